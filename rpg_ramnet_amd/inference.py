"""Streaming inference and offline evaluation drivers: the counterparts of the hot loop of ``RAM_Net/test.py:205-395``
and of the file pairing / aggregation of ``RAM_Net/evaluation.py:295-397``.

test.py's contract: batch 1, ONE data package per model call, the recurrent state lives in the objects the
model returns and is fed back on the next call; it is reset whenever the dataset index of ``ConcatDatasetCustom``
increases (a new recording); the first two packages of every recording are not saved ("such that the temporal
dependencies of the network are settled").  Written files (the layout evaluation.py globs):

    <out>/npy/<key>/depth_{idx:010d}.npy                      prediction, normalised log depth, float32 [1, H, W]
    <out>/ground_truth/npy/depth_<key>/frame_{idx:010d}.npy   target, same encoding
"""
import glob
import os
from os.path import join

import numpy as np
import torch

from .metrics import depth_metrics


def empty_states(every_x_rgb_frame):
    """State dictionaries of a fresh recording (test.py:215-222)."""
    lstm = {}
    for k in range(every_x_rgb_frame):
        lstm['events{}'.format(k)] = None
        lstm['depth{}'.format(k)] = None
    lstm['image'] = None
    return {'image': None}, lstm


def _time_batched_ok(model):
    return (type(model).__name__ == "ERGB2DepthRecurrent" and not bool(model.baseline) and model.recurrent_block_type == "conv"
            and torch.device(model.gpu).type == "cuda")


def stream_dataset(model, dataset, every_x_rgb_frame, output_folder=None, settle=2, calculate_scale=False,
                   reg_factor=5.70378, clip_distance=1000.0, max_items=None, time_batched="auto"):
    """Run ``model`` over ``dataset`` (items ``(sequence, dataset_idx)``, sequence_length 1) the way test.py does and
    optionally write predictions / targets as .npy.  Returns {'items', 'saved', 'scale': (mean, min, max) or None}.

    time_batched ("auto" = whenever the model allows it): a data package IS one group of ``graph.TimeBatchedStream`` — its K event
    grids encode at batch K, its K + 1 decodes run as one chain, only the state updates are sequential — with the same outputs
    bit for bit; False calls ``model(package, ...)`` as test.py:230-232 does."""
    was_training = model.training
    model.eval()
    use_tb = _time_batched_ok(model) if time_batched == "auto" else bool(time_batched)
    tb, K = None, every_x_rgb_frame
    n = len(dataset) if max_items is None else min(len(dataset), max_items)
    scale = np.empty(n) if calculate_scale else None
    saved = 0
    with torch.no_grad():
        prev_dataset_idx = -1
        for idx in range(n):
            item, dataset_idx = dataset[idx]
            if dataset_idx > prev_dataset_idx:
                prev_super, prev_lstm = empty_states(every_x_rgb_frame)
                sequence_idx = 0
            package = {k: v[None, :] for k, v in item[0].items()}
            if use_tb:
                hw = tuple(package['image'].shape[2:])
                if tb is None or (tb.H, tb.W) != hw:                # a ConcatDatasetCustom may mix recordings of different resolution:
                    from .graph import TimeBatchedStream          # the runtime's static buffers are per resolution -> a new one
                    tb = TimeBatchedStream(model, 1, hw[0], hw[1], max_events=K + 1)
                if sequence_idx == 0:
                    tb.reset()                                    # a new recording starts from the zero state
                for k in range(K):
                    tb.push_events(package['events{}'.format(k)])
                out = tb.wait(tb.push_image(package['image']))    # [K + 1, 1, 1, H, W]: views of a static buffer, valid until the second-next
                #                                                   group of this shape (they are written to disk right below)
                preds = {'events{}'.format(k): out[k] for k in range(K)}
                preds['image'] = out[K]
                new_super, new_lstm = prev_super, prev_lstm        # (the runtime carries the state)
            else:
                preds, new_super, new_lstm = model(package, prev_super['image'], prev_lstm)
            if output_folder and sequence_idx >= settle:
                for key, img in preds.items():
                    d = join(output_folder, "npy", key)
                    os.makedirs(d, exist_ok=True)
                    np.save(join(d, 'depth_{:010d}.npy'.format(idx)), img[0].cpu().numpy())
                for key, value in package.items():
                    if 'depth' in key:
                        d = join(output_folder, "ground_truth/npy", key)
                        os.makedirs(d, exist_ok=True)
                        np.save(join(d, 'frame_{:010d}.npy'.format(idx)), value[0].cpu().numpy())
                saved += 1
            if calculate_scale:      # least-squares scale between metric prediction and target (test.py:365-378; last key wins there, too)
                for key, img in preds.items():
                    p = np.exp(reg_factor * (img[0][0].cpu().numpy() - np.float32(1.0))) * clip_distance
                    t = np.exp(reg_factor * (package['depth_' + key][0][0].cpu().numpy() - np.float32(1.0))) * clip_distance
                    scale[idx] = np.sum(p * t) / np.sum(p * p)
            prev_super, prev_lstm = new_super, new_lstm
            sequence_idx += 1
            prev_dataset_idx = dataset_idx
    model.train(was_training)
    return {"items": n, "saved": saved,
            "scale": (float(np.mean(scale)), float(np.min(scale)), float(np.max(scale))) if calculate_scale else None}


def evaluate_folders(predictions_dir, targets_dir, clip_distance, reg_factor, crop_ymax=None, prediction_offset=0,
                     target_offset=0, cutoffs=(10, 20, 30, 80, 250, 500), device="cuda:0"):
    """evaluation.py:295-397 on device: sorted ``*.npy`` of both folders are paired by position, rows ``[:crop_ymax]`` kept,
    both converted to metric depth, per-file metrics averaged over files — for all pixels and for every depth cut-off
    (keys ``"<cutoff>_<metric>"``).  Files without a valid pixel inside a cut-off are skipped for that cut-off."""
    p_files = sorted(glob.glob(join(predictions_dir, '*.npy')))[prediction_offset:]
    t_files = sorted(glob.glob(join(targets_dir, '*.npy')))[target_offset:]
    assert len(p_files) > 0 and len(t_files) > 0
    sums, counts = {}, {}
    for p_file, t_file in zip(p_files, t_files):
        t, p = np.load(t_file)[0], np.load(p_file)[0]
        if crop_ymax is not None:
            t, p = t[:crop_ymax], p[:crop_ymax]
        assert p.shape == t.shape
        pt, tt = torch.from_numpy(p).to(device), torch.from_numpy(t).to(device)
        for cut in (None,) + tuple(cutoffs):
            m = depth_metrics(pt, tt, clip_distance, reg_factor, cutoff=float("inf") if cut is None else float(cut))
            if m.get("n", 0) == 0:
                continue
            pre = "" if cut is None else "%d_" % cut
            for k, v in m.items():
                if k != "n":
                    sums[pre + k] = sums.get(pre + k, 0.0) + v
            counts[pre] = counts.get(pre, 0) + 1
    out = {k: v / counts[k.split("_")[0] + "_" if k.split("_")[0].isdigit() else ""] for k, v in sums.items()}
    out["files"] = len(p_files)
    return out
