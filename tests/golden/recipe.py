"""Seeded synthetic-input recipes shared by make_golden.py (which feeds them to the reference) and by the
tests (which feed the very same arrays to the oracle / the HIP path).  numpy Generator streams are
bit-reproducible across machines, so large inputs need not be stored in the fixtures."""
import numpy as np
import torch


def make_item(rng, B, H, W, K, c_ev, c_img, with_depth=False, nan_frac=0.0):
    item = {}
    for k in range(K):
        item["events%d" % k] = torch.from_numpy(rng.standard_normal((B, c_ev, H, W)).astype(np.float32))
    item["image"] = torch.from_numpy(rng.random((B, c_img, H, W)).astype(np.float32))
    if with_depth:
        for key in ["image"] + ["events%d" % k for k in range(K)]:
            d = np.clip(1.0 + np.log(rng.uniform(0.02, 1.0, (B, 1, H, W))) / 3.70378, 0, 1).astype(np.float32)
            if nan_frac > 0:
                d[rng.random(d.shape) < nan_frac] = np.nan
            item["depth_" + key] = torch.from_numpy(d)
    return item



def synth_events(rng, n, W, H, t0=0.0, t1=0.05):
    ev = np.empty((n, 4), np.float64)
    ev[:, 0] = np.sort(rng.uniform(t0, t1, n))
    ev[:, 1] = rng.integers(0, W, n)
    ev[:, 2] = rng.integers(0, H, n)
    ev[:, 3] = rng.integers(0, 2, n)
    return ev




def make_dataset_dir(root, n_seq=2, n_frames=14, H=20, W=28, bins=5, seed=0):
    """A tiny EventScape-shaped directory tree (SURVEY 8f-4 file patterns): per sequence events/voxels/*_{idx:04d}_voxel.npy
    + timestamps.txt, depth/data/*_{idx:04d}_depth.npy + timestamps.txt, rgb/data/*_{idx:04d}_image.png.  Seeded, so the
    golden generator and the tests build byte-identical trees."""
    import os
    from PIL import Image
    rng = np.random.default_rng(seed)
    for s in range(n_seq):
        base = os.path.join(root, "Town%02d_sequence_%d" % (s + 1, s))
        ev, dp, im = (os.path.join(base, p) for p in ("events/voxels", "depth/data", "rgb/data"))
        for d in (ev, dp, im):
            os.makedirs(d, exist_ok=True)
        n = n_frames + 2 * s
        stamps = 12.5 + s + 0.04 * np.arange(n)
        table = np.stack([np.arange(n, dtype=np.float64), stamps], 1)
        np.savetxt(os.path.join(ev, "timestamps.txt"), table, fmt="%.6f")
        np.savetxt(os.path.join(dp, "timestamps.txt"), table, fmt="%.6f")
        for i in range(n):
            vox = rng.standard_normal((bins, H, W)).astype(np.float32)
            vox[rng.random((bins, H, W)) < 0.7] = 0.0                     # sparse, like event grids
            if i == 3:
                vox[:] = 0.0                                              # an empty grid: normalisation must leave it alone
            np.save(os.path.join(ev, "05_%03d_%04d_voxel.npy" % (s, i)), vox)
            depth = (rng.random((H, W)) ** 3 * 1500.0).astype(np.float32)  # metres, some beyond clip_distance = 1000
            depth[rng.random((H, W)) < 0.05] = np.nan
            np.save(os.path.join(dp, "05_%03d_%04d_depth.npy" % (s, i)), depth)
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            Image.fromarray(rgb).save(os.path.join(im, "05_%03d_%04d_image.png" % (s, i)))
    return root


FOLDERS = dict(event_folder="events/voxels", depth_folder="depth/data", frame_folder="rgb/data")
# (name, sequence index in sorted(os.listdir), item index, RNG seed, transform spec, dataset kwargs)
DATASET_CASES = [
    ("ramnet_center", 0, 0, 1, ("center", 16), dict(sequence_length=2, step_size=1, every_x_rgb_frame=3, clip_distance=1000.0,
                                                    reg_factor=5.70378, loss_composition=["image", "events2"])),
    ("ramnet_center_i2", 0, 2, 2, ("center", 16), dict(sequence_length=2, step_size=1, every_x_rgb_frame=3, clip_distance=1000.0,
                                                       reg_factor=5.70378, loss_composition=["image", "events2"])),
    ("ramnet_train_aug", 1, 1, 3, ("train", 16), dict(sequence_length=3, step_size=2, every_x_rgb_frame=2, clip_distance=1000.0,
                                                      reg_factor=5.70378, loss_composition=["image", "events1"])),
    ("ramnet_nocrop_scale", 0, 1, 4, None, dict(sequence_length=2, step_size=1, every_x_rgb_frame=2, clip_distance=80.0,
                                                reg_factor=3.70378, scale_factor=0.5, normalize=False)),
    ("baseline_rgb", 0, 1, 5, ("center", 16), dict(sequence_length=2, step_size=1, every_x_rgb_frame=3, clip_distance=1000.0,
                                                   baseline="rgb", loss_composition="image")),
    ("baseline_e", 0, 1, 6, ("center", 16), dict(sequence_length=2, step_size=1, every_x_rgb_frame=3, clip_distance=1000.0,
                                                 baseline="e", loss_composition="image")),
    ("baseline_ergb0", 0, 0, 7, ("center", 16), dict(sequence_length=3, step_size=1, every_x_rgb_frame=3, clip_distance=1000.0,
                                                     baseline="ergb0", loss_composition="image")),
    ("time_window", 1, 0, 8, ("center", 16), dict(sequence_length=2, step_size=1, every_x_rgb_frame=2, clip_distance=1000.0,
                                                  start_time=0.1, stop_time=0.45)),
]


def flatten_sequence(prefix, seq, out):
    for l, package in enumerate(seq):
        for k, v in package.items():
            out["%s/%d/%s" % (prefix, l, k)] = v.numpy() if v is not None else np.zeros(0, np.float32)
