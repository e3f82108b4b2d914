#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference.

Run only in the build container (where /root/reference exists):
    python tests/golden/make_golden.py
The fixtures (.npz = inputs + reference outputs) are committed; the reference never travels.
Nothing here copies reference source: the reference modules are imported and called.

Shims needed to import the reference in this image:
  torch.utils.tensorboard (absent), kornia.filters.sobel (absent), skimage.measure (absent),
  torchvision / cv2 (absent), np.int (removed alias).  ``model.gpu`` is overwritten with the CPU device.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from recipe import make_item, synth_events  # noqa: E402  (shared with the tests)
REF = "/root/reference/RAM_Net"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    sys.path.insert(0, REF)
    _stub("torch.utils.tensorboard", SummaryWriter=object)
    _stub("kornia")
    _stub("kornia.filters")
    _stub("kornia.filters.sobel", spatial_gradient=None, sobel=None)
    _stub("skimage")
    _stub("skimage.measure", compare_ssim=None)
    _stub("torchvision", utils=types.SimpleNamespace())
    _stub("cv2")
    _stub("tqdm", tqdm=lambda x, **k: x)
    if not hasattr(np, "int"):
        np.int = int
    import model.model as mm
    import model.submodules as sub
    import model.loss as loss
    import model.metric as metric
    import utils.event_tensor_utils as etu
    import trainer.lstm_trainer as lt
    import evaluation as ev  # noqa: F401  (argparse only runs under __main__)
    return mm, sub, loss, metric, etu, lt, ev


def load_cfg(name):
    with open(os.path.join(REF, "configs", name)) as f:
        return json.load(f)


def model_cfg(full, **over):
    c = dict(full["model"])
    c["gpu"] = 0
    c["every_x_rgb_frame"] = full["data_loader"]["train"]["every_x_rgb_frame"]
    c["baseline"] = full["data_loader"]["train"]["baseline"]
    c["loss_composition"] = full["trainer"]["loss_composition"]
    c.update(over)
    return c


def build(mm, arch, cfg, seed=0):
    torch.manual_seed(seed)
    m = getattr(mm, arch)(cfg)
    m.gpu = torch.device("cpu")
    return m


def empty_lstm(K):
    d = {"image": None}
    for k in range(K):
        d["events%d" % k] = None
        d["depth%d" % k] = None
    return d


def flat_states(prefix, supers, out):
    for key, lst in supers.items():
        if lst is None:
            continue
        for i, s in enumerate(lst):
            if isinstance(s, (list, tuple)):
                out["%s.%s.%d.h" % (prefix, key, i)] = s[0].detach().numpy()
                out["%s.%s.%d.c" % (prefix, key, i)] = s[1].detach().numpy()
            else:
                out["%s.%s.%d" % (prefix, key, i)] = s.detach().numpy()


def weight_checksums(sd):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------
def gen_network(mm, tag, arch, cfg, B, H, W, store_weights, seed_inputs=1, calls=2, lean=False):
    """Two consecutive forward calls (state carry)."""
    m = build(mm, arch, cfg).eval()
    K = cfg["every_x_rgb_frame"]
    baseline, lc = cfg["baseline"], cfg["loss_composition"]
    as_img = baseline == "ergb0" or (baseline == "e" and lc == "image")
    n_ev = 0 if (baseline == "rgb" or arch == "ERGB2Depth") else (K - 1 if as_img else K)
    c_ev = cfg["num_bins_rgb"] if as_img else cfg["num_bins_events"]
    rng = np.random.default_rng(seed_inputs)
    out = {"config": np.array(json.dumps(cfg)), "arch": np.array(arch),
           "recipe": np.array([seed_inputs, B, H, W, n_ev, c_ev, cfg["num_bins_rgb"], calls])}
    sd = m.state_dict()
    if store_weights:
        for k, v in sd.items():
            out["w." + k] = v.numpy()
    else:
        for k, v in weight_checksums(sd).items():
            out["wsum." + k] = v
    prev_super, prev_lstm = None, empty_lstm(K)
    with torch.no_grad():
        for c in range(calls):
            item = make_item(rng, B, H, W, n_ev, c_ev, cfg["num_bins_rgb"])
            for k, v in ({} if lean else item).items():   # lean: inputs are regenerated from "recipe"
                out["in%d.%s" % (c, k)] = v.numpy()
            preds, supers, lstms = m(item, prev_super, prev_lstm)
            for k, v in preds.items():
                out["pred%d.%s" % (c, k)] = v.numpy()
            if arch == "ERGB2DepthRecurrent":
                if not lean:
                    flat_states("super%d" % c, {"image": supers["image"]}, out)
                prev_super, prev_lstm = supers["image"], lstms
    np.savez_compressed(os.path.join(HERE, "net_%s.npz" % tag), **out)
    print("net_%s.npz" % tag, "params", sum(p.numel() for p in m.parameters()))


def gen_grads(mm, lt, loss_mod, tag, cfg, B, H, W, L, store_weights, nan_frac):
    """BPTT over L packages through the reference's own LSTMTrainer.forward_pass_sequence (SI loss only:
    the multi-scale gradient loss needs kornia, absent here)."""
    m = build(mm, "ERGB2DepthRecurrent", cfg).train()
    K = cfg["every_x_rgb_frame"]
    rng = np.random.default_rng(7)
    seq = [make_item(rng, B, H, W, K, cfg["num_bins_events"], cfg["num_bins_rgb"], True, nan_frac) for _ in range(L)]
    t = object.__new__(lt.LSTMTrainer)
    t.model, t.loss, t.loss_params = m, loss_mod.scale_invariant_loss, {"weight": 1.0, "n_lambda": 1.0}
    t.metrics, t.calculate_total_metrics = [], []
    t.every_x_rgb_frame, t.loss_composition, t.loss_weights = K, cfg["loss_composition"], [1, 1]
    t.gpu, t.baseline, t.state_combination = torch.device("cpu"), cfg["baseline"], cfg["state_combination"]
    t.use_grad_loss, t.use_mse_loss, t.state_preview_flag = False, False, False
    losses = t.forward_pass_sequence(seq)[0]
    m.zero_grad()
    losses["loss"].backward()
    out = {"config": np.array(json.dumps(cfg)), "reported_loss": losses["loss"].detach().numpy(),
           "L": np.array(L)}
    for l, item in enumerate(seq):
        for k, v in item.items():
            out["in%d.%s" % (l, k)] = v.numpy()
    sd = m.state_dict()
    for k, v in (sd.items() if store_weights else []):
        out["w." + k] = v.numpy()
    if not store_weights:
        for k, v in weight_checksums(sd).items():
            out["wsum." + k] = v
    for k, p in m.named_parameters():
        g = p.grad
        if store_weights or g.numel() <= 4096:
            out["g." + k] = g.numpy()
        else:  # full-width model: keep norms + a strided sample (small fixture)
            out["gnorm." + k] = np.array([float(g.double().norm()), float(g.double().sum())])
            out["gsample." + k] = g.flatten()[::997].numpy()
    np.savez_compressed(os.path.join(HERE, "grads_%s.npz" % tag), **out)
    print("grads_%s.npz" % tag, "loss", float(losses["loss"]))


def gen_primitives(sub):
    rng = np.random.default_rng(3)
    torch.manual_seed(3)
    out = {}

    def t(*s):
        return torch.from_numpy(rng.standard_normal(s).astype(np.float32))

    def put(name, mod, outs, **ins):
        for k, v in mod.state_dict().items():
            out["%s.w.%s" % (name, k)] = v.numpy()
        for k, v in ins.items():
            out["%s.in.%s" % (name, k)] = v.numpy()
        for k, v in outs.items():
            out["%s.out.%s" % (name, k)] = v.detach().numpy()

    with torch.no_grad():
        x = t(2, 5, 12, 14)
        m = sub.ConvLayer(5, 8, 5, 1, 2)
        put("conv_s1", m, {"y": m(x)}, x=x)
        x = t(2, 8, 12, 14)
        m = sub.ConvLayer(8, 16, 5, 2, 2)
        put("conv_s2", m, {"y": m(x)}, x=x)
        x = t(2, 8, 11, 13)      # odd sizes: (H-1)//2+1 output
        put("conv_s2_odd", m, {"y": m(x)}, x=x)
        m = sub.ConvLayer(8, 1, 1, activation=None)
        put("conv_1x1", m, {"y": m(x)}, x=x)
        x = t(2, 8, 6, 7)
        m = sub.UpsampleConvLayer(8, 4, 5, padding=2)
        put("upconv", m, {"y": m(x)}, x=x)
        m = sub.TransposedConvLayer(8, 4, 5, padding=2)
        put("tconv", m, {"y": m(x)}, x=x)
        m = sub.ResidualBlock(8, 8)
        put("resblock", m, {"y": m(x.clone())}, x=x)
        h = t(2, 8, 6, 7)
        m = sub.ConvGRU(8, 8, 3)
        put("convgru", m, {"h": m(x, h), "h_from_none": m(x, None)}, x=x, h=h)
        c = t(2, 8, 6, 7)
        m = sub.ConvLSTM(8, 8, 3)
        h1, c1 = m(x, (h, c))
        h0, c0 = m(x, None)
        put("convlstm", m, {"h": h1, "c": c1, "h_from_none": h0, "c_from_none": c0}, x=x, h=h, c=c)
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **out)
    print("primitives.npz")


def gen_loss_metrics(loss_mod, metric, ev):
    rng = np.random.default_rng(11)
    out = {}
    for i, nan_frac in enumerate([0.0, 0.2, 0.9]):
        p = rng.random((3, 1, 20, 24)).astype(np.float32)
        tg = rng.random((3, 1, 20, 24)).astype(np.float32)
        tg[rng.random(tg.shape) < nan_frac] = np.nan
        pt = torch.from_numpy(p).requires_grad_(True)
        l = loss_mod.scale_invariant_loss(pt, torch.from_numpy(tg), 1.0, 1.0)
        l.backward()
        l2 = loss_mod.scale_invariant_loss(torch.from_numpy(p), torch.from_numpy(tg), 0.5, 0.85)
        out["si%d.pred" % i], out["si%d.target" % i] = p, tg
        out["si%d.loss" % i], out["si%d.grad" % i] = l.detach().numpy(), pt.grad.numpy()
        out["si%d.loss_w05_l085" % i] = l2.numpy()
        out["si%d.mse" % i] = loss_mod.mse_loss(torch.from_numpy(p), torch.from_numpy(tg)).numpy()
        for fn in ["abs_rel_diff", "squ_rel_diff", "rms_linear", "scale_invariant_error", "mean_error",
                   "median_error"]:
            out["si%d.%s" % (i, fn)] = np.array(getattr(metric, fn)(p, tg))
    # metric depth from normalised log depth (evaluation.py:74-96)
    tg = rng.random((20, 24)).astype(np.float32)
    pr = rng.random((20, 24)).astype(np.float32)
    for clip, reg in [(80.0, 3.70378), (1000.0, 5.70378)]:
        t2, p2 = ev.prepare_depth_data(tg.copy(), pr.copy(), clip, reg_factor=reg)
        out["depth%d.target_in" % int(clip)], out["depth%d.pred_in" % int(clip)] = tg, pr
        out["depth%d.target" % int(clip)], out["depth%d.pred" % int(clip)] = t2, p2
        out["depth%d.abs_rel" % int(clip)] = np.array(metric.abs_rel_diff(p2, t2))
    np.savez_compressed(os.path.join(HERE, "loss_metrics.npz"), **out)
    print("loss_metrics.npz")


def gen_voxel(etu):
    rng = np.random.default_rng(5)
    out = {}
    cases = {
        "rand": (synth_events(rng, 20000, 346, 260), 5, 346, 260),
        "rand10": (synth_events(rng, 5000, 64, 48), 10, 64, 48),
        "onebin": (synth_events(rng, 1000, 32, 24), 1, 32, 24),
        "single": (np.array([[0.5, 3, 2, 1]], np.float64), 5, 8, 6),
        "same_t": (np.array([[1.0, 0, 0, 0], [1.0, 7, 5, 1], [1.0, 7, 5, 0]], np.float64), 3, 8, 6),
        "corners": (np.array([[0.0, 0, 0, 1], [0.25, 7, 0, 0], [0.5, 0, 5, 1], [1.0, 7, 5, 1]], np.float64),
                    5, 8, 6),
    }
    # polarity given as -1/+1 and exact-integer normalised timestamps
    e = synth_events(rng, 2000, 40, 30)
    e[:, 3] = e[:, 3] * 2 - 1
    e[:, 0] = np.round(e[:, 0] / 0.05 * 4) / 4 * 0.05
    e[0, 0], e[-1, 0] = 0.0, 0.05
    cases["int_ts_pm1"] = (e, 5, 40, 30)
    dev = torch.device("cpu")
    etu.CudaTimer = etu.Timer          # shim: the reference times normalisation with CUDA events
    for name, (evs, bins, W, H) in cases.items():
        g_t = etu.events_to_voxel_grid_pytorch(evs.copy(), bins, W, H, dev).numpy()
        g_n = etu.events_to_voxel_grid(evs.copy(), bins, W, H)
        out["%s.events" % name] = evs
        out["%s.dims" % name] = np.array([bins, W, H])
        out["%s.grid_torch" % name] = g_t
        out["%s.grid_numpy" % name] = g_n
    # normalisation (EventPreprocessor.__call__, event_tensor_utils.py:52-66)
    opts = types.SimpleNamespace(no_normalize=False, hot_pixels_file=None, flip=False)
    pre = etu.EventPreprocessor(opts)
    g = torch.from_numpy(out["rand.grid_torch"])[None]
    out["rand.normalized"] = pre(g)[0].numpy()
    z = torch.zeros(1, 2, 4, 4)
    out["zeros.normalized"] = pre(z)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "voxel.npz"), **out)
    print("voxel.npz")


def main():
    mm, sub, loss_mod, metric, etu, lt, ev = import_reference()
    ramnet = load_cfg("train_e2depth_si_grad_loss_statenet_ergb.json")
    lc = dict(loss_composition=["image", "events1"])   # K=2: supervise the frame and the last event grid
    # (4) BASELINE.json configs[4] wiring: 10-bin event voxel grids (the head layer then has 10 input channels)
    gen_network(mm, "seeded_ramnet_bins10", "ERGB2DepthRecurrent", model_cfg(ramnet, num_bins_events=10, every_x_rgb_frame=2),
                1, 32, 48, False)
    gen_grads(mm, lt, loss_mod, "seeded_ramnet_bins10", model_cfg(ramnet, num_bins_events=10, every_x_rgb_frame=2, **lc),
              2, 16, 24, 2, False, 0.2)
    if "--only-bins10" in sys.argv:
        return
    b_e = load_cfg("train_e2depth_si_grad_loss_statenet_baseline_e.json")
    b_ergb = load_cfg("train_e2depth_si_grad_loss_statenet_baseline_ergb.json")
    b_rgb = load_cfg("train_e2depth_si_grad_loss_statenet_baseline_rgb.json")
    b_nr = load_cfg("train_e2depth_si_grad_loss_statenet_baseline_ergb_no_recurrent.json")

    # (1) explicit-weight, narrow (base_num_channels=4) variants: pin the oracle independent of RNG
    small = dict(base_num_channels=4, every_x_rgb_frame=2)
    gen_network(mm, "small_gru", "ERGB2DepthRecurrent", model_cfg(ramnet, **small), 2, 24, 32, True)
    gen_network(mm, "small_lstm", "ERGB2DepthRecurrent",
                model_cfg(ramnet, state_combination="convlstm", **small), 2, 24, 32, True)
    gen_network(mm, "small_gru_enclstm", "ERGB2DepthRecurrent",
                model_cfg(ramnet, recurrent_block_type="convlstm", **small), 2, 24, 32, True)
    gen_network(mm, "small_tconv", "ERGB2DepthRecurrent",
                model_cfg(ramnet, use_upsample_conv=False, **small), 2, 24, 32, True)
    gen_network(mm, "small_base_rgb", "ERGB2DepthRecurrent", model_cfg(b_rgb, **small), 2, 24, 32, True)
    gen_network(mm, "small_base_e", "ERGB2DepthRecurrent", model_cfg(b_e, base_num_channels=4,
                                                                      every_x_rgb_frame=3), 2, 24, 32, True)
    gen_network(mm, "small_base_ergb0", "ERGB2DepthRecurrent", model_cfg(b_ergb, base_num_channels=4,
                                                                          every_x_rgb_frame=3), 2, 24, 32, True)
    gen_network(mm, "small_unet", "ERGB2Depth", model_cfg(b_nr, base_num_channels=4), 2, 32, 32, True, calls=1)
    gen_network(mm, "small_unet_concat", "ERGB2Depth",
                model_cfg(b_nr, base_num_channels=4, skip_type="concat"), 2, 32, 32, True, calls=1)

    # (2) released width (base 32), seeded init (torch.manual_seed(0)); weights identified by checksums
    gen_network(mm, "seeded_ramnet", "ERGB2DepthRecurrent", model_cfg(ramnet), 1, 32, 48, False)
    gen_network(mm, "seeded_ramnet_lstm", "ERGB2DepthRecurrent",
                model_cfg(ramnet, state_combination="convlstm", every_x_rgb_frame=2), 1, 32, 48, False)
    gen_network(mm, "seeded_base_rgb", "ERGB2DepthRecurrent", model_cfg(b_rgb), 1, 32, 48, False)
    gen_network(mm, "seeded_unet", "ERGB2Depth", model_cfg(b_nr), 1, 32, 48, False, calls=1)
    # BASELINE.json configs[0]: single 256x256 frame + 1 event bin-grid forward (plumbing case)
    gen_network(mm, "config1_256", "ERGB2DepthRecurrent", model_cfg(ramnet, every_x_rgb_frame=1),
                1, 256, 256, False, lean=True)

    # (3) BPTT gradients through the reference trainer's loss assembly
    gen_grads(mm, lt, loss_mod, "small_gru", model_cfg(ramnet, **small, **lc), 2, 24, 32, 2, True, 0.2)
    gen_grads(mm, lt, loss_mod, "small_lstm", model_cfg(ramnet, state_combination="convlstm", **small, **lc),
              2, 24, 32, 2, True, 0.2)
    gen_grads(mm, lt, loss_mod, "seeded_ramnet", model_cfg(ramnet, every_x_rgb_frame=2, **lc), 2, 16, 24, 2, False, 0.2)

    gen_primitives(sub)
    gen_loss_metrics(loss_mod, metric, ev)
    gen_voxel(etu)


if __name__ == "__main__":
    main()
