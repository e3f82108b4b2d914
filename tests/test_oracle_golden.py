"""CPU: the oracle (oracle/) reproduces every golden vector generated from the reference itself."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref, ramnet_ref, voxel_ref
from recipe import make_item

TOL = dict(rtol=1e-5, atol=1e-6)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def sd_from(z):
    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}


def run_net(z, sd):
    cfg = json.loads(str(z["config"]))
    arch = str(z["arch"])
    fwd = ramnet_ref.forward_recurrent if arch == "ERGB2DepthRecurrent" else ramnet_ref.forward_unet
    seed, B, H, W, n_ev, c_ev, c_img, calls = [int(v) for v in z["recipe"]]
    rng = np.random.default_rng(seed)
    prev_super, prev_lstm = None, ramnet_ref.empty_states_lstm(cfg["every_x_rgb_frame"])
    results = []
    with torch.no_grad():
        for c in range(calls):
            item = make_item(rng, B, H, W, n_ev, c_ev, c_img)
            for k, v in item.items():           # the regenerated inputs ARE the stored ones
                if "in%d.%s" % (c, k) in z.files:
                    assert np.array_equal(v.numpy(), z["in%d.%s" % (c, k)])
            preds, supers, lstms = fwd(sd, cfg, item, prev_super, prev_lstm)
            results.append((preds, supers))
            prev_super, prev_lstm = supers["image"], lstms
    return results


SMALL = ["small_gru", "small_lstm", "small_gru_enclstm", "small_tconv", "small_base_rgb", "small_base_e",
         "small_base_ergb0", "small_unet", "small_unet_concat"]


@pytest.mark.parametrize("tag", SMALL)
def test_network_explicit_weights(golden_dir, tag):
    z = load(golden_dir, "net_%s.npz" % tag)
    res = run_net(z, sd_from(z))
    n = 0
    for c, (preds, supers) in enumerate(res):
        for k, v in preds.items():
            np.testing.assert_allclose(v.numpy(), z["pred%d.%s" % (c, k)], **TOL)
            n += 1
        for name in [f for f in z.files if f.startswith("super%d.image." % c)]:
            parts = name.split(".")
            s = supers["image"][int(parts[2])]
            s = s[{"h": 0, "c": 1}[parts[3]]] if len(parts) == 4 else s
            np.testing.assert_allclose(s.numpy(), z[name], **TOL)
            n += 1
    assert n >= 1


def seeded_sd(cfg, arch, z):
    """Seeded full-width weights: the model class reproduces the reference's torch.manual_seed(0) initialisation
    (checksums in the fixture prove it), so its CPU state_dict feeds the oracle."""
    from rpg_ramnet_amd.model import model as mm
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in getattr(mm, arch)(cfg).state_dict().items()}
    for k, v in sd.items():
        got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
        np.testing.assert_allclose(got, z["wsum." + k], rtol=1e-5, atol=1e-7, err_msg=k)
    return sd


@pytest.mark.parametrize("tag", ["seeded_ramnet", "seeded_ramnet_lstm", "seeded_base_rgb", "seeded_ramnet_bins10", "seeded_unet"])
def test_network_seeded_full_width(golden_dir, tag):
    """Released width (base 32) incl. the 10-bin configs[4] wiring: oracle vs the reference's predictions and states."""
    z = load(golden_dir, "net_%s.npz" % tag)
    cfg = json.loads(str(z["config"]))
    res = run_net(z, seeded_sd(cfg, str(z["arch"]), z))
    for c, (preds, supers) in enumerate(res):
        assert len(preds) >= 1
        for k, v in preds.items():
            np.testing.assert_allclose(v.numpy(), z["pred%d.%s" % (c, k)], **TOL)
        for name in [f for f in z.files if f.startswith("super%d.image." % c)]:
            parts = name.split(".")
            s = supers["image"][int(parts[2])]
            s = s[{"h": 0, "c": 1}[parts[3]]] if len(parts) == 4 else s
            np.testing.assert_allclose(s.numpy(), z[name], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["seeded_ramnet", "seeded_ramnet_bins10"])
def test_bptt_gradients_seeded_full_width(golden_dir, tag):
    """Full-width BPTT gradients through the reference trainer's loss assembly (norms + strided samples in the fixture)."""
    z = load(golden_dir, "grads_%s.npz" % tag)
    cfg = json.loads(str(z["config"]))
    sd = {k: v.requires_grad_(True) for k, v in seeded_sd(cfg, "ERGB2DepthRecurrent", z).items()}
    seq = []
    for l in range(int(z["L"])):
        pre = "in%d." % l
        seq.append({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
    total, _ = ramnet_ref.sequence_loss(sd, cfg, seq, cfg["loss_composition"], [1, 1])
    total.backward()
    np.testing.assert_allclose(len(cfg["loss_composition"]) * float(total.detach()), float(z["reported_loss"]), rtol=1e-5)
    n = 0
    for k, v in sd.items():
        if v.grad is None:
            continue
        if "g." + k in z.files:
            np.testing.assert_allclose(v.grad.numpy(), z["g." + k], rtol=2e-3, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_allclose(float(v.grad.double().norm()), float(z["gnorm." + k][0]), rtol=1e-3, err_msg=k)
            np.testing.assert_allclose(v.grad.flatten()[::997].numpy(), z["gsample." + k], rtol=2e-3,
                                       atol=1e-3 * float(np.abs(z["gsample." + k]).max()) + 1e-9, err_msg=k)
        n += 1
    assert n >= 60


def test_primitives(golden_dir):
    z = load(golden_dir, "primitives.npz")

    def w(name):
        return {k[len(name) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + ".w.")}

    def i(name, k):
        return torch.from_numpy(z["%s.in.%s" % (name, k)])

    def chk(name, k, v):
        np.testing.assert_allclose(v.numpy(), z["%s.out.%s" % (name, k)], **TOL)

    with torch.no_grad():
        pre = lambda d: {"L." + k: v for k, v in d.items()}  # noqa: E731
        chk("conv_s1", "y", ramnet_ref.conv_layer(pre(w("conv_s1")), "L", i("conv_s1", "x"), 1, 2))
        chk("conv_s2", "y", ramnet_ref.conv_layer(pre(w("conv_s2")), "L", i("conv_s2", "x"), 2, 2))
        chk("conv_s2_odd", "y", ramnet_ref.conv_layer(pre(w("conv_s2_odd")), "L", i("conv_s2_odd", "x"), 2, 2))
        chk("conv_1x1", "y", ramnet_ref.conv_layer(pre(w("conv_1x1")), "L", i("conv_1x1", "x"), 1, 0, relu=False))
        chk("upconv", "y", ramnet_ref.upsample_conv_layer(pre(w("upconv")), "L", i("upconv", "x")))
        chk("tconv", "y", ramnet_ref.transposed_conv_layer(pre(w("tconv")), "L", i("tconv", "x")))
        chk("resblock", "y", ramnet_ref.residual_block(pre(w("resblock")), "L", i("resblock", "x")))
        g = pre(w("convgru"))
        chk("convgru", "h", ramnet_ref.conv_gru(g, "L", i("convgru", "x"), i("convgru", "h")))
        chk("convgru", "h_from_none", ramnet_ref.conv_gru(g, "L", i("convgru", "x"), None))
        l = pre(w("convlstm"))
        h, c = ramnet_ref.conv_lstm(l, "L", i("convlstm", "x"), (i("convlstm", "h"), i("convlstm", "c")))
        chk("convlstm", "h", h)
        chk("convlstm", "c", c)
        h, c = ramnet_ref.conv_lstm(l, "L", i("convlstm", "x"), None)
        chk("convlstm", "h_from_none", h)
        chk("convlstm", "c_from_none", c)


@pytest.mark.parametrize("tag", ["small_gru", "small_lstm"])
def test_bptt_gradients_through_trainer_loss(golden_dir, tag):
    z = load(golden_dir, "grads_%s.npz" % tag)
    cfg = json.loads(str(z["config"]))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_from(z).items()}
    L = int(z["L"])
    seq = []
    for l in range(L):
        pre = "in%d." % l
        seq.append({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
    total, _ = ramnet_ref.sequence_loss(sd, cfg, seq, cfg["loss_composition"], [1, 1])
    total.backward()
    # reported loss = (#keys) x the differentiated scalar (aliasing quirk, lstm_trainer.py:281,210-224)
    np.testing.assert_allclose(len(cfg["loss_composition"]) * float(total.detach()), float(z["reported_loss"]), rtol=1e-5)
    for k, v in sd.items():
        np.testing.assert_allclose(v.grad.numpy(), z["g." + k], rtol=2e-4, atol=1e-7, err_msg=k)


def test_si_loss_and_metrics(golden_dir):
    z = load(golden_dir, "loss_metrics.npz")
    for i in range(3):
        p, t = torch.from_numpy(z["si%d.pred" % i]), torch.from_numpy(z["si%d.target" % i])
        np.testing.assert_allclose(loss_ref.scale_invariant_loss(p, t).numpy(), z["si%d.loss" % i], rtol=1e-5)
        np.testing.assert_allclose(loss_ref.scale_invariant_loss(p, t, 0.5, 0.85).numpy(),
                                   z["si%d.loss_w05_l085" % i], rtol=1e-5)
        np.testing.assert_allclose(loss_ref.si_loss_grad(p, t).numpy(), z["si%d.grad" % i], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(loss_ref.mse_loss(p, t).numpy(), z["si%d.mse" % i], rtol=1e-5)
        for fn in ["abs_rel_diff", "squ_rel_diff", "rms_linear", "scale_invariant_error", "mean_error",
                   "median_error"]:
            np.testing.assert_allclose(getattr(loss_ref, fn)(p.numpy(), t.numpy()), z["si%d.%s" % (i, fn)],
                                       rtol=1e-6)
    for clip, reg in [(80, 3.70378), (1000, 5.70378)]:
        t2, p2 = loss_ref.prepare_depth_data(z["depth%d.target_in" % clip], z["depth%d.pred_in" % clip],
                                             float(clip), reg)
        np.testing.assert_allclose(t2, z["depth%d.target" % clip], rtol=1e-6)
        np.testing.assert_allclose(p2, z["depth%d.pred" % clip], rtol=1e-6)
        np.testing.assert_allclose(loss_ref.abs_rel_diff(p2, t2), z["depth%d.abs_rel" % clip], rtol=1e-6)


def test_evaluation_table_vs_reference_add_to_metrics(golden_dir):
    """oracle evaluation_table == the reference's add_to_metrics (evaluation.py:201-241) on its seeded maps: ten rows, NaN rows NaN."""
    z = load(golden_dir, "eval_metrics.npz")
    for tag in ("plain", "cut30", "nan", "nan_cut20", "mvsec"):
        clip, reg, cutoff = (float(v) for v in z[tag + ".params"])
        t, p = loss_ref.prepare_depth_data(z[tag + ".target_in"], z[tag + ".pred_in"], clip, reg)
        got = loss_ref.evaluation_table(t, p, np.nan_to_num(t) < cutoff)
        assert len(got) == 10
        for k, v in got.items():
            np.testing.assert_allclose(v, float(z["%s.%s" % (tag, k)]), rtol=1e-6, err_msg="%s %s" % (tag, k), equal_nan=True)
        assert np.isnan(got["median_diff"]) == np.isnan(got["RMS_log"]) == tag.startswith("nan")


def test_log_and_mse_losses_and_trainer_assembly(golden_dir):
    """scale_invariant_log_loss (model/loss.py:12-15), mse_loss at full / half resolution and the trainer's assembly with the mse
    term (lstm_trainer.py:152-226) against vectors produced by the reference's own functions and LSTMTrainer methods
    (tests/golden/make_golden_losses.py); values and gradients (autograd through the restatement)."""
    z = load(golden_dir, "loss_extra.npz")
    for i in range(3):
        t = torch.from_numpy(z["c%d.target" % i])
        for lam in (100, 85):
            p = torch.from_numpy(z["c%d.pred" % i]).requires_grad_(True)
            l = loss_ref.scale_invariant_log_loss(p, t, lam / 100.0)
            l.backward()
            np.testing.assert_allclose(l.detach().numpy(), z["c%d.silog%d.loss" % (i, lam)], rtol=1e-5)
            np.testing.assert_allclose(p.grad.numpy(), z["c%d.silog%d.grad" % (i, lam)], rtol=1e-4, atol=1e-8)
        for f in (100, 50):
            p = torch.from_numpy(z["c%d.pred" % i]).requires_grad_(True)
            l = loss_ref.mse_loss_downsampled(p, t, f / 100.0)
            l.backward()
            np.testing.assert_allclose(l.detach().numpy(), z["c%d.mse%d.loss" % (i, f)], rtol=1e-6)
            np.testing.assert_allclose(p.grad.numpy(), z["c%d.mse%d.grad" % (i, f)], rtol=1e-5, atol=1e-10)
    preds = [torch.from_numpy(z["asm.pred%d" % l]).requires_grad_(True) for l in range(2)]
    tgts = [torch.from_numpy(z["asm.target%d" % l]) for l in range(2)]
    total = loss_ref.total_batch_loss(preds, tgts, [float(w) for w in z["asm.weights"]], 2, loss_params={"weight": 1.0, "n_lambda": 1.0},
                                      mse={"weight": float(z["asm.mse_weight"]), "downsampling_factor": float(z["asm.mse_factor"])})
    total.backward()
    np.testing.assert_allclose(total.detach().numpy(), z["asm.loss"], rtol=1e-5)
    for l in range(2):
        np.testing.assert_allclose(preds[l].grad.numpy(), z["asm.grad%d" % l], rtol=1e-4, atol=1e-9)


VOX = ["rand", "rand10", "onebin", "single", "same_t", "corners", "int_ts_pm1"]


@pytest.mark.parametrize("name", VOX)
def test_voxel_grid(golden_dir, name):
    z = load(golden_dir, "voxel.npz")
    ev = z["%s.events" % name]
    bins, W, H = [int(v) for v in z["%s.dims" % name]]
    keep = ev.copy()
    g = voxel_ref.events_to_voxel_grid(ev, bins, W, H)
    assert np.array_equal(ev, keep)                      # no in-place mutation of the caller's array
    ref = z["%s.grid_torch" % name]
    # same float32 votes accumulated in the same order -> bit-exact against the reference's CPU index_add_
    assert np.array_equal(g, ref), np.abs(g - ref).max()
    assert np.array_equal(g != 0, ref != 0)
    np.testing.assert_allclose(g, z["%s.grid_numpy" % name], atol=2e-6)
    # the oracle's int64 indices == the reference's own (what its two index_add_ calls received: make_golden_voxel_indices.py)
    il, _, okl, ir, _, okr = voxel_ref.voxel_votes(ev, bins, W, H)
    zi = load(golden_dir, "voxel_indices.npz")
    assert np.array_equal(il[okl], zi["%s.ref_idx_left" % name]) and np.array_equal(ir[okr], zi["%s.ref_idx_right" % name])


def test_voxel_normalisation(golden_dir):
    z = load(golden_dir, "voxel.npz")
    n = voxel_ref.normalize_nonzero(z["rand.grid_torch"])
    np.testing.assert_allclose(n, z["rand.normalized"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(voxel_ref.normalize_nonzero(np.zeros((2, 4, 4), np.float32)), z["zeros.normalized"])


def test_msg_loss_restatement_is_consistent():
    """multi_scale_grad_loss is PARITY UNPINNED (kornia absent): only self-consistency is checked."""
    rng = np.random.default_rng(0)
    p = torch.from_numpy(rng.random((2, 1, 32, 32)).astype(np.float32))
    t = p.clone()
    assert float(loss_ref.multi_scale_grad_loss(p, t)) == 0.0
    ramp = torch.arange(32, dtype=torch.float32)[None, None, None, :].expand(2, 1, 32, 32).contiguous()
    g = loss_ref.spatial_gradient(ramp)
    assert torch.allclose(g[:, :, 0, :, 1:-1], torch.ones(2, 1, 32, 30))     # d/dx of a unit ramp
    assert torch.allclose(g[:, :, 1], torch.zeros(2, 1, 32, 32))


# ------------------------------------------------------------------------------------------------ norm 'BN' / 'IN'
NORM_NETS = ["small_gru_bn", "small_gru_in", "small_lstm_tconv_bn", "small_unet_bn", "small_unet_in"]


def norm_items(z, c):
    pre = "in%d." % c
    return {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}


def is_buffer(k):
    return k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")


@pytest.mark.parametrize("tag", NORM_NETS)
def test_norm_layers_training_and_eval_calls(golden_dir, tag):
    """BatchNorm / InstanceNorm variants (submodules.py:13-24, 29-30, 188-193): two training-mode calls (batch / per-image statistics,
    running buffers updated in place) and one eval-mode call (running statistics) of the reference, predictions and buffers."""
    z = load(golden_dir, "norm_%s.npz" % tag)
    cfg = json.loads(str(z["config"]))
    arch, sd = str(z["arch"]), {k: v.clone() for k, v in sd_from(z).items()}
    fwd = ramnet_ref.forward_recurrent if arch == "ERGB2DepthRecurrent" else ramnet_ref.forward_unet
    prev_super, prev_lstm = None, ramnet_ref.empty_states_lstm(cfg["every_x_rgb_frame"])
    ncalls = int(z["train_calls"]) + 1
    with torch.no_grad():
        for c in range(ncalls):
            preds, supers, lstms = fwd(sd, dict(cfg, training=c < ncalls - 1), norm_items(z, c), prev_super, prev_lstm)
            for k, v in preds.items():
                np.testing.assert_allclose(v.numpy(), z["pred%d.%s" % (c, k)], rtol=1e-4, atol=2e-6, err_msg="call %d %s" % (c, k))
            for k in [f for f in z.files if f.startswith("buf%d." % c)]:
                np.testing.assert_allclose(sd[k[5:]].numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)
            if arch == "ERGB2DepthRecurrent":
                prev_super, prev_lstm = supers["image"], lstms
    assert any(is_buffer(k) for k in sd) or cfg["norm"] == "IN"


@pytest.mark.parametrize("tag", ["small_gru_bn", "small_gru_in"])
def test_norm_layers_bptt_gradients(golden_dir, tag):
    z = load(golden_dir, "norm_grads_%s.npz" % tag)
    cfg = json.loads(str(z["config"]))
    sd = {k: (v.clone() if is_buffer(k) else v.clone().requires_grad_(True)) for k, v in sd_from(z).items()}
    seq = [norm_items(z, l) for l in range(int(z["L"]))]
    total, _ = ramnet_ref.sequence_loss(sd, dict(cfg, training=True), seq, cfg["loss_composition"], [1, 1])
    total.backward()
    np.testing.assert_allclose(len(cfg["loss_composition"]) * float(total.detach()), float(z["reported_loss"]), rtol=1e-5)
    n = 0
    for k, v in sd.items():
        if is_buffer(k):
            np.testing.assert_allclose(v.numpy(), z["buf." + k], rtol=1e-5, atol=1e-6, err_msg=k)
            continue
        want = z["g." + k]
        got = v.grad.numpy()          # (only the REPORTED loss carries the aliased-dict factor, lstm_trainer.py:281)
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=1e-5 * max(1.0, float(np.abs(want).max())), err_msg=k)
        n += 1
    assert n > 20
