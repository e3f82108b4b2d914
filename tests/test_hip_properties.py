"""GPU: size-independent properties at BASELINE.json's full sizes (256x344 crop of 346x260, batch 8, 200k events/grid),
where the CPU oracle would take minutes: homogeneity, batch-permutation equivariance, invariances, checksums."""
import numpy as np
import pytest
import torch

from recipe import make_item, synth_events
from util import assert_close, build_hip_model, ref_cfg

pytestmark = pytest.mark.gpu
H, W = 256, 344


def dev():
    return torch.device("cuda:0")


def test_full_size_conv_is_positively_homogeneous_and_shift_consistent():
    """relu(conv(s*x)) == s*relu(conv(x)) for s > 0 and zero bias, on a 256x344 map (21.5 x 32 tiles: partial tiles)."""
    from rpg_ramnet_amd.model.submodules import ConvLayer
    torch.manual_seed(0)
    m = ConvLayer(32, 64, 5, 2, 2).to(dev())
    with torch.no_grad():
        m.conv2d.bias.zero_()
        x = torch.randn(2, H, W, 32, device=dev())
        y1, y2 = m(x), m(3.0 * x)
        assert y1.shape == (2, H // 2, W // 2, 64)
        assert_close(y2.cpu().numpy(), (3.0 * y1).cpu().numpy(), 1e-5, "homogeneity")
        # translation by one output pixel (= 2 input pixels) away from the border
        xs = torch.roll(x, shifts=(2, 2), dims=(1, 2))
        ys = m(xs)
        # (a shift by one output pixel moves every pixel to another position of its Winograd tile — 2 x 4 since the encoders' views run the
        # F(2x4,3x3) kernel (round 5; column-transform coefficients up to 8): the two evaluations round differently, 1.9e-6 measured; 1e-6 held
        # for the 2 x 2 tiles of F(2x2))
        assert_close(ys[:, 4:-4, 4:-4].cpu().numpy(), torch.roll(y1, shifts=(1, 1), dims=(1, 2))[:, 4:-4, 4:-4].cpu().numpy(),
                     4e-6, "shift equivariance")


def test_full_size_network_is_batch_permutation_equivariant_and_deterministic():
    cfg, _ = ref_cfg("net_seeded_ramnet.npz", every_x_rgb_frame=2)
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
    rng = np.random.default_rng(0)
    item = make_item(rng, 4, H, W, 2, 5, 1)
    perm = torch.tensor([2, 0, 3, 1])
    lstm = {"events0": None, "events1": None, "image": None}
    with torch.no_grad():
        p1, s1, _ = model(item, None, lstm)
        p1b, _, _ = model(item, None, lstm)
        p2, s2, _ = model({k: v[perm] for k, v in item.items()}, None, lstm)
    for k in p1:
        assert torch.equal(p1[k], p1b[k]), "forward must be deterministic (no atomics on the forward path)"
        assert torch.equal(p1[k][perm], p2[k]), "batch elements must not interact"
        assert p1[k].shape == (4, 1, H, W) and float(p1[k].min()) >= 0.0 and float(p1[k].max()) <= 1.0
    for a, b in zip(s1["image"], s2["image"]):
        assert torch.equal(a[perm], b)


def test_full_size_si_loss_is_shift_invariant_and_weight_linear():
    """lambda = 1: SI(d + c) == SI(d); SI scales linearly with `weight`; NaN targets are ignored."""
    from rpg_ramnet_amd import ops
    g = torch.Generator(device=dev()).manual_seed(0)
    pred = torch.rand(8, 1, H, W, device=dev(), generator=g)
    tgt = torch.rand(8, 1, H, W, device=dev(), generator=g)
    tgt[torch.rand(8, 1, H, W, device=dev(), generator=g) < 0.2] = float("nan")
    l0 = float(ops.scale_invariant_loss(pred, tgt, 1.0, 1.0))
    l1 = float(ops.scale_invariant_loss(pred + 0.37, tgt, 1.0, 1.0))
    l2 = float(ops.scale_invariant_loss(pred, tgt, 2.5, 1.0))
    assert abs(l1 - l0) <= 1e-5 * abs(l0) + 1e-8
    assert abs(l2 - 2.5 * l0) <= 1e-6 * abs(l2)
    d = (pred - tgt)[~torch.isnan(tgt)].double()
    ref = float((d * d).mean() - d.mean() ** 2)
    assert abs(l0 - ref) <= 1e-5 * abs(ref)


def test_full_size_voxel_grid_checksums():
    """200k events -> 5 x 260 x 346: sum of the grid == sum of all valid votes, per-bin sums match, polarity symmetry."""
    from oracle import voxel_ref
    from rpg_ramnet_amd import voxel
    rng = np.random.default_rng(1)
    ev = synth_events(rng, 200000, 346, 260)
    g = voxel.events_to_voxel_grid(torch.from_numpy(ev).to(dev()), 5, 346, 260)
    il, vl, okl, ir, vr, okr = voxel_ref.voxel_votes(ev, 5, 346, 260)
    total = float(vl[okl].astype(np.float64).sum() + vr[okr].astype(np.float64).sum())
    assert abs(float(g.double().sum()) - total) < 1e-2
    per_bin = np.zeros(5)
    np.add.at(per_bin, il[okl] // (346 * 260), vl[okl].astype(np.float64))
    np.add.at(per_bin, ir[okr] // (346 * 260), vr[okr].astype(np.float64))
    np.testing.assert_allclose(g.double().sum(dim=(1, 2)).cpu().numpy(), per_bin, atol=1e-2)
    flipped = ev.copy()
    flipped[:, 3] = 1 - flipped[:, 3]
    g2 = voxel.events_to_voxel_grid(torch.from_numpy(flipped).to(dev()), 5, 346, 260)
    assert float((g + g2).abs().max()) < 1e-4                  # flipping every polarity negates the grid
    n = voxel.normalize_nonzero(g)
    nz = n[n != 0].double()
    assert abs(float(nz.mean())) < 1e-4 and abs(float(nz.std(unbiased=False)) - 1.0) < 1e-3
    assert torch.equal(n == 0, g == 0)
