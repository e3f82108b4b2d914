"""GPU: every fused HIP operator (forward, backward-data, backward-weights) against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref, ramnet_ref, voxel_ref
from rpg_ramnet_amd import _hip as Hh
from util import assert_close, load_golden, nchw, nhwc
import torch_restatements as tr

pytestmark = pytest.mark.gpu
TOL = 2e-4   # exact-fp32 MFMA vs PyTorch CPU; summation order differs


def dev():
    return torch.device("cuda:0")


def run_pair(module, oracle_fn, inputs, seed=0, grad_floor=0.0):
    """module: rpg_ramnet_amd layer (NHWC, cuda); oracle_fn(sd, *nchw_cpu_inputs) -> tensor or tuple.
    grad_floor: parameter gradients below this fraction of the largest one count as cancellation noise (a bias in front of a
    normalisation has an analytically ZERO gradient: both sides return rounding residue)."""
    module = module.to(dev())
    params = dict(module.named_parameters())          # (buffers — the running statistics of norm layers — carry no gradient)
    sd = {k: (v.detach().cpu().clone().requires_grad_(True) if k in params else v.detach().cpu().clone())
          for k, v in module.state_dict().items()}
    cpu_in = [None if t is None else t.clone().requires_grad_(True) for t in inputs]
    gpu_in = [None if t is None else nhwc(t).to(dev()).requires_grad_(True) for t in inputs]
    ref = oracle_fn(sd, *cpu_in)
    got = module(*gpu_in)
    ref = ref if isinstance(ref, tuple) else (ref,)
    got = got if isinstance(got, tuple) else (got,)
    g = torch.Generator().manual_seed(seed)
    loss_ref_, loss_got = 0, 0
    for r, o in zip(ref, got):
        o_nchw = o if o.shape == r.shape else o.permute(0, 3, 1, 2)
        assert_close(o_nchw.detach().cpu().numpy(), r.detach().numpy(), TOL, "forward")
        wgt = torch.randn(r.shape, generator=g)
        # an output that one side clips to exactly 0 (ReLU) and the other leaves a rounding error above it sits on the kink of
        # the activation: its derivative is 0 on one side and 1 on the other, so it gets no upstream gradient on either
        kink = (o_nchw.detach().cpu() == 0) != (r.detach() == 0)
        assert float(kink.float().mean()) < 1e-4
        wgt[kink] = 0.0
        loss_ref_ = loss_ref_ + (r * wgt).sum()
        loss_got = loss_got + (o_nchw * wgt.to(dev())).sum()
    loss_ref_.backward()
    loss_got.backward()
    for i, (c, gi) in enumerate(zip(cpu_in, gpu_in)):
        if c is not None and c.grad is not None:
            assert gi.grad is not None, "missing input grad %d" % i
            assert_close(nchw(gi.grad).cpu().numpy(), c.grad.numpy(), TOL, "grad input %d" % i)
    gmax = max(float(sd[k].grad.abs().max()) for k in params)
    for k, p in module.named_parameters():
        assert p.grad is not None, "missing grad for " + k
        assert_close(p.grad.cpu().numpy(), sd[k].grad.numpy(), TOL, "grad " + k, floor=grad_floor * gmax)
    for k, v in module.state_dict().items():           # running statistics updated by a training-mode forward
        if k not in params:
            np.testing.assert_allclose(v.cpu().numpy(), sd[k].numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


SHAPES = [(2, 16, 32), (1, 8, 16), (3, 9, 37), (2, 24, 20)]


@pytest.fixture(params=["s2d", "s2d_materialised", "direct"])
def stride2_algo(request):
    """5x5 stride-2 layers: 3x3 Winograd over the space-to-depth view read in place (default; channel counts that are not a
    power of two >= 32 materialise it), over the materialised view, and the direct stride-2 kernels."""
    from rpg_ramnet_amd import ops
    old, oldf = ops.get_space_to_depth(), ops._S2D_FUSED
    ops.set_space_to_depth(request.param != "direct")
    ops.set_space_to_depth_fused(request.param == "s2d")
    yield request.param
    ops.set_space_to_depth(old)
    ops.set_space_to_depth_fused(oldf)


@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("cin,cout,stride", [(32, 64, 1), (32, 64, 2), (64, 32, 1), (8, 32, 1), (128, 128, 2), (4, 32, 1), (8, 16, 2), (64, 96, 2)])
def test_conv_layer(B, H, W, cin, cout, stride, stride2_algo):
    from rpg_ramnet_amd.model.submodules import ConvLayer
    torch.manual_seed(1)
    m = ConvLayer(cin, cout, 5, stride, 2)
    x = torch.randn(B, cin, H, W)
    run_pair(m, lambda sd, a: torch.relu(torch.nn.functional.conv2d(a, sd["conv2d.weight"], sd["conv2d.bias"], stride, 2)), [x])


@pytest.mark.parametrize("cin", [1, 5, 6])
def test_head_conv_padded_input(cin):
    """Model inputs: NCHW with 1/5/6 channels -> NHWC zero-padded to 4/8 -> 5x5 conv (no input gradient)."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvLayer
    torch.manual_seed(2)
    m = ConvLayer(cin, 32, 5, 1, 2).to(dev())
    x = torch.randn(2, cin, 20, 28)
    y = m(ops.pack_input(x, dev()))
    w, b = m.conv2d.weight.detach().cpu().requires_grad_(True), m.conv2d.bias.detach().cpu().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.conv2d(x, w, b, 1, 2))
    assert_close(nchw(y).detach().cpu().numpy(), ref.detach().numpy(), TOL, "head forward")
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    (nchw(y) * wgt.to(dev())).sum().backward()
    assert_close(m.conv2d.weight.grad.cpu().numpy(), w.grad.numpy(), TOL, "head dW")
    assert_close(m.conv2d.bias.grad.cpu().numpy(), b.grad.numpy(), TOL, "head db")


@pytest.fixture(params=["folded", "folded_dgrad", "folded_direct", "direct"])
def upsample_algo(request):
    """UpsampleConvLayer forward: four 4x4 parity convolutions of the low-res input — on the Winograd F(2x2,4x4) kernel where
    the channel counts allow (default) or on the direct kernel — and the direct 5x5 kernel with the bilinear loader, all against
    the oracle."""
    from rpg_ramnet_amd import ops
    old, oldw, oldg, oldd = ops.get_fold_upsample(), ops._FOLD_WINO, ops._FOLD_WINO_WGRAD, ops._FOLD_DGRAD
    ops.set_fold_upsample(request.param != "direct")
    ops.set_fold_winograd(request.param in ("folded", "folded_dgrad"))
    ops.set_fold_winograd_wgrad(request.param in ("folded", "folded_dgrad"))
    ops.set_fold_dgrad(request.param == "folded_dgrad")           # backward-data through the folded operator's adjoint
    yield request.param
    ops.set_fold_dgrad(oldd)
    ops.set_fold_upsample(old)
    ops.set_fold_winograd(oldw)
    ops.set_fold_winograd_wgrad(oldg)


@pytest.mark.parametrize("B,H,W", [(2, 8, 16), (1, 5, 11), (2, 16, 24), (1, 4, 4), (1, 32, 43)])
@pytest.mark.parametrize("cin,cout,skip", [(64, 32, True), (64, 32, False), (256, 128, True), (128, 64, True), (48, 64, False)])
def test_upsample_conv(B, H, W, cin, cout, skip, upsample_algo):
    from rpg_ramnet_amd.model.submodules import UpsampleConvLayer
    torch.manual_seed(3)
    m = UpsampleConvLayer(cin, cout, 5, padding=2)
    x, s = torch.randn(B, cin, H, W), (torch.randn(B, cin, H, W) if skip else None)

    def oracle(sd, a, sk=None):
        return ramnet_ref.upsample_conv_layer({"L." + k: v for k, v in sd.items()}, "L", a if sk is None else a + sk)

    run_pair(m, oracle, [x, s] if skip else [x])


@pytest.mark.parametrize("cin,cout", [(64, 32), (128, 64), (256, 128)])
def test_fold_weight_algebra_kernels_equal_their_torch_statements(cin, cout):
    """The folded decoder's once-per-step weight algebra runs as library kernels since round 4 (torch.einsum chains
    before): backward-data Winograd pack, border matrices (+ transposes), and the fold of the pass's gradient workspaces into the 5x5
    `.grad` — each against the plain-torch statement it replaced (tr.pack_fold_wino_dgrad / border_matrices / fold_unpack_torch)."""
    from rpg_ramnet_amd import ops
    torch.manual_seed(3)
    w = (torch.randn(cout, cin, 5, 5) * 0.1).to(dev())
    L = Hh.lib()
    out = torch.empty(100 * cout * cin, device=dev())
    Hh.check(L.ramnet_pack_weight_fold_wino_dgrad(ops._p(w), ops._p(out), cout, cin, ops._st()), "pack dgrad")
    assert_close(out.cpu().numpy(), tr.pack_fold_wino_dgrad(w).cpu().numpy(), 1e-6, "fold dgrad pack")
    rows, cols = (torch.empty(2, 5 * cin, 2 * cout, device=dev()) for _ in range(2))
    rows_t, cols_t = (torch.empty(2, 2 * cout, 5 * cin, device=dev()) for _ in range(2))
    Hh.check(L.ramnet_pack_border_weights(ops._p(w), ops._p(rows), ops._p(cols), ops._p(rows_t), ops._p(cols_t), cout, cin, ops._st()), "border")
    r_ref, c_ref = tr.border_matrices(w)
    assert_close(rows.cpu().numpy(), r_ref.cpu().numpy(), 1e-6, "border rows")
    assert_close(cols.cpu().numpy(), c_ref.cpu().numpy(), 1e-6, "border cols")
    assert torch.equal(rows_t, rows.transpose(1, 2)) and torch.equal(cols_t, cols.transpose(1, 2))
    w4 = torch.randn(64 * cin * cout, device=dev())
    dU = torch.randn(100 * cin * cout, device=dev())
    wr, wc = torch.randn(2, 5 * cin, 2 * cout, device=dev()), torch.randn(2, 5 * cin, 2 * cout, device=dev())
    for use_du in (True, False):
        ref = tr.fold_unpack_torch(w4, dU if use_du else None, wr, wc, cout, cin, cin)
        g = torch.full((cout, cin, 5, 5), 0.5, device=dev())
        a4, aU, ar, ac = w4.clone(), dU.clone(), wr.clone(), wc.clone()
        Hh.check(L.ramnet_fold_unpack_wgrad(ops._p(a4), ops._p(aU) if use_du else None, ops._p(ar), ops._p(ac), ops._p(g), cout, cin, cin, ops._st()), "unfold")
        assert_close((g - 0.5).cpu().numpy(), ref.cpu().numpy(), 2e-5, "fold unpack (dU=%s)" % use_du)
        assert float(a4.abs().max()) == 0.0 and float(ar.abs().max()) == 0.0 and float(ac.abs().max()) == 0.0     # zeroed for the next pass
        assert (float(aU.abs().max()) == 0.0) == use_du


def test_upsample_conv_without_activation(upsample_algo):
    """The op also serves a linear upsample-conv (no ReLU): folded frame corrections and gradients without a mask."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(4)
    B, H, W, cin, cout = 2, 6, 9, 32, 48
    w = torch.nn.Parameter((torch.randn(cout, cin, 5, 5) * 0.05).to(dev()))
    b = torch.nn.Parameter((torch.randn(cout) * 0.1).to(dev()))
    cp = ops.ConvParam([w], [b])
    x, s = torch.randn(B, cin, H, W), torch.randn(B, cin, H, W)
    xg, sg = nhwc(x).to(dev()).requires_grad_(True), nhwc(s).to(dev()).requires_grad_(True)
    y = ops.ConvAct.apply(xg, sg, w, b, cp, 1, False, True)
    xr, sr = x.double().requires_grad_(True), s.double().requires_grad_(True)
    wr, br = w.detach().cpu().double().requires_grad_(True), b.detach().cpu().double().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr + sr, scale_factor=2, mode="bilinear", align_corners=False), wr, br, 1, 2)
    assert_close(nchw(y).detach().cpu().numpy(), ref.detach().numpy(), TOL, "linear upsample-conv forward")
    g = torch.randn(ref.shape)
    (ref * g.double()).sum().backward()
    (nchw(y) * g.to(dev())).sum().backward()
    assert_close(w.grad.cpu().numpy(), wr.grad.numpy(), TOL, "dW")
    assert_close(b.grad.cpu().numpy(), br.grad.numpy(), TOL, "db")
    assert_close(nchw(xg.grad).cpu().numpy(), xr.grad.numpy(), TOL, "dx")
    assert_close(nchw(sg.grad).cpu().numpy(), sr.grad.numpy(), TOL, "dskip")


@pytest.mark.parametrize("B,H,W", [(2, 8, 16), (1, 5, 11)])
@pytest.mark.parametrize("cin,cout,skip", [(64, 32, True), (32, 16, False)])
def test_transposed_conv(B, H, W, cin, cout, skip):
    from rpg_ramnet_amd.model.submodules import TransposedConvLayer
    torch.manual_seed(8)
    m = TransposedConvLayer(cin, cout, 5, padding=2)
    x, s = torch.randn(B, cin, H, W), (torch.randn(B, cin, H, W) if skip else None)

    def oracle(sd, a, sk=None):
        return ramnet_ref.transposed_conv_layer({"L." + k: v for k, v in sd.items()}, "L", a if sk is None else a + sk)

    run_pair(m, oracle, [x, s] if skip else [x])


@pytest.fixture(params=["winograd", "direct", "winograd2x4", "winograd2x4_split"])
def algo3x3(request):
    """3x3 stride-1 layers: Winograd F(2x2,3x3) (default at these sizes), the direct implicit GEMM, F(2x4,3x3) (the fine-scale
    kernel of the training batch, forced here for every eligible launch) and F(2x4,3x3) with split bf16 operands (csrc/conv_wino6s.hip),
    all against the oracle at the same tolerances."""
    from rpg_ramnet_amd import ops
    old = ops.get_winograd()
    ops.set_winograd(request.param != "direct")
    if request.param.startswith("winograd2x4"):
        ops.set_winograd_2x4("force")
        ops.set_wgrad_winograd_2x4("force")         # ... and the F(2x4,3x3) backward-weights kernel (csrc/conv_wgrad_wino6.hip)
        ops.set_split_operands(request.param.endswith("_split"))
    yield request.param
    ops.set_winograd(old)
    ops.set_winograd_2x4("auto")
    ops.set_wgrad_winograd_2x4("auto")
    ops.set_split_operands(False)


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 7, 13), (2, 9, 43), (1, 2, 2), (1, 32, 8), (2, 8, 32), (1, 64, 86)])
@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 128), (40, 64), (128, 64), (36, 64)])
def test_winograd2x4_conv3x3_raw(B, H, W, cin, cout):
    """F(2x4,3x3) forward and backward-data launches (csrc/conv_wino6.hip; loaders PLAIN / RELUMASK, epilogues RES_RELU / LINEAR /
    beta accumulation; the three workgroup tile shapes 16x16, 32x8, 8x32; ragged maps and reduction depths that are not multiples
    of 8) against float64 F.conv2d and against F(2x2,3x3); the library must report the r6 kernel for every launch.  "split": the same
    launches with three-term bf16 splits of both operands on the bf16 matrix pipe (csrc/conv_wino6s.hip, ops.set_split_operands) at the
    SAME tolerance; its error against float64 is printed beside the exact-fp32 kernel's."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops, _hip as Hh
    torch.manual_seed(11)
    w = torch.randn(cout, cin, 3, 3) * 0.1
    b = torch.randn(cout) * 0.1
    cp = ops.ConvParam([torch.nn.Parameter(w.to(dev()))], [torch.nn.Parameter(b.to(dev()))])
    x = torch.randn(B, cin, H, W)
    xg = nhwc(x).to(dev()).contiguous()
    res = torch.randn(B, cout, H, W)
    resg = nhwc(res).to(dev()).contiguous()
    taps, tapsd = ops.Taps.get("conv", 3, 1), ops.Taps.get("dgrad1", 3, 1)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    outs = {}
    for mode in ("force", "split", "off"):
        ops.set_winograd_2x4("force" if mode == "split" else mode)
        ops.set_split_operands(mode == "split")
        try:
            kern = []
            y = torch.full((B, H, W, cout), float("nan"), device=dev())
            ops.conv_launch(xg, taps, cp.fwd(), y, cout, bias=cp.bias(), epi=Hh.EPI_RES_RELU, e0=resg)
            kern.append(Hh.lib().ramnet_last_kernel().decode())
            dx = torch.full((B, H, W, cin), float("nan"), device=dev())
            if cin % 64 == 0:
                ops.conv_launch(y, tapsd, cp.bwd(), dx, cin, xm=resg, in_mode=Hh.IN_RELUMASK)
                kern.append(Hh.lib().ramnet_last_kernel().decode())
                acc = xg.clone()
                ops.conv_launch(y, tapsd, cp.bwd(), acc, cin, beta=1.0)
                kern.append(Hh.lib().ramnet_last_kernel().decode())
            else:
                acc = None
        finally:
            ops.set_winograd_2x4("auto")
            ops.set_split_operands(False)
        assert all(k.startswith({"force": "conv_wino_r6_kernel<", "split": "conv_wino_r6s_kernel<", "off": "conv_wino_r_kernel"}[mode]) for k in kern), kern
        outs[mode] = (y, dx, acc)
    yref = torch.relu(ref + res.double())
    dy = torch.where(res > 0, yref, torch.zeros_like(yref))                     # the RELUMASK loader of the backward pass
    dxref = F.conv_transpose2d(dy, w.double(), None, 1, 1)
    accref = x.double() + F.conv_transpose2d(yref, w.double(), None, 1, 1)
    err = lambda a, r: float((nchw(a).cpu().double() - r).abs().max() / r.abs().max())
    print("max-norm error vs float64: forward exact-fp32 %.2e, split operands %.2e" % (err(outs["force"][0], yref), err(outs["split"][0], yref)) +
          ("; backward-data %.2e / %.2e" % (err(outs["force"][1], dxref), err(outs["split"][1], dxref)) if outs["force"][2] is not None else ""))
    for mode in ("force", "split", "off"):
        y, dx, acc = outs[mode]
        assert_close(nchw(y).cpu().numpy(), yref.numpy(), TOL, "forward 2x4=%s" % mode)
        if acc is not None:
            assert_close(nchw(dx).cpu().numpy(), dxref.numpy(), TOL, "dgrad 2x4=%s" % mode)
            assert_close(nchw(acc).cpu().numpy(), accref.numpy(), TOL, "dgrad beta 2x4=%s" % mode)
    # rounding of F(2x4,3x3) in fp32 stays within a few ulp of F(2x2,3x3)
    assert_close(outs["force"][0].cpu().numpy(), outs["off"][0].cpu().numpy(), 2e-5, "F(2x4) vs F(2x2)")
    assert_close(outs["split"][0].cpu().numpy(), outs["force"][0].cpu().numpy(), 2e-5, "F(2x4) split operands vs exact fp32")


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 7, 13), (2, 9, 43), (1, 2, 2)])
@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 96), (40, 20), (128, 256), (36, 64), (68, 36)])
def test_winograd_conv3x3_raw(B, H, W, cin, cout):
    """Winograd forward and backward-data launches (all loader / epilogue combinations the 3x3 layers use) against
    float64 F.conv2d and against the direct kernel; channel counts that are not multiples of the 8 / 64 blocking."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops, _hip as Hh
    torch.manual_seed(11)
    w = torch.randn(cout, cin, 3, 3) * 0.1
    b = torch.randn(cout) * 0.1
    cp = ops.ConvParam([torch.nn.Parameter(w.to(dev()))], [torch.nn.Parameter(b.to(dev()))])
    x = torch.randn(B, cin, H, W)
    xg = nhwc(x).to(dev()).contiguous()
    res = torch.randn(B, cout, H, W)
    resg = nhwc(res).to(dev()).contiguous()
    taps, tapsd = ops.Taps.get("conv", 3, 1), ops.Taps.get("dgrad1", 3, 1)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    outs = {}
    for on in (True, False):
        ops.set_winograd(on)
        try:
            y = torch.full((B, H, W, cout), float("nan"), device=dev())
            ops.conv_launch(xg, taps, cp.fwd(), y, cout, bias=cp.bias(), epi=Hh.EPI_RES_RELU, e0=resg)
            dx = torch.full((B, H, W, cin), float("nan"), device=dev())
            ops.conv_launch(y, tapsd, cp.bwd(), dx, cin, xm=resg, in_mode=Hh.IN_RELUMASK)
            acc = xg.clone()
            ops.conv_launch(y, tapsd, cp.bwd(), acc, cin, beta=1.0)
        finally:
            ops.set_winograd(True)
        outs[on] = (y, dx, acc)
    yref = torch.relu(ref + res.double())
    dy = torch.where(res > 0, yref, torch.zeros_like(yref))                     # the RELUMASK loader of the backward pass
    dxref = F.conv_transpose2d(dy, w.double(), None, 1, 1)
    accref = x.double() + F.conv_transpose2d(yref, w.double(), None, 1, 1)
    for on in (True, False):
        y, dx, acc = outs[on]
        assert_close(nchw(y).cpu().numpy(), yref.numpy(), TOL, "forward winograd=%s" % on)
        assert_close(nchw(dx).cpu().numpy(), dxref.numpy(), TOL, "dgrad winograd=%s" % on)
        assert_close(nchw(acc).cpu().numpy(), accref.numpy(), TOL, "dgrad beta winograd=%s" % on)
    # rounding of F(2x2,3x3) in fp32 stays within a few ulp of the direct kernel
    assert_close(outs[True][0].cpu().numpy(), outs[False][0].cpu().numpy(), 2e-5, "winograd vs direct")


@pytest.mark.parametrize("B,H,W", [(1, 32, 43), (1, 16, 22), (2, 9, 20), (1, 8, 43)])
@pytest.mark.parametrize("cin,cout", [(256, 256), (128, 64), (512, 128), (200, 64)])
def test_winograd_split_reduction(B, H, W, cin, cout):
    """Latency-bound F(2x2,3x3) launches (batch-1 streaming on the coarse scales) split their channel reduction over 2-4 workgroups per
    output tile (ramnet_conv_desc.splitk_ws): forward RES_RELU, backward-data RELUMASK and LINEAR + beta against float64 and against the
    unsplit launch; the partials are joined in split order — bit-reproducible —, the arrival counters return to zero (the workspace is
    reused by every repetition), and the library reports a workspace for at least one launch of every case."""
    import ctypes as C
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops, _hip as Hh
    torch.manual_seed(12)
    w = torch.randn(cout, cin, 3, 3) * 0.05
    b = torch.randn(cout) * 0.1
    cp = ops.ConvParam([torch.nn.Parameter(w.to(dev()))], [torch.nn.Parameter(b.to(dev()))])
    x = torch.randn(B, cin, H, W)
    xg = nhwc(x).to(dev()).contiguous()
    res = torch.randn(B, cout, H, W)
    resg = nhwc(res).to(dev()).contiguous()
    taps, tapsd = ops.Taps.get("conv", 3, 1), ops.Taps.get("dgrad1", 3, 1)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    outs, split = {}, []
    for on in (True, True, False):
        ops.set_winograd_split(on)
        try:
            y = torch.full((B, H, W, cout), float("nan"), device=dev())
            d = ops._conv_desc(xg, taps, cp.fwd(), y, cout, bias=cp.bias(), epi=Hh.EPI_RES_RELU, e0=resg)
            split.append(int(d.splitk_floats))
            Hh.check(Hh.lib().ramnet_conv_launch(C.byref(d), ops._st()), "ramnet_conv_launch")
            dx = torch.full((B, H, W, cin), float("nan"), device=dev())
            d = ops._conv_desc(y, tapsd, cp.bwd(), dx, cin, xm=resg, in_mode=Hh.IN_RELUMASK)
            split.append(int(d.splitk_floats))
            Hh.check(Hh.lib().ramnet_conv_launch(C.byref(d), ops._st()), "ramnet_conv_launch")
            acc = xg.clone()
            ops.conv_launch(y, tapsd, cp.bwd(), acc, cin, beta=1.0)
        finally:
            ops.set_winograd_split(True)
        outs.setdefault(on, []).append((y, dx, acc))
    assert (max(split[:4]) > 0 or cin < 256) and split[4:] == [0, 0], split          # (reductions of 32 chunks and more split)
    yref = torch.relu(ref + res.double())
    dy = torch.where(res > 0, yref, torch.zeros_like(yref))
    dxref = F.conv_transpose2d(dy, w.double(), None, 1, 1)
    accref = x.double() + F.conv_transpose2d(yref, w.double(), None, 1, 1)
    for on in (True, False):
        y, dx, acc = outs[on][0]
        assert_close(nchw(y).cpu().numpy(), yref.numpy(), TOL, "forward split=%s" % on)
        assert_close(nchw(dx).cpu().numpy(), dxref.numpy(), TOL, "dgrad split=%s" % on)
        assert_close(nchw(acc).cpu().numpy(), accref.numpy(), TOL, "dgrad beta split=%s" % on)
    for a, c in zip(outs[True][0], outs[True][1]):
        assert torch.equal(a, c), "split launches are bit-reproducible"
    assert_close(outs[True][0][0].cpu().numpy(), outs[False][0][0].cpu().numpy(), 2e-5, "split vs unsplit")
    for ws in cp.__dict__.get("_splitk", {}).values():      # counters (first floats of the workspace) back at zero
        assert int(ws[:64].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("B,H,W,C", [(1, 32, 43, 256), (1, 16, 22, 128)])
def test_conv_gru_and_encoder_split_reduction_equals_unsplit(B, H, W, C):
    """The other loaders / epilogues that split at batch 1 — ConvGRU gates (concatenation + sigmoid), candidate (h * r product + GRU
    blend) and a stride-2 5x5 encoder over its space-to-depth view: split on against off (2e-5), and the oracle runs of test_conv_gru /
    test_conv_layer at (2, 4, 43, 256) / (128, 128, 2) execute the split launches against float64."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvGRU, ConvLayer
    torch.manual_seed(6)
    gru, enc = ConvGRU(C, C, 3).to(dev()), ConvLayer(C // 2, C, 5, stride=2, padding=2).to(dev())
    xin = torch.randn(B, 2 * H, 2 * W, C // 2, device=dev())
    h0 = torch.tanh(torch.randn(B, H, W, C, device=dev()))
    outs = {}
    for on in (True, False):
        ops.set_winograd_split(on)
        try:
            with torch.no_grad():
                e = enc(xin)
                outs[on] = (e, gru(e, h0))
        finally:
            ops.set_winograd_split(True)
    used = {k: bool(cp.__dict__.get("_splitk")) for m in (gru, enc) for k, cp in m._cps.items()}
    assert used["ur"] or used["o"], used
    for a, c in zip(outs[True], outs[False]):
        assert_close(a.cpu().numpy(), c.cpu().numpy(), 2e-5, "split vs unsplit")


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 32, 43, 256, 128), (1, 32, 43, 128, 64), (1, 16, 22, 256, 128), (2, 8, 12, 128, 64)])
def test_folded_decoder_split_reduction_equals_unsplit(B, H, W, cin, cout):
    """The folded upsample-conv decoders split their channel reduction at batch 1 as well (csrc/conv_wino24.hip; decoder 0: 96 workgroups
    of 16 chunks -> 192 of 8): split against unsplit (2e-5), repeated launches bit-identical, counters back at zero; test_upsample_conv at
    (1, 32, 43) x (256, 128) runs the split launch against the float64 oracle."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import UpsampleConvLayer
    torch.manual_seed(9)
    m = UpsampleConvLayer(cin, cout, 5, padding=2).to(dev())
    x, sk = torch.randn(B, H, W, cin, device=dev()), torch.randn(B, H, W, cin, device=dev())
    outs = []
    for on in (True, True, False):
        ops.set_winograd_split(on)
        try:
            with torch.no_grad():
                outs.append(m(x, sk))
        finally:
            ops.set_winograd_split(True)
    pools = [cp.__dict__.get("_splitk", {}) for cp in m._cps.values()]
    assert any(pools), "the launch did not split"
    assert torch.equal(outs[0], outs[1])
    assert_close(outs[0].cpu().numpy(), outs[2].cpu().numpy(), 2e-5, "split vs unsplit")
    for pool in pools:
        for ws in pool.values():
            assert int(ws[:64].view(torch.int32).abs().sum()) == 0


@pytest.fixture(params=[1, 2])
def wgrad_nf(request):
    """Both workgroup shapes of the Winograd backward-weights kernel: 32 x 32 channels (three per CU: launches on their own stream order) and
    32 x 64 (two per CU: the co-scheduled training step, ops.set_wgrad_overlap(True))."""
    Hh.check(Hh.lib().ramnet_set_option(b"wgrad_wino_nf", request.param), "set_option")
    yield request.param
    Hh.check(Hh.lib().ramnet_set_option(b"wgrad_wino_nf", 1), "set_option")


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 7, 13), (2, 9, 43), (1, 2, 2), (3, 32, 48), (2, 24, 10), (1, 64, 86)])
@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 96), (40, 20), (128, 256)])
def test_winograd_wgrad_raw(B, H, W, cin, cout, wgrad_nf):
    """Winograd backward-weights (+bias, + ReLU mask on the gradient) against float64 autograd and the direct kernel."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(12)
    x = torch.randn(B, cin, H, W)
    dy = torch.randn(B, cout, H, W)
    y = torch.randn(B, cout, H, W)                     # forward output whose sign is the ReLU mask
    g = torch.where(y > 0, dy, torch.zeros_like(dy)).double()
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, bias, 1, 1) * g).sum().backward()
    xg, dyg, yg = (nhwc(t).to(dev()).contiguous() for t in (x, dy, y))
    taps = ops.Taps.get("conv", 3, 1)
    for wino in (True, False, "2x4"):                   # F(2x2,3x3), direct, F(2x4,3x3) (csrc/conv_wgrad_wino6.hip)
        ws = torch.zeros(max(24 * cin * cout, Hh.lib().ramnet_wgrad_wino2x4_ws_floats(cin, cout)), device=dev())
        ws.wino, ws.wino6 = wino is True, wino == "2x4"
        bws = torch.zeros(cout, device=dev())
        for _ in range(2):                              # accumulates over launches (BPTT time steps)
            ops.wgrad_launch(xg, taps, dyg, ws, cout, gmask=yg, dbias=bws)
        assert Hh.lib().ramnet_last_kernel().decode().startswith(
            {True: "conv_wgrad_wino_r_kernel", False: "conv_wgrad_kernel", "2x4": "conv_wgrad_wino_r6_kernel"}[wino])
        grad = torch.zeros(cout, cin, 3, 3, device=dev())
        L = Hh.lib()
        if wino == "2x4":
            Hh.check(L.ramnet_unpack_wgrad_wino2x4(ops._p(ws), ops._p(grad), cout, cin, cin, cout, 0, ops._st()), "unpack")
        elif wino:
            Hh.check(L.ramnet_unpack_wgrad_wino(ops._p(ws), ops._p(grad), cout, cin, cin, cout, 0, ops._st()), "unpack")
        else:
            Hh.check(L.ramnet_unpack_wgrad(ops._p(ws), ops._p(grad), cout, cin, cin, cout, 0, 3, 3, ops._st()), "unpack")
        assert_close(grad.cpu().numpy(), 2 * w.grad.numpy(), TOL, "dW winograd=%s" % wino)
        assert_close(bws.cpu().numpy(), 2 * bias.grad.numpy(), TOL, "db winograd=%s" % wino)


@pytest.mark.parametrize("B,H,W,cin,cout", [(4, 32, 48, 64, 128), (2, 9, 43, 96, 64)])
def test_winograd_wgrad_slabs_bit_reproducible(B, H, W, cin, cout, wgrad_nf):
    """the tile splits of the Winograd backward-weights launch join per-split slabs by plain read-modify-write
    (ramnet_wgrad_desc.dw_slabs) instead of atomic adds.  Two passes over the same data — three accumulating launches each, as over
    BPTT time steps — give BIT-IDENTICAL weight and bias gradients, equal to float64 autograd; the atomic form (one workspace) still
    works and agrees to rounding."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(13)
    x = torch.randn(B, cin, H, W)
    dy = torch.randn(B, cout, H, W)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, bias, 1, 1) * dy.double()).sum().backward()
    xg, dyg = (nhwc(t).to(dev()).contiguous() for t in (x, dy))
    taps = ops.Taps.get("conv", 3, 1)
    L = Hh.lib()
    slabs = L.ramnet_wgrad_wino_slabs(cin, cout)
    assert slabs > 1
    n = 16 * cin * cout
    res = []
    for _ in range(2):
        ws = torch.zeros(slabs * n, device=dev())
        ws.wino, ws.slabs = True, slabs
        bws = torch.zeros(slabs * cout, device=dev())
        for _k in range(3):
            ops.wgrad_launch(xg, taps, dyg, ws, cout, dbias=bws)
        assert float(ws[n:].abs().max()) > 0                        # more than one slab really is in use
        Hh.check(L.ramnet_reduce_slabs(ops._p(ws), slabs, n, ops._st()), "reduce")
        Hh.check(L.ramnet_reduce_slabs(ops._p(bws), slabs, cout, ops._st()), "reduce")
        assert float(ws[n:].abs().max()) == 0.0                    # the fold leaves the other slabs zeroed for the next pass
        grad = torch.zeros(cout, cin, 3, 3, device=dev())
        Hh.check(L.ramnet_unpack_wgrad_wino(ops._p(ws), ops._p(grad), cout, cin, cin, cout, 0, ops._st()), "unpack")
        res.append((grad.cpu().numpy(), bws[:cout].cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert_close(res[0][0], 3 * w.grad.numpy(), TOL, "dW slabs")
    assert_close(res[0][1], 3 * bias.grad.numpy(), TOL, "db slabs")
    ws1 = torch.zeros(n, device=dev())
    ws1.wino = True
    b1 = torch.zeros(cout, device=dev())
    for _k in range(3):
        ops.wgrad_launch(xg, taps, dyg, ws1, cout, dbias=b1)
    g1 = torch.zeros(cout, cin, 3, 3, device=dev())
    Hh.check(L.ramnet_unpack_wgrad_wino(ops._p(ws1), ops._p(g1), cout, cin, cin, cout, 0, ops._st()), "unpack")
    assert_close(g1.cpu().numpy(), res[0][0], 1e-5, "atomic form vs slabs")


@pytest.mark.parametrize("B,H,W,cin,cout", [(4, 32, 48, 64, 128), (2, 9, 43, 96, 64), (8, 16, 22, 128, 128)])
def test_winograd2x4_wgrad_slabs_bit_reproducible(B, H, W, cin, cout):
    """The F(2x4,3x3) backward-weights kernel (csrc/conv_wgrad_wino6.hip) with per-split slabs in the BLOCKED layout of round 5
    (ramnet_wgrad_wino2x4_ws_floats: one MFMA lane's 16 accumulators = 64 contiguous bytes, 16-byte read-modify-write joins, tile splits dealt
    to the XCDs from 8 splits up): two passes of three accumulating launches give BIT-IDENTICAL weight and bias gradients, equal to float64
    autograd; the atomic form (one slab) agrees to rounding; a 2-segment launch (ramnet_wgrad_desc.segs) equals two launches to rounding."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(14)
    x = torch.randn(B, cin, H, W)
    dy = torch.randn(B, cout, H, W)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, bias, 1, 1) * dy.double()).sum().backward()
    xg, dyg = (nhwc(t).to(dev()).contiguous() for t in (x, dy))
    taps = ops.Taps.get("conv", 3, 1)
    L = Hh.lib()
    slabs, n = L.ramnet_wgrad_wino2x4_slabs(cin, cout), L.ramnet_wgrad_wino2x4_ws_floats(cin, cout)
    assert slabs > 1

    def unpack(ws):
        grad = torch.zeros(cout, cin, 3, 3, device=dev())
        Hh.check(L.ramnet_unpack_wgrad_wino2x4(ops._p(ws), ops._p(grad), cout, cin, cin, cout, 0, ops._st()), "unpack")
        return grad.cpu().numpy()
    res = []
    for _ in range(2):
        ws = torch.zeros(slabs * n, device=dev())
        ws.wino, ws.wino6, ws.slabs = False, True, slabs
        bws = torch.zeros(slabs * cout, device=dev())
        for _k in range(3):
            ops.wgrad_launch(xg, taps, dyg, ws, cout, dbias=bws)
        assert Hh.lib().ramnet_last_kernel().decode().startswith("conv_wgrad_wino_r6_kernel")
        assert float(ws[n:].abs().max()) > 0                        # more than one slab really is in use
        Hh.check(L.ramnet_reduce_slabs(ops._p(ws), slabs, n, ops._st()), "reduce")
        Hh.check(L.ramnet_reduce_slabs(ops._p(bws), slabs, cout, ops._st()), "reduce")
        res.append((unpack(ws), bws[:cout].cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert_close(res[0][0], 3 * w.grad.numpy(), TOL, "dW slabs")
    assert_close(res[0][1], 3 * bias.grad.numpy(), TOL, "db slabs")
    ws1 = torch.zeros(n, device=dev())
    ws1.wino, ws1.wino6 = False, True
    b1 = torch.zeros(cout, device=dev())
    for _k in range(3):
        ops.wgrad_launch(xg, taps, dyg, ws1, cout, dbias=b1)
    assert_close(unpack(ws1), res[0][0], 1e-5, "atomic form vs slabs")
    # two segments in one launch (the same tensors twice) + one plain launch == three launches
    ws2 = torch.zeros(slabs * n, device=dev())
    ws2.wino, ws2.wino6, ws2.slabs = False, True, slabs
    b2 = torch.zeros(slabs * cout, device=dev())
    segs = (Hh.WgradSeg * 2)()
    for sg in segs:
        sg.x0, sg.dout = ops._p(xg), ops._p(dyg)
    ops.wgrad_launch(xg, taps, dyg, ws2, cout, dbias=b2, segs=segs)
    ops.wgrad_launch(xg, taps, dyg, ws2, cout, dbias=b2)
    Hh.check(L.ramnet_reduce_slabs(ops._p(ws2), slabs, n, ops._st()), "reduce")
    Hh.check(L.ramnet_reduce_slabs(ops._p(b2), slabs, cout, ops._st()), "reduce")
    assert_close(unpack(ws2), res[0][0], 1e-5, "two segments + one launch vs three launches")
    assert_close(b2[:cout].cpu().numpy(), res[0][1], 1e-5, "bias: two segments + one launch vs three launches")


@pytest.mark.parametrize("B,H,W,cin,cout", [(4, 32, 48, 64, 128), (2, 9, 43, 96, 64), (8, 16, 22, 128, 128), (1, 5, 7, 64, 32), (3, 13, 20, 160, 192)])
def test_direct_split_wgrad_raw(B, H, W, cin, cout):
    """RAMNET_ALGO_DIRECT_SPLIT (csrc/conv_wgrad_dsplit.hip): the direct 3x3 backward-weights kernel on the bf16 matrix pipe — both operands
    split into three bf16 terms when a strip is staged, six of the nine partial products — against float64 autograd at the tolerance of the
    exact-fp32 kernels; per-split slabs in the blocked layout give BIT-IDENTICAL gradients over two passes of three accumulating launches, the
    atomic form (one slab) agrees to rounding, and the error against float64 is printed beside the exact F(2x4,3x3) kernel's."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(15)
    x = torch.randn(B, cin, H, W)
    dy = torch.randn(B, cout, H, W)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, bias, 1, 1) * dy.double()).sum().backward()
    xg, dyg = (nhwc(t).to(dev()).contiguous() for t in (x, dy))
    taps = ops.Taps.get("conv", 3, 1)
    L = Hh.lib()
    slabs, n = L.ramnet_wgrad_dsplit_slabs(cin, cout), L.ramnet_wgrad_dsplit_ws_floats(cin, cout)
    assert slabs > 1

    def unpack(ws):
        grad = torch.zeros(cout, cin, 3, 3, device=dev())
        Hh.check(L.ramnet_unpack_wgrad_dsplit(ops._p(ws), ops._p(grad), cout, cin, cin, cout, 0, ops._st()), "unpack")
        return grad.cpu().numpy()
    res = []
    for _ in range(2):
        ws = torch.zeros(slabs * n, device=dev())
        ws.wino, ws.wino6, ws.wg_dsplit, ws.slabs = False, False, True, slabs
        bws = torch.zeros(slabs * cout, device=dev())
        for _k in range(3):
            ops.wgrad_launch(xg, taps, dyg, ws, cout, dbias=bws)
        assert Hh.lib().ramnet_last_kernel().decode().startswith("conv_wgrad_dsplit_kernel")
        Hh.check(L.ramnet_reduce_slabs(ops._p(ws), slabs, n, ops._st()), "reduce")
        Hh.check(L.ramnet_reduce_slabs(ops._p(bws), slabs, cout, ops._st()), "reduce")
        assert float(ws[n:].abs().max()) == 0.0
        res.append((unpack(ws), bws[:cout].cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert_close(res[0][0], 3 * w.grad.numpy(), TOL, "dW slabs")
    assert_close(res[0][1], 3 * bias.grad.numpy(), TOL, "db slabs")
    ws1 = torch.zeros(n, device=dev())
    ws1.wino, ws1.wino6, ws1.wg_dsplit = False, False, True
    b1 = torch.zeros(cout, device=dev())
    for _k in range(3):
        ops.wgrad_launch(xg, taps, dyg, ws1, cout, dbias=b1)
    assert_close(unpack(ws1), res[0][0], 1e-5, "atomic form vs slabs")
    # the exact-fp32 F(2x4,3x3) kernel on the same tensors: both errors against float64 side by side
    n6 = L.ramnet_wgrad_wino2x4_ws_floats(cin, cout)
    ws6 = torch.zeros(n6, device=dev())
    ws6.wino, ws6.wino6, ws6.wg_dsplit = False, True, False
    b6 = torch.zeros(cout, device=dev())
    ops.wgrad_launch(xg, taps, dyg, ws6, cout, dbias=b6)
    assert Hh.lib().ramnet_last_kernel().decode().startswith("conv_wgrad_wino_r6_kernel")
    g6 = torch.zeros(cout, cin, 3, 3, device=dev())
    Hh.check(L.ramnet_unpack_wgrad_wino2x4(ops._p(ws6), ops._p(g6), cout, cin, cin, cout, 0, ops._st()), "unpack")
    ref = w.grad.numpy()
    e_split = np.abs(res[0][0] / 3 - ref).max() / np.abs(ref).max()
    e_exact = np.abs(g6.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("dW max-norm error vs float64: direct split %.3e, exact F(2x4,3x3) %.3e" % (e_split, e_exact))
    assert e_split < 4 * max(e_exact, 1e-7)


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 16, 64), (1, 7, 13, 32), (2, 4, 43, 256)])
def test_residual_block(B, H, W, C, algo3x3):
    from rpg_ramnet_amd.model.submodules import ResidualBlock
    torch.manual_seed(4)
    m = ResidualBlock(C, C)
    run_pair(m, lambda sd, a: ramnet_ref.residual_block({"L." + k: v for k, v in sd.items()}, "L", a),
             [torch.randn(B, C, H, W)])


def _randomise_norm(m, seed):
    """Non-trivial affine parameters and running statistics for every norm member of m."""
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.InstanceNorm2d)):
            with torch.no_grad():
                if mod.weight is not None:
                    mod.weight.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
                    mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
                if mod.running_mean is not None:
                    mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                    mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("norm", ["BN", "IN"])
@pytest.mark.parametrize("layer", ["conv_s2", "conv_s1_c12", "pred", "upconv", "upconv_skip", "tconv", "resblock"])
def test_norm_layers(layer, norm, training):
    """ConvLayer / UpsampleConvLayer / TransposedConvLayer / ResidualBlock with `norm` 'BN' | 'IN' (submodules.py:13-24, 29-30, 52-62,
    82-94, 188-193, 203-210): forward, input and parameter gradients (incl. the BatchNorm affine pair) and the running-buffer update
    against the oracle, in training mode (statistics of the input) and in eval mode (running statistics)."""
    from rpg_ramnet_amd.model import submodules as S
    torch.manual_seed(11)
    B, Hh, W = 2, 12, 20
    nk = dict(norm=norm, training=training)
    if layer == "conv_s2":
        m, ins = S.ConvLayer(32, 64, 5, 2, 2, norm=norm), [torch.randn(B, 32, Hh, W)]
        fn = lambda sd, a: ramnet_ref.conv_layer(sd, "L", a, 2, 2, **nk)  # noqa: E731
    elif layer == "conv_s1_c12":       # a channel count that is not a multiple of 16 (scalar path of the kernels: 12 = 3 quads)
        m, ins = S.ConvLayer(8, 12, 3, 1, 1, norm=norm), [torch.randn(B, 8, Hh, W)]
        fn = lambda sd, a: ramnet_ref.conv_layer(sd, "L", a, 1, 1, **nk)  # noqa: E731
    elif layer == "pred":              # conv1x1 -> norm -> (no activation): ONE output channel
        m, ins = S.ConvLayer(32, 1, 1, activation=None, norm=norm), [torch.randn(B, 32, Hh, W)]
        fn = lambda sd, a: ramnet_ref.conv_layer(sd, "L", a, 1, 0, relu=False, **nk)  # noqa: E731
    elif layer in ("upconv", "upconv_skip"):
        m = S.UpsampleConvLayer(64, 32, 5, padding=2, norm=norm)
        ins = [torch.randn(B, 64, Hh, W)] + ([torch.randn(B, 64, Hh, W)] if layer == "upconv_skip" else [])
        fn = lambda sd, a, b=None: ramnet_ref.upsample_conv_layer(sd, "L", a if b is None else a + b, **nk)  # noqa: E731
    elif layer == "tconv":
        m, ins = S.TransposedConvLayer(64, 32, 5, padding=2, norm=norm), [torch.randn(B, 64, Hh, W)]
        fn = lambda sd, a: ramnet_ref.transposed_conv_layer(sd, "L", a, **nk)  # noqa: E731
    else:
        m, ins = S.ResidualBlock(64, 64, norm=norm), [torch.randn(B, 64, Hh, W)]
        fn = lambda sd, a: ramnet_ref.residual_block(sd, "L", a, **nk)  # noqa: E731
    _randomise_norm(m, 5)
    m.train(training)
    run_pair(m, lambda sd, *a: fn({"L." + k: v for k, v in sd.items()}, *a), ins, grad_floor=1e-2)


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 16, 64), (1, 7, 13, 32), (2, 4, 43, 256), (1, 16, 32, 128)])
def test_conv_gru(B, H, W, C, algo3x3):
    from rpg_ramnet_amd.model.submodules import ConvGRU
    torch.manual_seed(5)
    m = ConvGRU(C, C, 3)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)     # non-zero biases (the reference initialises them to 0)
    run_pair(m, lambda sd, a, h: ramnet_ref.conv_gru({"L." + k: v for k, v in sd.items()}, "L", a, h),
             [torch.randn(B, C, H, W), torch.tanh(torch.randn(B, C, H, W))])


@pytest.fixture
def direct_split_wgrad():
    """Split operands everywhere they exist: F(2x4,3x3) forward / backward-data with split operands (csrc/conv_wino6s.hip) AND the direct
    split backward-weights kernel (csrc/conv_wgrad_dsplit.hip, ops.set_split_wgrad — off by default: profiles/r06_dsplit_notes.md)."""
    from rpg_ramnet_amd import ops
    ops.set_winograd_2x4("force")
    ops.set_wgrad_winograd_2x4("force")
    ops.set_split_operands(True)
    ops.set_split_wgrad(True)
    yield
    ops.set_split_wgrad(False)
    ops.set_split_operands(False)
    ops.set_winograd_2x4("auto")
    ops.set_wgrad_winograd_2x4("auto")


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 16, 64), (1, 7, 13, 32), (2, 4, 43, 256), (1, 16, 32, 128)])
def test_conv_gru_direct_split_backward_weights(B, H, W, C, direct_split_wgrad):
    """ConvGRU (concatenated inputs, both gate convolutions and the candidate) with the direct split backward-weights kernel: state, input
    and parameter gradients against the float64 oracle at the tolerance of the exact kernels (test_direct_split_wgrad_raw pins the kernel symbol)."""
    from rpg_ramnet_amd.model.submodules import ConvGRU
    torch.manual_seed(5)
    m = ConvGRU(C, C, 3)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    run_pair(m, lambda sd, a, h: ramnet_ref.conv_gru({"L." + k: v for k, v in sd.items()}, "L", a, h),
             [torch.randn(B, C, H, W), torch.tanh(torch.randn(B, C, H, W))])


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 16, 64), (2, 4, 43, 256)])
def test_residual_block_direct_split_backward_weights(B, H, W, C, direct_split_wgrad):
    """Residual block (ReLU masks on the input of the second layer's backward-weights launch and on both gradients: the masked
    instantiations of the kernel) against the float64 oracle."""
    from rpg_ramnet_amd.model.submodules import ResidualBlock
    torch.manual_seed(4)
    m = ResidualBlock(C, C)
    run_pair(m, lambda sd, a: ramnet_ref.residual_block({"L." + k: v for k, v in sd.items()}, "L", a), [torch.randn(B, C, H, W)])


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 32, 32, 64), (1, 18, 26, 64, 128)])
def test_stride2_encoder_direct_split_backward_weights(B, H, W, cin, cout, direct_split_wgrad):
    """Stride-2 5x5 encoder over its space-to-depth view (RAMNET_IN_S2D) with the direct split backward-weights kernel."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvLayer
    torch.manual_seed(1)
    m = ConvLayer(cin, cout, 5, 2, 2)
    assert ops.get_space_to_depth()
    run_pair(m, lambda sd, a: torch.relu(torch.nn.functional.conv2d(a, sd["conv2d.weight"], sd["conv2d.bias"], 2, 2)), [torch.randn(B, cin, H, W)])


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 16, 64), (1, 33, 45, 128), (8, 5, 43, 256)])
def test_conv_gru_backward_stage_b_fused_equals_unfused(B, H, W, C, algo3x3):
    """Stage B of the ConvGRU backward (dpr = d(h.r) h r (1-r), dh = dh'(1-u) + d(h.r) r; submodules.py:448-452 differentiated) runs in
    the epilogue of the candidate convolution's backward-data launch (RAMNET_EPI_GRU_BWD) for hidden sizes that are multiples of 64 and
    as its own launch (ramnet_gru_bwd_b) otherwise: both forms against each other on every 3x3 algorithm (the same sums in the same
    order up to the fused multiply-adds of the last step: 1e-5), and test_conv_gru compares each with the float64 oracle."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvGRU
    torch.manual_seed(11)
    m = ConvGRU(C, C, 3).to(dev())
    x0, h0 = torch.randn(B, H, W, C, device=dev()), torch.tanh(torch.randn(B, H, W, C, device=dev()))
    wgt = torch.randn(B, H, W, C, device=dev())
    res = {}
    for on in (True, False):
        ops.set_gru_bwd_fused(on)
        try:
            m.zero_grad()
            x, h = x0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
            (m(x, h) * wgt).sum().backward()
            res[on] = [x.grad.clone(), h.grad.clone()] + [p.grad.clone() for p in m.parameters()]
        finally:
            ops.set_gru_bwd_fused(True)
    for a, c in zip(res[True], res[False]):
        assert_close(a.cpu().numpy(), c.cpu().numpy(), 1e-5, "fused vs unfused stage B")


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 16, 64), (1, 33, 45, 128), (8, 5, 43, 256), (1, 7, 13, 32), (2, 6, 9, 12)])
def test_conv_gru_materialised_h_r_equals_the_loader_product(B, H, W, C, algo3x3):
    """RAMNET_EPI_SIGMOID_HR (ABI 21): the gates launch also writes h.r, the candidate convolution W_o*[x, h.r] (submodules.py:450) and its
    backward-weights launch read the plain concatenation [x | h.r] — against the loader form (RAMNET_IN_CAT_MUL: the product formed while
    the patch is staged) on every 3x3 algorithm: the same single-rounded products enter the same sums, so state, input gradients and (Winograd
    kernels: per-split slabs) weight gradients are BIT-identical; test_conv_gru compares either form with the float64 oracle."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvGRU
    torch.manual_seed(17)
    m = ConvGRU(C, C, 3).to(dev())
    x0, h0 = torch.randn(B, H, W, C, device=dev()), torch.tanh(torch.randn(B, H, W, C, device=dev()))
    wgt = torch.randn(B, H, W, C, device=dev())
    res = {}
    for on in (True, False):
        ops.set_gru_materialize_hr(on)
        try:
            m.zero_grad()
            x, h = x0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
            y = m(x, h)
            (y * wgt).sum().backward()
            res[on] = [y.detach().clone(), x.grad.clone(), h.grad.clone()] + [p.grad.clone() for p in m.parameters()]
        finally:
            ops.set_gru_materialize_hr(True)
    for k, (a, c) in enumerate(zip(res[True], res[False])):
        if k >= 3 and algo3x3 == "direct":      # (the direct backward-weights kernel joins its tile splits with atomic adds: order-dependent sums)
            assert_close(a.cpu().numpy(), c.cpu().numpy(), 1e-5, "materialised h.r vs loader product, weight gradients")
        else:
            assert torch.equal(a, c), "materialised h.r vs loader product (tensor %d)" % k


@pytest.mark.parametrize("cell", ["convgru", "convlstm"])
def test_overwriting_a_state_slot_before_backward_raises(cell):
    """The cells write their new state into a caller's buffer (`out=`: the slots of the time-batched forward, ops.arena_slots) behind
    autograd's back; the write is declared (ctx.mark_dirty), so a later write to a slot whose OLD content a pending backward saved — the next
    cell keeps its input state for its backward (submodules.py:436-454 differentiated) — raises instead of differentiating the overwritten
    state.  Slots are aliases with version counters of their own: writing slot 1 does not invalidate what was saved of slot 0."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvGRU, ConvLSTM
    torch.manual_seed(2)
    B, H, W, C = 1, 8, 12, 32
    m = (ConvGRU if cell == "convgru" else ConvLSTM)(C, C, 3).to(dev())
    pair = cell == "convlstm"

    def slots():
        bufs = [ops.arena_slots(2, (B, H, W, C), dev()) for _ in range(2 if pair else 1)]
        return bufs, (lambda k: tuple(b[1][k] for b in bufs) if pair else bufs[0][1][k])

    def first(st):
        return st[0] if pair else st

    x = [torch.randn(B, H, W, C, device=dev(), requires_grad=True) for _ in range(3)]
    h0 = torch.zeros(B, H, W, C, device=dev())
    h0 = (h0, h0.clone()) if pair else h0
    # (1) the legal pattern: slot 0, then slot 1 from slot 0 — backward runs, and the buffer holds both states consecutively
    bufs, slot = slots()
    s0 = m(x[0], h0, slot(0))
    s1 = m(x[1], s0, slot(1))
    assert first(s0).data_ptr() == bufs[0][0].data_ptr() and first(s1).data_ptr() == bufs[0][0][1].data_ptr()
    first(s1).sum().backward()
    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) for t in x[:2])
    # (2) slot 0 rewritten while the second cell's backward still needs its old content
    bufs, slot = slots()
    s0 = m(x[0], h0, slot(0))
    s1 = m(x[1], s0, slot(1))
    m(x[2], h0, slot(0))
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        first(s1).sum().backward()


@pytest.mark.parametrize("n,defer,B,H,W,C", [(5, 5, 2, 8, 16, 64), (5, 3, 1, 33, 45, 64), (6, 48, 2, 16, 22, 128), (3, 2, 8, 5, 43, 256)])
def test_conv_gru_deferred_backward_weights_equal_per_update_launches(n, defer, B, H, W, C):
    """ops.set_wgrad_defer: the backward-weights launches of n chained updates of ONE ConvGRU cell (submodules.py:436-454) queued and
    reduced by multi-segment launches (ramnet_wgrad_desc.segs; F(2x4,3x3) kernel, per-split slabs) — queues that fill during the pass and a
    remainder flushed when the engine finishes — against one launch per update: the same sums in another order (1e-5 of the tensor's
    maximum), repeated runs bit-identical, input gradients untouched (they do not depend on the weight-gradient launches)."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvGRU
    torch.manual_seed(13)
    m = ConvGRU(C, C, 3).to(dev())
    xs = [torch.randn(B, H, W, C, device=dev()) for _ in range(n)]
    h0 = torch.tanh(torch.randn(B, H, W, C, device=dev()))
    wgt = torch.randn(B, H, W, C, device=dev())
    res = []
    ops.set_wgrad_winograd_2x4("force")
    try:
        for d in (defer, defer, 0):
            ops.set_wgrad_defer(d)
            m.zero_grad()
            h = h0.clone().requires_grad_(True)
            st = h
            for x in xs:
                st = m(x, st)
            (st * wgt).sum().backward()
            res.append([h.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    finally:
        ops.set_wgrad_defer(0)
        ops.set_wgrad_winograd_2x4("auto")
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c), "deferred launches are bit-reproducible"
    assert torch.equal(res[0][0], res[2][0])
    for a, c in zip(res[0][1:], res[2][1:]):
        assert_close(a.cpu().numpy(), c.cpu().numpy(), 1e-5, "deferred vs per-update backward-weights")


@pytest.mark.parametrize("n,B,H,W,C,wide", [(5, 2, 8, 12, 64, True), (3, 1, 7, 13, 32, False), (8, 1, 4, 6, 8, True), (2, 3, 5, 9, 128, True)])
def test_time_fan_gradient_kernel(n, B, H, W, C, wide):
    """ramnet_cat_batch_add behind ops.TimeFan / ops.TimeSplit: the gradient of a time-batched feature = the gradients of its n slices
    (channel slices of wider tensors, as the [dx | dh] of a ConvGRU backward hands them over: ld = 2C) one behind the other along the batch
    axis, plus the gradient of the batched consumer — bit-equal to torch.cat + add (one addition per element)."""
    from rpg_ramnet_amd import ops
    torch.manual_seed(n)
    x = torch.randn(n * B, H, W, C, device=dev(), requires_grad=True)
    wide_g = [torch.randn(B, H, W, 2 * C if wide else C, device=dev()) for _ in range(n)]
    g_all = torch.randn(n * B, H, W, C, device=dev())
    out = ops.TimeFan.apply(x * 1.0, n)
    torch.autograd.backward([out[0]] + list(out[1:]), [g_all] + [g[..., :C] for g in wide_g])
    ref = g_all + torch.cat([g[..., :C] for g in wide_g], 0)
    assert torch.equal(x.grad, ref)
    x.grad = None
    parts = ops.TimeSplit.apply(x * 1.0, n)
    torch.autograd.backward(list(parts), [g[..., :C] for g in wide_g])
    assert torch.equal(x.grad, torch.cat([g[..., :C] for g in wide_g], 0))


@pytest.mark.parametrize("n,B,H,W,C,wide", [(5, 2, 8, 12, 64, True), (3, 1, 7, 13, 32, False), (2, 3, 5, 9, 128, True)])
def test_time_fan_gradient_kernel_with_relu_mask(n, B, H, W, C, wide):
    """ramnet_cat_batch_add_masked (ABI 24): the same sum times (feature > 0) — bit-equal to torch's cat + add + where."""
    from rpg_ramnet_amd import ops
    torch.manual_seed(n)
    x = torch.randn(n * B, H, W, C, device=dev(), requires_grad=True)
    wide_g = [torch.randn(B, H, W, 2 * C if wide else C, device=dev()) for _ in range(n)]
    g_all = torch.randn(n * B, H, W, C, device=dev())
    keep = x.detach() > 0
    zero = torch.zeros((), device=dev())
    out = ops.TimeFan.apply(x * 1.0, n, True)
    torch.autograd.backward([out[0]] + list(out[1:]), [g_all] + [g[..., :C] for g in wide_g])
    assert torch.equal(x.grad, torch.where(keep, g_all + torch.cat([g[..., :C] for g in wide_g], 0), zero))
    x.grad = None
    parts = ops.TimeSplit.apply(x * 1.0, n, True)
    torch.autograd.backward(list(parts), [g[..., :C] for g in wide_g])
    assert torch.equal(x.grad, torch.where(keep, torch.cat([g[..., :C] for g in wide_g], 0), zero))
    x.grad = None
    out = ops.TimeFan.apply(x * 1.0, n, True)
    out[0].backward(g_all)                                                   # the batched consumer alone: one relu_bwd launch
    assert torch.equal(x.grad, torch.where(keep, g_all, zero))


@pytest.mark.parametrize("B,C,H,W", [(3, 5, 16, 24), (2, 5, 7, 9), (2, 1, 16, 24), (2, 10, 8, 12), (1, 8, 4, 4), (40, 5, 64, 88)])
def test_pack_input_is_the_padded_channels_last_copy(B, C, H, W):
    """ops.pack_input (model.py:177,200: the model's NCHW input -> NHWC with channels zero-padded to a multiple of 4): every layout of the
    repack kernel against torch, bit for bit."""
    from rpg_ramnet_amd import ops
    torch.manual_seed(C)
    x = torch.randn(B, C, H, W, device=dev())
    y = ops.pack_input(x, dev())
    cp = (C + 3) // 4 * 4
    ref = torch.zeros(B, H, W, cp, device=dev())
    ref[..., :C] = x.permute(0, 2, 3, 1)
    assert y.shape == ref.shape and torch.equal(y, ref)


@pytest.mark.parametrize("B,H,W,C,n", [(4, 64, 96, 32, 2), (2, 33, 47, 32, 1), (6, 16, 24, 64, 3)])
def test_pred_sigmoid_si_matches_torch_and_is_bit_reproducible(B, H, W, C, n):
    """ops.PredSigmoidSI (prediction layer + scale-invariant loss of its n batch segments in the layer's own launches, statenet.py:313 +
    model/loss.py:6-9) against plain torch in float64 — losses, dx, dw, db — and, ABI 24: the weight / bias gradients are joined in a fixed
    order through the forward's scratch, so two backward passes give the same bits (the bias gradient is a cancelling sum: with fp32 atomics
    its last digits followed the arrival order of the workgroups)."""
    from rpg_ramnet_amd import ops
    torch.manual_seed(B + C)
    x0 = torch.randn(B, H, W, C, device=dev())
    w0 = (torch.randn(1, C, 1, 1, device=dev()) * 0.2)
    b0 = torch.randn(1, device=dev()) * 0.1
    tg = [torch.rand(B // n, 1, H, W, device=dev()) for _ in range(n)]
    tg[0][0, 0, :3, :5] = float("nan")
    dyv = torch.randn(B, 1, H, W, device=dev()) * 1e-3
    coef = [0.7 + 0.3 * i for i in range(n)]

    def run():
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = ops.PredSigmoidSI.apply(x, w, b, 1.0, 0.85, False, *tg)
        total = sum(c * l for c, l in zip(coef, out[1:])) + (out[0] * dyv).sum()
        total.backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in out], x.grad.clone(), w.grad.clone(), b.grad.clone()

    o1, dx1, dw1, db1 = run()
    o2, dx2, dw2, db2 = run()
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2) and torch.equal(dx1, dx2) and all(torch.equal(a, c) for a, c in zip(o1, o2))
    # mask_x: dx leaves with the ReLU mask of x itself applied (x = the last decoder's output on the path); everything else unchanged
    xm, wm, bm = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    outm = ops.PredSigmoidSI.apply(xm, wm, bm, 1.0, 0.85, True, *tg)
    (sum(c * l for c, l in zip(coef, outm[1:])) + (outm[0] * dyv).sum()).backward()
    assert torch.equal(xm.grad, torch.where(x0 > 0, dx1, torch.zeros((), device=dev()))) and torch.equal(wm.grad, dw1) and torch.equal(bm.grad, db1)
    # float64 torch
    x, w, b = x0.double().requires_grad_(True), w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    y = torch.sigmoid((x * w.view(1, 1, 1, C)).sum(-1, keepdim=True) + b).permute(0, 3, 1, 2)
    losses = []
    for i in range(n):
        d = y[i * (B // n):(i + 1) * (B // n)] - tg[i].double()
        d = d[~torch.isnan(d)]
        losses.append((d ** 2).mean() - 0.85 * d.mean() ** 2)
    (sum(c * l for c, l in zip(coef, losses)) + (y * dyv.double()).sum()).backward()
    assert_close(o1[0].cpu().numpy(), y.detach().cpu().numpy(), 1e-5, "prediction")
    for i in range(n):
        assert abs(float(o1[1 + i]) - float(losses[i])) <= 1e-5 * abs(float(losses[i])), "loss %d" % i
    assert_close(dx1.cpu().numpy(), x.grad.cpu().numpy(), 2e-4, "dx")
    assert_close(dw1.cpu().numpy().reshape(-1), w.grad.cpu().numpy().reshape(-1), 2e-4, "dw")
    assert abs(float(db1) - float(b.grad)) <= 2e-4 * float(x.grad.abs().sum() / C), "db (against the size of its terms)"


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 16, 64), (1, 7, 13, 32), (2, 4, 43, 256)])
def test_conv_lstm(B, H, W, C):
    from rpg_ramnet_amd.model.submodules import ConvLSTM

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.L = ConvLSTM(C, C, 3)

        def forward(self, x, h, c):
            return self.L(x, (h, c))

    torch.manual_seed(6)
    m = Wrap()
    run_pair(m, lambda sd, a, h, c: ramnet_ref.conv_lstm(sd, "L", a, (h, c)),
             [torch.randn(B, C, H, W), torch.tanh(torch.randn(B, C, H, W)), torch.randn(B, C, H, W)])


@pytest.mark.parametrize("scale", [30.0, 300.0])
def test_gate_nonlinearities_saturate_cleanly(scale):
    """The epilogues evaluate sigmoid / tanh on v_exp_f32 + v_rcp_f32 (common.hpp).  Pre-activations far outside the usual range
    (|z| up to several hundred: exp overflows to inf on one side, underflows to 0 on the other) must give exactly the saturated
    values the reference's torch.sigmoid / torch.tanh give, never NaN, and stay within 1e-6 absolute everywhere else."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.submodules import ConvGRU, ConvLSTM
    torch.manual_seed(17)
    B, C, H, W = 1, 32, 8, 12
    x, h, c = torch.randn(B, C, H, W), torch.tanh(torch.randn(B, C, H, W)), torch.randn(B, C, H, W)
    gru, lstm = ConvGRU(C, C, 3), ConvLSTM(C, C, 3)
    with torch.no_grad():
        for m in (gru, lstm):
            for p in m.parameters():
                if p.dim() == 1:
                    p.copy_(torch.linspace(-scale, scale, p.numel())[torch.randperm(p.numel())])      # biases spanning +-scale
    with torch.no_grad():
        got = gru.to(dev())(nhwc(x).to(dev()), nhwc(h).to(dev()))
        ref = ramnet_ref.conv_gru({"L." + k: v.cpu().double() for k, v in gru.state_dict().items()}, "L", x.double(), h.double())
        got = got[0] if isinstance(got, tuple) else got
        ref = ref[0] if isinstance(ref, tuple) else ref
        g = nchw(got).cpu().double()
        assert torch.isfinite(g).all()
        assert float((g - ref).abs().max()) < 5e-6
        hl, cl = lstm.to(dev())(nhwc(x).to(dev()), (nhwc(h).to(dev()), nhwc(c).to(dev())))
        rh, rc = ramnet_ref.conv_lstm({"L." + k: v.cpu().double() for k, v in lstm.state_dict().items()}, "L", x.double(), (h.double(), c.double()))
        for a, r in ((hl, rh), (cl, rc)):
            a = nchw(a).cpu().double()
            assert torch.isfinite(a).all()
            assert float((a - r).abs().max()) < 5e-6 * max(1.0, float(r.abs().max()))
        # 1x1 prediction head + sigmoid: exact 0 / 1 where torch saturates
        xp = nhwc(torch.randn(1, 32, 6, 8) * scale).to(dev())
        w, b = torch.randn(1, 32, 1, 1), torch.zeros(1)
        y = ops.PredSigmoid.apply(xp, w.to(dev()), b.to(dev()))
        r = torch.sigmoid(torch.nn.functional.conv2d(nchw(xp).cpu().double(), w.double(), b.double()))
        assert torch.isfinite(y).all()
        assert float(y.min()) >= 0.0 and float(y.max()) <= 1.0
        assert float((y.cpu().double() - r).abs().max()) < 1e-4       # (the fp32 pre-activation of magnitude ~scale * 6 rounds at 1e-5 * scale)


def test_pred_sigmoid():
    from rpg_ramnet_amd import ops
    torch.manual_seed(7)
    x = torch.randn(2, 32, 12, 20)
    w, b = (torch.randn(1, 32, 1, 1) * 0.2), torch.randn(1) * 0.1
    xg = nhwc(x).to(dev()).requires_grad_(True)
    wg, bg = w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    y = ops.PredSigmoid.apply(xg, wg, bg)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.sigmoid(torch.nn.functional.conv2d(xc, wc, bc))
    assert_close(y.detach().cpu().numpy(), ref.detach().numpy(), TOL, "pred")
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    (y * wgt.to(dev())).sum().backward()
    assert_close(nchw(xg.grad).cpu().numpy(), xc.grad.numpy(), TOL, "pred dx")
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), TOL, "pred dw")
    assert_close(bg.grad.cpu().numpy(), bc.grad.numpy(), TOL, "pred db")


def test_si_loss_golden_and_grad():
    from rpg_ramnet_amd import ops
    z = load_golden("loss_metrics.npz")
    for i in range(3):
        p = torch.from_numpy(z["si%d.pred" % i]).to(dev()).requires_grad_(True)
        t = torch.from_numpy(z["si%d.target" % i]).to(dev())
        l = ops.scale_invariant_loss(p, t, 1.0, 1.0)
        np.testing.assert_allclose(float(l.detach()), float(z["si%d.loss" % i]), rtol=1e-5)
        (3.0 * l).backward()
        np.testing.assert_allclose(p.grad.cpu().numpy(), 3.0 * z["si%d.grad" % i], rtol=1e-4, atol=1e-9)
        l2 = ops.scale_invariant_loss(p.detach(), t, 0.5, 0.85)
        np.testing.assert_allclose(float(l2), float(z["si%d.loss_w05_l085" % i]), rtol=1e-5)


def test_log_and_mse_losses_vs_reference_golden():
    """ops.scale_invariant_log_loss (model/loss.py:12-15) and ops.mse_loss at full / half resolution (model/loss.py:18-19 behind the
    bilinear x0.5 resize of lstm_trainer.py:173-181, fused into the reduction) — value and gradient against vectors produced by the
    reference's own functions (tests/golden/loss_extra.npz); even and odd map sizes, 0 / 20 / 60 % NaN targets."""
    from rpg_ramnet_amd import ops
    z = load_golden("loss_extra.npz")
    for i in range(3):
        t = torch.from_numpy(z["c%d.target" % i]).to(dev())
        for lam in (100, 85):
            p = torch.from_numpy(z["c%d.pred" % i]).to(dev()).requires_grad_(True)
            l = ops.scale_invariant_log_loss(p, t, lam / 100.0)
            np.testing.assert_allclose(float(l.detach()), float(z["c%d.silog%d.loss" % (i, lam)]), rtol=1e-5)
            (2.0 * l).backward()
            np.testing.assert_allclose(p.grad.cpu().numpy(), 2.0 * z["c%d.silog%d.grad" % (i, lam)], rtol=1e-4, atol=1e-8)
        for f in (100, 50):
            p = torch.from_numpy(z["c%d.pred" % i]).to(dev()).requires_grad_(True)
            l = ops.mse_loss(p, t, f / 100.0)
            np.testing.assert_allclose(float(l.detach()), float(z["c%d.mse%d.loss" % (i, f)]), rtol=1e-5)
            (2.0 * l).backward()
            np.testing.assert_allclose(p.grad.cpu().numpy(), 2.0 * z["c%d.mse%d.grad" % (i, f)], rtol=1e-4, atol=1e-10)
    with pytest.raises(NotImplementedError):
        ops.mse_loss(p, t, 0.25)


def test_trainer_loss_assembly_with_mse_term_vs_reference_golden():
    """trainer.sequence_loss with config['mse_loss'] = {weight 0.7, downsampling_factor 0.5}: loss value and the gradient that reaches
    the predictions equal what the reference's LSTMTrainer.calculate_losses / calculate_total_batch_loss produce (loss_extra.npz 'asm.*':
    two supervised maps with weights 1.0 / 0.5, SI loss + mse term).  The model is a stub that replays the stored predictions as the two
    keys of ONE package, so this side divides by L = 1 where the fixture divided by L = 2."""
    from rpg_ramnet_amd import trainer
    z = load_golden("loss_extra.npz")
    preds = [torch.from_numpy(z["asm.pred%d" % l]).to(dev()).requires_grad_(True) for l in range(2)]

    class Replay:
        every_x_rgb_frame, gpu = 1, dev()

        def __call__(self, item, prev_super, prev_lstm):
            return {"events0": preds[0], "image": preds[1]}, {"image": None}, prev_lstm
    item = {"depth_events0": torch.from_numpy(z["asm.target0"]), "depth_image": torch.from_numpy(z["asm.target1"])}
    total, reported = trainer.sequence_loss(Replay(), [item], ["events0", "image"], [float(w) for w in z["asm.weights"]],
                                            mse_loss={"weight": float(z["asm.mse_weight"]), "downsampling_factor": float(z["asm.mse_factor"])})
    np.testing.assert_allclose(float(total.detach()), 2.0 * float(z["asm.loss"]), rtol=1e-5)
    total.backward()
    for l in range(2):
        np.testing.assert_allclose(preds[l].grad.cpu().numpy(), 2.0 * z["asm.grad%d" % l], rtol=1e-4, atol=1e-9)


VOX = ["rand", "rand10", "onebin", "single", "same_t", "corners", "int_ts_pm1"]


@pytest.mark.parametrize("name", VOX)
def test_voxel_grid_golden(name):
    from rpg_ramnet_amd import voxel
    z = load_golden("voxel.npz")
    ev = z["%s.events" % name]
    bins, W, H = [int(v) for v in z["%s.dims" % name]]
    il, vl, okl, ir, vr, okr = voxel_ref.voxel_votes(ev, bins, W, H)
    gl, gr = voxel.voxel_indices(torch.from_numpy(ev).to(dev()), bins, W, H)
    assert np.array_equal(gl.cpu().numpy(), np.where(okl, il, -1)), "left indices must be bit-exact"
    assert np.array_equal(gr.cpu().numpy(), np.where(okr, ir, -1)), "right indices must be bit-exact"
    # ... and against the REFERENCE's own index arrays: what events_to_voxel_grid_pytorch handed to its two index_add_ calls
    # (event_tensor_utils.py:170-181), recorded by tests/golden/make_golden_voxel_indices.py — the valid votes in event order
    zi = load_golden("voxel_indices.npz")
    gln, grn = gl.cpu().numpy(), gr.cpu().numpy()
    assert np.array_equal(gln[gln >= 0], zi["%s.ref_idx_left" % name]), "left indices vs the reference's"
    assert np.array_equal(grn[grn >= 0], zi["%s.ref_idx_right" % name]), "right indices vs the reference's"
    g = voxel.events_to_voxel_grid(torch.from_numpy(ev).to(dev()), bins, W, H).cpu().numpy()
    ref = z["%s.grid_torch" % name]
    assert np.array_equal(g != 0, ref != 0) or np.abs(g - ref).max() < 1e-5
    np.testing.assert_allclose(g, ref, atol=1e-5)        # same votes; atomic accumulation order differs
    if name == "rand":
        n = voxel.normalize_nonzero(torch.from_numpy(ref).to(dev())).cpu().numpy()
        np.testing.assert_allclose(n, z["rand.normalized"], rtol=1e-4, atol=1e-5)


def test_voxel_empty_and_zero_grid():
    from rpg_ramnet_amd import voxel
    g = voxel.events_to_voxel_grid(torch.zeros(0, 4, dtype=torch.float64, device=dev()), 5, 16, 12)
    assert g.shape == (5, 12, 16) and float(g.abs().sum()) == 0.0
    z = voxel.normalize_nonzero(torch.zeros(2, 4, 4, device=dev()))
    assert float(z.abs().sum()) == 0.0


def test_voxel_grids_batched_equals_per_grid():
    """ramnet_voxelize_batch / ramnet_normalize_nonzero_batch: G ragged event lists (one empty, one single event) in one launch
    == the per-grid entry points (same votes; only the order of the atomic sums may differ) and the oracle."""
    from recipe import synth_events
    from rpg_ramnet_amd import voxel
    rng = np.random.default_rng(8)
    W, H, bins = 44, 36, 5
    lists = [synth_events(rng, n, W, H) for n in (5000, 1, 0, 777, 20000)]
    got = voxel.events_to_voxel_grids([torch.from_numpy(e).to(dev()) for e in lists], bins, W, H)
    gotn = voxel.events_to_voxel_grids(lists, bins, W, H, dev(), normalize=True)
    assert got.shape == (5, bins, H, W)
    for g, gn, ev in zip(got, gotn, lists):
        one = voxel.events_to_voxel_grid(torch.from_numpy(ev).to(dev()), bins, W, H)
        np.testing.assert_allclose(g.cpu().numpy(), one.cpu().numpy(), atol=1e-5)
        ref = voxel_ref.events_to_voxel_grid(ev, bins, W, H) if len(ev) else np.zeros((bins, H, W), np.float32)
        np.testing.assert_allclose(g.cpu().numpy(), ref, atol=1e-5)
        np.testing.assert_allclose(gn.cpu().numpy(), voxel_ref.normalize_nonzero(ref), rtol=1e-4, atol=2e-5)


def test_voxel_row_band_kernel_equals_oracle():
    """Launches of >= 16 grids run the row-band kernel (votes resolved in LDS, every cell written once, no zero-fill): 20 ragged
    lists at the real sensor size (260 x 346: 12 bands, the last one partial) incl. an empty list, a single event, events on the
    last row / column, the band boundaries, polarity 0 and fractional coordinates — against the oracle per grid, and against the
    global-atomic form (the per-grid entry point)."""
    from recipe import synth_events
    from rpg_ramnet_amd import voxel
    rng = np.random.default_rng(18)
    W, H, bins = 346, 260, 5
    sizes = [30000, 1, 0, 777, 20000, 5, 4096, 4097, 8191, 12345] + [3000] * 10
    lists = [synth_events(rng, n, W, H) for n in sizes]
    lists[3][:7, 1] = W - 1
    lists[3][:7, 2] = H - 1                    # last row / column
    lists[3][7:11, 2] = [0.0, 22.0, 23.0, 259.0]                # band boundaries (23 rows per band)
    lists[4][:4, 2] = [-0.5, 0.5, 22.9, 259.5]                  # (long long)y truncates towards zero: -0.5 is row 0
    lists[4][5:7, 1] = [345.9, 0.4]
    assert len(lists) >= 16
    got = voxel.events_to_voxel_grids([torch.from_numpy(e).to(dev()) for e in lists], bins, W, H)
    for i, (g, ev) in enumerate(zip(got, lists)):
        ref = voxel_ref.events_to_voxel_grid(ev, bins, W, H) if len(ev) else np.zeros((bins, H, W), np.float32)
        np.testing.assert_allclose(g.cpu().numpy(), ref, atol=2e-5, err_msg="grid %d vs oracle" % i)
        one = voxel.events_to_voxel_grid(torch.from_numpy(ev).to(dev()), bins, W, H)
        np.testing.assert_allclose(g.cpu().numpy(), one.cpu().numpy(), atol=2e-5, err_msg="grid %d vs the atomic form" % i)


def test_si_loss_is_bit_reproducible_and_reusable():
    """The single-launch statistics (per-workgroup partials in a library scratch, folded in a fixed order by the last arriver, self-resetting
    ticket): the same inputs give the same bits call after call, different inputs in between do not leak, and a full-resolution batch
    with 20 % NaN targets matches the float64 value."""
    from rpg_ramnet_amd import ops
    g = torch.Generator(device=dev()).manual_seed(5)
    p1, t1 = torch.rand(8, 1, 256, 344, device=dev(), generator=g), torch.rand(8, 1, 256, 344, device=dev(), generator=g)
    t1[torch.rand(t1.shape, device=dev(), generator=g) < 0.2] = float("nan")
    p2, t2 = torch.rand(3, 1, 17, 23, device=dev(), generator=g), torch.rand(3, 1, 17, 23, device=dev(), generator=g)
    a = ops.scale_invariant_loss(p1, t1)
    small = ops.scale_invariant_loss(p2, t2)
    b = ops.scale_invariant_loss(p1, t1)
    assert torch.equal(a, b)
    d = (p1 - t1).double()[~torch.isnan(t1)]
    np.testing.assert_allclose(float(a), float((d * d).mean() - d.mean() ** 2), rtol=1e-6)
    d2 = (p2 - t2).double()
    np.testing.assert_allclose(float(small), float((d2 * d2).mean() - d2.mean() ** 2), rtol=1e-6)


def test_voxel_sorted_form_exact_and_bit_reproducible():
    """Batches of >= 2 lists run the sorted form (csrc/loss_voxel.hip: chunk-local counting sort by row band, votes summed as 64-bit fixed
    point with integer LDS atomics).  (a) the same launch twice gives bit-identical grids; (b) every cell equals the EXACT sum of the
    reference's fp32 votes (oracle votes accumulated in float64) rounded once — within one fp32 ulp of the cell (the fixed point is 2^-40:
    only votes below 2^-16 are rounded, by < 1e-12); (c) events outside the image and NaN coordinates cast no vote; a hot pixel that fires
    6000 times with mixed polarities is summed without loss."""
    from recipe import synth_events
    from rpg_ramnet_amd import voxel
    rng = np.random.default_rng(29)
    W, H, bins = 346, 260, 5
    sizes = [60000, 2, 0, 9000, 8192, 8193]
    lists = [synth_events(rng, n, W, H) for n in sizes]
    hot = lists[0]
    hot[1000:7000, 1], hot[1000:7000, 2] = 17.0, 133.0                       # one hot pixel, 6000 events, random polarity
    lists[3][:5, 1] = [-3.0, 346.0, 1e9, 5.0, float("nan")]                  # outside / NaN x
    lists[3][5:9, 2] = [-1.0, 260.0, -7.5, 259.99]                           # outside y (and the last row)
    dev_lists = [torch.from_numpy(e).to(dev()) for e in lists]
    a = voxel.events_to_voxel_grids(dev_lists, bins, W, H)
    b = voxel.events_to_voxel_grids(dev_lists, bins, W, H)
    assert torch.equal(a, b), "the sorted form must be bit-reproducible"
    for i, (g, ev) in enumerate(zip(a, lists)):
        if len(ev) == 0:
            assert float(g.abs().sum()) == 0.0
            continue
        with np.errstate(invalid="ignore"):
            il, vl, okl, ir, vr, okr = voxel_ref.voxel_votes(ev, bins, W, H)
            inside = (ev[:, 1] > -1) & (ev[:, 1] < W) & (ev[:, 2] > -1) & (ev[:, 2] < H)        # (the reference would raise on the others)
        exact = np.zeros(bins * H * W, np.float64)
        np.add.at(exact, il[okl & inside], vl[okl & inside].astype(np.float64))
        np.add.at(exact, ir[okr & inside], vr[okr & inside].astype(np.float64))
        want = exact.astype(np.float32).reshape(bins, H, W)
        got = g.cpu().numpy()
        ulp = np.spacing(np.maximum(np.abs(want), np.float32(1e-30)))
        assert np.all(np.abs(got - want) <= ulp), "grid %d: max |diff| %.3e" % (i, float(np.abs(got - want).max()))
        assert np.array_equal(got == 0, want == 0) or np.abs(got[(got == 0) != (want == 0)]).max() < 1e-12


def test_voxel_row_band_fallback_form():
    """ramnet_set_option("voxel_sorted", 0) (and any launch inside a stream capture that finds no scratch) keeps the row-band kernels:
    same grids up to the order of their fp32 LDS atomics."""
    from rpg_ramnet_amd import voxel
    from recipe import synth_events
    L = Hh.lib()
    rng = np.random.default_rng(3)
    lists = [synth_events(rng, n, 346, 260) for n in [5000, 0, 1, 7000] + [1500] * 14]
    assert L.ramnet_get_option(b"voxel_sorted") == 1 and L.ramnet_get_option(b"no_such_option") == -1
    Hh.check(L.ramnet_set_option(b"voxel_sorted", 0), "set_option")
    try:
        got = voxel.events_to_voxel_grids([torch.from_numpy(e).to(dev()) for e in lists], 5, 346, 260)
    finally:
        Hh.check(L.ramnet_set_option(b"voxel_sorted", 1), "set_option")
    for g, ev in zip(got, lists):
        ref = voxel_ref.events_to_voxel_grid(ev, 5, 346, 260) if len(ev) else np.zeros((5, 260, 346), np.float32)
        np.testing.assert_allclose(g.cpu().numpy(), ref, atol=2e-5)
    assert L.ramnet_set_option(b"no_such_option", 1) != 0


@pytest.mark.parametrize("B,H,W,nan_frac", [(2, 32, 48, 0.0), (3, 24, 40, 0.2), (1, 16, 16, 0.5)])
def test_multi_scale_grad_loss_vs_oracle(B, H, W, nan_frac):
    """Next-row component.  PARITY UNPINNED against kornia itself; checked against the oracle restatement
    (value and gradient through torch autograd in float64)."""
    from rpg_ramnet_amd import ops
    rng = np.random.default_rng(5)
    p = rng.random((B, 1, H, W)).astype(np.float32)
    t = rng.random((B, 1, H, W)).astype(np.float32)
    t[rng.random(t.shape) < nan_frac] = np.nan
    pg = torch.from_numpy(p).to(dev()).requires_grad_(True)
    l = ops.multi_scale_grad_loss(pg, torch.from_numpy(t).to(dev()))
    (2.5 * l).backward()
    pc = torch.from_numpy(p).double().requires_grad_(True)
    lr = loss_ref.multi_scale_grad_loss(pc, torch.from_numpy(t).double())
    (2.5 * lr).backward()
    np.testing.assert_allclose(float(l.detach()), float(lr.detach()), rtol=2e-5)
    assert_close(pg.grad.cpu().numpy(), pc.grad.numpy(), 1e-4, "msg grad")


def test_depth_metrics_vs_reference_golden():
    """Next-row component: metric depth + Abs-Rel on device vs the reference's prepare_depth_data +
    abs_rel_diff (golden), and vs the oracle / a numpy restatement of evaluation.py:201-241 for the remaining sums."""
    from rpg_ramnet_amd import metrics
    z = load_golden("loss_metrics.npz")
    eps = 1e-5

    def reference_sums(t, p, mask):
        """add_to_metrics (evaluation.py:211-238) on the masked pixels; NaN targets dropped where metric.py drops them."""
        tm, pm = t[mask], p[mask]
        ok = ~np.isnan(tm)
        ratio = np.maximum(tm / (pm + eps), pm / (tm + eps))
        ld = np.log(tm[ok] + eps) - np.log(pm[ok] + eps)
        with np.errstate(invalid="ignore"):
            out = {"threshold_delta_1.25": np.mean(ratio <= 1.25), "threshold_delta_1.25^2": np.mean(ratio <= 1.25 ** 2),
                   "threshold_delta_1.25^3": np.mean(ratio <= 1.25 ** 3)}
        out.update({"abs_rel_diff": loss_ref.abs_rel_diff(pm, tm), "squ_rel_diff": loss_ref.squ_rel_diff(pm, tm),
                    "RMS_linear": loss_ref.rms_linear(pm, tm), "RMS_log": np.sqrt((ld ** 2).mean()),
                    "SILog": loss_ref.scale_invariant_error(np.log(pm + eps), np.log(tm + eps)),
                    "mean_depth_error": loss_ref.mean_error(pm, tm)})
        return out, int(ok.sum())

    for clip, reg in [(80, 3.70378), (1000, 5.70378)]:
        t_in, p_in = z["depth%d.target_in" % clip], z["depth%d.pred_in" % clip]
        m = metrics.depth_metrics(torch.from_numpy(p_in).to(dev()), torch.from_numpy(t_in).to(dev()), float(clip), reg)
        np.testing.assert_allclose(m["abs_rel_diff"], float(z["depth%d.abs_rel" % clip]), rtol=2e-5)
        t, p = loss_ref.prepare_depth_data(t_in, p_in, float(clip), reg)
        want, n = reference_sums(t.astype(np.float64), p.astype(np.float64), np.ones(t.shape, bool))
        assert m["n"] == n == t_in.size
        for k, v in want.items():
            np.testing.assert_allclose(m[k], v, rtol=5e-5, atol=1e-7, err_msg=k)
    tn = z["depth80.target_in"].copy()
    tn[::3, ::2] = np.nan
    m = metrics.depth_metrics(torch.from_numpy(z["depth80.pred_in"]).to(dev()), torch.from_numpy(tn).to(dev()), 80.0, 3.70378, cutoff=30.0)
    t, p = loss_ref.prepare_depth_data(tn, z["depth80.pred_in"], 80.0, 3.70378)
    want, n = reference_sums(t.astype(np.float64), p.astype(np.float64), np.nan_to_num(t) < 30.0)     # strict <, NaN -> inside
    assert m["n"] == n
    for k, v in want.items():
        if k == "RMS_log":                      # plain np.mean over the mask in the reference: a NaN target inside -> NaN
            assert np.isnan(m[k]) and np.isnan(m["median_diff"])
            continue
        np.testing.assert_allclose(m[k], v, rtol=5e-5, atol=1e-7, err_msg=k)


def test_depth_metrics_table_vs_reference_add_to_metrics():
    """All ten rows of the reference's evaluation table (evaluation.py:201-241) against ITS `add_to_metrics`, run on seeded maps by
    tests/golden/make_golden_eval_metrics.py: `median_diff`, and `RMS_log` / `median_diff` NaN as soon as the mask holds a NaN target."""
    from rpg_ramnet_amd import metrics
    z = load_golden("eval_metrics.npz")
    keys = ("abs_rel_diff", "squ_rel_diff", "RMS_linear", "RMS_log", "SILog", "mean_depth_error", "median_diff", "threshold_delta_1.25",
            "threshold_delta_1.25^2", "threshold_delta_1.25^3")
    for tag in ("plain", "cut30", "nan", "nan_cut20", "mvsec"):
        clip, reg, cutoff = (float(v) for v in z[tag + ".params"])
        m = metrics.depth_metrics(torch.from_numpy(z[tag + ".pred_in"]).to(dev()), torch.from_numpy(z[tag + ".target_in"]).to(dev()),
                                  clip, reg, cutoff=cutoff)
        for k in keys:
            want = float(z["%s.%s" % (tag, k)])
            if np.isnan(want):
                assert k in ("RMS_log", "median_diff") and np.isnan(m[k]), (tag, k, m[k])
            else:
                np.testing.assert_allclose(m[k], want, rtol=1e-4, atol=2e-5, err_msg="%s %s" % (tag, k))
        assert np.isnan(float(z[tag + ".RMS_log"])) == tag.startswith("nan")
