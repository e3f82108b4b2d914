"""GPU: the C ABI driven DIRECTLY with ctypes and raw device pointers (no rpg_ramnet_amd.ops in between), the way a
maintainer of another host language would bind it."""
import ctypes as C

import numpy as np
import pytest
import torch

from rpg_ramnet_amd import _hip
import torch_restatements as tr

pytestmark = pytest.mark.gpu


def ptr(t):
    return C.c_void_p(t.data_ptr())


def test_conv_forward_through_raw_descriptor():
    L = _hip.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    B, H, W, Cin, Cout, k = 2, 12, 20, 32, 64, 3
    x = torch.randn(B, H, W, Cin, device=dev)                       # NHWC
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.1              # OIHW
    b = torch.randn(Cout, device=dev) * 0.1
    n = L.ramnet_packed_weight_elems(Cout, Cin, k, k, 0, 1)
    wp = torch.empty(n, device=dev)
    assert L.ramnet_pack_weight(ptr(w), ptr(wp), Cout, Cin, k, k, 0, 1, st) == 0
    y = torch.empty(B, H, W, Cout, device=dev)
    d = _hip.ConvDesc()
    d.x0, d.ld0, d.C0, d.in_mode = ptr(x), Cin, Cin, _hip.IN_PLAIN
    d.B, d.Hin, d.Win, d.stride = B, H, W, 1
    taps = [(kh - 1, kw - 1, kh * k + kw) for kh in range(k) for kw in range(k)]
    d.ntaps = len(taps)
    for i, (dy, dx, wt) in enumerate(taps):
        d.dy[i], d.dx[i], d.wtap[i] = dy, dx, wt
    d.w, d.bias, d.Cout = ptr(wp), ptr(b), Cout
    d.Ho, d.Wo, d.HoF, d.WoF = H, W, H, W
    d.osy, d.osx, d.ooy, d.oox = 1, 1, 0, 0
    d.epi, d.beta, d.out, d.ldo = _hip.EPI_RELU, 0.0, ptr(y), Cout
    rc = L.ramnet_conv_launch(C.byref(d), st)
    assert rc == 0, L.ramnet_last_error()
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).cpu(), w.cpu(), b.cpu(), 1, 1))
    err = (y.permute(0, 3, 1, 2).cpu() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-4
    # a descriptor the kernel cannot serve is refused with a message, nothing is launched
    d.ld0 = 30
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001 and b"bad argument" in L.ramnet_last_error()


def test_loss_and_voxel_entry_points_raw():
    L = _hip.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(1)
    pred, tgt = torch.rand(4096, device=dev, generator=g), torch.rand(4096, device=dev, generator=g)
    tgt[::7] = float("nan")
    stats = torch.empty(4, dtype=torch.float64, device=dev)
    loss = torch.empty((), device=dev)
    assert L.ramnet_si_loss_fwd(ptr(pred), ptr(tgt), 4096, 1.0, 1.0, ptr(stats), ptr(loss), st) == 0
    d = (pred - tgt)[~torch.isnan(tgt)].double()
    np.testing.assert_allclose(float(loss), float((d * d).mean() - d.mean() ** 2), rtol=1e-5)
    assert int(stats[2]) == int((~torch.isnan(tgt)).sum())
    ev = torch.tensor([[0.0, 1, 1, 1], [0.5, 2, 3, 0], [1.0, 7, 5, 1]], dtype=torch.float64, device=dev)
    grid = torch.empty(3, 6, 8, device=dev)
    assert L.ramnet_voxelize(ptr(ev), 3, 3, 8, 6, ptr(grid), st) == 0
    exp = np.zeros((3, 6, 8), np.float32)
    exp[0, 1, 1], exp[1, 3, 2], exp[2, 5, 7] = 1.0, -1.0, 1.0          # t -> bins 0, 1, 2 exactly (dt = 0)
    assert np.array_equal(grid.cpu().numpy(), exp)


def test_winograd_conv_and_wgrad_through_raw_descriptors():
    """RAMNET_ALGO_WINOGRAD end to end with raw pointers: pack U = G g G^T, forward launch, backward-weights launch into
    the transformed-domain workspace, fold into OIHW; checked against float64 autograd."""
    L = _hip.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(3)
    B, H, W, Cin, Cout = 2, 14, 22, 64, 96
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
    b = torch.randn(Cout, device=dev) * 0.1
    wp = torch.empty(L.ramnet_packed_weight_elems_wino(Cout, Cin, 0, 1), device=dev)
    assert L.ramnet_pack_weight_wino(ptr(w), ptr(wp), Cout, Cin, 0, 1, st) == 0
    y = torch.empty(B, H, W, Cout, device=dev)
    d = _hip.ConvDesc()
    d.x0, d.ld0, d.C0, d.in_mode = ptr(x), Cin, Cin, _hip.IN_PLAIN
    d.B, d.Hin, d.Win, d.stride, d.ntaps = B, H, W, 1, 9
    for i in range(9):
        d.dy[i], d.dx[i], d.wtap[i] = i // 3 - 1, i % 3 - 1, i
    d.w, d.bias, d.Cout = ptr(wp), ptr(b), Cout
    d.Ho, d.Wo, d.HoF, d.WoF = H, W, H, W
    d.osy, d.osx = 1, 1
    d.epi, d.out, d.ldo, d.algo = _hip.EPI_LINEAR, ptr(y), Cout, _hip.ALGO_WINOGRAD
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    xr = x.permute(0, 3, 1, 2).cpu().double()
    wr, br = w.cpu().double().requires_grad_(True), b.cpu().double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, wr, br, 1, 1)
    assert float((y.permute(0, 3, 1, 2).cpu() - ref.detach()).abs().max() / ref.abs().max()) < 2e-4
    # the Winograd kernel only serves the dense 3x3 stride-1 window
    d.stride = 2
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001
    # backward-weights
    dy = torch.randn(B, H, W, Cout, device=dev)
    (ref * dy.permute(0, 3, 1, 2).cpu().double()).sum().backward()
    ws, dbias = torch.zeros(16 * Cin * Cout, device=dev), torch.zeros(Cout, device=dev)
    g = _hip.WgradDesc()
    g.x0, g.ld0, g.C0, g.in_mode = ptr(x), Cin, Cin, _hip.IN_PLAIN
    g.B, g.Hin, g.Win, g.ntaps, g.stride = B, H, W, 9, 1
    for i in range(9):
        g.dy[i], g.dx[i] = i // 3 - 1, i % 3 - 1
    g.dout, g.ldg, g.Cout, g.Ho, g.Wo = ptr(dy), Cout, Cout, H, W
    g.dw, g.dbias, g.algo = ptr(ws), ptr(dbias), _hip.ALGO_WINOGRAD
    assert L.ramnet_wgrad_launch(C.byref(g), st) == 0, L.ramnet_last_error()
    grad = torch.zeros(Cout, Cin, 3, 3, device=dev)
    assert L.ramnet_unpack_wgrad_wino(ptr(ws), ptr(grad), Cout, Cin, Cin, Cout, 0, st) == 0
    assert float((grad.cpu() - wr.grad).abs().max() / wr.grad.abs().max()) < 2e-4
    assert float((dbias.cpu() - br.grad).abs().max() / br.grad.abs().max()) < 2e-4


def test_split_operand_winograd_through_raw_descriptors():
    """RAMNET_ALGO_WINOGRAD_2X4_SPLIT (ABI 22) with raw pointers: the library accepts the descriptor (ramnet_conv_wino_split_ok), packs the
    three bf16 planes (ramnet_pack_weight_wino2x4_split: 6 bytes per Winograd-domain weight), runs conv_wino_r6s_kernel and matches float64 at the
    bar of the exact-fp32 algorithms; a concatenation whose boundary is not a multiple of 16 channels is refused (16-channel chunks)."""
    L = _hip.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(4)
    B, H, W, C0, C1, Cout = 2, 18, 26, 32, 48, 128
    Cin = C0 + C1
    x0, x1 = torch.randn(B, H, W, C0, device=dev), torch.randn(B, H, W, C1, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
    b = torch.randn(Cout, device=dev) * 0.1
    n4 = L.ramnet_packed_weight_elems_wino2x4_split(Cout, Cin, 0)
    assert n4 * 4 == ((Cin + 15) // 16) * (Cout // 64) * 24 * 2 * 3 * 1024          # [chunk][block][24 positions][2 halves][3 planes][64 lanes x 16 B]
    wp = torch.empty(n4, device=dev)
    assert L.ramnet_pack_weight_wino2x4_split(ptr(w), ptr(wp), Cout, Cin, 0, st) == 0
    y = torch.full((B, H, W, Cout), float("nan"), device=dev)
    d = _hip.ConvDesc()
    d.x0, d.x1, d.ld0, d.ld1, d.C0, d.C1, d.in_mode = ptr(x0), ptr(x1), C0, C1, C0, C1, _hip.IN_CAT
    d.B, d.Hin, d.Win, d.stride, d.ntaps = B, H, W, 1, 9
    for i in range(9):
        d.dy[i], d.dx[i], d.wtap[i] = i // 3 - 1, i % 3 - 1, i
    d.w, d.bias, d.Cout = ptr(wp), ptr(b), Cout
    d.Ho, d.Wo, d.HoF, d.WoF = H, W, H, W
    d.osy, d.osx = 1, 1
    d.epi, d.out, d.ldo, d.algo = _hip.EPI_RELU, ptr(y), Cout, _hip.ALGO_WINOGRAD
    assert L.ramnet_conv_wino_split_ok(C.byref(d), 1) == 1
    d.algo = _hip.ALGO_WINOGRAD_2X4_SPLIT
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    assert L.ramnet_last_kernel().decode().startswith("conv_wino_r6s_kernel<")
    xr = torch.cat([x0, x1], 3).permute(0, 3, 1, 2).cpu().double()
    ref = torch.relu(torch.nn.functional.conv2d(xr, w.cpu().double(), b.cpu().double(), 1, 1))
    assert float((y.permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-4
    # a chunk of 16 channels must lie in one tensor of the concatenation
    d.C0, d.C1, d.ld0 = 24, 56, 24
    d.algo = _hip.ALGO_WINOGRAD
    assert L.ramnet_conv_wino_split_ok(C.byref(d), 1) == 0
    d.algo = _hip.ALGO_WINOGRAD_2X4_SPLIT
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001


def test_space_to_depth_view_through_raw_descriptors():
    """RAMNET_IN_S2D / out_s2d: the Winograd kernels read and write the space-to-depth view of a full-resolution NHWC
    tensor in place; checked against the materialised view (ramnet_space_to_depth2) fed to the same kernels and against
    torch.  (ramnet_hip.h: ramnet_in_mode, ramnet_conv_desc.out_s2d.)"""
    L, st = _hip.lib(), None
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, H, W, Cc, Cout = 2, 12, 20, 32, 64           # logical 3x3 layer: [B, 6, 10, 128] -> [B, 6, 10, 64]
    Hl, Wl, Cin = H // 2, W // 2, 4 * Cc
    x = torch.randn(B, H, W, Cc, device=dev)
    deep = torch.empty(B, Hl, Wl, Cin, device=dev)
    assert L.ramnet_space_to_depth2(ptr(x), ptr(deep), B, H, W, Cc, 0, st) == 0
    ref_deep = x.view(B, Hl, 2, Wl, 2, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B, Hl, Wl, Cin)
    assert torch.equal(deep, ref_deep)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
    wp = torch.empty(L.ramnet_packed_weight_elems_wino(Cout, Cin, 0, 1), device=dev)
    assert L.ramnet_pack_weight_wino(ptr(w), ptr(wp), Cout, Cin, 0, 1, st) == 0

    def conv_desc(src, ld, c0, mode, out, cout, wpk):
        d = _hip.ConvDesc()
        d.x0, d.ld0, d.C0, d.in_mode = ptr(src), ld, c0, mode
        d.B, d.Hin, d.Win, d.stride, d.ntaps = B, Hl, Wl, 1, 9
        for i in range(9):
            d.dy[i], d.dx[i], d.wtap[i] = i // 3 - 1, i % 3 - 1, i
        d.w, d.Cout = ptr(wpk), cout
        d.Ho, d.Wo, d.HoF, d.WoF = Hl, Wl, out.shape[1], out.shape[2]
        d.osy, d.osx = 1, 1
        d.epi, d.out, d.ldo, d.algo = _hip.EPI_LINEAR, ptr(out), out.shape[3], _hip.ALGO_WINOGRAD
        return d

    # forward: in-place view == materialised view, bit for bit (same kernel, same operands)
    y_view, y_deep = torch.empty(B, Hl, Wl, Cout, device=dev), torch.empty(B, Hl, Wl, Cout, device=dev)
    d = conv_desc(x, Cc, Cc, _hip.IN_S2D, y_view, Cout, wp)
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    d = conv_desc(deep, Cin, Cin, _hip.IN_PLAIN, y_deep, Cout, wp)
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    assert torch.equal(y_view, y_deep)
    ref = torch.nn.functional.conv2d(ref_deep.permute(0, 3, 1, 2).cpu().double(), w.cpu().double(), None, 1, 1)
    assert float((y_view.permute(0, 3, 1, 2).cpu() - ref).abs().max() / ref.abs().max()) < 2e-4
    # the direct kernel does not know the view
    d = conv_desc(x, Cc, Cc, _hip.IN_S2D, y_view, Cout, wp)
    d.algo = _hip.ALGO_DIRECT
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001

    # backward-data: 4*Cc logical output channels stored at their full-resolution pixels
    dy = torch.randn(B, Hl, Wl, Cout, device=dev)
    wt = torch.empty(L.ramnet_packed_weight_elems_wino(Cout, Cin, 1, 1), device=dev)
    assert L.ramnet_pack_weight_wino(ptr(w), ptr(wt), Cout, Cin, 1, 1, st) == 0
    dx_view, dx_deep = torch.empty(B, H, W, Cc, device=dev), torch.empty(B, Hl, Wl, Cin, device=dev)
    d = conv_desc(dy, Cout, Cout, _hip.IN_PLAIN, dx_view, Cin, wt)
    d.out_s2d = Cc
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    d = conv_desc(dy, Cout, Cout, _hip.IN_PLAIN, dx_deep, Cin, wt)
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    back = torch.empty(B, H, W, Cc, device=dev)
    assert L.ramnet_space_to_depth2(ptr(dx_deep), ptr(back), B, H, W, Cc, 1, st) == 0
    assert torch.equal(dx_view, back)
    d = conv_desc(dy, Cout, Cout, _hip.IN_PLAIN, dx_view, Cin, wt)
    d.out_s2d, d.epi = Cc, _hip.EPI_RELU                     # only the plain store knows the view
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001

    # backward-weights: same workspace from the view as from the materialised tensor (atomic order aside)
    def wgrad(src, ld, c0, mode):
        ws = torch.zeros(16 * Cin * Cout, device=dev)
        g = _hip.WgradDesc()
        g.x0, g.ld0, g.C0, g.in_mode = ptr(src), ld, c0, mode
        g.B, g.Hin, g.Win, g.ntaps, g.stride = B, Hl, Wl, 9, 1
        for i in range(9):
            g.dy[i], g.dx[i] = i // 3 - 1, i % 3 - 1
        g.dout, g.ldg, g.Cout, g.Ho, g.Wo = ptr(dy), Cout, Cout, Hl, Wl
        g.dw, g.algo = ptr(ws), _hip.ALGO_WINOGRAD
        assert L.ramnet_wgrad_launch(C.byref(g), st) == 0, L.ramnet_last_error()
        return ws
    a, b = wgrad(x, Cc, Cc, _hip.IN_S2D), wgrad(deep, Cin, Cin, _hip.IN_PLAIN)
    torch.cuda.synchronize()
    assert float((a - b).abs().max() / b.abs().max()) < 1e-5

    # s2d_5x5: with the 3x3 view of a REAL 5x5 stride-2 filter (11 of 36 slices zero) the kernels skip the Winograd positions the
    # zeros annihilate — the results are those of the dense evaluation, bit for bit (forward, backward-data); backward-weights is
    # dense either way and its 5x5 gradient only reads the slices the filter owns
    from rpg_ramnet_amd.ops import s2d_weights, s2d_weights_adjoint
    w3 = s2d_weights(torch.randn(Cout, Cc, 5, 5, device=dev) * 0.1).contiguous()
    assert L.ramnet_pack_weight_wino(ptr(w3), ptr(wp), Cout, Cin, 0, 1, st) == 0
    assert L.ramnet_pack_weight_wino(ptr(w3), ptr(wt), Cout, Cin, 1, 1, st) == 0
    outs = []
    for flag in (0, 1):
        yv, dxv = torch.empty(B, Hl, Wl, Cout, device=dev), torch.empty(B, H, W, Cc, device=dev)
        d = conv_desc(x, Cc, Cc, _hip.IN_S2D, yv, Cout, wp)
        d.s2d_5x5 = flag
        assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
        d = conv_desc(dy, Cout, Cout, _hip.IN_PLAIN, dxv, Cin, wt)
        d.out_s2d, d.s2d_5x5 = Cc, flag
        assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
        ws = torch.zeros(16 * Cin * Cout, device=dev)
        g = _hip.WgradDesc()
        g.x0, g.ld0, g.C0, g.in_mode = ptr(x), Cc, Cc, _hip.IN_S2D
        g.B, g.Hin, g.Win, g.ntaps, g.stride = B, Hl, Wl, 9, 1
        for i in range(9):
            g.dy[i], g.dx[i] = i // 3 - 1, i % 3 - 1
        g.dout, g.ldg, g.Cout, g.Ho, g.Wo = ptr(dy), Cout, Cout, Hl, Wl
        g.dw, g.algo = ptr(ws), _hip.ALGO_WINOGRAD
        assert L.ramnet_wgrad_launch(C.byref(g), st) == 0, L.ramnet_last_error()
        g3 = torch.zeros(Cout, Cin, 3, 3, device=dev)
        assert L.ramnet_unpack_wgrad_wino(ptr(ws), ptr(g3), Cout, Cin, Cin, Cout, 0, st) == 0
        outs.append((yv, dxv, s2d_weights_adjoint(g3, Cc)))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][2] - outs[1][2]).abs().max() / outs[0][2].abs().max()) < 1e-5
    d = conv_desc(deep, Cin, Cin, _hip.IN_PLAIN, y_deep, Cout, wp)       # the flag needs the in-place view (its group geometry)
    d.s2d_5x5 = 1
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001


@pytest.mark.parametrize("cin,cpad", [(5, 8), (1, 4), (3, 4)])
def test_head_layer_through_raw_descriptors(cin, cpad):
    """RAMNET_ALGO_HEAD: the 5x5 head layers (statenet.py:160-175) forward and backward-weights with raw descriptors,
    against torch in float64 and against the generic kernel's workspace layout."""
    L, st = _hip.lib(), None
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    B, H, W, Cout = 2, 21, 45, 32
    x = torch.zeros(B, H, W, cpad, device=dev)
    x[..., :cin] = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(Cout, cin, 5, 5, device=dev) * 0.2
    b = torch.randn(Cout, device=dev) * 0.1
    assert L.ramnet_head_supported(cin, Cout) == 1 and L.ramnet_head_supported(2, Cout) == 0 and L.ramnet_head_supported(cin, 64) == 0
    wp = torch.empty(L.ramnet_packed_weight_elems_head(cin), device=dev)
    assert L.ramnet_pack_weight_head(ptr(w), ptr(wp), Cout, cin, st) == 0
    y = torch.empty(B, H, W, Cout, device=dev)
    d = _hip.ConvDesc()
    d.x0, d.ld0, d.C0, d.in_mode = ptr(x), cpad, cpad, _hip.IN_PLAIN
    d.B, d.Hin, d.Win, d.stride, d.ntaps = B, H, W, 1, 25
    for i in range(25):
        d.dy[i], d.dx[i], d.wtap[i] = i // 5 - 2, i % 5 - 2, i
    d.w, d.bias, d.Cout = ptr(wp), ptr(b), Cout
    d.Ho, d.Wo, d.HoF, d.WoF = H, W, H, W
    d.osy, d.osx = 1, 1
    d.epi, d.out, d.ldo, d.algo, d.head_cin = _hip.EPI_RELU, ptr(y), Cout, _hip.ALGO_HEAD, cin
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    xr = x[..., :cin].permute(0, 3, 1, 2).cpu().double()
    wr, br = w.cpu().double().requires_grad_(True), b.cpu().double().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.conv2d(xr, wr, br, 1, 2))
    assert float((y.permute(0, 3, 1, 2).cpu() - ref.detach()).abs().max() / ref.abs().max()) < 1e-5
    d.stride = 2                                    # only the dense stride-1 window
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001
    d.stride, d.head_cin = 1, 2                     # unsupported channel count
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001

    dy = torch.randn(B, H, W, Cout, device=dev)
    (ref * dy.permute(0, 3, 1, 2).cpu().double()).sum().backward()
    ws, dbias = torch.zeros(25 * cpad * Cout, device=dev), torch.zeros(Cout, device=dev)
    g = _hip.WgradDesc()
    g.x0, g.ld0, g.C0, g.in_mode = ptr(x), cpad, cpad, _hip.IN_PLAIN
    g.B, g.Hin, g.Win, g.ntaps, g.stride = B, H, W, 25, 1
    for i in range(25):
        g.dy[i], g.dx[i] = i // 5 - 2, i % 5 - 2
    g.dout, g.ldg, g.gmask, g.ldgm, g.Cout, g.Ho, g.Wo = ptr(dy), Cout, ptr(y), Cout, Cout, H, W
    g.dw, g.dbias, g.algo, g.head_cin = ptr(ws), ptr(dbias), _hip.ALGO_HEAD, cin
    assert L.ramnet_wgrad_launch(C.byref(g), st) == 0, L.ramnet_last_error()
    grad = torch.zeros(Cout, cin, 5, 5, device=dev)
    assert L.ramnet_unpack_wgrad(ptr(ws), ptr(grad), Cout, cin, cpad, Cout, 0, 5, 5, st) == 0      # same layout as the generic kernel
    assert float((grad.cpu() - wr.grad).abs().max() / wr.grad.abs().max()) < 1e-4
    assert float((dbias.cpu() - br.grad).abs().max() / br.grad.abs().max()) < 1e-4


@pytest.mark.parametrize("Cin,Cout", [(64, 64), (64, 32)])
def test_folded_upsample_conv_winograd_through_raw_descriptors(Cin, Cout):
    """RAMNET_ALGO_WINOGRAD24 with raw descriptors: forward of conv5x5(bilinear_x2(x)) away from the border (the border
    corrections are a separate GEMM, `frame`), and the Winograd-domain backward-weights, against torch in float64; the C pack
    function against the documented layout (ops.pack_fold_wino)."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    L, st = _hip.lib(), None
    dev = torch.device("cuda:0")
    torch.manual_seed(13)
    B, H, W = 2, 9, 14
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 5, 5, device=dev) * 0.1
    b = torch.randn(Cout, device=dev) * 0.1
    assert L.ramnet_fold_wino_supported(Cout, Cin) == 1 and L.ramnet_fold_wino_supported(48, Cin) == 0
    wp = torch.empty(L.ramnet_packed_weight_elems_fold_wino(Cout, Cin), device=dev)
    assert L.ramnet_pack_weight_fold_wino(ptr(w), ptr(wp), Cout, Cin, st) == 0
    assert float((wp - tr.pack_fold_wino(w)).abs().max()) < 1e-6
    xpad = torch.empty(B, H + 4, W + 4, Cin, device=dev)
    assert L.ramnet_pad2_sum(ptr(x), None, ptr(xpad), B, H, W, Cin, st) == 0
    y = torch.zeros(B, 2 * H, 2 * W, Cout, device=dev)
    d = _hip.ConvDesc()
    d.x0, d.ld0, d.C0, d.in_mode = ptr(xpad), Cin, Cin, _hip.IN_PLAIN
    d.B, d.Hin, d.Win, d.stride, d.ntaps = B, H + 4, W + 4, 1, 16
    d.w, d.bias, d.Cout = ptr(wp), ptr(b), Cout
    d.Ho, d.Wo, d.HoF, d.WoF = H, W, 2 * H, 2 * W
    d.osy, d.osx = 1, 1
    d.epi, d.out, d.ldo, d.algo = _hip.EPI_LINEAR, ptr(y), Cout, _hip.ALGO_WINOGRAD24
    assert L.ramnet_conv_launch(C.byref(d), st) == 0, L.ramnet_last_error()
    xr = x.permute(0, 3, 1, 2).cpu().double()
    wr, br = w.cpu().double(), b.cpu().double()
    ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False), wr, br, 1, 2)
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref)[:, :, 2:-2, 2:-2].abs().max() / ref.abs().max()) < 2e-5          # the frame needs the border GEMMs
    d.Cout = 48                                                                                # unsupported channel count
    assert L.ramnet_conv_launch(C.byref(d), st) == 10001

    # backward-weights: dU -> dW4 = G^T dU G must equal the gradient of the four 4x4 parity filters
    dy = torch.randn(B, 2 * H, 2 * W, Cout, device=dev)
    ws, dbias = torch.zeros(4 * 25 * Cin * Cout, device=dev), torch.zeros(Cout, device=dev)
    g = _hip.WgradDesc()
    g.x0, g.ld0, g.C0, g.in_mode = ptr(xpad), Cin, Cin, _hip.IN_PLAIN
    g.B, g.Hin, g.Win, g.ntaps, g.stride = B, H + 4, W + 4, 16, 1
    g.dout, g.ldg, g.Cout, g.Ho, g.Wo, g.HoG, g.WoG = ptr(dy), Cout, Cout, H, W, 2 * H, 2 * W
    g.dw, g.dbias, g.algo = ptr(ws), ptr(dbias), _hip.ALGO_WINOGRAD24
    assert L.ramnet_wgrad_launch(C.byref(g), st) == 0, L.ramnet_last_error()
    G = torch.tensor(tr.W24_G, dtype=torch.float64)
    d4 = torch.einsum("at,bs,pqabio->pqtsio", G, G, ws.view(2, 2, 5, 5, Cin, Cout).cpu().double())
    xp = xpad.permute(0, 3, 1, 2).cpu().double()
    dyr = dy.permute(0, 3, 1, 2).cpu().double()
    for py in range(2):
        for px in range(2):
            w4 = torch.zeros(Cout, Cin, 4, 4, dtype=torch.float64, requires_grad=True)
            out = F.conv2d(xp[:, :, py:py + H + 3, px:px + W + 3], w4)                         # [B][Cout][H][W]
            (out * dyr[:, :, py::2, px::2]).sum().backward()
            got4 = d4[py, px].permute(3, 2, 0, 1)                                              # [co][ci][t][s]
            assert float((got4 - w4.grad).abs().max() / w4.grad.abs().max()) < 2e-5
    assert float((dbias.cpu().double() - dyr.sum((0, 2, 3))).abs().max() / dyr.sum((0, 2, 3)).abs().max()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(5504, 256, 1280), (22, 40, 100), (96, 64, 320), (33, 8, 20)])
def test_gemm_entry_point_raw(M, N, K):
    """ramnet_gemm (border corrections of the folded upsample-conv): C = A B, C += A^T B with ragged sizes, against float64."""
    L = _hip.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(M + N)
    a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    c = torch.full((M, N), float("nan"), device=dev)
    assert L.ramnet_gemm(ptr(a), ptr(b), ptr(c), M, N, K, K, N, N, 0, 0, 1, 0, 0, 0, st) == 0          # K >= 128: four waves per block, fixed-order join; below: one wave
    ref = a.double().cpu() @ b.double().cpu()
    assert float((c.cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-5
    c1 = torch.full((M, N), float("nan"), device=dev)
    assert L.ramnet_gemm(ptr(a), ptr(b), ptr(c1), M, N, K, K, N, N, 0, 0, 1, 0, 0, 0, st) == 0
    assert torch.equal(c, c1), "the plain product must be bit-reproducible"
    # weight-gradient form: W[K][N] += A[R = M][K]^T G[R][N], twice (accumulation over BPTT steps); reduction split + atomics
    g = torch.randn(M, N, device=dev)
    w = torch.zeros(K, N, device=dev)
    for _ in range(2):
        assert L.ramnet_gemm(ptr(a), ptr(g), ptr(w), K, N, M, K, N, N, 1, 1, 1, 0, 0, 0, st) == 0
    refw = 2 * (a.double().cpu().t() @ g.double().cpu())
    assert float((w.cpu().double() - refw).abs().max() / refw.abs().max()) < 1e-5
    # a batch of two products in one launch (the two sides of a border)
    a2, b2 = torch.randn(2, M, K, device=dev), torch.randn(2, K, N, device=dev)
    c2 = torch.full((2, M, N), float("nan"), device=dev)
    assert L.ramnet_gemm(ptr(a2), ptr(b2), ptr(c2), M, N, K, K, N, N, 0, 0, 2, M * K, K * N, M * N, st) == 0
    ref2 = torch.bmm(a2.double().cpu(), b2.double().cpu())
    assert float((c2.cpu().double() - ref2).abs().max() / ref2.abs().max()) < 1e-5


@pytest.mark.parametrize("groups_mode", ["batch", "instance"])
@pytest.mark.parametrize("C_", [32, 6])
def test_norm_entry_points_raw(groups_mode, C_):
    """ramnet_norm_partial / _apply / _bwd with raw pointers (16-byte channel quads and the scalar path): batch / instance
    normalisation + ReLU forward and backward against float64 autograd of the textbook formula."""
    L = _hip.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(3)
    B, H, W = 3, 9, 14
    x = torch.randn(B, H, W, C_, device=dev) * 2 + 0.5
    gamma, beta = torch.rand(C_, device=dev) + 0.5, torch.randn(C_, device=dev) * 0.2
    dy = torch.randn(B, H, W, C_, device=dev)
    G = 1 if groups_mode == "batch" else B
    npix = B * H * W // G
    nslab = L.ramnet_norm_slabs(G, npix, C_)
    assert nslab >= 1
    part = torch.empty(G, nslab, C_, 2, dtype=torch.float64, device=dev)
    assert L.ramnet_norm_partial(ptr(x), C_, None, 0, 0, ptr(x), C_, G, npix, C_, nslab, ptr(part), st) == 0, L.ramnet_last_error()
    s = part.sum(1)
    mean = s[..., 0] / npix
    var = s[..., 1] / npix - mean * mean
    xd = x.double().cpu().requires_grad_(True)
    xr = xd.reshape(G, npix, C_)
    mu, vv = xr.mean(1, keepdim=True), xr.var(1, unbiased=False, keepdim=True)
    np.testing.assert_allclose(mean.cpu().numpy(), mu[:, 0].detach().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(var.cpu().numpy(), vv[:, 0].detach().numpy(), rtol=1e-5)
    eps = 1e-5
    rstd = (var + eps).rsqrt()
    scale = (rstd * gamma.double()).float().contiguous()
    shift = (beta.double() - mean * rstd * gamma.double()).float().contiguous()
    # the same vectors from the library's own finalize kernel, with torch's running-buffer update
    rm, rv = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    mr = torch.empty(2, G, C_, dtype=torch.float64, device=dev)
    ss = torch.empty(2, G, C_, device=dev)
    assert L.ramnet_norm_finalize(ptr(part), G, nslab, C_, npix, eps, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1, 0, ptr(nbt),
                                  ptr(mr[0]), ptr(mr[1]), ptr(ss[0]), ptr(ss[1]), st) == 0, L.ramnet_last_error()
    np.testing.assert_allclose(mr[0].cpu().numpy(), mean.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(mr[1].cpu().numpy(), rstd.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(ss[0].cpu().numpy(), scale.reshape(G, C_).cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(ss[1].cpu().numpy(), shift.reshape(G, C_).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * mean.mean(0).cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * (var * npix / (npix - 1)).mean(0)).cpu().numpy(), rtol=1e-6)
    assert int(nbt) == 1
    y = torch.empty_like(x)
    assert L.ramnet_norm_apply(ptr(x), C_, ptr(scale), ptr(shift), None, 0, 1, ptr(y), C_, G, npix, C_, st) == 0, L.ramnet_last_error()
    ref = torch.relu((xr - mu) / (vv + eps).sqrt() * gamma.double().cpu() + beta.double().cpu()).reshape(B, H, W, C_)
    assert float((y.cpu().double() - ref.detach()).abs().max()) < 2e-5
    (ref * dy.double().cpu()).sum().backward()
    assert L.ramnet_norm_partial(ptr(dy), C_, ptr(y), C_, 1, ptr(x), C_, G, npix, C_, nslab, ptr(part), st) == 0
    s = part.sum(1)
    s1 = s[..., 0]
    s2 = rstd * (s[..., 1] - mean * s1)
    g64 = gamma.double()
    c1 = (g64 * rstd).float().contiguous()
    c2d = -g64 * rstd * rstd * s2 / npix
    c3 = (-g64 * rstd * s1 / npix - c2d * mean).float().contiguous()
    c2 = c2d.float().contiguous()
    cc = torch.empty(3, G, C_, device=dev)
    dgb = torch.empty(2, C_, device=dev)
    assert L.ramnet_norm_finalize_bwd(ptr(part), G, nslab, C_, npix, ptr(mr[0]), ptr(mr[1]), ptr(gamma), 1, ptr(cc[0]), ptr(cc[1]), ptr(cc[2]),
                                      ptr(dgb[0]), ptr(dgb[1]), st) == 0, L.ramnet_last_error()
    for got, want in ((cc[0], c1), (cc[1], c2), (cc[2], c3)):
        np.testing.assert_allclose(got.cpu().numpy(), want.reshape(G, C_).cpu().numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(dgb[0].cpu().numpy(), s2.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dgb[1].cpu().numpy(), s1.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-6)
    dx = torch.empty_like(x)
    assert L.ramnet_norm_bwd(ptr(dy), C_, ptr(y), C_, 1, ptr(x), C_, ptr(c1), ptr(c2), ptr(c3), ptr(dx), C_, None, 0, G, npix, C_, st) == 0, \
        L.ramnet_last_error()
    # pixels on the ReLU kink of either side are excluded (the fp32 output is 0 where the fp64 one is 1e-8)
    ok = ((y.cpu() > 0) == (ref.detach() > 0)).all(dim=-1, keepdim=True).expand_as(dx)
    err = ((dx.cpu().double() - xd.grad)[ok]).abs().max() / xd.grad.abs().max()
    assert float(err) < 1e-4, float(err)
    # refused: a row stride below the channel count
    assert L.ramnet_norm_apply(ptr(x), C_ - 1, ptr(scale), ptr(shift), None, 0, 1, ptr(y), C_, G, npix, C_, st) == 10001
