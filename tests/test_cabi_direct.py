"""GPU: the C ABI driven DIRECTLY with ctypes and raw device pointers (no rpg_ramnet_amd.ops in between), the way a
maintainer of another host language would bind it."""
import ctypes as C

import numpy as np
import pytest
import torch

from rpg_ramnet_amd import _hip

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
    d.epi, d.beta, d.out, d.ldo, d.precision = _hip.EPI_RELU, 0.0, ptr(y), Cout, _hip.PREC_F32
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
    stats = torch.empty(3, dtype=torch.float64, device=dev)
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
