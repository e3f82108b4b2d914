#!/usr/bin/env python3
"""Isolated launches of the prediction layer + scale-invariant statistics (csrc/pointwise.hip: pred_sigmoid_si_fwd / _bwd) at the bench shape
(two supervised segments of 8 x 256 x 344 pixels, 32 channels): microseconds and the fraction of the 8 TB/s HBM peak for the algorithmic
bytes (4 (C + 1) per pixel forward, 4 (2 C + 3) backward), over the workgroup cap of the forward launch (ramnet_set_option "pred_si_cap").
Usage (GPU box): python tools/bench_pred_si.py [caps ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as H  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3        # us


def main():
    dev = torch.device("cuda:0")
    caps = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048, 4096]
    B, Hh, W, Cc, n = 16, 256, 344, 32, 2
    torch.manual_seed(0)
    x = torch.randn(B, Hh, W, Cc, device=dev, requires_grad=True)
    w = torch.randn(1, Cc, 1, 1, device=dev, requires_grad=True) * 0.1
    w = w.detach().requires_grad_(True)
    b = torch.zeros(1, device=dev, requires_grad=True)
    tg = [torch.rand(B // n, 1, Hh, W, device=dev) for _ in range(n)]
    # something else through the caches between two launches, as in the step (the input was written by the last decoder, not re-read hot)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    fwd_bytes, bwd_bytes = B * Hh * W * 4 * (Cc + 2), B * Hh * W * 4 * (2 * Cc + 3)
    ref = None
    for cap in caps:
        H.check(H.lib().ramnet_set_option(b"pred_si_cap", cap), "set_option")

        def f():
            return ops.PredSigmoidSI.apply(x, w, b, 1.0, 0.5, False, *tg)

        def f_cold():
            flush.zero_()
            return f()
        # the raw library call (no autograd, no allocation): what the launch itself takes
        L = H.lib()
        import ctypes as C
        xd, wd, bd = x.detach(), w.detach(), b.detach()
        y = torch.empty(B, 1, Hh, W, device=dev)
        seg_pix = (B // n) * Hh * W
        scratch = torch.zeros(L.ramnet_pred_si_scratch_doubles(seg_pix, n), device=dev, dtype=torch.float64)
        stats = torch.empty(n, 4, device=dev, dtype=torch.float64)
        lossb = torch.empty(n, device=dev)
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in tg])
        st = torch.cuda.current_stream().cuda_stream

        def raw():
            H.check(L.ramnet_pred_sigmoid_si_fwd(xd.data_ptr(), Cc, Cc, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), seg_pix, n, arr, 1.0, 0.5,
                                                 scratch.data_ptr(), stats.data_ptr(), lossb.data_ptr(), st), "raw")
        t_raw = timeit(raw, 200)
        print("cap %6d: raw launch %7.1f us (%.3f)" % (cap, t_raw, fwd_bytes / t_raw / 8e6))
        out = f()
        got = torch.stack([o.detach() for o in out[1:]]).cpu()
        if ref is None:
            ref = got
        t_hot = timeit(lambda: f())
        t_cold = timeit(f_cold) - timeit(lambda: flush.zero_())
        print("cap %6d: forward hot %7.1f us (%.3f of 8 TB/s)   behind a 512 MB fill %7.1f us (%.3f)   losses %s (max diff to the first cap %.2e)" % (
            cap, t_hot, fwd_bytes / t_hot / 8e6, t_cold, fwd_bytes / t_cold / 8e6, got.tolist(), (got - ref).abs().max().item()))
    H.check(H.lib().ramnet_set_option(b"pred_si_cap", 256), "set_option")
    raw()
    dy = torch.randn(B, 1, Hh, W, device=dev)
    dx = torch.empty(B, Hh, W, Cc, device=dev)
    dw, db, gs = torch.zeros(Cc, device=dev), torch.zeros(1, device=dev), torch.ones(n, device=dev)
    for cap in [64, 128, 256, 512, 1024, 2048, 4096]:
        H.check(L.ramnet_set_option(b"pred_si_bwd_cap", cap), "set_option")
        scratch_b = torch.zeros(L.ramnet_pred_si_scratch_doubles(seg_pix, n), device=dev, dtype=torch.float64)
        for joined in (False, True):
            with_dy = True

            def rawb():
                H.check(L.ramnet_pred_sigmoid_si_bwd(xd.data_ptr(), Cc, Cc, wd.data_ptr(), y.data_ptr(), dy.data_ptr() if with_dy else None, seg_pix, n, arr,
                                                     stats.data_ptr(), gs.data_ptr(), 1.0, 0.5, dx.data_ptr(), Cc, dw.data_ptr(), db.data_ptr(),
                                                     scratch_b.data_ptr() if joined else None, 0, st), "rawb")
            t = timeit(rawb, 200)
            nb = B * Hh * W * 4 * (2 * Cc + 2 + (1 if with_dy else 0))
            print("bwd cap %6d %s: raw launch %7.1f us (%.3f of 8 TB/s)" % (cap, "joined through scratch" if joined else "fp32 atomics (<= 512 workgroups)", t, nb / t / 8e6))
    H.check(L.ramnet_set_option(b"pred_si_bwd_cap", 1024), "set_option")
    out = f()
    loss = out[1] + out[2]
    t_bwd = timeit(lambda: torch.autograd.grad(loss, x, retain_graph=True))
    print("backward (grad of both losses w.r.t. x; includes autograd's own work): %7.1f us (%.3f of 8 TB/s)" % (t_bwd, bwd_bytes / t_bwd / 8e6))


if __name__ == "__main__":
    main()
