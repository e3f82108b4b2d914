"""Isolated backward-weights launches of one ConvGRU update at the bench shape (B = 8, 256 x 344): exact-fp32 F(2x4,3x3)
(csrc/conv_wgrad_wino6.hip) against the direct split-operand kernel (csrc/conv_wgrad_dsplit.hip), per-split slabs, ms per launch.
    python tools/bench_wgrad_dsplit.py [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as Hh   # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
L = Hh.lib()
taps = ops.Taps.get("conv", 3, 1)
B = int(os.environ.get("B", "8"))
shapes = [("gru0 gates", 128, 172, 128, 128), ("gru0 cand", 128, 172, 128, 64), ("gru1 gates", 64, 86, 256, 256), ("gru1 cand", 64, 86, 256, 128),
          ("gru2 gates", 32, 43, 512, 512), ("gru2 cand", 32, 43, 512, 256)]
tot = {"f2x4": 0.0, "wg_dsplit": 0.0}
for name, H, W, cin, cout in shapes:
    x = torch.randn(B, H, W, cin, device=dev)
    dy = torch.randn(B, H, W, cout, device=dev)
    row = []
    for kind in ("f2x4", "wg_dsplit"):
        if kind == "f2x4":
            slabs, n = L.ramnet_wgrad_wino2x4_slabs(cin, cout), L.ramnet_wgrad_wino2x4_ws_floats(cin, cout)
        else:
            slabs, n = L.ramnet_wgrad_dsplit_slabs(cin, cout), L.ramnet_wgrad_dsplit_ws_floats(cin, cout)
        ws = torch.zeros(slabs * n, device=dev)
        ws.wino, ws.wino6, ws.wg_dsplit, ws.slabs = False, kind == "f2x4", kind == "wg_dsplit", slabs
        bws = torch.zeros(slabs * cout, device=dev)
        for _ in range(3):
            ops.wgrad_launch(x, taps, dy, ws, cout, dbias=bws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.wgrad_launch(x, taps, dy, ws, cout, dbias=bws)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot[kind] += ms
        row.append("%s %.3f ms (%s, %d slabs)" % (kind, ms, L.ramnet_last_kernel().decode(), slabs))
    print("%-11s Cin %3d Cout %3d %3dx%3d: %s" % (name, cin, cout, H, W, " | ".join(row)), flush=True)
print("six launches: f2x4 %.3f ms, dsplit %.3f ms (x%.2f)" % (tot["f2x4"], tot["wg_dsplit"], tot["f2x4"] / tot["wg_dsplit"]))
