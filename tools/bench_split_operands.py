#!/usr/bin/env python3
"""The six forward launches of one ConvGRU state update (gates and candidate at the three scales, submodules.py:447-452) and their four
backward-data launches per scale, isolated, at the bench batch: exact-fp32 F(2x4,3x3) (csrc/conv_wino6.hip) against the split-operand
form (csrc/conv_wino6s.hip), with the error of both against float64 on a sample.
Usage (GPU box): python tools/bench_split_operands.py [--batch 8] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as H  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=344)
    ap.add_argument("--scales", default="0,1,2")
    ap.add_argument("--quick", action="store_true", help="split-operand forward launches only, no reference (ablation builds: tools/abl_wino6s.sh)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    taps, tapsd = ops.Taps.get("conv", 3, 1), ops.Taps.get("dgrad1", 3, 1)
    ops.set_winograd_2x4("force")
    tot = {False: 0.0, True: 0.0}
    for i in [int(s) for s in a.scales.split(",")]:
        C = 64 << i
        Hh, Ww = a.height >> (i + 1), a.width >> (i + 1)
        torch.manual_seed(i)
        wur = torch.nn.Parameter(torch.randn(2 * C, 2 * C, 3, 3, device=dev) * (0.5 / (18 * C) ** 0.5))
        bur = torch.nn.Parameter(torch.randn(2 * C, device=dev) * 0.1)
        wo = torch.nn.Parameter(torch.randn(C, 2 * C, 3, 3, device=dev) * (0.5 / (18 * C) ** 0.5))
        bo = torch.nn.Parameter(torch.randn(C, device=dev) * 0.1)
        cp_ur, cp_o = ops.ConvParam([wur], [bur]), ops.ConvParam([wo], [bo])
        x = torch.randn(B, Hh, Ww, C, device=dev)
        h = torch.tanh(torch.randn(B, Hh, Ww, C, device=dev))
        ur, hr, hn, o = (torch.empty(B, Hh, Ww, n, device=dev) for n in (2 * C, C, C, C))
        dpo, dpur = torch.randn(B, Hh, Ww, C, device=dev), torch.randn(B, Hh, Ww, 2 * C, device=dev)
        dxh = torch.zeros(B, Hh, Ww, 2 * C, device=dev)
        launches = {
            "gates": lambda: ops.conv_launch(x, taps, cp_ur.fwd(), ur, 2 * C, x1=h, in_mode=H.IN_CAT, C1=C, bias=cp_ur.bias(), epi=H.EPI_SIGMOID_HR, e1=h, o1=hr),
            "cand": lambda: ops.conv_launch(x, taps, cp_o.fwd(), hn, C, x1=hr, in_mode=H.IN_CAT, C1=C, bias=cp_o.bias(), epi=H.EPI_GRU_BLEND, e0=ur, e1=h, o1=o),
            "cand_bwd": lambda: ops.conv_launch(dpo, tapsd, cp_o.bwd(), dxh, 2 * C, epi=H.EPI_GRU_BWD, e0=ur, e1=h, o1=dpur) if C % 64 == 0 else None,
            "gates_bwd": lambda: ops.conv_launch(dpur, tapsd, cp_ur.bwd(), dxh, 2 * C, beta=0.0),
        }
        res, outs = {}, {}
        if a.quick:
            ops.set_split_operands(True)
            tg, tc = timeit(launches["gates"], a.reps), timeit(launches["cand"], a.reps)
            print("scale %d  gates %.4f ms  cand %.4f ms   [%s]" % (i, tg, tc, H.lib().ramnet_last_kernel().decode()))
            tot[True] += tg + tc
            continue
        for split in (False, True):
            ops.set_split_operands(split)
            for name, fn in launches.items():
                res[(name, split)] = timeit(fn, a.reps)
                res[(name, split, "k")] = H.lib().ramnet_last_kernel().decode()
            launches["gates"](), launches["cand"]()
            torch.cuda.synchronize()
            outs[split] = (ur.clone(), hn.clone())
        ops.set_split_operands(False)
        # float64 reference of the gates on the first image (CPU, a few seconds)
        import torch.nn.functional as F
        xin = torch.cat([x[:1], h[:1]], 3).permute(0, 3, 1, 2).double().cpu()
        ref = torch.sigmoid(F.conv2d(xin, wur.detach().double().cpu(), bur.detach().double().cpu(), 1, 1)).permute(0, 2, 3, 1)
        e32 = float((outs[False][0][:1].double().cpu() - ref).abs().max())
        esp = float((outs[True][0][:1].double().cpu() - ref).abs().max())
        dsp = float((outs[True][1] - outs[False][1]).abs().max())
        gf_g, gf_c = 2.0 * B * Hh * Ww * 9 * 2 * C * 2 * C / 1e9, 2.0 * B * Hh * Ww * 9 * 2 * C * C / 1e9
        print("scale %d  C=%3d %3dx%3d  [%s | %s]" % (i, C, Hh, Ww, res[("gates", False, "k")], res[("gates", True, "k")]))
        for name, gf in (("gates", gf_g), ("cand", gf_c), ("cand_bwd", gf_g), ("gates_bwd", gf_g)):
            t0, t1 = res[(name, False)], res[(name, True)]
            print("   %-10s exact fp32 %.4f ms (%6.1f TF/s alg.)   split operands %.4f ms (%6.1f)   x%.2f" % (name, t0, gf / t0, t1, gf / t1, t0 / t1))
        print("   gates max |err| vs float64 (sigmoid outputs): exact fp32 %.2e, split %.2e;  new state split vs exact: %.2e" % (e32, esp, dsp))
        for split in (False, True):
            tot[split] += res[("gates", split)] + res[("cand", split)]
    if a.quick:
        print("six forward launches, split operands: %.4f ms" % tot[True])
        return
    print("six forward launches of one update: exact fp32 %.4f ms, split operands %.4f ms: x%.2f" % (tot[False], tot[True], tot[False] / tot[True]))


if __name__ == "__main__":
    main()
