#!/usr/bin/env python3
"""Headline benchmark: RAM-Net training step on MI355X (BASELINE.json configs[1]/[2]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one optimizer step over a batch of B=8 sequences of L=8 data packages per GPU (each package = 5 event
voxel grids of 5 bins + 1 gray frame at 256x344, the reference's centre crop of 346x260 — the raw size does not run
through the reference network): forward through ERGB2DepthRecurrent, scale-invariant loss on
['image','events4'], BPTT backward, gradient all-reduce over RCCL (N>1), Adam.  Metric: depth samples/s, one sample
= one data package of one batch element; value = N*B*L*K / max-over-ranks(time).
Inputs are synthetic (device-generated event lists -> HIP voxel scatter-add -> nonzero normalisation; U[0,1) frames;
log-depth targets) and resident in HBM before the timed region.  Weights: seeded random init (no checkpoints offline).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RELEASED = dict(num_bins_rgb=1, num_bins_events=5, skip_type="sum", recurrent_block_type="conv",
                state_combination="convgru", spatial_resolution=[112, 112], num_encoders=3, base_num_channels=32,
                num_residual_blocks=2, use_upsample_conv=True, norm="none")
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq-len", type=int, default=8)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=344)
    ap.add_argument("--events-per-grid", type=int, default=200000)
    ap.add_argument("--bins", type=int, default=5, help="event voxel-grid bins (configs[4] uses 10)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for 1-GPU functional tests)")
    ap.add_argument("--mode", choices=["train", "infer", "stream"], default="train",
                    help="train: configs[1]/[2] step; infer: forward over the same sequences; stream: configs[3]-style batch-1 "
                         "asynchronous inference, irregular number of event grids per frame, persistent state")
    ap.add_argument("--state", choices=["convgru", "convlstm"], default="convgru")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-overlap-wgrad", dest="overlap_wgrad", action="store_false",
                    help="run backward-weights on the main stream instead of co-scheduling it with backward-data on a side stream "
                         "(the default schedule; per-kernel durations then include co-scheduled time)")
    ap.add_argument("--no-overlap-decoder", dest="overlap_decoder", action="store_false",
                    help="run the decoders on the main stream instead of a second stream concurrent with the next state update "
                         "(ops.set_decoder_overlap; default schedule)")
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default="f32")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra (untimed-for-value) overlap / bf16x3 measurements")
    return ap.parse_args()


def synth_sequence(model, B, L, H, W, K, bins, n_events, seed):
    """SURVEY section 8d recipe, generated on device; returns a list of L item dicts (NCHW device tensors)."""
    from rpg_ramnet_amd import voxel
    dev = model.gpu
    g = torch.Generator(device=dev).manual_seed(seed)
    seq = []
    for _ in range(L):
        item = {}
        for k in range(K):
            grids = []
            for _b in range(B):
                ev = torch.empty(n_events, 4, device=dev, dtype=torch.float64)
                ev[:, 0] = torch.sort(torch.rand(n_events, device=dev, generator=g, dtype=torch.float64) * 0.05)[0]
                ev[:, 1] = torch.randint(0, W, (n_events,), device=dev, generator=g).double()
                ev[:, 2] = torch.randint(0, H, (n_events,), device=dev, generator=g).double()
                ev[:, 3] = torch.randint(0, 2, (n_events,), device=dev, generator=g).double()
                grids.append(voxel.normalize_nonzero(voxel.events_to_voxel_grid(ev, bins, W, H)))
            item["events%d" % k] = torch.stack(grids)
        item["image"] = torch.rand(B, 1, H, W, device=dev, generator=g)
        for key in ("image", "events%d" % (K - 1)):
            u = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.98 + 0.02
            item["depth_" + key] = torch.clamp(1.0 + torch.log(u) / 3.70378, 0.0, 1.0)
        seq.append(item)
    return seq


class KernelTimer:
    """HIP-event timing of every MFMA launch on the launch stream, keyed by kernel symbol."""

    def __init__(self):
        self.rec = []
        self.on = False
        self.only = None        # when set: bracket only launches of this kernel symbol (keeps the timed region unperturbed)

    def install(self):
        from rpg_ramnet_amd import ops
        timer = self
        conv0, wgrad0 = ops.conv_launch, ops.wgrad_launch

        def variant(Cout, epi, B, Ho, Wo):
            """Mirror of the tile-configuration choice in ramnet_conv_launch (csrc/conv_igemm.hip)."""
            from rpg_ramnet_amd import _hip as Hh
            if epi == Hh.EPI_LSTM:
                return "conv_igemm_kernel<128,128,4,1,1>"
            cp = (Cout + 31) // 32 * 32
            bm, bn = 128, (128 if cp % 128 == 0 else 64 if cp % 64 == 0 else 32)

            def blocks(m, n):
                return -(-Wo // 16) * -(-Ho // (m // 16)) * B * (cp // n)
            if bn >= 64:
                if blocks(bm, bn) < 768:
                    bm = 64
                if blocks(bm, bn) < 768 and bn == 128:
                    bn = 64
            if bn == 32 and -(-Wo // 16) * -(-Ho // 16) * B >= 1024:
                bm = 256
            return "conv_igemm_kernel<%d,%d,%s,1>" % (bm, bn, "4,1" if bn == 32 else "2,2")

        def conv(x0, taps, w, out, Cout, **kw):
            if not timer.on:
                return conv0(x0, taps, w, out, Cout, **kw)
            Ho, Wo = kw.get("Ho") or out.shape[1], kw.get("Wo") or out.shape[2]
            name = variant(Cout, kw.get("epi", 0), x0.shape[0], Ho, Wo)
            if ops.uses_winograd(taps, w, kw.get("stride", 1), kw.get("epi", 0), kw.get("in_mode", 0),
                                 kw.get("C0") or x0.shape[3], kw.get("C1", 0)):
                name = "conv_wino_kernel"
            if ops.uses_head(taps, w, kw.get("stride", 1), kw.get("epi", 0), kw.get("in_mode", 0)):
                name = "conv_head_fwd_kernel"
            nclass = 1
            if kw.get("wino24"):        # one launch = the four output parities of a folded decoder layer
                name, nclass = "conv_wino24_kernel", 4
            if timer.only is not None and name != timer.only:
                return conv0(x0, taps, w, out, Cout, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            conv0(x0, taps, w, out, Cout, **kw)
            e.record()
            cin = ((kw.get("C0") or x0.shape[3]) + kw.get("C1", 0)) * (4 if kw.get("in_mode", 0) == 6 else 1)   # IN_S2D: 4 parities
            nout = Cout * (4 if kw.get("epi") == 5 else 1)
            timer.rec.append((name, s, e, 2.0 * nclass * x0.shape[0] * Ho * Wo * taps.flop_taps * cin * nout))

        def wgrad(x0, taps, dout, dw, Cout, **kw):
            if not timer.on:
                return wgrad0(x0, taps, dout, dw, Cout, **kw)
            cin = ((kw.get("C0") or x0.shape[3]) + kw.get("C1", 0)) * (4 if kw.get("in_mode", 0) == 6 else 1)
            tpm = 1 if cin > 16 else 8 if cin <= 4 else 4 if cin <= 8 else 2      # mirror of ramnet_wgrad_launch
            ntt = -(-taps.n // tpm)
            if Cout <= 32 or taps.n > 9:
                per = -(-ntt // 4)
                name = "conv_wgrad_kernel<%d,1,1>" % (1 if per <= 1 else 3 if per <= 3 else 4 if per <= 4 else 7)
            elif (taps.n > 1 and cin >= 64 and
                  -(-dout.shape[2] // 16) * -(-dout.shape[1] // 8) * dout.shape[0] * -(-cin // 64) * -(-Cout // 64) >= 2048):
                name = "conv_wgrad_kernel<9,2,2>"
            else:
                per = -(-ntt // 2)
                name = "conv_wgrad_kernel<%d,2,1>" % (1 if per <= 1 else 5)
            if getattr(dw, "wino", False):         # Winograd backward-weights (template flags: loader operand, ReLU mask on dy)
                from rpg_ramnet_amd import _hip as Hh
                name = "conv_wgrad_wino_kernel<%d,%d>" % (kw.get("in_mode", 0) in (Hh.IN_RELUMASK, Hh.IN_CAT_MUL), kw.get("gmask") is not None)
            nclass = 1
            if kw.get("wino24"):
                name, nclass = "conv_wgrad_wino24_kernel", 4
            if getattr(dw, "head_cin", 0) and ops.get_head_kernel() and taps.head and kw.get("stride", 1) == 1 and kw.get("in_mode", 0) == 0 \
                    and kw.get("gview") is None:
                name = "conv_head_wgrad_kernel"
            if timer.only is not None and name != timer.only:
                return wgrad0(x0, taps, dout, dw, Cout, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            wgrad0(x0, taps, dout, dw, Cout, **kw)
            e.record()
            ho, wo = kw.get("Ho") or dout.shape[1], kw.get("Wo") or dout.shape[2]       # (parity sub-grid of the folded upsample-conv)
            timer.rec.append((name, s, e, 2.0 * nclass * dout.shape[0] * ho * wo * taps.flop_taps * cin * Cout))

        multi0 = ops.conv_launch_multi

        def multi(x0, w, out, Cout, classes, **kw):
            if not timer.on:
                return multi0(x0, w, out, Cout, classes, **kw)
            tiles = sum(-(-Wo // 16) * -(-Ho // 8) for _, Ho, Wo, _ in classes) * x0.shape[0]
            cp = (Cout + 31) // 32 * 32
            bm, bn = 128, (128 if cp % 128 == 0 else 64 if cp % 64 == 0 else 32)
            if bn >= 64:
                if tiles * (cp // bn) < 768:
                    bm = 64
                    tiles = sum(-(-Wo // 16) * -(-Ho // 4) for _, Ho, Wo, _ in classes) * x0.shape[0]
                if tiles * (cp // bn) < 768 and bn == 128:
                    bn = 64
            if bn == 32 and sum(-(-Wo // 16) * -(-Ho // 16) for _, Ho, Wo, _ in classes) * x0.shape[0] >= 1024:
                bm = 256
            name = "conv_igemm_kernel<%d,%d,%s,1>" % (bm, bn, "4,1" if bn == 32 else "2,2")
            if timer.only is not None and name != timer.only:
                return multi0(x0, w, out, Cout, classes, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            multi0(x0, w, out, Cout, classes, **kw)
            e.record()
            cin = (kw.get("C0") or x0.shape[3]) + kw.get("C1", 0)
            fl = sum(2.0 * x0.shape[0] * Ho * Wo * taps.flop_taps * cin * Cout for taps, Ho, Wo, _ in classes)
            timer.rec.append((name, s, e, fl))

        ops.conv_launch, ops.wgrad_launch, ops.conv_launch_multi = conv, wgrad, multi

    def summary(self):
        agg = {}
        for name, s, e, fl in self.rec:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e) * 1e-3
            a[2] += fl
        return agg


def cpu_baseline(cfg, H, W, K, args):
    """The oracle (CPU restatement, fixture-pinned to the reference) timed on this box's host cores on a bounded
    sample of the same workload: one training step (fwd + SI loss + BPTT bwd) of B=1, L=1 at the bench resolution."""
    from oracle import ramnet_ref
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipe import make_item
    from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):      # the constructor prints (like the reference's); stdout carries ONE JSON line
        m = ERGB2DepthRecurrent(cfg)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    rng = np.random.default_rng(0)
    cores = torch.get_num_threads()
    B, L = 2, 2                                   # ~10-30 s of CPU work on the GPU box's host cores
    seq = [make_item(rng, B, H, W, K, cfg["num_bins_events"], 1, True, 0.0) for _ in range(L)]
    lc = cfg["loss_composition"]
    t0 = time.time()
    if args.mode == "train":
        total, _ = ramnet_ref.sequence_loss(sd, cfg, seq, lc, [1, 1])
        total.backward()
    else:
        with torch.no_grad():
            ramnet_ref.forward_recurrent(sd, cfg, seq[0], None, ramnet_ref.empty_states_lstm(K))
    dt = time.time() - t0
    return {"value": B * L / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%s step, B=%d L=%d K=%d %dx%d fp32 torch-CPU oracle, %.1f s" % (args.mode, B, L, K, H, W, dt)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("RAMNET_BENCH_SINGLE_DEVICE") == "1":     # functional test of the N>1 path on a 1-GPU box
            local = 0
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent
    from rpg_ramnet_amd.parallel import FlatGradReducer
    from rpg_ramnet_amd.trainer import sequence_loss, empty_states_lstm

    from rpg_ramnet_amd import ops
    ops.set_precision(args.precision)
    ops.set_wgrad_overlap(args.overlap_wgrad)
    ops.set_decoder_overlap(args.overlap_decoder)
    K, bins, B, L, H, W = 5, args.bins, args.batch, args.seq_len, args.height, args.width
    cfg = dict(RELEASED, num_bins_events=bins, gpu=local, every_x_rgb_frame=K, baseline=False, loss_composition=["image", "events4"],
               state_combination=args.state)
    torch.manual_seed(0)                       # identical initial weights on every rank (train.py:203)
    with contextlib.redirect_stdout(sys.stderr):
        model = ERGB2DepthRecurrent(cfg)
    model = model.to(model.gpu)
    seq = synth_sequence(model, B, L, H, W, K, bins, args.events_per_grid, seed=1000 + rank)
    timer = KernelTimer()
    if not args.no_kernel_timing:
        timer.install()

    if args.mode == "train":
        model.train()
        reducer = FlatGradReducer(model)
        opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=0)

        def step():
            reducer.zero()
            total, _ = sequence_loss(model, seq, cfg["loss_composition"], [1, 1])
            total.backward()
            reducer.all_reduce()
            reducer.wait()
            opt.step()
            return total
    elif args.mode == "stream":
        model.eval()
        rs = np.random.default_rng(7)
        sched = [int(v) for v in rs.integers(1, 9, size=L)]        # event grids before each frame, drawn from {1..8}
        stream_state = {"s": model.init_states(B, H, W)}

        def step():
            st = stream_state["s"]                                 # state persists ACROSS steps (one long stream)
            with torch.no_grad():
                for l, item in enumerate(seq):
                    for k in range(sched[l]):
                        st, _ = model.update_events(item["events%d" % (k % K)], st)
                        pred = model.decode(st)
                    st, _ = model.update_image(item["image"], st)
                    pred = model.decode(st)
            stream_state["s"] = st
            return pred.mean()
    else:
        model.eval()

        def step():
            prev_super, prev_lstm = None, empty_states_lstm(K)
            with torch.no_grad():
                for item in seq:
                    preds, supers, prev_lstm = model(item, prev_super, prev_lstm)
                    prev_super = supers["image"]
            return preds["image"].mean()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up steps bracket EVERY MFMA launch with HIP events (per-kernel table + choice of the dominant symbol); in the
    # timed region only the dominant symbol's launches are bracketed: ~7000 event pairs per step cost 3-4 % of the step.
    last = None
    timer.on = not args.no_kernel_timing
    for _ in range(args.warmup):
        last = step()
    fence()
    warm = timer.summary()
    timer.rec = []
    if warm:
        timer.only = max(warm.items(), key=lambda kv: kv[1][1])[0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    timer.on = False
    agg_timed = timer.summary()          # dominant kernel over the timed region (the extras below reuse the timer)
    timer.rec = []
    if world > 1:
        t = torch.tensor([dt], device=model.gpu, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss_val = float(last.detach())
    assert np.isfinite(loss_val), "non-finite loss"

    # ---- extras (outside the timed region that defines `value`): the same step with (a) backward-weights overlapped on a
    # side stream, (b) the bf16x3 split-operand contraction for forward/backward-data.  2 steps each, 1 warm-up.
    extras = {}
    if not args.no_extras and args.mode == "train" and world == 1:
        def measure(n=2):
            step()
            fence()
            t = time.perf_counter()
            for _ in range(n):
                lv = step()
            fence()
            e = (time.perf_counter() - t) / n
            return {"value": B * L / e, "ms_per_step": 1e3 * e, "final_loss": float(lv.detach())}
        single = not (args.overlap_wgrad or args.overlap_decoder)          # the other schedule: everything on one stream
        ops.set_wgrad_overlap(single)
        ops.set_decoder_overlap(single)
        if not single and timer.only is not None:      # ... with the dominant kernel bracketed: its un-co-scheduled duration
            timer.on, timer.rec = True, []
        extras["multi_stream" if single else "single_stream"] = measure()
        if timer.on:
            timer.on = False
            iso = timer.summary().get(timer.only)
            timer.rec = []
            if iso:
                extras["single_stream"]["dominant_kernel"] = {
                    "kernel": timer.only, "launches": iso[0], "avg_launch_ms": 1e3 * iso[1] / iso[0],
                    "achieved": iso[2] / iso[1] / 1e12, "unit": "TFLOP/s", "frac": iso[2] / iso[1] / 1e12 / F32_MFMA_PEAK_TFLOPS,
                    "note": "same kernel with every launch on ONE stream: the timed region co-schedules three streams, so its "
                            "per-kernel wall durations include time shared with other kernels"}
        ops.set_wgrad_overlap(args.overlap_wgrad)
        ops.set_decoder_overlap(args.overlap_decoder)
        if args.precision == "f32":
            ops.set_precision("bf16x3")
            extras["bf16x3_fwd_dgrad"] = dict(measure(), note="forward parity <= 1e-3 vs reference goldens is tested in "
                                              "tests/test_hip_model.py[bf16x3]; backward-weights stays fp32")
            ops.set_precision("f32")

    if rank == 0:
        samples = world * B * L * args.steps
        if args.mode == "stream":
            updates = world * B * (sum(sched) + L) * args.steps
        out = {"metric": "depth samples/sec (346x260 cropped to %dx%d, 5-bin grids, K=5 grids + 1 frame per sample; %s)"
                         % (H, W, {"train": "training step", "infer": "inference", "stream": "asynchronous streaming inference"}[args.mode]),
               "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.precision == "f32" else "bf16x3(fwd,dgrad)+f32(wgrad)", "data": "synthetic",
               "config": {"workload": "EventScape-shaped 346x260 -> %dx%d crop, 5 event bins, K=5, batch %d/GPU, seq-len %d, "
                                      "1xMI355X-per-rank %s, state=%s, SI loss on [image,events4], Adam, backward-weights %s"
                                      % (H, W, B, L, args.mode, args.state, "co-scheduled on a side stream" if args.overlap_wgrad else "on the main stream") +
                                      (", decoders on a second stream" if args.overlap_decoder and args.mode == "train" else ""),
                          "global_batch": B * world, "seq_len": L, "parallelism": "dp%d" % world},
               "final_loss": loss_val, "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
        if args.mode == "stream":
            out["stream"] = {"updates_per_s": updates / dt, "ms_per_update_and_decode": 1e3 * dt / (updates / (world * B)),
                             "grids_per_frame": sched, "note": "one update = fold one event grid or frame into the persistent "
                             "state + decode one depth map; samples/s counts frames"}
        if extras:
            out["extras"] = extras
        agg = agg_timed
        if agg:
            dom = max(agg.items(), key=lambda kv: kv[1][1])
            name, (n, secs, flops) = dom
            ach = flops / secs / 1e12
            traffic = None
            try:    # HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_p_pmc_hbm_traffic.json")))      # tools/pmc_traffic.py
                ent = pmc.get(name.replace(",1>", ",0>") if name.startswith("conv_igemm") else name)
                if ent:
                    traffic = (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ent.values()) /
                               max(1, sum(v["launches"] for v in ent.values())))
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                               "traffic_note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KB, launch-weighted over the kernel's grids, "
                                               "from profiles/r01_p_pmc_*; gfx950 FETCH_SIZE counts 1/2 of wide reads (MI355X_MICROARCH.md)",
                               "launches": n, "avg_launch_ms": 1e3 * secs / n,
                               "algorithmic_gflop_per_launch": flops / n / 1e9}
            iso = extras.get("single_stream", {}).get("dominant_kernel")
            if iso and iso["kernel"] == name:      # the same kernel without co-scheduled neighbours (single-stream extras pass)
                out["roofline"].update({"isolated_achieved": iso["achieved"], "isolated_frac": iso["frac"],
                                        "isolated_avg_launch_ms": iso["avg_launch_ms"],
                                        "schedule_note": "the timed region runs three streams (main, decoders, backward-weights): achieved/frac use "
                                                         "wall durations that include time shared with co-scheduled kernels; isolated_* = same "
                                                         "kernel, same launches, one stream (extras.single_stream)"})
            if "wino" in name:   # Winograd F(2x2,3x3) executes 16/36 of the algorithmic multiplies on the MFMA pipe
                out["roofline"]["mfma_flop_executed_frac_of_peak"] = out["roofline"].get("isolated_achieved", ach) / 2.25 / F32_MFMA_PEAK_TFLOPS
                out["roofline"]["note"] = ("achieved/frac count the ALGORITHMIC FLOP of the convolution (2*B*H*W*9*Cin*Cout, SURVEY 8d) as the "
                                           "contract asks; the kernel executes 1/2.25 of them (Winograd), see mfma_flop_executed_frac_of_peak")
            src = warm if warm else agg
            if warm and args.warmup > 0:      # overlap-proof view: all MFMA FLOP of a step over the step's wall time
                step_flop = sum(v[2] for v in warm.values()) / args.warmup
                out["step_mfma"] = {"algorithmic_tflop_per_step": step_flop / 1e12,
                                    "achieved": step_flop / (dt / args.steps) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": step_flop / (dt / args.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS}
            out["kernels_warmup_steps"] = {k: {"launches": v[0], "ms": 1e3 * v[1], "tflops": v[2] / v[1] / 1e12}
                                           for k, v in src.items()}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, H, W, K, args)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
