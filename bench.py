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

Roofline accounting.  The convolutions are MFMA-bound.  Every MFMA launch is classified by the kernel the
library reports (ramnet_last_kernel) and carries two FLOP counts: ALGORITHMIC (2*B*Ho*Wo*taps*Cin*Cout of the layer it
stands for) and EXECUTED (what the MFMA pipe really multiplies: Winograd F(2x2,3x3) 16/36 of the 3x3 taps —
12.25/25 for forward / backward-data of the stride-2 5x5 encoders run as 3x3 over the space-to-depth view with the positions
of its zero slices skipped (16/25 for their backward-weights) —,
F(2x2,4x4) 25/100 of a folded decoder).
`roofline.achieved/frac` use the EXECUTED count (a fraction of the fp32 MFMA peak that cannot exceed 1);
`roofline.algorithmic_achieved` is the layer-level rate.  `traffic` comes from the newest tracked
profiles/r*_pmc_hbm_traffic.json (separate rocprofv3 --pmc passes of this command, tools/pmc_traffic.py).
"""
import argparse
import contextlib
import glob
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
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense fp32
HBM_PEAK_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq-len", type=int, default=8)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=344)
    ap.add_argument("--full-frame", action="store_true",
                    help="the raw 260 x 346 frame (SURVEY 8d 'also report 264x352'): inputs and targets at 260 x 346, model.set_full_frame() reflect-pads "
                         "them to 264 x 352 inside the input repack and crops the predictions back (utils/inference_utils.py:287-314)")
    ap.add_argument("--events-per-grid", type=int, default=200000)
    ap.add_argument("--bins", type=int, default=5, help="event voxel-grid bins (configs[4] uses 10)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for 1-GPU functional tests)")
    ap.add_argument("--mode", choices=["train", "infer", "stream"], default="train",
                    help="train: configs[1]/[2] step; infer: forward over the same sequences; stream: configs[3]-style batch-1 "
                         "asynchronous inference, irregular number of event grids per frame, persistent state")
    ap.add_argument("--state", choices=["convgru", "convlstm"], default="convgru")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the process group and run the bucketed gradient all-reduce even with ONE rank: executes the RCCL "
                         "init + side-stream collective path on a single-GPU box (tests/test_hip_model.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-infer-baseline", action="store_true",
                    help="train mode: also time the CPU oracle's inference (B=8, no_grad, 1 warm + 3 timed packages, ~30 s) and print it as "
                         "cpu_baseline_infer beside extras.stream_b1 (the infer / stream modes always print it as cpu_baseline)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-overlap-wgrad", dest="overlap_wgrad", action="store_false",
                    help="run backward-weights on the main stream instead of co-scheduling it with backward-data on a side stream "
                         "(the default schedule; per-kernel durations then include co-scheduled time)")
    ap.add_argument("--no-overlap-decoder", dest="overlap_decoder", action="store_false",
                    help="run the decoders on the main stream instead of a second stream concurrent with the next state update "
                         "(ops.set_decoder_overlap; default schedule)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra (untimed-for-value) single-stream / hipGraph measurements")
    ap.add_argument("--dp-exact-loss", action="store_true",
                    help="N>1: all-reduce the SI-loss statistics (sum d, sum d^2, n per supervised map) before the backward so that the "
                         "N-rank gradient equals the single-rank gradient on the concatenated batch (model/loss.py:9 takes mean(d)^2 over "
                         "the WHOLE batch; SURVEY 8e).  Default: per-rank loss + gradient averaging (standard DDP)")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, initialise the process group, let every rank report in over the collective backend, print the "
                         "JSON line (value null) and exit before any GPU work: the launch contract alone (CPU test)")
    ap.add_argument("--wino2x4", default="auto", help="F(2x4,3x3) kernel selection for A/B runs: auto (library heuristics) | off | force, "
                    "optionally ,min_wgs (ops.set_winograd_2x4)")
    ap.add_argument("--split-operands", action="store_true",
                    help="F(2x4,3x3) forward / backward-data launches on the bf16 matrix pipe with three-term bf16 splits of both operands "
                         "(ops.set_split_operands, csrc/conv_wino6s.hip: fp32-level accuracy, parity suites at unchanged tolerances).  The default "
                         "line stays exact fp32 and carries this variant as extras.split_operands")
    ap.add_argument("--wgrad-2x4", default="auto", choices=["auto", "off", "force"], help="A/B: F(2x4,3x3) backward-weights (ops.set_wgrad_winograd_2x4)")
    ap.add_argument("--no-time-batching", action="store_true", help="A/B: the pass-by-pass package loop instead of the time-batched forward (ops.set_time_batching(False))")
    ap.add_argument("--no-relu-premask", action="store_true", help="A/B: ReLU mask of the time-batched encoders in the loaders of their backward launches "
                    "(round 5) instead of in the fan-in of their gradient (ops.set_relu_premask)")
    ap.add_argument("--time-batch-max-decodes", type=int, default=0, help="A/B: decodes per chain of the time-batched forward (0 = a whole group)")
    ap.add_argument("--no-configs4-extra", action="store_true", help="skip extras.configs4_shape (a ~20 s run of the 480x640 / 10-bin / B=4 / L=16 workload in a process of its own)")
    ap.add_argument("--no-gru-bwd-fused", action="store_true", help="A/B: ConvGRU backward stage B as its own launch (ops.set_gru_bwd_fused(False))")
    ap.add_argument("--wgrad-wino-nf", type=int, default=0, help="A/B: 32-channel output blocks per workgroup of the Winograd backward-weights kernel (1 or 2)")
    ap.add_argument("--wgrad-wino-blocks", type=int, default=0, help="A/B: workgroups per Winograd backward-weights launch (<= 384; default 384)")
    ap.add_argument("--wgrad-atomic", action="store_true", help="A/B: Winograd backward-weights splits meet by atomic adds instead of per-split slabs")
    ap.add_argument("--wgrad-defer", type=int, default=2, help="ConvGRU cell updates per deferred multi-segment backward-weights launch "
                    "(ops.set_wgrad_defer; 0 = every update launches its own; 2 since round 6: 241.8 against 240.7 samples/s same-box, "
                    "profiles/r06_h_tuning_notes.md section 6; round 5 measured 5 and more as losses)")
    ap.add_argument("--resident-inputs", action="store_true",
                    help="train: time the step on input tensors that are already resident in HBM (the definition of `value` up to round 4) "
                         "instead of the default, which voxelises fresh event lists and uploads frames / targets every step beside the compute "
                         "(InputSide); the default line carries the resident-input number as extras.resident_inputs")
    ap.add_argument("--input-stream", dest="input_inline", action="store_false",
                    help="A/B: the next step's voxelisation on a low-priority input stream beside the compute (round 5's schedule) instead of the "
                         "step's own stream, in front of its forward pass (default since round 6: a fourth stream of 160 KB-LDS workgroups beside "
                         "the three compute streams cost 1.6 % — 229.7 / 230.8 against 233.6 / 234.2 samples/s on one box, "
                         "profiles/r06_sched_ab.txt — and 2.7 % with split operands, whose workgroups need a whole CU)")
    ap.add_argument("--graph", action="store_true",
                    help="train: the timed step replays ONE hipGraph (gradient zero-fill, forward, loss, BPTT backward, gradient fold; "
                         "rpg_ramnet_amd.graph.GraphedTrainStep) instead of ~7000 eager launches; per-kernel HIP events are then "
                         "unavailable (no roofline object).  stream / infer modes always report both eager and graph replay")
    return ap.parse_args()


def subprocess_measure(extra, note, steps=2, warmup=1, timeout=240):
    """A short run of this script in a process of its own with `extra` arguments appended: the default line then carries a driver-timed
    number for a second workload / variant without disturbing the timed region of the headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-extras",
           "--no-kernel-timing"] + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["WORLD_SIZE"] = "1"
    t = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "peak_hbm_gb": d["peak_hbm_gb"],
                "final_loss": d.get("final_loss"), "workload": d["config"]["workload"], "wall_s": time.perf_counter() - t, "note": note}
    except Exception as ex:     # noqa: BLE001
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}


def configs4_measure():
    """BASELINE configs[4]'s per-GPU workload (480 x 640, 10-bin grids, batch 4, sequence length 16: ~100 GB of HBM)."""
    return subprocess_measure(["--height", "480", "--width", "640", "--bins", "10", "--batch", "4", "--seq-len", "16"],
                              "configs[4] shape on ONE GPU (its 8-GPU form shards sequences like configs[2]); same binary, a separate process")


def split_operands_measure():
    """the same training step with the F(2x4,3x3) forward / backward-data launches on split bf16 operands."""
    return subprocess_measure(["--split-operands"], "the same training step with the F(2x4,3x3) forward / backward-data launches (ConvGRU, residual blocks, "
                              "space-to-depth encoders: 45.7 % of the GPU time) on v_mfma_f32_32x32x16_bf16 with three-term bf16 splits of both operands "
                              "(six of nine partial products, fp32 accumulation; csrc/conv_wino6s.hip).  fp32-level accuracy: the operator, model, "
                              "long-horizon and B=8 L=8 gradient parity tests run this variant at unchanged tolerances (tests/test_hip_ops.py "
                              "algo3x3=winograd2x4_split, tests/test_hip_fullsize.py f2x4_split / test_bench_step_B8_L8_vs_oracle_split_operands).  "
                              "NOT the headline: `value` is exact fp32 on v_mfma_f32_32x32x2_f32; backward-weights stays exact fp32 here too", steps=3)


def full_frame_measure():
    """SURVEY 8d config 2 'also report 264x352': the raw 260 x 346 frame, reflect-padded to 264 x 352 inside the input repack."""
    return subprocess_measure(["--full-frame"], "the same training step on the raw 346x260 frame (reflect-padded to 264x352 in the input repack, "
                              "predictions cropped back; utils/inference_utils.py:287-314); same binary, a separate process", steps=3)


def synth_sequence(model, B, L, H, W, K, bins, n_events, seed, keep_events=None):
    """SURVEY section 8d recipe, generated on device; returns a list of L item dicts (NCHW device tensors).  The B x K event
    lists of a package go through ONE batched scatter-add launch and one batched nonzero normalisation.  keep_events: a list that
    receives the packed event lists of every package (voxel.pack_event_lists) — the raw input of the per-step input pipeline."""
    from rpg_ramnet_amd import voxel
    dev = model.gpu
    g = torch.Generator(device=dev).manual_seed(seed)
    seq = []
    for _ in range(L):
        item, lists = {}, []
        for _k in range(K * B):
            ev = torch.empty(n_events, 4, device=dev, dtype=torch.float64)
            ev[:, 0] = torch.sort(torch.rand(n_events, device=dev, generator=g, dtype=torch.float64) * 0.05)[0]
            ev[:, 1] = torch.randint(0, W, (n_events,), device=dev, generator=g).double()
            ev[:, 2] = torch.randint(0, H, (n_events,), device=dev, generator=g).double()
            ev[:, 3] = torch.randint(0, 2, (n_events,), device=dev, generator=g).double()
            lists.append(ev)
        if keep_events is not None:
            packed = voxel.pack_event_lists(lists, dev)
            keep_events.append(packed)
            grids = voxel.events_to_voxel_grids_packed(*packed, bins, W, H, normalize=True).view(K, B, bins, H, W)
        else:
            grids = voxel.events_to_voxel_grids(lists, bins, W, H, dev, normalize=True).view(K, B, bins, H, W)
        for k in range(K):
            item["events%d" % k] = grids[k]
        item["image"] = torch.rand(B, 1, H, W, device=dev, generator=g)
        for key in ("image", "events%d" % (K - 1)):
            u = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.98 + 0.02
            item["depth_" + key] = torch.clamp(1.0 + torch.log(u) / 3.70378, 0.0, 1.0)
        seq.append(item)
    return seq


# ---------------------------------------------------------------------------------------------------------------- timing
# 3x3 view of a 5x5 stride-2 layer: the parity group (a, b) of a channel keeps (4 - a) x (4 - b) of the 16 Winograd positions
S2D_SPARSE_FACTOR = (16 + 12 + 12 + 9) / 64.0


def winograd_factor(kernel):
    """Multiplies the MFMA pipe executes per multiply of the tap list the launch was given: F(2x2,3x3) 16 per 36,
    F(2x2,4x4) 25 per 64; direct kernels 1."""
    if kernel.startswith(("conv_wino_kernel", "conv_wino_r_kernel", "conv_wgrad_wino_kernel", "conv_wgrad_wino_r_kernel")):
        return 16.0 / 36.0
    # (conv_wino_r6s_kernel, --split-operands: the same 24 products, each as six bf16 partial products — counted here as fp32-equivalent FLOP)
    if kernel.startswith(("conv_wino_r6_kernel", "conv_wino_r6s_kernel", "conv_wgrad_wino_r6_kernel")):      # F(2x4,3x3): a 4 x 6 grid of products per 2 x 4 outputs x 9 taps
        return 24.0 / 72.0
    if kernel.startswith(("conv_wino24_kernel", "conv_wgrad_wino24_kernel")):
        return 25.0 / 64.0
    return 1.0


# algorithmic HBM bytes of the HBM-bound library calls, from their C arguments (include/ramnet_hip.h; SURVEY 8d figures:
# voxelizer 32 B/event + 2 atomic fp32 RMW (16 B) + grid zero-fill; SI loss 8 B/pixel; pred 4*(C+1) B/pixel)
HBM_CALLS = {
    "ramnet_voxelize": lambda a: 48.0 * a[1] + 4.0 * a[2] * a[3] * a[4],
    "ramnet_voxelize_batch": lambda a: None,           # bytes filled in by the caller (needs the event count)
    "ramnet_si_loss_fwd": lambda a: 8.0 * a[2],
    "ramnet_si_loss_bwd": lambda a: 12.0 * a[2],
    "ramnet_pred_sigmoid_fwd": lambda a: (4.0 * a[2] + 4.0) * a[6],
    "ramnet_pred_sigmoid_bwd": lambda a: (8.0 * a[2] + 8.0) * a[10],
    # prediction layer + SI loss in one pass (round 5): x read, y written, target read / x read, dx written, y and target read
    "ramnet_pred_sigmoid_si_fwd": lambda a: (4.0 * a[2] + 8.0) * a[6] * a[7],
    "ramnet_pred_sigmoid_si_bwd": lambda a: (8.0 * a[2] + 8.0) * a[6] * a[7],
    "ramnet_gru_bwd_a": lambda a: 28.0 * a[8] * a[7],      # (a[9] = leading dimension of dh')
    "ramnet_gru_bwd_a2": lambda a: 28.0 * a[8] * a[7],     # (stage B runs in the epilogue of the candidate convolution's backward-data launch)
    "ramnet_gru_bwd_b": lambda a: 24.0 * a[6] * a[5],
    "ramnet_cat_batch_add": lambda a: 4.0 * a[1] * a[2] * a[3] * (3 if a[5] else 2),
    "ramnet_cat_batch_add_masked": lambda a: 4.0 * a[1] * a[2] * a[3] * (4 if a[5] else 3),      # (+ the feature itself: its ReLU mask)
    "ramnet_pad2_sum": lambda a: 4.0 * a[6] * a[3] * ((2 if a[1] else 1) * a[4] * a[5] + (a[4] + 4) * (a[5] + 4)),
    "ramnet_relu_bwd": lambda a: 12.0 * a[3],
    "ramnet_unpad2_fold": lambda a: 4.0 * a[5] * a[2] * (a[3] * a[4] + (a[3] + 4) * (a[4] + 4)),
    "ramnet_nchw_to_nhwc_pad": lambda a: 4.0 * a[2] * a[4] * a[5] * (a[3] + a[6]),
}


class KernelTimer:
    """HIP-event timing on the launch stream.  MFMA launches (ops.conv_launch / wgrad_launch / conv_launch_multi) are keyed by
    the kernel symbol the library reports and carry algorithmic + executed FLOP; HBM-bound library calls (HBM_CALLS) are keyed
    by their entry point and carry algorithmic bytes."""

    def __init__(self):
        self.rec = []           # (key, start, end, alg_flop, exec_flop, bytes, tag)
        self.on = False
        self.only = None        # when set: bracket only launches of this kernel symbol (keeps the timed region unperturbed)
        self.hbm = False        # also bracket the HBM-bound calls (warm-up steps only)
        self.names = {}         # launch signature -> kernel symbol
        self.ext = {}           # raw stream handle -> torch.cuda.ExternalStream (event brackets on explicitly named launch streams)

    def _bracket(self, fn, name_of, sig, alg, tap_ratio, tag=None, nbytes=0.0, sparse=1.0, stream=None):
        """sig: hashable launch signature -> kernel symbol, learnt while every launch is bracketed (warm-up); with `only` set,
        launches whose signature maps to another symbol run un-bracketed.  tap_ratio = taps in the launch's list / taps of
        the layer it stands for (16/25 folded decoder, 36/25 space-to-depth encoder, else 1)."""
        if self.only is not None and self.names.get(sig, self.only) != self.only:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = None
        if stream is not None:            # the launch names its stream (backward-weights on the side stream): bracket it THERE
            raw = getattr(stream, "value", stream)
            st = self.ext.get(raw)
            if st is None:
                st = self.ext[raw] = torch.cuda.ExternalStream(raw)
        s.record(st) if st is not None else s.record()
        fn()
        e.record(st) if st is not None else e.record()
        name = self.names[sig] = name_of()
        if self.only is not None and name != self.only:
            return
        if name.startswith(("conv_wino_r6_kernel", "conv_wino_r6s_kernel")):      # the F(2x4) kernels run the space-to-depth views dense (no skipped zero slices)
            sparse = 1.0
        self.rec.append((name, s, e, alg, alg * tap_ratio * sparse * winograd_factor(name), nbytes, tag))

    def install(self):
        from rpg_ramnet_amd import ops, _hip as Hh
        timer = self
        conv0, wgrad0, multi0 = ops.conv_launch, ops.wgrad_launch, ops.conv_launch_multi

        def last():
            return Hh.lib().ramnet_last_kernel().decode()

        def cin_of(x0, kw, w=None):
            c = ((kw.get("C0") or x0.shape[3]) + kw.get("C1", 0)) * (4 if kw.get("in_mode", 0) == Hh.IN_S2D else 1)
            if isinstance(w, ops.PackRef) and not w.transposed:
                c = min(c, w.cp.Cin)              # head layers: 1 / 5 / 10 real channels inside a padded quad
            return c

        def sig_of(kind, x0, taps, Cout, kw):
            return (kind, tuple(x0.shape), id(taps), Cout) + tuple(kw.get(k) if not torch.is_tensor(kw.get(k)) else 1 for k in
                                                                  ("stride", "in_mode", "C0", "C1", "epi", "Ho", "Wo", "wino24", "out_s2d", "beta", "frame", "gview")) + (len(kw["segs"]) if kw.get("segs") is not None else 0,)

        def conv_bytes(x0, taps, Cout, kw, cin, Ho, Wo, nout, nclass):
            """Operand bytes of one forward / backward-data launch: every input, mask and epilogue operand read once, every
            output written once, the weights once (what the launch has to move if nothing is fetched twice)."""
            px_in, px_out = x0.shape[0] * x0.shape[1] * x0.shape[2], x0.shape[0] * Ho * Wo * nclass
            mode, c1 = kw.get("in_mode", 0), kw.get("C1", 0)
            rd = px_in * ((kw.get("C0") or x0.shape[3]) + c1)
            if mode == Hh.IN_RELUMASK:
                rd += px_in * x0.shape[3]
            elif mode == Hh.IN_CAT_MUL:
                rd += px_in * c1
            epi = kw.get("epi", Hh.EPI_LINEAR)
            extra = {Hh.EPI_GRU_BLEND: 3, Hh.EPI_RES_RELU: 1, Hh.EPI_LSTM: 2 + (4 if kw.get("o2") is not None else 0),
                     Hh.EPI_SIGMOID_HR: 1}.get(epi, 0)      # (h read and h.r written for the reset gate's half of the 2C outputs)
            extra += 1 if kw.get("beta") else 0
            wr = px_out * Cout * (1 + extra)
            return 4.0 * (rd + wr + nclass * taps.n * cin * nout)

        def conv(x0, taps, w, out, Cout, **kw):
            if not timer.on:
                return conv0(x0, taps, w, out, Cout, **kw)
            Ho, Wo = kw.get("Ho") or out.shape[1], kw.get("Wo") or out.shape[2]
            parity4 = kw.get("in_mode", 0) == Hh.IN_PARITY4
            nclass = 4 if (kw.get("wino24") and not parity4) else 1
            cin, nout = cin_of(x0, kw, w), Cout * (4 if kw.get("epi") == Hh.EPI_LSTM else 1)
            ratio = taps.n / float(taps.flop_taps)
            sparse = 1.0
            if ops._S2D_SPARSE and (kw.get("in_mode", 0) == Hh.IN_S2D or kw.get("out_s2d")):
                sparse = S2D_SPARSE_FACTOR      # positions of the zero slices are not issued (ramnet_conv_desc.s2d_5x5; F(2x2) kernel only)
            if parity4:     # decoder backward-data: stands for a 5x5 convolution over the full-resolution gradient; runs 16 taps
                alg = 2.0 * x0.shape[0] * x0.shape[1] * x0.shape[2] * 25 * x0.shape[3] * Cout      # over 4*C0 channels on the
                ratio = 16.0 * 4 * Ho * Wo / (25.0 * x0.shape[1] * x0.shape[2])                    # padded low-resolution grid
            else:
                alg = 2.0 * nclass * x0.shape[0] * Ho * Wo * taps.flop_taps * cin * nout
            tag = None
            if kw.get("epi") in (Hh.EPI_SIGMOID, Hh.EPI_SIGMOID_HR, Hh.EPI_GRU_BLEND) and kw.get("in_mode", 0) in (Hh.IN_CAT, Hh.IN_CAT_MUL):
                tag = "gru_fwd_C%d" % (kw.get("C1") or 0)
            timer._bracket(lambda: conv0(x0, taps, w, out, Cout, **kw), last, sig_of("c", x0, taps, Cout, kw) + (isinstance(w, ops.PackRef),),
                           alg, ratio, tag, conv_bytes(x0, taps, Cout, kw, cin, Ho, Wo, nout, nclass), sparse)

        def wgrad(x0, taps, dout, dw, Cout, **kw):
            if not timer.on:
                return wgrad0(x0, taps, dout, dw, Cout, **kw)
            ho, wo = kw.get("Ho") or dout.shape[1], kw.get("Wo") or dout.shape[2]     # (parity sub-grid of a folded decoder)
            nclass = 4 if kw.get("wino24") else 1
            cin = getattr(dw, "head_cin", 0) or cin_of(x0, kw)
            nseg = len(kw["segs"]) if kw.get("segs") is not None else 1       # a multi-segment launch covers nseg launches' tensors
            alg = 2.0 * nclass * nseg * dout.shape[0] * ho * wo * taps.flop_taps * cin * Cout
            # operand bytes of a backward-weights launch: input (+ its mask / product operand) and output gradient (+ ReLU mask) read once,
            # the gradient workspace — [16] (Winograd), [taps] (direct) or [4][25] (folded decoder) x Cin x Cout — read and written once
            px_in, px_out = nseg * x0.shape[0] * x0.shape[1] * x0.shape[2], nseg * dout.shape[0] * ho * wo * nclass
            mode, c1 = kw.get("in_mode", 0), kw.get("C1", 0)
            rd = px_in * ((kw.get("C0") or x0.shape[3]) + c1) * (2 if mode == Hh.IN_RELUMASK else 1) + (px_in * c1 if mode == Hh.IN_CAT_MUL else 0)
            rd += px_out * Cout * (2 if kw.get("gmask") is not None else 1)
            npos = 100 if kw.get("wino24") else 24 if getattr(dw, "wino6", False) else 16 if getattr(dw, "wino", False) else taps.n
            timer._bracket(lambda: wgrad0(x0, taps, dout, dw, Cout, **kw), last,
                           sig_of("w", x0, taps, Cout, kw) + (getattr(dw, "wino", False), getattr(dw, "wino6", False), getattr(dw, "head_cin", 0)),
                           alg, taps.n / float(taps.flop_taps), None, 4.0 * (rd + 2 * npos * cin * Cout), stream=kw.get("stream"))

        def multi(x0, w, out, Cout, classes, **kw):
            if not timer.on:
                return multi0(x0, w, out, Cout, classes, **kw)
            cin = cin_of(x0, kw)
            alg = sum(2.0 * x0.shape[0] * Ho * Wo * taps.flop_taps * cin * Cout for taps, Ho, Wo, _ in classes)
            ex = sum(2.0 * x0.shape[0] * Ho * Wo * taps.n * cin * Cout for taps, Ho, Wo, _ in classes)
            timer._bracket(lambda: multi0(x0, w, out, Cout, classes, **kw), last,
                           sig_of("m", x0, classes[0][0], Cout, kw) + (len(classes), classes[0][1], classes[0][2]), alg, ex / alg)

        ops.conv_launch, ops.wgrad_launch, ops.conv_launch_multi = conv, wgrad, multi

        def tracer(name, fn, args):
            f = HBM_CALLS.get(name)
            if f is None or not (timer.on and timer.hbm):
                return fn(*args)
            vals = [getattr(a, "value", a) for a in args]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            timer.rec.append((name, s, e, 0.0, 0.0, f(vals) or 0.0, None))
            return rc
        self.tracer = tracer

    def summary(self, by_tag=False):
        agg = {}
        for name, s, e, alg, ex, nbytes, tag in self.rec:
            key = tag if by_tag else name
            if key is None:
                continue
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e) * 1e-3
            a[2] += alg
            a[3] += ex
            a[4] += nbytes
        return agg


def newest_pmc_traffic():
    """The PMC summary (tools/pmc_traffic.py) that belongs to this tree: profiles/CURRENT_PMC names it (tools/profile_round.sh's copy step
    writes that one line); without the pointer, the file of the highest (round, then modification time) — never plain lexicographic order,
    in which `r04_zzzz` beats the newer `r04_final`.  {kernel: {grid: {launches, hbm_bytes_per_launch}}}."""
    ptr = os.path.join(ROOT, "profiles", "CURRENT_PMC")
    if os.path.exists(ptr):
        name = open(ptr).read().split()[0]
        return name, json.load(open(os.path.join(ROOT, "profiles", name)))
    files = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_hbm_traffic.json"))
    if not files:
        raise FileNotFoundError("no profiles/r*_pmc_hbm_traffic.json: run the rocprofv3 --pmc passes (tools/pmc_traffic.py) and commit it")
    best = max(files, key=lambda f: (os.path.basename(f)[:3], os.path.getmtime(f)))
    return os.path.basename(best), json.load(open(best))


def pmc_launches_of(pmc, name):
    """Launches of kernel symbol `name` in the PMC pass (ONE step, no warm-up: tools/profile_round.sh)."""
    want = name.replace(" ", "")
    n = 0
    for k, ent in pmc.items():
        kk = k.replace(" ", "").replace("(bool)", "").replace("true", "1").replace("false", "0")
        if kk == want or kk.split("<")[0] == want:
            n += sum(v["launches"] for v in ent.values())
    return n


def traffic_of(pmc, name, fetch_x2=False):
    """Launch-weighted HBM bytes per launch of kernel symbol `name` (template arguments: rocprofv3 prints ", " separators):
    FETCH_SIZE + WRITE_SIZE as counted — FETCH_SIZE = TCC_EA0_RDREQ x 64 B (MI355X_MICROARCH.md), which profiles/r05_zx_pmc_l2.txt confirms
    request by request for these kernels (118.9 MB per launch of the dominant kernel either way).  fetch_x2: the guide's correction for
    128-byte streaming reads (2 x FETCH_SIZE + WRITE_SIZE) — it does not apply to the 32-byte-per-pixel patch loads of the convolution
    kernels and is only kept as a second figure."""
    want = name.replace(" ", "")
    hit = None
    for k, ent in pmc.items():
        kk = k.replace(" ", "").replace("(bool)", "").replace("true", "1").replace("false", "0")
        if kk == want or kk.split("<")[0] == want:
            hit = ent if hit is None else {**hit, **{g + "'": v for g, v in ent.items()}}
    if not hit:
        return None
    n = sum(v["launches"] for v in hit.values())
    if fetch_x2:
        return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit.values()) / max(1, n)
    return sum((v["fetch_kb_raw"] + v["write_kb"]) * 1024.0 * v["launches"] for v in hit.values()) / max(1, n)


def cpu_baseline(cfg, H, W, K, args, mode=None):
    """The oracle (CPU restatement, fixture-pinned to the reference) timed on this box's host cores on a bounded sample of the
    same workload.  One untimed warm-up at B=1, L=1 first.
    train: ONE full training step — forward, SI loss on [image, events4], BPTT backward over the packages, Adam — at B=8, L=2
    (two data packages per sequence: a quarter of the L=8 step, which is linear in L; ~50-60 s of CPU work on the 128 host cores).
    infer / stream: forward under no_grad at B=8 with persistent state: 1 warm-up package, then 3 timed packages (BASELINE.md asks 3 + 5;
    bounded to keep the run to minutes: a package-batch is ~5-8 s here)."""
    from oracle import ramnet_ref
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipe import make_item
    from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent
    mode = mode or args.mode
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):      # the constructor prints (like the reference's); stdout carries ONE JSON line
        m = ERGB2DepthRecurrent(cfg)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=3e-4)
    rng = np.random.default_rng(0)
    cores = torch.get_num_threads()
    lc = cfg["loss_composition"]

    def train_step(B, L):
        seq = [make_item(rng, B, H, W, K, cfg["num_bins_events"], 1, True, 0.0) for _ in range(L)]
        t0 = time.time()
        opt.zero_grad()
        total, _ = ramnet_ref.sequence_loss(sd, cfg, seq, lc, [1, 1])
        total.backward()
        opt.step()
        return time.time() - t0

    def infer(B, warm, timed):
        seq = [make_item(rng, B, H, W, K, cfg["num_bins_events"], 1, True, 0.0) for _ in range(warm + timed)]
        with torch.no_grad():
            prev, lstm = None, ramnet_ref.empty_states_lstm(K)
            for i, item in enumerate(seq):
                if i == warm:
                    t0 = time.time()
                _, supers, lstm = ramnet_ref.forward_recurrent(sd, cfg, item, prev, lstm)
                prev = supers["image"]
        return time.time() - t0

    if mode == "train":
        warm = train_step(1, 1)
        B, L = args.batch, min(2, args.seq_len)
        dt = train_step(B, L)
        return {"value": B * L / dt, "unit": "samples/s", "cores": cores, "kind": "port",
                "sample": "train step (fwd, SI loss, BPTT bwd, Adam), B=%d L=%d K=%d %dx%d fp32 torch-CPU oracle (own restatement, fixture-pinned "
                          "to the reference), %.1f s after a %.1f s B=1 L=1 warm-up step" % (B, L, K, H, W, dt, warm)}
    B, warm_n, timed_n = args.batch, 1, 3
    dt = infer(B, warm_n, timed_n)
    return {"value": B * timed_n / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "inference (no_grad, persistent state), B=%d K=%d %dx%d fp32 torch-CPU oracle (own restatement, fixture-pinned to the "
                      "reference): %d warm-up + %d timed packages, %.1f s" % (B, K, H, W, warm_n, timed_n, dt)}


def stream_b1_measure(model, seq, K, H, W, timer, frames=8, reps=3):
    """BASELINE configs[3] at its own shape: batch-1 asynchronous streaming inference, persistent state, an irregular number of
    event grids (1..8) before each frame, hipGraph replays with the decode of update k pipelined behind update k+1
    (graph.GraphedStream).  Returns ms per update+decode and the MFMA roofline of the whole chain: FLOP counted by the launch
    hooks over ONE eager pass of the same schedule (every launch of the chain), divided by the replayed wall time."""
    from rpg_ramnet_amd.graph import GraphedStream, TimeBatchedStream
    was_training = model.training
    model.eval()
    rs = np.random.default_rng(7)
    sched = [int(v) for v in rs.integers(1, 9, size=frames)]
    items = [{k: v[:1].contiguous() for k, v in seq[l % len(seq)].items() if not k.startswith("depth_")} for l in range(frames)]
    n_upd = sum(sched) + frames

    def eager():
        st = model.init_states(1, H, W)
        with torch.no_grad():
            for l, item in enumerate(items):
                for k in range(sched[l]):
                    st, _ = model.update_events(item["events%d" % (k % K)], st)
                    pred = model.decode(st)
                st, _ = model.update_image(item["image"], st)
                pred = model.decode(st)
        return pred

    on0, only0, hbm0, rec0 = timer.on, timer.only, timer.hbm, timer.rec
    timer.on, timer.only, timer.hbm, timer.rec = True, None, False, []
    eager()
    torch.cuda.synchronize()
    agg = timer.summary()
    timer.on, timer.only, timer.hbm, timer.rec = on0, only0, hbm0, rec0
    alg, ex, nl = sum(v[2] for v in agg.values()), sum(v[3] for v in agg.values()), sum(v[0] for v in agg.values())
    def runner(g):
        def run():
            for l, item in enumerate(items):
                for k in range(sched[l]):
                    pred = g.update_events(item["events%d" % (k % K)])
                pred = g.update_image(item["image"])
            return g.wait(pred)
        return run

    def clock(fn):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    tb = TimeBatchedStream(model, 1, H, W, max_events=8)

    def run_tb():
        for l, item in enumerate(items):
            for k in range(sched[l]):
                tb.push_events(item["events%d" % (k % K)])
            pred = tb.push_image(item["image"])
        return tb.wait(pred)

    run_tb()                                            # first pass records the graphs of every group shape of the schedule
    dt = clock(run_tb)
    dt2 = clock(runner(GraphedStream(model, 1, H, W, pipelined=True)))
    dt1 = clock(runner(GraphedStream(model, 1, H, W, pipelined=False)))

    def live(g):                                        # a live sensor: the host waits for every depth map before the next measurement
        def run():
            for l, item in enumerate(items):
                for k in range(sched[l]):
                    g.update_events(item["events%d" % (k % K)])
                    torch.cuda.synchronize()
                g.update_image(item["image"])
                torch.cuda.synchronize()
        return run
    from rpg_ramnet_amd.graph import LatencyStream
    dtl = clock(live(LatencyStream(model, 1, H, W)))
    dtl1 = clock(live(GraphedStream(model, 1, H, W, pipelined=False)))
    dte = clock(eager)
    model.train(was_training)
    out = {"workload": "configs[3]: batch 1, %dx%d, persistent ConvGRU state, %d event grids + %d frames per pass (grids per frame %s), "
                       "update + decode per measurement" % (H, W, sum(sched), frames, sched),
           "ms_per_update_and_decode": 1e3 * dt / n_upd, "updates_per_s": n_upd / dt, "runtime": "graph.TimeBatchedStream: the event grids "
           "up to the next frame form a group — encoders at batch n and decoders at batch n+1 in one launch chain each (they do not depend "
           "on the order of the updates), state updates one by one per scale; groups pipelined (hipGraph replays, four streams)",
           "latency_ms_update_then_decode": 1e3 * dtl / n_upd, "latency_ms_serial_graph_replays": 1e3 * dtl1 / n_upd,
           "serial_ms_per_update_and_decode": 1e3 * dt1 / n_upd, "two_stage_ms_per_update_and_decode": 1e3 * dt2 / n_upd,

           "eager_ms_per_update_and_decode": 1e3 * dte / n_upd, "mfma_launches_per_update": nl / float(n_upd),
           "note": "ms_per_update_and_decode = period of the pipelined stream (throughput over a recorded stream, test.py:205-232); "
                   "latency_ms_update_then_decode = what a live sensor sees: one measurement in, the host synchronises on its depth map "
                   "(graph.LatencyStream: the fine scales' state updates on a second stream beside the coarse update, the residual blocks "
                   "and decoder 0); latency_ms_serial_graph_replays = the same with one update graph + one decode graph on one stream "
                   "(graph.GraphedStream); serial_ms = those replays back to back without the host wait; "
                   "two_stage = decode of update k beside update k+1 (graph.GraphedStream, the round-2 runtime)"}
    if ex > 0:
        out["roofline"] = {"bound": "mfma", "kernel": "whole update+decode chain (%d MFMA launches per update)" % round(nl / float(n_upd)),
                           "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "algorithmic_gflop_per_update": alg / n_upd / 1e9, "executed_gflop_per_update": ex / n_upd / 1e9,
                           "achieved": ex / dt / 1e12, "frac": ex / dt / 1e12 / F32_MFMA_PEAK_TFLOPS,
                           "frac_executed": ex / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_achieved": alg / dt / 1e12,
                           "frac_algorithmic": alg / dt / 1e12 / F32_MFMA_PEAK_TFLOPS,
                           "note": "MFMA FLOP of every launch of the chain (executed = what the pipe multiplies, Winograd 16/36 and 25/100; "
                                   "algorithmic = SURVEY 8d layer count) over the wall time of the pipelined hipGraph replay"}
    return out


def input_side_measure(model, step, seq, args, K, bins, B, L, H, W, rank, n=2):
    """The training step WITH its input side in the timed loop: per step, L batched voxel scatter-adds of
    fresh event lists (B*K lists of --events-per-grid events each, resident in HBM as the loader's raw `*_events.npy` would be after
    their upload), the batched nonzero normalisation, and the host-to-device copies of the frames and the two target maps from
    pinned memory — then the step itself (forward, loss, BPTT backward, all-reduce, Adam) on those tensors."""
    from rpg_ramnet_amd import voxel
    dev = model.gpu
    g = torch.Generator(device=dev).manual_seed(77 + rank)
    n_ev = args.events_per_grid
    lists = []
    for _ in range(K * B):
        ev = torch.empty(n_ev, 4, device=dev, dtype=torch.float64)
        ev[:, 0] = torch.sort(torch.rand(n_ev, device=dev, generator=g, dtype=torch.float64) * 0.05)[0]
        ev[:, 1] = torch.randint(0, W, (n_ev,), device=dev, generator=g).double()
        ev[:, 2] = torch.randint(0, H, (n_ev,), device=dev, generator=g).double()
        ev[:, 3] = torch.randint(0, 2, (n_ev,), device=dev, generator=g).double()
        lists.append(ev)
    from rpg_ramnet_amd.data import DevicePrefetcher
    host = [{k: v.cpu().pin_memory() for k, v in item.items() if not k.startswith("events")} for item in seq]

    def host_sequences():               # the loader's side: the same pinned host sequence over and over
        while True:
            yield host
    uploads = iter(DevicePrefetcher(host_sequences(), dev))      # frames / targets of step n + 1 go up on a copy stream during step n

    def full():
        frames = next(uploads)
        for l in range(L):
            grids = voxel.events_to_voxel_grids(lists, bins, W, H, dev, normalize=True).view(K, B, bins, H, W)
            for k in range(K):
                seq[l]["events%d" % k] = grids[k]
            for k, v in frames[l].items():
                seq[l][k] = v
        return step()

    full()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        lv = full()
    torch.cuda.synchronize()
    e = (time.perf_counter() - t) / n
    return {"value": B * L / e, "ms_per_step": 1e3 * e, "final_loss": float(lv.detach()),
            "note": "per step: %d voxelize_batch launches (%d lists x %d events each, on-device event lists), batched nonzero "
                    "normalisation, H2D of %d frames + %d target maps from pinned host memory on a copy stream during the previous step (data.DevicePrefetcher), then the training step" %
                    (L, K * B, n_ev, L * B, 2 * L * B)}


class InputSide:
    """The input side of a training step INSIDE the timed loop (reference: utils/event_tensor_utils.py:120-187,
    data_loader/dataset.py:254-415): per step, the L x (B*K) event lists of the NEXT step's sequences — resident in HBM, as the loader's
    raw `*_events.npy` are after their upload; two alternating sets — go through L batched voxel scatter-adds + L batched nonzero
    normalisations on an input stream of their own, and the frames / target maps of the next step come up from pinned host memory on a copy
    stream (data.DevicePrefetcher), all beside the compute of the current step; the step then trains on tensors that did not exist one step
    earlier.  Grid buffers alternate between two sets: set p is rewritten only after the step that read it has finished (event wait)."""

    def __init__(self, model, seq, events_a, events_b, K, B, bins, H, W, inline=False):
        from rpg_ramnet_amd.data import DevicePrefetcher
        self.inline = inline            # voxelise on the step's own stream, in front of its forward pass (A/B: --input-inline)
        self.dev, self.K, self.B, self.bins, self.H, self.W, self.L = model.gpu, K, B, bins, H, W, len(seq)
        self.events = [events_a, events_b]
        # the input stream at the LOWEST priority the device offers: its workgroups (up to 160 KB of LDS each) should take the CUs the three
        # compute streams leave idle, not displace their workgroups
        try:
            prio = int(os.environ.get("RAMNET_INPUT_PRIORITY", max(torch.cuda.Stream.priority_range())))
        except Exception:      # noqa: BLE001
            prio = 0
        self.priority = prio
        self.stream = torch.cuda.Stream(device=self.dev, priority=prio)
        self.grids = [[torch.empty(K * B, bins, H, W, device=self.dev) for _ in range(self.L)] for _ in range(2)]
        self.scratch = [[torch.empty(3 * K * B, device=self.dev, dtype=torch.float64) for _ in range(self.L)] for _ in range(2)]
        self.ready, self.free = [None, None], [None, None]
        self.static = [{k: v for k, v in item.items() if not k.startswith("events")} for item in seq]
        host = [{k: v.cpu().pin_memory() for k, v in item.items()} for item in self.static]

        def host_sequences():               # the loader's side: the same pinned host sequence over and over
            while True:
                yield host
        self.uploads = iter(DevicePrefetcher(host_sequences(), self.dev))
        self.n = 0
        self.prepare(0)

    def prepare(self, p):
        """Enqueue the voxelisation of one step's inputs into grid set p on the input stream."""
        from rpg_ramnet_amd import voxel
        st = torch.cuda.current_stream() if self.inline else self.stream
        if self.free[p] is not None:
            st.wait_event(self.free[p])
        with torch.cuda.stream(st):
            for l in range(self.L):
                voxel.events_to_voxel_grids_packed(*self.events[p][l], self.bins, self.W, self.H, out=self.grids[p][l], normalize=True,
                                                   scratch=self.scratch[p][l])
            self.ready[p] = st.record_event()

    def next_sequence(self):
        """The sequence of this step (grids of set p, fresh uploads of frames / targets); starts the next step's input work."""
        p = self.n & 1
        self.n += 1
        torch.cuda.current_stream().wait_event(self.ready[p])
        frames = next(self.uploads)
        seq = []
        for l in range(self.L):
            item = dict(frames[l])
            g = self.grids[p][l].view(self.K, self.B, self.bins, self.H, self.W)
            for k in range(self.K):
                item["events%d" % k] = g[k]
            seq.append(item)
        self.prepare(p ^ 1)
        self._cur = p
        return seq

    def done(self):
        """The step that read the current grid set has been enqueued completely: the set may be rewritten behind it."""
        self.free[self._cur] = torch.cuda.current_stream().record_event()


class ClockSampler:
    """Best-effort effective shader clock during the timed region: a background thread polls `rocm-smi --showclocks
    --json` and keeps the sclk readings; the roofline keeps the 2.4 GHz spec peak, `clock_ghz` says what the chip really ran at
    (the largest launches draw the power budget: 2.10-2.16 GHz measured with GRBM_GUI_ACTIVE, tools/effective_clock.py)."""

    def __init__(self, period=0.25):
        import shutil
        import threading
        self.exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self.samples, self.stop, self.period = [], threading.Event(), period
        self.thread = threading.Thread(target=self._run, daemon=True) if self.exe else None

    def _read(self):
        import re
        import subprocess
        try:
            out = subprocess.run([self.exe, "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            card = next(iter(json.loads(out[out.index("{"):]).values()))      # (a warning line may precede the JSON)
            for k, v in card.items():
                if "sclk" in k.lower():
                    m = re.search(r"(\d+(?:\.\d+)?)\s*mhz", str(v).lower())
                    if m:
                        return float(m.group(1)) * 1e-3
        except Exception:      # noqa: BLE001 — a missing / differently formatted tool must not touch the measurement
            return None
        return None

    def _run(self):
        while not self.stop.is_set():
            v = self._read()
            if v:
                self.samples.append(v)
            self.stop.wait(self.period)

    def __enter__(self):
        if self.thread:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.thread:
            self.thread.join(timeout=10)

    def ghz(self):
        return (sum(self.samples) / len(self.samples)) if self.samples else None


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with no torch.distributed.run environment around it: start the N ranks here (one process per
    GPU, rendezvous on 127.0.0.1, a free port) by re-executing this file under torch.distributed.run with the same arguments; rank
    0's JSON line is the child's stdout.  Under torch.distributed.run (WORLD_SIZE set) this is never reached."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # the children's stdout carries rank 0's JSON line; anything else a library prints there (gloo's connection banner) goes to stderr, so that
    # this process prints exactly ONE line on stdout as the single-process run does
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
    return p.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend_note = None
    if world > 1 or args.force_collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        ndev = torch.cuda.device_count()
        if os.environ.get("RAMNET_BENCH_SINGLE_DEVICE") == "1":     # functional test of the N>1 path on a 1-GPU box
            local = 0
        elif 0 < ndev < world:                                      # fewer GPUs than ranks (a 1-GPU box asked for --gpus 2): ranks share devices
            local = local % ndev
        if args.backend == "nccl" and ndev < world:
            # RCCL refuses two ranks on one device ("Duplicate GPU detected"): the functional run of the N>1 path goes over gloo, and says so
            args.backend = "gloo"
            backend_note = "gloo (fewer devices than ranks: %d < %d; RCCL needs one device per rank)" % (ndev, world)
        if args.launch_check:
            dist.init_process_group(args.backend)
            got = [None] * world
            dist.all_gather_object(got, rank)
            if rank == 0:
                print(json.dumps({"metric": "launch check", "value": None, "n_gpus": world, "rccl_ranks_seen": sorted(got),
                                  "backend": backend_note or args.backend}))
            dist.barrier()
            dist.destroy_process_group()
            return
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: pass the number of ranks torch.distributed.run starts" % (args.gpus, world)
    from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent
    from rpg_ramnet_amd.parallel import FlatGradReducer
    from rpg_ramnet_amd.trainer import sequence_loss, empty_states_lstm, LOSS_SEMANTICS

    from rpg_ramnet_amd import ops, _hip as Hh
    ops.set_wgrad_overlap(args.overlap_wgrad)
    ops.set_decoder_overlap(args.overlap_decoder)
    ops.set_wgrad_slabs(not args.wgrad_atomic)
    ops.set_wgrad_winograd_2x4(args.wgrad_2x4)
    ops.set_gru_bwd_fused(not args.no_gru_bwd_fused)
    if args.wgrad_defer >= 0:
        ops.set_wgrad_defer(args.wgrad_defer)
    ops.set_time_batching(not args.no_time_batching, max_decodes=args.time_batch_max_decodes)
    if args.wgrad_wino_nf:
        Hh.check(Hh.lib().ramnet_set_option(b"wgrad_wino_nf", args.wgrad_wino_nf), "set_option")
    if args.wgrad_wino_blocks:
        Hh.check(Hh.lib().ramnet_set_option(b"wgrad_wino_blocks", args.wgrad_wino_blocks), "set_option")
    w24 = args.wino2x4.split(",")
    ops.set_winograd_2x4(w24[0], min_wgs=int(w24[1]) if len(w24) > 1 else None)
    ops.set_split_operands(args.split_operands)
    ops.set_relu_premask(not args.no_relu_premask)
    if args.full_frame:
        assert args.mode == "train", "--full-frame: the training step (the graph runtimes of the other modes have static 8-aligned buffers)"
        args.height, args.width, args.no_extras = 260, 346, True
    K, bins, B, L, H, W = 5, args.bins, args.batch, args.seq_len, args.height, args.width
    cfg = dict(RELEASED, num_bins_events=bins, gpu=local, every_x_rgb_frame=K, baseline=False, loss_composition=["image", "events4"],
               state_combination=args.state)
    torch.manual_seed(0)                       # identical initial weights on every rank (train.py:203)
    with contextlib.redirect_stdout(sys.stderr):
        model = ERGB2DepthRecurrent(cfg)
    model = model.to(model.gpu)
    if args.full_frame:
        model.set_full_frame(True)
    timer = KernelTimer()
    if not args.no_kernel_timing:
        timer.install()
        Hh.set_tracer(timer.tracer)
    # input side (outside the timed region): B*K event lists per package through the batched voxeliser, bracketed once
    synth_sequence(model, B, 1, H, W, K, bins, args.events_per_grid, seed=999)      # untimed: the library allocates its sort scratch on first use
    torch.cuda.synchronize()
    timer.on = timer.hbm = not args.no_kernel_timing
    want_input_side = args.mode == "train" and not args.resident_inputs and not args.graph
    events_a = [] if want_input_side else None
    seq = synth_sequence(model, B, L, H, W, K, bins, args.events_per_grid, seed=1000 + rank, keep_events=events_a)
    torch.cuda.synchronize()
    vox = timer.summary().get("ramnet_voxelize_batch")
    timer.rec, timer.on, timer.hbm = [], False, False
    inp, use_inp = None, {"on": want_input_side}
    if want_input_side:
        events_b = []
        synth_sequence(model, B, L, H, W, K, bins, args.events_per_grid, seed=5000 + rank, keep_events=events_b)
        inp = InputSide(model, seq, events_a, events_b, K, B, bins, H, W, inline=args.input_inline)
        torch.cuda.synchronize()

    ranks_seen = [0]
    if world > 1 or args.force_collective:      # every rank reports in over the collective backend: the driver's SCALE run can confirm N ranks took part
        t = torch.full((1,), float(rank), device=model.gpu)
        got = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        ranks_seen = sorted(int(v.item()) for v in got)

    graphed = {}
    if args.mode == "train":
        model.train()
        reducer = FlatGradReducer(model, always_collective=args.force_collective)
        # the reference's optimizer (configs/*.json: Adam, lr 3e-4, weight_decay 0) as ONE fused multi-tensor kernel per step: the
        # foreach form's ~11 launches with their host work sit exposed at the step boundary (GPU idle ~3 ms per step, +0.8 %)
        opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=0, fused=True)

        def step():
            live = inp is not None and use_inp["on"]
            s_ = inp.next_sequence() if live else seq         # (live: grids voxelised / frames uploaded beside the previous step)
            reducer.zero()
            total, _ = sequence_loss(model, s_, cfg["loss_composition"], [1, 1], dp_exact=args.dp_exact_loss)
            total.backward()
            if live:
                inp.done()
            reducer.all_reduce()
            reducer.wait()
            opt.step()
            return total

        def make_graph_step():
            from rpg_ramnet_amd.graph import GraphedTrainStep
            g = GraphedTrainStep(model, seq, cfg["loss_composition"], [1, 1], reducer=reducer, warmup=1)

            def gstep():
                total, _ = g()
                reducer.all_reduce()
                reducer.wait()
                opt.step()
                return total
            return gstep
        if args.graph:
            timer.on = False
            step = make_graph_step()
            args.no_kernel_timing = True
    elif args.mode == "stream":
        model.eval()
        rs = np.random.default_rng(7)
        sched = [int(v) for v in rs.integers(1, 9, size=L)]        # event grids before each frame, drawn from {1..8}
        stream_state = {"s": model.init_states(B, H, W)}

        def eager_step():
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

        from rpg_ramnet_amd.graph import GraphedStream, TimeBatchedStream
        # hipGraph replays, time-batched groups (graph.TimeBatchedStream); the per-measurement runtimes (decode of update k beside
        # update k+1: GraphedStream pipelined; strictly serial replays) are extras
        tb = TimeBatchedStream(model, B, H, W, max_events=8)
        gs_two = GraphedStream(model, B, H, W, pipelined=True)
        gs_serial = GraphedStream(model, B, H, W, pipelined=False)

        def run_stream(g):
            for l, item in enumerate(seq):
                for k in range(sched[l]):
                    pred = g.update_events(item["events%d" % (k % K)])
                pred = g.update_image(item["image"])
            return g.wait(pred).mean()

        def step():
            for l, item in enumerate(seq):
                for k in range(sched[l]):
                    tb.push_events(item["events%d" % (k % K)])
                pred = tb.push_image(item["image"])
            return tb.wait(pred).mean()
        graphed["eager"] = eager_step
        graphed["graph_serial"] = lambda: run_stream(gs_serial)
        graphed["graph_two_stage"] = lambda: run_stream(gs_two)
    else:
        model.eval()

        def eager_step():
            prev_super, prev_lstm = None, empty_states_lstm(K)
            with torch.no_grad():
                for item in seq:
                    preds, supers, prev_lstm = model(item, prev_super, prev_lstm)
                    prev_super = supers["image"]
            return preds["image"].mean()

        from rpg_ramnet_amd.graph import GraphedPackage, TimeBatchedStream
        gp = GraphedPackage(model, seq[0]) if args.state == "convgru" else None      # one hipGraph per data package (ConvGRU states)

        def package_step():
            gp.reset()
            for item in seq:
                preds = gp(item)
            return preds["image"].mean()
        # a data package is one group of the time-batched runtime: its K event grids encode at batch K*B, its K+1 decodes run as
        # one chain at batch (K+1)*B, the state updates one by one at batch B (graph.TimeBatchedStream)
        tbi = TimeBatchedStream(model, B, H, W, max_events=K + 1)

        def step():
            tbi.reset()
            for item in seq:
                for k in range(K):
                    tbi.push_events(item["events%d" % k])
                out = tbi.push_image(item["image"])
            return tbi.wait(out)[K].mean()
        graphed["eager"] = eager_step
        if gp is not None:
            graphed["graph_per_package"] = package_step

    def fence():
        torch.cuda.synchronize()
        if world > 1 or args.force_collective:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up steps bracket EVERY MFMA launch and every HBM-bound call with HIP events (per-kernel tables + choice of the
    # dominant symbol); in the timed region only the dominant symbol's launches are bracketed: ~7000 event pairs per step cost
    # 3-4 % of the step, ~1000 cost < 0.5 %.
    last = None
    # (stream / infer: the steps replay hipGraphs — and record them on first use —, where HIP events cannot be bracketed; their
    # FLOP accounting comes from one eager pass below)
    timer.on = timer.hbm = not args.no_kernel_timing and args.mode == "train"
    # (the clock is sampled over the WARM-UP steps — the same workload, untimed: nothing polls the driver inside the timed region)
    with ClockSampler() if (rank == 0 and args.mode == "train" and args.warmup > 0) else contextlib.nullcontext() as clock:
        for _ in range(args.warmup):
            last = step()
        fence()
    warm = {k: v for k, v in timer.summary().items() if not k.startswith("ramnet_")}
    warm_hbm = {k: v for k, v in timer.summary().items() if k.startswith("ramnet_")}
    timer.rec, timer.hbm = [], False
    if warm:
        # dominant kernel = the symbol with the most executed MFMA work per step (what rocprofv3 --stats ranks first too); the
        # co-scheduled event durations of the warm-up inflate forward and backward launches differently and would make the
        # choice flip between the instantiations of the same kernel from run to run
        timer.only = max(warm.items(), key=lambda kv: kv[1][3])[0]
    collective = args.mode == "train" and (world > 1 or args.force_collective)
    if collective:
        reducer.timing = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0     # this rank's own time to finish its K steps (before the barrier)
    fence()
    dt = time.perf_counter() - t0
    timer.on = False
    agg_timed = timer.summary()          # dominant kernel over the timed region (the extras below reuse the timer)
    timer.rec = []
    per_rank = None
    if world > 1 or args.force_collective:
        t = torch.tensor([dt], device=model.gpu, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        # where a non-ideal scaling curve would come from: every rank's own step time and the span of its gradient collectives
        span = reducer.span_ms() if collective else None
        mine = torch.tensor([1e3 * dt_local / args.steps, span if span is not None else -1.0], device=model.gpu, dtype=torch.float64)
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        ms = [float(g[0]) for g in got]
        per_rank = {"step_ms": [round(v, 3) for v in ms], "step_ms_max": max(ms), "step_ms_min": min(ms),
                    "grad_allreduce_span_ms_last_step": [round(float(g[1]), 3) if float(g[1]) >= 0 else None for g in got],
                    "note": "step_ms = each rank's own wall time per step up to ITS device synchronisation, before the barrier; the "
                            "all-reduce span (side stream, first collective issued -> last done, last timed step) includes the wait of "
                            "early buckets for later folds"}
    loss_val = float(last.detach())
    assert np.isfinite(loss_val), "non-finite loss"

    # ---- extras (outside the timed region that defines `value`): the same step with every launch on ONE stream — per-kernel
    # durations without co-scheduled neighbours (dominant kernel, and the ConvGRU state update per scale).  2 steps, 1 warm-up.
    extras, gru_step, iso_hbm = {}, None, None
    if not args.no_extras and args.mode == "train" and world == 1 and (args.overlap_wgrad or args.overlap_decoder):
        ops.set_wgrad_overlap(False)
        ops.set_decoder_overlap(False)
        use_inp["on"] = False                 # (ONE stream: inputs resident, no input stream beside it)
        step()
        fence()
        timer.only, only = None, timer.only
        timer.on, timer.rec = not args.no_kernel_timing, []
        timer.hbm = timer.on
        t = time.perf_counter()
        for _ in range(2):
            lv = step()
        fence()
        e = (time.perf_counter() - t) / 2
        timer.on = timer.hbm = False
        extras["single_stream"] = {"value": B * L / e, "ms_per_step": 1e3 * e, "final_loss": float(lv.detach())}
        iso = timer.summary().get(only)
        iso_hbm = {k: v for k, v in timer.summary().items() if k.startswith("ramnet_")}
        gru_step = timer.summary(by_tag=True)
        timer.rec, timer.only = [], only
        if iso:
            extras["single_stream"]["dominant_kernel"] = {
                "kernel": only, "launches": iso[0], "avg_launch_ms": 1e3 * iso[1] / iso[0],
                "achieved": iso[3] / iso[1] / 1e12, "unit": "TFLOP/s (executed)", "frac": iso[3] / iso[1] / 1e12 / F32_MFMA_PEAK_TFLOPS,
                "algorithmic_achieved": iso[2] / iso[1] / 1e12}
        ops.set_wgrad_overlap(args.overlap_wgrad)
        ops.set_decoder_overlap(args.overlap_decoder)
        use_inp["on"] = want_input_side

    def measure(fn, n=2):
        fn()
        fence()
        t = time.perf_counter()
        for _ in range(n):
            lv = fn()
        fence()
        e = (time.perf_counter() - t) / n
        return {"value": B * L / e, "ms_per_step": 1e3 * e, "final_loss": float(lv.detach())}

    if not args.no_extras and world == 1:
        timer.on = False
        if "eager" in graphed:      # stream / infer: the timed region replays hipGraphs; the same work launch by launch
            extras["eager_launches"] = measure(graphed["eager"])
            if "graph_serial" in graphed:
                extras["graph_update_then_decode"] = dict(measure(graphed["graph_serial"]), note="hipGraph replays with the decode of "
                                                          "update k BEFORE update k+1 on one stream (the timed region overlaps them)")
            if "graph_per_package" in graphed:
                extras["graph_per_package"] = dict(measure(graphed["graph_per_package"]), note="one hipGraph replay per data package "
                                                   "(graph.GraphedPackage, the round-2 runtime)")
            if "graph_two_stage" in graphed:
                extras["graph_two_stage"] = dict(measure(graphed["graph_two_stage"]), note="decode of update k on a second stream beside "
                                                 "update k+1 (graph.GraphedStream pipelined, the round-2 runtime)")
        elif args.mode == "train" and not args.graph:
            if inp is not None:          # the definition of `value` up to round 4: the same step on tensors resident in HBM
                use_inp["on"] = False
                extras["resident_inputs"] = dict(measure(step, n=3), note="the step on input tensors already resident in HBM (no voxelisation, "
                                                 "no uploads in the loop): `value` of rounds 1-4; `value` now includes the input side")
                use_inp["on"] = True
            else:
                try:
                    extras["with_input_side"] = input_side_measure(model, step, seq, args, K, bins, B, L, H, W, rank)
                except Exception as ex:     # noqa: BLE001
                    extras["with_input_side"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            try:
                extras["stream_b1"] = stream_b1_measure(model, seq, K, H, W, timer)
            except Exception as ex:     # noqa: BLE001
                extras["stream_b1"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            try:
                extras["graph_replay"] = dict(measure(make_graph_step()), note="the same step as ONE hipGraph replay (zero-fill, forward, "
                                              "loss, BPTT backward, gradient fold) + eager Adam; rpg_ramnet_amd.graph.GraphedTrainStep")
            except Exception as ex:     # noqa: BLE001 — an extra must not take the headline measurement down
                extras["graph_replay"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            if (H, W, bins, B, L) == (256, 344, 5, 8, 8) and not args.full_frame and not args.no_configs4_extra:
                extras["split_operands"] = split_operands_measure()
                extras["full_frame_264x352"] = full_frame_measure()
                extras["configs4_shape"] = configs4_measure()

    stream_roof = None
    if args.mode in ("stream", "infer") and not args.no_kernel_timing and "eager" in graphed:
        timer.on, timer.only, timer.hbm, timer.rec = True, None, False, []
        graphed["eager"]()
        torch.cuda.synchronize()
        agg = timer.summary()
        timer.on, timer.rec = False, []
        salg, sex, snl = sum(v[2] for v in agg.values()), sum(v[3] for v in agg.values()), sum(v[0] for v in agg.values())
        per = dt / args.steps
        stream_roof = {"bound": "mfma", "kernel": "whole forward chain (%d MFMA launches per step)" % snl, "peak": F32_MFMA_PEAK_TFLOPS,
                       "unit": "TFLOP/s", "achieved": sex / per / 1e12, "frac": sex / per / 1e12 / F32_MFMA_PEAK_TFLOPS,
                       "frac_executed": sex / per / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_achieved": salg / per / 1e12,
                       "frac_algorithmic": salg / per / 1e12 / F32_MFMA_PEAK_TFLOPS,
                       "executed_tflop_per_step": sex / 1e12, "algorithmic_tflop_per_step": salg / 1e12, "traffic": None,
                       "note": "MFMA FLOP of every launch of one step (launch hooks over one eager pass) over the timed wall time per step "
                               "of the hipGraph replay; per-kernel HIP events do not exist inside a graph replay"}
    if rank == 0:
        samples = world * B * L * args.steps
        if args.mode == "stream":
            updates = world * B * (sum(sched) + L) * args.steps
        geom = ("346x260 full frame, reflect-padded to 264x352 in the input repack" if args.full_frame else
                "346x260 cropped to %dx%d" % (H, W) if (H, W) == (256, 344) else "%dx%d" % (W, H))
        out = {"metric": "depth samples/sec (%s, %d-bin grids, K=5 grids + 1 frame per sample; %s)"
                         % (geom, bins, {"train": "training step", "infer": "inference", "stream": "asynchronous streaming inference"}[args.mode]),
               "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if not args.split_operands else "f32 (3x3 forward / backward-data products as 3-term bf16 splits on the bf16 MFMA, fp32 accumulation)",
               "data": "synthetic",
               "abs_rel": "unverifiable: no checkpoint/dataset in the image (README.md:59-68 are URLs); the Abs-Rel formula and depth post-processing are "
                          "pinned against the reference's own outputs (tests/test_hip_ops.py::test_depth_metrics_*)",
               "input_side": ("in the timed loop: per step %d batched voxel scatter-adds of %d fresh on-device event lists (%d events each) + nonzero "
                              "normalisation %s, H2D of %d frames + %d target maps from pinned memory on a copy stream beside the "
                              "previous step's compute (bench.InputSide)" % (L, K * B, args.events_per_grid, "on the step's own stream in front of its forward pass"
                                                                             if args.input_inline else "on a low-priority input stream beside the compute", L * B, 2 * L * B)) if (inp is not None) else
                             "resident: the step's input tensors are in HBM before the timed region",
               "config": {"workload": "EventScape-shaped %s, %d event bins, K=5, batch %d/GPU, seq-len %d, "
                                      "1xMI355X-per-rank %s, state=%s, SI loss on [image,events4], Adam, backward-weights %s"
                                      % (("346x260 -> 256x344 crop" if (H, W) == (256, 344) else geom), bins, B, L, args.mode, args.state, "co-scheduled on a side stream" if args.overlap_wgrad else "on the main stream") +
                                      (", decoders on a second stream" if args.overlap_decoder and args.mode == "train" else ""),
                          "global_batch": B * world, "seq_len": L, "parallelism": "dp%d" % world},
               "final_loss": loss_val, "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
               "rccl_ranks_seen": ranks_seen, "backend": ((backend_note or args.backend) if (world > 1 or args.force_collective) else None),
               "loss_semantics": LOSS_SEMANTICS[bool(args.dp_exact_loss and args.mode == "train")]}
        if args.mode == "train" and (world > 1 or args.force_collective):
            out["grad_allreduce"] = {"buckets": len(reducer.buckets), "mb": reducer.flat.numel() * 4 / 1e6,
                                     "buckets_issued_during_the_fold": reducer.early_buckets, "steps_run": args.steps + args.warmup,
                                     "note": "parallel.FlatGradReducer: bucket k goes to the side stream as soon as the end-of-backward fold "
                                             "has finalised the last parameter of buckets 0..k (SURVEY 8e); the rest in all_reduce()"}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if args.mode == "stream":
            out["stream"] = {"updates_per_s": updates / dt, "ms_per_update_and_decode": 1e3 * dt / (updates / (world * B)),
                             "grids_per_frame": sched, "note": "one update = fold one event grid or frame into the persistent "
                             "state + decode one depth map; samples/s counts frames"}
        if extras:
            out["extras"] = extras
            sp = extras.get("split_operands")
            if isinstance(sp, dict) and sp.get("value"):     # the optional second arithmetic, beside — never instead of — the exact-fp32 `value`
                out["split_operands"] = {"value": sp["value"], "unit": sp["unit"], "ms_per_step": sp["ms_per_step"], "vs_exact_fp32": sp["value"] / (samples / dt),
                                         "final_loss": sp.get("final_loss"),
                                         "evidence": "parity suites at unchanged tolerances incl. the whole GPU suite with the variant forced on (profiles/"
                                                     "r06_gputest_split_everywhere.log), error vs float64 beside the exact kernel's (profiles/r06_split_design.md)"}
        if stream_roof:
            out["roofline"] = stream_roof
            if args.mode == "stream":
                n_upd = (sum(sched) + L) * B
                out["roofline"]["algorithmic_gflop_per_update"] = stream_roof["algorithmic_tflop_per_step"] * 1e3 / n_upd
                out["roofline"]["executed_gflop_per_update"] = stream_roof["executed_tflop_per_step"] * 1e3 / n_upd
        if agg_timed:
            name, (n, secs, alg, ex, opb) = max(agg_timed.items(), key=lambda kv: kv[1][1])
            pmc_file, pmc = newest_pmc_traffic()
            traffic = traffic_of(pmc, name)
            # a summary recorded for another build (different launches per step of the kernels it is quoted for) is refused, not quoted
            pmc_stale = []
            if warm and args.warmup > 0:
                for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])[:3]:
                    have, want = pmc_launches_of(pmc, k), v[0] / args.warmup
                    if have and abs(have - want) > 0.01 * want:
                        pmc_stale.append("%s: %d launches per step in %s, %d now" % (k, have, pmc_file, want))
            if pmc_stale:
                traffic = None
            out["roofline"] = {
                "bound": "mfma", "kernel": name, "achieved": ex / secs / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ex / secs / 1e12 / F32_MFMA_PEAK_TFLOPS, "frac_executed": ex / secs / 1e12 / F32_MFMA_PEAK_TFLOPS,
                "frac_algorithmic": alg / secs / 1e12 / F32_MFMA_PEAK_TFLOPS,
                "clock_ghz": clock.ghz() if clock is not None else None, "clock_source": "rocm-smi --showclocks sclk, mean over the warm-up steps "
                "(None: tool absent); peak = spec clock 2.4 GHz",
                "traffic": traffic, "traffic_fetch_size_x2": None if pmc_stale else traffic_of(pmc, name, fetch_x2=True),
                "traffic_source": "profiles/" + pmc_file,
                "operand_bytes_per_launch": opb / n, "traffic_over_operand_bytes": (traffic / (opb / n)) if traffic and opb else None,
                "launches": n, "avg_launch_ms": 1e3 * secs / n,
                "executed_gflop_per_launch": ex / n / 1e9, "algorithmic_gflop_per_launch": alg / n / 1e9,
                "algorithmic_achieved": alg / secs / 1e12,
                "note": "frac_algorithmic = SURVEY 8d quantity (algorithmic FLOP per launch / average launch duration / peak; can exceed the "
                        "executed fraction by the Winograd factor 72/24 = 3 of F(2x4,3x3)); frac = frac_executed = pipe utilisation.  "
                        "achieved/frac = EXECUTED MFMA FLOP (F(2x4,3x3): 24/72 of the 3x3 layer's, F(2x2,3x3) 16/36, folded decoders 25/64) over "
                        "HIP-event durations of this kernel in the timed region, which co-schedules three streams (main, decoders, "
                        "backward-weights) — wall durations include time shared with other kernels; extras.single_stream.dominant_kernel "
                        "= same launches on one stream.  algorithmic_achieved = layer-level rate (SURVEY 8d count).  traffic = HBM bytes "
                        "per launch, TCC_EA_RDREQ x 64 B + WRITE_SIZE (= FETCH_SIZE + WRITE_SIZE as counted), launch-weighted over the kernel's grids; traffic_fetch_size_x2 "
                        "= 2*FETCH_SIZE + WRITE_SIZE, the guide's correction for 128-byte streaming reads, which these 32-byte-per-pixel patch loads are not (L2 hit rate "
                        "90 %, nothing is fetched twice: profiles/r05_zx_pmc_l2.txt); operand_bytes_per_launch = inputs, masks and epilogue operands read once + outputs "
                        "written once + weights once (backward-weights: input, gradient and their masks read once + the gradient workspace read and written "
                        "once; its per-split slabs repeat that last term per split), averaged over the same launches"}
            if pmc_stale:
                out["roofline"]["traffic_refused"] = pmc_stale
            if warm and args.warmup > 0:      # the three kernels with the most GPU time: what they have to move against what the counters saw
                top = []
                for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])[:3]:
                    tr = None if pmc_stale else traffic_of(pmc, k)
                    top.append({"kernel": k, "launches_per_step": v[0] / args.warmup, "avg_launch_ms": 1e3 * v[1] / v[0],
                                "mfma_frac_executed": v[3] / v[1] / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                "operand_bytes_per_launch": v[4] / v[0], "traffic": tr,
                                "traffic_over_operand_bytes": (tr / (v[4] / v[0])) if (tr and v[4]) else None})
                out["roofline"]["top3"] = top
            iso_k = extras.get("single_stream", {}).get("dominant_kernel") if extras else None
            if iso_k:       # the same launches with nothing else on the chip (one-stream pass of the same process)
                out["roofline"]["one_stream"] = {"avg_launch_ms": iso_k["avg_launch_ms"], "frac_executed": iso_k["frac"],
                                                 "frac_algorithmic": iso_k["algorithmic_achieved"] / F32_MFMA_PEAK_TFLOPS, "launches": iso_k["launches"]}
            if warm and args.warmup > 0:      # overlap-proof view: all MFMA FLOP of a step over the step's wall time
                step_alg = sum(v[2] for v in warm.values()) / args.warmup
                step_ex = sum(v[3] for v in warm.values()) / args.warmup
                out["step_mfma"] = {"executed_tflop_per_step": step_ex / 1e12, "algorithmic_tflop_per_step": step_alg / 1e12,
                                    "achieved": step_ex / (dt / args.steps) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": step_ex / (dt / args.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                    "algorithmic_achieved": step_alg / (dt / args.steps) / 1e12}
                out["kernels_warmup_steps"] = {
                    k: {"launches": v[0], "ms": 1e3 * v[1], "executed_tflops": v[3] / v[1] / 1e12,
                        "mfma_frac": v[3] / v[1] / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_tflops": v[2] / v[1] / 1e12,
                        "hbm_bytes_per_launch": traffic_of(pmc, k)} for k, v in warm.items()}
                src_hbm = iso_hbm if iso_hbm else warm_hbm       # one-stream pass when available: durations without co-scheduled kernels
                hb = {k[len("ramnet_"):]: {"launches": v[0], "avg_us": 1e6 * v[1] / v[0], "algorithmic_mb_per_launch": v[4] / v[0] / 1e6,
                                           "achieved_tb_s": v[4] / v[1] / 1e12, "hbm_frac": v[4] / v[1] / 1e12 / HBM_PEAK_TBS}
                      for k, v in src_hbm.items() if v[4] > 0}
                if vox:     # input side: B*K grids per launch, 48 B per event (32 B read + two fp32 atomic RMW) + the grid zero-fill
                    nb = L * (48.0 * args.events_per_grid * B * K + 4.0 * B * K * bins * H * W)
                    hb["voxelize_batch"] = {"launches": vox[0], "avg_us": 1e6 * vox[1] / vox[0], "algorithmic_mb_per_launch": nb / vox[0] / 1e6,
                                            "achieved_tb_s": nb / vox[1] / 1e12, "hbm_frac": nb / vox[1] / 1e12 / HBM_PEAK_TBS,
                                            "note": "one launch = the %d grids of a package-batch (%d events each); outside the timed step" % (B * K, args.events_per_grid)}
                out["hbm_kernels"] = dict(hb, peak_tb_s=HBM_PEAK_TBS, note="HBM-bound kernels against the 8 TB/s HBM spec peak (6.3 TB/s "
                                          "achievable, MI355X_MICROARCH.md); " + ("one-stream pass (extras.single_stream)" if iso_hbm else
                                          "warm-up steps on the three-stream schedule: durations include co-scheduled time") +
                                          "; si_loss_* move 5-8 MB per launch and are launch-latency bound")
            if gru_step:
                # ConvGRU step algorithmic bytes = x read + h read + h' write (+ weights once); FLOP = 3 convs 3x3 2C->C
                gs = {}
                for tag, v in sorted(gru_step.items()):
                    C = int(tag.split("C")[-1])
                    div = C // 32
                    npx = B * (H // div) * (W // div)
                    nbytes = 3.0 * 4 * C * npx + 4.0 * 27 * 2 * C * C
                    steps_n = v[0] / 2.0                       # two launches per state update
                    ms = 1e3 * v[1] / steps_n
                    gs["C%d" % C] = {"ms": ms, "algorithmic_bytes": nbytes, "hbm_frac": nbytes / (ms * 1e-3) / 1e12 / HBM_PEAK_TBS,
                                     "mfma_frac": v[3] / v[1] / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                     "algorithmic_tflops": v[2] / v[1] / 1e12, "launches_per_step": v[0] / steps_n}
                out["roofline"]["gru_step"] = dict(gs, note="forward ConvGRU state update per scale (submodules.py:436-454), single-stream "
                                                   "pass: the step is MFMA-bound (530-810 FLOP/B vs a 20 FLOP/B ridge), so its HBM fraction "
                                                   "is low BECAUSE it is compute-bound (SURVEY 8d); mfma_frac = executed-MFMA fraction")
        if not args.no_cpu_baseline and world == 1:
            # (full-frame mode: the oracle, like the reference network, only takes multiples of 8 — it is timed at the padded size)
            out["cpu_baseline"] = cpu_baseline(cfg, 264 if args.full_frame else H, 352 if args.full_frame else W, K, args)
            if args.cpu_infer_baseline and args.mode == "train":
                out["cpu_baseline_infer"] = cpu_baseline(cfg, H, W, K, args, mode="infer")
        print(json.dumps(out))
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
