#!/usr/bin/env python3
"""Where does the gradient deviation of a full-resolution training step come from?  One BPTT step of BASELINE configs[1] at
B=2, L=2 (256x344, K=5) through (a) the HIP path with the Winograd kernels, (b) the HIP path with the direct exact-fp32 kernels,
(c) the oracle in float32 — each against the oracle in float64.  Prints, per variant, the worst per-tensor error (relative to
the tensor's largest entry, floored at 1 % of the largest gradient in the model) and the median over tensors."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from oracle import ramnet_ref  # noqa: E402  (test infrastructure; this tool is a diagnostic, not product code)
from recipe import make_item  # noqa: E402
from util import build_hip_model, ref_cfg  # noqa: E402
from rpg_ramnet_amd import ops  # noqa: E402
from rpg_ramnet_amd.trainer import sequence_loss  # noqa: E402

cfg, _ = ref_cfg("net_seeded_ramnet.npz", every_x_rgb_frame=5, loss_composition=["image", "events4"])
B, H, W, L = 2, 256, 344, 2
rng = np.random.default_rng(11)
seq = [make_item(rng, B, H, W, 5, 5, 1, True, 0.0) for _ in range(L)]
model = build_hip_model("ERGB2DepthRecurrent", cfg).train()
lc = cfg["loss_composition"]


def hip():
    model.zero_grad()
    total, _ = sequence_loss(model, seq, lc, [1, 1])
    total.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}


def oracle(dtype):
    sd = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in model.state_dict().items()}
    t, _ = ramnet_ref.sequence_loss(sd, cfg, [{k: v.to(dtype) for k, v in it.items()} for it in seq], lc, [1, 1])
    t.backward()
    return {k: v.grad.double() for k, v in sd.items()}


ref = oracle(torch.float64)
gmax = max(float(v.abs().max()) for v in ref.values())
runs = {"hip winograd": hip()}
ops.set_winograd(False), ops.set_fold_winograd(False), ops.set_fold_winograd_wgrad(False), ops.set_fold_dgrad(False), ops.set_space_to_depth(False)
runs["hip direct fp32"] = hip()
runs["oracle fp32"] = oracle(torch.float32)
for name, g in runs.items():
    errs = sorted(((float((g[k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), 1e-2 * gmax), k) for k in ref), reverse=True)
    print("%-16s worst %.2e (%s)  median %.2e" % (name, errs[0][0], errs[0][1].split("recurrent.")[-1], errs[len(errs) // 2][0]))
