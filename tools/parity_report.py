#!/usr/bin/env python3
"""Max relative error (max|got-ref| / max|ref|) of the HIP path against the reference's golden predictions and states,
for the direct and the Winograd kernels.  Usage (GPU box): python tools/parity_report.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from recipe import make_item  # noqa: E402
from util import build_hip_model, ref_cfg  # noqa: E402
from oracle import ramnet_ref  # noqa: E402
from rpg_ramnet_amd import ops  # noqa: E402


def worst(tag):
    cfg, z = ref_cfg("net_%s.npz" % tag)
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
    seed, B, H, W, n_ev, c_ev, c_img, calls = [int(v) for v in z["recipe"]]
    rng = np.random.default_rng(seed)
    prev_super, prev_lstm = None, ramnet_ref.empty_states_lstm(cfg["every_x_rgb_frame"])
    w = {"pred": 0.0, "state": 0.0}
    with torch.no_grad():
        for c in range(calls):
            preds, supers, lstms = model(make_item(rng, B, H, W, n_ev, c_ev, c_img), prev_super, prev_lstm)
            for k, v in preds.items():
                r = z["pred%d.%s" % (c, k)]
                w["pred"] = max(w["pred"], float(np.abs(v.cpu().numpy() - r).max() / np.abs(r).max()))
            for name in [f for f in z.files if f.startswith("super%d.image." % c)]:
                parts = name.split(".")
                s = supers["image"][int(parts[2])]
                s = s[{"h": 0, "c": 1}[parts[3]]] if len(parts) == 4 else s
                w["state"] = max(w["state"], float(np.abs(s.cpu().numpy() - z[name]).max() / np.abs(z[name]).max()))
            prev_super, prev_lstm = supers["image"], lstms
    return w


for tag in ("seeded_ramnet", "seeded_ramnet_lstm", "config1_256"):
    for on in (False, True):
        ops.set_winograd(on)
        r = worst(tag)
        print("%-20s %-8s predictions %.2e   states %.2e   (bar: 1e-3)" % (tag, "winograd" if on else "direct", r["pred"], r["state"]))
