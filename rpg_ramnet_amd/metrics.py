"""Depth evaluation on device: metric depth from normalised log depth and the error metrics of
RAM_Net/evaluation.py:74-96, :201-241 / RAM_Net/model/metric.py:8-33 (Abs-Rel is the accuracy half of the headline
metric), one fused HIP reduction per call."""
import math

import torch

from . import _hip as H
from .ops import _p, _st


def depth_metrics(prediction, target, clip_distance, reg_factor, cutoff=float("inf")):
    """prediction / target: normalised log-depth tensors (any shape, same numel; NaN targets are skipped).
    Returns a dict with the reference's metric names (means over valid pixels with metric depth <= cutoff)."""
    p = prediction.detach().float().contiguous()
    t = target.detach().to(p.device).float().contiguous()
    assert p.numel() == t.numel()
    out = torch.empty(10, device=p.device, dtype=torch.float64)
    H.check(H.lib().ramnet_depth_metrics(_p(p), _p(t), p.numel(), float(clip_distance), float(reg_factor),
                                         float(min(cutoff, 3.0e38)), _p(out), _st()), "ramnet_depth_metrics")
    n, ar, sr, se, l2, l1, ae, d1, d2, d3 = out.cpu().tolist()
    if n == 0:
        return {"n": 0}
    return {"n": int(n), "abs_rel_diff": ar / n, "squ_rel_diff": sr / n, "rms_linear": math.sqrt(se / n),
            "scale_invariant_error": l2 / n - (l1 / n) ** 2, "mean_error": ae / n,
            "threshold_1.25": d1 / n, "threshold_1.25^2": d2 / n, "threshold_1.25^3": d3 / n}
