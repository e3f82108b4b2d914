"""Depth evaluation on device: metric depth from normalised log depth (RAM_Net/evaluation.py:74-96) and the error metrics
of `add_to_metrics` (evaluation.py:201-241, RAM_Net/model/metric.py:8-33) with the reference's key names, comparison
operators and epsilons; one fused HIP reduction per call (Abs-Rel is the accuracy half of the headline metric).

Deviations, stated: `median_diff` is not computed (not a sum); a NaN target makes the reference's `RMS_log` NaN (plain
`np.mean`) while it is skipped here like in every other metric (the simulation data the paper's tables use has no NaNs)."""
import math

import torch

from . import _hip as H
from .ops import _p, _st


def depth_metrics(prediction, target, clip_distance, reg_factor, cutoff=float("inf")):
    """prediction / target: normalised log-depth tensors (any shape, same numel).  Pixels with metric target depth
    `nan_to_num(t) < cutoff` form the mask (evaluation.py:367); NaN targets are skipped by the means (metric.py) but stay in
    the denominator of the delta thresholds (`np.mean(ratio <= ...)`, evaluation.py:224-226)."""
    p = prediction.detach().float().contiguous()
    t = target.detach().to(p.device).float().contiguous()
    assert p.numel() == t.numel()
    out = torch.empty(11, device=p.device, dtype=torch.float64)
    H.check(H.lib().ramnet_depth_metrics(_p(p), _p(t), p.numel(), float(clip_distance), float(reg_factor),
                                         float(min(cutoff, 3.0e38)), _p(out), _st()), "ramnet_depth_metrics")
    n, nmask, ar, sr, se, l2, l1, ae, d1, d2, d3 = out.cpu().tolist()
    if n == 0:
        return {"n": 0}
    return {"n": int(n), "abs_rel_diff": ar / n, "squ_rel_diff": sr / n, "RMS_linear": math.sqrt(se / n),
            "RMS_log": math.sqrt(l2 / n), "SILog": l2 / n - (l1 / n) ** 2, "mean_depth_error": ae / n,
            "threshold_delta_1.25": d1 / nmask, "threshold_delta_1.25^2": d2 / nmask, "threshold_delta_1.25^3": d3 / nmask}
