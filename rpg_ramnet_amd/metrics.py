"""Depth evaluation on device: metric depth from normalised log depth (RAM_Net/evaluation.py:74-96) and the error metrics
of `add_to_metrics` (evaluation.py:201-241, RAM_Net/model/metric.py:8-33) with the reference's key names, comparison
operators and epsilons; one fused HIP reduction per call (Abs-Rel is the accuracy half of the headline metric).  All ten rows of the
reference's table, NaN behaviour included: `RMS_log` and `median_diff` are plain `np.mean` / `np.median` over the masked pixels
(evaluation.py:214, 241), so ONE NaN target inside the mask makes them NaN, while the metric.py means skip it
(pinned by tests/golden/eval_metrics.npz, the reference's own `add_to_metrics` run on seeded maps)."""
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
    clean = n == nmask                        # no NaN target inside the mask
    return {"n": int(n), "abs_rel_diff": ar / n, "squ_rel_diff": sr / n, "RMS_linear": math.sqrt(se / n),
            "RMS_log": math.sqrt(l2 / n) if clean else float("nan"), "SILog": l2 / n - (l1 / n) ** 2, "mean_depth_error": ae / n,
            "median_diff": _median_diff(p, t, float(clip_distance), float(reg_factor), float(cutoff)) if clean else float("nan"),
            "threshold_delta_1.25": d1 / nmask, "threshold_delta_1.25^2": d2 / nmask, "threshold_delta_1.25^3": d3 / nmask}


def _median_diff(p, t, clip, reg, cutoff):
    """|median(target) - median(prediction)| of the masked metric depths (evaluation.py:241; np.median: the mean of the two middle
    values of an even count).  Called for NaN-free masks only; two device sorts — the table is filled once per evaluated frame."""
    tm = torch.exp(reg * (t.reshape(-1) - 1.0)) * clip
    pm = (torch.exp(reg * (p.reshape(-1) - 1.0)) * clip).clamp(math.exp(-reg) * clip, clip)
    keep = tm < cutoff
    med = []
    for v in (tm[keep], pm[keep]):
        s, k = torch.sort(v)[0], v.numel()
        med.append(0.5 * (s[(k - 1) // 2] + s[k // 2]))
    return float((med[0] - med[1]).abs())
