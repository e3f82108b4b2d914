#!/usr/bin/env python3
"""tests/golden/eval_metrics.npz: the table `add_to_metrics` of the REFERENCE's evaluation.py (:201-241) fills — all ten metrics incl.
`median_diff` and the NaN behaviour of `RMS_log` — on seeded metric-depth maps, with and without NaN targets and distance cutoffs
(evaluation.py:365-378 masks `np.nan_to_num(target) < cutoff`).  Imports the reference (build container only) and calls its functions:
prepare_depth_data, then add_to_metrics(idx, {}, target, prediction, mask)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    _, _, _, _, _, _, ev = import_reference()
    rng = np.random.default_rng(23)
    out = {}
    cases = [("plain", 0.0, np.inf, 80.0, 3.70378), ("cut30", 0.0, 30.0, 80.0, 3.70378), ("nan", 0.15, np.inf, 80.0, 3.70378),
             ("nan_cut20", 0.3, 20.0, 80.0, 3.70378), ("mvsec", 0.0, 10.0, 1000.0, 5.70378)]
    for tag, nan_frac, cutoff, clip, reg in cases:
        t_in = rng.random((36, 50)).astype(np.float32)
        p_in = np.clip(t_in + 0.08 * rng.standard_normal(t_in.shape), 0, 1).astype(np.float32)
        if nan_frac:
            t_in[rng.random(t_in.shape) < nan_frac] = np.nan
        t, p = ev.prepare_depth_data(t_in.copy(), p_in.copy(), clip, reg_factor=reg)
        mask = np.nan_to_num(t) < cutoff
        with np.errstate(all="ignore"):
            m = ev.add_to_metrics(0, {}, t, p, mask, prefix="_")
        out["%s.target_in" % tag], out["%s.pred_in" % tag] = t_in, p_in
        out["%s.params" % tag] = np.array([clip, reg, cutoff if np.isfinite(cutoff) else 3.0e38])
        for k in ("abs_rel_diff", "squ_rel_diff", "RMS_linear", "RMS_log", "SILog", "mean_depth_error", "median_diff", "threshold_delta_1.25",
                  "threshold_delta_1.25^2", "threshold_delta_1.25^3"):
            out["%s.%s" % (tag, k)] = np.float64(m["_" + k])
        print(tag, {k: float(out["%s.%s" % (tag, k)]) for k in ("abs_rel_diff", "RMS_log", "median_diff")})
    np.savez_compressed(os.path.join(HERE, "eval_metrics.npz"), **out)
    print("eval_metrics.npz")


if __name__ == "__main__":
    main()
