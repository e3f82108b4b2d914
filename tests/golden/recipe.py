"""Seeded synthetic-input recipes shared by make_golden.py (which feeds them to the reference) and by the
tests (which feed the very same arrays to the oracle / the HIP path).  numpy Generator streams are
bit-reproducible across machines, so large inputs need not be stored in the fixtures."""
import numpy as np
import torch


def make_item(rng, B, H, W, K, c_ev, c_img, with_depth=False, nan_frac=0.0):
    item = {}
    for k in range(K):
        item["events%d" % k] = torch.from_numpy(rng.standard_normal((B, c_ev, H, W)).astype(np.float32))
    item["image"] = torch.from_numpy(rng.random((B, c_img, H, W)).astype(np.float32))
    if with_depth:
        for key in ["image"] + ["events%d" % k for k in range(K)]:
            d = np.clip(1.0 + np.log(rng.uniform(0.02, 1.0, (B, 1, H, W))) / 3.70378, 0, 1).astype(np.float32)
            if nan_frac > 0:
                d[rng.random(d.shape) < nan_frac] = np.nan
            item["depth_" + key] = torch.from_numpy(d)
    return item



def synth_events(rng, n, W, H, t0=0.0, t1=0.05):
    ev = np.empty((n, 4), np.float64)
    ev[:, 0] = np.sort(rng.uniform(t0, t1, n))
    ev[:, 1] = rng.integers(0, W, n)
    ev[:, 2] = rng.integers(0, H, n)
    ev[:, 3] = rng.integers(0, 2, n)
    return ev


