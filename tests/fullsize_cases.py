"""Inputs of the full-resolution runs of tests/test_hip_fullsize.py, shared by the tests (GPU box) and by
tests/golden/make_golden_fullsize.py, which evaluates the float64 / float32 CPU oracle on them ONCE and commits what it returns as
tests/golden/fullsize.npz — oracle outputs on seeded inputs, like every other fixture: the oracle's 6 minutes of host time per run of
the suite are spent when the fixture is made, not on the GPU box.  Everything is seeded: numpy generators for the
inputs, torch.manual_seed(0) for the weights (CPU generator: the same numbers on every machine of this image)."""
import zlib

import numpy as np
import torch

from recipe import make_item

H, W = 256, 344

# tag -> (fixture with the config, config overrides, batch, height, width, packages, NaN fraction of the targets)
STEP_CASES = {
    "config1_B2_L2": ("net_seeded_ramnet.npz", dict(every_x_rgb_frame=5, loss_composition=["image", "events4"]), 2, H, W, 2, 0.0),
    "bench_B8_L1": ("net_seeded_ramnet.npz", dict(every_x_rgb_frame=5, loss_composition=["image", "events4"]), 8, H, W, 1, 0.0),
    "bench_B8_L8": ("net_seeded_ramnet.npz", dict(every_x_rgb_frame=5, loss_composition=["image", "events4"]), 8, H, W, 8, 0.0),
    "config4_B1_L2": ("net_seeded_ramnet_bins10.npz", dict(every_x_rgb_frame=2, loss_composition=["image", "events1"]), 1, 480, 640, 2, 0.2),
}
N_GRAD, N_MAP = 1024, 8192            # sampled entries per gradient tensor / per prediction or state map


def step_sequence(cfg, Bn, Hn, Wn, L, nan_frac):
    rng = np.random.default_rng(11)
    K = cfg["every_x_rgb_frame"]
    return [make_item(rng, Bn, Hn, Wn, K, cfg["num_bins_events"], cfg["num_bins_rgb"], True, nan_frac) for _ in range(L)]


def sample_idx(name, numel, n):
    """The entries of tensor `name` the fixture holds: all of them up to n, else n distinct seeded positions (sorted)."""
    if numel <= n:
        return np.arange(numel)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(rng.choice(numel, n, replace=False))


def long_horizon_items(K=5, L=8):
    rng = np.random.default_rng(21)
    return [make_item(rng, 1, H, W, K, 5, 1) for _ in range(L)]


def stream_200_schedule():
    """The 200-update irregular stream: ("events" | "rgb", tensor) measurements and ("check", tag) marks, in order."""
    rng = np.random.default_rng(33)
    n, steps = 0, []
    while n < 200:
        for _ in range(int(rng.integers(1, 9))):
            steps.append(("events", torch.from_numpy(rng.standard_normal((1, 5, H, W)).astype(np.float32))))
            n += 1
        steps.append(("rgb", torch.from_numpy(rng.random((1, 1, H, W)).astype(np.float32))))
        n += 1
        if n % 40 < 9:
            steps.append(("check", "after %d updates" % n))
    steps.append(("check", "after %d updates (end)" % n))
    return steps
