"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny): the 'relative fp32' error the north star bounds by 1e-3."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close(a, b, tol, what="", floor=0.0):
    """floor: absolute scale below which a reference magnitude is treated as cancellation noise (e.g. the SI-loss
    gradient of a bias is a sum that cancels to ~0); pass a fraction of the largest gradient in the model."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    e = float(np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-30))
    assert e <= tol, "%s: rel err %.3e > %.1e (max|ref| %.3e)" % (what, e, tol, float(np.abs(np.asarray(b)).max()))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def ref_cfg(name, **over):
    """Model config the way train.py:198-201 assembles it, from the fixture (no /root/reference access)."""
    z = load_golden(name)
    cfg = json.loads(str(z["config"]))
    cfg.update(over)
    return cfg, z


def build_hip_model(arch, cfg, seed=0):
    from rpg_ramnet_amd.model import model as mm
    torch.manual_seed(seed)
    m = getattr(mm, arch)(cfg)
    return m.to(m.gpu)
