"""Event list -> voxel grid on device (HIP scatter-add), drop-in for
RAM_Net/utils/event_tensor_utils.py:120-187 (`events_to_voxel_grid_pytorch`) and the nonzero
normalisation of `EventPreprocessor.__call__` (:52-66).  Unlike the reference, the caller's event array is
NOT modified in place."""
import ctypes as C

import numpy as np
import torch

from . import _hip as H
from .ops import _p, _st


def _events(events, device):
    if isinstance(events, np.ndarray):
        events = torch.from_numpy(np.ascontiguousarray(events, dtype=np.float64))
    assert events.dim() == 2 and events.shape[1] == 4
    return events.to(device=device, dtype=torch.float64).contiguous()


def events_to_voxel_grid(events, num_bins, width, height, device=None):
    """events: [N,4] (t, x, y, p) sorted by t (numpy or tensor).  Returns float32 [num_bins, height, width]."""
    assert num_bins > 0 and width > 0 and height > 0
    device = torch.device(device) if device is not None else (events.device if torch.is_tensor(events) else torch.device("cuda:0"))
    ev = _events(events, device)
    grid = torch.empty(num_bins, int(height), int(width), device=device, dtype=torch.float32)
    H.check(H.lib().ramnet_voxelize(_p(ev) if ev.numel() else None, ev.shape[0], num_bins, int(width), int(height),
                                    _p(grid), _st()), "ramnet_voxelize")
    return grid


def voxel_indices(events, num_bins, width, height):
    """Flat int64 indices of the left/right votes (-1 where the reference's validity mask drops the vote)."""
    ev = _events(events, events.device)
    il = torch.empty(ev.shape[0], device=ev.device, dtype=torch.int64)
    ir = torch.empty_like(il)
    H.check(H.lib().ramnet_voxel_indices(_p(ev), ev.shape[0], num_bins, int(width), int(height), _p(il), _p(ir), _st()),
            "ramnet_voxel_indices")
    return il, ir


def normalize_nonzero(grid):
    """Zero mean / unit std over the non-zero entries; zeros stay zero.  Returns a new tensor."""
    out = grid.detach().clone().float().contiguous()
    scratch = torch.empty(3, device=out.device, dtype=torch.float64)
    H.check(H.lib().ramnet_normalize_nonzero(_p(out), out.numel(), _p(scratch), _st()), "ramnet_normalize_nonzero")
    return out
