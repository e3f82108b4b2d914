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


def pack_event_lists(event_lists, device):
    """A batch of event lists -> (one [sum N, 4] float64 device tensor, int64 offsets [G + 1] on the device, the longest list): the form a
    loader uploads once and events_to_voxel_grids_packed() consumes without a concatenation copy per call."""
    device = torch.device(device)
    evs = [_events(e, device) for e in event_lists]
    counts = [int(e.shape[0]) for e in evs]
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int64).to(device)
    cat = torch.cat(evs) if sum(counts) else None
    return cat, off, max(counts)


def events_to_voxel_grids_packed(cat, off, max_count, num_bins, width, height, out=None, normalize=False, scratch=None):
    """events_to_voxel_grids() on lists that are already packed (pack_event_lists): ONE scatter-add launch + one batched nonzero
    normalisation on the current stream, no allocation when `out` ([G, num_bins, height, width] fp32) and `scratch` (3 G doubles) are given
    — the per-step form of an input pipeline that prepares the next step's grids on a side stream."""
    G, n = int(off.shape[0]) - 1, num_bins * int(height) * int(width)
    device = off.device
    grids = out if out is not None else torch.empty(G, num_bins, int(height), int(width), device=device, dtype=torch.float32)
    assert tuple(grids.shape) == (G, num_bins, int(height), int(width)) and grids.is_contiguous() and grids.dtype == torch.float32
    L = H.lib()
    H.check(L.ramnet_voxelize_batch(_p(cat), _p(off), G, int(max_count), num_bins, int(width), int(height), _p(grids), _st()),
            "ramnet_voxelize_batch")
    if normalize:
        assert n % 4 == 0, "batched normalisation: bins * height * width must be a multiple of 4"
        if scratch is None:
            scratch = torch.empty(3 * G, device=device, dtype=torch.float64)
        H.check(L.ramnet_normalize_nonzero_batch(_p(grids), G, n, _p(scratch), _st()), "ramnet_normalize_nonzero_batch")
    return grids


def events_to_voxel_grids(event_lists, num_bins, width, height, device=None, normalize=False):
    """A batch of event lists -> [G, num_bins, height, width] in ONE scatter-add launch (+ one batched nonzero normalisation):
    the B x K grids of a batch of data packages.  Per grid identical to events_to_voxel_grid / normalize_nonzero."""
    assert len(event_lists) > 0 and num_bins > 0 and width > 0 and height > 0
    if device is None:
        device = event_lists[0].device if torch.is_tensor(event_lists[0]) else torch.device("cuda:0")
    device = torch.device(device)
    evs = [_events(e, device) for e in event_lists]
    counts = [int(e.shape[0]) for e in evs]
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int64).to(device)
    cat = torch.cat(evs) if sum(counts) else None
    G, n = len(evs), num_bins * int(height) * int(width)
    grids = torch.empty(G, num_bins, int(height), int(width), device=device, dtype=torch.float32)
    L = H.lib()
    H.check(L.ramnet_voxelize_batch(_p(cat), _p(off), G, max(counts), num_bins, int(width), int(height), _p(grids), _st()),
            "ramnet_voxelize_batch")
    if normalize:
        if n % 4:
            return torch.stack([normalize_nonzero(g) for g in grids])
        scratch = torch.empty(3 * G, device=device, dtype=torch.float64)
        H.check(L.ramnet_normalize_nonzero_batch(_p(grids), G, n, _p(scratch), _st()), "ramnet_normalize_nonzero_batch")
    return grids
