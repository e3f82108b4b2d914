#!/usr/bin/env python3
"""the REFERENCE's own flat voxel indices for the event lists of voxel.npz.

`events_to_voxel_grid_pytorch` (utils/event_tensor_utils.py:120-187) never returns its indices; it hands them to two
`Tensor.index_add_` calls (left vote :170-173, right vote :178-181).  This script imports the reference (build container only), runs
that function on the events stored in voxel.npz with `index_add_` wrapped by a recorder, and writes what the reference passed —
the int64 index arrays of the VALID votes, in event order — to voxel_indices.npz.  Nothing of the reference is copied: its function is
called.  The GPU test (tests/test_hip_ops.py::test_voxel_grid_golden) asserts that ramnet_voxel_indices, with its invalid votes
(-1) dropped, equals these arrays bit for bit."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    _, _, _, _, etu, _, _ = import_reference()
    etu.CudaTimer = etu.Timer
    z = np.load(os.path.join(HERE, "voxel.npz"))
    names = sorted({k.split(".")[0] for k in z.files if k.endswith(".events")})
    out = {}
    real = torch.Tensor.index_add_
    for name in names:
        ev = z["%s.events" % name]
        bins, W, H = [int(v) for v in z["%s.dims" % name]]
        seen = []

        def spy(self, dim, index, source, *a, **kw):
            seen.append(index.detach().clone().numpy().astype(np.int64))
            return real(self, dim, index, source, *a, **kw)
        torch.Tensor.index_add_ = spy
        try:
            g = etu.events_to_voxel_grid_pytorch(ev.copy(), bins, W, H, torch.device("cpu")).numpy()
        finally:
            torch.Tensor.index_add_ = real
        assert len(seen) == 2 and np.array_equal(g, z["%s.grid_torch" % name])
        out["%s.ref_idx_left" % name], out["%s.ref_idx_right" % name] = seen
    np.savez_compressed(os.path.join(HERE, "voxel_indices.npz"), **out)
    print("voxel_indices.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
