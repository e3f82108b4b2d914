"""TEST INFRASTRUCTURE ONLY — CPU restatement (oracle) of the RAM-Net hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported CPU baseline.  The product package
(``rpg_ramnet_amd``) never imports this package and fails loudly when its HIP
library is missing.

Pinning: every function here is checked against golden vectors produced by
importing the reference (``/root/reference/RAM_Net``) in the build container —
see ``tests/golden/make_golden.py`` and ``tests/test_oracle_golden.py``.  The
only un-pinned piece is ``multi_scale_grad_loss`` (its arithmetic lives in
kornia 0.4.0, which is absent from ``/root/reference`` and from this image):
that function says "parity unpinned" in its docstring.
"""
