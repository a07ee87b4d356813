"""CPU oracle for the SubGNN hot path -- TEST INFRASTRUCTURE ONLY.

Every function here is a plain numpy / pure-Python / torch-CPU restatement of the
reference algorithm (mims-harvard/SubGNN), citing the reference file:line it follows.
Nothing under ``subgnn_amd/`` imports this package: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may, and there
only as the checker / the timed CPU baseline -- never as the product path.

Parity pins (see DESIGN.md section "Oracle"):
  * pinned against outputs of the imported reference (tests/golden/*.npz, produced by
    tests/golden/make_goldens.py in the build container): graph order, CC ids, border
    sets, shortest-path similarities, degree sequences, triangular walks (given the draw
    tape), N/P/S anchor tensors (given the tape), get_anchor_patches, SG_MPN forward +
    grads, full forward logits / loss / grads, _pad_collate.
  * PARITY UNPINNED: fastdtw==0.3.4 (absent from the reference tree and from this image;
    restated from its published algorithm in oracle/fastdtw_restate.py) and the float
    summation order of torch-scatter's scatter-add (PyG 1.6.1, also absent).
    fastdtw's predecessor rule on ties is a switch (``tie_order`` 0 / 1 / 2, described in
    fastdtw_restate.py).  The product and the oracle default to 2, the shape the compiled variant most
    plausibly has: the pure-Python module (rule 0) raises on the empty series of the reference's padded
    component rows, so the reference's numbers came from the compiled one.  All three are property-tested
    against exact DTW; the goldens g7 / g11 are a self-consistency pin under rule 0 (generated through the
    restated pure-Python module), tests/golden/ties.npz pins rules 1 and 2 the same way.
"""
