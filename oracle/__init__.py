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
"""
