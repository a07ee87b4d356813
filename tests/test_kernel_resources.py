"""No kernel of libsubgnn_hip.so spills vector registers (round 4: the shipped degree-sequence instantiation spilled 4 VGPRs under a
5-waves-per-SIMD budget -- 59 MB of scratch traffic per launch for 8 MB of output -- and nothing in the suite noticed).  Reads the
code objects' metadata notes of the built objects; no GPU."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import kernel_resources as KR                                      # noqa: E402

LIBDIR = os.path.join(REPO, 'subgnn_amd', 'lib')
# (round 4's build had one accepted entry -- the 32-row DTW instantiation with 2 registers in scratch; the one-division cost
# function of round 5 freed them: no kernel of the library spills)
ALLOWED_VGPR_SPILLS = {}
# library kernels instantiated from headers (rocPRIM's sort): not ours to tune; their scratch is the library's choice
FOREIGN = ('rocprim::',)


def _objects():
    from subgnn_amd import build
    build.build(verbose=False)
    return sorted(os.path.join(LIBDIR, f) for f in os.listdir(LIBDIR) if f.endswith('.o'))


def test_no_kernel_spills_vector_registers():
    seen = 0
    bad = []
    for o in _objects():
        for k in KR.kernels(o):
            name = k['demangled']
            if any(f in name for f in FOREIGN):
                continue
            seen += 1
            allowed = max([v for p, v in ALLOWED_VGPR_SPILLS.items() if p in name], default=0)
            if k['vgpr_spill_count'] > allowed:
                bad.append('%s: %d VGPRs spilled (%d B scratch per lane), %d VGPRs' % (name[:100], k['vgpr_spill_count'],
                                                                                       k['private_segment_fixed_size'], k['vgpr_count']))
    assert seen > 100, 'the notes parser found only %d kernels' % seen
    assert not bad, '\n'.join(bad)


@pytest.mark.parametrize('pattern,max_vgprs', [
    ('degseq_wave_kernel<true, false, true>', 96),           # 5 wavefronts per SIMD (8 loads in flight)
    ('degseq_wave_kernel<true, false, false>', 96),          # 5
    ('dtw_similarity_reg_kernel<20, 0', 168),                # 3
])
def test_hot_kernels_keep_their_occupancy(pattern, max_vgprs):
    hits = [k for o in _objects() for k in KR.kernels(o) if pattern in k['demangled']]
    assert hits, pattern
    for k in hits:
        assert k['vgpr_count'] <= max_vgprs, (k['demangled'], k['vgpr_count'])
        assert k['vgpr_spill_count'] == 0


# Scalar-register spills of the hot kernels (the first test read VGPR spills only).  A spilled SGPR is a
# v_writelane / v_readlane pair -- vector-issue slots in kernels that are bound by vector issue -- so the counts are RATCHETED: a
# kernel may not spill more scalars than it did when its budget below was written (round 6 figures of the shipped build), and
# none of them may use a scratch segment.  Lowering a budget is welcome; raising one needs a reason next to it.
SGPR_SPILL_BUDGET = [
    ('dtw_similarity_reg_kernel<20, 0, 3, true>', 38),      # the benchmark's DTW launch (three tie rules share the budget)
    ('dtw_similarity_reg_kernel<20, 1, 3, true>', 38),
    ('dtw_similarity_reg_kernel<20, 2, 3, true>', 38),
    ('dtw_similarity_reg_kernel<12, ', 38),
    ('degseq_wave_kernel<true, false, true>', 38),          # the structure-channel CSR gather as the pass runs it (round 6: 18 -> 38 with the
    #                                                         five kernel arguments of the hub bitmaps and the per-node records, which made
    #                                                         the launch 21 % faster; the spilled scalars are the rarely used pointers)
    ('degseq_wave_kernel<true, false, false>', 107),        # its streaming form (measured beside the pass, not in it)
    ('khop1_sample_kernel<false>', 36),
    ('msbfs_level_kernel', 41),
    ('head_fwd_kernel', 38), ('head_bwd_kernel<1>', 31),
    ('mpn_fwd_kernel', 0), ('update_fwd_kernel', 0), ('update_bwd_dx_kernel', 0), ('update_bwd_dw_kernel', 0),
    ('contract_rows_partial_kernel', 0), ('rows_gemm_kernel', 0), ('rows_gemm_nt_kernel', 0), ('lstm_fwd_kernel', 0),
    ('lstm_bwd_kernel', 0), ('readout_sum_fwd_many_kernel', 0), ('optim_adam_kernel', 0), ('optim_sumsq_kernel', 0),
]


def test_hot_kernels_keep_their_scalar_spill_budget_and_use_no_scratch():
    ks = [k for o in _objects() for k in KR.kernels(o)]
    for pattern, budget in SGPR_SPILL_BUDGET:
        hits = [k for k in ks if pattern in k['demangled']]
        assert hits, pattern
        for k in hits:
            assert k['sgpr_spill_count'] <= budget, (k['demangled'][:100], k['sgpr_spill_count'], budget)
            assert k['private_segment_fixed_size'] == 0, (k['demangled'][:100], k['private_segment_fixed_size'])


def test_no_kernel_of_ours_uses_a_scratch_segment():
    """private_segment_fixed_size == 0 for every kernel of the library (round 5: masked_sum_slot_bwd_kernel<float4> kept a 32-byte
    stack slot for a select between two float4s, with no spill flagged)."""
    bad = [(k['demangled'][:100], k['private_segment_fixed_size']) for o in _objects() for k in KR.kernels(o)
           if k['private_segment_fixed_size'] > 0 and not any(f in k['demangled'] for f in FOREIGN)]
    assert not bad, bad
