"""Counter-based draw tape, host side (twin of the device functions in csrc/common.h).

Randomness in the sampling kernels is a pure function  draw64(seed, stream, item, j):
``seed`` the run seed, ``stream`` which sampler / split / layer is drawing, ``item`` the
independent unit (walk number, row * slots + slot, subgraph number), ``j`` that unit's own
draw counter (the neighbourhood anchors use two: draw 0 = rank of the pick among the row's
ascending entries, draw 1 = the "every variate negative" event of the PAD rule).  It replaces the three global
serial RNG streams of the reference (anchor_patch_samplers.py:70-106,177,189,206-222,326),
which cannot be consumed in parallel; see DESIGN.md "Draw tape".
"""
MASK64 = (1 << 64) - 1

STREAM_STRUCT_START = 1
STREAM_STRUCT_PATCH = 2
STREAM_WALK_INT = 3
STREAM_WALK_BOR = 4
STREAM_N_INT = 5
STREAM_N_BOR = 6
STREAM_P_INT = 7
STREAM_P_EXT = 8
STREAM_S_PICK = 9

SPLIT_CODE = {'train': 0, 'val': 1, 'test': 2}


def stream_id(kind, split=0, layer=0, epoch=0):
    """``epoch``: resample epoch (resample_anchor_patches draws fresh anchors after every validation
    epoch, SubGNN.py:453-460); 0 for the draws of prepare_data / prepare_test_data."""
    if isinstance(split, str):
        split = SPLIT_CODE[split]
    assert 0 <= layer < 256 and 0 <= epoch < 65536
    return (kind << 32) | (split << 24) | (epoch << 8) | layer
