"""Counter-based draw tape (oracle side) -- test infrastructure.

The reference consumes three global serial RNG streams in data-dependent amounts
(anchor_patch_samplers.py:70,74,78,80,98,100,102,103,106,177,189,206,208,222,326), so
"same seed" alone cannot give bit-exact parity for a parallel implementation
(hard part 2).  Parity is therefore defined *given a draw tape*: every
random decision is a pure function  draw64(seed, stream, item, j)  of

    seed    run seed
    stream  which sampler / split / layer / side is drawing (see STREAM_* below)
    item    which independent unit draws (walk number, matrix row * slots + slot, ...)
    j       the unit's own draw counter (walk: j-th draw of that walk; N anchors: node id)

The golden harness (tests/golden/make_goldens.py) monkey-patches np.random.choice,
random.uniform and torch.randn inside the imported reference modules to read this tape;
the HIP kernels and the C oracle regenerate the same function on the fly.
The product-side twin of this file is subgnn_amd/tape.py (same constants; the product
never imports oracle/).
"""
import numpy as np

MASK64 = (1 << 64) - 1
K_STREAM = 0x9E3779B97F4A7C15
K_ITEM = 0xD1B54A32D192ED03
K_DRAW = 0x8CB92BA72F3D8DD7
M1 = 0xBF58476D1CE4E5B9
M2 = 0x94D049BB133111EB

# stream ids (kind, plus per-call qualifiers folded in by stream_id())
STREAM_STRUCT_START = 1     # aps:222  start nodes of structure patches (ignored by walk patches)
STREAM_STRUCT_PATCH = 2     # aps:231  walks that *are* the structure anchor patches
STREAM_WALK_INT = 3         # aps:150  internal walks over a patch
STREAM_WALK_BOR = 4         # aps:150  border walks over a patch
STREAM_N_INT = 5            # aps:177  neighbourhood anchors, inside
STREAM_N_BOR = 6            # aps:189  neighbourhood anchors, border
STREAM_P_INT = 7            # aps:208  position anchors, inside
STREAM_P_EXT = 8            # aps:206  position anchors, border (shared)
STREAM_S_PICK = 9           # aps:326  which presampled structure patches a layer uses

SPLIT_CODE = {'train': 0, 'val': 1, 'test': 2}


def stream_id(kind, split=0, layer=0, epoch=0):
    """Fold (kind, split, layer, resample epoch) into one 64-bit stream number."""
    if isinstance(split, str):
        split = SPLIT_CODE[split]
    return (kind << 32) | (split << 24) | (epoch << 8) | layer


def mix64(z):
    z &= MASK64
    z = ((z ^ (z >> 30)) * M1) & MASK64
    z = ((z ^ (z >> 27)) * M2) & MASK64
    return z ^ (z >> 31)


def draw64(seed, stream, item, j):
    h = mix64((seed & MASK64) ^ ((stream * K_STREAM) & MASK64))
    h = mix64((h + item * K_ITEM) & MASK64)
    h = mix64((h + j * K_DRAW) & MASK64)
    return h


def draw32(seed, stream, item, j):
    return draw64(seed, stream, item, j) >> 32


def choice_index(seed, stream, item, j, n):
    """Uniform index in [0, n): high half of u32 * n (no rejection: one draw per choice)."""
    return (draw32(seed, stream, item, j) * n) >> 32


def uniform01(seed, stream, item, j):
    """random.uniform(0, 1) stand-in: u32 * 2**-32 as a float64 (exact)."""
    return draw32(seed, stream, item, j) * (1.0 / 4294967296.0)


def nanchor_pick(seed, stream, item, n, has_pad):
    """Neighbourhood-anchor law (twin of common.h sgnn_nanchor_index / sgnn_nanchor_allneg).

    The reference draws one N(0,1) variate per column of the padded id row, zeroes the PAD columns
    and takes the argmax (aps:177-179, 189-191).  In law: every non-PAD entry equally likely,
    except that PAD wins when all n real variates are negative (probability 2**-n) and the row has
    a PAD column.  The tape states that with two draws of the (row, slot) item:
      draw 1: all negative  iff  n <= 32 and the top n bits of u32 are zero;
      draw 0: index (u32 * n) >> 32 into the non-PAD entries in ASCENDING id order.
    Returns that index, or -1 when PAD wins (n == 0 included)."""
    if n == 0:
        return -1
    if has_pad and n <= 32 and (draw32(seed, stream, item, 1) >> (32 - n)) == 0:
        return -1
    return choice_index(seed, stream, item, 0, n)


def item_state(seed, stream, item):
    h = mix64((seed & MASK64) ^ ((stream * K_STREAM) & MASK64))
    return mix64((h + item * K_ITEM) & MASK64)


# ---- vectorised numpy versions (uint64 wrap-around arithmetic) -------------------------

def _mix64_np(z):
    z = z.astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(M1)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(M2)
    return z ^ (z >> np.uint64(31))


def draw64_np(seed, stream, item, j):
    with np.errstate(over='ignore'):
        item = np.asarray(item).astype(np.uint64)
        j = np.asarray(j).astype(np.uint64)
        h0 = np.uint64(mix64((seed & MASK64) ^ ((stream * K_STREAM) & MASK64)))
        h = _mix64_np(h0 + item * np.uint64(K_ITEM))
        h = _mix64_np(h + j * np.uint64(K_DRAW))
    return h
