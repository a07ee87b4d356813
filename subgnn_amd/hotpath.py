"""Sparse precompute for base graphs where the reference's dense structures cannot exist.

The reference's prepare_data needs a dense float64 APSP matrix (N x N,
prepare_dataset/precompute_graph_metrics.py:23,69), (S, C, N) similarity slabs
(SubGNN/SubGNN.py:763) and a dense N x N adjacency per patch (SubGNN/subgraph_utils.py:136):
impossible at N = 1M.  ``prepare_sparse`` computes the same quantities only where the model
reads them, all on the GPU:

  * N-internal anchors lie inside their component        -> similarity 0
  * N-border anchors lie in the k-hop border             -> similarity = hop level, a by-product
                                                            of the border BFS (sgnn_khop_border)
  * P-border anchors are a handful of shared nodes       -> one bit-parallel multi-source BFS
                                                            (sgnn_bfs_hops) + a min over members
  * P-internal anchors are nodes of the same subgraph    -> 0 for single-component subgraphs,
                                                            multi-source BFS otherwise
  * structure similarities                               -> unchanged (degree sequences + DTW)

The result plugs into SubGNN.forward through the per-edge similarity dict (see
SubGNN._run_mpn_layer_fused).  Values are identical to what the dense path would gather from
its slabs (tests/test_gpu_hotpath.py).
"""
import time

import collections

import torch

from . import ops, tape, gamma, subgraph_utils
from . import anchor_patch_samplers as aps

MAX_PINT_BYTES = 16 << 30          # hop table (max_id+1) x (#distinct P-internal anchors) uint8 kept in HBM


class StageTimer:
    """HIP-event stage timing on the current stream (no host synchronisation until read)."""

    def __init__(self, enabled=True):
        self.enabled, self.marks, self.host = enabled, [], []

    def mark(self, name):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))
            self.host.append(time.perf_counter())          # when the host got here (it runs ahead of the device)

    def host_summary(self):
        """ms between the host reaching consecutive marks: a stage whose host time exceeds its device
        time is bound by launching, not by the kernels."""
        return {n1: 1e3 * (t1 - t0) for (n1, _), t0, t1 in zip(self.marks[1:], self.host[:-1], self.host[1:])}

    def summary(self):
        out = {}
        for (n0, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out


def _hand_over(stream, *objs):
    """Tensors produced on a side stream and consumed on ``stream``: tell the caching allocator."""
    for o in objs:
        if o is None:
            continue
        if isinstance(o, torch.Tensor):
            if o.is_cuda:
                o.record_stream(stream)
            for nm in ('_sgnn_ids32', '_sgnn_sorted', '_sgnn_both', '_sgnn_all', '_sgnn_mask', '_sgnn_mask_u8'):
                extra = getattr(o, nm, None)
                if extra is not None:
                    _hand_over(stream, extra)
        elif isinstance(o, ops.Ragged):
            _hand_over(stream, o.ptr, o.nodes)
        elif isinstance(o, dict):
            _hand_over(stream, *o.values())
        elif isinstance(o, (list, tuple)):
            _hand_over(stream, *o)


BFS_LEVEL_MARGIN = 2
WALKS_ON_SIDE_MAX_SETS = 16384      # component rows up to which the structure walks run on the side stream ahead of the position search


def _bfs_levels(model, key, max_hops):
    """How many BFS levels to enqueue for the position channel's search ``key``.  The kernels run without a host
    synchronisation, so every enqueued level costs two launches (expand, which also reduces the previous level over the
    sets, and commit) whether or not the search has ended -- with the
    default cap of 32 hops and a small-world graph that is ~26 empty levels (~0.3 ms of a 19 ms pass).  The first
    search of a kind runs the full cap and its status (last productive level) is read back once; later ones enqueue
    (levels needed so far + BFS_LEVEL_MARGIN productive levels + the one empty level that proves the search has ended)
    and are verified BEFORE the pass is consumed (``_verify_bfs`` in install_pass)."""
    hint = model.__dict__.setdefault('_bfs_level_hint', {})
    k = hint.get(key)
    return max_hops if k is None else min(max_hops, k + BFS_LEVEL_MARGIN + 1)


BFS_PUSH_MARGIN = 1


def _bfs_push_levels(model, key):
    """How many levels of the search ``key`` may still push (each is a second launch, the commit): the level at which the
    first search of the kind switched to pulling, plus a margin; -1 (all of them) until that is known or when it never
    pulled.  Beyond them the device pulls whatever the frontier -- same results (ops.bfs_min_hops_to_sets)."""
    k = model.__dict__.setdefault('_bfs_push_hint', {}).get(key)
    return -1 if not k else int(k) - 1 + BFS_PUSH_MARGIN


def _bfs_note(model, st, key, status, max_hops, enqueued, redo):
    """First search of a kind: read its status (one blocking read-back) and keep the depth.  Hinted search: copy the
    status into pinned host memory behind the search (no wait here) and leave a check on the pass; ``redo`` runs the
    search again with the full cap."""
    hint = model.__dict__.setdefault('_bfs_level_hint', {})
    if torch.cuda.is_current_stream_capturing():
        # the pass is being recorded into a hipGraph (GraphedPasses): the status travels to a pinned buffer that belongs to the
        # recording (every replay rewrites it); no event -- the replay's own completion event orders the read
        if hint.get(key) is None:
            raise RuntimeError('a recorded pass needs the BFS depth of an earlier eager pass (GraphedPasses warms up first)')
        pool = model.__dict__.setdefault('_bfs_status_pool', [])
        if not pool:                                 # (pinning allocates host memory: not allowed while a stream is capturing)
            raise RuntimeError('no pinned status buffer left for a recorded search: GraphedPasses._record provides them')
        host = pool.pop()
        host.copy_(status, non_blocking=True)
        st.bfs_checks.append((key, host, None, max_hops, enqueued, None))
        return
    if hint.get(key) is None:
        last, more, first_pull = status.tolist()[:3]
        model.__dict__.setdefault('_bfs_push_hint', {})[key] = first_pull
        if more:
            raise RuntimeError('position-channel BFS: level %d still reached new nodes -- hparams["max_bfs_hops"] = %d '
                               'is smaller than the depth of this graph from the anchors' % (max_hops, max_hops))
        hint[key] = last
        return
    # one pinned status buffer per check, taken from a free list and returned by _verify_bfs once it has been read: any
    # number of passes may be in flight (PassPipeline.start can run ahead as far as the caller likes)
    pool = model.__dict__.setdefault('_bfs_status_pool', [])
    host = pool.pop() if pool else torch.empty(4, dtype=torch.int32).pin_memory()
    host.copy_(status, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    st.bfs_checks.append((key, host, ev, max_hops, enqueued, redo))


class BfsLevelsExhausted(RuntimeError):
    """A recorded pass's hinted position-channel search ran out of levels (the hint has been raised to the cap)."""


def _verify_bfs(model, st, keep=False):
    """Before a prepared pass becomes the model's: did every hinted position-channel search end within the levels it
    was given?  The status was copied to pinned memory right behind the search -- by now (the pass was prepared a step
    ago under the pipeline, or the caller is about to wait for the table anyway) the copy has landed and the event
    wait costs nothing.  A search that ran out of levels is REPEATED with the full cap and its similarities replaced
    before anything reads them; only a graph deeper than hparams['max_bfs_hops'] itself is an error."""
    hint = model.__dict__.setdefault('_bfs_level_hint', {})
    checks = list(st.bfs_checks)
    # every status is READ first (all copies have landed once the last event has), and the pinned buffers of an eager pass go
    # back to the free list in one place whatever happens below: a raise in the middle of the loop used to leave the buffers
    # of the remaining checks outside the pool while st.bfs_checks still listed the ones already returned
    results = []
    try:
        for key, host, ev, cap, enqueued, redo in checks:
            if ev is not None:
                ev.synchronize()
            results.append((int(host[0]), int(host[1])))
            if int(host[2]) > 0:                     # the level that switched to pull: the push hint only grows
                ph = model.__dict__.setdefault('_bfs_push_hint', {})
                ph[key] = max(ph.get(key) or 0, int(host[2]))
    finally:
        if not keep:
            st.bfs_checks = []
            pool = model.__dict__.setdefault('_bfs_status_pool', [])
            for key, host, ev, cap, enqueued, redo in checks:
                if ev is not None:                       # (a recording's buffers -- ev None -- belong to the recording: release_checks)
                    pool.append(host)
    for (key, host, ev, cap, enqueued, redo), (last, more) in zip(checks, results):
        if more and enqueued < cap:
            if redo is None:                         # a recorded pass cannot repeat a search: its owner records again
                hint[key] = cap
                raise BfsLevelsExhausted(key)
            last, more = redo(cap)[:2]
            model.__dict__['_bfs_redone'] = model.__dict__.get('_bfs_redone', 0) + 1
        if more:
            raise RuntimeError('position-channel BFS: level %d still reached new nodes -- hparams["max_bfs_hops"] = %d '
                               'is smaller than the depth of this graph from the anchors' % (cap, cap))
        hint[key] = max(hint.get(key) or 0, last)                  # the hint only grows: anchors are redrawn every pass


def release_checks(model, checks):
    """The pinned status buffers of a recording that is being dropped go back to the free list (GraphedPasses)."""
    pool = model.__dict__.setdefault('_bfs_status_pool', [])
    for c in checks:
        if c[1] is not None and not any(c[1] is h for h in pool):
            pool.append(c[1])


def check_pending(model):
    """Kept for callers of round 2: the hinted searches are verified inside install_pass now, nothing is pending."""
    return None


def _pint_sims_streamed(g, uniq, inv, cc_sets, S, C, max_hops, chunk_bytes=1 << 30):
    """P-internal similarities of multi-component subgraphs when the (distinct anchors x nodes) hop table would not fit
    (round 2 raised NotImplementedError above 16 GiB; the reference has no such limit short of its N x N matrix,
    SubGNN.py:752-781): the distinct anchors go through the fused search ``sgnn_bfs_min_hops_to_sets`` a block of
    sources at a time -- no hop table, the largest temporary is (S*C, block) floats -- and every (subgraph, slot)
    whose anchor falls into the block takes its column.  One multi-source BFS per block: slow for millions of distinct
    anchors, but bounded in memory."""
    U = int(uniq.numel())
    rows = S * C
    block = max(64, min(U, (chunk_bytes // (4 * max(rows, 1))) // 64 * 64))
    w = torch.zeros((S, C, inv.shape[1]), dtype=torch.float32, device=inv.device)
    src = uniq.to(torch.int32).contiguous()
    rix = (torch.arange(S, device=inv.device).view(S, 1, 1) * C + torch.arange(C, device=inv.device).view(1, C, 1))
    for a in range(0, U, block):
        b = min(U, a + block)
        part, status = ops.bfs_min_hops_to_sets(g, src[a:b], cc_sets, max_hops=max_hops, want_status=True)   # (S*C, b - a)
        if int(status[1]):
            raise RuntimeError('position-channel BFS: level %d still reached new nodes -- hparams["max_bfs_hops"] = %d '
                               'is smaller than the depth of this graph from the anchors' % (max_hops, max_hops))
        hit = (inv >= a) & (inv < b)                                               # (S, A)
        col = (inv - a).clamp(0, b - a - 1).view(S, 1, -1).expand(S, C, -1)
        got = part[rix.expand(S, C, inv.shape[1]), col]
        w = torch.where(hit.view(S, 1, -1), got, w)
    return w


def _deal_rows(shard, n, compute, tail_shape, dtype, device, key=None):
    """Strong scaling: rows [0, n) of a result every rank needs (walks over the shared structure patches) computed as
    ``world`` equal shares -- ``compute(lo, hi)`` -> rows lo..hi-1 -- and all-gathered (ranks in order, equal row counts
    known on the host: no size exchange).  Bit-identical to computing all rows on every rank when ``compute`` keys its
    draws by global row numbers.  ``key``: names the exchange for an emulated rank (dist.EmulatedPeers)."""
    from . import dist as sdist
    per = -(-n // shard.world)
    lo = min(shard.rank * per, n)
    hi = min(lo + per, n)
    part = compute(lo, hi) if hi > lo else torch.zeros((0,) + tuple(tail_shape), dtype=dtype, device=device)
    emu = getattr(shard, 'emulator', None)
    if emu is not None:
        def everybody():
            return torch.cat([compute(min(r * per, n), min(r * per + per, n)) for r in range(shard.world) if min(r * per, n) < n], 0)
        return emu.exchange(key, part, lo, hi, everybody)
    if part.shape[0] < per:
        part = torch.cat([part, part.new_zeros((per - part.shape[0],) + tuple(part.shape[1:]))], 0)
    return sdist.all_gather_rows(part.contiguous(), equal_rows=True)[:n]


def _emulated_position_sims(g, anchors, cc_sets, cc_ids, shard, max_hops, key):
    """_dealt_position_sims for ONE process standing in for a rank (dist.EmulatedPeers): the rank's own search -- its share of
    the sources over EVERY rank's components -- runs as in the real form; the gathered component ids come from the emulator
    (supplied by its owner: only the other ranks could compute them) and the other ranks' columns for this rank's rows from a
    search over all sources recorded on the first pass."""
    from . import dist as sdist
    emu = shard.emulator
    S, C, Lc = cc_ids.shape
    A = anchors.numel()
    rows = S * C
    mine = cc_ids.reshape(rows, Lc)
    all_cc = emu.exchange(('cc_ids', key), mine, shard.rank * rows, (shard.rank + 1) * rows, lambda: emu.provided['cc_ids_all'])
    if all_cc.shape[0] != rows * shard.world or all_cc.shape[1] != Lc:
        raise ValueError('EmulatedPeers: the supplied component ids are %s, this rank holds %s of %d ranks' % (tuple(all_cc.shape), (rows, Lc), shard.world))
    all_sets = ops.Ragged.from_padded(all_cc)
    a, b = sdist.shard_range(A, shard.rank, shard.world)
    status = torch.zeros(4, dtype=torch.int32, device=cc_ids.device)
    part = None
    if b > a:
        part, status = ops.bfs_min_hops_to_sets(g, anchors[a:b].to(torch.int32).contiguous(), all_sets, max_hops=max_hops, want_status=True)

    def everybody():
        full, st = ops.bfs_min_hops_to_sets(g, anchors.to(torch.int32).contiguous(), cc_sets, max_hops=max_hops, want_status=True)
        if int(st[1]):
            raise RuntimeError('position-channel BFS: hparams["max_bfs_hops"] = %d is smaller than the depth of this graph' % max_hops)
        return full
    block = part[shard.rank * rows:(shard.rank + 1) * rows] if part is not None else torch.zeros((rows, 0), dtype=torch.float32, device=cc_ids.device)
    return emu.exchange(('P_out', key), block, a, b, everybody, dim=1), status


def _dealt_position_sims(g, anchors, cc_sets, cc_ids, shard, max_hops):
    """P-border similarities with the BFS sources dealt across ranks (strong scaling): this rank runs the
    multi-source BFS for ITS share of the shared anchors only -- one 64-source word instead of
    ceil(A / 64) -- but for every rank's components (the padded component tensors are all-gathered:
    4 MB at the benchmark size), and an all-to-all hands every rank all anchors' columns for its own
    rows.  Values are identical to the replicated form."""
    from . import dist as sdist
    S, C, Lc = cc_ids.shape
    A = anchors.numel()
    all_cc = sdist.all_gather_rows(cc_ids.reshape(S * C, Lc), equal_rows=True)
    all_sets = ops.Ragged.from_padded(all_cc)
    a, b = sdist.shard_range(A, shard.rank, shard.world)
    width = (A + shard.world - 1) // shard.world
    part = torch.zeros((all_sets.n, width), dtype=torch.float32, device=cc_ids.device)
    status = torch.zeros(4, dtype=torch.int32, device=cc_ids.device)
    if b > a:
        part[:, :b - a], status = ops.bfs_min_hops_to_sets(g, anchors[a:b].to(torch.int32).contiguous(), all_sets,
                                                           max_hops=max_hops, want_status=True)
    # a graph deeper than max_hops from some rank's anchors must stop EVERY rank (the replicated form raises for it)
    status = sdist.all_reduce_max_(status)
    got = sdist.all_to_all_row_blocks(part)                         # block j: rank j's anchors, my rows
    rows = S * C
    cols = [got[j * rows:(j + 1) * rows, :sdist.shard_range(A, j, shard.world)[1] - sdist.shard_range(A, j, shard.world)[0]]
            for j in range(shard.world)]
    return torch.cat(cols, dim=1), status


class PassState:
    """What one pass of ``prepare_pass`` produced for a split: the per-pass attributes of the model (component ids,
    anchors, walks, similarity rows) by name, not yet visible to the model.  ``install_pass`` makes them the model's."""

    def __init__(self, split):
        self.split = split
        self.attrs = {}            # attribute name -> value
        self.per_split = {}        # attribute name (a dict keyed by split on the model) -> this split's value
        self.sim_cols = None       # (anchors_structure, {layer: device index tensor}) when new patches were drawn
        self.dtw_inputs = None     # between prepare_pass(defer_dtw=True) and finish_pass: what the DTW launches read
        self.bfs_checks = []       # hinted position-channel searches to verify before the pass is consumed (_verify_bfs)
        self.pool_reused = False   # the pass re-picked from an earlier pass's structure-patch pool (prepare_pass(pool=...))

    def tensors(self):
        return [self.attrs, self.per_split, self.sim_cols[1] if self.sim_cols else None]


def pool_of(st):
    """The structure-patch pool a full pass built (patches, walks, similarity rows of every pool patch): what a later pass may
    reuse instead of rebuilding it (``prepare_pass(pool=...)``)."""
    sp = st.split
    return {'structure_anchors': st.attrs['structure_anchors'],
            'int_w': st.attrs['int_structure_anchor_random_walks'], 'bor_w': st.attrs['bor_structure_anchor_random_walks'],
            'int_sims': st.attrs[sp + '_int_struc_similarities'], 'bor_sims': st.attrs[sp + '_bor_struc_similarities']}


def prepare_pass(model, split='train', timer=None, shard=None, defer_dtw=False, pool=None, epoch=None):
    """The sampling + similarity half of a pass (everything that does not read the embedding table), for one split
    (SubGNN.py:1024-1063 semantics, sparse similarities) -> PassState.  The model's per-pass attributes are left
    alone (only its per-split caches of pass-invariant facts are filled), so the pass can be prepared while the
    previous one is still training on the model (PassPipeline).
    ``shard`` (dist.Shard): the model holds one rank's block of the split's subgraphs; draws then read the
    tape items of the GLOBAL subgraph numbers and padded widths are reduced over ranks, so that the
    sharded passes together reproduce the single-rank pass bit for bit.

    Two HIP streams (hparams['overlap_streams'], default True): after the components, the main stream runs the
    border BFS + neighbourhood draws and the component degree sequences, a side stream runs the structure patches,
    the position channel's multi-source BFS and the triangular walks; they join before the DTW.  The border kernel
    (1024-thread workgroups around an LDS bitmap) and the walk kernels leave most wavefront slots of a CU empty and
    the BFS kernels are memory-bound with small workgroups: on the benchmark the three stages take 4.7 ms back to
    back and 4.0 ms overlapped (17.9 -> 17.0 ms per pass).  The DTW cannot share a CU (it holds every vector
    register at three wavefronts of 168 registers per SIMD), so nothing is overlapped with it.  The strong-scaling form issues
    collectives inside the position block (side stream) and around the dealt patches / walks (main stream).
    ``pool`` (``pool_of`` an earlier pass of the same split): the structure-patch pool is REUSED -- no patches, walks, degree
    sequences or DTW launches; the pass re-picks its layers' patches from the pool (init_anchors_structure) and reads the
    pool's similarity rows.  The reference's design: max_sim_epochs x n_anchor_patches_structure x n_layers patches are sampled
    and scored once (SubGNN.py:783-833, anchor_patch_samplers.py:210-243) so that re-picking is free (aps:316-328)."""
    hp, g, dev = model.hparams, model.networkx_graph, model.device
    if pool is not None and (shard is not None and shard.deal_shared):
        raise ValueError('pool reuse is not combined with the dealt (strong-scaling) form')
    seed = int(hp.get('seed', 0)) & tape.MASK64
    # the resample epoch of the draws (SubGNN.py:453-460: resample_anchor_patches draws fresh N / P anchors and re-picks the structure
    # patches after every validation epoch; the patches and their walks are not drawn again): the model's, unless the caller says
    ep = int(model.__dict__.get('_resample_epoch', 0)) if epoch is None else int(epoch)
    st = PassState(split)
    st.pool_reused = pool is not None
    t = timer or StageTimer(False)
    L = hp['n_layers']
    main = torch.cuda.current_stream()
    # (the dealt -- strong-scaling -- form overlaps too: its exchanges are issued by this one host thread in one fixed order on
    # every rank, and torch's process group runs them on its own stream behind whichever stream issued them: the position search
    # with its all-to-all on the side stream beside the border / walks / DTW chain.  hparams['overlap_streams_dealt'] = False
    # keeps the dealt form on one stream, as rounds 2-5 ran it)
    if hp.get('overlap_streams', True) and (hp.get('overlap_streams_dealt', True) or not (shard is not None and shard.deal_shared)):
        if getattr(model, '_side_stream', None) is None:
            model._side_stream = torch.cuda.Stream()
        side = model._side_stream
    else:
        side = main
    t.mark('start')
    # ---- components (main) ---------------------------------------------------------------
    subs = ops.Ragged.from_lists(getattr(model, split + '_sub_G'), dev) if not hasattr(model, '_subs_' + split) \
        else getattr(model, '_subs_' + split)
    setattr(model, '_subs_' + split, subs)
    labels = ops.cc_labels(g, subs)
    # the padded (components per subgraph, component length) is a property of the split's subgraph lists (and, sharded,
    # of all ranks'): read back on the first pass, kept afterwards -- one statistics launch and one host round trip less
    kept = model.__dict__.setdefault('_cc_dims', {})
    tag = (id(getattr(model, split + '_sub_G')), subs.n)
    hit = kept.get(split)
    cc_ids = subgraph_utils.components_from_labels(subs.ptr, subs.nodes, labels, subs.max_len,
                                                   dims_reduce=shard.reduce_max if shard is not None else None,
                                                   dims=hit[1] if hit is not None and hit[0] == tag else None)
    kept[split] = (tag, tuple(cc_ids.shape[1:]))
    st.attrs[split + '_cc_ids'] = cc_ids
    S, C, Lc = cc_ids.shape
    if getattr(model, '_deterministic', ops.DETERMINISTIC):
        # the component-embedding backward scatters per member id in sorted order: the members of a split's
        # components are the same every pass, so their order is computed on the first pass and kept
        mo = model.__dict__.setdefault('_cc_member_order', {})
        if mo.get(split) is None or mo[split][0].numel() != cc_ids.numel():
            mo[split] = ops.sort_edges_by_key(cc_ids.reshape(-1).to(torch.int32), g.max_id)
        cc_ids._sgnn_member_order = mo[split]
    cc_ids._sgnn_ids32 = cc_ids.reshape(-1).to(torch.int32)        # what the component-embedding kernel reads
    base = shard.start if shard is not None else 0             # global number of this rank's first subgraph
    cc_sets = ops.Ragged.from_padded(cc_ids.reshape(S * C, Lc))
    real = (cc_ids[:, :, 0] != 0)
    # which component rows are real, as forward reads it (SubGNN._forward: bool (S, C) and its uint8 twin): made here, beside the
    # sampling stages, and carried on the tensor -- two small launches less per training half
    cc_ids._sgnn_mask = real
    cc_ids._sgnn_mask_u8 = real.reshape(-1).to(torch.uint8)
    t.mark('components')
    # dispatch order of the component sets, heaviest (largest total degree) first: a property of the
    # split's subgraphs and the graph, computed once per split and kept.  The set kernels whose cost is
    # the members' degree sum (degree sequences, border BFS) take their sets in this order.
    orders = model.__dict__.setdefault('_degseq_order', {})
    if orders.get(split) is None or orders[split].numel() != cc_sets.n:
        orders[split] = ops.heaviest_first(g, cc_sets)
        t.mark('set_dispatch_order(first pass only)')
    set_order = orders[split]
    sims = {}
    a_sets = ai = ae = None
    det = bool(getattr(model, '_deterministic', ops.DETERMINISTIC))
    # ---- side stream: position channel + structure patches / walks -------------------------
    side.wait_stream(main)
    with torch.cuda.stream(side):
        # The structure patches keep the walks' full width (no trim to the longest walk: that was a host round trip, and
        # with a second prepared pass queued on this stream the host waited for that pass's DTW launch there).
        new_patches = hp['use_structure'] and pool is None and (split != 'test' or getattr(model, 'structure_anchors', None) is None)
        structure_anchors = getattr(model, 'structure_anchors', None) if pool is None else pool['structure_anchors']
        if new_patches:
            if shard is not None and shard.deal_shared and hp['structure_patch_type'] == 'triangular_random_walk':
                # strong scaling: every rank walks an eighth of the shared patches (tape items = global walk numbers)
                n_p = hp['max_sim_epochs'] * hp['n_anchor_patches_structure'] * hp['n_layers']
                structure_anchors = _deal_rows(shard, n_p, lambda lo, hi: aps.sample_structure_anchor_patches(
                    hp, g, dev, hp['max_sim_epochs'], trim=False, share=(lo, hi)), (hp['sample_walk_len'],), torch.int64, dev,
                    key=('S_patches', split))
                st.attrs['structure_anchors'] = structure_anchors
            else:
                structure_anchors = st.attrs['structure_anchors'] = aps.sample_structure_anchor_patches(hp, g, dev, hp['max_sim_epochs'], trim=False)
            if side is main:
                t.mark('S_patches_walks')
        patch_ev = None
        if hp['use_structure'] and side is not main:
            patch_ev = torch.cuda.Event()
            patch_ev.record(side)                       # the patches exist: their walks run on the main stream (below) and
            #                                             wait for the patches only, not for the position BFS queued next
        def position_block():
            """Position channel: the shared P-border anchors, the per-subgraph P-internal draws and the multi-source BFS."""
            if hp['use_position']:
                anchors_pos_ext = getattr(model, 'anchors_pos_ext', None)
                if anchors_pos_ext is None or split != 'test':
                    anchors_pos_ext = st.attrs['anchors_pos_ext'] = aps.init_anchors_pos_ext(hp, g, dev, epoch=ep)
                    if det:
                        for v in anchors_pos_ext.values():
                            ops.presort_ids(v, g.max_id)
                pint = {l: ops.choice_ragged(subs, hp['n_anchor_patches_pos_in'], seed,
                                             tape.stream_id(tape.STREAM_P_INT, split, l, ep), item_base=base) for l in range(L)}
                st.per_split['anchors_pos_int'] = pint
                for l in range(L):
                    if shard is not None and shard.deal_shared:
                        cap = hp.get('max_bfs_hops', 32)
                        if getattr(shard, 'emulator', None) is not None:
                            w, status = _emulated_position_sims(g, anchors_pos_ext[l], cc_sets, cc_ids, shard, cap, (split, l))
                        else:
                            w, status = _dealt_position_sims(g, anchors_pos_ext[l], cc_sets, cc_ids, shard, cap)
                        w = w.view(S, C, -1)
                        # full cap, all ranks' worst status: verified before the pass is consumed like the hinted searches
                        # (nothing to repeat: running out of levels here means max_bfs_hops is too small)
                        model.__dict__.setdefault('_bfs_level_hint', {}).setdefault(('P_out_dealt', split, l), 0)
                        _bfs_note(model, st, ('P_out_dealt', split, l), status, cap, cap, None)
                    else:
                        cap = hp.get('max_bfs_hops', 32)
                        nlev = _bfs_levels(model, ('P_out', split, l), cap)
                        src = anchors_pos_ext[l].to(torch.int32).contiguous()
                        w, status = ops.bfs_min_hops_to_sets(g, src, cc_sets, max_hops=nlev, want_status=True,
                                                             push_levels=_bfs_push_levels(model, ('P_out', split, l)))

                        def redo(levels, src=src, l=l, sims=sims):
                            # the hinted search ran out of levels: the same search with the full cap, similarities replaced.
                            # It runs on the INSTALLING stream: what it reads was allocated on the preparation stream
                            _hand_over(torch.cuda.current_stream(), src, cc_sets)
                            w2, st2 = ops.bfs_min_hops_to_sets(g, src, cc_sets, max_hops=levels, want_status=True)
                            sims[('P', 'out', l)] = w2.view(S, C, -1).contiguous()
                            return st2.tolist()
                        _bfs_note(model, st, ('P_out', split, l), status, cap, nlev, redo)
                        w = w.view(S, C, -1)
                    # (rows of padded components need no masking: an empty set receives no level in msbfs_set_reduce and
                    # keeps the 0 the output was cleared to -- test_sparse_prepare_equals_dense_prepare checks the raw rows)
                    sims[('P', 'out', l)] = w.contiguous()
                    if C == 1:
                        sims[('P', 'in', l)] = ops.ZeroSims((S, C, hp['n_anchor_patches_pos_in']), dev)
                    else:
                        uniq, inv = torch.unique(pint[l], return_inverse=True)
                        if uniq.numel() * (g.max_id + 1) <= MAX_PINT_BYTES:
                            d = ops.bfs_hops(g, uniq.to(torch.int32).contiguous(), max_hops=hp.get('max_bfs_hops', 32),
                                             node_major=True)
                            full = ops.min_hops_to_sets(d, cc_sets, node_major=True).view(S, C, -1)     # (S, C, U)
                            w = torch.gather(full, 2, inv.view(S, 1, -1).expand(S, C, -1))
                        else:
                            w = _pint_sims_streamed(g, uniq, inv, cc_sets, S, C, hp.get('max_bfs_hops', 32))
                        sims[('P', 'in', l)] = (w * real.unsqueeze(-1)).contiguous()
                if side is main:
                    t.mark('P_bfs_sims')
        # Where the position search runs.  It is memory-bound (pull levels gather a 32-byte row per edge) and the DTW launches are
        # bound by fp64 vector issue: started TOGETHER the two share the chip (tools/dtw_overlap_probe.py: DTW 4.6 ms + search
        # 1.5 ms take 5.0 ms side by side, 69 % of the search hidden; degree sequences, walks and the table's Adam hide
        # completely, the one-hop border kernel 38 %).  hparams['bfs_beside_dtw'] queues the search behind an event the main
        # stream records right before the DTW launches.  Measured in the pipelined schedule (round 4): the border stage drops
        # from 2.66 to 1.90 ms and the DTW stage grows from 5.2 to 6.2 -- the step stays at 9.4-9.5 ms, because the training
        # half of the previous pass already fills what the DTW launch leaves free: the device is saturated by total work, the
        # schedule only moves it around.  Default False (the search beside the border kernel, as before).
        late_bfs = bool(hp.get('bfs_beside_dtw', False)) and side is not main and hp['use_structure'] and hp['use_position'] \
            and not defer_dtw and not (shard is not None and shard.deal_shared)
        wos = hp.get('walks_on_side_stream')            # None: by size (below); True / False: forced
        walks_on_side = (bool(wos) if wos is not None else cc_sets.n <= WALKS_ON_SIDE_MAX_SETS) and side is not main and hp['use_structure']

        def structure_walks(degree_sequences=True):
            """Walks over the structure patches and the per-layer picks (+ the patches' degree sequences unless the caller runs
            them elsewhere).  Which stream: see ``walks_on_side`` below."""
            nonlocal a_sets, ai, ae
            if new_patches and shard is not None and shard.deal_shared:
                # the walks over the shared patches, dealt: a rank builds the node views of ITS patches only
                W_, T_ = hp['n_triangular_walks'], hp['random_walk_len']

                def both(lo, hi):
                    mine = structure_anchors[lo:hi].contiguous()
                    v = aps.patch_node_views(mine)
                    iw, bw = aps.perform_random_walks_both(hp, g, mine, v, first_patch=lo)
                    return torch.stack([bw, iw], 1)
                walks = _deal_rows(shard, structure_anchors.shape[0], both, (2, W_, T_), torch.int64, dev, key=('S_walks', split))
                bor_w = st.attrs['bor_structure_anchor_random_walks'] = walks[:, 0].contiguous()
                int_w = st.attrs['int_structure_anchor_random_walks'] = walks[:, 1].contiguous()
            elif new_patches:
                views = aps.patch_node_views(structure_anchors)
                # (internal and border walks in ONE launch: 1050 + 1050 one-workgroup-per-CU walks fill the chip's rounds better)
                int_w, bor_w = aps.perform_random_walks_both(hp, g, structure_anchors, views)
                st.attrs['bor_structure_anchor_random_walks'], st.attrs['int_structure_anchor_random_walks'] = bor_w, int_w
            if pool is not None:
                int_w, bor_w = pool['int_w'], pool['bor_w']
                st.attrs['structure_anchors'] = structure_anchors
                st.attrs['int_structure_anchor_random_walks'], st.attrs['bor_structure_anchor_random_walks'] = int_w, bor_w
            if new_patches or pool is not None:
                a_struct = st.attrs['anchors_structure'] = aps.init_anchors_structure(hp, structure_anchors, int_w, bor_w,
                                                                                      indices_on_device=True, epoch=ep)
                if det:
                    # forward runs the LSTM once over a layer's internal AND border walks (SubGNN._structure_anchor_embeddings):
                    # the stacked walks and the sort their embedding lookup's backward needs are made here
                    for v in a_struct.values():
                        v[2]._sgnn_both = ops.presort_ids(torch.cat([v[2], v[3]], 0), g.max_id)
                    if L > 1:                            # (forward runs ONE LSTM pass over the walks of all layers)
                        a_struct[0][2]._sgnn_all = ops.presort_ids(
                            torch.cat([t for l in range(L) for t in (a_struct[l][2], a_struct[l][3])], 0), g.max_id)
                # the column upload is a blocking host->device copy: do it here, before the long DTW
                # launches are queued, so that the host is free to queue forward/backward behind them
                st.sim_cols = (a_struct, model.sim_cols_of(a_struct))
            if pool is not None or not degree_sequences:
                return                                   # the pool's similarity rows exist: no degree sequences, no DTW
            patch_degree_sequences()

        def patch_degree_sequences():
            """The structure patches' degree sequences: what the DTW launches read of the patches (not the walks)."""
            nonlocal a_sets, ai, ae
            a_sets = ops.Ragged.from_padded(structure_anchors)
            ai, ae = ops.degree_sequence(g, a_sets, sort=True, use_degree_dict=g.full_degree is not None)
        # Where the walks run.  For shards of up to WALKS_ON_SIDE_MAX_SETS component rows (round 6): on the SIDE stream, right
        # behind the patches and AHEAD of the position search.  Nothing on the main chain reads the walks -- the DTW launches read
        # the patches' degree sequences only -- so they leave it; and they no longer run BESIDE the search: a walk is one
        # 1024-thread workgroup around a 125 KB LDS bitmap, it needs 16 free wavefront slots on one CU at once, and the search's
        # small workgroups kept every CU's slots busy -- at 6 250 subgraphs the walks' stage took 1.3-1.45 ms on the main chain
        # for 0.27 ms of kernel: the pass 3.08 -> 2.79 ms (same box, back to back).  At the benchmark's 50k subgraphs the main
        # chain is the long one either way and the walks hide better there (8.82 vs 8.87 ms pipelined, 9.31 vs 9.60 sequential):
        # they stay on the main chain, as rounds 3-5 had them.  hparams['walks_on_side_stream'] = True / False forces either.
        if walks_on_side:
            structure_walks(degree_sequences=False)
        if not late_bfs:
            position_block()
        if hp['use_structure'] and side is main:
            structure_walks()
            t.mark('S_patches_walks')
    # ---- main stream: neighbourhood channel + component degree sequences --------------------
    if hp['use_neighborhood']:
        k = hp['neigh_sample_border_size']
        has_pad_c = (cc_sets.lengths < Lc).to(torch.uint8)
        cc_canon = ops.sort_ragged(cc_sets)                     # the draw ranks the ascending members
        ni, nb, plans = {}, {}, {}
        for l in range(L):
            ni[l] = ops.sample_anchors_ragged(cc_canon, hp['n_anchor_patches_N_in'], seed,
                                              tape.stream_id(tape.STREAM_N_INT, split, l, ep), has_pad_c,
                                              canonical=True, item_base=base * C).view(S, C, -1)
            sims[('N', 'in', l)] = ops.ZeroSims(ni[l].shape, dev)
            # border BFS fused with the border-anchor draw (rank query on the visited bitmap): the
            # border is never materialised
            # the padded border matrix's width (the largest border: the PAD rule needs it) is a property of the split's
            # components and the graph, not of the draw: reduced (over the ranks too) on the first pass, kept afterwards
            widths = model.__dict__.setdefault('_border_width', {})
            wkey = (split, k, cc_sets.n, shard.world if shard is not None else 1)
            a, w, counts = ops.khop_border_sample(g, cc_sets, k, hp['n_anchor_patches_N_out'], seed,
                                                  tape.stream_id(tape.STREAM_N_BOR, split, l, ep),     # taken dynamically: a dispatch order buys nothing here
                                                  item_base=base * C,
                                                  count_reduce=shard.reduce_max if shard is not None else None,
                                                  width=widths.get(wkey))
            if wkey not in widths:
                wmax = counts.max().view(1)
                widths[wkey] = shard.reduce_max(wmax) if shard is not None else wmax
            nb[l] = a.view(S, C, -1)
            sims[('N', 'out', l)] = w.view(S, C, -1).contiguous()
            D = hp['node_embed_size']
            if det and hp.get('fused_forward', True) and D % 4 == 0 and D <= 256 and (D // 4) & (D // 4 - 1) == 0:
                # the border layer's table gradient is a sorted scatter: its edge list and order are known here
                plan = ops.mpn_edge_plan(sims[('N', 'out', l)], nb[l], cc_ids._sgnn_mask_u8, R=S * C,
                                         A=nb[l].shape[-1], D=D, max_key=g.max_id, sims_per_edge=True)
                plan['anchors'] = nb[l]
                plans[('N', False, l)] = plan
        st.per_split['anchors_neigh_int'], st.per_split['anchors_neigh_border'] = ni, nb
        st.per_split['_mpn_edge_plans'] = plans
        t.mark('border_bfs+N_anchors')
    ci = ce = None
    if hp['use_structure'] and pool is None:
        ci, ce = ops.degree_sequence(g, cc_sets, sort=True, use_degree_dict=g.full_degree is not None,
                                     order=set_order)
        t.mark('degree_sequences')
    if hp['use_structure'] and side is not main:
        main.wait_event(patch_ev)
        _hand_over(main, structure_anchors)
        if walks_on_side:
            if pool is None:
                patch_degree_sequences()
            t.mark('S_patch_degree_sequences')
        else:
            structure_walks()
            t.mark('S_walks')
    # ---- structure similarities (main), then the join ---------------------------------------
    # The DTW launches read the degree sequences only, nothing the side stream computes: they are queued BEFORE the join, so a
    # position search that is still running (small shards: the search does not shrink with the shard, everything else does)
    # overlaps with them instead of delaying them (6 250 subgraphs: the main chain reaches this point after ~1.3 ms, the search
    # ends at ~1.6).  At the benchmark's 50k subgraphs the search has long finished when the main chain gets here.
    st.dtw_inputs = (cc_sets, ci, ce, a_sets, ai, ae, (S, C)) if (hp['use_structure'] and pool is None) else None
    if pool is not None and hp['use_structure']:
        st.attrs[split + '_int_struc_similarities'], st.attrs[split + '_bor_struc_similarities'] = pool['int_sims'], pool['bor_sims']
    if late_bfs:
        dtw_ev = torch.cuda.Event()
        dtw_ev.record(main)                                 # the DTW launches are queued right behind this ...
        with torch.cuda.stream(side):
            side.wait_event(dtw_ev)                         # ... and the position search starts with them
            position_block()
    if not defer_dtw:
        finish_pass(model, st, t)
    if side is not main:
        main.wait_stream(side)
        _hand_over(main, sims, st.attrs.get('anchors_pos_ext'), st.per_split.get('anchors_pos_int'),
                   st.attrs.get('structure_anchors'), st.attrs.get('int_structure_anchor_random_walks'),
                   st.attrs.get('bor_structure_anchor_random_walks'), st.sim_cols[1] if st.sim_cols else None)
        for v in (st.attrs.get('anchors_structure') or {}).values():
            _hand_over(main, v[0], v[2], v[3])
    if side is not main:
        t.mark('side_stream_join(S_patches,P_bfs)')
    st.attrs[split + '_neigh_pos_similarities'] = sims if sims else None
    st.attrs[split + '_N_border'] = None
    return st


def finish_pass(model, st, timer=None):
    """The structure similarities of a prepared pass (the DTW launches: the longest stage, and one that shares a CU
    with nothing -- it holds every vector register at three wavefronts of 168 registers per SIMD).  Separate from prepare_pass
    (``defer_dtw``) for a caller that wants to queue it elsewhere.  Measured for the pipeline: queued behind the
    training half of the previous pass (so that it cannot starve that half's small kernels) the step took 16.7 ms,
    queued freely 15.8 -- the sampling stages and the training kernels do not overlap as well as DTW and training do."""
    t = timer or StageTimer(False)
    split = st.split
    if st.dtw_inputs is not None:
        cc_sets, ci, ce, a_sets, ai, ae, (S, C) = st.dtw_inputs
        st.dtw_inputs = None
        mx, my = max(cc_sets.max_len, 1), max(a_sets.max_len, 1)
        tie = int(model.hparams['dtw_tie_order'])
        # Grouping repeated component sequences pays on the internal side (2.7k distinct rows among the
        # benchmark's 50k) and is pure overhead on the external side (nearly all distinct).  Which it is
        # depends only on the split's components and the graph: decided on the first pass, kept.
        group = model.__dict__.setdefault('_dtw_group_rows', {})
        xprep = model.__dict__.setdefault('_dtw_x_prep', {})
        ent = group.get(split)
        if ent is None or ent[0] != cc_sets.n:
            # the first pass of a split does not group (a choice of kernels, never of values) and does not WAIT for the answer
            # either: the distinct-row counts travel to pinned memory behind its launches (75 ms of read-back on the driver's
            # box in round 5) and a later pass picks them up
            ent = group[split] = (cc_sets.n, False, False,
                                  (ops.distinct_rows_async(cc_sets.ptr, ci, mx), ops.distinct_rows_async(cc_sets.ptr, ce, mx)))
            xprep[split] = ({}, {})
            t.mark('dtw_row_grouping_counts_queued(first pass only)')
        elif ent[3] is not None and not torch.cuda.is_current_stream_capturing():
            ent = _settle_dtw_grouping(model, split, wait=False)
        # the component side of the DTW calls (grouping of repeated degree sequences, processing order) depends on the
        # split's components only -- the same every pass: kept from the first one (xprep), like the dispatch orders
        st.attrs[split + '_int_struc_similarities'] = \
            ops.dtw_similarity(cc_sets.ptr, ci, mx, a_sets.ptr, ai, my, tie, dedupe=ent[1], x_prep=xprep[split][0]).view(S, C, -1)
        st.attrs[split + '_bor_struc_similarities'] = \
            ops.dtw_similarity(cc_sets.ptr, ce, mx, a_sets.ptr, ae, my, tie, dedupe=ent[2], x_prep=xprep[split][1]).view(S, C, -1)
        t.mark('dtw')
    elif split + '_int_struc_similarities' not in st.attrs:
        st.attrs[split + '_int_struc_similarities'] = None
        st.attrs[split + '_bor_struc_similarities'] = None
    return st


def _settle_dtw_grouping(model, split, wait):
    """Pick up the distinct-row counts the split's first pass sent to pinned memory and decide which DTW side groups repeated
    rows (a choice of kernels, never of values).  ``wait``: block until the copies have landed -- what a recording of the
    preparation does BEFORE its capture starts (an event wait inside a capture invalidates it)."""
    group = model.__dict__.setdefault('_dtw_group_rows', {})
    ent = group.get(split)
    if ent is None or ent[3] is None:
        return ent
    fr = [ops.distinct_rows_ready(p, wait=wait) for p in ent[3]]
    if all(f is not None for f in fr):
        ent = group[split] = (ent[0], fr[0] <= 0.5, fr[1] <= 0.5, None)
        model.__dict__.setdefault('_dtw_x_prep', {})[split] = ({}, {})
    return ent


def install_pass(model, st, timer=None):
    """Make a prepared pass the model's current one, then the table-dependent tail: the component embeddings are
    the first stage of a pass to read the embedding table -- a sharded optimizer's all-gather of the updated table
    (dist.ShardedTableAdam) travels under everything prepare_pass does."""
    t = timer or StageTimer(False)
    _verify_bfs(model, st)                       # before anything of the pass is visible (repeats a search if it must)
    for k, v in st.attrs.items():
        setattr(model, k, v)
    for k, v in st.per_split.items():
        d = getattr(model, k, None)
        if d is None:
            d = {}
            setattr(model, k, d)
        d[st.split] = v
    if st.sim_cols is not None:
        model.set_sim_cols(*st.sim_cols)
    model._build_sim_cols()
    if getattr(model, '_table_sync', None) is not None:
        model._table_sync()
        t.mark('table_all_gather_wait')
    model.init_all_embeddings(split=st.split, trainable=model.hparams['trainable_cc'], lazy=True)
    t.mark('cc_embed')
    model.__dict__['_sparse_prepared'] = True       # (a resample is then a new sparse pass: SubGNN._prepare_anchors_only)
    model._bump_generation()
    return t


def prepare_sparse(model, split='train', timer=None, shard=None):
    """HIP-only prepare_data for one split: prepare_pass + install_pass."""
    t = timer or StageTimer(False)
    install_pass(model, prepare_pass(model, split, t, shard), t)
    return t


class PassPipeline:
    """Passes in flight: while the model trains on pass k (forward, backward, optimizer on the caller's stream), the
    sampling + similarity half of pass k + 1 -- which reads neither the parameters nor anything pass k writes -- runs
    on a second HIP stream.  The small kernels of the training half leave most of the chip idle; the benchmark's
    prepare-then-train takes 11.3 ms back to back and 10.2 ms this way (round 3; the step is close to the device's
    total work: see DESIGN section 4, "What the pipeline is bound by").

        pipe = PassPipeline(model, 'train', shard)
        pipe.start()                      # pass 0 is being prepared
        for step in ...:
            pipe.install()                # wait for the prepared pass, make it the model's, component embeddings
            pipe.start()                  # the next pass's sampling stages start on the side stream ...
            loss = training step          # ... while this one trains
            optimizer step

    The anchors a pass draws depend on (seed, split, layer, resample epoch) only, so the pipelined schedule draws
    what the sequential one draws.  Collectives of the prepared pass (the width reductions of a sharded pass) must use
    their own communicator (``dist.Shard(collectives=...)``): they run concurrently with the gradient exchange."""

    def __init__(self, model, split='train', shard=None, pool_epochs=1):
        """``pool_epochs`` > 1: the structure-patch pool (patches, walks, DTW rows) is rebuilt every ``pool_epochs`` passes only
        and re-picked from in between (prepare_pass(pool=...)); the reference's own schedule with hparams['max_sim_epochs']."""
        self.model, self.split, self.shard = model, split, shard
        self.stream = torch.cuda.Stream()
        self.pending = collections.deque()         # (state, its stage timer, event: prepared) in preparation order
        self.timer = None                          # stage timer of the pass installed last
        self.pool_epochs, self._n_started, self._pool = max(1, int(pool_epochs)), 0, None

    @property
    def state(self):
        return self.pending[0][0] if self.pending else None

    def start(self, timed=False, after=None):
        """Queue the next pass's sampling + similarity stages on the side stream.  ``after``: an event on the caller's
        stream to start behind (the end of ``install``) instead of behind everything queued there so far -- for a caller
        that has already queued the training half (CapturedTraining), which the preparation does not depend on.
        May be called again before ``install``: the passes are prepared one after the other on the side stream and
        installed in that order -- with two in flight the side stream (the longer chain) never waits for the host."""
        main = torch.cuda.current_stream()
        if after is not None:
            self.stream.wait_event(after)
        else:
            self.stream.wait_stream(main)
        timer = StageTimer(timed)
        reuse = self._pool if (self.pool_epochs > 1 and self._n_started % self.pool_epochs != 0) else None
        with torch.cuda.stream(self.stream):
            state = prepare_pass(self.model, self.split, timer, self.shard, pool=reuse)
            if reuse is None and self.pool_epochs > 1 and self.model.hparams['use_structure']:
                self._pool = pool_of(state)
            self._n_started += 1
            timer.mark('prepared')
            done = torch.cuda.Event()
            done.record()
        self.pending.append((state, timer, done))

    def install(self, timer=None, installer=None):
        """``installer(state, timer)``: what makes the prepared pass the model's (default install_pass; CapturedTraining.install
        for a recorded training half)."""
        if not self.pending:
            raise RuntimeError('PassPipeline.install without a started pass')
        st, self.timer, done = self.pending.popleft()
        main = torch.cuda.current_stream()
        main.wait_event(done)
        _hand_over(main, *st.tensors())
        for v in (st.attrs.get('anchors_structure') or {}).values():
            _hand_over(main, v[0], v[2], v[3])
        if installer is not None:
            installer(st, timer)
            return timer
        return install_pass(self.model, st, timer)


_TENSOR_ATTRS = ('_sgnn_ids32', '_sgnn_sorted', '_sgnn_both', '_sgnn_all', '_sgnn_member_order', '_sgnn_mask', '_sgnn_mask_u8')


def _copy_into(dst, src, path, replaced, memo=None):
    """Copy the tensors of ``src`` (a prepared pass's nest of dicts / tuples / tensors) into the tensors of ``dst`` where
    shape, dtype and device agree -- the addresses a recorded training half reads stay valid -- and return what to keep
    at this place.  Anything that cannot be copied is replaced and reported in ``replaced``.  ``memo``: storage already
    copied in this call (a pass holds some tensors under several names)."""
    memo = {} if memo is None else memo
    if isinstance(src, torch.Tensor):
        if isinstance(dst, torch.Tensor) and dst.shape == src.shape and dst.dtype == src.dtype and dst.device == src.device:
            key = (src.data_ptr(), src.numel(), src.dtype, dst.data_ptr())
            if dst.data_ptr() != src.data_ptr() and key not in memo:
                pairs = memo.get('__pairs__')
                if pairs is not None and dst.is_contiguous() and src.is_contiguous():
                    pairs.append((dst, src))         # copied together at the end (install_pass_static: one launch per dtype)
                else:
                    dst.copy_(src)
                memo[key] = True
            for nm in _TENSOR_ATTRS:
                v = getattr(src, nm, None)
                if v is not None:
                    setattr(dst, nm, _copy_into(getattr(dst, nm, None), v, path + '.' + nm, replaced, memo))
            return dst
        replaced.append(path)
        return src
    if isinstance(src, dict):
        if not isinstance(dst, dict):
            replaced.append(path)
            return src
        for k, v in src.items():
            dst[k] = _copy_into(dst.get(k), v, '%s[%r]' % (path, k), replaced, memo)
        return dst
    if isinstance(src, (list, tuple)):
        if not isinstance(dst, (list, tuple)) or len(dst) != len(src):
            replaced.append(path)
            return src
        return type(src)(_copy_into(d, v, '%s[%d]' % (path, i), replaced, memo) for i, (d, v) in enumerate(zip(dst, src)))
    return src                                   # ZeroSims, numbers, None


# what a recorded training half never reads through a kernel (shapes only): may change shape between passes
_SHAPE_ONLY = ('structure_anchors', 'anchors_structure', 'int_structure_anchor_random_walks', 'bor_structure_anchor_random_walks')


def install_pass_static(model, st, timer=None):
    """install_pass for a model whose training half is replayed from a hipGraph: the pass's tensors are COPIED into the
    tensors of the pass that was installed when the graph was recorded (same shapes from pass to pass: the split's
    subgraphs and the anchor counts do not change), so every address the recording reads stays valid.  Returns the
    list of places where that was not possible (a shape changed): the caller records again."""
    t = timer or StageTimer(False)
    _verify_bfs(model, st)
    # the ~40 tensors of a pass are copied by ONE multi-tensor launch per dtype (torch._foreach_copy_) instead of one small
    # launch each: at shard size the copies were 0.67 ms of a 3.5 ms pass, all of it launch overhead
    replaced, memo = [], {'__pairs__': []}
    for k, v in st.attrs.items():
        setattr(model, k, _copy_into(getattr(model, k, None), v, k, replaced, memo))
    for k, v in st.per_split.items():
        d = getattr(model, k, None)
        if d is None:
            d = {}
            setattr(model, k, d)
        d[st.split] = _copy_into(d.get(st.split), v, '%s[%s]' % (k, st.split), replaced, memo)
    if st.sim_cols is not None:
        cur = model.__dict__.get('_sim_cols_static') or getattr(model, '_sim_col_cache', None)
        cols = _copy_into(cur, st.sim_cols[1], '_sim_cols', replaced, memo)
        model.__dict__['_sim_cols_static'] = cols
        model.set_sim_cols(model.anchors_structure, cols)
    by_type = {}
    for d_, s_ in memo['__pairs__']:
        by_type.setdefault((d_.dtype, d_.device), ([], []))
        by_type[(d_.dtype, d_.device)][0].append(d_.view(-1))
        by_type[(d_.dtype, d_.device)][1].append(s_.view(-1))
    for dsts, srcs in by_type.values():
        torch._foreach_copy_(dsts, srcs)
    model._build_sim_cols()
    model.init_all_embeddings(split=st.split, trainable=model.hparams['trainable_cc'], lazy=True)
    t.mark('install_copies')
    model.__dict__['_sparse_prepared'] = True
    model._bump_generation()
    return [r for r in replaced if not r.startswith(_SHAPE_ONLY) or '_sgnn_both' in r or '_sgnn_all' in r]


class CapturedTraining:
    """The training half of a pass -- component embeddings, the three channels, head, loss, backward, gradient clipping,
    Adam (SubGNN.training_step -> backward -> optim.ClipAdam.step) -- recorded once into a hipGraph and replayed per pass.
    ~150 launches become one: the host needs ~0.1 ms instead of ~4 to queue them, so that under PassPipeline the training
    kernels of pass k are on the device BEFORE the host starts queueing the preparation of pass k + 1 (queued behind the
    preparation they started 4 ms late and ran into the DTW launch, which shares a CU with nothing).

        trainer = CapturedTraining(model, ClipAdam(..., capturable=True))
        per pass:  trainer.install(prepared_state);  loss, acc = trainer.step()

    ``install`` keeps the recording's addresses valid (install_pass_static); a shape change records again.  The first
    ``warmup`` steps run eagerly (lazy initialisations must not land in the recording).  Same kernels, same order, same
    arithmetic as the eager step: losses and parameters are bit-equal (tests/test_gpu_hotpath.py)."""

    def __init__(self, model, optimizer, split='train', warmup=2):
        """``optimizer`` None: the recording ends with the backward pass -- the data-parallel form, whose gradient exchange and
        (sharded) optimizer run eagerly behind every replay: collectives are not recorded.  The gradients are then the recording's
        STATIC tensors: ``step`` re-attaches them to the parameters after every replay (``p.grad = ...``), so the caller may
        set them to None when it is done with them -- and must, before the next step."""
        if optimizer is not None and not getattr(optimizer, 'capturable', False):
            raise ValueError('CapturedTraining needs ClipAdam(capturable=True): a host step count cannot be replayed')
        if optimizer is not None and getattr(model, '_table_sync', None) is not None:
            raise ValueError('a recording that contains the optimizer is the single-rank form (collectives are not recorded)')
        if optimizer is None and model.hparams.get('dp_gather_embeddings', False):
            raise ValueError('the replicated head gathers embeddings inside forward: not recordable')
        self.model, self.opt, self.split = model, optimizer, split
        self.static_grads = None
        self.graph, self.loss, self.acc = None, None, None
        self._warm_left = int(warmup)
        self._installed = False
        self.recordings = 0
        self.last_changed = None

    def install(self, st, timer=None):
        if not self._installed or self.graph is None:
            install_pass(self.model, st, timer)
            self._installed = True
            return
        changed = install_pass_static(self.model, st, timer)
        if changed:
            self.graph = None                    # a tensor the recording reads was replaced: record again
            self.last_changed = changed

    def _body(self):
        m = self.model
        out = m.training_step(full_split_batch(m, self.split), 0)
        m.backward(None, out['loss'], None, 0)
        if self.opt is not None:
            self.opt.step()
            self.opt.zero_grad(set_to_none=True)
        return out['loss'].detach(), out['log']['train_acc'].detach()

    def step(self):
        """-> (loss, accuracy): the recording's static outputs once it exists (clone to keep past the next step)."""
        sync = getattr(self.model, '_table_sync', None)
        if sync is not None:
            sync()                               # (a sharded optimizer's all-gather of the table: the training half reads it)
        if self.opt is None and any(p.grad is not None for p in self.model.parameters()):
            raise RuntimeError('CapturedTraining without an optimizer: the caller must set the gradients to None after its update')
        if self.graph is None:
            if self._warm_left > 0:
                self._warm_left -= 1
                return self._body()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    self.loss, self.acc = self._body()
            except Exception:
                from .graph_step import abandon_capture
                self.loss = self.acc = None
                abandon_capture(self.model, self.opt)      # (buffers of the dead capture must not reach an eager step)
                raise
            self.graph = g
            self.recordings += 1
            if self.opt is None:
                self.static_grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]
        self.graph.replay()
        if self.static_grads is not None:
            for p, gr in self.static_grads:
                p.grad = gr
        self.model.invalidate_half_table()
        return self.loss, self.acc


class GraphedPasses:
    """BOTH halves of a pass replayed from hipGraphs, two slots: the sampling + similarity half (prepare_pass: ~130 launches
    on two streams) recorded into one graph that writes the pass's tensors at FIXED addresses, the training half
    (CapturedTraining's body: ~135 launches) recorded into a second graph that reads exactly those addresses -- nothing is
    installed or copied between them (CapturedTraining alone copies every pass into its recording's tensors: 0.7 ms of a 3.5 ms
    pass at shard size).  Two such pairs alternate, so that slot B's preparation replays on a second stream while slot A
    trains:

        passes = GraphedPasses(model, ClipAdam(..., capturable=True))
        per pass:  loss, acc = passes.step()

    A step queues two graph launches: at the strong-scaling shard size (6 250 subgraphs, where the pass is bound by the
    host's ~265 launches) 4.7 ms eager -> 3.5 ms with the training half recorded -> see DESIGN 4 for this form.  Same kernels,
    same order, same arithmetic as prepare_pass + install_pass + the eager step: losses and parameters are bit-equal
    (tests/test_gpu_hotpath.py).  What a recording cannot contain -- a host round trip (torch.unique of the P-internal
    anchors of multi-component subgraphs), collectives -- raises at record time; the caller falls back to PassPipeline.
    The hinted position-channel searches are verified after every replay (status in pinned memory); one that ran out of
    levels raises the hint and both slots are recorded again."""

    class _Slot:
        def __init__(self):
            self.prep = self.train = self.state = self.loss = self.acc = None
            self.prep_done, self.train_done = torch.cuda.Event(), torch.cuda.Event()
            self.checks = []
            self.prepared = False

    def __init__(self, model, optimizer, split='train', warmup=2):
        if not getattr(optimizer, 'capturable', False):
            raise ValueError('GraphedPasses needs ClipAdam(capturable=True): a host step count cannot be replayed')
        if getattr(model, '_table_sync', None) is not None:
            raise ValueError('GraphedPasses is the single-rank form (collectives are not recorded)')
        self.model, self.opt, self.split = model, optimizer, split
        self.slots = [None, None]
        self.stream = torch.cuda.Stream()
        self.k = 0
        if int(warmup) < 2:
            raise ValueError('GraphedPasses needs two eager passes before it records (the second one runs the kernels the first one chose)')
        self._warm_left = int(warmup)
        self.recordings = 0

    def _body(self):
        m = self.model
        out = m.training_step(full_split_batch(m, self.split), 0)
        m.backward(None, out['loss'], None, 0)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        return out['loss'].detach(), out['log']['train_acc'].detach()

    def _record(self, i):
        """Slot i: record its preparation, run it once (a capture executes nothing), make its tensors the model's, record the
        training half on them."""
        slot = self._Slot()
        pool = self.model.__dict__.setdefault('_bfs_status_pool', [])
        while len(pool) < 4 * max(1, int(self.model.hparams['n_layers'])):      # pinned buffers the recorded searches will take
            pool.append(torch.empty(4, dtype=torch.int32).pin_memory())
        torch.cuda.synchronize()
        _settle_dtw_grouping(self.model, self.split, wait=True)     # (the recording keeps whatever kernels this decides on)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            slot.state = prepare_pass(self.model, self.split)
        slot.prep = g
        slot.checks = list(slot.state.bfs_checks)
        g.replay()
        torch.cuda.synchronize()
        try:
            install_pass(self.model, slot.state)             # (verifies the searches of the replay above)
        except BfsLevelsExhausted:
            release_checks(self.model, slot.checks)
            return self._record(i)                           # (the hint is the cap now: this happens at most once)
        slot.state.bfs_checks = list(slot.checks)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            slot.loss, slot.acc = self._body()
        slot.train = g2
        slot.prepared = True                                  # the replay above IS this slot's next pass
        slot.prep_done.record()
        self.slots[i] = slot
        self.recordings += 1
        return slot

    def step(self):
        """One pass -> (loss, accuracy): the recording's static outputs (clone to keep past the slot's next use)."""
        if self._warm_left > 0:
            # eager passes first: per-split caches, BFS depth hints, lazy initialisations and the allocator's growth must not
            # land in a recording
            self._warm_left -= 1
            install_pass(self.model, prepare_pass(self.model, self.split))
            out = self._body()
            # the first pass of a split sends its distinct-row counts to pinned memory and a LATER pass picks the DTW kernels'
            # row grouping from them: settled here, so that the next warm-up pass builds the kept x-side preparation of the
            # decided form eagerly (its first call indexes with a mask: not recordable)
            _settle_dtw_grouping(self.model, self.split, wait=True)
            return out
        i = self.k & 1
        main = torch.cuda.current_stream()
        slot = self.slots[i] or self._record(i)
        if not slot.prepared:                                  # (only the very first use of slot 1: nothing was queued for it yet)
            self._queue_prep(slot, main)
        main.wait_event(slot.prep_done)
        slot.prep_done.synchronize()                           # prepared a step ago: no wait in steady state
        slot.state.bfs_checks = list(slot.checks)
        try:
            _verify_bfs(self.model, slot.state, keep=True)
        except BfsLevelsExhausted:
            torch.cuda.synchronize()
            for sl in self.slots:                              # the dropped recordings' pinned status buffers return to the pool
                if sl is not None:
                    release_checks(self.model, sl.checks)
            self.slots = [None, None]                          # deeper searches from now on: record again
            return self.step()
        slot.train.replay()
        slot.train_done.record(main)
        slot.prepared = False
        self.model.invalidate_half_table()
        other = self.slots[1 - i]
        if other is not None:                                  # its preparation, beside this pass's training
            self._queue_prep(other, main)
        self.k += 1
        return slot.loss, slot.acc

    def _queue_prep(self, slot, main):
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(slot.train_done)            # the slot's tensors are still being trained on until then
            slot.prep.replay()
            slot.prep_done.record(self.stream)
        slot.prepared = True


def _device_labels(model, split):
    """The split's labels on the device, uploaded once (a pageable host->device copy blocks the host
    until the stream reaches it -- behind the DTW launches that is the whole DTW)."""
    src = getattr(model, split + '_sub_G_label')
    cache = model.__dict__.setdefault('_device_label_cache', {})
    if split not in cache or cache[split][0] is not src:
        cache[split] = (src, src.to(model.device))
    return cache[split][1]


def _whole_split_index(model, S):
    """arange(S) as the (S, 1) ``subgraph_idx`` of a batch, kept, and marked as the identity selection."""
    memo = model.__dict__.setdefault('_whole_split_idx', {})
    if S not in memo:
        idx = torch.arange(S, device=model.device).view(-1, 1)
        idx._sgnn_identity = S
        memo[S] = idx
    return memo[S]


def full_split_batch(model, split):
    """The whole split as one batch (the large-shard launch shape of the benchmark)."""
    S = getattr(model, split + '_cc_ids').shape[0]
    dev = model.device
    return {'subgraph_ids': None, 'cc_ids': getattr(model, split + '_cc_ids'),
            'N_border': None, 'NP_sim': getattr(model, split + '_neigh_pos_similarities'),
            'I_S_sim': getattr(model, split + '_int_struc_similarities'),
            'B_S_sim': getattr(model, split + '_bor_struc_similarities'),
            'subgraph_idx': _whole_split_index(model, S),
            'label': _device_labels(model, split)}
