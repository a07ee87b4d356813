/*
 * subgnn_hip.h -- C ABI of libsubgnn_hip.so: the MI355X (gfx950) kernels behind SubGNN's
 * anchor-patch sampling + three-channel subgraph message-passing hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (torch tensors on the host
 *     side); nothing is allocated or freed inside; every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*), graph-capturable, and stateless: the library keeps no settings between
 *     calls (where two kernels compute the same values, the choice is an argument of the call: `kernel`, `pull_alpha`);
 *   - return value: 0 = launched, negative = SGNN_ERR_* (argument errors are detected on the
 *     host before any launch; nothing throws across the ABI);
 *   - node ids are 1-based, 0 = PAD (reference config.py:9, SubGNN/SubGNN.py:554-559);
 *   - the base graph is CSR indexed by node id: rowptr int64[max_id + 2] (row 0 = PAD, empty),
 *     col int32[nnz]; `col` keeps networkx neighbour order (needed by the walks),
 *     `col_sorted` is the same rows sorted ascending (adjacency tests by binary search);
 *   - ragged sets (connected components, anchor patches, subgraphs) are
 *     set_ptr int64[n_sets + 1] + set_nodes int32[set_ptr[n_sets]], PAD already stripped,
 *     order preserved, duplicates kept -- never the reference's dense zero padding;
 *   - randomness is the counter-based draw tape draw64(seed, stream, item, j) (DESIGN.md,
 *     "Draw tape"; oracle twin: oracle/tape.py).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference root, mims-harvard/SubGNN).
 */
#ifndef SUBGNN_HIP_H
#define SUBGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGNN_OK                  0
#define SGNN_ERR_BAD_ARG        -1
#define SGNN_ERR_SET_TOO_LARGE  -2   /* a set exceeds the size class a kernel supports */
#define SGNN_ERR_NNZ_TOO_LARGE  -3   /* nnz >= 2^31 (kernels index col with 32 bits) */
#define SGNN_ERR_LAUNCH         -4   /* hipGetLastError() after a launch */
#define SGNN_ERR_UNSUPPORTED_D  -5   /* embedding width not supported by the vector path */

#define SGNN_ABI_VERSION 11
int sgnn_abi_version(void);
/* Load the code objects of every translation unit of the library on the current device (one empty launch each on ``stream``):
 * what the first call of each kernel family would otherwise pay, 5-25 ms at a time, inside the reference's one-time
 * prepare_data (SubGNN/SubGNN.py:1024-1063).  Idempotent and cheap when they are loaded.  0 = ok. */
int sgnn_warm_up(void* stream);
/* last hip error string for SGNN_ERR_LAUNCH (static storage) */
const char* sgnn_last_error(void);

/* ---------------------------------------------------------------------------------------
 * a10  Structure-channel CSR gather: degree sequences of node sets.
 * Replaces gamma.get_degree_sequence (SubGNN/gamma.py:21-49) as called for every anchor patch
 * and every CC row at SubGNN/SubGNN.py:802-809.
 *   internal[i] = #neighbours of set_nodes[i] inside its set (self loop counts 2, networkx)
 *   external[i] = full_degree[id] - internal[i]; full_degree == NULL -> degree from the CSR
 * One entry per listed node (duplicates kept); if `sorted`, each set's entries are written
 * in ascending order (gamma.py:35,48).  out_external may be NULL.
 * self_loops (nullable): uint8[max_id + 1], number of self-loop entries in each node's CSR row;
 * when given, the kernel does not have to test every streamed neighbour against the row's owner.
 * max_set_size: upper bound on set length known to the caller (<= 64 selects the wavefront-per-set kernel; a set
 * that breaks the promise gets INT32_MIN in all its outputs, never stale memory).
 * set_order (nullable): int32[n_sets], a permutation: the order in which sets are handed to the
 * hardware dispatcher (results are unaffected); heaviest-first shortens the tail of the launch.
 * ------------------------------------------------------------------------------------- */
int sgnn_degree_sequence(const int64_t* rowptr, const int32_t* col, int64_t nnz,
                         const int32_t* full_degree, const uint8_t* self_loops,
                         const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                         int64_t max_set_size, int sorted,
                         int32_t* out_internal, int32_t* out_external, const int32_t* set_order, void* stream);
/* Same, for callers that also hold the CSR with every row's neighbour ids in ascending order
 * (col_sorted, same rowptr; a simple graph: no id twice in a row).  Lists of >= 512 entries are then
 * not streamed at all: every member of the set binary-searches its id in the list (log2(deg)
 * dependent loads for the whole set instead of deg/64 wave loads and table probes) -- on scale-free
 * graphs the hub lists carry most of the bytes.  Results are identical. */
int sgnn_degree_sequence_sorted_rows(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted,
                                     int64_t nnz, const int32_t* full_degree, const uint8_t* self_loops,
                                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                     int64_t max_set_size, int sorted, int32_t* out_internal,
                                     int32_t* out_external, const int32_t* set_order, void* stream);
/* Same, with membership BITMAPS for the long lists (built once per graph by the caller): hub_index[v] >= 0 numbers the
 * lists that have one, -1 = no bitmap for v; hub_bits holds one row of hub_words 32-bit words per NODE ID x (rows 0 ..
 * max id), bit hub_index[v] of row x = "x is in v's neighbour list" (laid out by node: one member's lookups against all
 * the hubs of its set share a cache line).  node_info (may be NULL; 16-byte aligned): one record of four int32 per node id
 * -- {row start, degree, hub number (0xffffff = none) | self-loop entries << 24, full degree} -- read INSTEAD of one line
 * each out of rowptr, hub_index, self_loops and full_degree (info_degree != 0: external = the record's full degree - internal,
 * else degree + self loops - internal).  A list of >= sgnn_degree_sequence_search_threshold() entries that has a bitmap is neither streamed
 * nor searched: every member of the set reads its one bit (gamma.get_degree_sequence, gamma.py:21-49: the membership test
 * `w in subgraph` for a hub's neighbours, asked from the set's side).  Such lists WITHOUT a bitmap are searched when
 * col_sorted is given (may be NULL), streamed otherwise.  Results are identical. */
int sgnn_degree_sequence_hub_bitmaps(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted,
                                     int64_t nnz, const int32_t* full_degree, const uint8_t* self_loops,
                                     const int32_t* hub_index, const uint32_t* hub_bits, int64_t hub_words,
                                     const int32_t* node_info, int info_degree,
                                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                     int64_t max_set_size, int sorted, int32_t* out_internal,
                                     int32_t* out_external, const int32_t* set_order, void* stream);
int64_t sgnn_degree_sequence_search_threshold(void);
/* Sets of more than 2048 entries (components of subgraphs with thousands of nodes): the calls above leave them alone;
 * this one, issued after either on the same stream, writes their degrees UNSORTED (same counting rules; the membership
 * table lives in the workspace, sgnn_degree_sequence_huge_workspace_bytes(set_ptr[n_sets])); ordering such a set's slice
 * is the caller's (any segment sort). */
int64_t sgnn_degree_sequence_huge_workspace_bytes(int64_t total_entries);
int sgnn_degree_sequence_huge(const int64_t* rowptr, const int32_t* col, int64_t nnz, const int32_t* full_degree,
                              const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                              int64_t total_entries, int32_t* out_internal, int32_t* out_external,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a7  Connected components of induced subgraphs.
 * Replaces nx.connected_components(nx.subgraph(G, ids)) at SubGNN/SubGNN.py:589-592.
 * out_label[i] = smallest position (within its subgraph) of a node in the same component as
 * position i; duplicates of a node share a label.  Needs col_sorted.
 * ------------------------------------------------------------------------------------- */
int sgnn_cc_labels(const int64_t* rowptr, const int32_t* col_sorted, int64_t nnz,
                   const int64_t* sub_ptr, const int32_t* sub_nodes, int64_t n_subgraphs,
                   int64_t max_sub_len /* longest subgraph, 0 = unknown; subgraphs > 2048 nodes get -1: sgnn_cc_labels_huge */,
                   int32_t* out_label, void* stream);
/* Canonical order inside every set: ids ascending, equal ids in their original relative order.
 * The neighbourhood-anchor draw ranks "the ascending members" of a component / border set where the
 * reference walks a python set (SubGNN/anchor_patch_samplers.py:60-75, sample_neighborhood_anchor_patch);
 * this is the order both sides agree on.  out_pos (nullable): flat index the id came from, for
 * payloads that travel with the ids (hop labels).  max_set_size <= 1024, else SGNN_ERR_SET_TOO_LARGE
 * (the caller sorts (set, id) keys device-wide instead).  out_nodes must not alias set_nodes. */
int sgnn_sort_sets(const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int64_t max_set_size,
                   int32_t* out_nodes, int32_t* out_pos, void* stream);
/* labels -> the padded component tensor of SubGNN.initialize_cc_ids (SubGNN/SubGNN.py:575-607) in
 * canonical order: components by the position of their first node, nodes in subgraph order,
 * duplicates dropped, PAD = 0.  Two steps: _stats gives, per subgraph, the number of components and
 * the longest one (the caller takes the maxima C and L and zero-fills out (n_subgraphs, C, L) int64);
 * sgnn_cc_compact writes the ids.  max_sub_len: longest subgraph (0 = unknown); subgraphs of more than 2048 nodes are
 * skipped here and served by sgnn_cc_compact_huge below. */
int sgnn_cc_compact_stats(const int64_t* sub_ptr, const int32_t* sub_nodes, const int32_t* labels,
                          int64_t n_subgraphs, int64_t max_sub_len, int32_t* out_n_components,
                          int32_t* out_longest, void* stream);
int sgnn_cc_compact(const int64_t* sub_ptr, const int32_t* sub_nodes, const int32_t* labels,
                    int64_t n_subgraphs, int64_t max_sub_len, int64_t C, int64_t L, int64_t* out, void* stream);
/* The same three steps for subgraphs of MORE than 2048 nodes (the reference pads to any size, SubGNN/SubGNN.py:575-607):
 * the calls above leave such subgraphs alone (labels -1, no statistics, no rows) and these fill them in -- call them
 * after their counterpart on the same stream.  State lives in the caller's workspace
 * (sgnn_cc_huge_workspace_bytes(total_nodes), total_nodes = sub_ptr[n_subgraphs]; any content): id -> first position
 * table, union-find parents, ranks, counters, each subgraph at its own offset.  sgnn_cc_compact_huge: write == 0 gives
 * the statistics (out_n_components / out_longest), write != 0 the rows of out (n_subgraphs, C, L). */
int64_t sgnn_cc_huge_workspace_bytes(int64_t total_nodes);
int sgnn_cc_labels_huge(const int64_t* rowptr, const int32_t* col, int64_t nnz, const int64_t* sub_ptr,
                        const int32_t* sub_nodes, int64_t n_subgraphs, int64_t total_nodes, int32_t* out_label,
                        void* workspace, int64_t workspace_bytes, void* stream);
int sgnn_cc_compact_huge(const int64_t* sub_ptr, const int32_t* sub_nodes, const int32_t* labels,
                         int64_t n_subgraphs, int64_t total_nodes, int write, int64_t C, int64_t L,
                         int32_t* out_n_components, int32_t* out_longest, int64_t* out, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a8  k-hop border of a component, and the hop level of each border node.
 * Replaces subgraph_utils.get_component_border_neighborhood_set (SubGNN/subgraph_utils.py:
 * 146-176) / SubGNN.initialize_border_sets (SubGNN/SubGNN.py:673-700).
 * Two-pass protocol: call with out_nodes == NULL to get out_count[s]; prefix-sum on the host
 * into out_ptr; call again with out_ptr/out_nodes (and optionally out_hop) to fill.  Entries
 * of a set come in discovery order (not sorted).
 * ego_dict_mode != 0 reproduces the `ego_graphs.txt` path (su:168-174): 1 hop only and every
 * border id is (true id - 1), differenced against the 1-based component (id 0 can appear).
 * bitmap_in_lds != 0 keeps the visited bitmap in the CU's LDS (allowed iff
 * sgnn_khop_border_bitmap_fits_lds(max_id)); the results are the same either way.
 * workspace: sgnn_khop_border_workspace_bytes(max_id, n_sets, bitmap_in_lds) bytes: per-workgroup
 * BFS queues and, for the global variant, the visited bitmaps, which must be ZERO on entry (they
 * are left zeroed, so one buffer can be reused across calls); the LDS variant needs no
 * initialisation.
 * ------------------------------------------------------------------------------------- */
int64_t sgnn_khop_border_workspace_bytes(int64_t max_id, int64_t n_sets, int bitmap_in_lds);
int sgnn_khop_border_bitmap_fits_lds(int64_t max_id);
int sgnn_khop_border(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                     int k, int ego_dict_mode,
                     int64_t* out_count, const int64_t* out_ptr, int32_t* out_nodes, uint8_t* out_hop,
                     void* workspace, int64_t workspace_bytes, int bitmap_in_lds, void* stream);
/* one-pass materialisation: slice s of `arena` (starting at arena_off[s], guaranteed by the caller to
 * hold the border: for k = 1, sum of the members' degrees always does) is used as the BFS queue, so
 * the border is written by the BFS itself; out_count[s] entries are valid, in discovery order. */
int sgnn_khop_border_arena(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                           const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int k,
                           const int64_t* arena_off, int32_t* arena, int64_t* out_count,
                           void* workspace, int64_t workspace_bytes, int bitmap_in_lds, void* stream);
/* a8 + a4 fused: k-hop border BFS and, without materialising the border, the neighbourhood-border
 * anchor draw of anchor_patch_samplers.sample_neighborhood_anchor_patch(sample_inside=False)
 * (anchor_patch_samplers.py:184-194) over it, under the neighbourhood-anchor law stated at
 * sgnn_sample_anchors_padded.  For set s and slot i (tape item (item_base + s)*n_slots+i; item_base = the
 * number of the first set when the sets are rows of a larger, sharded matrix): out_anchor = the k-th
 * smallest border id (a rank query on the visited bitmap -- no sort, no per-node hashing), out_hop =
 * its hop level (the N-border similarity, = the APSP row-min of SubGNN.py:772 on that column),
 * out_allneg = the item's "every variate negative" draw -- the caller applies the PAD rule of
 * aps:190 (PAD wins when it is set AND the padded row is longer than this border, i.e.
 * out_count[s] < max count).  An empty border yields anchor 0.
 * col_sorted (nullable): the rows in ascending order.  k = 1 takes a specialised kernel (bits ORed without
 * return value, a 16-lane group per slot); with col_sorted it also serves id ranges beyond the LDS bitmap
 * (~1.2 M ids) by processing the range in slices -- without it such graphs keep the bitmap in `workspace`.
 * bitmap_in_lds: 0 = bitmap in workspace, 1 = in LDS, > 1 = in LDS using at most this many bytes (smaller
 * slices; a test hook that needs no global state).  workspace: sgnn_khop_border_sample_workspace_bytes. */
int64_t sgnn_khop_border_sample_workspace_bytes(int64_t max_id, int64_t n_sets, int k, int rows_sorted, int bitmap_in_lds);
int sgnn_khop_border_sample(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz, int64_t max_id,
                            const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int k,
                            int64_t n_slots, uint64_t seed, uint64_t stream_id, int64_t item_base,
                            int64_t* out_anchor, uint8_t* out_hop, uint8_t* out_allneg, int64_t* out_count,
                            const int32_t* set_order /* nullable: dispatch order of the sets, a permutation */,
                            void* workspace, int64_t workspace_bytes, int bitmap_in_lds, void* stream);
/* The PAD rule of anchor_patch_samplers.py:189-191 on the drawn anchors, in place: slot := PAD where out_allneg is set and the
 * set's border is smaller than the padded matrix's width (width: device scalar, = max out_count, MAX-reduced over ranks when
 * the sets are one shard); sims (n_sets, n_slots) float32 := hop level, 0 on PAD. */
int sgnn_khop_sample_finish(int64_t* anchor, const uint8_t* hop, const uint8_t* allneg, const int64_t* counts,
                            const int64_t* width, int64_t n_sets, int64_t n_slots, float* sims, void* stream);

/* Padded id rows (n_rows, row_len) int64 -> ragged sets: PAD (0) entries stripped -- or, with mask (uint8, same shape), the
 * entries whose mask is 0 -- order kept (how gamma.py:27, anchor_patch_samplers.py:131 and SubGNN.py:769 strip PAD).
 * sgnn_pack_rows_count writes the rows' kept counts; the caller's exclusive prefix sum of them is ptr (n_rows + 1);
 * sgnn_pack_rows_write writes the kept ids (as int32) behind ptr[row]. */
/* keep (n_rows, row_len) uint8: 1 where ids[r, i] != PAD and no earlier entry of the row holds the same id -- the node view of
 * a patch (unique nodes of a walk, first occurrence first: anchor_patch_samplers.py:131-138) as a mask for sgnn_pack_rows_*. */
int sgnn_first_occurrence_mask(const int64_t* ids, int64_t n_rows, int64_t row_len, uint8_t* keep, void* stream);
/* The flagged entries (flags uint8, aligned with set_nodes) of every ragged set, order kept -- a patch's in-border nodes
 * (subgraph_utils.py:126-144 via sgnn_patch_in_border).  counts != NULL: write the sets' flagged counts; counts == NULL: write
 * the flagged ids behind out_ptr[set] (the caller's exclusive prefix sum of the counts). */
int sgnn_filter_sets(const int64_t* set_ptr, const int32_t* set_nodes, const uint8_t* flags, int64_t n_sets, int64_t* counts,
                     const int64_t* out_ptr, int32_t* out_nodes, void* stream);
int sgnn_pack_rows_count(const int64_t* ids, const uint8_t* mask, int64_t n_rows, int64_t row_len, int64_t* counts,
                         void* stream);
int sgnn_pack_rows_write(const int64_t* ids, const uint8_t* mask, int64_t n_rows, int64_t row_len, const int64_t* ptr,
                         int32_t* nodes, void* stream);
/* The same packings in ONE launch for up to sgnn_pack_fused_max_rows() rows / sets (the few hundred structure patches of a pass:
 * anchor_patch_samplers.py:131-138 node views, subgraph_utils.py:126-144 in-border sets, gamma.py:27 PAD stripping): counts,
 * prefix sum, packed write and the zeroed tail of the arena by one workgroup.  mode 0: keep non-PAD ids; 1: keep where
 * mask != 0; 2: keep non-PAD ids that no earlier entry of the row repeats (first occurrence).  ptr: n_rows + 1 entries
 * (all written); nodes: an arena of n_rows * row_len + 1 entries (all written).  sgnn_filter_sets_fused: out_nodes is an arena
 * of arena_entries entries (>= the flagged total), all written. */
int64_t sgnn_pack_fused_max_rows(void);
int64_t sgnn_pack_fused_max_entries(void);   /* n_rows * row_len (the sets' arena for the filter) may not exceed this: entries are staged in LDS */
int sgnn_pack_rows_fused(const int64_t* ids, const uint8_t* mask, int mode, int64_t n_rows, int64_t row_len, int64_t* ptr,
                         int32_t* nodes, void* stream);
int sgnn_filter_sets_fused(const int64_t* set_ptr, const int32_t* set_nodes, const uint8_t* flags, int64_t n_sets,
                           int64_t arena_entries, int64_t* out_ptr, int32_t* out_nodes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a4  Neighbourhood anchor sampling from padded id matrices.
 * Replaces anchor_patch_samplers.sample_neighborhood_anchor_patch (anchor_patch_samplers.py:
 * 163-198).  The reference draws one N(0,1) variate per column, zeroes the PAD columns and takes
 * the argmax (aps:177-179,189-191); in law: each non-PAD entry equally likely, except that PAD wins
 * when all n real variates are negative (probability 2^-n) and the row has a PAD column.  The
 * tape's neighbourhood-anchor law states exactly that with two draws of item r*n_slots+i:
 *   draw 0: index (u32 * n) >> 32 into the row's non-PAD entries in ASCENDING id order;
 *   draw 1: "every variate negative" iff n <= 32 and the top n bits of u32 are zero.
 * ids: (n_rows, L) int64 in CANONICAL form -- non-PAD entries ascending, PADs (0) last;
 * out (n_rows, n_slots).  (subgnn_amd.ops canonicalises arbitrary rows before the call.)
 * ------------------------------------------------------------------------------------- */
int sgnn_sample_anchors_padded(const int64_t* ids, int64_t n_rows, int64_t L, int64_t n_slots,
                               uint64_t seed, uint64_t stream_id, int64_t* out, void* stream);
/* same law on ragged sets, each ascending; row_has_pad[r] (nullable = all 1) says whether the
 * padded row would hold a PAD.  item_base: number of the first set within the whole (possibly
 * sharded) matrix -- set r draws as tape item (item_base + r)*n_slots + slot, so a shard of the rows
 * reproduces the draws of the unsharded call. */
int sgnn_sample_anchors_ragged(const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                               const uint8_t* row_has_pad, int64_t n_slots,
                               uint64_t seed, uint64_t stream_id, int64_t item_base, int64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * a5/a6  Uniform draws with replacement from a list (position anchors, structure picks).
 * Replaces np.random.choice(seq, n, replace=True) at anchor_patch_samplers.py:206,208,326.
 * list r draws as tape item item_base + r from seq[ptr[r] .. ptr[r+1]); out (n_items, n_draws) int64.
 * ------------------------------------------------------------------------------------- */
int sgnn_choice_ragged(const int64_t* ptr, const int32_t* seq, int64_t n_items, int64_t n_draws,
                       uint64_t seed, uint64_t stream_id, int64_t item_base, int64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * a1-a3  Triangular random walks.
 * Replaces anchor_patch_samplers.triangular_random_walk / perform_random_walks /
 * sample_structure_anchor_patches (anchor_patch_samplers.py:20-158, 210-243).
 *   mode 0  'graph' : start = uniform over node_order (aps:70 with the whole graph)
 *   mode 1  'inside': walk inside the induced subgraph of patch p = item / walks_per_patch;
 *                     start uniform over the patch's node view  patch_ptr/patch_nodes (unique)
 *   mode 2  'border': start uniform over in_border nodes (inb_ptr/inb_nodes); neighbours
 *                     restricted to in_border U (V \ patch) (aps:141-143)
 * out: (n_items, walk_len) int64, PAD filled.  One tape item per walk (item = item_base + walk index: a rank that runs a
 * share of a launch's walks -- with the patches of that share only, for modes 1 and 2 -- draws what the whole launch would).
 * max_id: largest node id (rowptr has max_id + 2 entries); when the id range fits an LDS bitmap
 * (~1.1 M ids) a workgroup-per-walk kernel is used (adjacency to the previous node = one bit
 * test), else a wavefront-per-walk kernel (binary search in the sorted list); max_id <= 0 or
 * kernel = 1 select the latter (kernel = 0: pick by graph size).  Both give the same walks.
 * ------------------------------------------------------------------------------------- */
int sgnn_triangular_walks(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz,
                          const int32_t* node_order, int64_t n_nodes,
                          const int64_t* patch_ptr, const int32_t* patch_nodes,
                          const int64_t* inb_ptr, const int32_t* inb_nodes,
                          int mode, int64_t n_items, int64_t walks_per_patch, int64_t walk_len, double beta,
                          uint64_t seed, uint64_t stream_id, int64_t item_base, int64_t max_id, int kernel, int64_t* out,
                          void* stream);
/* The internal AND the border walks over the same patches (aps:118-158, which the reference calls twice: inside = True / False)
 * in ONE launch: out (2, n_items, walk_len) int64, [0] = internal walks (tape stream stream_id_int), [1] = border walks
 * (stream_id_bor) -- the walks the two calls of sgnn_triangular_walks (modes 1 and 2) produce.  Only where the graph's id
 * bitmap fits LDS (max_id < ~1.1 M): SGNN_ERR_SET_TOO_LARGE otherwise, and the caller makes the two calls. */
int sgnn_triangular_walks_both(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz,
                               const int64_t* patch_ptr, const int32_t* patch_nodes, const int64_t* inb_ptr,
                               const int32_t* inb_nodes, int64_t n_items, int64_t walks_per_patch, int64_t walk_len,
                               double beta, uint64_t seed, uint64_t stream_id_int, uint64_t stream_id_bor,
                               int64_t item_base, int64_t max_id, int64_t* out, void* stream);

/* in-border nodes of a patch (subgraph_utils.get_border_nodes, subgraph_utils.py:126-144, with
 * its id-1 / node-order indexing quirk): out_flag[i] = 1 iff patch_nodes[i] is a border node.
 * node_order[i] = id at position i of G.nodes(); node_pos[id] = position (0-based). */
int sgnn_patch_in_border(const int64_t* rowptr, const int32_t* col, int64_t nnz,
                         const int32_t* node_order, const int32_t* node_pos, int64_t n_nodes,
                         const int64_t* patch_ptr, const int32_t* patch_nodes, int64_t n_patches,
                         uint8_t* out_flag, void* stream);
/* Patches of more than 2048 nodes (ego-graph patches around hubs): the call above marks their flags 255 and this one,
 * issued after it on the same stream, fills them in; the membership table lives in the workspace
 * (sgnn_patch_in_border_huge_workspace_bytes(total_nodes), total_nodes = patch_ptr[n_patches]). */
int64_t sgnn_patch_in_border_huge_workspace_bytes(int64_t total_nodes);
int sgnn_patch_in_border_huge(const int64_t* rowptr, const int32_t* col, int64_t nnz,
                              const int32_t* node_order, const int32_t* node_pos, int64_t n_nodes,
                              const int64_t* patch_ptr, const int32_t* patch_nodes, int64_t n_patches,
                              int64_t total_nodes, uint8_t* out_flag, void* workspace, int64_t workspace_bytes,
                              void* stream);

/* ---------------------------------------------------------------------------------------
 * a9  Shortest-path similarities.
 * Replaces SubGNN.compute_shortest_path_similarities (SubGNN/SubGNN.py:752-781), dense-parity
 * form: out[r, :] = min over v in set r of apsp[v-1, :] (float64 in, float32 out), empty set
 * -> PAD.  apsp: (n, n_cols) row-major.
 * ------------------------------------------------------------------------------------- */
int sgnn_sp_similarity_dense(const double* apsp, int64_t n_cols,
                             const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                             float* out, void* stream);
/* sparse form for graphs where the N x N matrix cannot exist: hop distance from each of
 * n_sources source nodes to every node, by a level-synchronous bit-parallel multi-source BFS
 * (64 sources per machine word).  dist: (n_sources, max_id + 1) uint8, 255 = not reached within
 * max_hops (the reference's matrix holds 0 for unreachable pairs, precompute_graph_metrics.py:
 * 20-25; sgnn_min_hops_to_sets applies that convention).
 * workspace: sgnn_bfs_hops_workspace_bytes(max_id, n_sources, max_hops) bytes (any content).
 * The expansion is direction-optimising: a level pulls (every incomplete node ORs its neighbours'
 * frontier words) instead of pushing once the frontier's edge volume exceeds 1/alpha of all edges.
 * pull_alpha: that alpha (negative = the default, 32; 0 = always push); results do not depend on it. */
int64_t sgnn_bfs_hops_workspace_bytes(int64_t max_id, int64_t n_sources, int max_hops);
int sgnn_bfs_hops(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                  const int32_t* sources, int64_t n_sources, int max_hops, int node_major, int pull_alpha,
                  uint8_t* dist, void* workspace, int64_t workspace_bytes, void* stream);
/* out[r, a] = min over v in set r of (dist[a, v] == 255 ? 0 : dist[a, v])  (float32; empty set -> 0) */
/* node_major != 0: dist is laid out (max_id + 1, n_sources) instead -- the sources of a node are
 * contiguous, which coalesces both the BFS writes and the per-(set, source) min below */
int sgnn_min_hops_to_sets(const uint8_t* dist, int64_t n_sources, int64_t max_id, int node_major,
                          const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                          float* out, void* stream);
/* both steps in one: out[r, s] = min over the members of set r of the hop distance from source s
 * (0 when some member is unreachable from s: the reference's matrix holds 0 there and its row-min
 * runs over it) -- compute_shortest_path_similarities
 * (SubGNN/SubGNN.py:752-781) restricted to the anchor columns, without the (sources x nodes) hop
 * table: after every BFS level each set ORs its members' new frontier words and records the level for
 * the sources it sees for the first time.  out: (n_sets, n_sources) float32.
 * All max_hops levels are enqueued without a host synchronisation (levels behind the last productive one exit at
 * once).  out_status (nullable, int32[4], device): [0] = the last level that found anything, [1] = 1 if level
 * max_hops itself still did -- the caller enqueued too few levels and the result may be incomplete, [2] = the first
 * level that pulled (0: none), [3] = 0.  A caller that repeats the same search (the same graph and anchors every pass)
 * can read [0] once and pass max_hops = [0] + a margin afterwards, checking [1] at its next synchronisation point.
 * push_levels: the levels that may still push -- each costs a commit launch besides its expand launch; beyond them a
 * level pulls whatever its frontier (one launch: a pull level writes the next version of the seen rows itself).
 * < 0: all of them.  Results do not depend on it; a repeated search passes [2] + a margin. */
int64_t sgnn_bfs_min_hops_workspace_bytes(int64_t max_id, int64_t n_sources, int max_hops, int64_t n_sets);
int sgnn_bfs_min_hops_to_sets(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                              const int32_t* sources, int64_t n_sources, int max_hops, int pull_alpha, int push_levels,
                              const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                              float* out, int32_t* out_status, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a11  Structure similarity: 1 / (1 + fastdtw(x, y, radius=1, dist=calc_dist)).
 * Replaces gamma.calc_dist / gamma.calc_dtw (SubGNN/gamma.py:51-59) and the all-pairs driver
 * SubGNN.compute_structure_patch_similarities (SubGNN/SubGNN.py:783-833).
 * x = CC degree sequences (x_ptr/x_val, n_x rows), y = anchor degree sequences (n_y rows);
 * out (n_x, n_y) float32; rows with an empty x are PAD (SubGNN.py:831).  fp64 DP.
 * tie_order 0 = (i-1,j),(i,j-1),(i-1,j-1) first minimum (pure-Python fastdtw 0.3.4).
 * x_order (nullable, int32[n_x], a permutation): processing order of the x rows -- results are
 * unaffected; putting similar rows next to each other keeps a wavefront's lanes in step.
 * workspace: sgnn_dtw_workspace_bytes(n_x, max_x_len, n_y, max_y_len) bytes (any content).
 * ------------------------------------------------------------------------------------- */
/* Keys whose ascending order is a good processing order (x_order) for sgnn_dtw_similarity: (length, up to six
 * entries of the row's twice-halved series on a log scale), packed into an int64 per x row; empty rows sort first.
 * Sorting by them is the caller's (any sort). */
int sgnn_dtw_order_keys(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t* out_keys, void* stream);
int64_t sgnn_dtw_workspace_bytes(int64_t n_x, int64_t max_x_len, int64_t n_y, int64_t max_y_len);
/* kernel: 0 = pick by size (x rows of at most 32 entries: the register-resident kernel), 1 = the general
 * (workspace-resident) kernel whatever the size -- same values, bit for bit */
int sgnn_dtw_similarity(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t max_x_len,
                        const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int64_t max_y_len,
                        int tie_order, int kernel, const int32_t* x_order, float* out, void* workspace,
                        int64_t workspace_bytes, void* stream);
/* Same, for callers whose x rows are mostly empty (repeated rows given length 0 by a grouping step):
 * x_live_range (device, int64[2] = {first, count}, nullable) names the positions of the processing
 * order x_order that hold the non-empty rows -- x_order must list the empty rows first -- and only those
 * are computed; the caller zero-fills (PAD) the output beforehand.  The range lives on the device so
 * that no host round trip is needed to learn it. */
int sgnn_dtw_similarity_live(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t max_x_len,
                             const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int64_t max_y_len,
                             int tie_order, int kernel, const int32_t* x_order, const int64_t* x_live_range, float* out,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a12  CC embedding initialisation: sum or max of member node embeddings.
 * Replaces SubGNN.initialize_cc_embeddings (SubGNN/SubGNN.py:609-622).  E: (n_emb_rows, D) f32.
 * aggregator 0 = sum, 1 = max.  For max, a row shorter than padded_len also competes with the
 * zero PAD row (SubGNN.py:622); out_arg (n_sets, D) int32 receives the winning node id
 * (0 = PAD) for the backward pass.
 * ------------------------------------------------------------------------------------- */
int sgnn_cc_embed_fwd(const float* E, int64_t D,
                      const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                      int aggregator, int64_t padded_len, float* out, int32_t* out_arg, void* stream);
/* same with the table stored as IEEE half (rows, D): read as half, accumulated and returned in fp32
 * (BASELINE.json configs[4]: "fp16 embeddings"); the backward is sgnn_cc_embed_bwd (fp32 gradient). */
int sgnn_cc_embed_fwd_f16(const uint16_t* E_half, int64_t D,
                          const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                          int aggregator, int64_t padded_len, float* out, int32_t* out_arg, void* stream);
/* grad_E (n_emb_rows, D) is accumulated into (float atomics), row PAD untouched */
int sgnn_cc_embed_bwd(const float* grad_out, int64_t D,
                      const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                      int aggregator, const int32_t* arg, float* grad_E, void* stream);

/* ---------------------------------------------------------------------------------------
 * a13+a15  One anchor -> component message-passing layer (gather, weight, aggregate, read-out).
 * Replaces the body of SG_MPN.forward / propagate / message / generate_pos_struc_embeddings
 * (SubGNN/subgraph_mpn.py:105-174, 227-231) together with the anchor gather of
 * anchor_patch_samplers.get_anchor_patches / embed_anchor_patch (anchor_patch_samplers.py:
 * 333-411).  For row r (= b*C + c) and anchor slot a:
 *     edge(r,a)  -- see `src`
 *     w          = sims[r*sims_ld + column(r,a)]
 *     agg[r,:]   = sum_a edge * w * x(r,a,:)
 *     z[r,a]     = edge ? w * <wp, x(r,a,:)> + bp : bp       (pre-activation read-out)
 * Anchor rows x(r,a,:) come from one of three sources:
 *   SGNN_SRC_DENSE  x = anchor_embeds[(r*A+a)*D ..]  (the reference's materialised (B,C,A,D)
 *                   tensor); edge = edge_mask[r*A+a]
 *   SGNN_SRC_GATHER x = E[ids[(r/id_div)*A+a]*D ..]; edge = id != 0 && row_mask[r]
 *   SGNN_SRC_SHARED x = X[a*D ..] (P-border / structure anchors shared by every row);
 *                   edge = row_mask[r] && (ids == NULL || ids[a] != 0)
 * column(r,a): sim_col != NULL -> sim_col[a]; else sims_per_edge -> a; else id - 1.
 * The Linear(2D->D)+ReLU update (subgraph_mpn.py:233-239) is a plain GEMM done by the caller.
 * ------------------------------------------------------------------------------------- */
#define SGNN_SRC_DENSE  0
#define SGNN_SRC_GATHER 1
#define SGNN_SRC_SHARED 2

typedef struct sgnn_mpn_args {
    int32_t src;               /* SGNN_SRC_* */
    int32_t sims_per_edge;     /* 1: sims is (R, A) already gathered */
    int64_t R, A, D;
    const float*   x;          /* DENSE: (R,A,D) anchor_embeds; GATHER: E (rows,D); SHARED: X (A,D) */
    const int64_t* ids;        /* DENSE/GATHER: (R/id_div, A) anchor ids; SHARED: (A) or NULL */
    int64_t        id_div;     /* rows sharing one ids row (C for P-internal, else 1) */
    const uint8_t* edge_mask;  /* DENSE: (R,A) */
    const uint8_t* row_mask;   /* GATHER/SHARED: (R) cc_embed_mask, may be NULL (= all real) */
    const float*   sims;       /* (R, sims_ld) */
    int64_t        sims_ld;
    const int64_t* sim_col;    /* (A) or NULL */
    const float*   wp;         /* (D) linear_position.weight */
    const float*   bp;         /* (1) linear_position.bias   */
    int32_t        x_f16;      /* GATHER only: x points at an IEEE half table (rows, D), read as
                                * half and accumulated in fp32 (gradients stay fp32) */
    const float*   z_act;      /* nullable, (R, A): the read-out as sgnn_mpn_fwd wrote it under SGNN_MPN_RELU_Z.  When set, every
                                * backward entry point takes grad_z through that relu: an entry whose z_act is not > 0 counts as 0 */
    int32_t        flags;      /* SGNN_MPN_WP_PARTIAL: sgnn_mpn_bwd (DENSE) writes grad_wp as per-row partial sums (R, D)
                                * for the caller to add up in a fixed order, instead of adding into (D) with atomics.
                                * SGNN_MPN_RELU_Z: sgnn_mpn_fwd writes the read-out AFTER its non-linearity, relu(z) (mpn:122-131
                                * generate_pos_struc_embeddings applies it next), instead of the pre-activation */
} sgnn_mpn_args;
#define SGNN_MPN_WP_PARTIAL 1
#define SGNN_MPN_RELU_Z     2

/* Batch-sized calls (a few hundred rows) split the anchors of a row over sgnn_mpn_fwd_chunks(args) chunks so that
 * the launch fills the chip; chunk c writes its partial aggregate to agg + c * R * D and the caller adds the
 * chunks up (a fixed order: no atomics).  agg: (chunks, R, D); 1 chunk for shard-sized calls. */
int sgnn_mpn_fwd_chunks(const sgnn_mpn_args* args);
int sgnn_mpn_fwd(const sgnn_mpn_args* args, float* agg /*(chunks,R,D)*/, float* z /*(R,A)*/, void* stream);
/* The bodies of ONE message-passing layer (up to sgnn_mpn_fwd_many_max_bodies(): they read the layer below only) in one launch:
 * args is a HOST array of n argument blocks, agg[k] / z[k] (HOST arrays of DEVICE pointers) as sgnn_mpn_fwd writes them for
 * args[k] (agg[k]: sgnn_mpn_fwd_chunks(&args[k]) chunks; z[k] nullable).  Every body needs R > 0 and A > 0. */
int64_t sgnn_mpn_fwd_many_max_bodies(void);
int sgnn_mpn_fwd_many(int64_t n, const sgnn_mpn_args* args, float* const* agg, float* const* z, void* stream);
/* grad_x: DENSE (R,A,D) written; GATHER (rows,D) accumulated with float atomics, row PAD
 * untouched (the atomics-free form of GATHER: sgnn_mpn_bwd_edges + sgnn_scatter_add_rows_sorted); SHARED (A,D)
 * accumulated.  grad_wp (D) accumulated, or (R,D) written with SGNN_MPN_WP_PARTIAL.  Any of them may be NULL. */
int sgnn_mpn_bwd(const sgnn_mpn_args* args, const float* grad_agg, const float* grad_z,
                 float* grad_x, float* grad_wp, void* stream);

/* ---------------------------------------------------------------------------------------
 * a16 (optional ff_attn read-out)  Additive-attention scores over a subgraph's components.
 * Replaces attention.AdditiveAttention._forward_internal (SubGNN/attention.py:130-139) as used at
 * SubGNN/SubGNN.py:298-301:  out[r] = sum_j v[j] * tanh(qW[r / rows_per_batch, j] + (X U)[r, j]).
 * X (R, H) component embeddings, U (H, H) = _u_matrix, qW (R / rows_per_batch, H) = vector @ _w_matrix,
 * v (H) = _v_vector.
 * ------------------------------------------------------------------------------------- */
/* exact f32 form: the caller's BLAS computes XU = X U (a plain dense GEMM); this is everything after it, fused:
 * out[r] = sum_j v_j tanh(XU[r, j] + qW[r / rows_per_batch, j]). */
int sgnn_attn_scores_epilogue(const float* XU, const float* qW, const float* v, int64_t R, int64_t H,
                              int64_t rows_per_batch, float* out, void* stream);
/* half operands on the matrix cores, one kernel (v_mfma_f32_32x32x16_f16, fp32 accumulate; X and U are rounded to IEEE
 * half on the way in; four wavefronts per workgroup share each 32-column panel of U through LDS): the fp16 form of
 * BASELINE.json configs[4].  H <= 640, else SGNN_ERR_UNSUPPORTED_D (the caller takes the exact form).
 * workspace: sgnn_attn_scores_f16_workspace_bytes(H) bytes (the transposed half copy of U). */
int64_t sgnn_attn_scores_f16_workspace_bytes(int64_t H);
int sgnn_attn_scores_fwd_f16(const float* X, const float* U, const float* qW, const float* v,
                             int64_t R, int64_t H, int64_t rows_per_batch, float* out,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a14 (walk aggregator)  The recurrence of one bidirectional LSTM layer, whole sequence per launch.
 * Replaces the nn.LSTM(bidirectional=True, batch_first=True) inside the reference's LSTM module
 * (SubGNN/SubGNN.py:60-88) as called by aggregate_structure_anchor_patch
 * (SubGNN/anchor_patch_samplers.py:413-433) on (patches x walks, walk_len, D).  h0 = c0 = 0.
 * The non-recurrent contractions are plain GEMMs over all (sequence, step) rows and stay with the
 * caller's BLAS: the input projection before the forward call, dx / dW_ih / dW_hh / db after the
 * backward call.  Gate order i, f, g, o (torch).
 *   pre_x  (2, B, T, 4H)   x W_ih^T + b_ih per direction (0 = forward, 1 = reverse), direction-major: each direction's
 *                          rows are one GEMM's contiguous output
 *   whh_f, whh_r (4H, H)   weight_hh per direction;  bhh_f, bhh_r (4H), nullable: bias_hh per direction, added here (the
 *                          caller needs no concatenated weights and no summed biases: five small launches per layer less)
 *   y      (B, T, 2H)      [forward h_t | reverse h_t]  (torch's output layout)
 *   gates  (2, B, T, 4H), cell (2, B, T, H), hprev (2, B, T, H) = h_{t-1} in the direction's order:
 *          activations kept for the backward pass
 * backward: dy (B, T, 2H) -> dgates (2, B, T, 4H), the gradient w.r.t. pre_x.  Then, with dG[d] = dgates[d] viewed (B T, 4H):
 * dx = dG[0] W_ih_f + dG[1] W_ih_r,  dW_ih[d] = dG[d]^T x,  dW_hh[d] = dG[d]^T hprev[d],  db[d] = column sums of dG[d].
 * Hidden sizes 32, 64, 128 (sgnn_lstm_supported), any input size; else SGNN_ERR_UNSUPPORTED_D -- the
 * caller keeps the library LSTM for those.
 * ------------------------------------------------------------------------------------- */
int sgnn_lstm_supported(int64_t hidden_size);
int sgnn_lstm_fwd(const float* pre_x, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                  int64_t B, int64_t T, int64_t hidden_size, float* y, float* gates, float* cell, float* hprev, void* stream);
int sgnn_lstm_bwd(const float* whh_f, const float* whh_r, const float* gates, const float* cell, const float* dy,
                  int64_t B, int64_t T, int64_t hidden_size, float* dgates, void* stream);

/* The dense products around the recurrence, and the walk aggregator's tail (anchor_patch_samplers.py:413-433 / SubGNN.py:60-88:
 * embedding lookup -> LSTM -> last step (or sum over steps) -> Linear -> sum over a patch's walks), in this library's launches:
 *   sgnn_rows_gemm     out[z] (R, N) = X[row(r)] W_z^T + b_z for z = 0 (and 1 when W1 is given): fp32 MFMA; row(r) = ids[r] when
 *                      ids (int64, nullable) is given -- the gather of the input rows is the operand load -- and the gathered
 *                      rows are written to x_copy (R, K; nullable).  X rows of ldx floats, K % 8 == 0; W_z (N, K) row-major.
 *   sgnn_rows_gemm_nt  out (R, N) = A[0] W0 + A[1] W1 with A (2, R, K) and W_z (K, N) row-major (dx of both directions).
 *   sgnn_lstm_tail_fwd X (n_patches, D) = s W_lin^T + n_walks b_lin, s[p] = sum over the patch's walks of y[walk][T - 1] (last_only)
 *                      or of every step; y (n_patches n_walks, T, H2); s_out (n_patches, H2) kept for the backward.
 *   sgnn_lstm_tail_bwd dy (same shape as y, every element written), dW_lin (D, H2), db_lin (D) (both nullable) from dX and s. */
int sgnn_rows_gemm(const float* X, const int64_t* ids, int64_t ldx, int64_t R, int64_t K, const float* W0, const float* W1,
                   const float* b0, const float* b1, int64_t N, float* out, float* x_copy, void* stream);
int sgnn_rows_gemm_nt(const float* A, int64_t R, int64_t K, const float* W0, const float* W1, int64_t N, float* out, void* stream);
int sgnn_lstm_tail_fwd(const float* y, int64_t n_patches, int64_t n_walks, int64_t T, int64_t H2, int last_only,
                       const float* W_lin, const float* b_lin, int64_t D, float* s_out, float* X, void* stream);
int sgnn_lstm_tail_bwd(const float* dX, const float* s, int64_t n_patches, int64_t n_walks, int64_t T, int64_t H2,
                       int last_only, const float* W_lin, int64_t D, float* dy, float* dW_lin, float* db_lin, void* stream);

/* ---------------------------------------------------------------------------------------
 * a16  Masked sum over the components of a subgraph (subgraph_utils.masked_sum,
 * SubGNN/subgraph_utils.py:213-237, as used at SubGNN/SubGNN.py:303).  x (B,C,H), mask (B,C)
 * -> out (B,H); backward scatters grad_out to the real components.
 * ------------------------------------------------------------------------------------- */
int sgnn_masked_sum_fwd(const float* x, const uint8_t* mask, int64_t B, int64_t C, int64_t H,
                        float* out, void* stream);
int sgnn_masked_sum_bwd(const float* grad_out, const uint8_t* mask, int64_t B, int64_t C, int64_t H,
                        float* grad_x, void* stream);

/* ---------------------------------------------------------------------------------------
 * a16b  The tail of the forward pass without the (B, C, H) concatenation (SubGNN/SubGNN.py:286-312: cat of the
 * channel outputs, then masked_sum): every piece is summed over a subgraph's real components straight into its
 * column slot of the (B, H) subgraph embedding (row stride out_ld / grad_ld, in floats).
 *   sgnn_masked_sum_slot_fwd/_bwd: a piece that exists as a tensor x (B, C, W) (component embeddings).
 *   sgnn_readout_sum_fwd/_bwd: the read-out of a layer over SHARED anchors (subgraph_mpn.py:122-131: position read-out
 *     of the messages + relu) when only the read-out is consumed.  With s[a] = X[a,:] . wp (A values, the caller's)
 *       out[b, a] = sum over real components c of  relu(W[b,c,a] * s[a] + bp[0]),
 *       W[b,c,a] = sims[(b C + c) sims_ld + (sim_col ? sim_col[a] : a)]    (sims NULL: W = 0)
 *     row_mask (B C, nullable): 0 = padded component (contributes nothing).  Backward: grad_s[a] = sum_r g [z>0] W,
 *     grad_bp = sum g [z>0], by per-row-block partials added in block order (no atomics: bit-reproducible);
 *     either output may be NULL.  workspace: sgnn_readout_sum_bwd_workspace_bytes.
 * ------------------------------------------------------------------------------------- */
int sgnn_masked_sum_slot_fwd(const float* x, const uint8_t* mask, int64_t B, int64_t C, int64_t W, float* out,
                             int64_t out_ld, void* stream);
int sgnn_masked_sum_slot_bwd(const float* grad_out, int64_t grad_ld, const uint8_t* mask, int64_t B, int64_t C,
                             int64_t W, float* grad_x, void* stream);
/* The same for n_pieces tensors in one launch (per 96 pieces): xs[i] (B, C, widths[i]) summed into columns
 * [offsets[i], offsets[i] + widths[i]) of out (B, out_ld); backward: grad_xs[i] (B, C, widths[i]) written from those columns of
 * grad_out (a null grad_xs[i] is skipped).  xs / grad_xs / widths / offsets are HOST arrays (of DEVICE pointers).  The form a
 * batch-sized step uses: 22 pieces of a 4-layer model are one launch each way instead of 22. */
int sgnn_masked_sum_slots_fwd(const float* const* xs, const int64_t* widths, const int64_t* offsets, int64_t n_pieces,
                              const uint8_t* mask, int64_t B, int64_t C, float* out, int64_t out_ld, void* stream);
int sgnn_masked_sum_slots_bwd(const float* grad_out, int64_t grad_ld, const uint8_t* mask, int64_t B, int64_t C,
                              float* const* grad_xs, const int64_t* widths, const int64_t* offsets, int64_t n_pieces, void* stream);

/* ---------------------------------------------------------------------------------------
 * a17  Loss + accuracy of a step: nn.CrossEntropyLoss() with its default mean reduction (SubGNN/SubGNN.py:133, applied
 * at SubGNN.py:1116-1124) and subgraph_utils.calc_accuracy (SubGNN/subgraph_utils.py:108-124: argmax == label, mean)
 * in one pass over logits (B, K) row-major, labels int64 (B) in [0, K) or -100 (nn.CrossEntropyLoss's ignore_index: the row
 * contributes nothing and the mean is over the other rows; any other label outside [0, K) makes the loss NaN -- the library
 * raises).  lse (B + 1 floats): the rows' log-sum-exp and, in lse[B], the number of rows not ignored, kept for the
 * backward.  loss, accuracy: one float each (accuracy nullable; over all B rows).
 * grad_logits = (softmax - onehot) * grad_loss[0] / lse[B].
 * Partial sums are added in a fixed order: bit-reproducible.
 * ------------------------------------------------------------------------------------- */
int64_t sgnn_cross_entropy_workspace_bytes(int64_t B);
int sgnn_cross_entropy_fwd(const float* logits, const int64_t* labels, int64_t B, int64_t K, float* lse, float* loss,
                           float* accuracy, void* workspace, int64_t workspace_bytes, void* stream);
int sgnn_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* lse, const float* grad_loss,
                           int64_t B, int64_t K, float* grad_logits, void* stream);

/* Column sums of a row-major matrix x (R rows of A floats, row stride ld): out[a] = sum_r x[r, a] -- the bias gradients of the
 * head's Linear layers over a shard's rows (autograd of SubGNN/SubGNN.py:304-312).  Row-block partials added in block order:
 * bit-reproducible.  workspace: sgnn_column_sum_workspace_bytes. */
int64_t sgnn_column_sum_workspace_bytes(int64_t R, int64_t A);
int sgnn_column_sum(const float* x, int64_t ld, int64_t R, int64_t A, float* out, void* workspace, int64_t workspace_bytes,
                    void* stream);

int sgnn_readout_sum_fwd(const float* sims, int64_t sims_ld, const int64_t* sim_col, const float* s, const float* bp,
                         const uint8_t* row_mask, int64_t B, int64_t C, int64_t A, float* out, int64_t out_ld,
                         void* stream);
int64_t sgnn_readout_sum_bwd_workspace_bytes(int64_t B, int64_t C, int64_t A);
int sgnn_readout_sum_bwd(const float* grad_out, int64_t grad_ld, const float* sims, int64_t sims_ld,
                         const int64_t* sim_col, const float* s, const float* bp, const uint8_t* row_mask, int64_t B,
                         int64_t C, int64_t A, float* grad_s, float* grad_bp, void* workspace, int64_t workspace_bytes,
                         void* stream);

/* Every read-out piece of a step in ONE launch each way (n <= sgnn_readout_many_max() pieces; all arrays HOST arrays with one
 * entry per piece, of DEVICE pointers where they hold pointers).  Piece k: similarity rows sims[k] (ld sims_ld[k]; NULL = all
 * zero), columns sim_col[k] (nullable), anchor embeddings X[k] (A[k], D) with read-out weight wp[k] (D) -- the scores
 * s[a] = X[a, :] . wp, 0 where ids[k][a] == 0 (ids nullable) or X[k] is NULL, are computed by the call into s[k] (A[k] floats) --
 * bias bp[k], row mask row_mask[k] (nullable), column slot off[k] .. off[k] + A[k] of the (B, out_ld) embedding.
 * Backward: grad_X[k] (A[k], D), grad_wp[k] (D), grad_bp[k] (1), each nullable; tickets: RO_MAX_PIECES uint32, zero before the
 * first call and left zero.  Same sums in the same order as the single-piece calls. */
int64_t sgnn_readout_many_max(void);
int sgnn_readout_many_fwd(int64_t n, const float* const* sims, const int64_t* sims_ld, const int64_t* const* sim_col,
                          const float* const* X, const float* const* wp, const float* const* bp,
                          const int64_t* const* ids, const uint8_t* const* row_mask, float* const* s, const int64_t* A,
                          const int64_t* off, int64_t D, int64_t B, int64_t C, float* out, int64_t out_ld, void* stream);
int64_t sgnn_readout_many_bwd_workspace_bytes(int64_t n, const int64_t* A, int64_t B, int64_t C);
int sgnn_readout_many_bwd(int64_t n, const float* grad_out, int64_t grad_ld, const float* const* sims, const int64_t* sims_ld,
                          const int64_t* const* sim_col, const float* const* X, const float* const* wp,
                          const float* const* bp, const int64_t* const* ids, const uint8_t* const* row_mask,
                          float* const* s, const int64_t* A, const int64_t* off, int64_t D, int64_t B, int64_t C,
                          float* const* grad_X, float* const* grad_wp, float* const* grad_bp, void* workspace,
                          int64_t workspace_bytes, unsigned* tickets, void* stream);

/* ---------------------------------------------------------------------------------------
 * a18  Embedding-table gradient without atomics (the backward of every op that gathers table rows:
 * autograd of SubGNN/SubGNN.py:609-622, anchor_patch_samplers.py:404-411, subgraph_mpn.py:227-231 as
 * run by loss.backward(), SubGNN/SubGNN.py:1163-1164).
 *   table[key[e], :] += c1[e] * G[row(e), :] + c2[e] * v[:]      for e = order[0], order[1], ...
 * order: the edge numbers sorted STABLY by target key; key_sorted[p] = key of order[p] (ascending); key 0
 * (the PAD row) contributes nothing.  row(e) = edge_row[e], or e / edges_per_row when edge_row is NULL.
 * c1 NULL = 1; c2 / v NULL = no second term; G NULL = only the second term.  arg (nullable, (rows, D) int32):
 * column d of edge e counts only where arg[row(e), d] == key (the max aggregator's argmax).
 * Every table row has one writer and a fixed summation order: results are bit-reproducible.
 * sgnn_mpn_bwd_edges: keys and coefficients of a GATHER message-passing layer's backward (c1 = the edge
 * weight w, c2 = w * grad_z; masked or zero-weight edges get key 0); sgnn_mpn_bwd_wp_partial: that layer's
 * read-out weight gradient as per-row partial sums (R, partial_ld) for the caller to add up; partial_ld = D, or D + 1: then
 * column D holds the row's sum of grad_z (through args->z_act), the per-row partial of the read-out bias's gradient.
 * ------------------------------------------------------------------------------------- */
int64_t sgnn_scatter_add_rows_workspace_bytes(int64_t n_edges, int64_t D);
/* The stable sort the scatter needs: key_sorted / order (int32, n_edges each) from keys in [0, max_key] -- a radix
 * sort over the bits max_key has (rocPRIM), positions as the payload.  An order that does not change between
 * passes (the component members of a split) can be computed once and kept. */
int64_t sgnn_sort_edges_by_key_workspace_bytes(int64_t n_edges, int64_t max_key);
int sgnn_sort_edges_by_key(const int32_t* keys, int64_t n_edges, int64_t max_key, int32_t* key_sorted, int32_t* order,
                           void* workspace, int64_t workspace_bytes, void* stream);
int sgnn_scatter_add_rows_sorted(const int32_t* order, const int32_t* key_sorted, int64_t n_edges,
                                 const int32_t* edge_row, int64_t edges_per_row,
                                 const float* G, int64_t D, const float* c1, const float* c2, const float* v,
                                 const int32_t* arg, float* table,
                                 void* workspace, int64_t workspace_bytes, void* stream);
/* n_lists scatters into ONE table as one sorted list: the edge lists are concatenated (list order, then position), sorted once
 * (stable) and scattered once -- the form a batch-sized training step uses for the 6-18 short lists (1-30 k edges: one per layer
 * body, the component embeddings, the shared anchors' lookups) it adds to the embedding table's gradient: one pack launch, one
 * sort, two scatter launches instead of 6-18 times (sort + two).  keys / n_edges / edge_row / edges_per_row / G / c1 / c2 / v:
 * HOST arrays over the lists, each entry as the argument of that name of sgnn_scatter_add_rows_sorted (DEVICE pointers; every
 * G[k] has D columns; no argmax form).  Every table row has one writer and a fixed order of addition. */
int64_t sgnn_scatter_add_rows_multi_workspace_bytes(int64_t total_edges, int64_t D, int64_t max_key);
int sgnn_scatter_add_rows_multi(int64_t n_lists, const int32_t* const* keys, const int64_t* n_edges,
                                const int32_t* const* edge_row, const int64_t* edges_per_row, const float* const* G,
                                const float* const* c1, const float* const* c2, const float* const* v, int64_t D,
                                int64_t max_key, float* table, void* workspace, int64_t workspace_bytes, void* stream);
/* SHARED source, batch-sized calls, without atomics: per-row-tile partials in the workspace, added in tile order
 * (grad_x (A, D), grad_wp (D) and grad_bp (1) -- the read-out bias's gradient, the sum of every entry of grad_z taken through
 * args->z_act -- are OVERWRITTEN, each may be NULL). */
int64_t sgnn_mpn_bwd_shared_det_workspace_bytes(int64_t R, int64_t A, int64_t D);
int sgnn_mpn_bwd_shared_det(const struct sgnn_mpn_args* args, const float* grad_agg, const float* grad_z,
                            float* grad_x, float* grad_wp, float* grad_bp, void* workspace, int64_t workspace_bytes,
                            void* stream);
int sgnn_mpn_bwd_edges(const struct sgnn_mpn_args* args, const float* grad_z, int32_t* out_keys, float* out_c1,
                       float* out_c2, void* stream);
int sgnn_mpn_bwd_wp_partial(const struct sgnn_mpn_args* args, const float* grad_z, float* partial, int64_t partial_ld,
                            void* stream);
/* sgnn_mpn_bwd_edges for up to sgnn_mpn_fwd_many_max_bodies() GATHER bodies in one launch (HOST arrays of n entries; out_c2[k] /
 * grad_z[k] nullable as in the single form): the edge lists feed the step's combined table-gradient scatter, which runs when the
 * table's gradient is handed over -- until then they can wait for each other. */
int sgnn_mpn_bwd_edges_many(int64_t n, const struct sgnn_mpn_args* args, const float* const* grad_z, int32_t* const* out_keys,
                            float* const* out_c1, float* const* out_c2, void* stream);

/* ---------------------------------------------------------------------------------------
 * a12  update(): out = relu([x | aggr] W^T + b) and its backward (SubGNN/subgraph_mpn.py:233-241 with the
 * nn.Linear(2 D, D) of subgraph_mpn.py:33), one row per component.  x, aggr, out, grad_*: (R, D) float32 row-major;
 * W (D, 2 D); b (D), nullable in the forward.  D in {32, 64, 128} (else SGNN_ERR_UNSUPPORTED_D: the caller keeps
 * the library-GEMM form).  fp32 operands and accumulation (v_mfma_f32_32x32x2_f32).
 * Backward: dpre = grad_out * (out > 0); [grad_x | grad_aggr] = dpre W (either may be NULL: not computed);
 * grad_W = dpre^T [x | aggr], grad_b = column sums of dpre (either may be NULL) -- contracted over row blocks whose
 * partial sums (workspace) are added in block order: bit-reproducible.
 * ------------------------------------------------------------------------------------- */
int sgnn_update_fwd(const float* x, const float* aggr, const float* W, const float* b, int64_t R, int64_t D,
                    float* out, void* stream);
/* The forward with aggr as the anchor-chunk partial aggregates sgnn_mpn_fwd writes, aggr_chunks (n_chunks, R, D): added in
 * chunk order while they are loaded (no separate reduction launch per layer) and, when aggr_sum (R, D) is given, written out
 * once for sgnn_update_bwd.  n_chunks > 1 needs R <= sgnn_update_fwd_chunks_max_rows() (the batch-sized launch shape: the
 * only one sgnn_mpn_fwd splits). */
int64_t sgnn_update_fwd_chunks_max_rows(void);
int sgnn_update_fwd_chunks(const float* x, const float* aggr_chunks, int64_t n_chunks, const float* W, const float* b,
                           int64_t R, int64_t D, float* out, float* aggr_sum, void* stream);
/* The update layers of n bodies of ONE message-passing layer (up to sgnn_update_many_max_bodies() channel sides of a batch-sized
 * step: same R <= sgnn_update_fwd_chunks_max_rows(), same D) in one launch each way -- forward 1 launch, backward 3 (dx, dW
 * partials, reduce) instead of n and 3 n.  All pointer tables are HOST arrays of n DEVICE pointers with the meaning of the
 * single-body arguments; aggr_chunks[k] is (n_chunks[k], R, D), its sum goes to aggr_sum[k] (required when n_chunks[k] > 1).
 * Backward: grad_out[k] NULL = body k received no gradient and is skipped; else grad_x[k] / grad_aggr[k] may be NULL and
 * grad_W[k], grad_b[k] are both written; workspace n * sgnn_update_bwd_workspace_bytes(R, D). */
int64_t sgnn_update_many_max_bodies(void);
int sgnn_update_fwd_many(int64_t n, const float* const* x, const float* const* aggr_chunks, const int64_t* n_chunks,
                         const float* const* W, const float* const* b, int64_t R, int64_t D, float* const* out,
                         float* const* aggr_sum, void* stream);
int sgnn_update_bwd_many(int64_t n, const float* const* grad_out, const float* const* out, const float* const* x,
                         const float* const* aggr, const float* const* W, int64_t R, int64_t D, float* const* grad_x,
                         float* const* grad_aggr, float* const* grad_W, float* const* grad_b, void* workspace,
                         int64_t workspace_bytes, void* stream);
int64_t sgnn_update_bwd_workspace_bytes(int64_t R, int64_t D);
int sgnn_update_bwd(const float* grad_out, const float* out, const float* x, const float* aggr, const float* W,
                    int64_t R, int64_t D, float* grad_x, float* grad_aggr, float* grad_W, float* grad_b,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a18  Adam on one large parameter in one pass (torch.optim.Adam's rule, no weight decay / amsgrad:
 * SubGNN/SubGNN.py:1156-1161), with the caller's clip coefficient (train_config.py: gradient_clip_val) applied on the
 * fly: grad_scale (nullable) is a DEVICE scalar, so the clip needs no host round trip.  step = 1 for the first update.
 * zero_grad != 0: the gradient is zeroed in the same pass (the buffer can be handed out again without a fill).
 * All four arrays float32[n], 16-byte aligned.
 * ------------------------------------------------------------------------------------- */
/* a18b  The caller's clip_grad_norm_ (train_config.py: Trainer(gradient_clip_val) -> torch.nn.utils.clip_grad_norm_):
 * coefficient = min(1, max_norm / (total + 1e-6)), total = 2-norm over all gradients.  sgnn_grad_sumsq: sums of squares of
 * one large float32 gradient (16-byte aligned) as sgnn_grad_sumsq_partials() per-workgroup values; sgnn_clip_coefficient adds
 * them (any number of such arrays laid end to end) and the squares of other_norms (the small parameters' 2-norms) in a fixed
 * order and writes the DEVICE scalars coef (for sgnn_adam_step's grad_scale) and total_norm (nullable). */
int64_t sgnn_grad_sumsq_partials(void);
int sgnn_grad_sumsq(const float* grad, int64_t n, float* partial, void* stream);
int sgnn_clip_coefficient(const float* partial, int64_t n_partial, const float* other_norms, int64_t n_other,
                          float max_norm, float* coef, float* total_norm, void* stream);

int sgnn_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, int64_t step, const float* grad_scale, int zero_grad, void* stream);
/* The same with the step count in DEVICE memory (int64, starts at 0): incremented on the stream right before the update, read by
 * the update for its bias corrections -- the form a step recorded into a hipGraph replays (a host counter would be frozen at
 * its value at recording time). */
int sgnn_adam_step_counted(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                           float beta2, float eps, int64_t* step_counter, const float* grad_scale, int zero_grad,
                           void* stream);

/* a18c  The whole optimizer tail of a step -- clip_grad_norm_ over ALL parameters (train_config.py: Trainer(gradient_clip_val))
 * followed by torch.optim.Adam over all of them (SubGNN/SubGNN.py:1156-1161) -- in two launches per 72 tensors.  The tensor lists
 * are HOST arrays of DEVICE pointers (float32, 4-byte aligned at least; 16-byte aligned tensors take the vector path), numels[i]
 * elements each.
 *   sgnn_optim_partials(numels, n)          how many floats sgnn_optim_sumsq writes (one per workgroup; -1 = bad argument)
 *   sgnn_optim_sumsq(grads, ...)            per-workgroup sums of squares of every gradient -> partial[]; the step counts
 *                                           step_counters[counter_slots[i]] (step_counters nullable: DEVICE int64[]; counter_slots
 *                                           nullable: HOST int64[n_tensors], default i) each advance by one in the same launch
 *   sgnn_optim_count(step_counters, ...)    the advance alone (a step without clipping)
 *   sgnn_optim_adam(...)                    every workgroup adds partial[0..n_partial) in one fixed order, forms
 *                                           coefficient = min(1, max_norm / (sqrt(sum) + 1e-6)) and updates its chunk with the
 *                                           gradient times the coefficient (max_norm <= 0: no clipping, partial unused).
 *                                           Step counts: EXACTLY ONE of steps (HOST int64[n_tensors], >= 1: this update's
 *                                           number, per tensor) and step_counters (DEVICE, tensor i reads step_counters[counter_slots[i]] after
 *                                           sgnn_optim_sumsq / sgnn_optim_count advanced them -- the form a recorded step replays).
 *                                           zero_grad (nullable HOST int32[n_tensors]): != 0 zeroes that gradient in the same
 *                                           pass.  coef_out (nullable DEVICE float[2]): the coefficient and the total norm.
 *                                           row_lens / row_seen (both nullable HOST arrays): tensor i is a table of rows of
 *                                           row_lens[i] floats (a power of two, 4..256) with one DEVICE byte per row,
 *                                           row_seen[i] (zero-initialised by the caller, owned by the optimizer state): set once
 *                                           the row has had a non-zero gradient.  A row with an all-zero gradient and a clear
 *                                           byte has m = v = 0 and Adam's update of it is exactly zero: it is skipped after the
 *                                           read of its gradient (one such table per launch; bit-identical to the dense update).
 * The gradients themselves are NOT scaled in memory (clip_grad_norm_ scales them in place; here they are consumed). */
int64_t sgnn_optim_partials(const int64_t* numels, int64_t n_tensors);
int sgnn_optim_sumsq(const float* const* grads, const int64_t* numels, int64_t n_tensors, float* partial,
                     int64_t* step_counters, const int64_t* counter_slots, void* stream);
int sgnn_optim_count(int64_t* step_counters, const int64_t* counter_slots, int64_t n_tensors, void* stream);
int sgnn_optim_adam(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* numels, const int32_t* zero_grad, int64_t n_tensors, float lr, float beta1, float beta2,
                    float eps, const int64_t* steps, const int64_t* step_counters, const int64_t* counter_slots,
                    const int64_t* row_lens, unsigned char* const* row_seen,
                    const float* partial, int64_t n_partial, float max_norm, float* coef_out, void* stream);


/* A batch's rows of up to sgnn_gather_rows_many_max() per-split tensors in ONE launch (the row gathers of _pad_collate,
 * SubGNN/SubGNN.py:1068-1114: component ids, border ids, the channels' similarity rows, labels): dst[t][i, :] =
 * src[t][idx[i], :] with rows as raw bytes -- row_bytes[t] each, src_rows[t] rows in the source.  An index outside the source
 * (index_select raises for it) yields a ZERO row and sets *out_of_range (nullable DEVICE int32, never cleared here) to 1: the
 * caller polls the flag.  n * B up to 2^30 (the pair travels in blockIdx.x).
 * src / dst / row_bytes / src_rows: HOST arrays (src / dst of DEVICE pointers); idx: DEVICE int64[B]. */
int64_t sgnn_gather_rows_many_max(void);
int sgnn_gather_rows_many(int64_t n, const void* const* src, void* const* dst, const int64_t* row_bytes,
                          const int64_t* src_rows, const int64_t* idx, int64_t B, int32_t* out_of_range, void* stream);

/* ---------------------------------------------------------------------------------------
 * a16/a17 fused  The MLP head of SubGNN.forward + the loss of a training step (SubGNN/SubGNN.py:304-312: lin -> relu -> dropout
 * -> lin2 -> relu -> dropout -> lin3; SubGNN.py:1116-1124: nn.CrossEntropyLoss + subgraph_utils.calc_accuracy) behind the first
 * layer's GEMM: z1 = x W1^T + b1 (B, H1) comes from the caller's library GEMM, as does dx = dz1 W1 in the backward.
 *   sgnn_head_supported(H1, H2, K)   1 when the widths fit the kernels' LDS tables (H1, H2 <= 128, K <= 32)
 *   sgnn_head_blocks(B)              workgroups of either launch; the backward writes sgnn_head_partial_floats(H1, H2, K) floats
 *                                    per workgroup: [gW3 (K H2) | gb3 (K) | gW2 (H2 H1) | gb2 (H2) | gb1 (H1)], to be added over the
 *                                    workgroups in order (sgnn_reduce_partials)
 *   sgnn_head_fwd    a1 = drop(relu(z1)) (B, H1), a2 = drop(relu(a1 W2^T + b2)) (B, H2), logits = a2 W3^T + b3 (B, K); with labels
 *                    (int64 in [0, K) or -100, nullable) also lse (B + 1 floats, [B] = rows counted) and out = [mean loss,
 *                    accuracy over all B rows, rows counted].  p: dropout probability of both layers; p > 0 reads rng =
 *                    DEVICE int64[2] {seed, step} and ADVANCES step by one (the last workgroup does, after every workgroup has read
 *                    it): masks are a pure function of (seed, step, layer, element), so a step replayed from a hipGraph draws new
 *                    ones.  workspace: sgnn_head_fwd_workspace_bytes(B) bytes, its last 16 bytes ZERO before the first call (a
 *                    ticket the launch leaves zero).  Loss partials are added in workgroup order: bit-reproducible.
 *   sgnn_head_bwd    dlogits = (softmax - onehot) grad_loss[0] / rows[0] (+ grad_logits, nullable) -> dz1 (B, H1) and the
 *                    per-workgroup partials above.  grad_loss nullable (then only grad_logits flows).
 * ------------------------------------------------------------------------------------- */
int sgnn_head_supported(int64_t H1, int64_t H2, int64_t K);
int64_t sgnn_head_blocks(int64_t B);
int64_t sgnn_head_partial_floats(int64_t H1, int64_t H2, int64_t K);
int64_t sgnn_head_fwd_workspace_bytes(int64_t B);
int sgnn_head_fwd(const float* z1, int64_t B, int64_t H1, int64_t H2, int64_t K, const float* W2, const float* b2,
                  const float* W3, const float* b3, const int64_t* labels, float p, int64_t* rng, float* a1, float* a2,
                  float* logits, float* lse, float* out, void* workspace, int64_t workspace_bytes, void* stream);
int sgnn_head_bwd(const float* logits, const float* lse, const int64_t* labels, const float* grad_loss,
                  const float* grad_logits, const float* rows, const float* a1, const float* a2, const float* W2,
                  const float* W3, int64_t B, int64_t H1, int64_t H2, int64_t K, float p, float* dz1, float* partial,
                  void* stream);

/* A^T B for tall operands (the weight gradients of Linear / LSTM layers: outputs of a few thousand elements contracted over
 * thousands of rows -- a library GEMM runs them on a handful of workgroups): job k contracts A[k] (R[k], M[k]) with B[k]
 * (R[k], N[k]), row strides lda / ldb floats, over blocks of rows on the matrix cores (fp32 MFMA, fp32 accumulate) and writes
 * part[k] = (sgnn_contract_rows_blocks(R[k], M[k], N[k]), M[k], N[k]) block partials; sgnn_reduce_partials adds them in block order (a fixed
 * order: bit-reproducible).  colsum_part (nullable; entries nullable): job k also writes (blocks, M[k]) partial COLUMN SUMS of A[k]
 * (a bias gradient rides along with its weight's).  Up to sgnn_contract_rows_max_jobs() jobs per launch.  All arrays are HOST arrays (of DEVICE
 * pointers where they hold pointers).
 * sgnn_reduce_partials: out[k][j] = sum_b part[k][b * n[k] + j] over n_blocks[k] blocks, up to sgnn_reduce_partials_max_jobs()
 * jobs per launch. */
int64_t sgnn_contract_rows_max_jobs(void);
int64_t sgnn_contract_rows_blocks(int64_t R, int64_t M, int64_t N);
int sgnn_contract_rows_partial(int64_t n_jobs, const float* const* A, const float* const* B, const int64_t* lda,
                               const int64_t* ldb, const int64_t* M, const int64_t* N, const int64_t* R,
                               float* const* part, float* const* colsum_part, void* stream);
int64_t sgnn_reduce_partials_max_jobs(void);
int sgnn_reduce_partials(int64_t n_jobs, const float* const* part, const int64_t* n_blocks, const int64_t* n,
                         float* const* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): streaming copy of n_bytes with 4 or 16 bytes per lane.
 * The rocprofv3 memory-side counters (FETCH_SIZE / WRITE_SIZE) are calibrated on it -- a known byte
 * count in the access width of the CSR gather -- before they are read as HBM traffic of
 * sgnn_degree_sequence (tools/degseq_hbm_probe.py, bench.py roofline.traffic).
 * ------------------------------------------------------------------------------------- */
int sgnn_probe_stream_copy(const void* src, void* dst, int64_t n_bytes, int bytes_per_lane, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUBGNN_HIP_H */
