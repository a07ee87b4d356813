/*
 * oracle_c.c -- plain-C CPU restatement of the heavy integer / fp64 pieces of the SubGNN hot
 * path.  TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, never by the product (subgnn_amd/).
 *
 *   oc_degree_sequence   gamma.get_degree_sequence           (SubGNN/gamma.py:21-49)
 *   oc_fastdtw_sim       gamma.calc_dist / calc_dtw           (SubGNN/gamma.py:51-59) over
 *                        fastdtw==0.3.4, radius 1 -- PARITY UNPINNED (package absent; restated
 *                        from its published algorithm, see oracle/fastdtw_restate.py)
 *   oc_sp_similarity     compute_shortest_path_similarities   (SubGNN/SubGNN.py:752-781)
 *
 * Checked against the numpy/pure-Python oracle (which is pinned by the reference goldens) in
 * tests/test_oracle_c.py.  Build: oracle/Makefile -> oracle/_build/liboracle_c.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* sets are ragged: set_ptr[n_sets+1], set_nodes[]; one output per listed node (duplicates kept) */
void oc_degree_sequence(const int64_t* rowptr, const int32_t* col, const int32_t* full_degree,
                        const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int64_t max_id,
                        int sorted, int32_t* out_int, int32_t* out_ext)
{
    uint8_t* member = (uint8_t*)calloc((size_t)max_id + 2, 1);
    for (int64_t s = 0; s < n_sets; ++s) {
        const int64_t b = set_ptr[s], e = set_ptr[s + 1];
        for (int64_t i = b; i < e; ++i) member[set_nodes[i]] = 1;
        for (int64_t i = b; i < e; ++i) {
            const int32_t v = set_nodes[i];
            int32_t d = 0, selfc = 0;
            for (int64_t k = rowptr[v]; k < rowptr[v + 1]; ++k) {
                const int32_t w = col[k];
                if (w == v) { d += 2; ++selfc; }             /* networkx: a self loop counts twice */
                else if (member[w]) ++d;
            }
            const int32_t full = full_degree ? full_degree[v] : (int32_t)(rowptr[v + 1] - rowptr[v]) + selfc;
            out_int[i] = d;
            if (out_ext) out_ext[i] = full - d;
        }
        for (int64_t i = b; i < e; ++i) member[set_nodes[i]] = 0;
        if (sorted) {
            qsort(out_int + b, (size_t)(e - b), sizeof(int32_t), cmp_i32);
            if (out_ext) qsort(out_ext + b, (size_t)(e - b), sizeof(int32_t), cmp_i32);
        }
    }
    free(member);
}

static double calc_dist(double a, double b) {
    const double mx = a > b ? a : b, mn = a > b ? b : a;
    return (mx + 1.0) / (mn + 1.0) - 1.0;
}

/* windowed DTW; lo/hi per row (hi < lo = empty row).  Writes the warp path's per-row column range
 * into first/last and returns D[lx-1][ly-1].  pred order per tie_order (see fastdtw_restate.py). */
static double dtw_window(const double* x, int lx, const double* y, int ly, const int* lo, const int* hi,
                         int tie_order, int* first, int* last)
{
    const int order[2][3] = {{0, 1, 2}, {2, 0, 1}};      /* 0=(i-1,j) 1=(i,j-1) 2=(i-1,j-1) */
    double* D = (double*)malloc(sizeof(double) * (size_t)lx * ly);
    uint8_t* P = (uint8_t*)malloc((size_t)lx * ly);
    for (int64_t q = 0; q < (int64_t)lx * ly; ++q) D[q] = INFINITY;
    for (int i = 0; i < lx; ++i) {
        for (int j = lo[i]; j <= hi[i]; ++j) {
            const double dt = calc_dist(x[i], y[j]);
            double c[3];
            c[0] = (i > 0) ? D[(int64_t)(i - 1) * ly + j] : INFINITY;
            c[1] = (j > 0) ? D[(int64_t)i * ly + j - 1] : INFINITY;
            c[2] = (i > 0 && j > 0) ? D[(int64_t)(i - 1) * ly + j - 1] : ((i == 0 && j == 0) ? 0.0 : INFINITY);
            int best;
            double bc;
            if (tie_order == 2) {                        /* predecessor costs compared with <=: diagonal, (i-1,j), (i,j-1) */
                best = (c[2] <= c[0] && c[2] <= c[1]) ? 2 : (c[0] <= c[1] ? 0 : 1);
                bc = c[best] + dt;
            } else {
                best = order[tie_order][0];
                bc = c[best] + dt;
                for (int t = 1; t < 3; ++t) {
                    const int o = order[tie_order][t];
                    if (c[o] + dt < bc) { bc = c[o] + dt; best = o; }
                }
            }
            D[(int64_t)i * ly + j] = bc;
            P[(int64_t)i * ly + j] = (uint8_t)best;
        }
    }
    const double res = D[(int64_t)(lx - 1) * ly + ly - 1];
    if (first) {
        for (int i = 0; i < lx; ++i) { first[i] = ly; last[i] = -1; }
        int i = lx - 1, j = ly - 1;
        while (i >= 0 && j >= 0) {
            if (last[i] < j) last[i] = j;
            if (first[i] > j) first[i] = j;
            const int d = P[(int64_t)i * ly + j];
            if (d == 0) --i; else if (d == 1) --j; else { --i; --j; }
        }
    }
    free(D); free(P);
    return res;
}

static double fastdtw_rec(const double* x, int lx, const double* y, int ly, int tie_order, int* first, int* last)
{
    int* lo = (int*)malloc(sizeof(int) * lx);
    int* hi = (int*)malloc(sizeof(int) * lx);
    double res;
    if (lx < 3 || ly < 3) {
        for (int i = 0; i < lx; ++i) { lo[i] = 0; hi[i] = ly - 1; }
        res = dtw_window(x, lx, y, ly, lo, hi, tie_order, first, last);
    } else {
        const int lxc = lx / 2, lyc = ly / 2;
        double* xs = (double*)malloc(sizeof(double) * lxc);
        double* ys = (double*)malloc(sizeof(double) * lyc);
        for (int i = 0; i < lxc; ++i) xs[i] = (x[2 * i] + x[2 * i + 1]) / 2.0;
        for (int i = 0; i < lyc; ++i) ys[i] = (y[2 * i] + y[2 * i + 1]) / 2.0;
        int* cf = (int*)malloc(sizeof(int) * lxc);
        int* cl = (int*)malloc(sizeof(int) * lxc);
        fastdtw_rec(xs, lxc, ys, lyc, tie_order, cf, cl);
        /* expand_window with radius 1 on a monotone path: one contiguous run per fine row */
        int prev_lo = 0;
        for (int i = 0; i < lx; ++i) {
            const int ci = i / 2;
            int ca = ci - 1; if (ca < 0) ca = 0; if (ca > lxc - 1) ca = lxc - 1;
            int cb = ci + 1; if (cb > lxc - 1) cb = lxc - 1;
            int l = 2 * (cf[ca] - 1), h = 2 * (cl[cb] + 1) + 1;
            if (l < prev_lo) l = prev_lo;
            if (l < 0) l = 0;
            if (h > ly - 1) h = ly - 1;
            lo[i] = l; hi[i] = h;
            if (h >= l) prev_lo = l;
        }
        res = dtw_window(x, lx, y, ly, lo, hi, tie_order, first, last);
        free(xs); free(ys); free(cf); free(cl);
    }
    free(lo); free(hi);
    return res;
}

/* out[r*n_y + a] = (float)(1/(1+fastdtw(x_r, y_a))); empty x or y -> 0 (PAD) */
void oc_fastdtw_sim(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x,
                    const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int tie_order, float* out)
{
    /* rows are independent: the host's cores share them (bench.py's cpu_baseline leg; results do not depend on it) */
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < n_x; ++r) {
        const int lx = (int)(x_ptr[r + 1] - x_ptr[r]);
        double* x = (double*)malloc(sizeof(double) * (lx > 0 ? lx : 1));
        for (int i = 0; i < lx; ++i) x[i] = (double)x_val[x_ptr[r] + i];
        for (int64_t a = 0; a < n_y; ++a) {
            const int ly = (int)(y_ptr[a + 1] - y_ptr[a]);
            if (lx == 0 || ly == 0) { out[r * n_y + a] = 0.f; continue; }
            double* y = (double*)malloc(sizeof(double) * ly);
            for (int i = 0; i < ly; ++i) y[i] = (double)y_val[y_ptr[a] + i];
            const double d = fastdtw_rec(x, lx, y, ly, tie_order, NULL, NULL);
            out[r * n_y + a] = (float)(1.0 / (d + 1.0));
            free(y);
        }
        free(x);
    }
}

/* dense-parity shortest-path similarity: out[r, :] = min over members of apsp[v-1, :] */
void oc_sp_similarity(const double* apsp, int64_t n_cols, const int64_t* set_ptr, const int32_t* set_nodes,
                      int64_t n_sets, float* out)
{
    for (int64_t r = 0; r < n_sets; ++r) {
        const int64_t b = set_ptr[r], e = set_ptr[r + 1];
        for (int64_t c = 0; c < n_cols; ++c) {
            float res = 0.f;
            if (e > b) {
                double m = apsp[(int64_t)(set_nodes[b] - 1) * n_cols + c];
                for (int64_t i = b + 1; i < e; ++i) {
                    const double v = apsp[(int64_t)(set_nodes[i] - 1) * n_cols + c];
                    if (v < m) m = v;
                }
                res = (float)m;
            }
            out[r * n_cols + c] = res;
        }
    }
}

/* Position-channel similarities the way the sparse path defines them (subgnn_amd/hotpath.py; values identical to the
 * reference's dense gather, SubGNN/SubGNN.py:763-772): one breadth-first search per source over the CSR graph (ids
 * 1..n, row v of rowptr / col = the neighbours of id v, row 0 empty: oracle/graph.py), then out[set, source] = min over the set's members of the hop count,
 * 0 if some member is unreachable (the reference's matrix holds 0 there and its row-min runs over it).
 * Sources are independent: shared among the host's cores. */
void oc_bfs_min_hops_to_sets(const int64_t* rowptr, const int32_t* col, int64_t n, const int32_t* sources, int64_t n_src,
                             const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, float* out)
{
#pragma omp parallel
    {
        int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
        int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
#pragma omp for schedule(dynamic, 1)
        for (int64_t s = 0; s < n_src; ++s) {
            for (int64_t v = 0; v <= n; ++v) dist[v] = -1;
            int64_t head = 0, tail = 0;
            dist[sources[s]] = 0;
            queue[tail++] = sources[s];
            while (head < tail) {
                const int32_t u = queue[head++];
                for (int64_t e = rowptr[u]; e < rowptr[u + 1]; ++e) {
                    const int32_t w = col[e];
                    if (dist[w] < 0) { dist[w] = dist[u] + 1; queue[tail++] = w; }
                }
            }
            for (int64_t r = 0; r < n_sets; ++r) {
                int32_t m = 0;
                int first = 1, missing = 0;
                for (int64_t i = set_ptr[r]; i < set_ptr[r + 1]; ++i) {
                    const int32_t d = dist[set_nodes[i]];
                    if (d < 0) { missing = 1; break; }
                    if (first || d < m) { m = d; first = 0; }
                }
                out[r * n_src + s] = (missing || first) ? 0.f : (float)m;
            }
        }
        free(dist); free(queue);
    }
}
