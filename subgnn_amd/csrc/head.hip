// a16/a17  The MLP head of SubGNN.forward and the loss of a training step, fused (reference SubGNN/SubGNN.py:304-312:
//   h = dropout(relu(lin(x)));  h = dropout(relu(lin2(h)));  logits = lin3(h)
// and SubGNN.py:1116-1124 nn.CrossEntropyLoss + subgraph_utils.calc_accuracy).
//
// As library calls the head + loss of one step was 9 launches forward (three GEMMs, two clamps, two dropouts, loss, finish) and
// 22 backward (per layer: mask scale, threshold, two GEMMs, a block-sum, two launches for the bias) -- at a batch of 64
// subgraphs every one of them a 4-5 us floor around nanoseconds of arithmetic, and at 50k rows the weight-gradient GEMMs
// (64 x 566 outputs contracted over 50k rows) run on a handful of workgroups.  Here the FIRST layer's two large contractions
// stay dense library GEMMs (z1 = x W1^T + b1 forward, dx = dz1 W1 backward: plain GEMMs over all rows, the caller's), and
// everything behind them is one kernel each way:
//   head_fwd_kernel   per chunk of 32 rows: a1 = drop(relu(z1)), a2 = drop(relu(a1 W2^T + b2)), logits = a2 W3^T + b3, and the
//                     rows' log-sum-exp / loss / hit; W2 and W3 sit transposed in LDS; per-workgroup loss sums, and the LAST
//                     workgroup to finish (a ticket) adds them in workgroup order: no finish launch, bit-reproducible.
//   head_bwd_kernel   dlogits = (softmax - onehot) g / rows, back through both small layers to dz1 (B, H1), and per-workgroup
//                     partial sums of gW3, gb3, gW2, gb2, gb1 (each thread owns fixed elements of its workgroup's slice).
//   contract_rows_partial_kernel   gW1 = dz1^T x: fp32 MFMA (v_mfma_f32_32x32x2_f32) over row blocks, per-block partials -- a
//                     general A^T B for tall operands, several jobs per launch (the LSTM's four weight gradients use it too).
//   partials_reduce_kernel         block partials -> gradients, several jobs per launch, fixed order.
// Dropout is a counter-based mask (common.h's mixer over (seed, step, layer, element)): the step counter lives in device
// memory and is advanced by the forward kernel's last workgroup, so a step replayed from a hipGraph draws fresh masks; what
// the backward needs of a mask is in the saved activations (a == 0 <=> dropped or not activated).  p = 0 touches no random
// state at all.
#include "common.h"

#define HEAD_RB 32
#define HEAD_THREADS 256
#define HEAD_MAX_H 128
#define HEAD_MAX_K 32
#define HEAD_MAX_BLOCKS 1024        // workgroups per launch (each walks its chunks; the backward writes one partial slice per workgroup)
#define HEAD_SALT_2 0x51ED270B3C6EF372ull

struct HeadFwd {
    const float* z1; const float* W2; const float* b2; const float* W3; const float* b3;
    const int64_t* labels; int64_t* rng;
    int64_t B; int H1, H2, K; float p;
    float* a1; float* a2; float* logits; float* lse; float* partial; unsigned* ticket; float* out;
};

struct HeadBwd {
    const float* logits; const float* lse; const int64_t* labels; const float* g_loss; const float* g_logits; const float* rows;
    const float* a1; const float* a2; const float* W2; const float* W3;
    int64_t B; int H1, H2, K; float scale;
    float* dz1; float* partial;
};

__device__ __forceinline__ bool head_keep(uint64_t h0, uint64_t idx, float p)
{
    const uint32_t u = (uint32_t)(sgnn_tape_h1(h0, idx) >> 40);           // 24 random bits
    return (float)u * (1.f / 16777216.f) >= p;
}

// LDS layouts (floats; widths padded to multiples of 4 -- `p` -- so that every contraction reads float4s; pad columns are zero):
//   forward   W2s [H2e][H1p + 4] (row-major, H2e = H2 rounded up to 2: a thread computes two output columns), W3s [K][H2p + 4],
//             a1s [RB][H1p], a2s [RB][H2p], lgs [RB][K]
//   backward  W2t [H1e][H2p + 4] (TRANSPOSED: W2t[k][j]), W3s [K][H2p], dls [RB][K], dz2s / a2s [RB][H2p], a1s / dz1s [RB][H1p]
// A thread of the two H1 x H2 contractions owns a 4-row x 2-column block of the output: per four contraction positions it reads
// six float4s from LDS for 32 fused multiply-adds (one output per thread with two LDS reads per multiply-add ran at a seventh
// of the update layer's rate: 88 us forward at 50k rows).  The +4 row padding spreads a wavefront's float4 reads over the banks.
__device__ __forceinline__ float head_dot4(const float4 a, const float4 b, float acc)
{
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); return fmaf(a.w, b.w, acc);
}

__global__ __launch_bounds__(HEAD_THREADS) void head_fwd_kernel(const HeadFwd A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H1 = A.H1, H2 = A.H2, K = A.K, tid = threadIdx.x;
    const int H1p = (H1 + 3) & ~3, H2p = (H2 + 3) & ~3, H2e = (H2 + 1) & ~1, ld2 = H1p + 4, ld3 = H2p + 4;
    float* W2s = smem;                                   // [H2e][ld2]
    float* W3s = W2s + H2e * ld2;                        // [K][ld3]
    float* a1s = W3s + K * ld3;                          // [RB][H1p]
    float* a2s = a1s + HEAD_RB * H1p;                    // [RB][H2p]
    float* lgs = a2s + HEAD_RB * H2p;                    // [RB][K]
    __shared__ int s_last;
    for (int idx = tid; idx < H2e * ld2; idx += HEAD_THREADS) {
        const int j = idx / ld2, k = idx - j * ld2;
        W2s[idx] = (j < H2 && k < H1) ? A.W2[j * H1 + k] : 0.f;
    }
    for (int idx = tid; idx < K * ld3; idx += HEAD_THREADS) {
        const int c = idx / ld3, k = idx - c * ld3;
        W3s[idx] = k < H2 ? A.W3[c * H2 + k] : 0.f;
    }
    for (int idx = tid; idx < HEAD_RB * (H1p + H2p); idx += HEAD_THREADS) a1s[idx] = 0.f;      // (a1s and a2s are adjacent: pads stay zero)
    const bool drop = A.p > 0.f && A.rng != nullptr;
    uint64_t h01 = 0, h02 = 0;
    if (drop) {
        const uint64_t seed = (uint64_t)A.rng[0], step = (uint64_t)A.rng[1];
        h01 = sgnn_tape_h0(seed, step);
        h02 = sgnn_tape_h0(seed ^ HEAD_SALT_2, step);
    }
    const float keep_scale = drop ? 1.f / (1.f - A.p) : 1.f;
    float loss = 0.f, hit = 0.f, cnt = 0.f;
    __syncthreads();
    const int64_t n_chunks = (A.B + HEAD_RB - 1) / HEAD_RB;
    const int ncp = H2e / 2;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t row0 = chunk * HEAD_RB;
        const int nr = (int)(A.B - row0 < HEAD_RB ? A.B - row0 : HEAD_RB);
        for (int idx = tid; idx < HEAD_RB * H1; idx += HEAD_THREADS) {
            const int r = idx / H1, k = idx - r * H1;
            float v = 0.f;
            if (r < nr) {
                const int64_t e = (row0 + r) * H1 + k;
                v = fmaxf(A.z1[e], 0.f);
                if (drop) v = head_keep(h01, (uint64_t)e, A.p) ? v * keep_scale : 0.f;
                A.a1[e] = v;
            }
            a1s[r * H1p + k] = v;
        }
        __syncthreads();
        for (int item = tid; item < (HEAD_RB / 4) * ncp; item += HEAD_THREADS) {
            const int rg = item / ncp, cp = item - rg * ncp;
            float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            const float* ar = a1s + 4 * rg * H1p;
            const float* w0 = W2s + 2 * cp * ld2;
            const float* w1 = w0 + ld2;
            for (int k = 0; k < H1p; k += 4) {
                const float4 wa = *reinterpret_cast<const float4*>(w0 + k), wb = *reinterpret_cast<const float4*>(w1 + k);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 av = *reinterpret_cast<const float4*>(ar + r * H1p + k);
                    acc[r][0] = head_dot4(av, wa, acc[r][0]);
                    acc[r][1] = head_dot4(av, wb, acc[r][1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int row = 4 * rg + r, j = 2 * cp + c;
                    if (row < nr && j < H2) {
                        float v = fmaxf(acc[r][c] + (A.b2 ? A.b2[j] : 0.f), 0.f);
                        const int64_t e = (row0 + row) * H2 + j;
                        if (drop) v = head_keep(h02, (uint64_t)e, A.p) ? v * keep_scale : 0.f;
                        A.a2[e] = v;
                        a2s[row * H2p + j] = v;
                    }
                }
        }
        __syncthreads();
        for (int idx = tid; idx < nr * K; idx += HEAD_THREADS) {
            const int r = idx / K, c = idx - r * K;
            float acc = 0.f;
            const float* ar = a2s + r * H2p;
            const float* wr = W3s + c * ld3;
            for (int k = 0; k < H2p; k += 4)
                acc = head_dot4(*reinterpret_cast<const float4*>(ar + k), *reinterpret_cast<const float4*>(wr + k), acc);
            acc += A.b3 ? A.b3[c] : 0.f;
            A.logits[row0 * K + idx] = acc;
            lgs[idx] = acc;
        }
        __syncthreads();
        if (A.labels && tid < nr) {
            const float* x = lgs + tid * K;
            float m = x[0];
            int am = 0;
            for (int k = 1; k < K; ++k) { const float v = x[k]; if (v > m) { m = v; am = k; } }       // first maximum, as argmax
            float sum = 0.f;
            for (int k = 0; k < K; ++k) sum += expf(x[k] - m);
            const float l = m + logf(sum);
            A.lse[row0 + tid] = l;
            const int64_t y = A.labels[row0 + tid];
            if (y >= 0 && y < K) { loss += l - x[y]; hit += (am == (int)y) ? 1.f : 0.f; cnt += 1.f; }
            else if (y != -100ll) { loss = __builtin_nanf(""); cnt += 1.f; }
        }
        __syncthreads();
    }
    if (!A.labels && !drop) return;
    if (tid < 64) {                                      // (rows of a chunk sit in lanes 0..31 of wavefront 0)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { loss += __shfl_xor(loss, o, 64); hit += __shfl_xor(hit, o, 64); cnt += __shfl_xor(cnt, o, 64); }
        if (tid == 0) {
            A.partial[3 * blockIdx.x] = loss; A.partial[3 * blockIdx.x + 1] = hit; A.partial[3 * blockIdx.x + 2] = cnt;
            __threadfence();
            s_last = (atomicAdd(A.ticket, 1u) == gridDim.x - 1) ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_last || tid >= 64) return;
    __threadfence();
    const volatile float* part = A.partial;
    float a = 0.f, h = 0.f, n = 0.f;
    for (unsigned k = tid; k < gridDim.x; k += 64) { a += part[3 * k]; h += part[3 * k + 1]; n += part[3 * k + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); h += __shfl_xor(h, o, 64); n += __shfl_xor(n, o, 64); }
    if (tid == 0) {
        if (A.labels) {
            A.out[0] = a / n;                            // every row ignored: 0 / 0 = NaN, as the library gives
            A.out[1] = h / (float)A.B;
            A.out[2] = n;
            A.lse[A.B] = n;
        }
        if (drop) A.rng[1] += 1;                         // every workgroup has read the step: the next launch draws new masks
        *A.ticket = 0u;
    }
}

// partial layout of one workgroup: [gW3 (K H2) | gb3 (K) | gW2 (H2 H1) | gb2 (H2) | gb1 (H1)]
__host__ __device__ static inline int64_t head_partial_floats(int H1, int H2, int K) { return (int64_t)K * H2 + K + (int64_t)H2 * H1 + H2 + H1; }

// NB: 4 x 4 blocks of gW2 a thread keeps in registers (1: up to 64 x 64 weights, 2: 128 x 64, 4: 128 x 128)
template <int NB>
__global__ __launch_bounds__(HEAD_THREADS) void head_bwd_kernel(const HeadBwd A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H1 = A.H1, H2 = A.H2, K = A.K, tid = threadIdx.x;
    const int H1p = (H1 + 3) & ~3, H2p = (H2 + 3) & ~3, H1e = (H1 + 1) & ~1, ldt = H2p + 4;
    float* W2t = smem;                                   // [H1e][ldt]: W2t[k][j] = W2[j][k]
    float* W3s = W2t + H1e * ldt;                        // [K][H2p]
    float* dls = W3s + K * H2p;                          // [RB][K]
    float* dz2s = dls + HEAD_RB * K;                     // [RB][H2p]
    float* a2s = dz2s + HEAD_RB * H2p;                   // [RB][H2p]
    float* a1s = a2s + HEAD_RB * H2p;                    // [RB][H1p]
    float* dz1s = a1s + HEAD_RB * H1p;                   // [RB][H1p]
    for (int idx = tid; idx < H1e * ldt; idx += HEAD_THREADS) {
        const int k = idx / ldt, j = idx - k * ldt;
        W2t[idx] = (k < H1 && j < H2) ? A.W2[j * H1 + k] : 0.f;
    }
    for (int idx = tid; idx < K * H2p; idx += HEAD_THREADS) {
        const int c = idx / H2p, k = idx - c * H2p;
        W3s[idx] = k < H2 ? A.W3[c * H2 + k] : 0.f;
    }
    for (int idx = tid; idx < HEAD_RB * (2 * H2p + 2 * H1p); idx += HEAD_THREADS) dz2s[idx] = 0.f;    // (adjacent: all pads zero)
    const int64_t P = head_partial_floats(H1, H2, K);
    float* part = A.partial + (int64_t)blockIdx.x * P;
    const float gscale = (A.labels && A.g_loss) ? A.g_loss[0] / A.rows[0] : 0.f;
    const int64_t n_chunks = (A.B + HEAD_RB - 1) / HEAD_RB;
    const int nkp = H1e / 2, nbk = H1p / 4, nblocks = (H2p / 4) * nbk;
    const int n_small = K * H2 + K + H2 + H1;
    const int64_t oW2 = (int64_t)K * H2 + K, ob2 = oW2 + (int64_t)H2 * H1;
    float accW[NB][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) accW[nb][v] = 0.f;
    bool first = true;
    __syncthreads();
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t row0 = chunk * HEAD_RB;
        const int nr = (int)(A.B - row0 < HEAD_RB ? A.B - row0 : HEAD_RB);
        for (int idx = tid; idx < nr * K; idx += HEAD_THREADS) {
            const int r = idx / K, c = idx - r * K;
            float v = 0.f;
            if (A.labels && A.g_loss) {
                const int64_t y = A.labels[row0 + r];
                if (y >= 0 && y < K) v = (expf(A.logits[row0 * K + idx] - A.lse[row0 + r]) - (c == (int)y ? 1.f : 0.f)) * gscale;
            }
            if (A.g_logits) v += A.g_logits[row0 * K + idx];
            dls[idx] = v;
        }
        for (int idx = tid; idx < nr * H2; idx += HEAD_THREADS) { const int r = idx / H2; a2s[r * H2p + idx - r * H2] = A.a2[row0 * H2 + idx]; }
        for (int idx = tid; idx < nr * H1; idx += HEAD_THREADS) { const int r = idx / H1; a1s[r * H1p + idx - r * H1] = A.a1[row0 * H1 + idx]; }
        __syncthreads();
        for (int idx = tid; idx < nr * H2; idx += HEAD_THREADS) {
            const int r = idx / H2, k = idx - r * H2;
            float acc = 0.f;
            for (int c = 0; c < K; ++c) acc = fmaf(dls[r * K + c], W3s[c * H2p + k], acc);
            dz2s[r * H2p + k] = a2s[r * H2p + k] > 0.f ? acc * A.scale : 0.f;
        }
        __syncthreads();
        for (int item = tid; item < (HEAD_RB / 4) * nkp; item += HEAD_THREADS) {
            const int rg = item / nkp, kp = item - rg * nkp;
            if (4 * rg >= nr) continue;
            float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            const float* dr = dz2s + 4 * rg * H2p;
            const float* w0 = W2t + 2 * kp * ldt;
            const float* w1 = w0 + ldt;
            for (int j = 0; j < H2p; j += 4) {
                const float4 wa = *reinterpret_cast<const float4*>(w0 + j), wb = *reinterpret_cast<const float4*>(w1 + j);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 dv = *reinterpret_cast<const float4*>(dr + r * H2p + j);
                    acc[r][0] = head_dot4(dv, wa, acc[r][0]);
                    acc[r][1] = head_dot4(dv, wb, acc[r][1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int row = 4 * rg + r, k = 2 * kp + c;
                    if (row < nr && k < H1) {
                        const float v = a1s[row * H1p + k] > 0.f ? acc[r][c] * A.scale : 0.f;
                        A.dz1[(row0 + row) * H1 + k] = v;
                        dz1s[row * H1p + k] = v;
                    }
                }
        }
        __syncthreads();
        // gW2 += dz2^T a1 over the chunk's rows: a thread's 4 x 4 blocks live in registers for the whole launch
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int b = tid + HEAD_THREADS * nb;
            if (b < nblocks) {
                const int jb = b / nbk, kb = b - jb * nbk;
                for (int r = 0; r < nr; ++r) {
                    const float4 dz = *reinterpret_cast<const float4*>(dz2s + r * H2p + 4 * jb);
                    const float4 av = *reinterpret_cast<const float4*>(a1s + r * H1p + 4 * kb);
                    const float d4[4] = {dz.x, dz.y, dz.z, dz.w}, a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) accW[nb][4 * jj + kk] = fmaf(d4[jj], a4[kk], accW[nb][4 * jj + kk]);
                }
            }
        }
        // the small gradients [gW3 | gb3 | gb2 | gb1]: element e belongs to thread e % 256 for the whole launch, its running sum
        // sits in the workgroup's own slice of the partials (a few elements per thread: not worth registers)
        for (int e = tid; e < n_small; e += HEAD_THREADS) {
            int q = e;
            float s = 0.f;
            if (q < K * H2) {
                const int c = q / H2, k = q - c * H2;
                for (int r = 0; r < nr; ++r) s = fmaf(dls[r * K + c], a2s[r * H2p + k], s);
            } else if ((q -= K * H2) < K) {
                for (int r = 0; r < nr; ++r) s += dls[r * K + q];
            } else if ((q -= K) < H2) {
                for (int r = 0; r < nr; ++r) s += dz2s[r * H2p + q];
            } else {
                q -= H2;
                for (int r = 0; r < nr; ++r) s += dz1s[r * H1p + q];
            }
            float* dst = part + (e < K * H2 + K ? e : ob2 + (e - (K * H2 + K)));
            *dst = first ? s : *dst + s;
        }
        first = false;
        __syncthreads();
    }
    // the gW2 blocks of this workgroup's slice
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = tid + HEAD_THREADS * nb;
        if (b < nblocks) {
            const int jb = b / nbk, kb = b - jb * nbk;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int j = 4 * jb + jj, k = 4 * kb + kk;
                    if (j < H2 && k < H1) part[oW2 + (int64_t)j * H1 + k] = accW[nb][4 * jj + kk];
                }
        }
    }
}

// ---- A^T B for tall operands: partial sums over row blocks (fp32 MFMA) ----------------------------------------------------------
// part[blk][m][n] = sum over the block's rows r of A[r][m] B[r][n].  blockIdx.x = row block (four wavefronts of `wr` rows each),
// blockIdx.y = (tile of 32 m, group of CR_NT tiles of 32 n), blockIdx.z = job.  Step s of a wavefront covers rows r0 + 2 s +
// (lane / 32); CR_STEPS steps' operands are requested together.  The four wavefronts' sums are added in wavefront order through
// LDS: a fixed order, so are the blocks in partials_reduce_kernel.
#define CR_NT 4
#define CR_STEPS 8
#define CR_MAX_JOBS 8
typedef float cr_f32x16 __attribute__((ext_vector_type(16)));
struct CrJobs {
    const float* A[CR_MAX_JOBS]; const float* Bm[CR_MAX_JOBS]; float* part[CR_MAX_JOBS];
    float* colsum[CR_MAX_JOBS];          // nullable: (blocks, M) partial column sums of A (a bias gradient rides along with its weight's)
    long long lda[CR_MAX_JOBS], ldb[CR_MAX_JOBS], R[CR_MAX_JOBS];
    int M[CR_MAX_JOBS], N[CR_MAX_JOBS], wr[CR_MAX_JOBS];
};

__device__ __forceinline__ int cr_acc_row(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

__global__ __launch_bounds__(256) void contract_rows_partial_kernel(const CrJobs J)
{
    const int z = blockIdx.z;
    const int M = J.M[z], N = J.N[z], wr = J.wr[z];
    const int64_t R = J.R[z];
    const int mtiles = (M + 31) / 32, ntiles = (N + 31) / 32, ngroups = (ntiles + CR_NT - 1) / CR_NT;
    if ((int64_t)blockIdx.x * 4 * wr >= R || (int)blockIdx.y >= mtiles * ngroups) return;
    const int mt = blockIdx.y % mtiles, ng = blockIdx.y / mtiles;
    __shared__ float s_acc[CR_NT * 16 * 64];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* __restrict__ A = J.A[z];
    const float* __restrict__ Bm = J.Bm[z];
    const int64_t lda = J.lda[z], ldb = J.ldb[z];
    const int64_t r0 = (int64_t)blockIdx.x * 4 * wr + (int64_t)wave * wr;
    const int64_t r_end = r0 + wr < R ? r0 + wr : R;
    const int m = mt * 32 + i;
    const bool m_ok = m < M;
    cr_f32x16 acc[CR_NT];
#pragma unroll
    for (int c = 0; c < CR_NT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    float bsum = 0.f;                                        // column sum of A over this wavefront's rows (used by n-group 0 only)
    for (int64_t rb = r0; rb < r_end; rb += 2 * CR_STEPS) {
        float av[CR_STEPS], bv[CR_STEPS][CR_NT];
#pragma unroll
        for (int u = 0; u < CR_STEPS; ++u) {
            const int64_t r = rb + 2 * u + h;
            const bool live = r < r_end;
            av[u] = (live && m_ok) ? A[r * lda + m] : 0.f;
#pragma unroll
            for (int c = 0; c < CR_NT; ++c) {
                const int n = (ng * CR_NT + c) * 32 + i;
                bv[u][c] = (live && n < N) ? Bm[r * ldb + n] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < CR_STEPS; ++u) {
            bsum += av[u];
#pragma unroll
            for (int c = 0; c < CR_NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][c], acc[c], 0, 0, 0);
        }
    }
    bsum += __shfl_xor(bsum, 32, 64);                        // the two row halves
    __shared__ float s_b[64];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < CR_NT; ++c)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int idx = (c * 16 + v) * 64 + lane;
                    if (w > 0) acc[c][v] += s_acc[idx];
                    if (w < 3) s_acc[idx] = acc[c][v];
                }
            if (w > 0) bsum += s_b[lane];
            if (w < 3) s_b[lane] = bsum;
        }
        __syncthreads();
    }
    if (wave != 3) return;
    if (J.colsum[z] && ng == 0 && h == 0 && m_ok) J.colsum[z][(int64_t)blockIdx.x * M + m] = bsum;
    float* __restrict__ dst = J.part[z] + (int64_t)blockIdx.x * M * N;
#pragma unroll
    for (int c = 0; c < CR_NT; ++c) {
        const int n = (ng * CR_NT + c) * 32 + i;
        if (n >= N) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int mm = mt * 32 + cr_acc_row(v, h);
            if (mm < M) dst[(int64_t)mm * N + n] = acc[c][v];
        }
    }
}

// out[j] = sum over blocks of part[block * n + j], in a fixed order: a workgroup owns 64 outputs of one job; its four wavefronts
// each add up a contiguous quarter of the blocks, and the quarters are added in order through LDS
#define PR_MAX_JOBS 8
struct PrJobs {
    const float* part[PR_MAX_JOBS]; float* out[PR_MAX_JOBS];
    long long n_blocks[PR_MAX_JOBS], n[PR_MAX_JOBS];
    unsigned first_group[PR_MAX_JOBS + 1];
    int n_jobs;
};

__global__ __launch_bounds__(256) void partials_reduce_kernel(const PrJobs J)
{
    __shared__ float s_q[4 * 64];
    int job = 0;
    while (job + 1 < J.n_jobs && blockIdx.x >= J.first_group[job + 1]) ++job;
    const float* __restrict__ part = J.part[job];
    const int64_t n = J.n[job], n_blocks = J.n_blocks[job];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t j = (int64_t)(blockIdx.x - J.first_group[job]) * 64 + o;
    const int64_t per = (n_blocks + 3) / 4;
    const int64_t b0 = q * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    float s = 0.f;
    if (j < n) {
#pragma unroll 8
        for (int64_t b = b0; b < b1; ++b) s += part[b * n + j];
    }
    s_q[q * 64 + o] = s;
    __syncthreads();
    if (q == 0 && j < n) J.out[job][j] = ((s_q[o] + s_q[64 + o]) + s_q[128 + o]) + s_q[192 + o];
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
extern "C" int sgnn_head_supported(int64_t H1, int64_t H2, int64_t K)
{
    return (H1 >= 1 && H1 <= HEAD_MAX_H && H2 >= 1 && H2 <= HEAD_MAX_H && K >= 1 && K <= HEAD_MAX_K) ? 1 : 0;
}

static inline int64_t head_blocks(int64_t B)
{
    const int64_t chunks = (B + HEAD_RB - 1) / HEAD_RB;
    return chunks < 1 ? 1 : (chunks > HEAD_MAX_BLOCKS ? HEAD_MAX_BLOCKS : chunks);
}

extern "C" int64_t sgnn_head_blocks(int64_t B) { return B < 0 ? -1 : head_blocks(B); }

extern "C" int64_t sgnn_head_partial_floats(int64_t H1, int64_t H2, int64_t K)
{
    return sgnn_head_supported(H1, H2, K) ? head_partial_floats((int)H1, (int)H2, (int)K) : -1;
}

/* workspace of the forward: 3 floats per workgroup + the ticket (4 bytes, at the end: ZERO before the first call, left zero) */
extern "C" int64_t sgnn_head_fwd_workspace_bytes(int64_t B) { return B < 0 ? -1 : 3 * head_blocks(B) * 4 + 16; }

static int head_set_lds(const void* kernel, size_t bytes)
{
    if (bytes > 64 * 1024) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -1;
    }
    return 0;
}

extern "C" int sgnn_head_fwd(const float* z1, int64_t B, int64_t H1, int64_t H2, int64_t K, const float* W2, const float* b2,
                             const float* W3, const float* b3, const int64_t* labels, float p, int64_t* rng, float* a1, float* a2,
                             float* logits, float* lse, float* out, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!z1 || !W2 || !W3 || !a1 || !a2 || !logits || B < 1 || !(p >= 0.f && p < 1.f)) return SGNN_ERR_BAD_ARG;
    if (!sgnn_head_supported(H1, H2, K)) return SGNN_ERR_UNSUPPORTED_D;
    if (labels && (!lse || !out)) return SGNN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < sgnn_head_fwd_workspace_bytes(B)) return SGNN_ERR_BAD_ARG;
    HeadFwd A;
    A.z1 = z1; A.W2 = W2; A.b2 = b2; A.W3 = W3; A.b3 = b3; A.labels = labels; A.rng = (p > 0.f) ? rng : nullptr;
    A.B = B; A.H1 = (int)H1; A.H2 = (int)H2; A.K = (int)K; A.p = p;
    A.a1 = a1; A.a2 = a2; A.logits = logits; A.lse = lse; A.out = out;
    const int64_t nb = head_blocks(B);
    A.partial = (float*)workspace;
    A.ticket = (unsigned*)((char*)workspace + 3 * nb * 4);
    const int64_t H1p = (H1 + 3) & ~3ll, H2p = (H2 + 3) & ~3ll, H2e = (H2 + 1) & ~1ll;
    const size_t lds = (size_t)(H2e * (H1p + 4) + K * (H2p + 4) + HEAD_RB * (H1p + H2p + K)) * 4;
    if (head_set_lds((const void*)head_fwd_kernel, lds) != 0) return SGNN_ERR_LAUNCH;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)nb), dim3(HEAD_THREADS), lds, (hipStream_t)stream, A);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_head_bwd(const float* logits, const float* lse, const int64_t* labels, const float* grad_loss,
                             const float* grad_logits, const float* rows, const float* a1, const float* a2, const float* W2,
                             const float* W3, int64_t B, int64_t H1, int64_t H2, int64_t K, float p, float* dz1, float* partial,
                             void* stream)
{
    if (!a1 || !a2 || !W2 || !W3 || !dz1 || !partial || B < 1 || !(p >= 0.f && p < 1.f)) return SGNN_ERR_BAD_ARG;
    if (!sgnn_head_supported(H1, H2, K)) return SGNN_ERR_UNSUPPORTED_D;
    if (grad_loss && (!labels || !logits || !lse || !rows)) return SGNN_ERR_BAD_ARG;
    HeadBwd A;
    A.logits = logits; A.lse = lse; A.labels = labels; A.g_loss = grad_loss; A.g_logits = grad_logits; A.rows = rows;
    A.a1 = a1; A.a2 = a2; A.W2 = W2; A.W3 = W3; A.B = B; A.H1 = (int)H1; A.H2 = (int)H2; A.K = (int)K;
    A.scale = 1.f / (1.f - p);
    A.dz1 = dz1; A.partial = partial;
    const int64_t H1p = (H1 + 3) & ~3ll, H2p = (H2 + 3) & ~3ll, H1e = (H1 + 1) & ~1ll;
    const size_t lds = (size_t)(H1e * (H2p + 4) + K * H2p + HEAD_RB * (K + 2 * H2p + 2 * H1p)) * 4;
    const int64_t nblocks = (H2p / 4) * (H1p / 4);
#define HEAD_BWD(NB) do { if (head_set_lds((const void*)head_bwd_kernel<NB>, lds) != 0) return SGNN_ERR_LAUNCH; \
                          hipLaunchKernelGGL(head_bwd_kernel<NB>, dim3((unsigned)head_blocks(B)), dim3(HEAD_THREADS), lds, (hipStream_t)stream, A); } while (0)
    if (nblocks <= HEAD_THREADS) HEAD_BWD(1); else if (nblocks <= 2 * HEAD_THREADS) HEAD_BWD(2); else HEAD_BWD(4);
#undef HEAD_BWD
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// rows per wavefront of a job: enough row blocks that blocks x output tiles fill the chip twice (~512 workgroups), no more -- every
// block writes an M x N partial that the reduction reads back (a 512 x 128 output over 3 700 rows in 64-row blocks was 58
// partials of 256 KB: 15 MB written and read for a 1 MB result) -- and at most 128 rows (16 a multiple: the step of the loop)
static inline int cr_wave_rows(int64_t R, int64_t M, int64_t N)
{
    const int64_t tiles = ((M + 31) / 32) * ((((N + 31) / 32) + CR_NT - 1) / CR_NT);
    int64_t blocks = 512 / (tiles < 1 ? 1 : tiles);
    if (blocks < 1) blocks = 1;
    int64_t wr = (R + 4 * blocks - 1) / (4 * blocks);
    wr = (wr + 15) / 16 * 16;
    return (int)(wr < 16 ? 16 : (wr > 128 ? 128 : wr));
}

extern "C" int64_t sgnn_contract_rows_max_jobs(void) { return CR_MAX_JOBS; }
extern "C" int64_t sgnn_contract_rows_blocks(int64_t R, int64_t M, int64_t N)
{
    if (R < 0 || M < 1 || N < 1) return -1;
    const int64_t per = 4 * cr_wave_rows(R, M, N);
    return (R + per - 1) / per;
}

extern "C" int sgnn_contract_rows_partial(int64_t n_jobs, const float* const* A, const float* const* Bm, const int64_t* lda,
                                          const int64_t* ldb, const int64_t* M, const int64_t* N, const int64_t* R,
                                          float* const* part, float* const* colsum_part, void* stream)
{
    if (n_jobs < 1 || n_jobs > CR_MAX_JOBS || !A || !Bm || !lda || !ldb || !M || !N || !R || !part) return SGNN_ERR_BAD_ARG;
    CrJobs J;
    unsigned gx = 0, gy = 0;
    for (int k = 0; k < n_jobs; ++k) {
        if (!A[k] || !Bm[k] || !part[k] || M[k] < 1 || N[k] < 1 || M[k] > (1 << 20) || N[k] > (1 << 20) || R[k] < 0 || lda[k] < M[k] || ldb[k] < N[k])
            return SGNN_ERR_BAD_ARG;
        J.A[k] = A[k]; J.Bm[k] = Bm[k]; J.part[k] = part[k]; J.lda[k] = lda[k]; J.ldb[k] = ldb[k]; J.R[k] = R[k];
        J.colsum[k] = colsum_part ? colsum_part[k] : nullptr;
        J.M[k] = (int)M[k]; J.N[k] = (int)N[k]; J.wr[k] = cr_wave_rows(R[k], M[k], N[k]);
        const int64_t nb = sgnn_contract_rows_blocks(R[k], M[k], N[k]);
        const int64_t tiles = ((M[k] + 31) / 32) * ((((N[k] + 31) / 32) + CR_NT - 1) / CR_NT);
        if (nb > 0x7fffffff || tiles > 65535) return SGNN_ERR_BAD_ARG;
        if ((unsigned)nb > gx) gx = (unsigned)nb;
        if ((unsigned)tiles > gy) gy = (unsigned)tiles;
    }
    for (int k = (int)n_jobs; k < CR_MAX_JOBS; ++k) { J.A[k] = J.Bm[k] = nullptr; J.part[k] = nullptr; J.colsum[k] = nullptr; J.lda[k] = J.ldb[k] = J.R[k] = 0; J.M[k] = J.N[k] = 0; J.wr[k] = 16; }
    if (gx == 0) return SGNN_OK;
    hipLaunchKernelGGL(contract_rows_partial_kernel, dim3(gx, gy, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, J);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int64_t sgnn_reduce_partials_max_jobs(void) { return PR_MAX_JOBS; }

extern "C" int sgnn_reduce_partials(int64_t n_jobs, const float* const* part, const int64_t* n_blocks, const int64_t* n,
                                    float* const* out, void* stream)
{
    if (n_jobs < 1 || n_jobs > PR_MAX_JOBS || !part || !n_blocks || !n || !out) return SGNN_ERR_BAD_ARG;
    PrJobs J;
    J.n_jobs = (int)n_jobs;
    unsigned groups = 0;
    for (int k = 0; k < n_jobs; ++k) {
        if (!part[k] || !out[k] || n_blocks[k] < 0 || n[k] < 0 || (n[k] + 63) / 64 > 0x3fffffff) return SGNN_ERR_BAD_ARG;
        J.part[k] = part[k]; J.out[k] = out[k]; J.n_blocks[k] = n_blocks[k]; J.n[k] = n[k];
        J.first_group[k] = groups;
        groups += (unsigned)((n[k] + 63) / 64);
    }
    for (int k = (int)n_jobs; k < PR_MAX_JOBS; ++k) { J.part[k] = nullptr; J.out[k] = nullptr; J.n_blocks[k] = J.n[k] = 0; J.first_group[k] = groups; }
    J.first_group[PR_MAX_JOBS] = groups;
    if (groups == 0) return SGNN_OK;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, J);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(head)
