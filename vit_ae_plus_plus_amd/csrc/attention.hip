// Multi-head self-attention core (reference op K7, model/vit.py:117-121):
//   attn = softmax((q @ k^T) * hd^-0.5) ; o = attn @ v      (no mask, no dropout, non-causal)
// on the packed QKV activation [B, N, 3, H, hd] produced by the QKV GEMM (model/vit.py:114) and
// writing o as [B, N, H*hd] (the layout the output projection consumes, model/vit.py:121).
//
// Round-1 kernels: flash-style (never materialises N x N in HBM), fp32 VALU arithmetic, one
// query/key row per TPR lanes with the head dimension split over those lanes, K/V (or Q/dO) tiles
// broadcast from LDS.  Forward keeps a running max / sum per row and emits the log-sum-exp;
// backward recomputes P from (q, k, lse) in two kernels: dQ (+ delta = rowsum(dO * O)) with a
// thread per query row, then dK/dV with a thread per key row.  Attention is ~1-3 % of the path's
// FLOPs; the MFMA version is scheduled after the GEMM work.
#include <cstdlib>
#include "common.hpp"
#include "vitae_hip.h"

namespace {

constexpr int AT_THREADS = 128;
constexpr int AT_KT = 32;   // keys (or queries) staged per LDS tile

template <int TPR>
__device__ __forceinline__ float part_sum(float v) {
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Cooperative load of `rows` rows x HD floats (row stride ld) into LDS [AT_KT][HD].
template <int HD>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, long ld, int rows, float mul) {
    constexpr int V = HD / 4;
    for (int idx = threadIdx.x; idx < rows * V; idx += AT_THREADS) {
        const int j = idx / V, q4 = idx % V;
        f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)j * ld + q4 * 4);
        v *= mul;
        *reinterpret_cast<f32x4*>(dst + j * HD + q4 * 4) = v;
    }
}

template <int HD, int TPR>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                              float* __restrict__ lse, int N, int H, float scale) {
    constexpr int DPT = HD / TPR, ROWS = AT_THREADS / TPR;
    __shared__ __attribute__((aligned(16))) float Ks[AT_KT * HD];
    __shared__ __attribute__((aligned(16))) float Vs[AT_KT * HD];
    __shared__ float S[AT_KT * AT_THREADS];
    const int b = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
    const int row = blockIdx.x * ROWS + tid / TPR, d0 = (tid % TPR) * DPT;
    const int D = H * HD;
    const long ld = 3L * D;
    const float* base = qkv + (long)b * N * ld + h * HD;
    const bool live = row < N;
    float q[DPT], acc[DPT];
#pragma unroll
    for (int d = 0; d < DPT; ++d) {
        q[d] = live ? base[(long)row * ld + d0 + d] * scale : 0.f;
        acc[d] = 0.f;
    }
    float m = -1e30f, l = 0.f;
    for (int c0 = 0; c0 < N; c0 += AT_KT) {
        const int kt = min(AT_KT, N - c0);
        __syncthreads();
        stage_rows<HD>(Ks, base + (long)c0 * ld + D, ld, kt, 1.f);
        stage_rows<HD>(Vs, base + (long)c0 * ld + 2 * D, ld, kt, 1.f);
        __syncthreads();
        float cmax = -1e30f;
        for (int j = 0; j < kt; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DPT; ++d) s += q[d] * Ks[j * HD + d0 + d];
            s = part_sum<TPR>(s);
            S[j * AT_THREADS + tid] = s;
            cmax = fmaxf(cmax, s);
        }
        const float mn = fmaxf(m, cmax);
        const float corr = expf(m - mn);
        l *= corr;
#pragma unroll
        for (int d = 0; d < DPT; ++d) acc[d] *= corr;
        m = mn;
        for (int j = 0; j < kt; ++j) {
            const float p = expf(S[j * AT_THREADS + tid] - m);
            l += p;
#pragma unroll
            for (int d = 0; d < DPT; ++d) acc[d] += p * Vs[j * HD + d0 + d];
        }
    }
    if (live) {
        const float inv = 1.f / l;
        float* orow = o + ((long)b * N + row) * D + h * HD + d0;
#pragma unroll
        for (int d = 0; d < DPT; ++d) orow[d] = acc[d] * inv;
        if (d0 == 0) lse[((long)b * H + h) * N + row] = m + logf(l);
    }
}

template <int HD, int TPR>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                 const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                 float* __restrict__ dqkv, float* __restrict__ delta,
                                                                 int N, int H, float scale) {
    constexpr int DPT = HD / TPR, ROWS = AT_THREADS / TPR;
    __shared__ __attribute__((aligned(16))) float Ks[AT_KT * HD];
    __shared__ __attribute__((aligned(16))) float Vs[AT_KT * HD];
    const int b = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
    const int row = blockIdx.x * ROWS + tid / TPR, d0 = (tid % TPR) * DPT;
    const int D = H * HD;
    const long ld = 3L * D;
    const float* base = qkv + (long)b * N * ld + h * HD;
    const bool live = row < N;
    float q[DPT], g[DPT], dq[DPT];
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < DPT; ++d) {
        const long oi = ((long)b * N + row) * D + h * HD + d0 + d;
        q[d] = live ? base[(long)row * ld + d0 + d] * scale : 0.f;
        g[d] = live ? d_o[oi] : 0.f;
        dl += live ? g[d] * o[oi] : 0.f;
        dq[d] = 0.f;
    }
    dl = part_sum<TPR>(dl);
    const float L = live ? lse[((long)b * H + h) * N + row] : 0.f;
    for (int c0 = 0; c0 < N; c0 += AT_KT) {
        const int kt = min(AT_KT, N - c0);
        __syncthreads();
        stage_rows<HD>(Ks, base + (long)c0 * ld + D, ld, kt, 1.f);
        stage_rows<HD>(Vs, base + (long)c0 * ld + 2 * D, ld, kt, 1.f);
        __syncthreads();
        for (int j = 0; j < kt; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DPT; ++d) {
                s += q[d] * Ks[j * HD + d0 + d];
                dp += g[d] * Vs[j * HD + d0 + d];
            }
            s = part_sum<TPR>(s);
            dp = part_sum<TPR>(dp);
            const float ds = expf(s - L) * (dp - dl);
#pragma unroll
            for (int d = 0; d < DPT; ++d) dq[d] += ds * Ks[j * HD + d0 + d];
        }
    }
    if (live) {
        float* out = dqkv + ((long)b * N + row) * ld + h * HD + d0;
#pragma unroll
        for (int d = 0; d < DPT; ++d) out[d] = dq[d] * scale;
        if (d0 == 0) delta[((long)b * H + h) * N + row] = dl;
    }
}

template <int HD, int TPR>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  float* __restrict__ dqkv, int N, int H, float scale) {
    constexpr int DPT = HD / TPR, ROWS = AT_THREADS / TPR;
    __shared__ __attribute__((aligned(16))) float Qs[AT_KT * HD];
    __shared__ __attribute__((aligned(16))) float Gs[AT_KT * HD];
    __shared__ float Ls[AT_KT], Ds[AT_KT];
    const int b = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
    const int row = blockIdx.x * ROWS + tid / TPR, d0 = (tid % TPR) * DPT;
    const int D = H * HD;
    const long ld = 3L * D;
    const float* base = qkv + (long)b * N * ld + h * HD;
    const bool live = row < N;
    float k[DPT], v[DPT], dk[DPT], dv[DPT];
#pragma unroll
    for (int d = 0; d < DPT; ++d) {
        k[d] = live ? base[(long)row * ld + D + d0 + d] : 0.f;
        v[d] = live ? base[(long)row * ld + 2 * D + d0 + d] : 0.f;
        dk[d] = 0.f; dv[d] = 0.f;
    }
    const float* gbase = d_o + (long)b * N * D + h * HD;
    const float* lrow = lse + ((long)b * H + h) * N;
    const float* drow = delta + ((long)b * H + h) * N;
    for (int c0 = 0; c0 < N; c0 += AT_KT) {
        const int qt = min(AT_KT, N - c0);
        __syncthreads();
        stage_rows<HD>(Qs, base + (long)c0 * ld, ld, qt, scale);
        stage_rows<HD>(Gs, gbase + (long)c0 * D, D, qt, 1.f);
        if (tid < qt) { Ls[tid] = lrow[c0 + tid]; Ds[tid] = drow[c0 + tid]; }
        __syncthreads();
        for (int i = 0; i < qt; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DPT; ++d) {
                s += Qs[i * HD + d0 + d] * k[d];
                dp += Gs[i * HD + d0 + d] * v[d];
            }
            s = part_sum<TPR>(s);
            dp = part_sum<TPR>(dp);
            const float p = expf(s - Ls[i]);
            const float ds = p * (dp - Ds[i]);
#pragma unroll
            for (int d = 0; d < DPT; ++d) {
                dv[d] += p * Gs[i * HD + d0 + d];
                dk[d] += ds * Qs[i * HD + d0 + d];
            }
        }
    }
    if (live) {
        float* out = dqkv + ((long)b * N + row) * ld + h * HD + d0;
#pragma unroll
        for (int d = 0; d < DPT; ++d) { out[D + d] = dk[d]; out[2 * D + d] = dv[d]; }
    }
}

template <int HD, int TPR>
int launch_fwd(const float* qkv, float* o, float* lse, int B, int N, int H, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)HD);
    dim3 grid(cdiv(N, AT_THREADS / TPR), H, B);
    hipLaunchKernelGGL((attn_fwd_kernel<HD, TPR>), grid, dim3(AT_THREADS), 0, st, qkv, o, lse, N, H, scale);
    return vitae_launch_status();
}

template <int HD, int TPR>
int launch_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta,
               int B, int N, int H, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)HD);
    dim3 grid(cdiv(N, AT_THREADS / TPR), H, B);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, TPR>), grid, dim3(AT_THREADS), 0, st, qkv, o, d_o, lse, dqkv, delta, N,
                       H, scale);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, TPR>), grid, dim3(AT_THREADS), 0, st, qkv, d_o, lse, delta, dqkv, N, H,
                       scale);
    return vitae_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same three kernels on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate —
// the arithmetic of the VALU kernels above, 157 TFLOP/s peak instead of an LDS read per FMA).  Head sizes 32 and 64.
//
// A wave owns 32 rows (queries in the forward and dQ kernels, keys in dK/dV) and walks the other axis in tiles of 32 staged in
// LDS; four waves of a workgroup share the staged tiles.  Everything is kept TRANSPOSED with respect to the wave's own rows, so
// that they are the MFMA result's COLUMNS: C[row][col = lane & 31] puts one query (key) per lane, its softmax statistics,
// log-sum-exp and delta are per-lane scalars, and a finished score register is already the B operand (k-slot = lane >> 5, column =
// lane & 31) of the next product — P never moves between registers.  The contraction order inside an MFMA chain is free, so
//   * the head dimension is split in halves by lane >> 5 (step dd of a chain multiplies d = (lane >> 5) * HD/2 + dd): a lane reads
//     its operand row with 16-byte LDS loads, and the per-row fragments (Q, dO; K, V) are HD/2 registers;
//   * the second products contract over the 32 staged rows in the order of the result registers (step r pairs row crow(r, 0) on
//     the lower and crow(r, 1) on the upper half wave), so the A operand is a plain row read of the staged tile.
constexpr int FM_TILE = 32;

// exp on the transcendental unit (v_exp_f32 of x * log2 e; relative error <= 2e-6 over the arguments a softmax produces): ocml's expf is
// ~25 VALU instructions per value, and 16 of them per lane and 32-key tile cost as much as the tile's 32 MFMAs (N = 1729 forward
// 358 -> 299 us).  Measured and NOT kept: workgroups of two waves for short sequences (N = 217: 18.9 vs 18.8 us) and the whole other
// axis resident in LDS without per-tile barriers (17.5 vs 18.9 us) — at N = 217 a workgroup is bound by its own 7 x (32 MFMAs + softmax).
__device__ __forceinline__ float fm_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ int fm_crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// 32 rows x HD floats (global row stride ld, rows >= nrows are zero) -> registers -> LDS [32][HD + 4]
template <int HD> struct FmStage {
    static constexpr int LD = HD + 4, PIECES = FM_TILE * HD / 4 / 256;
    f32x4 r[PIECES];
    __device__ __forceinline__ void load(const float* __restrict__ src, long ld, int row0, int nrows, float mul) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int pidx = threadIdx.x + 256 * j, row = pidx / (HD / 4), c4 = pidx % (HD / 4);
            const bool ok = row0 + row < nrows;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)(ok ? row0 + row : nrows - 1) * ld + c4 * 4);
            r[j] = ok ? v * mul : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void store(float* dst) const {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int pidx = threadIdx.x + 256 * j, row = pidx / (HD / 4), c4 = pidx % (HD / 4);
            *reinterpret_cast<f32x4*>(dst + row * LD + c4 * 4) = r[j];
        }
    }
};

// fragment of this lane's own row: x[row][hi * HD/2 + 0 .. HD/2) (row clamped into the tensor), times mul
template <int HD>
__device__ __forceinline__ void fm_row_frag(float (&f)[HD / 2], const float* __restrict__ src, long ld, int row, int nrows, int hi, float mul) {
    const float* p = src + (long)min(row, nrows - 1) * ld + hi * (HD / 2);
#pragma unroll
    for (int q = 0; q < HD / 8; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[4 * q + e] = v[e] * mul;
    }
}

// C[staged row][own row] += sum_d T[staged row][d] * f[own row][d]  (T: LDS tile [32][HD + 4])
template <int HD>
__device__ __forceinline__ f32x16 fm_dot_rows(const float* T, const float (&f)[HD / 2], int l31, int hi) {
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    const float* row = T + l31 * (HD + 4) + hi * (HD / 2);
#pragma unroll
    for (int q = 0; q < HD / 8; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], f[4 * q + e], c, 0, 0, 0);
    }
    return c;
}

// acc[t][.] (rows d = 32 t + ., columns = own rows) += sum over the 32 staged rows of T[row][d] * w[row][own row]
template <int HD>
__device__ __forceinline__ void fm_accumulate(f32x16 (&acc)[HD / 32], const float* T, const f32x16& w, int l31, int hi) {
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(T[fm_crow(r, hi) * (HD + 4) + 32 * t + l31], w[r], acc[t], 0, 0, 0);
}

// out[own row][32 t + crow(r, hi)] = acc[t][r] * mul: four 16-byte stores per 32 columns
template <int HD>
__device__ __forceinline__ void fm_store_rows(float* __restrict__ out, const f32x16 (&acc)[HD / 32], int hi, float mul) {
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(out + 32 * t + 8 * g + 4 * hi) =
                f32x4{acc[t][4 * g] * mul, acc[t][4 * g + 1] * mul, acc[t][4 * g + 2] * mul, acc[t][4 * g + 3] * mul};
}

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_f32mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                               float* __restrict__ lse, int N, int H, float scale) {
    constexpr int LD = HD + 4;
    __shared__ __attribute__((aligned(16))) float Ks[FM_TILE * LD];
    __shared__ __attribute__((aligned(16))) float Vs[FM_TILE * LD];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int i = blockIdx.x * 128 + wave * 32 + l31;           // this lane's query
    const int D = H * HD;
    const long ld = 3L * D;
    const float* base = qkv + (long)b * N * ld + h * HD;
    float qf[HD / 2];
    fm_row_frag<HD>(qf, base, ld, i, N, hi, scale);
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m = -1e30f, l = 0.f;
    FmStage<HD> sk, sv;
    sk.load(base + D, ld, 0, N, 1.f);
    sv.load(base + 2 * D, ld, 0, N, 1.f);
    for (int c0 = 0; c0 < N; c0 += FM_TILE) {
        __syncthreads();
        sk.store(Ks);
        sv.store(Vs);
        __syncthreads();
        if (c0 + FM_TILE < N) {
            sk.load(base + D, ld, c0 + FM_TILE, N, 1.f);
            sv.load(base + 2 * D, ld, c0 + FM_TILE, N, 1.f);
        }
        f32x16 s = fm_dot_rows<HD>(Ks, qf, l31, hi);          // s[r] = q_i . k_(c0 + crow(r, hi))
        float cmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (c0 + fm_crow(r, hi) >= N) s[r] = -1e30f;
            cmax = fmaxf(cmax, s[r]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float mn = fmaxf(m, cmax);
        const float corr = fm_exp(m - mn);
        m = mn;
        l *= corr;
#pragma unroll
        for (int t = 0; t < HD / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= corr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fm_exp(s[r] - mn);
            l += s[r];
        }
        fm_accumulate<HD>(oacc, Vs, s, l31, hi);
    }
    l += __shfl_xor(l, 32, 64);
    if (i < N) {
        fm_store_rows<HD>(o + ((long)b * N + i) * D + h * HD, oacc, hi, 1.f / l);
        if (hi == 0) lse[((long)b * H + h) * N + i] = m + logf(l);
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_f32mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                  const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                  float* __restrict__ dqkv, float* __restrict__ delta,
                                                                  int N, int H, float scale) {
    constexpr int LD = HD + 4;
    __shared__ __attribute__((aligned(16))) float Ks[FM_TILE * LD];
    __shared__ __attribute__((aligned(16))) float Vs[FM_TILE * LD];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int i = blockIdx.x * 128 + wave * 32 + l31;
    const int D = H * HD;
    const long ld = 3L * D;
    const float* base = qkv + (long)b * N * ld + h * HD;
    float qf[HD / 2], gf[HD / 2];
    fm_row_frag<HD>(qf, base, ld, i, N, hi, scale);
    fm_row_frag<HD>(gf, d_o + (long)b * N * D + h * HD, D, i, N, hi, 1.f);
    float dl = 0.f;
    {
        float of[HD / 2];
        fm_row_frag<HD>(of, o + (long)b * N * D + h * HD, D, i, N, hi, 1.f);
#pragma unroll
        for (int d = 0; d < HD / 2; ++d) dl += gf[d] * of[d];
    }
    dl += __shfl_xor(dl, 32, 64);
    const float L = lse[((long)b * H + h) * N + min(i, N - 1)];
    f32x16 dq[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
    FmStage<HD> sk, sv;
    sk.load(base + D, ld, 0, N, 1.f);
    sv.load(base + 2 * D, ld, 0, N, 1.f);
    for (int c0 = 0; c0 < N; c0 += FM_TILE) {
        __syncthreads();
        sk.store(Ks);
        sv.store(Vs);
        __syncthreads();
        if (c0 + FM_TILE < N) {
            sk.load(base + D, ld, c0 + FM_TILE, N, 1.f);
            sv.load(base + 2 * D, ld, c0 + FM_TILE, N, 1.f);
        }
        f32x16 s = fm_dot_rows<HD>(Ks, qf, l31, hi);
        const f32x16 dp = fm_dot_rows<HD>(Vs, gf, l31, hi);   // dp[r] = dO_i . v_(c0 + crow(r, hi))
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = c0 + fm_crow(r, hi) < N ? fm_exp(s[r] - L) : 0.f;
            s[r] = p * (dp[r] - dl);
        }
        fm_accumulate<HD>(dq, Ks, s, l31, hi);
    }
    if (i < N) {
        fm_store_rows<HD>(dqkv + ((long)b * N + i) * ld + h * HD, dq, hi, scale);
        if (hi == 0) delta[((long)b * H + h) * N + i] = dl;
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_f32mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   float* __restrict__ dqkv, int N, int H, float scale) {
    constexpr int LD = HD + 4;
    __shared__ __attribute__((aligned(16))) float Qs[FM_TILE * LD];
    __shared__ __attribute__((aligned(16))) float Gs[FM_TILE * LD];
    __shared__ __attribute__((aligned(16))) float Ls[FM_TILE], Ds[FM_TILE];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int j = blockIdx.x * 128 + wave * 32 + l31;           // this lane's key
    const int D = H * HD;
    const long ld = 3L * D;
    const float* base = qkv + (long)b * N * ld + h * HD;
    const float* gbase = d_o + (long)b * N * D + h * HD;
    const float* lrow = lse + ((long)b * H + h) * N;
    const float* drow = delta + ((long)b * H + h) * N;
    float kf[HD / 2], vf[HD / 2];
    fm_row_frag<HD>(kf, base + D, ld, j, N, hi, 1.f);
    fm_row_frag<HD>(vf, base + 2 * D, ld, j, N, hi, 1.f);
    f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }
    FmStage<HD> sq, sg;
    float nl = 0.f, nd = 0.f;                                    // threads 0..31 carry the tile's lse / delta
    auto load_stats = [&](int c0) {
        if (threadIdx.x < FM_TILE) {
            const bool ok = c0 + (int)threadIdx.x < N;
            nl = ok ? lrow[c0 + threadIdx.x] : 0.f;
            nd = ok ? drow[c0 + threadIdx.x] : 0.f;
        }
    };
    sq.load(base, ld, 0, N, scale);
    sg.load(gbase, D, 0, N, 1.f);
    load_stats(0);
    for (int c0 = 0; c0 < N; c0 += FM_TILE) {
        __syncthreads();
        sq.store(Qs);
        sg.store(Gs);
        if (threadIdx.x < FM_TILE) { Ls[threadIdx.x] = nl; Ds[threadIdx.x] = nd; }
        __syncthreads();
        if (c0 + FM_TILE < N) {
            sq.load(base, ld, c0 + FM_TILE, N, scale);
            sg.load(gbase, D, c0 + FM_TILE, N, 1.f);
            load_stats(c0 + FM_TILE);
        }
        f32x16 s = fm_dot_rows<HD>(Qs, kf, l31, hi);          // s[r] = q_(c0 + crow(r, hi)) . k_j  (q pre-scaled)
        f32x16 dp = fm_dot_rows<HD>(Gs, vf, l31, hi);         // dp[r] = dO_(c0 + crow(r, hi)) . v_j
        // rows of the tile beyond N were staged as zeros (dO = 0, q = 0, delta = 0): their P and dS multiply zeros below
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + 8 * g + 4 * hi);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = fm_exp(s[4 * g + e] - l4[e]);
                s[4 * g + e] = p;
                dp[4 * g + e] = p * (dp[4 * g + e] - d4[e]);
            }
        }
        fm_accumulate<HD>(dv, Gs, s, l31, hi);                // dV_j += sum_i P_ij dO_i
        fm_accumulate<HD>(dk, Qs, dp, l31, hi);               // dK_j += sum_i dS_ij (scale q_i)
    }
    if (j < N) {
        float* out = dqkv + ((long)b * N + j) * ld + h * HD;
        fm_store_rows<HD>(out + D, dk, hi, 1.f);
        fm_store_rows<HD>(out + 2 * D, dv, hi, 1.f);
    }
}

template <int HD>
int launch_fwd_f32mfma(const float* qkv, float* o, float* lse, int B, int N, int H, hipStream_t st) {
    hipLaunchKernelGGL((attn_fwd_f32mfma_kernel<HD>), dim3(cdiv(N, 128), H, B), dim3(256), 0, st, qkv, o, lse, N, H, 1.0f / sqrtf((float)HD));
    return vitae_launch_status();
}

template <int HD>
int launch_bwd_f32mfma(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta,
                       int B, int N, int H, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)HD);
    const dim3 grid(cdiv(N, 128), H, B);
    hipLaunchKernelGGL((attn_bwd_dq_f32mfma_kernel<HD>), grid, dim3(256), 0, st, qkv, o, d_o, lse, dqkv, delta, N, H, scale);
    hipLaunchKernelGGL((attn_bwd_dkv_f32mfma_kernel<HD>), grid, dim3(256), 0, st, qkv, d_o, lse, delta, dqkv, N, H, scale);
    return vitae_launch_status();
}

// head sizes the matrix-core kernels serve (VITAE_ATTN_F32_MFMA=0: the VALU kernels everywhere, for A/B timing)
bool f32mfma_on() {
    static const bool on = !(getenv("VITAE_ATTN_F32_MFMA") && atoi(getenv("VITAE_ATTN_F32_MFMA")) == 0);
    return on;
}
}  // namespace

extern "C" int vitae_sdpa_fwd(const float* qkv, float* o, float* lse, int B, int N, int H, int head_dim,
                              void* stream) {
    if (!qkv || !o || !lse || B <= 0 || N <= 0 || H <= 0) return VITAE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = !(((uintptr_t)qkv | (uintptr_t)o) & 15);
    if (f32mfma_on() && vec && head_dim == 32) return launch_fwd_f32mfma<32>(qkv, o, lse, B, N, H, st);
    if (f32mfma_on() && vec && head_dim == 64) return launch_fwd_f32mfma<64>(qkv, o, lse, B, N, H, st);
    switch (head_dim) {
        case 16: return launch_fwd<16, 1>(qkv, o, lse, B, N, H, st);
        case 32: return launch_fwd<32, 1>(qkv, o, lse, B, N, H, st);
        case 64: return launch_fwd<64, 2>(qkv, o, lse, B, N, H, st);
        case 128: return launch_fwd<128, 4>(qkv, o, lse, B, N, H, st);
        default: return VITAE_ERR_UNSUPPORTED_SHAPE;
    }
}

extern "C" int vitae_sdpa_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                              float* delta, int B, int N, int H, int head_dim, void* stream) {
    if (!qkv || !o || !d_o || !lse || !dqkv || !delta || B <= 0 || N <= 0 || H <= 0) return VITAE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = !(((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) & 15);
    if (f32mfma_on() && vec && head_dim == 32) return launch_bwd_f32mfma<32>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
    if (f32mfma_on() && vec && head_dim == 64) return launch_bwd_f32mfma<64>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
    switch (head_dim) {
        case 16: return launch_bwd<16, 1>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        case 32: return launch_bwd<32, 1>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        case 64: return launch_bwd<64, 2>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        case 128: return launch_bwd<128, 4>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        default: return VITAE_ERR_UNSUPPORTED_SHAPE;
    }
}
