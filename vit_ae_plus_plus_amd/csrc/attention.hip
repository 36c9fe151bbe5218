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

}  // namespace

extern "C" int vitae_sdpa_fwd(const float* qkv, float* o, float* lse, int B, int N, int H, int head_dim,
                              void* stream) {
    if (!qkv || !o || !lse || B <= 0 || N <= 0 || H <= 0) return VITAE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
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
    switch (head_dim) {
        case 16: return launch_bwd<16, 1>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        case 32: return launch_bwd<32, 1>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        case 64: return launch_bwd<64, 2>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        case 128: return launch_bwd<128, 4>(qkv, o, d_o, lse, dqkv, delta, B, N, H, st);
        default: return VITAE_ERR_UNSUPPORTED_SHAPE;
    }
}
