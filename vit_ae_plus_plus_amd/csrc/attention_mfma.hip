// bf16-MFMA flash attention (forward, dQ, dK/dV) for head_dim 32 / 64 — the throughput-mode
// implementation of reference op K7 (model/vit.py:117-121).  fp32 tensors in HBM (same layouts as
// attention.hip), operands rounded to bf16 while being staged, fp32 accumulation and softmax.
//
// Everything is arranged so that NO cross-lane data movement is needed between the two matrix
// products of a tile (v_mfma_f32_32x32x16_bf16; C layout: lane holds column j = lane & 31 and the 16
// rows crow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi):
//   forward / dQ : S^T[key, q] = K Q^T  -> the lane's column is ONE query, so the softmax statistics
//                  (running max / sum, lse, delta) are per-lane scalars, and the 16 registers
//                  (bf16-packed) are directly the B operand of  O^T[d, q] += V^T[d, key] P^T[key, q];
//   dK/dV        : S[q, key] = Q K^T    -> the lane's column is ONE key; P and dS registers are the
//                  B operands of dV^T[d, key] += dO^T[d, q] P[q, key], dK^T[d, key] += Q^T[d, q] dS[q, key].
// The reduction-slot -> row mapping of those second products is whatever the C layout dictates
// (slot (s, hi, e) <-> row crow(8 s + e, hi)); the A operands (V^T, K^T, dO^T, Q^T) are gathered to
// match with ds_read_b64_tr_b16 (hardware transpose read: within a 16-lane group lane i receives
// element i of the four rows addressed by lanes 4 j .. 4 j + 3) from row-major bf16 LDS tiles.
// LDS rows are padded to HD + 8 elements: conflict-free for both the b128 fragment reads and the
// transpose reads.  Rows beyond N are zero-filled so stale LDS bits can never reach an MFMA.
#include <cstdlib>
#include "common.hpp"
#include "vitae_hip.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LN2 = 0.69314718055994530942f;
constexpr int CH = 128;   // rows (keys or queries) staged per LDS chunk

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ bf16x8 cvt8(const f32x4 a, const f32x4 b, float mul) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)(a[e] * mul); o[4 + e] = (__bf16)(b[e] * mul); }
    return o;
}

// 8 consecutive floats of one row -> bf16x8 (zeros when !valid)
__device__ __forceinline__ bf16x8 load_frag(const float* __restrict__ p, bool valid, float mul) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (valid) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
    return cvt8(a, b, mul);
}

// the same from a bf16 source (the qkv GEMM's bf16 copy): one 16-byte load, no conversion unless a scale is folded in
__device__ __forceinline__ bf16x8 load_frag(const __bf16* __restrict__ p, bool valid, float mul) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    if (valid) v = *reinterpret_cast<const bf16x8*>(p);
    if (mul != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * mul);
    }
    return v;
}

// rows [r0, r0 + CH) of a [*, HD] bf16 matrix (row stride ld) -> bf16 LDS tile [CH][HD + 8]: 16-byte copies
template <int HD>
__device__ __forceinline__ void stage_bf16(__bf16* dst, const __bf16* __restrict__ src, long ld, int r0, int N, float mul) {
    constexpr int LD = HD + 8, V = HD / 8;
    for (int idx = threadIdx.x; idx < CH * V; idx += 256) {
        const int row = idx / V, c8 = idx % V;
        *reinterpret_cast<bf16x8*>(dst + row * LD + c8 * 8) = load_frag(src + (long)(r0 + row) * ld + c8 * 8, r0 + row < N, mul);
    }
}

// rows [r0, r0 + CH) of a [*, HD] fp32 matrix (row stride ld) -> bf16 LDS tile [CH][HD + 8]
template <int HD>
__device__ __forceinline__ void stage_bf16(__bf16* dst, const float* __restrict__ src, long ld, int r0, int N, float mul) {
    constexpr int LD = HD + 8, V = HD / 4;
    for (int idx = threadIdx.x; idx < CH * V; idx += 256) {
        const int row = idx / V, c4 = idx % V;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < N) v = *reinterpret_cast<const f32x4*>(src + (long)(r0 + row) * ld + c4 * 4);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)(v[e] * mul);
        *reinterpret_cast<bf16x4*>(dst + row * LD + c4 * 4) = o;
    }
}

// The same two in two halves: issue() puts a chunk's global loads in flight (into registers), commit() converts and writes the
// LDS tile.  The streaming kernels issue chunk c + 1 right after the barrier that publishes chunk c and commit it behind the
// tiles of chunk c: the global latency of a chunk (1-2 us, every workgroup of a launch staging at the same moments) no longer
// sits between two barriers with all four waves waiting.
template <int HD, typename QT> struct ChunkStager;
template <int HD> struct ChunkStager<HD, float> {
    static constexpr int LD = HD + 8, V = HD / 4, NL = CH * V / 256;
    f32x4 r[NL];
    __device__ __forceinline__ void issue(const float* __restrict__ src, long ld, int r0, int N) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = threadIdx.x + 256 * i, row = idx / V, c4 = idx % V;
            r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r0 + row < N) r[i] = *reinterpret_cast<const f32x4*>(src + (long)(r0 + row) * ld + c4 * 4);
        }
    }
    __device__ __forceinline__ void commit(__bf16* dst, float mul) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = threadIdx.x + 256 * i, row = idx / V, c4 = idx % V;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(r[i][e] * mul);
            *reinterpret_cast<bf16x4*>(dst + row * LD + c4 * 4) = o;
        }
    }
};
template <int HD> struct ChunkStager<HD, __bf16> {
    static constexpr int LD = HD + 8, V = HD / 8, NL = CH * V / 256;
    bf16x8 r[NL];
    __device__ __forceinline__ void issue(const __bf16* __restrict__ src, long ld, int r0, int N) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = threadIdx.x + 256 * i, row = idx / V, c8 = idx % V;
            r[i] = load_frag(src + (long)(r0 + row) * ld + c8 * 8, r0 + row < N, 1.f);
        }
    }
    __device__ __forceinline__ void commit(__bf16* dst, float mul) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = threadIdx.x + 256 * i, row = idx / V, c8 = idx % V;
            bf16x8 v = r[i];
            if (mul != 1.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * mul);
            }
            *reinterpret_cast<bf16x8*>(dst + row * LD + c8 * 8) = v;
        }
    }
};

// A operand (32 x 16 slice of T^T) for reduction slots (s2, hi, e) <-> rows rowbase + crow(8 s2 + e, hi),
// output rows i <-> columns colbase + (lane & 31) of the row-major LDS tile T (row stride LD).
template <int LD>
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* T, int rowbase, int s2, int colbase, int lane) {
    const int gg = lane >> 4, li = lane & 15;
    const int row = rowbase + 16 * s2 + 4 * (gg >> 1) + (li >> 2);
    const int col = colbase + 16 * (gg & 1) + 4 * (li & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(T + row * LD + col));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(T + (row + 8) * LD + col));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.b;
}

__device__ __forceinline__ void pack16(const f32x16& p, bf16x8 (&f)[2]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { f[0][e] = (__bf16)p[e]; f[1][e] = (__bf16)p[8 + e]; }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// Column sums of a lane-per-row fragment set (the qkv bias gradient, fused into the kernels that produce dqkv):
// v[i] of the 32 lanes sharing `hi` are added by a butterfly over lane bits 0..4, the four waves' totals meet in
// LDS (`red`, 4 * 16 * NT * 2 floats), and one thread per column issues a single atomicAdd.
// Every wave of the block must call this (it contains a barrier).
template <int NV>
__device__ __forceinline__ void colsum_to(float (&v)[NV], bool valid, float* __restrict__ out, float* red, int hi, int l31,
                                          int wave) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x = valid ? v[i] : 0.f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        v[i] = x;
    }
    __syncthreads();
    if (l31 == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int nt = i / 16, r = i % 16;
            red[wave * (2 * NV) + 32 * nt + 8 * (r >> 2) + 4 * hi + (r & 3)] = v[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * NV) {
        const int c = threadIdx.x;
        atomicAdd(out + c, red[c] + red[2 * NV + c] + red[4 * NV + c] + red[6 * NV + c]);
    }
}

// ------------------------------------------------------------------------------- forward
template <int HD, typename QT = float>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const QT* __restrict__ qkv, float* __restrict__ o,
                                                            __bf16* __restrict__ o16,
                                                            float* __restrict__ lse, int N, int H, float scale) {
    constexpr int LD = HD + 8, NKK = HD / 16, NT = HD / 32;
    __shared__ __attribute__((aligned(16))) __bf16 Ks[CH * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[CH * LD];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int D = H * HD;
    const long ld = 3L * D;
    const QT* base = qkv + (long)b * N * ld + h * HD;
    const int q0 = (blockIdx.x * 4 + wave) * 32, qrow = q0 + l31;
    const bool qvalid = qrow < N, wave_live = q0 < N;
    bf16x8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = load_frag(base + (long)qrow * ld + 16 * kk + 8 * hi, qvalid, scale * LOG2E);
    f32x16 oacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) oacc[nt] = zero16();
    float m = -1e30f, lsum = 0.f;
    ChunkStager<HD, QT> sk, sv;
    sk.issue(base + D, ld, 0, N);
    sv.issue(base + 2 * D, ld, 0, N);
    for (int c0 = 0; c0 < N; c0 += CH) {
        __syncthreads();
        sk.commit(Ks, 1.f);
        sv.commit(Vs, 1.f);
        __syncthreads();
        if (c0 + CH < N) {                                  // the next chunk's loads fly under this chunk's tiles
            sk.issue(base + D, ld, c0 + CH, N);
            sv.issue(base + 2 * D, ld, c0 + CH, N);
        }
        if (!wave_live) continue;
        const int kend = min(CH, N - c0);
        for (int kt = 0; kt * 32 < kend; ++kt) {
            f32x16 s = zero16();
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ks[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[kk], s, 0, 0, 0);
            }
            if (c0 + kt * 32 + 32 > N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (c0 + kt * 32 + crow(r, hi) >= N) s[r] = -1e30f;
            }
            float tmax = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mn = fmaxf(m, tmax);
            const float corr = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mn); psum += s[r]; }
            lsum = lsum * corr + psum;
            bf16x8 pf[2];
            pack16(s, pf);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[nt][r] *= corr;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 a = tr_frag<LD>(Vs, kt * 32, s2, 32 * nt, lane);
                    oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[s2], oacc[nt], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_live) return;
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    if (qvalid) {
        const float inv = 1.f / ltot;
        float* orow = o + ((long)b * N + qrow) * D + h * HD;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {oacc[nt][4 * g] * inv, oacc[nt][4 * g + 1] * inv, oacc[nt][4 * g + 2] * inv, oacc[nt][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(orow + 32 * nt + 8 * g + 4 * hi) = v;
                if (o16) {
                    bf16x4 v16;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v16[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>(o16 + ((long)b * N + qrow) * D + h * HD + 32 * nt + 8 * g + 4 * hi) = v16;
                }
            }
        if (hi == 0) lse[((long)b * H + h) * N + qrow] = (m + __builtin_amdgcn_logf(ltot)) * LN2;
    }
}

// ------------------------------------------------------------------------------- backward: dQ (+ delta)
template <int HD, typename QT = float>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const QT* __restrict__ qkv, const float* __restrict__ o,
                                                               const float* __restrict__ d_o, const float* __restrict__ lse,
                                                               float* __restrict__ dqkv, __bf16* __restrict__ dqkv16,
                                                               float* __restrict__ dbias, float* __restrict__ delta, int N,
                                                               int H, float scale) {
    constexpr int LD = HD + 8, NKK = HD / 16, NT = HD / 32;
    __shared__ __attribute__((aligned(16))) __bf16 Ks[CH * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[CH * LD];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int D = H * HD;
    const long ld = 3L * D;
    const QT* base = qkv + (long)b * N * ld + h * HD;
    const int q0 = (blockIdx.x * 4 + wave) * 32, qrow = q0 + l31;
    const bool qvalid = qrow < N, wave_live = q0 < N;
    bf16x8 qf[NKK], gf[NKK];
    float dl = 0.f;
    const float* grow = d_o + ((long)b * N + qrow) * D + h * HD;
    const float* orow = o + ((long)b * N + qrow) * D + h * HD;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int c = 16 * kk + 8 * hi;
        qf[kk] = load_frag(base + (long)qrow * ld + c, qvalid, scale * LOG2E);
        f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, o0 = g0, o1 = g0;
        if (qvalid) {
            g0 = *reinterpret_cast<const f32x4*>(grow + c); g1 = *reinterpret_cast<const f32x4*>(grow + c + 4);
            o0 = *reinterpret_cast<const f32x4*>(orow + c); o1 = *reinterpret_cast<const f32x4*>(orow + c + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dl += g0[e] * o0[e] + g1[e] * o1[e];
        gf[kk] = cvt8(g0, g1, 1.f);
    }
    dl += __shfl_xor(dl, 32, 64);
    const float Lq = qvalid ? lse[((long)b * H + h) * N + qrow] * LOG2E : 0.f;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
    ChunkStager<HD, QT> sk, sv;
    sk.issue(base + D, ld, 0, N);
    sv.issue(base + 2 * D, ld, 0, N);
    for (int c0 = 0; c0 < N; c0 += CH) {
        __syncthreads();
        sk.commit(Ks, 1.f);
        sv.commit(Vs, 1.f);
        __syncthreads();
        if (c0 + CH < N) {
            sk.issue(base + D, ld, c0 + CH, N);
            sv.issue(base + 2 * D, ld, c0 + CH, N);
        }
        if (!wave_live) continue;
        const int kend = min(CH, N - c0);
        for (int kt = 0; kt * 32 < kend; ++kt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 ak = *reinterpret_cast<const bf16x8*>(&Ks[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(&Vs[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, gf[kk], dp, 0, 0, 0);
            }
            const bool tail = c0 + kt * 32 + 32 > N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(s[r] - Lq);
                if (tail && c0 + kt * 32 + crow(r, hi) >= N) p = 0.f;
                s[r] = p * (dp[r] - dl);
            }
            bf16x8 dsf[2];
            pack16(s, dsf);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 a = tr_frag<LD>(Ks, kt * 32, s2, 32 * nt, lane);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsf[s2], acc[nt], 0, 0, 0);
                }
        }
    }
    if (dbias) {
        float cs[16 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[16 * nt + r] = acc[nt][r] * scale;
        colsum_to<16 * NT>(cs, wave_live && qvalid, dbias + h * HD, reinterpret_cast<float*>(Ks), hi, l31, wave);
    }
    if (!wave_live || !qvalid) return;
    const long ooff = ((long)b * N + qrow) * ld + h * HD;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[nt][4 * g] * scale, acc[nt][4 * g + 1] * scale, acc[nt][4 * g + 2] * scale, acc[nt][4 * g + 3] * scale};
            if (dqkv) *reinterpret_cast<f32x4*>(dqkv + ooff + 32 * nt + 8 * g + 4 * hi) = v;
            if (dqkv16) {
                bf16x4 v16;
#pragma unroll
                for (int e = 0; e < 4; ++e) v16[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x4*>(dqkv16 + ((long)b * N + qrow) * ld + h * HD + 32 * nt + 8 * g + 4 * hi) = v16;
            }
        }
    if (hi == 0) delta[((long)b * H + h) * N + qrow] = dl;
}

// ------------------------------------------------------------------------------- backward: dK, dV
template <int HD, typename QT = float>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const QT* __restrict__ qkv, const float* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dqkv, __bf16* __restrict__ dqkv16,
                                                                float* __restrict__ dbias, int N, int H, float scale) {
    constexpr int LD = HD + 8, NKK = HD / 16, NT = HD / 32;
    __shared__ __attribute__((aligned(16))) __bf16 Qs[CH * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Gs[CH * LD];
    __shared__ __attribute__((aligned(16))) float Ls[CH];
    __shared__ __attribute__((aligned(16))) float Ds[CH];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int D = H * HD;
    const long ld = 3L * D;
    const QT* base = qkv + (long)b * N * ld + h * HD;
    const int k0 = (blockIdx.x * 4 + wave) * 32, krow = k0 + l31;
    const bool kvalid = krow < N, wave_live = k0 < N;
    bf16x8 kf[NKK], vf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        kf[kk] = load_frag(base + (long)krow * ld + D + 16 * kk + 8 * hi, kvalid, 1.f);
        vf[kk] = load_frag(base + (long)krow * ld + 2 * D + 16 * kk + 8 * hi, kvalid, 1.f);
    }
    f32x16 dk[NT], dv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dk[nt] = zero16(); dv[nt] = zero16(); }
    const float* gbase = d_o + (long)b * N * D + h * HD;
    const float* lrow = lse + ((long)b * H + h) * N;
    const float* drow = delta + ((long)b * H + h) * N;
    ChunkStager<HD, QT> sq;
    ChunkStager<HD, float> sg;
    float nl = 1e30f, nd = 0.f;                            // lse / delta of this thread's query of the NEXT chunk (threads < CH)
    auto issue_chunk = [&](int c) {
        sq.issue(base, ld, c, N);
        sg.issue(gbase, D, c, N);
        if (threadIdx.x < CH) {
            const int q = c + threadIdx.x;
            nl = q < N ? lrow[q] * LOG2E : 1e30f;           // invalid query -> P = exp2(-huge) = 0
            nd = q < N ? drow[q] : 0.f;
        }
    };
    issue_chunk(0);
    for (int c0 = 0; c0 < N; c0 += CH) {
        __syncthreads();
        sq.commit(Qs, scale * LOG2E);
        sg.commit(Gs, 1.f);
        if (threadIdx.x < CH) { Ls[threadIdx.x] = nl; Ds[threadIdx.x] = nd; }
        __syncthreads();
        if (c0 + CH < N) issue_chunk(c0 + CH);
        if (!wave_live) continue;
        const int qend = min(CH, N - c0);
        for (int qt = 0; qt * 32 < qend; ++qt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 aq = *reinterpret_cast<const bf16x8*>(&Qs[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                const bf16x8 ag = *reinterpret_cast<const bf16x8*>(&Gs[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, vf[kk], dp, 0, 0, 0);
            }
            f32x16 p;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 L4 = *reinterpret_cast<const f32x4*>(&Ls[qt * 32 + 8 * g + 4 * hi]);
                const f32x4 D4 = *reinterpret_cast<const f32x4*>(&Ds[qt * 32 + 8 * g + 4 * hi]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(s[4 * g + e] - L4[e]);
                    p[4 * g + e] = pe;
                    s[4 * g + e] = pe * (dp[4 * g + e] - D4[e]);
                }
            }
            bf16x8 pf[2], dsf[2];
            pack16(p, pf);
            pack16(s, dsf);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 ag = tr_frag<LD>(Gs, qt * 32, s2, 32 * nt, lane);
                    dv[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, pf[s2], dv[nt], 0, 0, 0);
                    const bf16x8 aq = tr_frag<LD>(Qs, qt * 32, s2, 32 * nt, lane);
                    dk[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, dsf[s2], dk[nt], 0, 0, 0);
                }
        }
    }
    if (dbias) {
        float ck[16 * NT], cv[16 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { ck[16 * nt + r] = dk[nt][r] * LN2; cv[16 * nt + r] = dv[nt][r]; }
        colsum_to<16 * NT>(ck, wave_live && kvalid, dbias + D + h * HD, reinterpret_cast<float*>(Qs), hi, l31, wave);
        colsum_to<16 * NT>(cv, wave_live && kvalid, dbias + 2 * D + h * HD, reinterpret_cast<float*>(Gs), hi, l31, wave);
    }
    if (!wave_live || !kvalid) return;
    const long ooff = ((long)b * N + krow) * ld + h * HD;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * nt + 8 * g + 4 * hi;
            f32x4 vk = {dk[nt][4 * g] * LN2, dk[nt][4 * g + 1] * LN2, dk[nt][4 * g + 2] * LN2, dk[nt][4 * g + 3] * LN2};
            f32x4 vv = {dv[nt][4 * g], dv[nt][4 * g + 1], dv[nt][4 * g + 2], dv[nt][4 * g + 3]};
            if (dqkv) {
                *reinterpret_cast<f32x4*>(dqkv + ooff + D + d) = vk;       // Qs carried scale*log2e: dK = sum dS * scale * Q
                *reinterpret_cast<f32x4*>(dqkv + ooff + 2 * D + d) = vv;
            }
            if (dqkv16) {
                bf16x4 k16, v16;
#pragma unroll
                for (int e = 0; e < 4; ++e) { k16[e] = (__bf16)vk[e]; v16[e] = (__bf16)vv[e]; }
                __bf16* o16 = dqkv16 + ((long)b * N + krow) * ld + h * HD;
                *reinterpret_cast<bf16x4*>(o16 + D + d) = k16;
                *reinterpret_cast<bf16x4*>(o16 + 2 * D + d) = v16;
            }
        }
}

// ------------------------------------------------------------------------------- backward, one launch
// For sequences whose whole head fits in LDS (N <= ~512 at hd 32, ~256 at hd 64 — every model of the reference):
// Q (pre-scaled), K, V, dO of one (batch, head) are staged ONCE as bf16 tiles, delta = rowsum(O * dO) and the
// log-sum-exp go to LDS, and the waves of the block then split the 32-row work items: dQ tiles (as the dq kernel)
// and dK/dV tiles (as the dkv kernel).  One launch instead of two, no delta round trip through HBM, and all four
// waves have work at N = 55 (two dQ + two dK/dV items).  qkv-bias column sums are collected with LDS atomics and
// leave the block as one global atomic per column.
template <int HD, typename QT = float>
__global__ __launch_bounds__(256) void attn_bwd_fused_kernel(const QT* __restrict__ qkv, const float* __restrict__ o,
                                                             const float* __restrict__ d_o, const float* __restrict__ lse,
                                                             float* __restrict__ dqkv, __bf16* __restrict__ dqkv16,
                                                             float* __restrict__ dbias, int N, int H, float scale, int NP) {
    constexpr int LD = HD + 8, NKK = HD / 16, NT = HD / 32, V4 = HD / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    __bf16* Qs = reinterpret_cast<__bf16*>(fsm);
    __bf16* Ks = Qs + NP * LD;
    __bf16* Vs = Ks + NP * LD;
    __bf16* Gs = Vs + NP * LD;
    float* Ls = reinterpret_cast<float*>(Gs + NP * LD);
    float* Ds = Ls + NP;
    float* Cs = Ds + NP;                          // [3 * HD] column sums of dq | dk | dv
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int D = H * HD;
    const long ld = 3L * D;
    const QT* base = qkv + (long)b * N * ld + h * HD;
    const float* gbase = d_o + (long)b * N * D + h * HD;
    const float* obase = o + (long)b * N * D + h * HD;
    // ---- stage the four operand tiles (rows >= N zero) and the per-row scalars
    for (int idx = threadIdx.x; idx < NP * V4; idx += 256) {
        const int row = idx / V4, c4 = (idx % V4) * 4;
        f32x4 g = {0.f, 0.f, 0.f, 0.f}, ov = g;
        if (row < N) ov = *reinterpret_cast<const f32x4*>(obase + (long)row * D + c4);
        bf16x4 q16, k16, v16, g16;
        if constexpr (sizeof(QT) == 2) {
            // bf16 qkv (the GEMM's own bf16 copy): k and v pass through, q takes the softmax scale
#pragma unroll
            for (int e = 0; e < 4; ++e) { q16[e] = (__bf16)0.f; k16[e] = (__bf16)0.f; v16[e] = (__bf16)0.f; }
            if (row < N) {
                const QT* r = base + (long)row * ld + c4;
                q16 = *reinterpret_cast<const bf16x4*>(r);
                k16 = *reinterpret_cast<const bf16x4*>(r + D);
                v16 = *reinterpret_cast<const bf16x4*>(r + 2 * D);
                g = *reinterpret_cast<const f32x4*>(gbase + (long)row * D + c4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { q16[e] = (__bf16)((float)q16[e] * (scale * LOG2E)); g16[e] = (__bf16)g[e]; }
        } else {
            f32x4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q;
            if (row < N) {
                const QT* r = base + (long)row * ld + c4;
                q = *reinterpret_cast<const f32x4*>(r);
                k = *reinterpret_cast<const f32x4*>(r + D);
                v = *reinterpret_cast<const f32x4*>(r + 2 * D);
                g = *reinterpret_cast<const f32x4*>(gbase + (long)row * D + c4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q16[e] = (__bf16)(q[e] * (scale * LOG2E)); k16[e] = (__bf16)k[e]; v16[e] = (__bf16)v[e]; g16[e] = (__bf16)g[e];
            }
        }
        *reinterpret_cast<bf16x4*>(Qs + row * LD + c4) = q16;
        *reinterpret_cast<bf16x4*>(Ks + row * LD + c4) = k16;
        *reinterpret_cast<bf16x4*>(Vs + row * LD + c4) = v16;
        *reinterpret_cast<bf16x4*>(Gs + row * LD + c4) = g16;
        // delta = rowsum(O * dO) on the way: a row's V4 pieces sit in V4 consecutive lanes (NP * V4 is a multiple of 256: every
        // lane of every wave takes part), so O is read as coalesced as dO — a per-row loop with one thread per row read 2 x HD
        // floats per lane at a row stride and cost the launch a microsecond or two
        float dl = g[0] * ov[0] + g[1] * ov[1] + g[2] * ov[2] + g[3] * ov[3];
#pragma unroll
        for (int of = 1; of < V4; of <<= 1) dl += __shfl_xor(dl, of, 64);
        if ((idx % V4) == 0) Ds[row] = dl;         // rows >= N: 0
    }
    for (int row = threadIdx.x; row < NP; row += 256)
        Ls[row] = row < N ? lse[((long)b * H + h) * N + row] * LOG2E : 1e30f;      // invalid query: P = exp2(s - huge) = 0
    if (threadIdx.x < 3 * HD) Cs[threadIdx.x] = 0.f;
    __syncthreads();

    const int T = NP / 32;
    for (int item = blockIdx.x * 4 + wave; item < 2 * T; item += gridDim.x * 4) {
        if (item < T) {
            // ---------------- dQ tile `item`
            const int qrow = item * 32 + l31;
            const bool qvalid = qrow < N;
            bf16x8 qf[NKK], gf[NKK];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                qf[kk] = *reinterpret_cast<const bf16x8*>(&Qs[qrow * LD + 16 * kk + 8 * hi]);
                gf[kk] = *reinterpret_cast<const bf16x8*>(&Gs[qrow * LD + 16 * kk + 8 * hi]);
            }
            const float Lq = Ls[qrow], dl = Ds[qrow];
            f32x16 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
            for (int kt = 0; kt < T; ++kt) {
                f32x16 sc = zero16(), dp = zero16();
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 ak = *reinterpret_cast<const bf16x8*>(&Ks[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(&Vs[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[kk], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, gf[kk], dp, 0, 0, 0);
                }
                const bool tail = kt * 32 + 32 > N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(sc[r] - Lq);
                    if (tail && kt * 32 + crow(r, hi) >= N) p = 0.f;
                    sc[r] = p * (dp[r] - dl);
                }
                bf16x8 dsf[2];
                pack16(sc, dsf);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 a = tr_frag<LD>(Ks, kt * 32, s2, 32 * nt, lane);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsf[s2], acc[nt], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] *= scale;
            if (dbias) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float x = qvalid ? acc[nt][r] : 0.f;
#pragma unroll
                        for (int of = 16; of > 0; of >>= 1) x += __shfl_xor(x, of, 64);
                        if (l31 == 0) atomicAdd(&Cs[32 * nt + 8 * (r >> 2) + 4 * hi + (r & 3)], x);
                    }
            }
            if (qvalid) {
                float* out = dqkv + ((long)b * N + qrow) * ld + h * HD;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                        if (dqkv) *reinterpret_cast<f32x4*>(out + 32 * nt + 8 * g + 4 * hi) = v;
                        if (dqkv16) {
                            bf16x4 v16;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v16[e] = (__bf16)v[e];
                            *reinterpret_cast<bf16x4*>(dqkv16 + ((long)b * N + qrow) * ld + h * HD + 32 * nt + 8 * g + 4 * hi) = v16;
                        }
                    }
            }
        } else {
            // ---------------- dK / dV tile `item - T`
            const int kt0 = item - T;
            const int krow = kt0 * 32 + l31;
            const bool kvalid = krow < N;
            bf16x8 kf[NKK], vf[NKK];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                kf[kk] = *reinterpret_cast<const bf16x8*>(&Ks[krow * LD + 16 * kk + 8 * hi]);
                vf[kk] = *reinterpret_cast<const bf16x8*>(&Vs[krow * LD + 16 * kk + 8 * hi]);
            }
            f32x16 dk[NT], dv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { dk[nt] = zero16(); dv[nt] = zero16(); }
            for (int qt = 0; qt < T; ++qt) {
                f32x16 sc = zero16(), dp = zero16();
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 aq = *reinterpret_cast<const bf16x8*>(&Qs[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    const bf16x8 ag = *reinterpret_cast<const bf16x8*>(&Gs[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[kk], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, vf[kk], dp, 0, 0, 0);
                }
                f32x16 p;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 L4 = *reinterpret_cast<const f32x4*>(&Ls[qt * 32 + 8 * g + 4 * hi]);
                    const f32x4 D4 = *reinterpret_cast<const f32x4*>(&Ds[qt * 32 + 8 * g + 4 * hi]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(sc[4 * g + e] - L4[e]);
                        p[4 * g + e] = pe;
                        sc[4 * g + e] = pe * (dp[4 * g + e] - D4[e]);
                    }
                }
                bf16x8 pf[2], dsf[2];
                pack16(p, pf);
                pack16(sc, dsf);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 ag = tr_frag<LD>(Gs, qt * 32, s2, 32 * nt, lane);
                        dv[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, pf[s2], dv[nt], 0, 0, 0);
                        const bf16x8 aq = tr_frag<LD>(Qs, qt * 32, s2, 32 * nt, lane);
                        dk[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, dsf[s2], dk[nt], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dk[nt][r] *= LN2;     // Qs carried scale*log2e: dK = sum dS * scale * Q
            if (dbias) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float x = kvalid ? dk[nt][r] : 0.f, y = kvalid ? dv[nt][r] : 0.f;
#pragma unroll
                        for (int of = 16; of > 0; of >>= 1) { x += __shfl_xor(x, of, 64); y += __shfl_xor(y, of, 64); }
                        if (l31 == 0) {
                            const int c = 32 * nt + 8 * (r >> 2) + 4 * hi + (r & 3);
                            atomicAdd(&Cs[HD + c], x);
                            atomicAdd(&Cs[2 * HD + c], y);
                        }
                    }
            }
            if (kvalid) {
                float* out = dqkv + ((long)b * N + krow) * ld + h * HD;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d = 32 * nt + 8 * g + 4 * hi;
                        f32x4 vk = {dk[nt][4 * g], dk[nt][4 * g + 1], dk[nt][4 * g + 2], dk[nt][4 * g + 3]};
                        f32x4 vv = {dv[nt][4 * g], dv[nt][4 * g + 1], dv[nt][4 * g + 2], dv[nt][4 * g + 3]};
                        if (dqkv) {
                            *reinterpret_cast<f32x4*>(out + D + d) = vk;
                            *reinterpret_cast<f32x4*>(out + 2 * D + d) = vv;
                        }
                        if (dqkv16) {
                            bf16x4 k16, v16;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { k16[e] = (__bf16)vk[e]; v16[e] = (__bf16)vv[e]; }
                            __bf16* o16 = dqkv16 + ((long)b * N + krow) * ld + h * HD;
                            *reinterpret_cast<bf16x4*>(o16 + D + d) = k16;
                            *reinterpret_cast<bf16x4*>(o16 + 2 * D + d) = v16;
                        }
                    }
            }
        }
    }
    if (dbias) {
        __syncthreads();
        if (threadIdx.x < 3 * HD) {
            const int part = threadIdx.x / HD, c = threadIdx.x % HD;
            atomicAdd(dbias + part * D + h * HD + c, Cs[threadIdx.x]);
        }
    }
}

}  // namespace

// Returns VITAE_ERR_UNSUPPORTED_SHAPE for head dims without an MFMA instantiation (caller falls back
// to the fp32 VALU kernels of attention.hip — same results to bf16 round-off).
template <typename QT>
static int sdpa_mfma_fwd_launch(const QT* qkv, float* o, void* o_bf16, float* lse, int B, int N, int H, int head_dim, void* stream) {
    if (!qkv || !o || !lse || B <= 0 || N <= 0 || H <= 0) return VITAE_ERR_INVALID_ARG;
    if ((((long)H * head_dim) & 7) || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const float scale = 1.0f / sqrtf((float)head_dim);
    dim3 grid(cdiv(N, 128), H, B);
    hipStream_t st = (hipStream_t)stream;
    __bf16* o16 = reinterpret_cast<__bf16*>(o_bf16);
    if (head_dim == 32) hipLaunchKernelGGL((attn_fwd_mfma_kernel<32, QT>), grid, dim3(256), 0, st, qkv, o, o16, lse, N, H, scale);
    else if (head_dim == 64) hipLaunchKernelGGL((attn_fwd_mfma_kernel<64, QT>), grid, dim3(256), 0, st, qkv, o, o16, lse, N, H, scale);
    else return VITAE_ERR_UNSUPPORTED_SHAPE;
    return vitae_launch_status();
}

extern "C" int vitae_sdpa_mfma_fwd(const float* qkv, float* o, void* o_bf16, float* lse, int B, int N, int H, int head_dim,
                                   void* stream) {
    return sdpa_mfma_fwd_launch<float>(qkv, o, o_bf16, lse, B, N, H, head_dim, stream);
}

// the same with q | k | v read from the bf16 copy the qkv GEMM writes (no fp32 qkv in HBM at all)
extern "C" int vitae_sdpa_mfma_fwd_bf16in(const void* qkv_bf16, float* o, void* o_bf16, float* lse, int B, int N, int H,
                                          int head_dim, void* stream) {
    return sdpa_mfma_fwd_launch<__bf16>(reinterpret_cast<const __bf16*>(qkv_bf16), o, o_bf16, lse, B, N, H, head_dim, stream);
}

// one-launch backward from bf16 q | k | v; VITAE_ERR_UNSUPPORTED_SHAPE when the head does not fit LDS (the caller then keeps an
// fp32 qkv and uses vitae_sdpa_mfma_bwd).  vitae_sdpa_bwd_fused_fits() answers that up front.
extern "C" int vitae_sdpa_bwd_fused_fits(int N, int head_dim) {
    static const int fused_on = getenv("VITAE_ATTN_BWD_FUSED") ? atoi(getenv("VITAE_ATTN_BWD_FUSED")) : 1;
    const int NP = cdiv(N, 32) * 32;
    const size_t lds = (size_t)4 * NP * (head_dim + 8) * 2 + (size_t)2 * NP * 4 + (size_t)3 * head_dim * 4;
    return fused_on && (head_dim == 32 || head_dim == 64) && lds <= 150 * 1024;
}

// dq + dkv kernels (any N): the head streams through LDS in 128-row chunks; delta_ws [B * H * N] floats
template <typename QT>
static int sdpa_bwd_two_kernels(const QT* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, __bf16* g16,
                                float* dqkv_colsum_accum, float* delta_ws, int B, int N, int H, int head_dim, float scale,
                                hipStream_t st) {
    dim3 grid(cdiv(N, 128), H, B);
    if (head_dim == 32) {
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<32, QT>), grid, dim3(256), 0, st, qkv, o, d_o, lse, dqkv, g16, dqkv_colsum_accum, delta_ws, N, H, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<32, QT>), grid, dim3(256), 0, st, qkv, d_o, lse, delta_ws, dqkv, g16, dqkv_colsum_accum, N, H, scale);
    } else if (head_dim == 64) {
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<64, QT>), grid, dim3(256), 0, st, qkv, o, d_o, lse, dqkv, g16, dqkv_colsum_accum, delta_ws, N, H, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<64, QT>), grid, dim3(256), 0, st, qkv, d_o, lse, delta_ws, dqkv, g16, dqkv_colsum_accum, N, H, scale);
    } else {
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    }
    return vitae_launch_status();
}

extern "C" int vitae_sdpa_mfma_bwd_bf16in(const void* qkv_bf16, const float* o, const float* d_o, const float* lse, float* dqkv,
                                          void* dqkv_bf16, float* dqkv_colsum_accum, float* delta_ws, int B, int N, int H,
                                          int head_dim, void* stream) {
    if (!qkv_bf16 || !o || !d_o || !lse || (!dqkv && !dqkv_bf16) || B <= 0 || N <= 0 || H <= 0) return VITAE_ERR_INVALID_ARG;
    if ((((long)H * head_dim) & 7) || ((uintptr_t)qkv_bf16 & 15) || ((uintptr_t)o & 15) || ((uintptr_t)d_o & 15) ||
        ((uintptr_t)dqkv & 15) || (head_dim != 32 && head_dim != 64))
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    const float scale = 1.0f / sqrtf((float)head_dim);
    if (!vitae_sdpa_bwd_fused_fits(N, head_dim)) {         // the head does not fit LDS: the two streaming kernels
        if (!delta_ws) return VITAE_ERR_INVALID_ARG;
        return sdpa_bwd_two_kernels<__bf16>(reinterpret_cast<const __bf16*>(qkv_bf16), o, d_o, lse, dqkv, reinterpret_cast<__bf16*>(dqkv_bf16),
                                            dqkv_colsum_accum, delta_ws, B, N, H, head_dim, scale, (hipStream_t)stream);
    }
    const int NP = cdiv(N, 32) * 32;
    const size_t lds = (size_t)4 * NP * (head_dim + 8) * 2 + (size_t)2 * NP * 4 + (size_t)3 * head_dim * 4;
    const int items = 2 * (NP / 32);
    int G = cdiv(items, 4);
    if ((long)G * H * B > 1024) G = cdiv(items, 8);
    dim3 fgrid(G, H, B);
    hipStream_t st = (hipStream_t)stream;
    const __bf16* q16 = reinterpret_cast<const __bf16*>(qkv_bf16);
    __bf16* g16 = reinterpret_cast<__bf16*>(dqkv_bf16);
    if (head_dim == 32) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<32, __bf16>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)attr;
        hipLaunchKernelGGL((attn_bwd_fused_kernel<32, __bf16>), fgrid, dim3(256), lds, st, q16, o, d_o, lse, dqkv, g16,
                           dqkv_colsum_accum, N, H, scale, NP);
    } else {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<64, __bf16>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)attr;
        hipLaunchKernelGGL((attn_bwd_fused_kernel<64, __bf16>), fgrid, dim3(256), lds, st, q16, o, d_o, lse, dqkv, g16,
                           dqkv_colsum_accum, N, H, scale, NP);
    }
    return vitae_launch_status();
}

extern "C" int vitae_sdpa_mfma_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                                   void* dqkv_bf16, float* dqkv_colsum_accum, float* delta_ws, int B, int N, int H,
                                   int head_dim, void* stream) {
    // dqkv (fp32) may be NULL when the bf16 copy is all the caller consumes (the one-launch backward only)
    if (!qkv || !o || !d_o || !lse || (!dqkv && !dqkv_bf16) || !delta_ws || B <= 0 || N <= 0 || H <= 0) return VITAE_ERR_INVALID_ARG;
    if ((((long)H * head_dim) & 3) || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15) || ((uintptr_t)d_o & 15) ||
        ((uintptr_t)dqkv & 15))
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    const float scale = 1.0f / sqrtf((float)head_dim);
    dim3 grid(cdiv(N, 128), H, B);
    hipStream_t st = (hipStream_t)stream;
    __bf16* g16 = reinterpret_cast<__bf16*>(dqkv_bf16);
    // whole head resident in LDS -> the one-launch backward
    static const int fused_on = getenv("VITAE_ATTN_BWD_FUSED") ? atoi(getenv("VITAE_ATTN_BWD_FUSED")) : 1;
    const int NP = cdiv(N, 32) * 32;
    const size_t lds = (size_t)4 * NP * (head_dim + 8) * 2 + (size_t)2 * NP * 4 + (size_t)3 * head_dim * 4;
    if (fused_on && (head_dim == 32 || head_dim == 64) && lds <= 150 * 1024) {
        const int items = 2 * (NP / 32);
        int G = cdiv(items, 4);                                 // one item per wave ...
        if ((long)G * H * B > 1024) G = cdiv(items, 8);          // ... two when that many workgroups would queue up
        dim3 fgrid(G, H, B);
        if (head_dim == 32) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<32>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)attr;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<32>), fgrid, dim3(256), lds, st, qkv, o, d_o, lse, dqkv, g16,
                               dqkv_colsum_accum, N, H, scale, NP);
        } else {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<64>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)attr;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<64>), fgrid, dim3(256), lds, st, qkv, o, d_o, lse, dqkv, g16,
                               dqkv_colsum_accum, N, H, scale, NP);
        }
        return vitae_launch_status();
    }
    return sdpa_bwd_two_kernels<float>(qkv, o, d_o, lse, dqkv, g16, dqkv_colsum_accum, delta_ws, B, N, H, head_dim, scale, st);
}
