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
template <int HD, typename QT, int CHR = CH> struct ChunkStager;
template <int HD, int CHR> struct ChunkStager<HD, float, CHR> {
    static constexpr int LD = HD + 8, V = HD / 4, NL = CHR * V / 256;
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
template <int HD, int CHR> struct ChunkStager<HD, __bf16, CHR> {
    static constexpr int LD = HD + 8, V = HD / 8, NL = CHR * V / 256;
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

// ------------------------------------------------------------------------------- streaming kernels (any N), round 4
// What changed against the round-3 kernels (68 us forward / 97 + 85 us backward at N = 1729, hd = 32, ~90 VALU instructions per
// 32 x 32 tile against 4-8 MFMAs, two workgroup barriers per four tiles).  Counters (profiles/round4_attention.txt) say these
// kernels are bound by instruction ISSUE of their SIMD — a wave-64 VALU instruction holds it ~4 clocks, an MFMA ~21-32, and the
// sum over the resident waves reaches ~80 % of the SIMD's time whatever the occupancy — so the work is counted in instructions:
//   * the softmax reference point rides in the MFMA's C operand: S' = K Q^T + (-m) costs nothing (the C input of
//     v_mfma is a register block of its own — 16 VGPRs holding -m broadcast — so no per-tile initialisation either), and the
//     backward's S - lse and dP - delta come out of the matrix pipe the same way: exp2(S') with NO subtraction per element;
//   * the running maximum is only a reference point: it moves when a tile's maximum exceeds it by more than 2^SM_THR (deferred
//     rescale, cdna_hip_programming.md T13; the first tile always sets it) — no per-tile exp2(m - m'), no per-tile O *= corr;
//   * a wave owns RB blocks of 32 rows (queries in forward / dQ, keys in dK/dV): every LDS fragment read serves RB MFMAs;
//   * K/V (Q/dO) chunks are double-buffered in LDS: ONE workgroup barrier per 128-row chunk, the chunk after next already in
//     flight in registers;
//   * the half-wave exchange of the tile maximum is a v_permlane32_swap (VALU), not a ds_bpermute.
constexpr float SM_THR = 8.f;

__device__ __forceinline__ f32x16 bcast16(float x) {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = x;
    return v;
}

// max over the two lanes that hold the same query (lane, lane ^ 32)
__device__ __forceinline__ float half_pair_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

__device__ __forceinline__ float max16(const f32x16& s) {
    float a = fmaxf(fmaxf(s[0], s[1]), s[2]), b = fmaxf(fmaxf(s[3], s[4]), s[5]);
    float c = fmaxf(fmaxf(s[6], s[7]), s[8]), d = fmaxf(fmaxf(s[9], s[10]), s[11]);
    a = fmaxf(fmaxf(a, s[12]), s[13]); b = fmaxf(fmaxf(b, s[14]), s[15]);
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// ------------------------------------------------------------------------------- forward
// One STEP of a wave = QB query blocks x KT key tiles (32 x 32 each): all QB * KT score products are issued back to back as
// independent accumulate chains (a dependent v_mfma pair costs its 64-clock result latency, an independent one 32 clocks of issue),
// one maximum / rescale decision and one pass of exp2 cover the whole step, and the P V products alternate between accumulators.
template <int HD, int QB, int KT, int CHK, typename QT = float>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const QT* __restrict__ qkv, float* __restrict__ o,
                                                            __bf16* __restrict__ o16,
                                                            float* __restrict__ lse, int N, int H, float scale) {
    constexpr int LD = HD + 8, NKK = HD / 16, NT = HD / 32;
    constexpr int NACC = (QB * NT >= 2) ? 1 : 2;            // independent P V accumulate chains per step: at least two
    static_assert(CHK % (32 * KT) == 0, "a chunk is a whole number of steps");
    __shared__ __attribute__((aligned(16))) __bf16 Ks[2][CHK * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[2][CHK * LD];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int D = H * HD;
    const long ld = 3L * D;
    const QT* base = qkv + (long)b * N * ld + h * HD;
    const int q0 = (blockIdx.x * 4 + wave) * (32 * QB);
    const bool wave_live = q0 < N;
    bf16x8 qf[QB][NKK];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int qrow = q0 + 32 * qb + l31;
            qf[qb][kk] = load_frag(base + (long)qrow * ld + 16 * kk + 8 * hi, qrow < N, scale * LOG2E);
        }
    f32x16 oacc[QB][NT][NACC], negm[QB];
    float m[QB], lsum[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        negm[qb] = zero16(); m[qb] = 0.f; lsum[qb] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int a = 0; a < NACC; ++a) oacc[qb][nt][a] = zero16();
    }
    ChunkStager<HD, QT, CHK> sk, sv;
    const int nch = (N + CHK - 1) / CHK;
    sk.issue(base + D, ld, 0, N);
    sv.issue(base + 2 * D, ld, 0, N);
    sk.commit(Ks[0], 1.f);
    sv.commit(Vs[0], 1.f);
    if (nch > 1) { sk.issue(base + D, ld, CHK, N); sv.issue(base + 2 * D, ld, CHK, N); }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int cur = c & 1, c0 = c * CHK;
        if (c + 1 < nch) {                                   // chunk c + 1 -> the buffer chunk c - 1 was read from (free since the last barrier)
            sk.commit(Ks[cur ^ 1], 1.f);
            sv.commit(Vs[cur ^ 1], 1.f);
            if (c + 2 < nch) { sk.issue(base + D, ld, c0 + 2 * CHK, N); sv.issue(base + 2 * D, ld, c0 + 2 * CHK, N); }
        }
        if (wave_live) {
            const __bf16* Kc = Ks[cur];
            const __bf16* Vc = Vs[cur];
            const int kend = min(CHK, N - c0);
            for (int k0 = 0; k0 < kend; k0 += 32 * KT) {
                bf16x8 kfr[KT][NKK];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int kk = 0; kk < NKK; ++kk) kfr[kt][kk] = *reinterpret_cast<const bf16x8*>(&Kc[(k0 + kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                f32x16 s[QB][KT];
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
                            s[qb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kt][kk], qf[qb][kk], kk == 0 ? negm[qb] : s[qb][kt], 0, 0, 0);
                if (c0 + k0 + 32 * KT > N) {                 // keys past the sequence (their K rows are zero in LDS)
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) if (c0 + k0 + kt * 32 + crow(r, hi) >= N) s[qb][kt][r] = -1e30f;
                }
                const bool first = c == 0 && k0 == 0;
                float tm[QB];
                bool need = first;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    float t = max16(s[qb][0]);
#pragma unroll
                    for (int kt = 1; kt < KT; ++kt) t = fmaxf(t, max16(s[qb][kt]));
                    tm[qb] = half_pair_max(t);
                    need |= tm[qb] > SM_THR;
                }
                if (__any(need)) {
                    // move the reference point: the first step sets it to the step's maximum, later ones only ever raise it
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const float delta = first ? tm[qb] : fmaxf(tm[qb], 0.f);
                        const float corr = __builtin_amdgcn_exp2f(-delta);
                        m[qb] += delta;
                        lsum[qb] *= corr;
                        negm[qb] = bcast16(-m[qb]);
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) s[qb][kt][r] -= delta;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int a = 0; a < NACC; ++a)
#pragma unroll
                                for (int r = 0; r < 16; ++r) oacc[qb][nt][a][r] *= corr;
                    }
                }
                bf16x8 pf[QB][KT][2];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    float ps = 0.f;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            s[qb][kt][r] = __builtin_amdgcn_exp2f(s[qb][kt][r]);
                            ps += s[qb][kt][r];
                        }
                        pack16(s[qb][kt], pf[qb][kt]);
                    }
                    lsum[qb] += ps;
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const bf16x8 a = tr_frag<LD>(Vc, k0 + kt * 32, s2, 32 * nt, lane);
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb)
                                oacc[qb][nt][(2 * kt + s2) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[qb][kt][s2], oacc[qb][nt][(2 * kt + s2) % NACC], 0, 0, 0);
                        }
            }
        }
        __syncthreads();                                     // chunk c + 1 is visible, chunk c's buffer is free
    }
    if (!wave_live) return;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0 + 32 * qb + l31;
        const float ltot = lsum[qb] + __shfl_xor(lsum[qb], 32, 64);
        if (qrow < N) {
            const float inv = 1.f / ltot;
            float* orow = o + ((long)b * N + qrow) * D + h * HD;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (NACC == 2 ? oacc[qb][nt][0][4 * g + e] + oacc[qb][nt][NACC - 1][4 * g + e] : oacc[qb][nt][0][4 * g + e]) * inv;
                    *reinterpret_cast<f32x4*>(orow + 32 * nt + 8 * g + 4 * hi) = v;
                    if (o16) {
                        bf16x4 v16;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v16[e] = (__bf16)v[e];
                        *reinterpret_cast<bf16x4*>(o16 + ((long)b * N + qrow) * D + h * HD + 32 * nt + 8 * g + 4 * hi) = v16;
                    }
                }
            if (hi == 0) lse[((long)b * H + h) * N + qrow] = (m[qb] + __builtin_amdgcn_logf(ltot)) * LN2;
        }
    }
}

// ------------------------------------------------------------------------------- backward: dQ (+ delta)
template <int HD, int QB, typename QT = float>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const QT* __restrict__ qkv, const float* __restrict__ o,
                                                               const float* __restrict__ d_o, const float* __restrict__ lse,
                                                               float* __restrict__ dqkv, __bf16* __restrict__ dqkv16,
                                                               float* __restrict__ dbias, float* __restrict__ delta, int N,
                                                               int H, float scale) {
    constexpr int LD = HD + 8, NKK = HD / 16, NT = HD / 32;
    __shared__ __attribute__((aligned(16))) __bf16 Ks[2][CH * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[2][CH * LD];
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int D = H * HD;
    const long ld = 3L * D;
    const QT* base = qkv + (long)b * N * ld + h * HD;
    const int q0 = (blockIdx.x * 4 + wave) * (32 * QB);
    const bool wave_live = q0 < N;
    bf16x8 qf[QB][NKK], gf[QB][NKK];
    f32x16 negL[QB], negD[QB], acc[QB][NT];
    float dlv[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0 + 32 * qb + l31;
        const bool qvalid = qrow < N;
        const float* grow = d_o + ((long)b * N + qrow) * D + h * HD;
        const float* orow = o + ((long)b * N + qrow) * D + h * HD;
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int cc = 16 * kk + 8 * hi;
            qf[qb][kk] = load_frag(base + (long)qrow * ld + cc, qvalid, scale * LOG2E);
            f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, o0 = g0, o1 = g0;
            if (qvalid) {
                g0 = *reinterpret_cast<const f32x4*>(grow + cc); g1 = *reinterpret_cast<const f32x4*>(grow + cc + 4);
                o0 = *reinterpret_cast<const f32x4*>(orow + cc); o1 = *reinterpret_cast<const f32x4*>(orow + cc + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dl += g0[e] * o0[e] + g1[e] * o1[e];
            gf[qb][kk] = cvt8(g0, g1, 1.f);
        }
        dl += __shfl_xor(dl, 32, 64);
        dlv[qb] = dl;
        const float Lq = qvalid ? lse[((long)b * H + h) * N + qrow] * LOG2E : 0.f;
        negL[qb] = bcast16(-Lq);
        negD[qb] = bcast16(-dl);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[qb][nt] = zero16();
    }
    ChunkStager<HD, QT> sk, sv;
    const int nch = (N + CH - 1) / CH;
    sk.issue(base + D, ld, 0, N);
    sv.issue(base + 2 * D, ld, 0, N);
    sk.commit(Ks[0], 1.f);
    sv.commit(Vs[0], 1.f);
    if (nch > 1) { sk.issue(base + D, ld, CH, N); sv.issue(base + 2 * D, ld, CH, N); }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int cur = c & 1, c0 = c * CH;
        if (c + 1 < nch) {
            sk.commit(Ks[cur ^ 1], 1.f);
            sv.commit(Vs[cur ^ 1], 1.f);
            if (c + 2 < nch) { sk.issue(base + D, ld, c0 + 2 * CH, N); sv.issue(base + 2 * D, ld, c0 + 2 * CH, N); }
        }
        if (wave_live) {
            const __bf16* Kc = Ks[cur];
            const __bf16* Vc = Vs[cur];
            const int kend = min(CH, N - c0);
            for (int kt = 0; kt * 32 < kend; ++kt) {
                bf16x8 kfr[NKK], vfr[NKK];
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    kfr[kk] = *reinterpret_cast<const bf16x8*>(&Kc[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    vfr[kk] = *reinterpret_cast<const bf16x8*>(&Vc[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                }
                const bool tail = c0 + kt * 32 + 32 > N;
                bf16x8 dsf[QB][2];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    // S' = K Q^T - lse and dP' = V dO^T - delta straight out of the matrix pipe
                    f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0], qf[qb][0], negL[qb], 0, 0, 0);
                    f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0], gf[qb][0], negD[qb], 0, 0, 0);
#pragma unroll
                    for (int kk = 1; kk < NKK; ++kk) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kk], qf[qb][kk], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kk], gf[qb][kk], dp, 0, 0, 0);
                    }
                    if (tail) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) if (c0 + kt * 32 + crow(r, hi) >= N) s[r] = -1e30f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]) * dp[r];
                    pack16(s, dsf[qb]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 a = tr_frag<LD>(Kc, kt * 32, s2, 32 * nt, lane);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) acc[qb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsf[qb][s2], acc[qb][nt], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0 + 32 * qb + l31;
        const bool qvalid = qrow < N;
        if (dbias) {
            float cs[16 * NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cs[16 * nt + r] = acc[qb][nt][r] * scale;
            colsum_to<16 * NT>(cs, wave_live && qvalid, dbias + h * HD, reinterpret_cast<float*>(&Ks[0][0]), hi, l31, wave);
        }
        if (!wave_live || !qvalid) continue;
        const long ooff = ((long)b * N + qrow) * ld + h * HD;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[qb][nt][4 * g] * scale, acc[qb][nt][4 * g + 1] * scale, acc[qb][nt][4 * g + 2] * scale, acc[qb][nt][4 * g + 3] * scale};
                if (dqkv) *reinterpret_cast<f32x4*>(dqkv + ooff + 32 * nt + 8 * g + 4 * hi) = v;
                if (dqkv16) {
                    bf16x4 v16;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v16[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>(dqkv16 + ooff + 32 * nt + 8 * g + 4 * hi) = v16;
                }
            }
        if (hi == 0) delta[((long)b * H + h) * N + qrow] = dlv[qb];
    }
}

// ------------------------------------------------------------------------------- backward: dK, dV
template <int HD, int KB, typename QT = float>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const QT* __restrict__ qkv, const float* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dqkv, __bf16* __restrict__ dqkv16,
                                                                float* __restrict__ dbias, int N, int H, float scale) {
    constexpr int LD = HD + 8, NKK = HD / 16, NT = HD / 32;
    __shared__ __attribute__((aligned(16))) __bf16 Qs[2][CH * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Gs[2][CH * LD];
    __shared__ __attribute__((aligned(16))) float Ls[2][CH];      // -lse * log2(e) of the chunk's queries (the C operand of S)
    __shared__ __attribute__((aligned(16))) float Ds[2][CH];      // -delta (the C operand of dP)
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int D = H * HD;
    const long ld = 3L * D;
    const QT* base = qkv + (long)b * N * ld + h * HD;
    const int k0 = (blockIdx.x * 4 + wave) * (32 * KB);
    const bool wave_live = k0 < N;
    bf16x8 kf[KB][NKK], vf[KB][NKK];
    f32x16 dk[KB][NT], dv[KB][NT];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int krow = k0 + 32 * kb + l31;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            kf[kb][kk] = load_frag(base + (long)krow * ld + D + 16 * kk + 8 * hi, krow < N, 1.f);
            vf[kb][kk] = load_frag(base + (long)krow * ld + 2 * D + 16 * kk + 8 * hi, krow < N, 1.f);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { dk[kb][nt] = zero16(); dv[kb][nt] = zero16(); }
    }
    const float* gbase = d_o + (long)b * N * D + h * HD;
    const float* lrow = lse + ((long)b * H + h) * N;
    const float* drow = delta + ((long)b * H + h) * N;
    ChunkStager<HD, QT> sq;
    ChunkStager<HD, float> sg;
    float nl = -1e30f, nd = 0.f;                           // -lse / -delta of this thread's query of the staged chunk (threads < CH)
    auto issue_chunk = [&](int c0) {
        sq.issue(base, ld, c0, N);
        sg.issue(gbase, D, c0, N);
        if (threadIdx.x < CH) {
            const int q = c0 + threadIdx.x;
            nl = q < N ? -lrow[q] * LOG2E : -1e30f;          // invalid query -> P = exp2(-huge) = 0
            nd = q < N ? -drow[q] : 0.f;
        }
    };
    auto commit_chunk = [&](int buf) {
        sq.commit(Qs[buf], scale * LOG2E);
        sg.commit(Gs[buf], 1.f);
        if (threadIdx.x < CH) { Ls[buf][threadIdx.x] = nl; Ds[buf][threadIdx.x] = nd; }
    };
    const int nch = (N + CH - 1) / CH;
    issue_chunk(0);
    commit_chunk(0);
    if (nch > 1) issue_chunk(CH);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int cur = c & 1, c0 = c * CH;
        if (c + 1 < nch) {
            commit_chunk(cur ^ 1);
            if (c + 2 < nch) issue_chunk(c0 + 2 * CH);
        }
        if (wave_live) {
            const __bf16* Qc = Qs[cur];
            const __bf16* Gc = Gs[cur];
            const float* Lc = Ls[cur];
            const float* Dc = Ds[cur];
            const int qend = min(CH, N - c0);
            for (int qt = 0; qt * 32 < qend; ++qt) {
                bf16x8 aq[NKK], ag[NKK];
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    aq[kk] = *reinterpret_cast<const bf16x8*>(&Qc[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    ag[kk] = *reinterpret_cast<const bf16x8*>(&Gc[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                }
                f32x16 cL, cD;                               // rows crow(r, hi) of the tile: four 16-byte groups each
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 L4 = *reinterpret_cast<const f32x4*>(&Lc[qt * 32 + 8 * g + 4 * hi]);
                    const f32x4 D4 = *reinterpret_cast<const f32x4*>(&Dc[qt * 32 + 8 * g + 4 * hi]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cL[4 * g + e] = L4[e]; cD[4 * g + e] = D4[e]; }
                }
                bf16x8 pf[KB][2], dsf[KB][2];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[0], kf[kb][0], cL, 0, 0, 0);
                    f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[0], vf[kb][0], cD, 0, 0, 0);
#pragma unroll
                    for (int kk = 1; kk < NKK; ++kk) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[kk], kf[kb][kk], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag[kk], vf[kb][kk], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r]); dp[r] *= s[r]; }
                    pack16(s, pf[kb]);
                    pack16(dp, dsf[kb]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 agt = tr_frag<LD>(Gc, qt * 32, s2, 32 * nt, lane);
                        const bf16x8 aqt = tr_frag<LD>(Qc, qt * 32, s2, 32 * nt, lane);
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) {
                            dv[kb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(agt, pf[kb][s2], dv[kb][nt], 0, 0, 0);
                            dk[kb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aqt, dsf[kb][s2], dk[kb][nt], 0, 0, 0);
                        }
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int krow = k0 + 32 * kb + l31;
        const bool kvalid = krow < N;
        if (dbias) {
            float ck[16 * NT], cv[16 * NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { ck[16 * nt + r] = dk[kb][nt][r] * LN2; cv[16 * nt + r] = dv[kb][nt][r]; }
            colsum_to<16 * NT>(ck, wave_live && kvalid, dbias + D + h * HD, reinterpret_cast<float*>(&Qs[0][0]), hi, l31, wave);
            colsum_to<16 * NT>(cv, wave_live && kvalid, dbias + 2 * D + h * HD, reinterpret_cast<float*>(&Gs[0][0]), hi, l31, wave);
        }
        if (!wave_live || !kvalid) continue;
        const long ooff = ((long)b * N + krow) * ld + h * HD;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * nt + 8 * g + 4 * hi;
                f32x4 vk = {dk[kb][nt][4 * g] * LN2, dk[kb][nt][4 * g + 1] * LN2, dk[kb][nt][4 * g + 2] * LN2, dk[kb][nt][4 * g + 3] * LN2};
                f32x4 vv = {dv[kb][nt][4 * g], dv[kb][nt][4 * g + 1], dv[kb][nt][4 * g + 2], dv[kb][nt][4 * g + 3]};
                if (dqkv) {
                    *reinterpret_cast<f32x4*>(dqkv + ooff + D + d) = vk;       // Qs carried scale*log2e: dK = sum dS * scale * Q
                    *reinterpret_cast<f32x4*>(dqkv + ooff + 2 * D + d) = vv;
                }
                if (dqkv16) {
                    bf16x4 k16, v16;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { k16[e] = (__bf16)vk[e]; v16[e] = (__bf16)vv[e]; }
                    *reinterpret_cast<bf16x4*>(dqkv16 + ooff + D + d) = k16;
                    *reinterpret_cast<bf16x4*>(dqkv16 + ooff + 2 * D + d) = v16;
                }
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
#ifndef VITAE_ATTN_FUSED_MINW
#define VITAE_ATTN_FUSED_MINW 2     // waves per SIMD the one-launch backward's registers are cut for.  3 (hd 64: 208 -> 168 VGPRs, 5 spilled,
                                    // three workgroups per CU) was measured: 768 heads of 55 tokens 17.4 -> 16.6 us, 96 heads 6.8 -> 7.1: not kept
#endif
__global__ __launch_bounds__(256, VITAE_ATTN_FUSED_MINW) void attn_bwd_fused_kernel(const QT* __restrict__ qkv, const float* __restrict__ o,
                                                             const float* __restrict__ d_o, const float* __restrict__ lse,
                                                             int N, int H, float scale, int NP,      // (what the loads need: inside the 14 preloaded dwords)
                                                             float* __restrict__ dqkv, __bf16* __restrict__ dqkv16,
                                                             float* __restrict__ dbias) {
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
        if ((idx % V4) == 0) Ds[row] = -dl;        // rows >= N: 0.  NEGATED: the C operand of the dP products
    }
    for (int row = threadIdx.x; row < NP; row += 256)
        Ls[row] = row < N ? -lse[((long)b * H + h) * N + row] * LOG2E : -1e30f;    // (negated) invalid query: P = exp2(s - huge) = 0
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
            // S - lse and dP - delta come out of the matrix pipe: -lse / -delta of the lane's query are the C operands
            const f32x16 negL = bcast16(Ls[qrow]), negD = bcast16(Ds[qrow]);
            f32x16 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
            for (int kt = 0; kt < T; ++kt) {
                f32x16 sc, dp;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 ak = *reinterpret_cast<const bf16x8*>(&Ks[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(&Vs[(kt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[kk], kk == 0 ? negL : sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, gf[kk], kk == 0 ? negD : dp, 0, 0, 0);
                }
                if (kt * 32 + 32 > N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (kt * 32 + crow(r, hi) >= N) sc[r] = -1e30f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]) * dp[r];
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
                f32x16 cL, cD;                               // -lse / -delta of the tile's query rows crow(r, hi): the C operands
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 L4 = *reinterpret_cast<const f32x4*>(&Ls[qt * 32 + 8 * g + 4 * hi]);
                    const f32x4 D4 = *reinterpret_cast<const f32x4*>(&Ds[qt * 32 + 8 * g + 4 * hi]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cL[4 * g + e] = L4[e]; cD[4 * g + e] = D4[e]; }
                }
                f32x16 sc, dp;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 aq = *reinterpret_cast<const bf16x8*>(&Qs[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    const bf16x8 ag = *reinterpret_cast<const bf16x8*>(&Gs[(qt * 32 + l31) * LD + 16 * kk + 8 * hi]);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[kk], kk == 0 ? cL : sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, vf[kk], kk == 0 ? cD : dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r]); dp[r] *= sc[r]; }
                bf16x8 pf[2], dsf[2];
                pack16(sc, pf);
                pack16(dp, dsf);
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
// rows (32-row blocks) per wave of the streaming kernels: two blocks halve the LDS fragment reads per MFMA and the workgroup count; one
// block keeps short sequences spread over more waves.  VITAE_ATTN_RB = 1 / 2 forces it (tools/attn_bench.py).
static int attn_row_blocks(int N, int head_dim, int which /* 0 fwd, 1 dq, 2 dkv */) {
    static const int forced = getenv("VITAE_ATTN_RB") ? atoi(getenv("VITAE_ATTN_RB")) : 0;
    if (forced == 1 || forced == 2) return (forced == 2 && head_dim == 64 && which == 2) ? 1 : forced;
    (void)N;
    return 1;   // two blocks per wave (253-256 VGPRs: one wave per SIMD and half the workgroups) measured 177 against 164 us at N = 1729
}

template <typename QT>
static int sdpa_mfma_fwd_launch(const QT* qkv, float* o, void* o_bf16, float* lse, int B, int N, int H, int head_dim, void* stream) {
    if (!qkv || !o || !lse || B <= 0 || N <= 0 || H <= 0) return VITAE_ERR_INVALID_ARG;
    if ((((long)H * head_dim) & 7) || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const float scale = 1.0f / sqrtf((float)head_dim);
    // forward variants (VITAE_ATTN_FWD forces one): 0: 1 query block x 1 key tile per step (short sequences); 1: 1 x 2; 2: 2 x 2
    // (256-key chunks — 1 x 4 and 2 x 2 — measured no faster at N = 1729: 64.0 / 58.7 against 56.9 us; they halve the resident workgroups)
    static const int forced = getenv("VITAE_ATTN_FWD") ? atoi(getenv("VITAE_ATTN_FWD")) : -1;
    // (two query blocks per wave also for short sequences once there are enough heads to fill the chip with half the workgroups:
    // batch 32 decoder, N = 217: 22.6 -> 18.6 us; batch 4: 6.7 -> 8.9 us)
    const int var = forced >= 0 ? forced : (N >= 512 ? (head_dim == 32 ? 2 : 1) : (head_dim == 32 && N > 128 && (long)H * B >= 512) ? 2 : 0);
    const int qb = var == 2 ? 2 : 1;
    dim3 grid(cdiv(N, 128 * qb), H, B);
    hipStream_t st = (hipStream_t)stream;
    __bf16* o16 = reinterpret_cast<__bf16*>(o_bf16);
#define VITAE_FWD(HD_, QB_, KT_, CHK_) hipLaunchKernelGGL((attn_fwd_mfma_kernel<HD_, QB_, KT_, CHK_, QT>), grid, dim3(256), 0, st, qkv, o, o16, lse, N, H, scale)
    // (N <= 64 — the encoder of the 96^3 / patch-16 model keeps 55 tokens — takes a 64-key chunk: half the LDS, so three workgroups
    // share a CU instead of two and the 768 heads of a batch-32 step are one round)
    static const int chk64 = getenv("VITAE_ATTN_FWD_CHK64") ? atoi(getenv("VITAE_ATTN_FWD_CHK64")) : 1;
    if (var == 0 && N <= 64 && chk64) {
        if (head_dim == 32) VITAE_FWD(32, 1, 1, 64);
        else if (head_dim == 64) VITAE_FWD(64, 1, 1, 64);
        else return VITAE_ERR_UNSUPPORTED_SHAPE;
        return vitae_launch_status();
    }
    if (head_dim == 32) {
        switch (var) {
            case 1: VITAE_FWD(32, 1, 2, 128); break;
            case 2: VITAE_FWD(32, 2, 2, 128); break;
            default: VITAE_FWD(32, 1, 1, 128); break;
        }
    } else if (head_dim == 64) {
        switch (var) {
            case 1: VITAE_FWD(64, 1, 2, 128); break;
            case 2: VITAE_FWD(64, 2, 1, 128); break;
            default: VITAE_FWD(64, 1, 1, 128); break;
        }
    } else {
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    }
#undef VITAE_FWD
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

// Workgroups per (batch, head) of the one-launch backward: every workgroup stages the WHOLE head (Q, K, V, dO), so with many heads
// one workgroup per head is best — it walks all the head's items, four at a time (batch 32 decoder, 512 heads x 14 items: 42.2 us
// with two workgroups per head, 31.3 with one); with few heads the items are spread over up to items / 4 workgroups to fill the chip.
static int fused_bwd_groups(int items, int H, int B) {
    static const int gforce = getenv("VITAE_ATTN_BWD_G") ? atoi(getenv("VITAE_ATTN_BWD_G")) : 0;
    if (gforce > 0) return gforce;
    const long heads = (long)H * B;
    int g = cdiv(items, 4), fill = cdiv(512, heads);
    if (g > fill) g = fill;
    return g < 1 ? 1 : g;
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
    const int rq = attn_row_blocks(N, head_dim, 1), rk = attn_row_blocks(N, head_dim, 2);
    dim3 gq(cdiv(N, 128 * rq), H, B), gk(cdiv(N, 128 * rk), H, B);
#define VITAE_ATTN_BWD(HD_, RQ_, RK_)                                                                                                       \
    do {                                                                                                                                   \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<HD_, RQ_, QT>), gq, dim3(256), 0, st, qkv, o, d_o, lse, dqkv, g16, dqkv_colsum_accum, delta_ws, N, H, scale); \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<HD_, RK_, QT>), gk, dim3(256), 0, st, qkv, d_o, lse, delta_ws, dqkv, g16, dqkv_colsum_accum, N, H, scale);   \
    } while (0)
    if (head_dim == 32) {
        if (rq == 2 && rk == 2) VITAE_ATTN_BWD(32, 2, 2);
        else if (rq == 2) VITAE_ATTN_BWD(32, 2, 1);
        else if (rk == 2) VITAE_ATTN_BWD(32, 1, 2);
        else VITAE_ATTN_BWD(32, 1, 1);
    } else if (head_dim == 64) {
        if (rq == 2) VITAE_ATTN_BWD(64, 2, 1);
        else VITAE_ATTN_BWD(64, 1, 1);
    } else {
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    }
#undef VITAE_ATTN_BWD
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
    const int G = fused_bwd_groups(items, H, B);
    dim3 fgrid(G, H, B);
    hipStream_t st = (hipStream_t)stream;
    const __bf16* q16 = reinterpret_cast<const __bf16*>(qkv_bf16);
    __bf16* g16 = reinterpret_cast<__bf16*>(dqkv_bf16);
    if (head_dim == 32) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<32, __bf16>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)attr;
        hipLaunchKernelGGL((attn_bwd_fused_kernel<32, __bf16>), fgrid, dim3(256), lds, st, q16, o, d_o, lse, N, H, scale, NP, dqkv, g16,
                           dqkv_colsum_accum);
    } else {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<64, __bf16>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)attr;
        hipLaunchKernelGGL((attn_bwd_fused_kernel<64, __bf16>), fgrid, dim3(256), lds, st, q16, o, d_o, lse, N, H, scale, NP, dqkv, g16,
                           dqkv_colsum_accum);
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
        const int G = fused_bwd_groups(items, H, B);
        dim3 fgrid(G, H, B);
        if (head_dim == 32) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<32>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)attr;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<32>), fgrid, dim3(256), lds, st, qkv, o, d_o, lse, N, H, scale, NP, dqkv, g16,
                               dqkv_colsum_accum);
        } else {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<64>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)attr;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<64>), fgrid, dim3(256), lds, st, qkv, o, d_o, lse, N, H, scale, NP, dqkv, g16,
                               dqkv_colsum_accum);
        }
        return vitae_launch_status();
    }
    return sdpa_bwd_two_kernels<float>(qkv, o, d_o, lse, dqkv, g16, dqkv_colsum_accum, delta_ws, B, N, H, head_dim, scale, st);
}
