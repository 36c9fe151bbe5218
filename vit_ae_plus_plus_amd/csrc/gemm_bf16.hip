// Throughput-mode GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for every dense contraction of the
// path — same contract as gemm.hip:
//   C[M,N] (+)= epi( sum_k A(m,k) * B(n,k) + bias[n] ) (+ residual)
// A is fp32 in HBM (activations / gradients) and is rounded to bf16 while being staged; B is either
// fp32 or a bf16 weight shadow kept by the AdamW kernel (halves the dominant HBM / L2 traffic).
// Operand storage flags as in gemm.hip (*_KC: element (row,k) at ptr[row*ld + k], else ptr[k*ld + row]).
//
// MI355X-specific structure
//  * tiles 64 x 64 (BK 256) or 64 x 128 (BK 128), 4 waves (2 x 2), wave tile 32 x BN/2: M is only
//    ~440-870 rows on this path, so small M tiles keep >=256 workgroups in flight;
//  * LDS images keep the GLOBAL orientation of each operand (no transposing stores): k-contiguous
//    tiles are read with ds_read_b128, row-contiguous tiles (dgrad's W, both wgrad operands) with the
//    hardware transpose read ds_read_b64_tr_b16 — no transposed copies of weights or activations exist;
//    row strides are padded (+8 / +16 elements) so both read kinds are bank-conflict free;
//  * the K loop is latency bound at these sizes (a CU fetches ~bytes-in-flight / ~1.2 us), so a k-tile is
//    a deep "phase": up to ~100 KB of 16-byte loads per workgroup in flight while the previous phase's
//    MFMAs run from LDS (single register set + single LDS buffer, two barriers per phase);
//  * workgroup ids are remapped so that all M-tiles of one weight panel run on the same XCD (its L2
//    then serves the panel to the 7-14 workgroups that share it);
//  * split-K partials go to a workspace and are reduced deterministically (shared with gemm.hip).
#include <cstdlib>
#include "common.hpp"
#include "vitae_hip.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A; long lda;
    const void* B; long ldb;
    float* C; long ldc;
    int M, N, K;
    int k_per_split, splits;
    const float* bias;
    const float* residual; long ldr;
    float* aux; long ldaux;
    int epi, accumulate;
    float* ws;
    float* colsum;   // optional: colsum[k] += sum_m A(m,k)  (bias gradient riding on the dgrad's A tiles)
    int tiles_m, tiles_n, n_per_xcd;
};

__device__ __forceinline__ void epilogue_store(const GemmArgs& p, float v, int m, int n) {
    if (p.bias) v += p.bias[n];
    const int kind = p.epi & 15;
    const bool auxd = (p.epi & VITAE_EPI_AUX_DERIV) != 0;      // aux holds GELU'(pre-activation) instead of the pre-activation
    if (kind == VITAE_EPI_GELU) {
        if (auxd) {                                            // (the pdf's exp only when the derivative is what gets saved)
            float y, dy;
            gelu_erf_both(v, y, dy);
            p.aux[(long)m * p.ldaux + n] = dy;
            v = y;
        } else {
            p.aux[(long)m * p.ldaux + n] = v;
            v = gelu_erf(v);
        }
    } else if (kind == VITAE_EPI_DGELU) {
        const float a = p.aux[(long)m * p.ldaux + n];
        v *= auxd ? a : gelu_erf_grad(a);
    } else if (kind == VITAE_EPI_RELU_MASK) {
        v = p.aux[(long)m * p.ldaux + n] > 0.f ? v : 0.f;
    }
    if (p.residual) v += p.residual[(long)m * p.ldr + n];
    float* c = p.C + (long)m * p.ldc + n;
    if (p.accumulate) v += *c;
    *c = v;
}

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// LDS row stride (bf16 elements) of a row-contiguous tile [k][n].  ds_read_b64_tr_b16 is serviced in two
// 32-lane groups touching 4 consecutive k-rows x 64 B; fully conflict-free needs a stride of 16 or 48
// dwords mod 64 (n + 32 elements), which was MEASURED SLOWER end to end: the larger tiles (98 KB) drop the
// paired dgrad+wgrad launch from 2 to 1 workgroup per CU.  n + 8 / n + 16 keeps 2-way conflicts
// (SQ_LDS_BANK_CONFLICT = 28 % of SQ_LDS_IDX_ACTIVE in profiles/) but two resident workgroups.
__host__ __device__ constexpr int ld_pad(int n) { return n == 128 ? n + 16 : n + 8; }

// ---------------------------------------------------------------- operand staging
// One operand tile = ROWS rows x BK k.  Per thread NPIECE 16-byte global pieces.
// X3 (fp32 operands only): the LDS image is TWO bf16 tiles, hi = bf16(x) and lo = bf16(x - hi), ONE elements apart — the
// split-operand form of the fp32-accurate mode (VITAE_PREC_BF16X3): x = hi + lo to 2^-17, and the k-loop computes
// hi.hi + hi.lo + lo.hi with fp32 accumulation.
template <int ROWS, int BK, int NT, bool KC, bool BF16, bool X3 = false> struct Stage {
    static_assert(!(X3 && BF16), "the split form splits fp32 operands");
    static constexpr int EPP = BF16 ? 8 : 4;                       // elements per 16-byte piece
    static constexpr int NPIECE = ROWS * BK / EPP / NT;
    static constexpr int LD = KC ? BK + 8 : ld_pad(ROWS);          // LDS row stride in bf16 elements (conflict-free)
    static constexpr int ONE = (KC ? ROWS : BK) * LD;              // elements of one bf16 tile image
    static constexpr int BYTES = ONE * 2 * (X3 ? 2 : 1);
    u32x4 r[NPIECE];
    unsigned ok;   // bit j: piece j is inside the operand (others are zero-filled at store time)

    // Loads are UNCONDITIONAL (indices clamped into the operand) and the zero-fill of out-of-range
    // pieces happens in store(): a per-piece `if (inside) load` makes hipcc branch around every load and
    // drain vmcnt(0) each k-tile, which serialises the whole prefetch pipeline (measured: 17 us floor).
    __device__ __forceinline__ void load(const void* __restrict__ P, long ld, int rows, int r0, int k0, int kend) {
        ok = 0u;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const int p = threadIdx.x + NT * j;
            int row, k;
            if (KC) { row = p / (BK / EPP); k = (p % (BK / EPP)) * EPP; }
            else { k = p / (ROWS / EPP); row = (p % (ROWS / EPP)) * EPP; }
            const int gr = r0 + row, gk = k0 + k;
            ok |= (gr < rows && gk < kend) ? (1u << j) : 0u;
            // clamp to the last in-range piece (kend and rows are multiples of the piece width)
            const int cr = min(gr, rows - (KC ? 1 : EPP)), ck = min(gk, kend - (KC ? EPP : 1));
            const long off = KC ? ((long)cr * ld + ck) : ((long)ck * ld + cr);
            r[j] = BF16 ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(P) + off)
                        : *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(P) + off);
        }
    }

    // colsum[k0 + k] += sum over this tile's rows of the fp32 operand (k-contiguous fp32 tiles only).
    // Every piece of a thread has the same k (piece stride 256 is a multiple of the pieces per row).
    __device__ __forceinline__ void colsum_accumulate(float* __restrict__ out, int k0, int kend) const {
        static_assert(!BF16, "column sums ride on fp32 tiles");
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            union { u32x4 u; f32x4 f; } in;
            in.u = r[j];
            if ((ok >> j) & 1u) s += in.f;
        }
        const int k = k0 + (threadIdx.x % (BK / EPP)) * EPP;
        if (k < kend) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(out + k + e, s[e]);
        }
    }

    __device__ __forceinline__ void store(__bf16* lds) const {
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const int p = threadIdx.x + NT * j;
            int row, k;
            if (KC) { row = p / (BK / EPP); k = (p % (BK / EPP)) * EPP; }
            else { k = p / (ROWS / EPP); row = (p % (ROWS / EPP)) * EPP; }
            __bf16* dst = KC ? (lds + row * LD + k) : (lds + k * LD + row);
            const bool inside = (ok >> j) & 1u;
            u32x4 v = r[j];
            if (!inside) v = u32x4{0u, 0u, 0u, 0u};
            if (BF16) {
                *reinterpret_cast<u32x4*>(dst) = v;
            } else {
                union { u32x4 u; f32x4 f; } in;
                in.u = v;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)in.f[e];
                *reinterpret_cast<bf16x4*>(dst) = o;
                if constexpr (X3) {
                    bf16x4 l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = (__bf16)(in.f[e] - (float)o[e]);
                    *reinterpret_cast<bf16x4*>(dst + ONE) = l;
                }
            }
        }
    }
};

// MFMA operand fragment (32 rows x 16 k) for rows row0 + (lane & 31), k-slots kk*16 + 8*hi + e
template <bool KC, int LD>
__device__ __forceinline__ bf16x8 frag(const __bf16* T, int row0, int kk, int lane) {
    if (KC) {
        return *reinterpret_cast<const bf16x8*>(T + (row0 + (lane & 31)) * LD + kk * 16 + 8 * (lane >> 5));
    } else {
        const int gg = lane >> 4, li = lane & 15;
        const int k = kk * 16 + 8 * (gg >> 1) + (li >> 2);
        const int col = row0 + 16 * (gg & 1) + 4 * (li & 3);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        union { s16x4 s[2]; bf16x8 b; } u;
        u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(T + k * LD + col));
        u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(T + (k + 4) * LD + col));
        return u.b;
    }
}

// Tile configurations (BM x BN x BK, threads): 64x64x256/256 (default), 64x128x128/256 (large GEMMs),
// 32x64x256/128 (small GEMMs: twice the workgroups, 2-3 resident per CU so one's MFMA loop overlaps
// another's load wait).  Waves form a (BM/32) x (rest) grid; wave tile 32 x (32*FN).
template <int BM, int BN, int BK, int NT, bool A_KC, bool B_KC, bool B_BF16, bool X3 = false> struct TileCfg {
    using SA = Stage<BM, BK, NT, A_KC, false, X3>;
    using SB = Stage<BN, BK, NT, B_KC, B_BF16, X3>;
    static constexpr int SMEM = SA::BYTES + SB::BYTES;
    static constexpr int KS = NT > 256 ? NT / 256 : 1;            // wave groups splitting each phase's k range
    static constexpr int WM = BM / 32, WN = (NT / 64 / KS) / WM, FN = BN / WN / 32;
};

template <int BM, int BN, int BK, int NT, bool A_KC, bool B_KC, bool B_BF16, bool X3 = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int bid, const int zid, unsigned char* smem) {
    using CFG = TileCfg<BM, BN, BK, NT, A_KC, B_KC, B_BF16, X3>;
    using SA = typename CFG::SA;
    using SB = typename CFG::SB;
    constexpr int FN = CFG::FN, WN = CFG::WN, KS = CFG::KS;
    // XCD-aware tile order: hardware places workgroup b on XCD b % 8; give each XCD whole weight panels.
    const int xcd = bid & 7, local = bid >> 3;
    const int tn = xcd + 8 * (local / p.tiles_m), tm = local % p.tiles_m;
    if (tn >= p.tiles_n) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kh = wave / (NT / 64 / KS), wq = wave % (NT / 64 / KS);   // k-group, position in the wave grid
    const int wm = wq / WN, wn = wq % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    f32x16 acc[FN];
#pragma unroll
    for (int f = 0; f < FN; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;

    // One k-tile of BK = 128 / 256 is a whole "phase": all of its 16-byte loads (up to ~100 KB per
    // workgroup) are in flight together while the previous phase's MFMAs run from LDS.  Per-CU fetch
    // rate here is set by bytes in flight (latency bound), so few large phases beat many small tiles,
    // and one register set + one LDS buffer is the structure hipcc keeps intact (it collapses a
    // two-register-set software pipeline into a single in-flight tile).
    SA ra;
    SB rb;
    __bf16* at = reinterpret_cast<__bf16*>(smem);
    __bf16* bt = reinterpret_cast<__bf16*>(smem + SA::BYTES);
    if (nk > 0) {
        ra.load(p.A, p.lda, p.M, m0, kbeg, kend);
        rb.load(p.B, p.ldb, p.N, n0, kbeg, kend);
    }
    const bool do_colsum = A_KC && p.colsum != nullptr && tn == 0;
    for (int t = 0; t < nk; ++t) {
        if constexpr (A_KC) { if (do_colsum) ra.colsum_accumulate(p.colsum, kbeg + t * BK, kend); }
        ra.store(at);
        rb.store(bt);
        __syncthreads();
        if (t + 1 < nk) {
            ra.load(p.A, p.lda, p.M, m0, kbeg + (t + 1) * BK, kend);
            rb.load(p.B, p.ldb, p.N, n0, kbeg + (t + 1) * BK, kend);
        }
#pragma unroll
        for (int kq = 0; kq < BK / 16 / KS; ++kq) {
            const int kk = kh * (BK / 16 / KS) + kq;
            const bf16x8 fa = frag<A_KC, SA::LD>(at, wm * 32, kk, lane);
            bf16x8 fal;
            if constexpr (X3) fal = frag<A_KC, SA::LD>(at + SA::ONE, wm * 32, kk, lane);
#pragma unroll
            for (int f = 0; f < FN; ++f) {
                const bf16x8 fb = frag<B_KC, SB::LD>(bt, wn * (32 * FN) + f * 32, kk, lane);
                if constexpr (X3) {   // the two cross terms first: they are 2^-8 of the main term
                    const bf16x8 fbl = frag<B_KC, SB::LD>(bt + SB::ONE, wn * (32 * FN) + f * 32, kk, lane);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fbl, acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal, fb, acc[f], 0, 0, 0);
                }
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[f], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if constexpr (KS > 1) {
        // the k-groups' partial tiles meet in LDS (tile buffers are free after the last barrier)
        float* red = reinterpret_cast<float*>(smem);
        if (kh > 0) {
#pragma unroll
            for (int f = 0; f < FN; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((kh - 1) * (NT / 64 / KS) + wq) * FN + f) * 1024 + r * 64 + lane] = acc[f][r];
        }
        __syncthreads();
        if (kh > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g)
#pragma unroll
            for (int f = 0; f < FN; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][r] += red[(((g - 1) * (NT / 64 / KS) + wq) * FN + f) * 1024 + r * 64 + lane];
    }
#pragma unroll
    for (int f = 0; f < FN; ++f) {
        const int n = n0 + wn * (32 * FN) + f * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + crow(r, hi);
            if (m >= p.M) continue;
            if (p.splits > 1) p.ws[((long)zid * p.M + m) * p.N + n] = acc[f][r];
            else epilogue_store(p, acc[f][r], m, n);
        }
    }
}

template <int BM, int BN, int BK, int NT, bool A_KC, bool B_KC, bool B_BF16, bool X3 = false>
__global__ __launch_bounds__(NT) void gemm_bf16_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[TileCfg<BM, BN, BK, NT, A_KC, B_KC, B_BF16, X3>::SMEM];
    gemm_body<BM, BN, BK, NT, A_KC, B_KC, B_BF16, X3>(p, blockIdx.x, blockIdx.z, smem);
}

// One launch, two independent GEMMs that consume the same dy: the dgrad (dy @ W, bf16 weight shadow read
// row-contiguous) and the wgrad (dy^T @ x).  Each alone leaves most CUs idle or latency-stalled at these
// sizes; together they double the resident workgroups per CU and share one launch boundary.
template <int NT, int BM1, int BN1, int BK1, int BM2, int BN2, int BK2>
__global__ __launch_bounds__(NT) void gemm_bf16_pair_kernel(const GemmArgs p1, const GemmArgs p2, const int nb1) {
    constexpr int S1 = TileCfg<BM1, BN1, BK1, NT, true, false, true>::SMEM;
    constexpr int S2 = TileCfg<BM2, BN2, BK2, NT, false, false, false>::SMEM;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S1 > S2 ? S1 : S2];
    if ((int)blockIdx.x < nb1) gemm_body<BM1, BN1, BK1, NT, true, false, true>(p1, blockIdx.x, 0, smem);
    else gemm_body<BM2, BN2, BK2, NT, false, false, false>(p2, blockIdx.x - nb1, 0, smem);
}

__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const GemmArgs p) {
    const long total = (long)p.M * p.N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.ws[(long)s * total + i];
        epilogue_store(p, v, (int)(i / p.N), (int)(i % p.N));
    }
}

template <int BM, int BN, int BK, int NT, bool B_BF16, bool X3 = false>
void launch(const GemmArgs& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t st) {
    dim3 block(NT);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NT, true, true, B_BF16, X3>), grid, block, 0, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NT, true, false, B_BF16, X3>), grid, block, 0, st, p);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NT, false, true, B_BF16, X3>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NT, false, false, B_BF16, X3>), grid, block, 0, st, p);
}

}  // namespace

// Tile configuration: cfg 0 = 32x64x256 (128 threads), 1 = 64x64x256, 2 = 64x128x128,
// 3 = 64x64x256 with 512 threads (two wave groups split each phase's k range: 2 waves per SIMD hide the
// LDS / MFMA latency of the fragment loop; partial tiles meet in LDS once).
static inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static inline int pick_cfg(int M, int N) {
    static const int big = env_int("VITAE_BN128_MIN_TILES", 512), small = env_int("VITAE_BM32_MAX_TILES", 0);
    static const int w8 = env_int("VITAE_GEMM_8WAVES", 0);   // faster on L2-warm operands, slower in the real (cold-weight) step
    if (N >= 128 && (long)cdiv(M, 64) * cdiv(N, 128) >= big) return 2;
    if ((long)cdiv(M, 64) * cdiv(N, 64) < small) return 0;
    return w8 ? 3 : 1;
}
static inline int cfg_bm(int c) { return c == 0 ? 32 : 64; }
static inline int cfg_nt(int c) { return c == 0 ? 128 : (c == 3 ? 512 : 256); }
static inline int cfg_bn(int c) { return c == 2 ? 128 : 64; }
static inline int cfg_bk(int c) { return c == 2 ? 128 : 256; }

extern "C" int vitae_gemm_bf16_pick_split_k(int M, int N, int K) {
    const int c = pick_cfg(M, N);
    const long tiles = (long)cdiv(M, cfg_bm(c)) * cdiv(N, cfg_bn(c));
    if (tiles >= 256 || K < 1024) return 1;
    long s = (512 + tiles - 1) / tiles;
    const long max_by_k = K / 512;   // keep >= 2 phases per split
    if (s > max_by_k) s = max_by_k;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
}

extern "C" int vitae_gemm_bf16(int a_kcontig, int b_kcontig, const float* A, long lda, const void* B, long ldb,
                               int b_is_bf16, float* C, long ldc, int M, int N, int K, const float* bias,
                               const float* residual, long ldr, int epi, float* aux, long ldaux, int accumulate,
                               int split_k, float* splitk_ws, float* a_colsum_accum, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    if (a_colsum_accum && !a_kcontig) return VITAE_ERR_INVALID_ARG;
    if (epi != VITAE_EPI_NONE && !aux) return VITAE_ERR_INVALID_ARG;
    if (epi & VITAE_EPI_AUX_BF16) return VITAE_ERR_UNSUPPORTED_SHAPE;      // (fp32 aux only in this family)
    if ((epi & VITAE_EPI_AUX_DERIV) && (epi & 15) != VITAE_EPI_GELU && (epi & 15) != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;   // (as vitae_gemm_glds does)
    const int b_epp = b_is_bf16 ? 8 : 4;
    const int a_vec = a_kcontig ? K : M, b_vec = b_kcontig ? K : N;
    if ((a_vec & 3) || (lda & 3) || (b_vec % b_epp) || (ldb % b_epp)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (split_k < 1) split_k = 1;
    if ((epi & 15) == VITAE_EPI_GELU) split_k = 1;
    GemmArgs p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    const int c = pick_cfg(M, N);
    const int bm = cfg_bm(c), bn = cfg_bn(c), bk = cfg_bk(c);
    int kps = cdiv(cdiv(K, split_k), bk) * bk;
    split_k = cdiv(K, kps);
    if (split_k > 1 && !splitk_ws) return VITAE_ERR_INVALID_ARG;
    p.k_per_split = kps; p.splits = split_k;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux;
    p.epi = epi; p.accumulate = accumulate; p.ws = splitk_ws; p.colsum = a_colsum_accum;
    p.tiles_m = cdiv(M, bm); p.tiles_n = cdiv(N, bn);
    p.n_per_xcd = cdiv(p.tiles_n, 8);
    dim3 grid(8 * p.n_per_xcd * p.tiles_m, 1, split_k);
    hipStream_t st = (hipStream_t)stream;
    const bool akc = a_kcontig != 0, bkc = b_kcontig != 0;
    if (c == 2) { if (b_is_bf16) launch<64, 128, 128, 256, true>(p, akc, bkc, grid, st); else launch<64, 128, 128, 256, false>(p, akc, bkc, grid, st); }
    else if (c == 3) { if (b_is_bf16) launch<64, 64, 256, 512, true>(p, akc, bkc, grid, st); else launch<64, 64, 256, 512, false>(p, akc, bkc, grid, st); }
    else if (c == 1) { if (b_is_bf16) launch<64, 64, 256, 256, true>(p, akc, bkc, grid, st); else launch<64, 64, 256, 256, false>(p, akc, bkc, grid, st); }
    else { if (b_is_bf16) launch<32, 64, 256, 128, true>(p, akc, bkc, grid, st); else launch<32, 64, 256, 128, false>(p, akc, bkc, grid, st); }
    if (split_k > 1) {
        const long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(blocks), dim3(256), 0, st, p);
    }
    return vitae_launch_status();
}

// Split of the reduction for the split-operand kernels below: their k-loop is bound by the latency of each 128-deep phase's fp32
// loads (one phase in flight per workgroup), so a GEMM with fewer workgroups than CUs is cut until it has them (swept in the step:
// 256 workgroups / 256-deep splits 8.79 ms; 512: 9.27; 768: 9.48; never: 10.38; the bf16 rule of this file: 9.13).
extern "C" int vitae_gemm_bf16x3_pick_split_k(int M, int N, int K) {
    static const int want = env_int("VITAE_X3_SPLIT_WGS", 256), min_k = env_int("VITAE_X3_SPLIT_MIN_K", 256);
    const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
    if (tiles >= want || K < 2 * min_k) return 1;
    long s = (want + tiles - 1) / tiles;
    if (s > K / min_k) s = K / min_k;
    if (s > 16) s = 16;
    return s < 1 ? 1 : (int)s;
}

// fp32-accurate mode (VITAE_PREC_BF16X3; vitae_gemm / vitae_linear_* with prec = 2 land here): both operands fp32 in HBM, split
// into bf16 hi + lo while they are staged, three MFMAs per fragment pair (hi.hi + hi.lo + lo.hi; the dropped lo.lo term is
// 2^-16 of a product).  The reference computes this path in fp32 (autocast off, utils/train_one_epoch.py:50): this mode keeps
// its losses within the 1e-4 bound at 8/3 of the exact-fp32 MFMA rate (v_mfma_f32_32x32x2_f32: 157 TFLOP/s).
// Tiles: 64 x 64 x 128 (two tile images per operand: 70 KB, two workgroups per CU) or 64 x 128 x 64 for GEMMs with many tiles.
extern "C" int vitae_gemm_bf16x3(int a_kcontig, int b_kcontig, const float* A, long lda, const float* B, long ldb,
                                 float* C, long ldc, int M, int N, int K, const float* bias, const float* residual, long ldr,
                                 int epi, float* aux, long ldaux, int accumulate, int split_k, float* splitk_ws, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    if (epi != VITAE_EPI_NONE && !aux) return VITAE_ERR_INVALID_ARG;
    if (epi & VITAE_EPI_AUX_BF16) return VITAE_ERR_UNSUPPORTED_SHAPE;      // (fp32 aux only in this family)
    if ((epi & VITAE_EPI_AUX_DERIV) && (epi & 15) != VITAE_EPI_GELU && (epi & 15) != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;   // (as vitae_gemm_glds does)
    const int a_vec = a_kcontig ? K : M, b_vec = b_kcontig ? K : N;
    if ((a_vec & 3) || (lda & 3) || (b_vec & 3) || (ldb & 3)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (split_k < 1) split_k = 1;
    if ((epi & 15) == VITAE_EPI_GELU) split_k = 1;
    static const int big = env_int("VITAE_X3_BN128_MIN_TILES", 512), wide_bk = env_int("VITAE_X3_BN128_BK", 64);
    const bool wide = N >= 128 && (long)cdiv(M, 64) * cdiv(N, 128) >= big;
    const int bn = wide ? 128 : 64, bk = wide ? wide_bk : 128;   // (64 x 64 x 256 — 135 KB, one workgroup per CU — measured no faster)
    GemmArgs p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    int kps = cdiv(cdiv(K, split_k), bk) * bk;
    split_k = cdiv(K, kps);
    if (split_k > 1 && !splitk_ws) return VITAE_ERR_INVALID_ARG;
    p.k_per_split = kps; p.splits = split_k;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux;
    p.epi = epi; p.accumulate = accumulate; p.ws = splitk_ws; p.colsum = nullptr;
    p.tiles_m = cdiv(M, 64); p.tiles_n = cdiv(N, bn);
    p.n_per_xcd = cdiv(p.tiles_n, 8);
    dim3 grid(8 * p.n_per_xcd * p.tiles_m, 1, split_k);
    hipStream_t st = (hipStream_t)stream;
    const bool akc = a_kcontig != 0, bkc = b_kcontig != 0;
    if (!wide) launch<64, 64, 128, 256, false, true>(p, akc, bkc, grid, st);
    else if (bk == 64) launch<64, 128, 64, 256, false, true>(p, akc, bkc, grid, st);
    else launch<64, 128, 128, 256, false, true>(p, akc, bkc, grid, st);
    if (split_k > 1) {
        const long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(blocks), dim3(256), 0, st, p);
    }
    return vitae_launch_status();
}

// Backward of one Linear in ONE launch (bf16 MFMA): dx[M,K] (+)= epi(dy[M,N] @ W[N,K]) with W read from its
// bf16 shadow, dW[N,K] (+)= dy^T @ x, db[N] += colsum(dy).  No split-K (the two GEMMs fill the chip together).
extern "C" int vitae_linear_bwd_pair_bf16(const float* dy, const void* w_bf16, const float* x, float* dx, float* dw,
                                          float* db_accum, int M, int N, int K, int epi, float* aux,
                                          int dx_accumulate, int dw_accumulate, void* stream) {
    if (!dy || !w_bf16 || !x || !dx || !dw || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    if (epi != VITAE_EPI_NONE && !aux) return VITAE_ERR_INVALID_ARG;
    if (epi & VITAE_EPI_AUX_BF16) return VITAE_ERR_UNSUPPORTED_SHAPE;      // (fp32 aux only in this family)
    if ((epi & VITAE_EPI_AUX_DERIV) && (epi & 15) != VITAE_EPI_GELU && (epi & 15) != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;   // (as vitae_gemm_glds does)
    if ((N & 7) || (K & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)dy & 15) || ((uintptr_t)w_bf16 & 15) || ((uintptr_t)x & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    GemmArgs p1, p2;
    // dgrad: rows M, cols K (in-features), reduction N
    p1.A = dy; p1.lda = N; p1.B = w_bf16; p1.ldb = K; p1.C = dx; p1.ldc = K;
    p1.M = M; p1.N = K; p1.K = N;
    int c1 = pick_cfg(M, K), c2 = pick_cfg(N, K);
    if (cfg_nt(c1) != cfg_nt(c2)) {   // one block size per launch: fall back to the 256-thread shapes
        if (c1 == 0 || c1 == 3) c1 = 1;
        if (c2 == 0 || c2 == 3) c2 = 1;
    }
    const int bm1 = cfg_bm(c1), bn1 = cfg_bn(c1), bk1 = cfg_bk(c1);
    p1.k_per_split = cdiv(N, bk1) * bk1; p1.splits = 1;
    p1.bias = nullptr; p1.residual = nullptr; p1.ldr = 0; p1.aux = aux; p1.ldaux = K; p1.epi = epi;
    p1.accumulate = dx_accumulate; p1.ws = nullptr; p1.colsum = db_accum;
    p1.tiles_m = cdiv(M, bm1); p1.tiles_n = cdiv(K, bn1); p1.n_per_xcd = cdiv(p1.tiles_n, 8);
    // wgrad: rows N (out-features), cols K, reduction M (tokens)
    p2.A = dy; p2.lda = N; p2.B = x; p2.ldb = K; p2.C = dw; p2.ldc = K;
    p2.M = N; p2.N = K; p2.K = M;
    const int bm2 = cfg_bm(c2), bn2 = cfg_bn(c2), bk2 = cfg_bk(c2);
    p2.k_per_split = cdiv(M, bk2) * bk2; p2.splits = 1;
    p2.bias = nullptr; p2.residual = nullptr; p2.ldr = 0; p2.aux = nullptr; p2.ldaux = 0; p2.epi = VITAE_EPI_NONE;
    p2.accumulate = dw_accumulate; p2.ws = nullptr; p2.colsum = nullptr;
    p2.tiles_m = cdiv(N, bm2); p2.tiles_n = cdiv(K, bn2); p2.n_per_xcd = cdiv(p2.tiles_n, 8);
    const int nb1 = 8 * p1.n_per_xcd * p1.tiles_m, nb2 = 8 * p2.n_per_xcd * p2.tiles_m;
    dim3 grid(nb1 + nb2);
    hipStream_t st = (hipStream_t)stream;
    if (c1 == 0) hipLaunchKernelGGL((gemm_bf16_pair_kernel<128, 32, 64, 256, 32, 64, 256>), grid, dim3(128), 0, st, p1, p2, nb1);
    else if (c1 == 3) hipLaunchKernelGGL((gemm_bf16_pair_kernel<512, 64, 64, 256, 64, 64, 256>), grid, dim3(512), 0, st, p1, p2, nb1);
    else if (c1 == 1 && c2 == 1) hipLaunchKernelGGL((gemm_bf16_pair_kernel<256, 64, 64, 256, 64, 64, 256>), grid, dim3(256), 0, st, p1, p2, nb1);
    else if (c1 == 1) hipLaunchKernelGGL((gemm_bf16_pair_kernel<256, 64, 64, 256, 64, 128, 128>), grid, dim3(256), 0, st, p1, p2, nb1);
    else if (c2 == 1) hipLaunchKernelGGL((gemm_bf16_pair_kernel<256, 64, 128, 128, 64, 64, 256>), grid, dim3(256), 0, st, p1, p2, nb1);
    else hipLaunchKernelGGL((gemm_bf16_pair_kernel<256, 64, 128, 128, 64, 128, 128>), grid, dim3(256), 0, st, p1, p2, nb1);
    return vitae_launch_status();
}

// fp32 -> bf16 shadow (engine init; afterwards the AdamW kernel keeps the shadow current)
namespace {
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        reinterpret_cast<bf16x4*>(dst)[i] = o;
    }
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = (__bf16)src[i];
}
}  // namespace

namespace {
// hi | lo planes of `count` equally spaced [rows, K] tensors, side by side: out[t][r][0:K] = bf16(x), out[t][r][K:2K] = bf16(x - float(bf16(x)))
__global__ __launch_bounds__(256) void cast_bf16_hilo_kernel(const float* __restrict__ src, __bf16* __restrict__ out, long rows, int K, long stride, int count) {
    const long k4 = K / 4, per = rows * k4, tot = per * count;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < tot; j += (long)gridDim.x * 256) {
        const long t = j / per, i = j % per, r = i / k4, c = i % k4;
        const f32x4 v = reinterpret_cast<const f32x4*>(src + t * stride)[i];
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (__bf16)v[e]; l[e] = (__bf16)(v[e] - (float)h[e]); }
        __bf16* row = out + (t * rows + r) * (2L * K);
        reinterpret_cast<bf16x4*>(row)[c] = h;
        reinterpret_cast<bf16x4*>(row + K)[c] = l;
    }
}
}  // namespace

extern "C" int vitae_cast_bf16_hilo(const float* src, void* out_bf16, long rows, int K, long stride, int count, void* stream) {
    if (!src || !out_bf16 || rows <= 0 || K <= 0 || count <= 0 || (K & 3) || (stride & 3) || ((uintptr_t)src & 15) || ((uintptr_t)out_bf16 & 7)) return VITAE_ERR_INVALID_ARG;
    long blocks = (rows * (K / 4) * count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cast_bf16_hilo_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<__bf16*>(out_bf16), rows, K, stride, count);
    return vitae_launch_status();
}

extern "C" int vitae_cast_bf16(const float* src, void* dst_bf16, long n, void* stream) {
    if (!src || !dst_bf16 || n <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst_bf16 & 7)) return VITAE_ERR_INVALID_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<__bf16*>(dst_bf16), n);
    return vitae_launch_status();
}
