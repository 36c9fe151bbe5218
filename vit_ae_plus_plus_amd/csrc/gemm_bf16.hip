// Throughput-mode GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for every dense contraction of the
// path — same contract as gemm.hip:
//   C[M,N] (+)= epi( sum_k A(m,k) * B(n,k) + bias[n] ) (+ residual)
// A is fp32 in HBM (activations / gradients) and is rounded to bf16 while being staged; B is either
// fp32 or a bf16 weight shadow kept by the AdamW kernel (halves the dominant HBM / L2 traffic).
// Operand storage flags as in gemm.hip (*_KC: element (row,k) at ptr[row*ld + k], else ptr[k*ld + row]).
//
// MI355X-specific structure
//  * tiles 64 x BN (BN = 128 or 64), BK = 64, 4 waves (2 x 2), wave tile 32 x BN/2: M is only
//    ~440-870 rows on this path, so small M tiles keep >100 workgroups in flight;
//  * LDS images keep the GLOBAL orientation of each operand (no transposing stores): k-contiguous
//    tiles are read with ds_read_b128, row-contiguous tiles (dgrad's W, both wgrad operands) with the
//    hardware transpose read ds_read_b64_tr_b16 — no transposed copies of weights or activations exist;
//    row strides are padded (+8 / +16 elements) so both read kinds are bank-conflict free;
//  * the K loop is latency bound at these sizes, so global loads run TWO k-tiles ahead of the MFMAs
//    (two register sets + double-buffered LDS, one barrier per k-tile);
//  * workgroup ids are remapped so that all M-tiles of one weight panel run on the same XCD (its L2
//    then serves the panel to the 7-14 workgroups that share it);
//  * split-K partials go to a workspace and are reduced deterministically (shared with gemm.hip).
#include "common.hpp"
#include "vitae_hip.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 64, BK = 64;

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
    int tiles_m, tiles_n, n_per_xcd;
};

__device__ __forceinline__ void epilogue_store(const GemmArgs& p, float v, int m, int n) {
    if (p.bias) v += p.bias[n];
    if (p.epi == VITAE_EPI_GELU) {
        p.aux[(long)m * p.ldaux + n] = v;
        v = gelu_erf(v);
    } else if (p.epi == VITAE_EPI_DGELU) {
        v *= gelu_erf_grad(p.aux[(long)m * p.ldaux + n]);
    } else if (p.epi == VITAE_EPI_RELU_MASK) {
        v = p.aux[(long)m * p.ldaux + n] > 0.f ? v : 0.f;
    }
    if (p.residual) v += p.residual[(long)m * p.ldr + n];
    float* c = p.C + (long)m * p.ldc + n;
    if (p.accumulate) v += *c;
    *c = v;
}

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// LDS row stride (elements) for a tile whose contiguous dimension has `n` elements
__host__ __device__ constexpr int ld_pad(int n) { return n == 128 ? n + 16 : n + 8; }

// ---------------------------------------------------------------- operand staging
// One operand tile = ROWS rows x BK k.  Per thread NPIECE 16-byte global pieces.
template <int ROWS, bool KC, bool BF16> struct Stage {
    static constexpr int EPP = BF16 ? 8 : 4;                       // elements per 16-byte piece
    static constexpr int NPIECE = ROWS * BK / EPP / 256;
    static constexpr int LD = KC ? ld_pad(BK) : ld_pad(ROWS);      // LDS row stride in bf16 elements
    static constexpr int BYTES = (KC ? ROWS : BK) * LD * 2;
    u32x4 r[NPIECE];

    __device__ __forceinline__ void load(const void* __restrict__ P, long ld, int rows, int r0, int k0, int kend) {
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const int p = threadIdx.x + 256 * j;
            int row, k;
            if (KC) { row = p / (BK / EPP); k = (p % (BK / EPP)) * EPP; }
            else { k = p / (ROWS / EPP); row = (p % (ROWS / EPP)) * EPP; }
            const int gr = r0 + row, gk = k0 + k;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gr < rows && gk < kend) {
                const long off = KC ? ((long)gr * ld + gk) : ((long)gk * ld + gr);
                v = BF16 ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(P) + off)
                         : *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(P) + off);
            }
            r[j] = v;
        }
    }

    __device__ __forceinline__ void store(__bf16* lds) const {
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const int p = threadIdx.x + 256 * j;
            int row, k;
            if (KC) { row = p / (BK / EPP); k = (p % (BK / EPP)) * EPP; }
            else { k = p / (ROWS / EPP); row = (p % (ROWS / EPP)) * EPP; }
            __bf16* dst = KC ? (lds + row * LD + k) : (lds + k * LD + row);
            if (BF16) {
                *reinterpret_cast<u32x4*>(dst) = r[j];
            } else {
                union { u32x4 u; f32x4 f; } in;
                in.u = r[j];
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)in.f[e];
                *reinterpret_cast<bf16x4*>(dst) = o;
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

template <int BN, bool A_KC, bool B_KC, bool B_BF16>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs p) {
    using SA = Stage<BM, A_KC, false>;
    using SB = Stage<BN, B_KC, B_BF16>;
    constexpr int FN = BN / 64;                         // 32-col fragments per wave (wave tile 32 x BN/2)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (SA::BYTES + SB::BYTES)];
    // XCD-aware tile order: hardware places workgroup b on XCD b % 8; give each XCD whole weight panels.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int tn = xcd + 8 * (local / p.tiles_m), tm = local % p.tiles_m;
    if (tn >= p.tiles_n) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    f32x16 acc[FN];
#pragma unroll
    for (int f = 0; f < FN; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;

    SA a0, a1;
    SB b0, b1;
    auto As = [&](int s) { return reinterpret_cast<__bf16*>(smem + s * (SA::BYTES + SB::BYTES)); };
    auto Bs = [&](int s) { return reinterpret_cast<__bf16*>(smem + s * (SA::BYTES + SB::BYTES) + SA::BYTES); };
    auto compute = [&](int s) {
        const __bf16* at = As(s);
        const __bf16* bt = Bs(s);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const bf16x8 fa = frag<A_KC, SA::LD>(at, wm * 32, kk, lane);
#pragma unroll
            for (int f = 0; f < FN; ++f) {
                const bf16x8 fb = frag<B_KC, SB::LD>(bt, wn * (BN / 2) + f * 32, kk, lane);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[f], 0, 0, 0);
            }
        }
    };

    if (nk > 0) {
        a0.load(p.A, p.lda, p.M, m0, kbeg, kend);
        b0.load(p.B, p.ldb, p.N, n0, kbeg, kend);
    }
    if (nk > 1) {
        a1.load(p.A, p.lda, p.M, m0, kbeg + BK, kend);
        b1.load(p.B, p.ldb, p.N, n0, kbeg + BK, kend);
    }
    if (nk > 0) { a0.store(As(0)); b0.store(Bs(0)); }
    __syncthreads();
    // steady state, unrolled by two so both register sets are statically indexed:
    //   tile t   lives in LDS stage t & 1, tile t+1 in registers (set (t+1) & 1), tile t+2 being issued.
    for (int t = 0; t < nk; t += 2) {
        if (t + 2 < nk) {
            a0.load(p.A, p.lda, p.M, m0, kbeg + (t + 2) * BK, kend);
            b0.load(p.B, p.ldb, p.N, n0, kbeg + (t + 2) * BK, kend);
        }
        compute(0);
        if (t + 1 < nk) { a1.store(As(1)); b1.store(Bs(1)); }
        __syncthreads();
        if (t + 1 >= nk) break;
        if (t + 3 < nk) {
            a1.load(p.A, p.lda, p.M, m0, kbeg + (t + 3) * BK, kend);
            b1.load(p.B, p.ldb, p.N, n0, kbeg + (t + 3) * BK, kend);
        }
        compute(1);
        if (t + 2 < nk) { a0.store(As(0)); b0.store(Bs(0)); }
        __syncthreads();
    }

#pragma unroll
    for (int f = 0; f < FN; ++f) {
        const int n = n0 + wn * (BN / 2) + f * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + crow(r, hi);
            if (m >= p.M) continue;
            if (p.splits > 1) p.ws[((long)blockIdx.z * p.M + m) * p.N + n] = acc[f][r];
            else epilogue_store(p, acc[f][r], m, n);
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const GemmArgs p) {
    const long total = (long)p.M * p.N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.ws[(long)s * total + i];
        epilogue_store(p, v, (int)(i / p.N), (int)(i % p.N));
    }
}

template <int BN, bool B_BF16>
void launch(const GemmArgs& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t st) {
    dim3 block(256);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<BN, true, true, B_BF16>), grid, block, 0, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<BN, true, false, B_BF16>), grid, block, 0, st, p);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<BN, false, true, B_BF16>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BN, false, false, B_BF16>), grid, block, 0, st, p);
}

}  // namespace

extern "C" int vitae_gemm_bf16_pick_split_k(int M, int N, int K) {
    const int bn = N >= 512 ? 128 : 64;
    const long tiles = (long)cdiv(M, BM) * cdiv(N, bn);
    if (tiles >= 128 || K < 1024) return 1;
    long s = (256 + tiles - 1) / tiles;
    const long max_by_k = K / 512;   // keep >= 8 k-tiles per split
    if (s > max_by_k) s = max_by_k;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
}

extern "C" int vitae_gemm_bf16(int a_kcontig, int b_kcontig, const float* A, long lda, const void* B, long ldb,
                               int b_is_bf16, float* C, long ldc, int M, int N, int K, const float* bias,
                               const float* residual, long ldr, int epi, float* aux, long ldaux, int accumulate,
                               int split_k, float* splitk_ws, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    if (epi != VITAE_EPI_NONE && !aux) return VITAE_ERR_INVALID_ARG;
    const int b_epp = b_is_bf16 ? 8 : 4;
    const int a_vec = a_kcontig ? K : M, b_vec = b_kcontig ? K : N;
    if ((a_vec & 3) || (lda & 3) || (b_vec % b_epp) || (ldb % b_epp)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (split_k < 1) split_k = 1;
    if (epi == VITAE_EPI_GELU) split_k = 1;
    GemmArgs p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    int kps = cdiv(cdiv(K, split_k), BK) * BK;
    split_k = cdiv(K, kps);
    if (split_k > 1 && !splitk_ws) return VITAE_ERR_INVALID_ARG;
    p.k_per_split = kps; p.splits = split_k;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux;
    p.epi = epi; p.accumulate = accumulate; p.ws = splitk_ws;
    const int bn = N >= 512 ? 128 : 64;
    p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(N, bn);
    p.n_per_xcd = cdiv(p.tiles_n, 8);
    dim3 grid(8 * p.n_per_xcd * p.tiles_m, 1, split_k);
    hipStream_t st = (hipStream_t)stream;
    const bool akc = a_kcontig != 0, bkc = b_kcontig != 0;
    if (bn == 128) { if (b_is_bf16) launch<128, true>(p, akc, bkc, grid, st); else launch<128, false>(p, akc, bkc, grid, st); }
    else { if (b_is_bf16) launch<64, true>(p, akc, bkc, grid, st); else launch<64, false>(p, akc, bkc, grid, st); }
    if (split_k > 1) {
        const long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(blocks), dim3(256), 0, st, p);
    }
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

extern "C" int vitae_cast_bf16(const float* src, void* dst_bf16, long n, void* stream) {
    if (!src || !dst_bf16 || n <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst_bf16 & 7)) return VITAE_ERR_INVALID_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<__bf16*>(dst_bf16), n);
    return vitae_launch_status();
}
