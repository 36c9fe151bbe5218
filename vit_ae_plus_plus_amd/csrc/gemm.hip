// MFMA GEMM for the dense contractions of the ViT-MAE path (reference ops K1/K6/K8/K9/K10/K13/K23,
// SURVEY §2.3): every nn.Linear / the Conv3d-as-GEMM patch embedding, forward, dgrad and wgrad.
//
//   C[M,N] (+)= epi( sum_k A(m,k) * B(n,k) + bias[n] ) (+ residual)
//
// All tensors are fp32 in HBM.  PREC selects the matrix-core arithmetic:
//   PREC 0: v_mfma_f32_32x32x2_f32   (exact fp32 FMA chain; the parity mode, 157 TF/s peak)
//   PREC 1: v_mfma_f32_32x32x16_bf16 (operands rounded to bf16 while being staged into LDS,
//                                     fp32 accumulate; the throughput mode, 2.5 PF/s peak)
// Operand storage: *_KC = true  -> element (row,k) at ptr[row*ld + k]   ("k-contiguous")
//                  *_KC = false -> element (row,k) at ptr[k*ld + row]   ("row-contiguous")
// so forward (x[M,K], W[N,K]) is KC/KC, dgrad (dy[M,N], W[N,K] read as B(k_out, n)) is KC/!KC and
// wgrad (dy[T,N] as A(n,t), x[T,K] as B(k,t)) is !KC/!KC — no transposed copies are ever made.
//
// Block = 256 threads = 4 waves (2x2), each wave one 32x32 accumulator tile; block tile 64x64,
// BK = 32, register-staged double-buffered LDS (one barrier per k-tile).  fp32 tiles are stored
// k-major in LDS (stride 68 floats) so the one-float-per-lane MFMA operands are read as 32
// consecutive dwords (conflict-free); bf16 tiles row-major (stride 40 bf16) read as ds_read_b128.
// Optional split-K (grid.z) writes partial tiles to a workspace that `splitk_reduce_kernel`
// sums deterministically before applying the epilogue.
#include "common.hpp"
#include "vitae_hip.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32;   // (BK = 64 — 32 MFMAs per barrier, 70 KB of LDS — measured SLOWER in the step: fp32 mode 11.2 -> 12.1 ms)
constexpr int NPC = BK / 16;               // 16-byte pieces per thread and operand tile
constexpr int SM32 = 68;   // fp32 LDS row stride (floats), k-major tile [BK][SM32]
constexpr int SK16 = BK + 8;   // bf16 LDS row stride (elements), row-major tile [64][SK16]

struct GemmArgs {
    const float* A; long lda;
    const float* B; long ldb;
    float* C; long ldc;
    int M, N, K;
    int k_per_split, splits;
    const float* bias;
    const float* residual; long ldr;
    float* aux; long ldaux;
    int epi;          // VITAE_EPI_*
    int accumulate;   // C += result
    float* ws;        // split-K partials [splits][M][N]
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

template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int rows, int r0,
                                          int k0, int kend, f32x4 (&v)[NPC]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int i = tid + 256 * j;
        int row, k;
        if (KC) { row = i / (BK / 4); k = (i % (BK / 4)) * 4; } else { k = i >> 4; row = (i & 15) * 4; }
        const int gr = r0 + row, gk = k0 + k;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (gr < rows && gk < kend) {
            const float* src = KC ? (P + (long)gr * ld + gk) : (P + (long)gk * ld + gr);
            x = *reinterpret_cast<const f32x4*>(src);
        }
        v[j] = x;
    }
}

template <int PREC, bool KC>
__device__ __forceinline__ void store_tile(void* lds, const f32x4 (&v)[NPC]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int i = tid + 256 * j;
        int row, k;
        if (KC) { row = i / (BK / 4); k = (i % (BK / 4)) * 4; } else { k = i >> 4; row = (i & 15) * 4; }
        if (PREC == 0) {
            float* s = reinterpret_cast<float*>(lds);
            if (KC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[(k + e) * SM32 + row] = v[j][e];
            } else {
                *reinterpret_cast<f32x4*>(&s[k * SM32 + row]) = v[j];
            }
        } else {
            __bf16* s = reinterpret_cast<__bf16*>(lds);
            if (KC) {
                bf16x4 b;
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (__bf16)v[j][e];
                *reinterpret_cast<bf16x4*>(&s[row * SK16 + k]) = b;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[(row + e) * SK16 + k] = (__bf16)v[j][e];
            }
        }
    }
}

template <int PREC> struct TileBytes { static constexpr int value = (PREC == 0) ? BK * SM32 * 4 : 64 * SK16 * 2; };

template <int PREC, bool A_KC, bool B_KC>
// (leading scalars: what the first tile loads need, preloaded into SGPRs with the dispatch — build.py; the descriptor's own scalar
// loads are waited for by the epilogue only)
__global__ __launch_bounds__(256) void gemm_kernel(const float* hA, const float* hB, int hlda, int hldb, int hM, int hN, int hK, int hkps,
                                                   const GemmArgs p) {
    constexpr int TB = TileBytes<PREC>::value;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TB];   // [buf][A|B]
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * hkps;
    const int kend = min(hK, kbeg + hkps);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    f32x4 ra[NPC], rb[NPC];
    if (nk > 0) {
        load_tile<A_KC>(hA, hlda, hM, m0, kbeg, kend, ra);
        load_tile<B_KC>(hB, hldb, hN, n0, kbeg, kend, rb);
        store_tile<PREC, A_KC>(smem, ra);
        store_tile<PREC, B_KC>(smem + TB, rb);
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const bool more = (t + 1 < nk);
        if (more) {
            load_tile<A_KC>(hA, hlda, hM, m0, kbeg + (t + 1) * BK, kend, ra);
            load_tile<B_KC>(hB, hldb, hN, n0, kbeg + (t + 1) * BK, kend, rb);
        }
        unsigned char* cur = smem + (t & 1) * 2 * TB;
        if (PREC == 0) {
            const float* as = reinterpret_cast<const float*>(cur) + wm * 32 + l31;
            const float* bs = reinterpret_cast<const float*>(cur + TB) + wn * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float a = as[(kk * 2 + hi) * SM32];
                const float b = bs[(kk * 2 + hi) * SM32];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        } else {
            const __bf16* as = reinterpret_cast<const __bf16*>(cur) + (wm * 32 + l31) * SK16 + hi * 8;
            const __bf16* bs = reinterpret_cast<const __bf16*>(cur + TB) + (wn * 32 + l31) * SK16 + hi * 8;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(as + kk * 16);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(bs + kk * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
        }
        if (more) {
            unsigned char* nxt = smem + ((t + 1) & 1) * 2 * TB;
            store_tile<PREC, A_KC>(nxt, ra);
            store_tile<PREC, B_KC>(nxt + TB, rb);
        }
        __syncthreads();
    }

    const int n = n0 + wn * 32 + l31;
    if (n >= p.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= p.M) continue;
        if (p.splits > 1) p.ws[((long)blockIdx.z * p.M + m) * p.N + n] = acc[r];
        else epilogue_store(p, acc[r], m, n);
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const long total = (long)p.M * p.N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.ws[(long)s * total + i];
        epilogue_store(p, v, (int)(i / p.N), (int)(i % p.N));
    }
}

template <int PREC>
void launch_gemm(const GemmArgs& p, bool a_kc, bool b_kc, hipStream_t st) {
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), p.splits), block(256);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<PREC, true, true>), grid, block, 0, st, p.A, p.B, (int)p.lda, (int)p.ldb, p.M, p.N, p.K, p.k_per_split, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<PREC, true, false>), grid, block, 0, st, p.A, p.B, (int)p.lda, (int)p.ldb, p.M, p.N, p.K, p.k_per_split, p);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<PREC, false, true>), grid, block, 0, st, p.A, p.B, (int)p.lda, (int)p.ldb, p.M, p.N, p.K, p.k_per_split, p);
    else hipLaunchKernelGGL((gemm_kernel<PREC, false, false>), grid, block, 0, st, p.A, p.B, (int)p.lda, (int)p.ldb, p.M, p.N, p.K, p.k_per_split, p);
}

}  // namespace

extern "C" long vitae_gemm_workspace_floats(int M, int N, int K, int split_k) {
    return split_k > 1 ? (long)split_k * M * N : 0;
}

extern "C" int vitae_gemm_pick_split_k(int M, int N, int K) {
    // Fill the 256 CUs (aim for >= ~2 blocks per CU) when the output has few 64x64 tiles and K is long.
    const long tiles = (long)cdiv(M, BM) * cdiv(N, BN);
    if (tiles >= 256 || K < 512) return 1;
    long s = (512 + tiles - 1) / tiles;
    const long max_by_k = K / 256;   // keep >= 8 k-tiles per split
    if (s > max_by_k) s = max_by_k;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
}

extern "C" int vitae_gemm(int prec, int a_kcontig, int b_kcontig,
                          const float* A, long lda, const float* B, long ldb,
                          float* C, long ldc, int M, int N, int K,
                          const float* bias, const float* residual, long ldr,
                          int epi, float* aux, long ldaux, int accumulate,
                          int split_k, float* splitk_ws, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    if (prec == VITAE_PREC_BF16X3)   // split-operand bf16 MFMA (csrc/gemm_bf16.hip)
        return vitae_gemm_bf16x3(a_kcontig, b_kcontig, A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, epi, aux, ldaux,
                                 accumulate, split_k, splitk_ws, stream);
    if (prec != 0 && prec != 1) return VITAE_ERR_INVALID_ARG;
    if (epi != VITAE_EPI_NONE && !aux) return VITAE_ERR_INVALID_ARG;
    if (epi & VITAE_EPI_AUX_BF16) return VITAE_ERR_UNSUPPORTED_SHAPE;      // (fp32 aux only in this family)
    if ((epi & VITAE_EPI_AUX_DERIV) && (epi & 15) != VITAE_EPI_GELU && (epi & 15) != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;   // (as vitae_gemm_glds does)
    // 16-byte vector loads run along the contiguous dimension of each operand.
    const int a_vec = a_kcontig ? K : M, b_vec = b_kcontig ? K : N;
    if ((a_vec & 3) || (b_vec & 3) || (lda & 3) || (ldb & 3)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (split_k < 1) split_k = 1;
    if ((epi & 15) == VITAE_EPI_GELU) split_k = 1;   // non-linear epilogue needs the full sum anyway (done in reduce) — keep simple
    GemmArgs p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    int kps = cdiv(cdiv(K, split_k), BK) * BK;
    split_k = cdiv(K, kps);
    if (split_k > 1 && !splitk_ws) return VITAE_ERR_INVALID_ARG;
    p.k_per_split = kps; p.splits = split_k;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux;
    p.epi = epi; p.accumulate = accumulate; p.ws = splitk_ws;
    hipStream_t st = (hipStream_t)stream;
    if (prec == 0) launch_gemm<0>(p, a_kcontig != 0, b_kcontig != 0, st);
    else launch_gemm<1>(p, a_kcontig != 0, b_kcontig != 0, st);
    if (split_k > 1) {
        const long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    }
    return vitae_launch_status();
}

// ---- the reference-facing linear entry points (nn.Linear, model/vit.py:85-96,107-122) -----------
extern "C" int vitae_linear_fwd(int prec, const float* x, const float* w, const float* bias, float* y,
                                int M, int N, int K, int epi, float* aux,
                                const float* residual, int split_k, float* ws, void* stream) {
    return vitae_gemm(prec, 1, 1, x, K, w, K, y, N, M, N, K, bias, residual, N, epi, aux, N, 0,
                      split_k, ws, stream);
}

extern "C" int vitae_linear_bwd_input(int prec, const float* dy, const float* w, float* dx,
                                      int M, int N, int K, int epi, float* aux, int accumulate,
                                      int split_k, float* ws, void* stream) {
    // dx[M,K] = dy[M,N] @ W[N,K]  (reduction over N; W read row-contiguous as B(k, n))
    return vitae_gemm(prec, 1, 0, dy, N, w, K, dx, K, M, K, N, nullptr, nullptr, 0, epi, aux, K,
                      accumulate, split_k, ws, stream);
}

extern "C" int vitae_linear_bwd_weight(int prec, const float* dy, const float* x, float* dw,
                                       int M, int N, int K, int accumulate, int split_k, float* ws,
                                       void* stream) {
    // dW[N,K] = dy[M,N]^T @ x[M,K]  (reduction over the M tokens)
    return vitae_gemm(prec, 0, 0, dy, N, x, K, dw, K, N, K, M, nullptr, nullptr, 0, VITAE_EPI_NONE,
                      nullptr, 0, accumulate, split_k, ws, stream);
}
