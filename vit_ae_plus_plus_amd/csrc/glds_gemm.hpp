// What the LDS-DMA GEMM kernels share: the launch descriptor of one GEMM problem, the XCD-aware workgroup count and the
// register-level epilogue of one 32x32 accumulator fragment (gemm_glds.hip: 64-row tiles; gemm_bt.hip: the big tiles).
#pragma once
#include "common.hpp"
#include "glds_tiles.hpp"
#include "vitae_hip.h"

namespace vglds {

struct GArgs {
    const __bf16* A; long lda;
    const __bf16* B; long ldb;
    float* C; long ldc;
    __bf16* C16; long ldc16;
    int M, N, K;
    int k_per_split, splits;
    const float* bias;
    const float* residual; long ldr;
    float* aux; long ldaux;
    int epi, accumulate;
    float* ws;           // split-K: [VITAE_GLDS_TICKETS ints of tile tickets (zero between launches)][partial tiles]
    float* out_colsum;   // optional: out_colsum[n] += sum_m (epilogue result)(m, n)  (bias gradient of the NEXT Linear)
    float* a_rowsum;     // optional: a_rowsum[m] += sum_k A(m, k): in a wgrad (A = dy^T) this is colsum(dy), the bias gradient
                         // of THIS Linear, obtained with one extra MFMA against a ones operand in the tn == 0 workgroups
    int vec_epi;         // all epilogue arrays are 16-byte addressable by 4-column groups (N, the leading dimensions and the
                         // base pointers allow it): the tile goes through LDS and leaves row-major, 16 bytes per lane
    int tiles_m, tiles_n;
    double* sqacc = nullptr;     // optional: the sum of squares of the stored result (the gradient norm's share of a wgrad) is added to
                                 // sqacc[(blockIdx.x & sq_mask) * sq_stride] — ONE address (mask 0) or spread slots: same-address double
                                 // atomics retire one per ~10 ns (round 6: 576 of them were 5.8 of the 16 us of a weight-gradient launch)
    int sq_mask = 0, sq_stride = 0;
    int tile0 = 0;               // index of this problem's first tile among the tickets / partial tiles of a grouped launch
    int aux16 = 0;               // aux holds bf16 (VITAE_EPI_AUX_BF16): the saved GELU pre-activation at half the bytes
    int exact = 0;               // GELU / GELU' through erff (the fp32-grade modes) instead of the 1.5e-7 polynomial
    int auxd = 0;                // aux holds GELU'(pre-activation) (VITAE_EPI_AUX_DERIV): GELU saves it, GELU' multiplies by it
    const __bf16* B2 = nullptr;  // second plane of the B operand (lo = bf16(W - bf16(W)), same layout as B): vitae_gemm_glds_w2
    int a_kwrap = 0;             // > 0: the A operand is a_kwrap wide and k wraps (gemm_bt.hip: a_wrap) — two-plane weights side by side in B
    long long* dbg;      // optional (tools/gemm_phase_probe.py): 8 s_memtime stamps per workgroup
    int xcd_m;           // 0: an XCD owns column tiles tn = xcd (mod 8) and walks every row tile (its L2 holds 1/8 of B and all
                         // of A); 1: it owns row tiles tm = xcd (mod 8) instead — picked when A is the larger operand
};

// where this workgroup adds its share of the gradient norm
__device__ __forceinline__ double* sq_slot(const GArgs& p) { return p.sqacc + (long)((int)blockIdx.x & p.sq_mask) * p.sq_stride; }

// workgroups of one launch (per k-split) under either XCD mapping
inline int glds_blocks(const GArgs& p) {
    return p.xcd_m ? 8 * cdiv(p.tiles_m, 8) * p.tiles_n : 8 * cdiv(p.tiles_n, 8) * p.tiles_m;
}

// Epilogue of one 32x32 accumulator fragment: column n, rows mbase + crow(r, hi).  The reads the epilogue needs
// (aux / residual / old C) are issued eight rows at a time, from clamped addresses, BEFORE the dependent stores, so
// their latencies overlap — one dependent load -> store per element made the residual GEMMs 2x slower than the bare
// product, while batching all 16 rows x 3 arrays at once cost 150 extra VGPRs (one workgroup per CU for the 64x128
// tiles).  Returns the column sum of the stored values.
__device__ __forceinline__ float epilogue_frag(const GArgs& p, const float (&v)[16], int mbase, int n, int hi, float& sqsum) {
    // 32-bit element offsets from the (wave-uniform) base pointers: one VGPR per address instead of a 64-bit pair
    // per row and array (the launchers reject operands with more than 2^31 elements)
    const int nc = min(n, p.N - 1);
    const int ldaux = (int)p.ldaux, ldr = (int)p.ldr, ldc = (int)p.ldc, ldc16 = (int)p.ldc16;
    const bool need_aux = p.epi == VITAE_EPI_DGELU || p.epi == VITAE_EPI_RELU_MASK;
    const bool acc_c = p.C && p.accumulate;
    const float bias = p.bias ? p.bias[nc] : 0.f;
    const bool ncol = n < p.N;
    float csum = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float ax[8], rs[8], co[8];
        int mrow[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) mrow[q] = min(mbase + crow(8 * half + q, hi), p.M - 1);
        if (need_aux) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                ax[q] = p.aux16 ? (float)reinterpret_cast<const __bf16*>(p.aux)[mrow[q] * ldaux + nc] : p.aux[mrow[q] * ldaux + nc];
        }
        if (p.residual) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rs[q] = p.residual[mrow[q] * ldr + nc];
        }
        if (acc_c) {
#pragma unroll
            for (int q = 0; q < 8; ++q) co[q] = p.C[mrow[q] * ldc + nc];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mbase + crow(8 * half + q, hi);
            if (!(ncol && m < p.M)) continue;
            float x = v[8 * half + q] + bias;
            if (p.epi == VITAE_EPI_GELU) {
                float c, d;
                gelu_gate(x, c, d);
                const float sv = p.auxd ? fmaf(x, d, c) : x;
                if (p.aux16) reinterpret_cast<__bf16*>(p.aux)[m * ldaux + n] = (__bf16)sv;
                else p.aux[m * ldaux + n] = sv;
                x *= c;
            } else if (p.epi == VITAE_EPI_DGELU) {
                x *= p.auxd ? ax[q] : gelu_fast_grad(ax[q]);
            } else if (p.epi == VITAE_EPI_RELU_MASK) {
                x = ax[q] > 0.f ? x : 0.f;
            } else if (p.epi == VITAE_EPI_RELU) {
                x = fmaxf(x, 0.f);
            }
            if (p.residual) x += rs[q];
            if (p.C) {
                if (acc_c) x += co[q];
                p.C[m * ldc + n] = x;
            }
            if (p.C16) p.C16[m * ldc16 + n] = (__bf16)x;
            csum += x;
            sqsum += x * x;
        }
        asm volatile("" ::: "memory");   // keep the next batch's loads behind these stores (register pressure)
    }
    return csum;
}

// gemm_bt.hip: the big-tile kernels (id 0: 256x256 on 8 waves; 3: 128x128 on 4 waves; wave-specialised: 4: 128x128, 5: 64x64).  `p` is a complete
// descriptor of ONE unsplit problem; tiles_m / tiles_n are set by the launcher.
bool bt_tile_dims(int id, int& bm, int& bn);
int bt_launch(GArgs p, int a_kc, int b_kc, int id, hipStream_t st);
int bt_wgrad_group_launch(GArgs* ps, int n, int splits, hipStream_t st, int kind);
// gemm_bt.hip: input gradient + weight gradient of one Linear as ONE launch of wave-specialised 64 x 64 workgroups (tile id 5)
int ws64_pair_launch(GArgs p1, GArgs p2, hipStream_t st);
// ... the same as persistent workgroups walking a tile queue with the tiles pipelined across each other (round 6)
int ws64q_pair_launch(GArgs p1, GArgs p2, hipStream_t st);
// gemm_bt.hip: fp32 operands split into bf16 hi + lo by the producer waves of a wave-specialised 64 x 64 workgroup (fp32x3 mode);
// p.A / p.B point at FLOATS here, p.K any multiple of 4
int wsx3_launch(GArgs p, int a_kc, int b_kc, hipStream_t st);
// gemm_bt.hip: forward form with a two-plane (hi + lo) weight operand on the wave-specialised 64 x 64 workgroup
int ws64_w2_launch(GArgs p, hipStream_t st);
int wsx3_slots();

}  // namespace vglds
