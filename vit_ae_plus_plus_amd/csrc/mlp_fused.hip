// The MLP of a transformer block (reference: Mlp3D.forward, model/vit.py:90-96 — fc1, exact GELU, fc2 — inside
// Block.forward, model/vit.py:143) as ONE launch forward and ONE launch backward.
//
// Why: at the batch sizes of BASELINE config 2 (M = 440 encoder / 868 decoder token rows) every kernel of the step costs
// ~8-12 us whatever it computes (launch boundary + first-tile latency + store drain), so the step time is the NUMBER of
// dependent launches.  fc1 -> GELU -> fc2 (+ the LayerNorm that follows) were three launches forward and three backward;
// here a workgroup owns (64-row panel, 128-wide slice of the hidden dimension), keeps the GELU tile in LDS, and multiplies it
// straight into its share of the fc2 product:
//   forward :  h = y2[panel] W1[slice]^T + b1 ;  act = gelu(h) ;  slab[slice][panel] = act W2[:, slice]^T
//   backward:  dact = dxo[panel] W2[:, slice] ;  dh = dact * gelu'(h) ;  slab[slice][panel] = dh W1[slice, :]
// The H/128 partial results ("slabs", fp32) are summed by the LayerNorm kernel that consumes the result anyway
// (vitae_layernorm_{fwd,bwd}_slabs in norm.hip: the launch-boundary reduce) — no atomics, no in-launch waiting.
// The weight gradients (dW2 = dxo^T act, dW1 = dh^T y2) need act / dh for ALL rows: both are saved in bf16 and the
// products run off the critical path (vitae_wgrad_group_glds).
//
// Layout choices for gfx950:
//   * every product is computed TRANSPOSED (D[hidden or n][m]): the MFMA result then holds, per lane, one token row m and
//     runs of 4 consecutive hidden / output columns — 8-byte LDS writes of the bf16 GELU tile in exactly the k-contiguous
//     image the second product reads, and 16-byte global stores of the fp32 slab, with no LDS transpose pass;
//   * all operands arrive by LDS-DMA (glds_tiles.hpp); phase 1 streams 64-deep k-tiles of the weight slice (128 rows) and
//     of the activation panel (64 rows) through NST1 stages; phase 2 streams one 64-column tile of the other weight per
//     step through a ring that aliases the phase-1 stages and is primed while the GELU epilogue runs;
//   * blocks b -> (XCD b & 7): an XCD owns the hidden slices j = xcd (mod 8) and walks all panels under them, so each
//     weight slice is fetched from memory by ONE L2 and re-used by every panel.
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "glds_tiles.hpp"
#include "vitae_hip.h"

namespace {

using namespace vglds;

constexpr int HS = 128;                 // hidden slice per workgroup
constexpr int NST1 = 4;                 // phase-1 stages (A 16 KB + B 8 KB each)
constexpr int ST1 = 128 * BK * 2 + 64 * BK * 2;
constexpr int R2 = 6;                   // phase-2 ring slots (one 64-column weight tile x 128 k = 16 KB each)
constexpr int SL2 = 2 * 64 * BK * 2;
constexpr int REGION0 = NST1 * ST1 > R2 * SL2 ? NST1 * ST1 : R2 * SL2;   // 96 KB
constexpr int IMG = 2 * 64 * BK * 2;    // [2 k-tiles][64 rows][64 k] bf16 = 16 KB
constexpr int SMEM = REGION0 + 2 * IMG + 1024;   // 129 KB: one workgroup per CU (the last KB holds the fc1 bias slice)

struct MlpArgs {
    const __bf16* X;     // fwd: y2 (LayerNorm output) [Mpad, d]; bwd: dxo = gradient of the block output [Mpad, d]
    const __bf16* W1;    // fc1.weight [H, d]
    const __bf16* W2;    // fc2.weight [d, H]
    const float* b1;     // fc1.bias [H] (fwd)
    __bf16* hpre;        // fc1 pre-activation [Mpad, H]: written fwd, read bwd
    __bf16* out16;       // fwd: act = gelu(h) [Mpad, H]; bwd: dh [Mpad, H]
    float* slabs;        // [S][Mpad][d]
    int M, Mpad, d, H, panels, S;
    long long* dbg;      // optional (tools/mlp_fused_probe.py): 8 s_memtime stamps per workgroup
};

long long* g_dbg = nullptr;

template <int N> __device__ __forceinline__ void waitv() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// address of the 8-byte half `hi` of 16-byte chunk c of row m inside a [2][64][64] k-contiguous image
__device__ __forceinline__ int img_off(int kt, int m, int c, int hi) {
    return kt * (64 * 128) + m * 128 + ((c ^ swz<true, 8>(m)) << 4) + hi * 8;
}

template <bool BWD, int NT>   // NT = d / 64
__global__ __launch_bounds__(256) void mlp_fused_kernel(const MlpArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];   // the ONLY LDS object
    unsigned char* imgX = smem + REGION0;          // act (fwd) / dh (bwd): operand of phase 2
    unsigned char* imgY = imgX + IMG;              // fc1 pre-activation tile
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int j = xcd + 8 * (local / p.panels), pn = local % p.panels;
    if (j >= p.S) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // phase 1: hidden half (64) x token half (32)
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = pn * 64, h0 = j * HS;
    const int d = p.d, H = p.H;
    auto stamp = [&](int i) {
        if (p.dbg && threadIdx.x == 0) p.dbg[(long)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- small operands first, also by DMA: an ordinary (VGPR-destination) load beside LDS-DMAs makes hipcc drain the
    //      whole DMA queue (vmcnt(0)) at its first use (cdna_hip_programming.md, "Pipelining across barriers")
    float* biasL = reinterpret_cast<float*>(imgY + IMG);
    if constexpr (!BWD) {
        if (wave == 0)      // 128 floats = 32 lanes x 16 B; the upper lanes re-load the same bytes into the spare half
            __builtin_amdgcn_global_load_lds(p.b1 + h0 + 4 * (lane & 31), (__attribute__((address_space(3))) void*)biasL, 16, 0, 0);
    } else {
        // the saved pre-activation tile [64 rows][128 hidden] -> imgY, by DMA
        dma_tile<64, true, 4>(p.hpre, H, p.Mpad, m0, h0, imgY, wave, lane);
        dma_tile<64, true, 4>(p.hpre, H, p.Mpad, m0, h0 + 64, imgY + 64 * 128, wave, lane);
    }

    // ---- phase 1: D1[hidden 128][m 64] = A(hidden, k) * B(m, k), k over d
    //      fwd: A = W1 rows h0.. (k-contiguous);  bwd: A = W2 columns h0.. (element (k = n, hidden) at W2[n * H + hidden])
    f32x16 acc[2][2];      // [even / odd 16-deep k-slice][fragment f: hidden wm * 64 + f * 32 ..]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[h][f][i] = 0.f;
    constexpr int G1 = 6;   // DMA instructions per wave per stage (4 for the 128-row tile + 2 for the 64-row tile)
    // piece i (0..5) of the stage of k-tile t.  One workgroup per CU means nobody else overlaps this workgroup's DMA issue
    // (the texture addresser takes a 1 KB piece in >= 16 clocks, 24 KB per k-tile = 384+ clocks) with its MFMAs (256 clocks
    // per k-tile): the pieces of tile t + NST1 - 1 are therefore interleaved, one by one, with the MFMAs of tile t.
    auto piece1 = [&](int t, int i) {
        unsigned char* st = smem + (t % NST1) * ST1;
        if (i < 4) {
            if constexpr (BWD) dma_piece<128, false, 4>(p.W2, H, H, h0, t * BK, st, wave, lane, i);
            else dma_piece<128, true, 4>(p.W1, d, H, h0, t * BK, st, wave, lane, i);
        } else {
            dma_piece<64, true, 4>(p.X, d, p.Mpad, m0, t * BK, st + 128 * BK * 2, wave, lane, i - 4);
        }
    };
#pragma unroll
    for (int t = 0; t < NST1 - 1; ++t)
        if (t < NT)
#pragma unroll
            for (int i = 0; i < G1; ++i) piece1(t, i);
    auto step1 = [&](int t, auto more) {          // more: std::true_type while tile t + NST1 - 1 exists
        constexpr bool MORE = decltype(more)::value;
        const unsigned char* at = smem + (t % NST1) * ST1;
        const unsigned char* bt = at + 128 * BK * 2;
        bf16x8 fa[4][2], fb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int f = 0; f < 2; ++f) fa[kk][f] = frag<128, !BWD>(at, wm * 64 + f * 32, kk, lane);
            fb[kk] = frag<64, true>(bt, wn * 32, kk, lane);
        }
        if constexpr (BWD) {            // transposing reads are inline asm (glds_tiles.hpp): order them by hand
            frags_ready();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int f = 0; f < 2; ++f) frag_tie(fa[kk][f]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                acc[kk & 1][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][f], fb[kk], acc[kk & 1][f], 0, 0, 0);
                if constexpr (MORE) {
                    if (2 * kk + f < G1) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece1(t + NST1 - 1, 2 * kk + f);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
    };
    constexpr int MAIN1 = NT - (NST1 - 1) > 0 ? NT - (NST1 - 1) : 0;
#pragma unroll 1
    for (int t = 0; t < MAIN1; ++t) {
        waitv<(NST1 - 2) * G1>();
        __builtin_amdgcn_s_barrier();
        step1(t, std::true_type{});
    }
#pragma unroll
    for (int t = MAIN1; t < NT; ++t) {
        const int younger = min(NT - 1 - t, NST1 - 2);
        if (younger >= 2) waitv<2 * G1>();
        else if (younger == 1) waitv<G1>();
        else waitv<0>();
        __builtin_amdgcn_s_barrier();
        step1(t, std::false_type{});
    }
    // every DMA has landed (the last step waited for vmcnt(0)); all waves must be done with the stages before the ring
    // of phase 2 overwrites them
    __builtin_amdgcn_s_barrier();
    stamp(1);

    // ---- phase-2 ring: tile nt = output columns nt * 64 .. of the OTHER weight, k = this workgroup's 128 hidden units
    //      fwd: W2 rows nt * 64.. (k-contiguous, k offset h0);  bwd: W1 columns nt * 64.. (element (k, n) at W1[(h0 + k) * d + n])
    constexpr int G2 = 4;
    auto piece2 = [&](int nt, int i) {             // piece i (0..3): k-tile i >> 1, instruction i & 1
        unsigned char* sl = smem + (nt % R2) * SL2 + (i >> 1) * (64 * BK * 2);
        if constexpr (BWD) dma_piece<64, false, 4>(p.W1, d, d, nt * 64, h0 + (i >> 1) * 64, sl, wave, lane, i & 1);
        else dma_piece<64, true, 4>(p.W2, H, d, nt * 64, h0 + (i >> 1) * 64, sl, wave, lane, i & 1);
    };
#pragma unroll
    for (int nt = 0; nt < R2 - 1; ++nt)
        if (nt < NT)
#pragma unroll
            for (int i = 0; i < G2; ++i) piece2(nt, i);

    // ---- epilogue 1 (under the ring's first DMAs): bias + GELU (fwd) or * GELU'(h) (bwd); lane = token row, registers =
    //      runs of 4 consecutive hidden units -> 8-byte writes into the k-contiguous images
    {
        const int m = wn * 32 + l31;
        const bool valid = m0 + m < p.M;            // pad rows: zeros (the weight-gradient products reduce over them)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 4 * f + g;            // 16-byte chunk of the 64-wide k-tile wm
                const int off = img_off(wm, m, c, hi);
                bf16x4 o, hq;
                if constexpr (!BWD) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(biasL + wm * 64 + f * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float h = acc[0][f][4 * g + e] + acc[1][f][4 * g + e] + b4[e];
                        float cdf, pdf;
                        gelu_gate(h, cdf, pdf);
                        hq[e] = (__bf16)(valid ? h : 0.f);
                        o[e] = (__bf16)(valid ? h * cdf : 0.f);
                    }
                    *reinterpret_cast<bf16x4*>(imgY + off) = hq;
                } else {
                    hq = *reinterpret_cast<const bf16x4*>(imgY + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dact = acc[0][f][4 * g + e] + acc[1][f][4 * g + e];
                        const float h = (float)hq[e];
                        float cdf, pdf;
                        gelu_gate(h, cdf, pdf);
                        o[e] = (__bf16)(valid ? dact * fmaf(h, pdf, cdf) : 0.f);
                    }
                }
                *reinterpret_cast<bf16x4*>(imgX + off) = o;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // images complete (raw barrier: the ring's DMAs stay in flight)
    stamp(2);

    // ---- images -> global, row-major, 16 bytes per lane (8 lanes cover a 128-byte row piece)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + 256 * i;       // chunk id: [kt 2][row 64][chunk 8]
        const int kt = q >> 9, row = (q >> 3) & 63, c = q & 7;
        const int off = img_off(kt, row, c, 0);
        const long go = (long)(m0 + row) * H + h0 + kt * 64 + c * 8;
        *reinterpret_cast<bf16x8*>(p.out16 + go) = *reinterpret_cast<const bf16x8*>(imgX + off);
        if constexpr (!BWD) *reinterpret_cast<bf16x8*>(p.hpre + go) = *reinterpret_cast<const bf16x8*>(imgY + off);
    }
    stamp(3);

    // ---- phase 2: D2[n 64][m 64] per tile = A(n, k) * B(m, k) with B = imgX (registers, loaded once).  Per tile a wave has
    //      8 MFMAs (256 clocks of matrix pipe), 4 DMA pieces of tile nt + R2 - 1 and the 4 slab stores of tile nt - 1 (8 KB of
    //      texture-addresser work): all three are interleaved, the stores one tile late so they never wait for their MFMAs.
    const int wm2 = wave >> 1, wn2 = wave & 1;     // n half (32) x token half (32)
    bf16x8 fx[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) fx[kk] = frag<64, true>(imgX + (kk >> 2) * (64 * 128), wn2 * 32, kk & 3, lane);
    float* slab = p.slabs + ((long)j * p.Mpad + m0 + wn2 * 32 + l31) * d + wm2 * 32 + 4 * hi;
    f32x4 pend[4];                                  // results of the previous tile, stored during this one
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        // tile nt must have landed; counted on the DMA instructions only (the slab / image stores in between are
        // younger-or-older vector-memory operations too: ignoring them waits for slightly more, never for less)
        const int younger = min(NT - 1 - nt, R2 - 2);
        switch (younger) {
            case 4: waitv<4 * G2>(); break;
            case 3: waitv<3 * G2>(); break;
            case 2: waitv<2 * G2>(); break;
            case 1: waitv<G2>(); break;
            default: waitv<0>(); break;
        }
        if (nt == 4) stamp(8);
        __builtin_amdgcn_s_barrier();
        if (nt == 4) stamp(9);
        const unsigned char* sl = smem + (nt % R2) * SL2;
        bf16x8 fw[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) fw[kk] = frag<64, !BWD>(sl + (kk >> 2) * (64 * BK * 2), wm2 * 32, kk & 3, lane);
        if constexpr (BWD) {
            frags_ready();
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) frag_tie(fw[kk]);
        }
        if (nt == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(10); }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 a0, a1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk], fx[kk], a1, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk], fx[kk], a0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk & 1) {
                if (nt + R2 - 1 < NT) piece2(nt + R2 - 1, kk >> 1);
            } else if (nt > 0) {
                *reinterpret_cast<f32x4*>(slab + (nt - 1) * 64 + 8 * (kk >> 1)) = pend[kk >> 1];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) pend[g][e] = a0[4 * g + e] + a1[4 * g + e];
        if (nt == 0) stamp(4);
        if (nt == 3) stamp(7);
        if (nt == 4) stamp(11);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(slab + (NT - 1) * 64 + 8 * g) = pend[g];
    stamp(5);
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(6); }
}

template <bool BWD>
int launch_mlp(const MlpArgs& p, hipStream_t st) {
    dim3 grid(8 * cdiv(p.S, 8) * p.panels), block(256);
    switch (p.d / 64) {
        case 8: hipLaunchKernelGGL((mlp_fused_kernel<BWD, 8>), grid, block, 0, st, p); break;
        case 12: hipLaunchKernelGGL((mlp_fused_kernel<BWD, 12>), grid, block, 0, st, p); break;
        case 16: hipLaunchKernelGGL((mlp_fused_kernel<BWD, 16>), grid, block, 0, st, p); break;
        default: return VITAE_ERR_UNSUPPORTED_SHAPE;
    }
    return vitae_launch_status();
}

int check(const MlpArgs& p) {
    if (!p.X || !p.W1 || !p.W2 || !p.hpre || !p.out16 || !p.slabs || p.M <= 0) return VITAE_ERR_INVALID_ARG;
    if (p.Mpad < p.M || (p.Mpad % 64) || (p.H % HS) || (p.d != 512 && p.d != 768 && p.d != 1024)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if ((long)p.Mpad * p.H >= (1L << 31)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    for (const void* q : {(const void*)p.X, (const void*)p.W1, (const void*)p.W2, (const void*)p.hpre, (const void*)p.out16, (const void*)p.slabs})
        if ((uintptr_t)q & 15) return VITAE_ERR_UNSUPPORTED_SHAPE;
    return VITAE_OK;
}

}  // namespace

extern "C" int vitae_mlp_fused_supported(int d, int H) {
    return (d == 512 || d == 768 || d == 1024) && H % HS == 0 && H > 0;
}

extern "C" int vitae_mlp_fused_slabs(int H) { return H / HS; }

// profiling hook: when set, every workgroup of the following launches writes 8 shader-clock stamps (phase boundaries) to buf
extern "C" int vitae_mlp_fused_set_debug(void* buf) { g_dbg = reinterpret_cast<long long*>(buf); return VITAE_OK; }

extern "C" int vitae_mlp_fused_fwd(const void* y16, const void* w1_16, const float* b1, const void* w2_16, void* hpre16,
                                   void* act16, float* slabs, int M, int Mpad, int d, int H, void* stream) {
    MlpArgs p;
    p.X = reinterpret_cast<const __bf16*>(y16); p.W1 = reinterpret_cast<const __bf16*>(w1_16);
    p.W2 = reinterpret_cast<const __bf16*>(w2_16); p.b1 = b1;
    p.hpre = reinterpret_cast<__bf16*>(hpre16); p.out16 = reinterpret_cast<__bf16*>(act16); p.slabs = slabs;
    p.M = M; p.Mpad = Mpad; p.d = d; p.H = H; p.panels = Mpad / 64; p.S = H / HS; p.dbg = g_dbg;
    if (!b1 || ((uintptr_t)b1 & 15)) return VITAE_ERR_INVALID_ARG;
    if (int rc = check(p)) return rc;
    return launch_mlp<false>(p, (hipStream_t)stream);
}

extern "C" int vitae_mlp_fused_bwd(const void* dxo16, const void* w1_16, const void* w2_16, const void* hpre16, void* dh16,
                                   float* slabs, int M, int Mpad, int d, int H, void* stream) {
    MlpArgs p;
    p.X = reinterpret_cast<const __bf16*>(dxo16); p.W1 = reinterpret_cast<const __bf16*>(w1_16);
    p.W2 = reinterpret_cast<const __bf16*>(w2_16); p.b1 = nullptr;
    p.hpre = reinterpret_cast<__bf16*>(const_cast<void*>(hpre16)); p.out16 = reinterpret_cast<__bf16*>(dh16); p.slabs = slabs;
    p.M = M; p.Mpad = Mpad; p.d = d; p.H = H; p.panels = Mpad / 64; p.S = H / HS; p.dbg = g_dbg;
    if (int rc = check(p)) return rc;
    return launch_mlp<true>(p, (hipStream_t)stream);
}
