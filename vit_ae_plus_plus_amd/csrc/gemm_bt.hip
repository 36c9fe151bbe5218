// Big-tile bf16 GEMM (128x128 ... 256x256 per workgroup) for the launches that have enough rows to fill the chip with such
// tiles: decoder_pred at every batch size, every Linear of the step from batch 8-16 up, the patch-8 model.
//   C[M,N] (+)= epi( sum_k A(m,k) * B(n,k) + bias[n] ) (+ residual)      (operand storage flags as in gemm_glds.hip)
//
// Why a second kernel family: a 64x64 tile moves 16 KB of operands L2 -> LDS per 64-deep k-step for 64x64x64 MACs, and a CU
// takes at most ~46-56 B/clk from L2 (profiles/round2_probe_dma_pattern.txt; the L2's 34.5 TB/s over 256 CUs): ~330 clocks
// of feed for 128 clocks of MFMA.  A 256x256 tile moves 64 KB for 16x the MACs (1300 clocks of feed under 2048 of MFMA).
//
// Schedule (the 8-phase structure of cdna_hip_programming.md §5 "256^2 template", re-derived for this operand layout):
//   * a k-tile (64 deep) of each operand is staged as TWO half-tiles (A-lo | A-hi, B-lo | B-hi: HM = BM/2, HN = BN/2 rows
//     each), and every wave owns rows of BOTH halves: its (BM/WM) x (BN/WN) outputs are four quadrants (A-half x B-half);
//   * a k-tile is four phases, one quadrant each: P1 reads A-lo + B-lo fragments, P2 B-hi, P3 A-hi, P4 nothing (B-lo kept):
//     (A0,B0) (A0,B1) (A1,B1) (A1,B0) — at most A-half + 2 B-half fragments live (64 VGPRs at 256x256);
//   * two LDS buffers only, yet ~5 phases of prefetch distance: a half-tile's slot is free two phases after its last
//     fragment read, so the DMA of k-tile t + 2 starts in P3 of k-tile t, half-tile by half-tile (A-lo, B-lo, then B-hi and
//     A-hi in P1 / P2 of k-tile t + 1), always 4 half-tiles (64 KB at 256x256) in flight, retired by COUNTED vmcnt waits;
//   * 8-wave workgroups run as two groups of four waves (one per SIMD each) staggered by one barrier: while one group
//     issues its 8 MFMAs of a quadrant, the other issues DMA + fragment reads, so a SIMD's matrix pipe always has one of
//     its two waves in an MFMA segment.  4-wave workgroups (128x128) are not staggered: two of them share a CU.
// Hazards, by barrier count (slot = interval between two workgroup barriers; a phase = load slot + MFMA slot; group 1 runs one
// slot behind group 0): a half-tile needed by the reads of phase g + 1 is waited for (each wave: its own DMA pieces) in the
// load slot of phase g, i.e. >= 1 barrier before anyone reads it; a half-tile is re-staged >= 2 phases after its last read,
// i.e. >= 2 barriers after the lgkmcnt(0) that retired the slower group's reads.
#include <cstdlib>
#include <type_traits>
#include "glds_gemm.hpp"

namespace vglds {

// Two-plane weights on every tile (round 5): B = [W hi | W lo] side by side ([N, 2 Ka]) and the reduction runs over 2 Ka with the A
// operand (k-contiguous, Ka wide) WRAPPING — k-tile t of the second half re-reads the activations of k-tile t - Ka / 64:
// y = x16 Whi^T + x16 Wlo^T out of one unmodified k-loop.  p.a_kwrap = Ka (a multiple of 64; 0 = no wrap).
__device__ __forceinline__ int a_wrap(const GArgs& p, int k0) { return (p.a_kwrap && k0 >= p.a_kwrap) ? k0 - p.a_kwrap : k0; }

template <int BM, int BN, int WM, int WN> struct BtCfg {
    static constexpr int NW = WM * WN, NT = 64 * NW;
    static constexpr int HM = BM / 2, HN = BN / 2;                          // rows of a half-tile
    static constexpr int FM = HM / (32 * WM), FN = HN / (32 * WN);          // 32x32 fragments of a wave inside one half
    static constexpr int A_HALF = HM * BK * 2, B_HALF = HN * BK * 2;        // bytes
    static constexpr int BUF = 2 * A_HALF + 2 * B_HALF, SMEM = 2 * BUF;
    static_assert(FM >= 1 && FN >= 1 && HM == 32 * WM * FM && HN == 32 * WN * FN, "wave arrangement does not tile the halves");
};

// Epilogue of ONE wave's part of a quadrant ((32 FM) x (32 FN) outputs), WAVE-PRIVATE: the wave parks its fragments in its own LDS
// region (Tw, row stride 32 FN floats: conflict-free for the ds_write_b32 of the MFMA layout — 32 consecutive columns per half
// wave — and for the ds_read_b128 lane groups of the row-major read-back), and re-reads them row-major: 8 FN lanes own one row
// (four consecutive columns each), so C, the bf16 copy, aux, residual and the old C move 16 bytes per lane in whole 128-byte
// (FN = 1) lines.  No workgroup barrier anywhere: a wave's LDS operations execute in order, and nobody else touches Tw.
// History (tools/bt_phase_probe.py, 256x256 tile, clocks per quadrant): workgroup-wide staging with every epilogue kind behind
// run-time branches: park 1900 + rows 5800 (5100 with the global stores removed: instruction-latency bound); specialised on the
// epilogue kind / full tiles: 1900 + 3200; wave-private: see DESIGN.md.
// KIND: which epilogue (one instantiation each, so that NO load in the row loop sits behind a run-time condition: hipcc waits
// vmcnt(0) in front of every conditionally executed load, and on gfx9 that also waits for every global store issued before —
// the first version spent ~4000 clocks per quadrant waiting for its own stores):
//   0 plain (bias), 1 + residual, 2 + old C (accumulate), 3 GELU (aux <- pre-activation), 4 GELU' (aux read), 5 ReLU mask (aux
//   read), 6 ReLU, 7 anything else (every option behind run-time checks: correct, slow).
// All global loads of the wave's part (one 16-byte load per row pass) are issued BEFORE the first store.
// AUX16 (kinds 3 / 4 / 5 / 7): the aux array holds bf16.  A template parameter, not a run-time test: with `if (p.aux16)` around the
// aux loads hipcc put every one of them behind a branch and an `s_waitcnt vmcnt(0)` — eight serialised memory round trips per
// quadrant in the GELU' epilogue (round 4, ISA of gemm_bt_kernel<256, 256, 2, 4, true, false>).
template <int FM, int FN, int KIND, bool FULL, bool AUX16>
__device__ __forceinline__ void bt_wave_rows(const GArgs& p, int mb, int nb, const float* Tw, int lane, float& sqs, f32x4& csum, const f32x4 bias4) {
    constexpr int S = 32 * FN, LPR = 8 * FN, RPI = 64 / LPR, PASSES = 32 * FM / RPI, PB = PASSES >= 4 ? 4 : PASSES;
    const int cg = lane % LPR, rr = lane / LPR;
    const int n = nb + 4 * cg;
    const bool ncol = FULL || n < p.N;
    const int nc = ncol ? n : 0;
    const int ldaux = (int)p.ldaux, ldr = (int)p.ldr, ldc = (int)p.ldc, ldc16 = (int)p.ldc16;
    const int epi = KIND == 7 ? p.epi : KIND == 3 ? VITAE_EPI_GELU : KIND == 4 ? VITAE_EPI_DGELU : KIND == 5 ? VITAE_EPI_RELU_MASK
                  : KIND == 6 ? VITAE_EPI_RELU : VITAE_EPI_NONE;
    const bool need_aux = KIND == 7 ? (p.epi == VITAE_EPI_DGELU || p.epi == VITAE_EPI_RELU_MASK) : (KIND == 4 || KIND == 5);
    const bool has_res = KIND == 7 ? p.residual != nullptr : KIND == 1;
    const bool acc_c = KIND == 7 ? (p.C && p.accumulate) : KIND == 2;
    // one array per operand kind; an instantiation other than 7 uses at most one of them (and loads all its passes up front;
    // kind 7 loads batch by batch: three arrays of all passes would spill next to 128 live accumulators)
    constexpr int LDN = KIND == 7 ? PB : PASSES;
    f32x4 ax[AUX16 ? 1 : LDN], rs[LDN], co[LDN];
    bf16x4 ax16[AUX16 ? LDN : 1];                  // raw: converted where it is used, so that the loads of a batch fly together
    auto load_ops = [&](int q0) {
#pragma unroll
        for (int q = 0; q < LDN; ++q) {
            const int mr = mb + rr + (q0 + q) * RPI;
            const int mc = FULL ? mr : min(mr, p.M - 1);
            if (need_aux) {
                if constexpr (AUX16) ax16[q] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.aux) + mc * ldaux + nc);
                else ax[q] = *reinterpret_cast<const f32x4*>(p.aux + mc * ldaux + nc);
            }
            if (has_res) rs[q] = *reinterpret_cast<const f32x4*>(p.residual + mc * ldr + nc);
            if (acc_c) co[q] = *reinterpret_cast<const f32x4*>(p.C + mc * ldc + nc);
        }
    };
    if constexpr (KIND != 7) load_ops(0);
    const float* Trow = Tw + rr * S + 4 * cg;
    // LDS reads ahead of the first store: the whole part when it is four passes, else batch by batch (eight passes of x next to
    // the 128 live accumulators of the 256x256 tile spill)
    constexpr bool XALL = PASSES <= 4;
    f32x4 x[XALL ? PASSES : PB];
    if constexpr (XALL) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) x[q] = *reinterpret_cast<const f32x4*>(Trow + q * RPI * S);
    }
#pragma unroll
    for (int pb = 0; pb < PASSES; pb += PB) {
        if constexpr (KIND == 7) load_ops(pb);
        constexpr int LB = KIND == 7 ? 0 : 1;      // index of this batch's first pass in the operand arrays: pb * LB
        constexpr int XB = XALL ? 1 : 0;
        if constexpr (!XALL) {
#pragma unroll
            for (int q = 0; q < PB; ++q) x[q] = *reinterpret_cast<const f32x4*>(Trow + (pb + q) * RPI * S);
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int m = mb + rr + (pb + q) * RPI;
            const bool ok = FULL || (ncol && m < p.M);
            f32x4 v = x[pb * XB + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bias4[e];
            f32x4 axv = {0.f, 0.f, 0.f, 0.f};
            if (need_aux) {
                if constexpr (AUX16) {
                    const bf16x4 h = ax16[pb * LB + q];
                    axv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                } else {
                    axv = ax[pb * LB + q];
                }
            }
            if (epi == VITAE_EPI_GELU) {
                f32x4 y, dy;
                if (p.exact) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float ye, de; gelu_erf_both(v[e], ye, de); y[e] = ye; dy[e] = de; }
                } else {
                    gelu_fast4(v, y, dy);
                }
                const f32x4 sv = p.auxd ? dy : v;                // what the backward gets: GELU'(x) (VITAE_EPI_AUX_DERIV) or x itself
                if (ok) {
                    if constexpr (AUX16) {
                        bf16x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (__bf16)sv[e];
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.aux) + m * ldaux + n) = h;
                    } else {
                        *reinterpret_cast<f32x4*>(p.aux + m * ldaux + n) = sv;
                    }
                }
                v = y;
            } else if (epi == VITAE_EPI_DGELU) {
                f32x4 g = axv;                                   // (aux already holds the derivative: a multiply)
                if (!p.auxd) {
                    if (p.exact) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = gelu_erf_grad(axv[e]);
                    } else {
                        g = gelu_fast_grad4(axv);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= g[e];
            } else if (epi == VITAE_EPI_RELU_MASK) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = axv[e] > 0.f ? v[e] : 0.f;
            } else if (epi == VITAE_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rs[pb * LB + q][e];
            }
            if (acc_c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += co[pb * LB + q][e];
            }
            if (ok) {
#ifndef VITAE_BT_STORE
#define VITAE_BT_STORE 0          // experiments: 1 = no fp32 store (the value is kept alive), 2 = non-temporal store
#endif
                if (p.C) {
                    if (VITAE_BT_STORE == 1) asm volatile("" ::"v"(v));
                    else if (VITAE_BT_STORE == 2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.C + m * ldc + n));
                    else *reinterpret_cast<f32x4*>(p.C + m * ldc + n) = v;
                }
                if (p.C16) {
                    bf16x4 v16;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v16[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>(p.C16 + m * ldc16 + n) = v16;
                }
                if (p.out_colsum) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[e] += v[e];
                }
                if (p.sqacc) sqs += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
    }
}

__device__ __forceinline__ int bt_epilogue_kind(const GArgs& p) {
    const bool res = p.residual != nullptr, accc = p.C && p.accumulate;
    if (p.epi == VITAE_EPI_NONE) return res && accc ? 7 : res ? 1 : accc ? 2 : 0;
    if (res || accc) return 7;
    return p.epi == VITAE_EPI_GELU ? 3 : p.epi == VITAE_EPI_DGELU ? 4 : p.epi == VITAE_EPI_RELU_MASK ? 5 : p.epi == VITAE_EPI_RELU ? 6 : 7;
}

// csum: the column sums of this part are ADDED to it (lane layout: the four columns nb + 4 (lane % (8 FN)) ..., one partial per row
// group).  defer_colsum: nothing is flushed — the caller folds the parts of a whole workgroup tile and issues ONE atomic per column
// (bt_tail); otherwise the wave folds its own row groups and adds its sums to p.out_colsum itself (csum should come in as zero).
template <int FM, int FN>
__device__ __forceinline__ void bt_wave_epilogue(const GArgs& p, int kind, int mb, int nb, const float* Tw, int lane, float& sqs, const f32x4 bias4,
                                                 f32x4& csum, const bool defer_colsum) {
    constexpr int LPR = 8 * FN;
    const bool full = mb + 32 * FM <= p.M && nb + 32 * FN <= p.N;
#define VITAE_BT_ROWS(E)                                                                       \
    case E:                                                                                    \
        if (full) bt_wave_rows<FM, FN, E, true, false>(p, mb, nb, Tw, lane, sqs, csum, bias4);  \
        else bt_wave_rows<FM, FN, E, false, false>(p, mb, nb, Tw, lane, sqs, csum, bias4);      \
        break;
#define VITAE_BT_ROWS_AUX(E)                                                                   \
    case E:                                                                                    \
        if (p.aux16) {                                                                         \
            if (full) bt_wave_rows<FM, FN, E, true, true>(p, mb, nb, Tw, lane, sqs, csum, bias4);  \
            else bt_wave_rows<FM, FN, E, false, true>(p, mb, nb, Tw, lane, sqs, csum, bias4);      \
        } else {                                                                               \
            if (full) bt_wave_rows<FM, FN, E, true, false>(p, mb, nb, Tw, lane, sqs, csum, bias4); \
            else bt_wave_rows<FM, FN, E, false, false>(p, mb, nb, Tw, lane, sqs, csum, bias4);     \
        }                                                                                      \
        break;
    switch (kind) {
        VITAE_BT_ROWS(0) VITAE_BT_ROWS(1) VITAE_BT_ROWS(2) VITAE_BT_ROWS_AUX(3) VITAE_BT_ROWS_AUX(4) VITAE_BT_ROWS_AUX(5) VITAE_BT_ROWS(6)
        default:
            if (p.aux16) bt_wave_rows<FM, FN, 7, false, true>(p, mb, nb, Tw, lane, sqs, csum, bias4);
            else bt_wave_rows<FM, FN, 7, false, false>(p, mb, nb, Tw, lane, sqs, csum, bias4);
            break;
    }
#undef VITAE_BT_ROWS
#undef VITAE_BT_ROWS_AUX
    if (defer_colsum) return;                      // (the caller folds csum over its parts: by reference, never through a pointer — that put it in scratch)
    if (p.out_colsum) {
        // lanes with equal (lane % LPR) hold the same four columns: fold the row groups, then one atomic per column
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int d = LPR; d < 64; d <<= 1) csum[e] += __shfl_xor(csum[e], d, 64);
        const int n = nb + 4 * (lane % LPR);
        if (lane < LPR && n < p.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(p.out_colsum + n + e, csum[e]);
        }
    }
}

// the fragments of one quadrant -> the wave's own LDS region
template <int FM, int FN, int NACC>
__device__ __forceinline__ void bt_park_quadrant(const f32x16 (&acc)[NACC][FM * FN], int lane, float* Tw) {
    constexpr int S = 32 * FN;
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[0][fm * FN + fn][r];
                if (NACC == 2) v += acc[NACC - 1][fm * FN + fn][r];
                Tw[(fm * 32 + crow(r, hi)) * S + fn * 32 + l31] = v;
            }
}

// What follows the k-loop of a big-tile workgroup: the in-launch split-K fix-up (tiles below 256 x 256) and the wave-private
// epilogue, quadrant by quadrant.  `wave` / threadIdx.x index the NW = WM * WN waves that hold accumulators (the wave-specialised
// kernel's producer waves have left by now: a barrier only counts the waves still alive).
template <int BM, int BN, int WM, int WN, int NACC, class Stamp>
__device__ __forceinline__ void bt_tail(const GArgs& p, f32x16 (&acc)[2][2][NACC][BtCfg<BM, BN, WM, WN>::FM * BtCfg<BM, BN, WM, WN>::FN],
                                        unsigned char* smem, const int m0, const int n0, const int tm, const int tn, const int zid,
                                        const int wave, const int lane, Stamp stamp) {
    using Cf = BtCfg<BM, BN, WM, WN>;
    constexpr int NW = Cf::NW, HM = Cf::HM, HN = Cf::HN, FM = Cf::FM, FN = Cf::FN, NF = FM * FN;
    const int wm = wave / WN, wn = wave % WN;
    // epilogue: every wave on its own, quadrant by quadrant through its own 32 FM x 32 FN floats of LDS (the stages are free once
    // everybody has left the k-loop).  The loop over quadrants is ROLLED; only the register -> LDS part depends on which
    // accumulators are meant.
    constexpr int TW = 32 * FM * 32 * FN;          // floats per wave
    static_assert(NW * TW * 4 + 64 <= Cf::SMEM, "wave-private staging (+ the norm share of each wave) fits the operand stages");
    float* Tw = reinterpret_cast<float*>(smem) + wave * TW;
    float sqs = 0.f;
    const int kind = bt_epilogue_kind(p);
    if constexpr (BM * BN < 256 * 256) if (p.splits > 1) {     // (the 256x256 tile: 256 KB of partials per split and workgroup - not offered)
        // Split-K inside the launch: every split parks its partial tile in the workspace (fragment order: 16 bytes per lane,
        // lane-contiguous, WRITE-THROUGH stores — sc1 — so the data is on the memory side of the eight L2s without a release
        // fence), drains, takes a ticket; the LAST arriver of a tile re-reads all partials with sc1 loads in split order
        // (bitwise reproducible whatever the arrival order) and runs the epilogue.  cdna_hip_programming.md §6 Guideline 16 R1.
        constexpr int NT = 64 * NW, GROUPS = 4 * NF * 4;                  // 16-byte groups per lane: quadrants x fragments x 4
        const int tile = p.tile0 + tm * p.tiles_n + tn;
        float* part = p.ws + VITAE_GLDS_TICKETS + (long)tile * p.splits * (NT * GROUPS * 4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, p.splits * (NT * GROUPS * 16), 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int toff = (int)threadIdx.x * 16;
        if constexpr (NACC == 2) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[a][b][0][f][i] += acc[a][b][NACC - 1][f][i];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][0][f][4 * g + e];
                        const int grp = ((a * 2 + b) * NF + f) * 4 + g;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (zid * GROUPS + grp) * (NT * 16) + toff, 0, 16);
                    }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its own stores
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(reinterpret_cast<int*>(p.ws) + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != p.splits - 1) return;
        __syncthreads();                                          // everyone has read the flag before the staging area is reused
        // all GROUPS loads of one split in flight together (a first version with one quadrant's four at a time cost the last
        // arriver 4 x splits dependent round trips to the memory side, ~1 us each)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        acc[a][b][0][f][i] = 0.f;
                        if (NACC == 2) acc[a][b][NACC - 1][f][i] = 0.f;
                    }
#pragma unroll 1
        for (int sp = 0; sp < p.splits; ++sp) {
            f32x4 v[GROUPS];
#pragma unroll
            for (int i = 0; i < GROUPS; ++i)
                v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (sp * GROUPS + i) * (NT * 16) + toff, 0, 16));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[a][b][0][f][4 * g + e] += v[((a * 2 + b) * NF + f) * 4 + g][e];
        }
        if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<int*>(p.ws) + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    // the bias of the wave's two column ranges, fetched under the barrier (issued inside the quadrant loop it cost every
    // quadrant one exposed memory latency)
    f32x4 biasq[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + b * HN + wn * 32 * FN + 4 * (lane % (8 * FN));
        biasq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) biasq[b] = *reinterpret_cast<const f32x4*>(p.bias + n);
    }
    __syncthreads();                               // every wave is done with the operand stages
    stamp(17);
    float* cred = reinterpret_cast<float*>(smem) + NW * TW + 16;              // column-sum partials [2 A-halves x WM][2 B-halves][HN]
    static_assert((NW * TW + 16 + 4 * WM * HN) * 4 <= Cf::SMEM, "column-sum staging fits behind the wave-private regions");
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        switch (q) {
            case 0: bt_park_quadrant<FM, FN, NACC>(acc[0][0], lane, Tw); break;
            case 1: bt_park_quadrant<FM, FN, NACC>(acc[0][1], lane, Tw); break;
            case 2: bt_park_quadrant<FM, FN, NACC>(acc[1][0], lane, Tw); break;
            default: bt_park_quadrant<FM, FN, NACC>(acc[1][1], lane, Tw); break;
        }
        __builtin_amdgcn_wave_barrier();           // compiler-only: the lanes' writes stay in front of the other lanes' reads
        f32x4 cs = {0.f, 0.f, 0.f, 0.f};
        bt_wave_epilogue<FM, FN>(p, kind, m0 + (q >> 1) * HM + wm * 32 * FM, n0 + (q & 1) * HN + wn * 32 * FN, Tw, lane, sqs, (q & 1) ? biasq[1] : biasq[0],
                                 cs, true);
        if (p.out_colsum) {
            // this quadrant's column sums, folded over the wave's row groups, parked per (A-half, wave row): nothing is carried in
            // registers across the quadrants (eight more live VGPRs spilled next to the 128 accumulators of the 256 x 256 tile)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int d = 8 * FN; d < 64; d <<= 1) cs[e] += __shfl_xor(cs[e], d, 64);
            if (lane < 8 * FN) *reinterpret_cast<f32x4*>(cred + (((q >> 1) * WM + wm) * 2 + (q & 1)) * HN + wn * 32 * FN + 4 * lane) = cs;
        }
        __builtin_amdgcn_wave_barrier();
        stamp(18 + q);
    }
    if (p.out_colsum) {
        // Column sums of the stored tile (the bias gradient of the Linear in front): the 2 WM parked partials of every column are
        // added and ONE value per column and workgroup goes to memory.  (Round 5: one atomic per wave, quadrant and column — 2048
        // per 256 x 256 tile — cost the fc2 input-gradient launches of a batch-32 step 5-10 us each, tools/epi_ablate.py.)
        __syncthreads();
        for (int j = threadIdx.x; j < 2 * HN; j += 64 * NW) {
            const int b = j / HN, c = j % HN, n = n0 + b * HN + c;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 2 * WM; ++w) t += cred[(w * 2 + b) * HN + c];
            if (n < p.N) atomicAdd(p.out_colsum + n, t);
        }
        __syncthreads();                            // (the norm share below reuses words in front)
    }
    if (p.sqacc) {
        // ONE double atomic per workgroup: every launch of a step adds to the same address, and the L2 serialises them — with one
        // per wave a 700-workgroup launch queued 2800 of them (the ws64 pair launches: 27 us instead of 17 inside the step)
        sqs = wave_sum(sqs);
        float* red = reinterpret_cast<float*>(smem) + NW * TW;
        if (lane == 0) red[wave] = sqs;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[w];
            atomicAdd(sq_slot(p), (double)tot);
        }
    }
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(25); }
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_bt_body(const GArgs& p, const int bid, const int zid, unsigned char* smem) {
    using Cf = BtCfg<BM, BN, WM, WN>;
    constexpr int NW = Cf::NW, HM = Cf::HM, HN = Cf::HN, FM = Cf::FM, FN = Cf::FN, NF = FM * FN;
    constexpr int A_HALF = Cf::A_HALF, B_HALF = Cf::B_HALF, BUF = Cf::BUF;
    constexpr bool STAGGER = NW == 8;
    constexpr int PA = pieces<HM, A_KC, NW>(), PB = pieces<HN, B_KC, NW>();
    constexpr int NACC = NF == 1 ? 2 : 1;          // one fragment per quadrant: even / odd k-slices on separate accumulators
    constexpr bool ASM_READS = !A_KC || !B_KC;     // transposing reads are inline asm: ordered by hand
    // XCD-aware AND balanced tile map (block b runs on XCD b % 8): the T = tiles_m * tiles_n tiles are cut into eight contiguous
    // chunks of the linear order (sizes differ by at most one), XCD x takes chunk x.  The linear order walks the SHORTER operand's
    // tiles fastest, so a chunk covers whole column tiles (all row tiles under them) — or whole row tiles when A is the larger
    // operand (xcd_m): an XCD's L2 holds 1/8 of one operand and streams the other.  (The 64-row kernels give XCD x the column
    // tiles x, x + 8, ...: with 18 column tiles two XCDs get three columns and six get two — a 1.5x longer tail.)
    const int T = p.tiles_m * p.tiles_n;
    const int xq = T >> 3, xr = T & 7, xcd = bid & 7;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    if ((bid >> 3) >= xq + (xcd < xr ? 1 : 0)) return;
    const int tn = p.xcd_m ? lin % p.tiles_n : lin / p.tiles_m;
    const int tm = p.xcd_m ? lin / p.tiles_n : lin % p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * p.k_per_split;
    const int nk = (min(p.K, kbeg + p.k_per_split) - kbeg) / BK;         // >= 2 (launcher)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool late = STAGGER && wave >= 4;         // group 1: one barrier behind
    // tools/bt_phase_probe.py: shader-clock stamps of wave 0 / wave 4 (one per stagger group), 32 per (workgroup, group)
    auto stamp = [&](int i) {
        if (p.dbg && lane == 0 && (wave & 3) == 0) p.dbg[(((long)zid * (8 * ((T + 7) >> 3)) + bid) * 2 + (wave >> 2)) * 32 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    f32x16 acc[2][2][NACC][NF];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int h = 0; h < NACC; ++h)
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][h][f][i] = 0.f;

    // half-tile positions in issue order: 0 A-lo, 1 B-lo, 2 B-hi, 3 A-hi; LDS image of a buffer: [A-lo][A-hi][B-lo][B-hi]
    // one DMA instruction (piece j of this wave) of a half-tile
    auto issue_piece = [&](auto pos_c, int tile, int buf, int j) {
        constexpr int POS = decltype(pos_c)::value;
        constexpr int OFF = POS == 0 ? 0 : POS == 3 ? A_HALF : POS == 1 ? 2 * A_HALF : 2 * A_HALF + B_HALF;
        unsigned char* dst = smem + buf * BUF + OFF;
        const int k0 = kbeg + tile * BK;
        if constexpr (POS == 0 || POS == 3) dma_piece<HM, A_KC, NW>(p.A, p.lda, p.M, m0 + (POS == 3 ? HM : 0), a_wrap(p, k0), dst, wave, lane, j);
        else dma_piece<HN, B_KC, NW>(p.B, p.ldb, p.N, n0 + (POS == 2 ? HN : 0), k0, dst, wave, lane, j);
    };
    auto issue = [&](auto pos_c, int tile, int buf) {
        constexpr int POS = decltype(pos_c)::value;
#pragma unroll
        for (int j = 0; j < ((POS == 0 || POS == 3) ? PA : PB); ++j) issue_piece(pos_c, tile, buf, j);
    };
    bf16x8 fa[FM][BK / 16], fb0[FN][BK / 16], fb1[FN][BK / 16];
    auto read_a = [&](int buf, int half) {
        const unsigned char* T = smem + buf * BUF + half * A_HALF;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
            for (int f = 0; f < FM; ++f) fa[f][kk] = frag<HM, A_KC>(T, wm * 32 * FM + f * 32, kk, lane);
    };
    auto read_b = [&](int buf, int half, bf16x8 (&fb)[FN][BK / 16]) {
        const unsigned char* T = smem + buf * BUF + 2 * A_HALF + half * B_HALF;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
            for (int f = 0; f < FN; ++f) fb[f][kk] = frag<HN, B_KC>(T, wn * 32 * FN + f * 32, kk, lane);
    };
    // MFMA slot of one quadrant; the DMA pieces of half-tile POS (k-tile `tile`, buffer `buf`; POS < 0: none) are issued BETWEEN
    // the MFMAs: a piece costs the wave 60-180 clocks of issue, which the matrix pipe spends on the MFMAs already queued —
    // in the load slot the same pieces made that slot (DMA + up to 12 fragment reads) longer than the partner group's 256
    // clocks of MFMA, and the matrix pipe idled at every other barrier (tools/bt_phase_probe.py: 3500 -> clocks per k-tile)
    auto mma = [&](f32x16 (&c)[NACC][NF], bf16x8 (&fb)[FN][BK / 16], auto pos_c, int tile, int buf, int st = -1) {
        constexpr int POS = decltype(pos_c)::value;
        constexpr int NP = POS < 0 ? 0 : (POS == 0 || POS == 3) ? PA : PB, NM = (BK / 16) * NF;
        constexpr int NPD = NP > 0 ? NP : 1, STEP = NM / NPD > 0 ? NM / NPD : 1;
        if (st >= 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(st); }
        if constexpr (ASM_READS) {
            frags_ready();
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                for (int f = 0; f < FM; ++f) frag_tie(fa[f][kk]);
#pragma unroll
                for (int f = 0; f < FN; ++f) frag_tie(fb[f][kk]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    c[kk % NACC][fm * FN + fn] =
                        __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[fm][kk], fb[fn][kk], c[kk % NACC][fm * FN + fn], 0, 0, 0);
                    if constexpr (NP > 0) {
                        const int m = (kk * FM + fm) * FN + fn;
                        if (m % STEP == 0 && m / STEP < NP) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(pos_c, tile, buf, m / STEP);
                            if (m / STEP == NP - 1 || m == NM - 1)
                                for (int r = m / STEP + 1; r < NP && m == NM - 1; ++r) issue_piece(pos_c, tile, buf, r);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
        __builtin_amdgcn_s_setprio(0);
        if (st >= 0) stamp(st + 1);
    };
    // the barrier that ends a load slot; the MFMA slot of a staggered workgroup ends with a second one
    auto load_done = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_done = [&]() {
        if constexpr (STAGGER) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr int W4 = 2 * (PA + PB);               // DMA instructions of four consecutive half-tiles
    using NoDma = std::integral_constant<int, -1>;

    // one k-tile = four phases.  MODE 0: steady state (k-tiles t + 1 and t + 2 exist); 1: t == nk - 2; 2: t == nk - 1.
    // Issue order of the half-tiles: ... A-lo(t) B-lo(t) B-hi(t) A-hi(t) A-lo(t+1) ...; phase g = 4 t + p issues element g + 6 (in its
    // MFMA slot) and waits (in its load slot, i.e. BEFORE that issue) for element g + 2, which phase g + 1 reads: the three
    // elements g + 3 .. g + 5 stay in flight.
#ifndef VITAE_BT_LMASK
#define VITAE_BT_LMASK 0          // bit p set: the DMA of phase p + 1 is issued in its LOAD slot (before the counted wait), else between
                                  // the MFMAs.  Measured (256x256, clocks per k-tile): 0 (all between the MFMAs) 2880; 8: 3150; 2: 3080; 10: 3370; 15: 3530 — a DMA
                                  // issued by the wave in its load slot stalls the MFMA issue of its SIMD partner (that partner's MFMA slot: 320 -> 560)
#endif
    constexpr bool L1 = VITAE_BT_LMASK & 1, L2 = VITAE_BT_LMASK & 2, L3 = VITAE_BT_LMASK & 4, L4 = VITAE_BT_LMASK & 8;
    constexpr int PB_HI = PB, PA_HI = PA, PA_LO = PA, PB_LO = PB;
    auto ktile = [&](auto mode_c, int t) {
        constexpr int MODE = decltype(mode_c)::value;
        const int buf = t & 1;
        const int sb = (MODE == 0 && p.dbg && t == 2) ? 3 : -100;       // stamps 3..14 in k-tile 2
        // P1: (A-lo, B-lo); issues B-hi(t + 1); needs B-hi(t) for P2: younger A-hi(t), A-lo(t+1), B-lo(t+1) (+ its own)
        if constexpr (MODE <= 1 && L1) issue(std::integral_constant<int, 2>{}, t + 1, buf ^ 1);
        read_a(buf, 0);
        read_b(buf, 0, fb0);
        if constexpr (MODE <= 1) wait_vmcnt<2 * PA + PB + (L1 ? PB_HI : 0)>(); else wait_vmcnt<PA>();
        load_done();
        if constexpr (MODE <= 1 && !L1) mma(acc[0][0], fb0, std::integral_constant<int, 2>{}, t + 1, buf ^ 1, sb);
        else mma(acc[0][0], fb0, NoDma{}, 0, 0, sb);
        mma_done();
        if (sb >= 0) stamp(sb + 2);
        // P2: (A-lo, B-hi); issues A-hi(t + 1); needs A-hi(t) for P3: younger A-lo(t+1), B-lo(t+1), B-hi(t+1) (+ its own)
        if constexpr (MODE <= 1 && L2) issue(std::integral_constant<int, 3>{}, t + 1, buf ^ 1);
        read_b(buf, 1, fb1);
        if constexpr (MODE <= 1) wait_vmcnt<PA + 2 * PB + (L2 ? PA_HI : 0)>(); else wait_vmcnt<0>();
        load_done();
        if constexpr (MODE <= 1 && !L2) mma(acc[0][1], fb1, std::integral_constant<int, 3>{}, t + 1, buf ^ 1, sb + 3);
        else mma(acc[0][1], fb1, NoDma{}, 0, 0, sb + 3);
        mma_done();
        if (sb >= 0) stamp(sb + 5);
        // P3: (A-hi, B-hi); issues A-lo(t + 2); P4 reads nothing
        if constexpr (MODE == 0 && L3) issue(std::integral_constant<int, 0>{}, t + 2, buf);
        read_a(buf, 1);
        load_done();
        if constexpr (MODE == 0 && !L3) mma(acc[1][1], fb1, std::integral_constant<int, 0>{}, t + 2, buf, sb + 6);
        else mma(acc[1][1], fb1, NoDma{}, 0, 0, sb + 6);
        mma_done();
        if (sb >= 0) stamp(sb + 8);
        // P4: (A-hi, B-lo); issues B-lo(t + 2); needs A-lo, B-lo(t+1) for P1: younger B-hi(t+1), A-hi(t+1), A-lo(t+2) (+ its own)
        if constexpr (MODE == 0 && L4) issue(std::integral_constant<int, 1>{}, t + 2, buf);
        if constexpr (MODE == 0) wait_vmcnt<2 * PA + PB + (L4 ? PB_LO : 0)>(); else if constexpr (MODE == 1) wait_vmcnt<PA + PB>();
        load_done();
        if constexpr (MODE == 0 && !L4) mma(acc[1][0], fb0, std::integral_constant<int, 1>{}, t + 2, buf, sb + 9);
        else mma(acc[1][0], fb0, NoDma{}, 0, 0, sb + 9);
        mma_done();
        if (sb >= 0) stamp(sb + 11);
    };

    // prologue: k-tile 0 completely, A-lo and B-lo of k-tile 1
    issue(std::integral_constant<int, 0>{}, 0, 0);
    issue(std::integral_constant<int, 1>{}, 0, 0);
    issue(std::integral_constant<int, 2>{}, 0, 0);
    issue(std::integral_constant<int, 3>{}, 0, 0);
    issue(std::integral_constant<int, 0>{}, 1, 1);
    issue(std::integral_constant<int, 1>{}, 1, 1);
    stamp(1);
    wait_vmcnt<W4>();                               // A-lo(0), B-lo(0) landed
    load_done();
    if (late) load_done();                          // group 1 starts one barrier later
    stamp(2);
#pragma unroll 1
    for (int t = 0; t < nk - 2; ++t) {
        if (t == 2) stamp(15);
        ktile(std::integral_constant<int, 0>{}, t);
    }
    ktile(std::integral_constant<int, 1>{}, nk - 2);
    ktile(std::integral_constant<int, 2>{}, nk - 1);
    stamp(16);
    if (STAGGER && !late) load_done();              // group 0 meets group 1's last barrier

    bt_tail<BM, BN, WM, WN, NACC>(p, acc, smem, m0, n0, tm, tn, zid, wave, lane, stamp);
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_bt_kernel(const GArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BtCfg<BM, BN, WM, WN>::SMEM];      // the ONLY LDS object
    gemm_bt_body<BM, BN, WM, WN, A_KC, B_KC>(p, blockIdx.x, blockIdx.z, smem);
}

// ---- wave-specialised 128 x 128 tile (tile id 4, round 4) --------------------------------------------------------------------
// Why: on the 128 x 128 tile above every wave issues its own LDS-DMA pieces between its MFMAs.  A piece occupies the issuing wave
// for 60-180 clocks (the CU's texture addresser takes ~45 B/clk, and the wave sits at the instruction until its 1 KB is accepted):
// eight pieces per wave and k-tile are 500-1400 clocks during which that wave issues no MFMA — a lone workgroup runs ~1465 clocks
// per 64-deep k-tile for 512 clocks of MFMA, and two workgroups per CU are no faster (LABNOTES round-4 notes).  Here the roles are
// split: waves 0-3 (one per SIMD) only read fragments and issue MFMAs, waves 4-7 (their SIMD partners) only issue DMA.  A blocked
// DMA instruction stalls nobody but its own wave; the matrix pipe of the SIMD keeps running on the consumer wave.
//   * S stages of one whole k-tile (A 128 x 64 | B 128 x 64 = 32 KB); producers run S - 1 tiles ahead;
//   * ONE workgroup barrier per k-tile.  B_t (t = 0 .. nk - 1) promises: tile t has landed (every producer waited for its own
//     pieces of it with a counted vmcnt before arriving) and every fragment read of tile t - 1 has retired (every consumer waited
//     lgkmcnt(0) before arriving), so the producers restage tile t - 1's slot with tile t + S - 1 right after it;
//   * a consumer crosses B_{t+1} in the MIDDLE of k-tile t (after the MFMAs of k-slices 0-1, with the fragments of slices 2-3
//     already in registers) and reads slices 0-1 of tile t + 1 under the MFMAs of slices 2-3: no fragment-read latency is exposed
//     at a tile boundary;
//   * producers leave after the last barrier; the consumers run the family's tail (in-launch split-K fix-up, wave-private epilogue).
#ifndef VITAE_WS_STAGES
#define VITAE_WS_STAGES 3         // (3 / 4 / 5 stages measured alike: the loop is bound by the L2 -> LDS feed, not by its latency)
#endif
#ifndef VITAE_WS_TAIL8
#define VITAE_WS_TAIL8 1          // unsplit launches: the producer waves take half of the epilogue (quadrants A-hi x B-lo / B-hi of their SIMD partner)
#endif
#ifndef VITAE_WS_TAIL_ALIAS
#define VITAE_WS_TAIL_ALIAS 0     // 1: the tail's LDS (parked quadrants, wave-private staging) ALIASES the operand stages behind one more barrier, so
                                  // that 4 or 5 stages fit.  Measured with the interleaved consumer (tools/probes/r5_ws_ab5.sh, us, 3 / 3 aliased / 4 / 5
                                  // stages): 3520 x 768 x 3072 27.0 / 27.8 / 27.8 / 27.8, its dgrad 28.6 / 28.5 / 28.6 / 28.6, 6944 x 512 x 2048 22.3 /
                                  // 22.8 / 22.8 / 22.7 — a deeper pipeline buys nothing: the loop sits on the CU's L2 -> LDS feed (32 KB per ~870 clocks =
                                  // 37 B/clk, what tools/probes/mfma_rate.hip gets for bare MFMAs beside the same DMA stream)
#endif
#ifndef VITAE_WS_INTERLEAVE
#define VITAE_WS_INTERLEAVE 1     // fragment reads one per MFMA gap (round 5) instead of bursts of eight between the MFMA blocks
#endif
#ifndef VITAE_WS_FINE_STAMPS
#define VITAE_WS_FINE_STAMPS 0    // 1: s_memtime stamps inside ONE k-tile of the consumer (variant library for tools/ws_phase_probe.py fine)
#endif
#ifndef VITAE_WS_ABLATE
#define VITAE_WS_ABLATE 0         // timing experiments (results are garbage): 1 no MFMA, 2 half of the DMA pieces, 4 no fragment reads,
                                  // 8 ONE producer wave issues every piece (correct results), 16 consumer wave 0 (the producer's SIMD partner) skips its MFMAs
#endif
template <bool A_KC, bool B_KC, int S>
__device__ __forceinline__ void gemm_ws_body(const GArgs& p, const int bid, const int zid, unsigned char* smem) {
    constexpr int BM = 128, BN = 128, NWC = 4, NWP = (VITAE_WS_ABLATE & 8) ? 1 : 4;
    constexpr int A_T = BM * BK * 2, B_T = BN * BK * 2, STG = A_T + B_T;
    constexpr int PA = pieces<BM, A_KC, NWP>(), PB = pieces<BN, B_KC, NWP>(), PT = ((VITAE_WS_ABLATE & 32) ? 1 : PA) + ((VITAE_WS_ABLATE & 2) ? 0 : PB);
    constexpr int TAILOFF = VITAE_WS_TAIL_ALIAS ? 0 : S * STG;       // byte offset of the tail's LDS
    static_assert(S >= 3 && S <= 5 && (VITAE_WS_TAIL_ALIAS ? S * STG : S * STG + (VITAE_WS_TAIL8 ? 12 * 4096 + 64 : 0)) <= 160 * 1024 && (S - 2) * PT <= 63, "stages fit LDS, three tiles of pieces fit the vmcnt field");
    const int T = p.tiles_m * p.tiles_n;
    const int xq = T >> 3, xr = T & 7, xcd = bid & 7;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    if ((bid >> 3) >= xq + (xcd < xr ? 1 : 0)) return;
    const int tn = p.xcd_m ? lin % p.tiles_n : lin / p.tiles_m;
    const int tm = p.xcd_m ? lin / p.tiles_n : lin % p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * p.k_per_split;
    const int nk = (min(p.K, kbeg + p.k_per_split) - kbeg) / BK;         // >= 2 (launcher)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stamp = [&](int i) {
        if (p.dbg && lane == 0 && (wave & 3) == 0) p.dbg[(((long)zid * (8 * ((T + 7) >> 3)) + bid) * 2 + (wave >> 2)) * 32 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    if (wave >= NWC) {
        // ---------------- producers: DMA only ----------------
        const int pw = wave - NWC;
        auto issue_tile = [&](int t, int stage) {
            unsigned char* dst = smem + stage * STG;
            const int k0 = kbeg + t * BK;
#pragma unroll
            for (int j = 0; j < ((VITAE_WS_ABLATE & 32) ? 1 : PA); ++j) dma_piece<BM, A_KC, NWP>(p.A, p.lda, p.M, m0, a_wrap(p, k0), dst, pw, lane, j);
#pragma unroll
            for (int j = 0; j < ((VITAE_WS_ABLATE & 2) ? 0 : PB); ++j) dma_piece<BN, B_KC, NWP>(p.B, p.ldb, p.N, n0, k0, dst + A_T, pw, lane, j);
        };
        auto wait_tiles = [&](int fly) {       // all but the youngest `fly` tiles of this wave's pieces have landed
            if (S >= 5 && fly >= 3) wait_vmcnt<(S >= 5 ? 3 : 1) * PT>();
            else if (S >= 4 && fly >= 2) wait_vmcnt<(S >= 4 ? 2 : 1) * PT>();
            else if (fly >= 1) wait_vmcnt<PT>();
            else wait_vmcnt<0>();
        };
        const int npre = min(S - 1, nk);
        const bool active = pw < NWP;
        for (int t = 0; t < npre; ++t) if (active) issue_tile(t, t);
        wait_tiles(npre - 1);
        barrier();                                                       // B_0
        int stage = npre % S;                                            // slot of tile t + S - 1 (= tile t - 1's)
#pragma unroll 1
        for (int t = 0; t + 1 < nk; ++t) {
            const bool pr = p.dbg && t >= 3 && t < 9;                    // tools/ws_phase_probe.py: who arrives last at B_4 .. B_9
            if (t + S - 1 < nk) {
                if (active) issue_tile(t + S - 1, stage);
                stage = stage + 1 == S ? 0 : stage + 1;
            }
            if (pr) stamp(2 + 3 * (t - 3));                               // pieces issued
            wait_tiles(min(t + S - 1, nk - 1) - (t + 1));                // B_{t+1}: tile t + 1 has landed
            if (pr) stamp(3 + 3 * (t - 3));                               // arrival
            barrier();
            if (pr) stamp(4 + 3 * (t - 3));                               // departure
        }
        if (!VITAE_WS_TAIL8 || p.splits > 1) return;                     // (split launches: the consumers' fix-up has barriers of its own)
        // the epilogue of the partner's two A-hi quadrants, parked by it in this wave's two LDS regions behind the stages
        const int wmp = pw >> 1, wnp = pw & 1;
        f32x4 bq[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + b * 64 + wnp * 32 + 4 * (lane % 8);
            bq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) bq[b] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
        const int kind = bt_epilogue_kind(p);
        const float* Pq = reinterpret_cast<const float*>(smem + TAILOFF) + pw * 2048;
        float sqs = 0.f;
        if (VITAE_WS_TAIL_ALIAS) barrier();                              // E: every consumer is past its last fragment read (the tail aliases the stages)
        barrier();                                                       // T: the partner's quadrants are in LDS
#pragma unroll 1
        for (int b = 0; b < 2; ++b) { f32x4 cz = {0.f, 0.f, 0.f, 0.f}; bt_wave_epilogue<1, 1>(p, kind, m0 + 64 + wmp * 32, n0 + b * 64 + wnp * 32, Pq + b * 1024, lane, sqs, b ? bq[1] : bq[0], cz, false); }
        if (p.sqacc) {                                                   // one atomic per workgroup (see bt_tail)
            sqs = wave_sum(sqs);
            float* red = reinterpret_cast<float*>(smem + TAILOFF + 12 * 4096);
            if (lane == 0) red[wave] = sqs;
            barrier();
        }
        return;
    }

    // ---------------- consumers: fragment reads + MFMA ----------------
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2][1][1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][0][0][i] = 0.f;
    bf16x8 fa[BK / 16][2], fb[BK / 16][2];
    if (VITAE_WS_ABLATE & 4) {
#pragma unroll
        for (int k = 0; k < BK / 16; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) { fa[k][h][e] = (__bf16)(float)lane; fb[k][h][e] = (__bf16)1.f; }
    }
    constexpr int RK = (A_KC ? 2 : 4) + (B_KC ? 2 : 4);                 // LDS instructions of one k-slice's fragments
    constexpr int W2 = 2 * RK > 15 ? 15 : 2 * RK;
    auto rd = [&](const unsigned char* TA, auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        const unsigned char* TB = TA + A_T;
        if (VITAE_WS_ABLATE & 4) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) fa[kk][h] = frag_asm<BM, A_KC>(TA, h * 64 + wm * 32, kk, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h) fb[kk][h] = frag_asm<BN, B_KC>(TB, h * 64 + wn * 32, kk, lane);
    };
    auto mm = [&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) { frag_tie(fa[kk][h]); frag_tie(fb[kk][h]); }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (!(VITAE_WS_ABLATE & 1) && !((VITAE_WS_ABLATE & 16) && wave == 0)) acc[a][b][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][a], fb[kk][b], acc[a][b][0][0], 0, 0, 0);
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    barrier();                                                           // B_0: tile 0 is in LDS
    rd(smem, K0{}); rd(smem, K1{});
    int stage = 0;
#if VITAE_WS_INTERLEAVE
    // Round 5 (tools/ws_phase_probe.py): the k-loop was bound by the CONSUMER — every barrier found the producers waiting 200-330
    // clocks, the interval was 1080 clocks for 512 of MFMA.  An in-order wave that issues eight fragment reads in a burst sits at
    // them for 100-300 clocks (four waves share the LDS port with the DMA's writes) while its matrix pipe drains.  Here the reads of
    // the two slices ahead are issued ONE FRAGMENT PER MFMA GAP: the four MFMAs of a slice carry the eight fragments of the next
    // pair of slices (each read issues in the 32-clock shadow of the MFMA in front of it).
    //   slice 0 MFMAs + reads of slices 2, 3 | slice 1 MFMAs | all reads of tile t retired, B_{t+1} | slice 2 MFMAs + reads of the
    //   next tile's slices 0, 1 | slice 3 MFMAs
    const FragOff<BM, A_KC> offa = frag_offsets<BM, A_KC>(wm * 32, lane);
    const FragOff<BN, B_KC> offb = frag_offsets<BN, B_KC>(wn * 32, lane);
    typedef __attribute__((address_space(3))) const unsigned char lds_u8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_u8*)smem;
    // Eight MFMAs (k-slices S0, S0 + 1) that carry the eight fragments of slices R0, R0 + 1 of the tile at LDS address `rb`:
    // gap:       0        1      2      3        4      5     6  7
    // fragments  a0 b0    a0'    b0'    a1 b1    a1'    b1'   -  -      (x' = the half 64 rows further; the last two gaps stay
    // empty so that the lgkmcnt(0) behind the block finds every read retired)
    auto half = [&](auto s0_c, const unsigned rb, auto r0_c, bool do_rd) {
        constexpr int S0 = decltype(s0_c)::value, R0 = decltype(r0_c)::value;
#pragma unroll
        for (int k = S0; k < S0 + 2; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) { frag_tie(fa[k][h]); frag_tie(fb[k][h]); }
        __builtin_amdgcn_sched_barrier(0);
        unsigned aa = 0, ab = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = S0 + (j >> 2), a = (j >> 1) & 1, b = j & 1;
            acc[a][b][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][a], fb[kk][b], acc[a][b][0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (do_rd && j < 6) {
                if (j == 0) {
                    aa = frag_base<BM, A_KC, R0>(rb, offa, 0); ab = frag_base<BN, B_KC, R0>(rb + A_T, offb, 0);
                    fa[R0][0] = frag_rd<BM, A_KC, R0, 0>(aa); fb[R0][0] = frag_rd<BN, B_KC, R0, 0>(ab);
                }
                if (j == 1) { if (!A_KC) aa = frag_base<BM, A_KC, R0>(rb, offa, 1); fa[R0][1] = frag_rd<BM, A_KC, R0, A_KC ? 1 : 0>(aa); }
                if (j == 2) { if (!B_KC) ab = frag_base<BN, B_KC, R0>(rb + A_T, offb, 1); fb[R0][1] = frag_rd<BN, B_KC, R0, B_KC ? 1 : 0>(ab); }
                if (j == 3) {
                    aa = frag_base<BM, A_KC, R0 + 1>(rb, offa, 0); ab = frag_base<BN, B_KC, R0 + 1>(rb + A_T, offb, 0);
                    fa[R0 + 1][0] = frag_rd<BM, A_KC, R0 + 1, 0>(aa); fb[R0 + 1][0] = frag_rd<BN, B_KC, R0 + 1, 0>(ab);
                }
                if (j == 4) { if (!A_KC) aa = frag_base<BM, A_KC, R0 + 1>(rb, offa, 1); fa[R0 + 1][1] = frag_rd<BM, A_KC, R0 + 1, A_KC ? 1 : 0>(aa); }
                if (j == 5) { if (!B_KC) ab = frag_base<BN, B_KC, R0 + 1>(rb + A_T, offb, 1); fb[R0 + 1][1] = frag_rd<BN, B_KC, R0 + 1, B_KC ? 1 : 0>(ab); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        // entry: the fragments of slices 0, 1 of tile t were read under the MFMAs of the previous tile's slices 2, 3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        half(K0{}, lds0 + stage * STG, K2{}, true);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // every read of tile t has retired (the last one two MFMAs ago)
        const bool more = t + 1 < nk;
        if (more) {
            barrier();                                                   // B_{t+1}
            stage = stage + 1 == S ? 0 : stage + 1;
        }
        __builtin_amdgcn_s_setprio(1);
        half(K2{}, lds0 + stage * STG, K0{}, more);
        __builtin_amdgcn_s_setprio(0);
    }
#else
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const unsigned char* TA = smem + stage * STG;
#if VITAE_WS_FINE_STAMPS
        const bool fine = p.dbg && t == 12;                              // tools/ws_phase_probe.py fine: the pieces of ONE k-tile (perturbs the pipeline)
        if (fine) stamp(21);
#endif
        __builtin_amdgcn_sched_barrier(0);
        rd(TA, K2{}); rd(TA, K3{});
        __builtin_amdgcn_sched_barrier(0);
#if VITAE_WS_FINE_STAMPS
        if (fine) stamp(22);
#endif
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(W2) : "memory");     // slices 0-1 (read one barrier ago) are in registers
        __builtin_amdgcn_sched_barrier(0);
#if VITAE_WS_FINE_STAMPS
        if (fine) stamp(23);
#endif
        __builtin_amdgcn_s_setprio(1);
        mm(K0{}); mm(K1{});
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#if VITAE_WS_FINE_STAMPS
        if (fine) stamp(24);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // every read of tile t has retired: its slot may be restaged
        if (t + 1 < nk) {
            const bool pr = p.dbg && t >= 3 && t < 9;
            if (pr) stamp(2 + 3 * (t - 3));                               // arrival at B_{t+1}
#if VITAE_WS_FINE_STAMPS
            if (fine) stamp(25);
#endif
            barrier();                                                   // B_{t+1}
            if (pr) stamp(3 + 3 * (t - 3));                               // departure
#if VITAE_WS_FINE_STAMPS
            if (fine) stamp(26);
#endif
            stage = stage + 1 == S ? 0 : stage + 1;
            const unsigned char* TN = smem + stage * STG;
            rd(TN, K0{}); rd(TN, K1{});
            __builtin_amdgcn_sched_barrier(0);
#if VITAE_WS_FINE_STAMPS
            if (fine) stamp(27);
#endif
        }
        __builtin_amdgcn_s_setprio(1);
        mm(K2{}); mm(K3{});
        __builtin_amdgcn_s_setprio(0);
#if VITAE_WS_FINE_STAMPS
        if (fine) stamp(28);
#endif
    }
#endif
    stamp(20);
    if (VITAE_WS_TAIL8 && p.splits == 1) {
        // unsplit launch: quadrants (A-hi, B-lo) and (A-hi, B-hi) go to the SIMD partner (a producer wave, idle by now) through
        // its two 4 KB regions behind the stages; this wave keeps (A-lo, B-lo) and (A-lo, B-hi).  Nothing here touches the stages,
        // which slower waves may still be reading.
        float* Pq = reinterpret_cast<float*>(smem + TAILOFF) + wave * 2048;
        float* Tw = reinterpret_cast<float*>(smem + TAILOFF) + 4 * 2048 + wave * 1024;
        if (VITAE_WS_TAIL_ALIAS) barrier();                              // E (no DMA is in flight: the producers waited for the last tile before B_{nk-1})
        f32x4 bq[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + b * 64 + wn * 32 + 4 * (lane % 8);
            bq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) bq[b] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
        const int kind = bt_epilogue_kind(p);
        bt_park_quadrant<1, 1, 1>(acc[1][0], lane, Pq);
        bt_park_quadrant<1, 1, 1>(acc[1][1], lane, Pq + 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the parked values are in LDS before the partner is released
        barrier();                                                       // T
        float sqs = 0.f;
#pragma unroll 1
        for (int b = 0; b < 2; ++b) {
            if (b == 0) bt_park_quadrant<1, 1, 1>(acc[0][0], lane, Tw);
            else bt_park_quadrant<1, 1, 1>(acc[0][1], lane, Tw);
            __builtin_amdgcn_wave_barrier();
            { f32x4 cz = {0.f, 0.f, 0.f, 0.f}; bt_wave_epilogue<1, 1>(p, kind, m0 + wm * 32, n0 + b * 64 + wn * 32, Tw, lane, sqs, b ? bq[1] : bq[0], cz, false); }
            __builtin_amdgcn_wave_barrier();
        }
        if (p.sqacc) {
            sqs = wave_sum(sqs);
            float* red = reinterpret_cast<float*>(smem + TAILOFF + 12 * 4096);
            if (lane == 0) red[wave] = sqs;
            barrier();
            if (threadIdx.x == 0) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) tot += red[w];
                atomicAdd(sq_slot(p), (double)tot);
            }
        }
        return;
    }
    bt_tail<BM, BN, 2, 2, 1>(p, acc, smem, m0, n0, tm, tn, zid, wave, lane, stamp);
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(const GArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[VITAE_WS_TAIL_ALIAS ? VITAE_WS_STAGES * 32768 : VITAE_WS_STAGES * 32768 + (VITAE_WS_TAIL8 ? 12 * 4096 + 64 : 0)];      // the ONLY LDS object
    gemm_ws_body<A_KC, B_KC, VITAE_WS_STAGES>(p, blockIdx.x, blockIdx.z, smem);
}

// ---- a 128 x 256 tile of the same structure for the WEIGHT-GRADIENT form (tile id 6; both operands row-contiguous) ---------------
// Why (round 5): the wave-specialised loop is bound by the CU's L2 -> LDS rate (37 B/clk), so what a k-tile costs is its BYTES:
// 32 KB for 128 x 128 (870 clocks), 48 KB for 128 x 256 (~1300) — 1.34x the outputs per byte.  A block's four weight gradients are
// 432 tiles of 128 x 128 on 256 one-workgroup-per-CU slots (1.69 rounds = 2); as 216 tiles of 128 x 256 they are ONE round.  The
// consumers keep the 2 x 2 arrangement of the family's tail (BtCfg<128, 256, 2, 2>: a wave owns rows a * 64 + wm * 32 and columns
// b * 128 + wn * 64 + f * 32, a, b, f in {0, 1}: eight 32 x 32 accumulators); eight MFMAs per k-slice carry the six fragments of a
// slice two ahead, one per MFMA gap.  Three stages of 48 KB; the tail (split fix-up, wave-private epilogue) aliases them.
template <int S>
__device__ __forceinline__ void gemm_wsw_body(const GArgs& p, const int bid, const int zid, unsigned char* smem) {
    constexpr int BM = 128, BN = 256, NWC = 4, NWP = 4;
    constexpr int A_T = BM * BK * 2, B_T = BN * BK * 2, STG = A_T + B_T;
    constexpr int PA = pieces<BM, false, NWP>(), PB = pieces<BN, false, NWP>(), PT = PA + PB;
    static_assert(S == 3 && S * STG <= 160 * 1024 && (S - 2) * PT <= 63, "three stages fit LDS, one tile of pieces fits the vmcnt field");
    const int T = p.tiles_m * p.tiles_n;
    const int xq = T >> 3, xr = T & 7, xcd = bid & 7;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    if ((bid >> 3) >= xq + (xcd < xr ? 1 : 0)) return;
    const int tn = p.xcd_m ? lin % p.tiles_n : lin / p.tiles_m;
    const int tm = p.xcd_m ? lin / p.tiles_n : lin % p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * p.k_per_split;
    const int nk = (min(p.K, kbeg + p.k_per_split) - kbeg) / BK;         // >= 2 (launcher)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stamp = [&](int) {};

    if (wave >= NWC) {
        // ---------------- producers: DMA only (the protocol of gemm_ws_body) ----------------
        const int pw = wave - NWC;
        auto issue_tile = [&](int t, int stage) {
            unsigned char* dst = smem + stage * STG;
            const int k0 = kbeg + t * BK;
#pragma unroll
            for (int j = 0; j < PA; ++j) dma_piece<BM, false, NWP>(p.A, p.lda, p.M, m0, k0, dst, pw, lane, j);
#pragma unroll
            for (int j = 0; j < PB; ++j) dma_piece<BN, false, NWP>(p.B, p.ldb, p.N, n0, k0, dst + A_T, pw, lane, j);
        };
        const int npre = min(S - 1, nk);
        for (int t = 0; t < npre; ++t) issue_tile(t, t);
        if (npre > 1) wait_vmcnt<PT>(); else wait_vmcnt<0>();
        barrier();                                                       // B_0
        int stage = npre % S;
#pragma unroll 1
        for (int t = 0; t + 1 < nk; ++t) {
            if (t + S - 1 < nk) {
                issue_tile(t + S - 1, stage);
                stage = stage + 1 == S ? 0 : stage + 1;
            }
            if (min(t + S - 1, nk - 1) - (t + 1) >= 1) wait_vmcnt<PT>(); else wait_vmcnt<0>();     // B_{t+1}: tile t + 1 has landed
            barrier();
        }
        return;                                                          // (a barrier only counts living waves: the tail belongs to the consumers)
    }

    // ---------------- consumers ----------------
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2][1][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][0][f][i] = 0.f;
    bf16x8 fa[BK / 16][2], fb[BK / 16][4];
    // per-lane offsets inside a stage, computed once: A fragments at rows wm * 32 (+ 64), B fragments at columns wn * 64 + {0, 32, 128, 160}
    const FragOff<BM, false> offa = frag_offsets<BM, false>(wm * 32, lane);
    unsigned ob[4];
    {
        constexpr int LB = BN * 2;
        const int gg = lane >> 4, li = lane & 15;
        const int kl = 8 * (gg >> 1) + (li >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = wn * 64 + (i >> 1) * 128 + (i & 1) * 32 + 16 * (gg & 1) + 4 * (li & 3);
            ob[i] = A_T + kl * LB + (((col >> 3) ^ swz<false, BN / 8>(kl)) << 4) + (col & 7) * 2;
        }
    }
    typedef __attribute__((address_space(3))) const unsigned char lds_u8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_u8*)smem;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    // all six fragments of k-slice KK of the tile at LDS address rb
    auto rd_slice = [&](const unsigned rb, auto kk_c) {
        constexpr int KK = decltype(kk_c)::value;
        fa[KK][0] = frag_rd<BM, false, KK, 0>(rb + offa.o[0]);
        fa[KK][1] = frag_rd<BM, false, KK, 0>(rb + offa.o[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[KK][i] = frag_rd<BN, false, KK, 0>(rb + ob[i]);
    };
    // Sixteen MFMAs (k-slices S0, S0 + 1) that carry the twelve fragments of slices R0, R0 + 1 of the tile at `rb`, one per gap
    // (the last four gaps stay empty: the lgkmcnt(0) behind the block finds every read retired)
    auto half = [&](auto s0_c, const unsigned rb, auto r0_c, bool do_rd) {
        constexpr int S0 = decltype(s0_c)::value, R0 = decltype(r0_c)::value;
#pragma unroll
        for (int k = S0; k < S0 + 2; ++k) {
            frag_tie(fa[k][0]); frag_tie(fa[k][1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) frag_tie(fb[k][i]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kk = S0 + (j >> 3), a = (j >> 2) & 1, bf = j & 3;
            acc[a][bf >> 1][0][bf & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][a], fb[kk][bf], acc[a][bf >> 1][0][bf & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (do_rd && j < 12) {
                if (j == 0) fa[R0][0] = frag_rd<BM, false, R0, 0>(rb + offa.o[0]);
                if (j == 1) fb[R0][0] = frag_rd<BN, false, R0, 0>(rb + ob[0]);
                if (j == 2) fa[R0][1] = frag_rd<BM, false, R0, 0>(rb + offa.o[1]);
                if (j == 3) fb[R0][1] = frag_rd<BN, false, R0, 0>(rb + ob[1]);
                if (j == 4) fb[R0][2] = frag_rd<BN, false, R0, 0>(rb + ob[2]);
                if (j == 5) fb[R0][3] = frag_rd<BN, false, R0, 0>(rb + ob[3]);
                if (j == 6) fa[R0 + 1][0] = frag_rd<BM, false, R0 + 1, 0>(rb + offa.o[0]);
                if (j == 7) fb[R0 + 1][0] = frag_rd<BN, false, R0 + 1, 0>(rb + ob[0]);
                if (j == 8) fa[R0 + 1][1] = frag_rd<BM, false, R0 + 1, 0>(rb + offa.o[1]);
                if (j == 9) fb[R0 + 1][1] = frag_rd<BN, false, R0 + 1, 0>(rb + ob[1]);
                if (j == 10) fb[R0 + 1][2] = frag_rd<BN, false, R0 + 1, 0>(rb + ob[2]);
                if (j == 11) fb[R0 + 1][3] = frag_rd<BN, false, R0 + 1, 0>(rb + ob[3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    barrier();                                                           // B_0: tile 0 is in LDS
    rd_slice(lds0, K0{}); rd_slice(lds0, K1{});
    int stage = 0;
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        half(K0{}, lds0 + stage * STG, K2{}, true);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // every read of tile t has retired
        const bool more = t + 1 < nk;
        if (more) {
            barrier();                                                   // B_{t+1}
            stage = stage + 1 == S ? 0 : stage + 1;
        }
        __builtin_amdgcn_s_setprio(1);
        half(K2{}, lds0 + stage * STG, K0{}, more);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bt_tail<BM, BN, 2, 2, 1>(p, acc, smem, m0, n0, tm, tn, zid, wave, lane, stamp);
}

// ---- the same structure on a 64 x 64 tile (tile id 5): the few-row launches of the batch-4 / batch-8 step ----------------------
// At 440-1736 token rows a launch is 84-700 workgroups of a handful of k-tiles each, and a workgroup's k-step is a dependent
// chain (DMA issue, fragment reads, MFMAs, wait + barrier: ~610 clocks per 64-deep step in gemm_glds.hip, ~880 in its pipelined
// form inside the step).  With the roles split the chain is gone: a k-step costs what its 16 KB of operands cost the CU's
// L2 -> LDS path (~400 clocks).  Measured alone (us, ws64 / 64-row family / hipBLASLt): 440 x 2304 x 768 7.2 / 10.4 / 8.3,
// 440 x 768 x 768 6.5 / 8.4 / 7.2, 868 x 2048 x 512 7.8 / 9.7 / 9.4, weight-gradient form 2304 x 768 x 448 7.7 / 9.4.
// Each consumer wave owns ONE 32 x 32 fragment (two accumulators: even / odd k-slices).  80 KB of LDS and 97 VGPRs: two workgroups
// share a CU.  In-launch split-K as in the family (partials through write-through stores, ticket, last arriver sums in split
// order); RS: the consumer waves of column tile 0 add the row sums of A (one more MFMA against ones per k-slice) = the bias
// gradient colsum(dy) of a weight-gradient launch.
#ifndef VITAE_WS64_STAGES
#define VITAE_WS64_STAGES 4         // 64 KB (the epilogue aliases the stages): TWO workgroups per CU, three k-tiles in flight each
#endif
constexpr int WS64_SMEM = VITAE_WS64_STAGES * 16384;
// W2 (round 5, forward form only): the B operand (a weight matrix) comes as TWO bf16 planes, hi = bf16(W) and lo = bf16(W - hi)
// (p.B / p.B2, same layout): a stage is [A | B hi | B lo] and every k-slice takes two MFMAs, x16 Wlo^T + x16 Whi^T — the weight
// enters at ~2^-17 instead of 2^-9 while the activations stay bf16.  Why: tools/bf16_rounding_ablation.py — the bf16 schedule's
// loss error against the fp32 reference is carried by the rounding of the WEIGHTS in the forward (1.3e-4 of 1.5e-4 on the total;
// activations 9e-6, the whole backward 4e-6), and the decoder's fc1 alone is 1.2e-4 of it.
// What a workgroup needs before its first DMA can leave (tile decoding + operand addressing).  The stand-alone kernels take these as
// LEADING SCALAR kernel arguments: with -amdgpu-kernarg-preload-count=16 they arrive in SGPRs with the dispatch, while the by-value
// descriptor behind them is fetched by scalar loads that nothing on the producers' path waits for (round 6).
struct WsHot {
    const __bf16* A; const __bf16* B; const __bf16* B2;        // (B2: the lo plane of a two-plane weight, else unused)
    int lda, ldb, M, N, K, tiles_m, tiles_n, xcd_m, kps;      // xcd_m: bit 0 the XCD map, bit 1 "p.dbg is set"
};
__device__ __forceinline__ WsHot ws_hot(const GArgs& p) {
    return WsHot{p.A, p.B, p.B2, (int)p.lda, (int)p.ldb, p.M, p.N, p.K, p.tiles_m, p.tiles_n, p.xcd_m | (p.dbg ? 2 : 0), p.k_per_split};
}
// the stand-alone kernels' leading scalars (13 dwords; 14 can be preloaded): the tile counts follow from M and N (64 x 64 tiles)
#define WS_HOT_PARAMS const __bf16* A, const __bf16* B, const __bf16* B2, int lda, int ldb, int M, int N, int K, int xcd_m, int kps
#define WS_HOT_FROM_PARAMS WsHot{A, B, B2, lda, ldb, M, N, K, (M + 63) >> 6, (N + 63) >> 6, xcd_m, kps}
#define WS_HOT_ARGS(p) (p).A, (p).B, (p).B2, (int)(p).lda, (int)(p).ldb, (p).M, (p).N, (p).K, (p).xcd_m | ((p).dbg ? 2 : 0), (p).k_per_split
template <bool A_KC, bool B_KC, int S, bool RS, bool W2 = false>
__device__ __forceinline__ void gemm_ws64_body(const GArgs& p, const WsHot& h, const int bid, const int zid, unsigned char* smem) {
#ifndef VITAE_WS64_PRODUCERS
#define VITAE_WS64_PRODUCERS 4
#endif
    constexpr int BM = 64, BN = 64, NWC = 4, NWP = VITAE_WS64_PRODUCERS;
    constexpr int A_T = BM * BK * 2, B_T = BN * BK * 2, STG = A_T + (W2 ? 2 : 1) * B_T;
    constexpr int PA = pieces<BM, A_KC, NWP>(), PB = pieces<BN, B_KC, NWP>(), PT = PA + (W2 ? 2 : 1) * PB;
    static_assert(!W2 || (A_KC && B_KC && !RS), "two weight planes: the forward form");
    static_assert(S >= 3 && S <= 5 && (S - 2) * PT <= 63, "stage count");
    const int T = h.tiles_m * h.tiles_n;
    const int xq = T >> 3, xr = T & 7, xcd = bid & 7;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    if ((bid >> 3) >= xq + (xcd < xr ? 1 : 0)) return;
    const int tn = (h.xcd_m & 1) ? lin % h.tiles_n : lin / h.tiles_m;
    const int tm = (h.xcd_m & 1) ? lin / h.tiles_n : lin % h.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * h.kps;
    const int nk = (min(h.K, kbeg + h.kps) - kbeg) / BK;         // >= 2 (launcher)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // tools/ws64_phase_probe.py: s_memtime stamps of consumer wave 0 (slots 0-7) and producer wave 4 (8-15) of every workgroup,
    // 16 per workgroup indexed by its position in the launch (blockIdx.x + gridDim.x * blockIdx.z)
    auto stamp = [&](int i) {
        if ((h.xcd_m & 2) && (threadIdx.x & 255) == 0) p.dbg[((long)blockIdx.z * gridDim.x + blockIdx.x) * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    if (wave >= NWC) {
        // ---------------- producers ----------------
        const int pw = wave - NWC;
        stamp(8);
        auto issue_tile = [&](int t, int stage) {
            unsigned char* dst = smem + stage * STG;
            const int k0 = kbeg + t * BK;
#pragma unroll
            for (int j = 0; j < PA; ++j) dma_piece<BM, A_KC, NWP>(h.A, h.lda, h.M, m0, k0, dst, pw, lane, j);
#pragma unroll
            for (int j = 0; j < PB; ++j) dma_piece<BN, B_KC, NWP>(h.B, h.ldb, h.N, n0, k0, dst + A_T, pw, lane, j);
            if constexpr (W2) {
#pragma unroll
                for (int j = 0; j < PB; ++j) dma_piece<BN, B_KC, NWP>(h.B2, h.ldb, h.N, n0, k0, dst + A_T + B_T, pw, lane, j);
            }
        };
        auto wait_tiles = [&](int fly) {
            if (S >= 5 && fly >= 3) wait_vmcnt<(S >= 5 ? 3 : 1) * PT>();
            else if (S >= 4 && fly >= 2) wait_vmcnt<(S >= 4 ? 2 : 1) * PT>();
            else if (fly >= 1) wait_vmcnt<PT>();
            else wait_vmcnt<0>();
        };
        const int npre = min(S - 1, nk);
        for (int t = 0; t < npre; ++t) issue_tile(t, t);
        stamp(9);                                                        // prologue issued
        wait_tiles(npre - 1);
        stamp(10);                                                       // first k-tile landed
        barrier();                                                       // B_0
        int stage = npre % S;
#pragma unroll 1
        for (int t = 0; t + 1 < nk; ++t) {
            if (t + S - 1 < nk) {
                issue_tile(t + S - 1, stage);
                stage = stage + 1 == S ? 0 : stage + 1;
            }
            wait_tiles(min(t + S - 1, nk - 1) - (t + 1));
            barrier();                                                   // B_{t+1}
        }
        stamp(11);                                                       // last k-tile handed over
        return;
    }
    // ---------------- consumers ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[h][i] = 0.f;
    const bool rowsum = RS && p.a_rowsum != nullptr && tn == 0 && wn == 0;   // wave-uniform; every k-split adds its share
    f32x16 accx;
#pragma unroll
    for (int i = 0; i < 16; ++i) accx[i] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
    bf16x8 fa[BK / 16], fb[BK / 16], fb2[W2 ? BK / 16 : 1];
    constexpr int RK = (A_KC ? 1 : 2) + (W2 ? 2 : 1) * (B_KC ? 1 : 2);
    auto rd = [&](const unsigned char* TA, auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        fa[kk] = frag_asm<BM, A_KC>(TA, wm * 32, kk, lane);
        fb[kk] = frag_asm<BN, B_KC>(TA + A_T, wn * 32, kk, lane);
        if constexpr (W2) fb2[kk] = frag_asm<BN, B_KC>(TA + A_T + B_T, wn * 32, kk, lane);
    };
    auto mm = [&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        frag_tie(fa[kk]); frag_tie(fb[kk]);
        if constexpr (W2) {                                              // the small term first
            frag_tie(fb2[kk]);
            acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb2[kk], acc[kk & 1], 0, 0, 0);
        }
        acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[kk], acc[kk & 1], 0, 0, 0);
        if constexpr (RS) {
            if (rowsum) accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], ones, accx, 0, 0, 0);
        }
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    // the bias of this wave's four-column groups, fetched now (after the loop it would cost an exposed memory latency)
    const int nq = n0 + wn * 32 + 4 * (lane % 8);
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nq < p.N) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nq);
    stamp(0);
    barrier();                                                           // B_0
    stamp(1);
    rd(smem, K0{}); rd(smem, K1{});
    int stage = 0;
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const unsigned char* TA = smem + stage * STG;
        __builtin_amdgcn_sched_barrier(0);
        rd(TA, K2{}); rd(TA, K3{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * RK) : "memory");     // slices 0-1 are in registers
        __builtin_amdgcn_sched_barrier(0);
        mm(K0{}); mm(K1{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // every read of tile t has retired
        if (t + 1 < nk) {
            barrier();                                                   // B_{t+1}
            stage = stage + 1 == S ? 0 : stage + 1;
            const unsigned char* TN = smem + stage * STG;
            rd(TN, K0{}); rd(TN, K1{});
            __builtin_amdgcn_sched_barrier(0);
        }
        mm(K2{}); mm(K3{});
    }
    if (RS && rowsum && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + crow(r, hi);
            if (m < p.M) atomicAdd(p.a_rowsum + m, accx[r]);
        }
    }
    stamp(2);                                                            // k-loop done
    f32x16 accs[1][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) accs[0][0][i] = acc[0][i] + acc[1][i];
    // the epilogue's LDS (4 KB per wave, the split-K flag, the norm shares) ALIASES the stages: every consumer is past its last
    // fragment read at this barrier, every DMA has landed, and the producers are gone — so the whole LDS budget is prefetch depth
    __syncthreads();
    float* Tw = reinterpret_cast<float*>(smem) + wave * 1024;
    if (p.splits > 1) {
        // (the producers have left: these barriers count the four consumer waves only)
        const int tile = p.tile0 + tm * p.tiles_n + tn;
        float* part = p.ws + VITAE_GLDS_TICKETS + (long)tile * p.splits * (BM * BN);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, p.splits * (BM * BN * 4), 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int toff = (int)threadIdx.x * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {accs[0][0][4 * g], accs[0][0][4 * g + 1], accs[0][0][4 * g + 2], accs[0][0][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (zid * 4 + g) * (256 * 16) + toff, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem + 4 * 4096);
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(reinterpret_cast<int*>(p.ws) + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        stamp(3);                                                        // partials parked, ticket taken
        if (*flag != p.splits - 1) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) accs[0][0][i] = 0.f;
#pragma unroll 1
        for (int sp0 = 0; sp0 < p.splits; sp0 += 4) {
            f32x4 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sp = min(sp0 + u, p.splits - 1);               // (clamped: a repeated load, never added)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    v[u][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (sp * 4 + g) * (256 * 16) + toff, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sp0 + u < p.splits) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) accs[0][0][4 * g + e] += v[u][g][e];
                }
        }
        if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<int*>(p.ws) + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(4);                                                            // epilogue starts
    bt_park_quadrant<1, 1, 1>(accs, lane, Tw);
    __builtin_amdgcn_wave_barrier();
    float sqs = 0.f;
    { f32x4 cz = {0.f, 0.f, 0.f, 0.f}; bt_wave_epilogue<1, 1>(p, bt_epilogue_kind(p), m0 + wm * 32, n0 + wn * 32, Tw, lane, sqs, bias4, cz, false); }
    stamp(5);                                                            // stores issued
    if (p.sqacc) {                                                       // one atomic per workgroup (see bt_tail)
        sqs = wave_sum(sqs);
        float* red = reinterpret_cast<float*>(smem + 4 * 4096 + 64);
        if (lane == 0) red[wave] = sqs;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sq_slot(p), (double)((red[0] + red[1]) + (red[2] + red[3])));
    }
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(6); }      // stores drained
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * (4 + VITAE_WS64_PRODUCERS), 2) void gemm_ws64_kernel(WS_HOT_PARAMS, const GArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WS64_SMEM];      // the ONLY LDS object
    const WsHot h = WS_HOT_FROM_PARAMS;
    gemm_ws64_body<A_KC, B_KC, VITAE_WS64_STAGES, false>(p, h, blockIdx.x, blockIdx.z, smem);
}

constexpr int WS64W2_STAGES = 3;            // [A | B hi | B lo] = 24 KB per stage: three stages = 72 KB, two workgroups per CU
__global__ __launch_bounds__(64 * (4 + VITAE_WS64_PRODUCERS), 2) void gemm_ws64_w2_kernel(WS_HOT_PARAMS, const GArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WS64W2_STAGES * 24576];      // the ONLY LDS object
    gemm_ws64_body<true, true, WS64W2_STAGES, false, true>(p, WS_HOT_FROM_PARAMS, blockIdx.x, blockIdx.z, smem);
}

// p: a complete forward-form descriptor (both operands k-contiguous, vec_epi set, p.B2 = the lo plane, p.splits k-ranges)
int ws64_w2_launch(GArgs p, hipStream_t st) {
    if (!p.vec_epi || !p.B2 || p.a_rowsum || (p.K % BK) || p.splits < 1) return VITAE_ERR_UNSUPPORTED_SHAPE;
    p.k_per_split = cdiv(cdiv(p.K, p.splits), BK) * BK;
    p.splits = cdiv(p.K, p.k_per_split);
    if (p.K - (p.splits - 1) * p.k_per_split < 2 * BK || p.k_per_split < 2 * BK) return VITAE_ERR_UNSUPPORTED_SHAPE;
    p.tiles_m = cdiv(p.M, 64); p.tiles_n = cdiv(p.N, 64); p.tile0 = 0;
    if (p.splits > 1 && (!p.ws || (long)p.tiles_m * p.tiles_n > VITAE_GLDS_TICKETS || p.epi == VITAE_EPI_GELU)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const dim3 grid(8 * cdiv((long)p.tiles_m * p.tiles_n, 8), 1, p.splits), block(64 * (4 + VITAE_WS64_PRODUCERS));
    hipLaunchKernelGGL(gemm_ws64_w2_kernel, grid, block, 0, st, WS_HOT_ARGS(p), p);
    return vitae_launch_status();
}

// ---- fp32x3 on the same structure (round 4) ------------------------------------------------------------------------------------
// The fp32-grade fast mode multiplies fp32 operands as bf16 hi + lo pairs (hi.hi + hi.lo + lo.hi, fp32 accumulate).  Here the
// PRODUCER waves — idle VALUs — do the split: they load the fp32 operand tiles with ordinary 16-byte loads (two k-tiles ahead,
// in registers), form hi = bf16(x), lo = bf16(x - hi) and write both images into LDS in exactly the layout the LDS-DMA leaves
// (same source permutation, same fragment reads).  Nothing upstream changes: every producer of an fp32 activation stays as it is.
// A stage = [A hi | B hi | A lo | B lo] = 32 KB; the look-ahead lives in registers, so TWO stages suffice (64 KB: two workgroups
// per CU).  The reduction length is any multiple of 4 (token counts are not padded on the fp32 side): the tail is zero-filled.
#ifndef VITAE_X3_PRODUCERS
#define VITAE_X3_PRODUCERS 4      // producer waves of the fp32x3 workgroup (8: half the conversion work per wave, one workgroup per CU)
#endif
// The loads are UNCONDITIONAL (clamped addresses) and the zero-fill of the reduction's tail is applied when the values are
// converted: a load behind a condition — or a select right behind it — makes hipcc wait for every load in flight (vmcnt(0)), which
// exposed a full memory latency per k-tile (0.78 us per tile instead of 0.35).
// ... and they are INLINE ASM: hipcc's own vmcnt bookkeeping across the loop's back edge waited for the younger register set too
// (vmcnt(7) .. vmcnt(0) in front of the older set's conversion), so a tile's loads had half a k-tile of flight time.  The caller
// counts: 8 loads per wave and tile, `wait_vmcnt<8>` in front of a conversion while the other set is in flight, then x3_tie.
__device__ __forceinline__ f32x4 x3_ld(const float* ptr) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
    return r;
}
__device__ __forceinline__ void x3_tie(f32x4& v) { asm volatile("" : "+v"(v)); }

template <int ROWS, bool KC>
__device__ __forceinline__ void x3_load_piece(const float* __restrict__ P, long ld, int rows, int K, int r0, int k0, int pw, int lane, int j,
                                              f32x4 (&v)[2]) {
    constexpr int LINE_CH = KC ? BK / 8 : ROWS / 8, LPI = 64 / LINE_CH, NI = pieces<ROWS, KC, VITAE_X3_PRODUCERS>();
    const int inst = pw * NI + j;
    const int line = inst * LPI + lane / LINE_CH;
    const int slot = lane % LINE_CH;
    const int chunk = slot ^ swz<KC, LINE_CH>(line);
    if (KC) {
        const int gr = min(r0 + line, rows - 1);
        const int k = k0 + chunk * 8;
        const float* src = P + (long)gr * ld;
        v[0] = x3_ld(src + min(k, K - 4));
        v[1] = x3_ld(src + min(k + 4, K - 4));
    } else {
        // rows are a multiple of 4 only (fp32 activations are not padded): each 4-row group clamped on its own — groups past
        // the operand read a valid group and are never stored
        const int g0 = min(r0 + chunk * 8, rows - 4), g1 = min(r0 + chunk * 8 + 4, rows - 4);
        const float* src = P + (long)min(k0 + line, K - 1) * ld;
        v[0] = x3_ld(src + g0);
        v[1] = x3_ld(src + g1);
    }
}
template <int ROWS, bool KC>
__device__ __forceinline__ void x3_store_piece(const f32x4 (&v)[2], int K, int k0, bool full, unsigned char* hi_img, unsigned char* lo_img, int pw, int lane, int j) {
    constexpr int LINE_CH = KC ? BK / 8 : ROWS / 8, LPI = 64 / LINE_CH, NI = pieces<ROWS, KC, VITAE_X3_PRODUCERS>();
    const int inst = pw * NI + j;
    const int line = inst * LPI + lane / LINE_CH;
    const int chunk = (lane % LINE_CH) ^ swz<KC, LINE_CH>(line);
    // which of the two 4-element groups lie inside the reduction (K % 4 == 0)
    const int k = KC ? k0 + chunk * 8 : k0 + line;
    const bool in0 = k < K, in1 = KC ? k + 4 < K : k < K;
    // 24 VALU operations per 8 elements (the naive form compiled to 40, and the producers' conversion was the k-loop's bound):
    // pairs converted once (v_cvt_pk_bf16_f32), the two hi values of a pair taken back out of the packed word by a shift / a mask
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x0 = v[q >> 1][2 * (q & 1)], x1 = v[q >> 1][2 * (q & 1) + 1];
        if (!full) {
            const bool in = (q >> 1) ? in1 : in0;
            x0 = in ? x0 : 0.f; x1 = in ? x1 : 0.f;
        }
        const unsigned ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
        const float h0 = __builtin_bit_cast(float, ph << 16), h1 = __builtin_bit_cast(float, ph & 0xffff0000u);
        h[q] = ph;
        l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0 - h0, x1 - h1}, bf16x2));
    }
    *reinterpret_cast<u32x4*>(hi_img + inst * 1024 + lane * 16) = h;
    *reinterpret_cast<u32x4*>(lo_img + inst * 1024 + lane * 16) = l;
}

template <bool A_KC, bool B_KC, bool RS>
__device__ __forceinline__ void gemm_wsx3_body(const GArgs& p, const WsHot& h, const int bid, const int zid, unsigned char* smem) {
    constexpr int BM = 64, BN = 64, NWC = 4;
    constexpr int IMG = 64 * BK * 2, STG = 4 * IMG;                      // [A hi | B hi | A lo | B lo]
    constexpr int PA = pieces<BM, A_KC, VITAE_X3_PRODUCERS>(), PB = pieces<BN, B_KC, VITAE_X3_PRODUCERS>();
    const float* Af = reinterpret_cast<const float*>(h.A);
    const float* Bf = reinterpret_cast<const float*>(h.B);
    const int T = h.tiles_m * h.tiles_n;
    const int xq = T >> 3, xr = T & 7, xcd = bid & 7;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    if ((bid >> 3) >= xq + (xcd < xr ? 1 : 0)) return;
    const int tn = (h.xcd_m & 1) ? lin % h.tiles_n : lin / h.tiles_m;
    const int tm = (h.xcd_m & 1) ? lin / h.tiles_n : lin % h.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zid * h.kps;
    const int kend = min(h.K, kbeg + h.kps);
    const int nk = (kend - kbeg + BK - 1) / BK;                          // >= 1; the last tile may be partly past kend (zero-filled)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if (wave >= NWC) {
        // ---------------- producers: fp32 loads two tiles ahead, split, LDS writes ----------------
        const int pw = wave - NWC;
        f32x4 ra[2][PA][2], rb[2][PB][2];
        auto load = [&](int t, auto set_c) {
            constexpr int SET = decltype(set_c)::value;
            const int k0 = kbeg + t * BK;
#pragma unroll
            for (int j = 0; j < PA; ++j) x3_load_piece<BM, A_KC>(Af, h.lda, h.M, kend, m0, k0, pw, lane, j, ra[SET][j]);
#pragma unroll
            for (int j = 0; j < PB; ++j) x3_load_piece<BN, B_KC>(Bf, h.ldb, h.N, kend, n0, k0, pw, lane, j, rb[SET][j]);
        };
        auto store = [&](int t, int stage, auto set_c, bool other_in_flight) {
            constexpr int SET = decltype(set_c)::value;
            unsigned char* st = smem + stage * STG;
            const int k0 = kbeg + t * BK;
            if (other_in_flight) wait_vmcnt<PA * 2 + PB * 2>(); else wait_vmcnt<0>();
#pragma unroll
            for (int j = 0; j < PA; ++j) { x3_tie(ra[SET][j][0]); x3_tie(ra[SET][j][1]); }
#pragma unroll
            for (int j = 0; j < PB; ++j) { x3_tie(rb[SET][j][0]); x3_tie(rb[SET][j][1]); }
            const bool full = k0 + BK <= kend;          // wave-uniform: only the last tile of a ragged reduction masks
#pragma unroll
            for (int j = 0; j < PA; ++j) x3_store_piece<BM, A_KC>(ra[SET][j], kend, k0, full, st, st + 2 * IMG, pw, lane, j);
#pragma unroll
            for (int j = 0; j < PB; ++j) x3_store_piece<BN, B_KC>(rb[SET][j], kend, k0, full, st + IMG, st + 3 * IMG, pw, lane, j);
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        load(0, S0{});
        if (nk > 1) load(1, S1{});
        // tile t goes into stage t & 1 after B_{t-1} (every read of tile t - 2 retired before it); B_t publishes it
#pragma unroll 1
        for (int t = 0; t < nk; t += 2) {
            store(t, 0, S0{}, t + 1 < nk);
            if (t + 2 < nk) load(t + 2, S0{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            barrier();                                                   // B_t
            if (t + 1 < nk) {
                store(t + 1, 1, S1{}, t + 2 < nk);
                if (t + 3 < nk) load(t + 3, S1{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                barrier();                                               // B_{t+1}
            }
        }
        return;
    }
    // ---------------- consumers: hi / lo fragments, three MFMAs per pair ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[h][i] = 0.f;
    const bool rowsum = RS && p.a_rowsum != nullptr && tn == 0 && wn == 0;
    f32x16 accx;
#pragma unroll
    for (int i = 0; i < 16; ++i) accx[i] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
    bf16x8 fah[BK / 16], fal[BK / 16], fbh[BK / 16], fbl[BK / 16];
    constexpr int RK = 2 * ((A_KC ? 1 : 2) + (B_KC ? 1 : 2));
    auto rd = [&](const unsigned char* ST, auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        fah[kk] = frag_asm<BM, A_KC>(ST, wm * 32, kk, lane);
        fbh[kk] = frag_asm<BN, B_KC>(ST + IMG, wn * 32, kk, lane);
        fal[kk] = frag_asm<BM, A_KC>(ST + 2 * IMG, wm * 32, kk, lane);
        fbl[kk] = frag_asm<BN, B_KC>(ST + 3 * IMG, wn * 32, kk, lane);
    };
    auto mm = [&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        frag_tie(fah[kk]); frag_tie(fbh[kk]); frag_tie(fal[kk]); frag_tie(fbl[kk]);
        // the two cross terms first: they are 2^-8 of the main term
        acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[kk], fbh[kk], acc[kk & 1], 0, 0, 0);
        acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[kk], fbl[kk], acc[kk & 1], 0, 0, 0);
        acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[kk], fbh[kk], acc[kk & 1], 0, 0, 0);
        if constexpr (RS) {
            if (rowsum) {
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[kk], ones, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[kk], ones, accx, 0, 0, 0);
            }
        }
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    const int nq = n0 + wn * 32 + 4 * (lane % 8);
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nq < p.N) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nq);
    barrier();                                                           // B_0
    rd(smem, K0{}); rd(smem, K1{});
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const unsigned char* ST = smem + (t & 1) * STG;
        __builtin_amdgcn_sched_barrier(0);
        rd(ST, K2{}); rd(ST, K3{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * RK > 15 ? 15 : 2 * RK) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        mm(K0{}); mm(K1{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // every read of tile t has retired
        if (t + 1 < nk) {
            barrier();                                                   // B_{t+1}
            const unsigned char* SN = smem + ((t + 1) & 1) * STG;
            rd(SN, K0{}); rd(SN, K1{});
            __builtin_amdgcn_sched_barrier(0);
        }
        mm(K2{}); mm(K3{});
    }
    if (RS && rowsum && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + crow(r, hi);
            if (m < p.M) atomicAdd(p.a_rowsum + m, accx[r]);
        }
    }
    f32x16 accs[1][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) accs[0][0][i] = acc[0][i] + acc[1][i];
    __syncthreads();                                                     // (consumers only: the producers have left) stages are free
    float* Tw = reinterpret_cast<float*>(smem) + wave * 1024;
    if (p.splits > 1) {
        const int tile = p.tile0 + tm * p.tiles_n + tn;
        float* part = p.ws + VITAE_GLDS_TICKETS + (long)tile * p.splits * (BM * BN);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, p.splits * (BM * BN * 4), 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int toff = (int)threadIdx.x * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {accs[0][0][4 * g], accs[0][0][4 * g + 1], accs[0][0][4 * g + 2], accs[0][0][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (zid * 4 + g) * (256 * 16) + toff, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem + 4 * 4096);
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(reinterpret_cast<int*>(p.ws) + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != p.splits - 1) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) accs[0][0][i] = 0.f;
#pragma unroll 1
        for (int sp0 = 0; sp0 < p.splits; sp0 += 4) {
            f32x4 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sp = min(sp0 + u, p.splits - 1);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    v[u][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (sp * 4 + g) * (256 * 16) + toff, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sp0 + u < p.splits) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) accs[0][0][4 * g + e] += v[u][g][e];
                }
        }
        if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<int*>(p.ws) + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bt_park_quadrant<1, 1, 1>(accs, lane, Tw);
    __builtin_amdgcn_wave_barrier();
    float sqs = 0.f;
    { f32x4 cz = {0.f, 0.f, 0.f, 0.f}; bt_wave_epilogue<1, 1>(p, bt_epilogue_kind(p), m0 + wm * 32, n0 + wn * 32, Tw, lane, sqs, bias4, cz, false); }
    if (p.sqacc) {
        sqs = wave_sum(sqs);
        float* red = reinterpret_cast<float*>(smem + 4 * 4096 + 64);
        if (lane == 0) red[wave] = sqs;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sq_slot(p), (double)((red[0] + red[1]) + (red[2] + red[3])));
    }
}

template <bool A_KC, bool B_KC, bool RS>
__global__ __launch_bounds__(64 * (4 + VITAE_X3_PRODUCERS), VITAE_X3_PRODUCERS == 4 ? 4 : 3) void gemm_wsx3_kernel(WS_HOT_PARAMS, const GArgs p) {      // (4 producers: 4 waves per SIMD = <= 128 VGPRs, two workgroups per CU — the row-sum variant took 132)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 4 * 8192];      // the ONLY LDS object
    gemm_wsx3_body<A_KC, B_KC, RS>(p, WS_HOT_FROM_PARAMS, blockIdx.x, blockIdx.z, smem);
}

// resident workgroup slots of the chip for this kernel (the split-K rule fills about that many)
int wsx3_slots() { return VITAE_X3_PRODUCERS == 4 ? 512 : 256; }

// p: a complete descriptor with FLOAT operands behind p.A / p.B (vec_epi set); p.splits k-ranges, multiples of 64 except the last
int wsx3_launch(GArgs p, int a_kc, int b_kc, hipStream_t st) {
    if (!p.vec_epi || (p.K & 3) || p.splits < 1 || p.C16) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (!a_kc && b_kc) return VITAE_ERR_UNSUPPORTED_SHAPE;
    p.k_per_split = cdiv(cdiv(p.K, p.splits), BK) * BK;
    p.splits = cdiv(p.K, p.k_per_split);
    p.tiles_m = cdiv(p.M, 64); p.tiles_n = cdiv(p.N, 64); p.tile0 = 0;
    if (p.splits > 1 && (!p.ws || (long)p.tiles_m * p.tiles_n > VITAE_GLDS_TICKETS || p.epi == VITAE_EPI_GELU)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (p.a_rowsum && (a_kc || b_kc)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const dim3 grid(8 * cdiv((long)p.tiles_m * p.tiles_n, 8), 1, p.splits), block(64 * (4 + VITAE_X3_PRODUCERS));
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_wsx3_kernel<true, true, false>), grid, block, 0, st, WS_HOT_ARGS(p), p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_wsx3_kernel<true, false, false>), grid, block, 0, st, WS_HOT_ARGS(p), p);
    else if (p.a_rowsum) hipLaunchKernelGGL((gemm_wsx3_kernel<false, false, true>), grid, block, 0, st, WS_HOT_ARGS(p), p);
    else hipLaunchKernelGGL((gemm_wsx3_kernel<false, false, false>), grid, block, 0, st, WS_HOT_ARGS(p), p);
    return vitae_launch_status();
}

// Backward of one Linear as ONE launch of such workgroups: the first nb1 * p1.splits blocks compute the input gradient
// dx = epi(dy W) (A = dy k-contiguous, B = W row-contiguous; its long reduction cut into p1.splits), the rest the weight gradient
// dW (+)= dy^T x (both row-contiguous; RS: + colsum(dy)).
// The leading scalars are what either half needs before its first DMA (WsHot), in 14 dwords so that all of them are preloaded: the
// halves of one Linear's backward SHARE dy (A of both), its leading dimension, the leading dimension of W / x, the width K of the layer
// (N of both), and the weight gradient's rows are the input gradient's reduction length (ws64_pair_launch checks exactly that).
//   packed = nb1 | p1.splits << 20 | p1.xcd_m << 24 | p2.xcd_m << 25 | (dbg set) << 26
template <bool RS>
__global__ __launch_bounds__(64 * (4 + VITAE_WS64_PRODUCERS), 2) void gemm_ws64_pair_kernel(const __bf16* A, const __bf16* B1, const __bf16* B2, int lda, int ldb,
                                                                                           int M1, int N1, int K1, int K2, int kps1, int packed,
                                                                                           const GArgs p1, const GArgs p2) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WS64_SMEM];
    const int nb1 = packed & 0xfffff, splits1 = (packed >> 20) & 15, dbg2 = (packed >> 25) & 2;
    if ((int)blockIdx.x < nb1 * splits1) {
        const WsHot h{A, B1, nullptr, lda, ldb, M1, N1, K1, (M1 + 63) >> 6, (N1 + 63) >> 6, ((packed >> 24) & 1) | dbg2, kps1};
        gemm_ws64_body<true, false, VITAE_WS64_STAGES, false>(p1, h, blockIdx.x % nb1, blockIdx.x / nb1, smem);
    } else {
        const WsHot h{A, B2, nullptr, lda, ldb, K1, N1, K2, (K1 + 63) >> 6, (N1 + 63) >> 6, ((packed >> 25) & 1) | dbg2, K2};
        gemm_ws64_body<false, false, VITAE_WS64_STAGES, RS>(p2, h, blockIdx.x - nb1 * splits1, 0, smem);
    }
}

// p1 / p2: complete descriptors of the two halves (vec_epi set, K % 64 == 0); p1.splits k-ranges of >= 2 k-tiles each
int ws64_pair_launch(GArgs p1, GArgs p2, hipStream_t st) {
    if (!p1.vec_epi || !p2.vec_epi || (p1.K % BK) || (p2.K % BK) || p2.K < 2 * BK || p1.a_rowsum) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (p1.splits < 1) p1.splits = 1;
    p1.k_per_split = cdiv(cdiv(p1.K, p1.splits), BK) * BK;
    p1.splits = cdiv(p1.K, p1.k_per_split);
    if (p1.K - (p1.splits - 1) * p1.k_per_split < 2 * BK || p1.k_per_split < 2 * BK) return VITAE_ERR_UNSUPPORTED_SHAPE;
    p1.tiles_m = cdiv(p1.M, 64); p1.tiles_n = cdiv(p1.N, 64); p1.tile0 = 0;
    p2.tiles_m = cdiv(p2.M, 64); p2.tiles_n = cdiv(p2.N, 64); p2.tile0 = 0;
    p2.k_per_split = p2.K; p2.splits = 1;
    if (p1.splits > 1 && (!p1.ws || (long)p1.tiles_m * p1.tiles_n > VITAE_GLDS_TICKETS)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const int nb1 = 8 * cdiv((long)p1.tiles_m * p1.tiles_n, 8), nb2 = 8 * cdiv((long)p2.tiles_m * p2.tiles_n, 8);
    const dim3 grid(nb1 * p1.splits + nb2), block(64 * (4 + VITAE_WS64_PRODUCERS));
    // what the kernel's leading scalars assume (the two halves of ONE Linear's backward: gemm_ws64_pair_kernel)
    if (p1.A != p2.A || p1.lda != p2.lda || p1.ldb != p2.ldb || p2.M != p1.K || p2.N != p1.N || nb1 >= (1 << 20) || p1.splits > 15 ||
        p1.lda >= (1L << 31) || p1.ldb >= (1L << 31)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const int packed = nb1 | p1.splits << 20 | (p1.xcd_m & 1) << 24 | (p2.xcd_m & 1) << 25 | ((p1.dbg || p2.dbg) ? 1 << 26 : 0);
    if (p2.a_rowsum) hipLaunchKernelGGL((gemm_ws64_pair_kernel<true>), grid, block, 0, st, p1.A, p1.B, p2.B, (int)p1.lda, (int)p1.ldb, p1.M, p1.N, p1.K,
                                        p2.K, p1.k_per_split, packed, p1, p2);
    else hipLaunchKernelGGL((gemm_ws64_pair_kernel<false>), grid, block, 0, st, p1.A, p1.B, p2.B, (int)p1.lda, (int)p1.ldb, p1.M, p1.N, p1.K,
                            p2.K, p1.k_per_split, packed, p1, p2);
    return vitae_launch_status();
}

// ---- the paired launch as PERSISTENT workgroups with the tiles pipelined ACROSS each other (round 6) ------------
// tools/ws64_phase_probe.py on gemm_ws64_pair_kernel (batch 4, encoder fc2: 912 tiles on 512 slots): a workgroup lives ~10.5 k clocks
// of which the k-loop is 4.0-5.8 k — 1.4 k of prologue issue + 0.5 k until the first k-tile lands before the first MFMA, 1.5-2.4 k of
// epilogue and 1.6 k until its stores have drained behind it, and the 400 second-round workgroups pay all of that again behind the
// first round.  Here a launch is at most one workgroup per slot and every workgroup walks a queue of tiles (virtual ids bid, bid + G,
// ...: input-gradient tiles first, then weight-gradient tiles; G a multiple of 8 so a workgroup stays on the XCD chunk of its tiles):
//   * the producer waves see ONE flat stream of k-tiles — behind the last k-tile of a tile they go straight on with the first ones
//     of the next tile, so those have landed when the consumers come out of their epilogue;
//   * the epilogue is wave-private: the split-K hand-over takes one ticket per WAVE (a wave's 32 x 32 quadrant of the partials is
//     its own), the gradient norm's share is carried across the tiles in a register and leaves once per workgroup.  It parks its
//     fragments in the stage of the tile's LAST k-tile — the one stage the producers cannot restage before the consumers reach the
//     next tile's first barrier — behind one more workgroup barrier per tile (E: every consumer's reads of that stage have retired);
//     both roles count nk + 1 barriers per tile;
//   * DMA addresses: one 32-bit offset per lane and piece, computed once per tile; the k-tile enters through the instruction's
//     scalar offset (the stand-alone kernel recomputed ~40 instructions of address arithmetic per piece).
// What the first builds taught (tools/pair_bench.py, tools/ws64_phase_probe.py with VITAE_WS64Q=1):
//   * two by-value descriptors (2 x 58 dwords) stayed live in SGPRs across the tile loop: 170-180 spilled into VGPR lanes, ~300
//     v_readlane on every epilogue path, the launch 20 % slower than one tile per workgroup.  Now ONE kernel-argument struct is read
//     through the kernarg segment pointer, the epilogue's fields re-read per tile (the pointer is laundered through an empty asm);
//   * a select between fields of the two structs is canonicalised into a select of ADDRESSES, and hipcc then copied both structs to
//     scratch (488 bytes, 172 VGPRs): everything the tile decoding and the producers need sits in a header of plain values (QHead),
//     loaded once at entry in one batch — a second, dependent batch of scalar loads (other cache lines of the segment) cost the
//     prologue another ~1 k clocks;
//   * three stages (two k-tiles in flight) ran the k-loop at 695 clocks per k-tile against 462 with four: the DMA latency needs three
//     in flight.  Four stages of 16 KB with the epilogue aliased as above: 64 KB, two workgroups per CU as before.
#ifndef VITAE_WSQ_STAGES
#define VITAE_WSQ_STAGES 4
#endif
constexpr int WSQ_S = VITAE_WSQ_STAGES, WSQ_STG = 16384, WSQ_MISC = WSQ_S * WSQ_STG, WSQ_SMEM = WSQ_MISC + 64;
static_assert(2 * WSQ_SMEM <= 160 * 1024, "two workgroups per CU");

struct QTile { int kind, m0, n0, tm, tn, zid, kbeg, nk; };       // kind 0: a tile of the input gradient (p1), 1: of the weight gradient (p2)
struct QHead {
    int nb1, nv, nd, tm1, tn1, xm1, kps1, K1, tm2, tn2, xm2, K2;                 // tile decoding
    // ... without a division: x / d = umulhi(x, floor(2^32 / d) + 1), exact for x * d < 2^32 (all of these are below 2^16).  The
    // compiler's expansion of four scalar divisions by run-time values was ~450 instructions = ~2000 clocks in front of the first
    // DMA piece of BOTH roles (tools/ws64_phase_probe.py, second build).
    unsigned mg_nb1, dv1, mg1, dv2, mg2;                                          // dv: tiles_n if xcd_m else tiles_m
    const __bf16 *A1, *B1; int lda1, ldb1, M1, N1;                               // input gradient: dy (k-contiguous), W (row-contiguous)
    const __bf16 *A2, *B2; int lda2, ldb2, M2, N2;                               // weight gradient: dy, x (both row-contiguous)
    long long* dbg; double* sqacc; int sq_mask, sq_stride;
};
struct PairArgs { QHead h; GArgs p1, p2; };
typedef __attribute__((address_space(4))) const PairArgs* wsq_kargs_t;
__device__ __forceinline__ const PairArgs* wsq_kargs() {
    wsq_kargs_t k = (wsq_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));                  // (whatever is loaded through the result is loaded HERE, not at kernel entry)
    return (const PairArgs*)k;
}
// virtual id -> tile (wave-uniform; H: the header's values); ok = false: a padding id of the XCD chunk map
#define WSQ_DECODE(v, q, ok)                                                                                               \
    do {                                                                                                                   \
        const bool w_ = (v) >= h_nd;                                                                                       \
        const int zid_ = w_ ? 0 : (int)__umulhi((unsigned)(v), h_mg_nb1);                                                  \
        const int bid_ = w_ ? (v) - h_nd : (v) - zid_ * h_nb1;                                                             \
        const int tiles_m_ = w_ ? h_tm2 : h_tm1, tiles_n_ = w_ ? h_tn2 : h_tn1, xcd_m_ = w_ ? h_xm2 : h_xm1;               \
        const unsigned dv_ = w_ ? h_dv2 : h_dv1, mg_ = w_ ? h_mg2 : h_mg1;                                                 \
        const int T_ = tiles_m_ * tiles_n_;                                                                                \
        const int xq_ = T_ >> 3, xr_ = T_ & 7, xcd_ = bid_ & 7;                                                            \
        (ok) = (bid_ >> 3) < xq_ + (xcd_ < xr_ ? 1 : 0);                                                                   \
        const int lin_ = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (bid_ >> 3);             \
        const int qa_ = (int)__umulhi((unsigned)lin_, mg_), qb_ = lin_ - qa_ * (int)dv_;                                   \
        (q).tn = xcd_m_ ? qb_ : qa_;                                                                                       \
        (q).tm = xcd_m_ ? qa_ : qb_;                                                                                       \
        (q).m0 = (q).tm * 64; (q).n0 = (q).tn * 64; (q).zid = zid_; (q).kind = w_ ? 1 : 0;                                 \
        const int kps_ = w_ ? h_K2 : h_kps1, K_ = w_ ? h_K2 : h_K1;                                                        \
        (q).kbeg = zid_ * kps_;                                                                                            \
        (q).nk = (min(K_, (q).kbeg + kps_) - (q).kbeg) >> 6;                                                               \
    } while (0)

// one tile on the four consumer waves: k-loop (one workgroup barrier per k-tile, crossed in the middle of a k-tile), barrier E,
// wave-private epilogue out of the last k-tile's stage
template <bool A_KC, bool RS>
__device__ __forceinline__ void wsq_consume(const GArgs& pk, const QTile& q, unsigned char* smem, int& cstage, float& sqs,
                                            const int wave, const int lane, long long* dbg) {
    // what the epilogue reads of the descriptor, loaded NOW (one batch of scalar loads that lands under the k-loop; pinned by empty
    // asms): left to the compiler the loads sat in front of their first use, an exposed scalar-memory round trip inside every epilogue
    GArgs p = pk;
#define WSQ_PIN(x) asm volatile("" : "+s"(x))
    WSQ_PIN(p.C); WSQ_PIN(p.ldc); WSQ_PIN(p.C16); WSQ_PIN(p.ldc16); WSQ_PIN(p.M); WSQ_PIN(p.N); WSQ_PIN(p.aux); WSQ_PIN(p.ldaux);
    WSQ_PIN(p.epi); WSQ_PIN(p.aux16); WSQ_PIN(p.auxd); WSQ_PIN(p.exact); WSQ_PIN(p.accumulate); WSQ_PIN(p.residual); WSQ_PIN(p.ldr);
    WSQ_PIN(p.out_colsum); WSQ_PIN(p.sqacc); WSQ_PIN(p.splits); WSQ_PIN(p.ws); WSQ_PIN(p.tile0); WSQ_PIN(p.tiles_n); WSQ_PIN(p.a_rowsum);
#undef WSQ_PIN
    constexpr bool B_KC = false;
    constexpr int BM = 64, BN = 64, A_T = BM * BK * 2;
    // tools/ws64_phase_probe.py (VITAE_WS64Q=1): the stamps of gemm_ws64_body, for the FIRST tile of every workgroup (dbg null after it)
    auto stamp = [&](int i) {
        if (dbg && threadIdx.x == 0) dbg[(long)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = q.m0, n0 = q.n0, nk = q.nk;
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[h][i] = 0.f;
    const bool rowsum = RS && p.a_rowsum != nullptr && q.tn == 0 && wn == 0;
    f32x16 accx;
#pragma unroll
    for (int i = 0; i < 16; ++i) accx[i] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
    bf16x8 fa[BK / 16], fb[BK / 16];
    constexpr int RK = (A_KC ? 1 : 2) + 2;
    auto rd = [&](const unsigned char* TA, auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        fa[kk] = frag_asm<BM, A_KC>(TA, wm * 32, kk, lane);
        fb[kk] = frag_asm<BN, B_KC>(TA + A_T, wn * 32, kk, lane);
    };
    auto mm = [&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        frag_tie(fa[kk]); frag_tie(fb[kk]);
        acc[kk & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[kk], acc[kk & 1], 0, 0, 0);
        if constexpr (RS) {
            if (rowsum) accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], ones, accx, 0, 0, 0);
        }
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    const f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};                           // (neither half of a backward has a bias: refused by the launcher)
    stamp(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (the previous tile's epilogue is done with its LDS region)
    barrier();                                                           // first k-tile of this tile has landed
    stamp(1);
    rd(smem + cstage * WSQ_STG, K0{}); rd(smem + cstage * WSQ_STG, K1{});
    int last = cstage;
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const unsigned char* TA = smem + cstage * WSQ_STG;
        __builtin_amdgcn_sched_barrier(0);
        rd(TA, K2{}); rd(TA, K3{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * RK) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        mm(K0{}); mm(K1{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // every read of this k-tile has retired
        last = cstage;
        cstage = cstage + 1 == WSQ_S ? 0 : cstage + 1;
        if (t + 1 < nk) {
            barrier();
            const unsigned char* TN = smem + cstage * WSQ_STG;
            rd(TN, K0{}); rd(TN, K1{});
            __builtin_amdgcn_sched_barrier(0);
        }
        mm(K2{}); mm(K3{});
    }
    barrier();                                                           // E: nobody reads the last k-tile's stage any more
    if (RS && rowsum && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + crow(r, hi);
            if (m < p.M) atomicAdd(p.a_rowsum + m, accx[r]);
        }
    }
    stamp(2);
    f32x16 accs[1][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) accs[0][0][i] = acc[0][i] + acc[1][i];
    float* Tw = reinterpret_cast<float*>(smem + last * WSQ_STG) + wave * 1024;    // this wave's 4 KB of the stage nobody restages before the next tile's first barrier
    if (p.splits > 1) {
        // split-K hand-over per WAVE: this wave's quadrant of the partial tile leaves as write-through stores, the wave takes the
        // ticket of (tile, quadrant), and the LAST wave to arrive for that quadrant sums all partials in split order
        const int tile = p.tile0 + q.tm * p.tiles_n + q.tn;
        float* part = p.ws + VITAE_GLDS_TICKETS + (long)tile * p.splits * (BM * BN);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, p.splits * (BM * BN * 4), 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int toff = (wave * 64 + lane) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {accs[0][0][4 * g], accs[0][0][4 * g + 1], accs[0][0][4 * g + 2], accs[0][0][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (q.zid * 4 + g) * (256 * 16) + toff, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int* ticket = reinterpret_cast<int*>(p.ws) + tile * 4 + wave;
        int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        stamp(3);
        if (tk != p.splits - 1) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) accs[0][0][i] = 0.f;
#pragma unroll 1
        for (int sp0 = 0; sp0 < p.splits; sp0 += 4) {
            f32x4 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sp = min(sp0 + u, p.splits - 1);               // (clamped: a repeated load, never added)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    v[u][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (sp * 4 + g) * (256 * 16) + toff, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sp0 + u < p.splits) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) accs[0][0][4 * g + e] += v[u][g][e];
                }
        }
        if (lane == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    stamp(4);
    bt_park_quadrant<1, 1, 1>(accs, lane, Tw);
    __builtin_amdgcn_wave_barrier();
    { f32x4 cz = {0.f, 0.f, 0.f, 0.f}; bt_wave_epilogue<1, 1>(p, bt_epilogue_kind(p), m0 + wm * 32, n0 + wn * 32, Tw, lane, sqs, bias4, cz, false); }
    __builtin_amdgcn_wave_barrier();
    stamp(5);
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(6); }
}

template <bool RS>
__global__ __launch_bounds__(512, 4) void gemm_ws64q_pair_kernel(const PairArgs unused_by_name) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WSQ_SMEM];
    constexpr int S = WSQ_S, PT = 4;                                      // DMA instructions per wave and k-tile: 2 for A + 2 for B
    // the header as VALUES that are not loads (each passed through an empty asm: zero instructions — as readfirstlane results they cost a
    // v_mov + v_readfirstlane + hazard nops each, ~450 clocks at entry): a struct copy went to scratch again, indexed through selected
    // addresses (184-200 bytes of private segment, the launch twice as slow)
    const QHead* hp_ = &((const PairArgs*)(wsq_kargs_t)__builtin_amdgcn_kernarg_segment_ptr())->h;
#define WSQ_VAL(T, name, field) T name = hp_->field; asm volatile("" : "+s"(name))
    WSQ_VAL(int, h_nb1, nb1); WSQ_VAL(int, h_nv, nv); WSQ_VAL(int, h_nd, nd); WSQ_VAL(int, h_tm1, tm1); WSQ_VAL(int, h_tn1, tn1); WSQ_VAL(int, h_xm1, xm1);
    WSQ_VAL(int, h_kps1, kps1); WSQ_VAL(int, h_K1, K1); WSQ_VAL(int, h_tm2, tm2); WSQ_VAL(int, h_tn2, tn2); WSQ_VAL(int, h_xm2, xm2); WSQ_VAL(int, h_K2, K2);
    WSQ_VAL(unsigned, h_mg_nb1, mg_nb1); WSQ_VAL(unsigned, h_dv1, dv1); WSQ_VAL(unsigned, h_mg1, mg1); WSQ_VAL(unsigned, h_dv2, dv2); WSQ_VAL(unsigned, h_mg2, mg2);
    WSQ_VAL(long long*, h_dbg, dbg);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, nv = h_nv;
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if (wave >= 4) {
        // ---------------- producers: one flat stream of k-tiles over all tiles of this workgroup ----------------
        // (their share of the header is loaded in THIS branch only: live here, not in the consumers' registers)
        WSQ_VAL(const __bf16*, h_A1, A1); WSQ_VAL(const __bf16*, h_B1, B1); WSQ_VAL(const __bf16*, h_A2, A2); WSQ_VAL(const __bf16*, h_B2, B2);
        WSQ_VAL(int, h_lda1, lda1); WSQ_VAL(int, h_ldb1, ldb1); WSQ_VAL(int, h_M1, M1); WSQ_VAL(int, h_N1, N1);
        WSQ_VAL(int, h_lda2, lda2); WSQ_VAL(int, h_ldb2, ldb2); WSQ_VAL(int, h_M2, M2); WSQ_VAL(int, h_N2, N2);
        const int pw = wave - 4;
        // geometry of this wave's two pieces of a 64-row operand tile: piece j = instruction pw * 2 + j = 8 LDS lines of 128 bytes
        int line[2], slot = lane & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) line[j] = (pw * 2 + j) * 8 + (lane >> 3);
        int v = blockIdx.x;
        QTile q;
        bool have = false;
        int ti = 0, issued = 0, handed = 0, istage = 0;
        int hleft = 0, hnext = 0;                                        // k-tiles left to hand over in the tile the consumers are in; nk of the tile after it
        int voA[2], voB[2], ksA = 0, ksB = 0;
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((__bf16*)nullptr, 0, 0, 0x00020000), rB = rA;
        auto enter = [&]() {
            // advance to the next valid tile of this workgroup's queue and set up its DMA offsets
            have = false;
            while (v < nv) {
                WSQ_DECODE(v, q, have);
                if (have) break;
                v += G;
            }
            if (!have) return;
            ti = 0;
            {   // (as selects of VALUES: `if (hleft == 0) hleft = nk; else hnext = nk;` became a store through a selected ADDRESS — two
                // ints in scratch, and every scratch access of the producer loop drained the DMA queue with s_waitcnt vmcnt(0))
                const bool first_ = hleft == 0;
                const int nk_ = q.nk, hl_ = hleft, hn_ = hnext;
                hleft = first_ ? nk_ : hl_;
                hnext = first_ ? hn_ : nk_;
            }
            // B of both kinds is row-contiguous ([k][rows], columns n0 ..): W for the input gradient, x for the weight gradient.  The
            // per-kind values are BLENDED arithmetically (kind is 0 / 1): a select or an if / else over them was turned into a two-entry
            // table in scratch indexed by the kind, and a scratch load inside this loop waits with vmcnt(0) — the whole DMA queue.
            const int k = q.kind;
            const int ldb = h_ldb1 + (h_ldb2 - h_ldb1) * k, Nb = h_N1 + (h_N2 - h_N1) * k;
            const unsigned long long b1 = (unsigned long long)h_B1, b2 = (unsigned long long)h_B2, a1 = (unsigned long long)h_A1, a2 = (unsigned long long)h_A2;
            rB = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(b1 + (b2 - b1) * (unsigned long long)k), 0, 0x7fffffff, 0x00020000);
            rA = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(a1 + (a2 - a1) * (unsigned long long)k), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cK = slot ^ swz<true, 8>(line[j]), cR = slot ^ swz<false, 8>(line[j]);
                voB[j] = ((q.kbeg + line[j]) * ldb + min(q.n0 + cR * 8, Nb - 8)) * 2;
                // A: dy — k-contiguous rows m0 .. for the input gradient (kind 0), row-contiguous columns m0 .. for the weight gradient
                const int a_kc = (min(q.m0 + line[j], h_M1 - 1) * h_lda1 + q.kbeg + cK * 8) * 2;
                const int a_rc = ((q.kbeg + line[j]) * h_lda2 + min(q.m0 + cR * 8, h_M2 - 8)) * 2;
                voA[j] = a_kc + (a_rc - a_kc) * k;
            }
            ksA = BK * 2 + (BK * h_lda2 * 2 - BK * 2) * k;
            ksB = BK * ldb * 2;
        };
        auto issue = [&]() {
            unsigned char* dst = smem + istage * WSQ_STG + pw * 2048;
            const int sA = ti * ksA, sB = ti * ksB;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voA[j], sA, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(dst + 8192 + j * 1024), 16, voB[j], sB, 0, 0);
            istage = istage + 1 == S ? 0 : istage + 1;
            ++issued;
            if (++ti == q.nk) { v += G; enter(); }
        };
        long long* dbg = h_dbg;
        auto pstamp = [&](int i) { if (dbg && threadIdx.x == 256) dbg[(long)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime(); };
        pstamp(8);
        enter();
        while (have && issued < S - 1) issue();
        pstamp(9);
#pragma unroll 1
        while (handed < issued) {
            // k-tile `handed` has landed once at most (issued - handed - 1) younger k-tiles are still in flight
            const int fly = issued - handed - 1;
            if (S >= 4 && fly >= 2) wait_vmcnt<2 * PT>();
            else if (fly >= 1) wait_vmcnt<PT>();
            else wait_vmcnt<0>();
            if (handed == 0) pstamp(10);
            barrier();                                                   // this k-tile is the consumers'; every read of the one before it has retired
            ++handed;
            if (have) issue();                                           // ... so that one's stage is restaged right away
            if (--hleft == 0) {
                barrier();                                               // E of the tile just handed over completely
                hleft = hnext; hnext = 0;
            }
            if (dbg && handed == 1) dbg = nullptr, pstamp(11);
        }
        return;
    }
    // ---------------- consumers ----------------
    if (threadIdx.x == 0) { *reinterpret_cast<float*>(smem + WSQ_MISC) = 0.f; *reinterpret_cast<int*>(smem + WSQ_MISC + 4) = 0; }
    int cstage = 0;
    float sqs = 0.f;
    bool any = false;
    long long* dbg = h_dbg;
#pragma unroll 1
    for (int v = blockIdx.x; v < nv; v += G) {
        QTile q;
        bool ok;
        WSQ_DECODE(v, q, ok);
        if (!ok) continue;
        any = true;
        const PairArgs* ka = wsq_kargs();
        // (the lane index is laundered per tile: otherwise the per-lane address arithmetic of BOTH tile kinds — fragment offsets, epilogue
        // rows — is hoisted out of the tile loop and stays live across it: 145-170 VGPRs, or spills inside the k-loop under the 128 cap)
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        if (q.kind == 0) wsq_consume<true, false>(ka->p1, q, smem, cstage, sqs, wave, lane_t, dbg);
        else wsq_consume<false, RS>(ka->p2, q, smem, cstage, sqs, wave, lane_t, dbg);
        dbg = nullptr;
    }
    WSQ_VAL(double*, h_sqacc, sqacc); WSQ_VAL(int, h_sq_mask, sq_mask); WSQ_VAL(int, h_sq_stride, sq_stride);
    if (any && h_sqacc) {
        // the gradient norm's share of every weight-gradient tile of this workgroup: ONE atomic (LDS meeting point without a barrier:
        // a wave's LDS operations execute in order, so whoever draws the last ticket sees all four sums)
        sqs = wave_sum(sqs);
        float* red = reinterpret_cast<float*>(smem + WSQ_MISC);
        int* cnt = reinterpret_cast<int*>(smem + WSQ_MISC + 4);
        if (lane == 0) {
            __hip_atomic_fetch_add(red, sqs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int c = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (c == 3) {
                const float tot = __hip_atomic_load(red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                atomicAdd(h_sqacc + (long)((int)blockIdx.x & h_sq_mask) * h_sq_stride, (double)tot);
            }
        }
    }
}

// p1 / p2 as for ws64_pair_launch
int ws64q_pair_launch(GArgs p1, GArgs p2, hipStream_t st) {
    if (!p1.vec_epi || !p2.vec_epi || (p1.K % BK) || (p2.K % BK) || p1.a_rowsum || p1.sqacc || p1.bias || p2.bias) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (p1.splits < 1) p1.splits = 1;
    p1.k_per_split = cdiv(cdiv(p1.K, p1.splits), BK) * BK;
    p1.splits = cdiv(p1.K, p1.k_per_split);
    // every tile at least WSQ_S - 1 k-tiles deep: the producers run that far ahead, and only ONE tile boundary may lie inside that window
    if (p2.K < (WSQ_S - 1) * BK || p1.k_per_split < (WSQ_S - 1) * BK || p1.K - (p1.splits - 1) * p1.k_per_split < (WSQ_S - 1) * BK) return VITAE_ERR_UNSUPPORTED_SHAPE;
    p1.tiles_m = cdiv(p1.M, 64); p1.tiles_n = cdiv(p1.N, 64); p1.tile0 = 0;
    p2.tiles_m = cdiv(p2.M, 64); p2.tiles_n = cdiv(p2.N, 64); p2.tile0 = 0;
    p2.k_per_split = p2.K; p2.splits = 1;
    // one ticket per (tile, consumer wave)
    if (p1.splits > 1 && (!p1.ws || (long)p1.tiles_m * p1.tiles_n * 4 > VITAE_GLDS_TICKETS)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (p1.M < 8 || p1.N < 8 || p2.M < 8 || p2.N < 8) return VITAE_ERR_UNSUPPORTED_SHAPE;        // (clamped 8-column groups of the row-contiguous operands)
    const int nb1 = 8 * cdiv((long)p1.tiles_m * p1.tiles_n, 8), nb2 = 8 * cdiv((long)p2.tiles_m * p2.tiles_n, 8);
    const int nv = nb1 * p1.splits + nb2;
    if (nv >= 65536 || p1.tiles_m * p1.tiles_n >= 65536 || p2.tiles_m * p2.tiles_n >= 65536) return VITAE_ERR_UNSUPPORTED_SHAPE;     // (the multiply-high divisions)
    auto magic = [](unsigned d) { return (unsigned)((1ull << 32) / d) + 1u; };
    const unsigned dv1 = p1.xcd_m ? p1.tiles_n : p1.tiles_m, dv2 = p2.xcd_m ? p2.tiles_n : p2.tiles_m;
    static const int slots = getenv("VITAE_WSQ_SLOTS") ? atoi(getenv("VITAE_WSQ_SLOTS")) : 512;       // two workgroups on each of the 256 CUs
    const int g = nv < slots ? nv : slots / 8 * 8;
    const dim3 grid(g), block(512);
    PairArgs a;
    a.h = QHead{nb1, nv, nb1 * p1.splits, p1.tiles_m, p1.tiles_n, p1.xcd_m, p1.k_per_split, p1.K, p2.tiles_m, p2.tiles_n, p2.xcd_m, p2.K,
                magic((unsigned)nb1), dv1, magic(dv1), dv2, magic(dv2),
                p1.A, p1.B, (int)p1.lda, (int)p1.ldb, p1.M, p1.N, p2.A, p2.B, (int)p2.lda, (int)p2.ldb, p2.M, p2.N,
                p1.dbg, p2.sqacc, p2.sq_mask, p2.sq_stride};
    a.p1 = p1; a.p2 = p2;
    if (p2.a_rowsum) hipLaunchKernelGGL((gemm_ws64q_pair_kernel<true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm_ws64q_pair_kernel<false>), grid, block, 0, st, a);
    return vitae_launch_status();
}

// Up to four weight-gradient problems dW_i[N_i, K_i] (+)= dy_i^T x_i of one transformer block (same reduction length: the padded
// token count) as ONE launch of 128x128 tiles: together they have enough tiles that the reduction needs no split (batch 32
// encoder: 432 tiles) or a split of two (decoder: 192) where each of them alone wanted 3-8 — and the in-launch split-K fix-up was
// half of those launches (tools/wgrad_split_sweep.py).  Block ranges start at multiples of 8: every problem keeps its XCD map.
struct BtGroup { GArgs p[4]; int start[5]; };

__global__ __launch_bounds__(256, 2) void gemm_bt_wgrad_group_kernel(const BtGroup g) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BtCfg<128, 128, 2, 2>::SMEM];
    const int b = blockIdx.x;
    const int i = (b >= g.start[1]) + (b >= g.start[2]) + (b >= g.start[3]);
    gemm_bt_body<128, 128, 2, 2, false, false>(g.p[i], b - g.start[i], blockIdx.z, smem);
}

// The same group on the wave-specialised 128 x 128 workgroup (one per CU): a k-tile of the weight-gradient form costs it ~900
// clocks against ~1300 per tile for the two co-resident ping-pong workgroups above (both operands come through the transposing
// reads, and there the waves that wait for them are the ones that multiply).
__global__ __launch_bounds__(512, 2) void gemm_ws_wgrad_group_kernel(const BtGroup g) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[VITAE_WS_TAIL_ALIAS ? VITAE_WS_STAGES * 32768 : VITAE_WS_STAGES * 32768 + (VITAE_WS_TAIL8 ? 12 * 4096 + 64 : 0)];
    const int b = blockIdx.x;
    const int i = (b >= g.start[1]) + (b >= g.start[2]) + (b >= g.start[3]);
    gemm_ws_body<false, false, VITAE_WS_STAGES>(g.p[i], b - g.start[i], blockIdx.z, smem);
}

// ... and on 128 x 256 tiles of it (gemm_wsw_body): 216 tiles instead of 432 for an encoder block — one round of the 256 slots
__global__ __launch_bounds__(512, 2) void gemm_wsw_wgrad_group_kernel(const BtGroup g) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * 49152];
    const int b = blockIdx.x;
    const int i = (b >= g.start[1]) + (b >= g.start[2]) + (b >= g.start[3]);
    gemm_wsw_body<3>(g.p[i], b - g.start[i], blockIdx.z, smem);
}

template <int BM, int BN, int WM, int WN>
static void bt_launch_cfg(const GArgs& p, bool a_kc, bool b_kc, hipStream_t st) {
    const dim3 grid(8 * cdiv((long)p.tiles_m * p.tiles_n, 8), 1, p.splits), block(64 * WM * WN);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bt_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_bt_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, st, p);
    else if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm_bt_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, st, p);
}

bool bt_tile_dims(int id, int& bm, int& bn) {
    switch (id) {
        case 0: bm = 256; bn = 256; return true;
        case 3: bm = 128; bn = 128; return true;
        case 4: bm = 128; bn = 128; return true;       // wave-specialised (4 MFMA waves + 4 DMA waves, one workgroup per CU)
        case 5: bm = 64; bn = 64; return true;         // ... on a 64 x 64 tile (unsplit launches)
        case 6: bm = 128; bn = 256; return true;       // ... on a 128 x 256 tile, weight-gradient form only
        default: return false;
    }
}

// p: a complete problem descriptor (no a_rowsum); tiles_m / tiles_n are set here
int bt_launch(GArgs p, int a_kc, int b_kc, int id, hipStream_t st) {
    int bm, bn;
    if (!bt_tile_dims(id, bm, bn)) return VITAE_ERR_INVALID_ARG;
    if (!p.vec_epi || p.a_rowsum || (p.K % BK) || p.splits < 1) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (!a_kc && b_kc) return VITAE_ERR_UNSUPPORTED_SHAPE;
    // every split gets k_per_split (a multiple of 64) except the last; each needs >= 2 k-tiles
    p.k_per_split = cdiv(cdiv(p.K, p.splits), BK) * BK;
    p.splits = cdiv(p.K, p.k_per_split);
    if (p.K - (p.splits - 1) * p.k_per_split < 2 * BK || p.k_per_split < 2 * BK) return VITAE_ERR_UNSUPPORTED_SHAPE;
    p.tiles_m = cdiv(p.M, bm); p.tiles_n = cdiv(p.N, bn);
    if (p.splits > 1 && (!p.ws || id == 0 || (long)p.tiles_m * p.tiles_n > VITAE_GLDS_TICKETS || p.epi == VITAE_EPI_GELU)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    // (128x64 / 64x128 on TWO waves, three workgroups per CU, were tried for the batch-8 encoder shapes where the vendor library uses
    // 128x64 / 128x96 macro tiles: 13.9 vs 13.4 us for the 64-row family on qkv, 15.7 vs 14.4 on fc1 — not kept.)
    // (256x128 and 128x256 on eight waves were built and measured too: four MFMAs per phase against the same barrier / DMA
    // overhead as eight — 2150 clocks per k-tile for 1024 of MFMA — never the best tile on any shape of the step: not kept)
    if (id == 6 && (a_kc || b_kc)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (id == 0) bt_launch_cfg<256, 256, 2, 4>(p, a_kc, b_kc, st);
    else if (id == 6) {
        // (through the group kernel as a group of one: the stand-alone instantiation of the same body came out of hipcc with 136
        // spilled registers, the group one with 4)
        BtGroup g;
        const int total = 8 * cdiv(p.tiles_m * p.tiles_n, 8);
        for (int i = 0; i < 4; ++i) { g.p[i] = p; g.start[i] = i ? total : 0; }
        g.start[4] = total;
        hipLaunchKernelGGL(gemm_wsw_wgrad_group_kernel, dim3(total, 1, p.splits), dim3(512), 0, st, g);
    } else if (id == 5) {
        const dim3 grid(8 * cdiv((long)p.tiles_m * p.tiles_n, 8), 1, p.splits), block(64 * (4 + VITAE_WS64_PRODUCERS));
        if (a_kc && b_kc) hipLaunchKernelGGL((gemm_ws64_kernel<true, true>), grid, block, 0, st, WS_HOT_ARGS(p), p);
        else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_ws64_kernel<true, false>), grid, block, 0, st, WS_HOT_ARGS(p), p);
        else hipLaunchKernelGGL((gemm_ws64_kernel<false, false>), grid, block, 0, st, WS_HOT_ARGS(p), p);
    } else if (id == 4) {
        const dim3 grid(8 * cdiv((long)p.tiles_m * p.tiles_n, 8), 1, p.splits), block(512);
        if (a_kc && b_kc) hipLaunchKernelGGL((gemm_ws_kernel<true, true>), grid, block, 0, st, p);
        else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_ws_kernel<true, false>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_ws_kernel<false, false>), grid, block, 0, st, p);
    } else bt_launch_cfg<128, 128, 2, 2>(p, a_kc, b_kc, st);
    return vitae_launch_status();
}

// n <= 4 complete weight-gradient descriptors (A = dy16 [K, M] row-contiguous, B = x16 [K, N], same K, vec_epi set); `splits`
// k-ranges for all of them; ws: tickets + partial tiles for the sum of their tiles
int bt_wgrad_group_launch(GArgs* ps, int n, int splits, hipStream_t st, int kind /* 0 ping-pong 128x128, 1 ws 128x128, 2 ws 128x256 */) {
    if (n < 1 || n > 4) return VITAE_ERR_INVALID_ARG;
    BtGroup g;
    int total = 0, tiles = 0;
    const int K = ps[0].K;
    int kps = cdiv(cdiv(K, splits), BK) * BK;
    splits = cdiv(K, kps);
    if (K - (splits - 1) * kps < 2 * BK || kps < 2 * BK) return VITAE_ERR_UNSUPPORTED_SHAPE;
    for (int i = 0; i < 4; ++i) {
        g.start[i] = total;
        if (i >= n) { g.p[i] = g.p[0]; continue; }
        GArgs& p = ps[i];
        if (!p.vec_epi || p.a_rowsum || p.K != K || (K % BK)) return VITAE_ERR_UNSUPPORTED_SHAPE;
        p.k_per_split = kps; p.splits = splits;
        p.tiles_m = cdiv(p.M, 128); p.tiles_n = cdiv(p.N, kind == 2 ? 256 : 128);
        p.tile0 = tiles;
        tiles += p.tiles_m * p.tiles_n;
        total += 8 * cdiv(p.tiles_m * p.tiles_n, 8);
        g.p[i] = p;
    }
    g.start[4] = total;
    for (int i = n; i < 4; ++i) g.start[i] = total;
    if (splits > 1 && (!ps[0].ws || tiles > VITAE_GLDS_TICKETS)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (kind == 2) hipLaunchKernelGGL(gemm_wsw_wgrad_group_kernel, dim3(total, 1, splits), dim3(512), 0, st, g);
    else if (kind == 1) hipLaunchKernelGGL(gemm_ws_wgrad_group_kernel, dim3(total, 1, splits), dim3(512), 0, st, g);
    else hipLaunchKernelGGL(gemm_bt_wgrad_group_kernel, dim3(total, 1, splits), dim3(256), 0, st, g);
    return vitae_launch_status();
}

}  // namespace vglds
