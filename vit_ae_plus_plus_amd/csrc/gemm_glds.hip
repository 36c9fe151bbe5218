// bf16 x bf16 GEMM with a 3-stage LDS-DMA pipeline (global_load_lds, 16 B per lane) — the throughput
// kernel of the path once activations are kept in bf16:
//   C[M,N] (+)= epi( sum_k A(m,k) * B(n,k) + bias[n] ) (+ residual)      (fp32 accumulate / fp32 C,
//   optional bf16 copy C16 for the next GEMM's operand)
// Operand storage flags as in gemm.hip (*_KC: element (row,k) at ptr[row*ld + k], else ptr[k*ld + row]).
//
// Why this shape on MI355X: at M ~ 440-870 a workgroup's K loop is latency bound (a CU fetches
// ~bytes-in-flight per ~1 us), and hipcc collapses register-staged multi-tile prefetch into one tile in
// flight.  LDS-DMA needs no registers: two 64-deep k-tiles (2 x 16-24 KB per workgroup) stay in
// flight behind the tile being multiplied (a third in-flight stage bought nothing, while 48 KB instead of
// 64 KB of LDS lets three workgroups share a CU: pred fwd 73 -> 50 us), retired with COUNTED s_waitcnt
// vmcnt(N) + a raw s_barrier
// (a __syncthreads() would drain the DMA queue; cdna_hip_programming.md §5 "Pipelining across barriers").
// The DMA writes LDS lane-linearly (wave-uniform base + lane*16), so bank conflicts are avoided by
// permuting the SOURCE addresses: 16-byte chunk c of tile row r is stored at chunk slot c ^ (r & 7),
// and the fragment reads (ds_read_b128 for k-contiguous tiles, ds_read_b64_tr_b16 for row-contiguous
// ones) apply the same XOR.  K (and the k-range of each split) must be a multiple of 64 — the engine
// pads token counts of its bf16 activation buffers to 64 with zero rows for the wgrad reductions.
#include <cstdlib>
#include "common.hpp"
#include "glds_tiles.hpp"
#include "glds_gemm.hpp"
#include "vitae_hip.h"

namespace {

using namespace vglds;

long long* g_gemm_dbg = nullptr;
double* g_wgrad_sqacc = nullptr;   // vitae_gemm_glds_set_wgrad_sqnorm(_spread): picked up by every weight-gradient launch while set
int g_wgrad_sq_mask = 0, g_wgrad_sq_stride = 0;
static inline void set_sq(vglds::GArgs& p) { p.sqacc = g_wgrad_sqacc; p.sq_mask = g_wgrad_sq_mask; p.sq_stride = g_wgrad_sq_stride; }

// Row-major epilogue.  The MFMA result has one COLUMN per lane (16 rows of it), so storing from registers means 4-byte
// accesses (2-byte for the bf16 copy), two 128-byte row pieces per instruction.  Here the finished tile is parked in
// LDS (the stage buffers are free by then) and re-read row-major: every lane owns four consecutive columns of a row, so
// C, the bf16 copy, the saved pre-activation, the residual and the old C all move 16 bytes per lane (8 for bf16) in
// 256-byte contiguous runs.  Loads of PB passes are issued together before the dependent stores, as in epilogue_frag.
// Operands of the row-major epilogue that do not depend on the product (bias, residual), fetched at KERNEL START for 64x64
// tiles: issued after the k-loop they cost one exposed memory latency (~800-1500 clocks of a ~5000-clock epilogue).
struct EpiPre { f32x4 bias, res[4]; bool have; };

template <int BM, int BN, int NW>
__device__ __forceinline__ void epilogue_prefetch(const GArgs& p, int m0, int n0, EpiPre& e) {
    constexpr int NT = 64 * NW, CG = BN / 4, RPP = NT / CG, PASSES = BM / RPP;
    e.have = false;
    if constexpr (BM == 64 && BN == 64 && NW == 4) {
        static_assert(PASSES == 4, "prefetch geometry");
        if (!p.vec_epi) return;
        e.have = true;
        const int cg = threadIdx.x % CG, r0 = threadIdx.x / CG;
        const int n = n0 + 4 * cg, nc = n < p.N ? n : 0;
        e.bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) e.bias = *reinterpret_cast<const f32x4*>(p.bias + nc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e.res[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.residual) e.res[q] = *reinterpret_cast<const f32x4*>(p.residual + min(m0 + r0 + q * RPP, p.M - 1) * (int)p.ldr + nc);
        }
    }
}

template <int BM, int BN, int NW, int NF>
__device__ __forceinline__ void epilogue_rows(const GArgs& p, const float (&a)[NF][16], int m0, int n0, int wm, int wn,
                                              int lane, unsigned char* smem, const EpiPre& pre) {
    constexpr int WAVES_M = NW / 2, FN = BN / 64, FM = BM / (32 * WAVES_M);
    constexpr int LDT = BN + 4, NT = 64 * NW, CG = BN / 4, RPP = NT / CG, PASSES = BM / RPP, PB = BN > 64 ? 2 : 4;
    static_assert(NF == FM * FN && PASSES % PB == 0, "tile / thread geometry");
    float* T = reinterpret_cast<float*>(smem);
    float* cs = T + BM * LDT;
    const int l31 = lane & 31, hi = lane >> 5;
    auto estamp = [&](int i) {             // tools/gemm_phase_probe.py: epilogue phases (one launch, z = 0 only)
        if (p.dbg && threadIdx.x == 0 && blockIdx.z == 0) p.dbg[(long)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    __syncthreads();                       // every wave is done with the operand stages
    estamp(12);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                T[(wm * (BM / WAVES_M) + fm * 32 + crow(r, hi)) * LDT + wn * (BN / 2) + fn * 32 + l31] = a[fm * FN + fn][r];
    if (p.out_colsum && (int)threadIdx.x < BN) cs[threadIdx.x] = 0.f;
    __syncthreads();
    estamp(13);
    const int cg = threadIdx.x % CG, r0 = threadIdx.x / CG;
    const int n = n0 + 4 * cg;
    const bool ncol = n < p.N;                                   // N % 4 == 0 here: the whole group is in or out
    const int nc = ncol ? n : 0;
    const int ldaux = (int)p.ldaux, ldr = (int)p.ldr, ldc = (int)p.ldc, ldc16 = (int)p.ldc16;
    const bool need_aux = p.epi == VITAE_EPI_DGELU || p.epi == VITAE_EPI_RELU_MASK;
    const bool acc_c = p.C && p.accumulate;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (pre.have) bias4 = pre.bias;
    else if (p.bias && ncol) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    float sqs = 0.f;
#pragma unroll
    for (int pb = 0; pb < PASSES; pb += PB) {
        f32x4 ax[PB], rs[PB], co[PB];
        bf16x4 ax16[PB];       // a bf16 aux stays raw until it is used: converting right behind the load made hipcc wait for every load in turn
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int mc = min(m0 + r0 + (pb + q) * RPP, p.M - 1);
            if (need_aux) {
                if (p.aux16) ax16[q] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.aux) + mc * ldaux + nc);
                else ax[q] = *reinterpret_cast<const f32x4*>(p.aux + mc * ldaux + nc);
            }
            if (p.residual) {
                if (pre.have) rs[q] = pre.res[(pb + q) & 3];
                else rs[q] = *reinterpret_cast<const f32x4*>(p.residual + mc * ldr + nc);
            }
            if (acc_c) co[q] = *reinterpret_cast<const f32x4*>(p.C + mc * ldc + nc);
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int row = r0 + (pb + q) * RPP, m = m0 + row;
            if (!(ncol && m < p.M)) continue;
            f32x4 x = *reinterpret_cast<const f32x4*>(&T[row * LDT + 4 * cg]);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] += bias4[e];
            if (p.epi == VITAE_EPI_GELU) {
                f32x4 y, dy;
                gelu_fast4(x, y, dy);
                const f32x4 sv = p.auxd ? dy : x;              // what the backward gets: GELU'(x) or x itself
                if (p.aux16) {
                    bf16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (__bf16)sv[e];
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.aux) + m * ldaux + n) = h;
                } else {
                    *reinterpret_cast<f32x4*>(p.aux + m * ldaux + n) = sv;
                }
                x = y;
            } else if (p.epi == VITAE_EPI_DGELU) {
                if (p.aux16) ax[q] = f32x4{(float)ax16[q][0], (float)ax16[q][1], (float)ax16[q][2], (float)ax16[q][3]};
                const f32x4 g = p.auxd ? ax[q] : gelu_fast_grad4(ax[q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] *= g[e];
            } else if (p.epi == VITAE_EPI_RELU_MASK) {
                if (p.aux16) ax[q] = f32x4{(float)ax16[q][0], (float)ax16[q][1], (float)ax16[q][2], (float)ax16[q][3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = ax[q][e] > 0.f ? x[e] : 0.f;
            } else if (p.epi == VITAE_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
            }
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] += rs[q][e];
            }
#ifndef VITAE_EPI_ABLATE
#define VITAE_EPI_ABLATE 0       // timing ablations of the row-major epilogue: 1 = no global stores, 2 = fp32 store only
#endif
            if (p.C) {
                if (acc_c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] += co[q][e];
                }
                if (VITAE_EPI_ABLATE != 1) *reinterpret_cast<f32x4*>(p.C + m * ldc + n) = x;
            }
            if (p.C16 && VITAE_EPI_ABLATE == 0) {
                bf16x4 x16;
#pragma unroll
                for (int e = 0; e < 4; ++e) x16[e] = (__bf16)x[e];
                *reinterpret_cast<bf16x4*>(p.C16 + m * ldc16 + n) = x16;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += x[e];
            if (p.sqacc) sqs += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
        }
        asm volatile("" ::: "memory");   // keep the next batch's loads behind these stores (register pressure)
    }
    if (p.sqacc) {
        // the gradient norm's share of this tile (every stored element exactly once): one double atomic per workgroup, instead
        // of a separate pass over the 0.5 GB gradient arena (grad_sqnorm_kernel: 45 us per bucket, the last one exposed)
        const float tot = block_sum_256(sqs, cs + BN);
        if (threadIdx.x == 0) atomicAdd(sq_slot(p), (double)tot);
    }
    if (p.out_colsum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&cs[4 * cg + e], csum[e]);      // LDS: RPP-way per column
        __syncthreads();
        if ((int)threadIdx.x < BN && n0 + (int)threadIdx.x < p.N) atomicAdd(p.out_colsum + n0 + threadIdx.x, cs[threadIdx.x]);
    }
}

// Workgroup tile BM x BN (64x64 or 64x128), four waves in a 2 x 2 arrangement: each wave owns (BM/2) x (BN/2) = FM x FN
// accumulator fragments of 32x32.  (The 128x128 and 256x256 tiles are a kernel family of their own: gemm_bt.hip.)
#ifndef VITAE_GLDS_NS_WIDE
#define VITAE_GLDS_NS_WIDE 2
#endif
#ifndef VITAE_GLDS_NACC_BIG
#define VITAE_GLDS_NACC_BIG 1
#endif
#ifndef VITAE_GLDS_NS_PAIR
#define VITAE_GLDS_NS_PAIR 0     // > 0: stages of the 64x64 tiles inside the paired (dgrad + wgrad) launch.  Measured in the step
                                 // (ms): 2 stages (32 KB, four workgroups per CU) 5.85 although 7 % faster in the L2-warm
                                 // micro-benchmark; 4 stages 5.65; 3 stages (the default) 5.52.  The same for the forward tile.
#endif
template <int BM, int BN, int NW = 4, bool PAIR = false> struct GCfg {
    // stages: 3 for 64x64 (48 KB, three workgroups per CU); the wider 4-wave tiles take 2 (48 KB for 64x128 -> three
    // workgroups per CU instead of two: decoder_pred fwd 49.8 -> 45.6 us) — occupancy beats prefetch depth there
    static constexpr int NST = (BM * BN > 64 * 64) ? VITAE_GLDS_NS_WIDE : (PAIR && VITAE_GLDS_NS_PAIR > 0) ? VITAE_GLDS_NS_PAIR : NS;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES, SMEM = NST * STAGE;
};

template <int BM, int BN, bool A_KC, bool B_KC, int NW = 4, bool RS = false, bool PAIR = false, int PIPE = 0>
__device__ __forceinline__ void gemm_glds_body(const GArgs& p, const int bid, const int zid, unsigned char* smem) {
    constexpr int WAVES_M = NW / 2, NT = 64 * NW;          // waves: WAVES_M x 2
    constexpr int FM = BM / (32 * WAVES_M), FN = BN / 64, NF = FM * FN;
    constexpr int A_BYTES = GCfg<BM, BN, NW, PAIR>::A_BYTES, STAGE = GCfg<BM, BN, NW, PAIR>::STAGE;
    constexpr int NST = PIPE ? PIPE : GCfg<BM, BN, NW, PAIR>::NST;
    static_assert(!PIPE || (A_KC && B_KC && PIPE >= 4 && !RS), "the pipelined loop: k-contiguous operands, >= 4 stages");
    constexpr int G = (BM + BN) / (8 * NW);                // DMA instructions per wave per stage
    const int xcd = bid & 7, local = bid >> 3;
    const int tn = p.xcd_m ? local % p.tiles_n : xcd + 8 * (local / p.tiles_m);
    const int tm = p.xcd_m ? xcd + 8 * (local / p.tiles_n) : local % p.tiles_m;
    if (tn >= p.tiles_n || tm >= p.tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BN;
    auto stamp = [&](int i) {
        if (p.dbg && threadIdx.x == 0) p.dbg[((long)zid * gridDim.x + bid) * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int kbeg = zid * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg) / BK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // two accumulators per output fragment (even / odd 16-deep k-slices): consecutive MFMAs never depend
    // on each other, so the matrix pipe is not serialised on the 32x32 accumulate latency
    // (with four or more fragments per wave the fragments themselves are the independent chains: one set)
    constexpr int NACC = NF >= 4 ? VITAE_GLDS_NACC_BIG : 2;
    f32x16 acc[NACC][NF];
#pragma unroll
    for (int h = 0; h < NACC; ++h)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[h][f][i] = 0.f;

    // RS instantiations only (the extra accumulators cost the wide tiles a workgroup per CU)
    const bool rowsum = RS && p.a_rowsum != nullptr && tn == 0 && wn == 0;   // wave-uniform; every k-split adds its share
    f32x16 accx[RS ? FM : 1];
#pragma unroll
    for (int f = 0; f < (RS ? FM : 1); ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) accx[f][i] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    auto issue = [&](int t) {
        unsigned char* st = smem + (t % NST) * STAGE;
        dma_tile<BM, A_KC, NW>(p.A, p.lda, p.M, m0, kbeg + t * BK, st, wave, lane);
        dma_tile<BN, B_KC, NW>(p.B, p.ldb, p.N, n0, kbeg + t * BK, st + A_BYTES, wave, lane);
    };
    EpiPre epre;
    epilogue_prefetch<BM, BN, NW>(p, m0, n0, epre);     // older than every DMA: the counted waits below never see these loads
    const int pre = min(nk, NST - 1);
    for (int t = 0; t < pre; ++t) issue(t);
    stamp(1);

    if constexpr (PIPE != 0) {
        // Software-pipelined k-loop for launches with ONE (or two) workgroups per CU, where nothing else overlaps a
        // workgroup's phases: the fragments of tile t + 1 are read from LDS while the MFMAs of tile t run from registers
        // (two fragment sets), so the LDS read latency (~250 clocks per step for a 64x64 tile: 32 KB through a 256 B/clk
        // LDS) and the wait for the next tile's DMA are no longer in series with the MFMAs.  Needs tile t + 1 landed one
        // step earlier than the classic loop, hence one more stage (NST = 4) to keep two tiles in flight.
        bf16x8 f0a[BK / 16][FM], f0b[BK / 16][FN], f1a[BK / 16][FM], f1b[BK / 16][FN];
        auto rd = [&](int t, bf16x8 (&fa)[BK / 16][FM], bf16x8 (&fb)[BK / 16][FN]) {
            const unsigned char* at = smem + (t % NST) * STAGE;
            const unsigned char* bt = at + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                for (int f = 0; f < FM; ++f) fa[kk][f] = frag<BM, true>(at, wm * (BM / WAVES_M) + f * 32, kk, lane);
#pragma unroll
                for (int f = 0; f < FN; ++f) fb[kk][f] = frag<BN, true>(bt, wn * (BN / 2) + f * 32, kk, lane);
            }
        };
        auto mm = [&](bf16x8 (&fa)[BK / 16][FM], bf16x8 (&fb)[BK / 16][FN]) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                    for (int fn = 0; fn < FN; ++fn)
                        acc[kk % NACC][fm * FN + fn] =
                            __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][fm], fb[kk][fn], acc[kk % NACC][fm * FN + fn], 0, 0, 0);
        };
        // tile 0
        {
            const int younger = min(nk - 1, NST - 2);
            if (younger >= 2) wait_vmcnt<2 * G>();
            else if (younger == 1) wait_vmcnt<G>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            stamp(2);
            rd(0, f0a, f0b);
        }
        auto step = [&](int t, bf16x8 (&ca)[BK / 16][FM], bf16x8 (&cb)[BK / 16][FN], bf16x8 (&na)[BK / 16][FM],
                        bf16x8 (&nb)[BK / 16][FN]) {
            const bool next = t + 1 < nk;
            if (next) {
                // tile t + 1 must have landed; issued and younger: tiles t + 2 .. min(nk - 1, t + NST - 2)
                if (min(nk - 2 - t, NST - 3) >= 1) wait_vmcnt<G>();
                else wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();      // everyone's pieces of tile t + 1 landed; the stage of tile t - 1 is free
            if (t + NST - 1 < nk) issue(t + NST - 1);
            if (next) rd(t + 1, na, nb);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
        };
        int t = 0;
        for (; t + 1 < nk; t += 2) {
            step(t, f0a, f0b, f1a, f1b);
            step(t + 1, f1a, f1b, f0a, f0b);
        }
        if (t < nk) step(t, f0a, f0b, f1a, f1b);
    } else
    for (int t = 0; t < nk; ++t) {
        // tile t must have landed: allow the (up to two) younger stages to stay in flight
        const int younger = min(nk - 1 - t, NST - 2);
        if (younger >= 3) wait_vmcnt<3 * G>();
        else if (younger == 2) wait_vmcnt<2 * G>();
        else if (younger == 1) wait_vmcnt<G>();
        else wait_vmcnt<0>();
        if (t == 4) stamp(8);
        __builtin_amdgcn_s_barrier();          // everyone's DMA pieces of tile t landed; tile t-1 fully consumed
        if (t == 0) stamp(2);
        if (t == 4) stamp(9);
        const bool more = t + NST - 1 < nk;    // the stage tile t-1 used is refilled with tile t + NST - 1 ...
        if (more) issue(t + NST - 1);
        if (t == 4) stamp(10);
        const unsigned char* at = smem + (t % NST) * STAGE;
        const unsigned char* bt = at + A_BYTES;
        bf16x8 fa[BK / 16][FM], fb[BK / 16][FN];
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
            for (int f = 0; f < FM; ++f) fa[kk][f] = frag<BM, A_KC>(at, wm * (BM / WAVES_M) + f * 32, kk, lane);
#pragma unroll
            for (int f = 0; f < FN; ++f) fb[kk][f] = frag<BN, B_KC>(bt, wn * (BN / 2) + f * 32, kk, lane);
        }
        // all fragment reads of the tile are issued before the first MFMA (hipcc otherwise recycles three fragment
        // registers and waits for a fresh LDS read in front of every MFMA: one LDS latency per MFMA)
        if constexpr (!A_KC || !B_KC) {     // transposing reads are inline asm (glds_tiles.hpp): order them by hand
            frags_ready();
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                for (int f = 0; f < FM; ++f) frag_tie(fa[kk][f]);
#pragma unroll
                for (int f = 0; f < FN; ++f) frag_tie(fb[kk][f]);
            }
        }
        if (t == 4 && p.dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(11); }
        __builtin_amdgcn_sched_barrier(0);
        // (DMA pieces issued one by one between these MFMAs instead of as a block above: measured 4.91 vs 4.87 ms per step
        // here, where the GEMMs mostly share a CU — removed; the big tiles of gemm_bt.hip, alone on their CU, do issue that way)
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
                    acc[kk % NACC][fm * FN + fn] =
                        __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][fm], fb[kk][fn], acc[kk % NACC][fm * FN + fn], 0, 0, 0);
        if (t == 3) stamp(12);
        if (t == 4) stamp(13);
        if constexpr (RS) {
            if (rowsum) {
#pragma unroll
                for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm)
                        accx[fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][fm], ones, accx[fm], 0, 0, 0);
            }
        }
    }
    stamp(3);
    if (RS && rowsum && l31 == 0) {
        // every column of accx holds the row sums of this workgroup's k-range
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WAVES_M) + fm * 32 + crow(r, hi);
                if (m < p.M) atomicAdd(p.a_rowsum + m, accx[RS ? fm : 0][r]);
            }
    }

    float a[NF][16];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[f][r] = NACC == 2 ? acc[0][f][r] + acc[NACC - 1][f][r] : acc[0][f][r];

    if (p.splits > 1) {
        // Split-K fix-up without a second launch: every split parks its partial tile (fragment order, coalesced),
        // takes a ticket, and the LAST one to arrive sums all partials in split order (bitwise reproducible whatever
        // the arrival order) and runs the epilogue.  Fences: release before the ticket, acquire after it.
        // Partials and tickets move with agent-scope relaxed atomics (sc1: written through to / read from the
        // memory side, coherent across the 8 XCD L2s); a full release/acquire fence pair instead would write back
        // and invalidate the whole L2 per workgroup (measured 3x slower than no split at all).
        // Round 3: the partials move 16 bytes per lane (write-through buffer stores / sc1 buffer loads instead of 4-byte agent
        // atomics), and the last arriver keeps the groups of FOUR splits in flight together: one round trip to the memory side
        // per four splits instead of one per split (the fix-up cost the last arriver ~7000 clocks, tools/gemm_phase_probe.py).
        const int tile = tm * p.tiles_n + tn;
        float* part = p.ws + VITAE_GLDS_TICKETS + ((long)tile * p.splits) * (BM * BN);
        int* ticket = reinterpret_cast<int*>(p.ws) + tile;
        constexpr int GRP = NF * 4;                                   // 16-byte groups per lane and split
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, p.splits * (BM * BN * 4), 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int toff = (int)threadIdx.x * 16;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v = {a[f][4 * g4], a[f][4 * g4 + 1], a[f][4 * g4 + 2], a[f][4 * g4 + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (zid * GRP + f * 4 + g4) * (NT * 16) + toff, 0, 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's partials are out
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        stamp(4);
        if (*flag != p.splits - 1) return;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[f][r] = 0.f;
        constexpr int SB = NF == 1 ? 4 : 2;                           // splits in flight together
#pragma unroll 1
        for (int sp0 = 0; sp0 < p.splits; sp0 += SB) {
            f32x4 v[SB][GRP];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int sp = min(sp0 + u, p.splits - 1);             // (clamped: a repeated load, never added)
#pragma unroll
                for (int i = 0; i < GRP; ++i)
                    v[u][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (sp * GRP + i) * (NT * 16) + toff, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < SB; ++u)
                if (sp0 + u < p.splits) {
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[f][4 * g4 + e] += v[u][f * 4 + g4][e];
                }
        }
        if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }

    stamp(5);
    if constexpr (BM * BN <= 64 * 128 && NW == 4) {
        if (p.vec_epi) {
            epilogue_rows<BM, BN, NW, NF>(p, a, m0, n0, wm, wn, lane, smem, epre);
            stamp(6);
            if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(7); }
            return;
        }
    }
    float sqsum = 0.f;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int n = n0 + wn * (BN / 2) + fn * 32 + l31;
        float csum = 0.f;
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) csum += epilogue_frag(p, a[fm * FN + fn], m0 + wm * (BM / WAVES_M) + fm * 32, n, hi, sqsum);
        if (p.out_colsum) {
            csum += __shfl_xor(csum, 32, 64);
            if (hi == 0 && n < p.N) atomicAdd(p.out_colsum + n, csum);
        }
    }
    if (p.sqacc) {
        sqsum = wave_sum(sqsum);
        if (lane == 0) atomicAdd(sq_slot(p), (double)sqsum);
    }
}

template <int BM, int BN, bool A_KC, bool B_KC, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_glds_kernel(const GArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[GCfg<BM, BN, NW>::SMEM];   // the ONLY LDS object
    gemm_glds_body<BM, BN, A_KC, B_KC, NW>(p, blockIdx.x, blockIdx.z, smem);
}

// forward form (both operands k-contiguous) with the software-pipelined k-loop: launches of at most ~2 workgroups per CU
#ifndef VITAE_GLDS_PIPE_STAGES
#define VITAE_GLDS_PIPE_STAGES 4
#endif
__global__ __launch_bounds__(256) void gemm_glds_pipe_kernel(const GArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[VITAE_GLDS_PIPE_STAGES * GCfg<64, 64, 4>::STAGE];   // the ONLY LDS object
    gemm_glds_body<64, 64, true, true, 4, false, false, VITAE_GLDS_PIPE_STAGES>(p, blockIdx.x, blockIdx.z, smem);
}

// dgrad (dy @ W: A k-contiguous, B = bf16 weights read row-contiguous) and wgrad (dy^T @ x: both operands
// row-contiguous) of one Linear in one launch — they share dy, and together they double the resident
// workgroups per CU.
template <int BM1, int BN1, int BM2, int BN2, bool RS = false>
__global__ __launch_bounds__(256) void gemm_glds_pair_kernel(const GArgs p1, const GArgs p2, const int nb1) {
    constexpr int S1 = GCfg<BM1, BN1, 4, true>::SMEM, S2 = GCfg<BM2, BN2, 4, true>::SMEM;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[S1 > S2 ? S1 : S2];
    // nb1 = workgroups of one dgrad split; the dgrad's long reduction (N of the Linear) is cut into p1.splits
    if ((int)blockIdx.x < nb1 * p1.splits) gemm_glds_body<BM1, BN1, true, false, 4, false, true>(p1, blockIdx.x % nb1, blockIdx.x / nb1, smem);
    else gemm_glds_body<BM2, BN2, false, false, 4, RS, true>(p2, blockIdx.x - nb1 * p1.splits, 0, smem);
}

template <int BM, int BN, int NW = 4>
void launch(const GArgs& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t st) {
    dim3 block(64 * NW);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, true, true, NW>), grid, block, 0, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, true, false, NW>), grid, block, 0, st, p);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, false, true, NW>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, false, false, NW>), grid, block, 0, st, p);
}

// out[n] += sum_m dy16[m, n] (rows m < M of a [*, N] bf16 matrix): fallback for the bias gradient when the paired launch
// has no row-sum instantiation for its tile shapes
// (N % 4 == 0.)  A workgroup owns 256 columns (a lane: four of them, 8-byte loads, four rows in flight) and 4 x rows_per_wave
// rows — its four waves take interleaved rows and meet in LDS, so that ONE atomic per column leaves a workgroup: the launch time
// followed the number of device-scope atomics (13.5 us for 253 k of them at [3520, 2304] with 2-byte loads and 32 rows per
// workgroup; 36 us for 507 k), not the 16 MB it reads.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ dy, float* __restrict__ out, int M, int N,
                                                          int rows_per_wave) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * 4 * rows_per_wave, r1 = min(M, r0 + 4 * rows_per_wave);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        int m = r0 + wave;                                  // rows r0 + wave, + 4, + 8, ...
        const __bf16* p = dy + (long)m * N + n;
        for (; m + 12 < r1; m += 16, p += 16 * (long)N) {
            bf16x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x4*>(p + 4 * u * (long)N);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += (float)v[u][e];
        }
        for (; m < r1; m += 4, p += 4 * (long)N) {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += (float)v[e];
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) {
        const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(out + n + e, t[e]);
    }
}

inline void launch_colsum_bf16(const void* dy16, float* out, int M, int N, hipStream_t st) {
    const int gx = cdiv(N, 256);
    static const int tgt = getenv("VITAE_COLSUM_BLOCKS") ? atoi(getenv("VITAE_COLSUM_BLOCKS")) : 320;
    int rpw = cdiv(cdiv(M, cdiv(tgt, gx)), 4);
    if (rpw < 4) rpw = 4;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, cdiv(M, 4 * rpw)), dim3(256), 0, st, reinterpret_cast<const __bf16*>(dy16), out, M, N, rpw);
}

struct Tile { int bm, bn, id; };

// Every XCD's L2 fetches its own copy of whatever its workgroups read: partition the LARGER operand across the XCDs.
// (wgrad of qkv / fc1 / decoder_pred: A = dy^T is 3-32x the size of B = x; 8 copies of it were most of the launch's
// memory-side traffic.)
// (Tried as well: all workgroups of one k-split on one XCD group, so that no operand slice is fetched by more than
// 8 / splits XCDs — less traffic again, but 5-25 % slower at K <= 3072 and only 3-5 % faster at K = 16384: not kept.)
// the row-major epilogue moves 4-column groups: every array it touches must be addressable that way
inline int vec_epilogue_ok(const GArgs& p) {
    static const int on = getenv("VITAE_GLDS_VEC_EPILOGUE") ? atoi(getenv("VITAE_GLDS_VEC_EPILOGUE")) : 1;
    if (!on || (p.N & 3)) return 0;
    auto ok = [](const void* ptr, long ld, int align) { return !ptr || (!(ld & 3) && !((uintptr_t)ptr & (align - 1))); };
    return ok(p.C, p.ldc, 16) && ok(p.C16, p.ldc16, 8) && ok(p.aux, p.ldaux, 16) && ok(p.residual, p.ldr, 16) && ok(p.bias, 0, 16);
}

inline int xcd_by_rows(int M, int N) {
    static const int mode = getenv("VITAE_GLDS_XCD_ROWS") ? atoi(getenv("VITAE_GLDS_XCD_ROWS")) : -1;
    if (mode >= 0) return mode;
    return M > N;
}

// 0: 64x64, 1: 64x128 — the wider tile once it still yields enough workgroups for 256 CUs.  (A 128x128 tile inside THIS
// kernel — 4 waves / 96 KB or 8 waves / 128 KB of LDS, one workgroup per CU — lost to 64x128 on every large GEMM of the step,
// decoder_pred fwd 54.7 -> 72.5 / 53.8 us: removed in round 3; the big tiles that do win have their own schedule, gemm_bt.hip.
// Round 4, the few-column shapes again with 8 waves / 3 stages and no split: 3520 x 768 x 3072 38.6 us (planner's choice 39.1 in
// the same cold-operand loop), x 768: 18.1 (14.2), 6944 x 512 x 2048: 32.8 (34.5) — ~1580 clocks per 128 x 128 x 64 k-tile.)
inline Tile pick_tile(int M, int N) {
    static const int wide_min = getenv("VITAE_GLDS_WIDE_MIN_TILES") ? atoi(getenv("VITAE_GLDS_WIDE_MIN_TILES")) : 400;
    if (N >= 128 && (long)cdiv(M, 64) * cdiv(N, 128) >= wide_min) return {64, 128, 1};
    return {64, 64, 0};
}

// ---- which kernel family serves a problem: the 64-row tiles of this file or the big tiles of gemm_bt.hip ---------------------
// A cost model in shader clocks, fitted to tools/bt_bench.py / tools/bt_phase_probe.py measurements:
//   big tile, per workgroup: prologue + k-tiles x clocks per 64-deep k-tile + epilogue (+ split-K fix-up); a launch takes
//   ceil(workgroups / resident slots) rounds of that.  256x256: 3000 + 2950 nk + 14500, one workgroup per CU; 128x128: 4500 +
//   1800 nk + 7500, two per CU, split-K fix-up 6000 + 2200 per split (write-through drain, ticket, one round trip per split).
//   64-row tiles: throughput bound tiles x (5100 + 400 nk) / 256 (64x64; 2800 + 930 nk for 64x128) or, with few tiles, the
//   latency of one workgroup 8000 + 760 nk (+ 7000 for the in-launch split-K fix-up).
// g_bt_mode: -1 the model decides, -2 never, 0 / 3: that tile wherever it is eligible (vitae_gemm_glds_set_bt_tile; VITAE_BT_TILE).
int g_bt_mode = getenv("VITAE_BT_TILE") ? atoi(getenv("VITAE_BT_TILE")) : -1;

struct BtPlan { int tile, split; double clocks; };

inline int old_split_rule(int M, int N, int K) {
    static const int min_kt = getenv("VITAE_GLDS_SPLIT_MIN_KT") ? atoi(getenv("VITAE_GLDS_SPLIT_MIN_KT")) : 8;
    static const int target = getenv("VITAE_GLDS_SPLIT_BLOCKS") ? atoi(getenv("VITAE_GLDS_SPLIT_BLOCKS")) : 384;
    const Tile t = pick_tile(M, N);
    const long tiles = (long)cdiv(M, t.bm) * cdiv(N, t.bn);
    if (tiles >= target / 2 || tiles > VITAE_GLDS_TICKETS) return 1;
    long s = (target + tiles - 1) / tiles;
    const long max_by_k = K / (BK * min_kt);   // >= min_kt k-tiles per split
    if (s > max_by_k) s = max_by_k;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
}

inline double est_64row(int M, int N, int K, bool allow_split, bool wgrad_form) {
    const Tile t = pick_tile(M, N);
    const int s = allow_split ? old_split_rule(M, N, K) : 1;
    const double nk = (double)K / BK / s, wgs = (double)cdiv(M, t.bm) * cdiv(N, t.bn) * s;
    // both operands row-contiguous (the weight-gradient form): every fragment comes through the transposing LDS read, two
    // instructions per 16-byte fragment — 650 instead of 400 clocks per k-tile and resident workgroup (tools/wgrad_split_sweep.py:
    // enc fc1 wgrad at batch 32, 576 tiles x 55 k-tiles, 39.8 us)
    // (64 x 128: refitted in round 4 — 1736 x 2048 x 512 takes 14.3 us = 24 k of these clocks, the first fit said 17.9 k and kept the
    // wave-specialised tiles, 11.9-13.2 us there, out)
    const double c = t.id == 1 ? 3800 + 1250 * nk : 5100 + (wgrad_form ? 650 : 400) * nk;
    const double lat = 8000 + (t.id == 1 ? 1000 : 760) * nk + (s > 1 ? 7000 : 0);
    const double thr = wgs * c / 256;
    return thr > lat ? thr : lat;
}

// cap: floats of split-K workspace a plan may need (< 0: no limit — the caller sizes the workspace from the plan's split)
inline BtPlan bt_plan(int M, int N, int K, int a_kc, int b_kc, bool allow_split, long cap = -1, bool allow_ws64 = true, int only = -1) {
    // only >= 0: the best split of THAT tile family alone, by the planner's own (unforced) rules and without the 64-row comparison
    BtPlan best{-1, 1, 1e30};
    if (g_bt_mode == -2 || K < 2 * BK || (K % BK) || (!a_kc && b_kc) || (N & 3)) return best;
    const int nkt = K / BK;
    static const int ws_on = getenv("VITAE_BT_WS") ? atoi(getenv("VITAE_BT_WS")) : 1;      // 0: the wave-specialised tile only when forced
    static const double ws_kt = getenv("VITAE_BT_WS_KT") ? atof(getenv("VITAE_BT_WS_KT")) : 800.0;
    static const int ws_min_rows = getenv("VITAE_BT_WS_MIN_ROWS") ? atoi(getenv("VITAE_BT_WS_MIN_ROWS")) : 2048;
    static const int ws_min_rows_fwd = getenv("VITAE_BT_WS_MIN_ROWS_FWD") ? atoi(getenv("VITAE_BT_WS_MIN_ROWS_FWD")) : 800;      // forward launches are never un-paired: batch 16 7.62 -> 7.51 ms, batch 8 neutral
    static const int ws_long_k = getenv("VITAE_BT_WS_LONG_K") ? atoi(getenv("VITAE_BT_WS_LONG_K")) : 8192;   // ... or a very long reduction (decoder_pred's input gradient, the patch embedding: 37.9 vs 41.7 / 33.8 vs 38.6 us at batch 4)
    static const double ws_kt_w = getenv("VITAE_BT_WS_KT_W") ? atof(getenv("VITAE_BT_WS_KT_W")) : 900.0;     // both operands row-contiguous (every fragment through two transposing reads; round 5, reads one per MFMA gap: 1280 -> 890 clocks per k-tile)
    static const double ws_fix = getenv("VITAE_BT_WS_FIX") ? atof(getenv("VITAE_BT_WS_FIX")) : 14000.0;
    static const int ws64_on = getenv("VITAE_BT_WS64") ? atoi(getenv("VITAE_BT_WS64")) : 1;
    // in-launch split-K fix-up of the 128 x 128 tile (partials out through write-through stores, ticket, the last arriver's reads):
    // refitted in round 5 — 3072 x 768 x 3520 (weight-gradient form) at split 3 takes 36 us = 86 k clocks, the first fit (6000 + 2200 s)
    // priced it at 58 k and kept the wave-specialised tile (28 us) out; 9000 + 4000 s also keeps 3456 x 768 x 16384 at split 3 (105 us) in front of the ws tile (118)
    static const double ws64_kt_w = getenv("VITAE_BT_WS64_KT_W") ? atof(getenv("VITAE_BT_WS64_KT_W")) : 650.0;
    static const int ws_long_any = getenv("VITAE_BT_WS_LONG_ANY") ? atoi(getenv("VITAE_BT_WS_LONG_ANY")) : 1;
    static const double bt_fix0 = getenv("VITAE_BT_FIX0") ? atof(getenv("VITAE_BT_FIX0")) : 9000.0;
    static const double bt_fix1 = getenv("VITAE_BT_FIX1") ? atof(getenv("VITAE_BT_FIX1")) : 4000.0;
    static const double wsw_kt = getenv("VITAE_BT_WSW_KT") ? atof(getenv("VITAE_BT_WSW_KT")) : 1350.0;     // 128 x 256 ws tile (weight-gradient form): 48 KB per k-tile at the CU's 37 B/clk
    static const int wsw_on = getenv("VITAE_BT_WSW") ? atoi(getenv("VITAE_BT_WSW")) : 1;
    for (int id : {0, 3, 4, 5, 6}) {
        if (only >= 0 && id != only) continue;
        if (id == 6 && (a_kc || b_kc || (g_bt_mode < 0 && !wsw_on))) continue;
        if (id == 5 && g_bt_mode != 5 && (!ws64_on || !allow_ws64)) continue;
        if (g_bt_mode >= 0 && id != g_bt_mode) continue;
        if (id == 4 && g_bt_mode < 0 && !ws_on) continue;
        // (the wave-specialised tile needs many token rows: at batch 8 — 880 / 1736 rows — it un-pairs launches the 64-row family
        // serves as well and the step loses 7 %)
        // (round 5: a very long reduction qualifies at any row count — decoder_pred's input gradient at batch 8, 1736 x 512 x 16384, ran
        // 76.8 us on 128 x 128 with split 4 against 51.0 here; such launches are never paired anyway)
        if (id == 4 && g_bt_mode < 0 && (a_kc ? M : K) < ((a_kc && b_kc) ? ws_min_rows_fwd : ws_min_rows) && !(K >= ws_long_k && (ws_long_any || (a_kc ? M : K) < 1024))) continue;
        if (id == 6 && g_bt_mode < 0 && K < ws_min_rows && K < ws_long_k && (long)M * N < (1L << 22)) continue;      // (as the 128 x 128 ws tile: many token rows only — or a big output: decoder_pred's 16384 x 512 at batch 8, 36 against 41 us)
        int bm, bn;
        bt_tile_dims(id, bm, bn);
        // (a FORCED tile — tests, tools — is held to what the kernel itself needs: two k-tiles per split, any M / N)
        const bool forced = g_bt_mode >= 0;
        if (!forced && (M < bm / 2 || N < bn / 2)) continue;
        const long tiles = (long)cdiv(M, bm) * cdiv(N, bn);
        static const int fsplit = getenv("VITAE_BT_SPLIT") ? atoi(getenv("VITAE_BT_SPLIT")) : 0;   // tools: with a forced tile, this split only
        for (int s = 1; s <= (id >= 3 && allow_split ? (id == 6 ? 4 : 8) : 1); ++s) {
            if (forced && fsplit > 0 && s != fsplit) continue;
            const int kps = cdiv(cdiv(K, s), BK) * BK;
            if (cdiv(K, kps) != s || kps < (forced ? 2 : 4) * BK || K - (s - 1) * kps < 2 * BK) continue;
            if (s > 1 && (tiles > VITAE_GLDS_TICKETS || (cap >= 0 && VITAE_GLDS_TICKETS + tiles * s * bm * bn > cap))) continue;
            const double wgs = (double)tiles * s, slots = id == 3 ? 512 : 256;
            const double rounds = (double)cdiv((long)wgs, (long)slots);
            const double nk = (double)nkt / s;
            const double per = id == 0 ? 3000 + ((!a_kc && !b_kc) ? 4300 : 2950) * nk + 14500      // (weight-gradient form: every fragment through two transposing reads — 768 x 16384 x 3456: 105-114 us on 192 tiles)
                             : id == 6 ? ws_fix + 4000 + wsw_kt * nk + (s > 1 ? 9000 + 4000 * s : 0)
                             : id == 4 ? ws_fix + ((!a_kc && !b_kc) ? ws_kt_w : ws_kt) * nk + (s > 1 ? 6000 + 2200 * s : 0)
                                       : 4500 + 1800 * nk + 7500 + (s > 1 ? bt_fix0 + bt_fix1 * s : 0);
            double clk = rounds * per;
            if (id == 5) {
                // wave-specialised 64 x 64: the latency of one workgroup (two share a CU) or, with many, the CUs' L2 -> LDS feed
                // (fitted to tools/probes/r4_ws64.py / r4_ws64_split.sh: 440 x 768 x 3072 at splits 1 / 2 / 4 / 6 = 16.0 / 12.5 / 11.5 / 14.4 us)
                // (weight-gradient form: every fragment through two transposing reads — 3072 x 768 x 3520 takes 37.9 us = 91 k clocks on 576
                // workgroups: 650 per k-tile, as on the 64-row tiles)
                const double over = wgs > 256 ? (wgs - 256) / 256 : 0, kt64 = (!a_kc && !b_kc && wgs >= 512) ? ws64_kt_w : 420;     // (... once every CU holds two such workgroups — 2048 x 512 x 6976 at split 2, exactly 512, takes 32.4 us = 650 per k-tile; 768 x 768 x 3520 at split 3, 432 workgroups, stays at 14.6 us)
                const double lat = (6500 + kt64 * nk + (s > 1 ? 3000 + 1000 * s : 0)) * (1 + 0.2 * over), thr = wgs * (3000 + kt64 * nk) / 256;
                clk = lat > thr ? lat : thr;
                // (16384-deep reductions stream an operand from HBM: 868 x 512 x 16384 at split 3 takes 42 us against this model's 21 —
                // the 128 x 128 tiles, modelled 2x low as well, take 36-38)
                if (nkt >= 128) clk *= 1.25;
            }
            if (clk < best.clocks) best = BtPlan{id, s, clk};
        }
    }
    if (best.tile < 0 || g_bt_mode >= 0 || only >= 0) return best;
    if (best.clocks >= 0.92 * est_64row(M, N, K, allow_split, !a_kc && !b_kc) * (nkt >= 128 ? 1.25 : 1.0)) return BtPlan{-1, 1, 0.0};     // (the same HBM-stream factor as above)
    return best;
}

template <int BM1, int BN1>
void launch_pair(int id2, dim3 grid, hipStream_t st, const GArgs& p1, const GArgs& p2, int nb1) {
    dim3 block(256);
    if (id2 == 0) hipLaunchKernelGGL((gemm_glds_pair_kernel<BM1, BN1, 64, 64>), grid, block, 0, st, p1, p2, nb1);
    else hipLaunchKernelGGL((gemm_glds_pair_kernel<BM1, BN1, 64, 128>), grid, block, 0, st, p1, p2, nb1);
}

}  // namespace

extern "C" int vitae_gemm_glds_set_bt_tile(int mode) {
    if (mode < -2 || mode > 6 || mode == 1 || mode == 2) return VITAE_ERR_INVALID_ARG;
    g_bt_mode = mode;
    return VITAE_OK;
}

extern "C" int vitae_gemm_glds_bt_choice(int a_kcontig, int b_kcontig, int M, int N, int K) { return bt_plan(M, N, K, a_kcontig, b_kcontig, true).tile; }

extern "C" int vitae_gemm_glds_pick_split_k(int M, int N, int K) {
    // (the forward / dgrad / wgrad forms share one plan: the model does not depend on the operand storage)
    const BtPlan bp = bt_plan(M, N, K, 1, 1, true);
    if (bp.tile >= 0) return bp.split;
    return old_split_rule(M, N, K);
}

// the same for a given operand form (the plan depends on it: transposing fragment reads cost the weight-gradient form more per k-tile)
extern "C" int vitae_gemm_glds_pick_split_k_form(int a_kcontig, int b_kcontig, int M, int N, int K) {
    const BtPlan bp = bt_plan(M, N, K, a_kcontig, b_kcontig, true);
    if (bp.tile >= 0) return bp.split;
    return old_split_rule(M, N, K);
}

extern "C" long vitae_gemm_glds_ws_floats(int M, int N, int split_k) {
    if (split_k <= 1) return 0;
    const Tile t = pick_tile(M, N);
    // whichever family serves the problem: the 128-row padding of the big tiles covers the 64-row one
    const long small = (long)cdiv(M, t.bm) * cdiv(N, t.bn) * t.bm * t.bn, big = (long)cdiv(M, 128) * cdiv(N, 128) * 128 * 128;
    const long wide = (long)cdiv(M, 128) * cdiv(N, 256) * 128 * 256;          // (the 128 x 256 tile of the weight-gradient form)
    long m = small > big ? small : big;
    if (wide > m) m = wide;
    return VITAE_GLDS_TICKETS + m * split_k;
}

static int gemm_glds_launch(int a_kcontig, int b_kcontig, const void* A16, long lda, const void* B16, long ldb,
                            float* C, long ldc, void* C16, long ldc16, int M, int N, int K, const float* bias,
                            const float* residual, long ldr, int epi, float* aux, long ldaux, int accumulate,
                            int split_k, float* splitk_ws, float* out_colsum_accum, void* stream,
                            const BtPlan* forced = nullptr) {
    if (!A16 || !B16 || (!C && !C16) || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    const int aux16 = (epi & VITAE_EPI_AUX_BF16) != 0, auxd = (epi & VITAE_EPI_AUX_DERIV) != 0;
    epi &= ~(VITAE_EPI_AUX_BF16 | VITAE_EPI_AUX_DERIV);
    if (epi != VITAE_EPI_NONE && epi != VITAE_EPI_RELU && !aux) return VITAE_ERR_INVALID_ARG;
    if (auxd && epi != VITAE_EPI_GELU && epi != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;
    if (K % BK) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if ((long)M * (ldc > N ? ldc : N) >= (1L << 31) || (long)M * ldaux >= (1L << 31) || (long)M * ldr >= (1L << 31) ||
        (long)M * ldc16 >= (1L << 31))
        return VITAE_ERR_UNSUPPORTED_SHAPE;   // epilogue uses 32-bit element offsets
    const int a_vec = a_kcontig ? K : M, b_vec = b_kcontig ? K : N;
    if ((a_vec & 7) || (lda & 7) || (b_vec & 7) || (ldb & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)A16 & 15) || ((uintptr_t)B16 & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    // (LDS-DMA pieces carry 32-bit byte offsets from the operand base)
    if ((long)(a_kcontig ? M : K) * lda >= (1L << 30) || (long)(b_kcontig ? N : K) * ldb >= (1L << 30)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (split_k < 1) split_k = 1;
    if (epi == VITAE_EPI_GELU) split_k = 1;
    GArgs p;
    p.A = reinterpret_cast<const __bf16*>(A16); p.lda = lda;
    p.B = reinterpret_cast<const __bf16*>(B16); p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.C16 = reinterpret_cast<__bf16*>(C16); p.ldc16 = ldc16;
    p.M = M; p.N = N; p.K = K;
    int kps = cdiv(cdiv(K, split_k), BK) * BK;
    split_k = cdiv(K, kps);
    if (split_k > 1 && !splitk_ws) return VITAE_ERR_INVALID_ARG;
    p.k_per_split = kps; p.splits = split_k;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux;
    p.epi = epi; p.aux16 = aux16; p.auxd = auxd; p.accumulate = accumulate; p.ws = splitk_ws; p.out_colsum = out_colsum_accum; p.a_rowsum = nullptr;
    p.dbg = g_gemm_dbg;
    const Tile t = pick_tile(M, N);
    p.tiles_m = cdiv(M, t.bm); p.tiles_n = cdiv(N, t.bn);
    p.xcd_m = xcd_by_rows(M, N);
    p.vec_epi = vec_epilogue_ok(p);
    if (!a_kcontig && !b_kcontig) set_sq(p);      // the weight-gradient form (dy^T @ x)
    if (p.vec_epi) {
        const BtPlan bp = forced ? *forced : bt_plan(M, N, K, a_kcontig, b_kcontig, epi != VITAE_EPI_GELU);
        if (bp.tile >= 0 && bp.split == split_k) {
            p.splits = split_k;
            return bt_launch(p, a_kcontig, b_kcontig, bp.tile, (hipStream_t)stream);
        }
    }
    if (split_k > 1 && (long)p.tiles_m * p.tiles_n > VITAE_GLDS_TICKETS) return VITAE_ERR_UNSUPPORTED_SHAPE;
    dim3 grid(glds_blocks(p), 1, split_k);
    hipStream_t st = (hipStream_t)stream;
    static const int pipe_max = getenv("VITAE_GLDS_PIPE_MAX_WGS") ? atoi(getenv("VITAE_GLDS_PIPE_MAX_WGS")) : 512;
    if (t.id == 0 && a_kcontig && b_kcontig && (long)grid.x * split_k <= pipe_max) {
        hipLaunchKernelGGL(gemm_glds_pipe_kernel, grid, dim3(256), 0, st, p);
        return vitae_launch_status();
    }
    if (t.id == 1) launch<64, 128>(p, a_kcontig != 0, b_kcontig != 0, grid, st);
    else launch<64, 64>(p, a_kcontig != 0, b_kcontig != 0, grid, st);
    return vitae_launch_status();
}

extern "C" int vitae_gemm_glds(int a_kcontig, int b_kcontig, const void* A16, long lda, const void* B16, long ldb,
                               float* C, long ldc, void* C16, long ldc16, int M, int N, int K, const float* bias,
                               const float* residual, long ldr, int epi, float* aux, long ldaux, int accumulate,
                               int split_k, float* splitk_ws, float* out_colsum_accum, void* stream) {
    return gemm_glds_launch(a_kcontig, b_kcontig, A16, lda, B16, ldb, C, ldc, C16, ldc16, M, N, K, bias, residual, ldr, epi, aux, ldaux,
                            accumulate, split_k, splitk_ws, out_colsum_accum, stream);
}

// plan of the two-plane launch: the big tiles see an ordinary forward problem with a reduction of 2 K; tile 5 / no big tile: the
// two-plane workgroup of gemm_bt.hip (ws64_w2_launch) on the K-deep reduction
static BtPlan w2_plan(int M, int N, int K, bool allow_split) {
    BtPlan bp = bt_plan(M, N, 2 * K, 1, 1, allow_split);
    if (bp.tile == 5 || bp.tile < 0) {
        bp.tile = 5;
        const BtPlan b1 = bt_plan(M, N, K, 1, 1, allow_split);
        bp.split = (b1.tile == 5) ? b1.split : 1;
    }
    return bp;
}

extern "C" int vitae_gemm_glds_w2_pick_split_k(int M, int N, int K) { return w2_plan(M, N, K, true).split; }

extern "C" int vitae_gemm_glds_w2(const void* A16, long lda, const void* B16_hilo, float* C, long ldc, void* C16,
                                  long ldc16, int M, int N, int K, const float* bias, const float* residual, long ldr, int epi, float* aux,
                                  long ldaux, int accumulate, int split_k, float* splitk_ws, float* out_colsum_accum, void* stream) {
    if (!A16 || !B16_hilo || (!C && !C16) || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    const int aux16 = (epi & VITAE_EPI_AUX_BF16) != 0, auxd = (epi & VITAE_EPI_AUX_DERIV) != 0;
    epi &= ~(VITAE_EPI_AUX_BF16 | VITAE_EPI_AUX_DERIV);
    if (epi != VITAE_EPI_NONE && epi != VITAE_EPI_RELU && !aux) return VITAE_ERR_INVALID_ARG;
    if (auxd && epi != VITAE_EPI_GELU && epi != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;
    if ((K % BK) || K < 2 * BK || (lda & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if ((long)M * (ldc > N ? ldc : N) >= (1L << 31) || (long)M * ldaux >= (1L << 31) || (long)M * ldr >= (1L << 31) || (long)M * ldc16 >= (1L << 31))
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)A16 & 15) || ((uintptr_t)B16_hilo & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if ((long)M * lda >= (1L << 30) || (long)N * 2 * K >= (1L << 30)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (split_k < 1 || epi == VITAE_EPI_GELU) split_k = 1;
    if (split_k > 1 && !splitk_ws) return VITAE_ERR_INVALID_ARG;
    GArgs p;
    p.A = reinterpret_cast<const __bf16*>(A16); p.lda = lda;
    p.B = reinterpret_cast<const __bf16*>(B16_hilo); p.ldb = 2L * K;
    p.C = C; p.ldc = ldc; p.C16 = reinterpret_cast<__bf16*>(C16); p.ldc16 = ldc16;
    p.M = M; p.N = N; p.K = K; p.k_per_split = K; p.splits = split_k;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux;
    p.epi = epi; p.aux16 = aux16; p.auxd = auxd; p.accumulate = accumulate; p.ws = splitk_ws; p.out_colsum = out_colsum_accum; p.a_rowsum = nullptr;
    p.dbg = nullptr; p.tiles_m = 0; p.tiles_n = 0;
    p.xcd_m = xcd_by_rows(M, N);
    p.vec_epi = vec_epilogue_ok(p);
    if (!p.vec_epi) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const BtPlan bp = w2_plan(M, N, K, epi != VITAE_EPI_GELU);
    if (bp.tile != 5 && bp.split == split_k) {
        // a big tile on the reduction of 2 K: [W hi | W lo] is one k-contiguous operand, the activations wrap
        p.K = 2 * K; p.a_kwrap = K;
        return bt_launch(p, 1, 1, bp.tile, (hipStream_t)stream);
    }
    p.B2 = p.B + K;                    // the lo plane: same rows, K columns further
    return ws64_w2_launch(p, (hipStream_t)stream);
}

// fp32x3 on the wave-specialised 64 x 64 workgroup (csrc/gemm_bt.hip: gemm_wsx3_kernel): the contract of vitae_gemm with fp32
// operands multiplied as bf16 hi + lo pairs; the producer waves split them while they stage.  K any multiple of 4; in-launch
// split-K with the workspace layout of vitae_gemm_glds (vitae_gemm_glds_ws_floats(M, N, split_k) floats, first VITAE_GLDS_TICKETS
// words zero before the first use).  a_rowsum_accum (weight-gradient form, a_kcontig = b_kcontig = 0): [M] += sum_k A(m, k), the
// bias gradient colsum(dy) beside dW = dy^T x.  VITAE_ERR_UNSUPPORTED_SHAPE: the caller keeps vitae_gemm_bf16x3.
extern "C" int vitae_gemm_wsx3_pick_split_k(int M, int N, int K) {
    const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
    const int nk = cdiv(K, BK);
    static const int want = getenv("VITAE_X3WS_SPLIT_WGS") ? atoi(getenv("VITAE_X3WS_SPLIT_WGS")) : wsx3_slots() * 7 / 8;
    long s = want / (tiles > 0 ? tiles : 1);
    if (s > 8) s = 8;
    while (s > 1 && nk / s < 4) --s;
    if (tiles > VITAE_GLDS_TICKETS) s = 1;
    return s < 1 ? 1 : (int)s;
}

extern "C" int vitae_gemm_wsx3(int a_kcontig, int b_kcontig, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                               int M, int N, int K, const float* bias, const float* residual, long ldr, int epi, float* aux,
                               long ldaux, int accumulate, int split_k, float* splitk_ws, float* out_colsum_accum,
                               float* a_rowsum_accum, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    const int auxd = (epi & VITAE_EPI_AUX_DERIV) != 0;
    epi &= ~VITAE_EPI_AUX_DERIV;
    if (epi != VITAE_EPI_NONE && epi != VITAE_EPI_RELU && !aux) return VITAE_ERR_INVALID_ARG;
    if (auxd && epi != VITAE_EPI_GELU && epi != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;
    if (epi & VITAE_EPI_AUX_BF16) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const int a_vec = a_kcontig ? K : M, b_vec = b_kcontig ? K : N;
    if ((a_vec & 3) || (lda & 3) || (b_vec & 3) || (ldb & 3) || (K & 3) || M < 8 || N < 8) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if ((long)M * (ldc > N ? ldc : N) >= (1L << 31) || (long)M * ldaux >= (1L << 31) || (long)M * ldr >= (1L << 31)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (split_k < 1 || epi == VITAE_EPI_GELU) split_k = 1;
    if (split_k > 1 && !splitk_ws) return VITAE_ERR_INVALID_ARG;
    GArgs p;
    p.A = reinterpret_cast<const __bf16*>(A); p.lda = lda;      // (floats behind these two: gemm_wsx3_body casts them back)
    p.B = reinterpret_cast<const __bf16*>(B); p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.C16 = nullptr; p.ldc16 = 0;
    p.M = M; p.N = N; p.K = K; p.k_per_split = K; p.splits = split_k;
    p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux;
    p.epi = epi; p.aux16 = 0; p.auxd = auxd; p.exact = 1; p.accumulate = accumulate; p.ws = splitk_ws; p.out_colsum = out_colsum_accum;
    p.a_rowsum = a_rowsum_accum; p.dbg = nullptr;
    p.tiles_m = 0; p.tiles_n = 0;
    p.xcd_m = xcd_by_rows(M, N);
    p.vec_epi = vec_epilogue_ok(p);
    if (!a_kcontig && !b_kcontig) set_sq(p);
    return wsx3_launch(p, a_kcontig, b_kcontig, (hipStream_t)stream);
}

// Backward of one Linear on bf16 operands in ONE launch: dx[M,K] = epi(dy16[M,N] @ W16[N,K]) (fp32 dx and/or
// bf16 dx16; optional colsum of the result = bias gradient of the layer in front), dW[N,K] (+)= dy16^T @ x16.
// Mpad = token count rounded up to 64: rows M..Mpad-1 of dy16 and x16 must be zero (wgrad reduces over them).
extern "C" int vitae_linear_bwd_pair_pick_split_k(int M, int Mpad, int N, int K) {
    // cut the dgrad reduction (N / 64 k-tiles on only ceil(M/64) * K/64 workgroups) down to about the length of the
    // wgrad's (Mpad / 64 k-tiles), which otherwise finishes long before it
    static const int target = getenv("VITAE_PAIR_SPLIT_TARGET") ? atoi(getenv("VITAE_PAIR_SPLIT_TARGET")) : 10;
    if (target <= 0) return 1;
    const int ks1 = N / BK, ks2 = Mpad / BK;
    int s = (ks1 + (ks2 > target ? ks2 : target) / 2) / (ks2 > target ? ks2 : target);
    if (s > 8) s = 8;
    while (s > 1 && ks1 / s < 4) --s;
    const Tile t1 = pick_tile(M, K);
    if ((long)cdiv(M, t1.bm) * cdiv(K, t1.bn) > VITAE_GLDS_TICKETS) s = 1;
    return s < 1 ? 1 : s;
}

extern "C" int vitae_linear_bwd_pair_glds(const void* dy16, const void* w16, const void* x16, float* dx, void* dx16,
                                          float* dw, void* dw16, int M, int Mpad, int N, int K, int epi, float* aux,
                                          float* dx_colsum_accum, float* dy_colsum_accum, int dx_accumulate, int dw_accumulate,
                                          int split_k, float* splitk_ws, long splitk_ws_floats, void* stream) {
    if (!dy16 || !w16 || (!x16 && dw) || (!dx && !dx16) || M <= 0 || N <= 0 || K <= 0) return VITAE_ERR_INVALID_ARG;
    const int aux16 = (epi & VITAE_EPI_AUX_BF16) != 0, auxd = (epi & VITAE_EPI_AUX_DERIV) != 0;
    epi &= ~(VITAE_EPI_AUX_BF16 | VITAE_EPI_AUX_DERIV);
    if (epi != VITAE_EPI_NONE && !aux) return VITAE_ERR_INVALID_ARG;
    if (auxd && epi != VITAE_EPI_DGELU) return VITAE_ERR_INVALID_ARG;
    const int epi_flags = (aux16 ? VITAE_EPI_AUX_BF16 : 0) | (auxd ? VITAE_EPI_AUX_DERIV : 0);
    if (dx_accumulate && !dx) return VITAE_ERR_INVALID_ARG;
    if ((N % BK) || (Mpad % BK) || (K & 7) || Mpad < M) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if ((long)(M > N ? M : N) * K >= (1L << 31)) return VITAE_ERR_UNSUPPORTED_SHAPE;   // 32-bit epilogue offsets
    if ((long)Mpad * N >= (1L << 30) || (long)Mpad * K >= (1L << 30) || (long)N * K >= (1L << 30)) return VITAE_ERR_UNSUPPORTED_SHAPE;   // 32-bit DMA byte offsets
    // what the split-K plans of either half may use: exactly what the caller says the workspace holds
    const long cap = splitk_ws && splitk_ws_floats > 0 ? splitk_ws_floats : 0;
    if (split_k > 1 && vitae_gemm_glds_ws_floats(M, K, split_k) > cap) return VITAE_ERR_INVALID_ARG;
    if (((uintptr_t)dy16 & 15) || ((uintptr_t)w16 & 15) || ((uintptr_t)x16 & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    if (!dw) {
        // dW deferred (vitae_wgrad_group_bt collects a block's weight gradients into one launch): the input gradient alone, through
        // the planner of vitae_gemm_glds
        const BtPlan pd = bt_plan(M, K, N, 1, 0, true, cap);
        int sp = pd.tile >= 0 ? pd.split : old_split_rule(M, K, N);
        while (pd.tile < 0 && sp > 1 && vitae_gemm_glds_ws_floats(M, K, sp) > cap) --sp;
        return gemm_glds_launch(1, 0, dy16, N, w16, K, dx, K, dx16, K, M, K, N, nullptr, nullptr, 0, epi | epi_flags,
                                aux, K, dx_accumulate, sp, splitk_ws, dx_colsum_accum, stream, &pd);
    }
    // both halves as wave-specialised 64 x 64 workgroups of ONE launch (gemm_bt.hip: gemm_ws64_pair_kernel), the input gradient cut `split` ways
    auto ws64_pair = [&](int split) -> int {
        GArgs p1, p2;
        p1.A = reinterpret_cast<const __bf16*>(dy16); p1.lda = N;
        p1.B = reinterpret_cast<const __bf16*>(w16); p1.ldb = K;
        p1.C = dx; p1.ldc = K; p1.C16 = reinterpret_cast<__bf16*>(dx16); p1.ldc16 = K;
        p1.M = M; p1.N = K; p1.K = N; p1.k_per_split = N; p1.splits = split;
        p1.bias = nullptr; p1.residual = nullptr; p1.ldr = 0; p1.aux = aux; p1.ldaux = K; p1.epi = epi; p1.aux16 = aux16; p1.auxd = auxd;
        p1.accumulate = dx_accumulate != 0; p1.ws = splitk_ws; p1.out_colsum = dx_colsum_accum; p1.a_rowsum = nullptr; p1.dbg = g_gemm_dbg;
        p1.xcd_m = xcd_by_rows(M, K); p1.tiles_m = 0; p1.tiles_n = 0;
        p1.vec_epi = vec_epilogue_ok(p1);
        p2.A = reinterpret_cast<const __bf16*>(dy16); p2.lda = N;
        p2.B = reinterpret_cast<const __bf16*>(x16); p2.ldb = K;
        p2.C = dw; p2.ldc = K; p2.C16 = reinterpret_cast<__bf16*>(dw16); p2.ldc16 = K;
        p2.M = N; p2.N = K; p2.K = Mpad; p2.k_per_split = Mpad; p2.splits = 1;
        p2.bias = nullptr; p2.residual = nullptr; p2.ldr = 0; p2.aux = nullptr; p2.ldaux = 0; p2.epi = VITAE_EPI_NONE;
        p2.accumulate = dw_accumulate; p2.ws = nullptr; p2.out_colsum = nullptr; p2.a_rowsum = dy_colsum_accum; p2.dbg = g_gemm_dbg;
        set_sq(p2);
        p2.xcd_m = xcd_by_rows(N, K); p2.tiles_m = 0; p2.tiles_n = 0;
        p2.vec_epi = vec_epilogue_ok(p2);
        if (!p1.vec_epi || !p2.vec_epi) return VITAE_ERR_UNSUPPORTED_SHAPE;
        // VITAE_WS64Q=1: the persistent form (round 6: csrc/gemm_bt.hip gemm_ws64q_pair_kernel — correct, measured 6-10 % SLOWER than
        // one tile per workgroup on the batch-4 / batch-8 pair launches, so it is opt-in; read per call so that a test can switch it)
        const char* wsq_env = getenv("VITAE_WS64Q");
        const int wsq = wsq_env ? atoi(wsq_env) : 0;
        const int rc = wsq ? ws64q_pair_launch(p1, p2, (hipStream_t)stream) : VITAE_ERR_UNSUPPORTED_SHAPE;
        return rc != VITAE_ERR_UNSUPPORTED_SHAPE ? rc : ws64_pair_launch(p1, p2, (hipStream_t)stream);
    };
    if (g_bt_mode == -1 || g_bt_mode == 5) {
        // Few token rows: the planner puts BOTH halves on that tile
        const BtPlan pd = bt_plan(M, K, N, 1, 0, true, cap), pw = bt_plan(N, K, Mpad, 0, 0, false, cap);
        if (pd.tile == 5 && pw.tile == 5) {
            const int rc = ws64_pair(pd.split);
            if (rc != VITAE_ERR_UNSUPPORTED_SHAPE) return rc;
        }
    }
    if (g_bt_mode != -2) {
        // When the big tiles serve either half, the halves go out as two launches of their own (each fills the chip; the
        // paired launch exists to double the resident workgroups of two SMALL problems): dx = epi(dy16 @ W16) and
        // dW (+)= dy16^T @ x16 through the planner of vitae_gemm_glds, bias gradient by the column-sum kernel.
        const BtPlan pd = bt_plan(M, K, N, 1, 0, true, cap, false), pw = bt_plan(N, K, Mpad, 0, 0, true, cap, false);
        if (pd.tile >= 0 || pw.tile >= 0) {
            // each half: the plan's split when the plan is a big tile, else the 64-row family's own rule, shrunk to the workspace
            auto fit = [&](const BtPlan& bp, int m, int n, int k) {
                if (bp.tile >= 0) return bp.split;
                int sp = old_split_rule(m, n, k);
                while (sp > 1 && vitae_gemm_glds_ws_floats(m, n, sp) > cap) --sp;
                return sp;
            };
            int rc = gemm_glds_launch(1, 0, dy16, N, w16, K, dx, K, dx16, K, M, K, N, nullptr, nullptr, 0, epi | epi_flags,
                                      aux, K, dx_accumulate, fit(pd, M, K, N), splitk_ws, dx_colsum_accum, stream, &pd);
            if (rc != VITAE_OK) return rc;
            rc = gemm_glds_launch(0, 0, dy16, N, x16, K, dw, K, dw16, K, N, K, Mpad, nullptr, nullptr, 0, VITAE_EPI_NONE, nullptr, 0,
                                  dw_accumulate, fit(pw, N, K, Mpad), splitk_ws, nullptr, stream, &pw);
            if (rc != VITAE_OK) return rc;
            if (dy_colsum_accum)
                launch_colsum_bf16(dy16, dy_colsum_accum, M, N, (hipStream_t)stream);
            return vitae_launch_status();
        }
    }
    static const int pair_ws64_rest = getenv("VITAE_PAIR_WS64_REST") ? atoi(getenv("VITAE_PAIR_WS64_REST")) : 1;
    if (g_bt_mode == -1 && pair_ws64_rest) {
        // Neither the big tiles nor (for both halves) the planner's 64 x 64 choice: what used to fall through to the 64-row pair kernel
        // below.  Round 6, ViT-L/16 on 128^3 (BASELINE config 4, 516 token rows x 1024 / 4096): the wave-specialised pair takes 26.7 /
        // 27.6 us on fc2 / fc1 where that kernel takes 38.3 / 36.7 (tools/pair_bench.py model=L128) — the planner had priced the weight
        // gradient alone on 128 x 128 tiles a little cheaper, which un-paired nothing and only kept the faster pair out.
        const BtPlan p5 = bt_plan(M, K, N, 1, 0, true, cap, true, 5);
        if (p5.tile == 5) {
            const int rc = ws64_pair(p5.split);
            if (rc != VITAE_ERR_UNSUPPORTED_SHAPE) return rc;
        }
    }
    GArgs p1, p2;
    p1.A = reinterpret_cast<const __bf16*>(dy16); p1.lda = N;
    p1.B = reinterpret_cast<const __bf16*>(w16); p1.ldb = K;
    p1.C = dx; p1.ldc = K; p1.C16 = reinterpret_cast<__bf16*>(dx16); p1.ldc16 = K;
    if (split_k < 1 || !splitk_ws) split_k = 1;
    const int kps = cdiv(cdiv(N, split_k), BK) * BK;
    split_k = cdiv(N, kps);
    p1.M = M; p1.N = K; p1.K = N; p1.k_per_split = kps; p1.splits = split_k;
    p1.bias = nullptr; p1.residual = nullptr; p1.ldr = 0; p1.aux = aux; p1.ldaux = K; p1.epi = epi; p1.aux16 = aux16; p1.auxd = auxd; p1.accumulate = dx_accumulate != 0;
    p1.ws = splitk_ws; p1.out_colsum = dx_colsum_accum; p1.a_rowsum = nullptr; p1.dbg = nullptr; p2.dbg = nullptr;
    const Tile t1 = pick_tile(M, K);
    p1.tiles_m = cdiv(M, t1.bm); p1.tiles_n = cdiv(K, t1.bn);
    p1.xcd_m = xcd_by_rows(M, K);
    p1.vec_epi = vec_epilogue_ok(p1);
    p2.A = reinterpret_cast<const __bf16*>(dy16); p2.lda = N;
    p2.B = reinterpret_cast<const __bf16*>(x16); p2.ldb = K;
    p2.C = dw; p2.ldc = K; p2.C16 = reinterpret_cast<__bf16*>(dw16); p2.ldc16 = K;
    p2.M = N; p2.N = K; p2.K = Mpad; p2.k_per_split = Mpad; p2.splits = 1;
    p2.bias = nullptr; p2.residual = nullptr; p2.ldr = 0; p2.aux = nullptr; p2.ldaux = 0; p2.epi = VITAE_EPI_NONE;
    p2.accumulate = dw_accumulate; p2.ws = nullptr; p2.out_colsum = nullptr; p2.a_rowsum = dy_colsum_accum;
    set_sq(p2);
    const Tile t2 = pick_tile(N, K);
    p2.tiles_m = cdiv(N, t2.bm); p2.tiles_n = cdiv(K, t2.bn);
    p2.xcd_m = xcd_by_rows(N, K);
    p2.vec_epi = vec_epilogue_ok(p2);
    const int nb1 = glds_blocks(p1), nb2 = glds_blocks(p2);
    dim3 grid(nb1 * split_k + nb2);
    hipStream_t st = (hipStream_t)stream;
    if (dy_colsum_accum && t1.id == 0 && t2.id == 0) {
        // bias gradient colsum(dy) on the wgrad workgroups: one extra MFMA against a ones operand
        hipLaunchKernelGGL((gemm_glds_pair_kernel<64, 64, 64, 64, true>), grid, dim3(256), 0, st, p1, p2, nb1);
        return vitae_launch_status();
    }
    p2.a_rowsum = nullptr;
    if (t1.id == 1) launch_pair<64, 128>(t2.id, grid, st, p1, p2, nb1);
    else launch_pair<64, 64>(t2.id, grid, st, p1, p2, nb1);
    if (dy_colsum_accum)
        launch_colsum_bf16(dy16, dy_colsum_accum, M, N, st);
    return vitae_launch_status();
}

// Weight gradients of up to four Linears with the same token count in ONE launch of 128x128 tiles (csrc/gemm_bt.hip: ping-pong
// workgroups, two per CU, or wave-specialised ones, one per CU): for i < n,
// dw[i][N[i], K[i]] (+)= dy16[i][Mpad, N[i]]^T x16[i][Mpad, K[i]]; optional bf16 copies dw16[i]; optional bias gradients
// dy_colsum[i][N[i]] += column sums of the first M rows of dy16[i] (one small launch each).  Pointer arrays and N / K on the HOST.
// Workgroup kind and split of the reduction are chosen for the SUM of the tiles.
extern "C" int vitae_wgrad_group_bt(int n, const void* const* dy16, const void* const* x16, float* const* dw, void* const* dw16,
                                    float* const* dy_colsum, const int* N, const int* K, int M, int Mpad, int dw_accumulate,
                                    float* splitk_ws, long splitk_ws_floats, void* stream) {
    if (n < 1 || n > 4 || !dy16 || !x16 || !dw || !N || !K || M <= 0 || Mpad < M) return VITAE_ERR_INVALID_ARG;
    if ((Mpad % BK) || Mpad < 4 * BK) return VITAE_ERR_UNSUPPORTED_SHAPE;
    GArgs ps[4];
    long tiles = 0, tiles2 = 0;      // 128 x 128 tiles; 128 x 256 tiles
    for (int i = 0; i < n; ++i) {
        if (!dy16[i] || !x16[i] || !dw[i] || N[i] <= 0 || K[i] <= 0) return VITAE_ERR_INVALID_ARG;
        if ((N[i] & 7) || (K[i] & 7) || (long)N[i] * K[i] >= (1L << 31) || (long)Mpad * N[i] >= (1L << 30) || (long)Mpad * K[i] >= (1L << 30)) return VITAE_ERR_UNSUPPORTED_SHAPE;
        if (((uintptr_t)dy16[i] & 15) || ((uintptr_t)x16[i] & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
        GArgs& p = ps[i];
        p.A = reinterpret_cast<const __bf16*>(dy16[i]); p.lda = N[i];
        p.B = reinterpret_cast<const __bf16*>(x16[i]); p.ldb = K[i];
        p.C = dw[i]; p.ldc = K[i];
        p.C16 = reinterpret_cast<__bf16*>(dw16 ? dw16[i] : nullptr); p.ldc16 = K[i];
        p.M = N[i]; p.N = K[i]; p.K = Mpad; p.k_per_split = Mpad; p.splits = 1;
        p.bias = nullptr; p.residual = nullptr; p.ldr = 0; p.aux = nullptr; p.ldaux = 0; p.epi = VITAE_EPI_NONE;
        p.accumulate = dw_accumulate; p.ws = splitk_ws; p.out_colsum = nullptr; p.a_rowsum = nullptr; p.dbg = nullptr;
        set_sq(p);
        p.xcd_m = xcd_by_rows(N[i], K[i]);
        p.tiles_m = cdiv(N[i], 128); p.tiles_n = cdiv(K[i], 128);
        p.vec_epi = vec_epilogue_ok(p);
        if (!p.vec_epi) return VITAE_ERR_UNSUPPORTED_SHAPE;
        tiles += (long)p.tiles_m * p.tiles_n;
        tiles2 += (long)cdiv(N[i], 128) * cdiv(K[i], 256);
    }
    // Two workgroup kinds (csrc/gemm_bt.hip), each with its own split of the reduction; the cheaper by the clocks fitted to
    // tools/wgrad_group_bench.py wins:
    //   ping-pong 128 x 128, two per CU: 12000 + 2650 per k-tile (both co-resident workgroups advance one) + the split fix-up
    //     (batch 32: encoder block, 432 tiles x 55 k-tiles unsplit, 66 us; decoder block, 192 x 109 at split 2, 73 us);
    //   wave-specialised 128 x 128, one per CU: 14000 + 900 per k-tile + fix-up;
    //   wave-specialised 128 x 256, one per CU: 18000 + 1350 per k-tile + fix-up (half the tiles: an encoder block's 216 are ONE round).
    const long cap = splitk_ws && splitk_ws_floats > 0 ? splitk_ws_floats : 0;
    const int nkt = Mpad / BK;
    auto split_ok = [&](int s, int kind) {
        if (s == 1) return true;
        const int kps = cdiv(cdiv(Mpad, s), BK) * BK;
        const long nt = kind == 2 ? tiles2 : tiles, area = kind == 2 ? 128 * 256 : 128 * 128;
        return cdiv(Mpad, kps) == s && kps >= 8 * BK && Mpad - (s - 1) * kps >= 2 * BK && nt <= VITAE_GLDS_TICKETS &&
               VITAE_GLDS_TICKETS + nt * s * area <= cap;
    };
    static const int env_ws = getenv("VITAE_WGRAD_GROUP_WS") ? atoi(getenv("VITAE_WGRAD_GROUP_WS")) : -1;          // 0 / 1: that kind only
    const int force_ws = g_bt_mode == 3 ? 0 : g_bt_mode == 4 ? 1 : g_bt_mode == 6 ? 2 : env_ws;                                       // (a forced tile 3 / 4 — tests, tools — forces the kind)
    static const int force_split = getenv("VITAE_WGRAD_GROUP_SPLIT") ? atoi(getenv("VITAE_WGRAD_GROUP_SPLIT")) : 0;
    static const double ws_kt = getenv("VITAE_WGRAD_GROUP_WS_KT") ? atof(getenv("VITAE_WGRAD_GROUP_WS_KT")) : 900.0;
    static const double bt_kt = getenv("VITAE_WGRAD_GROUP_BT_KT") ? atof(getenv("VITAE_WGRAD_GROUP_BT_KT")) : 2650.0;
    static const double wsw_kt = getenv("VITAE_WGRAD_GROUP_WSW_KT") ? atof(getenv("VITAE_WGRAD_GROUP_WSW_KT")) : 1300.0;
    // The wave-specialised kinds are bound by whichever is longer: a CU's own L2 -> LDS rate (900 / 1300 clocks per k-tile of 32 /
    // 48 KB) over its rounds, or the CHIP's — a grouped launch keeps every CU streaming and the eight L2s deliver ~13 TB/s between
    // them (5400 B/clk: tools/wgrad_group_bench.py, encoder block at batch 32: 432 x 55 x 32 KB in 62 us, 216 x 55 x 48 KB in 53).
    static const double chip_bpc = getenv("VITAE_WGRAD_GROUP_CHIP_BPC") ? atof(getenv("VITAE_WGRAD_GROUP_CHIP_BPC")) : 5400.0;
    int split = 1, ws_tile = 0;
    double best = 1e30;
    for (int kind = 0; kind < 3; ++kind) {
        if (force_ws >= 0 && kind != force_ws) continue;
        for (int s = 1; s <= (kind == 2 ? 4 : 8); ++s) {
            if ((force_split > 0 && s != force_split) || !split_ok(s, kind)) continue;
            const double nk = (double)nkt / s, wgs = (double)(kind == 2 ? tiles2 : tiles) * s, rounds = (double)cdiv((long)wgs, kind ? 256L : 512L);
            double clk;
            if (kind == 0) clk = rounds * (12000 + bt_kt * nk + (s > 1 ? 9000 + 4000 * s : 0));
            else {
                const double own = rounds * (kind == 2 ? wsw_kt : ws_kt) * nk, chip = wgs * nk * (kind == 2 ? 49152.0 : 32768.0) / chip_bpc;
                // fixed part (fill, tail): 20 k clocks for the 128 x 256 tile; for 128 x 128 it shrinks with the length of the k-loop
                // (5.6 k / 14.7 k / 22 k at 55 / 28 / 14 k-tiles: a tile's tail runs under its successor's k-loop)
                const double fix1 = 27500 - 400 * nk > 5000 ? 27500 - 400 * nk : 5000;
                clk = (own > chip ? own : chip) + (kind == 2 ? 20000 : fix1) + (s > 1 ? (kind == 2 ? 9000 + 4000 * s : 6000 + 2200 * s) : 0);
                if (kind == 2) clk *= 1.03;        // (ties go to the smaller tile)
            }
            if (clk < best) { best = clk; split = s; ws_tile = kind; }
        }
    }
    if (best >= 1e30) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const int rc = bt_wgrad_group_launch(ps, n, split, (hipStream_t)stream, ws_tile);
    if (rc != VITAE_OK) return rc;
    if (dy_colsum)
        for (int i = 0; i < n; ++i)
            if (dy_colsum[i]) launch_colsum_bf16(dy16[i], dy_colsum[i], M, N[i], (hipStream_t)stream);
    return vitae_launch_status();
}

// profiling hook (tools/gemm_phase_probe.py): with a device buffer of 8 long long per workgroup set, vitae_gemm_glds launches
// record shader-clock stamps at their phase boundaries; NULL switches it off
extern "C" int vitae_gemm_glds_set_debug(void* buf) { g_gemm_dbg = reinterpret_cast<long long*>(buf); return VITAE_OK; }

// While set (NULL clears): every weight-gradient launch of this file — the wgrad half of vitae_linear_bwd_pair_glds, and
// vitae_gemm_glds in its dy^T @ x form — adds the sum of squares of the gradient tile it stores to *slot (a double: the caller's
// acc[VITAE_ACC_GRADSQ]).  Process-global launch-time state of the (single) thread that issues the step; captured graphs keep
// the value it had at capture.  Not for split-K wgrads (the paired launch never splits its wgrad half).
extern "C" int vitae_gemm_glds_set_wgrad_sqnorm(double* slot) { g_wgrad_sqacc = slot; g_wgrad_sq_mask = 0; g_wgrad_sq_stride = 0; return VITAE_OK; }
// The same over n_slots (a power of two) addresses `stride` doubles apart: workgroup b adds to slots[(b mod n_slots) * stride].  Double
// atomics on ONE address retire one per ~10 ns whoever issues them (tools/pair_bench.py, round 6: the 576 weight-gradient workgroups of
// an encoder fc2 launch spent 5.8 of their 16 us there); the step's accumulator block carries VITAE_ACC_SQ_SLOTS such slots and
// vitae_grad_norm_finalize / vitae_opt_tail sum them.
extern "C" int vitae_gemm_glds_set_wgrad_sqnorm_spread(double* slots, int n_slots, int stride) {
    if (slots && (n_slots < 1 || (n_slots & (n_slots - 1)) || stride < 1)) return VITAE_ERR_INVALID_ARG;
    g_wgrad_sqacc = slots; g_wgrad_sq_mask = slots ? n_slots - 1 : 0; g_wgrad_sq_stride = slots ? stride : 0;
    return VITAE_OK;
}
