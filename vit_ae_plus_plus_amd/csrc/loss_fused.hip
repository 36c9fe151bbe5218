// The whole loss chain on the prediction — forward sums AND the gradient — in ONE pass over it (C = 4 channels):
//   masked-voxel reconstruction MSE        model/vit_autoenc.py:226-227   (+ its gradient)
//   3D Sobel magnitude, channel sum         model/model_utils/sobel_filter.py:37-45, model/vit_autoenc.py:221-223
//   edge-map MSE against the target's map   model/vit_autoenc.py:224-225   (+ its gradient through the Sobel magnitude)
//   dpred = mask 2 g_recon (pred - img) / (P mask.sum()) + d(edge mse)/d pred              (fp32 and bf16)
// Both MSEs are linear in their upstream gradient, so the backward needs no scalar from the forward and the two meet in one
// kernel: the unpatchified prediction (57 MB at batch 4) and its edge map never exist in HBM, and one launch replaces
// loss_fwd_fused_kernel + loss_bwd_fused_kernel (csrc/loss.hip: 80 + 149 us on the step's main chain).
//
// Why this shape on MI355X.  The two-kernel version was LDS-bandwidth bound: every thread fetched the 3x3 neighbourhoods of its
// stencils from LDS (196 B of LDS traffic per output value; profiles/round3_loss_pmc.txt).  Here the three directions of the 3x3x3
// stencils are served by three different mechanisms, none of which re-reads a neighbourhood:
//   x: the 64 lanes of a wave are 64 consecutive x — neighbours come from DPP wave shifts (v_add_f32_dpp ... wave_shr:1), no memory;
//   y: a thread MARCHES along y and keeps the two previous rows of every partial result in registers;
//   z: the NW waves of a workgroup are NW consecutive z-planes; the forward stencil loads the three planes it needs straight from
//      global memory (16 bytes = the four channels of a voxel, patchify order), the transposed stencil of the backward exchanges two
//      16-byte values per thread and row through LDS (double-buffered, one barrier per row).
// The four channels of a voxel are processed together (the edge map sums the channels' magnitudes: model_utils/sobel_filter.py:45),
// so E_pred, its error against the target's map and the per-channel gradient field are all formed in registers.
// Halo: lanes 0, 1, 62, 63 and the first / last wave compute halo values only (outputs: at most 60 x by NW - 2 planes per workgroup).
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "vitae_hip.h"

namespace {

#ifndef VITAE_LOSS_NW
#define VITAE_LOSS_NW 8
#endif
#ifndef VITAE_LOSS_MINW
#define VITAE_LOSS_MINW 2        // waves per SIMD the register budget is cut for (2: 174 VGPRs, no spills; 4: 128 with 64 spilled)
#endif
constexpr int NW = VITAE_LOSS_NW;      // waves = z-planes per workgroup (NW - 2 of them produce outputs)
#ifndef VITAE_LOSS_ABLATE
#define VITAE_LOSS_ABLATE 0          // timing ablations (wrong results): 1 no barrier / LDS exchange, 2 no global loads, 3 no stores
#endif
#ifndef VITAE_LOSS_DEPTH
#define VITAE_LOSS_DEPTH 4           // load slots of the march = steps a load flies ahead of its use (2 or 4; 18 VGPRs per slot: 175 / 240 VGPRs;
                                     // 4: batch 32 475 -> 453 us, batch 4 72 -> 66)
#endif
constexpr int NT = 64 * NW;
constexpr int XO_MAX = 60;             // output columns of a 64-lane row
constexpr int LD = VITAE_LOSS_DEPTH;
static_assert(LD == 2 || LD == 4, "the march's parity ring is two deep: load slots 2 or 4");

// Workgroup -> piece of the volume.  Pieces are numbered x-tile fastest, then y-segment, z-tile, batch element; neighbours in that
// order share their halo planes / rows / columns (a z-tile reads NW + 2 planes for NW - 2 outputs).  The dispatcher deals
// consecutive workgroup ids round-robin over the 8 XCDs, each with an L2 of its own, so with the identity map every halo is fetched
// from memory once per workgroup (FETCH_SIZE 2.0x the inputs for the loss kernel, 3.0x for the target's).  Here XCD k (ids = k mod 8)
// walks the CONTIGUOUS chunk k of the pieces in dispatch order: the pieces in flight on one XCD are neighbours and find each
// other's halos in their L2.  (chunk = 0: identity map, tools.)
struct Pieces {
    int n, chunk, xtiles, nseg, zt;
};
__device__ __forceinline__ bool piece_of(const Pieces& pc, int& xt, int& yseg, int& ztile, int& b) {
    int id = blockIdx.x;
    if (pc.chunk > 0) id = (id & 7) * pc.chunk + (id >> 3);
    if (id >= pc.n) return false;
    xt = id % pc.xtiles; id /= pc.xtiles;
    yseg = id % pc.nseg; id /= pc.nseg;
    ztile = id % pc.zt; b = id / pc.zt;
    return true;
}
// (the knob is read at EVERY launch on purpose — tests/test_gpu_ops.py flips it inside one process to show the results do not
// depend on the map; a getenv is ~0.1 us of host time per launch.  The 8 is the XCD count of the one chip this library is built for.)
inline Pieces make_pieces(int xtiles, int nseg, int zt, int B, const char* env, unsigned& grid) {
    const char* e = getenv(env);
    const int on = e ? atoi(e) : 1;
    Pieces pc;
    pc.xtiles = xtiles; pc.nseg = nseg; pc.zt = zt;
    pc.n = xtiles * nseg * zt * B;
    pc.chunk = on ? cdiv(pc.n, 8) : 0;
    grid = on ? (unsigned)pc.chunk * 8u : (unsigned)pc.n;
    return pc;
}

struct FGeom {
    int Lz, Hy, Wx, p, g1, g2, L;
    long P, pred_bstride, V;
    int xo, xtiles, tys;               // outputs per x-tile, number of x-tiles, output rows per y-segment
    float inv_count, inv_pm;           // 1 / (B V), 1 / (P mask.sum())
    Pieces pc;
};

// lane i <- lane i - 1 (x - 1) / lane i + 1 (x + 1); lanes without a source read 0 (halo lanes: their results are never used)
__device__ __forceinline__ float lft(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float rgt(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ f32x4 lft(f32x4 v) { return f32x4{lft(v[0]), lft(v[1]), lft(v[2]), lft(v[3])}; }
__device__ __forceinline__ f32x4 rgt(f32x4 v) { return f32x4{rgt(v[0]), rgt(v[1]), rgt(v[2]), rgt(v[3])}; }
__device__ __forceinline__ f32x4 smooth_x(f32x4 v) { return lft(v) + 2.f * v + rgt(v); }        // [1 2 1] along x

struct State {
    // loaded LD steps ahead (slot = consuming step mod LD; one step of ~200 VALU instructions on two waves per SIMD does not
    // cover an HBM round trip: 77 -> ... us with two)
    f32x4 in[LD][3];                  // prediction at (x, row, z - 1 | z | z + 1)
    float et[LD];                     // target edge map at the row whose gradient field is formed in that step
    f32x4 img[LD];                    // image values / mask flag of the row that leaves in that step
    float mk[LD];
    f32x4 P0[2], P1[2], P2[2];        // forward partials (z and x applied) of the two previous rows
    f32x4 F0[2], F1[2], F2[2];        // gradient field of the two previous rows
    f32x4 ctr[2];                     // the prediction itself at (x, row, z) of the two previous rows (reconstruction term)
};

// (row / p, row % p) of a clamped row index that advances by one per step, without a division per step (uniform: SGPRs)
struct RowIdx {
    int q, r;
    __device__ __forceinline__ void init(int row, int p, int Hy) { const int c = min(max(row, 0), Hy - 1); q = c / p; r = c % p; }
    // the (unclamped) row becomes `row`: the clamped one moves only inside (0, Hy)
    __device__ __forceinline__ void advance(int row, int p, int Hy) {
        if (row > 0 && row < Hy) { if (++r == p) { r = 0; ++q; } }
    }
};

// F32OUT: the fp32 gradient is stored too (false: the bf16 copy only — all the bf16 step consumes; 57 of its 213 MB at batch 4)
template <bool F32OUT>
__global__ __launch_bounds__(NT, VITAE_LOSS_MINW) void loss_fwd_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ imgs,
                                                            const float* __restrict__ mask, const float* __restrict__ Et,
                                                            const float* __restrict__ hp, float* __restrict__ dpred,
                                                            __bf16* __restrict__ dpred16, float* __restrict__ nonfinite,
                                                            double* __restrict__ acc, const FGeom g) {
    __shared__ f32x4 sU[2][NW][64];
    __shared__ f32x4 sW[2][NW][64];
    __shared__ float red[2][NW];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Lz = g.Lz, Hy = g.Hy, Wx = g.Wx, p = g.p;
    int xt, yseg, ztile, b;
    if (!piece_of(g.pc, xt, yseg, ztile, b)) return;        // (the grid is rounded up to a multiple of 8: whole workgroups leave)
    const int x0 = xt * g.xo, ys = yseg * g.tys, z0 = ztile * (NW - 2);
    const int x = x0 - 2 + lane, z = z0 - 1 + w;
    const int rows = min(g.tys, Hy - ys);                 // output rows of this segment
    const float ce = 2.f * hp[VITAE_HP_G_EDGE] * g.inv_count, cr = 2.f * hp[VITAE_HP_G_RECON] * g.inv_pm;

    // x / z parts of every address (constant along the march).  The prediction is read through a buffer resource over this
    // batch element's slab: a lane / plane outside the volume gets an out-of-range offset and the load returns zeros (the zero
    // padding of the convolutions) — no select behind the load.
    const bool okx = x >= 0 && x < Wx;
    const int xc = min(max(x, 0), Wx - 1);
    const int lx = xc / p, ex = (xc % p) * 4;
    const float* pb = pred + (long)b * g.pred_bstride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pb), 0, (int)((long)g.L * g.P * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u, OOB_ROW = 0x40000000u;   // (slab < 1 GB: vitae_loss_fwd_bwd_supported; the two never wrap)
    unsigned zoffb[3];                                      // byte offset of (plane zq, patch column lx, in-patch x) inside the slab
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int zq = z - 1 + j;
        const bool ok = okx && zq >= 0 && zq < Lz;
        const int zc = min(max(zq, 0), Lz - 1);
        zoffb[j] = ok ? 4u * (unsigned)((int)(((long)(zc / p) * g.g1 * g.g2 + lx) * g.P) + (zc % p) * p * p * 4 + ex) : OOB;
    }
    const bool okzc = z >= 0 && z < Lz;
    const int zc1 = min(max(z, 0), Lz - 1);
    const float* etp = Et + ((long)b * g.V) + (long)zc1 * Hy * Wx + xc;        // edge maps: row r at etp[r * Wx]
    const float* imp = imgs + ((long)b * 4 * g.V) + (long)zc1 * Hy * Wx + xc;  // images: channel c, row r at imp[c * V + r * Wx]
    const float* mkp = mask + (long)b * g.L + (zc1 / p) * g.g1 * g.g2 + lx;    // mask: row r at mkp[(r / p) * g2]
    const bool outw = w >= 1 && w <= NW - 2;                                   // (wave-uniform) this plane produces outputs
    const bool owner_xz = lane >= 2 && lane < 2 + g.xo && okx && outw && okzc; // this thread's column produces outputs
    const long dbase = (long)b * g.pred_bstride + ((long)(zc1 / p) * g.g1 * g.g2 + lx) * g.P + (zc1 % p) * p * p * 4 + ex;
    const int rstride = g.g2 * (int)g.P, rin = p * 4;      // element offset of row r inside the slab: (r / p) * rstride + (r % p) * rin

    State s;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        s.P0[q] = s.P1[q] = s.P2[q] = s.F0[q] = s.F1[q] = s.F2[q] = s.ctr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < LD; ++q) { s.img[q] = f32x4{0.f, 0.f, 0.f, 0.f}; s.et[q] = s.mk[q] = 0.f; }
    float sq = 0.f, rc = 0.f, chk = 0.f;

    // rows of the NEXT step's loads: input row yy, field row yy - 1, leaving row yy - 2 (clamped into the volume)
    RowIdx ri, rf, ro;
    int yyn = ys - 2;                                       // input row of the next issue()
    ri.init(yyn, p, Hy); rf.init(yyn - 1, p, Hy); ro.init(yyn - 2, p, Hy);

    // Every load of the march is UNCONDITIONAL (rows outside the volume: an out-of-range buffer offset returns the zero
    // padding; halo planes and steps past the end read clamped, valid addresses they do not use): with a load under a branch
    // inside the loop hipcc's s_waitcnt for the loads of two steps ago came out as vmcnt(1..5) — it waited for the loads just
    // issued as well, and the two-step prefetch was one in name only.
    auto issue = [&](auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
#if VITAE_LOSS_ABLATE != 2
        const unsigned roffb = (yyn >= 0 && yyn < Hy) ? 4u * (unsigned)(ri.q * rstride + ri.r * rin) : OOB_ROW;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            s.in[sl][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, zoffb[j] + roffb, 0, 0));
        s.et[sl] = etp[(rf.q * p + rf.r) * Wx];
        const int orow = (ro.q * p + ro.r) * Wx;
#pragma unroll
        for (int c = 0; c < 4; ++c) s.img[sl][c] = imp[(long)c * g.V + orow];
        s.mk[sl] = mkp[ro.q * g.g2];
#else
#pragma unroll
        for (int j = 0; j < 3; ++j) s.in[sl][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
        ++yyn;
        ri.advance(yyn, p, Hy); rf.advance(yyn - 1, p, Hy); ro.advance(yyn - 2, p, Hy);
    };

    RowIdx rs_out;                                          // the row that leaves in the CURRENT step (for the store address)
    rs_out.init(ys - 4, p, Hy);

    // (Round 5, measured and removed: finishing a row one step late — its neighbours' u / w requested behind the barrier, consumed
    // behind the next row's forward + field work — does not move the kernel: 474 vs 475 us at batch 32.  The march is bound by
    // instruction ISSUE, ~350 VALU + ~145 scalar instructions per row and wave on two lockstepped waves per SIMD, not by the LDS
    // round trip.)
    auto step = [&](int t, auto SLOTC, auto OUTC) {
        constexpr bool OUT = decltype(OUTC)::value;             // false: the first four rows of the march (nothing leaves yet)
        constexpr int sl = decltype(SLOTC)::value, cur = sl & 1, oth = cur ^ 1;
        const int yy = ys - 2 + t;
        // ---- forward partials of input row yy: z, then x
        const f32x4 c0 = s.in[sl][1];
        const f32x4 sz = s.in[sl][0] + 2.f * c0 + s.in[sl][2], dz = s.in[sl][2] - s.in[sl][0];
        const f32x4 im = s.img[sl];
        const float mk = s.mk[sl], etv = s.et[sl];
        issue(SLOTC);                                       // slot sl is free: the loads of step t + LD fly under LD steps
        const f32x4 szl = lft(sz), szr = rgt(sz);
        const f32x4 n0 = szl - szr;                         // d(x) s(z)
        const f32x4 n1 = szl + 2.f * sz + szr;              // s(x) s(z)
        const f32x4 n2 = smooth_x(dz);                      // s(x) d(z)
        // ---- Sobel components at row yy - 1 (rows: slot cur = yy - 2, slot oth = yy - 1, n* = yy)
        const f32x4 g0 = s.P0[cur] + 2.f * s.P0[oth] + n0;  // d(x) s(y) s(z)
        const f32x4 g1 = n1 - s.P1[cur];                    // s(x) d(y) s(z): row y + 1 minus row y - 1
        const f32x4 g2 = s.P2[cur] + 2.f * s.P2[oth] + n2;  // s(x) s(y) d(z)
        const f32x4 m2 = g0 * g0 + g1 * g1 + g2 * g2;
        const float ep = (__builtin_amdgcn_sqrtf(m2[0]) + __builtin_amdgcn_sqrtf(m2[1])) +
                         (__builtin_amdgcn_sqrtf(m2[2]) + __builtin_amdgcn_sqrtf(m2[3]));
        const int yf = yy - 1;
        const bool fin = okx && okzc && yf >= 0 && yf < Hy; // the field is exactly 0 outside the volume
        const float err = ep - etv;
        sq += (owner_xz && fin && yf >= ys && yf < ys + rows) ? err * err : 0.f;
        const float de = ce * err;
        f32x4 f0, f1, f2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float f = fin ? de * __builtin_amdgcn_rsqf(m2[c]) : 0.f;      // |g| = 0 -> inf -> NaN, like the reference's sqrt backward
            f0[c] = f * g0[c]; f1[c] = f * g1[c]; f2[c] = f * g2[c];
        }
        // ---- transposed stencils at row yy - 2: y (registers), x (DPP); z goes through LDS below
        const f32x4 h0 = s.F0[cur] + 2.f * s.F0[oth] + f0;  // s(y) F0
        const f32x4 h1 = s.F1[cur] - f1;                    // row y - 1 minus row y + 1 of F1
        const f32x4 h2 = s.F2[cur] + 2.f * s.F2[oth] + f2;  // s(y) F2
        const f32x4 u = (rgt(h0 + h1) + lft(h1 - h0)) + 2.f * h1;   // e(x) s(y) F0 + s(x) d(y) F1 = (rgt - lft) h0 + (lft + 2 + rgt) h1: two shifts, not four
        const f32x4 wv = smooth_x(h2);                      // s(x) s(y) F2
        const f32x4 pc = s.ctr[cur];                        // prediction at (x, yy - 2, z)
        // rotate: slot cur now holds row yy
        s.P0[cur] = n0; s.P1[cur] = n1; s.P2[cur] = n2;
        s.F0[cur] = f0; s.F1[cur] = f1; s.F2[cur] = f2;
        s.ctr[cur] = c0;
        const int oq = rs_out.q, orr = rs_out.r;
        rs_out.advance(yy - 1, p, Hy);
        if constexpr (!OUT) return;
#if VITAE_LOSS_ABLATE != 1
        sU[cur][w][lane] = u;
        sW[cur][w][lane] = wv;
        __syncthreads();
#endif
        if (outw) {
#if VITAE_LOSS_ABLATE == 1
            const f32x4 um = u, up = wv, wm = u, wp = wv;
#else
            const f32x4 um = sU[cur][w - 1][lane], up = sU[cur][w + 1][lane];
            const f32x4 wm = sW[cur][w - 1][lane], wp = sW[cur][w + 1][lane];
#endif
            f32x4 val = (um + 2.f * u + up) + (wm - wp);    // s(z) [..] + (plane z - 1 minus plane z + 1) of s(x) s(y) F2
            const f32x4 d = pc - im;
            const float mflag = mk != 0.f ? 1.f : 0.f;
            val = (cr * mflag) * d + val;
            if (owner_xz && (VITAE_LOSS_ABLATE != 3 || val[0] == 1234.5f)) {
                rc += mflag * ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
                chk += (val[0] + val[1]) + (val[2] + val[3]);          // non-finite as soon as one stored value is
                const long doff = dbase + (long)(oq * rstride + orr * rin);
                if (F32OUT) *reinterpret_cast<f32x4*>(dpred + doff) = val;
                if (!F32OUT || dpred16) {
                    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<bf4*>(dpred16 + doff) = bf4{(__bf16)val[0], (__bf16)val[1], (__bf16)val[2], (__bf16)val[3]};
                }
            }
        }
    };

    const int nsteps = rows + 4;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2 % LD>;
    using I3 = std::integral_constant<int, 3 % LD>;
    issue(I0{});
    issue(I1{});
    if constexpr (LD == 4) { issue(I2{}); issue(I3{}); }
    step(0, I0{}, std::false_type{});
    step(1, I1{}, std::false_type{});
    step(2, I2{}, std::false_type{});
    step(3, I3{}, std::false_type{});
    int t = 4;
    if constexpr (LD == 4) {
        for (; t + 3 < nsteps; t += 4) {
            step(t, I0{}, std::true_type{});
            step(t + 1, I1{}, std::true_type{});
            step(t + 2, I2{}, std::true_type{});
            step(t + 3, I3{}, std::true_type{});
        }
        if (t < nsteps) step(t++, I0{}, std::true_type{});
        if (t < nsteps) step(t++, I1{}, std::true_type{});
        if (t < nsteps) step(t++, I2{}, std::true_type{});
    } else {
        for (; t + 1 < nsteps; t += 2) {
            step(t, I0{}, std::true_type{});
            step(t + 1, I1{}, std::true_type{});
        }
        if (t < nsteps) step(t, I0{}, std::true_type{});
    }

    const bool bad = !(fabsf(chk) <= 3.4028234e38f);         // NaN or inf
    sq = wave_sum(sq);
    rc = wave_sum(rc);
    __syncthreads();
    if (lane == 0) { red[0][w] = sq; red[1][w] = rc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, r = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) { a += red[0][i]; r += red[1][i]; }
        atomicAdd(acc + VITAE_ACC_EDGE, (double)a);
        atomicAdd(acc + VITAE_ACC_RECON, (double)(r / (float)g.P));
    }
    if (bad && nonfinite) *nonfinite = __builtin_nanf("");   // benign race: every writer stores the same value
}

// ---------------------------------------------------------------------------------------------------------------------
// The target's edge map in ONE pass over the input: E_tgt = sum_c |Sobel(gauss11(img_c))| (model/model_utils/gaussian_filter.py:16-26,
// sobel_filter.py:37-45, model/vit_autoenc.py:221-223) — replaces blur_xy + blur_z + sobel_mag_tiled (csrc/loss.hip: 57 MB read and
// written twice for the blurred intermediates; 89 + 44 + 59 us on the target branch at batch 4).  Same organisation as the
// kernel above: lanes = x (the 11-tap blur along x is a chain of five DPP wave shifts to either side), the thread marches along y
// (the blur along y as eleven running sums — each new row feeds all of them, the oldest leaves complete; the Sobel rows in a
// two-deep ring), waves = z-planes (the blur along z reads its eleven planes straight from global memory, one dword per channel
// and plane: neighbouring waves ask for ten of the same eleven lines, so they come from the L1; the Sobel's z-stencil exchanges the
// blurred value through LDS).  Zero padding twice, as the reference's two convolutions do it: the INPUT is zero outside the volume
// (out-of-range buffer offsets), and the BLURRED volume is zero outside it as well (forced before the Sobel).
constexpr int TAPS = 11, RADB = 5;
constexpr int XO_T = 64 - 2 * (RADB + 1);      // 52 output columns of a 64-lane row

struct TGeom {
    int Lz, Hy, Wx;
    long V;
    int xo, xtiles, tys;
    float k[TAPS];
    Pieces pc;
};

#ifndef VITAE_TARGET_MINW
#define VITAE_TARGET_MINW 2
#endif
__global__ __launch_bounds__(NT, VITAE_TARGET_MINW) void target_edge_kernel(const float* __restrict__ imgs, float* __restrict__ Et, const TGeom g) {
    // Round 4: the 18 planes (8 waves x 11 taps overlap that much) x 4 channels x 64 columns a workgroup needs of an input row are
    // staged ONCE in LDS, channel-interleaved (16 bytes per voxel): a thread issues 12 dword loads per row instead of 44 (the round-3
    // kernel was bound by its texture-path instructions: 99 us at batch 4, 45 us without them) and fetches its eleven taps with
    // eleven ds_read_b128.  Two row buffers: a row is written one step before it is read, and every step ends in a barrier.
    constexpr int NPL = NW + TAPS - 1;                      // 18
    __shared__ f32x4 sB[2][NW][64];
    __shared__ f32x4 sIn[2][NPL][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Lz = g.Lz, Hy = g.Hy, Wx = g.Wx;
    int xt, yseg, ztile, b;
    if (!piece_of(g.pc, xt, yseg, ztile, b)) return;
    const int x0 = xt * g.xo, ys = yseg * g.tys, z0 = ztile * (NW - 2);
    const int x = x0 - (RADB + 1) + lane, z = z0 - 1 + w;
    const int rows = min(g.tys, Hy - ys);
    const bool okx = x >= 0 && x < Wx, okzc = z >= 0 && z < Lz;
    const int xc = min(max(x, 0), Wx - 1);
    const bool outw = w >= 1 && w <= NW - 2;
    const bool owner_xz = lane >= RADB + 1 && lane < RADB + 1 + g.xo && okx && outw && okzc;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(imgs + (long)b * 4 * g.V), 0, (int)(g.V * 16), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // this wave stages planes w, w + 8 and (w < 2) w + 16 of the 18: plane index pl <-> z0 - 6 + pl; out-of-range offsets return the
    // zero padding of the reference's first convolution
    constexpr int NSTG = (NPL + NW - 1) / NW;               // 3
    unsigned offp[NSTG];
#pragma unroll
    for (int k = 0; k < NSTG; ++k) {
        const int pl = w + NW * k, zq = z0 - (RADB + 1) + pl;
        offp[k] = (okx && pl < NPL && zq >= 0 && zq < Lz) ? 4u * (unsigned)((long)zq * Hy * Wx + xc) : OOB;
    }
    float* etp = Et + (long)b * g.V + (long)min(max(z, 0), Lz - 1) * Hy * Wx + xc;
    float kk[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) kk[j] = g.k[j];

    f32x4 stg[2][NSTG];                                    // this thread's share of two rows in flight (four channels per voxel): a row is loaded
                                                           // TWO steps before it is written to LDS (one set: one step, and every step began
                                                           // with s_waitcnt vmcnt(0) on loads ~1000 clocks old)
    f32x4 A[TAPS];                                         // running sums of the blur along y: A[j] belongs to row (current row - 5 + j)
    f32x4 P0[2], P1[2], P2[2];                             // Sobel partials (z and x applied) of the two previous blurred rows
#pragma unroll
    for (int j = 0; j < TAPS; ++j) A[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    P0[0] = P0[1] = P1[0] = P1[1] = P2[0] = P2[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    int yyn = ys - (RADB + 1);                              // input row of the next issue()
    auto issue = [&](auto SETC) {                           // (every load unconditional: see the kernel above)
        constexpr int set = decltype(SETC)::value;
        const int yc = min(max(yyn, 0), Hy - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned soff = 4u * (unsigned)((long)c * g.V + (long)yc * Wx);
#pragma unroll
            for (int k = 0; k < NSTG; ++k) stg[set][k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, offp[k], soff, 0));
        }
        ++yyn;
    };
    auto commit = [&](int buf, auto SETC) {                 // a staged row -> LDS
        constexpr int set = decltype(SETC)::value;
#pragma unroll
        for (int k = 0; k < NSTG; ++k)
            if (w + NW * k < NPL) sIn[buf][w + NW * k][lane] = stg[set][k];
    };

    auto step = [&](int t, auto PARC, auto SOBC, auto OUTC) {
        constexpr int cur = decltype(PARC)::value, oth = cur ^ 1;
        constexpr bool SOB = decltype(SOBC)::value, OUT = decltype(OUTC)::value;
        const int yy = ys - (RADB + 1) + t;                  // input row of this step
        // ---- blur along z (the row is zero outside the volume)
        const float rowok = (yy >= 0 && yy < Hy) ? 1.f : 0.f;
        f32x4 bz = kk[0] * sIn[cur][w][lane];
#pragma unroll
        for (int j = 1; j < TAPS; ++j) bz += kk[j] * sIn[cur][w + j][lane];
        bz *= rowok;
        commit(oth, PARC);                                  // row t + 1 (loaded during step t - 2) -> the other buffer
        issue(PARC);                                        // row t + 3's loads fly under two steps
        // ---- blur along x: five wave shifts to either side
        f32x4 bx = kk[RADB] * bz, l = bz, r = bz;
#pragma unroll
        for (int d = 1; d <= RADB; ++d) {
            l = lft(l); r = rgt(r);                         // l = bz(x - d), r = bz(x + d)
            bx += kk[RADB - d] * l + kk[RADB + d] * r;
        }
        // ---- blur along y: the new row feeds the eleven running sums; A[0] (row yy - 5) is complete
#pragma unroll
        for (int j = 0; j < TAPS - 1; ++j) A[j] = A[j + 1] + kk[TAPS - 1 - j] * bx;
        A[TAPS - 1] = kk[0] * bx;
        if constexpr (!SOB) { __syncthreads(); return; }    // the first ten rows of the march: no blurred row yet (the barrier publishes the staged row)
        const int yb = yy - RADB;                           // the blurred row
        const bool bok = okx && okzc && yb >= 0 && yb < Hy; // the blurred volume is zero outside the volume, too
        f32x4 bl = A[0];
        if (!bok) bl = f32x4{0.f, 0.f, 0.f, 0.f};
        sB[cur][w][lane] = bl;
        __syncthreads();                                    // (also publishes the input row staged above)
        if (outw) {
            const f32x4 bm = sB[cur][w - 1][lane], bp = sB[cur][w + 1][lane];
            const f32x4 sz = bm + 2.f * bl + bp, dz = bp - bm;
            const f32x4 szl = lft(sz), szr = rgt(sz);
            const f32x4 n0 = szl - szr, n1 = szl + 2.f * sz + szr, n2 = smooth_x(dz);
            if constexpr (OUT) {
                const f32x4 g0 = P0[cur] + 2.f * P0[oth] + n0;
                const f32x4 g1 = n1 - P1[cur];
                const f32x4 g2 = P2[cur] + 2.f * P2[oth] + n2;
                const f32x4 m2 = g0 * g0 + g1 * g1 + g2 * g2;
                const float e = (__builtin_amdgcn_sqrtf(m2[0]) + __builtin_amdgcn_sqrtf(m2[1])) +
                                (__builtin_amdgcn_sqrtf(m2[2]) + __builtin_amdgcn_sqrtf(m2[3]));
                const int yo = yb - 1;                      // the row whose edge value is complete
                if (owner_xz && yo < Hy) etp[(long)yo * Wx] = e;
            }
            P0[cur] = n0; P1[cur] = n1; P2[cur] = n2;
        }
    };

    // rows: input ys - 6 + t; blurred ys - 11 + t (t >= 10: ys - 1); edge ys - 12 + t (t >= 12: ys)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using F = std::false_type;
    using T = std::true_type;
    issue(I0{});                                            // row 0 -> buffer 0; rows 1 (set 0) and 2 (set 1) in flight
    commit(0, I0{});
    issue(I0{});
    issue(I1{});
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 10; t += 2) { step(t, I0{}, F{}, F{}); step(t + 1, I1{}, F{}, F{}); }
    step(10, I0{}, T{}, F{});
    step(11, I1{}, T{}, F{});
    const int nsteps = rows + 12;
    int t = 12;
#pragma unroll 1
    for (; t + 1 < nsteps; t += 2) {
        step(t, I0{}, T{}, T{});
        step(t + 1, I1{}, T{}, T{});
    }
    if (t < nsteps) step(t, I0{}, T{}, T{});
}

}  // namespace

// 1 when vitae_loss_fwd_bwd serves this geometry (else the caller keeps vitae_loss_fwd_fused + vitae_loss_bwd_fused)
extern "C" int vitae_loss_fwd_bwd_supported(int C, int Lz, int Hy, int Wx, int p) {
    if (C != 4 || p <= 0 || Lz % p || Hy % p || Wx % p) return 0;
    const long per_batch = ((long)(Lz / p) * (Hy / p) * (Wx / p) + 1) * p * p * p * 4;   // incl. the cls row of the decoder output
    return per_batch < (1L << 28) && cdiv(Lz, NW - 2) <= 65535;   // (slab bytes < 2^30: the out-of-range offsets of the kernel never wrap)
}

extern "C" int vitae_loss_fwd_bwd(const float* pred, long pred_bstride, const float* imgs, const float* mask, const float* edge_tgt,
                                  const float* hp, float* dpred, void* dpred_bf16, float* nonfinite_flag, double* acc,
                                  float mask_sum, int B, int C, int Lz, int Hy, int Wx, int p, void* stream) {
    if (!pred || !imgs || !mask || !edge_tgt || !hp || (!dpred && !dpred_bf16) || !acc || B <= 0 || B > 65535 || p <= 0 || mask_sum <= 0.f ||
        Lz % p || Hy % p || Wx % p)
        return VITAE_ERR_INVALID_ARG;
    if (!vitae_loss_fwd_bwd_supported(C, Lz, Hy, Wx, p) || pred_bstride >= (1L << 31) || (pred_bstride & 3) ||
        ((uintptr_t)pred & 15) || ((uintptr_t)dpred & 15) || ((uintptr_t)dpred_bf16 & 7))
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    FGeom g;
    g.Lz = Lz; g.Hy = Hy; g.Wx = Wx; g.p = p; g.g1 = Hy / p; g.g2 = Wx / p; g.L = (Lz / p) * g.g1 * g.g2;
    g.P = (long)p * p * p * 4; g.pred_bstride = pred_bstride; g.V = (long)Lz * Hy * Wx;
    g.xtiles = cdiv(Wx, XO_MAX); g.xo = cdiv(Wx, g.xtiles);
    const int zt = cdiv(Lz, NW - 2);
    // y-segments: one workgroup per CU is enough (measured: 124 us with 256 workgroups, 145 with 1024 at batch 4 — every segment
    // re-does 4 rows of the march); rows per segment a multiple of 4 and at least 8
    static const int target = getenv("VITAE_LOSS_WGS") ? atoi(getenv("VITAE_LOSS_WGS")) : 256;
    int nseg = cdiv(target, g.xtiles * zt * B);
    if (nseg < 1) nseg = 1;
    int tys = cdiv(cdiv(Hy, nseg), 4) * 4;
    if (tys < 8) tys = 8;
    if (tys > Hy) tys = Hy;
    g.tys = tys;
    nseg = cdiv(Hy, tys);
    g.inv_count = 1.0f / (float)((long)B * g.V);
    g.inv_pm = 1.0f / ((float)g.P * mask_sum);
    unsigned grid;
    g.pc = make_pieces(g.xtiles, nseg, zt, B, "VITAE_LOSS_XCD", grid);
    if (dpred)
        hipLaunchKernelGGL(loss_fwd_bwd_kernel<true>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, pred, imgs, mask, edge_tgt,
                           hp, dpred, reinterpret_cast<__bf16*>(dpred_bf16), nonfinite_flag, acc, g);
    else
        hipLaunchKernelGGL(loss_fwd_bwd_kernel<false>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, pred, imgs, mask, edge_tgt,
                           hp, dpred, reinterpret_cast<__bf16*>(dpred_bf16), nonfinite_flag, acc, g);
    return vitae_launch_status();
}

// edge_tgt[B, Lz, Hy, Wx] = sum_c |Sobel(gauss(imgs[:, c]))| in one launch (4 channels, 11 taps); 1 / 0 = served / not
extern "C" int vitae_target_edge_supported(int C, int ntaps, int Lz, int Hy, int Wx) {
    return C == 4 && ntaps == TAPS && (long)Lz * Hy * Wx * 16 < (1L << 31) && cdiv(Lz, NW - 2) <= 65535;
}

extern "C" int vitae_target_edge(const float* imgs, float* edge_tgt, const float* taps_host, int ntaps, int B, int C, int Lz, int Hy,
                                 int Wx, void* stream) {
    if (!imgs || !edge_tgt || !taps_host || B <= 0 || B > 65535 || Lz <= 0 || Hy <= 0 || Wx <= 0) return VITAE_ERR_INVALID_ARG;
    if (!vitae_target_edge_supported(C, ntaps, Lz, Hy, Wx)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    TGeom g;
    g.Lz = Lz; g.Hy = Hy; g.Wx = Wx; g.V = (long)Lz * Hy * Wx;
    for (int i = 0; i < TAPS; ++i) g.k[i] = taps_host[i];
    g.xtiles = cdiv(Wx, XO_T); g.xo = cdiv(Wx, g.xtiles);
    const int zt = cdiv(Lz, NW - 2);
    static const int target = getenv("VITAE_TARGET_WGS") ? atoi(getenv("VITAE_TARGET_WGS")) : 256;
    int nseg = cdiv(target, g.xtiles * zt * B);
    if (nseg < 1) nseg = 1;
    int tys = cdiv(cdiv(Hy, nseg), 4) * 4;                 // (every segment re-does 12 rows of the march)
    if (tys < 16) tys = 16;
    if (tys > Hy) tys = Hy;
    g.tys = tys;
    nseg = cdiv(Hy, tys);
    unsigned grid;
    g.pc = make_pieces(g.xtiles, nseg, zt, B, "VITAE_TARGET_XCD", grid);
    hipLaunchKernelGGL(target_edge_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, imgs, edge_tgt, g);
    return vitae_launch_status();
}
