// Row / column normalisation kernels: LayerNorm (reference op K5: partial(nn.LayerNorm, eps=1e-6),
// model/vit_autoenc.py:292,300,308 used at model/vit.py:132,135,142-143 and vit_autoenc.py:36,51)
// and the predictor's BatchNorm1d + ReLU (op K23, vit_autoenc.py:263-268).  HBM-bound:
// LayerNorm keeps the row in registers (one wave per row, 16-byte loads when D % 256 == 0),
// statistics by wave shuffles; parameter gradients are wave-partial sums + one atomicAdd per
// column per block into the (pre-zeroed) gradient arena.
#include <cstdlib>
#include <initializer_list>
#include "common.hpp"
#include "vitae_hip.h"

namespace {

constexpr int LN_MAX_PER_LANE = 16;   // D <= 1024

// ------------------------------------------------------------------ LayerNorm forward
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            __bf16* __restrict__ y16,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int M, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (long)row * D;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = c < D ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            const float o = (v[i] - mean) * rstd * w[c] + b[c];
            if (y) y[(long)row * D + c] = o;
            if (y16) y16[(long)row * D + c] = (__bf16)o;
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// D = 256 * NV: every lane owns NV float4 groups (columns 4 * (lane + 64 i) ..): 16-byte loads of x, w, b, 16-byte
// stores of y and 8-byte stores of the bf16 copy instead of 4- and 2-byte ones.
template <int NV>
// (argument order: what the first loads and the bounds need sits inside the first 14 dwords, which arrive preloaded in SGPRs — build.py)
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, int M, float eps, const float* __restrict__ w,
                                                                const float* __restrict__ b, float* __restrict__ y,
                                                                __bf16* __restrict__ y16, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out) {
    constexpr int D = 256 * NV;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long)row * D);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    f32x4 v[NV], wv[NV], bv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[lane + 64 * i]; wv[i] = w4[lane + 64 * i]; bv[i] = b4[lane + 64 * i];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * wv[i][e] + bv[i][e];
        if (y) reinterpret_cast<f32x4*>(y + (long)row * D)[lane + 64 * i] = o;
        if (y16) {
            bf16x4 o16;
#pragma unroll
            for (int e = 0; e < 4; ++e) o16[e] = (__bf16)o[e];
            reinterpret_cast<bf16x4*>(y16 + (long)row * D)[lane + 64 * i] = o16;
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// ------------------------------------------------------------------ LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w;  dw += dy * xhat;  db += dy.
// Each wave walks rows row0, row0 + nwaves, ...; per-lane column partials are combined over the
// block's 4 waves in LDS and added atomically.  `dx_accumulate` adds into dx (residual joins).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db,
                                                            __bf16* __restrict__ dx16, float* __restrict__ dx_colsum,
                                                            int M, int D, int dx_accumulate) {
    __shared__ float red[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    float pw[LN_MAX_PER_LANE], pb[LN_MAX_PER_LANE], wv[LN_MAX_PER_LANE], pc[LN_MAX_PER_LANE];
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        pw[i] = 0.f; pb[i] = 0.f; pc[i] = 0.f;
        const int c = lane + 64 * i;
        wv[i] = c < D ? w[c] : 0.f;
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += nwaves) {
        const float mu = mean[row], rs = rstd[row];
        const float* xr = x + (long)row * D;
        const float* dyr = dy + (long)row * D;
        float xh[LN_MAX_PER_LANE], g[LN_MAX_PER_LANE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            float d = 0.f, xv = 0.f;
            if (c < D) { d = dyr[c]; xv = (xr[c] - mu) * rs; }
            xh[i] = xv; g[i] = d * wv[i];
            pw[i] += d * xv; pb[i] += d;
            s1 += g[i]; s2 += g[i] * xv;
        }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
        float* dxr = dx + (long)row * D;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                float r = rs * (g[i] - s1 - xh[i] * s2);
                if (dx_accumulate) r += dxr[c];
                dxr[c] = r;
                if (dx16) dx16[(long)row * D + c] = (__bf16)r;
                pc[i] += r;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        if (64 * i >= D) break;
        __syncthreads();
        red[0][wave][lane] = pw[i]; red[1][wave][lane] = pb[i]; red[2][wave][lane] = pc[i];
        __syncthreads();
        if (wave == 0) {
            const int c = lane + 64 * i;
            if (c < D) {
                atomicAdd(dw + c, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
                atomicAdd(db + c, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
                if (dx_colsum) atomicAdd(dx_colsum + c, red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane]);
            }
        }
    }
}

// D <= 768 (every model of the reference): each wave owns RPW consecutive rows whose loads are all issued up front;
// column partials accumulate in registers over the wave's rows and meet the other waves' in LDS in a single round
// (7.6 us vs 10.8 us for the generic kernel at [440, 768], 8.8 vs 13.2 at [868, 512]).
template <int RPW>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                                 float* __restrict__ dw, float* __restrict__ db,
                                                                 __bf16* __restrict__ dx16, float* __restrict__ dx_colsum,
                                                                 int M, int D, int dx_accumulate) {
    extern __shared__ float red[];   // [3][4][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * RPW;
    constexpr int NL = 12;           // D <= 768 in this variant
    float dv[RPW][NL], xv[RPW][NL], ov[RPW][NL], mu[RPW], rs[RPW], wv[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { const int c = lane + 64 * i; wv[i] = c < D ? w[c] : 0.f; }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, M - 1);
        mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < D;
            dv[r][i] = ok ? dy[(long)row * D + c] : 0.f;
            xv[r][i] = ok ? x[(long)row * D + c] : 0.f;
            ov[r][i] = (ok && dx_accumulate) ? dx[(long)row * D + c] : 0.f;
        }
    }
    float pw[NL], pb[NL], pc[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { pw[i] = 0.f; pb[i] = 0.f; pc[i] = 0.f; }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int c = lane + 64 * i;
            const float xh = c < D ? (xv[r][i] - mu[r]) * rs[r] : 0.f;
            const float g = dv[r][i] * wv[i];
            pw[i] += dv[r][i] * xh; pb[i] += dv[r][i];
            s1 += g; s2 += g * xh;
            xv[r][i] = xh; dv[r][i] = g;
        }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float o = rs[r] * (dv[r][i] - s1 - xv[r][i] * s2) + ov[r][i];
                dx[(long)row * D + c] = o;
                if (dx16) dx16[(long)row * D + c] = (__bf16)o;
                pc[i] += o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int c = lane + 64 * i;
        if (c < D) { red[(0 * 4 + wave) * D + c] = pw[i]; red[(1 * 4 + wave) * D + c] = pb[i]; red[(2 * 4 + wave) * D + c] = pc[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dw + c, red[0 * D + c] + red[1 * D + c] + red[2 * D + c] + red[3 * D + c]);
        atomicAdd(db + c, red[4 * D + c] + red[5 * D + c] + red[6 * D + c] + red[7 * D + c]);
        if (dx_colsum) atomicAdd(dx_colsum + c, red[8 * D + c] + red[9 * D + c] + red[10 * D + c] + red[11 * D + c]);
    }
}

// The same with 16-byte accesses for D = 256 * NV (lane owns the float4 groups lane + 64 i): 3 loads per array and row
// instead of 12, 16-byte dx stores, 8-byte bf16 stores.
template <int RPW, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                     const float* __restrict__ w, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                                     float* __restrict__ dw, float* __restrict__ db,
                                                                     __bf16* __restrict__ dx16, float* __restrict__ dx_colsum,
                                                                     int M, int dx_accumulate) {
    constexpr int D = 256 * NV;
    extern __shared__ float red[];   // [3][4][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 wv[NV];
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
#pragma unroll
    for (int i = 0; i < NV; ++i) wv[i] = w4[lane + 64 * i];
    f32x4 pw[NV], pb[NV], pc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { pw[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[i] = pw[i]; pc[i] = pw[i]; }
    // a workgroup takes every gridDim.x-th group of 4 * RPW rows: its d(gamma) / d(beta) / column-sum partials meet in LDS and
    // leave as 3 D atomics ONCE per workgroup — with one group per workgroup a batch-32 launch (M = 6944: 868 workgroups)
    // spent two thirds of its time on 1.3 M device-scope atomics (29.7 us against 10 us of HBM time)
    const int ngroups = (M + 4 * RPW - 1) / (4 * RPW);
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int row0 = (grp * 4 + wave) * RPW;
    f32x4 dv[RPW][NV], xv[RPW][NV], ov[RPW][NV];
    float mu[RPW], rs[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, M - 1);
        mu[r] = mean[row]; rs[r] = rstd[row];
        const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy + (long)row * D);
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x + (long)row * D);
        const f32x4* o4 = reinterpret_cast<const f32x4*>(dx + (long)row * D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            dv[r][i] = dy4[lane + 64 * i];
            xv[r][i] = x4[lane + 64 * i];
            if (dx_accumulate) ov[r][i] = o4[lane + 64 * i];
            else ov[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[r][i][e] - mu[r]) * rs[r];
                const float d = dv[r][i][e];
                const float g = d * wv[i][e];
                pw[i][e] += d * xh; pb[i][e] += d;
                s1 += g; s2 += g * xh;
                xv[r][i][e] = xh; dv[r][i][e] = g;
            }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = rs[r] * (dv[r][i][e] - s1 - xv[r][i][e] * s2) + ov[r][i][e];
                pc[i][e] += o[e];
            }
            reinterpret_cast<f32x4*>(dx + (long)row * D)[lane + 64 * i] = o;
            if (dx16) {
                bf16x4 o16;
#pragma unroll
                for (int e = 0; e < 4; ++e) o16[e] = (__bf16)o[e];
                reinterpret_cast<bf16x4*>(dx16 + (long)row * D)[lane + 64 * i] = o16;
            }
        }
    }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        *reinterpret_cast<f32x4*>(&red[(0 * 4 + wave) * D + c]) = pw[i];
        *reinterpret_cast<f32x4*>(&red[(1 * 4 + wave) * D + c]) = pb[i];
        *reinterpret_cast<f32x4*>(&red[(2 * 4 + wave) * D + c]) = pc[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dw + c, red[0 * D + c] + red[1 * D + c] + red[2 * D + c] + red[3 * D + c]);
        atomicAdd(db + c, red[4 * D + c] + red[5 * D + c] + red[6 * D + c] + red[7 * D + c]);
        if (dx_colsum) atomicAdd(dx_colsum + c, red[8 * D + c] + red[9 * D + c] + red[10 * D + c] + red[11 * D + c]);
    }
}

// Round 4: the same backward WITHOUT atomics.  d(gamma) / d(beta) / column-sum partials of a workgroup leave as ONE coalesced
// 3 D-float record in a workspace (`part`: [gridDim.x][3][D]) and a later launch (ln_grad_reduce_kernel, once per backward phase,
// all LayerNorms of the phase together) sums the records into the gradient arena — nothing on this path waits for those sums before
// the optimiser does.  With no atomics to ration, a launch uses up to two workgroups per CU, and the rows of the NEXT pair are in
// flight while the current pair is reduced (the round-3 kernel exposed one memory round trip per row group of a 128-workgroup grid:
// 20 us at [3464, 768] against 7.5 us of HBM time, 3 D x 128 device-scope atomics on top).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_part_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                                 int M, int dx_accumulate,          // (14 dwords up to here: preloaded)
                                                                 float* __restrict__ part, __bf16* __restrict__ dx16) {
    constexpr int D = 256 * NV, RPW = 2;
    extern __shared__ float red[];   // [3][4][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 wv[NV];
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
#pragma unroll
    for (int i = 0; i < NV; ++i) wv[i] = w4[lane + 64 * i];
    f32x4 pw[NV], pb[NV], pc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { pw[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[i] = pw[i]; pc[i] = pw[i]; }
    const int ngroups = (M + 4 * RPW - 1) / (4 * RPW);
    f32x4 dv[2][RPW][NV], xv[2][RPW][NV], ov[2][RPW][NV];
    float mu[2][RPW], rs[2][RPW];
    auto load = [&](int grp, f32x4 (&d)[RPW][NV], f32x4 (&xx)[RPW][NV], f32x4 (&oo)[RPW][NV], float (&m_)[RPW], float (&r_)[RPW]) {
        const int row0 = (grp * 4 + wave) * RPW;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = min(row0 + r, M - 1);
            m_[r] = mean[row]; r_[r] = rstd[row];
            const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy + (long)row * D);
            const f32x4* x4 = reinterpret_cast<const f32x4*>(x + (long)row * D);
            const f32x4* o4 = reinterpret_cast<const f32x4*>(dx + (long)row * D);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                d[r][i] = dy4[lane + 64 * i];
                xx[r][i] = x4[lane + 64 * i];
                if (dx_accumulate) oo[r][i] = o4[lane + 64 * i];
                else oo[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto work = [&](int grp, f32x4 (&d)[RPW][NV], f32x4 (&xx)[RPW][NV], f32x4 (&oo)[RPW][NV], float (&m_)[RPW], float (&r_)[RPW]) {
        const int row0 = (grp * 4 + wave) * RPW;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = row0 + r;
            if (row >= M) break;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xx[r][i][e] - m_[r]) * r_[r];
                    const float dd = d[r][i][e];
                    const float g = dd * wv[i][e];
                    pw[i][e] += dd * xh; pb[i][e] += dd;
                    s1 += g; s2 += g * xh;
                    xx[r][i][e] = xh; d[r][i][e] = g;
                }
            s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = r_[r] * (d[r][i][e] - s1 - xx[r][i][e] * s2) + oo[r][i][e];
                    pc[i][e] += o[e];
                }
                reinterpret_cast<f32x4*>(dx + (long)row * D)[lane + 64 * i] = o;
                if (dx16) {
                    bf16x4 o16;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o16[e] = (__bf16)o[e];
                    reinterpret_cast<bf16x4*>(dx16 + (long)row * D)[lane + 64 * i] = o16;
                }
            }
        }
    };
    // two register sets, the loop unrolled by two: while set A is worked on, set B's loads are in flight
    int grp = blockIdx.x;
    if (grp < ngroups) load(grp, dv[0], xv[0], ov[0], mu[0], rs[0]);
    for (; grp < ngroups; grp += 2 * gridDim.x) {
        const int g1 = grp + gridDim.x, g2 = grp + 2 * gridDim.x;
        if (g1 < ngroups) load(g1, dv[1], xv[1], ov[1], mu[1], rs[1]);
        work(grp, dv[0], xv[0], ov[0], mu[0], rs[0]);
        if (g2 < ngroups) load(g2, dv[0], xv[0], ov[0], mu[0], rs[0]);
        if (g1 < ngroups) work(g1, dv[1], xv[1], ov[1], mu[1], rs[1]);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        *reinterpret_cast<f32x4*>(&red[(0 * 4 + wave) * D + c]) = pw[i];
        *reinterpret_cast<f32x4*>(&red[(1 * 4 + wave) * D + c]) = pb[i];
        *reinterpret_cast<f32x4*>(&red[(2 * 4 + wave) * D + c]) = pc[i];
    }
    __syncthreads();
    float* rec = part + (long)blockIdx.x * 3 * D;
    for (int c4 = threadIdx.x; c4 < 3 * D / 4; c4 += 256) {
        const int k = c4 / (D / 4), c = 4 * (c4 % (D / 4));
        const f32x4 a = *reinterpret_cast<const f32x4*>(&red[(k * 4 + 0) * D + c]), b = *reinterpret_cast<const f32x4*>(&red[(k * 4 + 1) * D + c]);
        const f32x4 cc = *reinterpret_cast<const f32x4*>(&red[(k * 4 + 2) * D + c]), d = *reinterpret_cast<const f32x4*>(&red[(k * 4 + 3) * D + c]);
        *reinterpret_cast<f32x4*>(rec + k * D + c) = (a + b) + (cc + d);
    }
}

// out_k[c] += sum over the G records of instance i of part_i[g][k][c], k = 0 (d gamma), 1 (d beta), 2 (column sum of dx; optional),
// for up to LN_RED_MAX LayerNorm instances in one launch: grid (instances, blocks of 16 float4 column groups of the 3 D).
// Summation order is fixed: bitwise reproducible.
constexpr int LN_RED_MAX = 48;
struct LnRedDesc { const float* part; float* dw; float* db; float* cs; int G, D; };
struct LnRedArgs { LnRedDesc d[LN_RED_MAX]; };

__global__ __launch_bounds__(256) void ln_grad_reduce_kernel(const LnRedArgs a) {
    // a workgroup = 16 float4 column groups (256-byte row pieces) x 16 record lanes: lane j of a column group sums records j, j + 16, ...
    // (all of its loads in flight together for G <= 128), the 16 partial sums meet in LDS in a fixed order
    __shared__ f32x4 red[16][16];
    const LnRedDesc& t = a.d[blockIdx.x];
    const int D = t.D, n4 = 3 * D / 4;
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c4 = blockIdx.y * 16 + cg;
    const bool ok = c4 < n4;
    const int k = ok ? c4 / (D / 4) : 0, c = ok ? 4 * (c4 % (D / 4)) : 0;
    float* out = k == 0 ? t.dw : k == 1 ? t.db : t.cs;
    const float* src = t.part + k * D + c;
    const long stride = 3L * D;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (ok && out) {
        int g = rl;
        for (; g + 7 * 16 < t.G; g += 8 * 16) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)(g + 16 * u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; g < t.G; g += 16) acc += *reinterpret_cast<const f32x4*>(src + (long)g * stride);
    }
    red[rl][cg] = acc;
    __syncthreads();
    if (rl == 0 && ok && out) {
        f32x4 s = red[0][cg];
#pragma unroll
        for (int j = 1; j < 16; ++j) s += red[j][cg];
        f32x4* o4 = reinterpret_cast<f32x4*>(out + c);
        *o4 = *o4 + s;
    }
}

// ------------------------------------------------------------------ column sum (bias gradients)
// out[n] += sum_m dy[m, n].  Threads own columns (coalesced rows), blocks own row slabs.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, long ld, float* __restrict__ out,
                                                     int M, int N, int rows_per_block) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    for (int m = r0; m < r1; ++m) s += dy[(long)m * ld + n];
    atomicAdd(out + n, s);
}

// ------------------------------------------------------------------ BatchNorm1d (+ReLU), training mode
// R rows (R = B * Ne tokens of one view: 220 at config 2, 1760 at batch 32) x D features.  Round 4: a workgroup owns 64 feature
// columns (16 lanes x float4: 256-byte row pieces) and its 16 row groups walk the rows with four loads in flight; the statistics
// meet in LDS.  Two passes (mean, then centred variance) like ATen's CPU batch_norm; saves mean / rstd, updates the running stats
// with momentum and the unbiased variance (nn.BatchNorm1d semantics).  Optional bf16 copy of the output (the next Linear's GEMM
// operand).  The round-3 kernels read 64-byte row pieces with one load in flight on 48 workgroups: 124 / 235 us at [1760, 768].
constexpr int BN_CG = 16, BN_RG = 16, BN_COLS = 4 * BN_CG;     // 16 column groups of 4 x 16 row groups = 256 threads, 64 columns

__device__ __forceinline__ f32x4 bn_col_reduce4(f32x4 v, f32x4 (*red)[BN_CG], int cg, int rg) {
    __syncthreads();
    red[rg][cg] = v;
    __syncthreads();
    f32x4 s = red[0][cg];
#pragma unroll
    for (int g = 1; g < BN_RG; ++g) s += red[g][cg];
    return s;
}

__global__ __launch_bounds__(256) void bn1d_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y, __bf16* __restrict__ y16,
                                                            float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var,
                                                            long long* __restrict__ num_batches_tracked,
                                                            int R, int D, float eps, float momentum) {
    __shared__ f32x4 red[BN_RG][BN_CG];
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int cg = threadIdx.x % BN_CG, rg = threadIdx.x / BN_CG;
    const int c = blockIdx.x * BN_COLS + 4 * cg;
    const bool ok = c < D;                                  // D % 4 == 0 (launcher)
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 s = z;
    if (ok) {
        int r = rg;
        for (; r + 3 * BN_RG < R; r += 4 * BN_RG) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (long)(r + u * BN_RG) * D + c);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (; r < R; r += BN_RG) s += *reinterpret_cast<const f32x4*>(x + (long)r * D + c);
    }
    const f32x4 mean = bn_col_reduce4(s, red, cg, rg) / (float)R;
    f32x4 q = z;
    if (ok) {
        int r = rg;
        for (; r + 3 * BN_RG < R; r += 4 * BN_RG) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (long)(r + u * BN_RG) * D + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) { const f32x4 d = v[u] - mean; q += d * d; }
        }
        for (; r < R; r += BN_RG) { const f32x4 d = *reinterpret_cast<const f32x4*>(x + (long)r * D + c) - mean; q += d * d; }
    }
    q = bn_col_reduce4(q, red, cg, rg);
    if (!ok) return;
    f32x4 rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) rstd[e] = rsqrtf(q[e] / R + eps);
    const f32x4 g = *reinterpret_cast<const f32x4*>(w + c), be = *reinterpret_cast<const f32x4*>(b + c);
    const f32x4 sc = rstd * g;
    for (int r = rg; r < R; r += BN_RG) {
        f32x4 v = (*reinterpret_cast<const f32x4*>(x + (long)r * D + c) - mean) * sc + be;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        *reinterpret_cast<f32x4*>(y + (long)r * D + c) = v;
        if (y16) {
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4*>(y16 + (long)r * D + c) = h;
        }
    }
    if (rg == 0) {
        *reinterpret_cast<f32x4*>(save_mean + c) = mean;
        *reinterpret_cast<f32x4*>(save_rstd + c) = rstd;
        if (run_mean) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                run_mean[c + e] = (1.f - momentum) * run_mean[c + e] + momentum * mean[e];
                run_var[c + e] = (1.f - momentum) * run_var[c + e] + momentum * (q[e] / (R > 1 ? R - 1 : 1));
            }
        }
    }
}

// eval mode (nn.BatchNorm1d with track_running_stats, model.eval()): normalise with the RUNNING statistics
__global__ __launch_bounds__(256) void bn1d_relu_eval_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, const float* __restrict__ run_mean,
                                                             const float* __restrict__ run_var, float* __restrict__ y,
                                                             long n, int D, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % D);
    const float v = (x[i] - run_mean[c]) * rsqrtf(run_var[c] + eps) * w[c] + b[c];
    y[i] = v > 0.f ? v : 0.f;
}

// dy arrives for the ReLU output y; ReLU mask = (y > 0).  dx = w*rstd*(g - mean(g) - xhat*mean(g*xhat)).  Optional bf16 copy of dx.
__global__ __launch_bounds__(256) void bn1d_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_rstd, float* __restrict__ dx, __bf16* __restrict__ dx16,
                                                            float* __restrict__ dw, float* __restrict__ db, int R, int D) {
    __shared__ f32x4 red[BN_RG][BN_CG];
    const int cg = threadIdx.x % BN_CG, rg = threadIdx.x / BN_CG;
    const int c = blockIdx.x * BN_COLS + 4 * cg;
    const bool ok = c < D;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 mean = ok ? *reinterpret_cast<const f32x4*>(save_mean + c) : z, rstd = ok ? *reinterpret_cast<const f32x4*>(save_rstd + c) : z;
    const f32x4 g = ok ? *reinterpret_cast<const f32x4*>(w + c) : z;
    f32x4 s1 = z, s2 = z;
    if (ok) {
        int r = rg;
        for (; r + BN_RG < R; r += 2 * BN_RG) {
            f32x4 yv[2], dv[2], xv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long i = (long)(r + u * BN_RG) * D + c;
                yv[u] = *reinterpret_cast<const f32x4*>(y + i); dv[u] = *reinterpret_cast<const f32x4*>(dy + i); xv[u] = *reinterpret_cast<const f32x4*>(x + i);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = yv[u][e] > 0.f ? dv[u][e] : 0.f;
                    s1[e] += d; s2[e] += d * (xv[u][e] - mean[e]) * rstd[e];
                }
        }
        for (; r < R; r += BN_RG) {
            const long i = (long)r * D + c;
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + i), dv = *reinterpret_cast<const f32x4*>(dy + i), xv = *reinterpret_cast<const f32x4*>(x + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = yv[e] > 0.f ? dv[e] : 0.f;
                s1[e] += d; s2[e] += d * (xv[e] - mean[e]) * rstd[e];
            }
        }
    }
    s1 = bn_col_reduce4(s1, red, cg, rg);
    s2 = bn_col_reduce4(s2, red, cg, rg);
    if (!ok) return;
    if (rg == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { atomicAdd(dw + c + e, s2[e]); atomicAdd(db + c + e, s1[e]); }
    }
    const f32x4 m1 = s1 / (float)R, m2 = s2 / (float)R;
    for (int r = rg; r < R; r += BN_RG) {
        const long i = (long)r * D + c;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + i), dv = *reinterpret_cast<const f32x4*>(dy + i), xv = *reinterpret_cast<const f32x4*>(x + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = yv[e] > 0.f ? dv[e] : 0.f;
            o[e] = g[e] * rstd[e] * (d - m1[e] - (xv[e] - mean[e]) * rstd[e] * m2[e]);
        }
        *reinterpret_cast<f32x4*>(dx + i) = o;
        if (dx16) {
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (__bf16)o[e];
            *reinterpret_cast<bf16x4*>(dx16 + i) = h;
        }
    }
}


// ---- the same two ops with the ROWS split over workgroups (round 5) -----------------------------------------------------------
// The kernels above give a 64-column strip ALL R rows: D / 64 = 12 workgroups for the predictor (D = 768) whatever R is — 12 of
// 256 CUs stream 1760 rows at batch 32 / patch 8 (45 us forward, 78 us backward per view, 0.36 TB/s).  Here a launch is
// (D / 64) x RS workgroups, and an op is two launches with RS partial records per column in between:
//   forward:  (1) per split, the local mean and the local sum of squared deviations (two passes over rows that stay in L2);
//             (2) every workgroup merges the RS records in split order with Chan's update (mean = sum n_i mean_i / R,
//                 M2 = sum M2_i + n_i (mean_i - mean)^2 — no E[x^2] - E[x]^2 cancellation), then normalises its own rows;
//   backward: (1) per split, sum(g) and sum(g xhat) (g = dy where y > 0); (2) the sums of all splits in order, then dx of its rows.
// Deterministic (no atomics between the launches); split 0 writes the statistics / adds the parameter gradients.
struct BnSplit { int RS, rps; };     // row splits, rows per split

__device__ __forceinline__ void bn_split_rows(const BnSplit sp, int R, int& r0, int& r1) {
    r0 = blockIdx.y * sp.rps; r1 = min(R, r0 + sp.rps);
}

__global__ __launch_bounds__(256) void bn1d_stats_part_kernel(const float* __restrict__ x, float* __restrict__ part, int R, int D, const BnSplit sp) {
    __shared__ f32x4 red[BN_RG][BN_CG];
    const int cg = threadIdx.x % BN_CG, rg = threadIdx.x / BN_CG;
    const int c = blockIdx.x * BN_COLS + 4 * cg;
    const bool ok = c < D;
    int r0, r1;
    bn_split_rows(sp, R, r0, r1);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 s = z;
    if (ok)
        for (int r = r0 + rg; r < r1; r += BN_RG) s += *reinterpret_cast<const f32x4*>(x + (long)r * D + c);
    const f32x4 mean = bn_col_reduce4(s, red, cg, rg) / (float)max(r1 - r0, 1);
    f32x4 q = z;
    if (ok)
        for (int r = r0 + rg; r < r1; r += BN_RG) { const f32x4 d = *reinterpret_cast<const f32x4*>(x + (long)r * D + c) - mean; q += d * d; }
    q = bn_col_reduce4(q, red, cg, rg);
    if (ok && rg == 0) {
        *reinterpret_cast<f32x4*>(part + ((long)blockIdx.y * 2 + 0) * D + c) = mean;
        *reinterpret_cast<f32x4*>(part + ((long)blockIdx.y * 2 + 1) * D + c) = q;
    }
}

__global__ __launch_bounds__(256) void bn1d_relu_apply_part_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                                   const float* __restrict__ part, float* __restrict__ y, __bf16* __restrict__ y16,
                                                                   float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                                   float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                   long long* __restrict__ num_batches_tracked, int R, int D, float eps,
                                                                   float momentum, const BnSplit sp) {
    if (num_batches_tracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int cg = threadIdx.x % BN_CG, rg = threadIdx.x / BN_CG;
    const int c = blockIdx.x * BN_COLS + 4 * cg;
    if (c >= D) return;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 mean = z;
    for (int i = 0; i < sp.RS; ++i) {
        const float n = (float)(min(R, (i + 1) * sp.rps) - i * sp.rps);
        mean += n * *reinterpret_cast<const f32x4*>(part + ((long)i * 2 + 0) * D + c);
    }
    mean = mean / (float)R;
    f32x4 q = z;
    for (int i = 0; i < sp.RS; ++i) {
        const float n = (float)(min(R, (i + 1) * sp.rps) - i * sp.rps);
        const f32x4 d = *reinterpret_cast<const f32x4*>(part + ((long)i * 2 + 0) * D + c) - mean;
        q += *reinterpret_cast<const f32x4*>(part + ((long)i * 2 + 1) * D + c) + n * (d * d);
    }
    f32x4 rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) rstd[e] = rsqrtf(q[e] / R + eps);
    const f32x4 g = *reinterpret_cast<const f32x4*>(w + c), be = *reinterpret_cast<const f32x4*>(b + c);
    const f32x4 sc = rstd * g;
    int r0, r1;
    bn_split_rows(sp, R, r0, r1);
    for (int r = r0 + rg; r < r1; r += BN_RG) {
        f32x4 v = (*reinterpret_cast<const f32x4*>(x + (long)r * D + c) - mean) * sc + be;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        *reinterpret_cast<f32x4*>(y + (long)r * D + c) = v;
        if (y16) {
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4*>(y16 + (long)r * D + c) = h;
        }
    }
    if (blockIdx.y == 0 && rg == 0) {
        *reinterpret_cast<f32x4*>(save_mean + c) = mean;
        *reinterpret_cast<f32x4*>(save_rstd + c) = rstd;
        if (run_mean) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                run_mean[c + e] = (1.f - momentum) * run_mean[c + e] + momentum * mean[e];
                run_var[c + e] = (1.f - momentum) * run_var[c + e] + momentum * (q[e] / (R > 1 ? R - 1 : 1));
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn1d_relu_bwd_sums_part_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                                      const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                                      float* __restrict__ part, int R, int D, const BnSplit sp) {
    __shared__ f32x4 red[BN_RG][BN_CG];
    const int cg = threadIdx.x % BN_CG, rg = threadIdx.x / BN_CG;
    const int c = blockIdx.x * BN_COLS + 4 * cg;
    const bool ok = c < D;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 mean = ok ? *reinterpret_cast<const f32x4*>(save_mean + c) : z, rstd = ok ? *reinterpret_cast<const f32x4*>(save_rstd + c) : z;
    int r0, r1;
    bn_split_rows(sp, R, r0, r1);
    f32x4 s1 = z, s2 = z;
    if (ok)
        for (int r = r0 + rg; r < r1; r += BN_RG) {
            const long i = (long)r * D + c;
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + i), dv = *reinterpret_cast<const f32x4*>(dy + i), xv = *reinterpret_cast<const f32x4*>(x + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = yv[e] > 0.f ? dv[e] : 0.f;
                s1[e] += d; s2[e] += d * (xv[e] - mean[e]) * rstd[e];
            }
        }
    s1 = bn_col_reduce4(s1, red, cg, rg);
    s2 = bn_col_reduce4(s2, red, cg, rg);
    if (ok && rg == 0) {
        *reinterpret_cast<f32x4*>(part + ((long)blockIdx.y * 2 + 0) * D + c) = s1;
        *reinterpret_cast<f32x4*>(part + ((long)blockIdx.y * 2 + 1) * D + c) = s2;
    }
}

__global__ __launch_bounds__(256) void bn1d_relu_bwd_apply_part_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                                       const float* __restrict__ w, const float* __restrict__ save_mean,
                                                                       const float* __restrict__ save_rstd, const float* __restrict__ part,
                                                                       float* __restrict__ dx, __bf16* __restrict__ dx16, float* __restrict__ dw,
                                                                       float* __restrict__ db, int R, int D, const BnSplit sp) {
    const int cg = threadIdx.x % BN_CG, rg = threadIdx.x / BN_CG;
    const int c = blockIdx.x * BN_COLS + 4 * cg;
    if (c >= D) return;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 s1 = z, s2 = z;
    for (int i = 0; i < sp.RS; ++i) {
        s1 += *reinterpret_cast<const f32x4*>(part + ((long)i * 2 + 0) * D + c);
        s2 += *reinterpret_cast<const f32x4*>(part + ((long)i * 2 + 1) * D + c);
    }
    if (blockIdx.y == 0 && rg == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { atomicAdd(dw + c + e, s2[e]); atomicAdd(db + c + e, s1[e]); }
    }
    const f32x4 mean = *reinterpret_cast<const f32x4*>(save_mean + c), rstd = *reinterpret_cast<const f32x4*>(save_rstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
    const f32x4 m1 = s1 / (float)R, m2 = s2 / (float)R;
    int r0, r1;
    bn_split_rows(sp, R, r0, r1);
    for (int r = r0 + rg; r < r1; r += BN_RG) {
        const long i = (long)r * D + c;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + i), dv = *reinterpret_cast<const f32x4*>(dy + i), xv = *reinterpret_cast<const f32x4*>(x + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = yv[e] > 0.f ? dv[e] : 0.f;
            o[e] = g[e] * rstd[e] * (d - m1[e] - (xv[e] - mean[e]) * rstd[e] * m2[e]);
        }
        *reinterpret_cast<f32x4*>(dx + i) = o;
        if (dx16) {
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (__bf16)o[e];
            *reinterpret_cast<bf16x4*>(dx16 + i) = h;
        }
    }
}

// row splits of a launch: ~384 workgroups, at least 32 rows each
inline BnSplit bn_split_plan(int R, int D) {
    const int strips = cdiv(D, BN_COLS);
    int rs = cdiv(384, strips);
    if (rs > cdiv(R, 32)) rs = cdiv(R, 32);
    if (rs < 1) rs = 1;
    BnSplit sp;
    sp.rps = cdiv(R, rs);
    sp.RS = cdiv(R, sp.rps);
    return sp;
}

}  // namespace

extern "C" int vitae_layernorm_fwd(const float* x, const float* w, const float* b, float* y, void* y_bf16, float* mean,
                                   float* rstd, int M, int D, float eps, void* stream) {
    if (!x || !w || !b || (!y && !y_bf16) || !mean || !rstd || M <= 0 || D <= 0) return VITAE_ERR_INVALID_ARG;
    if (D > 64 * LN_MAX_PER_LANE) return VITAE_ERR_UNSUPPORTED_SHAPE;
    __bf16* y16 = reinterpret_cast<__bf16*>(y_bf16);
    static const int vec_on = getenv("VITAE_LN_VEC") ? atoi(getenv("VITAE_LN_VEC")) : 1;     // 0: 4-byte accesses (A/B)
    const bool aligned = vec_on && !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)b | (uintptr_t)y) & 15) && !((uintptr_t)y16 & 7);
    hipStream_t st = (hipStream_t)stream;
    if (aligned && D == 768) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<3>, dim3(cdiv(M, 4)), dim3(256), 0, st, x, M, eps, w, b, y, y16, mean, rstd);
    else if (aligned && D == 512) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<2>, dim3(cdiv(M, 4)), dim3(256), 0, st, x, M, eps, w, b, y, y16, mean, rstd);
    else if (aligned && D == 1024) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<4>, dim3(cdiv(M, 4)), dim3(256), 0, st, x, M, eps, w, b, y, y16, mean, rstd);
    else if (aligned && D == 256) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<1>, dim3(cdiv(M, 4)), dim3(256), 0, st, x, M, eps, w, b, y, y16, mean, rstd);
    else hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x, w, b, y, y16, mean, rstd, M, D, eps);
    return vitae_launch_status();
}

extern "C" int vitae_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean,
                                   const float* rstd, float* dx, float* dw, float* db, void* dx_bf16,
                                   float* dx_colsum_accum, int M, int D, int dx_accumulate, void* stream) {
    if (!dy || !x || !w || !mean || !rstd || !dx || !dw || !db || M <= 0 || D <= 0) return VITAE_ERR_INVALID_ARG;
    if (D > 64 * LN_MAX_PER_LANE) return VITAE_ERR_UNSUPPORTED_SHAPE;
    __bf16* dx16v = reinterpret_cast<__bf16*>(dx_bf16);
    static const int vec_on = getenv("VITAE_LN_VEC") ? atoi(getenv("VITAE_LN_VEC")) : 1;
    const bool aligned = vec_on && !(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)w | (uintptr_t)dx) & 15) && !((uintptr_t)dx16v & 7);
    if (aligned && (D == 768 || D == 512 || D == 256)) {
        const size_t lds = (size_t)12 * D * sizeof(float);
        hipStream_t st = (hipStream_t)stream;
        static const int cap = getenv("VITAE_LN_BWD_BLOCKS") ? atoi(getenv("VITAE_LN_BWD_BLOCKS")) : 128;   // (sweep: 64 / 128 / 192 / 256 / 512 / all -> 18.3 / 13.0 / 13.3 / 13.1 / 18.1 / 18.6 us at M = 3520, D = 768; 8.2 / 7.5 / 8.8 / 9.8 / 9.5 / 9.5 at M = 1736, D = 512)
        const int nb = min(cdiv(M, 8), cap < 1 ? 1 : cap);
        if (D == 768) hipLaunchKernelGGL((layernorm_bwd_rows_vec_kernel<2, 3>), dim3(nb), dim3(256), lds, st, dy, x, w, mean, rstd, dx, dw, db, dx16v, dx_colsum_accum, M, dx_accumulate);
        else if (D == 512) hipLaunchKernelGGL((layernorm_bwd_rows_vec_kernel<2, 2>), dim3(nb), dim3(256), lds, st, dy, x, w, mean, rstd, dx, dw, db, dx16v, dx_colsum_accum, M, dx_accumulate);
        else hipLaunchKernelGGL((layernorm_bwd_rows_vec_kernel<2, 1>), dim3(nb), dim3(256), lds, st, dy, x, w, mean, rstd, dx, dw, db, dx16v, dx_colsum_accum, M, dx_accumulate);
        return vitae_launch_status();
    }
    if (D <= 768) {
        hipLaunchKernelGGL(layernorm_bwd_rows_kernel<2>, dim3(cdiv(M, 8)), dim3(256), (size_t)12 * D * sizeof(float),
                           (hipStream_t)stream, dy, x, w, mean, rstd, dx, dw, db, reinterpret_cast<__bf16*>(dx_bf16),
                           dx_colsum_accum, M, D, dx_accumulate);
        return vitae_launch_status();
    }
    int blocks = cdiv(M, 4);
    if (blocks > 256) blocks = 256;   // one row per wave up to 1024 rows (row work dominates; capping at 48 blocks doubled the time)
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, w, mean, rstd,
                       dx, dw, db, reinterpret_cast<__bf16*>(dx_bf16), dx_colsum_accum, M, D, dx_accumulate);
    return vitae_launch_status();
}

// workgroups (= partial records) of vitae_layernorm_bwd_part for M rows
extern "C" int vitae_layernorm_bwd_part_records(int M) {
    static const int cap = getenv("VITAE_LN_PART_BLOCKS") ? atoi(getenv("VITAE_LN_PART_BLOCKS")) : 256;   // (128 / 256 / 512 workgroups at [3464, 768]: 12.5 / 10.0 / 9.5 us, but the reduce reads 1.2 / 2.3 / 4.1 MB per LayerNorm: 0.6 / 1.4 / 2.4 us)
    const int ng = cdiv(M, 8);
    return ng < cap ? ng : cap;
}

extern "C" int vitae_layernorm_bwd_part(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                                        float* dx, float* part, void* dx_bf16, int M, int D, int dx_accumulate, void* stream) {
    if (!dy || !x || !w || !mean || !rstd || !dx || !part || M <= 0) return VITAE_ERR_INVALID_ARG;
    if (D != 768 && D != 512 && D != 256 && D != 1024) return VITAE_ERR_UNSUPPORTED_SHAPE;
    __bf16* dx16v = reinterpret_cast<__bf16*>(dx_bf16);
    if ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)w | (uintptr_t)dx | (uintptr_t)part) & 15) || ((uintptr_t)dx16v & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const int nb = vitae_layernorm_bwd_part_records(M);
    const size_t lds = (size_t)12 * D * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (D == 768) hipLaunchKernelGGL(layernorm_bwd_part_kernel<3>, dim3(nb), dim3(256), lds, st, dy, x, w, mean, rstd, dx, M, dx_accumulate, part, dx16v);
    else if (D == 512) hipLaunchKernelGGL(layernorm_bwd_part_kernel<2>, dim3(nb), dim3(256), lds, st, dy, x, w, mean, rstd, dx, M, dx_accumulate, part, dx16v);
    else if (D == 256) hipLaunchKernelGGL(layernorm_bwd_part_kernel<1>, dim3(nb), dim3(256), lds, st, dy, x, w, mean, rstd, dx, M, dx_accumulate, part, dx16v);
    else hipLaunchKernelGGL(layernorm_bwd_part_kernel<4>, dim3(nb), dim3(256), lds, st, dy, x, w, mean, rstd, dx, M, dx_accumulate, part, dx16v);
    return vitae_launch_status();
}

// n instances; the arrays live on the HOST (copied into the launch's arguments)
extern "C" int vitae_ln_grad_reduce(int n, const float* const* part, float* const* dw, float* const* db, float* const* dx_colsum,
                                    const int* records, const int* D, void* stream) {
    if (n <= 0 || !part || !dw || !db || !records || !D) return VITAE_ERR_INVALID_ARG;
    for (int i0 = 0; i0 < n; i0 += LN_RED_MAX) {
        LnRedArgs a;
        const int m = n - i0 < LN_RED_MAX ? n - i0 : LN_RED_MAX;
        int dmax = 0;
        for (int i = 0; i < m; ++i) {
            if (!part[i0 + i] || records[i0 + i] <= 0 || (D[i0 + i] & 3)) return VITAE_ERR_INVALID_ARG;
            a.d[i] = LnRedDesc{part[i0 + i], dw[i0 + i], db[i0 + i], dx_colsum ? dx_colsum[i0 + i] : nullptr, records[i0 + i], D[i0 + i]};
            if (D[i0 + i] > dmax) dmax = D[i0 + i];
        }
        hipLaunchKernelGGL(ln_grad_reduce_kernel, dim3(m, cdiv(3 * dmax / 4, 16)), dim3(256), 0, (hipStream_t)stream, a);
    }
    return vitae_launch_status();
}

extern "C" int vitae_colsum_accum(const float* dy, long ld, float* out, int M, int N, void* stream) {
    if (!dy || !out || M <= 0 || N <= 0) return VITAE_ERR_INVALID_ARG;
    const int rpb = 32;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 256), cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, dy, ld, out,
                       M, N, rpb);
    return vitae_launch_status();
}

static bool bn_vec_ok(int D, std::initializer_list<const void*> ptrs) {
    if (D & 3) return false;
    for (const void* q : ptrs) if ((uintptr_t)q & 15) return false;
    return true;
}

extern "C" int vitae_bn1d_relu_fwd(const float* x, const float* w, const float* b, float* y, void* y_bf16, float* save_mean,
                                   float* save_rstd, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, int R, int D, float eps, float momentum,
                                   void* stream) {
    if (!x || !w || !b || !y || !save_mean || !save_rstd || R <= 0 || D <= 0) return VITAE_ERR_INVALID_ARG;
    if (!bn_vec_ok(D, {x, w, b, y, save_mean, save_rstd}) || ((uintptr_t)y_bf16 & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    hipLaunchKernelGGL(bn1d_relu_fwd_kernel, dim3(cdiv(D, BN_COLS)), dim3(256), 0, (hipStream_t)stream, x, w, b, y,
                       reinterpret_cast<__bf16*>(y_bf16), save_mean, save_rstd, running_mean, running_var, num_batches_tracked, R, D, eps, momentum);
    return vitae_launch_status();
}

extern "C" int vitae_bn1d_relu_eval(const float* x, const float* w, const float* b, const float* running_mean,
                                    const float* running_var, float* y, int R, int D, float eps, void* stream) {
    if (!x || !w || !b || !running_mean || !running_var || !y || R <= 0 || D <= 0) return VITAE_ERR_INVALID_ARG;
    const long n = (long)R * D;
    hipLaunchKernelGGL(bn1d_relu_eval_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, running_mean,
                       running_var, y, n, D, eps);
    return vitae_launch_status();
}

extern "C" int vitae_bn1d_relu_bwd(const float* dy, const float* x, const float* y, const float* w,
                                   const float* save_mean, const float* save_rstd, float* dx, void* dx_bf16, float* dw,
                                   float* db, int R, int D, void* stream) {
    if (!dy || !x || !y || !w || !save_mean || !save_rstd || !dx || !dw || !db || R <= 0 || D <= 0)
        return VITAE_ERR_INVALID_ARG;
    if (!bn_vec_ok(D, {dy, x, y, w, save_mean, save_rstd, dx}) || ((uintptr_t)dx_bf16 & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    hipLaunchKernelGGL(bn1d_relu_bwd_kernel, dim3(cdiv(D, BN_COLS)), dim3(256), 0, (hipStream_t)stream, dy, x, y, w,
                       save_mean, save_rstd, dx, reinterpret_cast<__bf16*>(dx_bf16), dw, db, R, D);
    return vitae_launch_status();
}

// Row-split forms (two launches each): for many rows — one 64-column strip per workgroup leaves 244 of 256 CUs idle at D = 768.
// ws: vitae_bn1d_split_ws_floats(R, D) floats, used between the two launches of a call only.
extern "C" long vitae_bn1d_split_ws_floats(int R, int D) {
    if (R <= 0 || D <= 0) return 0;
    return (long)bn_split_plan(R, D).RS * 2 * D;
}

extern "C" int vitae_bn1d_relu_fwd_split(const float* x, const float* w, const float* b, float* y, void* y_bf16, float* save_mean,
                                         float* save_rstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                                         int R, int D, float eps, float momentum, float* ws, void* stream) {
    if (!x || !w || !b || !y || !save_mean || !save_rstd || !ws || R <= 0 || D <= 0) return VITAE_ERR_INVALID_ARG;
    if (!bn_vec_ok(D, {x, w, b, y, save_mean, save_rstd, ws}) || ((uintptr_t)y_bf16 & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const BnSplit sp = bn_split_plan(R, D);
    const dim3 grid(cdiv(D, BN_COLS), sp.RS);
    hipLaunchKernelGGL(bn1d_stats_part_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ws, R, D, sp);
    hipLaunchKernelGGL(bn1d_relu_apply_part_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, ws, y, reinterpret_cast<__bf16*>(y_bf16),
                       save_mean, save_rstd, running_mean, running_var, num_batches_tracked, R, D, eps, momentum, sp);
    return vitae_launch_status();
}

extern "C" int vitae_bn1d_relu_bwd_split(const float* dy, const float* x, const float* y, const float* w, const float* save_mean,
                                         const float* save_rstd, float* dx, void* dx_bf16, float* dw, float* db, int R, int D,
                                         float* ws, void* stream) {
    if (!dy || !x || !y || !w || !save_mean || !save_rstd || !dx || !dw || !db || !ws || R <= 0 || D <= 0) return VITAE_ERR_INVALID_ARG;
    if (!bn_vec_ok(D, {dy, x, y, w, save_mean, save_rstd, dx, ws}) || ((uintptr_t)dx_bf16 & 7)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const BnSplit sp = bn_split_plan(R, D);
    const dim3 grid(cdiv(D, BN_COLS), sp.RS);
    hipLaunchKernelGGL(bn1d_relu_bwd_sums_part_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, y, save_mean, save_rstd, ws, R, D, sp);
    hipLaunchKernelGGL(bn1d_relu_bwd_apply_part_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, y, w, save_mean, save_rstd, ws, dx,
                       reinterpret_cast<__bf16*>(dx_bf16), dw, db, R, D, sp);
    return vitae_launch_status();
}
