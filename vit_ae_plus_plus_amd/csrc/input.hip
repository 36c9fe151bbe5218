// Input-side normalisation of the volumes, on the GPU instead of in the DataLoader workers:
//   dataset/brats_dataset/brats.py:26-32   whole-sample z-score (unbiased variance) or min-max to [-1, 1]
//   dataset/egd_dataset/egd.py:44-50       per-channel z-score, same min-max
//   */brats.py:34-37, */egd.py:52-55       min-max to [0, 1]
// A "group" is one contiguous run of n elements that is normalised together (a sample, or one channel of a sample).
// HBM-bound streaming: one statistics pass (double accumulators; min / max through order-preserving integer
// atomics) and one apply pass.
#include "common.hpp"
#include "vitae_hip.h"

namespace {

__device__ __forceinline__ unsigned int float_key(float f) {          // monotone float -> uint map
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ws per group: [0] sum, [1] sum of squares (double), [2] = {min key, max key} packed as two uint32
__global__ __launch_bounds__(256) void norm_init_kernel(double* __restrict__ ws, int G) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    ws[3 * g] = 0.0; ws[3 * g + 1] = 0.0;
    unsigned int* mm = reinterpret_cast<unsigned int*>(ws + 3 * g + 2);
    mm[0] = 0xffffffffu; mm[1] = 0u;
}

__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x, double* __restrict__ ws, long n) {
    __shared__ double rs[4], rq[4];
    __shared__ unsigned int rmin[4], rmax[4];
    const int g = blockIdx.y;
    const float* xg = x + (long)g * n;
    double s = 0.0, q = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    const long n4 = ((uintptr_t)xg & 15) ? 0 : n / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xg);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = x4[i];
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ps += v[e]; pq += v[e] * v[e]; lo = fminf(lo, v[e]); hi = fmaxf(hi, v[e]); }
        s += (double)ps; q += (double)pq;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = xg[i];
        s += (double)v; q += (double)v * (double)v; lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64);
        lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { rs[w] = s; rq[w] = q; rmin[w] = float_key(lo); rmax[w] = float_key(hi); }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ws + 3 * g, rs[0] + rs[1] + rs[2] + rs[3]);
        atomicAdd(ws + 3 * g + 1, rq[0] + rq[1] + rq[2] + rq[3]);
        unsigned int* mm = reinterpret_cast<unsigned int*>(ws + 3 * g + 2);
        atomicMin(mm, min(min(rmin[0], rmin[1]), min(rmin[2], rmin[3])));
        atomicMax(mm + 1, max(max(rmax[0], rmax[1]), max(rmax[2], rmax[3])));
    }
}

__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const double* __restrict__ ws, long n, int mode) {
    const int g = blockIdx.y;
    float sub, mul, add = 0.f;
    if (mode == VITAE_NORM_ZSCORE) {
        const double mean = ws[3 * g] / (double)n;
        const double var = (ws[3 * g + 1] - ws[3 * g] * mean) / (double)(n - 1);      // unbiased, as torch.var
        sub = (float)mean; mul = (float)(1.0 / sqrt(var));
    } else {
        const unsigned int* mm = reinterpret_cast<const unsigned int*>(ws + 3 * g + 2);
        const float lo = key_float(mm[0]), hi = key_float(mm[1]);
        sub = lo; mul = 1.f / (hi - lo);
        if (mode == VITAE_NORM_MINMAX_PM1) { mul *= 2.f; add = -1.f; }
    }
    const float* xg = x + (long)g * n;
    float* yg = y + (long)g * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        yg[i] = (xg[i] - sub) * mul + add;
}

// ---- augmentations of the pre-training scripts (k_fold_cross_valid_combined_brats.py:93-97: torchio RandomAffine ->
// RandomNoise -> RandomGamma on the raw item), as batch kernels.  The random parameters are drawn on the host
// (utils/augment.py); the kernels are deterministic functions of (input, parameters, noise tensor).

// Affine resampling, linear interpolation (what sitk.Resample with sitkLinear does for torchio's Affine): output voxel
// o = (l, h, w) takes the input at the continuous index s = A_b (l, h, w, 1)^T.  A point is inside the buffer when
// -0.5 <= s_d < n_d - 0.5 on every axis (ITK's IsInsideBuffer); outside points get the pad value; the two neighbours
// on each axis are clamped to the valid indices (ITK's LinearInterpolateImageFunction at the half-voxel border).
__global__ __launch_bounds__(256) void affine_resample_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ mats, const double* __restrict__ ws,
                                                              float pad_const, int C, int Lz, int Hy, int Wx) {
    // One thread per output voxel and channel (blockIdx.z): its eight taps are independent loads in flight together.
    // Measured at 32 x 4 x 96^3 (453 MB in, 453 MB out): 0.83 ms = 1.1 TB/s.  A channel loop inside the thread (taps of
    // different channels become dependent round trips) took 0.93 ms; a 64 x 4 voxel workgroup per plane with division-free
    // 32-bit indexing 0.99 ms.  The limit is the L2 -> L1 traffic of the taps (each wave-load touches 2-3 lines), which
    // only an LDS-staged source box would cut; at 12 k volumes/s for both views against a 0.7 k volumes/s training step
    // that is left alone.
    const int b = blockIdx.y, c = blockIdx.z;
    const long V = (long)Lz * Hy * Wx;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int w = (int)(v % Wx), h = (int)((v / Wx) % Hy), l = (int)(v / ((long)Wx * Hy));
    const float* A = mats + 12 * b;
    const float fl = (float)l, fh = (float)h, fw = (float)w;
    const float sl = A[0] * fl + A[1] * fh + A[2] * fw + A[3];
    const float sh = A[4] * fl + A[5] * fh + A[6] * fw + A[7];
    const float sw = A[8] * fl + A[9] * fh + A[10] * fw + A[11];
    const float* xc = x + ((long)b * C + c) * V;
    float* yc = y + ((long)b * C + c) * V;
    const bool inside = sl >= -0.5f && sl < (float)Lz - 0.5f && sh >= -0.5f && sh < (float)Hy - 0.5f && sw >= -0.5f &&
                        sw < (float)Wx - 0.5f;
    if (!inside) {
        yc[v] = ws ? key_float(reinterpret_cast<const unsigned int*>(ws + 3 * b + 2)[0]) : pad_const;   // the sample's minimum
        return;
    }
    const float bl = floorf(sl), bh = floorf(sh), bw = floorf(sw);
    const float tl = sl - bl, th = sh - bh, tw = sw - bw;
    const int l0 = max((int)bl, 0), l1 = min((int)bl + 1, Lz - 1);
    const int h0 = max((int)bh, 0), h1 = min((int)bh + 1, Hy - 1);
    const int w0 = max((int)bw, 0), w1 = min((int)bw + 1, Wx - 1);
    const long o00 = ((long)l0 * Hy + h0) * Wx, o01 = ((long)l0 * Hy + h1) * Wx;
    const long o10 = ((long)l1 * Hy + h0) * Wx, o11 = ((long)l1 * Hy + h1) * Wx;
    const float p000 = xc[o00 + w0], p001 = xc[o00 + w1], p010 = xc[o01 + w0], p011 = xc[o01 + w1];
    const float p100 = xc[o10 + w0], p101 = xc[o10 + w1], p110 = xc[o11 + w0], p111 = xc[o11 + w1];
    const float a00 = p000 + tw * (p001 - p000), a01 = p010 + tw * (p011 - p010);
    const float a10 = p100 + tw * (p101 - p100), a11 = p110 + tw * (p111 - p110);
    const float a0 = a00 + th * (a01 - a00), a1 = a10 + th * (a11 - a10);
    yc[v] = a0 + tl * (a1 - a0);
}

// y = gamma(x + std_b * noise): RandomNoise (additive N(0, std_b)) then RandomGamma (sign(v) |v|^gamma_b, torchio's form
// for images with negative values; gamma_b = 1 is an exact pass-through).  noise may be NULL (std ignored).
__global__ __launch_bounds__(256) void noise_gamma_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                          float* __restrict__ y, const float* __restrict__ stds,
                                                          const float* __restrict__ gammas, long n) {
    const int b = blockIdx.y;
    const float sd = stds ? stds[b] : 0.f, gm = gammas ? gammas[b] : 1.f;
    const float* xb = x + (long)b * n;
    const float* nb = noise ? noise + (long)b * n : nullptr;
    float* yb = y + (long)b * n;
    // |v|^g = exp2(g log2|v|) on the transcendental units (v_log_f32 / v_exp_f32, ~1e-6 relative): powf's software
    // path made this HBM stream VALU-bound
    auto f = [&](float v, float z) {
        if (nb) v += sd * z;
        if (gm != 1.f) v = copysignf(__builtin_amdgcn_exp2f(gm * __builtin_amdgcn_logf(fabsf(v))), v);
        return v;
    };
    const bool vec = (n & 3) == 0 && !((uintptr_t)xb & 15) && !((uintptr_t)yb & 15) && !(nb && ((uintptr_t)nb & 15));
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xb);
        const f32x4* n4 = reinterpret_cast<const f32x4*>(nb);
        f32x4* y4 = reinterpret_cast<f32x4*>(yb);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (long)gridDim.x * 256) {
            const f32x4 a = x4[i];
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (nb) z = n4[i];
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = f(a[e], z[e]);
            y4[i] = r;
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) yb[i] = f(xb[i], nb ? nb[i] : 0.f);
}

}  // namespace

extern "C" int vitae_volume_minmax(const float* x, double* ws, int groups, long n, void* stream) {
    if (!x || !ws || groups <= 0 || n <= 0 || groups > 65535) return VITAE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    long per = (n / 4 + 255) / 256;
    int bx = (int)(per < 1 ? 1 : per > 512 ? 512 : per);
    hipLaunchKernelGGL(norm_init_kernel, dim3(cdiv(groups, 256)), dim3(256), 0, st, ws, groups);
    hipLaunchKernelGGL(norm_stats_kernel, dim3(bx, groups), dim3(256), 0, st, x, ws, n);
    return vitae_launch_status();
}

extern "C" int vitae_affine_resample(const float* x, float* y, const float* mats, const double* minmax_ws, float pad_value,
                                     int B, int C, int Lz, int Hy, int Wx, void* stream) {
    if (!x || !y || !mats || x == y || B <= 0 || C <= 0 || Lz <= 0 || Hy <= 0 || Wx <= 0) return VITAE_ERR_INVALID_ARG;
    if (B > 65535 || C > 65535) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const long V = (long)Lz * Hy * Wx;
    hipLaunchKernelGGL(affine_resample_kernel, dim3(cdiv(V, 256), B, C), dim3(256), 0, (hipStream_t)stream, x, y, mats, minmax_ws,
                       pad_value, C, Lz, Hy, Wx);
    return vitae_launch_status();
}

extern "C" int vitae_noise_gamma(const float* x, const float* noise, float* y, const float* stds, const float* gammas, int B,
                                 long n, void* stream) {
    if (!x || !y || B <= 0 || B > 65535 || n <= 0 || (noise && !stds)) return VITAE_ERR_INVALID_ARG;
    long per = (n / 4 + 255) / 256;
    int bx = (int)(per > 2048 ? 2048 : per < 1 ? 1 : per);
    hipLaunchKernelGGL(noise_gamma_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, x, noise, y, stds, gammas, n);
    return vitae_launch_status();
}

extern "C" int vitae_normalize_volumes(const float* x, float* y, double* ws, int groups, long n, int mode, void* stream) {
    if (!x || !y || !ws || groups <= 0 || n <= 1 || groups > 65535) return VITAE_ERR_INVALID_ARG;
    if (mode != VITAE_NORM_ZSCORE && mode != VITAE_NORM_MINMAX_PM1 && mode != VITAE_NORM_MINMAX_01) return VITAE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    long per = (n / 4 + 255) / 256;
    int bx = (int)(per < 1 ? 1 : per > 512 ? 512 : per);
    hipLaunchKernelGGL(norm_init_kernel, dim3(cdiv(groups, 256)), dim3(256), 0, st, ws, groups);
    hipLaunchKernelGGL(norm_stats_kernel, dim3(bx, groups), dim3(256), 0, st, x, ws, n);
    hipLaunchKernelGGL(norm_apply_kernel, dim3(bx, groups), dim3(256), 0, st, x, y, ws, n, mode);
    return vitae_launch_status();
}
