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

}  // namespace

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
