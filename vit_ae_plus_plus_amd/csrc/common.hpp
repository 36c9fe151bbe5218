// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of libvitae_hip.so.
// Wave = 64 lanes everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VITAE_OK 0
#define VITAE_ERR_INVALID_ARG (-1)
#define VITAE_ERR_UNSUPPORTED_SHAPE (-2)
#define VITAE_ERR_LAUNCH (-3)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

static inline int vitae_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VITAE_OK : VITAE_ERR_LAUNCH;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over a 256-thread block (4 waves); every thread gets the result. `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Phi(x) (the GELU gate) and phi(x) from ONE exponential, branch-free: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 —
// far below the bf16 rounding of the results it is used for; the exact-fp32 parity mode keeps erff above).  ocml's erff
// costs ~140 clocks per value in a GEMM epilogue (32 values per lane were 5 us of a 21 us kernel); this is ~20 VALU
// operations.  For x < 0 the tail form 0.5 * poly * e is used directly, so small gates keep their relative precision.
__device__ __forceinline__ void gelu_gate(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * x * x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float tail = 0.5f * poly * t * e;          // 1 - Phi(|x|)
    cdf = x >= 0.f ? 1.0f - tail : tail;
    pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_fast(float x) { float c, d; gelu_gate(x, c, d); return x * c; }
__device__ __forceinline__ float gelu_fast_grad(float x) { float c, d; gelu_gate(x, c, d); return fmaf(x, d, c); }

// The same gate on FOUR values with packed fp32 VALU (v_pk_mul / v_pk_fma / v_pk_add: two lanes' worth of work per issue slot)
// and bare v_exp_f32 / v_rcp_f32.  Round 5: a GEMM epilogue that applies GELU to a 256 x 256 tile is VALU-bound — 128 values per
// lane at ~25 issue slots each (ocml's range-reduced expf alone was 9 of them) cost the fc1 launches of a batch-32 step 8.5-9.4 us
// of their 37 us (tools/epi_ablate.py).  exp(-x^2/2) = 2^(-x^2 log2(e)/2) is at most 1, and below 2^-126 (|x| > 13.2) a flushed zero
// is the right gate, so no range handling is needed; the 0.5 of the tail is folded into the polynomial's coefficients.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_gate4(const f32x4 x, f32x4& cdf, f32x4& pdf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 v = {x[2 * h], x[2 * h + 1]};
        const f32x2 av = __builtin_elementwise_max(v, -v);
        const f32x2 a = (v * v) * -0.72134752044448170368f;                     // -x^2 / 2 * log2(e)
        f32x2 e, t;
        const f32x2 den = __builtin_elementwise_fma(av, (f32x2)(0.3275911f * 0.70710678118654752440f), (f32x2)(1.0f));
#pragma unroll
        for (int i = 0; i < 2; ++i) { e[i] = __builtin_amdgcn_exp2f(a[i]); t[i] = __builtin_amdgcn_rcpf(den[i]); }
        f32x2 poly = __builtin_elementwise_fma((f32x2)(0.5f * 1.061405429f), t, (f32x2)(0.5f * -1.453152027f));
        poly = __builtin_elementwise_fma(poly, t, (f32x2)(0.5f * 1.421413741f));
        poly = __builtin_elementwise_fma(poly, t, (f32x2)(0.5f * -0.284496736f));
        poly = __builtin_elementwise_fma(poly, t, (f32x2)(0.5f * 0.254829592f));
        const f32x2 tail = (poly * t) * e;                                       // 1 - Phi(|x|)
        const f32x2 up = 1.0f - tail;
        const f32x2 pd = e * 0.39894228040143267794f;
#pragma unroll
        for (int i = 0; i < 2; ++i) { cdf[2 * h + i] = v[i] >= 0.f ? up[i] : tail[i]; pdf[2 * h + i] = pd[i]; }
    }
}
// y = GELU(x), dy = GELU'(x) from one gate (the forward epilogue that saves the derivative for the backward: VITAE_EPI_AUX_DERIV)
__device__ __forceinline__ void gelu_fast4(const f32x4 x, f32x4& y, f32x4& dy) {
    f32x4 c, d;
    gelu_gate4(x, c, d);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 v = {x[2 * h], x[2 * h + 1]}, cc = {c[2 * h], c[2 * h + 1]}, dd = {d[2 * h], d[2 * h + 1]};
        const f32x2 yy = v * cc, gg = __builtin_elementwise_fma(v, dd, cc);
        y[2 * h] = yy[0]; y[2 * h + 1] = yy[1]; dy[2 * h] = gg[0]; dy[2 * h + 1] = gg[1];
    }
}
__device__ __forceinline__ f32x4 gelu_fast4(const f32x4 x) { f32x4 y, dy; gelu_fast4(x, y, dy); return y; }
__device__ __forceinline__ f32x4 gelu_fast_grad4(const f32x4 x) { f32x4 y, dy; gelu_fast4(x, y, dy); return dy; }
// exact-erf pair (fp32-grade modes): gelu_erf(x) = (0.5 x)(1 + erf) == x (0.5 (1 + erf)) bit for bit (scaling by 0.5 is exact)
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& dy) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    y = x * cdf;
    dy = cdf + x * pdf;
}
