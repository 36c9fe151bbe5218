// Optimiser-side streaming kernels over the flat parameter / gradient arenas (HBM-bound):
//   global gradient L2 norm     utils/misc.py:265-266, 280-292  (reference op K25)
//   AdamW step                  torch.optim.AdamW(betas=(0.9, 0.95)) created at
//                               k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:168-169 (op K26)
// Hyper-parameters live in the device `hp` block (VITAE_HP_*), so a captured step graph can be
// replayed while lr / bias corrections change.  The step is skipped when the gradient norm is not
// finite — the effect of GradScaler.step's inf check (utils/misc.py:267).
#include <cstdlib>
#include "common.hpp"
#include "vitae_hip.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Gradients are read either as fp32 or as bf16 (the wire copy of a data-parallel all-reduce done in bf16: the
// optimiser then consumes the reduced values directly instead of a converted fp32 copy).
template <typename G> __device__ __forceinline__ f32x4 grad4(const G* g, long i);
template <> __device__ __forceinline__ f32x4 grad4<float>(const float* g, long i) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
}
template <> __device__ __forceinline__ f32x4 grad4<__bf16>(const __bf16* g, long i) {
    const bf16x4 h = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(g) + i);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// where workgroup b adds its share of the squared gradient norm: spread slots (same-address double atomics retire one per ~10 ns)
__device__ __forceinline__ double* gradsq_slot(double* acc) {
    return acc + VITAE_ACC_SQ_BASE + ((int)blockIdx.x & (VITAE_ACC_SQ_SLOTS - 1)) * VITAE_ACC_SQ_STRIDE;
}

template <typename G>
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const G* __restrict__ g, long n, double* __restrict__ acc) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = grad4<G>(g, i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += (float)g[i] * (float)g[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(gradsq_slot(acc), (double)s);
}

// one wave: lane s reads spread slot s (vitae_hip.h VITAE_ACC_SQ_*), lane 0 adds acc[GRADSQ]
__global__ __launch_bounds__(64) void grad_norm_finalize_kernel(const double* __restrict__ acc, float* __restrict__ out) {
    static_assert(VITAE_ACC_SQ_SLOTS == 64, "one slot per lane");
    double v = acc[VITAE_ACC_SQ_BASE + threadIdx.x * VITAE_ACC_SQ_STRIDE];
    if (threadIdx.x == 0) v += acc[VITAE_ACC_GRADSQ];
#pragma unroll
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)sqrt(v);
}

// 1 - beta^t on the device.  A NEGATIVE slot carries -(1 - beta) from the host's double (relative error 6e-8; fp32(0.999) itself is
// 1.3e-8 off, which is 1.3e-5 of 1 - 0.999^t for small t): 1 - beta^t = -expm1(t log1p(-(1 - beta))).  Slot == 0: from fp32 beta.
__device__ __forceinline__ void device_bias_corrections(float b1, float b2, float t, float& bc1, float& bc2) {
    const float omb1 = bc1 < 0.f ? -bc1 : 1.f - b1, omb2 = bc2 < 0.f ? -bc2 : 1.f - b2;
    bc1 = -expm1f(t * log1pf(-omb1));
    bc2 = -expm1f(t * log1pf(-omb2));
}

// The moments are stored either as fp32 or as bf16 (round 6: `S`).  bf16 moments: the step is COMPUTED in fp32 from the stored
// values — m_new and v_new enter the parameter update unrounded — and only what is written back for the next step is rounded, so the
// stored moment carries a relative error of 2^-9 per step that decays with beta (tools/opt_state_ablation.py: the pinned ViT-B
// trajectory moves by 2e-7..2e-6, an 80-step loss curve by 1e-7, where another masking seed moves it by 1e-3).  8 B per parameter
// less HBM traffic (30 -> 22).  Round-to-nearest would stall an update smaller than half an ulp: the launchers refuse 1 - beta < 2^-6.
template <typename S> __device__ __forceinline__ f32x4 state4_ld(const S* s, long i);
template <> __device__ __forceinline__ f32x4 state4_ld<float>(const float* s, long i) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s) + i);
}
template <> __device__ __forceinline__ f32x4 state4_ld<__bf16>(const __bf16* s, long i) {
    const bf16x4 h = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(s) + i);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <typename S> __device__ __forceinline__ void state4_st(S* s, long i, const f32x4 v);
template <> __device__ __forceinline__ void state4_st<float>(float* s, long i, const f32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(s) + i);
}
template <> __device__ __forceinline__ void state4_st<__bf16>(__bf16* s, long i, const f32x4 v) {
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
    __builtin_nontemporal_store(h, reinterpret_cast<bf16x4*>(s) + i);
}

// torch.optim.AdamW (single-tensor form): p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
template <bool NT, typename G, int U = 8, typename S = float>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const G* __restrict__ g, S* __restrict__ m,
                                                    S* __restrict__ v, __bf16* __restrict__ shadow, long n,
                                                    const float* __restrict__ hp, const float* __restrict__ gnorm,
                                                    float weight_decay, const double* __restrict__ gacc = nullptr) {
    if (gnorm) { const float gn = gnorm[0]; if (!(gn == gn) || fabsf(gn) == INFINITY) return; }
    if (gacc) {
        // the gate straight from the accumulator (round 6): what grad_norm_finalize_kernel would have written for this bucket — every
        // wave sums the 64 spread slots + acc[GRADSQ] itself (65 L2 hits) instead of one more 4-5 us node in front of every bucket
        double q = gacc[VITAE_ACC_SQ_BASE + (threadIdx.x & 63) * VITAE_ACC_SQ_STRIDE];
        if ((threadIdx.x & 63) == 0) q += gacc[VITAE_ACC_GRADSQ];
#pragma unroll
        for (int d = 32; d; d >>= 1) q += __shfl_xor(q, d, 64);
        const float gn = (float)sqrt(q);
        if (!(gn == gn) || fabsf(gn) == INFINITY) return;
    }
    const float lr = hp[VITAE_HP_LR], b1 = hp[VITAE_HP_BETA1], b2 = hp[VITAE_HP_BETA2], eps = hp[VITAE_HP_EPS];
    float bc1 = hp[VITAE_HP_BC1], bc2 = hp[VITAE_HP_BC2];
    if (bc1 <= 0.f) {            // the host left the bias corrections to the device: t = applied steps + 1 (vitae_hip.h VITAE_HP_STEP)
        const float t = hp[VITAE_HP_STEP] + 1.f;
        device_bias_corrections(b1, b2, t, bc1, bc2);
    }
    const float sq_bc2 = sqrtf(bc2);
    const float gs = hp[VITAE_HP_GRAD_MUL];
    const float decay = 1.0f - lr * weight_decay, step = lr / bc1;
    const long n4 = n / 4;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    // two independent 16-byte groups per thread and iteration: 8 loads in flight before the first dependent use.
    // The moments stream through (nothing re-reads them for a whole step): non-temporal loads and stores keep them
    // from evicting the bf16 shadow / activations out of L2 and the MALL.
    // U independent 16-byte groups per thread and iteration (4 U loads in flight before the first dependent use): the pass
    // runs on one workgroup per CU, so its rate is set by the bytes each wave keeps in flight
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += U * stride) {
        long idx[U];
        bool live[U];
        f32x4 pp[U], mm[U], vv[U], gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = i0 + u * stride < n4;
            idx[u] = live[u] ? i0 + u * stride : i0;
            pp[u] = p4[idx[u]];
            mm[u] = state4_ld<S>(m, idx[u]);
            vv[u] = state4_ld<S>(v, idx[u]);
            gg[u] = grad4<G>(g, idx[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) break;
            const long i = idx[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = gg[u][e] * gs;
                pp[u][e] *= decay;
                mm[u][e] = mm[u][e] + (1.f - b1) * (ge - mm[u][e]);   // exp_avg.lerp_(grad, 1 - beta1)
                vv[u][e] = b2 * vv[u][e] + (1.f - b2) * ge * ge;
                pp[u][e] -= step * (mm[u][e] / (sqrtf(vv[u][e]) / sq_bc2 + eps));
            }
            p4[i] = pp[u];
            state4_st<S>(m, i, mm[u]);
            state4_st<S>(v, i, vv[u]);
            if (shadow) {
                bf16x4 sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) sh[e] = (__bf16)pp[u][e];
                reinterpret_cast<bf16x4*>(shadow)[i] = sh;
            }
        }
    }
    if (blockIdx.x == 0) {
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gg = (float)g[i] * gs;
            float pp = p[i] * decay;
            const float mo = (float)m[i];
            const float mm = mo + (1.f - b1) * (gg - mo);
            const float vv = b2 * (float)v[i] + (1.f - b2) * gg * gg;
            pp -= step * (mm / (sqrtf(vv) / sq_bc2 + eps));
            p[i] = pp; m[i] = (S)mm; v[i] = (S)vv;
            if (shadow) shadow[i] = (__bf16)pp;
        }
    }
}

__global__ void opt_count_bump_kernel(float* __restrict__ hp, const float* __restrict__ gnorm) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float gn = gnorm ? gnorm[0] : 0.f;
        if (gn == gn && fabsf(gn) != INFINITY) hp[VITAE_HP_STEP] += 1.f;
    }
}

// ---- the tail of a step: tokens + vectors (0.2 M elements) in two launches
template <typename G>
__global__ __launch_bounds__(256) void opt_tail_norm_kernel(const G* __restrict__ g, long n, double* __restrict__ acc, float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ int last;
    float s = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = grad4<G>(g, i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += (float)g[i] * (float)g[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        atomicAdd(gradsq_slot(acc), (double)s);
        __threadfence();
        last = atomicAdd(reinterpret_cast<int*>(acc + VITAE_ACC_TICKET_A), 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        // every other workgroup's double atomic is ordered before its ticket: read the slots through the atomic unit too
        double tot = atomicAdd(acc + VITAE_ACC_SQ_BASE + threadIdx.x * VITAE_ACC_SQ_STRIDE, 0.0);
        if (threadIdx.x == 0) tot += atomicAdd(acc + VITAE_ACC_GRADSQ, 0.0);
#pragma unroll
        for (int d = 32; d; d >>= 1) tot += __shfl_xor(tot, d, 64);
        if (threadIdx.x == 0) out[0] = (float)sqrt(tot);
    }
}

template <typename G, typename S = float>
__global__ __launch_bounds__(256) void opt_tail_adamw_kernel(float* __restrict__ p, const G* __restrict__ g, S* __restrict__ m,
                                                             S* __restrict__ v, __bf16* __restrict__ shadow, long n_decay, long n_plain,
                                                             float* __restrict__ hp, double* __restrict__ acc,
                                                             const float* __restrict__ gnorm, float weight_decay) {
    const float gn = gnorm[0];
    const bool finite = gn == gn && fabsf(gn) != INFINITY;
    if (finite) {
        const float lr = hp[VITAE_HP_LR], b1 = hp[VITAE_HP_BETA1], b2 = hp[VITAE_HP_BETA2], eps = hp[VITAE_HP_EPS];
        float bc1 = hp[VITAE_HP_BC1], bc2 = hp[VITAE_HP_BC2];
        if (bc1 <= 0.f) {
            const float t = hp[VITAE_HP_STEP] + 1.f;
            device_bias_corrections(b1, b2, t, bc1, bc2);
        }
        const float sq_bc2 = sqrtf(bc2), gs = hp[VITAE_HP_GRAD_MUL], step = lr / bc1;
        const long n = n_decay + n_plain;                  // both segment lengths are multiples of 4 (arena alignment)
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
            const float decay = 1.0f - lr * (i < n_decay ? weight_decay : 0.f);
            f32x4 pp = *reinterpret_cast<f32x4*>(p + i), mm = state4_ld<S>(m, i / 4), vv = state4_ld<S>(v, i / 4);
            const f32x4 gg = grad4<G>(g, i / 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = gg[e] * gs;
                pp[e] *= decay;
                mm[e] = mm[e] + (1.f - b1) * (ge - mm[e]);
                vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
                pp[e] -= step * (mm[e] / (sqrtf(vv[e]) / sq_bc2 + eps));
            }
            *reinterpret_cast<f32x4*>(p + i) = pp;
            state4_st<S>(m, i / 4, mm);
            state4_st<S>(v, i / 4, vv);
            if (shadow) {
                bf16x4 sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) sh[e] = (__bf16)pp[e];
                *reinterpret_cast<bf16x4*>(shadow + i) = sh;
            }
        }
    }
    // the LAST workgroup to get here counts the step (every workgroup has read hp[STEP] by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = atomicAdd(reinterpret_cast<int*>(acc + VITAE_ACC_TICKET_B), 1) == (int)gridDim.x - 1;
        if (last && finite) hp[VITAE_HP_STEP] += 1.f;
    }
}

// ---- the head of a step: hp upload from the pinned ring, masking noise, zeroing — one launch inside the captured step
__device__ __forceinline__ void philox4x32_10(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void step_prologue_kernel(float* __restrict__ hp, const float* __restrict__ ring, int slots,
                                                            const long long* __restrict__ seq, float* __restrict__ noise, long n_noise,
                                                            unsigned long long seed, double* __restrict__ acc,
                                                            unsigned int* __restrict__ zp, long zwords) {
    const long long s = *seq;
    const float* src = ring + (s % slots) * VITAE_HP_COUNT;      // pinned host memory, read over the link (64 bytes)
    if (blockIdx.x == 0) {
        if (threadIdx.x < VITAE_HP_HOST_COUNT) hp[threadIdx.x] = src[threadIdx.x];
        for (int i = threadIdx.x; i < VITAE_ACC_COUNT; i += 256) acc[i] = 0.0;
    }
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
    if (noise && src[VITAE_HP_NOISE_KEEP] == 0.f) {
        // four uniforms per Philox call: counter = (group index, step sequence number), key = seed; [0, 1) with 24 random bits
        for (long i = tid; i * 4 < n_noise; i += nth) {
            unsigned r[4];
            philox4x32_10((unsigned)seed, (unsigned)(seed >> 32), (unsigned)i, (unsigned)(i >> 32), (unsigned)s, (unsigned)(s >> 32), r);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i * 4 + e < n_noise) noise[i * 4 + e] = (float)(r[e] >> 8) * (1.0f / 16777216.0f);
        }
    }
    if (zp) {
        const long n4 = zwords / 4;
        u32x4_t* z4 = reinterpret_cast<u32x4_t*>(zp);
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (long i = tid; i < n4; i += nth) z4[i] = z;
        if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < zwords; i += 256) zp[i] = 0u;
    }
}

__global__ void step_epilogue_kernel(long long* __restrict__ seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *seq += 1;
}

}  // namespace

template <typename G>
static int grad_sqnorm_launch(const G* grads, long n, double* acc, float* norm_out, void* stream) {
    if (!grads || !acc || n <= 0 || ((uintptr_t)grads & 15)) return VITAE_ERR_INVALID_ARG;
    long blocks = (n / 4 + 255) / 256;
    static const long maxg = getenv("VITAE_GRADNORM_MAX_BLOCKS") ? atol(getenv("VITAE_GRADNORM_MAX_BLOCKS")) : 2048;
    if (blocks > maxg) blocks = maxg;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_sqnorm_kernel<G>, dim3((int)blocks), dim3(256), 0, st, grads, n, acc);
    if (norm_out) hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(64), 0, st, acc, norm_out);
    return vitae_launch_status();
}

extern "C" int vitae_grad_norm_finalize(const double* acc, float* norm_out, void* stream) {
    if (!acc || !norm_out) return VITAE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, norm_out);
    return vitae_launch_status();
}

extern "C" int vitae_grad_sqnorm(const float* grads, long n, double* acc, float* norm_out, void* stream) {
    return grad_sqnorm_launch<float>(grads, n, acc, norm_out, stream);
}

extern "C" int vitae_grad_sqnorm_bf16(const void* grads_bf16, long n, double* acc, float* norm_out, void* stream) {
    return grad_sqnorm_launch<__bf16>(reinterpret_cast<const __bf16*>(grads_bf16), n, acc, norm_out, stream);
}

template <typename G, typename S = float>
static int adamw_launch(float* params, const G* grads, S* exp_avg, S* exp_avg_sq, void* shadow_bf16, long n,
                        const float* hp, const float* grad_norm, float weight_decay, void* stream, const double* gacc = nullptr) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !hp || n <= 0) return VITAE_ERR_INVALID_ARG;
    if (((uintptr_t)params & 15) || (((uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & (4 * sizeof(S) - 1))) return VITAE_ERR_INVALID_ARG;
    if (((uintptr_t)grads & (4 * sizeof(G) - 1)) || ((uintptr_t)shadow_bf16 & 7)) return VITAE_ERR_INVALID_ARG;
    long blocks = (n / 4 + 255) / 256;
    // one 256-thread workgroup per CU: as fast alone as any larger grid (5.4-5.7 TB/s) and it leaves the CUs' other wave
    // slots to the backward kernels this pass runs beside (in-backward optimiser); fewer workgroups lose bandwidth
    static const long maxb = getenv("VITAE_ADAMW_MAX_BLOCKS") ? atol(getenv("VITAE_ADAMW_MAX_BLOCKS")) : 256;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    // 30 B/element of HBM traffic; 5.0-5.7 TB/s for every grid size >= 256 workgroups and cache policy tried (the
    // read-only grad-norm pass reaches 5.4 TB/s on the same box), i.e. this kernel sits at the achievable HBM rate.
    // groups in flight per thread: 2 -> 8 took the pass from 5.0 to 6.1 TB/s alone and the step from 5.58 to 5.18 ms (it
    // spends less time beside the backward kernels it slows down); 12 and 16 are slower again (5.21 / 5.32 ms)
    static const int unroll = getenv("VITAE_ADAMW_UNROLL") ? atoi(getenv("VITAE_ADAMW_UNROLL")) : 8;
    if (unroll == 4)
        hipLaunchKernelGGL((adamw_kernel<true, G, 4, S>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                           exp_avg_sq, reinterpret_cast<__bf16*>(shadow_bf16), n, hp, grad_norm, weight_decay, gacc);
    else if (unroll == 12)
        hipLaunchKernelGGL((adamw_kernel<true, G, 12, S>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                           exp_avg_sq, reinterpret_cast<__bf16*>(shadow_bf16), n, hp, grad_norm, weight_decay, gacc);
    else if (unroll == 8)
        hipLaunchKernelGGL((adamw_kernel<true, G, 8, S>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                           exp_avg_sq, reinterpret_cast<__bf16*>(shadow_bf16), n, hp, grad_norm, weight_decay, gacc);
    else
        hipLaunchKernelGGL((adamw_kernel<true, G, 2, S>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                           exp_avg_sq, reinterpret_cast<__bf16*>(shadow_bf16), n, hp, grad_norm, weight_decay, gacc);
    return vitae_launch_status();
}

// bf16 moments (round 6): the same pass at 22 instead of 30 bytes per parameter.  grads: fp32, or the bf16 wire copy when grads_bf16.
extern "C" int vitae_adamw_step_s16(float* params, const void* grads, int grads_bf16, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                                    void* shadow_bf16, long n, const float* hp, const float* grad_norm, float weight_decay, void* stream) {
    __bf16* m = reinterpret_cast<__bf16*>(exp_avg_bf16);
    __bf16* v = reinterpret_cast<__bf16*>(exp_avg_sq_bf16);
    if (grads_bf16)
        return adamw_launch<__bf16, __bf16>(params, reinterpret_cast<const __bf16*>(grads), m, v, shadow_bf16, n, hp, grad_norm, weight_decay, stream);
    return adamw_launch<float, __bf16>(params, reinterpret_cast<const float*>(grads), m, v, shadow_bf16, n, hp, grad_norm, weight_decay, stream);
}

// ... gated by the gradient-norm accumulator itself (acc: the double[VITAE_ACC_COUNT] block; the step is skipped when the sum of
// acc[VITAE_ACC_GRADSQ] and the VITAE_ACC_SQ_* slots is not finite — what vitae_grad_norm_finalize + grad_norm would say, without that launch)
extern "C" int vitae_adamw_step_s16_acc(float* params, const void* grads, int grads_bf16, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                                        void* shadow_bf16, long n, const float* hp, const double* acc, float weight_decay, void* stream) {
    if (!acc) return VITAE_ERR_INVALID_ARG;
    __bf16* m = reinterpret_cast<__bf16*>(exp_avg_bf16);
    __bf16* v = reinterpret_cast<__bf16*>(exp_avg_sq_bf16);
    if (grads_bf16)
        return adamw_launch<__bf16, __bf16>(params, reinterpret_cast<const __bf16*>(grads), m, v, shadow_bf16, n, hp, nullptr, weight_decay, stream, acc);
    return adamw_launch<float, __bf16>(params, reinterpret_cast<const float*>(grads), m, v, shadow_bf16, n, hp, nullptr, weight_decay, stream, acc);
}

extern "C" int vitae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                void* shadow_bf16, long n, const float* hp, const float* grad_norm,
                                float weight_decay, void* stream) {
    return adamw_launch<float>(params, grads, exp_avg, exp_avg_sq, shadow_bf16, n, hp, grad_norm, weight_decay, stream);
}

extern "C" int vitae_adamw_step_bf16g(float* params, const void* grads_bf16, float* exp_avg, float* exp_avg_sq,
                                      void* shadow_bf16, long n, const float* hp, const float* grad_norm,
                                      float weight_decay, void* stream) {
    return adamw_launch<__bf16>(params, reinterpret_cast<const __bf16*>(grads_bf16), exp_avg, exp_avg_sq, shadow_bf16, n, hp,
                                grad_norm, weight_decay, stream);
}

extern "C" int vitae_opt_count_bump(float* hp, const float* grad_norm, void* stream) {
    if (!hp) return VITAE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(opt_count_bump_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hp, grad_norm);
    return vitae_launch_status();
}

extern "C" int vitae_opt_tail(float* params, const void* grads, int grads_bf16, void* exp_avg, void* exp_avg_sq, int state_bf16, void* shadow_bf16,
                              long n_decay, long n_plain, float* hp, double* acc, float* norm_out, float weight_decay, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !hp || !acc || !norm_out || n_decay < 0 || n_plain < 0 || n_decay + n_plain <= 0)
        return VITAE_ERR_INVALID_ARG;
    const uintptr_t smask = state_bf16 ? 7 : 15;
    if ((n_decay & 3) || (n_plain & 3) || (((uintptr_t)params | (uintptr_t)grads) & 15) || (((uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & smask) ||
        ((uintptr_t)shadow_bf16 & 7))
        return VITAE_ERR_UNSUPPORTED_SHAPE;
    const long n = n_decay + n_plain;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipStream_t st = (hipStream_t)stream;
    __bf16* sh = reinterpret_cast<__bf16*>(shadow_bf16);
    float *m32 = reinterpret_cast<float*>(exp_avg), *v32 = reinterpret_cast<float*>(exp_avg_sq);
    __bf16 *m16 = reinterpret_cast<__bf16*>(exp_avg), *v16 = reinterpret_cast<__bf16*>(exp_avg_sq);
    if (grads_bf16) {
        const __bf16* g = reinterpret_cast<const __bf16*>(grads);
        hipLaunchKernelGGL(opt_tail_norm_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, st, g, n, acc, norm_out);
        if (state_bf16) hipLaunchKernelGGL((opt_tail_adamw_kernel<__bf16, __bf16>), dim3((int)blocks), dim3(256), 0, st, params, g, m16, v16, sh, n_decay, n_plain, hp, acc, norm_out, weight_decay);
        else hipLaunchKernelGGL((opt_tail_adamw_kernel<__bf16, float>), dim3((int)blocks), dim3(256), 0, st, params, g, m32, v32, sh, n_decay, n_plain, hp, acc, norm_out, weight_decay);
    } else {
        const float* g = reinterpret_cast<const float*>(grads);
        hipLaunchKernelGGL(opt_tail_norm_kernel<float>, dim3((int)blocks), dim3(256), 0, st, g, n, acc, norm_out);
        if (state_bf16) hipLaunchKernelGGL((opt_tail_adamw_kernel<float, __bf16>), dim3((int)blocks), dim3(256), 0, st, params, g, m16, v16, sh, n_decay, n_plain, hp, acc, norm_out, weight_decay);
        else hipLaunchKernelGGL((opt_tail_adamw_kernel<float, float>), dim3((int)blocks), dim3(256), 0, st, params, g, m32, v32, sh, n_decay, n_plain, hp, acc, norm_out, weight_decay);
    }
    return vitae_launch_status();
}

extern "C" int vitae_step_prologue(float* hp, const float* hp_ring, int ring_slots, const long long* step_seq, float* noise, long n_noise,
                                   long long seed, double* acc, void* zero_ptr, long zero_bytes, void* stream) {
    if (!hp || !hp_ring || ring_slots <= 0 || !step_seq || !acc || n_noise < 0 || zero_bytes < 0 || (zero_bytes & 3) ||
        ((uintptr_t)zero_ptr & 15))
        return VITAE_ERR_INVALID_ARG;
    const long work = (n_noise / 4 > zero_bytes / 16 ? n_noise / 4 : zero_bytes / 16);
    long blocks = (work + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(step_prologue_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, hp, hp_ring, ring_slots, step_seq,
                       noise, n_noise, (unsigned long long)seed, acc, reinterpret_cast<unsigned int*>(zero_ptr), zero_bytes / 4);
    return vitae_launch_status();
}

extern "C" int vitae_step_epilogue(long long* step_seq, void* stream) {
    if (!step_seq) return VITAE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(step_epilogue_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_seq);
    return vitae_launch_status();
}

namespace {
// Own fill kernel instead of hipMemsetAsync: as a captured graph memset node the latter left a few words
// of a 787 KB region unwritten on ROCm 7.2 (seen as garbage in the cls_token gradient under graph replay).
__global__ __launch_bounds__(256) void zero_kernel(unsigned int* __restrict__ p, long nwords) {
    const long n4 = nwords / 4;
    u32x4_t* p4 = reinterpret_cast<u32x4_t*>(p);
    const u32x4_t z = {0u, 0u, 0u, 0u};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p4[i] = z;
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < nwords; i += 256) p[i] = 0u;
}
}  // namespace

extern "C" int vitae_memset_zero(void* ptr, long bytes, void* stream) {
    if (!ptr || bytes < 0 || (bytes & 3) || ((uintptr_t)ptr & 15)) return VITAE_ERR_INVALID_ARG;
    if (bytes == 0) return VITAE_OK;
    const long nwords = bytes / 4;
    long blocks = (nwords / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned int*>(ptr), nwords);
    return vitae_launch_status();
}

extern "C" int vitae_abi_version(void) { return VITAE_ABI_VERSION; }

extern "C" const char* vitae_build_arch(void) { return "gfx950"; }
