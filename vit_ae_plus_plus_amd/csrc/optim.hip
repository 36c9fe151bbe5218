// Optimiser-side streaming kernels over the flat parameter / gradient arenas (HBM-bound):
//   global gradient L2 norm     utils/misc.py:265-266, 280-292  (reference op K25)
//   AdamW step                  torch.optim.AdamW(betas=(0.9, 0.95)) created at
//                               k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:168-169 (op K26)
// Hyper-parameters live in the device `hp` block (VITAE_HP_*), so a captured step graph can be
// replayed while lr / bias corrections change.  The step is skipped when the gradient norm is not
// finite — the effect of GradScaler.step's inf check (utils/misc.py:267).
#include "common.hpp"
#include "vitae_hip.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ acc) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n / 4;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = g4[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc + VITAE_ACC_GRADSQ, (double)s);
}

__global__ void grad_norm_finalize_kernel(const double* __restrict__ acc, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)sqrt(acc[VITAE_ACC_GRADSQ]);
}

// torch.optim.AdamW (single-tensor form): p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, __bf16* __restrict__ shadow, long n,
                                                    const float* __restrict__ hp, const float* __restrict__ gnorm,
                                                    float weight_decay) {
    if (gnorm) { const float gn = gnorm[0]; if (!(gn == gn) || fabsf(gn) == INFINITY) return; }
    const float lr = hp[VITAE_HP_LR], b1 = hp[VITAE_HP_BETA1], b2 = hp[VITAE_HP_BETA2], eps = hp[VITAE_HP_EPS];
    const float bc1 = hp[VITAE_HP_BC1], sq_bc2 = sqrtf(hp[VITAE_HP_BC2]);
    const float gs = hp[VITAE_HP_GRAD_MUL];
    const float decay = 1.0f - lr * weight_decay, step = lr / bc1;
    const long n4 = n / 4;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
        const f32x4 gg = g4[i] * gs;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pp[e] *= decay;
            mm[e] = mm[e] + (1.f - b1) * (gg[e] - mm[e]);   // exp_avg.lerp_(grad, 1 - beta1)
            vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
            pp[e] -= step * (mm[e] / (sqrtf(vv[e]) / sq_bc2 + eps));
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (shadow) {
            bf16x4 sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) sh[e] = (__bf16)pp[e];
            reinterpret_cast<bf16x4*>(shadow)[i] = sh;
        }
    }
    if (blockIdx.x == 0) {
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gg = g[i] * gs;
            float pp = p[i] * decay;
            const float mm = m[i] + (1.f - b1) * (gg - m[i]);
            const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
            pp -= step * (mm / (sqrtf(vv) / sq_bc2 + eps));
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (shadow) shadow[i] = (__bf16)pp;
        }
    }
}

}  // namespace

extern "C" int vitae_grad_sqnorm(const float* grads, long n, double* acc, float* norm_out, void* stream) {
    if (!grads || !acc || n <= 0 || ((uintptr_t)grads & 15)) return VITAE_ERR_INVALID_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((int)blocks), dim3(256), 0, st, grads, n, acc);
    if (norm_out) hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(64), 0, st, acc, norm_out);
    return vitae_launch_status();
}

extern "C" int vitae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                void* shadow_bf16, long n, const float* hp, const float* grad_norm,
                                float weight_decay, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !hp || n <= 0) return VITAE_ERR_INVALID_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return VITAE_ERR_INVALID_ARG;
    if ((uintptr_t)shadow_bf16 & 7) return VITAE_ERR_INVALID_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, reinterpret_cast<__bf16*>(shadow_bf16), n, hp, grad_norm, weight_decay);
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
