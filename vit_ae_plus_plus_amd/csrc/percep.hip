// Building blocks of the perceptual-loss hook (reference: model/model_utils/perceptual_loss.py:46-77 — every z-slice of
// every channel of the predicted and the target volume goes through VGG16's first ten 3x3 convolutions; the loss is the mean
// MSE of the relu1_2 / relu2_2 / relu3_3 / relu4_3 feature maps).  It is a no-gradient logging term (torch.no_grad at
// model/vit_autoenc.py:229-230, weight 0 in the shipped configuration), so only a forward exists.
// The convolutions run on the bf16 LDS-DMA GEMM (gemm_glds.hip, bias + ReLU epilogue) over an im2col matrix in NHWC order:
// these kernels build that matrix, pool, and reduce the squared feature difference.  All HBM-bound index kernels:
// 16-byte accesses, one pass each.
#include "common.hpp"
#include "vitae_hip.h"

namespace {

// First layer.  The reference feeds each (batch, z) slice of ONE channel as a 3-channel image of three identical copies
// (perceptual_loss.py:50-52), so conv1_1 sees the 3x3 neighbourhood only: A[(img, y, x), ky * 3 + kx] (K padded to 64 with
// zeros; the matching weight is the sum of conv1_1's weight over its three input channels).  img = v * B * Z + b * Z + z
// with v = 0 for `vol0` (the prediction) and 1 for `vol1` (the target); images img0 .. img0 + n_img - 1 of each.
__global__ __launch_bounds__(256) void percep_im2col_first_kernel(const float* __restrict__ vol0, const float* __restrict__ vol1,
                                                                  __bf16* __restrict__ A, int C, int ch, int Z, int H, int W,
                                                                  int img0, int n_img) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    const long rows = 2L * n_img * H * W;
    if (row >= rows) return;
    const int x = (int)(row % W), y = (int)((row / W) % H);
    const long im = row / ((long)W * H);
    const int v = (int)(im / n_img), gi = img0 + (int)(im % n_img);
    const int b = gi / Z, z = gi % Z;
    const float* src = (v ? vol1 : vol0) + (((long)b * C + ch) * Z + z) * (long)H * W;
    __bf16 out[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) out[i] = (__bf16)0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) out[ky * 3 + kx] = (__bf16)src[(long)yy * W + xx];
        }
    bf16x8* dst = reinterpret_cast<bf16x8*>(A + row * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<bf16x8*>(out + 8 * i);
}

// A[(n, y, x), (ky, kx, c)] = in[n, y + ky - 1, x + kx - 1, c] (zero padding); in is NHWC bf16, Cin % 8 == 0.
__global__ __launch_bounds__(256) void percep_im2col_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ A, long rows,
                                                            int H, int W, int Cin) {
    const int cg = Cin / 8;                       // 16-byte groups per tap
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * 9 * cg) return;
    const int g = (int)(t % cg);
    const int tap = (int)((t / cg) % 9);
    const long row = t / (9L * cg);
    const int x = (int)(row % W), y = (int)((row / W) % H);
    const long n = row / ((long)W * H);
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        v = *reinterpret_cast<const bf16x8*>(in + ((n * H + yy) * W + xx) * Cin + g * 8);
    *reinterpret_cast<bf16x8*>(A + (row * 9 + tap) * Cin + g * 8) = v;
}

// 2x2 / stride 2 max pooling, NHWC bf16
__global__ __launch_bounds__(256) void percep_maxpool_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ out, long n_img,
                                                             int H, int W, int C) {
    const int cg = C / 8, Ho = H / 2, Wo = W / 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_img * Ho * Wo * cg) return;
    const int g = (int)(t % cg);
    const long p = t / cg;
    const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho);
    const long n = p / ((long)Wo * Ho);
    const __bf16* s = in + ((n * H + 2 * yo) * W + 2 * xo) * C + g * 8;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(s), b = *reinterpret_cast<const bf16x8*>(s + C);
    const bf16x8 c = *reinterpret_cast<const bf16x8*>(s + (long)W * C), d = *reinterpret_cast<const bf16x8*>(s + (long)W * C + C);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)fmaxf(fmaxf((float)a[i], (float)b[i]), fmaxf((float)c[i], (float)d[i]));
    *reinterpret_cast<bf16x8*>(out + p * C + g * 8) = o;
}

// acc[0] += sum_i (a[i] - b[i])^2, double accumulation
__global__ __launch_bounds__(256) void percep_sqdiff_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, long n8,
                                                            double* __restrict__ acc) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const bf16x8 x = reinterpret_cast<const bf16x8*>(a)[i], y = reinterpret_cast<const bf16x8*>(b)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)x[e] - (float)y[e]; s += d * d; }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)s);
}

}  // namespace

extern "C" int vitae_percep_im2col_first(const float* vol_pred, const float* vol_target, void* A16, int B, int C, int channel,
                                         int Z, int H, int W, int img0, int n_img, void* stream) {
    if (!vol_pred || !vol_target || !A16 || B <= 0 || C <= 0 || channel < 0 || channel >= C || n_img <= 0 || img0 < 0 ||
        img0 + n_img > B * Z)
        return VITAE_ERR_INVALID_ARG;
    const long rows = 2L * n_img * H * W;
    hipLaunchKernelGGL(percep_im2col_first_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, vol_pred, vol_target,
                       reinterpret_cast<__bf16*>(A16), C, channel, Z, H, W, img0, n_img);
    return vitae_launch_status();
}

extern "C" int vitae_percep_im2col(const void* in16, void* A16, long n_img, int H, int W, int Cin, void* stream) {
    if (!in16 || !A16 || n_img <= 0 || H <= 0 || W <= 0) return VITAE_ERR_INVALID_ARG;
    if (Cin % 8) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const long rows = n_img * H * W;
    hipLaunchKernelGGL(percep_im2col_kernel, dim3(cdiv(rows * 9 * (Cin / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(in16), reinterpret_cast<__bf16*>(A16), rows, H, W, Cin);
    return vitae_launch_status();
}

extern "C" int vitae_percep_maxpool2(const void* in16, void* out16, long n_img, int H, int W, int C, void* stream) {
    if (!in16 || !out16 || n_img <= 0) return VITAE_ERR_INVALID_ARG;
    if ((C % 8) || (H & 1) || (W & 1)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    hipLaunchKernelGGL(percep_maxpool_kernel, dim3(cdiv(n_img * (H / 2) * (W / 2) * (C / 8), 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const __bf16*>(in16), reinterpret_cast<__bf16*>(out16), n_img, H, W, C);
    return vitae_launch_status();
}

extern "C" int vitae_percep_sqdiff(const void* a16, const void* b16, long n, double* acc, void* stream) {
    if (!a16 || !b16 || !acc || n <= 0) return VITAE_ERR_INVALID_ARG;
    if (n % 8) return VITAE_ERR_UNSUPPORTED_SHAPE;
    long blocks = cdiv(n / 8, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(percep_sqdiff_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(a16), reinterpret_cast<const __bf16*>(b16), n / 8, acc);
    return vitae_launch_status();
}
