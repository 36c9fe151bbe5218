// MFMA issue-rate probe (round 5): what does ONE wave per SIMD get out of v_mfma_f32_32x32x16_bf16 on gfx950, as a function of
//   ACCS   independent accumulators cycled (the ws128 consumer: 4; ws64: 2; bt256: 8)
//   MODE 0 bare MFMAs   1 + a co-resident idle wave per SIMD (parked at a barrier)   2 + a co-resident wave issuing LDS-DMA pieces
//        3 + 8 ds_read_b128 per 8 MFMAs (the ws consumer's fragment reads)            4 = 2 and 3 together
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACCS, int MODE>
__global__ __launch_bounds__(512) void probe(const f32x4* __restrict__ src, int iters, long long* out, float* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[96 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4) {
        if (MODE == 2 || MODE == 4) {
            // producer: 8 pieces (1 KB each) per "k-tile", three tiles in flight, paced by the barrier like the ws kernel
            const long base = ((long)blockIdx.x * 7919 * 64) % 1000000;
            for (int t = 0; t < iters; ++t) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const long idx = base + ((long)(t % 64) * 32 + (wave - 4) * 8 + p) * 64 + lane;
                    __builtin_amdgcn_global_load_lds(src + idx, (__attribute__((address_space(3))) void*)(smem + ((t % 3) * 32 + (wave - 4) * 8 + p) * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int t = 0; t < iters; ++t) __builtin_amdgcn_s_barrier();
        }
        return;
    }
    f32x16 acc[ACCS];
#pragma unroll
    for (int a = 0; a < ACCS; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { fa[k][e] = (__bf16)(float)(lane + k); fb[k][e] = (__bf16)1.0f; }
    const unsigned lbase = (unsigned)(lane * 16 + wave * 4096);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < iters; ++t) {
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[k]) : "v"(lbase), "n"(k * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[k]) : "v"(lbase), "n"(16384 + k * 1024));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(fa[k])); asm volatile("" : "+v"(fb[k])); }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 16; ++m)
            acc[m % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m & 3], fb[(m >> 2) & 3], acc[m % ACCS], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (MODE != 0) __builtin_amdgcn_s_barrier();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < ACCS; ++a) s += acc[a][0] + acc[a][7];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

template <int ACCS, int MODE>
void run(const f32x4* src, int blocks) {
    const int iters = 2000;
    long long* out; float* sink;
    hipMalloc(&out, blocks * 8); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<ACCS, MODE>), dim3(blocks), dim3(512), 0, 0, src, iters, out, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<ACCS, MODE>), dim3(blocks), dim3(512), 0, 0, src, iters, out, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double clk = (double)h[blocks / 2] / iters;
    printf("accs %d mode %d blocks %3d : %7.1f clk per 16 MFMAs (%5.1f per MFMA), wall %.3f ms -> %.2f GHz by s_memtime, %.0f TF/s\n", ACCS, MODE, blocks, clk, clk / 16,
           ms, h[blocks / 2] / (ms * 1e6), blocks * 4.0 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(sink);
}

int main() {
    f32x4* src; hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20);
    for (int blocks : {1, 256}) {
        run<4, 0>(src, blocks); run<2, 0>(src, blocks); run<8, 0>(src, blocks); run<16, 0>(src, blocks);
        run<4, 1>(src, blocks); run<4, 2>(src, blocks); run<4, 3>(src, blocks); run<4, 4>(src, blocks);
        run<8, 4>(src, blocks);
    }
    return 0;
}
