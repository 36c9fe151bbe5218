// Feed-rate probe: how fast can ONE workgroup (NW waves) move L2-resident bytes into LDS on gfx950,
//   mode 0: global_load_lds (LDS-DMA, 16 B/lane), D stages in flight, counted vmcnt
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging), D stages in flight
// Each iteration moves PIECES x 1 KB per wave.  Prints shader clocks per iteration and bytes/clk/CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_rate.hip -o /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NW, int PIECES, int DEPTH, int MODE>
__global__ __launch_bounds__(64 * NW) void probe(const f32x4* __restrict__ src, long n16, int iters, long long* out, float* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NW * PIECES * DEPTH * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long base = ((long)blockIdx.x * 7919 * 64) % (n16 - (long)iters * NW * PIECES * 64 - 64);
    auto issue = [&](int t) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const long idx = base + ((long)t * NW * PIECES + wave * PIECES + p) * 64 + lane;
            unsigned char* dst = smem + (((t % DEPTH) * NW + wave) * PIECES + p) * 1024;
            if (MODE == 0) __builtin_amdgcn_global_load_lds(src + idx, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    f32x4 acc = {0, 0, 0, 0};
    long long t0 = 0, t1 = 0;
    if (MODE == 0) {
        for (int t = 0; t < DEPTH - 1; ++t) issue(t);
        __builtin_amdgcn_s_barrier();
        t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < iters; ++t) {
            if (DEPTH >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (DEPTH - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            if (t + DEPTH - 1 < iters) issue(t + DEPTH - 1);
            // consume a little so the compiler keeps the LDS
            acc += *reinterpret_cast<f32x4*>(smem + (((t % DEPTH) * NW + wave) * PIECES) * 1024 + lane * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    } else {
        f32x4 r[DEPTH][PIECES];
        auto load = [&](int t, f32x4 (&dst)[PIECES]) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) dst[p] = src[base + ((long)t * NW * PIECES + wave * PIECES + p) * 64 + lane];
        };
#pragma unroll
        for (int t = 0; t < DEPTH - 1; ++t) load(t, r[t]);
        __builtin_amdgcn_s_barrier();
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int t0i = 0; t0i < iters; t0i += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int t = t0i + u;
                load(t + DEPTH - 1, r[(u + DEPTH - 1) % DEPTH]);
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    *reinterpret_cast<f32x4*>(smem + (((u % DEPTH) * NW + wave) * PIECES + p) * 1024 + lane * 16) = r[u][p];
                __builtin_amdgcn_s_barrier();
                acc += *reinterpret_cast<f32x4*>(smem + (((u % DEPTH) * NW + wave) * PIECES) * 1024 + ((lane + 1) & 63) * 16);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t0; out[blockIdx.x * 2 + 1] = t1; }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int NW, int PIECES, int DEPTH, int MODE>
void run(const f32x4* src, long n16, int blocks, long long* dout, float* sink) {
    const int iters = 64;
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL((probe<NW, PIECES, DEPTH, MODE>), dim3(blocks), dim3(64 * NW), 0, 0, src, n16, iters, dout, sink);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), dout, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> d;
    for (int b = 0; b < blocks; ++b) d.push_back(double(h[2 * b + 1] - h[2 * b]) / iters);
    std::sort(d.begin(), d.end());
    const double med = d[d.size() / 2], kb = NW * PIECES;
    printf("mode %d  waves %d  pieces/wave %d  depth %d  blocks %4d : %7.1f clk/iter (max %7.1f)  %5.1f B/clk/CU  (%d KB per iter)\n", MODE, NW, PIECES,
           DEPTH, blocks, med, d.back(), kb * 1024 / med, (int)kb);
}

int main() {
    const long n16 = (1L << 20) / 16 * 4;   // 4 MB: fits one XCD's L2
    f32x4* src; long long* dout; float* sink;
    hipMalloc(&src, n16 * 16 + (1 << 22)); hipMalloc(&dout, 4096 * 16); hipMalloc(&sink, 64);
    hipMemset(src, 0, n16 * 16 + (1 << 22));
    for (int blocks : {1, 256}) {
        run<4, 2, 3, 0>(src, n16, blocks, dout, sink);
        run<4, 4, 3, 0>(src, n16, blocks, dout, sink);
        run<4, 6, 3, 0>(src, n16, blocks, dout, sink);
        run<4, 6, 4, 0>(src, n16, blocks, dout, sink);
        run<4, 8, 3, 0>(src, n16, blocks, dout, sink);
        run<8, 2, 3, 0>(src, n16, blocks, dout, sink);
        run<8, 3, 3, 0>(src, n16, blocks, dout, sink);
        run<8, 4, 3, 0>(src, n16, blocks, dout, sink);
        run<16, 2, 3, 0>(src, n16, blocks, dout, sink);
        run<4, 4, 2, 1>(src, n16, blocks, dout, sink);
        run<4, 6, 2, 1>(src, n16, blocks, dout, sink);
        run<4, 4, 4, 1>(src, n16, blocks, dout, sink);
        run<8, 4, 2, 1>(src, n16, blocks, dout, sink);
        run<8, 2, 4, 1>(src, n16, blocks, dout, sink);
    }
    return 0;
}
