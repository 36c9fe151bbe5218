// Does the SOURCE ADDRESS PATTERN of an LDS-DMA piece (64 lanes x 16 B) change what it costs the issuing wave?
//   pattern 0: 1 KB contiguous                      (the dma_rate.hip baseline)
//   pattern 1: 8 rows x 128 B, row stride LD bytes  (a k-contiguous GEMM operand tile), lanes in order
//   pattern 2: the same with the 16-byte chunks of a row XOR-permuted (the bank-conflict swizzle of glds_tiles.hpp)
//   pattern 3: 4 k-lines x 256 B, line stride LD    (a row-contiguous operand, 128-wide tile)
//   pattern 4: 8 k-lines x 128 B permuted           (row-contiguous operand, 64-wide tile)
// One workgroup of 4 waves, 4 pieces per wave and iteration, 3 stages in flight, L2-resident source.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, long ld, int iters, long long* out, float* sink) {
    constexpr int NW = 4, PIECES = 4, DEPTH = 3;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NW * PIECES * DEPTH * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto issue = [&](int t) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int inst = wave * PIECES + p;         // 16 instructions cover a 128-row x 128-B (or 64 x 256-B) tile
            long off;
            if (PATTERN == 0) off = ((long)t * 16 + inst) * 1024 + lane * 16;
            else if (PATTERN == 1 || PATTERN == 2) {
                const int line = inst * 8 + lane / 8, slot = lane % 8;
                const int chunk = PATTERN == 2 ? (slot ^ ((line >> 1) & 7)) : slot;
                off = (long)line * ld + (long)t * 128 + chunk * 16;
            } else if (PATTERN == 3) {
                const int line = inst * 4 + lane / 16, slot = lane % 16;
                const int chunk = slot ^ ((line & 3) << 2);
                off = ((long)t * 64 + line) * ld + chunk * 16;
            } else {
                const int line = inst * 8 + lane / 8, slot = lane % 8;
                const int chunk = slot ^ (((line >> 1) & 1) << 2);
                off = ((long)t * 128 + line) * ld + chunk * 16;
            }
            unsigned char* dst = smem + (((t % DEPTH) * NW + wave) * PIECES + p) * 1024;
            __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    f32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < DEPTH - 1; ++t) issue(t);
    __builtin_amdgcn_s_barrier();
    long long iss = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < iters; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (DEPTH - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        const long long a = __builtin_amdgcn_s_memtime();
        if (t + DEPTH - 1 < iters) issue(t + DEPTH - 1);
        iss += __builtin_amdgcn_s_memtime() - a;
        acc += *reinterpret_cast<f32x4*>(smem + (((t % DEPTH) * NW + wave) * PIECES) * 1024 + lane * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = iss; }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int PATTERN>
void run(const char* name, const char* src, long ld, int blocks, long long* dout, float* sink) {
    const int iters = 10;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<PATTERN>), dim3(blocks), dim3(256), 0, 0, src, ld, iters, dout, sink);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), dout, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> d, e;
    for (int b = 0; b < blocks; ++b) { d.push_back(double(h[2 * b]) / iters); e.push_back(double(h[2 * b + 1]) / iters); }
    std::sort(d.begin(), d.end()); std::sort(e.begin(), e.end());
    printf("%-58s ld %5ld blocks %4d : %7.1f clk/iter, of which issuing the 4 pieces %6.1f  (%5.1f B/clk/CU)\n", name, ld, blocks,
           d[d.size() / 2], e[e.size() / 2], 16384.0 / d[d.size() / 2]);
}

int main() {
    char* src; long long* dout; float* sink;
    hipMalloc(&src, 64L << 20); hipMalloc(&dout, 4096 * 16); hipMalloc(&sink, 64);
    hipMemset(src, 0, 64L << 20);
    for (int blocks : {1, 84, 256}) {
        run<0>("0: 1 KB contiguous per piece", src, 0, blocks, dout, sink);
        for (long ld : {1536L, 6144L, 1024L, 4096L}) {
            run<1>("1: 8 rows x 128 B per piece, lanes in order", src, ld, blocks, dout, sink);
            run<2>("2: 8 rows x 128 B per piece, chunks XOR-permuted", src, ld, blocks, dout, sink);
            run<3>("3: 4 k-lines x 256 B per piece, permuted", src, ld, blocks, dout, sink);
            run<4>("4: 8 k-lines x 128 B per piece, permuted", src, ld, blocks, dout, sink);
        }
    }
    return 0;
}
