// What does ONE dependent kernel node of a replayed HIP graph cost when the kernel itself does nothing?  (round 6)
// A chain of N kernel nodes on one stream, captured and replayed; per node: grid x block, bytes of kernel arguments, and what the
// kernel does before it exits:
//   MODE 0 nothing   1 one dependent global load per thread (a memory round trip)   2 an LDS-using workgroup of 64 KB (slot allocation)
// Prints us per node = (replay time) / N.  The batch-4 training step is ~330 dependent nodes of 5-20 us each: this is the floor
// under every one of them.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/chain_rate.hip -o /tmp/chain_rate && /tmp/chain_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Big { long long v[40]; };   // 320 bytes of kernel arguments (two GArgs descriptors are ~400)

template <int MODE>
__global__ void node(const float* __restrict__ src, float* __restrict__ dst, int n) {
    if (MODE == 1) {
        const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
        const float v = src[i];
        if (v == 123456.f) dst[i] = v;
    }
    if (MODE == 2) {
        __shared__ unsigned char smem[64 * 1024];
        smem[threadIdx.x] = (unsigned char)threadIdx.x;
        __syncthreads();
        if (smem[(threadIdx.x + 1) & 255] == 77 && src[0] == 123456.f) dst[0] = 1.f;
    }
}

template <int MODE>
__global__ void node_big(const Big b, const float* __restrict__ src, float* __restrict__ dst, int n) {
    if (b.v[3] == 987654321LL) dst[0] = 2.f;
    if (MODE == 1) {
        const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
        const float v = src[i];
        if (v == 123456.f) dst[i] = v;
    }
}

template <class F>
static float time_chain(F launch, int nodes, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < nodes; ++i) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a, st); hipGraphLaunch(ge, st); hipEventRecord(b, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1e3f / nodes;
}

int main() {
    hipStream_t st; hipStreamCreate(&st);
    const int n = 1 << 22;
    float *src, *dst; hipMalloc(&src, n * 4); hipMalloc(&dst, n * 4); hipMemset(src, 0, n * 4);
    const int nodes = 300;
    Big big{};
    struct Cfg { int grid, block; } cfgs[] = {{1, 64}, {256, 256}, {110, 256}, {512, 512}, {912, 512}, {2048, 256}};
    for (auto c : cfgs) {
        const float t0 = time_chain([&](hipStream_t s) { hipLaunchKernelGGL(node<0>, dim3(c.grid), dim3(c.block), 0, s, src, dst, n); }, nodes, st);
        const float t1 = time_chain([&](hipStream_t s) { hipLaunchKernelGGL(node<1>, dim3(c.grid), dim3(c.block), 0, s, src, dst, n); }, nodes, st);
        const float t2 = time_chain([&](hipStream_t s) { hipLaunchKernelGGL(node<2>, dim3(c.grid), dim3(c.block), 0, s, src, dst, n); }, nodes, st);
        const float t3 = time_chain([&](hipStream_t s) { hipLaunchKernelGGL(node_big<0>, dim3(c.grid), dim3(c.block), 0, s, big, src, dst, n); }, nodes, st);
        const float t4 = time_chain([&](hipStream_t s) { hipLaunchKernelGGL(node_big<1>, dim3(c.grid), dim3(c.block), 0, s, big, src, dst, n); }, nodes, st);
        printf("grid %5d x %3d: empty %.2f us/node | one load %.2f | 64 KB LDS + barrier %.2f | 320 B kernarg %.2f | kernarg + load %.2f\n",
               c.grid, c.block, t0, t1, t2, t3, t4);
    }
    // the same chain issued eagerly (no graph): what the graph saves
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(node<0>, dim3(256), dim3(256), 0, st, src, dst, n);
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(node<0>, dim3(256), dim3(256), 0, st, src, dst, n);
        hipEventRecord(b, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("eager chain 256 x 256 empty: %.2f us/launch\n", ms * 1e3f / nodes);
    }
    return 0;
}
