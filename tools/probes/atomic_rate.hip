// How fast are coalesced fp32 atomicAdds to global memory on gfx950?  (round 5: would a one-launch attention backward at N = 1729 afford
// dQ as atomics — 3.5 M elements x 7 key blocks = 25 M adds per layer?)  Each wave adds 64 consecutive floats per instruction; the
// target region (elems floats) is swept `reps` times by different workgroups at different times, like the key blocks would.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_rate.hip -o build/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* dst, long elems, int reps) {
    // grid = (elems / (256 * 16)) * reps blocks: block b sweeps chunk (b % nchunks) — rep r of a chunk is r * nchunks blocks later
    const long nchunks = elems / (256 * 16);
    const long chunk = blockIdx.x % nchunks;
    float* p = dst + chunk * (256 * 16) + (threadIdx.x >> 6) * (64 * 16) + (threadIdx.x & 63);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (MODE == 0) atomicAdd(p + 64 * i, 1.0f);
        else if (MODE == 1) __builtin_amdgcn_global_atomic_fadd_f32(p + 64 * i, 1.0f);      // no-return form
        else p[64 * i] += 1.0f;                                                              // plain read-modify-write (racy: bandwidth yardstick)
    }
}
template <int MODE> void run(float* d, long elems, int reps) {
    const long blocks = elems / (256 * 16) * reps;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(d, 0, elems * 4);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, elems, reps);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, elems, reps);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d elems %8ld reps %d : %8.1f us  (%.1f G adds/s, %.2f TB/s of 4-byte adds)\n", MODE, elems, reps, ms * 1e3, elems * reps / ms / 1e6, elems * reps * 4.0 / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 64 << 20);
    for (long elems : {3541504L, 1105920L}) for (int reps : {1, 7}) { run<0>(d, elems, reps); run<1>(d, elems, reps); run<2>(d, elems, reps); }
    return 0;
}
