#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // natural addressing: lane l reads 8 bytes at element 4*l
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * threadIdx.x));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
}
