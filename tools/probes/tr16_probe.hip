#include <hip/hip_runtime.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + threadIdx.x * 4));
  out[threadIdx.x * 4 + 0] = v[0]; out[threadIdx.x * 4 + 1] = v[1]; out[threadIdx.x * 4 + 2] = v[2]; out[threadIdx.x * 4 + 3] = v[3];
}
int main() {
  short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
}
