// Probe: what does ONE wave per SIMD sustain on v_mfma_f32_32x32x2_f32 / 32x32x16_bf16, with 2 or 4 accumulators, with
// and without LDS reads between the MFMAs?  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int WAVES, bool LDS, bool BF>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, int iters) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64 * WAVES) sm[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float a = threadIdx.x * 1e-3f;
  const float* p = sm + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float b[NACC];
#pragma unroll
      for (int n = 0; n < NACC; ++n) b[n] = LDS ? p[((it + u) & 31) * 128 + n * 32] : a + (float)(u + n);
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if constexpr (BF) {
          bf16x8 av, bv;
          for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b[n]; }
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[n], 0, 0, 0);
        } else {
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[n], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int NACC, int WAVES, bool LDS, bool BF>
void run(const char* name, float* out) {
  const int iters = 2000, grid = 256 * (WAVES >= 4 ? 1 : 4 / WAVES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NACC, WAVES, LDS, BF>), dim3(grid), dim3(64 * WAVES), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NACC, WAVES, LDS, BF>), dim3(grid), dim3(64 * WAVES), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)grid * WAVES * iters * 16.0 * NACC;
  const double flop = mf * (BF ? 32768.0 : 4096.0);
  printf("%-44s %8.3f ms  %7.1f TF  (%.1f cycles/MFMA/wave at 2.4 GHz)\n", name, ms, flop / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * 16.0 * NACC));
}

int main() {
  float* out;
  hipMalloc(&out, 1 << 24);
  run<2, 4, false, false>("f32 32x32x2, 2 acc, 4 waves/CU, no LDS", out);
  run<4, 4, false, false>("f32 32x32x2, 4 acc, 4 waves/CU, no LDS", out);
  run<2, 4, true, false>("f32 32x32x2, 2 acc, 4 waves/CU, LDS reads", out);
  run<4, 4, true, false>("f32 32x32x2, 4 acc, 4 waves/CU, LDS reads", out);
  run<2, 8, false, false>("f32 32x32x2, 2 acc, 8 waves/CU, no LDS", out);
  run<4, 8, true, false>("f32 32x32x2, 4 acc, 8 waves/CU, LDS reads", out);
  run<2, 4, false, true>("bf16 32x32x16, 2 acc, 4 waves/CU, no LDS", out);
  run<4, 4, false, true>("bf16 32x32x16, 4 acc, 4 waves/CU, no LDS", out);
  run<4, 8, false, true>("bf16 32x32x16, 4 acc, 8 waves/CU, no LDS", out);
  return 0;
}
