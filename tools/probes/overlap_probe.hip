// Probe: do the MFMAs of one wave per SIMD overlap with VALU / LDS / global-load work of three other waves on the same
// SIMD (the consumer / producer split of conv32p / convp16)?  One 16-wave workgroup per CU; waves 0-3 run `mi` groups of
// 64 MFMAs (4 accumulators), waves 4-15 run `wi` iterations of the chosen filler.  Times: MFMA waves alone, fillers alone,
// both.  hipcc --offload-arch=gfx950 -O3 tools/probes/overlap_probe.hip -o /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF, int FILL>  // FILL 0: VALU fma chain, 1: LDS writes + reads, 2: global loads
__global__ __launch_bounds__(1024) void probe(float* out, const float* in, int mi, int wi) {
  __shared__ float sm[16384];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  if (wave < 4) {
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n)
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const float a = lane * 1e-3f;
    for (int it = 0; it < mi; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (BF) {
            bf16x8 av, bv;
            for (int e = 0; e < 8; ++e) {
              av[e] = (__bf16)a;
              bv[e] = (__bf16)(a + u);
            }
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[n], 0, 0, 0);
          } else {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a + u, acc[n], 0, 0, 0);
          }
        }
    }
    for (int n = 0; n < 4; ++n)
      for (int r = 0; r < 16; ++r) s += acc[n][r];
  } else {
    float x = lane * 0.5f, y = 1.0001f;
    for (int it = 0; it < wi; ++it) {
      if (FILL == 0) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x = fmaf(x, y, 0.25f);
      } else if (FILL == 1) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          sm[(wave * 1024 + u * 64 + lane) & 16383] = x;
          x += sm[(wave * 1024 + ((u + 5) & 15) * 64 + lane) & 16383];
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) x += in[((size_t)(blockIdx.x * 16 + wave) * 4096 + (it & 7) * 512 + u * 64 + lane) & 0xFFFFF];
      }
    }
    s = x;
  }
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = s;
}

template <bool BF, int FILL>
void run(const char* name, float* out, const float* in, int mi, int wi) {
  auto t = [&](int a, int b) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<BF, FILL>), dim3(256), dim3(1024), 0, 0, out, in, a ? 1 : 0, b ? 1 : 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<BF, FILL>), dim3(256), dim3(1024), 0, 0, out, in, a, b);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
  };
  const float m = t(mi, 0), f = t(0, wi), both = t(mi, wi);
  printf("%-34s MFMA alone %7.3f ms, filler alone %7.3f ms, both %7.3f ms  (sum %.3f, max %.3f)\n", name, m, f, both, m + f,
         m > f ? m : f);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&in, 4 << 20);
  hipMemset(in, 0, 4 << 20);
  run<true, 0>("bf16 MFMA + VALU", out, in, 2000, 3000);
  run<true, 1>("bf16 MFMA + LDS", out, in, 2000, 3000);
  run<true, 2>("bf16 MFMA + global loads", out, in, 2000, 6000);
  run<false, 0>("fp32 MFMA + VALU", out, in, 1000, 3000);
  run<false, 1>("fp32 MFMA + LDS", out, in, 1000, 3000);
  run<false, 2>("fp32 MFMA + global loads", out, in, 1000, 6000);
  return 0;
}
