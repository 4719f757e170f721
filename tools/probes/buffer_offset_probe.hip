// Probe for the flat-image staging finding of DESIGN.md 4.9: a raw buffer load whose VGPR offset is NEGATIVE (a row shift
// in front of the start of a channel row) combined with a positive instruction immediate (the column group, `offset:256`).
// Every lane loads element (lane + 64 q - shift) of a 1024-float row filled with 1 + index; elements in front of the row
// (negative index) must read as 0 through the descriptor's range check, everything else as its value.
//   form A: one VGPR offset per group, immediate folded by the compiler (what conv_stage.h did for the flat mode)
//   form B: the whole offset kept in the VGPR (an empty asm stops the folding: what it does now)
// Prints, per form, the lanes whose in-range elements came back wrong.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/buffer_offset_probe.hip -o /tmp/buffer_offset_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <bool KEEP>
__global__ void probe(const float* row, int n, int shift, float* out) {
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, n * 4, 0x00020000);
  const int vrow = (lane - shift) * 4;  // negative for lane < shift
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int off = vrow + 256 * q;
    if (KEEP) asm volatile("" : "+v"(off));
    out[q * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
  }
}

int main() {
  const int n = 1024;
  float h[n], *d, *o, r[256];
  for (int i = 0; i < n; ++i) h[i] = 1.f + i;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shift : {58, 70, 122, 130}) {
    for (int keep = 0; keep < 2; ++keep) {
      if (keep)
        hipLaunchKernelGGL(probe<true>, dim3(1), dim3(64), 0, 0, d, n, shift, o);
      else
        hipLaunchKernelGGL(probe<false>, dim3(1), dim3(64), 0, 0, d, n, shift, o);
      hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
      int bad = 0;
      printf("shift %3d  %s:", shift, keep ? "offset kept in the VGPR      " : "column group in the immediate");
      for (int j = 0; j < 256; ++j) {
        const int idx = j - shift;
        const float want = idx < 0 ? 0.f : 1.f + idx;
        if (r[j] != want) {
          if (bad < 8) printf("  element %d (lane %d, group %d) = %g, expected %g;", idx, j & 63, j >> 6, r[j], want);
          ++bad;
        }
      }
      printf(bad ? "  [%d wrong]\n" : "  all correct\n", bad);
    }
  }
  return 0;
}
