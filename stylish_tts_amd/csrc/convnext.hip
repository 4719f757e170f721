// Fused GeneratorConvNeXtBlock for C = 32 channels at the 75T frame rate (conv_next.py:80-93):
//   dwconv k7 -> AdaLN(eps 1e-6) -> Linear 32->128 -> Snake -> GRN(over time) -> Linear 128->32 -> + x
// Nine of these run on [B,32,75T] and carry most of the vocoder's activation traffic.
// GRN needs sum_t h^2 per (b, channel) over the WHOLE utterance, so the block is two launches:
//   pass 1 recomputes h = snake(pw1(adaln(dw(x)))) tile by tile and writes per-tile sum-of-squares;
//   pass 2 recomputes h, scales it by the GRN factor and feeds it STRAIGHT from the MFMA accumulator
//          registers into the second GEMM (the D fragment of GEMM-1 is already a legal B fragment of
//          GEMM-2 when the reduction index is enumerated as (q, hi) -> row (q&3)+8(q>>2)+4hi),
// so x is read once per pass and y written once: the 4C intermediate never touches HBM or even LDS, and pass 2 takes its
// residual from the raw tile it staged (round 6: the normalised tile has a buffer of its own -- bf16 [position][channel] in the
// bf16 mode, fp32 in the fp32 mode -- where it used to overwrite the raw one; the second read of x was 61 of 182 MB per launch
// at c5, profiles/r05_c5-bf16_pmc_traffic.json).  The kernel is csrc/convnext_kernel.h.
#include "convnext_kernel.h"

namespace sty {

void launch_convnext32_bf16(const Cnx32Args& a, dim3 grid, int pass, hipStream_t st);

int convnext32_ntiles(int T) { return cdiv(T, CNX_TT); }

int launch_convnext32(const Cnx32Args& a, int B, int pass, hipStream_t st) {
  if (a.h16 && a.T % 2) {
    set_error("convnext32: h16 (the bf16 copy of h for the lean backward) needs an even T: it is stored in column pairs");
    return STY_EINVAL;
  }
  dim3 grid(a.ntiles, B);
  // per position: dw 2*7*32, pw1 2*32*128, pw2 2*128*32 (pass 2 only); x read once, y written once (pass 2)
  const double pos = (double)B * a.T;
  const double flops = pos * (448.0 + 8192.0 + (pass == 2 ? 8192.0 : 0.0));
  const double bytes = pos * 32 * 4.0 * (pass == 2 ? 2.0 : 1.0);
  ProfScope prof(pass == 1 ? (a.bf16 ? "convnext32_pass1_kernel<true>" : "convnext32_pass1_kernel<false>")
                           : (a.bf16 ? "convnext32_pass2_kernel<true>" : "convnext32_pass2_kernel<false>"),
                 flops, bytes, st);
  if (a.bf16)
    launch_convnext32_bf16(a, grid, pass, st);  // convnext16.hip (its own compiler flags)
  else if (pass == 1)
    hipLaunchKernelGGL((convnext32_kernel<false, false>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((convnext32_kernel<true, false>), grid, dim3(256), 0, st, a);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// pwconv2 [C][4C] -> A fragments of the chained GEMM: w2a[j][q][hi][co] = W2[co][32j + (q&3)+8(q>>2) + 4hi]
// and b2eff[co] = b2[co] + sum_ch W2[co][ch] * grn_beta[ch].
__global__ void pack_w2a_kernel(const float* __restrict__ w2, const float* __restrict__ b2,
                                const float* __restrict__ grn_beta, int C, float* __restrict__ w2a,
                                float* __restrict__ b2eff) {
  const int C4 = 4 * C;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C4 * C) {
    const int co = i % C, rest = i / C;
    const int hi = rest & 1, q = (rest >> 1) & 15, j = rest >> 5;
    const int ch = 32 * j + (q & 3) + 8 * (q >> 2) + 4 * hi;
    w2a[i] = w2[(size_t)co * C4 + ch];
  }
}
// one wave per output channel (coalesced row read + shuffle reduction; the per-thread serial loop this replaces
// took 56 us at C = 256)
__global__ __launch_bounds__(64) void b2eff_kernel(const float* __restrict__ w2, const float* __restrict__ b2,
                                                   const float* __restrict__ grn_beta, int C4,
                                                   float* __restrict__ b2eff) {
  const int co = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f;
  for (int ch = lane; ch < C4; ch += 64) acc = fmaf(w2[(size_t)co * C4 + ch], grn_beta[ch], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) b2eff[co] = b2[co] + acc;
}

// the same two kernels for up to W2A_MAXJ blocks in ONE launch (the job list travels as a kernel argument): sixteen ConvNeXt
// blocks made 32 launches of a few microseconds at the head of every training step's forward
__global__ __launch_bounds__(256) void pack_w2a_multi_kernel(W2aJobs jobs) {
  const int j = blockIdx.y;
  const int C = jobs.C[j], C4 = 4 * C;
  const float* w2 = jobs.w2[j];
  const int npack = (C4 * C + 255) / 256;  // blocks that do the fragment pack; the next C / 4 blocks the b2eff rows
  if ((int)blockIdx.x < npack) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C4 * C) {
      const int co = i % C, rest = i / C;
      const int hi = rest & 1, q = (rest >> 1) & 15, jj = rest >> 5;
      const int ch = 32 * jj + (q & 3) + 8 * (q >> 2) + 4 * hi;
      jobs.w2a[j][i] = w2[(size_t)co * C4 + ch];
    }
    return;
  }
  const int co = ((int)blockIdx.x - npack) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // a wave per output channel
  if (co >= C) return;
  const float* gb = jobs.gb[j];
  float acc = 0.f;
  for (int ch = lane; ch < C4; ch += 64) acc = fmaf(w2[(size_t)co * C4 + ch], gb[ch], acc);  // (b2eff_kernel's order)
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) jobs.b2eff[j][co] = jobs.b2[j][co] + acc;
}
int launch_pack_w2a_multi(const W2aJobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return STY_OK;
  int maxb = 0;
  for (int j = 0; j < jobs.n; ++j) {
    const int C = jobs.C[j], nb = (4 * C * C + 255) / 256 + (C + 3) / 4;
    maxb = nb > maxb ? nb : maxb;
  }
  hipLaunchKernelGGL(pack_w2a_multi_kernel, dim3(maxb, jobs.n), dim3(256), 0, st, jobs);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

int launch_pack_w2a(const float* w2, const float* b2, const float* grn_beta, int C, float* w2a, float* b2eff,
                    hipStream_t st) {
  hipLaunchKernelGGL(pack_w2a_kernel, dim3(cdiv(4 * C * C, 256)), dim3(256), 0, st, w2, b2, grn_beta, C, w2a, b2eff);
  hipLaunchKernelGGL(b2eff_kernel, dim3(C), dim3(64), 0, st, w2, b2, grn_beta, 4 * C, b2eff);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
