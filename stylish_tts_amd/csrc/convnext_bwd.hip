// Backward of the fused GeneratorConvNeXtBlock at C = 32 (conv_next.py:80-93), recompute-based:
//   forward   u = dw7(x), xh = LN(u), xn = (1+g) xh + b, h0 = W1 xn + b1, h = snake(h0; alpha),
//             s = GRN scale(sum_t h^2), y = W2 (h s) + b2eff + x
// The training graph used to keep u, xn, h0, h (4C channels!) and ran ~20 kernels per block over them; at the 75T
// frame rate the nine C = 32 blocks were the largest single cost of a c3 step (~65 of 216 ms).  Here the forward is
// the fused two-pass inference kernel (nothing but x and the per-(b, channel) GRN scale is kept) and the backward
// recomputes the block tile by tile on the matrix cores:
//   pass 1   U = W2^T gY (per tile), ds[b,ch] partial = sum_t U h            (GRN needs it over the whole utterance)
//   -> grn_bwd: coef[b,ch], d gamma
//   pass 2   gH = U s + coef h, gH0 = gH snake'(h0); chained GEMM gXn = W1^T gH0 (accumulator fragment as B operand);
//            LayerNorm backward per column in registers -> gU; writes gU, xn, h s and gH0 (the operands of the two
//            weight-gradient GEMMs, which run on wgrad_k1_kernel) and per-tile partials of d alpha, d(gamma,beta)_AdaLN.
// The depthwise-conv backward runs on the existing kernels from gU.  HBM traffic per time column: ~1 100 floats for
// forward + backward against ~3 300 before.
#include <type_traits>

#include "sty_common.h"

namespace sty {

constexpr int CB_TT = 256;

// BF: bf16 compute mode -- GEMM-1 (h), U = W2^T gY and (pass 2) the chained gXn = W1^T gH0 each become two
// v_mfma_f32_32x32x16_bf16 per 32-row block instead of sixteen v_mfma_f32_32x32x2_f32: operands rounded to bf16 on the
// way in (eight reduction elements per lane: for the chained GEMM the lane's accumulator registers 8 s .. 8 s + 7, rows
// R(hi, 8 s + e), with the A fragment gathered in the same row order), fp32 accumulation, everything else unchanged.
// LEAN (pass 2, bf16 mode): the lean backward -- h s is not written and d alpha is not accumulated here (see the note on
// the lean backward at the end of this file): ten of the ~45 vector instructions per hidden element and a quarter of the
// kernel's output bytes.
template <int PASS, bool BF, bool LEAN = false>
__global__ __launch_bounds__(256, 2) void convnext32_bwd_kernel(Cnx32BwdArgs a) {
  constexpr int LW = CB_TT + 6, LG = CB_TT + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                 // [32][LW]  x tile, then x-hat
  float* gys = xs + 32 * LW;       // [32][LG]  gY tile
  float* rstd_s = gys + 32 * LG;   // [256]
  float* red = rstd_s + 256;       // [4][128]
  float* red2 = red + 4 * 128;     // [4][64]
  float* prm = red2 + 4 * 64;      // [4][128] b1, alpha, GRN scale, coef of this batch row (read as LDS broadcasts:
                                   // as global loads inside the element loops they were 55 % of the wave cycles)
  float* gbs = prm + 5 * 128;      // [64] 1 + gamma | beta of the AdaLN   (prm row 4: 1 / alpha)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  // FUSED (lean pass 2 with a.gx): overlapping tiles -- 256 columns computed, the owned ones [own_lo, own_hi) stored and summed,
  // and the input gradient gX = gY + dwconv^T(gU) written from this kernel (the epilogue at the end): every owned column finds
  // the gU of its three neighbours on either side in this workgroup's LDS
  const bool fused = PASS == 2 && a.gx != nullptr;
  const int b = blockIdx.y, t0 = blockIdx.x * (fused ? CNX_BWD_FUSED_STRIDE : CB_TT), T = a.T;
  const int own_lo = fused && blockIdx.x > 0 ? 4 : 0, own_hi = fused ? 4 + CNX_BWD_FUSED_STRIDE : CB_TT;
  const bool gy16 = a.gy16;  // a two-byte output gradient (kernel-uniform)
  const void* xb = reinterpret_cast<const char*>(a.x) + (size_t)b * 32 * T * 4;
  const void* gb_ = reinterpret_cast<const char*>(a.gy) + (size_t)b * 32 * T * (gy16 ? 2 : 4);
  // pass-2 outputs go through buffer descriptors of this batch row (32-bit offsets: the 64-bit per-row store
  // addresses of the flat form cost ~80 spilled VGPRs)
  __amdgpu_buffer_rsrc_t r_hs, r_g0, r_xn, r_gu, r_gx;
  if (PASS == 2) {
    // (out_bf16: the two 128-channel outputs are bf16 tensors [B][128][T] in the same buffers)
    const int esz = a.out_bf16 ? 2 : 4;
    r_hs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.hs) + (size_t)b * 128 * T * esz, 0, 128 * T * esz,
                                             0x00020000);
    r_g0 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.gh0) + (size_t)b * 128 * T * esz, 0, 128 * T * esz,
                                             0x00020000);
    const int xsz = a.xn16 ? 2 : 4;
    r_xn = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.xn) + (size_t)b * 32 * T * xsz, 0, 32 * T * xsz, 0x00020000);
    const int usz = a.gu16 ? 2 : 4, gsz = a.gx16 ? 2 : 4;
    r_gu = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.gu) + (size_t)b * 32 * T * usz, 0, 32 * T * usz, 0x00020000);
    r_gx = __builtin_amdgcn_make_buffer_rsrc(fused ? reinterpret_cast<char*>(a.gx) + (size_t)b * 32 * T * gsz
                                                   : reinterpret_cast<char*>(a.gu), 0, 32 * T * gsz, 0x00020000);
  }
  auto bst = [](__amdgpu_buffer_rsrc_t rs, float v, int off) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, 0);
  };
  if (tid < 128) {
    prm[tid] = a.b1[tid];
    prm[128 + tid] = a.alpha[tid];
    prm[256 + tid] = a.scale[b * 128 + tid];
    prm[384 + tid] = PASS == 2 ? a.coef[b * 128 + tid] : 0.f;
    prm[512 + tid] = 1.0f / a.alpha[tid];  // (a full-precision division per element and (n, j) block otherwise: ~10 VALU)
  } else if (tid < 192) {
    const int c = tid - 128;
    gbs[c] = c < 32 ? 1.f + a.gb[b * 64 + c] : a.gb[b * 64 + c];
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float v[4][5], g[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 4 * (half * 4 + i);
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int j = lane + 64 * q, t = t0 - 3 + j;
        v[i][q] = (j < LW && t >= 0 && t < T) ? reinterpret_cast<const float*>(xb)[(size_t)row * T + t] : 0.f;
      }
      if (gy16) {  // two-byte gY: a lane loads TWO adjacent columns (one dword; t0 and T are even), half the requests
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int t = t0 + 2 * (lane + 64 * q);
          const unsigned u = t < T ? reinterpret_cast<const unsigned*>(gb_)[((size_t)row * T + t) >> 1] : 0u;
          g[i][2 * q] = sty_bf_lo(u);
          g[i][2 * q + 1] = sty_bf_hi(u);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t0 + lane + 64 * q;
          g[i][q] = t < T ? reinterpret_cast<const float*>(gb_)[(size_t)row * T + t] : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 4 * (half * 4 + i);
#pragma unroll
      for (int q = 0; q < 5; ++q)
        if (lane + 64 * q < LW) xs[row * LW + lane + 64 * q] = v[i][q];
      if (gy16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gys[row * LG + 2 * (lane + 64 * (q >> 1)) + (q & 1)] = g[i][q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) gys[row * LG + lane + 64 * q] = g[i][q];
      }
    }
  }
  __syncthreads();
  {  // depthwise k7 + LayerNorm statistics: one thread per time column
    float u[32];
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      float acc = a.dw_b[c];
#pragma unroll
      for (int k = 0; k < 7; ++k) acc = fmaf(a.dw_w[c * 7 + k], xs[c * LW + tid + k], acc);
      u[c] = acc;
      mean += acc;
    }
    mean *= (1.0f / 32.0f);
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float d = u[c] - mean;
      var += d * d;
    }
    const float rstd = 1.0f / sqrtf(var * (1.0f / 32.0f) + 1e-6f);
    __syncthreads();
    rstd_s[tid] = rstd;
    const int t = t0 + tid;
    const bool own1 = t < T && tid >= own_lo && tid < own_hi;
    if (PASS == 2 && a.xn16) {
      // two-byte xn: a lane pair (columns t, t + 1; t0 and the owned range are multiples of four) shares its values so that
      // every lane stores ONE dword per channel PAIR -- the even lane (c, t .. t + 1), the odd lane (c + 1, t - 1 .. t)
      const bool odd = tid & 1;
      const int voff = own1 ? ((odd ? T + t - 1 : t)) * 2 : 0x7FFFFF00;
#pragma unroll
      for (int c = 0; c < 32; c += 2) {
        const float xh0 = (u[c] - mean) * rstd, xh1 = (u[c + 1] - mean) * rstd;
        xs[c * LW + 3 + tid] = xh0;
        xs[(c + 1) * LW + 3 + tid] = xh1;
        const float v0 = fmaf(xh0, gbs[c], gbs[32 + c]), v1 = fmaf(xh1, gbs[c + 1], gbs[33 + c]);
        const float got = sty_pair_swap(odd ? v0 : v1);  // even: the partner's channel c; odd: the partner's channel c + 1
        const unsigned two = odd ? sty_pack2_bf16(got, v1) : sty_pack2_bf16(v0, got);
        __builtin_amdgcn_raw_buffer_store_b32(two, r_xn, voff, c * T * 2, 0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float xh = (u[c] - mean) * rstd;
        xs[c * LW + 3 + tid] = xh;
        if (PASS == 2 && own1) bst(r_xn, fmaf(xh, gbs[c], gbs[32 + c]), (c * T + t) * 4);
      }
    }
  }
  __syncthreads();
  const int tw = wave * 64;
  // This wave's 64 columns are processed as two independent 32-column passes (n): with both in flight the kernel
  // needed 256 VGPRs + 672 B of scratch per lane; one at a time it fits with room to spare, at the price of reading
  // the (L1-resident, 48 KB) weight fragments twice.
  if (tid < 128) {
    red[tid] = red[128 + tid] = red[256 + tid] = red[384 + tid] = 0.f;
  }
  if (PASS == 2 && tid < 64) red2[tid] = red2[64 + tid] = red2[128 + tid] = red2[192 + tid] = 0.f;
  __syncthreads();
#pragma unroll 1
  for (int n = 0; n < 2; ++n) {
    const int tl = tw + n * 32 + l31, t = t0 + tl;
    const bool okT = t < T;                                // the column exists: it is computed (a neighbour may need its gU)
    const bool ok = okT && tl >= own_lo && tl < own_hi;    // ... and this tile owns it: stored, summed
    // lane part of the bf16 output offsets; columns past the end: outside the descriptor, the store is dropped by the range
    // check (no exec-mask branch around each of the 128 stores per lane and pass)
    const int vst16 = ok ? (4 * hi * T + t) * 2 : 0x7FFFFF00;
    const int vst32 = ok ? (4 * hi * T + t) * 4 : 0x7FFFFF00;
    // paired two-byte stores (lean gH0, two-byte gU): the even lane of a pair stores (its row, columns t, t + 1), the odd lane
    // (the next row, columns t - 1, t); a pair is owned or not as a whole
    const bool oddl = l31 & 1;
    const int vstp = ok ? (4 * hi * T + (oddl ? T + t - 1 : t)) * 2 : 0x7FFFFF00;
    const float okf = okT ? 1.f : 0.f;  // (the hi half of the wave holds the rows 4 further down)
    const float ownf = ok ? 1.f : 0.f;
    // B fragments: normalised input (AdaLN affine applied on the way) and the output gradient
    float bx[16], by[16];
    bf16x8 bxf[2], byf[2];
    if constexpr (BF) {
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        float vx[8], vy[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = 16 * s_ + 8 * hi + e;
          vx[e] = fmaf(xs[c * LW + 3 + tl], gbs[c], gbs[32 + c]);
          vy[e] = gys[c * LG + tl];
        }
        bxf[s_] = sty_pack_bf16(vx[0], vx[1], vx[2], vx[3], vx[4], vx[5], vx[6], vx[7]);
        byf[s_] = sty_pack_bf16(vy[0], vy[1], vy[2], vy[3], vy[4], vy[5], vy[6], vy[7]);
      }
    } else {
#pragma unroll
      for (int c2 = 0; c2 < 16; ++c2) {
        const int c = 2 * c2 + hi;
        bx[c2] = fmaf(xs[c * LW + 3 + tl], gbs[c], gbs[32 + c]);
        by[c2] = gys[c * LG + tl];
      }
    }
    f32x16 gxn;
#pragma unroll
    for (int r = 0; r < 16; ++r) gxn[r] = 0.f;
    // bf16 mode: the A fragments of the three GEMMs come READY MADE from a.wfrag (cnx_frag_pack_kernel: bf16, MFMA lane order,
    // one 16-byte load per fragment; rounds 3-4: sixteen strided dword loads + eight v_cvt_pk per fragment and block), and the
    // six fragments of block j + 1 are REQUESTED IN THE MIDDLE of block j's element loop.  vmcnt counts loads and stores in one
    // in-order queue: a load issued behind the sixteen gH0 stores of an element loop is only "back" when those stores have
    // been acknowledged by the memory system, so with the loads at the top of a block every block began with a full write
    // latency, and the chained GEMM's fragments -- loaded behind the loop -- with another.  Requested after row 7, they wait
    // for eight stores issued half a loop earlier and have the other half of the loop to arrive.  24 registers.
    bf16x8 wf1[2], wf2[2], wf3[2];
    auto load_w = [&](int jj) {
      if constexpr (BF) {
        const bf16x8* fr = reinterpret_cast<const bf16x8*>(a.wfrag) + (size_t)jj * 128 + lane;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          wf1[s_] = fr[s_ * 64];
          wf2[s_] = fr[512 + s_ * 64];
          if (PASS == 2) wf3[s_] = fr[1024 + s_ * 64];
        }
      }
    };
    load_w(0);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      f32x16 h, uu;
#pragma unroll
      for (int r = 0; r < 16; ++r) h[r] = uu[r] = 0.f;
      bf16x8 awf[2];
      if constexpr (BF) {
        // this block's fragments: requested half an element loop ago (load_w), or just now for the first block of a pass
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          awf[s_] = wf3[s_];
          h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[s_], bxf[s_], h, 0, 0, 0);
          uu = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[s_], byf[s_], uu, 0, 0, 0);
        }
      } else {
        const float* w1row = a.w1p + hi * 128 + j * 32 + l31;  // [ci][ch]
        const float* w2row = a.w2 + hi * 128 + j * 32 + l31;   // raw pwconv2.weight [co][ch]
        float av[16], a2[16];
#pragma unroll
        for (int c2 = 0; c2 < 16; ++c2) {
          av[c2] = w1row[(2 * c2) * 128];
          a2[c2] = w2row[(2 * c2) * 128];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < 16; ++c2) {
          h = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c2], bx[c2], h, 0, 0, 0);
          uu = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[c2], by[c2], uu, 0, 0, 0);
        }
      }
      // Snake argument range check once per 32 x 32 block (wave-uniform) instead of a branch per element
      bool slow = false;
      if constexpr (BF) {
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          amax = fmaxf(amax, fabsf(prm[128 + ch] * (h[r] + prm[ch])));
        }
        slow = __any(amax > 8192.0f);
      }
      // The element loop exists twice: the ordinary block (every Snake argument in the hardware sine's range) without the
      // library-sine path in it -- that path is a function call per element, and merely having it in the loop body cost the
      // fast path its schedule (registers saved around the calls, a wait in front of every branch) -- and the rare block with it.
      auto elem_loop = [&](auto slow_c) {
        const bool SLOWP = slow_c;  // (a compile-time constant after inlining, except in the STY_CNX_BWD_OLD build variant)
        float g0e = 0.f;
  #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float bias = prm[ch], al = prm[128 + ch], ral = prm[512 + ch];
          const float sc = prm[256 + ch];
          const float z = h[r] + bias;
          float s2, s2a = 0.f;
          if (PASS == 1) {
            s2 = (BF && !SLOWP) ? sty_sin2_hw(al * z) : sty_sin2(al * z);
          } else {  // sin^2 and sin(2 a z) = 2 sin cos from one range reduction
            float sn, cs;
            if (BF && !SLOWP)
              sty_sincos_hw(al * z, sn, cs);
            else
              sty_sincos(al * z, sn, cs);
            s2 = sn * sn;
            s2a = 2.f * sn * cs;
          }
          const float hv = fmaf(ral, s2, z);
          float rsum = 0.f;
          if (PASS == 1) {
            rsum = ok ? uu[r] * hv : 0.f;
          } else {
            const float cf = prm[384 + ch];
            
#ifdef STY_CNX_BWD_OLD
          const float gH = ok ? fmaf(uu[r], sc, cf * hv) : 0.f;
#else
          const float gH = okf * fmaf(uu[r], sc, cf * hv);  // (a multiply, not a select: hipcc turned `ok ? ... : 0` into an
              // exec-mask branch around the parameter reads of EVERY element, and the basic-block boundary kept the LDS reads of the next
              // rows from being requested early: three exposed LDS round trips per element; columns past the end hold finite values)
#endif
            const float g0 = gH * (1.f + s2a);
            if constexpr (!LEAN) rsum = ownf * gH * (z * s2a - s2 * ral) * ral;  // (d alpha: owned columns only)
            if (LEAN) {
              // gH0 leaves as DWORDS: rows r, r + 1 (r even) are adjacent channels and a lane pair holds adjacent columns, so
              // after one swap the even lane stores (row r: columns t, t + 1) and the odd lane (row r + 1: columns t - 1, t) --
              // eight stores per 32 x 32 block and lane where there were sixteen 2-byte ones
              if ((r & 1) == 0) {
                g0e = g0;
              } else {
                const float got = sty_pair_swap(oddl ? g0e : g0);
                const unsigned two = oddl ? sty_pack2_bf16(got, g0) : sty_pack2_bf16(g0e, got);
                const int srow = (j * 32 + ((r - 1) & 3) + 8 * ((r - 1) >> 2)) * T * 2;
                __builtin_amdgcn_raw_buffer_store_b32(two, r_g0, vstp, srow, 0);
              }
            } else if (ok) {
              if (BF && a.out_bf16) {
                const bf16x8 pk = sty_pack_bf16(hv * sc, g0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
                const unsigned two = __builtin_bit_cast(uint4, pk).x;
                // row offset in the SCALAR offset (wave-uniform part of ch; the hi half of the wave is 4 rows further: in
                // the lane part), so that the per-lane offset is the same for all sixteen rows
                const int srow = (j * 32 + (r & 3) + 8 * (r >> 2)) * T * 2;
                __builtin_amdgcn_raw_buffer_store_b16((short)(two & 0xffffu), r_hs, vst16, srow, 0);
                __builtin_amdgcn_raw_buffer_store_b16((short)(two >> 16), r_g0, vst16, srow, 0);
              } else {
                bst(r_hs, hv * sc, (ch * T + t) * 4);
                bst(r_g0, g0, (ch * T + t) * 4);
              }
            }
            h[r] = g0;
          }
          if constexpr (!(PASS == 2 && LEAN)) {
            rsum = sty_half_sum_to_lane31(rsum);
            if (l31 == 31) atomicAdd(&red[wave * 128 + ch], rsum);  // one writer per slot (the same lane in both passes): a
                                                                     // ds_add_f32 instead of read / add / write
          }
          if (BF && r == 7 && j < 3) load_w(j + 1);
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the per-channel loads / store addresses of later
                                                                // rows from being hoisted (that cost 106 spilled VGPRs)
        }
      };
#ifdef STY_CNX_BWD_OLD
      elem_loop(slow);
#else
      if (slow)
        elem_loop(std::true_type{});
      else
        elem_loop(std::false_type{});
#endif
      if (PASS == 2) {  // gXn[ci][t] += sum_ch W1[ch][ci] gH0[ch][t]: the gH0 fragment is the B operand
        if constexpr (BF) {
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_) {
            const bf16x8 af = awf[s_];
            const bf16x8 bf = sty_pack_bf16(h[8 * s_], h[8 * s_ + 1], h[8 * s_ + 2], h[8 * s_ + 3], h[8 * s_ + 4],
                                            h[8 * s_ + 5], h[8 * s_ + 6], h[8 * s_ + 7]);
            gxn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, gxn, 0, 0, 0);
          }
        } else {
          float aw[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) aw[q] = a.w1[(size_t)(j * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi) * 32 + l31];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 16; ++q) gxn = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[q], h[q], gxn, 0, 0, 0);
        }
      }
    }
    if (PASS == 2) {
      // LayerNorm backward over the 32 channels of this column (fragment rows: 16 registers x 2 half-waves)
      float gxh[16], xh[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2) + 4 * hi;
        xh[r] = xs[c * LW + 3 + tl];
        gxh[r] = gxn[r] * gbs[c];
        s1 += gxh[r];
        s2 = fmaf(gxh[r], xh[r], s2);
      }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      s1 *= (1.0f / 32.0f);
      s2 *= (1.0f / 32.0f);
      const float rs = rstd_s[tl];
      float gue = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float guv = rs * (gxh[r] - s1 - xh[r] * s2);
        if (a.gu16) {
          if ((r & 1) == 0) {
            gue = guv;
          } else {
            const float got = sty_pair_swap(oddl ? gue : guv);
            const unsigned two = oddl ? sty_pack2_bf16(got, guv) : sty_pack2_bf16(gue, got);
            __builtin_amdgcn_raw_buffer_store_b32(two, r_gu, vstp, (((r - 1) & 3) + 8 * ((r - 1) >> 2)) * T * 2, 0);
          }
        } else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, guv), r_gu, vst32, ((r & 3) + 8 * (r >> 2)) * T * 4, 0);
        // FUSED: gU takes x-hat's place in LDS (this lane was the cell's only reader); columns past the end hold zero
        if (fused) xs[c * LW + 3 + tl] = okf * guv;
        float v = ok ? gxn[r] * xh[r] : 0.f, w = ok ? gxn[r] : 0.f;
        v = sty_half_sum_to_lane31(v);
        w = sty_half_sum_to_lane31(w);
        if (l31 == 31) {
          red2[wave * 64 + c] += v;
          red2[wave * 64 + 32 + c] += w;
        }
      }
    }
  }
  __syncthreads();
  if (fused) {
    // gX[c][t] = gY[c][t] + sum_k w[c][k] gU[c][t + 3 - k] (the transpose of u[t] = sum_k w[k] x[t + k - 3]): one thread per
    // column pair; gU of local column j sits in xs cell 3 + j, the cells left of the first
    // tile hold the zeros loaded for t < 0.  Replaces dwconv7_bwd_dx_kernel's pass over gU, gY and gX (0.48 GB per block).
    // A thread takes TWO adjacent columns of sixteen channels: the eight cells both windows live in come as four ds_read_b64
    // (one thread per column and 7 + 1 ds_read_b32 per channel: +51 us per launch; the owned range and T are multiples of four,
    // so a pair is owned or not as a whole), the pair leaves as one 8-byte store.
    typedef unsigned u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
    const int p2 = 2 * (tid & 127), half = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int t = t0 + p2;
    const bool own2 = t < T && p2 >= own_lo && p2 < own_hi;
    const int voff = own2 ? t * 4 : 0x7FFFFF00, voff16 = own2 ? t * 2 : 0x7FFFFF00;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = 16 * half + i;
      const float2* wp = reinterpret_cast<const float2*>(xs + c * LW + p2);  // cells p2 .. p2 + 7 = gU of columns p2 - 3 .. p2 + 4
      const float2 q0 = wp[0], q1 = wp[1], q2 = wp[2], q3 = wp[3];
      const float win[8] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
      float a0 = gys[c * LG + p2], a1 = gys[c * LG + p2 + 1];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float w = a.dw_w[c * 7 + k];
        a0 = fmaf(w, win[6 - k], a0);
        a1 = fmaf(w, win[7 - k], a1);
      }
      if (a.gx16)
        __builtin_amdgcn_raw_buffer_store_b32(sty_pack2_bf16(a0, a1), r_gx, voff16, c * T * 2, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1)}, r_gx,
                                              voff, c * T * 4, 0);
    }
  }
  if (tid < 128 && !(PASS == 2 && LEAN)) {
    const double s = (double)red[tid] + (double)red[128 + tid] + (double)red[256 + tid] + (double)red[384 + tid];
    // pass 1: ds partial; pass 2: d alpha partial
    a.part[((size_t)b * 128 + tid) * a.ntiles + blockIdx.x] = s;
  }
  if (PASS == 2) {
    __syncthreads();
    if (tid < 64)
      a.part_gb[((size_t)b * 64 + tid) * a.ntiles + blockIdx.x] =
          (double)red2[tid] + (double)red2[64 + tid] + (double)red2[128 + tid] + (double)red2[192 + tid];
  }
}

// sum the per-tile partials in a fixed order
//  mode 0: out[b][ch] = sum_tiles part          (ds of the GRN scale)
//  mode 1: out[ch] += sum_b sum_tiles part      (d alpha)
//  mode 2: out[b][ch] += sum_tiles part         (d (gamma | beta) of the AdaLN projection output)
// one wave per output element: lanes stride over the partials (coalesced), fp64 shuffle reduction (fixed order)
__global__ __launch_bounds__(64) void cnx_partial_sum_kernel(const double* __restrict__ part, int B, int C, int ntiles,
                                                             int mode, float* __restrict__ out) {
  const int i = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  if (mode == 1) {
    for (int b = 0; b < B; ++b) {
      const double* p = part + ((size_t)b * C + i) * ntiles;
      for (int k = lane; k < ntiles; k += 64) s += p[k];
    }
  } else {
    const double* p = part + (size_t)i * ntiles;
    for (int k = lane; k < ntiles; k += 64) s += p[k];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    if (mode == 0)
      out[i] = (float)s;
    else
      out[i] += (float)s;
  }
}

int launch_cnx_partial_sum(const double* part, int B, int C, int ntiles, int mode, float* out, hipStream_t st) {
  const int n = mode == 1 ? C : B * C;
  hipLaunchKernelGGL(cnx_partial_sum_kernel, dim3(n), dim3(64), 0, st, part, B, C, ntiles, mode, out);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// bf16 A fragments of a block's three backward GEMMs in the lane order v_mfma_f32_32x32x16_bf16 reads them (lane = (l31, hi), eight
// reduction elements per lane and k-step s):
//   m = 0  h  = W1 xn      element e: w1p[16 s + 8 hi + e][32 j + l31]
//   m = 1  U  = W2^T gY    element e: w2raw[16 s + 8 hi + e][32 j + l31]
//   m = 2  gXn += W1^T gH0 element e: w1raw[32 j + R(8 s + e, hi)][l31],  R(q, hi) = (q & 3) + 8 (q >> 2) + 4 hi (the row order
//          of an accumulator fragment's registers: the chained GEMM takes gH0 straight from them)
// One thread per (m, j, s, lane); rounded to nearest even once, as the kernel's own v_cvt_pk did per block and tile.
__global__ __launch_bounds__(256) void cnx_frag_pack_kernel(const float* __restrict__ w1p, const float* __restrict__ w2raw,
                                                            const float* __restrict__ w1raw, bf16x8* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * 4 * 2 * 64) return;
  const int lane = i & 63, s_ = (i >> 6) & 1, j = (i >> 7) & 3, mtx = i >> 9;
  const int l31 = lane & 31, hi = lane >> 5;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (mtx == 2) {
      const int q = 8 * s_ + e;
      v[e] = w1raw[(size_t)(32 * j + (q & 3) + 8 * (q >> 2) + 4 * hi) * 32 + l31];
    } else {
      v[e] = (mtx == 0 ? w1p : w2raw)[(size_t)(16 * s_ + 8 * hi + e) * 128 + 32 * j + l31];
    }
  }
  out[i] = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}
int launch_cnx_frag_pack(const float* w1p, const float* w2raw, const float* w1raw, void* wfrag, hipStream_t st) {
  hipLaunchKernelGGL(cnx_frag_pack_kernel, dim3(6), dim3(256), 0, st, w1p, w2raw, w1raw, reinterpret_cast<bf16x8*>(wfrag));
  STY_LAUNCH_CHECK();
  return STY_OK;
}

int convnext32_bwd_ntiles(int T, int fused) {
  if (!fused) return cdiv(T, CB_TT);
  const int n = cdiv(T - 4, CNX_BWD_FUSED_STRIDE);  // the last tile owns up to 248 (n - 1) + 252 >= T
  return n < 1 ? 1 : n;
}

int launch_convnext32_bwd(const Cnx32BwdArgs& a, int B, int pass, hipStream_t st) {
  const bool fused_ = pass == 2 && a.gx;
  if (a.gx && (!fused_ || a.gx == a.gy || a.T % 4)) {
    set_error("convnext32_bwd: the fused input gradient needs pass 2, T %% 4 == 0 and gx != gy");
    return STY_EINVAL;
  }
  if (a.x16) {
    set_error("convnext32_bwd: a two-byte x is not built");
    return STY_EINVAL;
  }
  if ((a.gx16 && !fused_) || ((a.gx16 || a.gu16 || a.gy16) && a.T % 2)) {
    set_error("convnext32_bwd: two-byte gX needs the fused input gradient; two-byte gX / gU an even T");
    return STY_EINVAL;
  }
  if (a.xn16 && (pass != 2 || a.T % 2)) {
    set_error("convnext32_bwd: two-byte xn needs pass 2 and an even T");
    return STY_EINVAL;
  }
  if (a.ntiles != convnext32_bwd_ntiles(a.T, fused_)) {
    set_error("convnext32_bwd: ntiles does not match convnext32_bwd_ntiles(T, fused)");
    return STY_EINVAL;
  }
  if (a.bf16 && !a.wfrag) {
    set_error("convnext32_bwd: bf16 mode needs the packed weight fragments (launch_cnx_frag_pack)");
    return STY_EINVAL;
  }
  constexpr size_t lds = (32 * (CB_TT + 6) + 32 * (CB_TT + 1) + 256 + 4 * 128 + 4 * 64 + 5 * 128 + 64) * sizeof(float);
  static bool raised = false;
  if (!raised) {
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext32_bwd_kernel<1, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext32_bwd_kernel<2, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext32_bwd_kernel<1, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext32_bwd_kernel<2, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext32_bwd_kernel<2, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised = true;
  }
  dim3 grid(a.ntiles, B);
  const double pos = (double)B * a.T;
  // per position: dw 448, GEMM-1 8192, U 8192 (+ gXn 8192 in pass 2); pass 1 reads x, gY; pass 2 also writes
  // h s, gH0 (128 each), xn, gU (32 each)
  const double flops = pos * (448.0 + 16384.0 + (pass == 2 ? 8192.0 : 0.0));
  // (h s and gH0 as bf16 in the bf16 mode: 2 x 128 x 2 bytes instead of 2 x 128 x 4)
  const bool lean = pass == 2 && a.bf16 && a.out_bf16 && a.lean;
  const double bytes = pos * ((a.x16 ? 2.0 : 4.0) * 32.0 + (a.gy16 ? 2.0 : 4.0) * 32.0 +
                              (pass == 2 ? (a.xn16 ? 2.0 : 4.0) * 32.0 + (a.gu16 ? 2.0 : 4.0) * 32.0 + (fused_ ? (a.gx16 ? 2.0 : 4.0) * 32.0 : 0.0) +
                                                           (a.out_bf16 ? 2.0 : 4.0) * (lean ? 128.0 : 256.0)
                                                     : 0.0));
  ProfScope prof(pass == 1 ? (a.bf16 ? "convnext32_bwd_kernel<1,true>" : "convnext32_bwd_kernel<1,false>")
                           : (a.bf16 ? "convnext32_bwd_kernel<2,true>" : "convnext32_bwd_kernel<2,false>"),
                 flops, bytes, st);
  if (pass == 1 && a.bf16)
    hipLaunchKernelGGL((convnext32_bwd_kernel<1, true>), grid, dim3(256), lds, st, a);
  else if (pass == 1)
    hipLaunchKernelGGL((convnext32_bwd_kernel<1, false>), grid, dim3(256), lds, st, a);
  else if (lean)
    hipLaunchKernelGGL((convnext32_bwd_kernel<2, true, true>), grid, dim3(256), lds, st, a);
  else if (a.bf16)
    hipLaunchKernelGGL((convnext32_bwd_kernel<2, true>), grid, dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((convnext32_bwd_kernel<2, false>), grid, dim3(256), lds, st, a);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// =====================================================================================================================
// The lean backward (bf16 mode).  Two row sums over the whole utterance used to cost a kernel pass and a tenth of the
// second one; both are diagonals of products the weight-gradient GEMMs compute anyway:
//   ds[b,ch]   = sum_t U h,  U = W2^T gY        = sum_co W2[co,ch] M_b[co,ch],   M_b = gY_b h_b^T  (32 x 128 per utterance)
//                -> pass 1 (GEMM-1 + U + Snake per element) becomes ONE bandwidth-bound GEMM over (gY fp32, h bf16 kept by
//                   the forward: wgrad_cnx_kernel in its per-utterance mode), and the same M gives the weight gradient
//                   dW2[co,ch] = sum_b s[b,ch] M_b[co,ch] -- the separate (h s, gY) GEMM and the h s output of pass 2 go away;
//   d alpha[ch] = sum_t gH (z sin(2 a z) - sin^2(a z) / a) / a  =  (sum_t z gH0 - sum_t gH h) / a,   h = z + sin^2(a z) / a,
//                gH0 = gH (1 + sin(2 a z)), and with z = W1 xn + b1, gH = U s + coef h:
//                sum_t z gH0 = <W1[ch,:], dW1[ch,:]> + b1[ch] db1[ch],     sum_t gH h = s ds + coef sum_t h^2
//                -> no per-element work at all: a 128-thread kernel after the dW1 GEMM.
// The forward pays 256 bytes per position to keep h.  d alpha inherits the bf16 rounding of the dW1 GEMM's operands
// (gH0, xn) instead of being summed from fp32 terms: the same error class as every weight gradient of the mode.
// =====================================================================================================================
__device__ __forceinline__ float cnx_bf16_round(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  return __builtin_bit_cast(float, (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
}
// partial: [B * SB] planes of (128 ch x 32 co + 32 bias sums) floats from wgrad_cnx_kernel<true> in per-utterance mode.
// One workgroup per (hidden channel, utterance): thread = (co, plane group of 8).  M_b[ch][co] = sum of the utterance's SB
// planes (fixed order); ds[b][ch] = sum_co bf(W2[co][ch]) M_b; plane 0 of the utterance is OVERWRITTEN with s[b][ch] M_b (and
// its bias slots with the utterance's bias sums), so that a plain slice reduction over the B utterances gives dW2 and db2.
__global__ __launch_bounds__(256) void cnx_m_finish_kernel(float* __restrict__ partial, int SB, const float* __restrict__ w2raw,
                                                           const float* __restrict__ scale, float* __restrict__ ds) {
  __shared__ float acc_s[8][32];
  const int ch = blockIdx.x, b = blockIdx.y, co = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const size_t stride = 4096 + 32;
  float* base = partial + (size_t)b * SB * stride;
  float mb = 0.f, bs = 0.f;
  for (int s_ = sg; s_ < SB; s_ += 8) {
    mb += base[(size_t)s_ * stride + ch * 32 + co];
    if (ch == 0) bs += base[(size_t)s_ * stride + 4096 + co];
  }
  acc_s[sg][co] = mb;
  __syncthreads();
  if (sg == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += acc_s[k][co];
    float d = cnx_bf16_round(w2raw[(size_t)co * 128 + ch]) * t;  // U = bf(W2)^T bf(gY) in the kernels
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) d += __shfl_xor(d, o);  // over the 32 output channels (one half-wave)
    if (co == 0) ds[(size_t)b * 128 + ch] = d;
    base[ch * 32 + co] = scale[(size_t)b * 128 + ch] * t;
  }
  if (ch == 0) {  // (uniform per workgroup) bias gradient of pwconv2 = sum of gY: the planes' by-product
    __syncthreads();
    acc_s[sg][co] = bs;
    __syncthreads();
    if (sg == 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += acc_s[k][co];
      base[4096 + co] = t;
    }
  }
}
void launch_wgrad_reduce_planes(const float* partial, int nslices, size_t plane, size_t stride, int nb, float* gwp,
                                float* gbias, hipStream_t st);  // wgrad.hip
int launch_cnx_m_finish(float* partial, int B, int SB, const float* w2raw, const float* scale, float* ds, float* gw2,
                        float* gb2, hipStream_t st) {
  hipLaunchKernelGGL(cnx_m_finish_kernel, dim3(128, B), dim3(256), 0, st, partial, SB, w2raw, scale, ds);
  if (gw2) launch_wgrad_reduce_planes(partial, B, 4096, (size_t)SB * (4096 + 32), gb2 ? 32 : 0, gw2, gb2, st);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// gw1: this block's packed dW1 [32 ci][128 ch], gb1 [128] (zero before the block's weight-gradient GEMM: each conv runs once
// per step); part: the forward's GRN partials [B][128][nseg][2] (slot 1 = sum of h^2 of a tile)
__global__ __launch_bounds__(256) void cnx_dalpha_kernel(const float* __restrict__ w1raw, const float* __restrict__ b1,
                                                         const float* __restrict__ alpha, const float* __restrict__ gw1,
                                                         const float* __restrict__ gb1, const float* __restrict__ scale,
                                                         const float* __restrict__ ds, const float* __restrict__ coef,
                                                         const double* __restrict__ part, int nseg, int B,
                                                         float* __restrict__ dalpha) {
  // one workgroup per hidden channel; the B x nseg tile sums of h^2 are spread over the threads (fixed order: deterministic)
  __shared__ double red[4];
  const int ch = blockIdx.x, tid = threadIdx.x;
  double bt = 0.0;
  for (int i = tid; i < B * nseg; i += 256) {
    const int b = i / nseg, k = i - b * nseg;
    bt += (double)coef[(size_t)b * 128 + ch] * part[(((size_t)b * 128 + ch) * nseg + k) * 2 + 1];
  }
  for (int b = tid; b < B; b += 256) bt += (double)scale[(size_t)b * 128 + ch] * (double)ds[(size_t)b * 128 + ch];
  double a = 0.0;
  if (tid < 32) a = (double)cnx_bf16_round(w1raw[(size_t)ch * 32 + tid]) * (double)gw1[tid * 128 + ch];
  if (tid == 32) a = (double)b1[ch] * (double)gb1[ch];
  double v = a - bt;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) dalpha[ch] += (float)((red[0] + red[1] + red[2] + red[3]) / (double)alpha[ch]);
}
int launch_cnx_dalpha(const float* w1raw, const float* b1, const float* alpha, const float* gw1, const float* gb1,
                      const float* scale, const float* ds, const float* coef, const double* part, int nseg, int B,
                      float* dalpha, hipStream_t st) {
  hipLaunchKernelGGL(cnx_dalpha_kernel, dim3(128), dim3(256), 0, st, w1raw, b1, alpha, gw1, gb1, scale, ds, coef, part, nseg,
                     B, dalpha);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
