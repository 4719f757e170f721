// AdamW over a flat fp32 bucket (parameters, gradients and both moments are contiguous and equally laid out), the
// arithmetic order of torch.optim.AdamW's single-tensor path (torch/optim/adamw.py -> adam.py _single_tensor_adam):
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2)*g*g;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// as used by the reference (train/optimizers.py:110-118: AdamW, eps 1e-9, betas (0.85, 0.99), weight decay 1e-4).
// One pass: 4 reads + 3 writes per element, HBM-bound (28 B/element).
#include "sty_common.h"

namespace sty {

// lr_mult (optional, DEVICE): the learning rate of this launch is lr * *lr_mult -- the discriminators' rate multiplier
// (optimizers.py:54-65) computed on the device by disc_lr_track_kernel, so that the host never reads a loss back; the
// two scalars that depend on the rate are then formed here, in the host path's arithmetic (lr_d, wd, bc1 given).
__global__ __launch_bounds__(256) void adamw_kernel(size_t n4, size_t n, float4* __restrict__ p,
                                                    const float4* __restrict__ g, float4* __restrict__ m,
                                                    float4* __restrict__ v, float decay, float w1, float b2, float w2,
                                                    float step_size, float bc2s, float eps, float gscale,
                                                    const double* __restrict__ lr_mult, double lr_d, float wd, double bc1) {
  if (lr_mult) {
    const float lr = (float)(lr_d * lr_mult[0]);
    decay = 1.0f - lr * wd;
    step_size = (float)((double)lr / bc1);
  }
  const size_t stride = (size_t)gridDim.x * 256;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= gscale;  // 1 / world_size after a SUM all-reduce (1.0f is exact: single-GPU results are unchanged)
    pp *= decay;
    mm = mm + w1 * (gg - mm);
    vv = vv * b2 + w2 * gg * gg;
    const float denom = sqrtf(vv) / bc2s + eps;
    pp = pp - step_size * (mm / denom);
  };
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pv = p[i], mv = m[i], vv = v[i];
    const float4 gv = g[i];
    upd(pv.x, gv.x, mv.x, vv.x);
    upd(pv.y, gv.y, mv.y, vv.y);
    upd(pv.z, gv.z, mv.z, vv.z);
    upd(pv.w, gv.w, mv.w, vv.w);
    p[i] = pv;
    m[i] = mv;
    v[i] = vv;
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    float* ps = reinterpret_cast<float*>(p);
    const float* gs = reinterpret_cast<const float*>(g);
    float* ms = reinterpret_cast<float*>(m);
    float* vs = reinterpret_cast<float*>(v);
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) upd(ps[i], gs[i], ms[i], vs[i]);
  }
}

// EMA of a tracked discriminator loss and the learning-rate multiplier it sets (train/losses.py:236-256, :287;
// DiscriminatorLossHelper.get_disc_lr_multiplier): state[0] = last_loss, state[1] = the multiplier of last_loss BEFORE
// this step's value is folded in (the order of train/stage.py: the optimizer steps with the multiplier of the previous
// EMA, then the helper tracks the new loss).  Double precision, as the reference's Python floats.
__global__ void disc_lr_track_kernel(double* state, const float* loss, double ideal, double f_max, double h_min,
                                     double x_max, double x_min) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double last = state[0];
  const double x = fabs(last - ideal);
  double mult;
  if (last > ideal + x_max)
    mult = f_max;
  else if (last < ideal - x_min)
    mult = h_min;
  else if (last > ideal)
    mult = fmin(pow(f_max, x / x_max), f_max);
  else
    mult = fmax(pow(h_min, x / x_min), h_min);
  state[1] = mult;
  if (loss) state[0] = last * 0.95 + (double)loss[0] * 0.05;
}

int launch_adamw(size_t n, float* p, const float* g, float* m, float* v, double lr_d, float beta1, float beta2, float eps,
                 float weight_decay, int step, float grad_scale, const double* lr_mult, hipStream_t st) {
  const float lr = (float)lr_d;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2s = (float)sqrt(bc2);
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  ProfScope prof("adamw_kernel", 12.0 * n, 28.0 * n, st);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n4, n, reinterpret_cast<float4*>(p),
                     reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v),
                     1.0f - lr * weight_decay, 1.0f - beta1, beta2, 1.0f - beta2, step_size, bc2s, eps, grad_scale,
                     lr_mult, lr_d, weight_decay, bc1);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty

extern "C" int sty_adamw_step(size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream) {
  using namespace sty;
  if (!p || !g || !m || !v || step < 1 || ((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) {
    set_error("sty_adamw_step: null / unaligned (16 B) buffer or step < 1");
    return STY_EINVAL;
  }
  if (n == 0) return STY_OK;
  return launch_adamw(n, p, g, m, v, (double)lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr,
                      reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sty_adamw_step_scaled(size_t n, float* p, const float* g, float* m, float* v, double lr,
                                     const double* lr_mult, float beta1, float beta2, float eps, float weight_decay,
                                     int step, float grad_scale, void* stream) {
  using namespace sty;
  if (!p || !g || !m || !v || !lr_mult || step < 1 ||
      ((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) {
    set_error("sty_adamw_step_scaled: null / unaligned (16 B) buffer, no multiplier or step < 1");
    return STY_EINVAL;
  }
  if (n == 0) return STY_OK;
  return launch_adamw(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale, lr_mult,
                      reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sty_disc_lr_track(double* state, const float* loss, double ideal_loss, double f_max, double h_min,
                                 double x_max, double x_min, void* stream) {
  using namespace sty;
  if (!state || x_max <= 0.0 || x_min <= 0.0) {
    set_error("sty_disc_lr_track: null state or empty band");
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(disc_lr_track_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), state, loss,
                     ideal_loss, f_max, h_min, x_max, x_min);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
