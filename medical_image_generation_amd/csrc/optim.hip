// Fused optimizer over the flat fp32 parameter arena: global grad-norm (clip_grad_norm_, T-LDM:177) + Adam / AdamW
// (torch.optim defaults, T-LDM:121, T-AE:470, T-DDPM:383) in two launches instead of ~8 elementwise ops x 332 tensors.
// HBM-bound: 4 arrays read + 3 written, 16-byte accesses.  Step count and clip norm live in device memory so the
// whole train step can be captured in a hipGraph and replayed.
#include "common.h"
#include "medimgen_hip.h"

namespace {

__global__ void __launch_bounds__(256) k_sumsq(const f32x4* __restrict__ x, int64_t n4, const float* __restrict__ tail, int ntail,
                                               float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = x[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) acc += tail[threadIdx.x] * tail[threadIdx.x];
  float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale;
  int decoupled;
  const float* sumsq;  // may be null: no clipping
  const float* step;   // device scalar holding the 1-based step count (already incremented)
};

__global__ void __launch_bounds__(256) k_adam(AdamArgs a) {
  const float t = *a.step;
  const float bc1 = 1.f - powf(a.beta1, t), bc2 = 1.f - powf(a.beta2, t);
  const float step_size = a.lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  // grad_scale: the gradient buffer holds grad_scale^-1 times the gradient (a SUM all-reduce over `world` ranks with
  // grad_scale = 1/world is the data-parallel mean); the norm of the true gradient is grad_scale * sqrt(sumsq)
  float clip = a.grad_scale;
  if (a.sumsq) {  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    float c = a.max_norm / (a.grad_scale * sqrtf(*a.sumsq) + 1e-6f);
    clip = (c < 1.f ? c : 1.f) * a.grad_scale;
  }
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
    float p = a.p[i], g = a.g[i] * clip, m = a.m[i], v = a.v[i];
    if (a.decoupled)
      p *= 1.f - a.lr * a.weight_decay;
    else
      g += a.weight_decay * p;
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    p -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
    a.p[i] = p;
    a.m[i] = m;
    a.v[i] = v;
  }
}

__global__ void k_step_inc(float* step) { *step += 1.f; }
__global__ void __launch_bounds__(256) k_scale(float* __restrict__ x, float alpha, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= alpha;
}
// y += alpha * x (gradient accumulation over micro-batches: T-LDM:173-180 sums the micro-step gradients)
__global__ void __launch_bounds__(256) k_axpy(f32x4* __restrict__ y, const f32x4* __restrict__ x, float alpha, int64_t n4, float* __restrict__ yt,
                                              const float* __restrict__ xt, int ntail) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 a = y[i], b = x[i];
    a[0] += alpha * b[0]; a[1] += alpha * b[1]; a[2] += alpha * b[2]; a[3] += alpha * b[3];
    y[i] = a;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) yt[threadIdx.x] += alpha * xt[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_scale_by_clip(float* g, int64_t n, const float* sumsq, float max_norm) {
  float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
  c = c < 1.f ? c : 1.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] *= c;
}

}  // namespace

extern "C" {

int mi_sumsq_f32(const float* x, int64_t n, float* out, int accumulate, hipStream_t st) {
  if (n <= 0 || ((uintptr_t)x & 15)) return MI_ERR_BAD_ARG;
  if (!accumulate) {
    hipError_t e = mi_zero_fill_f32(out, 1, 1, 1, st);
    if (e != hipSuccess) return (int)e;
  }
  int64_t n4 = n / 4;
  int grid = (int)((n4 + 255) / 256);
  grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
  hipLaunchKernelGGL(k_sumsq, dim3(grid), dim3(256), 0, st, (const f32x4*)x, n4, x + n4 * 4, (int)(n - n4 * 4), out);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int decoupled_weight_decay, const float* grad_sumsq, float max_norm, float grad_scale,
                 float* step_counter, hipStream_t st) {
  if (n <= 0 || !step_counter || !(grad_scale > 0.f)) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, st, step_counter);
  AdamArgs a{param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, decoupled_weight_decay,
             grad_sumsq, step_counter};
  int grid = (int)((n + 255) / 256);
  grid = grid > 4096 ? 4096 : grid;
  hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_scale_f32(float* x, float alpha, int64_t n, hipStream_t st) {
  if (n <= 0 || !x) return MI_ERR_BAD_ARG;
  int grid = (int)((n + 255) / 256);
  hipLaunchKernelGGL(k_scale, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, st, x, alpha, n);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_axpy_f32(float* y, const float* x, float alpha, int64_t n, hipStream_t st) {
  if (n <= 0 || !y || !x || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return MI_ERR_BAD_ARG;
  int64_t n4 = n / 4;
  int grid = (int)((n4 + 255) / 256);
  grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
  hipLaunchKernelGGL(k_axpy, dim3(grid), dim3(256), 0, st, (f32x4*)y, (const f32x4*)x, alpha, n4, y + n4 * 4, x + n4 * 4, (int)(n - n4 * 4));
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_clip_grad_by_norm(float* grad, int64_t n, const float* grad_sumsq, float max_norm, hipStream_t st) {
  if (n <= 0) return MI_ERR_BAD_ARG;
  int grid = (int)((n + 255) / 256);
  grid = grid > 4096 ? 4096 : grid;
  hipLaunchKernelGGL(k_scale_by_clip, dim3(grid), dim3(256), 0, st, grad, n, grad_sumsq, max_norm);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
