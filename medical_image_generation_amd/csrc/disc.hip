// Kernels of the PatchDiscriminator path (train_autoencoder.py:371-397, 416-423, 600: `generative.networks.nets.PatchDiscriminator` --
// k4 convs with stride 2 / 1 and padding 1, BatchNorm, LeakyReLU(0.2) -- and `PatchAdversarialLoss("least_squares")`).
//
// The discriminator is a small side network of the autoencoder's GAN step (5 convs, ~3 ms of a ~25 ms step), so its convolutions are
// lowered to the library's bf16 NT GEMM instead of getting conv kernels of their own:
//     patches[voxel][tap * C + c] = x[voxel * s + tap - p][c]         (mi_im2col3d, zero outside the tensor)
//     y[voxel][co] = patches . W2[co][:] + bias                        (mi_gemm_nt_bf16; W2 = the torch weight with the taps made the slow K axis)
//     dpatches = dy . W2^T^T (NT against W2 transposed),  dx = fold of dpatches (mi_col2im3d: a gather, no atomics)
//     dW2 = dy^T . patches (NT on the two transposed matrices), un-permuted into the torch layout by mi_disc_wgrad_unpack
// BatchNorm in training mode over (N, D, H, W) of an NDHWC tensor IS GroupNorm with one channel per group on the [1][N*V][C] view:
// the GroupNorm kernels (groupnorm.hip, activation code 2 = LeakyReLU(0.2)) serve it; mi_bn_running_update keeps the module's buffers.
#include "common.h"
#include "medimgen_hip.h"

namespace {

constexpr int kT = 256;

// one thread per (output voxel, tap, 8-channel piece); C % 8 != 0 (the 1-channel image): one thread per (voxel, tap), scalar copies
__global__ void __launch_bounds__(kT) k_im2col3d(const bf16* __restrict__ x, int xcs, bf16* __restrict__ out, int N, int D, int H, int W, int C,
                                                 int Do, int Ho, int Wo, int k, int s, int p, int64_t total) {
  const int T = k * k * k, C8 = (C + 7) / 8;
  const bool vec = (C & 7) == 0 && (xcs & 7) == 0;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int c8 = (int)(i % C8);
    int64_t r = i / C8;
    const int t = (int)(r % T); r /= T;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do);
    const int n = (int)(r / Do);
    const int td = t / (k * k), th = (t / k) % k, tw = t % k;
    const int id = od * s + td - p, ih = oh * s + th - p, iw = ow * s + tw - p;
    const bool in = (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const int64_t vox = ((int64_t)(n * Do + od) * Ho + oh) * Wo + ow;
    bf16* dst = out + (vox * T + t) * C + c8 * 8;
    const bf16* src = x + (((int64_t)(n * D + id) * H + ih) * W + iw) * xcs + c8 * 8;
    if (vec) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (in) v = *(const u32x4*)src;
      *(u32x4*)dst = v;
    } else {
      for (int j = 0; j < 8 && c8 * 8 + j < C; ++j) dst[j] = in ? src[j] : f2bf(0.f);
    }
  }
}

// dx[voxel][c] = sum over the (output voxel, tap) pairs that read it of dpatches[..][tap * C + c]: a gather, one thread per (input voxel,
// 8-channel piece); C % 8 != 0: scalar.
__global__ void __launch_bounds__(kT) k_col2im3d(const bf16* __restrict__ dp, bf16* __restrict__ dx, int dcs, int N, int D, int H, int W, int C, int Do,
                                                 int Ho, int Wo, int k, int s, int p, int64_t total) {
  const int T = k * k * k, C8 = (C + 7) / 8;
  const bool vec = (C & 7) == 0 && (dcs & 7) == 0;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int c8 = (int)(i % C8);
    int64_t r = i / C8;
    const int iw = (int)(r % W); r /= W;
    const int ih = (int)(r % H); r /= H;
    const int id = (int)(r % D);
    const int n = (int)(r / D);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int td = 0; td < k; ++td) {
      const int nd = id + p - td;
      if (nd < 0 || nd % s) continue;
      const int od = nd / s;
      if (od >= Do) continue;
      for (int th = 0; th < k; ++th) {
        const int nh = ih + p - th;
        if (nh < 0 || nh % s) continue;
        const int oh = nh / s;
        if (oh >= Ho) continue;
        for (int tw = 0; tw < k; ++tw) {
          const int nw = iw + p - tw;
          if (nw < 0 || nw % s) continue;
          const int ow = nw / s;
          if (ow >= Wo) continue;
          const int t = (td * k + th) * k + tw;
          const bf16* src = dp + ((((int64_t)(n * Do + od) * Ho + oh) * Wo + ow) * T + t) * C + c8 * 8;
          if (vec) {
            const F8 f = unpack8(*(const u32x4*)src);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f.v[j];
          } else {
            for (int j = 0; j < 8 && c8 * 8 + j < C; ++j) acc[j] += bf2f(src[j]);
          }
        }
      }
    }
    bf16* dst = dx + (((int64_t)(n * D + id) * H + ih) * W + iw) * dcs + c8 * 8;
    if (vec) {
      F8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
      *(u32x4*)dst = pack8(o);
    } else {
      for (int j = 0; j < 8 && c8 * 8 + j < C; ++j) dst[j] = f2bf(acc[j]);
    }
  }
}

// torch weight [Co][Ci][T] fp32 -> w2 [Cop][T * Ci] bf16 (rows co >= Co zero) and w2t [T * Ci][Cop] bf16
__global__ void __launch_bounds__(kT) k_disc_pack(const float* __restrict__ w, bf16* __restrict__ w2, bf16* __restrict__ w2t, int Co, int Cop, int Ci,
                                                  int T) {
  const int64_t total = (int64_t)Cop * T * Ci;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int ci = (int)(i % Ci);
    int64_t r = i / Ci;
    const int t = (int)(r % T);
    const int co = (int)(r / T);
    const bf16 v = f2bf(co < Co ? w[((int64_t)co * Ci + ci) * T + t] : 0.f);
    w2[i] = v;
    w2t[((int64_t)t * Ci + ci) * Cop + co] = v;
  }
}
// dw[Co][Ci][T] += dw2[Cop][T * Ci]
__global__ void __launch_bounds__(kT) k_disc_wgrad_unpack(const float* __restrict__ dw2, float* __restrict__ dw, int Co, int Ci, int T) {
  const int64_t total = (int64_t)Co * Ci * T;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int t = (int)(i % T);
    int64_t r = i / T;
    const int ci = (int)(r % Ci);
    const int co = (int)(r / Ci);
    dw[i] += dw2[((int64_t)co * T + t) * Ci + ci];
  }
}

// LeakyReLU(slope) on bf16, 8 elements per thread; bwd: dx = dy * (x > 0 ? 1 : slope)
__global__ void __launch_bounds__(kT) k_leaky_fwd(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t n8, float slope) {
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kT) {
    F8 f = unpack8(x[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = f.v[j] > 0.f ? f.v[j] : slope * f.v[j];
    y[i] = pack8(f);
  }
}
__global__ void __launch_bounds__(kT) k_leaky_bwd(const u32x4* __restrict__ x, const u32x4* __restrict__ dy, u32x4* __restrict__ dx, int64_t n8, float slope) {
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kT) {
    const F8 f = unpack8(x[i]);
    F8 g = unpack8(dy[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) g.v[j] = f.v[j] > 0.f ? g.v[j] : slope * g.v[j];
    dx[i] = pack8(g);
  }
}

// nn.BatchNorm's buffers after a training-mode forward: running = (1 - m) running + m batch, the variance unbiased (x count / (count - 1));
// mean_rstd: [C][2] of the statistics pass (rstd = 1 / sqrt(var_biased + eps))
__global__ void __launch_bounds__(kT) k_bn_running(const float* __restrict__ mean_rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                                   int64_t* __restrict__ nbt, int C, float eps, float momentum, double count) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1];
  const double var = 1.0 / ((double)rstd * rstd) - (double)eps;
  const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
  rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
  rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
}

// PatchAdversarialLoss(criterion="least_squares") on the logits channel 0 of a [nvox][cs] tensor: a = LeakyReLU(slope)(logit) (upstream
// applies LeakyReLU(0.05) in front of the MSE unless no_activation_leastsq; slope 1 = no activation), *loss += weight * mean((a - target)^2),
// dlogits[v][0] = weight * 2 (a - target) / nvox * a'(logit), the other cs - 1 (padding) channels of dlogits = 0.
__global__ void __launch_bounds__(kT) k_ls_gan_loss(const bf16* __restrict__ logits, int cs, int64_t nvox, float target, float slope,
                                                    bf16* __restrict__ dlogits, float* __restrict__ loss, float weight) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv = 1.f / (float)nvox;
  for (int64_t v = blockIdx.x * (int64_t)kT + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * kT) {
    const float l = bf2f(logits[v * cs]);
    const float a = l > 0.f ? l : slope * l;
    const float d = a - target;
    acc += d * d;
    if (dlogits) {
      dlogits[v * cs] = f2bf(weight * 2.f * d * inv * (l > 0.f ? 1.f : slope));
      for (int c = 1; c < cs; ++c) dlogits[v * cs + c] = f2bf(0.f);
    }
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, weight * s * inv);
}

inline int grid_for(int64_t total) {
  int64_t g = (total + kT - 1) / kT;
  return (int)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int mi_im2col3d(const void* x, int x_cs, void* patches, int N, int D, int H, int W, int C, int k, int s, int p, hipStream_t st) {
  if (!x || !patches || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0 || p < 0 || x_cs < C) return MI_ERR_BAD_ARG;
  const int Do = (D + 2 * p - k) / s + 1, Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  if (Do <= 0 || Ho <= 0 || Wo <= 0) return MI_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * Do * Ho * Wo * k * k * k * ((C + 7) / 8);
  hipLaunchKernelGGL(k_im2col3d, dim3(grid_for(total)), dim3(kT), 0, st, (const bf16*)x, x_cs, (bf16*)patches, N, D, H, W, C, Do, Ho, Wo, k, s, p, total);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_col2im3d(const void* dpatches, void* dx, int dx_cs, int N, int D, int H, int W, int C, int k, int s, int p, hipStream_t st) {
  if (!dpatches || !dx || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0 || p < 0 || dx_cs < C) return MI_ERR_BAD_ARG;
  const int Do = (D + 2 * p - k) / s + 1, Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  if (Do <= 0 || Ho <= 0 || Wo <= 0) return MI_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * D * H * W * ((C + 7) / 8);
  hipLaunchKernelGGL(k_col2im3d, dim3(grid_for(total)), dim3(kT), 0, st, (const bf16*)dpatches, (bf16*)dx, dx_cs, N, D, H, W, C, Do, Ho, Wo, k, s, p, total);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_disc_pack_weights(const float* w, void* w2, void* w2t, int Cout, int Cout_padded, int Cin, int taps, hipStream_t st) {
  if (!w || !w2 || !w2t || Cout <= 0 || Cout_padded < Cout || Cin <= 0 || taps <= 0) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_disc_pack, dim3(grid_for((int64_t)Cout_padded * taps * Cin)), dim3(kT), 0, st, w, (bf16*)w2, (bf16*)w2t, Cout, Cout_padded, Cin, taps);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_disc_wgrad_unpack(const float* dw2, float* dw, int Cout, int Cin, int taps, hipStream_t st) {
  if (!dw2 || !dw || Cout <= 0 || Cin <= 0 || taps <= 0) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_disc_wgrad_unpack, dim3(grid_for((int64_t)Cout * Cin * taps)), dim3(kT), 0, st, dw2, dw, Cout, Cin, taps);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_leaky_relu_fwd(const void* x, void* y, int64_t n, float slope, hipStream_t st) {
  if (!x || !y || n <= 0 || (n & 7)) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_leaky_fwd, dim3(grid_for(n / 8)), dim3(kT), 0, st, (const u32x4*)x, (u32x4*)y, n / 8, slope);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_leaky_relu_bwd(const void* x, const void* dy, void* dx, int64_t n, float slope, hipStream_t st) {
  if (!x || !dy || !dx || n <= 0 || (n & 7)) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_leaky_bwd, dim3(grid_for(n / 8)), dim3(kT), 0, st, (const u32x4*)x, (const u32x4*)dy, (u32x4*)dx, n / 8, slope);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_ls_gan_loss(const void* logits, int cs, int64_t nvox, float target, float act_slope, void* dlogits, float* loss, float weight, hipStream_t st) {
  if (!logits || cs <= 0 || nvox <= 0) return MI_ERR_BAD_ARG;
  int64_t g = (nvox + kT - 1) / kT;
  hipLaunchKernelGGL(k_ls_gan_loss, dim3((int)(g > 1024 ? 1024 : g)), dim3(kT), 0, st, (const bf16*)logits, cs, nvox, target, act_slope, (bf16*)dlogits, loss,
                     weight);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_bn_running_update(const float* mean_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int C, float eps,
                         float momentum, int64_t count, hipStream_t st) {
  if (!mean_rstd || !running_mean || !running_var || C <= 0 || count <= 0) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_bn_running, dim3((C + kT - 1) / kT), dim3(kT), 0, st, mean_rstd, running_mean, running_var, num_batches_tracked, C, eps, momentum,
                     (double)count);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
