// Training-time augmentation of the data path on the GPU: the transform list of
// medimgen/data_processing.py:748-859 (define_nnunet_transformations) applied to patches that already live in HBM.
//
// The reference runs batchgeneratorsv2 transforms on the CPU inside DataLoader workers.  That package is not under /root/reference:
// the arithmetic below restates its published transforms (PARITY UNPINNED; checked against oracle/data.py, which states the same
// arithmetic with torch's own interpolate / grid_sample / conv / pad).  One (sample, channel) PLANE of V = D*H*W fp32 voxels per call
// (2-D patches: D = 1); every kernel is one streaming pass, HBM-bound: 4 B read + 4 B written per voxel (stats: 4 B read).
// Per-plane statistics stay on the device (`stats` = {min, max, mean, unbiased std}): the kernels that need them take the pointer, so a
// transform chain never synchronises with the host.
#include "common.h"
#include "medimgen_hip.h"

namespace {

constexpr int kT = 256;
constexpr int kStatBlocks = 256;  // partial records per plane (workspace: kStatBlocks * 4 floats)
constexpr int kMaxTaps = 33;

inline int grid_for(int64_t total, int cap = 4096) {
  int64_t g = (total + kT - 1) / kT;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- plane statistics: min, max, sum, sum of squares per block, finished in fp64 by one block
__global__ void __launch_bounds__(kT) k_plane_partial(const float* __restrict__ x, int64_t V, float* __restrict__ part) {
  __shared__ float red[4][4];
  float mn = INFINITY, mx = -INFINITY, s = 0.f, q = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < V; i += (int64_t)gridDim.x * kT) {
    const float v = x[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    float* r = red[threadIdx.x >> 6];
    r[0] = mn; r[1] = mx; r[2] = s; r[3] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + 4 * blockIdx.x;
    o[0] = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
    o[1] = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
    o[2] = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    o[3] = (red[0][3] + red[1][3]) + (red[2][3] + red[3][3]);
  }
}
__global__ void __launch_bounds__(64) k_plane_finish(const float* __restrict__ part, int nblocks, int64_t V, float* __restrict__ stats) {
  float mn = INFINITY, mx = -INFINITY;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) {
    mn = fminf(mn, part[4 * i]);
    mx = fmaxf(mx, part[4 * i + 1]);
    s += (double)part[4 * i + 2];
    q += (double)part[4 * i + 3];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if (threadIdx.x == 0) {
    const double mean = s / (double)V;
    double var = V > 1 ? (q - s * mean) / (double)(V - 1) : 0.0;  // torch.Tensor.std(): Bessel's correction
    if (var < 0.0) var = 0.0;
    stats[0] = mn; stats[1] = mx; stats[2] = (float)mean; stats[3] = (float)sqrt(var);
  }
}

// ---- pointwise ops, in place.  a / b: stats records (device), aux: a second plane (noise)
template <int OP>
__global__ void __launch_bounds__(kT) k_pointwise(float* __restrict__ x, int64_t V, float p0, const float* __restrict__ a,
                                                  const float* __restrict__ b, const float* __restrict__ aux) {
  float mn = 0.f, mx = 0.f, mean = 0.f, sd = 0.f, mean_b = 0.f, sd_b = 0.f;
  if (OP == MI_AUG_CONTRAST || OP == MI_AUG_GAMMA || OP == MI_AUG_RESTORE_STATS) { mn = a[0]; mx = a[1]; mean = a[2]; sd = a[3]; }
  if (OP == MI_AUG_RESTORE_STATS) { mean_b = b[2]; sd_b = b[3]; }
  const float rng = mx - mn;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < V; i += (int64_t)gridDim.x * kT) {
    float v = x[i];
    if (OP == MI_AUG_SCALE) v *= p0;                                                       // MultiplicativeBrightnessTransform
    else if (OP == MI_AUG_CONTRAST) v = fminf(fmaxf((v - mean) * p0 + mean, mn), mx);     // ContrastTransform(preserve_range=True)
    else if (OP == MI_AUG_GAMMA) v = powf((v - mn) / fmaxf(rng, 1e-7f), p0) * rng + mn;    // GammaTransform, no inversion
    else if (OP == MI_AUG_RESTORE_STATS) v = (v - mean) * (sd_b / fmaxf(sd, 1e-7f)) + mean_b;  // its retain_stats tail: a = now, b = before
    else if (OP == MI_AUG_ADD_NOISE) v += p0 * aux[i];                                     // GaussianNoiseTransform (p0 = sigma)
    else if (OP == MI_AUG_CLAMP01) v = fminf(fmaxf(v, 0.f), 1.f);                          // DATA:595
    x[i] = v;
  }
}

// ---- GaussianBlurTransform: one axis of the separable filter, reflect padding (torch pad mode "reflect": the edge voxel is not repeated)
struct Taps {
  float w[kMaxTaps];
  int n;
};
__global__ void __launch_bounds__(kT) k_blur_axis(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W, int axis, Taps t) {
  const int64_t V = (int64_t)D * H * W;
  const int n = axis == 0 ? D : (axis == 1 ? H : W);
  const int64_t stride = axis == 0 ? (int64_t)H * W : (axis == 1 ? W : 1);
  const int r = t.n / 2;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < V; i += (int64_t)gridDim.x * kT) {
    const int p = (int)((i / stride) % n);
    const float* base = x + (i - p * stride);
    float acc = 0.f;
    for (int j = 0; j < t.n; ++j) {
      int q = p + j - r;
      q = q < 0 ? -q : (q >= n ? 2 * (n - 1) - q : q);
      acc += t.w[j] * base[q * stride];
    }
    y[i] = acc;
  }
}

// ---- SimulateLowResolutionTransform: interpolate(mode="nearest-exact") to (lD, lH, lW), then interpolate(mode="trilinear" /
// "bilinear", align_corners=False) back -- as ONE gather: the coarse grid is never materialised (a coarse voxel IS a source voxel)
__device__ __forceinline__ void lin_src(int o, int n_out, int n_low, int n_in, int& i0, int& i1, float& w1) {
  // coarse coordinate of fine voxel o (area_pixel_compute_source_index, align_corners=False, clamped at 0)
  float s = ((float)n_low / (float)n_out) * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  int l0 = (int)s;
  l0 = l0 > n_low - 1 ? n_low - 1 : l0;
  const int l1 = l0 + 1 < n_low ? l0 + 1 : l0;
  w1 = s - (float)l0;
  // nearest-exact: coarse index l reads source voxel floor((l + 0.5) * n_in / n_low)
  const float sc = (float)n_in / (float)n_low;
  i0 = (int)floorf(((float)l0 + 0.5f) * sc);
  i1 = (int)floorf(((float)l1 + 0.5f) * sc);
  i0 = i0 > n_in - 1 ? n_in - 1 : i0;
  i1 = i1 > n_in - 1 ? n_in - 1 : i1;
}
__global__ void __launch_bounds__(kT) k_lowres(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W, int lD, int lH, int lW) {
  const int64_t V = (int64_t)D * H * W;
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < V; i += (int64_t)gridDim.x * kT) {
    const int xw = (int)(i % W);
    const int64_t t = i / W;
    const int yh = (int)(t % H), zd = (int)(t / H);
    int z0, z1, y0, y1, x0, x1;
    float wz, wy, wx;
    lin_src(zd, D, lD, D, z0, z1, wz);
    lin_src(yh, H, lH, H, y0, y1, wy);
    lin_src(xw, W, lW, W, x0, x1, wx);
    auto at = [&](int a, int b, int c) { return x[((int64_t)a * H + b) * W + c]; };
    // accumulation order of upsample_trilinear3d: w-pairs, then h, then d
    const float c00 = (1.f - wx) * at(z0, y0, x0) + wx * at(z0, y0, x1), c01 = (1.f - wx) * at(z0, y1, x0) + wx * at(z0, y1, x1);
    const float c10 = (1.f - wx) * at(z1, y0, x0) + wx * at(z1, y0, x1), c11 = (1.f - wx) * at(z1, y1, x0) + wx * at(z1, y1, x1);
    y[i] = (1.f - wz) * ((1.f - wy) * c00 + wy * c01) + wz * ((1.f - wy) * c10 + wy * c11);
  }
}

// ---- SpatialTransform (rotation / scaling, no elastic deformation): y[o] = trilinear(x, A (o - c_out) + c_in), zeros outside
// (grid_sample(mode="bilinear", padding_mode="zeros", align_corners=False) on the centred grid); c = (extent - 1) / 2 per axis
struct Affine {
  float a[9];
};
__global__ void __launch_bounds__(kT) k_affine_sample(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W, int oD, int oH,
                                                      int oW, Affine m) {
  const int64_t V = (int64_t)oD * oH * oW;
  const float ci0 = 0.5f * (D - 1), ci1 = 0.5f * (H - 1), ci2 = 0.5f * (W - 1);
  const float co0 = 0.5f * (oD - 1), co1 = 0.5f * (oH - 1), co2 = 0.5f * (oW - 1);
  for (int64_t i = blockIdx.x * (int64_t)kT + threadIdx.x; i < V; i += (int64_t)gridDim.x * kT) {
    const int xw = (int)(i % oW);
    const int64_t t = i / oW;
    const float g0 = (float)(t / oH) - co0, g1 = (float)(t % oH) - co1, g2 = (float)xw - co2;
    const float s0 = m.a[0] * g0 + m.a[1] * g1 + m.a[2] * g2 + ci0;
    const float s1 = m.a[3] * g0 + m.a[4] * g1 + m.a[5] * g2 + ci1;
    const float s2 = m.a[6] * g0 + m.a[7] * g1 + m.a[8] * g2 + ci2;
    const float f0 = floorf(s0), f1 = floorf(s1), f2 = floorf(s2);
    const int z0 = (int)f0, y0 = (int)f1, x0 = (int)f2;
    const float wz = s0 - f0, wy = s1 - f1, wx = s2 - f2;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = z0 + (k >> 2), yy = y0 + ((k >> 1) & 1), xx = x0 + (k & 1);
      const float w = ((k >> 2) ? wz : 1.f - wz) * (((k >> 1) & 1) ? wy : 1.f - wy) * ((k & 1) ? wx : 1.f - wx);
      if ((unsigned)z < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += w * x[((int64_t)z * H + yy) * W + xx];
    }
    y[i] = acc;
  }
}

}  // namespace

extern "C" {

int64_t mi_aug_stats_workspace_bytes(void) { return (int64_t)kStatBlocks * 4 * sizeof(float); }

int mi_aug_plane_stats(const float* x, int64_t V, float* stats, float* workspace, hipStream_t st) {
  if (!x || !stats || !workspace || V <= 0) return MI_ERR_BAD_ARG;
  const int nb = grid_for(V, kStatBlocks);
  hipLaunchKernelGGL(k_plane_partial, dim3(nb), dim3(kT), 0, st, x, V, workspace);
  hipLaunchKernelGGL(k_plane_finish, dim3(1), dim3(64), 0, st, (const float*)workspace, nb, V, stats);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_aug_pointwise(float* x, int64_t V, int op, float p0, const float* stats_a, const float* stats_b, const float* aux, hipStream_t st) {
  if (!x || V <= 0) return MI_ERR_BAD_ARG;
  const bool need_a = op == MI_AUG_CONTRAST || op == MI_AUG_GAMMA || op == MI_AUG_RESTORE_STATS;
  if ((need_a && !stats_a) || (op == MI_AUG_RESTORE_STATS && !stats_b) || (op == MI_AUG_ADD_NOISE && !aux)) return MI_ERR_BAD_ARG;
  const dim3 g(grid_for(V)), b(kT);
  switch (op) {
    case MI_AUG_SCALE: hipLaunchKernelGGL(k_pointwise<MI_AUG_SCALE>, g, b, 0, st, x, V, p0, stats_a, stats_b, aux); break;
    case MI_AUG_CONTRAST: hipLaunchKernelGGL(k_pointwise<MI_AUG_CONTRAST>, g, b, 0, st, x, V, p0, stats_a, stats_b, aux); break;
    case MI_AUG_GAMMA: hipLaunchKernelGGL(k_pointwise<MI_AUG_GAMMA>, g, b, 0, st, x, V, p0, stats_a, stats_b, aux); break;
    case MI_AUG_RESTORE_STATS: hipLaunchKernelGGL(k_pointwise<MI_AUG_RESTORE_STATS>, g, b, 0, st, x, V, p0, stats_a, stats_b, aux); break;
    case MI_AUG_ADD_NOISE: hipLaunchKernelGGL(k_pointwise<MI_AUG_ADD_NOISE>, g, b, 0, st, x, V, p0, stats_a, stats_b, aux); break;
    case MI_AUG_CLAMP01: hipLaunchKernelGGL(k_pointwise<MI_AUG_CLAMP01>, g, b, 0, st, x, V, p0, stats_a, stats_b, aux); break;
    default: return MI_ERR_BAD_ARG;
  }
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_aug_blur_axis(const float* x, float* y, int D, int H, int W, int axis, const float* taps_host, int ntaps, hipStream_t st) {
  if (!x || !y || x == y || D <= 0 || H <= 0 || W <= 0 || axis < 0 || axis > 2 || !taps_host) return MI_ERR_BAD_ARG;
  const int n = axis == 0 ? D : (axis == 1 ? H : W);
  if (ntaps < 1 || ntaps > kMaxTaps || !(ntaps & 1) || ntaps / 2 >= n) return MI_ERR_BAD_ARG;  // reflect padding needs radius < extent
  Taps t;
  t.n = ntaps;
  for (int j = 0; j < ntaps; ++j) t.w[j] = taps_host[j];
  hipLaunchKernelGGL(k_blur_axis, dim3(grid_for((int64_t)D * H * W)), dim3(kT), 0, st, x, y, D, H, W, axis, t);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_aug_lowres(const float* x, float* y, int D, int H, int W, int lD, int lH, int lW, hipStream_t st) {
  if (!x || !y || x == y || D <= 0 || H <= 0 || W <= 0 || lD <= 0 || lH <= 0 || lW <= 0) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_lowres, dim3(grid_for((int64_t)D * H * W)), dim3(kT), 0, st, x, y, D, H, W, lD, lH, lW);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_aug_affine_sample(const float* x, float* y, int D, int H, int W, int oD, int oH, int oW, const float* a_host, hipStream_t st) {
  if (!x || !y || x == y || D <= 0 || H <= 0 || W <= 0 || oD <= 0 || oH <= 0 || oW <= 0 || !a_host) return MI_ERR_BAD_ARG;
  Affine m;
  for (int j = 0; j < 9; ++j) m.a[j] = a_host[j];
  hipLaunchKernelGGL(k_affine_sample, dim3(grid_for((int64_t)oD * oH * oW)), dim3(kT), 0, st, x, y, D, H, W, oD, oH, oW, m);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
