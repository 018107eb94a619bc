// Token-wise kernels of the cross-attention path (SpatialTransformer / BasicTransformerBlock, UNet:178-342): nn.LayerNorm forward and
// backward over [M tokens][C channels] bf16 rows, and the GEGLU gate of the feed-forward (monai MLPBlock(act="GEGLU"), UNet:211:
// x, gate = chunk(2); x * gelu(gate), exact erf form = F.gelu's default).  HBM-bound streaming kernels: one wave per token row,
// 16-byte pieces on the lanes, fp32 statistics.
#include "common.h"
#include "medimgen_hip.h"

namespace {

// piece p of a row = channels 8p .. 8p+7; lane handles pieces lane, lane + 64 (C <= 1024)
template <int NP>
__global__ void __launch_bounds__(256) k_layernorm_fwd(const bf16* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16* __restrict__ y, int ldy, float* __restrict__ mean_rstd,
                                                       int64_t M, int C, float eps) {
  const int lane = threadIdx.x & 63, C8 = C >> 3;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  F8 v[NP];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = lane + 64 * i;
    if (p < C8) {
      v[i] = unpack8(*(const u32x4*)(x + row * ldx + p * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) { s += v[i].v[j]; q += v[i].v[j] * v[i].v[j]; }
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  const float mean = s / C;
  float var = q / C - mean * mean;
  var = var < 0.f ? 0.f : var;
  const float rstd = rsqrtf(var + eps);
  if (lane == 0) { mean_rstd[2 * row] = mean; mean_rstd[2 * row + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = lane + 64 * i;
    if (p < C8) {
      F8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = (v[i].v[j] - mean) * rstd * gamma[p * 8 + j] + beta[p * 8 + j];
      *(u32x4*)(y + row * ldy + p * 8) = pack8(o);
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dgamma += sum_rows dy * xhat,  dbeta += sum_rows dy.
// Each wave walks `rows_per_wave` rows keeping the per-channel sums of its pieces in registers; the four waves fold through LDS and
// the block adds C x 2 values with fp32 atomics.
template <int NP>
__global__ void __launch_bounds__(256) k_layernorm_bwd(const bf16* __restrict__ dy, int ldd, const bf16* __restrict__ x, int ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean_rstd, bf16* __restrict__ dx,
                                                       int ldo, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t M, int C,
                                                       int rows_per_wave) {
  extern __shared__ float sm[];  // [4][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, C8 = C >> 3;
  float ag[NP][8], ab[NP][8], gm[NP][8];
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ag[i][j] = ab[i][j] = 0.f;
      const int p = lane + 64 * i;
      gm[i][j] = p < C8 ? gamma[p * 8 + j] : 0.f;
    }
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  for (int64_t row = r0; row < r0 + rows_per_wave && row < M; ++row) {
    const float mean = mean_rstd[2 * row], rstd = mean_rstd[2 * row + 1];
    F8 xh[NP], g[NP];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = lane + 64 * i;
      if (p < C8) {
        const F8 xv = unpack8(*(const u32x4*)(x + row * ldx + p * 8)), dv = unpack8(*(const u32x4*)(dy + row * ldd + p * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i].v[j] = (xv.v[j] - mean) * rstd;
          g[i].v[j] = dv.v[j] * gm[i][j];
          m1 += g[i].v[j];
          m2 += g[i].v[j] * xh[i].v[j];
          ag[i][j] += dv.v[j] * xh[i].v[j];
          ab[i][j] += dv.v[j];
        }
      }
    }
    m1 = wave_sum(m1) / C;
    m2 = wave_sum(m2) / C;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = lane + 64 * i;
      if (p < C8) {
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = rstd * (g[i].v[j] - m1 - xh[i].v[j] * m2);
        *(u32x4*)(dx + row * ldo + p * 8) = pack8(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = lane + 64 * i;
    if (p < C8)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[(wave * 2 + 0) * C + p * 8 + j] = ag[i][j];
        sm[(wave * 2 + 1) * C + p * 8 + j] = ab[i][j];
      }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, (sm[0 * C + c] + sm[2 * C + c]) + (sm[4 * C + c] + sm[6 * C + c]));
    atomicAdd(dbeta + c, (sm[1 * C + c] + sm[3 * C + c]) + (sm[5 * C + c] + sm[7 * C + c]));
  }
}

__device__ __forceinline__ float gelu_f(float g) { return 0.5f * g * (1.f + erff(g * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float g) {  // Phi(g) + g * phi(g)
  return 0.5f * (1.f + erff(g * 0.70710678118654752f)) + g * 0.3989422804014327f * __expf(-0.5f * g * g);
}
// h [M][2F] -> y [M][F] = h[:, :F] * gelu(h[:, F:])
__global__ void __launch_bounds__(256) k_geglu_fwd(const bf16* __restrict__ h, bf16* __restrict__ y, int64_t M, int F8n) {
  const int64_t total = M * F8n;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / F8n;
    const int f = (int)(i - m * F8n);
    const F8 a = unpack8(*(const u32x4*)(h + (m * 2 * F8n + f) * 8)), g = unpack8(*(const u32x4*)(h + (m * 2 * F8n + F8n + f) * 8));
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = a.v[j] * gelu_f(g.v[j]);
    *(u32x4*)(y + i * 8) = pack8(o);
  }
}
// dh[:, :F] = dy * gelu(gate);  dh[:, F:] = dy * a * gelu'(gate)
__global__ void __launch_bounds__(256) k_geglu_bwd(const bf16* __restrict__ h, const bf16* __restrict__ dy, bf16* __restrict__ dh, int64_t M,
                                                   int F8n) {
  const int64_t total = M * F8n;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / F8n;
    const int f = (int)(i - m * F8n);
    const F8 a = unpack8(*(const u32x4*)(h + (m * 2 * F8n + f) * 8)), g = unpack8(*(const u32x4*)(h + (m * 2 * F8n + F8n + f) * 8));
    const F8 d = unpack8(*(const u32x4*)(dy + i * 8));
    F8 da, dg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      da.v[j] = d.v[j] * gelu_f(g.v[j]);
      dg.v[j] = d.v[j] * a.v[j] * gelu_grad_f(g.v[j]);
    }
    *(u32x4*)(dh + (m * 2 * F8n + f) * 8) = pack8(da);
    *(u32x4*)(dh + (m * 2 * F8n + F8n + f) * 8) = pack8(dg);
  }
}

}  // namespace

extern "C" {

int mi_layernorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean_rstd, int64_t M, int C,
                     float eps, hipStream_t st) {
  if (!x || !gamma || !beta || !y || !mean_rstd || M <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldy & 7) || ldx < C || ldy < C) return MI_ERR_BAD_ARG;
  if (C > 1024) return MI_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((M + 3) / 4));
  if (C <= 512)
    hipLaunchKernelGGL(k_layernorm_fwd<1>, grid, dim3(256), 0, st, (const bf16*)x, ldx, gamma, beta, (bf16*)y, ldy, mean_rstd, M, C, eps);
  else
    hipLaunchKernelGGL(k_layernorm_fwd<2>, grid, dim3(256), 0, st, (const bf16*)x, ldx, gamma, beta, (bf16*)y, ldy, mean_rstd, M, C, eps);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* mean_rstd, void* dx, int lddx,
                     float* dgamma, float* dbeta, int64_t M, int C, hipStream_t st) {
  if (!dy || !x || !gamma || !mean_rstd || !dx || !dgamma || !dbeta || M <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7))
    return MI_ERR_BAD_ARG;
  if (C > 1024) return MI_ERR_UNSUPPORTED;
  int rpw = (int)((M + 4 * 1024 - 1) / (4 * 1024));  // ~1024 blocks
  if (rpw < 1) rpw = 1;
  const dim3 grid((unsigned)((M + 4ll * rpw - 1) / (4ll * rpw)));
  const size_t lds = sizeof(float) * 8 * (size_t)C;
  if (C <= 512)
    hipLaunchKernelGGL(k_layernorm_bwd<1>, grid, dim3(256), lds, st, (const bf16*)dy, lddy, (const bf16*)x, ldx, gamma, mean_rstd, (bf16*)dx, lddx,
                       dgamma, dbeta, M, C, rpw);
  else
    hipLaunchKernelGGL(k_layernorm_bwd<2>, grid, dim3(256), lds, st, (const bf16*)dy, lddy, (const bf16*)x, ldx, gamma, mean_rstd, (bf16*)dx, lddx,
                       dgamma, dbeta, M, C, rpw);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_geglu_fwd(const void* h, void* y, int64_t M, int F, hipStream_t st) {
  if (!h || !y || M <= 0 || F <= 0 || (F & 7)) return MI_ERR_BAD_ARG;
  const int64_t total = M * (F / 8);
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(k_geglu_fwd, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, st, (const bf16*)h, (bf16*)y, M, F / 8);
  MI_CHECK_LAUNCH();
  return 0;
}
int mi_geglu_bwd(const void* h, const void* dy, void* dh, int64_t M, int F, hipStream_t st) {
  if (!h || !dy || !dh || M <= 0 || F <= 0 || (F & 7)) return MI_ERR_BAD_ARG;
  const int64_t total = M * (F / 8);
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(k_geglu_bwd, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, st, (const bf16*)h, (const bf16*)dy, (bf16*)dh, M, F / 8);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
