// GroupNorm (+SiLU) statistics / apply / backward on NDHWC bf16 activations (UNet:628,648,377,1932; AEKL:157,167,451,604).
//
// HBM-bound.  Statistics are two-level: per-(n, V-chunk, channel) fp32 partial sums written to a workspace by a
// streaming kernel (16-byte loads, 8 channels per lane, LDS reduction over the voxel lanes of the block), then one
// small block per (n, group) finishes in fp64 and emits the per-(n, channel) affine pair
//     scale = gamma * rstd,   shift = beta - mean * scale
// that conv prologues / the apply kernel consume, so "normalise" is a single FMA per element downstream.
// Backward uses the same two-level shape: partial sums of du and du*x, a per-(n, group) finish producing the three
// coefficients of  dx = a*du + b*x + c  per (n, channel), and a streaming apply.
#include <stdlib.h>

#include "common.h"
#include "medimgen_hip.h"

namespace {

constexpr int kT = 256;

// activation behind the affine: 0 none, 1 SiLU (UNet / AEKL), 2 LeakyReLU(0.2) (the PatchDiscriminator's BatchNorm layers)
template <int ACT>
__device__ __forceinline__ float act_f(float u) {
  if constexpr (ACT == 1) return silu_f(u);
  else if constexpr (ACT == 2) return u > 0.f ? u : 0.2f * u;
  else return u;
}
template <int ACT>
__device__ __forceinline__ float act_grad_f(float u) {
  if constexpr (ACT == 1) return silu_grad_f(u);
  else if constexpr (ACT == 2) return u > 0.f ? 1.f : 0.2f;
  else return 1.f;
}

struct Geo {
  int C8;    // channel groups of 8
  int rows;  // voxel lanes per block = kT / C8
};

__device__ __forceinline__ void load_ss(const float* ss, int c0, float* sc, float* sh) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = ss[2 * (c0 + j)];
    sh[j] = ss[2 * (c0 + j) + 1];
  }
}

// Block tail: 16 per-thread partials (8 channels x 2 moments) -> out[c][2].  Lanes that own the same 8 channels sit C8
// apart in a wave, so when C8 is a power of two the voxel lanes are folded with xor-shuffles (log2(64/C8) steps) and LDS only
// sees one row per wave; otherwise every row goes through LDS.
// out: this image's [C][chunks][2] slab -- chunk-fastest, so the finish kernels read each channel's partials as one contiguous run
// sums64 (mi_gn_bwd_fused): instead of the chunk's slot of `out`, the block's totals are added to the image's [C][2] fp64 sums with one
// hardware atomic per value (the consumer needs no per-chunk fold: no finalize launch).
__device__ __forceinline__ void block_fold(float (&a)[8], float (&b)[8], float* sm, float* out, int chunks, int chunk, int C, int C8, int rows,
                                           int r, int cg, double* sums64 = nullptr) {
  const bool pow2 = (C8 & (C8 - 1)) == 0 && C8 <= 64;
  if (pow2) {
    for (int m = C8; m < 64; m <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] += __shfl_xor(a[j], m, 64);
        b[j] += __shfl_xor(b[j], m, 64);
      }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < C8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[(wave * C + lane * 8 + j) * 2] = a[j];
        sm[(wave * C + lane * 8 + j) * 2 + 1] = b[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += kT) {
      const float v = (sm[i] + sm[2 * C + i]) + (sm[4 * C + i] + sm[6 * C + i]);
      if (sums64) unsafeAtomicAdd(sums64 + i, (double)v);
      else out[((int64_t)(i >> 1) * chunks + chunk) * 2 + (i & 1)] = v;
    }
  } else {
    if (r < rows) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[(r * C + cg * 8 + j) * 2] = a[j];
        sm[(r * C + cg * 8 + j) * 2 + 1] = b[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += kT) {
      float acc = 0.f;
      for (int k = 0; k < rows; ++k) acc += sm[k * C * 2 + i];
      if (sums64) unsafeAtomicAdd(sums64 + i, (double)acc);
      else out[((int64_t)(i >> 1) * chunks + chunk) * 2 + (i & 1)] = acc;
    }
  }
}

// partial[n][c][chunk][2] = (sum x, sum x^2) over the chunk's voxels
__global__ void __launch_bounds__(kT) k_gn_partial(const bf16* __restrict__ x, int cstride, float* __restrict__ partial, int C,
                                                   int64_t V, int64_t vchunk, int sweep) {
  extern __shared__ float sm[];  // [rows][C][2]
  const int C8 = C / 8, rows = kT / C8;
  const int cg = threadIdx.x % C8, r = threadIdx.x / C8;
  const int n = blockIdx.y;
  // sweep: the blocks of an image advance through it TOGETHER (block b takes voxel lanes b*rows + r of every stride of gridDim.x*rows
  // voxels) instead of each block walking its own chunk: memory is touched front to back in time, so that the pass that follows can
  // walk back to front and meet the lines this one touched last while they are still in the 256 MiB Infinity Cache (mi_gn_bwd)
  const int64_t step = sweep ? (int64_t)gridDim.x * rows : rows;
  const int64_t v0 = sweep ? (int64_t)blockIdx.x * rows : blockIdx.x * vchunk, v1 = sweep ? V : ((v0 + vchunk < V) ? v0 + vchunk : V);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (r < rows) {
    const bf16* base = x + (int64_t)n * V * cstride + cg * 8;
    for (int64_t v = v0 + r; v < v1; v += 4 * step) {  // 4 independent 16-byte loads in flight per lane
      u32x4 raw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int64_t vk = v + (int64_t)k * step;
        u32x4 z = {0u, 0u, 0u, 0u};
        raw[k] = vk < v1 ? *(const u32x4*)(base + vk * cstride) : z;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        F8 f = unpack8(raw[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s[j] += f.v[j];
          q[j] += f.v[j] * f.v[j];
        }
      }
    }
  }
  block_fold(s, q, sm, partial + (int64_t)n * gridDim.x * C * 2, gridDim.x, blockIdx.x, C, C8, rows, r, cg);
}

// one block (256 threads) per (n, g): up to 1024 chunk partials per channel, so the loads are spread over 4 waves
__device__ __forceinline__ void block_sum2_d(double& a, double& b, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[(threadIdx.x >> 6) * 2] = a;
    red[(threadIdx.x >> 6) * 2 + 1] = b;
  }
  __syncthreads();
  a = (red[0] + red[2]) + (red[4] + red[6]);
  b = (red[1] + red[3]) + (red[5] + red[7]);
}

// (partial_b, chunks_b, Cb): optional second source holding the channels [Ca, Ca + Cb) -- see mi_gn_stats_from_partial
__global__ void __launch_bounds__(256) k_gn_finalize(const float* __restrict__ partial, int chunks, int Ca, const float* __restrict__ partial_b,
                                                     int chunks_b, int C, int G, int64_t V, float eps, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ scale_shift,
                                                     float* __restrict__ mean_rstd) {
  __shared__ double red[8];
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  double s = 0.0, q = 0.0;
  {  // the group's channels are adjacent: its (sum, sumsq) pairs are one contiguous run per source (a group may straddle the two)
    const int c0 = g * cpg, c1 = c0 + cpg;
    const int a0 = c0 < Ca ? c0 : Ca, a1 = c1 < Ca ? c1 : Ca;  // channels [a0, a1) come from source a, [b0, b1) from source b
    const float2* p = (const float2*)(partial + ((int64_t)n * Ca + a0) * chunks * 2);
    for (int i = threadIdx.x; i < (a1 - a0) * chunks; i += 256) {
      const float2 v = p[i];
      s += (double)v.x;
      q += (double)v.y;
    }
    if (c1 > Ca) {
      const int b0 = (c0 > Ca ? c0 : Ca) - Ca, b1 = c1 - Ca;
      const float2* pb = (const float2*)(partial_b + ((int64_t)n * (C - Ca) + b0) * chunks_b * 2);
      for (int i = threadIdx.x; i < (b1 - b0) * chunks_b; i += 256) {
        const float2 v = pb[i];
        s += (double)v.x;
        q += (double)v.y;
      }
    }
  }
  block_sum2_d(s, q, red);
  const double m = (double)V * cpg;
  const double mean = s / m;
  double var = q / m - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    mean_rstd[((int64_t)n * G + g) * 2] = (float)mean;
    mean_rstd[((int64_t)n * G + g) * 2 + 1] = rstd;
  }
  for (int i = threadIdx.x; i < cpg; i += 256) {
    int c = g * cpg + i;
    float sc = gamma[c] * rstd;
    scale_shift[((int64_t)n * C + c) * 2] = sc;
    scale_shift[((int64_t)n * C + c) * 2 + 1] = beta[c] - (float)mean * sc;
  }
}

// y = act(x * scale + shift).  grid = (gx, N) with gx * kT a multiple of C8: a thread keeps ONE channel octet of ONE image for its
// whole loop, so its 16 scale / shift values stay in registers, the loop has no integer division, and 4 independent 16-byte loads
// are in flight per lane.
template <int ACT>
__global__ void __launch_bounds__(kT) k_gn_apply(const bf16* __restrict__ x, int xcs, const float* __restrict__ scale_shift,
                                                 bf16* __restrict__ y, int ycs, int C8, int64_t V, int rev) {
  const int C = C8 * 8;
  const int tid = blockIdx.x * kT + threadIdx.x;
  const int cg = tid % C8, R = gridDim.x * kT / C8;
  const int64_t n = blockIdx.y;
  float sc[8], sh[8];
  load_ss(scale_shift + n * C * 2, cg * 8, sc, sh);
  // rev: walk the image back to front (voxel V-1-v): right behind a statistics pass that swept it front to back, the lines that
  // pass touched last are met first, while they are still in the Infinity Cache
  const bf16* xb = x + n * V * xcs + cg * 8 + (rev ? (V - 1) * xcs : 0);
  bf16* yb = y + n * V * ycs + cg * 8 + (rev ? (V - 1) * ycs : 0);
  const int64_t xs = rev ? -(int64_t)xcs : xcs, ys = rev ? -(int64_t)ycs : ycs;
  for (int64_t v = tid / C8; v < V; v += 4 * (int64_t)R) {
    u32x4 raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v + k * (int64_t)R < V) raw[k] = *(const u32x4*)(xb + (v + k * (int64_t)R) * xs);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (v + k * (int64_t)R >= V) break;
      F8 f = unpack8(raw[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float u = f.v[j] * sc[j] + sh[j];
        f.v[j] = act_f<ACT>(u);
      }
      *(u32x4*)(yb + (v + k * (int64_t)R) * ys) = pack8(f);
    }
  }
}

// backward partials: (sum du, sum du*x) per (n, chunk, c);  du = g * silu'(x*scale+shift)  (or g when !silu)
template <int ACT>
__global__ void __launch_bounds__(kT) k_gn_bwd_partial(const bf16* __restrict__ g, int gcs, const bf16* __restrict__ x, int xcs,
                                                       const float* __restrict__ scale_shift, float* __restrict__ partial, int C,
                                                       int64_t V, int64_t vchunk, int sweep, double* __restrict__ sums64, int nrep) {
  extern __shared__ float sm[];
  const int C8 = C / 8, rows = kT / C8;
  const int cg = threadIdx.x % C8, r = threadIdx.x / C8;
  const int n = blockIdx.y;
  const int64_t step = sweep ? (int64_t)gridDim.x * rows : rows;  // (see k_gn_partial)
  const int64_t v0 = sweep ? (int64_t)blockIdx.x * rows : blockIdx.x * vchunk, v1 = sweep ? V : ((v0 + vchunk < V) ? v0 + vchunk : V);
  float s1[8], s2[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (r < rows) {
    load_ss(scale_shift + (int64_t)n * C * 2, cg * 8, sc, sh);
    const bf16* xb = x + (int64_t)n * V * xcs + cg * 8;
    const bf16* gb = g + (int64_t)n * V * gcs + cg * 8;
    for (int64_t v = v0 + r; v < v1; v += 4 * step) {  // 8 independent 16-byte loads in flight per lane
      u32x4 rx[4], rg[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int64_t vk = v + (int64_t)k * step;
        u32x4 z = {0u, 0u, 0u, 0u};
        rx[k] = vk < v1 ? *(const u32x4*)(xb + vk * xcs) : z;
        rg[k] = vk < v1 ? *(const u32x4*)(gb + vk * gcs) : z;  // g = 0 beyond the chunk -> contributes nothing
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        F8 fx = unpack8(rx[k]), fg = unpack8(rg[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float du = fg.v[j];
          if (ACT) du *= act_grad_f<ACT>(fx.v[j] * sc[j] + sh[j]);
          s1[j] += du;
          s2[j] += du * fx.v[j];
        }
      }
    }
  }
  // fused form: the block's totals go to replica (block % nrep) of the image's sums -- 1024 blocks adding into ONE [C][2] record
  // serialise on its few cache lines (round 3: +1.0 ms per step); nrep records cut every address's queue by nrep
  block_fold(s1, s2, sm, partial + (int64_t)n * gridDim.x * C * 2, gridDim.x, blockIdx.x, C, C8, rows, r, cg,
             sums64 ? sums64 + ((int64_t)(blockIdx.x % nrep) * gridDim.y + n) * C * 2 : nullptr);
}

// per (n, g): coefficients of dx = a*du + b*x + c per channel, and the affine-parameter gradients
__global__ void __launch_bounds__(256) k_gn_bwd_finalize(const float* __restrict__ partial, int chunks, int C, int G, int64_t V,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                         float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ float sm[];  // [cpg][2] channel sums
  __shared__ double pd[512];     // [thread][2], then [channel][8][2]
  __shared__ double tot[2];
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const float mean = mean_rstd[((int64_t)n * G + g) * 2], rstd = mean_rstd[((int64_t)n * G + g) * 2 + 1];
  // All channels of the group at once (was: one block-wide fp64 reduction, two barriers, per channel in turn -- 8 serial rounds for
  // the 256-channel levels, ~9 us per launch, 51 launches per C4 step): `per` threads per channel read its chunks (consecutive threads,
  // consecutive float2), 8 threads per channel fold the strips, one thread per channel finishes.  cpg <= 32 (C <= 1024 at 32 groups);
  // wider groups take several rounds of 32 channels.
  double m1 = 0.0, m2 = 0.0;
  for (int i0 = 0; i0 < cpg; i0 += 32) {
    const int nc = cpg - i0 < 32 ? cpg - i0 : 32, per = 256 / nc;  // per >= 8
    const int i = threadIdx.x / per, j = threadIdx.x - i * per;
    double s1 = 0.0, s2 = 0.0;
    if (i < nc) {
      const float2* p = (const float2*)(partial + ((int64_t)n * C + g * cpg + i0 + i) * chunks * 2);
      for (int ch = j; ch < chunks; ch += per) {
        const float2 v = p[ch];
        s1 += (double)v.x;
        s2 += (double)v.y;
      }
    }
    __syncthreads();  // (pd / sm of the previous round have been read)
    pd[2 * threadIdx.x] = s1;
    pd[2 * threadIdx.x + 1] = s2;
    __syncthreads();
    double f1 = 0.0, f2 = 0.0;
    const int ci = threadIdx.x >> 3, k = threadIdx.x & 7;
    if (ci < nc)
      for (int jj = k; jj < per; jj += 8) {
        f1 += pd[2 * (ci * per + jj)];
        f2 += pd[2 * (ci * per + jj) + 1];
      }
    __syncthreads();
    pd[2 * threadIdx.x] = f1;
    pd[2 * threadIdx.x + 1] = f2;
    __syncthreads();
    if (threadIdx.x < nc) {
      double t1 = 0.0, t2 = 0.0;
      for (int k2 = 0; k2 < 8; ++k2) {
        t1 += pd[2 * (threadIdx.x * 8 + k2)];
        t2 += pd[2 * (threadIdx.x * 8 + k2) + 1];
      }
      const double s2hat = (double)rstd * (t2 - (double)mean * t1);  // sum du * xhat
      sm[2 * (i0 + threadIdx.x)] = (float)t1;
      sm[2 * (i0 + threadIdx.x) + 1] = (float)s2hat;
      pd[2 * threadIdx.x] = (double)gamma[g * cpg + i0 + threadIdx.x] * t1;
      pd[2 * threadIdx.x + 1] = (double)gamma[g * cpg + i0 + threadIdx.x] * s2hat;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a1 = i0 ? tot[0] : 0.0, a2 = i0 ? tot[1] : 0.0;
      for (int c2 = 0; c2 < nc; ++c2) { a1 += pd[2 * c2]; a2 += pd[2 * c2 + 1]; }
      tot[0] = a1; tot[1] = a2;
    }
  }
  __syncthreads();
  m1 = tot[0]; m2 = tot[1];
  __syncthreads();
  const double m = (double)V * cpg;
  m1 /= m;
  m2 /= m;
  const float b = (float)(-(double)rstd * rstd * m2);
  const float c0 = (float)((double)rstd * rstd * m2 * mean - (double)rstd * m1);
  for (int i = threadIdx.x; i < cpg; i += 256) {
    int c = g * cpg + i;
    float* o = coef + ((int64_t)n * C + c) * 3;
    o[0] = rstd * gamma[c];
    o[1] = b;
    o[2] = c0;
    if (dbeta) atomicAdd(dbeta + c, sm[2 * i]);
    if (dgamma) atomicAdd(dgamma + c, sm[2 * i + 1]);
  }
}

// dx = a*du + b*x + c (+ add); grid as in k_gn_apply: 40 per-channel constants stay in registers
template <int ACT, int U, bool NT>
__global__ void __launch_bounds__(kT) k_gn_bwd_apply(const bf16* __restrict__ g, int gcs, const bf16* __restrict__ x, int xcs,
                                                     const float* __restrict__ scale_shift, const float* __restrict__ coef,
                                                     const bf16* __restrict__ add, int acs, const bf16* __restrict__ add2, int a2cs,
                                                     bf16* __restrict__ dx, int dcs, int C8, int64_t V, int rev,
                                                     const double* __restrict__ sums64, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_rstd, int G, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int nrep) {
  extern __shared__ double ssum[];  // fused form only: this image's [C][2] sums, replicas folded in a fixed order
  const int C = C8 * 8;
  const int tid = blockIdx.x * kT + threadIdx.x;
  const int cg = tid % C8, R = gridDim.x * kT / C8;
  const int64_t n = blockIdx.y;
  float sc[8], sh[8], ca[8], cb[8], cc[8];
  load_ss(scale_shift + n * C * 2, cg * 8, sc, sh);
  if (sums64) {
    // mi_gn_bwd_fused: what k_gn_bwd_finalize computes per (n, group), recomputed here by every thread for the groups its channel octet
    // touches, from the image's [C][2] fp64 sums (sum du, sum du x) -- a few dozen L2 loads per thread instead of a launch of G blocks
    // in the dependency chain of every norm (51 launches of ~9 us per C4 step)
    const int cpg = C / G;
    const double m = (double)V * cpg;
    for (int i = threadIdx.x; i < 2 * C; i += kT) {
      double a = 0.0;
      for (int r = 0; r < nrep; ++r) a += sums64[((int64_t)r * gridDim.y + n) * C * 2 + i];
      ssum[i] = a;
    }
    __syncthreads();
    const double* sn = ssum;
    int gprev = -1;
    float b = 0.f, c0 = 0.f, rstd = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j, gi = c / cpg;
      if (gi != gprev) {  // (channels of an octet are consecutive: its groups come in order)
        gprev = gi;
        const float mean = mean_rstd[(n * G + gi) * 2];
        rstd = mean_rstd[(n * G + gi) * 2 + 1];
        double m1 = 0.0, m2 = 0.0;
        for (int i = 0; i < cpg; ++i) {
          const int ci = gi * cpg + i;
          const double s1 = sn[2 * ci], s2 = sn[2 * ci + 1];
          m1 += (double)gamma[ci] * s1;
          m2 += (double)gamma[ci] * ((double)rstd * (s2 - (double)mean * s1));
        }
        m1 /= m; m2 /= m;
        b = (float)(-(double)rstd * rstd * m2);
        c0 = (float)((double)rstd * rstd * m2 * mean - (double)rstd * m1);
      }
      ca[j] = rstd * gamma[c]; cb[j] = b; cc[j] = c0;
      if (tid / C8 == 0) {  // one thread per (image, octet): the affine-parameter gradients of its channels
        const float mean = mean_rstd[(n * G + gi) * 2];
        const double s1 = sn[2 * c], s2 = sn[2 * c + 1];
        if (dbeta) atomicAdd(dbeta + c, (float)s1);
        if (dgamma) atomicAdd(dgamma + c, (float)((double)rstd * (s2 - (double)mean * s1)));
      }
    }
  } else {
    const float* cf = coef + (n * C + cg * 8) * 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ca[j] = cf[3 * j]; cb[j] = cf[3 * j + 1]; cc[j] = cf[3 * j + 2]; }
  }
  const bf16* xb = x + n * V * xcs + cg * 8;
  const bf16* gb = g + n * V * gcs + cg * 8;
  const bf16* ab = add ? add + n * V * acs + cg * 8 : nullptr;
  const bf16* a2b = add2 ? add2 + n * V * a2cs + cg * 8 : nullptr;  // second pending branch of x's gradient (e.g. a slice of d(concat))
  bf16* db = dx + n * V * dcs + cg * 8;
  for (int64_t v = tid / C8; v < V; v += U * (int64_t)R) {
    u32x4 rx[U], rg[U], ra[U], ra2[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t vf = v + k * (int64_t)R, vk = rev ? V - 1 - vf : vf;  // rev: back to front (see k_gn_apply)
      if (vf < V) {
        rx[k] = *(const u32x4*)(xb + vk * xcs);
        rg[k] = NT ? __builtin_nontemporal_load((const u32x4*)(gb + vk * gcs)) : *(const u32x4*)(gb + vk * gcs);
        if (ab) ra[k] = NT ? __builtin_nontemporal_load((const u32x4*)(ab + vk * acs)) : *(const u32x4*)(ab + vk * acs);
        if (a2b) ra2[k] = NT ? __builtin_nontemporal_load((const u32x4*)(a2b + vk * a2cs)) : *(const u32x4*)(a2b + vk * a2cs);
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t vf = v + k * (int64_t)R, vk = rev ? V - 1 - vf : vf;
      if (vf >= V) break;
      F8 fx = unpack8(rx[k]), fg = unpack8(rg[k]), o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float du = fg.v[j];
        if (ACT) du *= act_grad_f<ACT>(fx.v[j] * sc[j] + sh[j]);
        o.v[j] = ca[j] * du + cb[j] * fx.v[j] + cc[j];
      }
      if (ab) {
        F8 fa = unpack8(ra[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] += fa.v[j];
      }
      if (a2b) {
        F8 fa = unpack8(ra2[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] += fa.v[j];
      }
      if (NT) __builtin_nontemporal_store(pack8(o), (u32x4*)(db + vk * dcs));
      else *(u32x4*)(db + vk * dcs) = pack8(o);
    }
  }
}

// x-extent of the (gx, N) grid of the streaming apply kernels: ~4096 blocks in all, gx * kT a multiple of C8 (see k_gn_apply);
// total = 16-byte pieces of ONE image
inline int64_t apply_grid(int64_t total, int C8, int N) {
  int64_t grid = (total + kT - 1) / kT;
  static const int per_cu = [] { const char* e = getenv("MI_GN_BLOCKS_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 16; }();  // A/B knob
  const int64_t cap = 256 * per_cu / N > 64 ? 256 * per_cu / N : 64;
  if (grid > cap) grid = cap;
  int m = C8;  // kT * grid % C8 == 0  <=>  grid % (C8 / gcd(C8, kT)) == 0
  for (int a = kT, b = C8; b;) { int t = a % b; a = b; b = t; m = C8 / a; }
  if (grid >= m) grid -= grid % m;
  else grid = m;  // tiny tensors: a few idle blocks
  return grid;
}

inline int64_t pick_vchunk(int64_t V) {
  // ~1024 blocks whatever the volume: a 16^3 x 256-channel tensor cut into 2048-voxel chunks would be streamed by 2 CUs
  static const int nchunks = [] { const char* e = getenv("MI_GN_CHUNKS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();  // A/B knobs
  static const int nchunks_big = [] { const char* e = getenv("MI_GN_CHUNKS_BIG"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
  const int nc = (nchunks_big && V >= (1 << 20)) ? nchunks_big : nchunks;
  int64_t vc = (V + nc - 1) / nc;
  if (vc < 16) vc = 16;
  return vc;
}
// MI_GN_SWEEP (A/B knob, default 0 = every block walks its own chunk, both passes front to back).  Bit 0: the statistics passes sweep an
// image front to back with all blocks abreast and the backward's apply pass walks it back to front, so that it meets the lines the
// partial pass touched last while they could still be in the 256 MiB Infinity Cache; bit 1: the forward apply pass walks back to front
// too.  Measured (round 3, profiles/r03b_ab_gn_sweep.log, same box, interleaved): 24.20 ms/step with 0, 24.25 with 1, 24.26 with 3;
// mi_gn_bwd at 32 ch x 128^3 back to back: 139 us (0) vs 146 (1).  The two passes of a 268 MB working set gain nothing from the order.
inline int gn_sweep() {
  static const int v = [] { const char* e = getenv("MI_GN_SWEEP"); return e ? atoi(e) : 0; }();
  return v;
}
inline bool bad_c(int C, int G) { return C <= 0 || (C & 7) || C > 2048 || G <= 0 || C % G != 0; }

// ---- small tensors (the 16^3 level of the C4 net; 2-D nets): statistics -> coefficients -> apply in ONE launch.  Three launches of
// ~6 us each (launch floor: the tensors are 2 MB) plus their boundaries cost ~22 us per norm and there are ~20 such norms per pass.
// One 1024-thread workgroup per (image, group): a group's channels are an aligned run of 16 or 32 bytes per voxel, every thread issues
// ALL its loads up front (<= 8 pieces of 16 bytes; the group's 64-128 KB stay in registers), one block-wide fp64 fold gives mean / rstd,
// and the activated tensor is written from the registers.  (Round 2 tried one 256-thread block per group looping over the voxels:
// a chain of load latencies, slower than the three launches.)
constexpr int kSmallT = 1024, kSmallP = 8;
__device__ __forceinline__ void block_sum2_d16(double& a, double& b, double* red) {  // 16 waves
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[(threadIdx.x >> 6) * 2] = a;
    red[(threadIdx.x >> 6) * 2 + 1] = b;
  }
  __syncthreads();
  a = b = 0.0;
#pragma unroll
  for (int w = 0; w < kSmallT / 64; ++w) {
    a += red[2 * w];
    b += red[2 * w + 1];
  }
}
template <int ACT>
__global__ void __launch_bounds__(kSmallT) k_gn_small_fwd(const bf16* __restrict__ x, int xcs, bf16* __restrict__ y, int ycs, int C, int G, int V,
                                                          float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ scale_shift, float* __restrict__ mean_rstd) {
  __shared__ double red[2 * kSmallT / 64];
  const int g = blockIdx.x, n = blockIdx.y;
  const int cpg = C / G, pcs = cpg >> 3;            // 16-byte pieces per voxel of this group (1 or 2)
  const int total = V * pcs;                        // <= kSmallP * kSmallT
  const int q = threadIdx.x % pcs;                  // this thread's octet inside the group: the same for all its pieces (1024 % pcs == 0)
  const bf16* xb = x + (int64_t)n * V * xcs + g * cpg + q * 8;
  u32x4 raw[kSmallP];
#pragma unroll
  for (int i = 0; i < kSmallP; ++i) {
    const int idx = threadIdx.x + i * kSmallT;
    u32x4 z = {0u, 0u, 0u, 0u};
    raw[i] = idx < total ? *(const u32x4*)(xb + (int64_t)(idx / pcs) * xcs) : z;
  }
  float s = 0.f, sq = 0.f;
#pragma unroll
  for (int i = 0; i < kSmallP; ++i) {
    const F8 f = unpack8(raw[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s += f.v[j];
      sq = fmaf(f.v[j], f.v[j], sq);
    }
  }
  double ds = (double)s, dq = (double)sq;
  block_sum2_d16(ds, dq, red);
  const double m = (double)V * cpg;
  const double mean = ds / m;
  double var = dq / m - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = g * cpg + q * 8 + j;
    sc[j] = gamma[c] * rstd;
    sh[j] = beta[c] - (float)mean * sc[j];
  }
  if (threadIdx.x < pcs) {  // one thread per octet publishes what the backward (and any other consumer of the statistics) reads
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = g * cpg + q * 8 + j;
      scale_shift[((int64_t)n * C + c) * 2] = sc[j];
      scale_shift[((int64_t)n * C + c) * 2 + 1] = sh[j];
    }
    if (threadIdx.x == 0) {
      mean_rstd[((int64_t)n * G + g) * 2] = (float)mean;
      mean_rstd[((int64_t)n * G + g) * 2 + 1] = rstd;
    }
  }
  bf16* yb = y + (int64_t)n * V * ycs + g * cpg + q * 8;
#pragma unroll
  for (int i = 0; i < kSmallP; ++i) {
    const int idx = threadIdx.x + i * kSmallT;
    if (idx < total) {
      F8 f = unpack8(raw[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = act_f<ACT>(f.v[j] * sc[j] + sh[j]);
      *(u32x4*)(yb + (int64_t)(idx / pcs) * ycs) = pack8(f);
    }
  }
}
inline bool small_ok(int N, int64_t V, int C, int G) {
  if (bad_c(C, G) || N <= 0 || V <= 0) return false;
  const int cpg = C / G;
  return (cpg == 8 || cpg == 16) && V * (cpg >> 3) <= (int64_t)kSmallP * kSmallT;
}

}  // namespace

extern "C" {

int64_t mi_gn_workspace_bytes(int N, int64_t V, int C) {
  int64_t vc = pick_vchunk(V);
  return (int64_t)N * ((V + vc - 1) / vc) * C * 2 * (int64_t)sizeof(float);
}

int mi_gn_stats(const void* x, int x_cstride, int N, int64_t V, int C, int G, float eps, const float* gamma, const float* beta,
                float* scale_shift, float* mean_rstd, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if (bad_c(C, G) || N <= 0 || V <= 0 || (x_cstride & 7) || x_cstride < C) return MI_ERR_BAD_ARG;
  if (workspace_bytes < mi_gn_workspace_bytes(N, V, C)) return MI_ERR_BAD_ARG;
  int64_t vc = pick_vchunk(V);
  int chunks = (int)((V + vc - 1) / vc);
  int rows = kT / (C / 8);
  hipLaunchKernelGGL(k_gn_partial, dim3(chunks, N), dim3(kT), sizeof(float) * (size_t)rows * C * 2, st, (const bf16*)x, x_cstride,
                     (float*)workspace, C, V, vc, gn_sweep() & 1);
  hipLaunchKernelGGL(k_gn_finalize, dim3(N * G), dim3(256), 0, st, (const float*)workspace, chunks, C, (const float*)nullptr, 0, C, G, V, eps,
                     gamma, beta, scale_shift, mean_rstd);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_gn_stats_from_partial(const float* pa, int chunks_a, int Ca, const float* pb, int chunks_b, int Cb, int N, int64_t V, int G, float eps,
                             const float* gamma, const float* beta, float* scale_shift, float* mean_rstd, hipStream_t st) {
  const int C = Ca + Cb;
  if (!pa || chunks_a <= 0 || Ca <= 0 || Cb < 0 || (Cb > 0 && (!pb || chunks_b <= 0)) || bad_c(C, G) || N <= 0 || V <= 0) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_gn_finalize, dim3(N * G), dim3(256), 0, st, pa, chunks_a, Ca, pb, chunks_b, C, G, V, eps, gamma, beta, scale_shift,
                     mean_rstd);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_gn_apply(const void* x, int x_cstride, const float* scale_shift, void* y, int y_cstride, int N, int64_t V, int C, int silu,
                hipStream_t st) {
  if (C <= 0 || (C & 7) || (x_cstride & 7) || (y_cstride & 7) || N <= 0 || V <= 0) return MI_ERR_BAD_ARG;
  int64_t grid = apply_grid(V * (C / 8), C / 8, N);
  if (silu < 0 || silu > 2) return MI_ERR_BAD_ARG;  // activation code: 0 none, 1 SiLU, 2 LeakyReLU(0.2)
  auto k = silu == 1 ? k_gn_apply<1> : (silu == 2 ? k_gn_apply<2> : k_gn_apply<0>);
  hipLaunchKernelGGL(k, dim3((int)grid, N), dim3(kT), 0, st, (const bf16*)x, x_cstride, scale_shift, (bf16*)y, y_cstride, C / 8, V,
                     gn_sweep() & 2 ? 1 : 0);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_gn_small_supported(int N, int64_t V, int C, int G) { return small_ok(N, V, C, G) ? 1 : 0; }

int mi_gn_small_fwd(const void* x, int x_cstride, void* y, int y_cstride, int N, int64_t V, int C, int G, float eps, const float* gamma,
                    const float* beta, float* scale_shift, float* mean_rstd, int act, hipStream_t st) {
  if (!small_ok(N, V, C, G)) return MI_ERR_UNSUPPORTED;
  if (!x || !y || !gamma || !beta || !scale_shift || !mean_rstd || (x_cstride & 7) || (y_cstride & 7) || x_cstride < C || y_cstride < C ||
      act < 0 || act > 2)
    return MI_ERR_BAD_ARG;
  auto k = act == 1 ? k_gn_small_fwd<1> : (act == 2 ? k_gn_small_fwd<2> : k_gn_small_fwd<0>);
  hipLaunchKernelGGL(k, dim3(G, N), dim3(kSmallT), 0, st, (const bf16*)x, x_cstride, (bf16*)y, y_cstride, C, G, (int)V, eps, gamma, beta,
                     scale_shift, mean_rstd);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_gn_bwd(const void* g, int g_cstride, const void* x, int x_cstride, int N, int64_t V, int C, int G, const float* gamma,
              const float* scale_shift, const float* mean_rstd, int silu, const void* add, int add_cstride, const void* add2, int add2_cstride,
              void* dx, int dx_cstride, float* dgamma, float* dbeta, float* coef, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if (bad_c(C, G) || N <= 0 || V <= 0 || (x_cstride & 7) || (g_cstride & 7) || (dx_cstride & 7) || (add && (add_cstride & 7)) ||
      (add2 && (!add || (add2_cstride & 7))))
    return MI_ERR_BAD_ARG;
  if (workspace_bytes < mi_gn_workspace_bytes(N, V, C)) return MI_ERR_BAD_ARG;
  int64_t vc = pick_vchunk(V);
  int chunks = (int)((V + vc - 1) / vc);
  int rows = kT / (C / 8);
  if (silu < 0 || silu > 2) return MI_ERR_BAD_ARG;
  auto kp = silu == 1 ? k_gn_bwd_partial<1> : (silu == 2 ? k_gn_bwd_partial<2> : k_gn_bwd_partial<0>);
  hipLaunchKernelGGL(kp, dim3(chunks, N), dim3(kT), sizeof(float) * (size_t)rows * C * 2, st, (const bf16*)g, g_cstride, (const bf16*)x,
                     x_cstride, scale_shift, (float*)workspace, C, V, vc, gn_sweep() & 1, (double*)nullptr, 1);
  hipLaunchKernelGGL(k_gn_bwd_finalize, dim3(N * G), dim3(256), sizeof(float) * 2 * (size_t)(C / G), st, (const float*)workspace, chunks, C,
                     G, V, gamma, mean_rstd, coef, dgamma, dbeta);
  int64_t grid = apply_grid(V * (C / 8), C / 8, N);
  // MI_GN_VARIANT (A/B; bit 0 = 4 pieces in flight per stream instead of 2, bit 1 = non-temporal loads of the streams this pass reads for
  // the last time (g, the pending gradient branches) and non-temporal stores of dx).  Measured round 3, same box, interleaved
  // (profiles/r03i_ab_gn_variant.log): 22.96 ms/step with 0, 23.03 with 1, 22.75 with 2 (the default), 22.87 with 3; mi_gn_bwd at
  // 32 ch x 128^3 back to back 132.8 / 142.0 / 124.4 / 129.8 us.  The same hints on the forward apply pass and on the backward's partial
  // pass measured within noise (profiles/r03i_ab_gn_nt2.log) and were not kept.
  static const int variant = [] { const char* e = getenv("MI_GN_VARIANT"); return e ? atoi(e) : 2; }();
  auto ka = silu == 1 ? k_gn_bwd_apply<1, 2, false> : (silu == 2 ? k_gn_bwd_apply<2, 2, false> : k_gn_bwd_apply<0, 2, false>);
  if (variant == 1) ka = silu == 1 ? k_gn_bwd_apply<1, 4, false> : (silu == 2 ? k_gn_bwd_apply<2, 4, false> : k_gn_bwd_apply<0, 4, false>);
  else if (variant == 2) ka = silu == 1 ? k_gn_bwd_apply<1, 2, true> : (silu == 2 ? k_gn_bwd_apply<2, 2, true> : k_gn_bwd_apply<0, 2, true>);
  else if (variant == 3) ka = silu == 1 ? k_gn_bwd_apply<1, 4, true> : (silu == 2 ? k_gn_bwd_apply<2, 4, true> : k_gn_bwd_apply<0, 4, true>);
  hipLaunchKernelGGL(ka, dim3((int)grid, N), dim3(kT), 0, st, (const bf16*)g, g_cstride, (const bf16*)x, x_cstride, scale_shift, coef,
                     (const bf16*)add, add_cstride, (const bf16*)add2, add2_cstride, (bf16*)dx, dx_cstride, C / 8, V, gn_sweep() & 1,
                     (const double*)nullptr, (const float*)nullptr, (const float*)nullptr, G, (float*)nullptr, (float*)nullptr, 1);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_gn_bwd_fused(const void* g, int g_cstride, const void* x, int x_cstride, int N, int64_t V, int C, int G, const float* gamma,
                    const float* scale_shift, const float* mean_rstd, int silu, const void* add, int add_cstride, const void* add2,
                    int add2_cstride, void* dx, int dx_cstride, float* dgamma, float* dbeta, double* sums_zeroed, hipStream_t st) {
  if (bad_c(C, G) || N <= 0 || V <= 0 || (x_cstride & 7) || (g_cstride & 7) || (dx_cstride & 7) || (add && (add_cstride & 7)) ||
      (add2 && (!add || (add2_cstride & 7))) || !sums_zeroed || !gamma || !mean_rstd)
    return MI_ERR_BAD_ARG;
  int64_t vc = pick_vchunk(V);
  int chunks = (int)((V + vc - 1) / vc);
  int rows = kT / (C / 8);
  if (silu < 0 || silu > 2) return MI_ERR_BAD_ARG;
  // replicas of the sums: one per 64 blocks of the partial pass, at most MI_GN_FUSED_REPLICAS (16; the caller's buffer is sized for that)
  static const int rep_div = [] { const char* e = getenv("MI_GN_FUSED_REP_DIV"); return e && atoi(e) > 0 ? atoi(e) : 64; }();
  int nrep = chunks / rep_div;
  nrep = nrep < 1 ? 1 : (nrep > MI_GN_FUSED_REPLICAS ? MI_GN_FUSED_REPLICAS : nrep);
  auto kp = silu == 1 ? k_gn_bwd_partial<1> : (silu == 2 ? k_gn_bwd_partial<2> : k_gn_bwd_partial<0>);
  hipLaunchKernelGGL(kp, dim3(chunks, N), dim3(kT), sizeof(float) * (size_t)rows * C * 2, st, (const bf16*)g, g_cstride, (const bf16*)x,
                     x_cstride, scale_shift, (float*)nullptr, C, V, vc, 0, sums_zeroed, nrep);
  int64_t grid = apply_grid(V * (C / 8), C / 8, N);
  static const int variant = [] { const char* e = getenv("MI_GN_VARIANT"); return e ? atoi(e) : 2; }();
  auto ka = silu == 1 ? k_gn_bwd_apply<1, 2, false> : (silu == 2 ? k_gn_bwd_apply<2, 2, false> : k_gn_bwd_apply<0, 2, false>);
  if (variant == 1) ka = silu == 1 ? k_gn_bwd_apply<1, 4, false> : (silu == 2 ? k_gn_bwd_apply<2, 4, false> : k_gn_bwd_apply<0, 4, false>);
  else if (variant == 2) ka = silu == 1 ? k_gn_bwd_apply<1, 2, true> : (silu == 2 ? k_gn_bwd_apply<2, 2, true> : k_gn_bwd_apply<0, 2, true>);
  else if (variant == 3) ka = silu == 1 ? k_gn_bwd_apply<1, 4, true> : (silu == 2 ? k_gn_bwd_apply<2, 4, true> : k_gn_bwd_apply<0, 4, true>);
  hipLaunchKernelGGL(ka, dim3((int)grid, N), dim3(kT), sizeof(double) * 2 * (size_t)C, st, (const bf16*)g, g_cstride, (const bf16*)x, x_cstride,
                     scale_shift, (const float*)nullptr, (const bf16*)add, add_cstride, (const bf16*)add2, add2_cstride, (bf16*)dx, dx_cstride,
                     C / 8, V, 0, (const double*)sums_zeroed, gamma, mean_rstd, G, dgamma, dbeta, nrep);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
