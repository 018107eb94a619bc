// 3x3x3 stride-1 pad-1 convolutions with ONE channel on one side: the network's input conv (1 -> C, UNet:1820-1828, AEKL:393) and
// output conv (C -> 1, UNet:1935-1943, AEKL:609).  On the implicit-GEMM kernels one side of the GEMM is padded from 1 to 32, so
// 31/32 of the MFMA and LDS work is zeros (0.2 ms per call at 128^3 for 3.6 GFLOP); these are plain streaming kernels instead:
//   k_c1_expand  : y[v][c]  = a[c] + sum_t s[v + t] W[c][t]          forward of 1 -> C, and data gradient of C -> 1 (taps flipped)
//   k_c1_reduce  : y[v]     = a    + sum_t sum_c x[v + t][c] W[c][t]  forward of C -> 1:  Y[u][t] = x[u][:] . W[:][t] on the MFMA for
//                                                                    the tile + halo, then 27 shifted LDS reads per output voxel
// (Weight gradients stay on the table-driven MFMA kernel: a VALU formulation needs 216 accumulators per thread and measured slower.)
// W is the fp32 master weight in torch layout ([C][1][27] or [1][C][27]: index c * 27 + t either way) read through the scalar cache.
#include "common.h"
#include "conv_common.h"
#include "medimgen_hip.h"

namespace {

struct C1Args {
  int N, D, H, W, C;
  const bf16* s;   // single-channel tensor, voxel pitch s_cs
  int s_cs;
  const bf16* m;   // C-channel tensor, voxel pitch m_cs
  int m_cs;
  const float* w;  // [C][27] fp32
  const float* addvec;  // bias (+ per-image vector): a[n * av_stride + c]; may be null
  int av_stride;
  bf16* y;
  int y_cs;
};

// ---------------------------------------------------------------------------------------------- 1 -> C (and dgrad of C -> 1)
// Thread = voxel (its 27 neighbours live in registers, weights come through the scalar cache).  Neighbour loads are UNCONDITIONAL on
// clamped coordinates and masked afterwards: a predicated load per tap becomes 27 dependent branch + wait sequences.  The wave's
// 64 voxels x C channels go through a wave-private LDS tile so that the global stores are contiguous 16-byte pieces.
constexpr int kExpPitch = 64 * 2 + 16;  // bytes per voxel row of the staging tile (C <= 64)
template <int FLIP>
__global__ void __launch_bounds__(256) k_c1_expand(C1Args a) {
  __shared__ __attribute__((aligned(16))) char stage[4][64 * kExpPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)a.N * a.D * a.H * a.W;
  const int64_t v0 = blockIdx.x * 256ll + wave * 64;  // first voxel of this wave
  if (v0 >= total) return;
  const int64_t v = v0 + lane < total ? v0 + lane : total - 1;
  const int wx = (int)(v % a.W);
  int64_t t1 = v / a.W;
  const int hy = (int)(t1 % a.H);
  t1 /= a.H;
  const int dz = (int)(t1 % a.D), n = (int)(t1 / a.D);
  const bf16* sn = a.s + (int64_t)n * a.D * a.H * a.W * a.s_cs;
  float xs[27];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int z = dz + kd - 1, yy = hy + kh - 1, x = wx + kw - 1;
        const int zc = min(max(z, 0), a.D - 1), yc = min(max(yy, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
        const float val = bf2f(sn[(((int64_t)zc * a.H + yc) * a.W + xc) * a.s_cs]);
        xs[(kd * 3 + kh) * 3 + kw] = (z == zc && yy == yc && x == xc) ? val : 0.f;
      }
  char* my = stage[wave];
  for (int c0 = 0; c0 < a.C; c0 += 8) {
    F8 acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] = a.addvec ? a.addvec[(int64_t)n * a.av_stride + c0 + j] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* wr = a.w + (c0 + j) * 27;  // uniform address: scalar loads
#pragma unroll
      for (int t = 0; t < 27; ++t) acc.v[j] = fmaf(xs[t], wr[FLIP ? 26 - t : t], acc.v[j]);
    }
    *(u32x4*)(my + lane * kExpPitch + c0 * 2) = pack8(acc);
  }
  // (wave-private tile: LDS operations of one wave complete in order, no barrier needed)
  const int C8 = a.C / 8;
  for (int i = lane; i < 64 * C8; i += 64) {
    const int vox = i / C8, ch = i % C8;
    if (v0 + vox < total) *(u32x4*)(a.y + (v0 + vox) * a.y_cs + ch * 8) = *(const u32x4*)(my + vox * kExpPitch + ch * 16);
  }
}

// ---------------------------------------------------------------------------------------------- C -> 1 forward
constexpr int kTD = 4, kTH = 8, kTW = 8, kHD = kTD + 2, kHH = kTH + 2, kHW = kTW + 2, kHalo = kHD * kHH * kHW;  // 600 halo voxels
constexpr int kYP = 33;  // Y row pitch in floats (odd: the 27 shifted reads of a wave hit distinct banks)

template <int NK>  // k-steps of 16 channels
__global__ void __launch_bounds__(256) k_c1_reduce(C1Args a) {
  extern __shared__ float Y[];  // [608][kYP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int tw = (a.W + kTW - 1) / kTW, th = (a.H + kTH - 1) / kTH, td = (a.D + kTD - 1) / kTD;
  int tile = blockIdx.x;
  const int tx = tile % tw; tile /= tw;
  const int ty = tile % th; tile /= th;
  const int tz = tile % td, n = tile / td;
  const int z0 = tz * kTD - 1, y0 = ty * kTH - 1, x0 = tx * kTW - 1;  // halo origin
  // A fragments: row = tap r, k = channel ks*16 + 8h + j
  bf16x8 wf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    F8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ks * 16 + 8 * h + j;
      f.v[j] = (r < 27 && c < a.C) ? a.w[c * 27 + r] : 0.f;
    }
    wf[ks] = __builtin_bit_cast(bf16x8, pack8(f));
  }
  const bf16* xn = a.m + (int64_t)n * a.D * a.H * a.W * a.m_cs;
  constexpr int NB = ((kHalo + 31) / 32 + 3) / 4;  // halo blocks of 32 voxels per wave
  u32x4 raw[NB][NK];
  bool ok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {  // all loads first (clamped addresses, masked below): one memory latency per tile, not one per block
    const int hu = (wave + 4 * i) * 32 + r;
    const int hx = hu % kHW, hy = (hu / kHW) % kHH, hz = hu / (kHW * kHH);
    const int z = z0 + hz, yy = y0 + hy, x = x0 + hx;
    const int zc = min(max(z, 0), a.D - 1), yc = min(max(yy, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
    ok[i] = hu < kHalo && z == zc && yy == yc && x == xc;
    const bf16* px = xn + (((int64_t)zc * a.H + yc) * a.W + xc) * a.m_cs;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int c = min(ks * 16 + 8 * h, a.C - 8);
      raw[i][ks] = *(const u32x4*)(px + c);
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int blk = wave + 4 * i;
    if (blk >= (kHalo + 31) / 32) break;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      u32x4 b = raw[i][ks];
      if (!(ok[i] && ks * 16 + 8 * h < a.C)) b = u32x4{0u, 0u, 0u, 0u};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    // lane (voxel r, h) holds taps (e & 3) + 8 (e >> 2) + 4h
    const int hu = blk * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) Y[hu * kYP + (e & 3) + 8 * (e >> 2) + 4 * h] = acc[e];
  }
  __syncthreads();
  const int ox = threadIdx.x & 7, oy = (threadIdx.x >> 3) & 7, oz = threadIdx.x >> 6;
  const int z = tz * kTD + oz, yy = ty * kTH + oy, x = tx * kTW + ox;
  if (z >= a.D || yy >= a.H || x >= a.W) return;
  float o = a.addvec ? a.addvec[(int64_t)n * a.av_stride] : 0.f;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) o += Y[(((oz + kd) * kHH + oy + kh) * kHW + ox + kw) * kYP + (kd * 3 + kh) * 3 + kw];
  a.y[((((int64_t)n * a.D + z) * a.H + yy) * a.W + x) * a.y_cs] = f2bf(o);
}

bool c1_dims_ok(int N, int D, int H, int W, int C) {
  return N > 0 && D > 0 && H > 0 && W > 0 && C >= 8 && C <= 64 && (C & 7) == 0 && (int64_t)N * D * H * W < (1ll << 31);
}

}  // namespace

int mi_launch_c1_expand(const void* s, int s_cs, const float* w, const float* addvec, int av_stride, void* y, int y_cs, int N, int D, int H,
                        int W, int C, int flip, hipStream_t st) {
  if (!c1_dims_ok(N, D, H, W, C) || (y_cs & 7)) return MI_ERR_UNSUPPORTED;
  C1Args a{};
  a.N = N; a.D = D; a.H = H; a.W = W; a.C = C;
  a.s = (const bf16*)s; a.s_cs = s_cs; a.w = w; a.addvec = addvec; a.av_stride = av_stride; a.y = (bf16*)y; a.y_cs = y_cs;
  const int64_t total = (int64_t)N * D * H * W;
  dim3 grid((unsigned)((total + 255) / 256));
  if (flip) hipLaunchKernelGGL(k_c1_expand<1>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_c1_expand<0>, grid, dim3(256), 0, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_launch_c1_reduce(const void* x, int x_cs, const float* w, const float* addvec, int av_stride, void* y, int y_cs, int N, int D, int H,
                        int W, int C, hipStream_t st) {
  if (!c1_dims_ok(N, D, H, W, C) || (x_cs & 7)) return MI_ERR_UNSUPPORTED;
  C1Args a{};
  a.N = N; a.D = D; a.H = H; a.W = W; a.C = C;
  a.m = (const bf16*)x; a.m_cs = x_cs; a.w = w; a.addvec = addvec; a.av_stride = av_stride; a.y = (bf16*)y; a.y_cs = y_cs;
  const int64_t tiles = (int64_t)N * ((D + kTD - 1) / kTD) * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
  if (tiles >= (1ll << 31)) return MI_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * (size_t)((kHalo + 31) / 32 * 32) * kYP;
  const int nk = (C + 15) / 16;
  auto k = nk == 1 ? k_c1_reduce<1> : nk == 2 ? k_c1_reduce<2> : nk == 3 ? k_c1_reduce<3> : k_c1_reduce<4>;
  static bool attr_done[4] = {false, false, false, false};
  if (!attr_done[nk - 1]) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done[nk - 1] = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256), lds, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}
