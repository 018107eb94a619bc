// 3x3x3 stride-1 pad-1 convolutions with ONE channel on one side: the network's input conv (1 -> C, UNet:1820-1828, AEKL:393) and
// output conv (C -> 1, UNet:1935-1943, AEKL:609).  On the implicit-GEMM kernels one side of the GEMM is padded from 1 to 32, so
// 31/32 of the MFMA and LDS work is zeros (0.2 ms per call at 128^3 for 3.6 GFLOP); these are plain streaming kernels instead:
//   k_c1_expand  : y[v][c]  = a[c] + sum_t s[v + t] W[c][t]          forward of 1 -> C, and data gradient of C -> 1 (taps flipped)
//   k_c1_reduce  : y[v]     = a    + sum_t sum_c x[v + t][c] W[c][t]  forward of C -> 1:  Y[u][t] = x[u][:] . W[:][t] on the MFMA for
//                                                                    the tile + halo, then 27 shifted LDS reads per output voxel
//   k_c1_wgrad   : dW[c][t] += sum_v m[v][c] s[v +- t]   weight gradient of both (m = the C-channel tensor, s = the single-channel one):
//                  D[c][t] = M^T[c][vox] B[vox][t] on the MFMA with the 27 taps as the N axis; the "im2col" B fragment is gathered from a
//                  600-voxel LDS halo of s.  One HBM pass over m (the table-driven MFMA kernel padded the 1-channel side to 32: 199 us
//                  per call at 128^3 for a 134 MB read).
// W is the fp32 master weight in torch layout ([C][1][27] or [1][C][27]: index c * 27 + t either way) read through the scalar cache.
#include <stdlib.h>

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
  // k_c1_expand_mfma<1, 0> only: per-channel (sum, sum of squares) of the bf16 OUTPUT in the layout of the GroupNorm partials,
  // stats[n][c][chunk][2], chunk = blockIdx.x, stats_chunks = gridDim.x (every entry written); null: none
  float* stats;
  int stats_chunks;
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

// ---------------------------------------------------------------------------------------------- 1 -> C on the MFMA
// The same product with the 27 taps as the K axis: D[c][v] = W[c][t] B[t][v], B[t][v] = s[v + t] gathered from a wave-private LDS
// halo (3 slices x 10 x 10) of the single-channel tensor; two k-steps of 16 taps (27 padded to 32) per block of 32 voxels.  The
// weight rows are permuted like k_conv27's (MFMA row rho <-> channel 16 ((rho >> 2) & 1) + (rho & 3) + 4 (rho >> 3)) so that a
// lane's 16 accumulator registers are 16 CONSECUTIVE channels of its voxel: the result is stored straight from the registers as
// two 16-byte pieces per lane, no staging tile, no 864-FMA VALU loop.  One HBM write pass (the VALU kernel above: 86 us at 128^3
// for a 134 MB write).
constexpr int kTD = 4, kTH = 8, kTW = 8, kHD = kTD + 2, kHH = kTH + 2, kHW = kTW + 2, kHalo = kHD * kHH * kHW;  // 600 halo voxels
template <int NCB, int FLIP>
__global__ void __launch_bounds__(256) k_c1_expand_mfma(C1Args a, int ntiles) {
  constexpr int HS = 3 * kHH * kHW;
  __shared__ bf16 sh[4][2][HS + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, v = lane & 31, h = lane >> 5;
  const int tw = (a.W + kTW - 1) / kTW, th = (a.H + kTH - 1) / kTH, td = (a.D + kTD - 1) / kTD;
  // A fragments: row rho = v (lane & 31) carries channel crow; k = tap ks * 16 + 8 h + j
  const int crow = 16 * ((v >> 2) & 1) + (v & 3) + 4 * (v >> 3);
  bf16x8 wf[NCB][2];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      F8 f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tap = ks * 16 + 8 * h + j, c = cb * 32 + crow;
        f.v[j] = (tap < 27 && c < a.C) ? a.w[c * 27 + (FLIP ? 26 - tap : tap)] : 0.f;
      }
      wf[cb][ks] = __builtin_bit_cast(bf16x8, pack8(f));
    }
  // halo offsets of this lane's 8 taps per k-step, relative to the voxel's own halo position
  int toff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = ks * 16 + 8 * h + j, t2 = tap < 27 ? tap : 13;
      toff[ks][j] = (t2 / 9) * (kHH * kHW) + ((t2 / 3) % 3) * kHW + t2 % 3;
    }
  bf16 sreg[5];
  auto issue = [&](int tile) {
    int r = tile;
    const int tx = r % tw; r /= tw;
    const int ty = r % th; r /= th;
    const int tz = r % td, n = r / td;
    const int z = tz * kTD + wave, y0 = ty * kTH, x0 = tx * kTW;
    const bf16* sn = a.s + (int64_t)n * a.D * a.H * a.W * a.s_cs;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int hi = i * 64 + lane;
      const int hx = hi % kHW, hy = (hi / kHW) % kHH, hz = hi / (kHW * kHH);
      const int zz = z + hz - 1, yy = y0 + hy - 1, xx = x0 + hx - 1;
      const bool in = hi < HS && (unsigned)zz < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      sreg[i] = in ? sn[(((int64_t)zz * a.H + yy) * a.W + xx) * a.s_cs] : f2bf(0.f);
    }
  };
  // GroupNorm sums of the rounded output (the network's input conv feeds the first norm1 and, as a skip, the last concatenation:
  // 74 + 66 us of statistics passes per C4 step).  A lane's channels are fixed (16 h + e); per image: fold the 32 lanes of a half
  // wave, the 4 waves through LDS, one entry per workgroup.
  constexpr bool ST = NCB == 1 && FLIP == 0;
  __shared__ float sred[4][2][32];
  float sa[16], sq[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) sa[e] = sq[e] = 0.f;
  int sn = -1;
  auto stats_flush = [&](int img) {  // workgroup-uniform call sites
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) {
        sa[e] += __shfl_xor(sa[e], m, 64);
        sq[e] += __shfl_xor(sq[e], m, 64);
      }
    }
    __syncthreads();
    if (v == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { sred[wave][0][16 * h + e] = sa[e]; sred[wave][1][16 * h + e] = sq[e]; }
    }
    __syncthreads();
    if (threadIdx.x < 32 && (int)threadIdx.x < a.C) {
      const int c = threadIdx.x;
      const float s0 = (sred[0][0][c] + sred[1][0][c]) + (sred[2][0][c] + sred[3][0][c]);
      const float s1 = (sred[0][1][c] + sred[1][1][c]) + (sred[2][1][c] + sred[3][1][c]);
      *(float2*)(a.stats + (((int64_t)img * a.C + c) * a.stats_chunks + blockIdx.x) * 2) = make_float2(s0, s1);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) sa[e] = sq[e] = 0.f;
  };
  if (ST && a.stats && threadIdx.x < 32 && (int)threadIdx.x < a.C)  // (a workgroup may not see tiles of every image)
    for (int img = 0; img < a.N; ++img) *(float2*)(a.stats + (((int64_t)img * a.C + threadIdx.x) * a.stats_chunks + blockIdx.x) * 2) = make_float2(0.f, 0.f);
  int tile = blockIdx.x, buf = 0;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    bf16* hb = sh[wave][buf];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int hi = i * 64 + lane;
      if (hi < HS) hb[hi] = sreg[i];
    }
    int r = tile;
    const int tx = r % tw; r /= tw;
    const int ty = r % th; r /= th;
    const int tz = r % td, n = r / td;
    const int z = tz * kTD + wave;
    if (ST && a.stats && n != sn) {  // (the image index never decreases along a workgroup's walk)
      if (sn >= 0) stats_flush(sn);
      sn = n;
    }
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {  // 32 voxels: rows 4 blk .. 4 blk + 3 of this wave's d-slice
      const int row = 4 * blk + (v >> 3), col = v & 7;
      const int base = row * kHW + col;  // halo position of (slice 0, row, col); tap (1,1,1) is the voxel itself
      f32x16 acc[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int c = cb * 32 + 16 * h + e;
          acc[cb][e] = (a.addvec && c < a.C) ? a.addvec[(int64_t)n * a.av_stride + c] : 0.f;
        }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 bfr;
#pragma unroll
        for (int j = 0; j < 8; ++j) bfr[j] = (ks * 16 + 8 * h + j) < 27 ? hb[base + toff[ks][j]] : f2bf(0.f);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][ks], bfr, acc[cb], 0, 0, 0);
      }
      const int y = ty * kTH + row, x = tx * kTW + col;
      if (z < a.D && y < a.H && x < a.W) {
        bf16* out = a.y + ((((int64_t)n * a.D + z) * a.H + y) * a.W + x) * a.y_cs;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int c0 = cb * 32 + 16 * h;
          F8 lo, hi;
#pragma unroll
          for (int e = 0; e < 8; ++e) { lo.v[e] = acc[cb][e]; hi.v[e] = acc[cb][8 + e]; }
          const u32x4 plo = pack8(lo), phi = pack8(hi);
          if (c0 < a.C) *(u32x4*)(out + c0) = plo;
          if (c0 + 8 < a.C) *(u32x4*)(out + c0 + 8) = phi;
          if (ST && a.stats) {  // statistics of the ROUNDED values: what the consumer's GroupNorm will read (channels >= C hold zeros)
            const F8 rl = unpack8(plo), rh = unpack8(phi);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              sa[e] += rl.v[e]; sq[e] = fmaf(rl.v[e], rl.v[e], sq[e]);
              sa[8 + e] += rh.v[e]; sq[8 + e] = fmaf(rh.v[e], rh.v[e], sq[8 + e]);
            }
          }
        }
      }
    }
  }
  if (ST && a.stats && sn >= 0) stats_flush(sn);
}

// ---------------------------------------------------------------------------------------------- C -> 1 forward
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

// ---------------------------------------------------------------------------------------------- weight gradient of both
// dW[c][t] += sum_v m[v][c] * s[v + d(t)],  d(t) = tap offset (FLIP: -offset: the C -> 1 conv, where m = x and s = dy);
// column 27 of the product is m against a column of ones: sum_v m[v][c] (the bias gradient of the 1 -> C conv);
// sum_v s[v] (the bias gradient of the C -> 1 conv) is reduced on the side.  Persistent workgroups: 4 waves, wave w owns d-slice w
// of every 4x8x8 tile (4 k-steps of 16 voxels = 2 rows of 8), accumulators live across the workgroup's tiles; per-workgroup slabs
// [C x 32 + 1] fp32 are folded by k_c1_wgrad_reduce (512 workgroups adding into the same 1 K addresses with atomics serialise).
struct C1WgArgs {
  int N, D, H, W, C;
  const bf16* m; int m_cs;   // C-channel tensor
  const bf16* s; int s_cs;   // single-channel tensor
  float* part;               // [gridDim.x][NCB * 1024 + 32] slabs
  int ntiles, flip;
};
// Everything a wave touches in the loop is wave-private (its d-slice of m: 64 voxels x NCB*32 channels, and the 3 halo slices of s
// it needs), so there is no barrier in the loop (LDS operations of one wave complete in order) and the four waves drift apart,
// which is what hides the HBM latency; the next tile's loads are issued before the current tile's MFMAs.
template <int NCB>
__global__ void __launch_bounds__(256) k_c1_wgrad(C1WgArgs a) {
  constexpr int PITCH = NCB * 64, MT = 64 * PITCH, HS = 3 * kHH * kHW, HSB = (HS * 2 + 15) / 16 * 16, WV = 2 * MT + 2 * HSB;
  constexpr int PVMAX = NCB * 4;  // 16-byte pieces per voxel
  extern __shared__ __attribute__((aligned(16))) char c1w_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = lane & 31, h = lane >> 5;
  char* my = c1w_lds + wave * WV;
  const int tw = (a.W + kTW - 1) / kTW, th = (a.H + kTH - 1) / kTH, td = (a.D + kTD - 1) / kTD;
  const int PV = a.C >> 3;
  const int tt = a.flip ? 26 - t : t;  // tap of this lane's B column (27 = ones, 28.. = zeros)
  const int kd = tt / 9, kh = (tt / 3) % 3, kw = tt % 3;
  for (int i = lane; i < 2 * MT / 16; i += 64) ((u32x4*)my)[i] = u32x4{0u, 0u, 0u, 0u};  // channels >= C of a ragged C stay zero
  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
  float s_acc = 0.f;

  u32x4 mreg[PVMAX];
  bf16 sreg[5];
  auto issue = [&](int tile) {  // this wave's loads of `tile`: m pieces (coalesced: consecutive lanes = consecutive 16 bytes) and s halo
    int r = tile;
    const int tx = r % tw; r /= tw;
    const int ty = r % th; r /= th;
    const int tz = r % td, n = r / td;
    const int z = tz * kTD + wave, y0 = ty * kTH, x0 = tx * kTW;
    const bf16* mn = a.m + (int64_t)n * a.D * a.H * a.W * a.m_cs;
    const bf16* sn = a.s + (int64_t)n * a.D * a.H * a.W * a.s_cs;
#pragma unroll
    for (int i = 0; i < PVMAX; ++i) {
      const int idx = i * 64 + lane, v = idx / PV, p = idx - v * PV;
      const int y = y0 + (v >> 3), x = x0 + (v & 7);
      const bool in = i < PV && v < 64 && z < a.D && y < a.H && x < a.W;
      mreg[i] = in ? *(const u32x4*)(mn + (((int64_t)z * a.H + y) * a.W + x) * a.m_cs + p * 8) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int hi = i * 64 + lane;  // halo index inside this wave's 3 slices
      const int hx = hi % kHW, hy = (hi / kHW) % kHH, hz = hi / (kHW * kHH);
      const int zz = z + hz - 1, yy = y0 + hy - 1, xx = x0 + hx - 1;
      const bool in = hi < HS && (unsigned)zz < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      sreg[i] = in ? sn[(((int64_t)zz * a.H + yy) * a.W + xx) * a.s_cs] : f2bf(0.f);
    }
  };
  int tile = blockIdx.x, buf = 0;
  if (tile < a.ntiles) issue(tile);
  for (; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
    char* mt = my + buf * MT;
    bf16* sh = (bf16*)(my + 2 * MT + buf * HSB);
#pragma unroll
    for (int i = 0; i < PVMAX; ++i) {
      const int idx = i * 64 + lane, v = idx / PV, p = idx - v * PV;
      if (i < PV && v < 64) *(u32x4*)(mt + v * PITCH + p * 16) = mreg[i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int hi = i * 64 + lane;
      if (hi < HS) {
        sh[hi] = sreg[i];
        const int hx = hi % kHW, hy = (hi / kHW) % kHH, hz = hi / (kHW * kHH);
        if (hz == 1 && hy >= 1 && hy <= kTH && hx >= 1 && hx <= kTW) s_acc += bf2f(sreg[i]);  // this wave's own d-slice
      }
    }
    if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);  // in flight under this tile's MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // k-step: 16 voxels = (row 2q + h, col j), j = 0..7
      const int v0 = (2 * q + h) * 8;
      bf16x8 bfr;
      const int hb = (kd * kHH + 2 * q + h + kh) * kHW + kw;
#pragma unroll
      for (int j = 0; j < 8; ++j) bfr[j] = t < 27 ? sh[hb + j] : (t == 27 ? f2bf(1.f) : f2bf(0.f));
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        bf16x8 af;  // A[row c = cb*32 + t][k = 8h + j] = m[voxel v0 + j][c]
#pragma unroll
        for (int j = 0; j < 8; ++j) af[j] = *(const bf16*)(mt + (v0 + j) * PITCH + (cb * 32 + t) * 2);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[cb], 0, 0, 0);
      }
    }
  }
  // fold the 4 waves through LDS (the tile buffers are dead), write this workgroup's slab: part[c * 32 + t]
  __syncthreads();
  float* red = (float*)c1w_lds;  // [4][NCB * 1024] + [4]
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave * NCB * 1024 + cb * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + t] = acc[cb][e];
  s_acc = wave_sum(s_acc);
  if (lane == 0) red[4 * NCB * 1024 + wave] = s_acc;
  __syncthreads();
  float* out = a.part + (int64_t)blockIdx.x * (NCB * 1024 + 32);
  for (int i = threadIdx.x; i < NCB * 1024; i += 256)
    out[i] = (red[i] + red[NCB * 1024 + i]) + (red[2 * NCB * 1024 + i] + red[3 * NCB * 1024 + i]);
  if (threadIdx.x == 0) out[NCB * 1024] = (red[4 * NCB * 1024] + red[4 * NCB * 1024 + 1]) + (red[4 * NCB * 1024 + 2] + red[4 * NCB * 1024 + 3]);
}
// dw[c][t] += sum_slabs part[c * 32 + t] (t < 27);  colsum_m[c] += column 27;  colsum_s[0] += the s sums
__global__ void __launch_bounds__(256) k_c1_wgrad_reduce(const float* __restrict__ part, int nslab, int slab, int C, float* __restrict__ dw,
                                                         float* __restrict__ colsum_m, float* __restrict__ colsum_s) {
  // 32 outputs per block, the slabs dealt over 8 thread groups (one thread walking all 512 slabs is 128 dependent load latencies)
  __shared__ float red[8][32];
  const int i = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  const bool live = i <= C * 32;
  const int idx = i < C * 32 ? i : slab - 32;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (live) {
    int k = g;
    for (; k + 24 < nslab; k += 32) {
      s0 += part[(int64_t)k * slab + idx];
      s1 += part[(int64_t)(k + 8) * slab + idx];
      s2 += part[(int64_t)(k + 16) * slab + idx];
      s3 += part[(int64_t)(k + 24) * slab + idx];
    }
    for (; k < nslab; k += 8) s0 += part[(int64_t)k * slab + idx];
  }
  red[g][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0 || !live) return;
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) v += red[j][threadIdx.x];
  if (i == C * 32) { if (colsum_s) colsum_s[0] += v; return; }
  const int c = i >> 5, tp = i & 31;
  if (tp < 27) dw[c * 27 + tp] += v;
  else if (tp == 27 && colsum_m) colsum_m[c] += v;
}

bool c1_dims_ok(int N, int D, int H, int W, int C) {
  return N > 0 && D > 0 && H > 0 && W > 0 && C >= 8 && C <= 64 && (C & 7) == 0 && (int64_t)N * D * H * W < (1ll << 31);
}

}  // namespace

static int c1_use_mfma() {
  static const int v = [] { const char* e = getenv("MI_C1_EXPAND_MFMA"); return e ? atoi(e) : 1; }();
  return v;
}
// chunks of the statistics the 1 -> C forward emits (0: this shape does not emit them)
int mi_c1_expand_stats_chunks(int N, int D, int H, int W, int C) {
  static const int on = [] { const char* e = getenv("MI_C1_STATS"); return e ? atoi(e) : 1; }();  // 0: separate statistics pass (A/B runs)
  if (!on || !c1_use_mfma() || !c1_dims_ok(N, D, H, W, C) || C > 32 || N > 16) return 0;
  const int64_t tiles = (int64_t)N * ((D + kTD - 1) / kTD) * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
  if (tiles >= (1ll << 31)) return 0;
  return (int)(tiles < 2048 ? tiles : 2048);
}
int mi_launch_c1_expand(const void* s, int s_cs, const float* w, const float* addvec, int av_stride, void* y, int y_cs, int N, int D, int H,
                        int W, int C, int flip, hipStream_t st, float* stats) {
  if (!c1_dims_ok(N, D, H, W, C) || (y_cs & 7)) return MI_ERR_UNSUPPORTED;
  C1Args a{};
  a.N = N; a.D = D; a.H = H; a.W = W; a.C = C;
  a.s = (const bf16*)s; a.s_cs = s_cs; a.w = w; a.addvec = addvec; a.av_stride = av_stride; a.y = (bf16*)y; a.y_cs = y_cs;
  a.stats = stats; a.stats_chunks = stats ? mi_c1_expand_stats_chunks(N, D, H, W, C) : 0;
  if (stats && (flip || a.stats_chunks == 0)) return MI_ERR_UNSUPPORTED;
  const int use_mfma = c1_use_mfma();
  const int64_t tiles = (int64_t)N * ((D + kTD - 1) / kTD) * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
  if (use_mfma && tiles < (1ll << 31)) {
    const int grid = (int)(tiles < 2048 ? tiles : 2048);  // persistent: 8 workgroups of 4 waves per CU
    const int ncb = (C + 31) / 32;
    if (ncb == 1) {
      if (flip) hipLaunchKernelGGL((k_c1_expand_mfma<1, 1>), dim3(grid), dim3(256), 0, st, a, (int)tiles);
      else hipLaunchKernelGGL((k_c1_expand_mfma<1, 0>), dim3(grid), dim3(256), 0, st, a, (int)tiles);
    } else {
      if (flip) hipLaunchKernelGGL((k_c1_expand_mfma<2, 1>), dim3(grid), dim3(256), 0, st, a, (int)tiles);
      else hipLaunchKernelGGL((k_c1_expand_mfma<2, 0>), dim3(grid), dim3(256), 0, st, a, (int)tiles);
    }
    MI_CHECK_LAUNCH();
    return 0;
  }
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

// Weight gradient of the 1 -> C conv (flip = 0: m = dy, s = x; colsum_m = the bias gradient [C]) or of the C -> 1 conv (flip = 1:
// m = x, s = dy; colsum_s = the bias gradient [1]).  dw: fp32 [C][27] torch layout (either conv), accumulated.  part: scratch of
// mi_c1_wgrad_scratch_floats(C) floats.
int mi_c1_wgrad_slabs() { return 512; }
int64_t mi_c1_wgrad_scratch_floats(int C) { return (int64_t)mi_c1_wgrad_slabs() * (((C + 31) / 32) * 1024 + 32); }
int mi_launch_c1_wgrad(const void* m, int m_cs, const void* s1, int s_cs, float* dw, float* colsum_m, float* colsum_s, float* part, int N, int D,
                       int H, int W, int C, int flip, hipStream_t st) {
  if (!c1_dims_ok(N, D, H, W, C) || !part) return MI_ERR_UNSUPPORTED;
  const int64_t tiles = (int64_t)N * ((D + kTD - 1) / kTD) * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
  if (tiles >= (1ll << 31)) return MI_ERR_UNSUPPORTED;
  C1WgArgs a{};
  a.N = N; a.D = D; a.H = H; a.W = W; a.C = C;
  a.m = (const bf16*)m; a.m_cs = m_cs; a.s = (const bf16*)s1; a.s_cs = s_cs; a.part = part; a.ntiles = (int)tiles; a.flip = flip;
  const int ncb = (C + 31) / 32;
  int grid = mi_c1_wgrad_slabs();
  if (grid > tiles) grid = (int)tiles;
  {
    const int pitch = ncb * 64, mt = 64 * pitch, hsb = (3 * kHH * kHW * 2 + 15) / 16 * 16;
    size_t lds = 4 * (size_t)(2 * mt + 2 * hsb);
    const size_t need_red = sizeof(float) * (size_t)(4 * ncb * 1024 + 4);
    if (lds < need_red) lds = need_red;
    static bool attr_done[2] = {false, false};
    auto kern = ncb == 1 ? k_c1_wgrad<1> : k_c1_wgrad<2>;
    if (!attr_done[ncb - 1]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_done[ncb - 1] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  }
  const int slab = ncb * 1024 + 32;
  hipLaunchKernelGGL(k_c1_wgrad_reduce, dim3((C * 32 + 1 + 31) / 32), dim3(256), 0, st, part, grid, slab, C, dw, colsum_m, colsum_s);
  MI_CHECK_LAUNCH();
  return 0;
}
