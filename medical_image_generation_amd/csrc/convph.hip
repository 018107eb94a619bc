// Factor-2 phase convolutions for gfx950 on the role-specialised LDS-DMA kernel of conv27.hip (conv27_kernel.h, MODE 1 / 2).
//
// Two layer types of the U-Net / autoencoder "with strides" are 3x3x3 convolutions across a change of resolution by 2:
//   * Upsample.forward (UNet:569-588, AEKL Upsample): nearest x2 interpolation, then a k3 s1 p1 conv on the fine grid.  Every fine voxel
//     2i + p (phase p in {0,1}^3) only ever sees the 2x2x2 coarse voxels i + p + t - 1, each through the SUM of the taps that land on it:
//         y[2i + p] = sum_{t in {0,1}^3} A[p][t] x[i + p + t - 1],     A[p][t] = sum of W[k] over k in S(p_d,t_d) x S(p_h,t_h) x S(p_w,t_w),
//         S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
//     -- 8 taps per output voxel instead of 27 and no up-sampled tensor in HBM (zero padding of the fine tensor = zero padding of the
//     coarse one).  The reference counts 27 taps on the fine grid (torch.utils.flop_counter); that stays the flop convention.
//   * Downsample.forward (UNet:524-531, AEKL Downsample): k3 s2 p1 conv, y[o] = sum_k W[k] x[2o + k - 1]; by input class c (parity of the
//     fine coordinate) y[i] += sum_t B[c][t] x[2(i + (1 - c) + t - 1) + c] with B[1][0] = W[0], B[1][1] = W[2], B[0][0] = W[1], B[0][1] = 0.
// Both come in two directions that are each other's transposes, so two kernel modes serve four operations:
//   MODE 1 "scatter" (tile grid = coarse input, phases of the fine output): Upsample+conv forward; data gradient of the k3 s2 conv
//   MODE 2 "gather"  (tile grid = coarse output, classes of the fine input): k3 s2 conv forward; data gradient of Upsample+conv
// An image of the kernel is (tile, phase or class, 32-channel chunk) with the 8 taps of a 2x2x2 box; in MODE 1 the halo image of a
// (tile, chunk) serves all 8 phases (one fetch when the layer has one or two chunks), in MODE 2 the LDS-DMA gathers the class's
// sub-lattice (space-to-depth done by the DMA's per-lane source address: no re-laid-out copy of the tensor in HBM).
// Weights: fp32 master [Cout][Cin][27] -> bf16 A fragments [cout group][phase][chunk][tap][k-step][cout block], each the sum over its
// tap set in fp32, rounded once (k_pack_phase).
#include "conv27_kernel.h"

namespace {

template <int NCB, int MODE>
__global__ void __launch_bounds__(512, 2) k_convph(ConvArgs a) {
  conv27_body<NCB, 0, MODE>(a);
}

template <int NCB, int MODE>
int launchph(ConvArgs a, int ntiles, int ny, hipStream_t st) {
  using KK = K<NCB, MODE>;
  a.ntiles = ntiles;
  a.dbg = 0;
  a.stats = nullptr; a.stats_chunks = 0;
  const int gx = mi_conv27_grid_x(ntiles, ny);
  auto kern = k_convph<NCB, MODE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  static_assert(KK::LDS_TOTAL <= 160 * 1024, "LDS budget");
  hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), (size_t)KK::LDS_TOTAL, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}

// One fragment per 64 threads: fragment f = ((y * 8 + pc) * nchunks + ch) * (8 * 2 * NCB) + (t * 2 + ks) * NCB + cb holds
// A[row -> kernel-out channel (y * NCB + cb) * 32 + perm(rho)][k -> kernel-in channel ch * 32 + ks * 16 + 8h + j] = the sum of the
// torch weights W[o][i][tap] over the taps of masks[pc * 8 + t] (bit k = tap k of the 27); tr: kernel-out = torch-in (data gradients).
__global__ void __launch_bounds__(256) k_pack_phase(const float* __restrict__ w, u32x4* __restrict__ out, const unsigned* __restrict__ masks,
                                                    int nfrags, int NCB, int nchunks, int Ko, int Ki, int Co_t, int Ci_t, int tr) {
  const int gid = blockIdx.x * 256 + threadIdx.x, f = gid >> 6, lane = gid & 63;
  if (f >= nfrags) return;
  int r = f;
  const int cb = r % NCB; r /= NCB;
  const int ks = r & 1; r >>= 1;
  const int t = r & 7; r >>= 3;
  const int ch = r % nchunks; r /= nchunks;
  const int pc = r & 7, y = r >> 3;
  const unsigned m = masks[pc * 8 + t];
  const int rho = lane & 31;
  const int crow = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);  // conv27 row permutation: a lane's 16 accumulators = 16 consecutive channels
  const int ko = (y * NCB + cb) * 32 + crow, ki0 = ch * 32 + ks * 16 + (lane >> 5) * 8;
  F8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ki = ki0 + j;
    const int o = tr ? ki : ko, i = tr ? ko : ki;  // torch (out, in)
    float s = 0.f;
    if (ko < Ko && ki < Ki && o < Co_t && i < Ci_t) {
      const float* p = w + ((int64_t)o * Ci_t + i) * 27;
      for (unsigned mm = m; mm; mm &= mm - 1) s += p[__builtin_ctz(mm)];
    }
    v.v[j] = s;
  }
  out[gid] = pack8(v);
}

}  // namespace

int mi_launch_convph(const ConvArgs& a, int NCB, int mode, int ntiles, int ny, hipStream_t st) {
  if (a.g.TD != 4 || a.g.TH != 8 || a.g.TW != 8 || (a.x_cs & 7) || (a.Cin & 7) || (a.y_cs & 7) || (a.Cout & 7) || a.y_bytes == 0 || a.res) return MI_ERR_BAD_ARG;
  if (mode == 1) return NCB == 2 ? launchph<2, 1>(a, ntiles, ny, st) : launchph<1, 1>(a, ntiles, ny, st);
  if (mode == 2) return NCB == 2 ? launchph<2, 2>(a, ntiles, ny, st) : launchph<1, 2>(a, ntiles, ny, st);
  return MI_ERR_BAD_ARG;
}

int mi_launch_pack_phase(const float* w, void* out, const unsigned* d_masks, int nfrags, int NCB, int nchunks, int Ko, int Ki, int Co_t, int Ci_t,
                         int tr, hipStream_t st) {
  if (!w || !out || !d_masks || nfrags <= 0) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_pack_phase, dim3((nfrags * 64 + 255) / 256), dim3(256), 0, st, w, (u32x4*)out, d_masks, nfrags, NCB, nchunks, Ko, Ki, Co_t,
                     Ci_t, tr);
  MI_CHECK_LAUNCH();
  return 0;
}
