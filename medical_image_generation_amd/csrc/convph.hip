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
  static const int dbg = mi_diag_knob("MI_CPH_DBG") & 1;  // ablation knob: 1 = no store epilogue (results are garbage)
  a.dbg = dbg;
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

// Weight pack.  Fragment f = ((y * 8 + pc) * nchunks + ch) * (8 * 2 * NCB) + (t * 2 + ks) * NCB + cb holds
// A[row -> kernel-out channel (y * NCB + cb) * 32 + perm(rho)][k -> kernel-in channel ch * 32 + ks * 16 + 8h + j] = the sum of the torch
// weights W[o][i][tap] over the taps of masks[pc * 8 + t] (bit k = tap k of the 27); tr: kernel-out = torch-in (data gradients).
// A block owns one (32 kernel-out, 16 kernel-in) block: its 32 x 16 x 27 source weights are read once, coalesced (rows of 432 resp.
// 864 contiguous floats), into LDS and all 64 (phase, tap) fragments are summed from there -- the per-fragment gather from global
// memory (8 floats per lane at a 108-byte stride, up to 8 taps each) took 150 us for a 256 -> 256 layer, more than the kernel it feeds.
// jobs: every phase side of a network in ONE launch (blockIdx.x runs over the jobs' blocks; 12 launches of ~17 us each were 0.2 ms / step)
__global__ void __launch_bounds__(256) k_pack_phase(const PhasePackJob* __restrict__ jobs, int njobs) {
  extern __shared__ float sm[];
  int ji = 0;
  while (ji + 1 < njobs && (int)blockIdx.x >= jobs[ji + 1].block0) ++ji;
  const PhasePackJob jb = jobs[ji];
  const float* __restrict__ w = jb.w;
  u32x4* __restrict__ out = (u32x4*)jb.out;
  const unsigned* __restrict__ masks = jb.masks;
  const int NCB = jb.NCB, nchunks = jb.nchunks, Ko = jb.Ko, Ki = jb.Ki, Co_t = jb.Co_t, Ci_t = jb.Ci_t, tr = jb.tr;
  const int bx = blockIdx.x - jb.block0;
  const int kb = bx / (nchunks * 2), kk = bx % (nchunks * 2);
  const int y = kb / NCB, cb = kb % NCB, ch = kk >> 1, ks = kk & 1;
  const int ko0 = kb * 32, ki0 = kk * 16;
  const int rows = tr ? 16 : 32, ni = tr ? 32 : 16;         // torch rows (out channels) / in channels of the block
  const int o0 = tr ? ki0 : ko0, i0 = tr ? ko0 : ki0;
  const int run = ni * 27, pitch = run + 1;
  const int nvalid = (Ci_t - i0 < ni ? (Ci_t - i0 > 0 ? Ci_t - i0 : 0) : ni) * 27;
  for (int e0 = threadIdx.x; e0 < rows * run; e0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256, o = e / run, c = e - o * run;
      v[u] = (e < rows * run && o0 + o < Co_t && c < nvalid) ? w[((int64_t)(o0 + o) * Ci_t + i0) * 27 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256, o = e / run, c = e - o * run;
      if (e < rows * run) sm[o * pitch + c] = v[u];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rho = lane & 31, h = lane >> 5;
  const int crow = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);  // conv27 row permutation: a lane's 16 accumulators = 16 consecutive channels
  const bool row_ok = ko0 + crow < Ko;
  for (int c = blockIdx.y * 32 + wave; c < blockIdx.y * 32 + 32; c += 4) {  // blockIdx.y = half of the phases: 32 (phase, tap) over the 4 waves
    const int pc = c >> 3, t = c & 7;
    const unsigned m = masks[c];
    F8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kil = 8 * h + j;
      const int o = tr ? kil : crow, i = tr ? crow : kil;
      float sacc = 0.f;
      if (row_ok && ki0 + kil < Ki)
        for (unsigned mm = m; mm; mm &= mm - 1) sacc += sm[o * pitch + i * 27 + __builtin_ctz(mm)];
      v.v[j] = sacc;
    }
    const int f = ((y * 8 + pc) * nchunks + ch) * (16 * NCB) + (t * 2 + ks) * NCB + cb;
    out[(int64_t)f * 64 + lane] = pack8(v);
  }
}

}  // namespace

int mi_launch_convph(const ConvArgs& a, int NCB, int mode, int ntiles, int ny, hipStream_t st) {
  if (a.g.TD != 4 || a.g.TH != 8 || a.g.TW != 8 || (a.x_cs & 7) || (a.Cin & 7) || (a.y_cs & 7) || (a.Cout & 7) || a.y_bytes == 0 || a.res) return MI_ERR_BAD_ARG;
  if (mode == 1) return NCB == 2 ? launchph<2, 1>(a, ntiles, ny, st) : launchph<1, 1>(a, ntiles, ny, st);
  if (mode == 2) return NCB == 2 ? launchph<2, 2>(a, ntiles, ny, st) : launchph<1, 2>(a, ntiles, ny, st);
  return MI_ERR_BAD_ARG;
}

int mi_pack_phase_blocks(int nfrags, int NCB, int nchunks) { (void)NCB; (void)nchunks; return nfrags / 64; }  // = ny * NCB * nchunks * 2

int mi_launch_pack_phase(const PhasePackJob* d_jobs, int njobs, int total_blocks, hipStream_t st) {
  if (!d_jobs || njobs <= 0 || total_blocks <= 0) return MI_ERR_BAD_ARG;
  const size_t lds = sizeof(float) * (size_t)(16 * (32 * 27 + 1) > 32 * (16 * 27 + 1) ? 16 * (32 * 27 + 1) : 32 * (16 * 27 + 1));
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)k_pack_phase, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(k_pack_phase, dim3(total_blocks, 2), dim3(256), lds, st, d_jobs, njobs);
  MI_CHECK_LAUNCH();
  return 0;
}
