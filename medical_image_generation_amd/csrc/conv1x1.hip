// 1x1x1 convolution (forward and data gradient) as a streaming GEMM over voxels: the ResnetBlock skip connections
// (UNet:664-672; AEKL nin_shortcut 175-187) and the KL-VAE's quant / post-quant convs (AEKL:723-749).
//
// HBM-bound (a 96->32 skip conv at 128^3 moves 537 MB for 13 GFLOP), so there is no halo image and no LDS staging of the
// activations: the MFMA B fragment of a 32-voxel block -- lane (voxel r, half h) needs channels 16 ks + 8 h .. + 7 of voxel r,
// 16 contiguous bytes -- is loaded straight from global memory (buffer_load with range check: ragged tail and channel padding
// read zeros), double-buffered in registers one block ahead.  The packed weight fragments (a few KiB) sit in LDS for the whole
// kernel.  Output channels use the conv27 row permutation, so a lane holds 16 consecutive channels of its voxel and writes
// 32 contiguous bytes; the two lanes of a voxel complete its 64-byte segment.
#include <stdlib.h>

#include "conv_common.h"
#include "medimgen_hip.h"

namespace {

template <int NCH>  // 32-channel input chunks (K = 32 * NCH)
__global__ void __launch_bounds__(256) k_conv1x1(ConvArgs a, int ncb_total, int NCB, int nfrags, int64_t nvox, int64_t vox_per_image, int stage) {
  constexpr int K16 = 2 * NCH;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // stage: the block's [32 voxels][Cout] bf16 tile goes through a wave-private LDS tile and leaves as lane-linear 16-byte pieces
  // (consecutive lanes = consecutive bytes of a voxel, then the next voxel).  Straight from the accumulators a store instruction
  // writes 64 separate 16-byte fragments (lane = (voxel, half), 32 bytes apart, voxel pitch = Cout * 2): measured 2.8 TB/s on the
  // write-heavy 32 -> 96 data gradient of a skip conv where the forward direction (96 -> 32) streams at 5.1.
  const int tile_pitch = a.Cout * 2 + 16;  // bytes per voxel row of the tile (+16: rows start 4 banks apart)
  char* tile = lds + (size_t)nfrags * 1024 + (size_t)wave * 32 * tile_pitch;
  const int r = lane & 31, h = lane >> 5;
  // packed fragments -> LDS, same order: [cout group y][chunk][ks][cb], 1 KiB each
  for (int i = threadIdx.x; i < nfrags * 64; i += 256) ((u32x4*)lds)[i] = a.wpk[i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, a.y_bytes);
  const bool vec_ok = (a.y_cs & 7) == 0 && (a.Cout & 7) == 0 && a.y_bytes != 0;
  const int64_t nblocks = (nvox + 31) >> 5;
  const int64_t bstep = (int64_t)gridDim.x * 4;
  int64_t blk = (int64_t)blockIdx.x * 4 + wave;
  auto load_block = [&](u32x4 (&b)[K16], int64_t bk) {
    const int64_t v = bk * 32 + r;
#pragma unroll
    for (int kk = 0; kk < K16; ++kk) {
      const int c = kk * 16 + 8 * h;
      const bool ok = (v < nvox) & (c + 8 <= a.Cin);
      const unsigned off = ok ? (unsigned)((v * a.x_cs + c) * 2) : 0xfffffff0u;
      b[kk] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
    }
  };
  u32x4 bcur[K16], bnext[K16];
  if (blk < nblocks) load_block(bcur, blk);
  for (; blk < nblocks; blk += bstep) {
    if (blk + bstep < nblocks) load_block(bnext, blk + bstep);  // in flight under this block's MFMAs and stores
    const int64_t v = blk * 32 + r;
    const int64_t n = v / vox_per_image;
    for (int c32 = 0; c32 < ncb_total; ++c32) {
      const int yg = c32 / NCB, cb = c32 - yg * NCB;
      const int co = c32 * 32 + 16 * h;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = (a.addvec && v < nvox && co + e < a.Cout) ? a.addvec[n * a.addvec_stride + co + e] : 0.f;
#pragma unroll
      for (int kk = 0; kk < K16; ++kk) {
        const int frag = ((yg * NCH + (kk >> 1)) * 2 + (kk & 1)) * NCB + cb;
        const u32x4 af = *(const u32x4*)(lds + frag * 1024 + lane * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bcur[kk]), acc, 0, 0, 0);
      }
      F8 lo, hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) { lo.v[e] = acc[e]; hi.v[e] = acc[8 + e]; }
      if (stage) {
        if (co + 8 <= a.Cout) *(u32x4*)(tile + r * tile_pitch + co * 2) = pack8(lo);
        if (co + 16 <= a.Cout) *(u32x4*)(tile + r * tile_pitch + co * 2 + 16) = pack8(hi);
      } else if (vec_ok) {
        const bool in0 = (v < nvox) & (co + 8 <= a.Cout), in1 = (v < nvox) & (co + 16 <= a.Cout);
        const unsigned o0 = in0 ? (unsigned)((v * a.y_cs + co) * 2) : 0xfffffff0u;
        const unsigned o1 = in1 ? (unsigned)((v * a.y_cs + co + 8) * 2) : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pack8(lo)), ry, (int)o0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pack8(hi)), ry, (int)o1, 0, 0);
      } else if (v < nvox) {
        bf16* yp = a.y + v * a.y_cs + co;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (co + e < a.Cout) yp[e] = f2bf(acc[e]);
      }
    }
    if (stage) {  // (wave-private tile: this wave's LDS operations complete in order, no barrier)
      const int ppv = a.Cout >> 3;  // 16-byte pieces per voxel
      const int64_t v0 = blk * 32;
      for (int i = lane; i < 32 * ppv; i += 64) {
        const int row = i / ppv, col = i - row * ppv;
        const u32x4 piece = *(const u32x4*)(tile + row * tile_pitch + col * 16);
        const unsigned off = (v0 + row < nvox) ? (unsigned)(((v0 + row) * a.y_cs + col * 8) * 2) : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, piece), ry, (int)off, 0, 0);
      }
    }
#pragma unroll
    for (int kk = 0; kk < K16; ++kk) bcur[kk] = bnext[kk];
  }
}

template <int NCH>
int launch11(const ConvArgs& a, int ncb_total, int NCB, int nfrags, int64_t nvox, int64_t vpi, hipStream_t st) {
  size_t lds = (size_t)nfrags * 1024;
  // staged stores (see the kernel): whole channel octets, 16-byte aligned pitch, a tile per wave that fits beside the weights
  static const char* stage_env = getenv("MI_C11_STAGE");
  const size_t tile_bytes = (size_t)4 * 32 * (a.Cout * 2 + 16);
  const int stage = (!stage_env || atoi(stage_env)) && (a.y_cs & 7) == 0 && (a.Cout & 7) == 0 && a.y_bytes != 0 && a.Cout >= 16 && lds + tile_bytes <= 64 * 1024;
  if (stage) lds += tile_bytes;
  auto kern = k_conv1x1<NCH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  int64_t nblocks = (nvox + 31) / 32;
  int grid = (int)((nblocks + 3) / 4);
  if (grid > 256 * 6) grid = 256 * 6;  // ~6 workgroups (24 waves) per CU keep enough loads in flight
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, ncb_total, NCB, nfrags, nvox, vpi, stage);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// a: ConvArgs as filled for the table-driven kernel (x / Cin = tensor the loader reads, y / Cout = tensor produced, wpk packed with
// the perm16 row order); NCB = cout blocks per packed group, ny = groups.
int mi_launch_conv1x1(const ConvArgs& a, int NCB, int ny, hipStream_t st) {
  if ((a.x_cs & 7) || (a.Cin & 7) || a.res || a.ss) return MI_ERR_BAD_ARG;
  const int nch = a.nchunks;
  const int nfrags = ny * nch * 2 * NCB;
  if (nfrags * 1024 > 96 * 1024) return MI_ERR_UNSUPPORTED;
  const int64_t vpi = (int64_t)a.Do * a.Ho * a.Wo, nvox = vpi * a.N;
  const int ncb_total = (a.Cout + 31) / 32;
  switch (nch) {
    case 1: return launch11<1>(a, ncb_total, NCB, nfrags, nvox, vpi, st);
    case 2: return launch11<2>(a, ncb_total, NCB, nfrags, nvox, vpi, st);
    case 3: return launch11<3>(a, ncb_total, NCB, nfrags, nvox, vpi, st);
    case 4: return launch11<4>(a, ncb_total, NCB, nfrags, nvox, vpi, st);
    case 6: return launch11<6>(a, ncb_total, NCB, nfrags, nvox, vpi, st);
    default: return MI_ERR_UNSUPPORTED;
  }
}

// ------------------------------------------------------------------------------------------------ 1x1x1 weight gradient
// dW[co][ci] += sum_v dy[v][co] * x[v][ci]  (+ optional column sums of dy = bias gradient).  Also HBM-bound: x and dy are read
// exactly ONCE, with perfectly coalesced 16-byte loads of the flat [voxel][channel] arrays (register prefetch one tile ahead),
// and every (cout block, cin chunk) pair is accumulated by the same workgroup from the same LDS images.  A tile = 256 voxels;
// the images are stored per 32-channel chunk with 64-byte voxels so that a ds_read_b64_tr_b16 (4 voxels x 16 channels,
// delivered voxel-major = the MFMA k axis) covers every bank once.  Wave w owns k-steps 2w, 2w+1 of the tile's 16; the 8 waves'
// accumulators are folded through LDS at the end and added to dW with fp32 atomics (one 32x32 block per pair and workgroup).
namespace {

struct W11Args {
  const bf16* x; int x_cs; int Cin;
  const bf16* dy; int dy_cs; int Cout;
  float* dw;            // [Cout][Cin]
  float* colsum; int colsum_stride;
  int64_t nvox, vox_per_image;
  int ci0, co0;         // first input chunk / output block handled by this launch slice (blockIdx.y / z add to them)
};

__device__ __forceinline__ void tr_read8(u32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr));
}

template <int NCO, int NCI>
__global__ void __launch_bounds__(512, 2) k_wgrad1x1(W11Args p) {
  constexpr int IMG = 256 * 64;  // one 32-channel chunk image of a tile
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [NCI] x images, [NCO] dy images
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ci_base = (p.ci0 + blockIdx.y * NCI) * 32, co_base = (p.co0 + blockIdx.z * NCO) * 32;
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned l0 = (unsigned)(size_t)(lds_char*)lds;
  const int64_t ntiles = (p.nvox + 255) >> 8;
  // loader geometry: piece = 16 bytes = 8 channels of one voxel; a tile has 256 * 4 pieces per chunk image
  constexpr int XP = NCI * 1024 / 512, YP = NCO * 1024 / 512;  // pieces per thread
  u32x4 xr[XP], yr[YP];
  auto load_tile = [&](int64_t t) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int pc = threadIdx.x + 512 * i;             // piece index inside the tile: voxel-major over NCI*4 pieces per voxel
      const int v = pc / (NCI * 4), q = pc - v * (NCI * 4);
      const int64_t gv = t * 256 + v;
      const int c = ci_base + q * 8;
      u32x4 z = {0u, 0u, 0u, 0u};
      if (gv < p.nvox && c + 8 <= p.Cin) z = *(const u32x4*)(p.x + gv * p.x_cs + c);
      xr[i] = z;
    }
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      const int pc = threadIdx.x + 512 * i;
      const int v = pc / (NCO * 4), q = pc - v * (NCO * 4);
      const int64_t gv = t * 256 + v;
      const int c = co_base + q * 8;
      u32x4 z = {0u, 0u, 0u, 0u};
      if (gv < p.nvox && c + 8 <= p.Cout) z = *(const u32x4*)(p.dy + gv * p.dy_cs + c);
      yr[i] = z;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int pc = threadIdx.x + 512 * i;
      const int v = pc / (NCI * 4), q = pc - v * (NCI * 4);
      *(u32x4*)(lds + (q >> 2) * IMG + v * 64 + (q & 3) * 16) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      const int pc = threadIdx.x + 512 * i;
      const int v = pc / (NCO * 4), q = pc - v * (NCO * 4);
      *(u32x4*)(lds + (NCI + (q >> 2)) * IMG + v * 64 + (q & 3) * 16) = yr[i];
    }
  };
  f32x16 acc[NCO][NCI], cs[NCO];
#pragma unroll
  for (int o = 0; o < NCO; ++o) {
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[o][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCI; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[o][c][e] = 0.f;
  }
  const bool do_cs = p.colsum != nullptr && blockIdx.y == 0 && p.ci0 == 0;
  // transposed-read lane roles (as in conv.hip): 16-lane group gq -> channel half (gq & 1), voxel half (gq >> 1); lane 4q+p -> voxel q, channels 4p
  const int gq = lane >> 4, qv = (lane >> 2) & 3, pp = lane & 3;
  const unsigned lane_off = ((gq >> 1) * 8 + qv) * 64 + ((gq & 1) * 16 + pp * 4) * 2;  // k-step = 16 voxels: halves of 8
  int64_t t = blockIdx.x;
  if (t < ntiles) load_tile(t);
  int64_t cs_img = t < ntiles ? (t * 256) / p.vox_per_image : 0;
  auto cs_flush = [&](int64_t img) {
    if ((lane & 31) == 0) {
      const int hh = lane >> 5;
#pragma unroll
      for (int o = 0; o < NCO; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co_base + o * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (co < p.Cout) atomicAdd(p.colsum + img * p.colsum_stride + co, cs[o][e]);
        }
    }
#pragma unroll
    for (int o = 0; o < NCO; ++o)
#pragma unroll
      for (int e = 0; e < 16; ++e) cs[o][e] = 0.f;
  };
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();  // previous tile consumed
    store_tile();
    __syncthreads();
    if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);  // in flight under the MFMAs
    if (do_cs && p.colsum_stride != 0) {  // per-image sums: flush when the image changes (a tile never straddles images: host check)
      const int64_t img = (t * 256) / p.vox_per_image;
      if (img != cs_img) { cs_flush(cs_img); cs_img = img; }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned koff = (unsigned)((2 * wave + s) * 16 * 64) + lane_off;  // 16 voxels per k-step
      u32x2 fa[NCO][2], fb[NCI][2];
#pragma unroll
      for (int o = 0; o < NCO; ++o) {
        tr_read8(fa[o][0], l0 + (NCI + o) * IMG + koff);
        tr_read8(fa[o][1], l0 + (NCI + o) * IMG + koff + 4 * 64);
      }
#pragma unroll
      for (int c = 0; c < NCI; ++c) {
        tr_read8(fb[c][0], l0 + c * IMG + koff);
        tr_read8(fb[c][1], l0 + c * IMG + koff + 4 * 64);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int o = 0; o < NCO; ++o) asm volatile("" : "+v"(fa[o][0]), "+v"(fa[o][1]));
#pragma unroll
      for (int c = 0; c < NCI; ++c) asm volatile("" : "+v"(fb[c][0]), "+v"(fb[c][1]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < NCO; ++o) {
        const u32x4 ra = {fa[o][0][0], fa[o][0][1], fa[o][1][0], fa[o][1][1]};
#pragma unroll
        for (int c = 0; c < NCI; ++c) {
          const u32x4 rb = {fb[c][0][0], fb[c][0][1], fb[c][1][0], fb[c][1][1]};
          acc[o][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), acc[o][c], 0, 0, 0);
        }
        if (do_cs) {
          const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
          cs[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, ones), cs[o], 0, 0, 0);
        }
      }
    }
  }
  if (do_cs) {  // final column sums: fold the 8 waves through LDS first -- one atomic per channel and workgroup, not per wave
    float* csr = (float*)lds;  // [8][NCO * 32]
    __syncthreads();
    if ((lane & 31) == 0) {
      const int hh = lane >> 5;
#pragma unroll
      for (int o = 0; o < NCO; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) csr[wave * (NCO * 32) + o * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh] = cs[o][e];
    }
    __syncthreads();
    if (threadIdx.x < NCO * 32) {
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) sum += csr[wv * (NCO * 32) + threadIdx.x];
      const int co = co_base + threadIdx.x;
      const int64_t img = p.colsum_stride != 0 ? cs_img : 0;
      if (co < p.Cout) atomicAdd(p.colsum + img * p.colsum_stride + co, sum);
    }
  }
  // fold the 8 waves through LDS, pair by pair; D map: col = lane & 31 -> ci, row (e&3) + 8(e>>2) + 4h -> co
  float* red = (float*)lds;  // 8 x 1024 floats
#pragma unroll
  for (int o = 0; o < NCO; ++o)
#pragma unroll
    for (int c = 0; c < NCI; ++c) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[o][c][e];
      __syncthreads();
      for (int f = threadIdx.x; f < 1024; f += 512) {
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) sum += red[wv * 1024 + f];
        const int e = f >> 6, ln = f & 63;
        const int co = co_base + o * 32 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5), ci = ci_base + c * 32 + (ln & 31);
        if (co < p.Cout && ci < p.Cin) atomicAdd(p.dw + (int64_t)co * p.Cin + ci, sum);
      }
    }
}

// Wide layers (the latent UNet's q/k/v Linear: 512 -> 1536 over 16 k tokens).  k_wgrad1x1's eight waves all hold the workgroup's whole
// 64 x 64 block and split the voxels, so a launch re-reads its operands (Cout / 64 + Cin / 64) / 2 times from the Infinity Cache: 805 MB,
// 32 flop per byte, ~75 us.  Here the waves own SUB-BLOCKS of a 128 x 128 workgroup block (wave -> two cout blocks x one cin block,
// every wave walks all 16 k-steps of a tile): half the re-reads, no fold across waves in the epilogue.  Cin, Cout multiples of 128.
__global__ void __launch_bounds__(512, 1) k_wgrad1x1_wide(W11Args p) {
  constexpr int IMG = 256 * 64, NB = 4;  // 32-channel chunk image of a tile; 4 x-chunks + 4 dy-chunks = 128 KB
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ci_base = blockIdx.y * 128, co_base = blockIdx.z * 128;
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned l0 = (unsigned)(size_t)(lds_char*)lds;
  const int64_t ntiles = (p.nvox + 255) >> 8;
  constexpr int PP = NB * 1024 / 512;  // 16-byte pieces per thread and operand
  u32x4 xr[PP], yr[PP];
  auto load_tile = [&](int64_t t) {
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const int pc = threadIdx.x + 512 * i;  // voxel-major over the 16 pieces of a voxel
      const int v = pc >> 4, q = pc & 15;
      const int64_t gv = t * 256 + v;
      u32x4 z = {0u, 0u, 0u, 0u};
      xr[i] = gv < p.nvox ? *(const u32x4*)(p.x + gv * p.x_cs + ci_base + q * 8) : z;
      yr[i] = gv < p.nvox ? *(const u32x4*)(p.dy + gv * p.dy_cs + co_base + q * 8) : z;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const int pc = threadIdx.x + 512 * i;
      const int v = pc >> 4, q = pc & 15;
      *(u32x4*)(lds + (q >> 2) * IMG + v * 64 + (q & 3) * 16) = xr[i];
      *(u32x4*)(lds + (NB + (q >> 2)) * IMG + v * 64 + (q & 3) * 16) = yr[i];
    }
  };
  const int wo = wave >> 2, wc = wave & 3;  // cout blocks {2 wo, 2 wo + 1}, cin block wc
  f32x16 acc[2], cs[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[o][e] = cs[o][e] = 0.f;
  const bool do_cs = p.colsum != nullptr && blockIdx.y == 0 && wc == 0;  // (one column of waves sees every voxel of its cout blocks)
  const int gq = lane >> 4, qv = (lane >> 2) & 3, pp = lane & 3;
  const unsigned lane_off = ((gq >> 1) * 8 + qv) * 64 + ((gq & 1) * 16 + pp * 4) * 2;
  int64_t t = blockIdx.x;
  if (t < ntiles) load_tile(t);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();  // previous tile consumed
    store_tile();
    __syncthreads();
    if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);  // in flight under the MFMAs
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const unsigned koff = (unsigned)(ks * 16 * 64) + lane_off;
      u32x2 fa[2][2], fb[2];
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        tr_read8(fa[o][0], l0 + (NB + 2 * wo + o) * IMG + koff);
        tr_read8(fa[o][1], l0 + (NB + 2 * wo + o) * IMG + koff + 4 * 64);
      }
      tr_read8(fb[0], l0 + wc * IMG + koff);
      tr_read8(fb[1], l0 + wc * IMG + koff + 4 * 64);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int o = 0; o < 2; ++o) asm volatile("" : "+v"(fa[o][0]), "+v"(fa[o][1]));
      asm volatile("" : "+v"(fb[0]), "+v"(fb[1]));
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 rb = {fb[0][0], fb[0][1], fb[1][0], fb[1][1]};
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const u32x4 ra = {fa[o][0][0], fa[o][0][1], fa[o][1][0], fa[o][1][1]};
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), acc[o], 0, 0, 0);
        if (do_cs) {
          const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
          cs[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, ones), cs[o], 0, 0, 0);
        }
      }
    }
  }
  // D map: col = lane & 31 -> ci, row (e & 3) + 8 (e >> 2) + 4 (lane >> 5) -> co
  const int hh = lane >> 5;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co_base + (2 * wo + o) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh, ci = ci_base + wc * 32 + (lane & 31);
      atomicAdd(p.dw + (int64_t)co * p.Cin + ci, acc[o][e]);
      if (do_cs && (lane & 31) == 0) atomicAdd(p.colsum + co, cs[o][e]);
    }
  }
}

template <int NCO, int NCI>
int launch_w11(const W11Args& p, int gy, int gz, hipStream_t st) {
  const size_t lds = (size_t)(NCO + NCI) * 256 * 64 > 32768 ? (size_t)(NCO + NCI) * 256 * 64 : 32768;
  auto kern = k_wgrad1x1<NCO, NCI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int64_t ntiles = (p.nvox + 255) / 256;
  // splits over the voxel tiles: every split ends in (NCO * NCI * 1024) float atomics per workgroup, so wide layers (many channel pairs)
  // want few of them.  MI_W11_MIN_SPLIT / MI_W11_WGS: A/B knobs
  // (32 / 512 until late in round 3; 4: C3b 58.0 -> 57.2 ms, C5 48.3 -> 47.5, C4 21.51 -> 21.44, profiles/r04a_ab_w11_split*.log)
  static const int min_split = [] { const char* e = getenv("MI_W11_MIN_SPLIT"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
  static const int want_wgs = [] { const char* e = getenv("MI_W11_WGS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
  int gx = want_wgs / (gy * gz);
  if (gx < min_split) gx = min_split;
  if (gx > ntiles) gx = (int)ntiles;
  hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(512), lds, st, p);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// dw[Cout][Cin] (fp32) += dy^T x over all N * V voxels; colsum as mi_conv_wgrad.  Returns MI_ERR_UNSUPPORTED for shapes it does not cover.
int mi_launch_wgrad1x1(const void* x, int x_cs, int Cin, const void* dy, int dy_cs, int Cout, int N, int64_t V, float* dw, float* colsum,
                       int colsum_stride, hipStream_t st) {
  if ((x_cs & 7) || (dy_cs & 7) || (Cin & 7) || (Cout & 7)) return MI_ERR_UNSUPPORTED;
  if (colsum && colsum_stride != 0 && (V & 255)) return MI_ERR_UNSUPPORTED;  // tiles must not straddle images for per-image sums
  W11Args p;
  p.x = (const bf16*)x; p.x_cs = x_cs; p.Cin = Cin; p.dy = (const bf16*)dy; p.dy_cs = dy_cs; p.Cout = Cout;
  p.dw = dw; p.colsum = colsum; p.colsum_stride = colsum_stride; p.nvox = (int64_t)N * V; p.vox_per_image = V;
  p.ci0 = 0; p.co0 = 0;
  {  // wide layers: 128 x 128 workgroup blocks (k_wgrad1x1_wide); MI_W11_WIDE_MIN: smallest channel count that takes it (0: never)
    static const int wide_min = [] { const char* e = getenv("MI_W11_WIDE_MIN"); return e ? atoi(e) : 256; }();
    if (wide_min > 0 && Cin >= wide_min && Cout >= wide_min && Cin % 128 == 0 && Cout % 128 == 0 && (!colsum || colsum_stride == 0)) {
      static bool attr_set = false;
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_wgrad1x1_wide, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
      }
      const int gy = Cin / 128, gz = Cout / 128;
      const int64_t ntiles = (p.nvox + 255) / 256;
      static const int wide_wgs = [] { const char* e = getenv("MI_W11_WIDE_WGS"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
      int gx = wide_wgs / (gy * gz);
      if (gx < 1) gx = 1;
      if (gx > ntiles) gx = (int)ntiles;
      hipLaunchKernelGGL(k_wgrad1x1_wide, dim3(gx, gy, gz), dim3(512), 8 * 256 * 64, st, p);
      MI_CHECK_LAUNCH();
      return 0;
    }
  }
  const int nci = (Cin + 31) / 32, nco = (Cout + 31) / 32;
  const int NCO = nco % 2 == 0 ? 2 : 1;
  const int NCI = (NCO == 1 && nci % 3 == 0) ? 3 : (nci % 2 == 0 ? 2 : 1);  // (2 x 3 pairs of accumulators would spill)
  const int gy = nci / NCI, gz = nco / NCO;
  if (NCO == 1) {
    if (NCI == 1) return launch_w11<1, 1>(p, gy, gz, st);
    if (NCI == 2) return launch_w11<1, 2>(p, gy, gz, st);
    return launch_w11<1, 3>(p, gy, gz, st);
  }
  if (NCI == 1) return launch_w11<2, 1>(p, gy, gz, st);
  return launch_w11<2, 2>(p, gy, gz, st);
}

// Weight / bias gradient of a Linear layer y = x W^T + b on row-major bf16 activations (AttentionBlock's to_q/k/v, UNet:379-381):
// dW[out][in] += dy^T x, db[out] += column sums of dy -- the same single-pass kernel (no transposes, no split-K GEMM).
extern "C" int mi_linear_wgrad_bf16(const void* x, int ldx, int in_features, const void* dy, int ldy, int out_features, int64_t rows, float* dw,
                                    float* dbias, hipStream_t st) {
  if (!x || !dy || !dw || rows <= 0 || in_features <= 0 || out_features <= 0 || ldx < in_features || ldy < out_features) return MI_ERR_BAD_ARG;
  return mi_launch_wgrad1x1(x, ldx, in_features, dy, ldy, out_features, 1, rows, dw, dbias, 0, st);
}
