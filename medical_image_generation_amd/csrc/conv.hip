// 3-D / 2-D convolution for the UNet / AutoencoderKL hot path as ONE table-driven implicit GEMM on bf16 MFMA
// (replaces every `Convolution(conv_only=True)` -> nn.Conv{2,3}d call site: UNet:510,557,630,650,664,1820,1935; AEKL:67-86,
// 121,158-187,372,454,523,606,723-749).
//
// Formulation.  Every supported convolution (per-axis (k,s,p) in {(3,1,1), (3,2,1), (1,1,0)}), its data gradient and its
// weight gradient are instances of
//        out[n, o, CO] (+)= sum_{chunk, tap}  Wp[CO, chunk, tap] * in[n, o + delta(tap), chunk-channels]
// i.e. a stride-1 gather with a per-(output-channel-group, input-channel-chunk) TAP LIST:
//   * k3 s1:       27 taps (delta in {-1,0,1}^3);      k1: one tap;
//   * k3 s2 fwd:   runs on the space-to-depth image of x (8*Cin channels); the chunk's parity class q selects its taps
//                  ((q=0,d=0,t=1), (q=1,d=-1,t=0), (q=1,d=0,t=2) per axis) -> exactly 27*Cin MACs per output, no waste;
//   * dgrad s1:    the same kernel on dy with taps mirrored and the weight matrix transposed at pack time;
//   * dgrad s2:    produces the depth image of dx (output-channel group = parity class), then depth-to-space.
// The host-side plan (mi_conv_plan_*) builds the tables and owns the packed weights; the device code never branches on the
// convolution type.
//
// Mapping to CDNA4.  GEMM view: D[co, voxel] = A[co, k] * B[k, voxel], A = packed weights, B = activations, so each lane of
// the 32x32x16 accumulator holds 4x4 CONTIGUOUS output channels of one voxel (8-byte NDHWC stores, no transpose).
//   * B operand: the workgroup stages a halo tile (TD+2)x(TH+2)x(TW+2) voxels x 32 channels of x in LDS ONCE per channel
//     chunk and every tap reads it at a wave-uniform byte offset -> each activation byte is fetched ~2x from L2/HBM instead
//     of 27x.  Voxel pitch 80 B and row pitch == 128 (mod 256) make the 4x8-voxel fragment reads (ds_read_b128)
//     bank-conflict-free.  Staging goes through registers (issue-early / write-late) because it also applies the fused
//     GroupNorm-affine + SiLU prologue (x*scale[n,c]+shift[n,c] -> silu) in fp32 -- the activated tensor never exists in HBM.
//   * A operand: fragments are pre-packed in consumption order (one coalesced 1 KiB global_load_dwordx4 per fragment, served
//     from L2 -- weights are shared by every workgroup) and double-buffered in registers across taps.
//   * Epilogue: + (bias [+ time-embedding]) per (n, co), + residual, cast, store.
// Weight gradient: D[co, ci] += dY^T[co, vox] * X[vox, ci]; both operands need the voxel index on the MFMA k axis, which the
// NDHWC LDS images provide through ds_read_b64_tr_b16 (hardware transposed read); taps are distributed over the 4 waves,
// partial sums over tile ranges go to an fp32 slab that a reduce kernel folds into the torch-layout gradient.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "conv_common.h"
#include "medimgen_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ halo staging
// Each thread owns NP 16-byte pieces of the halo image: piece pc = tid + 256*i -> LDS voxel pc >> 2, channel part pc & 3.
// The voxel's coordinates inside the LDS image do not depend on the tile, so they are decoded ONCE per kernel and kept
// packed (10 bits per axis); global offsets and validity are rebuilt per tile with a handful of integer ops.
template <int NP>
struct Stage {
  int pk[NP];      // (hd << 20) | (hh << 10) | hw, or -1 for pieces beyond the image
  u32x4 pre[NP];   // staged data (next image), in flight while the current image is consumed
  unsigned valid;  // bit i: piece i lies inside the tensor (prologue applies only there: zero padding stays zero)
};

template <int NP, int NT = 256>
__device__ __forceinline__ void stage_init(Stage<NP>& s, const Geom& g) {
  const int HVOX = g.HD * g.HH * g.HW;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int v = (threadIdx.x + NT * i) >> 2;
    int hdz = v / (g.HH * g.HW);
    int rem = v - hdz * (g.HH * g.HW);
    int hhz = rem / g.HW, hwz = rem - hhz * g.HW;
    s.pk[i] = v < HVOX ? ((hdz << 20) | (hhz << 10) | hwz) : -1;
  }
  s.valid = 0;
}

// Wave-uniform buffer descriptor (raw, no stride): 32-bit per-lane byte offsets, hardware range check (an offset beyond
// num_records loads zeros -- which is exactly the conv's zero padding), scalar soffset for wave-uniform displacements.
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int NP>
__device__ __forceinline__ void stage_load(Stage<NP>& s, const ConvArgs& a, int n, int d0, int h0, int w0, int src_c0) {
  const Geom& g = a.g;
  const int part = threadIdx.x & 3;
  const int c = src_c0 + part * 8;
  const bool vec = ((a.x_cs & 7) == 0) && (c + 8 <= a.Cin);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  unsigned valid = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pk = s.pk[i];
    const int gd = d0 - g.hd + (pk >> 20), gh = h0 - g.hh + ((pk >> 10) & 1023), gw = w0 - g.hw + (pk & 1023);
    const bool ok = pk >= 0 && gd >= 0 && gd < a.Di && gh >= 0 && gh < a.Hi && gw >= 0 && gw < a.Wi;
    valid |= ok ? (1u << i) : 0u;
    const unsigned off = ok ? (unsigned)((((n * a.Di + gd) * a.Hi + gh) * a.Wi + gw) * a.x_cs + c) * 2u : 0xffffffffu;
    if (vec) {
      s.pre[i] = buf_load16(rx, off, 0);  // out-of-range offset -> zeros
    } else {  // ragged channel counts (Cin = 1, 4, ...): element-wise with masking
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        const bf16* p = a.x + (off >> 1);
        F8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = (c + j < a.Cin) ? bf2f(p[j]) : 0.f;
        v = pack8(f);
      }
      s.pre[i] = v;
    }
  }
  s.valid = valid;
}

template <int NP>
__device__ __forceinline__ void stage_store(Stage<NP>& s, const ConvArgs& a, int n, int src_c0, char* lds) {
  const Geom& g = a.g;
  const int part = threadIdx.x & 3;
  float sc[8], sh[8];
  if (a.ss) {
    const int c = (src_c0 % a.ss_C) + part * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bool in = c + j < a.ss_C;
      sc[j] = in ? a.ss[((int64_t)n * a.ss_C + c + j) * 2] : 0.f;
      sh[j] = in ? a.ss[((int64_t)n * a.ss_C + c + j) * 2 + 1] : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pk = s.pk[i];
    if (pk < 0) continue;
    u32x4 v = s.pre[i];
    if (a.ss && ((s.valid >> i) & 1u)) {
      F8 f = unpack8(v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float u = f.v[j] * sc[j] + sh[j];
        f.v[j] = a.pro_silu ? silu_f(u) : u;
      }
      v = pack8(f);
    }
    *(u32x4*)(lds + (pk >> 20) * g.slice + ((pk >> 10) & 1023) * g.row + (pk & 1023) * g.vox + part * 16) = v;
  }
}

// voxel block b (32 voxels = 4 rows x 8 cols of one slice) -> tile-relative (d, h, w) of its corner
__device__ __forceinline__ void block_origin(const Geom& g, int b, int& bd, int& bh, int& bw) {
  const int bpw = g.TW / 8, bph = g.TH / 4;
  bw = (b % bpw) * 8;
  bh = ((b / bpw) % bph) * 4;
  bd = b / (bpw * bph);
}

// ------------------------------------------------------------------------------------------------ forward / dgrad kernel
// LDS reads the compiler may not move or wait for: issue-early / wait-late is placed by hand (cdna guide 5.7, form (ii):
// the wait statement names every destination "+v", and a sched_barrier keeps the MFMAs on their side of it).
template <int OFF>
__device__ __forceinline__ void lds_read16_async(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int VB>
__device__ __forceinline__ void lds_wait_frags(u32x4 (&f)[2][VB]) {
  if constexpr (VB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[1][0]));
  __builtin_amdgcn_sched_barrier(0);
}
// wait for one fragment set while the NEXT set's 2*VB reads (issued after it; LDS returns in order) stay in flight
template <int VB>
__device__ __forceinline__ void lds_wait_frags_keep1(u32x4 (&f)[2][VB]) {
  if constexpr (VB == 2)
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]));
  else
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0][0]), "+v"(f[1][0]));
  __builtin_amdgcn_sched_barrier(0);
}

template <int OFF>
__device__ __forceinline__ void lds_read16_async(u32x4& dst, unsigned addr);
template <int VB>
__device__ __forceinline__ void lds_wait_frags(u32x4 (&f)[2][VB]);
template <int VB>
__device__ __forceinline__ void lds_wait_frags_keep1(u32x4 (&f)[2][VB]);

constexpr int F27_ROW = 896, F27_SLICE = 8960;  // LDS pitches of the 4x8x8 (+1 halo) tile every k3-s1 3-D conv uses
template <int T, int FLIP>
constexpr int f27_off() {  // byte offset of tap T's window (mirrored for the data gradient)
  constexpr int U = FLIP ? 26 - T : T;
  return (U / 9) * F27_SLICE + ((U / 3) % 3) * F27_ROW + (U % 3) * VOXB;
}

template <int T, int FLIP, int VB>
__device__ __forceinline__ void f27_issue(u32x4 (&f)[2][VB], const unsigned (&baddr)[VB]) {
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    lds_read16_async<f27_off<T, FLIP>()>(f[0][vb], baddr[vb]);
    lds_read16_async<f27_off<T, FLIP>() + 32>(f[1][vb], baddr[vb]);
  }
}
// taps T..26, fully unrolled by template recursion so every LDS offset is an immediate
template <int T, int NCB, int VB, int RING, int FLIP>
__device__ __forceinline__ void f27_taps(f32x16 (&acc)[VB][NCB], u32x4 (&wa)[RING][2][NCB], u32x4 (&fb)[2][2][VB],
                                         const unsigned (&baddr)[VB], __amdgpu_buffer_rsrc_t rw, unsigned wsoff) {
  if constexpr (T < 27) {
    constexpr int q = T % RING, cur = T & 1;
    if constexpr (T + 1 < 27) f27_issue<T + 1, FLIP, VB>(fb[cur ^ 1], baddr);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int vb = 0; vb < VB; ++vb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[vb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[q][ks][cb]),
                                                                __builtin_bit_cast(bf16x8, fb[cur][ks][vb]), acc[vb][cb], 0, 0, 0);
    if constexpr (T + RING < 27) {
      const unsigned wlane = (threadIdx.x & 63) * 16u;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) wa[q][ks][cb] = buf_load16(rw, wlane, wsoff + (((T + RING) * 2 + ks) * NCB + cb) * 1024u);
    }
    if constexpr (T + 1 < 27) lds_wait_frags<VB>(fb[cur ^ 1]);
    f27_taps<T + 1, NCB, VB, RING, FLIP>(acc, wa, fb, baddr, rw, wsoff);
  }
}

// lookahead-2 variant: three fragment sets; while tap T computes, taps T+1 and T+2 are in flight
template <int T, int NCB, int VB, int RING, int FLIP>
__device__ __forceinline__ void f27_taps3(f32x16 (&acc)[VB][NCB], u32x4 (&wa)[RING][2][NCB], u32x4 (&fb)[3][2][VB],
                                          const unsigned (&baddr)[VB], __amdgpu_buffer_rsrc_t rw, unsigned wsoff) {
  if constexpr (T < 27) {
    constexpr int q = T % RING, cur = T % 3;
    if constexpr (T + 2 < 27) f27_issue<T + 2, FLIP, VB>(fb[(T + 2) % 3], baddr);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int vb = 0; vb < VB; ++vb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[vb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[q][ks][cb]),
                                                                __builtin_bit_cast(bf16x8, fb[cur][ks][vb]), acc[vb][cb], 0, 0, 0);
    if constexpr (T + RING < 27) {
      const unsigned wlane = (threadIdx.x & 63) * 16u;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) wa[q][ks][cb] = buf_load16(rw, wlane, wsoff + (((T + RING) * 2 + ks) * NCB + cb) * 1024u);
    }
    if constexpr (T + 2 < 27) lds_wait_frags_keep1<VB>(fb[(T + 1) % 3]);
    else if constexpr (T + 1 < 27) lds_wait_frags<VB>(fb[(T + 1) % 3]);
    f27_taps3<T + 1, NCB, VB, RING, FLIP>(acc, wa, fb, baddr, rw, wsoff);
  }
}

template <int NCB, int VB, int NP, int RING, int WPS, int MODE>  // MODE 0: table-driven taps, 1: k3-s1 forward, 2: k3-s1 dgrad
__global__ void __launch_bounds__(256, WPS) k_conv_igemm(ConvArgs a) {
  constexpr bool FULL27 = MODE != 0;
  constexpr int FLIP = MODE == 2;  // WPS = 2: two workgroups per CU overlap staging with MFMA
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const Geom& g = a.g;
  const int y = blockIdx.y;
  int tile_last, tile_step;
  int tile = first_tile(a.ntiles, tile_last, tile_step);
  if (tile >= tile_last) return;  // whole workgroup, before any barrier

  int bbase[VB], bvh[VB], bvw[VB], bvd[VB];  // this lane's voxel inside the tile (LDS byte address / coordinates)
#pragma unroll
  for (int vb = 0; vb < VB; ++vb) {
    int bd, bh, bw;
    block_origin(g, wave * VB + vb, bd, bh, bw);
    bvd[vb] = bd; bvh[vb] = bh + (r >> 3); bvw[vb] = bw + (r & 7);
    bbase[vb] = bd * g.slice + bvh[vb] * g.row + bvw[vb] * g.vox + h * 16;
  }
  const int* hdr = a.hdr + (int64_t)y * a.nchunks * 4;
  const int cls = y / a.ogpq, cls_base = cls * a.outc_q, cls_lim = cls_base + a.outc_q;

  int n, d0, h0, w0;
  tile_origin(g, tile, n, d0, h0, w0);
  Stage<NP> st;
  stage_init<NP>(st, g);
  stage_load<NP>(st, a, n, d0, h0, w0, hdr[2]);
  int n_pre = n;  // batch index of the data held in st.pre
  bool first_img = true;

  while (true) {
    f32x16 acc[VB][NCB];
#pragma unroll
    for (int vb = 0; vb < VB; ++vb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[vb][cb][e] = 0.f;
    const int next_tile = tile + tile_step;
    int nn = 0, nd0 = 0, nh0 = 0, nw0 = 0;

    for (int ch = 0; ch < a.nchunks; ++ch) {
      const int tap_begin = hdr[ch * 4 + 0], ntaps = hdr[ch * 4 + 1], src_c0 = hdr[ch * 4 + 2], wfrag = hdr[ch * 4 + 3];
      // weight ring: the first RING taps' fragments are requested BEFORE the staging phase, which hides their L2 latency
      const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wpk, a.wpk_bytes);
      const unsigned wsoff = (unsigned)wfrag * 1024u;  // scalar byte offset of this chunk's first fragment
      const unsigned wlane = lane * 16u;
      u32x4 wa[RING][2][NCB];
#pragma unroll
      for (int q = 0; q < RING; ++q)
        if (q < ntaps) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) wa[q][ks][cb] = buf_load16(rw, wlane, wsoff + ((q * 2 + ks) * NCB + cb) * 1024u);
        }
      __syncthreads();  // every wave is done reading the previous tile image
      if (a.dbg != 1 || first_img) stage_store<NP>(st, a, n_pre, src_c0, lds);
      __syncthreads();
      first_img = false;
      // next image (next chunk of this tile, or chunk 0 of this workgroup's next tile): in flight under the MFMAs below
      if (a.dbg == 1) {
        if (ch + 1 >= a.nchunks && next_tile < tile_last) tile_origin(g, next_tile, nn, nd0, nh0, nw0);
      } else if (ch + 1 < a.nchunks) {
        stage_load<NP>(st, a, n, d0, h0, w0, hdr[(ch + 1) * 4 + 2]);
      } else if (next_tile < tile_last) {
        tile_origin(g, next_tile, nn, nd0, nh0, nw0);
        stage_load<NP>(st, a, nn, nd0, nh0, nw0, hdr[2]);
        n_pre = nn;
      }
      if (a.dbg == 2) continue;
      if constexpr (FULL27) {
        // k3 s1 in all three axes: the 27 taps are a compile-time loop nest (offsets = i*slice + j*row + k*VOXB, mirrored
        // for the data gradient), so the whole chunk is straight-line code the scheduler can software-pipeline.
        // B fragments are double-buffered by hand: tap t+1's LDS reads are issued before tap t's MFMAs and waited for after
        // them (left to itself hipcc puts each ds_read right in front of its consumer: one LDS round trip per MFMA).  The
        // tap offsets are immediates, so one address register per voxel block serves all 27 taps.
        typedef __attribute__((address_space(3))) char lds_char;
        const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds;
        unsigned baddr[VB];
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
          baddr[vb] = lds0 + bbase[vb];
          asm volatile("" : "+v"(baddr[vb]));  // keep it ONE register: no per-tap address hoisting
        }
        if constexpr (NCB == 1) {  // registers to spare: look two taps ahead
          u32x4 fb[3][2][VB];
          f27_issue<0, FLIP, VB>(fb[0], baddr);
          f27_issue<1, FLIP, VB>(fb[1], baddr);
          lds_wait_frags_keep1<VB>(fb[0]);
          f27_taps3<0, NCB, VB, RING, FLIP>(acc, wa, fb, baddr, rw, wsoff);
        } else {
          u32x4 fb[2][2][VB];
          f27_issue<0, FLIP, VB>(fb[0], baddr);
          lds_wait_frags<VB>(fb[0]);
          f27_taps<0, NCB, VB, RING, FLIP>(acc, wa, fb, baddr, rw, wsoff);
        }
      } else {
      for (int t0 = 0; t0 < ntaps; t0 += RING) {
#pragma unroll
        for (int q = 0; q < RING; ++q) {
          const int t = t0 + q;
          if (t < ntaps) {  // wave-uniform
            const int toff = a.taps[tap_begin + t];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
              for (int vb = 0; vb < VB; ++vb) {
                bf16x8 fb = *(const bf16x8*)(lds + bbase[vb] + toff + ks * 32);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                  acc[vb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[q][ks][cb]), fb, acc[vb][cb], 0, 0, 0);
              }
            }
            if (t + RING < ntaps) {  // refill this slot RING taps ahead
#pragma unroll
              for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                  wa[q][ks][cb] = buf_load16(rw, wlane, wsoff + (((t + RING) * 2 + ks) * NCB + cb) * 1024u);
            }
          }
        }
      }
      }
    }

    // epilogue: lane holds voxel r of each block, channels co_base + 8*grp + 4*h + (0..3) for grp = 0..3
    if (a.dbg != 3)
#pragma unroll
    for (int vb = 0; vb < VB; ++vb) {
      const int od = d0 + bvd[vb], oh = h0 + bvh[vb], ow = w0 + bvw[vb];
      if (od >= a.Do || oh >= a.Ho || ow >= a.Wo) continue;
      const int64_t vox = ((int64_t)(n * a.Do + od) * a.Ho + oh) * a.Wo + ow;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
          const int co = cls_base + ((y - cls * a.ogpq) * NCB + cb) * 32 + (a.perm16 ? h * 16 + grp * 4 : grp * 8 + h * 4);
          if (co >= cls_lim) continue;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[vb][cb][grp * 4 + i];
          const bool full = (co + 4 <= cls_lim) && ((a.y_cs & 3) == 0);
          if (a.addvec) {
            const float* av = a.addvec + (int64_t)n * a.addvec_stride + co;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (co + i < cls_lim) v[i] += av[i];
          }
          if (a.res) {
            const bf16* rp = a.res + vox * a.res_cs + co;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (co + i < cls_lim) v[i] += bf2f(rp[i]);
          }
          bf16* yp = a.y + vox * a.y_cs + co;
          if (full) {
            u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
            *(u32x2*)yp = o;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (co + i < cls_lim) yp[i] = f2bf(v[i]);
          }
        }
      }
    }
    if (next_tile >= tile_last) break;
    tile = next_tile; n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
  }
}

// ------------------------------------------------------------------------------------------------ weight-gradient kernel
struct WgradArgs {
  ConvArgs c;         // x side: loader geometry / tables (NCB = 1 tables); y/res/addvec unused
  const bf16* dy; int dy_cs;
  unsigned dy_bytes;  // size of dy for its buffer descriptor (LDS-DMA kernel)
  int nbuf;           // LDS-DMA kernel: image ring depth (2; 4 for 1x1 pairs)
  int flags;          // LDS-DMA kernel: tiles handed over by LDS counters instead of a barrier per tile (MI_WGRAD_FLAGS, default 1)
  float* cs_part;     // [nsplit][N][Cout] per-workgroup column sums (tap pairs): plain stores, folded by cs_reduce_block (trailing blocks of the reduce launch) -- 256 workgroups
                      // adding to the same 128-byte line with atomics serialise for tens of microseconds
  float* part;        // [nsplit][npairs][MAXTAPS? -> ntaps_of_pair][32][32] laid out by pair_off
  const int* pair_off;  // [npairs] float offset of the pair's slab inside one split's partial
  int64_t split_stride; // floats per split
  int ntiles, nsplit, npairs;
  int contig;           // k_conv_wgrad2: every XCD class owns a contiguous eighth of the tiles (needs ntiles % 8 == nsplit % 8 == 0)
  int dbg;              // ablation knob (MI_WGRAD_DBG): 1 = stage only the first tile, 2 = skip the MFMA loop
  float* colsum;        // optional: colsum[n * colsum_stride + co] += sum over voxels of dy (bias / time-embedding gradient)
  int colsum_stride;
  // LDS-DMA kernel, factor-2 layers read in place (no space-to-depth / up-sampled copy in HBM): image voxel u of a pair is tensor voxel
  // sx * u + class (x image; k3 s2 conv: sx = 2) resp. sy * u + phase (dY image; Upsample + conv: sy = 2), the class / phase of the
  // pair = bits 2..0 (d, h, w) of its header word 3.  1 / 1: dense images.
  int sx, sy;
  int cs_chunks;        // the first cs_chunks chunk slots of a cout block own the dY column sums (1; Upsample + conv: its 8 phases)
};

// ds_read_b64_tr_b16 (4 voxels x 16 channels delivered channel-per-lane) issued through asm so that a whole k-step's
// reads can be put in flight before the previous k-step's MFMAs; waited for by wg_wait (cdna guide 5.7 form (ii)).
__device__ __forceinline__ void tr_read_async(u32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int MAXT>
struct WgFrags {
  u32x2 a[2];
  u32x2 b[MAXT ? MAXT : 1][2];  // (MAXT = 0: a wave that only accumulates the dY column sums)
};
template <int MAXT>
__device__ __forceinline__ void wg_issue(WgFrags<MAXT>& f, unsigned ya, unsigned xa, const int (&toff)[MAXT ? MAXT : 1], int vox4) {
  tr_read_async(f.a[0], ya);
  tr_read_async(f.a[1], ya + vox4);
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    tr_read_async(f.b[t][0], xa + toff[t]);
    tr_read_async(f.b[t][1], xa + toff[t] + vox4);
  }
}
template <int MAXT>
__device__ __forceinline__ void wg_wait(WgFrags<MAXT>& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]));
#pragma unroll
  for (int t = 0; t < MAXT; ++t) asm volatile("" : "+v"(f.b[t][0]), "+v"(f.b[t][1]));
  __builtin_amdgcn_sched_barrier(0);
}
// CS: this wave also owns the fused bias gradient.  The column sums of dY over the k-step's 16 voxels are one more MFMA of the
// dY fragment against an all-ones B fragment (every column of the result holds sum_k dY[k][co]) -- no LDS pass, no VALU.
template <int MAXT, bool CS>
__device__ __forceinline__ void wg_mfma(f32x16 (&acc)[MAXT ? MAXT : 1], const WgFrags<MAXT>& f, f32x16& cs) {
  u32x4 ra = {f.a[0][0], f.a[0][1], f.a[1][0], f.a[1][1]};
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    u32x4 rb = {f.b[t][0][0], f.b[t][0][1], f.b[t][1][0], f.b[t][1][1]};
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), acc[t], 0, 0, 0);
  }
  if constexpr (CS) {
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};  // bf16 1.0 x 8
    cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, ones), cs, 0, 0, 0);
  }
}

template <int NPY>
struct StageY {
  u32x4 pre[NPY];
};

template <int NPY, int NT = 256>
__device__ __forceinline__ void stagey_load(StageY<NPY>& sy, const WgradArgs& w, int y, int n, int d0, int h0, int w0) {
  const ConvArgs& a = w.c;
  const Geom& g = a.g;
  const int NVOX = g.TD * g.TH * g.TW;
  const int part = threadIdx.x & 3, co = y * 32 + part * 8;
#pragma unroll
  for (int i = 0; i < NPY; ++i) {
    int v = (threadIdx.x + NT * i) >> 2;
    int vd = v / (g.TH * g.TW), rem = v - vd * (g.TH * g.TW);
    int vh = rem / g.TW, vw = rem - vh * g.TW;
    int od = d0 + vd, oh = h0 + vh, ow = w0 + vw;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (v < NVOX && od < a.Do && oh < a.Ho && ow < a.Wo && co < a.Cout) {
      const bf16* p = w.dy + ((int64_t)((n * a.Do + od) * a.Ho + oh) * a.Wo + ow) * w.dy_cs + co;
      if (co + 8 <= a.Cout && (w.dy_cs & 7) == 0) {
        val = *(const u32x4*)p;
      } else {
        F8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = (co + j < a.Cout) ? bf2f(p[j]) : 0.f;
        val = pack8(f);
      }
    }
    sy.pre[i] = val;
  }
}

template <int NPY, int NT = 256>
__device__ __forceinline__ void stagey_store(const StageY<NPY>& sy, const Geom& g, char* ldy) {
  const int NVOX = g.TD * g.TH * g.TW;
  const int part = threadIdx.x & 3;
#pragma unroll
  for (int i = 0; i < NPY; ++i) {
    int v = (threadIdx.x + NT * i) >> 2;
    if (v < NVOX) *(u32x4*)(ldy + v * g.vox + part * 16) = sy.pre[i];  // dY image: [TD][TH][TW] voxels, dense
  }
}

// One pass over the tile's 16 k-steps (16 voxels = 2 h-rows x 8 w of one slice) for a wave that owns NTAP taps.
// Reads of k-step s+1 are in flight under the MFMAs of k-step s (two statically named fragment sets, loop unrolled in pairs).
template <int NTAP, bool CS>
__device__ __forceinline__ void wg_tile(f32x16 (&acc)[NTAP ? NTAP : 1], const int (&toff)[NTAP ? NTAP : 1], f32x16& cs, const Geom& g,
                                        unsigned lds0, unsigned ldy0, int kh, int q, int chan_b, int dyrow, int dyslice, int dbg) {
  int sd = 0, sh = 0, sw = 0;
  auto addr_y = [&]() { return ldy0 + sd * dyslice + (sh + kh) * dyrow + (sw + q) * g.vox + chan_b; };
  auto addr_x = [&]() { return lds0 + sd * g.slice + (sh + kh) * g.row + (sw + q) * g.vox + chan_b; };
  auto advance = [&]() {
    sw += 8;
    if (sw >= g.TW) { sw = 0; sh += 2; if (sh >= g.TH) { sh = 0; ++sd; } }
  };
  const int ksteps = g.TD * (g.TH / 2) * (g.TW / 8);  // even (checked on the host)
  const int vox4 = 4 * g.vox;
  WgFrags<NTAP> fA, fB;
  wg_issue<NTAP>(fA, addr_y(), addr_x(), toff, vox4);
  wg_issue<NTAP>(fB, addr_y(), addr_x(), toff, vox4);
  wg_wait<NTAP>(fA);
  wg_wait<NTAP>(fB);
  for (int s = 0; s < ksteps; s += 2) {
    advance();
    if (dbg != 4) wg_issue<NTAP>(fB, addr_y(), addr_x(), toff, vox4);
    if (dbg != 3) wg_mfma<NTAP, CS>(acc, fA, cs);
    if (dbg != 4) wg_wait<NTAP>(fB);
    advance();
    if (s + 2 < ksteps && dbg != 4) wg_issue<NTAP>(fA, addr_y(), addr_x(), toff, vox4);
    if (dbg != 3) wg_mfma<NTAP, CS>(acc, fB, cs);
    if (s + 2 < ksteps && dbg != 4) wg_wait<NTAP>(fA);
  }
}

// 1-tap pairs (1x1 convs): splitting taps over waves would leave 7 of 8 waves idle, so the 16 k-steps are split instead
// (wave w takes k-steps 2w, 2w+1) and the 8 partial accumulators are folded through LDS once, after the last tile.
template <bool CS>
__device__ __forceinline__ void wg_tile_ksplit(f32x16 (&acc)[1], const int (&toff)[1], f32x16& cs, const Geom& g, unsigned lds0,
                                               unsigned ldy0, int kh, int q, int chan_b, int dyrow, int dyslice, int wave) {
  int sd = 0, sh = 0, sw = 0;
  auto advance = [&]() {
    sw += 8;
    if (sw >= g.TW) { sw = 0; sh += 2; if (sh >= g.TH) { sh = 0; ++sd; } }
  };
  for (int i = 0; i < 2 * wave; ++i) advance();
  const int vox4 = 4 * g.vox;
  WgFrags<1> fA, fB;
  wg_issue<1>(fA, ldy0 + sd * dyslice + (sh + kh) * dyrow + (sw + q) * g.vox + chan_b,
              lds0 + sd * g.slice + (sh + kh) * g.row + (sw + q) * g.vox + chan_b, toff, vox4);
  advance();
  wg_issue<1>(fB, ldy0 + sd * dyslice + (sh + kh) * dyrow + (sw + q) * g.vox + chan_b,
              lds0 + sd * g.slice + (sh + kh) * g.row + (sw + q) * g.vox + chan_b, toff, vox4);
  wg_wait<1>(fA);
  wg_wait<1>(fB);
  wg_mfma<1, CS>(acc, fA, cs);
  wg_mfma<1, CS>(acc, fB, cs);
}

// 3-D k3 convs always use the 4x8x8 (+1 halo) tile with 64-byte voxels: row 640, slice 6400 for the x image, 512 / 4096
// for the dY image.  Every k-step / half-row offset is then a ds_read immediate; one address register per tap, set once.
constexpr int WG3_XROW = 640, WG3_XSLICE = 6400, WG3_YROW = 512, WG3_YSLICE = 4096;
template <int OFF>
__device__ __forceinline__ void tr_read_async_i(u32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int S, int NTAP>
__device__ __forceinline__ void wg3_issue(WgFrags<NTAP>& f, unsigned ybase, const unsigned (&xbase)[NTAP]) {
  constexpr int OY = (S >> 2) * WG3_YSLICE + (S & 3) * 2 * WG3_YROW;
  constexpr int OX = (S >> 2) * WG3_XSLICE + (S & 3) * 2 * WG3_XROW;
  tr_read_async_i<OY>(f.a[0], ybase);
  tr_read_async_i<OY + 256>(f.a[1], ybase);
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    tr_read_async_i<OX>(f.b[t][0], xbase[t]);
    tr_read_async_i<OX + 256>(f.b[t][1], xbase[t]);
  }
}
template <int S, int NTAP, bool CS>
__device__ __forceinline__ void wg3_steps(f32x16 (&acc)[NTAP], f32x16& cs, WgFrags<NTAP>& fcur, WgFrags<NTAP>& fnext, unsigned ybase,
                                          const unsigned (&xbase)[NTAP]) {
  if constexpr (S < 16) {
    if constexpr (S + 1 < 16) wg3_issue<S + 1, NTAP>(fnext, ybase, xbase);
    wg_mfma<NTAP, CS>(acc, fcur, cs);
    if constexpr (S + 1 < 16) wg_wait<NTAP>(fnext);
    wg3_steps<S + 1, NTAP, CS>(acc, cs, fnext, fcur, ybase, xbase);
  }
}
template <int NTAP, bool CS>
__device__ __forceinline__ void wg3_tile(f32x16 (&acc)[NTAP], const int (&toff)[NTAP], f32x16& cs, unsigned lds0, unsigned ldy0, int kh,
                                         int q, int chan_b) {
  const unsigned lane_y = ldy0 + kh * WG3_YROW + q * 64 + chan_b;
  unsigned xbase[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) xbase[t] = lds0 + kh * WG3_XROW + q * 64 + chan_b + toff[t];
  WgFrags<NTAP> fA, fB;
  wg3_issue<0, NTAP>(fA, lane_y, xbase);
  wg_wait<NTAP>(fA);
  wg3_steps<0, NTAP, CS>(acc, cs, fA, fB, lane_y, xbase);
}

// 512 threads = 8 waves = 2 per SIMD: ds_read_b64_tr_b16 is issue-bound for a lone wave (an 8-byte LDS read needs several
// waves per SIMD to approach its rate), so two waves alternate LDS reads and MFMAs on every SIMD.  The pair's taps are dealt
// round-robin over the 8 waves (27 taps -> 4,4,4,3,3,3,3,3: every SIMD gets 7 or 6); each wave keeps its taps' 32x32
// accumulators for the whole tile range of the workgroup.
template <int NP, int NPY, bool GEO3D>
__global__ void __launch_bounds__(512, 2) k_conv_wgrad(WgradArgs w) {
  constexpr int NT = 512, MAXT = 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const ConvArgs& a = w.c;
  const Geom& g = a.g;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ldy = lds + g.lds_bytes;  // dY tile image
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned lds0 = (unsigned)(size_t)(lds_char*)lds, ldy0 = lds0 + g.lds_bytes;
  // 1-D grid, XCD-aware: blocks whose ids are equal mod 8 are observed to share an XCD (speed only), and the pairs
  // (cout block, cin chunk) of ONE tile range read the same x / dY lines -- so they get consecutive slots of one residue
  // class and hit the same L2 instead of fetching every 64-byte slice of a voxel line through a different one.
  int pair, split;
  if (w.nsplit >= 8) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair = slot % w.npairs;                 // (y, chunk)
    split = (slot / w.npairs) * 8 + xcd;    // tile-range index
  } else {  // few tile ranges (wide layers): every pair reads (nearly) all tiles anyway -- just spread the blocks evenly
    pair = blockIdx.x / w.nsplit;
    split = blockIdx.x % w.nsplit;
  }
  if (split >= w.nsplit || pair >= w.npairs) return;
  const int y = pair / a.nchunks;
  const int* hdr = a.hdr + (int64_t)pair * 4;
  const int tap_begin = hdr[0], ntaps = hdr[1], src_c0 = hdr[2];

  // transposed-read lane roles: 16-lane group gq -> channel half (gq&1), k half (gq>>1); lane 4q+p -> voxel row q, chan 4p
  const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int kh = gq >> 1;                           // which of the 2 h-rows of the k-step
  const int chan_b = ((gq & 1) * 16 + pp * 4) * 2;  // byte offset of the 4-channel piece inside the voxel

  f32x16 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  int toff[MAXT];
  int nt = 0;  // taps of this wave: wave, wave + 8, ...
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    int ti = wave + 8 * t;
    toff[t] = a.taps[tap_begin + (ti < ntaps ? ti : 0)];
    nt += ti < ntaps ? 1 : 0;
  }
  const bool ksplit = !GEO3D && ntaps == 1;  // 1x1 conv: split the k-steps over the waves instead of the taps
  const int dyrow = g.TW * g.vox, dyslice = g.TH * dyrow;

  int tile = split;
  if (tile >= w.ntiles) return;
  int n, d0, h0, w0;
  tile_origin(g, tile, n, d0, h0, w0);
  // fused bias / time-embedding gradient (column sums of dY): the workgroup that owns the first cin chunk of its cout block
  // adds one all-ones MFMA per k-step (wg_mfma<.., CS>).  Tap pairs: wave 7 -- it never has more taps than the others --
  // keeps the sums in a spare accumulator; 1x1 pairs (k-steps split over the waves): every wave sums its own k-steps.
  // Flushed with atomics when the image index changes (tiles are visited in increasing order) and at the end.
  const bool do_colsum = w.colsum != nullptr && (pair % a.nchunks) == 0;
  const bool cs_wave = do_colsum && (ksplit || wave == 7);
  f32x16& cs = acc[3];  // never a tap accumulator on a column-sum wave (wave 7 owns <= 3 taps; 1x1 pairs use acc[0] only)
  auto cs_flush = [&](int img) {  // every column of `cs` holds the same sums: lanes 0 / 32 own rows 4h + (e&3) + 8(e>>2)
    if ((lane & 31) == 0) {
      const int hh = lane >> 5;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = y * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (co < a.Cout) {
          if (w.cs_part && !ksplit) w.cs_part[((int64_t)split * a.N + img) * a.Cout + co] = cs[e];
          else atomicAdd(w.colsum + (int64_t)img * w.colsum_stride + co, cs[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.f;
  };
  int cs_n = n;
  if (cs_wave && w.cs_part && !ksplit) {  // rows of images this workgroup never reaches must read as zero
    for (int i = lane; i < a.N * 32; i += 64) {
      const int img = i >> 5, co = y * 32 + (i & 31);
      if (co < a.Cout) w.cs_part[((int64_t)split * a.N + img) * a.Cout + co] = 0.f;
    }
  }
  Stage<NP> st;
  StageY<NPY> sy;
  stage_init<NP, NT>(st, g);
  stage_load<NP>(st, a, n, d0, h0, w0, src_c0);
  stagey_load<NPY, NT>(sy, w, y, n, d0, h0, w0);
  bool first = true;
  int n_img = n;  // image of the tile that is in LDS during the current iteration
  while (true) {
    __syncthreads();  // previous tile fully consumed
    if ((w.dbg != 1 && w.dbg < 3) || first) {
      stage_store<NP>(st, a, n, src_c0, lds);
      stagey_store<NPY, NT>(sy, g, ldy);
    }
    __syncthreads();
    if (cs_wave && n_img != cs_n) {  // wave-uniform
      cs_flush(cs_n);
      cs_n = n_img;
    }
    const int next = tile + w.nsplit;
    if (next < w.ntiles) {  // next tile's loads fly under this tile's MFMAs
      tile_origin(g, next, n, d0, h0, w0);
      n_img = n;
      if (w.dbg != 1 && w.dbg < 3) {
        stage_load<NP>(st, a, n, d0, h0, w0, src_c0);
        stagey_load<NPY, NT>(sy, w, y, n, d0, h0, w0);
      }
    }
    first = false;
    if (w.dbg != 2) {  // wave-uniform dispatch on this wave's tap count
      f32x16(&a3)[3] = reinterpret_cast<f32x16(&)[3]>(acc);
      const int(&t3)[3] = reinterpret_cast<const int(&)[3]>(toff);
      if constexpr (GEO3D) {  // 27 taps: waves 0-2 own 4, waves 3-7 own 3
        if (nt == 4) wg3_tile<4, false>(acc, toff, cs, lds0, ldy0, kh, q, chan_b);
        else if (cs_wave) wg3_tile<3, true>(a3, t3, cs, lds0, ldy0, kh, q, chan_b);
        else wg3_tile<3, false>(a3, t3, cs, lds0, ldy0, kh, q, chan_b);
      } else {
        f32x16(&a2)[2] = reinterpret_cast<f32x16(&)[2]>(acc);
        const int(&t2)[2] = reinterpret_cast<const int(&)[2]>(toff);
        f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc);
        const int(&t1)[1] = reinterpret_cast<const int(&)[1]>(toff);
        if (ksplit) {
          if (cs_wave) wg_tile_ksplit<true>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, wave);
          else wg_tile_ksplit<false>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, wave);
        } else if (cs_wave) {  // wave 7: at most 3 taps (host checks ntaps < 32)
          if (nt == 3) wg_tile<3, true>(a3, t3, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
          else if (nt == 2) wg_tile<2, true>(a2, t2, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
          else if (nt == 1) wg_tile<1, true>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
          else wg_tile<0, true>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
        } else {
          if (nt == 4) wg_tile<4, false>(acc, toff, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
          else if (nt == 3) wg_tile<3, false>(a3, t3, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
          else if (nt == 2) wg_tile<2, false>(a2, t2, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
          else if (nt == 1) wg_tile<1, false>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, w.dbg);
        }
      }
    }
    if (next >= w.ntiles) break;
    tile = next;
  }
  if (cs_wave) cs_flush(cs_n);

  // partial slab: [tap][co 32][ci 32]; D map: col = lane&31 -> ci, row -> co
  float* out = w.part + (int64_t)split * w.split_stride + w.pair_off[pair];
  const int r = lane & 31, h = lane >> 5;
  if (ksplit) {  // fold the 8 waves' partial accumulators of the single tap through LDS
    float* red = (float*)lds;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[0][e];
    __syncthreads();
    for (int f = threadIdx.x; f < 1024; f += NT) {
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) sum += red[wv * 1024 + f];
      const int e = f >> 6, ln = f & 63;
      out[((e & 3) + 8 * (e >> 2) + 4 * (ln >> 5)) * 32 + (ln & 31)] = sum;
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    int ti = wave + 8 * t;
    if (ti >= ntaps) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int co = (e & 3) + 8 * (e >> 2) + 4 * h;
      out[((int64_t)ti * 32 + co) * 32 + r] = acc[t][e];
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient, LDS-DMA version
// Same arithmetic as k_conv_wgrad, different plumbing (what conv27.hip showed to matter): the 8 compute waves only read LDS
// and issue MFMAs; 4 loader waves (2 for the x halo image, 2 for the dY tile) fill DOUBLE-BUFFERED dense images by LDS-DMA
// (buffer_load ... lds; the hardware range check supplies zero padding and the ragged volume edge) one tile ahead, and there is
// ONE s_barrier per tile: it publishes tile i+1's images (loaders wait vmcnt(0) first) and frees tile i's buffers.
typedef __attribute__((address_space(3))) void lds_void_t;
#ifdef MI_WG2_DIAG_BAR  // ablation build: where do the waves of workgroup 0 spend their cycles? (tools/diag/wg2_diag.py)
__device__ unsigned long long g_wg2_clk[8];
#define WG2_T0() const unsigned long long tt0__ = __builtin_amdgcn_s_memtime()
#define WG2_T1(acc) (acc) += __builtin_amdgcn_s_memtime() - tt0__
#else
#define WG2_T0() do {} while (0)
#define WG2_T1(acc) do {} while (0)
#endif

template <int MAXP>
struct DmaPieces {
  int pk[MAXP];        // per piece of this wave: packed coordinates of the lane's voxel ((a << 20) | (b << 10) | c) or -1
  unsigned off[MAXP];  // ... and the byte offset of the lane's 16 bytes from the image's origin voxel (0 for lanes past the image)
};
// (A loader wave shares its SIMD's vector issue with two compute waves' MFMAs: per-piece address arithmetic is what it cannot
// afford.  Tiles whose image lies inside the tensor use off[] + a scalar base and no per-lane checks; see k_conv27.)

// x halo image: piece i (1 KiB) -> halo voxels 16 i .. 16 i + 15, lane -> voxel 16 i + (lane >> 2), 16-byte part lane & 3
template <int MAXP>
__device__ __forceinline__ void dma_init_x(DmaPieces<MAXP>& d, const Geom& g, int first, int stride, int npieces, int lane, int Hi, int Wi, int cs, int sx = 1,
                                           int slices = 0) {  // slices: depth of the image when it is not the tile's whole halo (k_conv_wgrad3's groups)
  const int hvox = (slices ? slices : g.HD) * g.HH * g.HW;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int i = first + stride * k;
    const int v = i * 16 + (lane >> 2);
    const int hd = v / (g.HH * g.HW), rem = v - hd * (g.HH * g.HW), hh = rem / g.HW, hw = rem - hh * g.HW;
    d.pk[k] = (i < npieces && v < hvox) ? ((hd << 20) | (hh << 10) | hw) : -1;
    d.off[k] = (i < npieces && v < hvox) ? (unsigned)(((sx * hd * Hi + sx * hh) * Wi + sx * hw) * cs + (lane & 3) * 8) * 2u : 0u;
  }
}
// sx, cls: image voxel u = tensor voxel sx * u + (cls bit 2 / 1 / 0 along d / h / w) (WgradArgs::sx)
template <int MAXP>
__device__ __forceinline__ void dma_issue_x(const DmaPieces<MAXP>& d, const ConvArgs& a, char* dst, int first, int stride, int npieces, int lane,
                                            int n, int d0, int h0, int w0, int src_c0, bool valid = true, int sx = 1, int cls = 0, int slices = 0) {
  const Geom& g = a.g;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const int c = src_c0 + (lane & 3) * 8;
  const int od = sx * (d0 - g.hd) + ((cls >> 2) & 1), oh = sx * (h0 - g.hh) + ((cls >> 1) & 1), ow = sx * (w0 - g.hw) + (cls & 1);
  const unsigned base = (unsigned)((((n * a.Di + od) * a.Hi + oh) * a.Wi + ow) * a.x_cs + src_c0) * 2u;  // scalar (mod 2^32)
  const bool interior = valid & (od >= 0) & (od + sx * ((slices ? slices : g.HD) - 1) < a.Di) & (oh >= 0) & (oh + sx * (g.HH - 1) < a.Hi) & (ow >= 0) &
                        (ow + sx * (g.HW - 1) < a.Wi) & (src_c0 + 32 <= a.Cin);
  if (interior) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int i = first + stride * k;
      if (i >= npieces) break;  // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(dst + i * 1024), 16, d.off[k], (int)base, 0, 0);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int i = first + stride * k;
    if (i >= npieces) break;  // wave-uniform
    const int pk = d.pk[k];
    const int gd = od + sx * (pk >> 20), gh = oh + sx * ((pk >> 10) & 1023), gw = ow + sx * (pk & 1023);
    const bool ok = valid & (pk >= 0) & ((unsigned)gd < (unsigned)a.Di) & ((unsigned)gh < (unsigned)a.Hi) & ((unsigned)gw < (unsigned)a.Wi) & (c + 8 <= a.Cin);
    const unsigned off = ok ? d.off[k] + base : 0xfffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(dst + i * 1024), 16, off, 0, 0, 0);
  }
}
// dY tile image: [TD][TH][TW] voxels x 64 B, channels y*32 ..
template <int MAXP>
__device__ __forceinline__ void dma_init_y(DmaPieces<MAXP>& d, const Geom& g, int first, int stride, int npieces, int lane, int Ho, int Wo, int cs, int sy = 1) {
  const int nvox = g.TD * g.TH * g.TW;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int i = first + stride * k;
    const int v = i * 16 + (lane >> 2);
    const int vd = v / (g.TH * g.TW), rem = v - vd * (g.TH * g.TW), vh = rem / g.TW, vw = rem - vh * g.TW;
    d.pk[k] = (i < npieces && v < nvox) ? ((vd << 20) | (vh << 10) | vw) : -1;
    d.off[k] = (i < npieces && v < nvox) ? (unsigned)(((sy * vd * Ho + sy * vh) * Wo + sy * vw) * cs + (lane & 3) * 8) * 2u : 0u;
  }
}
template <int MAXP>
__device__ __forceinline__ void dma_issue_y(const DmaPieces<MAXP>& d, const WgradArgs& w, char* dst, int first, int stride, int npieces, int lane,
                                            int y, int n, int d0, int h0, int w0, bool valid = true, int sy = 1, int phase = 0) {
  const ConvArgs& a = w.c;
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(w.dy, w.dy_bytes);
  const int co = y * 32 + (lane & 3) * 8;
  const Geom& g = a.g;
  const int yd = sy * d0 + ((phase >> 2) & 1), yh = sy * h0 + ((phase >> 1) & 1), yw = sy * w0 + (phase & 1);  // dY voxel of the tile origin
  const unsigned base = (unsigned)((((n * a.Do + yd) * a.Ho + yh) * a.Wo + yw) * w.dy_cs + y * 32) * 2u;  // scalar (mod 2^32)
  const bool interior = valid & (yd + sy * (g.TD - 1) < a.Do) & (yh + sy * (g.TH - 1) < a.Ho) & (yw + sy * (g.TW - 1) < a.Wo) & (y * 32 + 32 <= a.Cout);
  if (interior) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int i = first + stride * k;
      if (i >= npieces) break;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_void_t*)(dst + i * 1024), 16, d.off[k], (int)base, 0, 0);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int i = first + stride * k;
    if (i >= npieces) break;
    const int pk = d.pk[k];
    const int od = yd + sy * (pk >> 20), oh = yh + sy * ((pk >> 10) & 1023), ow = yw + sy * (pk & 1023);
    const bool ok = valid & (pk >= 0) & (od < a.Do) & (oh < a.Ho) & (ow < a.Wo) & (co + 8 <= a.Cout);
    const unsigned off = ok ? d.off[k] + base : 0xfffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_void_t*)(dst + i * 1024), 16, off, 0, 0, 0);
  }
}

// Tile hand-off by counters in LDS instead of a workgroup barrier per tile (w.flags).  The barrier made all eight compute waves wait
// for the slowest of them at every tile (measured: 15 % of the kernel inside it, profiles/r02f_wgrad2_barrier_cycles.log) although
// nothing couples them: a compute wave only needs "tile i has landed" (ready[slot], +1 per loader wave behind its vmcnt wait) and a
// loader only needs "all eight compute waves are done with the tile whose slot I am about to refill" (freed[slot], +1 per compute
// wave).  LDS operations of all waves go through one in-order pipeline, so a counter update issued after the data is in LDS (loader:
// after its vmcnt wait) or after the reads have returned (compute: its MFMAs consumed them) is seen no earlier than that data.
__device__ __forceinline__ void flag_wait(volatile unsigned* f, unsigned target) {
  while (*f < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void flag_bump(unsigned* f, int lane) {
  if (lane == 0) atomicAdd(f, 1u);
}

template <bool GEO3D>
__global__ void __launch_bounds__(768, 3) k_conv_wgrad2(WgradArgs w) {
  constexpr int MAXT = 4, MAXPX = 10, MAXPY = 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const ConvArgs& a = w.c;
  const Geom& g = a.g;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nvox = g.TD * g.TH * g.TW;
  const int px = (g.lds_bytes + 1023) >> 10, py = (nvox * 64 + 1023) >> 10;  // 1-KiB pieces per image
  const int XB = px << 10, YB = py << 10;
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned lds_base = (unsigned)(size_t)(lds_char*)lds;
  unsigned* const fl_ready = (unsigned*)(lds + w.nbuf * (XB + YB));  // [4] tiles landed per slot (x 4 loader waves), [4] freed (x 8 compute waves)
  unsigned* const fl_freed = fl_ready + 4;
  const bool flags = w.flags != 0;
  if (flags && threadIdx.x < 8) fl_ready[threadIdx.x] = 0u;  // (ordered before any use by the prologue barrier)
  int pair, split;
  if (w.nsplit >= 8) {  // XCD-aware placement: see k_conv_wgrad
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair = slot % w.npairs;
    split = (slot / w.npairs) * 8 + xcd;
  } else {
    pair = blockIdx.x / w.nsplit;
    split = blockIdx.x % w.nsplit;
  }
  if (split >= w.nsplit || pair >= w.npairs || split >= w.ntiles) return;  // whole workgroup, before any barrier
  // Tile walk.  Workgroups that share blockIdx.x % 8 are observed to share an XCD (speed only).  Interleaved (split, split + nsplit,
  // ...) the 8 XCDs work on tiles t, t + 1, ..., t + 7: neighbours along W sit on different L2s and every XCD fetches its own copy of
  // the shared halo columns.  Contiguous: XCD class x owns tiles [x T/8, (x + 1) T/8) and its workgroups stride through them, so
  // the tiles in flight on one L2 are neighbours in W, H and (with the blocked order of tile_origin) D.
  int t0 = split, tstep = w.nsplit, tend = w.ntiles;
  if (w.contig && w.nsplit >= 8) {
    const int tpx = w.ntiles >> 3;
    t0 = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3) / w.npairs;
    tstep = w.nsplit >> 3;
    tend = ((blockIdx.x & 7) + 1) * tpx;
  }
  const int y = pair / a.nchunks;
  const int* hdr = a.hdr + (int64_t)pair * 4;
  const int tap_begin = hdr[0], ntaps = hdr[1], src_c0 = hdr[2];
  const int cls = (w.sx > 1 || w.sy > 1) ? hdr[3] & 7 : 0;  // class of the x image / phase of the dY image (factor-2 layers read in place)
  const bool ksplit = !GEO3D && ntaps == 1;

  if (wave >= 8) {  // ---------------------------------------------------------------- loader waves
    // every loader wave issues a quarter of the x pieces AND a quarter of the dY pieces: an LDS-DMA issue costs its SIMD ~100 cycles
    // that the compute waves there cannot use, so the four SIMDs carry the same load
    const int first = wave - 8;
    DmaPieces<MAXPX> dx;
    DmaPieces<MAXPY> dyp;
    dma_init_x<MAXPX>(dx, g, first, 4, px, lane, a.Hi, a.Wi, a.x_cs, w.sx);
    dma_init_y<MAXPY>(dyp, g, first, 4, py, lane, a.Ho, a.Wo, w.dy_cs, w.sy);
    // ring of NB image pairs (w.nbuf: 2, or 4 for the 1x1 k-split pairs whose tiles are all loads and hardly any MFMA): tile i
    // lives in slot i % NB and is requested NB-1 tiles ahead.  Requests beyond the last tile are still issued (all lanes out of
    // range: zeros into a free slot) so that the counted vmcnt below stays exact.
    const int NB = w.nbuf;
    TileWalk walk;  // requests go t0, t0 + tstep, ... in order: stepped, not divided (conv_common.h)
    walk_init(walk, g, t0, tstep);
    bool first_request = true;
    auto request = [&](int t, int slot) {
      int n, d0, h0, w0;
      const bool valid = t < tend;
      if (!first_request) walk_step(walk, g);
      first_request = false;
      walk_origin(walk, g, n, d0, h0, w0);  // (past the end: every lane is masked, the origin does not matter)
      dma_issue_x<MAXPX>(dx, a, lds + slot * XB, first, 4, px, lane, n, d0, h0, w0, src_c0, valid, w.sx, w.sx > 1 ? cls : 0);
      dma_issue_y<MAXPY>(dyp, w, lds + NB * XB + slot * YB, first, 4, py, lane, y, n, d0, h0, w0, valid, w.sy, w.sy > 1 ? cls : 0);
    };
    auto wait_next = [&]() {  // everything but the NB-2 youngest tiles of this wave has landed (4 + 4 pieces per tile and wave when NB = 4)
      if (NB == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int tile = t0, slot = 0;
    for (int k = 0; k < NB - 1; ++k) request(tile + k * tstep, k);
    if (NB == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // prologue: tile 0 has landed
    if (flags) flag_bump(fl_ready + 0, lane);
    int it = 0;  // index of the current tile in this workgroup's sequence
    [[maybe_unused]] unsigned long long lseg[3] = {0, 0, 0}, lt0 = 0;
#ifdef MI_WG2_DIAG_BAR
    lt0 = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
      const int next = tile + tstep;
      const int free_slot = slot == 0 ? NB - 1 : slot - 1;  // released by the barrier that ended the previous iteration
      if (flags && it > 0) {  // free_slot held tile it-1, its ((it-1)/NB + 1)-th tenant: all eight compute waves must have left it
        WG2_T0(); flag_wait(fl_freed + free_slot, 8u * (unsigned)((it - 1) / NB + 1)); WG2_T1(lseg[2]);
      }
      { WG2_T0(); request(tile + (NB - 1) * tstep, free_slot); WG2_T1(lseg[0]); }
      { WG2_T0(); wait_next(); WG2_T1(lseg[1]); }  // tile it+1 has landed
      if (flags) flag_bump(fl_ready + (slot + 1 == NB ? 0 : slot + 1), lane);
      else { WG2_T0(); __builtin_amdgcn_s_barrier(); WG2_T1(lseg[2]); }
      if (next >= tend) break;
      tile = next;
      slot = slot + 1 == NB ? 0 : slot + 1;
      ++it;
    }
    if (flags) __builtin_amdgcn_s_barrier();  // every compute wave has finished its last tile (the images are dead after this)
#ifdef MI_WG2_DIAG_BAR
    if (blockIdx.x == 0 && wave == 8 && lane == 0) {
      g_wg2_clk[0] = __builtin_amdgcn_s_memtime() - lt0;
      g_wg2_clk[1] = lseg[0]; g_wg2_clk[2] = lseg[1]; g_wg2_clk[3] = lseg[2];
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ksplit) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }  // the compute waves' fold
    return;
  }

  // ------------------------------------------------------------------------------------ compute waves (as k_conv_wgrad)
  const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int kh = gq >> 1;
  const int chan_b = ((gq & 1) * 16 + pp * 4) * 2;
  f32x16 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  int toff[MAXT];
  int nt = 0;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    int ti = wave + 8 * t;
    toff[t] = a.taps[tap_begin + (ti < ntaps ? ti : 0)];
    nt += ti < ntaps ? 1 : 0;
  }
  const int dyrow = g.TW * g.vox, dyslice = g.TH * dyrow;
  int tile = t0, n, d0, h0, w0, buf = 0;
  const int NB = w.nbuf;
  TileWalk walk;
  walk_init(walk, g, t0, tstep);
  walk_origin(walk, g, n, d0, h0, w0);
  const bool do_colsum = w.colsum != nullptr && (pair % a.nchunks) < w.cs_chunks;
  const int cs_slab = split * w.cs_chunks + (pair % a.nchunks);  // (column-sum slab of this workgroup: one per split and owning chunk slot)
  const bool cs_wave = do_colsum && (ksplit || wave == 7);
  f32x16& cs = acc[3];
  auto cs_flush = [&](int img) {
    if ((lane & 31) == 0) {
      const int hh = lane >> 5;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = y * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (co < a.Cout) {
          if (w.cs_part && !ksplit) w.cs_part[((int64_t)cs_slab * a.N + img) * a.Cout + co] = cs[e];
          else atomicAdd(w.colsum + (int64_t)img * w.colsum_stride + co, cs[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.f;
  };
  int cs_n = n;
  if (cs_wave && w.cs_part && !ksplit) {  // rows of images this workgroup never reaches must read as zero
    for (int i = lane; i < a.N * 32; i += 64) {
      const int img = i >> 5, co = y * 32 + (i & 31);
      if (co < a.Cout) w.cs_part[((int64_t)cs_slab * a.N + img) * a.Cout + co] = 0.f;
    }
  }
  __builtin_amdgcn_s_barrier();  // prologue: tile 0 is in slot 0
  [[maybe_unused]] unsigned long long cbar = 0, ct0 = 0, ntl = 0;
#ifdef MI_WG2_DIAG_BAR
  ct0 = __builtin_amdgcn_s_memtime();
#endif
  int it = 0;
  while (true) {
    ++ntl;
    if (flags) { WG2_T0(); flag_wait(fl_ready + buf, 4u * (unsigned)(it / NB + 1)); WG2_T1(cbar); }
    if (cs_wave && n != cs_n) {
      cs_flush(cs_n);
      cs_n = n;
    }
    const unsigned lds0 = lds_base + buf * XB, ldy0 = lds_base + NB * XB + buf * YB;
    {
      f32x16(&a3)[3] = reinterpret_cast<f32x16(&)[3]>(acc);
      const int(&t3)[3] = reinterpret_cast<const int(&)[3]>(toff);
      if constexpr (GEO3D) {
        if (nt == 4) wg3_tile<4, false>(acc, toff, cs, lds0, ldy0, kh, q, chan_b);
        else if (cs_wave) wg3_tile<3, true>(a3, t3, cs, lds0, ldy0, kh, q, chan_b);
        else wg3_tile<3, false>(a3, t3, cs, lds0, ldy0, kh, q, chan_b);
      } else {
        f32x16(&a2)[2] = reinterpret_cast<f32x16(&)[2]>(acc);
        const int(&t2)[2] = reinterpret_cast<const int(&)[2]>(toff);
        f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc);
        const int(&t1)[1] = reinterpret_cast<const int(&)[1]>(toff);
        if (ksplit) {
          if (cs_wave) wg_tile_ksplit<true>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, wave);
          else wg_tile_ksplit<false>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, wave);
        } else if (cs_wave) {
          if (nt == 3) wg_tile<3, true>(a3, t3, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
          else if (nt == 2) wg_tile<2, true>(a2, t2, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
          else if (nt == 1) wg_tile<1, true>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
          else wg_tile<0, true>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
        } else {
          if (nt == 4) wg_tile<4, false>(acc, toff, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
          else if (nt == 3) wg_tile<3, false>(a3, t3, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
          else if (nt == 2) wg_tile<2, false>(a2, t2, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
          else if (nt == 1) wg_tile<1, false>(a1, t1, cs, g, lds0, ldy0, kh, q, chan_b, dyrow, dyslice, 0);
        }
      }
    }
    if (flags) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the tile have returned
      flag_bump(fl_freed + buf, lane);
    } else {
      WG2_T0(); __builtin_amdgcn_s_barrier(); WG2_T1(cbar);  // this tile's buffers are free; the next tile's images have landed
    }
    const int next = tile + tstep;
    if (next >= tend) break;
    tile = next;
    buf = buf + 1 == NB ? 0 : buf + 1;
    ++it;
    walk_step(walk, g);
    walk_origin(walk, g, n, d0, h0, w0);
  }
#ifdef MI_WG2_DIAG_BAR
  if (blockIdx.x == 0 && wave == 0 && lane == 0) { g_wg2_clk[4] = __builtin_amdgcn_s_memtime() - ct0; g_wg2_clk[5] = cbar; g_wg2_clk[6] = ntl; g_wg2_clk[7] = nt; }
#endif
  if (flags) __builtin_amdgcn_s_barrier();  // (pairs with the loaders': the images are dead, the k-split fold may reuse them)
  if (cs_wave) cs_flush(cs_n);

  float* out = w.part + (int64_t)split * w.split_stride + w.pair_off[pair];
  const int r = lane & 31, h = lane >> 5;
  if (ksplit) {  // fold the 8 waves' partial accumulators of the single tap through LDS (the images are dead by now)
    float* red = (float*)lds;
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[0][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int f = threadIdx.x; f < 1024; f += 512) {
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) sum += red[wv * 1024 + f];
      const int e = f >> 6, ln = f & 63;
      out[((e & 3) + 8 * (e >> 2) + 4 * (ln >> 5)) * 32 + (ln & 31)] = sum;
    }
    __builtin_amdgcn_s_barrier();
    return;
  }
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    int ti = wave + 8 * t;
    if (ti >= ntaps) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int co = (e & 3) + 8 * (e >> 2) + 4 * h;
      out[((int64_t)ti * 32 + co) * 32 + r] = acc[t][e];
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient, rolling x halo
// k3 s1 3-D layers.  k_conv_wgrad2's loader waves are its critical path (56 LDS-DMA pieces per tile at ~180 cycles each over four
// waves = 2500 cycles against ~2200 for the 54 MFMAs of a compute wave), and 40 of the 56 are the x halo image -- a third of which
// the PREVIOUS tile already had when a workgroup walks along D.  Here a workgroup owns a contiguous run of tiles in d-fastest order,
// and the halo lives in a ring of three GROUPS of 4 d-slices (25,600 B = 25 whole pieces): the tile at depth d0 reads slices
// d0-1 .. d0+4 = its first group and half of the second; the next tile along D reuses the second group as its first, so a step
// fetches ONE group (25 pieces instead of 40).  The six slices must be contiguous for the compute waves' immediates: behind the
// third group sits a copy of the first two slices of whatever group slot 0 holds (written with it: 13 more pieces every third
// step), so a tile whose groups are slots (2, 0) reads straight on.  A tile that starts a column (or the run) needs two fresh
// groups; the second one can only be requested once the previous tile has left its slot (exposed, once per column).
constexpr int WR_GROUP = 4 * WG3_XSLICE;  // 25,600 B
constexpr int WR_XRING = 88 * 1024;       // 3 groups + the 2-slice copy = 89,600 B, rounded up (the copy's 13th piece runs 512 B past it)
constexpr int WR_YB = 16 * 1024;          // dY tile image (two of them)
constexpr int WR_LDS = WR_XRING + 2 * WR_YB + 64;

#ifdef MI_WG3_DIAG_CLK  // `make diag`: 100 MHz real-time stamps of workgroup 0's phases + the first entry / last exit over all workgroups
__device__ unsigned long long g_wg3_clk[16];
__device__ unsigned long long g_wg3_span[6][1024];  // entry / exit / end of compute wave 0's tile loop / its prologue barrier, per workgroup
#define WG3_STAMP(i) do { if (blockIdx.x == 0 && lane == 0) g_wg3_clk[i] = wall_clock64(); } while (0)
#else
#define WG3_STAMP(i) do { } while (0)
#endif

__global__ void __launch_bounds__(768, 3) k_conv_wgrad3(WgradArgs w) {
  constexpr int MAXT = 4, MAXPG = 7, MAXPY = 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const ConvArgs& a = w.c;
  const Geom& g = a.g;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned lds_base = (unsigned)(size_t)(lds_char*)lds;
  unsigned* const fl_ready = (unsigned*)(lds + WR_XRING + 2 * WR_YB);  // tiles landed (x 4 loader waves); [1]: tiles left (x 8 compute waves)
  unsigned* const fl_freed = fl_ready + 1;
  if (threadIdx.x < 2) fl_ready[threadIdx.x] = 0u;  // (ordered before any use by the prologue barrier)
  int pair, split;
  if (w.nsplit >= 8) {  // XCD-aware placement: see k_conv_wgrad
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair = slot % w.npairs;
    split = (slot / w.npairs) * 8 + xcd;
  } else {
    pair = blockIdx.x / w.nsplit;
    split = blockIdx.x % w.nsplit;
  }
  if (split >= w.nsplit || pair >= w.npairs) return;  // whole workgroup, before any barrier
#ifdef MI_WG3_DIAG_CLK
  if (threadIdx.x == 0) {
    const unsigned long long t = wall_clock64();
    atomicMin(&g_wg3_clk[14], t);
    if (blockIdx.x < 1024) g_wg3_span[0][blockIdx.x] = t;
  }
#endif
  // run of tiles [t0, tend) in d-fastest order; the splits of one XCD class get neighbouring runs (their h / w halos meet in that L2)
  const int run = (w.nsplit >= 8 && (w.nsplit & 7) == 0) ? (split & 7) * (w.nsplit >> 3) + (split >> 3) : split;
  const int t0 = (int)((int64_t)run * w.ntiles / w.nsplit), tend = (int)((int64_t)(run + 1) * w.ntiles / w.nsplit);  // (nsplit <= ntiles: never empty)
  const int y = pair / a.nchunks;
  const int* hdr = a.hdr + (int64_t)pair * 4;
  const int tap_begin = hdr[0], ntaps = hdr[1], src_c0 = hdr[2];
  // digits of tile t0
  int td = t0 % g.tilesD, tw = (t0 / g.tilesD) % g.tilesW, th = (t0 / (g.tilesD * g.tilesW)) % g.tilesH, n = t0 / (g.tilesD * g.tilesW * g.tilesH);
  auto step_digits = [&]() -> bool {  // to the next tile of the run; true: it continues the column (the halo rolls)
    if (++td < g.tilesD) return true;
    td = 0;
    if (++tw == g.tilesW) { tw = 0; if (++th == g.tilesH) { th = 0; ++n; } }
    return false;
  };

  if (wave >= 8) {  // ---------------------------------------------------------------- loader waves
    const int first = wave - 8;
    if (wave == 8) WG3_STAMP(8);
    DmaPieces<MAXPG> dg;
    DmaPieces<MAXPY> dyp;
    dma_init_x<MAXPG>(dg, g, first, 4, 25, lane, a.Hi, a.Wi, a.x_cs, 1, 4);
    dma_init_y<MAXPY>(dyp, g, first, 4, 16, lane, a.Ho, a.Wo, w.dy_cs, 1);
    if (wave == 8) WG3_STAMP(9);
    auto group = [&](int slot, int dfirst) {  // slices dfirst .. dfirst+3 of the current (n, th, tw) column's halo -> group slot
      dma_issue_x<MAXPG>(dg, a, lds + slot * WR_GROUP, first, 4, 25, lane, n, dfirst + g.hd, th * g.TH, tw * g.TW, src_c0, true, 1, 0, 4);
      if (slot == 0) dma_issue_x<MAXPG>(dg, a, lds + 3 * WR_GROUP, first, 4, 13, lane, n, dfirst + g.hd, th * g.TH, tw * g.TW, src_c0, true, 1, 0, 4);
    };
    auto dytile = [&](int ybuf) {
      dma_issue_y<MAXPY>(dyp, w, lds + WR_XRING + ybuf * WR_YB, first, 4, 16, lane, y, n, td * g.TD, th * g.TH, tw * g.TW, true, 1, 0);
    };
    int G = 0, it = 0;
    group(0, td * g.TD - 1);
    group(1, td * g.TD + 3);
    dytile(0);
    if (wave == 8) WG3_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 8) WG3_STAMP(11);
    __builtin_amdgcn_s_barrier();  // prologue
    flag_bump(fl_ready, lane);
    for (int t = t0 + 1; t < tend; ++t, ++it) {  // request tile t = number it+1 of the run while number it is being consumed
      const bool rolls = step_digits();
      if (it > 0) flag_wait(fl_freed, 8u * (unsigned)it);  // tiles 0 .. it-1 have been left: group slot (G + 2) % 3 and dY buffer (it + 1) & 1 are free
      group((G + 2) % 3, rolls ? td * g.TD + 3 : td * g.TD - 1);
      dytile((it + 1) & 1);
      if (!rolls) {  // a new column: its second group goes where tile `it` still reads its first
        flag_wait(fl_freed, 8u * (unsigned)(it + 1));
        group(G % 3, td * g.TD + 3);
      }
      G += rolls ? 1 : 2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flag_bump(fl_ready, lane);
    }
    __builtin_amdgcn_s_barrier();  // every compute wave has finished its last tile
    return;
  }

  // ------------------------------------------------------------------------------------ compute waves (as k_conv_wgrad2<true>)
  if (wave == 0) WG3_STAMP(0);
  const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int kh = gq >> 1;
  const int chan_b = ((gq & 1) * 16 + pp * 4) * 2;
  f32x16 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  int toff[MAXT];
  int nt = 0;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    int ti = wave + 8 * t;
    toff[t] = a.taps[tap_begin + (ti < ntaps ? ti : 0)];
    nt += ti < ntaps ? 1 : 0;
  }
  const bool do_colsum = w.colsum != nullptr && (pair % a.nchunks) < w.cs_chunks;
  const int cs_slab = split * w.cs_chunks + (pair % a.nchunks);
  const bool cs_wave = do_colsum && wave == 7;
  f32x16& cs = acc[3];
  auto cs_flush = [&](int img) {
    if ((lane & 31) == 0) {
      const int hh = lane >> 5;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = y * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (co < a.Cout) {
          if (w.cs_part) w.cs_part[((int64_t)cs_slab * a.N + img) * a.Cout + co] = cs[e];
          else atomicAdd(w.colsum + (int64_t)img * w.colsum_stride + co, cs[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.f;
  };
  int cs_n = n;
  if (cs_wave && w.cs_part) {  // rows of images this workgroup never reaches must read as zero
    for (int i = lane; i < a.N * 32; i += 64) {
      const int img = i >> 5, co = y * 32 + (i & 31);
      if (co < a.Cout) w.cs_part[((int64_t)cs_slab * a.N + img) * a.Cout + co] = 0.f;
    }
  }
  if (wave == 0) WG3_STAMP(1);
  __builtin_amdgcn_s_barrier();  // prologue: the first tile is in group slots 0, 1 and dY buffer 0
  if (wave == 0) WG3_STAMP(2);
#ifdef MI_WG3_DIAG_CLK
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg3_span[3][blockIdx.x] = wall_clock64();
#endif
  int G = 0, it = 0;
  for (int t = t0; t < tend; ++t, ++it) {
    flag_wait(fl_ready, 4u * (unsigned)(it + 1));
    if (cs_wave && n != cs_n) {
      cs_flush(cs_n);
      cs_n = n;
    }
    const unsigned lds0 = lds_base + (G % 3) * WR_GROUP, ldy0 = lds_base + WR_XRING + (it & 1) * WR_YB;
    {
      f32x16(&a3)[3] = reinterpret_cast<f32x16(&)[3]>(acc);
      const int(&t3)[3] = reinterpret_cast<const int(&)[3]>(toff);
      if (nt == 4) wg3_tile<4, false>(acc, toff, cs, lds0, ldy0, kh, q, chan_b);
      else if (cs_wave) wg3_tile<3, true>(a3, t3, cs, lds0, ldy0, kh, q, chan_b);
      else wg3_tile<3, false>(a3, t3, cs, lds0, ldy0, kh, q, chan_b);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the tile have returned
    flag_bump(fl_freed, lane);
    if (t + 1 < tend) G += step_digits() ? 1 : 2;
  }
  if (wave == 0) WG3_STAMP(3);
#ifdef MI_WG3_DIAG_CLK
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg3_span[2][blockIdx.x] = wall_clock64();
#endif
  __builtin_amdgcn_s_barrier();  // (pairs with the loaders')
  if (wave == 0) WG3_STAMP(4);
#ifdef MI_WG3_DIAG_CLK
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg3_span[4][blockIdx.x] = wall_clock64();
#endif
  if (cs_wave) cs_flush(cs_n);
  float* out = w.part + (int64_t)split * w.split_stride + w.pair_off[pair];
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    int ti = wave + 8 * t;
    if (ti >= ntaps) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int co = (e & 3) + 8 * (e >> 2) + 4 * h;
      out[((int64_t)ti * 32 + co) * 32 + r] = acc[t][e];
    }
  }
#ifdef MI_WG3_DIAG_CLK
  if (wave == 0) WG3_STAMP(5);
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg3_span[5][blockIdx.x] = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wave == 0) WG3_STAMP(6);
  if (lane == 0) {
    const unsigned long long t = wall_clock64();
    atomicMax(&g_wg3_clk[15], t);
    if (blockIdx.x < 1024) atomicMax(&g_wg3_span[1][blockIdx.x], t);
  }
#endif
}

// Both instantiations spelled out: with only the implicit ones (the ternary / if-else in the launcher) hipcc 7.2 emitted the host stub
// of just one of them (the other stayed an undefined symbol of the shared object; no diagnostic).
template __global__ void k_conv_wgrad2<true>(WgradArgs);
template __global__ void k_conv_wgrad2<false>(WgradArgs);

// ------------------------------------------------------------------------------------------------ weight gradient of Upsample + conv
// dW'[p][t][co][ci] = sum_i dY[2i + p][co] x[i + p + t - 1][ci]  (p: output phase, t: tap of the 2x2x2 box; convph.hip), folded into the 27
// torch taps by k_wgrad_reduce_up.  Workgroup = (cout block, cin chunk) x a share of the COARSE tiles; the 8 compute waves are the 8
// box taps, each with 8 accumulators (one per phase); 4 loader waves.  Per tile the coarse x halo image is fetched ONCE (2 slots) and
// the 8 phase sub-lattices of the fine dY stream through a ring of 4 slots (LDS-DMA with doubled voxel steps: no up-sampled x, no
// re-laid-out dY in HBM); hand-off by the LDS counters of k_conv_wgrad2.  (A first version ran the phases as separate pairs of
// k_conv_wgrad2: 54 KB of images per 16 MFMAs of a wave, 4.4 us per tile of pure fetch latency: 565 us for the 64 -> 64 layer at 128^3.)
// Registers: a wave's accumulators for all 8 phases (128) + the column-sum accumulator do not fit the 168 VGPRs of a 12-wave workgroup
// (228 bytes of scratch per lane), so a workgroup takes the 4 phases of one d-parity (`half`): 64 accumulator registers, and the two
// halves of a (cout block, cin chunk) run as separate workgroups that each fetch the x halo.
constexpr int WGU_XSLOT = 40960, WGU_YSLOT = 16384, WGU_NBY = 4;
__global__ void __launch_bounds__(768, 3) k_wgrad_up(WgradArgs w) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const ConvArgs& a = w.c;
  const Geom& g = a.g;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned lds_base = (unsigned)(size_t)(lds_char*)lds;
  unsigned* const fl = (unsigned*)(lds + 2 * WGU_XSLOT + WGU_NBY * WGU_YSLOT);  // x_ready[2], x_freed[2], y_ready[4], y_freed[4]
  unsigned* const x_ready = fl, * const x_freed = fl + 2, * const y_ready = fl + 4, * const y_freed = fl + 8;
  if (threadIdx.x < 12) fl[threadIdx.x] = 0u;
  int pair, split;  // pair = ((y * nch + ch) * 2 + half)
  if (w.nsplit >= 8) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair = slot % w.npairs;
    split = (slot / w.npairs) * 8 + xcd;
  } else {
    pair = blockIdx.x / w.nsplit;
    split = blockIdx.x % w.nsplit;
  }
  if (split >= w.nsplit || pair >= w.npairs || split >= w.ntiles) return;  // whole workgroup, before any barrier
  int t0 = split, tstep = w.nsplit, tend = w.ntiles;
  if (w.contig && w.nsplit >= 8) {
    const int tpx = w.ntiles >> 3;
    t0 = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3) / w.npairs;
    tstep = w.nsplit >> 3;
    tend = ((blockIdx.x & 7) + 1) * tpx;
  }
  const int nch = a.nchunks;  // 32-channel chunks of x
  const int half = pair & 1, yc = pair >> 1;
  const int y = yc / nch, src_c0 = (yc % nch) * 32;
  const int T = (tend - t0 + tstep - 1) / tstep;  // tiles of this workgroup (>= 1)
  const int S = 4 * T;                            // steps = (tile, phase of this half)

  if (wave >= 8) {  // ---------------------------------------------------------------- loader waves
    const int first = wave - 8;
    DmaPieces<10> dx;
    DmaPieces<4> dyp;
    dma_init_x<10>(dx, g, first, 4, 40, lane, a.Hi, a.Wi, a.x_cs);
    dma_init_y<4>(dyp, g, first, 4, 16, lane, a.Ho, a.Wo, w.dy_cs, 2);
    TileWalk wx, wy;  // tiles of the x requests / of the dY requests (the dY requests run up to 3 steps ahead: maybe in the next tile)
    walk_init(wx, g, t0, tstep);
    walk_init(wy, g, t0, tstep);
    int itx = 0, ity = 0;  // tile iteration the walks stand at
    auto request_x = [&](int it) {  // (called with it = 0, 1, 2, ...)
      if (it > itx) { walk_step(wx, g); itx = it; }
      int n, d0, h0, w0;
      walk_origin(wx, g, n, d0, h0, w0);
      dma_issue_x<10>(dx, a, lds + (it & 1) * WGU_XSLOT, first, 4, 40, lane, n, d0, h0, w0, src_c0, it < T);
    };
    auto request_y = [&](int s) {  // (called with s = 0, 1, 2, ...)
      const int it = s >> 2;
      if (it > ity) { walk_step(wy, g); ity = it; }
      int n, d0, h0, w0;
      walk_origin(wy, g, n, d0, h0, w0);
      dma_issue_y<4>(dyp, w, lds + 2 * WGU_XSLOT + (s & 3) * WGU_YSLOT, first, 4, 16, lane, y, n, d0, h0, w0, s < S, 2, half * 4 + (s & 3));
    };
    request_x(0);
    request_y(0); request_y(1); request_y(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // x(0) and dY(0) have landed
    __builtin_amdgcn_s_barrier();                     // prologue (also orders the counters' zero-fill)
    flag_bump(x_ready + 0, lane);
    flag_bump(y_ready + 0, lane);
    for (int s = 0;; ++s) {
      // (1) requests of step s + 3: the dY slot's previous tenant was step s - 1; a new tile's x slot held tile it' - 2
      const int s3 = s + 3;
      bool xnew = false;
      if ((s3 & 3) == 0) {
        const int itn = s3 >> 2;
        if (itn >= 2) flag_wait(x_freed + (itn & 1), 8u * (unsigned)(itn >> 1));
        request_x(itn);
        xnew = true;
      }
      if (s3 >= WGU_NBY) flag_wait(y_freed + (s3 & 3), 8u * (unsigned)(s3 >> 2));
      request_y(s3);
      // (2) dY(s + 1) has landed: younger are dY(s + 2), dY(s + 3) and the x image if it went out with one of them
      if (xnew || ((s + 2) & 3) == 0) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (s + 1 >= S) break;
      if (((s + 1) & 3) == 0) flag_bump(x_ready + (((s + 1) >> 2) & 1), lane);  // (older than dY(s + 1): landed as well)
      flag_bump(y_ready + ((s + 1) & 3), lane);
    }
    __builtin_amdgcn_s_barrier();  // every compute wave has finished its last step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ------------------------------------------------------------------------------------ compute waves: wave = box tap t
  const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int kh = gq >> 1;
  const int chan_b = ((gq & 1) * 16 + pp * 4) * 2;
  f32x16 acc[4], cs;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) cs[e] = 0.f;
  // tap t = wave of the box, shifted by this half's d-phase
  const unsigned tapoff = (unsigned)((((wave >> 2) & 1) + half) * WG3_XSLICE + ((wave >> 1) & 1) * WG3_XROW + (wave & 1) * 64);
  // the dY column sums (bias gradient) of phase half * 4 + j are accumulated by wave j, for the pairs of the first cin chunk
  const bool do_colsum = w.colsum != nullptr && (yc % nch) == 0 && wave < 4;
  const int cs_slab = split * 8 + half * 4 + (wave & 3);
  TileWalk walk;
  walk_init(walk, g, t0, tstep);
  int n, d0, h0, w0;
  walk_origin(walk, g, n, d0, h0, w0);
  int cs_n = n;
  auto cs_flush = [&](int img) {
    if ((lane & 31) == 0) {
      const int hh = lane >> 5;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = y * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (co < a.Cout) w.cs_part[((int64_t)cs_slab * a.N + img) * a.Cout + co] = cs[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.f;
  };
  if (do_colsum) {  // rows of images this workgroup never reaches must read as zero
    for (int i = lane; i < a.N * 32; i += 64) {
      const int img = i >> 5, co = y * 32 + (i & 31);
      if (co < a.Cout) w.cs_part[((int64_t)cs_slab * a.N + img) * a.Cout + co] = 0.f;
    }
  }
  __builtin_amdgcn_s_barrier();  // prologue
  for (int it = 0; it < T; ++it) {
    if (do_colsum && n != cs_n) { cs_flush(cs_n); cs_n = n; }
    flag_wait(x_ready + (it & 1), 4u * (unsigned)((it >> 1) + 1));
    const unsigned xb = lds_base + (it & 1) * WGU_XSLOT + tapoff;
    const unsigned yb0 = lds_base + 2 * WGU_XSLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // phase half * 4 + j: (h, w) phase = j
      flag_wait(y_ready + j, 4u * (unsigned)(it + 1));
      const int one[1] = {((j >> 1) & 1) * WG3_XROW + (j & 1) * 64};
      f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc[j]);
      if (do_colsum && wave == j) wg3_tile<1, true>(a1, one, cs, xb, yb0 + j * WGU_YSLOT, kh, q, chan_b);
      else wg3_tile<1, false>(a1, one, cs, xb, yb0 + j * WGU_YSLOT, kh, q, chan_b);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      flag_bump(y_freed + j, lane);
    }
    flag_bump(x_freed + (it & 1), lane);
    if (it + 1 < T) { walk_step(walk, g); walk_origin(walk, g, n, d0, h0, w0); }
  }
  __builtin_amdgcn_s_barrier();  // (pairs with the loaders')
  if (do_colsum) cs_flush(cs_n);
  float* out = w.part + (int64_t)split * w.split_stride + ((int64_t)yc * 64 + half * 32 + wave) * 1024;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
      out[(int64_t)p * 8 * 1024 + co * 32 + r] = acc[p][e];
    }
}

// ------------------------------------------------------------------------------------------------ weight pack / unpack
// frag_items[f] = {src_tap, co0, ci0, flags}: fragment f holds A[row = co0 + r][k = ci0 + 8h + j] (kernel channel indices);
// torch weight W[Co_t][Ci_t][KT]; flags bit0: transposed (kernel-out = torch-in), i.e. dgrad.
__device__ __forceinline__ void pack_one(const float* __restrict__ w, u32x4* __restrict__ out, const int* __restrict__ items, int nfrags,
                                         u32x4* __restrict__ out2, const int* __restrict__ items2, int nfrags2, int Co_t, int Ci_t, int KT,
                                         int gid) {
  // one conv: BOTH fragment tables (forward, then data-gradient)
  int f = gid >> 6, lane = gid & 63;
  if (f >= nfrags) {
    f -= nfrags;
    if (f >= nfrags2) return;
    out = out2; items = items2;
    gid = f * 64 + lane;
  }
  const int* it = items + f * 4;
  // flags bit 1 (conv27.hip): MFMA row rho carries output channel 16h + e with rho = (e&3) + 8(e>>2) + 4h, so that a lane's 16
  // accumulator registers are 16 consecutive channels
  const int rho = lane & 31;
  const int crow = (it[3] & 2) ? 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3) : rho;
  int tap = it[0], co = it[1] + crow, ci0 = it[2] + (lane >> 5) * 8, tr = it[3] & 1;
  F8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int ci = ci0 + j;
    int o = tr ? ci : co, i = tr ? co : ci;  // torch (out, in) indices
    v.v[j] = (tap >= 0 && o < Co_t && i < Ci_t) ? w[((int64_t)o * Ci_t + i) * KT + tap] : 0.f;
  }
  out[gid] = pack8(v);
}
__global__ void __launch_bounds__(256) k_pack_weights(const float* __restrict__ w, u32x4* __restrict__ out,
                                                      const int* __restrict__ items, int nfrags, u32x4* __restrict__ out2,
                                                      const int* __restrict__ items2, int nfrags2, int Co_t, int Ci_t, int KT,
                                                      float* __restrict__ c1w, int c1n) {
  if (c1w && blockIdx.x == 0)  // single-channel convs (conv_c1.hip) use the fp32 weights as they are
    for (int i = threadIdx.x; i < c1n; i += 256) c1w[i] = w[i];
  pack_one(w, out, items, nfrags, out2, items2, nfrags2, Co_t, Ci_t, KT, blockIdx.x * 256 + threadIdx.x);
}
// every conv of a network in ONE launch (the per-conv launches are mostly launch floor: 67 convs, ~10 us each):
// The batch pack, staged through LDS.  Per fragment the gather above reads 8 floats per lane at a 108-byte stride (one tap of
// 8 input channels): 41 M weights x 2 tables take 0.4 ms, bound by the texture addresser, not by HBM.  A GROUP is the set of
// fragments that share (cout block of 32, cin block of 16) -- one per tap: its source is 32 (or, transposed for the data gradient,
// 16) contiguous runs of `ni * KT` floats, read coalesced into LDS once, then every tap's fragment is assembled from LDS and stored
// as a contiguous 1 KiB.
struct PackGroup {
  const float* w; u32x4* out;      // the conv's fp32 master weight, the fragment table (forward or data gradient) it packs into
  int o0, i0;                      // first torch (out, in) channel of the block
  int tr, perm;                    // transposed (data gradient); conv27 row permutation
  int Co_t, Ci_t, KT;
  int ntaps, tap_off;              // gtaps[tap_off + k] = {src_tap, fragment index}
  float* c1w; int c1n;             // fp32 copy for the single-channel kernels (first group of such a conv; else null)
};
__global__ void __launch_bounds__(256) k_pack_groups(const PackGroup* __restrict__ groups, const int2* __restrict__ gtaps) {
  extern __shared__ float sm[];
  const PackGroup g = groups[blockIdx.x];
  if (g.c1w)
    for (int i = threadIdx.x; i < g.c1n; i += 256) g.c1w[i] = g.w[i];
  const int no = g.tr ? 16 : 32, ni = g.tr ? 32 : 16;
  const int run = ni * g.KT, pitch = run + 1;
  const int nvalid = (g.Ci_t - g.i0 < ni ? (g.Ci_t - g.i0 > 0 ? g.Ci_t - g.i0 : 0) : ni) * g.KT;
  {  // all rows as one flat index space, 8 independent loads in flight per thread (a row-by-row loop is one memory latency per row)
    const int total = no * run;
    const float* base = g.w + ((int64_t)g.o0 * g.Ci_t + g.i0) * g.KT;
    const int64_t row_stride = (int64_t)g.Ci_t * g.KT;
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256, o = e / run, c = e - o * run;
        v[u] = (e < total && g.o0 + o < g.Co_t && c < nvalid) ? base[o * row_stride + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256, o = e / run, c = e - o * run;
        if (e < total) sm[o * pitch + c] = v[u];
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rho = lane & 31, h = lane >> 5;
  const int crow = g.perm ? 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3) : rho;
  for (int ti = wave; ti < g.ntaps; ti += 4) {
    const int2 tf = gtaps[g.tap_off + ti];
    F8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * h + j;
      const int o = g.tr ? k : crow, i = g.tr ? crow : k;
      v.v[j] = tf.x >= 0 ? sm[o * pitch + i * g.KT + tf.x] : 0.f;
    }
    g.out[(int64_t)tf.y * 64 + lane] = pack8(v);
  }
}

// Both fragment tables of a conv from ONE read of its weights.  The forward table wants (32 cout x 16 cin) blocks, the data-gradient table
// (16 cout x 32 cin) blocks of the same tensor: packed group by group every weight was read twice (441 M parameters: 2.0 ms per step of
// the latent UNet, 2.6 TB/s).  A SUPER-GROUP is the 32 x 32 (cout x cin) block of the torch weight with all its taps -- up to 110 KB of
// LDS, read with 16-byte loads where the block is 16-byte aligned -- and the (up to four) groups that live in it.
struct PackSuper {
  const float* w;
  int O0, I0, Co_t, Ci_t, KT;
  int ng;
  int g[4];  // indices into the group table
};
constexpr int kPackSuperT = 1024;
// The block is staged in LDS as bf16 (rounded on the way in: the same RNE conversion the fragments get, so the tables are bit for bit
// what the fp32-staged version wrote): 55 KB instead of 110, so TWO blocks live on a CU and one's loads run under the other's stores
// (one block per CU alternated a load phase and a store phase: 2.8 TB/s on the latent UNet's 441 M parameters).
__global__ void __launch_bounds__(kPackSuperT) k_pack_super(const PackSuper* __restrict__ supers, const PackGroup* __restrict__ groups,
                                                    const int2* __restrict__ gtaps) {
  extern __shared__ unsigned short sm16[];
  const PackSuper su = supers[blockIdx.x];
  const int run = 32 * su.KT, pitch = run + 4;  // (a multiple of 4 halves: a float4 piece is stored as one aligned 8-byte write)
  const int nci = su.Ci_t - su.I0 < 32 ? su.Ci_t - su.I0 : 32, nvalid = nci * su.KT;
  const int nco = su.Co_t - su.O0 < 32 ? su.Co_t - su.O0 : 32;
  const int64_t row_stride = (int64_t)su.Ci_t * su.KT;
  const float* base = su.w + ((int64_t)su.O0 * su.Ci_t + su.I0) * su.KT;
  const bool vec = (((uintptr_t)base & 15) == 0) && (row_stride & 3) == 0 && (nvalid & 3) == 0;
  if (vec) {
    const int q = nvalid >> 2, total = nco * q;  // float4 pieces
    for (int e0 = threadIdx.x; e0 < total; e0 += 5 * kPackSuperT) {
      f32x4 v[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int e = e0 + u * kPackSuperT, o = e / q, c4 = e - o * q;
        v[u] = e < total ? *(const f32x4*)(base + o * row_stride + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int e = e0 + u * kPackSuperT, o = e / q, c4 = e - o * q;
        if (e < total) {
          u32x2 pk = {pack2(v[u][0], v[u][1]), pack2(v[u][2], v[u][3])};
          *(u32x2*)(sm16 + o * pitch + 4 * c4) = pk;
        }
      }
    }
  } else {
    const int total = nco * nvalid;
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * kPackSuperT) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * kPackSuperT, o = e / nvalid, c = e - o * nvalid;
        v[u] = e < total ? base[o * row_stride + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * kPackSuperT, o = e / nvalid, c = e - o * nvalid;
        if (e < total) sm16[o * pitch + c] = __builtin_bit_cast(unsigned short, (bf16)v[u]);
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rho = lane & 31, h = lane >> 5;
  for (int gi = 0; gi < su.ng; ++gi) {
    const PackGroup g = groups[su.g[gi]];
    if (g.c1w)
      for (int i = threadIdx.x; i < g.c1n; i += kPackSuperT) g.c1w[i] = g.w[i];
    const int oo = g.o0 - su.O0, io = g.i0 - su.I0;  // this group's corner inside the block (0 or 16)
    const int crow = g.perm ? 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3) : rho;
    for (int ti = wave; ti < g.ntaps; ti += kPackSuperT / 64) {
      const int2 tf = gtaps[g.tap_off + ti];
      unsigned hv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        const int o = oo + (g.tr ? k : crow), i = io + (g.tr ? crow : k);
        hv[j] = (tf.x >= 0 && o < nco && i < nci) ? (unsigned)sm16[o * pitch + i * su.KT + tf.x] : 0u;
      }
      const u32x4 out = {hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
      g.out[(int64_t)tf.y * 64 + lane] = out;
    }
  }
}

// colsum[n * stride + co] += sum_split cs_part[split][n][co]   (stride 0: one row for the whole batch); block = 32 channels x 8 split groups.
// Rides as extra trailing blocks of the weight-gradient reduce launch (one launch per conv instead of two: ~50 launches of a
// 5 us latency-bound kernel per step).  part == nullptr: none.
struct CsReduce {
  const float* part;
  int nsplit, N, Cout;
  float* out;
  int stride;
  int nbx;  // blocks along the channel axis; the N images are the slow index
};
__device__ __forceinline__ void cs_reduce_block(const CsReduce& c, int b) {
  __shared__ float cs_red[8][32];
  const int co = (b % c.nbx) * 32 + (threadIdx.x & 31), n = b / c.nbx, kg = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f;
  if (co < c.Cout) {
    int k = kg;
    for (; k + 8 < c.nsplit; k += 16) {
      s0 += c.part[((int64_t)k * c.N + n) * c.Cout + co];
      s1 += c.part[((int64_t)(k + 8) * c.N + n) * c.Cout + co];
    }
    if (k < c.nsplit) s0 += c.part[((int64_t)k * c.N + n) * c.Cout + co];
  }
  cs_red[kg][threadIdx.x & 31] = s0 + s1;
  __syncthreads();
  if (kg == 0 && co < c.Cout) {
    const int l = threadIdx.x;
    const float v = ((cs_red[0][l] + cs_red[1][l]) + (cs_red[2][l] + cs_red[3][l])) + ((cs_red[4][l] + cs_red[5][l]) + (cs_red[6][l] + cs_red[7][l]));
    if (c.stride == 0) atomicAdd(c.out + co, v);  // N rows fold into one
    else c.out[(int64_t)n * c.stride + co] += v;
  }
}

// dW[co][ci][tap] += sum_split part[split][pair][ti][co_l][ci_l];  uitems[(pair, ti)] = {src_tap, co0, ci0, _}
// block = (256 / KG) elements x KG split groups: each thread sums nsplit/KG slabs with 4 independent loads in flight, LDS folds
// the groups.  (One thread per element walking all 256 slabs serially is pure load latency; with few slabs -- wide layers --
// fewer groups keep the block count and the idle threads down.)
template <int KG>
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ part, int64_t split_stride, int nsplit,
                                                      const int* __restrict__ uitems, int nitems, float* __restrict__ dw, int Co_t,
                                                      int Ci_t, int KT, CsReduce cs, int nmain) {
  constexpr int EL = 256 / KG, BPI = 1024 / EL;  // elements per block, blocks per 32x32 item
  __shared__ float red[KG][EL];
  if ((int)blockIdx.x >= nmain) { cs_reduce_block(cs, blockIdx.x - nmain); return; }  // whole block: the column-sum fold
  const int item = blockIdx.x / BPI, l = threadIdx.x % EL, e = (blockIdx.x % BPI) * EL + l, kg = threadIdx.x / EL;
  const float* p = part + (int64_t)item * 1024 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = kg;
  for (; k + 3 * KG < nsplit; k += 4 * KG) {
    s0 += p[(int64_t)k * split_stride];
    s1 += p[(int64_t)(k + KG) * split_stride];
    s2 += p[(int64_t)(k + 2 * KG) * split_stride];
    s3 += p[(int64_t)(k + 3 * KG) * split_stride];
  }
  for (; k < nsplit; k += KG) s0 += p[(int64_t)k * split_stride];
  red[kg][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kg == 0) {
    const int* it = uitems + item * 4;
    const int tap = it[0], co = it[1] + (e >> 5), ci = it[2] + (e & 31);
    if (tap >= 0 && co < Co_t && ci < Ci_t) {
      float v = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < KG; ++g2) v += red[g2][l];
      dw[((int64_t)co * Ci_t + ci) * KT + tap] += v;
    }
  }
}
// Few slabs, wide layers (256 -> 256: 1.8 M weights, 4 slabs): the element-per-thread kernel above spends its time in the scattered
// read-modify-write of dw (one 4-byte element per 108-byte stride).  Here a block owns ONE output row of a (cout, cin) pair with all
// its taps: the slab sums go through LDS and land in dw as one contiguous run of 32 * KT floats.
// RR rows per block (round 3, late): with one row per block a 512 -> 512 layer launches 8192 blocks of ~3 loads per thread each -- the
// launch is bound by block turnover and load latency (1.5 TB/s on the latent UNet's layers); RR rows share the block's fixed costs, put RR
// independent loads in flight per thread and read RR * 128 contiguous bytes per (tap, split).
template <int RR>
__global__ void __launch_bounds__(256) k_wgrad_reduce_rows(const float* __restrict__ part, int64_t split_stride, int nsplit,
                                                           const int* __restrict__ uitems, const int* __restrict__ pair_off, int npairs,
                                                           float* __restrict__ dw, int Co_t, int Ci_t, int KT, CsReduce cs, int nmain) {
  __shared__ float sm[RR][32][33];  // [row][ci_l][tap]  (KT <= 32)
  __shared__ unsigned present;      // taps this pair owns (strided convs split the taps over several pairs)
  if ((int)blockIdx.x >= nmain) { cs_reduce_block(cs, blockIdx.x - nmain); return; }  // whole block: the column-sum fold
  constexpr int GPP = 32 / RR;  // row groups per pair
  const int pair = blockIdx.x / GPP, co_l0 = (blockIdx.x % GPP) * RR;
  const int off = pair_off[pair];
  const int nt = (int)(((pair + 1 < npairs ? (int64_t)pair_off[pair + 1] : split_stride) - off) >> 10);
  const int* it0 = uitems + (off >> 10) * 4;
  const int co0 = it0[1] + co_l0, ci0 = it0[2];
  if (co0 >= Co_t) return;  // whole block
  if (threadIdx.x == 0) present = 0u;
  __syncthreads();
  const int ci_l = threadIdx.x & 31;
  for (int t = threadIdx.x >> 5; t < nt; t += 8) {
    const int tap = it0[t * 4];
    if (tap < 0) continue;
    const float* p = part + off + t * 1024 + co_l0 * 32 + ci_l;
    float s[RR][4];
#pragma unroll
    for (int r = 0; r < RR; ++r) s[r][0] = s[r][1] = s[r][2] = s[r][3] = 0.f;
    int k = 0;
    for (; k + 3 < nsplit; k += 4) {  // (four running sums per row, folded pairwise: the order the one-row kernel used)
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        s[r][0] += p[(int64_t)k * split_stride + r * 32];
        s[r][1] += p[(int64_t)(k + 1) * split_stride + r * 32];
        s[r][2] += p[(int64_t)(k + 2) * split_stride + r * 32];
        s[r][3] += p[(int64_t)(k + 3) * split_stride + r * 32];
      }
    }
    for (; k < nsplit; ++k)
#pragma unroll
      for (int r = 0; r < RR; ++r) s[r][0] += p[(int64_t)k * split_stride + r * 32];
#pragma unroll
    for (int r = 0; r < RR; ++r) sm[r][ci_l][tap] = (s[r][0] + s[r][1]) + (s[r][2] + s[r][3]);
    if (ci_l == 0) atomicOr(&present, 1u << tap);
  }
  __syncthreads();
  const unsigned mask = present;
  const int nci = Ci_t - ci0 < 32 ? Ci_t - ci0 : 32;
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    if (co0 + r >= Co_t) break;
    float* row = dw + ((int64_t)(co0 + r) * Ci_t + ci0) * KT;
    for (int e = threadIdx.x; e < nci * KT; e += 256) {
      const int c = e / KT, tap = e - c * KT;
      if (mask >> tap & 1u) row[e] += sm[r][c][tap];
    }
  }
}

// Upsample + conv: k_wgrad_up produced dW'[phase p][box tap t] (slab item (pair * 8 + p) * 8 + t); torch tap k = (kd, kh, kw) is the sum
// of the 8 (p, t) whose tap set holds it -- per axis k = 0: (p,t) in {(0,0), (1,0)}, k = 1: {(0,1), (1,0)}, k = 2: {(0,1), (1,1)} (the sets
// S(p,t) of convph.hip read backwards).  A block owns one output row of a (cout block, cin chunk) pair: 8 thread groups split the
// slabs, each thread sums its share of the row's 64 items for one input channel (64 independent accumulators), LDS folds the groups,
// and the 27 taps leave as one contiguous run of 32 * 27 floats.
__global__ void __launch_bounds__(256) k_wgrad_reduce_up(const float* __restrict__ part, int64_t split_stride, int nsplit, int nch, float* __restrict__ dw,
                                                         int Co_t, int Ci_t, CsReduce cs, int nmain) {
  __shared__ float red[8][16][33];
  __shared__ float items[64][33];
  __shared__ float outp[32][28];
  if ((int)blockIdx.x >= nmain) { cs_reduce_block(cs, blockIdx.x - nmain); return; }
  const int yc = blockIdx.x >> 5, co_l = blockIdx.x & 31;
  const int y = yc / nch, ch = yc % nch;
  const int co = y * 32 + co_l, ci0 = ch * 32;
  if (co >= Co_t) return;  // whole block
  const int ci_l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const float* base = part + (int64_t)yc * 64 * 1024 + co_l * 32 + ci_l;
  for (int i0 = 0; i0 < 64; i0 += 16) {  // 16 items at a time (LDS)
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
    for (int sp = grp; sp < nsplit; sp += 8) {
      const float* q = base + (int64_t)sp * split_stride + (int64_t)i0 * 1024;
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] += q[i * 1024];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[grp][i][ci_l] = a[i];
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 32; e += 256) {
      const int i = e >> 5, c = e & 31;
      items[i0 + i][c] = ((red[0][i][c] + red[1][i][c]) + (red[2][i][c] + red[3][i][c])) + ((red[4][i][c] + red[5][i][c]) + (red[6][i][c] + red[7][i][c]));
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < 27 * 32; e += 256) {
    const int k = e >> 5, c = e & 31;
    const int ka[3] = {k / 9, (k / 3) % 3, k % 3};
    float v = 0.f;
    for (int m = 0; m < 8; ++m) {  // one of the two (p, t) options per axis
      int p = 0, t = 0;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const int pa = (m >> (2 - ax)) & 1, ta = ka[ax] == 0 ? 0 : (ka[ax] == 1 ? 1 - pa : 1);  // k in S(pa, ta)
        p = p * 2 + pa; t = t * 2 + ta;
      }
      v += items[p * 8 + t][c];
    }
    outp[c][k] = v;
  }
  __syncthreads();
  const int nci = Ci_t - ci0 < 32 ? Ci_t - ci0 : 32;
  float* row = dw + ((int64_t)co * Ci_t + ci0) * 27;
  for (int e = threadIdx.x; e < nci * 27; e += 256) row[e] += outp[e / 27][e % 27];
}

int env_int(const char* name, int dflt);
// cs: the conv's column-sum partials to fold in the same launch (cs.part == nullptr: none)
inline void launch_wgrad_reduce(const float* part, int64_t split_stride, int nsplit, const int* uitems, int nitems, float* dw, int Co_t, int Ci_t,
                                int KT, const int* pair_off, int npairs, CsReduce cs, hipStream_t st) {
  static const int rows_kernel = env_int("MI_WGRAD_REDUCE_ROWS", 1);
  cs.nbx = (cs.Cout + 31) / 32;
  const int extra = cs.part ? cs.nbx * cs.N : 0;
  if (nsplit >= 32)
    hipLaunchKernelGGL(k_wgrad_reduce<8>, dim3(nitems * 32 + extra), dim3(256), 0, st, part, split_stride, nsplit, uitems, nitems, dw, Co_t, Ci_t, KT,
                       cs, nitems * 32);
  else if (rows_kernel && KT <= 32 && npairs >= 8) {
    // rows of a (cout block, cin chunk) pair per block.  Same-box (profiles/r04a_ab_reduce_rr*.log): the latent UNet (256-576 pairs per layer:
    // 8-18 k one-row blocks) 56.25 ms with 1, 55.85 with 2, 55.74 with 4, 56.17 with 8; the C4 net (16-64 pairs) 21.37 with 1, 21.49 with 4:
    // few pairs need every block they can get.  MI_WGRAD_REDUCE_RR overrides.
    static const int rr_env = env_int("MI_WGRAD_REDUCE_RR", 0);
    const int rr = rr_env > 0 ? rr_env : (npairs >= 128 ? 4 : 1);
    auto kr = rr >= 8 ? k_wgrad_reduce_rows<8> : rr >= 4 ? k_wgrad_reduce_rows<4> : rr >= 2 ? k_wgrad_reduce_rows<2> : k_wgrad_reduce_rows<1>;
    const int gpp = 32 / (rr >= 8 ? 8 : rr >= 4 ? 4 : rr >= 2 ? 2 : 1);
    hipLaunchKernelGGL(kr, dim3(npairs * gpp + extra), dim3(256), 0, st, part, split_stride, nsplit, uitems, pair_off, npairs, dw, Co_t, Ci_t, KT,
                       cs, npairs * gpp);
  }
  else if (nsplit >= 8)
    hipLaunchKernelGGL(k_wgrad_reduce<4>, dim3(nitems * 16 + extra), dim3(256), 0, st, part, split_stride, nsplit, uitems, nitems, dw, Co_t, Ci_t, KT,
                       cs, nitems * 16);
  else
    hipLaunchKernelGGL(k_wgrad_reduce<1>, dim3(nitems * 4 + extra), dim3(256), 0, st, part, split_stride, nsplit, uitems, nitems, dw, Co_t, Ci_t, KT,
                       cs, nitems * 4);
}

// ------------------------------------------------------------------------------------------------ host-side plan
struct AxisCombo { int q, delta, t; };

struct Tables {
  std::vector<int> hdr, taps, frag_items;  // kernel tables + pack items
  int ny = 0, nchunks = 0, nfrags = 0;
  int* d_hdr = nullptr; int* d_taps = nullptr; int* d_items = nullptr;
  u32x4* d_wpk = nullptr;
};

int rup(int v, int m) { return (v + m - 1) / m * m; }

Geom make_geom(int VB, int Do, int Ho, int Wo, const int halo[3], int N, int vox = VOXB) {
  Geom g;
  g.vox = vox;
  // tile = 4*VB blocks of 4x8 voxels (VB blocks per wave); single-slice volumes (2-D nets) put all blocks in one slice
  if (Do == 1) { g.TD = 1; g.TH = 16; g.TW = 8 * VB; }
  else { g.TD = 2 * VB; g.TH = 8; g.TW = 8; }
  g.hd = halo[0]; g.hh = halo[1]; g.hw = halo[2];
  g.HD = g.TD + 2 * g.hd; g.HH = g.TH + 2 * g.hh; g.HW = g.TW + 2 * g.hw;
  int row = g.HW * vox;
  int rr = rup(row, 256) + 128;           // == 128 (mod 256), >= row  (may overshoot by < 256)
  if (rr - 256 >= row) rr -= 256;
  // 64-byte voxels (wgrad): a transposed read covers 4 consecutive voxels = 256 contiguous bytes = every bank once,
  // whatever the alignment -> dense rows
  g.row = vox == VOXB ? rr : row;
  g.slice = g.HH * g.row;
  g.lds_bytes = g.HD * g.slice;
  g.tilesD = (Do + g.TD - 1) / g.TD; g.tilesH = (Ho + g.TH - 1) / g.TH; g.tilesW = (Wo + g.TW - 1) / g.TW;
  (void)N;
  static const int hb_env = env_int("MI_TILE_HB", 2);  // rows of tiles per h-block of the tile walk (conv_common.h: tile_origin)
  g.hb = (hb_env > 1 && g.tilesD > 1 && g.tilesH % hb_env == 0) ? hb_env : 1;
  return g;
}

}  // namespace

// One direction of a factor-2 layer on the phase kernels of convph.hip
struct PhaseSide {
  int mode = 0;            // 1 scatter / 2 gather; 0: this direction does not run there
  int ncb = 1, ny = 0, nchunks = 0, nfrags = 0;
  int Ko = 0, Ki = 0;      // kernel output / input channels
  int tr = 0;              // kernel-out = torch-in (data gradients)
  Geom g;                  // tile grid = the coarse side
  u32x4* d_w = nullptr;    // packed fragments [cout group][phase][chunk][tap][k-step][cout block]
  unsigned* d_masks = nullptr;  // [phase * 8 + tap] -> bit k set: torch tap k is summed into this fragment
  PhasePackJob* d_job = nullptr;  // one-job table for a pack of this side alone (mi_conv_pack_weights)
  const float* job_w = nullptr;   // ... the weight pointer it was written for
};

struct mi_conv_plan {
  int N, Di, Hi, Wi, Cin, Cout, k[3], s[3], p[3];
  bool up = false;         // nearest x2 interpolation in front of the k3 s1 p1 conv (mi_upconv_plan_create): Di.. coarse, Do.. = 2 Di..
  PhaseSide ph_fwd, ph_dg;
  mi_conv_plan* up_inner = nullptr;  // upconv: plain k3 s1 p1 plan on the fine grid (fallback when a tensor is not 16-byte pitched)
  bool up_wg = false;      // upconv: weight gradient by phase pairs on the coarse grid (wg tables / slabs of this plan)
  int up_nch = 0;          // ... its 32-channel input chunks
  int* d_hdr2 = nullptr;   // k3 s2 conv: weight-gradient pair headers for reading x in place (channel offset inside the class, class code)
  bool s2_wg_direct = false;
  bf16* d_xup = nullptr;   // upconv: nearest-upsampled x for up_inner
  int Do, Ho, Wo;
  int f[3], Q;            // space-to-depth factors of x for strided axes
  int Dp, Hp, Wp;         // depth-side dims of x (== Do.. for supported (k,s,p))
  int KT;
  int ncb_fwd, ncb_dg;
  Geom g_fwd, g_dg, g_wg;
  Tables fwd, dg, wg;     // wg: NCB = 1 tables on the fwd geometry (weights unused)
  std::vector<int> wg_pair_off, wg_uitems;
  int* d_pair_off = nullptr; int* d_uitems = nullptr;
  int64_t wg_split_stride = 0; int wg_nsplit = 0, wg_nitems = 0;
  float* d_part = nullptr;
  float* d_cspart = nullptr;  // [wg_nsplit][N][Cout] column-sum partials of the weight-gradient kernels
  bf16* d_xs = nullptr;   // space-to-depth image of x (strided convs)
  bf16* d_dxs = nullptr;  // depth image of dx
  bool strided = false;
  bool full27 = false;  // k3 s1 p1 on all three axes: compile-time tap nest
  // conv_c1.hip: one channel on one side (network input / output conv).  Those kernels read fp32 weights [C][27]: the pack
  // kernels copy the master weight into d_c1w
  bool c1_in = false, c1_out = false, c1_packed = false;
  float* d_c1w = nullptr;
  float* d_c1part = nullptr;  // per-workgroup slabs of the single-channel weight-gradient kernel (conv_c1.hip)
  bool v27_fwd = false, v27_dg = false;  // forward / data gradient run on conv27.hip (LDS-DMA kernel); weights packed with perm16
  bool v11_fwd = false, v11_dg = false;  // 1x1x1: forward / data gradient run on conv1x1.hip (streaming GEMM); weights packed with perm16
  // 1x1x1 convs whose weights do not fit conv1x1.hip's LDS-resident form (shortcut convs of the coarse levels: 512 -> 256 ...): forward and
  // data gradient on the NT GEMM (gemm.hip) with plain bf16 copies of the weight, [Cout][Cin] and its transpose, refreshed at pack time
  bool g11 = false;
  bf16* d_w11 = nullptr;
  bf16* d_w11t = nullptr;
  void* d_g11job = nullptr;        // one-entry job table of the single-plan pack (the per-step path is the batch)
  const float* g11job_w = nullptr;
};

namespace {

bool axis_combos_fwd(int k, int s, int p, std::vector<AxisCombo>& out, int& f) {
  out.clear();
  if (k == 3 && s == 1 && p == 1) { f = 1; for (int t = 0; t < 3; ++t) out.push_back({0, t - 1, t}); return true; }
  if (k == 1 && s == 1 && p == 0) { f = 1; out.push_back({0, 0, 0}); return true; }
  if (k == 3 && s == 2 && p == 1) { f = 2; out.push_back({0, 0, 1}); out.push_back({1, -1, 0}); out.push_back({1, 0, 2}); return true; }
  return false;
}
// dgrad: dx-depth[j][q] = sum W[t]^T dy[j + delta]
bool axis_combos_dgrad(int k, int s, int p, std::vector<AxisCombo>& out) {
  out.clear();
  if (k == 3 && s == 1 && p == 1) { for (int t = 0; t < 3; ++t) out.push_back({0, 1 - t, t}); return true; }
  if (k == 1 && s == 1 && p == 0) { out.push_back({0, 0, 0}); return true; }
  if (k == 3 && s == 2 && p == 1) { out.push_back({0, 0, 1}); out.push_back({1, 1, 0}); out.push_back({1, 0, 2}); return true; }
  return false;
}

int tap_lds_off(const Geom& g, int dd, int dh, int dw) {
  return (dd + g.hd) * g.slice + (dh + g.hh) * g.row + (dw + g.hw) * g.vox;
}

// Build tables for:  out channels = OutC (grouped in blocks of 32*NCB; group -> (qo, co-block) when out is a depth image),
// in channels = InC per parity class qi.  `combos[axis]` lists (q, delta, t); `q_on_input` tells whether q partitions the
// INPUT channels (fwd on s2d image) or the OUTPUT channels (dgrad producing a depth image).
void build_tables(Tables& T, const Geom& g, int NCB, const std::vector<AxisCombo> combos[3], const int f[3], const int k[3], bool q_on_input,
                  int InC, int OutC, bool transposed, bool perm16 = false) {
  const int Q = f[0] * f[1] * f[2];
  const int in_chunks_per_q = (InC + KC - 1) / KC;
  const int out_groups_per_q = (OutC + 32 * NCB - 1) / (32 * NCB);
  T.nchunks = q_on_input ? Q * in_chunks_per_q : in_chunks_per_q;
  T.ny = q_on_input ? out_groups_per_q : Q * out_groups_per_q;
  T.hdr.assign((size_t)T.ny * T.nchunks * 4, 0);
  T.taps.clear(); T.frag_items.clear();
  int nfrag = 0;
  for (int y = 0; y < T.ny; ++y)
    for (int ch = 0; ch < T.nchunks; ++ch) {
      int q = q_on_input ? ch / in_chunks_per_q : y / out_groups_per_q;
      int cin0 = (q_on_input ? ch % in_chunks_per_q : ch) * KC;   // kernel-input channel within the class
      int cog = q_on_input ? y : y % out_groups_per_q;
      int qd = q / (f[1] * f[2]), qh = (q / f[2]) % f[1], qw = q % f[2];
      int* H = &T.hdr[((size_t)y * T.nchunks + ch) * 4];
      H[0] = (int)T.taps.size();
      H[2] = q_on_input ? q * InC + cin0 : cin0;  // source channel offset in the tensor the loader reads
      H[3] = nfrag;
      int nt = 0;
      for (auto& cd : combos[0]) { if (cd.q != qd) continue;
        for (auto& chh : combos[1]) { if (chh.q != qh) continue;
          for (auto& cw : combos[2]) { if (cw.q != qw) continue;
            T.taps.push_back(tap_lds_off(g, cd.delta, chh.delta, cw.delta));
            int src_tap = (cd.t * k[1] + chh.t) * k[2] + cw.t;
            for (int ks = 0; ks < 2; ++ks)
              for (int cb = 0; cb < NCB; ++cb) {
                T.frag_items.push_back(src_tap);
                T.frag_items.push_back((cog * NCB + cb) * 32);
                T.frag_items.push_back(cin0 + ks * 16);
                T.frag_items.push_back((transposed ? 1 : 0) | (perm16 ? 2 : 0));
                ++nfrag;
              }
            ++nt;
          } } }
      H[1] = nt;
    }
  T.nfrags = nfrag;
}

int upload(const std::vector<int>& v, int** d) {
  size_t bytes = (v.size() ? v.size() : 1) * sizeof(int);
  hipError_t e = hipMalloc((void**)d, bytes);
  if (e != hipSuccess) return (int)e;
  if (v.size()) e = hipMemcpy(*d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
  return (int)e;
}
int upload_tables(Tables& T, bool with_weights) {
  int e;
  if ((e = upload(T.hdr, &T.d_hdr))) return e;
  if ((e = upload(T.taps, &T.d_taps))) return e;
  if ((e = upload(T.frag_items, &T.d_items))) return e;
  if (with_weights) {
    hipError_t he = hipMalloc((void**)&T.d_wpk, (size_t)(T.nfrags ? T.nfrags : 1) * 1024);
    if (he != hipSuccess) return (int)he;
  }
  return 0;
}
void free_tables(Tables& T) {
  if (T.d_hdr) (void)hipFree(T.d_hdr);
  if (T.d_taps) (void)hipFree(T.d_taps);
  if (T.d_items) (void)hipFree(T.d_items);
  if (T.d_wpk) (void)hipFree(T.d_wpk);
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

template <int NCB, int VB, int RING, int WPS, int MODE>
int launch_igemm(ConvArgs a, int ntiles, int ny, hipStream_t st) {
  if (MODE != 0 && (a.g.row != F27_ROW || a.g.slice != F27_SLICE || a.g.TD != 4 || a.g.TH != 8 || a.g.TW != 8)) return MI_ERR_BAD_ARG;
  const int hv = a.g.HD * a.g.HH * a.g.HW;
  const int np = (hv * 4 + 255) / 256;
  a.ntiles = ntiles;
  static const int dbg = mi_diag_knob("MI_IGEMM_DBG");
  a.dbg = dbg;
  // persistent grid: ~2 workgroups per CU in total, a multiple of 8 per cout group (one slot set per XCD residue class)
  int gx = (256 * WPS / ny + 7) / 8 * 8;
  if (gx < 8) gx = 8;
  int need = (ntiles + 7) / 8 * 8;
  if (gx > need) gx = need;
  dim3 grid(gx, ny), blk(256);
  size_t lds = (size_t)a.g.lds_bytes;
#define MI_LAUNCH_NP(NPV)                                                                                  \
  do {                                                                                                     \
    auto kern = k_conv_igemm<NCB, VB, NPV, RING, WPS, MODE>;                                               \
    static int lds_ok = 0; /* raise the dynamic-LDS limit once per instantiation (not a stream op) */      \
    if ((int)lds > lds_ok) {                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return (int)e;                                                                  \
      lds_ok = 160 * 1024;                                                                                 \
    }                                                                                                      \
    hipLaunchKernelGGL(kern, grid, blk, lds, st, a);                                                       \
  } while (0)
  if constexpr (MODE != 0) {  // fixed 4x8x8 (+halo) geometry: 2400 pieces -> 10 per thread
    if (np != 10) return MI_ERR_BAD_ARG;
    MI_LAUNCH_NP(10);
  } else {
    if (np <= 4) MI_LAUNCH_NP(4);
    else if (np <= 6) MI_LAUNCH_NP(6);
    else if (np <= 10) MI_LAUNCH_NP(10);
    else if (np <= 16) MI_LAUNCH_NP(16);
    else return MI_ERR_BAD_ARG;
  }
#undef MI_LAUNCH_NP
  MI_CHECK_LAUNCH();
  return 0;
}

int launch_igemm_any(const ConvArgs& a, int NCB, int mode, int ntiles, int ny, hipStream_t st) {
  static const int wps2 = env_int("MI_CONV_NCB2_WPS", 2);  // tuning knob for the 64-cout-per-workgroup variant
  if (mode == 1) {
    if (NCB == 2) return wps2 == 1 ? launch_igemm<2, 2, 3, 1, 1>(a, ntiles, ny, st) : launch_igemm<2, 2, 2, 2, 1>(a, ntiles, ny, st);
    static const int ring1 = env_int("MI_CONV_NCB1_RING", 3);  // weight-fragment ring depth (taps) of the 32-cout variant
    return ring1 == 3 ? launch_igemm<1, 2, 3, 2, 1>(a, ntiles, ny, st) : launch_igemm<1, 2, 6, 2, 1>(a, ntiles, ny, st);
  }
  if (mode == 2) {
    if (NCB == 2) return wps2 == 1 ? launch_igemm<2, 2, 3, 1, 2>(a, ntiles, ny, st) : launch_igemm<2, 2, 2, 2, 2>(a, ntiles, ny, st);
    static const int ring1 = env_int("MI_CONV_NCB1_RING", 3);
    return ring1 == 3 ? launch_igemm<1, 2, 3, 2, 2>(a, ntiles, ny, st) : launch_igemm<1, 2, 6, 2, 2>(a, ntiles, ny, st);
  }
  if (NCB == 2) return launch_igemm<2, 2, 2, 2, 0>(a, ntiles, ny, st);
  return launch_igemm<1, 2, 4, 2, 0>(a, ntiles, ny, st);
}

// ---- phase convolutions (convph.hip) -----------------------------------------------------------------------------------------------
enum PhaseKind { UC_FWD, UC_DG, S2_FWD, S2_DG };
// torch taps (bit k of 3) summed into box tap t of phase / class b along one axis (derivations: convph.hip)
unsigned axis_set(PhaseKind kind, int b, int t) {
  static const unsigned S[2][2] = {{1u, 6u}, {3u, 4u}};  // S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
  switch (kind) {
    case UC_FWD: return S[b][t];
    case UC_DG: return S[b][1 - t];
    case S2_FWD: return b ? (t ? 4u : 1u) : (t ? 0u : 2u);
    case S2_DG: return b ? (t ? 1u : 4u) : (t ? 2u : 0u);
  }
  return 0u;
}
int phase_side_create(PhaseSide& ps, PhaseKind kind, int mode, int tr, int Ko, int Ki, int gd, int gh, int gw, int N) {
  const int halo[3] = {1, 1, 1};
  ps.g = make_geom(2, gd, gh, gw, halo, N);
  if (ps.g.TD != 4 || ps.g.TH != 8 || ps.g.TW != 8) return MI_ERR_UNSUPPORTED;
  ps.Ko = Ko; ps.Ki = Ki; ps.tr = tr;
  ps.ncb = Ko > 32 ? 2 : 1;
  const int64_t tiles = (int64_t)N * ps.g.tilesD * ps.g.tilesH * ps.g.tilesW;
  if (tiles * ((Ko + 63) / 64) <= 128) ps.ncb = 1;  // few tiles: 32 output channels per workgroup fill more CUs
  ps.ny = (Ko + 32 * ps.ncb - 1) / (32 * ps.ncb);
  ps.nchunks = (Ki + 31) / 32;
  ps.nfrags = ps.ny * 8 * ps.nchunks * 8 * 2 * ps.ncb;
  unsigned masks[64];
  for (int pc = 0; pc < 8; ++pc)
    for (int t = 0; t < 8; ++t) {
      const unsigned sd = axis_set(kind, (pc >> 2) & 1, (t >> 2) & 1), sh = axis_set(kind, (pc >> 1) & 1, (t >> 1) & 1), sw = axis_set(kind, pc & 1, t & 1);
      unsigned m = 0;
      for (int kd = 0; kd < 3; ++kd)
        for (int kh = 0; kh < 3; ++kh)
          for (int kw = 0; kw < 3; ++kw)
            if (((sd >> kd) & 1) && ((sh >> kh) & 1) && ((sw >> kw) & 1)) m |= 1u << ((kd * 3 + kh) * 3 + kw);
      masks[pc * 8 + t] = m;
    }
  if (hipMalloc((void**)&ps.d_masks, sizeof(masks)) != hipSuccess) return (int)hipErrorOutOfMemory;
  if (hipMemcpy(ps.d_masks, masks, sizeof(masks), hipMemcpyHostToDevice) != hipSuccess) return (int)hipErrorUnknown;
  if (hipMalloc((void**)&ps.d_w, (size_t)ps.nfrags * 1024) != hipSuccess) return (int)hipErrorOutOfMemory;
  ps.mode = mode;
  return 0;
}
void phase_side_free(PhaseSide& ps) {
  if (ps.d_w) (void)hipFree(ps.d_w);
  if (ps.d_masks) (void)hipFree(ps.d_masks);
  if (ps.d_job) (void)hipFree(ps.d_job);
  ps.d_w = nullptr; ps.d_masks = nullptr; ps.d_job = nullptr; ps.mode = 0;
}
PhasePackJob phase_job(const PhaseSide& ps, const float* w, int Co_t, int Ci_t, int block0) {
  return PhasePackJob{w, ps.d_w, ps.d_masks, ps.ncb, ps.nchunks, ps.Ko, ps.Ki, Co_t, Ci_t, ps.tr, block0};
}
// pack of one side alone (plan.pack: first use of a plan, tests); the per-step path is the batch (mi_conv_pack_batch_run)
int phase_side_pack(PhaseSide& ps, const float* w, int Co_t, int Ci_t, hipStream_t st) {
  if (!ps.mode) return 0;
  if (!ps.d_job || ps.job_w != w) {  // (synchronous upload: not during a capture -- the trainers pack through the batch there)
    if (!ps.d_job && hipMalloc((void**)&ps.d_job, sizeof(PhasePackJob)) != hipSuccess) return (int)hipErrorOutOfMemory;
    const PhasePackJob j = phase_job(ps, w, Co_t, Ci_t, 0);
    if (hipMemcpy(ps.d_job, &j, sizeof(j), hipMemcpyHostToDevice) != hipSuccess) return (int)hipErrorUnknown;
    ps.job_w = w;
  }
  return mi_launch_pack_phase(ps.d_job, 1, mi_pack_phase_blocks(ps.nfrags, ps.ncb, ps.nchunks), st);
}
// x: kernel input [N][id][ih][iw] with pitch x_cs, y: kernel output [N][od][oh][ow] with pitch y_cs (one of the two grids is twice the other)
int phase_side_run(const PhaseSide& ps, const void* x, int x_cs, int id, int ih, int iw, void* y, int y_cs, int od, int oh, int ow, int N,
                   const float* addvec, int addvec_stride, hipStream_t st) {
  if (!ps.mode || (x_cs & 7) || (y_cs & 7)) return MI_ERR_UNSUPPORTED;
  const int64_t xb = (int64_t)N * id * ih * iw * x_cs * 2, yb = (int64_t)N * od * oh * ow * y_cs * 2;
  if (xb >= (1ll << 32) || yb >= (1ll << 32) || (int64_t)ps.nfrags * 1024 >= (1ll << 32)) return MI_ERR_UNSUPPORTED;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16*)x; a.x_cs = x_cs; a.N = N; a.Di = id; a.Hi = ih; a.Wi = iw; a.Cin = ps.Ki;
  a.y = (bf16*)y; a.y_cs = y_cs; a.Cout = ps.Ko; a.Do = od; a.Ho = oh; a.Wo = ow;
  a.ogpq = ps.ny; a.outc_q = ps.Ko;
  a.wpk = ps.d_w; a.wpk_bytes = (unsigned)ps.nfrags * 1024u; a.nchunks = ps.nchunks;
  a.addvec = addvec; a.addvec_stride = addvec_stride;
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
  a.perm16 = 1;
  a.g = ps.g;
  const int ntiles = N * a.g.tilesD * a.g.tilesH * a.g.tilesW;
  return mi_launch_convph(a, ps.ncb, ps.mode, ntiles, ps.ny, st);
}

// Weight gradient of Upsample + conv on the coarse grid (k_wgrad_up): pair = (cout block y, cin chunk ch); slabs of 64 items per pair
int upconv_wgrad_tables(mi_conv_plan* P) {
  const int halo[3] = {1, 1, 1};
  P->g_wg = make_geom(2, P->Di, P->Hi, P->Wi, halo, P->N, 64);
  const Geom& g = P->g_wg;
  if (g.TD != 4 || g.TH != 8 || g.TW != 8 || g.row != WG3_XROW || g.slice != WG3_XSLICE) return MI_ERR_UNSUPPORTED;
  const int ny = (P->Cout + 31) / 32, nch = (P->Cin + 31) / 32;
  P->up_nch = nch;
  P->wg.ny = ny; P->wg.nchunks = nch;
  const int npairs = ny * nch * 2;  // (cout block, cin chunk, d-parity half of the phases)
  P->wg_split_stride = (int64_t)ny * nch * 64 * 1024;
  const int ntiles = P->N * g.tilesD * g.tilesH * g.tilesW;
  int nsplit = 1;
  {  // (the cost model of mi_conv_plan_create: rounds per XCD x (tiles per workgroup + fixed costs); a tile here is 8 phase steps)
    int64_t best = -1;
    for (int ns = 1; ns <= 256 && ns <= ntiles; ++ns) {
      const int per_xcd = ns >= 8 ? npairs * ((ns + 7) / 8) : (npairs * ns + 7) / 8;
      const int64_t rounds = (per_xcd + 31) / 32, cost = rounds * ((ntiles + ns - 1) / ns + 2);
      if (best < 0 || cost < best) { best = cost; nsplit = ns; }
    }
  }
  static const int ns_env = env_int("MI_WGU_NSPLIT", 0);
  if (ns_env > 0 && ns_env <= ntiles) nsplit = ns_env;
  while (nsplit > 1 && (int64_t)nsplit * P->wg_split_stride * 4 > (256ll << 20)) nsplit /= 2;
  P->wg_nsplit = nsplit;
  if (hipMalloc((void**)&P->d_part, (size_t)nsplit * P->wg_split_stride * 4) != hipSuccess) return (int)hipErrorOutOfMemory;
  if (hipMalloc((void**)&P->d_cspart, (size_t)nsplit * 8 * P->N * P->Cout * 4) != hipSuccess) return (int)hipErrorOutOfMemory;
  return 0;
}

int upconv_wgrad(mi_conv_plan* P, const void* x, int x_cs, const void* dy, int dy_cs, float* dw, float* dy_colsum, int dy_colsum_stride, hipStream_t st) {
  if (dy_colsum && dy_colsum_stride != 0 && dy_colsum_stride < P->Cout) return MI_ERR_BAD_ARG;
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  ConvArgs& a = w.c;
  a.x = (const bf16*)x; a.x_cs = x_cs; a.N = P->N; a.Di = P->Di; a.Hi = P->Hi; a.Wi = P->Wi; a.Cin = P->Cin;
  a.Cout = P->Cout; a.Do = P->Do; a.Ho = P->Ho; a.Wo = P->Wo;  // dY: the fine grid
  a.nchunks = P->up_nch;
  a.g = P->g_wg;
  const int64_t xb = (int64_t)P->N * P->Di * P->Hi * P->Wi * x_cs * 2, dyb = (int64_t)P->N * P->Do * P->Ho * P->Wo * dy_cs * 2;
  if (xb >= (1ll << 32) || dyb >= (1ll << 32)) return MI_ERR_UNSUPPORTED;
  a.x_bytes = (unsigned)xb;
  w.dy = (const bf16*)dy; w.dy_cs = dy_cs; w.dy_bytes = (unsigned)dyb;
  w.sx = 1; w.sy = 2; w.cs_chunks = 1;
  w.part = P->d_part; w.split_stride = P->wg_split_stride;
  w.ntiles = P->N * a.g.tilesD * a.g.tilesH * a.g.tilesW;
  w.nsplit = P->wg_nsplit;
  static const int contig_env = env_int("MI_WGRAD_CONTIG", 1);
  w.contig = contig_env && (w.ntiles % 8 == 0) && (w.nsplit % 8 == 0) && w.nsplit <= w.ntiles;
  w.colsum = dy_colsum; w.colsum_stride = dy_colsum_stride;
  w.cs_part = dy_colsum ? P->d_cspart : nullptr;
  w.npairs = P->wg.ny * P->up_nch * 2;
  const size_t lds = 2 * WGU_XSLOT + WGU_NBY * WGU_YSLOT + 64;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)k_wgrad_up, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  dim3 grid(w.nsplit >= 8 ? w.npairs * ((w.nsplit + 7) / 8) * 8 : w.npairs * w.nsplit);
  hipLaunchKernelGGL(k_wgrad_up, grid, dim3(768), lds, st, w);
  CsReduce cs{w.cs_part, P->wg_nsplit * 8, P->N, P->Cout, dy_colsum, dy_colsum_stride, 0};
  cs.nbx = (cs.Cout + 31) / 32;
  const int extra = cs.part ? cs.nbx * cs.N : 0, nmain = P->wg.ny * P->up_nch * 32;
  hipLaunchKernelGGL(k_wgrad_reduce_up, dim3(nmain + extra), dim3(256), 0, st, P->d_part, P->wg_split_stride, P->wg_nsplit, P->up_nch, dw, P->Cout,
                     P->Cin, cs, nmain);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" {

int mi_conv_plan_create(mi_conv_plan** out, int N, int Di, int Hi, int Wi, int Cin, int Cout, const int* k, const int* s, const int* p) {
  if (!out || N <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0) return MI_ERR_BAD_ARG;
  mi_conv_plan* P = new mi_conv_plan();
  P->N = N; P->Di = Di; P->Hi = Hi; P->Wi = Wi; P->Cin = Cin; P->Cout = Cout;
  std::vector<AxisCombo> cf[3], cdg[3];
  int dims[3] = {Di, Hi, Wi}, od[3], halo_f[3] = {0, 0, 0}, halo_d[3] = {0, 0, 0};
  for (int a = 0; a < 3; ++a) {
    P->k[a] = k[a]; P->s[a] = s[a]; P->p[a] = p[a];
    if (!axis_combos_fwd(k[a], s[a], p[a], cf[a], P->f[a]) || !axis_combos_dgrad(k[a], s[a], p[a], cdg[a])) {
      delete P;
      return MI_ERR_UNSUPPORTED;
    }
    od[a] = (dims[a] + 2 * p[a] - k[a]) / s[a] + 1;
    for (auto& c : cf[a]) if (c.delta != 0) halo_f[a] = 1;
    for (auto& c : cdg[a]) if (c.delta != 0) halo_d[a] = 1;
  }
  P->Do = od[0]; P->Ho = od[1]; P->Wo = od[2];
  P->Q = P->f[0] * P->f[1] * P->f[2];
  P->strided = P->Q > 1;
  P->Dp = (Di + P->f[0] - 1) / P->f[0]; P->Hp = (Hi + P->f[1] - 1) / P->f[1]; P->Wp = (Wi + P->f[2] - 1) / P->f[2];
  P->KT = k[0] * k[1] * k[2];
  P->full27 = true;
  for (int a = 0; a < 3; ++a) P->full27 = P->full27 && k[a] == 3 && s[a] == 1 && p[a] == 1;
  // parity classes start at q*Cin: 16-byte loads need Cin % 8 == 0.  A class that is not a multiple of 32 channels is
  // read together with the head of the next class; those extra k-rows meet zero weights (pack masks ci >= Cin).
  if (P->strided && (Cin % 8) != 0) { delete P; return MI_ERR_UNSUPPORTED; }
  // 64 output channels per workgroup run ~10 % more MFMA per second than 32 (fewer LDS bytes per MFMA), but a channel count that is an
  // odd multiple of 32 pads its last workgroup row with 32 dead channels: for 96 that is a third more MFMAs than needed (the data
  // gradient of the 96 -> 32 conv of the finest up-block: 434 us as 64 + 32(+32 dead), ~345 as 3 x 32)
  P->ncb_fwd = Cout > 32 && Cout != 96 ? 2 : 1;
  {  // only the 32-channel variant carries the GroupNorm sums in its epilogue (above): up to this many output channels the forward runs
     // as rows of 32 so that the consumer's statistics pass disappears (A/B knob; 32 = the plain rule)
    static const int ncb1_max = env_int("MI_NCB1_FWD_MAXC", 32);
    if (Cout <= ncb1_max && (Cout % 32) == 0) P->ncb_fwd = 1;
  }
  P->ncb_dg = Cin > 32 && Cin != 96 ? 2 : 1;
  {  // few tiles (16^3 levels): 64 output channels per workgroup would leave most CUs without a workgroup -> 32 per workgroup
    const int64_t tiles = (int64_t)N * ((od[0] + 3) / 4) * ((od[1] + 7) / 8) * ((od[2] + 7) / 8);
    if (od[0] > 1 && tiles * ((Cout + 63) / 64) <= 128) P->ncb_fwd = 1;
    if (od[0] > 1 && tiles * ((Cin + 63) / 64) <= 128) P->ncb_dg = 1;
  }
  // forward: loader reads x (or its depth image: dims Dp.., Q*Cin channels); outputs on the (Do,Ho,Wo) grid
  P->g_fwd = make_geom(2, P->Do, P->Ho, P->Wo, halo_f, N);
  static const int use27 = env_int("MI_CONV27", 1);  // 0: keep every conv on the table-driven kernel (A/B runs)
  {
    static const int use_c1 = env_int("MI_CONV_C1", 1);
    auto c_ok = [](int c) { return c >= 8 && c <= 64 && (c & 7) == 0; };
    P->c1_in = use_c1 && P->full27 && Cin == 1 && c_ok(Cout);
    P->c1_out = use_c1 && P->full27 && Cout == 1 && c_ok(Cin);
    if (P->c1_in || P->c1_out) {
      const int C = P->c1_in ? Cout : Cin;
      if (hipMalloc((void**)&P->d_c1w, (size_t)27 * C * 4) != hipSuccess) { mi_conv_plan_destroy(P); return (int)hipErrorOutOfMemory; }
      if (hipMalloc((void**)&P->d_c1part, (size_t)mi_c1_wgrad_scratch_floats(C) * 4) != hipSuccess) { mi_conv_plan_destroy(P); return (int)hipErrorOutOfMemory; }
    }
  }
  const bool geo27 = P->full27 && P->g_fwd.TD == 4 && P->g_fwd.TH == 8 && P->g_fwd.TW == 8;
  P->v27_fwd = use27 && geo27 && (Cin % 8) == 0 && (Cout % 8) == 0;  // whole channel octets on both sides (else: table-driven kernel)
  P->v27_dg = use27 && geo27 && (Cout % 8) == 0 && (Cin % 8) == 0;
  {
    static const int use11 = env_int("MI_CONV1X1", 1);
    const bool k1 = P->KT == 1 && !P->strided;
    auto chunks_ok = [](int c) { int n = (c + 31) / 32; return n == 1 || n == 2 || n == 3 || n == 4 || n == 6; };
    P->v11_fwd = use11 && k1 && (Cin % 8) == 0 && chunks_ok(Cin) && ((Cout + 31) / 32) * ((Cin + 31) / 32) * 2 <= 96;
    P->v11_dg = use11 && k1 && (Cout % 8) == 0 && chunks_ok(Cout) && ((Cout + 31) / 32) * ((Cin + 31) / 32) * 2 <= 96;
    static const int use_g11 = env_int("MI_CONV1X1_GEMM", 1);  // 0: the table-driven kernel for those layers (A/B runs)
    P->g11 = use_g11 && k1 && (!P->v11_fwd || !P->v11_dg) && (Cin % 8) == 0 && (Cout % 8) == 0;
    if (P->g11 && (hipMalloc((void**)&P->d_w11, (size_t)Cin * Cout * 2) != hipSuccess || hipMalloc((void**)&P->d_w11t, (size_t)Cin * Cout * 2) != hipSuccess)) {
      mi_conv_plan_destroy(P);
      return (int)hipErrorOutOfMemory;
    }
  }
  build_tables(P->fwd, P->g_fwd, P->ncb_fwd, cf, P->f, P->k, true, Cin, Cout, false, P->v27_fwd || P->v11_fwd);
  // dgrad: loader reads dy (Do,Ho,Wo,Cout); outputs the depth image of dx on the (Dp,Hp,Wp) grid with Q*Cin channels
  P->g_dg = make_geom(2, P->Dp, P->Hp, P->Wp, halo_d, N);
  build_tables(P->dg, P->g_dg, P->ncb_dg, cdg, P->f, P->k, false, Cout, Cin, true, P->v27_dg || P->v11_dg);
  // wgrad: forward geometry, 32-cout groups
  P->g_wg = make_geom(2, P->Do, P->Ho, P->Wo, halo_f, N, 64);
  build_tables(P->wg, P->g_wg, 1, cf, P->f, P->k, true, Cin, Cout, false);
  int e;
  if ((e = upload_tables(P->fwd, true)) || (e = upload_tables(P->dg, true)) || (e = upload_tables(P->wg, false))) { mi_conv_plan_destroy(P); return e; }
  // wgrad partial slabs: per pair, ntaps * 1024 floats
  int npairs = P->wg.ny * P->wg.nchunks;
  int64_t off = 0;
  P->wg_pair_off.resize(npairs);
  for (int pr = 0; pr < npairs; ++pr) {
    P->wg_pair_off[pr] = (int)off;
    int nt = P->wg.hdr[(size_t)pr * 4 + 1];
    int frag0 = P->wg.hdr[(size_t)pr * 4 + 3];
    for (int t = 0; t < nt; ++t) {  // item = first fragment (ks = 0, cb = 0) of the tap
      const int* it = &P->wg.frag_items[(size_t)(frag0 + t * 2) * 4];
      P->wg_uitems.push_back(it[0]); P->wg_uitems.push_back(it[1]); P->wg_uitems.push_back(it[2]); P->wg_uitems.push_back(0);
    }
    off += (int64_t)nt * 1024;
  }
  P->wg_split_stride = off;
  P->wg_nitems = (int)(off / 1024);
  int ntiles = N * P->g_wg.tilesD * P->g_wg.tilesH * P->g_wg.tilesW;
  // One workgroup per CU (112 KB of LDS each), and workgroup i of a launch goes to XCD i % 8: an XCD that is dealt more than its 32
  // CUs' worth runs a second round.  Rounding 256 / npairs UP (what this did) deals 33 workgroups to five XCDs for 3 pairs (96 -> 32),
  // and the launch takes twice as long: 554 us where 32 -> 32, a third of the work, takes 123.  Cost of a split = rounds x (tiles per
  // workgroup + 6: prologue, accumulator flush and the slab it adds to the reduce, in tile times -- three rounds of 32 tiles measured
  // 383 us against ~305 for one round of 103); ties go to the coarser split.
  int nsplit = 1;
  {
    int64_t best = -1;
    for (int ns = 1; ns <= 256 && ns <= ntiles; ++ns) {
      const int per_xcd = ns >= 8 ? npairs * ((ns + 7) / 8) : (npairs * ns + 7) / 8;  // (k_conv_wgrad[2]: XCD-aware placement from 8 splits on)
      // (+ optionally the slab round trip: every split writes its slab and the reduce kernel reads it back; MI_WGRAD_SLAB_COST =
      // hundredths of a tile time per MB and split -- 2 x 1 MB / 3 TB/s = 0.67 us = ~22; measured: no setting moves the step, default 0)
      static const int slab_cost = env_int("MI_WGRAD_SLAB_COST", 0);
      const int64_t rounds = (per_xcd + 31) / 32;
      const int64_t cost = 100 * rounds * ((ntiles + ns - 1) / ns + 6) + (int64_t)slab_cost * ns * (off * 4 / (1 << 20));
      if (best < 0 || cost < best) { best = cost; nsplit = ns; }
    }
  }
  static const int old_split = env_int("MI_WGRAD_SPLIT_OLD", 0);  // A/B knob: the round-1 choice
  if (old_split) nsplit = (256 + npairs - 1) / npairs;
  static const int force_split = env_int("MI_WGRAD_NSPLIT", 0);  // probe knob: this many splits for every layer of at most 64 tiles
  if (force_split && ntiles <= 64) nsplit = force_split;
  if (nsplit > ntiles) nsplit = ntiles;
  while (nsplit > 1 && (int64_t)nsplit * off * 4 > (256ll << 20)) nsplit /= 2;  // cap the slab at 256 MiB
  if (nsplit < 1) nsplit = 1;
  P->wg_nsplit = nsplit;
  if ((e = upload(P->wg_pair_off, &P->d_pair_off)) || (e = upload(P->wg_uitems, &P->d_uitems))) { mi_conv_plan_destroy(P); return e; }
  if (hipMalloc((void**)&P->d_part, (size_t)nsplit * off * 4) != hipSuccess) { mi_conv_plan_destroy(P); return (int)hipErrorOutOfMemory; }
  if (hipMalloc((void**)&P->d_cspart, (size_t)nsplit * N * Cout * 4) != hipSuccess) { mi_conv_plan_destroy(P); return (int)hipErrorOutOfMemory; }
  {  // k3 s2 p1 on all three axes: forward and data gradient on the phase kernels (no space-to-depth copies of x / dx)
    static const int use_ph = env_int("MI_CONVPH", 1);
    bool s2all = Di > 1;
    for (int a = 0; a < 3; ++a) s2all = s2all && k[a] == 3 && s[a] == 2 && p[a] == 1;
    if (use_ph && s2all && (Cin % 8) == 0 && (Cout % 8) == 0) {
      int e1 = phase_side_create(P->ph_fwd, S2_FWD, 2, 0, Cout, Cin, P->Do, P->Ho, P->Wo, N);
      int e2 = e1 ? e1 : phase_side_create(P->ph_dg, S2_DG, 1, 1, Cin, Cout, P->Do, P->Ho, P->Wo, N);
      if (e1 || e2) { phase_side_free(P->ph_fwd); phase_side_free(P->ph_dg); if ((e1 ? e1 : e2) != MI_ERR_UNSUPPORTED) { mi_conv_plan_destroy(P); return e1 ? e1 : e2; } }
      if (P->ph_fwd.mode && (Cin % 32) == 0) {  // weight gradient: the class images are gathered from x in place (k_conv_wgrad2, sx = 2)
        std::vector<int> h2 = P->wg.hdr;
        const int cpq = Cin / 32;  // chunks per class
        for (int pr = 0; pr < P->wg.ny * P->wg.nchunks; ++pr) {
          const int ch = pr % P->wg.nchunks, q = ch / cpq;
          h2[(size_t)pr * 4 + 2] = (ch % cpq) * 32;
          h2[(size_t)pr * 4 + 3] = q;  // (qd, qh, qw) = bits 2, 1, 0: all three factors are 2
        }
        if ((e = upload(h2, &P->d_hdr2))) { mi_conv_plan_destroy(P); return e; }
        P->s2_wg_direct = true;
      }
    }
  }
  if (P->strided) {
    size_t nb = (size_t)N * P->Dp * P->Hp * P->Wp * P->Q * Cin * 2;
    if (hipMalloc((void**)&P->d_xs, nb) != hipSuccess || hipMalloc((void**)&P->d_dxs, nb) != hipSuccess) { mi_conv_plan_destroy(P); return (int)hipErrorOutOfMemory; }
  }
  *out = P;
  return 0;
}

int mi_conv_plan_destroy(mi_conv_plan* P) {
  if (!P) return 0;
  phase_side_free(P->ph_fwd); phase_side_free(P->ph_dg);
  if (P->d_hdr2) (void)hipFree(P->d_hdr2);
  if (P->up_inner) mi_conv_plan_destroy(P->up_inner);
  if (P->d_xup) (void)hipFree(P->d_xup);
  free_tables(P->fwd); free_tables(P->dg); free_tables(P->wg);
  if (P->d_pair_off) (void)hipFree(P->d_pair_off);
  if (P->d_uitems) (void)hipFree(P->d_uitems);
  if (P->d_part) (void)hipFree(P->d_part);
  if (P->d_cspart) (void)hipFree(P->d_cspart);
  if (P->d_c1w) (void)hipFree(P->d_c1w);
  if (P->d_w11) (void)hipFree(P->d_w11);
  if (P->d_w11t) (void)hipFree(P->d_w11t);
  if (P->d_g11job) (void)hipFree(P->d_g11job);
  if (P->d_c1part) (void)hipFree(P->d_c1part);
  if (P->d_xs) (void)hipFree(P->d_xs);
  if (P->d_dxs) (void)hipFree(P->d_dxs);
  delete P;
  return 0;
}

// Upsample.forward of the reference (UNet:569-588 / AEKL Upsample): nearest x2 interpolation on all three axes, then the k3 s1 p1
// `Convolution`.  x: [N][D][H][W][Cin] coarse, y: [N][2D][2H][2W][Cout].  The plan answers mi_conv_fwd / mi_conv_dgrad / mi_conv_wgrad /
// mi_conv_pack_weights like any other (weights: the conv's torch tensor [Cout][Cin][3][3][3]).
int mi_upconv_plan_create(mi_conv_plan** out, int N, int D, int H, int W, int Cin, int Cout) {
  if (!out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return MI_ERR_BAD_ARG;
  if (D == 1) return MI_ERR_UNSUPPORTED;  // 2-D nets: upsample kernel + plain conv (engine.upsample)
  mi_conv_plan* P = new mi_conv_plan();
  P->up = true;
  P->N = N; P->Di = D; P->Hi = H; P->Wi = W; P->Cin = Cin; P->Cout = Cout;
  P->Do = 2 * D; P->Ho = 2 * H; P->Wo = 2 * W;
  for (int a = 0; a < 3; ++a) { P->k[a] = 3; P->s[a] = 1; P->p[a] = 1; P->f[a] = 1; }
  P->Q = 1; P->KT = 27; P->Dp = D; P->Hp = H; P->Wp = W;
  const int k3[3] = {3, 3, 3}, s1[3] = {1, 1, 1}, p1[3] = {1, 1, 1};
  int e = mi_conv_plan_create(&P->up_inner, N, 2 * D, 2 * H, 2 * W, Cin, Cout, k3, s1, p1);
  if (e) { P->up_inner = nullptr; mi_conv_plan_destroy(P); return e; }
  if (hipMalloc((void**)&P->d_xup, (size_t)N * 8 * D * H * W * Cin * 2) != hipSuccess) { mi_conv_plan_destroy(P); return (int)hipErrorOutOfMemory; }
  static const int use_ph = env_int("MI_CONVPH", 1);
  if (use_ph && (Cin % 8) == 0 && (Cout % 8) == 0) {
    int e1 = phase_side_create(P->ph_fwd, UC_FWD, 1, 0, Cout, Cin, D, H, W, N);
    int e2 = e1 ? e1 : phase_side_create(P->ph_dg, UC_DG, 2, 1, Cin, Cout, D, H, W, N);
    if (e1 || e2) { phase_side_free(P->ph_fwd); phase_side_free(P->ph_dg); if ((e1 ? e1 : e2) != MI_ERR_UNSUPPORTED) { mi_conv_plan_destroy(P); return e1 ? e1 : e2; } }
    if (P->ph_fwd.mode) {
      const int ew = upconv_wgrad_tables(P);
      if (ew && ew != MI_ERR_UNSUPPORTED) { mi_conv_plan_destroy(P); return ew; }
      P->up_wg = ew == 0;
    }
  }
  *out = P;
  return 0;
}

int mi_conv_plan_out_dims(const mi_conv_plan* P, int* dims3) {
  if (!P || !dims3) return MI_ERR_BAD_ARG;
  dims3[0] = P->Do; dims3[1] = P->Ho; dims3[2] = P->Wo;
  return 0;
}

// fp32 master weight [Cout][Cin][kd][kh][kw] -> packed bf16 fragments for forward and dgrad
}  // extern "C"

// bf16 copies of a 1x1 weight for the GEMM path: o[co][ci] and ot[ci][co]; one block per 32 x 32 tile, all such convs in one launch
struct G11Job { const float* w; bf16* o; bf16* ot; int Co, Ci, block0, bx; };
namespace {
__global__ void __launch_bounds__(256) k_pack_gemm11(const G11Job* __restrict__ jobs, int njobs) {
  __shared__ float t[32][33];
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].block0) ++j;
  const G11Job g = jobs[j];
  const int b = blockIdx.x - g.block0, bi = b % g.bx, bo = b / g.bx;  // tile (co block bo, ci block bi)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = bo * 32 + r, ci = bi * 32 + tx;
    const float v = (co < g.Co && ci < g.Ci) ? g.w[(int64_t)co * g.Ci + ci] : 0.f;
    t[r][tx] = v;
    if (co < g.Co && ci < g.Ci) g.o[(int64_t)co * g.Ci + ci] = f2bf(v);
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = bi * 32 + r, co = bo * 32 + tx;
    if (co < g.Co && ci < g.Ci) g.ot[(int64_t)ci * g.Co + co] = f2bf(t[tx][r]);
  }
}
G11Job g11_job(const mi_conv_plan* P, const float* w, int block0) {
  G11Job g;
  g.w = w; g.o = P->d_w11; g.ot = P->d_w11t; g.Co = P->Cout; g.Ci = P->Cin; g.block0 = block0; g.bx = (P->Cin + 31) / 32;
  return g;
}
int g11_blocks(const mi_conv_plan* P) { return ((P->Cin + 31) / 32) * ((P->Cout + 31) / 32); }
}  // namespace

extern "C" {

int mi_conv_pack_weights(mi_conv_plan* P, const float* w, hipStream_t st) {
  if (!P || !w) return MI_ERR_BAD_ARG;
  {
    int e = phase_side_pack(P->ph_fwd, w, P->Cout, P->Cin, st);
    if (!e) e = phase_side_pack(P->ph_dg, w, P->Cout, P->Cin, st);
    if (e) return e;
    if (P->up) return mi_conv_pack_weights(P->up_inner, w, st);
  }
  hipLaunchKernelGGL(k_pack_weights, dim3(((P->fwd.nfrags + P->dg.nfrags) * 64 + 255) / 256), dim3(256), 0, st, w, P->fwd.d_wpk,
                     P->fwd.d_items, P->fwd.nfrags, P->dg.d_wpk, P->dg.d_items, P->dg.nfrags, P->Cout, P->Cin, P->KT, P->d_c1w,
                     P->d_c1w ? 27 * P->Cin * P->Cout : 0);
  if (P->g11) {  // (synchronous upload of the one-entry job table: not during a capture -- the trainers pack through the batch there)
    if (!P->d_g11job || P->g11job_w != w) {
      if (!P->d_g11job && hipMalloc(&P->d_g11job, sizeof(G11Job)) != hipSuccess) return (int)hipErrorOutOfMemory;
      const G11Job h = g11_job(P, w, 0);
      if (hipMemcpy(P->d_g11job, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return (int)hipErrorUnknown;
      P->g11job_w = w;
    }
    hipLaunchKernelGGL(k_pack_gemm11, dim3(g11_blocks(P)), dim3(256), 0, st, (const G11Job*)P->d_g11job, 1);
  }
  P->c1_packed = true;
  MI_CHECK_LAUNCH();
  return 0;
}

struct mi_pack_batch {
  PackSuper* d_supers = nullptr;  // groups paired by the 32 x 32 weight block they read (k_pack_super); nsupers == 0: k_pack_groups
  int nsupers = 0;
  size_t lds_super = 0;
  G11Job* d_g11 = nullptr;  // 1x1 convs on the GEMM path: one more launch for all of them
  int ng11 = 0, g11_blocks = 0;
  PackGroup* d_groups = nullptr;
  int2* d_gtaps = nullptr;
  int ngroups = 0;
  size_t lds = 0;
  PhasePackJob* d_phase = nullptr;  // phase-kernel sides of the batch's plans: one more launch for all of them
  int nphase = 0, phase_blocks = 0;
};
namespace {
// groups of one fragment table: fragments keyed by (co0, ci0) in order of first appearance
void add_groups(std::vector<PackGroup>& groups, std::vector<int2>& gtaps, const Tables& T, const float* w, const mi_conv_plan* P) {
  std::unordered_map<int64_t, int> index;  // (co0, ci0) -> group
  const size_t g0 = groups.size();
  std::vector<std::vector<int2>> taps;
  for (int f = 0; f < T.nfrags; ++f) {
    const int* it = &T.frag_items[(size_t)f * 4];
    const int64_t key = ((int64_t)it[1] << 32) | (unsigned)it[2];
    auto found = index.find(key);
    int gi = found == index.end() ? -1 : found->second;
    if (gi < 0) {
      gi = (int)taps.size();
      index.emplace(key, gi);
      taps.emplace_back();
      PackGroup g;
      memset(&g, 0, sizeof(g));
      g.w = w; g.out = T.d_wpk;
      g.tr = it[3] & 1; g.perm = (it[3] >> 1) & 1;
      g.o0 = g.tr ? it[2] : it[1];   // torch out channel: kernel cin when transposed
      g.i0 = g.tr ? it[1] : it[2];
      g.Co_t = P->Cout; g.Ci_t = P->Cin; g.KT = P->KT;
      groups.push_back(g);
    }
    taps[(size_t)gi].push_back(int2{it[0], f});
  }
  for (size_t k = 0; k < taps.size(); ++k) {
    groups[g0 + k].ntaps = (int)taps[k].size();
    groups[g0 + k].tap_off = (int)gtaps.size();
    gtaps.insert(gtaps.end(), taps[k].begin(), taps[k].end());
  }
}
}  // namespace
int mi_conv_pack_batch_create(mi_pack_batch** out, mi_conv_plan* const* plans, const float* const* weights, int n) {
  if (!out || !plans || !weights || n <= 0) return MI_ERR_BAD_ARG;
  std::vector<PackGroup> groups;
  std::vector<int2> gtaps;
  size_t lds = 0;
  std::vector<PhasePackJob> phase;
  int phase_blocks = 0;
  std::vector<G11Job> g11;
  int g11_nblocks = 0;
  std::vector<PackSuper> supers;
  size_t lds_super = 0;
  static const int use_super = env_int("MI_PACK_SUPER", 1);  // 0: one block per group, every weight read once per table (A/B runs)
  bool super_ok = use_super != 0;
  for (int i = 0; i < n; ++i) {
    mi_conv_plan* P = plans[i];
    if (!P || !weights[i]) return MI_ERR_BAD_ARG;
    for (PhaseSide* ps : {&P->ph_fwd, &P->ph_dg})
      if (ps->mode) {
        phase.push_back(phase_job(*ps, weights[i], P->Cout, P->Cin, phase_blocks));
        phase_blocks += mi_pack_phase_blocks(ps->nfrags, ps->ncb, ps->nchunks);
      }
    if (P->g11) {
      g11.push_back(g11_job(P, weights[i], g11_nblocks));
      g11_nblocks += g11_blocks(P);
    }
    if (P->up) P = P->up_inner;  // (its tables pack like any k3 s1 conv's)
    P->c1_packed = true;
    const size_t g0 = groups.size();
    add_groups(groups, gtaps, P->fwd, weights[i], P);
    add_groups(groups, gtaps, P->dg, weights[i], P);
    if (P->d_c1w && groups.size() > g0) { groups[g0].c1w = P->d_c1w; groups[g0].c1n = 27 * P->Cin * P->Cout; }
    {  // super-groups of this plan: groups keyed by the 32 x 32 block of the torch weight they read
      std::unordered_map<int64_t, int> at;
      for (size_t gi = g0; gi < groups.size(); ++gi) {
        const PackGroup& g = groups[gi];
        const int O0 = g.o0 & ~31, I0 = g.i0 & ~31;
        const int64_t key = ((int64_t)O0 << 32) | (unsigned)I0;
        auto f = at.find(key);
        if (f == at.end()) {
          PackSuper su;
          memset(&su, 0, sizeof(su));
          su.w = weights[i]; su.O0 = O0; su.I0 = I0; su.Co_t = P->Cout; su.Ci_t = P->Cin; su.KT = P->KT;
          at.emplace(key, (int)supers.size());
          supers.push_back(su);
          f = at.find(key);
        }
        PackSuper& su = supers[(size_t)f->second];
        if (su.ng < 4) su.g[su.ng++] = (int)gi;
        else super_ok = false;  // (more than four groups in one block: not a layout this file produces)
      }
      const size_t need_su = sizeof(unsigned short) * 32 * (size_t)(32 * P->KT + 4);  // (bf16 staging: k_pack_super)
      if (need_su > lds_super) lds_super = need_su;
    }
    const size_t need = sizeof(float) * 32 * (size_t)(16 * P->KT + 1) > sizeof(float) * 16 * (size_t)(32 * P->KT + 1)
                            ? sizeof(float) * 32 * (size_t)(16 * P->KT + 1) : sizeof(float) * 16 * (size_t)(32 * P->KT + 1);
    if (need > lds) lds = need;
  }
  if (lds > 64 * 1024) return MI_ERR_UNSUPPORTED;
  mi_pack_batch* B = new mi_pack_batch();
  B->ngroups = (int)groups.size(); B->lds = lds;
  B->nphase = (int)phase.size(); B->phase_blocks = phase_blocks;
  B->ng11 = (int)g11.size(); B->g11_blocks = g11_nblocks;
  if (use_super > 1) fprintf(stderr, "[pack batch] %d plans, %zu groups, %zu super-groups, paired: %d, LDS %zu B\n", n, groups.size(), supers.size(), (int)super_ok, lds_super);
  if (super_ok && !supers.empty() && lds_super <= 150 * 1024) {
    if (hipMalloc((void**)&B->d_supers, sizeof(PackSuper) * supers.size()) != hipSuccess ||
        hipMemcpy(B->d_supers, supers.data(), sizeof(PackSuper) * supers.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_pack_super, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      mi_conv_pack_batch_destroy(B);
      return (int)hipErrorOutOfMemory;
    }
    B->nsupers = (int)supers.size(); B->lds_super = lds_super;
  }
  if (B->ng11) {
    if (hipMalloc((void**)&B->d_g11, sizeof(G11Job) * g11.size()) != hipSuccess ||
        hipMemcpy(B->d_g11, g11.data(), sizeof(G11Job) * g11.size(), hipMemcpyHostToDevice) != hipSuccess) {
      mi_conv_pack_batch_destroy(B);
      return (int)hipErrorOutOfMemory;
    }
  }
  if (B->nphase) {
    if (hipMalloc((void**)&B->d_phase, sizeof(PhasePackJob) * phase.size()) != hipSuccess ||
        hipMemcpy(B->d_phase, phase.data(), sizeof(PhasePackJob) * phase.size(), hipMemcpyHostToDevice) != hipSuccess) {
      mi_conv_pack_batch_destroy(B);
      return (int)hipErrorOutOfMemory;
    }
  }
  if (hipMalloc((void**)&B->d_groups, sizeof(PackGroup) * (groups.size() ? groups.size() : 1)) != hipSuccess ||
      hipMalloc((void**)&B->d_gtaps, sizeof(int2) * (gtaps.size() ? gtaps.size() : 1)) != hipSuccess) {
    mi_conv_pack_batch_destroy(B);
    return (int)hipErrorOutOfMemory;
  }
  hipError_t e = hipMemcpy(B->d_groups, groups.data(), sizeof(PackGroup) * groups.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(B->d_gtaps, gtaps.data(), sizeof(int2) * gtaps.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) { mi_conv_pack_batch_destroy(B); return (int)e; }
  *out = B;
  return 0;
}
int mi_conv_pack_batch_run(mi_pack_batch* B, hipStream_t st) {
  if (!B) return MI_ERR_BAD_ARG;
  if (B->nsupers) hipLaunchKernelGGL(k_pack_super, dim3(B->nsupers), dim3(kPackSuperT), B->lds_super, st, B->d_supers, B->d_groups, B->d_gtaps);
  else if (B->ngroups) hipLaunchKernelGGL(k_pack_groups, dim3(B->ngroups), dim3(256), B->lds, st, B->d_groups, B->d_gtaps);
  MI_CHECK_LAUNCH();
  if (B->ng11) hipLaunchKernelGGL(k_pack_gemm11, dim3(B->g11_blocks), dim3(256), 0, st, (const G11Job*)B->d_g11, B->ng11);
  MI_CHECK_LAUNCH();
  if (B->nphase) return mi_launch_pack_phase(B->d_phase, B->nphase, B->phase_blocks, st);
  return 0;
}
int mi_conv_pack_batch_destroy(mi_pack_batch* B) {
  if (!B) return 0;
  if (B->d_groups) (void)hipFree(B->d_groups);
  if (B->d_gtaps) (void)hipFree(B->d_gtaps);
  if (B->d_phase) (void)hipFree(B->d_phase);
  if (B->d_g11) (void)hipFree(B->d_g11);
  if (B->d_supers) (void)hipFree(B->d_supers);
  delete B;
  return 0;
}

int mi_conv_fwd_stats_chunks(const mi_conv_plan* P) {
  if (P && !P->up && P->c1_in) return mi_c1_expand_stats_chunks(P->N, P->Di, P->Hi, P->Wi, P->Cout);  // the network's input conv (conv_c1.hip)
  if (!P || P->up || !P->v27_fwd || P->N > 16 || P->ncb_fwd != 1) return 0;
  return 4 * mi_conv27_grid_x(P->N * P->g_fwd.tilesD * P->g_fwd.tilesH * P->g_fwd.tilesW, P->fwd.ny);
}

int mi_conv_fwd(mi_conv_plan* P, const void* x, int x_cs, const float* scale_shift, int silu, const float* addvec, int addvec_stride,
                const void* res, int res_cs, void* y, int y_cs, float* out_stats, hipStream_t st) {
  if (!P || !x || !y || x_cs < P->Cin || y_cs < P->Cout) return MI_ERR_BAD_ARG;
  if (out_stats && (!mi_conv_fwd_stats_chunks(P) || scale_shift || ((x_cs & 7) && !P->c1_in))) return MI_ERR_UNSUPPORTED;  // conv27 / 1 -> C paths only
  if (P->ph_fwd.mode && !scale_shift && !res) {  // Upsample + conv (scatter) / k3 s2 conv (gather) on the phase kernels
    const int e = phase_side_run(P->ph_fwd, x, x_cs, P->Di, P->Hi, P->Wi, y, y_cs, P->Do, P->Ho, P->Wo, P->N, addvec, addvec_stride, st);
    if (e != MI_ERR_UNSUPPORTED) return e;
  }
  if (P->up) {  // fallback: materialise the nearest-upsampled tensor, plain k3 s1 conv on the fine grid
    if (x_cs != P->Cin) return MI_ERR_UNSUPPORTED;
    int e = mi_upsample_nearest_fwd(x, P->d_xup, P->N, P->Di, P->Hi, P->Wi, P->Cin, 2, 2, 2, st);
    if (e) return e;
    return mi_conv_fwd(P->up_inner, P->d_xup, P->Cin, scale_shift, silu, addvec, addvec_stride, res, res_cs, y, y_cs, nullptr, st);
  }
  if ((P->c1_in || P->c1_out) && P->c1_packed && !scale_shift && !res) {
    if (out_stats && !P->c1_in) return MI_ERR_UNSUPPORTED;
    int e = P->c1_in ? mi_launch_c1_expand(x, x_cs, P->d_c1w, addvec, addvec_stride, y, y_cs, P->N, P->Di, P->Hi, P->Wi, P->Cout, 0, st, out_stats)
                     : mi_launch_c1_reduce(x, x_cs, P->d_c1w, addvec, addvec_stride, y, y_cs, P->N, P->Di, P->Hi, P->Wi, P->Cin, st);
    if (e != MI_ERR_UNSUPPORTED) return e;
  }
  if (out_stats && P->c1_in) return MI_ERR_UNSUPPORTED;  // (the table-driven fallback of a single-channel conv emits no sums: never silently)
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  const bf16* src = (const bf16*)x;
  int src_cs = x_cs;
  if (P->strided) {
    int e = mi_space_to_depth(x, x_cs, P->d_xs, P->N, P->Di, P->Hi, P->Wi, P->Cin, P->f[0], P->f[1], P->f[2], st);
    if (e) return e;
    src = P->d_xs;
    src_cs = P->Q * P->Cin;
  }
  a.x = src; a.x_cs = src_cs; a.N = P->N; a.Di = P->Dp; a.Hi = P->Hp; a.Wi = P->Wp; a.Cin = P->strided ? P->Q * P->Cin : P->Cin;
  if (!P->strided) { a.Di = P->Di; a.Hi = P->Hi; a.Wi = P->Wi; }
  a.y = (bf16*)y; a.y_cs = y_cs; a.Cout = P->Cout; a.Do = P->Do; a.Ho = P->Ho; a.Wo = P->Wo;
  a.ogpq = P->fwd.ny; a.outc_q = P->Cout;
  a.wpk = P->fwd.d_wpk; a.hdr = P->fwd.d_hdr; a.taps = P->fwd.d_taps; a.nchunks = P->fwd.nchunks;
  a.ss = scale_shift; a.ss_C = P->Cin; a.pro_silu = silu;
  a.addvec = addvec; a.addvec_stride = addvec_stride;
  a.res = (const bf16*)res; a.res_cs = res_cs;
  a.g = P->g_fwd;
  int ntiles = P->N * a.g.tilesD * a.g.tilesH * a.g.tilesW;
  a.wpk_bytes = (unsigned)P->fwd.nfrags * 1024u;
  {
    int64_t xb = (int64_t)a.N * a.Di * a.Hi * a.Wi * a.x_cs * 2;
    if (xb >= (1ll << 32)) return MI_ERR_UNSUPPORTED;
    a.x_bytes = (unsigned)xb;
  }
  a.perm16 = P->v27_fwd || P->v11_fwd;
  if (P->g11 && !P->v11_fwd && !scale_shift && !res && (x_cs & 7) == 0 && (y_cs & 7) == 0 && addvec_stride == 0 && !out_stats &&
      ((uintptr_t)x & 15) == 0 && (int64_t)P->N * P->Do * P->Ho * P->Wo < (1ll << 31))  // y[vox][co] = x[vox][:] . W[co][:] + bias
    return mi_gemm_nt_bf16(x, x_cs, 0, 0, P->d_w11, P->Cin, 0, 0, y, y_cs, 0, 0, addvec, nullptr, 0, 0, 0, P->N * P->Do * P->Ho * P->Wo, P->Cout, P->Cin, 1, 1,
                           1.0f, 0, 0, st);
  if (P->v11_fwd && !scale_shift && !res && (x_cs & 7) == 0) {
    const int64_t yb = (int64_t)P->N * P->Do * P->Ho * P->Wo * y_cs * 2;
    a.y_bytes = yb < (1ll << 32) ? (unsigned)yb : 0u;
    return mi_launch_conv1x1(a, P->ncb_fwd, P->fwd.ny, st);
  }
  if (P->v27_fwd && !scale_shift) {  // (the fused GroupNorm prologue lives in the register-staged kernel)
    if (x_cs & 7) return MI_ERR_UNSUPPORTED;
    const int64_t rb = res ? (int64_t)P->N * P->Do * P->Ho * P->Wo * res_cs * 2 : 0;
    a.res_bytes = rb < (1ll << 32) ? (unsigned)rb : 0u;
    const int64_t yb = (int64_t)P->N * P->Do * P->Ho * P->Wo * y_cs * 2;
    a.y_bytes = yb < (1ll << 32) ? (unsigned)yb : 0u;
    a.stats = out_stats; a.stats_chunks = out_stats ? mi_conv_fwd_stats_chunks(P) : 0;
    const int e27 = mi_launch_conv27(a, P->ncb_fwd, 0, ntiles, P->fwd.ny, st);
    if (e27 != MI_ERR_UNSUPPORTED || out_stats) return e27;  // (odd output pitch / >= 4 GiB output: the table-driven kernel below)
    a.stats = nullptr;
  }
  return launch_igemm_any(a, P->ncb_fwd, P->full27 ? 1 : 0, ntiles, P->fwd.ny, st);
}

int mi_conv_dgrad(mi_conv_plan* P, const void* dy, int dy_cs, void* dx, int dx_cs, hipStream_t st) {
  if (!P || !dy || !dx || dy_cs < P->Cout || dx_cs < P->Cin) return MI_ERR_BAD_ARG;
  if (P->ph_dg.mode) {  // data gradient of Upsample + conv (gather from the fine dy) / of the k3 s2 conv (scatter into the fine dx)
    const int e = phase_side_run(P->ph_dg, dy, dy_cs, P->Do, P->Ho, P->Wo, dx, dx_cs, P->Di, P->Hi, P->Wi, P->N, nullptr, 0, st);
    if (e != MI_ERR_UNSUPPORTED) return e;
  }
  if (P->up) {  // fallback: data gradient on the fine grid, then fold the 2x2x2 blocks (nearest-upsample backward)
    if (dx_cs != P->Cin) return MI_ERR_UNSUPPORTED;
    int e = mi_conv_dgrad(P->up_inner, dy, dy_cs, P->d_xup, P->Cin, st);
    if (e) return e;
    return mi_upsample_nearest_bwd(P->d_xup, dx, P->N, P->Di, P->Hi, P->Wi, P->Cin, 2, 2, 2, st);
  }
  if (P->c1_out && P->c1_packed) {  // dx[v][ci] = sum_t dy[v - t] W[0][ci][t]: the 1 -> C kernel with the taps flipped
    int e = mi_launch_c1_expand(dy, dy_cs, P->d_c1w, nullptr, 0, dx, dx_cs, P->N, P->Di, P->Hi, P->Wi, P->Cin, 1, st);
    if (e != MI_ERR_UNSUPPORTED) return e;
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16*)dy; a.x_cs = dy_cs; a.N = P->N; a.Di = P->Do; a.Hi = P->Ho; a.Wi = P->Wo; a.Cin = P->Cout;
  a.Do = P->Dp; a.Ho = P->Hp; a.Wo = P->Wp;
  a.Cout = P->Q * P->Cin;
  a.ogpq = P->dg.ny / P->Q; a.outc_q = P->Cin;
  if (P->strided) {
    if (dx_cs != P->Cin) return MI_ERR_UNSUPPORTED;
    a.y = P->d_dxs; a.y_cs = P->Q * P->Cin;
  } else {
    a.y = (bf16*)dx; a.y_cs = dx_cs;
  }
  a.wpk = P->dg.d_wpk; a.hdr = P->dg.d_hdr; a.taps = P->dg.d_taps; a.nchunks = P->dg.nchunks;
  a.g = P->g_dg;
  int ntiles = P->N * a.g.tilesD * a.g.tilesH * a.g.tilesW;
  a.wpk_bytes = (unsigned)P->dg.nfrags * 1024u;
  {
    int64_t xb = (int64_t)a.N * a.Di * a.Hi * a.Wi * a.x_cs * 2;
    if (xb >= (1ll << 32)) return MI_ERR_UNSUPPORTED;
    a.x_bytes = (unsigned)xb;
  }
  a.perm16 = P->v27_dg || P->v11_dg;
  if (P->g11 && !P->v11_dg && (dy_cs & 7) == 0 && (dx_cs & 7) == 0 && ((uintptr_t)dy & 15) == 0 &&
      (int64_t)P->N * P->Do * P->Ho * P->Wo < (1ll << 31))  // dx[vox][ci] = dy[vox][:] . W^T[ci][:]
    return mi_gemm_nt_bf16(dy, dy_cs, 0, 0, P->d_w11t, P->Cout, 0, 0, dx, dx_cs, 0, 0, nullptr, nullptr, 0, 0, 0, P->N * P->Do * P->Ho * P->Wo, P->Cin, P->Cout, 1,
                           1, 1.0f, 0, 0, st);
  if (P->v11_dg && (dy_cs & 7) == 0) {
    const int64_t yb = (int64_t)P->N * a.Do * a.Ho * a.Wo * a.y_cs * 2;
    a.y_bytes = yb < (1ll << 32) ? (unsigned)yb : 0u;
    return mi_launch_conv1x1(a, P->ncb_dg, P->dg.ny, st);
  }
  if (P->v27_dg) {
    if (dy_cs & 7) return MI_ERR_UNSUPPORTED;
    const int64_t yb = (int64_t)P->N * a.Do * a.Ho * a.Wo * a.y_cs * 2;
    a.y_bytes = yb < (1ll << 32) ? (unsigned)yb : 0u;
    const int e27 = mi_launch_conv27(a, P->ncb_dg, 1, ntiles, P->dg.ny, st);
    if (e27 != MI_ERR_UNSUPPORTED) return e27;
  }
  int e = launch_igemm_any(a, P->ncb_dg, P->full27 ? 2 : 0, ntiles, P->dg.ny, st);
  if (e) return e;
  if (P->strided) return mi_depth_to_space(P->d_dxs, dx, P->N, P->Di, P->Hi, P->Wi, P->Cin, P->f[0], P->f[1], P->f[2], st);
  return 0;
}

// dw (fp32, torch layout [Cout][Cin][kd][kh][kw]) += wgrad;  x side takes the same fused prologue as the forward
int mi_conv_wgrad(mi_conv_plan* P, const void* x, int x_cs, const float* scale_shift, int silu, const void* dy, int dy_cs, float* dw,
                  float* dy_colsum, int dy_colsum_stride, hipStream_t st) {
  if (!P || !x || !dy || !dw || x_cs < P->Cin || dy_cs < P->Cout) return MI_ERR_BAD_ARG;
  if (P->up && P->up_wg && !scale_shift && (x_cs & 7) == 0 && (dy_cs & 7) == 0) {
    const int e = upconv_wgrad(P, x, x_cs, dy, dy_cs, dw, dy_colsum, dy_colsum_stride, st);
    if (e != MI_ERR_UNSUPPORTED) return e;
  }
  if (P->up) {  // fallback: on the fine grid against the nearest-upsampled x
    if (x_cs != P->Cin || scale_shift) return MI_ERR_UNSUPPORTED;
    int e = mi_upsample_nearest_fwd(x, P->d_xup, P->N, P->Di, P->Hi, P->Wi, P->Cin, 2, 2, 2, st);
    if (e) return e;
    return mi_conv_wgrad(P->up_inner, P->d_xup, P->Cin, nullptr, 0, dy, dy_cs, dw, dy_colsum, dy_colsum_stride, st);
  }
  static const int use_w11 = env_int("MI_WGRAD1X1", 1);
  if (use_w11 && P->KT == 1 && !P->strided && !scale_shift && !(dy_colsum && dy_colsum_stride != 0 && dy_colsum_stride < P->Cout)) {
    int e = mi_launch_wgrad1x1(x, x_cs, P->Cin, dy, dy_cs, P->Cout, P->N, (int64_t)P->Di * P->Hi * P->Wi, dw, dy_colsum, dy_colsum_stride, st);
    if (e != MI_ERR_UNSUPPORTED) return e;
  }
  static const int use_c1w = env_int("MI_C1_WGRAD", 1);
  if (use_c1w && (P->c1_in || P->c1_out) && P->d_c1part && !scale_shift && !(dy_colsum && dy_colsum_stride != 0)) {
    // one channel on one side: a streaming kernel with the 27 taps as the GEMM's N axis (conv_c1.hip) instead of padding that side to 32
    int e = P->c1_in ? mi_launch_c1_wgrad(dy, dy_cs, x, x_cs, dw, dy_colsum, nullptr, P->d_c1part, P->N, P->Di, P->Hi, P->Wi, P->Cout, 0, st)
                     : mi_launch_c1_wgrad(x, x_cs, dy, dy_cs, dw, nullptr, dy_colsum, P->d_c1part, P->N, P->Di, P->Hi, P->Wi, P->Cin, 1, st);
    if (e != MI_ERR_UNSUPPORTED) return e;
  }
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.sx = w.sy = 1; w.cs_chunks = 1;
  ConvArgs& a = w.c;
  const bf16* src = (const bf16*)x;
  int src_cs = x_cs;
  static const int use_w2 = env_int("MI_WGRAD2", 1);
  // k3 s2 conv on all axes: the LDS-DMA kernel gathers each pair's class image from x in place (sx = 2) -- no space-to-depth copy
  bool direct = false;
  if (P->s2_wg_direct && use_w2 && !scale_shift && (x_cs & 7) == 0 && (dy_cs & 7) == 0 && (P->Cout & 7) == 0 && P->g_wg.vox == 64) {
    const int px = (P->g_wg.lds_bytes + 1023) / 1024, py = (P->g_wg.TD * P->g_wg.TH * P->g_wg.TW * 64 + 1023) / 1024;
    const int64_t dyb = (int64_t)P->N * P->Do * P->Ho * P->Wo * dy_cs * 2, xb = (int64_t)P->N * P->Di * P->Hi * P->Wi * x_cs * 2;
    direct = px <= 40 && py <= 16 && dyb < (1ll << 32) && xb < (1ll << 32);
  }
  if (P->strided && !direct) {
    int e = mi_space_to_depth(x, x_cs, P->d_xs, P->N, P->Di, P->Hi, P->Wi, P->Cin, P->f[0], P->f[1], P->f[2], st);
    if (e) return e;
    src = P->d_xs;
    src_cs = P->Q * P->Cin;
  }
  const bool s2d = P->strided && !direct;
  a.x = src; a.x_cs = src_cs; a.N = P->N;
  a.Di = s2d ? P->Dp : P->Di; a.Hi = s2d ? P->Hp : P->Hi; a.Wi = s2d ? P->Wp : P->Wi;
  a.Cin = s2d ? P->Q * P->Cin : P->Cin;
  a.Cout = P->Cout; a.Do = P->Do; a.Ho = P->Ho; a.Wo = P->Wo;
  a.hdr = direct ? P->d_hdr2 : P->wg.d_hdr; a.taps = P->wg.d_taps; a.nchunks = P->wg.nchunks;
  if (direct) w.sx = 2;
  a.ss = scale_shift; a.ss_C = P->Cin; a.pro_silu = silu;
  {
    int64_t xb = (int64_t)a.N * a.Di * a.Hi * a.Wi * a.x_cs * 2;
    if (xb >= (1ll << 32)) return MI_ERR_UNSUPPORTED;
    a.x_bytes = (unsigned)xb;
  }
  a.g = P->g_wg;
  w.dy = (const bf16*)dy; w.dy_cs = dy_cs;
  w.part = P->d_part; w.pair_off = P->d_pair_off; w.split_stride = P->wg_split_stride;
  w.ntiles = P->N * a.g.tilesD * a.g.tilesH * a.g.tilesW;
  w.nsplit = P->wg_nsplit;
  static const int contig_env = env_int("MI_WGRAD_CONTIG", 1);
  w.contig = contig_env && (w.ntiles % 8 == 0) && (w.nsplit % 8 == 0) && w.nsplit <= w.ntiles;
  static const int dbg = mi_diag_knob("MI_WGRAD_DBG");
  w.dbg = dbg;
  w.colsum = dy_colsum; w.colsum_stride = dy_colsum_stride;
  static const int cs_slab = env_int("MI_CS_SLAB", 1);
  bool cs_pairs_multi_tap = true;  // pairs that own the column sums (first chunk of a cout block) must be tap pairs, not k-split pairs
  for (int yy = 0; yy < P->wg.ny; ++yy) cs_pairs_multi_tap = cs_pairs_multi_tap && P->wg.hdr[((size_t)yy * P->wg.nchunks) * 4 + 1] > 1;
  w.cs_part = (cs_slab && dy_colsum && cs_pairs_multi_tap) ? P->d_cspart : nullptr;  if (dy_colsum && dy_colsum_stride != 0 && dy_colsum_stride < P->Cout) return MI_ERR_BAD_ARG;  // 0: one row for the whole batch
  const int hv = a.g.HD * a.g.HH * a.g.HW;
  const int np = (hv * 4 + 255) / 256;
  const int nvox = a.g.TD * a.g.TH * a.g.TW;
  const int npy = (nvox * 4 + 255) / 256;
  size_t lds = (size_t)a.g.lds_bytes + (size_t)nvox * a.g.vox;
  w.npairs = P->wg.ny * P->wg.nchunks;
  dim3 grid(w.nsplit >= 8 ? w.npairs * ((w.nsplit + 7) / 8) * 8 : w.npairs * w.nsplit), blk(512);
  if ((a.g.TD * (a.g.TH / 2) * (a.g.TW / 8)) % 2) return MI_ERR_BAD_ARG;  // the k loop is unrolled in pairs
  if (npy > 4 || np > 16) return MI_ERR_BAD_ARG;  // (256-thread counts; the kernel runs 512 threads: half of each)
  int max_taps = 0;
  for (size_t i = 1; i < P->wg.hdr.size(); i += 4) max_taps = P->wg.hdr[i] > max_taps ? P->wg.hdr[i] : max_taps;
  if (max_taps >= 32) return MI_ERR_UNSUPPORTED;  // wave 7 keeps a spare accumulator for the fused dY column sums
#define MI_LAUNCH_WG(NPV) MI_LAUNCH_WG_G(NPV, false)
#define MI_LAUNCH_WG_G(NPV, G3)                                                                                    \
  do {                                                                                                             \
    auto kern = k_conv_wgrad<NPV, 2, G3>;                                                                          \
    static int lds_ok = 0;                                                                                         \
    if ((int)lds > lds_ok) {                                                                                       \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return (int)e;                                                                          \
      lds_ok = 160 * 1024;                                                                                         \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, blk, lds, st, w);                                                               \
  } while (0)
  // LDS-DMA kernel (k_conv_wgrad2): whole 8-channel pieces only, no fused prologue
  {
    const int px = (a.g.lds_bytes + 1023) / 1024, py = (nvox * 64 + 1023) / 1024;
    const int64_t dyb = (int64_t)P->N * P->Do * P->Ho * P->Wo * dy_cs * 2;
    if (use_w2 && !scale_shift && a.g.vox == 64 && (a.x_cs & 7) == 0 && (a.Cin & 7) == 0 && (dy_cs & 7) == 0 && (P->Cout & 7) == 0 && px <= 40 &&
        py <= 16 && dyb < (1ll << 32)) {
      w.dy_bytes = (unsigned)dyb;
      w.nbuf = (P->KT == 1 && px == 16 && py == 16) ? 4 : 2;  // 1x1: tiles are pure loads -> three tiles in flight per workgroup
      static const int use_flags = env_int("MI_WGRAD_FLAGS", 1);
      w.flags = use_flags;
      const size_t lds2 = (size_t)(w.nbuf * (px + py)) * 1024 + 64;  // + the hand-off counters
      const bool geo3 = P->full27 && a.g.row == WG3_XROW && a.g.slice == WG3_XSLICE && a.g.TD == 4 && a.g.TH == 8 && a.g.TW == 8;
      static bool attr3 = false, attrg = false;
      bool& done = geo3 ? attr3 : attrg;
      if (!done) {
        hipError_t e = geo3 ? hipFuncSetAttribute((const void*)k_conv_wgrad2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                            : hipFuncSetAttribute((const void*)k_conv_wgrad2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
      }
      static const int use_roll = env_int("MI_WGRAD_ROLL", 1);  // 0: k_conv_wgrad2 for every layer (A/B runs)
      // (runs of more than two columns lose more to L2 locality -- the workgroups of an XCD are then columns apart -- than the saved
      // pieces return: 64->64 and 96->32 at 128^3 measured +1 / +2.5 %, 32->32 -6 %, 64->32 -3.5 %, the 64^3 .. 16^3 levels -1.5 .. -3 %)
      const bool roll = use_roll && geo3 && !direct && w.sx == 1 && w.sy == 1 && w.nbuf == 2 && w.flags && P->wg.hdr[1] == 27 && w.nsplit <= w.ntiles &&
                        a.g.tilesD >= 2 && (use_roll > 1 || (int64_t)w.ntiles <= 2ll * a.g.tilesD * w.nsplit);
      if (roll) {
        static bool attr_r = false;
        if (!attr_r) {
          hipError_t e = hipFuncSetAttribute((const void*)k_conv_wgrad3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          if (e != hipSuccess) return (int)e;
          attr_r = true;
        }
#ifdef MI_WG3_DIAG_CLK
        {
          unsigned long long h0[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, ~0ull, 0};
          (void)hipDeviceSynchronize();
          (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg3_clk), h0, sizeof(h0));
          static unsigned long long z[6][1024];
          (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg3_span), z, sizeof(z));
        }
#endif
        hipLaunchKernelGGL(k_conv_wgrad3, grid, dim3(768), (size_t)WR_LDS, st, w);
#ifdef MI_WG3_DIAG_CLK
        {
          unsigned long long h[16];
          (void)hipDeviceSynchronize();
          (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wg3_clk), sizeof(h));
          auto us = [&](int a, int b) { return (double)((long long)h[b] - (long long)h[a]) / 100.0; };
          fprintf(stderr, "[wgrad3 %d->%d ntiles %d nsplit %d] all workgroups: first entry -> last exit %.2f us | workgroup 0 (from the first entry): entry %.2f, "
                  "compute wave 0: setup %.2f, prologue barrier %.2f, tiles %.2f, final barrier %.2f, stores issued %.2f, stores landed %.2f | loader wave: "
                  "descriptors %.2f, prologue issue %.2f, landed %.2f us\n", P->Cin, P->Cout, w.ntiles, w.nsplit, us(14, 15), us(14, 0), us(0, 1), us(1, 2), us(2, 3), us(3, 4),
                  us(4, 5), us(5, 6), us(8, 9), us(9, 10), us(10, 11));
          static unsigned long long sp[6][1024];
          (void)hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_wg3_span), sizeof(sp));
          std::vector<double> ent, dur, ext, lp, pro, bar, iss;
          for (unsigned b = 0; b < grid.x && b < 1024; ++b)
            if (sp[1][b]) { bar.push_back((double)(sp[4][b] - h[14]) / 100.0); iss.push_back((double)(sp[5][b] - h[14]) / 100.0); lp.push_back((double)(sp[2][b] - h[14]) / 100.0); pro.push_back((double)(sp[3][b] - h[14]) / 100.0); ent.push_back((double)(sp[0][b] - h[14]) / 100.0); ext.push_back((double)(sp[1][b] - h[14]) / 100.0); dur.push_back((double)(sp[1][b] - sp[0][b]) / 100.0); }
          std::sort(ent.begin(), ent.end()); std::sort(dur.begin(), dur.end()); std::sort(ext.begin(), ext.end());
          std::sort(lp.begin(), lp.end()); std::sort(pro.begin(), pro.end()); std::sort(bar.begin(), bar.end()); std::sort(iss.begin(), iss.end());
          if (!ent.empty()) fprintf(stderr, "    final barrier passed min %.2f median %.2f max %.2f | wave 0 stores issued min %.2f median %.2f max %.2f us\n", bar.front(), bar[bar.size() / 2],
                                    bar.back(), iss.front(), iss[iss.size() / 2], iss.back());
          if (!ent.empty()) fprintf(stderr, "    prologue done min %.2f median %.2f max %.2f | tile loop done min %.2f median %.2f max %.2f us\n", pro.front(), pro[pro.size() / 2], pro.back(),
                                    lp.front(), lp[lp.size() / 2], lp.back());
          if (!ent.empty()) {
            auto qt = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
            fprintf(stderr, "    %zu workgroups: entry 10%% %.2f median %.2f 90%% %.2f max %.2f | exit min %.2f median %.2f max %.2f | lifetime min %.2f median %.2f 90%% %.2f max %.2f us\n",
                    ent.size(), qt(ent, 0.1), qt(ent, 0.5), qt(ent, 0.9), ent.back(), ext.front(), qt(ext, 0.5), ext.back(), dur.front(), qt(dur, 0.5), qt(dur, 0.9), dur.back());
          }
        }
#endif
      } else if (geo3) hipLaunchKernelGGL(k_conv_wgrad2<true>, grid, dim3(768), lds2, st, w);
      else hipLaunchKernelGGL(k_conv_wgrad2<false>, grid, dim3(768), lds2, st, w);
#ifdef MI_WG2_DIAG_BAR
      {
        unsigned long long h[8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wg2_clk), sizeof(h));
        fprintf(stderr, "[wgrad2] workgroup 0: compute wave 0: %llu cycles over %llu tiles (%llu taps), %llu inside the barrier | loader wave 8: %llu cycles = issue %llu + vmcnt wait %llu + barrier %llu\n",
                h[4], h[6], h[7], h[5], h[0], h[1], h[2], h[3]);
      }
#endif
      launch_wgrad_reduce(P->d_part, P->wg_split_stride, P->wg_nsplit, P->d_uitems, P->wg_nitems, dw, P->Cout, P->Cin, P->KT, P->d_pair_off,
                          P->wg.ny * P->wg.nchunks,
                          CsReduce{w.cs_part ? P->d_cspart : nullptr, P->wg_nsplit, P->N, P->Cout, dy_colsum, dy_colsum_stride, 0}, st);
      MI_CHECK_LAUNCH();
      return 0;
    }
  }
  const int np8 = (hv * 4 + 511) / 512;
  const bool geo3d = P->full27 && a.g.vox == 64 && a.g.row == WG3_XROW && a.g.slice == WG3_XSLICE && a.g.TD == 4 && a.g.TH == 8 &&
                     a.g.TW == 8 && np8 == 5;
  if (geo3d) MI_LAUNCH_WG_G(5, true);
  else if (np8 <= 2) MI_LAUNCH_WG(2);
  else if (np8 <= 3) MI_LAUNCH_WG(3);
  else if (np8 <= 5) MI_LAUNCH_WG(5);
  else MI_LAUNCH_WG(8);
#undef MI_LAUNCH_WG
#undef MI_LAUNCH_WG_G
  launch_wgrad_reduce(P->d_part, P->wg_split_stride, P->wg_nsplit, P->d_uitems, P->wg_nitems, dw, P->Cout, P->Cin, P->KT, P->d_pair_off,
                      P->wg.ny * P->wg.nchunks, CsReduce{w.cs_part ? P->d_cspart : nullptr, P->wg_nsplit, P->N, P->Cout, dy_colsum, dy_colsum_stride, 0},
                      st);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
