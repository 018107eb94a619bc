// Structures shared by the convolution translation units (conv.hip: plans, table-driven kernel, weight gradient;
// conv27.hip: the LDS-DMA 3x3x3 stride-1 kernel).  Plain data, passed to kernels by value.
#pragma once
#include "common.h"

namespace mi_conv {

constexpr int VOXB = 80;    // LDS bytes per voxel (32 ch * 2 B + 16 B pad)
constexpr int KC = 32;      // channels per chunk

struct Geom {           // tile + LDS geometry (host-computed, passed by value)
  int TD, TH, TW;       // output tile (voxels); TD * (TH/4) * (TW/8) == 4 * VB
  int hd, hh, hw;       // halo on each side per axis (0/1)
  int HD, HH, HW;       // LDS tile dims = T + 2*halo
  int vox, row, slice;  // byte pitches (voxel: 80 for the b128 fragment reads of fwd/dgrad, 64 for wgrad's transposed reads)
  int lds_bytes;
  int tilesD, tilesH, tilesW;
  int hb;               // tile walk: rows of tiles per h-block (tile_origin); 1 = plain (w, h, d) raster
};

struct ConvArgs {
  const bf16* x; int x_cs;
  int N, Di, Hi, Wi, Cin;       // tensor the loader reads (already space-to-depth'ed when strided); Cin = its channel count
  bf16* y; int y_cs; int Cout;  // channels the kernel produces (store masked to Cout)
  int ogpq, outc_q;             // output groups (blockIdx.y) per parity class, channels per class (== ny, Cout when no classes)
  int Do, Ho, Wo;
  const u32x4* wpk;             // packed A fragments (64 x 16 B each)
  const int* hdr;               // [ny][nchunks][4] = tap_begin, ntaps, src_c0, wfrag_begin
  const int* taps;              // LDS byte offsets
  int nchunks;
  const float* ss; int ss_C; int pro_silu;  // prologue affine [N][ss_C][2] (channel = src channel % ss_C) or null
  const float* addvec; int addvec_stride;   // fp32 [Cout] (stride 0) or [N] rows of pitch `stride`; null = none
  const bf16* res; int res_cs;
  unsigned y_bytes;             // size of the output tensor for its buffer descriptor (0: >= 4 GiB)
  unsigned res_bytes;           // size of the residual tensor for its buffer descriptor (0: >= 4 GiB, element-wise path)
  int ntiles;
  int perm16;  // weights packed with the conv27 row permutation (lane's 16 accumulator registers = 16 consecutive channels)
  int dbg;  // ablation knob (MI_IGEMM_DBG): 1 = stage only the first image, 2 = skip the MFMA loop, 3 = skip the epilogue
  unsigned x_bytes, wpk_bytes;  // sizes for the buffer descriptors (both < 4 GiB, checked on the host)
  // conv27 forward only: per-channel (sum, sum of squares) of the bf16 OUTPUT for the GroupNorm that consumes it, laid out as the
  // GroupNorm partials stats[n][c][chunk][2]; chunk = blockIdx.x * 4 + store wave, stats_chunks = 4 * gridDim.x.  null: none.
  float* stats; int stats_chunks;
  Geom g;
};


// XCD-aware tile walk of the persistent conv kernels: workgroups that share `blockIdx.x % 8` are observed to share an XCD
// (speed only), so each of the 8 residue classes owns a CONTIGUOUS tile range -- neighbouring tiles, whose halos overlap,
// are then served by the same L2.
static __device__ __forceinline__ int first_tile(int ntiles, int& last, int& step) {
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  last = (x + 1) * tpx < ntiles ? (x + 1) * tpx : ntiles;
  step = nslots;
  return x * tpx + slot;
}

// Tile index -> tile origin.  Within an image the walk is blocked: w fastest, then the g.hb rows of an h-block, then ALL depths, then
// the next h-block.  The 32 workgroups of an XCD work on 32 consecutive tiles (first_tile), so with 16 tiles per row and hb = 2 one
// "depth step" of an XCD is 32 tiles = 2.3 MB of halo reads + stores: the d-halo slices of depth td (a third of every halo image)
// were read one step earlier and are still in that XCD's 4 MiB L2.  The plain (w, h, d) raster returns to a tile's d-neighbour a
// whole depth layer (256 tiles, 18 MB at 128^3) later: those reads came from beyond L2.  Images stay outermost (the GroupNorm-sum /
// bias-gradient epilogues rely on the image index never decreasing along a workgroup's walk).
static __device__ __forceinline__ void tile_origin(const Geom& g, int tile, int& n, int& d0, int& h0, int& w0) {
  int tw = tile % g.tilesW; tile /= g.tilesW;
  int th, td;
  if (g.hb > 1) {
    const int hr = tile % g.hb; tile /= g.hb;
    td = tile % g.tilesD; tile /= g.tilesD;
    const int nhb = g.tilesH / g.hb;
    th = (tile % nhb) * g.hb + hr;
    n = tile / nhb;
  } else {
    th = tile % g.tilesH; tile /= g.tilesH;
    td = tile % g.tilesD;
    n = tile / g.tilesD;
  }
  d0 = td * g.TD; h0 = th * g.TH; w0 = tw * g.TW;
}

// The same walk, stepped without divisions.  A persistent workgroup visits tile0, tile0 + step, ...; tile_origin costs five integer
// divisions by run-time values = ~150 VALU instructions even on wave-uniform operands, and in the role kernels every wave pays them
// per tile out of the few vector-issue slots the MFMA stream leaves (measured in k_conv27: ~900 cycles per tile on every wave).
// TileWalk keeps the mixed-radix digits of the tile index (w, row inside the h-block, depth, h-block, image) and adds the digits
// of the step with carries: scalar adds and compares only.
struct TileWalk {
  int tw, hr, td, hbi, n;       // digits of the current tile
  int s_tw, s_hr, s_td, s_hbi, s_n;  // digits of the step
  int r_hr, r_hbi;              // radices that are not in Geom: rows per h-block, h-blocks
};
static __device__ __forceinline__ void walk_digits(const Geom& g, int hb, int nhb, int t, int& tw, int& hr, int& td, int& hbi, int& n) {
  tw = t % g.tilesW; t /= g.tilesW;
  hr = t % hb; t /= hb;
  td = t % g.tilesD; t /= g.tilesD;
  hbi = t % nhb;
  n = t / nhb;
}
static __device__ __forceinline__ void walk_init(TileWalk& k, const Geom& g, int tile0, int step) {
  k.r_hr = g.hb > 1 ? g.hb : g.tilesH;  // hb <= 1: one h-block of all rows = the plain (w, h, d) raster
  k.r_hbi = g.tilesH / k.r_hr;
  walk_digits(g, k.r_hr, k.r_hbi, tile0, k.tw, k.hr, k.td, k.hbi, k.n);
  walk_digits(g, k.r_hr, k.r_hbi, step, k.s_tw, k.s_hr, k.s_td, k.s_hbi, k.s_n);
}
static __device__ __forceinline__ void walk_step(TileWalk& k, const Geom& g) {
  int c;
  k.tw += k.s_tw;          c = k.tw >= g.tilesW;  k.tw -= c ? g.tilesW : 0;
  k.hr += k.s_hr + c;      c = k.hr >= k.r_hr;    k.hr -= c ? k.r_hr : 0;
  k.td += k.s_td + c;      c = k.td >= g.tilesD;  k.td -= c ? g.tilesD : 0;
  k.hbi += k.s_hbi + c;    c = k.hbi >= k.r_hbi;  k.hbi -= c ? k.r_hbi : 0;
  k.n += k.s_n + c;
}
static __device__ __forceinline__ void walk_origin(const TileWalk& k, const Geom& g, int& n, int& d0, int& h0, int& w0) {
  n = k.n; d0 = k.td * g.TD; h0 = (k.hbi * k.r_hr + k.hr) * g.TH; w0 = k.tw * g.TW;
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

}  // namespace mi_conv
using namespace mi_conv;

// conv27.hip: k3 s1 p1 3-D forward (flip = 0) / data gradient (flip = 1) on the 4x8x8 tile; a.g / tables as for the table-driven kernel
int mi_launch_conv27(const ConvArgs& a, int NCB, int flip, int ntiles, int ny, hipStream_t st);
int mi_conv27_grid_x(int ntiles, int ny);  // gridDim.x of that launch (4 statistics chunks per workgroup)
// convph.hip: factor-2 phase convolutions (mode 1 scatter: tile grid = coarse input, fine output; mode 2 gather: fine input, tile grid =
// coarse output) and their weight pack (fragment = fp32 sum of the master weights over a tap set, d_masks[phase * 8 + tap])
int mi_launch_convph(const ConvArgs& a, int NCB, int mode, int ntiles, int ny, hipStream_t st);
struct PhasePackJob {          // one phase side's weight pack (k_pack_phase); block0 = first blockIdx.x of the job in a batched launch
  const float* w; void* out; const unsigned* masks;
  int NCB, nchunks, Ko, Ki, Co_t, Ci_t, tr, block0;
};
int mi_pack_phase_blocks(int nfrags, int NCB, int nchunks);  // blocks (along x) of a job: (kernel-out blocks of 32) x (kernel-in blocks of 16)
int mi_launch_pack_phase(const PhasePackJob* d_jobs, int njobs, int total_blocks, hipStream_t st);
// conv1x1.hip: 1x1x1 forward / data gradient as a streaming GEMM over voxels
int mi_launch_conv1x1(const ConvArgs& a, int NCB, int ny, hipStream_t st);
int mi_launch_wgrad1x1(const void* x, int x_cs, int Cin, const void* dy, int dy_cs, int Cout, int N, int64_t V, float* dw, float* colsum,
                       int colsum_stride, hipStream_t st);
// conv_c1.hip: k3 s1 p1 3-D convs with one channel on one side (fp32 master weights [C][27] read directly)
int mi_launch_c1_expand(const void* s, int s_cs, const float* w, const float* addvec, int av_stride, void* y, int y_cs, int N, int D, int H,
                        int W, int C, int flip, hipStream_t st, float* stats = nullptr);  // stats: GroupNorm sums of the output (forward, C <= 32)
int mi_c1_expand_stats_chunks(int N, int D, int H, int W, int C);
int mi_launch_c1_wgrad(const void* m, int m_cs, const void* s1, int s_cs, float* dw, float* colsum_m, float* colsum_s, float* part, int N, int D,
                       int H, int W, int C, int flip, hipStream_t st);
int64_t mi_c1_wgrad_scratch_floats(int C);
int mi_launch_c1_reduce(const void* x, int x_cs, const float* w, const float* addvec, int av_stride, void* y, int y_cs, int N, int D, int H,
                        int W, int C, hipStream_t st);
