// 3x3x3 stride-1 convolution (forward and data gradient) for gfx950: one persistent 8-wave workgroup per CU with fixed
// wave roles; both MFMA operands are served from LDS that is filled by LDS-DMA (buffer_load ... lds).
// (Same call sites as conv.hip: every k3 s1 p1 `Convolution` of UNet:630-672,557-565,1935 / AEKL:158-187.)
//
// Why a second kernel.  In the table-driven kernel (conv.hip) every wave fetches its own copy of every weight fragment
// from L2 (0.5 KiB per MFMA: at full MFMA rate that alone is the CU's whole 64 B/clk vector-memory path), the halo image
// goes through 40 VGPRs + ds_write, and the epilogue stores 8 bytes per lane at a voxel stride (16 valid bytes per
// 128-byte line per instruction; with bias rows + residual it costs as much as the MFMAs).  Measured on MI355X, a
// 4-wave version of this design whose waves issued their own DMA and ran their own epilogue spent 20 % of its time
// issuing LDS-DMA and 13 % in the epilogue while the bare fragment-read + MFMA stream ran at 1.3-1.45 PFLOP/s: hence roles.
//
//   waves 0-3  COMPUTE: one per SIMD; wave w owns the d-slice w of the 4x8x8-voxel tile (two 4x8 voxel blocks) and NCB
//              32-channel output blocks.  Their instruction stream is: counted lgkmcnt, MFMA, one ds_read_b128 of the NEXT
//              tap's fragments -- a steady LDS trickle (1 KiB per MFMA slot per wave) instead of bursts.  At tile end the
//              accumulators go to a bf16 staging tile in LDS and are re-initialised with bias (+ time embedding).
//   waves 4-5  WEIGHT LOADERS: packed A fragments are 1 KiB and lane-linear = exactly one LDS-DMA instruction; a ring of
//              3 tap groups is shared by the compute waves, group J+2 is requested while group J is consumed and checked
//              (vmcnt(0) + s_barrier) one group before its first use, so fragment prefetch runs across group boundaries.
//              Single-chunk 32-channel layers keep all 27 taps resident: nothing is re-fetched after the first tile.
//   waves 6-7  HALO LOADERS + STORE EPILOGUE: the halo image 6x10x10 voxels x 32 channels is dense (64-byte voxels,
//              38.4 KB), double buffered, filled by LDS-DMA with the hardware range check supplying the zero padding; bank
//              conflicts of the 4x8-voxel fragment reads are removed by XOR-ing the 16-byte slot with the halo row
//              (slot ^= hh & 3) on the DMA's per-lane SOURCE address (the LDS destination of a DMA is lane-linear).
//              They also turn the previous tile's staging tile into full-line 16-byte stores (+ residual), a few pieces
//              per tap group, so the compute waves never wait for global memory.
// Every wave executes the same number of s_barrier (one per tap group + prologue + final); loaders wait for their own DMA
// with vmcnt before the barrier that publishes it (LDS-DMA data is ordered for another wave's ds_read only that way).
// Output channels are assigned to MFMA rows so that a lane's 16 accumulator registers are 16 CONSECUTIVE channels
// (row (e&3) + 8(e>>2) + 4h <-> channel 16h + e; the permutation is applied when the weights are packed).
// Rounding: conv + bias/temb is rounded to bf16 once (staging), the residual is added to that and rounded again -- the
// same two roundings as the reference's autocast path (conv output in bf16, then `skip + h`, UNet:701).
#include "conv27_kernel.h"

namespace {

template <int NCB, int FLIP>
__global__ void __launch_bounds__(512, 2) k_conv27(ConvArgs a) {
  conv27_body<NCB, FLIP, 0>(a);
}

template <int FLIP>
__global__ void __launch_bounds__(512, 2) k_conv27r(ConvArgs a) {
  conv27r_body<FLIP>(a);
}
// single-chunk 32-channel layers with at least a few tiles per column: the rolling-halo variant (conv27_kernel.h, MODE 3)
template <int FLIP>
int launch27r(ConvArgs a, int ntiles, int ny, hipStream_t st) {
  using KK = K<1, 3>;
  a.ntiles = ntiles;
  static const int dbg = mi_diag_knob("MI_C27_DBG");
  a.dbg = dbg & ~64;
  const int gx = mi_conv27_grid_x(ntiles, ny);
  if (a.stats && a.stats_chunks != 4 * gx) return MI_ERR_BAD_ARG;
  auto kern = k_conv27r<FLIP>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  static_assert(KK::LDS_TOTAL <= 160 * 1024, "LDS budget of the rolling variant");
  hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), (size_t)KK::LDS_TOTAL, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}

template <int NCB, int FLIP>
int launch27(ConvArgs a, int ntiles, int ny, hipStream_t st) {
  using KK = K<NCB, 0>;
  a.ntiles = ntiles;
  static const int dbg = mi_diag_knob("MI_C27_DBG");
  a.dbg = dbg;
  const int gx = mi_conv27_grid_x(ntiles, ny);
  if (a.stats && a.stats_chunks != 4 * gx) return MI_ERR_BAD_ARG;
  auto kern = k_conv27<NCB, FLIP>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), (size_t)KK::LDS_TOTAL, st, a);
  MI_CHECK_LAUNCH();
  if (a.dbg & 64) {  // diagnostic only: synchronises
    unsigned long long h[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c27_clk), sizeof(h));
    if (h[1]) fprintf(stderr, "[conv27<%d,%d>] main loop: %llu shader cycles in %.1f us -> %.0f MHz\n", NCB, FLIP, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
#ifdef MI_C27_DIAG_BAR
    fprintf(stderr, "[conv27<%d,%d>] cycles inside barriers (top 0 / 1 / 2..NG-2 / NG-1): compute wave 0: %llu %llu %llu %llu   helper wave 4: %llu %llu %llu %llu\n", NCB,
            FLIP, h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
    fprintf(stderr, "[conv27<%d,%d>] helper wave 4 segments: top-0 issue %llu, middle store slots %llu, last store slot %llu, final vmcnt wait %llu\n", NCB, FLIP, h[12],
            h[13], h[14], h[15]);
#endif
  }
  return 0;
}

}  // namespace

int mi_conv27_grid_x(int ntiles, int ny) {
  int gx = (256 / ny) / 8 * 8;  // one workgroup per CU over all cout groups, a multiple of 8 (one slot set per XCD class)
  if (gx < 8) gx = 8;
  const int need = (ntiles + 7) / 8 * 8;
  return gx > need ? need : gx;
}

int mi_launch_conv27(const ConvArgs& a, int NCB, int flip, int ntiles, int ny, hipStream_t st) {
  if (a.g.TD != 4 || a.g.TH != 8 || a.g.TW != 8 || (a.x_cs & 7) || (a.Cin & 7)) return MI_ERR_BAD_ARG;
  if ((a.y_cs & 7) || (a.Cout & 7) || a.y_bytes == 0) return MI_ERR_UNSUPPORTED;  // whole 16-byte pieces only (the caller falls back)
  if (a.res && (a.res_bytes == 0 || (a.res_cs & 7))) return MI_ERR_UNSUPPORTED;  // (>= 4 GiB is not reachable with the x_bytes limit)
  static const int use_roll = [] { const char* e = getenv("MI_C27_ROLL"); return e ? atoi(e) : 1; }();  // 0: the plain kernel (A/B runs)
  if (NCB == 1 && a.nchunks == 1 && use_roll && a.g.tilesD >= 4 && a.Cin >= 32)
    return flip ? launch27r<1>(a, ntiles, ny, st) : launch27r<0>(a, ntiles, ny, st);
  if (NCB == 1) return flip ? launch27<1, 1>(a, ntiles, ny, st) : launch27<1, 0>(a, ntiles, ny, st);
  if (NCB == 2) return flip ? launch27<2, 1>(a, ntiles, ny, st) : launch27<2, 0>(a, ntiles, ny, st);
  return MI_ERR_BAD_ARG;
}
