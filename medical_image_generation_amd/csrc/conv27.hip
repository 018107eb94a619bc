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
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"
#include "medimgen_hip.h"

namespace {

constexpr int HROW = 640, HSLICE = 6400;   // dense halo image: 10 voxels x 64 B per row, 10 rows per slice, 6 slices
constexpr int HALO_VOX = 600;
constexpr int HALO_BYTES = 40960;          // image (38400 B) rounded up to 40 whole 1-KiB DMA pieces
constexpr int RD = 3;                      // ring slots: weight groups are requested 2 groups ahead

template <int NCB> struct Cfg;
template <> struct Cfg<1> { static constexpr int GT = 9; };  // 3 groups of 9 taps: the ring holds a whole chunk
template <> struct Cfg<2> { static constexpr int GT = 3; };  // 9 groups of 3 taps

typedef __attribute__((address_space(3))) void lds_void;

// Ablation builds (tools/diag/c27_ablate.sh; never the shipped library): -DMI_C27_DIAG_A / _B drop the weight / activation fragment
// reads of the compute waves, _HALO / _W the helper waves' LDS-DMA, _NOBAR every s_barrier.  Results are garbage; every wave still
// runs the same loop to the same exit, and no address leaves its buffer.
#ifdef MI_C27_DIAG_NOBAR
#define C27_BARRIER() do {} while (0)
#else
#define C27_BARRIER() __builtin_amdgcn_s_barrier()
#endif
// -DMI_C27_DIAG_BAR: shader cycles spent inside the barriers, by position in the image (top 0 / top 1 / tops 2..NG-2 / top NG-1)
#ifdef MI_C27_DIAG_BAR
#define C27_BARRIER_T(acc) do { const unsigned long long t0__ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_barrier(); (acc) += __builtin_amdgcn_s_memtime() - t0__; } while (0)
#define C27_T0() const unsigned long long tt0__ = __builtin_amdgcn_s_memtime()
#define C27_T1(acc) (acc) += __builtin_amdgcn_s_memtime() - tt0__
#else
#define C27_BARRIER_T(acc) C27_BARRIER()
#define C27_T0() do {} while (0)
#define C27_T1(acc) do {} while (0)
#endif

__device__ unsigned long long g_c27_clk[16];  // diagnostic (MI_C27_DBG & 64): shader-clock and 100 MHz real-time ticks of workgroup 0's main loop

template <int NCB>
struct K {
  static constexpr int GT = Cfg<NCB>::GT, NG = 27 / GT;
  static constexpr int FRAGS = GT * 2 * NCB, GROUP_BYTES = FRAGS * 1024, RING = RD * GROUP_BYTES;
  static constexpr int VOXP = NCB * 64;        // staging bytes per voxel (bf16), 16-byte slots XOR-swizzled with the voxel index
  static constexpr int PV = NCB * 4;           // 16-byte pieces per voxel
  static constexpr int STG_WAVE = 64 * VOXP;   // one compute wave's 64 voxels
  static constexpr int RING0 = 2 * HALO_BYTES, STG0 = RING0 + RING, AV0 = STG0 + 4 * STG_WAVE;
  static constexpr int AV_WAVE = NCB * 2 * 64;  // per compute wave: bias (+ time embedding) of its accumulator channels [cb][h][16] fp32
  static constexpr int LDS_TOTAL = AV0 + 4 * AV_WAVE;
  static constexpr int PPT = (PV + NG - 2) / (NG - 1);  // store-epilogue pieces (of PV per helper wave and tile) processed per tap group
  // The next image's halo pieces are dealt over the first NH tops: an LDS-DMA instruction issued beside the compute waves' fragment
  // reads costs its wave 110-190 cycles, and a top that carries all 10 of them next to its weight pieces takes twice a tap group's
  // time (measured, 64->64 @128^3: the compute waves spent 12 % of the kernel inside the top-1 barrier).  Not at top NG-2: what is
  // issued there is retired only together with that top's stores.
#ifndef MI_C27_NH
#define MI_C27_NH 1
#endif
  static constexpr int NH = NG - 2 < MI_C27_NH ? (NG - 2 < 1 ? 1 : NG - 2) : MI_C27_NH;
  static constexpr int hbeg(int j) { return j >= NH ? 10 : (j * 10) / NH; }  // first piece of top j (10 = HPW)
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int OFF>
__device__ __forceinline__ void lds_read16(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read16_a(u32x4& dst, unsigned addr) {
#ifdef MI_C27_DIAG_A
  asm volatile("" : "+v"(dst) : "v"(addr));
#else
  lds_read16<OFF>(dst, addr);
#endif
}
template <int OFF>
__device__ __forceinline__ void lds_read16_b(u32x4& dst, unsigned addr) {
#ifdef MI_C27_DIAG_B
  asm volatile("" : "+v"(dst) : "v"(addr));
#else
  lds_read16<OFF>(dst, addr);
#endif
}

template <int NCB>
struct Frags {
  u32x4 b[2][2];    // [ks][vb]
  u32x4 a[2][NCB];  // [ks][cb]
};
template <int NCB>
__device__ __forceinline__ void wait_frags(Frags<NCB>& f) {
  if constexpr (NCB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]),
                   "+v"(f.a[1][1]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.a[0][0]), "+v"(f.a[1][0]));
  __builtin_amdgcn_sched_barrier(0);
}

// Fragment read Q of tap T (window mirrored for the data gradient), in FIRST-USE order of the tap's MFMAs:
//   NCB = 2:  b00 a00 a01 b01 | b10 a10 a11 b11        NCB = 1:  b00 a0 b01 | b10 a1 b11        (b[ks][vb], a[ks][cb])
// bb[vb][th][ks] = lane bases inside the halo buffer, ab = lane base inside the ring slot of T's group.
template <int Q, int T, int NCB, int FLIP>
__device__ __forceinline__ void issue_one(Frags<NCB>& f, const unsigned (&bb)[2][3][2], unsigned ab) {
  constexpr int U = FLIP ? 26 - T : T;
  constexpr int TH = (U / 3) % 3;
  constexpr int BOFF = (U / 9) * HSLICE + TH * HROW + (U % 3) * 64;
  constexpr int t = T % Cfg<NCB>::GT;
  constexpr int HALF = 2 + NCB;  // reads per k-step
  constexpr int ks = Q / HALF, q = Q % HALF;
  if constexpr (NCB == 2) {
    if constexpr (q == 0) lds_read16_b<BOFF>(f.b[ks][0], bb[0][TH][ks]);
    else if constexpr (q == 1) lds_read16_a<((t * 2 + ks) * 2 + 0) * 1024>(f.a[ks][0], ab);
    else if constexpr (q == 2) lds_read16_a<((t * 2 + ks) * 2 + 1) * 1024>(f.a[ks][1], ab);
    else lds_read16_b<BOFF>(f.b[ks][1], bb[1][TH][ks]);
  } else {
    if constexpr (q == 0) lds_read16_b<BOFF>(f.b[ks][0], bb[0][TH][ks]);
    else if constexpr (q == 1) lds_read16_a<(t * 2 + ks) * 1024>(f.a[ks][0], ab);
    else lds_read16_b<BOFF>(f.b[ks][1], bb[1][TH][ks]);
  }
}
template <int T, int NCB, int FLIP>
__device__ __forceinline__ void issue_frags(Frags<NCB>& f, const unsigned (&bb)[2][3][2], unsigned ab) {
  issue_one<0, T, NCB, FLIP>(f, bb, ab); issue_one<1, T, NCB, FLIP>(f, bb, ab); issue_one<2, T, NCB, FLIP>(f, bb, ab);
  issue_one<3, T, NCB, FLIP>(f, bb, ab); issue_one<4, T, NCB, FLIP>(f, bb, ab); issue_one<5, T, NCB, FLIP>(f, bb, ab);
  if constexpr (NCB == 2) { issue_one<6, T, NCB, FLIP>(f, bb, ab); issue_one<7, T, NCB, FLIP>(f, bb, ab); }
}

// One tap = 4*NCB MFMAs, each followed by its share of the NEXT tap's fragment reads (1 per MFMA; 2,2,1,1 for NCB = 1).
// LDS returns in order, so before MFMA k a COUNTED lgkmcnt suffices: (reads of this tap not needed yet) + (reads of the next tap
// already issued).  The wait names the MFMA's operands ("+v") and a sched_barrier pins the order (cdna guide 5.7 (ii), rule 18).
template <int KI, int NCB>
__device__ __forceinline__ void wait_operands(Frags<NCB>& f) {
  if constexpr (NCB == 2) {
    constexpr int ks = KI / 4, vb = (KI / 2) % 2, cb = KI % 2;
    constexpr int N = (KI == 3 || KI == 7) ? 7 : 6;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a[ks][cb]), "+v"(f.b[ks][vb]) : "i"(N));
  } else {
    constexpr int ks = KI / 2, vb = KI % 2;
    constexpr int N = KI == 0 ? 4 : 5;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a[ks][0]), "+v"(f.b[ks][vb]) : "i"(N));
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int KI, int TN, int NCB, int FLIP>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[2][NCB], Frags<NCB>& cur, Frags<NCB>& nxt, const unsigned (&bb)[2][3][2], unsigned ab) {
  wait_operands<KI, NCB>(cur);
  if constexpr (NCB == 2) {
    constexpr int ks = KI / 4, vb = (KI / 2) % 2, cb = KI % 2;
    acc[vb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[ks][cb]), __builtin_bit_cast(bf16x8, cur.b[ks][vb]),
                                                          acc[vb][cb], 0, 0, 0);
    issue_one<KI, TN, NCB, FLIP>(nxt, bb, ab);
  } else {
    constexpr int ks = KI / 2, vb = KI % 2;
    acc[vb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[ks][0]), __builtin_bit_cast(bf16x8, cur.b[ks][vb]),
                                                         acc[vb][0], 0, 0, 0);
    if constexpr (KI == 0) { issue_one<0, TN, NCB, FLIP>(nxt, bb, ab); issue_one<1, TN, NCB, FLIP>(nxt, bb, ab); }
    else if constexpr (KI == 1) { issue_one<2, TN, NCB, FLIP>(nxt, bb, ab); issue_one<3, TN, NCB, FLIP>(nxt, bb, ab); }
    else if constexpr (KI == 2) issue_one<4, TN, NCB, FLIP>(nxt, bb, ab);
    else issue_one<5, TN, NCB, FLIP>(nxt, bb, ab);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// MFMAs of the tap held (or arriving) in `cur`, interleaved with the reads of tap TN into `nxt`
template <int TN, int NCB, int FLIP>
__device__ __forceinline__ void tap_body(f32x16 (&acc)[2][NCB], Frags<NCB>& cur, Frags<NCB>& nxt, const unsigned (&bb)[2][3][2], unsigned ab) {
  mfma_step<0, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
  mfma_step<1, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
  mfma_step<2, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
  mfma_step<3, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
  if constexpr (NCB == 2) {
    mfma_step<4, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
    mfma_step<5, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
    mfma_step<6, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
    mfma_step<7, TN, NCB, FLIP>(acc, cur, nxt, bb, ab);
  }
}

// ------------------------------------------------------------------------------------------------ image sequence
// An "image" = one 32-channel chunk of one tile.  Every wave walks the same sequence (wave-uniform scalars).
struct Seq {
  int tile, ch, n, d0, h0, w0;          // current image
  int ntile, nch, nn, nd0, nh0, nw0;    // next image (ntile < 0: none)
  TileWalk walk;                        // digits of the furthest tile looked at (the next one once seq_next has stepped)
};
__device__ __forceinline__ void seq_next(Seq& q, const ConvArgs& a, int tile_step, int tile_last) {
  if (q.ch + 1 < a.nchunks) { q.ntile = q.tile; q.nch = q.ch + 1; q.nn = q.n; q.nd0 = q.d0; q.nh0 = q.h0; q.nw0 = q.w0; }
  else if (q.tile + tile_step < tile_last) { q.ntile = q.tile + tile_step; q.nch = 0; walk_step(q.walk, a.g); walk_origin(q.walk, a.g, q.nn, q.nd0, q.nh0, q.nw0); }
  else { q.ntile = -1; q.nch = 0; q.nn = q.nd0 = q.nh0 = q.nw0 = 0; }
}
__device__ __forceinline__ void seq_advance(Seq& q) {
  q.tile = q.ntile; q.ch = q.nch; q.n = q.nn; q.d0 = q.nd0; q.h0 = q.nh0; q.w0 = q.nw0;
}

// ------------------------------------------------------------------------------------------------ compute waves
template <int NCB>
struct CState {
  f32x16 acc[2][NCB];
  Frags<NCB> fr[2];
  unsigned bcur[2][3][2];      // halo fragment-read lane bases inside the buffer of the image being consumed
  unsigned abase, abase_next;  // A fragment-read bases: ring slot of the current / the next group
  int av_n;                    // image index the LDS bias table of this wave was loaded for
  int cur;                     // halo buffer of the current image
  int slot;                    // ring slot of the previous group
  int resident;                // single chunk whose weight groups all fit the ring: loaded once, inner barriers skipped
  unsigned long long bw[4];    // MI_C27_DIAG_BAR
};

template <int J, int NCB>
__device__ __forceinline__ void compute_top(CState<NCB>& s, int lane) {
  using KK = K<NCB>;
  // group J+1's weights (and, at the last group, the next image's halo) are in LDS for every wave.  With the whole chunk resident in
  // the ring the inner barriers publish nothing: without them the helper waves have the whole image, not one tap group, for the halo
  // DMA and the store epilogue (measured, 32->32 @128^3: the compute waves spent 16 % of the kernel inside the top-1 barrier waiting
  // for helpers still issuing the halo)
  if (J == 0 || J == KK::NG - 1 || !s.resident) C27_BARRIER_T(s.bw[J < 2 ? J : (J == KK::NG - 1 ? 3 : 2)]);
  const int sj = s.slot + 1 == RD ? 0 : s.slot + 1;  // slot of group J
  const int sj1 = sj + 1 == RD ? 0 : sj + 1;         // slot of group J+1
  s.abase = KK::RING0 + sj * KK::GROUP_BYTES + lane * 16;
  s.abase_next = KK::RING0 + sj1 * KK::GROUP_BYTES + lane * 16;
  s.slot = sj;
}

template <int T, int PAR, int NCB, int FLIP>
__device__ __forceinline__ void taps(CState<NCB>& s, int lane) {
  using KK = K<NCB>;
  if constexpr (T < 27) {
    constexpr int cur = (T + PAR) & 1;
    if constexpr (T % KK::GT == 0) compute_top<T / KK::GT, NCB>(s, lane);
    if constexpr (T + 1 < 27) {
#ifdef MI_C27_DIAG_TAPS8  // timing probe: what would an 8-tap (phase) convolution cost in this kernel's structure?  (taps 8..25 skipped)
      if constexpr (T < 8)
#endif
      tap_body<T + 1, NCB, FLIP>(s.acc, s.fr[cur], s.fr[cur ^ 1], s.bcur, (T + 1) % KK::GT == 0 ? s.abase_next : s.abase);
    } else {  // next: first tap of the next image -- other halo buffer, next group's slot
      const unsigned delta = s.cur ? 0u - (unsigned)HALO_BYTES : (unsigned)HALO_BYTES;
#pragma unroll
      for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) s.bcur[vb][th][ks] += delta;
      tap_body<0, NCB, FLIP>(s.acc, s.fr[cur], s.fr[cur ^ 1], s.bcur, s.abase_next);
    }
    taps<T + 1, PAR, NCB, FLIP>(s, lane);
  }
}

// bias (+ time embedding) table of this wave in LDS: [cb][h][16] fp32; the accumulators start from it (the MFMA adds on top)
template <int NCB>
__device__ __forceinline__ void load_av(CState<NCB>& s, const ConvArgs& a, char* lds, int y, int wave, int lane, int n) {
  using KK = K<NCB>;
  float* tab = (float*)(lds + KK::AV0 + wave * KK::AV_WAVE);
  for (int i = lane; i < NCB * 32; i += 64) {  // i = cb*32 + 16h + e  == channel offset inside this workgroup's cout range
    const int co = y * NCB * 32 + i;
    tab[i] = (a.addvec && co < a.Cout) ? a.addvec[(int64_t)n * a.addvec_stride + co] : 0.f;
  }
  s.av_n = n;
}
template <int NCB>
__device__ __forceinline__ void init_acc(CState<NCB>& s, char* lds, int wave, int lane) {
  using KK = K<NCB>;
  const float* tab = (const float*)(lds + KK::AV0 + wave * KK::AV_WAVE) + 16 * (lane >> 5);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *(const f32x4*)(tab + cb * 32 + 4 * q);
#pragma unroll
      for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int e = 0; e < 4; ++e) s.acc[vb][cb][4 * q + e] = v[e];
    }
}

// accumulators -> bf16 staging tile of this wave: voxel v = vb*32 + r, channels cb*32 + 16h + (0..15) = slots cb*4 + 2h, +1
template <int NCB>
__device__ __forceinline__ void stage_acc(CState<NCB>& s, char* lds, int wave, int lane) {
  using KK = K<NCB>;
  const int r = lane & 31, h = lane >> 5;
  char* stg = lds + KK::STG0 + wave * KK::STG_WAVE;
#pragma unroll
  for (int vb = 0; vb < 2; ++vb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int v = vb * 32 + r;
      F8 lo, hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) { lo.v[e] = s.acc[vb][cb][e]; hi.v[e] = s.acc[vb][cb][8 + e]; }
      const int s0 = cb * 4 + 2 * h, sw = v & (KK::PV - 1);
      *(u32x4*)(stg + v * KK::VOXP + ((s0 ^ sw) * 16)) = pack8(lo);
      *(u32x4*)(stg + v * KK::VOXP + (((s0 + 1) ^ sw) * 16)) = pack8(hi);
    }
}

template <int NCB, int FLIP>
__device__ __forceinline__ void compute_role(const ConvArgs& a, char* lds, int y, int wave, int lane, int tile0, int tile_step, int tile_last) {
  CState<NCB> s;
  {
    const int r = lane & 31, h = lane >> 5, row = r >> 3, col = r & 7;
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s.bcur[vb][th][ks] = wave * HSLICE + (vb * 4 + row) * HROW + col * 64 + (((ks * 2 + h) ^ ((row + th) & 3)) * 16);
        }
  }
  Seq q;
  q.tile = tile0; q.ch = 0;
  walk_init(q.walk, a.g, tile0, tile_step);
  walk_origin(q.walk, a.g, q.n, q.d0, q.h0, q.w0);
  load_av<NCB>(s, a, lds, y, wave, lane, q.n);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  init_acc<NCB>(s, lds, wave, lane);
  s.cur = 0;
  s.bw[0] = s.bw[1] = s.bw[2] = s.bw[3] = 0;
  s.resident = a.nchunks == 1 && K<NCB>::NG <= RD;
  s.slot = RD - 1;  // "slot of group -1": compute_top<0> steps to slot 0
  s.abase = s.abase_next = K<NCB>::RING0 + lane * 16;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing of this wave's is in flight across the barriers
  C27_BARRIER();  // prologue: first halo image and weight groups 0, 1 are in LDS
  issue_frags<0, NCB, FLIP>(s.fr[0], s.bcur, s.abase);
  wait_frags<NCB>(s.fr[0]);
  unsigned long long t0 = 0, r0 = 0;
  if (a.dbg & 64) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  int par = 0;
  while (true) {
    seq_next(q, a, tile_step, tile_last);
    if (par) taps<0, 1, NCB, FLIP>(s, lane);
    else taps<0, 0, NCB, FLIP>(s, lane);
    par ^= 1;
    if (q.ch == a.nchunks - 1) {  // tile finished (the next image's first fragments are already on their way)
      if (s.resident) C27_BARRIER_T(s.bw[1]);  // the helper waves have read the previous tile out of the staging tile (helper_role)
      if (!(a.dbg & 1)) stage_acc<NCB>(s, lds, wave, lane);
      if (q.ntile >= 0 && q.nn != s.av_n) {  // image index changed: once in thousands of tiles
        load_av<NCB>(s, a, lds, y, wave, lane, q.nn);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      init_acc<NCB>(s, lds, wave, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging written before the barrier that hands it to the store waves
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q.ntile < 0) break;
    s.cur ^= 1;
    seq_advance(q);
  }
  if ((a.dbg & 64) && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    g_c27_clk[0] = t1 - t0; g_c27_clk[1] = r1 - r0;
    for (int i = 0; i < 4; ++i) g_c27_clk[4 + i] = s.bw[i];
  }
  C27_BARRIER();  // final: the last tile's staging is complete
}

// ------------------------------------------------------------------------------------------------ helper waves (4-7)
// Issuing an LDS-DMA instruction costs the SIMD ~100 cycles that its compute wave cannot use (measured: with two waves issuing all
// weight pieces, the compute waves on those two SIMDs needed 60 cycles per MFMA instead of 32 and everybody waited for them at the
// barriers).  So every helper wave does a quarter of everything: a quarter of each weight group, a quarter of the halo image,
// one d-slice of the store epilogue.
//
// vmcnt bookkeeping of a helper (operations complete in issue order).  Per image, in program order:
//   top 0 :  [residual loads of the tile being stored: R]  [weights of group 2]  [halo quarter of the next image: HPW]
//   top J :  [weights of group J+2]            then (store slots 1 .. NG-2, after the barrier)  [PPT stores]
//   top NG-1 : [PPT stores] before the wait
// At top J the weights of group J+1 (issued at top J-1) must have landed: the only younger operations are, for J = 1, the halo
// quarter, and for J >= 2 the stores of slot J-1 (plus, at top NG-1, the slot issued just before the wait).  Waiting for them
// also retires everything older: the residual loads (usable from slot 1 on without any further wait) and, from top 2 on, the halo.
constexpr int HPW = 10;  // halo pieces per helper wave and image

template <int NCB>
__device__ __forceinline__ void issue_A(const ConvArgs& a, char* lds, int y, int hl, int lane, int ch, int j, int slot) {
  using KK = K<NCB>;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wpk, a.wpk_bytes);
  // fragments are packed [cout group][chunk][tap][ks][cb] and every chunk of a k3 s1 conv has all 27 taps: no table lookup
  // (a load inside the loop would be a VECTOR load -- the kernel stores to global memory -- and drain the DMA queue)
  const int wfrag = (y * a.nchunks + ch) * (27 * 2 * NCB) + j * KK::FRAGS;
#ifdef MI_C27_DIAG_W
  return;
#endif
#pragma unroll
  for (int i = 0; i < (KK::FRAGS + 3) / 4; ++i) {
    const int f = hl + 4 * i;
    if (f < KK::FRAGS)  // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + KK::RING0 + slot * KK::GROUP_BYTES + f * 1024), 16, lane * 16,
                                               (wfrag + f) * 1024, 0, 0);
  }
}

template <int K0, int K1>  // pieces K0 .. K1-1 of this wave's HPW
__device__ __forceinline__ void issue_halo(const ConvArgs& a, char* lds, const int (&hp)[HPW], const unsigned (&hoff)[HPW], int hl, int buf, int valid,
                                           int n, int d0, int h0, int w0, int src_c0) {
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
#ifdef MI_C27_DIAG_HALO
  return;
#endif
  // byte offset of the halo origin voxel (d0-1, h0-1, w0-1), channel src_c0: wave-uniform, lives in an SGPR (mod 2^32; tensors < 4 GiB)
  const unsigned base = (unsigned)((((n * a.Di + d0 - 1) * a.Hi + (h0 - 1)) * a.Wi + (w0 - 1)) * a.x_cs + src_c0) * 2u;
  // A helper wave shares its SIMD's vector issue with a compute wave that is issuing MFMAs: every VALU instruction here is paid for by
  // the whole workgroup at the next barrier (measured: with ~25 address instructions per piece the compute waves spent 16-20 % of
  // the kernel waiting for the helpers).  A tile whose halo lies inside the tensor needs none: the per-lane part of the address is a
  // kernel-lifetime constant (hoff) and the tile's part goes into the instruction's scalar offset.
  const bool interior = (valid != 0) & (d0 >= 1) & (d0 + 5 <= a.Di) & (h0 >= 1) & (h0 + 9 <= a.Hi) & (w0 >= 1) & (w0 + 9 <= a.Wi) & (src_c0 + 32 <= a.Cin);
  if (interior) {
#pragma unroll
    for (int k = K0; k < K1; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + buf * HALO_BYTES + (hl + 4 * k) * 1024), 16, hoff[k], (int)base, 0, 0);
    return;
  }
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int pk = hp[k];
    const int gd = d0 - 1 + ((pk >> 16) & 255), gh = h0 - 1 + ((pk >> 8) & 255), gw = w0 - 1 + (pk & 255);
    const int c = src_c0 + ((pk >> 24) & 3) * 8;
    const bool ok = (valid != 0) & (pk >= 0) & ((unsigned)gd < (unsigned)a.Di) & ((unsigned)gh < (unsigned)a.Hi) & ((unsigned)gw < (unsigned)a.Wi) &
                    (c + 8 <= a.Cin);  // (bitwise: one select, no branches)
    const unsigned off = ok ? hoff[k] + base : 0xfffffff0u;  // out of range -> zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + buf * HALO_BYTES + (hl + 4 * k) * 1024), 16, off, 0, 0, 0);
  }
}

// Store epilogue of one tile, this wave's d-slice (= compute wave hl's staging tile).  Piece p (0 .. PV-1): 16 voxels x PV slots per
// wave-instruction, so that consecutive lanes cover consecutive 16-byte slots of a voxel and then the next voxel along W.
template <int NCB>
struct Epi {
  int n, d0, h0, w0;        // the tile being stored
  int active;               // a tile is pending
  int full;                 // ... and lies inside the output with whole channel octets: no per-lane masks, addresses = lane constant + scalar
  unsigned ybase, rbase;    // byte offsets of the tile's origin voxel (channel y*NCB*32) in y / in the residual
  unsigned ylane, rlane, slane;  // this lane's constant part: voxel (lane / PV), slot (lane % PV) of a piece in y / residual / the staging tile
  unsigned yps, rps;        // bytes from one piece to the next (8 / PV rows of the tile)
  u32x4 res[K<NCB>::PV];    // residual pieces, loaded one tap group ahead of their use
  float sa[8], sq[8];       // a.stats: running sum / sum of squares of this lane's channel octet over the tiles of image sn
  int sn;
};
// (Statistics are compiled into the NCB = 1 variant only.  Measured same box, same call: with the code in both variants the C4 step
// takes 25.5 ms, 25.8 ms when it is compiled in but unused, 25.3 ms with it in neither variant or in NCB = 1 alone -- the NCB = 2
// kernel, 256 VGPRs and ~9000 instructions, loses more to the extra code than the fused statistics save.)
// Fold the lanes that share a channel octet (they sit PV apart) and store this wave's chunk of image e.sn; resets the sums.
template <int NCB>
__device__ __forceinline__ void stats_flush(Epi<NCB>& e, const ConvArgs& a, int y, int hl, int lane) {
  using KK = K<NCB>;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int m = KK::PV; m < 64; m <<= 1) {
      e.sa[j] += __shfl_xor(e.sa[j], m, 64);
      e.sq[j] += __shfl_xor(e.sq[j], m, 64);
    }
  }
  const int chunk = blockIdx.x * 4 + hl;
  if (lane < KK::PV) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = y * NCB * 32 + lane * 8 + j;
      if (c < a.Cout) *(float2*)(a.stats + (((int64_t)e.sn * a.Cout + c) * a.stats_chunks + chunk) * 2) = make_float2(e.sa[j], e.sq[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) e.sa[j] = e.sq[j] = 0.f;
}
// every image's entry of this wave's chunk starts at zero (a workgroup may not see tiles of every image)
template <int NCB>
__device__ __forceinline__ void stats_zero(const ConvArgs& a, int y, int hl, int lane) {
  using KK = K<NCB>;
  const int chunk = blockIdx.x * 4 + hl;
  if (lane < KK::PV)
    for (int n = 0; n < a.N; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = y * NCB * 32 + lane * 8 + j;
        if (c < a.Cout) *(float2*)(a.stats + (((int64_t)n * a.Cout + c) * a.stats_chunks + chunk) * 2) = make_float2(0.f, 0.f);
      }
}
template <int NCB, bool ST>
__device__ __forceinline__ void stats_begin_tile(Epi<NCB>& e, const ConvArgs& a, int y, int hl, int lane) {
  if (ST && a.stats && e.sn != e.n) {  // wave-uniform
    if (e.sn >= 0) stats_flush<NCB>(e, a, y, hl, lane);
    e.sn = e.n;
  }
}
// Per-lane border masks of piece p (tiles that stick out of the output only): false -> the lane's access is dropped
template <int NCB>
__device__ __forceinline__ bool epi_inside(const Epi<NCB>& e, const ConvArgs& a, int y, int hl, int lane, int p) {
  using KK = K<NCB>;
  const int q = p * 64 + lane, v = q / KK::PV, sidx = q % KK::PV;
  const int od = e.d0 + hl, oh = e.h0 + (v >> 3), ow = e.w0 + (v & 7);
  return (od < a.Do) & (oh < a.Ho) & (ow < a.Wo) & (y * NCB * 32 + sidx * 8 + 8 <= a.Cout);
}
template <int NCB>
__device__ __forceinline__ void epi_begin_tile(Epi<NCB>& e, const ConvArgs& a, int y, int n, int d0, int h0, int w0) {
  e.active = 1; e.n = n; e.d0 = d0; e.h0 = h0; e.w0 = w0;
  e.full = (d0 + 4 <= a.Do) & (h0 + 8 <= a.Ho) & (w0 + 8 <= a.Wo) & ((y + 1) * NCB * 32 <= a.Cout);
  const unsigned vox = (unsigned)(((n * a.Do + d0) * a.Ho + h0) * a.Wo + w0);
  e.ybase = (vox * (unsigned)a.y_cs + (unsigned)(y * NCB * 32)) * 2u;
  e.rbase = (vox * (unsigned)a.res_cs + (unsigned)(y * NCB * 32)) * 2u;
}
// The residual pieces are ordinary compiler-tracked buffer loads issued one tap group before their use, and only when there is a
// residual; the compiler places the vmcnt waits itself (conservatively: the helper wave waits for them almost at once, about a
// microsecond per tile of the layers that have a residual).  They used to be inline-asm loads that the compiler believed complete
// at issue, consumed behind the manual vmcnt plan.  That is unsound: whenever the compiler copies or re-homes the destination
// registers between issue and use (the phi copy behind a conditionally executed asm; live-range splits at 256 VGPRs; the home copy
// of a tied "+v" operand) it reads them before the data has landed -- NaNs at realistic sizes only (tests/test_kernels_gpu.py::
// test_conv_residual_at_size); 8^3 cases pass by luck because the loads return at once.
template <int NCB>
__device__ __forceinline__ void epi_issue_res(Epi<NCB>& e, const ConvArgs& a, int y, int hl, int lane, bool enable) {
  using KK = K<NCB>;
  if (!enable) return;  // (a pending tracked load makes the compiler drain the DMA queue at its waits: none without a residual)
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res, a.res ? a.res_bytes : 0u);
#pragma unroll
  for (int p = 0; p < KK::PV; ++p) {
    const unsigned so = e.rbase + (unsigned)p * e.rps;  // scalar
    if (e.full) {
      e.res[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)e.rlane, (int)so, 0));
    } else {
      const unsigned off = epi_inside<NCB>(e, a, y, hl, lane, p) ? e.rlane + so : 0xfffffff0u;
      e.res[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)off, 0, 0));
    }
  }
}
// The helper waves share their SIMDs' vector issue with the compute waves' MFMAs, so VALU instructions here are the scarce thing:
// an inside tile (e.full) costs no address arithmetic at all (lane constants + a scalar offset) and no statistics masks.
template <int P0, int CNT, int NCB, bool ST>
__device__ __forceinline__ void epi_process(Epi<NCB>& e, const ConvArgs& a, char* lds, int y, int hl, int lane, bool has_res) {
  using KK = K<NCB>;
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, a.y_bytes);
#pragma unroll
  for (int p = P0; p < P0 + CNT; ++p) {
    if (p >= KK::PV) break;
    u32x4 raw = *(const u32x4*)(lds + KK::STG0 + hl * KK::STG_WAVE + p * 1024 + e.slane);
    if (has_res) {
      F8 f = unpack8(raw), rr = unpack8(e.res[p]);
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] += rr.v[j];
      raw = pack8(f);
    }
    const unsigned so = e.ybase + (unsigned)p * e.yps;  // scalar
    // (stores are always issued -- masked lanes get an out-of-range offset: the store count is part of the vmcnt bookkeeping.  The
    // host only sends whole channel octets with an 8-aligned pitch here; ragged outputs stay on the table-driven kernel.)
    if (e.full) {
      if (ST && a.stats) {  // statistics of the ROUNDED values: what the consumer's GroupNorm will read
        const F8 f = unpack8(raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          e.sa[j] += f.v[j];
          e.sq[j] = fmaf(f.v[j], f.v[j], e.sq[j]);
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, raw), ry, (int)e.ylane, (int)so, 0);
    } else {
      const bool inside = epi_inside<NCB>(e, a, y, hl, lane, p);
      if (ST && a.stats) {
        const F8 f = unpack8(raw);
        const float mk = inside ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = mk * f.v[j];
          e.sa[j] += t;
          e.sq[j] = fmaf(t, f.v[j], e.sq[j]);
        }
      }
      const unsigned off = inside ? e.ylane + so : 0xfffffff0u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, raw), ry, (int)off, 0, 0);
    }
  }
}
// pieces of store slot J (1 .. NG-1)
template <int J, int NCB, bool ST>
__device__ __forceinline__ void epi_slot(Epi<NCB>& e, const ConvArgs& a, char* lds, int y, int hl, int lane, int j, bool has_res) {
  using KK = K<NCB>;
  if constexpr (J < KK::NG) {
    if (j == J) epi_process<(J - 1) * KK::PPT, KK::PPT, NCB, ST>(e, a, lds, y, hl, lane, has_res);
    else epi_slot<J + 1, NCB, ST>(e, a, lds, y, hl, lane, j, has_res);
  }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediates the helper's tops need)
__device__ __forceinline__ void wait_vm_dyn(int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;
    case 10: wait_vm<10>(); break;
    case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break;
    default: wait_vm<0>(); break;
  }
}
// halo pieces of top j (J0 <= j) of the NEXT image
template <int J0, int NCB>
__device__ __forceinline__ void halo_part(const ConvArgs& a, char* lds, const int (&hp)[HPW], const unsigned (&hoff)[HPW], int hl, int buf, const Seq& q, int j) {
  using KK = K<NCB>;
  if constexpr (J0 < KK::NH) {
    if (j == J0) {
#ifdef MI_C27_DIAG_HOT  // every halo image of this workgroup = the origin tile's: L2-warm after the first fetch
      issue_halo<KK::hbeg(J0), KK::hbeg(J0 + 1)>(a, lds, hp, hoff, hl, buf, q.ntile >= 0, 0, 4, 8, 8, q.nch * 32);
#else
      issue_halo<KK::hbeg(J0), KK::hbeg(J0 + 1)>(a, lds, hp, hoff, hl, buf, q.ntile >= 0, q.nn, q.nd0, q.nh0, q.nw0, q.nch * 32);
#endif
    } else halo_part<J0 + 1, NCB>(a, lds, hp, hoff, hl, buf, q, j);
  }
}

// ST: this instantiation carries the output-statistics code (forward kernel of the 32-channel variant only: see stats_flush)
template <int NCB, bool ST>
__device__ __forceinline__ void helper_role(const ConvArgs& a, char* lds, int y, int hl, int lane, int tile0, int tile_step, int tile_last) {
  using KK = K<NCB>;
  int hp[HPW];  // this lane's halo DMA pieces: (logical slot << 24) | (hd << 16) | (hh << 8) | hw, or -1
  unsigned hoff[HPW];  // ... and their byte offsets from the halo origin voxel
#pragma unroll
  for (int k = 0; k < HPW; ++k) {
    const int v = (hl + 4 * k) * 16 + (lane >> 2), p = lane & 3;
    const int hd = v / 100, rem = v - hd * 100, hh = rem / 10, hw = rem - hh * 10;
    hp[k] = v < HALO_VOX ? (((p ^ (hh & 3)) << 24) | (hd << 16) | (hh << 8) | hw) : -1;
    // (lanes past the image fetch the origin voxel into LDS padding nobody reads -- on the interior path; the border path masks them)
    hoff[k] = v < HALO_VOX ? (unsigned)(((hd * a.Hi + hh) * a.Wi + hw) * a.x_cs + (p ^ (hh & 3)) * 8) * 2u : 0u;
  }
  const bool has_res = a.res != nullptr;
  const bool resident = a.nchunks == 1 && KK::NG <= RD;  // the ring holds every group of the only chunk: load once
  int a_ch = 0, a_j = 0, slot = 0, issued = 0;
  auto issue_next_A = [&]() {
    if (!(resident && issued >= KK::NG)) issue_A<NCB>(a, lds, y, hl, lane, a_ch, a_j, slot);
    issued = issued < 1000 ? issued + 1 : issued;
    slot = slot + 1 == RD ? 0 : slot + 1;
    if (++a_j == KK::NG) { a_j = 0; a_ch = a_ch + 1 == a.nchunks ? 0 : a_ch + 1; }
  };
  Seq q;
  q.tile = tile0; q.ch = 0;
  walk_init(q.walk, a.g, tile0, tile_step);
  walk_origin(q.walk, a.g, q.n, q.d0, q.h0, q.w0);
  issue_halo<0, HPW>(a, lds, hp, hoff, hl, 0, 1, q.n, q.d0, q.h0, q.w0, 0);
  issue_next_A();
  issue_next_A();  // two groups ahead
  if (resident)    // ... or all of them: nothing publishes a group later (compute_top skips the inner barriers)
    for (int g = 2; g < KK::NG; ++g) issue_next_A();
  wait_vm<0>();
  C27_BARRIER();  // prologue
  Epi<NCB> e;
  e.active = 0; e.full = 0; e.n = e.d0 = e.h0 = e.w0 = 0;
  e.ybase = e.rbase = 0;
  {
    const int vl = lane / KK::PV, sidx = lane % KK::PV, row = vl >> 3, col = vl & 7;  // (pieces start at whole rows: v & (PV-1) == vl & (PV-1))
    e.ylane = (unsigned)(((hl * a.Ho + row) * a.Wo + col) * a.y_cs + sidx * 8) * 2u;
    e.rlane = (unsigned)(((hl * a.Ho + row) * a.Wo + col) * a.res_cs + sidx * 8) * 2u;
    e.slane = (unsigned)(vl * KK::VOXP + ((sidx ^ (vl & (KK::PV - 1))) * 16));
    e.yps = (unsigned)((8 / KK::PV) * a.Wo * a.y_cs) * 2u;
    e.rps = (unsigned)((8 / KK::PV) * a.Wo * a.res_cs) * 2u;
  }
#pragma unroll
  for (int p = 0; p < KK::PV; ++p) e.res[p] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 8; ++j) e.sa[j] = e.sq[j] = 0.f;
  e.sn = -1;
  if (ST && a.stats) stats_zero<NCB>(a, y, hl, lane);
  int cur = 0;
  [[maybe_unused]] unsigned long long hbw[4] = {0, 0, 0, 0}, hseg[4] = {0, 0, 0, 0};  // MI_C27_DIAG_BAR
  while (true) {
    seq_next(q, a, tile_step, tile_last);
    const bool epi = e.active != 0;  // wave-uniform
    if (resident) {
      // Resident weights: three barriers per image and the two jobs of this wave under different ones.  The halo of the next image
      // is issued after B0 and must have landed at the top-(NG-1) barrier (the compute waves prefetch their first fragments across
      // the image boundary); the previous tile's staging is stored AFTER that barrier and released by a third one that the compute
      // waves pass just before they overwrite the staging tile, a tap group later.  With both jobs due at the same barrier the
      // helpers were the critical path (measured, 32->32 @128^3: ~2000 cycles to issue 10 LDS-DMA pieces beside the MFMA stream +
      // ~1450 for the stores, against ~2300-2900 for the compute waves' two tap groups: they waited 16-23 % of the kernel there).
      C27_BARRIER_T(hbw[0]);
      { C27_T0();
      if (epi) stats_begin_tile<NCB, ST>(e, a, y, hl, lane);
      epi_issue_res<NCB>(e, a, y, hl, lane, epi && has_res);
      halo_part<0, NCB>(a, lds, hp, hoff, hl, cur ^ 1, q, 0);
      C27_T1(hseg[0]); }
      { C27_T0(); wait_vm<0>(); C27_T1(hseg[3]); }  // (the previous tile's stores are a whole image old)
      C27_BARRIER_T(hbw[3]);
      { C27_T0();
      if (epi) {
        epi_process<0, KK::PV, NCB, ST>(e, a, lds, y, hl, lane, has_res);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done
        e.active = 0;
      }
      C27_T1(hseg[2]); }
      C27_BARRIER_T(hbw[1]);
      if (q.ch == a.nchunks - 1 && !(a.dbg & 1)) epi_begin_tile<NCB>(e, a, y, q.n, q.d0, q.h0, q.w0);
      if (q.ntile < 0) break;
      cur ^= 1;
      seq_advance(q);
      continue;
    }
    // ---- top 0
    wait_vm<0>();  // group 1's weights are this wave's youngest operation
    C27_BARRIER_T(hbw[0]);
    { C27_T0();
    if (epi) stats_begin_tile<NCB, ST>(e, a, y, hl, lane);
    epi_issue_res<NCB>(e, a, y, hl, lane, epi && has_res);
    issue_next_A();
    halo_part<0, NCB>(a, lds, hp, hoff, hl, cur ^ 1, q, 0);
    C27_T1(hseg[0]); }
    // ---- tops 1 .. NG-2.  Issue order inside a top: weights of group j+2, halo part j, stores of slot j (after the barrier);
    // at top j the weights issued at top j-1 must have landed, i.e. everything but the halo part and the stores of top j-1.
    for (int j = 1; j < KK::NG - 1; ++j) {
      const int hprev = KK::hbeg(j) - KK::hbeg(j - 1);  // (values of a small table: j is a loop counter)
      wait_vm_dyn(hprev + ((epi && j >= 2) ? KK::PPT : 0));
      C27_BARRIER_T(hbw[j == 1 ? 1 : 2]);
      issue_next_A();
      halo_part<1, NCB>(a, lds, hp, hoff, hl, cur ^ 1, q, j);
      { C27_T0(); if (epi) epi_slot<1, NCB, ST>(e, a, lds, y, hl, lane, j, has_res); C27_T1(hseg[1]); }
    }
    // ---- top NG-1: the last store slot runs BEFORE the barrier (a single-chunk tile's compute waves overwrite the staging tile right
    // after it); the next image's halo and group NG's weights must have landed, the stores of the last two slots may fly
    { C27_T0();
    if (epi) {
      epi_process<(KK::NG - 2) * KK::PPT, KK::PPT, NCB, ST>(e, a, lds, y, hl, lane, has_res);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done
      e.active = 0;
    }
    C27_T1(hseg[2]); }
    { C27_T0();
    if (epi) wait_vm<2 * KK::PPT>(); else wait_vm<0>();  // younger than group NG's weights: the stores of the last two slots
    C27_T1(hseg[3]); }
    C27_BARRIER_T(hbw[3]);
    issue_next_A();
    if (q.ch == a.nchunks - 1 && !(a.dbg & 1)) epi_begin_tile<NCB>(e, a, y, q.n, q.d0, q.h0, q.w0);
    if (q.ntile < 0) break;
    cur ^= 1;
    seq_advance(q);
  }
  wait_vm<0>();
  C27_BARRIER();  // final: the last tile's staging is complete
#ifdef MI_C27_DIAG_BAR
  if ((a.dbg & 64) && blockIdx.x == 0 && blockIdx.y == 0 && hl == 0 && lane == 0)
    for (int i = 0; i < 4; ++i) { g_c27_clk[8 + i] = hbw[i]; g_c27_clk[12 + i] = hseg[i]; }
#endif
  if (e.active) {
    stats_begin_tile<NCB, ST>(e, a, y, hl, lane);
    epi_issue_res<NCB>(e, a, y, hl, lane, has_res);
    wait_vm<0>();
    epi_process<0, KK::PV, NCB, ST>(e, a, lds, y, hl, lane, has_res);
  }
  if (ST && a.stats && e.sn >= 0) stats_flush<NCB>(e, a, y, hl, lane);
  wait_vm<0>();
}

template <int NCB, int FLIP>
__global__ void __launch_bounds__(512, 2) k_conv27(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int y = blockIdx.y;
  int tile_last, tile_step;
  const int tile0 = first_tile(a.ntiles, tile_last, tile_step);
  if (tile0 >= tile_last) {  // whole workgroup, before any barrier
    if (NCB == 1 && FLIP == 0 && a.stats && wave >= 4) stats_zero<NCB>(a, y, wave - 4, lane);
    return;
  }
  if (a.dbg & 2) { if (wave < 4) __builtin_amdgcn_s_setprio(3); }   // experiment knobs (MI_C27_DBG): static wave priority
  if (a.dbg & 4) { if (wave >= 4) __builtin_amdgcn_s_setprio(3); }
  if (wave < 4) compute_role<NCB, FLIP>(a, lds, y, wave, lane, tile0, tile_step, tile_last);
  else helper_role<NCB, NCB == 1 && FLIP == 0>(a, lds, y, wave - 4, lane, tile0, tile_step, tile_last);
}

template <int NCB, int FLIP>
int launch27(ConvArgs a, int ntiles, int ny, hipStream_t st) {
  using KK = K<NCB>;
  a.ntiles = ntiles;
  static const char* dbg_env = getenv("MI_C27_DBG");
  a.dbg = dbg_env ? atoi(dbg_env) : 0;
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
  if (NCB == 1) return flip ? launch27<1, 1>(a, ntiles, ny, st) : launch27<1, 0>(a, ntiles, ny, st);
  if (NCB == 2) return flip ? launch27<2, 1>(a, ntiles, ny, st) : launch27<2, 0>(a, ntiles, ny, st);
  return MI_ERR_BAD_ARG;
}
