// Kernel body of the role-specialised LDS-DMA convolution, shared by conv27.hip (MODE 0: 3x3x3 stride 1) and convph.hip (MODE 1 / 2:
// the factor-2 phase convolutions).  Each translation unit instantiates its own kernels from it: separate objects, so that the
// instantiations of one family do not perturb the register allocation of the other.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"
#include "medimgen_hip.h"

namespace {

constexpr int HROW = 640, HSLICE = 6400;   // dense halo image: 10 voxels x 64 B per row, 10 rows per slice, 6 slices
constexpr int HALO_VOX = 600;
constexpr int HALO_BYTES = 40960;          // image (38400 B) rounded up to 40 whole 1-KiB DMA pieces
constexpr int RD = 3;                      // ring slots: weight groups are requested 2 groups ahead

// MODE 0: the 3x3x3 stride-1 convolution (27 taps per image).  MODE 1 / 2 (convph.hip): the factor-2 phase convolutions -- an image is
// one (tile, phase or class, 32-channel chunk) and has the 8 taps of a 2x2x2 box:
//   MODE 1 "scatter": input on the tile grid, output on the twice finer grid; output phase p in {0,1}^3 of tile voxel i is
//           y[2i + p] = sum_t A[p][t] x[i + p + t - 1]   (nearest-upsample + 3x3x3 conv: UNet:569-588; the data gradient of a k3 s2 p1 conv)
//   MODE 2 "gather": input on the twice finer grid, output on the tile grid; input class c in {0,1}^3 contributes
//           y[i] += sum_t A[c][t] x[2(i + (1 - c) + t - 1) + c]   (k3 s2 p1 conv: UNet:524-531; the data gradient of upsample + conv)
// In both, tap t of voxel v sits at halo coordinate v + s + t with a per-image shift s (p, resp. 1 - c): ONE unrolled 8-tap loop whose
// fragment-read lane bases are shifted per image.
template <int NCB, int MODE> struct Cfg;
template <> struct Cfg<1, 0> { static constexpr int GT = 9; };  // 3 groups of 9 taps: the ring holds a whole chunk
template <> struct Cfg<2, 0> { static constexpr int GT = 3; };  // 9 groups of 3 taps
template <> struct Cfg<1, 1> { static constexpr int GT = 4; };  // 2 groups of 4 taps (8 KB)
template <> struct Cfg<2, 1> { static constexpr int GT = 2; };  // 4 groups of 2 taps (8 KB)
template <> struct Cfg<1, 2> { static constexpr int GT = 4; };
template <> struct Cfg<2, 2> { static constexpr int GT = 2; };
template <> struct Cfg<1, 3> { static constexpr int GT = 9; };  // MODE 3: MODE 0 with a rolling halo (single-chunk layers; end of this file)
template <int MODE> constexpr bool kPhase = MODE == 1 || MODE == 2;
constexpr int RGROUP = 4 * HSLICE;          // MODE 3: a group of 4 d-slices = 25 whole 1-KiB DMA pieces
constexpr int RHALO = 3 * RGROUP + 13 * 1024;  // three groups + the copy of slot 0's first two slices (12.5 pieces, issued as 13)

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

template <int NCB, int MODE>
struct K {
  static constexpr int NT = kPhase<MODE> ? 8 : 27;  // taps per image
  static constexpr int GT = Cfg<NCB, MODE>::GT, NG = NT / GT;
  static_assert(NG * GT == NT && NG >= 2, "tap groups");
  static constexpr int FRAGS = GT * 2 * NCB, GROUP_BYTES = FRAGS * 1024, RING = RD * GROUP_BYTES;
  static constexpr int VOXP = NCB * 64;        // staging bytes per voxel (bf16), 16-byte slots XOR-swizzled with the voxel index
  static constexpr int PV = NCB * 4;           // 16-byte pieces per voxel
  static constexpr int STG_WAVE = 64 * VOXP;   // one compute wave's 64 voxels
  static constexpr int RING0 = MODE == 3 ? RHALO : 2 * HALO_BYTES, STG0 = RING0 + RING, AV0 = STG0 + 4 * STG_WAVE;
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
template <int Q, int T, int NCB, int FLIP, int MODE>
__device__ __forceinline__ void issue_one(Frags<NCB>& f, const unsigned (&bb)[2][3][2], unsigned ab) {
  constexpr int U = FLIP ? 26 - T : T;
  // MODE 0: tap U of the 3x3x3 window; MODE 1 / 2: tap T = (td, th, tw) of the 2x2x2 box (the image's shift is in the lane bases)
  constexpr int TH = kPhase<MODE> ? (T >> 1) & 1 : (U / 3) % 3;
  constexpr int BOFF = kPhase<MODE> ? (T >> 2) * HSLICE + TH * HROW + (T & 1) * 64 : (U / 9) * HSLICE + TH * HROW + (U % 3) * 64;
  constexpr int t = T % Cfg<NCB, MODE>::GT;
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
template <int T, int NCB, int FLIP, int MODE>
__device__ __forceinline__ void issue_frags(Frags<NCB>& f, const unsigned (&bb)[2][3][2], unsigned ab) {
  issue_one<0, T, NCB, FLIP, MODE>(f, bb, ab); issue_one<1, T, NCB, FLIP, MODE>(f, bb, ab); issue_one<2, T, NCB, FLIP, MODE>(f, bb, ab);
  issue_one<3, T, NCB, FLIP, MODE>(f, bb, ab); issue_one<4, T, NCB, FLIP, MODE>(f, bb, ab); issue_one<5, T, NCB, FLIP, MODE>(f, bb, ab);
  if constexpr (NCB == 2) { issue_one<6, T, NCB, FLIP, MODE>(f, bb, ab); issue_one<7, T, NCB, FLIP, MODE>(f, bb, ab); }
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
template <int KI, int TN, int NCB, int FLIP, int MODE>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[2][NCB], Frags<NCB>& cur, Frags<NCB>& nxt, const unsigned (&bb)[2][3][2], unsigned ab) {
  wait_operands<KI, NCB>(cur);
  if constexpr (NCB == 2) {
    constexpr int ks = KI / 4, vb = (KI / 2) % 2, cb = KI % 2;
    acc[vb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[ks][cb]), __builtin_bit_cast(bf16x8, cur.b[ks][vb]),
                                                          acc[vb][cb], 0, 0, 0);
    issue_one<KI, TN, NCB, FLIP, MODE>(nxt, bb, ab);
  } else {
    constexpr int ks = KI / 2, vb = KI % 2;
    acc[vb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[ks][0]), __builtin_bit_cast(bf16x8, cur.b[ks][vb]),
                                                         acc[vb][0], 0, 0, 0);
    if constexpr (KI == 0) { issue_one<0, TN, NCB, FLIP, MODE>(nxt, bb, ab); issue_one<1, TN, NCB, FLIP, MODE>(nxt, bb, ab); }
    else if constexpr (KI == 1) { issue_one<2, TN, NCB, FLIP, MODE>(nxt, bb, ab); issue_one<3, TN, NCB, FLIP, MODE>(nxt, bb, ab); }
    else if constexpr (KI == 2) issue_one<4, TN, NCB, FLIP, MODE>(nxt, bb, ab);
    else issue_one<5, TN, NCB, FLIP, MODE>(nxt, bb, ab);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// MFMAs of the tap held (or arriving) in `cur`, interleaved with the reads of tap TN into `nxt`
template <int TN, int NCB, int FLIP, int MODE>
__device__ __forceinline__ void tap_body(f32x16 (&acc)[2][NCB], Frags<NCB>& cur, Frags<NCB>& nxt, const unsigned (&bb)[2][3][2], unsigned ab) {
  mfma_step<0, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
  mfma_step<1, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
  mfma_step<2, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
  mfma_step<3, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
  if constexpr (NCB == 2) {
    mfma_step<4, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
    mfma_step<5, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
    mfma_step<6, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
    mfma_step<7, TN, NCB, FLIP, MODE>(acc, cur, nxt, bb, ab);
  }
}

// ------------------------------------------------------------------------------------------------ image sequence
// An "image" = one 32-channel chunk of one tile.  Every wave walks the same sequence (wave-uniform scalars).
// MODE 1 / 2: an image = one chunk of one phase (class) pc of one tile, walked chunk-fastest, then the 8 phases, then the next tile.
struct Seq {
  int tile, ch, n, d0, h0, w0;          // current image
  int ntile, nch, nn, nd0, nh0, nw0;    // next image (ntile < 0: none)
  TileWalk walk;                        // digits of the furthest tile looked at (the next one once seq_next has stepped)
  // MODE 1 / 2 only
  int pc, npc;                          // phase (class) of the current / next image: bit 2 = d, bit 1 = h, bit 0 = w
  int buf, nbuf;                        // halo buffer of the current / next image
  int nfetch;                           // the next image's halo has to be fetched (it is not what buffer nbuf holds already)
  int held0, held1;                     // key of the halo image each buffer holds (two scalars: a run-time indexed array would live in scratch)
};
// MODE 1 (scatter): the halo image belongs to (tile, chunk) and serves all 8 phases.  With one or two chunks both buffers keep their
// image for the whole tile (one chunk: the other buffer receives the next tile's; two chunks: buffer = chunk) and 8 phases share ONE
// fetch; with more chunks the buffers alternate and every image is fetched again (from L2).  MODE 2 (gather): every image has its
// own halo (the class's sub-lattice): the buffers alternate.
template <int MODE>
__device__ __forceinline__ void seq_next(Seq& q, const ConvArgs& a, int tile_step, int tile_last) {
  if (q.ch + 1 < a.nchunks) { q.ntile = q.tile; q.nch = q.ch + 1; q.nn = q.n; q.nd0 = q.d0; q.nh0 = q.h0; q.nw0 = q.w0; if (MODE) q.npc = q.pc; }
  else if (MODE != 0 && q.pc + 1 < 8) { q.ntile = q.tile; q.nch = 0; q.nn = q.n; q.nd0 = q.d0; q.nh0 = q.h0; q.nw0 = q.w0; q.npc = q.pc + 1; }
  else if (q.tile + tile_step < tile_last) { q.ntile = q.tile + tile_step; q.nch = 0; walk_step(q.walk, a.g); walk_origin(q.walk, a.g, q.nn, q.nd0, q.nh0, q.nw0); if (MODE) q.npc = 0; }
  else { q.ntile = -1; q.nch = 0; q.nn = q.nd0 = q.nh0 = q.nw0 = 0; if (MODE) q.npc = 0; }
  if constexpr (MODE != 0) {
    const bool newtile = q.ntile != q.tile;
    if (MODE == 1 && a.nchunks == 1) q.nbuf = newtile ? q.buf ^ 1 : q.buf;
    else if (MODE == 1 && a.nchunks == 2) q.nbuf = q.nch;
    else q.nbuf = q.buf ^ 1;
    const int key = (q.ntile * 8 + (MODE == 2 ? q.npc : 0)) * a.nchunks + q.nch;
    q.nfetch = q.ntile >= 0 && (q.nbuf ? q.held1 : q.held0) != key;
    if (q.nfetch) { if (q.nbuf) q.held1 = key; else q.held0 = key; }
  }
}
template <int MODE>
__device__ __forceinline__ void seq_init(Seq& q, const ConvArgs& a, int tile0, int tile_step) {
  q.tile = tile0; q.ch = 0;
  walk_init(q.walk, a.g, tile0, tile_step);
  walk_origin(q.walk, a.g, q.n, q.d0, q.h0, q.w0);
  if constexpr (MODE != 0) {
    q.pc = q.npc = 0; q.buf = q.nbuf = 0; q.nfetch = 0;
    q.held0 = (tile0 * 8) * a.nchunks; q.held1 = -1;  // the prologue fetches the first image into buffer 0
  }
}
template <int MODE>
__device__ __forceinline__ void seq_advance(Seq& q) {
  q.tile = q.ntile; q.ch = q.nch; q.n = q.nn; q.d0 = q.nd0; q.h0 = q.nh0; q.w0 = q.nw0;
  if constexpr (MODE != 0) { q.pc = q.npc; q.buf = q.nbuf; }
}
// an image that completes an accumulation: the tile's last chunk (MODE 0), a phase's last chunk (MODE 1), the last class's last chunk (MODE 2)
template <int MODE>
__device__ __forceinline__ bool seq_acc_done(const Seq& q, const ConvArgs& a) {
  return q.ch == a.nchunks - 1 && (MODE != 2 || q.pc == 7);
}

// ------------------------------------------------------------------------------------------------ compute waves
template <int NCB>
struct CState {
  f32x16 acc[2][NCB];
  Frags<NCB> fr[2];
  unsigned bcur[2][3][2];      // halo fragment-read lane bases inside the buffer of the image being consumed ([vb][tap row][k-step])
  unsigned bnxt[2][3][2];      // MODE 1 / 2: ... of the next image (its buffer and its shift)
  unsigned abase, abase_next;  // A fragment-read bases: ring slot of the current / the next group
  int av_n;                    // image index the LDS bias table of this wave was loaded for
  int cur;                     // halo buffer of the current image
  unsigned hdelta;             // MODE 0 / 3: byte step from the current image's halo to the next image's (set per image by the role)
  int slot;                    // ring slot of the previous group
  int resident;                // single chunk whose weight groups all fit the ring: loaded once, inner barriers skipped
  unsigned long long bw[4];    // MI_C27_DIAG_BAR
};

// MODE 1 / 2: lane bases of an image in halo buffer `buf` whose taps are shifted by s = (pc bit 2, bit 1, bit 0) voxels along (d, h, w):
// entry [vb][th][ks] serves the taps of box row th (0 / 1); the XOR swizzle goes by the halo ROW the read lands in (row + s_h + th)
template <int NCB>
__device__ __forceinline__ void phase_bases(unsigned (&bb)[2][3][2], int wave, int lane, int buf, int shift) {
  const int r = lane & 31, h = lane >> 5, row = r >> 3, col = r & 7;
  const int sd = (shift >> 2) & 1, sh = (shift >> 1) & 1, sw = shift & 1;
  const unsigned base = (unsigned)(buf * HALO_BYTES + (wave + sd) * HSLICE + sh * HROW + sw * 64);  // wave-uniform
#pragma unroll
  for (int vb = 0; vb < 2; ++vb)
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        bb[vb][th][ks] = base + (unsigned)((vb * 4 + row) * HROW + col * 64 + (((ks * 2 + h) ^ ((row + sh + th) & 3)) * 16));
#pragma unroll
  for (int vb = 0; vb < 2; ++vb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bb[vb][2][ks] = 0u;  // (row 2 does not exist in the 2x2x2 box: never read)
}
// shift of phase / class pc: MODE 1 (scatter) the phase itself, MODE 2 (gather) 1 - class per axis
template <int MODE>
__device__ __forceinline__ int phase_shift(int pc) { return MODE == 2 ? (pc ^ 7) : pc; }

template <int J, int NCB, int MODE>
__device__ __forceinline__ void compute_top(CState<NCB>& s, int lane) {
  using KK = K<NCB, MODE>;
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

template <int T, int PAR, int NCB, int FLIP, int MODE>
__device__ __forceinline__ void taps(CState<NCB>& s, int lane) {
  using KK = K<NCB, MODE>;
  if constexpr (T < KK::NT) {
    constexpr int cur = (T + PAR) & 1;
    if constexpr (T % KK::GT == 0) compute_top<T / KK::GT, NCB, MODE>(s, lane);
    if constexpr (T + 1 < KK::NT) {
#ifdef MI_C27_DIAG_TAPS8  // timing probe: what would an 8-tap (phase) convolution cost in this kernel's structure?  (taps 8..25 skipped)
      if constexpr (T < 8)
#endif
      tap_body<T + 1, NCB, FLIP, MODE>(s.acc, s.fr[cur], s.fr[cur ^ 1], s.bcur, (T + 1) % KK::GT == 0 ? s.abase_next : s.abase);
    } else if constexpr (!kPhase<MODE>) {  // next: first tap of the next image -- other halo buffer (MODE 3: next ring position), next group's slot
      const unsigned delta = s.hdelta;
#pragma unroll
      for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) s.bcur[vb][th][ks] += delta;
      tap_body<0, NCB, FLIP, MODE>(s.acc, s.fr[cur], s.fr[cur ^ 1], s.bcur, s.abase_next);
    } else {  // MODE 1 / 2: the next image's buffer and shift are in bnxt (compute_role)
#pragma unroll
      for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) s.bcur[vb][th][ks] = s.bnxt[vb][th][ks];
      tap_body<0, NCB, FLIP, MODE>(s.acc, s.fr[cur], s.fr[cur ^ 1], s.bcur, s.abase_next);
    }
    taps<T + 1, PAR, NCB, FLIP, MODE>(s, lane);
  }
}

// bias (+ time embedding) table of this wave in LDS: [cb][h][16] fp32; the accumulators start from it (the MFMA adds on top)
template <int NCB, int MODE>
__device__ __forceinline__ void load_av(CState<NCB>& s, const ConvArgs& a, char* lds, int y, int wave, int lane, int n) {
  using KK = K<NCB, MODE>;
  float* tab = (float*)(lds + KK::AV0 + wave * KK::AV_WAVE);
  for (int i = lane; i < NCB * 32; i += 64) {  // i = cb*32 + 16h + e  == channel offset inside this workgroup's cout range
    const int co = y * NCB * 32 + i;
    tab[i] = (a.addvec && co < a.Cout) ? a.addvec[(int64_t)n * a.addvec_stride + co] : 0.f;
  }
  s.av_n = n;
}
template <int NCB, int MODE>
__device__ __forceinline__ void init_acc(CState<NCB>& s, char* lds, int wave, int lane) {
  using KK = K<NCB, MODE>;
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
template <int NCB, int MODE>
__device__ __forceinline__ void stage_acc(CState<NCB>& s, char* lds, int wave, int lane) {
  using KK = K<NCB, MODE>;
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

template <int NCB, int FLIP, int MODE>
__device__ __forceinline__ void compute_role(const ConvArgs& a, char* lds, int y, int wave, int lane, int tile0, int tile_step, int tile_last) {
  using KK = K<NCB, MODE>;
  CState<NCB> s;
  if constexpr (MODE == 0) {
    const int r = lane & 31, h = lane >> 5, row = r >> 3, col = r & 7;
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s.bcur[vb][th][ks] = wave * HSLICE + (vb * 4 + row) * HROW + col * 64 + (((ks * 2 + h) ^ ((row + th) & 3)) * 16);
        }
  } else {
    phase_bases<NCB>(s.bcur, wave, lane, 0, phase_shift<MODE>(0));
  }
  Seq q;
  seq_init<MODE>(q, a, tile0, tile_step);
  load_av<NCB, MODE>(s, a, lds, y, wave, lane, q.n);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  init_acc<NCB, MODE>(s, lds, wave, lane);
  s.cur = 0;
  s.bw[0] = s.bw[1] = s.bw[2] = s.bw[3] = 0;
  s.resident = MODE == 0 && a.nchunks == 1 && KK::NG <= RD;
  s.slot = RD - 1;  // "slot of group -1": compute_top<0> steps to slot 0
  s.abase = s.abase_next = KK::RING0 + lane * 16;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing of this wave's is in flight across the barriers
  C27_BARRIER();  // prologue: first halo image and weight groups 0, 1 are in LDS
  issue_frags<0, NCB, FLIP, MODE>(s.fr[0], s.bcur, s.abase);
  wait_frags<NCB>(s.fr[0]);
  unsigned long long t0 = 0, r0 = 0;
  if (a.dbg & 64) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  int par = 0;
  while (true) {
    seq_next<MODE>(q, a, tile_step, tile_last);
    if constexpr (MODE != 0) phase_bases<NCB>(s.bnxt, wave, lane, q.nbuf, phase_shift<MODE>(q.npc));
    else s.hdelta = s.cur ? 0u - (unsigned)HALO_BYTES : (unsigned)HALO_BYTES;
    if constexpr ((KK::NT & 1) != 0) {  // odd tap count: the two fragment sets swap roles from image to image
      if (par) taps<0, 1, NCB, FLIP, MODE>(s, lane);
      else taps<0, 0, NCB, FLIP, MODE>(s, lane);
      par ^= 1;
    } else {
      taps<0, 0, NCB, FLIP, MODE>(s, lane);
    }
    if (seq_acc_done<MODE>(q, a)) {  // accumulation finished (the next image's first fragments are already on their way)
      if (s.resident) C27_BARRIER_T(s.bw[1]);  // the helper waves have read the previous tile out of the staging tile (helper_role)
      if (!(a.dbg & 1)) stage_acc<NCB, MODE>(s, lds, wave, lane);
      if (q.ntile >= 0 && q.nn != s.av_n) {  // image index changed: once in thousands of tiles
        load_av<NCB, MODE>(s, a, lds, y, wave, lane, q.nn);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      init_acc<NCB, MODE>(s, lds, wave, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging written before the barrier that hands it to the store waves
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q.ntile < 0) break;
    s.cur ^= 1;
    seq_advance<MODE>(q);
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
// MODE 1 / 2: the halo quarter is only there when the next image needs a fetch (Seq::nfetch), and with NG = 2 the last top follows
// top 0 directly: the halo is then younger than the weights it waits for, and only ONE store slot is younger than the halo.
constexpr int HPW = 10;  // halo pieces per helper wave and image

template <int NCB, int MODE>
__device__ __forceinline__ void issue_A(const ConvArgs& a, char* lds, int y, int hl, int lane, int pc, int ch, int j, int slot) {
  using KK = K<NCB, MODE>;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wpk, a.wpk_bytes);
  // fragments are packed [cout group][chunk][tap][ks][cb] and every chunk of a k3 s1 conv has all 27 taps: no table lookup
  // (a load inside the loop would be a VECTOR load -- the kernel stores to global memory -- and drain the DMA queue);
  // MODE 1 / 2: [cout group][phase][chunk][tap of the box][ks][cb]
  const int wfrag = kPhase<MODE> ? ((y * 8 + pc) * a.nchunks + ch) * (KK::NT * 2 * NCB) + j * KK::FRAGS : (y * a.nchunks + ch) * (27 * 2 * NCB) + j * KK::FRAGS;
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

// MODE 2 (gather): halo voxel (hd, hh, hw) of class c is input voxel 2 (origin + h) + c: the per-lane offsets hoff are built with
// doubled voxel steps (helper_role) and the class moves the scalar base; S = 2, (cd, chh, cw) = the class.  MODE 0 / 1: S = 1, class 0.
template <int K0, int K1, int MODE>  // pieces K0 .. K1-1 of this wave's HPW
__device__ __forceinline__ void issue_halo(const ConvArgs& a, char* lds, const int (&hp)[HPW], const unsigned (&hoff)[HPW], int hl, int buf, int valid,
                                           int n, int d0, int h0, int w0, int src_c0, int pc) {
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
#ifdef MI_C27_DIAG_HALO
  return;
#endif
  constexpr int S = MODE == 2 ? 2 : 1;
  const int cd = MODE == 2 ? (pc >> 2) & 1 : 0, chh = MODE == 2 ? (pc >> 1) & 1 : 0, cw = MODE == 2 ? pc & 1 : 0;
  const int od = S * (d0 - 1) + cd, oh = S * (h0 - 1) + chh, ow = S * (w0 - 1) + cw;  // input voxel of the halo origin
  // byte offset of the halo origin voxel, channel src_c0: wave-uniform, lives in an SGPR (mod 2^32; tensors < 4 GiB)
  const unsigned base = (unsigned)((((n * a.Di + od) * a.Hi + oh) * a.Wi + ow) * a.x_cs + src_c0) * 2u;
  // A helper wave shares its SIMD's vector issue with a compute wave that is issuing MFMAs: every VALU instruction here is paid for by
  // the whole workgroup at the next barrier (measured: with ~25 address instructions per piece the compute waves spent 16-20 % of
  // the kernel waiting for the helpers).  A tile whose halo lies inside the tensor needs none: the per-lane part of the address is a
  // kernel-lifetime constant (hoff) and the tile's part goes into the instruction's scalar offset.
  const bool interior = (valid != 0) & (od >= 0) & (od + S * 5 < a.Di) & (oh >= 0) & (oh + S * 9 < a.Hi) & (ow >= 0) & (ow + S * 9 < a.Wi) & (src_c0 + 32 <= a.Cin);
  if (interior) {
#pragma unroll
    for (int k = K0; k < K1; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + buf * HALO_BYTES + (hl + 4 * k) * 1024), 16, hoff[k], (int)base, 0, 0);
    return;
  }
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int pk = hp[k];
    const int gd = od + S * ((pk >> 16) & 255), gh = oh + S * ((pk >> 8) & 255), gw = ow + S * (pk & 255);
    const int c = src_c0 + ((pk >> 24) & 3) * 8;
    const bool ok = (valid != 0) & (pk >= 0) & ((unsigned)gd < (unsigned)a.Di) & ((unsigned)gh < (unsigned)a.Hi) & ((unsigned)gw < (unsigned)a.Wi) &
                    (c + 8 <= a.Cin);  // (bitwise: one select, no branches)
    const unsigned off = ok ? hoff[k] + base : 0xfffffff0u;  // out of range -> zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + buf * HALO_BYTES + (hl + 4 * k) * 1024), 16, off, 0, 0, 0);
  }
}

// Store epilogue of one tile, this wave's d-slice (= compute wave hl's staging tile).  Piece p (0 .. PV-1): 16 voxels x PV slots per
// wave-instruction, so that consecutive lanes cover consecutive 16-byte slots of a voxel and then the next voxel along W.
// MODE 1 (scatter): tile voxel v of phase p is output voxel 2 v + p: every voxel step of the lane constants and of the tile origin
// doubles and the phase moves the scalar base (SO = 2).
template <int NCB, int MODE>
struct Epi {
  int n, d0, h0, w0;        // the tile being stored
  int pc;                   // MODE 1: its phase
  int active;               // a tile is pending
  int full;                 // ... and lies inside the output with whole channel octets: no per-lane masks, addresses = lane constant + scalar
  unsigned ybase, rbase;    // byte offsets of the tile's origin voxel (channel y*NCB*32) in y / in the residual
  unsigned ylane, rlane, slane;  // this lane's constant part: voxel (lane / PV), slot (lane % PV) of a piece in y / residual / the staging tile
  unsigned yps, rps;        // bytes from one piece to the next (8 / PV rows of the tile)
  u32x4 res[K<NCB, MODE>::PV];    // residual pieces, loaded one tap group ahead of their use
  float sa[8], sq[8];       // a.stats: running sum / sum of squares of this lane's channel octet over the tiles of image sn
  int sn;
};
// (Statistics are compiled into the NCB = 1 variant only.  Measured same box, same call: with the code in both variants the C4 step
// takes 25.5 ms, 25.8 ms when it is compiled in but unused, 25.3 ms with it in neither variant or in NCB = 1 alone -- the NCB = 2
// kernel, 256 VGPRs and ~9000 instructions, loses more to the extra code than the fused statistics save.)
// Fold the lanes that share a channel octet (they sit PV apart) and store this wave's chunk of image e.sn; resets the sums.
template <int NCB, int MODE>
__device__ __forceinline__ void stats_flush(Epi<NCB, MODE>& e, const ConvArgs& a, int y, int hl, int lane) {
  using KK = K<NCB, MODE>;
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
template <int NCB, int MODE>
__device__ __forceinline__ void stats_zero(const ConvArgs& a, int y, int hl, int lane) {
  using KK = K<NCB, MODE>;
  const int chunk = blockIdx.x * 4 + hl;
  if (lane < KK::PV)
    for (int n = 0; n < a.N; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = y * NCB * 32 + lane * 8 + j;
        if (c < a.Cout) *(float2*)(a.stats + (((int64_t)n * a.Cout + c) * a.stats_chunks + chunk) * 2) = make_float2(0.f, 0.f);
      }
}
template <int NCB, int MODE, bool ST>
__device__ __forceinline__ void stats_begin_tile(Epi<NCB, MODE>& e, const ConvArgs& a, int y, int hl, int lane) {
  if (ST && a.stats && e.sn != e.n) {  // wave-uniform
    if (e.sn >= 0) stats_flush<NCB, MODE>(e, a, y, hl, lane);
    e.sn = e.n;
  }
}
// Per-lane border masks of piece p (tiles that stick out of the output only): false -> the lane's access is dropped
template <int NCB, int MODE>
__device__ __forceinline__ bool epi_inside(const Epi<NCB, MODE>& e, const ConvArgs& a, int y, int hl, int lane, int p) {
  using KK = K<NCB, MODE>;
  constexpr int SO = MODE == 1 ? 2 : 1;
  const int q = p * 64 + lane, v = q / KK::PV, sidx = q % KK::PV;
  const int od = SO * (e.d0 + hl) + (MODE == 1 ? (e.pc >> 2) & 1 : 0), oh = SO * (e.h0 + (v >> 3)) + (MODE == 1 ? (e.pc >> 1) & 1 : 0),
            ow = SO * (e.w0 + (v & 7)) + (MODE == 1 ? e.pc & 1 : 0);
  return (od < a.Do) & (oh < a.Ho) & (ow < a.Wo) & (y * NCB * 32 + sidx * 8 + 8 <= a.Cout);
}
template <int NCB, int MODE>
__device__ __forceinline__ void epi_begin_tile(Epi<NCB, MODE>& e, const ConvArgs& a, int y, int n, int d0, int h0, int w0, int pc) {
  constexpr int SO = MODE == 1 ? 2 : 1;
  e.active = 1; e.n = n; e.d0 = d0; e.h0 = h0; e.w0 = w0; e.pc = pc;
  const int pd = MODE == 1 ? (pc >> 2) & 1 : 0, ph = MODE == 1 ? (pc >> 1) & 1 : 0, pw = MODE == 1 ? pc & 1 : 0;
  const int od = SO * d0 + pd, oh = SO * h0 + ph, ow = SO * w0 + pw;  // output voxel of the tile origin
  e.full = (od + SO * 3 < a.Do) & (oh + SO * 7 < a.Ho) & (ow + SO * 7 < a.Wo) & ((y + 1) * NCB * 32 <= a.Cout);
  const unsigned vox = (unsigned)(((n * a.Do + od) * a.Ho + oh) * a.Wo + ow);
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
template <int NCB, int MODE>
__device__ __forceinline__ void epi_issue_res(Epi<NCB, MODE>& e, const ConvArgs& a, int y, int hl, int lane, bool enable) {
  using KK = K<NCB, MODE>;
  if (!enable) return;  // (a pending tracked load makes the compiler drain the DMA queue at its waits: none without a residual)
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res, a.res ? a.res_bytes : 0u);
#pragma unroll
  for (int p = 0; p < KK::PV; ++p) {
    const unsigned so = e.rbase + (unsigned)p * e.rps;  // scalar
    if (e.full) {
      e.res[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)e.rlane, (int)so, 0));
    } else {
      const unsigned off = epi_inside<NCB, MODE>(e, a, y, hl, lane, p) ? e.rlane + so : 0xfffffff0u;
      e.res[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)off, 0, 0));
    }
  }
}
// The helper waves share their SIMDs' vector issue with the compute waves' MFMAs, so VALU instructions here are the scarce thing:
// an inside tile (e.full) costs no address arithmetic at all (lane constants + a scalar offset) and no statistics masks.
template <int P0, int CNT, int NCB, int MODE, bool ST>
__device__ __forceinline__ void epi_process(Epi<NCB, MODE>& e, const ConvArgs& a, char* lds, int y, int hl, int lane, bool has_res) {
  using KK = K<NCB, MODE>;
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, a.y_bytes);
#pragma unroll
  for (int p = P0; p < P0 + CNT; ++p) {
    if (p >= KK::PV) {
      // The helper's vmcnt plan counts PPT stores per slot.  Where PV is not a multiple of NG - 1 (MODE 1 / 2 with 64 channels: 8 pieces
      // over 3 slots) the missing ones are issued with an out-of-range offset: dropped by the range check, counted like the others.
      if constexpr ((KK::NG - 1) * KK::PPT != KK::PV)
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{0, 0, 0, 0}, ry, (int)0xfffffff0u, 0, 0);
      continue;
    }
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
      const bool inside = epi_inside<NCB, MODE>(e, a, y, hl, lane, p);
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
template <int J, int NCB, int MODE, bool ST>
__device__ __forceinline__ void epi_slot(Epi<NCB, MODE>& e, const ConvArgs& a, char* lds, int y, int hl, int lane, int j, bool has_res) {
  using KK = K<NCB, MODE>;
  if constexpr (J < KK::NG) {
    if (j == J) epi_process<(J - 1) * KK::PPT, KK::PPT, NCB, MODE, ST>(e, a, lds, y, hl, lane, has_res);
    else epi_slot<J + 1, NCB, MODE, ST>(e, a, lds, y, hl, lane, j, has_res);
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
template <int J0, int NCB, int MODE>
__device__ __forceinline__ void halo_part(const ConvArgs& a, char* lds, const int (&hp)[HPW], const unsigned (&hoff)[HPW], int hl, int buf, const Seq& q, int j) {
  using KK = K<NCB, MODE>;
  if constexpr (J0 < KK::NH) {
    if (j == J0) {
#ifdef MI_C27_DIAG_HOT  // every halo image of this workgroup = the origin tile's: L2-warm after the first fetch
      issue_halo<KK::hbeg(J0), KK::hbeg(J0 + 1), MODE>(a, lds, hp, hoff, hl, buf, q.ntile >= 0, 0, 4, 8, 8, q.nch * 32, 0);
#else
      issue_halo<KK::hbeg(J0), KK::hbeg(J0 + 1), MODE>(a, lds, hp, hoff, hl, buf, q.ntile >= 0, q.nn, q.nd0, q.nh0, q.nw0, q.nch * 32, MODE ? q.npc : 0);
#endif
    } else halo_part<J0 + 1, NCB, MODE>(a, lds, hp, hoff, hl, buf, q, j);
  }
}

// ST: this instantiation carries the output-statistics code (forward kernel of the 32-channel variant only: see stats_flush)
template <int NCB, int MODE, bool ST>
__device__ __forceinline__ void helper_role(const ConvArgs& a, char* lds, int y, int hl, int lane, int tile0, int tile_step, int tile_last) {
  using KK = K<NCB, MODE>;
  constexpr int SI = MODE == 2 ? 2 : 1, SO = MODE == 1 ? 2 : 1;  // voxel step of the input (halo source) / of the output per tile voxel
  int hp[HPW];  // this lane's halo DMA pieces: (logical slot << 24) | (hd << 16) | (hh << 8) | hw, or -1
  unsigned hoff[HPW];  // ... and their byte offsets from the halo origin voxel
#pragma unroll
  for (int k = 0; k < HPW; ++k) {
    const int v = (hl + 4 * k) * 16 + (lane >> 2), p = lane & 3;
    const int hd = v / 100, rem = v - hd * 100, hh = rem / 10, hw = rem - hh * 10;
    hp[k] = v < HALO_VOX ? (((p ^ (hh & 3)) << 24) | (hd << 16) | (hh << 8) | hw) : -1;
    // (lanes past the image fetch the origin voxel into LDS padding nobody reads -- on the interior path; the border path masks them)
    hoff[k] = v < HALO_VOX ? (unsigned)(((SI * hd * a.Hi + SI * hh) * a.Wi + SI * hw) * a.x_cs + (p ^ (hh & 3)) * 8) * 2u : 0u;
  }
  const bool has_res = a.res != nullptr;
  const bool resident = MODE == 0 && a.nchunks == 1 && KK::NG <= RD;  // the ring holds every group of the only chunk: load once
  int a_pc = 0, a_ch = 0, a_j = 0, slot = 0, issued = 0;
  auto issue_next_A = [&]() {
    if (!(resident && issued >= KK::NG)) issue_A<NCB, MODE>(a, lds, y, hl, lane, a_pc, a_ch, a_j, slot);
    issued = issued < 1000 ? issued + 1 : issued;
    slot = slot + 1 == RD ? 0 : slot + 1;
    if (++a_j == KK::NG) {
      a_j = 0;
      if (++a_ch == a.nchunks) { a_ch = 0; if (MODE) a_pc = (a_pc + 1) & 7; }
    }
  };
  Seq q;
  seq_init<MODE>(q, a, tile0, tile_step);
  issue_halo<0, HPW, MODE>(a, lds, hp, hoff, hl, 0, 1, q.n, q.d0, q.h0, q.w0, 0, 0);
  issue_next_A();
  issue_next_A();  // two groups ahead
  if (resident)    // ... or all of them: nothing publishes a group later (compute_top skips the inner barriers)
    for (int g = 2; g < KK::NG; ++g) issue_next_A();
  wait_vm<0>();
  C27_BARRIER();  // prologue
  Epi<NCB, MODE> e;
  e.active = 0; e.full = 0; e.n = e.d0 = e.h0 = e.w0 = 0; e.pc = 0;
  e.ybase = e.rbase = 0;
  {
    const int vl = lane / KK::PV, sidx = lane % KK::PV, row = vl >> 3, col = vl & 7;  // (pieces start at whole rows: v & (PV-1) == vl & (PV-1))
    e.ylane = (unsigned)(((SO * hl * a.Ho + SO * row) * a.Wo + SO * col) * a.y_cs + sidx * 8) * 2u;
    e.rlane = (unsigned)(((hl * a.Ho + row) * a.Wo + col) * a.res_cs + sidx * 8) * 2u;
    e.slane = (unsigned)(vl * KK::VOXP + ((sidx ^ (vl & (KK::PV - 1))) * 16));
    e.yps = (unsigned)((8 / KK::PV) * SO * a.Wo * a.y_cs) * 2u;
    e.rps = (unsigned)((8 / KK::PV) * a.Wo * a.res_cs) * 2u;
  }
#pragma unroll
  for (int p = 0; p < KK::PV; ++p) e.res[p] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 8; ++j) e.sa[j] = e.sq[j] = 0.f;
  e.sn = -1;
  if (ST && a.stats) stats_zero<NCB, MODE>(a, y, hl, lane);
  int cur = 0;
  [[maybe_unused]] unsigned long long hbw[4] = {0, 0, 0, 0}, hseg[4] = {0, 0, 0, 0};  // MI_C27_DIAG_BAR
  while (true) {
    seq_next<MODE>(q, a, tile_step, tile_last);
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
      if (epi) stats_begin_tile<NCB, MODE, ST>(e, a, y, hl, lane);
      epi_issue_res<NCB, MODE>(e, a, y, hl, lane, epi && has_res);
      halo_part<0, NCB, MODE>(a, lds, hp, hoff, hl, cur ^ 1, q, 0);
      C27_T1(hseg[0]); }
      { C27_T0(); wait_vm<0>(); C27_T1(hseg[3]); }  // (the previous tile's stores are a whole image old)
      C27_BARRIER_T(hbw[3]);
      { C27_T0();
      if (epi) {
        epi_process<0, KK::PV, NCB, MODE, ST>(e, a, lds, y, hl, lane, has_res);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done
        e.active = 0;
      }
      C27_T1(hseg[2]); }
      C27_BARRIER_T(hbw[1]);
      if (q.ch == a.nchunks - 1 && !(a.dbg & 1)) epi_begin_tile<NCB, MODE>(e, a, y, q.n, q.d0, q.h0, q.w0, 0);
      if (q.ntile < 0) break;
      cur ^= 1;
      seq_advance<MODE>(q);
      continue;
    }
    const bool fetch = MODE ? q.nfetch != 0 : true;     // (MODE 0: issued even past the last image, masked: the count below stays exact)
    const int fbuf = MODE ? q.nbuf : cur ^ 1;
    const int hal = fetch ? HPW : 0;                    // halo pieces this wave issues at top 0
    // ---- top 0
    wait_vm<0>();  // group 1's weights are this wave's youngest operation
    C27_BARRIER_T(hbw[0]);
    { C27_T0();
    if (epi) stats_begin_tile<NCB, MODE, ST>(e, a, y, hl, lane);
    epi_issue_res<NCB, MODE>(e, a, y, hl, lane, epi && has_res);
    issue_next_A();
    if (fetch) halo_part<0, NCB, MODE>(a, lds, hp, hoff, hl, fbuf, q, 0);
    C27_T1(hseg[0]); }
    // ---- tops 1 .. NG-2.  Issue order inside a top: weights of group j+2, halo part j, stores of slot j (after the barrier);
    // at top j the weights issued at top j-1 must have landed, i.e. everything but the halo part and the stores of top j-1.
    for (int j = 1; j < KK::NG - 1; ++j) {
      const int hprev = MODE ? (j == 1 ? hal : 0) : KK::hbeg(j) - KK::hbeg(j - 1);  // (values of a small table: j is a loop counter)
      wait_vm_dyn(hprev + ((epi && j >= 2) ? KK::PPT : 0));
      C27_BARRIER_T(hbw[j == 1 ? 1 : 2]);
      issue_next_A();
      if constexpr (MODE == 0) halo_part<1, NCB, MODE>(a, lds, hp, hoff, hl, cur ^ 1, q, j);
      { C27_T0(); if (epi) epi_slot<1, NCB, MODE, ST>(e, a, lds, y, hl, lane, j, has_res); C27_T1(hseg[1]); }
    }
    // ---- top NG-1: the last store slot runs BEFORE the barrier (a single-chunk tile's compute waves overwrite the staging tile right
    // after it); the next image's halo and group NG's weights must have landed, the stores of the last two slots may fly
    { C27_T0();
    if (epi) {
      epi_process<(KK::NG - 2) * KK::PPT, KK::PPT, NCB, MODE, ST>(e, a, lds, y, hl, lane, has_res);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done
      e.active = 0;
    }
    C27_T1(hseg[2]); }
    { C27_T0();
    // younger than group NG's weights and the halo: the stores of the last two slots (NG = 2: there is only one slot, the one just issued)
    if (epi) wait_vm<(KK::NG >= 3 ? 2 : 1) * KK::PPT>(); else wait_vm<0>();
    C27_T1(hseg[3]); }
    C27_BARRIER_T(hbw[3]);
    issue_next_A();
    if (seq_acc_done<MODE>(q, a) && !(a.dbg & 1)) epi_begin_tile<NCB, MODE>(e, a, y, q.n, q.d0, q.h0, q.w0, MODE ? q.pc : 0);
    if (q.ntile < 0) break;
    cur ^= 1;
    seq_advance<MODE>(q);
  }
  wait_vm<0>();
  C27_BARRIER();  // final: the last tile's staging is complete
#ifdef MI_C27_DIAG_BAR
  if ((a.dbg & 64) && blockIdx.x == 0 && blockIdx.y == 0 && hl == 0 && lane == 0)
    for (int i = 0; i < 4; ++i) { g_c27_clk[8 + i] = hbw[i]; g_c27_clk[12 + i] = hseg[i]; }
#endif
  if (e.active) {
    stats_begin_tile<NCB, MODE, ST>(e, a, y, hl, lane);
    epi_issue_res<NCB, MODE>(e, a, y, hl, lane, has_res);
    wait_vm<0>();
    epi_process<0, KK::PV, NCB, MODE, ST>(e, a, lds, y, hl, lane, has_res);
  }
  if (ST && a.stats && e.sn >= 0) stats_flush<NCB, MODE>(e, a, y, hl, lane);
  wait_vm<0>();
}

// ------------------------------------------------------------------------------------------------ MODE 3: rolling halo along D
// Single-chunk 32-channel layers (weights resident).  A workgroup owns a contiguous run of tiles in d-fastest order and walks it column
// segment by column segment.  Inside a segment the halo lives in a ring of three GROUPS of 4 d-slices: the tile at depth d0 reads
// slices d0-1 .. d0+4 = its first group and half of its second; the next tile takes the second group as its first, so a step fetches
// ONE group (25 pieces) instead of the whole halo image (38): the helper waves' LDS-DMA issue is what the compute waves wait for at
// the barriers.  The six slices have to be contiguous for the fragment reads' immediates: behind the third group sits a
// copy of the first two slices of whatever slot 0 holds (written together with it), so a tile on slots (2, 0) reads straight on.
// A segment starts like the kernel does (both groups fetched, prologue barrier) and ends like it (final barrier, last tile stored).
struct RSeg {
  int t, tend;        // next tile of the run (d-fastest linear index), end of the run
  int n, th, tw, td;  // digits of tile t
};
__device__ __forceinline__ void rseg_init(RSeg& r, const Geom& g, int t0, int tend) {
  r.t = t0; r.tend = tend;
  r.td = t0 % g.tilesD; t0 /= g.tilesD;
  r.tw = t0 % g.tilesW; t0 /= g.tilesW;
  r.th = t0 % g.tilesH;
  r.n = t0 / g.tilesH;
}
__device__ __forceinline__ int rseg_len(const RSeg& r, const Geom& g) {  // tiles of the run left in the current column
  const int a = r.tend - r.t, b = g.tilesD - r.td;
  return a < b ? a : b;
}
__device__ __forceinline__ void rseg_advance(RSeg& r, const Geom& g, int len) {
  r.t += len; r.td += len;
  if (r.td >= g.tilesD) {
    r.td = 0;
    if (++r.tw == g.tilesW) { r.tw = 0; if (++r.th == g.tilesH) { r.th = 0; ++r.n; } }
  }
}

template <int FLIP>
__device__ __forceinline__ void compute_role_r(const ConvArgs& a, char* lds, int y, int wave, int lane, int t0, int tend) {
  using KK = K<1, 3>;
  CState<1> s;
  unsigned bl[2][3][2];
  {
    const int r = lane & 31, h = lane >> 5, row = r >> 3, col = r & 7;
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bl[vb][th][ks] = wave * HSLICE + (vb * 4 + row) * HROW + col * 64 + (((ks * 2 + h) ^ ((row + th) & 3)) * 16);
  }
  RSeg rs;
  rseg_init(rs, a.g, t0, tend);
  load_av<1, 3>(s, a, lds, y, wave, lane, rs.n);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  init_acc<1, 3>(s, lds, wave, lane);
  s.cur = 0; s.hdelta = 0u;
  s.bw[0] = s.bw[1] = s.bw[2] = s.bw[3] = 0;
  s.resident = 1;
  int par = 0;
  while (rs.t < rs.tend) {
    const int nseg = rseg_len(rs, a.g);
    if (rs.n != s.av_n) {  // image index changed (segment starts only: nothing of this wave's is in flight)
      load_av<1, 3>(s, a, lds, y, wave, lane, rs.n);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      init_acc<1, 3>(s, lds, wave, lane);
    }
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) s.bcur[vb][th][ks] = bl[vb][th][ks];  // the segment's first tile sits on group slots (0, 1)
    s.slot = RD - 1;
    s.abase = s.abase_next = KK::RING0 + lane * 16;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    C27_BARRIER();  // prologue of the segment: its first two groups (and, the first time, the weights) are in LDS
    if (par) { issue_frags<0, 1, FLIP, 3>(s.fr[1], s.bcur, s.abase); wait_frags<1>(s.fr[1]); }
    else { issue_frags<0, 1, FLIP, 3>(s.fr[0], s.bcur, s.abase); wait_frags<1>(s.fr[0]); }
    for (int u = 0; u < nseg; ++u) {
      const int c0 = u % 3, c1 = c0 == 2 ? 0 : c0 + 1;
      s.hdelta = (unsigned)((c1 - c0) * RGROUP);  // (past the segment's last tile: fragments that nobody consumes, inside the ring)
      if (par) taps<0, 1, 1, FLIP, 3>(s, lane);
      else taps<0, 0, 1, FLIP, 3>(s, lane);
      par ^= 1;
      C27_BARRIER_T(s.bw[1]);  // the helper waves have read the previous tile out of the staging tile
      if (!(a.dbg & 1)) stage_acc<1, 3>(s, lds, wave, lane);
      init_acc<1, 3>(s, lds, wave, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    C27_BARRIER();  // final of the segment: its last tile's staging is complete
    rseg_advance(rs, a.g, nseg);
  }
}

template <bool ST>
__device__ __forceinline__ void helper_role_r(const ConvArgs& a, char* lds, int y, int hl, int lane, int t0, int tend) {
  using KK = K<1, 3>;
  constexpr int GPW = 7;  // group pieces per helper wave: pieces hl, hl + 4, ... < 25
  int hp[GPW];
  unsigned hoff[GPW];
#pragma unroll
  for (int k = 0; k < GPW; ++k) {
    const int v = (hl + 4 * k) * 16 + (lane >> 2), p = lane & 3;  // < 400 for every piece < 25
    const int hd = v / 100, rem = v - hd * 100, hh = rem / 10, hw = rem - hh * 10;
    hp[k] = ((p ^ (hh & 3)) << 24) | (hd << 16) | (hh << 8) | hw;
    hoff[k] = (unsigned)(((hd * a.Hi + hh) * a.Wi + hw) * a.x_cs + (p ^ (hh & 3)) * 8) * 2u;
  }
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  // slices dfirst .. dfirst+3 of column (n, h0, w0) -> group slot; slot 0 also feeds the copy behind the third group
  auto group = [&](int slot, int n, int dfirst, int h0, int w0) {
#ifdef MI_C27_DIAG_HALO
    return;
#endif
    const int oh = h0 - 1, ow = w0 - 1;
    const unsigned base = (unsigned)((((n * a.Di + dfirst) * a.Hi + oh) * a.Wi + ow) * a.x_cs) * 2u;
    const bool interior = (dfirst >= 0) & (dfirst + 3 < a.Di) & (oh >= 0) & (oh + 9 < a.Hi) & (ow >= 0) & (ow + 9 < a.Wi) & (32 <= a.Cin);
#pragma unroll
    for (int k = 0; k < GPW; ++k) {
      const int i = hl + 4 * k;
      if (i >= 25) break;  // wave-uniform
      unsigned off = hoff[k] + base;
      if (!interior) {
        const int pk = hp[k];
        const int gd = dfirst + ((pk >> 16) & 255), gh = oh + ((pk >> 8) & 255), gw = ow + (pk & 255);
        const bool ok = ((unsigned)gd < (unsigned)a.Di) & ((unsigned)gh < (unsigned)a.Hi) & ((unsigned)gw < (unsigned)a.Wi) & (((pk >> 24) & 3) * 8 + 8 <= a.Cin);
        off = ok ? off : 0xfffffff0u;  // out of range -> zeros
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + slot * RGROUP + i * 1024), 16, off, 0, 0, 0);
      if (slot == 0 && i < 13) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + 3 * RGROUP + i * 1024), 16, off, 0, 0, 0);
    }
  };
  const bool has_res = a.res != nullptr;
  Epi<1, 3> e;
  e.active = 0; e.full = 0; e.n = e.d0 = e.h0 = e.w0 = 0; e.pc = 0;
  e.ybase = e.rbase = 0;
  {
    const int vl = lane / KK::PV, sidx = lane % KK::PV, row = vl >> 3, col = vl & 7;
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
  if (ST && a.stats) stats_zero<1, 3>(a, y, hl, lane);
  RSeg rs;
  rseg_init(rs, a.g, t0, tend);
  bool first = true;
  while (rs.t < rs.tend) {
    const int nseg = rseg_len(rs, a.g);
    const int n = rs.n, h0 = rs.th * 8, w0 = rs.tw * 8;
    group(0, n, rs.td * 4 - 1, h0, w0);
    group(1, n, rs.td * 4 + 3, h0, w0);
    if (first) {  // the weights: every tap group of the only chunk, once
#pragma unroll
      for (int j = 0; j < KK::NG; ++j) issue_A<1, 3>(a, lds, y, hl, lane, 0, 0, j, j);
      first = false;
    }
    wait_vm<0>();
    C27_BARRIER();  // prologue of the segment
    for (int u = 0; u < nseg; ++u) {
      const int d0 = (rs.td + u) * 4;
      const bool epi = e.active != 0;
      C27_BARRIER();
      if (epi) stats_begin_tile<1, 3, ST>(e, a, y, hl, lane);
      epi_issue_res<1, 3>(e, a, y, hl, lane, epi && has_res);
      if (u + 1 < nseg) group((u + 2) % 3, n, d0 + 7, h0, w0);  // the next tile's second group: slices (d0 + 4) + 3 ..
      wait_vm<0>();
      C27_BARRIER();
      if (epi) {
        epi_process<0, KK::PV, 1, 3, ST>(e, a, lds, y, hl, lane, has_res);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done
        e.active = 0;
      }
      C27_BARRIER();
      if (!(a.dbg & 1)) epi_begin_tile<1, 3>(e, a, y, n, d0, h0, w0, 0);
    }
    wait_vm<0>();
    C27_BARRIER();  // final of the segment
    if (e.active) {  // its last tile (the compute waves are at the next segment's prologue barrier, or done)
      stats_begin_tile<1, 3, ST>(e, a, y, hl, lane);
      epi_issue_res<1, 3>(e, a, y, hl, lane, has_res);
      wait_vm<0>();
      epi_process<0, KK::PV, 1, 3, ST>(e, a, lds, y, hl, lane, has_res);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      e.active = 0;
    }
    rseg_advance(rs, a.g, nseg);
  }
  if (ST && a.stats && e.sn >= 0) stats_flush<1, 3>(e, a, y, hl, lane);
  wait_vm<0>();
}

template <int FLIP>
__device__ __forceinline__ void conv27r_body(const ConvArgs& a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int y = blockIdx.y;
  // run of this workgroup: the workgroups of one XCD class (blockIdx.x % 8) get neighbouring runs
  const int nwg = gridDim.x, run = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int t0 = (int)((int64_t)run * a.ntiles / nwg), tend = (int)((int64_t)(run + 1) * a.ntiles / nwg);
  if (t0 >= tend) {  // whole workgroup, before any barrier
    if (FLIP == 0 && a.stats && wave >= 4) stats_zero<1, 3>(a, y, wave - 4, lane);
    return;
  }
  if (wave < 4) compute_role_r<FLIP>(a, lds, y, wave, lane, t0, tend);
  else helper_role_r<FLIP == 0>(a, lds, y, wave - 4, lane, t0, tend);
}

template <int NCB, int FLIP, int MODE>
__device__ __forceinline__ void conv27_body(const ConvArgs& a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int y = blockIdx.y;
  int tile_last, tile_step;
  const int tile0 = first_tile(a.ntiles, tile_last, tile_step);
  if (tile0 >= tile_last) {  // whole workgroup, before any barrier
    if (MODE == 0 && NCB == 1 && FLIP == 0 && a.stats && wave >= 4) stats_zero<NCB, MODE>(a, y, wave - 4, lane);
    return;
  }
  if (a.dbg & 2) { if (wave < 4) __builtin_amdgcn_s_setprio(3); }   // experiment knobs (MI_C27_DBG): static wave priority
  if (a.dbg & 4) { if (wave >= 4) __builtin_amdgcn_s_setprio(3); }
  if (wave < 4) compute_role<NCB, FLIP, MODE>(a, lds, y, wave, lane, tile0, tile_step, tile_last);
  else helper_role<NCB, MODE, MODE == 0 && NCB == 1 && FLIP == 0>(a, lds, y, wave - 4, lane, tile0, tile_step, tile_last);
}

}  // namespace
