// 3x3x3 stride-1 convolution (forward and data gradient) for gfx950: one persistent workgroup per CU, both MFMA operands
// served from LDS that is filled by LDS-DMA (buffer_load ... lds) -- no staging registers, no per-wave weight fetches.
// (Same call sites as conv.hip: every k3 s1 p1 `Convolution` of UNet:630-672,557-565,1935 / AEKL:158-187.)
//
// Why a second kernel.  In the table-driven kernel (conv.hip) every wave fetches its own copy of every weight fragment
// from L2 (0.5 KiB per MFMA: at full MFMA rate that alone is the CU's whole 64 B/clk vector-memory path), the halo image
// goes through 40 VGPRs + ds_write, and the epilogue stores 8 bytes per lane at a voxel stride (16 valid bytes per
// 128-byte line per instruction).  Here:
//   * A (weights): packed fragments are 1 KiB each and lane-linear, i.e. exactly one LDS-DMA wave-instruction.  The four
//     waves share them through an LDS ring of tap groups; a group is requested P groups ahead and is checked (counted
//     vmcnt + s_barrier) ONE group before its first use, so the fragment prefetch runs across group boundaries.
//     Single-chunk 32-channel layers keep all 27 taps resident (nothing is re-fetched after the first tile).
//   * B (activations): the halo image 6x10x10 voxels x 32 channels is dense (64-byte voxels, 38.4 KB) and double
//     buffered; it is filled by LDS-DMA with the hardware range check supplying the zero padding.  Bank conflicts of the
//     4x8-voxel ds_read_b128 fragment reads are removed by an XOR swizzle of the 16-byte slot with the halo row
//     (slot ^= hh & 3), applied on the DMA's per-lane SOURCE address (the LDS destination of a DMA is lane-linear).
//   * Epilogue: accumulators -> fp32 staging tile in LDS (wave-private) -> read back as 8 channels per lane so that four
//     consecutive lanes cover a voxel's 64 bytes and 32 lanes cover 8 voxels along W: full-line 16-byte stores; bias /
//     time-embedding and the residual are added in fp32 on the way (one rounding, residual read with the same full lines).
//   * Output channels are assigned to MFMA rows so that a lane's 16 accumulator registers are 16 CONSECUTIVE channels
//     (row (e&3) + 8(e>>2) + 4h <-> channel 16h + e; the permutation is applied when the weights are packed).
// All vector-memory operations of the main loop are issued in program order and waited for with counted vmcnt: a wait
// constant may only be too SMALL (stricter) when other operations (epilogue loads / stores) are in flight, never too large.
#include <stdlib.h>

#include "conv_common.h"
#include "medimgen_hip.h"

namespace {

constexpr int HROW = 640, HSLICE = 6400;   // dense halo image: 10 voxels x 64 B per row, 10 rows per slice, 6 slices
constexpr int HALO_VOX = 600;
constexpr int HALO_BYTES = 40960;          // image (38400 B) rounded up to 40 whole 1-KiB DMA pieces
constexpr int HP = 10;                     // halo pieces per wave and image
constexpr int EPI_PITCH = 144;             // staging: 32 fp32 channels per voxel + 16 B (conflict-free b128 writes)
constexpr int EPI_WAVE = 32 * EPI_PITCH;   // one 32-voxel block per pass
constexpr int RES_OPS = 8;                 // residual prefetch: at most VB * NCB * 2 loads per lane

template <int NCB> struct Cfg;
template <> struct Cfg<1> { static constexpr int GT = 9, P = 2; };  // 3 groups of 9 taps, whole chunk resident in the ring
template <> struct Cfg<2> { static constexpr int GT = 3, P = 3; };  // 9 groups of 3 taps, ring of 4

typedef __attribute__((address_space(3))) void lds_void;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int OFF>
__device__ __forceinline__ void lds_read16(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
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
  constexpr int HALF = 2 + NCB;      // reads per k-step
  constexpr int ks = Q / HALF, q = Q % HALF;
  if constexpr (NCB == 2) {
    if constexpr (q == 0) lds_read16<BOFF>(f.b[ks][0], bb[0][TH][ks]);
    else if constexpr (q == 1) lds_read16<((t * 2 + ks) * 2 + 0) * 1024>(f.a[ks][0], ab);
    else if constexpr (q == 2) lds_read16<((t * 2 + ks) * 2 + 1) * 1024>(f.a[ks][1], ab);
    else lds_read16<BOFF>(f.b[ks][1], bb[1][TH][ks]);
  } else {
    if constexpr (q == 0) lds_read16<BOFF>(f.b[ks][0], bb[0][TH][ks]);
    else if constexpr (q == 1) lds_read16<(t * 2 + ks) * 1024>(f.a[ks][0], ab);
    else lds_read16<BOFF>(f.b[ks][1], bb[1][TH][ks]);
  }
}
template <int T, int NCB, int FLIP>
__device__ __forceinline__ void issue_frags(Frags<NCB>& f, const unsigned (&bb)[2][3][2], unsigned ab) {
  issue_one<0, T, NCB, FLIP>(f, bb, ab); issue_one<1, T, NCB, FLIP>(f, bb, ab); issue_one<2, T, NCB, FLIP>(f, bb, ab);
  issue_one<3, T, NCB, FLIP>(f, bb, ab); issue_one<4, T, NCB, FLIP>(f, bb, ab); issue_one<5, T, NCB, FLIP>(f, bb, ab);
  if constexpr (NCB == 2) { issue_one<6, T, NCB, FLIP>(f, bb, ab); issue_one<7, T, NCB, FLIP>(f, bb, ab); }
}

// One tap = 4*NCB MFMAs, each followed by its share of the NEXT tap's fragment reads (1 per MFMA; 2,2,1,1 for NCB = 1): the LDS
// sees a steady trickle instead of a burst, and with one wave per SIMD the reads' latency hides under the following MFMAs.
// LDS returns in order, so before MFMA k a COUNTED lgkmcnt suffices: (reads of this tap not needed yet) + (reads of the next tap
// already issued).  The wait names the MFMA's operands ("+v") and a sched_barrier pins the order (cdna guide 5.7 (ii), rule 18).
template <int K, int NCB>
__device__ __forceinline__ void wait_operands(Frags<NCB>& f) {
  if constexpr (NCB == 2) {
    constexpr int ks = K / 4, vb = (K / 2) % 2, cb = K % 2;
    constexpr int N = (K == 3 || K == 7) ? 7 : 6;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a[ks][cb]), "+v"(f.b[ks][vb]) : "i"(N));
  } else {
    constexpr int ks = K / 2, vb = K % 2;
    constexpr int N = K == 0 ? 4 : 5;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a[ks][0]), "+v"(f.b[ks][vb]) : "i"(N));
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int K, int TN, int NCB, int FLIP>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[2][NCB], Frags<NCB>& cur, Frags<NCB>& nxt, const unsigned (&bb)[2][3][2], unsigned ab) {
  wait_operands<K, NCB>(cur);
  if constexpr (NCB == 2) {
    constexpr int ks = K / 4, vb = (K / 2) % 2, cb = K % 2;
    acc[vb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[ks][cb]), __builtin_bit_cast(bf16x8, cur.b[ks][vb]),
                                                            acc[vb][cb], 0, 0, 0);
    issue_one<K, TN, NCB, FLIP>(nxt, bb, ab);
  } else {
    constexpr int ks = K / 2, vb = K % 2;
    acc[vb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[ks][0]), __builtin_bit_cast(bf16x8, cur.b[ks][vb]),
                                                           acc[vb][0], 0, 0, 0);
    if constexpr (K == 0) { issue_one<0, TN, NCB, FLIP>(nxt, bb, ab); issue_one<1, TN, NCB, FLIP>(nxt, bb, ab); }
    else if constexpr (K == 1) { issue_one<2, TN, NCB, FLIP>(nxt, bb, ab); issue_one<3, TN, NCB, FLIP>(nxt, bb, ab); }
    else if constexpr (K == 2) issue_one<4, TN, NCB, FLIP>(nxt, bb, ab);
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

template <int NCB>
struct State {
  f32x16 acc[2][NCB];
  Frags<NCB> fr[2];
  unsigned lanebase[2][3][2];  // halo fragment-read bases relative to a buffer
  unsigned bcur[2][3][2];      // ... inside the buffer of the image being consumed
  unsigned abase, abase_next;  // A fragment-read bases: ring slot of the current / the next group
  int hp[HP];                  // this lane's halo DMA pieces: (slot << 24) | (hd << 16) | (hh << 8) | hw, or -1
  // wave-uniform
  int cur;                     // halo buffer of the current image
  int slot;                    // ring slot of the current group
  int a_ch, a_j;               // stream position (chunk, group) of the next weight group to request
  int tile, ch, n, d0, h0, w0; // current image
  int ntile, nch, nn, nd0, nh0, nw0;  // next image (ntile < 0: none)
  int res_pending;             // residual prefetch in flight (changes the vmcnt constants of the last groups)
  int st_pending;              // the previous tile's output stores (a fixed number) may still be in flight at the first group tops
  int av_n;                    // image index the bias / time-embedding registers were loaded for
  float av[NCB][8];            // addvec of this lane's 8 read-back channels per cout block
  u32x4 resv[2][NCB][2];       // prefetched residual pieces [vb][cb][i]
};

template <int NCB>
struct K {
  using C = Cfg<NCB>;
  static constexpr int GT = C::GT, P = C::P, NG = 27 / GT, RD = P + 1;
  static constexpr int FRAGS = GT * 2 * NCB, DA = (FRAGS + 3) / 4, GROUP_BYTES = FRAGS * 1024, RING = RD * GROUP_BYTES;
  static constexpr int RING0 = 2 * HALO_BYTES, EPI0 = RING0 + RING, LDS_TOTAL = EPI0 + 4 * EPI_WAVE;
};

// one weight group -> ring slot `slot` (this wave's share: fragments wave, wave + 4, ...; the tail repeats the last fragment so
// that every wave issues exactly DA operations)
template <int NCB>
__device__ __forceinline__ void issue_A(const ConvArgs& a, char* lds, int y, int wave, int lane, int ch, int j, int slot) {
  using KK = K<NCB>;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wpk, a.wpk_bytes);
  // fragments are packed [cout group][chunk][tap][ks][cb] and every chunk of a k3 s1 conv has all 27 taps: no table lookup
  // (a load inside the loop would be a VECTOR load -- the kernel stores to global memory -- and drain the DMA queue)
  const int wfrag = (y * a.nchunks + ch) * (27 * 2 * NCB) + j * KK::FRAGS;
#pragma unroll
  for (int i = 0; i < KK::DA; ++i) {
    int f = wave + 4 * i;
    f = f < KK::FRAGS ? f : KK::FRAGS - 1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + KK::RING0 + slot * KK::GROUP_BYTES + f * 1024), 16, lane * 16,
                                             (wfrag + f) * 1024, 0, 0);
  }
}

// halo image of (n, d0, h0, w0), channels [src_c0, src_c0 + 32) -> buffer `buf`; valid == 0 zero-fills (keeps the count exact)
template <int NCB>
__device__ __forceinline__ void issue_halo(const ConvArgs& a, char* lds, const int (&hp)[HP], int wave, int buf, int valid, int n, int d0,
                                           int h0, int w0, int src_c0) {
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
#pragma unroll
  for (int k = 0; k < HP; ++k) {
    const int pk = hp[k];
    const int gd = d0 - 1 + ((pk >> 16) & 255), gh = h0 - 1 + ((pk >> 8) & 255), gw = w0 - 1 + (pk & 255);
    const int c = src_c0 + ((pk >> 24) & 3) * 8;
    const bool ok = (valid != 0) & (pk >= 0) & ((unsigned)gd < (unsigned)a.Di) & ((unsigned)gh < (unsigned)a.Hi) & ((unsigned)gw < (unsigned)a.Wi) &
                    (c + 8 <= a.Cin);  // (bitwise: one select, no branches)
    const unsigned off = ok ? (unsigned)((((n * a.Di + gd) * a.Hi + gh) * a.Wi + gw) * a.x_cs + c) * 2u : 0xfffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + buf * HALO_BYTES + (wave + 4 * k) * 1024), 16, off, 0, 0, 0);
  }
}

template <int NCB>
__device__ __forceinline__ void mfmas(f32x16 (&acc)[2][NCB], const Frags<NCB>& f) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        acc[vb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[ks][cb]), __builtin_bit_cast(bf16x8, f.b[ks][vb]),
                                                              acc[vb][cb], 0, 0, 0);
}

// Top of group J of the current image: the NEXT group's weights (and, at the last group, the next image's halo) must have
// landed in every wave's view; then the slot of the previous group is refilled P groups ahead.
template <int J, int NCB>
__device__ __forceinline__ void group_top(State<NCB>& s, const ConvArgs& a, char* lds, int y, int wave, int lane) {
  using KK = K<NCB>;
  constexpr int X = (KK::P - 2) * KK::DA + ((J >= 1 && J <= KK::P - 1) ? HP : 0);
  constexpr int ST_OPS = 4 * NCB;  // output stores per lane and tile (epilogue, vector path)
  if (!(a.dbg & 2)) {
    if (s.res_pending) wait_vm<X + RES_OPS>();
    else if (J <= KK::P - 2 && s.st_pending) wait_vm<X + ST_OPS>();  // DMA_A(J+1) is older than those stores: they may stay in flight
    else wait_vm<X>();
  }
  if constexpr (J == KK::P - 2) s.st_pending = 0;
  if (!(a.dbg & 8)) __builtin_amdgcn_s_barrier();
  // on entry s.slot is the ring slot of group J-1: every wave is done with it, and (ring depth = P + 1) it is the slot of group J+P
  if (!(a.dbg & 32)) issue_A<NCB>(a, lds, y, wave, lane, s.a_ch, s.a_j, s.slot);
  if (++s.a_j == KK::NG) { s.a_j = 0; s.a_ch = s.a_ch + 1 == a.nchunks ? 0 : s.a_ch + 1; }
  if constexpr (J == 0) {
    if (!(a.dbg & 16)) issue_halo<NCB>(a, lds, s.hp, wave, s.cur ^ 1, s.ntile >= 0, s.nn, s.nd0, s.nh0, s.nw0, s.nch * 32);
  }
  const int sj = s.slot + 1 == KK::RD ? 0 : s.slot + 1;  // slot of group J
  const int sj1 = sj + 1 == KK::RD ? 0 : sj + 1;         // slot of group J+1 (checked by THIS top: its fragments may be prefetched)
  s.abase = KK::RING0 + sj * KK::GROUP_BYTES + lane * 16;
  s.abase_next = KK::RING0 + sj1 * KK::GROUP_BYTES + lane * 16;
  s.slot = sj;
}

// residual of the current tile in the read-back layout (8 channels = 16 bytes per lane: full lines), prefetched during the last
// chunk; consumed by the epilogue
template <int NCB>
__device__ __forceinline__ void issue_res(State<NCB>& s, const ConvArgs& a, int y, int wave, int lane) {
  const int piece = lane & 3;
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res, a.res_bytes);
#pragma unroll
  for (int vb = 0; vb < 2; ++vb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = i * 16 + (lane >> 2);
        const int od = s.d0 + wave, oh = s.h0 + vb * 4 + (v >> 3), ow = s.w0 + (v & 7);
        const int co = (y * NCB + cb) * 32 + piece * 8;
        const bool ok = (od < a.Do) & (oh < a.Ho) & (ow < a.Wo) & (co + 8 <= a.Cout);
        const unsigned off = ok ? (unsigned)(((((int64_t)(s.n * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.res_cs + co) * 2) : 0xfffffff0u;
        const u32x4 z = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)off, 0, 0));  // out of range -> 0
        s.resv[vb][cb][i] = z;
      }
}

template <int T, int PAR, int NCB, int FLIP>
__device__ __forceinline__ void taps(State<NCB>& s, const ConvArgs& a, char* lds, int y, int wave, int lane, bool fast_res) {
  using KK = K<NCB>;
  if constexpr (T < 27) {
    constexpr int cur = (T + PAR) & 1;
    if constexpr (T % KK::GT == 0) group_top<T / KK::GT, NCB>(s, a, lds, y, wave, lane);
    if constexpr (T == 18) {
      if (fast_res && s.ch == a.nchunks - 1) {  // wave-uniform
        issue_res<NCB>(s, a, y, wave, lane);
        s.res_pending = (KK::GT == 3);  // (GT = 9: no group top follows inside this image); RES_OPS == 2 * 2 * 2 loads
      }
    }
    if constexpr (T + 1 < 27) {
      tap_body<T + 1, NCB, FLIP>(s.acc, s.fr[cur], s.fr[cur ^ 1], s.bcur, (T + 1) % KK::GT == 0 ? s.abase_next : s.abase);
    } else {  // next: first tap of the next image -- other halo buffer, next group's slot
      const unsigned boff = (s.cur ^ 1) * HALO_BYTES;
#pragma unroll
      for (int vb = 0; vb < 2; ++vb)
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) s.bcur[vb][th][ks] = s.lanebase[vb][th][ks] + boff;
      tap_body<0, NCB, FLIP>(s.acc, s.fr[cur], s.fr[cur ^ 1], s.bcur, s.abase_next);
    }
    taps<T + 1, PAR, NCB, FLIP>(s, a, lds, y, wave, lane, fast_res);
  }
}

template <int NCB>
__device__ __forceinline__ void epilogue(State<NCB>& s, const ConvArgs& a, char* lds, int y, int wave, int lane, bool fast_res) {
  using KK = K<NCB>;
  const int r = lane & 31, h = lane >> 5, piece = lane & 3;
  char* stg = lds + KK::EPI0 + wave * EPI_WAVE;
  const bool vec_ok = (a.y_cs & 7) == 0 && (a.Cout & 7) == 0 && a.y_bytes != 0;  // wave-uniform
  if (a.addvec && s.av_n != s.n) {  // wave-uniform; the image index changes once in thousands of tiles
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = (y * NCB + cb) * 32 + piece * 8 + j;
        s.av[cb][j] = co < a.Cout ? a.addvec[(int64_t)s.n * a.addvec_stride + co] : 0.f;
      }
    s.av_n = s.n;
  }
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, a.y_bytes);
#pragma unroll
  for (int vb = 0; vb < 2; ++vb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      float* wp = (float*)(stg + r * EPI_PITCH + h * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {s.acc[vb][cb][4 * q], s.acc[vb][cb][4 * q + 1], s.acc[vb][cb][4 * q + 2], s.acc[vb][cb][4 * q + 3]};
        *(f32x4*)(wp + 4 * q) = v;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) s.acc[vb][cb][e] = 0.f;
      const int co = (y * NCB + cb) * 32 + piece * 8;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = i * 16 + (lane >> 2);
        const float* rp = (const float*)(stg + v * EPI_PITCH + piece * 32);
        f32x4 lo = *(const f32x4*)rp, hi = *(const f32x4*)(rp + 4);
        F8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f.v[j] = lo[j] + s.av[cb][j]; f.v[4 + j] = hi[j] + s.av[cb][4 + j]; }
        const int od = s.d0 + wave, oh = s.h0 + vb * 4 + (v >> 3), ow = s.w0 + (v & 7);
        const bool inside = (od < a.Do) & (oh < a.Ho) & (ow < a.Wo) & (co < a.Cout);
        const int64_t vox = ((int64_t)(s.n * a.Do + od) * a.Ho + oh) * a.Wo + ow;
        if (a.res) {
          if (fast_res) {
            F8 rr = unpack8(s.resv[vb][cb][i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) f.v[j] += rr.v[j];
          } else if (inside) {
            const bf16* rq = a.res + vox * a.res_cs + co;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (co + j < a.Cout) f.v[j] += bf2f(rq[j]);
          }
        }
        if (vec_ok) {  // always issued (masked lanes get an out-of-range offset): the store count is part of the vmcnt bookkeeping
          const unsigned off = inside ? (unsigned)((vox * a.y_cs + co) * 2) : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pack8(f)), ry, (int)off, 0, 0);
        } else if (inside) {
          bf16* yp = a.y + vox * a.y_cs + co;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (co + j < a.Cout) yp[j] = f2bf(f.v[j]);
        }
      }
    }
  s.res_pending = 0;
  s.st_pending = vec_ok;
}

template <int NCB, int FLIP>
__global__ void __launch_bounds__(256, 1) k_conv27(ConvArgs a) {
  using KK = K<NCB>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int y = blockIdx.y;
  int tile_last, tile_step;
  const int tile0 = first_tile(a.ntiles, tile_last, tile_step);
  if (tile0 >= tile_last) return;  // whole workgroup, before any barrier
  const Geom& g = a.g;
  const bool fast_res = a.res != nullptr && (a.res_cs & 7) == 0 && a.res_bytes != 0;  // wave-uniform

  State<NCB> s;
  {  // lane constants
    const int r = lane & 31, h = lane >> 5, row = r >> 3, col = r & 7;
#pragma unroll
    for (int vb = 0; vb < 2; ++vb)
#pragma unroll
      for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s.lanebase[vb][th][ks] = wave * HSLICE + (vb * 4 + row) * HROW + col * 64 + (((ks * 2 + h) ^ ((row + th) & 3)) * 16);
          s.bcur[vb][th][ks] = s.lanebase[vb][th][ks];
        }
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const int v = (wave + 4 * k) * 16 + (lane >> 2), p = lane & 3;
      const int hd = v / 100, rem = v - hd * 100, hh = rem / 10, hw = rem - hh * 10;
      s.hp[k] = v < HALO_VOX ? (((p ^ (hh & 3)) << 24) | (hd << 16) | (hh << 8) | hw) : -1;
    }
  }
#pragma unroll
  for (int vb = 0; vb < 2; ++vb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) s.acc[vb][cb][e] = 0.f;
  s.cur = 0; s.slot = 0; s.res_pending = 0; s.st_pending = 0; s.av_n = -1;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int j = 0; j < 8; ++j) s.av[cb][j] = 0.f;
  s.tile = tile0; s.ch = 0;
  tile_origin(g, tile0, s.n, s.d0, s.h0, s.w0);

  // prologue: first image + weight groups 0 .. P-1, everything landed before the first fragment read
  issue_halo<NCB>(a, lds, s.hp, wave, 0, 1, s.n, s.d0, s.h0, s.w0, 0);
  s.a_ch = 0; s.a_j = 0;
#pragma unroll
  for (int q = 0; q < KK::P; ++q) {
    issue_A<NCB>(a, lds, y, wave, lane, s.a_ch, s.a_j, q);
    if (++s.a_j == KK::NG) { s.a_j = 0; s.a_ch = s.a_ch + 1 == a.nchunks ? 0 : s.a_ch + 1; }
  }
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  s.slot = KK::RD - 1;  // "slot of group -1": group_top<0> steps to slot 0 and refills slot RD-1 with group P
  s.abase = s.abase_next = KK::RING0 + lane * 16;
  issue_frags<0, NCB, FLIP>(s.fr[0], s.bcur, s.abase);
  wait_frags<NCB>(s.fr[0]);
  int par = 0;
  while (true) {
    // next image
    if (s.ch + 1 < a.nchunks) { s.ntile = s.tile; s.nch = s.ch + 1; s.nn = s.n; s.nd0 = s.d0; s.nh0 = s.h0; s.nw0 = s.w0; }
    else if (s.tile + tile_step < tile_last) { s.ntile = s.tile + tile_step; s.nch = 0; tile_origin(g, s.ntile, s.nn, s.nd0, s.nh0, s.nw0); }
    else { s.ntile = -1; s.nch = 0; s.nn = s.nd0 = s.nh0 = s.nw0 = 0; }
    if (par) taps<0, 1, NCB, FLIP>(s, a, lds, y, wave, lane, fast_res);
    else taps<0, 0, NCB, FLIP>(s, a, lds, y, wave, lane, fast_res);
    par ^= 1;
    if (s.ch == a.nchunks - 1 && !(a.dbg & 1)) epilogue<NCB>(s, a, lds, y, wave, lane, fast_res);
    if (s.ntile < 0) break;
    s.cur ^= 1;
    s.tile = s.ntile; s.ch = s.nch; s.n = s.nn; s.d0 = s.nd0; s.h0 = s.nh0; s.w0 = s.nw0;
  }
  wait_vm<0>();  // nothing of this workgroup is in flight when it ends
}

template <int NCB, int FLIP>
int launch27(ConvArgs a, int ntiles, int ny, hipStream_t st) {
  using KK = K<NCB>;
  a.ntiles = ntiles;
  static const char* dbg_env = getenv("MI_C27_DBG");
  a.dbg = dbg_env ? atoi(dbg_env) : 0;
  int gx = (256 / ny) / 8 * 8;  // one workgroup per CU over all cout groups, a multiple of 8 (one slot set per XCD class)
  if (gx < 8) gx = 8;
  const int need = (ntiles + 7) / 8 * 8;
  if (gx > need) gx = need;
  auto kern = k_conv27<NCB, FLIP>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(256), (size_t)KK::LDS_TOTAL, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // namespace

int mi_launch_conv27(const ConvArgs& a, int NCB, int flip, int ntiles, int ny, hipStream_t st) {
  if (a.g.TD != 4 || a.g.TH != 8 || a.g.TW != 8 || (a.x_cs & 7) || (a.Cin & 7)) return MI_ERR_BAD_ARG;
  if (NCB == 1) return flip ? launch27<1, 1>(a, ntiles, ny, st) : launch27<1, 0>(a, ntiles, ny, st);
  if (NCB == 2) return flip ? launch27<2, 1>(a, ntiles, ny, st) : launch27<2, 0>(a, ntiles, ny, st);
  return MI_ERR_BAD_ARG;
}
