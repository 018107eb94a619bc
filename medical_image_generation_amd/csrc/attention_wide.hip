// Fused attention for ONE wide head (d = 512 / 768): the shape every latent UNet of the reference's planner has
// (configuration.py:894 `num_head_channels = [0, 512, 768]` through AttentionBlock._attention, UNet:406-416).  Same contract as
// attention.hip (packed qkv[B*S][3C] in, channels-last y (+ residual) out, no S x S matrix in HBM, no transposes), same "swapped"
// products with the query (or, for dK / dV, the key) on the lane.
//
// What is different: a 32 x d accumulator does not fit one wave (d = 512: 256 fp32 registers for O alone).  The head dimension is
// split in two halves over the waves of a workgroup -- 4 waves = 2 d-halves x 2 blocks of 32 rows:
//   * products that CONTRACT over d (S^T = K Q^T, dP^T = V dO^T) are computed as two partial 32 x 32 tiles, one per half, exchanged
//     through 4 KB of LDS and added (a + b == b + a: both halves hold bit-identical scores, so both run the same softmax and no
//     probability tile ever crosses waves);
//   * products that PRODUCE d (O^T, dQ^T, dK^T, dV^T) need no exchange: a wave accumulates its own half, DH/32 tiles of 32 x 32.
// The row-side operand of a wave (its 32 queries' half rows of Q, dO; or K, V for the key-outer kernel) lives in registers for the
// whole kernel; the other side streams through LDS in tiles of 32 rows, filled by LDS-DMA (one 1 KiB piece = one 512-wide row) so
// that no registers are spent on staging.  Row pitch 2 d + 48 bytes (== 48 mod 256): conflict-free for the 16-byte row-fragment
// reads AND for the transposed 8-byte reads of the same tile.
// Register budget of a wave at d = 512: 64 (row operand) + 128 (accumulators, AGPRs) forward; 128 + 128 for dQ; 128 + 256 for the
// fused dK / dV kernel.  At d = 768 dK and dV run as two passes of the same kernel (384 accumulator registers do not exist), and
// the tiles are single-buffered (two 50 KB tiles + the exchange area); d = 768 sits at 10^3 tokens in the reference's nets, where
// none of this matters.
#include "attention_common.h"

using namespace mi_attn;

namespace {

typedef __attribute__((address_space(3))) void lds_void;

template <int DH>
struct W {
  static constexpr int D = 2 * DH;
  static constexpr int ROWB = 2 * D;        // bytes of one operand row
  static constexpr int P = ROWB + 48;       // LDS row pitch
  static constexpr int TILE = 32 * P;
  static constexpr int DK = DH / 16, DB = DH / 32;
  static constexpr int NPIECE = (ROWB + 1023) / 1024, TAIL = ROWB % 1024;
  static constexpr int NDMA = 8 * NPIECE;   // LDS-DMA instructions per wave and tile
  static constexpr bool DBUF = DH <= 256;   // room for a second buffer of the tile that is read twice per iteration
  static_assert(D % 128 == 0, "pitch rule");
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// Rows r0 .. r0+31 of an S-row operand (base = row 0 of the image, first column of the head) into an LDS tile, by all 4 waves.
// Rows >= S come back as zeros through the buffer range check; the row offset is the instruction's scalar offset.
template <int DH>
__device__ __forceinline__ void tile_dma(char* tile, const bf16* base, int r0, int S, int ld, int wave, int lane) {
  typedef W<DH> T;
  const int rem = S - r0;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base + (int64_t)r0 * ld), 0, rem > 0 ? (rem > 32 ? 32 : rem) * ld * 2 : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wave + 4 * i;
#pragma unroll
    for (int p = 0; p < T::NPIECE; ++p) {
      if ((p + 1) * 1024 <= T::ROWB || lane * 16 < T::TAIL)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(tile + row * T::P + p * 1024), 16, lane * 16 + p * 1024, row * ld * 2, 0, 0);
    }
  }
}

// partial 32 x 32 tile -> this wave's slot of the exchange area (lane-linear 16-byte pieces), and the sum with the partner's
__device__ __forceinline__ void xch_put(char* slot, const f32x16& t, int lane) {
#pragma unroll
  for (int e = 0; e < 4; ++e) *(f32x4*)(slot + e * 1024 + lane * 16) = f32x4{t[4 * e], t[4 * e + 1], t[4 * e + 2], t[4 * e + 3]};
}
__device__ __forceinline__ void xch_add(const char* slot, f32x16& t, int lane) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 v = *(const f32x4*)(slot + e * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[4 * e + j] += v[j];
  }
}

// accumulators O^T / dQ^T / dK^T / dV^T (rows = channels of this wave's half, column = the lane's token) -> bf16 row `row` of out
template <int NB_>
__device__ __forceinline__ void store_half(const f32x16 (&acc)[NB_], float mul, const bf16* resid, bf16* out, int64_t row, int ldo, int col0,
                                           int h) {
#pragma unroll
  for (int i = 0; i < NB_; ++i)
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      const int col = col0 + i * 32 + grp * 8 + h * 4;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[i][grp * 4 + j] * mul;
      if (resid) {
        const bf16* rp = resid + row * ldo + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += bf2f(rp[j]);
      }
      u32x2 w = {pack2(v[0], v[1]), pack2(v[2], v[3])};
      *(u32x2*)(out + row * ldo + col) = w;
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int DH>
__global__ void __launch_bounds__(256) k_attnw_fwd(AttnArgs a) {
  typedef W<DH> T;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *kt = lds, *vt = lds + T::TILE, *xch = lds + 2 * T::TILE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const int dh = wave >> 1, qb = wave & 1;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int S = a.S;
  const int64_t rb = (int64_t)b * S;
  const int q = blockIdx.x * 64 + qb * 32 + r;
  const int qc = hd * T::D, kc = a.C + hd * T::D, vc = 2 * a.C + hd * T::D;
  const float c = a.scale * kLog2e;

  bf16x8 qf[T::DK];  // B operand of the partial S^T = K Q^T over this wave's half of d
#pragma unroll
  for (int ks = 0; ks < T::DK; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + q, rb + S, a.ld, qc + dh * DH + ks * 16 + h * 8));
  f32x16 o[T::DB];
#pragma unroll
  for (int i = 0; i < T::DB; ++i) o[i] = kZero16;
  float m = -INFINITY, l = 0.f;

  const int kbeg = blockIdx.z * a.tps * 32, kend = min(S, kbeg + a.tps * 32);
  const bf16 *kbase = a.qkv + rb * a.ld + kc, *vbase = a.qkv + rb * a.ld + vc;
  tile_dma<DH>(kt, kbase, kbeg, S, a.ld, wave, lane);
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    wait_vm<0>();
    __syncthreads();  // K(k0) is in LDS; every wave is past the previous tile's O^T += V^T P^T
    tile_dma<DH>(vt, vbase, k0, S, a.ld, wave, lane);  // lands during the score phase
    f32x16 st = kZero16;
#pragma unroll
    for (int ks = 0; ks < T::DK; ++ks)
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, T::P, 0, dh * T::DK + ks, lane), qf[ks], st, 0, 0, 0);
    xch_put(xch + wave * 4096, st, lane);
    __syncthreads();  // partial scores visible; the K tile is free
    const bool more = k0 + 32 < kend;
    if (more) tile_dma<DH>(kt, kbase, k0 + 32, S, a.ld, wave, lane);
    xch_add(xch + (wave ^ 2) * 4096, st, lane);
    if (k0 + 32 > S) {  // ragged last tile: keys >= S must not count (their K rows were read as zeros)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (k0 + (e & 3) + 8 * (e >> 2) + 4 * h >= S) st[e] = -INFINITY;
    }
    float tmax = st[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, st[e]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * c;
    if (__builtin_amdgcn_ballot_w64(tmax > m)) {
      const float mn = fmaxf(m, tmax);
      const float alpha = ex2(m - mn);
      l *= alpha;
      m = mn;
#pragma unroll
      for (int i = 0; i < T::DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    }
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = ex2(st[e] * c - m);
      st[e] = p;
      ps += p;
    }
    l += ps;
    if (more) wait_vm<T::NDMA>(); else wait_vm<0>();  // V(k0) has landed (the next K tile may still fly)
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = pack_acc8(st, s);
#pragma unroll
      for (int i = 0; i < T::DB; ++i)
        o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vt, T::P, 0, s, dh * T::DB + i, lane), pf, o[i], 0, 0, 0);
    }
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = lt > 0.f ? 1.f / lt : 0.f;
  if (q >= S) return;
  if (a.nsplit > 1) {
    if (h == 0 && dh == 0) *(float2*)(a.part_ml + (((int64_t)blockIdx.z * gridDim.y + bh) * S + q) * 2) = make_float2(m, lt);
    bf16* po = a.part + (int64_t)blockIdx.z * (gridDim.y / a.heads) * S * a.C;
    store_half(o, inv, nullptr, po, rb + q, a.C, qc + dh * DH, h);
    return;
  }
  if (h == 0 && dh == 0 && a.lse) a.lse[(int64_t)bh * S + q] = m + log2f(lt);
  store_half(o, inv, a.resid, a.y, rb + q, a.C, qc + dh * DH, h);
}

// ------------------------------------------------------------------------------------------------ backward, query outer: dQ
template <int DH>
__global__ void __launch_bounds__(256) k_attnw_dq(AttnArgs a) {
  typedef W<DH> T;
  constexpr int NB = T::DBUF ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *kt = lds, *vt = lds + NB * T::TILE, *xch = lds + (NB + 1) * T::TILE;  // xch: [wave][S^T | dP^T] 4 KB each
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const int dh = wave >> 1, qb = wave & 1;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int S = a.S;
  const int64_t rb = (int64_t)b * S;
  const int q = blockIdx.x * 64 + qb * 32 + r;
  const int qc = hd * T::D, kc = a.C + hd * T::D, vc = 2 * a.C + hd * T::D;
  const float c = a.scale * kLog2e;

  bf16x8 qf[T::DK], dof[T::DK];
#pragma unroll
  for (int ks = 0; ks < T::DK; ++ks) {
    qf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + q, rb + S, a.ld, qc + dh * DH + ks * 16 + h * 8));
    dof[ks] = __builtin_bit_cast(bf16x8, ld16(a.dy, rb + q, rb + S, a.C, qc + dh * DH + ks * 16 + h * 8));
  }
  const float lse = q < S ? a.lse[(int64_t)bh * S + q] : 0.f;
  const float drow = q < S ? a.dsum[(int64_t)bh * S + q] : 0.f;
  f32x16 dq[T::DB];
#pragma unroll
  for (int i = 0; i < T::DB; ++i) dq[i] = kZero16;

  const int kbeg = blockIdx.z * a.tps * 32, kend = min(S, kbeg + a.tps * 32);
  const bf16 *kbase = a.qkv + rb * a.ld + kc, *vbase = a.qkv + rb * a.ld + vc;
  tile_dma<DH>(kt, kbase, kbeg, S, a.ld, wave, lane);
  tile_dma<DH>(vt, vbase, kbeg, S, a.ld, wave, lane);
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    wait_vm<0>();
    __syncthreads();  // K(k0), V(k0) in LDS; every wave is past the previous tile's dQ^T += K^T dS^T
    const bool more = k0 + 32 < kend;
    char* kc_t = kt + cur * T::TILE;
    if (T::DBUF && more) tile_dma<DH>(kt + (cur ^ 1) * T::TILE, kbase, k0 + 32, S, a.ld, wave, lane);
    f32x16 st = kZero16, dp = kZero16;
#pragma unroll
    for (int ks = 0; ks < T::DK; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kc_t, T::P, 0, dh * T::DK + ks, lane), qf[ks], st, 0, 0, 0);   // S^T  = K Q^T
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vt, T::P, 0, dh * T::DK + ks, lane), dof[ks], dp, 0, 0, 0);    // dP^T = V dO^T
    }
    xch_put(xch + wave * 8192, st, lane);
    xch_put(xch + wave * 8192 + 4096, dp, lane);
    __syncthreads();  // partials visible; the V tile is free
    if (more) tile_dma<DH>(vt, vbase, k0 + 32, S, a.ld, wave, lane);
    xch_add(xch + (wave ^ 2) * 8192, st, lane);
    xch_add(xch + (wave ^ 2) * 8192 + 4096, dp, lane);
    // keys >= S need no mask: their K rows are zeros, so whatever dS^T holds there multiplies zeros in dQ^T += K^T dS^T
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = ex2(st[e] * c - lse) * (dp[e] - drow) * a.scale;  // dS^T = P (dP - D) scale
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 df = pack_acc8(st, s);
#pragma unroll
      for (int i = 0; i < T::DB; ++i)
        dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(kc_t, T::P, 0, s, dh * T::DB + i, lane), df, dq[i], 0, 0, 0);
    }
    if (T::DBUF) cur ^= 1;
    else {
      __syncthreads();  // single K buffer: free only now
      if (more) tile_dma<DH>(kt, kbase, k0 + 32, S, a.ld, wave, lane);
    }
  }
  if (q >= S) return;
  bf16* out = a.dqkv;
  int ldo = a.ld;
  if (a.nsplit > 1) out = a.part + (int64_t)blockIdx.z * (gridDim.y / a.heads) * S * 3 * a.C, ldo = 3 * a.C;
  store_half(dq, 1.f, nullptr, out, rb + q, ldo, qc + dh * DH, h);
}

// ------------------------------------------------------------------------------------------------ backward, key outer: dK, dV
// The key sits on the lane; Q / dO tiles of 32 queries stream through LDS.  WHICH: 0 = dK and dV, 1 = dK only, 2 = dV only.
template <int DH, int WHICH>
__global__ void __launch_bounds__(256) k_attnw_dkv(AttnArgs a) {
  typedef W<DH> T;
  constexpr bool DK_ = WHICH != 2, DV_ = WHICH != 1;
  constexpr int NB = T::DBUF ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *qt = lds, *dt = lds + NB * T::TILE, *xch = lds + (NB + 1) * T::TILE;
  float* lse_s = (float*)(xch + 4 * 8192);
  float* dsum_s = lse_s + 32;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const int dh = wave >> 1, kb = wave & 1;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int S = a.S;
  const int64_t rb = (int64_t)b * S;
  const int key = blockIdx.x * 64 + kb * 32 + r;
  const int qc = hd * T::D, kc = a.C + hd * T::D, vc = 2 * a.C + hd * T::D;
  const float c = a.scale * kLog2e;

  bf16x8 kf[T::DK], vf[DK_ ? T::DK : 1];  // B operands of S = Q K^T and dP = dO V^T over this wave's half of d
#pragma unroll
  for (int ks = 0; ks < T::DK; ++ks) {
    kf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + key, rb + S, a.ld, kc + dh * DH + ks * 16 + h * 8));
    if constexpr (DK_) vf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + key, rb + S, a.ld, vc + dh * DH + ks * 16 + h * 8));
  }
  f32x16 dk[DK_ ? T::DB : 1], dv[DV_ ? T::DB : 1];
#pragma unroll
  for (int i = 0; i < T::DB; ++i) {
    if constexpr (DK_) dk[i] = kZero16;
    if constexpr (DV_) dv[i] = kZero16;
  }

  const int qbeg = blockIdx.z * a.tps * 32, qend = min(S, qbeg + a.tps * 32);
  const bf16 *qbase = a.qkv + rb * a.ld + qc, *dbase = a.dy + rb * a.C + hd * T::D;
  tile_dma<DH>(qt, qbase, qbeg, S, a.ld, wave, lane);
  tile_dma<DH>(dt, dbase, qbeg, S, a.C, wave, lane);
  int cur = 0;
  for (int q0 = qbeg; q0 < qend; q0 += 32) {
    wait_vm<0>();
    __syncthreads();  // Q(q0), dO(q0) in LDS; every wave is past the previous tile
    const bool more = q0 + 32 < qend;
    char* qc_t = qt + cur * T::TILE;
    if (T::DBUF && more) tile_dma<DH>(qt + (cur ^ 1) * T::TILE, qbase, q0 + 32, S, a.ld, wave, lane);
    if (threadIdx.x < 32) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < S ? a.lse[(int64_t)bh * S + qq] : INFINITY;  // +inf -> p = 0 for padded queries
      dsum_s[threadIdx.x] = qq < S ? a.dsum[(int64_t)bh * S + qq] : 0.f;
    }
    f32x16 st = kZero16, dp = kZero16;
#pragma unroll
    for (int ks = 0; ks < T::DK; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qc_t, T::P, 0, dh * T::DK + ks, lane), kf[ks], st, 0, 0, 0);          // S  = Q K^T
      if constexpr (DK_) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dt, T::P, 0, dh * T::DK + ks, lane), vf[ks], dp, 0, 0, 0);  // dP = dO V^T
    }
    xch_put(xch + wave * 8192, st, lane);
    if (DK_) xch_put(xch + wave * 8192 + 4096, dp, lane);
    __syncthreads();  // partials (and the row terms) visible
    if (!DV_ && more) tile_dma<DH>(dt, dbase, q0 + 32, S, a.C, wave, lane);               // dK only: dO was needed for dP alone
    if (!DK_ && !T::DBUF && more) tile_dma<DH>(qt, qbase, q0 + 32, S, a.ld, wave, lane);  // dV only: Q was needed for S alone
    xch_add(xch + (wave ^ 2) * 8192, st, lane);
    if (DK_) xch_add(xch + (wave ^ 2) * 8192 + 4096, dp, lane);
    // keys >= S (lanes past the end) compute finite values that are never stored: no key mask needed
    f32x16 pr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ql = (e & 3) + 8 * (e >> 2) + 4 * h;
      const float p = ex2(st[e] * c - lse_s[ql]);
      pr[e] = p;
      if (DK_) st[e] = p * (dp[e] - dsum_s[ql]) * a.scale;  // dS
    }
    if constexpr (DV_) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = pack_acc8(pr, s);
#pragma unroll
        for (int i = 0; i < T::DB; ++i)
          dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(dt, T::P, 0, s, dh * T::DB + i, lane), pf, dv[i], 0, 0, 0);  // dV^T += dO^T P
      }
      if (DK_ || !T::DBUF) __syncthreads();  // the dO tile is free: its refill overlaps the dK product
      if (more) tile_dma<DH>(dt, dbase, q0 + 32, S, a.C, wave, lane);
    }
    if constexpr (DK_) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 df = pack_acc8(st, s);
#pragma unroll
        for (int i = 0; i < T::DB; ++i)
          dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qc_t, T::P, 0, s, dh * T::DB + i, lane), df, dk[i], 0, 0, 0);  // dK^T += Q^T dS
      }
      if (!T::DBUF) {
        __syncthreads();  // single Q buffer: free only now
        if (more) tile_dma<DH>(qt, qbase, q0 + 32, S, a.ld, wave, lane);
      }
    }
    if (T::DBUF) cur ^= 1;
  }
  if (key >= S) return;
  bf16* out = a.dqkv;
  int ldo = a.ld;
  if (a.nsplit > 1) out = a.part + (int64_t)blockIdx.z * (gridDim.y / a.heads) * S * 3 * a.C, ldo = 3 * a.C;
  if constexpr (DK_) store_half(dk, 1.f, nullptr, out, rb + key, ldo, kc + dh * DH, h);
  if constexpr (DV_) store_half(dv, 1.f, nullptr, out, rb + key, ldo, vc + dh * DH, h);
}

constexpr int kMaxSplit = 8;
// ~2 workgroups per CU over the grid, at least 8 tiles (256 rows of the reduction axis) per split
void pick_split(int B, int heads, int S, int& nsplit, int& tps) {
  const int base = (S + 63) / 64 * B * heads, ntiles = (S + 31) / 32;
  int want = 512 / (base > 0 ? base : 1);
  static const int env = [] { const char* e = getenv("MI_ATTNW_SPLIT"); return e ? atoi(e) : 0; }();
  if (env > 0) want = env;
  want = want < 1 ? 1 : (want > kMaxSplit ? kMaxSplit : want);
  if (!env && want > ntiles / 8) want = ntiles / 8 > 0 ? ntiles / 8 : 1;
  tps = (ntiles + want - 1) / want;
  nsplit = (ntiles + tps - 1) / tps;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t fwd_ws(int C, int heads, int B, int S, int ns) { return ns > 1 ? align256((size_t)ns * B * S * C * 2) + (size_t)ns * B * heads * S * 8 : 0; }
size_t bwd_ws(int C, int B, int S, int ns) { return ns > 1 ? (size_t)ns * B * S * 3 * C * 2 : 0; }

template <typename Kern>
int launchw(Kern kern, const AttnArgs& a, int B, size_t lds, hipStream_t st) {
  // (one attribute call per kernel instantiation: the static lives in this template instance)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.S + 63) / 64, B * a.heads, a.nsplit), dim3(256), lds, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}
template <int DH> constexpr size_t lds_fwd() { return 2 * W<DH>::TILE + 4 * 4096; }
template <int DH> constexpr size_t lds_bwd() { return ((W<DH>::DBUF ? 2 : 1) + 1) * W<DH>::TILE + 4 * 8192 + 256; }
static_assert(lds_bwd<256>() <= 160 * 1024 && lds_bwd<384>() <= 160 * 1024, "LDS budget");

}  // namespace

namespace mi_attn {

bool attnw_supported(int C, int heads) {
  static const int on = [] { const char* e = getenv("MI_ATTN_WIDE"); return e ? atoi(e) : 1; }();  // 0: materialised path (A/B runs)
  if (!on || heads <= 0 || C % heads) return false;
  const int d = C / heads;
  return d == 512 || d == 768;
}

int64_t attnw_workspace_bytes(int C, int heads, int B, int S) {
  if (!attnw_supported(C, heads) || B <= 0 || S <= 0) return 0;
  int ns, tps;
  pick_split(B, heads, S, ns, tps);
  const size_t f = fwd_ws(C, heads, B, S, ns), b = bwd_ws(C, B, S, ns);
  return (int64_t)(f > b ? f : b);
}

int attnw_fwd(AttnArgs a, int B, void* ws, int64_t ws_bytes, hipStream_t st) {
  if (B <= 0 || a.S <= 0 || (a.ld & 7) || a.ld < 3 * a.C) return MI_ERR_BAD_ARG;
  if ((int64_t)B * a.S * a.ld * 2 >= (1ll << 31)) return MI_ERR_UNSUPPORTED;  // 32-bit offsets inside one image's rows
  pick_split(B, a.heads, a.S, a.nsplit, a.tps);
  if (!ws || (size_t)ws_bytes < fwd_ws(a.C, a.heads, B, a.S, a.nsplit)) a.nsplit = 1, a.tps = (a.S + 31) / 32;
  a.part = (bf16*)ws;
  a.part_ml = (float*)((char*)ws + align256((size_t)a.nsplit * B * a.S * a.C * 2));
  const int e = a.C / a.heads == 512 ? launchw(k_attnw_fwd<256>, a, B, lds_fwd<256>(), st) : launchw(k_attnw_fwd<384>, a, B, lds_fwd<384>(), st);
  if (e) return e;
  if (a.nsplit > 1) attn_merge_fwd_launch(a, B, st);
  MI_CHECK_LAUNCH();
  return 0;
}

int attnw_bwd(AttnArgs a, int B, const bf16* y, const bf16* resid, void* ws, int64_t ws_bytes, hipStream_t st) {
  if (B <= 0 || a.S <= 0 || (a.ld & 7) || a.ld < 3 * a.C) return MI_ERR_BAD_ARG;
  if ((int64_t)B * a.S * a.ld * 2 >= (1ll << 31)) return MI_ERR_UNSUPPORTED;
  pick_split(B, a.heads, a.S, a.nsplit, a.tps);
  if (!ws || (size_t)ws_bytes < bwd_ws(a.C, B, a.S, a.nsplit)) a.nsplit = 1, a.tps = (a.S + 31) / 32;
  a.part = (bf16*)ws;
  attn_dsum_launch(a, y, resid, B, st);
  int e;
  if (a.C / a.heads == 512) {
    if ((e = launchw(k_attnw_dq<256>, a, B, lds_bwd<256>(), st))) return e;
    if ((e = launchw(k_attnw_dkv<256, 0>, a, B, lds_bwd<256>(), st))) return e;
  } else {
    if ((e = launchw(k_attnw_dq<384>, a, B, lds_bwd<384>(), st))) return e;
    if ((e = launchw(k_attnw_dkv<384, 1>, a, B, lds_bwd<384>(), st))) return e;
    if ((e = launchw(k_attnw_dkv<384, 2>, a, B, lds_bwd<384>(), st))) return e;
  }
  if (a.nsplit > 1) attn_merge_bwd_launch(a, B, st);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // namespace mi_attn
