// Fused self-attention for AttentionBlock._attention (UNet:406-416, AEKL:271-281): softmax(Q K^T * scale) V without
// materialising the S x S score matrix, forward and backward, for head dims 32 and 64 (wider single-head blocks -- the
// reference's latent UNet uses one head of 512 / 768 -- keep the GEMM + softmax path in gemm.hip).
//
// Operands live in the packed projection buffer qkv[B*S][3C] (Q | K | V, head h at column h*d), outputs go straight into
// the channels-last activation y[B][S][C] (+ residual), so no head split / merge copies exist.
//
// All three kernels use the "swapped" product S^T = K Q^T on mfma_f32_32x32x16_bf16 so that the QUERY index sits on the
// lane: running max / sum, LSE and the row term D of the backward are then lane-local scalars, and the probability tile
// (keys down the registers) is already the B operand of the next product that sums over keys (cdna guide: "an accumulator
// tile as the next MFMA's operand"), with the matching k-order produced by ds_read_b64_tr_b16 on the other operand.
//   forward            : O^T[dv,q]  += V^T[dv,key]  P^T[key,q]        (online softmax, exp2 domain)
//   backward, q outer  : dQ^T[d,q]  += K^T[d,key]   dS^T[key,q]
//   backward, kv outer : dV^T[dv,k] += dO^T[dv,q]   P[q,key] ,  dK^T[d,k] += Q^T[d,q] dS[q,key]   (key on the lane)
// One workgroup = 4 waves x 32 rows; K/V (or Q/dO) tiles of 64 rows are staged through LDS by all 4 waves.
#include "attention_common.h"

using namespace mi_attn;

namespace {

template <int D>
struct Tiles {  // LDS images of a 64-row tile of two operands
  static constexpr int PA = D * 2 + 16;             // row pitch for b128 row fragments (16 * odd)
  static constexpr int PB = D == 64 ? 192 : 64;     // row pitch for transposed reads (4 rows hit distinct 64-B bank groups)
  static constexpr int PIECES = 64 * D / 8 / 256;   // 16-byte pieces per thread per operand
};

// 64 rows r0.. of an S-row operand (base = row 0, first column of the head) into registers; rows >= S read as zeros through the
// buffer range check, and the per-tile part of the address lives in the (scalar) descriptor: no vector address math in the loops
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const bf16* base, int r0, int S, int ld) {
  const int rem = S - r0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base + (int64_t)r0 * ld), 0, rem > 0 ? rem * ld * 2 : 0, 0x00020000);
}
template <int D>
__device__ __forceinline__ void tile_load(u32x4 (&reg)[Tiles<D>::PIECES], const bf16* base, int r0, int S, int ld) {
  const __amdgpu_buffer_rsrc_t rs = tile_rsrc(base, r0, S, ld);
#pragma unroll
  for (int i = 0; i < Tiles<D>::PIECES; ++i) {
    int pc = threadIdx.x + 256 * i;
    int r = pc / (D / 8), c = (pc % (D / 8)) * 8;
    reg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * ld + c) * 2, 0, 0));
  }
}
template <int D>
__device__ __forceinline__ void tile_store(const u32x4 (&reg)[Tiles<D>::PIECES], char* tile, int pitch) {
#pragma unroll
  for (int i = 0; i < Tiles<D>::PIECES; ++i) {
    int pc = threadIdx.x + 256 * i;
    int r = pc / (D / 8), c = (pc % (D / 8)) * 8;
    *(u32x4*)(tile + r * pitch + c * 2) = reg[i];
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ void __launch_bounds__(256) k_attn_fwd(AttnArgs a) {
  constexpr int DK = D / 16, DVB = D / 32;
  typedef Tiles<D> T;
  __shared__ __attribute__((aligned(16))) char kt[64 * T::PA];
  __shared__ __attribute__((aligned(16))) char vt[64 * T::PB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int S = a.S;
  const int64_t rb = (int64_t)b * S;  // first row of this image in qkv / y
  const int q = blockIdx.x * 128 + wave * 32 + r;
  const int qc = hd * D, kc = a.C + hd * D, vc = 2 * a.C + hd * D;
  const float c = a.scale * kLog2e;

  bf16x8 qf[DK];  // B operand of S^T = K Q^T: lane (q, h) holds Q[q][ks*16 + 8h ..]
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + q, rb + S, a.ld, qc + ks * 16 + h * 8));

  f32x16 o[DVB];
#pragma unroll
  for (int i = 0; i < DVB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int kbeg = blockIdx.z * a.tps * 64, kend = min(S, kbeg + a.tps * 64);
  const bf16 *kbase = a.qkv + rb * a.ld + kc, *vbase = a.qkv + rb * a.ld + vc;
  u32x4 kr[T::PIECES], vr[T::PIECES];
  tile_load<D>(kr, kbase, kbeg, S, a.ld);
  tile_load<D>(vr, vbase, kbeg, S, a.ld);
  for (int k0 = kbeg; k0 < kend; k0 += 64) {
    __syncthreads();
    tile_store<D>(kr, kt, T::PA);
    tile_store<D>(vr, vt, T::PB);
    __syncthreads();
    if (k0 + 64 < kend) {
      tile_load<D>(kr, kbase, k0 + 64, S, a.ld);
      tile_load<D>(vr, vbase, k0 + 64, S, a.ld);
    }
    // S^T tile: 64 keys (2 blocks of 32) x 32 queries
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, T::PA, kb * 32, 0, lane), qf[0], kZero16, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < DK; ++ks) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, T::PA, kb * 32, ks, lane), qf[ks], st[kb], 0, 0, 0);
    }
    if (k0 + 64 > S) {  // ragged last tile: keys >= S must not count (their K rows were read as zeros)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= S) st[kb][e] = -INFINITY;
    }
    float tmax = st[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, st[kb][e]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * c;  // the two lane halves hold the other keys of the same query (c > 0)
    if (__builtin_amdgcn_ballot_w64(tmax > m)) {  // some query's running max moved: rescale (rare once the maxima settle)
      const float mn = fmaxf(m, tmax);
      const float alpha = ex2(m - mn);  // first tile: m = -inf -> 0
      l *= alpha;
      m = mn;
#pragma unroll
      for (int i = 0; i < DVB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    }
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = ex2(st[kb][e] * c - m);
        st[kb][e] = p;
        ps += p;
      }
    l += ps;
    // O^T += V^T P^T : per 16-key step the probability registers are the B operand, V^T comes from the transposed read
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = pack_acc8(st[kb], s);
#pragma unroll
        for (int i = 0; i < DVB; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vt, T::PB, kb * 32, s, i, lane), pf, o[i], 0, 0, 0);
      }
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = lt > 0.f ? 1.f / lt : 0.f;
  if (a.nsplit > 1) {
    if (q < S) {
      if (h == 0) *(float2*)(a.part_ml + (((int64_t)blockIdx.z * gridDim.y + bh) * S + q) * 2) = make_float2(m, lt);
      bf16* po = a.part + ((int64_t)blockIdx.z * gridDim.y / a.heads * S + rb + q) * a.C;
#pragma unroll
      for (int i = 0; i < DVB; ++i)
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
          const int col = hd * D + i * 32 + grp * 8 + h * 4;
          u32x2 w = {pack2(o[i][grp * 4] * inv, o[i][grp * 4 + 1] * inv), pack2(o[i][grp * 4 + 2] * inv, o[i][grp * 4 + 3] * inv)};
          *(u32x2*)(po + col) = w;
        }
    }
    return;
  }
  if (q < S) {
    if (h == 0 && a.lse) a.lse[(int64_t)bh * S + q] = m + log2f(lt);
#pragma unroll
    for (int i = 0; i < DVB; ++i)
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int col = hd * D + i * 32 + grp * 8 + h * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = o[i][grp * 4 + j] * inv;
        if (a.resid) {
          const bf16* rp = a.resid + (rb + q) * a.C + col;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += bf2f(rp[j]);
        }
        u32x2 w = {pack2(v[0], v[1]), pack2(v[2], v[3])};
        *(u32x2*)(a.y + (rb + q) * a.C + col) = w;
      }
  }
}

// ------------------------------------------------------------------------------------------------ backward: D = rowsum(dO * O)
// O = y - x (y stored with the residual).  One wave per row.
__global__ void __launch_bounds__(256) k_attn_dsum(const bf16* __restrict__ dy, const bf16* __restrict__ y, const bf16* __restrict__ x,
                                                   float* __restrict__ dsum, int C, int heads, int S, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int D = C / heads;
  const int64_t b = row / S, q = row % S;
  for (int hd = 0; hd < heads; ++hd) {
    float acc = 0.f;
    for (int j = lane; j < D; j += 64) {
      int col = hd * D + j;
      float o = bf2f(y[row * C + col]) - (x ? bf2f(x[row * C + col]) : 0.f);
      acc += bf2f(dy[row * C + col]) * o;
    }
    acc = wave_sum(acc);
    if (lane == 0) dsum[(b * heads + hd) * S + q] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ backward, query outer: dQ
template <int D>
__global__ void __launch_bounds__(256) k_attn_bwd_dq(AttnArgs a) {
  constexpr int DK = D / 16, DB = D / 32;
  typedef Tiles<D> T;
  __shared__ __attribute__((aligned(16))) char kt[64 * T::PB];   // K: row fragments for S^T AND transposed for dQ^T -> use PB pitch
  __shared__ __attribute__((aligned(16))) char vt[64 * T::PA];   // V: row fragments for dP^T
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int S = a.S;
  const int64_t rb = (int64_t)b * S;
  const int q = blockIdx.x * 128 + wave * 32 + r;
  const int qc = hd * D, kc = a.C + hd * D, vc = 2 * a.C + hd * D;
  const float c = a.scale * kLog2e;

  bf16x8 qf[DK], dof[DK];
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    qf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + q, rb + S, a.ld, qc + ks * 16 + h * 8));
    dof[ks] = __builtin_bit_cast(bf16x8, ld16(a.dy, rb + q, rb + S, a.C, hd * D + ks * 16 + h * 8));
  }
  const float lse = q < S ? a.lse[(int64_t)bh * S + q] : 0.f;
  const float dq_row = q < S ? a.dsum[(int64_t)bh * S + q] : 0.f;
  f32x16 dq[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;

  const int kbeg = blockIdx.z * a.tps * 64, kend = min(S, kbeg + a.tps * 64);
  const bf16 *kbase = a.qkv + rb * a.ld + kc, *vbase = a.qkv + rb * a.ld + vc;
  u32x4 kr[T::PIECES], vr[T::PIECES];
  tile_load<D>(kr, kbase, kbeg, S, a.ld);
  tile_load<D>(vr, vbase, kbeg, S, a.ld);
  for (int k0 = kbeg; k0 < kend; k0 += 64) {
    __syncthreads();
    tile_store<D>(kr, kt, T::PB);
    tile_store<D>(vr, vt, T::PA);
    __syncthreads();
    if (k0 + 64 < kend) {
      tile_load<D>(kr, kbase, k0 + 64, S, a.ld);
      tile_load<D>(vr, vbase, k0 + 64, S, a.ld);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, T::PB, kb * 32, 0, lane), qf[0], kZero16, 0, 0, 0);   // S^T  = K Q^T
      f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vt, T::PA, kb * 32, 0, lane), dof[0], kZero16, 0, 0, 0);  // dP^T = V dO^T
#pragma unroll
      for (int ks = 1; ks < DK; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, T::PB, kb * 32, ks, lane), qf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vt, T::PA, kb * 32, ks, lane), dof[ks], dp, 0, 0, 0);
      }
      // keys >= S need no mask: their K rows are zeros, so whatever dS^T holds there multiplies zeros in dQ^T += K^T dS^T
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = ex2(st[e] * c - lse) * (dp[e] - dq_row) * a.scale;  // dS^T = P (dP - D) scale
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 df = pack_acc8(st, s);
#pragma unroll
        for (int i = 0; i < DB; ++i) dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(kt, T::PB, kb * 32, s, i, lane), df, dq[i], 0, 0, 0);
      }
    }
  }
  bf16* out = a.dqkv;
  int ldo = a.ld;
  if (a.nsplit > 1) out = a.part + (int64_t)blockIdx.z * (gridDim.y / a.heads) * S * 3 * a.C, ldo = 3 * a.C;
  if (q < S) {
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int col = qc + i * 32 + grp * 8 + h * 4;
        u32x2 w = {pack2(dq[i][grp * 4], dq[i][grp * 4 + 1]), pack2(dq[i][grp * 4 + 2], dq[i][grp * 4 + 3])};
        *(u32x2*)(out + (rb + q) * ldo + col) = w;
      }
  }
}

// ------------------------------------------------------------------------------------------------ backward, key outer: dK, dV
// Same machinery with the roles of queries and keys exchanged: the KEY sits on the lane, S = Q K^T tiles have queries down
// the registers, so P and dS are the B operands of the products that sum over queries.  LSE / D are per register row here.
template <int D>
__global__ void __launch_bounds__(256) k_attn_bwd_dkv(AttnArgs a) {
  constexpr int DK = D / 16, DB = D / 32;
  typedef Tiles<D> T;
  __shared__ __attribute__((aligned(16))) char qt[64 * T::PB];   // Q tile (rows + transposed)
  __shared__ __attribute__((aligned(16))) char dt[64 * T::PB];   // dO tile (rows + transposed)
  __shared__ float lse_s[64], dsum_s[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int S = a.S;
  const int64_t rb = (int64_t)b * S;
  const int key = blockIdx.x * 128 + wave * 32 + r;
  const int qc = hd * D, kc = a.C + hd * D, vc = 2 * a.C + hd * D;
  const float c = a.scale * kLog2e;

  bf16x8 kf[DK], vf[DK];  // B operands of S = Q K^T and dP = dO V^T: lane (key, h) holds K[key][ks*16 + 8h ..]
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    kf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + key, rb + S, a.ld, kc + ks * 16 + h * 8));
    vf[ks] = __builtin_bit_cast(bf16x8, ld16(a.qkv, rb + key, rb + S, a.ld, vc + ks * 16 + h * 8));
  }
  f32x16 dk[DB], dv[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[i][e] = dv[i][e] = 0.f;

  const int qbeg = blockIdx.z * a.tps * 64, qend = min(S, qbeg + a.tps * 64);
  const bf16 *qbase = a.qkv + rb * a.ld + qc, *dbase = a.dy + rb * a.C + hd * D;
  u32x4 qr[T::PIECES], dr[T::PIECES];
  tile_load<D>(qr, qbase, qbeg, S, a.ld);
  tile_load<D>(dr, dbase, qbeg, S, a.C);
  for (int q0 = qbeg; q0 < qend; q0 += 64) {
    __syncthreads();
    tile_store<D>(qr, qt, T::PB);
    tile_store<D>(dr, dt, T::PB);
    if (threadIdx.x < 64) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < S ? a.lse[(int64_t)bh * S + qq] : INFINITY;  // +inf -> p = 0 for padded queries
      dsum_s[threadIdx.x] = qq < S ? a.dsum[(int64_t)bh * S + qq] : 0.f;
    }
    __syncthreads();
    if (q0 + 64 < qend) {
      tile_load<D>(qr, qbase, q0 + 64, S, a.ld);
      tile_load<D>(dr, dbase, q0 + 64, S, a.C);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qt, T::PB, qb * 32, 0, lane), kf[0], kZero16, 0, 0, 0);  // S  = Q K^T
      f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dt, T::PB, qb * 32, 0, lane), vf[0], kZero16, 0, 0, 0);  // dP = dO V^T
#pragma unroll
      for (int ks = 1; ks < DK; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qt, T::PB, qb * 32, ks, lane), kf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dt, T::PB, qb * 32, ks, lane), vf[ks], dp, 0, 0, 0);
      }
      // keys >= S (lanes past the end) compute garbage-free finite values that are never stored: no key mask needed
      f32x16 pr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ql = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const float p = ex2(st[e] * c - lse_s[ql]);
        pr[e] = p;
        st[e] = p * (dp[e] - dsum_s[ql]) * a.scale;  // dS
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = pack_acc8(pr, s), df = pack_acc8(st, s);
#pragma unroll
        for (int i = 0; i < DB; ++i) {
          dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(dt, T::PB, qb * 32, s, i, lane), pf, dv[i], 0, 0, 0);  // dV^T += dO^T P
          dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qt, T::PB, qb * 32, s, i, lane), df, dk[i], 0, 0, 0);  // dK^T += Q^T dS
        }
      }
    }
  }
  bf16* out = a.dqkv;
  int ldo = a.ld;
  if (a.nsplit > 1) out = a.part + (int64_t)blockIdx.z * (gridDim.y / a.heads) * S * 3 * a.C, ldo = 3 * a.C;
  if (key < S) {
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int off = i * 32 + grp * 8 + h * 4;
        u32x2 wk = {pack2(dk[i][grp * 4], dk[i][grp * 4 + 1]), pack2(dk[i][grp * 4 + 2], dk[i][grp * 4 + 3])};
        u32x2 wv = {pack2(dv[i][grp * 4], dv[i][grp * 4 + 1]), pack2(dv[i][grp * 4 + 2], dv[i][grp * 4 + 3])};
        *(u32x2*)(out + (rb + key) * ldo + kc + off) = wk;
        *(u32x2*)(out + (rb + key) * ldo + vc + off) = wv;
      }
  }
}

// ------------------------------------------------------------------------------------------------ merging the splits
// forward: y = sum_i w_i O_i / sum_i w_i (+ resid), w_i = l_i 2^(m_i - M);  lse = M + log2(sum_i w_i).  One thread per 8 columns.
__global__ void __launch_bounds__(256) k_attn_merge_fwd(const bf16* __restrict__ part, const float* __restrict__ ml, const bf16* __restrict__ resid,
                                                        bf16* __restrict__ y, float* __restrict__ lse, int C, int heads, int S, int64_t rows,
                                                        int nsplit) {
  const int c8 = C / 8, D = C / heads;
  const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= rows * c8) return;
  const int64_t row = idx / c8;
  const int col = (int)(idx - row * c8) * 8, hd = col / D;
  const int64_t b = row / S, q = row - b * S, BH = rows / S * heads;
  float mi[8], li[8], M = -INFINITY;
  for (int z = 0; z < nsplit; ++z) {
    const float2 v = *(const float2*)(ml + ((z * BH + b * heads + hd) * S + q) * 2);
    mi[z] = v.x, li[z] = v.y;
    M = fmaxf(M, v.x);
  }
  float L = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < nsplit; ++z) {
    const float w = mi[z] == -INFINITY ? 0.f : li[z] * exp2f(mi[z] - M);
    L += w;
    const F8 p = unpack8(*(const u32x4*)(part + (z * rows + row) * C + col));
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += w * p.v[j];
  }
  const float inv = L > 0.f ? 1.f / L : 0.f;
  u32x4 rr = {0u, 0u, 0u, 0u};
  if (resid) rr = *(const u32x4*)(resid + row * C + col);
  const F8 r = unpack8(rr);
  F8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = acc[j] * inv + r.v[j];
  *(u32x4*)(y + row * C + col) = pack8(o);
  if (lse && col == hd * D) lse[(b * heads + hd) * S + q] = M + log2f(L);
}
// backward: dqkv[row][0..3C) = sum over splits of the partial gradients
__global__ void __launch_bounds__(256) k_attn_merge_bwd(const bf16* __restrict__ part, bf16* __restrict__ dqkv, int ld, int C3, int64_t rows,
                                                        int nsplit) {
  const int c8 = C3 / 8;
  const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= rows * c8) return;
  const int64_t row = idx / c8;
  const int col = (int)(idx - row * c8) * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < nsplit; ++z) {
    const F8 p = unpack8(*(const u32x4*)(part + (z * rows + row) * C3 + col));
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += p.v[j];
  }
  F8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
  *(u32x4*)(dqkv + row * ld + col) = pack8(o);
}

constexpr int kMaxSplit = 8;
// enough workgroups for ~2 per CU (8 waves; measured best at S = 4096), at least 4 tiles (256 rows of the reduction axis) per split
void pick_split(int B, int heads, int S, int& nsplit, int& tps) {
  const int base = (S + 127) / 128 * B * heads, ntiles = (S + 63) / 64;
  int want = 512 / (base > 0 ? base : 1);
  static const int env = [] { const char* e = getenv("MI_ATTN_SPLIT"); return e ? atoi(e) : 0; }();
  if (env > 0) want = env;
  want = want < 1 ? 1 : (want > kMaxSplit ? kMaxSplit : want);
  if (!env && want > ntiles / 4) want = ntiles / 4 > 0 ? ntiles / 4 : 1;
  tps = (ntiles + want - 1) / want;
  nsplit = (ntiles + tps - 1) / tps;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t fwd_ws(int C, int heads, int B, int S, int ns) { return ns > 1 ? align256((size_t)ns * B * S * C * 2) + (size_t)ns * B * heads * S * 8 : 0; }
size_t bwd_ws(int C, int B, int S, int ns) { return ns > 1 ? (size_t)ns * B * S * 3 * C * 2 : 0; }

bool bad(int C, int heads, int S, int B, int ld) {
  if (heads <= 0 || C % heads) return true;
  const int d = C / heads;
  return (d != 32 && d != 64) || B <= 0 || S <= 0 || (ld & 7) || ld < 3 * C;
}

}  // namespace

namespace mi_attn {
void attn_merge_fwd_launch(const AttnArgs& a, int B, hipStream_t st) {
  const int64_t rows = (int64_t)B * a.S, n = rows * (a.C / 8);
  hipLaunchKernelGGL(k_attn_merge_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, a.part_ml, a.resid, a.y, a.lse, a.C, a.heads, a.S,
                     rows, a.nsplit);
}
void attn_merge_bwd_launch(const AttnArgs& a, int B, hipStream_t st) {
  const int64_t rows = (int64_t)B * a.S, n = rows * (3 * a.C / 8);
  hipLaunchKernelGGL(k_attn_merge_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, a.dqkv, a.ld, 3 * a.C, rows, a.nsplit);
}
void attn_dsum_launch(const AttnArgs& a, const bf16* y, const bf16* resid, int B, hipStream_t st) {
  const int64_t rows = (int64_t)B * a.S;
  hipLaunchKernelGGL(k_attn_dsum, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, a.dy, y, resid, a.dsum, a.C, a.heads, a.S, rows);
}
}  // namespace mi_attn

extern "C" {

int mi_attn_supported(int C, int heads) {
  return heads > 0 && C % heads == 0 && (C / heads == 32 || C / heads == 64 || attnw_supported(C, heads));
}

int64_t mi_attn_workspace_bytes(int C, int heads, int B, int S) {
  if (attnw_supported(C, heads)) return attnw_workspace_bytes(C, heads, B, S);
  if (bad(C, heads, S, B, 3 * C)) return 0;
  int ns, tps;
  pick_split(B, heads, S, ns, tps);
  const size_t f = fwd_ws(C, heads, B, S, ns), b = bwd_ws(C, B, S, ns);
  return (int64_t)(f > b ? f : b);
}

int mi_attn_fwd(const void* qkv, int ld, int C, int heads, int B, int S, float scale, const void* resid, void* y, float* lse, void* ws,
                int64_t ws_bytes, hipStream_t st) {
  if (!qkv || !y) return MI_ERR_BAD_ARG;
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.ld = ld; a.C = C; a.heads = heads; a.S = S; a.scale = scale;
  a.resid = (const bf16*)resid; a.y = (bf16*)y; a.lse = lse;
  if (attnw_supported(C, heads)) return attnw_fwd(a, B, ws, ws_bytes, st);
  if (bad(C, heads, S, B, ld)) return MI_ERR_BAD_ARG;
  pick_split(B, heads, S, a.nsplit, a.tps);
  if (!ws || (size_t)ws_bytes < fwd_ws(C, heads, B, S, a.nsplit)) a.nsplit = 1, a.tps = (S + 63) / 64;  // no scratch: one pass over all keys
  a.part = (bf16*)ws;
  a.part_ml = (float*)((char*)ws + align256((size_t)a.nsplit * B * S * C * 2));
  dim3 grid((S + 127) / 128, B * heads, a.nsplit), blk(256);
  if (C / heads == 64) hipLaunchKernelGGL(k_attn_fwd<64>, grid, blk, 0, st, a);
  else hipLaunchKernelGGL(k_attn_fwd<32>, grid, blk, 0, st, a);
  if (a.nsplit > 1) attn_merge_fwd_launch(a, B, st);
  MI_CHECK_LAUNCH();
  return 0;
}

// dy: gradient w.r.t. the attention output (same tensor as the gradient of y); y, resid: forward output and its residual input
// (O = y - resid is what the row term needs); dqkv receives dQ | dK | dV in the layout of qkv.  dsum: [B*heads][S] scratch.
int mi_attn_bwd(const void* qkv, int ld, int C, int heads, int B, int S, float scale, const void* y, const void* resid, const void* dy,
                const float* lse, float* dsum, void* dqkv, void* ws, int64_t ws_bytes, hipStream_t st) {
  if (!qkv || !y || !dy || !lse || !dsum || !dqkv) return MI_ERR_BAD_ARG;
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.ld = ld; a.C = C; a.heads = heads; a.S = S; a.scale = scale;
  a.dy = (const bf16*)dy; a.lse = const_cast<float*>(lse); a.dsum = dsum; a.dqkv = (bf16*)dqkv;
  if (attnw_supported(C, heads)) return attnw_bwd(a, B, (const bf16*)y, (const bf16*)resid, ws, ws_bytes, st);
  if (bad(C, heads, S, B, ld)) return MI_ERR_BAD_ARG;
  pick_split(B, heads, S, a.nsplit, a.tps);
  if (!ws || (size_t)ws_bytes < bwd_ws(C, B, S, a.nsplit)) a.nsplit = 1, a.tps = (S + 63) / 64;
  a.part = (bf16*)ws;
  attn_dsum_launch(a, (const bf16*)y, (const bf16*)resid, B, st);
  dim3 grid((S + 127) / 128, B * heads, a.nsplit), blk(256);
  if (C / heads == 64) {
    hipLaunchKernelGGL(k_attn_bwd_dq<64>, grid, blk, 0, st, a);
    hipLaunchKernelGGL(k_attn_bwd_dkv<64>, grid, blk, 0, st, a);
  } else {
    hipLaunchKernelGGL(k_attn_bwd_dq<32>, grid, blk, 0, st, a);
    hipLaunchKernelGGL(k_attn_bwd_dkv<32>, grid, blk, 0, st, a);
  }
  if (a.nsplit > 1) attn_merge_bwd_launch(a, B, st);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
