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
#include "common.h"
#include "medimgen_hip.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;

struct AttnArgs {
  const bf16* qkv;   // [B*S][ld]
  int ld, C, heads, S;
  float scale;
  const bf16* resid;  // forward: x (added to the output) or null
  bf16* y;            // forward: [B*S][C]
  float* lse;         // [B*heads][S], log2 domain: m + log2(l)
  // backward
  const bf16* dy;     // [B*S][C] gradient of the attention output (the residual branch is handled by the caller)
  const bf16* o;      // [B*S][C] forward output WITHOUT residual is not stored; o = y - x is recomputed from y and resid
  float* dsum;        // [B*heads][S]  D = rowsum(dO * O)
  bf16* dqkv;         // [B*S][ld]
};

// 16 bytes = 8 bf16 of row `row` at element offset `col` (ld in elements); zero when row >= limit
__device__ __forceinline__ u32x4 ld16(const bf16* base, int64_t row, int limit, int ld, int col) {
  u32x4 z = {0u, 0u, 0u, 0u};
  return row < limit ? *(const u32x4*)(base + row * ld + col) : z;
}

// transposed LDS read producing the A fragment of X^T for the permuted k order of an accumulator-as-B-operand product:
// element j of lane half h <- row 16*s + 8*(j>>2) + 4*h + (j&3), column (lane & 31) of the 32-column block `cblk`
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int row0, int s, int cblk, int lane) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = lane >> 5;
  const char* a = tile + (row0 + 16 * s + 4 * h + q) * pitch + (cblk * 32 + (gq & 1) * 16 + p * 4) * 2;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)a);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(a + 8 * pitch));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
// rows-as-A fragment: lane (r, h) reads 16 bytes of row r: columns ks*16 + 8h ..
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int pitch, int row0, int ks, int lane) {
  return *(const bf16x8*)(tile + (row0 + (lane & 31)) * pitch + ks * 32 + (lane >> 5) * 16);
}
__device__ __forceinline__ bf16x8 pack_acc8(const f32x16& a, int s) {  // registers 8s..8s+7 -> bf16x8 (k-step s as B operand)
  u32x4 r = {pack2(a[8 * s], a[8 * s + 1]), pack2(a[8 * s + 2], a[8 * s + 3]), pack2(a[8 * s + 4], a[8 * s + 5]),
             pack2(a[8 * s + 6], a[8 * s + 7])};
  return __builtin_bit_cast(bf16x8, r);
}

template <int D>
struct Tiles {  // LDS images of a 64-row tile of two operands
  static constexpr int PA = D * 2 + 16;             // row pitch for b128 row fragments (16 * odd)
  static constexpr int PB = D == 64 ? 192 : 64;     // row pitch for transposed reads (4 rows hit distinct 64-B bank groups)
  static constexpr int PIECES = 64 * D / 8 / 256;   // 16-byte pieces per thread per operand
};

// stage a 64-row tile [rows r0.., D cols at column offset col] into LDS with pitch P
template <int D>
__device__ __forceinline__ void tile_load(u32x4 (&reg)[Tiles<D>::PIECES], const bf16* base, int64_t row_base, int r0, int limit, int ld,
                                          int col) {
#pragma unroll
  for (int i = 0; i < Tiles<D>::PIECES; ++i) {
    int pc = threadIdx.x + 256 * i;
    int r = pc / (D / 8), c = (pc % (D / 8)) * 8;
    reg[i] = ld16(base, row_base + r0 + r, row_base + limit, ld, col + c);
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

  u32x4 kr[T::PIECES], vr[T::PIECES];
  tile_load<D>(kr, a.qkv, rb, 0, S, a.ld, kc);
  tile_load<D>(vr, a.qkv, rb, 0, S, a.ld, vc);
  for (int k0 = 0; k0 < S; k0 += 64) {
    __syncthreads();
    tile_store<D>(kr, kt, T::PA);
    tile_store<D>(vr, vt, T::PB);
    __syncthreads();
    if (k0 + 64 < S) {
      tile_load<D>(kr, a.qkv, rb, k0 + 64, S, a.ld, kc);
      tile_load<D>(vr, a.qkv, rb, k0 + 64, S, a.ld, vc);
    }
    // S^T tile: 64 keys (2 blocks of 32) x 32 queries
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[kb][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, T::PA, kb * 32, ks, lane), qf[ks], st[kb], 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float t = key < S ? st[kb][e] * c : -INFINITY;
        st[kb][e] = t;
        tmax = fmaxf(tmax, t);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));  // the two lane halves hold the other keys of the same query
    const float mn = fmaxf(m, tmax);
    const float alpha = mn == -INFINITY ? 1.f : exp2f(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = mn == -INFINITY ? 0.f : exp2f(st[kb][e] - mn);
        st[kb][e] = p;
        ps += p;
      }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int i = 0; i < DVB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
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

  u32x4 kr[T::PIECES], vr[T::PIECES];
  tile_load<D>(kr, a.qkv, rb, 0, S, a.ld, kc);
  tile_load<D>(vr, a.qkv, rb, 0, S, a.ld, vc);
  for (int k0 = 0; k0 < S; k0 += 64) {
    __syncthreads();
    tile_store<D>(kr, kt, T::PB);
    tile_store<D>(vr, vt, T::PA);
    __syncthreads();
    if (k0 + 64 < S) {
      tile_load<D>(kr, a.qkv, rb, k0 + 64, S, a.ld, kc);
      tile_load<D>(vr, a.qkv, rb, k0 + 64, S, a.ld, vc);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 st, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, T::PB, kb * 32, ks, lane), qf[ks], st, 0, 0, 0);   // S^T  = K Q^T
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vt, T::PA, kb * 32, ks, lane), dof[ks], dp, 0, 0, 0);  // dP^T = V dO^T
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const float p = key < S ? exp2f(st[e] * c - lse) : 0.f;
        st[e] = p * (dp[e] - dq_row) * a.scale;  // dS^T
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 df = pack_acc8(st, s);
#pragma unroll
        for (int i = 0; i < DB; ++i) dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(kt, T::PB, kb * 32, s, i, lane), df, dq[i], 0, 0, 0);
      }
    }
  }
  if (q < S) {
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int col = qc + i * 32 + grp * 8 + h * 4;
        u32x2 w = {pack2(dq[i][grp * 4], dq[i][grp * 4 + 1]), pack2(dq[i][grp * 4 + 2], dq[i][grp * 4 + 3])};
        *(u32x2*)(a.dqkv + (rb + q) * a.ld + col) = w;
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

  u32x4 qr[T::PIECES], dr[T::PIECES];
  tile_load<D>(qr, a.qkv, rb, 0, S, a.ld, qc);
  tile_load<D>(dr, a.dy, rb, 0, S, a.C, hd * D);
  for (int q0 = 0; q0 < S; q0 += 64) {
    __syncthreads();
    tile_store<D>(qr, qt, T::PB);
    tile_store<D>(dr, dt, T::PB);
    if (threadIdx.x < 64) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < S ? a.lse[(int64_t)bh * S + qq] : INFINITY;  // +inf -> p = 0 for padded queries
      dsum_s[threadIdx.x] = qq < S ? a.dsum[(int64_t)bh * S + qq] : 0.f;
    }
    __syncthreads();
    if (q0 + 64 < S) {
      tile_load<D>(qr, a.qkv, rb, q0 + 64, S, a.ld, qc);
      tile_load<D>(dr, a.dy, rb, q0 + 64, S, a.C, hd * D);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 st, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qt, T::PB, qb * 32, ks, lane), kf[ks], st, 0, 0, 0);  // S  = Q K^T
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dt, T::PB, qb * 32, ks, lane), vf[ks], dp, 0, 0, 0);  // dP = dO V^T
      }
      f32x16 pr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ql = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const float p = key < S ? exp2f(st[e] * c - lse_s[ql]) : 0.f;
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
  if (key < S) {
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int off = i * 32 + grp * 8 + h * 4;
        u32x2 wk = {pack2(dk[i][grp * 4], dk[i][grp * 4 + 1]), pack2(dk[i][grp * 4 + 2], dk[i][grp * 4 + 3])};
        u32x2 wv = {pack2(dv[i][grp * 4], dv[i][grp * 4 + 1]), pack2(dv[i][grp * 4 + 2], dv[i][grp * 4 + 3])};
        *(u32x2*)(a.dqkv + (rb + key) * a.ld + kc + off) = wk;
        *(u32x2*)(a.dqkv + (rb + key) * a.ld + vc + off) = wv;
      }
  }
}

bool bad(int C, int heads, int S, int B, int ld) {
  if (heads <= 0 || C % heads) return true;
  const int d = C / heads;
  return (d != 32 && d != 64) || B <= 0 || S <= 0 || (ld & 7) || ld < 3 * C;
}

}  // namespace

extern "C" {

int mi_attn_supported(int C, int heads) { return heads > 0 && C % heads == 0 && (C / heads == 32 || C / heads == 64); }

int mi_attn_fwd(const void* qkv, int ld, int C, int heads, int B, int S, float scale, const void* resid, void* y, float* lse,
                hipStream_t st) {
  if (bad(C, heads, S, B, ld) || !qkv || !y) return MI_ERR_BAD_ARG;
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.ld = ld; a.C = C; a.heads = heads; a.S = S; a.scale = scale;
  a.resid = (const bf16*)resid; a.y = (bf16*)y; a.lse = lse;
  dim3 grid((S + 127) / 128, B * heads), blk(256);
  if (C / heads == 64) hipLaunchKernelGGL(k_attn_fwd<64>, grid, blk, 0, st, a);
  else hipLaunchKernelGGL(k_attn_fwd<32>, grid, blk, 0, st, a);
  MI_CHECK_LAUNCH();
  return 0;
}

// dy: gradient w.r.t. the attention output (same tensor as the gradient of y); y, resid: forward output and its residual input
// (O = y - resid is what the row term needs); dqkv receives dQ | dK | dV in the layout of qkv.  dsum: [B*heads][S] scratch.
int mi_attn_bwd(const void* qkv, int ld, int C, int heads, int B, int S, float scale, const void* y, const void* resid, const void* dy,
                const float* lse, float* dsum, void* dqkv, hipStream_t st) {
  if (bad(C, heads, S, B, ld) || !qkv || !y || !dy || !lse || !dsum || !dqkv) return MI_ERR_BAD_ARG;
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.ld = ld; a.C = C; a.heads = heads; a.S = S; a.scale = scale;
  a.dy = (const bf16*)dy; a.lse = const_cast<float*>(lse); a.dsum = dsum; a.dqkv = (bf16*)dqkv;
  const int64_t rows = (int64_t)B * S;
  hipLaunchKernelGGL(k_attn_dsum, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, (const bf16*)dy, (const bf16*)y, (const bf16*)resid, dsum, C,
                     heads, S, rows);
  dim3 grid((S + 127) / 128, B * heads), blk(256);
  if (C / heads == 64) {
    hipLaunchKernelGGL(k_attn_bwd_dq<64>, grid, blk, 0, st, a);
    hipLaunchKernelGGL(k_attn_bwd_dkv<64>, grid, blk, 0, st, a);
  } else {
    hipLaunchKernelGGL(k_attn_bwd_dq<32>, grid, blk, 0, st, a);
    hipLaunchKernelGGL(k_attn_bwd_dkv<32>, grid, blk, 0, st, a);
  }
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
