// Shared by attention.hip (head dims 32 / 64) and attention_wide.hip (one head of 512 / 768): kernel arguments, the fragment-read idioms
// of the "swapped" products (see attention.hip) and the launchers of the split-merge kernels.
#pragma once
#include "common.h"
#include "medimgen_hip.h"

namespace mi_attn {

static constexpr float kLog2e = 1.4426950408889634f;
static constexpr f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
// v_exp_f32 without exp2f()'s denormal-range scaling: arguments here are <= 0 and results below 2^-126 may flush to zero
static __device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

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
  // split over the reduction axis (keys in forward / dQ, queries in dK,dV): blockIdx.z handles `tps` 64-row tiles and writes
  // a partial result; k_attn_merge_* combine them.  nsplit == 1: results go straight to y / dqkv.
  int nsplit, tps;
  bf16* part;         // forward: [nsplit][B*S][C] normalised partial outputs; backward: [nsplit][B*S][3C] partial dQ | dK | dV
  float* part_ml;     // forward: [nsplit][B*heads][S][2] running max (log2 domain) and sum of each partial
};

// 16 bytes = 8 bf16 of row `row` at element offset `col` (ld in elements); zero when row >= limit
static __device__ __forceinline__ u32x4 ld16(const bf16* base, int64_t row, int limit, int ld, int col) {
  u32x4 z = {0u, 0u, 0u, 0u};
  return row < limit ? *(const u32x4*)(base + row * ld + col) : z;
}

// transposed LDS read producing the A fragment of X^T for the permuted k order of an accumulator-as-B-operand product:
// element j of lane half h <- row 16*s + 8*(j>>2) + 4*h + (j&3), column (lane & 31) of the 32-column block `cblk`
static __device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int row0, int s, int cblk, int lane) {
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
static __device__ __forceinline__ bf16x8 row_frag(const char* tile, int pitch, int row0, int ks, int lane) {
  return *(const bf16x8*)(tile + (row0 + (lane & 31)) * pitch + ks * 32 + (lane >> 5) * 16);
}
static __device__ __forceinline__ bf16x8 pack_acc8(const f32x16& a, int s) {  // registers 8s..8s+7 -> bf16x8 (k-step s as B operand)
  u32x4 r = {pack2(a[8 * s], a[8 * s + 1]), pack2(a[8 * s + 2], a[8 * s + 3]), pack2(a[8 * s + 4], a[8 * s + 5]),
             pack2(a[8 * s + 6], a[8 * s + 7])};
  return __builtin_bit_cast(bf16x8, r);
}


// attention.hip: combine the partial results of the split kernels (forward: normalised partial outputs + running max / sum)
void attn_merge_fwd_launch(const AttnArgs& a, int B, hipStream_t st);
void attn_merge_bwd_launch(const AttnArgs& a, int B, hipStream_t st);
void attn_dsum_launch(const AttnArgs& a, const bf16* y, const bf16* resid, int B, hipStream_t st);
// attention_wide.hip
bool attnw_supported(int C, int heads);
int64_t attnw_workspace_bytes(int C, int heads, int B, int S);
int attnw_fwd(AttnArgs a, int B, void* ws, int64_t ws_bytes, hipStream_t st);
int attnw_bwd(AttnArgs a, int B, const bf16* y, const bf16* resid, void* ws, int64_t ws_bytes, hipStream_t st);

}  // namespace mi_attn
