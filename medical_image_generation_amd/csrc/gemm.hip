// Batched bf16 "NT" GEMM on MFMA 32x32x16 + the small kernels the attention block needs around it
// (tiled transpose, row softmax forward / backward).  Used by nn.Linear-shaped work: q/k/v projections, the time-embedding
// MLP, time_emb_proj, and the materialised attention products (AttentionBlock._attention, UNet:406-416).
//
//   C[z][m][n] = alpha * sum_k A[z][m][k] * B[z][n][k]  (+ bias[n]) (+ R[z][m][n])  (+ C_old when accumulate)
//
// 128x128x32 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles; A and B staged through padded LDS rows
// (80-byte pitch: conflict-free ds_read_b128 for the 32x32x16 operand maps) with register prefetch of the next K tile.
#include "common.h"
#include "medimgen_hip.h"

namespace {

constexpr int BK = 32, PITCH = 40;  // PITCH in bf16 elements (80 bytes)

struct GemmArgs {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* bias;
  const bf16* R;
  int M, N, K, lda, ldb, ldc, ldr;
  int Z2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
  float alpha;
  int out_f32, accumulate;
};

// WT = MFMA tiles per wave and dimension: 2 -> 128x128 block tile, 1 -> 64x64 (projections of a 16^3 level, M = 4096, N = 256:
// 64 workgroups of 128x128 would leave three quarters of the chip idle)
template <int WT>
__global__ void __launch_bounds__(256) k_gemm_nt(GemmArgs p) {
  constexpr int BM = 64 * WT, BN = 64 * WT;
  __shared__ __attribute__((aligned(16))) bf16 lds[2][2][BM * PITCH];  // [buf][A|B][row][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int z = blockIdx.z, z1 = z / p.Z2, z2 = z % p.Z2;
  const bf16* A = p.A + z1 * p.sA1 + z2 * p.sA2;
  const bf16* B = p.B + z1 * p.sB1 + z2 * p.sB2;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // staging map: 256 * WT 16-byte pieces per operand tile, WT per thread
  int prow[WT], pk[WT];
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    int pc = tid + 256 * i;
    prow[i] = pc >> 2;
    pk[i] = (pc & 3) * 8;
  }
  u32x4 ra[WT], rb[WT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < WT; ++i) {
      u32x4 zero = {0u, 0u, 0u, 0u};
      int k = k0 + pk[i];
      int am = m0 + prow[i], bn = n0 + prow[i];
      ra[i] = (am < p.M && k < p.K) ? *(const u32x4*)(A + (int64_t)am * p.lda + k) : zero;
      rb[i] = (bn < p.N && k < p.K) ? *(const u32x4*)(B + (int64_t)bn * p.ldb + k) : zero;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WT; ++i) {
      *(u32x4*)(&lds[buf][0][prow[i] * PITCH + pk[i]]) = ra[i];
      *(u32x4*)(&lds[buf][1][prow[i] * PITCH + pk[i]]) = rb[i];
    }
  };

  f32x16 acc[WT][WT];
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[WT], fb[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        fa[i] = *(const bf16x8*)(&lds[buf][0][(wm * 32 * WT + i * 32 + r) * PITCH + ks * 16 + h * 8]);
        fb[i] = *(const bf16x8*)(&lds[buf][1][(wn * 32 * WT + i * 32 + r) * PITCH + ks * 16 + h * 8]);
      }
#pragma unroll
      for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue. D map (32x32x16): col = lane & 31 -> n, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) -> m
  const int64_t coff = z1 * p.sC1 + z2 * p.sC2;
  const int64_t roff = z1 * p.sR1 + z2 * p.sR2;
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b) {
      const int n = n0 + wn * 32 * WT + b * 32 + r;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 32 * WT + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[a][b][e] * p.alpha + bv;
        if (p.R) v += bf2f(p.R[roff + (int64_t)m * p.ldr + n]);
        const int64_t ci = coff + (int64_t)m * p.ldc + n;
        if (p.out_f32) {
          float* c = (float*)p.C;
          c[ci] = p.accumulate ? c[ci] + v : v;
        } else {
          ((bf16*)p.C)[ci] = f2bf(v);
        }
      }
    }
}

// ------------------------------------------------------------------ batched 2-D transpose (bf16), 64x64 tiles
// Output rows are zero-filled from column R up to the next multiple of 8 that fits the pitch (R itself becomes the K axis of a GEMM,
// whose 16-byte operand loads need K % 8 == 0: ragged token counts such as 3^3 = 27).
__global__ void __launch_bounds__(256) k_transpose(const bf16* __restrict__ in, bf16* __restrict__ out, int R, int Cc, int ld_in,
                                                   int ld_out, int Z2, int64_t si1, int64_t si2, int64_t so1, int64_t so2) {
  __shared__ bf16 t[64][66];
  const int z = blockIdx.z, z1 = z / Z2, z2 = z % Z2;
  in += z1 * si1 + z2 * si2;
  out += z1 * so1 + z2 * so2;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int rr = i >> 6, cc = i & 63;
    t[rr][cc] = (r0 + rr < R && c0 + cc < Cc) ? in[(int64_t)(r0 + rr) * ld_in + c0 + cc] : f2bf(0.f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int cc = i >> 6, rr = i & 63;
    if (c0 + cc < Cc && r0 + rr < R) out[(int64_t)(c0 + cc) * ld_out + r0 + rr] = t[rr][cc];
    else if (c0 + cc < Cc && r0 + rr < min((R + 7) & ~7, ld_out)) out[(int64_t)(c0 + cc) * ld_out + r0 + rr] = f2bf(0.f);
  }
}

// ------------------------------------------------------------------ row softmax: fp32 scores -> bf16 probabilities
// ldp: row pitch of the bf16 matrices (>= cols); columns [cols, ldp) are written as zeros (K padding of the products that follow)
__global__ void __launch_bounds__(256) k_softmax_fwd(const float* __restrict__ s, bf16* __restrict__ pr, int cols, int ldp) {
  __shared__ float red[4];
  const float* row = s + (int64_t)blockIdx.x * cols;
  bf16* o = pr + (int64_t)blockIdx.x * ldp;
  float mx = -3.0e38f;
  for (int j = threadIdx.x; j < cols; j += 256) mx = fmaxf(mx, row[j]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) sum += __expf(row[j] - mx);
  sum = block_sum_256(sum, red);
  const float inv = 1.0f / sum;
  for (int j = threadIdx.x; j < cols; j += 256) o[j] = f2bf(__expf(row[j] - mx) * inv);
  for (int j = cols + threadIdx.x; j < ldp; j += 256) o[j] = f2bf(0.f);
}
// dS = P * (dP - sum_k dP_k P_k) * scale
__global__ void __launch_bounds__(256) k_softmax_bwd(const bf16* __restrict__ pr, const float* __restrict__ dp, bf16* __restrict__ ds,
                                                     int cols, int ldp, float scale) {
  __shared__ float red[4];
  const bf16* prow = pr + (int64_t)blockIdx.x * ldp;
  const float* drow = dp + (int64_t)blockIdx.x * cols;
  bf16* o = ds + (int64_t)blockIdx.x * ldp;
  float dot = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) dot += bf2f(prow[j]) * drow[j];
  dot = block_sum_256(dot, red);
  for (int j = threadIdx.x; j < cols; j += 256) o[j] = f2bf(bf2f(prow[j]) * (drow[j] - dot) * scale);
  for (int j = cols + threadIdx.x; j < ldp; j += 256) o[j] = f2bf(0.f);
}

}  // namespace

extern "C" {

int mi_gemm_nt_bf16(const void* A, int lda, int64_t sA1, int64_t sA2, const void* B, int ldb, int64_t sB1, int64_t sB2, void* C, int ldc,
                    int64_t sC1, int64_t sC2, const float* bias, const void* R, int ldr, int64_t sR1, int64_t sR2, int M, int N, int K,
                    int Z, int Z2, float alpha, int out_f32, int accumulate, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || Z <= 0 || Z2 <= 0 || Z % Z2) return MI_ERR_BAD_ARG;
  if ((K & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return MI_ERR_BAD_ARG;
  if ((sA1 | sA2 | sB1 | sB2) & 7) return MI_ERR_BAD_ARG;
  if (accumulate && !out_f32) return MI_ERR_BAD_ARG;
  GemmArgs p;
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.bias = bias; p.R = (const bf16*)R;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.Z2 = Z2; p.sA1 = sA1; p.sA2 = sA2; p.sB1 = sB1; p.sB2 = sB2; p.sC1 = sC1; p.sC2 = sC2; p.sR1 = sR1; p.sR2 = sR2;
  p.alpha = alpha; p.out_f32 = out_f32; p.accumulate = accumulate;
  if ((int64_t)ceil_div(N, 128) * ceil_div(M, 128) * Z < 256)  // fewer workgroups than CUs: quarter-size tiles
    hipLaunchKernelGGL(k_gemm_nt<1>, dim3(ceil_div(N, 64), ceil_div(M, 64), Z), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(k_gemm_nt<2>, dim3(ceil_div(N, 128), ceil_div(M, 128), Z), dim3(256), 0, st, p);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_transpose_bf16(const void* in, int ld_in, int64_t si1, int64_t si2, void* out, int ld_out, int64_t so1, int64_t so2, int R, int Cc,
                      int Z, int Z2, hipStream_t st) {
  if (R <= 0 || Cc <= 0 || Z <= 0 || Z2 <= 0 || Z % Z2) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_transpose, dim3(ceil_div(Cc, 64), ceil_div(R, 64), Z), dim3(256), 0, st, (const bf16*)in, (bf16*)out, R, Cc, ld_in,
                     ld_out, Z2, si1, si2, so1, so2);
  MI_CHECK_LAUNCH();
  return 0;
}

int mi_softmax_fwd(const float* scores, void* probs, int64_t rows, int cols, int ld_probs, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || rows > 2147483647LL || ld_probs < cols) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_softmax_fwd, dim3((int)rows), dim3(256), 0, st, scores, (bf16*)probs, cols, ld_probs);
  MI_CHECK_LAUNCH();
  return 0;
}
int mi_softmax_bwd(const void* probs, const float* dprobs, void* dscores, int64_t rows, int cols, int ld_probs, float scale,
                   hipStream_t st) {
  if (rows <= 0 || cols <= 0 || rows > 2147483647LL || ld_probs < cols) return MI_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_softmax_bwd, dim3((int)rows), dim3(256), 0, st, (const bf16*)probs, dprobs, (bf16*)dscores, cols, ld_probs, scale);
  MI_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
