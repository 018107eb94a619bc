// Shared device helpers for the medimgen gfx950 kernels.  CDNA4 only (wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MI_WAVE 64

#define MI_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)



__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving

// 8 bf16 <-> 8 floats through one 16-byte access
struct F8 {
  float v[8];
};
__device__ __forceinline__ F8 unpack8(u32x4 raw) {
  F8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __uint_as_float(raw[i] << 16);
    r.v[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
  }
  return r;
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  bf16x2 p;
  p[0] = (bf16)lo;
  p[1] = (bf16)hi;
  return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ u32x4 pack8(const F8& f) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack2(f.v[2 * i], f.v[2 * i + 1]);
  return r;
}

// sigmoid through v_exp_f32 / v_rcp_f32 (1 ulp each): an IEEE divide costs ~10 VALU instructions and a branch per element, which
// made the streaming GroupNorm kernels VALU-bound; exp2(+large) = inf -> rcp = 0 and exp2(-large) = 0 -> rcp(1) = 1 are the right limits
__device__ __forceinline__ float sigmoid_f(float u) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u)); }
__device__ __forceinline__ float silu_f(float u) { return u * sigmoid_f(u); }
// d/du [u * sigmoid(u)] = s * (1 + u * (1 - s))
__device__ __forceinline__ float silu_grad_f(float u) {
  float s = sigmoid_f(u);
  return s * (1.0f + u * (1.0f - s));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS. Result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Zero-fill as a KERNEL, never hipMemset[2D]Async: the train step is captured into hipGraphs, and on ROCm 7.2 a memset node of a
// graph that has been replayed stops zeroing its target once ANOTHER graph is captured in the process (measured: the bias-gradient
// slab of the captured step kept stale bytes -> inf gradient norm -> every later optimizer step clipped to nothing; tests/
// test_trainer_gpu.py::test_graph_survives_other_shape_forward).  A kernel node has no such state.
static __global__ void k_zero_f32_2d(float* __restrict__ x, int64_t ld, int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    x[r * ld + (i - r * cols)] = 0.f;
  }
}
static inline hipError_t mi_zero_fill_f32(float* x, int64_t ld, int rows, int cols, hipStream_t st) {
  const int64_t total = (int64_t)rows * cols;
  int grid = (int)((total + 255) / 256);
  grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
  hipLaunchKernelGGL(k_zero_f32_2d, dim3(grid), dim3(256), 0, st, x, ld, rows, cols);
  return hipGetLastError();
}

// Ablation knobs (MI_C27_DBG, MI_CPH_DBG, MI_WGRAD_DBG, MI_IGEMM_DBG) switch parts of a kernel off and make it compute GARBAGE: they are
// read in the diagnostic builds only (`make diag` / `make phdiag`: -DMI_DIAG_KNOBS); the shipped library ignores them.
#include <stdlib.h>
static inline int mi_diag_knob(const char* name) {
#ifdef MI_DIAG_KNOBS
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
#else
  (void)name;
  return 0;
#endif
}
