// probe: wall-clock TFLOP/s of an LDS-fed MFMA loop in the regime of k_conv27 (one compute wave per SIMD, every operand fragment
// read from LDS by ds_read_b128, 6 KiB of fragment reads per 64 voxels x 32 cout x 32 cin) for the two bf16 MFMA shapes:
//   A: 4 x v_mfma_f32_32x32x16_bf16  (2 voxel blocks of 32 x 1 cout block of 32 x 2 k-steps of 16)
//   B: 8 x v_mfma_f32_16x16x32_bf16  (4 voxel blocks of 16 x 2 cout blocks of 16 x 1 k-step of 32)
// Same flops, same LDS bytes, random bf16 data.  MI355X_MICROARCH.md (DVFS give-back 7) reports B 1.12-1.14x faster in wall time
// at equal cycles; this checks it on this pool's boxes before a kernel is rebuilt around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ void __launch_bounds__(256, 1) k(const unsigned* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned*)lds)[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = (const u32x4*)(lds + wave * 16384) + lane;
  float total = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[2] = {};
    for (int it = 0; it < iters; ++it) {
      const u32x4* p = base + (it & 3) * 64;  // 4 KiB windows: conflict-free lane-linear fragments
      u32x4 b00 = p[0], b01 = p[64], a0 = p[128], b10 = p[192], b11 = p[256], a1 = p[320];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b00), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b01), acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b10), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b11), acc[1], 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) total += acc[0][e] + acc[1][e];
  } else {
    f32x4 acc[4][2] = {};
    for (int it = 0; it < iters; ++it) {
      const u32x4* p = base + (it & 3) * 64;
      u32x4 b0 = p[0], b1 = p[64], b2 = p[128], b3 = p[192], a0 = p[256], a1 = p[320];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc[1][1], 0, 0, 0);
      acc[2][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b2), acc[2][0], 0, 0, 0);
      acc[2][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b2), acc[2][1], 0, 0, 0);
      acc[3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b3), acc[3][0], 0, 0, 0);
      acc[3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b3), acc[3][1], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) total += acc[i][j][e];
  }
  if (total == 1.2345f) sink[0] = total;
}

template <int SHAPE>
double run(const unsigned* src, float* sink, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(256), 65536, 0, src, sink, iters);
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(256), 65536, 0, src, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 16 * reps;
  return flops / (ms * 1e-3) / 1e12;
}
int main() {
  std::vector<unsigned> h(16384);
  srand(1);
  for (auto& v : h) {  // two random bf16 in [-2, 2) per dword (random sign, exponent, mantissa)
    auto r = [] { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; __builtin_memcpy(&u, &f, 4); return u >> 16; };
    v = r() | (r() << 16);
  }
  unsigned* src; float* sink;
  hipMalloc(&src, 65536); hipMalloc(&sink, 4);
  hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 200000;  // ~50 ms per launch
  for (int round = 0; round < 3; ++round) {
    double a = run<0>(src, sink, iters), b = run<1>(src, sink, iters);
    printf("round %d: 32x32x16 %.0f TFLOP/s | 16x16x32 %.0f TFLOP/s | ratio %.3f\n", round, a, b, b / a);
  }
  return 0;
}
