// probe: wall-clock TFLOP/s of an LDS-fed v_mfma_f32_32x32x16_bf16 loop (one wave per SIMD, 256 CUs, random bf16 data, every operand
// fragment read from LDS with ds_read_b128) as a function of the per-wave register tile VB voxel blocks x CB cout blocks:
// LDS bytes per MFMA = (VB + CB) / (VB * CB) KiB.  The ceiling any LDS-fed conv kernel of that tile shape can reach.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int VB, int CB, int PIPE>
__global__ void __launch_bounds__(256, 1) k(const unsigned* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned*)lds)[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = (const u32x4*)(lds + wave * 16384) + lane;
  f32x16 acc[VB][CB] = {};
  // software-pipelined: the fragments of iteration it + 1 are requested before the MFMAs of iteration it (PIPE = 1), as a hand-
  // scheduled kernel does; PIPE = 0 is the naive loop (load, wait, compute), which measures LDS latency, not LDS bandwidth
  u32x4 b[VB], a[CB], nb[VB], na[CB];
#pragma unroll
  for (int v = 0; v < VB; ++v) b[v] = base[64 * v];
#pragma unroll
  for (int c = 0; c < CB; ++c) a[c] = base[64 * (VB + c)];
  for (int it = 0; it < iters; ++it) {
    const u32x4* p = base + ((it + 1) & 1) * 64;
    if (PIPE) {
#pragma unroll
      for (int v = 0; v < VB; ++v) nb[v] = p[64 * v];
#pragma unroll
      for (int c = 0; c < CB; ++c) na[c] = p[64 * (VB + c)];
    }
#pragma unroll
    for (int v = 0; v < VB; ++v)
#pragma unroll
      for (int c = 0; c < CB; ++c)
        acc[v][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[c]), __builtin_bit_cast(bf16x8, b[v]), acc[v][c], 0, 0, 0);
    if (!PIPE) {
#pragma unroll
      for (int v = 0; v < VB; ++v) nb[v] = p[64 * v];
#pragma unroll
      for (int c = 0; c < CB; ++c) na[c] = p[64 * (VB + c)];
    }
#pragma unroll
    for (int v = 0; v < VB; ++v) b[v] = nb[v];
#pragma unroll
    for (int c = 0; c < CB; ++c) a[c] = na[c];
  }
  float total = 0.f;
  for (int v = 0; v < VB; ++v) for (int c = 0; c < CB; ++c) for (int e = 0; e < 16; ++e) total += acc[v][c][e];
  if (total == 1.2345f) sink[0] = total;
}

template <int VB, int CB, int PIPE>
void run(const unsigned* src, float* sink) {
  const int iters = 400000 / (VB * CB);
  hipFuncSetAttribute((const void*)k<VB, CB, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<VB, CB, PIPE>), dim3(256), dim3(256), 65536, 0, src, sink, iters);
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<VB, CB, PIPE>), dim3(256), dim3(256), 65536, 0, src, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 4 * (double)iters * VB * CB * 2.0 * 32 * 32 * 16 * reps;
  printf("VB %d x CB %d %s: %.2f KiB LDS per MFMA -> %.0f TFLOP/s\n", VB, CB, PIPE ? "pipelined" : "naive    ", (VB + CB) / (double)(VB * CB),
         flops / (ms * 1e-3) / 1e12);
}
int main() {
  std::vector<unsigned> h(16384);
  srand(1);
  for (auto& v : h) {
    auto r = [] { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; __builtin_memcpy(&u, &f, 4); return u >> 16; };
    v = r() | (r() << 16);
  }
  unsigned* src; float* sink;
  hipMalloc(&src, 65536); hipMalloc(&sink, 4);
  hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {
    run<2, 1, 0>(src, sink); run<2, 1, 1>(src, sink); run<2, 2, 0>(src, sink); run<2, 2, 1>(src, sink); run<4, 1, 1>(src, sink);
    run<4, 2, 1>(src, sink); run<4, 4, 1>(src, sink); run<8, 1, 1>(src, sink);
  }
  return 0;
}
