// probe: ds_read_b128 issue/return rate per wave with N reads allowed in flight (counted lgkmcnt), 4 waves per CU, 256 CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int INFLIGHT, int STRIDE>
__global__ void __launch_bounds__(256, 1) k(unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((unsigned*)lds)[i] = i;
  __syncthreads();
  typedef __attribute__((address_space(3))) char lc;
  unsigned addr = (unsigned)(size_t)(lc*)lds + (threadIdx.x & 63) * STRIDE + (threadIdx.x >> 6) * 8192;
  u32x4 r[8];
  unsigned acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r[j]) : "i"(INFLIGHT));
      acc += r[j][0];
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "i"(0));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 0x12345) sink[0] = acc;
}
template <int INFLIGHT, int STRIDE>
void run(const char* name) {
  unsigned long long* o; unsigned* s;
  hipMalloc(&o, 256 * 4 * 8); hipMalloc(&s, 4);
  hipLaunchKernelGGL((k<INFLIGHT, STRIDE>), dim3(256), dim3(256), 32768, 0, o, s);
  hipLaunchKernelGGL((k<INFLIGHT, STRIDE>), dim3(256), dim3(256), 32768, 0, o, s);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), o, 1024 * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : h) m += v; m /= 1024;
  printf("%s inflight<=%d stride %d: %.1f cycles per ds_read_b128 per wave (memtime ticks)\n", name, INFLIGHT, STRIDE, m / 2048.0);
}
int main() {
  run<0, 16>("b128"); run<1, 16>("b128"); run<3, 16>("b128"); run<6, 16>("b128"); run<7, 16>("b128"); run<7, 80>("b128 conflicts?");
  return 0;
}
