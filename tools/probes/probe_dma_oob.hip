#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// probe: buffer_load ... lds with out-of-range lanes: are zeros written to LDS?
__global__ void k(const unsigned* g, unsigned nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef __attribute__((address_space(3))) void lv;
  for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)lds)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)nbytes, 0x00020000);
  // lanes 0..31 in range, lanes 32..63 beyond num_records
  unsigned voff = threadIdx.x < 32 ? threadIdx.x * 16u : 0xfffffff0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lv*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)lds)[i];
}
int main() {
  unsigned *g, *o;
  hipMalloc(&g, 4096); hipMalloc(&o, 4096);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, 512u, o);
  hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
  printf("in-range lane 0: %x %x ; lane 31: %x ; OOB lane 32: %x %x ; lane 63: %x\n", h[0], h[1], h[31*4], h[32*4], h[32*4+1], h[63*4]);
  return 0;
}
