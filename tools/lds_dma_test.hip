// Checks the LDS layout of global_load_lds_dwordx4 on gfx950: lane l's 16 bytes land at
// M0 + 16 l (a 1-KB contiguous segment per wave instruction).
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_dma_test.hip -o tools/lds_dma_test.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(const double *g, double *out) {
  extern __shared__ double T[];
  for (int e = threadIdx.x; e < 1024; e += blockDim.x) T[e] = -1.0;
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)T;
  const uint32_t voff = (threadIdx.x & 63) * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *base = (const char *)(g + wave * 128);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(base), "s"(lds0 + 2048 + wave * 1024 * 2) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += blockDim.x) out[e] = T[e];
}
int main() {
  double h[1024], o[1024], *g, *out;
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMalloc(&g, sizeof(h)); hipMalloc(&out, sizeof(o));
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 128, 8192>>>(g, out);
  hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int i = 0; i < 128; ++i) if (o[256 + w * 256 + i] != w * 128 + i) ++bad;
  for (int i = 0; i < 256; ++i) if (o[i] != -1.0) ++bad;
  for (int i = 384; i < 512; ++i) if (o[i] != -1.0) ++bad;
  printf("lds dma layout: %s (bad=%d)  o[256..259]=%g %g %g %g o[512]=%g\n", bad ? "UNEXPECTED" : "contiguous", bad, o[256], o[257], o[258], o[259], o[512]);
  return bad != 0;
}
