// Probes global_load_lds_dword on gfx950 (the tile prefetch of k_hm2, csrc/kernels_hm.hip):
//   (a) lane l's 4 bytes land at M0 + inst_offset + 4 l (256 contiguous bytes per wave instruction),
//   (b) the instruction's immediate offset advances BOTH the global and the LDS address,
//   (c) M0 offsets beyond 64 KB address the upper part of a 160-KB allocation.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_dma_b32_test.hip -o tools/lds_dma_b32_test.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
constexpr int kN = 18000;  // doubles of LDS = 144 000 bytes
__global__ void k(const double *g, double *out) {
  extern __shared__ double T[];
  for (int e = threadIdx.x; e < kN; e += blockDim.x) T[e] = -1.0;
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)T;
  const uint32_t voff = (threadIdx.x & 63) * 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // wave w copies 64 doubles g[64 w ..] to T[c0 + 65 w ..] (pitch 65), c0 = 0 and 16000 (> 64 KB)
  for (int rep = 0; rep < 2; ++rep) {
    const char *base = (const char *)(g + wave * 64);
    const uint32_t l = lds0 + (rep ? 16000 * 8 : 0) + wave * 65 * 8;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\tglobal_load_lds_dword %0, %1 offset:256"
                 :: "v"(voff), "s"(base), "s"(l) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < kN; e += blockDim.x) out[e] = T[e];
}
int main() {
  static double h[256], o[kN];
  double *g, *out;
  for (int i = 0; i < 256; ++i) h[i] = 1000.0 + i;
  hipMalloc(&g, sizeof(h)); hipMalloc(&out, sizeof(o));
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kN * 8);
  k<<<1, 256, kN * 8>>>(g, out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
  hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int rep = 0; rep < 2; ++rep)
    for (int w = 0; w < 4; ++w)
      for (int i = 0; i < 65; ++i) {
        const double want = i < 64 ? 1000.0 + w * 64 + i : -1.0;
        // (the pad slot of wave w is overwritten by nobody: next wave starts at 65 (w + 1))
        if (o[(rep ? 16000 : 0) + w * 65 + i] != want) ++bad;
      }
  printf("lds dma b32: %s (bad=%d)  low: %g %g .. %g | %g   high: %g %g .. %g\n", bad ? "UNEXPECTED" : "as assumed", bad,
         o[0], o[1], o[63], o[64], o[16000], o[16001], o[16063]);
  return bad != 0;
}
