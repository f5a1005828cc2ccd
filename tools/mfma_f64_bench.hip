// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (the guides give
// no FP64 MFMA number; DESIGN.md quotes the result next to the 78.6 TF spec).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o /tmp/mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0,
                                          unsigned long long *clk) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 256) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;       // shader cycles
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;  // 100 MHz ticks
  }
}

template <int NACC>
void run(int blocks, int iters) {
  double *out;
  unsigned long long *clk, hclk[512];
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&clk, sizeof(hclk));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(out, iters, 1.0, 0.5, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, iters, 1.0, 0.5, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hclk, clk, sizeof(hclk), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { cyc += hclk[2 * i]; rt += hclk[2 * i + 1]; }
  const double ghz = cyc / rt * 0.1;  // cycles per 10 ns tick
  const int wps = (blocks + 255) / 256;
  double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks;
  printf("nacc=%d waves/SIMD=%d iters=%d  %.2f ms  %.2f TFLOP/s  in-kernel clock %.3f GHz  %.1f shader cycles per MFMA per SIMD\n",
         NACC, wps, iters, ms, flops / ms / 1e9, ghz, (cyc / 256) / ((double)NACC * iters * wps));
  hipFree(out);
}

int main() {
  run<1>(256, 200000);
  run<4>(256, 200000);
  run<16>(256, 100000);
  run<16>(512, 50000);
  run<4>(1024, 100000);
  run<8>(2048, 25000);
  return 0;
}
