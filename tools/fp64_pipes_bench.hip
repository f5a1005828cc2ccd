// Microbenchmark: FP64 throughput of the gfx950 pipes -- v_mfma_f64_16x16x4_f64,
// v_mfma_f64_4x4x4_4b_f64, v_fma_f64 (VALU) and MFMA+VALU side by side.
// The guides carry no FP64 numbers; DESIGN.md quotes these next to the 78.6 TF
// datasheet value.
// build: hipcc --offload-arch=gfx950 -O3 tools/fp64_pipes_bench.hip -o tools/fp64_pipes_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: all waves MFMA 16x16x4; 1: all waves VALU fma; 2: even waves MFMA, odd waves VALU;
// 3: MFMA 4x4x4; 4: each wave interleaves 1 MFMA with NV VALU fmas
template <int NACC, int NV>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0, int mode,
                                         unsigned long long *clk) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  d4 acc[NACC];
  double v[16];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0);
  const bool do_valu = mode == 1 || (mode == 2 && (wave & 1) == 1);
  if (mode == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(i * NV + j) & 15] = fma(a, b, v[(i * NV + j) & 15]);
      }
    }
  } else if (mode == 3) {
    double s4[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) s4[i] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) s4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) v[0] += s4[i];
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(a, b, v[i]);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 256) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int NACC, int NV>
void run(const char *label, int mode, int blocks, int iters) {
  double *out;
  unsigned long long *clk, hclk[512];
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&clk, sizeof(hclk));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC, NV><<<blocks, 256>>>(out, iters, 1.0, 0.5, mode, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, NV><<<blocks, 256>>>(out, iters, 1.0, 0.5, mode, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hclk, clk, sizeof(hclk), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { cyc += hclk[2 * i]; rt += hclk[2 * i + 1]; }
  const double waves = 4.0 * blocks;
  const double mf = 2.0 * 16 * 16 * 4 * (double)NACC * iters;  // per MFMA wave
  const double vf = 2.0 * 64 * 64 * (double)iters;             // per VALU wave
  double mfl = 0, vfl = 0;
  if (mode == 0) mfl = mf * waves;
  if (mode == 1) vfl = vf * waves;
  if (mode == 2) { mfl = mf * waves / 2; vfl = vf * waves / 2; }
  if (mode == 3) mfl = 2.0 * 4 * 4 * 4 * 4 * (double)NACC * iters * waves;
  if (mode == 4) { mfl = mf * waves; vfl = 2.0 * 64 * NV * (double)NACC * iters * waves; }
  printf("%-34s waves/SIMD=%d  %.2f ms  MFMA %.2f TF  VALU %.2f TF  total %.2f TF  clock %.3f GHz\n", label,
         (blocks + 255) / 256, ms, mfl / ms / 1e9, vfl / ms / 1e9, (mfl + vfl) / ms / 1e9,
         cyc / rt * 0.1);
  hipFree(out);
  hipFree(clk);
}

int main() {
  run<4, 0>("mfma16x16x4 nacc=4", 0, 1024, 50000);
  run<8, 0>("mfma16x16x4 nacc=8", 0, 2048, 25000);
  run<16, 0>("mfma16x16x4 nacc=16", 0, 512, 50000);
  run<4, 0>("valu fma_f64", 1, 256, 100000);
  run<4, 0>("valu fma_f64", 1, 512, 100000);
  run<4, 0>("valu fma_f64", 1, 1024, 50000);
  run<4, 0>("valu fma_f64", 1, 2048, 25000);
  run<4, 0>("mfma(even waves)+valu(odd waves)", 2, 512, 50000);
  run<4, 0>("mfma(even waves)+valu(odd waves)", 2, 1024, 50000);
  run<4, 0>("mfma(even waves)+valu(odd waves)", 2, 2048, 25000);
  run<8, 0>("mfma4x4x4 nacc=8", 3, 1024, 50000);
  run<8, 0>("mfma4x4x4 nacc=8", 3, 2048, 50000);
  run<8, 4>("interleave 1 mfma : 4 fma", 4, 512, 20000);
  run<8, 8>("interleave 1 mfma : 8 fma", 4, 512, 20000);
  run<8, 16>("interleave 1 mfma : 16 fma", 4, 512, 10000);
  run<8, 16>("interleave 1 mfma : 16 fma", 4, 1024, 10000);
  run<8, 32>("interleave 1 mfma : 32 fma", 4, 1024, 10000);
  return 0;
}
