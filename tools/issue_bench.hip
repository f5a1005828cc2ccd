// Microbenchmark: how fast can a THIRD wave on a SIMD issue instructions while two
// other waves keep the FP64 matrix pipe saturated with v_mfma_f64_4x4x4_4b_f64?
// (the situation of the producer waves of k_gram_mfma4).  Prints the third wave's
// cycles per instruction for several instruction kinds, with and without s_setprio.
// build: hipcc --offload-arch=gfx950 -O3 tools/issue_bench.hip -o tools/issue_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int KIND, int PRIO>
__global__ void __launch_bounds__(768) k(double *out, int iters, unsigned long long *cyc) {
  __shared__ double L[8192];
  for (int e = threadIdx.x; e < 8192; e += 768) L[e] = 1.0 + 1e-9 * e;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 8) {  // MFMA waves: 32 accumulators, register operands
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    const double a = 1.0 + lane * 1e-9, b = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 768 + threadIdx.x] = s;
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  double v[8];
  d2 w[8];
  int x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = 1.0 + i; x[i] = lane + i; w[i] = d2{0, 0}; }
  const double m = 1.0 + 1e-12 * lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int n = iters / 4;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) v[i] *= m;                                             // v_mul_f64
      if (KIND == 1) x[i] = (x[i] * 3 + lane) ^ it;                         // integer VALU
      if (KIND == 2) w[i] = *(const d2 *)(L + ((lane * 2 + i * 128 + it * 2) & 8190));  // ds_read_b128
      if (KIND == 3) L[(lane + i * 64 + wave * 512) & 8191] = v[i];          // ds_write_b64
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + x[i] + w[i].x + w[i].y;
  out[blockIdx.x * 768 + threadIdx.x] = s;
  if (threadIdx.x == 512) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int PRIO>
void run(const char *label, int iters) {
  double *out;
  unsigned long long *cyc, h[256];
  hipMalloc(&out, sizeof(double) * 256 * 768);
  hipMalloc(&cyc, sizeof(h));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND, PRIO><<<256, 768>>>(out, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND, PRIO><<<256, 768>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0;
  for (int i = 0; i < 256; ++i) c += h[i];
  const double ninstr = (iters / 4) * 8.0 * (KIND == 1 ? 3 : 1);
  printf("%-28s prio=%d  third wave: %.1f cycles/instr   MFMA waves: %.1f TFLOP/s\n", label, PRIO,
         c / 256 / ninstr, 2.0 * 256 * 32.0 * iters * 8 * 256 / ms / 1e9);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, 0>("v_mul_f64", 40000);
  run<0, 1>("v_mul_f64", 40000);
  run<1, 0>("int VALU (mul/add/xor)", 40000);
  run<1, 1>("int VALU (mul/add/xor)", 40000);
  run<2, 0>("ds_read_b128", 40000);
  run<2, 1>("ds_read_b128", 40000);
  run<3, 0>("ds_write_b64", 40000);
  run<3, 1>("ds_write_b64", 40000);
  return 0;
}
