// Microbenchmark: v_mfma_f64_4x4x4_4b_f64 consumer loop of k_gram_mfma4 in isolation --
// 32 accumulators per wave, 12 operands per K=4 step re-read from LDS, optional
// workgroup barrier every 4 steps, 1..3 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_lds_bench.hip -o tools/mfma4x4_lds_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kTP = 272;
static double g_fill = 0.0;

template <int THREADS, bool LDS, bool BARRIER, int NMUL = 0, int NVALU = 0>
__global__ void __launch_bounds__(THREADS) k(double *out, int chunks, const double *in) {
  __shared__ double T[16 * kTP];
  for (int e = threadIdx.x; e < 16 * kTP; e += THREADS) T[e] = in[e & 1023] + 1e-9 * e;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (NMUL > 0 && wave >= 8) {
    // producer stand-in: NMUL FP64 multiplies per chunk in 8 independent chains
    __builtin_amdgcn_s_setprio(3);
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1.0 + 1e-9 * (lane + i);
    const double m = 1.0 + 1e-12 * lane;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
      for (int q = 0; q < NMUL / 8; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= m;
      __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    return;
  }
  const int wm = (wave >> 2) & 1, wn = wave & 3;
  const int mk = lane >> 4, mblk = (lane >> 2) & 3, me = lane & 3;
  const int abase = mk * kTP + wm * 64 + mblk * 4 + me;
  int bbase[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bbase[r] = mk * kTP + 128 + wn * 32 + ((mblk + r) & 3) * 4 + me;
  double acc[4][2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  auto load_ops = [&](int step, double (&a)[4], double (&b)[2][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = T[abase + i * 16 + 4 * step * kTP];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) b[j][r] = T[bbase[r] + j * 16 + 4 * step * kTP];
  };
  auto mfma_step = [&](const double (&a)[4], const double (&b)[2][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j][r], acc[i][j][r], 0, 0, 0);
  };
  double a0[4], b0[2][4], a1[4], b1[2][4];
  int dummy = lane;
  load_ops(0, a0, b0);
  load_ops(1, a1, b1);
  for (int c = 0; c < chunks; ++c) {
#pragma unroll
    for (int s = 0; s < 4; s += 2) {
      if (LDS) load_ops(s + 1, a1, b1);
      // NVALU integer VALU instructions per chunk inside the consumer's own stream (the
      // LDS address arithmetic a dynamic buffer index costs)
#pragma unroll
      for (int q = 0; q < NVALU / 2; ++q) asm volatile("v_add_u32 %0, 1, %0" : "+v"(dummy));
      mfma_step(a0, b0);
      if (LDS) load_ops((s + 2) & 3, a0, b0);
      mfma_step(a1, b1);
      if (LDS) asm volatile("" ::: "memory");
    }
    if (BARRIER) __syncthreads();
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s + dummy;
}

template <int THREADS, bool LDS, bool BARRIER, int NMUL = 0, int NVALU = 0>
void run(const char *label, int chunks) {
  const int blocks = 256;
  double *out, *in;
  hipMalloc(&out, sizeof(double) * blocks * THREADS);
  hipMalloc(&in, sizeof(double) * 1024);
  { double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = g_fill != 0.0 ? g_fill * (1 + i % 7) : 0.0; if (g_fill < 0) for (int i = 0; i < 1024; ++i) h[i] = __builtin_nan(""); hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice); }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<THREADS, LDS, BARRIER, NMUL, NVALU><<<blocks, THREADS>>>(out, chunks, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<THREADS, LDS, BARRIER, NMUL, NVALU><<<blocks, THREADS>>>(out, chunks, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 256 * 128.0 * chunks * (NMUL > 0 ? 8 : THREADS / 64) * blocks;
  printf("%-52s waves/SIMD=%d  %.2f ms  %.2f TFLOP/s\n", label, THREADS / 256, ms, flops / ms / 1e9);
  hipFree(out);
  hipFree(in);
}

int main(int argc, char **argv) {
  g_fill = argc > 1 ? atof(argv[1]) : 0.0;
  run<256, false, false>("32 acc, register operands", 40000);
  run<512, false, false>("32 acc, register operands", 40000);
  run<768, false, false>("32 acc, register operands", 40000);
  run<256, true, false>("32 acc, 12 LDS operands/step", 40000);
  run<512, true, false>("32 acc, 12 LDS operands/step", 40000);
  run<768, true, false>("32 acc, 12 LDS operands/step", 40000);
  run<512, true, true>("32 acc, LDS operands, barrier per 4 steps", 40000);
  run<768, true, true>("32 acc, LDS operands, barrier per 4 steps", 40000);
  run<512, true, true, 0, 8>("8 consumers, barrier, + 8 int VALU / chunk", 40000);
  run<512, true, true, 0, 24>("8 consumers, barrier, + 24 int VALU / chunk", 40000);
  run<512, true, true, 0, 64>("8 consumers, barrier, + 64 int VALU / chunk", 40000);
  run<768, true, true, 8>("8 consumers + 4 waves x 8 v_mul_f64 / chunk", 40000);
  run<768, true, true, 64>("8 consumers + 4 waves x 64 v_mul_f64 / chunk", 40000);
  run<768, true, true, 256>("8 consumers + 4 waves x 256 v_mul_f64 / chunk", 40000);
  return 0;
}
