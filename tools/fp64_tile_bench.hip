// Microbenchmark: what a register-tiled v_fma_f64 outer-product loop sustains on
// gfx950 with operands (1) kept in registers, (2) re-read from LDS with
// ds_read_b128 every step (the access pattern of k_gram_valu), for an 8x8 and a
// 16x8 per-lane tile.  Sets the ceiling for the vector-pipe Gram kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/fp64_tile_bench.hip -o tools/fp64_tile_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int TM, int TN, bool LDS, int MINW>
__global__ void __launch_bounds__(256, MINW) k(double *out, int iters, const double *in) {
  __shared__ double T[256 * 10 * 2];
  for (int e = threadIdx.x; e < 256 * 10 * 2; e += 256) T[e] = in[e & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ly = lane >> 3, lx = lane & 7;
  const double *ta = T + ((wave >> 1) * 64 + 2 * ly) * 10;
  const double *tb = T + (128 + (wave & 1) * 64 + 2 * lx) * 10;
  double acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;
  d2 a[TM], b[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a[i] = *(const d2 *)(ta + ((i >> 1) * 16 + (i & 1)) % 120 * 10);
#pragma unroll
  for (int j = 0; j < TN; ++j) b[j] = *(const d2 *)(tb + ((j >> 1) * 16 + (j & 1)) * 10);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      if (LDS) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *(const d2 *)(ta + (((i >> 1) * 16 + (i & 1)) % 120) * 10 + 2 * kp);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const d2 *)(tb + ((j >> 1) * 16 + (j & 1)) * 10 + 2 * kp);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = fma(a[i].x, b[j].x, acc[i][j]);
          acc[i][j] = fma(a[i].y, b[j].y, acc[i][j]);
        }
    }
    if (LDS) asm volatile("" ::: "memory");
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TM, int TN, bool LDS, int MINW>
void run(const char *label, int blocks, int iters) {
  double *out, *in;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&in, sizeof(double) * 1024);
  hipMemset(in, 0, sizeof(double) * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<TM, TN, LDS, MINW><<<blocks, 256>>>(out, iters, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<TM, TN, LDS, MINW><<<blocks, 256>>>(out, iters, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 2 * TM * TN * 4.0 * iters * 256.0 * blocks;
  printf("%-44s blocks=%4d  %.2f ms  %.2f TFLOP/s\n", label, blocks, ms, flops / ms / 1e9);
  hipFree(out);
  hipFree(in);
}

int main() {
  run<8, 8, false, 2>("8x8 tile, operands in registers, 2 waves/SIMD", 512, 20000);
  run<8, 8, false, 2>("8x8 tile, operands in registers, 1 wave/SIMD", 256, 20000);
  run<8, 8, true, 2>("8x8 tile, ds_read_b128 operands, 2 waves/SIMD", 512, 20000);
  run<8, 8, true, 2>("8x8 tile, ds_read_b128 operands, 1 wave/SIMD", 256, 20000);
  run<16, 8, false, 1>("16x8 tile, operands in registers, 1 wave/SIMD", 256, 10000);
  run<16, 8, true, 1>("16x8 tile, ds_read_b128 operands, 1 wave/SIMD", 256, 10000);
  run<8, 4, true, 4>("8x4 tile, ds_read_b128 operands, 4 waves/SIMD", 1024, 20000);
  return 0;
}
