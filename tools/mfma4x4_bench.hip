// Microbenchmark + layout probe for v_mfma_f64_4x4x4_4b_f64 on gfx950:
// (1) rate with 64 independent accumulators per wave (the register budget of a
//     64 x 64 wave tile) at 1..3 waves per SIMD, with and without the CBSZ/ABID
//     A-block broadcast;  (2) operand / result lane layout, probed with one-hot inputs.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_bench.hip -o tools/mfma4x4_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int BCAST>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0) {
  double acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.0;
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = a0 + threadIdx.x * 1e-9 + i;
    b[i] = b0 - threadIdx.x * 1e-9 - i;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (BCAST) {
          acc[(i * 4 + j) * 4 + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[(i * 4 + j) * 4 + 0], 2, 0, 0);
          acc[(i * 4 + j) * 4 + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[(i * 4 + j) * 4 + 1], 2, 1, 0);
          acc[(i * 4 + j) * 4 + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[(i * 4 + j) * 4 + 2], 2, 2, 0);
          acc[(i * 4 + j) * 4 + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[(i * 4 + j) * 4 + 3], 2, 3, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[(i * 4 + j) * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[(i * 4 + j) * 4 + r], 0, 0, 0);
        }
      }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int BCAST>
void run(int blocks, int iters) {
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<BCAST><<<blocks, 256>>>(out, iters, 1.0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<BCAST><<<blocks, 256>>>(out, iters, 1.0, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 4 * 4 * 4 * 4 * 64.0 * iters * 4.0 * blocks;
  printf("mfma_f64_4x4x4 nacc=64 bcast=%d waves/SIMD=%d  %.2f ms  %.2f TFLOP/s\n", BCAST, blocks / 256, ms,
         flops / ms / 1e9);
  hipFree(out);
}

// layout probe: A = one-hot at lane la, B = one-hot at lane lb -> which lane of D gets 1?
__global__ void probe(const double *a, const double *b, double *d, int cbsz, int abid) {
  double c = 0.0;
  double r;
  if (cbsz == 0) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  else if (abid == 0) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], c, 2, 0, 0);
  else if (abid == 1) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], c, 2, 1, 0);
  else if (abid == 2) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], c, 2, 2, 0);
  else r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], c, 2, 3, 0);
  d[threadIdx.x] = r;
}

int main() {
  run<0>(256, 4000);
  run<0>(512, 4000);
  run<0>(768, 4000);
  run<1>(256, 4000);
  run<1>(512, 4000);
  run<1>(768, 4000);
  // layout: a[l] = 1 + l (distinct), b one-hot at lane lb: D lane m = sum_k A[i][k] B[k][j]
  double *da, *db, *dd;
  hipMalloc(&da, 64 * 8);
  hipMalloc(&db, 64 * 8);
  hipMalloc(&dd, 64 * 8);
  std::vector<double> ha(64), hb(64), hd(64);
  for (int cb = 0; cb < 2; ++cb)
    for (int abid = 0; abid < (cb ? 4 : 1); ++abid)
      for (int lb : {0, 1, 4, 5, 16, 21, 63}) {
        for (int l = 0; l < 64; ++l) { ha[l] = 100 + l; hb[l] = (l == lb) ? 1.0 : 0.0; }
        hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(da, db, dd, cb ? 2 : 0, abid);
        hipMemcpy(hd.data(), dd, 512, hipMemcpyDeviceToHost);
        printf("cbsz=%d abid=%d B one-hot lane %2d -> D nonzero:", cb ? 2 : 0, abid, lb);
        for (int l = 0; l < 64; ++l)
          if (hd[l] != 0.0) printf(" D[%d]=A[%d]", l, (int)hd[l] - 100);
        printf("\n");
      }
  return 0;
}
