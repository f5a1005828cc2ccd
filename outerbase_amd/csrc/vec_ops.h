// Element-wise maps and fixed-order sums over n-vectors in HBM, written as device lambdas at
// the call site (the likelihood classes of lpdf.cpp: residuals, observation standard
// deviations, weighted sums -- src/lpdfs/*.cpp are Armadillo expression templates over the
// same vectors).  Plain HBM-streaming kernels; reductions are two-stage with a fixed
// summation order, so results are run-to-run reproducible.
#pragma once
#include "obhip_internal.h"

namespace obhip {

template <class F>
__global__ void __launch_bounds__(256) k_vmap(uint64_t n, F f) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) f(i);
}

// f(i) for i in [0, n)
template <class F>
int vmap(uint64_t n, F f) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_vmap<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cur_stream(), n, f);
  OB_HIP(hipGetLastError());
  return 0;
}

constexpr int kSumBlocks = 512;

template <int K, class F>
__global__ void __launch_bounds__(256) k_vsum1(uint64_t n, F f, double *__restrict__ part) {
  __shared__ double red[K][256];
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    f(i, acc);
#pragma unroll
  for (int k = 0; k < K; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
#pragma unroll
      for (int k = 0; k < K; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x < K) part[(uint64_t)blockIdx.x * K + threadIdx.x] = red[threadIdx.x][0];
}

// One wave per sum (block k of the grid): lane l adds the partials of blocks l, l + 64, ... in
// ascending order, the 64 lane sums meet in a butterfly -- a fixed order as before, but 8
// independent loads per lane where one thread walked 512 dependent ones (17-25 us per sum: a fifth
// of a millisecond per obfit evaluation went there).
template <int K>
__global__ void __launch_bounds__(64) k_vsum2(const double *__restrict__ part, int nblk,
                                              double *__restrict__ out) {
  const int k = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += part[(uint64_t)b * K + k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (threadIdx.x == 0) out[k] = s;
}

// d_out[k] = sum_i (what f(i, acc) adds to acc[k]), k < K <= 64; d_part: kSumBlocks * K doubles
template <int K, class F>
int vsum(uint64_t n, F f, double *d_out, double *d_part) {
  static_assert(K >= 1 && K <= 64, "at most 64 simultaneous sums");
  const int nblk = (int)std::min<uint64_t>(kSumBlocks, std::max<uint64_t>(1, (n + 255) / 256));
  hipLaunchKernelGGL((k_vsum1<K, F>), dim3(nblk), dim3(256), 0, cur_stream(), n, f, d_part);
  hipLaunchKernelGGL(k_vsum2<K>, dim3(K), dim3(64), 0, cur_stream(), d_part, nblk, d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
