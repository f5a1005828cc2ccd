// Small n-vector kernels of libobhip: the synthetic benchmark generator, the
// standardisation of y (R/fitting.R:55-57) and the residual pieces of
// loglik_gauss::update / hessmult (src/lpdfs/loglik_gauss.cpp:110-145).
// All are plain HBM-streaming kernels; reductions are two-stage with a fixed
// summation order so results are run-to-run reproducible.
#include "obhip_internal.h"

namespace obhip {

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// u(i, j) = (splitmix64(splitmix64(seed) ^ (i d + j)) >> 11) 2^-53: the seed is hashed into a
// mask of the counter, so streams of different seeds do not overlap (with seed + counter the
// prediction rows of seed 43 were the training rows of seed 42 shifted by one coordinate)
__device__ __forceinline__ double synth_u(uint64_t seed, uint64_t i, uint64_t d, uint64_t j) {
  return (double)(splitmix64(splitmix64(seed) ^ (i * d + j)) >> 11) * 0x1.0p-53;
}

// x = 0.02 + 0.96 u (x 6.283185 on mat25ang dims);
// y = borehole8d(first 8 coordinates) + sum_{j>=8} 20/(j+1) sin(2 pi u_j)
// (Borehole-8d closed form: R/testfuncs.R:32-46).
__global__ void k_synth(uint64_t seed, uint64_t row0, uint64_t n, uint64_t d,
                        const int *__restrict__ kinds, double *__restrict__ x,
                        double *__restrict__ y) {
#pragma clang fp contract(off)  // x must be bit-identical on every rank and on the host
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t i = row0 + r;
  double b[8];
  double extra = 0.0;
  for (uint64_t j = 0; j < d; ++j) {
    const double u = synth_u(seed, i, d, j);
    const double xv = 0.02 + 0.96 * u;
    x[j * n + r] = kinds[j] == OBHIP_COV_MAT25ANG ? xv * 6.283185 : xv;
    if (j < 8)
      b[j] = xv;
    else
      extra += (20.0 / (double)(j + 1)) * sin(2.0 * 3.141592653589793 * u);
  }
  for (uint64_t j = d; j < 8; ++j) b[j] = 0.5;
  const double rw = b[0] * (0.15 - 0.05) + 0.05;
  const double rr = b[1] * (50000.0 - 100.0) + 100.0;
  const double Tu = b[2] * (115600.0 - 63070.0) + 63070.0;
  const double Hu = b[3] * (1110.0 - 990.0) + 990.0;
  const double Tl = b[4] * (116.0 - 63.1) + 63.1;
  const double Hl = b[5] * (820.0 - 700.0) + 700.0;
  const double L = b[6] * (1680.0 - 1120.0) + 1120.0;
  const double Kw = b[7] * (12045.0 - 9855.0) + 9855.0;
  const double m1 = 2.0 * 3.141592653589793 * Tu * (Hu - Hl);
  const double m2 = log(rr / rw);
  const double m3 = 1.0 + 2.0 * L * Tu / (m2 * rw * rw * Kw) + Tu / Tl;
  y[r] = (m1 / m2 / m3 - 77.0) + extra;
}

constexpr int kRedBlocks = 1024;

// stage 1: per-block partial (sum, sum of squares); stage 2: one block.
__global__ void __launch_bounds__(256)
k_sum2_stage1(const double *__restrict__ v, uint64_t n, double *__restrict__ part) {
  __shared__ double s1[256], s2[256];
  double a = 0.0, b = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const double t = v[i];
    a += t;
    b += t * t;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s1[threadIdx.x] += s1[threadIdx.x + off];
      s2[threadIdx.x] += s2[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s1[0];
    part[2 * blockIdx.x + 1] = s2[0];
  }
}

__global__ void __launch_bounds__(256)
k_sum2_stage2(const double *__restrict__ part, int nblk, double *__restrict__ out2) {
  __shared__ double s1[256], s2[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    a += part[2 * i];
    b += part[2 * i + 1];
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s1[threadIdx.x] += s1[threadIdx.x + off];
      s2[threadIdx.x] += s2[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = s1[0];
    out2[1] = s2[0];
  }
}

__global__ void k_affine(double *__restrict__ v, uint64_t n, double cent, double sca) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (v[i] - cent) / sca;
}

// r = -e2 * (yhat - y); diff (in place of yhat) kept for the sum of squares
__global__ void k_resid(const double *__restrict__ yhat, const double *__restrict__ y, uint64_t n,
                        double e2, double *__restrict__ r, double *__restrict__ diff) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double dlt = yhat[i] - y[i];
    diff[i] = dlt;
    r[i] = -e2 * dlt;
  }
}

__global__ void k_scale(double *__restrict__ v, uint64_t n, double c) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= c;
}

__global__ void k_fill(double *__restrict__ v, uint64_t n, double c) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = c;
}

}  // namespace

int launch_synth(uint64_t seed, uint64_t row0, uint64_t n, uint64_t d, const int *d_kinds,
                 double *d_x, double *d_y) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_synth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cur_stream(), seed,
                     row0, n, d, d_kinds, d_x, d_y);
  OB_HIP(hipGetLastError());
  return 0;
}

// out[i] = add + sum_{k < p} Z[k + i * ld]^2, one wave per column
__global__ void __launch_bounds__(256)
k_colnorm2(const double *__restrict__ Z, uint64_t ld, uint64_t p, uint64_t n, double add,
           double *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const double *z = Z + i * ld;
  double s = 0.0;
  for (uint64_t k = lane; k < p; k += 64) s = fma(z[k], z[k], s);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) out[i] = s + add;
}

int launch_colnorm2(const double *d_Z, uint64_t ld, uint64_t p, uint64_t n, double add,
                    double *d_out) {
  hipLaunchKernelGGL(k_colnorm2, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, cur_stream(), d_Z, ld,
                     p, n, add, d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

// out = sum_{i < n, k < p} A[k + i ld] B[k + i ld] (two stages, fixed order)
__global__ void __launch_bounds__(256)
k_dot_cols1(const double *__restrict__ A, const double *__restrict__ B, uint64_t ld, uint64_t p,
            uint64_t n, double *__restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const double *a = A + i * ld, *b = B + i * ld;
    for (uint64_t k = threadIdx.x; k < p; k += 256) s = fma(a[k], b[k], s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void k_dot_cols2(const double *__restrict__ part, int nblk, double *__restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[i];
    *out = s;
  }
}
__global__ void k_set_identity(double *__restrict__ A, uint64_t p) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < p * p) A[e] = (e / p == e % p) ? 1.0 : 0.0;
}

int launch_dot_cols(const double *d_A, const double *d_B, uint64_t ld, uint64_t p, uint64_t n,
                    double *d_out, double *d_part /* 4096 doubles */) {
  const int nblk = (int)std::min<uint64_t>(4096, std::max<uint64_t>(1, n));
  hipLaunchKernelGGL(k_dot_cols1, dim3(nblk), dim3(256), 0, cur_stream(), d_A, d_B, ld, p, n, d_part);
  hipLaunchKernelGGL(k_dot_cols2, dim3(1), dim3(64), 0, cur_stream(), d_part, nblk, d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_set_identity(double *d_A, uint64_t p) {
  hipLaunchKernelGGL(k_set_identity, dim3((unsigned)((p * p + 255) / 256)), dim3(256), 0, cur_stream(),
                     d_A, p);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_sum_sumsq(const double *d_v, uint64_t n, double *d_out2, double *d_part /* 2*kRedBlocks */) {
  const int nblk = (int)std::min<uint64_t>(kRedBlocks, std::max<uint64_t>(1, (n + 255) / 256));
  hipLaunchKernelGGL(k_sum2_stage1, dim3(nblk), dim3(256), 0, cur_stream(), d_v, n, d_part);
  hipLaunchKernelGGL(k_sum2_stage2, dim3(1), dim3(256), 0, cur_stream(), d_part, nblk, d_out2);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_affine(double *d_v, uint64_t n, double cent, double sca) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_affine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cur_stream(), d_v, n,
                     cent, sca);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_resid(const double *d_yhat, const double *d_y, uint64_t n, double e2, double *d_r,
                 double *d_diff) {
  hipLaunchKernelGGL(k_resid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cur_stream(), d_yhat,
                     d_y, n, e2, d_r, d_diff);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_scale(double *d_v, uint64_t n, double c) {
  hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cur_stream(), d_v, n, c);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_fill(double *d_v, uint64_t n, double c) {
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cur_stream(), d_v, n, c);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip

// ---- exact order statistics of row-sharded columns (sharded obfit's quantile knots) -------
// Keys: doubles mapped to uint64 so that unsigned order = numeric order.  One bisection step
// of the selection counts, for every (column, target) pair, the local elements whose key is
// <= the target's midpoint: a wave takes 64 rows of one column, a ballot per target gives
// the count of the wave, one lane accumulates it in LDS.  Integer counts: order-independent.
namespace obhip {

namespace {

__device__ __forceinline__ uint64_t order_key(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

constexpr int kQMaxT = 512;  // targets per column

__global__ void __launch_bounds__(256)
k_count_le(const double *__restrict__ x, uint64_t n, const uint64_t *__restrict__ mids, int T,
           unsigned long long *__restrict__ counts) {
  __shared__ uint64_t sm[kQMaxT];
  __shared__ unsigned int cnt[4][kQMaxT];
  const int l = blockIdx.y, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int t = threadIdx.x; t < T; t += 256) {
    sm[t] = mids[(size_t)l * T + t];
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][t] = 0;
  }
  __syncthreads();
  const double *col = x + (uint64_t)l * n;
  for (uint64_t r0 = ((uint64_t)blockIdx.x * 4 + wave) * 64; r0 < n; r0 += (uint64_t)gridDim.x * 256) {
    const uint64_t r = r0 + lane;
    const bool live = r < n;
    const uint64_t key = live ? order_key(col[r]) : ~0ull;
    for (int t = 0; t < T; ++t) {
      const unsigned long long m = __ballot(live && key <= sm[t]);
      if (lane == 0) cnt[wave][t] += (unsigned int)__popcll(m);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) {
    const unsigned long long c = (unsigned long long)cnt[0][t] + cnt[1][t] + cnt[2][t] + cnt[3][t];
    if (c) atomicAdd(&counts[(size_t)l * T + t], c);
  }
}

}  // namespace

__global__ void k_u64_to_f64(const unsigned long long *__restrict__ in, uint64_t n, double *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}

int launch_u64_to_f64(const unsigned long long *d_in, uint64_t n, double *d_out) {
  hipLaunchKernelGGL(k_u64_to_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cur_stream(), d_in, n,
                     d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

int quantile_max_targets() { return kQMaxT; }

// counts[l * T + t] += #{ i : key(x[i, l]) <= mids[l * T + t] } (counts zeroed by the caller)
int launch_count_le(const double *d_x, uint64_t n, uint64_t d, const uint64_t *d_mids, int T,
                    unsigned long long *d_counts) {
  if (T > kQMaxT) return fail(OBHIP_ERR_INVALID, "too many quantiles per column");
  if (n == 0) return 0;
  // each wave's counter is 32 bits: at most 2^32 rows per block
  const unsigned gx = (unsigned)std::min<uint64_t>(1024, (n + 255) / 256);
  hipLaunchKernelGGL(k_count_le, dim3(gx, (unsigned)d), dim3(256), 0, cur_stream(), d_x, n, d_mids, T,
                     d_counts);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
