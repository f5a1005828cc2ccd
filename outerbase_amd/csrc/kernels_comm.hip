// Pack / unpack kernels of the one-buffer exchange of back end A:
//   buf = [ upper triangle of G, row-major packed: p (p + 1) / 2 ][ g = B^T y : p ]
//         [ b1 = B^T 1 : p ][ sum y, sum y^2, n ][ zero padding ]
// and the finalisation that turns the globally summed pieces into the right-hand side of
// the standardised problem: with cent = sum y / n and sca = sd(y) (n - 1 denominator,
// R/fitting.R:55-57), B^T ((y - cent) / sca) = (B^T y - cent B^T 1) / sca, so that the
// standardisation of y over ALL ranks needs no exchange of its own.
// HBM-streaming kernels, fixed summation order.
#include "obhip_internal.h"

namespace obhip {

namespace {

__device__ __forceinline__ uint64_t tri_off(uint64_t i, uint64_t p) {
  return i * p - i * (i - 1) / 2;  // start of row i (entries j >= i) in the packed triangle
}

// one block per row i: G[i][i..p) -> buf[tri_off(i) ...]
__global__ void __launch_bounds__(256)
k_pack_tri(const double *__restrict__ G, uint64_t p, double *__restrict__ buf) {
  const uint64_t i = blockIdx.x;
  const double *src = G + i * p;
  double *dst = buf + tri_off(i, p) - i;  // dst[j] for j >= i
  for (uint64_t j = i + threadIdx.x; j < p; j += 256) dst[j] = src[j];
}

__global__ void __launch_bounds__(256)
k_pack_tail(const double *__restrict__ g, const double *__restrict__ b1,
            const double *__restrict__ sum2, double nlocal, uint64_t p, uint64_t pad,
            double *__restrict__ tail) {
  const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < p) {
    tail[e] = g[e];
    tail[p + e] = b1[e];
  } else if (e < p + pad) {
    const uint64_t k = e - p;  // 0, 1, 2: the scalars; beyond: padding
    tail[2 * p + k] = k == 0 ? sum2[0] : (k == 1 ? sum2[1] : (k == 2 ? nlocal : 0.0));
  }
}

// 64 x 64 tile (bi <= bj) of the packed triangle -> G[i][j] and, transposed through LDS,
// G[j][i]; both written in 512-byte row segments
__global__ void __launch_bounds__(256)
k_unpack_tri(const double *__restrict__ buf, uint64_t p, int nb, double *__restrict__ G) {
  __shared__ double S[64 * 65];
  int bi = 0, rem = blockIdx.x;
  while (rem >= nb - bi) {
    rem -= nb - bi;
    ++bi;
  }
  const int bj = bi + rem;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  const uint64_t j = (uint64_t)bj * 64 + c;
  for (int r = r4; r < 64; r += 4) {
    const uint64_t i = (uint64_t)bi * 64 + r;
    double v = 0.0;
    if (i < p && j < p && j >= i) {
      v = buf[tri_off(i, p) + (j - i)];
      G[i * p + j] = v;
    }
    S[r * 65 + c] = v;
  }
  __syncthreads();
  // mirror: row jj = 64 bj + r, column ii = 64 bi + c holds S[c][r]; strictly below the
  // diagonal only
  const uint64_t ii = (uint64_t)bi * 64 + c;
  for (int r = r4; r < 64; r += 4) {
    const uint64_t jj = (uint64_t)bj * 64 + r;
    if (jj < p && ii < p && ii < jj) G[jj * p + ii] = S[c * 65 + r];
  }
}

// The unpack of a row-sharded Newton fit: the summed packed triangle becomes the Hessian
// H = e2 G + diag(prec) in full symmetric storage (lpdfvec::hess_, fit.cpp:503-512) and its
// diagonal, in the one pass that has to touch the p x p matrix anyway.
__global__ void __launch_bounds__(256)
k_unpack_form(const double *__restrict__ buf, uint64_t p, int nb, double *__restrict__ H, double e2,
              const double *__restrict__ prec, double *__restrict__ diagH) {
  __shared__ double S[64 * 65];
  int bi = 0, rem = blockIdx.x;
  while (rem >= nb - bi) {
    rem -= nb - bi;
    ++bi;
  }
  const int bj = bi + rem;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  const uint64_t j = (uint64_t)bj * 64 + c;
  for (int r = r4; r < 64; r += 4) {
    const uint64_t i = (uint64_t)bi * 64 + r;
    double v = 0.0;
    if (i < p && j < p && j >= i) {
      v = e2 * buf[tri_off(i, p) + (j - i)];
      if (i == j) {
        v += prec[i];
        if (diagH) diagH[i] = v;
      }
      H[i * p + j] = v;
    }
    S[r * 65 + c] = v;
  }
  __syncthreads();
  const uint64_t ii = (uint64_t)bi * 64 + c;
  for (int r = r4; r < 64; r += 4) {
    const uint64_t jj = (uint64_t)bj * 64 + r;
    if (jj < p && ii < p && ii < jj) H[jj * p + ii] = S[c * 65 + r];
  }
}

// tail = [g p][b1 p][sum y, sum y^2, n]: g_out = (g - cent b1) / sca, meansd = cent, sca, n
__global__ void __launch_bounds__(256)
k_finalize_rhs(const double *__restrict__ tail, uint64_t p, double *__restrict__ g_out,
               double *__restrict__ meansd) {
  const double s1 = tail[2 * p], s2 = tail[2 * p + 1], n = tail[2 * p + 2];
  const double cent = s1 / n;
  // sum (y - cent)^2 = sum y^2 - n cent^2; relative rounding error eps (1 + cent^2 / var)
  const double var = fmax(s2 - n * cent * cent, 0.0) / (n - 1.0);
  const double sca = sqrt(var);
  const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < p) g_out[e] = (tail[e] - cent * tail[p + e]) / sca;
  if (e == 0) {
    meansd[0] = cent;
    meansd[1] = sca;
    meansd[2] = n;
  }
}

}  // namespace

uint64_t normal_eq_tail(uint64_t p) { return 2 * p + 3; }

int launch_pack_normal_eq(uint64_t p, bool with_tri, const double *d_G, const double *d_g,
                          const double *d_b1, const double *d_sum2, double nlocal, double *d_buf,
                          uint64_t count) {
  const uint64_t tri = p * (p + 1) / 2;
  if (count < tri + normal_eq_tail(p)) return fail(OBHIP_ERR_INVALID, "exchange buffer too small");
  hipStream_t st = cur_stream();
  if (with_tri) hipLaunchKernelGGL(k_pack_tri, dim3((unsigned)p), dim3(256), 0, st, d_G, p, d_buf);
  const uint64_t pad = count - tri - 2 * p;  // scalars + zero padding
  hipLaunchKernelGGL(k_pack_tail, dim3((unsigned)((p + pad + 255) / 256)), dim3(256), 0, st, d_g, d_b1,
                     d_sum2, nlocal, p, pad, d_buf + tri);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_unpack_normal_eq(uint64_t p, bool with_tri, const double *d_buf, double *d_G, double *d_g,
                            double *d_meansd) {
  const uint64_t tri = p * (p + 1) / 2;
  hipStream_t st = cur_stream();
  if (with_tri) {
    const int nb = (int)((p + 63) / 64);
    hipLaunchKernelGGL(k_unpack_tri, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, st, d_buf, p,
                       nb, d_G);
  }
  hipLaunchKernelGGL(k_finalize_rhs, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, st, d_buf + tri,
                     p, d_g, d_meansd);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_unpack_form(uint64_t p, const double *d_tri, double *d_H, double e2, const double *d_prec,
                       double *d_diagH) {
  ProfScope ps("unpack_form");
  const int nb = (int)((p + 63) / 64);
  hipLaunchKernelGGL(k_unpack_form, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, cur_stream(),
                     d_tri, p, nb, d_H, e2, d_prec, d_diagH);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
