// G = B^T B: dispatch over the Gram back ends and the reduction of their row-split partials.
//
// Replaces loglik_std::hess (src/lpdfs/loglik_std.cpp:170-173, the basismat.t() * basismat
// gemm on the n x p design matrix built by getm_, src/linalg.cpp:647-715).
//
//   default  kernels_gram_panel.hip: v_mfma_f64_4x4x4_4b_f64 tiles fed by LDS-direct loads
//            from a row-major design matrix staged in HBM -- all rows at once when n x p
//            doubles fit in half of the free memory, in row chunks otherwise;
//   3        kernels_gram_mfma4.hip: the same matrix-core tiles with the operand panels
//            generated inside the kernel from the factored basis (no staging memory; terms of
//            at most 8 factors on at most 128 used columns).
// Every back end covers the upper-triangular 128 x 128 tile pairs, splits the rows over
// several workgroups per pair, and writes per-block partial tiles; k_gram_reduce sums the
// partials in a fixed order (bit-reproducible, no atomics) and mirrors the result into the
// lower triangle.
#include "obhip_internal.h"

namespace obhip {

namespace {

constexpr int kGT = 128;  // output tile edge (terms)

// sum the row-split partials and scatter tile (I, J) and its mirror into the full symmetric
// p x p matrix; ACC: add to what G holds (later row chunks of the chunked panel back end)
template <bool ACC>
__global__ void __launch_bounds__(256)
k_gram_reduce(const double *__restrict__ part, int npairs, int nsplit, int nb, int p,
              double *__restrict__ G) {
  __shared__ double S[64 * 65];  // one 64 x 64 quadrant, so that the mirror goes out in rows too
  int I = 0, rem = blockIdx.x;
  while (rem >= nb - I) {
    rem -= nb - I;
    ++I;
  }
  const int J = I + rem;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  for (int qd = 0; qd < 4; ++qd) {
    const int qr = qd >> 1, qc = qd & 1;
    if (I == J && qr > qc) continue;  // diagonal tiles: the lower-left quadrant is the mirror
    // sum of the row-split partials, 64 consecutive doubles per wave load
    for (int r = r4; r < 64; r += 4) {
      const int e = (qr * 64 + r) * kGT + qc * 64 + c;
      double s = 0.0;
#pragma unroll 8
      for (int k = 0; k < nsplit; ++k) s += part[((uint64_t)k * npairs + blockIdx.x) * (kGT * kGT) + e];
      S[r * 65 + c] = s;
    }
    __syncthreads();
    const int gi0 = I * kGT + qr * 64, gj0 = J * kGT + qc * 64;
    for (int r = r4; r < 64; r += 4) {
      // G[gi0 + r][gj0 + c] and, mirrored, G[gj0 + r][gi0 + c] = S[c][r]
      if (gi0 + r < p && gj0 + c < p && !(I == J && qr == qc && c < r)) {
        double *g = &G[(uint64_t)(gi0 + r) * p + gj0 + c];
        *g = ACC ? *g + S[r * 65 + c] : S[r * 65 + c];
      }
      if (gj0 + r < p && gi0 + c < p && !(I == J && qr == qc && c >= r)) {
        double *g = &G[(uint64_t)(gj0 + r) * p + gi0 + c];
        *g = ACC ? *g + S[c * 65 + r] : S[c * 65 + r];
      }
    }
    __syncthreads();
  }
}

}  // namespace

int launch_gram_reduce(const double *part, int npairs, int nsplit, int nb, int p, double *d_G,
                       bool accumulate) {
  ProfScope ps("gram_reduce");
  if (accumulate)
    hipLaunchKernelGGL(k_gram_reduce<true>, dim3((unsigned)npairs), dim3(256), 0, cur_stream(), part,
                       npairs, nsplit, nb, p, d_G);
  else
    hipLaunchKernelGGL(k_gram_reduce<false>, dim3((unsigned)npairs), dim3(256), 0, cur_stream(), part,
                       npairs, nsplit, nb, p, d_G);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_gram_mfma4(const obhip_basis &b, obhip_terms &t, double *d_G);
bool gram_mfma4_supports(const obhip_terms &t);
int launch_gram_panel(const obhip_basis &b, obhip_terms &t, double *d_G);

// 0 / 4 = staged design matrix (whole or in row chunks), 3 = fused kernel
static int g_gram_backend = 0;
void set_gram_backend(int b) { g_gram_backend = b; }
int get_gram_backend() { return g_gram_backend; }

int launch_gram(const obhip_basis &b, obhip_terms &t, double *d_G) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (g_gram_backend == 3) {
    if (!gram_mfma4_supports(t))
      return fail(OBHIP_ERR_INVALID,
                  "fused Gram kernel: terms of at most 8 factors on at most 128 used basis columns");
    return launch_gram_mfma4(b, t, d_G);
  }
  return launch_gram_panel(b, t, d_G);
}

}  // namespace obhip
