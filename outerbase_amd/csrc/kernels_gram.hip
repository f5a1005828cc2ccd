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

// Sum the row-split partials of tile pair (I, J) and write it where the fit wants it (GramSink):
//   full symmetric p x p (the tile and its mirror) -- raw G, or, with `form`, the Hessian
//       H = e2 G + diag(prec) of lpdfvec::hess_ (fit.cpp:503-512, loglik_std.cpp:170-173,
//       logpr_gauss.cpp:153-158) and its diagonal, so that no further pass over p x p follows;
//   packed upper triangle (row i: entries j >= i at i p - i (i - 1) / 2 + (j - i)) -- the
//       exchange buffer of a row-sharded fit, written here directly.
// acc: add to what the destination holds (later row chunks of the chunked panel back end);
// form then applies to the accumulated sum (last chunk only).
__device__ __forceinline__ uint64_t tri_row(uint64_t i, uint64_t p) { return i * p - i * (i - 1) / 2; }

// Grid: (tile pairs, 4 quadrants x 4 slices of 16 rows).  Until round 4 a workgroup walked the four
// quadrants of its tile pair one after the other: 36 workgroups at p = 1024 (BASELINE configs[1]),
// each a chain of 4 x 16 x nsplit dependent loads -- 0.22 ms for 67 MB.  Same sums in the same order
// per element; only who computes them changed.
constexpr int kRedRows = 16;  // rows of a quadrant per workgroup
__global__ void __launch_bounds__(256)
k_gram_reduce(const double *__restrict__ part, int npairs, int nsplit, int nb, int p,
              double *__restrict__ G, int acc, int packed, int form, double e2,
              const double *__restrict__ prec, double *__restrict__ diagH) {
  __shared__ double S[kRedRows * 65];  // the slice, so that the mirror goes out in row segments too
  int I = 0, rem = blockIdx.x;
  while (rem >= nb - I) {
    rem -= nb - I;
    ++I;
  }
  const int J = I + rem;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  const int qd = (int)blockIdx.y / (64 / kRedRows), r0 = ((int)blockIdx.y % (64 / kRedRows)) * kRedRows;
  const int qr = qd >> 1, qc = qd & 1;
  if (I == J && qr > qc) return;  // diagonal tiles: the lower-left quadrant is the mirror
  const bool dq = I == J && qr == qc;  // quadrant on the diagonal of G
  const int gi0 = I * kGT + qr * 64, gj0 = J * kGT + qc * 64;
  // sum of the row-split partials, 64 consecutive doubles per wave load
  for (int r = r0 + r4; r < r0 + kRedRows; r += 4) {
    const int e = (qr * 64 + r) * kGT + qc * 64 + c;
    double s = 0.0;
#pragma unroll 8
    for (int k = 0; k < nsplit; ++k) s += part[((uint64_t)k * npairs + blockIdx.x) * (kGT * kGT) + e];
    const bool in = gi0 + r < p && gj0 + c < p;
    if (in && !(dq && c < r)) {  // the entry (gi0 + r, gj0 + c), j >= i
      const uint64_t i = (uint64_t)(gi0 + r), j = (uint64_t)(gj0 + c);
      double *g = packed ? &G[tri_row(i, p) + (j - i)] : &G[i * p + j];
      if (acc) s += *g;
      if (form) {
        s *= e2;
        if (i == j) {
          s += prec[i];
          if (diagH) diagH[i] = s;
        }
      }
      *g = s;
    }
    S[(r - r0) * 65 + c] = s;
  }
  if (packed) return;  // no mirror, and S is not read
  __syncthreads();
  // mirrored: G[gj0 + r][gi0 + r0 + cc] = S[cc][r], strictly below the diagonal of G only; a thread
  // takes column cc of the slice (16 consecutive doubles of a row of G per 16 threads)
  const int cc = threadIdx.x & (kRedRows - 1);
  for (int r = threadIdx.x / kRedRows; r < 64; r += 256 / kRedRows) {
    if (gj0 + r < p && gi0 + r0 + cc < p && !(dq && r0 + cc >= r))
      G[(uint64_t)(gj0 + r) * p + gi0 + r0 + cc] = S[cc * 65 + r];
  }
}

}  // namespace

int launch_gram_reduce(const double *part, int npairs, int nsplit, int nb, int p, const GramSink &sink,
                       bool accumulate, bool last) {
  ProfScope ps("gram_reduce");
  const int form = sink.form && last ? 1 : 0;
  if (form && !sink.prec) return fail(OBHIP_ERR_INVALID, "gram sink: form without the prior precisions");
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)npairs, 4 * (64 / kRedRows)), dim3(256), 0, cur_stream(), part, npairs,
                     nsplit, nb, p, sink.out, accumulate ? 1 : 0, sink.packed ? 1 : 0, form, sink.e2,
                     sink.prec, sink.diagH);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_gram_mfma4(const obhip_basis &b, obhip_terms &t, const GramSink &sink);
bool gram_mfma4_supports(const obhip_terms &t);
int launch_gram_panel(const obhip_basis &b, obhip_terms &t, const GramSink &sink, GramFuse *fuse);

// 0 / 4 = staged design matrix (whole or in row chunks), 3 = fused kernel
static int g_gram_backend = 0;
void set_gram_backend(int b) { g_gram_backend = b; }
int get_gram_backend() { return g_gram_backend; }

int launch_gram(const obhip_basis &b, obhip_terms &t, double *d_G) {
  GramSink sink;
  sink.out = d_G;
  return launch_gram_to(b, t, sink);
}

int launch_gram_to(const obhip_basis &b, obhip_terms &t, const GramSink &sink, GramFuse *fuse) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (g_gram_backend == 3) {
    if (!gram_mfma4_supports(t))
      return fail(OBHIP_ERR_INVALID,
                  "fused Gram kernel: terms of at most 8 factors on at most 128 used basis columns");
    return launch_gram_mfma4(b, t, sink);
  }
  return launch_gram_panel(b, t, sink, fuse);
}

}  // namespace obhip
