// X = L^-T for the Cholesky factor L of the total Hessian: the explicit triangular inverse
// behind predr_std's posterior variance b_i^T inv(H) b_i = || L^-1 b_i ||^2 (the reference
// forms inv(tothess) with arma::inv, src/lpdfs/loglik_std.cpp:227, and multiplies the design
// matrix by it, :251-255).  With X in hand the p^2 n' flop of the variance are one pass of the
// matrix-core kernel of kernels_gram_panel.hip (A^T Bm with the row norms as its epilogue)
// instead of a triangular solve with n' right-hand sides.
//
// L: lower triangle of a row-major p x p buffer with pitch ldl (what launch_newton_solve
// leaves in H).  X: row-major pp x pp (pp = p rounded up to 128, zero outside the upper
// triangle), X[c][i] = inv(L)[i][c] for c <= i.
//
// Blocked by 64: with W = inv(L), W_jj = inv(L_jj) and for i > j
//   W_ij = -inv(L_ii) sum_{k = j .. i - 1} L_ik W_kj.
//   k_trtri_diag  one wave per diagonal block: lane = column of the inverse, forward
//                 substitution over the rows with the block in LDS (broadcast reads);
//   k_trtri_cols  workgroup (j, s) owns the 16-column slice s of block column j of W and
//                 walks i = j + 1 ... : 4 waves split the k range of the sum, operands
//                 straight from L and from the part of X the workgroup itself wrote in
//                 earlier steps (v_mfma_f64_16x16x4_f64; a lane takes 16 CONSECUTIVE k, so
//                 every operand load is a 128-byte run), partial sums meet in LDS, then the
//                 64 x 64 by 64 x 16 product with inv(L_ii).
#include "obhip_internal.h"

namespace obhip {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int TB = 64;

__global__ void __launch_bounds__(64)
k_trtri_diag(const double *__restrict__ L, uint64_t ldl, int p, double *__restrict__ Dinv,
             double *__restrict__ X, uint64_t ldx) {
  __shared__ double Ls[TB * (TB + 1)];
  const int lane = threadIdx.x, j0 = blockIdx.x * TB, jb = min(TB, p - j0);
  for (int r = 0; r < TB; ++r) {
    double v = (r == lane) ? 1.0 : 0.0;  // identity padding beyond p
    if (r < jb && lane < jb && lane <= r) v = L[(uint64_t)(j0 + r) * ldl + j0 + lane];
    Ls[r * (TB + 1) + lane] = v;
  }
  __syncthreads();
  // lane c solves L x = e_c; x[r] = 0 for r < c
  double x[TB];
#pragma unroll
  for (int r = 0; r < TB; ++r) {
    double s = (r == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) s = fma(-Ls[r * (TB + 1) + k], x[k], s);  // broadcast reads
    x[r] = (r >= lane) ? s / Ls[r * (TB + 1) + r] : 0.0;
  }
  // Dinv[j]: inv(L_jj) row-major [r][c]; X diagonal block transposed: X[j0 + c][j0 + r]
  double *D = Dinv + (uint64_t)blockIdx.x * TB * TB;
#pragma unroll
  for (int r = 0; r < TB; ++r) {
    D[r * TB + lane] = x[r];
    if (r < jb && lane < jb) X[(uint64_t)(j0 + lane) * ldx + j0 + r] = x[r];
  }
}

__global__ void __launch_bounds__(256)
k_trtri_cols(const double *__restrict__ L, uint64_t ldl, int p, const double *__restrict__ Dinv,
             double *__restrict__ X, uint64_t ldx, int nb) {
  __shared__ double Tred[4][TB][17];  // per-wave partial T (64 x 16)
  __shared__ double Ts[TB][17];       // T = sum over the waves
  const int j = blockIdx.x, slice = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t16 = lane & 15, q = lane >> 4;
  const int c0 = j * TB + slice * 16;  // first column of the slice (row of X)
  if (c0 >= p) return;
  for (int i = j + 1; i < nb; ++i) {
    const int i0 = i * TB;
    // T_w = sum over this wave's k blocks of L_ik W_kj[:, slice]
    d4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = d4{0.0, 0.0, 0.0, 0.0};
    for (int kb = j + wave; kb < i; kb += 4) {
      const int k0 = kb * TB + q * 16;  // this lane's 16 consecutive k
      // B operand: W_kj[k][n] = X[c0 + n][k]
      double bv[16];
      {
        const double *src = X + (uint64_t)min(c0 + t16, p - 1) * ldx + k0;
#pragma unroll
        for (int s = 0; s < 16; ++s) bv[s] = src[s];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int row = min(i0 + m * 16 + t16, p - 1);
        const double *src = L + (uint64_t)row * ldl + k0;
        double av[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) av[s] = src[s];
#pragma unroll
        for (int s = 0; s < 16; ++s)
          acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc[m], 0, 0, 0);
      }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) Tred[wave][m * 16 + q + 4 * r][t16] = acc[m][r];
    __syncthreads();
    for (int e = tid; e < TB * 16; e += 256) {
      const int r = e >> 4, c = e & 15;
      Ts[r][c] = Tred[0][r][c] + Tred[1][r][c] + Tred[2][r][c] + Tred[3][r][c];
    }
    __syncthreads();
    // W_ij[:, slice] = -inv(L_ii) T: wave w makes rows 16 w .. 16 w + 15
    {
      const double *D = Dinv + (uint64_t)i * TB * TB;
      d4 o = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double av = D[(wave * 16 + t16) * TB + 4 * s + q];  // inv(L_ii)[m][k]
        const double bv = Ts[4 * s + q][t16];                     // T[k][n]
        o = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, o, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + wave * 16 + q + 4 * r, col = c0 + t16;  // W[row][col]
        if (row < p && col < p) X[(uint64_t)col * ldx + row] = -o[r];
      }
    }
    // the next step reads what this one wrote (same workgroup): make it visible
    __threadfence();
    __syncthreads();
  }
}

// out[c][r] = in[r][c], 64 x 64 tiles through LDS
__global__ void __launch_bounds__(256)
k_transpose(const double *__restrict__ in, uint64_t ldi, double *__restrict__ out, uint64_t ldo,
            int n) {
  __shared__ double S[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  for (int r = r4; r < 64; r += 4) {
    const int gr = bi * 64 + r, gc = bj * 64 + c;
    S[r][c] = (gr < n && gc < n) ? in[(uint64_t)gr * ldi + gc] : 0.0;
  }
  __syncthreads();
  for (int r = r4; r < 64; r += 4) {
    const int gr = bj * 64 + r, gc = bi * 64 + c;
    if (gr < n && gc < n) out[(uint64_t)gr * ldo + gc] = S[c][r];
  }
}

}  // namespace

// d_X: pp x pp doubles (pp = multiple of 128 >= p), overwritten: zero + L^-T in its upper
// triangle.  d_dinv: ((p + 63) / 64) * 4096 doubles of scratch.
int launch_trtri_lt(const double *d_L, uint64_t ldl, uint64_t p64, double *d_X, uint64_t pp,
                    double *d_dinv) {
  const int p = (int)p64, nb = (p + TB - 1) / TB;
  hipStream_t st = cur_stream();
  OB_HIP(hipMemsetAsync(d_X, 0, pp * pp * sizeof(double), st));
  ProfScope ps("predict_std_inverse");
  hipLaunchKernelGGL(k_trtri_diag, dim3((unsigned)nb), dim3(64), 0, st, d_L, ldl, p, d_dinv, d_X, pp);
  if (nb > 1)
    hipLaunchKernelGGL(k_trtri_cols, dim3((unsigned)nb, 4), dim3(256), 0, st, d_L, ldl, p, d_dinv, d_X,
                       pp, nb);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_transpose(const double *d_in, uint64_t ldi, double *d_out, uint64_t ldo, uint64_t n) {
  const unsigned nt = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(k_transpose, dim3(nt, nt), dim3(256), 0, cur_stream(), d_in, ldi, d_out, ldo,
                     (int)n);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
