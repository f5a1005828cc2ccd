// Dense SPD solve on gfx950: blocked right-looking Cholesky with the forward
// substitution carried along as an extra matrix row, then a blocked backward
// substitution.  Replaces `coeff += solve(h, r)` of lpdf::optnewton
// (src/fit.cpp:120; Armadillo -> LAPACK in the reference).
//
// H is p x p, row-major, full symmetric storage on entry; on exit its lower
// triangle holds L (H = L L^T), the strict upper triangle is scratch.
//
// Per 64-column block step j:
//   k_chol_panel : every workgroup (one wave) re-factorises the 64 x 64
//                  diagonal block in registers (lane = row, v_readlane
//                  broadcasts; ~2k dependent FMAs, cheaper than a launch
//                  boundary), then solves its 64 panel rows against L_jj^T.
//                  One extra "row" is the right-hand side z, which turns the
//                  forward substitution L z = rhs into part of the panel solve.
//   k_chol_update: trailing update A22 -= L21 L21^T (lower tiles only) on
//                  v_mfma_f64_16x16x4_f64, 128 x 128 tiles, K = 64; plus the
//                  matching update of z.
// Backward: k_chol_back per block from the last to the first: theta_j =
// L_jj^-T z_j, then z[0:j) -= L[j, 0:j)^T theta_j.
#include "obhip_internal.h"

namespace obhip {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = 64;
constexpr int LDP = NB + 1;  // padded LDS leading dimension

__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(64)
k_chol_panel(double *__restrict__ H, double *__restrict__ z, int p, int j0, int *__restrict__ info) {
  __shared__ double Ld[NB * LDP];
  __shared__ double P[NB * LDP];
  const int lane = threadIdx.x;
  const int jb = min(NB, p - j0);

  // diagonal block -> LDS (identity padding beyond jb)
  // (branch-free clamped addresses, 16 row loads in flight per batch: a load /
  // wait / store per row would cost one L2 round trip per row)
  {
    const double *src = H + (size_t)j0 * p + j0 + min(lane, jb - 1);
#pragma unroll
    for (int rb = 0; rb < NB; rb += 16) {
      double t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = src[(size_t)min(rb + i, jb - 1) * p];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = rb + i;
        Ld[r * LDP + lane] = (r < jb && lane < jb) ? t[i] : ((r == lane) ? 1.0 : 0.0);
      }
    }
  }
  __syncthreads();
  double a[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) a[k] = Ld[lane * LDP + k];

  // unblocked right-looking Cholesky, lane i owns row i
  bool bad = false;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const double piv = readlane_d(a[c], c);
    if (!(piv > 0.0)) bad = true;
    const double dinv = 1.0 / sqrt(piv);
    const double lc = (lane == c) ? sqrt(piv) : a[c] * dinv;
    a[c] = lc;
#pragma unroll
    for (int k = c + 1; k < NB; ++k) {
      const double lk = readlane_d(lc, k);
      a[k] = fma(-lc, lk, a[k]);
    }
  }
  if (bad && blockIdx.x == 0 && lane == 0) atomicMax(info, j0 + 1);
#pragma unroll
  for (int k = 0; k < NB; ++k) Ld[lane * LDP + k] = a[k];
  __syncthreads();

  if (blockIdx.x == 0) {
    // write L_jj (lower triangle) back
    for (int r = 0; r < jb; ++r)
      if (lane <= r) H[(size_t)(j0 + r) * p + j0 + lane] = Ld[r * LDP + lane];
    return;
  }

  // panel rows of this workgroup; the last workgroup carries the rhs row z
  const bool is_z = blockIdx.x == gridDim.x - 1;
  const int r0 = j0 + NB * (int)blockIdx.x;
  const int nrows = is_z ? 1 : min(NB, p - r0);
  {
    const int lc = min(lane, jb - 1);
    const double *src = is_z ? z + j0 + lc : H + (size_t)min(r0, p - 1) * p + j0 + lc;
    const int rmax = is_z ? 0 : min(NB, p - r0) - 1;  // clamp row offsets into the matrix
#pragma unroll
    for (int rb = 0; rb < NB; rb += 16) {
      double t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = src[(size_t)min(rb + i, max(rmax, 0)) * p];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = rb + i;
        P[r * LDP + lane] = (r < nrows && lane < jb) ? t[i] : 0.0;
      }
    }
  }
  __syncthreads();
  double x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) x[c] = P[lane * LDP + c];
  // x L_jj^T = h  (row-wise forward substitution)
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double s = x[c];
#pragma unroll
    for (int k = 0; k < c; ++k) s = fma(-x[k], Ld[c * LDP + k], s);
    x[c] = s / Ld[c * LDP + c];
  }
#pragma unroll
  for (int c = 0; c < NB; ++c) P[lane * LDP + c] = x[c];
  __syncthreads();
  for (int r = 0; r < nrows; ++r)
    if (lane < jb) {
      const double v = P[r * LDP + lane];
      if (is_z)
        z[j0 + lane] = v;
      else
        H[(size_t)(r0 + r) * p + j0 + lane] = v;
    }
}

__global__ void __launch_bounds__(256)
k_chol_update(double *__restrict__ H, double *__restrict__ z, int p, int j0, int nt, int npairs) {
  const int t0 = j0 + NB;
  if ((int)blockIdx.x >= npairs) {
    // z[c] -= sum_k z[j0 + k] * L[c][j0 + k]
    const int c = t0 + ((int)blockIdx.x - npairs) * 256 + (int)threadIdx.x;
    if (c < p) {
      double s = z[c];
      const double *lrow = H + (size_t)c * p + j0;
      for (int k = 0; k < NB; ++k) s = fma(-z[j0 + k], lrow[k], s);
      z[c] = s;
    }
    return;
  }
  // lower-triangular tile pair (bi >= bj)
  int bi = 0, rem = blockIdx.x;
  while (rem > bi) {
    rem -= bi + 1;
    ++bi;
  }
  const int bj = rem;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  if (bi == bj && wn > wm) return;  // strictly upper 64 x 64 quadrant
  const int t16 = lane & 15, q = lane >> 4;
  const int rbase = t0 + bi * 128 + wm * 64, cbase = t0 + bj * 128 + wn * 64;
  const double *pa[4], *pb[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    pa[f] = H + (size_t)min(p - 1, rbase + f * 16 + t16) * p + j0 + q;
    pb[f] = H + (size_t)min(p - 1, cbase + f * 16 + t16) * p + j0 + q;
  }
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    double a[4], b[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      a[f] = pa[f][4 * s];
      b[f] = pb[f][4 * s];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + i * 16 + q + 4 * r;
        const int col = cbase + j * 16 + t16;
        if (row < p && col < p && col <= row) H[(size_t)row * p + col] -= acc[i][j][r];
      }
}

__global__ void __launch_bounds__(256)
k_chol_back(const double *__restrict__ L, double *__restrict__ z, double *__restrict__ theta, int p,
            int j0) {
  __shared__ double Ld[NB * LDP];
  __shared__ double th[NB];
  const int jb = min(NB, p - j0);
  {
    double t[NB * NB / 256];
#pragma unroll
    for (int i = 0; i < NB * NB / 256; ++i) {
      const int e = threadIdx.x + i * 256;
      t[i] = L[(size_t)(j0 + min(e / NB, jb - 1)) * p + j0 + min(e % NB, jb - 1)];
    }
#pragma unroll
    for (int i = 0; i < NB * NB / 256; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / NB, c = e % NB;
      Ld[r * LDP + c] = (r < jb && c < jb && c <= r) ? t[i] : ((r == c) ? 1.0 : 0.0);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double zk = lane < jb ? z[j0 + lane] : 0.0;
    // L_jj^T theta = z, from the last unknown to the first
    for (int c = NB - 1; c >= 0; --c) {
      const double tc = readlane_d(zk, c) / Ld[c * LDP + c];
      if (lane == c) zk = tc;
      if (lane < c) zk = fma(-Ld[c * LDP + lane], tc, zk);
    }
    th[lane] = zk;
    if (blockIdx.x == 0 && lane < jb) theta[j0 + lane] = zk;
  }
  __syncthreads();
  const int c = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (c < j0) {
    double s = z[c];
    for (int k = 0; k < jb; ++k) s = fma(-L[(size_t)(j0 + k) * p + c], th[k], s);
    z[c] = s;
  }
}

__global__ void k_form_hessian(double *__restrict__ G, const double *__restrict__ prec, double e2,
                               int p, double *__restrict__ diagH) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)p * p) return;
  const int i = (int)(idx / p), j = (int)(idx % p);
  double v = e2 * G[idx];
  if (i == j) {
    v += prec[i];
    if (diagH) diagH[i] = v;
  }
  G[idx] = v;
}

}  // namespace

uint64_t newton_workspace_bytes(uint64_t p) { return (p + 64) * sizeof(double); }

int launch_form_hessian(uint64_t p, double *d_G, const double *d_prec, double e2, double *d_diagH) {
  ProfScope ps("form_hessian");
  const size_t total = (size_t)p * p;
  hipLaunchKernelGGL(k_form_hessian, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     cur_stream(), d_G, d_prec, e2, (int)p, d_diagH);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_newton_solve(uint64_t p64, double *d_H, const double *d_rhs, double *d_theta, void *d_ws,
                        uint64_t ws_bytes) {
  if (ws_bytes < newton_workspace_bytes(p64)) return fail(OBHIP_ERR_INVALID, "workspace too small");
  if (p64 > (1u << 30)) return fail(OBHIP_ERR_INVALID, "p too large");
  const int p = (int)p64;
  double *z = (double *)d_ws;
  int *info = (int *)(z + p);
  hipStream_t st = cur_stream();
  OB_HIP(hipMemcpyAsync(z, d_rhs, sizeof(double) * p, hipMemcpyDeviceToDevice, st));
  OB_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
  {
    ProfScope ps("cholesky");
    for (int j0 = 0; j0 < p; j0 += NB) {
      const int nrowblk = (p - j0 + NB - 1) / NB;  // block 0 = diagonal block
      hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)(nrowblk + 1)), dim3(64), 0, st, d_H, z, p, j0,
                         info);
      const int m = p - (j0 + NB);
      if (m > 0) {
        const int nt = (m + 127) / 128;
        const int npairs = nt * (nt + 1) / 2;
        const int nz = (m + 255) / 256;
        hipLaunchKernelGGL(k_chol_update, dim3((unsigned)(npairs + nz)), dim3(256), 0, st, d_H, z, p,
                           j0, nt, npairs);
      }
    }
    OB_HIP(hipGetLastError());
  }
  {
    ProfScope ps("backsolve");
    const int last = (p - 1) / NB * NB;
    for (int j0 = last; j0 >= 0; j0 -= NB) {
      const int nblk = std::max(1, (j0 + 255) / 256);
      hipLaunchKernelGGL(k_chol_back, dim3((unsigned)nblk), dim3(256), 0, st, d_H, z, d_theta, p, j0);
    }
    OB_HIP(hipGetLastError());
  }
  int h_info = 0;
  OB_HIP(hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, st));
  OB_HIP(hipStreamSynchronize(st));
  if (h_info != 0)
    return fail(OBHIP_ERR_NUMERIC, "Hessian is not positive definite (block starting at column " +
                                       std::to_string(h_info - 1) + ")");
  return 0;
}

}  // namespace obhip
