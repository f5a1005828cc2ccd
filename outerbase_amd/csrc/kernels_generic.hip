// Generic forms of the outer-product kernels for term sets the LDS-tiled kernels cannot take:
// the fast kernels (kernels_prod.hip, kernels_predict.hip, kernels_gram_panel.hip) stage the
// basis columns a term set USES for 64 rows in LDS, which bounds that number (Mu) at about
// 300 (152 KiB / 512 B), and the term-per-lane kernels keep at most 8 factors per term in
// registers.  The reference has neither limit (prodmm_ / tprodmm_ / getm_ walk the umat,
// src/linalg.cpp:57-131, 286-355, 647-715), so wide problems -- a few levels in each of a
// hundred dimensions, or terms with many factors -- take these kernels instead: the same
// products with every basis column read straight from the tile-blocked basemat in HBM / L2
// (a lane owns a row, so a wave reads one 512-byte column run per factor).  Any Mu up to the
// 65535 the uint16 column lists can index, any number of factors.  Slower than the tiled
// kernels (no reuse of a staged tile across terms beyond what L2 gives), correct everywhere.
#include "obhip_internal.h"

namespace obhip {

namespace {

constexpr int kGW = 4;  // waves per block

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// product of the W factors of term k at this lane's row; tile: bm + tile * Mc * 64
template <bool SQ>
__device__ __forceinline__ double term_prod_hbm(const double *__restrict__ tile, int lane,
                                                const uint16_t *__restrict__ cols,
                                                const uint32_t *__restrict__ ucol, int W, int k) {
  double v = 1.0;
  for (int j = 0; j < W; ++j) {
    const int u = __builtin_amdgcn_readfirstlane((int)cols[(size_t)k * W + j]);
    const int c = __builtin_amdgcn_readfirstlane((int)ucol[u]);
    const double x = tile[(size_t)c * kTileRows + lane];
    v *= SQ ? x * x : x;
  }
  return v;
}

// MODE 0: out = B a, 1: out = B^2 a, 2: out = B (column-major n x p, ob$getmat)
template <int MODE>
__global__ void __launch_bounds__(kGW * 64)
k_mm_generic(const double *__restrict__ bm, const double *__restrict__ scale, uint64_t Mc,
             const uint16_t *__restrict__ cols, const uint32_t *__restrict__ ucol, int W, int p,
             const double *__restrict__ a, uint64_t n, uint64_t ld, double *__restrict__ out) {
  __shared__ double red[kGW][kTileRows];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = blockIdx.x, row = tile * kTileRows + lane;
  const double *tb = bm + tile * Mc * kTileRows;
  const double s = scale[row];  // 0 in the padding rows
  double acc = 0.0;
  for (int k = wave; k < p; k += kGW) {
    const double v = term_prod_hbm<MODE == 1>(tb, lane, cols, ucol, W, k);
    if constexpr (MODE == 2) {
      if (row < n) out[(uint64_t)k * ld + row] = s * v;
    } else {
      acc = fma(a[k], v, acc);
    }
  }
  if constexpr (MODE != 2) {
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && row < n) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < kGW; ++w) tot += red[w][lane];
      out[row] = (MODE == 1 ? s * s : s) * tot;
    }
  }
}

// part[split][k] = sum over the split's rows of w_i s_i prod_k(i)   (SQ: s_i^2 prod^2)
template <bool SQ>
__global__ void __launch_bounds__(kGW * 64)
k_tmm_generic(const double *__restrict__ bm, const double *__restrict__ scale, uint64_t Mc,
              const uint16_t *__restrict__ cols, const uint32_t *__restrict__ ucol, int W, int p,
              const double *__restrict__ w, uint64_t n, uint64_t ntiles, uint64_t tps,
              uint64_t p_pad, double *__restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = blockIdx.y * kGW + wave;
  if (k >= p) return;
  const uint64_t t0 = (uint64_t)blockIdx.x * tps, t1 = min(ntiles, t0 + tps);
  double acc = 0.0;
  for (uint64_t tile = t0; tile < t1; ++tile) {
    const uint64_t row = tile * kTileRows + lane;
    const double s = scale[row];
    const double wi = row < n ? w[row] : 0.0;
    const double v = term_prod_hbm<SQ>(bm + tile * Mc * kTileRows, lane, cols, ucol, W, k);
    acc = fma(wi * (SQ ? s * s : s), v, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) part[(uint64_t)blockIdx.x * p_pad + k] = acc;
}

__global__ void __launch_bounds__(256)
k_tmm_generic_reduce(const double *__restrict__ part, int nsplit, uint64_t p_pad, int p,
                     double *__restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= p) return;
  double s = 0.0;
  for (int q = 0; q < nsplit; ++q) s += part[(uint64_t)q * p_pad + k];
  out[k] = s;
}

// B row-major [n_pad][p_pad]: lane = term, a wave writes 512-byte row segments
__global__ void __launch_bounds__(kGW * 64)
k_materialize_generic(const double *__restrict__ bm, const double *__restrict__ scale, uint64_t Mc,
                      const uint16_t *__restrict__ cols, const uint32_t *__restrict__ ucol, int W,
                      int p, uint64_t p_pad, double *__restrict__ B) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint64_t tile = blockIdx.x;
  const int k = (blockIdx.y * kGW + wave) * 64 + lane;
  if ((uint64_t)(blockIdx.y * kGW + wave) * 64 >= p_pad) return;
  const double *tb = bm + tile * Mc * kTileRows;
  const bool live = k < p;
  for (int r = 0; r < kTileRows; ++r) {
    double v = live ? scale[tile * kTileRows + r] : 0.0;
    if (live)
      for (int j = 0; j < W; ++j) v *= tb[(size_t)ucol[cols[(size_t)k * W + j]] * kTileRows + r];
    B[(tile * kTileRows + r) * p_pad + k] = v;
  }
}

}  // namespace

// t must be prepared for b (obhip_terms::prepare)
int launch_mm_generic(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, int mode,
                      uint64_t ld) {
  if (ld == 0) ld = b.n;
  const dim3 grid((unsigned)(b.n_pad / kTileRows));
#define OB_GMM(M_)                                                                                 \
  hipLaunchKernelGGL(k_mm_generic<M_>, grid, dim3(kGW * 64), 0, cur_stream(), b.bm.p, b.scale.p,    \
                     b.md.Mc, t.cols.p, t.ucol.p, (int)t.W, (int)t.p, d_a, b.n, ld, d_out)
  if (mode == 0) OB_GMM(0);
  else if (mode == 1) OB_GMM(1);
  else OB_GMM(2);
#undef OB_GMM
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_tmm_generic(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out,
                       bool squared) {
  const uint64_t ntiles = b.n_pad / kTileRows;
  const uint64_t yblocks = (t.p + kGW - 1) / kGW;
  uint64_t nsplit = std::max<uint64_t>(1, 4096 / std::max<uint64_t>(1, yblocks));
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(const_cast<obhip_basis &>(b).workspace(nsplit * t.p_pad * sizeof(double), (void **)&part));
  const dim3 grid((unsigned)nsplit, (unsigned)yblocks);
  if (squared)
    hipLaunchKernelGGL(k_tmm_generic<true>, grid, dim3(kGW * 64), 0, cur_stream(), b.bm.p, b.scale.p,
                       b.md.Mc, t.cols.p, t.ucol.p, (int)t.W, (int)t.p, d_a, b.n, ntiles, tps, t.p_pad,
                       part);
  else
    hipLaunchKernelGGL(k_tmm_generic<false>, grid, dim3(kGW * 64), 0, cur_stream(), b.bm.p, b.scale.p,
                       b.md.Mc, t.cols.p, t.ucol.p, (int)t.W, (int)t.p, d_a, b.n, ntiles, tps, t.p_pad,
                       part);
  hipLaunchKernelGGL(k_tmm_generic_reduce, dim3((unsigned)((t.p + 255) / 256)), dim3(256), 0,
                     cur_stream(), part, (int)nsplit, t.p_pad, (int)t.p, d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_materialize_generic(const obhip_basis &b, obhip_terms &t, double *d_B) {
  const dim3 grid((unsigned)(b.n_pad / kTileRows), (unsigned)((t.p_pad + kGW * 64 - 1) / (kGW * 64)));
  hipLaunchKernelGGL(k_materialize_generic, grid, dim3(kGW * 64), 0, cur_stream(), b.bm.p, b.scale.p,
                     b.md.Mc, t.cols.p, t.ucol.p, (int)t.W, (int)t.p, t.p_pad, d_B);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
