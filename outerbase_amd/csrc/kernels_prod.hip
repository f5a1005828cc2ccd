// Matrix-free outer-product kernels for gfx950:
//   getmat : B (n x p) materialised           getm_    src/linalg.cpp:647-715
//   mm     : out = B a   (and B^2 a)          prodmm_  src/linalg.cpp:57-131
//   tmm    : out = B^T a (and (B^2)^T a)      tprodmm_ src/linalg.cpp:286-355
// with B[i,k] = basescale[i] * prod_{l: t_kl>0} basemat[i, col(l, t_kl)].
//
// All three stage one 64-row tile of the used basemat columns in LDS
// ([column][row], one 512-byte run per column, straight from the tile-blocked
// HBM layout) and map lane = row, so every LDS read in the Hadamard product is
// a conflict-free ds_read_b64 with a wave-uniform column, and the term tables
// (column lists, coefficients) are wave-uniform scalar loads.
//
// tmm keeps a 64-term x 64-row block of partial sums in registers per wave
// (acc[t], lane = row mod 64) across all its row tiles and only reduces across
// lanes once at the end, so the per-element cost equals mm's.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

constexpr int kMaxMuLds = 304;  // 304 * 64 * 8 B = 152 KiB of the 160 KiB LDS

// product of the staged columns of term k for this lane's row
__device__ __forceinline__ double term_prod(const double *__restrict__ lds,
                                            const uint32_t *__restrict__ cw, int W2, int lane,
                                            double v) {
  for (int w = 0; w < W2; ++w) {
    const uint32_t c = cw[w];
    v *= lds[(c & 0xffffu) * kTileRows + lane];
    v *= lds[(c >> 16) * kTileRows + lane];
  }
  return v;
}

template <bool SQ>
__global__ void __launch_bounds__(256)
k_mm(const double *__restrict__ bm, const double *__restrict__ scale,
     const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc, const uint32_t *__restrict__ colsw,
     int W2, int p, const double *__restrict__ a, double *__restrict__ out, uint64_t n) {
  extern __shared__ double lds[];
  double *red = lds + (size_t)Mu * kTileRows;  // [4][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = blockIdx.x;
  stage_tile<SQ, false>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, 256);
  __syncthreads();
  double acc = 0.0;
  for (int k = wave; k < p; k += 4) acc += term_prod(lds, colsw + (size_t)k * W2, W2, lane, a[k]);
  red[wave * kTileRows + lane] = acc;
  __syncthreads();
  if (wave == 0) {
    const uint64_t row = tile * kTileRows + lane;
    if (row < n) {
      const double s = scale[row];
      out[row] = ((red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane])) * (SQ ? s * s : s);
    }
  }
}

__global__ void __launch_bounds__(256)
k_getmat(const double *__restrict__ bm, const double *__restrict__ scale,
         const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc, const uint32_t *__restrict__ colsw,
         int W2, int p, double *__restrict__ out, uint64_t n) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = blockIdx.x;
  stage_tile<false, false>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, 256);
  __syncthreads();
  const uint64_t row = tile * kTileRows + lane;
  const double s = row < n ? scale[row] : 0.0;
  for (int k = wave; k < p; k += 4) {
    const double v = term_prod(lds, colsw + (size_t)k * W2, W2, lane, s);
    if (row < n) out[(uint64_t)k * n + row] = v;
  }
}

// grid = (row splits, p_pad / 256); wave w of a block owns terms
// blockIdx.y*256 + w*64 ... +63.
template <bool SQ>
__global__ void __launch_bounds__(256)
k_tmm(const double *__restrict__ bm, const double *__restrict__ scale,
      const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc, const uint32_t *__restrict__ colsw,
      int W2, const double *__restrict__ a, uint64_t n, uint64_t ntiles, uint64_t tiles_per_split,
      uint64_t p_pad, double *__restrict__ part) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k0 = blockIdx.y * 256 + wave * 64;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  double acc[64];
#pragma unroll
  for (int t = 0; t < 64; ++t) acc[t] = 0.0;
  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();
    stage_tile<SQ, false>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, 256);
    __syncthreads();
    const uint64_t row = tile * kTileRows + lane;
    double vs = 0.0;
    if (row < n) {
      const double s = scale[row];
      vs = a[row] * (SQ ? s * s : s);  // b = basescale % a, linalg.cpp:305
    }
#pragma unroll
    for (int t = 0; t < 64; ++t)
      acc[t] += term_prod(lds, colsw + (size_t)(k0 + t) * W2, W2, lane, vs);
  }
  // cross-lane reduction, lane t keeps the total of term k0 + t
  double mine = 0.0;
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    double v = acc[t];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == t) mine = v;
  }
  part[(uint64_t)blockIdx.x * p_pad + k0 + lane] = mine;
}

__global__ void k_tmm_reduce(const double *__restrict__ part, int nsplit, uint64_t p_pad, int p,
                             double *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p) return;
  double s = 0.0;
  for (int r = 0; r < nsplit; ++r) s += part[(uint64_t)r * p_pad + k];
  out[k] = s;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    OB_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes));
  return 0;
}

int check_mu(const obhip_terms &t) {
  if (t.Mu > (uint64_t)kMaxMuLds)
    return fail(OBHIP_ERR_INVALID, "terms touch " + std::to_string(t.Mu) +
                                       " basis columns; at most " + std::to_string(kMaxMuLds) +
                                       " fit the LDS tile");
  return 0;
}

}  // namespace

int launch_getmat(const obhip_basis &b, obhip_terms &t, double *d_out) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  OB_TRY(check_mu(t));
  ProfScope ps("getmat");
  const size_t lds = t.Mu * kTileRows * sizeof(double);
  OB_TRY(set_lds(k_getmat, lds));
  hipLaunchKernelGGL(k_getmat, dim3((unsigned)(b.n_pad / kTileRows)), dim3(256), lds, cur_stream(),
                     b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p,
                     (int)(t.W / 2), (int)t.p, d_out, b.n);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_mm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, bool squared) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  OB_TRY(check_mu(t));
  ProfScope ps(squared ? "sqmm" : "mm");
  const size_t lds = (t.Mu * kTileRows + 4 * kTileRows) * sizeof(double);
  const dim3 grid((unsigned)(b.n_pad / kTileRows));
  if (squared) {
    OB_TRY(set_lds(k_mm<true>, lds));
    hipLaunchKernelGGL(k_mm<true>, grid, dim3(256), lds, cur_stream(), b.bm.p, b.scale.p, t.ucol.p,
                       (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, (int)(t.W / 2), (int)t.p,
                       d_a, d_out, b.n);
  } else {
    OB_TRY(set_lds(k_mm<false>, lds));
    hipLaunchKernelGGL(k_mm<false>, grid, dim3(256), lds, cur_stream(), b.bm.p, b.scale.p, t.ucol.p,
                       (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, (int)(t.W / 2), (int)t.p,
                       d_a, d_out, b.n);
  }
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_tmm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, bool squared) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  OB_TRY(check_mu(t));
  ProfScope ps(squared ? "sqtmm" : "tmm");
  const uint64_t ntiles = b.n_pad / kTileRows;
  const uint64_t pblocks = (t.p + 255) / 256;
  // enough blocks to fill 256 CUs a few times over, each with >= 4 tiles
  uint64_t nsplit = std::max<uint64_t>(1, (256 * 6 + pblocks - 1) / pblocks);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  const uint64_t p_pad = t.p_pad;  // multiple of 256 (obhip_terms::prepare)
  double *part = nullptr;
  OB_TRY(const_cast<obhip_basis &>(b).workspace(nsplit * p_pad * sizeof(double), (void **)&part));
  const size_t lds = t.Mu * kTileRows * sizeof(double);
  const dim3 grid((unsigned)nsplit, (unsigned)pblocks);
  if (squared) {
    OB_TRY(set_lds(k_tmm<true>, lds));
    hipLaunchKernelGGL(k_tmm<true>, grid, dim3(256), lds, cur_stream(), b.bm.p, b.scale.p, t.ucol.p,
                       (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, (int)(t.W / 2), d_a, b.n,
                       ntiles, tps, p_pad, part);
  } else {
    OB_TRY(set_lds(k_tmm<false>, lds));
    hipLaunchKernelGGL(k_tmm<false>, grid, dim3(256), lds, cur_stream(), b.bm.p, b.scale.p, t.ucol.p,
                       (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, (int)(t.W / 2), d_a, b.n,
                       ntiles, tps, p_pad, part);
  }
  OB_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 255) / 256)), dim3(256), 0, cur_stream(),
                     part, (int)nsplit, p_pad, (int)t.p, d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
