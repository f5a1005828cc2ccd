// Fused per-dimension basis evaluation for gfx950.
//
//   R_l = cov_l(x[:,l], knots_l) . rotmat_l ;  c_l = R_l[:,0]
//   basemat[:, level t of dim l] = R_l[:,t] / c_l     (t >= 1)
//   basescale = prod_l c_l
//
// replaces covf::cov (src/covfuncs.cpp:113-126,197-212,285-310),
// outermod::buildob (src/modandbase.cpp:285-298) and outerbase::build
// (src/modandbase.cpp:547-626) in one pass: the n x m kernel matrix, its
// product with rotmat and the normalisation never leave registers.
//
// Data layout in HBM (DESIGN.md "basemat"): rows are grouped in tiles of 64;
// within a tile each stored ("compact") column is a contiguous run of 64
// doubles: bm[(tile * Mc + col) * 64 + row_in_tile].  Compact column 0 is the
// all-ones column (level 0 of every dimension); level t >= 1 of dimension l is
// column dims[l].ccol0 + t - 1.  One lane owns one row, so every store is a
// fully coalesced 512-byte wave store and no LDS transposition is needed.
//
// Work split: block = 4 waves = one 64-row tile; wave w evaluates dimensions
// w, w+4, ...  All per-dimension tables (knot constants, rotmat columns) are
// wave-uniform and are fetched with scalar loads.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

__global__ void __launch_bounds__(256)
k_build_basis(const DimDesc *__restrict__ dims, const double *__restrict__ ka,
              const double *__restrict__ kb, const double *__restrict__ kc,
              const double *__restrict__ rot, const double *__restrict__ x, uint64_t n, int d,
              uint64_t Mc, double *__restrict__ bm, double *__restrict__ scale) {
  __shared__ double part[4][kTileRows];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = blockIdx.x;
  const uint64_t row = tile * kTileRows + lane;
  const bool valid = row < n;
  double *tile_out = bm + tile * Mc * kTileRows + lane;
  double sc = 1.0;
  for (int l = wave; l < d; l += 4) {
    const DimDesc D = dims[l];
    // padded rows evaluate at the first knot-free point of the domain; their
    // scale is forced to zero below so they never contribute.
    const double xv = valid ? x[(uint64_t)l * n + row] : 0.5;
    const double cl = build_dim_any(D, ka, kb, kc, rot, xv, StoreGlobal{tile_out});
    sc *= cl;  // modandbase.cpp:573
  }
  part[wave][lane] = sc;
  if (wave == 0) tile_out[0] = 1.0;  // the all-ones column (modandbase.cpp:574)
  __syncthreads();
  if (wave == 0) {
    const double s = part[0][lane] * part[1][lane] * part[2][lane] * part[3][lane];
    scale[row] = valid ? s : 0.0;
  }
}

// raw R_k = cov(x[:,k], knots_k) . rotmat_k, all m_k columns: ob$getbase(k)
// (modandbase.cpp:634-639; basemat block times basescalemat column).
template <int KIND>
__device__ __forceinline__ void getbase_dim(const DimDesc &D, const double *ka, const double *kb,
                                            const double *kc, const double *rot, double xv,
                                            uint64_t n, uint64_t row, double *__restrict__ out) {
  double a0, a1, a2;
  kernel_pre<KIND>(D, xv, a0, a1, a2);
  double acc[8];
  for (int c0 = 0; c0 < D.ncolp; c0 += 8) {
    dim_chunk<KIND>(D, ka, kb, kc, rot, a0, a1, a2, c0, acc);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c0 + c < D.ncol) out[(uint64_t)(c0 + c) * n + row] = acc[c];
  }
}

__global__ void __launch_bounds__(256)
k_getbase(DimDesc D, const double *__restrict__ ka, const double *__restrict__ kb,
          const double *__restrict__ kc, const double *__restrict__ rot,
          const double *__restrict__ xcol, uint64_t n, double *__restrict__ out) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const double xv = xcol[row];
  if (D.kind == OBHIP_COV_MAT25)
    getbase_dim<OBHIP_COV_MAT25>(D, ka, kb, kc, rot, xv, n, row, out);
  else if (D.kind == OBHIP_COV_MAT25POW)
    getbase_dim<OBHIP_COV_MAT25POW>(D, ka, kb, kc, rot, xv, n, row, out);
  else if (D.kind == kCovMat25Direct)
    getbase_dim<kCovMat25Direct>(D, ka, kb, kc, rot, xv, n, row, out);
  else if (D.kind == kCovMat25PowDirect)
    getbase_dim<kCovMat25PowDirect>(D, ka, kb, kc, rot, xv, n, row, out);
  else
    getbase_dim<OBHIP_COV_MAT25ANG>(D, ka, kb, kc, rot, xv, n, row, out);
}

}  // namespace

int launch_build_basis(obhip_basis &b) {
  ProfScope ps("build_basis");
  const uint64_t tiles = b.n_pad / kTileRows;
  hipLaunchKernelGGL(k_build_basis, dim3((unsigned)tiles), dim3(256), 0, cur_stream(),
                     b.md.dims.p, b.md.ka.p, b.md.kb.p, b.md.kc.p, b.md.rot.p, b.x.p, b.n,
                     (int)b.d, b.md.Mc, b.bm.p, b.scale.p);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_getbase(const obhip_basis &b, uint64_t k, double *d_out) {
  // full-width rotation table of dimension k only
  const obhip_model &m = *b.model;
  std::vector<int64_t> cap(m.d, 0);
  cap[k] = (int64_t)m.m_of(k) - 1;
  ModelDev md;
  OB_TRY(md.build(m, cap));
  ProfScope ps("getbase");
  hipLaunchKernelGGL(k_getbase, dim3((unsigned)((b.n + 255) / 256)), dim3(256), 0, cur_stream(),
                     md.dims_h[k], md.ka.p, md.kb.p, md.kc.p, md.rot.p, b.x.p + k * b.n, b.n, d_out);
  OB_HIP(hipGetLastError());
  OB_HIP(hipStreamSynchronize(cur_stream()));  // md is freed on return
  return 0;
}

}  // namespace obhip
