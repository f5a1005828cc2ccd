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
// Work split and the interval tables of mat25 / mat25pow: at k_build_basis.  In the knot-loop
// path all per-dimension tables (knot constants, rotmat columns) are wave-uniform and are
// fetched with scalar loads.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

// Work split: a block of 8 waves (4 until round 5) takes 8 consecutive 64-row tiles, one per wave; all waves walk
// the dimensions together, so that a dimension's interval tables (mat25 / mat25pow,
// build_dim_tab: [m sorted u][m + 1][levels][6], 12 KB at 40 knots and 6 levels) are staged once
// per block in LDS and the per-row table reads -- seven bisection steps and 3 x levels 16-byte
// entries, a different interval per lane -- are LDS gathers, not vector-memory ones (23 gather
// instructions per row and dimension through the texture path were as slow as the
// ~17-instruction-per-knot loop they replace).  A lane keeps the running basescale of its row
// across the dimensions, so no reduction over waves is needed.  Dimensions without tables
// (mat25ang, out-of-range hyper-parameters) or with tables beyond the LDS buffer (all levels
// kept: tables grow with levels x knots) take the scalar-operand knot loop.  The kernel is
// latency-bound (a chain of bisection steps and table reads per dimension; the next dimension's
// tables and x values are fetched under it), so the register budget is capped for the waves
// per CU the two 16-KB table buffers allow (24 in three blocks of eight).
constexpr int kBbTab = kIntervalTabMax;  // doubles of LDS for one dimension's tables

__device__ __forceinline__ int bb_tab_size(const DimDesc &D) {
  return ((D.m + 1) & ~1) + (D.m + 1) * D.ncol * 6;
}

// NW: waves (= 64-row tiles) per block.  More tiles per block stage a dimension's tables fewer
// times (20 x 16 KB per block at d = 20) and put more waves on a CU.
// (4 / 8 / 16 tiles per block at the headline shape: 0.476 / 0.426 / 0.655 ms -- sixteen spill)
template <int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 5 : 6)
k_build_basis(const DimDesc *__restrict__ dims, const double *__restrict__ ka,
              const double *__restrict__ kb, const double *__restrict__ kc,
              const double *__restrict__ rot, const double *__restrict__ tab,
              const double *__restrict__ x, uint64_t n, int d, uint64_t Mc, uint64_t ntiles,
              double *__restrict__ bm, double *__restrict__ scale) {
  __shared__ __attribute__((aligned(16))) double ltab[2][kBbTab];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = (uint64_t)blockIdx.x * NW + wave;
  const bool mine = tile < ntiles;  // (wave-uniform)
  const uint64_t row = tile * kTileRows + lane;
  const StoreGlobal store{bm + tile * Mc * kTileRows + lane};
  // The tables and the x values of dimension l + 1 are fetched into registers before dimension l
  // is evaluated and go to the other LDS buffer after it: one barrier per dimension, and the
  // fetch latency hides under the evaluation.
  constexpr int kPer = kBbTab / (NW * 64);
  static_assert(kBbTab % (NW * 64) == 0, "");
  double treg[kPer];
  double xnext = 0.5;
  auto fetch = [&](int l) {
    if (l >= d) return;
    const DimDesc D = dims[l];
    // padded rows evaluate at the first knot-free point of the domain; their
    // scale is forced to zero below so they never contribute.
    xnext = mine && row < n ? x[(uint64_t)l * n + row] : 0.5;
    if (D.tab >= 0 && bb_tab_size(D) <= kBbTab) {
      const int sz = bb_tab_size(D);
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int e = threadIdx.x + NW * 64 * i;
        treg[i] = e < sz ? tab[D.tab + e] : 0.0;
      }
    }
  };
  auto put = [&](int l) {
    if (l >= d) return;
    const DimDesc D = dims[l];
    if (D.tab >= 0 && bb_tab_size(D) <= kBbTab) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) ltab[l & 1][threadIdx.x + NW * 64 * i] = treg[i];
    }
  };
  fetch(0);
  put(0);
  double sc = 1.0;
  for (int l = 0; l < d; ++l) {
    DimDesc D = dims[l];
    const bool staged = D.tab >= 0 && bb_tab_size(D) <= kBbTab;
    if (staged) D.tab = 0;
    const double xv = xnext;
    __syncthreads();  // this dimension's tables are in place, the other buffer is free
    fetch(l + 1);
    if (mine) sc *= build_dim_any(D, ka, kb, kc, rot, staged ? (const double *)ltab[l & 1] : tab, xv, store);  // modandbase.cpp:573
    put(l + 1);
  }
  if (mine) {
    bm[tile * Mc * kTileRows + lane] = 1.0;  // the all-ones column (modandbase.cpp:574)
    scale[row] = row < n ? sc : 0.0;
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
  static const int nw = getenv("OBHIP_BB_WAVES") ? atoi(getenv("OBHIP_BB_WAVES")) : 8;
#define OB_BB(NW_)                                                                                             \
  hipLaunchKernelGGL(k_build_basis<NW_>, dim3((unsigned)((tiles + NW_ - 1) / NW_)), dim3(NW_ * 64), 0,        \
                     cur_stream(), b.md.dims.p, b.md.ka.p, b.md.kb.p, b.md.kc.p, b.md.rot.p, b.md.tab.p, b.x.p, \
                     b.n, (int)b.d, b.md.Mc, tiles, b.bm.p, b.scale.p)
  if (nw == 4)
    OB_BB(4);
  else
    OB_BB(8);
#undef OB_BB
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
