// Hyper-parameter gradients of the outer-product basis for gfx950:
//   basemat_gradhyp        outermod::buildob, gradient form   src/modandbase.cpp:306-327
//                          outerbase::build with dograd       src/modandbase.cpp:547-626
//   getmat_gradhyp         getmge_                            src/linalg.cpp:778-822
//   matmul_gradhyp         prodmmge_ / domultgesub_           src/linalg.cpp:139-276
//   tmatmul_gradhyp        tprodmmge_ / dotmultgesub_         src/linalg.cpp:362-471
//
// For hyper-parameter h of dimension l = hypmatch[h] the reference stores, per row,
//   basemat_gradhyp[:, gest[h] + t] = (d cov/d hyp_h . rotmat_l + cov . rotmat_gradhyp_h)[:, t] / c_l
// for EVERY level t (0 included), and its products amount to
//   d B[i,k] / d hyp_h = basescale_i . prod_{m != l} basemat[i, col(m, t_km)] . basemat_gradhyp[i, gest[h] + t_kl]
// i.e. the ordinary term product with dimension l's factor (the constant 1 at level 0)
// replaced by the gradient column.  So this file builds ONE combined tile-blocked array
// (basemat columns, then per hyper-parameter the gradient columns of levels 0..cap) and
// per hyper-parameter a *view* of the terms in which dimension l is dropped and a pseudo-
// dimension pointing at the gradient block carries level t_kl + 1; the value kernels
// k_mm / k_tmm / getmat (kernels_prod.hip) then compute gradient products unchanged.
//
// Who computes what:
//   getmat_gradhyp            getmat on the view, one pass per hyper-parameter
//   matmul / sqmm _gradhyp    with M = sum_k a_k P_k (one k_mm pass over all terms) and
//                             delta[h, t] = ge[h, t] - basemat[col(dim h, t)] ge[h, 0]:
//                             out[:, h] = s (ge[h, 0] M + sum_{k: t_kl > 0} a_k E_k delta[h, t_kl]),
//                             so per hyper-parameter only the terms that HAVE its dimension are
//                             multiplied out (k_mm on a restricted view whose pseudo-dimension
//                             points at the delta columns): nnz(terms) products instead of
//                             p x nhyp
//   tmatmul / sqtmm _gradhyp  k_tmm_ge0: the terms without the hyper-parameter's dimension, all
//                             hyper-parameters at once (products on the fly, MFMA contraction
//                             with the weight columns), + k_tmm_tl on the concatenated views
//                             restricted to the terms that have it.  k_bt_times_u (a streaming
//                             pass over the materialised design matrix) is the older form of
//                             the dense part, kept for terms of more than 8 factors
//   fallback: k_tmm on the full views, one pass per hyper-parameter
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <atomic>
#include <set>
#include <thread>

#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

// kernel value and its hyper-parameter derivatives at one knot (covfuncs.cpp:134-150,
// 220-243, 318-347); a0, a1 as in kernel_pre, lx = log(x) (mat25pow only)
template <int KIND>
__device__ __forceinline__ void kernel_value_grad(const DimDesc &D, double ka, double kb, double kd,
                                                  double a0, double a1, double lx, double &kv,
                                                  double &g0, double &g1) {
  constexpr double a = 2.0, b = 0.25;
  if (KIND == OBHIP_COV_MAT25ANG) {
    const double hs = a0 - ka, hc = a1 - kb;
    const double h = sqrt(hs * hs + hc * hc);
    const double e = exp(-h), w = e * (h + 1.0);
    kv = (1.0 + h + h * h * (1.0 / 3.0)) * e;
    g0 = a / 3 * hs * hs * w;
    g1 = a / 3 * hc * hc * w;
  } else {
    const double h = a0 - ka, ah = fabs(h);
    const double e = exp(-ah);
    kv = (1.0 + ah + ah * ah * (1.0 / 3.0)) * e;
    const double h2 = h * (1.0 + ah) * e;
    g0 = a / 3 * (h * h2);
    g1 = 0.0;
    if (KIND == OBHIP_COV_MAT25POW || KIND == kCovMat25PowDirect)
      // D.p0 = powv; a0 and ka are centred on D.p2 (device_common.h), t(x) itself is a0 + D.p2
      g1 = (lx * (a0 + D.p2) - kd) * (-(b * D.p0 / 3) * h2) + b / 3 * (h * h2);
  }
}

// one dimension, one row: R = cov . rot, Rt_h = dcov_h . rot + cov . rotg_h, everything
// divided by R[0]; basemat columns and gradient columns go to the combined tile
template <int KIND>
__device__ __forceinline__ double build_dim_grad(const DimDesc &D, const GradHyp *__restrict__ hy,
                                                 int nh, const double *__restrict__ ka,
                                                 const double *__restrict__ kb,
                                                 const double *__restrict__ kd,
                                                 const double *__restrict__ rot,
                                                 const double *__restrict__ rotg, double xv,
                                                 double *__restrict__ tile_out) {
  double a0, a1, a2;
  kernel_pre<KIND>(D, xv, a0, a1, a2);
  const double lx = (KIND == OBHIP_COV_MAT25POW || KIND == kCovMat25PowDirect) ? log(xv) : 0.0;
  double cl = 1.0, g00 = 0.0, g10 = 0.0;  // level-0 gradient columns of the two hyper-parameters
  for (int c0 = 0; c0 < D.ncolp; c0 += 8) {
    double r[8], t0[8], t1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = t0[c] = t1[c] = 0.0;
    const double *rp = rot + D.rotoff + c0;
    const double *g0p = rotg + hy[0].rotgoff + c0;
    const double *g1p = rotg + hy[nh - 1].rotgoff + c0;
    for (int j = 0; j < D.m; ++j) {
      double kv, d0, d1;
      kernel_value_grad<KIND>(D, ka[D.koff + j], kb[D.koff + j], kd[D.koff + j], a0, a1, lx, kv, d0,
                              d1);
      const size_t o = (size_t)j * D.ncolp;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const double rc = rp[o + c];
        r[c] = fma(kv, rc, r[c]);
        t0[c] = fma(d0, rc, fma(kv, g0p[o + c], t0[c]));
        if (KIND != OBHIP_COV_MAT25 && KIND != kCovMat25Direct) t1[c] = fma(d1, rc, fma(kv, g1p[o + c], t1[c]));
      }
    }
    if (c0 == 0) {
      cl = r[0];
      g00 = t0[0] / cl;
      g10 = t1[0] / cl;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int col = c0 + c;
      if (col < D.ncol) {
        const double bv = r[c] / cl, ge0 = t0[c] / cl, ge1 = t1[c] / cl;
        tile_out[(size_t)(hy[0].gecol + col) * kTileRows] = ge0;
        if (KIND != OBHIP_COV_MAT25 && KIND != kCovMat25Direct) tile_out[(size_t)(hy[1].gecol + col) * kTileRows] = ge1;
        if (col >= 1) {
          tile_out[(size_t)(D.ccol0 + col - 1) * kTileRows] = bv;
          tile_out[(size_t)(hy[0].dcol + col - 1) * kTileRows] = fma(-bv, g00, ge0);
          if (KIND != OBHIP_COV_MAT25 && KIND != kCovMat25Direct)
            tile_out[(size_t)(hy[1].dcol + col - 1) * kTileRows] = fma(-bv, g10, ge1);
        }
      }
    }
  }
  return cl;
}

// ---- interval tables of the gradient basis (mat25 / mat25pow) -------------------------------
// The same algebra as the value path's tables (device_common.h build_dim_tab, core.cpp): between
// two neighbouring knots every per-knot quantity of build_dim_grad is e^{-h} (below) or e^{-ah}
// (above) times a polynomial of degree <= 3 in h = u(x) - u_j:
//   kv = (1 + |h| + h^2 / 3) e            cov                       covfuncs.cpp:121-124
//   d0 = (a / 3) h^2 (1 + |h|) e          d cov / d hyp_0           covfuncs.cpp:134-150
//   d1 = -(b pw / 3) (L - kd_j) h (1 + |h|) e + (b / 3) h^2 (1 + |h|) e   (mat25pow, :220-243;
//        L = log(x) t(x) is a per-row scalar, kd_j = log(knot_j) t(knot_j))
// so with t = u(x) - (the largest u_j <= u(x)) the three knot sums per level are
//   r  = e^{-t} V-(t) + e^{t} V+(t)
//   t0 = e^{-t} C(t)  + e^{t} D(t)
//   t1 = e^{-t} (E(t) + L P(t)) + e^{t} (F(t) + L Q(t))
// with 6 + 8 (+ 8 + 6) host-built coefficients per (interval, level) (build_grad_tab below) -- O(1)
// per (row, level) where the knot loop spends ~45 instructions per (row, knot, 8 levels).
// A dimension's table is [mu sorted u][chunk 0][chunk 1] ...: the levels in chunks of `nck`, a chunk
// = [m + 1 intervals][nck levels][nc] -- as many levels as fit the LDS buffer beside the sorted u
// (every mat25 / mat25pow dimension of at most 127 knots gets tables: 128 intervals x 28
// coefficients are 3584 doubles per level).
struct GradTab {
  int off;      // offset (doubles) of the dimension's table in the table array, -1: knot loop
  int nc;       // coefficients per (interval, level): 14 (mat25) or 28 (mat25pow)
  int mu;       // sorted u, padded to an even length
  int nck;      // levels per chunk
  int nchunks;  // ceil(ncol / nck)
  int chunk;    // doubles per chunk = (m + 1) * nck * nc
};
constexpr int kGradTabMax = 16384;  // doubles of LDS for the sorted u + one chunk (128 KB)

// a row's place in a dimension's table: interval, local variable, the two exponentials, L
struct GradRow {
  int J;
  double t, em, ep, L;
};
template <int KIND>
__device__ __forceinline__ GradRow grad_row(const DimDesc &D, const double *__restrict__ tab, double xv) {
  constexpr bool POW = KIND == OBHIP_COV_MAT25POW;
  const double ux = (POW ? pow(xv, D.p0) / D.p1 : xv / D.p0) - D.p2;
  int lo = 0, hi = D.m;  // u_(lo-1) <= ux < u_(hi)
  for (int it = 0; it < 7; ++it) {  // m <= 127
    const int mid = (lo + hi) >> 1;
    const bool open = lo < hi;
    const bool le = tab[min(mid, D.m - 1)] <= ux;
    lo = open && le ? mid + 1 : lo;
    hi = open && !le ? mid : hi;
  }
  GradRow g;
  g.J = lo;
  g.t = ux - tab[max(g.J - 1, 0)];
  g.em = g.J == 0 ? 0.0 : exp(-g.t);
  g.ep = g.J == D.m ? 0.0 : exp(g.t);
  g.L = POW ? log(xv) * (ux + D.p2) : 0.0;
  return g;
}
// level 0's values, carried from the first chunk of a dimension to the later ones
struct GradLevel0 {
  double cl = 1.0, icl = 1.0, g00 = 0.0, g10 = 0.0;
};
// levels [c_lo, c_hi) of one dimension for one row; chunk: the staged chunk [m + 1][nck][NC]
template <int KIND>
__device__ __forceinline__ void build_dim_grad_tab(const DimDesc &D, const GradHyp *__restrict__ hy,
                                                   const double *__restrict__ chunk, int nck, int c_lo,
                                                   int c_hi, const GradRow &g, GradLevel0 &z,
                                                   double *__restrict__ tile_out) {
  constexpr bool POW = KIND == OBHIP_COV_MAT25POW;
  constexpr int NC = POW ? 28 : 14;
  typedef double dd2 __attribute__((ext_vector_type(2)));
  const double t = g.t, em = g.em, ep = g.ep, L = g.L;
  const dd2 *__restrict__ cf = (const dd2 *)(chunk + (size_t)g.J * nck * NC);
  double icl = z.icl, cl = z.cl, g00 = z.g00, g10 = z.g10;
#pragma unroll 1
  for (int c = c_lo; c < c_hi; ++c) {
    const dd2 *e = cf + (c - c_lo) * (NC / 2);
    const dd2 v0 = e[0], v1 = e[1], v2 = e[2];           // V-0 V-1 | V-2 V+0 | V+1 V+2
    const dd2 c0 = e[3], c1 = e[4], d0 = e[5], d1 = e[6];  // C0 C1 | C2 C3 | D0 D1 | D2 D3
    const double r = em * fma(t, fma(t, v1.x, v0.y), v0.x) + ep * fma(t, fma(t, v2.y, v2.x), v1.y);
    const double s0 = em * fma(t, fma(t, fma(t, c1.y, c1.x), c0.y), c0.x) +
                      ep * fma(t, fma(t, fma(t, d1.y, d1.x), d0.y), d0.x);
    double s1 = 0.0;
    if (POW) {
      const dd2 e0 = e[7], e1 = e[8], f0 = e[9], f1 = e[10];  // E0 E1 | E2 E3 | F0 F1 | F2 F3
      const dd2 p0 = e[11], pq = e[12], q1 = e[13];            // P0 P1 | P2 Q0 | Q1 Q2
      const double lo_ = fma(t, fma(t, fma(t, e1.y, e1.x), e0.y), e0.x) + L * fma(t, fma(t, pq.x, p0.y), p0.x);
      const double hi_ = fma(t, fma(t, fma(t, f1.y, f1.x), f0.y), f0.x) + L * fma(t, fma(t, q1.y, q1.x), pq.y);
      s1 = em * lo_ + ep * hi_;
    }
    if (c == 0) {
      cl = r;
      icl = 1.0 / r;
      g00 = s0 * icl;
      g10 = s1 * icl;
    }
    const double bv = r * icl, ge0 = s0 * icl, ge1 = s1 * icl;
    tile_out[(size_t)(hy[0].gecol + c) * kTileRows] = ge0;
    if (POW) tile_out[(size_t)(hy[1].gecol + c) * kTileRows] = ge1;
    if (c >= 1) {
      tile_out[(size_t)(D.ccol0 + c - 1) * kTileRows] = bv;
      tile_out[(size_t)(hy[0].dcol + c - 1) * kTileRows] = fma(-bv, g00, ge0);
      if (POW) tile_out[(size_t)(hy[1].dcol + c - 1) * kTileRows] = fma(-bv, g10, ge1);
    }
  }
  z.cl = cl, z.icl = icl, z.g00 = g00, z.g10 = g10;
}

// 16 waves = 16 row tiles per block; all waves walk the dimensions that have a table together,
// a dimension's sorted u and one chunk of its levels staged in LDS at a time.  Dimensions without a
// table (mat25ang, out-of-range hyper-parameters, more than 127 knots) are left to
// k_build_basis_grad, which then ran BEFORE this kernel on those dimensions alone
// (scale_has_part: scale holds their product).
__global__ void __launch_bounds__(1024)
k_build_basis_grad_tab(const DimDesc *__restrict__ dims, const GradHyp *__restrict__ hyps,
                       const int *__restrict__ hypst, const GradTab *__restrict__ gtabs,
                       const double *__restrict__ gtab, const double *__restrict__ x, uint64_t n, int d,
                       uint64_t Mtot, uint64_t ntiles, int scale_has_part, double *__restrict__ bm,
                       double *__restrict__ scale) {
  extern __shared__ double ltab[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = (uint64_t)blockIdx.x * 16 + wave;
  const bool mine = tile < ntiles;  // (wave-uniform)
  const uint64_t row = tile * kTileRows + lane;
  const bool valid = mine && row < n;
  double *tile_out = bm + tile * Mtot * kTileRows + lane;
  double sc = 1.0;
  for (int l = 0; l < d; ++l) {
    const GradTab T = gtabs[l];
    if (T.off < 0) continue;  // (block-uniform)
    const DimDesc D = dims[l];
    const GradHyp *hy = hyps + hypst[l];
    const bool m25 = D.kind == OBHIP_COV_MAT25;
    GradRow g;
    GradLevel0 z;
    // per chunk: ~1800 clocks of LDS-bound evaluation per wave and 8 levels follow, so the ~1 us the
    // copy is exposed for is ~10 % (prefetching it through registers made the kernel spill)
    for (int k = 0; k < T.nchunks; ++k) {
      __syncthreads();  // everyone is done with what the buffer held
      if (k == 0)
        for (int e = threadIdx.x; e < T.mu; e += 1024) ltab[e] = gtab[T.off + e];
      const double *src = gtab + T.off + T.mu + (size_t)k * T.chunk;
      for (int e = threadIdx.x; e < T.chunk; e += 1024) ltab[T.mu + e] = src[e];
      __syncthreads();
      if (!mine) continue;
      if (k == 0) {
        const double xv = valid ? x[(uint64_t)l * n + row] : 0.5;
        g = m25 ? grad_row<OBHIP_COV_MAT25>(D, ltab, xv) : grad_row<OBHIP_COV_MAT25POW>(D, ltab, xv);
      }
      const int c_lo = k * T.nck, c_hi = min(D.ncol, c_lo + T.nck);
      if (m25)
        build_dim_grad_tab<OBHIP_COV_MAT25>(D, hy, ltab + T.mu, T.nck, c_lo, c_hi, g, z, tile_out);
      else
        build_dim_grad_tab<OBHIP_COV_MAT25POW>(D, hy, ltab + T.mu, T.nck, c_lo, c_hi, g, z, tile_out);
    }
    sc *= z.cl;
  }
  if (mine) {
    tile_out[0] = 1.0;
    if (scale_has_part) sc *= scale[row];
    scale[row] = valid ? sc : 0.0;
  }
}

__global__ void __launch_bounds__(256)
k_build_basis_grad(const DimDesc *__restrict__ dims, const GradHyp *__restrict__ hyps,
                   const int *__restrict__ hypst, const double *__restrict__ ka,
                   const double *__restrict__ kb, const double *__restrict__ kd,
                   const double *__restrict__ rot, const double *__restrict__ rotg,
                   const double *__restrict__ x, uint64_t n, int d, uint64_t Mtot,
                   const int *__restrict__ dimsel /* d dimensions to take, or null: 0 .. d - 1 */,
                   double *__restrict__ bm, double *__restrict__ scale) {
  __shared__ double part[4][kTileRows];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = blockIdx.x;
  const uint64_t row = tile * kTileRows + lane;
  const bool valid = row < n;
  double *tile_out = bm + tile * Mtot * kTileRows + lane;
  double sc = 1.0;
  for (int li = wave; li < d; li += 4) {
    const int l = dimsel ? dimsel[li] : li;
    const DimDesc D = dims[l];
    const GradHyp *hy = hyps + hypst[l];
    const int nh = hypst[l + 1] - hypst[l];
    const double xv = valid ? x[(uint64_t)l * n + row] : 0.5;
    double cl;
    // (the gradient kernel takes one exp per knot anyway: the DIRECT kinds are the plain ones)
    if (D.kind == OBHIP_COV_MAT25 || D.kind == kCovMat25Direct)
      cl = build_dim_grad<kCovMat25Direct>(D, hy, nh, ka, kb, kd, rot, rotg, xv, tile_out);
    else if (D.kind == OBHIP_COV_MAT25POW || D.kind == kCovMat25PowDirect)
      cl = build_dim_grad<kCovMat25PowDirect>(D, hy, nh, ka, kb, kd, rot, rotg, xv, tile_out);
    else
      cl = build_dim_grad<OBHIP_COV_MAT25ANG>(D, hy, nh, ka, kb, kd, rot, rotg, xv, tile_out);
    sc *= cl;
  }
  part[wave][lane] = sc;
  if (wave == 0) tile_out[0] = 1.0;
  __syncthreads();
  if (wave == 0) {
    const double s = part[0][lane] * part[1][lane] * part[2][lane] * part[3][lane];
    scale[row] = valid ? s : 0.0;
  }
}

// ---- host: the interval tables of one dimension -------------------------------------------------
// Layout: [m sorted u, padded to an even length][m + 1 intervals][ncol levels][NC] with, per
// (interval J, level c), the polynomial coefficients (lowest power first) of
//   V- (3) V+ (3) | C (4) D (4) | E (4) F (4) P (3) Q (3)     (the last 14 for mat25pow only)
// in the local variable t = u(x) - ref_J, ref_J = u_(J-1) (u_(0) for J = 0): "-" sums the knots
// below (j < J, h = t + q_j, q_j = ref_J - u_j), "+" the knots above (ah = q_j - t, q_j = u_j -
// ref_J), each knot weighted by e^{-q_j} <= 1.  Built by two recurrences in extended precision,
// O(m) per level where a direct sum per interval would be O(m^2): going from interval J to J + 1
// the local variable moves by D = u_J - u_(J-1), so the "below" polynomial is shifted
// (P(t + D), a Taylor shift of a cubic) and damped by e^{-D} before knot J enters with q = 0;
// the "above" polynomials run the same way from the last interval down.  Every factor is <= 1
// and nothing is ever subtracted that the knot sum itself does not subtract.
typedef long double ld;
struct Cubic {
  ld c[4] = {0, 0, 0, 0};
  void shift(ld D) {  // P(t) -> P(t + D)
    const ld p0 = c[0], p1 = c[1], p2 = c[2], p3 = c[3];
    c[0] = p0 + D * (p1 + D * (p2 + D * p3));
    c[1] = p1 + D * (2 * p2 + 3 * D * p3);
    c[2] = p2 + 3 * D * p3;
  }
  void scale(ld f) {
    for (ld &v : c) v *= f;
  }
  void add(const Cubic &o, ld f) {
    for (int k = 0; k < 4; ++k) c[k] += f * o.c[k];
  }
  Cubic flipped() const {  // P(t) -> P(-t)
    Cubic r;
    r.c[0] = c[0], r.c[1] = -c[1], r.c[2] = c[2], r.c[3] = -c[3];
    return r;
  }
};

// us: the m centred knot positions u_j sorted ascending, ord their original indices; R, G0, G1:
// rotmat / rotmat_gradhyp columns of level c (indexed by the ORIGINAL knot index); kdv:
// log(knot_j) t(knot_j).  out: the chunks behind the sorted u, level c at chunk c / nck, entry
// ((J * nck) + c % nck) * NC of it.
void build_grad_tab(int m, int ncol, int nck, bool pw, double powv, const std::vector<double> &us,
                    const std::vector<int> &ord, const double *rot, const double *rotg0,
                    const double *rotg1, uint64_t ldr, const double *kdv, double *out) {
  const int NC = pw ? 28 : 14;
  const ld a3 = 2.0L / 3.0L, b3 = 0.25L / 3.0L, bp3 = 0.25L * (ld)powv / 3.0L;
  Cubic gkv, gcub, ghq;  // (1 + z + z^2 / 3), z^2 (1 + z), z (1 + z)
  gkv.c[0] = 1, gkv.c[1] = 1, gkv.c[2] = 1.0L / 3.0L;
  gcub.c[2] = 1, gcub.c[3] = 1;
  ghq.c[1] = 1, ghq.c[2] = 1;
  for (int c = 0; c < ncol; ++c) {
    const double *R = rot + (size_t)c * ldr, *G0 = rotg0 + (size_t)c * ldr, *G1 = pw ? rotg1 + (size_t)c * ldr : nullptr;
    // per-knot polynomials in z (z = h below, z = ah above); the sign of the z (1 + z) term flips
    // above, where h (1 + ah) = -ah (1 + ah)
    auto knot = [&](int j, bool above, Cubic (&f)[4]) {
      const int o = ord[j];
      const ld r = R[o], g0 = G0[o];
      for (Cubic &q : f) q = Cubic();
      f[0].add(gkv, r);                       // V
      f[1].add(gcub, a3 * r), f[1].add(gkv, g0);  // T0
      if (pw) {
        const ld sg = above ? -1.0L : 1.0L;
        f[2].add(ghq, sg * bp3 * (ld)kdv[o] * r), f[2].add(gcub, b3 * r), f[2].add(gkv, (ld)G1[o]);  // E / F
        f[3].add(ghq, -sg * bp3 * r);                                                                // P / Q
      }
    };
    std::vector<Cubic> below((size_t)(m + 1) * 4), above((size_t)(m + 1) * 4);
    Cubic cur[4], kn[4];
    // forward: below(J), J = 1 .. m (below(0) is empty)
    for (int J = 1; J <= m; ++J) {
      if (J >= 2) {
        const ld D = (ld)us[J - 1] - (ld)us[J - 2];
        const ld e = expl(-D);
        for (Cubic &q : cur) q.shift(D), q.scale(e);
      }
      knot(J - 1, false, kn);  // enters with q = 0: z = t
      for (int f = 0; f < 4; ++f) cur[f].add(kn[f], 1), below[(size_t)J * 4 + f] = cur[f];
    }
    // backward: above(J) for J = m - 1 .. 0 (above(m) is empty), as polynomials in t
    for (Cubic &q : cur) q = Cubic();
    for (int J = m - 1; J >= 0; --J) {
      // local variable of interval J against that of J + 1: t_J = t_(J+1) + D, D = ref_(J+1) - ref_J
      const ld refJ = us[J >= 1 ? J - 1 : 0], refJ1 = us[J];
      const ld D = refJ1 - refJ;
      const ld e = expl(-D);
      for (Cubic &q : cur) q.shift(-D), q.scale(e);
      // knot J enters with q_J = u_J - ref_J: polynomial in z = q_J - t -> in t
      knot(J, true, kn);
      const ld qJ = (ld)us[J] - refJ;
      const ld w = expl(-qJ);
      for (int f = 0; f < 4; ++f) {
        Cubic g = kn[f];
        g.shift(qJ);              // g(z = qJ + s)
        cur[f].add(g.flipped(), w);  // s = -t
        above[(size_t)J * 4 + f] = cur[f];
      }
    }
    for (int J = 0; J <= m; ++J) {
      double *e = out + (size_t)(c / nck) * ((size_t)(m + 1) * nck * NC) + ((size_t)J * nck + c % nck) * NC;
      const Cubic *lo = &below[(size_t)J * 4], *hi = &above[(size_t)J * 4];
      e[0] = (double)lo[0].c[0], e[1] = (double)lo[0].c[1], e[2] = (double)lo[0].c[2];
      e[3] = (double)hi[0].c[0], e[4] = (double)hi[0].c[1], e[5] = (double)hi[0].c[2];
      for (int k = 0; k < 4; ++k) e[6 + k] = (double)lo[1].c[k], e[10 + k] = (double)hi[1].c[k];
      if (pw) {
        for (int k = 0; k < 4; ++k) e[14 + k] = (double)lo[2].c[k], e[18 + k] = (double)hi[2].c[k];
        for (int k = 0; k < 3; ++k) e[22 + k] = (double)lo[3].c[k], e[25 + k] = (double)hi[3].c[k];
      }
    }
  }
}

}  // namespace

// (Re)build the gradient basis of b for the current model state.
int ensure_gradbasis(obhip_basis &b) {
  const obhip_model &m = *b.model;
  // like the reference's outerbase, b does not follow later changes of the model
  // (vignettes/learning.Rmd:48-54); mixing its tables with newer gradient tables would be
  // silently wrong, so ask for the rebuild instead
  if (b.md.model_version != m.version)
    return fail(OBHIP_ERR_STATE, "the model changed after this basis was built: call build() first");
  if (b.grad && b.grad->model_version == m.version) return 0;
  HostTimer ht_all("ensure_gradbasis (rebuild)");
  auto g = std::make_unique<obhip_gradbasis>();
  const uint64_t d = m.d, nh = m.nhyp();
  // tables: per hyper-parameter the rotmat_gradhyp block capped like ModelDev::rot
  std::vector<double> hrotg, hkd(m.M(), 0.0);
  std::vector<int> hhypst(d + 1);
  g->hyps_h.resize(nh);
  uint64_t gecol = b.md.Mc;
  for (uint64_t l = 0; l < d; ++l) {
    const DimDesc &D = b.md.dims_h[l];
    const uint64_t ml = m.m_of(l), o = m.knotptst[l];
    hhypst[l] = (int)m.hypst[l];
    for (uint64_t h = m.hypst[l]; h < m.hypst[l + 1]; ++h) {
      GradHyp &G = g->hyps_h[h];
      G.dim = (int)l;
      G.which = (int)(h - m.hypst[l]);
      G.rotgoff = (int)hrotg.size();
      G.gecol = (int)gecol;
      gecol += (uint64_t)D.ncol;
      hrotg.resize(hrotg.size() + ml * D.ncolp, 0.0);
      for (uint64_t j = 0; j < ml; ++j)
        for (int cc = 0; cc < D.ncol; ++cc)
          hrotg[G.rotgoff + j * D.ncolp + cc] = m.rotmat_gradhyp[(m.gest[h] + cc) * m.mmax + j];
    }
    if (m.kinds[l] == OBHIP_COV_MAT25POW)
      for (uint64_t j = 0; j < ml; ++j) {
        const double t = std::pow(m.knotpt[o + j], D.p0) / D.p1;
        hkd[o + j] = std::log(m.knotpt[o + j]) * t;  // covfuncs.cpp:234
      }
  }
  hhypst[d] = (int)m.hypst[d];
  for (uint64_t h = 0; h < nh; ++h) {  // the delta blocks behind all gradient blocks
    g->hyps_h[h].dcol = (int)gecol;
    gecol += (uint64_t)b.md.dims_h[m.hypmatch[h]].ncol - 1;
  }
  OB_TRY(g->hyps.upload(g->hyps_h.data(), nh));
  {
    std::vector<int> c0(nh);
    for (uint64_t h = 0; h < nh; ++h) c0[h] = g->hyps_h[h].gecol;
    if (nh) OB_TRY(g->ge0col.upload(c0.data(), nh));
  }
  OB_TRY(g->rotg.upload(hrotg.data(), hrotg.size()));
  OB_TRY(g->kd.upload(hkd.data(), hkd.size()));
  DevBuf<int> dhypst;
  OB_TRY(dhypst.upload(hhypst.data(), hhypst.size()));
  // interval tables of the mat25 / mat25pow dimensions (OBHIP_GRAD_KNOTLOOP=1: none, the
  // round-3 kernel -- A/B runs)
  const bool knotloop = getenv("OBHIP_GRAD_KNOTLOOP") && atoi(getenv("OBHIP_GRAD_KNOTLOOP")) != 0;
  HostTimer ht_tab("ensure_gradbasis: host tables + uploads");
  std::vector<GradTab> hgt(d);
  std::vector<int> rest;  // dimensions without a table: the knot loop
  std::vector<double> htab;
  std::vector<uint64_t> tabbed;  // dimensions with a table
  uint64_t htab_size = 0;
  bool any_tab = false;
  for (uint64_t l = 0; l < d; ++l) {
    const DimDesc &D = b.md.dims_h[l];
    const uint64_t ml = m.m_of(l);
    const bool pw = m.kinds[l] == OBHIP_COV_MAT25POW;
    const int nc = pw ? 28 : 14;
    const uint64_t mu = (ml + 1) / 2 * 2;
    const uint64_t per_level = (ml + 1) * (uint64_t)nc;
    hgt[l] = GradTab{-1, nc, (int)mu, 0, 0, 0};
    rest.push_back((int)l);
    // (D.kind differs from the model's kind when the knots spread too far for the separable
    // exponentials: those dimensions keep the per-knot exp)
    if (knotloop || (m.kinds[l] != OBHIP_COV_MAT25 && !pw) || D.kind != m.kinds[l] || ml > 127) continue;
    const uint64_t nck = std::min<uint64_t>((uint64_t)D.ncol, ((uint64_t)kGradTabMax - mu) / per_level);
    const uint64_t nchunks = ((uint64_t)D.ncol + nck - 1) / nck;
    const uint64_t size = mu + nchunks * nck * per_level;
    rest.pop_back();
    if (htab_size % 2) ++htab_size;
    hgt[l].off = (int)htab_size;
    hgt[l].nck = (int)nck;
    hgt[l].nchunks = (int)nchunks;
    hgt[l].chunk = (int)(nck * per_level);
    htab_size += size;
    tabbed.push_back(l);
    any_tab = true;
  }
  // the tables of the dimensions are independent (two extended-precision recurrences over the knots
  // per level each): filled on host threads, like the eigen-problems of obhip_model::build -- an
  // obfit function evaluation rebuilds them
  htab.assign(htab_size, 0.0);
  auto fill_dim = [&](uint64_t l) {
    const DimDesc &D = b.md.dims_h[l];
    const uint64_t ml = m.m_of(l), o = m.knotptst[l];
    const bool pw = m.kinds[l] == OBHIP_COV_MAT25POW;
    const uint64_t mu = (ml + 1) / 2 * 2;
    std::vector<int> ord(ml);
    std::vector<double> u(ml), us(ml);
    for (uint64_t j = 0; j < ml; ++j) {
      ord[j] = (int)j;
      u[j] = (pw ? std::pow(m.knotpt[o + j], D.p0) / D.p1 : m.knotpt[o + j] / D.p0) - D.p2;
    }
    std::stable_sort(ord.begin(), ord.end(), [&](int a2, int b2) { return u[a2] < u[b2]; });
    for (uint64_t j = 0; j < ml; ++j) us[j] = u[ord[j]];
    double *T = &htab[hgt[l].off];
    for (uint64_t j = 0; j < ml; ++j) T[j] = us[j];
    for (uint64_t j = ml; j < mu; ++j) T[j] = us[ml - 1];
    const uint64_t h0 = m.hypst[l];
    build_grad_tab((int)ml, D.ncol, hgt[l].nck, pw, D.p0, us, ord, &m.rotmat[o * m.mmax],
                   &m.rotmat_gradhyp[m.gest[h0] * m.mmax], pw ? &m.rotmat_gradhyp[m.gest[h0 + 1] * m.mmax] : nullptr,
                   m.mmax, &hkd[o], T + mu);
  };
  {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t nthr = std::min<uint64_t>({(uint64_t)tabbed.size(), (uint64_t)hw, 16});
    if (nthr <= 1) {
      for (uint64_t l : tabbed) fill_dim(l);
    } else {
      std::atomic<uint64_t> next{0};
      std::vector<std::thread> pool;
      for (uint64_t q = 0; q < nthr; ++q)
        pool.emplace_back([&] {
          for (uint64_t e = next++; e < tabbed.size(); e = next++) fill_dim(tabbed[e]);
        });
      for (std::thread &th : pool) th.join();
    }
  }
  DevBuf<GradTab> dgt;
  DevBuf<double> dtab;
  DevBuf<int> drest;
  if (any_tab) {
    OB_TRY(dgt.upload(hgt.data(), hgt.size()));
    OB_TRY(dtab.upload(htab.data(), htab.size()));
    if (!rest.empty()) OB_TRY(drest.upload(rest.data(), rest.size()));
  }

  // the combined array as an obhip_basis whose dimension table is extended by two
  // pseudo-dimensions per hyper-parameter: d + h, whose level j >= 1 is gradient level j - 1,
  // and d + nhyp + h, whose level j >= 1 is delta level j
  g->gb = std::make_unique<obhip_basis>();
  obhip_basis &gb = *g->gb;
  gb.model = b.model;
  gb.n = b.n;
  gb.n_pad = b.n_pad;
  gb.d = d + 2 * nh;
  gb.device = b.device;
  gb.md.cap = b.md.cap;
  gb.md.dims_h = b.md.dims_h;
  for (uint64_t h = 0; h < nh; ++h) {
    const DimDesc &D = b.md.dims_h[m.hypmatch[h]];
    DimDesc P = D;
    P.ccol0 = g->hyps_h[h].gecol;  // level j -> column gecol + j - 1
    P.ncol = D.ncol + 1;
    gb.md.cap.push_back((int64_t)D.ncol);
    gb.md.dims_h.push_back(P);
  }
  for (uint64_t h = 0; h < nh; ++h) {
    const DimDesc &D = b.md.dims_h[m.hypmatch[h]];
    DimDesc P = D;
    P.ccol0 = g->hyps_h[h].dcol;  // level j -> column dcol + j - 1
    gb.md.cap.push_back((int64_t)D.ncol - 1);
    gb.md.dims_h.push_back(P);
  }
  gb.md.Mc = gecol;
  gb.md.model_version = m.version;
  const uint64_t tiles = b.n_pad / kTileRows;
  OB_TRY(gb.bm.alloc(tiles * gecol * kTileRows));
  OB_TRY(gb.scale.alloc(b.n_pad));
  {
    ProfScope ps("build_basis_grad");
    if (any_tab) {
      // the dimensions without a table first (their product goes to scale), then the others
      if (!rest.empty())
        hipLaunchKernelGGL(k_build_basis_grad, dim3((unsigned)tiles), dim3(256), 0, cur_stream(),
                           b.md.dims.p, g->hyps.p, dhypst.p, b.md.ka.p, b.md.kb.p, g->kd.p, b.md.rot.p,
                           g->rotg.p, b.x.p, b.n, (int)rest.size(), gecol, drest.p, gb.bm.p, gb.scale.p);
      const size_t lds = (size_t)kGradTabMax * sizeof(double);
      OB_TRY(ensure_dyn_lds((const void *)k_build_basis_grad_tab, lds));
      hipLaunchKernelGGL(k_build_basis_grad_tab, dim3((unsigned)((tiles + 15) / 16)), dim3(1024), lds, cur_stream(),
                         b.md.dims.p, g->hyps.p, dhypst.p, dgt.p, dtab.p, b.x.p, b.n, (int)d, gecol, tiles,
                         rest.empty() ? 0 : 1, gb.bm.p, gb.scale.p);
    } else {
      hipLaunchKernelGGL(k_build_basis_grad, dim3((unsigned)tiles), dim3(256), 0, cur_stream(),
                         b.md.dims.p, g->hyps.p, dhypst.p, b.md.ka.p, b.md.kb.p, g->kd.p, b.md.rot.p,
                         g->rotg.p, b.x.p, b.n, (int)d, gecol, (const int *)nullptr, gb.bm.p, gb.scale.p);
    }
    OB_HIP(hipGetLastError());
    OB_HIP(hipStreamSynchronize(cur_stream()));  // dhypst and the tables are locals
  }
  g->model_version = m.version;
  static uint64_t next_id = 1;
  g->id = next_id++;
  b.grad = std::move(g);
  return 0;
}

namespace {

// squared store of the gradient basis (basematsq / basematsq_gradhyp, modandbase.cpp:581,
// 588-590): basemat columns squared, gradient column t of a hyper-parameter -> 2 * it *
// basemat level t of its dimension (the ones column at level 0); scale squared
__global__ void __launch_bounds__(256)
k_square_gradbasis(const double *__restrict__ src, const double *__restrict__ scale,
                   const uint32_t *__restrict__ pair, uint64_t Mc, uint64_t Mtot,
                   double *__restrict__ dst, double *__restrict__ scale_sq) {
  const uint64_t tile = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double *s = src + tile * Mtot * kTileRows + lane;
  double *o = dst + tile * Mtot * kTileRows + lane;
  for (uint64_t c = wave; c < Mtot; c += 4) {
    const double v = s[c * kTileRows];
    o[c * kTileRows] = c < Mc ? v * v : 2.0 * v * s[(uint64_t)pair[c - Mc] * kTileRows];
  }
  if (wave == 0) {
    const double sc = scale[tile * kTileRows + lane];
    scale_sq[tile * kTileRows + lane] = sc * sc;
  }
}

}  // namespace

int ensure_gradbasis_sq(obhip_basis &b) {
  OB_TRY(ensure_gradbasis(b));
  obhip_gradbasis &g = *b.grad;
  if (g.gbsq) return 0;
  const obhip_basis &gb = *g.gb;
  const uint64_t Mc = b.md.Mc, Mtot = gb.md.Mc;
  std::vector<uint32_t> pair(Mtot - Mc, 0u);
  for (size_t h = 0; h < g.hyps_h.size(); ++h) {
    const DimDesc &D = b.md.dims_h[g.hyps_h[h].dim];
    for (int t = 1; t < D.ncol; ++t) {
      pair[g.hyps_h[h].gecol - Mc + t] = (uint32_t)(D.ccol0 + t - 1);
      // delta_sq[t] = ge_sq[t] - basemat[t]^2 ge_sq[0] = 2 basemat[t] delta[t]
      pair[g.hyps_h[h].dcol - Mc + t - 1] = (uint32_t)(D.ccol0 + t - 1);
    }
  }
  DevBuf<uint32_t> dpair;
  OB_TRY(dpair.upload(pair.data(), pair.size()));
  auto sq = std::make_unique<obhip_basis>();
  sq->model = gb.model;
  sq->n = gb.n;
  sq->n_pad = gb.n_pad;
  sq->d = gb.d;
  sq->device = gb.device;
  sq->md.cap = gb.md.cap;
  sq->md.dims_h = gb.md.dims_h;
  sq->md.Mc = Mtot;
  sq->md.model_version = gb.md.model_version;
  const uint64_t tiles = gb.n_pad / kTileRows;
  OB_TRY(sq->bm.alloc(tiles * Mtot * kTileRows));
  OB_TRY(sq->scale.alloc(gb.n_pad));
  hipLaunchKernelGGL(k_square_gradbasis, dim3((unsigned)tiles), dim3(256), 0, cur_stream(), gb.bm.p,
                     gb.scale.p, dpair.p, Mc, Mtot, sq->bm.p, sq->scale.p);
  OB_HIP(hipGetLastError());
  OB_HIP(hipStreamSynchronize(cur_stream()));  // dpair is a local
  g.gbsq = std::move(sq);
  return 0;
}

namespace {

// out_gradhyp[h][i] += ge[h, 0]_i * M_i  (M = B a or B^2 a, already scaled by basescale)
__global__ void k_ge0_combine(const double *__restrict__ bm, uint64_t Mtot,
                              const int *__restrict__ ge0col, const double *__restrict__ M,
                              uint64_t n, double *__restrict__ outge) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const int h = blockIdx.y;
  const double g0 = bm[((row >> 6) * Mtot + (uint64_t)ge0col[h]) * kTileRows + (row & 63)];
  outge[(uint64_t)h * n + row] = fma(g0, M[row], outge[(uint64_t)h * n + row]);
}

}  // namespace

namespace {

// ---- transposed products: the dense part of every hyper-parameter in one streaming pass ---------
// out_gradhyp[k, h] = sum_i w_i dB_ik/dhyp_h.  For the terms WITHOUT hyper-parameter h's
// dimension that is sum_i (w_i ge[h, 0]_i) B_ik: a tall-skinny product of the materialised
// row-major design matrix (obhip_basis::bmat, as the Gram kernel uses it) with an n x nhyp
// weight matrix -- one pass over B for all hyper-parameters, HBM-bound.  The terms that do
// have the dimension are recomputed by k_tmm on a view restricted to them (grad_view_sparse).
// Thread = 4 consecutive terms x NH hyper-parameters; the row weights sit in lane registers
// (lane = row of the 64-row tile) and are broadcast with v_readlane.  SQ: squared stores
// (B^2; the level-0 gradient column of the squared store is 2 ge[h, 0]).
constexpr int kNHB = 20;  // hyper-parameters per pass

template <bool SQ>
__global__ void __launch_bounds__(512)
k_bt_times_u(const double *__restrict__ Bmat, uint64_t p_pad, const double *__restrict__ gtile,
             uint64_t Mtot, const int *__restrict__ ge0abs, int nhyp, int h0,
             const double *__restrict__ a, uint64_t n, uint64_t ntiles, uint64_t tiles_per_split,
             double *__restrict__ part) {
  const int tid = threadIdx.x, lane = tid & 63;
  const uint64_t k4 = (uint64_t)blockIdx.y * 2048 + (uint64_t)tid * 4;
  if (k4 >= p_pad) return;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  double acc[4][kNHB];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int h = 0; h < kNHB; ++h) acc[q][h] = 0.0;
  int gcol[kNHB];
#pragma unroll
  for (int h = 0; h < kNHB; ++h) gcol[h] = h0 + h < nhyp ? ge0abs[h0 + h] : -1;
  for (uint64_t tile = t0; tile < t1; ++tile) {
    const uint64_t row = tile * kTileRows + lane;
    const double av = row < n ? a[row] : 0.0;
    double uw[kNHB];
#pragma unroll
    for (int h = 0; h < kNHB; ++h)
      uw[h] = gcol[h] >= 0 ? av * gtile[(tile * Mtot + (uint64_t)gcol[h]) * kTileRows + lane] : 0.0;
    const double *brow = Bmat + tile * kTileRows * p_pad + k4;
#pragma unroll 4
    for (int r = 0; r < kTileRows; ++r) {
      const double4 v4 = *(const double4 *)(brow + (uint64_t)r * p_pad);
      double v[4] = {v4.x, v4.y, v4.z, v4.w};
      if (SQ) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= v[q];
      }
#pragma unroll
      for (int h = 0; h < kNHB; ++h) {
        const double wh = readlane_f64(uw[h], r);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][h] = fma(v[q], wh, acc[q][h]);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < kNHB; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      part[((uint64_t)blockIdx.x * kNHB + h) * p_pad + k4 + q] = acc[q][h];
}

// ---- the same dense part without the design matrix: products on the fly, MFMA contraction ------
// D[h][k] = sum_i (a_i s_i ge[h, 0]_i) P_k(i) is a (hyper-parameters x rows) x (rows x terms)
// product whose right factor never exists: with lane = (term t16 = l & 15, row r4 = l >> 4) a
// lane's term product at rows 4 s + r4 IS the B operand element B[k = r4][j = t16] of
// v_mfma_f64_16x16x4_f64, and the weight column of hyper-parameter t16 read at the same rows
// is the A operand element A[i = t16][k = r4].  Column addresses are loop-invariant per lane
// as in the term-per-lane kernels (the step is the immediate offset), so a (term, row) costs
// its WE column reads, WE - 1 multiplies and 1/64 of an MFMA per 16 hyper-parameters.  The
// weights (staged behind the tile's columns, premultiplied by a and the basescale) go through
// the same in-order LDS queue one step ahead; every s_waitcnt counts the reads behind its
// target exactly (ge_newer_*).  A wave holds 4 groups of 16 terms, a block 512 terms.
typedef double ge_d4 __attribute__((ext_vector_type(4)));
constexpr int kGeNU = 4, kGeSteps = 16, kGeInflight = 8;

// SQF: the unit's product is squared before it enters the MFMA (the squared stores' products are
// the squares of the plain ones: the double-buffered kernel stages the raw columns by LDS-direct loads)
// N4: further groups of FOUR hyper-parameters on v_mfma_f64_4x4x4_4b_f64 (a quarter of the matrix-pipe
// time of a 16-block: 20 hyper-parameters cost 1.25 blocks instead of 2).  There the product is the A
// operand -- lane (block (l >> 2) & 3, i = l & 3, k = l >> 4) holds A_blk[i][k], which IS term
// t16 = l & 15 at row r4 = l >> 4 -- and the weight of hyper-parameter l & 3 at the same row the B
// operand B_blk[k][j = l & 3], the same in all four blocks; D_blk[i][j] lands in lane (i = l >> 4,
// blk, j = l & 3): term 4 blk + (l >> 4), hyper-parameter l & 3.
template <int WE, int W, int NHB, bool SQF = false, int N4 = 0>
struct GePipe {
  static constexpr int NU = kGeNU, TOT = kGeSteps * kGeNU, NWT = NHB + N4;
  static constexpr int D = (kGeInflight / WE) > 0 ? kGeInflight / WE : 1;
  // units issued once step U has issued (the prologue issues units 0 .. D-2)
  static constexpr int issued(int U) { return U + D < TOT ? U + D : TOT; }
  static constexpr bool wnext(int s) { return s + 1 < kGeSteps; }  // weights of step s+1 exist
  // LDS reads queued behind the weights of step s when step s waits for them
  static constexpr int newer_w(int s) {
    const int units = s == 0 ? issued(0) : issued(s * NU) - issued((s - 1) * NU);
    const int v = units * WE + (wnext(s) ? NWT : 0);
    return v < 15 ? v : 15;
  }
  // LDS reads queued behind the reads of unit U when step U waits for them
  static constexpr int newer_u(int U) {
    const int units = (TOT - 1 - U) < (D - 1) ? (TOT - 1 - U) : (D - 1);
    int wr = 0;
    for (int S = (U - D + 1 > 0 ? U - D + 1 : 0); S <= U; ++S)
      if (S % NU == 0 && wnext(S / NU)) wr += NWT;
    const int v = units * WE + wr;
    return v < 15 ? v : 15;
  }

  template <int U, typename C>
  static __device__ __forceinline__ void issue(C &c, double (&buf)[12]) {
    constexpr int st = U / NU, unit = U % NU, s = (U % D) * WE;
#pragma unroll
    for (int j = 0; j < WE; ++j) buf[s + j] = tl_rd<st * 32>(c.ad[unit][W - WE + j]);
  }
  template <int S, typename C>
  static __device__ __forceinline__ void issue_w(C &c, double (&w)[12]) {
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb) w[(S % 2) * NWT + hb] = tl_rd<S * 32>(c.aw[hb]);
    if constexpr (N4 > 0) {
#pragma unroll
      for (int q = 0; q < N4; ++q) w[(S % 2) * NWT + NHB + q] = tl_rd<S * 32>(c.aw4[q]);
    }
  }
  template <int U, typename C>
  static __device__ __forceinline__ void steps(C &c, double (&buf)[12], double (&w)[12]) {
    if constexpr (U < TOT) {
      constexpr int st = U / NU, unit = U % NU, s = (U % D) * WE;
      if constexpr (U + D - 1 < TOT) issue<U + D - 1>(c, buf);
      if constexpr (unit == 0) {
        if constexpr (wnext(st)) issue_w<st + 1>(c, w);
        tl_waitn<newer_w(st), NWT, (st % 2) * NWT>(w);
      }
      tl_waitn<newer_u(U), WE, s>(buf);
      double v = buf[s];
#pragma unroll
      for (int j = 1; j < WE; ++j) v *= buf[s + j];
      if constexpr (SQF) v = v * v;
#pragma unroll
      for (int hb = 0; hb < NHB; ++hb)
        c.acc[unit][hb] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[(st % 2) * NWT + hb], v,
                                                               c.acc[unit][hb], 0, 0, 0);
      if constexpr (N4 > 0) {
#pragma unroll
        for (int q = 0; q < N4; ++q)
          c.acc4[unit][q] =
              __builtin_amdgcn_mfma_f64_4x4x4f64(v, w[(st % 2) * NWT + NHB + q], c.acc4[unit][q], 0, 0, 0);
      }
      steps<U + 1>(c, buf, w);
    }
  }
  template <int U, typename C>
  static __device__ __forceinline__ void prologue(C &c, double (&buf)[12]) {
    if constexpr (U < D - 1 && U < TOT) {
      issue<U>(c, buf);
      prologue<U + 1>(c, buf);
    }
  }
  template <typename C>
  static __device__ __forceinline__ void run(C &c) {
    double buf[12], w[12];
    issue_w<0>(c, w);
    prologue<0>(c, buf);
    steps<0>(c, buf, w);
  }
};

template <int W, int NHB, int N4 = 0>
struct GeCtx {
  uint32_t ad[kGeNU][W];
  uint32_t aw[NHB > 0 ? NHB : 1];
  ge_d4 acc[kGeNU][NHB > 0 ? NHB : 1];
  uint32_t aw4[N4 > 0 ? N4 : 1];
  double acc4[kGeNU][N4 > 0 ? N4 : 1];
};

template <int W, int NHB, bool SQF = false, int N4 = 0>
__device__ __forceinline__ void ge_tile(GeCtx<W, NHB, N4> &c, int we) {
  if (we == W) {
    GePipe<W, W, NHB, SQF, N4>::run(c);
  } else if (we == W - 1) {
    GePipe<W - 1, W, NHB, SQF, N4>::run(c);
  } else if (W >= 3 && we == W - 2) {
    GePipe<(W >= 3 ? W - 2 : 1), W, NHB, SQF, N4>::run(c);
  } else {
    GePipe<(W >= 4 ? W - 3 : 1), W, NHB, SQF, N4>::run(c);
  }
}

// NW: waves per block (16 where the tile leaves room for only one block per CU)
template <int W2, int NHB, bool SQ, int NW>
__global__ void __launch_bounds__(NW * 64, 4)
k_tmm_ge0(const double *__restrict__ bm, const double *__restrict__ scale,
          const uint32_t *__restrict__ ucol, int Mu, uint64_t Mtot,
          const uint32_t *__restrict__ colsw, const uint32_t *__restrict__ sperm,
          const int *__restrict__ ge0abs, int nhyp, int h0, const double *__restrict__ a, uint64_t n,
          uint64_t ntiles, uint64_t tiles_per_split, uint64_t p_pad, double *__restrict__ part) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2, HS = 16 * NHB;
  const int lane = threadIdx.x & 63, t16 = lane & 15, r4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  // my terms: slots (blockIdx.y * 8 + wave) * 64 + g * 16 + t16 of the sorted order
  GeCtx<W, NHB> c;
  int nzmax = 1;
#pragma unroll
  for (int g = 0; g < kGeNU; ++g) {
    const uint64_t slot = ((uint64_t)blockIdx.y * NW + wave) * 64 + g * 16 + t16;
    const bool ok = slot < p_pad;
    const uint64_t k = ok ? sperm[slot] : 0;
    uint32_t cw[W2];
#pragma unroll
    for (int w = 0; w < W2; ++w) {
      cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
      c.ad[g][2 * w] = (cw[w] & 0xffffu) * (kTlPitch * 8) + r4 * 8;
      c.ad[g][2 * w + 1] = (cw[w] >> 16) * (kTlPitch * 8) + r4 * 8;
    }
    nzmax = max(nzmax, tl_nnz<W2>(cw));
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb) c.acc[g][hb] = ge_d4{0.0, 0.0, 0.0, 0.0};
  }
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb) c.aw[hb] = (uint32_t)(Mu + hb * 16 + t16) * (kTlPitch * 8) + r4 * 8;
  const int we = tl_variant<W>(wave_max_i32(nzmax));
  const bool live = ((uint64_t)blockIdx.y * NW + wave) * 64 < p_pad;

  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();  // every wave is done with the previous tile
    {
      const double *src = bm + tile * Mtot * kTileRows + lane;
      // SQ: the squared stores formed while staging (basematsq = basemat^2, its level-0 gradient
      // column 2 ge[h, 0], scale^2: k_square_gradbasis) -- no second copy of the gradient basis
      for (int u = wave; u < Mu; u += NW) {
        const double v = src[(size_t)ucol[u] * kTileRows];
        lds[u * kTlPitch + lane] = SQ ? v * v : v;
      }
      const uint64_t row = tile * kTileRows + lane;
      double wr = 0.0;
      if (row < n) {
        const double sc = scale[row];
        wr = a[row] * (SQ ? sc * sc : sc);
      }
      for (int h = wave; h < HS; h += NW) {
        const int col = h0 + h < nhyp ? ge0abs[h0 + h] : -1;
        double gv = col >= 0 ? src[(size_t)col * kTileRows] : 0.0;
        if (SQ) gv = 2.0 * gv;
        lds[(Mu + h) * kTlPitch + lane] = col >= 0 ? wr * gv : 0.0;
      }
    }
    __syncthreads();
    if (live) ge_tile<W, NHB>(c, we);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: column (term) = lane & 15, row (hyper-parameter) =
  // (lane >> 4) + 4 * reg
#pragma unroll
  for (int g = 0; g < kGeNU; ++g) {
    const uint64_t slot = ((uint64_t)blockIdx.y * NW + wave) * 64 + g * 16 + t16;
    if (slot >= p_pad) continue;
    const uint64_t k = sperm[slot];
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        part[((uint64_t)blockIdx.x * HS + hb * 16 + r4 + 4 * v) * p_pad + k] = c.acc[g][hb][v];
  }
}

// The same with the NEXT tile on its way while this one is worked on: two tile buffers, one block of
// 16 waves per CU, the value columns fetched by LDS-direct loads (global_load_lds_dword as in k_hm2:
// no staging registers -- the kernel sits at 104-128 VGPRs -- and no ds_write), the 16 weight
// columns (one per wave) through two registers.  The columns arrive raw, so the squared stores'
// products are formed as squares of the plain products (GePipe<SQF>).  k_tmm_ge0 above loads its
// tile between two barriers: with two blocks per CU one computes while the other loads, but at a
// few microseconds of compute per tile the matrix pipe still idled half of the time.
// N4: hyper-parameters h0 + 16 .. h0 + 16 + 4 N4 ride along in groups of four (GePipe); NHB = 0:
// groups of four only (the last 1 to 8 hyper-parameters in a pass of their own, whose matrix
// instructions take a quarter or half of a 16-block's time).
template <int W2, bool SQ, int N4, int NHB = 1>
__global__ void __launch_bounds__(1024, 4)
k_tmm_ge0_db(const double *__restrict__ bm, const double *__restrict__ scale,
             const uint32_t *__restrict__ ucol, int Mu, uint64_t Mtot,
             const uint32_t *__restrict__ colsw, const uint32_t *__restrict__ sperm,
             const int *__restrict__ ge0abs, int nhyp, int h0, const double *__restrict__ a, uint64_t n,
             uint64_t ntiles, uint64_t tiles_per_split, uint64_t p_pad, double *__restrict__ part) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2, H16 = 16 * NHB, HS = H16 + 4 * N4, NW = 16;
  static_assert(NHB <= 1 && HS <= 24 && HS >= 4, "");
  const int lane = threadIdx.x & 63, t16 = lane & 15, r4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
  const uint32_t tile_bytes = (uint32_t)(Mu + HS) * (kTlPitch * 8);

  GeCtx<W, NHB, N4> c;
  int nzmax = 1;
#pragma unroll
  for (int g = 0; g < kGeNU; ++g) {
    const uint64_t slot = ((uint64_t)blockIdx.y * NW + wave) * 64 + g * 16 + t16;
    const bool ok = slot < p_pad;
    const uint64_t k = ok ? sperm[slot] : 0;
    uint32_t cw[W2];
#pragma unroll
    for (int w = 0; w < W2; ++w) {
      cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
      c.ad[g][2 * w] = lds0 + (cw[w] & 0xffffu) * (kTlPitch * 8) + r4 * 8;
      c.ad[g][2 * w + 1] = lds0 + (cw[w] >> 16) * (kTlPitch * 8) + r4 * 8;
    }
    nzmax = max(nzmax, tl_nnz<W2>(cw));
    c.acc[g][0] = ge_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < (N4 > 0 ? N4 : 1); ++q) c.acc4[g][q] = 0.0;
  }
  c.aw[0] = lds0 + (uint32_t)(Mu + t16) * (kTlPitch * 8) + r4 * 8;
#pragma unroll
  for (int q = 0; q < (N4 > 0 ? N4 : 1); ++q)
    c.aw4[q] = lds0 + (uint32_t)(Mu + H16 + 4 * q + (lane & 3)) * (kTlPitch * 8) + r4 * 8;
  const int we = tl_variant<W>(wave_max_i32(nzmax));
  const bool live = ((uint64_t)blockIdx.y * NW + wave) * 64 < p_pad;
  // wave h stages weight column h: a s (SQ: a s^2) times ge[h, 0] (SQ: 2 ge[h, 0]) of the row = lane
  // (the first waves a second one, column 16 + h, where groups of four ride along with a 16-block)
  constexpr bool kSecond = NHB > 0 && N4 > 0;
  const int wcol = (wave < HS && h0 + wave < nhyp) ? ge0abs[h0 + wave] : -1;
  const int wcol2 = (kSecond && wave < 4 * N4 && h0 + 16 + wave < nhyp) ? ge0abs[h0 + 16 + wave] : -1;

  // next tile -> the other buffer.  (The weight column's two loads are requested BEFORE the
  // LDS-direct loads and used at the top of the next tile: the compiler's vmcnt bookkeeping does
  // not see the inline-asm loads, a use right here would wait for all of them.)
  double gvn = 0.0, wrn = 0.0, gvn2 = 0.0;
  auto prefetch = [&](uint64_t tile, int bsel) {
    const char *tb = (const char *)(bm + tile * Mtot * kTileRows);
    const uint64_t row = tile * kTileRows + lane;
    gvn = wrn = gvn2 = 0.0;
    if (row < n && wcol >= 0) {
      const double sc = scale[row];
      wrn = a[row] * (SQ ? sc * sc : sc);
      gvn = ((const double *)tb)[(size_t)wcol * kTileRows + lane];
      if (kSecond && wcol2 >= 0) gvn2 = ((const double *)tb)[(size_t)wcol2 * kTileRows + lane];
    }
    const uint32_t l0 = lds0 + (bsel ? tile_bytes : 0u);
    for (int u = wave; u < Mu; u += NW) {
      const uint32_t col = __builtin_amdgcn_readfirstlane(ucol[u]);
      const uint64_t ga = (uint64_t)(tb + (size_t)col * (kTileRows * 8));
      const uint64_t gu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
      const uint32_t ld = (uint32_t)__builtin_amdgcn_readfirstlane((int)(l0 + (uint32_t)u * (kTlPitch * 8)));
// m0 is written here: on the clobber list so that the compiler never assumes a value of its own
// survives the statement (round-4 advice; m0 is a reserved register, hence the diagnostic)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\t"
                   "global_load_lds_dword %0, %1 offset:256"
                   :: "v"((uint32_t)lane * 4u), "s"((const char *)gu), "s"(ld) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  if (t0 < t1) prefetch(t0, 0);

  for (uint64_t tile = t0; tile < t1; ++tile) {
    const int bsel = (int)((tile - t0) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the tile has landed
    if (wave < HS)
      lds[(bsel ? tile_bytes / 8 : 0) + (Mu + wave) * kTlPitch + lane] = wrn * (SQ ? 2.0 * gvn : gvn);
    if (kSecond && wave < 4 * N4)
      lds[(bsel ? tile_bytes / 8 : 0) + (Mu + 16 + wave) * kTlPitch + lane] = wrn * (SQ ? 2.0 * gvn2 : gvn2);
    __syncthreads();  // tile and weights complete; every wave is done with the other buffer
    if (tile + 1 < t1) prefetch(tile + 1, bsel ^ 1);
    if (live) ge_tile<W, NHB, SQ, N4>(c, we);
    // on to the other buffer
    const uint32_t delta = bsel ? 0u - tile_bytes : tile_bytes;
#pragma unroll
    for (int g = 0; g < kGeNU; ++g)
#pragma unroll
      for (int j = 0; j < W; ++j) {
        c.ad[g][j] += delta;
        asm volatile("" : "+v"(c.ad[g][j]));
      }
    c.aw[0] += delta;
    asm volatile("" : "+v"(c.aw[0]));
    if constexpr (N4 > 0) {
#pragma unroll
      for (int q = 0; q < N4; ++q) {
        c.aw4[q] += delta;
        asm volatile("" : "+v"(c.aw4[q]));
      }
    }
  }
#pragma unroll
  for (int g = 0; g < kGeNU; ++g) {
    const uint64_t slot = ((uint64_t)blockIdx.y * NW + wave) * 64 + g * 16 + t16;
    if (NHB > 0 && slot < p_pad) {
      const uint64_t k = sperm[slot];
#pragma unroll
      for (int v = 0; v < 4; ++v) part[((uint64_t)blockIdx.x * HS + r4 + 4 * v) * p_pad + k] = c.acc[g][0][v];
    }
    if constexpr (N4 > 0) {
      // D layout of v_mfma_f64_4x4x4_4b_f64: term 4 blk + (lane >> 4), hyper-parameter lane & 3
      const uint64_t slot4 = ((uint64_t)blockIdx.y * NW + wave) * 64 + g * 16 + ((lane >> 2) & 3) * 4 + r4;
      if (slot4 < p_pad) {
        const uint64_t k4 = sperm[slot4];
#pragma unroll
        for (int q = 0; q < N4; ++q)
          part[((uint64_t)blockIdx.x * HS + H16 + 4 * q + (lane & 3)) * p_pad + k4] = c.acc4[g][q];
      }
    }
  }
}

// D[h0 + h][k] = sum of the row-split partials laid out [split][hs][p_pad]
__global__ void k_ge0_reduce(const double *__restrict__ part, int nsplit, int hs, uint64_t p_pad,
                             int p, int nh, double *__restrict__ out /* [nh][p] */) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y;
  if (k >= (uint64_t)p || h >= nh) return;
  double s = 0.0;
  for (int r = 0; r < nsplit; ++r) s += part[((uint64_t)r * hs + h) * p_pad + k];
  out[(uint64_t)h * p + k] = s;
}

// D[h0 + h][k] = sum of the row-split partials
__global__ void k_btu_reduce(const double *__restrict__ part, int nsplit, uint64_t p_pad, int p,
                             int nh, double *__restrict__ out /* [nh][p] */) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y;
  if (k >= (uint64_t)p || h >= nh) return;
  double s = 0.0;
  for (int r = 0; r < nsplit; ++r) s += part[((uint64_t)r * kNHB + h) * p_pad + k];
  out[(uint64_t)h * p + k] = s;
}

}  // namespace

namespace {
template <int W2, int NHB, bool SQ, int NW>
int run_tmm_ge0_sq(const obhip_basis &src, obhip_terms &t, const int *d_c0, int nhyp, int h0,
                   const double *d_a, dim3 grid, uint64_t ntiles, uint64_t tps, double *part) {
  const size_t lds = (t.Mu + 16 * NHB) * kTlPitch * sizeof(double);
  if (lds > 64 * 1024)
    OB_TRY(ensure_dyn_lds((const void *)k_tmm_ge0<W2, NHB, SQ, NW>, lds));
  hipLaunchKernelGGL((k_tmm_ge0<W2, NHB, SQ, NW>), grid, dim3(NW * 64), lds, cur_stream(), src.bm.p,
                     src.scale.p, t.ucol.p, (int)t.Mu, src.md.Mc, (const uint32_t *)t.cols.p,
                     t.sperm.p, d_c0, nhyp, h0, d_a, src.n, ntiles, tps, t.p_pad, part);
  OB_HIP(hipGetLastError());
  return 0;
}
template <int W2, bool SQ, int N4, int NHB = 1>
int run_tmm_ge0_db(const obhip_basis &src, obhip_terms &t, const int *d_c0, int nhyp, int h0,
                   const double *d_a, dim3 grid, uint64_t ntiles, uint64_t tps, double *part) {
  const size_t lds = 2 * (t.Mu + 16 * NHB + 4 * N4) * kTlPitch * sizeof(double);
  OB_TRY(ensure_dyn_lds((const void *)k_tmm_ge0_db<W2, SQ, N4, NHB>, lds));
  hipLaunchKernelGGL((k_tmm_ge0_db<W2, SQ, N4, NHB>), grid, dim3(1024), lds, cur_stream(), src.bm.p, src.scale.p,
                     t.ucol.p, (int)t.Mu, src.md.Mc, (const uint32_t *)t.cols.p, t.sperm.p, d_c0, nhyp, h0,
                     d_a, src.n, ntiles, tps, t.p_pad, part);
  OB_HIP(hipGetLastError());
  return 0;
}
template <int W2, int NHB>
int run_tmm_ge0(bool sq, int nw, int n4, const obhip_basis &src, obhip_terms &t, const int *d_c0, int nhyp,
                int h0, const double *d_a, dim3 grid, uint64_t ntiles, uint64_t tps, double *part) {
  if (nw == 32) {  // (16 waves, two tile buffers)
    if (n4 == -1) {  // groups of four only
      if (sq) return run_tmm_ge0_db<W2, true, 1, 0>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
      return run_tmm_ge0_db<W2, false, 1, 0>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
    }
    if (n4 == -2) {
      if (sq) return run_tmm_ge0_db<W2, true, 2, 0>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
      return run_tmm_ge0_db<W2, false, 2, 0>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
    }
    if (n4 == 1) {
      if (sq) return run_tmm_ge0_db<W2, true, 1>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
      return run_tmm_ge0_db<W2, false, 1>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
    }
    if (n4 == 2) {
      if (sq) return run_tmm_ge0_db<W2, true, 2>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
      return run_tmm_ge0_db<W2, false, 2>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
    }
    if (sq) return run_tmm_ge0_db<W2, true, 0>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
    return run_tmm_ge0_db<W2, false, 0>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
  }
  if (nw == 16) {
    if (sq) return run_tmm_ge0_sq<W2, NHB, true, 16>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
    return run_tmm_ge0_sq<W2, NHB, false, 16>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
  }
  if (sq) return run_tmm_ge0_sq<W2, NHB, true, 8>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
  return run_tmm_ge0_sq<W2, NHB, false, 8>(src, t, d_c0, nhyp, h0, d_a, grid, ntiles, tps, part);
}
}  // namespace

// the dense part without the design matrix (k_tmm_ge0); t prepared for b
bool tmm_ge0_supports(const obhip_terms &t) {
  const uint64_t w2 = t.W / 2;
  return w2 >= 1 && w2 <= 4 && (t.Mu + 32) * kTlPitch * sizeof(double) <= 156 * 1024;
}

// d_out: p x nhyp column-major (device) = sum_i a_i ge[h, 0]_i B_ik (squared: the squared stores,
// formed from the gradient basis while staging -- gbsq is not needed)
int launch_tmm_ge0(obhip_basis &b, obhip_terms &t, bool squared, const double *d_a, double *d_out) {
  obhip_gradbasis &g = *b.grad;
  const obhip_basis &src = *g.gb;
  const int nhyp = (int)b.model->nhyp();
  const DevBuf<int> &dc0 = g.ge0col;  // absolute columns in the combined array (resident: no upload,
                                      // and no synchronisation at the end of this function)
  const uint64_t ntiles = b.n_pad / kTileRows;
  // a tile of more than half the LDS leaves one block per CU: 16 waves in it instead of 8
  // (OBHIP_GE0_WAVES=8|16 forces either: A/B runs)
  // ... and where TWO tiles fit the LDS, 16 waves with the next tile prefetched into the second
  // buffer (k_tmm_ge0_db; nw = 32 stands for it below)
  static const int force_nw = getenv("OBHIP_GE0_WAVES") ? atoi(getenv("OBHIP_GE0_WAVES")) : 0;
  // Hyper-parameters beyond a multiple of 16 (d = 20 mat25: 4 of 20) in groups of four on the
  // 4 x 4 x 4 matrix instruction instead of a mostly empty 16-block.  OBHIP_GE0_FOURS: 0 = never,
  // 1 = 1 to 8 left over take a pass of their own (NHB = 0), 2 (default) = behind a 16-block they
  // ride along with it (the products formed once; 22-39 registers of the W2 = 2 kernel spilled
  // outside the read pipeline).  d = 20 mat25 at the headline terms, 20 hyper-parameters: dense part
  // 5.47 / 4.52 / 3.50 ms, obfit evaluation 30.8 / 29.8 / 28.5 ms (tools/r05_fours_ab.sh).
  static const int fours = getenv("OBHIP_GE0_FOURS") ? atoi(getenv("OBHIP_GE0_FOURS")) : 2;
  const size_t tile_lds = (t.Mu + 16) * kTlPitch * sizeof(double);
  const bool two_tiles = 2 * tile_lds <= 156 * 1024;
  bool one_per_cu = tile_lds > 80 * 1024;
  int nw = one_per_cu ? 16 : 8;
  if (two_tiles) nw = 32;
  if (force_nw == 8 || force_nw == 16 || (force_nw == 32 && two_tiles)) nw = force_nw;
  if (nw == 32) one_per_cu = true;
  const uint64_t tpb = (uint64_t)(nw == 32 ? 16 : nw) * 64;
  const uint64_t pblocks = (t.p_pad + tpb - 1) / tpb;
  const int ncu = device_cus(b.device);
  uint64_t nsplit = std::max<uint64_t>(1, (uint64_t)ncu * (one_per_cu ? 1 : 2) / pblocks);
  nsplit = std::min(nsplit, ntiles);
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  const dim3 grid((unsigned)nsplit, (unsigned)pblocks);
  ProfScope ps(squared ? "sqtmm_gradhyp_dense" : "tmm_gradhyp_dense");
  // 16 hyper-parameters per pass: with two 16-blocks per pass (NHB = 2) the 64 accumulator
  // registers of a wave no longer fit beside the pipeline and the compiler spills them in the
  // inner loop (measured 8.4 ms against 2 x 1.7 ms at C3)
  for (int h0 = 0; h0 < nhyp;) {
    const int rem = nhyp - h0;
    // n4 > 0: 16 + 4 n4 hyper-parameters in this pass; n4 < 0: 4 |n4| only
    int n4 = 0;
    if (fours == 2 && nw == 32 && rem > 16 && rem <= 24) n4 = (rem - 16 + 3) / 4;
    if (fours >= 1 && nw == 32 && rem <= 8) n4 = -((rem + 3) / 4);
    if (n4 > 0 && 2 * (t.Mu + 16 + 4 * n4) * kTlPitch * sizeof(double) > 156 * 1024) n4 = 0;
    const int nh = n4 > 0 ? rem : (n4 < 0 ? rem : std::min(16, rem));
    const int hs = n4 > 0 ? 16 + 4 * n4 : (n4 < 0 ? -4 * n4 : 16);
    double *part = nullptr;
    OB_TRY(b.workspace(nsplit * hs * t.p_pad * sizeof(double), (void **)&part));
    switch (t.W / 2) {
      case 1: OB_TRY((run_tmm_ge0<1, 1>(squared, nw, n4, src, t, dc0.p, nhyp, h0, d_a, grid, ntiles, tps, part))); break;
      case 2: OB_TRY((run_tmm_ge0<2, 1>(squared, nw, n4, src, t, dc0.p, nhyp, h0, d_a, grid, ntiles, tps, part))); break;
      case 3: OB_TRY((run_tmm_ge0<3, 1>(squared, nw, n4, src, t, dc0.p, nhyp, h0, d_a, grid, ntiles, tps, part))); break;
      default: OB_TRY((run_tmm_ge0<4, 1>(squared, nw, n4, src, t, dc0.p, nhyp, h0, d_a, grid, ntiles, tps, part))); break;
    }
    hipLaunchKernelGGL(k_ge0_reduce, dim3((unsigned)((t.p + 255) / 256), (unsigned)nh), dim3(256), 0,
                       cur_stream(), part, (int)nsplit, hs, t.p_pad, (int)t.p, nh,
                       d_out + (uint64_t)h0 * t.p);
    OB_HIP(hipGetLastError());
    h0 += nh;
  }
  return 0;
}

// d_out: p x nhyp column-major (device) = sum_i a_i ge[h, 0]_i B_ik (SQ: squared stores)
int launch_bt_times_ge0(obhip_basis &b, obhip_terms &t, bool squared, const double *d_a,
                        double *d_out) {
  OB_TRY(ensure_bmat(b, t));
  obhip_gradbasis &g = *b.grad;
  const obhip_basis &src = squared ? *g.gbsq : *g.gb;
  const int nhyp = (int)b.model->nhyp();
  std::vector<int> c0(nhyp);
  for (int h = 0; h < nhyp; ++h) c0[h] = g.hyps_h[h].gecol;  // absolute column in the combined array
  DevBuf<int> dc0;
  OB_TRY(dc0.upload(c0.data(), c0.size()));
  const uint64_t ntiles = b.n_pad / kTileRows;
  const unsigned gy = (unsigned)((t.p_pad + 2047) / 2048);
  uint64_t nsplit = std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 1024 / gy));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(b.workspace(nsplit * kNHB * t.p_pad * sizeof(double), (void **)&part));
  ProfScope ps(squared ? "sqtmm_gradhyp_dense" : "tmm_gradhyp_dense");
  for (int h0 = 0; h0 < nhyp; h0 += kNHB) {
    const int nh = std::min(kNHB, nhyp - h0);
    if (squared)
      hipLaunchKernelGGL(k_bt_times_u<true>, dim3((unsigned)nsplit, gy), dim3(512), 0, cur_stream(),
                         b.bmat.p, t.p_pad, src.bm.p, src.md.Mc, dc0.p, nhyp, h0, d_a, b.n, ntiles,
                         tps, part);
    else
      hipLaunchKernelGGL(k_bt_times_u<false>, dim3((unsigned)nsplit, gy), dim3(512), 0, cur_stream(),
                         b.bmat.p, t.p_pad, src.bm.p, src.md.Mc, dc0.p, nhyp, h0, d_a, b.n, ntiles,
                         tps, part);
    hipLaunchKernelGGL(k_btu_reduce, dim3((unsigned)((t.p + 255) / 256), (unsigned)nh), dim3(256), 0,
                       cur_stream(), part, (int)nsplit, t.p_pad, (int)t.p, nh,
                       d_out + (uint64_t)h0 * t.p);
  }
  OB_HIP(hipGetLastError());
  OB_HIP(hipStreamSynchronize(cur_stream()));  // dc0 is a local
  return 0;
}

// (k_tmm_d3, the fused pass over the terms that have a dimension: kernels_grad_d3.hip)

// Views restricted to the terms whose level in hyper-parameter h's dimension is non-zero,
// that dimension dropped and replaced by a pseudo-dimension: the gradient block at level
// t + 1 (ge_sviews) or the delta block at level t (ge_dviews).  idx receives the term
// indices.  nullptr when no term has the dimension.
static void build_sparse_views(obhip_terms &t, const obhip_basis &b) {
  const obhip_model &m = *b.model;
  const uint64_t nh = m.nhyp(), d = t.d, de = d + 2 * nh;
  // (keyed by the dimension of every hyper-parameter: the same terms may meet a basis of another
  // model with as many hyper-parameters laid out differently)
  std::vector<uint64_t> sig(m.hypmatch.begin(), m.hypmatch.end());
  if (t.ge_sviews.size() == nh && t.ge_dviews.size() == nh && t.ge_views_sig == sig) return;
  t.ge_views_sig = sig;
  t.ge_d3_cap.clear();
  t.ge_sviews.clear();
  t.ge_sviews.resize(nh);
  t.ge_dviews.clear();
  t.ge_dviews.resize(nh);
  t.ge_sidx.assign(nh, {});
  for (uint64_t hh = 0; hh < nh; ++hh) {
    const uint64_t l = m.hypmatch[hh];
    std::vector<uint32_t> &ix = t.ge_sidx[hh];
    for (uint64_t k = 0; k < t.p; ++k)
      if (t.lev[k * d + l] > 0) ix.push_back((uint32_t)k);
    if (ix.empty()) continue;
    for (int delta = 0; delta < 2; ++delta) {
      auto v = std::make_unique<obhip_terms>();
      v->no_share = true;  // (a view of the caller's terms: the per-hyper-parameter kernels take the plain tables)
      v->p = ix.size();
      v->d = de;
      v->lev.assign(v->p * de, 0);
      v->maxlev.assign(de, 0);
      const uint64_t pd = d + (delta ? nh : 0) + hh;
      for (uint64_t j = 0; j < v->p; ++j) {
        const uint64_t k = ix[j];
        uint64_t nnz = 1;
        for (uint64_t q = 0; q < d; ++q) {
          const uint32_t lv = q == l ? 0u : t.lev[k * d + q];
          v->lev[j * de + q] = lv;
          v->maxlev[q] = std::max<int64_t>(v->maxlev[q], lv);
          nnz += lv > 0;
        }
        const uint32_t gl = t.lev[k * d + l] + (delta ? 0 : 1);
        v->lev[j * de + pd] = gl;
        v->maxlev[pd] = std::max<int64_t>(v->maxlev[pd], gl);
        v->nnz_total += nnz;
        v->max_nnz = std::max(v->max_nnz, nnz);
      }
      (delta ? t.ge_dviews : t.ge_sviews)[hh] = std::move(v);
    }
  }
  // The gradient views are contracted a few hyper-parameters at a time: their term lists
  // concatenated into groups whose used columns (the value columns of the other dimensions
  // plus the groups' own gradient columns) stay within what the term-per-lane kernel can
  // prefetch and keep two workgroups per CU on (128 columns).  One list of all of them
  // (round 1) put ~300 columns and 3 p terms into a single pass: one workgroup per CU, no
  // prefetch, 16.5 ms per call at p = 4096, 16 hyper-parameters.
  const char *ge = getenv("OBHIP_GRAD_GROUP_MU");
  const size_t mu_cap = ge ? (size_t)std::max(1, atoi(ge)) : 128;
  auto group_views = [&](const std::vector<std::unique_ptr<obhip_terms>> &views,
                         std::vector<obhip_terms::GeGroup> &groups) {
    groups.clear();
    std::set<std::pair<uint32_t, uint32_t>> used;  // (dimension of the view, level > 0)
    auto close_group = [&](obhip_terms::GeGroup &g) {
      if (g.hyps.empty()) return;
      auto v = std::make_unique<obhip_terms>();
      v->no_share = true;  // (a view of the caller's terms: the per-hyper-parameter kernels take the plain tables)
      v->p = g.off.back();
      v->d = de;
      v->lev.resize(v->p * de);
      v->maxlev.assign(de, 0);
      for (size_t j = 0; j < g.hyps.size(); ++j) {
        const obhip_terms *sv = views[g.hyps[j]].get();
        std::copy(sv->lev.begin(), sv->lev.end(), v->lev.begin() + g.off[j] * de);
        for (uint64_t q = 0; q < de; ++q) v->maxlev[q] = std::max(v->maxlev[q], sv->maxlev[q]);
        v->nnz_total += sv->nnz_total;
        v->max_nnz = std::max(v->max_nnz, sv->max_nnz);
      }
      g.v = std::move(v);
      groups.push_back(std::move(g));
    };
    obhip_terms::GeGroup cur;
    cur.off.push_back(0);
    for (uint64_t hh = 0; hh < nh; ++hh) {
      const obhip_terms *sv = views[hh].get();
      if (!sv) continue;
      std::set<std::pair<uint32_t, uint32_t>> mine;
      for (uint64_t j = 0; j < sv->p; ++j)
        for (uint64_t q = 0; q < de; ++q)
          if (sv->lev[j * de + q] > 0) mine.insert({(uint32_t)q, sv->lev[j * de + q]});
      std::set<std::pair<uint32_t, uint32_t>> both = used;
      both.insert(mine.begin(), mine.end());
      if (!cur.hyps.empty() && both.size() + 1 > mu_cap) {
        close_group(cur);
        cur = obhip_terms::GeGroup();
        cur.off.push_back(0);
        used = mine;
      } else {
        used.swap(both);
      }
      cur.hyps.push_back(hh);
      cur.off.push_back(cur.off.back() + sv->p);
    }
    close_group(cur);
  };
  group_views(t.ge_sviews, t.ge_sgroups);
  group_views(t.ge_dviews, t.ge_dgroups);  // the same for the delta views (grad_mm_dot_dev)
}

// The groups of the fused gradient passes (obhip_terms::GeD3, k_tmm_d3) for the column layout of
// b's gradient basis; false when some view does not fit the kernel (more than 8 column slots, a
// tile beyond the prefetch registers): the callers then take the per-hyper-parameter passes.
static bool build_d3_groups(obhip_terms &t, const obhip_basis &b) {
  // OBHIP_GRAD_D3=0 (read per call: the tests switch it): the per-hyper-parameter passes
  const char *sw = getenv("OBHIP_GRAD_D3");
  if (sw && atoi(sw) == 0) return false;
  // the tables hold column numbers of the gradient basis: keyed by everything those depend on
  // (level caps, first column of every dimension, first delta column of every hyper-parameter)
  const obhip_model &m = *b.model;
  const obhip_gradbasis &g = *b.grad;
  const std::vector<DimDesc> &dims = b.md.dims_h;
  const uint64_t d = t.d;
  std::vector<int64_t> key = b.md.cap;
  for (uint64_t l = 0; l < d; ++l) key.push_back(dims[l].ccol0);
  for (const GradHyp &gh : g.hyps_h) {
    key.push_back(gh.dim);
    key.push_back(gh.dcol);
  }
  if (t.ge_d3_cap == key) return t.ge_d3_ok;
  HostTimer ht("build_d3_groups (rebuild)");
  build_sparse_views(t, b);
  t.ge_d3.clear();
  t.ge_d3_cap = key;
  t.ge_d3_ok = false;
  // two blocks per CU: 2 x Mu x 65 x 8 B <= 160 KB
  constexpr size_t kMuCap = 152;
  static_assert(kMuCap <= (size_t)kTlWaves * kD3Pre, "the tile is prefetched into registers");  // kD3Pre: obhip_internal.h
  struct Member {
    uint64_t l, h0;
    int nh;
    std::set<uint32_t> cols;
    uint64_t maxw = 0;
  };
  std::vector<Member> mem;
  for (uint64_t l = 0; l < d; ++l)
    for (uint64_t h0 = m.hypst[l]; h0 < m.hypst[l + 1]; h0 += 2) {
      Member M;
      M.l = l;
      M.h0 = h0;
      M.nh = (int)std::min<uint64_t>(2, m.hypst[l + 1] - h0);
      const std::vector<uint32_t> &ix = t.ge_sidx[h0];
      if (ix.empty()) continue;
      for (uint32_t k : ix) {
        uint64_t w = 1 + (uint64_t)M.nh;
        for (uint64_t q = 0; q < d; ++q) {
          const uint32_t lv = t.lev[(uint64_t)k * d + q];
          if (lv == 0) continue;
          // (bits 28-29: how the column is staged -- 0 as it is, 1 times the row's basescale (delta
          // columns), 2 times basescale x second weight (the dimension's OWN factor): k_tmm_d3)
          if (q == l) {
            M.cols.insert((uint32_t)(dims[q].ccol0 + lv - 1) | kD3Own);
            for (int j = 0; j < M.nh; ++j) M.cols.insert((uint32_t)(g.hyps_h[h0 + j].dcol + lv - 1) | kD3Delta);
          } else {
            M.cols.insert((uint32_t)(dims[q].ccol0 + lv - 1));
            ++w;
          }
        }
        M.maxw = std::max(M.maxw, w);
      }
      if (M.cols.size() + 1 > kMuCap || M.maxw > 8) return false;
      mem.push_back(std::move(M));
    }
  // consecutive members with the same number of delta columns share a launch while their columns fit
  size_t i = 0;
  while (i < mem.size()) {
    std::set<uint32_t> used = mem[i].cols;
    uint64_t maxw = mem[i].maxw;
    size_t j = i + 1;
    while (j < mem.size() && mem[j].nh == mem[i].nh) {
      std::set<uint32_t> both = used;
      both.insert(mem[j].cols.begin(), mem[j].cols.end());
      if (both.size() + 1 > kMuCap) break;
      used.swap(both);
      maxw = std::max(maxw, mem[j].maxw);
      ++j;
    }
    obhip_terms::GeD3 G;
    G.nh = mem[i].nh;
    G.off.push_back(0);
    for (size_t q = i; q < j; ++q) {
      G.hyp0.push_back(mem[q].h0);
      G.off.push_back(G.off.back() + t.ge_sidx[mem[q].h0].size());
    }
    auto v = std::make_unique<obhip_terms>();
    v->no_share = true;
    v->p = G.off.back();
    v->d = 0;
    v->p_pad = (v->p + 255) / 256 * 256;
    v->W = std::max<uint64_t>(4, (maxw + 1) / 2 * 2);
    std::vector<uint32_t> ul(1, 0u);  // the ones column first
    ul.insert(ul.end(), used.begin(), used.end());
    v->Mu = ul.size();
    std::map<uint32_t, uint16_t> pos;
    for (size_t u = 0; u < ul.size(); ++u) pos[ul[u]] = (uint16_t)u;
    const uint64_t W = v->W;
    std::vector<uint16_t> hc(v->p_pad * W, 0);
    std::vector<uint32_t> nz(v->p, 0), order(v->p_pad);
    for (size_t q = i; q < j; ++q) {
      const Member &M = mem[q];
      const std::vector<uint32_t> &ix = t.ge_sidx[M.h0];
      for (size_t e = 0; e < ix.size(); ++e) {
        const uint64_t k = ix[e], vt = G.off[q - i] + e;
        uint64_t others = 0;
        for (uint64_t a = 0; a < d; ++a) others += a != M.l && t.lev[k * d + a] > 0;
        uint64_t w = W - (others + 1 + (uint64_t)M.nh);
        nz[vt] = (uint32_t)(others + 1 + (uint64_t)M.nh);
        for (uint64_t a = 0; a < d; ++a) {  // the other factors keep their dimension order
          const uint32_t lv = t.lev[k * d + a];
          if (a != M.l && lv > 0) hc[vt * W + w++] = pos[(uint32_t)(dims[a].ccol0 + lv - 1)];
        }
        const uint32_t lv = t.lev[k * d + M.l];
        hc[vt * W + w++] = pos[(uint32_t)(dims[M.l].ccol0 + lv - 1) | kD3Own];
        for (int a = 0; a < M.nh; ++a)
          hc[vt * W + w++] = pos[(uint32_t)(g.hyps_h[M.h0 + a].dcol + lv - 1) | kD3Delta];
      }
    }
    for (uint64_t k = 0; k < v->p_pad; ++k) order[k] = (uint32_t)k;
    std::stable_sort(order.begin(), order.begin() + v->p, [&](uint32_t a, uint32_t c) { return nz[a] > nz[c]; });
    if (v->cols.upload(hc.data(), hc.size()) || v->sperm.upload(order.data(), order.size()) ||
        v->ucol.upload(ul.data(), ul.size()))
      return false;
    G.v = std::move(v);
    t.ge_d3.push_back(std::move(G));
    i = j;
  }
  t.ge_d3_ok = true;
  return true;
}

obhip_terms *grad_view_sparse(obhip_terms &t, const obhip_basis &b, uint64_t h,
                              const std::vector<uint32_t> **idx) {
  build_sparse_views(t, b);
  *idx = &t.ge_sidx[h];
  return t.ge_sviews[h].get();
}

obhip_terms *grad_view_delta(obhip_terms &t, const obhip_basis &b, uint64_t h,
                             const std::vector<uint32_t> **idx) {
  build_sparse_views(t, b);
  *idx = &t.ge_sidx[h];
  return t.ge_dviews[h].get();
}

// View of the terms for hyper-parameter h: dimension hypmatch[h] dropped, pseudo-dimension
// d + h at level t + 1 (so even level 0 picks up its gradient column).
obhip_terms *grad_view(obhip_terms &t, const obhip_basis &b, uint64_t h) {
  const obhip_model &m = *b.model;
  const uint64_t nh = m.nhyp(), d = t.d, de = d + 2 * nh;
  if (t.ge_views.size() != nh || t.ge_full_sig != m.hypmatch) {
    t.ge_views.clear();
    t.ge_views.resize(nh);
    t.ge_full_sig = m.hypmatch;
  }
  if (!t.ge_views[h]) {
    auto v = std::make_unique<obhip_terms>();
    v->no_share = true;
    v->p = t.p;
    v->d = de;
    v->lev.assign(t.p * de, 0);
    v->maxlev.assign(de, 0);
    const uint64_t l = m.hypmatch[h];
    for (uint64_t k = 0; k < t.p; ++k) {
      uint64_t nnz = 0;
      for (uint64_t q = 0; q < d; ++q) {
        const uint32_t lv = q == l ? 0u : t.lev[k * d + q];
        v->lev[k * de + q] = lv;
        v->maxlev[q] = std::max<int64_t>(v->maxlev[q], lv);
        nnz += lv > 0;
      }
      const uint32_t gl = t.lev[k * d + l] + 1;
      v->lev[k * de + d + h] = gl;
      v->maxlev[d + h] = std::max<int64_t>(v->maxlev[d + h], gl);
      ++nnz;
      v->nnz_total += nnz;
      v->max_nnz = std::max(v->max_nnz, nnz);
    }
    t.ge_views[h] = std::move(v);
  }
  return t.ge_views[h].get();
}

}  // namespace obhip

using namespace obhip;

namespace {

int d2h(void *dst, const void *src, size_t bytes) {
  OB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  return 0;
}

int check_grad_args(const obhip_basis *b, const obhip_terms *t) {
  if (!b || !t) return fail(OBHIP_ERR_INVALID, "gradhyp: null argument");
  if (t->d != b->model->d) return fail(OBHIP_ERR_INVALID, "terms and model disagree on d");
  for (uint64_t l = 0; l < t->d; ++l)
    if (t->maxlev[l] > b->md.cap[l])
      return fail(OBHIP_ERR_INVALID, "terms use a level above the basis' level cap");
  return 0;
}

// out_gradhyp (host, n x nhyp) = d(B a)/dhyp_h (squared: d(B^2 a)/dhyp_h); d_M receives
// B a (B^2 a).  One k_mm pass over all terms for M, per hyper-parameter one k_mm pass over
// the terms that have its dimension (delta view), then out[:, h] = ge[h, 0] M + that.
int mm_gradhyp_dev(obhip_basis &b, obhip_terms &t, bool squared, const double *a, const double *d_a,
                   double *d_M, DevBuf<double> &dge) {
  HostTimer ht("mm_gradhyp_dev");
  obhip_gradbasis &g = *b.grad;
  const obhip_basis &src = squared ? *g.gbsq : *g.gb;
  const uint64_t nh = b.model->nhyp(), n = b.n;
  DevBuf<double> dacat;
  DevBuf<int> dc0;
  OB_TRY(dge.alloc(n * nh));
  OB_HIP(hipMemsetAsync(dge.p, 0, n * nh * sizeof(double), cur_stream()));
  // coefficients of the restricted views, concatenated
  std::vector<double> acat;
  std::vector<uint64_t> off(nh + 1, 0);
  for (uint64_t h = 0; h < nh; ++h) {
    const std::vector<uint32_t> *idx = nullptr;
    grad_view_delta(t, b, h, &idx);
    for (uint32_t k : *idx) acat.push_back(a[k]);
    off[h + 1] = acat.size();
  }
  std::vector<int> c0(nh);
  for (uint64_t h = 0; h < nh; ++h) c0[h] = g.hyps_h[h].gecol;
  OB_TRY(dc0.upload(c0.data(), c0.size()));
  if (!acat.empty()) OB_TRY(dacat.upload(acat.data(), acat.size()));
  OB_TRY(launch_mm(b, t, d_a, d_M, squared));
  {
    ProfScope ps(squared ? "sqmm_gradhyp" : "mm_gradhyp");
    for (uint64_t h = 0; h < nh; ++h) {
      const std::vector<uint32_t> *idx = nullptr;
      obhip_terms *v = grad_view_delta(t, b, h, &idx);
      if (!v) continue;
      OB_TRY(launch_mm(src, *v, dacat.p + off[h], dge.p + h * n, false));
    }
    hipLaunchKernelGGL(k_ge0_combine, dim3((unsigned)((n + 255) / 256), (unsigned)nh), dim3(256), 0,
                       cur_stream(), src.bm.p, src.md.Mc, dc0.p, d_M, n, dge.p);
    OB_HIP(hipGetLastError());
  }
  OB_HIP(hipStreamSynchronize(cur_stream()));  // dc0, dacat are locals
  return 0;
}

int mm_gradhyp_all(obhip_basis &b, obhip_terms &t, bool squared, const double *a, const double *d_a,
                   double *d_M, double *out_gradhyp) {
  DevBuf<double> dge;
  OB_TRY(mm_gradhyp_dev(b, t, squared, a, d_a, d_M, dge));
  return d2h(out_gradhyp, dge.p, b.n * b.model->nhyp() * sizeof(double));
}

// part[h][blk] = partial sums of w_i G[h][i] over the rows blk, blk + gridDim.x, ...
__global__ void __launch_bounds__(256)
k_wdot1(const double *__restrict__ G, const double *__restrict__ w, uint64_t n,
        double *__restrict__ part) {
  __shared__ double red[256];
  const double *g = G + (uint64_t)blockIdx.y * n;
  double s = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    s = fma(w[i], g[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(uint64_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
__global__ void k_wdot2(const double *__restrict__ part, int nblk, double *__restrict__ out) {
  const int h = blockIdx.x;
  if (threadIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part[(uint64_t)h * nblk + i];
  out[h] = s;
}

// part[h][blk] = partial sums of w_i M_i ge[h, 0]_i (the level-0 gradient column of hyper-parameter
// h in the tiled gradient basis)
__global__ void __launch_bounds__(256)
k_wdot_ge0(const double *__restrict__ bm, uint64_t Mtot, const int *__restrict__ ge0col,
           const double *__restrict__ M, const double *__restrict__ w, uint64_t n,
           double *__restrict__ part) {
  __shared__ double red[256];
  const uint64_t c = (uint64_t)ge0col[blockIdx.y];
  double s = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    s = fma(w[i] * M[i], bm[((i >> 6) * Mtot + c) * kTileRows + (i & 63)], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(uint64_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

// out_host[h] = w^T d(B a)/dhyp_h without the n x nhyp matrix.  d(B a)/dhyp_h = ge[h, 0] % (B a) +
// B_delta,h a_h (mm_gradhyp_dev), and the likelihoods only ever contract it with one row vector
// (loglik_gauss.cpp:127, loglik_std.cpp:143, loglik_gda.cpp:141: gradhyp = yhat_gradhyp^T r), so
//   out[h] = sum_i w_i ge[h, 0]_i M_i  +  a_h^T (B_delta,h^T w):
// one pass of the term-per-lane B^T kernel per group of concatenated delta views (accumulators
// stay in the lanes: no cross-lane sum per row, which is what the nhyp restricted k_mm passes of
// mm_gradhyp_dev pay) and one streaming pass for the level-0 columns.  d_M = B a.
int mm_gradhyp_dot_dev(obhip_basis &b, obhip_terms &t, const double *a, const double *d_M,
                       const double *d_w, double *out_host) {
  HostTimer ht("mm_gradhyp_dot_dev");
  obhip_gradbasis &g = *b.grad;
  const obhip_basis &src = *g.gb;
  const uint64_t nh = b.model->nhyp(), n = b.n;
  if (nh == 0) return 0;
  constexpr int nblk = 256;
  std::vector<int> c0(nh);
  for (uint64_t h = 0; h < nh; ++h) c0[h] = g.hyps_h[h].gecol;
  DevBuf<int> dc0;
  DevBuf<double> dpart, dres, dall;
  OB_TRY(dc0.upload(c0.data(), c0.size()));
  OB_TRY(dpart.alloc(nh * nblk));
  OB_TRY(dres.alloc(nh));
  ProfScope ps("mm_gradhyp_dot");
  hipLaunchKernelGGL(k_wdot_ge0, dim3(nblk, (unsigned)nh), dim3(256), 0, cur_stream(), src.bm.p,
                     src.md.Mc, dc0.p, d_M, d_w, n, dpart.p);
  hipLaunchKernelGGL(k_wdot2, dim3((unsigned)nh), dim3(64), 0, cur_stream(), dpart.p, nblk, dres.p);
  OB_HIP(hipGetLastError());
  OB_TRY(d2h(out_host, dres.p, nh * sizeof(double)));
  const std::vector<uint32_t> *idx = nullptr;
  grad_view_delta(t, b, 0, &idx);  // builds the views and their groups
  std::vector<double> tmp;
  auto contract = [&](uint64_t h, const double *v) {  // out[h] += a_h^T v over the view's terms
    const std::vector<uint32_t> &ix = t.ge_sidx[h];
    long double s = 0;
    for (size_t q = 0; q < ix.size(); ++q) s += (long double)a[ix[q]] * v[q];
    out_host[h] += (double)s;
  };
  if (build_d3_groups(t, b)) {  // one pass per group of dimensions (k_tmm_d3)
    for (const obhip_terms::GeD3 &grp : t.ge_d3) {
      const uint64_t vp = grp.v->p;
      OB_TRY(dall.alloc(std::max<uint64_t>(2 * grp.nh * vp, dall.n)));
      OB_TRY(launch_tmm_d3(b, grp, 1, d_w, nullptr, dall.p));
      tmp.resize(grp.nh * vp);
      OB_TRY(d2h(tmp.data(), dall.p, tmp.size() * sizeof(double)));
      for (size_t j = 0; j < grp.hyp0.size(); ++j)
        for (int hh = 0; hh < grp.nh; ++hh) contract(grp.hyp0[j] + hh, tmp.data() + hh * vp + grp.off[j]);
    }
    return 0;
  }
  for (obhip_terms::GeGroup &grp : t.ge_dgroups) {
    obhip_terms *all = grp.v.get();
    if (all->prepare(src.md.cap, src.md.dims_h) == 0 && all->Mu <= 296) {
      OB_TRY(dall.alloc(std::max<uint64_t>(all->p, dall.n)));
      OB_TRY(launch_tmm(src, *all, d_w, dall.p, false));
      tmp.resize(all->p);
      OB_TRY(d2h(tmp.data(), dall.p, tmp.size() * sizeof(double)));
      for (size_t j = 0; j < grp.hyps.size(); ++j) contract(grp.hyps[j], tmp.data() + grp.off[j]);
      continue;
    }
    for (uint64_t h : grp.hyps) {  // beyond one LDS tile: hyper-parameter by hyper-parameter
      obhip_terms *v = grad_view_delta(t, b, h, &idx);
      if (!v) continue;
      OB_TRY(dall.alloc(std::max<uint64_t>(v->p, dall.n)));
      OB_TRY(launch_tmm(src, *v, d_w, dall.p, false));
      tmp.resize(v->p);
      OB_TRY(d2h(tmp.data(), dall.p, tmp.size() * sizeof(double)));
      contract(h, tmp.data());
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int obhip_basis_getmat_gradhyp(const obhip_basis *bc, const obhip_terms *tc, double *out) {
  OB_TRY(check_grad_args(bc, tc));
  if (!out) return fail(OBHIP_ERR_INVALID, "getmat_gradhyp: null argument");
  obhip_basis &b = *const_cast<obhip_basis *>(bc);
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  OB_TRY(ensure_gradbasis(b));
  DevBuf<double> tmp;
  OB_TRY(tmp.alloc(b.n * t.p));
  for (uint64_t h = 0; h < b.model->nhyp(); ++h) {
    OB_TRY(launch_getmat(*b.grad->gb, *grad_view(t, b, h), tmp.p));
    OB_TRY(d2h(out + h * b.n * t.p, tmp.p, b.n * t.p * sizeof(double)));
  }
  return 0;
}

int obhip_basis_mm_gradhyp(const obhip_basis *bc, const obhip_terms *tc, const double *a,
                           double *out, double *out_gradhyp) {
  OB_TRY(check_grad_args(bc, tc));
  if (!a || !out_gradhyp) return fail(OBHIP_ERR_INVALID, "mm_gradhyp: null argument");
  obhip_basis &b = *const_cast<obhip_basis *>(bc);
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  OB_TRY(ensure_gradbasis(b));
  DevBuf<double> da, dout;
  OB_TRY(da.upload(a, t.p));
  OB_TRY(dout.alloc(b.n));
  OB_TRY(mm_gradhyp_all(b, t, false, a, da.p, dout.p, out_gradhyp));
  if (out) OB_TRY(d2h(out, dout.p, b.n * sizeof(double)));
  return 0;
}

// w^T d(B a)/dhyp without moving the n x nhyp matrix to the host: what the likelihoods need
// of matmul_gradhyp (loglik_gauss.cpp:127, loglik_std.cpp:143: gradhyp = r^T yhat_gradhyp)
int obhip_basis_mm_gradhyp_dot(const obhip_basis *bc, const obhip_terms *tc, const double *a,
                               const double *w, double *out, double *out_dot) {
  OB_TRY(check_grad_args(bc, tc));
  if (!a || !w || !out_dot) return fail(OBHIP_ERR_INVALID, "mm_gradhyp_dot: null argument");
  obhip_basis &b = *const_cast<obhip_basis *>(bc);
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  OB_TRY(ensure_gradbasis(b));
  DevBuf<double> da, dw, dout;
  OB_TRY(da.upload(a, t.p));
  OB_TRY(dw.upload(w, b.n));
  OB_TRY(dout.alloc(b.n));
  OB_TRY(launch_mm(b, t, da.p, dout.p, false));
  OB_TRY(mm_gradhyp_dot_dev(b, t, a, dout.p, dw.p, out_dot));
  if (out) OB_TRY(d2h(out, dout.p, b.n * sizeof(double)));
  return 0;
}

static int tmm_gradhyp_all(obhip_basis &b, obhip_terms &t, bool squared, const double *d_a,
                           double *out_gradhyp);

// products on the squared stores: ob$sqmm_gradhyp / ob$sqtmm_gradhyp
// (modandbase.cpp:798-809, 845-856)
// a == nullptr with transposed: the all-ones vector, filled on the device (sqcolsums_gradhyp)
static int sq_gradhyp(const obhip_basis *bc, const obhip_terms *tc, const double *a,
                      double *out_gradhyp, bool transposed) {
  OB_TRY(check_grad_args(bc, tc));
  if ((!a && !transposed) || !out_gradhyp)
    return fail(OBHIP_ERR_INVALID, "sq*_gradhyp: null argument");
  obhip_basis &b = *const_cast<obhip_basis *>(bc);
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  // (the transposed product asks for the squared store itself, where it needs it)
  OB_TRY(transposed ? ensure_gradbasis(b) : ensure_gradbasis_sq(b));
  const uint64_t nin = transposed ? b.n : t.p, nout = transposed ? t.p : b.n;
  DevBuf<double> da, dout;
  if (a) {
    OB_TRY(da.upload(a, nin));
  } else {
    OB_TRY(da.alloc(nin));
    OB_TRY(launch_fill(da.p, nin, 1.0));
  }
  OB_TRY(dout.alloc(nout));
  if (transposed) return tmm_gradhyp_all(b, t, true, da.p, out_gradhyp);
  return mm_gradhyp_all(b, t, true, a, da.p, dout.p, out_gradhyp);
}

int obhip_basis_sqmm_gradhyp(const obhip_basis *b, const obhip_terms *t, const double *a,
                             double *out_gradhyp) {
  return sq_gradhyp(b, t, a, out_gradhyp, false);
}

int obhip_basis_sqtmm_gradhyp(const obhip_basis *b, const obhip_terms *t, const double *a,
                              double *out_gradhyp) {
  return sq_gradhyp(b, t, a, out_gradhyp, true);
}

int obhip_basis_sqcolsums_gradhyp(const obhip_basis *b, const obhip_terms *t, double *out_gradhyp) {
  if (!b) return fail(OBHIP_ERR_INVALID, "sqcolsums_gradhyp: null argument");
  return sq_gradhyp(b, t, nullptr, out_gradhyp, true);  // ones: modandbase.cpp:875-879
}

int obhip_basis_residvar_gradhyp(const obhip_basis *b, const obhip_terms *t, const obhip_model *m,
                                 double *out_gradhyp) {
  if (!b || !t || !m || !out_gradhyp) return fail(OBHIP_ERR_INVALID, "residvar_gradhyp: null argument");
  if (m != b->model) return fail(OBHIP_ERR_INVALID, "residvar_gradhyp: basis belongs to another model");
  const uint64_t p = t->p, n = b->n, nh = m->nhyp(), d = m->d;
  // varc = getvar(terms), lvarge = getlvar_gradhyp(terms) (modandbase.cpp:350-356, 364-379)
  std::vector<double> varc(p), col(p), tmp(n);
  for (uint64_t k = 0; k < p; ++k) {
    double sv = 0;
    for (uint64_t l = 0; l < d; ++l) sv += m->basisvar[m->knotptst[l] + t->lev[k * d + l]];
    varc[k] = std::exp(sv);
  }
  OB_TRY(sq_gradhyp(b, t, varc.data(), out_gradhyp, false));  // :909
  for (uint64_t i = 0; i < n * nh; ++i) out_gradhyp[i] = -out_gradhyp[i];
  for (uint64_t h = 0; h < nh; ++h) {  // :912-919: minus B^2 (lvarge % varc), column by column
    const uint64_t l = m->hypmatch[h];
    for (uint64_t k = 0; k < p; ++k)
      col[k] = varc[k] * m->logbasisvar_gradhyp[m->gest[h] + t->lev[k * d + l]];
    OB_TRY(obhip_basis_sqmm(b, t, col.data(), 1, tmp.data()));
    for (uint64_t i = 0; i < n; ++i) out_gradhyp[h * n + i] -= tmp[i];
  }
  return 0;
}

// out_gradhyp (host, p x nhyp) = sum_i a_i dB_ik/dhyp_h: one streaming pass over the design
// matrix for the terms without the hyper-parameter's dimension, k_tmm on the restricted views
// for those with it; one k_tmm pass per hyper-parameter when the design matrix does not fit
static int tmm_gradhyp_all(obhip_basis &b, obhip_terms &t, bool squared, const double *d_a,
                           double *out_gradhyp) {
  HostTimer ht("tmm_gradhyp_all");
  const uint64_t nh = b.model->nhyp(), p = t.p;
  DevBuf<double> dout;
  OB_TRY(dout.alloc(p * nh));
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  static const bool stream_b = getenv("OBHIP_GRAD_STREAM_B") != nullptr;  // the older dense pass
  const bool onfly = tmm_ge0_supports(t) && !stream_b;
  // the squared store of the gradient basis (a second array of its size) only where the older
  // passes run: k_tmm_ge0 and k_tmm_d3 form the squares themselves
  const bool fused = onfly && build_d3_groups(t, b);
  if (squared && !fused) OB_TRY(ensure_gradbasis_sq(b));
  const obhip_basis &src = squared && !fused ? *b.grad->gbsq : *b.grad->gb;
  if (!onfly && !gram_panel_supports(b, t)) {
    for (uint64_t h = 0; h < nh; ++h) {
      OB_TRY(launch_tmm(src, *grad_view(t, b, h), d_a, dout.p, false));
      OB_TRY(d2h(out_gradhyp + h * p, dout.p, p * sizeof(double)));
    }
    return 0;
  }
  if (onfly)
    OB_TRY(launch_tmm_ge0(b, t, squared, d_a, dout.p));
  else
    OB_TRY(launch_bt_times_ge0(b, t, squared, d_a, dout.p));
  OB_TRY(d2h(out_gradhyp, dout.p, p * nh * sizeof(double)));
  std::vector<double> tmp;
  const std::vector<uint32_t> *idx = nullptr;
  if (fused) {
    // the dense pass treated EVERY term as if it lacked the dimension (ge[h, 0] B, squared:
    // 2 ge[h, 0] B^2); the terms that have it add their delta part, squared: 2 u2 (k_tmm_d3)
    DevBuf<double> dall;
    for (const obhip_terms::GeD3 &grp : t.ge_d3) {
      const uint64_t vp = grp.v->p;
      OB_TRY(dall.alloc(std::max<uint64_t>(2 * grp.nh * vp, dall.n)));
      OB_TRY(launch_tmm_d3(b, grp, squared ? 2 : 1, d_a, d_a, dall.p));
      tmp.resize(grp.nh * vp);
      OB_TRY(d2h(tmp.data(), dall.p + (squared ? grp.nh * vp : 0), tmp.size() * sizeof(double)));
      for (size_t j = 0; j < grp.hyp0.size(); ++j)
        for (int hh = 0; hh < grp.nh; ++hh) {
          const uint64_t h = grp.hyp0[j] + hh;
          const std::vector<uint32_t> &ix = t.ge_sidx[h];
          const double *u = tmp.data() + hh * vp + grp.off[j];
          for (size_t q = 0; q < ix.size(); ++q) out_gradhyp[h * p + ix[q]] += squared ? 2.0 * u[q] : u[q];
        }
    }
    return 0;
  }
  // the terms that have the hyper-parameter's dimension: one pass per group of restricted
  // views (build_sparse_views), overwriting the dense pass' entries
  grad_view_sparse(t, b, 0, &idx);  // builds the views
  for (obhip_terms::GeGroup &g : t.ge_sgroups) {
    obhip_terms *all = g.v.get();
    if (all->prepare(src.md.cap, src.md.dims_h) == 0 && all->Mu <= 296) {
      DevBuf<double> dall;
      OB_TRY(dall.alloc(all->p));
      OB_TRY(launch_tmm(src, *all, d_a, dall.p, false));
      tmp.resize(all->p);
      OB_TRY(d2h(tmp.data(), dall.p, tmp.size() * sizeof(double)));
      for (size_t j = 0; j < g.hyps.size(); ++j) {
        const uint64_t h = g.hyps[j];
        const std::vector<uint32_t> &ix = t.ge_sidx[h];
        const double *src_h = tmp.data() + g.off[j];
        for (size_t q = 0; q < ix.size(); ++q) out_gradhyp[h * p + ix[q]] = src_h[q];
      }
      continue;
    }
    for (uint64_t h : g.hyps) {  // beyond one LDS tile: hyper-parameter by hyper-parameter
      obhip_terms *v = grad_view_sparse(t, b, h, &idx);
      if (!v) continue;
      OB_TRY(launch_tmm(src, *v, d_a, dout.p, false));
      tmp.resize(idx->size());
      OB_TRY(d2h(tmp.data(), dout.p, tmp.size() * sizeof(double)));
      for (size_t q = 0; q < tmp.size(); ++q) out_gradhyp[h * p + (*idx)[q]] = tmp[q];
    }
  }
  return 0;
}

int obhip_basis_tmm_gradhyp(const obhip_basis *bc, const obhip_terms *tc, const double *a,
                            double *out, double *out_gradhyp) {
  OB_TRY(check_grad_args(bc, tc));
  if (!a || !out_gradhyp) return fail(OBHIP_ERR_INVALID, "tmm_gradhyp: null argument");
  obhip_basis &b = *const_cast<obhip_basis *>(bc);
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  OB_TRY(ensure_gradbasis(b));
  DevBuf<double> da, dout;
  OB_TRY(da.upload(a, b.n));
  if (out) {
    OB_TRY(dout.alloc(t.p));
    OB_TRY(launch_tmm(b, t, da.p, dout.p, false));
    OB_TRY(d2h(out, dout.p, t.p * sizeof(double)));
  }
  return tmm_gradhyp_all(b, t, false, da.p, out_gradhyp);
}

}  // extern "C"

// ---- device-level forms for the likelihood classes (lpdf.cpp): inputs and the n-sized
// results stay in HBM, only p- and nhyp-sized results go to the host -----------------------
namespace obhip {

// dge (n x nhyp, device) = d(B a)/dhyp (squared: d(B^2 a)/dhyp), d_M = B a (B^2 a);
// a_host / d_a: the same p coefficients on the host and on the device
int grad_mm_dev(obhip_basis &b, obhip_terms &t, bool squared, const double *a_host, const double *d_a,
                double *d_M, DevBuf<double> &dge) {
  OB_TRY(check_grad_args(&b, &t));
  OB_TRY(squared ? ensure_gradbasis_sq(b) : ensure_gradbasis(b));
  return mm_gradhyp_dev(b, t, squared, a_host, d_a, d_M, dge);
}

// out_host[h] = w^T d(B a)/dhyp_h from d_M = B a (mm_gradhyp_dot_dev)
int grad_mm_dot_dev(obhip_basis &b, obhip_terms &t, const double *a_host, const double *d_M,
                    const double *d_w, double *out_host) {
  OB_TRY(check_grad_args(&b, &t));
  OB_TRY(ensure_gradbasis(b));
  return mm_gradhyp_dot_dev(b, t, a_host, d_M, d_w, out_host);
}

// Both hyper-gradient contractions of a likelihood in one sweep (k_tmm_d3 MODE 3):
//   out_dot (nhyp)       = w1^T d(B a)/dhyp            (grad_mm_dot_dev; d_M = B a)
//   out_sq  (p x nhyp)   = sum_i w2_i d(B^2)_ik/dhyp   (grad_tmm_host squared; d_w2 null: ones)
// kNotFused when the terms do not fit the fused kernels: the caller then asks for the two singly.
int grad_dual_dev(obhip_basis &b, obhip_terms &t, const double *a_host, const double *d_M, const double *d_w1,
                  const double *d_w2, double *out_dot, double *out_sq) {
  HostTimer ht("grad_dual_dev");
  OB_TRY(check_grad_args(&b, &t));
  OB_TRY(ensure_gradbasis(b));
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (!tmm_ge0_supports(t) || !build_d3_groups(t, b)) return kNotFused;
  obhip_gradbasis &g = *b.grad;
  const obhip_basis &src = *g.gb;
  const uint64_t nh = b.model->nhyp(), n = b.n, p = t.p;
  if (nh == 0) return 0;
  DevBuf<double> ones, dpart, dres;
  if (!d_w2) {
    OB_TRY(ones.alloc(n));
    OB_TRY(launch_fill(ones.p, n, 1.0));
    d_w2 = ones.p;
  }
  // The whole sweep is enqueued before the host reads anything: [nh level-0 dot products |
  // p x nh dense part | per group its 2 nh x view-terms block], ONE copy at the end (every blocking
  // copy in between would drain the queue: ten of them per evaluation in the first version).
  uint64_t total = nh + p * nh;
  std::vector<uint64_t> goff;
  for (const obhip_terms::GeD3 &grp : t.ge_d3) {
    goff.push_back(total);
    total += 2 * (uint64_t)grp.nh * grp.v->p;
  }
  constexpr int nblk = 256;
  OB_TRY(dpart.alloc(nh * nblk));
  OB_TRY(dres.alloc(total));
  hipLaunchKernelGGL(k_wdot_ge0, dim3(nblk, (unsigned)nh), dim3(256), 0, cur_stream(), src.bm.p, src.md.Mc,
                     g.ge0col.p, d_M, d_w1, n, dpart.p);
  hipLaunchKernelGGL(k_wdot2, dim3((unsigned)nh), dim3(64), 0, cur_stream(), dpart.p, nblk, dres.p);
  OB_HIP(hipGetLastError());
  OB_TRY(launch_tmm_ge0(b, t, true, d_w2, dres.p + nh));
  for (size_t gi = 0; gi < t.ge_d3.size(); ++gi) OB_TRY(launch_tmm_d3(b, t.ge_d3[gi], 3, d_w1, d_w2, dres.p + goff[gi]));
  std::vector<double> host(total);
  OB_TRY(d2h(host.data(), dres.p, total * sizeof(double)));
  std::copy(host.begin(), host.begin() + nh, out_dot);
  std::copy(host.begin() + nh, host.begin() + nh + p * nh, out_sq);
  for (size_t gi = 0; gi < t.ge_d3.size(); ++gi) {
    const obhip_terms::GeD3 &grp = t.ge_d3[gi];
    const uint64_t vp = grp.v->p;
    const double *tmp = host.data() + goff[gi];
    for (size_t j = 0; j < grp.hyp0.size(); ++j)
      for (int hh = 0; hh < grp.nh; ++hh) {
        const uint64_t h = grp.hyp0[j] + hh;
        const std::vector<uint32_t> &ix = t.ge_sidx[h];
        const double *u1 = tmp + hh * vp + grp.off[j];
        const double *u2 = tmp + (grp.nh + hh) * vp + grp.off[j];
        long double s2 = 0;
        for (size_t q = 0; q < ix.size(); ++q) {
          s2 += (long double)a_host[ix[q]] * u1[q];
          out_sq[h * p + ix[q]] += 2.0 * u2[q];
        }
        out_dot[h] += (double)s2;
      }
  }
  return 0;
}

// out_host[c] = sum_i w_i G[i + c n], c < ncol (G: n x ncol column-major, device)
int grad_wdot_dev(const double *d_G, const double *d_w, uint64_t n, uint64_t ncol, double *out_host) {
  if (ncol == 0) return 0;
  constexpr int nblk = 256;
  DevBuf<double> dpart, dres;
  OB_TRY(dpart.alloc(ncol * nblk));
  OB_TRY(dres.alloc(ncol));
  hipLaunchKernelGGL(k_wdot1, dim3(nblk, (unsigned)ncol), dim3(256), 0, cur_stream(), d_G, d_w, n,
                     dpart.p);
  hipLaunchKernelGGL(k_wdot2, dim3((unsigned)ncol), dim3(64), 0, cur_stream(), dpart.p, nblk, dres.p);
  OB_HIP(hipGetLastError());
  return d2h(out_host, dres.p, ncol * sizeof(double));
}

// out_host (p x nhyp column-major) = sum_i a_i dB_ik/dhyp_h (squared: of B^2)
int grad_tmm_host(obhip_basis &b, obhip_terms &t, bool squared, const double *d_a, double *out_host) {
  OB_TRY(check_grad_args(&b, &t));
  OB_TRY(ensure_gradbasis(b));  // (tmm_gradhyp_all asks for the squared store where it needs it)
  return tmm_gradhyp_all(b, t, squared, d_a, out_host);
}

}  // namespace obhip
