// Device-side helpers shared by the HIP kernels of libobhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "obhip_internal.h"

namespace obhip {

// ---- 1-D covariance kernels -----------------------------------------------------
// k(h) = (1 + h + h^2/3) exp(-h)  (src/covfuncs.cpp:121-124,207-210,306-308).
//
// mat25 / mat25pow: h = |t(x) - t(knot_j)| with t(x) = x / expLS (mat25,
// covfuncs.cpp:114-120) or x^powv / expLS (mat25pow, :198-206).  exp(-h) is
// separable around the sign of t(x) - t_j, so a row needs two exponentials
// (exp(+u(x)), exp(-u(x))) instead of one per knot; the knot factors exp(+-u_j) are
// precomputed on the host (ModelDev::build).  u = t - c is centred on the middle of the
// dimension's knots (D.p2 = c), which keeps |u_j| at half the knot range: the dimension
// takes this path while max |u_j| < 150 (the product of the decaying pair then never
// meets inf * 0, wherever x lies: exp(-|u(x)|) underflows to an honest 0 far outside).
// Hyper-parameters so small that the knots themselves spread beyond that (the reference's
// updatehyp accepts them; only the hyper-prior penalises them) switch the dimension to
// the DIRECT kinds below: one exp(-h) per knot like the reference.
//
// mat25ang: h = sqrt((sin x/ls_s - sin k_j/ls_s)^2 + (cos x/ls_c - cos k_j/ls_c)^2)
// (covfuncs.cpp:285-305): not separable, one exp per knot; ka = sin k_j/ls_s,
// kb = cos k_j/ls_c.
constexpr int kCovMat25Direct = 3, kCovMat25PowDirect = 4;  // DimDesc.kind on the device only
template <int KIND>
__device__ __forceinline__ void kernel_pre(const DimDesc &D, double xv, double &a0, double &a1,
                                           double &a2) {
  if (KIND == OBHIP_COV_MAT25 || KIND == kCovMat25Direct) {
    a0 = xv / D.p0 - D.p2;
    a1 = KIND == OBHIP_COV_MAT25 ? exp(a0) : 0.0;
    a2 = KIND == OBHIP_COV_MAT25 ? exp(-a0) : 0.0;
  } else if (KIND == OBHIP_COV_MAT25POW || KIND == kCovMat25PowDirect) {
    a0 = pow(xv, D.p0) / D.p1 - D.p2;
    a1 = KIND == OBHIP_COV_MAT25POW ? exp(a0) : 0.0;
    a2 = KIND == OBHIP_COV_MAT25POW ? exp(-a0) : 0.0;
  } else {
    a0 = sin(xv) / D.p0;
    a1 = cos(xv) / D.p1;
    a2 = 0.0;
  }
}

template <int KIND>
__device__ __forceinline__ double kernel_value(double ka, double kb, double kc, double a0,
                                               double a1, double a2) {
  double h, eh;
  if (KIND == OBHIP_COV_MAT25ANG) {
    const double hs = a0 - ka, hc = a1 - kb;
    h = sqrt(hs * hs + hc * hc);
    eh = exp(-h);
  } else if (KIND == kCovMat25Direct || KIND == kCovMat25PowDirect) {
    h = fabs(a0 - ka);
    eh = exp(-h);
  } else {
    const double dlt = a0 - ka;
    h = fabs(dlt);
    eh = dlt >= 0.0 ? a2 * kb : a1 * kc;
  }
  return (1.0 + h + h * h * (1.0 / 3.0)) * eh;
}

// ---- one dimension of the basis for one row ----------------------------------------
// One 8-column chunk of (kernel row vector) x rotmat for one dimension.
// a0..a2 are the per-row precomputed values (kernel_pre).  D and every table
// index are wave-uniform, so the knot constants and the rotmat entries are
// scalar loads and the FMAs take them as SGPR operands.
template <int KIND>
__device__ __forceinline__ void dim_chunk(const DimDesc &D, const double *__restrict__ ka,
                                          const double *__restrict__ kb,
                                          const double *__restrict__ kc,
                                          const double *__restrict__ rot, double a0,
                                          double a1, double a2, int c0, double (&acc)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.0;
  const double *rp = rot + D.rotoff + c0;
  // unrolled so that the scalar loads of several knots are in flight together: one knot is
  // only ~17 VALU instructions, far less than a scalar-load round trip
#pragma unroll 4
  for (int j = 0; j < D.m; ++j) {
    double kv = kernel_value<KIND>(ka[D.koff + j], kb[D.koff + j], kc[D.koff + j], a0, a1, a2);
    const double *r = rp + (size_t)j * D.ncolp;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = fma(kv, r[c], acc[c]);
  }
}

// stores level t >= 1 of a dimension (compact column ccol) for this lane's row
struct StoreGlobal {
  double *base;  // bm + tile * Mc * 64 + lane
  __device__ __forceinline__ void operator()(int ccol, double v) const {
    base[(size_t)ccol * kTileRows] = v;
  }
};
struct StoreLds {
  double *lds;      // [used column][64]
  const int *cpos;  // compact column -> used column or -1
  int lane;
  __device__ __forceinline__ void operator()(int ccol, double v) const {
    const int u = cpos[ccol];
    if (u >= 0) lds[u * kTileRows + lane] = v;
  }
};

struct StoreLdsPitch {  // the same into a [used column][65] tile (term-per-lane kernels)
  double *lds;
  const int *cpos;
  int lane;
  __device__ __forceinline__ void operator()(int ccol, double v) const {
    const int u = cpos[ccol];
    if (u >= 0) lds[u * 65 + lane] = v;
  }
};

// R = cov(x, knots) . rotmat for one row and one dimension; levels >= 1 are
// divided by level 0 (modandbase.cpp:297) and stored, level 0 is returned.
template <int KIND, typename Store>
__device__ __forceinline__ double build_dim(const DimDesc &D, const double *ka, const double *kb,
                                            const double *kc, const double *rot, double xv,
                                            const Store &store) {
  double a0, a1, a2;
  kernel_pre<KIND>(D, xv, a0, a1, a2);
  double acc[8];
  double cl = 1.0;
  for (int c0 = 0; c0 < D.ncolp; c0 += 8) {
    dim_chunk<KIND>(D, ka, kb, kc, rot, a0, a1, a2, c0, acc);
    if (c0 == 0) cl = acc[0];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int col = c0 + c;
      if (col >= 1 && col < D.ncol) store(D.ccol0 + col - 1, acc[c] / cl);
    }
  }
  return cl;
}

// mat25 / mat25pow from the interval tables (ModelDev::build, core.cpp): J = number of knots with
// u_j <= u(x) by bisection on the sorted u, t = u(x) - u_(J-1), and per level six table entries:
//   R[c] = e^{-t} (A0 + t (A1 + t A2)) + e^{+t} (B0 + t (t B2 - B1)).
// O(1) per (row, level) where the knot loop spends ~17 instructions per (row, knot); two
// exponentials per (row, dimension) as before.  No knots below (J = 0): the e^{-t} part is empty
// and e^{-t} may overflow (t < 0), so it is dropped; likewise e^{+t} for J = m.
template <int KIND, typename Store>
__device__ __forceinline__ double build_dim_tab(const DimDesc &D, const double *__restrict__ tab,
                                                double xv, const Store &store) {
  typedef double dd2 __attribute__((ext_vector_type(2)));
  const double ux = (KIND == OBHIP_COV_MAT25 ? xv / D.p0 : pow(xv, D.p0) / D.p1) - D.p2;
  const double *__restrict__ us = tab + D.tab;
  int J;
  double uref;  // u_(J-1) (u_(0) for J = 0)
  if (D.gwin > 0) {
    // (nearly) equidistant knots: the host has checked that the guess is within gwin - 1 of J for
    // every u (ModelDev::build), so J = (knots below the window) + (knots of the window <= u):
    // 2 gwin INDEPENDENT reads instead of seven dependent ones.  (NaN: J0 = 0, J = 0, as the
    // bisection gives.)
    double q = floor((ux - D.g0) * D.ginv) + 1.0;
    q = fmin(fmax(q, 0.0), (double)D.m);
    const int J0 = (int)q, w0 = max(J0 - 2, 0);
    // always the window of gwin = 2 (J0 - 2 .. J0 + 1: a superset of gwin = 1's), its four values
    // kept: the knot below u(x), u_(J-1), is one of them (J0 - 1 <= J <= J0 + 1), so the reference
    // point costs selects instead of a second, dependent read
    double wv[4];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = J0 - 2 + k;
      const bool in = idx >= 0 && idx < D.m;
      wv[k] = us[min(max(idx, 0), D.m - 1)];
      cnt += in && wv[k] <= ux ? 1 : 0;
    }
    J = w0 + cnt;
    const int kr = max(J - 1, 0) - (J0 - 2);  // 0 .. 3
    uref = kr <= 1 ? (kr == 0 ? wv[0] : wv[1]) : (kr == 2 ? wv[2] : wv[3]);
  } else {
    int lo = 0, hi = D.m;  // u_(lo-1) <= ux < u_(hi)
    for (int it = 0; it < 7; ++it) {  // m <= 127
      const int mid = (lo + hi) >> 1;
      const bool open = lo < hi;
      const bool le = us[min(mid, D.m - 1)] <= ux;
      lo = open && le ? mid + 1 : lo;
      hi = open && !le ? mid : hi;
    }
    J = lo;
    uref = us[max(J - 1, 0)];
  }
  const double t = ux - uref;
  const double em = J == 0 ? 0.0 : exp(-t), ep = J == D.m ? 0.0 : exp(t);
  const dd2 *__restrict__ cf = (const dd2 *)(us + ((D.m + 1) & ~1) + (size_t)J * D.ncol * 6);
  double cl = 1.0, icl = 1.0;
  for (int c = 0; c < D.ncol; ++c) {
    const dd2 e0 = cf[3 * c], e1 = cf[3 * c + 1], e2 = cf[3 * c + 2];  // A0 A1 | A2 B0 | B1 B2
    const double r = em * fma(t, fma(t, e1.x, e0.y), e0.x) + ep * fma(t, fma(t, e2.y, -e2.x), e1.y);
    if (c == 0) {
      cl = r;
      icl = 1.0 / r;  // one division per (row, dimension); the levels are scaled by the reciprocal
    } else {
      store(D.ccol0 + c - 1, r * icl);
    }
  }
  return cl;
}

template <typename Store>
__device__ __forceinline__ double build_dim_any(const DimDesc &D, const double *ka, const double *kb,
                                                const double *kc, const double *rot,
                                                const double *tab, double xv, const Store &store) {
  if (D.tab >= 0) {
    if (D.kind == OBHIP_COV_MAT25) return build_dim_tab<OBHIP_COV_MAT25>(D, tab, xv, store);
    return build_dim_tab<OBHIP_COV_MAT25POW>(D, tab, xv, store);
  }
  if (D.kind == OBHIP_COV_MAT25) return build_dim<OBHIP_COV_MAT25>(D, ka, kb, kc, rot, xv, store);
  if (D.kind == OBHIP_COV_MAT25POW)
    return build_dim<OBHIP_COV_MAT25POW>(D, ka, kb, kc, rot, xv, store);
  if (D.kind == kCovMat25Direct) return build_dim<kCovMat25Direct>(D, ka, kb, kc, rot, xv, store);
  if (D.kind == kCovMat25PowDirect)
    return build_dim<kCovMat25PowDirect>(D, ka, kb, kc, rot, xv, store);
  return build_dim<OBHIP_COV_MAT25ANG>(D, ka, kb, kc, rot, xv, store);
}

// ---- term tables in registers ---------------------------------------------------------
// A wave works on 64 terms at a time: lane j holds the packed column list of
// term k0 + j (W2 dwords, two uint16 used-column indices each); inside an
// unrolled 64-term loop the entries are broadcast with v_readlane.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

template <int W2>
__device__ __forceinline__ void load_cw(uint32_t (&cw)[W2], const uint32_t *__restrict__ colsw,
                                        int k) {
#pragma unroll
  for (int w = 0; w < W2; ++w) cw[w] = colsw[(size_t)k * W2 + w];
}

// v * product of the staged columns ([column][64 rows] LDS tile) of the term held
// by lane t, for this lane's row
template <int W2>
__device__ __forceinline__ double term_prod_rl(const double *__restrict__ lds,
                                               const uint32_t (&cw)[W2], int t, int lane, double v) {
#pragma unroll
  for (int w = 0; w < W2; ++w) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cw[w], t);
    v *= lds[(c & 0xffffu) * 64 + lane];
    v *= lds[(c >> 16) * 64 + lane];
  }
  return v;
}

// generic fallback (more than 8 columns per term): column words from memory
__device__ __forceinline__ double term_prod_mem(const double *__restrict__ lds,
                                                const uint32_t *__restrict__ cw, int W2, int lane,
                                                double v) {
  for (int w = 0; w < W2; ++w) {
    const uint32_t c = cw[w];
    v *= lds[(c & 0xffffu) * 64 + lane];
    v *= lds[(c >> 16) * 64 + lane];
  }
  return v;
}

// ---- LDS tile of basemat ----------------------------------------------------------
// A staged tile holds Mu "used" columns of one 64-row tile: element (u, r) lives
// at u * 64 + (r ^ swz(u)).  The XOR swizzle keeps the two access patterns of
// the consumers conflict-free:
//   - lane = row, column wave-uniform (mm / tmm / getmat): r ^ const is a
//     bijection of the lanes onto the 64 slots of the column;
//   - 16 different columns x 4 consecutive rows (Gram MFMA operands): lanes
//     that share a row differ in u, and swz moves them to different banks
//     whenever their columns differ mod 16.
__device__ __forceinline__ int tile_swz(int u) { return (u & 15) << 1; }
__device__ __forceinline__ int tile_idx(int u, int r) { return u * kTileRows + (r ^ tile_swz(u)); }

// Cooperative copy of one 64-row tile (columns ucol[0..Mu)) from HBM to LDS.
// SQUARE stages the squared values (basematsq, modandbase.cpp:581).
template <bool SQUARE, bool SWIZZLE>
__device__ __forceinline__ void stage_tile(double *__restrict__ lds, const double *__restrict__ bm_tile,
                                           const uint32_t *__restrict__ ucol, int Mu, int tid,
                                           int nthreads) {
  const int total = Mu * kTileRows;
  for (int e = tid; e < total; e += nthreads) {
    const int u = e >> 6, r = e & 63;
    double v = bm_tile[(size_t)ucol[u] * kTileRows + r];
    if (SQUARE) v *= v;
    lds[SWIZZLE ? tile_idx(u, r) : e] = v;
  }
}


// ---- term-per-lane kernels: shared pieces ------------------------------------------------
// (k_tmm_tl, k_mm_tl, k_materialize_tl in kernels_prod.hip, k_predict_tl in
// kernels_predict.hip; the layout and the hand-issued reads are described at k_tmm_tl)
constexpr int kTlThreads = 512, kTlWaves = kTlThreads / 64, kTlGP = 2, kTlPitch = 65;
constexpr int kTlPre = 16;    // prefetch registers per thread => Mu <= 8 * 16
constexpr int kTlChunk = 16;  // rows per unrolled chunk (code size)

template <int OFF>
__device__ __forceinline__ double tl_rd(uint32_t addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// ---- the read pipeline of the term-per-lane kernels ---------------------------------------
// s_waitcnt lgkmcnt(KEEP) tied to the N registers b[S .. S+N) it releases
template <int KEEP, int N, int S>
__device__ __forceinline__ void tl_waitn(double (&b)[12]) {
  static_assert(N >= 1 && N <= 8 && S + N <= 12, "");
  if constexpr (N == 1)
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[S]) : "n"(KEEP) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b[S]), "+v"(b[S + 1]) : "n"(KEEP) : "memory");
  else if constexpr (N == 3)
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(b[S]), "+v"(b[S + 1]), "+v"(b[S + 2]) : "n"(KEEP) : "memory");
  else if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(%4)"
                 : "+v"(b[S]), "+v"(b[S + 1]), "+v"(b[S + 2]), "+v"(b[S + 3])
                 : "n"(KEEP)
                 : "memory");
  else if constexpr (N == 5)
    asm volatile("s_waitcnt lgkmcnt(%5)"
                 : "+v"(b[S]), "+v"(b[S + 1]), "+v"(b[S + 2]), "+v"(b[S + 3]), "+v"(b[S + 4])
                 : "n"(KEEP)
                 : "memory");
  else if constexpr (N == 6)
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(b[S]), "+v"(b[S + 1]), "+v"(b[S + 2]), "+v"(b[S + 3]), "+v"(b[S + 4]),
                   "+v"(b[S + 5])
                 : "n"(KEEP)
                 : "memory");
  else if constexpr (N == 7)
    asm volatile("s_waitcnt lgkmcnt(%7)"
                 : "+v"(b[S]), "+v"(b[S + 1]), "+v"(b[S + 2]), "+v"(b[S + 3]), "+v"(b[S + 4]),
                   "+v"(b[S + 5]), "+v"(b[S + 6])
                 : "n"(KEEP)
                 : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(b[S]), "+v"(b[S + 1]), "+v"(b[S + 2]), "+v"(b[S + 3]), "+v"(b[S + 4]),
                   "+v"(b[S + 5]), "+v"(b[S + 6]), "+v"(b[S + 7])
                 : "n"(KEEP)
                 : "memory");
}

// One chunk of ROWS tile rows for the NU terms ("units") a lane holds.  A unit's product is
// the last WE of its W column slots (the term tables are right-aligned, so a wave whose terms
// all have at most WE factors skips the leading ones); D = 12 / WE units are in flight (LDS
// returns in order, lgkmcnt counts to 15; INFLIGHT caps the reads -- and their registers).
// The context C provides
//   uint32_t ad[NU][W]                       LDS byte addresses of row 0 of the chunk
//   template <int RR> void row()             once per row, before its first unit
//   template <int RR, int UNIT> void use(v)  the product of unit UNIT at row RR
// TAIL > 0: the last TAIL factors of a unit are NOT multiplied in but handed over one by one,
//   template <int RR, int UNIT, int S, bool LEAD> void use_tail(v, buf)
// with v the product of the WE - TAIL leading factors (LEAD false: there are none, v = 1) and
// buf[S .. S + TAIL) the others (the gradient passes of kernels_grad.hip: a term's factor in one
// dimension and that dimension's delta columns)
template <int WE, int W, int NU, int ROWS, int INFLIGHT = 12, int U0 = 0, int ROW0 = 0, int TAIL = 0>
struct TlPipe {
  static_assert(WE >= 1 && WE <= W && W <= 8 && INFLIGHT <= 12 && TAIL <= WE, "");
  static constexpr int D = (INFLIGHT / WE) > 0 ? INFLIGHT / WE : 1;
  static constexpr int TOT = ROWS * NU;

  template <int U, typename C>
  static __device__ __forceinline__ void issue(C &c, double (&buf)[12]) {
    constexpr int rr = U / NU, unit = U % NU, s = (U % D) * WE;
#pragma unroll
    for (int j = 0; j < WE; ++j) buf[s + j] = tl_rd<(ROW0 + rr) * 8>(c.ad[U0 + unit][W - WE + j]);
  }
  template <int U, typename C>
  static __device__ __forceinline__ void steps(C &c, double (&buf)[12]) {
    if constexpr (U < TOT) {
      constexpr int rr = U / NU, unit = U % NU, s = (U % D) * WE;
      if constexpr (U + D - 1 < TOT) issue<U + D - 1>(c, buf);
      if constexpr (unit == 0) c.template row<rr>();
      constexpr int newer = (TOT - 1 - U) < (D - 1) ? (TOT - 1 - U) : (D - 1);
      tl_waitn<newer * WE, WE, s>(buf);
      if constexpr (TAIL == 0) {
        double v = buf[s];
#pragma unroll
        for (int j = 1; j < WE; ++j) v *= buf[s + j];
        c.template use<rr, U0 + unit>(v);
      } else if constexpr (WE == TAIL) {
        c.template use_tail<rr, U0 + unit, s, false>(1.0, buf);
      } else {
        double v = buf[s];
#pragma unroll
        for (int j = 1; j < WE - TAIL; ++j) v *= buf[s + j];
        c.template use_tail<rr, U0 + unit, s + WE - TAIL, true>(v, buf);
      }
      steps<U + 1>(c, buf);
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
    double buf[12];
    prologue<0>(c, buf);
    steps<0>(c, buf);
  }
};

// Run the pipeline instantiated for we column reads per term on units [U0, U0 + NUH) of a
// lane.  Variants: W, W-1, W-2, W-3.  k_tmm_tl runs the two halves of its units separately,
// each with the width its own terms need.
// ROW0: first row of the chunk relative to the addresses in c.ad (an immediate of the reads)
template <int W, int NUH, int ROWS, int INFLIGHT, int U0, int ROW0 = 0, typename C>
__device__ __forceinline__ void tl_run_half(C &c, int we) {
  if (we == W) {
    TlPipe<W, W, NUH, ROWS, INFLIGHT, U0, ROW0>::run(c);
  } else if (we == W - 1) {
    TlPipe<W - 1, W, NUH, ROWS, INFLIGHT, U0, ROW0>::run(c);
  } else if (W >= 3 && we == W - 2) {
    TlPipe<(W >= 3 ? W - 2 : 1), W, NUH, ROWS, INFLIGHT, U0, ROW0>::run(c);
  } else {
    TlPipe<(W >= 4 ? W - 3 : 1), W, NUH, ROWS, INFLIGHT, U0, ROW0>::run(c);
  }
}
// the same with TAIL factors handed over singly: variants W .. max(TAIL, W - 3)
template <int W, int NUH, int ROWS, int INFLIGHT, int U0, int TAIL, typename C>
__device__ __forceinline__ void tl_run_tail(C &c, int we) {
  if (we == W || W - 1 < TAIL) {
    TlPipe<W, W, NUH, ROWS, INFLIGHT, U0, 0, TAIL>::run(c);
  } else if (we == W - 1 || W - 2 < TAIL) {
    TlPipe<(W - 1 >= TAIL ? W - 1 : W), W, NUH, ROWS, INFLIGHT, U0, 0, TAIL>::run(c);
  } else if (we == W - 2 || W - 3 < TAIL) {
    TlPipe<(W - 2 >= TAIL ? W - 2 : W), W, NUH, ROWS, INFLIGHT, U0, 0, TAIL>::run(c);
  } else {
    TlPipe<(W - 3 >= TAIL ? W - 3 : W), W, NUH, ROWS, INFLIGHT, U0, 0, TAIL>::run(c);
  }
}
template <int W, int TAIL>
__device__ __forceinline__ int tl_variant_tail(int nzmax) {
  return max(nzmax, max(TAIL, W - 3));
}
template <int W>
__device__ __forceinline__ int tl_variant(int nzmax) {  // smallest variant that covers nzmax
  return max(nzmax, max(1, W - 3));
}

// ---- shared sub-products: the read pipeline of a star ---------------------------------------
// (host side and the idea: csrc/share.cpp.)  A star is four terms; shape (P, S): P shared column
// reads whose product q every term takes over, then S reads per term.  S = 1, P >= 1: a family of
// terms that differ in one factor (P + 4 reads for four terms); P = 0: four unrelated terms of up
// to S factors (the plain form, what TlPipe does).  The context C provides
//   uint32_t ad[]                            flat LDS byte addresses of row 0 of the chunk: the star
//                                            at A0: [P shared][term 0: S][term 1: S] ... in read order
//   template <int RR> void row()             once per row, before its first product
//   template <int RR, int UNIT> void use(v)  the product of term UNIT - U0 of the star at row RR
// The reads of ROWS rows form one sequence; K of them are in flight (a ring of K registers: read i
// lands in slot i mod K and is issued once read i - K has been used; LDS returns in order, so
// s_waitcnt lgkmcnt(newest - last) releases a group, tied by "+v" to the registers it releases).
template <int KEEP, int N, int I0, int RING>
__device__ __forceinline__ void tl_waitr(double (&b)[RING]) {
  static_assert(N >= 1 && N <= 8 && KEEP >= 0 && KEEP <= 15, "");
#define OB_RG(j) "+v"(b[(I0 + (j)) % RING])
  if constexpr (N == 1)
    asm volatile("s_waitcnt lgkmcnt(%1)" : OB_RG(0) : "n"(KEEP) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_waitcnt lgkmcnt(%2)" : OB_RG(0), OB_RG(1) : "n"(KEEP) : "memory");
  else if constexpr (N == 3)
    asm volatile("s_waitcnt lgkmcnt(%3)" : OB_RG(0), OB_RG(1), OB_RG(2) : "n"(KEEP) : "memory");
  else if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(%4)" : OB_RG(0), OB_RG(1), OB_RG(2), OB_RG(3) : "n"(KEEP) : "memory");
  else if constexpr (N == 5)
    asm volatile("s_waitcnt lgkmcnt(%5)" : OB_RG(0), OB_RG(1), OB_RG(2), OB_RG(3), OB_RG(4) : "n"(KEEP) : "memory");
  else if constexpr (N == 6)
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : OB_RG(0), OB_RG(1), OB_RG(2), OB_RG(3), OB_RG(4), OB_RG(5)
                 : "n"(KEEP)
                 : "memory");
  else if constexpr (N == 7)
    asm volatile("s_waitcnt lgkmcnt(%7)"
                 : OB_RG(0), OB_RG(1), OB_RG(2), OB_RG(3), OB_RG(4), OB_RG(5), OB_RG(6)
                 : "n"(KEEP)
                 : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : OB_RG(0), OB_RG(1), OB_RG(2), OB_RG(3), OB_RG(4), OB_RG(5), OB_RG(6), OB_RG(7)
                 : "n"(KEEP)
                 : "memory");
#undef OB_RG
}

// A context with `static constexpr bool kFactored = true` takes the products apart instead:
//   template <int RR> void prefix(q)            the product of the shared factors (1 for a plain star),
//                                               once per row, after row() and before the row's terms
//   template <int RR, int UNIT> void leaf(g)    term UNIT's own factor (plain: its whole product)
// -- a sum over a star's terms is then q * sum_u c_u g_u (P + 4 multiply-adds per row instead of
// P + 7), and a product with a row weight g_u * (q w).
template <typename C, typename = void>
struct tl_factored : std::false_type {};
template <typename C>
struct tl_factored<C, std::void_t<decltype(C::kFactored)>> : std::bool_constant<C::kFactored> {};

template <int P, int S, int ROWS, int K, int A0 = 0, int U0 = 0, int ROW0 = 0>
struct TlStar {
  static constexpr int G = 4;
  static constexpr int R = P + G * S;             // reads per row
  static constexpr int T = ROWS * R;
  static constexpr int E = (P > 0 ? 1 : 0) + G;   // groups per row: the shared part, the terms
  static constexpr int NE = ROWS * E;
  static_assert(S >= 1 && S <= 8 && P >= 0 && P <= 7 && K <= 16 && K >= P && K >= S, "");
  static constexpr int ev_size(int e) { return (P > 0 && e % E == 0) ? P : S; }
  static constexpr int ev_last(int e) {  // index of the last read of group e
    const int row = e / E, g = e % E;
    if (P > 0) return row * R + (g == 0 ? P : P + g * S) - 1;
    return row * R + (g + 1) * S - 1;
  }
  template <int I, int END, typename C>
  static __device__ __forceinline__ void issue(C &c, double (&buf)[K]) {
    if constexpr (I < END) {
      buf[I % K] = tl_rd<(ROW0 + I / R) * 8>(c.ad[A0 + I % R]);
      issue<I + 1, END>(c, buf);
    }
  }
  template <int EV, typename C>
  static __device__ __forceinline__ void steps(C &c, double (&buf)[K], double &q) {
    if constexpr (EV < NE) {
      constexpr int last = ev_last(EV), size = ev_size(EV), first = last - size + 1;
      constexpr int prev_last = EV == 0 ? -1 : ev_last(EV == 0 ? 0 : EV - 1);
      constexpr int newest = prev_last + K < T - 1 ? prev_last + K : T - 1;  // issued so far
      constexpr int row = EV / E, g = EV % E;
      if constexpr (g == 0) c.template row<row>();
      tl_waitr<newest - last, size, first % K, K>(buf);
      double v = buf[first % K];
#pragma unroll
      for (int j = 1; j < size; ++j) v *= buf[(first + j) % K];
      if constexpr (tl_factored<C>::value) {
        if constexpr (P > 0 && g == 0) {
          c.template prefix<row>(v);
        } else {
          if constexpr (P == 0 && g == 0) c.template prefix<row>(1.0);
          c.template leaf<row, U0 + (P > 0 ? g - 1 : g)>(v);
        }
      } else if constexpr (P > 0 && g == 0) {
        q = v;
      } else {
        if constexpr (P > 0) v *= q;
        c.template use<row, U0 + (P > 0 ? g - 1 : g)>(v);
      }
      constexpr int upto = last + K < T - 1 ? last + K : T - 1;  // the group's registers are free
      issue<newest + 1, upto + 1>(c, buf);
      steps<EV + 1>(c, buf, q);
    }
  }
  template <typename C>
  static __device__ __forceinline__ void run(C &c) {
    double buf[K];
    double q = 1.0;
    issue<0, (K < T ? K : T)>(c, buf);
    steps<0>(c, buf, q);
  }
};

// the instantiation for a star-wave's shape (P | S << 8, obhip_terms::sh_shape): stars with
// P = 1 .. W - 1 shared factors, plain stars of W or W - 2 slots per term.  Wave-uniform.
template <int W, int ROWS, int K, int A0 = 0, int U0 = 0, int ROW0 = 0, typename C>
__device__ __forceinline__ void tl_star_run(C &c, uint32_t shape) {
  const int P = (int)(shape & 0xffu);
  constexpr int KS = K < W ? W : K;  // (a plain star of W slots needs W registers at least)
  if ((shape >> 8) != 1u) {
    if constexpr (W >= 4) {
      if ((int)(shape >> 8) == W - 2) {
        TlStar<0, W - 2, ROWS, KS, A0, U0, ROW0>::run(c);
        return;
      }
    }
    TlStar<0, W, ROWS, KS, A0, U0, ROW0>::run(c);
  } else if (P <= 1) {
    TlStar<1, 1, ROWS, K, A0, U0, ROW0>::run(c);
  } else if (P == 2) {
    if constexpr (W > 2) TlStar<2, 1, ROWS, K, A0, U0, ROW0>::run(c);
  } else if (P == 3) {
    if constexpr (W > 3) TlStar<3, 1, ROWS, K, A0, U0, ROW0>::run(c);
  } else if (P == 4) {
    if constexpr (W > 4) TlStar<4, 1, ROWS, K, A0, U0, ROW0>::run(c);
  } else if (P == 5) {
    if constexpr (W > 5) TlStar<5, 1, ROWS, K, A0, U0, ROW0>::run(c);
  } else if (P == 6) {
    if constexpr (W > 6) TlStar<6, 1, ROWS, K, A0, U0, ROW0>::run(c);
  } else {
    if constexpr (W > 7) TlStar<7, 1, ROWS, K, A0, U0, ROW0>::run(c);
  }
}

// Slot (position in obhip_terms::sperm, terms by falling number of factors) of unit u of a
// lane: a wave owns NU * 64 consecutive slots.  (Giving every wave one run of long and one
// run of short terms, to even out the reads per wave, was measured and did not help.)
template <int NU>
__device__ __forceinline__ uint64_t tl_slot(uint64_t block, int wave, int u, int lane) {
  return ((block * 8 + wave) * NU + u) * 64 + lane;
}

// number of factors of a term from its packed column words (used column 0 = the ones)
template <int W2>
__device__ __forceinline__ int tl_nnz(const uint32_t (&cw)[W2]) {
  int nz = 0;
#pragma unroll
  for (int w = 0; w < W2; ++w) nz += ((cw[w] & 0xffffu) != 0) + ((cw[w] >> 16) != 0);
  return nz;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

// a + b where, afterwards, lanes 0-31 hold the sum of a over lanes (l, l + 32) and lanes
// 32-63 the sum of b over (l - 32, l)
__device__ __forceinline__ double swap32_sum(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a),
                                                   (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a),
                                                   (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// the same one level down: 16-lane rows 0 and 2 end with the sum of a over rows (0,1) and
// (2,3), rows 1 and 3 with the sum of b
__device__ __forceinline__ double swap16_sum(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a),
                                                   (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a),
                                                   (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// v + (v rotated right by ROR lanes within every row of 16 lanes).  The two v_mov_b32_dpp are
// written out: through __builtin_amdgcn_update_dpp the compiler first zeroes the destination
// (two more VALU instructions per call) although row_ror with full row / bank masks writes every
// lane.  s_nop 4: a DPP read of a VGPR needs two wait states behind the VALU write of it, and a
// DPP instruction five behind a VALU write of EXEC (v_cmpx) -- the compiler's hazard recognizer
// does not look inside inline asm, so the larger of the two is paid here (3 more cycles).
template <int ROR>
__device__ __forceinline__ double row16_ror_add(double v) {
  static_assert(ROR >= 1 && ROR <= 15, "row_ror:1 .. row_ror:15");
  int rl, rh;
  asm volatile("s_nop 4\n\t"
               "v_mov_b32_dpp %0, %2 row_ror:%4 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %1, %3 row_ror:%4 row_mask:0xf bank_mask:0xf"
               : "=&v"(rl), "=&v"(rh)
               : "v"(__double2loint(v)), "v"(__double2hiint(v)), "n"(ROR));
  return v + __hiloint2double(rh, rl);
}


}  // namespace obhip
