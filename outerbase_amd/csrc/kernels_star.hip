// The outer-product kernels on shared sub-products (k_star): B a, B^T a, their squared forms, and
// B^T (c_a B a + c_b y) in one pass over the basis.
//
// What it computes (one template, the operation a parameter):
//   OP_HESS    the Hessian product of the PCG, B^T (B p)        loglik_gauss::hessmult,
//                                                                src/lpdfs/loglik_gauss.cpp:137-145
//   OP_UPDATE  the gradient pass of its update(): yhat = B theta,
//              B^T (e^{-2 sigma} (y - yhat)), sum (yhat - y)^2   loglik_gauss.cpp:117-125
//              (lpdf::optcg, src/fit.cpp:71-85, calls both once per iteration)
//   OP_MM      out = B a        (SQ: B^2 a)                      prodmm_,  src/linalg.cpp:57-131
//   OP_TMM     out = B^T a      (SQ: (B^2)^T a;                  tprodmm_, src/linalg.cpp:286-355
//              DUAL: B^T a and (B^2)^T a2 in one pass: the PCG's cold start, e^{-2 sigma} B^T y and
//              the preconditioner's sqcolsums, loglik_gauss.cpp:125,154-157)
// with B[i,k] = basescale[i] * prod_{l: t_kl > 0} basemat[i, col(l, t_kl)].
//
// Shared sub-products.  The terms are grouped into stars (csrc/share.cpp: four terms that share all
// factors but one; P + 4 LDS column reads and P + 4 multiply-adds per row for four terms where the
// term-per-lane kernels of rounds 1-4 spend 4 (P + 1) reads), one star per lane, 16 waves of 64
// lanes per workgroup; a workgroup takes 16 star-waves (4096 terms) and a range of 64-row tiles,
// double-buffered in LDS as [column][65] by LDS-direct loads.  Per tile:
//   phase A (OP_HESS, OP_UPDATE, OP_MM):  s_r = sum_k a_k prod_k = q sum_u a_u g_u per star (q the
//            product of the shared factors, g_u a term's own factor), 8 rows of per-lane
//            accumulators, reduced over the 64 lanes once per 8 rows (permlane swaps + DPP) into
//            red[wave][row] -- no barrier inside the phase;
//   middle:  lane = (row, part).  tot_r = sum over the 16 waves (DPP within 16 lanes); the row
//            weight w_r = c_a s_r^2 tot_r + c_b s_r y_r (OP_UPDATE also writes yhat and the residual
//            sum; OP_MM writes out_r; OP_TMM takes w_r = a_r s_r as given);
//   phase B (all but OP_MM):  acc_k += prod_k w_r = g_u (q w_r), the weight of a row one v_readlane
//            pair -- no cross-lane traffic at all.
// k_hm2 (kernels_hm.hip) forms every product ONCE and keeps the 4 x 4 products of four rows in
// registers between the two uses: a cross-lane butterfly, an LDS exchange and a workgroup barrier
// every four rows, all 16 waves in step (read pipelines 0.99 ms, reductions
// 0.52 ms, additive).  With stars a product costs 1.5 LDS reads instead of 3 -- cheap enough to
// form twice: two barriers per tile instead of seventeen, and the waves run their phases decoupled.
//
// Left-over terms (share.cpp: no family with four free members; 1-3 % of a downward-closed set)
// would need a star-wave of plain stars: 16 reads per row where a family star-wave has 6, the one
// wave the other fifteen wait for (measured by skipping it: 9 % of the Hessian product at the
// headline terms, 18 % at d = 8 with six-factor terms).  They are multiplied out in the MIDDLE step
// instead, lane = (row, left-over term): a_k prod_k joins the row's sum before the 16-lane
// reduction, and prod_k w_r goes to an accumulator of the lane's own, summed over the rows when the
// kernel ends.
//
// Squared forms need no squared tile: (q g)^2 = q^2 g^2.
//
// Term sets: grouped into stars (obhip_terms::sh.ok) of up to 6 factors, at most 192 left-over
// terms, two tiles of the used columns plus ~20 KB in LDS; OP_HESS / OP_UPDATE: 9 to 16 family
// star-waves (all terms in one workgroup); OP_MM / OP_TMM: 9 or more (workgroups along the terms).
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

constexpr int kStWaves = 16;
constexpr int kStRedPitch = 65;  // red[wave][65]: the middle step reads 16 waves' partials of a row
constexpr int kStLeftMax = 192;  // left-over terms the middle step takes: 12 per lane of a row's 16
constexpr int kStNL = kStLeftMax / 16;
constexpr int kStLeftMaxDual = 128;  // DUAL keeps two accumulators per left-over term: 8 per lane
enum { OP_HESS = 0, OP_UPDATE = 1, OP_MM = 2, OP_TMM = 3 };

// one wave instruction pair: the 512 bytes of a basis column (64 rows) from g to LDS at l
// m0 is written here: on the clobber list so that the compiler never assumes a value of its own
// survives the statement (m0 is a reserved register, hence the diagnostic)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void st_dma_col(const char *g /* uniform */, uint32_t voff /* 4 lane */,
                                           uint32_t l /* uniform */) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\t"
               "global_load_lds_dword %0, %1 offset:256"
               :: "v"(voff), "s"(g), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int NA, bool SQ>
struct StA {  // phase A: 8 rows of sum_u a_u prod_u = q sum_u a_u g_u  (SQ: q^2 sum_u a_u g_u^2)
  static constexpr bool kFactored = true;
  uint32_t (&ad)[NA];
  const double (&av)[4];
  double acc[8];
  double q, t;
  template <int RR>
  __device__ __forceinline__ void row() {}
  template <int RR>
  __device__ __forceinline__ void prefix(double qv) {
    q = SQ ? qv * qv : qv;
  }
  template <int RR, int UNIT>
  __device__ __forceinline__ void leaf(double g) {
    const double gg = SQ ? g * g : g;
    t = UNIT == 0 ? gg * av[0] : fma(gg, av[UNIT], t);
    if constexpr (UNIT == 3) acc[RR] = fma(q, t, acc[RR]);
  }
};
template <int NA, bool SQ, bool DUAL>
struct StB {  // phase B: acc_u += prod_u w_r = g_u (q w_r)  (SQ: g_u^2 (q^2 w_r); DUAL: both)
  static constexpr bool kFactored = true;
  uint32_t (&ad)[NA];
  double (&acc)[4];
  double (&acc2)[DUAL ? 4 : 1];
  double vs, vs2;  // w of row = lane (vs2: of the squared products)
  double vr, vr2;  // w of the current row, wave-uniform
  double qw, qw2;
  int rc;
  template <int RR>
  __device__ __forceinline__ void row() {
    vr = readlane_f64(vs, rc + RR);
    if constexpr (DUAL) vr2 = readlane_f64(vs2, rc + RR);
  }
  template <int RR>
  __device__ __forceinline__ void prefix(double qv) {
    qw = (SQ ? qv * qv : qv) * vr;
    if constexpr (DUAL) qw2 = qv * qv * vr2;
  }
  template <int RR, int UNIT>
  __device__ __forceinline__ void leaf(double g) {
    acc[UNIT] = fma(SQ ? g * g : g, qw, acc[UNIT]);
    if constexpr (DUAL) acc2[UNIT] = fma(g * g, qw2, acc2[UNIT]);
  }
};

struct StArgs {
  const double *bm, *scale;
  const uint32_t *ucol;
  int Mu;
  uint64_t Mc;
  const uint32_t *shcols, *shterm, *shshape;  // star tables (obhip_terms::sh_*)
  int nswf;                                   // family star-waves: the first nswf of the tables
  const uint32_t *left_term, *left_colsw;     // left-over terms, their columns as packed pairs
  int nleft;
  const double *a;    // coefficients (p): OP_HESS, OP_UPDATE, OP_MM
  int p;
  const double *rw;   // rows: y (OP_UPDATE), a (OP_TMM)
  const double *rw2;  // OP_TMM DUAL: a2 (null: ones)
  double ca, cb;
  uint64_t n, n_pad, ntiles, tiles_per_split, p_pad;
  double *part;       // [gridDim.x][p_pad] partial B^T sums (all but OP_MM)
  double *part2;      // DUAL: of the squared products
  double *out;        // OP_MM: B a (n), or with several workgroups along the terms mpart
  double *mpart;      // OP_MM: [gridDim.y][n_pad] unscaled partial row sums (k_mm_tl_sum scales), or null
  double *yhat;       // OP_UPDATE (may be null)
  double *sspart;     // OP_UPDATE: [gridDim.x] sums of squared residuals (may be null)
  const double *stop0, *stop1;
};

template <int W2, int K, int OP, bool SQ, bool DUAL>
__global__ void __launch_bounds__(kStWaves * 64, 4)
k_star(const uint32_t *__restrict__ ucol /* = A.ucol: a kernel argument of its own, so that the loads
          of the prefetch loop are scalar loads (through the struct they became vector loads, each with
          an s_waitcnt vmcnt(0) that also waited for the LDS-direct loads before it: +11 %) */,
       const StArgs A) {
  constexpr bool PA = OP != OP_TMM, PB = OP != OP_MM, RO = OP == OP_UPDATE;
  constexpr int NL = DUAL ? kStLeftMaxDual / 16 : kStNL;  // left-over terms per lane of the middle step
  static_assert(!DUAL || (OP == OP_TMM && !SQ), "DUAL: B^T a and (B^2)^T a2");
  static_assert(!SQ || OP == OP_MM || OP == OP_TMM, "squared forms: mm, tmm");
  // a launch enqueued before the host has read the step's break conditions (the PCG loop of
  // api.cpp): nothing to do when the iteration it was meant for will not happen
  if (A.stop0 != nullptr && (*A.stop0 != 0.0 || *A.stop1 != 0.0)) return;
  extern __shared__ double lds[];
  constexpr int W = 2 * W2, NA = 4 * W, WAVES = kStWaves;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Mu = A.Mu;
  const int tile_doubles = Mu * kTlPitch;
  double *wts = lds + 2 * (size_t)tile_doubles;      // [2 buffers][4][64] row data (see land())
  double *red = wts + 2 * 256;                        // [WAVES][65] per-wave row sums of phase A
  double *wrow = red + WAVES * kStRedPitch;           // [64 | 64] the row weights of phase B (DUAL: both)
  double *ssw = wrow + 128;                           // [WAVES] residual sums (epilogue)
  double *la = ssw + WAVES;                           // [nleft] coefficients of the left-over terms
  uint32_t *lad = (uint32_t *)(la + kStLeftMax);      // [nleft][W] their columns' offsets in a tile (doubles)
  uint32_t *landed = lad + kStLeftMax * W;            // waves whose share of a prefetched tile is in LDS
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
  const uint32_t tile_bytes = (uint32_t)tile_doubles * 8u;
  const uint64_t t0 = (uint64_t)blockIdx.x * A.tiles_per_split;
  const uint64_t t1 = min(A.ntiles, t0 + A.tiles_per_split);
  const int p = A.p;
  const uint64_t n = A.n, p_pad = A.p_pad;
  // the left-over terms belong to the first workgroup along the terms
  const int nleft = blockIdx.y == 0 ? A.nleft : 0;

  // star sigma = (16 blockIdx.y + wave) * 64 + lane of obhip_terms::sh_*: 4 W packed column
  // indices in the read order of the star-wave's shape, four term indices
  const int sw = (int)blockIdx.y * WAVES + wave;
  const uint64_t sg = (uint64_t)sw * 64 + lane;
  const bool live = sw < A.nswf;  // (whole waves beyond the family stars: they only stage)
  uint32_t ad[NA];
  double av[4], acc[4], acc2[DUAL ? 4 : 1];
#pragma unroll
  for (int i = 0; i < NA / 2; ++i) {
    const uint32_t cw = live ? A.shcols[sg * (NA / 2) + i] : 0u;  // column 0 = ones
    ad[2 * i] = lds0 + (cw & 0xffffu) * (kTlPitch * 8);
    ad[2 * i + 1] = lds0 + (cw >> 16) * (kTlPitch * 8);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t kt = live ? A.shterm[sg * 4 + u] : 0xffffffffu;  // (an empty star's: no term)
    av[u] = PA && kt < (uint32_t)p ? A.a[kt] : 0.0;
    acc[u] = 0.0;
    if constexpr (DUAL) acc2[u] = 0.0;
  }
  // the left-over terms: coefficients and column offsets in LDS, an accumulator per lane and slot
  for (int j = threadIdx.x; j < nleft; j += WAVES * 64) {
    const uint32_t kt = A.left_term[j];
    la[j] = PA && kt < (uint32_t)p ? A.a[kt] : 0.0;
  }
  for (int i = threadIdx.x; i < nleft * W2; i += WAVES * 64) {
    const uint32_t cw = A.left_colsw[i];
    lad[2 * i] = (cw & 0xffffu) * kTlPitch;
    lad[2 * i + 1] = (cw >> 16) * kTlPitch;
  }
  double accl[PB ? NL : 1], accl2[DUAL ? NL : 1];
#pragma unroll
  for (int jj = 0; jj < (PB ? NL : 1); ++jj) accl[jj] = 0.0;
#pragma unroll
  for (int jj = 0; jj < (DUAL ? NL : 1); ++jj) accl2[jj] = 0.0;
  const uint32_t shape = live ? (uint32_t)__builtin_amdgcn_readfirstlane((int)A.shshape[sw]) : (1u | (1u << 8));
  for (int i = threadIdx.x; i < WAVES * kStRedPitch; i += WAVES * 64) red[i] = 0.0;  // absent waves: zero
  if (threadIdx.x == 0) *landed = 0u;

  // next tile -> the other buffer, by LDS-direct loads; the last wave also fetches the rows' scale
  // and row vector(s) (requested BEFORE the LDS-direct loads and only used in land(): the
  // compiler's own vmcnt bookkeeping does not see the inline-asm loads)
  double scn = 0.0, yn = 0.0, y2n = 0.0;
  auto prefetch = [&](uint64_t tile, int bsel) {
    if (wave == WAVES - 1) {
      const uint64_t row = tile * kTileRows + lane;
      scn = yn = 0.0;
      y2n = 1.0;
      if (row < n) {
        scn = A.scale[row];
        if (RO || OP == OP_TMM) yn = A.rw[row];
        if (DUAL && A.rw2 != nullptr) y2n = A.rw2[row];
      }
    }
    const char *tb = (const char *)(A.bm + tile * A.Mc * kTileRows);
    const uint32_t l0 = lds0 + (bsel ? tile_bytes : 0u);
    for (int u = wave; u < Mu; u += WAVES) {
      const uint32_t col = __builtin_amdgcn_readfirstlane(ucol[u]);
      const uint64_t ga = (uint64_t)(tb + (size_t)col * (kTileRows * 8));
      const uint64_t gu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
      st_dma_col((const char *)gu, (uint32_t)lane * 4u,
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(l0 + (uint32_t)u * (kTlPitch * 8))));
    }
  };
  // Tile hand-over.  Two barriers per tile are data dependencies (phase A -> middle -> phase B).
  // A third -- "the next tile has landed and nobody reads the other buffer any more", at the top of
  // a tile -- would make every wave wait for the slowest one's phase B.  Instead: the next tile is
  // requested right after the first barrier of a tile (every wave is past the previous tile, whose
  // buffer it overwrites), a wave reports its share as landed after the first 16 rows of its last
  // phase (s_waitcnt vmcnt(0), then one LDS add), and before the next tile a wave only waits until
  // all 16 have reported -- which, as a rule, they did long ago.
  auto land = [&](int bnext) {  // my share of the prefetched tile (and, last wave, its row data)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == WAVES - 1) {  // lane = row
      double *wn = wts + bnext * 256;
      if (OP == OP_HESS || OP == OP_UPDATE) {  // vA = c_a s^2, vB = c_b s y  ->  w = vA tot + vB
        wn[lane] = A.ca * scn * scn;
        if (RO) {
          wn[64 + lane] = A.cb * scn * yn;
          wn[128 + lane] = scn;
          wn[192 + lane] = yn;
        }
      } else if (OP == OP_MM) {  // out = tot * s (SQ: s^2)
        wn[lane] = SQ ? scn * scn : scn;
      } else {  // OP_TMM: b = basescale % a, linalg.cpp:305
        wn[lane] = yn * (SQ ? scn * scn : scn);
        if (DUAL) wn[64 + lane] = y2n * scn * scn;
      }
    }
    if (lane == 0) __hip_atomic_fetch_add(landed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  __syncthreads();  // red, la, lad and the counter are initialised -- before any wave reports a share
  if (t0 < t1) {
    prefetch(t0, 0);
    land(0);
  }
  double ssacc = 0.0;  // lanes (lane & 15) == 0: sum over their rows of (yhat - y)^2

  for (uint64_t tile = t0; tile < t1; ++tile) {
    const int bsel = (int)((tile - t0) & 1);
    double *wt = wts + bsel * 256;
    {  // every wave's share of this tile is in LDS (reported by land(): 16 per tile)
      // (bounded: every wave reports unconditionally, so the wait cannot last; should it ever, the
      // wave goes on after ~0.3 s and poisons its results instead of hanging the GPU)
      const uint32_t want = (uint32_t)(tile - t0 + 1) * WAVES;
      int spins = 0;
      while (__builtin_amdgcn_readfirstlane(
                 (int)__hip_atomic_load(landed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < (int)want &&
             ++spins < (1 << 22))
        __builtin_amdgcn_s_sleep(2);
      if (spins >= (1 << 22)) acc[0] = __builtin_nan("");
    }
    if (!PB) {
      // OP_MM has no phase B under which the next tile could land: it is requested here, behind a
      // barrier of its own (every wave is through the previous tile's middle step, which reads that
      // tile), and reported at the end of phase A
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (tile + 1 < t1) prefetch(tile + 1, bsel ^ 1);
    }

    // ---- phase A: per-wave row sums of sum_k a_k prod_k, 8 rows at a time --------------------
    if (PA && live) {
      StA<NA, SQ> ca_{ad, av, {}, 1.0, 0.0};
      double *redw = red + wave * kStRedPitch;
#pragma unroll 1
      for (int rc = 0; rc < kTileRows; rc += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) ca_.acc[r] = 0.0;
        tl_star_run<W, 8, K>(ca_, shape);
        const int32_t last = !PB ? -(kTileRows - 8) * 8 + (bsel ? -(int32_t)tile_bytes : (int32_t)tile_bytes)
                                 : -(kTileRows - 8) * 8;  // back to row 0 (OP_MM: of the other buffer)
        const int32_t step = rc + 8 < kTileRows ? 8 * 8 : last;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          ad[i] += (uint32_t)step;
          asm volatile("" : "+v"(ad[i]));
        }
        // 8 accumulators x 64 lanes -> 2 registers whose 16-lane row q holds tile row
        // rc + i + 2 q, then the sum over the 16 lanes of the row
        double s4[4], s2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) s4[i] = swap32_sum(ca_.acc[i], ca_.acc[i + 4]);
#pragma unroll
        for (int i = 0; i < 2; ++i) s2[i] = swap16_sum(s4[i], s4[i + 2]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double v = row16_ror_add<8>(s2[i]);
          v = row16_ror_add<4>(v);
          v = row16_ror_add<2>(v);
          v = row16_ror_add<1>(v);
          if ((lane & 15) == 0) redw[rc + i + 2 * (lane >> 4)] = v;
        }
      }
    }
    if (!PB && tile + 1 < t1) land(bsel ^ 1);
    // (s_barrier behind an lgkmcnt wait only: __syncthreads would also wait for the vector-memory
    // counter, i.e. at the second barrier for the tile requested a moment before)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's row sums are in red -- and every wave is past the previous tile
    asm volatile("" ::: "memory");
    if (PB && tile + 1 < t1) prefetch(tile + 1, bsel ^ 1);

    // ---- middle: lane = (row, part); wave w takes rows 4 w .. 4 w + 3 -------------------------
    // part < 16 indexes the waves' partial sums and the left-over terms part, part + 16, ...
    double *wr = wrow;
    {
      const int row = 4 * wave + (lane >> 4), prt = lane & 15;
      const double *tb = lds + (bsel ? tile_doubles : 0) + row;  // this row in the tile
      auto left_prod = [&](int j) {
        double v = 1.0;
#pragma unroll
        for (int e = 0; e < W; ++e) v *= tb[lad[j * W + e]];
        return v;
      };
      double t = 0.0;
      if (PA) {
        t = red[prt * kStRedPitch + row];
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) {
          const int j = jj * 16 + prt;
          if (jj * 16 < nleft && j < nleft) {
            const double v = left_prod(j);
            t = fma(la[j], SQ ? v * v : v, t);
          }
          if (jj % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // (three terms' reads in flight, not twelve)
        }
        t = row16_ror_add<8>(t);
        t = row16_ror_add<4>(t);
        t = row16_ror_add<2>(t);
        t = row16_ror_add<1>(t);
      }
      double wv = PA ? wt[row] * t : wt[row];  // (every lane of the row: its left-over terms want the weight)
      double wv2 = 0.0;
      if (RO) wv += wt[64 + row];
      if (DUAL) wv2 = wt[64 + row];
      if (PB) {
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) {
          const int j = jj * 16 + prt;
          if (jj * 16 < nleft && j < nleft) {
            const double v = left_prod(j);
            accl[jj] = fma(SQ ? v * v : v, wv, accl[jj]);
            if constexpr (DUAL) accl2[jj] = fma(v * v, wv2, accl2[jj]);
          }
          if (jj % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (prt == 0) {
        const uint64_t grow = tile * kTileRows + row;
        if (RO) {
          if (grow < n) {
            const double yh = wt[128 + row] * t;
            if (A.yhat != nullptr) A.yhat[grow] = yh;
            const double dlt = yh - wt[192 + row];
            ssacc = fma(dlt, dlt, ssacc);
          }
        }
        if (OP == OP_MM) {
          if (A.mpart != nullptr)
            A.mpart[(uint64_t)blockIdx.y * A.n_pad + grow] = t;  // scaled by k_mm_tl_sum
          else if (grow < n)
            A.out[grow] = wv;
        } else {
          wr[row] = wv;
          if (DUAL) wr[64 + row] = wv2;
        }
      }
    }
    if (!PB) continue;  // (OP_MM: the next tile's top barrier comes before anything here is overwritten)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the row weights are complete (and red may be overwritten)
    asm volatile("" ::: "memory");

    // ---- phase B: acc_k += prod_k w_r -----------------------------------------------------------
    // after the last rows: row 0 of the other buffer
    const int32_t step_last = -(kTileRows - 16) * 8 + (bsel ? -(int32_t)tile_bytes : (int32_t)tile_bytes);
    if (live) {
      StB<NA, SQ, DUAL> cb_{ad, acc, acc2, wr[lane], DUAL ? wr[64 + lane] : 0.0, 0.0, 0.0, 0.0, 0.0, 0};
#pragma unroll 1
      for (int rc = 0; rc < kTileRows; rc += 16) {
        cb_.rc = rc;
        tl_star_run<W, 16, K>(cb_, shape);
        if (rc == 0 && tile + 1 < t1) land(bsel ^ 1);
        const int32_t step = rc + 16 < kTileRows ? 16 * 8 : step_last;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          ad[i] += (uint32_t)step;
          asm volatile("" : "+v"(ad[i]));
        }
      }
    } else if (tile + 1 < t1) {
      land(bsel ^ 1);
    }
  }
  if (PB && live) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t kt = A.shterm[sg * 4 + u];
      if (kt != 0xffffffffu) {
        A.part[(uint64_t)blockIdx.x * p_pad + kt] = acc[u];
        if constexpr (DUAL) A.part2[(uint64_t)blockIdx.x * p_pad + kt] = acc2[u];
      }
    }
  }
  if (PB && nleft > 0) {
    // left-over term j = jj * 16 + part: its sum over the rows sits in the lanes (wave, row group,
    // part) of all waves -- over the row groups by shuffles, over the waves through LDS (the tile
    // buffers are free now)
    __syncthreads();
    double *lred = lds;  // [DUAL ? 2 : 1][WAVES][NL][16]
#pragma unroll
    for (int jj = 0; jj < NL; ++jj) {
      double v = accl[jj];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) lred[(wave * NL + jj) * 16 + lane] = v;
      if constexpr (DUAL) {
        double v2 = accl2[jj];
        v2 += __shfl_xor(v2, 16, 64);
        v2 += __shfl_xor(v2, 32, 64);
        if (lane < 16) lred[WAVES * NL * 16 + (wave * NL + jj) * 16 + lane] = v2;
      }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nleft; j += WAVES * 64) {
      double s = 0.0, s2 = 0.0;
      for (int w = 0; w < WAVES; ++w) {
        s += lred[(w * NL + j / 16) * 16 + (j & 15)];
        if (DUAL) s2 += lred[WAVES * NL * 16 + (w * NL + j / 16) * 16 + (j & 15)];
      }
      A.part[(uint64_t)blockIdx.x * p_pad + A.left_term[j]] = s;
      if (DUAL) A.part2[(uint64_t)blockIdx.x * p_pad + A.left_term[j]] = s2;
    }
  }
  if (RO && A.sspart != nullptr) {
    // the residual sums sit in lanes 0, 16, 32, 48 of every wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ssacc += __shfl_xor(ssacc, off, 64);
    if (lane == 0) ssw[wave] = ssacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int w = 0; w < WAVES; ++w) s += ssw[w];
      A.sspart[blockIdx.x] = s;
    }
  }
}

// ---- the fused predictor on shared sub-products ---------------------------------------------------
// predictor$update + $mean + $var of pred_gauss (src/lpdfs/loglik_gauss.cpp:214-227: a fresh
// outerbase at the new rows, modandbase.cpp:547, then prodmm_ on basemat and on basematsq,
// linalg.cpp:57-131) in one kernel, as k_predict_tl (kernels_predict.hip) -- the basis tile only
// ever exists in LDS -- with the contraction of k_star: per 64-row tile the 16 waves evaluate the
// dimensions (lane = row, wave w takes dimensions w, w + 16, ...) into the [column][65] tile, then
// phase A with the stars in the lanes (mean: q sum_u theta_u g_u; VAR: q^2 sum_u cv_u g_u^2 beside
// it) and the middle step (sum over the waves, the left-over terms, the row's basescale).
template <int NA, bool VAR>
struct StP {
  static constexpr bool kFactored = true;
  uint32_t (&ad)[NA];
  const double (&th)[4];
  const double (&cv)[VAR ? 4 : 1];
  double acc[8];
  double accv[VAR ? 8 : 1];
  double q, t, t2;
  template <int RR>
  __device__ __forceinline__ void row() {}
  template <int RR>
  __device__ __forceinline__ void prefix(double qv) {
    q = qv;
  }
  template <int RR, int UNIT>
  __device__ __forceinline__ void leaf(double g) {
    t = UNIT == 0 ? g * th[0] : fma(g, th[UNIT], t);
    if constexpr (VAR) {
      const double gg = g * g;
      t2 = UNIT == 0 ? gg * cv[0] : fma(gg, cv[UNIT], t2);
    }
    if constexpr (UNIT == 3) {
      acc[RR] = fma(q, t, acc[RR]);
      if constexpr (VAR) accv[RR] = fma(q * q, t2, accv[RR]);
    }
  }
};

// 8 accumulators x 64 lanes -> tile rows rc .. rc + 7 of redw[.]
__device__ __forceinline__ void st_reduce8(const double (&acc)[8], double *__restrict__ redw, int rc, int lane) {
  double s4[4], s2[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) s4[i] = swap32_sum(acc[i], acc[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; ++i) s2[i] = swap16_sum(s4[i], s4[i + 2]);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    double v = row16_ror_add<8>(s2[i]);
    v = row16_ror_add<4>(v);
    v = row16_ror_add<2>(v);
    v = row16_ror_add<1>(v);
    if ((lane & 15) == 0) redw[rc + i + 2 * (lane >> 4)] = v;
  }
}

template <int W2, int K, bool VAR, bool PFX>
__global__ void __launch_bounds__(kStWaves * 64, 4)
k_star_predict(const DimDesc *__restrict__ dims, const double *__restrict__ ka, const double *__restrict__ kb,
               const double *__restrict__ kc, const double *__restrict__ rot, const double *__restrict__ tab,
               const int *__restrict__ cpos, int d, int Mu, const uint32_t *__restrict__ shcols,
               const uint32_t *__restrict__ shterm, const uint32_t *__restrict__ shshape, int nswf,
               const uint32_t *__restrict__ left_term, const uint32_t *__restrict__ left_colsw, int nleft, int p,
               const double *__restrict__ theta, const double *__restrict__ coeffvar, double e2sigma,
               const double *__restrict__ x, uint64_t n, uint64_t ntiles, uint64_t tiles_per_split,
               double *__restrict__ mean, double *__restrict__ var) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2, NA = 4 * W, WAVES = kStWaves;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *red = lds + (size_t)2 * Mu * kTlPitch;     // (two tile buffers;) [WAVES][65] mean partials
  double *redv = red + WAVES * kStRedPitch;           // [WAVES][65] variance partials
  double *reds = redv + WAVES * kStRedPitch;          // [2 buffers][WAVES][65] basescale partials
  double *la = reds + 2 * WAVES * kStRedPitch;        // [nleft] theta of the left-over terms
  double *lcv = la + kStLeftMax;                      // [nleft] coeffvar
  uint32_t *lad = (uint32_t *)(lcv + kStLeftMax);     // [nleft][W] column offsets (doubles)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  const uint64_t sg = (uint64_t)wave * 64 + lane;
  const bool live = wave < nswf;
  uint32_t ad[NA];
  double th[4], cv[VAR ? 4 : 1];
#pragma unroll
  for (int i = 0; i < NA / 2; ++i) {
    const uint32_t cw = live ? shcols[sg * (NA / 2) + i] : 0u;
    ad[2 * i] = lds0 + (cw & 0xffffu) * (kTlPitch * 8);
    ad[2 * i + 1] = lds0 + (cw >> 16) * (kTlPitch * 8);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t kt = live ? shterm[sg * 4 + u] : 0xffffffffu;
    th[u] = kt < (uint32_t)p ? theta[kt] : 0.0;
    if constexpr (VAR) cv[u] = kt < (uint32_t)p ? coeffvar[kt] : 0.0;
  }
  for (int j = threadIdx.x; j < nleft; j += WAVES * 64) {
    const uint32_t kt = left_term[j];
    la[j] = kt < (uint32_t)p ? theta[kt] : 0.0;
    lcv[j] = VAR && kt < (uint32_t)p ? coeffvar[kt] : 0.0;
  }
  for (int i = threadIdx.x; i < nleft * W2; i += WAVES * 64) {
    const uint32_t cw = left_colsw[i];
    lad[2 * i] = (cw & 0xffffu) * kTlPitch;
    lad[2 * i + 1] = (cw >> 16) * kTlPitch;
  }
  const uint32_t shape = live ? (uint32_t)__builtin_amdgcn_readfirstlane((int)shshape[wave]) : (1u | (1u << 8));
  for (int i = threadIdx.x; i < 2 * WAVES * kStRedPitch; i += WAVES * 64) red[i] = 0.0;  // red, redv: absent waves

  // The basis of tile T + 1 is evaluated by the waves as they come out of phase A of tile T, into
  // the other tile buffer (first version: build -> barrier -> phase A -> barrier -> middle -> barrier,
  // the three segments strictly one after the other in all 16 waves: 1.32 ms at the headline terms
  // against 0.82 for the same contraction from a stored basis).  The dimensions go to the waves from
  // the LAST wave down -- the star-waves are sorted by falling number of shared factors, so the
  // waves that have the fewest reads per row in phase A take the second round of dimensions.
  const int tile_doubles = Mu * kTlPitch;
  const uint32_t tile_bytes = (uint32_t)tile_doubles * 8u;
  // PFX: the row's inputs for the wave's first two dimensions of tile T + 1 are requested BEFORE
  // phase A of tile T -- the build is a latency chain (input from HBM, interval search, table
  // coefficients), and its first link, the longest, then lies behind phase A
  double xn[2] = {0.5, 0.5};
  auto fetch_x = [&](uint64_t tile) {
    const uint64_t row = tile * kTileRows + lane;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int l = WAVES - 1 - wave + q * WAVES;
      xn[q] = (row < n && l < d) ? x[(uint64_t)l * n + row] : 0.5;
    }
  };
  auto build = [&](uint64_t tile, int bsel) {  // lane = row
    const StoreLdsPitch store{lds + (bsel ? tile_doubles : 0), cpos, lane};
    const uint64_t row = tile * kTileRows + lane;
    const bool valid = row < n;
    double sc = 1.0;
    int q = 0;
    for (int l = WAVES - 1 - wave; l < d; l += WAVES, ++q) {
      double xv;
      if (PFX && q < 2)
        xv = q == 0 ? xn[0] : xn[1];
      else
        xv = valid ? x[(uint64_t)l * n + row] : 0.5;
      const DimDesc D = dims[l];
      sc *= build_dim_any(D, ka, kb, kc, rot, tab, xv, store);
    }
    if (wave == 0) store.lds[lane] = 1.0;  // used column 0 = all ones
    reds[(bsel * WAVES + wave) * kStRedPitch + lane] = sc;
  };
  if (t0 < t1) {
    if (PFX) fetch_x(t0);
    build(t0, 0);
  }
  __syncthreads();
  for (uint64_t tile = t0; tile < t1; ++tile) {
    const int bsel = (int)((tile - t0) & 1);
    if (PFX && tile + 1 < t1) fetch_x(tile + 1);
    if (live) {
      StP<NA, VAR> c{ad, th, cv, {}, {}, 1.0, 0.0, 0.0};
#pragma unroll 1
      for (int rc = 0; rc < kTileRows; rc += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          c.acc[r] = 0.0;
          if constexpr (VAR) c.accv[r] = 0.0;
        }
        tl_star_run<W, 8, K>(c, shape);
        // next chunk; after the last rows: row 0 of the other buffer
        const int32_t last = -(kTileRows - 8) * 8 + (bsel ? -(int32_t)tile_bytes : (int32_t)tile_bytes);
        const int32_t step = rc + 8 < kTileRows ? 8 * 8 : last;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          ad[i] += (uint32_t)step;
          asm volatile("" : "+v"(ad[i]));
        }
        st_reduce8(c.acc, red + wave * kStRedPitch, rc, lane);
        if constexpr (VAR) st_reduce8(c.accv, redv + wave * kStRedPitch, rc, lane);
      }
    }
    if (tile + 1 < t1) build(tile + 1, bsel ^ 1);  // (its buffer: tile - 1's, whose middle step is behind the last barrier)
    __syncthreads();  // every wave's row sums are in red / redv, the next tile is built
    {  // middle: lane = (row, part); wave w takes rows 4 w .. 4 w + 3
      const int row = 4 * wave + (lane >> 4), prt = lane & 15;
      const double *tb = lds + (bsel ? tile_doubles : 0) + row;
      double t = red[prt * kStRedPitch + row];
      double tv = VAR ? redv[prt * kStRedPitch + row] : 0.0;
#pragma unroll
      for (int jj = 0; jj < kStNL; ++jj) {
        const int j = jj * 16 + prt;
        if (jj * 16 < nleft && j < nleft) {
          double v = 1.0;
#pragma unroll
          for (int e = 0; e < W; ++e) v *= tb[lad[j * W + e]];
          t = fma(la[j], v, t);
          if (VAR) tv = fma(lcv[j], v * v, tv);
        }
        if (jj % 3 == 2) __builtin_amdgcn_sched_barrier(0);
      }
      double s = reds[(bsel * WAVES + prt) * kStRedPitch + row];  // the 16 waves' basescale factors of the row
      t = row16_ror_add<8>(t);
      t = row16_ror_add<4>(t);
      t = row16_ror_add<2>(t);
      t = row16_ror_add<1>(t);
      if (VAR) {
        tv = row16_ror_add<8>(tv);
        tv = row16_ror_add<4>(tv);
        tv = row16_ror_add<2>(tv);
        tv = row16_ror_add<1>(tv);
      }
      // product over the 16 lanes of the row (waves beyond the dimensions hold 1)
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) s *= __shfl_xor(s, off, 64);
      const uint64_t grow = tile * kTileRows + row;
      if (prt == 0 && grow < n) {
        mean[grow] = t * s;
        if (VAR) var[grow] = tv * (s * s) + e2sigma;  // loglik_gauss.cpp:224-225
      }
    }
    __syncthreads();  // the middle step is through with red / redv and with this tile's buffer
  }
}

}  // namespace

size_t star_lds_bytes(const obhip_terms &t) {
  return ((size_t)2 * t.Mu * kTlPitch + 2 * 256 + kStWaves * kStRedPitch + 128 + kStWaves + kStLeftMax) *
             sizeof(double) +
         ((size_t)kStLeftMax * t.W + 4) * sizeof(uint32_t);
}

// the term sets k_star takes: grouped into stars, at least 9 family star-waves (below that the
// kernels with one or two terms per lane keep all 16 waves busy), few left-over terms, two tiles in
// LDS (the epilogue's [2][16][12][16] sums of the left-over terms go through the tile buffers);
// one_block: all terms in one workgroup (the Hessian products)
bool star_supports(const obhip_terms &t, bool one_block, bool dual) {
  const int w2 = (int)(t.W / 2);
  return share_wanted() && t.sh.ok && w2 >= 1 && w2 <= 3 && t.sh.nsw_family >= 9 &&
         (!one_block || t.sh.nsw_family <= 16) && t.sh.nleft <= (uint64_t)(dual ? kStLeftMaxDual : kStLeftMax) &&
         star_lds_bytes(t) <= (size_t)156 * 1024 &&
         (size_t)2 * t.Mu * kTlPitch >= (size_t)2 * kStWaves * kStNL * 16;
}

namespace {
template <int W2, int K, int OP, bool SQ, bool DUAL>
int run_star(const StArgs &A, dim3 grid, size_t lds) {
  OB_TRY(ensure_dyn_lds((const void *)k_star<W2, K, OP, SQ, DUAL>, lds));
  hipLaunchKernelGGL((k_star<W2, K, OP, SQ, DUAL>), grid, dim3(kStWaves * 64), lds, cur_stream(), A.ucol, A);
  OB_HIP(hipGetLastError());
  return 0;
}
template <int OP, bool SQ, bool DUAL>
int dispatch_star(const obhip_terms &t, const StArgs &A, dim3 grid) {
  const size_t lds = star_lds_bytes(t);
  switch ((int)(t.W / 2)) {
    case 1: return run_star<1, 12, OP, SQ, DUAL>(A, grid, lds);
    // (DUAL: two accumulators per term and per left-over term -- fewer reads in flight instead of spills)
    case 2: return run_star<2, DUAL ? 8 : 12, OP, SQ, DUAL>(A, grid, lds);
    default: return run_star<3, DUAL ? 6 : 10, OP, SQ, DUAL>(A, grid, lds);  // (terms of 7 and 8 factors: not taken)
  }
}
StArgs star_args(const obhip_basis &b, const obhip_terms &t, uint64_t ntiles, uint64_t tps) {
  StArgs A{};
  A.bm = b.bm.p;
  A.scale = b.scale.p;
  A.ucol = t.ucol.p;
  A.Mu = (int)t.Mu;
  A.Mc = b.md.Mc;
  A.shcols = (const uint32_t *)t.sh_cols.p;
  A.shterm = t.sh_term.p;
  A.shshape = t.sh_shape.p;
  A.nswf = (int)t.sh.nsw_family;
  A.left_term = t.sh_left_term.p;
  A.left_colsw = (const uint32_t *)t.sh_left_cols.p;
  A.nleft = (int)t.sh.nleft;
  A.p = (int)t.p;
  A.n = b.n;
  A.n_pad = b.n_pad;
  A.ntiles = ntiles;
  A.tiles_per_split = tps;
  A.p_pad = t.p_pad;
  return A;
}
}  // namespace

size_t star_predict_lds_bytes(const obhip_terms &t) {
  return ((size_t)2 * t.Mu * kTlPitch + 4 * kStWaves * kStRedPitch + 2 * kStLeftMax) * sizeof(double) +
         (size_t)kStLeftMax * t.W * sizeof(uint32_t);
}
// the fused predictor takes: all family stars in one workgroup (9 .. 16 star-waves)
bool star_predict_supports(const obhip_terms &t) {
  const int w2 = (int)(t.W / 2);
  return share_wanted() && t.sh.ok && w2 >= 1 && w2 <= 3 && t.sh.nsw_family >= 9 && t.sh.nsw_family <= 16 &&
         t.sh.nleft <= (uint64_t)kStLeftMax && star_predict_lds_bytes(t) <= (size_t)156 * 1024;
}
int launch_star_predict(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x, uint64_t n,
                        double *d_mean, const double *d_coeffvar, double e2sigma, double *d_var) {
  const size_t lds = star_predict_lds_bytes(t);
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t ntiles = (n + kTileRows - 1) / kTileRows;
  uint64_t nsplit = std::min<uint64_t>(ntiles, (uint64_t)device_cus(dev));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  const bool wv = d_coeffvar != nullptr && d_var != nullptr;
  // (with the variance the kernel spills more for the two registers than the early request gains:
  // 1.83 -> 1.88 ms; without, 1.35 -> 1.24 ms on the same box)
  static const bool pfx_on = !(getenv("OBHIP_PREDICT_PFX") && atoi(getenv("OBHIP_PREDICT_PFX")) == 0);
  const bool pfx = pfx_on && !wv;
#define OB_SP(W2_, K_, VAR_)                                                                                  \
  do {                                                                                                        \
    if (pfx)                                                                                                  \
      OB_SP2(W2_, K_, VAR_, true);                                                                            \
    else                                                                                                      \
      OB_SP2(W2_, K_, VAR_, false);                                                                           \
  } while (0)
#define OB_SP2(W2_, K_, VAR_, PFX_)                                                                           \
  do {                                                                                                        \
    OB_TRY(ensure_dyn_lds((const void *)k_star_predict<W2_, K_, VAR_, PFX_>, lds));                           \
    hipLaunchKernelGGL((k_star_predict<W2_, K_, VAR_, PFX_>), dim3((unsigned)nsplit), dim3(kStWaves * 64), lds, \
                       cur_stream(), t.pred_md.dims.p, t.pred_md.ka.p, t.pred_md.kb.p, t.pred_md.kc.p,        \
                       t.pred_md.rot.p, t.pred_md.tab.p, t.cpos.p, (int)m.d, (int)t.Mu,                       \
                       (const uint32_t *)t.sh_cols.p, t.sh_term.p, t.sh_shape.p, (int)t.sh.nsw_family,        \
                       t.sh_left_term.p, (const uint32_t *)t.sh_left_cols.p, (int)t.sh.nleft, (int)t.p,       \
                       d_theta, d_coeffvar, e2sigma, d_x, n, ntiles, tps, d_mean, d_var);                     \
  } while (0)
  switch ((int)(t.W / 2) * 2 + (wv ? 1 : 0)) {
    case 2: OB_SP(1, 12, false); break;
    case 3: OB_SP(1, 12, true); break;
    case 4: OB_SP(2, 12, false); break;
    case 5: OB_SP(2, 10, true); break;  // (ring depth 8 and / or early inputs here: 1.79-1.86 ms either way)
    case 6: OB_SP(3, 10, false); break;
    default: OB_SP(3, 8, true); break;
  }
#undef OB_SP
#undef OB_SP2
  OB_HIP(hipGetLastError());
  return 0;
}

// B^T (c_a B a + c_b y): d_y null = the Hessian product
int launch_star_hess(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca,
                     double cb, double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles,
                     uint64_t tps, const double *stop0, const double *stop1) {
  StArgs A = star_args(b, t, ntiles, tps);
  A.a = d_a;
  A.rw = d_y;
  A.ca = ca;
  A.cb = cb;
  A.part = part;
  A.yhat = d_yhat;
  A.sspart = sspart;
  A.stop0 = stop0;
  A.stop1 = stop1;
  if (d_y != nullptr) return dispatch_star<OP_UPDATE, false, false>(t, A, dim3(nsplit));
  return dispatch_star<OP_HESS, false, false>(t, A, dim3(nsplit));
}

// part[nsplit][p_pad] = partial B^T a (squared: (B^2)^T a); part2 (may be null) = partial (B^2)^T a2
int launch_star_tmm(const obhip_basis &b, obhip_terms &t, const double *d_a, bool squared, double *part,
                    const double *d_a2, double *part2, unsigned nsplit, uint64_t ntiles, uint64_t tps) {
  StArgs A = star_args(b, t, ntiles, tps);
  A.rw = d_a;
  A.rw2 = d_a2;
  A.part = part;
  A.part2 = part2;
  const dim3 grid(nsplit, (unsigned)((t.sh.nsw_family + kStWaves - 1) / kStWaves));
  if (part2 != nullptr) return dispatch_star<OP_TMM, false, true>(t, A, grid);
  if (squared) return dispatch_star<OP_TMM, true, false>(t, A, grid);
  return dispatch_star<OP_TMM, false, false>(t, A, grid);
}

// d_out (n) = B a (squared: B^2 a); mpart ([blocks along the terms][n_pad]): unscaled partial row
// sums when the terms take more than one workgroup (the caller sums and scales: k_mm_tl_sum)
int launch_star_mm(const obhip_basis &b, obhip_terms &t, const double *d_a, bool squared, double *d_out,
                   double *mpart, unsigned nsplit, uint64_t ntiles, uint64_t tps) {
  StArgs A = star_args(b, t, ntiles, tps);
  A.a = d_a;
  A.out = d_out;
  A.mpart = mpart;
  const dim3 grid(nsplit, (unsigned)((t.sh.nsw_family + kStWaves - 1) / kStWaves));
  if (squared) return dispatch_star<OP_MM, true, false>(t, A, grid);
  return dispatch_star<OP_MM, false, false>(t, A, grid);
}

}  // namespace obhip
