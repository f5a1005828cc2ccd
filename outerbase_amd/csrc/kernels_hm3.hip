// B^T (c_a B a + c_b y) in one pass over the basis, third generation (k_hm3): two phases per tile
// on shared sub-products.
//
// What it computes: the Hessian product of the PCG (loglik_gauss::hessmult,
// src/lpdfs/loglik_gauss.cpp:137-145: B^T (B p)) and the gradient pass of its update()
// (loglik_gauss.cpp:117-125: yhat = B theta, B^T (e^{-2 sigma} (y - yhat))), which lpdf::optcg
// (src/fit.cpp:71-85) calls once per iteration each -- as k_hm2 (kernels_hm.hip) and k_hm_tl
// (kernels_prod.hip) do.
//
// Why a third kernel.  k_hm2 forms every term product ONCE per row and keeps the 4 x 4 products of
// a sub-chunk in registers between "s_r = sum_k a_k prod_k" and "acc_k += prod_k w_r"; the price is
// a cross-lane butterfly, an LDS exchange and a workgroup barrier every FOUR rows, in which all 16
// waves stand in step (read pipelines 0.99 ms, reduction phases 0.52 ms,
// simply additive).  With the terms grouped into stars (csrc/share.cpp: four terms that share all
// factors but one; P + 4 column reads and P + 3 multiplies for four terms) a term product costs
// 1.6 LDS reads instead of 3 -- cheap enough to form it TWICE, and then nothing has to be kept:
//   phase A (the tile's 64 rows, B a):   s_r += a_k prod_k with 8 rows of per-lane accumulators,
//            reduced over the 64 lanes once per 8 rows (permlane swaps + DPP, as k_mm_tl) into
//            red[wave][row] -- no barrier inside the phase;
//   middle:  tot_r = sum over the waves, w_r = c_a s_r^2 tot_r + c_b s_r y_r (16 waves x 4 rows,
//            lane = (row, wave partial)); the update() form also writes yhat and the residual sum;
//   phase B (the 64 rows again, B^T w):  acc_k += prod_k w_r, the weight of a row one v_readlane
//            pair (as k_tmm_tl) -- no cross-lane traffic at all.
// Two workgroup barriers per tile instead of seventeen (the hand-over from one tile to the next is
// a counter of landed shares, not a barrier: see the tile loop), ~60 VGPRs less than k_hm2 (no products
// kept), so 12 reads in flight per wave; the waves run their phases decoupled, LDS reads of one
// under the reductions of another.  The tile is double-buffered by LDS-direct loads as in k_hm2.
//
// The terms that found no family (share.cpp: "left over", 1-3 % of a downward-closed set) would
// need a star-wave of plain stars -- 16 reads per row where a family star-wave has 6, the one wave
// the other fifteen wait for at every barrier (measured by skipping it: 9 % of the kernel at the
// headline terms, 18 % at d = 8 with six-factor terms).  They are multiplied out in the MIDDLE step
// instead, lane = (row, left-over term): the lane adds a_k prod_k to its row's sum before the
// 16-lane reduction, and, once the row weight is known, prod_k w_r to an accumulator of its own
// (summed over the rows when the kernel ends) -- a product per (left-over term, row) where the
// plain star-wave formed two and dragged 212 proper family terms along.
//
// Terms: those obhip_terms::prepare grouped into stars (t.sh.ok) of up to 6 factors, 9 to 16
// family star-waves (one star per lane: 2049 .. 4096 terms as a rule), at most 192 left-over terms,
// two tiles of the used columns plus ~20 KB in LDS.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

constexpr int kHm3Waves = 16;
constexpr int kHm3RedPitch = 65;  // red[wave][65]: the middle step reads 16 waves' partials of a row

// one wave instruction pair: the 512 bytes of a basis column (64 rows) from g to LDS at l
// m0 is written here: on the clobber list so that the compiler never assumes a value of its own
// survives the statement (m0 is a reserved register, hence the diagnostic)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void hm3_dma_col(const char *g /* uniform */, uint32_t voff /* 4 lane */,
                                            uint32_t l /* uniform */) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\t"
               "global_load_lds_dword %0, %1 offset:256"
               :: "v"(voff), "s"(g), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int NA>
struct Hm3A {  // phase A: 8 rows of sum_u a_u prod_u = q sum_u a_u g_u
  static constexpr bool kFactored = true;
  uint32_t (&ad)[NA];
  const double (&av)[4];
  double acc[8];
  double q, t;
  template <int RR>
  __device__ __forceinline__ void row() {}
  template <int RR>
  __device__ __forceinline__ void prefix(double qv) {
    q = qv;
  }
  template <int RR, int UNIT>
  __device__ __forceinline__ void leaf(double g) {
    t = UNIT == 0 ? g * av[0] : fma(g, av[UNIT], t);
    if constexpr (UNIT == 3) acc[RR] = fma(q, t, acc[RR]);
  }
};
template <int NA>
struct Hm3B {  // phase B: acc_u += prod_u w_r = g_u (q w_r)
  static constexpr bool kFactored = true;
  uint32_t (&ad)[NA];
  double (&acc)[4];
  double vs;  // w of row = lane
  double vr;  // w of the current row, wave-uniform
  double qw;
  int rc;
  template <int RR>
  __device__ __forceinline__ void row() {
    vr = readlane_f64(vs, rc + RR);
  }
  template <int RR>
  __device__ __forceinline__ void prefix(double qv) {
    qw = qv * vr;
  }
  template <int RR, int UNIT>
  __device__ __forceinline__ void leaf(double g) {
    acc[UNIT] = fma(g, qw, acc[UNIT]);
  }
};

constexpr int kHm3LeftMax = 192;  // left-over terms the middle step takes: 12 per lane of a row's 16
constexpr int kHm3NL = kHm3LeftMax / 16;

// RO: the update() form (y, yhat, sum of squared residuals); without it the Hessian product
template <int W2, int K, bool RO>
__global__ void __launch_bounds__(kHm3Waves * 64, 4)
k_hm3(const double *__restrict__ bm, const double *__restrict__ scale, const uint32_t *__restrict__ ucol,
      int Mu, uint64_t Mc, const uint32_t *__restrict__ shcols, const uint32_t *__restrict__ shterm,
      const uint32_t *__restrict__ shshape, const double *__restrict__ a, int p,
      const double *__restrict__ y, double ca, double cb, uint64_t n, uint64_t ntiles,
      uint64_t tiles_per_split, uint64_t p_pad, double *__restrict__ part, double *__restrict__ yhat,
      double *__restrict__ sspart, const double *__restrict__ stop0, const double *__restrict__ stop1,
      int nswf /* family star-waves: the first nswf of the tables */, const uint32_t *__restrict__ left_term,
      const uint32_t *__restrict__ left_colsw /* nleft x W2 packed pairs */, int nleft) {
  // a launch enqueued before the host has read the step's break conditions (the PCG loop of
  // api.cpp): nothing to do when the iteration it was meant for will not happen
  if (stop0 != nullptr && (*stop0 != 0.0 || *stop1 != 0.0)) return;
  extern __shared__ double lds[];
  constexpr int W = 2 * W2, NA = 4 * W, WAVES = kHm3Waves;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile_doubles = Mu * kTlPitch;
  double *wts = lds + 2 * (size_t)tile_doubles;      // [2 buffers][vA 64 | vB 64 | s 64 | y 64]
  double *red = wts + 2 * 256;                        // [WAVES][65] per-wave row sums of phase A
  double *wrow = red + WAVES * kHm3RedPitch;          // [64] the row weights of phase B
  double *ssw = wrow + 64;                            // [WAVES] residual sums (epilogue)
  double *la = ssw + WAVES;                           // [nleft] coefficients of the left-over terms
  uint32_t *lad = (uint32_t *)(la + kHm3LeftMax);     // [nleft][W] their columns' offsets in a tile (doubles)
  uint32_t *landed = lad + kHm3LeftMax * W;           // waves whose share of a prefetched tile is in LDS
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
  const uint32_t tile_bytes = (uint32_t)tile_doubles * 8u;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  // star sigma = wave * 64 + lane of obhip_terms::sh_*: 4 W packed column indices in the read
  // order of the star-wave's shape, four term indices
  const uint64_t sg = (uint64_t)wave * 64 + lane;
  const bool live = wave < nswf;  // (whole waves beyond the family stars: they only stage)
  const bool ok = live;
  uint32_t ad[NA];
  double av[4], acc[4];
#pragma unroll
  for (int i = 0; i < NA / 2; ++i) {
    const uint32_t cw = ok ? shcols[sg * (NA / 2) + i] : 0u;  // column 0 = ones
    ad[2 * i] = lds0 + (cw & 0xffffu) * (kTlPitch * 8);
    ad[2 * i + 1] = lds0 + (cw >> 16) * (kTlPitch * 8);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t kt = ok ? shterm[sg * 4 + u] : 0xffffffffu;  // (an empty star's: no term)
    av[u] = kt < (uint32_t)p ? a[kt] : 0.0;
    acc[u] = 0.0;
  }
  // the left-over terms: coefficients and column offsets in LDS, an accumulator per lane and slot
  for (int j = threadIdx.x; j < nleft; j += WAVES * 64) {
    const uint32_t kt = left_term[j];
    la[j] = kt < (uint32_t)p ? a[kt] : 0.0;
  }
  for (int i = threadIdx.x; i < nleft * W2; i += WAVES * 64) {
    const uint32_t cw = left_colsw[i];
    lad[2 * i] = (cw & 0xffffu) * kTlPitch;
    lad[2 * i + 1] = (cw >> 16) * kTlPitch;
  }
  double accl[kHm3NL];
#pragma unroll
  for (int jj = 0; jj < kHm3NL; ++jj) accl[jj] = 0.0;
  const uint32_t shape = live ? (uint32_t)__builtin_amdgcn_readfirstlane((int)shshape[wave]) : (1u | (1u << 8));
  for (int i = threadIdx.x; i < WAVES * kHm3RedPitch; i += WAVES * 64) red[i] = 0.0;  // absent waves: zero
  if (threadIdx.x == 0) *landed = 0u;

  // next tile -> the other buffer, by LDS-direct loads; the last wave also fetches the rows' scale
  // and y (requested BEFORE the LDS-direct loads and only used at the top of the next tile: the
  // compiler's own vmcnt bookkeeping does not see the inline-asm loads)
  double scn = 0.0, yn = 0.0;
  auto prefetch = [&](uint64_t tile, int bsel) {
    if (wave == WAVES - 1) {
      const uint64_t row = tile * kTileRows + lane;
      scn = yn = 0.0;
      if (row < n) {
        scn = scale[row];
        if (RO) yn = y[row];
      }
    }
    const char *tb = (const char *)(bm + tile * Mc * kTileRows);
    const uint32_t l0 = lds0 + (bsel ? tile_bytes : 0u);
    for (int u = wave; u < Mu; u += WAVES) {
      const uint32_t col = __builtin_amdgcn_readfirstlane(ucol[u]);
      const uint64_t ga = (uint64_t)(tb + (size_t)col * (kTileRows * 8));
      const uint64_t gu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
      hm3_dma_col((const char *)gu, (uint32_t)lane * 4u,
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(l0 + (uint32_t)u * (kTlPitch * 8))));
    }
  };
  // Tile hand-over.  Two barriers per tile are data dependencies (phase A -> middle -> phase B);
  // the third of the first version -- "the next tile has landed and nobody reads the other buffer
  // any more", at the top of a tile -- made every wave wait for the slowest one's phase B.  Now:
  // the next tile is requested right after the first barrier of a tile (every wave is past the
  // previous tile, whose buffer it overwrites), a wave reports its share as landed after the first
  // 16 rows of its phase B (s_waitcnt vmcnt(0), then one LDS add), and before phase A of the next
  // tile a wave only waits until all 16 have reported -- which, as a rule, they did long ago: a
  // wave that is through with phase B goes on into the next tile while others still work.
  auto land = [&](int bnext) {  // my share of the prefetched tile (and, last wave, its weights)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == WAVES - 1) {  // per row (lane = row): vA = c_a s^2, vB = c_b s y  ->  w = vA tot + vB
      double *wn = wts + bnext * 256;
      wn[lane] = ca * scn * scn;
      if (RO) {
        wn[64 + lane] = cb * scn * yn;
        wn[128 + lane] = scn;
        wn[192 + lane] = yn;
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

    // ---- phase A: per-wave row sums of sum_k a_k prod_k, 8 rows at a time --------------------
    if (live) {
      Hm3A<NA> ca_{ad, av, {}, 1.0, 0.0};
      double *redw = red + wave * kHm3RedPitch;
#pragma unroll 1
      for (int rc = 0; rc < kTileRows; rc += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) ca_.acc[r] = 0.0;
        tl_star_run<W, 8, K>(ca_, shape);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          ad[i] += (rc + 8 < kTileRows) ? 8 * 8 : -(kTileRows - 8) * 8;  // next chunk, or back to row 0
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
    // (s_barrier behind an lgkmcnt wait only: __syncthreads would also wait for the vector-memory
    // counter, i.e. at the second barrier for the tile requested a moment before)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's row sums are in red -- and every wave is past the previous tile
    asm volatile("" ::: "memory");
    if (tile + 1 < t1) prefetch(tile + 1, bsel ^ 1);

    // ---- middle: tot_r over the waves, w_r; wave w takes rows 4 w .. 4 w + 3 ------------------
    // lane = (row, part): part < 16 indexes the waves' partial sums and the left-over terms
    // part, part + 16, ...
    {
      const int row = 4 * wave + (lane >> 4), prt = lane & 15;
      const double *tb = lds + (bsel ? tile_doubles : 0) + row;  // this row in the tile
      auto left_prod = [&](int j) {
        double v = 1.0;
#pragma unroll
        for (int e = 0; e < W; ++e) v *= tb[lad[j * W + e]];
        return v;
      };
      double t = red[prt * kHm3RedPitch + row];
#pragma unroll
      for (int jj = 0; jj < kHm3NL; ++jj) {
        const int j = jj * 16 + prt;
        if (jj * 16 < nleft && j < nleft) t = fma(la[j], left_prod(j), t);
        if (jj % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // (three terms' reads in flight, not twelve)
      }
      t = row16_ror_add<8>(t);
      t = row16_ror_add<4>(t);
      t = row16_ror_add<2>(t);
      t = row16_ror_add<1>(t);
      double wv = wt[row] * t;  // (every lane of the row: its left-over terms want the weight)
      if (RO) wv += wt[64 + row];
#pragma unroll
      for (int jj = 0; jj < kHm3NL; ++jj) {
        const int j = jj * 16 + prt;
        if (jj * 16 < nleft && j < nleft) accl[jj] = fma(left_prod(j), wv, accl[jj]);
        if (jj % 3 == 2) __builtin_amdgcn_sched_barrier(0);
      }
      if (prt == 0) {
        if (RO) {
          const uint64_t grow = tile * kTileRows + row;
          if (grow < n) {
            const double yh = wt[128 + row] * t;
            if (yhat != nullptr) yhat[grow] = yh;
            const double dlt = yh - wt[192 + row];
            ssacc = fma(dlt, dlt, ssacc);
          }
        }
        wrow[row] = wv;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the row weights are complete (and red may be overwritten)
    asm volatile("" ::: "memory");

    // ---- phase B: acc_k += prod_k w_r -----------------------------------------------------------
    // after the last rows: row 0 of the other buffer
    const int32_t step_last = -(kTileRows - 16) * 8 + (bsel ? -(int32_t)tile_bytes : (int32_t)tile_bytes);
    if (live) {
      Hm3B<NA> cb_{ad, acc, wrow[lane], 0.0, 0.0, 0};
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
  if (ok) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t kt = shterm[sg * 4 + u];
      if (kt != 0xffffffffu) part[(uint64_t)blockIdx.x * p_pad + kt] = acc[u];
    }
  }
  if (nleft > 0) {
    // left-over term j = jj * 16 + part: its sum over the rows sits in the lanes (wave, row group,
    // part) of all waves -- over the row groups by shuffles, over the waves through LDS (the tile
    // buffers are free now)
    __syncthreads();
    double *lred = lds;  // [WAVES][kHm3NL][16]
#pragma unroll
    for (int jj = 0; jj < kHm3NL; ++jj) {
      double v = accl[jj];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) lred[(wave * kHm3NL + jj) * 16 + lane] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nleft; j += WAVES * 64) {
      double s = 0.0;
      for (int w = 0; w < WAVES; ++w) s += lred[(w * kHm3NL + j / 16) * 16 + (j & 15)];
      part[(uint64_t)blockIdx.x * p_pad + left_term[j]] = s;
    }
  }
  if (RO && sspart != nullptr) {
    // the residual sums sit in lanes 0, 16, 32, 48 of every wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ssacc += __shfl_xor(ssacc, off, 64);
    if (lane == 0) ssw[wave] = ssacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int w = 0; w < WAVES; ++w) s += ssw[w];
      sspart[blockIdx.x] = s;
    }
  }
}

template <int W2, int K>
int run_hm3(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca, double cb,
            double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles, uint64_t tps,
            size_t lds, const double *stop0, const double *stop1) {
#define OB_HM3_LAUNCH(RO_)                                                                                 \
  do {                                                                                                     \
    OB_TRY(ensure_dyn_lds((const void *)k_hm3<W2, K, RO_>, lds));                                          \
    hipLaunchKernelGGL((k_hm3<W2, K, RO_>), dim3(nsplit), dim3(kHm3Waves * 64), lds, cur_stream(), b.bm.p, \
                       b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.sh_cols.p,             \
                       (const uint32_t *)t.sh_term.p, (const uint32_t *)t.sh_shape.p, d_a, (int)t.p, d_y,  \
                       ca, cb, b.n, ntiles, tps, t.p_pad, part, d_yhat, sspart, stop0, stop1,              \
                       (int)t.sh.nsw_family, (const uint32_t *)t.sh_left_term.p,                           \
                       (const uint32_t *)t.sh_left_cols.p, (int)t.sh.nleft);                               \
  } while (0)
  if (d_y != nullptr)
    OB_HM3_LAUNCH(true);
  else
    OB_HM3_LAUNCH(false);
#undef OB_HM3_LAUNCH
  OB_HIP(hipGetLastError());
  return 0;
}

size_t hm3_lds_bytes(const obhip_terms &t) {
  return ((size_t)2 * t.Mu * kTlPitch + 2 * 256 + kHm3Waves * kHm3RedPitch + 64 + kHm3Waves + kHm3LeftMax) *
             sizeof(double) +
         ((size_t)kHm3LeftMax * t.W + 4) * sizeof(uint32_t);
}

}  // namespace

// the term sets this kernel takes: grouped into stars, one family star per lane of 9 to 16 waves
// (below that k_hm2's one or two terms per lane keep all 16 waves busy), few left-over terms, two
// tiles in LDS
bool hm3_supports(const obhip_terms &t) {
  const int w2 = (int)(t.W / 2);
  // (the epilogue's [16][12][16] sums of the left-over terms go through the tile buffers: 24 KB)
  return share_wanted() && t.sh.ok && w2 >= 1 && w2 <= 3 && t.sh.nsw_family >= 9 && t.sh.nsw_family <= 16 &&
         t.sh.nleft <= (uint64_t)kHm3LeftMax && hm3_lds_bytes(t) <= (size_t)156 * 1024 &&
         (size_t)2 * t.Mu * kTlPitch >= (size_t)kHm3Waves * kHm3NL * 16;
}

int launch_hm3(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca,
               double cb, double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles,
               uint64_t tps, const double *stop0, const double *stop1) {
  const size_t lds = hm3_lds_bytes(t);
#define OB_HM3(W2_, K_) \
  return run_hm3<W2_, K_>(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, nsplit, ntiles, tps, lds, stop0, stop1)
  switch ((int)(t.W / 2)) {
    case 1: OB_HM3(1, 12);
    case 2: OB_HM3(2, 12);
    default: OB_HM3(3, 10);  // (terms of 7 and 8 factors: 25 spilled registers -- not taken, hm3_supports)
  }
#undef OB_HM3
}

}  // namespace obhip
