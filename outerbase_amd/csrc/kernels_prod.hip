// Matrix-free outer-product kernels for gfx950:
//   getmat : B (n x p) materialised           getm_    src/linalg.cpp:647-715
//   mm     : out = B a   (and B^2 a)          prodmm_  src/linalg.cpp:57-131
//   tmm    : out = B^T a (and (B^2)^T a)      tprodmm_ src/linalg.cpp:286-355
// with B[i,k] = basescale[i] * prod_{l: t_kl>0} basemat[i, col(l, t_kl)].
//
// Two generations live here.
//
// Term-per-lane (default for terms of up to 8 factors): k_tmm_tl, k_mm_tl, k_materialize_tl
// (and k_predict_tl in kernels_predict.hip).  lane = term, the LDS byte addresses of a term's
// columns are loop-invariant registers, the row is the immediate offset of the ds_read_b64;
// persistent blocks of 8 waves walk a range of 64-row tiles staged as [column][65].  The LDS
// pipe is the bound (80 % busy at the benchmark size).  Described at k_tmm_tl.
//
// Lane-per-row (first generation; terms of more than 8 factors, getmat): k_mm, k_tmm.  All
// stage one 64-row tile of the used basemat columns in LDS
// ([column][row], one 512-byte run per column, straight from the tile-blocked
// HBM layout) and map lane = row, so every LDS read in the Hadamard product is
// a conflict-free ds_read_b64 at a wave-uniform column.
//
// Term tables: a wave works on 64 terms at a time.  Lane j holds the packed
// column list (and for mm the coefficient) of term k0 + j in registers, loaded
// with one coalesced vector load per group; inside the fully unrolled 64-term
// loop the entries are broadcast with v_readlane into SGPRs.  No scalar memory
// load -- and therefore no load latency -- sits in the inner loop.
//
// tmm keeps a 64-term x 64-row block of partial sums in registers per wave
// (acc[t], lane = row mod 64) across all its row tiles and reduces across lanes
// once at the end; the next tile's global loads are in flight while the current tile
// is consumed.
#include <atomic>

#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

constexpr int kMaxMuLds = 304;  // 304 * 64 * 8 B = 152 KiB of the 160 KiB LDS
constexpr int kMaxW2 = 4;       // register-resident column lists: up to 8 columns per term

__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// product of the staged columns of the term held by lane t, for this lane's row
template <int W2>
__device__ __forceinline__ double term_prod_rl(const double *__restrict__ lds,
                                               const uint32_t (&cw)[W2], int t, int lane, double v) {
#pragma unroll
  for (int w = 0; w < W2; ++w) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cw[w], t);
    v *= lds[(c & 0xffffu) * kTileRows + lane];
    v *= lds[(c >> 16) * kTileRows + lane];
  }
  return v;
}

// generic fallback (more than 8 columns per term): column words from memory
__device__ __forceinline__ double term_prod_mem(const double *__restrict__ lds,
                                                const uint32_t *__restrict__ cw, int W2, int lane,
                                                double v) {
  for (int w = 0; w < W2; ++w) {
    const uint32_t c = cw[w];
    v *= lds[(c & 0xffffu) * kTileRows + lane];
    v *= lds[(c >> 16) * kTileRows + lane];
  }
  return v;
}

template <int W2>
__device__ __forceinline__ void load_cw(uint32_t (&cw)[W2], const uint32_t *__restrict__ colsw,
                                        int k) {
#pragma unroll
  for (int w = 0; w < W2; ++w) cw[w] = colsw[(size_t)k * W2 + w];
}

// ---- mm / getmat -----------------------------------------------------------------------
// MODE 0: out = B a;  MODE 1: out = B^2 a;  MODE 2: materialise B (a unused)
// 8 waves share one staged tile: 3 blocks x 8 waves per CU instead of 3 x 4 (the tile is
// what limits residency), which is what hides the LDS latency of the products
constexpr int kMmThreads = 512, kMmWaves = kMmThreads / 64;

template <int W2, int MODE>
__global__ void __launch_bounds__(kMmThreads)
k_mm(const double *__restrict__ bm, const double *__restrict__ scale,
     const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc, const uint32_t *__restrict__ colsw,
     int W2rt, int p, const double *__restrict__ a, double *__restrict__ out, uint64_t n,
     uint64_t ld /* MODE 2: leading dimension of the column-major result */) {
  extern __shared__ double lds[];
  double *red = lds + (size_t)Mu * kTileRows;  // [kMmWaves][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = blockIdx.x;
  const uint64_t row = tile * kTileRows + lane;
  if (MODE == 1)
    stage_tile<true, false>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, kMmThreads);
  else
    stage_tile<false, false>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, kMmThreads);
  __syncthreads();
  const double s = row < n ? scale[row] : 0.0;
  double acc = 0.0;
  const int ngroups = (p + 63) / 64;
  if constexpr (W2 > 0) {
    uint32_t cw[W2], cwn[W2];
    double av = 0.0, avn = 0.0;
    int g = wave;
    if (g < ngroups) {
      const int k = min(g * 64 + lane, p - 1);  // the table is padded, a[] is not
      load_cw(cw, colsw, g * 64 + lane);
      if (MODE != 2) av = a[k];
    }
    for (; g < ngroups; g += kMmWaves) {
      const int gn = g + kMmWaves;
      if (gn < ngroups) {  // next group's table while this one is consumed
        load_cw(cwn, colsw, gn * 64 + lane);
        if (MODE != 2) avn = a[min(gn * 64 + lane, p - 1)];
      }
      const int k0 = g * 64;
      const int cnt = min(64, p - k0);
      if (cnt == 64) {
#pragma unroll
        for (int t = 0; t < 64; ++t) {
          if (MODE == 2) {
            const double v = term_prod_rl<W2>(lds, cw, t, lane, s);
            if (row < n) out[(uint64_t)(k0 + t) * ld + row] = v;
          } else {
            acc += term_prod_rl<W2>(lds, cw, t, lane, readlane_f64(av, t));
          }
        }
      } else {
        for (int t = 0; t < cnt; ++t) {
          if (MODE == 2) {
            const double v = term_prod_rl<W2>(lds, cw, t, lane, s);
            if (row < n) out[(uint64_t)(k0 + t) * ld + row] = v;
          } else {
            acc += term_prod_rl<W2>(lds, cw, t, lane, readlane_f64(av, t));
          }
        }
      }
#pragma unroll
      for (int w = 0; w < W2; ++w) cw[w] = cwn[w];
      av = avn;
    }
  } else {
    for (int k = wave; k < p; k += kMmWaves) {
      if (MODE == 2) {
        const double v = term_prod_mem(lds, colsw + (size_t)k * W2rt, W2rt, lane, s);
        if (row < n) out[(uint64_t)k * ld + row] = v;
      } else {
        acc += term_prod_mem(lds, colsw + (size_t)k * W2rt, W2rt, lane, a[k]);
      }
    }
  }
  if (MODE == 2) return;
  red[wave * kTileRows + lane] = acc;
  __syncthreads();
  if (wave == 0 && row < n) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < kMmWaves; ++q) tot += red[q * kTileRows + lane];
    out[row] = tot * (MODE == 1 ? s * s : s);
  }
}

// ---- tmm -----------------------------------------------------------------------------------
// grid = (row splits, p_pad / 256); wave w of a block owns terms
// blockIdx.y*256 + w*64 ... +63.  kPre registers per thread hold the next tile.
constexpr int kTmmPre = 32;  // doubles per thread => Mu <= 128 with prefetch

template <int W2, bool SQ, bool PREFETCH>
__global__ void __launch_bounds__(256)
k_tmm(const double *__restrict__ bm, const double *__restrict__ scale,
      const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc, const uint32_t *__restrict__ colsw,
      int W2rt, const double *__restrict__ a, uint64_t n, uint64_t ntiles,
      uint64_t tiles_per_split, uint64_t p_pad, double *__restrict__ part) {
  extern __shared__ double lds[];
  int *lu = (int *)(lds + (size_t)Mu * kTileRows);  // [Mu] ucol[u] * 64
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k0 = blockIdx.y * 256 + wave * 64;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  for (int u = threadIdx.x; u < Mu; u += 256) lu[u] = (int)ucol[u] * kTileRows;
  __syncthreads();

  uint32_t cw[W2 > 0 ? W2 : 1];
  if constexpr (W2 > 0) load_cw(cw, colsw, k0 + lane);
  double acc[64];
#pragma unroll
  for (int t = 0; t < 64; ++t) acc[t] = 0.0;

  // element (u, r) of a tile: thread t moves row r = t & 63 of columns u = (t >> 6) + 4 q
  double pre[PREFETCH ? kTmmPre : 1];
  double prevs = 0.0;
  auto fetch = [&](uint64_t tile) {
    const double *src = bm + tile * Mc * kTileRows + lane;
#pragma unroll
    for (int q = 0; q < kTmmPre; ++q) {
      const int u = wave + 4 * q;
      pre[q] = u < Mu ? src[lu[u]] : 0.0;
    }
    const uint64_t row = tile * kTileRows + lane;
    prevs = 0.0;
    if (row < n) {
      const double s = scale[row];
      prevs = a[row] * (SQ ? s * s : s);  // b = basescale % a, linalg.cpp:305
    }
  };
  if (PREFETCH && t0 < t1) fetch(t0);

  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();  // every wave is done with the previous tile
    double vs;
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < kTmmPre; ++q) {
        const int u = wave + 4 * q;
        if (u < Mu) lds[u * kTileRows + lane] = SQ ? pre[q] * pre[q] : pre[q];
      }
      vs = prevs;
    } else {
      stage_tile<SQ, false>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, 256);
      const uint64_t row = tile * kTileRows + lane;
      vs = 0.0;
      if (row < n) {
        const double s = scale[row];
        vs = a[row] * (SQ ? s * s : s);
      }
    }
    __syncthreads();
    if (PREFETCH && tile + 1 < t1) fetch(tile + 1);
    if constexpr (W2 > 0) {
#pragma unroll
      for (int t = 0; t < 64; ++t) acc[t] += term_prod_rl<W2>(lds, cw, t, lane, vs);
    } else {
#pragma unroll
      for (int t = 0; t < 64; ++t)
        acc[t] += term_prod_mem(lds, colsw + (size_t)(k0 + t) * W2rt, W2rt, lane, vs);
    }
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

// ---- tmm, term-per-lane ---------------------------------------------------------------------
// The default B^T a kernel.  lane = term: the LDS byte addresses of a term's columns are
// loop-invariant VGPRs and the row is the immediate offset of the ds_read_b64, so the inner
// loop has no address arithmetic, no column broadcast and no cross-lane reduction; the
// weight a[r]*scale[r] of the row is one v_readlane pair shared by all the terms a wave
// holds.  Tile layout in LDS: [column][65] doubles -- with the odd pitch, lanes that read
// different columns hit banks (2 col + 2 r) mod 64, lanes that read the same column (the
// common case, neighbouring terms share factors) broadcast.
// A block of 8 waves owns up to 8 * NPAIR * 2 * 64 terms (4096 for NPAIR = 4) and a
// contiguous range of row tiles, so a tile is staged once per 4096 terms (the lane = row
// kernel above stages it once per 256).
// The reads are issued by hand (the compiler pairs the rows into ds_read2_b64, half the LDS
// rate, and spills the addresses): one "unit" = the W reads of one term for one row, D units
// in flight per wave (lgkmcnt counts to 15), LDS returns in order, so s_waitcnt
// lgkmcnt((D-1) W) releases the oldest unit; the "+v" operands tie each wait to the
// registers it releases.
// Terms are taken in the order of obhip_terms::sperm (falling number of factors): the
// NPAIR x 2 x 64 consecutive slots of a wave then hold terms of nearly one length, and the wave
// runs the pipeline instantiated for that length (TlPipe<WE, ...>): at the benchmark size
// 25 instead of 32 column reads per lane and row.
// DUAL: B^T a and (B^2)^T a2 in one pass -- the products are formed once, the squares are one more
// multiply and multiply-add per (term, row): the cold start of the PCG needs e^{-2 sigma} B^T y
// and the preconditioner's sqcolsums (loglik_gauss.cpp:125,154-157), which were two passes.
template <int W, int NU, bool DUAL = false>
struct TmmCtx {
  uint32_t ad[NU][W];
  double acc[NU];
  double acc2[DUAL ? NU : 1];
  double vs, vs2;  // weights of row = lane (vs2: of the squared products)
  double vr, vr2;  // weights of the current row, wave-uniform
  int rc;          // first row of the chunk
  template <int RR>
  __device__ __forceinline__ void row() {
    vr = readlane_f64(vs, rc + RR);
    if constexpr (DUAL) vr2 = readlane_f64(vs2, rc + RR);
  }
  template <int RR, int UNIT>
  __device__ __forceinline__ void use(double v) {
    acc[UNIT] = fma(v, vr, acc[UNIT]);
    if constexpr (DUAL) acc2[UNIT] = fma(v * v, vr2, acc2[UNIT]);
  }
};

template <int W, int NU, bool DUAL>
__device__ __forceinline__ void tmm_tile(TmmCtx<W, NU, DUAL> &c, int wea, int web) {
  // 8 units x W addresses already fill the register budget: 8 reads in flight instead of 12
  constexpr int kInflight = NU * W >= 32 ? 8 : 12;
#pragma unroll 1
  for (int rc = 0; rc < kTileRows; rc += kTlChunk) {
    c.rc = rc;
    if constexpr (NU == 1) {
      tl_run_half<W, 1, kTlChunk, kInflight, 0>(c, wea);
    } else {
      tl_run_half<W, NU / 2, kTlChunk, kInflight, 0>(c, wea);
      tl_run_half<W, NU / 2, kTlChunk, kInflight, NU / 2>(c, web);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int j = 0; j < W; ++j) {
        // next chunk, or back to row 0; opaque so that no second copy of the addresses
        // is kept (and spilled) across the tile loop
        c.ad[u][j] += (rc + kTlChunk < kTileRows) ? kTlChunk * 8 : -(kTileRows - kTlChunk) * 8;
        asm volatile("" : "+v"(c.ad[u][j]));
      }
  }
}

template <int W2, bool SQ, int NPAIR, bool PREFETCH, bool DUAL = false>
__global__ void __launch_bounds__(kTlThreads, 4)
k_tmm_tl(const double *__restrict__ bm, const double *__restrict__ scale,
         const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc,
         const uint32_t *__restrict__ colsw, const uint32_t *__restrict__ sperm,
         const double *__restrict__ a, uint64_t n, uint64_t ntiles, uint64_t tiles_per_split,
         uint64_t p_pad, double *__restrict__ part, const double *__restrict__ a2 = nullptr,
         double *__restrict__ part2 = nullptr) {
  static_assert(!(DUAL && SQ), "DUAL forms the squares from the plain products");
  extern __shared__ double lds[];
  constexpr int W = 2 * W2, NU = NPAIR * kTlGP;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  // my terms: NU * 64 consecutive slots of the sorted order (tl_slot); the two halves of the
  // units get their own pipeline width
  TmmCtx<W, NU, DUAL> c;
  int nza = 1, nzb = 1;
  bool live = false;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = tl_slot<NU>(blockIdx.y, wave, u, lane);
    const bool ok = slot < p_pad;
    live = live || ok;
    const uint64_t k = ok ? sperm[slot] : 0;
    c.acc[u] = 0.0;
    if constexpr (DUAL) c.acc2[u] = 0.0;
    uint32_t cw[W2];
#pragma unroll
    for (int w = 0; w < W2; ++w) {
      cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
      c.ad[u][2 * w] = (cw[w] & 0xffffu) * (kTlPitch * 8);
      c.ad[u][2 * w + 1] = (cw[w] >> 16) * (kTlPitch * 8);
    }
    if (NU == 1 || u < NU / 2)
      nza = max(nza, tl_nnz<W2>(cw));
    else
      nzb = max(nzb, tl_nnz<W2>(cw));
  }
  const int wea = tl_variant<W>(wave_max_i32(nza)), web = tl_variant<W>(wave_max_i32(nzb));
  live = wave_max_i32(live ? 1 : 0) != 0;

  // this wave stages columns u = wave + 8 q of every tile: their tile offsets, wave-uniform
  int lu[PREFETCH ? kTlPre : 1];
  double pre[PREFETCH ? kTlPre : 1];
  if (PREFETCH) {
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      lu[q] = u < Mu ? __builtin_amdgcn_readfirstlane((int)ucol[u] * kTileRows) : 0;
    }
  }
  double vsn = 0.0, vs2n = 0.0;
  auto weight = [&](uint64_t tile) {
    const uint64_t row = tile * kTileRows + lane;
    double v = 0.0;
    vs2n = 0.0;
    if (row < n) {
      const double sc = scale[row];
      v = a[row] * (SQ ? sc * sc : sc);  // b = basescale % a, linalg.cpp:305
      if (DUAL) vs2n = (a2 ? a2[row] : 1.0) * sc * sc;
    }
    return v;
  };
  auto fetch = [&](uint64_t tile) {
    const double *src = bm + tile * Mc * kTileRows + lane;
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      pre[q] = u < Mu ? src[lu[q]] : 0.0;
    }
    vsn = weight(tile);
  };
  if (PREFETCH && t0 < t1) fetch(t0);

  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();  // every wave is done with the previous tile
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < kTlPre; ++q) {
        const int u = wave + kTlWaves * q;
        if (u < Mu) lds[u * kTlPitch + lane] = SQ ? pre[q] * pre[q] : pre[q];
      }
      c.vs = vsn;
      c.vs2 = vs2n;
    } else {
      const double *src = bm + tile * Mc * kTileRows + lane;
      for (int u = wave; u < Mu; u += kTlWaves) {
        const double v = src[(size_t)ucol[u] * kTileRows];
        lds[u * kTlPitch + lane] = SQ ? v * v : v;
      }
      c.vs = weight(tile);
      c.vs2 = vs2n;
    }
    __syncthreads();
    if (PREFETCH && tile + 1 < t1) fetch(tile + 1);
    if (!live) continue;  // (whole waves beyond p_pad in the last block along p)
    tmm_tile<W, NU, DUAL>(c, wea, web);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = tl_slot<NU>(blockIdx.y, wave, u, lane);
    if (slot < p_pad) {
      part[(uint64_t)blockIdx.x * p_pad + sperm[slot]] = c.acc[u];
      if constexpr (DUAL) part2[(uint64_t)blockIdx.x * p_pad + sperm[slot]] = c.acc2[u];
    }
  }
}

// ---- row-major design matrix, term-per-lane ------------------------------------------------------
// B[row][k] for the materialised-B Gram kernel (kernels_gram_panel.hip).  lane = two adjacent
// terms, so a wave instruction stores 64 x 16 B = 1 KB of one row of B and no transpose is
// needed; products exactly as in k_tmm_tl, the row's basescale is the last factor.
// WITH_Y: B^T y on the way (acc[unit] += B[row][term] y[row]): the fit's only other pass over the
// basis (k_tmm_tl for B^T y, 1 ms per 1e6 rows) rides on products this kernel forms anyway; it is
// bound by its HBM writes, so the extra multiply-add per (term, row) is free.
template <int W, int NPAIR, bool WITH_Y>
struct MtCtx {
  uint32_t ad[NPAIR * 2][W];  // unit 2 q + i = term i of the lane's pair in pair-group q
  uint32_t koff[NPAIR];       // first of the lane's two terms in pair-group q
  double sc;                  // basescale of row = lane
  double sr;                  // basescale of the current row, wave-uniform
  double yl, yr;              // y of row = lane / of the current row
  double acc[WITH_Y ? NPAIR * 2 : 1];
  double v0;
  double *rowp;               // B + current row * p_pad (uniform)
  uint64_t p_pad;
  int rc;
  template <int RR>
  __device__ __forceinline__ void row() {
    sr = readlane_f64(sc, rc + RR);
    if constexpr (WITH_Y) yr = readlane_f64(yl, rc + RR);
    if constexpr (RR > 0) rowp += p_pad;
  }
  template <int RR, int UNIT>
  __device__ __forceinline__ void use(double v) {
    v *= sr;
    if constexpr (WITH_Y) acc[UNIT] = fma(v, yr, acc[UNIT]);
    if constexpr (UNIT % 2 == 0) {
      v0 = v;
    } else {
      double2 o;
      o.x = v0;
      o.y = v;
      *(double2 *)(rowp + koff[UNIT / 2]) = o;
    }
  }
};

template <int W2, int NPAIR, bool PREFETCH, bool WITH_Y>
__global__ void __launch_bounds__(kTlThreads, 4)
k_materialize_tl(const double *__restrict__ bm, const double *__restrict__ scale,
                 const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc,
                 const uint32_t *__restrict__ colsw, uint64_t n, uint64_t ntiles, uint64_t tiles_per_split,
                 uint64_t p_pad, double *__restrict__ out, const double *__restrict__ y,
                 double *__restrict__ ypart /* [gridDim.x][p_pad] partial B^T y */) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  // pair-group ((blockIdx.y * NPAIR + q) * 8 + wave) covers 128 terms, lane takes two
  MtCtx<W, NPAIR, WITH_Y> c;
  c.p_pad = p_pad;
  if constexpr (WITH_Y) {
#pragma unroll
    for (int u = 0; u < NPAIR * 2; ++u) c.acc[u] = 0.0;
  }
  bool any = false;
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) {
    const uint64_t k = (((uint64_t)blockIdx.y * NPAIR + q) * kTlWaves + wave) * 128 + 2 * lane;
    c.koff[q] = (uint32_t)min(k, p_pad - 2);
    any = any || k < p_pad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int w = 0; w < W2; ++w) {
        const uint32_t cw = k < p_pad ? colsw[(k + i) * W2 + w] : 0u;
        c.ad[2 * q + i][2 * w] = (cw & 0xffffu) * (kTlPitch * 8);
        c.ad[2 * q + i][2 * w + 1] = (cw >> 16) * (kTlPitch * 8);
      }
  }
  // pair-groups beyond p_pad exist only in the last block along p and only for whole waves
  const bool live = __builtin_amdgcn_readfirstlane((int)any) != 0;

  int lu[PREFETCH ? kTlPre : 1];
  double pre[PREFETCH ? kTlPre : 1];
  if (PREFETCH) {
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      lu[q] = u < Mu ? __builtin_amdgcn_readfirstlane((int)ucol[u] * kTileRows) : 0;
    }
  }
  double scn = 0.0, yn = 0.0;
  auto rowy = [&](uint64_t tile) {
    const uint64_t row = tile * kTileRows + lane;
    return WITH_Y && row < n ? y[row] : 0.0;
  };
  auto fetch = [&](uint64_t tile) {
    const double *src = bm + tile * Mc * kTileRows + lane;
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      pre[q] = u < Mu ? src[lu[q]] : 0.0;
    }
    scn = scale[tile * kTileRows + lane];  // 0 in padding rows
    yn = rowy(tile);
  };
  if (PREFETCH && t0 < t1) fetch(t0);

  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < kTlPre; ++q) {
        const int u = wave + kTlWaves * q;
        if (u < Mu) lds[u * kTlPitch + lane] = pre[q];
      }
      c.sc = scn;
      c.yl = yn;
    } else {
      const double *src = bm + tile * Mc * kTileRows + lane;
      for (int u = wave; u < Mu; u += kTlWaves) lds[u * kTlPitch + lane] = src[(size_t)ucol[u] * kTileRows];
      c.sc = scale[tile * kTileRows + lane];
      c.yl = rowy(tile);
    }
    __syncthreads();
    if (PREFETCH && tile + 1 < t1) fetch(tile + 1);
    if (!live) continue;
#pragma unroll 1
    for (int rc = 0; rc < kTileRows; rc += kTlChunk) {
      c.rc = rc;
      c.rowp = out + (tile * kTileRows + rc) * p_pad;
      TlPipe<W, W, NPAIR * 2, kTlChunk>::run(c);
#pragma unroll
      for (int u = 0; u < NPAIR * 2; ++u)
#pragma unroll
        for (int j = 0; j < W; ++j) {
          c.ad[u][j] += (rc + kTlChunk < kTileRows) ? kTlChunk * 8 : -(kTileRows - kTlChunk) * 8;
          asm volatile("" : "+v"(c.ad[u][j]));
        }
    }
  }
  if constexpr (WITH_Y) {
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) {
      const uint64_t k = (((uint64_t)blockIdx.y * NPAIR + q) * kTlWaves + wave) * 128 + 2 * lane;
      if (k < p_pad) {
        ypart[(uint64_t)blockIdx.x * p_pad + k] = c.acc[2 * q];
        ypart[(uint64_t)blockIdx.x * p_pad + k + 1] = c.acc[2 * q + 1];
      }
    }
  }
}

// ---- mm, term-per-lane ----------------------------------------------------------------------
// out = B a with the same lane = term layout: a wave holds NG terms per lane (addresses and
// coefficient in registers for the whole launch), accumulates a_k * prod over its terms for
// 8 rows at a time (acc[row], lane = term) and only then reduces across the 64 lanes: a
// butterfly with v_permlane32_swap / v_permlane16_swap takes the 16 accumulators to 4 (each
// 16-lane row of a register then belongs to one tile row), 4 DPP rotations finish the sum.
// That is ~5 VALU instructions per row and wave against 4 per (term, row) for the products,
// where the lane = row kernel spends 13 per (term, row) (column broadcast, address, product).
constexpr int kMlChunk = 8;  // rows per chunk: 8 accumulators, the register budget is tight

template <int W, int NG>
struct MmCtx {
  uint32_t ad[NG][W];
  double av[NG];
  double acc[kMlChunk];
  template <int RR>
  __device__ __forceinline__ void row() {}
  template <int RR, int UNIT>
  __device__ __forceinline__ void use(double v) {
    acc[RR] = fma(v, av[UNIT], acc[RR]);
  }
};

// one tile: 8-row chunks, after each the 8 accumulators x 64 lanes go to red[rc .. rc + 7]
template <int W, int NG>
__device__ __forceinline__ void mm_tile(MmCtx<W, NG> &c, int we, double *__restrict__ redw,
                                        int lane) {
  constexpr int kInflight = NG * W >= 32 ? 8 : 12;
#pragma unroll 1
  for (int rc = 0; rc < kTileRows; rc += kMlChunk) {
#pragma unroll
    for (int r = 0; r < kMlChunk; ++r) c.acc[r] = 0.0;
    tl_run_half<W, NG, kMlChunk, kInflight, 0>(c, we);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int j = 0; j < W; ++j) {
        c.ad[g][j] += (rc + kMlChunk < kTileRows) ? kMlChunk * 8 : -(kTileRows - kMlChunk) * 8;
        asm volatile("" : "+v"(c.ad[g][j]));
      }
    // 8 accumulators x 64 lanes -> 2 registers whose 16-lane row q holds tile row
    // rc + i + 2 q, then the sum over the 16 lanes of the row
    static_assert(kMlChunk == 8, "butterfly below reduces 8 rows");
    double s4[4], s2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) s4[i] = swap32_sum(c.acc[i], c.acc[i + 4]);
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

template <int W2, bool SQ, int NG, bool PREFETCH>
__global__ void __launch_bounds__(kTlThreads, 4)
k_mm_tl(const double *__restrict__ bm, const double *__restrict__ scale,
        const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc, const uint32_t *__restrict__ colsw,
        const uint32_t *__restrict__ sperm, const double *__restrict__ a, int p, uint64_t n,
        uint64_t n_pad, uint64_t ntiles,
        uint64_t tiles_per_split, uint64_t p_pad, double *__restrict__ out,
        double *__restrict__ part) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2;
  double *red = lds + (size_t)Mu * kTlPitch;  // [8 waves][64 rows]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  // my terms: NG * 64 consecutive slots of the sorted order (tl_slot)
  MmCtx<W, NG> c;
  int nzmax = 1;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const uint64_t slot = tl_slot<NG>(blockIdx.y, wave, g, lane);
    const bool ok = slot < p_pad;
    const uint64_t k = ok ? sperm[slot] : 0;
    c.av[g] = ok && k < (uint64_t)p ? a[k] : 0.0;
    uint32_t cw[W2];
#pragma unroll
    for (int w = 0; w < W2; ++w) {
      cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
      c.ad[g][2 * w] = (cw[w] & 0xffffu) * (kTlPitch * 8);
      c.ad[g][2 * w + 1] = (cw[w] >> 16) * (kTlPitch * 8);
    }
    nzmax = max(nzmax, tl_nnz<W2>(cw));
  }
  const int we = tl_variant<W>(wave_max_i32(nzmax));
  const bool live = (((uint64_t)blockIdx.y * kTlWaves + wave) * NG) * 64 < p_pad;

  int lu[PREFETCH ? kTlPre : 1];
  double pre[PREFETCH ? kTlPre : 1];
  if (PREFETCH) {
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      lu[q] = u < Mu ? __builtin_amdgcn_readfirstlane((int)ucol[u] * kTileRows) : 0;
    }
  }
  auto fetch = [&](uint64_t tile) {
    const double *src = bm + tile * Mc * kTileRows + lane;
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      pre[q] = u < Mu ? src[lu[q]] : 0.0;
    }
  };
  // row sums of a finished tile: wave 0, lane = row
  auto emit = [&](uint64_t tile) {
    const uint64_t row = tile * kTileRows + lane;
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < kTlWaves; ++q) tot += red[q * kTileRows + lane];
    if (part != nullptr) {
      part[(uint64_t)blockIdx.y * n_pad + row] = tot;  // scaled by k_mm_tl_sum
    } else if (row < n) {
      const double sc = scale[row];
      out[row] = tot * (SQ ? sc * sc : sc);
    }
  };
  if (PREFETCH && t0 < t1) fetch(t0);

  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();  // every wave is done with the previous tile, its row sums are in red
    if (tile > t0 && wave == 0) emit(tile - 1);
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < kTlPre; ++q) {
        const int u = wave + kTlWaves * q;
        if (u < Mu) lds[u * kTlPitch + lane] = SQ ? pre[q] * pre[q] : pre[q];
      }
    } else {
      const double *src = bm + tile * Mc * kTileRows + lane;
      for (int u = wave; u < Mu; u += kTlWaves) {
        const double v = src[(size_t)ucol[u] * kTileRows];
        lds[u * kTlPitch + lane] = SQ ? v * v : v;
      }
    }
    __syncthreads();  // tile staged; wave 0 has read red
    if (PREFETCH && tile + 1 < t1) fetch(tile + 1);
    double *redw = red + wave * kTileRows;
    if (!live)  // whole waves beyond p_pad in the last block along p
      redw[lane] = 0.0;
    else
      mm_tile<W, NG>(c, we, redw, lane);
  }
  __syncthreads();
  if (t0 < t1 && wave == 0) emit(t1 - 1);
}

// ---- B^T (c_a B a + c_b y), term-per-lane, ONE pass over the basis --------------------------------
// The Hessian product of the PCG (loglik_gauss::hessmult, loglik_gauss.cpp:137-145:
// B^T (B p) -- c_a = 1, c_b = 0) and the gradient pass of its update() (loglik_gauss.cpp:117-125:
// yhat = B theta, B^T (e^{-2 sigma} (y - yhat)) -- c_a = -e^{-2 sigma}, c_b = e^{-2 sigma}) as one
// kernel instead of k_mm_tl followed by k_tmm_tl.  Both are bound by the LDS column reads
// (W per term and row), and the two kernels read every column twice: once
// for B a, once for B^T r.  Here a block holds ALL terms (8 waves x NU x 64), takes a tile in
// chunks of 4 rows, and keeps the NU x 4 term products of a chunk in registers:
//   1. products as in k_mm_tl (TlPipe): prod[u][r], and s[r] += a_u prod[u][r];
//   2. s[r] summed over the 64 lanes (permlane swaps + DPP) and, through 32 doubles of LDS and
//      one s_barrier, over the 8 waves: tot_r = sum_k a_k prod_k(row r), wave-uniform;
//   3. w_r = c_a s_r^2 tot_r + c_b s_r y_r, and acc[u] += prod[u][r] w_r as in k_tmm_tl.
// Half the LDS reads of the two-kernel form, the tile staged once, no n-vector through HBM
// (yhat is written only when the caller wants it).
constexpr int kHmChunk = 4;

template <int W, int NU>
struct HmCtx {
  uint32_t ad[NU][W];
  double av[NU];
  double acc[NU];
  double prod[NU][kHmChunk];
  double s[kHmChunk];
  template <int RR>
  __device__ __forceinline__ void row() {}
  template <int RR, int UNIT>
  __device__ __forceinline__ void use(double v) {
    prod[UNIT][RR] = v;
    s[RR] = fma(v, av[UNIT], s[RR]);
  }
};

// one sub-chunk of 4 rows (ROW0 .. ROW0 + 3 beyond the rows the addresses point at): steps 1-3
template <int W, int NU, int ROW0, bool ROWS_OUT>
__device__ __forceinline__ void hm_subchunk(HmCtx<W, NU> &c, bool live, int wea, int web, double *rb,
                                            int wave, int lane, int rc /* tile row of ROW0 */,
                                            double vA, double vB, double &totrow) {
  constexpr int kInflight = NU * W >= 32 ? 8 : 12;
#pragma unroll
  for (int r = 0; r < kHmChunk; ++r) c.s[r] = 0.0;
  if (live) {
    if constexpr (NU == 1) {
      tl_run_half<W, 1, kHmChunk, kInflight, 0, ROW0>(c, wea);
    } else {
      tl_run_half<W, NU / 2, kHmChunk, kInflight, 0, ROW0>(c, wea);
      tl_run_half<W, NU / 2, kHmChunk, kInflight, NU / 2, ROW0>(c, web);
    }
  } else {
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < kHmChunk; ++r) c.prod[u][r] = 0.0;
  }
  // s[0..3] over the 64 lanes: 16-lane row q ends with the sum of s[q]
  static_assert(kHmChunk == 4, "the butterfly below reduces 4 rows");
  double v = swap16_sum(swap32_sum(c.s[0], c.s[2]), swap32_sum(c.s[1], c.s[3]));
  v = row16_ror_add<8>(v);
  v = row16_ror_add<4>(v);
  v = row16_ror_add<2>(v);
  v = row16_ror_add<1>(v);
  if ((lane & 15) == 0) rb[wave * kHmChunk + (lane >> 4)] = v;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (not vmcnt: the next tile's loads stay in flight)
  __builtin_amdgcn_s_barrier();
  // over the 8 waves: lane l takes red[wave (l & 31) / 4][row l % 4]; afterwards EVERY lane l holds
  // tot of row l % 4
  double t = rb[lane & 31];
  t = row16_ror_add<4>(t);
  t = row16_ror_add<8>(t);
  t += __shfl_xor(t, 16, 64);
  // w of row l % 4 in lane l: the row's vA, vB (lane = row) fetched by the lanes that want them
  const int src = rc + (lane & 3);
  double wl = __shfl(vA, src, 64) * t;
  if (ROWS_OUT) {
    wl += __shfl(vB, src, 64);
    if ((lane & ~3) == rc) totrow = t;  // lane = row keeps sum_k a_k prod_k of its row
  }
#pragma unroll
  for (int r = 0; r < kHmChunk; ++r) {
    const double wv = readlane_f64(wl, r);
#pragma unroll
    for (int u = 0; u < NU; ++u) c.acc[u] = fma(c.prod[u][r], wv, c.acc[u]);
  }
}

// ROWS_OUT: the update() form (y, yhat, sum of squared residuals); without it the Hessian product
template <int W2, int NU, bool PREFETCH, bool ROWS_OUT>
__global__ void __launch_bounds__(kTlThreads, 2)
k_hm_tl(const double *__restrict__ bm, const double *__restrict__ scale,
        const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc,
        const uint32_t *__restrict__ colsw, const uint32_t *__restrict__ sperm,
        const double *__restrict__ a, int p, const double *__restrict__ y, double ca, double cb,
        uint64_t n, uint64_t ntiles, uint64_t tiles_per_split, uint64_t p_pad,
        double *__restrict__ part, double *__restrict__ yhat, double *__restrict__ sspart) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2;
  double *red = lds + (size_t)Mu * kTlPitch;  // [2][8 waves][4 rows]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  HmCtx<W, NU> c;
  int nza = 1, nzb = 1;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = tl_slot<NU>(0, wave, u, lane);
    const bool ok = slot < p_pad;
    const uint64_t k = ok ? sperm[slot] : 0;
    c.acc[u] = 0.0;
    c.av[u] = ok && k < (uint64_t)p ? a[k] : 0.0;
    uint32_t cw[W2];
#pragma unroll
    for (int w = 0; w < W2; ++w) {
      cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
      c.ad[u][2 * w] = (cw[w] & 0xffffu) * (kTlPitch * 8);
      c.ad[u][2 * w + 1] = (cw[w] >> 16) * (kTlPitch * 8);
    }
    if (NU == 1 || u < NU / 2)
      nza = max(nza, tl_nnz<W2>(cw));
    else
      nzb = max(nzb, tl_nnz<W2>(cw));
  }
  const int wea = tl_variant<W>(wave_max_i32(nza)), web = tl_variant<W>(wave_max_i32(nzb));
  const bool live = ((uint64_t)wave * NU) * 64 < p_pad;  // (whole waves beyond p_pad: zeros)

  int lu[PREFETCH ? kTlPre : 1];
  double pre[PREFETCH ? kTlPre : 1];
  if (PREFETCH) {
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      lu[q] = u < Mu ? __builtin_amdgcn_readfirstlane((int)ucol[u] * kTileRows) : 0;
    }
  }
  // per row (lane = row): vA = c_a s^2, vB = c_b s y  ->  w = vA tot + vB
  double vAn = 0.0, vBn = 0.0, vA = 0.0, vB = 0.0;
  auto weights = [&](uint64_t tile) {
    const uint64_t row = tile * kTileRows + lane;
    vAn = vBn = 0.0;
    if (row < n) {
      const double sc = scale[row];
      vAn = ca * sc * sc;
      if (ROWS_OUT) vBn = cb * sc * y[row];
    }
  };
  auto fetch = [&](uint64_t tile) {
    const double *src = bm + tile * Mc * kTileRows + lane;
#pragma unroll
    for (int q = 0; q < kTlPre; ++q) {
      const int u = wave + kTlWaves * q;
      pre[q] = u < Mu ? src[lu[q]] : 0.0;
    }
    weights(tile);
  };
  if (PREFETCH && t0 < t1) fetch(t0);
  double ssacc = 0.0;  // wave 0: sum over its rows of (yhat - y)^2

  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();  // every wave is done with the previous tile
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < kTlPre; ++q) {
        const int u = wave + kTlWaves * q;
        if (u < Mu) lds[u * kTlPitch + lane] = pre[q];
      }
    } else {
      const double *src = bm + tile * Mc * kTileRows + lane;
      for (int u = wave; u < Mu; u += kTlWaves) lds[u * kTlPitch + lane] = src[(size_t)ucol[u] * kTileRows];
      weights(tile);
    }
    vA = vAn;
    vB = vBn;
    __syncthreads();
    if (PREFETCH && tile + 1 < t1) fetch(tile + 1);
    double totrow = 0.0;  // lane = row: sum_k a_k prod_k of this tile's row
#pragma unroll 1
    for (int rc = 0; rc < kTileRows; rc += 2 * kHmChunk) {
      // two sub-chunks per address update: the second reads at immediate row offsets 4 .. 7;
      // the cross-wave sums alternate between the two halves of red
      hm_subchunk<W, NU, 0, ROWS_OUT>(c, live, wea, web, red, wave, lane, rc, vA, vB, totrow);
      hm_subchunk<W, NU, kHmChunk, ROWS_OUT>(c, live, wea, web, red + kTlWaves * kHmChunk, wave, lane,
                                             rc + kHmChunk, vA, vB, totrow);
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int j = 0; j < W; ++j) {
          c.ad[u][j] += (rc + 2 * kHmChunk < kTileRows) ? 2 * kHmChunk * 8 : -(kTileRows - 2 * kHmChunk) * 8;
          asm volatile("" : "+v"(c.ad[u][j]));
        }
    }
    if (ROWS_OUT && wave == 0) {
      const uint64_t row = tile * kTileRows + lane;
      if (row < n) {
        const double yh = scale[row] * totrow;
        if (yhat != nullptr) yhat[row] = yh;
        const double dlt = yh - y[row];
        ssacc = fma(dlt, dlt, ssacc);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = tl_slot<NU>(0, wave, u, lane);
    if (slot < p_pad) part[(uint64_t)blockIdx.x * p_pad + sperm[slot]] = c.acc[u];
  }
  if (ROWS_OUT && wave == 0 && sspart != nullptr) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ssacc += __shfl_xor(ssacc, off, 64);
    if (lane == 0) sspart[blockIdx.x] = ssacc;
  }
}

// out[0] = sum of the per-block sums, in block order
__global__ void k_hm_ss(const double *__restrict__ sspart, int nblk, double *__restrict__ out) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += sspart[b];
    out[0] = s;
  }
}

// several blocks along p: out = scale * sum of their partial row sums
template <bool SQ>
__global__ void k_mm_tl_sum(const double *__restrict__ part, int pblocks, uint64_t n_pad,
                            const double *__restrict__ scale, uint64_t n,
                            double *__restrict__ out) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double tot = 0.0;
  for (int y = 0; y < pblocks; ++y) tot += part[(uint64_t)y * n_pad + row];
  const double sc = scale[row];
  out[row] = tot * (SQ ? sc * sc : sc);
}

// out[k] = sum over the row splits of part[split][k]: a block takes 64 terms, its 16 waves
// every 16th split each (coalesced 512-byte reads), LDS tree over the waves.  Fixed order,
// so the result does not depend on the launch.
constexpr int kRedThreads = 1024;
__global__ void __launch_bounds__(kRedThreads)
k_tmm_reduce(const double *__restrict__ part, int nsplit, uint64_t p_pad, int p,
             double *__restrict__ out) {
  __shared__ double red[kRedThreads / 64][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int r = w;
  for (; r + 48 < nsplit; r += 64) {
    s0 += part[(uint64_t)r * p_pad + k];
    s1 += part[(uint64_t)(r + 16) * p_pad + k];
    s2 += part[(uint64_t)(r + 32) * p_pad + k];
    s3 += part[(uint64_t)(r + 48) * p_pad + k];
  }
  for (; r < nsplit; r += 16) s0 += part[(uint64_t)r * p_pad + k];
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && k < p) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < kRedThreads / 64; ++q) tot += red[q][lane];
    out[k] = tot;
  }
}

// the same, and q[k] = e2 out[k] + prec[k] pv[k] (HmThen)
__global__ void __launch_bounds__(kRedThreads)
k_tmm_reduce_q(const double *__restrict__ part, int nsplit, uint64_t p_pad, int p, double *__restrict__ out,
               double e2, const double *__restrict__ prec, const double *__restrict__ pv, double *__restrict__ q) {
  __shared__ double red[kRedThreads / 64][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int r = w;
  for (; r + 48 < nsplit; r += 64) {
    s0 += part[(uint64_t)r * p_pad + k];
    s1 += part[(uint64_t)(r + 16) * p_pad + k];
    s2 += part[(uint64_t)(r + 32) * p_pad + k];
    s3 += part[(uint64_t)(r + 48) * p_pad + k];
  }
  for (; r < nsplit; r += 16) s0 += part[(uint64_t)r * p_pad + k];
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && k < p) {
    double tot = 0.0;
#pragma unroll
    for (int qq = 0; qq < kRedThreads / 64; ++qq) tot += red[qq][lane];
    out[k] = tot;
    q[k] = e2 * tot + prec[k] * pv[k];
  }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  return ensure_dyn_lds((const void *)kernel, bytes);
}

// more used columns than one LDS tile holds: the generic kernels (kernels_generic.hip)
bool beyond_lds(const obhip_terms &t) {
  return t.Mu > (uint64_t)kMaxMuLds || getenv("OBHIP_FORCE_GENERIC") != nullptr;  // env: tests
}

template <int W2, int MODE>
int run_mm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, uint64_t ld) {
  const size_t lds = (t.Mu * kTileRows + kMmWaves * kTileRows) * sizeof(double);
  OB_TRY(set_lds(k_mm<W2, MODE>, lds));
  hipLaunchKernelGGL((k_mm<W2, MODE>), dim3((unsigned)(b.n_pad / kTileRows)), dim3(kMmThreads), lds,
                     cur_stream(), b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc,
                     (const uint32_t *)t.cols.p, (int)(t.W / 2), (int)t.p, d_a, d_out, b.n, ld);
  OB_HIP(hipGetLastError());
  return 0;
}

template <int MODE>
int dispatch_mm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out,
                uint64_t ld = 0) {
  switch (t.W / 2) {
    case 1: return run_mm<1, MODE>(b, t, d_a, d_out, ld);
    case 2: return run_mm<2, MODE>(b, t, d_a, d_out, ld);
    case 3: return run_mm<3, MODE>(b, t, d_a, d_out, ld);
    case 4: return run_mm<4, MODE>(b, t, d_a, d_out, ld);
    default: return run_mm<0, MODE>(b, t, d_a, d_out, ld);
  }
}

template <int W2, bool SQ, bool PF>
int run_tmm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *part, dim3 grid,
            uint64_t ntiles, uint64_t tps) {
  const size_t lds = t.Mu * kTileRows * sizeof(double) + t.Mu * sizeof(int);
  OB_TRY(set_lds(k_tmm<W2, SQ, PF>, lds));
  hipLaunchKernelGGL((k_tmm<W2, SQ, PF>), grid, dim3(256), lds, cur_stream(), b.bm.p, b.scale.p,
                     t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, (int)(t.W / 2), d_a,
                     b.n, ntiles, tps, t.p_pad, part);
  OB_HIP(hipGetLastError());
  return 0;
}

template <bool SQ>
int dispatch_tmm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *part, dim3 grid,
                 uint64_t ntiles, uint64_t tps) {
  const bool pf = t.Mu <= 4 * (uint64_t)kTmmPre;
  switch (t.W / 2) {
    case 1: return pf ? run_tmm<1, SQ, true>(b, t, d_a, part, grid, ntiles, tps)
                      : run_tmm<1, SQ, false>(b, t, d_a, part, grid, ntiles, tps);
    case 2: return pf ? run_tmm<2, SQ, true>(b, t, d_a, part, grid, ntiles, tps)
                      : run_tmm<2, SQ, false>(b, t, d_a, part, grid, ntiles, tps);
    case 3: return pf ? run_tmm<3, SQ, true>(b, t, d_a, part, grid, ntiles, tps)
                      : run_tmm<3, SQ, false>(b, t, d_a, part, grid, ntiles, tps);
    case 4: return pf ? run_tmm<4, SQ, true>(b, t, d_a, part, grid, ntiles, tps)
                      : run_tmm<4, SQ, false>(b, t, d_a, part, grid, ntiles, tps);
    default: return run_tmm<0, SQ, false>(b, t, d_a, part, grid, ntiles, tps);
  }
}

}  // namespace

// d_out: column-major n x p with leading dimension ld (0: n)
int launch_getmat(const obhip_basis &b, obhip_terms &t, double *d_out, uint64_t ld) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (ld == 0) ld = b.n;
  ProfScope ps("getmat");
  if (beyond_lds(t)) return launch_mm_generic(b, t, nullptr, d_out, 2, ld);
  return dispatch_mm<2>(b, t, nullptr, d_out, ld);
}


template <int W2, bool SQ, int NG>
int run_mm_tl(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, double *part,
              dim3 grid, uint64_t ntiles, uint64_t tps) {
  const size_t lds = (t.Mu * kTlPitch + kTlWaves * kTileRows) * sizeof(double);
  const bool pf = t.Mu <= (uint64_t)kTlWaves * kTlPre;
  if (pf) {
    OB_TRY(set_lds(k_mm_tl<W2, SQ, NG, true>, lds));
    hipLaunchKernelGGL((k_mm_tl<W2, SQ, NG, true>), grid, dim3(kTlThreads), lds, cur_stream(), b.bm.p,
                       b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, t.sperm.p,
                       d_a, (int)t.p, b.n, b.n_pad, ntiles, tps, t.p_pad, d_out, part);
  } else {
    OB_TRY(set_lds(k_mm_tl<W2, SQ, NG, false>, lds));
    hipLaunchKernelGGL((k_mm_tl<W2, SQ, NG, false>), grid, dim3(kTlThreads), lds, cur_stream(), b.bm.p,
                       b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, t.sperm.p,
                       d_a, (int)t.p, b.n, b.n_pad, ntiles, tps, t.p_pad, d_out, part);
  }
  OB_HIP(hipGetLastError());
  return 0;
}

template <bool SQ>
int dispatch_mm_tl(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out,
                   double *part, dim3 grid, int ng, uint64_t ntiles, uint64_t tps) {
#define OB_ML(W2_, NG_) return run_mm_tl<W2_, SQ, NG_>(b, t, d_a, d_out, part, grid, ntiles, tps)
  switch (t.W / 2) {
    case 1: if (ng == 8) OB_ML(1, 8); if (ng == 4) OB_ML(1, 4); if (ng == 2) OB_ML(1, 2); OB_ML(1, 1);
    case 2: if (ng == 8) OB_ML(2, 8); if (ng == 4) OB_ML(2, 4); if (ng == 2) OB_ML(2, 2); OB_ML(2, 1);
    case 3: if (ng == 4) OB_ML(3, 4); if (ng == 2) OB_ML(3, 2); OB_ML(3, 1);
    default: if (ng == 4) OB_ML(4, 4); if (ng == 2) OB_ML(4, 2); OB_ML(4, 1);
  }
#undef OB_ML
}

// kernels_star.hip: the kernels on shared sub-products (stars of four terms), from 9 star-waves up
bool star_supports(const obhip_terms &t, bool one_block, bool dual = false);
int launch_star_hess(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca,
                     double cb, double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles,
                     uint64_t tps, const double *stop0, const double *stop1);
int launch_star_tmm(const obhip_basis &b, obhip_terms &t, const double *d_a, bool squared, double *part,
                    const double *d_a2, double *part2, unsigned nsplit, uint64_t ntiles, uint64_t tps);
int launch_star_mm(const obhip_basis &b, obhip_terms &t, const double *d_a, bool squared, double *d_out,
                   double *mpart, unsigned nsplit, uint64_t ntiles, uint64_t tps);

namespace {
bool hm3_wanted() {
  static const bool off = getenv("OBHIP_HM3") && atoi(getenv("OBHIP_HM3")) == 0;
  return !off;
}
// one workgroup per CU (two tiles of the used columns in LDS), a range of tiles each; the terms in
// workgroups of 16 star-waves
void star_grid(const obhip_basis &b, const obhip_terms &t, uint64_t &nsplit, uint64_t &pblocks, uint64_t &ntiles,
               uint64_t &tps) {
  ntiles = b.n_pad / kTileRows;
  pblocks = (t.sh.nsw_family + 15) / 16;
  nsplit = std::max<uint64_t>(1, (uint64_t)device_cus(b.device) / pblocks);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
  tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
}
}  // namespace

int mm_tl_supports(const obhip_terms &t) {
  const int w2 = (int)(t.W / 2);
  return w2 >= 1 && w2 <= kMaxW2 &&
         (t.Mu * kTlPitch + kTlWaves * kTileRows) * sizeof(double) <= 156 * 1024;
}

int launch_mm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, bool squared) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (beyond_lds(t)) {
    ProfScope ps(squared ? "sqmm" : "mm");
    return launch_mm_generic(b, t, d_a, d_out, squared ? 1 : 0, 0);
  }
  static const bool force_rows = getenv("OBHIP_MM_LANE_ROW") != nullptr;
  if (!force_rows && hm3_wanted() && star_supports(t, false)) {  // shared sub-products (kernels_star.hip)
    uint64_t nsplit, pblocks, ntiles, tps;
    star_grid(b, t, nsplit, pblocks, ntiles, tps);
    double *mpart = nullptr;
    if (pblocks > 1)
      OB_TRY(const_cast<obhip_basis &>(b).workspace(pblocks * b.n_pad * sizeof(double), (void **)&mpart));
    ProfScope ps(squared ? "sqmm" : "mm");
    OB_TRY(launch_star_mm(b, t, d_a, squared, d_out, mpart, (unsigned)nsplit, ntiles, tps));
    if (pblocks > 1) {
      const dim3 g2((unsigned)((b.n + 255) / 256));
      if (squared)
        hipLaunchKernelGGL(k_mm_tl_sum<true>, g2, dim3(256), 0, cur_stream(), mpart, (int)pblocks, b.n_pad,
                           b.scale.p, b.n, d_out);
      else
        hipLaunchKernelGGL(k_mm_tl_sum<false>, g2, dim3(256), 0, cur_stream(), mpart, (int)pblocks, b.n_pad,
                           b.scale.p, b.n, d_out);
      OB_HIP(hipGetLastError());
    }
    return 0;
  }
  if (!mm_tl_supports(t) || force_rows) {
    ProfScope ps(squared ? "sqmm" : "mm");
    return squared ? dispatch_mm<1>(b, t, d_a, d_out) : dispatch_mm<0>(b, t, d_a, d_out);
  }
  // term-per-lane kernel: a block of 8 waves x NG groups x 64 terms, as few blocks along p
  // as the register budget allows
  const int ngmax = t.W / 2 <= 2 ? 8 : 4;
  int ng = 1;
  while (ng < ngmax && (uint64_t)kTlWaves * ng * 64 < t.p_pad) ng *= 2;
  const uint64_t tpb = (uint64_t)kTlWaves * ng * 64;
  const uint64_t pblocks = (t.p_pad + tpb - 1) / tpb;
  const uint64_t ntiles = b.n_pad / kTileRows;
  uint64_t nsplit = std::max<uint64_t>(1, (uint64_t)device_cus(b.device) * 4 / pblocks);
  nsplit = std::min(nsplit, ntiles);
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  if (pblocks > 1)
    OB_TRY(const_cast<obhip_basis &>(b).workspace(pblocks * b.n_pad * sizeof(double), (void **)&part));
  const dim3 grid((unsigned)nsplit, (unsigned)pblocks);
  ProfScope ps(squared ? "sqmm" : "mm");
  if (squared)
    OB_TRY(dispatch_mm_tl<true>(b, t, d_a, d_out, part, grid, ng, ntiles, tps));
  else
    OB_TRY(dispatch_mm_tl<false>(b, t, d_a, d_out, part, grid, ng, ntiles, tps));
  if (pblocks > 1) {
    const dim3 g2((unsigned)((b.n + 255) / 256));
    if (squared)
      hipLaunchKernelGGL(k_mm_tl_sum<true>, g2, dim3(256), 0, cur_stream(), part, (int)pblocks,
                         b.n_pad, b.scale.p, b.n, d_out);
    else
      hipLaunchKernelGGL(k_mm_tl_sum<false>, g2, dim3(256), 0, cur_stream(), part, (int)pblocks,
                         b.n_pad, b.scale.p, b.n, d_out);
    OB_HIP(hipGetLastError());
  }
  return 0;
}

template <int W2, bool SQ, int NPAIR>
int run_tmm_tl(const obhip_basis &b, obhip_terms &t, const double *d_a, double *part, dim3 grid,
               uint64_t ntiles, uint64_t tps) {
  const size_t lds = t.Mu * kTlPitch * sizeof(double);
  static const bool nopf = getenv("OBHIP_TL_NOPREFETCH") != nullptr;
  const bool pf = !nopf && t.Mu <= (uint64_t)kTlWaves * kTlPre;
  if (pf) {
    OB_TRY(set_lds(k_tmm_tl<W2, SQ, NPAIR, true>, lds));
    hipLaunchKernelGGL((k_tmm_tl<W2, SQ, NPAIR, true>), grid, dim3(kTlThreads), lds, cur_stream(),
                       b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p,
                       t.sperm.p, d_a, b.n, ntiles, tps, t.p_pad, part);
  } else {
    OB_TRY(set_lds(k_tmm_tl<W2, SQ, NPAIR, false>, lds));
    hipLaunchKernelGGL((k_tmm_tl<W2, SQ, NPAIR, false>), grid, dim3(kTlThreads), lds, cur_stream(),
                       b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p,
                       t.sperm.p, d_a, b.n, ntiles, tps, t.p_pad, part);
  }
  OB_HIP(hipGetLastError());
  return 0;
}

// terms per block of k_tmm_tl: 8 waves x NPAIR x 2 groups of 64
template <bool SQ>
int dispatch_tmm_tl(const obhip_basis &b, obhip_terms &t, const double *d_a, double *part,
                    unsigned nsplit, int npair, uint64_t ntiles, uint64_t tps) {
  const uint64_t tpb = (uint64_t)kTlWaves * npair * kTlGP * 64;
  const dim3 grid(nsplit, (unsigned)((t.p_pad + tpb - 1) / tpb));
#define OB_TL(W2_, NP_) return run_tmm_tl<W2_, SQ, NP_>(b, t, d_a, part, grid, ntiles, tps)
  switch (t.W / 2) {
    case 1: if (npair == 4) OB_TL(1, 4); if (npair == 2) OB_TL(1, 2); OB_TL(1, 1);
    case 2: if (npair == 4) OB_TL(2, 4); if (npair == 2) OB_TL(2, 2); OB_TL(2, 1);
    case 3: if (npair == 2) OB_TL(3, 2); OB_TL(3, 1);
    default: if (npair == 2) OB_TL(4, 2); OB_TL(4, 1);
  }
#undef OB_TL
}

int tmm_tl_supports(const obhip_terms &t) {
  const int w2 = (int)(t.W / 2);
  return w2 >= 1 && w2 <= kMaxW2 && t.Mu * kTlPitch * sizeof(double) <= 156 * 1024;
}

template <int W2, int NPAIR, bool PF, bool WY>
int run_materialize_tl2(const obhip_basis &b, obhip_terms &t, double *d_B, dim3 grid, uint64_t ntiles,
                        uint64_t tps, const double *d_y, double *ypart) {
  const size_t lds = t.Mu * kTlPitch * sizeof(double);
  OB_TRY(set_lds(k_materialize_tl<W2, NPAIR, PF, WY>, lds));
  hipLaunchKernelGGL((k_materialize_tl<W2, NPAIR, PF, WY>), grid, dim3(kTlThreads), lds, cur_stream(),
                     b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, b.n,
                     ntiles, tps, t.p_pad, d_B, d_y, ypart);
  OB_HIP(hipGetLastError());
  return 0;
}
template <int W2, int NPAIR>
int run_materialize_tl(const obhip_basis &b, obhip_terms &t, double *d_B, dim3 grid, uint64_t ntiles,
                       uint64_t tps, const double *d_y, double *ypart) {
  const bool pf = t.Mu <= (uint64_t)kTlWaves * kTlPre;
  if (pf && d_y) return run_materialize_tl2<W2, NPAIR, true, true>(b, t, d_B, grid, ntiles, tps, d_y, ypart);
  if (pf) return run_materialize_tl2<W2, NPAIR, true, false>(b, t, d_B, grid, ntiles, tps, d_y, ypart);
  if (d_y) return run_materialize_tl2<W2, NPAIR, false, true>(b, t, d_B, grid, ntiles, tps, d_y, ypart);
  return run_materialize_tl2<W2, NPAIR, false, false>(b, t, d_B, grid, ntiles, tps, d_y, ypart);
}

bool materialize_tl_supports(const obhip_terms &t) { return tmm_tl_supports(t) != 0; }

// d_B: n_pad x p_pad doubles, row-major; t prepared by the caller.  d_y (n, may be null): also
// d_g (p) = B^T y, from the products the copy forms anyway.
int launch_materialize_tl(const obhip_basis &b, obhip_terms &t, double *d_B, const double *d_y,
                          double *d_g) {
  const int npmax = t.W / 2 <= 2 ? 4 : 2;
  int npair = 1;
  while (npair < npmax && (uint64_t)kTlWaves * npair * 128 < t.p_pad) npair *= 2;
  const uint64_t tpb = (uint64_t)kTlWaves * npair * 128;
  const uint64_t pblocks = (t.p_pad + tpb - 1) / tpb;
  const uint64_t ntiles = b.n_pad / kTileRows;
  uint64_t nsplit = std::max<uint64_t>(1, (uint64_t)device_cus(b.device) * 4 / pblocks);
  nsplit = std::min(nsplit, ntiles);
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  const dim3 grid((unsigned)nsplit, (unsigned)pblocks);
  double *ypart = nullptr;
  if (d_y) OB_TRY(const_cast<obhip_basis &>(b).workspace(nsplit * t.p_pad * sizeof(double), (void **)&ypart));
  int rc = 0;
#define OB_MT(W2_, NP_) rc = run_materialize_tl<W2_, NP_>(b, t, d_B, grid, ntiles, tps, d_y, ypart); break
  switch ((int)(t.W / 2) * 8 + npair) {
    case 1 * 8 + 4: OB_MT(1, 4);
    case 1 * 8 + 2: OB_MT(1, 2);
    case 1 * 8 + 1: OB_MT(1, 1);
    case 2 * 8 + 4: OB_MT(2, 4);
    case 2 * 8 + 2: OB_MT(2, 2);
    case 2 * 8 + 1: OB_MT(2, 1);
    case 3 * 8 + 2: OB_MT(3, 2);
    case 3 * 8 + 1: OB_MT(3, 1);
    case 4 * 8 + 2: OB_MT(4, 2);
    default: rc = run_materialize_tl<4, 1>(b, t, d_B, grid, ntiles, tps, d_y, ypart); break;
  }
#undef OB_MT
  OB_TRY(rc);
  if (d_y) {
    hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0, cur_stream(),
                       ypart, (int)nsplit, t.p_pad, (int)t.p, d_g);
    OB_HIP(hipGetLastError());
  }
  return 0;
}

// d_out (p) = B^T a and d_out2 (p) = (B^2)^T a2 (a2 null: ones -> sqcolsums) in ONE pass of
// k_tmm_tl<DUAL>; kNotFused when the terms do not fit that instantiation (more than 6 factors): the caller then makes the two passes.
int launch_tmm_dual(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, const double *d_a2,
                    double *d_out2) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  static const bool off = getenv("OBHIP_TMM_DUAL") && atoi(getenv("OBHIP_TMM_DUAL")) == 0;
  const int w2 = (int)(t.W / 2);
  if (!off && !beyond_lds(t) && hm3_wanted() && star_supports(t, false, true)) {  // shared sub-products
    uint64_t nsplit, pblocks, ntiles, tps;
    star_grid(b, t, nsplit, pblocks, ntiles, tps);
    double *part = nullptr;
    OB_TRY(const_cast<obhip_basis &>(b).workspace(2 * nsplit * t.p_pad * sizeof(double), (void **)&part));
    double *part2 = part + nsplit * t.p_pad;
    {
      ProfScope ps("tmm_dual");
      OB_TRY(launch_star_tmm(b, t, d_a, false, part, d_a2, part2, (unsigned)nsplit, ntiles, tps));
    }
    hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0, cur_stream(), part,
                       (int)nsplit, t.p_pad, (int)t.p, d_out);
    hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0, cur_stream(), part2,
                       (int)nsplit, t.p_pad, (int)t.p, d_out2);
    OB_HIP(hipGetLastError());
    return 0;
  }
  if (off || beyond_lds(t) || !tmm_tl_supports(t) || w2 > 3) return kNotFused;
  const bool pf = t.Mu <= (uint64_t)kTlWaves * kTlPre;  // (else the tile is loaded between the barriers)
  const uint64_t ntiles = b.n_pad / kTileRows, p_pad = t.p_pad;
  // six-slot terms (obfit's eight-dimensional sets): 4 terms per lane at 123 VGPRs, 8 would spill
  const int npmax = w2 <= 2 ? 4 : 2;
  int npair = 1;
  while (npair < npmax && (uint64_t)kTlWaves * npair * kTlGP * 64 < p_pad) npair *= 2;
  const uint64_t tpb = (uint64_t)kTlWaves * npair * kTlGP * 64;
  const uint64_t pblocks = (p_pad + tpb - 1) / tpb;
  uint64_t nsplit = std::max<uint64_t>(1, (uint64_t)device_cus(b.device) * 2 / pblocks);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(const_cast<obhip_basis &>(b).workspace(2 * nsplit * p_pad * sizeof(double), (void **)&part));
  double *part2 = part + nsplit * p_pad;
  const dim3 grid((unsigned)nsplit, (unsigned)pblocks);
  const size_t lds = t.Mu * kTlPitch * sizeof(double);
  {
    ProfScope ps("tmm_dual");
#define OB_TD2(W2_, NP_, PF_)                                                                               \
  do {                                                                                                       \
    OB_TRY(set_lds(k_tmm_tl<W2_, false, NP_, PF_, true>, lds));                                              \
    hipLaunchKernelGGL((k_tmm_tl<W2_, false, NP_, PF_, true>), grid, dim3(kTlThreads), lds, cur_stream(),    \
                       b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p,          \
                       t.sperm.p, d_a, b.n, ntiles, tps, p_pad, part, d_a2, part2);                          \
  } while (0)
#define OB_TD(W2_, NP_)                                                                                      \
  do {                                                                                                       \
    if (pf)                                                                                                  \
      OB_TD2(W2_, NP_, true);                                                                                \
    else                                                                                                     \
      OB_TD2(W2_, NP_, false);                                                                               \
  } while (0)
    if (w2 == 1) {
      if (npair == 4) OB_TD(1, 4); else if (npair == 2) OB_TD(1, 2); else OB_TD(1, 1);
    } else if (w2 == 2) {
      if (npair == 4) OB_TD(2, 4); else if (npair == 2) OB_TD(2, 2); else OB_TD(2, 1);
    } else {
      if (npair == 2) OB_TD(3, 2); else OB_TD(3, 1);
    }
#undef OB_TD2
#undef OB_TD
    OB_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0, cur_stream(), part,
                     (int)nsplit, p_pad, (int)t.p, d_out);
  hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0, cur_stream(), part2,
                     (int)nsplit, p_pad, (int)t.p, d_out2);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_tmm(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, bool squared) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (beyond_lds(t)) {
    ProfScope ps(squared ? "sqtmm" : "tmm");
    return launch_tmm_generic(b, t, d_a, d_out, squared);
  }
  const uint64_t ntiles = b.n_pad / kTileRows;
  const uint64_t p_pad = t.p_pad;  // multiple of 256 (obhip_terms::prepare)
  double *part = nullptr;
  static const bool force_rows = getenv("OBHIP_TMM_LANE_ROW") != nullptr;
  if (!force_rows && hm3_wanted() && star_supports(t, false)) {  // shared sub-products (kernels_star.hip)
    uint64_t nsplit, pblocks, nt, tps;
    star_grid(b, t, nsplit, pblocks, nt, tps);
    OB_TRY(const_cast<obhip_basis &>(b).workspace(nsplit * p_pad * sizeof(double), (void **)&part));
    {
      ProfScope ps(squared ? "sqtmm" : "tmm");
      OB_TRY(launch_star_tmm(b, t, d_a, squared, part, nullptr, nullptr, (unsigned)nsplit, nt, tps));
    }
    hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0,
                       cur_stream(), part, (int)nsplit, p_pad, (int)t.p, d_out);
    OB_HIP(hipGetLastError());
    return 0;
  }
  if (tmm_tl_supports(t) && !force_rows) {
    // term-per-lane kernel: the fewest blocks along p that the register budget allows
    const int npmax = t.W / 2 <= 2 ? 4 : 2;
    int npair = 1;
    while (npair < npmax && (uint64_t)kTlWaves * npair * kTlGP * 64 < p_pad) npair *= 2;
    const uint64_t tpb = (uint64_t)kTlWaves * npair * kTlGP * 64;
    const uint64_t pblocks = (p_pad + tpb - 1) / tpb;
    // two resident blocks per CU, one round
    uint64_t nsplit = std::max<uint64_t>(1, (uint64_t)device_cus(b.device) * 2 / pblocks);
    nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
    const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + tps - 1) / tps;
    OB_TRY(const_cast<obhip_basis &>(b).workspace(nsplit * p_pad * sizeof(double), (void **)&part));
    {
      ProfScope ps(squared ? "sqtmm" : "tmm");
      if (squared)
        OB_TRY(dispatch_tmm_tl<true>(b, t, d_a, part, (unsigned)nsplit, npair, ntiles, tps));
      else
        OB_TRY(dispatch_tmm_tl<false>(b, t, d_a, part, (unsigned)nsplit, npair, ntiles, tps));
    }
    hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0,
                       cur_stream(), part, (int)nsplit, p_pad, (int)t.p, d_out);
    OB_HIP(hipGetLastError());
    return 0;
  }
  const uint64_t pblocks = (t.p + 255) / 256;
  // enough blocks to fill 256 CUs a few times over, each with >= 4 tiles
  uint64_t nsplit = std::max<uint64_t>(1, (256 * 6 + pblocks - 1) / pblocks);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  OB_TRY(const_cast<obhip_basis &>(b).workspace(nsplit * p_pad * sizeof(double), (void **)&part));
  const dim3 grid((unsigned)nsplit, (unsigned)pblocks);
  {
    ProfScope ps(squared ? "sqtmm" : "tmm");
    if (squared)
      OB_TRY(dispatch_tmm<true>(b, t, d_a, part, grid, ntiles, tps));
    else
      OB_TRY(dispatch_tmm<false>(b, t, d_a, part, grid, ntiles, tps));
  }
  hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0,
                     cur_stream(), part, (int)nsplit, p_pad, (int)t.p, d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

// d_out (p) = B^T (c_a B a + c_b y) in one pass over the basis (k_hm_tl); d_yhat (n, may be
// null) = B a; d_ss (1, may be null) = sum (B a - y)^2.  Returns 1 when the terms do not fit the
// fused kernel (more terms than one block holds, wide terms, too many used columns): the caller
// then takes the two-kernel path.
template <int W2, int NU, bool PF, bool RO>
int run_hm_tl2(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca,
               double cb, double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles,
               uint64_t tps, size_t lds) {
  OB_TRY(set_lds(k_hm_tl<W2, NU, PF, RO>, lds));
  hipLaunchKernelGGL((k_hm_tl<W2, NU, PF, RO>), dim3(nsplit), dim3(kTlThreads), lds, cur_stream(), b.bm.p,
                     b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p, t.sperm.p, d_a,
                     (int)t.p, d_y, ca, cb, b.n, ntiles, tps, t.p_pad, part, d_yhat, sspart);
  OB_HIP(hipGetLastError());
  return 0;
}
template <int W2, int NU>
int run_hm_tl(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca,
              double cb, double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles,
              uint64_t tps) {
  const size_t lds = (t.Mu * kTlPitch + 2 * kTlWaves * kHmChunk) * sizeof(double);
  static const bool nopf = getenv("OBHIP_HM_NOPREFETCH") != nullptr;
  const bool pf = !nopf && t.Mu <= (uint64_t)kTlWaves * kTlPre;
  const bool ro = d_y != nullptr;
  if (pf && ro) return run_hm_tl2<W2, NU, true, true>(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, nsplit, ntiles, tps, lds);
  if (pf) return run_hm_tl2<W2, NU, true, false>(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, nsplit, ntiles, tps, lds);
  if (ro) return run_hm_tl2<W2, NU, false, true>(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, nsplit, ntiles, tps, lds);
  return run_hm_tl2<W2, NU, false, false>(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, nsplit, ntiles, tps, lds);
}

// kernels_hm.hip: the second-generation kernel (two tile buffers fed by LDS-direct loads, four
// waves per SIMD) and the terms it takes
bool hm2_supports(const obhip_terms &t, bool ro, int variant);
int launch_hm2(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca,
               double cb, double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles,
               uint64_t tps, int variant, const double *stop0, const double *stop1);

namespace {
bool hm2_wanted() {
  static const bool off = getenv("OBHIP_HESSMULT_FUSED") && atoi(getenv("OBHIP_HESSMULT_FUSED")) == 0;
  static const bool v1 = getenv("OBHIP_HM_V1") && atoi(getenv("OBHIP_HM_V1")) != 0;
  return !off && !v1;
}
int hm2_variant() {
  static const int variant = getenv("OBHIP_HM2_VARIANT") ? atoi(getenv("OBHIP_HM2_VARIANT")) : 0;
  return variant;
}
}  // namespace

// whether a Hessian product of these terms can be enqueued speculatively (it takes k_hm2, which
// honours the stop flags of launch_hessmult_fused)
bool hessmult_fused_skippable(const obhip_basis &b, obhip_terms &t) {
  if (t.prepare(b.md.cap, b.md.dims_h) != 0) return false;
  if (!hm2_wanted() || beyond_lds(t)) return false;
  return (hm3_wanted() && hm2_variant() == 0 && star_supports(t, true)) || hm2_supports(t, false, hm2_variant());
}

int launch_hessmult_fused(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y,
                          double ca, double cb, double *d_out, double *d_yhat, double *d_ss,
                          const double *d_stop0, const double *d_stop1, const HmThen *then) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  static const bool off = getenv("OBHIP_HESSMULT_FUSED") && atoi(getenv("OBHIP_HESSMULT_FUSED")) == 0;
  // OBHIP_HM_V1=1: the round-3 kernel (A/B runs); OBHIP_HM2_VARIANT: block shapes of k_hm2
  const int variant = hm2_variant();
  const int w2 = (int)(t.W / 2);
  // OBHIP_HM3=0: k_hm2 instead of the two-phase kernel on shared sub-products (A/B)
  const bool use3 = hm2_wanted() && hm3_wanted() && variant == 0 && !beyond_lds(t) && star_supports(t, true);
  const bool use2 = use3 || (hm2_wanted() && !beyond_lds(t) && hm2_supports(t, d_y != nullptr, variant));
  if (d_stop0 && !use2) return fail(OBHIP_ERR_STATE, "hessmult: stop flags need the k_hm2 / k_hm3 path");
  // (8 terms of 6 factors per lane spill and run at half the speed of the two-kernel form: measured)
  const int numax = w2 <= 2 ? 8 : 4;
  if (!use2 &&
      (off || beyond_lds(t) || w2 < 1 || w2 > kMaxW2 || t.p_pad > (uint64_t)kTlWaves * numax * 64 ||
       (t.Mu * kTlPitch + 2 * kTlWaves * kHmChunk) * sizeof(double) > 156 * 1024))
    return kNotFused;
  int nu = 1;
  while ((uint64_t)kTlWaves * nu * 64 < t.p_pad) nu *= 2;
  const uint64_t ntiles = b.n_pad / kTileRows;
  // one resident block per CU (the products of a chunk live in registers), one round
  uint64_t nsplit = std::max<uint64_t>(1, (uint64_t)device_cus(b.device));
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(const_cast<obhip_basis &>(b).workspace((nsplit * t.p_pad + nsplit) * sizeof(double), (void **)&part));
  if ((d_yhat || d_ss) && !d_y) return fail(OBHIP_ERR_INVALID, "hessmult: yhat / residual sum need y");
  double *sspart = d_ss ? part + nsplit * t.p_pad : nullptr;
  if (use3) {
    ProfScope ps("hessmult");
    OB_TRY(launch_star_hess(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, (unsigned)nsplit, ntiles, tps,
                            d_stop0, d_stop1));
  } else if (use2) {
    ProfScope ps("hessmult");
    OB_TRY(launch_hm2(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, (unsigned)nsplit, ntiles, tps, variant,
                      d_stop0, d_stop1));
  } else {
    ProfScope ps("hessmult");
#define OB_HM(W2_, NU_) OB_TRY((run_hm_tl<W2_, NU_>(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, (unsigned)nsplit, ntiles, tps))); break
    switch (w2 * 16 + nu) {
      case 1 * 16 + 1: OB_HM(1, 1);
      case 1 * 16 + 2: OB_HM(1, 2);
      case 1 * 16 + 4: OB_HM(1, 4);
      case 1 * 16 + 8: OB_HM(1, 8);
      case 2 * 16 + 1: OB_HM(2, 1);
      case 2 * 16 + 2: OB_HM(2, 2);
      case 2 * 16 + 4: OB_HM(2, 4);
      case 2 * 16 + 8: OB_HM(2, 8);
      case 3 * 16 + 1: OB_HM(3, 1);
      case 3 * 16 + 2: OB_HM(3, 2);
      case 3 * 16 + 4: OB_HM(3, 4);
      case 4 * 16 + 1: OB_HM(4, 1);
      case 4 * 16 + 2: OB_HM(4, 2);
      case 4 * 16 + 4: OB_HM(4, 4);
      default: return kNotFused;
    }
#undef OB_HM
  }
  if (then)
    hipLaunchKernelGGL(k_tmm_reduce_q, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0, cur_stream(),
                       part, (int)nsplit, t.p_pad, (int)t.p, d_out, then->e2, then->prec, then->pv, then->q);
  else
    hipLaunchKernelGGL(k_tmm_reduce, dim3((unsigned)((t.p + 63) / 64)), dim3(kRedThreads), 0, cur_stream(),
                       part, (int)nsplit, t.p_pad, (int)t.p, d_out);
  if (d_ss) hipLaunchKernelGGL(k_hm_ss, dim3(1), dim3(64), 0, cur_stream(), sspart, (int)nsplit, d_ss);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
