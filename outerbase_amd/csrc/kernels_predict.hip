// Fused predictor for gfx950: basis evaluation at new inputs and the B theta
// (and B^2 coeffvar) contraction in one kernel; the basis tile only ever exists
// in LDS.  Replaces predictor$update + $mean + $var of pred_gauss
// (src/lpdfs/loglik_gauss.cpp:214-227: a fresh outerbase (modandbase.cpp:547)
// followed by prodmm_ (linalg.cpp:57-131) on basemat and on basematsq).
//
// Per 64-row tile: 8 waves evaluate the dimensions (wave w takes w, w+8, ...),
// lane = row, writing only the columns the terms use into the LDS tile; after
// one barrier the 8 waves split the terms in groups of 64 with register-resident
// term tables, exactly as k_mm does.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

// 8 waves per 64-row tile (see kMmThreads in kernels_prod.hip)
constexpr int kPrThreads = 512, kPrWaves = kPrThreads / 64;

template <int W2, bool VAR>
__global__ void __launch_bounds__(kPrThreads)
k_predict(const DimDesc *__restrict__ dims, const double *__restrict__ ka,
          const double *__restrict__ kb, const double *__restrict__ kc,
          const double *__restrict__ rot, const double *__restrict__ tab, const int *__restrict__ cpos, int d, int Mu,
          const uint32_t *__restrict__ colsw, int W2rt, int p, const double *__restrict__ theta,
          const double *__restrict__ coeffvar, double e2sigma, const double *__restrict__ x,
          uint64_t n, double *__restrict__ mean, double *__restrict__ var) {
  extern __shared__ double lds[];
  double *red = lds + (size_t)Mu * kTileRows;   // [kPrWaves][64] scale partials, then mean partials
  double *redv = red + kPrWaves * kTileRows;    // [kPrWaves][64] variance partials
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t row = (uint64_t)blockIdx.x * kTileRows + lane;
  const bool valid = row < n;

  double sc = 1.0;
  const StoreLds store{lds, cpos, lane};
  for (int l = wave; l < d; l += kPrWaves) {
    const DimDesc D = dims[l];
    const double xv = valid ? x[(uint64_t)l * n + row] : 0.5;
    sc *= build_dim_any(D, ka, kb, kc, rot, tab, xv, store);
  }
  if (wave == 0) lds[lane] = 1.0;  // used column 0 = all ones
  red[wave * kTileRows + lane] = sc;
  __syncthreads();
  double s = 1.0;
#pragma unroll
  for (int q = 0; q < kPrWaves; ++q) s *= red[q * kTileRows + lane];
  __syncthreads();

  double am = 0.0, av = 0.0;
  if constexpr (W2 > 0) {
    const int ngroups = (p + 63) / 64;
    uint32_t cw[W2];
    for (int g = wave; g < ngroups; g += kPrWaves) {
      const int k0 = g * 64, cnt = min(64, p - k0);
      const int kk = min(k0 + lane, p - 1);
      load_cw(cw, colsw, k0 + lane);
      const double th = theta[kk];
      const double cv = VAR ? coeffvar[kk] : 0.0;
      if (cnt == 64) {
#pragma unroll
        for (int t = 0; t < 64; ++t) {
          const double pr = term_prod_rl<W2>(lds, cw, t, lane, 1.0);
          am = fma(readlane_f64(th, t), pr, am);
          if (VAR) av = fma(readlane_f64(cv, t), pr * pr, av);
        }
      } else {
        for (int t = 0; t < cnt; ++t) {
          const double pr = term_prod_rl<W2>(lds, cw, t, lane, 1.0);
          am = fma(readlane_f64(th, t), pr, am);
          if (VAR) av = fma(readlane_f64(cv, t), pr * pr, av);
        }
      }
    }
  } else {
    for (int k = wave; k < p; k += kPrWaves) {
      const double pr = term_prod_mem(lds, colsw + (size_t)k * W2rt, W2rt, lane, 1.0);
      am = fma(theta[k], pr, am);
      if (VAR) av = fma(coeffvar[k], pr * pr, av);
    }
  }
  red[wave * kTileRows + lane] = am;
  if (VAR) redv[wave * kTileRows + lane] = av;
  __syncthreads();
  if (wave == 0 && valid) {
    double tm = 0.0, tv = 0.0;
#pragma unroll
    for (int q = 0; q < kPrWaves; ++q) {
      tm += red[q * kTileRows + lane];
      if (VAR) tv += redv[q * kTileRows + lane];
    }
    mean[row] = tm * s;
    if (VAR) var[row] = tv * (s * s) + e2sigma;  // loglik_gauss.cpp:224-225
  }
}

// ---- term-per-lane predictor -------------------------------------------------------------------
// Default.  Persistent blocks of 8 waves; per 64-row tile the waves first evaluate the
// dimensions (lane = row) into the [column][65] LDS tile, then switch to lane = term for the
// contraction exactly as k_mm_tl does: theta_k (and coeffvar_k) and the column addresses of
// NG terms per lane stay in registers, 8 rows of accumulators per chunk, permlane-swap
// butterfly + DPP for the sum over the lanes.  More than 8 * NG * 64 terms: several passes
// over the same LDS tile with the tables reloaded.
template <int W, int NG, bool VAR>
struct PrCtx {
  uint32_t ad[NG][W];
  double th[NG];
  double cv[VAR ? NG : 1];
  double acc[8];
  double accv[VAR ? 8 : 1];
  template <int RR>
  __device__ __forceinline__ void row() {}
  template <int RR, int UNIT>
  __device__ __forceinline__ void use(double v) {
    acc[RR] = fma(v, th[UNIT], acc[RR]);
    if constexpr (VAR) accv[RR] = fma(v * v, cv[UNIT], accv[RR]);
  }
};

// 8 accumulators x 64 lanes -> tile rows rc .. rc + 7 of red[wave][.]
__device__ __forceinline__ void pr_reduce8(const double (&acc)[8], double *__restrict__ redw, int rc,
                                           int lane) {
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

template <int W2, int NG, bool VAR>
__global__ void __launch_bounds__(kTlThreads, 4)
k_predict_tl(const DimDesc *__restrict__ dims, const double *__restrict__ ka,
             const double *__restrict__ kb, const double *__restrict__ kc,
             const double *__restrict__ rot, const double *__restrict__ tab, const int *__restrict__ cpos, int d, int Mu,
             const uint32_t *__restrict__ colsw, const uint32_t *__restrict__ sperm, int p,
             uint64_t p_pad, int npass,
             const double *__restrict__ theta, const double *__restrict__ coeffvar, double e2sigma,
             const double *__restrict__ x, uint64_t n, uint64_t ntiles, uint64_t tiles_per_split,
             double *__restrict__ mean, double *__restrict__ var) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2;
  constexpr int kInflight = NG * W >= 32 ? 8 : 12;
  double *red = lds + (size_t)Mu * kTlPitch;    // [8 waves][64 rows] mean partials
  double *redv = red + kTlWaves * kTileRows;    // [8][64] variance partials
  double *reds = redv + kTlWaves * kTileRows;   // [8][64] scale partials
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  PrCtx<W, NG, VAR> c;
  int we = W;  // column slots the terms of this wave (and pass) need
  // slots: NG * 64 consecutive ones of the sorted order per wave (tl_slot)
  auto load_terms = [&](int pass) {
    int nzmax = 1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint64_t slot = tl_slot<NG>((uint64_t)pass, wave, g, lane);
      const bool ok = slot < p_pad;
      const uint64_t k = ok ? sperm[slot] : 0;
      const bool real = ok && k < (uint64_t)p;
      c.th[g] = real ? theta[k] : 0.0;
      if constexpr (VAR) c.cv[g] = real ? coeffvar[k] : 0.0;
      uint32_t cw[W2];
#pragma unroll
      for (int w = 0; w < W2; ++w) {
        cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
        c.ad[g][2 * w] = (cw[w] & 0xffffu) * (kTlPitch * 8);
        c.ad[g][2 * w + 1] = (cw[w] >> 16) * (kTlPitch * 8);
      }
      nzmax = max(nzmax, tl_nnz<W2>(cw));
    }
    we = tl_variant<W>(wave_max_i32(nzmax));
  };
  if (npass == 1) load_terms(0);

  double s_cur = 0.0;  // wave 0: basescale of row = lane of the tile being contracted
  auto emit = [&](uint64_t tile, double s) {  // wave 0, lane = row
    const uint64_t row = tile * kTileRows + lane;
    double tm = 0.0, tv = 0.0;
#pragma unroll
    for (int q = 0; q < kTlWaves; ++q) {
      tm += red[q * kTileRows + lane];
      if (VAR) tv += redv[q * kTileRows + lane];
    }
    if (row < n) {
      mean[row] = tm * s;
      if (VAR) var[row] = tv * (s * s) + e2sigma;  // loglik_gauss.cpp:224-225
    }
  };

  const StoreLdsPitch store{lds, cpos, lane};
  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();  // the previous tile's partials are complete, its LDS tile is free
    if (tile > t0 && wave == 0) emit(tile - 1, s_cur);
    {  // basis at the new rows, lane = row
      const uint64_t row = tile * kTileRows + lane;
      const bool valid = row < n;
      double sc = 1.0;
      for (int l = wave; l < d; l += kTlWaves) {
        const DimDesc D = dims[l];
        const double xv = valid ? x[(uint64_t)l * n + row] : 0.5;
        sc *= build_dim_any(D, ka, kb, kc, rot, tab, xv, store);
      }
      if (wave == 0) lds[lane] = 1.0;  // used column 0 = all ones
      reds[wave * kTileRows + lane] = sc;
    }
    __syncthreads();  // tile built; wave 0 has read the previous partials
    if (wave == 0) {
      double s = 1.0;
#pragma unroll
      for (int q = 0; q < kTlWaves; ++q) s *= reds[q * kTileRows + lane];
      s_cur = s;
    }
#pragma unroll 1
    for (int rc = 0; rc < kTileRows; rc += 8) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        c.acc[r] = 0.0;
        if constexpr (VAR) c.accv[r] = 0.0;
      }
      for (int pass = 0; pass < npass; ++pass) {
        if (npass > 1) {
          load_terms(pass);
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < W; ++j) c.ad[g][j] += rc * 8;
        }
        tl_run_half<W, NG, 8, kInflight, 0>(c, we);
      }
      if (npass == 1) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int j = 0; j < W; ++j) {
            c.ad[g][j] += (rc + 8 < kTileRows) ? 8 * 8 : -(kTileRows - 8) * 8;
            asm volatile("" : "+v"(c.ad[g][j]));
          }
      }
      pr_reduce8(c.acc, red + wave * kTileRows, rc, lane);
      if constexpr (VAR) pr_reduce8(c.accv, redv + wave * kTileRows, rc, lane);
    }
  }
  __syncthreads();
  if (t0 < t1 && wave == 0) emit(t1 - 1, s_cur);
}

template <int W2, int NG, bool VAR>
int run_predict_tl(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x,
                   uint64_t n, double *d_mean, const double *d_coeffvar, double e2sigma,
                   double *d_var, int npass) {
  const size_t lds = (t.Mu * kTlPitch + 3 * kTlWaves * kTileRows) * sizeof(double);
  OB_TRY(ensure_dyn_lds((const void *)k_predict_tl<W2, NG, VAR>, lds));
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int ncu = device_cus(dev);
  const uint64_t ntiles = (n + kTileRows - 1) / kTileRows;
  uint64_t nsplit = std::min<uint64_t>(ntiles, (uint64_t)ncu * 4);
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  hipLaunchKernelGGL((k_predict_tl<W2, NG, VAR>), dim3((unsigned)nsplit), dim3(kTlThreads), lds,
                     cur_stream(), t.pred_md.dims.p, t.pred_md.ka.p, t.pred_md.kb.p, t.pred_md.kc.p,
                     t.pred_md.rot.p, t.pred_md.tab.p, t.cpos.p, (int)m.d, (int)t.Mu, (const uint32_t *)t.cols.p,
                     t.sperm.p, (int)t.p, t.p_pad, npass, d_theta, d_coeffvar, e2sigma, d_x, n, ntiles, tps,
                     d_mean, d_var);
  OB_HIP(hipGetLastError());
  return 0;
}

template <bool VAR>
int dispatch_predict_tl(const obhip_model &m, obhip_terms &t, const double *d_theta,
                        const double *d_x, uint64_t n, double *d_mean, const double *d_coeffvar,
                        double e2sigma, double *d_var) {
  // the variance form carries twice the accumulators and coefficients: half the terms per lane
  const int ngmax = (t.W / 2 <= 2 ? 8 : 4) / (VAR ? 2 : 1);
  int ng = 1;
  while (ng < ngmax && (uint64_t)kTlWaves * ng * 64 < t.p_pad) ng *= 2;
  const uint64_t tpb = (uint64_t)kTlWaves * ng * 64;
  const int npass = (int)((t.p_pad + tpb - 1) / tpb);
#define OB_PR(W2_, NG_) \
  return run_predict_tl<W2_, NG_, VAR>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var, npass)
  // (the larger NG only exist without the variance)
  switch (t.W / 2) {
    case 1:
      if constexpr (!VAR) if (ng == 8) OB_PR(1, 8);
      if (ng == 4) OB_PR(1, 4);
      if (ng == 2) OB_PR(1, 2);
      OB_PR(1, 1);
    case 2:
      if constexpr (!VAR) if (ng == 8) OB_PR(2, 8);
      if (ng == 4) OB_PR(2, 4);
      if (ng == 2) OB_PR(2, 2);
      OB_PR(2, 1);
    case 3:
      if constexpr (!VAR) if (ng == 4) OB_PR(3, 4);
      if (ng == 2) OB_PR(3, 2);
      OB_PR(3, 1);
    default:
      if constexpr (!VAR) if (ng == 4) OB_PR(4, 4);
      if (ng == 2) OB_PR(4, 2);
      OB_PR(4, 1);
  }
#undef OB_PR
}

template <int W2, bool VAR>
int run_predict(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x,
                uint64_t n, double *d_mean, const double *d_coeffvar, double e2sigma, double *d_var) {
  const size_t lds = (t.Mu * kTileRows + 2 * kPrWaves * kTileRows) * sizeof(double);
  OB_TRY(ensure_dyn_lds((const void *)k_predict<W2, VAR>, lds));
  hipLaunchKernelGGL((k_predict<W2, VAR>), dim3((unsigned)((n + kTileRows - 1) / kTileRows)),
                     dim3(kPrThreads), lds, cur_stream(), t.pred_md.dims.p, t.pred_md.ka.p, t.pred_md.kb.p,
                     t.pred_md.kc.p, t.pred_md.rot.p, t.pred_md.tab.p, t.cpos.p, (int)m.d, (int)t.Mu,
                     (const uint32_t *)t.cols.p, (int)(t.W / 2), (int)t.p, d_theta, d_coeffvar,
                     e2sigma, d_x, n, d_mean, d_var);
  OB_HIP(hipGetLastError());
  return 0;
}

template <bool VAR>
int dispatch_predict(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x,
                     uint64_t n, double *d_mean, const double *d_coeffvar, double e2sigma,
                     double *d_var) {
  switch (t.W / 2) {
    case 1: return run_predict<1, VAR>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
    case 2: return run_predict<2, VAR>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
    case 3: return run_predict<3, VAR>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
    case 4: return run_predict<4, VAR>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
    default: return run_predict<0, VAR>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
  }
}

}  // namespace

bool star_predict_supports(const obhip_terms &t);
int launch_star_predict(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x, uint64_t n,
                        double *d_mean, const double *d_coeffvar, double e2sigma, double *d_var);

int launch_predict(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x,
                   uint64_t n, double *d_mean, const double *d_coeffvar, double e2sigma,
                   double *d_var) {
  if (t.pred_model != &m || t.pred_md.model_version != m.version) {
    OB_TRY(t.pred_md.build(m, t.maxlev));
    t.pred_model = &m;
  }
  OB_TRY(t.prepare(t.pred_md.cap, t.pred_md.dims_h));
  if (n == 0) return 0;
  ProfScope ps("predict");
  if (t.Mu > 296 || getenv("OBHIP_FORCE_GENERIC")) {
    // more used columns than the fused kernel's LDS tile holds: evaluate the basis at the new
    // rows into HBM (levels the terms use only) and take the products from there
    obhip_basis *bn = nullptr;
    OB_TRY(obhip_basis_create_dev(&bn, &m, d_x, n, t.maxlev.data()));
    int rc = launch_mm(*bn, t, d_theta, d_mean, false);
    if (!rc && d_coeffvar != nullptr && d_var != nullptr) {
      rc = launch_mm(*bn, t, d_coeffvar, d_var, true);
      if (!rc) rc = launch_affine(d_var, n, -e2sigma, 1.0);
    }
    obhip_basis_destroy(bn);  // synchronises the stream first
    return rc;
  }
  static const bool lane_row = getenv("OBHIP_PREDICT_LANE_ROW") != nullptr;
  static const bool no_star = getenv("OBHIP_HM3") && atoi(getenv("OBHIP_HM3")) == 0;
  if (!lane_row && !no_star && star_predict_supports(t))  // shared sub-products (kernels_star.hip)
    return launch_star_predict(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
  const int w2 = (int)(t.W / 2);
  if (!lane_row && w2 >= 1 && w2 <= 4 &&
      (t.Mu * kTlPitch + 3 * kTlWaves * kTileRows) * sizeof(double) <= 156 * 1024) {
    if (d_coeffvar != nullptr && d_var != nullptr)
      return dispatch_predict_tl<true>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
    return dispatch_predict_tl<false>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
  }
  if (d_coeffvar != nullptr && d_var != nullptr)
    return dispatch_predict<true>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
  return dispatch_predict<false>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
}

}  // namespace obhip
