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
          const double *__restrict__ rot, const int *__restrict__ cpos, int d, int Mu,
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
    sc *= build_dim_any(D, ka, kb, kc, rot, xv, store);
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

template <int W2, bool VAR>
int run_predict(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x,
                uint64_t n, double *d_mean, const double *d_coeffvar, double e2sigma, double *d_var) {
  const size_t lds = (t.Mu * kTileRows + 2 * kPrWaves * kTileRows) * sizeof(double);
  if (lds > 64 * 1024)
    OB_HIP(hipFuncSetAttribute((const void *)k_predict<W2, VAR>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_predict<W2, VAR>), dim3((unsigned)((n + kTileRows - 1) / kTileRows)),
                     dim3(kPrThreads), lds, cur_stream(), t.pred_md.dims.p, t.pred_md.ka.p, t.pred_md.kb.p,
                     t.pred_md.kc.p, t.pred_md.rot.p, t.cpos.p, (int)m.d, (int)t.Mu,
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

int launch_predict(const obhip_model &m, obhip_terms &t, const double *d_theta, const double *d_x,
                   uint64_t n, double *d_mean, const double *d_coeffvar, double e2sigma,
                   double *d_var) {
  if (t.pred_model != &m || t.pred_md.model_version != m.version) {
    OB_TRY(t.pred_md.build(m, t.maxlev));
    t.pred_model = &m;
  }
  OB_TRY(t.prepare(t.pred_md.cap, t.pred_md.dims_h));
  if (t.Mu > 296)
    return fail(OBHIP_ERR_INVALID, "terms touch too many basis columns for the LDS tile");
  if (n == 0) return 0;
  ProfScope ps("predict");
  if (d_coeffvar != nullptr && d_var != nullptr)
    return dispatch_predict<true>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
  return dispatch_predict<false>(m, t, d_theta, d_x, n, d_mean, d_coeffvar, e2sigma, d_var);
}

}  // namespace obhip
