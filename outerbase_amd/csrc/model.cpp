// Host-side model logic of libobhip: covariance functions on knot-sized
// inputs, the per-dimension eigen-model (outermod::build), term selection and
// term variances.  These are O(d m^3) / O(p^2) scalar problems that the
// reference also runs on the host; the data-parallel work lives in the
// kernels_*.hip files.
//
// Reference behaviour followed (not copied): src/covfuncs.cpp:35-50,113-126,
// 197-212,285-310; src/modandbase.cpp:128-276,350-356,387-440;
// src/interfaceR.cpp:53-149.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <unordered_set>

#include "obhip_internal.h"

namespace obhip {

// ---- covariance metadata ------------------------------------------------------
static const CovInfo kCovInfo[kNumCov] = {
    // mat25: covfuncs.cpp:87-111
    {1, {0, 0}, {-2.25, 0}, {1.5, 0}, {0.1, 1}, 0.0, 1.0},
    // mat25pow: covfuncs.cpp:166-195
    {2, {0, 0}, {-2.25, -1.25}, {1.5, 1.25}, {0.1, 0.01}, 0.0, 1.0},
    // mat25ang: covfuncs.cpp:254-283
    {2, {0, 0}, {-2.25, -2.25}, {1.5, 1.5}, {0.1, 0.1}, 0.0, 6.283185},
};

const CovInfo &cov_info(int kind) { return kCovInfo[kind]; }

static inline double mat25_of_h(double h) {
  return (1.0 + h + h * h / 3.0) * std::exp(-h);
}

void cov_host(int kind, const double *hyp, const double *x1, uint64_t n1,
              const double *x2, uint64_t n2, double *out) {
  const double a = 2.0, b = 0.25;  // covfuncs.h:42,53-54,66
  std::vector<double> t1(n1), t2(n2), u1, u2;
  if (kind == OBHIP_COV_MAT25) {
    const double ls = std::exp(a * hyp[0]);
    for (uint64_t i = 0; i < n1; ++i) t1[i] = x1[i] / ls;
    for (uint64_t j = 0; j < n2; ++j) t2[j] = x2[j] / ls;
  } else if (kind == OBHIP_COV_MAT25POW) {
    const double powv = std::exp(b * hyp[1]);
    const double ls = std::exp(a * hyp[0] + b * hyp[1]);
    for (uint64_t i = 0; i < n1; ++i) t1[i] = std::pow(x1[i], powv) / ls;
    for (uint64_t j = 0; j < n2; ++j) t2[j] = std::pow(x2[j], powv) / ls;
  } else {
    const double lss = std::exp(a * hyp[0]), lsc = std::exp(a * hyp[1]);
    u1.resize(n1);
    u2.resize(n2);
    for (uint64_t i = 0; i < n1; ++i) {
      t1[i] = std::sin(x1[i]) / lss;
      u1[i] = std::cos(x1[i]) / lsc;
    }
    for (uint64_t j = 0; j < n2; ++j) {
      t2[j] = std::sin(x2[j]) / lss;
      u2[j] = std::cos(x2[j]) / lsc;
    }
  }
  for (uint64_t j = 0; j < n2; ++j)
    for (uint64_t i = 0; i < n1; ++i) {
      double h;
      if (kind == OBHIP_COV_MAT25ANG) {
        const double hs = t1[i] - t2[j], hc = u1[i] - u2[j];
        h = std::sqrt(hs * hs + hc * hc);
      } else {
        h = std::fabs(t1[i] - t2[j]);
      }
      out[j * n1 + i] = mat25_of_h(h);
    }
}

// covf_*::cov_gradhyp (covfuncs.cpp:134-150, 220-243, 318-347)
void cov_gradhyp_host(int kind, const double *hyp, const double *x1, uint64_t n1, const double *x2,
                      uint64_t n2, double *out) {
  const double a = 2.0, b = 0.25;
  const uint64_t sl = n1 * n2;
  if (kind == OBHIP_COV_MAT25ANG) {
    const double lss = std::exp(a * hyp[0]), lsc = std::exp(a * hyp[1]);
    for (uint64_t j = 0; j < n2; ++j)
      for (uint64_t i = 0; i < n1; ++i) {
        const double hs = std::sin(x1[i]) / lss - std::sin(x2[j]) / lss;
        const double hc = std::cos(x1[i]) / lsc - std::cos(x2[j]) / lsc;
        const double h = std::sqrt(hs * hs + hc * hc);
        const double w = std::exp(-h) * (h + 1.0);
        out[j * n1 + i] = a / 3 * hs * hs * w;
        out[sl + j * n1 + i] = a / 3 * hc * hc * w;
      }
    return;
  }
  const bool pw = kind == OBHIP_COV_MAT25POW;
  const double powv = pw ? std::exp(b * hyp[1]) : 1.0;
  const double ls = pw ? std::exp(a * hyp[0] + b * hyp[1]) : std::exp(a * hyp[0]);
  for (uint64_t j = 0; j < n2; ++j) {
    const double t2 = (pw ? std::pow(x2[j], powv) : x2[j]) / ls;
    for (uint64_t i = 0; i < n1; ++i) {
      const double t1 = (pw ? std::pow(x1[i], powv) : x1[i]) / ls;
      const double h = t1 - t2;
      const double h2 = h * (1.0 + std::fabs(h)) * std::exp(-std::fabs(h));
      out[j * n1 + i] = a / 3 * (h * h2);
      if (pw) {
        const double g = std::log(x1[i]) * t1 - std::log(x2[j]) * t2;
        out[sl + j * n1 + i] = g * (-(b * powv / 3) * h2) + b / 3 * (h * h2);
      }
    }
  }
}

double cov_hyplpdf_host(int kind, const double *hyp) {
  const CovInfo &ci = kCovInfo[kind];
  double out = 0;
  for (int l = 0; l < ci.numhyp; ++l) {
    if (ci.hypub[l] < hyp[l]) return -std::numeric_limits<double>::infinity();
    if (ci.hyplb[l] > hyp[l]) return -std::numeric_limits<double>::infinity();
    out += 5 * std::log(ci.hypub[l] - hyp[l]);
    out += 5 * std::log(hyp[l] - ci.hyplb[l]);
  }
  for (int l = 0; l < ci.numhyp; ++l) {
    const double r = hyp[l] - ci.hyp0[l];
    out -= 0.5 * r * r / ci.hypvar[l];
  }
  return out;
}

// ---- cyclic Jacobi ------------------------------------------------------------
// One-sided accuracy is not needed here; the classical two-sided rotation with
// the stable tangent formula gives eigenvalues of the SPD knot covariances to
// high relative accuracy, which matters because levels are usable down to
// lambda_j / lambda_0 ~ 1e-11 (modandbase.cpp:245-248).
void jacobi_eigh(int n, std::vector<double> &a, std::vector<double> &w,
                 std::vector<double> &v) {
  v.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
  auto A = [&](int i, int j) -> double & { return a[(size_t)j * n + i]; };
  auto V = [&](int i, int j) -> double & { return v[(size_t)j * n + i]; };
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0;
    for (int q = 1; q < n; ++q)
      for (int p = 0; p < q; ++p) off += A(p, q) * A(p, q);
    if (off == 0.0) break;
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A(p, q);
        if (apq == 0.0) continue;
        const double app = A(p, p), aqq = A(q, q);
        // skip rotations that cannot change either diagonal entry
        if (std::fabs(apq) <= 1e-300 ||
            std::fabs(apq) < 1e-19 * std::sqrt(std::fabs(app * aqq)) ) {
          A(p, q) = A(q, p) = 0.0;
          continue;
        }
        rotated = true;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) /
                         (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        const double tau = s / (1.0 + c);
        A(p, p) = app - t * apq;
        A(q, q) = aqq + t * apq;
        A(p, q) = A(q, p) = 0.0;
        for (int r = 0; r < n; ++r) {
          if (r != p && r != q) {
            const double arp = A(r, p), arq = A(r, q);
            const double nrp = arp - s * (arq + tau * arp);
            const double nrq = arq + s * (arp - tau * arq);
            A(r, p) = A(p, r) = nrp;
            A(r, q) = A(q, r) = nrq;
          }
          const double vrp = V(r, p), vrq = V(r, q);
          V(r, p) = vrp - s * (vrq + tau * vrp);
          V(r, q) = vrq + s * (vrp - tau * vrq);
        }
      }
    if (!rotated) break;
  }
  // sort ascending
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(),
            [&](int x, int y) { return A(x, x) < A(y, y); });
  w.resize(n);
  std::vector<double> vs((size_t)n * n);
  for (int j = 0; j < n; ++j) {
    w[j] = A(idx[j], idx[j]);
    std::memcpy(&vs[(size_t)j * n], &v[(size_t)idx[j] * n], sizeof(double) * n);
  }
  v.swap(vs);
}

}  // namespace obhip

using namespace obhip;

// outermod::build, value part (modandbase.cpp:210-255)
int obhip_model::build() {
  if (!knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  const uint64_t Mtot = M();
  mmax = 0;
  for (uint64_t l = 0; l < d; ++l) mmax = std::max(mmax, m_of(l));
  rotmat.assign(mmax * Mtot, 0.0);
  basisvar.assign(Mtot, 0.0);
  maxlevel.assign(d, 0);
  // gradient bookkeeping (modandbase.cpp:183-197)
  hypmatch.assign(hypst[d], 0);
  gest.assign(hypst[d] + 1, 0);
  {
    uint64_t cur = 0;
    for (uint64_t l = 0; l < d; ++l)
      for (uint64_t h = hypst[l]; h < hypst[l + 1]; ++h) {
        hypmatch[h] = l;
        gest[h] = cur;
        cur += m_of(l);
      }
    gest[hypst[d]] = cur;
    rotmat_gradhyp.assign(mmax * cur, 0.0);
    logbasisvar_gradhyp.assign(cur, 0.0);
  }
  // The dimensions are independent eigen-problems writing disjoint slices; a BFGS run calls
  // this once per function evaluation (om$updatehyp), so they run on host threads.
  auto build_dim = [&](uint64_t k) {
    const uint64_t lenh = m_of(k), o = knotptst[k];
    const double *xs = &knotpt[o];
    std::vector<double> R(lenh * lenh), w, U;
    cov_host(kinds[k], &hyp[hypst[k]], xs, lenh, xs, lenh, R.data());
    jacobi_eigh((int)lenh, R, w, U);
    // descending order (":237-238"), sign fix (":241-242")
    std::vector<double> sr(lenh);
    std::vector<double> Ud(lenh * lenh);
    for (uint64_t j = 0; j < lenh; ++j) {
      sr[j] = w[lenh - 1 - j];
      std::memcpy(&Ud[j * lenh], &U[(lenh - 1 - j) * lenh], sizeof(double) * lenh);
    }
    const uint64_t halfw = lenh / 2;
    for (uint64_t j = 0; j < lenh; ++j) {
      const double key = Ud[j * lenh + halfw] +
                         (halfw + 1 < lenh ? 2.71828 * Ud[j * lenh + halfw + 1] : 0.0);
      const double sg = key > 0 ? 1.0 : (key < 0 ? -1.0 : 0.0);
      for (uint64_t i = 0; i < lenh; ++i) Ud[j * lenh + i] *= sg;
    }
    double mean = 0;
    for (double s : sr) mean += s;
    mean /= (double)lenh;
    const double minsv = 0.00000000001 * mean;
    int64_t ml = (int64_t)lenh - 1;
    for (uint64_t j = 0; j + 1 < lenh; ++j)
      if (sr[j] - sr[j + 1] < minsv) {
        ml = (int64_t)j;
        break;
      }
    maxlevel[k] = ml;
    const double lo = minsv / 1000, hi = (double)lenh * minsv / 1000;
    for (uint64_t j = 0; j < lenh; ++j) {
      const double lin = lenh > 1 ? lo + (hi - lo) * (double)j / (double)(lenh - 1) : lo;
      sr[j] += lin;
    }
    const double sq = std::sqrt((double)lenh);
    for (uint64_t j = 0; j < lenh; ++j) {
      const double dv = sr[j] / sq;
      for (uint64_t i = 0; i < lenh; ++i)
        rotmat[(o + j) * mmax + i] = Ud[j * lenh + i] / dv;
      basisvar[o + j] = std::log(sr[j] / (double)lenh);
    }
    // gradient matrices (modandbase.cpp:257-274) from the jittered sr and sign-fixed U:
    // UtdRV = U^T dR U; dlog(var) = diag(UtdRV) / sr; Ah = U (UtdRV % Fm) / (sr / sqrt(m)),
    // Fm[i][j] = 1 / (sr[j] - sr[i]) off the diagonal and -1 / sr[i] on it
    const uint64_t nh = hypst[k + 1] - hypst[k];
    std::vector<double> Rge(lenh * lenh * nh), T1(lenh * lenh), T2(lenh * lenh);
    cov_gradhyp_host(kinds[k], &hyp[hypst[k]], xs, lenh, xs, lenh, Rge.data());
    for (uint64_t l = 0; l < nh; ++l) {
      const double *dR = &Rge[l * lenh * lenh];  // symmetric
      // T1 = dR U (column j of U is Ud[j*lenh ..])
      for (uint64_t j = 0; j < lenh; ++j)
        for (uint64_t i = 0; i < lenh; ++i) {
          double acc = 0;
          for (uint64_t q = 0; q < lenh; ++q) acc += dR[q * lenh + i] * Ud[j * lenh + q];
          T1[j * lenh + i] = acc;
        }
      // T2 = (U^T T1) % Fm
      const uint64_t go = gest[hypst[k] + l];
      for (uint64_t j = 0; j < lenh; ++j)
        for (uint64_t i = 0; i < lenh; ++i) {
          double acc = 0;
          for (uint64_t q = 0; q < lenh; ++q) acc += Ud[i * lenh + q] * T1[j * lenh + q];
          if (i == j) logbasisvar_gradhyp[go + j] = acc / sr[j];
          T2[j * lenh + i] = acc * (i == j ? -1.0 / sr[i] : 1.0 / (sr[j] - sr[i]));
        }
      // Ah = U T2, columns scaled by sqrt(m) / sr[j]
      for (uint64_t j = 0; j < lenh; ++j) {
        const double dv = sr[j] / sq;
        for (uint64_t i = 0; i < lenh; ++i) {
          double acc = 0;
          for (uint64_t q = 0; q < lenh; ++q) acc += Ud[q * lenh + i] * T2[j * lenh + q];
          rotmat_gradhyp[(go + j) * mmax + i] = acc / dv;
        }
      }
    }
  };
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const uint64_t nthr = std::min<uint64_t>({d, (uint64_t)hw, 16});
  if (nthr <= 1) {
    for (uint64_t k = 0; k < d; ++k) build_dim(k);
  } else {
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> pool;
    for (uint64_t t = 0; t < nthr; ++t)
      pool.emplace_back([&] {
        for (uint64_t k = next++; k < d; k = next++) build_dim(k);
      });
    for (std::thread &th : pool) th.join();
  }
  version = obhip::next_model_version();
  return 0;
}

// ---- C ABI: covariance + model ------------------------------------------------
extern "C" {

int obhip_cov_numhyp(int kind, int *numhyp) {
  if (kind < 0 || kind >= kNumCov || !numhyp) return fail(OBHIP_ERR_INVALID, "bad covariance kind");
  *numhyp = kCovInfo[kind].numhyp;
  return 0;
}

int obhip_cov_info(int kind, double *hyp0, double *hyplb, double *hypub,
                   double *hypvar, double *lowbnd, double *uppbnd) {
  if (kind < 0 || kind >= kNumCov) return fail(OBHIP_ERR_INVALID, "bad covariance kind");
  const CovInfo &ci = kCovInfo[kind];
  for (int l = 0; l < ci.numhyp; ++l) {
    if (hyp0) hyp0[l] = ci.hyp0[l];
    if (hyplb) hyplb[l] = ci.hyplb[l];
    if (hypub) hypub[l] = ci.hypub[l];
    if (hypvar) hypvar[l] = ci.hypvar[l];
  }
  if (lowbnd) *lowbnd = ci.lowbnd;
  if (uppbnd) *uppbnd = ci.uppbnd;
  return 0;
}

int obhip_cov(int kind, const double *hyp, const double *x1, uint64_t n1,
              const double *x2, uint64_t n2, double *out) {
  if (kind < 0 || kind >= kNumCov) return fail(OBHIP_ERR_INVALID, "bad covariance kind");
  if (!hyp || (!x1 && n1) || (!x2 && n2) || (!out && n1 * n2))
    return fail(OBHIP_ERR_INVALID, "null argument");
  cov_host(kind, hyp, x1, n1, x2, n2, out);
  return 0;
}

int obhip_cov_gradhyp(int kind, const double *hyp, const double *x1, uint64_t n1, const double *x2,
                      uint64_t n2, double *out) {
  if (kind < 0 || kind >= kNumCov) return fail(OBHIP_ERR_INVALID, "bad covariance kind");
  if (!hyp || (!x1 && n1) || (!x2 && n2) || (!out && n1 * n2))
    return fail(OBHIP_ERR_INVALID, "null argument");
  cov_gradhyp_host(kind, hyp, x1, n1, x2, n2, out);
  return 0;
}

int obhip_cov_hyplpdf(int kind, const double *hyp, double *out) {
  if (kind < 0 || kind >= kNumCov || !hyp || !out) return fail(OBHIP_ERR_INVALID, "bad argument");
  *out = cov_hyplpdf_host(kind, hyp);
  return 0;
}

int obhip_model_get_knots(const obhip_model *m, uint64_t *knotptst, double *knotpt) {
  if (!m) return fail(OBHIP_ERR_INVALID, "null model");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  if (knotptst) std::copy(m->knotptst.begin(), m->knotptst.end(), knotptst);
  if (knotpt) std::copy(m->knotpt.begin(), m->knotpt.end(), knotpt);
  return 0;
}

int obhip_model_create(obhip_model **out, uint64_t d, const int *kinds) {
  if (!out || !kinds || d == 0) return fail(OBHIP_ERR_INVALID, "model_create: bad argument");
  for (uint64_t l = 0; l < d; ++l)
    if (kinds[l] < 0 || kinds[l] >= kNumCov)
      return fail(OBHIP_ERR_INVALID, "need to choose one of the existing cov functions");
  obhip_model *m = new obhip_model();
  m->d = d;
  m->kinds.assign(kinds, kinds + d);
  m->hypst.resize(d + 1);
  uint64_t cur = 0;
  for (uint64_t l = 0; l < d; ++l) {
    m->hypst[l] = cur;
    cur += kCovInfo[kinds[l]].numhyp;
  }
  m->hypst[d] = cur;
  m->hyp.resize(cur);
  for (uint64_t l = 0; l < d; ++l)
    for (int h = 0; h < kCovInfo[kinds[l]].numhyp; ++h)
      m->hyp[m->hypst[l] + h] = kCovInfo[kinds[l]].hyp0[h];
  *out = m;
  return 0;
}

int obhip_model_destroy(obhip_model *m) {
  delete m;
  return 0;
}

int obhip_model_set_knots(obhip_model *m, const uint64_t *knotptst,
                          const double *knotpt) {
  if (!m || !knotptst || !knotpt) return fail(OBHIP_ERR_INVALID, "set_knots: null argument");
  if (knotptst[0] != 0) return fail(OBHIP_ERR_INVALID, "set_knots: knotptst[0] must be 0");
  for (uint64_t l = 0; l < m->d; ++l) {
    if (knotptst[l + 1] < knotptst[l] + 2)
      return fail(OBHIP_ERR_INVALID, "set_knots: need at least 2 knots per dimension");
    const CovInfo &ci = kCovInfo[m->kinds[l]];
    for (uint64_t j = knotptst[l]; j < knotptst[l + 1]; ++j)
      if (!(knotpt[j] >= ci.lowbnd) || !(knotpt[j] <= ci.uppbnd))
        return fail(OBHIP_ERR_INVALID,
                    std::to_string(l + 1) + "knot point needs to be between " +
                        std::to_string(ci.lowbnd) + " and " + std::to_string(ci.uppbnd));
  }
  m->knotptst.assign(knotptst, knotptst + m->d + 1);
  m->knotpt.assign(knotpt, knotpt + knotptst[m->d]);
  m->knots_set = true;
  return m->build();
}

int obhip_model_set_hyp(obhip_model *m, const double *hyp, uint64_t nhyp) {
  if (!m || !hyp) return fail(OBHIP_ERR_INVALID, "set_hyp: null argument");
  if (nhyp != m->hyp.size()) return fail(OBHIP_ERR_INVALID, "wrongsized vector");
  m->hyp.assign(hyp, hyp + nhyp);
  if (m->knots_set) return m->build();
  return 0;
}

int obhip_model_get_hyp(const obhip_model *m, double *hyp) {
  if (!m || !hyp) return fail(OBHIP_ERR_INVALID, "get_hyp: null argument");
  std::copy(m->hyp.begin(), m->hyp.end(), hyp);
  return 0;
}

int obhip_model_dims(const obhip_model *m, uint64_t *d, uint64_t *M,
                     uint64_t *mmax, uint64_t *nhyp) {
  if (!m) return fail(OBHIP_ERR_INVALID, "model_dims: null model");
  if (d) *d = m->d;
  if (M) *M = m->M();
  if (mmax) *mmax = m->mmax;
  if (nhyp) *nhyp = m->hyp.size();
  return 0;
}

int obhip_model_get_rotation(const obhip_model *m, double *rotmat,
                             double *basisvar, int64_t *maxlevel) {
  if (!m) return fail(OBHIP_ERR_INVALID, "null model");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  if (rotmat) std::copy(m->rotmat.begin(), m->rotmat.end(), rotmat);
  if (basisvar) std::copy(m->basisvar.begin(), m->basisvar.end(), basisvar);
  if (maxlevel) std::copy(m->maxlevel.begin(), m->maxlevel.end(), maxlevel);
  return 0;
}

int obhip_model_set_rotation(obhip_model *m, const double *rotmat,
                             const double *basisvar, const int64_t *maxlevel) {
  if (!m || !rotmat || !basisvar || !maxlevel) return fail(OBHIP_ERR_INVALID, "null argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  m->rotmat.assign(rotmat, rotmat + m->mmax * m->M());
  m->basisvar.assign(basisvar, basisvar + m->M());
  for (uint64_t l = 0; l < m->d; ++l) {
    if (maxlevel[l] < 0 || (uint64_t)maxlevel[l] >= m->m_of(l))
      return fail(OBHIP_ERR_INVALID, "set_rotation: maxlevel out of range");
  }
  m->maxlevel.assign(maxlevel, maxlevel + m->d);
  m->version = obhip::next_model_version();
  return 0;
}

int obhip_model_grad_layout(const obhip_model *m, uint64_t *nhyp, uint64_t *hypmatch,
                            uint64_t *gest) {
  if (!m) return fail(OBHIP_ERR_INVALID, "null model");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  if (nhyp) *nhyp = m->nhyp();
  if (hypmatch) std::copy(m->hypmatch.begin(), m->hypmatch.end(), hypmatch);
  if (gest) std::copy(m->gest.begin(), m->gest.end(), gest);
  return 0;
}

int obhip_model_get_rotation_grad(const obhip_model *m, double *rotmat_gradhyp,
                                  double *logbasisvar_gradhyp) {
  if (!m) return fail(OBHIP_ERR_INVALID, "null model");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  if (rotmat_gradhyp)
    std::copy(m->rotmat_gradhyp.begin(), m->rotmat_gradhyp.end(), rotmat_gradhyp);
  if (logbasisvar_gradhyp)
    std::copy(m->logbasisvar_gradhyp.begin(), m->logbasisvar_gradhyp.end(), logbasisvar_gradhyp);
  return 0;
}

int obhip_model_set_rotation_grad(obhip_model *m, const double *rotmat_gradhyp,
                                  const double *logbasisvar_gradhyp) {
  if (!m || !rotmat_gradhyp || !logbasisvar_gradhyp)
    return fail(OBHIP_ERR_INVALID, "null argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  const uint64_t ng = m->gest.back();
  m->rotmat_gradhyp.assign(rotmat_gradhyp, rotmat_gradhyp + m->mmax * ng);
  m->logbasisvar_gradhyp.assign(logbasisvar_gradhyp, logbasisvar_gradhyp + ng);
  m->version = obhip::next_model_version();
  return 0;
}

int obhip_model_term_lvar_gradhyp(const obhip_model *m, const uint64_t *terms, uint64_t p,
                                  double *out) {
  if (!m || !terms || !out) return fail(OBHIP_ERR_INVALID, "term_lvar_gradhyp: null argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  for (uint64_t h = 0; h < m->nhyp(); ++h) {
    const uint64_t l = m->hypmatch[h];
    for (uint64_t k = 0; k < p; ++k) {
      const uint64_t t = terms[l * p + k];
      if (t >= m->m_of(l)) return fail(OBHIP_ERR_INVALID, "term level out of range");
      out[h * p + k] = m->logbasisvar_gradhyp[m->gest[h] + t];
    }
  }
  return 0;
}

int obhip_model_term_var(const obhip_model *m, const uint64_t *terms,
                         uint64_t p, double *out) {
  if (!m || !terms || !out) return fail(OBHIP_ERR_INVALID, "term_var: null argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  for (uint64_t k = 0; k < p; ++k) {
    double s = 0;
    for (uint64_t l = 0; l < m->d; ++l) {
      const uint64_t t = terms[l * p + k];
      if (t >= m->m_of(l)) return fail(OBHIP_ERR_INVALID, "term level out of range");
      s += m->basisvar[m->knotptst[l] + t];
    }
    out[k] = std::exp(s);
  }
  return 0;
}

int obhip_model_hyplpdf(const obhip_model *m, const double *hyp, uint64_t nhyp,
                        double *out) {
  if (!m || !hyp || !out) return fail(OBHIP_ERR_INVALID, "hyplpdf: null argument");
  if (nhyp != m->hyp.size()) {
    *out = -std::numeric_limits<double>::infinity();
    return 0;
  }
  double s = 0;
  for (uint64_t l = 0; l < m->d; ++l) s += cov_hyplpdf_host(m->kinds[l], hyp + m->hypst[l]);
  *out = s;
  return 0;
}

int obhip_model_hyplpdf_grad(const obhip_model *m, const double *hyp, uint64_t nhyp, double *out) {
  if (!m || !hyp || !out) return fail(OBHIP_ERR_INVALID, "hyplpdf_grad: null argument");
  for (uint64_t h = 0; h < m->hyp.size(); ++h) out[h] = 0.0;
  if (nhyp != m->hyp.size()) return 0;  // modandbase.cpp:111: zeros on a size mismatch
  for (uint64_t l = 0; l < m->d; ++l) {
    // covf::lpdf_gradhyp, covfuncs.cpp:53-70 (zeros outside the box)
    const CovInfo &ci = kCovInfo[m->kinds[l]];
    const double *hp = hyp + m->hypst[l];
    double *o = out + m->hypst[l];
    bool inside = true;
    for (int q = 0; q < ci.numhyp; ++q)
      if (ci.hypub[q] < hp[q] || ci.hyplb[q] > hp[q]) inside = false;
    if (!inside) continue;
    for (int q = 0; q < ci.numhyp; ++q)
      o[q] = -5.0 / (ci.hypub[q] - hp[q]) + 5.0 / (hp[q] - ci.hyplb[q]) -
             (hp[q] - ci.hyp0[q]) / ci.hypvar[q];
  }
  return 0;
}

}  // extern "C"

// ---- term selection -------------------------------------------------------------
namespace {

struct VecHash {
  size_t operator()(const std::vector<uint16_t> &v) const {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint16_t x : v) {
      h ^= x;
      h *= 0x100000001b3ull;
    }
    return (size_t)h;
  }
};

inline uint64_t splitmix64_next(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace

extern "C" int obhip_model_select_terms(const obhip_model *m, uint64_t p,
                                        uint64_t seed, uint64_t *terms_out) {
  if (!m || !terms_out) return fail(OBHIP_ERR_INVALID, "select_terms: null argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  const uint64_t d = m->d;
  typedef std::vector<uint16_t> Term;
  auto value = [&](const Term &t) {
    double s = 0;
    for (uint64_t l = 0; l < d; ++l) s += m->basisvar[m->knotptst[l] + t[l]];
    return s;
  };
  std::vector<Term> cand;   // open candidates, array order as in the reference
  std::vector<double> cval;
  std::unordered_set<Term, VecHash> chosen;
  cand.emplace_back(d, 0);
  cval.push_back(value(cand[0]));
  uint64_t rng = seed;
  std::vector<uint64_t> near;
  for (uint64_t k = 0; k < p; ++k) {
    if (cand.empty())
      return fail(OBHIP_ERR_INVALID, "select_terms: lattice exhausted before p terms");
    double mx = cval[0];
    for (double v : cval) mx = std::max(mx, v);
    const double mval = -0.1 + mx;  // modandbase.cpp:406
    uint64_t kstar = 0;
    if (seed == 0) {
      for (uint64_t i = 0; i < cval.size(); ++i)
        if (cval[i] > mval) {
          kstar = i;
          break;
        }
    } else {
      near.clear();
      for (uint64_t i = 0; i < cval.size(); ++i)
        if (cval[i] > mval) near.push_back(i);
      kstar = near[splitmix64_next(rng) % near.size()];
    }
    Term T = cand[kstar];
    for (uint64_t l = 0; l < d; ++l) terms_out[l * p + k] = T[l];
    chosen.insert(T);
    // remove by moving the last candidate into the hole (":414-417")
    const uint64_t last = cand.size() - 1;
    if (last > kstar) {
      cand[kstar] = cand[last];
      cval[kstar] = cval[last];
    }
    cand.pop_back();
    cval.pop_back();
    // children T + e_l whose parents are all chosen (":419-436")
    uint64_t nnzT = 0;
    for (uint64_t l = 0; l < d; ++l) nnzT += T[l] > 0;
    for (uint64_t l = 0; l < d; ++l) {
      if ((int64_t)T[l] >= m->maxlevel[l]) continue;
      const uint64_t nparents = nnzT + (T[l] < 1 ? 1 : 0);
      uint64_t have = 1;  // T itself
      Term S = T;
      S[l] += 1;
      for (uint64_t a = 0; a < d; ++a) {
        if (a == l || T[a] == 0) continue;
        S[a] -= 1;
        have += chosen.count(S);
        S[a] += 1;
      }
      if (have == nparents) {
        cand.push_back(S);
        cval.push_back(value(S));
      }
    }
  }
  return 0;
}
