// CPU restatement (C++/OpenMP, no Armadillo) of the reference's own loops --
// TEST INFRASTRUCTURE ONLY: a second, independent implementation that the tests
// compare with oracle/ob_oracle.py, and the `cpu_baseline` leg of bench.py.  The
// product (outerbase_amd/) never links or calls it.
//
// It keeps the reference's OpenMP schedule so that its timing is a fair stand-in
// for "the reference's CPU path": nthreads = omp_get_num_procs() unless given
// (src/modandbase.cpp:464), chunksize = max(32, min(1 + 2048/T, n/(4T) + 1)),
// tall/wide switch at 20 chunks (src/modandbase.cpp:504-512), one row chunk per
// loop iteration with thread-local accumulators merged at the end
// (src/linalg.cpp:96-125, 313-336).  Functions cite the lines they follow.
//
// Layouts are the reference's: column-major FP64, terms p x d column-major u64.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Sched {
  int64_t chunk, loops;
  bool vertpl;
};

Sched sched(int64_t n, int T) {  // outerbase::setloopvals_, modandbase.cpp:504-512
  Sched s;
  const int64_t maxchunk = 1 + 2048 / T, minchunk = 32;
  s.chunk = std::max(minchunk, std::min(maxchunk, n / (4 * (int64_t)T) + 1));
  s.loops = (n + s.chunk - 1) / s.chunk;
  s.vertpl = s.loops > 20;
  return s;
}

inline double mat25(double h) { return (1.0 + h + h * h / 3.0) * std::exp(-h); }

// covf::cov for one x against the knots of a dimension (covfuncs.cpp:113-126,197-212,285-310)
void cov_row(int kind, const double *hyp, double x, const double *knots, int m, double *out) {
  const double a = 2.0, b = 0.25;
  if (kind == 0) {
    const double ls = std::exp(a * hyp[0]);
    for (int j = 0; j < m; ++j) out[j] = mat25(std::fabs(x / ls - knots[j] / ls));
  } else if (kind == 1) {
    const double powv = std::exp(b * hyp[1]), ls = std::exp(a * hyp[0] + b * hyp[1]);
    const double xt = std::pow(x, powv) / ls;
    for (int j = 0; j < m; ++j) out[j] = mat25(std::fabs(xt - std::pow(knots[j], powv) / ls));
  } else {
    const double lss = std::exp(a * hyp[0]), lsc = std::exp(a * hyp[1]);
    const double sx = std::sin(x) / lss, cx = std::cos(x) / lsc;
    for (int j = 0; j < m; ++j) {
      const double hs = sx - std::sin(knots[j]) / lss, hc = cx - std::cos(knots[j]) / lsc;
      out[j] = mat25(std::sqrt(hs * hs + hc * hc));
    }
  }
}

}  // namespace

extern "C" {

int ob_cpu_num_procs() { return omp_get_num_procs(); }

// outerbase::build, value part (modandbase.cpp:547-598 tall branch; the short
// branch computes the same values): basemat n x M, basescale n.
void ob_cpu_build(int64_t n, int64_t d, const double *x, const int *kinds, const uint64_t *knotptst,
                  const double *knotpt, const uint64_t *hypst, const double *hyp,
                  const double *rotmat, int64_t mmax, int nthreads, double *basemat,
                  double *basescale) {
  const int T = nthreads > 0 ? nthreads : omp_get_num_procs();
  const Sched s = sched(n, T);
  for (int64_t i = 0; i < n; ++i) basescale[i] = 1.0;
#pragma omp parallel num_threads(T)
  {
    std::vector<double> kv, R;
#pragma omp for
    for (int64_t j = 0; j < s.loops; ++j) {
      const int64_t i0 = j * s.chunk, i1 = std::min((j + 1) * s.chunk, n);
      for (int64_t k = 0; k < d; ++k) {
        const int64_t o = (int64_t)knotptst[k], m = (int64_t)knotptst[k + 1] - o;
        kv.resize(m);
        R.resize(m);
        for (int64_t i = i0; i < i1; ++i) {
          cov_row(kinds[k], hyp + hypst[k], x[k * n + i], knotpt + o, (int)m, kv.data());
          // R = cov . rotmat (modandbase.cpp:294)
          for (int64_t c = 0; c < m; ++c) {
            const double *rc = rotmat + (o + c) * mmax;
            double acc = 0;
            for (int64_t q = 0; q < m; ++q) acc += kv[q] * rc[q];
            R[c] = acc;
          }
          const double c0 = R[0];
          basescale[i] *= c0;                                   // :573
          basemat[(o + 0) * n + i] = 1.0;                       // :574
          for (int64_t c = 1; c < m; ++c) basemat[(o + c) * n + i] = R[c] / c0;  // :297
        }
      }
    }
  }
}

// getm_ / domat_ (linalg.cpp:647-715)
void ob_cpu_getmat(int64_t n, int64_t p, int64_t d, const uint64_t *terms, const uint64_t *knotptst,
                   const double *basemat, const double *basescale, int nthreads, double *out) {
  const int T = nthreads > 0 ? nthreads : omp_get_num_procs();
#pragma omp parallel for num_threads(T)
  for (int64_t k = 0; k < p; ++k) {
    double *col = out + k * n;
    for (int64_t i = 0; i < n; ++i) col[i] = basescale[i];
    for (int64_t l = 0; l < d; ++l) {
      const uint64_t t = terms[l * p + k];
      if (t > 0) {
        const double *b = basemat + (knotptst[l] + t) * n;
        for (int64_t i = 0; i < n; ++i) col[i] *= b[i];
      }
    }
  }
}

// prodmm_ / domult_ (linalg.cpp:57-131): out = B a
void ob_cpu_mm(int64_t n, int64_t p, int64_t d, const uint64_t *terms, const uint64_t *knotptst,
               const double *basemat, const double *basescale, const double *a, int nthreads,
               double *out) {
  const int T = nthreads > 0 ? nthreads : omp_get_num_procs();
  const Sched s = sched(n, T);
  std::fill(out, out + n, 0.0);
  if (s.vertpl) {  // tall: one row chunk per iteration (linalg.cpp:96-113)
#pragma omp parallel num_threads(T)
    {
      std::vector<double> temp(s.chunk), acc(s.chunk);
#pragma omp for nowait
      for (int64_t lcv = 0; lcv < s.loops; ++lcv) {
        const int64_t i0 = lcv * s.chunk, len = std::min((lcv + 1) * s.chunk, n) - i0;
        std::fill(acc.begin(), acc.begin() + len, 0.0);
        for (int64_t k = 0; k < p; ++k) {
          std::fill(temp.begin(), temp.begin() + len, a[k]);
          for (int64_t l = 0; l < d; ++l) {
            const uint64_t t = terms[l * p + k];
            if (t > 0) {
              const double *b = basemat + (knotptst[l] + t) * n + i0;
              for (int64_t i = 0; i < len; ++i) temp[i] *= b[i];
            }
          }
          for (int64_t i = 0; i < len; ++i) acc[i] += temp[i];
        }
        for (int64_t i = 0; i < len; ++i) out[i0 + i] = acc[i] * basescale[i0 + i];
      }
    }
  } else {  // wide: terms split over threads, thread-local out (linalg.cpp:77-92)
#pragma omp parallel num_threads(T)
    {
      std::vector<double> temp(n), acc(n, 0.0);
#pragma omp for
      for (int64_t k = 0; k < p; ++k) {
        std::fill(temp.begin(), temp.end(), a[k]);
        for (int64_t l = 0; l < d; ++l) {
          const uint64_t t = terms[l * p + k];
          if (t > 0) {
            const double *b = basemat + (knotptst[l] + t) * n;
            for (int64_t i = 0; i < n; ++i) temp[i] *= b[i];
          }
        }
        for (int64_t i = 0; i < n; ++i) acc[i] += temp[i];
      }
#pragma omp critical
      for (int64_t i = 0; i < n; ++i) out[i] += acc[i];
    }
    for (int64_t i = 0; i < n; ++i) out[i] *= basescale[i];
  }
}

// tprodmm_ / dotmultsub_ (linalg.cpp:286-355): out = B^T a
void ob_cpu_tmm(int64_t n, int64_t p, int64_t d, const uint64_t *terms, const uint64_t *knotptst,
                const double *basemat, const double *basescale, const double *a, int nthreads,
                double *out) {
  const int T = nthreads > 0 ? nthreads : omp_get_num_procs();
  const Sched s = sched(n, T);
  std::fill(out, out + p, 0.0);
  std::vector<double> b(n);
  for (int64_t i = 0; i < n; ++i) b[i] = basescale[i] * a[i];  // :305
#pragma omp parallel num_threads(T)
  {
    std::vector<double> acc(p, 0.0), temp(s.vertpl ? s.chunk : n);
    if (s.vertpl) {
#pragma omp for nowait
      for (int64_t lcv = 0; lcv < s.loops; ++lcv) {
        const int64_t i0 = lcv * s.chunk, len = std::min((lcv + 1) * s.chunk, n) - i0;
        for (int64_t k = 0; k < p; ++k) {
          std::copy(b.begin() + i0, b.begin() + i0 + len, temp.begin());
          for (int64_t l = 0; l < d; ++l) {
            const uint64_t t = terms[l * p + k];
            if (t > 0) {
              const double *bm = basemat + (knotptst[l] + t) * n + i0;
              for (int64_t i = 0; i < len; ++i) temp[i] *= bm[i];
            }
          }
          double sum = 0;
          for (int64_t i = 0; i < len; ++i) sum += temp[i];
          acc[k] += sum;
        }
      }
    } else {
#pragma omp for nowait
      for (int64_t k = 0; k < p; ++k) {
        std::copy(b.begin(), b.end(), temp.begin());
        for (int64_t l = 0; l < d; ++l) {
          const uint64_t t = terms[l * p + k];
          if (t > 0) {
            const double *bm = basemat + (knotptst[l] + t) * n;
            for (int64_t i = 0; i < n; ++i) temp[i] *= bm[i];
          }
        }
        double sum = 0;
        for (int64_t i = 0; i < n; ++i) sum += temp[i];
        acc[k] += sum;
      }
    }
#pragma omp critical
    for (int64_t k = 0; k < p; ++k) out[k] += acc[k];
  }
}

}  // extern "C"
