// predr_std (src/lpdfs/loglik_std.cpp:218-256): prediction of the loglik_std model with the
// posterior covariance of the coefficients,
//   mean = B theta,   var_i = b_i^T C b_i + e^{2 sigma},   C = inv(total Hessian)
// (SURVEY.md 8f-2).  The reference forms C with arma::inv (:227) and
// rowsum((B C) % B) (:251-255); here H = L L^T is factorised by the library's own Cholesky
// (kernels_chol.hip) and var_i = || L^{-1} b_i ||^2 + e^{2 sigma}: one triangular solve with
// n right-hand sides on the row-major design matrix of the new points (which is exactly
// the column-major p x n matrix B^T), then a column norm.
//
// The triangular solve is a plain library call (rocBLAS dtrsm, p^2 n flop); rocBLAS is
// loaded at run time so that the hot path neither links nor needs it.  Every other step
// (basis, design matrix, Cholesky, norms) is this library's own HIP code.
#include <dlfcn.h>

#include <cmath>
#include <cstring>

#include "obhip_internal.h"

using namespace obhip;

namespace {

// the few rocBLAS entry points, by their documented C signatures (rocblas.h)
typedef void *rb_handle;
typedef int (*rb_create_t)(rb_handle *);
typedef int (*rb_destroy_t)(rb_handle);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_dtrsm64_t)(rb_handle, int side, int uplo, int trans, int diag, int64_t m, int64_t n,
                            const double *alpha, const double *A, int64_t lda, double *B,
                            int64_t ldb);
// enum values of rocblas-types.h
constexpr int kSideLeft = 141, kFillUpper = 121, kOpTranspose = 112, kDiagNonUnit = 131;

struct RocBlas {
  void *lib = nullptr;
  rb_create_t create = nullptr;
  rb_destroy_t destroy = nullptr;
  rb_set_stream_t set_stream = nullptr;
  rb_dtrsm64_t dtrsm = nullptr;
  rb_handle h = nullptr;
};

int rocblas(RocBlas **out) {
  static RocBlas rb;
  if (!rb.lib) {
    // a copy already in the process (e.g. the one PyTorch ships) wins
    const char *names[] = {"librocblas.so.5", "librocblas.so.4", "librocblas.so"};
    for (const char *nm : names)
      if ((rb.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!rb.lib)
      for (const char *nm : names)
        if ((rb.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!rb.lib)
      return fail(OBHIP_ERR_STATE, "predict_std needs rocBLAS (librocblas.so) for its triangular solve");
    rb.create = (rb_create_t)dlsym(rb.lib, "rocblas_create_handle");
    rb.destroy = (rb_destroy_t)dlsym(rb.lib, "rocblas_destroy_handle");
    rb.set_stream = (rb_set_stream_t)dlsym(rb.lib, "rocblas_set_stream");
    rb.dtrsm = (rb_dtrsm64_t)dlsym(rb.lib, "rocblas_dtrsm_64");
    if (!rb.create || !rb.set_stream || !rb.dtrsm) {
      rb.lib = nullptr;
      return fail(OBHIP_ERR_STATE, "rocBLAS lacks rocblas_dtrsm_64");
    }
    if (rb.create(&rb.h) != 0) {
      rb.lib = nullptr;
      return fail(OBHIP_ERR_HIP, "rocblas_create_handle failed");
    }
  }
  *out = &rb;
  return 0;
}

}  // namespace

namespace obhip {
int launch_colnorm2(const double *d_Z, uint64_t ld, uint64_t p, uint64_t n, double add,
                    double *d_out);
}

extern "C" int obhip_predict_std(const obhip_model *m, const obhip_terms *tc, const double *theta,
                                 const double *H, const double *x, uint64_t n, uint64_t ldx,
                                 double *mean, double sigma, double *var) {
  if (!m || !tc || !theta || !x || !mean || n == 0 || ldx < n)
    return fail(OBHIP_ERR_INVALID, "predict_std: bad argument");
  // the mean is the ordinary fused predictor
  OB_TRY(obhip_predict(m, tc, theta, x, n, ldx, mean, nullptr, sigma, nullptr));
  if (!var) return 0;
  if (!H) return fail(OBHIP_ERR_INVALID, "predict_std: var needs the total Hessian");
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  const uint64_t p = t.p;
  // basis at the new points up to the levels the terms use, then B row-major
  std::vector<int64_t> cap(t.maxlev);
  obhip_basis *b = nullptr;
  OB_TRY(obhip_basis_create(&b, m, x, n, ldx, cap.data()));
  struct Guard {
    obhip_basis *b;
    ~Guard() { obhip_basis_destroy(b); }
  } guard{b};
  OB_TRY(t.prepare(b->md.cap, b->md.dims_h));
  DevBuf<double> dB, dH, drhs, dth, dvar;
  DevBuf<char> ws;
  OB_TRY(dB.alloc(b->n_pad * t.p_pad));
  OB_TRY(launch_materialize_rows(*b, t, dB.p));
  // H = L L^T with the library's Cholesky (the solve it carries along is not used)
  OB_TRY(dH.upload(H, p * p));
  std::vector<double> zero(p, 0.0);
  OB_TRY(drhs.upload(zero.data(), p));
  OB_TRY(dth.alloc(p));
  const uint64_t wsb = newton_workspace_bytes(p);
  OB_TRY(ws.alloc(wsb));
  OB_TRY(launch_newton_solve(p, dH.p, drhs.p, dth.p, ws.p, wsb));
  // Z = L^{-1} B^T.  dH is row-major with L in its lower triangle = column-major upper
  // triangular A = L^T, so op(A) = A^T = L; dB is column-major p_pad x n_pad.
  RocBlas *rb = nullptr;
  OB_TRY(rocblas(&rb));
  if (rb->set_stream(rb->h, cur_stream()) != 0) return fail(OBHIP_ERR_HIP, "rocblas_set_stream failed");
  const double one = 1.0;
  {
    ProfScope ps("predict_std_trsm");
    if (rb->dtrsm(rb->h, kSideLeft, kFillUpper, kOpTranspose, kDiagNonUnit, (int64_t)p, (int64_t)n,
                  &one, dH.p, (int64_t)p, dB.p, (int64_t)t.p_pad) != 0)
      return fail(OBHIP_ERR_HIP, "rocblas_dtrsm failed");
  }
  OB_TRY(dvar.alloc(n));
  OB_TRY(launch_colnorm2(dB.p, t.p_pad, p, n, std::exp(2.0 * sigma), dvar.p));
  OB_HIP(hipMemcpyAsync(var, dvar.p, n * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  return 0;
}
