// Posterior-covariance quantities of the loglik_std model (SURVEY.md 8f-2 and the
// full-Hessian part of 8f-4), both built on H = L L^T from the library's own Cholesky
// (kernels_chol.hip) and on the row-major design matrix, which is exactly the column-major
// p x n matrix B^T:
//
//   obhip_predict_std   predr_std (src/lpdfs/loglik_std.cpp:218-256): mean = B theta,
//                       var_i = b_i^T inv(H) b_i + e^{2 sigma} = || L^{-1} b_i ||^2 + e^{2 sigma}
//                       (the reference forms inv(tothess), :227, and rowsum((B C) % B), :251-255)
//   obhip_margadj_full  lpdfvec::buildhess with the full Hessian (src/fit.cpp:270-299):
//                       -1/2 log det H and -1/2 tr(inv(H) dH) for every hyper-parameter and
//                       parameter, from Y = inv(H) B^T and streaming dot products instead of
//                       the reference's p x p x nhyp cubes
//
// The triangular solves with n right-hand sides are plain library calls (rocBLAS dtrsm,
// p^2 n flop each); rocBLAS is loaded at run time so that the hot path neither links nor
// needs it.  Every other step (basis, design matrices, Cholesky, norms, dot products) is
// this library's own HIP code.
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
constexpr int kSideLeft = 141, kFillUpper = 121, kOpNone = 111, kOpTranspose = 112, kDiagNonUnit = 131;

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
int launch_dot_cols(const double *d_A, const double *d_B, uint64_t ld, uint64_t p, uint64_t n,
                    double *d_out, double *d_part);
int launch_set_identity(double *d_A, uint64_t p);
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


// Marginal adjustment of lpdfvec(loglik_std, logpr_gauss) with the full Hessian
// (lpdfvec::buildhess, fit.cpp:270-299): val = -1/2 log det H and
//   gradhyp[l] = -1/2 tr(inv(H) dH/dhyp_l),  dH/dhyp_l = e^{-2 sigma}(B^T Bge_l + Bge_l^T B) - diag(lvarge_l prec)
// (loglik_std.cpp:180-192, logpr_gauss.cpp:165-173), likewise for the two para.  The
// reference forms the p x p x nhyp cubes and inv(H); here tr(inv(H) B^T Bge_l) =
// sum_i (inv(H) b_i) . bge_l,i needs Y = inv(H) B^T once (two triangular solves on the
// design matrix) and then one streaming dot product per hyper-parameter.
extern "C" int obhip_margadj_full(const obhip_basis *bc, const obhip_terms *tc, const obhip_model *m,
                                  const double *H, double sigma, double rho, double *val,
                                  double *gradhyp, double *gradpara) {
  if (!bc || !tc || !m || !H || !val) return fail(OBHIP_ERR_INVALID, "margadj_full: null argument");
  if (bc->model != m) return fail(OBHIP_ERR_INVALID, "margadj_full: basis belongs to another model");
  obhip_basis &b = *const_cast<obhip_basis *>(bc);
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  const uint64_t p = t.p, d = m->d, nh = m->nhyp();
  // H = L L^T
  DevBuf<double> dH, drhs, dth, dI, ddiag, dscal;
  DevBuf<char> ws;
  OB_TRY(dH.upload(H, p * p));
  std::vector<double> zero(p, 0.0);
  OB_TRY(drhs.upload(zero.data(), p));
  OB_TRY(dth.alloc(p));
  const uint64_t wsb = newton_workspace_bytes(p);
  OB_TRY(ws.alloc(wsb));
  OB_TRY(launch_newton_solve(p, dH.p, drhs.p, dth.p, ws.p, wsb));
  std::vector<double> ld(p);
  OB_HIP(hipMemcpy2DAsync(ld.data(), sizeof(double), dH.p, (p + 1) * sizeof(double), sizeof(double), p,
                          hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  double logdet = 0;
  for (double v : ld) logdet += 2.0 * std::log(v);
  *val = -0.5 * logdet;
  if (!gradhyp && !gradpara) return 0;

  RocBlas *rb = nullptr;
  OB_TRY(rocblas(&rb));
  if (rb->set_stream(rb->h, cur_stream()) != 0) return fail(OBHIP_ERR_HIP, "rocblas_set_stream failed");
  const double one = 1.0;
  // diag(inv(H)) = squared column norms of L^{-1}
  OB_TRY(dI.alloc(p * p));
  OB_TRY(launch_set_identity(dI.p, p));
  if (rb->dtrsm(rb->h, kSideLeft, kFillUpper, kOpTranspose, kDiagNonUnit, (int64_t)p, (int64_t)p, &one,
                dH.p, (int64_t)p, dI.p, (int64_t)p) != 0)
    return fail(OBHIP_ERR_HIP, "rocblas_dtrsm failed");
  OB_TRY(ddiag.alloc(p));
  OB_TRY(launch_colnorm2(dI.p, p, p, p, 0.0, ddiag.p));
  std::vector<double> hinv(p);
  OB_HIP(hipMemcpyAsync(hinv.data(), ddiag.p, p * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
  dI.release();
  // Y = inv(H) B^T on a copy of the design matrix (column-major p_pad x n_pad)
  OB_TRY(ensure_gradbasis(b));
  OB_TRY(ensure_bmat(b, t));
  const uint64_t nel = b.n_pad * t.p_pad;
  DevBuf<double> dY, dG;
  OB_TRY(dY.alloc(nel));
  OB_HIP(hipMemcpyAsync(dY.p, b.bmat.p, nel * sizeof(double), hipMemcpyDeviceToDevice, cur_stream()));
  if (rb->dtrsm(rb->h, kSideLeft, kFillUpper, kOpTranspose, kDiagNonUnit, (int64_t)p, (int64_t)b.n, &one,
                dH.p, (int64_t)p, dY.p, (int64_t)t.p_pad) != 0 ||
      rb->dtrsm(rb->h, kSideLeft, kFillUpper, kOpNone, kDiagNonUnit, (int64_t)p, (int64_t)b.n, &one, dH.p,
                (int64_t)p, dY.p, (int64_t)t.p_pad) != 0)
    return fail(OBHIP_ERR_HIP, "rocblas_dtrsm failed");
  OB_TRY(dscal.alloc(nh + 1 + 4096));
  double *part = dscal.p + nh + 1;
  // tr(inv(H) B^T B) and tr(inv(H) B^T Bge_l)
  OB_TRY(launch_dot_cols(b.bmat.p, dY.p, t.p_pad, p, b.n, dscal.p + nh, part));
  OB_TRY(dG.alloc(nel));
  for (uint64_t h = 0; h < nh; ++h) {
    OB_TRY(launch_materialize_rows(*b.grad->gb, *grad_view(t, b, h), dG.p));
    OB_TRY(launch_dot_cols(dG.p, dY.p, t.p_pad, p, b.n, dscal.p + h, part));
  }
  std::vector<double> q(nh + 1);
  OB_HIP(hipMemcpyAsync(q.data(), dscal.p, (nh + 1) * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  // prior parts (logpr_gauss.cpp:165-186): prec_k = 1 / (sd_k e^rho)^2
  const double e2 = std::exp(-2.0 * sigma);
  std::vector<double> prec(p);
  for (uint64_t k = 0; k < p; ++k) {
    double sv = 0;
    for (uint64_t l = 0; l < d; ++l) sv += m->basisvar[m->knotptst[l] + t.lev[k * d + l]];
    prec[k] = 1.0 / (std::exp(sv) * std::exp(2.0 * rho));
  }
  if (gradhyp)
    for (uint64_t h = 0; h < nh; ++h) {
      const uint64_t l = m->hypmatch[h];
      double pr = 0;
      for (uint64_t k = 0; k < p; ++k)
        pr += hinv[k] * prec[k] * m->logbasisvar_gradhyp[m->gest[h] + t.lev[k * d + l]];
      gradhyp[h] = -0.5 * (2.0 * e2 * q[h] - pr);
    }
  if (gradpara) {
    gradpara[0] = e2 * q[nh];  // -1/2 tr(inv(H) (-2 e^{-2 sigma} G)), loglik_std.cpp:199-203
    double pr = 0;
    for (uint64_t k = 0; k < p; ++k) pr += hinv[k] * prec[k];
    gradpara[1] = pr;          // -1/2 sum(-2 prec_k inv(H)_kk), logpr_gauss.cpp:181-186
  }
  return 0;
}
