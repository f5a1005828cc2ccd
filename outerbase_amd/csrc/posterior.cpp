// Posterior-covariance quantities of the loglik_std model (SURVEY.md 8f-2 and the
// full-Hessian part of 8f-4), all on this library's own kernels: H = L L^T by the Cholesky of
// kernels_chol.hip, X = L^-T by kernels_trtri.hip, and every p^2 n product as one pass of the
// FP64 matrix-core kernel of kernels_gram_panel.hip in its two-operand form (C = A^T Bm with
// both operands stored with the contraction index slow).
//
//   obhip_predict_std   predr_std (src/lpdfs/loglik_std.cpp:218-256): mean = B theta,
//                       var_i = b_i^T inv(H) b_i + e^{2 sigma} = || L^-1 b_i ||^2 + e^{2 sigma}.
//                       The reference forms inv(tothess) (arma::inv, :227) and
//                       rowsum((B inv(H)) % B) (:251-255); here Z = B L^-T is never stored:
//                       the kernel's epilogue squares and row-sums its 128 x 128 tiles.
//   obhip_margadj_full  lpdfvec::buildhess with the full Hessian (src/fit.cpp:270-299):
//                       -1/2 log det H and -1/2 tr(inv(H) dH) for every hyper-parameter and
//                       parameter, from Y = inv(H) B^T (inv(H) = L^-T L^-1 formed once) and
//                       streaming dot products instead of the reference's p x p x nhyp cubes.
#include <cmath>
#include <cstring>

#include "obhip_internal.h"
#include "vec_ops.h"

using namespace obhip;

namespace obhip {
int launch_dot_cols(const double *d_A, const double *d_B, uint64_t ld, uint64_t p, uint64_t n,
                    double *d_out, double *d_part);

static uint64_t pad128(uint64_t v) { return (v + 127) / 128 * 128; }

// f.L = Cholesky factor of H (lower triangle, row-major p x p), f.X = L^-T (pp x pp)
int post_factor_build(const double *d_H, uint64_t p, PostFactor &f, bool want_inverse) {
  f.p = p;
  f.pp = pad128(p);
  OB_TRY(f.L.alloc(p * p));
  OB_HIP(hipMemcpyAsync(f.L.p, d_H, p * p * sizeof(double), hipMemcpyDeviceToDevice, cur_stream()));
  DevBuf<double> rhs, th;
  DevBuf<char> ws;
  OB_TRY(rhs.alloc(p));
  OB_TRY(th.alloc(p));
  OB_HIP(hipMemsetAsync(rhs.p, 0, p * sizeof(double), cur_stream()));
  const uint64_t wsb = newton_workspace_bytes(p);
  OB_TRY(ws.alloc(wsb));
  OB_TRY(launch_newton_solve(p, f.L.p, rhs.p, th.p, ws.p, wsb));  // synchronises
  if (!want_inverse) return 0;
  DevBuf<double> dinv;
  OB_TRY(dinv.alloc((p + 63) / 64 * 4096));
  OB_TRY(f.X.alloc(f.pp * f.pp));
  OB_TRY(launch_trtri_lt(f.L.p, p, p, f.X.p, f.pp, dinv.p));
  OB_HIP(hipStreamSynchronize(cur_stream()));  // dinv is a local
  return 0;
}

// d_var[i] = || L^-1 b_i ||^2 + e2sigma at the n rows of d_x (column-major n x d, device)
int post_var_dev(const obhip_model &m, obhip_terms &t, const PostFactor &f, const double *d_x,
                 uint64_t n, double e2sigma, double *d_var) {
  const uint64_t p = f.p, pp = f.pp;
  if (t.p != p) return fail(OBHIP_ERR_INVALID, "posterior factor and terms disagree on p");
  // row chunks so that the term-major design matrix (pp x rows doubles) stays below 8 GB
  const uint64_t cmax = std::max<uint64_t>(128, ((8ull << 30) / (pp * sizeof(double))) / 128 * 128);
  const uint64_t ntj = pp / 128;
  DevBuf<double> Bcm, part;
  for (uint64_t r0 = 0; r0 < n; r0 += cmax) {
    const uint64_t nr = std::min(cmax, n - r0), npad = pad128(nr);
    obhip_basis *b = nullptr;
    // the chunk's rows of x: column-major with leading dimension n -> gather into a compact copy
    DevBuf<double> xc;
    const double *xsrc = d_x;
    if (r0 != 0 || nr != n) {
      OB_TRY(xc.alloc(nr * m.d));
      OB_HIP(hipMemcpy2DAsync(xc.p, nr * sizeof(double), d_x + r0, n * sizeof(double),
                              nr * sizeof(double), m.d, hipMemcpyDeviceToDevice, cur_stream()));
      xsrc = xc.p;
    }
    OB_TRY(obhip_basis_create_dev(&b, &m, xsrc, nr, t.maxlev.data()));
    struct Guard {
      obhip_basis *b;
      ~Guard() { obhip_basis_destroy(b); }
    } guard{b};
    OB_TRY(Bcm.alloc(pp * npad));
    OB_TRY(part.alloc(ntj * npad));
    OB_HIP(hipMemsetAsync(Bcm.p, 0, pp * npad * sizeof(double), cur_stream()));
    OB_TRY(launch_getmat(*b, t, Bcm.p, npad));  // B^T: term-major, rows contiguous
    {
      ProfScope ps("predict_std_gemm");
      OB_TRY(launch_atb(1, Bcm.p, npad, npad, f.X.p, pp, pp, pp, true, part.p, npad));
    }
    const double *pt = part.p;
    double *out = d_var + r0;
    OB_TRY(vmap(nr, [=] __device__(uint64_t i) {
      double s = e2sigma;
      for (uint64_t j = 0; j < ntj; ++j) s += pt[j * npad + i];
      out[i] = s;
    }));
    OB_HIP(hipStreamSynchronize(cur_stream()));
  }
  return 0;
}

}  // namespace obhip

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
  DevBuf<double> dH, dx, dvar;
  OB_TRY(dH.upload(H, p * p));
  PostFactor f;
  OB_TRY(post_factor_build(dH.p, p, f, true));
  if (ldx == n) {
    OB_TRY(dx.upload(x, n * m->d));
  } else {
    std::vector<double> xc(n * m->d);
    for (uint64_t l = 0; l < m->d; ++l) std::memcpy(&xc[l * n], x + l * ldx, n * sizeof(double));
    OB_TRY(dx.upload(xc.data(), xc.size()));
  }
  OB_TRY(dvar.alloc(n));
  OB_TRY(post_var_dev(*m, t, f, dx.p, n, std::exp(2.0 * sigma), dvar.p));
  OB_HIP(hipMemcpyAsync(var, dvar.p, n * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  return 0;
}

// Marginal adjustment of lpdfvec(loglik_std, logpr_gauss) with the full Hessian
// (lpdfvec::buildhess, fit.cpp:270-299): val = -1/2 log det H and
//   gradhyp[l] = -1/2 tr(inv(H) dH/dhyp_l),  dH/dhyp_l = e^{-2 sigma}(B^T Bge_l + Bge_l^T B) - diag(lvarge_l prec)
// (loglik_std.cpp:180-192, logpr_gauss.cpp:165-173), likewise for the two para.  The
// reference forms the p x p x nhyp cubes and inv(H) through eig_sym; here
// tr(inv(H) B^T Bge_l) = sum_{k, i} Y[k][i] Bge_l[i][k] with Y = inv(H) B^T formed once (term-
// major, like the design matrices it is contracted with) and one streaming dot product per
// hyper-parameter.
extern "C" int obhip_margadj_full(const obhip_basis *bc, const obhip_terms *tc, const obhip_model *m,
                                  const double *H, double sigma, double rho, double *val,
                                  double *gradhyp, double *gradpara) {
  if (!bc || !tc || !m || !H || !val) return fail(OBHIP_ERR_INVALID, "margadj_full: null argument");
  if (bc->model != m) return fail(OBHIP_ERR_INVALID, "margadj_full: basis belongs to another model");
  obhip_basis &b = *const_cast<obhip_basis *>(bc);
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  const uint64_t p = t.p, d = m->d, nh = m->nhyp();
  const bool grads = gradhyp || gradpara;
  DevBuf<double> dH;
  OB_TRY(dH.upload(H, p * p));
  PostFactor f;
  OB_TRY(post_factor_build(dH.p, p, f, grads));
  std::vector<double> ld(p);
  OB_HIP(hipMemcpy2DAsync(ld.data(), sizeof(double), f.L.p, (p + 1) * sizeof(double), sizeof(double), p,
                          hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  double logdet = 0;
  for (double v : ld) logdet += 2.0 * std::log(v);
  *val = -0.5 * logdet;
  if (!grads) return 0;

  // inv(H) = L^-T L^-1 = Linv^T Linv with Linv = X^T (row-major, k slow): one Gram-shaped product
  const uint64_t pp = f.pp, npad = pad128(b.n);
  DevBuf<double> Linv, Hinv;
  OB_TRY(Linv.alloc(pp * pp));
  OB_TRY(Hinv.alloc(pp * pp));
  OB_HIP(hipMemsetAsync(Linv.p, 0, pp * pp * sizeof(double), cur_stream()));
  OB_TRY(launch_transpose(f.X.p, pp, Linv.p, pp, p));
  OB_TRY(launch_atb(2, Linv.p, pp, pp, Linv.p, pp, pp, pp, false, Hinv.p, pp));
  std::vector<double> hinv(p);  // diag(inv(H))
  OB_HIP(hipMemcpy2DAsync(hinv.data(), sizeof(double), Hinv.p, (pp + 1) * sizeof(double),
                          sizeof(double), p, hipMemcpyDeviceToHost, cur_stream()));
  // B^T and Y = inv(H) B^T, both term-major pp x npad
  OB_TRY(ensure_gradbasis(b));
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  DevBuf<double> Bcm, Ycm, Gcm, dscal;
  OB_TRY(Bcm.alloc(pp * npad));
  OB_TRY(Ycm.alloc(pp * npad));
  OB_HIP(hipMemsetAsync(Bcm.p, 0, pp * npad * sizeof(double), cur_stream()));
  OB_TRY(launch_getmat(b, t, Bcm.p, npad));
  OB_TRY(launch_atb(2, Hinv.p, pp, pp, Bcm.p, npad, npad, pp, false, Ycm.p, npad));
  OB_TRY(dscal.alloc(nh + 1 + 4096));
  double *part = dscal.p + nh + 1;
  // tr(inv(H) B^T B) and tr(inv(H) B^T Bge_l): rows k < p, the n real columns of each
  OB_TRY(launch_dot_cols(Bcm.p, Ycm.p, npad, b.n, p, dscal.p + nh, part));
  OB_TRY(Gcm.alloc(pp * npad));
  for (uint64_t h = 0; h < nh; ++h) {
    OB_TRY(launch_getmat(*b.grad->gb, *grad_view(t, b, h), Gcm.p, npad));
    OB_TRY(launch_dot_cols(Gcm.p, Ycm.p, npad, b.n, p, dscal.p + h, part));
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
