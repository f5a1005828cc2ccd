/* A plain-C caller of libobhip.so: fit + predict on one problem through the HOST-buffer
 * entry points of include/obhip.h, with no Python marshalling in between -- the same calls
 * the Rcpp glue (glue/obhip_glue.cpp) makes.  tests/test_cabi.py builds this file with gcc,
 * hands it a golden fixture as a flat binary file and compares what it writes with the
 * fixture's expected values.
 *
 *   cabi_smoke <in.bin> <out.bin>
 *
 * in.bin : u64 d, M, mmax, nhyp, n, p, nnew; u64 kinds[d]; u64 knotptst[d + 1];
 *          f64 knotpt[M], hyp[nhyp], rotmat[mmax * M] (column-major), basisvar[M];
 *          i64 maxlevel[d]; f64 x[n * d] (column-major); u64 terms[p * d] (column-major);
 *          f64 y[n], xnew[nnew * d]
 * out.bin: f64 theta[p] (optnewton), mean[nnew], var_std[nnew], theta2[p] (obhip_fit_newton),
 *          mean2[nnew] (obhip_predict), theta_cg[p] (optcg, 12 iterations), var_gauss[nnew],
 *          f64 cg_iters, f64 val, f64 Ba[n] (obhip_basis_mm with a = theta)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "obhip.h"

#define CK(call)                                                                    \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ != 0) {                                                                 \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_,    \
              obhip_last_error());                                                  \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static void *rd(FILE *f, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return p;
}

int main(int argc, char **argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint64_t *hd = rd(f, 7 * 8);
  const uint64_t d = hd[0], M = hd[1], mmax = hd[2], nhyp = hd[3], n = hd[4], p = hd[5], nnew = hd[6];
  uint64_t *kinds64 = rd(f, d * 8), *knotptst = rd(f, (d + 1) * 8);
  double *knotpt = rd(f, M * 8), *hyp = rd(f, nhyp * 8), *rotmat = rd(f, mmax * M * 8);
  double *basisvar = rd(f, M * 8);
  int64_t *maxlevel = rd(f, d * 8);
  double *x = rd(f, n * d * 8);
  uint64_t *terms = rd(f, p * d * 8);
  double *y = rd(f, n * 8), *xnew = rd(f, nnew * d * 8);
  fclose(f);

  int ndev = 0;
  CK(obhip_device_count(&ndev));
  if (ndev < 1) {
    fprintf(stderr, "no HIP device\n");
    return 3;
  }
  if (obhip_abi_version() != 5) {
    fprintf(stderr, "unexpected ABI version %d\n", obhip_abi_version());
    return 1;
  }

  /* outermod: setcovfs, updatehyp, setknot (+ the fixture's eigen-rotation) */
  int *kinds = malloc(d * sizeof(int));
  for (uint64_t l = 0; l < d; ++l) kinds[l] = (int)kinds64[l];
  obhip_model *om = NULL;
  CK(obhip_model_create(&om, d, kinds));
  CK(obhip_model_set_hyp(om, hyp, nhyp));
  CK(obhip_model_set_knots(om, knotptst, knotpt));
  CK(obhip_model_set_rotation(om, rotmat, basisvar, maxlevel));
  uint64_t dd, MM, mm, nh;
  CK(obhip_model_dims(om, &dd, &MM, &mm, &nh));
  if (dd != d || MM != M || mm != mmax || nh != nhyp) {
    fprintf(stderr, "model dims disagree with the fixture\n");
    return 1;
  }
  /* the model's own deterministic term selection reproduces the fixture's terms */
  uint64_t *sel = malloc(p * d * 8);
  CK(obhip_model_select_terms(om, p, 0, sel));
  if (memcmp(sel, terms, p * d * 8) != 0) {
    fprintf(stderr, "selectterms differs from the fixture\n");
    return 1;
  }

  double *theta = malloc(p * 8), *mean = malloc(nnew * 8), *var_std = malloc(nnew * 8);
  double *theta2 = malloc(p * 8), *mean2 = malloc(nnew * 8), *theta_cg = malloc(p * 8);
  double *var_gauss = malloc(nnew * 8), *Ba = malloc(n * 8);

  /* lpdfvec(loglik_std, logpr_gauss)$optnewton(); predictor$update / mean / var */
  obhip_lpdf *lik = NULL, *pr = NULL, *lp = NULL;
  CK(obhip_loglik_create(&lik, OBHIP_LPDF_LOGLIK_STD, om, terms, p, y, x, n, n));
  CK(obhip_logpr_gauss_create(&pr, om, terms, p));
  CK(obhip_lpdfvec_create(&lp, lik, pr));
  CK(obhip_lpdf_optnewton(lp));
  uint64_t len = 0;
  CK(obhip_lpdf_get_vec(lp, OBHIP_VEC_COEFF, theta, p, &len));
  if (len != p) return 1;
  double val = 0;
  CK(obhip_lpdf_get_val(lp, &val));
  obhip_predictor *pd = NULL;
  CK(obhip_predictor_create(&pd, lp));
  CK(obhip_predictor_update(pd, xnew, nnew, nnew));
  CK(obhip_predictor_mean(pd, mean));
  CK(obhip_predictor_var(pd, var_std));
  CK(obhip_predictor_destroy(pd));

  /* the fused host-buffer entry points: outerbase + obhip_fit_newton + obhip_predict */
  double sigma = 0, rho = 0;
  CK(obhip_lpdf_get_vec(lik, OBHIP_VEC_PARA, &sigma, 1, NULL));
  CK(obhip_lpdf_get_vec(pr, OBHIP_VEC_PARA, &rho, 1, NULL));
  obhip_terms *t = NULL;
  obhip_basis *b = NULL;
  CK(obhip_terms_create(&t, om, terms, p));
  CK(obhip_basis_create(&b, om, x, n, n, NULL));
  CK(obhip_fit_newton(b, t, om, y, sigma, rho, theta2, NULL, NULL));
  CK(obhip_predict(om, t, theta2, xnew, nnew, nnew, mean2, NULL, sigma, NULL));
  CK(obhip_basis_mm(b, t, theta, 1, Ba));

  /* lpdfvec(logpr_gauss, loglik_gauss)$optcg(0, 12); pred_gauss variance */
  obhip_lpdf *likg = NULL, *lpg = NULL;
  uint64_t iters = 0;
  CK(obhip_loglik_create(&likg, OBHIP_LPDF_LOGLIK_GAUSS, om, terms, p, y, x, n, n));
  CK(obhip_lpdfvec_create(&lpg, pr, likg));
  CK(obhip_lpdf_optcg(lpg, 0.0, 12, &iters));
  CK(obhip_lpdf_get_vec(lpg, OBHIP_VEC_COEFF, theta_cg, p, NULL));
  CK(obhip_predictor_create(&pd, lpg));
  CK(obhip_predictor_update(pd, xnew, nnew, nnew));
  CK(obhip_predictor_var(pd, var_gauss));
  CK(obhip_predictor_destroy(pd));

  /* error convention: a status code and a message, nothing thrown */
  if (obhip_lpdf_update(lp, theta, p + 1) != OBHIP_ERR_INVALID || !obhip_last_error()[0]) {
    fprintf(stderr, "wrong-size update was not refused\n");
    return 1;
  }

  f = fopen(argv[2], "wb");
  if (!f) return 2;
  const double it = (double)iters;
  fwrite(theta, 8, p, f);
  fwrite(mean, 8, nnew, f);
  fwrite(var_std, 8, nnew, f);
  fwrite(theta2, 8, p, f);
  fwrite(mean2, 8, nnew, f);
  fwrite(theta_cg, 8, p, f);
  fwrite(var_gauss, 8, nnew, f);
  fwrite(&it, 8, 1, f);
  fwrite(&val, 8, 1, f);
  fwrite(Ba, 8, n, f);
  fclose(f);

  CK(obhip_lpdf_destroy(lpg));
  CK(obhip_lpdf_destroy(likg));
  CK(obhip_lpdf_destroy(lp));
  CK(obhip_lpdf_destroy(pr));
  CK(obhip_lpdf_destroy(lik));
  CK(obhip_basis_destroy(b));
  CK(obhip_terms_destroy(t));
  CK(obhip_model_destroy(om));
  printf("cabi_smoke ok: n=%llu p=%llu, %llu CG iterations\n", (unsigned long long)n,
         (unsigned long long)p, (unsigned long long)iters);
  return 0;
}
