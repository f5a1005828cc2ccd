/*
 * obhip.h -- C ABI of the MI355X-native outerbase hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one piece of
 * the reference's Rcpp-module surface (RCPP_MODULE(obmod),
 * src/interfaceR.cpp:661-793 of MattPlumlee/outerbase) or of the C++ objects
 * behind it.  The reference interface each function replaces is cited as
 * file:line relative to the reference checkout.  INTEGRATION.md shows the
 * Rcpp glue a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C, opaque handles, caller owns every host buffer, the library
 *     owns every device buffer behind a handle;
 *   - every function returns 0 on success, non-zero on failure; the message
 *     is in obhip_last_error() (thread-local).  Nothing throws across the ABI;
 *   - matrices that cross the ABI on the HOST side are column-major FP64,
 *     `terms` are column-major p x d unsigned 64-bit 0-based levels, exactly
 *     like the Armadillo mat / umat the reference marshals
 *     (SURVEY.md section 8b "Data marshalling");
 *   - functions with the suffix _dev take DEVICE pointers (HBM resident
 *     inputs/outputs) and enqueue on the stream set by obhip_set_stream();
 *     they do not synchronise.  Functions without the suffix take host
 *     pointers and return after the result is in the host buffer;
 *   - there is NO CPU fallback: without a visible gfx950 device every
 *     device-touching call fails with OBHIP_ERR_NO_DEVICE;
 *   - handles are NOT re-entrant: like the reference (one R main thread,
 *     SURVEY.md section 8b "Threading") a model / basis / terms / lpdf /
 *     communicator is used by one thread at a time.  Calls on a `const`
 *     handle may still fill caches behind it (device tables of the terms, the
 *     staged design matrix and task tables of a basis, the prior precisions);
 *     different handles may be used from different threads concurrently;
 *   - collective calls (anything taking an obhip_comm with more than one rank)
 *     must be made by every rank with the same sizes: an argument error on one
 *     rank returns on that rank while its peers wait in the collective.
 */
#ifndef OBHIP_H
#define OBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OBHIP_OK 0
#define OBHIP_ERR_INVALID 1     /* bad argument (std::range_error / Armadillo
                                   size error in the reference) */
#define OBHIP_ERR_NO_DEVICE 2   /* no HIP device / kernels unavailable */
#define OBHIP_ERR_HIP 3         /* a HIP runtime call failed */
#define OBHIP_ERR_STATE 4       /* object not ready (e.g. knots not set) */
#define OBHIP_ERR_NUMERIC 5     /* non-finite / not positive definite */

/* covariance kinds: listcov() R/fitting.R:6-8, setcovfs
 * src/interfaceR.cpp:53-73 */
#define OBHIP_COV_MAT25 0
#define OBHIP_COV_MAT25POW 1
#define OBHIP_COV_MAT25ANG 2

typedef struct obhip_model obhip_model; /* class outermod, modandbase.h:9-54 */
typedef struct obhip_basis obhip_basis; /* class outerbase, modandbase.h:57-125 */
typedef struct obhip_terms obhip_terms; /* a umat `terms` resident on device */
typedef struct obhip_lpdf obhip_lpdf;   /* class lpdf and descendants, fit.h:23-361 */
typedef struct obhip_predictor obhip_predictor; /* class predictor, fit.h:352-361 */
typedef struct obhip_comm obhip_comm;   /* the ranks of a row-sharded job (no reference
                                           counterpart: the reference is one process,
                                           modandbase.cpp:464) */

/* ---- library ----------------------------------------------------------- */
/* 5.  (2 -> 3: obhip_standardise_dev, obhip_destandardise_dev, obhip_fit_newton_count,
 * obhip_fit_newton_sharded_dev, obhip_source_hash added; 3 -> 4: obhip_comm_selftest_dev,
 * obhip_comm_exchange_path, obhip_comm_init_sim added, obhip_standardise_dev accepts an empty
 * shard when it has a communicator; 4 -> 5: obhip_terms_share_tables added; nothing removed.) */
int obhip_abi_version(void);
const char *obhip_last_error(void);
/* 16 hex digits of the SHA-256 over the library's sources at build time: which = 0 all of
 * them, 1 the files the Gram kernel is built from.  Profiles record it, and bench.py prints
 * PMC counters only when they were collected from the kernel that is running. */
const char *obhip_source_hash(int which);
int obhip_device_count(int *count);
int obhip_set_device(int device);
/* hipStream_t to launch on (NULL = default stream). */
int obhip_set_stream(void *hip_stream);
int obhip_synchronize(void);
/* Device temporaries of the calls below are recycled through a size-keyed pool (at most
 * OBHIP_POOL_MB, default 8192, of cached memory); this hands the cached blocks back to the
 * driver. */
int obhip_trim_pool(void);
/* Per-kernel hipEvent timing (used by bench.py for the roofline line).  obhip_profile_get: the
 * launches and device time of one profiled scope ("gram", "cholesky", "hessmult", ...); "*" = all
 * scopes together; "host_syncs" = the number of times the library has blocked on the device
 * since it was loaded (launches; total_ms = 0): the host round trips of a call sequence are the
 * difference of two readings. */
int obhip_profile_enable(int on);
int obhip_profile_reset(void);
int obhip_profile_get(const char *kernel, uint64_t *launches, double *total_ms);

/* ---- covariance functions (classes covf_mat25 / covf_mat25pow /
 * covf_mat25ang, src/covfuncs.h:34-67; module rows interfaceR.cpp:764-791) */
int obhip_cov_numhyp(int kind, int *numhyp);
/* hyp0, hyplb, hypub, hypvar (numhyp each), lowbnd, uppbnd
 * (covfuncs.cpp:87-111,166-195,254-283) */
int obhip_cov_info(int kind, double *hyp0, double *hyplb, double *hypub,
                   double *hypvar, double *lowbnd, double *uppbnd);
/* covf::cov (covfuncs.cpp:113-126,197-212,285-310): out is n1 x n2
 * column-major.  Host arithmetic (knot-sized problems). */
int obhip_cov(int kind, const double *hyp, const double *x1, uint64_t n1,
              const double *x2, uint64_t n2, double *out);
/* covf::cov_gradhyp (covfuncs.cpp:134-150,220-243,318-347; module row interfaceR.cpp:775):
 * out is n1 x n2 x numhyp, column-major slices.  Host arithmetic. */
int obhip_cov_gradhyp(int kind, const double *hyp, const double *x1, uint64_t n1,
                      const double *x2, uint64_t n2, double *out);
/* covf::lpdf (covfuncs.cpp:35-50) */
int obhip_cov_hyplpdf(int kind, const double *hyp, double *out);

/* ---- outermod ---------------------------------------------------------- */
/* new(outermod) + setcovfs(om, covnames): interfaceR.cpp:53-73,
 * outermod::hyp_init modandbase.cpp:128-153 */
int obhip_model_create(obhip_model **out, uint64_t d, const int *kinds);
int obhip_model_destroy(obhip_model *m);
/* setknot(om, knotlist): interfaceR.cpp:94-149.  knotptst has d+1 entries
 * (modandbase.h:18), knotpt has knotptst[d]. Triggers outermod::build. */
int obhip_model_set_knots(obhip_model *m, const uint64_t *knotptst,
                          const double *knotpt);
/* the knots as set (fields knotptst / knotpt, modandbase.h:18-19); either pointer may be NULL */
int obhip_model_get_knots(const obhip_model *m, uint64_t *knotptst, double *knotpt);
/* om$updatehyp(hyp): outermod::hyp_set modandbase.cpp:161-202 */
int obhip_model_set_hyp(obhip_model *m, const double *hyp, uint64_t nhyp);
/* gethyp(om): interfaceR.cpp:167-180 */
int obhip_model_get_hyp(const obhip_model *m, double *hyp);
/* d, M = total knots, mmax = max knots per dim, nhyp */
int obhip_model_dims(const obhip_model *m, uint64_t *d, uint64_t *M,
                     uint64_t *mmax, uint64_t *nhyp);
/* results of outermod::build (modandbase.cpp:210-276): rotmat is
 * mmax x M column-major (modandbase.h:44), basisvar M (:12), maxlevel d
 * (:29). */
int obhip_model_get_rotation(const obhip_model *m, double *rotmat,
                             double *basisvar, int64_t *maxlevel);
/* Replace the eigen-decomposition results with caller-supplied ones (the
 * reference takes them from LAPACK via arma::eig_sym, modandbase.cpp:236;
 * parity tests inject the oracle's so that both sides share one rotation). */
int obhip_model_set_rotation(obhip_model *m, const double *rotmat,
                             const double *basisvar, const int64_t *maxlevel);
/* Hyper-parameter gradient layout (outermod::hyp_set, modandbase.cpp:183-197): nhyp
 * hyper-parameters in all; hyper-parameter h belongs to dimension hypmatch[h] and owns
 * the column block [gest[h], gest[h+1]) (m_l columns) of the *_gradhyp arrays.
 * hypmatch: nhyp entries, gest: nhyp + 1; any pointer may be NULL. */
int obhip_model_grad_layout(const obhip_model *m, uint64_t *nhyp, uint64_t *hypmatch,
                            uint64_t *gest);
/* gradient part of outermod::build (modandbase.cpp:257-274): rotmat_gradhyp is
 * mmax x gest[nhyp] column-major (modandbase.h:45), logbasisvar_gradhyp gest[nhyp] */
int obhip_model_get_rotation_grad(const obhip_model *m, double *rotmat_gradhyp,
                                  double *logbasisvar_gradhyp);
/* counterpart of obhip_model_set_rotation for the gradient arrays */
int obhip_model_set_rotation_grad(obhip_model *m, const double *rotmat_gradhyp,
                                  const double *logbasisvar_gradhyp);
/* om$getlvar_gradhyp(terms): modandbase.cpp:364-379; out p x nhyp column-major */
int obhip_model_term_lvar_gradhyp(const obhip_model *m, const uint64_t *terms, uint64_t p,
                                  double *out);
/* om$selectterms(numele): modandbase.cpp:387-440.  seed==0: the reference's
 * RNG shuffle (modandbase.cpp:408) is replaced by the identity permutation;
 * seed!=0: SplitMix64-driven pick among the near-best candidates.
 * terms_out: p x d column-major. */
int obhip_model_select_terms(const obhip_model *m, uint64_t p, uint64_t seed,
                             uint64_t *terms_out);
/* om$getvar(terms): modandbase.cpp:350-356 */
int obhip_model_term_var(const obhip_model *m, const uint64_t *terms,
                         uint64_t p, double *out);
/* om$hyplpdf(hyp): modandbase.cpp:89-99 */
int obhip_model_hyplpdf(const obhip_model *m, const double *hyp, uint64_t nhyp,
                        double *out);

/* om$hyplpdf_grad(hyp): modandbase.cpp:106-118 (covf::lpdf_gradhyp, covfuncs.cpp:53-70) */
int obhip_model_hyplpdf_grad(const obhip_model *m, const double *hyp, uint64_t nhyp,
                             double *out);

/* ---- terms ------------------------------------------------------------- */
/* Upload a umat `terms` (p x d, column-major, 0-based levels) once; the
 * reference passes it by value on every call (modandbase.cpp:649,677,700). */
int obhip_terms_create(obhip_terms **out, const obhip_model *m,
                       const uint64_t *terms, uint64_t p);
int obhip_terms_destroy(obhip_terms *t);
int obhip_terms_info(const obhip_terms *t, uint64_t *p, uint64_t *d,
                     uint64_t *nnz_total, uint64_t *max_nnz);
/* highest level used per dimension (d entries) */
int obhip_terms_maxlevels(const obhip_terms *t, int64_t *levels);
/* How the term-per-lane kernels share sub-products among these terms (host only, no device
 * needed).  The reference multiplies every term out on its own (prodmm_ / tprodmm_,
 * linalg.cpp:57-131,286-355); selectterms' output is downward-closed (modandbase.cpp:419-436), so
 * its terms come in families that differ in one factor, and the library groups the p_pad = p
 * rounded up to 256 terms (the padding terms have no factors) into *stars*: four terms whose P
 * shared factors are read and multiplied once (csrc/share.cpp).  Terms that find no family with
 * four free members are *left over* (any term set is accepted; one that is not downward-closed
 * just leaves more over): they are listed, and also packed four at a time into plain stars
 * (nothing shared) behind the family stars.  Stars come in star-waves of 64, filled up with empty
 * stars.  info (11 entries):
 *   [0] p_pad, [1] left-over terms, [2] column reads per row of the family star-waves (sum of
 *   P + 4), [3] of the scheme without sharing (4 terms per lane, terms by falling number of
 *   factors: rounds 1-4), [4] W, the column slots per term, [5] LDS cycles of all star-waves' reads
 *   with their bank conflicts (2 per read at best; column u of the [column][65] tile lies in bank
 *   pair u mod 32), [6] the same before the search over the stars' half-waves and term orders that
 *   minimises them, [7] family star-waves, [8] plain star-waves, [9] reads per row of the latter, [10] the LDS
 *   cycles of [5] once the library has also chosen which tile index (bank pair) every used column
 *   gets -- what it uploads; the tables returned here keep the level-based ids (11 entries).
 * With S = 64 ([7] + [8]) stars in all -- S <= p_pad / 4 + 128 -- the optional outputs are:
 * term (4 S entries): the four term indices of star 0, of star 1, ... (0xffffffff: none);
 * shape ([7] + [8] entries): per star-wave P | S << 8 (family star-waves first: S = 1);
 * factor (S * 4 W entries): per star its column slots in read order -- [P shared][own of term 0]
 * [own of term 1] ... -- as 1 + (levels of the dimensions before l) + (level - 1) for level >= 1
 * of dimension l (levels counted up to obhip_terms_maxlevels), 0 = unused;
 * left ([1] entries): the left-over terms. */
int obhip_terms_share_tables(const obhip_terms *t, uint64_t *info, uint32_t *term, uint32_t *shape,
                             uint16_t *factor, uint32_t *left);

/* ---- outerbase --------------------------------------------------------- */
/* new(outerbase, om, x): modandbase.cpp:459-480 + build :547-626.
 * x is n x d column-major with leading dimension ldx (host).  levelcap
 * (d entries, or NULL) bounds the levels evaluated per dimension; NULL = all
 * knots-1 levels as the reference does.  The basis is built on the current
 * device. */
int obhip_basis_create(obhip_basis **out, const obhip_model *m, const double *x,
                       uint64_t n, uint64_t ldx, const int64_t *levelcap);
/* same with x already in HBM (column-major, ld = n) */
int obhip_basis_create_dev(obhip_basis **out, const obhip_model *m,
                           const double *d_x, uint64_t n,
                           const int64_t *levelcap);
/* ob$build(): rebuild after the model's hyp/knots changed
 * (modandbase.cpp:547; vignettes/learning.Rmd:48-54) */
int obhip_basis_rebuild(obhip_basis *b);
int obhip_basis_destroy(obhip_basis *b);
int obhip_basis_dims(const obhip_basis *b, uint64_t *n, uint64_t *d,
                     uint64_t *ncols_stored);
/* ob$getbase(k), k 1-based: modandbase.cpp:634-639; out n x m_k col-major */
int obhip_basis_getbase(const obhip_basis *b, uint64_t k, double *out);
/* ob$getmat(terms): getm_ linalg.cpp:647-715; out n x p col-major */
int obhip_basis_getmat(const obhip_basis *b, const obhip_terms *t, double *out);
/* ob$matmul(terms, a) / outerbase::mm: prodmm_ linalg.cpp:57-131 (vector)
 * and :481-557 (matrix, ncol > 1; a is p x ncol, out n x ncol) */
int obhip_basis_mm(const obhip_basis *b, const obhip_terms *t, const double *a,
                   uint64_t ncol, double *out);
/* ob$tmatmul(terms, a) / outerbase::tmm: tprodmm_ linalg.cpp:286-355 and
 * :567-637 (a is n x ncol, out p x ncol) */
int obhip_basis_tmm(const obhip_basis *b, const obhip_terms *t, const double *a,
                    uint64_t ncol, double *out);
/* outerbase::sqmm modandbase.cpp:784-790, sqtmm/sqtmmm :816-837,
 * sqcolsums :863-867, residvar :889-895 */
int obhip_basis_sqmm(const obhip_basis *b, const obhip_terms *t,
                     const double *a, uint64_t ncol, double *out);
int obhip_basis_sqtmm(const obhip_basis *b, const obhip_terms *t,
                      const double *a, uint64_t ncol, double *out);
int obhip_basis_sqcolsums(const obhip_basis *b, const obhip_terms *t,
                          double *out);
int obhip_basis_residvar(const obhip_basis *b, const obhip_terms *t,
                         const obhip_model *m, double *out);

/* device-pointer forms (single vector) used by the fit drivers, the
 * benchmark and the multi-GPU path */
int obhip_basis_mm_dev(const obhip_basis *b, const obhip_terms *t,
                       const double *d_a, double *d_out, int squared);
int obhip_basis_tmm_dev(const obhip_basis *b, const obhip_terms *t,
                        const double *d_a, double *d_out, int squared);

/* ---- hyper-parameter gradients ------------------------- */
/* The gradient basis (outerbase::build with dograd, modandbase.cpp:547-626) is built on
 * the first call and kept until the basis is rebuilt.  nhyp and the block layout come
 * from obhip_model_grad_layout.  The basis must have been built after the last change of
 * the model (OBHIP_ERR_STATE otherwise).
 * ob$getmat_gradhyp(terms): modandbase.cpp:663-669 (getmge_, linalg.cpp:778-822);
 * out: n x p x nhyp, column-major slices. */
int obhip_basis_getmat_gradhyp(const obhip_basis *b, const obhip_terms *t, double *out);
/* ob$matmul_gradhyp(terms, a): modandbase.cpp:725-744 (prodmmge_, linalg.cpp:219-276);
 * out (n, may be NULL) = B a, out_gradhyp n x nhyp column-major. */
int obhip_basis_mm_gradhyp(const obhip_basis *b, const obhip_terms *t, const double *a,
                           double *out, double *out_gradhyp);
/* w^T d(B a)/dhyp (nhyp values) with the n x nhyp matrix left on the device: the form the
 * likelihoods use matmul_gradhyp in (src/lpdfs/loglik_gauss.cpp:127, loglik_std.cpp:143).
 * out (n, B a) may be NULL. */
int obhip_basis_mm_gradhyp_dot(const obhip_basis *b, const obhip_terms *t, const double *a,
                               const double *w, double *out, double *out_dot);
/* ob$tmatmul_gradhyp(terms, a): modandbase.cpp:755-776 (tprodmmge_, linalg.cpp:395-471);
 * out (p, may be NULL) = B^T a, out_gradhyp p x nhyp column-major. */
int obhip_basis_tmm_gradhyp(const obhip_basis *b, const obhip_terms *t, const double *a,
                            double *out, double *out_gradhyp);

/* ob$sqmm_gradhyp(terms, a) / ob$sqtmm_gradhyp(terms, a): the same on the squared stores
 * basematsq / basescalesq / basematsq_gradhyp (modandbase.cpp:798-809, 845-856);
 * out_gradhyp n x nhyp resp. p x nhyp, column-major */
int obhip_basis_sqmm_gradhyp(const obhip_basis *b, const obhip_terms *t, const double *a,
                             double *out_gradhyp);
int obhip_basis_sqtmm_gradhyp(const obhip_basis *b, const obhip_terms *t, const double *a,
                              double *out_gradhyp);
/* ob$sqcolsums_gradhyp(terms): modandbase.cpp:875-879; p x nhyp */
int obhip_basis_sqcolsums_gradhyp(const obhip_basis *b, const obhip_terms *t, double *out_gradhyp);
/* ob$residvar_gradhyp(terms): modandbase.cpp:904-925; n x nhyp */
int obhip_basis_residvar_gradhyp(const obhip_basis *b, const obhip_terms *t, const obhip_model *m,
                                 double *out_gradhyp);

/* ---- Gram / Newton ("back end A") -------------------------------------- */
/* G = B^T B (loglik_std::hess without its e^{-2 sigma}, loglik_std.cpp:
 * 170-173) and g = B^T y (loglik_std::update at coeff = 0, :100-120).
 * d_G: p x p (symmetric, full storage), d_g: p, both DEVICE buffers owned by
 * the caller so that a multi-GPU caller can all-reduce them.  d_y may be
 * NULL (then d_g is untouched). */
int obhip_gram_dev(const obhip_basis *b, const obhip_terms *t,
                   const double *d_y, double *d_G, double *d_g);
/* Which kernel forms G: 0 / 4 = FP64 matrix cores (v_mfma_f64_4x4x4_4b_f64) fed from a
 * row-major design matrix staged in HBM -- all n_pad x p_pad doubles at once (kept with the
 * basis) when that fits in half of the free memory, in row chunks otherwise; 3 = the same
 * matrix-core tiles with the operand panels generated inside the kernel (no staging memory;
 * terms of at most 8 factors on at most 128 used basis columns).  Both give the same G up to
 * summation order; DESIGN.md has the measurements. */
int obhip_set_gram_backend(int backend);
/* bytes of device workspace obhip_newton_solve_dev needs for p terms */
int obhip_newton_workspace_bytes(uint64_t p, uint64_t *bytes);
/* One Newton step from coeff = 0 of lpdfvec(loglik_std, logpr_gauss)
 * (lpdf::optnewton fit.cpp:98-131): H = e^{-2 sigma} G + diag(1/(sd e^rho)^2)
 * (loglik_std.cpp:170-173, logpr_gauss.cpp:153-158, fit.cpp:503-512),
 * theta = solve(H, e^{-2 sigma} g) by Cholesky + two triangular solves.
 * d_G is overwritten by the Cholesky factor (lower triangle, row-major).
 * d_theta: p (device).  d_diagH (p, may be NULL) receives diag(H)
 * (lpdfvec::diaghess_, fit.cpp:557-566). */
int obhip_newton_solve_dev(const obhip_model *m, const obhip_terms *t,
                           double *d_G, const double *d_g, double sigma,
                           double rho, double *d_theta, double *d_diagH,
                           void *d_workspace, uint64_t workspace_bytes);
/* host-buffer convenience: the whole back end A on one device.
 * theta (p), diagH (p, may be NULL), H_out (p x p col-major, may be NULL). */
int obhip_fit_newton(const obhip_basis *b, const obhip_terms *t,
                     const obhip_model *m, const double *y, double sigma,
                     double rho, double *theta, double *diagH, double *H_out);

/* ---- matrix-free PCG ("back end B", what obfit runs) -------------------- */
/* lpdf::optcg (fit.cpp:37-96) on lpdfvec(logpr_gauss, loglik_gauss)
 * (loglik_gauss.cpp:110-157), domargadj = false.  theta is in/out (host, p);
 * iters_out receives the iteration count; diagH (p, may be NULL) the
 * preconditioner (lpdfvec::diaghess_); val_out (may be NULL) the value of the
 * objective at the result -- asking for it costs one more evaluation (two passes
 * over the basis) when the last iteration advanced it by the recurrence. */
int obhip_fit_cg(const obhip_basis *b, const obhip_terms *t,
                 const obhip_model *m, const double *y, double sigma,
                 double rho, double tol, uint64_t maxit, double *theta,
                 uint64_t *iters_out, double *diagH, double *val_out);
/* same with y / theta / diagH in HBM.  comm (may be NULL = one rank): the rows are sharded
 * over comm's ranks and every B^T a pass sums one p-vector (+ 2 scalars) over them
 * (SURVEY.md section 8e); all ranks return the same theta. */
int obhip_fit_cg_dev(const obhip_basis *b, const obhip_terms *t,
                     const obhip_model *m, const double *d_y, double sigma,
                     double rho, double tol, uint64_t maxit, double *d_theta,
                     uint64_t *iters_out, double *d_diagH, double *val_out,
                     obhip_comm *comm);

/* ---- multi-GPU: rows sharded over ranks, one process per GPU ------------
 * No reference counterpart (one process, OpenMP threads: modandbase.cpp:464,480); this is
 * the partitioning BASELINE.json's north_star prescribes.  Every rank holds a contiguous row
 * block of x / y and its own outerbase; outermod and terms are replicated.  Back end A
 * exchanges ONE buffer per fit, back end B one p-vector per B^T a pass; prediction needs no
 * communication. */
#define OBHIP_UNIQUE_ID_BYTES 128
#define OBHIP_TRANSPORT_NONE 0
#define OBHIP_TRANSPORT_RCCL 1 /* ncclReduceScatter + ncclAllGather over xGMI */
#define OBHIP_TRANSPORT_HOST 2 /* caller-supplied sum of a host buffer (MPI, gloo) */
#define OBHIP_TRANSPORT_SIM 3  /* N virtual ranks holding this rank's shard: sum = N x, on the device */
/* what sums a buffer of a given size (obhip_comm_exchange_path) */
#define OBHIP_EXCHANGE_NONE 0
#define OBHIP_EXCHANGE_PAIR 1      /* ncclReduceScatter + ncclAllGather, in place */
#define OBHIP_EXCHANGE_ALLREDUCE 2 /* ncclAllReduce */
#define OBHIP_EXCHANGE_HOST 3
#define OBHIP_EXCHANGE_SIM 4
/* rank 0 draws the communicator id (ncclGetUniqueId); the launcher hands the 128 bytes to
 * every rank */
int obhip_comm_unique_id(void *id);
/* RCCL communicator over the current device of each of the nranks processes
 * (ncclCommInitRank; collective: every rank must call it).  librccl is loaded at run time. */
int obhip_comm_init(obhip_comm **out, int nranks, int rank, const void *id);
/* the same ranks with a caller-supplied transport: fn(user, host_buf, count) sums count
 * doubles in place over all ranks (MPI_Allreduce under R, gloo in the one-GPU rehearsals);
 * the library stages device buffers through pinned memory */
typedef int (*obhip_host_allreduce_fn)(void *user, double *host_buf, uint64_t count);
int obhip_comm_init_host(obhip_comm **out, int nranks, int rank, obhip_host_allreduce_fn fn,
                         void *user);
int obhip_comm_destroy(obhip_comm *c);
/* rccl_ranks: what ncclCommCount reports (0 for the host transport); rccl_version:
 * ncclGetVersion; any pointer may be NULL */
int obhip_comm_info(const obhip_comm *c, int *nranks, int *rank, int *transport, int *rccl_ranks,
                    int *rccl_version);
/* in-place sum of count doubles in HBM over the ranks, on the library's stream */
int obhip_comm_allreduce_dev(obhip_comm *c, double *d_buf, uint64_t count);
/* A communicator of nranks VIRTUAL ranks that all hold this process's shard: every sum is one
 * device pass buf *= nranks, nothing leaves the GPU.  For timing, on one GPU, the step one rank
 * of an nranks-GPU job runs (exchange-buffer layout, unpack, replicated solve); the result is
 * the fit of the shard's rows repeated nranks times. */
int obhip_comm_init_sim(obhip_comm **out, int nranks);
/* path: OBHIP_EXCHANGE_* a buffer of count doubles takes on this communicator (large buffers
 * in equal 16-byte blocks per rank: the reduce-scatter / all-gather pair, unless
 * OBHIP_RCCL_ALLREDUCE=1 was set on any rank when the communicator was made -- the flags are
 * summed over the ranks at init -- or the self-test switched it off); selftest: 0 not run,
 * 1 passed, 2 passed after switching the pair off.  Either pointer may be NULL. */
int obhip_comm_exchange_path(const obhip_comm *c, uint64_t count, int *path, int *selftest);
/* Collective self-test of the exchange on a buffer of count doubles (pass the size of the real
 * exchange, obhip_fit_newton_count): rank r fills (r + 1) w_i with small integers w_i, the sums
 * over the ranks are compared ON THE DEVICE with the closed form nranks (nranks + 1) / 2 w_i,
 * once through the reduce-scatter / all-gather pair (if a buffer of this size takes it) and
 * once through the plain all-reduce; the mismatch counts are summed over the ranks so that all
 * take the same decision.  Pair wrong, all-reduce right: the communicator switches to the
 * all-reduce for every size, in this process, and the call succeeds.  All-reduce wrong:
 * OBHIP_ERR_STATE.  result (4 values, may be NULL): OBHIP_EXCHANGE_* in use afterwards for this
 * size, mismatching elements of the pair (-1: not tried), of the all-reduce, 1 if the pair was
 * switched off by this call. */
int obhip_comm_selftest_dev(obhip_comm *c, uint64_t count, int64_t *result);
/* The one exchange of back end A.  Buffer layout (count doubles, obhip_normal_eq_count):
 * [upper triangle of G_r = B_r^T B_r, row-major packed, p (p + 1) / 2][B_r^T y_r : p]
 * [B_r^T 1 : p][sum y_r, sum y_r^2, n_r][zero padding to 2 nranks].  Pack -> sum over ranks ->
 * unpack: d_G becomes the global Gram (full symmetric storage), d_g the right-hand side of
 * the problem with y standardised over ALL rows like obfit does (R/fitting.R:55-57):
 * B^T ((y - cent) / sca) = (B^T y - cent B^T 1) / sca; d_meansd receives cent, sca, n
 * (device, 3 doubles).  comm may be NULL (one rank: only the standardisation happens and the
 * triangle part of the buffer is never touched, so d_buf may point count - tail doubles
 * before a tail-sized allocation). */
/* Exact sample quantiles (R's quantile(), type 7) of every column of a row-sharded x: what
 * obfit's knot placement needs of the data (.genknotlist, R/fitting.R:177-185).  d_x: this
 * rank's n x d rows, column-major (device); probs: q values in [0, 1] (host); out: d x q,
 * out[l * q + j] (host), identical on every rank.  comm may be NULL (one rank). */
int obhip_quantiles_dev(obhip_comm *comm, const double *d_x, uint64_t n, uint64_t d,
                        const double *probs, uint64_t q, double *out);
int obhip_normal_eq_count(uint64_t p, int nranks, uint64_t *count);
/* ---- row-sharded Newton fit, start to finish on the device (ABI 3) ----------------------
 * obfit's standardisation of y (R/fitting.R:55-57: (y - mean(y)) / sd(y), n - 1 denominator)
 * over the rows of ALL ranks, two-pass like R's sd(): (sum y, n) and sum (y - mean)^2 are
 * summed over the ranks (24 bytes in two exchanges), nothing returns to the host.
 * d_y_raw, d_y: n doubles (device; may be the same buffer); d_meansd: mean, sd, rows of all
 * ranks (device, 3 doubles).  comm may be NULL (one rank).  With a communicator n = 0 is a legal
 * shard (it still takes part in the two sums; d_y_raw / d_y may then be NULL); fewer than two
 * rows over all ranks give sd = NaN as R's sd() does. */
int obhip_standardise_dev(obhip_comm *comm, const double *d_y_raw, uint64_t n, double *d_y,
                          double *d_meansd);
/* obpred's de-standardisation (R/fitting.R:152): v = mean + sd v, in place, mean and sd read
 * from d_meansd on the device */
int obhip_destandardise_dev(double *d_v, uint64_t n, const double *d_meansd);
/* doubles of exchange buffer obhip_fit_newton_sharded_dev needs: the packed upper triangle of
 * G, then g = B^T y, padded to equal 16-byte blocks per rank */
int obhip_fit_newton_count(uint64_t p, int nranks, uint64_t *count);
/* lpdf::optnewton (fit.cpp:98-131) of lpdfvec(loglik_std, logpr_gauss) from coeff = 0 on a
 * row shard: G_r = B_r^T B_r by the Gram kernels, whose reduction writes the packed upper
 * triangle straight into d_exbuf, g_r = B_r^T y behind it; ONE sum over the ranks; the unpack
 * forms H = e^{-2 sigma} G + diag(1 / (sd e^rho)^2) (loglik_std.cpp:170-173,
 * logpr_gauss.cpp:153-158, fit.cpp:503-512) in full symmetric storage in d_H; Cholesky and
 * the two triangular solves (replicated on every rank) give d_theta.  d_y: this rank's rows,
 * already standardised.  d_H (p x p; on return its lower triangle is the Cholesky factor),
 * d_g (p: B^T y over all ranks), d_theta (p), d_diagH (p, may be NULL), d_exbuf
 * (obhip_fit_newton_count doubles whose padding the caller zeroed once), d_workspace
 * (obhip_newton_workspace_bytes): device memory of the caller.  comm = NULL: one rank, the
 * reduction writes H itself, d_exbuf is not used. */
int obhip_fit_newton_sharded_dev(obhip_comm *comm, const obhip_basis *b, const obhip_terms *t,
                                 const obhip_model *m, const double *d_y, double sigma, double rho,
                                 double *d_H, double *d_g, double *d_theta, double *d_diagH,
                                 double *d_exbuf, uint64_t exbuf_count, void *d_workspace,
                                 uint64_t workspace_bytes);
int obhip_normal_eq_exchange_dev(obhip_comm *comm, uint64_t p, uint64_t n_local, double *d_G,
                                 double *d_g, const double *d_b1, const double *d_sum2,
                                 double *d_buf, uint64_t buf_count, double *d_meansd);

/* ---- predictor ---------------------------------------------------------- */
/* predictor$update(x) + $mean() (+ $var() of pred_gauss):
 * loglik_gauss.cpp:214-227, loglik_std.cpp:239-248.  The basis at xnew is
 * evaluated and contracted with theta in one fused kernel; it is never
 * written to HBM.  d_coeffvar (p) may be NULL; then d_var is untouched;
 * otherwise var = B^2 coeffvar + e^{2 sigma}. */
int obhip_predict_dev(const obhip_model *m, const obhip_terms *t,
                      const double *d_theta, const double *d_x, uint64_t n,
                      double *d_mean, const double *d_coeffvar, double sigma,
                      double *d_var);
int obhip_predict(const obhip_model *m, const obhip_terms *t,
                  const double *theta, const double *x, uint64_t n,
                  uint64_t ldx, double *mean, const double *coeffvar,
                  double sigma, double *var);
/* predr_std (loglik_std.cpp:218-256), the predictor of the loglik_std model with the full
 * posterior covariance of the coefficients: mean = B theta, var_i = b_i^T inv(H) b_i +
 * e^{2 sigma} (:249-256; the reference uses arma::inv; here H = L L^T by the library's own
 * Cholesky, X = L^-T by a blocked triangular inversion and || L^-1 b_i ||^2 in one pass of the
 * matrix-core kernel, csrc/posterior.cpp -- no BLAS library is involved).  H: total Hessian,
 * p x p symmetric (host); var may be NULL (then H may be too). */
int obhip_predict_std(const obhip_model *m, const obhip_terms *t, const double *theta,
                      const double *H, const double *x, uint64_t n, uint64_t ldx, double *mean,
                      double sigma, double *var);

/* Marginal adjustment of lpdfvec(loglik_std, logpr_gauss) with the full Hessian
 * (lpdfvec::buildhess, fit.cpp:270-299): val = -1/2 log det H, gradhyp[l] =
 * -1/2 sum(dH/dhyp_l % inv(H)) (nhyp entries) and gradpara = {noisescale, coeffscale}
 * parts (loglik_std.cpp:180-203, logpr_gauss.cpp:165-186).  H: total Hessian (host);
 * gradhyp / gradpara may be NULL.  inv(H) = L^-T L^-1 on the library's own kernels, like
 * obhip_predict_std. */
int obhip_margadj_full(const obhip_basis *b, const obhip_terms *t, const obhip_model *m,
                       const double *H, double sigma, double rho, double *val, double *gradhyp,
                       double *gradpara);

/* ---- the model layer: lpdf, loglik_*, logpr_gauss, lpdfvec, predictor ------------------
 * Module rows src/interfaceR.cpp:696-762; classes src/fit.h:23-361; arithmetic
 * src/fit.cpp:37-612 and src/lpdfs/{loglik_std,loglik_gauss,loglik_gda,logpr_gauss}.cpp.
 * One handle type for every descendant of class lpdf; y, yhat, the residuals and the
 * observation standard deviations of an object stay in HBM between calls, so a call moves
 * p-, nhyp- and npara-sized vectors only.  Objects reference each other like the C++
 * objects of the reference do (fit.h:133,153): the caller keeps the outermod alive as long
 * as a likelihood or prior built on it, and the two members as long as their lpdfvec. */
#define OBHIP_LPDF_LOGLIK_STD 0   /* class loglik_std,   src/lpdfs/loglik_std.cpp:41-203 */
#define OBHIP_LPDF_LOGLIK_GAUSS 1 /* class loglik_gauss, src/lpdfs/loglik_gauss.cpp:41-172 */
#define OBHIP_LPDF_LOGLIK_GDA 2   /* class loglik_gda,   src/lpdfs/loglik_gda.cpp:48-235 */
#define OBHIP_LPDF_LOGPR_GAUSS 3  /* class logpr_gauss,  src/lpdfs/logpr_gauss.cpp:41-186 */
#define OBHIP_LPDF_VEC 4          /* class lpdfvec,      src/fit.cpp:174-612 */
/* new(loglik_std | loglik_gauss | loglik_gda, om, terms, y, x): interfaceR.cpp:733-750.
 * terms p x d column-major, y n, x n x d column-major with leading dimension ldx (host). */
int obhip_loglik_create(obhip_lpdf **out, int kind, const obhip_model *om, const uint64_t *terms,
                        uint64_t p, const double *y, const double *x, uint64_t n, uint64_t ldx);
/* new(logpr_gauss, om, terms): interfaceR.cpp:752-756 */
int obhip_logpr_gauss_create(obhip_lpdf **out, const obhip_model *om, const uint64_t *terms,
                             uint64_t p);
/* new(lpdfvec, a, b): interfaceR.cpp:758-762, fit.cpp:174-200; para = [a.para, b.para] */
int obhip_lpdfvec_create(obhip_lpdf **out, obhip_lpdf *a, obhip_lpdf *b);
int obhip_lpdf_destroy(obhip_lpdf *l);
/* kind, nterms (field, interfaceR.cpp:709), npara, number of hyper-parameters of the model,
 * rows of the likelihood's data (0 for the prior); any pointer may be NULL */
int obhip_lpdf_dims(const obhip_lpdf *l, int *kind, uint64_t *nterms, uint64_t *npara,
                    uint64_t *nhyp, uint64_t *n);
/* boolean fields: compute_* (interfaceR.cpp:698-701; this ABI names them by what they do --
 * the reference module binds R's compute_gradpara to C++ compute_gradhyp and vice versa),
 * fullhess (read-only, :702), lpdfvec's domarg (:761), loglik_gda's dodiag (:748) */
#define OBHIP_FLAG_COMPUTE_VAL 0
#define OBHIP_FLAG_COMPUTE_GRAD 1
#define OBHIP_FLAG_COMPUTE_GRADHYP 2
#define OBHIP_FLAG_COMPUTE_GRADPARA 3
#define OBHIP_FLAG_FULLHESS 4
#define OBHIP_FLAG_DOMARG 5
#define OBHIP_FLAG_DODIAG 6
int obhip_lpdf_get_flag(const obhip_lpdf *l, int flag, int *value);
int obhip_lpdf_set_flag(obhip_lpdf *l, int flag, int value);
/* field val (interfaceR.cpp:703) */
int obhip_lpdf_get_val(const obhip_lpdf *l, double *val);
/* vector fields (interfaceR.cpp:704-708,737,743,749,755; para0 / paravar fit.h:59-60;
 * totdiaghess fit.h:33): out may be NULL to query the length */
#define OBHIP_VEC_COEFF 0
#define OBHIP_VEC_GRAD 1
#define OBHIP_VEC_GRADHYP 2
#define OBHIP_VEC_GRADPARA 3
#define OBHIP_VEC_PARA 4
#define OBHIP_VEC_PARA0 5
#define OBHIP_VEC_PARAVAR 6
#define OBHIP_VEC_TOTDIAGHESS 7
#define OBHIP_VEC_COEFFSD 8 /* logpr_gauss only */
#define OBHIP_VEC_YHAT 9    /* likelihoods only; copied from HBM */
int obhip_lpdf_get_vec(const obhip_lpdf *l, int which, double *out, uint64_t cap, uint64_t *len);
/* names of the parameters, getpara(lpdf): interfaceR.cpp:193-199 */
int obhip_lpdf_paraname(const obhip_lpdf *l, uint64_t i, const char **name);
/* the umat `terms` of the object (fit.h:31), p x d column-major */
int obhip_lpdf_terms(const obhip_lpdf *l, uint64_t *terms_out);
/* the outerbase a likelihood owns (member `ob`, fit.h:185,237,273) and the device form of
 * its terms; borrowed handles, valid until the next updateterms / destroy */
int obhip_lpdf_basis(obhip_lpdf *l, obhip_basis **b, obhip_terms **t);
/* Rows sharded over the ranks of comm (no reference counterpart): the
 * likelihood (loglik_gauss | loglik_std; given an lpdfvec, its likelihood) holds this
 * rank's rows, and every sum over rows -- val, grad, gradhyp, gradpara, hessmult,
 * diaghess*, hess, optcg, optnewton -- is summed over the ranks, so that all ranks see the
 * numbers of the whole data set; para0 = log(0.01 var(y)) takes var(y) over all rows.
 * Collective: every rank calls it, and afterwards the same methods in the same order.
 * loglik_gda refuses (obfit runs it on a subsample every rank holds); NULL detaches. */
int obhip_lpdf_set_comm(obhip_lpdf *l, obhip_comm *comm);
/* lpdf$setnthreads (interfaceR.cpp:710): accepted and ignored on the device */
int obhip_lpdf_setnthreads(obhip_lpdf *l, int nthreads);
/* lpdf$update(coeff): loglik_gauss.cpp:110-130, loglik_std.cpp:100-120, loglik_gda.cpp:117-153,
 * logpr_gauss.cpp:113-121, lpdfvec fit.cpp:323-380 (marginal adjustment included) */
int obhip_lpdf_update(obhip_lpdf *l, const double *coeff, uint64_t ncoeff);
/* lpdf$updateom() / updatepara(para) / updateterms(terms): interfaceR.cpp:714-716 */
int obhip_lpdf_updateom(obhip_lpdf *l);
int obhip_lpdf_updatepara(obhip_lpdf *l, const double *para, uint64_t npara);
int obhip_lpdf_updateterms(obhip_lpdf *l, const uint64_t *terms, uint64_t p);
/* lpdf$hessmult(g) -> p; diaghess() -> p; diaghessgradhyp() -> p x nhyp;
 * diaghessgradpara() -> p x npara (column-major): interfaceR.cpp:717-720 */
int obhip_lpdf_hessmult(obhip_lpdf *l, const double *g, double *out);
int obhip_lpdf_diaghess(obhip_lpdf *l, double *out);
int obhip_lpdf_diaghessgradhyp(obhip_lpdf *l, double *out);
int obhip_lpdf_diaghessgradpara(obhip_lpdf *l, double *out);
/* hess() (fit.h:81; loglik_std.cpp:170-173, logpr_gauss.cpp:167-172, lpdfvec::hess_
 * fit.cpp:503-512): p x p, formed on the device by the Gram kernels */
int obhip_lpdf_hess(obhip_lpdf *l, double *out);
/* lpdf$optcg(tol, maxepch) (fit.cpp:37-96; iters may be NULL) and lpdf$optnewton()
 * (fit.cpp:98-131) */
int obhip_lpdf_optcg(obhip_lpdf *l, double tol, uint64_t maxepch, uint64_t *iters);
int obhip_lpdf_optnewton(obhip_lpdf *l);
/* lpdf$paralpdf(para) / paralpdf_grad(para): fit.cpp:133-157, lpdfvec fit.cpp:470-496 */
int obhip_lpdf_paralpdf(const obhip_lpdf *l, const double *parap, uint64_t n, double *out);
int obhip_lpdf_paralpdf_grad(const obhip_lpdf *l, const double *parap, uint64_t n, double *out);

/* new(predictor, lpdf) (interfaceR.cpp:725-731, lpdf::pred fit.h:51-55): the predictor of the
 * likelihood -- predr_std (loglik_std.cpp:218-256), pred_gauss (loglik_gauss.cpp:196-227),
 * pred_gda (loglik_gda.cpp:247-281); an lpdfvec stands for its likelihood.  Starts at the
 * training inputs like the reference. */
int obhip_predictor_create(obhip_predictor **out, const obhip_lpdf *l);
int obhip_predictor_destroy(obhip_predictor *p);
int obhip_predictor_setnthreads(obhip_predictor *p, int nthreads);
/* predictor$update(x): x n x d column-major, leading dimension ldx (host) */
int obhip_predictor_update(obhip_predictor *p, const double *x, uint64_t n, uint64_t ldx);
int obhip_predictor_n(const obhip_predictor *p, uint64_t *n);
/* predictor$mean() / $var(): n values */
int obhip_predictor_mean(obhip_predictor *p, double *out);
int obhip_predictor_var(obhip_predictor *p, double *out);

/* ---- synthetic workload of BASELINE.md section 3 (benchmark input) ------ */
/* rows [row0, row0+n) of the counter-based SplitMix64 stream; d_x is n x d
 * column-major, d_y n (raw, not standardised).  kinds: d entries. */
int obhip_synth_xy_dev(uint64_t seed, uint64_t row0, uint64_t n, uint64_t d,
                       const int *kinds, double *d_x, double *d_y);
/* sum and sum of squares of a device vector (for standardising y) */
int obhip_sum_sumsq_dev(const double *d_v, uint64_t n, double *d_out2);
/* v = (v - cent) / sca in place */
int obhip_affine_dev(double *d_v, uint64_t n, double cent, double sca);

/* ---- device memory helpers for non-torch callers (Rcpp glue) ------------ */
int obhip_malloc(void **d_ptr, uint64_t bytes);
int obhip_free(void *d_ptr);
int obhip_memcpy_h2d(void *d_dst, const void *src, uint64_t bytes);
int obhip_memcpy_d2h(void *dst, const void *d_src, uint64_t bytes);
int obhip_memcpy_d2d(void *d_dst, const void *d_src, uint64_t bytes); /* async */

#ifdef __cplusplus
}
#endif
#endif /* OBHIP_H */
