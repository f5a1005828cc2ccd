// C ABI of libobhip: outerbase objects, the Gram/Newton and PCG fit drivers and
// the predictor.  Host logic only; every arithmetic pass over n rows is one of
// the HIP kernels in kernels_*.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "obhip_internal.h"
#include "vec_ops.h"

using namespace obhip;

namespace {

// om.getvar(terms) (modandbase.cpp:350-356) from the cached levels
std::vector<double> term_var(const obhip_model &m, const obhip_terms &t) {
  std::vector<double> v(t.p);
  for (uint64_t k = 0; k < t.p; ++k) {
    double s = 0;
    for (uint64_t l = 0; l < t.d; ++l) s += m.basisvar[m.knotptst[l] + t.lev[k * t.d + l]];
    v[k] = std::exp(s);
  }
  return v;
}

// logpr_gauss::diaghess (logpr_gauss.cpp:122-124)
std::vector<double> prior_prec(const obhip_model &m, const obhip_terms &t, double rho) {
  std::vector<double> v = term_var(m, t);
  const double sca = std::exp(rho);
  for (double &x : v) {
    const double sd = std::sqrt(x) * sca;
    x = 1.0 / (sd * sd);
  }
  return v;
}

int check_compat(const obhip_model *m, const obhip_terms *t) {
  if (!m || !t) return fail(OBHIP_ERR_INVALID, "null model/terms");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  if (t->d != m->d) return fail(OBHIP_ERR_INVALID, "terms and model dimensions differ");
  for (uint64_t l = 0; l < m->d; ++l)
    if (t->maxlev[l] >= (int64_t)m->m_of(l))
      return fail(OBHIP_ERR_INVALID, "terms use a level beyond the model's knots");
  return 0;
}

int basis_setup(obhip_basis *b, const obhip_model *m, const int64_t *levelcap) {
  std::vector<int64_t> cap;
  if (levelcap) cap.assign(levelcap, levelcap + m->d);
  OB_TRY(b->md.build(*m, cap));
  b->grad.reset();  // the gradient basis follows the value basis
  b->bmat_terms = 0;  // and so does the materialised design matrix
  const uint64_t tiles = b->n_pad / kTileRows;
  OB_TRY(b->bm.alloc(tiles * b->md.Mc * kTileRows));
  OB_TRY(b->scale.alloc(b->n_pad));
  return launch_build_basis(*b);
}

}  // namespace

namespace obhip {
std::vector<double> prior_prec_of(const obhip_model &m, const obhip_terms &t, double rho) {
  return prior_prec(m, t, rho);
}
int check_compat_of(const obhip_model *m, const obhip_terms *t) { return check_compat(m, t); }
}  // namespace obhip

namespace {

constexpr size_t kScratch = 2048;  // doubles of scratch of the two-stage reductions: taken from
                                   // the pool per call (keyed by device and stream)

int d2h(void *dst, const void *src, size_t bytes) {
  OB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  return 0;
}

}  // namespace

extern "C" {

// ---- outerbase ------------------------------------------------------------------
int obhip_basis_create(obhip_basis **out, const obhip_model *m, const double *x, uint64_t n,
                       uint64_t ldx, const int64_t *levelcap) {
  if (!out || !m || !x || n == 0 || ldx < n) return fail(OBHIP_ERR_INVALID, "basis_create: bad argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  OB_TRY(require_device());
  obhip_basis *b = new obhip_basis();
  b->model = m;
  b->n = n;
  b->n_pad = (n + kTileRows - 1) / kTileRows * kTileRows;
  b->d = m->d;
  (void)hipGetDevice(&b->device);
  int rc = 0;
  if (ldx == n) {
    rc = b->x.upload(x, n * m->d);
  } else {
    std::vector<double> xc(n * m->d);
    for (uint64_t l = 0; l < m->d; ++l) std::memcpy(&xc[l * n], x + l * ldx, n * sizeof(double));
    rc = b->x.upload(xc.data(), xc.size());
  }
  if (!rc) rc = basis_setup(b, m, levelcap);
  if (rc) {
    delete b;
    return rc;
  }
  *out = b;
  return 0;
}

int obhip_basis_create_dev(obhip_basis **out, const obhip_model *m, const double *d_x, uint64_t n,
                           const int64_t *levelcap) {
  if (!out || !m || !d_x || n == 0) return fail(OBHIP_ERR_INVALID, "basis_create_dev: bad argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  OB_TRY(require_device());
  obhip_basis *b = new obhip_basis();
  b->model = m;
  b->n = n;
  b->n_pad = (n + kTileRows - 1) / kTileRows * kTileRows;
  b->d = m->d;
  (void)hipGetDevice(&b->device);
  int rc = b->x.alloc(n * m->d);
  if (!rc && hipMemcpyAsync(b->x.p, d_x, n * m->d * sizeof(double), hipMemcpyDeviceToDevice,
                            cur_stream()) != hipSuccess)
    rc = fail(OBHIP_ERR_HIP, "copy of x failed");
  if (!rc) rc = basis_setup(b, m, levelcap);
  if (rc) {
    delete b;
    return rc;
  }
  *out = b;
  return 0;
}

int obhip_basis_rebuild(obhip_basis *b) {
  if (!b) return fail(OBHIP_ERR_INVALID, "null basis");
  std::vector<int64_t> cap = b->md.cap;
  if (b->model->d != b->d) return fail(OBHIP_ERR_INVALID, "model dimension changed");
  return basis_setup(b, b->model, cap.data());
}

int obhip_basis_destroy(obhip_basis *b) {
  if (b) (void)hipStreamSynchronize(cur_stream());
  delete b;
  return 0;
}

int obhip_basis_dims(const obhip_basis *b, uint64_t *n, uint64_t *d, uint64_t *ncols_stored) {
  if (!b) return fail(OBHIP_ERR_INVALID, "null basis");
  if (n) *n = b->n;
  if (d) *d = b->d;
  if (ncols_stored) *ncols_stored = b->md.Mc;
  return 0;
}

int obhip_basis_getbase(const obhip_basis *b, uint64_t k, double *out) {
  if (!b || !out) return fail(OBHIP_ERR_INVALID, "getbase: null argument");
  if (k < 1 || k > b->d) return fail(OBHIP_ERR_INVALID, "getbase: dimension out of range (1-based)");
  const uint64_t m = b->model->m_of(k - 1);
  DevBuf<double> tmp;
  OB_TRY(tmp.alloc(b->n * m));
  OB_TRY(launch_getbase(*b, k - 1, tmp.p));
  return d2h(out, tmp.p, b->n * m * sizeof(double));
}

int obhip_basis_getmat(const obhip_basis *b, const obhip_terms *t, double *out) {
  if (!b || !t || !out) return fail(OBHIP_ERR_INVALID, "getmat: null argument");
  OB_TRY(check_compat(b->model, t));
  DevBuf<double> tmp;
  OB_TRY(tmp.alloc(b->n * t->p));
  OB_TRY(launch_getmat(*b, *const_cast<obhip_terms *>(t), tmp.p));
  return d2h(out, tmp.p, b->n * t->p * sizeof(double));
}

static int mm_host(const obhip_basis *b, const obhip_terms *t, const double *a, uint64_t ncol,
                   double *out, bool transposed, bool squared) {
  if (!b || !t || !a || !out || ncol == 0) return fail(OBHIP_ERR_INVALID, "matmul: bad argument");
  OB_TRY(check_compat(b->model, t));
  const uint64_t nin = transposed ? b->n : t->p, nout = transposed ? t->p : b->n;
  DevBuf<double> din, dout;
  OB_TRY(dout.alloc(nout));
  for (uint64_t c = 0; c < ncol; ++c) {
    OB_TRY(din.upload(a + c * nin, nin));
    if (transposed)
      OB_TRY(launch_tmm(*b, *const_cast<obhip_terms *>(t), din.p, dout.p, squared));
    else
      OB_TRY(launch_mm(*b, *const_cast<obhip_terms *>(t), din.p, dout.p, squared));
    OB_TRY(d2h(out + c * nout, dout.p, nout * sizeof(double)));
  }
  return 0;
}

int obhip_basis_mm(const obhip_basis *b, const obhip_terms *t, const double *a, uint64_t ncol,
                   double *out) {
  return mm_host(b, t, a, ncol, out, false, false);
}
int obhip_basis_tmm(const obhip_basis *b, const obhip_terms *t, const double *a, uint64_t ncol,
                    double *out) {
  return mm_host(b, t, a, ncol, out, true, false);
}
int obhip_basis_sqmm(const obhip_basis *b, const obhip_terms *t, const double *a, uint64_t ncol,
                     double *out) {
  return mm_host(b, t, a, ncol, out, false, true);
}
int obhip_basis_sqtmm(const obhip_basis *b, const obhip_terms *t, const double *a, uint64_t ncol,
                      double *out) {
  return mm_host(b, t, a, ncol, out, true, true);
}

int obhip_basis_sqcolsums(const obhip_basis *b, const obhip_terms *t, double *out) {
  if (!b || !t || !out) return fail(OBHIP_ERR_INVALID, "sqcolsums: null argument");
  OB_TRY(check_compat(b->model, t));
  // tmm of the squared store with the all-ones vector (modandbase.cpp:864-866), the ones
  // filled on the device
  DevBuf<double> dones, dout;
  OB_TRY(dones.alloc(b->n));
  OB_TRY(dout.alloc(t->p));
  OB_TRY(launch_fill(dones.p, b->n, 1.0));
  OB_TRY(launch_tmm(*b, *const_cast<obhip_terms *>(t), dones.p, dout.p, true));
  return d2h(out, dout.p, t->p * sizeof(double));
}

int obhip_basis_residvar(const obhip_basis *b, const obhip_terms *t, const obhip_model *m,
                         double *out) {
  if (!b || !t || !m || !out) return fail(OBHIP_ERR_INVALID, "residvar: null argument");
  OB_TRY(check_compat(m, t));
  const std::vector<double> v = term_var(*m, *t);
  OB_TRY(mm_host(b, t, v.data(), 1, out, false, true));
  for (uint64_t i = 0; i < b->n; ++i) out[i] = 1.0 - out[i];  // modandbase.cpp:891
  return 0;
}

int obhip_basis_mm_dev(const obhip_basis *b, const obhip_terms *t, const double *d_a, double *d_out,
                       int squared) {
  if (!b || !t || !d_a || !d_out) return fail(OBHIP_ERR_INVALID, "mm_dev: null argument");
  OB_TRY(check_compat(b->model, t));
  return launch_mm(*b, *const_cast<obhip_terms *>(t), d_a, d_out, squared != 0);
}

int obhip_basis_tmm_dev(const obhip_basis *b, const obhip_terms *t, const double *d_a, double *d_out,
                        int squared) {
  if (!b || !t || !d_a || !d_out) return fail(OBHIP_ERR_INVALID, "tmm_dev: null argument");
  OB_TRY(check_compat(b->model, t));
  return launch_tmm(*b, *const_cast<obhip_terms *>(t), d_a, d_out, squared != 0);
}

// ---- Gram / Newton ----------------------------------------------------------------
int obhip_gram_dev(const obhip_basis *b, const obhip_terms *t, const double *d_y, double *d_G,
                   double *d_g) {
  if (!b || !t || !d_G) return fail(OBHIP_ERR_INVALID, "gram_dev: null argument");
  if (d_y && !d_g) return fail(OBHIP_ERR_INVALID, "gram_dev: d_g is null");
  OB_TRY(check_compat(b->model, t));
  OB_TRY(launch_gram(*b, *const_cast<obhip_terms *>(t), d_G));
  if (d_y) OB_TRY(launch_tmm(*b, *const_cast<obhip_terms *>(t), d_y, d_g, false));
  return 0;
}

int obhip_set_gram_backend(int backend) {
  if (backend != 0 && backend != 3 && backend != 4)
    return fail(OBHIP_ERR_INVALID, "gram backend must be 0 (automatic), 3 (fused) or 4 (staged design matrix)");
  set_gram_backend(backend);
  return 0;
}

int obhip_newton_workspace_bytes(uint64_t p, uint64_t *bytes) {
  if (!bytes) return fail(OBHIP_ERR_INVALID, "null argument");
  // z + info (Cholesky) + prior precision + scaled right-hand side
  *bytes = newton_workspace_bytes(p) + 2 * p * sizeof(double);
  return 0;
}

int obhip_newton_solve_dev(const obhip_model *m, const obhip_terms *t, double *d_G, const double *d_g,
                           double sigma, double rho, double *d_theta, double *d_diagH,
                           void *d_workspace, uint64_t workspace_bytes) {
  if (!m || !t || !d_G || !d_g || !d_theta || !d_workspace)
    return fail(OBHIP_ERR_INVALID, "newton_solve_dev: null argument");
  OB_TRY(check_compat(m, t));
  const uint64_t p = t->p;
  uint64_t need = 0;
  obhip_newton_workspace_bytes(p, &need);
  if (workspace_bytes < need) return fail(OBHIP_ERR_INVALID, "newton_solve_dev: workspace too small");
  double *d_prec = (double *)d_workspace;
  double *d_rhs = d_prec + p;
  void *d_cholws = d_rhs + p;
  const std::vector<double> prec = prior_prec(*m, *t, rho);
  const double e2 = std::exp(-2.0 * sigma);
  hipStream_t st = cur_stream();
  OB_HIP(hipMemcpyAsync(d_prec, prec.data(), p * sizeof(double), hipMemcpyHostToDevice, st));
  OB_HIP(hipStreamSynchronize(st));  // prec is a local
  // H = e^{-2 sigma} G + diag(prec)   (loglik_std.cpp:170-173 + logpr_gauss.cpp:153-158)
  OB_TRY(launch_form_hessian(p, d_G, d_prec, e2, d_diagH));
  // grad at coeff = 0: e^{-2 sigma} B^T y   (loglik_std.cpp:113-116)
  OB_HIP(hipMemcpyAsync(d_rhs, d_g, p * sizeof(double), hipMemcpyDeviceToDevice, st));
  OB_TRY(launch_scale(d_rhs, p, e2));
  return launch_newton_solve(p, d_G, d_rhs, d_theta, d_cholws, newton_workspace_bytes(p));
}

int obhip_fit_newton(const obhip_basis *b, const obhip_terms *t, const obhip_model *m,
                     const double *y, double sigma, double rho, double *theta, double *diagH,
                     double *H_out) {
  if (!b || !t || !m || !y || !theta) return fail(OBHIP_ERR_INVALID, "fit_newton: null argument");
  OB_TRY(check_compat(m, t));
  const uint64_t p = t->p;
  DevBuf<double> dy, dG, dg, dth, ddiag;
  DevBuf<char> ws;
  uint64_t wsb = 0;
  obhip_newton_workspace_bytes(p, &wsb);
  OB_TRY(dy.upload(y, b->n));
  OB_TRY(dG.alloc(p * p));
  OB_TRY(dg.alloc(p));
  OB_TRY(dth.alloc(p));
  OB_TRY(ddiag.alloc(p));
  OB_TRY(ws.alloc(wsb));
  OB_TRY(obhip_gram_dev(b, t, dy.p, dG.p, dg.p));
  if (H_out) {
    // the caller wants H itself (lpdfvec::hess_, fit.cpp:503-512): form it on a copy
    DevBuf<double> dH, dprec;
    OB_TRY(dH.alloc(p * p));
    OB_HIP(hipMemcpyAsync(dH.p, dG.p, p * p * sizeof(double), hipMemcpyDeviceToDevice, cur_stream()));
    const std::vector<double> prec = prior_prec(*m, *t, rho);
    OB_TRY(dprec.upload(prec.data(), p));
    OB_TRY(launch_form_hessian(p, dH.p, dprec.p, std::exp(-2.0 * sigma), nullptr));
    OB_TRY(d2h(H_out, dH.p, p * p * sizeof(double)));
  }
  OB_TRY(obhip_newton_solve_dev(m, t, dG.p, dg.p, sigma, rho, dth.p, ddiag.p, ws.p, wsb));
  OB_TRY(d2h(theta, dth.p, p * sizeof(double)));
  if (diagH) OB_TRY(d2h(diagH, ddiag.p, p * sizeof(double)));
  return 0;
}

// ---- PCG ----------------------------------------------------------------------------
// lpdf::optcg (fit.cpp:37-96) for lpdfvec(logpr_gauss, loglik_gauss | loglik_std) with every
// vector in HBM: theta, the gradient, the preconditioned gradient, the search direction and
// the Hessian products are p-vectors on the device, the step algebra of an iteration (three
// dot products, the break conditions, theta / gradient / direction updates) is ONE
// single-workgroup kernel, and the host reads back five scalars per iteration to learn
// whether the loop goes on.  With several ranks the p-vector of every B^T a pass is summed
// on the library's stream (RCCL) before the kernel that consumes it; nothing but the five
// scalars ever synchronises the host.
}  // extern "C"

namespace {

constexpr int kCgThreads = 1024;
enum { S_VAL = 0, S_VALDIFF, S_NUM, S_DENOM, S_ALPHA, S_DONE, S_NEXT, S_NUM0, S_GP, S_FINITE, S_VALO, S_ITERS, S_COUNT = 16 };

struct CgVecs {
  double *theta, *grad, *rm, *pv, *q, *mdiag;
  const double *prec;
  uint64_t p;
};

template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *red /* K * kCgThreads / 64 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = kCgThreads / 64;
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    if (lane == 0) red[k * nw + wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[k * nw + w];  // fixed order: reproducible
    v[k] = s;
  }
  __syncthreads();
}

// after the two passes of update(): hv = B^T r (summed over ranks), ss = sum (yhat - y)^2:
// grad = hv - theta prec (logpr_gauss.cpp:105), val (loglik_gauss.cpp:121 + logpr_gauss.cpp:101)
__global__ void __launch_bounds__(kCgThreads)
k_cg_eval(CgVecs v, const double *__restrict__ hv, double e2, double ntot_sigma, double logsd,
          double *__restrict__ scal) {
  __shared__ double red[kCgThreads / 64];
  double s[1] = {0.0};
  for (uint64_t k = threadIdx.x; k < v.p; k += kCgThreads) {
    const double c = v.theta[k];
    v.grad[k] = hv[k] - c * v.prec[k];
    s[0] = fma(c * c, v.prec[k], s[0]);
  }
  block_sum<1>(s, red);
  if (threadIdx.x == 0) {
    scal[S_VALO] = scal[S_VAL];
    scal[S_VAL] = -0.5 * e2 * hv[v.p + 1] - ntot_sigma - 0.5 * s[0] - logsd;
  }
}

// m = diaghess (e^{-2 sigma} sqcolsums + prior), rm = grad / m, pv = rm (fit.cpp:47-60)
__global__ void __launch_bounds__(kCgThreads)
k_cg_init(CgVecs v, const double *__restrict__ sq, double e2, double *__restrict__ scal) {
  __shared__ double red[2 * kCgThreads / 64];
  double bad[2] = {0.0, 0.0};  // non-finite entries of m, of grad
  for (uint64_t k = threadIdx.x; k < v.p; k += kCgThreads) {
    const double md = e2 * sq[k] + v.prec[k];
    v.mdiag[k] = md;
    const double r = v.grad[k] / md;
    v.rm[k] = r;
    v.pv[k] = r;
    if (!isfinite(md)) bad[0] += 1.0;
    if (!isfinite(v.grad[k])) bad[1] += 1.0;
  }
  block_sum<2>(bad, red);
  if (threadIdx.x == 0) {
    // fit.cpp:53-56 leaves only when m AND grad are both non-finite; with one of them the
    // reference goes on (and the first step's num / denom stop the loop below)
    scal[S_FINITE] = (bad[0] != 0.0 && bad[1] != 0.0) ? 0.0 : 1.0;
    scal[S_VALDIFF] = 10.0;
    scal[S_NUM0] = -1.0;
    scal[S_DONE] = 0.0;
    scal[S_NEXT] = 0.0;
    scal[S_ITERS] = 0.0;
  }
}

// q = e^{-2 sigma} (B^T B pv) + prec pv (lpdfvec::hessmult, fit.cpp:382-392)
__global__ void __launch_bounds__(256)
k_cg_q(CgVecs v, const double *__restrict__ raw, double e2) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < v.p) v.q[k] = e2 * raw[k] + v.prec[k] * v.pv[k];
}

// One iteration of fit.cpp:71-85.  PART 0: the whole step with gradient and value advanced
// by the recurrence (the objective is exactly quadratic); PART 1: up to theta += alpha pv
// (a full update() follows); PART 2: the rest of the step after that update().
// The whole step (PART 0) for p <= 4 x 1024 with every vector element in registers: ONE round of
// loads where the general form below re-reads what it has just written (its pointers may alias)
// -- four dependent trips to memory and 18 us per iteration at p = 4096, a chain the Hessian
// product waits behind.  Same sums in the same order (element k = thread + 1024 j, j ascending).
__device__ __forceinline__ void cg_iter_regs(const CgVecs &v, double tol, double *__restrict__ scal,
                                             double *red, double *sh) {
  constexpr int E = 4;
  double g[E], rm[E], q[E], pv[E], md[E], th[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const uint64_t k = threadIdx.x + (uint64_t)kCgThreads * j;
    const bool in = k < v.p;
    g[j] = in ? v.grad[k] : 0.0;
    rm[j] = in ? v.rm[k] : 0.0;
    q[j] = in ? v.q[k] : 0.0;
    pv[j] = in ? v.pv[k] : 0.0;
    md[j] = in ? v.mdiag[k] : 1.0;
    th[j] = in ? v.theta[k] : 0.0;
  }
  double d[3] = {0.0, 0.0, 0.0};  // num = grad . rm, denom = q . pv, gp = grad . pv
#pragma unroll
  for (int j = 0; j < E; ++j) {
    d[0] = fma(g[j], rm[j], d[0]);
    d[1] = fma(q[j], pv[j], d[1]);
    d[2] = fma(g[j], pv[j], d[2]);
  }
  block_sum<3>(d, red);
  if (threadIdx.x == 0) {
    const double num = d[0], denom = d[1];
    double num0 = scal[S_NUM0];
    bool done = num < tol && scal[S_VALDIFF] < tol;  // fit.cpp:73
    if (!done) {  // (the guards of the general form below)
      if (num0 < 0.0) num0 = num;
      if (num <= 1e-28 * num0 || !(num > 0.0) || !(denom > 0.0)) done = true;
    }
    scal[S_NUM0] = num0;
    scal[S_NUM] = num;
    scal[S_DENOM] = denom;
    scal[S_GP] = d[2];
    scal[S_DONE] = done ? 1.0 : 0.0;
    scal[S_ALPHA] = done ? 0.0 : num / denom;
    sh[0] = done ? 1.0 : 0.0;
    sh[1] = done ? 0.0 : num / denom;
    sh[2] = num;
  }
  __syncthreads();
  if (sh[0] != 0.0) return;
  const double alpha = sh[1], num = sh[2];
  if (threadIdx.x == 0) {
    scal[S_ITERS] += 1.0;
    // recurrence: val += alpha g.p - alpha^2 p.q / 2
    const double dv = alpha * d[2] - 0.5 * alpha * alpha * d[1];
    scal[S_VALO] = scal[S_VAL];
    scal[S_VAL] += dv;
  }
  double n2[2] = {0.0, 0.0};
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const uint64_t k = threadIdx.x + (uint64_t)kCgThreads * j;
    th[j] = fma(alpha, pv[j], th[j]);
    g[j] = fma(-alpha, q[j], g[j]);  // grad -= alpha q
    const double r = g[j] / md[j];   // rm = grad / m (fit.cpp:80-84)
    n2[0] = fma(-(alpha * q[j]), r, n2[0]);
    n2[1] = fma(g[j], r, n2[1]);
    rm[j] = r;
    if (k < v.p) {
      v.theta[k] = th[j];
      v.grad[k] = g[j];
      v.rm[k] = r;
    }
  }
  block_sum<2>(n2, red);
  const double beta = n2[0] / num;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const uint64_t k = threadIdx.x + (uint64_t)kCgThreads * j;
    if (k < v.p) v.pv[k] = fma(beta, pv[j], rm[j]);
  }
  if (threadIdx.x == 0) {
    const double vd = scal[S_VAL] - scal[S_VALO];
    scal[S_VALDIFF] = vd;
    scal[S_NEXT] = (n2[1] < tol && vd < tol) ? 1.0 : 0.0;
  }
}

template <int PART>
__global__ void __launch_bounds__(kCgThreads)
k_cg_iter(CgVecs v, double tol, double *__restrict__ scal) {
  __shared__ double red[3 * kCgThreads / 64];
  __shared__ double sh[4];
  // the loop has ended (a launch enqueued ahead of the host's look at the break conditions: the
  // batched form of fit_cg_dev_impl): nothing to do
  if (PART != 2 && (scal[S_DONE] != 0.0 || scal[S_NEXT] != 0.0)) return;
  if (PART == 0 && v.p <= 4 * (uint64_t)kCgThreads) {
    cg_iter_regs(v, tol, scal, red, sh);
    return;
  }
  if (PART != 2) {
    double d[3] = {0.0, 0.0, 0.0};  // num = grad . rm, denom = q . pv, gp = grad . pv
    for (uint64_t k = threadIdx.x; k < v.p; k += kCgThreads) {
      d[0] = fma(v.grad[k], v.rm[k], d[0]);
      d[1] = fma(v.q[k], v.pv[k], d[1]);
      d[2] = fma(v.grad[k], v.pv[k], d[2]);
    }
    block_sum<3>(d, red);
    if (threadIdx.x == 0) {
      const double num = d[0], denom = d[1];
      double num0 = scal[S_NUM0];
      bool done = num < tol && scal[S_VALDIFF] < tol;  // fit.cpp:73
      // Rounding floor: once the preconditioned gradient has dropped by 14 digits what is left
      // of it is rounding noise and the next direction a difference of two equal numbers; an
      // exactly vanishing gradient or a direction without curvature has no step length
      // either.  The reference has none of these guards and divides 0 / 0 there.
      if (!done) {
        if (num0 < 0.0) num0 = num;
        if (num <= 1e-28 * num0 || !(num > 0.0) || !(denom > 0.0)) done = true;
      }
      scal[S_NUM0] = num0;
      scal[S_NUM] = num;
      scal[S_DENOM] = denom;
      scal[S_GP] = d[2];
      scal[S_DONE] = done ? 1.0 : 0.0;
      scal[S_ALPHA] = done ? 0.0 : num / denom;
      sh[0] = done ? 1.0 : 0.0;
      sh[1] = done ? 0.0 : num / denom;
    }
    __syncthreads();
    if (sh[0] != 0.0) return;
    if (threadIdx.x == 0) scal[S_ITERS] += 1.0;  // steps made (the host's k when it enqueues ahead)
    const double alpha = sh[1];
    for (uint64_t k = threadIdx.x; k < v.p; k += kCgThreads) v.theta[k] = fma(alpha, v.pv[k], v.theta[k]);
    if (PART == 1) return;
    // recurrence: val += alpha g.p - alpha^2 p.q / 2, grad -= alpha q
    if (threadIdx.x == 0) {
      const double dv = alpha * scal[S_GP] - 0.5 * alpha * alpha * scal[S_DENOM];
      scal[S_VALO] = scal[S_VAL];
      scal[S_VAL] += dv;
    }
    for (uint64_t k = threadIdx.x; k < v.p; k += kCgThreads) v.grad[k] = fma(-alpha, v.q[k], v.grad[k]);
    __syncthreads();
  }
  // rm = grad / m, beta = -(alpha q) . rm / num, pv = rm + beta pv (fit.cpp:80-84); and the
  // tolerance test the NEXT iteration opens with (fit.cpp:72-73 needs grad and rm only), so
  // that the host can skip the Hessian product of a converged iterate
  const double alpha = scal[S_ALPHA], num = scal[S_NUM];
  double n2[2] = {0.0, 0.0};
  for (uint64_t k = threadIdx.x; k < v.p; k += kCgThreads) {
    const double r = v.grad[k] / v.mdiag[k];
    v.rm[k] = r;
    n2[0] = fma(-(alpha * v.q[k]), r, n2[0]);
    n2[1] = fma(v.grad[k], r, n2[1]);
  }
  block_sum<2>(n2, red);
  const double beta = n2[0] / num;
  for (uint64_t k = threadIdx.x; k < v.p; k += kCgThreads) v.pv[k] = fma(beta, v.pv[k], v.rm[k]);
  if (threadIdx.x == 0) {
    const double vd = scal[S_VAL] - scal[S_VALO];
    scal[S_VALDIFF] = vd;
    scal[S_NEXT] = (n2[1] < tol && vd < tol) ? 1.0 : 0.0;
  }
}

}  // namespace

namespace obhip {

// finite_out: 0 when the loop left through the non-finite exit of fit.cpp:53-56 (val = -inf)
int fit_cg_dev_impl(const obhip_basis *b, const obhip_terms *tc, const obhip_model *m,
                    const double *d_y, double sigma, double rho, double tol, uint64_t maxit,
                    double *d_theta, uint64_t *iters_out, double *d_diagH, double *val_out,
                    obhip_comm *comm, int *finite_out, double *d_sqcolsums_out) {
  if (!b || !tc || !m || !d_y || !d_theta) return fail(OBHIP_ERR_INVALID, "fit_cg_dev: null argument");
  OB_TRY(check_compat(m, tc));
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  const uint64_t n = b->n, p = t.p;
  const double e2 = std::exp(-2.0 * sigma);
  const std::vector<double> tv = term_var(*m, t);
  const std::vector<double> prec = prior_prec(*m, t, rho);
  double logsd = 0;  // sum(log(coeffsd * sca)), logpr_gauss.cpp:101
  for (uint64_t k = 0; k < p; ++k) logsd += std::log(std::sqrt(tv[k]) * std::exp(rho));

  DevBuf<double> yhat, r, tmp, dpv, vec, dprec, scal, ddiag;
  OB_TRY(ddiag.alloc(p));
  bool diag_done = false;  // the cold start takes sqcolsums along with B^T r (k_tmm_tl<DUAL>)
  OB_TRY(yhat.alloc(n));
  OB_TRY(r.alloc(n));
  OB_TRY(tmp.alloc(n));
  OB_TRY(dpv.alloc(p + 2));
  OB_TRY(vec.alloc(5 * p));  // grad, rm, pv, q, mdiag
  OB_TRY(dprec.upload(prec.data(), p));
  OB_TRY(scal.alloc(S_COUNT));
  DevBuf<double> redbuf;
  OB_TRY(redbuf.alloc(kScratch));
  double *red = redbuf.p;
  hipStream_t st = cur_stream();
  OB_HIP(hipMemsetAsync(scal.p, 0, S_COUNT * sizeof(double), st));
  CgVecs v{d_theta, vec.p, vec.p + p, vec.p + 2 * p, vec.p + 3 * p, vec.p + 4 * p, dprec.p, p};
  const bool many = comm != nullptr;

  // total number of rows over all ranks
  double ntot = (double)n;
  if (many) {
    OB_HIP(hipMemcpyAsync(dpv.p, &ntot, sizeof(double), hipMemcpyHostToDevice, st));
    OB_HIP(hipStreamSynchronize(st));
    OB_TRY(comm_allreduce(comm, dpv.p, 1));
    OB_TRY(d2h(&ntot, dpv.p, sizeof(double)));
  }
  // lpdfvec::update with compute_val, compute_grad (fit.cpp:323-363) at the theta in HBM
  bool theta_zero = false;  // B 0 = 0 needs no pass over the basis (first update of a cold start)
  {
    const double *th = d_theta;
    double *out = scal.p + S_GP;
    OB_TRY(vsum<1>(p, [=] __device__(uint64_t k, double (&acc)[1]) { acc[0] += fabs(th[k]); }, out, red));
    double s = 1.0;
    OB_TRY(d2h(&s, out, sizeof(double)));
    theta_zero = s == 0.0;
  }
  auto update = [&]() -> int {
    // one fused pass (k_hm_tl): B^T (e^{-2 sigma} (y - B theta)) and sum (B theta - y)^2
    int fused = theta_zero ? kNotFused
                           : launch_hessmult_fused(*b, t, v.theta, d_y, -e2, e2, dpv.p, nullptr, dpv.p + p + 1);
    if (fused != kNotFused) OB_TRY(fused);
    if (fused == kNotFused) {
      if (theta_zero)
        OB_TRY(launch_fill(yhat.p, n, 0.0));
      else
        OB_TRY(launch_mm(*b, t, v.theta, yhat.p, false));      // loglik_gauss.cpp:117
      OB_TRY(launch_resid(yhat.p, d_y, n, e2, r.p, tmp.p));    // :118-124
      // :125, and on the first update the preconditioner's sqcolsums in the same pass
      int dual = diag_done ? kNotFused : launch_tmm_dual(*b, t, r.p, dpv.p, nullptr, ddiag.p);
      if (dual != kNotFused) {
        OB_TRY(dual);
        diag_done = true;
      } else {
        OB_TRY(launch_tmm(*b, t, r.p, dpv.p, false));
      }
      OB_TRY(launch_sum_sumsq(tmp.p, n, dpv.p + p, red));
    }
    theta_zero = false;
    if (many) OB_TRY(comm_allreduce(comm, dpv.p, p + 2));
    hipLaunchKernelGGL(k_cg_eval, dim3(1), dim3(kCgThreads), 0, st, v, dpv.p, e2, ntot * sigma, logsd,
                       scal.p);
    OB_HIP(hipGetLastError());
    return 0;
  };
  // q = lpdfvec::hessmult(pv): loglik_gauss.cpp:137-145 + logpr_gauss.cpp:113-115
  // spec: enqueued BEFORE the host has read the break conditions of the step that precedes it (the
  // GPU then works through the host round trip); the kernel itself returns at once when the step
  // said stop (S_DONE) or that the next iteration would stop at its opening test (S_NEXT)
  const bool can_spec = hessmult_fused_skippable(*b, t);
  auto hessmult = [&](bool spec = false) -> int {
    // (one rank: q comes out of the reduction of the product's row-split partials; with several the
    // sum over ranks lies between the two)
    const HmThen then{e2, v.prec, v.pv, v.q};
    const int fused = launch_hessmult_fused(*b, t, v.pv, nullptr, 1.0, 0.0, dpv.p, nullptr, nullptr,
                                            spec ? scal.p + S_DONE : nullptr, spec ? scal.p + S_NEXT : nullptr,
                                            many ? nullptr : &then);
    if (fused != kNotFused) OB_TRY(fused);
    if (fused == kNotFused) {
      OB_TRY(launch_mm(*b, t, v.pv, yhat.p, false));
      OB_TRY(launch_tmm(*b, t, yhat.p, dpv.p, false));
    }
    if (many) OB_TRY(comm_allreduce(comm, dpv.p, p));
    if (many || fused == kNotFused)
      hipLaunchKernelGGL(k_cg_q, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, st, v, dpv.p, e2);
    OB_HIP(hipGetLastError());
    return 0;
  };

  OB_TRY(update());
  // m = diaghess(): e^{-2 sigma} sqcolsums + prior (loglik_gauss.cpp:154-157)
  if (!diag_done) {
    OB_TRY(launch_fill(tmp.p, n, 1.0));
    OB_TRY(launch_tmm(*b, t, tmp.p, ddiag.p, true));
  }
  if (many) OB_TRY(comm_allreduce(comm, ddiag.p, p));
  // (the likelihood's diaghess / diaghessgradpara are multiples of these sums: lpdfvec::optcg
  // hands them on instead of two more passes over the basis)
  if (d_sqcolsums_out)
    OB_HIP(hipMemcpyAsync(d_sqcolsums_out, ddiag.p, p * sizeof(double), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(kCgThreads), 0, st, v, ddiag.p, e2, scal.p);
  OB_HIP(hipGetLastError());
  double hs[S_COUNT];
  OB_TRY(d2h(hs, scal.p, sizeof hs));
  uint64_t k = 0;
  double val = hs[S_VAL];
  if (hs[S_FINITE] == 0.0) {
    val = -std::numeric_limits<double>::infinity();  // fit.cpp:53-56
  } else {
    OB_TRY(hessmult());
    // The objective is exactly quadratic in coeff, so the gradient and value after the step
    // follow from the Hessian product already in hand: grad -= alpha q, val += alpha g.p -
    // alpha^2 p.q / 2.  The reference re-evaluates both with a full update() (two more
    // passes over the basis) in every iteration (fit.cpp:79); here that happens every
    // `refresh` (16) iterations and, when the caller asks for the value, once at the end, which
    // halves the passes and keeps the same iterates up to rounding.  OBHIP_CG_REFRESH=1
    // restores the reference's schedule.
    const char *re = getenv("OBHIP_CG_REFRESH");
    const uint64_t refresh = re ? (uint64_t)std::max(1, atoi(re)) : 16;
    bool exact = true;  // grad / val come from update(), not from the recurrence
    // Small problems (the reference's own scale: n = 1000, p = 256 takes ~40 us of kernels per
    // iteration and ~90 us for the host to read the five scalars back): `batch` iterations are
    // enqueued before the host looks -- every kernel of an iteration returns at once when an earlier
    // one ended the loop (k_cg_iter's guard, the stop flags of the Hessian product), the device
    // counts the steps made (S_ITERS).  Same iterates, same iteration count, a fraction of the
    // host round trips.  OBHIP_CG_BATCH sets the batch (1 = off); default 8 below 2^22 rows x terms.
    const char *be = getenv("OBHIP_CG_BATCH");
    const uint64_t batch = be ? (uint64_t)std::max(1, atoi(be))
                              : ((double)b->n_pad * (double)t.p_pad <= 4194304.0 ? 8 : 1);
    const bool batched = batch > 1 && can_spec && !many;
    // one iteration with the host in the loop; returns kStop when the loop ends (k then counts the
    // iterations made), 0 to go on, an error code otherwise
    constexpr int kStop = -100;
    auto iteration = [&]() -> int {  // fit.cpp:71-85
      const bool full = (k + 1) % refresh == 0;
      if (full)
        hipLaunchKernelGGL(k_cg_iter<1>, dim3(1), dim3(kCgThreads), 0, st, v, tol, scal.p);
      else
        hipLaunchKernelGGL(k_cg_iter<0>, dim3(1), dim3(kCgThreads), 0, st, v, tol, scal.p);
      OB_HIP(hipGetLastError());
      if (full) {  // the break conditions come first: a converged iterate takes no update()
        OB_TRY(d2h(hs, scal.p, (S_NEXT + 1) * sizeof(double)));
        if (hs[S_DONE] != 0.0) return kStop;
        OB_TRY(update());
        hipLaunchKernelGGL(k_cg_iter<2>, dim3(1), dim3(kCgThreads), 0, st, v, tol, scal.p);
        OB_HIP(hipGetLastError());
      }
      const bool spec = can_spec && !full;
      if (spec) OB_TRY(hessmult(true));
      OB_TRY(d2h(hs, scal.p, (S_NEXT + 1) * sizeof(double)));  // the host sync of an iteration
      if (hs[S_DONE] != 0.0) return kStop;
      exact = full;
      ++k;
      if (hs[S_NEXT] != 0.0) return kStop;  // the next iteration would stop at once: its Hessian product is not needed
      if (!spec) OB_TRY(hessmult());
      return 0;
    };
    for (k = 0; k < maxit;) {
      if (batched && (k + 1) % refresh != 0) {
        // up to `batch` plain iterations, never across a refresh iteration (that one keeps the
        // host in the loop)
        uint64_t nb = 0;
        while (nb < batch && k + nb < maxit && (k + nb + 1) % refresh != 0) {
          hipLaunchKernelGGL(k_cg_iter<0>, dim3(1), dim3(kCgThreads), 0, st, v, tol, scal.p);
          OB_HIP(hipGetLastError());
          OB_TRY(hessmult(true));
          ++nb;
        }
        OB_TRY(d2h(hs, scal.p, (S_ITERS + 1) * sizeof(double)));  // ONE host sync per batch
        exact = false;
        if (hs[S_DONE] != 0.0 || hs[S_NEXT] != 0.0) {
          k = (uint64_t)hs[S_ITERS];
          break;
        }
        k += nb;
        continue;
      }
      const int rc = iteration();
      if (rc == kStop) break;
      if (rc != 0) return rc;
    }
    // the value reported is a true evaluation (callers that re-evaluate anyway pass
    // val_out = NULL and save the two passes)
    if (!exact && val_out) OB_TRY(update());
    OB_TRY(d2h(hs, scal.p, sizeof(double)));
    val = hs[S_VAL];
  }
  if (d_diagH)
    OB_HIP(hipMemcpyAsync(d_diagH, v.mdiag, p * sizeof(double), hipMemcpyDeviceToDevice, st));
  OB_HIP(hipStreamSynchronize(st));
  if (iters_out) *iters_out = k;
  if (val_out) *val_out = val;
  if (finite_out) *finite_out = hs[S_FINITE] != 0.0 ? 1 : 0;
  return 0;
}

}  // namespace obhip

extern "C" {

int obhip_fit_cg_dev(const obhip_basis *b, const obhip_terms *tc, const obhip_model *m,
                     const double *d_y, double sigma, double rho, double tol, uint64_t maxit,
                     double *d_theta, uint64_t *iters_out, double *d_diagH, double *val_out,
                     obhip_comm *comm) {
  return fit_cg_dev_impl(b, tc, m, d_y, sigma, rho, tol, maxit, d_theta, iters_out, d_diagH, val_out,
                         comm, nullptr);
}

int obhip_fit_cg(const obhip_basis *b, const obhip_terms *t, const obhip_model *m, const double *y,
                 double sigma, double rho, double tol, uint64_t maxit, double *theta,
                 uint64_t *iters_out, double *diagH, double *val_out) {
  if (!b || !t || !m || !y || !theta) return fail(OBHIP_ERR_INVALID, "fit_cg: null argument");
  DevBuf<double> dy, dth, ddiag;
  OB_TRY(dy.upload(y, b->n));
  OB_TRY(dth.upload(theta, t->p));
  OB_TRY(ddiag.alloc(t->p));
  OB_TRY(obhip_fit_cg_dev(b, t, m, dy.p, sigma, rho, tol, maxit, dth.p, iters_out, ddiag.p, val_out,
                          nullptr));
  OB_TRY(d2h(theta, dth.p, t->p * sizeof(double)));
  if (diagH) OB_TRY(d2h(diagH, ddiag.p, t->p * sizeof(double)));
  return 0;
}

// ---- predictor ------------------------------------------------------------------------
int obhip_predict_dev(const obhip_model *m, const obhip_terms *t, const double *d_theta,
                      const double *d_x, uint64_t n, double *d_mean, const double *d_coeffvar,
                      double sigma, double *d_var) {
  if (!m || !t || !d_theta || !d_x || !d_mean) return fail(OBHIP_ERR_INVALID, "predict_dev: null argument");
  OB_TRY(check_compat(m, t));
  OB_TRY(require_device());
  return launch_predict(*m, *const_cast<obhip_terms *>(t), d_theta, d_x, n, d_mean, d_coeffvar,
                        std::exp(2.0 * sigma), d_var);
}

int obhip_predict(const obhip_model *m, const obhip_terms *t, const double *theta, const double *x,
                  uint64_t n, uint64_t ldx, double *mean, const double *coeffvar, double sigma,
                  double *var) {
  if (!m || !t || !theta || !x || !mean || n == 0 || ldx < n)
    return fail(OBHIP_ERR_INVALID, "predict: bad argument");
  OB_TRY(check_compat(m, t));
  OB_TRY(require_device());
  DevBuf<double> dx, dth, dmean, dcv, dvar;
  if (ldx == n) {
    OB_TRY(dx.upload(x, n * m->d));
  } else {
    std::vector<double> xc(n * m->d);
    for (uint64_t l = 0; l < m->d; ++l) std::memcpy(&xc[l * n], x + l * ldx, n * sizeof(double));
    OB_TRY(dx.upload(xc.data(), xc.size()));
  }
  OB_TRY(dth.upload(theta, t->p));
  OB_TRY(dmean.alloc(n));
  const bool do_var = coeffvar && var;
  if (do_var) {
    OB_TRY(dcv.upload(coeffvar, t->p));
    OB_TRY(dvar.alloc(n));
  }
  OB_TRY(obhip_predict_dev(m, t, dth.p, dx.p, n, dmean.p, do_var ? dcv.p : nullptr, sigma,
                           do_var ? dvar.p : nullptr));
  OB_TRY(d2h(mean, dmean.p, n * sizeof(double)));
  if (do_var) OB_TRY(d2h(var, dvar.p, n * sizeof(double)));
  return 0;
}

// ---- synthetic workload ------------------------------------------------------------------
int obhip_synth_xy_dev(uint64_t seed, uint64_t row0, uint64_t n, uint64_t d, const int *kinds,
                       double *d_x, double *d_y) {
  if (!kinds || !d_x || !d_y || d == 0) return fail(OBHIP_ERR_INVALID, "synth: bad argument");
  OB_TRY(require_device());
  DevBuf<int> dk;
  OB_TRY(dk.upload(kinds, d));
  OB_TRY(launch_synth(seed, row0, n, d, dk.p, d_x, d_y));
  OB_HIP(hipStreamSynchronize(cur_stream()));  // dk is a local
  return 0;
}

int obhip_sum_sumsq_dev(const double *d_v, uint64_t n, double *d_out2) {
  if (!d_v || !d_out2) return fail(OBHIP_ERR_INVALID, "null argument");
  DevBuf<double> red;  // back to the pool on return: handed out again to this stream only
  OB_TRY(red.alloc(kScratch));
  return launch_sum_sumsq(d_v, n, d_out2, red.p);
}

int obhip_affine_dev(double *d_v, uint64_t n, double cent, double sca) {
  if (!d_v) return fail(OBHIP_ERR_INVALID, "null argument");
  return launch_affine(d_v, n, cent, sca);
}

}  // extern "C"
