// One Newton fit of lpdfvec(loglik_std, logpr_gauss) on row-sharded data, start to finish on
// the device (lpdf::optnewton, src/fit.cpp:98-131, with obfit's standardisation of y,
// R/fitting.R:55-57, over the rows of ALL ranks).
//
// What crosses ranks:
//   (sum y, n)            16 bytes   -> the mean of y over all rows
//   sum (y - mean)^2       8 bytes   -> its standard deviation, two-pass like R's sd()
//   [packed upper triangle of G_r = B_r^T B_r | g_r = B_r^T y_std,r]   p (p + 1) / 2 + p doubles
// and what touches the p x p matrix on either side of the exchange is one pass each: the
// reduction of the Gram kernel's row-split partials writes the packed triangle straight into
// the exchange buffer, and the unpack forms H = e^{-2 sigma} G + diag(prior) while it restores
// the full symmetric storage the Cholesky wants.  With one rank (comm = NULL) the reduction
// writes H itself and nothing is exchanged.  y is standardised BEFORE B^T y is taken, so no
// B^T 1 pass and no cancellation in (B^T y - mean B^T 1).
#include <cmath>

#include "obhip_internal.h"
#include "vec_ops.h"

using namespace obhip;

namespace obhip {
int launch_unpack_form(uint64_t p, const double *d_tri, double *d_H, double e2, const double *d_prec,
                       double *d_diagH);
std::vector<double> prior_prec_of(const obhip_model &m, const obhip_terms &t, double rho);
int check_compat_of(const obhip_model *m, const obhip_terms *t);
}  // namespace obhip

namespace {

// st = [sum y, n, sum (y - cent)^2] summed over ranks -> meansd = [cent, sca, n]
__global__ void k_meansd(const double *__restrict__ st, double *__restrict__ meansd) {
  if (threadIdx.x == 0) {
    const double n = st[1], cent = st[0] / n;
    meansd[0] = cent;
    meansd[1] = sqrt(st[2] / (n - 1.0));  // n - 1 denominator (R's sd); n = 1 gives NaN as R's does
    meansd[2] = n;
  }
}

__global__ void k_set1(double *dst, double v) {
  if (threadIdx.x == 0) *dst = v;
}

constexpr size_t kScratch = 2048;

}  // namespace

extern "C" {

int obhip_standardise_dev(obhip_comm *comm, const double *d_y_raw, uint64_t n, double *d_y,
                          double *d_meansd) {
  // A rank of a sharded job may hold no rows (fewer rows than ranks, a ragged last shard): it
  // must still take part in the two sums or its peers wait for ever, so with a communicator
  // n = 0 is legal and contributes (0, 0) and 0.  Fewer than two rows over ALL ranks give
  // sd = NaN, as R's sd() does (the count is known on the device only: rejecting it here would
  // put a host synchronisation into every fit).
  if (!d_meansd || (n != 0 && (!d_y_raw || !d_y))) return fail(OBHIP_ERR_INVALID, "standardise_dev: null argument");
  OB_TRY(require_device());
  if (!comm && n < 2) return fail(OBHIP_ERR_INVALID, "standardise_dev: the standard deviation needs two rows");
  DevBuf<double> st, red;
  OB_TRY(st.alloc(4));
  OB_TRY(red.alloc(kScratch));
  hipStream_t s = cur_stream();
  double *stp = st.p;
  const double *y = d_y_raw;
  OB_TRY(vsum<1>(n, [=] __device__(uint64_t i, double (&acc)[1]) { acc[0] += y[i]; }, stp, red.p));
  hipLaunchKernelGGL(k_set1, dim3(1), dim3(64), 0, s, stp + 1, (double)n);
  if (comm) OB_TRY(comm_allreduce(comm, stp, 2));
  OB_TRY(vsum<1>(n, [=] __device__(uint64_t i, double (&acc)[1]) {
    const double c = y[i] - stp[0] / stp[1];
    acc[0] = fma(c, c, acc[0]);
  }, stp + 2, red.p));
  if (comm) OB_TRY(comm_allreduce(comm, stp + 2, 1));
  hipLaunchKernelGGL(k_meansd, dim3(1), dim3(64), 0, s, stp, d_meansd);
  const double *ms = d_meansd;
  double *yo = d_y;
  OB_TRY(vmap(n, [=] __device__(uint64_t i) { yo[i] = (y[i] - ms[0]) / ms[1]; }));
  OB_HIP(hipGetLastError());
  return 0;
}

int obhip_destandardise_dev(double *d_v, uint64_t n, const double *d_meansd) {
  if (!d_v || !d_meansd) return fail(OBHIP_ERR_INVALID, "destandardise_dev: null argument");
  const double *ms = d_meansd;
  return vmap(n, [=] __device__(uint64_t i) { d_v[i] = fma(ms[1], d_v[i], ms[0]); });
}

int obhip_fit_newton_count(uint64_t p, int nranks, uint64_t *count) {
  if (!count || nranks < 1 || p == 0) return fail(OBHIP_ERR_INVALID, "fit_newton_count: bad argument");
  const uint64_t raw = p * (p + 1) / 2 + p;
  const uint64_t q = 2 * (uint64_t)nranks;  // equal 16-byte blocks for reduce-scatter
  *count = (raw + q - 1) / q * q;
  return 0;
}

int obhip_fit_newton_sharded_dev(obhip_comm *comm, const obhip_basis *b, const obhip_terms *tc,
                                 const obhip_model *m, const double *d_y, double sigma, double rho,
                                 double *d_H, double *d_g, double *d_theta, double *d_diagH,
                                 double *d_exbuf, uint64_t exbuf_count, void *d_workspace,
                                 uint64_t workspace_bytes) {
  if (!b || !tc || !m || !d_y || !d_H || !d_g || !d_theta || !d_workspace)
    return fail(OBHIP_ERR_INVALID, "fit_newton_sharded_dev: null argument");
  OB_TRY(require_device());
  // (knots set, same dimensions, no level beyond the model's knots: term_var below indexes the
  // model's tables with the terms' levels)
  OB_TRY(check_compat_of(m, tc));
  obhip_terms &t = *const_cast<obhip_terms *>(tc);
  if (b->model != m) return fail(OBHIP_ERR_INVALID, "fit_newton_sharded_dev: model / terms / basis do not belong together");
  const uint64_t p = t.p;
  uint64_t need = 0;
  obhip_newton_workspace_bytes(p, &need);
  if (workspace_bytes < need) return fail(OBHIP_ERR_INVALID, "fit_newton_sharded_dev: workspace too small");
  const uint64_t tri = p * (p + 1) / 2;
  if (comm) {
    uint64_t cnt = 0;
    OB_TRY(obhip_fit_newton_count(p, comm_nranks(comm), &cnt));
    if (!d_exbuf || exbuf_count < cnt) return fail(OBHIP_ERR_INVALID, "fit_newton_sharded_dev: exchange buffer too small");
    exbuf_count = cnt;
  }
  double *d_rhs = (double *)d_workspace + p;
  void *d_cholws = d_rhs + p;
  const double e2 = std::exp(-2.0 * sigma);
  hipStream_t st = cur_stream();
  // prior precisions: uploaded when the model state or rho changed, otherwise already in HBM
  if (!t.prec_dev.p || t.prec_model != m || t.prec_version != m->version || t.prec_rho != rho) {
    const std::vector<double> prec = prior_prec_of(*m, t, rho);
    OB_TRY(t.prec_dev.upload(prec.data(), p));  // (synchronises: prec is a local)
    t.prec_model = m;
    t.prec_version = m->version;
    t.prec_rho = rho;
  }
  const double *d_prec = t.prec_dev.p;
  GramSink sink;
  if (comm) {
    sink.out = d_exbuf;
    sink.packed = true;
  } else {
    sink.out = d_H;
    sink.form = true;
    sink.e2 = e2;
    sink.prec = d_prec;
    sink.diagH = d_diagH;
  }
  // B^T y: taken along by the staging pass of the design matrix when there is one (its
  // products are the entries of B), by its own pass over the basis otherwise
  double *g_dst = comm ? d_exbuf + tri : d_g;
  GramFuse fuse;
  fuse.y = d_y;
  fuse.g = g_dst;
  OB_TRY(launch_gram_to(*b, t, sink, &fuse));
  if (!fuse.done) OB_TRY(launch_tmm(*b, t, d_y, g_dst, false));
  if (comm) {
    {
      ProfScope ps("exchange");
      OB_TRY(comm_allreduce(comm, d_exbuf, exbuf_count));
    }
    OB_TRY(launch_unpack_form(p, d_exbuf, d_H, e2, d_prec, d_diagH));
    OB_HIP(hipMemcpyAsync(d_g, d_exbuf + tri, p * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  // grad at coeff = 0: e^{-2 sigma} B^T y   (loglik_std.cpp:113-116)
  const double *g = d_g;
  OB_TRY(vmap(p, [=] __device__(uint64_t k) { d_rhs[k] = e2 * g[k]; }));
  return launch_newton_solve(p, d_H, d_rhs, d_theta, d_cholws, newton_workspace_bytes(p));
}

}  // extern "C"
