// Internal declarations shared by the host logic and the HIP kernels of
// libobhip.  Nothing here is part of the ABI (include/obhip.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/obhip.h"

namespace obhip {

// ---- error plumbing ---------------------------------------------------------
int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);
#define OB_HIP(expr)                                                \
  do {                                                              \
    hipError_t _e = (expr);                                         \
    if (_e != hipSuccess) return obhip::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)
#define OB_TRY(expr)        \
  do {                      \
    int _rc = (expr);       \
    if (_rc != 0) return _rc; \
  } while (0)

int require_device();
hipStream_t cur_stream();
void set_cur_stream(hipStream_t s);  // (internal: a second stream inside one entry point; restored before it returns)
// model state stamps are drawn from one process-wide counter: device tables cached under
// (model address, stamp) can then never be taken for those of another model that came to live at
// the same address
uint64_t next_model_version();
int ensure_dyn_lds(const void *kernel, size_t bytes);  // dynamic LDS above 64 KB, granted once per kernel
int device_cus(int device);                            // compute units (cached per device)

// ---- profiling --------------------------------------------------------------
// every blocking wait of the library for the device goes through these (the macros below): counted,
// so that a caller can ask how many host round trips a call sequence made
// (obhip_profile_get("host_syncs"); bench.py reports it for the small-n obfit of configs[0])
extern std::atomic<uint64_t> g_host_syncs;
inline hipError_t counted_stream_sync(hipStream_t s) {
  g_host_syncs.fetch_add(1, std::memory_order_relaxed);
  return hipStreamSynchronize(s);
}
inline hipError_t counted_device_sync() {
  g_host_syncs.fetch_add(1, std::memory_order_relaxed);
  return hipDeviceSynchronize();
}
#define hipStreamSynchronize(s) obhip::counted_stream_sync(s)
#define hipDeviceSynchronize() obhip::counted_device_sync()

struct ProfScope {
  const char *name;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool active = false;
  explicit ProfScope(const char *n);
  ~ProfScope();
};

// host wall time of a scope, summed per name and printed at exit when OBHIP_HOST_TIMING is set
// (tuning aid: where an entry point spends its time between the kernels)
struct HostTimer {
  const char *name;
  double t0 = 0;
  static bool on() {
    static const bool v = getenv("OBHIP_HOST_TIMING") != nullptr;
    return v;
  }
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  }
  struct Table {
    std::map<std::string, std::pair<uint64_t, double>> t;
    ~Table() {
      for (auto &e : t)
        fprintf(stderr, "[host] %-36s %8llu x %10.3f ms = %10.1f ms\n", e.first.c_str(),
                (unsigned long long)e.second.first, e.second.second / e.second.first, e.second.second);
    }
  };
  static Table &table() {
    static Table tb;
    return tb;
  }
  explicit HostTimer(const char *n) : name(n) {
    if (on()) t0 = now();
  }
  ~HostTimer() {
    if (!on()) return;
    auto &e = table().t[name];
    e.first += 1;
    e.second += now() - t0;
  }
};

// ---- device memory pool -----------------------------------------------------------
// Every C-ABI call allocates its device temporaries (vectors of n or p doubles, an n x nhyp
// block, ...) and frees them on return; hipMalloc / hipFree cost far more than the kernels of
// a small call and hipFree synchronises the device.  Freed blocks are therefore kept, keyed
// by exact size, device and the stream they were last used on -- a block is handed out again
// only to work queued on that same stream, so stream order protects it -- up to
// OBHIP_POOL_MB (default: an eighth of the device's memory, at least 8192) of cached memory; larger blocks and the overflow go back to
// the driver, and a failed hipMalloc empties the pool and retries.
int pool_alloc(void **p, size_t bytes);
// stream: the one the block was allocated for (the tag under which it may be handed out again)
void pool_free(void *p, size_t bytes, hipStream_t stream);
void pool_trim();

// ---- device buffer ----------------------------------------------------------
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipStream_t s = nullptr;  // stream current at allocation
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) {
      // used on another stream since (obhip_set_stream between allocation and release): let
      // that work finish, the block goes back under the stream it was allocated for
      if (cur_stream() != s) (void)hipStreamSynchronize(cur_stream());
      pool_free(p, n * sizeof(T), s);
    }
    p = nullptr;
    n = 0;
  }
  int alloc(size_t count) {
    if (count == n && p) return 0;
    release();
    if (count == 0) return 0;
    int rc = pool_alloc((void **)&p, count * sizeof(T));
    if (rc) return rc;
    n = count;
    s = cur_stream();
    return 0;
  }
  int upload(const T *src, size_t count) {
    int rc = alloc(count);
    if (rc) return rc;
    if (count == 0) return 0;
    hipError_t e = hipMemcpyAsync(p, src, count * sizeof(T),
                                  hipMemcpyHostToDevice, cur_stream());
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
    e = hipStreamSynchronize(cur_stream());
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    return 0;
  }
};

// ---- covariance functions (host) --------------------------------------------
constexpr int kNumCov = 3;
struct CovInfo {
  int numhyp;
  double hyp0[2], hyplb[2], hypub[2], hypvar[2];
  double lowbnd, uppbnd;
};
const CovInfo &cov_info(int kind);
void cov_host(int kind, const double *hyp, const double *x1, uint64_t n1,
              const double *x2, uint64_t n2, double *out /* n1 x n2 col-major */);
void cov_gradhyp_host(int kind, const double *hyp, const double *x1, uint64_t n1, const double *x2,
                      uint64_t n2, double *out /* n1 x n2 x numhyp, col-major slices */);
double cov_hyplpdf_host(int kind, const double *hyp);

// symmetric eigen-decomposition (cyclic Jacobi), eigenvalues ascending like
// LAPACK dsyev / arma::eig_sym; a is n x n col-major (destroyed), v n x n.
void jacobi_eigh(int n, std::vector<double> &a, std::vector<double> &w,
                 std::vector<double> &v);

}  // namespace obhip

// ---- outermod -----------------------------------------------------------------
struct obhip_model {
  uint64_t d = 0;
  std::vector<int> kinds;
  std::vector<uint64_t> hypst;     // d+1
  std::vector<double> hyp;
  bool knots_set = false;
  std::vector<uint64_t> knotptst;  // d+1
  std::vector<double> knotpt;      // M
  uint64_t mmax = 0;
  std::vector<double> rotmat;      // mmax x M col-major
  std::vector<double> basisvar;    // M
  std::vector<int64_t> maxlevel;   // d
  uint64_t version = 0;            // bumped whenever build() runs
  // hyper-parameter gradients (modandbase.cpp:183-197,257-274): one block of m_l columns
  // per hyper-parameter; gest[h] = first column of hyper-parameter h, hypmatch[h] = its dim
  std::vector<uint64_t> hypmatch, gest;     // nhyp, nhyp + 1
  std::vector<double> rotmat_gradhyp;       // mmax x gest[nhyp] col-major
  std::vector<double> logbasisvar_gradhyp;  // gest[nhyp]
  uint64_t nhyp() const { return hypmatch.size(); }

  uint64_t M() const { return knotpt.size(); }
  uint64_t m_of(uint64_t l) const { return knotptst[l + 1] - knotptst[l]; }
  int build();
};

// ---- device view of a model restricted to per-dimension level caps ------------
namespace obhip {

struct DimDesc {
  int kind;
  int m;       // knots in this dim
  int koff;    // offset into the knot arrays
  int ncol;    // levels evaluated = cap+1 (level 0 included)
  int ncolp;   // ncol rounded up to a multiple of 8
  int rotoff;  // offset into rot (doubles)
  int ccol0;   // compact column of level 1 (level t -> ccol0 + t - 1)
  int tab;     // offset (doubles) of the dimension's interval tables in ModelDev::tab, -1: none
  int gwin;    // interval search of the tables: 0 = seven bisection steps; E > 0: the guess
               // J0 = floor((u - g0) ginv) + 1 is within E - 1 of the interval for every u (checked on
               // the host), so 2 E independent reads around it settle it in one round trip
  double p0, p1, p2;  // kernel constants (see kernels_basis.hip)
  double g0, ginv;    // the guess of gwin: first sorted knot (in u), (m - 1) / (last - first)
};

struct ModelDev {
  uint64_t model_version = ~0ull;
  const void *built_for = nullptr;  // the model the tables were built from
  std::vector<int64_t> cap;     // d
  std::vector<DimDesc> dims_h;
  uint64_t Mc = 0;              // compact columns incl. the ones column 0
  DevBuf<DimDesc> dims;
  DevBuf<double> ka, kb, kc;    // per-knot constants (M each)
  DevBuf<double> rot;           // per dim [m][ncolp]
  DevBuf<double> tab;           // per mat25 / mat25pow dim [m sorted u (even length)][m + 1][ncol][6]
  int build(const obhip_model &m, const std::vector<int64_t> &cap);
};

}  // namespace obhip

// ---- terms --------------------------------------------------------------------
// star tables of a term set (csrc/share.cpp): p_pad / 4 stars of four terms that share all
// factors but one (or, plain stars, nothing), in star-waves of 64
namespace obhip {
struct ShareTables {
  bool ok = false;
  uint64_t nstars = 0;       // 64 x (nsw_family + nsw_plain), the empty stars that fill the waves included
  uint64_t nsw_family = 0;   // star-waves of family stars (shape (P, 1)) -- they come first
  uint64_t nsw_plain = 0;    // star-waves of plain stars (shape (0, S)): the left-over terms four at a time
  uint64_t nleft = 0;        // left-over terms (no family with four free members)
  uint64_t reads = 0;        // column reads per row of the family star-waves: sum of P + 4
  uint64_t reads_left = 0;   // of the plain star-waves: sum of 4 S
  uint64_t reads_plain = 0;  // of the nnz-sorted scheme without sharing, 4 terms per lane (rounds 1-4)
  uint64_t lds_cycles = 0;   // LDS cycles of all star-waves' reads with their bank conflicts (2 per read at best)
  uint64_t lds_cycles0 = 0;  // the same before the half-wave / term-order search
  std::vector<uint16_t> cols;   // nstars x 4 W used-column indices, laid out for the star-wave's shape
  std::vector<uint32_t> term;   // nstars x 4 term indices (0xffffffff: none -- an empty star's)
  std::vector<uint32_t> shape;  // per star-wave: P | S << 8
  std::vector<uint32_t> left_term;  // nleft term indices
  std::vector<uint16_t> left_cols;  // nleft x W used-column indices, right-aligned (0 = ones)
  std::vector<uint16_t> relabel;    // used-column index the caller passed -> index the tables are written in
};
int build_share_tables(const uint16_t *hc, uint64_t p_pad, uint64_t W, ShareTables &out, bool renumber = false,
                       uint64_t ncol = 0);
bool share_wanted();  // OBHIP_SHARE=0: the kernels take the plain tables (A/B measurements)
}  // namespace obhip

struct obhip_terms {
  uint64_t uid = 0;                   // unique per object (caches keyed by terms use it)
  uint64_t p = 0, d = 0;
  std::vector<uint32_t> lev;          // p x d row-major levels
  std::vector<int64_t> maxlev;        // d
  uint64_t nnz_total = 0, max_nnz = 0;
  // device tables, rebuilt when the compact layout (caps) changes
  std::vector<int64_t> cached_cap;
  uint64_t W = 0;                     // padded column-list width
  uint64_t Mu = 0;                    // used compact columns
  obhip::DevBuf<uint16_t> cols;       // p_pad x W indices into the USED list
  obhip::DevBuf<uint32_t> ucol;       // Mu compact column ids (used list)
  obhip::DevBuf<uint32_t> sperm;      // p_pad: terms ordered by falling number of factors (stable)
  obhip::DevBuf<int32_t> cpos;        // compact column -> used index or -1 (Mc)
  uint64_t p_pad = 0;
  // star tables (shared sub-products, csrc/share.cpp); sh.ok false: the kernels take sperm / cols
  obhip::ShareTables sh;              // (host copies dropped after the upload; counts kept)
  bool no_share = false;              // views of another term set (gradient passes): no star tables
  obhip::DevBuf<uint16_t> sh_cols;    // nstars x 4 W
  obhip::DevBuf<uint32_t> sh_term;    // nstars x 4
  obhip::DevBuf<uint32_t> sh_shape;   // nstars / 64
  obhip::DevBuf<uint32_t> sh_left_term;  // nleft (at least one element)
  obhip::DevBuf<uint16_t> sh_left_cols;  // nleft x W
  // per hyper-parameter views for the gradient products (kernels_grad.hip)
  std::vector<std::unique_ptr<obhip_terms>> ge_views;
  // the same restricted to the terms that HAVE the hyper-parameter's dimension, with
  // their indices (transposed gradient products: the other terms come from one dense pass)
  std::vector<std::unique_ptr<obhip_terms>> ge_sviews;
  std::vector<std::vector<uint32_t>> ge_sidx;
  std::vector<uint64_t> ge_views_sig;  // hypmatch of the model the views were built for
  std::vector<uint64_t> ge_full_sig;   // the same for ge_views
  // restricted likewise, the dimension's factor replaced by the delta column (products B a)
  std::vector<std::unique_ptr<obhip_terms>> ge_dviews;
  // all ge_sviews concatenated (one B^T a pass for every hyper-parameter), offsets per h
  struct GeGroup {
    std::unique_ptr<obhip_terms> v;  // the ge_sviews of `hyps`, concatenated
    std::vector<uint64_t> hyps, off; // off[j]: first term of hyps[j] in v (off.size() = hyps.size() + 1)
  };
  std::vector<GeGroup> ge_sgroups;
  std::vector<GeGroup> ge_dgroups;  // the ge_dviews likewise (w^T d(B a)/dhyp as B_delta^T w)
  // the fused gradient passes (k_tmm_d3, kernels_grad.hip): per launch the terms that have one of a
  // few dimensions, each with that dimension's own factor moved to the END of its column list and
  // followed by the delta columns (level of the term) of up to two of the dimension's
  // hyper-parameters -- tables written by hand into v (no levels), keyed by the level caps
  struct GeD3 {
    std::unique_ptr<obhip_terms> v;
    int nh = 0;                       // delta columns per view-term (1 or 2)
    std::vector<uint64_t> hyp0, off;  // member j: hyper-parameters hyp0[j] .. hyp0[j] + nh - 1,
                                      // view-terms [off[j], off[j + 1]) = the terms ge_sidx[hyp0[j]]
  };
  std::vector<GeD3> ge_d3;
  std::vector<int64_t> ge_d3_cap;     // key of the column layout ge_d3 was built for (build_d3_groups)
  bool ge_d3_ok = false;              // false: some view does not fit the kernel -> older passes
  // device view of the model capped at maxlev, for the fused predictor
  obhip::ModelDev pred_md;
  const obhip_model *pred_model = nullptr;
  // prior precisions 1 / (sd e^rho)^2 of these terms on the device, for the model state and rho
  // they were last asked for (the device-side Newton fit: no upload, no host sync per fit)
  obhip::DevBuf<double> prec_dev;
  const obhip_model *prec_model = nullptr;
  uint64_t prec_version = 0;
  double prec_rho = 0.0;
  int prepare(const std::vector<int64_t> &cap, const std::vector<obhip::DimDesc> &dims);
};

// ---- hyper-parameter gradient tables (device) -------------------------------------
namespace obhip {
struct GradHyp {
  int dim;      // dimension of this hyper-parameter
  int which;    // its index within the dimension (0 or 1)
  int rotgoff;  // offset into rotg (doubles), block [m][ncolp] like ModelDev::rot
  int gecol;    // first gradient column of this hyper-parameter in the combined tile
  int dcol;     // first delta column: delta[t] = ge[t] - basemat[level t] ge[0], t >= 1
};
}  // namespace obhip

struct obhip_basis;
// gradient basis of an outerbase (outerbase::build with dograd, modandbase.cpp:547-626):
// a second tile-blocked array holding the basemat columns AND, behind them, for every
// hyper-parameter h the columns basemat_gradhyp[:, gest[h] + t], t = 0..cap -- laid out
// so that the ordinary product kernels can run on it with per-hyper-parameter term views.
struct obhip_gradbasis {
  uint64_t model_version = ~0ull;
  uint64_t id = 0;  // unique per build: tables derived from the column layout are keyed by it
  std::vector<obhip::GradHyp> hyps_h;
  obhip::DevBuf<obhip::GradHyp> hyps;
  obhip::DevBuf<int> ge0col;   // gecol of every hyper-parameter (the level-0 gradient columns), device
  obhip::DevBuf<double> rotg;  // per hyper-parameter [m][ncolp]
  obhip::DevBuf<double> kd;    // per knot: log(knot) * t(knot) (mat25pow), else 0
  std::unique_ptr<obhip_basis> gb;    // combined array + extended dimension table
  std::unique_ptr<obhip_basis> gbsq;  // its squared store (basematsq / basematsq_gradhyp)
};

// ---- outerbase ------------------------------------------------------------------
struct obhip_basis {
  const obhip_model *model = nullptr;
  uint64_t n = 0, n_pad = 0, d = 0;
  obhip::ModelDev md;
  obhip::DevBuf<double> x;      // column-major n x d (ld = n)
  obhip::DevBuf<double> bm;     // tile-blocked [n_pad/64][Mc][64]
  obhip::DevBuf<double> scale;  // n_pad (0 beyond n)
  obhip::DevBuf<char> work;     // scratch for split-reduction partials (grown on demand)
  obhip::DevBuf<double> bmat;   // row-major design matrix [n_pad][p_pad], staging of the
                                // materialised-B Gram kernel (allocated on first use)
  uint64_t bmat_terms = 0;      // uid of the terms bmat currently holds (0: none)
  obhip::DevBuf<uint64_t> gram_pairs;  // XCD-aware (tile pair, row split) task order of that kernel
  int gram_pairs_nb = -1, gram_pairs_ns = -1;
  bool gram_pairs_diag4 = false, gram_pairs_cont = false;
  std::unique_ptr<obhip_gradbasis> grad;  // built on first *_gradhyp call, dropped on rebuild
  int device = 0;
  int workspace(size_t bytes, void **out) {
    if (work.n < bytes) {
      // the previous kernels may still be reading the old buffer
      if (work.p && hipStreamSynchronize(obhip::cur_stream()) != hipSuccess)
        return obhip::fail(OBHIP_ERR_HIP, "stream sync failed");
      int rc = work.alloc(bytes + bytes / 4);
      if (rc) return rc;
    }
    *out = work.p;
    return 0;
  }
};

namespace obhip {

constexpr int kTileRows = 64;
// doubles of LDS k_build_basis has for one dimension's interval tables; ModelDev::build makes
// tables only for dimensions whose tables fit (larger ones would be per-lane global gathers, no
// faster than the knot loop, and cost the host O(knots^2 x levels) per hyper-parameter update)
constexpr int kIntervalTabMax = 2048;

// kernels_basis.hip
int launch_build_basis(obhip_basis &b);
int launch_getbase(const obhip_basis &b, uint64_t k, double *d_out /* n x m */);
// kernels_prod.hip
int launch_getmat(const obhip_basis &b, obhip_terms &t, double *d_out, uint64_t ld = 0);
int launch_mm(const obhip_basis &b, obhip_terms &t, const double *d_a,
              double *d_out, bool squared);
int launch_tmm(const obhip_basis &b, obhip_terms &t, const double *d_a,
               double *d_out, bool squared);
// d_out (p) = B^T (c_a B a + c_b y) in ONE pass over the basis; d_yhat (n) = B a and d_ss (1) =
// sum (B a - y)^2 when asked for.  Returns kNotFused (nothing done) when the terms do not fit the
// fused kernel: the caller then composes launch_mm / launch_tmm.
constexpr int kNotFused = -1;
// d_stop0 / d_stop1 (device scalars, may be null): the launch does nothing when either is non-zero
// at the time it RUNS -- for launches enqueued before the host knows whether they are needed; only
// where hessmult_fused_skippable says so
// then (may be null): q[k] = e2 d_out[k] + prec[k] pv[k] written by the reduction of the row-split
// partials as well -- the PCG's q = H pv (lpdfvec::hessmult, fit.cpp:382-392) without a launch of its own
struct HmThen {
  double e2;
  const double *prec, *pv;
  double *q;
};
int launch_hessmult_fused(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y,
                          double ca, double cb, double *d_out, double *d_yhat, double *d_ss,
                          const double *d_stop0 = nullptr, const double *d_stop1 = nullptr,
                          const HmThen *then = nullptr);
bool hessmult_fused_skippable(const obhip_basis &b, obhip_terms &t);
// d_out = B^T a and d_out2 = (B^2)^T a2 (a2 null: ones) in one pass; kNotFused: make two passes
int launch_tmm_dual(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, const double *d_a2,
                    double *d_out2);
// kernels_generic.hip: any number of used columns / factors, columns read from HBM
int launch_mm_generic(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out, int mode,
                      uint64_t ld);
int launch_tmm_generic(const obhip_basis &b, obhip_terms &t, const double *d_a, double *d_out,
                       bool squared);
int launch_materialize_generic(const obhip_basis &b, obhip_terms &t, double *d_B);
// kernels_gram.hip
// where k_gram_reduce puts the summed tiles: full symmetric p x p (raw G, or with `form` the
// Hessian e2 G + diag(prec) and its diagonal) or the packed upper triangle of a row-sharded
// fit's exchange buffer
struct GramSink {
  double *out = nullptr;
  bool packed = false;
  bool form = false;
  double e2 = 1.0;
  const double *prec = nullptr;  // p, device
  double *diagH = nullptr;       // p, device, may be null
};
// a fit's request to take g = B^T y along when the design matrix is staged (the staging pass
// forms every entry of B anyway); `done` tells the caller whether that pass ran and did it --
// not when the staged matrix of these terms was still valid, nor on the chunked / fused paths
struct GramFuse {
  const double *y = nullptr;  // n, device
  double *g = nullptr;        // p, device
  bool done = false;
};
int launch_gram(const obhip_basis &b, obhip_terms &t, double *d_G);
int launch_gram_to(const obhip_basis &b, obhip_terms &t, const GramSink &sink, GramFuse *fuse = nullptr);
int launch_gram_reduce(const double *part, int npairs, int nsplit, int nb, int p, const GramSink &sink,
                       bool accumulate, bool last);
void set_gram_backend(int b);
int get_gram_backend();
// kernels_gram_panel.hip
int launch_materialize_rows(const obhip_basis &b, obhip_terms &t, double *d_B, GramFuse *fuse = nullptr);
int ensure_bmat(obhip_basis &b, obhip_terms &t, GramFuse *fuse = nullptr);  // b.bmat = design matrix of (b, t)
bool gram_panel_supports(const obhip_basis &b, const obhip_terms &t);
// C = A^T Bm on the matrix cores (kernels_gram_panel.hip): mode 1 = row norms of C into
// out[J * ldo + i] per 128-column tile J, mode 2 = C stored row-major with ldo
int launch_atb(int mode, const double *A, uint64_t ldA, uint64_t M, const double *Bm, uint64_t ldB,
               uint64_t N, uint64_t K, bool tri, double *out, uint64_t ldo);
// kernels_trtri.hip: X (pp x pp, zeroed) = L^-T in its upper triangle; transpose of an n x n block
int launch_trtri_lt(const double *d_L, uint64_t ldl, uint64_t p, double *d_X, uint64_t pp,
                    double *d_dinv);
int launch_transpose(const double *d_in, uint64_t ldi, double *d_out, uint64_t ldo, uint64_t n);
// posterior.cpp: Cholesky factor of the total Hessian and X = L^-T for predr_std
struct PostFactor {
  uint64_t p = 0, pp = 0;  // pp = p rounded up to 128
  DevBuf<double> L;        // p x p row-major, lower triangle = L
  DevBuf<double> X;        // pp x pp row-major, upper triangle = L^-T, zero elsewhere
};
int post_factor_build(const double *d_H, uint64_t p, PostFactor &f, bool want_inverse);
int post_var_dev(const obhip_model &m, obhip_terms &t, const PostFactor &f, const double *d_x,
                 uint64_t n, double e2sigma, double *d_var);
// kernels_grad.hip
int ensure_gradbasis(obhip_basis &b);
int ensure_gradbasis_sq(obhip_basis &b);
obhip_terms *grad_view(obhip_terms &t, const obhip_basis &b, uint64_t h);
// device-level forms (inputs and n-sized results in HBM) for the likelihood classes
int grad_mm_dev(obhip_basis &b, obhip_terms &t, bool squared, const double *a_host, const double *d_a,
                double *d_M, DevBuf<double> &dge);
int grad_mm_dot_dev(obhip_basis &b, obhip_terms &t, const double *a_host, const double *d_M,
                    const double *d_w, double *out_host);
// staging class of a column of a k_tmm_d3 tile, in bits 28-29 of its entry in the group's ucol list
constexpr uint32_t kD3Delta = 1u << 28, kD3Own = 2u << 28, kD3ColMask = (1u << 28) - 1;
constexpr int kD3Pre = 20;  // k_tmm_d3: prefetch registers per thread => at most 8 * 20 columns per group
// one group of obhip_terms::ge_d3 (kernels_grad_d3.hip): d_out (device, [2 nh][v.p]) = u1 of the
// group's first hyper-parameter, of its second, u2 likewise; mode bit 0: u1, bit 1: u2
int launch_tmm_d3(obhip_basis &b, const obhip_terms::GeD3 &g, int mode, const double *d_w1, const double *d_w2,
                  double *d_out);
int grad_dual_dev(obhip_basis &b, obhip_terms &t, const double *a_host, const double *d_M, const double *d_w1,
                  const double *d_w2, double *out_dot, double *out_sq);
int grad_wdot_dev(const double *d_G, const double *d_w, uint64_t n, uint64_t ncol, double *out_host);
int grad_tmm_host(obhip_basis &b, obhip_terms &t, bool squared, const double *d_a, double *out_host);
// kernels_chol.hip
uint64_t newton_workspace_bytes(uint64_t p);
bool materialize_tl_supports(const obhip_terms &t);
int launch_materialize_tl(const obhip_basis &b, obhip_terms &t, double *d_B, const double *d_y = nullptr,
                          double *d_g = nullptr);
int launch_newton_solve(uint64_t p, double *d_H, const double *d_rhs,
                        double *d_theta, void *d_ws, uint64_t ws_bytes);
int launch_form_hessian(uint64_t p, double *d_G, const double *d_prec,
                        double e2, double *d_diagH);
// kernels_predict.hip
int launch_predict(const obhip_model &m, obhip_terms &t, const double *d_theta,
                   const double *d_x, uint64_t n, double *d_mean,
                   const double *d_coeffvar, double e2sigma, double *d_var);
// small vector kernels (kernels_misc.hip)
int launch_synth(uint64_t seed, uint64_t row0, uint64_t n, uint64_t d,
                 const int *d_kinds, double *d_x, double *d_y);
int launch_sum_sumsq(const double *d_v, uint64_t n, double *d_out2,
                     double *d_part /* 2048 doubles of scratch */);
int launch_affine(double *d_v, uint64_t n, double cent, double sca);
int launch_resid(const double *d_yhat, const double *d_y, uint64_t n, double e2,
                 double *d_r, double *d_diff);
int launch_scale(double *d_v, uint64_t n, double c);
int launch_fill(double *d_v, uint64_t n, double c);
// comm.cpp: in-place sum over the ranks of c (no-op for c == nullptr or one rank)
int comm_allreduce(obhip_comm *c, double *d_buf, uint64_t count);
int comm_nranks(const obhip_comm *c);
// obhip_fit_cg_dev with the non-finite exit of fit.cpp:53-56 reported (finite_out = 0, val = -inf)
int fit_cg_dev_impl(const obhip_basis *b, const obhip_terms *t, const obhip_model *m, const double *d_y,
                    double sigma, double rho, double tol, uint64_t maxit, double *d_theta,
                    uint64_t *iters_out, double *d_diagH, double *val_out, obhip_comm *comm,
                    int *finite_out,
                    double *d_sqcolsums_out = nullptr);

}  // namespace obhip
