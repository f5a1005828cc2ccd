// Reference-side binding of libobhip: replaces src/interfaceR.cpp of MattPlumlee/outerbase.
// Same module name (obmod), same free functions, same class / method / field names as
// RCPP_MODULE(obmod) (src/interfaceR.cpp:661-793), so every R file of the package
// (R/fitting.R, R/outersupport.R, the vignettes and tests) runs unchanged; each body is a
// thin call into the C ABI of include/obhip.h.  The C++ files of the reference
// (covfuncs.cpp, modandbase.cpp, linalg.cpp, fit.cpp, lpdfs/*.cpp) drop out of the build.
//
// Built by R CMD INSTALL with the Makevars shown in INTEGRATION.md (needs Rcpp only: no
// RcppArmadillo, no OpenMP).  R, Rcpp and RcppArmadillo are not in the image this repository
// is developed in, so this file has not been built against Rcpp there (its syntax and its calls
// into the ABI are checked by g++ -fsyntax-only against a declaration-only stand-in for
// <Rcpp.h>, tests/rcpp_stub/Rcpp.h -- test infrastructure); tests/cabi_smoke.c exercises
// the same sequence of ABI calls from plain C, and outerbase_amd/obmod.py is the same mapping
// over ctypes, which the test-suite runs.
//
// Line references: "ref" = file:line in the reference checkout.
#include <Rcpp.h>

#include <string>
#include <vector>

#include "obhip.h"

using namespace Rcpp;

namespace {

inline void ck(int rc) {  // status code -> R error (the reference throws C++ exceptions)
  if (rc != 0) stop(obhip_last_error());
}

// umat arrives from R as a numeric / integer matrix of 0-based levels
// (ref tests/testthat/test-obombasic.R:49); column-major is kept
std::vector<uint64_t> to_u64(const NumericMatrix &t) {
  std::vector<uint64_t> out(t.size());
  for (R_xlen_t i = 0; i < t.size(); ++i) {
    if (t[i] < 0) stop("terms must be non-negative levels");
    out[i] = (uint64_t)t[i];
  }
  return out;
}

int kind_of(const std::string &s) {
  if (s == "mat25") return OBHIP_COV_MAT25;
  if (s == "mat25pow") return OBHIP_COV_MAT25POW;
  if (s == "mat25ang") return OBHIP_COV_MAT25ANG;
  stop("need to choose one of the existing cov functions");
  return -1;
}

}  // namespace

// ---- covf, covf_mat25, covf_mat25pow, covf_mat25ang (ref src/covfuncs.h:4-67;
// module rows ref src/interfaceR.cpp:764-791) ------------------------------------------------
class covf {
public:
  int kind = -1;
  NumericVector hyp, hyplb, hypub, hyp0, hypvar;
  double lowbnd = 0, uppbnd = 0;
  covf() {}
  explicit covf(int k) : kind(k) {
    int nh = 0;
    ck(obhip_cov_numhyp(k, &nh));
    hyp0 = NumericVector(nh);
    hyplb = NumericVector(nh);
    hypub = NumericVector(nh);
    hypvar = NumericVector(nh);
    ck(obhip_cov_info(k, hyp0.begin(), hyplb.begin(), hypub.begin(), hypvar.begin(), &lowbnd, &uppbnd));
    hyp = clone(hyp0);
  }
  virtual ~covf() {}
  NumericMatrix cov(NumericVector x1, NumericVector x2) {           // ref covfuncs.cpp:113-126,197-212,285-310
    if (kind < 0) return NumericMatrix(0, 0);
    NumericMatrix out(x1.size(), x2.size());
    ck(obhip_cov(kind, hyp.begin(), x1.begin(), x1.size(), x2.begin(), x2.size(), out.begin()));
    return out;
  }
  NumericVector covdiag(NumericVector x) {                          // covmdiag, ref covfuncs.cpp:128,214,312
    return NumericVector(x.size(), 1.0);
  }
  NumericVector cov_gradhyp(NumericVector x1, NumericVector x2) {   // ref covfuncs.cpp:134-150,220-243,318-347
    if (kind < 0) return NumericVector(0);
    NumericVector out(Dimension(x1.size(), x2.size(), hyp.size()));
    ck(obhip_cov_gradhyp(kind, hyp.begin(), x1.begin(), x1.size(), x2.begin(), x2.size(), out.begin()));
    return out;
  }
};
class covf_mat25 : public covf { public: covf_mat25() : covf(OBHIP_COV_MAT25) {} };
class covf_mat25pow : public covf { public: covf_mat25pow() : covf(OBHIP_COV_MAT25POW) {} };
class covf_mat25ang : public covf { public: covf_mat25ang() : covf(OBHIP_COV_MAT25ANG) {} };

// ---- outermod (ref src/modandbase.h:9-54; module rows ref src/interfaceR.cpp:670-678) --------
class outermod {
public:
  obhip_model *h = nullptr;
  std::vector<std::string> covnames;
  outermod() {}
  ~outermod() { obhip_model_destroy(h); }
  void need() const {
    if (!h) throw std::range_error("Need to set cov. funcs before setting knots.");
  }
  uint64_t d() const { uint64_t v; ck(obhip_model_dims(h, &v, 0, 0, 0)); return v; }
  uint64_t nhyp() const { uint64_t v; ck(obhip_model_dims(h, 0, 0, 0, &v)); return v; }
  void updatehyp(NumericVector hyp) {                               // hyp_set, ref modandbase.cpp:161-202
    need();
    ck(obhip_model_set_hyp(h, hyp.begin(), hyp.size()));
  }
  NumericMatrix selectterms(unsigned int numele) {                  // ref modandbase.cpp:387-440
    need();
    const uint64_t dd = d();
    std::vector<uint64_t> t((uint64_t)numele * dd);
    // ties among the near-best candidates: the reference shuffles them with R's RNG
    // (ref modandbase.cpp:408); a seed drawn from R's RNG keeps set.seed() in charge
    const uint64_t seed = 1 + (uint64_t)(R::unif_rand() * 4294967295.0);
    ck(obhip_model_select_terms(h, numele, seed, t.data()));
    NumericMatrix out(numele, dd);
    std::copy(t.begin(), t.end(), out.begin());
    return out;
  }
  NumericVector getvar(NumericMatrix terms) {                       // ref modandbase.cpp:350-356
    need();
    std::vector<uint64_t> t = to_u64(terms);
    NumericVector out(terms.nrow());
    ck(obhip_model_term_var(h, t.data(), terms.nrow(), out.begin()));
    return out;
  }
  NumericMatrix getlvar_gradhyp(NumericMatrix terms) {              // ref modandbase.cpp:364-379
    need();
    std::vector<uint64_t> t = to_u64(terms);
    NumericMatrix out(terms.nrow(), nhyp());
    ck(obhip_model_term_lvar_gradhyp(h, t.data(), terms.nrow(), out.begin()));
    return out;
  }
  double hyplpdf(NumericVector hyp) {                               // ref modandbase.cpp:89-99
    need();
    double v;
    ck(obhip_model_hyplpdf(h, hyp.begin(), hyp.size(), &v));
    return v;
  }
  NumericVector hyplpdf_grad(NumericVector hyp) {                   // ref modandbase.cpp:106-118
    need();
    NumericVector out(nhyp());
    ck(obhip_model_hyplpdf_grad(h, hyp.begin(), hyp.size(), out.begin()));
    return out;
  }
};

// ref src/interfaceR.cpp:53-73
void setcovfs(outermod &om, StringVector covstr) {
  std::vector<int> k;
  om.covnames.clear();
  for (R_xlen_t l = 0; l < covstr.size(); ++l) {
    om.covnames.push_back(as<std::string>(covstr[l]));
    k.push_back(kind_of(om.covnames.back()));
  }
  obhip_model_destroy(om.h);
  om.h = nullptr;
  ck(obhip_model_create(&om.h, k.size(), k.data()));
}

// ref src/interfaceR.cpp:94-149 (dimension and range checks come back as the same messages)
void setknot(outermod &om, List L) {
  om.need();
  if ((uint64_t)L.size() != om.d()) {
    std::string m = "dim needs to match";
    m += std::to_string(om.d());
    m += ".";
    throw std::range_error(m);
  }
  std::vector<uint64_t> st(1, 0);
  std::vector<double> kp;
  for (R_xlen_t l = 0; l < L.size(); ++l) {
    NumericVector v = L[l];
    kp.insert(kp.end(), v.begin(), v.end());
    st.push_back(kp.size());
  }
  const int rc = obhip_model_set_knots(om.h, st.data(), kp.data());
  if (rc == OBHIP_ERR_INVALID) throw std::range_error(obhip_last_error());
  ck(rc);
}

// ref src/interfaceR.cpp:167-180
NumericVector gethyp(outermod &om) {
  om.need();
  const uint64_t nh = om.nhyp();
  NumericVector out(nh);
  ck(obhip_model_get_hyp(om.h, out.begin()));
  std::vector<uint64_t> hypmatch(nh), gest(nh + 1);
  ck(obhip_model_grad_layout(om.h, 0, hypmatch.data(), gest.data()));
  CharacterVector names(nh);
  uint64_t prev = ~0ull, within = 0;
  for (uint64_t k = 0; k < nh; ++k) {
    within = hypmatch[k] == prev ? within + 1 : 0;
    prev = hypmatch[k];
    const std::string &c = om.covnames[hypmatch[k]];
    const char *hn = c == "mat25" ? "scale" : c == "mat25pow" ? (within ? "power" : "scale")
                                                              : (within ? "cos.sc" : "sin.sc");
    names[k] = "inpt" + std::to_string(1 + hypmatch[k]) + "." + hn;   // ref covfuncs.cpp:93,172,260
  }
  out.names() = names;
  return out;
}

// ---- outerbase (ref src/modandbase.h:57-125; module rows ref src/interfaceR.cpp:680-694) ----
class outerbase {
public:
  obhip_basis *h = nullptr;
  const outermod &om;
  uint64_t n;
  int nthreads = 1;       // OpenMP knobs of the reference (ref modandbase.cpp:504-513): kept
  bool vertpl = false;    // as plain fields for scripts that set them; no meaning on the GPU
  unsigned int chunksize = 0, loopsize = 0;
  outerbase(const outermod &om_, NumericMatrix x) : om(om_), n(x.nrow()) {
    om.need();
    if ((uint64_t)x.ncol() != om.d()) stop("x must have one column per input dimension");
    ck(obhip_basis_create(&h, om.h, x.begin(), n, n, nullptr));     // copies x like `const mat xp`
  }
  ~outerbase() { obhip_basis_destroy(h); }
  struct Terms {                                                    // umat by value -> device tables
    obhip_terms *t = nullptr;
    uint64_t p;
    Terms(const outermod &om, const NumericMatrix &terms) : p(terms.nrow()) {
      std::vector<uint64_t> u = to_u64(terms);
      ck(obhip_terms_create(&t, om.h, u.data(), p));
    }
    ~Terms() { obhip_terms_destroy(t); }
  };
  void build() { ck(obhip_basis_rebuild(h)); }                      // ref modandbase.cpp:547-626
  NumericMatrix getbase(unsigned int k) {                           // ref modandbase.cpp:634-639 (1-based)
    if (k < 1 || k > om.d()) stop("dimension out of range (1-based)");
    std::vector<uint64_t> st(om.d() + 1);
    ck(obhip_model_get_knots(om.h, st.data(), nullptr));
    NumericMatrix out(n, st[k] - st[k - 1]);
    ck(obhip_basis_getbase(h, k, out.begin()));
    return out;
  }
  NumericMatrix getmat(NumericMatrix terms) {                       // ref modandbase.cpp:649-654, getm_ linalg.cpp:647-715
    Terms t(om, terms);
    NumericMatrix out(n, t.p);
    ck(obhip_basis_getmat(h, t.t, out.begin()));
    return out;
  }
  NumericVector matmul(NumericMatrix terms, NumericVector a) {      // mm_out, ref modandbase.cpp:687-692
    Terms t(om, terms);
    if ((uint64_t)a.size() != t.p) stop("non-conformable arguments");
    NumericVector out(n);
    ck(obhip_basis_mm(h, t.t, a.begin(), 1, out.begin()));
    return out;
  }
  NumericVector tmatmul(NumericMatrix terms, NumericVector a) {     // tmm_out, ref modandbase.cpp:711-716
    Terms t(om, terms);
    if ((uint64_t)a.size() != n) stop("non-conformable arguments");
    NumericVector out(t.p);
    ck(obhip_basis_tmm(h, t.t, a.begin(), 1, out.begin()));
    return out;
  }
  NumericVector getmat_gradhyp(NumericMatrix terms) {               // ref modandbase.cpp:663-669 (cube)
    Terms t(om, terms);
    NumericVector out(Dimension(n, t.p, om.nhyp()));
    ck(obhip_basis_getmat_gradhyp(h, t.t, out.begin()));
    return out;
  }
  NumericMatrix matmul_gradhyp(NumericMatrix terms, NumericVector a) {   // mm_gradhyp_out, ref modandbase.cpp:739-744
    Terms t(om, terms);
    if ((uint64_t)a.size() != t.p) stop("non-conformable arguments");
    NumericMatrix out(n, om.nhyp());
    ck(obhip_basis_mm_gradhyp(h, t.t, a.begin(), nullptr, out.begin()));
    return out;
  }
  NumericMatrix tmatmul_gradhyp(NumericMatrix terms, NumericVector a) {  // tmm_gradhyp_out, ref modandbase.cpp:771-776
    Terms t(om, terms);
    if ((uint64_t)a.size() != n) stop("non-conformable arguments");
    NumericMatrix out(t.p, om.nhyp());
    ck(obhip_basis_tmm_gradhyp(h, t.t, a.begin(), nullptr, out.begin()));
    return out;
  }
};

// ---- lpdf and descendants (ref src/fit.h:23-361; module rows ref src/interfaceR.cpp:696-762) --
class lpdf {
public:
  obhip_lpdf *h = nullptr;
  lpdf() {}                       // the module exposes the bare base class too (ref :697)
  virtual ~lpdf() { obhip_lpdf_destroy(h); }
  void need() const { if (!h) stop("this lpdf has no model behind it"); }
  // fields (ref :698-709)
  bool flag(int f) const { need(); int v; ck(obhip_lpdf_get_flag(h, f, &v)); return v != 0; }
  void set_flag(int f, bool v) { need(); ck(obhip_lpdf_set_flag(h, f, v)); }
  bool get_compute_val() { return flag(OBHIP_FLAG_COMPUTE_VAL); }
  void set_compute_val(bool v) { set_flag(OBHIP_FLAG_COMPUTE_VAL, v); }
  bool get_compute_grad() { return flag(OBHIP_FLAG_COMPUTE_GRAD); }
  void set_compute_grad(bool v) { set_flag(OBHIP_FLAG_COMPUTE_GRAD, v); }
  // bound by name; the reference binds R's compute_gradpara to C++ compute_gradhyp and
  // vice versa (ref :700-701), which only shows when the two are set differently
  bool get_compute_gradhyp() { return flag(OBHIP_FLAG_COMPUTE_GRADHYP); }
  void set_compute_gradhyp(bool v) { set_flag(OBHIP_FLAG_COMPUTE_GRADHYP, v); }
  bool get_compute_gradpara() { return flag(OBHIP_FLAG_COMPUTE_GRADPARA); }
  void set_compute_gradpara(bool v) { set_flag(OBHIP_FLAG_COMPUTE_GRADPARA, v); }
  bool get_fullhess() { return flag(OBHIP_FLAG_FULLHESS); }
  double get_val() { need(); double v; ck(obhip_lpdf_get_val(h, &v)); return v; }
  NumericVector vec(int which) const {
    need();
    uint64_t len = 0;
    ck(obhip_lpdf_get_vec(h, which, nullptr, 0, &len));
    NumericVector out(len);
    ck(obhip_lpdf_get_vec(h, which, out.begin(), len, nullptr));
    return out;
  }
  NumericVector get_coeff() { return vec(OBHIP_VEC_COEFF); }
  NumericVector get_grad() { return vec(OBHIP_VEC_GRAD); }
  NumericVector get_gradhyp() { return vec(OBHIP_VEC_GRADHYP); }
  NumericVector get_gradpara() { return vec(OBHIP_VEC_GRADPARA); }
  NumericVector get_para() { return vec(OBHIP_VEC_PARA); }
  unsigned int get_nterms() { need(); uint64_t v; ck(obhip_lpdf_dims(h, 0, &v, 0, 0, 0)); return v; }
  uint64_t npara() const { uint64_t v; ck(obhip_lpdf_dims(h, 0, 0, &v, 0, 0)); return v; }
  uint64_t nhyp() const { uint64_t v; ck(obhip_lpdf_dims(h, 0, 0, 0, &v, 0)); return v; }
  // methods (ref :710-722)
  void setnthreads(int k) { need(); ck(obhip_lpdf_setnthreads(h, k)); }
  void optcg(double tol, unsigned int maxepch) { need(); ck(obhip_lpdf_optcg(h, tol, maxepch, nullptr)); }   // ref fit.cpp:37-96
  void optnewton() { need(); ck(obhip_lpdf_optnewton(h)); }                                                // ref fit.cpp:98-131
  void update(NumericVector coeff) { need(); ck(obhip_lpdf_update(h, coeff.begin(), coeff.size())); }
  void updateom() { need(); ck(obhip_lpdf_updateom(h)); }
  void updatepara(NumericVector para) { need(); ck(obhip_lpdf_updatepara(h, para.begin(), para.size())); }
  void updateterms(NumericMatrix terms) {
    need();
    std::vector<uint64_t> t = to_u64(terms);
    ck(obhip_lpdf_updateterms(h, t.data(), terms.nrow()));
  }
  NumericVector hessmult(NumericVector g) {
    need();
    if ((uint64_t)g.size() != get_nterms()) stop("non-conformable arguments");
    NumericVector out(g.size());
    ck(obhip_lpdf_hessmult(h, g.begin(), out.begin()));
    return out;
  }
  NumericVector diaghess() { need(); NumericVector out(get_nterms()); ck(obhip_lpdf_diaghess(h, out.begin())); return out; }
  NumericMatrix diaghessgradhyp() {
    need();
    NumericMatrix out(get_nterms(), nhyp());
    ck(obhip_lpdf_diaghessgradhyp(h, out.begin()));
    return out;
  }
  NumericMatrix diaghessgradpara() {
    need();
    NumericMatrix out(get_nterms(), npara());
    ck(obhip_lpdf_diaghessgradpara(h, out.begin()));
    return out;
  }
  double paralpdf(NumericVector p) { need(); double v; ck(obhip_lpdf_paralpdf(h, p.begin(), p.size(), &v)); return v; }   // ref fit.cpp:133-139
  NumericVector paralpdf_grad(NumericVector p) {                                                                        // ref fit.cpp:146-157
    need();
    NumericVector out(npara());
    ck(obhip_lpdf_paralpdf_grad(h, p.begin(), p.size(), out.begin()));
    return out;
  }
};

// ref src/interfaceR.cpp:193-199
NumericVector getpara(lpdf &logpdf) {
  NumericVector out = logpdf.get_para();
  CharacterVector names(out.size());
  for (R_xlen_t k = 0; k < out.size(); ++k) {
    const char *nm = nullptr;
    ck(obhip_lpdf_paraname(logpdf.h, k, &nm));
    names[k] = nm;
  }
  out.names() = names;
  return out;
}

class loglik : public lpdf {
public:
  loglik(int kind, const outermod &om, NumericMatrix terms, NumericVector y, NumericMatrix x) {
    om.need();
    if (x.nrow() != y.size()) stop("x and y dims do not align");
    std::vector<uint64_t> t = to_u64(terms);
    ck(obhip_loglik_create(&h, kind, om.h, t.data(), terms.nrow(), y.begin(), x.begin(), x.nrow(), x.nrow()));
  }
  NumericVector get_yhat() { return vec(OBHIP_VEC_YHAT); }
};
class loglik_std : public loglik {      // ref src/lpdfs/loglik_std.cpp:41-203
public:
  loglik_std(const outermod &om, NumericMatrix terms, NumericVector y, NumericMatrix x)
      : loglik(OBHIP_LPDF_LOGLIK_STD, om, terms, y, x) {}
};
class loglik_gauss : public loglik {    // ref src/lpdfs/loglik_gauss.cpp:41-172
public:
  loglik_gauss(const outermod &om, NumericMatrix terms, NumericVector y, NumericMatrix x)
      : loglik(OBHIP_LPDF_LOGLIK_GAUSS, om, terms, y, x) {}
};
class loglik_gda : public loglik {      // ref src/lpdfs/loglik_gda.cpp:48-235
public:
  loglik_gda(const outermod &om, NumericMatrix terms, NumericVector y, NumericMatrix x)
      : loglik(OBHIP_LPDF_LOGLIK_GDA, om, terms, y, x) {}
  bool get_dodiag() { return flag(OBHIP_FLAG_DODIAG); }
  void set_dodiag(bool v) { set_flag(OBHIP_FLAG_DODIAG, v); }
};
class logpr_gauss : public lpdf {       // ref src/lpdfs/logpr_gauss.cpp:41-186
public:
  logpr_gauss(const outermod &om, NumericMatrix terms) {
    om.need();
    std::vector<uint64_t> t = to_u64(terms);
    ck(obhip_logpr_gauss_create(&h, om.h, t.data(), terms.nrow()));
  }
  NumericVector get_coeffsd() { return vec(OBHIP_VEC_COEFFSD); }
};
class lpdfvec : public lpdf {           // ref src/fit.cpp:174-612
public:
  lpdfvec(lpdf &a, lpdf &b) { a.need(); b.need(); ck(obhip_lpdfvec_create(&h, a.h, b.h)); }
  bool get_domarg() { return flag(OBHIP_FLAG_DOMARG); }
  void set_domarg(bool v) { set_flag(OBHIP_FLAG_DOMARG, v); }
};

// ---- predictor (ref src/fit.h:352-361; module rows ref src/interfaceR.cpp:725-731) -----------
class predictor {
public:
  obhip_predictor *h = nullptr;
  predictor(const lpdf &logpdf) {
    logpdf.need();
    const int rc = obhip_predictor_create(&h, logpdf.h);
    if (rc == OBHIP_ERR_INVALID) throw std::invalid_argument(obhip_last_error());   // ref fit.h:53
    ck(rc);
  }
  ~predictor() { obhip_predictor_destroy(h); }   // (the reference never frees pred, ref fit.h:354-356)
  void update(NumericMatrix x) { ck(obhip_predictor_update(h, x.begin(), x.nrow(), x.nrow())); }
  NumericVector mean() { uint64_t n; ck(obhip_predictor_n(h, &n)); NumericVector out(n); ck(obhip_predictor_mean(h, out.begin())); return out; }
  NumericVector var() { uint64_t n; ck(obhip_predictor_n(h, &n)); NumericVector out(n); ck(obhip_predictor_var(h, out.begin())); return out; }
  void setnthreads(int k) { ck(obhip_predictor_setnthreads(h, k)); }
};

RCPP_EXPOSED_CLASS(outerbase)
RCPP_EXPOSED_CLASS(outermod)
RCPP_EXPOSED_CLASS(lpdf)

RCPP_MODULE(obmod) {                                       // ref src/interfaceR.cpp:661-793
  function("setcovfs", &setcovfs, "type ?setcovfs");
  function("setknot", &setknot, "type ?setknot");
  function("gethyp", &gethyp, "type ?gethyp");
  function("getpara", &getpara, "type ?getpara");

  class_<outermod>("outermod")
      .constructor()
      .method("updatehyp", &outermod::updatehyp)
      .method("selectterms", &outermod::selectterms)
      .method("getvar", &outermod::getvar)
      .method("getlvar_gradhyp", &outermod::getlvar_gradhyp)
      .method("hyplpdf", &outermod::hyplpdf)
      .method("hyplpdf_grad", &outermod::hyplpdf_grad);

  class_<outerbase>("outerbase")
      .constructor<const outermod &, NumericMatrix>()
      .field("nthreads", &outerbase::nthreads)
      .field("vertpl", &outerbase::vertpl)
      .field_readonly("chunksize", &outerbase::chunksize)
      .field_readonly("loopsize", &outerbase::loopsize)
      .method("getbase", &outerbase::getbase)
      .method("getmat", &outerbase::getmat)
      .method("build", &outerbase::build)
      .method("matmul", &outerbase::matmul)
      .method("tmatmul", &outerbase::tmatmul)
      .method("getmat_gradhyp", &outerbase::getmat_gradhyp)
      .method("matmul_gradhyp", &outerbase::matmul_gradhyp)
      .method("tmatmul_gradhyp", &outerbase::tmatmul_gradhyp);

  class_<lpdf>("lpdf")
      .constructor()
      .property("compute_val", &lpdf::get_compute_val, &lpdf::set_compute_val)
      .property("compute_grad", &lpdf::get_compute_grad, &lpdf::set_compute_grad)
      .property("compute_gradpara", &lpdf::get_compute_gradpara, &lpdf::set_compute_gradpara)
      .property("compute_gradhyp", &lpdf::get_compute_gradhyp, &lpdf::set_compute_gradhyp)
      .property("fullhess", &lpdf::get_fullhess)
      .property("val", &lpdf::get_val)
      .property("coeff", &lpdf::get_coeff)
      .property("grad", &lpdf::get_grad)
      .property("gradhyp", &lpdf::get_gradhyp)
      .property("gradpara", &lpdf::get_gradpara)
      .property("para", &lpdf::get_para)
      .property("nterms", &lpdf::get_nterms)
      .method("setnthreads", &lpdf::setnthreads)
      .method("optcg", &lpdf::optcg)
      .method("optnewton", &lpdf::optnewton)
      .method("update", &lpdf::update)
      .method("updateom", &lpdf::updateom)
      .method("updatepara", &lpdf::updatepara)
      .method("updateterms", &lpdf::updateterms)
      .method("hessmult", &lpdf::hessmult)
      .method("diaghess", &lpdf::diaghess)
      .method("diaghessgradhyp", &lpdf::diaghessgradhyp)
      .method("diaghessgradpara", &lpdf::diaghessgradpara)
      .method("paralpdf", &lpdf::paralpdf)
      .method("paralpdf_grad", &lpdf::paralpdf_grad);

  class_<predictor>("predictor")
      .constructor<const lpdf &>()
      .method("update", &predictor::update)
      .method("mean", &predictor::mean)
      .method("var", &predictor::var)
      .method("setnthreads", &predictor::setnthreads);

  class_<loglik_std>("loglik_std")
      .derives<lpdf>("lpdf")
      .constructor<const outermod &, NumericMatrix, NumericVector, NumericMatrix>()
      .property("yhat", &loglik_std::get_yhat);

  class_<loglik_gauss>("loglik_gauss")
      .derives<lpdf>("lpdf")
      .constructor<const outermod &, NumericMatrix, NumericVector, NumericMatrix>()
      .property("yhat", &loglik_gauss::get_yhat);

  class_<loglik_gda>("loglik_gda")
      .derives<lpdf>("lpdf")
      .constructor<const outermod &, NumericMatrix, NumericVector, NumericMatrix>()
      .property("dodiag", &loglik_gda::get_dodiag, &loglik_gda::set_dodiag)
      .property("yhat", &loglik_gda::get_yhat);

  class_<logpr_gauss>("logpr_gauss")
      .derives<lpdf>("lpdf")
      .constructor<const outermod &, NumericMatrix>()
      .property("coeffsd", &logpr_gauss::get_coeffsd);

  class_<lpdfvec>("lpdfvec")
      .derives<lpdf>("lpdf")
      .constructor<lpdf &, lpdf &>()
      .property("domarg", &lpdfvec::get_domarg, &lpdfvec::set_domarg);

  class_<covf>("covf")
      .constructor()
      .field("hyp", &covf::hyp)
      .field_readonly("hyplb", &covf::hyplb)
      .field_readonly("hypub", &covf::hypub)
      .field_readonly("hyp0", &covf::hyp0)
      .field_readonly("hypvar", &covf::hypvar)
      .field_readonly("lowbnd", &covf::lowbnd)
      .field_readonly("uppbnd", &covf::uppbnd)
      .method("cov", &covf::cov)
      .method("covdiag", &covf::covdiag)
      .method("cov_gradhyp", &covf::cov_gradhyp);

  class_<covf_mat25>("covf_mat25").derives<covf>("covf").constructor();
  class_<covf_mat25pow>("covf_mat25pow").derives<covf>("covf").constructor();
  class_<covf_mat25ang>("covf_mat25ang").derives<covf>("covf").constructor();
}
