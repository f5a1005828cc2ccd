"""Host-side mirror of the reference's Rcpp module `obmod`
(src/interfaceR.cpp:661-793 of MattPlumlee/outerbase): same class, method and
field names, same argument meaning (0-based term levels, 1-based getbase), so
tests read like the reference's own testthat files.  Every method forwards to
the C ABI in include/obhip.h; nothing is computed in Python except O(p) vector
algebra on results.

Scope: value paths only -- no hyper-gradient methods
(`*_gradhyp`), no loglik_gda, no marginal adjustment.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import call, ptr

_KINDS = {"mat25": 0, "mat25pow": 1, "mat25ang": 2}
_KIND_NAMES = {v: k for k, v in _KINDS.items()}
_HYPNAMES = {"mat25": ["scale"], "mat25pow": ["scale", "power"],
             "mat25ang": ["sin.sc", "cos.sc"]}


def listcov():
    """R/fitting.R:6-8"""
    return ["mat25pow", "mat25", "mat25ang"]


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _fmat(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _umat(terms):
    t = np.asarray(terms)
    if t.ndim != 2:
        raise ValueError("terms must be a p x d matrix")
    if np.any(t < 0):
        raise ValueError("terms must be non-negative levels")
    return np.asfortranarray(t.astype(np.uint64))


# ----------------------------------------------------------------------------
# covariance function classes (interfaceR.cpp:764-791)
# ----------------------------------------------------------------------------
class covf:
    kind = None

    def __init__(self):
        k = _KINDS[self.kind]
        nh = C.c_int(0)
        call("obhip_cov_numhyp", k, C.byref(nh))
        nh = nh.value
        self.hyp0 = np.zeros(nh)
        self.hyplb = np.zeros(nh)
        self.hypub = np.zeros(nh)
        self.hypvar = np.zeros(nh)
        lo, up = C.c_double(0), C.c_double(0)
        call("obhip_cov_info", k, ptr(self.hyp0), ptr(self.hyplb), ptr(self.hypub),
             ptr(self.hypvar), C.byref(lo), C.byref(up))
        self.lowbnd, self.uppbnd = lo.value, up.value
        self.hyp = self.hyp0.copy()

    def cov(self, x1, x2):
        x1, x2 = _f64(x1), _f64(x2)
        out = np.empty((len(x1), len(x2)), order="F")
        call("obhip_cov", _KINDS[self.kind], ptr(_f64(self.hyp)), ptr(x1), len(x1), ptr(x2),
             len(x2), ptr(out))
        return out

    def covdiag(self, x):
        return np.ones(len(x))  # covfuncs.cpp:128-132

    def lpdf(self, hyp):
        out = C.c_double(0)
        call("obhip_cov_hyplpdf", _KINDS[self.kind], ptr(_f64(hyp)), C.byref(out))
        return out.value


class covf_mat25(covf):
    kind = "mat25"


class covf_mat25pow(covf):
    kind = "mat25pow"


class covf_mat25ang(covf):
    kind = "mat25ang"


# ----------------------------------------------------------------------------
# outermod (interfaceR.cpp:670-678)
# ----------------------------------------------------------------------------
class outermod:
    def __init__(self):
        self._h = None
        self.covnames = []
        self.d = 0

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and _lib.lib is not None:
            _lib.lib.obhip_model_destroy(self._h)
            self._h = None

    def _need(self):
        if not self._h:
            raise RuntimeError("Need to set cov. funcs before setting knots.")

    # -- queries
    def dims(self):
        d, M, mmax, nh = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        call("obhip_model_dims", self._h, C.byref(d), C.byref(M), C.byref(mmax), C.byref(nh))
        return d.value, M.value, mmax.value, nh.value

    def rotation(self):
        d, M, mmax, _ = self.dims()
        rot = np.empty((mmax, M), order="F")
        bv = np.empty(M)
        ml = np.empty(d, dtype=np.int64)
        call("obhip_model_get_rotation", self._h, ptr(rot), ptr(bv), ptr(ml))
        return rot, bv, ml

    def set_rotation(self, rotmat, basisvar, maxlevel):
        rot = _fmat(rotmat)
        bv = _f64(basisvar)
        ml = np.ascontiguousarray(maxlevel, dtype=np.int64)
        call("obhip_model_set_rotation", self._h, ptr(rot), ptr(bv), ptr(ml))

    def grad_layout(self):
        """(hypmatch, gest): dimension and first gradient column of every hyper-parameter
        (modandbase.cpp:183-197)."""
        nh = C.c_uint64(0)
        call("obhip_model_grad_layout", self._h, C.byref(nh), None, None)
        hm = np.empty(nh.value, dtype=np.uint64)
        ge = np.empty(nh.value + 1, dtype=np.uint64)
        call("obhip_model_grad_layout", self._h, None, ptr(hm), ptr(ge))
        return hm.astype(np.int64), ge.astype(np.int64)

    def rotation_grad(self):
        """(rotmat_gradhyp, logbasisvar_gradhyp) of outermod::build (modandbase.cpp:257-274)."""
        _, _, mmax, _ = self.dims()
        ng = int(self.grad_layout()[1][-1])
        rg = np.empty((mmax, ng), order="F")
        lv = np.empty(ng)
        call("obhip_model_get_rotation_grad", self._h, ptr(rg), ptr(lv))
        return rg, lv

    def set_rotation_grad(self, rotmat_gradhyp, logbasisvar_gradhyp):
        rg = _fmat(rotmat_gradhyp)
        lv = _f64(logbasisvar_gradhyp)
        call("obhip_model_set_rotation_grad", self._h, ptr(rg), ptr(lv))

    @property
    def maxlevel(self):
        return self.rotation()[2]

    @property
    def basisvar(self):
        return self.rotation()[1]

    # -- module methods
    def updatehyp(self, hyp):
        self._need()
        hyp = _f64(hyp)
        call("obhip_model_set_hyp", self._h, ptr(hyp), len(hyp))

    def selectterms(self, numele, seed=0):
        self._need()
        d = self.d
        out = np.empty((int(numele), d), dtype=np.uint64, order="F")
        call("obhip_model_select_terms", self._h, int(numele), int(seed), ptr(out))
        return out.astype(np.int64)

    def getvar(self, terms):
        t = _umat(terms)
        out = np.empty(t.shape[0])
        call("obhip_model_term_var", self._h, ptr(t), t.shape[0], ptr(out))
        return out

    def getlvar_gradhyp(self, terms):
        """modandbase.cpp:364-379"""
        t = _umat(terms)
        nh = len(self.grad_layout()[0])
        out = np.empty((t.shape[0], nh), order="F")
        call("obhip_model_term_lvar_gradhyp", self._h, ptr(t), t.shape[0], ptr(out))
        return out

    def hyplpdf(self, hyp):
        hyp = _f64(hyp)
        out = C.c_double(0)
        call("obhip_model_hyplpdf", self._h, ptr(hyp), len(hyp), C.byref(out))
        return out.value

    def hyplpdf_grad(self, hyp):
        """modandbase.cpp:106-118"""
        hyp = _f64(hyp)
        out = np.zeros(self.dims()[3])
        call("obhip_model_hyplpdf_grad", self._h, ptr(hyp), len(hyp), ptr(out))
        return out


def setcovfs(om, covnames):
    """interfaceR.cpp:53-73"""
    covnames = [str(c) for c in covnames]
    for c in covnames:
        if c not in _KINDS:
            raise ValueError("need to choose one of the existing cov functions")
    if om._h:
        _lib.lib.obhip_model_destroy(om._h)
        om._h = None
    kinds = (C.c_int * len(covnames))(*[_KINDS[c] for c in covnames])
    h = C.c_void_p()
    call("obhip_model_create", C.byref(h), len(covnames), C.cast(kinds, C.c_void_p))
    om._h = h
    om.covnames = covnames
    om.d = len(covnames)


def setknot(om, knotlist):
    """interfaceR.cpp:94-149"""
    om._need()
    if len(knotlist) != om.d:
        raise ValueError("dim needs to match%d." % om.d)
    ks = [_f64(k) for k in knotlist]
    st = np.zeros(om.d + 1, dtype=np.uint64)
    st[1:] = np.cumsum([len(k) for k in ks])
    kp = np.concatenate(ks)
    call("obhip_model_set_knots", om._h, ptr(st), ptr(kp))
    om._knotptst = st.astype(np.int64)


def gethyp(om):
    """interfaceR.cpp:167-180 -> (values, names)"""
    om._need()
    nh = om.dims()[3]
    out = np.empty(nh)
    call("obhip_model_get_hyp", om._h, ptr(out))
    return out


def hypnames(om):
    names = []
    for l, c in enumerate(om.covnames):
        names += ["inpt%d.%s" % (l + 1, h) for h in _HYPNAMES[c]]
    return names


def getpara(logpdf):
    """interfaceR.cpp:193-199"""
    return np.array(logpdf.para, dtype=np.float64)


# ----------------------------------------------------------------------------
# terms handle cache
# ----------------------------------------------------------------------------
class _Terms:
    def __init__(self, om, terms):
        self.array = np.asarray(terms).astype(np.int64)
        t = _umat(terms)
        if t.shape[1] != om.d:
            raise ValueError("terms must have one column per input dimension")
        h = C.c_void_p()
        call("obhip_terms_create", C.byref(h), om._h, ptr(t), t.shape[0])
        self._h = h
        self.p = t.shape[0]
        self.d = t.shape[1]

    def maxlevels(self):
        out = np.empty(self.d, dtype=np.int64)
        call("obhip_terms_maxlevels", self._h, ptr(out))
        return out

    def info(self):
        p, d, nnz, mx = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        call("obhip_terms_info", self._h, C.byref(p), C.byref(d), C.byref(nnz), C.byref(mx))
        return dict(p=p.value, d=d.value, nnz_total=nnz.value, max_nnz=mx.value)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and _lib.lib is not None:
            _lib.lib.obhip_terms_destroy(self._h)
            self._h = None


def _terms_of(om, terms):
    return terms if isinstance(terms, _Terms) else _Terms(om, terms)


# ----------------------------------------------------------------------------
# outerbase (interfaceR.cpp:680-694)
# ----------------------------------------------------------------------------
class outerbase:
    def __init__(self, om, x, levelcap=None):
        om._need()
        self.om = om
        x = _fmat(x)
        if x.ndim != 2 or x.shape[1] != om.d:
            raise ValueError("x must be n x d")
        self.xp = x
        self.n_row = x.shape[0]
        cap = None if levelcap is None else np.ascontiguousarray(levelcap, dtype=np.int64)
        h = C.c_void_p()
        call("obhip_basis_create", C.byref(h), om._h, ptr(x), x.shape[0], x.shape[0], ptr(cap))
        self._h = h
        # OpenMP schedule fields of the reference (modandbase.cpp:504-513) have
        # no meaning on the device; kept readable for drop-in scripts.
        self.nthreads = 1
        self.vertpl = False

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and _lib.lib is not None:
            _lib.lib.obhip_basis_destroy(self._h)
            self._h = None

    def build(self):
        call("obhip_basis_rebuild", self._h)

    def getbase(self, k):
        m = int(np.diff(self.om._knotptst)[k - 1]) if 1 <= k <= self.om.d else 0
        if m == 0:
            raise ValueError("dimension out of range (1-based)")
        out = np.empty((self.n_row, m), order="F")
        call("obhip_basis_getbase", self._h, int(k), ptr(out))
        return out

    def getmat(self, terms):
        t = _terms_of(self.om, terms)
        out = np.empty((self.n_row, t.p), order="F")
        call("obhip_basis_getmat", self._h, t._h, ptr(out))
        return out

    def _mm(self, fn, terms, a, nin, nout):
        t = _terms_of(self.om, terms)
        a = _fmat(a)
        vec = a.ndim == 1
        a2 = a.reshape(-1, 1, order="F") if vec else a
        if a2.shape[0] != (t.p if nin == "p" else self.n_row):
            raise ValueError("non-conformable arguments")
        rows = self.n_row if nout == "n" else t.p
        out = np.empty((rows, a2.shape[1]), order="F")
        call(fn, self._h, t._h, ptr(a2), a2.shape[1], ptr(out))
        return out[:, 0].copy() if vec else out

    # -- hyper-parameter gradients (interfaceR.cpp:689-692) ------------------------------
    def getmat_gradhyp(self, terms):
        """n x p x nhyp cube (modandbase.cpp:663-669)."""
        t = _terms_of(self.om, terms)
        nh = len(self.om.grad_layout()[0])
        out = np.empty((self.n_row, t.p, nh), order="F")
        call("obhip_basis_getmat_gradhyp", self._h, t._h, ptr(out))
        return out

    def matmul_gradhyp(self, terms, a):
        """n x nhyp (mm_gradhyp_out, modandbase.cpp:739-744)."""
        t = _terms_of(self.om, terms)
        a = _f64(a)
        if a.shape[0] != t.p:
            raise ValueError("non-conformable arguments")
        nh = len(self.om.grad_layout()[0])
        out = np.empty((self.n_row, nh), order="F")
        call("obhip_basis_mm_gradhyp", self._h, t._h, ptr(a), None, ptr(out))
        return out

    def matmul_gradhyp_dot(self, terms, a, w):
        """w^T matmul_gradhyp(terms, a) (nhyp values) with the n x nhyp matrix left on the
        device -- the contraction the likelihoods make of it (loglik_gauss.cpp:127)."""
        t = _terms_of(self.om, terms)
        a, w = _f64(a), _f64(w)
        if a.shape[0] != t.p or w.shape[0] != self.n_row:
            raise ValueError("non-conformable arguments")
        out = np.empty(len(self.om.grad_layout()[0]))
        call("obhip_basis_mm_gradhyp_dot", self._h, t._h, ptr(a), ptr(w), None, ptr(out))
        return out

    def tmatmul_gradhyp(self, terms, a):
        """p x nhyp (tmm_gradhyp_out, modandbase.cpp:771-776)."""
        t = _terms_of(self.om, terms)
        a = _f64(a)
        if a.shape[0] != self.n_row:
            raise ValueError("non-conformable arguments")
        nh = len(self.om.grad_layout()[0])
        out = np.empty((t.p, nh), order="F")
        call("obhip_basis_tmm_gradhyp", self._h, t._h, ptr(a), None, ptr(out))
        return out

    def _sq_gradhyp(self, fn, terms, a, nin, nout):
        t = _terms_of(self.om, terms)
        a = _f64(a)
        if a.shape[0] != (t.p if nin == "p" else self.n_row):
            raise ValueError("non-conformable arguments")
        nh = len(self.om.grad_layout()[0])
        out = np.empty((self.n_row if nout == "n" else t.p, nh), order="F")
        call(fn, self._h, t._h, ptr(a), ptr(out))
        return out

    def sqmm_gradhyp(self, terms, a):       # modandbase.cpp:798-809
        return self._sq_gradhyp("obhip_basis_sqmm_gradhyp", terms, a, "p", "n")

    def sqtmm_gradhyp(self, terms, a):      # modandbase.cpp:845-856
        return self._sq_gradhyp("obhip_basis_sqtmm_gradhyp", terms, a, "n", "p")

    def sqcolsums_gradhyp(self, terms):     # modandbase.cpp:875-879
        t = _terms_of(self.om, terms)
        out = np.empty((t.p, len(self.om.grad_layout()[0])), order="F")
        call("obhip_basis_sqcolsums_gradhyp", self._h, t._h, ptr(out))
        return out

    def residvar_gradhyp(self, terms):      # modandbase.cpp:904-925
        t = _terms_of(self.om, terms)
        out = np.empty((self.n_row, len(self.om.grad_layout()[0])), order="F")
        call("obhip_basis_residvar_gradhyp", self._h, t._h, self.om._h, ptr(out))
        return out

    def matmul(self, terms, a):
        return self._mm("obhip_basis_mm", terms, a, "p", "n")

    def tmatmul(self, terms, a):
        return self._mm("obhip_basis_tmm", terms, a, "n", "p")

    def sqmm(self, terms, a):
        return self._mm("obhip_basis_sqmm", terms, a, "p", "n")

    def sqtmm(self, terms, a):
        return self._mm("obhip_basis_sqtmm", terms, a, "n", "p")

    def sqcolsums(self, terms):
        t = _terms_of(self.om, terms)
        out = np.empty(t.p)
        call("obhip_basis_sqcolsums", self._h, t._h, ptr(out))
        return out

    def residvar(self, terms):
        t = _terms_of(self.om, terms)
        out = np.empty(self.n_row)
        call("obhip_basis_residvar", self._h, t._h, self.om._h, ptr(out))
        return out


def rvar(y):
    y = np.asarray(y, dtype=np.float64)
    return float(np.sum((y - y.mean()) ** 2) / (len(y) - 1))


# ----------------------------------------------------------------------------
# lpdf family (interfaceR.cpp:696-762), value paths
# ----------------------------------------------------------------------------
class lpdf:
    def __init__(self):
        self.val = 0.0
        self.coeff = np.zeros(0)
        self.grad = np.zeros(0)
        self.para = np.zeros(0)
        self.nterms = 0
        self.fullhess = False
        self.compute_val = True
        self.compute_grad = True
        self.compute_gradhyp = False   # fit.h:36-38
        self.compute_gradpara = False
        self.gradhyp = np.zeros(0)
        self.gradpara = np.zeros(0)
        self.paranames = []

    def setnthreads(self, k):  # fit.h:57 (no-op on the device)
        return None

    def paralpdf(self, parap):
        parap = np.asarray(parap, dtype=np.float64)
        if len(parap) != len(self.para0):
            return -np.inf
        return float(-0.5 * np.sum((parap - self.para0) ** 2 / self.paravar))  # fit.cpp:133-139

    def paralpdf_grad(self, parap):
        parap = np.asarray(parap, dtype=np.float64)
        if len(parap) != len(self.para0):
            return np.zeros(len(self.para))
        return -(parap - self.para0) / self.paravar                            # fit.cpp:146-157

    def optcg(self, tol, maxepch):
        """lpdf::optcg (fit.cpp:37-96), generic form: diagonally preconditioned CG on
        update / hessmult / diaghess of this object (the n-passes run on the device through
        those methods, the p-vector algebra here).  lpdfvec overrides it with the
        device-resident loop when the likelihood has one noise level."""
        self.fullhess = False
        self.compute_gradhyp = self.compute_gradpara = False
        if len(self.coeff) != self.nterms:
            self.coeff = np.zeros(self.nterms)
        coeff = np.array(self.coeff, dtype=np.float64)
        self.update(coeff)
        m = self.diaghess()
        if not np.all(np.isfinite(m)) and not np.all(np.isfinite(self.grad)):
            self.val = -np.inf
            return
        rm = self.grad / m
        pv = rm.copy()
        q = self.hessmult(pv)
        valdiff = 10.0
        self.cgiters = 0
        num0 = None
        for _ in range(int(maxepch)):
            num = float(np.sum(self.grad * rm))
            if num < tol and valdiff < tol:
                break
            if num0 is None:
                num0 = num
            if num <= 1e-28 * num0:   # rounding floor, see obhip_fit_cg_dev
                break
            if not num > 0.0:     # exactly stationary: the next direction would be 0 / 0
                break
            denom = float(np.sum(q * pv))
            if not denom > 0.0:   # the direction cancelled to zero (p = 1, second iteration)
                break
            alpha = num / denom
            coeff = coeff + alpha * pv
            valo = self.val
            self.update(coeff)
            valdiff = self.val - valo
            rm = self.grad / m
            beta = -float(np.sum((alpha * q) * rm)) / num
            pv = rm + beta * pv
            q = self.hessmult(pv)
            self.cgiters += 1
        self.compute_gradhyp = self.compute_gradpara = True    # fit.cpp:87-93
        self.update(coeff)
        self.compute_gradhyp = self.compute_gradpara = False


class logpr_gauss(lpdf):
    """src/lpdfs/logpr_gauss.cpp:41-158"""

    def __init__(self, om, terms):
        super().__init__()
        self.om = om
        self.terms = np.asarray(terms).astype(np.int64)
        self.para0 = np.array([6.0])
        self.paravar = np.array([4.0])
        self.paranames = ["coeffscale"]
        self.para = self.para0.copy()
        self.nterms = self.terms.shape[0]
        self.updateom()

    def updateom(self):
        self.coeffsd = np.sqrt(self.om.getvar(self.terms))
        self.coefflvarge = self.om.getlvar_gradhyp(self.terms)   # logpr_gauss.cpp:80

    def updatepara(self, para):
        self.para = np.array(para, dtype=np.float64).reshape(-1)

    def updateterms(self, terms):
        self.terms = np.asarray(terms).astype(np.int64)
        self.nterms = self.terms.shape[0]
        self.updateom()

    def update(self, coeff):
        self.coeff = np.array(coeff, dtype=np.float64)
        sca = math.exp(self.para[0])
        stdresid = self.coeff / (self.coeffsd * sca)
        self.val = float(-0.5 * np.sum(stdresid ** 2) - np.sum(np.log(self.coeffsd * sca)))
        if self.compute_gradhyp:      # logpr_gauss.cpp:102
            self.gradhyp = (0.5 * self.coefflvarge).T @ (stdresid ** 2 - 1)
        if self.compute_gradpara:     # :103
            self.gradpara = np.array([np.sum(stdresid ** 2) - len(self.coeffsd)])
        self.grad = -1.0 * stdresid / (self.coeffsd * sca)

    def diaghess(self):
        return 1.0 / np.square(self.coeffsd * math.exp(self.para[0]))

    def diaghessgradhyp(self):        # logpr_gauss.cpp:131-135
        return -self.coefflvarge / np.square(self.coeffsd * math.exp(self.para[0]))[:, None]

    def diaghessgradpara(self):       # logpr_gauss.cpp:143-145
        return (-2.0 / np.square(self.coeffsd * math.exp(self.para[0])))[:, None]

    def hessmult(self, g):
        return np.asarray(g) / np.square(self.coeffsd * math.exp(self.para[0]))


class _loglik(lpdf):
    def __init__(self, om, terms, y, x):
        super().__init__()
        self.om = om
        self.y = _f64(y)
        self.x = _fmat(x)
        self._t = _Terms(om, terms)
        self.terms = self._t.array
        self.nterms = self._t.p
        self.para0 = np.array([math.log(0.01 * rvar(self.y))])  # loglik_std.cpp:51
        self.paravar = np.array([1.0])
        self.paranames = ["noisescale"]
        self.para = self.para0.copy()
        # the basis is evaluated up to the highest level the terms use
        self.ob = outerbase(om, self.x, levelcap=self._t.maxlevels())
        self.yhat = np.zeros(len(self.y))

    def updateom(self):
        self.ob.build()

    def updatepara(self, para):
        self.para = np.array(para, dtype=np.float64).reshape(-1)

    def updateterms(self, terms):
        self._t = _Terms(self.om, terms)
        self.terms = self._t.array
        self.nterms = self._t.p
        self.ob = outerbase(self.om, self.x, levelcap=self._t.maxlevels())

    def update(self, coeff):
        # loglik_gauss.cpp:110-130 / loglik_std.cpp:100-120
        self.coeff = np.array(coeff, dtype=np.float64)
        s = self.para[0]
        self.yhat = self.ob.matmul(self._t, self.coeff)
        resid = math.exp(-s) * (self.yhat - self.y)
        self.val = float(-0.5 * np.sum(resid ** 2) - len(self.y) * s)
        r2 = -math.exp(-s) * resid
        self.grad = self.ob.tmatmul(self._t, r2)
        if self.compute_gradhyp:      # loglik_gauss.cpp:114-117,127
            self.gradhyp = self.ob.matmul_gradhyp_dot(self._t, self.coeff, r2)
        if self.compute_gradpara:     # :128
            self.gradpara = np.array([np.sum(resid ** 2) - len(self.y)])

    def hessmult(self, g):
        v = self.ob.matmul(self._t, np.asarray(g, dtype=np.float64))
        return self.ob.tmatmul(self._t, math.exp(-2 * self.para[0]) * v)

    def diaghess(self):
        return math.exp(-2 * self.para[0]) * self.ob.sqcolsums(self._t)

    def diaghessgradhyp(self):        # loglik_gauss.cpp:158-161
        return math.exp(-2 * self.para[0]) * self.ob.sqcolsums_gradhyp(self._t)

    def diaghessgradpara(self):       # loglik_gauss.cpp:169-172
        return (-2 * math.exp(-2 * self.para[0]) * self.ob.sqcolsums(self._t))[:, None]


class loglik_gauss(_loglik):
    """src/lpdfs/loglik_gauss.cpp:41-157 (matrix-free)"""


class loglik_std(_loglik):
    """src/lpdfs/loglik_std.cpp:41-173.  The reference materialises the design
    matrix; here it stays factored and hess() runs the fused Gram kernel."""

    def hess(self):
        G = _gram_host(self.ob, self._t)
        return math.exp(-2 * self.para[0]) * G


class loglik_gda(_loglik):
    """src/lpdfs/loglik_gda.cpp:48-235: Gaussian likelihood whose per-observation variance
    adds the residual variance of the truncated expansion (the diagonal adjustment,
    field `dodiag`), composed from the device products (residvar, sqtmm, the *_gradhyp
    family); the n-vectors live on the host as in the Rcpp module."""

    def __init__(self, om, terms, y, x):
        super().__init__(om, terms, y, x)
        self.para0 = np.array([0.5 * math.log(0.01 * rvar(self.y)), 0.0])   # :58-60
        self.paravar = np.array([4.0, 4.0])
        self.paranames = ["noisescale", "lik.coeffscale"]
        self.para = self.para0.copy()
        self.dodiag = True
        self._redostd = True

    def updateom(self):
        super().updateom()
        self._redostd = True

    def updatepara(self, para):
        super().updatepara(para)
        self._redostd = True

    def updateterms(self, terms):
        super().updateterms(terms)
        self._redostd = True

    def _buildstd(self):              # :215-235
        if not self._redostd:
            return
        e0, e1 = math.exp(2 * self.para[0]), math.exp(2 * self.para[1])
        rterms = self.ob.residvar(self._t)
        obsvar = np.full(len(self.y), e0)
        if self.dodiag:
            obsvar = obsvar + e1 * rterms
        self.obssd = np.sqrt(obsvar)
        if self.dodiag:
            self.obssd_gradhyp = self.ob.residvar_gradhyp(self._t) * ((e1 * 0.5) / self.obssd)[:, None]
        self.obssd_gradpara = np.zeros((len(self.y), 2))
        self.obssd_gradpara[:, 0] = e0 / self.obssd
        if self.dodiag:
            self.obssd_gradpara[:, 1] = e1 * rterms / self.obssd
        self._redostd = False

    def update(self, coeff):          # :117-153
        self.coeff = np.array(coeff, dtype=np.float64)
        self.yhat = self.ob.matmul(self._t, self.coeff)
        self._buildstd()
        r = (self.yhat - self.y) / self.obssd
        r2 = np.square(r)
        self.val = float(-0.5 * np.sum(r2) - np.sum(np.log(self.obssd)))
        r = -r / self.obssd
        r2 = r2 / self.obssd
        self.grad = self.ob.tmatmul(self._t, r)
        if self.compute_gradhyp:
            self.gradhyp = self.ob.matmul_gradhyp_dot(self._t, self.coeff, r)
            if self.dodiag:
                self.gradhyp = self.gradhyp + r2 @ self.obssd_gradhyp \
                    - (1.0 / self.obssd) @ self.obssd_gradhyp
        if self.compute_gradpara:
            self.gradpara = r2 @ self.obssd_gradpara - (1.0 / self.obssd) @ self.obssd_gradpara

    def hessmult(self, g):            # :160-169
        v = self.ob.matmul(self._t, np.asarray(g, dtype=np.float64))
        return self.ob.tmatmul(self._t, v / np.square(self.obssd))

    def diaghess(self):               # :177-180
        self._buildstd()
        return self.ob.sqtmm(self._t, 1.0 / np.square(self.obssd))

    def diaghessgradhyp(self):        # :187-200
        self._buildstd()
        temp = 1.0 / np.square(self.obssd)
        lh = self.ob.sqtmm_gradhyp(self._t, temp)
        if self.dodiag:
            lh = lh + self.ob.sqtmm(self._t, self.obssd_gradhyp * (temp * (-2.0 / self.obssd))[:, None])
        return lh

    def diaghessgradpara(self):       # :207-214
        self._buildstd()
        temp = (1.0 / np.square(self.obssd)) * (-2.0 / self.obssd)
        return self.ob.sqtmm(self._t, self.obssd_gradpara * temp[:, None])


def _gram_host(ob, t):
    """B^T B through the device Gram kernel, returned to the host."""
    import torch
    if not torch.cuda.is_available():
        raise _lib.ObhipError(2, "no HIP device visible: libobhip has no CPU fallback")
    G = torch.empty((t.p, t.p), dtype=torch.float64, device="cuda")
    call("obhip_gram_dev", ob._h, t._h, None, ptr(G), None)
    torch.cuda.synchronize()
    return G.cpu().numpy()


class lpdfvec(lpdf):
    """src/fit.cpp:174-380 for a (likelihood, prior) pair; domarg switches the marginal
    adjustment on in its diagonal form (what obfit uses, R/fitting.R:110)."""

    def __init__(self, a, b):
        super().__init__()
        liks = [o for o in (a, b) if isinstance(o, _loglik)]
        prs = [o for o in (a, b) if isinstance(o, logpr_gauss)]
        if len(liks) != 1 or len(prs) != 1:
            raise ValueError("lpdfvec needs one likelihood and one logpr_gauss")
        self.lpdflist = [a, b]
        self.loglik, self.logpr = liks[0], prs[0]
        self.terms = self.loglik.terms
        self.nterms = self.loglik.nterms
        self.para = np.concatenate([a.para, b.para])
        self.para0 = np.concatenate([a.para0, b.para0])
        self.paravar = np.concatenate([a.paravar, b.paravar])
        self.paranames = a.paranames + b.paranames
        self.domarg = True            # fit.h:98 (field domargadj)
        self.coeff = np.zeros(self.nterms)
        self.totdiaghess = None
        self.tothess = None

    def _sigma_rho(self):
        return float(self.loglik.para[0]), float(self.logpr.para[0])

    def updateom(self):
        for o in self.lpdflist:
            o.updateom()

    def updatepara(self, para):
        para = np.array(para, dtype=np.float64).reshape(-1)
        n0 = len(self.lpdflist[0].para)
        self.lpdflist[0].updatepara(para[:n0])
        self.lpdflist[1].updatepara(para[n0:])
        self.para = para

    def updateterms(self, terms):
        for o in self.lpdflist:
            o.updateterms(terms)
        self.terms = self.loglik.terms
        self.nterms = self.loglik.nterms
        self.coeff = np.zeros(self.nterms)

    def update(self, coeff):
        self.coeff = np.array(coeff, dtype=np.float64)
        for o in self.lpdflist:       # fit.cpp:323-328: the flags are pushed down
            o.compute_gradhyp = self.compute_gradhyp
            o.compute_gradpara = self.compute_gradpara
            o.update(self.coeff)
        self.val = sum(o.val for o in self.lpdflist)
        self.grad = self.lpdflist[0].grad + self.lpdflist[1].grad
        if self.compute_gradhyp:      # fit.cpp:339-342,347-352: summed over the list
            self.gradhyp = self.lpdflist[0].gradhyp + self.lpdflist[1].gradhyp
        if self.compute_gradpara:     # concatenated like para
            self.gradpara = np.concatenate([self.lpdflist[0].gradpara, self.lpdflist[1].gradpara])
        if self.domarg:
            self._margadj()

    def _settotdiaghess(self, D):     # lpdfvec::settotdiaghess: the members see it too
        self.totdiaghess = D
        for o in self.lpdflist:
            o.totdiaghess = D

    def _margadj(self):
        """Marginal adjustment (lpdfvec::buildhess fit.cpp:252-299, margadj :371-380):
        -1/2 sum log diag(H) and its hyp / para gradients in the diagonal form; with the
        full Hessian (after optnewton) -1/2 log det H and -1/2 tr(inv(H) dH)."""
        if self.fullhess:
            # -1/2 log det H and -1/2 tr(inv(H) dH) on the device (obhip_margadj_full)
            lik, pr = self.loglik, self.logpr
            nh = len(lik.om.grad_layout()[0])
            H = _fmat(self.hess())
            val = C.c_double(0)
            gh, gp = np.zeros(nh), np.zeros(2)
            want_g = self.compute_gradhyp or self.compute_gradpara
            call("obhip_margadj_full", lik.ob._h, lik._t._h, lik.om._h, ptr(H), float(lik.para[0]),
                 float(pr.para[0]), C.byref(val), ptr(gh) if want_g else None,
                 ptr(gp) if want_g else None)
            self.val += val.value
            if self.compute_gradhyp:
                self.gradhyp = self.gradhyp + gh
            if self.compute_gradpara:
                order = [gp[0] if o is lik else gp[1] for o in self.lpdflist]
                self.gradpara = self.gradpara + np.array(order)
            return
        D = self.diaghess()
        self._settotdiaghess(D)
        self.val += float(-0.5 * np.sum(np.log(D)))
        if self.compute_gradhyp:
            dgh = self.lpdflist[0].diaghessgradhyp() + self.lpdflist[1].diaghessgradhyp()
            self.gradhyp = self.gradhyp - 0.5 * np.sum(dgh / D[:, None], axis=0)
        if self.compute_gradpara:
            dgp = np.concatenate([self.lpdflist[0].diaghessgradpara(),
                                  self.lpdflist[1].diaghessgradpara()], axis=1)
            self.gradpara = self.gradpara - 0.5 * np.sum(dgp / D[:, None], axis=0)

    def hessmult(self, g):
        return self.lpdflist[0].hessmult(g) + self.lpdflist[1].hessmult(g)

    def diaghess(self):
        return self.lpdflist[0].diaghess() + self.lpdflist[1].diaghess()

    def hess(self):
        H = self.loglik.hess()
        H[np.diag_indices_from(H)] += self.logpr.diaghess()
        return H

    def optnewton(self):
        """lpdf::optnewton (fit.cpp:98-131): Gram + Cholesky on the device."""
        if not isinstance(self.loglik, loglik_std):
            raise RuntimeError("optnewton needs a loglik_std (loglik_gauss never builds a Hessian)")
        self.fullhess = True
        sigma, rho = self._sigma_rho()
        p = self.nterms
        theta = np.zeros(p)
        diagH = np.zeros(p)
        call("obhip_fit_newton", self.loglik.ob._h, self.loglik._t._h, self.loglik.om._h,
             ptr(self.loglik.y), sigma, rho, ptr(theta), ptr(diagH), None)
        self._settotdiaghess(diagH)
        self.compute_gradhyp = self.compute_gradpara = True    # fit.cpp:122-128
        self.update(theta)
        self.compute_gradhyp = self.compute_gradpara = False

    def optcg(self, tol, maxepch):
        """lpdf::optcg (fit.cpp:37-96): matrix-free PCG, device-resident for a likelihood
        with one noise level; the generic loop of lpdf.optcg otherwise (loglik_gda)."""
        if isinstance(self.loglik, loglik_gda):
            return lpdf.optcg(self, tol, maxepch)
        self.fullhess = False
        sigma, rho = self._sigma_rho()
        p = self.nterms
        theta = np.array(self.coeff if len(self.coeff) == p else np.zeros(p), dtype=np.float64)
        diagH = np.zeros(p)
        iters = C.c_uint64(0)
        val = C.c_double(0)
        call("obhip_fit_cg", self.loglik.ob._h, self.loglik._t._h, self.loglik.om._h,
             ptr(self.loglik.y), sigma, rho, float(tol), int(maxepch), ptr(theta), C.byref(iters),
             ptr(diagH), C.byref(val))
        self._settotdiaghess(diagH)
        self.cgiters = iters.value
        self.compute_gradhyp = self.compute_gradpara = True    # fit.cpp:87-93
        self.update(theta)
        self.compute_gradhyp = self.compute_gradpara = False


class predictor:
    """interfaceR.cpp:725-731: the predictor that belongs to the likelihood (lpdf::pred):
    pred_gauss (loglik_gauss.cpp:196-227), predr_std (loglik_std.cpp:218-256) with the full
    posterior covariance, pred_gda (loglik_gda.cpp:247-281)."""

    def __init__(self, logpdf):
        lik = logpdf.loglik if isinstance(logpdf, lpdfvec) else logpdf
        if not isinstance(lik, _loglik):
            raise ValueError("cannot produce a predictor from this obj.")  # fit.h:53
        self.om = lik.om
        self._t = lik._t
        self._lik = lik
        self.coeff = np.array(lik.coeff if len(lik.coeff) == lik.nterms
                              else np.zeros(lik.nterms), dtype=np.float64)
        self.para = np.array(lik.para, dtype=np.float64)
        self.sigma = float(lik.para[0])
        # lpdfvec::settotdiaghess / settothess hand the Hessian pieces to the members
        td = getattr(logpdf, "totdiaghess", None)
        if td is None:
            td = getattr(lik, "totdiaghess", None)
        self.totdiaghess = None if td is None else np.asarray(td, dtype=np.float64)
        self.coeffvar = (1.0 / self.totdiaghess) if td is not None else np.zeros(lik.nterms)
        self.tothess = None
        if isinstance(lik, loglik_std) and isinstance(logpdf, lpdfvec) and logpdf.fullhess:
            self.tothess = _fmat(logpdf.hess())      # loglik_std.cpp:226-227 (didfulltothess)
        self.x = lik.x
        self._mean = None
        self._var = None

    def setnthreads(self, k):
        return None

    def update(self, x):
        x = _fmat(x)
        if x.ndim != 2 or x.shape[1] != self.om.d:
            raise ValueError("x must be n x d")
        n = x.shape[0]
        self._mean = np.empty(n)
        self._var = np.empty(n)
        lik = self._lik
        if isinstance(lik, loglik_std) and self.tothess is not None:
            # predr_std::var with coeffcov = inv(tothess), loglik_std.cpp:249-256
            call("obhip_predict_std", self.om._h, self._t._h, ptr(_f64(self.coeff)),
                 ptr(self.tothess), ptr(x), n, n, ptr(self._mean), self.sigma, ptr(self._var))
        else:
            # pred_gauss::var = B^2 (1 / totdiaghess) + e^{2 sigma} (loglik_gauss.cpp:224-225);
            # predr_std without a full Hessian puts totdiaghess ITSELF on the diagonal of
            # coeffcov (loglik_std.cpp:228-232) -- kept as the reference has it
            cv = self.coeffvar
            if isinstance(lik, loglik_std):
                cv = self.totdiaghess if self.totdiaghess is not None else np.zeros(lik.nterms)
            cv = _f64(cv)
            call("obhip_predict", self.om._h, self._t._h, ptr(_f64(self.coeff)), ptr(x), n, n,
                 ptr(self._mean), ptr(cv), self.sigma, ptr(self._var))
            if isinstance(lik, loglik_gda) and lik.dodiag:
                # pred_gda::var adds the residual variance of the truncated expansion
                # (loglik_gda.cpp:276-281)
                obn = outerbase(self.om, x, levelcap=self._t.maxlevels())
                self._var = self._var + math.exp(2 * self.para[1]) * obn.residvar(self._t)
        self.x = x

    def mean(self):
        if self._mean is None:
            self.update(self.x)
        return self._mean

    def var(self):
        if self._var is None:
            self.update(self.x)
        return self._var
