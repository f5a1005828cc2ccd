"""Host-side mirror of the reference's Rcpp module `obmod`
(src/interfaceR.cpp:661-793 of MattPlumlee/outerbase): same class, method and
field names, same argument meaning (0-based term levels, 1-based getbase), so
tests read like the reference's own testthat files.  Every method forwards to
the C ABI in include/obhip.h; nothing is computed in Python except O(p) vector
algebra on results.

Everything is a holder of a libobhip handle: outermod / outerbase / the lpdf family
(loglik_std, loglik_gauss, loglik_gda, logpr_gauss, lpdfvec) / predictor, value and
hyper-gradient paths and the marginal adjustment included.  The likelihood objects keep
their n-vectors in HBM (csrc/lpdf.cpp); a method call moves p-sized vectors only.
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

    def cov_gradhyp(self, x1, x2):
        """n1 x n2 x numhyp cube (covfuncs.cpp:134-150,220-243,318-347)"""
        x1, x2 = _f64(x1), _f64(x2)
        out = np.empty((len(x1), len(x2), len(self.hyp)), order="F")
        call("obhip_cov_gradhyp", _KINDS[self.kind], ptr(_f64(self.hyp)), ptr(x1), len(x1),
             ptr(x2), len(x2), ptr(out))
        return out

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

    def knots(self):
        """the knot points as set (list of one array per dimension)"""
        d, M, _, _ = self.dims()
        st = np.empty(d + 1, dtype=np.uint64)
        kp = np.empty(M)
        call("obhip_model_get_knots", self._h, ptr(st), ptr(kp))
        st = st.astype(np.int64)
        return [kp[st[l]:st[l + 1]].copy() for l in range(d)]

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
    def __init__(self, om, x, levelcap=None, _borrow=None):
        om._need()
        self.om = om
        self._owned = _borrow is None
        if _borrow is not None:      # the outerbase a likelihood owns (member `ob`)
            self._h, self.n_row = _borrow
            self.xp = None
            self.nthreads = 1
            self.vertpl = False
            return
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
        if getattr(self, "_h", None) and getattr(self, "_owned", False) and _lib is not None \
                and _lib.lib is not None:
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
# lpdf family (interfaceR.cpp:696-762): holders of an obhip_lpdf handle
# ----------------------------------------------------------------------------
_LOGLIK_STD, _LOGLIK_GAUSS, _LOGLIK_GDA, _LOGPR_GAUSS, _LPDF_VEC = range(5)
_FLAGS = {"compute_val": 0, "compute_grad": 1, "compute_gradhyp": 2, "compute_gradpara": 3,
          "fullhess": 4, "domarg": 5, "dodiag": 6}
_VECS = {"coeff": 0, "grad": 1, "gradhyp": 2, "gradpara": 3, "para": 4, "para0": 5, "paravar": 6,
         "totdiaghess": 7, "coeffsd": 8, "yhat": 9}


class lpdf:
    """class lpdf (fit.h:23-90, module rows interfaceR.cpp:696-723).  Fields are read from
    the object behind the handle on every access."""
    _h = None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and _lib.lib is not None:
            _lib.lib.obhip_lpdf_destroy(self._h)
            self._h = None

    # -- fields ---------------------------------------------------------------------------
    def __getattr__(self, name):
        if name in _FLAGS:
            v = C.c_int(0)
            call("obhip_lpdf_get_flag", self._h, _FLAGS[name], C.byref(v))
            return bool(v.value)
        if name in _VECS:
            n = C.c_uint64(0)
            call("obhip_lpdf_get_vec", self._h, _VECS[name], None, 0, C.byref(n))
            out = np.empty(n.value)
            call("obhip_lpdf_get_vec", self._h, _VECS[name], ptr(out), n.value, None)
            return out
        if name == "val":
            v = C.c_double(0)
            call("obhip_lpdf_get_val", self._h, C.byref(v))
            return v.value
        if name in ("nterms", "npara"):
            return self._dims()[name]
        if name == "paranames":
            out = []
            for i in range(self._dims()["npara"]):
                s = C.c_char_p()
                call("obhip_lpdf_paraname", self._h, i, C.byref(s))
                out.append(s.value.decode())
            return out
        if name == "terms":
            dd = self._dims()
            t = np.empty((dd["nterms"], self.om.d), dtype=np.uint64, order="F")
            call("obhip_lpdf_terms", self._h, ptr(t))
            return t.astype(np.int64)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _FLAGS:
            call("obhip_lpdf_set_flag", self._h, _FLAGS[name], int(bool(value)))
        else:
            object.__setattr__(self, name, value)

    def _dims(self):
        k, nt, npar, nh, n = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        call("obhip_lpdf_dims", self._h, C.byref(k), C.byref(nt), C.byref(npar), C.byref(nh),
             C.byref(n))
        return dict(kind=k.value, nterms=nt.value, npara=npar.value, nhyp=nh.value, n=n.value)

    # -- methods (interfaceR.cpp:710-722) ---------------------------------------------------
    def setnthreads(self, k):  # fit.h:57 (no meaning on the device)
        call("obhip_lpdf_setnthreads", self._h, int(k))

    def set_comm(self, comm):
        """Rows sharded over the ranks of `comm` (an obhip_comm handle, driver.make_comm):
        every sum over rows is summed over the ranks.  No reference counterpart."""
        call("obhip_lpdf_set_comm", self._h, comm)
        self._comm = comm

    def update(self, coeff):
        c = _f64(coeff)
        call("obhip_lpdf_update", self._h, ptr(c), len(c))

    def updateom(self):
        call("obhip_lpdf_updateom", self._h)

    def updatepara(self, para):
        pp = _f64(np.asarray(para, dtype=np.float64).reshape(-1))
        call("obhip_lpdf_updatepara", self._h, ptr(pp), len(pp))

    def updateterms(self, terms):
        t = _umat(terms)
        call("obhip_lpdf_updateterms", self._h, ptr(t), t.shape[0])

    def hessmult(self, g):
        g = _f64(g)
        if len(g) != self.nterms:
            raise ValueError("non-conformable arguments")
        out = np.empty(len(g))
        call("obhip_lpdf_hessmult", self._h, ptr(g), ptr(out))
        return out

    def diaghess(self):
        out = np.empty(self.nterms)
        call("obhip_lpdf_diaghess", self._h, ptr(out))
        return out

    def diaghessgradhyp(self):
        dd = self._dims()
        out = np.empty((dd["nterms"], dd["nhyp"]), order="F")
        call("obhip_lpdf_diaghessgradhyp", self._h, ptr(out))
        return out

    def diaghessgradpara(self):
        dd = self._dims()
        out = np.empty((dd["nterms"], dd["npara"]), order="F")
        call("obhip_lpdf_diaghessgradpara", self._h, ptr(out))
        return out

    def hess(self):
        p = self.nterms
        out = np.empty((p, p), order="F")
        call("obhip_lpdf_hess", self._h, ptr(out))
        return out

    def optcg(self, tol, maxepch):
        it = C.c_uint64(0)
        call("obhip_lpdf_optcg", self._h, float(tol), int(maxepch), C.byref(it))
        self.cgiters = it.value

    def optnewton(self):
        call("obhip_lpdf_optnewton", self._h)

    def paralpdf(self, parap):
        pp = _f64(np.asarray(parap, dtype=np.float64).reshape(-1))
        out = C.c_double(0)
        call("obhip_lpdf_paralpdf", self._h, ptr(pp), len(pp), C.byref(out))
        return out.value

    def paralpdf_grad(self, parap):
        pp = _f64(np.asarray(parap, dtype=np.float64).reshape(-1))
        out = np.zeros(self.npara)
        call("obhip_lpdf_paralpdf_grad", self._h, ptr(pp), len(pp), ptr(out))
        return out


class logpr_gauss(lpdf):
    """src/lpdfs/logpr_gauss.cpp:41-186 (interfaceR.cpp:752-756)"""

    def __init__(self, om, terms):
        om._need()
        self.om = om
        t = _umat(terms)
        if t.shape[1] != om.d:
            raise ValueError("terms must have one column per input dimension")
        h = C.c_void_p()
        call("obhip_logpr_gauss_create", C.byref(h), om._h, ptr(t), t.shape[0])
        self._h = h


class _loglik(lpdf):
    _kind = None

    def __init__(self, om, terms, y, x):
        om._need()
        self.om = om
        y = _f64(y)
        x = _fmat(x)
        t = _umat(terms)
        if x.ndim != 2 or x.shape[1] != om.d or x.shape[0] != len(y):
            raise ValueError("x must be n x d and y must have n entries")
        if t.shape[1] != om.d:
            raise ValueError("terms must have one column per input dimension")
        h = C.c_void_p()
        call("obhip_loglik_create", C.byref(h), self._kind, om._h, ptr(t), t.shape[0], ptr(y),
             ptr(x), x.shape[0], x.shape[0])
        self._h = h
        self.y = y
        self.x = x

    @property
    def ob(self):
        """the outerbase the likelihood owns (member `ob`, fit.h:185)"""
        b = C.c_void_p()
        call("obhip_lpdf_basis", self._h, C.byref(b), None)
        return outerbase(self.om, None, _borrow=(b, len(self.y)))

    @property
    def _t(self):
        return _Terms(self.om, self.terms)


class loglik_gauss(_loglik):
    """src/lpdfs/loglik_gauss.cpp:41-172 (matrix-free; interfaceR.cpp:739-743)"""
    _kind = _LOGLIK_GAUSS


class loglik_std(_loglik):
    """src/lpdfs/loglik_std.cpp:41-203 (interfaceR.cpp:733-737).  The reference materialises
    the design matrix; here it stays factored and hess() runs the Gram kernel."""
    _kind = _LOGLIK_STD


class loglik_gda(_loglik):
    """src/lpdfs/loglik_gda.cpp:48-235 (interfaceR.cpp:745-750): Gaussian likelihood whose
    per-observation variance adds the residual variance of the truncated expansion (field
    `dodiag`)."""
    _kind = _LOGLIK_GDA


class lpdfvec(lpdf):
    """src/fit.cpp:174-612 (interfaceR.cpp:758-762): a pair of lpdfs; field `domarg`."""

    def __init__(self, a, b):
        if not isinstance(a, lpdf) or not isinstance(b, lpdf):
            raise ValueError("lpdfvec needs two lpdf objects")
        h = C.c_void_p()
        call("obhip_lpdfvec_create", C.byref(h), a._h, b._h)
        self._h = h
        self.lpdflist = [a, b]          # keeps the members alive (fit.h:133 holds references)
        self.om = a.om
        liks = [o for o in (a, b) if isinstance(o, _loglik)]
        prs = [o for o in (a, b) if isinstance(o, logpr_gauss)]
        self.loglik = liks[0] if liks else None
        self.logpr = prs[0] if prs else None


class predictor:
    """interfaceR.cpp:725-731: the predictor that belongs to the likelihood (lpdf::pred):
    pred_gauss (loglik_gauss.cpp:196-227), predr_std (loglik_std.cpp:218-256) with the full
    posterior covariance after optnewton, pred_gda (loglik_gda.cpp:247-281)."""

    def __init__(self, logpdf):
        if not isinstance(logpdf, lpdf):
            raise ValueError("cannot produce a predictor from this obj.")
        h = C.c_void_p()
        call("obhip_predictor_create", C.byref(h), logpdf._h)      # fit.h:53 for non-likelihoods
        self._h = h
        self.om = logpdf.om

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and _lib.lib is not None:
            _lib.lib.obhip_predictor_destroy(self._h)
            self._h = None

    def setnthreads(self, k):
        call("obhip_predictor_setnthreads", self._h, int(k))

    def update(self, x):
        x = _fmat(x)
        if x.ndim != 2 or x.shape[1] != self.om.d:
            raise ValueError("x must be n x d")
        call("obhip_predictor_update", self._h, ptr(x), x.shape[0], x.shape[0])

    def _n(self):
        n = C.c_uint64(0)
        call("obhip_predictor_n", self._h, C.byref(n))
        return n.value

    def mean(self):
        out = np.empty(self._n())
        call("obhip_predictor_mean", self._h, ptr(out))
        return out

    def var(self):
        out = np.empty(self._n())
        call("obhip_predictor_var", self._h, ptr(out))
        return out
