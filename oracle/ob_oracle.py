"""
CPU oracle for the outerbase hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference algorithm (R package
MattPlumlee/outerbase v0.1.1).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; the product path
(outerbase_amd/) never does.

Every function cites the reference file:line it follows (paths relative to
the reference checkout).  The reference itself (C++11 + Armadillo + Rcpp)
cannot be built in this environment (no R, Rcpp, Armadillo), and it ships no
golden vectors, so this oracle is pinned by

  * the identities asserted by the reference's own tests
    (tests/testthat/test-obombasic.R:21-78, test-covf.R:1-65), restated in
    tests/test_oracle.py, and
  * mathematical invariants of the algorithm (see tests/test_oracle.py).

For `solve`, `inv`, the Gram, `optnewton`, `optcg` and `predictor` the
reference has no test at all: PARITY UNPINNED for those.

The eigen-decomposition (Armadillo `eig_sym` -> LAPACK dsyev*) is a
third-party dependency of the reference, not vendored and not version-pinned
(DESCRIPTION:16 "LinkingTo: Rcpp, RcppArmadillo"); numpy.linalg.eigh (LAPACK
dsyevd) restates it.
"""
import math
import numpy as np

# ----------------------------------------------------------------------------
# 1-D covariance kernels                                   src/covfuncs.cpp
# ----------------------------------------------------------------------------
COV_KINDS = ("mat25", "mat25pow", "mat25ang")
_A = 2.0      # covfuncs.h:42,53,66   `double a = 2.;`
_B = 0.25     # covfuncs.h:54         `double b = 0.25;`

# (numhyp, hyp0, hyplb, hypub, hypvar, lowbnd, uppbnd)
COV_INFO = {
    # covfuncs.cpp:87-111
    "mat25": dict(numhyp=1, hyp0=[0.0], hyplb=[-2.25], hypub=[1.5],
                  hypvar=[0.1], lowbnd=0.0, uppbnd=1.0,
                  hypnames=["scale"]),
    # covfuncs.cpp:166-195
    "mat25pow": dict(numhyp=2, hyp0=[0.0, 0.0], hyplb=[-2.25, -1.25],
                     hypub=[1.5, 1.25], hypvar=[0.1, 0.01], lowbnd=0.0,
                     uppbnd=1.0, hypnames=["scale", "power"]),
    # covfuncs.cpp:254-283
    "mat25ang": dict(numhyp=2, hyp0=[0.0, 0.0], hyplb=[-2.25, -2.25],
                     hypub=[1.5, 1.5], hypvar=[0.1, 0.1], lowbnd=0.0,
                     uppbnd=6.283185, hypnames=["sin.sc", "cos.sc"]),
}


def _mat25_of_h(h):
    # covfuncs.cpp:121-124  h = (1 + h + h^2/3) * exp(-h)
    return (1.0 + h + np.square(h) / 3.0) * np.exp(-h)


def cov(kind, x1, x2, hyp):
    """covf_*::cov -- n x m kernel matrix.

    mat25    covfuncs.cpp:113-126
    mat25pow covfuncs.cpp:197-212
    mat25ang covfuncs.cpp:285-310
    """
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    hyp = np.asarray(hyp, dtype=np.float64)
    if kind == "mat25":
        expLS = math.exp(_A * hyp[0])
        h = np.abs((x1 / expLS)[:, None] - (x2 / expLS)[None, :])
        return _mat25_of_h(h)
    if kind == "mat25pow":
        powv = math.exp(_B * hyp[1])
        expLS = math.exp(_A * hyp[0] + _B * hyp[1])
        x1t = np.power(x1, powv) / expLS
        x2t = np.power(x2, powv) / expLS
        h = np.abs(x1t[:, None] - x2t[None, :])
        return _mat25_of_h(h)
    if kind == "mat25ang":
        expLSs = math.exp(_A * hyp[0])
        expLSc = math.exp(_A * hyp[1])
        hs = (np.sin(x1) / expLSs)[:, None] - (np.sin(x2) / expLSs)[None, :]
        hc = (np.cos(x1) / expLSc)[:, None] - (np.cos(x2) / expLSc)[None, :]
        h = np.sqrt(np.square(hs) + np.square(hc))
        return _mat25_of_h(h)
    raise ValueError("unknown covariance " + str(kind))


def covmdiag(kind, x):
    """covf_*::covmdiag -- all ones (covfuncs.cpp:128-132,214-218,312-316)."""
    return np.ones(len(x))


def cov_gradhyp(kind, x1, x2, hyp):
    """covf_*::cov_gradhyp -- n x m x numhyp cube.

    covfuncs.cpp:134-150 (mat25), :220-243 (mat25pow), :318-347 (mat25ang).
    Checked against finite differences as tests/testthat/test-covf.R does.
    """
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    hyp = np.asarray(hyp, dtype=np.float64)
    if kind == "mat25":
        expLS = math.exp(_A * hyp[0])
        h = (x1 / expLS)[:, None] - (x2 / expLS)[None, :]
        h2 = h * (1 + np.abs(h)) * np.exp(-np.abs(h))
        return (_A / 3 * (h * h2))[:, :, None]
    if kind == "mat25pow":
        powv = math.exp(_B * hyp[1])
        expLS = math.exp(_A * hyp[0] + _B * hyp[1])
        x1t = np.power(x1, powv) / expLS
        x2t = np.power(x2, powv) / expLS
        h = x1t[:, None] - x2t[None, :]
        h2 = h * (1 + np.abs(h)) * np.exp(-np.abs(h))
        g1 = (np.log(x1) * x1t)[:, None] - (np.log(x2) * x2t)[None, :]
        g1 = g1 * (-(_B * powv / 3) * h2)
        hh = h * h2
        g1 = g1 + (_B / 3) * hh
        g0 = (_A / 3) * hh
        return np.stack([g0, g1], axis=2)
    if kind == "mat25ang":
        expLSs = math.exp(_A * hyp[0])
        expLSc = math.exp(_A * hyp[1])
        hs = (np.sin(x1) / expLSs)[:, None] - (np.sin(x2) / expLSs)[None, :]
        hc = (np.cos(x1) / expLSc)[:, None] - (np.cos(x2) / expLSc)[None, :]
        h = np.sqrt(np.square(hs) + np.square(hc))
        w = np.exp(-h) * (h + 1)
        return np.stack([(_A / 3) * np.square(hs) * w,
                         (_A / 3) * np.square(hc) * w], axis=2)
    raise ValueError("unknown covariance " + str(kind))


def cov_hyp_lpdf(kind, hypp):
    """covf::lpdf -- hyper-prior (covfuncs.cpp:35-50)."""
    info = COV_INFO[kind]
    hypp = np.asarray(hypp, dtype=np.float64)
    if len(hypp) != info["numhyp"]:
        return -np.inf
    out = 0.0
    for l in range(len(hypp)):
        if info["hypub"][l] < hypp[l] or info["hyplb"][l] > hypp[l]:
            return -np.inf
        out += 5 * math.log(info["hypub"][l] - hypp[l])
        out += 5 * math.log(hypp[l] - info["hyplb"][l])
    out -= 0.5 * float(np.sum(np.square(hypp - np.asarray(info["hyp0"]))
                              / np.asarray(info["hypvar"])))
    return out


def cov_hyp_lpdf_grad(kind, hypp):
    """covf::lpdf_gradhyp (covfuncs.cpp:53-70): zeros outside the box."""
    info = COV_INFO[kind]
    hypp = np.asarray(hypp, dtype=np.float64)
    out = np.zeros(info["numhyp"])
    if len(hypp) != info["numhyp"]:
        return out
    for l in range(len(hypp)):
        if info["hypub"][l] < hypp[l] or info["hyplb"][l] > hypp[l]:
            return np.zeros(info["numhyp"])
        out[l] -= 5 / (info["hypub"][l] - hypp[l])
        out[l] += 5 / (hypp[l] - info["hyplb"][l])
    return out - (hypp - np.asarray(info["hyp0"])) / np.asarray(info["hypvar"])


# ----------------------------------------------------------------------------
# outermod                                     src/modandbase.cpp:67-440
# ----------------------------------------------------------------------------
class OuterMod:
    """Restatement of class outermod (src/modandbase.h:9-54)."""

    def __init__(self):
        self.d = 0
        self.kinds = []
        self.hyp = np.zeros(0)
        self.hypst = np.zeros(1, dtype=np.int64)
        self.knotpt = np.zeros(0)
        self.knotptst = np.zeros(1, dtype=np.int64)
        self.rotmat = None
        self.basisvar = None
        self.maxlevel = None

    # interfaceR.cpp:53-73 (setcovfs) + modandbase.cpp:128-153 (hyp_init)
    def setcovfs(self, kinds):
        for k in kinds:
            if k not in COV_INFO:
                raise ValueError("unknown covariance " + str(k))
        self.kinds = list(kinds)
        self.d = len(kinds)
        st = [0]
        for k in kinds:
            st.append(st[-1] + COV_INFO[k]["numhyp"])
        self.hypst = np.asarray(st, dtype=np.int64)
        self.hyp = np.concatenate([COV_INFO[k]["hyp0"] for k in kinds]) \
            .astype(np.float64)
        self.rotmat = None

    # interfaceR.cpp:94-149 (setknot)
    def setknot(self, knotlist):
        if self.d == 0:
            raise RuntimeError("Need to set cov. funcs before setting knots.")
        if len(knotlist) != self.d:
            raise ValueError("dim needs to match %d." % self.d)
        for l, kn in enumerate(knotlist):
            kn = np.asarray(kn, dtype=np.float64)
            info = COV_INFO[self.kinds[l]]
            # covfuncs.h:23-27 inputcheck
            if kn.min() < info["lowbnd"] or kn.max() > info["uppbnd"]:
                raise ValueError("%d knot point needs to be between %f and %f"
                                 % (l + 1, info["lowbnd"], info["uppbnd"]))
        st = [0]
        for kn in knotlist:
            st.append(st[-1] + len(kn))
        self.knotptst = np.asarray(st, dtype=np.int64)
        self.knotpt = np.concatenate([np.asarray(k, dtype=np.float64)
                                      for k in knotlist])
        self.build()

    # modandbase.cpp:161-202 (hyp_set)
    def hyp_set(self, hyp):
        self.hyp = np.asarray(hyp, dtype=np.float64).copy()
        if len(self.knotpt):
            self.build()

    def hyp_of(self, k):
        return self.hyp[self.hypst[k]:self.hypst[k + 1]]

    def knots_of(self, k):
        return self.knotpt[self.knotptst[k]:self.knotptst[k + 1]]

    # modandbase.cpp:89-99 (hyplpdf)
    def hyplpdf(self, hypp):
        hypp = np.asarray(hypp, dtype=np.float64)
        if len(hypp) != len(self.hyp):
            return -np.inf
        return sum(cov_hyp_lpdf(self.kinds[l],
                                hypp[self.hypst[l]:self.hypst[l + 1]])
                   for l in range(self.d))

    # modandbase.cpp:106-118 (hyplpdf_grad)
    def hyplpdf_grad(self, hypp):
        hypp = np.asarray(hypp, dtype=np.float64)
        out = np.zeros(len(self.hyp))
        if len(hypp) == len(self.hyp):
            for l in range(self.d):
                sl = slice(self.hypst[l], self.hypst[l + 1])
                out[sl] = cov_hyp_lpdf_grad(self.kinds[l], hypp[sl])
        return out

    # modandbase.cpp:210-276 (build), value part :219-255
    def build(self):
        d = self.d
        M = len(self.knotpt)
        mmax = int(np.max(np.diff(self.knotptst)))
        self.rotmat = np.zeros((mmax, M))
        self.basisvar = np.zeros(M)
        self.maxlevel = np.zeros(d, dtype=np.int64)
        # gradient bookkeeping, hyp_set modandbase.cpp:183-197: per dimension one block
        # of m_l columns per hyper-parameter; gest[h] = first column of hyper-parameter
        # h, hypmatch[h] = its dimension
        nh = int(self.hypst[d])
        self.hypmatch = np.zeros(nh, dtype=np.int64)
        self.gest = np.zeros(nh + 1, dtype=np.int64)
        self.knotptstge = np.zeros(d + 1, dtype=np.int64)
        cur = 0
        for l in range(d):
            self.knotptstge[l] = cur
            for h in range(self.hypst[l], self.hypst[l + 1]):
                self.hypmatch[h] = l
                self.gest[h] = cur
                cur += self.knotptst[l + 1] - self.knotptst[l]
        self.knotptstge[d] = cur
        self.gest[nh] = cur
        self.rotmat_gradhyp = np.zeros((mmax, cur))
        self.logbasisvar_gradhyp = np.zeros(cur)
        for k in range(d):
            xsh = self.knots_of(k)
            lenh = len(xsh)
            R = cov(self.kinds[k], xsh, xsh, self.hyp_of(k))       # :232
            sr, U = np.linalg.eigh(R)                               # :236
            sr = sr[::-1].copy()                                    # :237
            U = U[:, ::-1].copy()                                   # :238
            halfw = lenh // 2                                       # :241
            U = U * np.sign(U[halfw, :] + 2.71828 * U[halfw + 1, :])[None, :]
            minsv = 0.00000000001 * np.mean(sr)                     # :245
            small = np.nonzero(-np.diff(sr) < minsv)[0]             # :246
            self.maxlevel[k] = small[0] if len(small) else lenh - 1  # :247-248
            sr = sr + np.linspace(minsv / 1000, lenh * minsv / 1000, lenh)
            o = self.knotptst[k]
            self.rotmat[:lenh, o:o + lenh] = \
                U / (sr / math.sqrt(lenh))[None, :]                 # :252-254
            self.basisvar[o:o + lenh] = np.log(sr / lenh)           # :255
            # gradient matrices, :257-274 (sr and U are the jittered / sign-fixed ones)
            Rge = cov_gradhyp(self.kinds[k], xsh, xsh, self.hyp_of(k))  # :258
            Fm = np.tile(sr[None, :], (lenh, 1))                    # :260
            np.fill_diagonal(Fm, 0.0)                               # :261
            Fm = Fm - sr[:, None]                                   # :262
            Fm = 1.0 / Fm                                           # :263
            og = self.knotptstge[k]
            for l in range(self.hypst[k + 1] - self.hypst[k]):
                UtdRV = U.T @ Rge[:, :, l] @ U                      # :267
                self.logbasisvar_gradhyp[og + l * lenh:og + (l + 1) * lenh] = \
                    np.diag(UtdRV) / sr                             # :268-269
                Ah = U @ (UtdRV * Fm)                               # :270
                Ah = Ah / (sr / math.sqrt(lenh))[None, :]           # :271
                self.rotmat_gradhyp[:lenh, og + l * lenh:og + (l + 1) * lenh] = Ah

    # modandbase.cpp:285-298 (buildob, value form)
    def buildob(self, xcol, k):
        lenh = self.knotptst[k + 1] - self.knotptst[k]
        o = self.knotptst[k]
        R = cov(self.kinds[k], xcol, self.knots_of(k), self.hyp_of(k))
        R = R @ self.rotmat[:lenh, o:o + lenh]
        R[:, 1:] = R[:, 1:] / R[:, 0:1]
        return R

    # modandbase.cpp:306-327 (buildob, gradient form): R as above and Rt[:, :, h] =
    # (d cov/d hyp_h . rotmat + cov . rotmat_gradhyp_h) / R[:, 0] for EVERY column
    # (column 0 included; it is not the derivative of the normalised ratio)
    def buildob_grad(self, xcol, k):
        lenh = self.knotptst[k + 1] - self.knotptst[k]
        o = self.knotptst[k]
        rot = self.rotmat[:lenh, o:o + lenh]
        R = cov(self.kinds[k], xcol, self.knots_of(k), self.hyp_of(k))      # :310
        Rt = cov_gradhyp(self.kinds[k], xcol, self.knots_of(k), self.hyp_of(k))  # :312
        for h in range(self.hypst[k], self.hypst[k + 1]):
            j = h - self.hypst[k]
            Rt[:, :, j] = Rt[:, :, j] @ rot + \
                R @ self.rotmat_gradhyp[:lenh, self.gest[h]:self.gest[h + 1]]  # :316-319
        R = R @ rot                                                           # :321
        Rt = Rt / R[:, 0:1, None]                                             # :324-325
        R[:, 1:] = R[:, 1:] / R[:, 0:1]                                       # :326
        return R, Rt

    # modandbase.cpp:364-379 (getlvar_gradhyp)
    def getlvar_gradhyp(self, terms):
        terms = np.asarray(terms, dtype=np.int64)
        out = np.zeros((terms.shape[0], len(self.hypmatch)))
        for h in range(len(self.hypmatch)):
            out[:, h] = self.logbasisvar_gradhyp[self.gest[h] + terms[:, self.hypmatch[h]]]
        return out

    # modandbase.cpp:336-342 (totvar)
    def totvar(self, x):
        return np.ones(np.asarray(x).shape[0])

    # modandbase.cpp:350-356 (getvar)
    def getvar(self, terms):
        terms = np.asarray(terms, dtype=np.int64)
        idx = self.knotptst[:self.d][None, :] + terms
        return np.exp(np.sum(self.basisvar[idx], axis=1))

    # modandbase.cpp:387-440 (selectterms)
    def selectterms(self, numele, rng=None):
        """Greedy best-first selection over the downward-closed lattice.

        The reference breaks near-ties (within 0.1 of the best open
        candidate) with Armadillo `shuffle`, i.e. R's RNG
        (modandbase.cpp:406-409), which cannot be reproduced outside R.
        `rng=None` replaces the shuffle by the identity permutation (lowest
        candidate index wins); a numpy Generator gives a random pick.
        """
        d = self.d
        st = self.knotptst[:d]
        terms = np.zeros((numele, d), dtype=np.int64)
        cap = 10 * numele + d + 1
        pterms = np.zeros((cap, d), dtype=np.int64)
        ptv = np.zeros(cap)
        ptv[0] = np.sum(self.basisvar[st + pterms[0]])
        npot = 1
        nd = 0
        for _ in range(numele):
            mval = -0.1 + np.max(ptv[:npot])                        # :406
            islarge = np.nonzero(ptv[:npot] > mval)[0]              # :407
            if rng is not None:
                kstar = int(rng.permutation(islarge)[0])            # :408
            else:
                kstar = int(islarge[0])
            terms[nd] = pterms[kstar]                               # :410
            nd += 1
            npot -= 1
            if npot > kstar:                                        # :414-417
                pterms[kstar] = pterms[npot]
                ptv[kstar] = ptv[npot]
            T = terms[nd - 1]
            pt = T[None, :] - terms[:nd]                            # :419
            gt = (np.sum(pt, axis=1) == 0) & \
                 (np.sum(np.abs(pt), axis=1) == 2)                  # :420-421
            for l in range(d):
                if T[l] < self.maxlevel[l]:                         # :423
                    gt2 = gt & (pt[:, l] == -1)                     # :425
                    h3 = int(np.sum(T > 0)) + int(T[l] < 1)         # :426
                    h4 = 1 + int(np.sum(gt2))                       # :427
                    if h3 == h4:
                        if npot >= cap:
                            # the reference would overrun its fixed
                            # 10*numele buffer here (Armadillo throws)
                            raise OverflowError("candidate list overflow")
                        pterms[npot] = T
                        pterms[npot, l] += 1
                        ptv[npot] = np.sum(self.basisvar[st + pterms[npot]])
                        npot += 1
        return terms


# ----------------------------------------------------------------------------
# outerbase                                   src/modandbase.cpp:459-922
# ----------------------------------------------------------------------------
class OuterBase:
    """Restatement of class outerbase (src/modandbase.h:57-125), value
    (non-gradient) parts."""

    def __init__(self, om, x, dograd=False):
        self.om = om
        self.xp = np.array(x, dtype=np.float64, order="F")
        self.dograd = dograd
        self.build()

    # modandbase.cpp:547-626 (build); the tall/short OpenMP branches compute
    # the same values, so one restatement covers both.
    def build(self):
        om = self.om
        n = self.xp.shape[0]
        M = len(om.knotpt)
        self.basemat = np.zeros((n, M), order="F")
        self.basescalemat = np.zeros((n, om.d), order="F")
        self.basescale = np.ones(n)
        if self.dograd:
            self.basemat_gradhyp = np.zeros((n, om.knotptstge[om.d]), order="F")
            self.basematsq_gradhyp = np.zeros((n, om.knotptstge[om.d]), order="F")
        for k in range(om.d):
            if self.dograd:
                R, Rt = om.buildob_grad(self.xp[:, k], k)           # :568
                for h in range(om.hypst[k], om.hypst[k + 1]):
                    self.basemat_gradhyp[:, om.gest[h]:om.gest[h + 1]] = \
                        Rt[:, :, h - om.hypst[k]]                   # :585-587
            else:
                R = om.buildob(self.xp[:, k], k)
            self.basescalemat[:, k] = R[:, 0]                       # :572
            self.basescale *= R[:, 0]                               # :573
            R[:, 0] = 1.0                                           # :574
            if self.dograd:
                for h in range(om.hypst[k], om.hypst[k + 1]):       # :588-590, R[:, 0] is 1 here
                    self.basematsq_gradhyp[:, om.gest[h]:om.gest[h + 1]] = \
                        2 * (Rt[:, :, h - om.hypst[k]] * R)
            self.basemat[:, om.knotptst[k]:om.knotptst[k + 1]] = R  # :578
        self.basematsq = np.square(self.basemat)                    # :581
        self.basescalesq = np.square(self.basescale)                # :597

    # modandbase.cpp:634-639 (getbase; 1-based dim like the reference)
    def getbase(self, dim):
        k = dim - 1
        om = self.om
        out = self.basemat[:, om.knotptst[k]:om.knotptst[k + 1]].copy()
        return out * self.basescalemat[:, k:k + 1]


def _colprod(terms, knotptst, basemat, k):
    """temp = prod over l with terms[k,l]>0 of basemat[:, off_l+terms[k,l]]
    (linalg.cpp:73-76, 295-298, 659-661)."""
    temp = np.ones(basemat.shape[0])
    for l in range(terms.shape[1]):
        t = terms[k, l]
        if t > 0:
            temp = temp * basemat[:, knotptst[l] + t]
    return temp


def getm(terms, basemat, basescale, knotptst):
    """getm_/domat_ (linalg.cpp:647-715): materialise B (n x p)."""
    terms = np.asarray(terms, dtype=np.int64)
    out = np.empty((basemat.shape[0], terms.shape[0]), order="F")
    for k in range(terms.shape[0]):
        out[:, k] = _colprod(terms, knotptst, basemat, k)
    return out * basescale[:, None]


def prodmm(terms, a, basemat, basescale, knotptst):
    """prodmm_/domult_ (linalg.cpp:57-131; matrix form :481-557): B.a."""
    terms = np.asarray(terms, dtype=np.int64)
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        out = np.zeros(basemat.shape[0])
        for k in range(terms.shape[0]):
            out += a[k] * _colprod(terms, knotptst, basemat, k)
        return out * basescale
    out = np.zeros((basemat.shape[0], a.shape[1]))
    for k in range(terms.shape[0]):
        out += _colprod(terms, knotptst, basemat, k)[:, None] * a[k][None, :]
    return out * basescale[:, None]


def tprodmm(terms, a, basemat, basescale, knotptst):
    """tprodmm_/dotmultsub_ (linalg.cpp:286-355; matrix form :567-637):
    B^T.a."""
    terms = np.asarray(terms, dtype=np.int64)
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        b = basescale * a
        out = np.zeros(terms.shape[0])
        for k in range(terms.shape[0]):
            out[k] = np.sum(b * _colprod(terms, knotptst, basemat, k))
        return out
    b = a * basescale[:, None]
    out = np.zeros((terms.shape[0], a.shape[1]))
    for k in range(terms.shape[0]):
        out[k] = _colprod(terms, knotptst, basemat, k) @ b
    return out


def _colprod_ge(terms, knotptst, basemat, basematge, gest, hypmatch, k, h):
    """One term's column of dB/d hyp_h without basescale: the product of
    _colprod with dimension hypmatch[h]'s factor replaced by the gradient column
    gest[h] + level -- level 0 included (dotmultgesub_, linalg.cpp:375-384; in
    domultgesub_ :150-161 the level-0 case is spread over `- temp % ge[0]` per term and
    `+ ge[0] % out` at the end, prodmmge_ :271-272, which sums to the same)."""
    lm = hypmatch[h]
    temp = np.ones(basemat.shape[0])
    for m in range(terms.shape[1]):
        t = terms[k, m]
        if t > 0 and m != lm:
            temp = temp * basemat[:, knotptst[m] + t]
    return temp * basematge[:, gest[h] + terms[k, lm]]


def getmge(terms, basemat, basescale, knotptst, basematge, gest, hypmatch):
    """getmge_ (linalg.cpp:778-822, the non-chunked branch :812-817; its chunked
    branch :788-810 assigns a zero buffer and cannot be what is meant): the cube
    dB/d hyp, n x p x nhyp."""
    terms = np.asarray(terms, dtype=np.int64)
    nh = len(hypmatch)
    out = np.empty((basemat.shape[0], terms.shape[0], nh))
    for h in range(nh):
        for k in range(terms.shape[0]):
            out[:, k, h] = _colprod_ge(terms, knotptst, basemat, basematge, gest, hypmatch, k, h)
    return out * basescale[:, None, None]


def prodmmge(terms, a, basemat, basescale, knotptst, basematge, gest, hypmatch):
    """prodmmge_ (linalg.cpp:219-276): B.a and its gradient, n x nhyp."""
    terms = np.asarray(terms, dtype=np.int64)
    a = np.asarray(a, dtype=np.float64)
    nh = len(hypmatch)
    out = np.zeros(basemat.shape[0])
    outge = np.zeros((basemat.shape[0], nh))
    for k in range(terms.shape[0]):
        temp = a[k] * _colprod(terms, knotptst, basemat, k)          # :148-152
        out += temp                                                  # :153
        for h in range(nh):                                          # :154-163
            if terms[k, hypmatch[h]] > 0:
                tempalt = a[k] * _colprod_ge(terms, knotptst, basemat, basematge, gest,
                                             hypmatch, k, h)
                outge[:, h] += tempalt - temp * basematge[:, gest[h]]
    for h in range(nh):
        outge[:, h] += basematge[:, gest[h]] * out                   # :271-272
    return out * basescale, outge * basescale[:, None]               # :273-274


def tprodmmge(terms, a, basemat, basescale, knotptst, basematge, gest, hypmatch):
    """tprodmmge_ (linalg.cpp:395-471): B^T.a and its gradient, p x nhyp."""
    terms = np.asarray(terms, dtype=np.int64)
    b = basescale * np.asarray(a, dtype=np.float64)                  # :410
    nh = len(hypmatch)
    out = np.zeros(terms.shape[0])
    outge = np.zeros((terms.shape[0], nh))
    for k in range(terms.shape[0]):
        out[k] = np.sum(b * _colprod(terms, knotptst, basemat, k))   # :371-375
        for h in range(nh):                                          # :377-385
            outge[k, h] = np.dot(b, _colprod_ge(terms, knotptst, basemat, basematge, gest,
                                                hypmatch, k, h))
    return out, outge


def ob_getmat_gradhyp(ob, terms):          # modandbase.cpp:663-669
    om = ob.om
    return getmge(terms, ob.basemat, ob.basescale, om.knotptst, ob.basemat_gradhyp,
                  om.gest, om.hypmatch)


def ob_mm_gradhyp(ob, terms, a):           # modandbase.cpp:725-744
    om = ob.om
    return prodmmge(terms, a, ob.basemat, ob.basescale, om.knotptst, ob.basemat_gradhyp,
                    om.gest, om.hypmatch)


def ob_tmm_gradhyp(ob, terms, a):          # modandbase.cpp:755-776
    om = ob.om
    return tprodmmge(terms, a, ob.basemat, ob.basescale, om.knotptst, ob.basemat_gradhyp,
                     om.gest, om.hypmatch)


def ob_sqmm_gradhyp(ob, terms, a):         # modandbase.cpp:798-809
    om = ob.om
    return prodmmge(terms, a, ob.basematsq, ob.basescalesq, om.knotptst,
                    ob.basematsq_gradhyp, om.gest, om.hypmatch)[1]


def ob_sqtmm_gradhyp(ob, terms, a):        # modandbase.cpp:845-856
    om = ob.om
    return tprodmmge(terms, a, ob.basematsq, ob.basescalesq, om.knotptst,
                     ob.basematsq_gradhyp, om.gest, om.hypmatch)[1]


def ob_sqcolsums_gradhyp(ob, terms):       # modandbase.cpp:875-879
    return ob_sqtmm_gradhyp(ob, terms, np.ones(ob.xp.shape[0]))


def ob_residvar_gradhyp(ob, terms):        # modandbase.cpp:904-925
    varc = ob.om.getvar(terms)
    outge = -ob_sqmm_gradhyp(ob, terms, varc)
    lv2 = ob.om.getlvar_gradhyp(terms) * varc[:, None]
    return outge - prodmm(terms, lv2, ob.basematsq, ob.basescalesq, ob.om.knotptst)


def ob_getmat(ob, terms):      # modandbase.cpp:649-654
    return getm(terms, ob.basemat, ob.basescale, ob.om.knotptst)


def ob_mm(ob, terms, a):       # modandbase.cpp:677-692
    return prodmm(terms, a, ob.basemat, ob.basescale, ob.om.knotptst)


def ob_tmm(ob, terms, a):      # modandbase.cpp:700-716
    return tprodmm(terms, a, ob.basemat, ob.basescale, ob.om.knotptst)


def ob_sqmm(ob, terms, a):     # modandbase.cpp:784-790
    return prodmm(terms, a, ob.basematsq, ob.basescalesq, ob.om.knotptst)


def ob_sqtmm(ob, terms, a):    # modandbase.cpp:816-837
    return tprodmm(terms, a, ob.basematsq, ob.basescalesq, ob.om.knotptst)


def ob_sqcolsums(ob, terms):   # modandbase.cpp:863-867
    return ob_sqtmm(ob, terms, np.ones(ob.xp.shape[0]))


def ob_residvar(ob, terms):    # modandbase.cpp:889-895
    return 1.0 - ob_sqmm(ob, terms, ob.om.getvar(terms))


# ----------------------------------------------------------------------------
# Gram / Newton / predict slice      src/lpdfs/loglik_std.cpp, logpr_gauss.cpp
# ----------------------------------------------------------------------------
def rvar(y):
    """R's var(): n-1 denominator (arma::var default too,
    loglik_std.cpp:51)."""
    y = np.asarray(y, dtype=np.float64)
    return float(np.sum(np.square(y - np.mean(y))) / (len(y) - 1))


def default_sigma(y):
    """loglik_std.cpp:51 / loglik_gauss.cpp:46: para0 = log(0.01*var(y))."""
    return math.log(0.01 * rvar(y))


DEFAULT_RHO = 6.0   # logpr_gauss.cpp:48


def prior_prec(om, terms, rho):
    """logpr_gauss::diaghess (logpr_gauss.cpp:122-124):
    1/(coeffsd*sca)^2, coeffsd = sqrt(getvar(terms)) (:55), sca=exp(rho)
    (:54)."""
    coeffsd = np.sqrt(om.getvar(terms))
    return 1.0 / np.square(coeffsd * math.exp(rho))


def gram(ob, terms, y=None):
    """Unscaled G = B^T B (loglik_std.cpp:170-173 without the e^{-2 sigma})
    and g = B^T y (loglik_std.cpp:114-116 at coeff = 0)."""
    B = ob_getmat(ob, terms)
    G = B.T @ B
    g = None if y is None else B.T @ np.asarray(y, dtype=np.float64)
    return G, g


def total_hess(ob, terms, sigma, rho):
    """lpdfvec::hess_ (fit.cpp:503-512) = loglik_std::hess
    (loglik_std.cpp:170-173) + logpr_gauss::hess (logpr_gauss.cpp:153-158)."""
    G, _ = gram(ob, terms)
    H = math.exp(-2 * sigma) * G
    H[np.diag_indices_from(H)] += prior_prec(ob.om, terms, rho)
    return H


def total_grad(ob, terms, y, sigma, rho, coeff):
    """lpdfvec::update gradient sum (fit.cpp:323-363):
    loglik_std::update (loglik_std.cpp:100-120) +
    logpr_gauss::update (logpr_gauss.cpp:98-106)."""
    yhat = ob_mm(ob, terms, coeff)
    resid = -math.exp(-sigma) * (math.exp(-sigma) * (yhat - y))
    g = ob_tmm(ob, terms, resid)
    coeffsd = np.sqrt(ob.om.getvar(terms))
    sca = math.exp(rho)
    stdresid = coeff / (coeffsd * sca)
    return g - stdresid / (coeffsd * sca)


def fit_newton(ob, terms, y, sigma=None, rho=DEFAULT_RHO, coeff0=None):
    """lpdf::optnewton on lpdfvec(loglik_std, logpr_gauss)
    (fit.cpp:98-131): one Newton step coeff += solve(H, grad)."""
    y = np.asarray(y, dtype=np.float64)
    if sigma is None:
        sigma = default_sigma(y)
    p = np.asarray(terms).shape[0]
    coeff = np.zeros(p) if coeff0 is None else np.array(coeff0, float)
    H = total_hess(ob, terms, sigma, rho)
    r = total_grad(ob, terms, y, sigma, rho, coeff)
    coeff = coeff + np.linalg.solve(H, r)                      # fit.cpp:120
    return coeff, H


def lpdf_val(ob, terms, y, sigma, rho, coeff):
    """val of lpdfvec without marginal adjustment: loglik_gauss/std val
    (loglik_gauss.cpp:121, loglik_std.cpp:111) + logpr_gauss val
    (logpr_gauss.cpp:101)."""
    yhat = ob_mm(ob, terms, coeff)
    n = len(y)
    v = -0.5 * np.sum(np.square(math.exp(-sigma) * (yhat - y))) - n * sigma
    coeffsd = np.sqrt(ob.om.getvar(terms))
    sca = math.exp(rho)
    v += -0.5 * np.sum(np.square(coeff / (coeffsd * sca))) \
        - np.sum(np.log(coeffsd * sca))
    return float(v)


def loglik_update(ob, terms, y, sigma, coeff):
    """loglik_gauss::update / loglik_std::update with compute_gradhyp and
    compute_gradpara (loglik_gauss.cpp:110-130, loglik_std.cpp:100-120): val, grad,
    gradhyp, gradpara.  ob must have been built with dograd."""
    yhat, yhatge = ob_mm_gradhyp(ob, terms, coeff)
    resid = math.exp(-sigma) * (yhat - y)
    val = float(-0.5 * np.sum(np.square(resid)) - len(y) * sigma)
    r2 = -math.exp(-sigma) * resid
    grad = ob_tmm(ob, terms, r2)
    gradhyp = r2 @ yhatge
    gradpara = np.array([np.sum(np.square(resid)) - len(y)])
    return val, grad, gradhyp, gradpara


def logpr_update(om, terms, rho, coeff):
    """logpr_gauss::update with compute_gradhyp / compute_gradpara
    (logpr_gauss.cpp:98-108; coefflvarge = getlvar_gradhyp, :80)."""
    coeffsd = np.sqrt(om.getvar(terms))
    sca = math.exp(rho)
    stdresid = coeff / (coeffsd * sca)
    val = float(-0.5 * np.sum(np.square(stdresid)) - np.sum(np.log(coeffsd * sca)))
    gradhyp = (0.5 * om.getlvar_gradhyp(terms)).T @ (np.square(stdresid) - 1)
    gradpara = np.array([np.sum(np.square(stdresid)) - len(coeffsd)])
    grad = -stdresid / (coeffsd * sca)
    return val, grad, gradhyp, gradpara


def loglik_gda_update(ob, terms, y, para, coeff, dodiag=True):
    """loglik_gda::buildstd + update + diaghess family (loglik_gda.cpp:117-235): returns a
    dict with val, grad, gradhyp, gradpara, obssd, diaghess, diaghessgradhyp,
    diaghessgradpara.  ob needs dograd."""
    e0, e1 = math.exp(2 * para[0]), math.exp(2 * para[1])
    rterms = ob_residvar(ob, terms)                                   # :221
    obsvar = np.full(len(y), e0) + (e1 * rterms if dodiag else 0.0)   # :219-222
    obssd = np.sqrt(obsvar)
    n, nh = len(y), len(ob.om.hypmatch)
    ge = ob_residvar_gradhyp(ob, terms) * ((e1 * 0.5) / obssd)[:, None] if dodiag \
        else np.zeros((n, nh))                                        # :224-227
    gp = np.zeros((n, 2))
    gp[:, 0] = e0 / obssd                                             # :229
    if dodiag:
        gp[:, 1] = e1 * rterms / obssd                                # :230
    yhat, yhatge = ob_mm_gradhyp(ob, terms, coeff)                    # :122-125
    r = (yhat - y) / obssd
    r2 = np.square(r)
    out = {"val": float(-0.5 * np.sum(r2) - np.sum(np.log(obssd))), "obssd": obssd}
    r = -r / obssd
    r2 = r2 / obssd
    out["grad"] = ob_tmm(ob, terms, r)
    out["gradhyp"] = r @ yhatge + (r2 @ ge - (1 / obssd) @ ge if dodiag else 0.0)   # :139-145
    out["gradpara"] = r2 @ gp - (1 / obssd) @ gp                                     # :147-150
    w = 1 / np.square(obssd)
    out["diaghess"] = ob_sqtmm(ob, terms, w)                                         # :177-180
    out["diaghessgradhyp"] = ob_sqtmm_gradhyp(ob, terms, w) + \
        (ob_sqtmm(ob, terms, ge * (w * (-2 / obssd))[:, None]) if dodiag else 0.0)  # :187-200
    out["diaghessgradpara"] = ob_sqtmm(ob, terms, gp * (w * (-2 / obssd))[:, None])  # :207-214
    return out


def margadj_diag(ob, terms, sigma, rho):
    """Marginal adjustment of lpdfvec in its diagonal form (lpdfvec::buildhess,
    fit.cpp:252-268, added by margadj :371-380): with the total Hessian diagonal
    D = e^{-2 sigma} sqcolsums + 1/(sd e^rho)^2,
      val += -1/2 sum log D,  gradhyp += -1/2 sum_k dD_k/dhyp / D_k,  same for para.
    Returns (val, gradhyp, gradpara_loglik, gradpara_logpr); ob needs dograd."""
    om = ob.om
    prec = prior_prec(om, terms, rho)
    D = math.exp(-2 * sigma) * ob_sqcolsums(ob, terms) + prec
    dgh = math.exp(-2 * sigma) * ob_sqcolsums_gradhyp(ob, terms) \
        - om.getlvar_gradhyp(terms) * prec[:, None]         # loglik_gauss.cpp:158-161, logpr :131-135
    dgp_lik = -2 * math.exp(-2 * sigma) * ob_sqcolsums(ob, terms)   # loglik_gauss.cpp:169-172
    dgp_pr = -2 * prec                                              # logpr_gauss.cpp:143-145
    return (float(-0.5 * np.sum(np.log(D))), -0.5 * np.sum(dgh / D[:, None], axis=0),
            float(-0.5 * np.sum(dgp_lik / D)), float(-0.5 * np.sum(dgp_pr / D)))


def margadj_full(ob, terms, sigma, rho):
    """Marginal adjustment of lpdfvec with the full Hessian (lpdfvec::buildhess,
    fit.cpp:270-299; loglik_std::hessgradhyp / hessgradpara loglik_std.cpp:180-203,
    logpr_gauss::hessgradhyp / hessgradpara logpr_gauss.cpp:165-186):
      val = -1/2 log det H,  gradhyp[l] = -1/2 sum(dH/dhyp_l % inv(H)),  same for para.
    Returns (val, gradhyp, gradpara_loglik, gradpara_logpr); ob needs dograd."""
    om = ob.om
    B = ob_getmat(ob, terms)
    Bge = ob_getmat_gradhyp(ob, terms)
    prec = prior_prec(om, terms, rho)
    e2 = math.exp(-2 * sigma)
    H = e2 * (B.T @ B) + np.diag(prec)
    heig, hvec = np.linalg.eigh(H)                       # :277-281
    Hinv = (hvec / heig[None, :]) @ hvec.T
    val = float(-0.5 * np.sum(np.log(heig)))             # :283
    lv = om.getlvar_gradhyp(terms)
    gh = np.zeros(Bge.shape[2])
    for l in range(Bge.shape[2]):
        S = e2 * (B.T @ Bge[:, :, l])
        dH = S + S.T - np.diag(lv[:, l] * prec)          # loglik_std.cpp:185-189, logpr :168-171
        gh[l] = -0.5 * np.sum(dH * Hinv)                 # fit.cpp:286-290
    gp_lik = float(-0.5 * np.sum((-2 * e2 * (B.T @ B)) * Hinv))       # loglik_std.cpp:199-203
    gp_pr = float(-0.5 * np.sum(-2 * prec * np.diag(Hinv)))           # logpr_gauss.cpp:181-186
    return val, gh, gp_lik, gp_pr


def fit_cg(ob, terms, y, sigma=None, rho=DEFAULT_RHO, tol=1e-10, maxit=100,
           coeff0=None):
    """lpdf::optcg on lpdfvec(logpr_gauss, loglik_gauss) (fit.cpp:37-96) with
    loglik_gauss::{update,hessmult,diaghess} (loglik_gauss.cpp:110-157) and
    domargadj = false.  Returns (coeff, iterations, diaghess)."""
    y = np.asarray(y, dtype=np.float64)
    if sigma is None:
        sigma = default_sigma(y)
    p = np.asarray(terms).shape[0]
    coeff = np.zeros(p) if coeff0 is None else np.array(coeff0, float)
    prec = prior_prec(ob.om, terms, rho)
    e2 = math.exp(-2 * sigma)

    def update(c):
        return (lpdf_val(ob, terms, y, sigma, rho, c),
                total_grad(ob, terms, y, sigma, rho, c))

    def hessmult(v):
        return e2 * ob_tmm(ob, terms, ob_mm(ob, terms, v)) + prec * v

    val, grad = update(coeff)
    m = e2 * ob_sqcolsums(ob, terms) + prec                    # fit.cpp:51
    rm = grad / m
    pvec = rm.copy()
    q = hessmult(pvec)
    valdiff = 10.0
    k = 0
    for k in range(maxit):                                      # fit.cpp:71
        num = float(np.sum(grad * rm))
        if num < tol and valdiff < tol:
            break
        denom = float(np.sum(q * pvec))
        alpha = num / denom
        coeff = coeff + alpha * pvec
        valo = val
        val, grad = update(coeff)
        valdiff = val - valo
        rm = grad / m
        num2 = -float(np.sum((alpha * q) * rm))
        beta = num2 / num
        pvec = rm + beta * pvec
        q = hessmult(pvec)
    else:
        k = maxit
    return coeff, k, m


def predict_mean(om, terms, coeff, xnew):
    """pred_gauss/predr_std::update + mean (loglik_gauss.cpp:214-222,
    loglik_std.cpp:239-248)."""
    ob = OuterBase(om, xnew)
    return ob_mm(ob, terms, coeff)


def predict_var_std(om, terms, H, sigma, xnew):
    """predr_std::var (loglik_std.cpp:249-256): rowsum((B H^-1) .* B) +
    e^{2 sigma}, coeffcov = inv(tothess) (:227)."""
    ob = OuterBase(om, xnew)
    B = ob_getmat(ob, terms)
    return np.sum((B @ np.linalg.inv(H)) * B, axis=1) + math.exp(2 * sigma)


def predict_var_gauss(om, terms, diagH, sigma, xnew):
    """pred_gauss::var (loglik_gauss.cpp:223-227): B^2 (1/diagH) +
    e^{2 sigma}."""
    ob = OuterBase(om, xnew)
    return ob_sqmm(ob, terms, 1.0 / diagH) + math.exp(2 * sigma)


# ----------------------------------------------------------------------------
# R harness pieces                                        R/fitting.R
# ----------------------------------------------------------------------------
def quantile7(x, probs):
    """R's stats::quantile.default, type = 7 (what .genknotlist calls, R/fitting.R:177-185;
    base R, not part of the reference checkout): index = 1 + (n - 1) p, lo = floor(index),
    hi = ceiling(index), h = index - lo, (1 - h) x[lo] + h x[hi] on the sorted sample (1-based;
    the 4 eps fuzz of quantile.default belongs to types 4-6, 8 and 9, type 7 has none).  Agrees
    with numpy's method="linear" to the last bit or two."""
    xs = np.sort(np.asarray(x, dtype=np.float64))
    n = len(xs)
    probs = np.asarray(probs, dtype=np.float64)
    index = 1.0 + (n - 1) * probs
    lo = np.floor(index).astype(np.int64)
    hi = np.ceil(index).astype(np.int64)
    h = index - lo
    lo = np.clip(lo, 1, n) - 1
    hi = np.clip(hi, 1, n) - 1
    out = xs[lo].copy()
    nz = (h > 0) & (xs[hi] != out)
    out[nz] = (1 - h[nz]) * xs[lo[nz]] + h[nz] * xs[hi[nz]]
    return out


def genknotlist(bassize, x):
    """.genknotlist (R/fitting.R:177-185)."""
    x = np.asarray(x, dtype=np.float64)
    out = []
    for k in range(x.shape[1]):
        b = int(bassize[k])
        probs = np.linspace(0, 1, b) * b / (b + 1) + 0.5 / (b + 1)
        out.append(quantile7(x[:, k], probs))
    return out


def getsteps(numb, sampsize, sigtonoiseratio=1e-3, tol=0.001):
    """.getsteps (R/fitting.R:188-195)."""
    r = math.sqrt(numb / sampsize)
    # R evaluates (1 + r)^2 / 0 to Inf and min(1000, Inf) = 1000
    kapp = math.inf if r == 1.0 else (1 + r) ** 2 / (1 - r) ** 2
    kapp = min(1000, kapp)
    iterest = 0.5 * math.sqrt(kapp) * math.log(2 * sampsize *
                                                sigtonoiseratio / tol)
    return int(math.ceil(2 * iterest))


def borehole8d(x):
    """obtest_borehole8d (R/testfuncs.R:32-46)."""
    x = np.asarray(x, dtype=np.float64)
    rw = x[:, 0] * (0.15 - 0.05) + 0.05
    r = x[:, 1] * (50000 - 100) + 100
    Tu = x[:, 2] * (115600 - 63070) + 63070
    Hu = x[:, 3] * (1110 - 990) + 990
    Tl = x[:, 4] * (116 - 63.1) + 63.1
    Hl = x[:, 5] * (820 - 700) + 700
    L = x[:, 6] * (1680 - 1120) + 1120
    Kw = x[:, 7] * (12045 - 9855) + 9855
    m1 = 2 * math.pi * Tu * (Hu - Hl)
    m2 = np.log(r / rw)
    m3 = 1 + 2 * L * Tu / (m2 * np.square(rw) * Kw) + Tu / Tl
    return m1 / m2 / m3 - 77


# ----------------------------------------------------------------------------
# Synthetic workload of BASELINE.md section 3 (defined by this repo, not by
# the reference; restated here independently of outerbase_amd/data.py so the
# tests can cross-check the device generator).
# ----------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def splitmix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_u(seed, row0, nrows, d):
    """u(i,j) = (splitmix64(splitmix64(seed) ^ (i*d + j)) >> 11) * 2^-53: the hashed seed
    masks the counter, so that the streams of two seeds do not overlap."""
    i = np.arange(row0, row0 + nrows, dtype=np.uint64)[:, None]
    j = np.arange(d, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ctr = splitmix64(np.uint64(seed)) ^ (i * np.uint64(d) + j)
    return (splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def synth_xy(seed, row0, nrows, kinds):
    d = len(kinds)
    u = synth_u(seed, row0, nrows, d)
    x = 0.02 + 0.96 * u
    for j, kd in enumerate(kinds):
        if kd == "mat25ang":
            x[:, j] *= 6.283185
    u8 = np.zeros((nrows, 8))
    u8[:, :min(8, d)] = (0.02 + 0.96 * u)[:, :min(8, d)]
    if d < 8:
        u8[:, d:] = 0.5
    y = borehole8d(u8)
    for j in range(8, d):
        y = y + (20.0 / (j + 1)) * np.sin(2 * math.pi * u[:, j])
    return x, y


def bench_knots(kinds, m=40):
    out = []
    for kd in kinds:
        g = 0.001 + 0.025 * np.arange(m) if m <= 40 else np.linspace(0.001, 0.976, m)
        if kd == "mat25ang":
            g = g * 6.283185
        out.append(g)
    return out
