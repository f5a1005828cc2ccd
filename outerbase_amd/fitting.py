"""obfit / obpred and the BFGS hyper-parameter loop -- the reference's R-side harness
(R/fitting.R:27-195, R/outersupport.R:30-226) over the device-backed module mirror
(obmod.py).  Same names, argument meaning and stop() conditions; errors are ValueError,
the reference's warning() calls are warnings.warn.

Everything data-sized runs on the GPU through libobhip (basis builds, B a, B^T a and
their hyper-parameter gradients, the PCG solves); this file only holds the scalar
optimisation logic the reference keeps in R.
"""
import ctypes as C
import math
import warnings

import numpy as np

from . import _lib, obmod
from ._lib import call
from .obmod import (gethyp, getpara, loglik_gauss, loglik_gda, logpr_gauss, lpdfvec, outermod,
                    predictor, setcovfs, setknot)

_COVS = obmod.listcov()                         # R/fitting.R:6-8


# ---- BFGS_std (R/outersupport.R:30-171) -------------------------------------------------
def BFGS_std(funcw, parlist, B=None, lr=0.1, verbose=0, **kw):
    """Minimise funcw(parlist, **kw) -> {"val", "gval"} with BFGS and the reference's
    Wolfe-condition bisection; parlist is a dict of 1-d arrays, kept in key order.
    Infinite / NaN values are stepped away from like the reference does."""
    c1, c2, numatte0 = 0.0001, 0.9, 5
    keys = list(parlist.keys())
    sizes = [len(np.atleast_1d(parlist[k])) for k in keys]

    def flat(pl):
        return np.concatenate([np.atleast_1d(np.asarray(pl[k], dtype=np.float64)) for k in keys])

    def relist(v):
        out, o = {}, 0
        for k, m in zip(keys, sizes):
            out[k] = v[o:o + m].copy()
            o += m
        return out

    def gflat(optid):
        # R: unlist(NULL) is empty and sum(dirc * empty) is 0
        return flat(optid["gval"]) if optid["gval"] is not None else np.zeros(sum(sizes))

    def isna(v):
        return v is None or (isinstance(v, float) and math.isnan(v))

    parv = flat(parlist)
    optid = funcw(relist(parv), **kw)
    valo = optid["val"]
    go = gflat(optid)
    resetB = True
    if B is None:
        B = np.diag(1.0 / np.sqrt(go ** 2 + 0.001))
    else:
        resetB = False
    twice = False
    lr0 = lr00 = lr
    log = [dict(iter=0, val=valo, lr=lr)]
    if np.any(np.isnan(go)):
        raise ValueError("initial gradient was undefined, stopping.")
    dirc = np.zeros_like(go)
    k = 0
    for k in range(1, 101):
        dirc = -B @ go
        st = lr * dirc
        parvp = parv + st
        optid = funcw(relist(parvp), **kw)

        def wolfe(oid, step):
            w1 = (oid["val"] - valo) - c1 * step * float(dirc @ go)
            w2 = -float(dirc @ gflat(oid)) + c2 * float(dirc @ go)
            return w1, w2

        w1, w2 = wolfe(optid, lr)
        numatte, lrlb, lrub, lrh = numatte0, 0.0, math.inf, lr
        optidh = optid
        while numatte > 0 and (isna(w1) or isna(w2) or w1 > 0 or w2 > 0):
            if isna(w1) or isna(w2) or w1 > 0:
                lrub = lrh
                lrh = 0.5 * (lrlb + lrub)
            else:
                lrlb = lrh
                lrh = 0.5 * (lrlb + lrub) if math.isfinite(lrub) else 2 * lrlb
            parvp = parv + lrh * dirc
            optidh = funcw(relist(parvp), **kw)
            w1, w2 = wolfe(optidh, lrh)
            numatte -= 1
        if isna(w1) or isna(w2):
            raise ValueError("something is very wrong... stuck on NAs")
        if w1 > 0:
            if resetB:
                c2 = c2 ** 0.5
                lr0 = lr0 / 10
                lr = lr0
            if lr0 < lr00 / (10 ** 2 + 1):
                break
            optid = funcw(relist(parv), **kw)
            valo = optid["val"]
            go = gflat(optid)
            B = np.diag(1.0 / np.sqrt(0.001 + go ** 2))
            resetB = True
            log.append(dict(iter=k, val=None, lr=lr))
            if verbose > 0:
                print("restarted hessian")
        else:
            if lr != lrh:
                lr = lrh
                st = parvp - parv
                parv = parvp
                optid = optidh
            else:
                parv = parvp
            if k > 2 and float(st @ go) > -len(go) / 4 and twice:
                break
            elif k > 2 and float(st @ go) > -len(go) / 4:
                twice = True
            goo = go
            valo = optid["val"]
            go = gflat(optid)
            yv = go - goo
            log.append(dict(iter=k, val=valo, w1=w1, w2=w2, lr=lr))
            if verbose > 1:
                print(log[-1])
            if resetB:
                B = float(st @ yv) / float(yv @ yv) * np.eye(len(parv))
                resetB = False
            cvh = 1.0 / float(st @ yv)
            M1 = np.eye(len(go)) - cvh * np.outer(st, yv)
            B = M1 @ B @ M1.T + cvh * np.outer(st, st)
            lr = lr ** 0.9   # drift toward 1
    optid = funcw(relist(parv), **kw)
    if verbose > 0:
        print("num iter: %d  obj start: %g  obj end: %g  final learning rate: %g"
              % (k, log[0]["val"], optid["val"], lr))
    return dict(parlist=relist(parv), B=B, lr=lr, optid=optid, log=log)


# ---- .lpdfwrapper / BFGS_lpdf (R/outersupport.R:173-226) --------------------------------
def _lpdfwrapper(parlist, om, logpdf, newt=False, cgsteps=100, cgtol=0.001):
    regpara = logpdf.paralpdf(parlist["para"])
    reghyp = om.hyplpdf(parlist["hyp"])
    if math.isfinite(regpara) and math.isfinite(reghyp):
        om.updatehyp(parlist["hyp"])
        logpdf.updateom()
        logpdf.updatepara(parlist["para"])
        if newt:
            logpdf.optnewton()
        else:
            logpdf.optcg(cgtol, cgsteps)
        gval = {"hyp": -logpdf.gradhyp - om.hyplpdf_grad(parlist["hyp"]),
                "para": -logpdf.gradpara - logpdf.paralpdf_grad(parlist["para"])}
        return {"val": -logpdf.val - reghyp - regpara, "gval": gval}
    return {"val": math.inf, "gval": None}


def BFGS_lpdf(om, logpdf, parlist=None, newt=False, cgsteps=100, cgtol=0.001, **kw):
    """om and logpdf end up at the optimum; the return value is information only.  As in
    the reference (R/outersupport.R:206-216) cgsteps / cgtol are accepted but the wrapper
    runs with its own defaults (they are not forwarded by BFGS_std's call)."""
    parlist = dict(parlist or {})
    if parlist.get("hyp") is None:
        parlist["hyp"] = gethyp(om)
    if parlist.get("para") is None:
        parlist["para"] = getpara(logpdf)
    parlist = {"hyp": np.asarray(parlist["hyp"], dtype=np.float64),
               "para": np.asarray(parlist["para"], dtype=np.float64)}
    _lpdfwrapper(parlist, om, logpdf, newt=newt)   # start by aligning para
    return BFGS_std(_lpdfwrapper, parlist, om=om, newt=newt, logpdf=logpdf, **kw)


# ---- helpers of R/fitting.R:158-195 -------------------------------------------------------
def _checkcov(covname, xcol):
    if covname not in _COVS:
        raise ValueError("covariances must be from listcov()")
    cf = getattr(obmod, "covf_" + covname)()
    lo, hi = cf.lowbnd, cf.uppbnd
    if xcol.min() < lo or xcol.max() > hi:
        raise ValueError("x ranges exceed limits of covariance functions: the limits are between "
                         "%g and %g, try rescaling" % (lo, hi))
    if xcol.max() - xcol.min() < (hi - lo) / 20:
        raise ValueError("x are too small for ranges: the limits are between %g and %g, "
                         "try rescaling" % (lo, hi))


class _DeviceCopy:
    """x (n x d, column-major) in HBM for the calls below that take device pointers."""

    def __init__(self, x):
        xf = np.asfortranarray(x, dtype=np.float64)
        self.n, self.d = xf.shape
        self.ptr = C.c_void_p()
        call("obhip_malloc", C.byref(self.ptr), max(8, xf.nbytes))
        call("obhip_memcpy_h2d", self.ptr, xf.ctypes.data, xf.nbytes)

    def close(self):
        if self.ptr:
            _lib.lib.obhip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()


def _quantiles(dx, probs, comm):
    """d x q exact type-7 quantiles of the columns of x over the rows of ALL ranks
    (obhip_quantiles_dev: bisection on counts, no rank ever sorts)."""
    probs = np.ascontiguousarray(probs, dtype=np.float64)
    out = np.empty((dx.d, len(probs)))
    call("obhip_quantiles_dev", comm, dx.ptr, dx.n, dx.d, probs.ctypes.data, len(probs),
         out.ctypes.data)
    return out


def _sum_over_ranks(vals, comm):
    """element-wise sum of a few host numbers over the ranks (identity without a communicator)"""
    v = np.ascontiguousarray(vals, dtype=np.float64)
    if comm is None:
        return v
    d = C.c_void_p()
    call("obhip_malloc", C.byref(d), v.nbytes)
    try:
        call("obhip_memcpy_h2d", d, v.ctypes.data, v.nbytes)
        call("obhip_comm_allreduce_dev", comm, d, len(v))
        call("obhip_memcpy_d2h", v.ctypes.data, d, v.nbytes)
    finally:
        _lib.lib.obhip_free(d)
    return v


def _genknotlist(bassize, x, comm=None, dx=None):
    """.genknotlist (R/fitting.R:177-185): quantile(x_k, seq(0,1,len=b)*b/(b+1)+0.5/(b+1)),
    R's default type 7, over the rows of all ranks, evaluated on the device."""
    own = dx is None
    if own:
        dx = _DeviceCopy(x)
    try:
        bs = [int(b) for b in bassize]
        qs = {}
        for b in sorted(set(bs)):
            probs = np.linspace(0, 1, b) * b / (b + 1) + 0.5 / (b + 1)
            qs[b] = _quantiles(dx, probs, comm)
        return [qs[b][k].copy() for k, b in enumerate(bs)]
    finally:
        if own:
            dx.close()


def _getsteps(numb, sampsize, sigtonoiseratio=1e-3, tol=0.001):
    r = math.sqrt(numb / sampsize)
    # R: (1 + r)^2 / (1 - r)^2 is Inf at r = 1 (numb == sampsize) and min(1000, Inf) = 1000
    kapp = 1000.0 if r == 1.0 else min(1000.0, (1 + r) ** 2 / (1 - r) ** 2)
    return int(math.ceil(2 * 0.5 * math.sqrt(kapp) * math.log(2 * sampsize * sigtonoiseratio / tol)))


# ---- obfit / obpred (R/fitting.R:27-155) ---------------------------------------------------
def obfit(x, y, numb=100, verbose=0, covnames=None, hyp=None, numberopts=2, nthreads=None,
          seed=None, comm=None, row0=0):
    """Fit an outerbase model with hyper-parameter learning.  `seed` drives the row subsample
    of the first stage (R's sample(), R/fitting.R:81); nthreads is accepted and ignored (the
    work runs on the GPU).

    Row-sharded fits (no reference counterpart): every rank passes its
    own contiguous rows x, y, its obhip_comm (driver.make_comm) and row0, the index of its
    first row in the whole data set; all ranks pass the same seed.  y is standardised, the
    knots are placed at quantiles and the first stage's rows are drawn over ALL rows, every
    rank ends with the same model, and obpred predicts at whatever rows a rank holds."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n_local, d = x.shape
    if n_local != len(y):
        raise ValueError("x and y dims do not align")
    if comm is not None and seed is None:
        raise ValueError("a row-sharded fit needs the same seed on every rank")
    # sums over all rows: n, sum y, then sum (y - mean)^2 (two passes like R's sd())
    n = int(round(_sum_over_ranks([n_local], comm)[0]))
    if n < d:
        raise ValueError("dimension larger than sample size has not been tested")
    if n > 10 ** 6:
        raise ValueError("sample size should be less than 1000000")
    if d > 200:
        raise ValueError("dimension should be less than 200")
    if n > 10 ** 5:
        warnings.warn("sample size is larger than has been tested")
    if d > 20:
        warnings.warn("more than 20 dimensions has not been tested")
    if d == 1:
        raise ValueError("dimension must be larger than 1")
    if d == 2:
        raise ValueError("dimension 2 has not been tested")
    if numb < 2 * d:
        raise ValueError("number of basis functions should be less than twice the dimension")
    if numb > 100000:
        raise ValueError("number of basis functions is beyond testing")
    if numb > 5000:
        warnings.warn("number of basis functions is large, might take time to fit.")
    if numb > n:
        warnings.warn("number of basis functions larger than sample size, this has not been "
                      "thoroughly tested")
    if nthreads is not None:
        if math.ceil(nthreads) < 1:
            raise ValueError("nthreads must be bigger than 1")
        if math.ceil(nthreads) > 100:
            raise ValueError("nthreads should be small than 100 (for now).")
    y_cent = float(_sum_over_ranks([y.sum()], comm)[0]) / n
    y_sca = math.sqrt(float(_sum_over_ranks([np.sum((y - y_cent) ** 2)], comm)[0]) / (n - 1))
    y = (y - y_cent) / y_sca
    if covnames is not None and len(covnames) != d:
        raise ValueError("cov names must be same size as columns in x")
    if covnames is None:
        covnames = [_COVS[0]] * d
    for c in covnames:
        if c not in _COVS:
            raise ValueError("covariances must be from listcov()")
    if comm is None:                                  # argument guards, as the R code makes them
        xrange_ = np.stack([x.min(axis=0), x.max(axis=0)], axis=1)
        dx = None
    else:
        dx = _DeviceCopy(x)
        xrange_ = _quantiles(dx, [0.0, 1.0], comm)    # min and max of every column, all ranks
    for k in range(d):
        _checkcov(covnames[k], xrange_[k])
    if dx is None:
        dx = _DeviceCopy(x)
    om = outermod()
    setcovfs(om, covnames)
    if hyp is not None and len(hyp) == len(gethyp(om)):
        om.updatehyp(hyp)
    setknot(om, _genknotlist([40] * d, x, comm, dx))  # 40 knot points for each dim
    numbr = min(n // 2, numb, 80 * d)
    terms = om.selectterms(numbr)                     # small number of terms
    ssr = min(n, 3 * numbr)
    logpr = logpr_gauss(om, terms)
    rng = np.random.default_rng(seed)
    subsetinds = rng.choice(n, size=ssr, replace=False)
    if comm is None:
        yr, xr = y[subsetinds], x[subsetinds, :]
    else:
        # the drawn rows live on different ranks: every rank fills in the ones it holds and
        # the buffer is summed, so that all ranks run the first stage on the same rows
        buf = np.zeros((ssr, d + 1))
        mine = (subsetinds >= row0) & (subsetinds < row0 + n_local)
        loc = subsetinds[mine] - row0
        buf[mine, :d] = x[loc, :]
        buf[mine, d] = y[loc]
        buf = _sum_over_ranks(buf.ravel(), comm).reshape(ssr, d + 1)
        yr, xr = buf[:, d].copy(), buf[:, :d].copy()
    loglik = loglik_gda(om, terms, yr, xr)
    loglik.dodiag = True
    logpdf = lpdfvec(logpr, loglik)
    if verbose > 0:
        print("doing partial optimization")
    optinfo = BFGS_lpdf(om, logpdf, verbose=verbose, cgsteps=100)

    terms = om.selectterms(numb)
    bassize = np.ceil(np.maximum(16, np.minimum(70, 2 * terms.max(axis=0))))
    setknot(om, _genknotlist(bassize, x, comm, dx))
    dx.close()
    loglik_faster = loglik_gauss(om, terms, y, x)
    if comm is not None:
        loglik_faster.set_comm(comm)
    logpdf_faster = lpdfvec(logpr, loglik_faster)
    logpdf_faster.domarg = True
    Bm = optinfo["B"][:-1, :-1]                       # one fewer para: strip the last one off
    Bm = len(yr) / n * Bm                             # decrease scale
    logpdf_faster.updatepara(getpara(logpdf)[:2])
    lr = optinfo["lr"]
    for k in range(numberopts):
        # var(y) of the standardised y over all rows is 1 (R/fitting.R:115-116)
        nsteps = _getsteps(numb, n, 1.0 / math.exp(2 * getpara(logpdf_faster)[1]))
        if verbose > 0:
            print("doing optimization", k + 1, "(max number of cg steps", nsteps, ")")
        terms = om.selectterms(numb)
        logpdf_faster.updateterms(terms)
        optinfo = BFGS_lpdf(om, logpdf_faster, verbose=verbose, B=Bm, lr=lr / 2, cgsteps=nsteps)
        Bm, lr = optinfo["B"], optinfo["lr"]
    return dict(y_cent=y_cent, y_sca=y_sca, om=om, predobj=predictor(loglik_faster),
                logpdf=logpdf_faster, optinfo=optinfo)


def rvar_ratio(y, noisescale):
    """var(y) / exp(2 para[2]) of R/fitting.R:115-116"""
    return float(np.var(y, ddof=1) / math.exp(2 * noisescale))


def obpred(obmodel, x):
    """mean and var at new x (R/fitting.R:149-155)."""
    p = obmodel["predobj"]
    p.update(np.asarray(x, dtype=np.float64))
    return dict(mean=obmodel["y_cent"] + obmodel["y_sca"] * p.mean(),
                var=obmodel["y_sca"] ** 2 * p.var())
