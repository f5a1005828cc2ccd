"""CPU restatement of the reference's R-side harness and of the lpdf object model it drives --
TEST INFRASTRUCTURE ONLY (the product path must never import this; checked by
tests/test_host_logic.py).  NumPy throughout, sized for n of a few hundred.

  lpdf classes   src/fit.h:23-361, src/fit.cpp:37-612, src/lpdfs/{logpr_gauss,loglik_gauss,
                 loglik_gda}.cpp  -> LogprGauss, LoglikGauss, LoglikGda, LpdfVec (update, optcg,
                 hessmult, the diagonal marginal adjustment, the parameter priors)
  BFGS_std       R/outersupport.R:30-171
  .lpdfwrapper   R/outersupport.R:206-226
  BFGS_lpdf      R/outersupport.R:192-203
  obfit / obpred R/fitting.R:27-155 (the random row subset of :81 is an argument: R's sample()
                 is not reproducible outside R)

Written from the reference text, independently of outerbase_amd/fitting.py (which it checks).
Parity unpinned: the reference ships no fixture for any of this.
"""
import math

import numpy as np

import ob_oracle as O


# ---- lpdf base: the Gaussian priors on para (fit.cpp:133-158) ---------------------------------
class _Lpdf:
    def paralpdf(self, parap):
        parap = np.atleast_1d(np.asarray(parap, dtype=np.float64))
        if len(parap) != len(self.para0):
            return -math.inf
        return float(-0.5 * np.sum(np.square(parap - self.para0) / self.paravar))

    def paralpdf_grad(self, parap):
        parap = np.atleast_1d(np.asarray(parap, dtype=np.float64))
        out = np.zeros(len(self.para))
        if len(parap) != len(self.para0):
            return out
        return out - (parap - self.para0) / self.paravar


# ---- logpr_gauss (logpr_gauss.cpp:41-146) -------------------------------------------------------
class LogprGauss(_Lpdf):
    def __init__(self, om, terms):
        self.om = om
        self.terms = np.asarray(terms)
        self.para0 = np.array([6.0])                    # :48
        self.paravar = np.array([4.0])                  # :50
        self.para = self.para0.copy()
        self.updateom()

    def updateom(self):                                 # :68-71
        self.coeffsd = np.sqrt(self.om.getvar(self.terms))
        self.coefflvarge = self.om.getlvar_gradhyp(self.terms)

    def updatepara(self, para):                         # :78-81
        self.para = np.array(para, dtype=np.float64)

    def updateterms(self, terms):                       # :88-93
        self.terms = np.asarray(terms)
        self.updateom()

    def update(self, coeff, want_gradhyp):              # :98-108
        sca = math.exp(self.para[0])
        stdresid = coeff / (self.coeffsd * sca)
        self.val = float(-0.5 * np.sum(np.square(stdresid)) - np.sum(np.log(self.coeffsd * sca)))
        self.grad = -stdresid / (self.coeffsd * sca)
        if want_gradhyp:
            self.gradhyp = (0.5 * self.coefflvarge).T @ (np.square(stdresid) - 1)
            self.gradpara = np.array([np.sum(np.square(stdresid)) - len(self.coeffsd)])

    def hessmult(self, g):                              # :113-115
        return g / np.square(self.coeffsd * math.exp(self.para[0]))

    def diaghess(self):                                 # :122-124
        return 1.0 / np.square(self.coeffsd * math.exp(self.para[0]))

    def diaghessgradhyp(self):                          # :131-135
        return -self.coefflvarge / np.square(self.coeffsd * math.exp(self.para[0]))[:, None]

    def diaghessgradpara(self):                         # :143-145
        return (-2.0 / np.square(self.coeffsd * math.exp(self.para[0])))[:, None]


# ---- loglik_gauss (loglik_gauss.cpp:41-172) -----------------------------------------------------
class LoglikGauss(_Lpdf):
    def __init__(self, om, terms, y, x):
        self.om = om
        self.terms = np.asarray(terms)
        self.y = np.asarray(y, dtype=np.float64)
        self.ob = O.OuterBase(om, x, dograd=True)
        self.para0 = np.array([math.log(0.01 * O.rvar(self.y))])   # :46
        self.paravar = np.array([1.0])                             # :48
        self.para = self.para0.copy()

    def updateom(self):                                 # :68-70
        self.ob.build()

    def updatepara(self, para):                         # :87-92
        self.para = np.array(para, dtype=np.float64)

    def updateterms(self, terms):                       # :99-102
        self.terms = np.asarray(terms)

    def update(self, coeff, want_gradhyp):              # :110-130
        obssd = math.exp(self.para[0])
        if want_gradhyp:
            yhat, yhatge = O.ob_mm_gradhyp(self.ob, self.terms, coeff)
        else:
            yhat = O.ob_mm(self.ob, self.terms, coeff)
        r = (yhat - self.y) / obssd
        r2 = np.square(r)
        self.val = float(-0.5 * np.sum(r2) - len(self.y) * math.log(obssd))
        r = -r / obssd
        self.grad = O.ob_tmm(self.ob, self.terms, r)
        if want_gradhyp:
            self.gradhyp = r @ yhatge
            self.gradpara = np.array([np.sum(r2) - len(self.y)])

    def hessmult(self, g):                              # :137-145
        t = O.ob_mm(self.ob, self.terms, g) / math.exp(2 * self.para[0])
        return O.ob_tmm(self.ob, self.terms, t)

    def diaghess(self):                                 # :154-157
        return math.exp(-2 * self.para[0]) * O.ob_sqcolsums(self.ob, self.terms)

    def diaghessgradhyp(self):                          # :165-168
        return math.exp(-2 * self.para[0]) * O.ob_sqcolsums_gradhyp(self.ob, self.terms)

    def diaghessgradpara(self):                         # :176-179
        return (-2 * math.exp(-2 * self.para[0]) * O.ob_sqcolsums(self.ob, self.terms))[:, None]


# ---- loglik_gda (loglik_gda.cpp:49-235) ---------------------------------------------------------
class LoglikGda(_Lpdf):
    def __init__(self, om, terms, y, x):
        self.om = om
        self.terms = np.asarray(terms)
        self.y = np.asarray(y, dtype=np.float64)
        self.ob = O.OuterBase(om, x, dograd=True)
        self.doda = True                                            # :52 (R: loglik$dodiag)
        self.para0 = np.array([0.5 * math.log(0.01 * O.rvar(self.y)), 0.0])   # :58-59
        self.paravar = np.array([4.0, 4.0])                         # :61-62
        self.para = self.para0.copy()

    def updateom(self):                                 # :86-89
        self.ob.build()

    def updatepara(self, para):                         # :96-99
        self.para = np.array(para, dtype=np.float64)

    def updateterms(self, terms):                       # :106-110
        self.terms = np.asarray(terms)

    def buildstd(self):                                 # :216-235
        e0, e1 = math.exp(2 * self.para[0]), math.exp(2 * self.para[1])
        n = len(self.y)
        rterms = O.ob_residvar(self.ob, self.terms)
        obsvar = np.full(n, e0)
        if self.doda:
            obsvar = obsvar + e1 * rterms
        self.obssd = np.sqrt(obsvar)
        if self.doda:
            self.obssd_gradhyp = O.ob_residvar_gradhyp(self.ob, self.terms) * \
                ((e1 * 0.5) / self.obssd)[:, None]
        self.obssd_gradpara = np.zeros((n, 2))
        self.obssd_gradpara[:, 0] = e0 / self.obssd
        if self.doda:
            self.obssd_gradpara[:, 1] = e1 * rterms / self.obssd

    def update(self, coeff, want_gradhyp):              # :117-150
        if want_gradhyp:
            yhat, yhatge = O.ob_mm_gradhyp(self.ob, self.terms, coeff)
        else:
            yhat = O.ob_mm(self.ob, self.terms, coeff)
        self.buildstd()
        sd = self.obssd
        r = (yhat - self.y) / sd
        r2 = np.square(r)
        self.val = float(-0.5 * np.sum(r2) - np.sum(np.log(sd)))
        r = -r / sd
        r2 = r2 / sd
        self.grad = O.ob_tmm(self.ob, self.terms, r)
        if want_gradhyp:
            self.gradhyp = r @ yhatge
            if self.doda:
                self.gradhyp = self.gradhyp + r2 @ self.obssd_gradhyp - (1 / sd) @ self.obssd_gradhyp
            self.gradpara = r2 @ self.obssd_gradpara - (1 / sd) @ self.obssd_gradpara

    def hessmult(self, g):                              # :160-169
        t = O.ob_mm(self.ob, self.terms, g) / self.obssd / self.obssd
        return O.ob_tmm(self.ob, self.terms, t)

    def diaghess(self):                                 # :177-180
        return O.ob_sqtmm(self.ob, self.terms, 1 / np.square(self.obssd))

    def diaghessgradhyp(self):                          # :187-200
        w = 1 / np.square(self.obssd)
        lh = O.ob_sqtmm_gradhyp(self.ob, self.terms, w)
        if self.doda:
            lh = lh + O.ob_sqtmm(self.ob, self.terms, self.obssd_gradhyp * (w * (-2 / self.obssd))[:, None])
        return lh

    def diaghessgradpara(self):                         # :207-214
        w = (1 / np.square(self.obssd)) * (-2 / self.obssd)
        return O.ob_sqtmm(self.ob, self.terms, self.obssd_gradpara * w[:, None])


# ---- lpdfvec (fit.cpp:174-460) ------------------------------------------------------------------
class LpdfVec(_Lpdf):
    """new(lpdfvec, a, b): lpdflist = [a, b], para = [a.para, b.para] (fit.cpp:174-198)."""

    def __init__(self, a, b):
        self.list = [a, b]
        self.nterms = a.terms.shape[0]
        self.domarg = True                              # fit.h: domargadj defaults to true
        self._slices()
        self.para = np.concatenate([a.para, b.para])
        self.coeff = np.zeros(0)
        self.redohess = True

    def _slices(self):
        n0 = len(self.list[0].para)
        self.sl = [slice(0, n0), slice(n0, n0 + len(self.list[1].para))]

    def paralpdf(self, parap):                          # fit.cpp:470-478
        parap = np.asarray(parap, dtype=np.float64)
        return sum(l.paralpdf(parap[s]) for l, s in zip(self.list, self.sl))

    def paralpdf_grad(self, parap):                     # fit.cpp:487-496
        parap = np.asarray(parap, dtype=np.float64)
        return np.concatenate([l.paralpdf_grad(parap[s]) for l, s in zip(self.list, self.sl)])

    def updateom(self):                                 # fit.cpp:207-210
        for l in self.list:
            l.updateom()
        self.redohess = True

    def updatepara(self, para):                         # fit.cpp:219-228
        para = np.asarray(para, dtype=np.float64)
        self.para = para.copy()
        for l, s in zip(self.list, self.sl):
            l.updatepara(para[s])
        self.redohess = True

    def updateterms(self, terms):                       # fit.cpp:237-244
        for l in self.list:
            l.updateterms(terms)
        self.nterms = np.asarray(terms).shape[0]
        self.redohess = True

    def buildhess(self):                                # fit.cpp:252-268 (diagonal form)
        if not self.redohess:
            return
        self.diaghessv = self.list[0].diaghess() + self.list[1].diaghess()
        if self.domarg:
            dgh = self.list[0].diaghessgradhyp() + self.list[1].diaghessgradhyp()
            dgp = np.concatenate([self.list[0].diaghessgradpara(), self.list[1].diaghessgradpara()], axis=1)
            self.val_margadj = float(-0.5 * np.sum(np.log(self.diaghessv)))
            self.gradhyp_margadj = -0.5 * np.sum(dgh / self.diaghessv[:, None], axis=0)
            self.gradpara_margadj = -0.5 * np.sum(dgp / self.diaghessv[:, None], axis=0)
        self.redohess = False

    def update(self, coeff, want_gradhyp=False):        # fit.cpp:323-363, margadj :371-380
        self.coeff = np.array(coeff, dtype=np.float64)
        for l in self.list:
            l.update(self.coeff, want_gradhyp)
        self.buildhess()
        self.val = sum(l.val for l in self.list)
        self.grad = self.list[0].grad + self.list[1].grad
        if want_gradhyp:
            self.gradhyp = self.list[0].gradhyp + self.list[1].gradhyp
            self.gradpara = np.concatenate([l.gradpara for l in self.list])
        if self.domarg:
            self.val += self.val_margadj
            if want_gradhyp:
                self.gradhyp = self.gradhyp + self.gradhyp_margadj
                self.gradpara = self.gradpara + self.gradpara_margadj

    def hessmult(self, g):                              # fit.cpp:382-392
        return self.list[0].hessmult(g) + self.list[1].hessmult(g)

    def optcg(self, tol, maxepch):                      # lpdf::optcg, fit.cpp:37-96
        if len(self.coeff) != self.nterms:
            self.coeff = np.zeros(self.nterms)
        coeff = self.coeff.copy()
        self.update(coeff)
        m = self.diaghessv
        if not np.all(np.isfinite(m)) and not np.all(np.isfinite(self.grad)):
            self.val = -math.inf
            return 0
        rm = self.grad / m
        p = rm.copy()
        q = self.hessmult(p)
        valdiff = 10.0
        k = 0
        while k < maxepch:
            num = float(np.sum(self.grad * rm))
            if num < tol and valdiff < tol:
                break
            denom = float(np.sum(q * p))
            alpha = num / denom
            coeff = coeff + alpha * p
            valo = self.val
            self.update(coeff)
            valdiff = self.val - valo
            rm = self.grad / m
            num2 = -float(np.sum((alpha * q) * rm))
            beta = num2 / num
            p = rm + beta * p
            q = self.hessmult(p)
            k += 1
        self.update(coeff, want_gradhyp=True)           # fit.cpp:87-93
        return k


# ---- BFGS_std (R/outersupport.R:30-171) ---------------------------------------------------------
def bfgs_std(funcw, parlist, B=None, lr=0.1, **kw):
    """parlist: dict name -> 1-d array (R's list; unlist / relist keep the order of the names).
    Returns dict(parlist, B, lr, optid, trace); trace has one entry per outer iteration:
    (k, objective after the step or None on a restart, learning rate, wolfe 1, wolfe 2, number of
    line-search evaluations)."""
    c1, c2, numatte0 = 0.0001, 0.9, 5
    names = list(parlist)
    lens = [len(np.atleast_1d(parlist[nm])) for nm in names]

    def unlist(pl):
        return np.concatenate([np.atleast_1d(np.asarray(pl[nm], dtype=np.float64)) for nm in names])

    def relist(v):
        out, at = {}, 0
        for nm, ln in zip(names, lens):
            out[nm] = np.array(v[at:at + ln])
            at += ln
        return out

    def ungrad(optid):      # unlist(NULL) is empty: sum(dirc * numeric(0)) is 0 in R
        return unlist(optid["gval"]) if optid["gval"] is not None else None

    def dot(a, b):
        return 0.0 if b is None else float(np.sum(a * b))

    def is_na(v):
        return isinstance(v, float) and math.isnan(v)

    parv = unlist(parlist)
    optid = funcw(relist(parv), **kw)
    valo = optid["val"]
    go = ungrad(optid)
    if go is None or np.any(np.isnan(go)):
        raise ValueError("initial gradient was undefined, stopping.")
    resetB = B is None
    if B is None:
        B = np.diag(1 / np.sqrt(go ** 2 + 0.001))
    twice = False
    lr0 = lr00 = lr
    trace = [(0, valo, lr, None, None, 0)]
    for k in range(1, 101):
        dirc = -(B @ go)
        st = lr * dirc
        parvp = parv + st
        optid = funcw(relist(parvp), **kw)
        w1 = (optid["val"] - valo) - c1 * lr * dot(dirc, go)
        w2 = -dot(dirc, ungrad(optid)) + c2 * dot(dirc, go)
        numatte, lrlb, lrub, lrh = numatte0, 0.0, math.inf, lr
        optidh = optid
        nls = 0
        while numatte > 0 and (is_na(w1) or is_na(w2) or w1 > 0 or w2 > 0):
            if is_na(w1) or is_na(w2) or w1 > 0:
                lrub = lrh
                lrh = 0.5 * (lrlb + lrub)
            else:
                lrlb = lrh
                lrh = 0.5 * (lrlb + lrub) if math.isfinite(lrub) else 2 * lrlb
            parvp = parv + lrh * dirc
            optidh = funcw(relist(parvp), **kw)
            w1 = (optidh["val"] - valo) - c1 * lrh * dot(dirc, go)
            w2 = -dot(dirc, ungrad(optidh)) + c2 * dot(dirc, go)
            numatte -= 1
            nls += 1
        if is_na(w1) or is_na(w2):
            raise ValueError("something is very wrong... stuck on NAs")
        if w1 > 0:
            if resetB:
                c2 = c2 ** 0.5
                lr0 = lr0 / 10
                lr = lr0
            if lr0 < lr00 / (10 ** 2 + 1):
                break
            optid = funcw(relist(parv), **kw)          # do not feed it extra info
            valo = optid["val"]
            go = ungrad(optid)
            B = np.diag(1 / np.sqrt(0.001 + go ** 2))
            resetB = True
            trace.append((k, None, lr, None, None, nls))
        else:
            if lr != lrh:
                lr = lrh
                st = parvp - parv
                parv = parvp
                optid = optidh
            else:
                parv = parvp
            small = k > 2 and float(np.sum(st * go)) > -len(go) / 4
            if small and twice:
                break
            if small:
                twice = True
            goo = go
            valo = optid["val"]
            go = ungrad(optid)
            yv = go - goo
            trace.append((k, valo, lr, w1, w2, nls))
            if resetB:
                B = float(np.sum(st * yv)) / float(np.sum(yv * yv)) * np.eye(len(parv))
                resetB = False
            cvh = 1 / float(np.sum(st * yv))
            M1 = np.eye(len(go)) - cvh * np.outer(st, yv)
            B = M1 @ B @ M1.T + cvh * np.outer(st, st)
            lr = lr ** 0.9                              # drift toward 1
    optid = funcw(relist(parv), **kw)                  # finish by evaluating
    return {"parlist": relist(parv), "B": B, "lr": lr, "optid": optid, "trace": trace}


# ---- .lpdfwrapper, BFGS_lpdf (R/outersupport.R:192-226) ----------------------------------------
def lpdfwrapper(parlist, om, logpdf, newt=False, cgsteps=100, cgtol=0.001):
    regpara = logpdf.paralpdf(parlist["para"])
    reghyp = om.hyplpdf(parlist["hyp"])
    if math.isfinite(regpara) and math.isfinite(reghyp):
        om.hyp_set(parlist["hyp"])                      # om$updatehyp
        logpdf.updateom()
        logpdf.updatepara(parlist["para"])
        if newt:
            raise NotImplementedError("obfit never asks for Newton steps")
        logpdf.optcg(cgtol, cgsteps)
        gval = {"hyp": -logpdf.gradhyp - om.hyplpdf_grad(parlist["hyp"]),
                "para": -logpdf.gradpara - logpdf.paralpdf_grad(parlist["para"])}
        return {"val": -logpdf.val - reghyp - regpara, "gval": gval}
    return {"val": math.inf, "gval": None}


def bfgs_lpdf(om, logpdf, parlist=None, newt=False, cgsteps=100, cgtol=0.001, **kw):
    """cgsteps / cgtol stop here, as in the reference: BFGS_std is called with om, newt, logpdf and
    what came in `...` only, so the wrapper always runs with its own defaults (100, 0.001)."""
    parlist = dict(parlist or {})
    if parlist.get("hyp") is None:
        parlist["hyp"] = om.hyp.copy()                  # gethyp(om)
    if parlist.get("para") is None:
        parlist["para"] = logpdf.para.copy()            # getpara(logpdf)
    parlist = {"hyp": parlist["hyp"], "para": parlist["para"]}
    lpdfwrapper(parlist, om, logpdf, newt=newt)         # start by aligning para
    return bfgs_std(lpdfwrapper, parlist, om=om, newt=newt, logpdf=logpdf, **kw)


# ---- obfit / obpred (R/fitting.R:27-155) --------------------------------------------------------
def obfit(x, y, numb, covnames, subsetinds, numberopts=2, om=None):
    """The two-stage fit.  subsetinds: the ssr rows R draws with sample(length(y), ssr)
    (R/fitting.R:81); om: an OuterMod to use (tests pass one whose eigen-rotation is shared with
    the device model), a fresh one by default.  Returns the model list plus the BFGS traces."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, d = x.shape
    y_cent = float(np.mean(y))
    y_sca = float(np.std(y, ddof=1))
    y = (y - y_cent) / y_sca                            # :55-57
    if om is None:
        om = O.OuterMod()
    om.setcovfs(covnames)
    om.setknot(O.genknotlist([40] * d, x))              # :75
    numbr = min(n // 2, numb, 80 * d)                   # :77
    terms = om.selectterms(numbr)
    ssr = min(n, 3 * numbr)
    if len(subsetinds) != ssr:
        raise ValueError("subsetinds must hold min(n, 3 numbr) = %d rows" % ssr)
    logpr = LogprGauss(om, terms)
    yr, xr = y[subsetinds], x[subsetinds, :]
    loglik = LoglikGda(om, terms, yr, xr)
    loglik.doda = True
    logpdf = LpdfVec(logpr, loglik)
    opt1 = bfgs_lpdf(om, logpdf)                        # :94-95 (cgsteps stops in BFGS_lpdf)
    terms = om.selectterms(numb)                        # :97
    bassize = np.ceil(np.maximum(16, np.minimum(70, 2 * terms.max(axis=0))))
    om.setknot(O.genknotlist(bassize, x))               # :101-104
    loglik_faster = LoglikGauss(om, terms, y, x)
    logpdf_faster = LpdfVec(logpr, loglik_faster)
    logpdf_faster.domarg = True
    B = opt1["B"][:-1, :-1]                             # :110-111
    B = len(yr) / n * B                                 # :113
    logpdf_faster.updatepara(logpdf.para[:2])           # :114
    lr = opt1["lr"]
    traces = [opt1["trace"]]
    opt = opt1
    for _ in range(numberopts):                         # :118-131
        terms = om.selectterms(numb)
        logpdf_faster.updateterms(terms)
        opt = bfgs_lpdf(om, logpdf_faster, B=B, lr=lr / 2)
        B, lr = opt["B"], opt["lr"]
        traces.append(opt["trace"])
    return {"y_cent": y_cent, "y_sca": y_sca, "om": om, "loglik": loglik_faster, "logpdf": logpdf_faster,
            "terms": terms, "optinfo": opt, "traces": traces}


def obpred_mean(obmodel, x):
    """obpred(...)$mean (R/fitting.R:149-152; pred_gauss::mean loglik_gauss.cpp:214-222)."""
    lg = obmodel["loglik"]
    return obmodel["y_cent"] + obmodel["y_sca"] * O.predict_mean(obmodel["om"], lg.terms, obmodel["logpdf"].coeff,
                                                                 np.asarray(x, dtype=np.float64))
