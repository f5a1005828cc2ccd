#!/usr/bin/env python
"""Generates tests/golden/*.npz from the CPU oracle (oracle/ob_oracle.py).

The reference ships no golden vectors and cannot be built here (SURVEY.md
section 8c), so these fixtures pin the ORACLE (against drift) and give the GPU
tests fixed data to hit; they are not outputs of the reference itself.
Run:  python oracle/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ob_oracle as O  # noqa: E402

LEAD = 8   # leading levels of each dimension's basemat block kept in the fixture
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: (kinds, knots per dim, n, p, hyp shift, seed)
    "ref_basic_d8": (["mat25pow"] + ["mat25"] * 7, None, 15, 20, 0.0, 42),   # test-obombasic.R
    "mixed_d3": (["mat25", "mat25pow", "mat25ang"], 24, 300, 40, 0.15, 7),
    "mat25_d8": (["mat25"] * 8, 40, 300, 36, -0.2, 11),
}


def make(name):
    kinds, m, n, p, shift, seed = CASES[name]
    d = len(kinds)
    rng = np.random.default_rng(seed)
    om = O.OuterMod()
    om.setcovfs(kinds)
    if m is None:
        knots = [np.arange(0.001, 0.999, 0.025)] * d
    else:
        knots = O.bench_knots(kinds, m)
    hyp = om.hyp + shift * np.linspace(-1, 1, len(om.hyp))
    om.hyp_set(hyp)
    om.setknot(knots)
    x = 0.02 + 0.96 * rng.random((n, d))
    for j, k in enumerate(kinds):
        if k == "mat25ang":
            x[:, j] *= 6.283185
    terms = om.selectterms(p)
    ob = O.OuterBase(om, x)
    B = O.ob_getmat(ob, terms)
    a = rng.standard_normal(p)
    v = rng.standard_normal(n)
    y = B @ (np.sqrt(om.getvar(terms)) * rng.standard_normal(p)) + 0.05 * rng.standard_normal(n)
    y = (y - y.mean()) / y.std(ddof=1)
    sigma = O.default_sigma(y)
    theta, H = O.fit_newton(ob, terms, y, sigma=sigma)
    theta_cg, iters, diagH = O.fit_cg(ob, terms, y, sigma=sigma, tol=0.0, maxit=12)  # fixed iteration count
    xnew = 0.02 + 0.96 * rng.random((9, d))
    for j, k in enumerate(kinds):
        if k == "mat25ang":
            xnew[:, j] *= 6.283185
    G, g = O.gram(ob, terms, y)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        kinds=np.array(kinds), knotpt=om.knotpt, knotptst=om.knotptst, hyp=om.hyp,
        rotmat=om.rotmat, basisvar=om.basisvar, maxlevel=om.maxlevel,
        x=x, terms=terms, termvar=om.getvar(terms),
        basemat_lead=np.stack([ob.basemat[:, om.knotptst[l]:om.knotptst[l] + LEAD] for l in range(d)], axis=1),
        basescale=ob.basescale, basescalemat=ob.basescalemat,
        B=B, a=a, Ba=O.ob_mm(ob, terms, a), v=v, Btv=O.ob_tmm(ob, terms, v),
        sqBa=O.ob_sqmm(ob, terms, np.abs(a)), sqcolsums=O.ob_sqcolsums(ob, terms),
        G=G, g=g, y=y, sigma=sigma, rho=O.DEFAULT_RHO, H=H, theta=theta,
        theta_cg=theta_cg, cg_iters=iters, diagH=diagH,
        xnew=xnew, mean=O.predict_mean(om, terms, theta, xnew),
        var_gauss=O.predict_var_gauss(om, terms, np.diag(H), sigma, xnew),
        var_std=O.predict_var_std(om, terms, H, sigma, xnew))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for nm in CASES:
        make(nm)
        print("wrote", nm)
