"""Wide problems: term sets that touch more basis columns than one LDS tile holds (about 300)
or carry more than 8 factors per term.  The reference has no such limit (its loops walk the
umat, src/linalg.cpp:57-131, 286-355, 647-715); here the column-from-HBM kernels
(csrc/kernels_generic.hip) and the chunked design-matrix Gram take over.  Everything against
the CPU oracle."""
import math

import numpy as np
import pytest

from conftest import make_pair, sample_x

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


def wide_case(rng, d, m, p, max_nnz, maxlev):
    kinds = [["mat25", "mat25pow", "mat25ang"][k % 3] for k in range(d)]
    knots = []
    for kd in kinds:
        g = np.linspace(0.03, 0.97, m)
        knots.append(g * 6.283185 if kd == "mat25ang" else g)
    om_o, om_d = make_pair(kinds, knots)
    terms = np.zeros((p, d), dtype=np.int64)
    for k in range(1, p):
        nnz = int(rng.integers(1, max_nnz + 1))
        dims = rng.choice(d, size=nnz, replace=False)
        terms[k, dims] = rng.integers(1, maxlev + 1, size=nnz)
    terms = np.unique(terms, axis=0)
    return kinds, om_o, om_d, terms


@pytest.mark.parametrize("d,m,p,max_nnz,maxlev", [
    (120, 6, 700, 3, 4),     # ~440 distinct (dimension, level) columns: beyond the LDS tile
    (60, 8, 500, 12, 6),     # up to 12 factors per term and ~350 columns
])
def test_wide_term_sets_match_oracle(d, m, p, max_nnz, maxlev):
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(d + p)
    kinds, om_o, om_d, terms = wide_case(rng, d, m, p, max_nnz, maxlev)
    p = terms.shape[0]
    used = sum(len(np.unique(terms[:, l][terms[:, l] > 0])) for l in range(d)) + 1
    assert used > 304                       # the tiled kernels' limit (kMaxMuLds)
    n = 777
    x = sample_x(rng, n, kinds)
    bo = O.OuterBase(om_o, x)
    B = O.ob_getmat(bo, terms)
    bd = ob.outerbase(om_d, x)
    a, v = rng.standard_normal(p), rng.standard_normal(n)
    tol = 1e-9
    assert relerr(bd.getmat(terms), B) < tol
    assert relerr(bd.matmul(terms, a), B @ a) < tol
    assert relerr(bd.tmatmul(terms, v), B.T @ v) < tol
    assert relerr(bd.sqmm(terms, np.abs(a)), (B * B) @ np.abs(a)) < tol
    assert relerr(bd.sqcolsums(terms), (B * B).sum(axis=0)) < tol
    # Gram + Newton fit + predictor (fused predictor's fallback: basis at the new rows in HBM)
    y = B @ (0.1 * rng.standard_normal(p)) + 0.05 * rng.standard_normal(n)
    y = (y - y.mean()) / y.std(ddof=1)
    lik = ob.loglik_std(om_d, terms, y, x)
    e2 = math.exp(-2 * lik.para[0])
    G = lik.hess() / e2
    assert relerr(G, B.T @ B) < tol and np.array_equal(G, G.T)
    lp = ob.lpdfvec(lik, ob.logpr_gauss(om_d, terms))
    lp.domarg = False
    lp.optnewton()
    theta_o, H_o = O.fit_newton(bo, terms, y, sigma=float(lik.para[0]))
    xnew = sample_x(rng, 150, kinds)
    pred = ob.predictor(lp)
    pred.update(xnew)
    assert relerr(pred.mean(), O.predict_mean(om_o, terms, theta_o, xnew)) < 1e-6
    # the matrix-free PCG fit and its predictor variance on the same wide terms
    likg = ob.loglik_gauss(om_d, terms, y, x)
    lpg = ob.lpdfvec(ob.logpr_gauss(om_d, terms), likg)
    lpg.optcg(1e-13, 3000)
    predg = ob.predictor(lpg)
    predg.update(xnew)
    assert relerr(predg.mean(), O.predict_mean(om_o, terms, theta_o, xnew)) < 1e-5
    want = O.predict_var_gauss(om_o, terms, lpg.totdiaghess, float(likg.para[0]), xnew)
    assert relerr(predg.var(), want) < tol


def test_chunked_gram_equals_whole_at_headline_width():
    """p = 4096 (528 tile pairs) with the design matrix staged in three ragged row chunks:
    same G as with all rows staged at once, to summation order."""
    import os
    import torch
    from outerbase_amd._lib import call
    from outerbase_amd.driver import HotPath
    hp = HotPath(["mat25"] * 20, 40, 4096, 50_000)
    hp.setup()
    hp.step()
    torch.cuda.synchronize()
    try:
        G1 = hp.G.clone()                       # the Cholesky factor overwrote hp.G: recompute
        call("obhip_gram_dev", hp.basis, hp.t._h, None, G1.data_ptr(), None)
        os.environ["OBHIP_GRAM_CHUNK_ROWS"] = str(64 * 300)
        try:
            call("obhip_basis_rebuild", hp.basis)   # drops the staged matrix
            G2 = torch.empty_like(G1)
            call("obhip_gram_dev", hp.basis, hp.t._h, None, G2.data_ptr(), None)
            torch.cuda.synchronize()
        finally:
            del os.environ["OBHIP_GRAM_CHUNK_ROWS"]
        assert torch.equal(G2, G2.T)
        assert float((G1 - G2).abs().max() / G1.abs().max()) < 1e-13
    finally:
        hp.close()
