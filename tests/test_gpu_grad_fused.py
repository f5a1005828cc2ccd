"""The fused hyper-gradient passes (k_tmm_d3, csrc/kernels_grad.hip) against the oracle and against
the per-hyper-parameter passes they replace (OBHIP_GRAD_D3=0, read per call).

What is contracted: matmul_gradhyp / tmatmul_gradhyp / sqcolsums_gradhyp (prodmmge_, tprodmmge_:
/root/reference/src/linalg.cpp:219-276, 395-471; modandbase.cpp:798-879) and, through them, the
likelihood's gradhyp = yhat_gradhyp^T r (loglik_gauss.cpp:120-127) and lpdfvec's marginal
adjustment -1/2 sum diaghessgradhyp / diaghess (fit.cpp:262-268).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import make_pair, knots_for, sample_x

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


def random_terms(rng, p, d, maxlev, max_nnz):
    t = np.zeros((p, d), dtype=np.int64)
    for k in range(1, p):
        nnz = int(rng.integers(1, max_nnz + 1))
        dims = rng.choice(d, size=nnz, replace=False)
        t[k, dims] = rng.integers(1, maxlev + 1, size=nnz)
    return t


CASES = {
    # one delta column per view-term (mat25: one hyper-parameter per dimension), up to 4 factors
    "mat25_d10_w4": (["mat25"] * 10, 12, 6, 4, 700, 500),
    # two delta columns (mat25pow / mat25ang), up to 6 factors: the 8-slot column lists obfit's
    # eight-dimensional terms take
    "pow_ang_d8_w6": (["mat25pow", "mat25ang"] * 4, 12, 5, 6, 900, 450),
    # mixed: groups of one and of two hyper-parameters per dimension in one term set, 2 factors
    "mixed_d6_w2": (["mat25", "mat25pow", "mat25ang"] * 2, 16, 9, 2, 300, 333),
}


@pytest.mark.parametrize("case", list(CASES))
def test_fused_passes_match_oracle_and_the_passes_they_replace(case, monkeypatch):
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd import _lib
    kinds, m, maxlev, max_nnz, p, n = CASES[case]
    rng = np.random.default_rng(p + n)
    om_o, om_d = make_pair(kinds, knots_for(kinds, m))
    terms = random_terms(rng, p, len(kinds), maxlev, max_nnz)
    x = sample_x(rng, n, kinds)
    a, v = rng.standard_normal(p), rng.standard_normal(n)
    bo = O.OuterBase(om_o, x, dograd=True)
    want = (O.ob_mm_gradhyp(bo, terms, a)[1], O.ob_tmm_gradhyp(bo, terms, v)[1],
            O.ob_sqcolsums_gradhyp(bo, terms))
    got, launches = {}, {}
    for path in ("fused", "views"):
        if path == "views":
            monkeypatch.setenv("OBHIP_GRAD_D3", "0")
        bd = ob.outerbase(om_d, x)
        _lib.call("obhip_profile_reset")
        _lib.call("obhip_profile_enable", 1)
        dot = bd.matmul_gradhyp_dot(terms, a, v)
        got[path] = (bd.matmul_gradhyp(terms, a), bd.tmatmul_gradhyp(terms, v), bd.sqcolsums_gradhyp(terms), dot)
        cnt, ms = C.c_uint64(0), C.c_double(0)
        _lib.call("obhip_profile_get", b"tmm_d3", C.byref(cnt), C.byref(ms))
        _lib.call("obhip_profile_enable", 0)
        launches[path] = cnt.value
    # the fused kernel is what ran (three contractions, at least one group each), and only there
    assert launches["fused"] >= 3 and launches["views"] == 0, launches
    # (levels of 8 and more lose digits in the knot sums on the float64 oracle's side: see
    # test_gradhyp_products_match_oracle)
    tol = 2e-7 if maxlev >= 8 else 1e-9
    for path in got:
        for q in range(3):
            assert relerr(got[path][q], want[q]) < tol, (path, q)
        assert relerr(got[path][3], want[0].T @ v) < tol, path
    # the two device paths sum the same products in another order
    for q in (1, 2):
        assert relerr(got["fused"][q], got["views"][q]) < 1e-12, q


def _lpdf_state(lp):
    return dict(val=float(lp.val), grad=np.array(lp.grad), gradhyp=np.array(lp.gradhyp),
                gradpara=np.array(lp.gradpara))


@pytest.mark.parametrize("kinds", [["mat25pow"] * 4 + ["mat25"] * 2, ["mat25ang", "mat25", "mat25pow"]])
def test_likelihood_gradients_with_the_fused_sweep_and_deferred_adjustment(kinds, monkeypatch):
    """lpdfvec(loglik_gauss, logpr_gauss) with the marginal adjustment on: value, gradient,
    hyper-gradient and parameter gradient (fit.cpp:319-380) through a sequence that exercises the
    caches -- update, update again, new parameters, new hyper-parameters (updateom), new terms --
    fused against per-hyper-parameter passes, and the first state against the oracle."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(len(kinds))
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20))
    n, p = 600, 200
    x, y = O.synth_xy(42, 0, n, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(p)
    terms2 = om_o.selectterms(p + 40)[20:20 + p]
    coeff = 0.05 * rng.standard_normal(p)
    hyp0 = ob.gethyp(om_d)
    sigma, rho = -1.1, 4.0

    def run():
        # (a fresh pair: the first state runs on the oracle's rotation, updatehyp below puts the
        # library's own eigen-model in its place)
        _, om_d = make_pair(kinds, knots_for(kinds, 20))
        lik = ob.loglik_gauss(om_d, terms, y, x)
        pr = ob.logpr_gauss(om_d, terms)
        lp = ob.lpdfvec(lik, pr)
        lp.updatepara([sigma, rho])
        out = []

        def step(with_grad=True):
            lp.compute_gradhyp = lp.compute_gradpara = with_grad
            lp.update(coeff)
            if with_grad:
                out.append(_lpdf_state(lp))
        step()
        step()                                   # nothing changed: the caches answer
        step(with_grad=False)                    # (a value-only update between two gradient ones)
        lp.updatepara([sigma + 0.3, rho - 0.5])
        step()
        om_d.updatehyp(hyp0 + 0.05)
        lp.updateom()
        step(with_grad=False)                    # the rebuild happens in an update without gradients
        step()
        lp.updateterms(terms2)
        step()
        dh = np.array(lp.diaghessgradhyp())      # the accessor forms what is still pending
        return out, dh

    fused, dh_f = run()
    monkeypatch.setenv("OBHIP_GRAD_D3", "0")
    views, dh_v = run()
    assert len(fused) == len(views) == 5
    for s, (f, v) in enumerate(zip(fused, views)):
        assert abs(f["val"] - v["val"]) <= 1e-12 * abs(v["val"]), s
        for key in ("grad", "gradhyp", "gradpara"):
            assert relerr(f[key], v[key]) < 1e-10, (s, key)
    assert relerr(dh_f, dh_v) < 1e-11
    assert relerr(fused[0]["gradhyp"], fused[1]["gradhyp"]) == 0.0
    # the first state against the oracle (same rotation)
    bo = O.OuterBase(om_o, x, dograd=True)
    v0, g0, gh0, gp0 = O.loglik_update(bo, terms, y, sigma, coeff)
    pv, pg, pgh, pgp = O.logpr_update(om_o, terms, rho, coeff)
    mv, mgh, mgp_lik, mgp_pr = O.margadj_diag(bo, terms, sigma, rho)
    want_gh = np.asarray(gh0) + np.asarray(pgh) + np.asarray(mgh)
    assert abs(fused[0]["val"] - (v0 + pv + mv)) < 1e-9 * abs(v0 + pv + mv)
    assert relerr(fused[0]["gradhyp"], want_gh) < 1e-8
    assert relerr(fused[0]["grad"], np.asarray(g0) + np.asarray(pg)) < 1e-10


def test_views_too_wide_for_the_fused_kernel_take_the_older_passes():
    """Terms of 9 factors make a view of 8 other factors + 1 + 1 columns: beyond the 8 column slots
    of k_tmm_d3, so build_d3_groups declines and the per-hyper-parameter passes answer."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25"] * 10
    rng = np.random.default_rng(9)
    om_o, om_d = make_pair(kinds, knots_for(kinds, 10))
    p, n = 120, 200
    terms = random_terms(rng, p, 10, 3, 9)
    terms[1, :9] = 1                               # one term with nine factors for certain
    x = sample_x(rng, n, kinds)
    v = rng.standard_normal(n)
    bo = O.OuterBase(om_o, x, dograd=True)
    bd = ob.outerbase(om_d, x)
    assert relerr(bd.tmatmul_gradhyp(terms, v), O.ob_tmm_gradhyp(bo, terms, v)[1]) < 1e-9
    assert relerr(bd.sqcolsums_gradhyp(terms), O.ob_sqcolsums_gradhyp(bo, terms)) < 1e-9


def test_one_terms_handle_on_bases_of_two_models():
    """The view tables cached in a terms handle hold column numbers of a gradient basis; the same
    handle may meet the basis of another model with as many hyper-parameters laid out differently
    (here 2 + 1 + 1 against 1 + 2 + 1): the caches are keyed by that layout."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(4)
    n, p = 300, 150
    terms = random_terms(rng, p, 3, 4, 3)
    v, a = rng.standard_normal(n), rng.standard_normal(p)
    handle = None
    for kinds in (["mat25pow", "mat25", "mat25"], ["mat25", "mat25pow", "mat25"], ["mat25pow", "mat25", "mat25"]):
        om_o, om_d = make_pair(kinds, knots_for(kinds, 14))
        assert len(om_o.hypmatch) == 4
        if handle is None:
            handle = ob.obmod._Terms(om_d, terms)
        x = sample_x(rng, n, kinds)
        bo = O.OuterBase(om_o, x, dograd=True)
        bd = ob.outerbase(om_d, x)
        assert relerr(bd.tmatmul_gradhyp(handle, v), O.ob_tmm_gradhyp(bo, terms, v)[1]) < 1e-7, kinds     # (a stale table gives errors of order one)
        assert relerr(bd.sqcolsums_gradhyp(handle), O.ob_sqcolsums_gradhyp(bo, terms)) < 1e-7, kinds     # (a stale table gives errors of order one)
        assert relerr(bd.matmul_gradhyp_dot(handle, a, v), O.ob_mm_gradhyp(bo, terms, a)[1].T @ v) < 1e-7, kinds     # (a stale table gives errors of order one)
        assert relerr(bd.matmul_gradhyp(handle, a), O.ob_mm_gradhyp(bo, terms, a)[1]) < 1e-7, kinds     # (a stale table gives errors of order one)
