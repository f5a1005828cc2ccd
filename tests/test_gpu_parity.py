"""GPU parity tests: every kernel behind the C ABI against the CPU oracle on the
same seeded inputs.  All arithmetic is FP64; tolerances are stated per test.
The reference's own identities (tests/testthat/test-obombasic.R) are restated
through the device path in test_reference_identities.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

from conftest import KNOTS_REF, knots_for, make_pair, sample_x

pytestmark = pytest.mark.gpu

MIXED = ["mat25pow", "mat25", "mat25ang", "mat25", "mat25pow", "mat25ang", "mat25", "mat25"]


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


def random_terms(rng, p, d, maxlev, max_nnz):
    """terms with a controlled number of non-zero levels (incl. the constant)."""
    t = np.zeros((p, d), dtype=np.int64)
    for k in range(1, p):
        nnz = int(rng.integers(1, max_nnz + 1))
        dims = rng.choice(d, size=nnz, replace=False)
        t[k, dims] = rng.integers(1, maxlev + 1, size=nnz)
    return t


@pytest.fixture(scope="module")
def mixed_pair():
    return make_pair(MIXED, knots_for(MIXED), hyp=None)


def test_getbase_all_kernels(mixed_pair):
    """covf::cov x rotmat for all three kernels, every level (covfuncs.cpp:113-310,
    modandbase.cpp:285-298,634-639).  The sum over knots cancels catastrophically for
    trailing levels (rotmat ~ 1/lambda), so the tolerance is relative to
    sum_j |k_j| |rot_jc| per column."""
    import ob_oracle as O
    import outerbase_amd as ob
    om_o, om_d = mixed_pair
    rng = np.random.default_rng(1)
    x = sample_x(rng, 100, MIXED)
    bo = O.OuterBase(om_o, x)
    bd = ob.outerbase(om_d, x)
    for k in range(1, len(MIXED) + 1):
        got = bd.getbase(k)
        want = bo.getbase(k)
        o = om_o.knotptst[k - 1]
        m = om_o.knotptst[k] - o
        K = O.cov(MIXED[k - 1], x[:, k - 1], om_o.knots_of(k - 1), om_o.hyp_of(k - 1))
        bound = np.abs(K) @ np.abs(om_o.rotmat[:m, o:o + m])
        assert np.all(np.abs(got - want) <= 1e-13 * bound + 1e-300), "dim %d" % k


@pytest.mark.parametrize("scale_hyp", ["lower bound", -3.3, -3.5])
def test_extreme_length_scales_and_inputs_outside_the_domain(scale_hyp):
    """mat25 / mat25pow evaluate exp(-|t(x) - t_j|) as a product of per-row and per-knot
    exponentials (device_common.h).  At the hyper-parameter lower bound, below it (updatehyp
    accepts that, only the hyper-prior penalises it) and for inputs outside [0, 1] the factors
    must not overflow into inf * 0: centred exponents up to a knot spread of 300, one exp per
    knot beyond.  The oracle evaluates exp(-h) directly like the reference."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25", "mat25pow", "mat25"]
    knots = knots_for(kinds, 20)
    lb = [O.COV_INFO[k]["hyplb"] for k in kinds]
    hyp = np.concatenate([np.asarray(O.COV_INFO[k]["hyp0"], dtype=float) for k in kinds])
    sc = lb[0][0] if scale_hyp == "lower bound" else scale_hyp
    hyp[0] = sc                      # mat25 scale
    hyp[1] = sc                      # mat25pow scale
    om_o, om_d = make_pair(kinds, knots, hyp=hyp)
    rng = np.random.default_rng(12)
    x = sample_x(rng, 200, kinds)
    if scale_hyp != -3.5:
        # outside the kernel's domain (at -3.5 every kernel value of such a row underflows to
        # zero and the reference's own R[:, 1:] / R[:, 0] is 0 / 0, modandbase.cpp:297)
        x[:5, 0] = [-0.02, 1.03, 0.0, 1.0, 1.3]
    x[:3, 2] = [-0.5, 2.0, 1.0001]
    bo = O.OuterBase(om_o, x)
    bd = ob.outerbase(om_d, x)
    for k in (1, 2, 3):
        got, want = bd.getbase(k), bo.getbase(k)
        assert np.all(np.isfinite(got))
        o, m = om_o.knotptst[k - 1], om_o.knotptst[k] - om_o.knotptst[k - 1]
        K = O.cov(kinds[k - 1], x[:, k - 1], om_o.knots_of(k - 1), om_o.hyp_of(k - 1))
        bound = np.abs(K) @ np.abs(om_o.rotmat[:m, o:o + m])
        assert np.all(np.abs(got - want) <= 1e-12 * bound + 1e-290), "dim %d" % k
    terms = om_o.selectterms(40)
    a = rng.standard_normal(40)
    B = O.ob_getmat(bo, terms)
    assert np.all(np.isfinite(bd.matmul(terms, a)))
    assert relerr(bd.matmul(terms, a), B @ a) < 1e-9
    # the fused predictor evaluates the same kernels in LDS
    lik = ob.loglik_gauss(om_d, terms, rng.standard_normal(200), x)
    lik.update(a)
    pred = ob.predictor(lik)
    pred.update(x)
    assert relerr(pred.mean(), B @ a) < 1e-9
    # and the gradient basis (its kernel takes one exp per knot) stays finite and consistent
    g = bd.matmul_gradhyp(terms, a)
    assert np.all(np.isfinite(g))


@pytest.mark.parametrize("hyp_shift", [0.0, 0.3])
def test_getmat_matmul_tmatmul(hyp_shift):
    import ob_oracle as O
    import outerbase_amd as ob
    hyp = None
    if hyp_shift:
        o0, _ = make_pair(MIXED, knots_for(MIXED))
        hyp = o0.hyp + hyp_shift * np.linspace(-1, 1, len(o0.hyp))
    om_o, om_d = make_pair(MIXED, knots_for(MIXED), hyp=hyp)
    rng = np.random.default_rng(2)
    n = 333
    x = sample_x(rng, n, MIXED)
    terms = om_o.selectterms(150)
    assert np.array_equal(terms, om_d.selectterms(150))
    bo = O.OuterBase(om_o, x)
    bd = ob.outerbase(om_d, x)
    B = O.ob_getmat(bo, terms)
    assert relerr(bd.getmat(terms), B) < 1e-11
    a = rng.standard_normal(150)
    v = rng.standard_normal(n)
    assert relerr(bd.matmul(terms, a), O.ob_mm(bo, terms, a)) < 1e-11
    assert relerr(bd.tmatmul(terms, v), O.ob_tmm(bo, terms, v)) < 1e-11
    assert relerr(bd.sqmm(terms, np.abs(a)), O.ob_sqmm(bo, terms, np.abs(a))) < 1e-11
    assert relerr(bd.sqtmm(terms, v), O.ob_sqtmm(bo, terms, v)) < 1e-11
    assert relerr(bd.sqcolsums(terms), O.ob_sqcolsums(bo, terms)) < 1e-11
    assert np.max(np.abs(bd.residvar(terms) - O.ob_residvar(bo, terms))) < 1e-11
    # matrix forms (linalg.cpp:481-637)
    A = rng.standard_normal((150, 3))
    V = rng.standard_normal((n, 2))
    assert relerr(bd.matmul(terms, A), O.ob_mm(bo, terms, A)) < 1e-11
    assert relerr(bd.tmatmul(terms, V), O.ob_tmm(bo, terms, V)) < 1e-11


def test_reference_identities():
    """tests/testthat/test-obombasic.R:21-78 through the device path: n=15, d=8,
    p=20, knots seq(.001,.999,.025), covs mat25pow + 7 x mat25; summed absolute
    differences below 0.01 (the reference's tolerance), plus the intended
    tmatmul identity the reference forgot to check (test-obombasic.R:60)."""
    import outerbase_amd as ob
    d = 8
    kinds = ["mat25pow"] + ["mat25"] * (d - 1)
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, [KNOTS_REF] * d)
    om.updatehyp(ob.gethyp(om))
    rng = np.random.default_rng(42)
    x = rng.random((15, d))
    terms = om.selectterms(20)
    b = ob.outerbase(om, x)
    theta = np.sqrt(om.getvar(terms) / 20) * rng.standard_normal(20)
    viabase = np.ones((15, 20))
    for k in range(1, d + 1):
        viabase *= b.getbase(k)[:, terms[:, k - 1]]
    B = b.getmat(terms)
    assert np.sum(np.abs(B - viabase)) < 0.01
    v1 = B @ theta
    assert np.sum(np.abs(v1 - b.matmul(terms, theta))) < 0.01
    assert np.sum(np.abs(B.T @ v1 - b.tmatmul(terms, v1))) < 0.01
    # and far tighter than the reference asks
    assert relerr(B, viabase) < 1e-10
    assert relerr(b.matmul(terms, theta), v1) < 1e-12


@pytest.mark.parametrize("n,p", [(200, 100), (10000, 100), (200, 1000), (10000, 2000)])
def test_shapes_of_reference_gradient_tests(n, p):
    """the four (n, p) shapes test-obomgrad.R:72-105 uses to hit both OpenMP
    schedules of the reference; here they hit single/multi tile and
    single/multi term-block paths."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25"] * 8
    om_o, om_d = make_pair(kinds, [np.arange(0.001, 0.999, 0.05)] * 8)  # test-lpdf.R:89
    rng = np.random.default_rng(n + p)
    x = rng.random((n, 8))
    terms = om_o.selectterms(p)
    bo = O.OuterBase(om_o, x)
    bd = ob.outerbase(om_d, x, levelcap=terms.max(axis=0))
    a = rng.standard_normal(p)
    v = rng.standard_normal(n)
    # 20 knots per dimension: 1000+ terms reach levels with eigenvalue ratio ~1e-8,
    # i.e. rotmat entries ~1e6, so the sum over knots cancels 6 digits in BOTH
    # implementations; agreement is limited by that conditioning, not by the kernels.
    assert relerr(bd.matmul(terms, a), O.ob_mm(bo, terms, a)) < 1e-7
    assert relerr(bd.tmatmul(terms, v), O.ob_tmm(bo, terms, v)) < 1e-7


@pytest.mark.parametrize("n", [1, 63, 64, 65, 130])
@pytest.mark.parametrize("p", [1, 129, 300])
def test_ragged_sizes(n, p):
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25", "mat25pow", "mat25ang"]
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20))
    rng = np.random.default_rng(7 * n + p)
    x = sample_x(rng, n, kinds)
    terms = om_o.selectterms(p)
    bo = O.OuterBase(om_o, x)
    bd = ob.outerbase(om_d, x)
    a = rng.standard_normal(p)
    v = rng.standard_normal(n)
    # 3 dimensions x 20 knots: 129+ terms climb to levels with rotmat ~1e5
    # (see test_shapes_of_reference_gradient_tests); p = 1 is exact to rounding.
    tol = 1e-12 if p == 1 else 1e-7
    assert relerr(bd.matmul(terms, a), O.ob_mm(bo, terms, a)) < tol
    assert relerr(bd.tmatmul(terms, v), O.ob_tmm(bo, terms, v)) < tol
    if n > 1:
        lik = ob.loglik_std(om_d, terms, v, x)
        G = lik.hess() * math.exp(2 * lik.para[0])
        B = O.ob_getmat(bo, terms)
        assert relerr(G, B.T @ B) < tol
        assert np.array_equal(G, G.T)


@pytest.fixture
def gram_backend(request):
    """3 = fused 4x4x4 matrix-core kernel, 4 = staged-design-matrix 4x4x4 matrix-core
    kernel (0 = automatic = 4, in row chunks when memory is short)."""
    from outerbase_amd import _lib
    _lib.call("obhip_set_gram_backend", request.param)
    yield request.param
    _lib.call("obhip_set_gram_backend", 0)


@pytest.mark.parametrize("gram_backend", [3, 4], indirect=True)
@pytest.mark.parametrize("n,p", [(2, 1), (65, 127), (200, 128), (1000, 129), (5000, 700)])
def test_gram_backends(gram_backend, n, p):
    """both Gram kernels on single/multi tile pairs, ragged edges and multiple
    row splits, against B^T B of the oracle; exact symmetry of the result."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25", "mat25pow", "mat25ang", "mat25"]
    om_o, om_d = make_pair(kinds, knots_for(kinds, 40))
    rng = np.random.default_rng(n + 31 * p)
    x = sample_x(rng, n, kinds)
    terms = om_o.selectterms(p)
    y = rng.standard_normal(n)
    lik = ob.loglik_std(om_d, terms, y, x)
    G = lik.hess() * math.exp(2 * lik.para[0])
    B = O.ob_getmat(O.OuterBase(om_o, x), terms)
    # 700 terms in 4 dimensions reach level ~10 (eigenvalue ratio ~1e-9): both sides
    # lose digits in the knot sum; low-level cases agree to ~1e-13
    assert relerr(G, B.T @ B) < (1e-7 if p > 200 else 1e-11)
    assert np.array_equal(G, G.T)


@pytest.mark.parametrize("gram_backend", [3, 4], indirect=True)
@pytest.mark.parametrize("max_nnz", [1, 2, 3, 5, 6, 8])
def test_gram_term_widths(gram_backend, max_nnz):
    """terms with 1..8 non-zero levels (MFMA kernel template widths W = 2, 4, 6,
    8; runtime width in the vector-pipe kernel)."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25"] * 10
    om_o, om_d = make_pair(kinds, knots_for(kinds, 24))
    rng = np.random.default_rng(max_nnz)
    n, p = 500, 200
    x = sample_x(rng, n, kinds)
    terms = random_terms(rng, p, 10, 6, max_nnz)
    bo = O.OuterBase(om_o, x)
    y = rng.standard_normal(n)
    lik = ob.loglik_std(om_d, terms, y, x)
    G = lik.hess() * math.exp(2 * lik.para[0])
    B = O.ob_getmat(bo, terms)
    # 24 knots, levels up to 6 (eigenvalue ratio ~3e-7): ~1e-10 agreement
    assert relerr(G, B.T @ B) < 2e-9
    a = rng.standard_normal(p)
    assert relerr(lik.ob.matmul(terms, a), B @ a) < 2e-9
    assert relerr(lik.ob.tmatmul(terms, y), B.T @ y) < 2e-9


def test_gram_wide_terms():
    """a term with 9 non-zero levels: the fused kernel (widths up to 8) refuses it
    loudly; the staged-design-matrix kernel (and with it the automatic choice) and the
    matrix-free path take it."""
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd import _lib
    kinds = ["mat25"] * 12
    om_o, om_d = make_pair(kinds, knots_for(kinds, 16))
    rng = np.random.default_rng(3)
    x = sample_x(rng, 100, kinds)
    terms = np.zeros((5, 12), dtype=np.int64)
    terms[1, :9] = 1
    terms[2, 3] = 2
    lik = ob.loglik_std(om_d, terms, rng.standard_normal(100), x)
    B = O.ob_getmat(O.OuterBase(om_o, x), terms)
    try:
        _lib.call("obhip_set_gram_backend", 3)
        with pytest.raises(ob.ObhipError):
            lik.hess()
        for backend in (1, 2, 5):            # the pruned generations are gone from the ABI
            with pytest.raises(ob.ObhipError):
                _lib.call("obhip_set_gram_backend", backend)
        for backend in (0, 4):
            _lib.call("obhip_set_gram_backend", backend)
            G = lik.hess() * math.exp(2 * lik.para[0])
            assert relerr(G, B.T @ B) < 1e-11
    finally:
        _lib.call("obhip_set_gram_backend", 0)
    # the matrix-free path (and with it the CG fit) has no such limit
    assert relerr(lik.ob.matmul(terms, np.ones(5)), B @ np.ones(5)) < 1e-11
    v = rng.standard_normal(100)
    assert relerr(lik.ob.tmatmul(terms, v), B.T @ v) < 1e-11


def _fit_case(rng, kinds, n, p, m=40):
    import ob_oracle as O
    om_o, om_d = make_pair(kinds, knots_for(kinds, m))
    x, y = O.synth_xy(42, 0, n, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    xnew, _ = O.synth_xy(43, 0, 257, kinds)
    terms = om_o.selectterms(p)
    return om_o, om_d, x, y, xnew, terms


@pytest.mark.parametrize("kinds,n,p", [
    (["mat25pow"] * 8, 1000, 256),                      # BASELINE config 1
    (["mat25"] * 10, 3000, 300),
    (["mat25", "mat25pow", "mat25ang"] * 3, 2000, 200),
])
def test_newton_fit_and_predict(kinds, n, p):
    """optnewton (fit.cpp:98-131) + predictor (loglik_gauss.cpp:214-227):
    predictions within 1e-6 relative of the oracle (north_star tolerance)."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(5)
    om_o, om_d, x, y, xnew, terms = _fit_case(rng, kinds, n, p)
    bo = O.OuterBase(om_o, x)
    theta_o, H_o = O.fit_newton(bo, terms, y)
    lik = ob.loglik_std(om_d, terms, y, x)
    pr = ob.logpr_gauss(om_d, terms)
    lp = ob.lpdfvec(lik, pr)
    lp.optnewton()
    # Hessian itself
    assert relerr(lp.hess(), H_o) < 1e-10
    # predictions at new inputs and at the training inputs
    pred = ob.predictor(lp)
    pred.update(xnew)
    mean_o = O.predict_mean(om_o, terms, theta_o, xnew)
    assert relerr(pred.mean(), mean_o) < 1e-6
    assert relerr(lik.yhat, O.ob_mm(bo, terms, theta_o)) < 1e-6
    # predr_std variance with the full posterior covariance (loglik_std.cpp:249-256)
    var_o = O.predict_var_std(om_o, terms, H_o, O.default_sigma(y), xnew)
    assert relerr(pred.var(), var_o) < 1e-7
    # Newton stationarity: H theta = e^{-2 sigma} B^T y
    rhs = math.exp(-2 * O.default_sigma(y)) * O.ob_tmm(bo, terms, y)
    assert relerr(H_o @ lp.coeff, rhs) < 1e-7


def test_cg_fit_matches_oracle_iterations():
    """optcg (fit.cpp:37-96): same iteration count and iterate as the oracle at
    the reference's iteration cap (.getsteps, R/fitting.R:188-195); and at tight
    tolerance the CG solution converges to the Newton one."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25"] * 8
    rng = np.random.default_rng(6)
    om_o, om_d, x, y, xnew, terms = _fit_case(rng, kinds, 2000, 150)
    bo = O.OuterBase(om_o, x)
    steps = O.getsteps(150, 2000, O.rvar(y) / math.exp(2 * O.default_sigma(y)))
    th_o, it_o, m_o = O.fit_cg(bo, terms, y, tol=1e-3, maxit=steps)
    lik = ob.loglik_gauss(om_d, terms, y, x)
    pr = ob.logpr_gauss(om_d, terms)
    lp = ob.lpdfvec(pr, lik)
    lp.optcg(1e-3, steps)
    assert lp.cgiters == it_o
    assert relerr(lp.totdiaghess, m_o) < 1e-10
    assert relerr(lik.yhat, O.ob_mm(bo, terms, th_o)) < 1e-6
    # tight CG == Newton on predictions
    lp.optcg(1e-14, 2000)
    th_n, _ = O.fit_newton(bo, terms, y)
    assert relerr(lik.yhat, O.ob_mm(bo, terms, th_n)) < 1e-6


@pytest.mark.parametrize("m,cap", [(120, None), (120, 3), (40, None)])
def test_many_knots_with_and_without_interval_tables(m, cap):
    """The interval tables of mat25 / mat25pow are built only while a dimension's table fits the
    LDS buffer of k_build_basis (2048 doubles; round-3 advice: larger ones cost the host
    O(knots^2 x levels) per hyper-parameter update and were read by per-lane global gathers, no
    faster than the knot loop).  120 knots with every level kept: the knot loop; capped at 3
    levels: tables of 121 intervals; 40 knots with every level: the knot loop again.  The
    well-separated low levels against the oracle either way."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25", "mat25pow", "mat25"]
    knots = [np.linspace(0.001, 0.976, m)] * 3
    om_o, om_d = make_pair(kinds, knots)
    rng = np.random.default_rng(m)
    x = sample_x(rng, 500, kinds)
    terms = random_terms(rng, 60, 3, 3, 3)
    levelcap = None if cap is None else np.full(3, cap, dtype=np.int64)
    bd = ob.outerbase(om_d, x, levelcap=levelcap)
    want = O.ob_getmat(O.OuterBase(om_o, x), terms)
    assert relerr(bd.getmat(terms), want) < 1e-9
    a = rng.standard_normal(60)
    assert relerr(bd.matmul(terms, a), want @ a) < 1e-9


def test_level_caps_do_not_change_results():
    import outerbase_amd as ob
    kinds = ["mat25"] * 6
    _, om_d = make_pair(kinds, knots_for(kinds, 30))
    rng = np.random.default_rng(8)
    x = sample_x(rng, 400, kinds)
    terms = om_d.selectterms(120)
    full = ob.outerbase(om_d, x)
    capped = ob.outerbase(om_d, x, levelcap=terms.max(axis=0))
    a = rng.standard_normal(120)
    # (to rounding, not to the bit: with all 30 levels kept a dimension's interval tables no longer
    # fit k_build_basis's LDS buffer and it takes the knot loop, the capped basis takes the tables)
    assert relerr(capped.matmul(terms, a), full.matmul(terms, a)) < 1e-10
    with pytest.raises(ob.ObhipError):
        ob.outerbase(om_d, x, levelcap=np.zeros(6, dtype=np.int64)).matmul(terms, a)


def test_rebuild_after_hyp_change():
    """vignettes/learning.Rmd:48-54: an outerbase does not follow the outermod
    until build() is called again."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25", "mat25pow", "mat25"]
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20), share_rotation=False)
    rng = np.random.default_rng(9)
    x = sample_x(rng, 100, kinds)
    terms = om_d.selectterms(30)
    b = ob.outerbase(om_d, x)
    before = b.getmat(terms)
    om_d.updatehyp(ob.gethyp(om_d) + 0.2)
    assert np.array_equal(b.getmat(terms), before)
    b.build()
    after = b.getmat(terms)
    assert relerr(after, before) > 1e-3
    om_o.hyp_set(om_o.hyp + 0.2)
    om_d.set_rotation(om_o.rotmat, om_o.basisvar, om_o.maxlevel)
    b.build()
    assert relerr(b.getmat(terms), O.ob_getmat(O.OuterBase(om_o, x), terms)) < 1e-10


def test_own_eigensolver_low_levels():
    """Without sharing the rotation: the library's Jacobi eigen-solver against
    LAPACK on the levels term selection actually uses (well separated
    eigenvalues); trailing levels are numerically undetermined in both."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25"] * 5
    om_o, om_d = make_pair(kinds, knots_for(kinds, 40), share_rotation=False)
    rng = np.random.default_rng(10)
    x = sample_x(rng, 200, kinds)
    terms = om_o.selectterms(300)
    assert np.array_equal(terms, om_d.selectterms(300))
    assert terms.max() <= 12
    got = ob.outerbase(om_d, x, levelcap=terms.max(axis=0)).getmat(terms)
    want = O.ob_getmat(O.OuterBase(om_o, x), terms)
    assert relerr(got, want) < 1e-6


def test_synthetic_generator_matches_oracle():
    import ctypes as C
    import torch
    import ob_oracle as O
    from outerbase_amd import _lib
    kinds = ["mat25", "mat25ang", "mat25pow"] * 4
    kid = {"mat25": 0, "mat25pow": 1, "mat25ang": 2}
    n, d = 1000, len(kinds)
    x = torch.empty((d, n), dtype=torch.float64, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    arr = (C.c_int * d)(*[kid[k] for k in kinds])
    _lib.call("obhip_synth_xy_dev", 42, 5000, n, d, C.cast(arr, C.c_void_p), x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    xo, yo = O.synth_xy(42, 5000, n, kinds)
    assert np.array_equal(x.cpu().numpy().T, xo)
    assert relerr(y.cpu().numpy(), yo) < 1e-13


# ---- hyper-parameter gradients (tests/testthat/test-obomgrad.R) ------------
@pytest.mark.parametrize("kinds,n,p", [
    (["mat25pow"] + ["mat25"] * 7, 200, 100),          # test-obomgrad.R "short, skinny"
    (["mat25", "mat25pow", "mat25ang"], 333, 150),     # every covariance, ragged n
    (["mat25"] * 5, 1000, 700),                        # wide
])
def test_gradhyp_products_match_oracle(kinds, n, p):
    """getmat_gradhyp / matmul_gradhyp / tmatmul_gradhyp (getmge_, prodmmge_, tprodmmge_:
    linalg.cpp:778-822, 219-276, 395-471) against the oracle on the same rotation."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(n + p)
    hyp = None
    om_o, om_d = make_pair(kinds, knots_for(kinds, 24), hyp=hyp)
    x = sample_x(rng, n, kinds)
    terms = om_o.selectterms(p)
    bo = O.OuterBase(om_o, x, dograd=True)
    bd = ob.outerbase(om_d, x)
    a = rng.standard_normal(p)
    v = rng.standard_normal(n)
    mean_o, mge_o = O.ob_mm_gradhyp(bo, terms, a)
    g_o, gge_o = O.ob_tmm_gradhyp(bo, terms, v)
    # levels of ~8 and more (eigenvalue ratio ~1e-8 at 24 knots) lose digits in the knot
    # sums on both sides (see test_gram_backends); low-level cases agree to ~1e-12.  Since
    # round 4 the device takes those sums from interval tables built in extended precision
    # (0.15 eps x the conditioning bound against 0.5 eps for a knot loop:
    # test_gradient_basis_tables_and_knot_loop_against_extended_precision), so what is left
    # at high levels is the float64 oracle's own loss: 1.3e-7 here
    tol = 2e-7 if terms.max() >= 8 else 1e-9
    assert relerr(bd.matmul_gradhyp(terms, a), mge_o) < tol
    assert relerr(bd.tmatmul_gradhyp(terms, v), gge_o) < tol
    if n * p <= 60000:
        assert relerr(bd.getmat_gradhyp(terms), O.ob_getmat_gradhyp(bo, terms)) < tol


@pytest.mark.parametrize("kind", ["mat25", "mat25pow"])
def test_gradient_basis_tables_and_knot_loop_against_extended_precision(kind, monkeypatch):
    """The gradient basis of a dimension (outermod::buildob with gradients, modandbase.cpp:306-327)
    from the interval tables (k_build_basis_grad_tab) and from the knot loop (k_build_basis_grad,
    OBHIP_GRAD_KNOTLOOP=1), each against the same sums taken in extended precision on the host --
    the float64 oracle loses as many digits at high levels as the knot loop does, so it cannot
    tell which of the two is right.  A one-dimensional model makes the raw columns visible:
    getmat = cov . rotmat, getmat_gradhyp = dcov . rotmat + cov . rotmat_gradhyp.  Errors are
    measured against the conditioning of the knot sums, sum_j |k_j| |rot_jc| per entry."""
    import ob_oracle as O
    import outerbase_amd as ob
    ld = np.longdouble
    knots = [0.001 + 0.025 * np.arange(40)]
    om_o, om_d = make_pair([kind], knots)
    nlev = 16
    terms = np.arange(nlev, dtype=np.int64)[:, None]
    rng = np.random.default_rng(5)
    x = sample_x(rng, 400, [kind])
    rot, _, _ = om_d.rotation()
    rotg, _ = om_d.rotation_grad()
    hyp = np.asarray(ob.gethyp(om_d), dtype=ld)
    kn, xv = knots[0].astype(ld), x[:, 0].astype(ld)
    rot = rot[:, :nlev].astype(ld)
    nh = len(hyp)
    rotg = [rotg[:, h * 40:h * 40 + nlev].astype(ld) for h in range(nh)]
    if kind == "mat25":
        t1, t2 = xv / np.exp(2 * hyp[0]), kn / np.exp(2 * hyp[0])
    else:
        powv, els = np.exp(ld(0.25) * hyp[1]), np.exp(2 * hyp[0] + ld(0.25) * hyp[1])
        t1, t2 = np.power(xv, powv) / els, np.power(kn, powv) / els
    h = t1[:, None] - t2[None, :]
    ah = np.abs(h)
    e = np.exp(-ah)
    K = (1 + ah + ah * ah / 3) * e
    h2 = h * (1 + ah) * e
    dK = [ld(2) / 3 * h * h2]
    if kind == "mat25pow":
        g1 = (np.log(xv) * t1)[:, None] - (np.log(kn) * t2)[None, :]
        dK.append(g1 * (-(ld(0.25) * powv / 3) * h2) + ld(0.25) / 3 * h * h2)
    R = K @ rot
    bound_R = np.abs(K) @ np.abs(rot)
    Rt = [dK[q] @ rot + K @ rotg[q] for q in range(nh)]
    bound_Rt = [np.abs(dK[q]) @ np.abs(rot) + np.abs(K) @ np.abs(rotg[q]) for q in range(nh)]
    worst = {}
    for path in ("tables", "knot loop"):
        if path == "knot loop":
            monkeypatch.setenv("OBHIP_GRAD_KNOTLOOP", "1")
        b = ob.outerbase(om_d, x, levelcap=np.array([nlev - 1]))
        got_R = b.getmat(terms).astype(ld)
        got = b.getmat_gradhyp(terms)            # n x p x nhyp
        err = [float(np.max(np.abs(got_R - R) / bound_R))]
        err += [float(np.max(np.abs(got[:, :, q].astype(ld) - Rt[q]) / bound_Rt[q])) for q in range(nh)]
        worst[path] = err
    print(kind, "error / conditioning bound (value, d/dhyp...):", worst)
    for path, err in worst.items():
        assert max(err) < 2e-13, (path, err)


@pytest.mark.parametrize("kinds,nhyp", [
    (["mat25"] * 18 + ["mat25pow", "mat25ang", "mat25pow", "mat25ang"], 26),   # 16 + 10: two 16-blocks
    (["mat25"] * 20, 20),                       # 16 + 4: one group of four rides along (the headline's d)
    (["mat25"] * 16 + ["mat25pow"] * 3, 22),    # 16 + 6: two groups of four ride along
    (["mat25pow"] * 3, 6),                      # groups of four only
    (["mat25"] * 3, 3),
])
def test_gradhyp_with_more_hyperparameters_than_one_pass_holds(kinds, nhyp):
    """More hyper-parameters than one 16-block of the dense pass holds, and the shapes at which the
    last few are contracted in groups of four on the 4 x 4 x 4 matrix instruction (k_tmm_ge0_db,
    csrc/kernels_grad.hip): matmul / tmatmul / sqcolsums _gradhyp against the oracle."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(26)
    om_o, om_d = make_pair(kinds, knots_for(kinds, 16))
    assert len(om_o.hypmatch) == nhyp
    n, p = 300, 260
    x = sample_x(rng, n, kinds)
    terms = om_o.selectterms(p)
    bo = O.OuterBase(om_o, x, dograd=True)
    bd = ob.outerbase(om_d, x)
    a, v = rng.standard_normal(p), rng.standard_normal(n)
    tol = 1e-9 if terms.max() < 8 else 2e-7
    assert relerr(bd.matmul_gradhyp(terms, a), O.ob_mm_gradhyp(bo, terms, a)[1]) < tol
    assert relerr(bd.tmatmul_gradhyp(terms, v), O.ob_tmm_gradhyp(bo, terms, v)[1]) < tol
    assert relerr(bd.sqcolsums_gradhyp(terms), O.ob_sqcolsums_gradhyp(bo, terms)) < tol


@pytest.mark.parametrize("ss,nterms", [(400, 100),        # testmultgrad's defaults
                                        (200, 100),        # "short, skinny" (test-obomgrad.R:72)
                                        (10000, 100),      # "tall, skinny"  (:81)
                                        (200, 1000),       # "short, wide"   (:90)
                                        (10000, 2000)])    # "tall, wide"    (:99)
def test_gradhyp_follows_the_reference_finite_difference_test(ss, nterms):
    """testmultgrad of test-obomgrad.R:21-67 on the device at the reference's four shapes: the
    gradient along a random direction equals the difference quotient of two rebuilt bases
    (updatehyp + build).  Central differences and 1e-3 of the largest entry here; the
    reference takes a forward difference and accepts a relative difference of 1."""
    import outerbase_amd as ob
    from conftest import KNOTS_REF
    d = 8
    kinds = ["mat25pow"] + ["mat25"] * (d - 1)
    rng = np.random.default_rng(42)
    x = rng.random((ss, d))
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, [KNOTS_REF] * d)
    hyp0 = ob.gethyp(om)
    terms = om.selectterms(nterms)
    obp = ob.outerbase(om, x)
    theta = 0.01 * rng.standard_normal(nterms)
    y = rng.standard_normal(ss)
    mge = obp.matmul_gradhyp(terms, theta)
    gge = obp.tmatmul_gradhyp(terms, y)
    # (eps: the eigenvectors of the rebuilt bases carry ~1e-9 of solver noise at the high
    # levels wide term sets reach, so the quotient is only good to ~1e-9 / eps)
    eps, hypp = 1e-4, rng.random(len(hyp0)) - 0.5
    vals = []
    for sgn in (1.0, -1.0):
        om.updatehyp(hyp0 + sgn * eps * hypp)
        with pytest.raises(ob.ObhipError):
            obp.matmul_gradhyp(terms, theta)      # stale basis: must be rebuilt first
        obp.build()
        vals.append((obp.matmul(terms, theta), obp.tmatmul(terms, y)))
    fd_m = (vals[0][0] - vals[1][0]) / (2 * eps)
    fd_t = (vals[0][1] - vals[1][1]) / (2 * eps)
    assert np.max(np.abs(fd_m - mge @ hypp)) < 1e-3 * np.max(np.abs(fd_m))
    assert np.max(np.abs(fd_t - gge @ hypp)) < 1e-3 * np.max(np.abs(fd_t))


def test_lpdf_gradhyp_matches_oracle():
    """lpdfvec(loglik_gauss, logpr_gauss)$update with compute_gradhyp / compute_gradpara
    (fit.cpp:319-352, loglik_gauss.cpp:110-130, logpr_gauss.cpp:98-108) and om$hyplpdf_grad."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25pow", "mat25", "mat25ang", "mat25"]
    rng = np.random.default_rng(17)
    om_o, om_d = make_pair(kinds, knots_for(kinds, 24))
    x, y = O.synth_xy(42, 0, 500, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(120)
    coeff = 0.05 * rng.standard_normal(120)
    lik = ob.loglik_gauss(om_d, terms, y, x)
    pr = ob.logpr_gauss(om_d, terms)
    lp = ob.lpdfvec(lik, pr)
    lp.domarg = False                     # the adjustment has its own test below
    lp.compute_gradhyp = lp.compute_gradpara = True
    lp.updatepara([-1.3, 5.0])
    lp.update(coeff)
    bo = O.OuterBase(om_o, x, dograd=True)
    v, g, gh, gp = O.loglik_update(bo, terms, y, -1.3, coeff)
    pv, pg, pgh, pgp = O.logpr_update(om_o, terms, 5.0, coeff)
    assert abs(lp.val - (v + pv)) < 1e-10 * abs(v + pv)
    assert relerr(lp.grad, g + pg) < 1e-10
    assert relerr(lp.gradhyp, gh + pgh) < 1e-8
    assert relerr(lp.gradpara, np.concatenate([gp, pgp])) < 1e-11
    hyp = ob.gethyp(om_d) + 0.1
    assert np.allclose(om_d.hyplpdf_grad(hyp), om_o.hyplpdf_grad(hyp), rtol=1e-14)


def test_squared_store_gradients_match_oracle():
    """ob$sqmm_gradhyp, sqtmm_gradhyp, sqcolsums_gradhyp, residvar_gradhyp
    (modandbase.cpp:798-809, 845-856, 875-879, 904-925) and the lpdfs' diaghessgradhyp."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25", "mat25pow", "mat25ang", "mat25"]
    rng = np.random.default_rng(23)
    om_o, om_d = make_pair(kinds, knots_for(kinds, 24))
    n, p = 260, 90
    x = sample_x(rng, n, kinds)
    terms = om_o.selectterms(p)
    a, v = np.abs(rng.standard_normal(p)), rng.standard_normal(n)
    bo = O.OuterBase(om_o, x, dograd=True)
    bd = ob.outerbase(om_d, x)
    assert relerr(bd.sqmm_gradhyp(terms, a), O.ob_sqmm_gradhyp(bo, terms, a)) < 1e-9
    assert relerr(bd.sqtmm_gradhyp(terms, v), O.ob_sqtmm_gradhyp(bo, terms, v)) < 1e-9
    assert relerr(bd.sqcolsums_gradhyp(terms), O.ob_sqcolsums_gradhyp(bo, terms)) < 1e-9
    assert relerr(bd.residvar_gradhyp(terms), O.ob_residvar_gradhyp(bo, terms)) < 1e-9
    y = rng.standard_normal(n)
    lik = ob.loglik_gauss(om_d, terms, y, x)
    s = float(lik.para[0])
    assert relerr(lik.diaghessgradhyp(), math.exp(-2 * s) * O.ob_sqcolsums_gradhyp(bo, terms)) < 1e-9
    pr = ob.logpr_gauss(om_d, terms)
    want = -om_o.getlvar_gradhyp(terms) / (om_o.getvar(terms) * math.exp(2 * 6.0))[:, None]
    assert relerr(pr.diaghessgradhyp(), want) < 1e-12


def test_lpdfvec_marginal_adjustment_diagonal_form():
    """logpdf$domarg = TRUE as in obfit (R/fitting.R:108-110): val / gradhyp / gradpara of
    lpdfvec(logpr_gauss, loglik_gauss) with the diagonal marginal adjustment
    (fit.cpp:252-268, 371-380) against the oracle, and the hyp gradient against a
    difference quotient of val through updatehyp + updateom."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25pow", "mat25", "mat25"]
    rng = np.random.default_rng(31)
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20))
    x, y = O.synth_xy(42, 0, 400, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(40)          # levels <= 7: val stays smooth enough in hyp
    coeff = 0.05 * rng.standard_normal(40)
    pr = ob.logpr_gauss(om_d, terms)
    lik = ob.loglik_gauss(om_d, terms, y, x)
    lp = ob.lpdfvec(pr, lik)                      # obfit's order: prior first
    lp.domarg = True
    lp.compute_gradhyp = lp.compute_gradpara = True
    rho, sigma = 4.5, -1.1
    lp.updatepara([rho, sigma])
    lp.update(coeff)
    bo = O.OuterBase(om_o, x, dograd=True)
    v, g, gh, gp = O.loglik_update(bo, terms, y, sigma, coeff)
    pv, pg, pgh, pgp = O.logpr_update(om_o, terms, rho, coeff)
    mv, mgh, mgp_lik, mgp_pr = O.margadj_diag(bo, terms, sigma, rho)
    assert abs(lp.val - (v + pv + mv)) < 1e-10 * abs(v + pv + mv)
    assert relerr(lp.gradhyp, gh + pgh + mgh) < 1e-8
    assert relerr(lp.gradpara, np.array([pgp[0] + mgp_pr, gp[0] + mgp_lik])) < 1e-10
    # difference quotient on the device (no shared rotation: own eigen-model throughout)
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, knots_for(kinds, 20))
    hyp0 = ob.gethyp(om)
    pr2, lik2 = ob.logpr_gauss(om, terms), ob.loglik_gauss(om, terms, y, x)
    lp2 = ob.lpdfvec(pr2, lik2)
    lp2.domarg = True
    lp2.compute_gradhyp = True
    lp2.updatepara([rho, sigma])
    lp2.update(coeff)
    gh0 = lp2.gradhyp.copy()
    # val carries ~1e-9 relative rounding noise from the high-level basis columns, so the
    # quotient needs a step of 1e-4 (CPU oracle: 2e-5 agreement at 1e-4, 1e-3 at 1e-5)
    eps, hd = 1e-4, rng.random(len(hyp0)) - 0.5
    vals = []
    for sgn in (1.0, -1.0):
        om.updatehyp(hyp0 + sgn * eps * hd)
        lp2.updateom()
        lp2.update(coeff)
        vals.append(lp2.val)
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - gh0 @ hd) < 1e-3 * abs(fd)


def test_loglik_gda_matches_oracle():
    """loglik_gda (src/lpdfs/loglik_gda.cpp:117-235) composed from the device products: val,
    grad, gradhyp, gradpara and the diaghess family against the oracle; the generic optcg
    on lpdfvec(logpr, loglik_gda) reaches a stationary point of the combined objective."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25pow", "mat25", "mat25"]
    rng = np.random.default_rng(2)
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20))
    x, y = O.synth_xy(42, 0, 300, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(30)
    coeff = 0.05 * rng.standard_normal(30)
    para = np.array([-1.2, 0.3])
    lik = ob.loglik_gda(om_d, terms, y, x)
    assert np.allclose(lik.para0, [0.5 * math.log(0.01 * np.var(y, ddof=1)), 0.0])
    lik.updatepara(para)
    lik.compute_gradhyp = lik.compute_gradpara = True
    lik.update(coeff)
    r = O.loglik_gda_update(O.OuterBase(om_o, x, dograd=True), terms, y, para, coeff)
    assert abs(lik.val - r["val"]) < 1e-10 * abs(r["val"])
    assert relerr(lik.grad, r["grad"]) < 1e-9
    assert relerr(lik.gradhyp, r["gradhyp"]) < 1e-8
    assert relerr(lik.gradpara, r["gradpara"]) < 1e-10
    assert relerr(lik.diaghess(), r["diaghess"]) < 1e-10
    assert relerr(lik.diaghessgradhyp(), r["diaghessgradhyp"]) < 1e-8
    assert relerr(lik.diaghessgradpara(), r["diaghessgradpara"]) < 1e-10
    pr = ob.logpr_gauss(om_d, terms)
    lp = ob.lpdfvec(pr, lik)
    lp.optcg(1e-12, 500)
    assert np.max(np.abs(lp.grad)) < 1e-5 * np.max(np.abs(lik.ob.tmatmul(terms, y)))
    assert lp.gradhyp.shape == (4,) and lp.gradpara.shape == (3,)


@pytest.mark.parametrize("n", [1, 2, 7, 200, 100003])
def test_device_quantiles_are_r_type_7(n):
    """.genknotlist (R/fitting.R:177-185): quantile(x, probs), R's default type 7 -- exact
    order statistics by bisection on counts (obhip_quantiles_dev), against the oracle's
    restatement and numpy's 'linear' method, ties, negative values and +-0 included."""
    import ob_oracle as O
    from outerbase_amd import fitting as F
    rng = np.random.default_rng(n)
    x = rng.random((n, 4))
    x[:, 1] = np.round(x[:, 1], 1)                  # heavy ties
    x[:, 2] = rng.standard_normal(n) * 1e3          # negative values, wide range
    x[::3, 3] = 0.0
    x[1::3, 3] = -0.0
    if n >= 200:
        kn = F._genknotlist([40, 16, 70, 5], x)
        ref = O.genknotlist([40, 16, 70, 5], x)
        assert all(np.array_equal(a, b) for a, b in zip(kn, ref))
    dx = F._DeviceCopy(x)
    probs = np.array([0.0, 1e-9, 0.1, 0.25, 1 / 3, 0.5, 0.9, 1 - 1e-12, 1.0])
    got = F._quantiles(dx, probs, None)
    dx.close()
    want = np.stack([O.quantile7(x[:, k], probs) for k in range(4)])
    assert np.array_equal(got, want)
    lin = np.stack([np.quantile(x[:, k], probs) for k in range(4)])
    assert np.allclose(got, lin, rtol=1e-14, atol=1e-300)


def test_obfit_and_obpred_end_to_end():
    """obfit / obpred (R/fitting.R:27-155) on the Borehole function (R/testfuncs.R:32-46) as
    the package's own examples use it: hyper-parameters move, predictions on fresh points
    are far better than the mean predictor, variances are positive."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(42)
    x = rng.random((400, 8))
    y = O.borehole8d(x)
    obmodel = ob.obfit(x, y, numb=100, seed=1)
    xt = rng.random((200, 8))
    pred = ob.obpred(obmodel, xt)
    yt = O.borehole8d(xt)
    rmse = math.sqrt(np.mean((pred["mean"] - yt) ** 2))
    assert rmse < 0.05 * np.std(yt)
    assert np.all(pred["var"] > 0)
    assert np.max(np.abs(ob.gethyp(obmodel["om"]))) > 1e-3     # moved away from the defaults
    # standardised residuals are of order one (var is a calibrated scale, not decoration)
    z = (pred["mean"] - yt) / np.sqrt(pred["var"])
    assert 0.05 < np.sqrt(np.mean(z ** 2)) < 20


def test_predr_std_full_posterior_covariance():
    """predictor of lpdfvec(loglik_std, logpr_gauss) after optnewton = predr_std with
    coeffcov = inv(tothess): var = rowsum((B C) % B) + e^{2 sigma} (loglik_std.cpp:218-256),
    against the oracle; without the full Hessian the reference's diagonal fallback."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25pow", "mat25", "mat25ang", "mat25"]
    rng = np.random.default_rng(77)
    om_o, om_d = make_pair(kinds, knots_for(kinds, 24))
    x, y = O.synth_xy(42, 0, 700, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(130)
    lik = ob.loglik_std(om_d, terms, y, x)
    lp = ob.lpdfvec(lik, ob.logpr_gauss(om_d, terms))
    lp.optnewton()
    xnew, _ = O.synth_xy(43, 0, 333, kinds)
    pred = ob.predictor(lp)
    pred.update(xnew)
    bo = O.OuterBase(om_o, x)
    sigma = float(lik.para[0])
    theta_o, H_o = O.fit_newton(bo, terms, y, sigma=sigma)
    assert relerr(pred.mean(), O.predict_mean(om_o, terms, theta_o, xnew)) < 1e-9
    want = O.predict_var_std(om_o, terms, H_o, sigma, xnew)
    assert relerr(pred.var(), want) < 1e-8
    assert np.all(pred.var() > math.exp(2 * sigma))


def test_pred_gda_adds_the_residual_variance():
    """pred_gda::var (loglik_gda.cpp:276-281) = B^2 coeffvar + e^{2 para0} + e^{2 para1} residvar."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25", "mat25pow", "mat25"]
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20))
    x, y = O.synth_xy(42, 0, 300, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(40)
    lik = ob.loglik_gda(om_d, terms, y, x)
    lp = ob.lpdfvec(ob.logpr_gauss(om_d, terms), lik)
    lp.optcg(1e-10, 200)
    pred = ob.predictor(lik)
    xnew, _ = O.synth_xy(43, 0, 111, kinds)
    pred.update(xnew)
    bn = O.OuterBase(om_o, xnew)
    cv = 1.0 / lp.totdiaghess
    want = O.ob_sqmm(bn, terms, cv) + math.exp(2 * lik.para[0]) + \
        math.exp(2 * lik.para[1]) * O.ob_residvar(bn, terms)
    assert relerr(pred.var(), want) < 1e-9
    assert relerr(pred.mean(), O.ob_mm(bn, terms, lp.coeff)) < 1e-9


def test_full_hessian_marginal_adjustment_after_optnewton():
    """optnewton on lpdfvec(loglik_std, logpr_gauss) with the marginal adjustment on (the
    default): val, gradhyp and gradpara include -1/2 log det H and -1/2 tr(inv(H) dH)
    (fit.cpp:122-128, 270-299), against the oracle."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25pow", "mat25", "mat25ang"]
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20))
    x, y = O.synth_xy(42, 0, 350, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(45)
    lik = ob.loglik_std(om_d, terms, y, x)
    pr = ob.logpr_gauss(om_d, terms)
    lp = ob.lpdfvec(lik, pr)
    assert lp.domarg
    lp.optnewton()
    sigma, rho = float(lik.para[0]), float(pr.para[0])
    bo = O.OuterBase(om_o, x, dograd=True)
    theta, _ = O.fit_newton(bo, terms, y, sigma=sigma, rho=rho)
    v, g, gh, gp = O.loglik_update(bo, terms, y, sigma, theta)
    pv, pg, pgh, pgp = O.logpr_update(om_o, terms, rho, theta)
    mv, mgh, mgl, mgp = O.margadj_full(bo, terms, sigma, rho)
    assert abs(lp.val - (v + pv + mv)) < 1e-9 * abs(v + pv + mv)
    assert relerr(lp.gradhyp, gh + pgh + mgh) < 1e-7
    assert relerr(lp.gradpara, np.array([gp[0] + mgl, pgp[0] + mgp])) < 1e-8


@pytest.mark.parametrize("p", [1, 2, 63, 64, 65, 127, 128, 129, 1000, 3000, 4096, 4097, 4160, 8192, 8257, 8385])
def test_newton_solve_sizes_against_library_solve(p):
    """obhip_newton_solve_dev (Cholesky + two triangular solves, fit.cpp:120) on random SPD
    systems of awkward sizes (single element, one block, ragged blocks, tile edges)
    against torch.linalg.solve on the same H."""
    import torch
    import outerbase_amd as ob
    from outerbase_amd._lib import call
    kinds = ["mat25"] * 6
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, knots_for(kinds, 40))
    terms = om.selectterms(p)
    t = ob.obmod._Terms(om, terms)
    torch.manual_seed(p)
    A = torch.randn((p, p + 3), dtype=torch.float64, device="cuda")
    G = A @ A.T + 0.5 * torch.eye(p, dtype=torch.float64, device="cuda")
    g = torch.randn(p, dtype=torch.float64, device="cuda")
    sigma, rho = 0.3, 2.0
    e2 = math.exp(-2 * sigma)
    prec = torch.from_numpy(1.0 / (om.getvar(terms) * math.exp(2 * rho))).cuda()
    H = e2 * G + torch.diag(prec)
    want = torch.linalg.solve(H, e2 * g)
    wsb = C.c_uint64(0)
    call("obhip_newton_workspace_bytes", p, C.byref(wsb))
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")
    th = torch.empty(p, dtype=torch.float64, device="cuda")
    dH = torch.empty(p, dtype=torch.float64, device="cuda")
    Gc = G.clone()
    call("obhip_newton_solve_dev", om._h, t._h, Gc.data_ptr(), g.data_ptr(), sigma, rho,
         th.data_ptr(), dH.data_ptr(), ws.data_ptr(), wsb.value)
    torch.cuda.synchronize()
    cond = float(torch.linalg.cond(H))
    assert float((th - want).norm() / want.norm()) < 1e-13 * max(cond, 10.0)
    assert float((dH - torch.diagonal(H)).abs().max() / torch.diagonal(H).abs().max()) < 1e-14


@pytest.mark.parametrize("panels", [1, 2, 4, 8])
def test_cholesky_schedules_with_one_to_eight_panels_per_pass(panels):
    """The blocked Cholesky with 1, 2, 4 and 8 panels per trailing pass (csrc/kernels_chol.hip: the
    default is 1 below p = 4096, 2 from there, 4 from 8192) at sizes where a pass ends inside a
    panel, right behind one, and in the middle of a pass; the solution and L itself against torch.
    The setting is read once per process: a child process per schedule."""
    import os
    import subprocess
    import sys
    sizes = [1, 64, 65, 129, 200, 257, 449, 513, 640, 1000, 1217]
    env = dict(os.environ, OBHIP_CHOL_PANELS=str(panels))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chol_schedule_worker.py")
    r = subprocess.run([sys.executable, worker] + [str(v) for v in sizes], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith(("ok", "BAD"))]
    assert [int(ln[1]) for ln in lines] == sizes, r.stdout
    assert all(ln[0] == "ok" for ln in lines), r.stdout


def test_cholesky_with_more_workgroups_than_the_gpu_holds():
    """p = 16448: 258 panel workgroups per step on 256 CUs and a ragged last block.  Guards
    the in-place write-back race found at p >= 16384 (late workgroups re-factorising a
    diagonal block that workgroup 0 had already overwritten with L_jj)."""
    import torch
    import outerbase_amd as ob
    from outerbase_amd._lib import call
    from outerbase_amd.driver import bench_knots
    p = 16448
    kinds = (["mat25", "mat25pow", "mat25ang"] * 14)[:40]
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, bench_knots(kinds, 40))
    terms = om.selectterms(p)
    t = ob.obmod._Terms(om, terms)
    torch.manual_seed(0)
    A = torch.randn((p, 256), dtype=torch.float64, device="cuda")
    G = A @ A.T + 10.0 * torch.eye(p, dtype=torch.float64, device="cuda")
    g = torch.randn(p, dtype=torch.float64, device="cuda")
    sigma, rho = 0.0, 20.0
    prec = torch.from_numpy(1.0 / (om.getvar(terms) * math.exp(2 * rho))).cuda()
    want = torch.linalg.solve(G + torch.diag(prec), g)
    wsb = C.c_uint64(0)
    call("obhip_newton_workspace_bytes", p, C.byref(wsb))
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")
    th = torch.empty(p, dtype=torch.float64, device="cuda")
    dH = torch.empty(p, dtype=torch.float64, device="cuda")
    call("obhip_newton_solve_dev", om._h, t._h, G.data_ptr(), g.data_ptr(), sigma, rho,
         th.data_ptr(), dH.data_ptr(), ws.data_ptr(), wsb.value)
    torch.cuda.synchronize()
    assert float((th - want).norm() / want.norm()) < 1e-10


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_against_oracle(seed):
    """Seeded fuzz over shapes the other tests do not enumerate: random d, n, p, covariance
    mix and knot counts; products, Gram, Newton fit, predictor and one gradient product
    against the oracle."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.integers(3, 11))
    kinds = [["mat25", "mat25pow", "mat25ang"][int(k)] for k in rng.integers(0, 3, d)]
    m = int(rng.integers(8, 31))
    n = int(rng.choice([1, 2, 63, 64, 65, 130, 777, 2049]))
    p = int(rng.choice([1, 2, 5, 64, 100, 129, 300]))
    om_o, om_d = make_pair(kinds, knots_for(kinds, m))
    x = sample_x(rng, n, kinds)
    terms = om_o.selectterms(p)
    bo = O.OuterBase(om_o, x, dograd=True)
    bd = ob.outerbase(om_d, x)
    a, v = rng.standard_normal(p), rng.standard_normal(n)
    # high levels (few knots, many terms) lose digits on both sides, see test_gram_backends
    tol = 1e-6 if terms.max() >= 8 else 1e-9
    assert relerr(bd.matmul(terms, a), O.ob_mm(bo, terms, a)) < tol
    assert relerr(bd.tmatmul(terms, v), O.ob_tmm(bo, terms, v)) < tol
    assert relerr(bd.getmat(terms), O.ob_getmat(bo, terms)) < tol
    assert relerr(bd.sqcolsums(terms), O.ob_sqcolsums(bo, terms)) < tol
    assert relerr(bd.matmul_gradhyp(terms, a), O.ob_mm_gradhyp(bo, terms, a)[1]) < 100 * tol
    assert relerr(bd.tmatmul_gradhyp(terms, v), O.ob_tmm_gradhyp(bo, terms, v)[1]) < 100 * tol
    # the contraction the likelihoods make of it, reduced on the device
    assert relerr(bd.matmul_gradhyp_dot(terms, a, v), v @ O.ob_mm_gradhyp(bo, terms, a)[1]) < 100 * tol
    aa = np.abs(a)
    assert relerr(bd.sqmm(terms, aa), O.ob_sqmm(bo, terms, aa)) < tol
    assert relerr(bd.sqtmm(terms, v), O.ob_sqtmm(bo, terms, v)) < tol
    assert relerr(bd.residvar(terms), O.ob_residvar(bo, terms)) < 100 * tol
    assert relerr(bd.sqcolsums_gradhyp(terms), O.ob_sqcolsums_gradhyp(bo, terms)) < 100 * tol
    assert relerr(bd.sqmm_gradhyp(terms, aa), O.ob_sqmm_gradhyp(bo, terms, aa)) < 100 * tol
    y = rng.standard_normal(n)
    if n >= 2:
        lik = ob.loglik_std(om_d, terms, y, x)
        lp = ob.lpdfvec(lik, ob.logpr_gauss(om_d, terms))
        lp.domarg = False
        B = O.ob_getmat(bo, terms)
        assert relerr(lp.hess() - np.diag(lp.logpr.diaghess()), math.exp(-2 * lik.para[0]) * (B.T @ B)) < tol
        lp.optnewton()
        theta_o, H_o = O.fit_newton(bo, terms, y, sigma=float(lik.para[0]))
        xnew = sample_x(rng, 37, kinds)
        pred = ob.predictor(lp)
        pred.update(xnew)
        assert relerr(pred.mean(), O.predict_mean(om_o, terms, theta_o, xnew)) < 1e-6
        if p <= 100:
            # the matrix-free PCG reaches the same solution (tight tolerance, generous cap)
            likg = ob.loglik_gauss(om_d, terms, y, x)
            lpg = ob.lpdfvec(ob.logpr_gauss(om_d, terms), likg)
            lpg.optcg(1e-14, 4000)
            assert relerr(likg.yhat, O.ob_mm(bo, terms, theta_o)) < 1e-5


@pytest.mark.parametrize("seed", range(24))
def test_random_terms_caps_and_gram_backends(seed):
    """Second fuzz: arbitrary (not downward closed) term sets with up to 8 factors, a basis
    built up to per-dimension level caps, every Gram back end, the Gauss predictor variance
    and a rebuild after updatehyp, all against the oracle."""
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd import _lib
    rng = np.random.default_rng(5000 + seed)
    d = int(rng.integers(3, 13))
    kinds = [["mat25", "mat25pow", "mat25ang"][int(k)] for k in rng.integers(0, 3, d)]
    m = int(rng.integers(10, 25))
    n = int(rng.choice([3, 64, 100, 257, 1500]))
    p = int(rng.choice([2, 7, 65, 128, 257]))
    maxlev = int(rng.integers(1, 6))
    max_nnz = int(rng.integers(1, min(8, d) + 1))
    hyp = None
    om_o, om_d = make_pair(kinds, knots_for(kinds, m), hyp=hyp)
    x = sample_x(rng, n, kinds)
    terms = random_terms(rng, p, d, maxlev, max_nnz)
    terms = np.unique(terms, axis=0)
    p = terms.shape[0]
    bo = O.OuterBase(om_o, x)
    caps = np.maximum(terms.max(axis=0), rng.integers(0, 3, d))      # at least what the terms use
    caps = np.minimum(caps, m - 1)
    bd = ob.outerbase(om_d, x, levelcap=caps)
    a, v = rng.standard_normal(p), rng.standard_normal(n)
    B = O.ob_getmat(bo, terms)
    tol = 1e-9
    assert relerr(bd.matmul(terms, a), B @ a) < tol
    assert relerr(bd.tmatmul(terms, v), B.T @ v) < tol
    y = rng.standard_normal(n)
    lik = ob.loglik_std(om_d, terms, y, x)
    e2 = math.exp(-2 * lik.para[0])
    try:
        for backend in (3, 4, 0):
            _lib.call("obhip_set_gram_backend", backend)
            G = lik.hess() / e2
            assert relerr(G, B.T @ B) < tol, backend
            assert np.array_equal(G, G.T)
        # the design matrix staged in row chunks of 128 (what happens when n x p doubles do
        # not fit in memory), and the column-from-HBM kernels that take over beyond the LDS tile
        for var in ("OBHIP_GRAM_CHUNK_ROWS", "OBHIP_FORCE_GENERIC"):
            os.environ[var] = "128"
            try:
                lik2 = ob.loglik_std(om_d, terms, y, x)
                G = lik2.hess() / e2
                assert relerr(G, B.T @ B) < tol, var
                assert np.array_equal(G, G.T)
                if var == "OBHIP_FORCE_GENERIC":
                    assert relerr(bd.matmul(terms, a), B @ a) < tol
                    assert relerr(bd.tmatmul(terms, v), B.T @ v) < tol
                    assert relerr(bd.getmat(terms), B) < tol
                    assert relerr(bd.sqmm(terms, np.abs(a)), (B * B) @ np.abs(a)) < tol
                    assert relerr(bd.sqtmm(terms, v), (B * B).T @ v) < tol
            finally:
                del os.environ[var]
    finally:
        _lib.call("obhip_set_gram_backend", 0)
    # pred_gauss variance with an arbitrary coefficient variance
    likg = ob.loglik_gauss(om_d, terms, y, x)
    lpg = ob.lpdfvec(ob.logpr_gauss(om_d, terms), likg)
    lpg.optcg(1e-12, 50)
    pred = ob.predictor(lpg)
    xnew = sample_x(rng, 19, kinds)
    pred.update(xnew)
    want = O.predict_var_gauss(om_o, terms, lpg.totdiaghess, float(likg.para[0]), xnew)
    assert relerr(pred.var(), want) < tol
    # hyper-parameters move: the basis follows only after build()
    hyp2 = ob.gethyp(om_d) + 0.05 * rng.standard_normal(len(ob.gethyp(om_d)))
    om_o.hyp_set(hyp2)
    om_d.updatehyp(hyp2)
    om_d.set_rotation(om_o.rotmat, om_o.basisvar, om_o.maxlevel)
    bd.build()
    B2 = O.ob_getmat(O.OuterBase(om_o, x), terms)
    assert relerr(bd.matmul(terms, a), B2 @ a) < tol


@pytest.mark.parametrize("max_nnz,p,maxlev", [
    (2, 64, 3), (2, 3000, 12), (2, 20000, 14),       # W = 2: NPAIR 1 / 2 / 4, two blocks along p
    (4, 300, 4), (4, 2300, 6), (4, 9000, 6),          # W = 4, up to three blocks along p
    (6, 700, 3), (6, 2500, 5),                        # W = 6: NPAIR 1 / 2
    (8, 1030, 3), (8, 4200, 3),                       # W = 8, three blocks along p
    (4, 3000, 14),                                    # > 128 used columns: staging without prefetch
])
def test_term_per_lane_variants(max_nnz, p, maxlev):
    """Every instantiation of the term-per-lane kernels (k_tmm_tl for B^T a, k_mm_tl for B a:
    columns per term, terms per lane, prefetch on / off, several blocks along p), plain and
    squared, against the oracle's design matrix on ragged row counts."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(31 * max_nnz + p)
    d = 10
    kinds = ["mat25", "mat25pow", "mat25ang", "mat25", "mat25"] * 2
    om_o, om_d = make_pair(kinds, knots_for(kinds, 16))
    terms = np.unique(random_terms(rng, p, d, maxlev, max_nnz), axis=0)
    if maxlev == 14:
        assert len({(l, t) for l in range(d) for t in set(terms[:, l]) if t > 0}) > 128
    for n in (1, 191, 1100, 70000):
        if n > 2000 and terms.shape[0] > 3000:
            continue            # the oracle's n x p design matrix
        x = sample_x(rng, n, kinds)
        B = O.ob_getmat(O.OuterBase(om_o, x), terms)
        bd = ob.outerbase(om_d, x)
        v = rng.standard_normal(n)
        # high levels lose digits in cov x rotmat on both sides (see test_gram_backends), so
        # the oracle pins the result loosely and the device's own design matrix (getmat, the
        # lane = row kernel) pins the contraction tightly
        # (levels 12-14 of 16 knots lie beyond `maxlevel`, where lambda_j / lambda_0 < 1e-11 and
        # the eigenpairs are rounding noise amplified by rotmat ~ 1 / lambda: kernel variants
        # are what this case is for, not parity)
        tol = 1e-5 if maxlev >= 12 else (1e-6 if maxlev >= 5 else 1e-9)
        Bd = bd.getmat(terms)
        got, gotsq = bd.tmatmul(terms, v), bd.sqtmm(terms, v)
        assert relerr(got, B.T @ v) < tol
        assert relerr(gotsq, (B * B).T @ v) < tol
        assert relerr(got, Bd.T @ v) < 1e-12
        assert relerr(gotsq, (Bd * Bd).T @ v) < 1e-12
        a = rng.standard_normal(terms.shape[0])
        got, gotsq = bd.matmul(terms, a), bd.sqmm(terms, np.abs(a))
        assert relerr(got, B @ a) < tol
        assert relerr(gotsq, (B * B) @ np.abs(a)) < tol
        assert relerr(got, Bd @ a) < 1e-12
        assert relerr(gotsq, (Bd * Bd) @ np.abs(a)) < 1e-12
        # the fused predictor (k_predict_tl; several passes over the terms when they exceed
        # one block's registers): mean = B a, var = B^2 cv + e^{2 sigma}
        from outerbase_amd._lib import call, ptr
        mean, var = np.empty(n), np.empty(n)
        cv, sig = np.abs(a) + 0.1, -0.3
        xf = np.asfortranarray(x)
        tt = ob.obmod._Terms(om_d, terms)
        call("obhip_predict", om_d._h, tt._h, ptr(a), ptr(xf), n, n, ptr(mean), ptr(cv), sig, ptr(var))
        assert relerr(mean, Bd @ a) < 1e-11
        assert relerr(var, (Bd * Bd) @ cv + math.exp(2 * sig)) < 1e-11
        call("obhip_predict", om_d._h, tt._h, ptr(a), ptr(xf), n, n, ptr(mean), None, sig, None)
        assert relerr(mean, Bd @ a) < 1e-11
        # the fused Hessian product (k_hm_tl: B^T (e^{-2 sigma} B a) in one pass when all terms
        # fit one block, the two kernels otherwise) through loglik_gauss$hessmult, and its
        # update() form (B^T (e^{-2 sigma} (y - B theta)), sum of squared residuals) through
        # three iterations of the device PCG against the oracle's lpdf::optcg
        if n >= 2:
            lik = ob.loglik_gauss(om_d, terms, v, x)
            e2 = math.exp(-2 * lik.para[0])
            assert relerr(lik.hessmult(a), e2 * (Bd.T @ (Bd @ a))) < 1e-11
            if n <= 1100 and maxlev < 12:
                sig0 = float(lik.para[0])
                th_o, it_o, m_o = O.fit_cg(O.OuterBase(om_o, x), terms, v, sigma=sig0, tol=1e-30, maxit=3)
                th = np.zeros(terms.shape[0])
                dh = np.empty(terms.shape[0])
                its = C.c_uint64(0)
                yv = np.ascontiguousarray(v)
                call("obhip_fit_cg", bd._h, tt._h, om_d._h, ptr(yv), sig0, 6.0, 1e-30, 3, ptr(th),
                     C.byref(its), ptr(dh), None)
                assert its.value == it_o == 3
                assert relerr(th, th_o) < (1e-5 if maxlev >= 5 else 1e-8)
                # the preconditioner e^{-2 sigma} sqcolsums + prior (fit.cpp:51): since round 4 taken
                # along with B^T r by the cold start's one pass (k_tmm_tl<DUAL>) where the terms allow
                assert relerr(dh, m_o) < 1e-10


@pytest.mark.parametrize("d,maxlev,want_mu", [(26, 5, 131), (30, 4, 121), (16, 6, 97)])
def test_fused_hessian_product_between_its_two_lds_limits(d, maxlev, want_mu):
    """k_hm2 keeps two tiles of the used columns in LDS: up to 147 columns for the Hessian product
    at 4 terms per lane, 118 for its update() form (which also keeps the coefficients there).
    131 columns: the Hessian products take k_hm2, the update() passes fall back to k_hm_tl;
    121: likewise; 97: both take k_hm2.  loglik_gauss$hessmult against B^T (B a) from getmat, and
    three iterations of the device PCG (update() at the start, Hessian products after) against
    the oracle's lpdf::optcg."""
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd._lib import call, ptr
    kinds = ["mat25"] * d
    om_o, om_d = make_pair(kinds, knots_for(kinds, 20))
    rng = np.random.default_rng(d)
    n, p = 700, 3000
    x = sample_x(rng, n, kinds)
    terms = random_terms(rng, p, d, maxlev, 4)
    for l in range(d):                      # every level of every dimension is used somewhere
        terms[1 + l * maxlev:1 + (l + 1) * maxlev, :] = 0
        terms[1 + l * maxlev:1 + (l + 1) * maxlev, l] = np.arange(1, maxlev + 1)
    tt = ob.obmod._Terms(om_d, terms)
    assert 1 + int(tt.maxlevels().sum()) == want_mu
    v = rng.standard_normal(n)
    a = rng.standard_normal(p)
    lik = ob.loglik_gauss(om_d, terms, v, x)
    Bd = lik.ob.getmat(terms)
    e2 = math.exp(-2 * lik.para[0])
    assert relerr(lik.hessmult(a), e2 * (Bd.T @ (Bd @ a))) < 1e-11
    sig0 = float(lik.para[0])
    th_o, it_o, m_o = O.fit_cg(O.OuterBase(om_o, x), terms, v, sigma=sig0, tol=1e-30, maxit=3)
    th, dh, its = np.zeros(p), np.empty(p), C.c_uint64(0)
    call("obhip_fit_cg", lik.ob._h, tt._h, om_d._h, ptr(np.ascontiguousarray(v)), sig0, 6.0, 1e-30, 3,
         ptr(th), C.byref(its), ptr(dh), None)
    assert its.value == it_o == 3
    assert relerr(th, th_o) < 1e-7 and relerr(dh, m_o) < 1e-10


@pytest.mark.parametrize("seed", range(16))
def test_random_lpdf_level_quantities(seed):
    """Third fuzz: the lpdf-level values and gradients (loglik_gauss / loglik_std /
    loglik_gda, logpr_gauss, both marginal adjustments, predr_std) on random shapes,
    including more terms than rows."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.integers(3, 9))
    kinds = [["mat25", "mat25pow", "mat25ang"][int(k)] for k in rng.integers(0, 3, d)]
    m = int(rng.integers(12, 25))
    n = int(rng.choice([20, 65, 200, 513]))
    p = int(rng.choice([3, 30, 64, 90]))
    om_o, om_d = make_pair(kinds, knots_for(kinds, m))
    x = sample_x(rng, n, kinds)
    y = rng.standard_normal(n)
    terms = om_o.selectterms(p)
    coeff = 0.1 * rng.standard_normal(p)
    sigma, rho = float(rng.uniform(-2, 0.5)), float(rng.uniform(1, 6))
    bo = O.OuterBase(om_o, x, dograd=True)
    gtol = 1e-5 if terms.max() >= 8 else 1e-7
    # diagonal marginal adjustment on lpdfvec(logpr, loglik_gauss)
    pr = ob.logpr_gauss(om_d, terms)
    lik = ob.loglik_gauss(om_d, terms, y, x)
    lp = ob.lpdfvec(pr, lik)
    lp.compute_gradhyp = lp.compute_gradpara = True
    lp.updatepara([rho, sigma])
    lp.update(coeff)
    v, g, gh, gp = O.loglik_update(bo, terms, y, sigma, coeff)
    pv, pg, pgh, pgp = O.logpr_update(om_o, terms, rho, coeff)
    mv, mgh, mgl, mgp = O.margadj_diag(bo, terms, sigma, rho)
    assert abs(lp.val - (v + pv + mv)) < 1e-9 * abs(v + pv + mv)
    assert relerr(lp.grad, g + pg) < 1e-8
    assert relerr(lp.gradhyp, gh + pgh + mgh) < gtol
    assert relerr(lp.gradpara, np.array([pgp[0] + mgp, gp[0] + mgl])) < 1e-8
    # full marginal adjustment after optnewton on lpdfvec(loglik_std, logpr)
    liks = ob.loglik_std(om_d, terms, y, x)
    prs = ob.logpr_gauss(om_d, terms)
    lps = ob.lpdfvec(liks, prs)
    lps.updatepara([sigma, rho])
    lps.optnewton()
    theta, H = O.fit_newton(bo, terms, y, sigma=sigma, rho=rho)
    v, g, gh, gp = O.loglik_update(bo, terms, y, sigma, theta)
    pv, pg, pgh, pgp = O.logpr_update(om_o, terms, rho, theta)
    fv, fgh, fgl, fgp = O.margadj_full(bo, terms, sigma, rho)
    assert abs(lps.val - (v + pv + fv)) < 1e-8 * abs(v + pv + fv)
    assert relerr(lps.gradhyp, gh + pgh + fgh) < 10 * gtol
    assert relerr(lps.gradpara, np.array([gp[0] + fgl, pgp[0] + fgp])) < 1e-6
    xnew = sample_x(rng, 23, kinds)
    pred = ob.predictor(lps)
    pred.update(xnew)
    assert relerr(pred.var(), O.predict_var_std(om_o, terms, H, sigma, xnew)) < 1e-6
    # loglik_gda
    para = np.array([sigma, float(rng.uniform(-1, 1))])
    lg = ob.loglik_gda(om_d, terms, y, x)
    lg.updatepara(para)
    lg.compute_gradhyp = lg.compute_gradpara = True
    lg.update(coeff)
    r = O.loglik_gda_update(bo, terms, y, para, coeff)
    # residvar = 1 - B^2 var cancels to ~1e-7 relative when the expansion explains almost
    # everything (modandbase.cpp:889-895), and val / gradients inherit that
    assert abs(lg.val - r["val"]) < 1e-7 * abs(r["val"])
    assert relerr(lg.gradhyp, r["gradhyp"]) < max(gtol, 1e-6)
    assert relerr(lg.gradpara, r["gradpara"]) < 1e-6
    assert relerr(lg.diaghessgradhyp(), r["diaghessgradhyp"]) < max(gtol, 1e-6)


@pytest.mark.parametrize("seed", range(6))
def test_obfit_on_random_smooth_functions(seed):
    """obfit / obpred end to end on random smooth test functions of 3..7 inputs with all
    three covariance families in play: finite, positive variances, and a fit far better
    than the mean predictor."""
    import outerbase_amd as ob
    rng = np.random.default_rng(300 + seed)
    d = int(rng.integers(3, 8))
    n = int(rng.choice([150, 400, 900]))
    covnames = [["mat25pow", "mat25", "mat25ang"][int(k)] for k in rng.integers(0, 3, d)]
    x = 0.05 + 0.9 * rng.random((n + 200, d))
    for j, c in enumerate(covnames):
        if c == "mat25ang":
            x[:, j] *= 6.283185
    w = rng.standard_normal(d)
    ph = rng.uniform(0, 3, d)

    def f(z):
        u = z.copy()
        for j, c in enumerate(covnames):
            u[:, j] = np.sin(z[:, j]) if c == "mat25ang" else z[:, j]
        return np.sin(u @ w + 0.5) + 0.3 * np.cos(2 * u[:, 0] + ph[0]) * u[:, 1] + 0.1 * (u ** 2) @ np.abs(w)

    y = f(x)
    numb = int(max(2 * d, min(80, n // 3)))
    m = ob.obfit(x[:n], y[:n], numb=numb, covnames=covnames, seed=seed)
    pred = ob.obpred(m, x[n:])
    assert np.all(np.isfinite(pred["mean"])) and np.all(np.isfinite(pred["var"]))
    assert np.all(pred["var"] > 0)
    rmse = math.sqrt(np.mean((pred["mean"] - y[n:]) ** 2))
    assert rmse < 0.5 * np.std(y[n:])


def test_device_memory_pool_recycles_and_trims():
    """The temporaries of the C-ABI calls come from a size-keyed pool (csrc/core.cpp): results
    must not depend on whether a block is fresh or recycled, and obhip_trim_pool hands the
    cached memory back to the driver."""
    import torch
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd import _lib
    rng = np.random.default_rng(77)
    kinds = ["mat25", "mat25pow", "mat25ang"]
    om_o, om_d = make_pair(kinds, knots_for(kinds, 12))
    terms = om_o.selectterms(40)
    n = 300000                     # 2.4 MB vectors: blocks large enough to see in hipMemGetInfo
    x = sample_x(rng, n, kinds)
    bd = ob.outerbase(om_d, x)
    a, v = rng.standard_normal(40), rng.standard_normal(n)
    first = (bd.matmul(terms, a), bd.tmatmul(terms, v), bd.matmul_gradhyp(terms, a))
    _lib.call("obhip_trim_pool")
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):             # same sizes again: recycled blocks, fill patterns of older calls
        again = (bd.matmul(terms, a), bd.tmatmul(terms, v), bd.matmul_gradhyp(terms, a))
        for f, g in zip(first, again):
            assert np.array_equal(f, g)
    cached = free0 - torch.cuda.mem_get_info()[0]
    assert cached > 0              # the temporaries stayed with the pool ...
    _lib.call("obhip_trim_pool")
    assert torch.cuda.mem_get_info()[0] >= free0 - (1 << 20)   # ... and went back on request
    assert relerr(first[0][:2000], O.ob_mm(O.OuterBase(om_o, x[:2000]), terms, a)) < 1e-9


# ---- tests/testthat/test-lpdf.R on the device objects -------------------------------------
def _lpdf_getvals(ob, om, obj, coeff, hyp, para):
    """getvals (test-lpdf.R:21-37)."""
    obj.compute_gradhyp = True
    obj.compute_gradpara = True
    om.updatehyp(hyp)
    obj.updateom()
    obj.updatepara(para)
    obj.update(coeff)
    return dict(val=obj.val, grad=obj.grad, gradhyp=obj.gradhyp, gradpara=obj.gradpara,
                diaghess=obj.diaghess(), diaghessgradhyp=obj.diaghessgradhyp(),
                diaghessgradpara=obj.diaghessgradpara())


def _lpdf_perturbvals(ob, om, obj, coeff, coeffp, hyp, hypp, para, parap, ep):
    """perturbvals (test-lpdf.R:39-72): differences of the value and of the Hessian diagonal
    along random directions against ep times the analytic directional derivatives."""
    gv = lambda c, h, q: _lpdf_getvals(ob, om, obj, c, h, q)
    L = gv(coeff, hyp, para)
    L_coeff = gv(coeff + ep * coeffp, hyp, para)
    L_para = gv(coeff, hyp, para + ep * parap)
    L_hyp = gv(coeff, hyp + ep * hypp, para)
    L2 = gv(coeff, hyp, para)
    grad = (L["grad"] + L2["grad"] + L_coeff["grad"]) / 3
    gradhyp = (L["gradhyp"] + L2["gradhyp"] + L_para["gradhyp"]) / 3
    gradpara = (L["gradpara"] + L2["gradpara"] + L_para["gradpara"]) / 3
    chk = {}
    chk["coeff"] = (L_coeff["val"] - L["val"], ep * float(np.sum(grad * coeffp)))
    chk["hyp"] = (L_hyp["val"] - L["val"], ep * float(np.sum(gradhyp * hypp)))
    chk["para"] = (L_para["val"] - L["val"], ep * float(np.sum(gradpara * parap)))
    chk["rep"] = (L["val"], L2["val"])
    chk["dhcoeff"] = (L_hyp["diaghess"] - L["diaghess"], ep * (L["diaghessgradhyp"] @ hypp))
    chk["dhpara"] = (L_para["diaghess"] - L["diaghess"], ep * (L["diaghessgradpara"] @ parap))
    return chk


def _all_equal(target, current, tolerance):
    """testthat::expect_equal / all.equal.numeric: mean relative difference
    sum|target - current| / sum|target| below the tolerance (absolute when target is ~0)."""
    target, current = np.atleast_1d(target).astype(float), np.atleast_1d(current).astype(float)
    xy = float(np.sum(np.abs(target - current)))
    xn = float(np.sum(np.abs(target)))
    if np.isfinite(xn) and xn > np.finfo(float).eps ** 0.5 * target.size:
        xy /= xn
    else:
        xy /= target.size
    return xy < tolerance


@pytest.mark.parametrize("ss,nterms", [(200, 100),       # "short, skinny" (test-lpdf.R:132-168)
                                        (10000, 100),     # "tall, skinny"  (:171-206)
                                        (200, 1000)])     # "short, wide"   (:209-245)
def test_reference_lpdf_perturbation_suite(ss, nterms):
    """fulltest of tests/testthat/test-lpdf.R:76-126 on the device: the prior, the Gaussian
    likelihood and their lpdfvec at the reference's three shapes; values move along random
    coeff / hyp / para directions by ep times the analytic directional derivative, the
    Hessian diagonal moves by ep times diaghessgradhyp / diaghessgradpara, and a repeated
    evaluation reproduces the value -- every expectation of the reference's file, with its
    tolerance (0.01 relative), none of them downgraded to a warning."""
    import outerbase_amd as ob
    import ob_oracle as O
    d, ep = 8, 1e-4
    rng = np.random.default_rng(42)
    xo = rng.uniform(size=(ss, d))
    yo = O.borehole8d(xo) + 77.0                                  # test-lpdf.R:1-15 (no offset)
    y = (yo - yo.mean()) / yo.std(ddof=1)
    om = ob.outermod()
    ob.setcovfs(om, ["mat25"] * d)
    ob.setknot(om, [np.arange(0.001, 0.999, 0.05)] * d)           # :89
    hyp = ob.gethyp(om)
    om.updatehyp(hyp)
    terms = om.selectterms(nterms)
    logpr = ob.logpr_gauss(om, terms)
    loglik = ob.loglik_gauss(om, terms, y, xo)
    sdy = y.std(ddof=1)
    coeff = sdy / 100 * rng.standard_normal(nterms)
    coeffp = sdy / 100 * rng.standard_normal(nterms)
    hypp = rng.uniform(size=hyp.size) - 0.5
    prpara, likpara = np.array([math.log(1.0)]), np.array([math.log(0.1)])
    prinfo = _lpdf_perturbvals(ob, om, logpr, coeff, coeffp, hyp, hypp, prpara,
                               rng.standard_normal(1), ep)
    likinfo = _lpdf_perturbvals(ob, om, loglik, coeff, coeffp, hyp, hypp, likpara,
                                rng.standard_normal(1), ep)
    logpdf = ob.lpdfvec(loglik, logpr)
    vecpara = ob.getpara(logpdf)
    vecinfo = _lpdf_perturbvals(ob, om, logpdf, coeff, coeffp, hyp, hypp, vecpara,
                                rng.standard_normal(vecpara.size), ep)
    for name, info in (("pr", prinfo), ("lik", likinfo), ("vec", vecinfo)):
        for what in ("coeff", "hyp", "para"):
            assert _all_equal(info[what][0], info[what][1], 0.01), (name, what, info[what])
    assert _all_equal(likinfo["dhcoeff"][0], likinfo["dhcoeff"][1], 0.01)
    assert _all_equal(likinfo["dhpara"][0], likinfo["dhpara"][1], 0.01)
    assert _all_equal(likinfo["rep"][0], likinfo["rep"][1], 1e-12)


@pytest.mark.parametrize("p,bad_at", [(40, 17), (300, 5), (300, 299), (4200, 2100)])
def test_newton_solve_reports_a_hessian_that_is_not_positive_definite(p, bad_at):
    """A negative pivot anywhere (first block, last ragged block, middle of a two-panel
    schedule) comes back as an error naming the 64-column block, not as NaNs: the reference's
    solve() (fit.cpp:120) throws there too."""
    import torch
    import outerbase_amd as ob
    from outerbase_amd._lib import call
    kinds = ["mat25"] * 6
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, knots_for(kinds, 40))
    terms = om.selectterms(p)
    t = ob.obmod._Terms(om, terms)
    torch.manual_seed(p + bad_at)
    A = torch.randn((p, p + 3), dtype=torch.float64, device="cuda")
    G = A @ A.T + 0.5 * torch.eye(p, dtype=torch.float64, device="cuda")
    G[bad_at, bad_at] = -1e15         # indefinite whatever the prior adds to the diagonal
    g = torch.randn(p, dtype=torch.float64, device="cuda")
    wsb = C.c_uint64(0)
    call("obhip_newton_workspace_bytes", p, C.byref(wsb))
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")
    th = torch.empty(p, dtype=torch.float64, device="cuda")
    dH = torch.empty(p, dtype=torch.float64, device="cuda")
    with pytest.raises(ob.ObhipError, match="not positive definite") as ei:
        call("obhip_newton_solve_dev", om._h, t._h, G.data_ptr(), g.data_ptr(), 0.3, 2.0,
             th.data_ptr(), dH.data_ptr(), ws.data_ptr(), wsb.value)
    assert "column %d" % (bad_at // 64 * 64) in str(ei.value)


def test_host_buffer_predr_std_and_the_small_helpers():
    """obhip_predict_std on host buffers (the entry the Rcpp glue would use for predr_std,
    loglik_std.cpp:218-256) against the oracle, with x passed with a leading dimension larger
    than n; and obhip_basis_dims / obhip_memcpy_d2d / obhip_synchronize, which nothing else
    calls directly."""
    import torch
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd._lib import call, ptr
    kinds = ["mat25", "mat25pow", "mat25ang", "mat25"]
    om_o, om_d = make_pair(kinds, knots_for(kinds, 24))
    x, y = O.synth_xy(42, 0, 600, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    terms = om_o.selectterms(150)
    bo = O.OuterBase(om_o, x)
    sigma = math.log(0.1)
    theta, H = O.fit_newton(bo, terms, y, sigma=sigma)
    n, ldx = 257, 300
    xnew, _ = O.synth_xy(43, 0, n, kinds)
    xpad = np.full((ldx, len(kinds)), np.nan, order="F")
    xpad[:n] = xnew
    t = ob.obmod._Terms(om_d, terms)
    mean, var = np.empty(n), np.empty(n)
    Hs = np.asfortranarray(0.5 * (H + H.T))
    call("obhip_predict_std", om_d._h, t._h, ptr(np.ascontiguousarray(theta)), ptr(Hs), ptr(xpad),
         n, ldx, ptr(mean), sigma, ptr(var))
    assert relerr(mean, O.predict_mean(om_o, terms, theta, xnew)) < 1e-10
    assert relerr(var, O.predict_var_std(om_o, terms, H, sigma, xnew)) < 1e-8
    # mean only: H and var may be NULL
    mean2 = np.empty(n)
    call("obhip_predict_std", om_d._h, t._h, ptr(np.ascontiguousarray(theta)), None, ptr(xpad), n, ldx,
         ptr(mean2), sigma, None)
    assert np.array_equal(mean, mean2)
    # the small helpers
    bd = ob.outerbase(om_d, xnew)
    nn, dd, nc = C.c_uint64(), C.c_uint64(), C.c_uint64()
    call("obhip_basis_dims", bd._h, C.byref(nn), C.byref(dd), C.byref(nc))
    assert (nn.value, dd.value) == (n, len(kinds)) and nc.value >= 1 + len(kinds)
    a = torch.arange(1000, dtype=torch.float64, device="cuda")
    b = torch.zeros_like(a)
    call("obhip_memcpy_d2d", C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), 8000)
    call("obhip_synchronize")
    assert torch.equal(a, b)


@pytest.mark.parametrize("n,p", [(3000, 1100), (700, 520), (2500, 2048), (1500, 4096)])
def test_gram_diagonal_tiles_packed_four_into_three_blocks(n, p, monkeypatch):
    """The staged-design-matrix Gram kernel gives four consecutive diagonal 128 x 128 tiles to
    three workgroups (a diagonal tile has three distinct 64 x 64 quadrants, not four).  Same G
    as with one workgroup per diagonal tile (OBHIP_GRAM_DIAG4=0) up to summation order, exactly
    symmetric, and the oracle's B^T B; p = 1100: two groups and a left-over tile, 520: one group
    and one left-over, 2048: four groups, 4096 (the headline's p): eight groups."""
    import ob_oracle as O
    import outerbase_amd as ob
    kinds = ["mat25"] * 8
    om_o, om_d = make_pair(kinds, knots_for(kinds, 30))
    rng = np.random.default_rng(p)
    x = sample_x(rng, n, kinds)
    terms = om_o.selectterms(p)
    y = rng.standard_normal(n)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("OBHIP_GRAM_DIAG4", flag)
        lik = ob.loglik_std(om_d, terms, y, x)
        got[flag] = lik.hess() * math.exp(2 * lik.para[0])
        assert np.array_equal(got[flag], got[flag].T)
    scale = np.max(np.abs(got["0"]))
    assert np.max(np.abs(got["1"] - got["0"])) < 1e-13 * scale
    B = O.ob_getmat(O.OuterBase(om_o, x), terms)
    # (4096 terms of eight dimensions reach levels whose knot sums lose digits on both sides, as in
    # test_gram_backends: the two schedules pin each other to 1e-13 above, the oracle loosely)
    assert relerr(got["1"], B.T @ B) < (1e-9 if p <= 2048 else 1e-6)


def test_obfit_trajectory_matches_harness_oracle(monkeypatch):
    """The whole two-stage obfit flow (R/fitting.R:27-137: loglik_gda stage on a row subset,
    then two BFGS runs on loglik_gauss; BFGS_std, .lpdfwrapper, lpdf::optcg underneath) on the
    device against its independent CPU restatement oracle/ob_harness.py, iteration by iteration:
    Borehole, n = 400, d = 8, 100 terms, the same row subset on both sides.  Per BFGS iteration
    the objective value and the step length must agree to 1e-6 relative for as long as the two
    runs take the same branch of the Wolfe bisection (a restart on one side and a step on the
    other ends the comparison: from there the trajectories are different optimisations), and that
    must hold for at least the first 10 iterations.  The device model takes the oracle's
    eigen-rotation at every hyper-parameter update, as every parity test does (DESIGN.md
    section 6), and the PCG re-evaluates value and gradient every iteration like the reference
    (OBHIP_CG_REFRESH=1)."""
    import ob_oracle as O
    import ob_harness as H
    import outerbase_amd as ob
    from outerbase_amd import fitting, obmod
    monkeypatch.setenv("OBHIP_CG_REFRESH", "1")
    rng = np.random.default_rng(11)
    n, d, numb, seed = 400, 8, 100, 5
    x = 0.02 + 0.96 * rng.random((n, d))
    y = O.borehole8d(x)
    covnames = ["mat25pow"] * d

    # the device model mirrors an oracle outermod and takes its rotation after every change
    mirrors = {}

    def sync(om):
        m = mirrors[id(om)]
        om.set_rotation(m.rotmat, m.basisvar, m.maxlevel)
        om.set_rotation_grad(m.rotmat_gradhyp, m.logbasisvar_gradhyp)

    real_setknot, real_updatehyp = obmod.setknot, obmod.outermod.updatehyp

    def setknot(om, knotlist):
        real_setknot(om, knotlist)
        m = O.OuterMod()
        m.setcovfs(om.covnames)
        m.hyp = obmod.gethyp(om).copy()
        m.setknot([np.asarray(k, dtype=np.float64) for k in knotlist])
        mirrors[id(om)] = m
        sync(om)

    def updatehyp(self, hyp):
        real_updatehyp(self, hyp)
        if id(self) in mirrors:
            mirrors[id(self)].hyp_set(np.asarray(hyp, dtype=np.float64))
            sync(self)

    monkeypatch.setattr(fitting, "setknot", setknot)
    monkeypatch.setattr(obmod.outermod, "updatehyp", updatehyp)
    logs = []
    real_bfgs = fitting.BFGS_std

    def bfgs(*a, **k):
        out = real_bfgs(*a, **k)
        logs.append(out["log"])
        return out
    monkeypatch.setattr(fitting, "BFGS_std", bfgs)

    got = fitting.obfit(x, y, numb=numb, covnames=covnames, seed=seed)
    numbr = min(n // 2, numb, 80 * d)
    sub = np.random.default_rng(seed).choice(n, size=min(n, 3 * numbr), replace=False)
    want = H.obfit(x, y, numb, covnames, sub)
    assert len(logs) == 3 and len(want["traces"]) == 3

    def rel(a, b):
        return abs(a - b) / max(abs(a), abs(b), 1e-300)
    matched, total, same_path = [], 0, True
    for run, (lg, tr) in enumerate(zip(logs, want["traces"])):
        k = 0
        for dv, ov in zip(lg, tr):
            d_restart, o_restart = dv["val"] is None, ov[1] is None
            if d_restart != o_restart:
                break
            if not d_restart and (rel(dv["val"], ov[1]) > 1e-6 or rel(dv["lr"], ov[2]) > 1e-6):
                break
            if d_restart and rel(dv["lr"], ov[2]) > 1e-6:
                break
            k += 1
        matched.append((k, len(lg), len(tr)))
        total += k
        if k < min(len(lg), len(tr)) or len(lg) != len(tr):
            same_path = False
            break
    # the first BFGS run (the loglik_gda stage) agrees for at least 10 iterations (or to its end)
    k0, nd0, no0 = matched[0]
    assert k0 >= min(11, nd0, no0), "trajectories part at iteration %d of the first run: %r" % (k0, matched)
    if same_path:
        # identical branch sequences throughout: the final model is the same model
        hyp_d, hyp_o = ob.gethyp(got["om"]), want["om"].hyp
        assert np.max(np.abs(hyp_d - hyp_o)) < 1e-6 * max(1.0, np.max(np.abs(hyp_o)))
        xt = 0.02 + 0.96 * np.random.default_rng(3).random((50, d))
        pd_, po = ob.obpred(got, xt)["mean"], H.obpred_mean(want, xt)
        assert np.max(np.abs(pd_ - po)) < 1e-6 * np.max(np.abs(po))
    print("obfit trajectory parity: matched iterations per BFGS run (matched, device, oracle):", matched,
          "identical branch sequence:", same_path)
