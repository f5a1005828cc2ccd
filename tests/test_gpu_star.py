"""The kernels on shared sub-products (csrc/kernels_star.hip, star tables of csrc/share.cpp) against
the oracle and against the device's own lane = row design matrix (getmat), on term sets that make
them run: downward-closed sets of selectterms (src/modandbase.cpp:387-440) from 2049 terms up, also
with left-over terms, more terms than one workgroup holds, terms of six factors -- and the sets
they must NOT take (not downward-closed beyond the left-over budget, fewer than nine star-waves),
where the kernels of rounds 1-4 answer instead.

Products checked: B a, B^2 a (prodmm_, src/linalg.cpp:57-131), B^T a, (B^2)^T a (tprodmm_,
linalg.cpp:286-355), the PCG's Hessian product and update() pass (loglik_gauss.cpp:117-145), its
cold start's dual pass (preconditioner, loglik_gauss.cpp:154-157), the fused predictor with and
without the variance (loglik_gauss.cpp:214-227)."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import knots_for, make_pair, sample_x

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


def share_info(om_d, terms):
    import outerbase_amd as ob
    from outerbase_amd._lib import call, ptr
    info = np.zeros(11, dtype=np.uint64)
    tt = ob.obmod._Terms(om_d, terms)          # (kept alive through the call)
    call("obhip_terms_share_tables", tt._h, ptr(info), None, None, None, None)
    return dict(p_pad=int(info[0]), nleft=int(info[1]), reads=int(info[2]), reads_plain=int(info[3]),
                W=int(info[4]), nswf=int(info[7]))


def check_products(om_o, om_d, kinds, terms, n, rng, tol_oracle):
    """every product of the term-per-lane family on n rows: tight against getmat (the lane = row
    kernel, one product per (term, row) in the reference's factor order), loose against the oracle"""
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd._lib import call, ptr
    p = terms.shape[0]
    x = sample_x(rng, n, kinds)
    bd = ob.outerbase(om_d, x)
    Bd = bd.getmat(terms)
    B = O.ob_getmat(O.OuterBase(om_o, x), terms)
    assert relerr(Bd, B) < tol_oracle
    v = rng.standard_normal(n)
    a = rng.standard_normal(p)
    assert relerr(bd.tmatmul(terms, v), Bd.T @ v) < 1e-12
    assert relerr(bd.sqtmm(terms, v), (Bd * Bd).T @ v) < 1e-12
    assert relerr(bd.matmul(terms, a), Bd @ a) < 1e-12
    assert relerr(bd.sqmm(terms, np.abs(a)), (Bd * Bd) @ np.abs(a)) < 1e-12
    assert relerr(bd.matmul(terms, a), B @ a) < tol_oracle
    # the fused predictor, mean and variance
    tt = ob.obmod._Terms(om_d, terms)
    mean, var = np.empty(n), np.empty(n)
    cv, sig = np.abs(a) + 0.1, -0.3
    xf = np.asfortranarray(x)
    call("obhip_predict", om_d._h, tt._h, ptr(a), ptr(xf), n, n, ptr(mean), ptr(cv), sig, ptr(var))
    assert relerr(mean, Bd @ a) < 1e-11
    assert relerr(var, (Bd * Bd) @ cv + math.exp(2 * sig)) < 1e-11
    call("obhip_predict", om_d._h, tt._h, ptr(a), ptr(xf), n, n, ptr(mean), None, sig, None)
    assert relerr(mean, Bd @ a) < 1e-11
    if n < 2:
        return bd          # (loglik_gauss' default noise level is log(0.01 var(y)))
    # the Hessian product and three PCG iterations (update() form at the start, the dual pass for
    # the preconditioner) against the oracle's lpdf::optcg
    lik = ob.loglik_gauss(om_d, terms, v, x)
    e2 = math.exp(-2 * lik.para[0])
    # (4096 terms of size ~1 with random signs: B a cancels four digits, in NumPy's order and in the
    # kernel's: 1e-10 here where the single products above hold 1e-12)
    assert relerr(lik.hessmult(a), e2 * (Bd.T @ (Bd @ a))) < 1e-10
    sig0 = float(lik.para[0])
    th_o, it_o, m_o = O.fit_cg(O.OuterBase(om_o, x), terms, v, sigma=sig0, tol=1e-30, maxit=3)
    th, dh, its = np.zeros(p), np.empty(p), C.c_uint64(0)
    call("obhip_fit_cg", bd._h, tt._h, om_d._h, ptr(np.ascontiguousarray(v)), sig0, 6.0, 1e-30, 3, ptr(th),
         C.byref(its), ptr(dh), None)
    assert its.value == it_o == 3
    # (against the oracle: its design matrix differs from the device's by tol_oracle at high levels)
    assert relerr(th, th_o) < max(1e-8, 100 * tol_oracle) and relerr(dh, m_o) < max(1e-9, tol_oracle)
    # ... and tightly against the device's own design matrix: m = e^{-2 sigma} sqcolsums + prior
    prec = 1.0 / (om_o.getvar(terms) * math.exp(12.0))
    assert relerr(dh, e2 * (Bd * Bd).sum(0) + prec) < 1e-11
    return bd


@pytest.mark.parametrize("kinds,p,want_w", [
    (["mat25"] * 20, 4096, 4),          # the headline term set: 16 star-waves, 44 left-over terms
    (["mat25"] * 20, 2500, 4),          # 10 star-waves: six waves of the workgroup only stage
    (["mat25pow"] * 8, 3000, 6),        # six-factor terms
    (["mat25", "mat25pow", "mat25ang"] * 4, 3300, 4),
])
def test_star_kernels_on_selected_terms(kinds, p, want_w):
    om_o, om_d = make_pair(kinds, knots_for(kinds, 40))
    terms = om_o.selectterms(3 * p)
    # at most 12 levels per dimension (still downward-closed): two tiles of the used columns fit LDS,
    # as in obfit's term sets at this shape
    terms = terms[terms.max(1) <= 12][:p]
    assert len(terms) == p
    info = share_info(om_d, terms)
    assert info["W"] == want_w and 9 <= info["nswf"] <= 16 and info["nleft"] <= 192      # k_star's domain
    assert info["reads"] <= 0.65 * info["reads_plain"]
    assert 2 * (1 + int(terms.max(0).sum())) * 65 * 8 + 24000 <= 156 * 1024             # ... and its LDS
    rng = np.random.default_rng(p)
    # (high levels lose digits in the knot sums on both sides: the oracle pins loosely there, the
    # device's own design matrix tightly)
    tol = 1e-5 if terms.max() >= 12 else (1e-6 if terms.max() >= 5 else 1e-9)
    for n in (1, 300, 5000):
        check_products(om_o, om_d, kinds, terms, n, rng, tol)


def test_star_kernels_with_more_terms_than_one_workgroup_holds():
    """p = 9000 at d = 20: 35 family star-waves, three workgroups along the terms for B a / B^T a
    (their partial row sums go through k_mm_tl_sum); the Hessian product needs all terms in one
    workgroup and takes the two-kernel form."""
    kinds = ["mat25"] * 20
    om_o, om_d = make_pair(kinds, knots_for(kinds, 40))
    terms = om_o.selectterms(9000)
    info = share_info(om_d, terms)
    assert info["nswf"] > 32 and info["nleft"] <= 192
    rng = np.random.default_rng(9)
    check_products(om_o, om_d, kinds, terms, 700, rng, 1e-6)


def test_term_sets_the_star_kernels_take_and_refuse():
    """A downward-closed set with a handful of foreign terms (caller-supplied, NOT downward-closed):
    they are left over and multiplied out in the middle step.  With more foreign terms than the
    left-over budget (192), and with an entirely random set, the star kernels step aside -- the
    results are the same either way."""
    import outerbase_amd as ob
    kinds = ["mat25"] * 14
    om_o, om_d = make_pair(kinds, knots_for(kinds, 40))
    base = om_o.selectterms(3000)
    rng = np.random.default_rng(3)

    def foreign(count):
        t = np.zeros((count, len(kinds)), dtype=np.int64)
        for k in range(count):
            dims = rng.choice(len(kinds), size=4, replace=False)
            t[k, dims] = rng.integers(2, 5, size=4)
        return t
    have = {tuple(r) for r in base}
    for count, in_domain in ((40, True), (400, False)):
        extra = np.array([r for r in foreign(3 * count) if tuple(r) not in have][:count])
        terms = np.vstack([base, extra])
        info = share_info(om_d, terms)
        assert info["nleft"] >= count
        assert (info["nleft"] <= 192) == in_domain
        check_products(om_o, om_d, kinds, terms, 400, rng, 1e-9)
    # nothing shared at all
    t = np.unique(np.vstack([np.zeros((1, len(kinds)), dtype=np.int64), foreign(2500)]), axis=0)
    assert share_info(om_d, t)["nleft"] > 192
    check_products(om_o, om_d, kinds, t, 300, rng, 1e-9)


def test_star_kernels_over_many_tiles():
    """n = 200 000 rows (3125 tiles: every workgroup walks a dozen of them, so the tile hand-over by
    the counter of landed shares runs thousands of times) at the headline terms: B a, B^T a,
    sqcolsums and the Hessian product against the design matrix built by getmat (the lane = row
    kernel of round 1) in blocks of 20 000 rows."""
    import outerbase_amd as ob
    kinds = ["mat25"] * 20
    om_o, om_d = make_pair(kinds, knots_for(kinds, 40))
    terms = om_o.selectterms(4096)
    rng = np.random.default_rng(5)
    n = 200_000
    x = sample_x(rng, n, kinds)
    bd = ob.outerbase(om_d, x)
    a = rng.standard_normal(4096)
    v = rng.standard_normal(n)
    Ba = np.empty(n)
    Btv = np.zeros(4096)
    sq = np.zeros(4096)
    for r0 in range(0, n, 20_000):
        blk = ob.outerbase(om_d, x[r0:r0 + 20_000]).getmat(terms)
        Ba[r0:r0 + 20_000] = blk @ a
        Btv += blk.T @ v[r0:r0 + 20_000]
        sq += (blk * blk).T @ np.ones(blk.shape[0])
    assert relerr(bd.matmul(terms, a), Ba) < 1e-12
    assert relerr(bd.tmatmul(terms, v), Btv) < 1e-11
    assert relerr(bd.sqtmm(terms, np.ones(n)), sq) < 1e-11
    lik = ob.loglik_gauss(om_d, terms, v, x)
    e2 = math.exp(-2 * lik.para[0])
    want = np.zeros(4096)
    for r0 in range(0, n, 20_000):
        blk = ob.outerbase(om_d, x[r0:r0 + 20_000]).getmat(terms)
        want += blk.T @ Ba[r0:r0 + 20_000]
    assert relerr(lik.hessmult(a), e2 * want) < 1e-11
