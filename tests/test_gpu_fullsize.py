"""BASELINE.json configs at FULL size (configs[2]: d=20, n=1e6, p=4096; also configs[1],
one rank's shard of configs[4] and configs[0]) through size-independent properties: the oracle cannot produce a reference at this size
in seconds, but the kernels must agree with EACH OTHER (the Gram / Cholesky
kernels vs the matrix-free kernels vs the fused predictor)."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hot():
    import torch
    from outerbase_amd.driver import HotPath
    hp = HotPath(["mat25"] * 20, 40, 4096, 1_000_000)
    hp.setup()
    hp.step()
    torch.cuda.synchronize()
    yield hp
    hp.close()


def _vec(torch, n):
    return torch.empty(n, dtype=torch.float64, device="cuda")


def test_standardised_targets(hot):
    # standardised over all rows by obhip_standardise_dev: two passes like R's sd() -- the mean
    # from (sum y, n), then the centred sum of squares (24 bytes cross the ranks)
    y = hot.standardised_targets().cpu().numpy()
    assert abs(y.mean()) < 1e-12
    assert abs(y.var(ddof=1) - 1.0) < 1e-11


def test_gram_is_consistent_with_matrix_free_kernels(hot):
    import torch
    from outerbase_amd._lib import call
    p, n = hot.p, hot.n
    G = torch.empty((p, p), dtype=torch.float64, device="cuda")
    g = _vec(torch, p)
    y = hot.standardised_targets()
    call("obhip_gram_dev", hot.basis, hot.t._h, y.data_ptr(), G.data_ptr(), g.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(G, G.T)
    # diag(B^T B) == column sums of B^2 (modandbase.cpp:863-867)
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    sq = _vec(torch, p)
    call("obhip_basis_tmm_dev", hot.basis, hot.t._h, ones.data_ptr(), sq.data_ptr(), 1)
    torch.cuda.synchronize()
    assert float(((torch.diagonal(G) - sq).abs() / sq).max()) < 1e-11
    # G a == B^T (B a)
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.standard_normal(p)).cuda()
    Ba, BtBa = _vec(torch, n), _vec(torch, p)
    call("obhip_basis_mm_dev", hot.basis, hot.t._h, a.data_ptr(), Ba.data_ptr(), 0)
    call("obhip_basis_tmm_dev", hot.basis, hot.t._h, Ba.data_ptr(), BtBa.data_ptr(), 0)
    torch.cuda.synchronize()
    Ga = G @ a
    assert float((Ga - BtBa).abs().max() / Ga.abs().max()) < 1e-11
    # g == B^T y through an independent launch
    g2 = _vec(torch, p)
    call("obhip_basis_tmm_dev", hot.basis, hot.t._h, y.data_ptr(), g2.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(g, g2)
    # the right-hand side the fit used, (B^T y_raw - cent B^T 1) / sca, is the same vector
    assert float((hot.g - g).abs().max() / g.abs().max()) < 1e-11


def test_mm_is_linear(hot):
    import torch
    from outerbase_amd._lib import call
    rng = np.random.default_rng(1)
    a = torch.from_numpy(rng.standard_normal(hot.p)).cuda()
    b = torch.from_numpy(rng.standard_normal(hot.p)).cuda()
    outs = []
    for v in (a, b, 2.0 * a - 3.0 * b):
        o = _vec(torch, hot.n)
        call("obhip_basis_mm_dev", hot.basis, hot.t._h, v.data_ptr(), o.data_ptr(), 0)
        outs.append(o)
    torch.cuda.synchronize()
    err = (outs[2] - (2.0 * outs[0] - 3.0 * outs[1])).abs().max() / outs[2].abs().max()
    assert float(err) < 1e-12


def test_newton_solution_is_stationary(hot):
    """H theta = e^{-2 sigma} B^T y with H applied matrix-free: checks the Gram,
    Cholesky and triangular-solve kernels at p = 4096 against mm / tmm."""
    import torch
    from outerbase_amd._lib import call
    e2 = math.exp(-2 * hot.sigma)
    tmp, hv = _vec(torch, hot.n), _vec(torch, hot.p)
    call("obhip_basis_mm_dev", hot.basis, hot.t._h, hot.theta.data_ptr(), tmp.data_ptr(), 0)
    call("obhip_basis_tmm_dev", hot.basis, hot.t._h, tmp.data_ptr(), hv.data_ptr(), 0)
    torch.cuda.synchronize()
    prec = torch.from_numpy(1.0 / (hot.om.getvar(hot.terms) * math.exp(2 * hot.rho))).cuda()
    lhs = e2 * hv + prec * hot.theta
    rhs = e2 * hot.g
    assert float((lhs - rhs).norm() / rhs.norm()) < 1e-10


def test_fused_predictor_equals_stored_basis_path(hot):
    """predict at the TRAINING rows through k_predict (basis rebuilt in LDS) must
    equal k_mm on the stored basemat."""
    import torch
    from outerbase_amd._lib import call
    via_mm, via_pred = _vec(torch, hot.n), _vec(torch, hot.n)
    call("obhip_basis_mm_dev", hot.basis, hot.t._h, hot.theta.data_ptr(), via_mm.data_ptr(), 0)
    call("obhip_predict_dev", hot.om._h, hot.t._h, hot.theta.data_ptr(), hot.x.data_ptr(), hot.n,
         via_pred.data_ptr(), None, hot.sigma, None)
    torch.cuda.synchronize()
    assert float((via_mm - via_pred).abs().max() / via_mm.abs().max()) < 1e-12
    # and the fit explains the data: residual variance well below the prior noise guess
    resid = via_mm - hot.standardised_targets()
    assert float(resid.var()) < 0.5


def test_gradient_products_are_adjoint_at_full_size(hot):
    """v^T (d(B a)/dhyp_h) == a^T (d(B^T v)/dhyp_h) for every hyper-parameter: the one-pass
    kernel k_mmge against the streaming + restricted-view path of the transposed products,
    at n = 1e6, p = 4096 (and the same for the squared stores)."""
    import outerbase_amd as ob
    rng = np.random.default_rng(5)
    x = np.ascontiguousarray(hot.x.cpu().numpy().T)          # n x d
    b = ob.outerbase(hot.om, x, levelcap=hot.t.maxlevels())
    a = rng.standard_normal(hot.p)
    v = rng.standard_normal(hot.n)
    lhs = v @ b.matmul_gradhyp(hot.t, a)
    rhs = a @ b.tmatmul_gradhyp(hot.t, v)
    assert np.max(np.abs(lhs - rhs)) < 1e-10 * np.max(np.abs(lhs))
    aa = np.abs(a)
    lhs = v @ b.sqmm_gradhyp(hot.t, aa)
    rhs = aa @ b.sqtmm_gradhyp(hot.t, v)
    assert np.max(np.abs(lhs - rhs)) < 1e-10 * np.max(np.abs(lhs))


# ---- the other BASELINE.json configs ----------------------------------------------------------
def _fit_is_stationary(hp, tol):
    """Newton stationarity with H applied matrix-free + fused predictor vs stored basis."""
    import torch
    from outerbase_amd._lib import call
    e2 = math.exp(-2 * hp.sigma)
    tmp, hv = _vec(torch, hp.n), _vec(torch, hp.p)
    call("obhip_basis_mm_dev", hp.basis, hp.t._h, hp.theta.data_ptr(), tmp.data_ptr(), 0)
    call("obhip_basis_tmm_dev", hp.basis, hp.t._h, tmp.data_ptr(), hv.data_ptr(), 0)
    torch.cuda.synchronize()
    prec = torch.from_numpy(1.0 / (hp.om.getvar(hp.terms) * math.exp(2 * hp.rho))).cuda()
    lhs = e2 * hv + prec * hp.theta
    rhs = e2 * hp.g
    assert float((lhs - rhs).norm() / rhs.norm()) < tol
    via_pred = _vec(torch, hp.n)
    call("obhip_predict_dev", hp.om._h, hp.t._h, hp.theta.data_ptr(), hp.x.data_ptr(), hp.n,
         via_pred.data_ptr(), None, hp.sigma, None)
    torch.cuda.synchronize()
    assert float((tmp - via_pred).abs().max() / tmp.abs().max()) < 1e-11


def test_config1_d10_n1e5_p1024():
    """BASELINE.json configs[1]: d=10, n=1e5, p=1024, mat25."""
    import torch
    from outerbase_amd.driver import HotPath
    hp = HotPath(["mat25"] * 10, 40, 1024, 100_000)
    hp.setup()
    hp.step()
    torch.cuda.synchronize()
    try:
        _fit_is_stationary(hp, 1e-10)
    finally:
        hp.close()


def test_config3_one_rank_shard_d20_1p25e6_rows_p4096():
    """BASELINE.json configs[3] (d=20, n=1e7, p=4096, rows sharded over 8 GPUs): one rank's
    1.25e6-row shard (rank 3's rows of the stream) through the same path as the headline --
    41 GB design matrix, Gram, Cholesky, fused predictor -- checked by the properties that
    hold at any size.  The 8-rank sum itself is what tests/test_00_two_rank_device.py
    exercises on the device path."""
    import torch
    from outerbase_amd.driver import HotPath, shard_rows
    row0, n = shard_rows(3, 8, 10_000_000)
    assert (row0, n) == (3_750_000, 1_250_000)
    hp = HotPath(["mat25"] * 20, 40, 4096, n, row0=row0, n_total=n)
    hp.setup()
    hp.step()
    torch.cuda.synchronize()
    try:
        _fit_is_stationary(hp, 1e-10)
        y = hp.standardised_targets()
        assert abs(float(y.mean())) < 1e-12 and abs(float(y.var()) - 1.0) < 1e-11
    finally:
        hp.close()


def test_config4_one_rank_shard_d40_p16384_mixed_covariances():
    """BASELINE.json configs[4] (d=40, n=1e6, p=16384, mixed covariance functions over 8
    GPUs): the 125 000-row shard of one rank -- p = 16384 through the Gram, Cholesky and
    predict kernels (8256 tile pairs, 256 panel steps, all three covariances)."""
    import torch
    from outerbase_amd.driver import HotPath
    kinds = (["mat25", "mat25pow", "mat25ang"] * 14)[:40]
    hp = HotPath(kinds, 40, 16384, 125_000)
    hp.setup()
    hp.step()
    torch.cuda.synchronize()
    try:
        assert hp.terms_info["max_nnz"] <= 8
        # high levels of the periodic kernel: the conditioning note of test_gpu_parity applies
        _fit_is_stationary(hp, 1e-8)
    finally:
        hp.close()


def test_config0_borehole_obfit():
    """BASELINE.json configs[0]: Borehole d=8, n=1000, p=256 through obfit / obpred."""
    import ob_oracle as O
    import outerbase_amd as ob
    rng = np.random.default_rng(0)
    x = rng.random((1000, 8))
    y = O.borehole8d(x)
    m = ob.obfit(x, y, numb=256, seed=0)
    xt = rng.random((500, 8))
    pred = ob.obpred(m, xt)
    yt = O.borehole8d(xt)
    assert math.sqrt(np.mean((pred["mean"] - yt) ** 2)) < 0.01 * np.std(yt)
    assert np.all(pred["var"] > 0)
