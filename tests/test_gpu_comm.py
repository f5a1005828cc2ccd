"""libobhip's communicator on one GPU: the RCCL transport with a single rank (every RCCL
entry point the N > 1 path uses is called -- ncclGetUniqueId, ncclCommInitRank,
ncclCommCount, ncclReduceScatter + ncclAllGather, ncclAllReduce -- and a one-rank sum is
the identity), and the pack / unpack kernels of the one-buffer exchange at the headline
p = 4096 against a NumPy restatement of the layout."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _comm_rccl_single():
    from outerbase_amd._lib import call
    uid = np.zeros(128, dtype=np.uint8)
    call("obhip_comm_unique_id", uid.ctypes.data)
    assert uid.any()
    h = C.c_void_p()
    call("obhip_comm_init", C.byref(h), 1, 0, uid.ctypes.data)
    return h


@pytest.mark.parametrize("p", [1, 63, 300, 4096])
def test_exchange_roundtrip_through_rccl_single_rank(p):
    import torch
    from outerbase_amd._lib import call, lib
    rng = np.random.default_rng(p)
    A = rng.standard_normal((p, p))
    G = A + A.T
    g, b1 = rng.standard_normal(p), rng.standard_normal(p)
    n = 5000.0
    y = 3.0 + 2.0 * rng.standard_normal(int(n))
    sums = np.array([y.sum(), (y * y).sum()])
    comm = _comm_rccl_single()
    try:
        nr, rk, tr, rr, rv = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        call("obhip_comm_info", comm, C.byref(nr), C.byref(rk), C.byref(tr), C.byref(rr), C.byref(rv))
        assert (nr.value, rk.value, tr.value, rr.value) == (1, 0, 1, 1) and rv.value > 0
        cnt = C.c_uint64(0)
        call("obhip_normal_eq_count", p, 1, C.byref(cnt))
        dG = torch.from_numpy(G).cuda()
        dg, db1 = torch.from_numpy(g).cuda(), torch.from_numpy(b1).cuda()
        ds = torch.from_numpy(sums).cuda()
        buf = torch.full((cnt.value,), float("nan"), dtype=torch.float64, device="cuda")
        ms = torch.zeros(3, dtype=torch.float64, device="cuda")
        # scribble over the lower triangle: the unpack must restore it from the upper one
        dG2 = dG.clone()
        call("obhip_normal_eq_exchange_dev", comm, p, int(n), dG2.data_ptr(), dg.data_ptr(),
             db1.data_ptr(), ds.data_ptr(), buf.data_ptr(), cnt.value, ms.data_ptr())
        torch.cuda.synchronize()
        hb = buf.cpu().numpy()
        tri = p * (p + 1) // 2
        assert np.array_equal(hb[:tri], G[np.triu_indices(p)])
        assert np.array_equal(hb[tri:tri + p], g) and np.array_equal(hb[tri + p:tri + 2 * p], b1)
        assert np.array_equal(hb[tri + 2 * p:tri + 2 * p + 3], [sums[0], sums[1], n])
        assert not np.isnan(hb).any() and not hb[tri + 2 * p + 3:].any()
        assert np.array_equal(dG2.cpu().numpy(), G)
        cent, sd = y.mean(), y.std(ddof=1)
        got = ms.cpu().numpy()
        assert abs(got[0] - cent) < 1e-13 * abs(cent) and abs(got[1] - sd) < 1e-12 * sd and got[2] == n
        want = (g - got[0] * b1) / got[1]
        assert np.max(np.abs(dg.cpu().numpy() - want)) < 1e-14 * np.max(np.abs(want))
        # plain in-place sums of other sizes (reduce-scatter + all-gather above 4096 doubles
        # per rank, all-reduce below and for ragged counts)
        for count in (3, 4097, 8192, 1 << 20):
            v = torch.from_numpy(rng.standard_normal(count)).cuda()
            w = v.clone()
            call("obhip_comm_allreduce_dev", comm, w.data_ptr(), count)
            torch.cuda.synchronize()
            assert torch.equal(v, w)
    finally:
        lib.obhip_comm_destroy(comm)


def test_exchange_without_communicator_only_standardises():
    import torch
    from outerbase_amd._lib import call
    p = 257
    rng = np.random.default_rng(1)
    G = rng.standard_normal((p, p))
    g, b1 = rng.standard_normal(p), rng.standard_normal(p)
    y = 10.0 + rng.standard_normal(999)
    cnt = C.c_uint64(0)
    call("obhip_normal_eq_count", p, 1, C.byref(cnt))
    tri = p * (p + 1) // 2
    dG = torch.from_numpy(G).cuda()
    dg, db1 = torch.from_numpy(g).cuda(), torch.from_numpy(b1).cuda()
    ds = torch.tensor([y.sum(), (y * y).sum()], dtype=torch.float64, device="cuda")
    tail = torch.zeros(cnt.value - tri, dtype=torch.float64, device="cuda")
    ms = torch.zeros(3, dtype=torch.float64, device="cuda")
    call("obhip_normal_eq_exchange_dev", None, p, len(y), dG.data_ptr(), dg.data_ptr(),
         db1.data_ptr(), ds.data_ptr(), tail.data_ptr() - 8 * tri, cnt.value, ms.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dG.cpu().numpy(), G)          # untouched, not even symmetrised
    cent, sd = y.mean(), y.std(ddof=1)
    want = (g - cent * b1) / sd
    assert np.max(np.abs(dg.cpu().numpy() - want)) < 1e-11 * np.max(np.abs(want))
    assert abs(float(ms[1]) - sd) < 1e-10 * sd


def test_one_rank_host_communicator_without_callback_is_the_identity():
    """obhip_comm_init_host accepts fn = NULL for one rank; every exchange is then a no-op
    (round 2 called the null pointer)."""
    import torch
    from outerbase_amd._lib import call, lib
    h = C.c_void_p()
    call("obhip_comm_init_host", C.byref(h), 1, 0, None, None)
    try:
        v = torch.arange(5000, dtype=torch.float64, device="cuda")
        w = v.clone()
        call("obhip_comm_allreduce_dev", h, w.data_ptr(), 5000)
        torch.cuda.synchronize()
        assert torch.equal(v, w)
        # a two-rank communicator still needs its callback
        h2 = C.c_void_p()
        assert lib.obhip_comm_init_host(C.byref(h2), 2, 0, None, None) != 0
    finally:
        lib.obhip_comm_destroy(h)


@pytest.mark.parametrize("shift", [0.0, 1e7])
def test_standardise_dev_is_two_pass(shift):
    """(y - mean) / sd with the n - 1 denominator (R/fitting.R:55-57) as R's sd() computes it:
    centred sum of squares, so a mean 1e7 standard deviations away costs no digits (the
    one-pass sum y^2 - n mean^2 of round 2 lost 14 of them there)."""
    import torch
    from outerbase_amd._lib import call
    rng = np.random.default_rng(3)
    y = shift + rng.standard_normal(70001)
    dy = torch.from_numpy(y).cuda()
    out = torch.empty_like(dy)
    ms = torch.zeros(3, dtype=torch.float64, device="cuda")
    call("obhip_standardise_dev", None, dy.data_ptr(), len(y), out.data_ptr(), ms.data_ptr())
    torch.cuda.synchronize()
    cent, sd = y.mean(), y.std(ddof=1)
    got = ms.cpu().numpy()
    assert abs(got[0] - cent) <= 1e-15 * abs(cent) + 1e-16 and abs(got[1] - sd) < 1e-12 * sd and got[2] == len(y)
    want = (y - cent) / sd
    assert np.max(np.abs(out.cpu().numpy() - want)) < (1e-8 if shift else 1e-13)
    back = out.clone()
    call("obhip_destandardise_dev", back.data_ptr(), len(y), ms.data_ptr())
    torch.cuda.synchronize()
    assert np.max(np.abs(back.cpu().numpy() - y)) < 1e-15 * (abs(shift) + 10)


@pytest.mark.parametrize("n,p,chunk", [(3000, 300, 0), (1500, 129, 0), (5000, 260, 1024), (700, 1, 0)])
def test_fit_newton_sharded_entry_equals_the_composed_calls(n, p, chunk, monkeypatch):
    """obhip_fit_newton_sharded_dev -- the Gram reduction writing H (one rank) or the packed
    exchange triangle (communicator), the unpack forming H -- against the round-2 composition
    obhip_gram_dev + obhip_newton_solve_dev (k_gram_reduce to full storage, k_form_hessian) and
    against the oracle; also with the design matrix staged in row chunks."""
    import torch
    import ob_oracle as O
    from conftest import make_pair, knots_for
    from outerbase_amd._lib import call, lib
    from outerbase_amd import obmod
    kinds = ["mat25", "mat25pow", "mat25ang", "mat25"]
    om_o, om = make_pair(kinds, knots_for(kinds, 20))
    terms = om_o.selectterms(p)
    t = obmod._Terms(om, terms)
    x, y = O.synth_xy(7, 0, n, kinds)
    y = (y - y.mean()) / y.std(ddof=1)
    sigma, rho = math.log(0.01), 6.0
    dx = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    dy = torch.from_numpy(y).cuda()
    if chunk:
        monkeypatch.setenv("OBHIP_GRAM_CHUNK_ROWS", str(chunk))
    basis = C.c_void_p()
    call("obhip_basis_create_dev", C.byref(basis), om._h, dx.data_ptr(), n, t.maxlevels().ctypes.data)
    wsb = C.c_uint64(0)
    call("obhip_newton_workspace_bytes", p, C.byref(wsb))
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")

    def bufs():
        return (torch.full((p, p), float("nan"), dtype=torch.float64, device="cuda"),
                torch.empty(p, dtype=torch.float64, device="cuda"),
                torch.empty(p, dtype=torch.float64, device="cuda"),
                torch.empty(p, dtype=torch.float64, device="cuda"))
    try:
        # round-2 composition
        G0, g0, th0, dh0 = bufs()
        call("obhip_gram_dev", basis, t._h, dy.data_ptr(), G0.data_ptr(), g0.data_ptr())
        Graw = G0.cpu().numpy().copy()
        call("obhip_newton_solve_dev", om._h, t._h, G0.data_ptr(), g0.data_ptr(), sigma, rho,
             th0.data_ptr(), dh0.data_ptr(), ws.data_ptr(), wsb.value)
        # one rank, no communicator: the reduction writes H
        G1, g1, th1, dh1 = bufs()
        call("obhip_fit_newton_sharded_dev", None, basis, t._h, om._h, dy.data_ptr(), sigma, rho,
             G1.data_ptr(), g1.data_ptr(), th1.data_ptr(), dh1.data_ptr(), None, 0, ws.data_ptr(), wsb.value)
        torch.cuda.synchronize()
        assert torch.equal(th0, th1) and torch.equal(dh0, dh1) and torch.equal(g0, g1)
        # a one-rank communicator: packed triangle -> exchange (identity) -> unpack forms H
        comm = C.c_void_p()
        call("obhip_comm_init_host", C.byref(comm), 1, 0, None, None)
        cnt = C.c_uint64(0)
        call("obhip_fit_newton_count", p, 1, C.byref(cnt))
        ex = torch.zeros(cnt.value, dtype=torch.float64, device="cuda")
        G2, g2, th2, dh2 = bufs()
        call("obhip_fit_newton_sharded_dev", comm, basis, t._h, om._h, dy.data_ptr(), sigma, rho,
             G2.data_ptr(), g2.data_ptr(), th2.data_ptr(), dh2.data_ptr(), ex.data_ptr(), cnt.value,
             ws.data_ptr(), wsb.value)
        torch.cuda.synchronize()
        lib.obhip_comm_destroy(comm)
        tri = p * (p + 1) // 2
        hb = ex.cpu().numpy()
        assert np.array_equal(hb[:tri], Graw[np.triu_indices(p)])       # the raw G, packed
        assert np.array_equal(hb[tri:tri + p], g0.cpu().numpy()) and not hb[tri + p:].any()
        assert torch.equal(th0, th2) and torch.equal(dh0, dh2) and torch.equal(g0, g2)
        assert torch.equal(torch.tril(G0), torch.tril(G2))                 # the same factor
        # and the oracle
        theta_o, _ = O.fit_newton(O.OuterBase(om_o, x), terms, y, sigma=sigma)
        assert np.max(np.abs(th1.cpu().numpy() - theta_o)) < 1e-6 * np.max(np.abs(theta_o))
    finally:
        lib.obhip_basis_destroy(basis)


def _selftest(comm, count):
    from outerbase_amd._lib import lib
    res = (C.c_int64 * 4)()
    rc = lib.obhip_comm_selftest_dev(comm, count, C.cast(res, C.c_void_p))
    return rc, [int(v) for v in res]


def _path(comm, count):
    from outerbase_amd._lib import call
    path, st = C.c_int(), C.c_int()
    call("obhip_comm_exchange_path", comm, count, C.byref(path), C.byref(st))
    return path.value, st.value


@pytest.mark.parametrize("transport", ["rccl", "host", "sim"])
def test_comm_selftest_single_rank(transport):
    """obhip_comm_selftest_dev on a buffer of the headline exchange size (p = 4096: 8.4e6
    doubles): closed-form sums through the reduce-scatter / all-gather pair and through
    ncclAllReduce, compared on the device.  One rank, so the sums are the values themselves --
    what is checked here is the plumbing (both RCCL paths are called, the counts come back, the
    path in use is reported); N > 1 runs the same code from bench.py before its warm-up."""
    from outerbase_amd._lib import call, lib
    cnt = C.c_uint64(0)
    call("obhip_fit_newton_count", 4096, 1, C.byref(cnt))
    if transport == "rccl":
        comm = _comm_rccl_single()
    elif transport == "host":
        comm = C.c_void_p()
        call("obhip_comm_init_host", C.byref(comm), 1, 0, None, None)
    else:
        comm = C.c_void_p()
        call("obhip_comm_init_sim", C.byref(comm), 8)
    try:
        assert _path(comm, cnt.value)[1] == 0                       # not run yet
        rc, res = _selftest(comm, cnt.value)
        assert rc == 0, lib.obhip_last_error()
        want_path = {"rccl": 1, "host": 3, "sim": 4}[transport]
        assert res == [want_path, 0 if transport == "rccl" else -1, 0, 0]
        assert _path(comm, cnt.value) == (want_path, 1)
        # small buffers (the 24 bytes of the standardisation) take the plain all-reduce
        if transport == "rccl":
            assert _path(comm, 3)[0] == 2
            assert _selftest(comm, 3) == (0, [2, -1, 0, 0])
    finally:
        lib.obhip_comm_destroy(comm)


def test_comm_selftest_switches_a_wrong_pair_off_in_process():
    """The reduce-scatter / all-gather pair returning one wrong element while ncclAllReduce is
    right: the self-test reports the mismatch, switches this communicator to ncclAllReduce for
    every size without a relaunch, and the next exchange of the same buffer is exact.  The fault
    injector exists in the TEST build of the library only (libobhip_testing.so: comm.cpp compiled
    with -DOBHIP_TESTING, armed through obhip_testing_fault_inject_pair), so the case runs in a
    child process that loads that build (tests/fault_inject_worker.py); the shipping library has
    no such switch (test_host_logic.py::test_shipping_library_has_no_fault_injector)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OBHIP_TEST_LIBRARY="testing")
    r = subprocess.run([sys.executable, os.path.join(here, "fault_inject_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fault injection ok" in r.stdout


def test_comm_selftest_fails_loudly_when_the_transport_sums_wrongly():
    """A host transport whose callback loses one element: OBHIP_ERR_STATE with a message."""
    from outerbase_amd import _lib
    from outerbase_amd._lib import call, lib

    def bad_sum(user, host_ptr, count):
        buf = np.ctypeslib.as_array(C.cast(host_ptr, C.POINTER(C.c_double)), shape=(count,))
        buf[count // 3] = 0.0
        return 0
    cb = _lib.HOST_ALLREDUCE_FN(bad_sum)
    comm = C.c_void_p()
    call("obhip_comm_init_host", C.byref(comm), 1, 0, C.cast(cb, C.c_void_p), None)
    try:
        rc, _ = _selftest(comm, 50000)
        assert rc == 4                                                # OBHIP_ERR_STATE
        assert b"wrong sums" in lib.obhip_last_error()
    finally:
        lib.obhip_comm_destroy(comm)


def test_standardise_empty_shard_takes_part_in_both_sums():
    """A rank without rows (fewer rows than ranks, a ragged last shard) must still enter the
    two sums of obhip_standardise_dev, or its peers wait for ever (round-3 advice): with a
    communicator n = 0 is legal.  The peer here is the host callback, which adds the other
    rank's (sum y, n) and then its centred sum of squares."""
    import torch
    from outerbase_amd import _lib
    from outerbase_amd._lib import call, lib
    rng = np.random.default_rng(11)
    y = 5.0 + rng.standard_normal(1234)
    calls = []

    def peer(user, host_ptr, count):
        buf = np.ctypeslib.as_array(C.cast(host_ptr, C.POINTER(C.c_double)), shape=(count,))
        calls.append((count, buf.copy()))
        if count == 2:
            buf += [y.sum(), len(y)]
        else:
            buf += ((y - y.mean()) ** 2).sum()
        return 0
    cb = _lib.HOST_ALLREDUCE_FN(peer)
    comm = C.c_void_p()
    call("obhip_comm_init_host", C.byref(comm), 2, 0, C.cast(cb, C.c_void_p), None)
    try:
        ms = torch.zeros(3, dtype=torch.float64, device="cuda")
        call("obhip_standardise_dev", comm, None, 0, None, ms.data_ptr())
        torch.cuda.synchronize()
        assert [c for c, _ in calls] == [2, 1]
        assert not calls[0][1].any() and not calls[1][1].any()       # this rank contributed zeros
        got = ms.cpu().numpy()
        assert abs(got[0] - y.mean()) < 1e-14 * abs(y.mean()) and got[2] == len(y)
        assert abs(got[1] - y.std(ddof=1)) < 1e-13
        # without a communicator an empty input stays an error
        assert lib.obhip_standardise_dev(None, None, 0, None, ms.data_ptr()) == 1
    finally:
        lib.obhip_comm_destroy(comm)


def test_sharded_newton_rejects_terms_beyond_the_models_levels():
    """obhip_fit_newton_sharded_dev with terms whose levels the model does not have (made for a
    model with more knots): OBHIP_ERR_INVALID before anything indexes the model's tables
    (round-3 advice: the entry skipped check_compat)."""
    import torch
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd import obmod
    from outerbase_amd._lib import call, lib
    kinds = ["mat25"] * 3
    big, small = ob.outermod(), ob.outermod()
    for om, m in ((big, 40), (small, 8)):
        ob.setcovfs(om, kinds)
        ob.setknot(om, O.bench_knots(kinds, m))
    terms = big.selectterms(200)
    assert terms.max() >= 8                                          # beyond the small model
    t_big = obmod._Terms(big, terms)
    x, y = O.synth_xy(3, 0, 500, kinds)
    dx = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    dy = torch.from_numpy(y).cuda()
    basis = C.c_void_p()
    call("obhip_basis_create_dev", C.byref(basis), small._h, dx.data_ptr(), 500, None)
    p = 200
    wsb = C.c_uint64(0)
    call("obhip_newton_workspace_bytes", p, C.byref(wsb))
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")
    H = torch.empty((p, p), dtype=torch.float64, device="cuda")
    g, th = (torch.empty(p, dtype=torch.float64, device="cuda") for _ in range(2))
    try:
        rc = lib.obhip_fit_newton_sharded_dev(None, basis, t_big._h, small._h, dy.data_ptr(), math.log(0.01),
                                              6.0, H.data_ptr(), g.data_ptr(), th.data_ptr(), None, None, 0,
                                              ws.data_ptr(), wsb.value)
        assert rc == 1 and b"beyond the model" in lib.obhip_last_error()
    finally:
        lib.obhip_basis_destroy(basis)


def test_sim_ranks_fit_is_the_fit_of_the_shard_repeated():
    """obhip_comm_init_sim(N): N virtual ranks that all hold this process's rows -- the fit must
    be the one-rank fit of those rows repeated N times (same mean / sd up to the n - 1
    denominator, G and B^T y N-fold), through the real exchange-buffer layout and unpack."""
    import torch
    from outerbase_amd.driver import HotPath
    kinds = ["mat25", "mat25pow", "mat25"]
    N, n, p = 4, 3000, 150
    sim = HotPath(kinds, 20, p, n, rank=0, world=N, transport="sim", row0=0, n_total=N * n)
    sim.setup()
    assert sim.comm_info()["path"].startswith("sim")
    assert sim.comm_selftest()["allreduce_mismatches"] == 0
    sim.step()
    torch.cuda.synchronize()
    one = HotPath(kinds, 20, p, N * n, terms=sim.terms)
    one.setup()
    # the same rows N times: x, xnew, y of the one-rank job are the shard tiled
    one.x.copy_(sim.x.repeat(1, N))
    one.xnew.copy_(sim.xnew.repeat(1, N))
    one.y_raw.copy_(sim.y_raw.repeat(N))
    one.step()
    torch.cuda.synchronize()
    assert abs(one.y_cent - sim.y_cent) < 1e-13 * abs(sim.y_cent) and abs(one.y_sca - sim.y_sca) < 1e-12
    ts, to = sim.theta.cpu().numpy(), one.theta.cpu().numpy()
    assert np.max(np.abs(ts - to)) < 1e-8 * np.max(np.abs(to))
    ms, mo = sim.mean.cpu().numpy(), one.mean[:n].cpu().numpy()
    assert np.max(np.abs(ms - mo)) < 1e-9 * np.max(np.abs(mo))
    assert sim.newton_residual_rel() < 1e-10
    sim.close()
    one.close()
