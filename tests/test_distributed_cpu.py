"""N > 1 path on CPU: two gloo ranks shard the rows of the synthetic stream and exchange
ONE buffer per fit.  Two contracts are rehearsed: the ABI-3 flow the driver runs
(obhip_standardise_dev + obhip_fit_newton_sharded_dev: 24 bytes for mean / sd of y, then
[packed upper triangle of G_r][B_r^T y_r of the standardised y][padding];
test_two_rank_abi3_flow_equals_single_process) and the ABI-2 composition that is still exported
(include/obhip.h, obhip_normal_eq_exchange_dev): [packed upper triangle of G_r][B_r^T y_r][B_r^T 1]
[sum y_r, sum y_r^2, n_r][padding].  The local arithmetic comes from the CPU oracle here
(no GPU in this tier); what is checked is the sharding contract of SURVEY.md section 8e:
contiguous row blocks by rank, the buffer size the library reports, that the summed buffer
reproduces the single-process fit INCLUDING the standardisation of y over all rows
(B^T ((y - cent) / sca) = (B^T y - cent B^T 1) / sca), replicated solve,
communication-free prediction.  The device path itself runs with two ranks in
tests/test_00_two_rank_device.py (-m gpu).
"""
import ctypes as C
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ob_oracle as O

KINDS = ["mat25", "mat25pow", "mat25", "mat25ang"]
ROWS_TOTAL = 601          # ragged: 300 + 301
P = 60


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    om = O.OuterMod()
    om.setcovfs(KINDS)
    om.setknot(O.bench_knots(KINDS, 24))
    return om


def pack(G, g, b1, sums, nloc, count):
    """the exchange buffer of one rank (numpy restatement of k_pack_tri / k_pack_tail)"""
    p = len(g)
    iu = np.triu_indices(p)                   # row-major upper triangle, j >= i
    buf = np.zeros(count)
    tri = p * (p + 1) // 2
    buf[:tri] = G[iu]
    buf[tri:tri + p] = g
    buf[tri + p:tri + 2 * p] = b1
    buf[tri + 2 * p:tri + 2 * p + 3] = [sums[0], sums[1], nloc]
    return buf


def unpack(buf, p):
    """-> G (full symmetric), standardised right-hand side, cent, sca, n
    (k_unpack_tri / k_finalize_rhs)"""
    tri = p * (p + 1) // 2
    G = np.zeros((p, p))
    iu = np.triu_indices(p)
    G[iu] = buf[:tri]
    G = G + np.triu(G, 1).T
    g, b1 = buf[tri:tri + p], buf[tri + p:tri + 2 * p]
    s1, s2, n = buf[tri + 2 * p:tri + 2 * p + 3]
    cent = s1 / n
    sca = math.sqrt(max(s2 - n * cent * cent, 0.0) / (n - 1.0))
    return G, (g - cent * b1) / sca, cent, sca, n


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    from outerbase_amd import _lib, driver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om = _model()
        terms = om.selectterms(P)
        row0, nrow = driver.shard_rows(rank, world, ROWS_TOTAL)
        x, y = O.synth_xy(42, row0, nrow, KINDS)
        ob = O.OuterBase(om, x)
        G, g = O.gram(ob, terms, y)                       # raw y: B_r^T y_r
        _, b1 = O.gram(ob, terms, np.ones(nrow))
        cnt = C.c_uint64(0)
        _lib.call("obhip_normal_eq_count", P, world, C.byref(cnt))
        buf = pack(G, g, b1, (y.sum(), (y * y).sum()), nrow, cnt.value)
        # the library's communicator object with the host transport; its collective needs
        # device memory, so here the same callback sums the host buffer directly
        comm, cb = driver.make_comm(rank, world, "host")
        nr, rk, tr = C.c_int(), C.c_int(), C.c_int()
        _lib.call("obhip_comm_info", comm, C.byref(nr), C.byref(rk), C.byref(tr), None, None)
        assert (nr.value, rk.value, tr.value) == (world, rank, 2)
        assert cb(None, buf.ctypes.data, len(buf)) == 0
        _lib.call("obhip_comm_destroy", comm)
        Gt, gt, cent, sd, ntot = unpack(buf, P)
        sigma = math.log(0.01)
        H = math.exp(-2 * sigma) * Gt + np.diag(O.prior_prec(om, terms, O.DEFAULT_RHO))
        theta = np.linalg.solve(H, math.exp(-2 * sigma) * gt)
        xnew, _ = O.synth_xy(43, row0, 20, KINDS)
        mean = cent + sd * O.predict_mean(om, terms, theta, xnew)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), theta=theta, mean=mean, cent=cent,
                 sd=sd, ntot=ntot, row0=row0, count=cnt.value)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_row_sharding_equals_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["row0"]) == 0 and int(r1["row0"]) == 300 and int(r0["ntot"]) == ROWS_TOTAL
    # equal blocks for the reduce-scatter: a multiple of 2 * world doubles
    assert int(r0["count"]) % (2 * world) == 0
    assert int(r0["count"]) >= P * (P + 1) // 2 + 2 * P + 3
    # replicated solve: identical on both ranks
    assert np.array_equal(r0["theta"], r1["theta"])
    # single-process reference on all rows
    om = _model()
    terms = om.selectterms(P)
    x, y = O.synth_xy(42, 0, ROWS_TOTAL, KINDS)
    cent, sd = y.mean(), y.std(ddof=1)
    assert abs(r0["cent"] - cent) < 1e-12 * abs(cent) and abs(r0["sd"] - sd) < 1e-12 * sd
    y = (y - cent) / sd
    theta, _ = O.fit_newton(O.OuterBase(om, x), terms, y, sigma=math.log(0.01))
    assert np.max(np.abs(r0["theta"] - theta)) < 1e-8 * np.max(np.abs(theta))
    for r in (r0, r1):
        xnew, _ = O.synth_xy(43, int(r["row0"]), 20, KINDS)
        want = cent + sd * O.predict_mean(om, terms, theta, xnew)
        assert np.max(np.abs(r["mean"] - want)) < 1e-6 * np.max(np.abs(want))


def _worker_v3(rank, world, port, out_dir):
    """One rank of the ABI-3 flow (obhip_standardise_dev + obhip_fit_newton_sharded_dev) with the
    local arithmetic from the CPU oracle: (sum y, n) summed -> mean; sum (y - mean)^2 summed ->
    sd (two-pass, like R's sd()); y standardised BEFORE B^T y; ONE buffer [packed upper triangle
    of G_r][B_r^T y_r][zero padding] summed; H = e^{-2 sigma} G + prior formed while unpacking."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    from outerbase_amd import _lib, driver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om = _model()
        terms = om.selectterms(P)
        row0, nrow = driver.shard_rows(rank, world, ROWS_TOTAL)
        x, y = O.synth_xy(42, row0, nrow, KINDS)
        comm, cb = driver.make_comm(rank, world, "host")

        def allsum(vals):
            v = np.ascontiguousarray(vals, dtype=np.float64)
            assert cb(None, v.ctypes.data, len(v)) == 0
            return v
        s1, ntot = allsum([y.sum(), float(nrow)])              # 16 bytes
        cent = s1 / ntot
        sd = math.sqrt(allsum([np.sum((y - cent) ** 2)])[0] / (ntot - 1.0))   # 8 bytes
        ys = (y - cent) / sd
        G, g = O.gram(O.OuterBase(om, x), terms, ys)
        cnt = C.c_uint64(0)
        _lib.call("obhip_fit_newton_count", P, world, C.byref(cnt))
        tri = P * (P + 1) // 2
        buf = np.zeros(cnt.value)
        buf[:tri] = G[np.triu_indices(P)]                       # what k_gram_reduce writes (packed)
        buf[tri:tri + P] = g                                    # what the staging pass writes
        allsum_buf = allsum(buf)
        _lib.call("obhip_comm_destroy", comm)
        sigma = math.log(0.01)
        e2 = math.exp(-2 * sigma)
        H = np.zeros((P, P))                                    # k_unpack_form
        H[np.triu_indices(P)] = e2 * allsum_buf[:tri]
        H = H + np.triu(H, 1).T
        H[np.diag_indices(P)] += O.prior_prec(om, terms, O.DEFAULT_RHO)
        theta = np.linalg.solve(H, e2 * allsum_buf[tri:tri + P])
        xnew, _ = O.synth_xy(43, row0, 20, KINDS)
        mean = cent + sd * O.predict_mean(om, terms, theta, xnew)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), theta=theta, mean=mean, cent=cent,
                 sd=sd, ntot=ntot, row0=row0, count=cnt.value, pad=allsum_buf[tri + P:])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_abi3_flow_equals_single_process(tmp_path):
    """The exchange contract of obhip_fit_newton_sharded_dev under gloo: 24 bytes for the
    standardisation, one [triangle | B^T y] buffer, replicated solve."""
    world = 2
    port = _free_port()
    mp.spawn(_worker_v3, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["ntot"]) == ROWS_TOTAL and int(r0["count"]) % (2 * world) == 0
    assert int(r0["count"]) >= P * (P + 1) // 2 + P and not r0["pad"].any()
    assert np.array_equal(r0["theta"], r1["theta"])
    om = _model()
    terms = om.selectterms(P)
    x, y = O.synth_xy(42, 0, ROWS_TOTAL, KINDS)
    cent, sd = y.mean(), y.std(ddof=1)
    assert abs(r0["cent"] - cent) < 1e-14 * abs(cent) and abs(r0["sd"] - sd) < 1e-14 * sd
    theta, _ = O.fit_newton(O.OuterBase(om, x), terms, (y - cent) / sd, sigma=math.log(0.01))
    assert np.max(np.abs(r0["theta"] - theta)) < 1e-9 * np.max(np.abs(theta))
    for r in (r0, r1):
        xnew, _ = O.synth_xy(43, int(r["row0"]), 20, KINDS)
        want = cent + sd * O.predict_mean(om, terms, theta, xnew)
        assert np.max(np.abs(r["mean"] - want)) < 1e-9 * np.max(np.abs(want))


def test_shards_cover_the_rows_exactly_once():
    from outerbase_amd import driver
    for n, w in [(1_000_000, 8), (10_000_000, 8), (6001, 2), (7, 3), (1_000_000, 3)]:
        blocks = [driver.shard_rows(r, w, n) for r in range(w)]
        assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == n
        for a, b in zip(blocks, blocks[1:]):
            assert a[0] + a[1] == b[0]
        assert max(b[1] for b in blocks) - min(b[1] for b in blocks) <= 1


def test_getsteps_matches_reference_formula():
    from outerbase_amd import driver, fitting
    for numb, n, ratio in [(4096, 1e6, 1e4), (300, 400, 1e-3), (100, 100000, 5.0),
                           (500, 500, 1e-3)]:       # numb == n: R's min(1000, Inf) = 1000
        assert driver.getsteps(numb, n, ratio) == O.getsteps(numb, n, ratio)
        assert fitting._getsteps(numb, n, ratio) == O.getsteps(numb, n, ratio)
