"""N > 1 path on CPU: two gloo ranks shard the rows of the synthetic stream, and
the orchestration helpers of outerbase_amd.driver (the ones HotPath runs over
RCCL on a GPU node) merge their statistics and normal equations.  Local
arithmetic comes from the CPU oracle here -- the point is the sharding contract
(SURVEY.md section 8e): row blocks by rank, global standardisation of y, one
sum of G and g, replicated solve, communication-free prediction.
"""
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
ROWS = 300
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


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    from outerbase_amd import driver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om = _model()
        terms = om.selectterms(P)
        row0, nrow = driver.shard_rows(rank, ROWS)
        x, y = O.synth_xy(42, row0, nrow, KINDS)
        y = y.copy()

        def reduce_floats(vals):
            t = torch.tensor(vals, dtype=torch.float64)
            dist.all_reduce(t)
            return t.tolist()

        def centre_and_sumsq(cent):
            y[:] = y - cent
            return float(np.sum(y * y))

        cent, sd = driver.global_standardise(lambda: float(np.sum(y)), centre_and_sumsq,
                                             float(ROWS * world), reduce_floats)
        y /= sd
        ob = O.OuterBase(om, x)
        G, g = O.gram(ob, terms, y)
        Gt, gt = torch.from_numpy(G), torch.from_numpy(g)
        driver.merge_normal_equations(Gt, gt, dist.all_reduce)
        sigma = math.log(0.01)
        H = math.exp(-2 * sigma) * Gt.numpy() + np.diag(O.prior_prec(om, terms, O.DEFAULT_RHO))
        theta = np.linalg.solve(H, math.exp(-2 * sigma) * gt.numpy())
        xnew, _ = O.synth_xy(43, row0, 20, KINDS)
        mean = cent + sd * O.predict_mean(om, terms, theta, xnew)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), theta=theta, mean=mean, cent=cent, sd=sd)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_row_sharding_equals_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # replicated solve: identical on both ranks
    assert np.array_equal(r0["theta"], r1["theta"])
    # single-process reference on all 2 * ROWS rows
    om = _model()
    terms = om.selectterms(P)
    x, y = O.synth_xy(42, 0, ROWS * world, KINDS)
    cent, sd = y.mean(), y.std(ddof=1)
    assert abs(r0["cent"] - cent) < 1e-12 * abs(cent) and abs(r0["sd"] - sd) < 1e-12 * sd
    y = (y - cent) / sd
    theta, _ = O.fit_newton(O.OuterBase(om, x), terms, y, sigma=math.log(0.01))
    for rank, r in enumerate((r0, r1)):
        xnew, _ = O.synth_xy(43, rank * ROWS, 20, KINDS)
        want = cent + sd * O.predict_mean(om, terms, theta, xnew)
        assert np.max(np.abs(r["mean"] - want)) < 1e-6 * np.max(np.abs(want))


def test_getsteps_matches_reference_formula():
    from outerbase_amd import driver
    for numb, n, ratio in [(4096, 1e6, 1e4), (300, 400, 1e-3), (100, 100000, 5.0)]:
        assert driver.getsteps(numb, n, ratio) == O.getsteps(numb, n, ratio)
