"""The two CPU implementations under oracle/ -- the NumPy restatement and the
C++/OpenMP one that keeps the reference's chunk schedule -- must agree before
either is used to judge the GPU path.  Covers both
the tall (row-chunk) and the wide (term-parallel) branch of the schedule."""
import subprocess
import os

import numpy as np
import pytest

import ob_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cpu():
    import ob_cpu
    if not ob_cpu.available():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ob_cpu


@pytest.mark.parametrize("n,p,threads", [(100, 40, 4), (3000, 200, 4), (700, 300, 1)])
def test_cpp_port_matches_numpy_oracle(cpu, n, p, threads):
    kinds = ["mat25", "mat25pow", "mat25ang", "mat25"]
    om = O.OuterMod()
    om.setcovfs(kinds)
    om.setknot(O.bench_knots(kinds, 30))
    x, _ = O.synth_xy(1, 0, n, kinds)
    terms = om.selectterms(p)
    ob = O.OuterBase(om, x)
    bm, bs = cpu.build(om, x, threads)
    lead = np.concatenate([np.arange(om.knotptst[l], om.knotptst[l] + 6) for l in range(4)])
    assert np.allclose(bm[:, lead], ob.basemat[:, lead], rtol=1e-9, atol=1e-9)
    assert np.allclose(bs, ob.basescale, rtol=1e-13)
    rng = np.random.default_rng(n)
    a, v = rng.standard_normal(p), rng.standard_normal(n)
    B = O.ob_getmat(ob, terms)
    sc = np.abs(B).max()
    assert np.max(np.abs(cpu.getmat(om, terms, bm, bs, threads) - B)) < 1e-9 * sc
    assert np.max(np.abs(cpu.mm(om, terms, bm, bs, a, threads) - B @ a)) < 1e-9 * np.abs(B @ a).max()
    assert np.max(np.abs(cpu.tmm(om, terms, bm, bs, v, threads) - B.T @ v)) < 1e-9 * np.abs(B.T @ v).max()
