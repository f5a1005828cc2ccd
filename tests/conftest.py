import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    # device_count() does not initialise the HIP runtime in this process (is_available()
    # does): tests that start their own rank processes must be able to do so from a parent
    # that has not touched the GPU yet
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


KNOTS_REF = np.arange(0.001, 0.999, 0.025)  # tests/testthat/test-obombasic.R:34


def make_pair(kinds, knotlist, hyp=None, share_rotation=True):
    """(oracle OuterMod, device outermod) on the same covs / knots / hyp.  With
    share_rotation the device model takes the oracle's eigen-decomposition so
    that parity is about the kernels, not about LAPACK vs Jacobi noise in
    near-null eigenvectors (SURVEY.md section 7 'Hard parts')."""
    import ob_oracle as O
    import outerbase_amd as ob
    om_o = O.OuterMod()
    om_o.setcovfs(kinds)
    if hyp is not None:
        om_o.hyp_set(hyp)
    om_o.setknot(knotlist)
    om_d = ob.outermod()
    ob.setcovfs(om_d, kinds)
    if hyp is not None:
        om_d.updatehyp(hyp)
    ob.setknot(om_d, knotlist)
    if share_rotation:
        om_d.set_rotation(om_o.rotmat, om_o.basisvar, om_o.maxlevel)
        om_d.set_rotation_grad(om_o.rotmat_gradhyp, om_o.logbasisvar_gradhyp)
    return om_o, om_d


def knots_for(kinds, m=40):
    import ob_oracle as O
    return O.bench_knots(kinds, m)


def sample_x(rng, n, kinds):
    x = 0.02 + 0.96 * rng.random((n, len(kinds)))
    for j, k in enumerate(kinds):
        if k == "mat25ang":
            x[:, j] *= 6.283185
    return x
