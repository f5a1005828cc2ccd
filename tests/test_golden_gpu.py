"""Device path against the committed fixtures (tests/golden/*.npz, written by
oracle/gen_golden.py): every array of the hot path, all three kernels."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_device_reproduces_golden(path):
    import outerbase_amd as ob
    g = np.load(path)
    kinds = [str(k) for k in g["kinds"]]
    st = g["knotptst"]
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    om.updatehyp(g["hyp"])
    ob.setknot(om, [g["knotpt"][st[l]:st[l + 1]] for l in range(len(kinds))])
    om.set_rotation(g["rotmat"], g["basisvar"], g["maxlevel"])
    terms = g["terms"]
    assert np.array_equal(om.selectterms(terms.shape[0]), terms)
    assert relerr(om.getvar(terms), g["termvar"]) < 1e-14
    b = ob.outerbase(om, g["x"])
    for k in range(1, len(kinds) + 1):
        want = g["basemat_lead"][:, k - 1, :6] * g["basescalemat"][:, k - 1:k]
        assert relerr(b.getbase(k)[:, :6], want) < 2e-9
    assert relerr(b.getmat(terms), g["B"]) < 2e-9
    assert relerr(b.matmul(terms, g["a"]), g["Ba"]) < 2e-9
    assert relerr(b.tmatmul(terms, g["v"]), g["Btv"]) < 2e-9
    assert relerr(b.sqmm(terms, np.abs(g["a"])), g["sqBa"]) < 2e-9
    assert relerr(b.sqcolsums(terms), g["sqcolsums"]) < 2e-9
    lik = ob.loglik_std(om, terms, g["y"], g["x"])
    assert abs(lik.para[0] - float(g["sigma"])) < 1e-13
    lp = ob.lpdfvec(lik, ob.logpr_gauss(om, terms))
    assert relerr(lp.hess(), g["H"]) < 2e-9
    lp.optnewton()
    pred = ob.predictor(lp)
    pred.update(g["xnew"])
    assert relerr(pred.mean(), g["mean"]) < 1e-6          # north_star tolerance
    assert relerr(pred.var(), g["var_std"]) < 1e-7         # predr_std: b^T inv(H) b + e^{2 sigma}
    likg = ob.loglik_gauss(om, terms, g["y"], g["x"])
    lpg = ob.lpdfvec(ob.logpr_gauss(om, terms), likg)
    lpg.optcg(0.0, 12)   # tol 0: exactly 12 iterations on both sides
    assert lpg.cgiters == int(g["cg_iters"])
    assert relerr(lpg.totdiaghess, g["diagH"]) < 2e-9
    assert relerr(likg.yhat, g["B"] @ g["theta_cg"]) < 1e-6
    predg = ob.predictor(lpg)                              # pred_gauss: B^2 / diag(H) + e^{2 sigma}
    predg.update(g["xnew"])
    assert relerr(predg.var(), g["var_gauss"]) < 1e-9
