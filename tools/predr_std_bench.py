"""predr_std at n' = 1e5, p = 4096 (d = 20): time of the posterior-variance path
(loglik_std.cpp:249-256) after optnewton -- the p^2 n' flop of || L^-1 b_i ||^2."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import outerbase_amd as ob
from outerbase_amd import _lib
from outerbase_amd.driver import bench_knots

nnew = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d, n = 20, 20000
kinds = ["mat25"] * d
om = ob.outermod(); ob.setcovfs(om, kinds); ob.setknot(om, bench_knots(kinds, 40))
terms = om.selectterms(p)
rng = np.random.default_rng(0)
x = 0.02 + 0.96 * rng.random((n, d)); y = rng.standard_normal(n)
lik = ob.loglik_std(om, terms, y, x)
lp = ob.lpdfvec(lik, ob.logpr_gauss(om, terms))
lp.domarg = False
lp.optnewton()
pred = ob.predictor(lp)
xnew = 0.02 + 0.96 * rng.random((nnew, d))
for it in range(2):
    pred.update(xnew)
    _lib.call("obhip_profile_reset"); _lib.call("obhip_profile_enable", 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = pred.var()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out = {}
    for name in ("materialize_B", "cholesky", "predict_std_trsm", "predict_std_inverse", "predict_std_gemm", "predict", "getmat"):
        cnt, ms = C.c_uint64(), C.c_double()
        _lib.call("obhip_profile_get", name.encode(), C.byref(cnt), C.byref(ms))
        if cnt.value: out[name] = round(ms.value, 3)
    _lib.call("obhip_profile_enable", 0)
print("n'=%d p=%d: predictor var() %.1f ms wall (host buffers in and out); kernels ms %s; %.1f TFLOP/s on p^2 n' flop"
      % (nnew, p, dt * 1e3, out, p * p * nnew / dt / 1e12))
print("var range", float(v.min()), float(v.max()))
