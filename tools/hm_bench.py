"""Times the likelihood's Hessian product (lpdf$hessmult: B^T (e^{-2 sigma} B g), the PCG's inner
pass) on terms of a given shape (tuning aid; OBHIP_HM2_VARIANT / OBHIP_HM_V1 select the kernel).

  python tools/hm_bench.py [n] [p] [d] [cov] [maxlev]      defaults 1000000 4096 8 mat25pow 12
"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import outerbase_amd as ob
from outerbase_amd import _lib
from outerbase_amd.driver import bench_knots

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cov = sys.argv[4] if len(sys.argv) > 4 else "mat25pow"
kinds = [cov] * d
om = ob.outermod()
ob.setcovfs(om, kinds)
ob.setknot(om, bench_knots(kinds, 40))
# (obfit's term sets at this shape keep fewer levels than selectterms at the default
# hyper-parameters: its length scales grow and the eigenvalues fall faster)
maxlev = int(sys.argv[5]) if len(sys.argv) > 5 else 12
terms = om.selectterms(4 * p)
terms = terms[terms.max(1) <= maxlev][:p]      # (<= 147 used columns: what k_hm2's two tile buffers hold)
assert len(terms) == p
rng = np.random.default_rng(0)
x = 0.02 + 0.96 * rng.random((n, d))
y = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]
lik = ob.loglik_gauss(om, terms, y, x)
g = rng.standard_normal(p)
ref = lik.hessmult(g)
torch.cuda.synchronize()
_lib.call("obhip_profile_reset")
_lib.call("obhip_profile_enable", 1)
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    out = lik.hessmult(g)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
cnt, ms = C.c_uint64(0), C.c_double(0)
_lib.call("obhip_profile_get", b"hessmult", C.byref(cnt), C.byref(ms))
_lib.call("obhip_profile_enable", 0)
A = (terms > 0).sum(1)
print("n = %d, p = %d, d = %d %s, factors per term %.2f (max %d): hessmult %.3f ms on the device, %.3f ms wall; checksum %.12e"
      % (n, p, d, cov, A.mean(), A.max(), ms.value / max(1, cnt.value), wall, float(np.dot(out, g))))
