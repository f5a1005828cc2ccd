"""One gradient evaluation of obfit's second stage at scale -- loglik_gauss + logpr_gauss in an
lpdfvec, update() with compute_gradhyp / compute_gradpara after updateom() -- timed by phase with
the library's hipEvent scopes (obhip_profile_*) and by wall clock (tuning aid).

  python tools/grad_eval_bench.py [n] [p] [d] [cov]      defaults 1000000 4096 8 mat25pow
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
reps = 5
kinds = [cov] * d
om = ob.outermod()
ob.setcovfs(om, kinds)
ob.setknot(om, bench_knots(kinds, 40))
terms = om.selectterms(p)
rng = np.random.default_rng(0)
x = 0.02 + 0.96 * rng.random((n, d))
y = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.1 * rng.standard_normal(n)
lik = ob.loglik_gauss(om, terms, y, x)
pr = ob.logpr_gauss(om, terms)
vec = ob.lpdfvec(lik, pr)
coeff = 0.01 * rng.standard_normal(p)
hyp = ob.gethyp(om)
vec.compute_gradhyp = True
vec.compute_gradpara = True
vec.update(coeff)                      # first use: views, tables, instantiations
torch.cuda.synchronize()
_lib.call("obhip_profile_reset")
_lib.call("obhip_profile_enable", 1)
t0 = time.perf_counter()
for i in range(reps):
    om.updatehyp(hyp + 0.001 * (i + 1))
    vec.updateom()
    vec.compute_gradhyp = True
    vec.compute_gradpara = True
    vec.update(coeff)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
print("n = %d, p = %d, d = %d %s: updatehyp + updateom + update with hyper-gradients %.2f ms" % (n, p, d, cov, wall))
names = ("build_basis", "build_basis_grad", "mm", "tmm", "sqtmm", "tmm_d3", "sqtmm_gradhyp_dense",
         "tmm_gradhyp_dense", "mm_gradhyp_dot", "mm_gradhyp", "sqmm_gradhyp")
for name in names:
    cnt, ms = C.c_uint64(0), C.c_double(0)
    _lib.call("obhip_profile_get", name.encode(), C.byref(cnt), C.byref(ms))
    if cnt.value:
        print("  %-22s %5.1f launches  %8.3f ms each  %8.3f ms per evaluation"
              % (name, cnt.value / reps, ms.value / cnt.value, ms.value / reps))
_lib.call("obhip_profile_enable", 0)
print("gradhyp", np.array2string(np.asarray(vec.gradhyp), precision=10))
