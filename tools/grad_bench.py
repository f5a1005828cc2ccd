"""Times the hyper-parameter gradient products at the benchmark size (tuning aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import outerbase_amd as ob
from outerbase_amd.driver import bench_knots

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kinds = ["mat25"] * 20
om = ob.outermod()
ob.setcovfs(om, kinds)
ob.setknot(om, bench_knots(kinds, 40))
terms = om.selectterms(p)
rng = np.random.default_rng(0)
x = 0.02 + 0.96 * rng.random((n, 20))
t = ob.obmod._Terms(om, terms)
b = ob.outerbase(om, x, levelcap=t.maxlevels())
a, v = rng.standard_normal(p), rng.standard_normal(n)
for name, fn in (("matmul", lambda: b.matmul(t, a)), ("tmatmul", lambda: b.tmatmul(t, v)),
                 ("matmul_gradhyp (20 hyper-parameters, incl. first-use gradient basis)", lambda: b.matmul_gradhyp(t, a)),
                 ("matmul_gradhyp", lambda: b.matmul_gradhyp(t, a)),
                 ("tmatmul_gradhyp", lambda: b.tmatmul_gradhyp(t, v)),
                 ("sqcolsums_gradhyp (incl. first-use squared store)", lambda: b.sqcolsums_gradhyp(t)),
                 ("sqcolsums_gradhyp", lambda: b.sqcolsums_gradhyp(t))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print("%-75s %8.1f ms" % (name, (time.perf_counter() - t0) * 1e3))
