"""Times the gradient-basis build (outerbase::build with dograd, src/modandbase.cpp:306-327,547-626)
at an obfit-like shape and at the headline shape: hipEvent time of the build kernels through
obhip_profile_*.  Run once with OBHIP_GRAD_KNOTLOOP=1 (round-3 kernel) and once without."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import outerbase_amd as ob
from outerbase_amd import _lib
from outerbase_amd.driver import bench_knots

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(0)
for label, kinds, m, p in (("obfit-like: d=8 mat25pow, 40 knots, 4096 terms", ["mat25pow"] * 8, 40, 4096),
                           ("obfit-like: d=8 mat25pow, 70 knots, 4096 terms", ["mat25pow"] * 8, 70, 4096),
                           ("headline: d=20 mat25, 40 knots, 4096 terms", ["mat25"] * 20, 40, 4096)):
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, bench_knots(kinds, m))
    terms = om.selectterms(p)
    t = ob.obmod._Terms(om, terms)
    x = 0.02 + 0.96 * rng.random((n, len(kinds)))
    b = ob.outerbase(om, x, levelcap=t.maxlevels())
    a = rng.standard_normal(p)
    b.matmul_gradhyp(t, a)                     # first use builds the gradient basis (warm-up)
    res = []
    for rep in range(3):
        om.updatehyp(ob.gethyp(om) + 0.01)     # a hyper-parameter update invalidates it
        b.build()
        _lib.call("obhip_profile_reset")
        _lib.call("obhip_profile_enable", 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.matmul_gradhyp(t, a)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        cnt, ms = C.c_uint64(0), C.c_double(0)
        _lib.call("obhip_profile_get", b"build_basis_grad", C.byref(cnt), C.byref(ms))
        _lib.call("obhip_profile_enable", 0)
        res.append(ms.value)
        walls = locals().get("walls", [])
        walls.append(wall)
    print("%-52s levels %s  build_basis_grad %.3f ms (min of 3: %s); whole first matmul_gradhyp after a rebuild "
          "(host tables + build + products + %d MB result copy) %.1f ms  [OBHIP_GRAD_KNOTLOOP=%s]" % (
        label, t.maxlevels().tolist(), min(res), ", ".join("%.3f" % r for r in res),
        8 * n * len(ob.gethyp(om)) // 1000000, min(walls[-3:]),
        os.environ.get("OBHIP_GRAD_KNOTLOOP", "0")))
