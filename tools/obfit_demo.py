"""obfit / obpred on the Borehole function, printed (demo and timing aid)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def borehole8d(x):
    """Borehole function on the unit cube (R/testfuncs.R:32-46)."""
    rw = x[:, 0] * (0.15 - 0.05) + 0.05
    r = x[:, 1] * (50000 - 100) + 100
    Tu = x[:, 2] * (115600 - 63070) + 63070
    Hu = x[:, 3] * (1110 - 990) + 990
    Tl = x[:, 4] * (116 - 63.1) + 63.1
    Hl = x[:, 5] * (820 - 700) + 700
    L = x[:, 6] * (1680 - 1120) + 1120
    Kw = x[:, 7] * (12045 - 9855) + 9855
    m1 = 2 * np.pi * Tu * (Hu - Hl)
    m2 = np.log(r / rw)
    m3 = 1 + 2 * L * Tu / (m2 * rw ** 2 * Kw) + Tu / Tl
    return m1 / m2 / m3 - 77


import outerbase_amd as ob
from outerbase_amd import fitting
_calls = {"n": 0, "t": 0.0}
_orig = fitting._lpdfwrapper
def _counted(*a, **k):
    t0 = time.time()
    r = _orig(*a, **k)
    _calls["n"] += 1
    _calls["t"] += time.time() - t0
    return r
fitting._lpdfwrapper = _counted
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
numb = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(42)
x = rng.random((n, 8)); y = borehole8d(x)
if os.environ.get("OBFIT_CPROFILE"):  # where the host time of the fit goes, by C-ABI entry point
    from outerbase_amd import _lib
    _t = {}
    _orig_call = _lib.call
    def _timed(name, *a):
        t0 = time.perf_counter()
        r = _orig_call(name, *a)
        e = _t.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t0
        return r
    import outerbase_amd.obmod as _om, outerbase_amd.fitting as _ft
    for _m in (_lib, _om, _ft):
        if hasattr(_m, "call"): setattr(_m, "call", _timed)
t0=time.time(); m = ob.obfit(x, y, numb=numb, seed=1, verbose=1); t1=time.time()
if os.environ.get("OBFIT_CPROFILE"):
    for k, v in sorted(_t.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%-40s %6d calls %9.1f ms total %8.2f ms each" % (k, v[0], 1e3 * v[1], 1e3 * v[1] / v[0]))
xt = rng.random((200, 8)); pred = ob.obpred(m, xt); yt = borehole8d(xt)
print("function evaluations (updatehyp + updateom + updatepara + optcg + gradients): %d, %.1f ms each, %.2f s of the fit" % (_calls["n"], 1e3 * _calls["t"] / max(1, _calls["n"]), _calls["t"]))
print("fit s", t1-t0, "rmse/sd", math.sqrt(np.mean((pred["mean"]-yt)**2))/np.std(yt), "hyp", ob.gethyp(m["om"]), "para", ob.getpara(m["logpdf"]))
z=(pred["mean"]-yt)/np.sqrt(pred["var"]); print("rms z", np.sqrt(np.mean(z**2)))
