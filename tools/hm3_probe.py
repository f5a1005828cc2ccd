"""Probe: how much of k_hm3's time is the star-wave that holds the plain stars (the terms that
found no family)?  Times loglik_gauss$hessmult on the headline term set and on the same set
without the terms of its plain stars (their places go to padding terms: cheap stars of the empty
family).  python tools/hm3_probe.py [d] [cov] [p]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import outerbase_amd as ob
from outerbase_amd import _lib
from outerbase_amd._lib import call, ptr
from outerbase_amd.driver import bench_knots

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cov = sys.argv[2] if len(sys.argv) > 2 else "mat25"
p = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
n = 1000000
kinds = [cov] * d
om = ob.outermod()
ob.setcovfs(om, kinds)
ob.setknot(om, bench_knots(kinds, 40))
terms = om.selectterms(p)
if cov == "mat25pow":
    t4 = om.selectterms(4 * p)
    terms = t4[t4.max(1) <= 12][:p]
rng = np.random.default_rng(0)
x = 0.02 + 0.96 * rng.random((n, d))
y = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]


def share(tt):
    info = np.zeros(7, dtype=np.uint64)
    call("obhip_terms_share_tables", tt._h, ptr(info), None, None, None)
    p_pad, nplain = int(info[0]), int(info[1])
    term = np.zeros(p_pad, dtype=np.uint32)
    call("obhip_terms_share_tables", tt._h, ptr(info), ptr(term), None, None)
    return info, term[:4 * nplain]


def timeit(tr, label):
    lik = ob.loglik_gauss(om, tr, y, x)
    g = rng.standard_normal(len(tr))
    lik.hessmult(g)
    torch.cuda.synchronize()
    _lib.call("obhip_profile_reset")
    _lib.call("obhip_profile_enable", 1)
    for _ in range(20):
        lik.hessmult(g)
    torch.cuda.synchronize()
    cnt, ms = C.c_uint64(0), C.c_double(0)
    _lib.call("obhip_profile_get", b"hessmult", C.byref(cnt), C.byref(ms))
    _lib.call("obhip_profile_enable", 0)
    info, _ = share(ob.obmod._Terms(om, tr))
    print("%s: p = %d, share info %s: hessmult %.4f ms" % (label, len(tr), info.tolist(), ms.value / max(1, cnt.value)))


info, plain = share(ob.obmod._Terms(om, terms))
timeit(terms, "all terms")
keep = np.ones(len(terms), dtype=bool)
keep[plain[plain < len(terms)]] = False
timeit(terms[keep], "without the %d terms of the plain stars" % int((~keep).sum()))
