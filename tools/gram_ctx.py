"""Gram kernel time in three contexts: back to back on a cached design matrix, after a fresh
basis build + materialisation, and inside whole fit+predict steps (kernel tuning aid)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from outerbase_amd import _lib
from outerbase_amd.driver import HotPath
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
hp = HotPath(["mat25"] * 20, 40, 4096, n)
hp.setup()
hp.standardise()

def prof(name):
    cnt, pm = C.c_uint64(), C.c_double()
    _lib.call("obhip_profile_get", name.encode(), C.byref(cnt), C.byref(pm))
    return pm.value / max(1, cnt.value), cnt.value

def measure(label, fn, reps=6):
    fn()
    torch.cuda.synchronize()
    _lib.call("obhip_profile_reset")
    _lib.call("obhip_profile_enable", 1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    g, c = prof("gram")
    m, _ = prof("materialize_B")
    _lib.call("obhip_profile_enable", 0)
    print("%-28s gram %.3f ms (%d calls)  materialize %.3f ms" % (label, g, c, m))

h = C.c_void_p()
_lib.call("obhip_basis_create_dev", C.byref(h), hp.om._h, hp.x.data_ptr(), n, hp.caps.ctypes.data)
measure("cached B, back to back", lambda: _lib.call("obhip_gram_dev", h, hp.t._h, None, hp.G.data_ptr(), None))
_lib.call("obhip_basis_destroy", h)

def fresh():
    hh = C.c_void_p()
    _lib.call("obhip_basis_create_dev", C.byref(hh), hp.om._h, hp.x.data_ptr(), n, hp.caps.ctypes.data)
    _lib.call("obhip_gram_dev", hh, hp.t._h, None, hp.G.data_ptr(), None)
    _lib.call("obhip_basis_destroy", hh)
measure("fresh basis + materialise", fresh)

for gap_ms in (1.0, 2.5, 5.0, 20.0):
    def fresh_gap():
        fresh()
        torch.cuda.synchronize()
        time.sleep(gap_ms * 1e-3)
    measure("fresh + %.1f ms idle" % gap_ms, fresh_gap)
measure("whole step", lambda: hp.step())
