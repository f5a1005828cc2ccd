"""Times obhip_gram_dev alone on the benchmark workload (kernel tuning aid)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from outerbase_amd import _lib
from outerbase_amd.driver import HotPath
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
backend = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hp = HotPath(["mat25"] * 20, 40, 4096, n)
hp.setup()
hp.standardise()
h = C.c_void_p()
_lib.call("obhip_basis_create_dev", C.byref(h), hp.om._h, hp.x.data_ptr(), n, hp.caps.ctypes.data)
_lib.call("obhip_set_gram_backend", backend)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.call("obhip_gram_dev", h, hp.t._h, None, hp.G.data_ptr(), None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
ms = (t1 - t0) * 1e3
_lib.call("obhip_profile_reset")
_lib.call("obhip_profile_enable", 1)
_lib.call("obhip_gram_dev", h, hp.t._h, None, hp.G.data_ptr(), None)
torch.cuda.synchronize()
for name in ("materialize_B", "gram", "gram_reduce"):
    cnt, pm = C.c_uint64(), C.c_double()
    _lib.call("obhip_profile_get", name.encode(), C.byref(cnt), C.byref(pm))
    if cnt.value:
        print("  %-14s %.3f ms" % (name, pm.value / cnt.value))
_lib.call("obhip_profile_enable", 0)
print("source_hash_gram=%s" % _lib.lib.obhip_source_hash(1).decode())
print("backend %d n=%d gram %.2f ms  %.2f TFLOP/s (dbg=%s)" % (backend, n, ms, n * 4096.0 * 4097 / ms / 1e9, os.environ.get("OBHIP_GRAM_DBG")))
