"""Times the one-buffer exchange of back end A (pack, RCCL reduce-scatter + all-gather with a
single rank = copies, unpack, right-hand side) at p = 4096 on one GPU: the part of a fit that
only exists with more than one rank, minus the wire."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from outerbase_amd._lib import call, lib
p = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
uid = np.zeros(128, dtype=np.uint8)
call("obhip_comm_unique_id", uid.ctypes.data)
comm = C.c_void_p()
call("obhip_comm_init", C.byref(comm), 1, 0, uid.ctypes.data)
cnt = C.c_uint64(0)
call("obhip_normal_eq_count", p, 1, C.byref(cnt))
G = torch.randn((p, p), dtype=torch.float64, device="cuda"); G = G + G.T
g = torch.randn(p, dtype=torch.float64, device="cuda"); b1 = torch.randn(p, dtype=torch.float64, device="cuda")
s = torch.tensor([10.0, 500.0], dtype=torch.float64, device="cuda")
buf = torch.zeros(cnt.value, dtype=torch.float64, device="cuda")
ms = torch.zeros(3, dtype=torch.float64, device="cuda")
def run():
    call("obhip_normal_eq_exchange_dev", comm, p, 100000, G.data_ptr(), g.data_ptr(), b1.data_ptr(),
         s.data_ptr(), buf.data_ptr(), cnt.value, ms.data_ptr())
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("p=%d: exchange with one RCCL rank %.3f ms per fit (%.1f MB buffer)" % (p, e0.elapsed_time(e1) / 20, cnt.value * 8 / 1e6))
lib.obhip_comm_destroy(comm)
