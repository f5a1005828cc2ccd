"""Child process of test_comm_selftest_switches_a_wrong_pair_off_in_process: loads the TEST build
of the library (OBHIP_TEST_LIBRARY=testing -> outerbase_amd/libobhip_testing.so), arms the fault
injector behind the reduce-scatter / all-gather pair and checks that the exchange self-test sees
the wrong element, switches the communicator to ncclAllReduce in-process and sums exactly
afterwards."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    assert os.environ.get("OBHIP_TEST_LIBRARY") == "testing"
    import torch
    from outerbase_amd import _lib
    from outerbase_amd._lib import call, lib
    assert _lib.LIB_PATH.endswith("libobhip_testing.so")
    lib.obhip_testing_fault_inject_pair.argtypes = [C.c_int]
    lib.obhip_testing_fault_inject_pair.restype = None
    uid = np.zeros(128, dtype=np.uint8)
    call("obhip_comm_unique_id", uid.ctypes.data)
    comm = C.c_void_p()
    call("obhip_comm_init", C.byref(comm), 1, 0, uid.ctypes.data)

    def selftest(count):
        res = (C.c_int64 * 4)()
        rc = lib.obhip_comm_selftest_dev(comm, count, C.cast(res, C.c_void_p))
        return rc, [int(v) for v in res]

    def path(count):
        pth, st = C.c_int(), C.c_int()
        call("obhip_comm_exchange_path", comm, count, C.byref(pth), C.byref(st))
        return pth.value, st.value

    try:
        count = 1 << 20
        v = torch.arange(count, dtype=torch.float64, device="cuda")
        w = v.clone()
        call("obhip_comm_allreduce_dev", comm, w.data_ptr(), count)
        torch.cuda.synchronize()
        assert torch.equal(v, w)                                      # disarmed: exact
        lib.obhip_testing_fault_inject_pair(1)
        w = v.clone()
        call("obhip_comm_allreduce_dev", comm, w.data_ptr(), count)
        torch.cuda.synchronize()
        assert not torch.equal(v, w)                                  # the fault is real
        rc, res = selftest(count)
        assert rc == 0, lib.obhip_last_error()
        assert res == [2, 1, 0, 1], res                               # all-reduce in use, pair: 1 bad
        assert path(count) == (2, 2)
        w = v.clone()
        call("obhip_comm_allreduce_dev", comm, w.data_ptr(), count)
        torch.cuda.synchronize()
        assert torch.equal(v, w)
    finally:
        lib.obhip_comm_destroy(comm)
    print("fault injection ok")


if __name__ == "__main__":
    main()
