"""libobhip's communicator on one GPU: the RCCL transport with a single rank (every RCCL
entry point the N > 1 path uses is called -- ncclGetUniqueId, ncclCommInitRank,
ncclCommCount, ncclReduceScatter + ncclAllGather, ncclAllReduce -- and a one-rank sum is
the identity), and the pack / unpack kernels of the one-buffer exchange at the headline
p = 4096 against a NumPy restatement of the layout."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _comm_rccl_single():
    from outerbase_amd._lib import call
    uid = np.zeros(128, dtype=np.uint8)
    call("obhip_comm_unique_id", uid.ctypes.data)
    assert uid.any()
    h = C.c_void_p()
    call("obhip_comm_init", C.byref(h), 1, 0, uid.ctypes.data)
    return h


@pytest.mark.parametrize("p", [1, 63, 300, 4096])
def test_exchange_roundtrip_through_rccl_single_rank(p):
    import torch
    from outerbase_amd._lib import call, lib
    rng = np.random.default_rng(p)
    A = rng.standard_normal((p, p))
    G = A + A.T
    g, b1 = rng.standard_normal(p), rng.standard_normal(p)
    n = 5000.0
    y = 3.0 + 2.0 * rng.standard_normal(int(n))
    sums = np.array([y.sum(), (y * y).sum()])
    comm = _comm_rccl_single()
    try:
        nr, rk, tr, rr, rv = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        call("obhip_comm_info", comm, C.byref(nr), C.byref(rk), C.byref(tr), C.byref(rr), C.byref(rv))
        assert (nr.value, rk.value, tr.value, rr.value) == (1, 0, 1, 1) and rv.value > 0
        cnt = C.c_uint64(0)
        call("obhip_normal_eq_count", p, 1, C.byref(cnt))
        dG = torch.from_numpy(G).cuda()
        dg, db1 = torch.from_numpy(g).cuda(), torch.from_numpy(b1).cuda()
        ds = torch.from_numpy(sums).cuda()
        buf = torch.full((cnt.value,), float("nan"), dtype=torch.float64, device="cuda")
        ms = torch.zeros(3, dtype=torch.float64, device="cuda")
        # scribble over the lower triangle: the unpack must restore it from the upper one
        dG2 = dG.clone()
        call("obhip_normal_eq_exchange_dev", comm, p, int(n), dG2.data_ptr(), dg.data_ptr(),
             db1.data_ptr(), ds.data_ptr(), buf.data_ptr(), cnt.value, ms.data_ptr())
        torch.cuda.synchronize()
        hb = buf.cpu().numpy()
        tri = p * (p + 1) // 2
        assert np.array_equal(hb[:tri], G[np.triu_indices(p)])
        assert np.array_equal(hb[tri:tri + p], g) and np.array_equal(hb[tri + p:tri + 2 * p], b1)
        assert np.array_equal(hb[tri + 2 * p:tri + 2 * p + 3], [sums[0], sums[1], n])
        assert not np.isnan(hb).any() and not hb[tri + 2 * p + 3:].any()
        assert np.array_equal(dG2.cpu().numpy(), G)
        cent, sd = y.mean(), y.std(ddof=1)
        got = ms.cpu().numpy()
        assert abs(got[0] - cent) < 1e-13 * abs(cent) and abs(got[1] - sd) < 1e-12 * sd and got[2] == n
        want = (g - got[0] * b1) / got[1]
        assert np.max(np.abs(dg.cpu().numpy() - want)) < 1e-14 * np.max(np.abs(want))
        # plain in-place sums of other sizes (reduce-scatter + all-gather above 4096 doubles
        # per rank, all-reduce below and for ragged counts)
        for count in (3, 4097, 8192, 1 << 20):
            v = torch.from_numpy(rng.standard_normal(count)).cuda()
            w = v.clone()
            call("obhip_comm_allreduce_dev", comm, w.data_ptr(), count)
            torch.cuda.synchronize()
            assert torch.equal(v, w)
    finally:
        lib.obhip_comm_destroy(comm)


def test_exchange_without_communicator_only_standardises():
    import torch
    from outerbase_amd._lib import call
    p = 257
    rng = np.random.default_rng(1)
    G = rng.standard_normal((p, p))
    g, b1 = rng.standard_normal(p), rng.standard_normal(p)
    y = 10.0 + rng.standard_normal(999)
    cnt = C.c_uint64(0)
    call("obhip_normal_eq_count", p, 1, C.byref(cnt))
    tri = p * (p + 1) // 2
    dG = torch.from_numpy(G).cuda()
    dg, db1 = torch.from_numpy(g).cuda(), torch.from_numpy(b1).cuda()
    ds = torch.tensor([y.sum(), (y * y).sum()], dtype=torch.float64, device="cuda")
    tail = torch.zeros(cnt.value - tri, dtype=torch.float64, device="cuda")
    ms = torch.zeros(3, dtype=torch.float64, device="cuda")
    call("obhip_normal_eq_exchange_dev", None, p, len(y), dG.data_ptr(), dg.data_ptr(),
         db1.data_ptr(), ds.data_ptr(), tail.data_ptr() - 8 * tri, cnt.value, ms.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dG.cpu().numpy(), G)          # untouched, not even symmetrised
    cent, sd = y.mean(), y.std(ddof=1)
    want = (g - cent * b1) / sd
    assert np.max(np.abs(dg.cpu().numpy() - want)) < 1e-11 * np.max(np.abs(want))
    assert abs(float(ms[1]) - sd) < 1e-10 * sd
