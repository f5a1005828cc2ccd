"""One rank of a row-sharded obfit (outerbase_amd.fitting.obfit with comm), started as its own
process by tests/test_00_two_rank_device.py; ranks share the one GPU, libobhip's host
transport over gloo carries the sums.  usage: obfit_worker.py <out_dir> <n_total> <numb>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    out_dir, n_total, numb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import ob_oracle as O
    import outerbase_amd as ob
    from outerbase_amd.driver import make_comm, shard_rows
    rng = np.random.default_rng(17)
    x = rng.random((n_total, 8))
    y = O.borehole8d(x)
    xt = np.random.default_rng(18).random((300, 8))
    row0, n = shard_rows(rank, world, n_total)
    comm, cb = make_comm(rank, world, "host")
    m = ob.obfit(x[row0:row0 + n], y[row0:row0 + n], numb=numb, seed=5, comm=comm, row0=row0)
    pred = ob.obpred(m, xt)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), hyp=ob.gethyp(m["om"]),
             para=ob.getpara(m["logpdf"]), mean=pred["mean"], var=pred["var"],
             y_cent=m["y_cent"], y_sca=m["y_sca"], coeff=m["logpdf"].coeff,
             knots0=np.asarray(m["om"].knots()[0]), truth=O.borehole8d(xt))
    del m, pred
    if comm is not None:
        from outerbase_amd import _lib
        _lib.lib.obhip_comm_destroy(comm)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
