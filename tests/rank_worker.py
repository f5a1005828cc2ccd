"""One rank of a row-sharded DEVICE fit + predict (outerbase_amd.driver.HotPath), started
as its own process by tests/test_00_two_rank_device.py: world ranks share the one GPU of the
test box, torch.distributed (gloo) is the launcher's control plane and libobhip's host
transport carries the exchange buffer.  Writes theta / predictions / standardisation of
this rank to <out>/rank<r>.npz.

usage: rank_worker.py <out_dir> <backend: newton|cg> <n_total> <p> <knots> <kinds,comma>
                      [cg_tol cg_maxit]
       (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the environment)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, backend, n_total, p, knots, kinds = sys.argv[1:7]
    n_total, p, knots = int(n_total), int(p), int(knots)
    kinds = kinds.split(",")
    cg_tol = float(sys.argv[7]) if len(sys.argv) > 7 else 1e-10
    cg_maxit = int(sys.argv[8]) if len(sys.argv) > 8 else None
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from outerbase_amd.driver import HotPath, shard_rows
    row0, n = shard_rows(rank, world, n_total)
    hp = HotPath(kinds, knots, p, n, rank=rank, world=world, backend=backend, row0=row0,
                 n_total=n_total, transport="host" if world > 1 else None, cg_tol=cg_tol,
                 cg_maxit=cg_maxit)
    hp.setup()
    hp.step()
    hp.step()          # a second step: the exchange buffer and the basis are reused
    torch.cuda.synchronize()
    info = hp.comm_info()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), theta=hp.theta.cpu().numpy(),
             mean=hp.mean.cpu().numpy(), cent=hp.y_cent, sd=hp.y_sca, row0=row0, n=n,
             G00=hp.G[:8, :8].cpu().numpy(), ranks=info["ranks"],
             iters=-1 if hp.cg_iters is None else hp.cg_iters)
    hp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
