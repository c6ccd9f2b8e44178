"""worker of tests/test_gpu_sharded.py::test_two_processes_one_gpu -- one rank of a ShardedSweep run.
Launched by torch.distributed.run; every rank uses GPU 0 (the test box has one), so the process
group is gloo (RCCL refuses two ranks on one device); everything else is the production path:
HipShard, the aliasing tensors, the engine's own HIP stream, svinet_amd/sharded.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    path, n, k, sweeps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    # optional: mini-batch steps instead of sweeps -- "step:<windows per block>:<kappa>"
    mode = sys.argv[6] if len(sys.argv) > 6 else "sweep"
    import torch
    import torch.distributed as dist
    from svinet_amd.host_api import Setup
    from svinet_amd.sharded import HipShard, ShardedStep, ShardedSweep
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    setup = Setup(path, n, k)
    if mode == "kshard":   # every rank a column slice of all rows (svinet_amd/ksharded.py)
        from svinet_amd.ksharded import KShard, KShardedSweep
        ks = KShard(setup, rank, world, 0, use_validation_stop=False)
        run = KShardedSweep(ks, dist)
        run.init()
        run.sweep(sweeps)
        ks.engine.synchronize()
        torch.cuda.synchronize()
        g, lam, conv = ks.engine.state()
        c = ks.engine.control()
        np.savez(out + ".%d.npz" % rank, gamma=g, lam=lam, conv=conv, member=ks.engine.communities(), k0=ks.k0, k1=ks.k1,
                 iter=c.iter, annealing=c.annealing, rows=ks.engine.rows())
        dist.barrier()
        dist.destroy_process_group()
        return
    # whole sweeps: work-balanced blocks (the default); mini-batch steps: the equal blocks they need
    shard = HipShard(setup, rank, world, 0, equal=mode.startswith("step"), use_validation_stop=False)
    if mode.startswith("step"):
        _, nwin, kappa = mode.split(":")
        bn = (shard.B + int(nwin) - 1) // int(nwin)
        shard.engine.set_stochastic(batch_nodes=bn, tau0=1.0, kappa=float(kappa), shard_block=shard.B)
        run = ShardedStep(shard, dist)
        run.step(sweeps)
    else:
        run = ShardedSweep(shard, dist)
        run.sweep(sweeps)
    run.gather_communities()
    shard.engine.synchronize()
    torch.cuda.synchronize()
    g, lam, conv = shard.engine.state()
    c = shard.engine.control()
    np.savez(out + ".%d.npz" % rank, gamma=g, lam=lam, conv=conv, member=shard.engine.communities(),
             iter=c.iter, annealing=c.annealing, rows=shard.engine.rows(), mphi=shard.engine.aux(2))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
