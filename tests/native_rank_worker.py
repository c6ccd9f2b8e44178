"""worker of tests/test_gpu_native_ranks.py -- ONE RANK of the library's own multi-rank drivers
(svils_comm_init + svils_sweep_sharded / svils_step_sharded / svils_sweep_ksharded: the collectives are issued
inside libsvils on the engine's stream).  Every rank is a separate process on GPU 0; the transport is
tests/fakerccl (SVILS_RCCL_LIBRARY, set by the test), because RCCL refuses two ranks on one device.  No
torch.distributed here: the ranks only share the 128-byte communicator id, which rank 0 leaves in a file.

argv: path n k count out rank world mode
mode: sweep | step:<windows per block>:<kappa> | kshard | kshard-log | kshard-lowt (link_thresh = 0.3) |
      kstep:<windows>:<kappa> (mini-batch steps on the K-sharded layout) |
      sweep-stop (node blocks with the validation stop rule ON: the sequence of the CLI's do_on_stop -- the host SEES the
      stop, gathers the tags, reads them at once)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def comm_id(out, rank):
    from svinet_amd import _svils
    f = out + ".id"
    if rank == 0:
        cid = _svils.comm_unique_id()
        with open(f + ".tmp", "wb") as fh:
            fh.write(cid)
        os.rename(f + ".tmp", f)
        return cid
    for _ in range(6000):
        if os.path.exists(f):
            return open(f, "rb").read()
        time.sleep(0.01)
    raise RuntimeError("no communicator id from rank 0")


def main():
    path, n, k, count, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rank, world, mode = int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
    assert os.environ.get("SVILS_RCCL_LIBRARY"), "the test names the transport"
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    from svinet_amd.sharded import balanced_bounds, block_size, node_block
    setup = Setup(path, n, k, link_thresh=0.3 if mode == "kshard-lowt" else 0.5)
    extra = {}
    if mode.startswith("kshard") or mode.startswith("kstep"):
        from svinet_amd.ksharded import column_slices
        k0, k1 = column_slices(k, world)[rank]
        eng = _svils.Engine(setup.n, setup.k, ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta,
                            link_thresh=setup.link_thresh, lt_min_deg=setup.lt_min_deg, device=0, k_slice=(k0, k1),
                            use_validation_stop=False)
        eng.set_graph(setup.links)
        eng.set_validation(setup.validation_sorted)
        eng.set_state(np.ascontiguousarray(setup.gamma[:, k0:k1]), np.ascontiguousarray(setup.lam[k0:k1]))
        if mode == "kshard-log":
            eng.ksh_log_domain(True)
        eng.comm_init(comm_id(out, rank), rank, world)
        eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
        if mode.startswith("kstep"):
            _, nwin, kappa = mode.split(":")
            eng.set_stochastic(batch_nodes=(n + int(nwin) - 1) // int(nwin), tau0=4.0, kappa=float(kappa), node_tau0=2.0, node_kappa=0.5)
        eng.ksh_init_state()
        row0 = eng.validation_row()           # collective: the constructor-time likelihood row
        if mode.startswith("kstep"):
            eng.step_ksharded(count)
        else:
            eng.sweep_ksharded(count)
        extra = dict(k0=k0, k1=k1, row0=row0)
        # the host-staged gather the CLI uses for its files: every rank's slice of the tags, rank by rank
        mine = np.ascontiguousarray(eng.communities(), dtype=np.uint8)
        wmax = max(b - a for a, b in column_slices(k, world))
        send = np.zeros((n, wmax), dtype=np.uint8)
        send[:, :k1 - k0] = mine
        extra["gathered"] = eng.allgather_host(send, world)
    else:
        B = block_size(n, world)
        if mode.startswith("step") or mode == "sweep-equal":   # mini-batch steps need the equal blocks svils_comm_init assumes
            eng = setup.engine(device=0, node_block=node_block(n, world, rank), n_alloc=B * world, use_validation_stop=False)
        else:                                                   # whole sweeps: the work-balanced blocks, declared before the communicator
            bounds = balanced_bounds(setup.links, n, world)
            eng = setup.engine(device=0, node_block=(int(bounds[rank]), int(bounds[rank + 1])), use_validation_stop=(mode == "sweep-stop"))
            eng.set_node_blocks(rank, world, bounds)
            extra_bounds = bounds
        eng.comm_init(comm_id(out, rank), rank, world)
        if not os.environ.get("NATIVE_RANK_NO_TIMING"):       # (timing brackets keep the sweeps eager: graphs need it off)
            eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
        if mode.startswith("step"):
            _, nwin, kappa = mode.split(":")
            bn = (B + int(nwin) - 1) // int(nwin)
            eng.set_stochastic(batch_nodes=bn, tau0=1.0, kappa=float(kappa), shard_block=B)
            eng.step_sharded(count)
        else:
            eng.sweep_sharded(count)
        early = {}
        if mode == "sweep-stop":
            # svinet -gpus N at its stop (host/linksampling.cc: the control block says `stopped` -> do_on_stop: gather, then the
            # files): the getters may skip their wait once the stop has been SEEN, but not for the gather enqueued since
            c = eng.control()
            assert c.stopped == 1, "the run was meant to reach its stop rule"
            eng.gather_communities()
            early = dict(member_early=eng.communities())      # no synchronize() in between: the getter itself must wait
        else:
            eng.gather_communities()
        extra = dict(mphi=eng.aux(2), **early)
    eng.synchronize()
    ci = eng.comm_query()
    assert ci["rank"] == rank and ci["nranks"] == world and ci["hip_device"] == 0
    extra.update(row_comm=ci["row_communicator"], comm_nranks=ci["nranks"])
    g, lam, conv = eng.state()
    c = eng.control()
    np.savez(out + ".%d.npz" % rank, gamma=g, lam=lam, conv=conv, member=eng.communities(), iter=c.iter,
             annealing=c.annealing, rows=eng.rows(), exchanges=eng.timing()["exchange"][1], **extra)
    eng.close()


if __name__ == "__main__":
    main()
