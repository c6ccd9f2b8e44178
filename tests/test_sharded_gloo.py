"""world_size-2 (and 3) gloo runs of the node-block exchange protocol in
svinet_amd/sharded.py on CPU, with the numpy shard double standing in for the
HIP engine; the result must equal the single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from svinet_amd.host_api import Setup
from svinet_amd.sharded import ShardedSweep, block_size, node_block
from shard_double import NumpyShard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, n, k, sweeps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = Setup(path, n, k)
        shard = NumpyShard(s.n, s.k, s.ones, s.ones_prob, s.eta, s.links, s.validation_sorted,
                           s.gamma, s.lam, rank, world)
        ShardedSweep(shard, dist).sweep(sweeps)
        if rank == 0:
            np.savez(out, gamma=shard.t_gamma.numpy()[:s.n], lam=shard.lam,
                     conv=shard.conv[shard.parity].numpy()[:s.n], rows=np.array(shard.rows),
                     annealing=shard.annealing, it=shard.iter)
        # every rank must hold the same replicated state after the exchanges
        t = shard.t_gamma.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert torch.equal(t, shard.t_gamma)
    finally:
        dist.destroy_process_group()


def test_block_partition():
    assert block_size(10, 4) == 3
    assert [node_block(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert node_block(2, 4, 3) == (2, 2)


@pytest.mark.parametrize("world,sweeps", [(2, 25), (3, 70)])
def test_sharded_equals_oracle(graph_files, tmp_path, world, sweeps):
    """LFR n=1000 k=28; 70 sweeps crosses the annealing switch and the converged shortcuts."""
    path, n, k = graph_files["lfr"], 1000, 28
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), path, n, k, sweeps, out), nprocs=world, join=True)
    got = np.load(out)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    assert int(got["it"]) == ref.iter and bool(got["annealing"]) == ref.annealing
    assert np.array_equal(got["conv"], ref.converged)
    np.testing.assert_allclose(got["gamma"], ref.gamma, rtol=1e-8)
    np.testing.assert_allclose(got["lam"], ref.lam, rtol=1e-8)
    np.testing.assert_allclose(got["rows"], ref.rows[1:, 9], rtol=1e-9)
