"""world_size-2 (and 3) gloo runs of the K-sharded exchange protocol of svinet_amd/ksharded.py on CPU, with the numpy
slice double standing in for the HIP engine: KShardedSweep's phase / all-reduce order in separate processes must give
the single-process oracle's state."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from svinet_amd.host_api import Setup
from svinet_amd.ksharded import KShardedSweep, column_slices
from kshard_double import NumpyKShard


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
        tl = 2.0 * np.bincount(np.asarray(s.links).ravel(), minlength=s.n)
        shard = NumpyKShard(s.n, s.k, s.ones, s.ones_prob, s.eta, s.links, tl, s.validation_sorted, s.gamma, s.lam, rank, world)
        run = KShardedSweep(shard, dist)
        run.init()
        run.sweep(sweeps)
        np.savez(out + ".%d.npz" % rank, gamma=shard.gamma, lam=shard.lam, conv=shard.conv, rows=np.array(shard.rows),
                 annealing=shard.annealing, it=shard.iter)
    finally:
        dist.destroy_process_group()


def test_column_slices():
    assert column_slices(10, 4) == [(0, 2), (2, 5), (5, 7), (7, 10)]
    assert column_slices(512, 8)[3] == (192, 256)


@pytest.mark.parametrize("world,sweeps", [(2, 25), (3, 70)])
def test_ksharded_gloo_equals_oracle(graph_files, tmp_path, world, sweeps):
    """LFR n=1000 k=28; 70 sweeps crosses the annealing switch and the converged shortcuts (quirk Q2 included)."""
    path, n, k = graph_files["lfr"], 1000, 28
    out = str(tmp_path / "ks")
    mp.spawn(_worker, args=(world, _free_port(), path, n, k, sweeps, out), nprocs=world, join=True)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    got = [np.load(out + ".%d.npz" % r) for r in range(world)]
    g = np.concatenate([x["gamma"] for x in got], 1)
    lam = np.concatenate([x["lam"] for x in got], 0)
    np.testing.assert_allclose(g, ref.gamma, rtol=1e-9)
    np.testing.assert_allclose(lam, ref.lam, rtol=1e-9)
    for x in got:
        assert np.array_equal(x["conv"], ref.converged)
        assert bool(x["annealing"]) == ref.annealing and int(x["it"]) == ref.iter
        np.testing.assert_allclose(x["rows"], ref.rows[1:, 9], rtol=1e-9)
