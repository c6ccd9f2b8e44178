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
from svinet_amd.sharded import ShardedSweep, balanced_bounds, block_size, node_block
from shard_double import NumpyShard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, n, k, sweeps, out, balanced=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = Setup(path, n, k)
        shard = NumpyShard(s.n, s.k, s.ones, s.ones_prob, s.eta, s.links, s.validation_sorted,
                           s.gamma, s.lam, rank, world, balanced=balanced)
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


def test_balanced_blocks_of_the_headline_graph(graph_files):
    """svils_balance_node_blocks (host code of the library) on ca-AstroPh: the reference numbers nodes by first
    appearance, equal-count blocks give rank 0 of 8 2.96 x the mean number of CSR entries; the balanced cut keeps every
    rank within 10 % of the mean at worlds 2, 4 and 8, and the bounds are a partition of [0, n)."""
    s = Setup(graph_files["astroph"], 17903, 20)
    deg = np.bincount(np.asarray(s.links).ravel(), minlength=s.n)
    B = block_size(s.n, 8)
    eq = np.array([deg[r * B:(r + 1) * B].sum() for r in range(8)], dtype=np.float64)
    assert 2.9 < eq.max() / eq.mean() < 3.0
    for world in (1, 2, 4, 8):
        b = balanced_bounds(s.links, s.n, world).astype(np.int64)
        assert b[0] == 0 and b[-1] == s.n and np.all(np.diff(b) > 0)
        ent = np.array([deg[b[r]:b[r + 1]].sum() for r in range(world)], dtype=np.float64)
        assert np.all(np.abs(ent / ent.mean() - 1.0) < 0.10), (world, ent / ent.mean())


@pytest.mark.parametrize("world,sweeps,balanced", [(2, 25, True), (3, 70, True), (2, 12, False)])
def test_sharded_equals_oracle(graph_files, tmp_path, world, sweeps, balanced):
    """LFR n=1000 k=28; 70 sweeps crosses the annealing switch and the converged shortcuts.  Work-balanced blocks (the
    default of every driver) and the equal blocks."""
    path, n, k = graph_files["lfr"], 1000, 28
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), path, n, k, sweeps, out, balanced), nprocs=world, join=True)
    got = np.load(out)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    assert int(got["it"]) == ref.iter and bool(got["annealing"]) == ref.annealing
    assert np.array_equal(got["conv"], ref.converged)
    np.testing.assert_allclose(got["gamma"], ref.gamma, rtol=1e-8)
    np.testing.assert_allclose(got["lam"], ref.lam, rtol=1e-8)
    np.testing.assert_allclose(got["rows"], ref.rows[1:, 9], rtol=1e-9)


class _FakeStepEngine:
    """records the phase order and hands out scripted windows: the exchange logic of ShardedStep is
    what is under test here, not the kernels (those run in tests/test_gpu_sharded.py)"""

    def __init__(self, windows):
        self.windows, self.t, self.calls = windows, 0, []

    def step_phase(self, ph):
        self.calls.append(ph)
        if ph == 3:          # PHASE_D closes the step
            self.t += 1

    def step_window(self):
        return self.windows[self.t % len(self.windows)]


class _FakeShard:
    def __init__(self, rank, world, B, cols, windows):
        self.rank, self.world, self.B, self.n_alloc = rank, world, B, B * world
        self.engine = _FakeStepEngine(windows)
        mk = lambda c, dt: torch.zeros(self.n_alloc, c, dtype=dt)
        self.rows, self.mphi = [mk(cols, torch.float64)], mk(cols, torch.float64)
        self.xflags = mk(4, torch.int32)        # packed flags (SVILS_BUF_XFLAGS)
        self.kvec_a, self.kvec_c = torch.zeros(cols, dtype=torch.float64), torch.zeros(3 * cols, dtype=torch.float64)
        self.sweeps = 0

    def end_sweep(self):
        self.sweeps += 1


def _step_worker(rank, world, port, B, cols, windows, out):
    from svinet_amd.sharded import ShardedStep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = _FakeShard(rank, world, B, cols, windows)
        drv = ShardedStep(sh, dist)
        for t, (b, e) in enumerate(windows):
            # what this rank's kernels would have produced in its own window
            lo, hi = rank * B + b, rank * B + e
            sh.rows[0][lo:hi] = 100.0 * (t + 1) + rank
            sh.mphi[lo:hi] = -(100.0 * (t + 1) + rank)
            sh.xflags[lo:hi] = 7 * (t + 1) + rank
            sh.kvec_a[:] = rank + 1.0
            sh.kvec_c[:] = 2.0 * (rank + 1)
            drv.step(1)
            assert sh.kvec_a[0].item() == world * (world + 1) / 2 and sh.kvec_c[0].item() == world * (world + 1)
        assert sh.engine.calls == [0, 1, 4, 2, 3] * len(windows)      # A, B, EXPAND, C, D
        np.savez(out + ".%d.npz" % rank, gamma=sh.rows[0].numpy(), mphi=sh.mphi.numpy(), conv=sh.xflags.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_step_exchanges_only_the_windows(tmp_path, world):
    B, cols = 10, 4
    windows = [(0, 4), (4, 8), (8, 10)]
    out = str(tmp_path / "st")
    mp.spawn(_step_worker, args=(world, _free_port(), B, cols, windows, out), nprocs=world, join=True)
    got = [np.load(out + ".%d.npz" % r) for r in range(world)]
    for g in got[1:]:                                   # replicated after the exchanges
        for key in ("gamma", "mphi", "conv"):
            assert np.array_equal(g[key], got[0][key])
    gam = got[0]["gamma"]
    for t, (b, e) in enumerate(windows):
        for r in range(world):
            assert np.all(gam[r * B + b:r * B + e] == 100.0 * (t + 1) + r)
            assert np.all(got[0]["mphi"][r * B + b:r * B + e] == -(100.0 * (t + 1) + r))
