"""-m gpu: K-sharded sweeps (every rank a column slice of all rows) against the oracle, with all ranks as
virtual ranks in one process on one GPU: the device kernels of svinet_amd/csrc/svils_ksh.h and the phase
order of the C ABI; the protocol itself is pinned on the CPU by tests/test_ksharded_protocol.py."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph,world,k,sweeps", [("lfr", 2, 28, 70), ("lfr", 4, 100, 6), ("lfr", 3, 130, 5),
                                                   ("astroph", 4, 200, 3)])
def test_ksharded_virtual_ranks_equal_oracle(graph_files, graph, world, k, sweeps):
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, sweep_virtual
    path, n = graph_files[graph], {"lfr": 1000, "astroph": 17903}[graph]
    setup = Setup(path, n, k)
    shards = [KShard(setup, r, world, 0, use_validation_stop=False) for r in range(world)]
    init_virtual(shards)
    sweep_virtual(shards, sweeps)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    states = [s.engine.state() for s in shards]
    g = np.concatenate([st[0] for st in states], 1)
    lam = np.concatenate([st[1] for st in states], 0)
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    for st in states:                                  # flags are replicated, identical on every rank
        assert np.array_equal(st[2], ref.converged)
    for s in shards:
        c = s.engine.control()
        assert c.iter == ref.iter and bool(c.annealing) == ref.annealing and c.sweeps_done == sweeps
        assert (c.links_dense, c.links_sparse, c.links_shortcut) == ref.link_counts()
        np.testing.assert_allclose(s.engine.rows()[:, 1:], np.asarray(ref.rows)[1:sweeps + 1, 1:], rtol=1e-9, atol=1e-13)
        assert np.array_equal(s.engine.aux(3), ref.active_comms)
