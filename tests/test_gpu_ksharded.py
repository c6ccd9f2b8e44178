"""-m gpu: K-sharded sweeps (every rank a column slice of all rows) against the oracle, with all ranks as
virtual ranks in one process on one GPU: the device kernels of svinet_amd/csrc/svils_ksh.h and the phase
order of the C ABI; the protocol itself is pinned on the CPU by tests/test_ksharded_protocol.py."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph,world,k,sweeps", [("lfr", 2, 28, 70), ("lfr", 4, 100, 6), ("lfr", 3, 130, 5),
                                                   ("astroph", 4, 200, 3), ("lfr", 7, 28, 20),   # slices of <= 64 columns (down to 4): k_phi_ksh16
                                                   ("astroph", 2, 200, 3), ("lfr", 2, 300, 30),   # 100 / 150 columns: k_phi_ksh<2>, <4>
                                                   ("lfr", 2, 600, 4)])         # 300 columns: k_phi_ksh<8>
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
        np.testing.assert_allclose(s.engine.rows()[:, 1:], np.asarray(ref.rows)[1:sweeps + 1, 1:], rtol=1e-8, atol=1e-11)
        assert np.array_equal(s.engine.aux(3), ref.active_comms)


def test_ksharded_astroph_k200_four_shards_whole_trajectory(graph_files):
    """BASELINE config 4 (ca-AstroPh, K = 200) on FOUR virtual K-shards of 50 columns over the reference's whole natural
    run: annealing switch (sweep 24), shortcut regime, and the replicated stop rule firing on every shard at the
    oracle's sweep (27)."""
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, sweep_virtual
    path, n, k, world = graph_files["astroph"], 17903, 200, 4
    setup = Setup(path, n, k)
    shards = [KShard(setup, r, world, 0, use_validation_stop=True) for r in range(world)]
    init_virtual(shards)
    ref = O.LinkSampling(O.Network(path, n), k)
    n_ref = 0
    while True:
        rc = ref.sweep()
        n_ref += 1
        if rc == 2:
            break
        assert n_ref < 200
    sweep_virtual(shards, n_ref + 3)                   # the sweeps after the stop are no-ops on every shard
    states = [s.engine.state() for s in shards]
    g = np.concatenate([st[0] for st in states], 1)
    lam = np.concatenate([st[1] for st in states], 0)
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    want = ref.communities()
    for s, st in zip(shards, states):
        c = s.engine.control()
        assert c.stopped == 1 and c.sweeps_done == n_ref and c.iter == ref.iter and bool(c.annealing) == ref.annealing
        assert np.array_equal(st[2], ref.converged)
        np.testing.assert_allclose(s.engine.rows()[:, 1:], np.asarray(ref.rows)[1:, 1:], rtol=1e-8, atol=1e-11)
        assert np.array_equal(s.engine.communities(), want[:, s.k0:s.k1])


@pytest.mark.parametrize("world,k,sweeps", [(2, 28, 40), (3, 100, 5)])
def test_ksharded_processes_one_gpu(graph_files, tmp_path, world, k, sweeps):
    """svinet_amd/ksharded.py end to end in separate processes (one per rank, as on a multi-GPU node), all on
    GPU 0 with a gloo group: the column slices put together equal the oracle's state, the replicated flags,
    likelihood rows and community tags agree on every rank."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py")
    out = str(tmp_path / "kstate")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29650 + world), worker,
                        graph_files["lfr"], "1000", str(k), str(sweeps), out, "kshard"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    states = [np.load(out + ".%d.npz" % rk) for rk in range(world)]
    g = np.concatenate([s["gamma"] for s in states], 1)
    lam = np.concatenate([s["lam"] for s in states], 0)
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    want = ref.communities()
    for s in states:
        assert np.array_equal(s["conv"], ref.converged)
        assert int(s["iter"]) == ref.iter and bool(s["annealing"]) == ref.annealing
        np.testing.assert_allclose(s["rows"][:, 1:], ref.rows[1:, 1:], rtol=1e-9, atol=1e-13)
        assert np.array_equal(s["member"], want[:, int(s["k0"]):int(s["k1"])])   # each rank tags its own columns


@pytest.mark.parametrize("log_domain", [False, True])
def test_native_ksharded_driver_world1(graph_files, log_domain):
    """svils_comm_init + svils_ksh_init_state + svils_sweep_ksharded with a communicator of ONE rank holding
    every column (RCCL never saw more than one rank of this code on the hardware available): equals the
    plain engine's sweeps at 1e-12 (the two phi forms differ in rounding) and the oracle's flags."""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    path, n, k = graph_files["lfr"], 1000, 100
    setup = Setup(path, n, k)
    eng = _svils.Engine(n, k, ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta, link_thresh=setup.link_thresh,
                        lt_min_deg=setup.lt_min_deg, use_validation_stop=False, k_slice=(0, k))
    eng.set_graph(setup.links)
    eng.set_validation(setup.validation_sorted)
    eng.set_state(setup.gamma, setup.lam)
    eng.ksh_log_domain(log_domain)
    eng.comm_init(_svils.comm_unique_id(), 0, 1)
    eng.ksh_init_state()
    plain = setup.engine(use_validation_stop=False)
    # the constructor's validation_likelihood row (src/linksampling.cc:149-150) of a K-sharded state: collective
    np.testing.assert_allclose(eng.validation_row(), plain.validation_row(), rtol=1e-12, atol=0)
    eng.sweep_ksharded(8)
    plain.sweep(8)
    np.testing.assert_allclose(eng.validation_row(), plain.validation_row(), rtol=1e-11, atol=0)   # between two sweeps
    # svils_comm_allgather_host through the one-rank communicator (device staging + ncclAllGather)
    blob = np.arange(12345, dtype=np.float64)
    assert np.array_equal(eng.allgather_host(blob, 1)[0], blob)
    g1, l1, c1 = eng.state()
    g2, l2, c2 = plain.state()
    assert np.max(np.abs(g1 - g2) / g2) < 1e-12 and np.max(np.abs(l1 - l2) / np.abs(l2)) < 1e-12
    assert np.array_equal(c1, c2)
    np.testing.assert_allclose(eng.rows()[:, 1:], plain.rows()[:, 1:], rtol=1e-11, atol=1e-14)
    assert np.array_equal(eng.communities(), plain.communities())


@pytest.mark.parametrize("world,k,sweeps,log_domain", [(2, 28, 60, False), (3, 100, 40, False), (2, 300, 30, False),
                                                        (2, 28, 60, True), (2, 300, 30, True)])
def test_ksharded_active_set_path(graph_files, world, k, sweeps, log_domain):
    """the active-set branch (src/linksampling.cc:634-681) from the first sweep on (sparse_after_iter = 0, as in the
    authors' shipped runs): the union of two active sets spans the slices, an empty union is empty on every rank"""
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, sweep_virtual
    path, n = graph_files["lfr"], 1000
    setup = Setup(path, n, k)
    shards = [KShard(setup, r, world, 0, use_validation_stop=False, sparse_after_iter=0, log_domain=log_domain) for r in range(world)]
    init_virtual(shards)
    sweep_virtual(shards, sweeps)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False, sparse_after_iter=0)
    counts = []
    for _ in range(sweeps):
        ref.sweep()
        counts.append(ref.link_counts())
    assert sum(c[1] for c in counts) > 0, "the test must exercise the active-set branch"
    states = [s.engine.state() for s in shards]
    g = np.concatenate([st[0] for st in states], 1)
    lam = np.concatenate([st[1] for st in states], 0)
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    for st, s in zip(states, shards):
        assert np.array_equal(st[2], ref.converged)
        got = [tuple(int(x) for x in r[:3]) for r in s.engine.sweep_stats(0, sweeps)]
        assert got == counts


def test_ksharded_underflowing_denominator_is_loud():
    """rows of (almost) disjoint support: every e^x_k of a link underflows on every rank.  The K-sharded layout has no
    log-domain detour across ranks; it must stop with an error, not drop the link silently."""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, sweep_virtual
    k, world = 200, 2
    n = 2 * k + 20
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    chords = np.stack([np.arange(n), (np.arange(n) + 7) % n], 1)
    pairs = np.concatenate([ring, chords]).astype(np.int32)
    setup = Setup(n=n, k=k, pairs=pairs, heldout_ratio=0.0)
    g = np.full((n, k), 5e-4)
    g[np.arange(n), np.arange(n) % k] = 50.0
    lam = np.tile([3.0, 2.0], (k, 1))
    shards = [KShard(setup, r, world, 0, use_validation_stop=False, log_domain=False) for r in range(world)]
    for s in shards:
        s.engine.set_state(np.ascontiguousarray(g[:, s.k0:s.k1]), np.ascontiguousarray(lam[s.k0:s.k1]))
    init_virtual(shards)
    with pytest.raises(_svils.SvilsError, match="underflowed"):
        sweep_virtual(shards, 1)       # the first host entry that looks at the control block reports it
        shards[0].engine.control()


@pytest.mark.parametrize("k,world", [(200, 2), (64, 2), (800, 4)])
def test_ksharded_log_domain_handles_underflowing_rows(k, world):
    """the same rows of disjoint support in the log-domain mode (per-link max exchanged first): two sweeps equal the oracle's
    sequential log-sum-exp.  K = 800 takes the mode by default (k_total > 700)."""
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, sweep_virtual
    n = 2 * k + 20
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    chords = np.stack([np.arange(n), (np.arange(n) + 7) % n], 1)
    pairs = np.concatenate([ring, chords]).astype(np.int32)
    setup = Setup(n=n, k=k, pairs=pairs, heldout_ratio=0.0)
    ref = O.LinkSampling(O.Network(n=n, pairs=pairs), k, heldout_ratio=0.0, use_validation_stop=False)
    ref.set_skip_validation(True)
    g = np.full((n, k), 5e-4)
    g[np.arange(n), np.arange(n) % k] = 50.0
    lam = np.tile([3.0, 2.0], (k, 1))
    ref.set_gamma(g); ref.set_lambda(lam); ref.refresh()
    shards = [KShard(setup, r, world, 0, use_validation_stop=False, log_domain=None if k > 700 else True) for r in range(world)]
    assert all(s.log_domain for s in shards)
    for s in shards:
        s.engine.set_state(np.ascontiguousarray(g[:, s.k0:s.k1]), np.ascontiguousarray(lam[s.k0:s.k1]))
    init_virtual(shards)
    for _ in range(2):
        ref.sweep()
        sweep_virtual(shards, 1)
        states = [s.engine.state() for s in shards]
        gg = np.concatenate([st[0] for st in states], 1)
        ll = np.concatenate([st[1] for st in states], 0)
        assert np.isfinite(gg).all() and np.isfinite(ll).all()
        np.testing.assert_allclose(gg, ref.gamma, rtol=1e-7)
        np.testing.assert_allclose(ll, ref.lam, rtol=1e-7)
        for st in states:
            assert np.array_equal(st[2], ref.converged)


def test_ksharded_log_domain_equals_product_form(graph_files):
    """on an ordinary model the two forms agree (LFR K=100, 3 ranks, 40 sweeps) and both equal the oracle"""
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, sweep_virtual
    path, n, k, world, sweeps = graph_files["lfr"], 1000, 100, 3, 40
    setup = Setup(path, n, k)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    for mode in (True, False):
        shards = [KShard(setup, r, world, 0, use_validation_stop=False, log_domain=mode) for r in range(world)]
        init_virtual(shards)
        sweep_virtual(shards, sweeps)
        states = [s.engine.state() for s in shards]
        g = np.concatenate([st[0] for st in states], 1)
        assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
        assert all(np.array_equal(st[2], ref.converged) for st in states)
        assert (lambda c: (c.links_dense, c.links_sparse, c.links_shortcut))(shards[0].engine.control()) == ref.link_counts()


@pytest.mark.parametrize("world,k,sweeps,thresh,min_deg", [(2, 28, 40, 0.3, 0), (3, 100, 12, 0.2, 2), (4, 28, 25, 0.05, 0),
                                                           (2, 300, 8, 0.3, 1)])
def test_ksharded_argmax_tagging_below_one_half(graph_files, world, k, sweeps, thresh, min_deg):
    """link_thresh < 1/2: a phi above the threshold need not be the link's maximum, so the tag goes to the first strict
    maximum over ALL columns (src/linksampling.cc:704-717, src/matrix.hh:521-532) -- the per-link maximum travels in the
    log-domain exchange, the lowest column attaining it next to the denominators (SVILS_KSH_EARG, MIN).  Tags, flags,
    counters and state equal the oracle's; thresh = 0.05 tags nearly every link, so second-largest memberships above
    the threshold (which the >= 1/2 rule would also tag) must NOT be tagged."""
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, sweep_virtual
    path, n = graph_files["lfr"], 1000
    setup = Setup(path, n, k, link_thresh=thresh, lt_min_deg=min_deg)
    shards = [KShard(setup, r, world, 0, use_validation_stop=False) for r in range(world)]
    assert all(s.log_domain for s in shards)       # forced: the maximum is what the log-domain exchange carries
    init_virtual(shards)
    sweep_virtual(shards, sweeps)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False, link_thresh=thresh, lt_min_deg=min_deg)
    for _ in range(sweeps):
        ref.sweep()
    states = [s.engine.state() for s in shards]
    g = np.concatenate([st[0] for st in states], 1)
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    want = ref.communities()
    assert want.sum() > 0
    got = np.concatenate([s.engine.communities() for s in shards], 1)
    assert np.array_equal(got, want)
    for st in states:
        assert np.array_equal(st[2], ref.converged)


@pytest.mark.parametrize("world,k,sweeps,log_domain", [(2, 28, 30, None), (3, 100, 8, None), (2, 300, 5, None), (3, 28, 30, True)])
def test_ksharded_minibatch_full_window_is_a_sweep(graph_files, world, k, sweeps, log_domain):
    """mini-batch steps on K-sharded handles with ONE window (all nodes) and step size 1 (kappa = 0) are full sweeps:
    the entry-indexed denominators (every CSR entry its own value), the window ranges of the exchanges, the blended
    finalise and lambda at rho = 1 -- against the oracle"""
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, step_virtual
    path, n = graph_files["lfr"], 1000
    setup = Setup(path, n, k)
    shards = [KShard(setup, r, world, 0, use_validation_stop=False, log_domain=log_domain) for r in range(world)]
    for s in shards:
        s.engine.set_stochastic(batch_nodes=0, tau0=1.0, kappa=0.0)
    init_virtual(shards)
    step_virtual(shards, sweeps)
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    states = [s.engine.state() for s in shards]
    g = np.concatenate([st[0] for st in states], 1)
    lam = np.concatenate([st[1] for st in states], 0)
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    for st in states:
        assert np.array_equal(st[2], ref.converged)
    for s in shards:
        c = s.engine.control()
        assert c.iter == ref.iter and bool(c.annealing) == ref.annealing
        np.testing.assert_allclose(s.engine.rows()[:, 1:], np.asarray(ref.rows)[1:sweeps + 1, 1:], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("world,k,nwin,steps,thresh", [(2, 28, 3, 60, 0.5), (3, 100, 4, 24, 0.5), (2, 64, 5, 25, 0.3)])
def test_ksharded_minibatch_windows_equal_the_plain_engine(graph_files, world, k, nwin, steps, thresh):
    """windows of n / nwin nodes with damped steps: the column slices of the K-sharded ranks, put together, equal
    svils_step on ONE plain handle with the same windows and step sizes (state, flags, likelihood rows, tags) -- the
    same algorithm, sharded by columns; link_thresh = 0.3 runs the arg-max tagging in steps"""
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, init_virtual, step_virtual
    path, n = graph_files["lfr"], 1000
    setup = Setup(path, n, k, link_thresh=thresh)
    bn = (n + nwin - 1) // nwin
    kw = dict(batch_nodes=bn, tau0=4.0, kappa=0.6, node_tau0=2.0, node_kappa=0.5)
    shards = [KShard(setup, r, world, 0, use_validation_stop=False) for r in range(world)]
    for s in shards:
        s.engine.set_stochastic(**kw)
    init_virtual(shards)
    step_virtual(shards, steps)
    plain = setup.engine(use_validation_stop=False)
    plain.set_stochastic(**kw)
    plain.step(steps)
    pg, pl, pc = plain.state()
    states = [s.engine.state() for s in shards]
    g = np.concatenate([st[0] for st in states], 1)
    lam = np.concatenate([st[1] for st in states], 0)
    assert np.max(np.abs(g - pg) / np.abs(pg)) < 1e-9
    assert np.max(np.abs(lam - pl) / np.abs(pl)) < 1e-9
    for st in states:
        assert np.array_equal(st[2], pc)
    np.testing.assert_allclose(shards[0].engine.rows()[:, 1:], plain.rows()[:, 1:], rtol=1e-8, atol=1e-11)
    assert np.array_equal(np.concatenate([s.engine.communities() for s in shards], 1), plain.communities())
    assert shards[0].engine.control().iter == steps


def test_ksharded_rejects_what_it_does_not_do(graph_files):
    """plain sweeps / steps and node-block mini-batches are refused on a K-sharded handle, not approximated; a handle
    with link_thresh < 1/2 cannot leave the log-domain exchange"""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    setup = Setup(graph_files["lfr"], 1000, 100)
    kw = dict(ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta, lt_min_deg=0, use_validation_stop=False, k_slice=(0, 50))
    low = _svils.Engine(1000, 100, link_thresh=0.3, **kw)
    assert low.ksh_log_domain() == 1
    with pytest.raises(_svils.SvilsError, match="link_thresh"):
        low.ksh_log_domain(False)
    eng = _svils.Engine(1000, 100, link_thresh=0.5, **kw)
    with pytest.raises(_svils.SvilsError, match="K-sharded"):
        eng.set_stochastic(batch_nodes=100, shard_block=500)
    eng.set_graph(setup.links)
    eng.set_validation(setup.validation_sorted)
    eng.set_state(np.ascontiguousarray(setup.gamma[:, :50]), np.ascontiguousarray(setup.lam[:50]))
    with pytest.raises(_svils.SvilsError, match="K-sharded"):
        eng.sweep(1)
