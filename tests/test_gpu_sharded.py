"""-m gpu: the HIP engine in node-block mode.  Two (three) virtual ranks share the
one GPU of the test box; the exchanges of svinet_amd/sharded.py are done
in-process on the torch tensors that alias the engines' device buffers.  The
result must equal the oracle (and hence the single-engine run)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _exchange_sum(ts):
    tot = ts[0].clone()
    for t in ts[1:]:
        tot += t
    for t in ts:
        t.copy_(tot)


@pytest.mark.parametrize("graph,world,k,sweeps,balanced", [("lfr", 2, 28, 70, True), ("lfr", 3, 28, 12, True), ("lfr", 2, 100, 5, True),
                                                            ("astroph", 4, 200, 3, True),   # BASELINE config 4's shape, sharded
                                                            ("astroph", 8, 20, 8, True),    # the headline graph on eight work-balanced blocks
                                                            ("lfr", 3, 28, 12, False)])     # equal blocks (s3 by node block)
def test_virtual_ranks_equal_oracle(graph_files, graph, world, k, sweeps, balanced):
    import torch
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    from svinet_amd.sharded import HipShard

    path, n = graph_files[graph], {"lfr": 1000, "astroph": 17903}[graph]
    setup = Setup(path, n, k)
    shards = [HipShard(setup, r, world, 0, equal=not balanced, use_validation_stop=False) for r in range(world)]
    bounds, bm = shards[0].bounds.astype(np.int64), shards[0].bmax
    if balanced and graph == "astroph":
        # what the blocks are for: CSR entries per rank within 10 % of the mean (equal blocks: rank 0 of 8 has 2.96 x)
        deg = np.bincount(np.asarray(setup.links).ravel(), minlength=n)
        ent = np.array([deg[bounds[r]:bounds[r + 1]].sum() for r in range(world)], dtype=np.float64)
        assert np.all(np.abs(ent / ent.mean() - 1.0) < 0.10), ent / ent.mean()

    def sync():
        for s in shards:
            s.engine.synchronize()
        torch.cuda.synchronize()

    for _ in range(sweeps):
        # two exchange points, the same while annealing and after: sum[k] travels with the staged rows
        for s in shards:
            s.phase(_svils.PHASE_A)
            s.phase(_svils.PHASE_B_LIGHT)
        sync()
        _exchange_sum([s.kvec_a for s in shards])
        for dst in range(world):
            for src in range(world):
                if src != dst:
                    shards[dst].gstage[src * bm:(src + 1) * bm].copy_(shards[src].gstage[src * bm:(src + 1) * bm])
        sync()
        for s in shards:
            s.phase(_svils.PHASE_EXPAND_ALL)
            s.phase(_svils.PHASE_C)
        sync()
        _exchange_sum([s.kvec_c for s in shards])
        sync()
        for s in shards:
            s.phase(_svils.PHASE_D)
            s.end_sweep()
    sync()

    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    if sweeps >= 60:
        assert not ref.annealing   # the run crossed the switch
    states = [s.engine.state() for s in shards]
    for g, lam, conv in states:
        assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
        assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
        assert np.array_equal(conv, ref.converged)
    # replicated state is bit-identical across ranks
    for g, lam, conv in states[1:]:
        assert np.array_equal(g, states[0][0]) and np.array_equal(lam, states[0][1])
    c = shards[0].engine.control()
    assert c.iter == ref.iter and bool(c.annealing) == ref.annealing
    # the derived per-node state every rank recomputed for every row: Elogpi, active counts
    for s in shards:
        assert np.array_equal(s.engine.aux(3), ref.active_comms)
        np.testing.assert_allclose(s.engine.aux(0), ref.elogpi, rtol=0, atol=1e-9)
    # communities: each rank tagged its own rows
    want = ref.communities()
    for r, s in enumerate(shards):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        assert np.array_equal(s.engine.communities()[lo:hi], want[lo:hi])


def test_native_rccl_driver_world1(graph_files):
    """the library's own multi-GPU driver (svils_comm_init / svils_sweep_sharded: RCCL all-reduce and
    all-gather on the engine's stream) on a communicator of ONE rank -- the transport this box can
    run -- equals the caller-driven protocol bit for bit and the plain engine up to summation order."""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    from svinet_amd.sharded import HipShard, ShardedSweep
    for key, n, k, sweeps in (("lfr", 1000, 28, 40), ("lfr", 1000, 64, 6)):
        setup = Setup(graph_files[key], n, k)
        eng = setup.engine(use_validation_stop=False, node_block=(0, n), n_alloc=n)
        eng.comm_init(_svils.comm_unique_id(), 0, 1)
        eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
        eng.sweep_sharded(sweeps)
        eng.gather_communities()
        eng.synchronize()
        # the collectives really ran: two exchange points per sweep, annealing or not (LFR K=28 leaves annealing after sweep 29)
        assert eng.timing()["exchange"][1] == 2 * sweeps
        plain = setup.engine(use_validation_stop=False)
        plain.sweep(sweeps)
        a, b = eng.state(), plain.state()
        np.testing.assert_allclose(a[0], b[0], rtol=1e-10)       # summation order only (fold vs k_colreduce)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-10)
        assert np.array_equal(a[2], b[2])
        assert np.array_equal(eng.communities(), plain.communities())
        np.testing.assert_allclose(eng.rows(), plain.rows(), rtol=1e-10)

        class _NoDist:          # the caller-driven protocol with a world of one: no exchange
            pass
        shard = HipShard(setup, 0, 1, 0, use_validation_stop=False)
        ShardedSweep(shard, _NoDist()).sweep(sweeps)
        c = shard.engine.state()
        # (the library's own driver leaves the K-vectors for its collectives from inside the kernels at K <= 32, the
        #  caller-driven phases through k_colreduce: another fixed summation order)
        np.testing.assert_allclose(a[0], c[0], rtol=1e-10)
        np.testing.assert_allclose(a[1], c[1], rtol=1e-10)
        assert np.array_equal(a[2], c[2])
        # ... and the same run replayed as hipGraphs with the REAL librccl's collectives captured inside (no timing
        # brackets: those keep the sweeps eager): bit-identical to the eager run
        geng = setup.engine(use_validation_stop=False, node_block=(0, n), n_alloc=n)
        geng.comm_init(_svils.comm_unique_id(), 0, 1)
        geng.sweep_sharded(sweeps)
        geng.synchronize()
        gs = geng.state()
        assert np.array_equal(a[0], gs[0]) and np.array_equal(a[1], gs[1]) and np.array_equal(a[2], gs[2])
        assert np.array_equal(eng.rows(), geng.rows())
    # wrong node block for the rank: refused, loudly
    eng = setup.engine(use_validation_stop=False)
    with pytest.raises(_svils.SvilsError):
        eng.comm_init(_svils.comm_unique_id(), 1, 2)
    # caller-given blocks that do not match the handle's: refused as well
    with pytest.raises(_svils.SvilsError):
        eng.set_node_blocks(0, 2, np.array([0, 400, n], dtype=np.uint32))


def test_sharded_driver_world1(graph_files):
    """ShardedSweep over a real process group of size 1 (nccl == RCCL) equals the plain engine."""
    import os
    import torch
    import torch.distributed as dist
    from svinet_amd.host_api import Setup
    from svinet_amd.sharded import HipShard, ShardedSweep

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        setup = Setup(graph_files["lfr"], 1000, 28)
        shard = HipShard(setup, 0, 1, 0, use_validation_stop=False)
        ShardedSweep(shard, dist).sweep(10)
        plain = setup.engine(use_validation_stop=False)
        plain.sweep(10)
        a, b = shard.engine.state(), plain.state()
        # same kernels; the sharded driver materialises the K-vectors with k_colreduce where the plain
        # engine folds the per-block partial rows in the consumers: a different (fixed) summation order
        np.testing.assert_allclose(a[0], b[0], rtol=1e-12)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-12)
        assert np.array_equal(a[2], b[2])
        # the aliasing tensors really see the device buffers
        shard.engine.synchronize()
        torch.cuda.synchronize()
        assert np.allclose(shard.rows[0][:1000, :28].cpu().numpy(), a[0], rtol=0, atol=0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,sweeps", [(2, 28, 40), (3, 64, 6)])
def test_two_processes_one_gpu(graph_files, tmp_path, world, k, sweeps):
    """svinet_amd/sharded.py end to end in separate processes (one per rank, as on a multi-GPU node),
    all on GPU 0 with a gloo group: the exchanged state equals the oracle's on every rank."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py")
    out = str(tmp_path / "state")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29600 + world), worker,
                        graph_files["lfr"], "1000", str(k), str(sweeps), out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    states = [np.load(out + ".%d.npz" % rk) for rk in range(world)]
    for s in states:
        assert np.max(np.abs(s["gamma"] - ref.gamma) / np.abs(ref.gamma)) < 1e-5
        assert np.max(np.abs(s["lam"] - ref.lam) / np.abs(ref.lam)) < 1e-5
        assert np.array_equal(s["conv"], ref.converged)
        assert int(s["iter"]) == ref.iter and bool(s["annealing"]) == ref.annealing
        np.testing.assert_allclose(s["rows"][:, 1:], ref.rows[1:, 1:], rtol=1e-7, atol=1e-12)
        assert np.array_equal(s["member"], ref.communities())
    for s in states[1:]:
        assert np.array_equal(s["gamma"], states[0]["gamma"]) and np.array_equal(s["lam"], states[0]["lam"])


def _run_workers(graph_files, tmp_path, world, k, steps, mode, port):
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py")
    out = str(tmp_path / "state")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), worker,
                        graph_files["lfr"], "1000", str(k), str(steps), out, mode],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return [np.load(out + ".%d.npz" % rk) for rk in range(world)]


@pytest.mark.parametrize("world,k,sweeps", [(2, 28, 30), (3, 64, 5)])
def test_sharded_minibatch_full_window_is_a_sweep(graph_files, tmp_path, world, k, sweeps):
    """sharded mini-batch steps with one window per block and step size 1 (kappa = 0) are full sweeps:
    separate processes, gloo group on GPU 0, state equal to the oracle's on every rank"""
    states = _run_workers(graph_files, tmp_path, world, k, sweeps, "step:1:0", 29610 + world)
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    for s in states:
        assert np.max(np.abs(s["gamma"] - ref.gamma) / np.abs(ref.gamma)) < 1e-5
        assert np.max(np.abs(s["lam"] - ref.lam) / np.abs(ref.lam)) < 1e-5
        assert np.array_equal(s["conv"], ref.converged)
        assert int(s["iter"]) == ref.iter and bool(s["annealing"]) == ref.annealing
        np.testing.assert_allclose(s["rows"][:, 1:], ref.rows[1:, 1:], rtol=1e-7, atol=1e-12)
    for s in states[1:]:
        assert np.array_equal(s["gamma"], states[0]["gamma"]) and np.array_equal(s["lam"], states[0]["lam"])


def test_sharded_minibatch_windows(graph_files, tmp_path):
    """3 windows per block, damped steps, 2 ranks: 90 steps = 30 passes over the nodes.  The replicated state
    is bit-identical on both ranks, every node has been updated (sum_k mphi = 1/2, quirk Q3), the held-out
    likelihood improves."""
    states = _run_workers(graph_files, tmp_path, 2, 28, 90, "step:3:0.5", 29620)
    a, b = states
    for key in ("gamma", "lam", "conv", "rows", "mphi"):
        assert np.array_equal(a[key], b[key]), key
    assert np.isfinite(a["gamma"]).all() and (a["gamma"] > 0).all() and (a["lam"] > 0).all()
    np.testing.assert_allclose(a["mphi"].sum(1), 0.5, rtol=1e-9)
    assert int(a["iter"]) == 90 and a["rows"].shape[0] == 90
    assert a["rows"][-1, 9] > a["rows"][0, 9]


@pytest.mark.parametrize("k", [28, 64])
def test_native_step_sharded_world1(graph_files, k):
    """svils_step_sharded (mini-batch steps with the exchanges inside the library: K-vector all-reduces and the
    broadcasts of every rank's window rows) with a communicator of ONE rank: equals svils_step on a plain handle
    (phase-split vs fused launches: summation order only), and the collectives really ran."""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    n, steps, bn = 1000, 24, 300
    setup = Setup(graph_files["lfr"], n, k)
    eng = setup.engine(use_validation_stop=False, node_block=(0, n), n_alloc=n)
    eng.comm_init(_svils.comm_unique_id(), 0, 1)        # before svils_set_stochastic: either order is accepted
    eng.set_stochastic(batch_nodes=bn, tau0=4.0, kappa=0.6, shard_block=n)
    eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
    eng.step_sharded(steps)
    eng.synchronize()
    assert eng.timing()["exchange"][1] == 3 * steps
    plain = setup.engine(use_validation_stop=False)
    plain.set_stochastic(batch_nodes=bn, tau0=4.0, kappa=0.6)
    plain.step(steps)
    a, b = eng.state(), plain.state()
    np.testing.assert_allclose(a[0], b[0], rtol=1e-9)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-9)
    assert np.array_equal(a[2], b[2])
    np.testing.assert_allclose(eng.rows(), plain.rows(), rtol=1e-9, atol=1e-12)
    with pytest.raises(_svils.SvilsError):
        plain.step_sharded(1)                            # no shard_block: not a node-block mini-batch handle
