"""-m gpu: the library's OWN multi-rank drivers with rank > 0.

svils_sweep_sharded, svils_step_sharded, svils_sweep_ksharded (+ log-domain), svils_comm_allgather_host,
svils_gather_communities and the CLI `svinet -gpus N [-kshard | -minibatch m]` issue their collectives inside
libsvils through a table of dlsym'd nccl* entry points.  On an 8-GPU node that table is RCCL; on the one-GPU test
box RCCL refuses two ranks on one device, so these tests bind it (SVILS_RCCL_LIBRARY) to tests/fakerccl -- a
tests-only transport that lets several processes on ONE GPU form a communicator -- and hold every rank's result
against the oracle.  What runs here that no other test runs: every `rank * B * ld` offset, the in-place
all-gathers, the world x 3 grouped broadcasts of exchange_windows with their roots, the collective staging of
svils_comm_allgather_host, the pipe hand-shake of the communicator id and the reaping logic of `svinet -gpus N`.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, "fakerccl", "libfakerccl.so")
SVINET = os.path.join(ROOT, "svinet_amd", "bin", "svinet")


def _env(tmp_path=None, sync=False):
    """The transport runs in its ASYNCHRONOUS mode (RCCL's stream-order contract and nothing more, every collective
    stretched by 300 us) unless a test asks for the synchronous one: a missing hipStreamWaitEvent between the library's
    compute and communication streams then gives a wrong answer instead of being hidden (tests/test_gpu_fakerccl_async.py
    shows that it does)."""
    if not os.path.exists(FAKE):
        import __graft_entry__ as ge
        ge.build_test_transport()
    env = dict(os.environ)
    env["SVILS_RCCL_LIBRARY"] = FAKE
    env["FAKERCCL_TIMEOUT_S"] = "180"
    env["FAKERCCL_ASYNC"] = "0" if sync else "1"
    env["FAKERCCL_DELAY_US"] = "0" if sync else "300"
    if tmp_path is not None:
        env["FAKERCCL_STATS"] = str(tmp_path / "fakerccl.stats")
    return env


def _read_stats(path, world):
    """-> (collectives rank 0 executed, summed over its communicators; number of communicators).  Every communicator must
    have been joined by every rank, and every rank must have executed the same number of collectives on it."""
    by_comm = {}
    for line in open(path):
        r, w, calls, moved, name = line.split()
        by_comm.setdefault(name, []).append((int(r), int(w), int(calls)))
    total = 0
    for name, rows in by_comm.items():
        assert sorted(x[0] for x in rows) == list(range(world)) and all(x[1] == world for x in rows), (name, rows)
        assert len({x[2] for x in rows}) == 1, (name, rows)
        total += rows[0][2]
    return total, len(by_comm)


def _run_ranks(tmp_path, path, n, k, count, world, mode, extra_env=None, sync=False):
    out = str(tmp_path / "state")
    worker = os.path.join(HERE, "native_rank_worker.py")
    env = _env(tmp_path, sync=sync)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, worker, path, str(n), str(k), str(count), out, str(r), str(world), mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = []
    for p in procs:
        try:
            _, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append(err)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, errs[r][-3000:])
    return [np.load(out + ".%d.npz" % r) for r in range(world)], _read_stats(env["FAKERCCL_STATS"], world)


def _oracle(path, n, k, sweeps, **kw):
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False, **kw)
    for _ in range(sweeps):
        ref.sweep()
    return ref


def _row_collectives(ref, n, world, chunks):
    """collectives that carry the rows of one sweep: one all-gather of the slices padded to the largest block while the
    blocks are near-equal (world * bmax <= 1.5 n), else one in-place broadcast per (non-empty) block with its exact
    count; chunks x world broadcasts when pipelined (exchange_rows_and_expand, svils_api.hip)"""
    if chunks != 1:
        return chunks * world
    from svinet_amd.sharded import balanced_bounds
    b = np.asarray(balanced_bounds(ref.links, n, world), dtype=np.int64)
    sizes = np.diff(b)
    return 1 if 2 * world * int(sizes.max()) <= 3 * n else int((sizes > 0).sum())


def _check_node_block(states, ref, n, world, tags=True):
    from svinet_amd.sharded import block_size
    B = block_size(n, world)
    for s in states:
        assert np.max(np.abs(s["gamma"] - ref.gamma) / np.abs(ref.gamma)) < 1e-9
        assert np.max(np.abs(s["lam"] - ref.lam) / np.abs(ref.lam)) < 1e-9
        assert np.array_equal(s["conv"], ref.converged)
        assert int(s["iter"]) == ref.iter and bool(s["annealing"]) == ref.annealing
        np.testing.assert_allclose(s["rows"][:, 1:], ref.rows[1:, 1:], rtol=1e-7, atol=1e-12)
        if tags:   # svils_gather_communities: every rank ends up with every block's tags
            assert np.array_equal(s["member"], ref.communities())
    for s in states[1:]:       # the replicated state is bit-identical across ranks
        assert np.array_equal(s["gamma"], states[0]["gamma"]) and np.array_equal(s["lam"], states[0]["lam"])
    return B


@pytest.mark.parametrize("graph,world,k,sweeps,chunks,sync", [("lfr", 2, 28, 40, 1, False), ("lfr", 3, 64, 6, 1, False), ("lfr", 3, 28, 35, 1, False),
                                                               ("astroph", 2, 200, 3, 1, False),
                                                               # the pipelined row exchange (chunks on a stream and a communicator of
                                                               # their own, each expanded while the next one travels): 3 and 5 chunks
                                                               # of uneven size
                                                               ("lfr", 3, 28, 35, 3, False), ("lfr", 2, 100, 8, 5, False), ("astroph", 3, 200, 3, 4, False),
                                                               # and once each on the synchronous transport
                                                               ("lfr", 2, 28, 40, 1, True), ("lfr", 3, 28, 35, 3, True),
                                                               # the world sizes the driver's scaling run uses (2, 4, 8): four and
                                                               # eight node blocks, the headline graph at its own K on eight
                                                               ("lfr", 4, 28, 35, 1, False), ("lfr", 8, 28, 35, 3, False),
                                                               ("astroph", 8, 20, 6, 1, False), ("astroph", 4, 200, 3, 2, False),
                                                               # BASELINE config 4 as its 8-GPU run executes it: ca-AstroPh K = 200 on
                                                               # eight balanced node blocks (exact-count row exchange: 8 x bmax = 3.0 n)
                                                               ("astroph", 8, 200, 3, 2, False)])
def test_native_sweep_sharded_ranks(graph_files, tmp_path, graph, world, k, sweeps, chunks, sync):
    """svils_sweep_sharded in `world` processes on WORK-BALANCED node blocks (svils_balance_node_blocks: blocks of
    different sizes): per sweep one grouped {all-reduce of sum[k], in-place all-gather of the staged rows padded to the
    largest block} and the all-reduce of s1,s2,s3 -- two exchange points in both phases of the run (LFR K=28 leaves
    annealing at sweep 29)"""
    path, n = graph_files[graph], {"lfr": 1000, "astroph": 17903}[graph]
    states, (calls, ncomm) = _run_ranks(tmp_path, path, n, k, sweeps, world, "sweep", {"SVILS_XCHUNKS": str(chunks)}, sync=sync)
    ref = _oracle(path, n, k, sweeps)
    _check_node_block(states, ref, n, world)
    if (graph, k, sweeps) == ("lfr", 28, 35) or (graph, k, sweeps) == ("lfr", 28, 40):
        assert not ref.annealing        # the run crossed the switch: both phases ran through the same two exchange points
    for s in states:
        assert int(s["exchanges"]) == 2 * sweeps
    # collectives on the wire: per sweep all-reduce(sum) + rows + all-reduce(s1,s2,s3), + the tag gather (one broadcast
    # per block); rows = 1 all-gather, or chunks x world broadcasts when pipelined
    # ... the chunks on a second communicator, whose id travelled as one more broadcast on the first
    rows = _row_collectives(ref, n, world, chunks)
    assert ncomm == (1 if chunks == 1 else 2)
    assert calls == (2 + rows) * sweeps + world + (0 if chunks == 1 else 1)
    for s in states:
        assert bool(s["row_comm"]) == (chunks > 1) and int(s["comm_nranks"]) == world
    # the blocks really differ in size on the headline graph (hubs first: rank 0 owns the fewest nodes)
    if graph == "astroph":
        from svinet_amd.host_api import Setup
        from svinet_amd.sharded import balanced_bounds
        b = balanced_bounds(Setup(path, n, k).links, n, world).astype(np.int64)
        assert np.diff(b)[0] * 2 < np.diff(b)[-1]


def test_tags_after_a_seen_stop_wait_for_the_gather(graph_files, tmp_path):
    """once a control block that says `stopped` has reached the host the getters skip their stream wait
    -- but svils_gather_communities, called after it (the CLI's do_on_stop), ENQUEUES broadcasts that write the other
    blocks' rows of the community bitmask.  On the asynchronous transport with every collective stretched by 30 ms a getter
    that does not wait reads rank 0's bitmask with the other blocks' rows stale; every rank's tags, read straight after the
    gather, must equal the oracle's."""
    path, n, k, world = graph_files["lfr"], 1000, 28, 3
    ref = O.LinkSampling(O.Network(path, n), k)
    nsw = 0
    while ref.sweep() != 2:
        nsw += 1
        assert nsw < 400
    states, _ = _run_ranks(tmp_path, path, n, k, nsw + 1 + 8, world, "sweep-stop", {"FAKERCCL_DELAY_US": "30000", "NATIVE_RANK_NO_TIMING": "1"})
    want = ref.communities()
    assert want.sum() > 100
    for s in states:
        assert int(s["iter"]) == ref.iter and np.array_equal(s["conv"], ref.converged)
        assert np.array_equal(s["member_early"], want)
        assert np.array_equal(s["member"], want)


@pytest.mark.parametrize("graph,world,k,sweeps,chunks", [("lfr", 2, 28, 70, 1), ("lfr", 3, 28, 45, 3), ("astroph", 4, 20, 12, 1),
                                                          ("lfr", 2, 100, 21, 2)])
def test_native_sweep_sharded_graph_replay(graph_files, tmp_path, graph, world, k, sweeps, chunks):
    """the same driver with its hipGraphs: no timing brackets, SVILS_GRAPH_AFTER=0 (conftest) -- the first sweep eager, the
    rest replayed as captured graphs WITH their collectives (and, pipelined, the fork to the communication stream and the
    second communicator inside the capture).  The tests' transport executes a captured collective at every replay.
    Every rank equals the oracle; the transport really ran a collective per exchange of every sweep."""
    path, n = graph_files[graph], {"lfr": 1000, "astroph": 17903}[graph]
    states, _ = _run_ranks(tmp_path, path, n, k, sweeps, world, "sweep",
                           {"SVILS_XCHUNKS": str(chunks), "NATIVE_RANK_NO_TIMING": "1", "SVILS_GRAPH_AFTER": "0", "FAKERCCL_EXECUTED": str(tmp_path / "exe")})
    ref = _oracle(path, n, k, sweeps)
    _check_node_block(states, ref, n, world)
    rows = _row_collectives(ref, n, world, chunks)
    executed = int(open(str(tmp_path / "exe")).read().split()[0])      # rank 0's count of collectives that RAN (eager + replayed)
    assert executed == (2 + rows) * sweeps + world + (0 if chunks == 1 else 1)
    captured = int(open(str(tmp_path / "exe")).read().split()[1])      # of which from captured graphs
    assert captured >= (2 + rows) * (sweeps - 1 - 3)                   # all but the eager first sweep and at most 3 single leftovers


@pytest.mark.parametrize("world,k,steps,mode", [(2, 28, 30, "step:1:0"), (3, 64, 5, "step:1:0")])
def test_native_step_sharded_full_window_is_a_sweep(graph_files, tmp_path, world, k, steps, mode):
    """svils_step_sharded with one window per block and step size 1: full sweeps, so every rank equals the oracle.
    The window rows travel as world x 3 in-place broadcasts, rank r the root of its own window."""
    path, n = graph_files["lfr"], 1000
    states, (calls, _) = _run_ranks(tmp_path, path, n, k, steps, world, mode)
    ref = _oracle(path, n, k, steps)
    _check_node_block(states, ref, n, world, tags=False)
    assert calls == steps * (2 + 3 * world) + world      # + the tag gather: one broadcast per block


@pytest.mark.parametrize("world", [2, 3])
def test_native_step_sharded_windows(graph_files, tmp_path, world):
    """3 windows per block, damped steps: 90 steps.  The replicated state is bit-identical on all ranks, every node
    has been updated (sum_k mphi = 1/2, quirk Q3), the held-out likelihood improves -- and with two ranks (where the
    order of a two-term sum cannot differ between transports) the whole run equals the caller-driven protocol of
    svinet_amd/sharded.py over gloo bit for bit: same kernels, same exchange points."""
    path, n, k, steps = graph_files["lfr"], 1000, 28, 90
    states, _ = _run_ranks(tmp_path, path, n, k, steps, world, "step:3:0.5")
    a = states[0]
    for b in states[1:]:
        for key in ("gamma", "lam", "conv", "rows", "mphi"):
            assert np.array_equal(a[key], b[key]), key
    assert np.isfinite(a["gamma"]).all() and (a["gamma"] > 0).all() and (a["lam"] > 0).all()
    np.testing.assert_allclose(a["mphi"].sum(1), 0.5, rtol=1e-9)
    assert int(a["iter"]) == steps and a["rows"].shape[0] == steps
    assert a["rows"][-1, 9] > a["rows"][0, 9]
    if world != 2:
        return
    out = str(tmp_path / "pystate")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", "29671", os.path.join(HERE, "shard_worker.py"),
                        path, str(n), str(k), str(steps), out, "step:3:0.5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    py = np.load(out + ".0.npz")
    for key in ("gamma", "lam", "conv", "rows"):
        assert np.array_equal(a[key], py[key]), key


@pytest.mark.parametrize("world,k,sweeps,mode", [(2, 28, 40, "kshard"), (3, 100, 6, "kshard"), (3, 130, 5, "kshard-log"),
                                                  (2, 28, 40, "kshard-log"), (3, 28, 30, "kshard-lowt"),
                                                  # four and eight column slices (K = 28 on eight ranks: slices of 3 and 4 columns)
                                                  (4, 28, 35, "kshard"), (8, 28, 35, "kshard"), (8, 200, 4, "kshard"),
                                                  # BASELINE config 4 K-sharded over eight ranks: ca-AstroPh, slices of 25 columns
                                                  (8, 200, 3, "kshard-astroph")])
def test_native_sweep_ksharded_ranks(graph_files, tmp_path, world, k, sweeps, mode):
    """svils_ksh_init_state + svils_sweep_ksharded in `world` processes (uneven slices at K=100/3, 130/3): the column
    slices put together equal the oracle, flags / rows / counters replicated; svils_validation_row and
    svils_comm_allgather_host (collective staging) with rank > 0"""
    path, n = graph_files["lfr"], 1000
    if mode == "kshard-astroph":
        path, n, mode = graph_files["astroph"], 17903, "kshard"
    states, (calls, _) = _run_ranks(tmp_path, path, n, k, sweeps, world, mode)
    ref = _oracle(path, n, k, sweeps, **({"link_thresh": 0.3} if mode == "kshard-lowt" else {}))
    g = np.concatenate([s["gamma"] for s in states], 1)
    lam = np.concatenate([s["lam"] for s in states], 0)
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    want = ref.communities()
    per_sweep = {"kshard": 4, "kshard-log": 5, "kshard-lowt": 6}[mode]   # + the max, + the arg-max (MIN) of every link
    for r, s in enumerate(states):
        assert np.array_equal(s["conv"], ref.converged)
        assert int(s["iter"]) == ref.iter and bool(s["annealing"]) == ref.annealing
        np.testing.assert_allclose(s["rows"][:, 1:], ref.rows[1:, 1:], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(s["row0"][1:], ref.rows[0, 1:], rtol=1e-9, atol=1e-13)   # constructor-time row
        k0, k1 = int(s["k0"]), int(s["k1"])
        assert np.array_equal(s["member"], want[:, k0:k1])
        # what svils_comm_allgather_host handed to this rank: every rank's tags, rank by rank
        for q, t in enumerate(states):
            q0, q1 = int(t["k0"]), int(t["k1"])
            assert np.array_equal(s["gathered"][q][:, :q1 - q0], want[:, q0:q1])
        assert int(s["exchanges"]) == per_sweep * sweeps + 2   # + init rows + validation row
    # + 2 collectives of the staging (the agreement all-reduce and the gather itself)
    assert calls == per_sweep * sweeps + 2 + 2


@pytest.mark.parametrize("world,k,nwin,steps", [(2, 28, 3, 45), (3, 100, 4, 16)])
def test_native_step_ksharded_ranks(graph_files, tmp_path, world, k, nwin, steps):
    """svils_step_ksharded in `world` processes: every rank steps through the same windows on its own columns, the
    exchanges carry the window's share of the buffers (svils_ksh_buffer_ptr's sub-ranges).  The slices put together equal
    svils_step on one plain handle with the same windows and step sizes."""
    from svinet_amd.host_api import Setup
    path, n = graph_files["lfr"], 1000
    states, (calls, _) = _run_ranks(tmp_path, path, n, k, steps, world, "kstep:%d:0.6" % nwin)
    setup = Setup(path, n, k)
    plain = setup.engine(use_validation_stop=False)
    plain.set_stochastic(batch_nodes=(n + nwin - 1) // nwin, tau0=4.0, kappa=0.6, node_tau0=2.0, node_kappa=0.5)
    plain.step(steps)
    pg, pl, pc = plain.state()
    g = np.concatenate([s["gamma"] for s in states], 1)
    lam = np.concatenate([s["lam"] for s in states], 0)
    assert np.max(np.abs(g - pg) / np.abs(pg)) < 1e-9
    assert np.max(np.abs(lam - pl) / np.abs(pl)) < 1e-9
    for s in states:
        assert np.array_equal(s["conv"], pc)
        assert int(s["iter"]) == steps
        np.testing.assert_allclose(s["rows"][:, 1:], plain.rows()[:, 1:], rtol=1e-8, atol=1e-11)
    assert calls == 4 * steps + 2 + 2        # per step den, rowx, q2v, vdot; + init rows, the constructor row, the staged gather


# ----------------------------------------------------------------------------------------- the CLI, forked ranks
def _cli(tmp_path, args, world, timeout=900, env=None):
    env = env or _env(tmp_path)
    cmd = [SVINET] + args + ["-gpus", str(world), "-device-list", ",".join(["0"] * world)]
    return subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=timeout)


def _cmp_numeric(path_a, path_b, skip, atol):
    a, b = np.loadtxt(path_a), np.loadtxt(path_b)
    assert a.shape == b.shape
    assert np.array_equal(a[:, :skip], b[:, :skip])
    np.testing.assert_allclose(a[:, skip:], b[:, skip:], rtol=1e-5, atol=atol)


@pytest.mark.parametrize("world,extra", [(2, []), (3, ["-sweep-batch", "7"]), (2, ["-kshard"]), (3, ["-kshard", "-sweep-batch", "4"]),
                                         # (eight forked ranks pass as well -- profiles/r04t -- but take 5 minutes on the tests' transport:
                                         #  sixteen spinning host threads under the box's 16-core quota; four are enough here)
                                         (4, []), (4, ["-kshard"])])
def test_cli_gpus_ranks_files_equal_oracle(graph_files, tmp_path, world, extra):
    """`svinet -gpus N [-kshard]` with N forked ranks on GPU 0 (the id through the pipes, every rank its own
    LinkSampling, the gathers behind the files): rank 0's files equal the oracle's writers, nobody else writes"""
    path, n, k, M = graph_files["lfr"], 1000, 28, 40
    r = _cli(tmp_path, ["-file", path, "-n", str(n), "-k", str(k), "-link-sampling", "-no-stop", "-max-iterations", str(M)] + extra, world)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = tmp_path / ("n%d-k%d-mmsb-linksampling" % (n, k))
    ref = O.LinkSampling(O.Network(path, n), k, use_validation_stop=False, max_iterations=M)
    while ref.sweep() == 0:
        pass
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    _cmp_numeric(d / "groups.txt", rd / "groups.txt", 2, 1.1e-3)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v = np.loadtxt(d / "validation.txt")
    assert v.shape == (M + 2, 11)                    # constructor row + one per sweep (quirk Q8: M + 1 sweeps)
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)
    assert len([x for x in os.listdir(str(tmp_path)) if x.endswith("-linksampling")]) == 1
    calls, _ = _read_stats(str(tmp_path / "fakerccl.stats"), world)
    assert calls > M


def test_cli_gpus_load_test_ranks(graph_files, tmp_path):
    """`svinet -gpus 2 -load-test FILE`: the test set on node-block shards (every rank holds the replicated state and
    records the same test rows; rank 0 writes test.txt) -- files against the oracle's counterpart"""
    net = O.Network(graph_files["lfr"], 1000)
    s2i, e = net.seq2id(), net.edges()
    tp = np.concatenate([e[11::89], [[5, 800], [40, 77]]]).astype(np.uint32)
    tf = tmp_path / "test_pairs.txt"
    tf.write_text("".join("%d\t%d\n" % (s2i[a], s2i[b]) for a, b in tp))
    M = 30
    r = _cli(tmp_path, ["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop", "-max-iterations", str(M),
                        "-load-test", str(tf), "-sweep-batch", "5"], 2)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    ref = O.LinkSampling(net, 28, use_validation_stop=False, max_iterations=M, test_pairs=tp)
    while ref.sweep() == 0:
        pass
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    t = np.loadtxt(d / "test.txt")
    assert t.shape == (M + 1, 11)
    np.testing.assert_allclose(np.delete(t, 1, axis=1), ref.test_rows, rtol=0, atol=6e-10)


def test_cli_gpus_minibatch_ranks(graph_files, tmp_path):
    """`svinet -gpus 2 -minibatch m`: svils_step_sharded behind the forked command line (relabelling on every rank,
    shard_block, the window broadcasts).  Two ranks stepping through windows of m nodes of their own blocks visit the
    rows in a different grouping than one process does, so there is no file to equal; what must hold: the run
    completes, rank 0 writes a finite model and the held-out likelihood improves."""
    path, n, k = graph_files["lfr"], 1000, 28
    r = _cli(tmp_path, ["-file", path, "-n", str(n), "-k", str(k), "-link-sampling", "-rfreq", "4", "-no-stop", "-max-iterations", "79",
                        "-minibatch", "125", "-tau0", "4", "-kappa", "0.5", "-nodetau0", "4", "-nodekappa", "0.5", "-sweep-batch", "4"], 2)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = tmp_path / ("n%d-k%d-mmsb-linksampling" % (n, k))
    g = np.loadtxt(d / "gamma.txt")[:, 2:]
    assert g.shape == (n, k) and np.isfinite(g).all() and (g > 0).all()
    v = np.loadtxt(d / "validation.txt")
    assert v.shape[0] >= 20 and v[-1, 10] > v[1, 10]


def test_cli_gpus_kshard_minibatch_ranks(graph_files, tmp_path):
    """`svinet -gpus 2 -kshard -minibatch m`: two forked ranks, every one a column slice, stepping through the same
    windows -- the files equal those of the plain one-process mini-batch run (same windows, same step sizes)"""
    path, n, k = graph_files["lfr"], 1000, 28
    args = ["-file", path, "-n", str(n), "-k", str(k), "-link-sampling", "-rfreq", "5", "-no-stop", "-max-iterations", "59",
            "-minibatch", "250", "-tau0", "2", "-kappa", "0.5", "-nodetau0", "2", "-nodekappa", "0.5", "-sweep-batch", "5"]
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ra = subprocess.run([SVINET] + args, cwd=str(tmp_path / "a"), capture_output=True, text=True, timeout=600)
    assert ra.returncode == 0, ra.stderr[-2000:]
    rb = _cli(tmp_path / "b", args + ["-kshard"], 2, env=_env(tmp_path))
    assert rb.returncode == 0, (rb.stdout[-2000:], rb.stderr[-3000:])
    da, db = tmp_path / "a" / "n1000-k28-mmsb-linksampling", tmp_path / "b" / "n1000-k28-mmsb-linksampling"
    _cmp_numeric(da / "gamma.txt", db / "gamma.txt", 2, 2.1e-5)
    _cmp_numeric(da / "lambda.txt", db / "lambda.txt", 1, 2.1e-5)
    assert (da / "communities.txt").read_text() == (db / "communities.txt").read_text()
    va, vb = np.loadtxt(da / "validation.txt"), np.loadtxt(db / "validation.txt")
    np.testing.assert_allclose(np.delete(va, 1, axis=1), np.delete(vb, 1, axis=1), rtol=0, atol=2e-9)


def test_cli_gpus_rank_failure_does_not_hang(graph_files, tmp_path):
    """a rank that dies (here: a device ordinal that does not exist) takes the others with it instead of leaving
    them in a collective for ever; the parent returns non-zero (reaping in completion order)"""
    import time
    env = _env(tmp_path)
    env["FAKERCCL_TIMEOUT_S"] = "600"          # the peers must be ended by the parent, not by the transport's timeout
    t0 = time.time()
    r = _cli(tmp_path, ["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop", "-max-iterations", "5"],
             2, timeout=300, env=env)
    assert r.returncode == 0                    # sanity: the same command with good devices runs
    cmd = [SVINET, "-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop", "-max-iterations", "5",
           "-label", "bad", "-gpus", "2", "-device-list", "0,99"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert time.time() - t0 < 120


def test_cli_gpus_sigterm_is_collective(graph_files, tmp_path):
    """SIGTERM = "save the model and go on" (src/main.cc:29-40, src/linksampling.cc:763-766).  With -gpus N the parent
    passes the signal on, the ranks see it at different sweeps and agree at their next poll (do_on_stop is collective
    there): the model is written while the run goes on, and the run still ends normally."""
    import signal
    import time
    env = _env(tmp_path)
    cmd = [SVINET, "-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop", "-max-iterations", "1500",
           "-gpus", "2", "-device-list", "0,0"]
    p = subprocess.Popen(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    t0 = time.time()
    while time.time() - t0 < 120 and not ((d / "validation.txt").exists() and len((d / "validation.txt").read_text().split("\n")) > 20):
        time.sleep(0.05)
    assert p.poll() is None, "the run ended before the signal"
    p.send_signal(signal.SIGTERM)
    t1 = time.time()
    while time.time() - t1 < 60 and not (d / "gamma.txt").exists():
        time.sleep(0.02)
    saved_early = (d / "gamma.txt").exists() and p.poll() is None
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, err[-3000:]
    assert "Got signal. Saving model and groups." in out
    assert saved_early
    v = np.loadtxt(d / "validation.txt")
    assert v.shape[0] == 1502


def _check_bench_line(stdout):
    import json
    line = [l for l in stdout.split("\n") if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 10 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["roofline"]["launches_timed"] >= 10 and out["exchange"]["ms_per_sweep"] > 0
    assert out["roofline"]["frac"] <= 1.0
    # the communicator as the bound library describes it: two ranks, each asked about its own
    rc = out["rccl"]
    assert rc["nranks"] == [2] and sorted(r["rank"] for r in rc["ranks"]) == [0, 1]
    assert len({r["pid"] for r in rc["ranks"]}) == 2 and rc["library"][0].endswith("libfakerccl.so")
    assert out["n1_same_box"]["value"] > 0 and out["cpu_baseline"]["value"] > 0
    return out


def test_bench_bare_gpus_2_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (what a driver that only knows the N = 1 command types):
    bench.py spawns its two ranks itself and rank 0 prints the one JSON line -- here in test mode (both ranks on GPU 0,
    the tests' transport in asynchronous mode)."""
    env = _env(tmp_path)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--test-one-gpu",
                        "--extra-list", "config4_astroph_k200"], env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.split("\n") if l.startswith("{")]) == 1      # ONE line, from rank 0
    out = _check_bench_line(r.stdout)
    assert "error" not in out["sharded_extra"]["config4_astroph_k200"]


def test_bench_multi_gpu_code_path_two_ranks(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), in its test mode:
    both ranks on GPU 0, the library's collectives on the tests' transport.  The N > 1 path of bench.py -- sharded
    runner, event pass, side records in both layouts, the JSON line -- with rank > 0 before any 8-GPU node runs it."""
    import json
    env = _env(tmp_path)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29691", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--test-one-gpu", "--extra-list", "config4_astroph_k200,minibatch_steps_astroph_k20,ksharded_config4_astroph_k200"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    out = _check_bench_line(r.stdout)
    ex = out["sharded_extra"]
    for name in ("config4_astroph_k200", "minibatch_steps_astroph_k20", "ksharded_config4_astroph_k200"):
        assert "error" not in ex[name], ex[name]
        assert ex[name]["value"] > 0


def test_bench_side_record_that_hangs_does_not_cost_the_line(tmp_path):
    """a side record that never comes back (here: a test hook; on a node: a collective that blocks) is caught by the
    per-record watchdog: the JSON line still carries the headline value, the records measured before it, an error entry
    for the hung one and "not run" for those behind it (bench.SIDE_RECORDS gives the order); every rank exits"""
    import json
    env = _env(tmp_path)
    env["BENCH_TEST_HANG_RECORD"] = "ksharded_config4_astroph_k200"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--test-one-gpu",
                        "--record-timeout", "20", "--no-cpu-baseline",
                        "--extra-list", "config4_astroph_k200,minibatch_steps_astroph_k20,ksharded_config4_astroph_k200"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["n_gpus"] == 2
    ex = out["sharded_extra"]
    assert list(ex)[:1] == ["config4_astroph_k200"] and ex["config4_astroph_k200"]["value"] > 0
    assert ex["config4_astroph_k200"]["model_ms_per_step"] > 0          # the cost model's prediction beside the record
    assert "did not finish" in ex["ksharded_config4_astroph_k200"]["error"]
    assert "not run" in ex["minibatch_steps_astroph_k20"]["error"]
    assert out["model_ms_per_step"] > 0 and out["measured_over_model"] > 0
    assert out["load_balance"]["max_over_mean"] < 1.1 and out["rccl"]["devices_unique"] is False   # test mode: both ranks on GPU 0
