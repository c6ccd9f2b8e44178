"""-m gpu: every N > 1 entry point of the library bound to the REAL librccl (/opt/rocm/lib/librccl.so*), with the one
world size a one-GPU box can form: a communicator of one rank.  What no other box has shown yet is real RCCL carrying
two ranks of this code; what CAN be shown here is that every call the multi-rank drivers make -- communicator
creation, the second communicator formed by broadcasting an id over the first, grouped all-reduce + all-gather,
grouped in-place broadcasts with per-rank counts on a stream of their own, MAX / MIN reductions, the host-staged
gather, and all of it under hipGraph capture -- is accepted by the real library, completes and gives the plain
engine's numbers.  (The multi-rank arithmetic itself is held to the oracle on the tests' transport,
tests/test_gpu_native_ranks.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _real_rccl(eng):
    q = eng.comm_query()
    assert "librccl" in q["library"] and "fakerccl" not in q["library"], q
    assert q["nranks"] == 1 and q["rank"] == 0 and q["rccl_version"] > 20000, q
    return q


@pytest.fixture(autouse=True)
def _no_test_transport(monkeypatch):
    monkeypatch.delenv("SVILS_RCCL_LIBRARY", raising=False)


@pytest.mark.parametrize("key,n,k,sweeps,chunks", [("lfr", 1000, 28, 70, 3), ("lfr", 1000, 100, 21, 4), ("astroph", 17903, 20, 40, 2)])
def test_pipelined_row_exchange_second_communicator_and_graphs(graph_files, monkeypatch, key, n, k, sweeps, chunks):
    """svils_sweep_sharded with the pipelined row exchange forced on (SVILS_XCHUNKS): the second communicator (rank 0's
    fresh id broadcast over the first), grouped in-place ncclBroadcast per chunk on the communication stream, k_expand_all
    per chunk behind its event -- first eagerly, then captured into hipGraphs WITH the fork to the communication stream
    (SVILS_GRAPH_AFTER=0, no timing brackets) and replayed.  Equals the plain engine."""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    monkeypatch.setenv("SVILS_XCHUNKS", str(chunks))
    setup = Setup(graph_files[key], n, k)
    plain = setup.engine(use_validation_stop=False)
    plain.sweep(sweeps)
    for timed in (True, False):       # eager (timing brackets keep it so) / graph replay
        eng = setup.engine(use_validation_stop=False, node_block=(0, n))
        eng.comm_init(_svils.comm_unique_id(), 0, 1)
        if timed:
            eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
        eng.sweep_sharded(sweeps)
        eng.gather_communities()
        eng.synchronize()
        q = _real_rccl(eng)
        assert q["row_communicator"] is True
        a, b = eng.state(), plain.state()
        np.testing.assert_allclose(a[0], b[0], rtol=1e-10)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-10)
        assert np.array_equal(a[2], b[2]) and np.array_equal(eng.communities(), plain.communities())
        np.testing.assert_allclose(eng.rows(), plain.rows(), rtol=1e-10)
        if timed:
            assert eng.timing()["exchange"][1] == 2 * sweeps
            first = a
        else:
            assert np.array_equal(a[0], first[0]) and np.array_equal(a[1], first[1])     # replay == eager, bit for bit
        eng.close()


@pytest.mark.parametrize("key,n,k,sweeps", [("lfr", 1000, 28, 40), ("astroph", 17903, 200, 6)])
def test_grouped_allreduce_and_exact_count_broadcasts(graph_files, monkeypatch, key, n, k, sweeps):
    """the row exchange of work-balanced blocks that are far from equal: ONE grouped launch of {all-reduce of sum[k], an
    in-place broadcast per block with its exact row count} (SVILS_EXACT_ROWS forces that form on the single block of a world
    of one) -- accepted by the real librccl eagerly and under hipGraph capture, equal to the plain engine"""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    monkeypatch.setenv("SVILS_EXACT_ROWS", "1")
    setup = Setup(graph_files[key], n, k)
    plain = setup.engine(use_validation_stop=False)
    plain.sweep(sweeps)
    first = None
    for timed in (True, False):       # eager (timing brackets keep it so) / graph replay
        eng = setup.engine(use_validation_stop=False, node_block=(0, n))
        eng.comm_init(_svils.comm_unique_id(), 0, 1)
        if timed:
            eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
        eng.sweep_sharded(sweeps)
        eng.synchronize()
        _real_rccl(eng)
        a, b = eng.state(), plain.state()
        np.testing.assert_allclose(a[0], b[0], rtol=1e-10)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-10)
        assert np.array_equal(a[2], b[2])
        np.testing.assert_allclose(eng.rows(), plain.rows(), rtol=1e-10)
        if timed:
            assert eng.timing()["exchange"][1] == 2 * sweeps
            first = a
        else:
            assert np.array_equal(a[0], first[0]) and np.array_equal(a[1], first[1])     # replay == eager, bit for bit
        eng.close()


@pytest.mark.parametrize("mode", ["sum", "log", "lowt"])
def test_ksharded_reductions(graph_files, mode):
    """svils_sweep_ksharded's all-reduces with the real library: SUM (product form), + MAX (log-domain denominators),
    + MIN (link_thresh < 1/2: the lowest column attaining a link's maximum); svils_comm_allgather_host's agreement
    all-reduce (uint32) and staged all-gather (bytes); mini-batch steps on the K-sharded layout"""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    n, k = 1000, 64
    lt = 0.3 if mode == "lowt" else 0.5
    setup = Setup(graph_files["lfr"], n, k, link_thresh=lt)

    def ksh():
        e = _svils.Engine(n, k, ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta, link_thresh=lt,
                          lt_min_deg=setup.lt_min_deg, use_validation_stop=False, k_slice=(0, k))
        e.set_graph(setup.links)
        e.set_validation(setup.validation_sorted)
        e.set_state(setup.gamma, setup.lam)
        if mode == "log":
            e.ksh_log_domain(True)
        e.comm_init(_svils.comm_unique_id(), 0, 1)
        return e

    eng = ksh()
    eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
    eng.ksh_init_state()
    eng.sweep_ksharded(12)
    eng.synchronize()
    _real_rccl(eng)
    per_sweep = {"sum": 4, "log": 5, "lowt": 6}[mode]
    assert eng.timing()["exchange"][1] == per_sweep * 12 + 1
    plain = setup.engine(use_validation_stop=False)
    plain.sweep(12)
    a, b = eng.state(), plain.state()
    assert np.max(np.abs(a[0] - b[0]) / b[0]) < 1e-11 and np.max(np.abs(a[1] - b[1]) / np.abs(b[1])) < 1e-11
    assert np.array_equal(a[2], b[2]) and np.array_equal(eng.communities(), plain.communities())
    blob = np.arange(70001, dtype=np.uint8)
    assert np.array_equal(eng.allgather_host(blob, 1)[0], blob)
    assert np.array_equal(eng.allgather_host(np.arange(9, dtype=np.float64), 1)[0], np.arange(9.0))   # smaller: the staging is re-used
    # mini-batch steps, same layout
    st = ksh()
    st.set_stochastic(batch_nodes=250, tau0=4.0, kappa=0.6)
    st.ksh_init_state()
    st.step_ksharded(8)
    ps = setup.engine(use_validation_stop=False)
    ps.set_stochastic(batch_nodes=250, tau0=4.0, kappa=0.6)
    ps.step(8)
    c, d = st.state(), ps.state()
    assert np.max(np.abs(c[0] - d[0]) / d[0]) < 1e-9 and np.array_equal(c[2], d[2])


def test_bench_force_sharded_line(tmp_path):
    """bench.py --force-sharded: the N > 1 code path of the bench (gloo control plane, svils_comm_init on the real
    librccl, the graph-replayed sharded sweeps in the timed region, event pass, RCCL evidence) with a world of one; the
    sharded sweep must not be far behind the plain engine's on the same box (two more launches and two one-rank
    collectives per sweep)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("SVILS_RCCL_LIBRARY", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--steps", "64", "--warmup", "5", "--no-extra",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    assert out["value"] > 0 and out["rccl"]["nranks"] == [1] and "librccl" in out["rccl"]["library"][0]
    assert out["rccl"]["devices_unique"] is True and out["load_balance"]["max_over_mean"] == 1.0
    ratio = out["n1_same_box"]["value"] / out["value"]
    assert ratio < 2.0, ratio        # (recorded figures: profiles/README.md)
