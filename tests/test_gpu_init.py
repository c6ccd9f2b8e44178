"""-m gpu: init_gamma2 (src/linksampling.cc:374-401) ON THE DEVICE -- svils_init_gamma regenerates the reference's MT19937
draws from jump-ahead states, normalises them per link and adds them into the gamma rows in the reference's order -- against the
host path (host/linksampling.cc: init_gamma2, itself pinned on the authors' constructor rows G3 / G4 and on the oracle):
the same bits."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SVINET = os.path.join(ROOT, "svinet_amd", "bin", "svinet")


@pytest.mark.parametrize("graph,n,k,streams", [("lfr", 1000, 28, (1, 624)), ("lfr", 1000, 28, (7, 624 * 200)), ("astroph", 17903, 20, (64, 624 * 99)),
                                                ("lfr", 1000, 200, (13, 624 * 737)), ("astroph", 17903, 100, (300, 624 * 106)),
                                                ("lfr", 1000, 600, (40, 624 * 719))])
def test_device_init_gamma_equals_the_host_path_bit_for_bit(graph_files, graph, n, k, streams):
    """K = 20 .. 600 (one to ten columns per lane), stream boundaries that fall inside links and inside twists, one stream alone"""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    s = Setup(graph_files[graph], n, k)
    edges = s.init_links()
    total = edges.shape[0] * k
    ns, per = streams
    if ns == 1:
        per = (total + 623) // 624 * 624
    ns = (total + per - 1) // per
    st = s.init_streams(ns, per)
    eng = _svils.Engine(s.n, s.k, ones=s.ones, ones_prob=s.ones_prob, eta=s.eta, use_validation_stop=False)
    eng.set_graph(s.links)
    eng.set_validation(s.validation_sorted)
    eng.init_gamma(edges, st, per, s.lam)
    g, lam, conv = eng.state()
    assert np.array_equal(g, s.gamma), float(np.max(np.abs(g - s.gamma)))
    assert np.array_equal(lam, s.lam) and not conv.any()
    # ... and the engine is in the state svils_set_state leaves: the constructor's likelihood row and the first sweeps agree
    ref = s.engine(use_validation_stop=False)
    assert np.array_equal(eng.validation_row(), ref.validation_row())
    eng.sweep(3)
    ref.sweep(3)
    assert np.array_equal(eng.state()[0], ref.state()[0]) and np.array_equal(eng.rows(), ref.rows())
    # a wrong cover of the stream is refused
    with pytest.raises(_svils.SvilsError):
        eng.init_gamma(edges, st[:1], per if ns > 1 else per - 624, s.lam)


def test_cli_device_init_leaves_the_same_files(graph_files, tmp_path):
    """`svinet ... -link-sampling` with init_gamma2 on the device (SVINET_INIT_DEVICE=1; by itself from E k >= 2^24 uniforms on)
    and on the host (=0): every file of the run byte for byte"""
    outs = []
    for flag in ("1", "0"):
        d = tmp_path / ("init" + flag)
        d.mkdir()
        env = dict(os.environ, SVINET_INIT_DEVICE=flag, SVINET_TRACE_LOOP="1")
        r = subprocess.run([SVINET, "-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop", "-max-iterations", "12"],
                           cwd=str(d), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("init_gamma2 on the device" in r.stderr) == (flag == "1")
        outs.append(d / "n1000-k28-mmsb-linksampling")
    for name in ("gamma.txt", "lambda.txt", "groups.txt", "communities.txt", "validation-edges.txt"):
        assert (outs[0] / name).read_bytes() == (outs[1] / name).read_bytes(), name
    va, vb = np.loadtxt(outs[0] / "validation.txt"), np.loadtxt(outs[1] / "validation.txt")
    assert np.array_equal(np.delete(va, 1, axis=1), np.delete(vb, 1, axis=1))      # (column 1 is the wall-clock duration)


def test_device_init_on_node_block_and_ksharded_handles(graph_files):
    """svils_init_gamma on the handles of the two multi-GPU layouts: a node-block handle (the replicated state: every row) and
    K-sharded handles (uneven column slices of every link's vector, divided by the sum over ALL columns) -- the host path's bits"""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    from svinet_amd.ksharded import KShard, column_slices, init_virtual, sweep_virtual
    s = Setup(graph_files["lfr"], 1000, 100)
    nb = s.engine(use_validation_stop=False, node_block=(250, 700))
    s.device_init(nb)
    assert np.array_equal(nb.state()[0], s.gamma)
    for (k0, k1) in column_slices(100, 3):
        e = _svils.Engine(s.n, s.k, ones=s.ones, ones_prob=s.ones_prob, eta=s.eta, use_validation_stop=False, k_slice=(k0, k1))
        e.set_graph(s.links)
        e.set_validation(s.validation_sorted)
        s.device_init(e, lam=np.ascontiguousarray(s.lam[k0:k1]))
        assert np.array_equal(e.state()[0], s.gamma[:, k0:k1]), (k0, k1)
    # a Setup WITHOUT a host gamma (what bench.py and the config-5 tests use at n = 1e6): its engines and K-shards draw on the
    # device and then run like the ones that were handed the host's array
    s2 = Setup(graph_files["lfr"], 1000, 100, host_gamma=False)
    with pytest.raises(AttributeError):
        s2.gamma
    a, b = s2.engine(use_validation_stop=False), s.engine(use_validation_stop=False)
    assert np.array_equal(a.state()[0], s.gamma)
    a.sweep(5)
    b.sweep(5)
    assert np.array_equal(a.state()[0], b.state()[0])
    sh2 = [KShard(s2, r, 3, 0, use_validation_stop=False) for r in range(3)]
    sh1 = [KShard(s, r, 3, 0, use_validation_stop=False) for r in range(3)]
    for sh in (sh1, sh2):
        init_virtual(sh)
        sweep_virtual(sh, 4)
    for x, y in zip(sh1, sh2):
        assert np.array_equal(x.engine.state()[0], y.engine.state()[0])
