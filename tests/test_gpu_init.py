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
