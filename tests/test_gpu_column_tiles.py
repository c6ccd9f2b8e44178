"""-m gpu: k above SVILS_MAX_K = 2048 on one device (include/svils.h, "column-tiled handle"): svils_create builds
ceil(k / 2048) column slices of every row -- the K-sharded layout with all its ranks on one stream -- and svils_sweep
drives their phases with the four exchanges summed in place.  The reference has no such limit short of its 16-bit
community ids (src/linksampling.cc:635), so these runs are held to the oracle like every other K."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _random_graph(rng, n, m):
    a = rng.integers(0, n, size=m)
    b = rng.integers(0, n, size=m)
    hub = np.stack([np.full(n // 3, int(rng.integers(0, n))), rng.integers(0, n, size=n // 3)], 1)
    return (np.concatenate([np.stack([a, b], 1), hub]) * 3 + 5).astype(np.int32)


@pytest.mark.parametrize("k,n", [(2049, 30), (2500, 80), (4100, 100)])     # two tiles (1024 + 1025, 1250 + 1250), three tiles
def test_tiled_handle_equals_oracle(k, n):
    from svinet_amd.host_api import Setup
    rng = np.random.default_rng(k)
    pairs = _random_graph(rng, n, 5 * n)
    s = Setup(n=n, k=k, pairs=pairs, heldout_ratio=0.05)
    ref = O.LinkSampling(O.Network(n=n, pairs=pairs), k, heldout_ratio=0.05, use_validation_stop=False)
    eng = s.engine(use_validation_stop=False)

    def both(nsw):
        for _ in range(nsw):
            ref.sweep()
        eng.sweep(nsw)

    def check(tag):
        g, lam, conv = eng.state()
        assert g.shape == (s.n, k) and lam.shape[0] == k, tag
        assert np.max(np.abs(g - ref.gamma) / ref.gamma) < 1e-7, tag
        assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-7, tag
        assert np.array_equal(conv, ref.converged), tag
        assert np.array_equal(eng.communities(), ref.communities()), tag
        c = eng.control()
        assert (c.links_dense, c.links_sparse, c.links_shortcut) == ref.link_counts(), tag

    # the constructor's likelihood row (svils_validation_row) before any sweep
    row0 = eng.validation_row()
    np.testing.assert_allclose(row0[1:], ref.rows[0, 1:], rtol=1e-7, atol=1e-12)
    both(3)
    check("dense")
    np.testing.assert_allclose(eng.rows()[:, 1:], ref.rows[1:, 1:], rtol=1e-7, atol=1e-12)
    # one-converged shortcuts, communities on both sides of a tile edge and k itself (quirk Q2 with pc == K) -- from the
    # seeded state again and past the annealing phase: with thousands of communities on a graph of a few hundred links the
    # annealing scale ones / sum[k] (src/linksampling.cc:541-542) takes gamma past 1e170 within three sweeps and to inf
    # within five, in the reference's arithmetic as much as here
    conv = np.zeros(s.n, dtype=np.uint32)
    idx = rng.choice(s.n, size=s.n // 3, replace=False)
    conv[idx] = rng.choice([1, k // 2, k // 2 + 1, k - 1, k], size=idx.size)
    ref.set_converged(conv)
    eng.set_state(s.gamma, s.lam, conv)
    ref.set_gamma(s.gamma); ref.set_lambda(s.lam); ref.refresh()
    ref.annealing = False
    eng.set_control(annealing=0)
    both(2)
    check("shortcuts")
    # active-set path
    ref.iter = 1500
    eng.set_control(iter=1500)
    both(2)
    check("sparse")
    # (node, community) pairs: by node, every pair of the tag matrix
    tags = eng.community_tags()
    m = ref.communities()
    want = np.argwhere(m != 0)
    assert np.array_equal(np.asarray(tags).reshape(-1, 2), want)


def test_tiled_handle_says_what_it_does_not_do():
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    rng = np.random.default_rng(3)
    s = Setup(n=20, k=2100, pairs=_random_graph(rng, 20, 80), heldout_ratio=0.05)
    eng = s.engine(use_validation_stop=False)
    with pytest.raises(_svils.SvilsError) as ei:
        eng.report_enqueue(0, 0, True)
    assert ei.value.code == -4 and "column-tiled" in str(ei.value)
    with pytest.raises(_svils.SvilsError) as ei:
        eng.set_stochastic(batch_nodes=4)
    assert ei.value.code == -4
    with pytest.raises(_svils.SvilsError) as ei:
        _svils.Engine(20, 70000, ones=5, ones_prob=0.1)
    assert ei.value.code == -4 and "SVILS_MAX_K_TOTAL" in str(ei.value)


def test_cli_k_above_max_k(graph_files, tmp_path):
    """the drop-in binary with -k 2100: files equal to the oracle's writers (synchronous loop, a column-tiled handle)"""
    from conftest import ROOT
    svinet = os.path.join(ROOT, "svinet_amd", "bin", "svinet")
    r = subprocess.run([svinet, "-file", graph_files["assort"], "-n", "75", "-k", "2100", "-link-sampling", "-no-stop", "-max-iterations", "3"],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n75-k2100-mmsb-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 2100, use_validation_stop=False, max_iterations=3)
    n = 0
    while ref.sweep() == 0:
        n += 1
    assert n == 4                                    # quirk Q8: N+1 sweeps
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    for name, skip, atol in (("gamma.txt", 2, 1.1e-5), ("lambda.txt", 1, 1.1e-5), ("groups.txt", 2, 1.1e-3)):
        a, b = np.loadtxt(d / name), np.loadtxt(rd / name)
        assert a.shape == b.shape and np.array_equal(a[:, :skip], b[:, :skip])
        np.testing.assert_allclose(a[:, skip:], b[:, skip:], rtol=1e-5, atol=atol)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v = np.loadtxt(d / "validation.txt")
    assert v.shape == (5, 11)
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)
