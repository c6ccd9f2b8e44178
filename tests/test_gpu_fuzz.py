"""-m gpu: randomized small graphs / community counts against the oracle (both
kernel layouts: lane-per-link for K <= 32, row-per-wavefront above), including
declared-but-absent nodes, duplicate and reversed input lines, hubs, tiny K, the
converged shortcuts (forced by seeding converged flags) and the active-set path."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _random_graph(rng, n, m):
    a = rng.integers(0, n, size=m)
    b = rng.integers(0, n, size=m)
    pairs = np.stack([a, b], 1)
    pairs = np.concatenate([pairs, pairs[: m // 7, ::-1], pairs[: m // 11]])      # reversed + duplicate lines
    hub = np.stack([np.full(n // 3, int(rng.integers(0, n))), rng.integers(0, n, size=n // 3)], 1)
    return (np.concatenate([pairs, hub]) * 3 + 5).astype(np.int32)                # non-contiguous external ids


CASES = [(seed, n, k) for seed, (n, k) in enumerate([(12, 1), (30, 2), (40, 3), (64, 5), (100, 8), (130, 9), (150, 16),
                                                    (200, 17), (257, 24), (300, 31), (300, 32), (120, 33), (90, 40),
                                                    (80, 63), (70, 64), (60, 65), (50, 70), (45, 129)])]


@pytest.mark.parametrize("seed,n,k", CASES)
def test_random_graph(seed, n, k):
    from svinet_amd.host_api import Setup
    rng = np.random.default_rng(1000 + seed)
    pairs = _random_graph(rng, n, 6 * n)
    declared = n + int(rng.integers(0, 5))
    hr = float(rng.choice([0.0, 0.02, 0.05]))
    s = Setup(n=declared, k=k, pairs=pairs, heldout_ratio=hr, seed=seed)
    ref = O.LinkSampling(O.Network(n=declared, pairs=pairs), k, heldout_ratio=hr, seed=seed, use_validation_stop=False)
    if s.validation_sorted.shape[0] == 0:
        ref.set_skip_validation(True)
    assert np.array_equal(s.links, ref.links) and np.array_equal(s.gamma, ref.gamma)
    eng = s.engine(use_validation_stop=False)

    def both(nsw):
        for _ in range(nsw):
            ref.sweep()
        eng.sweep(nsw)

    def check(tag):
        g, lam, conv = eng.state()
        assert np.max(np.abs(g - ref.gamma) / ref.gamma) < 1e-7, tag
        assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-7, tag
        assert np.array_equal(conv, ref.converged), tag
        assert np.array_equal(eng.communities(), ref.communities()), tag
        c = eng.control()
        assert (c.links_dense, c.links_sparse, c.links_shortcut) == ref.link_counts(), tag

    both(4)
    check("dense")
    # force the one-converged shortcuts: mark a third of the nodes converged on both sides
    conv = np.zeros(s.n, dtype=np.uint32)
    idx = rng.choice(s.n, size=s.n // 3, replace=False)
    conv[idx] = rng.integers(1, k + 1, size=idx.size)      # includes k itself => quirk Q2 (pc == K)
    g, lam, _ = eng.state()
    ref.set_converged(conv)
    eng.set_state(g, lam, conv)
    ref.set_gamma(g); ref.set_lambda(lam); ref.refresh()
    both(3)
    check("shortcuts")
    # active-set path
    ref.iter = 1500
    eng.set_control(iter=1500)
    both(3)
    check("sparse")


@pytest.mark.parametrize("k,n", [(257, 40), (300, 50), (384, 40), (400, 40), (513, 30), (600, 40), (768, 36), (800, 30), (1100, 30), (2048, 64)])
def test_large_k_layouts(k, n):
    """V = 8, V = 12 (K = 513..768), V = 16 and V = 32 register layouts (K up to SVILS_MAX_K)"""
    from svinet_amd.host_api import Setup
    rng = np.random.default_rng(k)
    pairs = _random_graph(rng, n, 5 * n)
    s = Setup(n=n, k=k, pairs=pairs, heldout_ratio=0.05)
    ref = O.LinkSampling(O.Network(n=n, pairs=pairs), k, heldout_ratio=0.05, use_validation_stop=False)
    eng = s.engine(use_validation_stop=False)
    for _ in range(3):
        ref.sweep()
    eng.sweep(3)
    g, lam, conv = eng.state()
    assert np.max(np.abs(g - ref.gamma) / ref.gamma) < 1e-7
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-7
    assert np.array_equal(eng.communities(), ref.communities())
    np.testing.assert_allclose(eng.rows()[:, 1:], ref.rows[1:, 1:], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("rfreq", [2, 3, 5])
def test_reportfreq(graph_files, rfreq):
    """-rfreq after -link-sampling: likelihood every rfreq sweeps, tagging only on the sweep
    before a report (src/linksampling.cc:768-787)"""
    from svinet_amd.host_api import Setup
    s = Setup(graph_files["assort"], 75, 4)
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 4, reportfreq=rfreq)
    eng = s.engine(reportfreq=rfreq)
    n = 0
    while ref.sweep() != 2:
        n += 1
        assert n < 3000
    eng.sweep(n + 1 + 4)
    c = eng.control()
    assert c.stopped == 1 and c.iter == ref.iter and c.sweeps_done == n + 1
    g, lam, conv = eng.state()
    assert np.max(np.abs(g - ref.gamma) / ref.gamma) < 1e-7
    assert np.array_equal(eng.communities(), ref.communities())
    rows = eng.rows()
    assert np.array_equal(rows[:, 0], ref.rows[1:, 0]) and np.all(rows[:, 0] % rfreq == 0)
    np.testing.assert_allclose(rows[:, 1:], ref.rows[1:, 1:], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("graph,n,k,rfreq,options", [
    ("assort", 75, 60, 2, None),             # row-per-wavefront kernels (K > 56): k_tail
    ("lfr", 1000, 64, 3, None),
    ("assort", 75, 4, 2, {"fused3": "0"}),   # small K as four launches per sweep: k_tail with the folded s3
    ("lfr", 1000, 28, 5, {"fused3": "0"}),
])
def test_reportfreq_where_the_tail_is_a_launch_of_its_own(graph_files, graph, n, k, rfreq, options):
    """-rfreq > 1 on the sweeps that end in k_tail: a sweep without a likelihood row runs the serial part (lambda,
    Elogbeta, link statistics, _iter++) on block 0 alone, a sweep with one on the last block behind the ticket --
    the run to its stop, every likelihood row and the final state against the oracle (src/linksampling.cc:768-787)"""
    from svinet_amd.host_api import Setup
    s = Setup(graph_files[graph], n, k)
    ref = O.LinkSampling(O.Network(graph_files[graph], n), k, reportfreq=rfreq)
    eng = s.engine(reportfreq=rfreq, options=options)
    m = 0
    while ref.sweep() != 2:
        m += 1
        assert m < 3000
    eng.sweep(m + 1 + 4)
    c = eng.control()
    assert c.stopped == 1 and c.iter == ref.iter and c.sweeps_done == m + 1
    g, lam, conv = eng.state()
    assert np.max(np.abs(g - ref.gamma) / ref.gamma) < 1e-7
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-7
    assert np.array_equal(eng.communities(), ref.communities())
    rows = eng.rows()
    assert np.array_equal(rows[:, 0], ref.rows[1:, 0]) and np.all(rows[:, 0] % rfreq == 0)
    np.testing.assert_allclose(rows[:, 1:], ref.rows[1:, 1:], rtol=1e-7, atol=1e-12)


def test_empty_and_tiny_graphs():
    """degenerate inputs through the C ABI: no training link at all, and an 8-node ring"""
    from svinet_amd._svils import Engine
    rng = np.random.default_rng(5)
    for k in (4, 40):
        n = 6
        eng = Engine(n, k, ones=0, ones_prob=0.0, use_validation_stop=False)
        eng.set_graph(np.zeros((0, 2), dtype=np.uint32))
        eng.set_validation(np.zeros((0, 3), dtype=np.uint32))
        eng.set_state(rng.uniform(0.5, 2.0, size=(n, k)), np.ones((k, 2)))
        eng.sweep(2)
        g, lam, conv = eng.state()
        assert np.all(g == 1.0 / k)                      # gammanext stays alpha without links (:532-533)
        assert np.all(lam[:, 0] == 1.0) and np.all(lam[:, 1] == 1.0)
        assert eng.control().iter == 2 and eng.control().links_dense == 0
        # the smallest non-degenerate graph: a ring of 8 nodes (with fewer nodes than 2*degree+1 the
        # reference's own non-link correction n - tl - 1 goes negative and it produces NaN)
        pairs = np.array([[i, (i + 1) % 8] for i in range(8)], dtype=np.int32)
        ref = O.LinkSampling(O.Network(n=8, pairs=pairs), k, heldout_ratio=0.0, use_validation_stop=False)
        ref.set_skip_validation(True)
        e2 = Engine(8, k, ones=8, ones_prob=ref.ones_prob, use_validation_stop=False)
        e2.set_graph(ref.links)
        e2.set_validation(np.zeros((0, 3), dtype=np.uint32))
        e2.set_state(ref.gamma, ref.lam)
        for _ in range(3):
            ref.sweep()
        e2.sweep(3)
        g, lam, conv = e2.state()
        assert np.allclose(g, ref.gamma, rtol=1e-10) and np.allclose(lam, ref.lam, rtol=1e-10)


@pytest.mark.parametrize("k,no_elogpi", [(20, False), (64, False), (200, False), (64, True), (200, True), (500, True)])
def test_softmax_rows_that_underflow(k, no_elogpi, monkeypatch):
    """Links whose endpoints have (almost) disjoint supports: every exp(x_k) underflows without a
    shift (x_k < -745 for all k).  The row-per-wavefront kernel computes its softmax without the max
    shift and must fall back to the shifted form for such rows; the lane-per-link kernel always shifts.
    Both must agree with the oracle's sequential log-sum-exp.
    no_elogpi: the handle stores no Elogpi rows (option skip_elogpi, by itself only from 256 MB of state on) -- its fast phi launch
    only raises DevCtrl::phi_redo and the launch behind it redoes the pass with the rows re-derived from gamma."""
    from svinet_amd.host_api import Setup
    if no_elogpi:
        monkeypatch.setenv("SVILS_SKIP_ELOGPI", "1")
    n = 2 * k + 20          # every community keeps some mass (an empty one makes E/sum[k] infinite in the reference too)
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    chords = np.stack([np.arange(n), (np.arange(n) + 7) % n], 1)
    pairs = np.concatenate([ring, chords]).astype(np.int32)
    s = Setup(n=n, k=k, pairs=pairs, heldout_ratio=0.0)
    ref = O.LinkSampling(O.Network(n=n, pairs=pairs), k, heldout_ratio=0.0, use_validation_stop=False)
    ref.set_skip_validation(True)
    # node i is concentrated on community i % k; everything else is tiny, so that psi(tiny) ~ -1/tiny = -2000
    g = np.full((n, k), 5e-4)
    g[np.arange(n), np.arange(n) % k] = 50.0
    lam = np.tile([3.0, 2.0], (k, 1))
    ref.set_gamma(g); ref.set_lambda(lam); ref.refresh()
    x = ref.elogpi[0] + ref.elogpi[1] + ref.elogbeta[:, 0]
    assert x.max() < -745, "the test must exercise the underflow path"
    eng = s.engine(use_validation_stop=False)
    assert eng.get_option("skip_elogpi") == (1 if no_elogpi else -1)
    eng.set_state(g, lam)
    for nsw in (1, 2):
        ref.sweep()
        eng.sweep(1)
        gg, ll, conv = eng.state()
        assert np.isfinite(gg).all() and np.isfinite(ll).all()
        np.testing.assert_allclose(gg, ref.gamma, rtol=1e-7)
        np.testing.assert_allclose(ll, ref.lam, rtol=1e-7)
        assert np.array_equal(conv, ref.converged)


@pytest.mark.parametrize("graph,n,k,sweeps", [("lfr", 1000, 28, 40), ("astroph", 17903, 20, 6), ("lfr", 1000, 8, 12)])
def test_small_k_on_a_graph_too_large_for_the_class_lists(graph_files, graph, n, k, sweeps, monkeypatch):
    """The class lists of the lane-per-link layout pack an entry index into 27 bits; a graph of 2^26 training links or
    more takes the row-per-wavefront kernels at small K too instead of being refused.  SVILS_LPL_MAX_ENTRIES
    lowers the switch-over point so that the fallback runs on the example graphs: against the oracle."""
    from svinet_amd.host_api import Setup
    monkeypatch.setenv("SVILS_LPL_MAX_ENTRIES", "1000")
    setup = Setup(graph_files[graph], n, k)
    eng = setup.engine(use_validation_stop=False)
    monkeypatch.delenv("SVILS_LPL_MAX_ENTRIES")
    eng.sweep(sweeps)
    ref = O.LinkSampling(O.Network(graph_files[graph], n), k, use_validation_stop=False)
    for _ in range(sweeps):
        ref.sweep()
    g, lam, conv = eng.state()
    assert np.max(np.abs(g - ref.gamma) / np.abs(ref.gamma)) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    assert np.array_equal(conv, ref.converged)
    c = eng.control()
    assert (c.links_dense, c.links_sparse, c.links_shortcut) == ref.link_counts()
    assert np.array_equal(eng.communities(), ref.communities())
    np.testing.assert_allclose(eng.rows()[:, 1:], ref.rows[1:, 1:], rtol=1e-8, atol=1e-11)
    # ... and it really was the other layout: the same run on the lane-per-link kernels differs in the last bits
    lpl = setup.engine(use_validation_stop=False)
    lpl.sweep(sweeps)
    assert not np.array_equal(lpl.state()[0], g)
