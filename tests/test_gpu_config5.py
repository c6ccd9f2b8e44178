"""-m gpu: BASELINE config 5 (synthetic MMSB, n = 1,000,000, k = 512) at FULL size on one GPU, and an
HBM-bound parity point against the oracle.

At n = 10^6 the oracle needs ~155 s per sweep: too slow to run beside the test, so the full-size parity check
compares with a DIGEST of the oracle's state that tools/make_config5_digest.py produced in the build container
(tests/golden/config5/).  The full-size run is also checked through the size-independent properties of the sweep
(tests/test_gpu_properties.py), and at n = 10^5, k = 512 (0.41 GB per n-by-k array, outside the 256 MB Infinity Cache,
the same kernels and layouts) against the oracle running live.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def config5():
    """the full-size graph, its planted truth and the host side (LinkSampling ls(env, network): held-out set, seeded
    gamma -- ~25 s of host work, 4.1 GB) ONCE for the three full-size tests; every test builds its own engines from it"""
    from svinet_amd import mmsbgen_sparse as G
    from svinet_amd.host_api import Setup
    n, k = 1_000_000, 512
    pairs, truth = G.generate(n, k, 24, return_truth=True)
    # host_gamma=False: the seeded gamma of every engine below is DRAWN ON THE DEVICE (svils_init_gamma: whole rows for the plain
    # engines, column slices for the eight K-shards) -- so the oracle digests also hold the device form of init_gamma2 at full size
    s = Setup(n=n, k=k, pairs=pairs, host_gamma=False)
    yield n, k, pairs, truth, s
    s.close()


def test_config5_full_size_invariants(config5):
    n, k, pairs, _, s = config5
    assert s.n == n and s.singles == 0                     # no isolated node, so -n 1000000 holds (Q9)
    L = int(s.nlinks)
    assert 1.0e7 < L < 6.0e7                               # below the reference's 6e7-link cap (Q6)
    # quirk Q5: n(n-1)/2 in uint32 arithmetic wraps at this size; the host reproduces the wrapped value
    assert s.total_pairs == float((np.uint64(n) * np.uint64(n - 1) % np.uint64(2**32)) // np.uint64(2))
    e1 = s.engine(use_validation_stop=False)
    e2 = s.engine(use_validation_stop=False)
    e1.sweep(2)
    e2.sweep(2)
    g1, l1, c1 = e1.state()
    g2, l2, c2 = e2.state()
    # bitwise run-to-run determinism (no floating-point atomics anywhere)
    assert np.array_equal(l1, l2) and np.array_equal(c1, c2) and np.array_equal(g1, g2)
    del g2
    assert np.isfinite(g1).all() and (g1 > 0).all()
    # phi rows are probability vectors: every link adds 2 to `_sum` (src/linksampling.cc:625,630,663,700)
    tot = float((l1[:, 0] - s.eta[0]).sum())
    assert abs(tot - 2.0 * L) < 1e-6 * L, (tot, 2 * L)
    # mean indicators: sum_k mphi[p][k] == 1/2 for every node with a training link (tl = 2 deg, quirk Q3)
    rs = e1.aux(2).sum(1)
    tl = e1.aux(4)
    assert np.allclose(rs[tl > 0], 0.5, rtol=0, atol=1e-12)
    deg = np.bincount(s.links.ravel(), minlength=s.n)
    assert np.array_equal(tl, 2.0 * deg)
    # the link-branch counters partition the training links
    c = e1.control()
    assert c.links_dense + c.links_sparse + c.links_shortcut == L and c.sweeps_done == 2 and c.rows == 2
    rows = e1.rows()
    assert np.isfinite(rows).all() and list(rows[:, 0]) == [0.0, 1.0] and rows[0, 2] == s.validation_sorted.shape[0]
    assert np.array_equal(rows, e2.rows())
    e1.close()
    e2.close()


def test_config5_full_size_against_oracle_digest(config5):
    """BASELINE config 5 at FULL size (n = 1,000,000, k = 512) against the ORACLE: tools/make_config5_digest.py ran the
    sequential oracle for 2 sweeps on this graph in the build container (~2.5 min per sweep, 25 GB -- too slow to repeat
    on the GPU box) and committed a digest of its state (tests/golden/config5/); the HIP run must reproduce it --
    lambda, the column sums of gamma and 64 fixed gamma rows to 1e-9 relative, every integer exactly
    (src/linksampling.cc:605-761)."""
    import hashlib
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5")
    meta = json.load(open(os.path.join(d, "digest.json")))
    dg = np.load(os.path.join(d, "digest.npz"))
    n, k, _, _, s = config5
    assert (n, k) == (meta["n"], meta["k"]) and meta["mean_degree"] == 24
    # the same graph, the same held-out set, the same wrapped total_pairs (quirk Q5) as the oracle saw
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert int(s.nlinks) == meta["nlinks"] and sha(s.links.astype(np.uint32)) == meta["links_sha256"]
    assert sha(s.validation_sorted.astype(np.uint32)) == meta["validation_sha256"]
    assert s.total_pairs == meta["total_pairs_uint32"] and s.ones_prob == meta["ones_prob"]
    eng = s.engine(use_validation_stop=False)
    row0 = eng.validation_row()
    eng.sweep(meta["sweeps"])
    g, lam, conv = eng.state()
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    assert rel(lam, dg["lam"]) < 1e-9
    assert rel(g.sum(0), dg["gamma_colsum"]) < 1e-9
    assert rel(g[dg["rows_idx"]], dg["gamma_rows"]) < 1e-9
    rs = g.sum(1)
    np.testing.assert_allclose([rs.min(), rs.max()], dg["gamma_rowsum_minmax"], rtol=1e-9)
    # integers: exactly
    assert np.array_equal(np.flatnonzero(conv).astype(np.uint32), dg["converged_idx"])
    assert np.array_equal(conv[conv > 0], dg["converged_val"])
    assert np.array_equal(np.bincount(eng.aux(3), minlength=k + 1), dg["active_hist"])
    st = eng.sweep_stats(0, meta["sweeps"])
    assert np.array_equal(st.astype(np.int64), dg["link_counts"])
    # likelihood rows: the constructor's and one per sweep (columns: iter, s/k, k, mean0, k0, mean1, k1, ...)
    want = dg["likelihood_rows"]
    np.testing.assert_allclose(row0[1:], want[0, 1:], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(eng.rows()[:, 1:], want[1:, 1:], rtol=1e-9, atol=1e-13)
    assert list(eng.rows()[:, 0]) == list(want[1:, 0])
    eng.close()


def _eight_kshards(s, world=8, state=None, control=None):
    """BASELINE config 5 as its 8-GPU run executes it: EIGHT K-sharded handles of 64 columns each (k_phi_ksh16,
    k_fin1/2_ksh<16,4>, k_s3_ksh16), here as virtual ranks on the one GPU with the four exchanges summed in rank order"""
    from svinet_amd.ksharded import KShard, init_virtual
    shards = [KShard(s, r, world, 0, use_validation_stop=False) for r in range(world)]
    if state is not None:
        g0, lam0, conv0 = state
        for sh in shards:
            sh.engine.set_state(np.ascontiguousarray(g0[:, sh.k0:sh.k1]), np.ascontiguousarray(lam0[sh.k0:sh.k1]), conv0)
    if control is not None:
        for sh in shards:
            sh.engine.set_control(**control)
    init_virtual(shards)
    return shards


def _kshard_state(shards):
    states = [sh.engine.state() for sh in shards]
    for st in states[1:]:
        assert np.array_equal(st[2], states[0][2])       # flags are replicated
    return np.concatenate([st[0] for st in states], 1), np.concatenate([st[1] for st in states], 0), states[0][2]


def test_config5_full_size_eight_kshards_against_oracle_digest(config5):
    """BASELINE config 5 at FULL size in the layout built for its 8-GPU run -- eight column slices of 64 -- against the
    same oracle digest as the plain engine (two sweeps from the seeded state): lambda, gamma column sums, 64 gamma rows
    to 1e-9, flags / active counts / link-branch counts of every sweep exactly, likelihood rows to 1e-9
    (src/linksampling.cc:605-761; the layout: DESIGN.md section 6)."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5")
    meta = json.load(open(os.path.join(d, "digest.json")))
    dg = np.load(os.path.join(d, "digest.npz"))
    n, k, _, _, s = config5
    from svinet_amd.ksharded import sweep_virtual
    shards = _eight_kshards(s)
    assert [sh.k1 - sh.k0 for sh in shards] == [64] * 8 and not shards[0].log_domain
    sweep_virtual(shards, meta["sweeps"])
    g, lam, conv = _kshard_state(shards)
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    assert rel(lam, dg["lam"]) < 1e-9
    assert rel(g.sum(0), dg["gamma_colsum"]) < 1e-9
    assert rel(g[dg["rows_idx"]], dg["gamma_rows"]) < 1e-9
    rs = g.sum(1)
    np.testing.assert_allclose([rs.min(), rs.max()], dg["gamma_rowsum_minmax"], rtol=1e-9)
    del g
    assert np.array_equal(np.flatnonzero(conv).astype(np.uint32), dg["converged_idx"])
    assert np.array_equal(conv[conv > 0], dg["converged_val"])
    want = dg["likelihood_rows"]
    for sh in (shards[0], shards[7]):                    # replicated on every rank: first and last
        e = sh.engine
        assert np.array_equal(np.bincount(e.aux(3), minlength=k + 1), dg["active_hist"])
        assert np.array_equal(e.sweep_stats(0, meta["sweeps"]).astype(np.int64), dg["link_counts"])
        np.testing.assert_allclose(e.rows()[:, 1:], want[1:, 1:], rtol=1e-9, atol=1e-13)
        assert list(e.rows()[:, 0]) == list(want[1:, 0])
    for sh in shards:
        sh.engine.close()


def test_config5_full_size_planted_state_eight_kshards_against_oracle_digest(config5):
    """... and from the planted near-converged state at _iter = 999 (four sweeps: O(1) shortcuts for 44 % of the links, s3's
    quirk Q2 across slice edges -- q2v --, the active-set branch past _iter = 1000 on bitmasks of one word per node):
    link-branch counts of every sweep and every flag exactly, lambda / gamma to 1e-9."""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, "golden", "config5")
    meta = json.load(open(os.path.join(d, "digest_planted.json")))
    dg = np.load(os.path.join(d, "digest_planted.npz"))
    spec = importlib.util.spec_from_file_location("make_config5_digest", os.path.join(os.path.dirname(here), "tools", "make_config5_digest.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    n, k, pairs, truth, s = config5
    from svinet_amd.ksharded import sweep_virtual
    state = tool.planted_state(pairs, truth, n, k)
    shards = _eight_kshards(s, state=state, control=dict(iter=meta["iter0"], annealing=0))
    del state
    nsw = meta["sweeps"]
    sweep_virtual(shards, nsw)
    g, lam, conv = _kshard_state(shards)
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    assert rel(lam, dg["lam"]) < 1e-9
    assert rel(g.sum(0), dg["gamma_colsum"]) < 1e-9
    assert rel(g[dg["rows_idx"]], dg["gamma_rows"]) < 1e-9
    del g
    assert np.array_equal(np.flatnonzero(conv).astype(np.uint32), dg["converged_idx"]) and dg["converged_idx"].size > 100_000
    assert np.array_equal(conv[conv > 0], dg["converged_val"])
    want = dg["likelihood_rows"]
    for sh in (shards[0], shards[7]):
        e = sh.engine
        st = e.sweep_stats(0, nsw).astype(np.int64)
        assert np.array_equal(st, dg["link_counts"]), (st, dg["link_counts"])
        assert np.array_equal(np.bincount(e.aux(3), minlength=k + 1), dg["active_hist"])
        np.testing.assert_allclose(e.rows()[:, 1:], want[1:, 1:], rtol=1e-9, atol=1e-10)   # (atol: see the plain-engine test)
    for sh in shards:
        sh.engine.close()


def test_config5_full_size_planted_state_against_oracle_digest(config5):
    """The same full-size graph in the regime a long run ends in, which two sweeps from the seeded state never reach:
    started from tools/make_config5_digest.py::planted_state (gamma = alpha + degree x planted membership, every third node
    pure and flagged converged) at _iter = 999 with annealing off, the oracle ran four sweeps -- two dense ones with O(1)
    shortcuts for 44 % of the links and the s3 pass on those flags (quirk Q2), then two past _iter = 1000 that also take the
    active-set branch (src/linksampling.cc:622-681,731-746) -- and its digest is committed; the HIP run must match: link-branch counts of
    every sweep and every flag exactly, lambda / gamma to 1e-9."""
    import hashlib
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, "golden", "config5")
    meta = json.load(open(os.path.join(d, "digest_planted.json")))
    dg = np.load(os.path.join(d, "digest_planted.npz"))
    spec = importlib.util.spec_from_file_location("make_config5_digest", os.path.join(os.path.dirname(here), "tools", "make_config5_digest.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    n, k, pairs, truth, s = config5
    assert (n, k) == (meta["n"], meta["k"]) and meta["mean_degree"] == 24
    g0, lam0, conv0 = tool.planted_state(pairs, truth, n, k)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(g0) == meta["gamma0_sha256"]
    assert sha(s.links.astype(np.uint32)) == meta["links_sha256"] and sha(s.validation_sorted.astype(np.uint32)) == meta["validation_sha256"]
    eng = s.engine(use_validation_stop=False)
    eng.set_state(g0, lam0, conv0)
    del g0
    eng.set_control(iter=meta["iter0"], annealing=0)
    nsw = meta["sweeps"]
    eng.sweep(nsw)
    st = eng.sweep_stats(0, nsw).astype(np.int64)
    assert np.array_equal(st, dg["link_counts"]), (st, dg["link_counts"])
    assert st[:, 2].min() > 1_000_000 and st[:, 1].max() > 100_000        # shortcut links in every sweep, the active-set branch past _iter = 1000
    g, lam, conv = eng.state()
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    assert rel(lam, dg["lam"]) < 1e-9
    assert rel(g.sum(0), dg["gamma_colsum"]) < 1e-9
    assert rel(g[dg["rows_idx"]], dg["gamma_rows"]) < 1e-9
    assert np.array_equal(np.flatnonzero(conv).astype(np.uint32), dg["converged_idx"]) and dg["converged_idx"].size > 100_000
    assert np.array_equal(conv[conv > 0], dg["converged_val"])
    assert np.array_equal(np.bincount(eng.aux(3), minlength=k + 1), dg["active_hist"])
    want = dg["likelihood_rows"]                        # row 0 is the seeded constructor's (before the planted state went in)
    # Absolute 1e-10 on the likelihood columns.  In this state 508 of a node's 512 pi components are EXACTLY equal (alpha / row sum),
    # so the reference's K^2 double loop (src/linksampling.hh:258-292, restated by the oracle) adds the same tiny product
    # ~2.6e5 times to a partial sum near 1: every one of those additions rounds the same way, and the loop's sum comes out
    # ~2e-12 low per non-link pair (measured: mean0 of every row -1.9e-12 with lambda / gamma equal to 1e-13; a numpy
    # restatement of the loop on such rows shows -7e-12 .. +2e-12 per pair).  The device evaluates the loop's exact
    # value, (sum pi_p)(sum pi_q) - sum pi_p pi_q beta; validation.txt prints these columns to 1e-9.
    np.testing.assert_allclose(eng.rows()[:, 1:], want[1:, 1:], rtol=1e-9, atol=1e-10)
    assert list(eng.rows()[:, 0]) == list(want[1:, 0])
    eng.close()


@pytest.fixture(scope="module")
def hbm_bound():
    """n = 1e5, k = 512 on the planted MMSB graph (0.41 GB per n-by-k array: outside the 256 MB Infinity Cache), the oracle
    run LIVE for three sweeps -- once, for the plain engine and for the eight node blocks below (~50 s of CPU)."""
    from oracle import oracle as O
    from svinet_amd import mmsbgen_sparse as G
    from svinet_amd.host_api import Setup
    n, k, nsweeps = 100_000, 512, 3     # flags set by prune() in one sweep steer the branches of the next (shortcut links, s3's quirk Q2)
    pairs = G.generate(n, k, 24)
    s = Setup(n=n, k=k, pairs=pairs)
    ref = O.LinkSampling(O.Network(n=n, pairs=pairs), k, use_validation_stop=False)
    assert np.array_equal(ref.links, s.links) and np.array_equal(ref.validation_sorted, s.validation_sorted)
    for _ in range(nsweeps):
        ref.sweep()
    yield n, k, nsweeps, pairs, s, ref
    s.close()


def test_hbm_bound_sweep_against_oracle(hbm_bound):
    """three sweeps at n = 1e5, k = 512 on the planted MMSB graph against the oracle RUNNING LIVE (the full-size runs
    above compare with committed digests): gamma / lambda within 1e-5 relative (the north-star bar; observed ~1e-13),
    flags, counters and the likelihood row equal.
    (n = 2e5 until round 4: 107 s of the suite for the oracle's three sweeps, now that the full size itself is covered.)"""
    n, k, nsweeps, pairs, s, ref = hbm_bound
    eng = s.engine(use_validation_stop=False)
    eng.sweep(nsweeps)
    g, lam, conv = eng.state()
    rg, rl = ref.gamma, ref.lam
    assert np.max(np.abs(g - rg) / rg) < 1e-5
    assert np.max(np.abs(lam - rl) / np.abs(rl)) < 1e-5
    assert np.max(np.abs(g - rg) / rg) < 1e-9            # what fp64 with re-ordered sums actually gives
    assert np.array_equal(conv, ref.converged)
    assert np.array_equal(eng.aux(3), ref.active_comms)
    c = eng.control()
    assert (c.links_dense, c.links_sparse, c.links_shortcut) == ref.link_counts()
    np.testing.assert_allclose(eng.rows()[:, 1:], ref.rows[1:, 1:], rtol=1e-9, atol=1e-13)
    eng.close()


def test_hbm_bound_eight_node_blocks_chunked_exchange_against_oracle(hbm_bound, tmp_path):
    """The node-block form of an HBM-bound K = 512 run as eight GPUs execute it: EIGHT native ranks (processes on this one
    GPU over the tests' transport) on work-balanced blocks, the row exchange FORCED into its pipelined form (option xchunks:
    chunks of every block on the communication stream and a second communicator, each expanded while the next travels --
    the form config 5's 4.1 GB payload takes by itself) -- every rank's replicated state against the live oracle."""
    import test_gpu_native_ranks as T
    n, k, nsweeps, pairs, s, ref = hbm_bound
    path = str(tmp_path / "mmsb_n100k.txt")
    with open(path, "w") as f:
        f.write("".join("%d\t%d\n" % (a, b) for a, b in np.asarray(pairs).tolist()))
    world, chunks = 8, 3
    states, (calls, ncomm) = T._run_ranks(tmp_path, path, n, k, nsweeps, world, "sweep", {"SVILS_XCHUNKS": str(chunks)})
    for st in states:
        assert np.max(np.abs(st["gamma"] - ref.gamma) / ref.gamma) < 1e-9
        assert np.max(np.abs(st["lam"] - ref.lam) / np.abs(ref.lam)) < 1e-9
        assert np.array_equal(st["conv"], ref.converged)
        np.testing.assert_allclose(st["rows"][:, 1:], ref.rows[1:, 1:], rtol=1e-9, atol=1e-13)
        assert bool(st["row_comm"]) and int(st["comm_nranks"]) == world
    for st in states[1:]:
        assert np.array_equal(st["gamma"], states[0]["gamma"]) and np.array_equal(st["lam"], states[0]["lam"])
    assert np.array_equal(states[0]["member"], ref.communities())
    # per sweep: all-reduce(sum) + chunks x world broadcasts + all-reduce(s1,s2,s3); + the tag gather and the second communicator's id
    assert ncomm == 2 and calls == (2 + chunks * world) * nsweeps + world + 1


@pytest.mark.parametrize("k", [20, 28])
def test_large_small_k_graph_against_oracle(k):
    """K <= 32 on a graph of more than 512 classification tiles (n = 60 000, ~1.4 M CSR entries): full sweeps
    keep four launches there, with the next sweep's classification riding spin-free on the s3 and tail
    launches in as many blocks as it needs.  Twelve sweeps (shortcut links appear, the link classes change from
    sweep to sweep) against the oracle: state, flags, per-sweep link counts and likelihood rows."""
    from oracle import oracle as O
    from svinet_amd import mmsbgen_sparse as G
    from svinet_amd.host_api import Setup
    n = 60_000
    pairs = G.generate(n, k, 24)
    s = Setup(n=n, k=k, pairs=pairs)
    assert 2 * int(s.nlinks) > 512 * 1024
    ref = O.LinkSampling(O.Network(n=n, pairs=pairs), k, use_validation_stop=False)
    eng = s.engine(use_validation_stop=False)
    counts = []
    for _ in range(12):
        ref.sweep()
        counts.append(ref.link_counts())
    eng.sweep(12)
    g, lam, conv = eng.state()
    assert np.max(np.abs(g - ref.gamma) / ref.gamma) < 1e-9
    assert np.max(np.abs(lam - ref.lam) / np.abs(ref.lam)) < 1e-9
    assert np.array_equal(conv, ref.converged)
    st = eng.sweep_stats(0, 12)
    assert [tuple(int(x) for x in r[:3]) for r in st] == counts
    np.testing.assert_allclose(eng.rows()[:, 1:], np.asarray(ref.rows)[1:13, 1:], rtol=1e-9, atol=1e-13)
    # the same through eager phase-by-phase launches of a second engine (bitwise: one code path per kernel)
    e2 = s.engine(use_validation_stop=False)
    for _ in range(12):
        e2.sweep(1)
    g2, l2, c2 = e2.state()
    assert np.array_equal(g, g2) and np.array_equal(lam, l2) and np.array_equal(conv, c2)
