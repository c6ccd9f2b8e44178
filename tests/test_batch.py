"""`-batch` (BASELINE config 1): the reference's all-pairs CPU engine, plumbing only.

* samplers pinned by the authors' shipped run (tests/golden/ref_assort_batch, data files of
  example/n75-k4-mmsb-batch.tgz; that revision printed sequence ids and used heldout ratio 0.1);
* sweep / likelihood: host C++ engine against the numpy restatement oracle/batch_oracle.py from
  the engine's own starting point (gsl_ran_gamma's stream is internal to GSL: parity unpinned);
* CLI: runs to the stop rule, writes the reference's files, finds the 4 planted blocks.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import batch_oracle as B
from svinet_amd.host_api import BatchEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SVINET = os.path.join(ROOT, "svinet_amd", "bin", "svinet")


def _pairs(path):
    return np.array([[int(x) for x in l.split()] for l in open(path) if l.strip()], dtype=np.uint32)


def test_samplers_match_shipped_run(graph_files):
    e = BatchEngine(graph_files["assort"], 75, 4, heldout_ratio=0.1)
    d = os.path.join(GOLDEN, "ref_assort_batch")
    assert np.array_equal(e.heldout, _pairs(os.path.join(d, "heldout-edges.txt")))
    assert np.array_equal(e.validation, _pairs(os.path.join(d, "validation-edges.txt")))
    assert e.heldout.shape[0] == 84 and len({tuple(r) for r in e.heldout}) == 83   # param.txt: 83 distinct


def test_sweep_matches_restatement(graph_files):
    e = BatchEngine(graph_files["assort"], 75, 4, heldout_ratio=0.1, eta_type="fromdata")
    assert e.eta == pytest.approx((214.75, 1.0))                       # shipped param.txt
    n, k = e.n, e.k
    adj = np.zeros((n, n), dtype=np.int64)
    adj[e.edges[:, 0], e.edges[:, 1]] = adj[e.edges[:, 1], e.edges[:, 0]] = 1
    skip = {tuple(r) for r in e.heldout} | {tuple(r) for r in e.validation}
    hsorted = sorted({tuple(int(x) for x in r) for r in e.heldout})
    g, lam = e.gamma, e.lam
    assert 0.6 < g.min() and g.max() < 1.5 and abs(g.mean() - 1) < 0.01    # Gamma(100, 1/100) cells
    np.testing.assert_allclose(e.rows[0][1:], B.heldout_row(g, lam, hsorted, adj, e.ones_prob), rtol=1e-12)
    # tolerance: the per-pair fixed point stops on a 1e-5 mean-change threshold, so a last-bit
    # difference in psi() can move one pair's exit by a round (observed 2e-10 relative on gamma)
    for it in range(1, 4):
        g, lam = B.sweep(g, lam, adj, skip, 1.0 / k, e.eta)
        e.sweep()
        assert not e.report()
        np.testing.assert_allclose(e.gamma, g, rtol=1e-7)
        np.testing.assert_allclose(e.lam, lam, rtol=1e-7)
        assert e.rows[it][0] == it
        np.testing.assert_allclose(e.rows[it][1:], B.heldout_row(g, lam, hsorted, adj, e.ones_prob), rtol=1e-7)
    # invariants of the update: every trained pair adds one unit of mass to each endpoint
    trained = n * (n - 1) // 2 - len(skip)
    assert abs((e.gamma - 1.0 / k).sum() - 2 * trained) < 1e-8


def test_cli_batch_runs_to_stop_and_finds_blocks(graph_files, tmp_path):
    r = subprocess.run([SVINET, "-file", graph_files["assort"], "-n", "75", "-k", "4", "-batch",
                        "-eta-type", "fromdata", "-heldout-ratio", "0.1", "-outdir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n75-k4-mmsb-batch"
    for f in ("param.txt", "heldout-edges.txt", "validation-edges.txt", "heldout.txt", "validation.txt", "max.txt",
              "gamma.txt", "lambda.txt", "groups.txt", "communities.txt", "summary.txt"):
        assert (d / f).exists(), f
    it, _, a, _, max_h, _, why = (d / "max.txt").read_text().split()
    assert int(it) > 75 and int(why) in (0, 1)                          # stop rule is armed after n sweeps
    rows = np.loadtxt(d / "heldout.txt")
    assert rows.shape == (int(it) + 1, 11) and rows[-1, 10] > rows[0, 10]
    gam = np.loadtxt(d / "gamma.txt")
    assert gam.shape == (75, 6) and np.all(gam[:, 2:] > 0)
    lam = np.loadtxt(d / "lambda.txt")
    assert lam.shape == (4, 3)
    groups = np.loadtxt(d / "groups.txt")
    assert groups.shape == (75, 7) and np.allclose(groups[:, 2:6].sum(1), 1, atol=2e-3)
    label = dict(zip(groups[:, 1].astype(int), groups[:, 6].astype(int)))
    # the generator's four blocks are contiguous id ranges (example/assort-75-4-results.png)
    found = []
    for lo, hi in ((2, 21), (24, 44), (48, 66), (67, 75)):
        ls = [label[i] for i in range(lo, hi + 1)]
        top = max(set(ls), key=ls.count)
        assert ls.count(top) >= 0.85 * len(ls)
        found.append(top)
    assert len(set(found)) == 4
    comm = [l.split() for l in (d / "communities.txt").read_text().splitlines() if l.strip()]
    assert 3 <= len(comm) <= 4 and all(len(set(c)) == len(c) for c in comm)


def test_cli_rejects_empty_graph(tmp_path):
    p = tmp_path / "empty.txt"
    p.write_text("")
    r = subprocess.run([SVINET, "-file", str(p), "-n", "10", "-k", "2", "-batch", "-outdir", str(tmp_path)],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no links" in r.stderr
