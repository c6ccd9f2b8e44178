"""-m gpu: `-load` resume parity and the on-disk model contract.

The reference resumes by reading gamma.txt / lambda.txt into _gamma / _lambda in the constructor
(`LinkSampling::load_model`, src/linksampling.cc:1266-1352, path = dir + "gamma.txt" with no
separator, src/env.hh:277-282) and then runs infer() as usual: `_iter` starts at 0 again (quirk
Q1), `_annealing_phase` is true (ctor init list, :33), `_converged` is zeroed at the top of
infer() (:559), the held-out sample is drawn again from the same seed.  Nothing else is restored.
The tests resume both the HIP path and the oracle from the SAME saved model and compare them.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu
SVINET = os.path.join(ROOT, "svinet_amd", "bin", "svinet")


def ref_parse_model(d, n, k):
    """What the reference's parsers make of <d>/gamma.txt and <d>/lambda.txt: fgets() into a
    32*k-byte buffer, strtod() token by token, the first 2 (gamma) / 1 (lambda) tokens of a line
    skipped (MMSBGen::load_model, src/mmsbgen.cc:73-150 -- the -gml consumer; the same loop in
    LinkSampling::load_model, src/linksampling.cc:1266-1352)."""
    sz = 32 * k

    def parse(path, skip, cols, rows):
        out = np.zeros((rows, cols))
        r = 0
        with open(path) as f:
            for line in f:
                # fgets(line, sz, f) would cut a longer line in two: the writer must stay below it
                assert len(line) < sz, "line of %d bytes does not fit the reference's %d-byte buffer" % (len(line), sz)
                toks = line.split()
                assert len(toks) >= skip + cols - 1, "error parsing gamma file"   # the reference's own check
                vals = [float(t) for t in toks]                                   # strtod
                out[r, :len(vals) - skip] = vals[skip:skip + cols]
                r += 1
        assert r == rows
        return out

    return parse(os.path.join(d, "gamma.txt"), 2, k, n), parse(os.path.join(d, "lambda.txt"), 1, 2, k)


def _as_printed(a):
    """values as "%.5f" prints them and strtod reads them back (save_model, src/linksampling.cc:804-837)"""
    return np.array([float("%.5f" % v) for v in a.ravel()]).reshape(a.shape)


@pytest.mark.parametrize("key,n,k,first,more", [("lfr", 1000, 28, 45, 9), ("assort", 75, 4, 6, 12)])
def test_resume_equals_oracle_resumed_from_the_same_model(graph_files, key, n, k, first, more):
    from svinet_amd.host_api import Setup
    s = Setup(graph_files[key], n, k)
    e1 = s.engine(use_validation_stop=False)
    e1.sweep(first)
    g, lam, conv1 = e1.state()
    if key == "lfr":
        assert (conv1 > 0).any()          # the saved run had converged nodes: they must NOT carry over
    g5, l5 = _as_printed(g), _as_printed(lam)

    # HIP path: a fresh handle loaded with the saved model, exactly what the CLI does for -load
    from svinet_amd._svils import Engine
    e2 = Engine(s.n, s.k, ones=s.ones, ones_prob=s.ones_prob, eta=s.eta, use_validation_stop=False)
    e2.set_graph(s.links)
    e2.set_validation(s.validation_sorted)
    e2.set_state(g5, l5)
    c = e2.control()
    assert c.iter == 0 and c.annealing == 1 and c.write_comm == 0 and c.nh == 0     # :33, Q1
    assert not e2.state()[2].any()                                                  # :559
    row0 = e2.validation_row()

    # oracle: same graph, same held-out sample, the same model poked in, expectations refreshed
    ref = O.LinkSampling(O.Network(graph_files[key], n), k, use_validation_stop=False)
    assert np.array_equal(ref.validation_sorted, s.validation_sorted)
    ref.set_gamma(g5)
    ref.set_lambda(l5)
    ref.refresh()
    assert ref.iter == 0 and ref.annealing and not ref.converged.any()
    for _ in range(more):
        ref.sweep()
    e2.sweep(more)
    g2, l2, conv2 = e2.state()
    assert np.max(np.abs(g2 - ref.gamma) / ref.gamma) < 1e-9
    assert np.max(np.abs(l2 - ref.lam) / np.abs(ref.lam)) < 1e-9
    assert np.array_equal(conv2, ref.converged)
    assert np.array_equal(e2.communities(), ref.communities())
    c = e2.control()
    assert c.iter == more == ref.iter and bool(c.annealing) == ref.annealing
    rows = e2.rows()
    assert rows.shape[0] == more and list(rows[:, 0]) == list(range(more))          # iterations count from 0 again
    np.testing.assert_allclose(rows[:, 1:], ref.rows[1:, 1:], rtol=1e-9, atol=1e-13)
    # the constructor-time row of a resumed run is the likelihood of the LOADED model at iteration 0
    assert row0[0] == 0 and np.isfinite(row0).all()
    # ... and differs from a run that was never interrupted (annealing restarted, flags dropped)
    if (conv1 > 0).any():
        e1.sweep(more)
        assert np.max(np.abs(e1.state()[0] - g2) / g2) > 1e-6


def _run(args, cwd):
    return subprocess.run([SVINET] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)


def test_cli_load_round_trip_and_parser_contract(graph_files, tmp_path):
    """save_model -> the reference's parser -> -load -> N more sweeps, on files, against the oracle"""
    common = ["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop"]
    r = _run(common + ["-max-iterations", "12", "-label", "first"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d1 = tmp_path / "n1000-k28-first-linksampling"
    # contract with MMSBGen::load_model (-gml) and LinkSampling::load_model: parse what we wrote
    g5, l5 = ref_parse_model(str(d1), 1000, 28)
    assert np.isfinite(g5).all() and (g5 > 0).all() and (l5 > 0).all()
    first_cols = np.loadtxt(d1 / "gamma.txt")[:, :2]
    assert np.array_equal(first_cols[:, 0], np.arange(1000))                        # seq, then the external id
    # the saved numbers are the state itself at print resolution
    ref1 = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, use_validation_stop=False, max_iterations=12)
    while ref1.sweep() == 0:
        pass
    np.testing.assert_allclose(g5, ref1.gamma, rtol=1e-5, atol=6e-6)
    np.testing.assert_allclose(l5, ref1.lam, rtol=1e-5, atol=6e-6)

    # resume: path is dir + "gamma.txt" -- the trailing separator is the caller's job (src/env.hh:277-282)
    r = _run(common + ["-max-iterations", "7", "-label", "resumed", "-load", str(d1) + "/"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d2 = tmp_path / "n1000-k28-resumed-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, use_validation_stop=False, max_iterations=7)
    ref.set_gamma(g5)
    ref.set_lambda(l5)
    ref.refresh()
    n = 0
    while ref.sweep() == 0:
        n += 1
    assert n == 8                                                                   # Q8: N + 1 sweeps, counted from 0
    rd = tmp_path / "ref_resumed"
    ref.write_model(str(rd))
    a, b = np.loadtxt(d2 / "gamma.txt"), np.loadtxt(rd / "gamma.txt")
    assert np.array_equal(a[:, :2], b[:, :2])
    np.testing.assert_allclose(a[:, 2:], b[:, 2:], rtol=1e-5, atol=1.1e-5)
    np.testing.assert_allclose(np.loadtxt(d2 / "lambda.txt"), np.loadtxt(rd / "lambda.txt"), rtol=1e-5, atol=1.1e-5)
    assert (d2 / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v = np.loadtxt(d2 / "validation.txt")
    assert v.shape == (9, 11) and list(v[:, 0].astype(int)) == [0] + list(range(8))  # ctor row + sweeps 0..7
    np.testing.assert_allclose(np.delete(v, 1, axis=1)[1:], ref.rows[1:], rtol=0, atol=6e-10)
    # without the separator the files are not found, loudly (no silent fresh start)
    r = _run(common + ["-max-iterations", "2", "-label", "nosep", "-load", str(d1)], str(tmp_path))
    assert r.returncode != 0 and "gamma.txt" in (r.stderr + r.stdout)
