"""-m gpu: the `svinet` command line end to end against the oracle's writers
(gamma.txt / lambda.txt / communities.txt / groups.txt / validation.txt formats of
src/linksampling.cc:804-917,996-1001,1452-1476)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
SVINET = os.path.join(ROOT, "svinet_amd", "bin", "svinet")


def _run(args, cwd, env=None):
    return subprocess.run([SVINET] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                          env=dict(os.environ, **(env or {})))


def _cmp_numeric(path_a, path_b, skip, atol):
    a, b = np.loadtxt(path_a), np.loadtxt(path_b)
    assert a.shape == b.shape
    assert np.array_equal(a[:, :skip], b[:, :skip])
    np.testing.assert_allclose(a[:, skip:], b[:, skip:], rtol=1e-5, atol=atol)


@pytest.mark.parametrize("batch,sync", [(0, False), (1, False), (7, False), (1, True), (7, True)])
def test_cli_max_iterations(graph_files, tmp_path, batch, sync):
    """batch 0 = the default (automatic chunks); sync = SVINET_SYNC_REPORTS=1, the loop that synchronises at every
    batch instead of collecting report snapshots while the device sweeps on -- the files are the same"""
    tfile = tmp_path / "timing.json"
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop",
              "-max-iterations", "20"] + (["-sweep-batch", str(batch)] if batch else []), str(tmp_path),
             env={"SVINET_SYNC_REPORTS": "1" if sync else "0", "SVINET_TIMING_FILE": str(tfile)})
    assert r.returncode == 0, r.stderr
    import json
    tm = json.loads(tfile.read_text())
    assert tm["pipelined"] == (not sync) and tm["ended_by"] == "max iterations"
    if not sync:
        assert tm["sweeps"] == 21 and tm["reports"] == tm["chunks"] and tm["sweeps_s"] > 0
        assert tm["chunks"] == (21 if batch == 1 else 3 if batch == 7 else tm["chunks"]) and 1 <= tm["communities_written"] <= tm["reports"]
    assert "+ Quitting: reached max iterations." in r.stdout
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, use_validation_stop=False, max_iterations=20)
    n = 0
    while ref.sweep() == 0:
        n += 1
    assert n == 21                                   # quirk Q8: N+1 sweeps
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    _cmp_numeric(d / "groups.txt", rd / "groups.txt", 2, 1.1e-3)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    # validation.txt: ctor row + one row per sweep; columns other than duration equal the oracle's
    v = np.loadtxt(d / "validation.txt")
    assert v.shape == (22, 11)
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)
    assert (d / "validation-edges.txt").read_bytes() == open(os.path.join(GOLDEN, "ref_lfr_k28", "heldout-edges.txt"), "rb").read()
    t = (d / "test.txt").read_text().split("\n")
    assert len([l for l in t if l]) == 21 and "-nan" in t[0]
    mx = (d / "max.txt").read_text().split("\t")
    assert mx[0] == "20" and len(mx) == 6


def test_cli_validation_stop_and_resume(graph_files, tmp_path):
    r = _run(["-file", graph_files["assort"], "-n", "75", "-k", "4", "-link-sampling"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n75-k4-mmsb-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 4)
    while ref.sweep() != 2:
        assert ref.iter < 2000
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v = np.loadtxt(d / "validation.txt")
    assert int(v[-1, 0]) == ref.iter
    # -load: resume from the saved model (path is dir + "gamma.txt", no separator added)
    r2 = _run(["-file", graph_files["assort"], "-n", "75", "-k", "4", "-link-sampling", "-label", "resumed",
               "-load", str(d) + "/", "-no-stop", "-max-iterations", "2"], str(tmp_path))
    assert r2.returncode == 0, r2.stderr
    g0 = np.loadtxt(d / "gamma.txt")
    g1 = np.loadtxt(tmp_path / "n75-k4-resumed-linksampling" / "gamma.txt")
    assert g0.shape == g1.shape and np.isfinite(g1).all()


def test_cli_accuracy_and_load_validation(graph_files, tmp_path):
    # -accuracy: all links train, validation_likelihood() is a no-op => validation.txt stays empty
    r = _run(["-file", graph_files["assort"], "-n", "75", "-k", "4", "-link-sampling", "-accuracy",
              "-max-iterations", "5", "-label", "acc"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n75-k4-acc-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 4, accuracy=True, max_iterations=5)
    while ref.sweep() == 0:
        pass
    rd = tmp_path / "ref_acc"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    assert (d / "validation.txt").read_text() == ""
    # -load-validation: pairs given as external ids, one "id<TAB>id" per line
    net = O.Network(graph_files["assort"], 75)
    s2i = net.seq2id()
    e = net.edges()
    pairs = [(int(s2i[a]), int(s2i[b])) for a, b in e[::97]] + [(int(s2i[0]), int(s2i[70])), (int(s2i[3]), int(s2i[66]))]
    vf = tmp_path / "val.txt"
    vf.write_text("".join("%d\t%d\n" % p for p in pairs))
    r = _run(["-file", graph_files["assort"], "-n", "75", "-k", "4", "-link-sampling", "-load-validation", str(vf),
              "-no-stop", "-max-iterations", "3", "-label", "lv"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n75-k4-lv-linksampling"
    ve = [l.split("\t") for l in (d / "validation-edges.txt").read_text().split("\n") if l]
    assert len(ve) == len(pairs)
    v = np.loadtxt(d / "validation.txt")
    assert v.shape == (5, 11) and int(v[0, 3]) == len(pairs)


def test_cli_max_iterations_one(graph_files, tmp_path):
    """-max-iterations 1 forces write_comm from the first sweep (src/linksampling.cc:581-582)"""
    r = _run(["-file", graph_files["assort"], "-n", "75", "-k", "4", "-link-sampling", "-no-stop",
              "-max-iterations", "1", "-label", "one"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n75-k4-one-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 4, use_validation_stop=False, max_iterations=1)
    n = 0
    while ref.sweep() == 0:
        n += 1
    assert n == 2
    rd = tmp_path / "ref_one"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()


def test_cli_minibatch_mode(graph_files, tmp_path):
    """-minibatch: mini-batch steps through the CLI (random node relabelling on the host side is
    undone in every output file)."""
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-rfreq", "5", "-no-stop",
              "-max-iterations", "199", "-minibatch", "200", "-tau0", "1", "-kappa", "0.5", "-nodetau0", "1",
              "-nodekappa", "0.5", "-sweep-batch", "5"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    gam = np.loadtxt(d / "gamma.txt")
    assert gam.shape == (1000, 30) and np.all(gam[:, 2:] > 0)
    rows = np.loadtxt(d / "validation.txt")
    assert rows[-1, 10] > rows[0, 10]
    assert np.array_equal(rows[1:, 0], np.arange(0, 200, 5))
    assert "link_sampling_minibatch_nodes: 200" in (d / "param.txt").read_text()
    # gamma rows are in sequence-id order again: a node's strongest community agrees with its neighbours' more
    # often than with random nodes'
    edges = np.array([[int(x) for x in l.split()] for l in open(graph_files["lfr"]) if l.strip()])
    ids = gam[:, 1].astype(int)
    top = dict(zip(ids, gam[:, 2:].argmax(1)))
    same = np.mean([top[a] == top[b] for a, b in edges])
    rng = np.random.default_rng(0)
    rnd = np.mean([top[a] == top[b] for a, b in zip(rng.permutation(ids), rng.permutation(ids))])
    assert same > rnd + 0.3


def test_cli_nmi(graph_files, tmp_path):
    """-nmi <ground truth>: ground_truth.txt in the reference's layout and one `mutual3:` line per
    communities.txt written; the LFR communities are recovered (the authors' run ends at 0.897)."""
    truth = os.path.join(GOLDEN, "graphs", "LFR-ground-truth-n1000-k28.txt")
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-nmi", truth], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    # byte-identical to the ground_truth.txt in the authors' shipped run
    assert (d / "ground_truth.txt").read_bytes() == open(os.path.join(GOLDEN, "ref_lfr_k28", "ground_truth.txt"), "rb").read()
    lines = (d / "mutual.txt").read_text().splitlines()
    rows = np.loadtxt(d / "validation.txt")
    assert all(l.startswith("mutual3:\t") for l in lines) and len(lines) == rows.shape[0] - 1   # row 0 is the constructor's
    vals = np.array([float(l.split("\t")[1]) for l in lines])
    assert vals[0] < 0.2 and vals[-1] > 0.8


def test_cli_kshard_world_of_one(graph_files, tmp_path):
    """`svinet -gpus 1 -kshard`: the K-sharded driver of the command line (slice hand-over, svils_ksh_init_state,
    the collective constructor row, svils_sweep_ksharded, the gathers behind communities.txt / gamma.txt /
    lambda.txt) with a world of one -- every file against the oracle's, as for the plain run."""
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop",
              "-max-iterations", "20", "-gpus", "1", "-kshard"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, use_validation_stop=False, max_iterations=20)
    while ref.sweep() == 0:
        pass
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    _cmp_numeric(d / "groups.txt", rd / "groups.txt", 2, 1.1e-3)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v = np.loadtxt(d / "validation.txt")
    assert v.shape == (22, 11)                       # constructor row + one per sweep
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)


def test_cli_kshard_minibatch_world_of_one(graph_files, tmp_path):
    """`svinet -gpus 1 -kshard -minibatch m`: svils_step_ksharded behind the command line (relabelling, slice hand-over in
    the relabelled order, the gathers behind the files); the same files as the plain mini-batch run"""
    args = ["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-rfreq", "5", "-no-stop",
            "-max-iterations", "99", "-minibatch", "200", "-tau0", "1", "-kappa", "0.5", "-nodetau0", "1",
            "-nodekappa", "0.5", "-sweep-batch", "5"]
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ra = _run(args, str(tmp_path / "a"))
    rb = _run(args + ["-gpus", "1", "-kshard"], str(tmp_path / "b"))
    assert ra.returncode == 0 and rb.returncode == 0, ra.stderr + rb.stderr
    da, db = tmp_path / "a" / "n1000-k28-mmsb-linksampling", tmp_path / "b" / "n1000-k28-mmsb-linksampling"
    _cmp_numeric(da / "gamma.txt", db / "gamma.txt", 2, 2.1e-5)
    _cmp_numeric(da / "lambda.txt", db / "lambda.txt", 1, 2.1e-5)
    assert (da / "communities.txt").read_text() == (db / "communities.txt").read_text()
    va, vb = np.loadtxt(da / "validation.txt"), np.loadtxt(db / "validation.txt")
    np.testing.assert_allclose(np.delete(va, 1, axis=1), np.delete(vb, 1, axis=1), rtol=0, atol=2e-9)


def test_cli_kshard_link_thresh_below_one_half(graph_files, tmp_path):
    """`svinet -kshard -link-thresh 0.3`: argmax tagging on the K-sharded layout; communities.txt equals the oracle's"""
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop", "-gpus", "1", "-kshard",
              "-link-thresh", "0.3", "-max-iterations", "20"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, use_validation_stop=False, max_iterations=20, link_thresh=0.3)
    while ref.sweep() == 0:
        pass
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)


def test_cli_sharded_code_path_world_of_one(graph_files, tmp_path):
    """`svinet -sharded`: the -gpus N driver of the command line (node-block handle, communicator, svils_sweep_sharded,
    svils_gather_communities) with one GPU -- the files against the oracle's, as for the plain run."""
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop",
              "-max-iterations", "40", "-sharded", "-sweep-batch", "7"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, use_validation_stop=False, max_iterations=40)
    while ref.sweep() == 0:
        pass
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v = np.loadtxt(d / "validation.txt")
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)


def test_cli_sharded_minibatch_world_of_one(graph_files, tmp_path):
    """`svinet -sharded -minibatch m`: svils_step_sharded behind the command line; the same files as the plain
    mini-batch run (gamma.txt / lambda.txt to the printed digits, the same communities)."""
    args = ["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-rfreq", "5", "-no-stop",
            "-max-iterations", "99", "-minibatch", "200", "-tau0", "1", "-kappa", "0.5", "-nodetau0", "1",
            "-nodekappa", "0.5", "-sweep-batch", "5"]
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ra = _run(args, str(tmp_path / "a"))
    rb = _run(args + ["-sharded"], str(tmp_path / "b"))
    assert ra.returncode == 0 and rb.returncode == 0, ra.stderr + rb.stderr
    da, db = tmp_path / "a" / "n1000-k28-mmsb-linksampling", tmp_path / "b" / "n1000-k28-mmsb-linksampling"
    _cmp_numeric(da / "gamma.txt", db / "gamma.txt", 2, 2.1e-5)
    _cmp_numeric(da / "lambda.txt", db / "lambda.txt", 1, 2.1e-5)
    assert (da / "communities.txt").read_text() == (db / "communities.txt").read_text()
    va, vb = np.loadtxt(da / "validation.txt"), np.loadtxt(db / "validation.txt")
    np.testing.assert_allclose(np.delete(va, 1, axis=1), np.delete(vb, 1, axis=1), rtol=0, atol=2e-9)


@pytest.mark.parametrize("sync", [False, True])
def test_cli_load_test(graph_files, tmp_path, sync):
    """-load-test <file> (external ids, "id<TAB>id"): test-edges.txt lists every line with the network's y, the pairs leave
    the training links, test.txt gets the likelihood of the test set per report -- none for the report that ends the
    run (src/linksampling.cc:777-781,1147-1182,1417-1450).  Against the oracle's counterpart, in both report loops."""
    net = O.Network(graph_files["lfr"], 1000)
    s2i, e = net.seq2id(), net.edges()
    tp = np.concatenate([e[7::97], [[3, 900], [17, 512], [3, 900], [999, 4]]]).astype(np.uint32)
    tf = tmp_path / "test_pairs.txt"
    tf.write_text("".join("%d\t%d\n" % (s2i[a], s2i[b]) for a, b in tp))
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-load-test", str(tf)], str(tmp_path),
             env={"SVINET_SYNC_REPORTS": "1" if sync else "0"})
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    ref = O.LinkSampling(net, 28, test_pairs=tp)
    while ref.sweep() != 2:
        assert ref.iter < 500
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v, t = np.loadtxt(d / "validation.txt"), np.loadtxt(d / "test.txt")
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)
    assert t.shape == (v.shape[0] - 2, 11)               # no constructor row, none for the stopping report
    np.testing.assert_allclose(np.delete(t, 1, axis=1), ref.test_rows, rtol=0, atol=6e-10)
    te = [l.split("\t") for l in (d / "test-edges.txt").read_text().split("\n") if l]
    assert len(te) == tp.shape[0]
    for (a, b), row in zip(tp, te):
        lo, hi = min(a, b), max(a, b)
        assert (int(row[0]), int(row[1]), int(row[2])) == (int(s2i[lo]), int(s2i[hi]), net.y(int(lo), int(hi)))


@pytest.mark.parametrize("blank", [False, True])
def test_cli_init_communities(graph_files, tmp_path, blank):
    """-init-communities <file> (one community per line, external ids; Network::load_init_communities,
    src/network.cc:374-440; LinkSampling::init_gamma_external, src/linksampling.cc:405-453): no random gamma, the model
    starts from the listed memberships.  Files against the oracle's counterpart; init_memberships.txt as the reference
    writes it.  blank: an EMPTY line is not an empty community in the reference -- sscanf("%[^\\n]") matches nothing,
    returns 0 and leaves the previous line in the scratch buffer, whose members become that community as well
    (src/network.cc:391-416)."""
    net = O.Network(graph_files["lfr"], 1000)
    s2i = net.seq2id()
    rng = np.random.default_rng(5)
    comms = [sorted(rng.choice(1000, size=int(rng.integers(20, 60)), replace=False).tolist()) for _ in range(28)]
    comms[5] = comms[5] + comms[5][:3]                   # a node listed twice on one line counts twice
    cf = tmp_path / "init.txt"
    lines = [" ".join(str(int(s2i[p])) for p in c) + "\n" for c in comms]
    if blank:
        lines[10] = "\n"
        comms[10] = list(comms[9])
    cf.write_text("".join(lines))
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-init-communities", str(cf),
              "-no-stop", "-max-iterations", "25"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    ref = O.LinkSampling(net, 28, use_validation_stop=False, max_iterations=25, init_communities=comms)
    while ref.sweep() == 0:
        pass
    rd = tmp_path / "ref"
    ref.write_model(str(rd))
    _cmp_numeric(d / "gamma.txt", rd / "gamma.txt", 2, 1.1e-5)
    _cmp_numeric(d / "lambda.txt", rd / "lambda.txt", 1, 1.1e-5)
    assert (d / "communities.txt").read_text() == (rd / "communities.txt").read_text()
    v = np.loadtxt(d / "validation.txt")
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)
    assert "use_init_communities: True" in (d / "param.txt").read_text()
    mem = [l.rstrip("\n").split("\t") for l in (d / "init_memberships.txt").read_text().split("\n") if l]
    assert len(mem) == 1000 and int(mem[0][0]) == int(s2i[0])
    of = {int(m[0]): [int(x) for x in m[1:] if x != ""] for m in mem}
    for c, nodes in enumerate(comms):
        for p in nodes:
            assert c in of[int(s2i[p])]


def _files_equal_but_duration(da, db, names=("gamma.txt", "lambda.txt", "groups.txt", "communities.txt", "max.txt")):
    for nme in names:
        if (da / nme).exists() or (db / nme).exists():
            if nme == "max.txt":   # iteration, duration, likelihood, why: drop the duration
                a, b = (da / nme).read_text().split(), (db / nme).read_text().split()
                assert a[:1] + a[2:] == b[:1] + b[2:], nme
            else:
                assert (da / nme).read_bytes() == (db / nme).read_bytes(), nme
    va, vb = np.loadtxt(da / "validation.txt"), np.loadtxt(db / "validation.txt")
    assert va.shape == vb.shape
    assert np.array_equal(np.delete(va, 1, axis=1), np.delete(vb, 1, axis=1))


@pytest.mark.parametrize("extra", [["-no-stop", "-max-iterations", "100"], ["-no-stop", "-max-iterations", "98"],
                                   ["-no-stop", "-max-iterations", "103", "-sweep-batch", "3"], []])
def test_cli_pipelined_rfreq_rows(graph_files, tmp_path, extra):
    """-rfreq 5 through the PIPELINED loop (the default): the device records a row for every sweep whose _iter is a
    multiple of 5 -- [iter, iter + batch), sweep 0 included -- and every one of them reaches validation.txt, also the
    last one before -max-iterations ends the run and the row on which the stop rule fires (extra == []: the run ends
    by its stop rule).  Same files as the synchronous loop; rows against the oracle."""
    args = ["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-rfreq", "5"] + extra
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ra = _run(args, str(tmp_path / "a"), env={"SVINET_SYNC_REPORTS": "0"})
    rb = _run(args, str(tmp_path / "b"), env={"SVINET_SYNC_REPORTS": "1"})
    assert ra.returncode == 0 and rb.returncode == 0, ra.stderr + rb.stderr
    da, db = tmp_path / "a" / "n1000-k28-mmsb-linksampling", tmp_path / "b" / "n1000-k28-mmsb-linksampling"
    _files_equal_but_duration(da, db)
    v = np.loadtxt(da / "validation.txt")
    if extra:
        m = int(extra[2])
        ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, use_validation_stop=False, max_iterations=m, reportfreq=5)
        while ref.sweep() == 0:
            pass
        assert np.array_equal(v[1:, 0], np.arange(0, m + 1, 5))      # sweeps 0, 5, ..., the last multiple of 5 <= m
    else:
        ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28, reportfreq=5)
        while ref.sweep() != 2:
            assert ref.iter < 2000
        assert v[-1, 0] % 5 == 0
    np.testing.assert_allclose(np.delete(v, 1, axis=1), ref.rows, rtol=0, atol=6e-10)


def test_cli_default_graph_threshold_in_a_pipelined_run(graph_files, tmp_path):
    """The product default (SVILS_GRAPH_AFTER unset = 128): hipGraph capture happens in the MIDDLE of a pipelined run, at
    the first chunk past 128 sweeps, with report copies in flight on the copy stream.  300 sweeps; files equal those of
    the synchronous loop."""
    args = ["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-no-stop", "-max-iterations", "299"]
    env = {k: v for k, v in os.environ.items() if k != "SVILS_GRAPH_AFTER"}
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ra = subprocess.run([SVINET] + args, cwd=str(tmp_path / "a"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                        timeout=900, env=dict(env, SVINET_SYNC_REPORTS="0"))
    rb = subprocess.run([SVINET] + args, cwd=str(tmp_path / "b"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                        timeout=900, env=dict(env, SVINET_SYNC_REPORTS="1"))
    assert ra.returncode == 0 and rb.returncode == 0, ra.stderr + rb.stderr
    da, db = tmp_path / "a" / "n1000-k28-mmsb-linksampling", tmp_path / "b" / "n1000-k28-mmsb-linksampling"
    _files_equal_but_duration(da, db)
    assert np.loadtxt(da / "validation.txt").shape == (301, 11)


def test_cli_threaded_file_writers(graph_files, tmp_path):
    """gamma.txt / groups.txt through the threaded writers (waves of blocks formatted by threads and copied into a mapping
    of the file by threads: what runs at n = 1e6, k = 512) on a small model: byte-identical to the one-thread writes;
    init_gamma2 with its draws spread over threads as well"""
    args = ["-file", graph_files["astroph"], "-n", "17903", "-k", "20", "-link-sampling", "-no-stop", "-max-iterations", "3"]
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    ra = _run(args, str(tmp_path / "a"), env={"SVINET_WRITE_THREADS": "1", "SVINET_INIT_THREADS": "1"})
    rb = _run(args, str(tmp_path / "b"), env={"SVINET_WRITE_THREADS": "5", "SVINET_INIT_THREADS": "6", "SVINET_INIT_CHUNK_LINKS": "2000"})
    assert ra.returncode == 0 and rb.returncode == 0, ra.stderr + rb.stderr
    da, db = tmp_path / "a" / "n17903-k20-mmsb-linksampling", tmp_path / "b" / "n17903-k20-mmsb-linksampling"
    for nme in ("gamma.txt", "groups.txt", "lambda.txt", "communities.txt", "validation-edges.txt"):
        assert (da / nme).read_bytes() == (db / nme).read_bytes(), nme
    assert (da / "gamma.txt").stat().st_size > 3_000_000
