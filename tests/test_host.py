"""Product host side (C++: Network / Env / LinkSampling ctor / CLI) on CPU:
against the reference's shipped goldens and against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from svinet_amd.host_api import Setup
from conftest import GOLDEN, ROOT

SVINET = os.path.join(ROOT, "svinet_amd", "bin", "svinet")


@pytest.mark.parametrize("key,n,k,hr", [("lfr", 1000, 28, 0.01), ("assort", 75, 4, 0.01),
                                        ("astroph", 17903, 20, 0.02), ("astroph", 17903, 20, 0.01)])
def test_setup_equals_oracle(graph_files, key, n, k, hr):
    s = Setup(graph_files[key], n, k, heldout_ratio=hr)
    net = O.Network(graph_files[key], n)
    ref = O.LinkSampling(net, k, heldout_ratio=hr)
    assert (s.n, s.ones, s.nlinks) == (net.n, net.ones, ref.nlinks)
    assert np.array_equal(s.edges, net.edges())
    assert np.array_equal(s.seq2id, net.seq2id())
    assert np.array_equal(s.validation_accept, ref.validation_accept)
    assert np.array_equal(s.validation_sorted, ref.validation_sorted)
    assert np.array_equal(s.links, ref.links)
    assert np.array_equal(s.gamma, ref.gamma)          # bit-identical init_gamma2
    assert np.array_equal(s.lam, ref.lam)
    assert s.ones_prob == ref.ones_prob and s.total_pairs == ref.total_pairs


def test_setup_matches_reference_goldens(graph_files):
    s = Setup(graph_files["astroph"], 17903, 20, heldout_ratio=0.02)
    gold = np.loadtxt(os.path.join(GOLDEN, "ref_astroph_k20", "heldout-edges.txt"), dtype=np.int64)
    va = s.validation_accept
    mine = np.stack([s.seq2id[va[:, 0]], s.seq2id[va[:, 1]], va[:, 2]], 1)
    assert np.array_equal(mine, gold)


@pytest.mark.parametrize("eta_type", ["uniform", "fromdata", "sparse", "dense"])
def test_eta_types(graph_files, eta_type):
    s = Setup(graph_files["assort"], 75, 4, eta_type=eta_type)
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 4, eta_type=eta_type)
    assert s.eta == ref.eta


def test_seed_changes_the_stream(graph_files):
    a = Setup(graph_files["assort"], 75, 4, seed=7)
    b = Setup(graph_files["assort"], 75, 4)
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 4, seed=7)
    assert np.array_equal(a.gamma, ref.gamma) and not np.array_equal(a.gamma, b.gamma)


def test_in_memory_pairs_and_singletons():
    pairs = np.array([[10, 11], [11, 12], [12, 10], [12, 13], [13, 10], [11, 13], [20, 21], [21, 22], [22, 20], [13, 20]], np.int32)
    s = Setup(n=9, k=3, pairs=pairs, heldout_ratio=0.0)     # declared n larger than the graph
    assert s.n == 7 and s.singles == 2 and s.ones == 10 and s.nlinks == 10
    assert s.validation_sorted.shape == (0, 3)
    ref = O.LinkSampling(O.Network(n=9, pairs=pairs), 3, heldout_ratio=0.0)
    assert np.array_equal(s.gamma, ref.gamma) and np.array_equal(s.links, ref.links)


def _run(args, cwd):
    return subprocess.run([SVINET] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_cli_usage_and_rejections(tmp_path):
    assert _run([], str(tmp_path)).returncode != 0
    r = _run(["-help"], str(tmp_path))
    assert r.returncode == 0 and "-link-sampling" in r.stdout
    r = _run(["-file", "x", "-n", "10", "-k", "2"], str(tmp_path))        # no engine selected
    assert r.returncode == 2 and "only the -link-sampling and -batch engines" in r.stderr
    r = _run(["-file", "x", "-n", "10", "-k", "2", "-link-sampling", "-rnode"], str(tmp_path))
    assert r.returncode == 2 and "-rnode" in r.stderr
    r = _run(["-file", "/nonexistent", "-n", "10", "-k", "2", "-link-sampling"], str(tmp_path))
    assert r.returncode != 0 and "error reading" in r.stderr


def test_cli_host_outputs_and_loud_failure_without_gpu(graph_files, tmp_path):
    """On a CPU-only box the CLI must do the reference's host-side setup (output
    dir, param.txt, validation-edges.txt identical to the shipped golden) and then
    fail loudly when it reaches the device -- never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_cli.py")
    r = _run(["-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling", "-seed", "0"], str(tmp_path))
    assert r.returncode != 0 and "no HIP device" in r.stderr
    d = tmp_path / "n1000-k28-mmsb-linksampling"
    assert d.is_dir()
    gold = open(os.path.join(GOLDEN, "ref_lfr_k28", "heldout-edges.txt"), "rb").read()
    assert (d / "validation-edges.txt").read_bytes() == gold
    params = dict(l.split(": ", 1) for l in (d / "param.txt").read_text().split("\n") if ": " in l)
    assert params["nodes"] == "1000" and params["groups"] == "28" and params["alpha"] == "0.035714286"
    assert params["total pairs"] == "499500.000000000" or params["total pairs"] == "499500"
    assert params["ones_prob"] == "0.059801802" and params["heldout_ratio"] == "0.010000000"
    assert params["validation pairs (1s and 0s)"] == "298" and params["network ones"] == "29871"
    assert os.path.islink(str(d / "network.dat"))
    for f in ("infer.log", "test-edges.txt", "logl.txt", "validation.txt", "test.txt"):
        assert (d / f).exists()


def test_output_dir_naming(graph_files, tmp_path):
    _run(["-file", graph_files["assort"], "-n", "75", "-k", "4", "-link-sampling", "-label", "run1", "-seed", "3"], str(tmp_path))
    assert (tmp_path / "n75-k4-run1-seed3-linksampling").is_dir()


def test_accuracy_mode_trains_on_all_links(graph_files):
    """-accuracy: held-out pairs are still drawn (RNG stream unchanged) but no link is held out"""
    s = Setup(graph_files["assort"], 75, 4, accuracy=True)
    ref = O.LinkSampling(O.Network(graph_files["assort"], 75), 4, accuracy=True)
    assert s.nlinks == s.ones == ref.nlinks
    assert np.array_equal(s.links, ref.links) and np.array_equal(s.gamma, ref.gamma)
    assert np.array_equal(s.validation_accept, ref.validation_accept)


def test_cli_strid(tmp_path):
    """-strid: arbitrary node names, numbered by first appearance, table in str2id.txt sorted by name"""
    g = tmp_path / "names.txt"
    names = ["n%02d" % i for i in range(30)]
    lines = ["%s\t%s" % (names[i], names[(i + 1) % 30]) for i in range(30)] + ["%s\t%s" % (names[i], names[(i + 5) % 30]) for i in range(30)]
    g.write_text("\n".join(lines) + "\n")
    r = _run(["-file", str(g), "-n", "30", "-k", "3", "-link-sampling", "-strid", "-label", "s"], str(tmp_path))
    d = tmp_path / "n30-k3-s-linksampling"
    table = [l.split("\t") for l in (d / "str2id.txt").read_text().split("\n") if l]
    assert [t[0] for t in table] == sorted(names)
    # first appearance order: n00, n01, n02, ... (n01 appears as second token of line 1)
    assert dict((a, int(b)) for a, b in table) == {nm: i for i, nm in enumerate(names)}
    ve = (d / "validation-edges.txt").read_text()
    assert ve.endswith("\n")


def test_nmi_matches_the_shipped_mutual_txt():
    """-nmi: the reference appends the external `mutual` program's output to mutual.txt after every
    communities.txt; the authors' shipped LFR run holds both files, so the last mutual.txt line pins the
    native implementation (svinet_amd/host/nmi.cc)."""
    import ctypes as C
    from svinet_amd import host_api
    L = host_api.load()
    L.svih_nmi.argtypes = [C.c_char_p, C.c_char_p]
    L.svih_nmi.restype = C.c_double
    d = os.path.join(GOLDEN, "ref_lfr_k28")
    truth = os.path.join(GOLDEN, "graphs", "LFR-ground-truth-n1000-k28.txt")
    v = L.svih_nmi(os.fsencode(os.path.join(d, "communities.txt")), os.fsencode(truth))
    last = open(os.path.join(d, "mutual.txt")).read().split()[-1]
    assert "%g" % v == last == "0.897372"
    # a cover against itself is 1; unreadable files are reported
    own = os.path.join(d, "communities.txt")
    assert L.svih_nmi(os.fsencode(own), b"/nonexistent") < 0


def test_cli_gpus_argument_checks(tmp_path, graph_files):
    """-gpus / -device-list / -kshard are validated before anything forks or touches a device"""
    import subprocess
    for extra in (["-gpus", "0"], ["-gpus", "-3", "-kshard"], ["-gpus", "2", "-device-list", "0"],
                  ["-gpus", "40", "-kshard"], ["-gpus", "2", "-device-list", "a,b"]):
        r = subprocess.run([SVINET, "-file", graph_files["lfr"], "-n", "1000", "-k", "28", "-link-sampling"] + extra,
                           cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "error" in r.stderr, extra


def test_fixed_point_formatter_writes_printf_bytes():
    """gamma.txt ("%.5f") and groups.txt ("%.3f") are written by a scaled-integer formatter with an snprintf fallback
    (svinet_amd/host/fixedfmt.hh): every value must come out byte for byte as printf writes it -- random magnitudes,
    values a hair on either side of a rounding tie at 5 and at 3 decimals, exact ties, zeros, negatives, huge, inf/nan."""
    import ctypes as C
    from svinet_amd import host_api
    L = host_api.load()
    L.svih_fixed_format_mismatches.argtypes = [C.c_void_p, C.c_uint64]
    L.svih_fixed_format_mismatches.restype = C.c_uint64
    rng = np.random.default_rng(7)
    m = 400_000
    k5 = rng.integers(0, 10**9, m).astype(np.float64)
    k3 = rng.integers(0, 10**7, m).astype(np.float64)
    tie5, tie3 = (k5 + 0.5) / 1e5, (k3 + 0.5) / 1e3
    vals = np.concatenate([
        rng.random(m), rng.random(m) * 1e3, np.exp(rng.random(m) * 60 - 40), rng.integers(0, 200000, m) + rng.random(m),
        tie5, np.nextafter(tie5, np.inf), np.nextafter(tie5, -np.inf), tie3, np.nextafter(tie3, np.inf), np.nextafter(tie3, -np.inf),
        -rng.random(1000), rng.random(1000) * 1e12,
        np.array([0.0, -0.0, 1e300, 5e-6, 4.9999999999e-6, 0.0005, 0.0015, 0.0025, 171798.69183, 171798.69185, 99999.999995,
                  0.9999995, np.inf, -np.inf, np.nan]),
    ])
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    assert L.svih_fixed_format_mismatches(vals.ctypes.data, vals.size) == 0


def test_mt19937_jump_ahead_equals_drawing():
    """svinet_amd/host/mtjump.hh: the generator `pos` outputs after the seed by jump-ahead (characteristic polynomial by
    Berlekamp-Massey, x^pos mod phi, Horner on the state) gives the same 3000 outputs as drawing `pos` values does -- block
    boundaries, a -seed other than the default, positions beyond 2^28"""
    import ctypes as C
    from svinet_amd import host_api
    L = host_api.load()
    L.svih_mt_jump_check.argtypes = [C.c_ulong, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    L.svih_mt_jump_check.restype = C.c_int
    for seed, pos in ((0, 0), (0, 1), (0, 623), (0, 624), (0, 625), (0, 1247), (0, 1248), (7, 123457), (4357, 10 ** 7 + 3),
                      (0, 3 * 10 ** 8 + 11)):
        assert L.svih_mt_jump_check(seed, pos, 3000, None) == 0, (seed, pos)


def test_init_gamma_threaded_draws_are_the_sequential_stream(graph_files):
    """init_gamma2 with the draws spread over threads (every thread jumps to its chunks of the ONE gsl_rng stream): gamma
    and the held-out pairs are bit-identical to the single-threaded loop, whatever the thread count and chunk size"""
    import os
    from svinet_amd.host_api import Setup

    def run(env):
        old = {k: os.environ.pop(k, None) for k in ("SVINET_INIT_THREADS", "SVINET_INIT_CHUNK_LINKS")}
        os.environ.update(env)
        try:
            s = Setup(graph_files["astroph"], 17903, 20)
            return np.array(s.gamma), np.array(s.validation_accept)
        finally:
            for k in env:
                os.environ.pop(k, None)
            os.environ.update({k: v for k, v in old.items() if v is not None})

    g0, v0 = run({"SVINET_INIT_THREADS": "1"})
    for env in ({"SVINET_INIT_THREADS": "4", "SVINET_INIT_CHUNK_LINKS": "1000"},
                {"SVINET_INIT_THREADS": "7", "SVINET_INIT_CHUNK_LINKS": "333"},      # ragged last round, idle threads
                {"SVINET_INIT_THREADS": "5", "SVINET_INIT_CHUNK_LINKS": "50"},       # a hub's run of 504 links spans ten chunks and two rounds
                {"SVINET_INIT_THREADS": "16", "SVINET_INIT_CHUNK_LINKS": "12311"}):  # one round only
        g, v = run(env)
        assert np.array_equal(g, g0) and np.array_equal(v, v0), env


def test_init_streams_are_the_sequential_stream(graph_files):
    """what svils_init_gamma is handed (host/linksampling.cc: init_streams): state s of the jump-ahead chain is the generator
    s * per_stream outputs behind state 0, and state 0 stands where init_gamma2 began -- checked by running the twist forward in
    numpy from state 0 and by reproducing the host's gamma from the regenerated draws (the device kernel's recipe, on the CPU)"""
    from svinet_amd.host_api import Setup

    def twist(x):
        x = x.copy()
        up, lo = np.uint32(0x80000000), np.uint32(0x7fffffff)
        def tw(u, v):
            y = (u & up) | (v & lo)
            return (y >> np.uint32(1)) ^ (np.uint32(0x9908b0df) * (y & np.uint32(1)))
        x[:227] = x[397:624] ^ tw(x[:227], x[1:228])
        x[227:454] = x[0:227] ^ tw(x[227:454], x[228:455])
        x[454:623] = x[227:396] ^ tw(x[454:623], x[455:624])
        x[623] = x[396] ^ tw(x[623:624], x[0:1])[0]
        return x

    def temper(y):
        y = y ^ (y >> np.uint32(11))
        y = y ^ ((y << np.uint32(7)) & np.uint32(0x9d2c5680))
        y = y ^ ((y << np.uint32(15)) & np.uint32(0xefc60000))
        return y ^ (y >> np.uint32(18))

    s = Setup(graph_files["lfr"], 1000, 28)
    per, ns = 624 * 5, 4
    st = s.init_streams(ns, per)
    x = st[0].copy()
    raw = []
    for b in range(5 * (ns - 1) + 2):
        if b % 5 == 0 and b // 5 < ns:
            assert np.array_equal(x, st[b // 5]), "state %d is not %d outputs behind state 0" % (b // 5, b * 624)
        x = twist(x)
        raw.append(temper(x))
    raw = np.concatenate(raw)
    # the first links' draws, normalised and added in drawing order, give the host's gamma rows of nodes no later link touches
    edges = s.init_links()
    k = 28
    nl = raw.shape[0] // k
    g = np.zeros((1000, k))
    for l in range(nl):
        u = raw[l * k:(l + 1) * k].astype(np.float64) / 4294967296.0
        v = u / u.sum()
        g[edges[l, 0]] += v
        g[edges[l, 1]] += v
    done = np.setdiff1d(np.unique(edges[:nl]), np.unique(edges[nl:]))
    assert done.size > 0 and np.array_equal(g[done], s.gamma[done])
    assert s.init_offset() > 0
