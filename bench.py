#!/usr/bin/env python
"""bench.py -- edge-updates/sec of the link-sampling sweep on MI355X.

  python bench.py --gpus 1 --steps 100 --warmup 5
  python bench.py --gpus N --steps K --warmup W            # bare: spawns its own N ranks (one process per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one sweep of LinkSampling::infer()'s loop (src/linksampling.cc:571-789)
over all training links: phi pass, mean indicators, s3 pass, lambda update,
expectations, prune, validation likelihood + stop rule -- nothing skipped.
One edge-update = one training link processed in one sweep.
Workload (BASELINE.json north_star): ca-AstroPh, n=17903, k=20, seeded init.
Inputs are resident in HBM before the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (fixture, n, k)
    "astroph-k20": ("ca-AstroPh.csv.gz", 17903, 20),
    "astroph-k200": ("ca-AstroPh.csv.gz", 17903, 200),
    "lfr-k28": ("LFR-network-n1000-k28.txt.gz", 1000, 28),
}


def _fixture(name):
    src = os.path.join(ROOT, "tests", "golden", "graphs", name)
    tmp = tempfile.NamedTemporaryFile(delete=False, suffix=".txt")
    with gzip.open(src, "rb") as f:
        tmp.write(f.read())
    tmp.close()
    return tmp.name


def _synthetic_pairs(n, mean_deg, seed):
    """benchmark-only sparse graph: ring (no isolated node) + uniform random pairs"""
    import numpy as np
    rng = np.random.default_rng(seed)
    m = n * mean_deg // 2 - n
    a = rng.integers(0, n, size=m, dtype=np.int64)
    b = rng.integers(0, n, size=m, dtype=np.int64)
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    return np.concatenate([ring, np.stack([a, b], 1)]).astype(np.int32)


# the sources a phi kernel is compiled from: a PMC measurement is valid for the tree only while these are unchanged
KERNEL_SOURCES = {
    "lpl": ("svinet_amd/csrc/svils_lpl.hip", "svinet_amd/csrc/svils_cls.h", "svinet_amd/csrc/svils_devutil.h",
            "svinet_amd/csrc/svils_internal.h"),                                   # K <= 56: k_phi_lpl & co
    "row": ("svinet_amd/csrc/svils_device.hip", "svinet_amd/csrc/svils_devutil.h", "svinet_amd/csrc/svils_internal.h"),
}


def kernel_source_hashes(k):
    import hashlib
    out = {}
    for rel in KERNEL_SOURCES["lpl" if k <= 56 else "row"]:
        with open(os.path.join(ROOT, rel), "rb") as f:
            out[rel] = hashlib.sha256(f.read()).hexdigest()[:16]
    return out


def _traffic(workload, k):
    """HBM bytes per phi launch (and per whole sweep) from the committed rocprofv3 --pmc summary of this workload, with
    its provenance.  -> (record or None, reason).  A record measured on OTHER kernel sources than the ones in this tree
    is refused: tools/pmc_traffic.py stores the hashes of the kernel's source files (KERNEL_SOURCES) next to the bytes,
    and they must equal the hashes of the files this run was built from (there is no .git on the GPU box to ask)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(workload)
    except Exception as exc:
        return None, "profiles/traffic.json unreadable (%r)" % (exc,)
    if not rec or "phi_hbm_bytes_per_launch" not in rec:
        return None, "no PMC record for this workload"
    have = kernel_source_hashes(k)
    if rec.get("source_hashes") != have:
        changed = sorted(f for f in have if (rec.get("source_hashes") or {}).get(f) != have[f])
        return None, ("PMC record of commit %s refused: %s changed since it was measured -- re-run tools/evidence.sh pmc"
                      % (rec.get("commit"), ", ".join(changed)))
    return rec, "source hashes match"


def _roofline_fields(rec, k, n_nodes, workload):
    """Turns a _phi_record into the line's roofline fields.  `frac` is the fraction of the 8 TB/s HBM peak by the bytes
    that REALLY crossed the L2's memory side per launch (PMC counters, valid for this tree's kernel sources) -- it cannot
    exceed 1.  Without a valid counter record it is the fraction by the pull design's own byte model.  SURVEY 8d's
    32*K-per-link figure stays beside it as frac_survey_model: it charges a push-style scatter this pull-style kernel
    never performs and counts cache-fed bytes as HBM bytes, so it may exceed 1."""
    out = dict(rec)
    out["achieved_survey_model"] = out.pop("achieved")
    out["frac_survey_model"] = out.pop("frac")
    out["survey_model"] = "32*K bytes x (dense + active-set links of the timed sweeps) / phi time (SURVEY 8d)"
    out["pull_model"] = _pull_model(rec, k, n_nodes)
    t = rec["avg_launch_us"] * 1e-6
    tr, why = _traffic(workload, k)
    if tr:
        out["traffic"] = tr["phi_hbm_bytes_per_launch"]
        out["traffic_source"] = {kk: tr.get(kk) for kk in ("source", "commit", "counters", "source_hashes")}
        out["achieved"] = tr["phi_hbm_bytes_per_launch"] / t / 1e9
        out["frac_basis"] = "hbm_counter"
        out["frac_hbm_counter"] = out["achieved"] / HBM_PEAK_GBS
        if tr.get("sweep_hbm_bytes"):
            out["sweep_traffic"] = tr["sweep_hbm_bytes"]
    else:
        out["traffic"] = None
        out["traffic_source"] = {"refused": why}
        out["achieved"] = out["pull_model"]["achieved"]
        out["frac_basis"] = "pull_model"
    out["frac"] = out["achieved"] / HBM_PEAK_GBS
    return out


def _phi_record(eng, k, label):
    """roofline numbers of the phi launches timed since enable_timing(): algorithmic bytes are 32*K per
    link that took a softmax branch (dense or active-set) IN THE TIMED SWEEPS -- shortcut links move
    no rows (src/linksampling.cc:622-631)."""
    from svinet_amd import _svils   # noqa: F401
    ms, n = eng.timing()["phi"]
    dense, sparse, shortcut = (int(x) for x in eng.timed_links())
    t = ms * 1e-3
    alg = 32.0 * k * (dense + sparse)
    achieved = alg / t / 1e9 if t > 0 else 0.0
    return {"window": label, "launches_timed": int(n), "avg_launch_us": (t / n * 1e6) if n else None,
            "links_in_timed_launches": {"dense": dense, "sparse": sparse, "shortcut": shortcut},
            "algorithmic_bytes_per_launch": (alg / n) if n else None,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}


def dense_only_window(setup, k, device, min_launches=10):
    """phi launches of sweeps 0..3 of the seeded run -- before any node is flagged converged every link
    takes the full softmax, the one window the 32*K byte model describes exactly.  One engine,
    put back to the seeded initial state between repetitions (set_state + the constructor's loop state),
    after an untimed first repetition: the first launches on a new stream pay one-off costs (code load,
    scratch set-up) that belong to no sweep."""
    from svinet_amd import _svils
    e = setup.engine(use_validation_stop=False, device=device)
    gamma0 = setup.gamma if setup.host_gamma else None
    tot_ms, tot_n, tot_links, reps = 0.0, 0, [0, 0, 0], 0
    first = True
    while tot_n < min_launches:
        if not first:
            _reseed(e, setup, gamma0)
        e.enable_timing(0 if first else (1 << _svils.KERNEL_PHI), 1)
        before = e.control().sweeps_done
        e.sweep(4)
        e.synchronize()
        if not first:
            ms, n = e.timing()["phi"]
            li = e.timed_links()
            tot_ms += ms; tot_n += n
            for j in range(3):
                tot_links[j] += int(li[j])
            assert int(e.sweep_stats(before, 4)[:, 2].sum()) == 0, "the dense-only window met a shortcut link"
            reps += 1
        first = False
    e.close()
    t = tot_ms * 1e-3
    alg = 32.0 * k * (tot_links[0] + tot_links[1])
    achieved = alg / t / 1e9
    rec = {"window": "sweeps 0..3 of the seeded run, %d repetitions from the re-seeded initial state" % reps,
           "launches_timed": tot_n, "avg_launch_us": t / tot_n * 1e6,
           "links_in_timed_launches": {"dense": tot_links[0], "sparse": tot_links[1], "shortcut": tot_links[2]},
           "algorithmic_bytes_per_launch": alg / tot_n, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "achieved_survey_model": achieved, "frac_survey_model": achieved / HBM_PEAK_GBS}
    # no PMC pass covers exactly these launches: the fraction is the pull design's own byte model (state cache-resident)
    rec["pull_model"] = _pull_model(rec, k, setup.n)
    rec["achieved"], rec["frac"], rec["frac_basis"] = rec["pull_model"]["achieved"], rec["pull_model"]["frac"], "pull_model"
    return rec


def _pull_model(rec, k, n_nodes):
    """The bytes the pull-style phi pass itself has to move: every directed softmax entry
    reads its neighbour's Elogpi row (8K), every node reads its own row once and writes its accumulated
    gammanext row once (16K per node) -- 16*K*L_softmax + 16*K*N per launch.  SURVEY 8d's 32*K-per-link model
    also charges the push-style scatter (a 16K read-modify-write per link) that this design never performs,
    which is why `frac` by that model can exceed 1 on HBM-bound sizes; this one cannot."""
    nl = rec["launches_timed"]
    li = rec["links_in_timed_launches"]
    b = 16.0 * k * (li["dense"] + li["sparse"]) / nl + 16.0 * k * n_nodes
    t = rec["avg_launch_us"] * 1e-6
    return {"bytes_per_launch": b, "achieved": b / t / 1e9, "frac": b / t / 1e9 / HBM_PEAK_GBS,
            "model": "16*K*L_softmax + 16*K*N (neighbour rows + own row + one gammanext row per node)"}


def _reseed(eng, setup, gamma0=None):
    """back to the seeded initial state of `LinkSampling ls(env, network)`: gamma / lambda of init_gamma2 /
    init_lambda, the constructor's loop state (src/linksampling.cc:19-33), no converged flag"""
    if setup.host_gamma:
        eng.set_state(setup.gamma if gamma0 is None else gamma0, setup.lam)
    else:
        setup.device_init(eng)          # drawn again on the device (svils_init_gamma: the same bits, ~30 ms at n = 1e6, k = 512)
    eng.set_control(iter=0, annealing=1, write_comm=0, nh=0, prev_h=-2147483647.0, max_h=-2147483647.0)


def repeated_windows(eng, setup, warmup, steps, reps, torch, runner=None, dist=None):
    """The SAME sweep window (sweeps warmup..warmup+steps of the seeded run) `reps` times from the re-seeded
    state, each repetition timed exactly like the first one (device sync on both sides of exactly `steps` sweeps,
    hipGraph replay, no per-kernel events).  A 20-sweep window is 1.2 ms of GPU time: a single one on a fresh
    lease sees clocks that have not settled, the median over repetitions does not."""
    import numpy as np
    gamma0 = setup.gamma if setup.host_gamma else None
    times, finals = [], []
    runner = runner or eng     # N > 1: the sharded driver (every rank re-seeds its replica of the state; MAX over ranks per window)
    for _ in range(reps):
        _reseed(eng, setup, gamma0)
        runner.sweep(warmup)
        times.append(_timed(runner, eng, steps, dist, torch))
        c = eng.control()
        finals.append((int(c.iter), int(c.links_dense), int(c.links_sparse), int(c.links_shortcut)))
    assert len(set(finals)) == 1, "repetitions of the same window ended in different states: %r" % (set(finals),)
    t = np.sort(np.asarray(times))
    return times, {"reps": reps, "median_ms_per_step": float(np.median(t)) / steps * 1e3,
                   "min_ms_per_step": float(t[0]) / steps * 1e3, "max_ms_per_step": float(t[-1]) / steps * 1e3,
                   "p10_ms_per_step": float(t[int(0.1 * (len(t) - 1))]) / steps * 1e3,
                   "p90_ms_per_step": float(t[int(round(0.9 * (len(t) - 1)))]) / steps * 1e3,
                   "end_state_identical": True}


HBM_BOUND_WORKLOAD = "synthetic:200000:512:24"
CONFIG5_WORKLOAD = "mmsb:1000000:512:24"     # BASELINE config 5: planted MMSB graph, n = 1e6, k = 512


def hbm_bound_record(device, sweeps=10, workload=HBM_BOUND_WORKLOAD):
    """The same phi pass on a state that cannot sit in the 256 MB Infinity Cache, measured in this invocation:
    the fraction of the 8 TB/s HBM peak by SURVEY 8d's algorithmic bytes, by the pull design's own byte model and
    by the PMC-counted traffic of this workload (committed profile, provenance given).
    Default: n = 2e5, k = 512 on a uniform random graph (0.82 GB per n-by-k array).
    workload = CONFIG5_WORKLOAD: BASELINE config 5 at full size -- the planted MMSB graph, n = 1e6, k = 512,
    4.1 GB per n-by-k array (3 resident), the size SURVEY 8d makes the >= 50 % HBM-roofline claim on."""
    from svinet_amd import _svils
    t0 = time.perf_counter()
    setup, _, _, n, k, data = _load_workload(workload)
    eng = setup.engine(use_validation_stop=False, device=device)
    eng.sweep(2)
    eng.synchronize()
    setup_s = time.perf_counter() - t0
    # whole sweeps first, no events: ms per sweep as svils_sweep runs them
    t1 = time.perf_counter()
    eng.sweep(sweeps)
    eng.synchronize()
    el = time.perf_counter() - t1
    # then the per-kernel pass (events around every launch)
    eng.enable_timing(0xff, 1)
    eng.sweep(sweeps)
    eng.synchronize()
    tm = eng.timing()
    rec = _phi_record(eng, k, "sweeps %d..%d of the seeded run" % (2 + sweeps, 2 + 2 * sweeps))
    L = int(setup.nlinks)
    row_bytes = eng.device_buffer(_svils.BUF_GAMMA)[2]          # the engine's own row stride
    rec.update({"workload": "%s: n=%d k=%d links/sweep=%d, state %.2f GB per n-by-k array (3 resident)"
                            % (workload, n, k, L, n * row_bytes / 1e9),
                "data": data,
                "ms_per_sweep": el / sweeps * 1e3, "edge_updates_per_s": L * sweeps / el,
                "kernels_us": {kk: v[0] / max(v[1], 1) * 1e3 for kk, v in tm.items() if v[1]},
                "setup_s": setup_s,
                "kernel": "k_phi<8,false,true> (row-per-wavefront, product form on exp(Elogpi) rows)"})
    if workload == CONFIG5_WORKLOAD:
        # a long untimed-by-events stretch of the same run: 250 further sweeps (~8 s of uninterrupted device work -- also what
        # lets an outside GPU-busy sampler see this command at all: everything else it does on the device lasts milliseconds)
        eng.enable_timing(0, 1)
        t2 = time.perf_counter()
        eng.sweep(250)
        eng.synchronize()
        el2 = time.perf_counter() - t2
        c2 = eng.control()
        rec["sustained"] = {"sweeps": 250, "first_sweep": 2 + 2 * sweeps, "ms_per_sweep": el2 / 250 * 1e3,
                            "edge_updates_per_s": L * 250 / el2,
                            "links_last_sweep": {"dense": int(c2.links_dense), "sparse": int(c2.links_sparse), "shortcut": int(c2.links_shortcut)}}
    rec = _roofline_fields(rec, k, n, workload)
    if rec.get("sweep_traffic"):
        # every kernel of the sweep, not only phi: PMC bytes of one whole sweep over the measured sweep time
        rec["sweep_achieved_counter"] = rec["sweep_traffic"] / (rec["ms_per_sweep"] * 1e-3) / 1e9
        rec["sweep_frac_hbm_counter"] = rec["sweep_achieved_counter"] / HBM_PEAK_GBS
    rec["note"] = ("pull-style phi reads one neighbour row per directed entry and writes each gammanext row once: the PMC-counted "
                   "HBM bytes sit within a few per cent of pull_model and below the 32*K-per-link model"
                   + ("" if n * row_bytes * 3 > 1e9 * 4 else "; at this size part of the traffic is fed by the 256 MB Infinity Cache "
                      "(FETCH_SIZE counts those hits): config5 is the all-HBM figure"))
    eng.close()
    setup.close()
    return rec


def cpu_baseline(path, pairs, n, k, warmup, steps, min_s=12.0, budget_s=25.0):
    """The oracle (a port of the reference's single-threaded loop, oracle/svinet_oracle.c) timed on this box's host
    cores over the same sweep window of the same seeded run -- repeated from the seeded state until >= min_s of CPU
    work have been timed (a 20-sweep window is ~1.6 s), bounded to ~budget_s."""
    from oracle import oracle as O
    net = O.Network(path, n) if path else O.Network(n=n, pairs=pairs)
    done, el, reps, per = 0, 0.0, 0, None
    while el < min_s and el < budget_s:
        ref = O.LinkSampling(net, k, use_validation_stop=False)
        t_w0 = time.perf_counter()
        for _ in range(warmup):
            ref.sweep()
        if per is None and warmup:
            per = (time.perf_counter() - t_w0) / warmup
        t0 = time.perf_counter()
        d = 0
        while d < steps:
            ref.sweep()
            d += 1
            if el + time.perf_counter() - t0 > budget_s and d >= 2:
                break
        el += time.perf_counter() - t0
        done += d
        reps += 1
        if d < steps:
            break
    return {"value": ref.nlinks * done / el, "unit": "edge-updates/s", "cores": 1, "kind": "port",
            "sample": "oracle (single-thread C port, -O2), sweeps %d..%d of the same seeded run, %d sweeps timed in %d "
                      "repetition(s) of the window, %.1f s" % (warmup, warmup + steps, done, reps, el),
            "host_cpus": os.cpu_count(), "warmup_s_per_sweep": per}


def cpu_baseline_allcores(path, pairs, n, k, warmup, steps, budget_s=12.0):
    """An ALL-CORES figure next to the contract's single-thread baseline (SURVEY 8d "optionally"): the oracle's sweep
    threaded with OpenMP (oracle/svinet_oracle_omp.c -- the reference's path itself has no threads; sums are taken in
    another order, results equal to rounding: tests/test_oracle_omp.py).  The thread count that is fastest on this
    box among a few tried is reported, with all tried counts listed."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")    # before libgomp loads: idle threads sleep instead of spinning
    from oracle import oracle as O
    net = O.Network(path, n) if path else O.Network(n=n, pairs=pairs)
    ncpu = os.cpu_count() or 1
    quota = None                                            # CPU-time limit of the container, in cores (None: no limit found)
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):        # cgroup v2
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q[0] == "max" else float(q[0]) / float(q[1])
        else:                                               # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
    except Exception:
        pass
    tried, t_all0 = {}, time.perf_counter()
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, 256)}):
        if time.perf_counter() - t_all0 > budget_s:
            break
        ref = O.LinkSampling(net, k, use_validation_stop=False)
        for _ in range(min(warmup, 5)):
            ref.sweep_omp(th)
        t0, d = time.perf_counter(), 0
        while d < steps and (d < 3 or time.perf_counter() - t0 < budget_s / 6):
            ref.sweep_omp(th)
            d += 1
        tried[th] = ref.nlinks * d / (time.perf_counter() - t0)
    best = max(tried, key=tried.get)
    return {"value": tried[best], "unit": "edge-updates/s", "cores": best, "kind": "port-openmp",
            "sample": "oracle sweep threaded with OpenMP (not the reference's code path, which is single-threaded; results "
                      "equal to rounding), first sweeps of the same seeded run, a few sweeps per thread count",
            "threads_tried": {str(t): v for t, v in tried.items()}, "host_cpus": ncpu,
            "cpu_affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_quota_cores": quota}


def cli_end_to_end(path, n, k, value_ms_per_step):
    """What the user of the DROP-IN gets: wall time of `svinet -file ca-AstroPh -n 17903 -k 20 -link-sampling` run to its
    stop rule, split into read / constructor / graph upload / sweeps (incl. the per-report files) / final files, from the
    CLI's own clocks (SVINET_TIMING_FILE).  Three forms: the default (report snapshots collected while the device sweeps
    on, automatic chunks), -sweep-batch 1 (one report per sweep, the reference's cadence, src/linksampling.cc:777-786)
    and the synchronous loop of earlier rounds at -sweep-batch 1 (a stream synchronisation + file rewrite per sweep)."""
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "svinet_amd", "bin", "svinet")
    out = {"command": "svinet -file ca-AstroPh.csv -n %d -k %d -link-sampling   (runs to the validation stop rule)" % (n, k),
           "library_ms_per_sweep": value_ms_per_step}
    out["scenarios"] = {"to_stop": "the default flags: the run ends on the validation stop rule (31 sweeps with this revision's inputs; "
                                   "99 in the authors' 2013 log)",
                        "300_sweeps": "-no-stop -max-iterations 299: long enough for graph replay (the library captures its hipGraphs once "
                                      "a handle has run 128 sweeps)"}
    for name, extra, env, scen in [(nm + "/" + sc, ex, en, sa)
                                   for sc, sa in (("to_stop", []), ("300_sweeps", ["-no-stop", "-max-iterations", "299"]))
                                   for nm, ex, en in (("default", [], {}), ("sweep_batch_1", ["-sweep-batch", "1"], {}),
                                                      ("sweep_batch_16", ["-sweep-batch", "16"], {}),
                                                      ("synchronous_sweep_batch_1", ["-sweep-batch", "1"], {"SVINET_SYNC_REPORTS": "1"}))]:
        # a run to the stop rule is a 2 ms window in a process that has just started: three runs, the one with the median
        # sweep time is the record (all three times are kept beside it); the 300-sweep runs are long enough for one
        runs = []
        for _ in range(3 if not scen else 1):
            d = tempfile.mkdtemp(prefix="svinet_cli_")
            try:
                tf = os.path.join(d, "timing.json")
                t0 = time.perf_counter()
                cli_env = {kk: vv for kk, vv in os.environ.items() if kk != "SVILS_GRAPH_AFTER"}     # the product's defaults
                cli_env.update(env, SVINET_TIMING_FILE=tf)
                r = subprocess.run([exe, "-file", path, "-n", str(n), "-k", str(k), "-link-sampling"] + extra + scen, cwd=d,
                                   env=cli_env, capture_output=True, text=True, timeout=600)
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    out[name] = {"error": r.stderr[-300:]}
                    runs = []
                    break
                tm = json.load(open(tf))
                tm["process_wall_s"] = wall
                runs.append(tm)
            finally:
                shutil.rmtree(d, ignore_errors=True)
        if not runs:
            continue
        runs.sort(key=lambda t: t["sweeps_s"])
        tm = runs[len(runs) // 2]
        if len(runs) > 1:
            tm["sweeps_s_of_every_run"] = [t["sweeps_s"] for t in runs]
        tm["ms_per_sweep"] = tm["sweeps_s"] / tm["sweeps"] * 1e3
        tm["vs_library_sweep"] = tm["ms_per_sweep"] / value_ms_per_step
        out[name] = tm
    return out


def _load_workload(name):
    """-> (setup, path, pairs, n, k, data description).  The generated graphs (hundreds of thousands of nodes and up) leave
    init_gamma2 to the device (Setup(host_gamma=False) -> svils_init_gamma, bit-identical): no n x k array on the host"""
    from svinet_amd.host_api import Setup
    path, pairs = None, None
    if name.startswith("synthetic"):
        _, sn, sk, sd = name.split(":")
        n, k = int(sn), int(sk)
        pairs = _synthetic_pairs(n, int(sd), 20240517)
        setup = Setup(n=n, k=k, pairs=pairs, host_gamma=False)
        data = "synthetic sparse graph (ring + uniform random pairs, seed 20240517), seeded init"
    elif name.startswith("mmsb"):
        from svinet_amd import mmsbgen_sparse
        _, sn, sk, sd = name.split(":")
        n, k = int(sn), int(sk)
        pairs = mmsbgen_sparse.generate(n, k, int(sd))
        setup = Setup(n=n, k=k, pairs=pairs, host_gamma=False)
        data = ("synthetic sparse MMSB graph (svinet_amd/mmsbgen_sparse.py: Dirichlet(0.05) top-4 memberships, "
                "Beta(4700.59,0.77) rates, Philox seed %d), seeded init" % mmsbgen_sparse.DEFAULT_SEED)
    else:
        if name not in WORKLOADS and name.startswith("astroph-k"):   # any K on ca-AstroPh
            WORKLOADS[name] = ("ca-AstroPh.csv.gz", 17903, int(name[len("astroph-k"):]))
        fixture, n, k = WORKLOADS[name]
        path = _fixture(fixture)
        setup = Setup(path, n, k)
        data = "reference example graph %s (fixture copy), seeded init (MT19937 4357)" % fixture
    return setup, path, pairs, n, k, data


class _Sharded:
    """One chain over all ranks: node-block sharding with the exchanges issued by the device library itself
    (svils_sweep_sharded: RCCL all-reduce / all-gather on the engine's stream between the phases of a sweep)."""

    def __init__(self, setup, rank, world, device, dist, equal_blocks=False):
        import numpy as np
        from svinet_amd import _svils
        from svinet_amd.sharded import balanced_bounds, block_size, equal_bounds
        # node blocks balanced by work (CSR entries + a per-node share: svils_balance_node_blocks); mini-batch steps keep
        # the equal blocks they need
        self.bounds = equal_bounds(setup.n, world) if equal_blocks else balanced_bounds(setup.links, setup.n, world)
        B = block_size(setup.n, world)
        self.eng = setup.engine(use_validation_stop=False, device=device,
                                node_block=(int(self.bounds[rank]), int(self.bounds[rank + 1])),
                                n_alloc=B * world if equal_blocks else 0)
        if not equal_blocks:
            self.eng.set_node_blocks(rank, world, self.bounds)
        ids = [_svils.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        self.eng.comm_init(ids[0], rank, world)
        deg = np.bincount(np.asarray(setup.links).ravel(), minlength=setup.n)
        ent = np.array([int(deg[int(self.bounds[r]):int(self.bounds[r + 1])].sum()) for r in range(world)], dtype=np.float64)
        self.balance = {"node_blocks": [int(x) for x in self.bounds], "csr_entries_per_rank": [int(x) for x in ent],
                        "max_over_mean": float(ent.max() / ent.mean()) if ent.sum() else None,
                        "cut": "equal node counts" if equal_blocks else "svils_balance_node_blocks (entries + 0.5 per node); s3 pass cut by link count"}

    def sweep(self, n):
        self.eng.sweep_sharded(n)


class _ShardedSteps(_Sharded):
    """the north_star's global step: mini-batch (Robbins-Monro) steps over the node-block shards, every rank a
    window of its own block per step, svils_step_sharded (all-reduce of the K-vectors, broadcasts of the touched
    gamma rows).  `sweep(n)` runs n steps; WINDOWS of them are one pass over the nodes."""
    WINDOWS = 8

    def __init__(self, setup, rank, world, device, dist):
        from svinet_amd.sharded import block_size
        super().__init__(setup, rank, world, device, dist, equal_blocks=True)
        B = block_size(setup.n, world)
        self.eng.set_stochastic(batch_nodes=(B + self.WINDOWS - 1) // self.WINDOWS, tau0=64.0, kappa=0.6, shard_block=B)

    def sweep(self, n):
        self.eng.step_sharded(n)


class _KSharded:
    """One chain over all ranks, K-sharded: rank r holds the columns [k r / G, k (r+1) / G) of every row and the
    library all-reduces the four coupling buffers itself (svils_sweep_ksharded)."""

    def __init__(self, setup, rank, world, device, dist):
        import numpy as np
        from svinet_amd import _svils
        from svinet_amd.ksharded import column_slices
        k0, k1 = column_slices(setup.k, world)[rank]
        self.eng = e = _svils.Engine(setup.n, setup.k, ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta,
                                     link_thresh=setup.link_thresh, lt_min_deg=setup.lt_min_deg,
                                     use_validation_stop=False, device=device, k_slice=(k0, k1))
        e.set_graph(setup.links)
        e.set_validation(setup.validation_sorted)
        if setup.host_gamma:
            e.set_state(np.ascontiguousarray(setup.gamma[:, k0:k1]), np.ascontiguousarray(setup.lam[k0:k1]))
        else:
            setup.device_init(e, lam=np.ascontiguousarray(setup.lam[k0:k1]))
        ids = [_svils.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        e.comm_init(ids[0], rank, world)
        e.ksh_init_state()

    def sweep(self, n):
        self.eng.sweep_ksharded(n)


# N > 1 side records, in the order they run: (name, workload, timed sweeps, layout).  BASELINE config 5 in the layout built for
# it comes FIRST -- it is the one configuration the design expects to scale, so it must not sit behind
# six other records under the global cut-off on the first real node; the two config-5 records share one host set-up (~25 s).
SIDE_RECORDS = (
    ("ksharded_config5_mmsb_n1m_k512", "mmsb:1000000:512:24", 5, "kshard"),
    ("config5_mmsb_n1m_k512", "mmsb:1000000:512:24", 5, "nodeblock"),
    ("config4_astroph_k200", "astroph-k200", 50, "nodeblock"),
    ("ksharded_config4_astroph_k200", "astroph-k200", 50, "kshard"),
    ("hbm_bound_n200k_k512", "synthetic:200000:512:24", 10, "nodeblock"),
    ("ksharded_hbm_bound_n200k_k512", "synthetic:200000:512:24", 10, "kshard"),
    # mini-batch steps on the headline graph: 8 windows per node block, 80 steps = 10 passes
    ("minibatch_steps_astroph_k20", "astroph-k20", 80, "steps"),
)
# seconds a record may need on top of the sweeps themselves (host set-up of its workload on every rank at once, graph capture,
# the one-GPU run beside it): what --extra-timeout scales with
SIDE_RECORD_BUDGET_S = {"mmsb:1000000:512:24": 150, "synthetic:200000:512:24": 45, "astroph-k200": 25, "astroph-k20": 20}


def side_record_plan(extra_list=""):
    """the records a run will take, in order, and the global cut-off that goes with them"""
    want = [x for x in extra_list.split(",") if x] if extra_list else None
    plan = [r for r in SIDE_RECORDS if want is None or r[0] in want]
    seen, budget = set(), 60
    for _, wl, _, _ in plan:
        budget += SIDE_RECORD_BUDGET_S[wl] // (2 if wl in seen else 1)    # a workload already set up costs the sweeps only
        seen.add(wl)
    return plan, budget


def model_prediction(workload, layout, world):
    """What tools/shard_cost.py PREDICTS for this record -- per-rank compute measured on ONE GPU + the stated link model --
    so that the first measured N > 1 number is read against it in the same line.  -> dict or None (no model row)."""
    try:
        with open(os.path.join(ROOT, "profiles", "shard_cost_model.json")) as f:
            m = json.load(f)
        row = m["rows"]["%s|%s|%d" % (workload, layout, world)]
    except Exception:
        return None
    out = dict(row)
    out["link_model"] = m.get("link_model")
    out["source"] = m.get("source")
    return out


def _timed(runner, eng, steps, dist, torch):
    """barrier + device sync on both sides of exactly `steps` sweeps; MAX over ranks"""
    eng.synchronize(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.sweep(steps)
    eng.synchronize(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return el


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process (which has not touched HIP or torch) becomes the
    launcher -- N children of the same command line, one per GPU, with the environment torch.distributed.run would
    have given them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on 127.0.0.1, a free port).  Rank 0
    owns stdout (the one JSON line); a rank that dies takes the others with it (exact PIDs, no patterns)."""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "BENCH_SELF_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what this pool's host driver supports
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                # the others usually leave by themselves for the same reason (and say why): a few seconds for that, then the end
                t_end = time.time() + 5.0
                while time.time() < t_end and any(procs[q].poll() is None for q in live):
                    time.sleep(0.05)
                still = [q for q in sorted(live) if procs[q].poll() is None]
                if still:
                    sys.stderr.write("bench.py: rank %d exited with %d; ending the other ranks\n" % (r, code))
                for q in still:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def _rccl_evidence(eng, dist, rank, world, local_rank, torch):
    """Did RCCL see N ranks on N devices?  Every rank asks the bound library about ITS communicator
    (svils_comm_query: ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion + the device's PCI bus
    id + the path the nccl* symbols came from); rank 0 puts the answers side by side."""
    mine = eng.comm_query()
    mine.update({"pid": os.getpid(), "local_rank": local_rank, "gpu": torch.cuda.get_device_name(local_rank),
                 "visible_devices": torch.cuda.device_count()})
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank != 0:
        return None
    return {"nranks": sorted({a["nranks"] for a in allr}), "rccl_version": sorted({a["rccl_version"] for a in allr}),
            "library": sorted({a["library"] for a in allr}),
            "distinct_devices": len({a["pci_bus_id"] for a in allr}),
            "ranks": [{kk: a[kk] for kk in ("rank", "hip_device", "pci_bus_id", "pid", "gpu", "row_communicator")} for a in allr],
            "control_plane": "torch.distributed gloo over 127.0.0.1 (communicator id, barriers, max-over-ranks of the "
                             "timings); every byte of the sweep's exchanges goes through the library's own RCCL "
                             "communicator(s), created by svils_comm_init from the broadcast id"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="astroph-k20",
                    help="|".join(WORKLOADS) + "|astroph-k<K>|synthetic:<n>:<k>:<mean_deg>|mmsb:<n>:<k>:<mean_deg>")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-period", type=int, default=9,
                    help="N=1: the event pass after the timed region brackets the phi launch with hipEvents on every P-th "
                         "sweep of the same window (those sweeps launch eagerly, the others replay hipGraphs); lowered "
                         "automatically so that at least 10 launches are timed")
    ap.add_argument("--no-hbm-bound", action="store_true", help="skip the HBM-bound sub-record (n=2e5, k=512; ~15 s)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the config-5 sub-record (planted MMSB graph, n=1e6, k=512 on this one GPU; ~60 s, 13 GB of HBM)")
    ap.add_argument("--reps", type=int, default=50,
                    help="N=1: repetitions of the timed window from the re-seeded state; `value` is their median (0: the single first window)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="take the N>1 code path (process group, RCCL communicator, sharded driver, side records) with "
                         "whatever world size there is -- a one-GPU box can exercise it with a world of one")
    ap.add_argument("--no-extra", action="store_true", help="N>1: skip the side records (config 4, HBM-bound size)")
    ap.add_argument("--extra-list", default="", help="N>1: comma-separated names of the side records to run (default: all)")
    ap.add_argument("--test-one-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cli", action="store_true", help="N=1: skip the cli_end_to_end record (three runs of the svinet binary to its stop rule, ~5 s)")
    ap.add_argument("--cli-only", action="store_true", help="print the cli_end_to_end record alone")
    ap.add_argument("--main-timeout", type=int, default=420,
                    help="N>1: seconds the communicator set-up + warm-up + timed sweeps may take before rank 0 prints "
                         "an error line and every rank exits")
    ap.add_argument("--extra-timeout", type=int, default=0,
                    help="N>1: seconds after the main measurement before a watchdog prints the JSON line and exits "
                         "(0: scaled with the side records requested -- side_record_plan)")
    ap.add_argument("--record-timeout", type=int, default=0,
                    help="N>1: seconds ONE side record may take; past it the record is marked as hung, the ones behind it as not run, "
                         "the JSON line is printed with everything measured so far and every rank exits (a rank stuck in a collective "
                         "cannot be brought back).  0: 200 s, or twice the record's set-up budget where that is more (config 5: eight "
                         "concurrent 25 s host set-ups under a 16-core quota)")
    args = ap.parse_args()
    if args.extra_timeout <= 0:
        # scaled with the records requested, but never more than five minutes behind the main measurement by default: the one JSON line
        # is owed to a driver whose own limit is unknown, and the records run in order of what they are worth (config 5 K-sharded first)
        # (--test-one-gpu: N ranks share ONE GPU, every record runs N times slower than on a node -- the rehearsal keeps the whole plan)
        plan_s = side_record_plan(args.extra_list)[1]
        args.extra_timeout = plan_s if getattr(args, "test_one_gpu", False) else min(plan_s, 300)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: be the launcher (before anything touches HIP in this process)
        raise SystemExit(_spawn_ranks(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (ranks started by a launcher other than _spawn_ranks: before HIP is touched)
    import torch
    from svinet_amd import _svils

    # the library loop measured here replays hipGraphs from its warm-up on (the product default captures them only once
    # a handle has run 128 sweeps: a capture costs more than a short run -- the cli_end_to_end record runs that default)
    os.environ.setdefault("SVILS_GRAPH_AFTER", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.test_one_gpu:
        # TEST MODE (tests/test_gpu_native_ranks.py): every rank on GPU 0, a gloo process group for the id broadcast and
        # the timing reductions, and the library's collectives on the tests-only transport SVILS_RCCL_LIBRARY names --
        # the N > 1 code path of this file on a one-GPU box.  Never a measurement.
        assert os.environ.get("SVILS_RCCL_LIBRARY"), "--test-one-gpu needs the tests' transport (SVILS_RCCL_LIBRARY)"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_sharded
    dist = None
    if multi:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # control plane only (communicator id, barriers, max-over-ranks of the timings): gloo.  The data path's RCCL
        # communicators are the library's own (svils_comm_init) -- no second set of RCCL communicators in this process.
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))

    # N > 1: RCCL has never carried more than one rank of this code on the hardware available to its builders.
    # If the communicator set-up or the timed sharded sweeps ever block, rank 0 still owes the driver a JSON
    # line: it says so (value null, the reason in "error") and every rank leaves instead of hanging the node.
    main_done = None
    fallback = {}    # what the main watchdog can still print if the replayed window blocks (N > 1)
    if multi:
        import threading
        main_done = threading.Event()

        def main_watchdog():
            if not main_done.wait(timeout=args.main_timeout):
                if rank == 0:
                    line = {"metric": "edge-updates/sec (link-sampling SVI step)", "value": None, "unit": "edge-updates/s", "n_gpus": world, "steps": args.steps,
                            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                            "error": "the sharded run did not finish within %d s (communicator set-up or a "
                                     "collective blocked); nothing was measured" % args.main_timeout}
                    if fallback.get("eager"):
                        # the eager window (every launch and collective enqueued one by one) was timed before the
                        # replayed one blocked: that number is a measurement of the same K sweeps and is what the line carries
                        e = fallback["eager"]
                        line.update({"value": e["value"], "ms_per_step": e["ms_per_step"], "data": fallback.get("data"),
                                     "config": fallback.get("config"), "eager_window": e,
                                     "error": "the hipGraph-replayed window did not finish within %d s; `value` is the EAGER window of the "
                                              "same %d sweeps (option sharded_graphs = 0), timed before it" % (args.main_timeout, args.steps)})
                    print(json.dumps(line), flush=True)
                sys.stderr.write("bench.py: rank %d left on the main watchdog\n" % rank)
                sys.stderr.flush()
                os._exit(3)

        threading.Thread(target=main_watchdog, daemon=True).start()

    # ---- inputs: product host side (C++), resident in HBM before timing ----
    setup, path, pairs, n, k, data = _load_workload(args.workload)
    L = int(setup.nlinks)
    V = int(setup.validation_sorted.shape[0])

    if args.cli_only:
        eng = setup.engine(use_validation_stop=False, device=local_rank)
        eng.sweep(args.warmup)
        el = _timed(eng, eng, args.steps, None, torch)
        print(json.dumps({"cli_end_to_end": cli_end_to_end(path, n, k, el / args.steps * 1e3)}), flush=True)
        os.unlink(path)
        return

    # N = 1: the plain engine (hipGraph replay).  N > 1: ONE chain, node-block sharded over the N ranks with
    # the RCCL exchanges inside the timed region -- strong scaling of the metric's own workload, whatever
    # its size (ca-AstroPh K=20 is a ~65 us sweep, four collectives per sweep cannot speed it up).
    if not multi:
        eng = setup.engine(use_validation_stop=False, device=local_rank)
        runner = eng
    else:
        runner = _Sharded(setup, rank, world, local_rank, dist)
        eng = runner.eng
    eager_window = None
    graphs_on = eng.get_option("sharded_graphs") != 0
    if multi and graphs_on:
        # N > 1, first: the SAME window with every launch and collective enqueued one by one (no capture).  RCCL under
        # hipGraph capture has never met more than one rank of this code; if the replayed window below blocks, the main
        # watchdog prints the line with this number instead of nothing.  It also pins replay == eager on the real links.
        eng.set_option("sharded_graphs", 0)
        runner.sweep(args.warmup)
        el_e = _timed(runner, eng, args.steps, dist, torch)
        c_e = eng.control()
        lam_e = eng.state()[1].copy()
        eager_window = {"value": L * args.steps / el_e, "unit": "edge-updates/s", "ms_per_step": el_e / args.steps * 1e3,
                        "what": "the same %d sweeps from the same re-seeded state, launched eagerly (option sharded_graphs = 0), "
                                "timed before the replayed window" % args.steps}
        fallback.update({"eager": eager_window, "data": data,
                         "config": {"workload": "%s: n=%d k=%d links/sweep=%d, sweeps %d..%d of the seeded run, node blocks x%d"
                                                % (args.workload, n, k, L, args.warmup, args.warmup + args.steps, world)}})
        eng.set_option("sharded_graphs", 1)
        _reseed(eng, setup)
    runner.sweep(args.warmup)
    period = max(1, min(args.event_period, args.steps // 10))   # >= 10 timed launches whenever steps >= 10
    # The timed region carries no hipEvents: svils_sweep replays whole sweeps as hipGraphs (N = 1) / the sharded
    # driver replays its phases with their collectives captured (N > 1).  Per-kernel timings come from event passes
    # of their own afterwards.
    elapsed = _timed(runner, eng, args.steps, dist, torch)
    ctrl = eng.control()
    multi_repeat = None
    if eager_window is not None:
        import numpy as np
        same = (int(ctrl.iter), int(ctrl.links_dense), int(ctrl.links_shortcut)) == (int(c_e.iter), int(c_e.links_dense), int(c_e.links_shortcut))
        eager_window["replayed_end_state_identical"] = bool(same and np.array_equal(lam_e, eng.state()[1]))
        # That first replayed window CAPTURED its graphs inside the timed region (a 64-sweep graph with its collectives is
        # milliseconds of capture + instantiation, once per handle): like the N = 1 line, `value` is the median over
        # repetitions of the same window from the re-seeded state -- the captures are behind them.
        first_replayed = elapsed
        nrep = max(1, min(args.reps, 10))
        times_m, multi_repeat = repeated_windows(eng, setup, args.warmup, args.steps, nrep, torch, runner=runner, dist=dist)
        multi_repeat["first_window_ms_per_step"] = first_replayed / args.steps * 1e3
        multi_repeat["first_window_note"] = "the first replayed window captured its hipGraphs inside the timed region"
        elapsed = float(np.median(np.asarray(times_m)))
        ctrl = eng.control()
    if main_done is not None:
        main_done.set()
    assert ctrl.sweeps_done >= args.warmup + args.steps, "sweeps were skipped"
    first_window_ms = elapsed / args.steps * 1e3

    same_window = None
    exch = None
    repeat = None
    if rank == 0 and not multi:
        # N = 1: the same window again, `--reps` times from the re-seeded state; `value` is their median
        if args.reps > 0:
            times, repeat = repeated_windows(eng, setup, args.warmup, args.steps, args.reps, torch)
            repeat["first_window_ms_per_step"] = first_window_ms
            import numpy as np
            elapsed = float(np.median(np.asarray(times)))
            ctrl = eng.control()
        # event pass of its own over the same window: every `period`-th sweep launches eagerly between hipEvents
        _reseed(eng, setup)
        eng.sweep(args.warmup)
        eng.enable_timing(1 << _svils.KERNEL_PHI, period)
        eng.sweep(args.steps)
        eng.synchronize()
        same_window = _phi_record(eng, k, "event pass over the same window: every %d-th sweep of sweeps %d..%d"
                                  % (period, args.warmup, args.warmup + args.steps))
        if same_window["launches_timed"] < 10:   # --steps below 10: top up after the window, labelled
            extra = 10 - same_window["launches_timed"]
            eng.enable_timing(1 << _svils.KERNEL_PHI, 1)
            eng.sweep(extra)
            eng.synchronize()
            more = _phi_record(eng, k, "")
            n0, n1 = same_window["launches_timed"], more["launches_timed"]
            t = (same_window["avg_launch_us"] or 0) * n0 + more["avg_launch_us"] * n1
            li = {kk: same_window["links_in_timed_launches"][kk] + more["links_in_timed_launches"][kk]
                  for kk in ("dense", "sparse", "shortcut")}
            alg = 32.0 * k * (li["dense"] + li["sparse"])
            ach = alg / (t * 1e-6) / 1e9
            same_window = {"window": same_window["window"] + " + the %d sweeps after it (to reach 10 launches)" % extra,
                           "launches_timed": n0 + n1, "avg_launch_us": t / (n0 + n1), "links_in_timed_launches": li,
                           "algorithmic_bytes_per_launch": alg / (n0 + n1), "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        eng.enable_timing(0, 1)

    if multi:
        # phi and exchange times from an event pass of its own right after the timed region (all ranks: the
        # sweeps are collective), so that the event records do not sit in the measured sweeps
        nev = max(10, min(args.steps, 20))
        eng.enable_timing((1 << _svils.KERNEL_PHI) | (1 << _svils.KERNEL_EXCHANGE), 1)
        runner.sweep(nev)
        eng.synchronize()
        if rank == 0:
            exch = (eng.timing()["exchange"][0], nev, eng.timing()["exchange"][1])
            same_window = _phi_record(eng, k, "the %d sweeps after the timed region (sweeps %d..%d), this rank's node block"
                                      % (nev, args.warmup + args.steps, args.warmup + args.steps + nev))
        eng.enable_timing(0, 1)

    rccl = n1_same_box = cpu_rec = None
    if multi:
        rccl = _rccl_evidence(eng, dist, rank, world, local_rank, torch)
        if rank == 0:
            # the SAME window on ONE GPU of this box (rank 0's), plain engine: what the N = 1 run of this box gives,
            # for the consistency check of the N > 1 value -- the other ranks wait at the barrier below
            e1 = setup.engine(use_validation_stop=False, device=local_rank)
            times1, rep1 = repeated_windows(e1, setup, args.warmup, args.steps, 10, torch)
            import numpy as np
            el1 = float(np.median(np.asarray(times1)))
            n1_same_box = {"value": L * args.steps / el1, "unit": "edge-updates/s", "ms_per_step": el1 / args.steps * 1e3,
                           "reps": 10, "what": "the same sweep window on rank 0's GPU alone (plain engine, hipGraph replay), "
                                               "median of 10 repetitions -- bench.py --gpus 1 measures exactly this"}
            e1.close()
            if not args.no_cpu_baseline:
                cpu_rec = cpu_baseline(path, pairs, n, k, args.warmup, args.steps, min_s=8.0, budget_s=15.0)
        dist.barrier()

    out = None
    if rank == 0:
        g, lam, conv = eng.state()
        out = {
            "metric": "edge-updates/sec (link-sampling SVI step)",
            "value": L * args.steps / elapsed,
            "unit": "edge-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": data,
            "config": {"workload": "%s: n=%d k=%d links/sweep=%d heldout_pairs=%d, sweeps %d..%d of the seeded run"
                                   % (args.workload, n, k, L, V, args.warmup, args.warmup + args.steps),
                       "parallelism": ("one chain, node blocks balanced by work x%d: per sweep ONE grouped launch {all-reduce of sum[k] (K doubles), "
                                       "the unscaled new rows (n*ld*8 B: an all-gather of slices padded to the largest block, or exact-count broadcasts "
                                       "where the blocks are far from equal; above 128 MB in chunks on a stream of their own, each expanded while the "
                                       "next travels)} and one all-reduce of s1,s2,s3 (3K doubles) -- two exchange points in both phases of the run; "
                                       "flags recomputed not exchanged; issued by the device library between the phases and replayed with them as "
                                       "hipGraphs (svils_sweep_sharded)" % world) if world > 1 else "single GPU",
                       "converged_nodes_at_end": int((conv > 0).sum()),
                       # how the last sweep's links were evaluated (src/linksampling.cc:622-719): full softmax,
                       # active-set path (_iter > 1000 only), O(1) shortcut of links with exactly one converged endpoint
                       "links_last_sweep": {"dense": int(ctrl.links_dense), "sparse": int(ctrl.links_sparse),
                                            "shortcut": int(ctrl.links_shortcut),
                                            "scope": "this rank's node block" if multi else "all links"}},
        }
        if multi_repeat is not None:
            repeat = multi_repeat
        if repeat is not None:
            out["repeat"] = repeat
            out["value_definition"] = ("median over %d repetitions of the timed window (each: exactly %d sweeps from the re-seeded "
                                       "state after %d warm-up sweeps, device sync on both sides, hipGraph replay, no events); "
                                       "first_window_ms_per_step is the single first window" % (repeat["reps"], args.steps, args.warmup))
        if args.test_one_gpu:
            out["test_mode"] = "all ranks on GPU 0 over the tests' transport: a code-path check, not a measurement"
        roof = {"bound": "hbm", "kernel": "k_phi_lpl (phi pass, A6)" if k <= 56 else "k_phi (phi pass, A6)"}
        if not multi:
            roof.update(_roofline_fields(same_window, k, n, args.workload))
        else:   # this rank's node block: no PMC record describes it
            roof.update(same_window)
            roof["achieved_survey_model"], roof["frac_survey_model"] = roof.pop("achieved"), roof.pop("frac")
            roof["pull_model"] = _pull_model(same_window, k, int(runner.bounds[rank + 1]) - int(runner.bounds[rank]))
            roof["traffic"], roof["frac_basis"] = None, "pull_model"
            roof["achieved"], roof["frac"] = roof["pull_model"]["achieved"], roof["pull_model"]["frac"]
        roof["timing"] = ("hipEvents around the phi launch on the engine's own stream, in an event pass of its own after the "
                          "timed region" + ("; sampled sweeps launch eagerly, the rest replay hipGraphs" if not multi else ""))
        row_bytes = eng.device_buffer(_svils.BUF_GAMMA)[2]
        resident = n * row_bytes < 200e6
        roof["note"] = ("frac = HBM bytes per phi launch (%s) / launch time / 8 TB/s.  The state of this workload (%.1f MB per n-by-k "
                        "array) is %s" % (
                            "PMC counters 2*FETCH_SIZE + WRITE_SIZE" if roof["frac_basis"] == "hbm_counter" else "pull_model, no valid PMC record",
                            n * row_bytes / 1e6,
                            "resident in the 4 MB-per-XCD L2s / 256 MB Infinity Cache: the launch is bound by the dependent-miss chain, "
                            "not by HBM (frac_survey_model is a cache-fed rate) -- the HBM-bound figures are hbm_bound and config5"
                            if resident else "larger than the 256 MB Infinity Cache"))
        out["roofline"] = roof
        if rccl is not None:
            out["rccl"] = rccl
            out["n1_same_box"] = n1_same_box
            out["speedup_vs_n1_same_box"] = out["value"] / n1_same_box["value"]
            if cpu_rec is not None:
                out["cpu_baseline"] = cpu_rec
                out["speedup_vs_cpu_1core"] = out["value"] / cpu_rec["value"]
            # N ranks must sit on N devices and RCCL must say so: otherwise `value` is not an N-GPU number and is withheld
            rccl["devices_unique"] = rccl["distinct_devices"] == world and rccl["nranks"] == [world]
            if not rccl["devices_unique"] and not args.test_one_gpu:
                out["error"] = ("the %d ranks do not sit on %d distinct devices of one communicator (distinct PCI bus ids: %d, "
                                "ncclCommCount: %s): value withheld" % (world, world, rccl["distinct_devices"], rccl["nranks"]))
                out["value_withheld"], out["value"] = out["value"], None
        if eager_window is not None:
            out["eager_window"] = eager_window
        if exch is not None:
            out["exchange"] = {"ms_per_sweep": exch[0] / exch[1], "exchange_points_per_sweep": exch[2] / exch[1],
                               "note": "hipEvent time of the RCCL collectives on the engine stream, measured in the event pass after the "
                                       "timed region (eager launches; the timed region replays hipGraphs).  Two exchange points per sweep: "
                                       "{rows + sum[k]} and {s1,s2,s3}"}
            out["load_balance"] = runner.balance
            mp = model_prediction(args.workload, "nodeblock", world)
            if mp is not None:
                out["model"] = mp
                out["model_ms_per_step"] = mp.get("predicted_ms_per_step")
                if mp.get("predicted_ms_per_step"):
                    out["measured_over_model"] = out["ms_per_step"] / mp["predicted_ms_per_step"]
        if not multi and n * k * 8 < 256e6:
            try:
                out["roofline_dense_only"] = dense_only_window(setup, k, local_rank)
            except Exception as exc:
                out["roofline_dense_only"] = {"error": repr(exc)[:200]}
        if not multi and not args.no_hbm_bound and args.workload != HBM_BOUND_WORKLOAD:
            try:
                out["hbm_bound"] = hbm_bound_record(local_rank)
            except Exception as exc:
                out["hbm_bound"] = {"error": repr(exc)[:200]}
        if not multi and not args.no_config5 and args.workload != CONFIG5_WORKLOAD:
            # BASELINE config 5 at full size on this one GPU: the size the >= 50 % HBM-roofline claim is made on
            try:
                out["config5"] = hbm_bound_record(local_rank, sweeps=5, workload=CONFIG5_WORKLOAD)
            except Exception as exc:
                out["config5"] = {"error": repr(exc)[:200]}
        if not multi and not args.no_cli and path and args.workload.startswith("astroph"):
            try:
                out["cli_end_to_end"] = cli_end_to_end(path, n, k, out["ms_per_step"])
            except Exception as exc:
                out["cli_end_to_end"] = {"error": repr(exc)[:200]}
        if not args.no_cpu_baseline and not multi:
            out["cpu_baseline"] = cpu_baseline(path, pairs, n, k, args.warmup, args.steps)
            out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
            try:
                out["cpu_baseline_allcores"] = cpu_baseline_allcores(path, pairs, n, k, args.warmup, args.steps)
            except Exception as e:      # the extra figure must never cost the contract's line
                out["cpu_baseline_allcores"] = {"value": None, "error": repr(e)}
    # The one JSON line is owed to the driver whatever happens below: a watchdog emits it (without the
    # side records) and leaves if a side measurement or the teardown ever blocks on a collective.
    import threading
    emit_lock = threading.Lock()
    state = {"emitted": False}

    def emit():
        with emit_lock:
            if rank == 0 and not state["emitted"]:
                sys.stdout.flush()
                print(json.dumps(out), flush=True)   # the one JSON line, after any RCCL banner
            state["emitted"] = True

    finished = threading.Event()
    current = {"name": None, "t0": 0.0, "todo": [], "limit": args.record_timeout or 200}   # the side record in progress (per-record watchdog)

    def watchdog():
        t_start = time.time()
        while not finished.wait(timeout=1.0):
            name = current["name"]
            hung = name is not None and time.time() - current["t0"] > current["limit"]
            if not hung and time.time() - t_start <= args.extra_timeout:
                continue
            if rank == 0:
                ex = out.setdefault("sharded_extra", {})
                if hung:
                    ex[name] = {"error": "did not finish within %d s (a collective or a kernel of this record blocked); the ranks left"
                                         % current["limit"]}
                    for later in current["todo"]:
                        ex.setdefault(later, {"error": "not run: the record %s before it hung" % name})
                else:
                    ex["watchdog"] = "side records cut off after %d s" % args.extra_timeout
            emit()
            sys.stderr.write("bench.py: rank %d left on the watchdog\n" % rank)
            sys.stderr.flush()
            os._exit(0)

    if multi:
        threading.Thread(target=watchdog, daemon=True).start()

    # N > 1 side records: the same sharded driver on the workloads SURVEY 8e expects to scale -- BASELINE
    # config 4 (ca-AstroPh K=200) and the HBM-bound size (n=2e5, k=512) -- one chain over the N ranks each.
    if multi and not args.no_extra:
        extra = {}
        if rank == 0:
            out["sharded_extra"] = extra   # filled as the records complete: the watchdog prints what is there
        plan, _ = side_record_plan(args.extra_list)
        setups = {}          # workload -> (setup, path, n, k): the two config-5 records share one host set-up
        for name, wl, wsteps, layout in plan:
            cls = {"kshard": _KSharded, "nodeblock": _Sharded, "steps": _ShardedSteps}[layout]
            wl_run = wl
            if args.test_one_gpu and wl == CONFIG5_WORKLOAD:
                # TEST MODE: N ranks share one box's host memory and one GPU -- eight 7 GB host set-ups of the full size took
                # the test box down (profiles/r07c): the flow is rehearsed on a tenth of the nodes, and the record says so
                wl_run = "mmsb:100000:512:24"
            current["todo"] = [r[0] for r in plan if r[0] != name and r[0] not in extra]
            current["limit"] = args.record_timeout or max(200, 2 * SIDE_RECORD_BUDGET_S[wl])
            current["t0"], current["name"] = time.time(), name
            try:
                if os.environ.get("BENCH_TEST_HANG_RECORD") == name:   # tests only: this record never comes back
                    while True:
                        time.sleep(1.0)
                if wl_run not in setups:
                    for old in list(setups):             # one workload resident at a time (4.1 GB of host state at config 5)
                        so, po = setups.pop(old)[:2]
                        so.close()
                        if po:
                            os.unlink(po)
                    s2, p2, _, n2, k2, _ = _load_workload(wl_run)
                    setups[wl_run] = (s2, p2, n2, k2)
                s2, p2, n2, k2 = setups[wl_run]
                r2 = cls(s2, rank, world, local_rank, dist)
                r2.sweep(3)
                if cls is not _ShardedSteps:
                    r2.sweep(wsteps)    # rehearsal: the hipGraphs of this many sweeps (either layout) are captured here, not in the timed region
                el2 = _timed(r2, r2.eng, wsteps, dist, torch)
                r2.eng.enable_timing((1 << _svils.KERNEL_PHI) | (1 << _svils.KERNEL_EXCHANGE), 1)
                nev2 = min(10, wsteps)
                r2.sweep(nev2)          # event pass after the timed sweeps (see above)
                r2.eng.synchronize()
                if rank == 0:
                    tm = r2.eng.timing()
                    per_step = int(s2.nlinks) / (cls.WINDOWS if cls is _ShardedSteps else 1)   # links a step updates, on average
                    extra[name] = {"value": per_step * wsteps / el2, "unit": "edge-updates/s", "steps": wsteps,
                                   "ms_per_step": el2 / wsteps * 1e3, "n": n2, "k": k2, "links": int(s2.nlinks),
                                   "phi_us_rank0": tm["phi"][0] / max(tm["phi"][1], 1) * 1e3,
                                   "exchange_ms_per_sweep_rank0": tm["exchange"][0] / nev2,
                                   "row_communicator": r2.eng.comm_query()["row_communicator"]}
                    if hasattr(r2, "balance"):
                        extra[name]["csr_entries_max_over_mean"] = r2.balance["max_over_mean"]
                    if wl_run != wl:
                        extra[name]["test_mode_stand_in"] = "%s instead of %s (one box's host memory cannot hold %d full-size set-ups)" % (wl_run, wl, world)
                    mp = model_prediction(wl, layout, world)
                    if mp is not None:
                        extra[name]["model"] = mp
                        extra[name]["model_ms_per_step"] = mp.get("predicted_ms_per_step")
                        if mp.get("predicted_ms_per_step"):
                            extra[name]["measured_over_model"] = extra[name]["ms_per_step"] / mp["predicted_ms_per_step"]
                    if cls is not _ShardedSteps:
                        # the same sweeps on rank 0's GPU alone (plain engine, hipGraph replay): what one GPU of THIS box does with
                        # this workload -- the other ranks wait at the barrier below
                        try:
                            e1 = s2.engine(use_validation_stop=False, device=local_rank)
                            e1.sweep(3)
                            e1.sweep(wsteps)
                            t1 = _timed(e1, e1, wsteps, None, torch)
                            e1.close()
                            extra[name]["n1_same_box_ms_per_step"] = t1 / wsteps * 1e3
                            extra[name]["speedup_vs_n1_same_box"] = t1 / el2
                        except Exception as exc1:   # (the barrier below must be reached whatever happens here)
                            extra[name]["n1_same_box_error"] = repr(exc1)[:200]
                dist.barrier()
                r2.eng.close()
            except Exception as exc:  # the main measurement must survive a failure here
                extra[name] = {"error": repr(exc)[:300]}
            current["name"] = None
        for so, po, _, _ in setups.values():
            so.close()
            if po:
                os.unlink(po)
    if path:
        os.unlink(path)
    if dist is not None:
        dist.barrier()
    emit()
    if dist is not None:
        dist.destroy_process_group()
    finished.set()


if __name__ == "__main__":
    main()
