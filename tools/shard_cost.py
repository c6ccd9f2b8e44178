#!/usr/bin/env python
"""Multi-GPU cost model inputs, MEASURED on ONE GPU: what rank 0 of G computes per sweep in the two sharded layouts
(all phases launched back to back, no exchange -- the exchange buffers keep whatever they hold, so this is timing
only), next to the plain engine, plus the bytes each layout exchanges per sweep and a predicted sweep time on
G = 2, 4, 8 GPUs from a stated link model.  Nothing here is a multi-GPU measurement.

  python tools/shard_cost.py [workload] [G,G,...] [--json FILE]

--json FILE merges this workload's rows into FILE (profiles/shard_cost_model.json: what bench.py prints as `model` beside every
N > 1 record -- key "<workload>|<nodeblock or kshard>|<G>").

Link model (MI355X_MICROARCH.md: 7 xGMI links x ~153 GB/s per GPU, point to point): a rank sends / receives over all
its links at once at EFF = 300 GB/s aggregate when the collective uses them all (all-gather / broadcast of node blocks
with G = 8: every peer is one hop away), EFF * (G - 1) / 7 with fewer peers; every collective costs LAT = 20 us on top.
  all-gather of S bytes in total : S (G - 1) / G received per rank
  all-reduce of M bytes          : 2 M (G - 1) / G moved per rank (reduce-scatter + all-gather)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import _load_workload
from svinet_amd import _svils
from svinet_amd.sharded import balanced_bounds, equal_bounds

import json
argv = list(sys.argv[1:])
json_out = None
if "--json" in argv:
    i = argv.index("--json")
    json_out = argv[i + 1]
    del argv[i:i + 2]
wl = argv[0] if len(argv) > 0 else "astroph-k200"
Gs = [int(x) for x in argv[1].split(",")] if len(argv) > 1 else [2, 4, 8]
rows = {}
EFF, LAT = 300e9, 20e-6
setup, _, _, n, k, _ = _load_workload(wl)
L, V = int(setup.nlinks), int(setup.validation_sorted.shape[0])
ld = (k + 15) // 16 * 16 if k > 56 else (k + 1) // 2 * 2     # the engine's row stride (packed rows at K <= 56)
steps = 20 if n * k < 5e7 else 6


def wall(fn, reps):
    fn(2)
    t0 = time.perf_counter()
    fn(reps)
    return (time.perf_counter() - t0) / reps * 1e3


def link_time(bytes_per_rank, G, ncoll):
    bw = EFF * min(1.0, (G - 1) / 7.0)
    return (bytes_per_rank / bw + ncoll * LAT) * 1e3


plain = setup.engine(use_validation_stop=False)
def run_plain(s):
    plain.sweep(s); plain.synchronize()
t_plain = wall(run_plain, steps)
plain.close()
print("# %s: n=%d k=%d links=%d held-out pairs=%d; plain engine on one GPU: %.3f ms per sweep" % (wl, n, k, L, V, t_plain))
print("# layout      G   compute/rank (ms)  of which expand   bytes exchanged per sweep (total)         collectives  links (ms)  predicted ms/sweep  speed-up vs 1 GPU")
for G in Gs:
    # ---- node blocks, balanced by work (svils_balance_node_blocks): every rank of G on small state, ranks 0 and G-1 on large
    bounds = balanced_bounds(setup.links, n, G)
    deg = np.bincount(np.asarray(setup.links).ravel(), minlength=n)
    ent = np.array([deg[int(bounds[r]):int(bounds[r + 1])].sum() for r in range(G)], dtype=np.float64)
    eq = equal_bounds(n, G)
    ent_eq = np.array([deg[int(eq[r]):int(eq[r + 1])].sum() for r in range(G)], dtype=np.float64)
    ranks = list(range(G)) if n * k < 5e7 else [0, G - 1]
    t_rank, t_exp = [], 0.0
    for r in ranks:
        e = setup.engine(use_validation_stop=False, node_block=(int(bounds[r]), int(bounds[r + 1])))
        e.set_node_blocks(r, G, bounds)
        def run_nb(s):
            for _ in range(s):
                for ph in (_svils.PHASE_A, _svils.PHASE_B_LIGHT, _svils.PHASE_EXPAND_ALL, _svils.PHASE_C, _svils.PHASE_D):
                    e.sweep_phase(ph)
            e.synchronize()
        t_rank.append(wall(run_nb, steps))
        if r == 0:
            def run_exp(s):
                for _ in range(s):
                    e.sweep_phase(_svils.PHASE_EXPAND_ALL)
                e.synchronize()
            t_exp = wall(run_exp, steps)
        e.close()
    t_nb = max(t_rank)
    bmax = int(np.max(np.diff(bounds.astype(np.int64))))
    # the staged rows: slices padded to the largest block (all-gather) while G * bmax <= 1.5 n, else the exact rows as G
    # grouped broadcasts (exchange_rows_and_expand, svils_api.hip)
    rows_total = (G * bmax if 2 * G * bmax <= 3 * n else n) * ld * 8
    kvec = 4 * k * 8
    recv = rows_total * (G - 1) / G + 2 * kvec * (G - 1) / G
    t_link = link_time(recv, G, 2)
    # the row exchange is pipelined against the expansion of the rows already there (large payloads): what stays exposed is the longer of the two
    t_pred = (t_nb - t_exp) + max(t_exp, link_time(rows_total * (G - 1) / G, G, 1)) + link_time(2 * kvec * (G - 1) / G, G, 1)
    print("node-block  %2d   %10.3f        %8.3f          %7.1f MB all-gather + %5.1f KB all-reduce   %6d    %8.3f    %10.3f        %6.2fx"
          % (G, t_nb, t_exp, rows_total / 1e6, kvec / 1e3, 2, t_link, t_pred, t_plain / t_pred))
    rows["%s|nodeblock|%d" % (wl, G)] = {
        "predicted_ms_per_step": t_pred, "compute_ms_per_rank": t_nb, "of_which_expand_ms": t_exp, "plain_engine_ms_per_step": t_plain,
        "predicted_speedup_vs_one_gpu": t_plain / t_pred, "bytes_exchanged_per_step": rows_total + 2 * kvec, "collectives_per_step": 2,
        "exposed_link_ms": max(t_exp, link_time(rows_total * (G - 1) / G, G, 1)) - t_exp + link_time(2 * kvec * (G - 1) / G, G, 1),
        "csr_entries_max_over_mean": float(ent.max() / ent.mean()),
        "how": "max over the measured ranks of {phi, light finalise, expand-all, s3, tail} launched back to back on ONE GPU with no exchange; "
               "row exchange pipelined against the expansion (the longer of the two stays exposed) + one K-vector all-reduce"}
    print("#   balance: CSR entries per rank / mean = %s (equal-count blocks: %s); compute per measured rank (ms): %s"
          % (" ".join("%.2f" % x for x in ent / ent.mean()), " ".join("%.2f" % x for x in ent_eq / ent_eq.mean()),
             " ".join("r%d %.3f" % (r, t) for r, t in zip(ranks, t_rank))))
    # ---- K-sharded: rank 0 of G
    k0, k1 = 0, k // G
    ks = _svils.Engine(n, k, ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta, link_thresh=setup.link_thresh,
                       lt_min_deg=setup.lt_min_deg, use_validation_stop=False, k_slice=(k0, k1))
    ks.set_graph(setup.links); ks.set_validation(setup.validation_sorted)
    if setup.host_gamma:
        ks.set_state(np.ascontiguousarray(setup.gamma[:, k0:k1]), np.ascontiguousarray(setup.lam[k0:k1]))
    else:
        setup.device_init(ks, lam=np.ascontiguousarray(setup.lam[k0:k1]))
    ks.ksh_init_state()
    log = ks.ksh_log_domain() == 1
    def run_ks(s):
        for _ in range(s):
            if log:
                ks.ksweep_phase(_svils.KPHASE_DENMAX)
            for ph in range(5):
                ks.ksweep_phase(ph)
        ks.synchronize()
    t_ks = wall(run_ks, steps)
    ks.close()
    ar = (L * (2 if log else 1) + 3 * n + k + V) * 8
    ncoll = 5 if log else 4
    t_link = link_time(2 * ar * (G - 1) / G, G, ncoll)
    t_pred = t_ks + t_link
    print("K-sharded   %2d   %10.3f        %8s          %7.1f MB all-reduce (den %s rowx q2 vdot)          %6d    %8.3f    %10.3f        %6.2fx"
          % (G, t_ks, "-", ar / 1e6, "+dmax" if log else "", ncoll, t_link, t_pred, t_plain / t_pred))
    rows["%s|kshard|%d" % (wl, G)] = {
        "predicted_ms_per_step": t_pred, "compute_ms_per_rank": t_ks, "plain_engine_ms_per_step": t_plain,
        "predicted_speedup_vs_one_gpu": t_plain / t_pred, "bytes_exchanged_per_step": ar, "collectives_per_step": ncoll,
        "exposed_link_ms": t_link,
        "how": "rank 0 of G: every phase of a K-sharded sweep launched back to back on ONE GPU with no exchange; + %d all-reduces "
               "(reduce-scatter + all-gather) of the coupling buffers, not overlapped" % ncoll}
if json_out:
    doc = {"rows": {}}
    if os.path.exists(json_out):
        try:
            doc = json.load(open(json_out))
        except Exception:
            pass
    doc.setdefault("rows", {}).update(rows)
    doc["link_model"] = {"eff_GBps": EFF / 1e9, "lat_us": LAT * 1e6,
                         "what": "a rank moves EFF aggregate over its xGMI links when a collective uses all 7 of them, EFF x (G-1)/7 with fewer "
                                 "peers; every collective costs LAT on top; all-gather of S: S (G-1)/G received; all-reduce of M: 2 M (G-1)/G moved. "
                                 "ASSUMPTIONS, nothing on real links has been measured"}
    doc["source"] = "tools/shard_cost.py (per-rank compute measured on one MI355X, commit %s)" % os.environ.get("EVIDENCE_COMMIT", "?")
    with open(json_out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
