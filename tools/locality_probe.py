#!/usr/bin/env python
"""Does a locality-aware node numbering pay on config 5?  (DESIGN.md "next": every XCD gathers neighbour rows from the whole
n-by-k table; a numbering that puts a node's neighbours near it would let the L2s / Infinity Cache serve them.)

  python tools/locality_probe.py make  [n] [k]        # build container: orderings -> tools/scratch_perm/*.npy
  python tools/locality_probe.py run   [n] [k]        # GPU box: sweep / per-kernel times per ordering, plain engine and K-shard rank 0 of 8

The graph is relabelled OUTSIDE the library (a different input with the same structure): timing only.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "scratch_perm")   # (created by `make`; .npy files of 4 MB each travel to the GPU box with the snapshot: delete them afterwards)
os.makedirs(D, exist_ok=True)
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 512
from svinet_amd import mmsbgen_sparse as G


def relabel(pairs, order):
    """order[i] = old id of the node that becomes new id i; pairs listed so that first appearance ~ the new numbering"""
    new = np.empty(n, dtype=np.int64)
    new[order] = np.arange(n)
    p = new[np.asarray(pairs, dtype=np.int64)]
    lo, hi = p.min(1), p.max(1)
    o = np.lexsort((lo, hi))
    return np.stack([lo[o], hi[o]], 1).astype(np.int32)


if mode == "make":
    pairs, truth = G.generate(n, k, 24, return_truth=True)
    comm, w = truth[0], truth[1]
    dom = comm[np.arange(n), np.argmax(w, 1)]
    np.save(os.path.join(D, "planted_%d_%d.npy" % (n, k)), np.argsort(dom, kind="stable").astype(np.int32))
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    P = np.asarray(pairs, dtype=np.int64)
    A = sp.coo_matrix((np.ones(len(P), dtype=np.int8), (P[:, 0], P[:, 1])), shape=(n, n)).tocsr()
    A = (A + A.T).tocsr()
    t0 = time.time()
    rcm = reverse_cuthill_mckee(A, symmetric_mode=True)
    print("rcm %.1f s" % (time.time() - t0))
    np.save(os.path.join(D, "rcm_%d_%d.npy" % (n, k)), np.asarray(rcm, dtype=np.int32))
    # label propagation, asynchronous in two halves, 6 rounds; then nodes sorted by label
    t0 = time.time()
    rng = np.random.default_rng(1)
    lab = np.arange(n, dtype=np.int64)
    src = np.concatenate([P[:, 0], P[:, 1]])
    dst = np.concatenate([P[:, 1], P[:, 0]])
    for it in range(8):
        half = rng.random(n) < 0.5
        key = src * n + lab[dst]
        u, c = np.unique(key, return_counts=True)
        node, l = u // n, u % n
        o = np.lexsort((l, -c, node))
        node, l = node[o], l[o]
        first = np.concatenate([[True], node[1:] != node[:-1]])
        best = lab.copy()
        best[node[first]] = l[first]
        lab = np.where(half, best, lab)
        print("lpa round %d: %d labels, %.1f s" % (it, len(np.unique(lab)), time.time() - t0))
    np.save(os.path.join(D, "lpa_%d_%d.npy" % (n, k)), np.argsort(lab, kind="stable").astype(np.int32))
    sys.exit(0)

from svinet_amd import _svils
from svinet_amd.host_api import Setup
pairs = G.generate(n, k, 24)
for name in ("orig", "planted", "lpa", "rcm"):
    if name == "orig":
        pr = pairs
    else:
        f = os.path.join(D, "%s_%d_%d.npy" % (name, n, k))
        if not os.path.exists(f):
            continue
        pr = relabel(pairs, np.load(f).astype(np.int64))
    s = Setup(n=n, k=k, pairs=pr)
    L = int(s.nlinks)
    lk = np.asarray(s.links, dtype=np.int64)
    span = float(np.median(np.abs(lk[:, 0] - lk[:, 1])))
    e = s.engine(use_validation_stop=False)
    e.sweep(2); e.synchronize()
    e.enable_timing(0xff, 1)
    e.sweep(5); e.synchronize()
    tm = e.timing()
    plain = {kk: v[0] / max(v[1], 1) for kk, v in tm.items() if v[1]}
    e.close()
    G8, r = 8, 0
    k0, k1 = k * r // G8, k * (r + 1) // G8
    ks = _svils.Engine(n, k, ones=s.ones, ones_prob=s.ones_prob, eta=s.eta, link_thresh=s.link_thresh, lt_min_deg=s.lt_min_deg,
                       use_validation_stop=False, k_slice=(k0, k1))
    ks.set_graph(s.links); ks.set_validation(s.validation_sorted)
    ks.set_state(np.ascontiguousarray(s.gamma[:, k0:k1]), np.ascontiguousarray(s.lam[k0:k1]))
    ks.ksh_init_state()
    def phases(m):
        for _ in range(m):
            for ph in range(5):
                ks.ksweep_phase(ph)
        ks.synchronize()
    phases(2)
    per = []
    for ph in range(5):
        ks.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ks.ksweep_phase(ph)
        ks.synchronize()
        per.append((time.perf_counter() - t0) / 5 * 1e3)
    t0 = time.perf_counter(); phases(5); tks = (time.perf_counter() - t0) / 5 * 1e3
    ks.close()
    print("%-8s links %d median |p-q| %.0f | plain ms: %s sum %.2f | kshard r0/8 ms/sweep %.3f phases DEN %.3f PHI+fin1 %.3f FIN+s3 %.3f LAM %.3f STOP %.3f"
          % (name, L, span, " ".join("%s=%.2f" % kv for kv in plain.items()), sum(plain.values()), tks, *per), flush=True)
    s.close()
