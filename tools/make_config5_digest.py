#!/usr/bin/env python
"""Full-size parity pin for BASELINE config 5 (planted MMSB graph, n = 1,000,000, k = 512).

Runs the ORACLE (oracle/svinet_oracle.c, the sequential restatement of src/linksampling.cc:556-790 -- the same
code every small parity test uses, not the threaded variant) for SWEEPS sweeps from the seeded initial state on the
graph of svinet_amd/mmsbgen_sparse.py, and writes a small digest under tests/golden/config5/ that the -m gpu test
tests/test_gpu_config5.py::test_config5_full_size_against_oracle_digest compares the HIP run with:

  lambda [512][2], the column sums of gamma [512], 64 fixed gamma rows [64][512], the converged flags (count + the
  indices), active_comms histogram, the three link counters of every sweep, the likelihood rows (constructor + one per
  sweep), SHA-256 of the training-link list and of the held-out pair list (so that the test knows it is looking at the
  same graph and the same held-out set).

Run in the BUILD container (needs ~25-30 GB of RAM and ~5 min per sweep on one core):
  python tools/make_config5_digest.py [sweeps=2]              -> digest.{json,npz}: from the seeded initial state
  python tools/make_config5_digest.py --planted [sweeps=4]    -> digest_planted.{json,npz}: from planted_state() with
        _iter = 999 -- converged flags, shortcut links and the active-set branch at full size
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, K, DEG = 1_000_000, 512, 24
ROWS_SEED = 20240517


def fixed_rows(n, count=64):
    """the 64 gamma rows the digest keeps (the test uses the same function)"""
    return np.sort(np.random.default_rng(ROWS_SEED).choice(n, size=count, replace=False)).astype(np.int64)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def planted_state(pairs, truth, n, k):
    """A state NEAR the planted solution (the regime a long run ends in, which 2 sweeps from the seeded state never
    reach at this size): gamma = alpha + degree x planted membership with every third node PURE in its strongest
    community and carrying that community's converged flag, lambda from the link budget, annealing off (as it is late in
    a run).  With _iter set to 999 the first two sweeps are dense with O(1) shortcuts for the links that have exactly one
    flagged endpoint (44 % of them) and the s3 pass runs on those flags (quirk Q2); the third and fourth (_iter > 1000)
    also take the active-set branch wherever both endpoints have fewer than K/10 active communities
    (src/linksampling.cc:622-681).  A pure function of the generator's output; the test recomputes it.
    -> gamma [n][k], lambda [k][2], converged [n]"""
    comm, w, _ = truth
    # the generator's memberships are four comparable components per node (Dirichlet(0.05) over 512 columns, truncated to
    # its top 4), which never collapse to one: every third node is made PURE in its strongest community here
    w = w.copy()
    pure = (np.arange(n) % 3) == 0
    w[pure, 0] = 1.0
    w[pure, 1:] = 0.0
    deg = np.bincount(pairs.ravel(), minlength=n).astype(np.float64)
    g = np.full((n, k), 1.0 / k)
    rows = np.repeat(np.arange(n), comm.shape[1])
    np.add.at(g, (rows, comm.ravel()), (deg[:, None] * w).ravel())
    lam = np.empty((k, 2))
    lam[:, 0] = 1.0 + 2.0 * pairs.shape[0] / k
    lam[:, 1] = 1.0 + 20.0 * pairs.shape[0] / k
    # ... and those nodes carry the converged flag of that community from the start (the sticky _converged[] of
    # src/linksampling.cc:455-475, handed over like -load would have to): left to itself the model spreads a pure node
    # over its neighbours' communities again within one sweep, and no flag would ever be set on this graph
    conv = np.zeros(n, dtype=np.uint32)
    conv[pure] = comm[pure, 0].astype(np.uint32) + 1
    return g, lam, conv


def main():
    planted = "--planted" in sys.argv
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    sweeps = int(nums[0]) if nums else (4 if planted else 2)
    from oracle import oracle as O
    from svinet_amd import mmsbgen_sparse as G
    t0 = time.time()
    pairs, truth = G.generate(N, K, DEG, return_truth=True)
    net = O.Network(n=N, pairs=pairs)
    ref = O.LinkSampling(net, K, use_validation_stop=False)
    g0 = None
    if planted:
        g0, lam0, conv0 = planted_state(pairs, truth, N, K)
        ref.set_gamma(g0)
        ref.set_lambda(lam0)
        ref.set_converged(conv0)
        ref.refresh()
        ref.annealing = False   # (while annealing, gammanext *= ones / sum[k] makes EVERY node active in every column of below-half-average mass)
        ref.iter = 999      # two dense sweeps (the first one's prune() sets the flags, the second takes the shortcuts), then _iter > 1000
    print("graph + constructor: %.0f s, %d training links" % (time.time() - t0, ref.nlinks), flush=True)
    counts = []
    for i in range(sweeps):
        t1 = time.time()
        ref.sweep()
        counts.append([int(x) for x in ref.link_counts()])
        print("sweep %d: %.0f s, links dense/sparse/shortcut %r" % (i, time.time() - t1, counts[-1]), flush=True)
    g = ref.gamma
    conv = ref.converged
    rows_idx = fixed_rows(N)
    out = os.path.join(ROOT, "tests", "golden", "config5")
    os.makedirs(out, exist_ok=True)
    stem = "digest_planted" if planted else "digest"
    np.savez_compressed(os.path.join(out, stem + ".npz"), lam=ref.lam, gamma_colsum=g.sum(0), gamma_rows=g[rows_idx],
                        rows_idx=rows_idx, converged_idx=np.flatnonzero(conv).astype(np.uint32),
                        converged_val=conv[conv > 0], active_hist=np.bincount(ref.active_comms, minlength=K + 1),
                        link_counts=np.asarray(counts, dtype=np.int64), likelihood_rows=ref.rows,
                        gamma_rowsum_minmax=np.asarray([g.sum(1).min(), g.sum(1).max()]))
    meta = {"n": N, "k": K, "mean_degree": DEG, "sweeps": sweeps, "nlinks": int(ref.nlinks),
            "nvalidation": int(ref.validation_sorted.shape[0]), "links_sha256": sha(ref.links),
            "validation_sha256": sha(ref.validation_sorted), "total_pairs_uint32": ref.total_pairs, "ones_prob": ref.ones_prob,
            "generator": "svinet_amd/mmsbgen_sparse.py seed %d" % G.DEFAULT_SEED,
            "oracle": "oracle/svinet_oracle.c, sequential sweep (orc_ls_sweep)", "made_by": "tools/make_config5_digest.py",
            "wall_s": round(time.time() - t0)}
    if planted:
        meta.update({"start": "planted_state() of this script (gamma, lambda, converged flags), _iter = 999, annealing off", "gamma0_sha256": sha(g0), "iter0": 999})
    json.dump(meta, open(os.path.join(out, stem + ".json"), "w"), indent=1)
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
