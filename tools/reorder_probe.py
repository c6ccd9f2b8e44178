#!/usr/bin/env python
"""Does a locality-improving node order pay on the HBM-bound sizes?  The same planted MMSB graph with its
generator's random ids, with ids sorted by the planted dominant community, and with a plain BFS order
(what a host could compute without ground truth); per-kernel hipEvent times of sweeps 3..13.

  python tools/reorder_probe.py [n] [k] [mean_deg]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svinet_amd import mmsbgen_sparse
from svinet_amd.host_api import Setup
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 512
deg = int(sys.argv[3]) if len(sys.argv) > 3 else 24
pairs, (comm, w, beta) = mmsbgen_sparse.generate(n, k, deg, return_truth=True)


def relabel(pairs, order):
    new = np.empty(n, dtype=np.int64)
    new[order] = np.arange(n)
    a, b = new[pairs[:, 0]], new[pairs[:, 1]]
    key = np.unique(np.minimum(a, b) * n + np.maximum(a, b))
    return np.stack([key // n, key % n], axis=1).astype(np.int32)


def bfs_order(pairs):
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    m = sp.coo_matrix((np.ones(len(pairs), dtype=np.int8), (pairs[:, 0], pairs[:, 1])), shape=(n, n)).tocsr()
    m = m + m.T
    return np.asarray(reverse_cuthill_mckee(m, symmetric_mode=True), dtype=np.int64)


def run(tag, pr):
    s = Setup(n=n, k=k, pairs=pr)
    e = s.engine(use_validation_stop=False)
    e.sweep(3); e.synchronize()
    e.enable_timing(0xff)
    e.sweep(10); e.synchronize()
    t = e.timing()
    print(tag, " ".join("%s=%.1fus" % (kk, v[0] / max(v[1], 1) * 1e3) for kk, v in t.items() if v[1]),
          "sum=%.2fms" % (sum(v[0] for v in t.values()) / 10), flush=True)
    e.close()


run("generator ids      ", pairs)
run("by planted community", relabel(pairs, np.argsort(comm[:, 0], kind="stable")))
t0 = time.perf_counter()
o = bfs_order(pairs)
print("rcm order computed in %.1f s" % (time.perf_counter() - t0))
run("reverse Cuthill-McKee", relabel(pairs, o))
