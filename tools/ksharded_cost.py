#!/usr/bin/env python
"""What the K-sharded form costs on ONE GPU (a single rank holding every column: all exchanges are no-ops),
next to the plain engine, and what one rank of G would run (a column slice of width k/G on the full graph:
virtual rank 0 alone, its exchange buffers left as they are -- timing only, the numbers are not a model).

  python tools/ksharded_cost.py [workload] [G]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import _load_workload
from svinet_amd import _svils
wl = sys.argv[1] if len(sys.argv) > 1 else "astroph-k200"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
setup, _, _, n, k, _ = _load_workload(wl)


def kengine(k0, k1):
    e = _svils.Engine(n, k, ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta, link_thresh=setup.link_thresh,
                      lt_min_deg=setup.lt_min_deg, use_validation_stop=False, k_slice=(k0, k1))
    e.set_graph(setup.links); e.set_validation(setup.validation_sorted)
    e.set_state(np.ascontiguousarray(setup.gamma[:, k0:k1]), np.ascontiguousarray(setup.lam[k0:k1]))
    return e


def timed(fn, steps):
    fn(3)
    t0 = time.perf_counter(); fn(steps); dt = time.perf_counter() - t0
    return dt / steps * 1e3


steps = 20
plain = setup.engine(use_validation_stop=False)
def run_plain(s): plain.sweep(s); plain.synchronize()
print("%s: plain engine                       %.3f ms per sweep" % (wl, timed(run_plain, steps)))
full = kengine(0, k)
full.ksh_init_state()
def run_full(s): full.sweep_ksharded(s); full.synchronize()
print("%s: K-sharded, one rank, all %d columns  %.3f ms per sweep" % (wl, k, timed(run_full, steps)))
w = k // G
part = kengine(0, w)
part.ksh_init_state()
import ctypes
def run_part(s):
    for _ in range(s):
        for ph in range(5):
            part.ksweep_phase(ph)
    part.synchronize()
try:
    print("%s: one rank of %d (columns 0..%d of %d)   %.3f ms per sweep of compute (exchanges not included)" % (wl, G, w, k, timed(run_part, steps)))
except Exception as exc:
    print("one-rank-of-%d timing: %r" % (G, exc))
