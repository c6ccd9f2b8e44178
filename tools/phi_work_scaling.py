#!/usr/bin/env python
"""Is the small-K phi pass work-proportional where it is a throughput problem?  A synthetic graph large enough
that every resident wave has many wave-items (n = 1e6, K = 20, mean degree 24), with a chosen fraction of the nodes
flagged converged before the sweep: a link with exactly one converged endpoint takes the O(1) shortcut
(src/linksampling.cc:622-631), so the softmax-link count falls as 1 - 2f(1-f).  One sweep per setting from the same
state (flags are sticky, so each setting starts from a fresh state); per-kernel hipEvent times.

  python tools/phi_work_scaling.py [n] [k] [deg]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svinet_amd.host_api import Setup
from bench import _synthetic_pairs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
deg = int(sys.argv[3]) if len(sys.argv) > 3 else 24
setup = Setup(n=n, k=k, pairs=_synthetic_pairs(n, deg, 20240517))
eng = setup.engine(use_validation_stop=False)
eng.sweep(2); eng.synchronize()
g0, l0, _ = eng.state()
rng = np.random.default_rng(1)
print("# synthetic n=%d k=%d links=%d; fraction of nodes flagged converged -> links by branch, us per launch" % (n, k, setup.nlinks))
for f in (0.0, 0.1, 0.25, 0.5):
    conv = np.zeros(n, dtype=np.uint32)
    idx = rng.random(n) < f
    conv[idx] = rng.integers(1, k + 1, size=int(idx.sum()), dtype=np.uint32)
    eng.set_state(g0, l0, conv)
    eng.sweep(1); eng.synchronize()            # classification of the forced flags happens here
    eng.set_state(g0, l0, conv)
    eng.enable_timing(0xff)
    eng.sweep(1); eng.synchronize()
    t = eng.timing()
    st = eng.sweep_stats(int(eng.control().sweeps_done) - 1, 1)[0]
    print("f=%.2f dense=%8d shortcut=%8d | " % (f, st[0], st[2]) +
          " ".join("%s=%.1f" % (kk, v[0] / max(v[1], 1) * 1e3) for kk, v in t.items() if v[1]))
