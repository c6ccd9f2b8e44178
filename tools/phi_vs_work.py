#!/usr/bin/env python
"""phi-kernel time against the work of the sweep: per-kernel hipEvent times averaged over windows of
sweeps, next to the dense / sparse / shortcut link counts of those sweeps (ca-AstroPh K=20 by default).

  python tools/phi_vs_work.py [workload] [last_sweep] [window] [sparse_after_iter]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import _load_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "astroph-k20"
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
win = int(sys.argv[3]) if len(sys.argv) > 3 else 50
sparse_after = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
setup, _, _, n, k, _ = _load_workload(wl)
eng = setup.engine(use_validation_stop=False, sparse_after_iter=sparse_after)
L = setup.nlinks
print("# %s: %d links; columns: sweeps, dense, sparse, shortcut (mean per sweep), then us per launch" % (wl, L))
done = 0
while done < last:
    eng.enable_timing(0xff)
    eng.sweep(win)
    eng.synchronize()
    t = eng.timing()
    st = eng.sweep_stats(done, win).astype(np.float64).mean(0)
    print("%5d..%-5d dense=%7.0f sparse=%7.0f short=%7.0f | " % (done, done + win, st[0], st[1], st[2]) +
          " ".join("%s=%.1f" % (kk, v[0] / max(v[1], 1) * 1e3) for kk, v in t.items() if v[1]))
    done += win
