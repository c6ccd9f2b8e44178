#!/usr/bin/env python
"""phi-kernel time against the work of the sweep: per-kernel hipEvent times averaged over windows of
sweeps, next to the dense / sparse / shortcut link counts of those sweeps (ca-AstroPh K=20 by default).

  python tools/phi_vs_work.py [workload] [last_sweep] [window]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svinet_amd.host_api import Setup
from bench import WORKLOADS, _fixture
wl = sys.argv[1] if len(sys.argv) > 1 else "astroph-k20"
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
win = int(sys.argv[3]) if len(sys.argv) > 3 else 50
if wl not in WORKLOADS and wl.startswith("astroph-k"):
    WORKLOADS[wl] = ("ca-AstroPh.csv.gz", 17903, int(wl[len("astroph-k"):]))
f, n, k = WORKLOADS[wl]
setup = Setup(_fixture(f), n, k)
eng = setup.engine(use_validation_stop=False)
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
