#!/usr/bin/env python
"""time mini-batch steps (svils_step) for one workload: tools/step_times.py mmsb:200000:512:24 20 [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svinet_amd.host_api import Setup
from svinet_amd import mmsbgen_sparse
_, sn, sk, sd = sys.argv[1].split(":")
n, k = int(sn), int(sk)
nb = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3 * nb
pairs = mmsbgen_sparse.generate(n, k, int(sd))
perm = np.random.default_rng(5).permutation(n).astype(np.int32)
p2 = np.sort(perm[pairs], axis=1)
p2 = p2[np.lexsort((p2[:, 1], p2[:, 0]))]
s = Setup(n=n, k=k, pairs=p2)
full = s.engine(use_validation_stop=False)
full.sweep(3); full.synchronize()
t0 = time.perf_counter(); full.sweep(5); full.synchronize(); tf = (time.perf_counter() - t0) / 5
e = s.engine(use_validation_stop=False, reportfreq=nb)
e.set_stochastic(batch_nodes=(n + nb - 1) // nb, tau0=1.0, kappa=0.5)
e.step(nb); e.synchronize()
t0 = time.perf_counter(); e.step(steps); e.synchronize(); ts = (time.perf_counter() - t0) / steps
L = int(s.nlinks)
print("%s: full sweep %.3f ms (%.3g edge-updates/s); %d windows: %.3f ms/step, %.3f ms per pass over the nodes (%.3g link evaluations/s)"
      % (sys.argv[1], tf * 1e3, L / tf, nb, ts * 1e3, ts * nb * 1e3, L / (ts * nb)))
