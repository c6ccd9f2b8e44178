#!/usr/bin/env python
"""ONE rank of a sharded run on ONE GPU, phases launched back to back with no exchange (the exchange buffers keep whatever
they hold: timing and traffic only, like tools/shard_cost.py) -- the command rocprofv3 wraps to get the per-kernel
durations (--kernel-trace --stats) and HBM bytes (--pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the kernels an
N-GPU run executes on each of its ranks.

  python tools/shard_rank.py <workload> <kshard|nodeblock> <G> <rank> [sweeps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import _load_workload
from svinet_amd import _svils
from svinet_amd.sharded import balanced_bounds

wl, layout, G, r = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
sweeps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
setup, _, _, n, k, _ = _load_workload(wl)
if layout == "kshard":
    k0, k1 = k * r // G, k * (r + 1) // G
    e = _svils.Engine(n, k, ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta, link_thresh=setup.link_thresh,
                      lt_min_deg=setup.lt_min_deg, use_validation_stop=False, k_slice=(k0, k1))
    e.set_graph(setup.links); e.set_validation(setup.validation_sorted)
    if setup.host_gamma:
        e.set_state(np.ascontiguousarray(setup.gamma[:, k0:k1]), np.ascontiguousarray(setup.lam[k0:k1]))
    else:
        setup.device_init(e, lam=np.ascontiguousarray(setup.lam[k0:k1]))
    e.ksh_init_state()
    log = e.ksh_log_domain() == 1
    phases = ([_svils.KPHASE_DENMAX] if log else []) + list(range(5))
    def sweep():
        for ph in phases:
            e.ksweep_phase(ph)
else:
    bounds = balanced_bounds(setup.links, n, G)
    e = setup.engine(use_validation_stop=False, node_block=(int(bounds[r]), int(bounds[r + 1])))
    e.set_node_blocks(r, G, bounds)
    def sweep():
        for ph in (_svils.PHASE_A, _svils.PHASE_B_LIGHT, _svils.PHASE_EXPAND_ALL, _svils.PHASE_C, _svils.PHASE_D):
            e.sweep_phase(ph)
for _ in range(2):
    sweep()
e.synchronize()
import time
t0 = time.perf_counter()
for _ in range(sweeps):
    sweep()
e.synchronize()
print("%s %s rank %d of %d: %.3f ms per sweep (compute only, %d sweeps)" % (wl, layout, r, G, (time.perf_counter() - t0) / sweeps * 1e3, sweeps))
