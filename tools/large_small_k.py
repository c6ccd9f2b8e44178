#!/usr/bin/env python
"""Small K on a large graph (synthetic, mean degree 24): per-kernel hipEvent times of sweeps 3..13 and the
graph-replay time per sweep.   python tools/large_small_k.py <n> <k>"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from svinet_amd.host_api import Setup
from bench import _synthetic_pairs
n,k=int(sys.argv[1]),int(sys.argv[2])
setup = Setup(n=n, k=k, pairs=_synthetic_pairs(n, 24, 20240517))
eng = setup.engine(use_validation_stop=False)
eng.sweep(3); eng.synchronize()
eng.enable_timing(0xff)
eng.sweep(10); eng.synchronize()
t = eng.timing()
print("n=%d k=%d" % (n, k), " ".join("%s=%.1f" % (kk, v[0] / max(v[1], 1) * 1e3) for kk, v in t.items() if v[1]))
import time
t0=time.perf_counter(); eng.sweep(20); eng.synchronize(); print("graph-replay sweep %.1f us" % ((time.perf_counter()-t0)/20*1e6))
