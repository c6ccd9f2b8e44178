import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from svinet_amd.host_api import Setup
from bench import _fixture
s = Setup(_fixture("ca-AstroPh.csv.gz"), 17903, 20)
e = s.engine(use_validation_stop=False)
e.sweep(50); e.synchronize()
t0 = time.perf_counter(); e.sweep(1000); t1 = time.perf_counter(); e.synchronize(); t2 = time.perf_counter()
print("enqueue 1000 sweeps: %.1f ms (%.1f us/sweep host), total %.1f ms (%.1f us/sweep)" % ((t1-t0)*1e3, (t1-t0)*1e3, (t2-t0)*1e3, (t2-t0)*1e3))
