#!/usr/bin/env python
"""per-kernel hipEvent timings of the sweep for one workload (A/B tool; SVILS_LIB selects a build)"""
import sys, os, gzip, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svinet_amd.host_api import Setup
from bench import WORKLOADS, _fixture, _synthetic_pairs
wl = sys.argv[1] if len(sys.argv) > 1 else "astroph-k20"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
if wl.startswith("synthetic"):
    _, sn, sk, sd = wl.split(":")
    setup = Setup(n=int(sn), k=int(sk), pairs=_synthetic_pairs(int(sn), int(sd), 20240517))
elif wl.startswith("mmsb"):
    from svinet_amd import mmsbgen_sparse
    _, sn, sk, sd = wl.split(":")
    setup = Setup(n=int(sn), k=int(sk), pairs=mmsbgen_sparse.generate(int(sn), int(sk), int(sd)))
else:
    if wl not in WORKLOADS and wl.startswith("astroph-k"):
        WORKLOADS[wl] = ("ca-AstroPh.csv.gz", 17903, int(wl[len("astroph-k"):]))
    f, n, k = WORKLOADS[wl]
    setup = Setup(_fixture(f), n, k)
eng = setup.engine(use_validation_stop=False)
eng.sweep(5); eng.synchronize()
eng.enable_timing(0xff)
eng.sweep(steps); eng.synchronize()
t = eng.timing()
tot = sum(v[0] for v in t.values())
print(os.environ.get("SVILS_LIB", "default"), wl, " ".join("%s=%.1fus" % (k, v[0] / max(v[1], 1) * 1e3) for k, v in t.items()), "sum=%.1fus" % (tot / steps * 1e3))
