#!/usr/bin/env python
"""Where a sweep's launches spend their time: wall-clock stamps (100 MHz) that thread 0 of every block
records at phase boundaries, from a -DSVILS_STAMPS build (python -m svinet_amd.build --stamps; run with
SVILS_LIB=svinet_amd/lib/libsvils_stamps.so).  Prints, per kernel, the stamps of the LAST eager sweep
relative to the earliest block start of that kernel: min / median / max over blocks, in microseconds.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svinet_amd.host_api import Setup
from bench import WORKLOADS, _fixture
wl = sys.argv[1] if len(sys.argv) > 1 else "astroph-k20"
nsweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f, n, k = WORKLOADS[wl]
setup = Setup(_fixture(f), n, k)
eng = setup.engine(use_validation_stop=False)
eng.sweep(nsweeps); eng.synchronize()
eng.sweep(1); eng.synchronize()        # eager (short call): the stamps of this sweep stay in the buffer
import ctypes as C
from svinet_amd import _svils
out = np.zeros(4 * 1024 * 8, dtype=np.uint64)
_svils._chk(_svils.load().svils_get_aux(eng._h, 5, out.ctypes.data))
st = out.reshape(4, 1024, 8).astype(np.float64)
names = ["phi", "finalize", "s3(+count role: slots 0,7)", "tail(+scatter role: slots 0,7)"]
t00 = None
for kk in range(4):
    a = st[kk]
    used = a[:, 0] > 0
    if not used.any():
        continue
    t0 = a[used, 0].min()
    if t00 is None:
        t00 = t0
    print("%s: %d blocks, first block starts at +%.2f us of the sweep" % (names[kk], used.sum(), (t0 - t00) / 100.0))
    for s in range(8):
        col = a[used, s]
        col = col[col > 0]
        if col.size:
            r = (col - t0) / 100.0
            print("   slot %d: n=%4d  min %7.2f  med %7.2f  max %7.2f us" % (s, col.size, r.min(), np.median(r), r.max()))
