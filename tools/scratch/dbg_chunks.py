import os, sys, gzip
sys.path.insert(0, os.getcwd())
import numpy as np
from svinet_amd import _svils
from svinet_amd.host_api import Setup
src='tests/golden/graphs/LFR-network-n1000-k28.txt.gz'
open('/tmp/lfr.txt','wb').write(gzip.open(src,'rb').read())
setup = Setup('/tmp/lfr.txt', 1000, 28)
plain = setup.engine(use_validation_stop=False); plain.sweep(5); pg = plain.state()[0]
for ch in (1, 3):
    os.environ['SVILS_XCHUNKS'] = str(ch)
    eng = setup.engine(use_validation_stop=False, node_block=(0, 1000), n_alloc=1000)
    eng.comm_init(_svils.comm_unique_id(), 0, 1)
    eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
    for i in range(5):
        eng.sweep_sharded(1); eng.synchronize()
        g = eng.state()[0]
        print(ch, i, np.isfinite(g).all(), g[0,:3])
    print('diff', np.nanmax(np.abs(g-pg)/pg))
