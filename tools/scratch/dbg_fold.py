import os, sys, gzip
sys.path.insert(0, os.getcwd())
import numpy as np
from svinet_amd import _svils
from svinet_amd.host_api import Setup
open('/tmp/lfr.txt','wb').write(gzip.open('tests/golden/graphs/LFR-network-n1000-k28.txt.gz','rb').read())
setup = Setup('/tmp/lfr.txt', 1000, 28)
plain = setup.engine(use_validation_stop=False); plain.sweep(30); pg = plain.state()[0]
for mode in ('0', '1', '1'):
    os.environ['SVILS_SHARD_FOLD'] = mode
    eng = setup.engine(use_validation_stop=False, node_block=(0, 1000), n_alloc=1000)
    eng.comm_init(_svils.comm_unique_id(), 0, 1)
    eng.enable_timing(1 << _svils.KERNEL_EXCHANGE)
    eng.sweep_sharded(30); eng.synchronize()
    print(mode, float(np.max(np.abs(eng.state()[0] - pg) / pg)))
    eng.close()
