"""The planted-state config-5 run against its committed oracle digest, column by column: which likelihood columns differ
and by how much (profiles/r04i_planted_state_likelihood_rows_diff.txt).  python tools/planted_state_diff.py  (GPU, ~1 min)"""
import sys, os, json, hashlib, importlib.util
import numpy as np
sys.path.insert(0, ".")
from svinet_amd import mmsbgen_sparse as G
from svinet_amd.host_api import Setup
d = "tests/golden/config5"
meta = json.load(open(os.path.join(d, "digest_planted.json")))
dg = np.load(os.path.join(d, "digest_planted.npz"))
spec = importlib.util.spec_from_file_location("m", "tools/make_config5_digest.py")
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
n, k = meta["n"], meta["k"]
pairs, truth = G.generate(n, k, meta["mean_degree"], return_truth=True)
g0, lam0, conv0 = tool.planted_state(pairs, truth, n, k)
s = Setup(n=n, k=k, pairs=pairs)
eng = s.engine(use_validation_stop=False)
eng.set_state(g0, lam0, conv0)
eng.set_control(iter=meta["iter0"], annealing=0)
nsw = meta["sweeps"]
for i in range(nsw):
    eng.sweep(1)
g, lam, conv = eng.state()
rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
print("lam rel", rel(lam, dg["lam"]), "colsum rel", rel(g.sum(0), dg["gamma_colsum"]), "rows rel", rel(g[dg["rows_idx"]], dg["gamma_rows"]))
want = dg["likelihood_rows"][1:]
got = eng.rows()
np.set_printoptions(precision=17, linewidth=250)
for r in range(nsw):
    print("row", r)
    print(" got ", got[r])
    print(" want", want[r])
    print(" abs ", got[r] - want[r])
    print(" rel ", (got[r] - want[r]) / np.where(want[r] != 0, want[r], 1))
la = lam[:, 0] / (lam[:, 0] + lam[:, 1]); lb = dg["lam"][:, 0] / (dg["lam"][:, 0] + dg["lam"][:, 1])
print("beta rel max", rel(la, lb), "lam0 rel", rel(lam[:, 0], dg["lam"][:, 0]), "lam1 rel", rel(lam[:, 1], dg["lam"][:, 1]))
