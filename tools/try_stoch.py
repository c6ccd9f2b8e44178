import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from svinet_amd import mmsbgen_sparse as G
from svinet_amd.host_api import Setup
from test_mmsbgen import nmi
n, k, alpha = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
pairs, (comm, w, _) = G.generate(n, k, 24, alpha=alpha, return_truth=True)
rng = np.random.default_rng(5)
perm = rng.permutation(n).astype(np.int32)
p2 = perm[pairs]; p2.sort(axis=1); p2 = p2[np.lexsort((p2[:,1], p2[:,0]))]
inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
def run(tag, eng, stepper, s, total, per):
    t0=time.time()
    for i in range(total // per):
        stepper(per)
        c = eng.control()
        if c.stopped: break
    eng.synchronize(); el=time.time()-t0
    g, lam, conv = eng.state()
    orig = inv[s.seq2id]           # original label of each seq node
    strong = w[orig, 0] > 0.9
    rows = eng.rows()
    print(tag, "iters", c.iter, "stopped", c.stopped, "annealing", c.annealing, "time %.2f" % el, "a first/last", rows[0,9], rows[-1,9], "nmi strong %.3f" % nmi(comm[orig,0][strong], g.argmax(1)[strong]), "conv", (conv>0).sum())
s = Setup(n=n, k=k, pairs=p2)
print("links", s.nlinks)
e = s.engine(); run("batch", e, e.sweep, s, 400, 10)
for bn_div, tau0, kappa, epochs in [(10, 16, 0.5, 40), (10, 1, 0.5, 40), (10, 64, 0.7, 40), (50, 64, 0.5, 40), (4, 4, 0.5, 40)]:
    nb = (n + (n//bn_div) - 1)//(n//bn_div)
    e = s.engine(reportfreq=nb)
    e.set_stochastic(batch_nodes=n//bn_div, tau0=tau0, kappa=kappa)
    run("stoch div%d tau%g kap%g" % (bn_div, tau0, kappa), e, e.step, s, epochs*nb, nb)
