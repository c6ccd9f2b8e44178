"""CPU: the K-sharded multi-GPU layout sketched in DESIGN.md section 6, as a numpy PROTOCOL against the oracle.

Not product code and not a substitute for it: this pins the design claim that a rank holding only the
columns [k0, k1) of every row can run the sweep of src/linksampling.cc:556-790 exactly, exchanging per
sweep nothing but
    one SUM of L doubles      (softmax denominators of the links that take the softmax branch),
    one SUM of 3N doubles     (row sums of the new gamma, active-community counts and index sums),
    one SUM of K doubles      (quirk Q2: mphi[q][pc] is added to s3[pc - 1], which may live on the rank to the left),
    one SUM of V doubles      (partial dot products of the held-out pairs),
instead of the N*K doubles the node-block layout all-gathers.  Every cross-rank value below goes through
`exchange()`, which also counts the doubles; everything else a rank touches is its own column slice or
replicated integers.  Dense path only (_iter <= 1000).
"""
import numpy as np
import pytest
from scipy.special import digamma

from oracle import oracle as O


class Exchange:
    def __init__(self):
        self.doubles = 0

    def sum(self, parts):
        self.doubles += int(np.asarray(parts[0]).size)
        tot = parts[0].copy()
        for x in parts[1:]:
            tot = tot + x
        return tot


class Rank:
    """one rank: the columns `cols` of gamma / Elogpi / mphi / lambda / Elogbeta"""

    def __init__(self, cols, ref):
        self.cols = cols
        self.k0, self.k1 = int(cols[0]), int(cols[-1]) + 1
        self.gamma = ref.gamma[:, cols].copy()
        self.elogpi = ref.elogpi[:, cols].copy()
        self.mphi = ref.mphi[:, cols].copy()
        self.lam = ref.lam[cols].copy()
        self.elogbeta = ref.elogbeta[cols].copy()

    def owns(self, k):
        return (k >= self.k0) & (k < self.k1)


def ksharded_sweep(ranks, rep, links, tl, consts, ex):
    """one sweep; `rep` = the replicated scalars / integer vectors (identical on every rank by construction)"""
    n, K, E, alpha, eta0, eta1 = consts["n"], consts["K"], consts["E"], consts["alpha"], consts["eta0"], consts["eta1"]
    p, q = links[:, 0], links[:, 1]
    conv = rep["conv"]
    pc, qc = conv[p], conv[q]
    sh1, sh2 = (pc != 0) & (qc == 0), (qc != 0) & (pc == 0)
    dense = ~(sh1 | sh2)
    c = np.where(sh1, pc, qc).astype(np.int64) - 1          # shortcut column (valid where sh1 | sh2)
    pd, qd = p[dense], q[dense]
    # ---- A6, pass 1: partial softmax denominators of the dense links, then ONE exchange
    e = [np.exp(r.elogpi[pd] + r.elogpi[qd] + r.elogbeta[:, 0]) for r in ranks]
    S = ex.sum([x.sum(1) for x in e])
    # ---- A6, pass 2 + A7 + A8, all column-local
    q2_parts, rows_parts = [], []
    for r, er in zip(ranks, e):
        kl = len(r.cols)
        phi = er / S[:, None]
        gn = np.full((n, kl), alpha)
        np.add.at(gn, pd, phi)
        np.add.at(gn, qd, phi)
        ssum = 2.0 * phi.sum(0)
        mine = (sh1 | sh2) & r.owns(c)
        np.add.at(gn, (p[mine], c[mine] - r.k0), 1.0)
        np.add.at(gn, (q[mine], c[mine] - r.k0), 1.0)
        np.add.at(ssum, c[mine] - r.k0, 2.0)
        has = tl > 0
        m = (gn[has] - alpha) / tl[has, None]
        r.mphi[has] = m                                      # rows without a training link stay stale
        s1, s2 = m.sum(0), (m * m).sum(0)
        gn[has] += (n - tl[has, None] - 1.0) * m
        if rep["annealing"]:
            gn[has] *= E / ssum
        s3 = (r.mphi[pd] * r.mphi[qd]).sum(0)
        q2 = np.zeros(K)                                     # Q2: the owner of column pc holds mphi[.][pc], the target is pc - 1
        for sel, cc, other in ((sh1, pc, q), (sh2, qc, p)):
            col = cc[sel].astype(np.int64)                   # = converged community + 1 (one past it)
            ok = (col < K) & r.owns(col)
            np.add.at(q2, col[ok] - 1, r.mphi[other[sel][ok], col[ok] - r.k0])
        q2_parts.append(q2)
        r._tmp = (gn, ssum, s1, s2, s3)
        rows_parts.append(np.stack([gn.sum(1), (gn - alpha >= 1.0).sum(1).astype(np.float64),
                                    ((gn - alpha >= 1.0) * (r.cols + 1.0)).sum(1)], 1))
    q2 = ex.sum(q2_parts)
    rows = ex.sum(rows_parts)                                # row sums, active counts, index sums: [n][3]
    rowsum, active, idx = rows[:, 0], rows[:, 1].astype(np.int64), rows[:, 2].astype(np.int64)
    # ---- lambda, A5, A9
    for r in ranks:
        gn, ssum, s1, s2, s3 = r._tmp
        r.lam = np.stack([eta0 + ssum, eta1 + (s1 * s1 - s2 - (s3 + q2[r.cols]))], 1)
        r.gamma = gn
        r.elogpi = digamma(gn) - digamma(rowsum)[:, None]
        r.elogbeta = digamma(r.lam) - digamma(r.lam.sum(1))[:, None]
    rep["conv"] = np.where(active == 1, idx, conv).astype(conv.dtype)
    rep["active"] = active
    # ---- A10: held-out likelihood from partial dot products, then the stop rule (replicated)
    vp, vq, vy = rep["vpairs"][:, 0], rep["vpairs"][:, 1], rep["vpairs"][:, 2]
    dots = ex.sum([(r.gamma[vp] * r.gamma[vq] * (r.lam[:, 0] / r.lam.sum(1))).sum(1) for r in ranks])
    pq = dots / (rowsum[vp] * rowsum[vq])
    u = np.log(np.maximum(np.where(vy != 0, pq, 1.0 - pq), 1e-30))
    a = consts["zeros_prob"] * u[vy == 0].mean() + consts["ones_prob"] * u[vy != 0].mean()
    stop = False
    if rep["iter"] > 10:
        prev = rep["prev_h"]
        if a > prev and prev != 0 and abs((a - prev) / prev) < 1e-5:
            stop = True
        elif a < prev:
            rep["nh"] += 1
        elif a > prev:
            rep["nh"] = 0
        if rep["nh"] > 2:
            stop = True
    rep["prev_h"] = a
    if rep["annealing"] and stop:
        rep["annealing"], rep["nh"], rep["prev_h"] = False, 0, 0.0
    rep["iter"] += 1
    return a


@pytest.mark.parametrize("world,sweeps", [(4, 70), (7, 12)])
def test_ksharded_protocol_equals_the_oracle(graph_files, world, sweeps):
    path, n, K = graph_files["lfr"], 1000, 28
    net = O.Network(path, n)
    ref = O.LinkSampling(net, K, use_validation_stop=False)
    links = ref.links.astype(np.int64)
    tl = ref.training_links
    eta0, eta1 = ref.eta
    consts = dict(n=n, K=K, E=float(len(net.edges())), alpha=1.0 / K, eta0=eta0, eta1=eta1,
                  ones_prob=ref.ones_prob, zeros_prob=1.0 - ref.ones_prob)
    ranks = [Rank(cols, ref) for cols in np.array_split(np.arange(K), world)]
    rep = dict(conv=ref.converged.copy(), active=ref.active_comms.copy(), annealing=True, iter=0, nh=0,
               prev_h=-2147483647.0, vpairs=ref.validation_sorted.astype(np.int64))
    ex = Exchange()
    saw_shortcut = saw_q2 = False
    for s in range(sweeps):
        before = ex.doubles
        a = ksharded_sweep(ranks, rep, links, tl, consts, ex)
        ref.sweep()
        assert abs(a - ref.rows[-1][9]) < 1e-9 * max(1.0, abs(a)), (s, a, ref.rows[-1][9])
        assert np.array_equal(rep["conv"], ref.converged), s
        assert np.array_equal(rep["active"], ref.active_comms), s
        assert rep["annealing"] == ref.annealing and rep["iter"] == ref.iter
        dense, _, short = ref.link_counts()
        saw_shortcut |= short > 0
        # the whole exchange of a sweep: dense denominators + 3N + K + V doubles, far below the N*K of a gamma gather
        assert ex.doubles - before == dense + 3 * n + K + len(rep["vpairs"])
        g = np.concatenate([r.gamma for r in ranks], 1)
        lam = np.concatenate([r.lam for r in ranks], 0)
        np.testing.assert_allclose(g, ref.gamma, rtol=1e-9)
        np.testing.assert_allclose(lam, ref.lam, rtol=1e-9)
    if sweeps >= 70:
        assert saw_shortcut and not ref.annealing      # the run crossed the annealing switch with shortcut links present
