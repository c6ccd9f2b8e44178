"""Sparse MMSB graph generator for the large link-sampling runs (BASELINE config 5).

The reference's generator (`MMSBGen::gen`, src/mmsbgen.cc:44-71, and
`draw_membership_indicators`, src/mmsbgen.hh:143-210) visits all n(n-1)/2 pairs:
pi_p ~ Dirichlet(alpha), beta_k ~ Beta(eta0_gen, eta1_gen) = Beta(4700.59, 0.77)
(src/env.hh:371-378); per pair z_p ~ Mult(pi_p), z_q ~ Mult(pi_q) and
y ~ Bernoulli(beta_k) iff z_p == z_q == k (epsilon is treated as 0).  That is
O(n^2) and cannot make an n = 10^6 graph.  This module draws from the same
process restricted to the pairs that can be links, in O(links):

  * pi_p ~ Dirichlet(alpha) truncated to its `top` largest components and renormalised;
  * P(z_p = z_q = k) = pi_p[k] * pi_q[k], so the links of community k are endpoint
    pairs drawn independently with probability proportional to pi_p[k] (resp. pi_q[k]);
    their number is Poisson with mean  rate * beta_k * (sum_p pi_p[k])^2 / 2, where the
    single scalar `rate` (< 1: the graph is a thinned version of the reference's process)
    is set so that the expected mean degree is `mean_deg`;
  * self pairs and duplicates are dropped; nodes left without a link are attached to a
    member of their strongest community, so every id in [0, n) appears and `-n n` holds
    (the reference renumbers away isolated nodes, src/main.cc:291).

Randomness: numpy's Philox counter-based generator keyed with `seed` (default
20240517); the output is a pure function of (n, k, mean_deg, alpha, top, seed) for a
given numpy version.  Output rows are (i, j) with i < j, sorted.
"""
import numpy as np

ETA0_GEN, ETA1_GEN = 4700.59, 0.77     # src/env.hh:371-378 (eta*_dense)
DEFAULT_SEED = 20240517


def memberships(n, k, alpha, top, rng, chunk=16384):
    """top-`top` truncated Dirichlet(alpha) rows: (comm[n][top] int32, w[n][top] float64, rows sum to 1)"""
    top = min(top, k)
    comm = np.empty((n, top), dtype=np.int32)
    w = np.empty((n, top), dtype=np.float64)
    for lo in range(0, n, chunk):
        hi = min(lo + chunk, n)
        # Dirichlet = normalised Gamma(alpha, 1) draws; only the top components survive, so the
        # normalisation is done after truncation.  log-space (Gamma(a) = Gamma(a+1) * U^(1/a))
        # keeps the tiny-alpha draws from underflowing to an all-zero row.
        g = np.log(rng.standard_gamma(alpha + 1.0, size=(hi - lo, k))) + np.log(rng.random((hi - lo, k))) / alpha
        idx = np.argpartition(g, k - top, axis=1)[:, k - top:]
        val = np.take_along_axis(g, idx, axis=1)
        val = np.exp(val - val.max(axis=1, keepdims=True))
        order = np.argsort(-val, axis=1, kind="stable")
        comm[lo:hi] = np.take_along_axis(idx, order, axis=1)
        val = np.take_along_axis(val, order, axis=1)
        w[lo:hi] = val / val.sum(axis=1, keepdims=True)
    return comm, w


def generate(n, k, mean_deg=24, alpha=0.05, top=4, seed=DEFAULT_SEED, return_truth=False):
    """-> int32 pairs [E][2] (i < j, sorted, unique); with return_truth also (comm, w, beta)."""
    rng = np.random.Generator(np.random.Philox(seed))
    comm, w = memberships(n, k, alpha, top, rng)
    beta = rng.beta(ETA0_GEN, ETA1_GEN, size=k)
    # membership entries grouped by community, with a running weight sum
    flat_c = comm.ravel()
    flat_w = w.ravel()
    flat_p = np.repeat(np.arange(n, dtype=np.int64), comm.shape[1])
    keep = flat_w > 0
    flat_c, flat_w, flat_p = flat_c[keep], flat_w[keep], flat_p[keep]
    order = np.argsort(flat_c, kind="stable")
    flat_c, flat_w, flat_p = flat_c[order], flat_w[order], flat_p[order]
    cum = np.cumsum(flat_w)
    end = np.zeros(k, dtype=np.float64)
    start = np.zeros(k, dtype=np.float64)
    cnt = np.bincount(flat_c, minlength=k)
    last = np.cumsum(cnt) - 1
    has = cnt > 0
    end[has] = cum[last[has]]
    start[has] = np.concatenate([[0.0], end[has][:-1]])
    mass = end - start                                   # sum_p pi_p[k]
    budget = beta * mass * mass / 2.0
    want = n * mean_deg / 2.0
    mk = rng.poisson(budget * (want / budget.sum()))
    ck = np.repeat(np.arange(k), mk)
    m = ck.shape[0]

    def endpoints():
        u = start[ck] + rng.random(m) * mass[ck]
        i = np.searchsorted(cum, u, side="right")
        i = np.minimum(i, last[ck])                      # guard the top edge of a community's span
        return flat_p[i]

    a, b = endpoints(), endpoints()
    ok = a != b
    a, b = a[ok], b[ok]
    key = np.unique(np.minimum(a, b) * n + np.maximum(a, b))
    # attach isolated nodes to a member of their strongest community
    deg = np.bincount(np.concatenate([key // n, key % n]), minlength=n)
    iso = np.nonzero(deg == 0)[0]
    if iso.size:
        c0 = comm[iso, 0]
        u = start[c0] + rng.random(iso.size) * mass[c0]
        j = flat_p[np.minimum(np.searchsorted(cum, u, side="right"), last[c0])]
        same = j == iso
        j[same] = (iso[same] + 1) % n
        key = np.unique(np.concatenate([key, np.minimum(iso, j) * n + np.maximum(iso, j)]))
    pairs = np.stack([key // n, key % n], axis=1).astype(np.int32)
    if return_truth:
        return pairs, (comm, w, beta)
    return pairs


def write_pairs(path, pairs):
    """the reference's input format: "%d\\t%d\\n" (Network::read, src/network.cc:10-116)"""
    np.savetxt(path, pairs, fmt="%d", delimiter="\t")


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("-n", type=int, required=True)
    ap.add_argument("-k", type=int, required=True)
    ap.add_argument("--mean-deg", type=int, default=24)
    ap.add_argument("--alpha", type=float, default=0.05)
    ap.add_argument("--top", type=int, default=4)
    ap.add_argument("--seed", type=int, default=DEFAULT_SEED)
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    p = generate(a.n, a.k, a.mean_deg, a.alpha, a.top, a.seed)
    write_pairs(a.out, p)
    print("wrote %d links over %d nodes to %s" % (p.shape[0], a.n, a.out))
