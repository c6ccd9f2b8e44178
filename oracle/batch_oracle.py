"""numpy restatement of the reference's `-batch` engine (MMSBInfer::batch_infer).

TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/): imported by tests/ to check the
host C++ engine svinet_amd/host/mmsbbatch.cc; never by the product.

Parity status: the held-out / validation samplers are pinned by the authors' shipped run
(example/n75-k4-mmsb-batch.tgz: heldout-edges.txt, validation-edges.txt -- that older revision
printed sequence ids).  Everything after the gamma initialisation is UNPINNED against the
reference: gamma starts from gsl_ran_gamma draws whose stream is internal to GSL,
so the sweep is checked engine-vs-restatement from a shared starting point only.

Follows: PhiComp::update_phis / update_phis_until_conv (src/mmsbinfer.hh:104-203),
MMSBInfer::batch_infer (src/mmsbinfer.cc:833-930), edge_likelihood (src/mmsbinfer.hh:634-668),
heldout_likelihood (src/mmsbinfer.cc:2086-2174).
"""
import numpy as np
from scipy.special import digamma

ONLINE_ITERATIONS = 50          # src/env.hh:415
MEAN_CHANGE_THRESH = 0.00001    # src/env.hh:337
EPSILON = 1e-30                 # src/env.hh:395


def dir_exp(a):
    """set_dir_exp (src/mmsbinfer.hh:563-580): psi(a_ij) - psi(sum_j a_ij)"""
    return digamma(a) - digamma(a.sum(1, keepdims=True))


def phis(elogpi, elogbeta, p, q, y):
    """per-pair fixed point for arrays of pairs (p[i], q[i], y[i]) -> phi1, phi2  [npairs][K]"""
    npairs, K = p.shape[0], elogpi.shape[1]
    yk = y[:, None].astype(np.float64)
    elogf = elogbeta[None, :, 0] * yk + elogbeta[None, :, 1] * (1 - yk)
    logeps = np.log(EPSILON)
    phi1 = np.full((npairs, K), 1.0 / K)
    phi2 = np.full((npairs, K), 1.0 / K)
    old1 = np.zeros((npairs, K))
    old2 = np.zeros((npairs, K))
    live = np.ones(npairs, dtype=bool)

    def update(b, c):
        a = np.exp(elogpi[c] + elogf * b + np.where(yk == 1, (1 - b) * logeps, 0.0))
        s = a.sum(1, keepdims=True)
        assert np.all(s > 0)
        return a / s

    for i in range(ONLINE_ITERATIONS):
        if i % 2 == 0:
            old1[live] = phi1[live]
            old2[live] = phi2[live]
        n1 = update(phi2, p)
        n2 = update(phi1, q)
        v1 = np.abs(n1 - old1).mean(1)
        v2 = np.abs(n2 - old2).mean(1)
        phi1[live] = n1[live]
        phi2[live] = n2[live]
        if i % 2 == 0:
            continue
        live &= ~((v1 < MEAN_CHANGE_THRESH) & (v2 < MEAN_CHANGE_THRESH))
        if not live.any():
            break
    return phi1, phi2


def sweep(gamma, lam, adj, skip, alpha, eta):
    """one pass of batch_infer's loop body: returns the new (gamma, lambda).
    adj: [n][n] 0/1 symmetric; skip: set of (p,q) p<q held out or in the validation set"""
    n, K = gamma.shape
    elogpi, elogbeta = dir_exp(gamma), dir_exp(lam)
    iu = np.triu_indices(n, 1)
    keep = np.array([(a, b) not in skip for a, b in zip(*iu)])
    p, q = iu[0][keep], iu[1][keep]
    y = adj[p, q]
    phi1, phi2 = phis(elogpi, elogbeta, p, q, y)
    gnext = np.full((n, K), alpha)
    np.add.at(gnext, p, phi1)
    np.add.at(gnext, q, phi2)
    lnext = np.tile(np.asarray(eta, dtype=np.float64), (K, 1))
    pp = phi1 * phi2
    lnext[:, 0] += (pp * (y[:, None] == 1)).sum(0)
    lnext[:, 1] += (pp * (y[:, None] == 0)).sum(0)
    return gnext, lnext


def edge_likelihood(gamma, lam, p, q, y):
    pi_p, pi_q = gamma[p] / gamma[p].sum(), gamma[q] / gamma[q].sum()
    beta = lam[:, 0] / lam.sum(1)
    if y == 1:
        s = float((pi_p * pi_q * beta).sum())
    else:
        rate = np.full((gamma.shape[1],) * 2, EPSILON)
        np.fill_diagonal(rate, beta)
        s = float((np.outer(pi_p, pi_q) * (1 - rate)).sum())
    return np.log(max(s, 1e-30))


def heldout_row(gamma, lam, pairs_sorted, adj, ones_prob):
    """columns 2..10 of a heldout.txt row (iteration and duration left out)"""
    u = np.array([edge_likelihood(gamma, lam, a, b, adj[a, b]) for a, b in pairs_sorted])
    y = np.array([adj[a, b] for a, b in pairs_sorted])
    m0, m1 = u[y == 0].mean(), u[y == 1].mean()
    z = 1 - ones_prob
    return [u.mean(), len(u), m0, int((y == 0).sum()), m1, int((y == 1).sum()), z * m0, ones_prob * m1,
            z * m0 + ones_prob * m1]
