"""The sparse MMSB generator that feeds the large link-sampling runs."""
import numpy as np
import pytest

from svinet_amd import mmsbgen_sparse as G


def nmi(a, b):
    """normalised mutual information (arithmetic-mean normalisation) of two labelings"""
    a = np.unique(a, return_inverse=True)[1]
    b = np.unique(b, return_inverse=True)[1]
    c = np.zeros((a.max() + 1, b.max() + 1))
    np.add.at(c, (a, b), 1)
    p = c / c.sum()
    pa, pb = p.sum(1), p.sum(0)
    nz = p > 0
    mi = (p[nz] * np.log(p[nz] / np.outer(pa, pb)[nz])).sum()
    ha = -(pa[pa > 0] * np.log(pa[pa > 0])).sum()
    hb = -(pb[pb > 0] * np.log(pb[pb > 0])).sum()
    return mi / max((ha + hb) / 2, 1e-300)


def test_generator_shape_and_determinism():
    n, k = 5000, 32
    p1, (comm, w, beta) = G.generate(n, k, 24, return_truth=True)
    p2 = G.generate(n, k, 24)
    assert np.array_equal(p1, p2)                                   # pure function of its arguments
    assert not np.array_equal(p1, G.generate(n, k, 24, seed=7))
    assert p1.dtype == np.int32 and p1.shape[1] == 2
    assert np.all(p1[:, 0] < p1[:, 1]) and p1.min() >= 0 and p1.max() < n
    key = p1[:, 0].astype(np.int64) * n + p1[:, 1]
    assert np.all(np.diff(key) > 0)                                 # sorted, unique
    deg = np.bincount(p1.ravel(), minlength=n)
    assert deg.min() >= 1                                           # -n n holds
    assert 0.85 * 24 < deg.mean() < 1.05 * 24
    assert comm.shape == (n, 4) and np.allclose(w.sum(1), 1) and np.all(np.diff(w, axis=1) <= 0)
    assert np.all((beta > 0.99) & (beta <= 1))
    # links join nodes that share a community
    share = (comm[p1[:, 0]][:, :, None] == comm[p1[:, 1]][:, None, :]).any((1, 2))
    assert share.mean() > 0.99


def test_k_smaller_than_top():
    p = G.generate(200, 2, 8)
    assert np.bincount(p.ravel(), minlength=200).min() >= 1


def test_oracle_recovers_planted_memberships():
    """the reference algorithm (oracle) run to its stop rule recovers the planted structure"""
    from oracle import oracle as O
    n, k = 1500, 8
    pairs, (comm, w, _) = G.generate(n, k, 24, return_truth=True)
    net = O.Network(n=n, pairs=pairs)
    ref = O.LinkSampling(net, k)
    for _ in range(200):
        if ref.sweep() == 2:
            break
    s2i = net.seq2id()
    strong = w[s2i, 0] > 0.9
    assert nmi(comm[s2i, 0][strong], ref.gamma.argmax(1)[strong]) > 0.9
