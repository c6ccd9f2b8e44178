"""CPU test double for svinet_amd.sharded.HipShard: the phases of a node-block sweep on one
node block in numpy (pull-style, dense path), exposing the same surface
(kvec_a, kvec_c, gstage, bounds, bmax, phase(), end_sweep()) with CPU torch tensors,
so ShardedSweep's exchange protocol can be exercised with gloo.
Test infrastructure only."""
import numpy as np
import torch
from scipy.special import digamma

from svinet_amd.sharded import balanced_bounds, equal_bounds


class NumpyShard:
    def __init__(self, n, k, ones, ones_prob, eta, links, validation_sorted, gamma, lam, rank, world, balanced=True):
        self.rank, self.world = rank, world
        self.n, self.k, self.ones = n, k, ones
        self.ones_prob, self.zeros_prob = ones_prob, 1 - ones_prob
        self.eta0, self.eta1 = eta
        self.alpha = 1.0 / k
        links = np.asarray(links, dtype=np.int64)
        # the library's own cut (host code of libsvils.so: no device needed), or the equal blocks
        self.bounds = (balanced_bounds(links, n, world) if balanced else equal_bounds(n, world)).astype(np.int64)
        self.bmax = int(np.diff(self.bounds).max())
        self.n_alloc = n
        self.lo, self.hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.P = np.concatenate([links[:, 0], links[:, 1]])      # directed entries
        self.Q = np.concatenate([links[:, 1], links[:, 0]])
        own = (self.P >= self.lo) & (self.P < self.hi)
        self.P, self.Q = self.P[own], self.Q[own]
        # the s3 pass is cut by link count, not by node block (svils_set_node_blocks with caller-given bounds)
        L = len(links)
        if balanced and world > 1:
            sl = slice(L * rank // world, L * (rank + 1) // world)
            self.UP, self.UQ = links[sl, 0], links[sl, 1]
        else:
            up = (links[:, 0] >= self.lo) & (links[:, 0] < self.hi)
            self.UP, self.UQ = links[up, 0], links[up, 1]
        self.deg = np.bincount(np.concatenate([links[:, 0], links[:, 1]]), minlength=n).astype(np.float64)
        self.val = np.asarray(validation_sorted, dtype=np.int64)

        def full(cols, dtype=torch.float64):
            return torch.zeros(self.n_alloc, cols, dtype=dtype)
        self.t_gamma, self.t_elogpi, self.t_mphi = full(k), full(k), full(k)
        self.t_gamma[:n] = torch.from_numpy(np.array(gamma))
        g = self.t_gamma[:n].numpy()
        self.t_elogpi[:n] = torch.from_numpy(digamma(g) - digamma(g.sum(1, keepdims=True)))
        self.gstage = torch.zeros(world * self.bmax, k, dtype=torch.float64)
        self.conv = torch.zeros(2, self.n_alloc, dtype=torch.int32)
        self.active = torch.zeros(self.n_alloc, 1, dtype=torch.int32)
        self.member = torch.zeros(self.n_alloc, 1, dtype=torch.int64)
        self.kvec_a = torch.zeros(k, dtype=torch.float64)
        self.kvec_c = torch.zeros(3 * k, dtype=torch.float64)
        self.lam = np.array(lam, dtype=np.float64)
        self.elogbeta0 = digamma(self.lam[:, 0]) - digamma(self.lam.sum(1))
        self.iter, self.annealing, self.parity, self.sweeps = 0, True, 0, 0
        self.prev_h, self.nh = -2147483647.0, 0
        self.rows = []

    # ---- surface shared with HipShard ----
    def end_sweep(self):
        self.sweeps += 1

    def phase(self, ph):
        # svils_phase: A, B, C, D, EXPAND, B_LIGHT, EXPAND_ALL
        (self._a, None, self._c, self._d, None, self._b_light, self._expand_all)[ph]()

    def _expand_all(self):
        """every row from the staged (unscaled) rows: annealing scale, gamma, Elogpi, mphi of the other blocks,
        prune() flags -- the same on every rank"""
        n, k = self.n, self.k
        st = self.gstage.numpy()
        raw = np.empty((n, k))
        for r in range(self.world):
            b0, b1 = int(self.bounds[r]), int(self.bounds[r + 1])
            raw[b0:b1] = st[r * self.bmax:r * self.bmax + (b1 - b0)]
        has = self.deg > 0
        scale = (self.ones / self.kvec_a.numpy()) if self.annealing else np.ones(k)
        g = np.where(has[:, None], raw * scale, raw)
        oth = np.ones(n, dtype=bool)
        oth[self.lo:self.hi] = False
        self.t_mphi.numpy()[:n][oth] = np.where(has[oth, None], (raw[oth] - self.alpha) / (n - 1.0), 0.0)
        self.t_gamma.numpy()[:n] = g
        self.t_elogpi.numpy()[:n] = digamma(g) - digamma(g.sum(1, keepdims=True))
        act = (g - self.alpha >= 1)
        cnt = act.sum(1)
        lastk = k - 1 - np.argmax(act[:, ::-1], axis=1)
        old = self.conv[self.parity].numpy()[:n]
        self.conv[self.parity ^ 1].numpy()[:n] = np.where(cnt == 1, lastk + 1, old)
        self.active.numpy()[:n, 0] = cnt

    # ---- phases ----
    def _a(self):
        n, k = self.n, self.k
        conv = self.conv[self.parity].numpy()[:n]
        el = self.t_elogpi.numpy()[:n]
        pc, qc = conv[self.P], conv[self.Q]
        short = (pc > 0) != (qc > 0)
        acc = np.zeros((n, k))
        dn = ~short
        x = el[self.P[dn]] + el[self.Q[dn]] + self.elogbeta0
        x -= x.max(1, keepdims=True)
        e = np.exp(x)
        np.add.at(acc, self.P[dn], e / e.sum(1, keepdims=True))
        c = np.where(pc > 0, pc, qc) - 1
        np.add.at(acc, (self.P[short], c[short]), 1.0)
        self.acc = acc
        self.kvec_a[:] = torch.from_numpy(acc[self.lo:self.hi].sum(0))

    def _b_light(self):
        n, k, lo, hi = self.n, self.k, self.lo, self.hi
        acc = self.acc[lo:hi]
        tl = 2.0 * self.deg[lo:hi][:, None]
        has = tl[:, 0] > 0
        m = np.where(tl > 0, acc / np.where(tl > 0, tl, 1.0), 0.0)
        g = self.alpha + acc + (n - tl - 1.0) * m          # unscaled
        g[~has] = self.alpha
        mph = self.t_mphi.numpy()
        mph[lo:hi][has] = m[has]
        self.gstage.numpy()[self.rank * self.bmax:self.rank * self.bmax + (hi - lo)] = g
        kc = self.kvec_c.numpy()
        kc[:k] = m[has].sum(0)
        kc[k:2 * k] = (m[has] ** 2).sum(0)

    def _c(self):
        n, k = self.n, self.k
        conv = self.conv[self.parity].numpy()[:n]
        mph = self.t_mphi.numpy()[:n]
        pc, qc = conv[self.UP], conv[self.UQ]
        s3 = np.zeros(k)
        a = (pc > 0) & (qc == 0)
        b = (pc == 0) & (qc > 0)
        d = ~(a | b)
        s3 += (mph[self.UP[d]] * mph[self.UQ[d]]).sum(0)
        for sel, cc, other in ((a, pc, self.UQ), (b, qc, self.UP)):
            idx = np.nonzero(sel)[0]
            cv = cc[idx]
            vals = np.where(cv < k, mph[other[idx], np.minimum(cv, k - 1)], 0.0)   # quirk Q2
            np.add.at(s3, cv - 1, vals)
        self.kvec_c.numpy()[2 * k:] = s3

    def _d(self):
        k = self.k
        kc = self.kvec_c.numpy()
        s1, s2, s3 = kc[:k], kc[k:2 * k], kc[2 * k:]
        self.lam = np.stack([self.eta0 + self.kvec_a.numpy(), self.eta1 + (s1 * s1 - s2 - s3)], 1)
        self.elogbeta0 = digamma(self.lam[:, 0]) - digamma(self.lam.sum(1))
        self.parity ^= 1
        if len(self.val):
            g = self.t_gamma.numpy()
            gp, gq, y = g[self.val[:, 0]], g[self.val[:, 1]], self.val[:, 2]
            beta = self.lam[:, 0] / self.lam.sum(1)
            pq = (gp * gq * beta).sum(1) / (gp.sum(1) * gq.sum(1))
            u = np.log(np.maximum(np.where(y == 1, pq, 1.0 - pq), 1e-30))
            a = self.zeros_prob * u[y == 0].mean() + self.ones_prob * u[y == 1].mean()
            self.rows.append(a)
            stop = False
            if self.iter > 10:
                if a > self.prev_h and self.prev_h != 0 and abs((a - self.prev_h) / self.prev_h) < 1e-5:
                    stop = True
                elif a < self.prev_h:
                    self.nh += 1
                elif a > self.prev_h:
                    self.nh = 0
                if self.nh > 2:
                    stop = True
            self.prev_h = a
            if self.annealing and stop:
                self.annealing, self.nh, self.prev_h = False, 0, 0.0
        self.iter += 1
