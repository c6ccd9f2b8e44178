"""CPU test double for svinet_amd.sharded.HipShard: the four sweep phases on one
node block in numpy (pull-style, dense path), exposing the same surface
(kvec_a, kvec_c, gather_list(), phase(), end_sweep()) with CPU torch tensors,
so ShardedSweep's exchange protocol can be exercised with gloo.
Test infrastructure only."""
import numpy as np
import torch
from scipy.special import digamma

from svinet_amd.sharded import block_size, node_block


class NumpyShard:
    def __init__(self, n, k, ones, ones_prob, eta, links, validation_sorted, gamma, lam, rank, world):
        self.rank, self.world = rank, world
        self.n, self.k, self.ones = n, k, ones
        self.ones_prob, self.zeros_prob = ones_prob, 1 - ones_prob
        self.eta0, self.eta1 = eta
        self.alpha = 1.0 / k
        self.B = block_size(n, world)
        self.n_alloc = self.B * world
        self.lo, self.hi = node_block(n, world, rank)
        links = np.asarray(links, dtype=np.int64)
        self.P = np.concatenate([links[:, 0], links[:, 1]])      # directed entries
        self.Q = np.concatenate([links[:, 1], links[:, 0]])
        own = (self.P >= self.lo) & (self.P < self.hi)
        self.P, self.Q = self.P[own], self.Q[own]
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
        self.conv = torch.zeros(2, self.n_alloc, dtype=torch.int32)
        self.active = torch.zeros(self.n_alloc, 1, dtype=torch.int32)
        self.amask = torch.zeros(self.n_alloc, 1, dtype=torch.int64)
        self.member = torch.zeros(self.n_alloc, 1, dtype=torch.int64)
        self.xflags = torch.zeros(self.n_alloc, 4, dtype=torch.int32)   # conv (new), active, amask lo/hi
        self.kvec_a = torch.zeros(k, dtype=torch.float64)
        self.kvec_c = torch.zeros(3 * k, dtype=torch.float64)
        self.lam = np.array(lam, dtype=np.float64)
        self.elogbeta0 = digamma(self.lam[:, 0]) - digamma(self.lam.sum(1))
        self.iter, self.annealing, self.parity, self.sweeps = 0, True, 0, 0
        self.prev_h, self.nh = -2147483647.0, 0
        self.rows = []

    # ---- surface shared with HipShard ----
    def gather_list(self):
        return [self.t_gamma, self.xflags]

    def end_sweep(self):
        self.sweeps += 1

    def phase(self, ph):
        (self._a, self._b, self._c, self._d, self._expand)[ph]()

    def _expand(self):
        # rows of the other ranks: flags unpacked, Elogpi and mphi from the gathered gamma
        n, lo, hi = self.n, self.lo, self.hi
        oth = np.ones(n, dtype=bool)
        oth[lo:hi] = False
        xf = self.xflags.numpy()[:n]
        self.conv[self.parity ^ 1].numpy()[:n][oth] = xf[oth, 0]
        self.active.numpy()[:n, 0][oth] = xf[oth, 1]
        g = self.t_gamma.numpy()[:n][oth]
        self.t_elogpi.numpy()[:n][oth] = digamma(g) - digamma(g.sum(1, keepdims=True))
        isc = (self.kvec_a.numpy() / self.ones) if self.annealing else 1.0
        self.t_mphi.numpy()[:n][oth] = (g * isc - self.alpha) / (n - 1.0)

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

    def _b(self):
        n, k, lo, hi = self.n, self.k, self.lo, self.hi
        acc = self.acc[lo:hi]
        tl = 2.0 * self.deg[lo:hi][:, None]
        has = tl[:, 0] > 0
        m = np.where(tl > 0, acc / np.where(tl > 0, tl, 1.0), 0.0)
        g = self.alpha + acc + (n - tl - 1.0) * m
        if self.annealing:
            g = g * (self.ones / self.kvec_a.numpy())
        g[~has] = self.alpha
        mph = self.t_mphi.numpy()
        mph[lo:hi][has] = m[has]
        self.t_gamma.numpy()[lo:hi] = g
        self.t_elogpi.numpy()[lo:hi] = digamma(g) - digamma(g.sum(1, keepdims=True))
        act = (g - self.alpha >= 1)
        cnt = act.sum(1)
        lastk = k - 1 - np.argmax(act[:, ::-1], axis=1)
        old = self.conv[self.parity].numpy()[lo:hi]
        self.conv[self.parity ^ 1].numpy()[lo:hi] = np.where(cnt == 1, lastk + 1, old)
        self.active.numpy()[lo:hi, 0] = cnt
        xf = self.xflags.numpy()
        xf[lo:hi, 0] = self.conv[self.parity ^ 1].numpy()[lo:hi]
        xf[lo:hi, 1] = cnt
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
