"""CPU test double for svinet_amd.ksharded.KShard: the five phases of a K-sharded sweep on one column slice in numpy
(dense path), exposing the surface KShardedSweep drives (engine.ksweep_phase / synchronize, buf[...] as CPU torch
tensors), so its exchange order can be exercised with gloo in several processes.  Test infrastructure only; the
arithmetic is the protocol of tests/test_ksharded_protocol.py cut at its exchange points."""
import numpy as np
import torch
from scipy.special import digamma

from svinet_amd import _svils
from svinet_amd.ksharded import column_slices


class _Engine:
    def __init__(self, owner):
        self.o = owner

    def ksweep_phase(self, ph):
        self.o.phase(int(ph))

    def synchronize(self):
        pass


class NumpyKShard:
    def __init__(self, n, k, ones, ones_prob, eta, links, tl, validation_sorted, gamma, lam, rank, world):
        self.torch = torch
        self.rank, self.world = rank, world
        self.n, self.K, self.E = n, k, float(ones)
        self.ones_prob, self.zeros_prob = ones_prob, 1.0 - ones_prob
        self.eta0, self.eta1 = eta
        self.alpha = 1.0 / k
        self.k0, self.k1 = column_slices(k, world)[rank]
        self.cols = np.arange(self.k0, self.k1)
        self.links = np.asarray(links, dtype=np.int64)
        self.tl = np.asarray(tl, dtype=np.float64)
        self.val = np.asarray(validation_sorted, dtype=np.int64)
        self.gamma = np.array(gamma)[:, self.k0:self.k1].copy()
        self.lam = np.array(lam)[self.k0:self.k1].copy()
        self.elogbeta = digamma(self.lam) - digamma(self.lam.sum(1))[:, None]
        self.mphi = np.zeros_like(self.gamma)
        self.elogpi = None
        self.conv = np.zeros(n, dtype=np.int64)
        self.active = np.zeros(n, dtype=np.int64)
        self.annealing, self.iter, self.nh, self.prev_h = True, 0, 0, -2147483647.0
        self.rows = []
        L, V = len(self.links), len(self.val)
        self.buf = {_svils.KSH_DEN: torch.zeros(L, dtype=torch.float64), _svils.KSH_ROWX: torch.zeros(3 * n, dtype=torch.float64),
                    _svils.KSH_Q2: torch.zeros(k, dtype=torch.float64), _svils.KSH_VDOT: torch.zeros(V, dtype=torch.float64)}
        self.engine = _Engine(self)
        self.stream = None
        self.log_domain = False

    def owns(self, c):
        return (c >= self.k0) & (c < self.k1)

    def phase(self, ph):
        n, K, alpha = self.n, self.K, self.alpha
        p, q = self.links[:, 0], self.links[:, 1]
        if ph == _svils.KPHASE_INIT_ROWS:
            self.buf[_svils.KSH_ROWX].view(n, 3)[:, 0] = torch.from_numpy(self.gamma.sum(1))
            return
        if ph == _svils.KPHASE_INIT_EXPAND:
            rs = self.buf[_svils.KSH_ROWX].view(n, 3)[:, 0].numpy()
            self.elogpi = digamma(self.gamma) - digamma(rs)[:, None]
            return
        pc, qc = self.conv[p], self.conv[q]
        sh1, sh2 = (pc != 0) & (qc == 0), (qc != 0) & (pc == 0)
        dense = ~(sh1 | sh2)
        pd, qd = p[dense], q[dense]
        if ph == _svils.KPHASE_DEN:
            self.e = np.exp(self.elogpi[pd] + self.elogpi[qd] + self.elogbeta[:, 0])
            den = np.zeros(len(self.links))
            den[dense] = self.e.sum(1)
            self.buf[_svils.KSH_DEN].copy_(torch.from_numpy(den))
        elif ph == _svils.KPHASE_PHI:
            S = self.buf[_svils.KSH_DEN].numpy()[dense]
            phi = self.e / S[:, None]
            kl = len(self.cols)
            gn = np.full((n, kl), alpha)
            np.add.at(gn, pd, phi)
            np.add.at(gn, qd, phi)
            ssum = 2.0 * phi.sum(0)
            c = np.where(sh1, pc, qc) - 1
            mine = (sh1 | sh2) & self.owns(c)
            np.add.at(gn, (p[mine], c[mine] - self.k0), 1.0)
            np.add.at(gn, (q[mine], c[mine] - self.k0), 1.0)
            np.add.at(ssum, c[mine] - self.k0, 2.0)
            has = self.tl > 0
            m = (gn[has] - alpha) / self.tl[has, None]
            self.mphi[has] = m
            self.s1, self.s2, self.ssum = m.sum(0), (m * m).sum(0), ssum
            gn[has] += (n - self.tl[has, None] - 1.0) * m
            if self.annealing:
                gn[has] *= self.E / ssum
            self.gamma = gn
            act = gn - alpha >= 1.0
            rx = np.stack([gn.sum(1), act.sum(1).astype(np.float64), (act * (self.cols + 1.0)).sum(1)], 1)
            self.buf[_svils.KSH_ROWX].copy_(torch.from_numpy(rx.reshape(-1)))
            self.link_counts = (int(dense.sum()), 0, int((sh1 | sh2).sum()))
        elif ph == _svils.KPHASE_FIN:
            rx = self.buf[_svils.KSH_ROWX].view(n, 3).numpy()
            rowsum, active, idx = rx[:, 0], rx[:, 1].astype(np.int64), rx[:, 2].astype(np.int64)
            self.rowsum = rowsum.copy()
            # s3 with the flags of the previous sweep, Q2 across the slice edge
            self.s3 = (self.mphi[pd] * self.mphi[qd]).sum(0)
            q2 = np.zeros(K)
            for sel, cc, other in ((sh1, pc, q), (sh2, qc, p)):
                col = cc[sel]
                ok = (col < K) & self.owns(col)
                np.add.at(q2, col[ok] - 1, self.mphi[other[sel][ok], col[ok] - self.k0])
            self.buf[_svils.KSH_Q2].copy_(torch.from_numpy(q2))
            self.elogpi = digamma(self.gamma) - digamma(rowsum)[:, None]
            self.conv = np.where(active == 1, idx, self.conv)
            self.active = active
        elif ph == _svils.KPHASE_LAMBDA:
            q2 = self.buf[_svils.KSH_Q2].numpy()
            self.lam = np.stack([self.eta0 + self.ssum, self.eta1 + (self.s1 * self.s1 - self.s2 - (self.s3 + q2[self.cols]))], 1)
            self.elogbeta = digamma(self.lam) - digamma(self.lam.sum(1))[:, None]
            vp, vq = self.val[:, 0], self.val[:, 1]
            dots = (self.gamma[vp] * self.gamma[vq] * (self.lam[:, 0] / self.lam.sum(1))).sum(1)
            self.buf[_svils.KSH_VDOT].copy_(torch.from_numpy(dots))
        elif ph == _svils.KPHASE_STOP:
            vp, vq, vy = self.val[:, 0], self.val[:, 1], self.val[:, 2]
            pq = self.buf[_svils.KSH_VDOT].numpy() / (self.rowsum[vp] * self.rowsum[vq])
            u = np.log(np.maximum(np.where(vy != 0, pq, 1.0 - pq), 1e-30))
            a = self.zeros_prob * u[vy == 0].mean() + self.ones_prob * u[vy != 0].mean()
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
