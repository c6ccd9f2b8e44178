"""K-sharded sweeps over the GPUs of one node: every rank keeps the columns [k0, k1) of ALL rows.

The node-block layout (sharded.py) replicates the n-by-k state and all-gathers the gamma rows every sweep
(n*k*8 bytes).  Here the state is split by columns instead; the columns of a row are coupled in four places
only, each a buffer of partials SUMmed over the ranks between two phases (DESIGN.md section 6,
svinet_amd/csrc/svils_ksh.h, tests/test_ksharded_protocol.py):

    DEN -> SUM den[L] -> PHI -> SUM rowx[3n] -> FIN -> SUM q2v[K] -> LAMBDA -> SUM vdot[V] -> STOP
    (log-domain mode, the default above K = 700: DENMAX -> MAX dmax[L] first, then the same)

The native driver is svils_comm_init + svils_ksh_init_state + svils_sweep_ksharded (RCCL all-reduces on the
engine's stream).  This module is the caller-driven form: `KShard` wraps one engine and exposes its exchange
buffers as torch tensors, `sweep_virtual` runs several of them in one process (tests), `KShardedSweep` runs
one per process over torch.distributed.
"""
import numpy as np

from . import _svils
from .sharded import _as_tensor


def column_slices(k, world):
    """contiguous, as even as possible: rank r holds [b[r], b[r+1])"""
    b = [(k * r) // world for r in range(world + 1)]
    return [(b[r], b[r + 1]) for r in range(world)]


class KShard:
    def __init__(self, setup, rank, world, device_index=0, log_domain=None, **engine_kw):
        """log_domain: None = the library's default (on above K = 700), True / False forces it"""
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        self.k0, self.k1 = column_slices(setup.k, world)[rank]
        from ._svils import Engine
        args = dict(ones=setup.ones, ones_prob=setup.ones_prob, eta=setup.eta,
                    link_thresh=setup.link_thresh, lt_min_deg=setup.lt_min_deg, device=device_index,
                    k_slice=(self.k0, self.k1))
        args.update(engine_kw)
        self.device_index = device_index
        self.engine = e = Engine(setup.n, setup.k, **args)
        e.set_graph(setup.links)
        e.set_validation(setup.validation_sorted)
        if getattr(setup, "host_gamma", True):
            e.set_state(np.ascontiguousarray(setup.gamma[:, self.k0:self.k1]), np.ascontiguousarray(setup.lam[self.k0:self.k1]))
        else:   # init_gamma2 on the device: this rank's column slice of every link's draws (svils_init_gamma)
            setup.device_init(e, lam=np.ascontiguousarray(setup.lam[self.k0:self.k1]))
        dev = torch.device("cuda", device_index)
        self.stream = torch.cuda.ExternalStream(e.stream(), device=dev)
        self.buf = {}
        if log_domain is not None:
            e.ksh_log_domain(log_domain)
        self.log_domain = e.ksh_log_domain()
        for which in (_svils.KSH_DEN, _svils.KSH_ROWX, _svils.KSH_Q2, _svils.KSH_VDOT, _svils.KSH_DMAX, _svils.KSH_EARG):
            p, n = e.ksh_buffer(which)
            self.buf[which] = _as_tensor(torch, p, 8 * n, "<f8", dev) if n else None


_ORDER = ((_svils.KPHASE_DEN, _svils.KSH_DEN), (_svils.KPHASE_PHI, _svils.KSH_ROWX), (_svils.KPHASE_FIN, _svils.KSH_Q2),
          (_svils.KPHASE_LAMBDA, _svils.KSH_VDOT), (_svils.KPHASE_STOP, None))


def _order(shard):
    """(phase, buffers to reduce after it) of one sweep; the log-domain mode exchanges the per-link max first, and a handle
    with link_thresh < 1/2 (argmax tagging) the lowest column attaining it (MIN) next to the denominators"""
    order = [(ph, (buf,) if buf is not None else ()) for ph, buf in _ORDER]
    if shard.buf.get(_svils.KSH_EARG) is not None:
        order[0] = (order[0][0], (_svils.KSH_DEN, _svils.KSH_EARG))
    if shard.log_domain:
        order.insert(0, (_svils.KPHASE_DENMAX, (_svils.KSH_DMAX,)))
    return order


def _sum_virtual(shards, which):
    ts = [s.buf[which] for s in shards]
    if ts[0] is None:
        return
    for s in shards:
        s.engine.synchronize()
    tot = ts[0].clone()
    for t in ts[1:]:
        if which == _svils.KSH_DMAX:
            tot = shards[0].torch.maximum(tot, t)
        elif which == _svils.KSH_EARG:
            tot = shards[0].torch.minimum(tot, t)
        else:
            tot += t
    for t in ts:
        t.copy_(tot)
    shards[0].torch.cuda.synchronize()


def init_virtual(shards):
    for s in shards:
        s.engine.ksweep_phase(_svils.KPHASE_INIT_ROWS)
    _sum_virtual(shards, _svils.KSH_ROWX)
    for s in shards:
        s.engine.ksweep_phase(_svils.KPHASE_INIT_EXPAND)


def sweep_virtual(shards, nsweeps=1):
    """all ranks in ONE process (tests): the exchanges are plain tensor sums in rank order"""
    for _ in range(nsweeps):
        for phase, bufs in _order(shards[0]):
            for s in shards:
                s.engine.ksweep_phase(phase)
            for which in bufs:
                _sum_virtual(shards, which)


def _window_tensor(shard, which):
    """the window's share of an exchange buffer while a mini-batch step is open (svils_ksh_buffer_ptr returns the
    sub-range; the tensors cached at construction alias the whole buffers)"""
    p, n = shard.engine.ksh_buffer(which)
    if not n:
        return None
    dev = shard.torch.device("cuda", shard.device_index)
    return _as_tensor(shard.torch, p, 8 * n, "<f8", dev)


def _reduce_virtual(shards, which, tensors):
    if tensors[0] is None:
        return
    torch = shards[0].torch
    for s in shards:
        s.engine.synchronize()
    tot = tensors[0].clone()
    for t in tensors[1:]:
        if which == _svils.KSH_DMAX:
            tot = torch.maximum(tot, t)
        elif which == _svils.KSH_EARG:
            tot = torch.minimum(tot, t)
        else:
            tot += t
    for t in tensors:
        t.copy_(tot)
    torch.cuda.synchronize()


def step_virtual(shards, nsteps=1):
    """mini-batch steps (svils_set_stochastic on every shard first) with all ranks in ONE process (tests): the phases of
    a sweep over the window every rank shares, the exchanges restricted to the window's share of the buffers"""
    for _ in range(nsteps):
        for phase, bufs in _order(shards[0]):
            for s in shards:
                s.engine.ksweep_phase(phase)
            for which in bufs:
                _reduce_virtual(shards, which, [_window_tensor(s, which) for s in shards])


class KShardedSweep:
    """one rank per process: the exchanges are torch.distributed all-reduces on the engine's stream"""

    def __init__(self, shard, dist, group=None):
        self.s, self.dist, self.group = shard, dist, group

    def _sum(self, which):
        t = self.s.buf[which]
        if t is None:
            return
        torch = self.s.torch
        op = (self.dist.ReduceOp.MAX if which == _svils.KSH_DMAX else
              self.dist.ReduceOp.MIN if which == _svils.KSH_EARG else self.dist.ReduceOp.SUM)
        if self.dist.get_backend(self.group) == "gloo":
            self.s.engine.synchronize()
            h = t.cpu() if t.is_cuda else t
            self.dist.all_reduce(h, op=op, group=self.group)
            if t.is_cuda:
                t.copy_(h)
                torch.cuda.synchronize()
        else:
            with torch.cuda.stream(self.s.stream):
                self.dist.all_reduce(t, op=op, group=self.group)

    def init(self):
        self.s.engine.ksweep_phase(_svils.KPHASE_INIT_ROWS)
        self._sum(_svils.KSH_ROWX)
        self.s.engine.ksweep_phase(_svils.KPHASE_INIT_EXPAND)

    def sweep(self, nsweeps=1):
        for _ in range(nsweeps):
            for phase, bufs in _order(self.s):
                self.s.engine.ksweep_phase(phase)
                for which in bufs:
                    self._sum(which)


class KShardedStep(KShardedSweep):
    """mini-batch steps, one rank per process (the caller-driven form of svils_step_ksharded)"""

    def step(self, nsteps=1):
        torch = self.s.torch
        for _ in range(nsteps):
            for phase, bufs in _order(self.s):
                self.s.engine.ksweep_phase(phase)
                for which in bufs:
                    t = _window_tensor(self.s, which)
                    if t is None:
                        continue
                    op = (self.dist.ReduceOp.MAX if which == _svils.KSH_DMAX else
                          self.dist.ReduceOp.MIN if which == _svils.KSH_EARG else self.dist.ReduceOp.SUM)
                    if self.dist.get_backend(self.group) == "gloo":
                        self.s.engine.synchronize()
                        h = t.cpu()
                        self.dist.all_reduce(h, op=op, group=self.group)
                        t.copy_(h)
                        torch.cuda.synchronize()
                    else:
                        with torch.cuda.stream(self.s.stream):
                            self.dist.all_reduce(t, op=op, group=self.group)
