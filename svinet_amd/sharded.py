"""Node-block sharding of the link-sampling sweep over the GPUs of one node.

One process per GPU.  Rank r owns the node block [r*B, (r+1)*B), B = ceil(n/G):
it evaluates phi for its nodes' CSR rows (pull-style: every link is evaluated
once per endpoint, so no floating-point scatter crosses GPUs) and finalises
its rows.  The reference has no distributed path; the exchange below is the
multi-GPU form of the sums inside LinkSampling::infer()
(src/linksampling.cc:605-761):

  phase A  phi pass over owned rows      -> all-reduce(SUM)  `sum[k]`           (K doubles)
  phase B  mean indicators, new gamma,   -> all-gather by node block of the gamma rows (ONE n-by-k
           Elogpi, prune over owned rows    array) and of the packed flags (converged, active count,
                                            active-set mask: ONE small buffer, SVILS_BUF_XFLAGS)
  EXPAND   flags of the other ranks' rows unpacked; their Elogpi = psi(gamma)-psi(sum) and
           m = (gamma/scale - alpha)/(n-1) re-derived locally from the gathered gamma (no exchange)
  phase C  s3 pass over owned upper rows -> all-reduce(SUM)  s1,s2,s3          (3K doubles)
  phase D  lambda, likelihood, stop rule    (replicated, identical on every rank)

Two drivers run this protocol.  The native one lives in the library
(svils_comm_init / svils_sweep_sharded: RCCL calls on the engine's stream, no Python
between the phases) and is what bench.py and the C++ CLI (`-gpus N`) use.  The one in
this file issues the same exchanges as `torch.distributed` calls (backend "nccl" ==
RCCL over xGMI on ROCm; "gloo" for the CPU protocol tests and for several ranks on one
GPU) on tensors that alias the engine's device buffers, on the engine's own HIP stream;
it also drives the mini-batch steps.
"""
import numpy as np

from . import _svils


def block_size(n, world):
    return (n + world - 1) // world


def node_block(n, world, rank):
    b = block_size(n, world)
    return min(rank * b, n), min((rank + 1) * b, n)


class _DevArray:
    """__cuda_array_interface__ view of a raw device pointer (no ownership)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _as_tensor(torch, ptr, nbytes, dtype_str, device):
    itemsize = {"<f8": 8, "<u4": 4, "<u8": 8, "<i4": 4, "<i8": 8}[dtype_str]
    arr = _DevArray(ptr, (nbytes // itemsize,), dtype_str)
    return torch.as_tensor(arr, device=device)


class HipShard:
    """Adapter: an svils Engine restricted to this rank's node block, plus torch
    tensors aliasing its exchange buffers."""

    def __init__(self, setup, rank, world, device_index, **engine_kw):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        n = setup.n
        self.B = block_size(n, world)
        self.n_alloc = self.B * world
        self.engine = setup.engine(device=device_index, node_block=node_block(n, world, rank),
                                   n_alloc=self.n_alloc, **engine_kw)
        dev = torch.device("cuda", device_index)
        self.stream = torch.cuda.ExternalStream(self.engine.stream(), device=dev)
        e = self.engine

        def t(which, ts):
            p, nb, rb = e.device_buffer(which)
            return _as_tensor(torch, p, nb, ts, dev), rb

        self.kvec_a, _ = t(_svils.BUF_KVEC_A, "<f8")
        self.kvec_c, _ = t(_svils.BUF_KVEC_C, "<f8")
        self.rows = []
        for which in (_svils.BUF_GAMMA,):     # Elogpi / mphi are re-derived by PHASE_EXPAND
            ten, rb = t(which, "<f8")
            self.rows.append(ten.view(self.n_alloc, rb // 8))
        ten, rb = t(_svils.BUF_MPHI, "<f8")   # exchanged only by mini-batch steps
        self.mphi = ten.view(self.n_alloc, rb // 8)
        mem, rb = t(_svils.BUF_MEMBER, "<i8")
        self.member = mem.view(self.n_alloc, rb // 8)
        # converged flag (the half prune() is writing), active count and active-set mask of every row,
        # packed on the device: the host needs no mirror of the device's buffer parity
        xf, rb = t(_svils.BUF_XFLAGS, "<i4")
        self.xflags = xf.view(self.n_alloc, rb // 4)

    def phase(self, ph):
        self.engine.sweep_phase(ph)

    def annealing(self):
        return self.engine.control().annealing != 0   # synchronises the engine's stream

    def gather_list(self):
        return self.rows + [self.xflags]

    def end_sweep(self):
        pass


class ShardedSweep:
    """Runs sweeps over `shard` (HipShard or a test double with the same
    surface), doing the exchanges with `dist` (torch.distributed)."""

    def __init__(self, shard, dist, group=None):
        self.s, self.dist, self.group = shard, dist, group
        self.world = shard.world

    def _allreduce(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _allgather_rows(self, t):
        if self.world > 1:
            B, r = self.s.B, self.s.rank
            mine = t[r * B:(r + 1) * B]
            if t.is_cuda:
                self.dist.all_gather_into_tensor(t, mine, group=self.group)   # in place
            else:
                self.dist.all_gather_into_tensor(t, mine.clone(), group=self.group)

    def _ctx(self):
        st = getattr(self.s, "stream", None)
        if st is None:
            import contextlib
            return contextlib.nullcontext()
        return self.s.torch.cuda.stream(st)

    def _annealing(self):
        """the replicated annealing flag (a shard without one is treated as always annealing)"""
        f = getattr(self.s, "annealing", None)
        return True if f is None else bool(f() if callable(f) else f)

    def sweep(self, nsweeps=1):
        """`sum[k]` is read between the phi pass and the finalise pass only while annealing (the ones/sum[k]
        scale, src/linksampling.cc:542); afterwards its one reader is lambda[k][0] in the tail, so its
        all-reduce moves next to the one of s1,s2,s3 (as svils_sweep_sharded does: one exchange point fewer per
        sweep).  The flag only goes from 1 to 0 inside a run and is the same on every rank; it is looked at
        when a call starts and every 16 sweeps until it is off."""
        s = self.s
        annealing = True
        with self._ctx():
            for i in range(nsweeps):
                if annealing and i % 16 == 0:
                    annealing = self._annealing()
                s.phase(_svils.PHASE_A)
                if annealing:
                    self._allreduce(s.kvec_a)
                s.phase(_svils.PHASE_B)
                for t in s.gather_list():
                    self._allgather_rows(t)
                s.phase(_svils.PHASE_EXPAND)
                s.phase(_svils.PHASE_C)
                if not annealing:
                    self._allreduce(s.kvec_a)
                self._allreduce(s.kvec_c)
                s.phase(_svils.PHASE_D)
                s.end_sweep()

    def gather_communities(self):
        with self._ctx():
            self._allgather_rows(self.s.member)


class ShardedStep(ShardedSweep):
    """Mini-batch (Robbins-Monro) steps over node-block shards: every rank takes the window at the same
    offset inside its own block (svils_step_window), so one step updates world x batch_nodes nodes.
    Exchanges per step: all-reduce of `sum[k]` (K doubles); all-gather of the WINDOWS' gamma and mphi
    rows and flags (world x batch_nodes rows, not n); all-reduce of s1,s2,s3 (3K doubles) -- the
    "K-vector lambda and touched gamma rows" of the global step.  The engines must have been put in
    mini-batch mode with shard_block = HipShard.B."""

    def _allgather_window(self, t, b, e):
        if self.world > 1 and e > b:
            B, r = self.s.B, self.s.rank
            outs = [t[q * B + b:q * B + e] for q in range(self.world)]
            mine = outs[r]
            self.dist.all_gather(outs, mine.clone(), group=self.group)   # the input aliases outs[r]

    def step(self, nsteps=1):
        s = self.s
        eng = s.engine
        with self._ctx():
            for _ in range(nsteps):
                eng.step_phase(_svils.PHASE_A)
                self._allreduce(s.kvec_a)
                eng.step_phase(_svils.PHASE_B)
                b, e = eng.step_window()
                for t in s.rows + [s.mphi, s.xflags]:
                    self._allgather_window(t, b, e)
                eng.step_phase(_svils.PHASE_EXPAND)
                eng.step_phase(_svils.PHASE_C)
                self._allreduce(s.kvec_c)
                eng.step_phase(_svils.PHASE_D)
                s.end_sweep()
