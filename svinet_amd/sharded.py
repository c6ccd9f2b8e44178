"""Node-block sharding of the link-sampling sweep over the GPUs of one node.

One process per GPU.  Rank r owns the CONTIGUOUS node block [bounds[r], bounds[r+1]); the blocks carry
equal WORK (CSR entries + a per-node share: svils_balance_node_blocks), not equal node counts -- the reference
numbers nodes by first appearance (src/network.cc:10-116), so on its example graphs the hubs sit at the low ids
and equal-count blocks would give rank 0 of 8 on ca-AstroPh three times the mean work.  A rank evaluates phi for its
nodes' CSR rows (pull-style: every link is evaluated once per endpoint, so no floating-point scatter crosses GPUs).
The reference has no distributed path; the exchange below is the multi-GPU form of the sums inside
LinkSampling::infer() (src/linksampling.cc:605-761):

  phase A        phi pass over owned rows                       -> partial `sum[k]`
  phase B_LIGHT  mean indicators, s1 / s2 partials, tags, the UNSCALED new rows of the owned block
                 -> exchange 1: all-reduce(SUM) `sum[k]` (K doubles) + all-gather of the staged rows
                    (ONE n-by-k array, slices padded to the largest block)
  EXPAND_ALL     every row, owned or not: annealing scale ones/sum[k], gamma, Elogpi = psi(gamma)-psi(sum),
                 m = (row - alpha)/(n-1) of the other blocks, prune() flags -- recomputed by every rank from the
                 same bytes, so flags are not exchanged
  phase C        s3 pass over this rank's share of the links     -> exchange 2: all-reduce(SUM) s1,s2,s3 (3K doubles)
  phase D        lambda, likelihood, stop rule                    (replicated, identical on every rank)

Two exchange points in both phases of a run (the annealing scale is applied behind the row exchange).

Two drivers run this protocol.  The native one lives in the library
(svils_comm_init / svils_sweep_sharded: RCCL calls on the engine's stream, no Python
between the phases, whole runs of sweeps replayed as hipGraphs) and is what bench.py and the C++ CLI
(`-gpus N`) use.  The one in this file issues the same exchanges as `torch.distributed` calls (backend "nccl" ==
RCCL over xGMI on ROCm; "gloo" for the CPU protocol tests and for several ranks on one
GPU) on tensors that alias the engine's device buffers, on the engine's own HIP stream;
it also drives the mini-batch steps (equal blocks, rows exchanged in place).
"""
import numpy as np

from . import _svils


def block_size(n, world):
    return (n + world - 1) // world


def node_block(n, world, rank):
    """the EQUAL blocks (mini-batch steps; what svils_comm_init assumes when no bounds were declared)"""
    b = block_size(n, world)
    return min(rank * b, n), min((rank + 1) * b, n)


def equal_bounds(n, world):
    return np.array([min(r * block_size(n, world), n) for r in range(world + 1)], dtype=np.uint32)


def balanced_bounds(links, n, world, node_weight=-1.0):
    """bounds[world + 1] of the work-balanced blocks: the library's own cut (svils_balance_node_blocks; host code,
    needs no device), so that every driver -- this file, bench.py, the C++ CLI -- owns the same rows"""
    return _svils.balance_node_blocks(links, n, world, node_weight)


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
    tensors aliasing its exchange buffers.  bounds: the node blocks of all ranks (default: balanced by work);
    equal=True takes the equal blocks mini-batch steps need (and allocates their in-place exchange rows)."""

    def __init__(self, setup, rank, world, device_index, bounds=None, equal=False, **engine_kw):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        n = setup.n
        self.B = block_size(n, world)
        if bounds is None:
            bounds = equal_bounds(n, world) if equal else balanced_bounds(setup.links, n, world)
        self.bounds = np.asarray(bounds, dtype=np.uint32)
        self.bmax = int(np.max(np.diff(self.bounds.astype(np.int64)))) if world else n
        self.n_alloc = self.B * world if equal else n
        self.engine = setup.engine(device=device_index, node_block=(int(self.bounds[rank]), int(self.bounds[rank + 1])),
                                   n_alloc=self.n_alloc, **engine_kw)
        # equal blocks stay "undeclared" (svils_comm_init's default): the handle may then run mini-batch steps too
        self.engine.set_node_blocks(rank, world, None if equal else self.bounds)
        dev = torch.device("cuda", device_index)
        self.stream = torch.cuda.ExternalStream(self.engine.stream(), device=dev)
        e = self.engine

        def t(which, ts):
            p, nb, rb = e.device_buffer(which)
            return _as_tensor(torch, p, nb, ts, dev), rb

        self.kvec_a, _ = t(_svils.BUF_KVEC_A, "<f8")
        self.kvec_c, _ = t(_svils.BUF_KVEC_C, "<f8")
        ten, rb = t(_svils.BUF_GSTAGE, "<f8")          # [world * bmax][ld]: slice r = rank r's unscaled new rows
        self.gstage = ten.view(world * max(self.bmax, 1), rb // 8)
        self.rows = []
        for which in (_svils.BUF_GAMMA,):     # exchanged in place by mini-batch steps only
            ten, rb = t(which, "<f8")
            self.rows.append(ten.view(self.n_alloc, rb // 8))
        ten, rb = t(_svils.BUF_MPHI, "<f8")   # exchanged only by mini-batch steps
        self.mphi = ten.view(self.n_alloc, rb // 8)
        mem, rb = t(_svils.BUF_MEMBER, "<i8")
        self.member = mem.view(self.n_alloc, rb // 8)
        # converged flag (the half prune() is writing), active count and active-set mask of every row,
        # packed on the device (mini-batch steps)
        xf, rb = t(_svils.BUF_XFLAGS, "<i4")
        self.xflags = xf.view(self.n_alloc, rb // 4)

    def phase(self, ph):
        self.engine.sweep_phase(ph)

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

    def _allgather_staged(self, t):
        """the slices of the staging buffer ([world * bmax][ld]); in place: the send block IS slice `rank`"""
        if self.world > 1:
            bm, r = self.s.bmax, self.s.rank
            mine = t[r * bm:(r + 1) * bm]
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

    def sweep(self, nsweeps=1):
        """two exchange points per sweep, the same in both phases of a run: `sum[k]` travels with the rows and the
        annealing scale ones / sum[k] (src/linksampling.cc:542) is applied behind the exchange, by EXPAND_ALL"""
        s = self.s
        with self._ctx():
            for _ in range(nsweeps):
                s.phase(_svils.PHASE_A)
                s.phase(_svils.PHASE_B_LIGHT)
                self._allreduce(s.kvec_a)
                self._allgather_staged(s.gstage)
                s.phase(_svils.PHASE_EXPAND_ALL)
                s.phase(_svils.PHASE_C)
                self._allreduce(s.kvec_c)
                s.phase(_svils.PHASE_D)
                s.end_sweep()

    def gather_communities(self):
        """every block's rows of the community bitmask (blocks differ in size: one broadcast per owner)"""
        if self.world <= 1:
            return
        with self._ctx():
            b = self.s.bounds
            for r in range(self.world):
                if int(b[r + 1]) > int(b[r]):
                    self.dist.broadcast(self.s.member[int(b[r]):int(b[r + 1])], src=r, group=self.group)


class ShardedStep(ShardedSweep):
    """Mini-batch (Robbins-Monro) steps over node-block shards: every rank takes the window at the same
    offset inside its own block (svils_step_window), so one step updates world x batch_nodes nodes.
    Exchanges per step: all-reduce of `sum[k]` (K doubles); all-gather of the WINDOWS' gamma and mphi
    rows and flags (world x batch_nodes rows, not n); all-reduce of s1,s2,s3 (3K doubles) -- the
    "K-vector lambda and touched gamma rows" of the global step.  The engines must have been put in
    mini-batch mode with shard_block = HipShard.B, on EQUAL blocks (HipShard(..., equal=True))."""

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
