"""ctypes binding of the C ABI in include/svils.h (libsvils.so, HIP/gfx950).

There is no CPU fallback: if the shared library is missing this module raises
at load time, and if no HIP device is present `Engine(...)` raises
`SvilsError` (SVILS_ERR_DEVICE).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVILS_LIB", os.path.join(_HERE, "lib", "libsvils.so"))  # SVILS_LIB: A/B kernel builds

KERNEL_NAMES = ("phi", "reduce_sum", "finalize", "s3", "validation", "reduce_s", "tail", "classify", "exchange")
KERNEL_PHI = 0
KERNEL_EXCHANGE = 8

# every symbol include/svils.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "svils_config_default", "svils_create", "svils_destroy", "svils_set_graph",
    "svils_set_validation", "svils_set_state", "svils_get_control", "svils_set_control",
    "svils_validation_row", "svils_sweep", "svils_synchronize", "svils_get_rows",
    "svils_get_state", "svils_get_communities", "svils_get_aux", "svils_enable_timing",
    "svils_get_timing", "svils_kernel_name", "svils_sweep_phase", "svils_device_buffer",
    "svils_stream", "svils_last_error", "svils_abi_version", "svils_debug_eval",
    "svils_set_timing_period", "svils_set_stochastic", "svils_step", "svils_stochastic_default", "svils_step_phase", "svils_step_window",
    "svils_get_sweep_stats", "svils_get_timed_links",
    "svils_comm_unique_id", "svils_comm_init", "svils_sweep_sharded", "svils_gather_communities",
    "svils_ksweep_phase", "svils_ksh_buffer_ptr", "svils_ksh_init_state", "svils_sweep_ksharded", "svils_ksh_log_domain",
    "svils_comm_allgather_host", "svils_step_sharded", "svils_step_ksharded", "svils_comm_query",
    "svils_report_enqueue", "svils_report_ready", "svils_report_fetch", "svils_report_test_rows",
    "svils_set_test", "svils_get_test_rows",
    "svils_report_tag_count", "svils_report_fetch_tags", "svils_get_community_tags",
    "svils_set_node_blocks", "svils_balance_node_blocks", "svils_prepare_graphs",
    "svils_set_option", "svils_get_option", "svils_option_table", "svils_init_gamma",
)


class SvilsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("svils error %d: %s" % (code, msg))
        self.code = code


class Stochastic(C.Structure):
    _fields_ = [("batch_nodes", C.c_uint32), ("node_tau0", C.c_double), ("node_kappa", C.c_double),
                ("tau0", C.c_double), ("kappa", C.c_double), ("seed", C.c_uint64), ("shard_block", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("k", C.c_uint32), ("ones", C.c_uint64),
        ("alpha", C.c_double), ("eta0", C.c_double), ("eta1", C.c_double),
        ("epsilon", C.c_double), ("link_thresh", C.c_double),
        ("lt_min_deg", C.c_uint32), ("reportfreq", C.c_uint32),
        ("use_validation_stop", C.c_int32),
        ("ones_prob", C.c_double), ("zeros_prob", C.c_double),
        ("device", C.c_int32), ("node_begin", C.c_uint32), ("node_end", C.c_uint32),
        ("n_alloc", C.c_uint32), ("sparse_after_iter", C.c_int32),
        ("k_begin", C.c_uint32), ("k_total", C.c_uint32),
    ]


class Control(C.Structure):
    _fields_ = [
        ("iter", C.c_uint32), ("annealing", C.c_int32), ("write_comm", C.c_int32),
        ("nh", C.c_int32), ("prev_h", C.c_double), ("max_h", C.c_double),
        ("stopped", C.c_int32), ("why", C.c_int32), ("sweeps_done", C.c_uint32),
        ("rows", C.c_uint32), ("links_dense", C.c_uint64), ("links_sparse", C.c_uint64),
        ("links_shortcut", C.c_uint64),
    ]


class CommInfo(C.Structure):
    _fields_ = [("nranks", C.c_int32), ("rank", C.c_int32), ("device", C.c_int32), ("version", C.c_int32),
                ("row_comm", C.c_int32), ("pci_bus_id", C.c_char * 32), ("library", C.c_char * 256)]


_lib = None


def load():
    """dlopen libsvils.so; raises OSError (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `python -m svinet_amd.build` "
                      "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.svils_last_error.restype = C.c_char_p
    L.svils_kernel_name.restype = C.c_char_p
    L.svils_kernel_name.argtypes = [C.c_int]
    L.svils_abi_version.restype = C.c_int
    L.svils_config_default.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint32]
    L.svils_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.svils_destroy.argtypes = [vp]
    L.svils_set_graph.argtypes = [vp, vp, C.c_uint64]
    L.svils_set_validation.argtypes = [vp, vp, C.c_uint64]
    L.svils_set_state.argtypes = [vp, vp, vp, vp]
    L.svils_get_control.argtypes = [vp, C.POINTER(Control)]
    L.svils_set_control.argtypes = [vp, C.POINTER(Control)]
    L.svils_validation_row.argtypes = [vp, vp]
    L.svils_sweep.argtypes = [vp, C.c_uint32]
    L.svils_set_stochastic.argtypes = [vp, C.POINTER(Stochastic)]
    L.svils_step.argtypes = [vp, C.c_uint32]
    L.svils_step_phase.argtypes = [vp, C.c_int]
    L.svils_step_window.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.svils_stochastic_default.argtypes = [C.POINTER(Stochastic), C.c_uint32]
    L.svils_stochastic_default.restype = None
    L.svils_synchronize.argtypes = [vp]
    L.svils_get_rows.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.svils_get_state.argtypes = [vp, vp, vp, vp]
    L.svils_get_communities.argtypes = [vp, vp]
    L.svils_get_aux.argtypes = [vp, C.c_int, vp]
    L.svils_enable_timing.argtypes = [vp, C.c_uint32]
    L.svils_get_timing.argtypes = [vp, vp, vp]
    L.svils_sweep_phase.argtypes = [vp, C.c_int]
    L.svils_device_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t)]
    L.svils_stream.argtypes = [vp, C.POINTER(vp)]
    L.svils_debug_eval.argtypes = [vp, C.c_int, vp, vp, C.c_uint32]
    L.svils_set_timing_period.argtypes = [vp, C.c_uint32]
    L.svils_get_sweep_stats.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.svils_get_timed_links.argtypes = [vp, vp]
    L.svils_comm_unique_id.argtypes = [vp]
    L.svils_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.svils_set_node_blocks.argtypes = [vp, C.c_int, C.c_int, vp]
    L.svils_prepare_graphs.argtypes = [vp, C.c_uint32]
    L.svils_balance_node_blocks.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int, C.c_double, vp]
    L.svils_sweep_sharded.argtypes = [vp, C.c_uint32]
    L.svils_gather_communities.argtypes = [vp]
    L.svils_comm_allgather_host.argtypes = [vp, vp, vp, C.c_size_t]
    L.svils_step_sharded.argtypes = [vp, C.c_uint32]
    L.svils_comm_query.argtypes = [vp, C.POINTER(CommInfo)]
    L.svils_report_enqueue.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
    L.svils_report_ready.argtypes = [vp, C.c_int]
    L.svils_report_fetch.argtypes = [vp, C.c_int, C.POINTER(Control), vp, C.POINTER(C.c_uint32), vp]
    L.svils_report_test_rows.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_uint32)]
    L.svils_set_test.argtypes = [vp, vp, C.c_uint64]
    L.svils_get_test_rows.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.svils_report_tag_count.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    L.svils_report_fetch_tags.argtypes = [vp, C.c_int, C.POINTER(Control), vp, C.POINTER(C.c_uint32), vp, C.c_uint64,
                                          C.POINTER(C.c_uint64)]
    L.svils_get_community_tags.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.svils_init_gamma.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_uint64, vp]
    L.svils_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.svils_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    L.svils_option_table.restype = C.c_char_p
    for name in EXPORTS:
        f = getattr(L, name)
        if name not in ("svils_last_error", "svils_kernel_name", "svils_abi_version", "svils_stochastic_default", "svils_option_table"):
            f.restype = C.c_int
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise SvilsError(rc, load().svils_last_error().decode("utf-8", "replace"))


BUF_KVEC_A, BUF_KVEC_C, BUF_GAMMA, BUF_ELOGPI, BUF_MPHI, BUF_CONV, BUF_ACTIVE, BUF_AMASK, \
    BUF_MEMBER, BUF_XFLAGS, BUF_GSTAGE = range(11)
COMM_ID_BYTES = 128


def comm_unique_id():
    """rank 0: a fresh ncclUniqueId (bytes) for Engine.comm_init on every rank"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(load().svils_comm_unique_id(buf))
    return buf.raw
PHASE_A, PHASE_B, PHASE_C, PHASE_D, PHASE_EXPAND, PHASE_B_LIGHT, PHASE_EXPAND_ALL = range(7)


def balance_node_blocks(links, n, world, node_weight=-1.0):
    """svils_balance_node_blocks: bounds[world + 1] of contiguous node blocks of equal WORK (CSR entries + node_weight per
    node; < 0 = the library's default).  Host code only: works without a device."""
    links = np.ascontiguousarray(links, dtype=np.uint32)
    b = np.zeros(world + 1, dtype=np.uint32)
    _chk(load().svils_balance_node_blocks(links.ctypes.data, links.shape[0], n, world, float(node_weight), b.ctypes.data))
    return b
KPHASE_DEN, KPHASE_PHI, KPHASE_FIN, KPHASE_LAMBDA, KPHASE_STOP, KPHASE_INIT_ROWS, KPHASE_INIT_EXPAND, KPHASE_DENMAX = range(8)
KSH_DEN, KSH_ROWX, KSH_Q2, KSH_VDOT, KSH_DMAX, KSH_EARG = range(6)


class Engine:
    """One svils_handle: the device-resident body of LinkSampling::infer()."""

    def __init__(self, n, k, ones, ones_prob, eta=(1.0, 1.0), link_thresh=0.5, lt_min_deg=0,
                 reportfreq=1, use_validation_stop=True, device=0, node_block=None, n_alloc=0,
                 sparse_after_iter=1000, k_slice=None, options=None):
        """k_slice = (k_begin, k_end): a K-sharded handle holding those columns of the `k` communities
        (gamma / lambda go in and out as that slice; alpha stays 1/k).
        options = {key: value}: rows of the library's option table set on the new handle (svils_set_option)."""
        L = load()
        cfg = Config()
        _chk(L.svils_config_default(C.byref(cfg), n, k))
        cfg.ones = int(ones)
        cfg.eta0, cfg.eta1 = float(eta[0]), float(eta[1])
        cfg.link_thresh = link_thresh
        cfg.lt_min_deg = lt_min_deg
        cfg.reportfreq = reportfreq
        cfg.use_validation_stop = int(use_validation_stop)
        cfg.ones_prob = float(ones_prob)
        cfg.zeros_prob = 1 - float(ones_prob)   # src/linksampling.cc:50
        cfg.device = device
        if node_block is not None:
            cfg.node_begin, cfg.node_end = node_block
        cfg.n_alloc = n_alloc
        cfg.sparse_after_iter = sparse_after_iter
        self.k_total = k
        if k_slice is not None:
            cfg.k_begin, cfg.k_total = int(k_slice[0]), int(k)
            cfg.k = int(k_slice[1]) - int(k_slice[0])
            k = cfg.k
        self.n, self.k = n, k
        self._h = C.c_void_p()
        _chk(L.svils_create(C.byref(cfg), C.byref(self._h)))
        for key, value in (options or {}).items():
            self.set_option(key, value)

    def close(self):
        if getattr(self, "_h", None):
            load().svils_destroy(self._h)
            self._h = None

    __del__ = close

    def set_graph(self, links):
        links = np.ascontiguousarray(links, dtype=np.uint32)
        assert links.ndim == 2 and links.shape[1] == 2
        self.nlinks = links.shape[0]
        _chk(load().svils_set_graph(self._h, links.ctypes.data, links.shape[0]))

    def ksweep_phase(self, phase):
        _chk(load().svils_ksweep_phase(self._h, int(phase)))

    def ksh_buffer(self, which):
        """-> (device pointer, number of doubles) of a K-sharded exchange buffer"""
        p, n = C.c_void_p(), C.c_size_t()
        _chk(load().svils_ksh_buffer_ptr(self._h, int(which), C.byref(p), C.byref(n)))
        return p.value, n.value

    def ksh_log_domain(self, on=None):
        """set (True / False) or query (None) the log-domain denominators of a K-sharded handle"""
        rc = load().svils_ksh_log_domain(self._h, -1 if on is None else int(bool(on)))
        if on is None:
            return bool(rc)
        _chk(rc)

    def ksh_init_state(self):
        _chk(load().svils_ksh_init_state(self._h))

    def step_ksharded(self, nsteps=1):
        """mini-batch steps of a K-sharded handle, the exchanges issued by the library (collective)"""
        _chk(load().svils_step_ksharded(self._h, int(nsteps)))

    def sweep_ksharded(self, nsweeps=1):
        _chk(load().svils_sweep_ksharded(self._h, int(nsweeps)))

    def set_validation(self, pairs_y):
        pairs_y = np.ascontiguousarray(pairs_y, dtype=np.uint32).reshape(-1, 3)
        _chk(load().svils_set_validation(self._h, pairs_y.ctypes.data, pairs_y.shape[0]))

    def set_test(self, pairs_y):
        """-load-test pairs [T][3] = (p, q, y) in map order: a test row per report (include/svils.h)"""
        pairs_y = np.ascontiguousarray(pairs_y, dtype=np.uint32).reshape(-1, 3)
        _chk(load().svils_set_test(self._h, pairs_y.ctypes.data, pairs_y.shape[0]))

    def test_rows(self, first=0, count=None):
        if count is None:
            count = self.control().rows - first
        out = np.zeros((count, 10), dtype=np.float64)
        if count:
            _chk(load().svils_get_test_rows(self._h, first, count, out.ctypes.data))
        return out

    def set_state(self, gamma, lam, converged=None):
        gamma = np.ascontiguousarray(gamma, dtype=np.float64)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        assert gamma.shape == (self.n, self.k) and lam.shape == (self.k, 2)
        cptr = None
        if converged is not None:
            converged = np.ascontiguousarray(converged, dtype=np.uint32)
            assert converged.shape == (self.n,)
            cptr = converged.ctypes.data
        _chk(load().svils_set_state(self._h, gamma.ctypes.data, lam.ctypes.data, cptr))

    def init_gamma(self, edges, mt_states, outputs_per_stream, lam):
        """init_gamma2 on the device (svils_init_gamma): edges [E][2] in drawing order, mt_states [S][624] uint32"""
        edges = np.ascontiguousarray(edges, dtype=np.uint32)
        st = np.ascontiguousarray(mt_states, dtype=np.uint32)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        assert edges.ndim == 2 and edges.shape[1] == 2 and st.ndim == 2 and st.shape[1] == 624 and lam.shape == (self.k, 2)
        _chk(load().svils_init_gamma(self._h, edges.ctypes.data, edges.shape[0], st.ctypes.data, st.shape[0],
                                     int(outputs_per_stream), lam.ctypes.data))

    def control(self):
        c = Control()
        _chk(load().svils_get_control(self._h, C.byref(c)))
        return c

    def set_control(self, **kw):
        c = self.control()
        for key, val in kw.items():
            setattr(c, key, val)
        _chk(load().svils_set_control(self._h, C.byref(c)))

    def validation_row(self):
        row = np.zeros(10, dtype=np.float64)
        _chk(load().svils_validation_row(self._h, row.ctypes.data))
        return row

    def sweep(self, nsweeps=1):
        _chk(load().svils_sweep(self._h, nsweeps))

    def prepare_graphs(self, max_sweeps=16):
        _chk(load().svils_prepare_graphs(self._h, max_sweeps))

    def set_stochastic(self, batch_nodes=0, tau0=1024.0, kappa=0.9, node_tau0=None, node_kappa=None, seed=0,
                       shard_block=0):
        """mini-batch mode (include/svils.h): windows of `batch_nodes` consecutive nodes per step;
        node step sizes default to the lambda ones when not given; shard_block = nodes per rank block on
        a node-block shard (then drive the steps with step_phase)"""
        cfg = Stochastic(batch_nodes, tau0 if node_tau0 is None else node_tau0,
                         kappa if node_kappa is None else node_kappa, tau0, kappa, seed, shard_block)
        _chk(load().svils_set_stochastic(self._h, C.byref(cfg)))

    def step(self, nsteps=1):
        _chk(load().svils_step(self._h, nsteps))

    def step_phase(self, phase):
        _chk(load().svils_step_phase(self._h, phase))

    def step_window(self):
        b, e = C.c_uint32(), C.c_uint32()
        _chk(load().svils_step_window(self._h, C.byref(b), C.byref(e)))
        return b.value, e.value

    def sweep_phase(self, phase):
        _chk(load().svils_sweep_phase(self._h, phase))

    def synchronize(self):
        _chk(load().svils_synchronize(self._h))

    def rows(self, first=0, count=None):
        if count is None:
            count = self.control().rows - first
        out = np.zeros((count, 10), dtype=np.float64)
        if count:
            _chk(load().svils_get_rows(self._h, first, count, out.ctypes.data))
        return out

    def state(self):
        g = np.zeros((self.n, self.k), dtype=np.float64)
        lam = np.zeros((self.k, 2), dtype=np.float64)
        conv = np.zeros(self.n, dtype=np.uint32)
        _chk(load().svils_get_state(self._h, g.ctypes.data, lam.ctypes.data, conv.ctypes.data))
        return g, lam, conv

    def communities(self):
        m = np.zeros((self.n, self.k), dtype=np.uint8)
        _chk(load().svils_get_communities(self._h, m.ctypes.data))
        return m

    def community_tags(self):
        """the same communities as (node, community) pairs [ntags][2] (svils_get_community_tags)"""
        nt = C.c_uint64()
        _chk(load().svils_get_community_tags(self._h, None, 0, C.byref(nt)))
        t = np.zeros((max(nt.value, 1), 2), dtype=np.uint32)
        _chk(load().svils_get_community_tags(self._h, t.ctypes.data, nt.value, C.byref(nt)))
        return t[:nt.value]

    def aux(self, which):
        shapes = {0: ((self.n, self.k), np.float64), 1: ((self.k, 2), np.float64),
                  2: ((self.n, self.k), np.float64), 3: ((self.n,), np.uint32),
                  4: ((self.n,), np.float64)}
        shape, dt = shapes[which]
        out = np.zeros(shape, dtype=dt)
        _chk(load().svils_get_aux(self._h, which, out.ctypes.data))
        return out

    def debug_eval(self, which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        _chk(load().svils_debug_eval(self._h, which, x.ctypes.data, out.ctypes.data, x.size))
        return out

    def enable_timing(self, mask, period=1):
        _chk(load().svils_enable_timing(self._h, mask))
        _chk(load().svils_set_timing_period(self._h, period))

    def timing(self):
        ms = np.zeros(len(KERNEL_NAMES), dtype=np.float64)
        cnt = np.zeros(len(KERNEL_NAMES), dtype=np.uint64)
        _chk(load().svils_get_timing(self._h, ms.ctypes.data, cnt.ctypes.data))
        return {KERNEL_NAMES[i]: (float(ms[i]), int(cnt[i])) for i in range(len(KERNEL_NAMES))}

    def sweep_stats(self, first=0, count=None):
        """(dense, sparse, shortcut) link counts of sweeps [first, first+count)"""
        if count is None:
            count = self.control().sweeps_done - first
        out = np.zeros((count, 3), dtype=np.uint64)
        if count:
            _chk(load().svils_get_sweep_stats(self._h, first, count, out.ctypes.data))
        return out

    def timed_links(self):
        """(dense, sparse, shortcut) links summed over the sweeps whose phi launch was timed"""
        out = np.zeros(3, dtype=np.uint64)
        _chk(load().svils_get_timed_links(self._h, out.ctypes.data))
        return out

    # ---- pipelined reports ----
    def report_enqueue(self, row_first, row_count, with_communities=True):
        t = C.c_int()
        _chk(load().svils_report_enqueue(self._h, row_first, row_count, int(with_communities), C.byref(t)))
        return t.value

    def report_ready(self, ticket):
        rc = load().svils_report_ready(self._h, ticket)
        if rc < 0:
            _chk(rc)
        return bool(rc)

    def report_test_rows(self, ticket, row_count):
        """the test rows of a report (before report_fetch, which frees the slot)"""
        nr = C.c_uint32()
        rows = np.zeros((max(row_count, 1), 10), dtype=np.float64)
        _chk(load().svils_report_test_rows(self._h, ticket, rows.ctypes.data, C.byref(nr)))
        return rows[:nr.value]

    def report_fetch(self, ticket, row_count, with_communities=True):
        """-> (Control, rows [have][10], member [n][k] or None)"""
        c, nr = Control(), C.c_uint32()
        rows = np.zeros((max(row_count, 1), 10), dtype=np.float64)
        m = np.zeros((self.n, self.k), dtype=np.uint8) if with_communities else None
        _chk(load().svils_report_fetch(self._h, ticket, C.byref(c), rows.ctypes.data, C.byref(nr),
                                       m.ctypes.data if with_communities else None))
        return c, rows[:nr.value], m

    def report_fetch_tags(self, ticket, row_count, cap=None):
        """-> (Control, rows [have][10], tags [ntags][2]); cap: room offered (default: what svils_report_tag_count says)"""
        nt = C.c_uint64()
        if cap is None:
            _chk(load().svils_report_tag_count(self._h, ticket, C.byref(nt)))
            cap = nt.value
        c, nr = Control(), C.c_uint32()
        rows = np.zeros((max(row_count, 1), 10), dtype=np.float64)
        t = np.zeros((max(cap, 1), 2), dtype=np.uint32)
        _chk(load().svils_report_fetch_tags(self._h, ticket, C.byref(c), rows.ctypes.data, C.byref(nr), t.ctypes.data, cap,
                                            C.byref(nt)))
        return c, rows[:nr.value], t[:nt.value]

    # ---- native multi-GPU driver (RCCL inside the library) ----
    def comm_init(self, comm_id, rank, world):
        assert len(comm_id) == COMM_ID_BYTES
        _chk(load().svils_comm_init(self._h, C.c_char_p(comm_id), rank, world))

    def set_node_blocks(self, rank, world, bounds=None):
        """declare the node blocks of all ranks (bounds[world + 1]; None = equal blocks of ceil(n / world) nodes)"""
        if bounds is None:
            _chk(load().svils_set_node_blocks(self._h, rank, world, None))
        else:
            b = np.ascontiguousarray(bounds, dtype=np.uint32)
            assert b.shape == (world + 1,)
            _chk(load().svils_set_node_blocks(self._h, rank, world, b.ctypes.data))

    def comm_query(self):
        """what the bound RCCL says about this handle's communicator (ncclCommCount / UserRank / CuDevice / GetVersion)"""
        ci = CommInfo()
        _chk(load().svils_comm_query(self._h, C.byref(ci)))
        return {"nranks": ci.nranks, "rank": ci.rank, "hip_device": ci.device, "rccl_version": ci.version,
                "row_communicator": bool(ci.row_comm), "pci_bus_id": ci.pci_bus_id.decode(),
                "library": ci.library.decode()}

    def sweep_sharded(self, nsweeps=1):
        _chk(load().svils_sweep_sharded(self._h, nsweeps))

    def set_option(self, key, value):
        """one row of the library's option table for this handle (include/svils.h: svils_set_option)"""
        _chk(load().svils_set_option(self._h, key.encode(), str(int(value)).encode()))

    def get_option(self, key):
        buf = C.create_string_buffer(64)
        _chk(load().svils_get_option(self._h, key.encode(), buf, 64))
        return int(buf.value.decode())

    def gather_communities(self):
        _chk(load().svils_gather_communities(self._h))

    def step_sharded(self, nsteps=1):
        _chk(load().svils_step_sharded(self._h, nsteps))

    def allgather_host(self, send, world):
        """every rank's `send` (equal byte counts), rank by rank: array of shape (world,) + send.shape.  Collective."""
        send = np.ascontiguousarray(send)
        recv = np.empty((world,) + send.shape, dtype=send.dtype)
        _chk(load().svils_comm_allgather_host(self._h, send.ctypes.data, recv.ctypes.data, send.nbytes))
        return recv

    def device_buffer(self, which):
        p, b, r = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _chk(load().svils_device_buffer(self._h, which, C.byref(p), C.byref(b), C.byref(r)))
        return p.value, b.value, r.value

    def stream(self):
        s = C.c_void_p()
        _chk(load().svils_stream(self._h, C.byref(s)))
        return s.value


def option_table():
    """the library's option table as a list of dicts (key, environment, default, settable, meaning)"""
    rows = load().svils_option_table().decode().strip().split("\n")
    head = rows[0].split("\t")
    return [dict(zip(head, r.split("\t"))) for r in rows[1:]]
