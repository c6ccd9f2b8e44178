"""ctypes front-end for oracle/libsvinet_oracle.so.

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from svinet_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsvinet_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "svinet_oracle.c")
    hdr = os.path.join(_HERE, "svinet_oracle.h")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsvinet_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class _Config(C.Structure):
    _fields_ = [("k", C.c_uint32), ("seed", C.c_double), ("heldout_ratio", C.c_double),
                ("link_thresh", C.c_double), ("lt_min_deg", C.c_uint32), ("eta_type", C.c_int),
                ("reportfreq", C.c_uint32), ("max_iterations", C.c_uint32),
                ("use_validation_stop", C.c_int), ("skip_init", C.c_int), ("accuracy", C.c_int),
                ("eta_override0", C.c_double), ("eta_override1", C.c_double), ("train_on_heldout", C.c_int),
                ("sparse_after_iter", C.c_int32),
                ("test_pairs", C.c_void_p), ("ntest", C.c_uint32),
                ("init_comm_ptr", C.c_void_p), ("init_comm_nodes", C.c_void_p), ("ninit_comm", C.c_uint32)]


_lib = None
_lib_omp = None
_SO_OMP = os.path.join(_HERE, "libsvinet_oracle_omp.so")


def lib_omp():
    """libsvinet_oracle_omp.so: the threaded sweep behind bench.py's cpu_baseline_allcores (NOT the reference's
    summation order -- see svinet_oracle_omp.c).  Handles come from lib(); the struct layout is the same translation unit."""
    global _lib_omp
    if _lib_omp is not None:
        return _lib_omp
    srcs = [os.path.join(_HERE, f) for f in ("svinet_oracle_omp.c", "svinet_oracle.c", "svinet_oracle.h")]
    try:
        if not os.path.exists(_SO_OMP) or os.path.getmtime(_SO_OMP) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libsvinet_oracle_omp.so"], stdout=subprocess.DEVNULL)
    except Exception:
        if not os.path.exists(_SO_OMP):
            raise
    L = C.CDLL(_SO_OMP)
    L.orc_ls_sweep_omp.restype = C.c_int
    L.orc_ls_sweep_omp.argtypes = [C.c_void_p, C.c_int]
    L.orc_omp_max_threads.restype = C.c_int
    _lib_omp = L
    return L

ETA_TYPES = {"uniform": 0, "fromdata": 1, "sparse": 2, "dense": 3}


def lib():
    global _lib
    if _lib is not None:
        return _lib
    try:
        build()
    except Exception:
        if not os.path.exists(_SO):
            raise
    L = C.CDLL(_SO)
    vp, u32, dbl = C.c_void_p, C.c_uint32, C.c_double
    P = C.POINTER

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_rng_new", vp, C.c_ulong)
    sig("orc_rng_free", None, vp)
    sig("orc_rng_get", u32, vp)
    sig("orc_rng_uniform", dbl, vp)
    sig("orc_rng_uniform_int", u32, vp, u32)
    sig("orc_digamma", dbl, dbl)
    sig("orc_net_read", vp, C.c_char_p, u32)
    sig("orc_net_from_pairs", vp, vp, C.c_uint64, u32)
    sig("orc_net_free", None, vp)
    sig("orc_net_n", u32, vp)
    sig("orc_net_ones", u32, vp)
    sig("orc_net_deg", u32, vp, u32)
    sig("orc_net_adj", P(u32), vp, u32)
    sig("orc_net_edges", P(u32), vp)
    sig("orc_net_seq2id", P(u32), vp)
    sig("orc_net_y", C.c_int, vp, u32, u32)
    sig("orc_config_default", None, P(_Config), u32)
    sig("orc_ls_create", vp, vp, P(_Config))
    sig("orc_ls_free", None, vp)
    sig("orc_ls_sweep", C.c_int, vp)
    sig("orc_ls_set_skip_validation", None, vp, C.c_int)
    for nm in ("n", "k", "nlinks", "nvalidation", "iter", "nrows", "ntest", "ntest_rows"):
        sig("orc_ls_" + nm, u32, vp)
    sig("orc_ls_links", P(u32), vp)
    sig("orc_ls_training_links", P(dbl), vp)
    for nm in ("gamma", "lambda", "elogpi", "elogbeta", "mphi", "fmap", "rows", "test_rows"):
        sig("orc_ls_" + nm, P(dbl), vp)
    for nm in ("converged", "active_comms", "validation_accept", "validation_sorted", "test_sorted"):
        sig("orc_ls_" + nm, P(u32), vp)
    sig("orc_ls_set_iter", None, vp, u32)
    sig("orc_ls_annealing", C.c_int, vp)
    sig("orc_ls_set_annealing", None, vp, C.c_int)
    sig("orc_ls_write_comm", C.c_int, vp)
    for nm in ("eta0", "eta1", "ones_prob", "total_pairs"):
        sig("orc_ls_" + nm, dbl, vp)
    sig("orc_ls_link_counts", None, vp, P(u32), P(u32), P(u32))
    sig("orc_ls_refresh", None, vp)
    sig("orc_ls_communities", u32, vp, vp)
    sig("orc_ls_write_model", C.c_int, vp, C.c_char_p)
    _lib = L
    return L


def _arr(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    a = np.ctypeslib.as_array(ptr, shape=(n,))
    return a.view(dtype).reshape(shape)


class Rng:
    def __init__(self, seed=0):
        self._h = lib().orc_rng_new(int(seed))

    def get(self):
        return lib().orc_rng_get(self._h)

    def uniform(self):
        return lib().orc_rng_uniform(self._h)

    def uniform_int(self, n):
        return lib().orc_rng_uniform_int(self._h, n)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_rng_free(self._h)
            self._h = None


def digamma(x):
    return lib().orc_digamma(float(x))


class Network:
    """Network::read (src/network.cc:10-116)."""

    def __init__(self, path=None, n=0, pairs=None):
        L = lib()
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int32)
            self._h = L.orc_net_from_pairs(pairs.ctypes.data, pairs.shape[0], n)
        else:
            self._h = L.orc_net_read(os.fsencode(path), n)
        if not self._h:
            raise IOError("cannot read network %r" % (path,))
        self.n = L.orc_net_n(self._h)
        self.ones = L.orc_net_ones(self._h)

    def edges(self):
        return _arr(lib().orc_net_edges(self._h), (self.ones, 2), np.uint32).copy()

    def seq2id(self):
        return _arr(lib().orc_net_seq2id(self._h), (self.n,), np.uint32).copy()

    def deg(self, p):
        return lib().orc_net_deg(self._h, p)

    def adj(self, p):
        # adjacency list length == deg for undirected graphs
        d = self.deg(p)
        return _arr(lib().orc_net_adj(self._h, p), (d,), np.uint32).copy()

    def y(self, a, b):
        return lib().orc_net_y(self._h, a, b)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_net_free(self._h)
            self._h = None


class LinkSampling:
    """LinkSampling ctor + infer() loop body (src/linksampling.cc:5-155,556-790)."""

    def __init__(self, net, k, seed=0, heldout_ratio=0.01, link_thresh=0.5, lt_min_deg=0,
                 eta_type="uniform", reportfreq=1, max_iterations=0, use_validation_stop=True,
                 skip_init=False, accuracy=False, eta_override=None, train_on_heldout=False, sparse_after_iter=1000,
                 test_pairs=None, init_communities=None):
        """test_pairs: [T][2] SEQUENCE ids as -load-test maps them (src/linksampling.cc:1417-1450);
        init_communities: list of lists of sequence ids, one per line of the -init-communities file"""
        L = lib()
        cfg = _Config()
        L.orc_config_default(C.byref(cfg), k)
        cfg.seed = seed
        cfg.heldout_ratio = heldout_ratio
        cfg.link_thresh = link_thresh
        cfg.lt_min_deg = lt_min_deg
        cfg.eta_type = ETA_TYPES[eta_type]
        cfg.reportfreq = reportfreq
        cfg.max_iterations = max_iterations
        cfg.use_validation_stop = int(use_validation_stop)
        cfg.skip_init = int(skip_init)
        cfg.accuracy = int(accuracy)
        if eta_override is not None:
            cfg.eta_override0, cfg.eta_override1 = eta_override
        cfg.train_on_heldout = int(train_on_heldout)
        cfg.sparse_after_iter = int(sparse_after_iter)
        if test_pairs is not None:
            self._tp = np.ascontiguousarray(test_pairs, dtype=np.uint32).reshape(-1, 2)
            cfg.test_pairs, cfg.ntest = self._tp.ctypes.data, self._tp.shape[0]
        if init_communities is not None:
            self._icp = np.concatenate([[0], np.cumsum([len(c) for c in init_communities])]).astype(np.uint32)
            self._icn = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint32) for c in init_communities] or [np.zeros(0, np.uint32)]))
            cfg.init_comm_ptr, cfg.init_comm_nodes, cfg.ninit_comm = self._icp.ctypes.data, self._icn.ctypes.data, len(init_communities)
        self.net = net
        self._h = L.orc_ls_create(net._h, C.byref(cfg))
        self.n = L.orc_ls_n(self._h)
        self.k = L.orc_ls_k(self._h)
        self.nlinks = L.orc_ls_nlinks(self._h)

    def sweep(self):
        return lib().orc_ls_sweep(self._h)

    def sweep_omp(self, nthreads):
        """threaded sweep (all-cores CPU figure only; rounding differs from sweep())"""
        return lib_omp().orc_ls_sweep_omp(self._h, int(nthreads))

    def set_skip_validation(self, skip):
        lib().orc_ls_set_skip_validation(self._h, int(skip))

    # views (valid until the next sweep swaps buffers -> copy)
    @property
    def gamma(self):
        return _arr(lib().orc_ls_gamma(self._h), (self.n, self.k), np.float64).copy()

    def set_gamma(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        assert g.shape == (self.n, self.k)
        C.memmove(lib().orc_ls_gamma(self._h), g.ctypes.data, g.nbytes)

    def set_lambda(self, lam):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        assert lam.shape == (self.k, 2)
        C.memmove(lib().orc_ls_lambda(self._h), lam.ctypes.data, lam.nbytes)

    def set_converged(self, conv):
        conv = np.ascontiguousarray(conv, dtype=np.uint32)
        C.memmove(lib().orc_ls_converged(self._h), conv.ctypes.data, conv.nbytes)

    def refresh(self):
        lib().orc_ls_refresh(self._h)

    @property
    def lam(self):
        return _arr(lib().orc_ls_lambda(self._h), (self.k, 2), np.float64).copy()

    @property
    def elogpi(self):
        return _arr(lib().orc_ls_elogpi(self._h), (self.n, self.k), np.float64).copy()

    @property
    def elogbeta(self):
        return _arr(lib().orc_ls_elogbeta(self._h), (self.k, 2), np.float64).copy()

    @property
    def mphi(self):
        return _arr(lib().orc_ls_mphi(self._h), (self.n, self.k), np.float64).copy()

    @property
    def fmap(self):
        return _arr(lib().orc_ls_fmap(self._h), (self.n, self.k), np.float64).copy()

    @property
    def converged(self):
        return _arr(lib().orc_ls_converged(self._h), (self.n,), np.uint32).copy()

    @property
    def active_comms(self):
        return _arr(lib().orc_ls_active_comms(self._h), (self.n,), np.uint32).copy()

    @property
    def links(self):
        return _arr(lib().orc_ls_links(self._h), (self.nlinks, 2), np.uint32).copy()

    @property
    def training_links(self):
        return _arr(lib().orc_ls_training_links(self._h), (self.n,), np.float64).copy()

    @property
    def validation_accept(self):
        v = lib().orc_ls_nvalidation(self._h)
        return _arr(lib().orc_ls_validation_accept(self._h), (v, 3), np.uint32).copy()

    @property
    def validation_sorted(self):
        v = lib().orc_ls_nvalidation(self._h)
        return _arr(lib().orc_ls_validation_sorted(self._h), (v, 3), np.uint32).copy()

    @property
    def rows(self):
        r = lib().orc_ls_nrows(self._h)
        return _arr(lib().orc_ls_rows(self._h), (r, 10), np.float64).copy()

    @property
    def test_sorted(self):
        t = lib().orc_ls_ntest(self._h)
        return _arr(lib().orc_ls_test_sorted(self._h), (t, 3), np.uint32).copy()

    @property
    def test_rows(self):
        r = lib().orc_ls_ntest_rows(self._h)
        return _arr(lib().orc_ls_test_rows(self._h), (r, 10), np.float64).copy()

    @property
    def iter(self):
        return lib().orc_ls_iter(self._h)

    @iter.setter
    def iter(self, v):
        lib().orc_ls_set_iter(self._h, int(v))

    @property
    def annealing(self):
        return bool(lib().orc_ls_annealing(self._h))

    @annealing.setter
    def annealing(self, v):
        lib().orc_ls_set_annealing(self._h, int(v))

    @property
    def write_comm(self):
        return bool(lib().orc_ls_write_comm(self._h))

    @property
    def eta(self):
        return lib().orc_ls_eta0(self._h), lib().orc_ls_eta1(self._h)

    @property
    def ones_prob(self):
        return lib().orc_ls_ones_prob(self._h)

    @property
    def total_pairs(self):
        return lib().orc_ls_total_pairs(self._h)

    def link_counts(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().orc_ls_link_counts(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def communities(self):
        out = np.zeros((self.n, self.k), dtype=np.uint8)
        lib().orc_ls_communities(self._h, out.ctypes.data)
        return out

    def write_model(self, d):
        os.makedirs(d, exist_ok=True)
        rc = lib().orc_ls_write_model(self._h, os.fsencode(d))
        if rc:
            raise IOError("write_model failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ls_free(self._h)
            self._h = None
