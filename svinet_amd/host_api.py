"""ctypes binding of the C++ host side (svinet_amd/host/capi.cc).

Gives Python the PRODUCT's own graph reader, held-out sampler, gamma/lambda
initialisation and training-link list (the reference's Network::read and
LinkSampling constructor, src/network.cc:10-116, src/linksampling.cc:5-155),
so that bench.py and the tests feed the device C ABI with product-made inputs.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsvinet_host.so")
ETA_TYPES = {"uniform": 0, "fromdata": 1, "sparse": 2, "dense": 3}


class Options(C.Structure):
    _fields_ = [("n", C.c_uint32), ("k", C.c_uint32), ("seed", C.c_double),
                ("heldout_ratio", C.c_double), ("link_thresh", C.c_double),
                ("lt_min_deg", C.c_uint32), ("eta_type", C.c_int32), ("accuracy", C.c_int32), ("defer_gamma", C.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: run `python -m svinet_amd.build`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double
    P = C.POINTER
    L.svih_options_default.argtypes = [P(Options), u32, u32]
    L.svih_options_default.restype = None
    L.svih_setup_from_file.argtypes = [C.c_char_p, P(Options)]
    L.svih_setup_from_file.restype = vp
    L.svih_setup_from_pairs.argtypes = [vp, u64, P(Options)]
    L.svih_setup_from_pairs.restype = vp
    L.svih_setup_free.argtypes = [vp]
    L.svih_setup_free.restype = None
    for name, res in (("n", u32), ("k", u32), ("ones", u32), ("singles", u32),
                      ("total_pairs", dbl), ("ones_prob", dbl), ("eta0", dbl), ("eta1", dbl),
                      ("seq2id", P(u32)), ("gamma", P(dbl)), ("lambda", P(dbl)),
                      ("nvalidation", u64), ("validation_sorted", P(u32)),
                      ("validation_accept", P(u32)), ("nlinks", u64), ("links", P(u32)),
                      ("edges", P(u32))):
        f = getattr(L, "svih_" + name)
        f.argtypes = [vp]
        f.restype = res
    L.svih_init_links.argtypes = [vp, vp]
    L.svih_init_links.restype = u64
    L.svih_init_offset.argtypes = [vp]
    L.svih_init_offset.restype = u64
    L.svih_init_streams.argtypes = [vp, u64, u64, vp]
    L.svih_init_streams.restype = C.c_int
    L.svih_deg.argtypes = [vp, u32]
    L.svih_deg.restype = u32
    _lib = L
    return L


def _arr(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


class Setup:
    """Host-side state of `LinkSampling ls(env, network)` before infer()."""

    def __init__(self, path=None, n=0, k=0, pairs=None, seed=0, heldout_ratio=0.01,
                 link_thresh=0.5, lt_min_deg=0, eta_type="uniform", accuracy=False, host_gamma=True):
        """host_gamma=False: init_gamma2 is NOT drawn on the host (2.9 s and a 4.1 GB array at n = 1e6, k = 512); engines get their
        gamma from svils_init_gamma (device_init), bit-identical to the host path"""
        L = load()
        o = Options()
        L.svih_options_default(C.byref(o), n, k)
        o.seed = seed
        o.heldout_ratio = heldout_ratio
        o.link_thresh = link_thresh
        o.lt_min_deg = lt_min_deg
        o.eta_type = ETA_TYPES[eta_type]
        o.accuracy = int(accuracy)
        o.defer_gamma = 0 if host_gamma else 1
        self.host_gamma = bool(host_gamma)
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
            self._h = L.svih_setup_from_pairs(pairs.ctypes.data, pairs.shape[0], C.byref(o))
        else:
            self._h = L.svih_setup_from_file(os.fsencode(path), C.byref(o))
        if not self._h:
            raise IOError("cannot read network %r" % (path,))
        self.link_thresh, self.lt_min_deg = link_thresh, lt_min_deg
        self.n = L.svih_n(self._h)
        self.k = L.svih_k(self._h)
        self.ones = L.svih_ones(self._h)
        self.singles = L.svih_singles(self._h)
        self.total_pairs = L.svih_total_pairs(self._h)
        self.ones_prob = L.svih_ones_prob(self._h)
        self.eta = (L.svih_eta0(self._h), L.svih_eta1(self._h))
        nv = L.svih_nvalidation(self._h)
        self.validation_sorted = _arr(L.svih_validation_sorted(self._h), (nv, 3), np.uint32)
        self.validation_accept = _arr(L.svih_validation_accept(self._h), (nv, 3), np.uint32)
        self.nlinks = L.svih_nlinks(self._h)
        self.links = _arr(L.svih_links(self._h), (self.nlinks, 2), np.uint32)
        self.seq2id = _arr(L.svih_seq2id(self._h), (self.n,), np.uint32)
        self.lam = _arr(L.svih_lambda(self._h), (self.k, 2), np.float64)

    @property
    def gamma(self):
        if not self.host_gamma:
            raise AttributeError("this Setup was made with host_gamma=False: gamma is drawn on the device (device_init)")
        return _arr(load().svih_gamma(self._h), (self.n, self.k), np.float64)

    def init_plan(self):
        """(streams, outputs per stream) as the drop-in binary cuts the init stream (host/linksampling.cc: init_gamma2_on_device)"""
        total = int(self.ones) * int(self.k)
        want = max(1, min(2048, total // (624 * 256)))
        per = ((total + want - 1) // want + 623) // 624 * 624
        return (total + per - 1) // per, per

    def device_init(self, eng, lam=None):
        """init_gamma2 on the device for `eng` (any handle: whole graph, node block, K-shard -- lam = its rows of lambda)"""
        if not hasattr(self, "_init_cache"):
            ns, per = self.init_plan()
            self._init_cache = (self.init_links(), self.init_streams(ns, per), per)
        edges, st, per = self._init_cache
        eng.init_gamma(edges, st, per, self.lam if lam is None else lam)

    @property
    def edges(self):
        return _arr(load().svih_edges(self._h), (self.ones, 2), np.uint32)

    def deg(self, p):
        return load().svih_deg(self._h, p)

    # ---- what svils_init_gamma takes (init_gamma2 on the device, csrc/svils_init.hip) ----
    def init_links(self):
        """every link (held-out ones included) in the order init_gamma2 draws for them: [E][2], p < q"""
        e = load().svih_init_links(self._h, None)
        out = np.empty((e, 2), dtype=np.uint32)
        load().svih_init_links(self._h, out.ctypes.data)
        return out

    def init_offset(self):
        """outputs the MT19937 stream had produced when init_gamma2 began (the held-out sampler's draws)"""
        return int(load().svih_init_offset(self._h))

    def init_streams(self, nstreams, per_stream):
        """[nstreams][624] MT19937 states, per_stream outputs apart, the first at init_offset() (jump-ahead, host/mtjump.hh)"""
        out = np.empty((nstreams, 624), dtype=np.uint32)
        if load().svih_init_streams(self._h, nstreams, per_stream, out.ctypes.data):
            raise RuntimeError("the jump-ahead machinery is unavailable")
        return out

    def engine(self, **kw):
        """An svils Engine loaded with this setup (graph, validation set, state)."""
        from ._svils import Engine
        args = dict(ones=self.ones, ones_prob=self.ones_prob, eta=self.eta,
                    link_thresh=self.link_thresh, lt_min_deg=self.lt_min_deg)
        args.update(kw)
        eng = Engine(self.n, self.k, **args)
        eng.set_graph(self.links)
        eng.set_validation(self.validation_sorted)
        if self.host_gamma:
            eng.set_state(self.gamma, self.lam)
        else:
            self.device_init(eng)
        return eng

    def close(self):
        if getattr(self, "_h", None):
            load().svih_setup_free(self._h)
            self._h = None

    __del__ = close


class BatchEngine:
    """The `-batch` engine of the host library (svinet_amd/host/mmsbbatch.cc): the reference's
    all-pairs CPU engine MMSBInfer::batch_infer (src/mmsbinfer.cc:833-930), plumbing only."""

    def __init__(self, path, n, k, seed=0, heldout_ratio=0.01, eta_type="uniform"):
        L = load()
        vp, u32, u64, dbl, P = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.POINTER
        if not hasattr(L, "_batch_ready"):
            L.svih_batch_from_file.argtypes = [C.c_char_p, P(Options)]
            L.svih_batch_from_file.restype = vp
            for name, res in (("free", None), ("n", u32), ("iter", u32), ("gamma", P(dbl)), ("lambda", P(dbl)),
                              ("nheldout", u64), ("heldout", P(u32)), ("nvalidation", u64),
                              ("validation", P(u32)), ("nrows", u64), ("rows", P(dbl)), ("sweep", None),
                              ("report", C.c_int), ("eta0", dbl), ("eta1", dbl), ("ones_prob", dbl),
                              ("edges", P(u32)), ("ones", u32)):
                f = getattr(L, "svih_batch_" + name)
                f.argtypes = [vp]
                f.restype = res
            L._batch_ready = True
        o = Options()
        L.svih_options_default(C.byref(o), n, k)
        o.seed, o.heldout_ratio, o.eta_type = seed, heldout_ratio, ETA_TYPES[eta_type]
        self._h = L.svih_batch_from_file(os.fsencode(path), C.byref(o))
        if not self._h:
            raise IOError("cannot read network %r" % (path,))
        self.n, self.k = L.svih_batch_n(self._h), k
        self.eta = (L.svih_batch_eta0(self._h), L.svih_batch_eta1(self._h))
        self.ones_prob = L.svih_batch_ones_prob(self._h)
        self.heldout = _arr(L.svih_batch_heldout(self._h), (L.svih_batch_nheldout(self._h), 2), np.uint32)
        self.validation = _arr(L.svih_batch_validation(self._h), (L.svih_batch_nvalidation(self._h), 2), np.uint32)
        self.edges = _arr(L.svih_batch_edges(self._h), (L.svih_batch_ones(self._h), 2), np.uint32)

    @property
    def gamma(self):
        return _arr(load().svih_batch_gamma(self._h), (self.n, self.k), np.float64)

    @property
    def lam(self):
        return _arr(load().svih_batch_lambda(self._h), (self.k, 2), np.float64)

    @property
    def rows(self):
        return _arr(load().svih_batch_rows(self._h), (load().svih_batch_nrows(self._h), 10), np.float64)

    @property
    def iter(self):
        return load().svih_batch_iter(self._h)

    def sweep(self):
        load().svih_batch_sweep(self._h)

    def report(self):
        return bool(load().svih_batch_report(self._h))

    def close(self):
        if getattr(self, "_h", None):
            load().svih_batch_free(self._h)
            self._h = None

    __del__ = close
