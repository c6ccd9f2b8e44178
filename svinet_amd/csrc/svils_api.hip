// svils_api.hip -- the C ABI of include/svils.h on top of the gfx950 kernels: the life cycle of a handle (create / destroy),
// graph, validation set and state upload, the getters and the hipEvent timing.  The other entry points live next door
// (svils_handle.h lists the translation units).  There is no CPU compute path: without a HIP device svils_create() fails.
#include "svils_handle.h"

namespace svils_impl {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int drain_timing(svils_handle *h) {
  for (int i = 0; i < SVILS_KERNEL_COUNT; ++i) {
    for (auto &ev : h->pending[i]) {
      HIPCHK(hipEventSynchronize(ev.b));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
      h->t_ms[i] += ms;
      h->t_n[i]++;
      h->freelist.push_back(ev);
    }
    h->pending[i].clear();
  }
  return 0;
}

int fault_error(uint32_t code) {
  if (code == 2u)
    return fail(SVILS_ERR_DEVICE, "K-sharded sweep: the softmax denominator of a link underflowed (rows of disjoint support); "
                                  "switch the log-domain exchange on for this model: svils_ksh_log_domain(h, 1) (the default above K = 700)");
  return fail(SVILS_ERR_DEVICE, "an in-launch hand-off between workgroups timed out (role blocks not co-resident on this device); the state is frozen");
}

int elogpi_rows(svils_handle *h, double **rows) {
  *rows = h->d.elogpi;
  if (h->d.ksh && !h->d.ksh_log) {   // K-sharded, product form: no stored Elogpi either -- from gamma and the summed row sums
    if (!h->elogpi_view) {
      int rc = dalloc(h, &h->elogpi_view, (size_t)h->geo.n_alloc * h->geo.ld);
      if (rc) return rc;
    }
    DeviceState dv = h->d;
    dv.elogpi = h->elogpi_view;
    dv.epi = nullptr;
    launch_ksh_phase(h->geo, dv, h->prm, 6, h->stream);
    HIPCHK(hipGetLastError());
    *rows = h->elogpi_view;
    return 0;
  }
  if (!h->d.skip_elogpi) return 0;
  if (!h->elogpi_view) {
    int rc = dalloc(h, &h->elogpi_view, (size_t)h->geo.n_alloc * h->geo.ld);
    if (rc) return rc;
  }
  DeviceState dv = h->d;
  dv.elogpi = h->elogpi_view;
  dv.epi = nullptr;                       // (k_dir_exp leaves the exp(Elogpi) rows of the state alone)
  launch_dir_exp(h->geo, dv, h->stream);
  HIPCHK(hipGetLastError());
  *rows = h->elogpi_view;
  return 0;
}

void drop_graphs_of(svils_handle *h) {
  for (auto &g_ : h->sgexec) if (g_) { (void)hipGraphExecDestroy(g_); g_ = nullptr; }
  if (h->gexec1) { (void)hipGraphExecDestroy(h->gexec1); h->gexec1 = nullptr; }
  if (h->gexecN) { (void)hipGraphExecDestroy(h->gexecN); h->gexecN = nullptr; }
  for (auto &g_ : h->gexecP) if (g_) { (void)hipGraphExecDestroy(g_); g_ = nullptr; }
}

// chunk a row segment [off, off+len) of node p into items of <= ch neighbours
void chunk_row(std::vector<Item> &items, uint32_t p, uint32_t off, uint32_t len, uint32_t ch,
               int32_t *next_slot, int32_t *first_slot, uint32_t *nsplit) {
  if (len == 0) { if (first_slot) { *first_slot = -1; *nsplit = 0; } return; }
  uint32_t nch = (len + ch - 1) / ch;
  if (nch <= 1 || !next_slot) {
    if (nch <= 1) {
      items.push_back(Item{p, off, len, -1});
      if (first_slot) { *first_slot = -1; *nsplit = 0; }
      return;
    }
  }
  uint32_t base = len / nch, rem = len % nch, o = off;
  if (first_slot) { *first_slot = *next_slot; *nsplit = nch; }
  for (uint32_t c = 0; c < nch; ++c) {
    uint32_t l = base + (c < rem ? 1u : 0u);
    int32_t slot = -1;
    if (next_slot) slot = (*next_slot)++;
    items.push_back(Item{p, o, l, slot});
    o += l;
  }
}

uint64_t tags_of_bits(const Geometry &g, const uint64_t *bits, uint32_t *tags, uint64_t cap) {
  uint64_t cnt = 0;
  for (uint32_t p = 0; p < g.n; ++p)
    for (int v = 0; v < g.V; ++v) {
      uint64_t b = bits[(size_t)p * g.kw + v];
      while (b) {
        const int lw = __builtin_ctzll(b);
        b &= b - 1;
        const uint32_t k = kmap_host(g.W, g.V, lw, v);
        if (k >= g.K) continue;
        if (tags && cnt < cap) { tags[2 * cnt] = p; tags[2 * cnt + 1] = k; }
        ++cnt;
      }
    }
  return cnt;
}

}  // namespace svils_impl

extern "C" {

const char *svils_last_error(void) { return g_err.c_str(); }
int svils_abi_version(void) { return SVILS_ABI_VERSION; }

const char *svils_kernel_name(int k) {
  static const char *names[SVILS_KERNEL_COUNT] = {"phi", "reduce_sum", "finalize", "s3",
                                                  "validation", "reduce_s", "tail", "classify", "exchange"};
  return (k >= 0 && k < SVILS_KERNEL_COUNT) ? names[k] : "?";
}

int svils_config_default(svils_config *cfg, uint32_t n, uint32_t k) {
  if (!cfg || k == 0) return fail(SVILS_ERR_ARG, "svils_config_default: bad arguments");
  memset(cfg, 0, sizeof(*cfg));
  cfg->n = n;
  cfg->k = k;
  cfg->alpha = (double)1 / k;   // src/env.hh:344
  cfg->eta0 = 1.0;              // eta_type "uniform", src/network.cc:236-238
  cfg->eta1 = 1.0;
  cfg->epsilon = 1e-30;         // src/env.hh:395
  cfg->link_thresh = 0.5;       // src/main.cc: link_thresh
  cfg->lt_min_deg = 0;
  cfg->reportfreq = 1;          // src/main.cc:149-153
  cfg->use_validation_stop = 1;
  cfg->ones_prob = 0.0;
  cfg->zeros_prob = 1.0;
  cfg->device = 0;
  cfg->node_begin = 0;
  cfg->node_end = n;
  cfg->n_alloc = 0;
  cfg->sparse_after_iter = 1000;   // src/linksampling.cc:634
  return 0;
}

int svils_create(const svils_config *cfg, svils_handle **out) {
  if (!cfg || !out) return fail(SVILS_ERR_ARG, "svils_create: null argument");
  *out = nullptr;
  if (cfg->n == 0 || cfg->k == 0) return fail(SVILS_ERR_ARG, "svils_create: n and k must be > 0");
  if (cfg->k > SVILS_MAX_K_TOTAL || cfg->k_total > SVILS_MAX_K_TOTAL)
    return fail(SVILS_ERR_UNSUPPORTED, "k=%u exceeds SVILS_MAX_K_TOTAL=%d (the reference's community ids are 16-bit, src/linksampling.cc:635)",
                std::max(cfg->k, cfg->k_total), SVILS_MAX_K_TOTAL);
  if (cfg->k > SVILS_MAX_K && cfg->k_total)
    return fail(SVILS_ERR_UNSUPPORTED, "K-sharded handle: a slice of k=%u columns exceeds SVILS_MAX_K=%d (use more slices)", cfg->k, SVILS_MAX_K);
  if (cfg->reportfreq == 0) return fail(SVILS_ERR_ARG, "reportfreq must be >= 1");
  uint32_t nb = cfg->node_begin, ne = cfg->node_end ? cfg->node_end : cfg->n;
  if (nb > ne || ne > cfg->n) return fail(SVILS_ERR_ARG, "bad node block [%u,%u) for n=%u", nb, ne, cfg->n);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(SVILS_ERR_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(SVILS_ERR_ARG, "device %d out of range (%d devices)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  if (cfg->k > SVILS_MAX_K) return tiles_create(cfg, out);   // column tiles on this device (svils_handle::tiles)

  svils_handle *h = new (std::nothrow) svils_handle();
  if (!h) return fail(SVILS_ERR_NOMEM, "out of host memory");
  h->cfg = *cfg;
  h->opt = options_from_env();   // the only place a handle looks at the environment (svils_options.h)
  Geometry &g = h->geo;
  g.n = cfg->n;
  g.n_alloc = std::max(cfg->n_alloc, cfg->n);
  g.K = cfg->k;
  g.ld = (cfg->k + 15u) & ~15u;  // rows are 128-byte aligned
  // Lane-per-link layout (K <= 56): rows packed at ld = K rounded to even (16-byte aligned, which is all the chunked
  // double2 loads need): 160-byte rows at K = 20 instead of a 256-byte stride.  The finalise pass leaves fewer bytes
  // dirty (the boundary behind it drains them), a phi row's two lines carry no padding and the n-by-k state fits one
  // XCD's L2 (2.9 instead of 4.6 MB per array on ca-AstroPh).  profiles/r03r_packed_rows.txt: ca-AstroPh K=20
  // 54.7 -> 53.7 us per sweep (phi 24.3 -> 22.6), LFR K=28 35.4 -> 34.5, ca-AstroPh K=28 76.7 -> 75.4.  A kernel
  // instantiated for more columns than K reads a few doubles of the next row (masked: Elogbeta = -inf there); every
  // allocation carries slack for the last row.  Option pack_rows = 0 restores the padded stride (A/B).
  if (h->opt.pack_rows && use_lpl(cfg->k) && !cfg->k_total) g.ld = (cfg->k + 1u) & ~1u;
  g.k10 = cfg->k / 10;           // integer division, src/linksampling.cc:465,634
  g.node_begin = nb;
  g.node_end = ne;
  g.K0 = 0;
  g.Kt = cfg->k;
  if (cfg->k_total) {   // K-sharded handle: a column slice of every row
    if ((uint64_t)cfg->k_begin + cfg->k > cfg->k_total || cfg->k_total > SVILS_MAX_K_TOTAL || nb != 0 || ne != cfg->n) {
      delete h;
      return fail(SVILS_ERR_ARG, "K-sharded handle: need k_begin + k <= k_total <= %d and the node block [0, n)", SVILS_MAX_K_TOTAL);
    }
    g.K0 = cfg->k_begin;
    g.Kt = cfg->k_total;
    g.k10 = cfg->k_total / 10;
  }
  if (!pick_layout(cfg->k, &g.W, &g.V)) { delete h; return fail(SVILS_ERR_UNSUPPORTED, "unsupported k"); }
  if (cfg->k_total) g.W = 64;    // the K-sharded kernels are row-per-wavefront whatever the slice width
  g.kw = (uint32_t)g.V;
  Params &p = h->prm;
  p.ones = cfg->ones; p.alpha = cfg->alpha; p.eta0 = cfg->eta0; p.eta1 = cfg->eta1;
  p.epsilon = cfg->epsilon; p.link_thresh = cfg->link_thresh; p.lt_min_deg = cfg->lt_min_deg;
  p.reportfreq = cfg->reportfreq; p.use_validation_stop = cfg->use_validation_stop;
  p.ones_prob = cfg->ones_prob; p.zeros_prob = cfg->zeros_prob;
  p.sparse_after = cfg->sparse_after_iter;
  memset(&h->d, 0, sizeof(h->d));

  int rc = 0;
  auto guard = [&](int r) { if (r && !rc) rc = r; };
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail(SVILS_ERR_DEVICE, "hipStreamCreate failed");
  }
  DeviceState &d = h->d;
  const size_t nk = (size_t)g.n_alloc * g.ld;
  guard(dalloc(h, &d.gamma, nk));
  d.gacc = d.gamma;   // full sweeps accumulate gammanext in place
  guard(dalloc(h, &d.elogpi, nk));
  // exp(Elogpi) for the product form of k_phi (K > 56): the phi pass trades its exps for multiplies (ca-AstroPh K=200:
  // 174 -> 120 us, n=2e5 K=512: 3.45 -> 3.02 ms).  The price is one more n-by-k write in the finalise pass.  Whole-graph
  // handles can afford it at any size since round 3: their sweeps no longer store the mean indicators (derive_m), so
  // the finalise pass writes three n-by-k arrays as before (n=1e6 K=512, profiles/r03e_derive_m.txt: phi -1.3 ms,
  // finalise +0.06 ms, s3 +0.2 ms per sweep).  Node-block handles (multi-GPU: mphi stays stored) keep the 1.5 GB limit:
  // beyond it their phi pass runs at the HBM gather ceiling either way and the extra write costs more than the exps.
  d.ksh = cfg->k_total ? 1 : 0;
  d.ksh_log = cfg->k_total > 700 ? 1 : 0;   // psi(1/K) < -745: concentrated memberships underflow the product form
  // link_thresh < 1/2: a phi above the threshold need not be the maximum, so the tag goes to the first strict maximum
  // over ALL columns (src/matrix.hh:521-532).  The log-domain exchange already carries the link's maximum; the lowest
  // column attaining it travels next to the denominators (SVILS_KSH_EARG, MIN) -- no further pass over the rows.
  d.ksh_lowt = (d.ksh && cfg->link_thresh < 0.5) ? 1 : 0;
  if (d.ksh_lowt) d.ksh_log = 1;
  h->derive_ok = h->opt.derive_m != 0;
#ifdef SVILS_TESTING
  h->d.inject_fault = h->opt.fault_inject;   // (libsvils_testing.so only)
#endif
  const bool whole_graph = g.node_begin == 0 && g.node_end == g.n;
  uint64_t epi_max_mb = (whole_graph && h->derive_ok) ? ~0ull >> 21 : 1536;
  if (h->opt.epi_max_mb >= 0) epi_max_mb = (uint64_t)h->opt.epi_max_mb;   // A/B knob (profiles/r03*)
  if (d.ksh || (!use_lpl(g.K) && nk * sizeof(double) <= epi_max_mb << 20)) guard(dalloc(h, &d.epi, nk));
  if (d.ksh) {
    guard(dalloc(h, &d.rowx, 3 * (size_t)g.n));
    guard(dalloc(h, &d.q2v, g.Kt));
  }
  {   // svils_internal.h: DeviceState::skip_elogpi
    const bool can = !d.ksh && d.epi && !use_lpl(g.K) && g.K <= 512u && cfg->link_thresh >= 0.5;
    const bool big = nk * sizeof(double) >= ((size_t)256 << 20);
    d.skip_elogpi = (can && (h->opt.skip_elogpi < 0 ? big : h->opt.skip_elogpi != 0)) ? 1 : 0;
  }
  if (d.skip_elogpi) d.gacc = d.elogpi;   // (the array is free: nothing stores or reads Elogpi rows on such a handle)
  guard(dalloc(h, &d.mphi, nk));
  guard(dalloc(h, &d.conv, 2 * (size_t)g.n_alloc));
  guard(dalloc(h, &d.active_cnt, g.n_alloc));
  guard(dalloc(h, &d.cflag, g.n_alloc));
  guard(dalloc(h, &d.cls_epoch, 4));
  guard(dalloc(h, &d.amask, (size_t)g.n_alloc * g.kw));
  guard(dalloc(h, &d.member, (size_t)g.n_alloc * g.kw));
  d.xf_ld = 2u + 2u * g.kw;
  guard(dalloc(h, &d.xflags, (size_t)g.n_alloc * d.xf_ld));
  guard(dalloc(h, &d.lambda, 2 * (size_t)g.K));
  guard(dalloc(h, &d.elogbeta, 2 * (size_t)g.K));
  guard(dalloc(h, &d.kvec_a, g.K));
  guard(dalloc(h, &d.iscale, g.K));
  guard(dalloc(h, &d.kvec_c, 3 * (size_t)g.K + 4));
  d.rows_cap = 1u << 16;
  guard(dalloc(h, &d.rows, (size_t)d.rows_cap * 10));
  guard(dalloc(h, &d.ctrl, 1));
  {
    double *lt = nullptr;
    guard(dalloc(h, &lt, 256, false));
    if (!rc) {
      // {1/c_i, ln c_i} at the centres of 128 equal sub-intervals of [1,2) (log_tab in svils_devutil.h)
      double tab[256];
      for (int i = 0; i < 128; ++i) {
        const double c = 1.0 + (i + 0.5) / 128.0;
        tab[2 * i] = 1.0 / c;
        tab[2 * i + 1] = std::log(c);
      }
      if (hipMemcpy(lt, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SVILS_ERR_DEVICE, "log table upload failed");
      d.logtab = lt;
    }
  }
  guard(dalloc(h, &h->row_scratch, 10));
  d.sweep_stats_cap = 1u << 12;
  guard(dalloc(h, &d.sweep_stats, (size_t)d.sweep_stats_cap * 4));
  guard(dalloc(h, &d.stamps, 4 * 1024 * 8));
  guard(dalloc(h, &d.tail_ctl, 4));
  guard(dalloc(h, &d.tail_part, SVILS_TAIL_BLOCKS * 4));
  d.nb_t = 1;
  if (rc) { svils_destroy(h); return rc; }
  DevCtrl c;
  memset(&c, 0, sizeof c);
  c.annealing = 1;             // _annealing_phase(true), src/linksampling.cc:33
  c.prev_h = -2147483647;      // :21
  c.max_h = -2147483647;       // :19
  c.iter = 0;                  // quirk Q1
  if (hipMemcpyAsync(d.ctrl, &c, sizeof c, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess) {
    svils_destroy(h);
    return fail(SVILS_ERR_DEVICE, "control block upload failed");
  }
  h->graph_after = h->opt.graph_after;
  h->shard_fold_ok = h->opt.shard_fold != 0;
  h->xchunks = h->opt.xchunks;
  *out = h;
  return 0;
}

int svils_destroy(svils_handle *h) {
  if (!h) return 0;
  (void)hipSetDevice(h->cfg.device);
  if (!h->tiles.empty()) {   // column tiles: tile 0 owns the stream, so it goes last
    for (size_t i = h->tiles.size(); i-- > 0;) svils_destroy(h->tiles[i]);
    delete h;
    return 0;
  }
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (int i = 0; i < SVILS_KERNEL_COUNT; ++i)
    for (auto &ev : h->pending[i]) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  for (auto &ev : h->freelist) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  if (h->gexec1) (void)hipGraphExecDestroy(h->gexec1);
  if (h->gexecN) (void)hipGraphExecDestroy(h->gexecN);
  for (auto &g_ : h->gexecP) if (g_) { (void)hipGraphExecDestroy(g_); g_ = nullptr; }
  for (auto &g_ : h->sgexec) if (g_) { (void)hipGraphExecDestroy(g_); g_ = nullptr; }
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);   // nothing of a communicator may still be enqueued
  comm_destroy(h);
  if (h->stage) (void)hipFree(h->stage);
  if (h->stage_flag) (void)hipFree(h->stage_flag);
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  for (hipEvent_t e : h->ev_chunk) (void)hipEventDestroy(e);
  if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
  for (auto &rs : h->rslot) {
    if (rs.dev) (void)hipFree(rs.dev);
    if (rs.host) (void)hipHostFree(rs.host);
    if (rs.packed) (void)hipEventDestroy(rs.packed);
    if (rs.landed) (void)hipEventDestroy(rs.landed);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  for (void *p : h->allocs) (void)hipFree(p);
  if (h->stream && !h->stream_shared) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int svils_set_graph(svils_handle *h, const uint32_t *links, uint64_t nlinks) {
  if (TILED(h)) {
    for (svils_handle *t : h->tiles) { int rc_ = svils_set_graph(t, links, nlinks); if (rc_) return rc_; }
    h->have_graph = true;
    return tiles_try_init(h);
  }
  if (!h || (!links && nlinks)) return fail(SVILS_ERR_ARG, "svils_set_graph: null argument");
  if (h->have_graph) return fail(SVILS_ERR_ARG, "svils_set_graph: graph already set");
  HIPCHK(hipSetDevice(h->cfg.device));
  Geometry &g = h->geo;
  const uint32_t n = g.n;
  // symmetric CSR; row x = {p < x, ascending} ++ {q > x in link-list order}: the order in
  // which the reference's link loop touches gammanext[x] (src/linksampling.cc:605-701)
  std::vector<uint64_t> rowptr(n + 1, 0);
  for (uint64_t l = 0; l < nlinks; ++l) {
    uint32_t p = links[2 * l], q = links[2 * l + 1];
    if (p >= q || q >= n) return fail(SVILS_ERR_ARG, "link %llu = (%u,%u): need p < q < n", (unsigned long long)l, p, q);
    if (l && links[2 * l - 2] > p) return fail(SVILS_ERR_ARG, "links must be sorted by first endpoint (link %llu)", (unsigned long long)l);
    rowptr[p + 1]++;
    rowptr[q + 1]++;
  }
  for (uint32_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
  std::vector<uint32_t> col(std::max<uint64_t>(2 * nlinks, 1));
  std::vector<uint32_t> upper(n, 0);
  std::vector<uint32_t> elink;   // K-sharded handles: training-link index of every CSR entry
  {
    std::vector<uint64_t> fill(rowptr.begin(), rowptr.end() - 1);
    // lower parts: links arrive sorted by p, so appending p to row q keeps ascending order
    if (h->d.ksh) elink.assign(std::max<uint64_t>(2 * nlinks, 1), 0);
    for (uint64_t l = 0; l < nlinks; ++l) {
      if (h->d.ksh) elink[fill[links[2 * l + 1]]] = (uint32_t)l;
      col[fill[links[2 * l + 1]]++] = links[2 * l];
    }
    for (uint32_t i = 0; i < n; ++i) upper[i] = (uint32_t)(fill[i] - rowptr[i]);
    for (uint64_t l = 0; l < nlinks; ++l) {
      if (h->d.ksh) elink[fill[links[2 * l]]] = (uint32_t)l;
      col[fill[links[2 * l]]++] = links[2 * l + 1];
    }
  }
  // work items over the owned node block
  const int G = 64 / g.W;
  const uint32_t ch = 32u * (uint32_t)G;
  std::vector<Item> items_phi, items_s3;
  std::vector<int32_t> split_first(n, -1);
  std::vector<uint32_t> split_cnt(n, 0);
  int32_t next_slot = 0;
  h->h_item_phi.assign((size_t)n + 1, 0);
  h->h_item_s3.assign((size_t)n + 1, 0);
  h->h_linkptr.assign((size_t)n + 1, 0);
  for (uint32_t p = 0; p < n; ++p) {
    h->h_item_phi[p] = (uint32_t)items_phi.size();
    h->h_item_s3[p] = (uint32_t)items_s3.size();
    const uint32_t deg = (uint32_t)(rowptr[p + 1] - rowptr[p]);
    h->h_linkptr[p + 1] = h->h_linkptr[p] + (deg - upper[p]);
    if (p < g.node_begin || p >= g.node_end) continue;
    chunk_row(items_phi, p, 0, deg, ch, &next_slot, &split_first[p], &split_cnt[p]);
    chunk_row(items_s3, p, upper[p], deg - upper[p], ch, nullptr, nullptr, nullptr);
  }
  h->h_item_phi[n] = (uint32_t)items_phi.size();
  h->h_item_s3[n] = (uint32_t)items_s3.size();
  DeviceState &d = h->d;
  d.nitems_phi = (uint32_t)items_phi.size();
  d.nitems_s3 = (uint32_t)items_s3.size();
  d.nslots = (uint32_t)next_slot;
  auto cap = [](uint64_t x, uint32_t lim) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(x, lim)); };
  // grids of the row-per-wavefront kernels: whole multiples of the resident block count
  {
    // every phi block leaves a partial row of `sum` for k_colreduce / k_tail: one resident round of blocks on
    // graphs of a few rounds of work (ca-AstroPh K=200: sweep 254 -> 241 us, K=100: 198 -> 185), four rounds
    // (better balance of the last round: phi 3085 -> 2951 us at n=2e5, K=512) on large ones
    const uint32_t res = rpw_resident_blocks(g, 0, h->cfg.device);
    const uint64_t want = ((uint64_t)d.nitems_phi + 3) / 4;
    d.nb_a = cap(want, (want < 16ull * res ? 1u : 4u) * res);
  }
  d.nb_b = cap(((uint64_t)(g.node_end - g.node_begin) + 4 * G - 1) / (4 * G), rpw_resident_blocks(g, 2, h->cfg.device));
  if (d.ksh) {
    // K-sharded handles: their finalise pass (k_fin1_ksh) is a small-register kernel of its own, sized here rather than by
    // k_finalize's occupancy: eight blocks per CU.  (It stays the weakest kernel of a rank-sweep -- 650 us at n = 1e6 on a
    // 64-column slice, 60 % of its wave cycles parked on memory, profiles/r07p_kshard_sq.txt; the grid is not why.)
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device);
    const uint32_t npb = g.V == 1 ? 16u : 4u;   // nodes per block: four per wavefront on slices of <= 64 columns
    d.nb_b = cap(((uint64_t)(g.node_end - g.node_begin) + npb - 1) / npb, 8u * (uint32_t)(cus > 0 ? cus : 256));
  }
  d.nb_c = cap((d.nitems_s3 + 3) / 4, 2 * rpw_resident_blocks(g, 1, h->cfg.device));
  // lane-per-link layout for small K: wave-items of 64 consecutive entries of a class list
  // The class lists pack an entry index into 27 bits: graphs of 2^26 training links or more take the
  // row-per-wavefront kernels at small K too (they index with 64 bits and have no such limit; slower per link at
  // K <= 56, but the reference's main use case -- small K on a large graph -- must not be refused).
  // Option lpl_max_entries lowers the switch-over point (tests exercise the fallback on small graphs with it).
  const uint64_t lpl_max_entries = std::min<uint64_t>(1ull << 27, h->opt.lpl_max_entries);
  d.lpl = (use_lpl(g.K) && !d.ksh && 2 * nlinks < lpl_max_entries) ? 1 : 0;
  {
    const size_t state_bytes = (size_t)g.n_alloc * g.ld * sizeof(double);
    d.wt = (d.lpl && state_bytes >= ((size_t)1 << 20) && state_bytes <= ((size_t)8 << 20)) ? 1 : 0;    // svils_internal.h: DeviceState::wt
    if (h->opt.wt >= 0) d.wt = (d.lpl && h->opt.wt != 0) ? 1 : 0;                                     // A/B knob
  }
  d.nlinks = nlinks;
  d.ent_begin = rowptr[g.node_begin];
  d.ent_end = rowptr[g.node_end];
  d.lpl_w0 = 0;
  d.lpl_nitems = (uint32_t)(((d.ent_end - d.ent_begin) + 63) >> 6) + 1u;
  {
    // owned links = those whose first endpoint is in the node block (list is sorted by p)
    uint64_t lb = 0, le = nlinks;
    while (lb < nlinks && links[2 * lb] < g.node_begin) ++lb;
    le = lb;
    while (le < nlinks && links[2 * le] < g.node_end) ++le;
    d.link_begin = lb;
    d.link_end = le;
  }
  std::vector<uint32_t> erow;
  if (d.lpl) {
    // classification tiles: 1024 entries, more on large graphs so that there are at most ~2048 tiles
    // (the scatter pass adds up the counts of all tiles below its own); erow / col padded to whole tiles
    d.cls_tile = 1024u * (uint32_t)std::max<uint64_t>(1, (2 * nlinks + 1024ull * 2048 - 1) / (1024ull * 2048));
    d.ent_pad = ((2 * nlinks + d.cls_tile - 1) / d.cls_tile) * d.cls_tile;
    if (d.ent_pad == 0) d.ent_pad = d.cls_tile;
    erow.assign(d.ent_pad, 0xffffffffu);
    col.resize(d.ent_pad, 0xffffffffu);
    for (uint32_t p = 0; p < n; ++p)
      for (uint64_t e = rowptr[p]; e < rowptr[p + 1]; ++e) erow[e] = p;
    d.cls_tile0 = (uint32_t)(d.ent_begin / d.cls_tile);
    d.cls_ntiles = d.ent_end > d.ent_begin
                       ? (uint32_t)((d.ent_end + d.cls_tile - 1) / d.cls_tile) - d.cls_tile0 : 0u;
    // one block per CU for the three passes: at most SVILS_FOLD_ROWS partial rows per K-vector
    const int nw = lpl_phi_waves(g.K);
    const uint32_t phi_items = d.lpl_nitems;
    d.nb_a = cap((phi_items + nw - 1) / nw, std::min<uint32_t>(SVILS_FOLD_ROWS, lpl_phi_resident_blocks(g.K, h->cfg.device)));
    // finalise pass: 12-wave blocks of 64 / lpl_finalize_group(K) nodes per wavefront, as many as the device holds at
    // once (one per CU at its register budget); larger graphs loop inside the blocks
    {
      int cus = 0;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device);
      d.fin_waves = lpl_finalize_waves(g.K, g.node_end - g.node_begin, cus > 0 ? (uint32_t)cus : 256u);
    }
    const uint32_t fnodes = d.fin_waves * (64u / (uint32_t)lpl_finalize_group(g.K));
    d.nb_b = cap(((uint64_t)(g.node_end - g.node_begin) + fnodes - 1) / fnodes,
                 std::min<uint32_t>(SVILS_FOLD_ROWS, lpl_finalize_resident_blocks(g.K, h->cfg.device)));
    d.s3_threads = lpl_s3_threads(g.K, d.link_end - d.link_begin);
    d.nb_c = cap((d.link_end - d.link_begin + d.s3_threads - 1) / d.s3_threads, 192);   // + up to 64 classification blocks
  }

  {
    // fixed-point scale of sum[k] <= 2 L (lane-per-link layout, svils_devutil.h: fx_add): 2^shift * (2 L + 2) < 2^61
    int bits = 1;
    while ((1ull << bits) < 2 * nlinks + 2) ++bits;
    d.fx_scale = std::ldexp(1.0, 61 - bits);
    d.fx_inv = std::ldexp(1.0, bits - 61);
  }
  int rc = 0;
  auto guard = [&](int r) { if (r && !rc) rc = r; };
  guard(dalloc(h, &d.rowptr, (size_t)n + 1, false));
  guard(dalloc(h, &d.col, col.size(), false));
  guard(dalloc(h, &d.upper, n, false));
  guard(dalloc(h, &d.items_phi, items_phi.size(), false));
  guard(dalloc(h, &d.items_s3, items_s3.size(), false));
  guard(dalloc(h, &d.split_first, n, false));
  guard(dalloc(h, &d.split_cnt, n, false));
  guard(dalloc(h, &d.parts, (size_t)d.nslots * g.ld));
  guard(dalloc(h, &d.part_cnt, (size_t)d.nslots * g.ld));
  if (d.lpl) {
    guard(dalloc(h, &d.erow, erow.size(), false));
    guard(dalloc(h, &d.links, std::max<uint64_t>(2 * nlinks, 1), false));
    guard(dalloc(h, &d.slot_f, 2 * (size_t)d.lpl_nitems * g.ld));
    guard(dalloc(h, &d.ghead, 2 * (size_t)g.n_alloc * g.ld));
    guard(dalloc(h, &d.gtail, 2 * (size_t)g.n_alloc * g.ld));
    guard(dalloc(h, &d.gacc1, (size_t)g.n_alloc * g.ld));
    guard(dalloc(h, &d.member_acc, g.n_alloc));
    if (h->prm.lt_min_deg > 0) guard(dalloc(h, &d.fcnt, (size_t)g.n_alloc * g.ld));
    const size_t own = std::max<uint64_t>(d.ent_end - d.ent_begin, 1);
    for (int l = 0; l < 2; ++l) {
      guard(dalloc(h, &d.cp[l], own, false));
      guard(dalloc(h, &d.cq[l], own, false));
    }
    guard(dalloc(h, &d.scol, own, false));
    for (int l = 0; l < 3; ++l) guard(dalloc(h, &d.npos[l], (size_t)g.n_alloc + 1));
    const uint32_t tiles_all = (uint32_t)(d.ent_pad / d.cls_tile);
    guard(dalloc(h, &d.tcnt, tiles_all));
    guard(dalloc(h, &d.cls_args, 8));
    guard(dalloc(h, &d.s3_ctl, 4));
    guard(dalloc(h, &d.cls_sync, 4));
    guard(dalloc(h, &d.tbase, tiles_all));
    guard(dalloc(h, &d.tpoll, tiles_all));
    if (g.K <= 32) guard(dalloc(h, &d.gacc0, (size_t)g.n_alloc * g.ld));   // three-launch sweeps accumulate beside gamma
    {
      // co-residency of the in-launch hand-off: the s3 launch's 64 + 1 role blocks next to whatever s3 blocks are still
      // running.  Not met on a partition of a few CUs; not knowable under a CU mask (the attribute still counts every
      // CU): both keep the four-launch sweep, whose passes never wait for another workgroup.  Option fused3 = 0 / 1 overrides.
      int assume_cus = 0;
#ifdef SVILS_TESTING
      assume_cus = h->opt.assume_cus;   // (libsvils_testing.so only)
#endif
      const uint32_t res = lpl_s3_resident_blocks(g.K, h->cfg.device, d.s3_threads, assume_cus);
      const bool masked = getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK");   // (the runtime's own variables, looked at once per graph)
      h->fused3_ok = res >= 2u * 65u && !masked;
      if (h->opt.fused3 >= 0) h->fused3_ok = h->opt.fused3 != 0;
    }
    // ltot [2][8] u32 | shist [2][K] u64, cleared together before a stand-alone classification
    const size_t shist_bytes = ((2 * (size_t)g.K * sizeof(unsigned long long)) + 63) / 64 * 64;
    h->cls_zero_bytes = 64 + shist_bytes + 2 * 8 * 2 * 64 * sizeof(long long);   // ... | sumfx [2][8][hi|lo][64] i64
    unsigned char *cz = nullptr;
    guard(dalloc(h, &cz, h->cls_zero_bytes));
    h->cls_zero = cz;
    if (cz) {
      d.ltot = reinterpret_cast<uint32_t *>(cz);
      d.shist = reinterpret_cast<unsigned long long *>(cz + 64);
      d.sumfx = reinterpret_cast<long long *>(cz + 64 + shist_bytes);

    }
  }
  guard(dalloc(h, &d.part_a, (size_t)d.nb_a * g.K));
  guard(dalloc(h, &d.part_links, (size_t)d.nb_a * 3));
  guard(dalloc(h, &d.part_b, (size_t)d.nb_b * 2 * g.K));
  guard(dalloc(h, &d.part_c, (size_t)d.nb_c * g.K));
  if (d.ksh) {
    if (nlinks >= (1ull << 32)) return fail(SVILS_ERR_UNSUPPORTED, "K-sharded handles index links with 32 bits");
    guard(dalloc(h, &d.elink, elink.size(), false));
    // one value per link on full sweeps; mini-batch steps index the same buffers by CSR entry (2 per link: ksh_ent)
    if (2 * nlinks >= (1ull << 32)) return fail(SVILS_ERR_UNSUPPORTED, "K-sharded handles index CSR entries with 32 bits");
    guard(dalloc(h, &d.den, std::max<uint64_t>(2 * nlinks, 1)));
    guard(dalloc(h, &d.dmax, std::max<uint64_t>(2 * nlinks, 1)));
    if (d.ksh_lowt) guard(dalloc(h, &d.earg, std::max<uint64_t>(2 * nlinks, 1)));
    guard(dalloc(h, &d.part_q2, d.nb_c));
  }
  if (rc) return rc;
  if (d.ksh) HIPCHK(hipMemcpyAsync(d.elink, elink.data(), elink.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d.rowptr, rowptr.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d.col, col.data(), col.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d.upper, upper.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  if (!items_phi.empty())
    HIPCHK(hipMemcpyAsync(d.items_phi, items_phi.data(), items_phi.size() * sizeof(Item), hipMemcpyHostToDevice, h->stream));
  if (!items_s3.empty())
    HIPCHK(hipMemcpyAsync(d.items_s3, items_s3.data(), items_s3.size() * sizeof(Item), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d.split_first, split_first.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d.split_cnt, split_cnt.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  h->cls_valid = false;
  if (d.lpl) {
    HIPCHK(hipMemcpyAsync(d.erow, erow.data(), erow.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    if (nlinks)
      HIPCHK(hipMemcpyAsync(d.links, links, 2 * nlinks * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->h_rowptr.swap(rowptr);
  h->h_upper.swap(upper);
  h->have_graph = true;
  return apply_s3_split(h);
}

int svils_set_validation(svils_handle *h, const uint32_t *pairs_y, uint64_t nv) {
  if (TILED(h)) {
    for (svils_handle *t : h->tiles) { int rc_ = svils_set_validation(t, pairs_y, nv); if (rc_) return rc_; }
    return 0;
  }
  if (!h || (!pairs_y && nv)) return fail(SVILS_ERR_ARG, "svils_set_validation: null argument");
  // captured kernel arguments hold the old validation pointers: drop every graph
  (void)hipStreamSynchronize(h->stream);
  if (h->gexec1) (void)hipGraphExecDestroy(h->gexec1);
  if (h->gexecN) (void)hipGraphExecDestroy(h->gexecN);
  for (auto &g_ : h->gexecP) if (g_) { (void)hipGraphExecDestroy(g_); g_ = nullptr; }
  h->gexec1 = h->gexecN = nullptr;

  if (nv > 0xffffffffull) return fail(SVILS_ERR_UNSUPPORTED, "too many validation pairs");
  HIPCHK(hipSetDevice(h->cfg.device));
  for (uint64_t i = 0; i < nv; ++i)
    if (pairs_y[3 * i] >= h->geo.n || pairs_y[3 * i + 1] >= h->geo.n || pairs_y[3 * i + 2] > 1)
      return fail(SVILS_ERR_ARG, "validation pair %llu out of range", (unsigned long long)i);
  DeviceState &d = h->d;
  int rc = dalloc(h, &d.vpairs, 3 * (size_t)nv, false);
  if (!rc) rc = dalloc(h, &d.uval, nv);
  if (!rc && d.ksh) rc = dalloc(h, &d.vdot, std::max<uint64_t>(nv, 1));
  if (rc) return rc;
  if (nv) HIPCHK(hipMemcpyAsync(d.vpairs, pairs_y, 3 * nv * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  d.nv = (uint32_t)nv;
  d.nb_t = tail_blocks(h->geo, d.nv);
  return 0;
}

int svils_set_state(svils_handle *h, const double *gamma, const double *lambda,
                    const uint32_t *converged) {
  if (TILED(h)) {
    if (!gamma || !lambda) return fail(SVILS_ERR_ARG, "svils_set_state: null argument");
    return tiles_set_state(h, gamma, lambda, converged);
  }
  if (!h || !gamma || !lambda) return fail(SVILS_ERR_ARG, "svils_set_state: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  const Geometry &g = h->geo;
  DeviceState &d = h->d;
  HIPCHK(hipMemsetAsync(d.gamma, 0, (size_t)g.n_alloc * g.ld * sizeof(double), h->stream));
  HIPCHK(hipMemcpy2DAsync(d.gamma, g.ld * sizeof(double), gamma, g.K * sizeof(double),
                          g.K * sizeof(double), g.n, hipMemcpyHostToDevice, h->stream));
  return state_arrived(h, lambda, converged);
}

}  // extern "C"
namespace svils_impl {
// what follows a new gamma on the device, however it got there (svils_set_state: uploaded; svils_init_gamma: drawn in place):
// lambda, the converged flags, the expectations, the bookkeeping of a handle whose state has just been replaced
int state_arrived(svils_handle *h, const double *lambda, const uint32_t *converged) {
  const Geometry &g = h->geo;
  DeviceState &d = h->d;
  h->mphi_stale = false;   // the stored rows have nothing to do with the new gamma (nor has the reference's _mphi after load_model)
  h->frozen = false;       // whatever stop the caller had seen belongs to the old state
  HIPCHK(hipMemcpyAsync(d.lambda, lambda, 2 * (size_t)g.K * sizeof(double), hipMemcpyHostToDevice, h->stream));
  DevCtrl c;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(&c, d.ctrl, sizeof c, hipMemcpyDeviceToHost));
  uint32_t *cur = d.conv + (size_t)c.parity * g.n_alloc;
  if (converged) HIPCHK(hipMemcpyAsync(cur, converged, g.n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  else HIPCHK(hipMemsetAsync(cur, 0, g.n * sizeof(uint32_t), h->stream));
  if (!d.ksh) launch_dir_exp(g, d, h->stream);   // K-sharded: the row sums cross ranks (svils_ksh_init_state)
  launch_lambda_exp(g, d, h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  h->cls_valid = false;   // the converged flags changed under the link classes
  h->cflag_dirty = true;
  h->have_state = true;
  return 0;
}
}  // namespace svils_impl
extern "C" {

int svils_get_control(svils_handle *h, svils_control *out) {
  if (TILED(h)) return svils_get_control(h->tiles[0], out);   // the loop control is replicated on the tiles
  if (!h || !out) return fail(SVILS_ERR_ARG, "svils_get_control: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc_ = settle(h)) return rc_;
  DevCtrl c;
  HIPCHK(hipMemcpy(&c, h->d.ctrl, sizeof c, hipMemcpyDeviceToHost));
  if (c.fault) return fault_error(c.fault);
  if (c.stopped) h->frozen = true;
  out->iter = c.iter; out->annealing = c.annealing; out->write_comm = c.write_comm; out->nh = c.nh;
  out->prev_h = c.prev_h; out->max_h = c.max_h; out->stopped = c.stopped; out->why = c.why;
  out->sweeps_done = c.sweeps_done; out->rows = c.rows;
  out->links_dense = c.links_dense; out->links_sparse = c.links_sparse; out->links_shortcut = c.links_shortcut;
  return 0;
}

int svils_set_control(svils_handle *h, const svils_control *in) {
  if (TILED(h)) {
    for (svils_handle *t : h->tiles) { int rc_ = svils_set_control(t, in); if (rc_) return rc_; }
    return 0;
  }
  if (!h || !in) return fail(SVILS_ERR_ARG, "svils_set_control: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  DevCtrl c;
  HIPCHK(hipMemcpy(&c, h->d.ctrl, sizeof c, hipMemcpyDeviceToHost));
  if (c.fault) return fault_error(c.fault);
  c.iter = in->iter; c.annealing = in->annealing; c.write_comm = in->write_comm; c.nh = in->nh;
  c.prev_h = in->prev_h; c.max_h = in->max_h;
  HIPCHK(hipMemcpy(h->d.ctrl, &c, sizeof c, hipMemcpyHostToDevice));
  h->frozen = false;      // the loop state is the caller's again: getters wait for the stream until a stop is seen anew
  h->cls_valid = false;   // _iter decides between the dense and the active-set class
  return 0;
}

int svils_validation_row(svils_handle *h, double *row10) {
  if (TILED(h)) return row10 ? tiles_validation_row(h, row10) : fail(SVILS_ERR_ARG, "svils_validation_row: null argument");
  if (!h || !row10) return fail(SVILS_ERR_ARG, "svils_validation_row: null argument");
  if (!h->have_state) return fail(SVILS_ERR_ARG, "svils_validation_row: call svils_set_state first");
  if (h->d.nv == 0) return fail(SVILS_ERR_ARG, "svils_validation_row: no validation set");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->d.ksh) return ksh_validation_row(h, row10);
  launch_validation(h->geo, h->d, h->prm, h->stream);
  launch_row_only(h->geo, h->d, h->prm, h->row_scratch, h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(row10, h->row_scratch, 10 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int svils_set_timing_period(svils_handle *h, uint32_t period) {
  NOT_TILED(h, "svils_set_timing_period");
  if (!h || period == 0) return fail(SVILS_ERR_ARG, "svils_set_timing_period: bad argument");
  h->tperiod = period;
  return 0;
}

int svils_synchronize(svils_handle *h) {
  if (TILED(h)) return svils_synchronize(h->tiles[0]);   // one stream for all tiles
  if (!h) return fail(SVILS_ERR_ARG, "svils_synchronize: null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  uint32_t fault = 0;
  HIPCHK(hipMemcpy(&fault, &h->d.ctrl->fault, sizeof fault, hipMemcpyDeviceToHost));
  if (fault) return fault_error(fault);
  return 0;
}

int svils_get_rows(svils_handle *h, uint32_t first, uint32_t count, double *rows) {
  if (TILED(h)) return svils_get_rows(h->tiles[0], first, count, rows);
  if (!h || (!rows && count)) return fail(SVILS_ERR_ARG, "svils_get_rows: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc_ = settle(h)) return rc_;
  DevCtrl c;
  HIPCHK(hipMemcpy(&c, h->d.ctrl, sizeof c, hipMemcpyDeviceToHost));
  if (c.fault) return fault_error(c.fault);
  if ((uint64_t)first + count > c.rows) return fail(SVILS_ERR_ARG, "rows [%u,%u) not recorded yet (have %u)", first, first + count, c.rows);
  if (c.rows - first > h->d.rows_cap) return fail(SVILS_ERR_ARG, "row %u already overwritten in the ring", first);
  // the ring wraps at rows_cap: at most two contiguous copies
  uint32_t done = 0;
  while (done < count) {
    const uint32_t slot = (first + done) % h->d.rows_cap;
    const uint32_t run = std::min(count - done, h->d.rows_cap - slot);
    HIPCHK(hipMemcpy(rows + (size_t)done * 10, h->d.rows + (size_t)slot * 10, (size_t)run * 10 * sizeof(double), hipMemcpyDeviceToHost));
    done += run;
  }
  return 0;
}

int svils_get_community_tags(svils_handle *h, uint32_t *tags, uint64_t cap, uint64_t *ntags) {
  if (TILED(h)) return ntags && (tags || !cap) ? tiles_get_community_tags(h, tags, cap, ntags) : fail(SVILS_ERR_ARG, "svils_get_community_tags: null argument");
  if (!h || !ntags || (!tags && cap)) return fail(SVILS_ERR_ARG, "svils_get_community_tags: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc_ = settle(h)) return rc_;
  const Geometry &g = h->geo;
  std::vector<uint64_t> bits((size_t)g.n * g.kw);
  HIPCHK(hipMemcpy(bits.data(), h->d.member, bits.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  *ntags = tags_of_bits(g, bits.data(), tags, cap);
  if (tags && *ntags > cap) return fail(SVILS_ERR_ARG, "svils_get_community_tags: %llu tags, room for %llu", (unsigned long long)*ntags, (unsigned long long)cap);
  return 0;
}

int svils_get_state(svils_handle *h, double *gamma, double *lambda, uint32_t *converged) {
  if (TILED(h)) return tiles_get_state(h, gamma, lambda, converged);
  if (!h) return fail(SVILS_ERR_ARG, "svils_get_state: null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc_ = settle(h)) return rc_;
  const Geometry &g = h->geo;
  if (gamma)
    HIPCHK(hipMemcpy2D(gamma, g.K * sizeof(double), h->d.gamma, g.ld * sizeof(double), g.K * sizeof(double), g.n, hipMemcpyDeviceToHost));
  if (lambda) HIPCHK(hipMemcpy(lambda, h->d.lambda, 2 * (size_t)g.K * sizeof(double), hipMemcpyDeviceToHost));
  if (converged) {
    DevCtrl c;
    HIPCHK(hipMemcpy(&c, h->d.ctrl, sizeof c, hipMemcpyDeviceToHost));
  if (c.fault) return fault_error(c.fault);
    HIPCHK(hipMemcpy(converged, h->d.conv + (size_t)c.parity * g.n_alloc, g.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  return 0;
}

int svils_get_communities(svils_handle *h, uint8_t *member) {
  if (TILED(h)) return member ? tiles_get_communities(h, member) : fail(SVILS_ERR_ARG, "svils_get_communities: null argument");
  if (!h || !member) return fail(SVILS_ERR_ARG, "svils_get_communities: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc_ = settle(h)) return rc_;
  const Geometry &g = h->geo;
  std::vector<uint64_t> bits((size_t)g.n * g.kw);
  HIPCHK(hipMemcpy(bits.data(), h->d.member, bits.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  memset(member, 0, (size_t)g.n * g.K);
  for (uint32_t p = 0; p < g.n; ++p)
    for (int v = 0; v < g.V; ++v) {
      uint64_t b = bits[(size_t)p * g.kw + v];
      while (b) {
        int lw = __builtin_ctzll(b);
        b &= b - 1;
        uint32_t k = kmap_host(g.W, g.V, lw, v);
        if (k < g.K) member[(size_t)p * g.K + k] = 1;
      }
    }
  return 0;
}

int svils_get_aux(svils_handle *h, int which, void *out) {
  NOT_TILED(h, "svils_get_aux");
  if (!h || !out) return fail(SVILS_ERR_ARG, "svils_get_aux: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const Geometry &g = h->geo;
  switch (which) {
    case 0: {
      double *el = nullptr;
      int rc = elogpi_rows(h, &el);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(h->stream));
      HIPCHK(hipMemcpy2D(out, g.K * sizeof(double), el, g.ld * sizeof(double), g.K * sizeof(double), g.n, hipMemcpyDeviceToHost));
      return 0;
    }
    case 1:
      HIPCHK(hipMemcpy(out, h->d.elogbeta, 2 * (size_t)g.K * sizeof(double), hipMemcpyDeviceToHost));
      return 0;
    case 2:
      if (h->mphi_stale) {
        launch_mphi_from_gamma(h->geo, h->d, h->prm, h->stream);
        HIPCHK(hipStreamSynchronize(h->stream));
        h->mphi_stale = false;
      }
      HIPCHK(hipMemcpy2D(out, g.K * sizeof(double), h->d.mphi, g.ld * sizeof(double), g.K * sizeof(double), g.n, hipMemcpyDeviceToHost));
      return 0;
    case 3:
      HIPCHK(hipMemcpy(out, h->d.active_cnt, g.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
      return 0;
    case 4: {
      if (!h->have_graph) return fail(SVILS_ERR_ARG, "svils_get_aux: graph not set");
      double *tl = (double *)out;
      for (uint32_t p = 0; p < g.n; ++p) tl[p] = 2.0 * (double)(h->h_rowptr[p + 1] - h->h_rowptr[p]);
      return 0;
    }
    case 5:   // profiling stamps (zeros unless the library was built with -DSVILS_STAMPS)
      HIPCHK(hipMemcpy(out, h->d.stamps, 4 * 1024 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      return 0;
    default:
      return fail(SVILS_ERR_ARG, "svils_get_aux: unknown selector %d", which);
  }
}

int svils_debug_eval(svils_handle *h, int which, const double *in, double *out, uint32_t n) {
  if (TILED(h)) return svils_debug_eval(h->tiles[0], which, in, out, n);
  if (!h || !in || !out || which < 0 || which > 3) return fail(SVILS_ERR_ARG, "svils_debug_eval: bad argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  double *din = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc((void **)&din, (size_t)std::max(n, 1u) * sizeof(double)));
  if (hipMalloc((void **)&dout, (size_t)std::max(n, 1u) * sizeof(double)) != hipSuccess) { (void)hipFree(din); return fail(SVILS_ERR_NOMEM, "hipMalloc failed"); }
  int rc = 0;
  if (hipMemcpy(din, in, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(SVILS_ERR_DEVICE, "upload failed");
  if (!rc) {
    launch_debug_eval(h->d, which, din, dout, n, h->stream);
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(SVILS_ERR_DEVICE, "debug eval failed: %s", hipGetErrorString(hipGetLastError()));
  }
  (void)hipFree(din);
  (void)hipFree(dout);
  return rc;
}

int svils_enable_timing(svils_handle *h, uint32_t kernel_mask) {
  NOT_TILED(h, "svils_enable_timing");
  if (!h) return fail(SVILS_ERR_ARG, "svils_enable_timing: null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  int rc = drain_timing(h);
  if (rc) return rc;
  h->tmask = kernel_mask;
  h->timed_sweeps.clear();
  for (int i = 0; i < SVILS_KERNEL_COUNT; ++i) { h->t_ms[i] = 0; h->t_n[i] = 0; }
  return 0;
}

int svils_get_timing(svils_handle *h, double *ms, uint64_t *launches) {
  NOT_TILED(h, "svils_get_timing");
  if (!h || !ms || !launches) return fail(SVILS_ERR_ARG, "svils_get_timing: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  int rc = drain_timing(h);
  if (rc) return rc;
  for (int i = 0; i < SVILS_KERNEL_COUNT; ++i) { ms[i] = h->t_ms[i]; launches[i] = h->t_n[i]; }
  return 0;
}

int svils_get_sweep_stats(svils_handle *h, uint32_t first, uint32_t count, uint64_t *out) {
  if (TILED(h)) return svils_get_sweep_stats(h->tiles[0], first, count, out);
  if (!h || (!out && count)) return fail(SVILS_ERR_ARG, "svils_get_sweep_stats: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  DevCtrl c;
  HIPCHK(hipMemcpy(&c, h->d.ctrl, sizeof c, hipMemcpyDeviceToHost));
  if (c.fault) return fault_error(c.fault);
  if ((uint64_t)first + count > c.sweeps_done)
    return fail(SVILS_ERR_ARG, "sweeps [%u,%u) not run yet (have %u)", first, first + count, c.sweeps_done);
  if (c.sweeps_done - first > h->d.sweep_stats_cap) return fail(SVILS_ERR_ARG, "sweep %u is no longer in the ring", first);
  for (uint32_t i = 0; i < count; ++i) {
    unsigned long long st[4];
    const uint32_t slot = (first + i) % h->d.sweep_stats_cap;
    HIPCHK(hipMemcpy(st, h->d.sweep_stats + (size_t)slot * 4, sizeof st, hipMemcpyDeviceToHost));
    if (st[3] != first + i) return fail(SVILS_ERR_DEVICE, "sweep statistics ring is inconsistent at sweep %u", first + i);
    out[3 * (size_t)i] = st[0]; out[3 * (size_t)i + 1] = st[1]; out[3 * (size_t)i + 2] = st[2];
  }
  return 0;
}

int svils_get_timed_links(svils_handle *h, uint64_t *out3) {
  NOT_TILED(h, "svils_get_timed_links");
  if (!h || !out3) return fail(SVILS_ERR_ARG, "svils_get_timed_links: null argument");
  out3[0] = out3[1] = out3[2] = 0;
  for (uint32_t sw : h->timed_sweeps) {
    uint64_t one[3];
    int rc = svils_get_sweep_stats(h, sw, 1, one);
    if (rc) return rc;
    out3[0] += one[0]; out3[1] += one[1]; out3[2] += one[2];
  }
  return 0;
}

int svils_device_buffer(svils_handle *h, svils_buffer which, void **dptr, size_t *bytes,
                        size_t *row_bytes) {
  NOT_TILED(h, "svils_device_buffer");
  if (!h || !dptr || !bytes || !row_bytes) return fail(SVILS_ERR_ARG, "svils_device_buffer: null argument");
  const Geometry &g = h->geo;
  const DeviceState &d = h->d;
  switch (which) {
    case SVILS_BUF_KVEC_A: *dptr = d.kvec_a; *bytes = g.K * sizeof(double); *row_bytes = *bytes; return 0;
    case SVILS_BUF_KVEC_C: *dptr = d.kvec_c; *bytes = 3 * (size_t)g.K * sizeof(double); *row_bytes = *bytes; return 0;
    case SVILS_BUF_GAMMA: *dptr = d.gamma; *row_bytes = g.ld * sizeof(double); *bytes = *row_bytes * g.n_alloc; return 0;
    case SVILS_BUF_ELOGPI: {   // (computed at this call on a handle that does not store them: DeviceState::skip_elogpi)
      double *el = nullptr;
      int rc = elogpi_rows(h, &el);
      if (rc) return rc;
      *dptr = el; *row_bytes = g.ld * sizeof(double); *bytes = *row_bytes * g.n_alloc; return 0;
    }
    case SVILS_BUF_MPHI:
      if (h->mphi_stale) {
        launch_mphi_from_gamma(h->geo, h->d, h->prm, h->stream);
        h->mphi_stale = false;
      }
      *dptr = d.mphi; *row_bytes = g.ld * sizeof(double); *bytes = *row_bytes * g.n_alloc; return 0;
    case SVILS_BUF_CONV: {
      // the buffer prune() writes during phase B = conv[parity ^ 1]; parity flips once per
      // sweep in phase D, and the host can mirror it as (sweeps_done & 1)
      *dptr = d.conv; *row_bytes = sizeof(uint32_t); *bytes = 2 * (size_t)g.n_alloc * sizeof(uint32_t); return 0;
    }
    case SVILS_BUF_ACTIVE: *dptr = d.active_cnt; *row_bytes = sizeof(uint32_t); *bytes = (size_t)g.n_alloc * sizeof(uint32_t); return 0;
    case SVILS_BUF_AMASK: *dptr = d.amask; *row_bytes = g.kw * sizeof(uint64_t); *bytes = *row_bytes * g.n_alloc; return 0;
    case SVILS_BUF_MEMBER: *dptr = d.member; *row_bytes = g.kw * sizeof(uint64_t); *bytes = *row_bytes * g.n_alloc; return 0;
    case SVILS_BUF_XFLAGS: *dptr = d.xflags; *row_bytes = d.xf_ld * sizeof(uint32_t); *bytes = *row_bytes * g.n_alloc; return 0;
    case SVILS_BUF_GSTAGE:
      if (!h->blocks_set) return fail(SVILS_ERR_ARG, "svils_device_buffer: SVILS_BUF_GSTAGE exists once the node blocks are declared (svils_set_node_blocks)");
      *dptr = d.gstage; *row_bytes = g.ld * sizeof(double); *bytes = *row_bytes * h->blk.bmax * h->blk.world; return 0;
    default: return fail(SVILS_ERR_ARG, "svils_device_buffer: unknown buffer %d", (int)which);
  }
}

int svils_stream(svils_handle *h, void **stream) {
  if (!h || !stream) return fail(SVILS_ERR_ARG, "svils_stream: null argument");
  *stream = (void *)h->stream;
  return 0;
}

}  // extern "C"
