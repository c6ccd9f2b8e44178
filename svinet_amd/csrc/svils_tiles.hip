// svils_tiles.hip -- column-tiled handles: k above SVILS_MAX_K on ONE device as ceil(k / SVILS_MAX_K) K-sharded slices on one
// stream, the exchanges of a K-sharded sweep summed in place by k_tiles_combine.
#include "svils_handle.h"

extern "C" {

// ---------------------------------------------------------------- column tiles: k > SVILS_MAX_K on one device
}  // extern "C"
namespace svils_impl {
struct TilePtrs { double *p[SVILS_MAX_TILES]; int n; };
// what an all-reduce over the "ranks" of a K-sharded run would leave: op 0 SUM (in tile order: reproducible), 1 MAX, 2 MIN
__global__ __launch_bounds__(256) void k_tiles_combine(TilePtrs t, size_t count, int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double v = t.p[0][i];
    for (int g = 1; g < t.n; ++g) {
      const double w = t.p[g][i];
      v = op == 0 ? v + w : op == 1 ? fmax(v, w) : fmin(v, w);
    }
    for (int g = 0; g < t.n; ++g) t.p[g][i] = v;
  }
}

int tiles_combine(svils_handle *h, svils_ksh_buffer which) {
  TilePtrs t{};
  t.n = (int)h->tiles.size();
  size_t count = 0;
  for (int i = 0; i < t.n; ++i) {
    void *p = nullptr;
    size_t n = 0;
    int rc = svils_ksh_buffer_ptr(h->tiles[(size_t)i], which, &p, &n);
    if (rc) return rc;
    if (i && n != count) return fail(SVILS_ERR_ARG, "column tiles: exchange buffer %d has different sizes on the tiles", (int)which);
    count = n;
    t.p[i] = (double *)p;
  }
  if (!count) return 0;
  const int op = which == SVILS_KSH_DMAX ? 1 : which == SVILS_KSH_EARG ? 2 : 0;
  const uint32_t nb = (uint32_t)std::min<size_t>((count + 255) / 256, 4096);
  hipLaunchKernelGGL(k_tiles_combine, dim3(nb), dim3(256), 0, h->tiles[0]->stream, t, count, op);
  HIPCHK(hipGetLastError());
  return 0;
}
int tiles_phase(svils_handle *h, svils_kphase ph) {
  for (svils_handle *t : h->tiles) {
    int rc = svils_ksweep_phase(t, ph);
    if (rc) return rc;
  }
  return 0;
}
// row sums and Elogpi of a freshly set state (svils_ksh_init_state over the tiles); needs graph and state on every tile
int tiles_try_init(svils_handle *h) {
  if (h->tiles_inited || !h->have_graph || !h->have_state) return 0;
  int rc;
  if ((rc = tiles_phase(h, SVILS_KPHASE_INIT_ROWS))) return rc;
  if ((rc = tiles_combine(h, SVILS_KSH_ROWX))) return rc;
  if ((rc = tiles_phase(h, SVILS_KPHASE_INIT_EXPAND))) return rc;
  h->tiles_inited = true;
  return 0;
}
int tiles_need_init(svils_handle *h, const char *who) {
  int rc = tiles_try_init(h);
  if (rc) return rc;
  if (!h->tiles_inited) return fail(SVILS_ERR_ARG, "%s: a column-tiled handle (k > SVILS_MAX_K) needs svils_set_graph and svils_set_state first", who);
  return 0;
}

int tiles_create(const svils_config *cfg, svils_handle **out) {
  if (cfg->k > SVILS_MAX_K_TOTAL) return fail(SVILS_ERR_UNSUPPORTED, "k=%u exceeds SVILS_MAX_K_TOTAL=%d (the reference's community ids are 16-bit, src/linksampling.cc:635)", cfg->k, SVILS_MAX_K_TOTAL);
  const uint32_t ne = cfg->node_end ? cfg->node_end : cfg->n;
  if (cfg->node_begin != 0 || ne != cfg->n || cfg->n_alloc > cfg->n)
    return fail(SVILS_ERR_UNSUPPORTED, "k=%u > SVILS_MAX_K=%d runs as column tiles of the whole graph: node blocks are not available (shard the columns instead: svils_config.k_total)", cfg->k, SVILS_MAX_K);
  const uint32_t G = (cfg->k + SVILS_MAX_K - 1) / SVILS_MAX_K;
  svils_handle *h = new (std::nothrow) svils_handle();
  if (!h) return fail(SVILS_ERR_NOMEM, "out of host memory");
  h->cfg = *cfg;
  h->geo.n = cfg->n;
  h->geo.K = h->geo.Kt = cfg->k;
  for (uint32_t r = 0; r < G; ++r) {
    svils_config c = *cfg;
    c.k_begin = (uint32_t)((uint64_t)cfg->k * r / G);
    c.k = (uint32_t)((uint64_t)cfg->k * (r + 1) / G) - c.k_begin;
    c.k_total = cfg->k;
    svils_handle *t = nullptr;
    int rc = svils_create(&c, &t);
    if (rc) { svils_destroy(h); return rc; }
    h->tiles.push_back(t);
    if (r) {   // one stream for all tiles: their phases and the sums between them are one sequence
      (void)hipStreamSynchronize(t->stream);
      (void)hipStreamDestroy(t->stream);
      t->stream = h->tiles[0]->stream;
      t->stream_shared = true;
    }
  }
  h->stream = h->tiles[0]->stream;
  h->stream_shared = true;
  *out = h;
  return 0;
}

int tiles_set_state(svils_handle *h, const double *gamma, const double *lambda, const uint32_t *converged) {
  const uint32_t n = h->cfg.n, K = h->cfg.k;
  std::vector<double> slice;
  for (svils_handle *t : h->tiles) {
    const uint32_t k0 = t->cfg.k_begin, w = t->cfg.k;
    slice.resize((size_t)n * w);
    for (uint32_t i = 0; i < n; ++i) memcpy(&slice[(size_t)i * w], gamma + (size_t)i * K + k0, (size_t)w * sizeof(double));
    int rc = svils_set_state(t, slice.data(), lambda + 2 * (size_t)k0, converged);
    if (rc) return rc;
  }
  h->have_state = true;
  h->tiles_inited = false;
  h->frozen = false;
  return tiles_try_init(h);
}

int tiles_get_state(svils_handle *h, double *gamma, double *lambda, uint32_t *converged) {
  const uint32_t n = h->cfg.n, K = h->cfg.k;
  std::vector<double> slice;
  for (svils_handle *t : h->tiles) {
    const uint32_t k0 = t->cfg.k_begin, w = t->cfg.k;
    if (gamma) slice.resize((size_t)n * w);
    int rc = svils_get_state(t, gamma ? slice.data() : nullptr, lambda ? lambda + 2 * (size_t)k0 : nullptr, t == h->tiles[0] ? converged : nullptr);
    if (rc) return rc;
    if (gamma)
      for (uint32_t i = 0; i < n; ++i) memcpy(gamma + (size_t)i * K + k0, &slice[(size_t)i * w], (size_t)w * sizeof(double));
  }
  return 0;
}

// one sweep = the phases of a K-sharded sweep on every tile, the exchanges summed in place (svils_sweep_ksharded)
int tiles_sweep(svils_handle *h, uint32_t nsweeps) {
  int rc = tiles_need_init(h, "svils_sweep");
  if (rc) return rc;
  svils_handle *t0 = h->tiles[0];
  if (nsweeps > (uint64_t)t0->d.rows_cap * t0->prm.reportfreq)
    return fail(SVILS_ERR_ARG, "svils_sweep: at most %llu sweeps per call", (unsigned long long)t0->d.rows_cap * t0->prm.reportfreq);
  for (uint32_t i = 0; i < nsweeps; ++i) {
    if (t0->d.ksh_log) {
      if ((rc = tiles_phase(h, SVILS_KPHASE_DENMAX))) return rc;
      if ((rc = tiles_combine(h, SVILS_KSH_DMAX))) return rc;
    }
    if ((rc = tiles_phase(h, SVILS_KPHASE_DEN))) return rc;
    if ((rc = tiles_combine(h, SVILS_KSH_DEN))) return rc;
    if (t0->d.ksh_lowt && (rc = tiles_combine(h, SVILS_KSH_EARG))) return rc;
    if ((rc = tiles_phase(h, SVILS_KPHASE_PHI))) return rc;
    if ((rc = tiles_combine(h, SVILS_KSH_ROWX))) return rc;
    if ((rc = tiles_phase(h, SVILS_KPHASE_FIN))) return rc;
    if ((rc = tiles_combine(h, SVILS_KSH_Q2))) return rc;
    if ((rc = tiles_phase(h, SVILS_KPHASE_LAMBDA))) return rc;
    if ((rc = tiles_combine(h, SVILS_KSH_VDOT))) return rc;
    if ((rc = tiles_phase(h, SVILS_KPHASE_STOP))) return rc;
  }
  return 0;
}

int ksh_validation_row_finish(svils_handle *h, double *row10);
int tiles_validation_row(svils_handle *h, double *row10) {
  int rc = tiles_need_init(h, "svils_validation_row");
  if (rc) return rc;
  for (svils_handle *t : h->tiles) {
    launch_ksh_phase(t->geo, t->d, t->prm, 8, t->stream);   // k_vdot_ksh alone
    HIPCHK(hipGetLastError());
  }
  if ((rc = tiles_combine(h, SVILS_KSH_VDOT))) return rc;
  return ksh_validation_row_finish(h->tiles[0], row10);
}

int tiles_get_communities(svils_handle *h, uint8_t *member) {
  const uint32_t n = h->cfg.n, K = h->cfg.k;
  std::vector<uint8_t> slice;
  for (svils_handle *t : h->tiles) {
    const uint32_t k0 = t->cfg.k_begin, w = t->cfg.k;
    slice.resize((size_t)n * w);
    int rc = svils_get_communities(t, slice.data());
    if (rc) return rc;
    for (uint32_t i = 0; i < n; ++i) memcpy(member + (size_t)i * K + k0, &slice[(size_t)i * w], w);
  }
  return 0;
}

// (node, community) pairs, by node, the communities of a node ascending
int tiles_get_community_tags(svils_handle *h, uint32_t *tags, uint64_t cap, uint64_t *ntags) {
  std::vector<uint64_t> keys;   // node << 32 | community
  std::vector<uint32_t> part;
  for (svils_handle *t : h->tiles) {
    uint64_t nt = 0;
    int rc = svils_get_community_tags(t, nullptr, 0, &nt);
    if (rc) return rc;
    part.resize(2 * (size_t)nt);
    if ((rc = svils_get_community_tags(t, part.data(), nt, &nt))) return rc;
    for (uint64_t i = 0; i < nt; ++i) keys.push_back((uint64_t)part[2 * i] << 32 | (uint64_t)(part[2 * i + 1] + t->cfg.k_begin));
  }
  std::sort(keys.begin(), keys.end());
  *ntags = keys.size();
  if (!tags) return 0;
  if (keys.size() > cap) return fail(SVILS_ERR_ARG, "svils_get_community_tags: %llu tags, room for %llu", (unsigned long long)keys.size(), (unsigned long long)cap);
  for (size_t i = 0; i < keys.size(); ++i) { tags[2 * i] = (uint32_t)(keys[i] >> 32); tags[2 * i + 1] = (uint32_t)keys[i]; }
  return 0;
}
}  // namespace svils_impl
