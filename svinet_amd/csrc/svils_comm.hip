// svils_comm.hip -- multi-GPU, node blocks: RCCL bound at run time, the node blocks of a run and the
// split of the s3 pass, node-block sweeps with their two exchange points (eager or captured into hipGraphs with the
// collectives inside), mini-batch steps over node blocks, and the gathers behind the final files.
#include "svils_handle.h"

namespace svils_impl {

// The staging of the row exchange, [world][bmax][ld]; slice `rank` is where the light finalise pass writes.
int apply_blocks(svils_handle *h, int rank, int world, const uint32_t *bounds, bool explicit_bounds) {
  const Geometry &g = h->geo;
  if (world < 1 || world > SVILS_MAX_WORLD || rank < 0 || rank >= world)
    return fail(SVILS_ERR_ARG, "node blocks: rank %d of %d (at most %d ranks)", rank, world, SVILS_MAX_WORLD);
  Blocks b{};
  b.world = (uint32_t)world;
  b.chunk = 0;
  b.nchunks = 1;
  if (bounds) {
    for (int r = 0; r <= world; ++r) b.bounds[r] = bounds[r];
  } else {   // equal blocks of ceil(n / world) nodes
    const uint32_t B = (g.n + (uint32_t)world - 1) / (uint32_t)world;
    for (int r = 0; r <= world; ++r) b.bounds[r] = (uint32_t)std::min<uint64_t>(g.n, (uint64_t)r * B);
  }
  if (b.bounds[0] != 0 || b.bounds[world] != g.n) return fail(SVILS_ERR_ARG, "node blocks: bounds must run from 0 to n = %u", g.n);
  b.bmax = 0;
  for (int r = 0; r < world; ++r) {
    if (b.bounds[r + 1] < b.bounds[r]) return fail(SVILS_ERR_ARG, "node blocks: bounds must not decrease (rank %d)", r);
    b.bmax = std::max(b.bmax, b.bounds[r + 1] - b.bounds[r]);
  }
  if (b.bounds[rank] != g.node_begin || b.bounds[rank + 1] != g.node_end)
    return fail(SVILS_ERR_ARG, "node blocks: rank %d of %d owns [%u,%u) but the handle was created for [%u,%u)", rank, world,
                b.bounds[rank], b.bounds[rank + 1], g.node_begin, g.node_end);
  if (h->blocks_set) {
    if (h->blk.world != b.world || memcmp(h->blk.bounds, b.bounds, sizeof(uint32_t) * (size_t)(world + 1)) != 0 || h->rank != rank)
      return fail(SVILS_ERR_ARG, "node blocks: already declared differently for this handle");
    if (explicit_bounds && !h->blocks_explicit) {
      h->blocks_explicit = true;
      return apply_s3_split(h);
    }
    return 0;
  }
  if (h->d.ksh) return fail(SVILS_ERR_ARG, "node blocks: a K-sharded handle holds every node");
  int rc = dalloc(h, &h->d.gstage, (size_t)world * std::max(b.bmax, 1u) * g.ld);
  if (rc) return rc;
  h->d.gown = h->d.gstage + (size_t)rank * b.bmax * g.ld;
  h->blk = b;
  h->rank = rank;
  h->world = world;
  h->blocks_set = true;
  h->blocks_explicit = explicit_bounds;
  return apply_s3_split(h);
}
// a whole-graph handle that never heard of blocks is a world of one
int ensure_blocks(svils_handle *h) {
  if (h->blocks_set) return 0;
  if (h->geo.node_begin != 0 || h->geo.node_end != h->geo.n)
    return fail(SVILS_ERR_ARG, "this node-block handle needs svils_set_node_blocks (or svils_comm_init) first");
  return apply_blocks(h, 0, 1, nullptr, false);
}

// Node-block sweeps with caller-given (work-balanced) blocks: the s3 pass is not tied to the node blocks -- it reads the
// replicated mean indicators of both endpoints and leaves a K-vector -- so the link list is simply cut into `world`
// equal runs.  (With first-appearance numbering the low blocks hold the upper ends of most links: blocks balanced by
// CSR entries would leave rank 0 with twice its share of the s3 pass.)
int apply_s3_split(svils_handle *h) {
  if (!h->have_graph || !h->blocks_set || !h->blocks_explicit || h->world <= 1) return 0;
  const Geometry &g = h->geo;
  DeviceState &d = h->d;
  const uint64_t L = d.nlinks;
  const uint64_t lb = L * (uint64_t)h->rank / (uint64_t)h->world, le = L * ((uint64_t)h->rank + 1) / (uint64_t)h->world;
  HIPCHK(hipStreamSynchronize(h->stream));
  drop_graphs_of(h);
  auto cap = [](uint64_t x, uint32_t lim) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(x, lim)); };
  d.link_begin = lb;
  d.link_end = le;
  uint32_t nb_c;
  if (d.lpl) {
    d.s3_threads = lpl_s3_threads(g.K, le - lb);   // the block shape goes by THIS rank's share of the links (12-wave blocks only pay beyond 192 x 512)
    nb_c = cap((le - lb + d.s3_threads - 1) / d.s3_threads, 192);
  } else {
    const int G = 64 / g.W;
    const uint32_t ch = 32u * (uint32_t)G;
    std::vector<Item> items;
    // first node whose links reach past lb
    uint32_t p = (uint32_t)(std::upper_bound(h->h_linkptr.begin(), h->h_linkptr.end(), lb) - h->h_linkptr.begin());
    p = p ? p - 1 : 0;
    for (; p < g.n && h->h_linkptr[p] < le; ++p) {
      const uint64_t a = std::max(lb, h->h_linkptr[p]), b = std::min(le, h->h_linkptr[p + 1]);
      if (b <= a) continue;
      chunk_row(items, p, h->h_upper[p] + (uint32_t)(a - h->h_linkptr[p]), (uint32_t)(b - a), ch, nullptr, nullptr, nullptr);
    }
    dfree(h, &d.items_s3);
    int rc = dalloc(h, &d.items_s3, items.size(), false);
    if (rc) return rc;
    if (!items.empty()) HIPCHK(hipMemcpyAsync(d.items_s3, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    d.nitems_s3 = (uint32_t)items.size();
    d.item0_s3 = 0;
    nb_c = cap((d.nitems_s3 + 3) / 4, 2 * rpw_resident_blocks(g, 1, h->cfg.device));
  }
  if (nb_c > d.nb_c) {
    dfree(h, &d.part_c);
    int rc = dalloc(h, &d.part_c, (size_t)nb_c * g.K);
    if (rc) return rc;
  }
  d.nb_c = nb_c;
  return 0;
}

}  // namespace svils_impl

extern "C" {

// ---------------------------------------------------------------- RCCL, bound at run time
}  // extern "C"
namespace svils_impl {
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.lib) return 0;
  // SVILS_RCCL_LIBRARY names the RCCL build to bind (a site's own librccl; tests/ point it at a transport that
  // lets several processes share one GPU -- tests/fakerccl).  A named library that does not load is an error:
  // there is no silent second choice.
  void *lib = nullptr;
  const char *named = getenv("SVILS_RCCL_LIBRARY");
  if (named && *named) {
    lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(SVILS_ERR_UNSUPPORTED, "SVILS_RCCL_LIBRARY=%s does not load (%s)", named, dlerror());
  } else {
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  }
  if (!lib) return fail(SVILS_ERR_UNSUPPORTED, "librccl not found (%s): the multi-GPU driver needs RCCL", dlerror());
#define BIND(F)                                                                        \
  do {                                                                                 \
    *(void **)(&g_rccl.F) = dlsym(lib, "nccl" #F);                                     \
    if (!g_rccl.F) return fail(SVILS_ERR_UNSUPPORTED, "librccl lacks nccl" #F);        \
  } while (0)
  BIND(GetUniqueId); BIND(CommInitRank); BIND(CommDestroy); BIND(AllReduce); BIND(AllGather); BIND(Broadcast);
  BIND(GroupStart); BIND(GroupEnd); BIND(GetErrorString);
#undef BIND
  *(void **)(&g_rccl.CommCount) = dlsym(lib, "ncclCommCount");
  *(void **)(&g_rccl.CommCuDevice) = dlsym(lib, "ncclCommCuDevice");
  *(void **)(&g_rccl.CommUserRank) = dlsym(lib, "ncclCommUserRank");
  *(void **)(&g_rccl.GetVersion) = dlsym(lib, "ncclGetVersion");
  g_rccl.lib = lib;
  return 0;
}

void comm_destroy(svils_handle *h) {
  if (h->comm_rows && g_rccl.lib) (void)g_rccl.CommDestroy(h->comm_rows);
  h->comm_rows = nullptr;
  if (h->comm && g_rccl.lib) (void)g_rccl.CommDestroy(h->comm);
  h->comm = nullptr;
}

// The second communicator (same ranks, same devices) that carries the chunked row exchange on comm_stream.  Collective:
// every rank reaches it at the same point of its first pipelined sweep.  Rank 0 draws a fresh unique id and hands it to
// the others over the first communicator (128 bytes through the staging word of the handle's stream).
int ensure_row_comm(svils_handle *h) {
  if (h->comm_rows || !h->comm) return 0;
  if (h->opt.one_comm) {   // A/B knob: rows share the first communicator
    h->comm_rows = nullptr;
    return 0;
  }
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  if (h->rank == 0) NCCLCHK(g_rccl.GetUniqueId(&id));
  unsigned char *dev = nullptr;
  HIPCHK(hipMalloc(&dev, sizeof id));
  HIPCHK(hipMemcpyAsync(dev, &id, sizeof id, hipMemcpyHostToDevice, h->stream));
  NCCLCHK(g_rccl.Broadcast(dev, dev, sizeof id, ncclUint8, 0, h->comm, h->stream));
  HIPCHK(hipMemcpyAsync(&id, dev, sizeof id, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  (void)hipFree(dev);
  NCCLCHK(g_rccl.CommInitRank(&h->comm_rows, h->world, id, h->rank));
  return 0;
}
}  // namespace svils_impl
extern "C" {

int svils_comm_unique_id(void *id128) {
  if (!id128) return fail(SVILS_ERR_ARG, "svils_comm_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == SVILS_COMM_ID_BYTES, "ncclUniqueId size");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return 0;
}

int svils_comm_init(svils_handle *h, const void *id128, int rank, int world) {
  NOT_TILED(h, "svils_comm_init");
  if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return fail(SVILS_ERR_ARG, "svils_comm_init: bad argument");
  if (h->comm) return fail(SVILS_ERR_ARG, "svils_comm_init: communicator already initialised");
  int rc;
  if (!h->d.ksh) {
    // the node blocks: what svils_set_node_blocks declared, else equal blocks of ceil(n / world) nodes
    if (h->blocks_set && ((int)h->blk.world != world || h->rank != rank))
      return fail(SVILS_ERR_ARG, "svils_comm_init: rank %d of %d, but svils_set_node_blocks declared rank %d of %u", rank, world,
                  h->rank, h->blk.world);
    if (!h->blocks_set && (rc = apply_blocks(h, rank, world, nullptr, false))) return rc;
  }
  rc = rccl_load();
  if (rc) return rc;
  HIPCHK(hipSetDevice(h->cfg.device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  NCCLCHK(g_rccl.CommInitRank(&h->comm, world, id, rank));
  h->rank = rank;
  h->world = world;
  return 0;
}

int svils_comm_query(svils_handle *h, svils_comm_info *out) {
  NOT_TILED(h, "svils_comm_query");
  if (!h || !out) return fail(SVILS_ERR_ARG, "svils_comm_query: null argument");
  if (!h->comm) return fail(SVILS_ERR_ARG, "svils_comm_query: the handle has no communicator (svils_comm_init)");
  memset(out, 0, sizeof *out);
  out->nranks = out->rank = out->device = out->version = -1;
  if (g_rccl.CommCount) NCCLCHK(g_rccl.CommCount(h->comm, &out->nranks));
  if (g_rccl.CommUserRank) NCCLCHK(g_rccl.CommUserRank(h->comm, &out->rank));
  if (g_rccl.CommCuDevice) NCCLCHK(g_rccl.CommCuDevice(h->comm, &out->device));
  if (g_rccl.GetVersion) { int v = -1; if (g_rccl.GetVersion(&v) == ncclSuccess) out->version = v; }
  out->row_comm = h->comm_rows ? 1 : 0;
  if (out->device >= 0) (void)hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, out->device);
  Dl_info di;
  if (dladdr((void *)g_rccl.AllReduce, &di) && di.dli_fname) snprintf(out->library, sizeof out->library, "%s", di.dli_fname);
  return 0;
}

}  // extern "C"
namespace svils_impl {
// the exchanges of one sharded sweep: K-vector all-reduces are latency-bound, the row
// gather carries N*ld*8 bytes; both all-gathers are in place (send block = own slice of the receive buffer)
int exchange_sum(svils_handle *h, double *v, size_t count) {
  if (!h->comm) return 0;
  Timed t(h, SVILS_KERNEL_EXCHANGE);
  NCCLCHK(g_rccl.AllReduce(v, v, count, ncclDouble, ncclSum, h->comm, h->stream));
  return 0;
}
// chunks of the pipelined row exchange: one (a grouped all-gather) while the whole n-by-k payload is below
// 256 MB, then one per 128 MB, at most eight
uint32_t exchange_chunks(const svils_handle *h) {
  if (h->xchunks) return h->xchunks;
  const uint64_t bytes = (uint64_t)h->geo.n * h->geo.ld * sizeof(double);
  return (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, bytes / (128ull << 20)));
}

// The ONE row exchange of a node-block sweep, between the light finalise pass and the s3 pass:
//   all-reduce(SUM) of `sum[k]` (K doubles)  +  the unscaled new rows of every block, staged in gstage [world][bmax][ld]
//   -> k_expand_all: annealing scale, gamma, Elogpi / exp(Elogpi), mean indicators of the other blocks, prune() flags of
//      EVERY row (computed redundantly from identical bytes: flags are not exchanged).
// Small payloads: one grouped launch {all-reduce, in-place all-gather of the slices padded to the largest block}.
// From 256 MB on the rows travel in C chunks on the communication stream and a second communicator (chunk c = rows
// [s c / C, s (c + 1) / C) of EVERY block of s rows: one grouped launch of `world` in-place broadcasts with the exact
// counts, rank r the root of its own rows); as soon as chunk c has arrived the compute stream expands it while chunk
// c + 1 is on the links.  Exposed: the first chunk's transfer and the last chunk's expansion.
int exchange_rows_and_expand(svils_handle *h) {
  const Geometry &g = h->geo;
  const DeviceState &d = h->d;
  Blocks b = h->blk;
  const uint32_t C = h->comm ? exchange_chunks(h) : 1u;
  if (C <= 1) {
    if (h->comm) {
      Timed t(h, SVILS_KERNEL_EXCHANGE);
      // The all-gather moves world * bmax rows.  Blocks balanced by WORK are far from equal in rows where the numbering
      // puts the hubs first (ca-AstroPh on 8 ranks: 574 ... 6 775 nodes, world * bmax = 3.0 n): beyond 1.5 n the rows go
      // as `world` in-place broadcasts with the exact counts in the same grouped launch (the form of the chunked
      // exchange below) -- n rows on the links instead of world * bmax.
      // (option row_exchange = 1 / 2 forces one form: A/B on real links, and the tests' way to put the grouped
      //  {all-reduce, broadcasts} launch through the real librccl on a world of one)
      const bool padded = h->opt.row_exchange != 2 && ((uint64_t)b.bmax * b.world * 2 <= 3 * (uint64_t)g.n || h->opt.row_exchange == 1);
      NCCLCHK(g_rccl.GroupStart());
      NCCLCHK(g_rccl.AllReduce(d.kvec_a, d.kvec_a, g.K, ncclDouble, ncclSum, h->comm, h->stream));
      if (padded) {
        NCCLCHK(g_rccl.AllGather(d.gown, d.gstage, (size_t)b.bmax * g.ld, ncclDouble, h->comm, h->stream));
      } else {
        for (int r = 0; r < h->world; ++r) {
          const size_t rows = b.bounds[r + 1] - b.bounds[r];
          if (!rows) continue;
          double *gp = d.gstage + (size_t)r * b.bmax * g.ld;
          NCCLCHK(g_rccl.Broadcast(gp, gp, rows * g.ld, ncclDouble, r, h->comm, h->stream));
        }
      }
      NCCLCHK(g_rccl.GroupEnd());
    }
    return run_phase(h, SVILS_PHASE_EXPAND_ALL, false, true);
  }
  if (!h->comm_stream) HIPCHK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  if (!h->ev_ready) HIPCHK(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
  {
    int rc = ensure_row_comm(h);
    if (rc) return rc;
  }
  ncclComm_t rows_comm = h->comm_rows ? h->comm_rows : h->comm;
  while (h->ev_chunk.size() < C) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->ev_chunk.push_back(e);
  }
  Timed t(h, SVILS_KERNEL_EXCHANGE);   // on the compute stream: from "rows may leave" to "last chunk expanded"
  // `sum` first, on the compute stream (k_expand_all reads it), and the rows leave behind it: the K doubles cost one
  // small-collective latency in front of a transfer of hundreds of megabytes, and no two collectives of this handle are
  // ever in flight on two streams at once (a transport that runs its host side on one thread per process -- the tests'
  // -- would otherwise see rank A inside the all-reduce and rank B inside the first broadcast, each waiting for the other)
  NCCLCHK(g_rccl.AllReduce(d.kvec_a, d.kvec_a, g.K, ncclDouble, ncclSum, h->comm, h->stream));
  HIPCHK(hipEventRecord(h->ev_ready, h->stream));
  HIPCHK(hipStreamWaitEvent(h->comm_stream, h->ev_ready, 0));
  b.nchunks = C;
  for (uint32_t c = 0; c < C; ++c) {
    NCCLCHK(g_rccl.GroupStart());
    for (int r = 0; r < h->world; ++r) {
      uint32_t lo, hi;
      chunk_range(b.bounds[r + 1] - b.bounds[r], c, C, &lo, &hi);
      if (hi <= lo) continue;
      double *gp = d.gstage + ((size_t)r * b.bmax + lo) * g.ld;
      NCCLCHK(g_rccl.Broadcast(gp, gp, (size_t)(hi - lo) * g.ld, ncclDouble, r, rows_comm, h->comm_stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    HIPCHK(hipEventRecord(h->ev_chunk[c], h->comm_stream));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_chunk[c], 0));
    b.chunk = c;
    launch_expand_all(g, d, h->prm, b, h->stream);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// one node-block sweep: two exchange points, whatever the annealing flag says (nothing here looks at the control block)
int sharded_sweep_once(svils_handle *h) {
  int rc;
  if ((rc = run_phase(h, SVILS_PHASE_A, false, true))) return rc;
  if ((rc = run_phase(h, SVILS_PHASE_B_LIGHT, false, true))) return rc;
  if ((rc = exchange_rows_and_expand(h))) return rc;
  if ((rc = run_phase(h, SVILS_PHASE_C, false, true))) return rc;
  if ((rc = exchange_sum(h, h->d.kvec_c, 3 * (size_t)h->geo.K))) return rc;
  return run_phase(h, SVILS_PHASE_D, false, true);
}

// `nsweeps` node-block sweeps, collectives included, captured into an executable graph.  RCCL's collectives are
// stream-capturable; the communication stream of the pipelined exchange forks from and joins the handle's stream through
// events, which capture follows.  Anything that fails ends the capture and the caller stays eager for good.
hipGraphExec_t capture_sharded(svils_handle *h, uint32_t nsweeps) {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  const uint64_t issued = h->sweeps_issued;
  const uint32_t saved = h->tmask;
  h->tmask = 0;
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed) != hipSuccess) { h->tmask = saved; (void)hipGetLastError(); return nullptr; }
  int rc = 0;
  for (uint32_t i = 0; i < nsweeps && !rc; ++i) rc = sharded_sweep_once(h);
  const hipError_t e = hipStreamEndCapture(h->stream, &graph);
  h->tmask = saved;
  h->sweeps_issued = issued;   // nothing ran
  if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); return nullptr; }
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { exec = nullptr; (void)hipGetLastError(); }
  (void)hipGraphDestroy(graph);
  return exec;
}
}  // namespace svils_impl
extern "C" {

int svils_sweep_sharded(svils_handle *h, uint32_t nsweeps) {
  NOT_TILED(h, "svils_sweep_sharded");
  if (!h) return fail(SVILS_ERR_ARG, "svils_sweep_sharded: null handle");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_sweep_sharded: set graph and state first");
  if (!h->comm && h->world != 1) return fail(SVILS_ERR_ARG, "svils_sweep_sharded: call svils_comm_init first");
  if (!h->comm && !(h->geo.node_begin == 0 && h->geo.node_end == h->geo.n))
    return fail(SVILS_ERR_ARG, "svils_sweep_sharded: a node-block handle needs svils_comm_init");
  if (h->stoch) return fail(SVILS_ERR_ARG, "svils_sweep_sharded: the handle is in mini-batch mode");
  if (h->d.ksh) return fail(SVILS_ERR_ARG, "svils_sweep_sharded: a K-sharded handle is driven by svils_sweep_ksharded");
  if (nsweeps > (uint64_t)h->d.rows_cap * h->prm.reportfreq)
    return fail(SVILS_ERR_ARG, "svils_sweep_sharded: at most %llu sweeps per call",
                (unsigned long long)h->d.rows_cap * h->prm.reportfreq);
  HIPCHK(hipSetDevice(h->cfg.device));
  int rc = ensure_blocks(h);
  if (rc) return rc;
  // The sweep has the same shape in both phases of a run (the annealing scale is applied on the device, behind the
  // exchange), so nothing here reads the control block and whole runs of sweeps replay as hipGraphs -- under the same
  // rule as svils_sweep: eager until the handle has run graph_after sweeps, timing brackets need eager launches.
  // Option sharded_graphs = 0 keeps every sweep eager.  Every rank takes the same decisions (same arguments, same
  // history, same options), so the ranks enqueue the same collectives in the same order whether they replay or launch.
  // (an option of the handle, changed with svils_set_option: bench.py times an eager window first and a replayed one after
  // it, so that a first contact with real multi-GPU RCCL that blocks under capture still leaves the eager number behind)
  const bool graphs_wanted = h->opt.sharded_graphs != 0;
  const bool warm = h->sgexec[0] != nullptr || h->sweeps_issued + nsweeps >= h->graph_after || nsweeps >= 64;
  uint32_t left = nsweeps;
  if (graphs_wanted && h->sgraphs_ok && h->tmask == 0 && nsweeps >= 4 && warm) {
    // the first sweep of a handle runs eagerly: lazily created objects (communication stream, second communicator,
    // events, the first stand-alone classification) must exist before a capture
    if (h->sweeps_issued == 0) { if ((rc = sharded_sweep_once(h))) return rc; --left; }
    if ((rc = ensure_classes(h))) return rc;
    for (int i = (int)svils_handle::kGraphMaxLog; i >= 0 && h->sgraphs_ok; --i) {
      const uint32_t m = 1u << i;
      if (left < m) continue;
      if (!h->sgexec[i]) {
        h->sgexec[i] = capture_sharded(h, m);
        if (!h->sgexec[i]) {
          if (i == 0) { h->sgraphs_ok = false; drop_graphs_of(h); }   // not even one sweep captures: eager from now on
          continue;
        }
      }
      for (; left >= m; left -= m) {
        HIPCHK(hipGraphLaunch(h->sgexec[i], h->stream));
        h->sweeps_issued += m;
      }
    }
  }
  for (; left > 0; --left)
    if ((rc = sharded_sweep_once(h))) return rc;
  return 0;
}

}  // extern "C"
namespace svils_impl {
int step_phase_impl(svils_handle *h, svils_phase phase, bool fused);

// the rows every rank touched in this step: the window [b, e) of every rank's block, for gamma, mphi and the
// packed flags -- one grouped launch of world broadcasts per array (rank r is the root of its own window)
int exchange_windows(svils_handle *h, uint32_t b, uint32_t e) {
  if (!h->comm || e <= b) return 0;
  Timed t(h, SVILS_KERNEL_EXCHANGE);
  const Geometry &g = h->geo;
  const DeviceState &d = h->d;
  const size_t B = g.n_alloc / (size_t)h->world, rows = e - b;
  NCCLCHK(g_rccl.GroupStart());
  for (int r = 0; r < h->world; ++r) {
    const size_t row0 = (size_t)r * B + b;
    double *gp = d.gamma + row0 * g.ld, *mp = d.mphi + row0 * g.ld;
    uint32_t *xp = d.xflags + row0 * d.xf_ld;
    NCCLCHK(g_rccl.Broadcast(gp, gp, rows * g.ld, ncclDouble, r, h->comm, h->stream));
    NCCLCHK(g_rccl.Broadcast(mp, mp, rows * g.ld, ncclDouble, r, h->comm, h->stream));
    NCCLCHK(g_rccl.Broadcast(xp, xp, rows * d.xf_ld, ncclUint32, r, h->comm, h->stream));
  }
  NCCLCHK(g_rccl.GroupEnd());
  return 0;
}
}  // namespace svils_impl
extern "C" {

// Mini-batch (Robbins-Monro) steps over node-block shards with the exchanges issued here: the global step of
// the north_star -- all-reduce of the K-vectors, the touched gamma (and mphi, flag) rows of every rank's window.
int svils_step_sharded(svils_handle *h, uint32_t nsteps) {
  NOT_TILED(h, "svils_step_sharded");
  if (!h) return fail(SVILS_ERR_ARG, "svils_step_sharded: null handle");
  if (!h->stoch) return fail(SVILS_ERR_ARG, "svils_step_sharded: call svils_set_stochastic first");
  if (!h->scfg.shard_block) return fail(SVILS_ERR_ARG, "svils_step_sharded: svils_set_stochastic needs shard_block (the node-block size)");
  if (!h->comm && h->geo.n_alloc != h->scfg.shard_block) return fail(SVILS_ERR_ARG, "svils_step_sharded: call svils_comm_init first");
  if (h->blocks_explicit)
    return fail(SVILS_ERR_ARG, "svils_step_sharded: mini-batch steps need the equal node blocks of svils_comm_init, not caller-given ones");
  if (h->comm && ((size_t)h->geo.n_alloc != (size_t)h->world * h->scfg.shard_block ||
                  h->scfg.shard_block != (h->geo.n + (uint32_t)h->world - 1) / (uint32_t)h->world))
    return fail(SVILS_ERR_ARG, "svils_step_sharded: need shard_block = ceil(n / world) = %u and n_alloc = world * shard_block (have %u, %u)",
                (h->geo.n + (uint32_t)h->world - 1) / (uint32_t)h->world, h->scfg.shard_block, h->geo.n_alloc);
  if (nsteps > (uint64_t)h->d.rows_cap * h->prm.reportfreq)
    return fail(SVILS_ERR_ARG, "svils_step_sharded: at most %llu steps per call",
                (unsigned long long)h->d.rows_cap * h->prm.reportfreq);
  const Geometry &g = h->geo;
  for (uint32_t s = 0; s < nsteps; ++s) {
    int rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_A, false))) return rc;
    if ((rc = exchange_sum(h, h->d.kvec_a, g.K))) return rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_B, false))) return rc;
    if ((rc = exchange_windows(h, h->sw_begin, h->sw_end))) return rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_EXPAND, false))) return rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_C, false))) return rc;
    if ((rc = exchange_sum(h, h->d.kvec_c, 3 * (size_t)g.K))) return rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_D, false))) return rc;
  }
  return 0;
}

int svils_comm_allgather_host(svils_handle *h, const void *send, void *recv, size_t bytes) {
  NOT_TILED(h, "svils_comm_allgather_host");
  if (!h || !send || !recv) return fail(SVILS_ERR_ARG, "svils_comm_allgather_host: null argument");
  if (!h->comm) {
    if (h->world != 1) return fail(SVILS_ERR_ARG, "svils_comm_allgather_host: call svils_comm_init first");
    memcpy(recv, send, bytes);
    return 0;
  }
  if (bytes == 0) return 0;
  HIPCHK(hipSetDevice(h->cfg.device));
  // the staging buffer persists and grows only when a larger payload comes (every rank passes the same `bytes`, so
  // they grow at the same call).  The ranks agree that everybody's allocation worked BEFORE the gather: a rank
  // that ran out of memory must not leave its peers blocked in the collective.
  if (bytes > h->stage_bytes) {
    if (h->stage) (void)hipFree(h->stage);
    h->stage = nullptr;
    h->stage_bytes = 0;
    if (!h->stage_flag) HIPCHK(hipMalloc(&h->stage_flag, sizeof(uint32_t)));
    const size_t want = bytes + bytes / 4;   // some head room: payloads of one run differ by little
    const uint32_t failed = hipMalloc(&h->stage, want * (size_t)h->world) == hipSuccess ? 0u : 1u;
    if (failed) { h->stage = nullptr; (void)hipGetLastError(); }
    HIPCHK(hipMemcpyAsync(h->stage_flag, &failed, sizeof failed, hipMemcpyHostToDevice, h->stream));
    NCCLCHK(g_rccl.AllReduce(h->stage_flag, h->stage_flag, 1, ncclUint32, ncclSum, h->comm, h->stream));
    uint32_t nfailed = 0;
    HIPCHK(hipMemcpyAsync(&nfailed, h->stage_flag, sizeof nfailed, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nfailed) {
      if (h->stage) (void)hipFree(h->stage);
      h->stage = nullptr;
      return fail(SVILS_ERR_DEVICE, "svils_comm_allgather_host: %u of %d ranks could not allocate %zu staging bytes", nfailed, h->world,
                  want * (size_t)h->world);
    }
    h->stage_bytes = want;
  }
  unsigned char *tmp = h->stage;
  HIPCHK(hipMemcpyAsync(tmp + (size_t)h->rank * bytes, send, bytes, hipMemcpyHostToDevice, h->stream));
  NCCLCHK(g_rccl.AllGather(tmp + (size_t)h->rank * bytes, tmp, bytes, ncclUint8, h->comm, h->stream));
  HIPCHK(hipMemcpyAsync(recv, tmp, bytes * (size_t)h->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int svils_gather_communities(svils_handle *h) {
  NOT_TILED(h, "svils_gather_communities");
  if (!h) return fail(SVILS_ERR_ARG, "svils_gather_communities: null handle");
  if (!h->comm) return h->world == 1 ? 0 : fail(SVILS_ERR_ARG, "svils_gather_communities: call svils_comm_init first");
  HIPCHK(hipSetDevice(h->cfg.device));
  const Geometry &g = h->geo;
  // every block's rows of the community bitmask, in place, with the exact counts (the blocks differ in size)
  NCCLCHK(g_rccl.GroupStart());
  for (int r = 0; r < h->world; ++r) {
    const size_t rows = h->blk.bounds[r + 1] - h->blk.bounds[r];
    if (!rows) continue;
    uint64_t *mp = h->d.member + (size_t)h->blk.bounds[r] * g.kw;
    NCCLCHK(g_rccl.Broadcast(mp, mp, rows * g.kw, ncclUint64, r, h->comm, h->stream));
  }
  NCCLCHK(g_rccl.GroupEnd());
  // The broadcasts WRITE state the getters read (the other blocks' rows of `member`): the no-wait window a seen stop
  // opened (svils_handle::frozen) is suspended until a getter has synchronised the stream behind them (settle()).
  h->writes_in_flight = true;
  return 0;
}

int svils_set_node_blocks(svils_handle *h, int rank, int world, const uint32_t *bounds) {
  NOT_TILED(h, "svils_set_node_blocks");
  if (!h) return fail(SVILS_ERR_ARG, "svils_set_node_blocks: null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  return apply_blocks(h, rank, world, bounds, bounds != nullptr);
}

int svils_balance_node_blocks(const uint32_t *links, uint64_t nlinks, uint32_t n, int world, double node_weight, uint32_t *bounds) {
  if ((!links && nlinks) || !bounds || n == 0 || world < 1 || world > SVILS_MAX_WORLD)
    return fail(SVILS_ERR_ARG, "svils_balance_node_blocks: bad argument (at most %d ranks)", SVILS_MAX_WORLD);
  if (node_weight < 0.0) node_weight = 0.5;
  std::vector<uint32_t> deg(n, 0);
  for (uint64_t l = 0; l < nlinks; ++l) {
    const uint32_t p = links[2 * l], q = links[2 * l + 1];
    if (p >= n || q >= n) return fail(SVILS_ERR_ARG, "svils_balance_node_blocks: link %llu names node %u / %u (n = %u)", (unsigned long long)l, p, q, n);
    deg[p]++;
    deg[q]++;
  }
  // cost of a node = its CSR entries (the phi pass evaluates each once) + node_weight (the per-node part of the finalise
  // pass, in units of one entry); cut r goes where the running cost is closest to r / world of the total
  const double total = 2.0 * (double)nlinks + node_weight * (double)n;
  bounds[0] = 0;
  double run = 0.0;
  uint32_t x = 0;
  for (int r = 1; r < world; ++r) {
    const double target = total * (double)r / (double)world;
    while (x < n) {
      const double c = (double)deg[x] + node_weight;
      if (run + c > target && (run + c - target) > (target - run)) break;   // taking x overshoots by more than stopping short
      run += c;
      ++x;
      if (run >= target) break;
    }
    bounds[r] = x;
  }
  bounds[world] = n;
  return 0;
}

}  // extern "C"
