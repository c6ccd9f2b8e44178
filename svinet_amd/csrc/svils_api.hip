// svils_api.hip -- the C ABI of include/svils.h on top of the gfx950 kernels.
// Host-side work here is plumbing only: argument checks, CSR construction,
// uploads/downloads, launch sequencing and hipEvent timing.  There is no CPU
// compute path: without a HIP device svils_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: librccl is dlopen()ed on first use

#include "svils_internal.h"
#include "svils_report.h"

using namespace svils;

namespace {

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

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(e_ == hipErrorOutOfMemory ? SVILS_ERR_NOMEM : SVILS_ERR_DEVICE,        \
                  "%s failed: %s", #expr, hipGetErrorString(e_));                        \
  } while (0)

// column-tiled handles (k > SVILS_MAX_K, svils_handle::tiles)
#define TILED(h) ((h) && !(h)->tiles.empty())
#define NOT_TILED(h, name) \
  do { if (TILED(h)) return fail(SVILS_ERR_UNSUPPORTED, name ": not available on a column-tiled handle (k > SVILS_MAX_K = %d)", SVILS_MAX_K); } while (0)

struct EvPair {
  hipEvent_t a, b;
};

}  // namespace

struct svils_handle {
  static constexpr uint32_t kGraphMaxLog = 6;
  svils_config cfg;
  Geometry geo;
  DeviceState d;
  Params prm;
  hipStream_t stream = nullptr;
  bool have_graph = false, have_state = false;
  // lane-per-link layout: the link classes on the device describe the sweep about to run (k_s3_lpl
  // refreshes them for the next sweep); cleared whenever flags / _iter / the window change under them
  bool cls_valid = false;
  // three-launch sweeps hand work between workgroups INSIDE a launch (classification role blocks of the s3 launch): only
  // where the device provably holds all of them at once -- decided when the graph is set (svils_set_graph)
  bool fused3_ok = true;
  bool shard_fold_ok = true;      // SVILS_SHARD_FOLD=0: node-block sweeps keep the k_colreduce launches (A/B knob)
  bool cflag_dirty = true;        // the host wrote converged flags (or nothing has yet): rebuild cflag[] before classifying
  bool derive_ok = true;          // SVILS_DERIVE_M=0 keeps the stored mean indicators everywhere (A/B knob)
  bool mphi_stale = false;        // whole sweeps (derive_m) left the stored mean indicators behind gamma: k_mphi_from_gamma on demand
  // The host has SEEN the stop (a fetched report or control block said `stopped`): every launch from the stopping sweep
  // on returns at once without touching the state, so the getters below read it without waiting for the no-op sweeps a
  // pipelined caller still has in flight behind the stop (two chunks of 16 sweeps in the drop-in binary: ~0.2 ms).
  bool frozen = false;
  // Column tiles (k > SVILS_MAX_K on ONE device): the handle the caller holds owns `tiles` K-sharded handles -- slices of
  // at most SVILS_MAX_K columns of every row, the layout of a K-sharded multi-GPU run with all its "ranks" on this device
  // and on one stream -- and drives their phases itself; the four exchanges of a K-sharded sweep become a sum over the
  // tiles' buffers (k_tiles_combine).  Nothing else of this struct is used by such a handle.
  std::vector<svils_handle *> tiles;
  bool stream_shared = false;     // a tile: its stream is tile 0's
  bool tiles_inited = false;      // the row sums / Elogpi of the tiles' state have been formed (needs graph and state)
  bool v_flush_needed = false;   // a three-launch sweep left its likelihood row / stop rule to the next launch
  bool v_flush_capture = false;  // ... and so do the sweeps captured in the hipGraphs
  void *cls_zero = nullptr;      // ltot + shist + scan descriptors, one contiguous block
  size_t cls_zero_bytes = 0;
  // native multi-GPU driver (svils_comm_init)
  ncclComm_t comm = nullptr;
  ncclComm_t comm_rows = nullptr;   // second communicator of the same ranks: the chunked row exchange on comm_stream
  int rank = 0, world = 1;
  // node-block sweeps: the row exchange runs on a stream of its own, in chunks, and the rows of a chunk are expanded
  // (k_expand) on the compute stream while the next chunk travels
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ready = nullptr;               // phase B done: the rows may leave
  std::vector<hipEvent_t> ev_chunk;            // chunk c has arrived
  uint32_t xchunks = 0;                        // 0: chosen from the payload (SVILS_XCHUNKS overrides)
  unsigned char *stage = nullptr;   // device staging of svils_comm_allgather_host: world x stage_bytes, grown collectively
  size_t stage_bytes = 0;
  uint32_t *stage_flag = nullptr;   // device word: "my allocation failed", summed over the ranks
  // node blocks of a node-block run (svils_set_node_blocks, or the equal blocks svils_comm_init assumes)
  Blocks blk{};
  bool blocks_set = false;
  bool blocks_explicit = false;     // bounds came from the caller (balanced): the s3 pass is split by link count, no mini-batch steps
  std::vector<uint32_t> h_upper;    // [n] offset of the first q > x inside row x (host copy, for the s3 split)
  // hipGraphs of node-block sweeps (with their collectives captured): [i] = 2^i sweeps
  hipGraphExec_t sgexec[kGraphMaxLog + 1] = {};
  bool sgraphs_ok = true;
  std::vector<uint32_t> timed_sweeps;   // sweeps_done index of every sweep whose phi launch was bracketed
  uint64_t sweeps_issued = 0;           // sweeps enqueued so far (== DevCtrl.sweeps_done unless stopped)
  // hipGraph replay of whole sweeps (host launch cost: 8 launches x ~7 us per sweep eager)
  static constexpr uint32_t kGraphSweeps = 8;
  hipGraphExec_t gexec1 = nullptr, gexecN = nullptr;   // 1 sweep / kGraphSweeps sweeps
  // other powers of two up to kGraphMax sweeps, captured on first use: 20 sweeps replay as 16 + 4, 100 as 64 + 32 + 4
  // (every graph launch is ~4.5 us of idle device: profiles/r03zb_graph_granularity.txt)
  hipGraphExec_t gexecP[kGraphMaxLog + 1] = {};        // [i]: 2^i sweeps (i = 0 and 3 stay null: gexec1, gexecN)
  bool graphs_ok = true;                               // false after a capture failure: stay eager
  uint32_t graph_after = 128;                          // sweeps a handle runs eagerly before it captures graphs (svils_sweep)
  std::vector<void *> allocs;
  // pipelined reports (svils_report_enqueue): staging slots, a copy stream, per-slot events
  struct ReportSlot {
    unsigned char *dev = nullptr, *host = nullptr;
    hipEvent_t packed = nullptr, landed = nullptr;
    bool busy = false, with_member = false;
    uint32_t row_first = 0, row_count = 0;
  };
  // -load-test (svils_set_test): a second pair set through the validation kernel, rows in a ring of their own
  uint32_t *t_pairs = nullptr;
  double *t_uval = nullptr, *t_rows = nullptr;
  uint32_t nt = 0, t_cap = 0;
  ReportSlot rslot[SVILS_REPORT_SLOTS];
  ReportLayout rlay{};
  hipStream_t copy_stream = nullptr;
  double *row_scratch = nullptr;  // device [10]
  // timing
  uint32_t tmask = 0;
  uint32_t tperiod = 1;   // bracket every tperiod-th sweep only
  std::vector<EvPair> pending[SVILS_KERNEL_COUNT];
  std::vector<EvPair> freelist;
  double t_ms[SVILS_KERNEL_COUNT] = {0};
  uint64_t t_n[SVILS_KERNEL_COUNT] = {0};
  std::vector<uint64_t> h_rowptr;  // kept for training_links / aux
  // mini-batch (Robbins-Monro) mode, svils_set_stochastic / svils_step
  bool stoch = false;
  svils_stochastic scfg{};
  uint64_t steps_done = 0;
  std::vector<uint64_t> h_linkptr;      // [n+1] first training link whose first endpoint is >= node
  std::vector<uint32_t> h_item_phi;     // [n+1] first phi item of a node (row-per-wavefront layout)
  std::vector<uint32_t> h_item_s3;      // [n+1] first s3 item of a node
  // the open mini-batch step (between svils_step_phase(A) and (D)): per-launch copies with the window set
  bool step_open = false;
  Geometry sg;
  DeviceState sd;
  Params sp;
  uint32_t sw_begin = 0, sw_end = 0;    // window relative to a rank's block
};

namespace {

template <class T>
int dalloc(svils_handle *h, T **p, size_t count, bool zero = true) {
  *p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T) + 512;   // slack: chunked row loads may run past the last row
  void *q = nullptr;
  HIPCHK(hipMalloc(&q, bytes));
  h->allocs.push_back(q);
  if (zero) HIPCHK(hipMemsetAsync(q, 0, bytes, h->stream));
  *p = (T *)q;
  return 0;
}

// give a dalloc()ed buffer back before svils_destroy (buffers that are re-sized by a later call)
template <class T>
void dfree(svils_handle *h, T **p) {
  if (!*p) return;
  auto it = std::find(h->allocs.begin(), h->allocs.end(), (void *)*p);
  if (it != h->allocs.end()) h->allocs.erase(it);
  (void)hipFree((void *)*p);
  *p = nullptr;
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

struct Timed {
  svils_handle *h;
  int k;
  EvPair ev{};
  bool on;
  Timed(svils_handle *h_, int k_) : h(h_), k(k_), on((h_->tmask >> k_) & 1u) {
    if (!on) return;
    if (h->freelist.empty()) {
      if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) { on = false; return; }
    } else {
      ev = h->freelist.back();
      h->freelist.pop_back();
    }
    (void)hipEventRecord(ev.a, h->stream);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(ev.b, h->stream);
    h->pending[k].push_back(ev);
  }
};

int fault_error(uint32_t code) {
  if (code == 2u)
    return fail(SVILS_ERR_DEVICE, "K-sharded sweep: the softmax denominator of a link underflowed (rows of disjoint support); "
                                  "switch the log-domain exchange on for this model: svils_ksh_log_domain(h, 1) (the default above K = 700)");
  return fail(SVILS_ERR_DEVICE, "an in-launch hand-off between workgroups timed out (role blocks not co-resident on this device); the state is frozen");
}

// ---- node blocks ------------------------------------------------------------------------------------------------
int apply_s3_split(svils_handle *h);
int ensure_classes(svils_handle *h);
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

// (re)classify the links of the sweep about to run from the flags as they stand
int classify_now(svils_handle *h, const Geometry &g, const DeviceState &d, const Params &prm) {
  Timed t(h, SVILS_KERNEL_CLASSIFY);
  if (h->cflag_dirty) {
    launch_cflag_rebuild(h->geo, h->d, h->stream);
    h->cflag_dirty = false;
  }
  HIPCHK(hipMemsetAsync(h->cls_zero, 0, h->cls_zero_bytes, h->stream));
  launch_classify(g, d, prm, h->stream);
  return 0;
}

// `fused`: the whole sweep is enqueued by this library with no exchange between the phases, so
// (small K) consumers fold the producers' partial rows themselves and k_s3_lpl classifies the
// links of the next sweep; otherwise the K-vectors are materialised for the caller's collectives.
int run_phase(svils_handle *h, svils_phase ph, const Geometry &g, const DeviceState &d0, const Params &prm,
              bool fused, bool shard = false) {
  hipStream_t s = h->stream;
  DeviceState d = d0;
  // whole full sweeps keep the mean indicators in derived form (svils_internal.h: derive_m); anything else -- sweeps
  // split at their exchange points, mini-batch steps -- works on the stored rows, brought up to date first
  // (lane-per-link layout, K <= 56: its s3 kernel runs at the register limit of 16-wave blocks and keeps the stored form)
  d.derive_m = (fused && !prm.stoch && !d.ksh && !d.lpl && h->derive_ok) ? 1 : 0;
  if (!d.derive_m && h->mphi_stale) {
    launch_mphi_from_gamma(h->geo, h->d, h->prm, s);
    h->mphi_stale = false;
  }
  if (d.derive_m && ph == SVILS_PHASE_B) h->mphi_stale = true;
  d.fold = (fused && d.lpl && g.K <= 32) ? 1 : 0;   // K = 33..64: K-vectors via k_colreduce (2K columns are too wide to fold)
  // Node-block sweeps issued by this library (svils_sweep_sharded), K <= 32: the K-vectors the collectives need are left by
  // the kernels themselves -- `sum` by the light finalise pass (block 0 adds the phi pass's per-XCD accumulators), s1 / s2 by
  // and s3 by the last block of the s3 launch to arrive -- instead of by two k_colreduce launches; the tail reads the
  // all-reduced vectors (no fold there).  The s3 launch then has at most 192 blocks (<= 192 partial rows for its last block).
  const bool shard_fold = shard && d.lpl && g.K <= 32 && !prm.stoch && h->shard_fold_ok;
  if (shard_fold && (ph == SVILS_PHASE_A || ph == SVILS_PHASE_B_LIGHT)) d.fold = 1;
  d.shard_c = (shard_fold && ph == SVILS_PHASE_C) ? 1 : 0;   // (d.fold stays 0 there: the LAST block leaves s1, s2 and s3)
  // Three launches per sweep when this library drives whole full sweeps at K <= 32: the work of k_tail is
  // split between the last s3 block (lambda, loop control) and a role of the NEXT phi launch (likelihood,
  // stop rule), and the phi pass accumulates beside gamma so that it may run before the stop rule has spoken.
  // ... on graphs of up to 512 classification tiles (half a million CSR entries): there the whole next-sweep
  // classification fits the <= 64 co-resident role blocks of the s3 launch with at most two tiles per worker.
  // Larger graphs keep four launches, where the two classification passes ride spin-free on the s3 and tail
  // launches with as many blocks as they need (n=1e6, K=20: s3 launch 1740 -> see profiles/r02h).
  // (a handle with a test set keeps four launches: the deferred stop rule would come too late for the test row)
  d.fused3 = (d.fold && !prm.stoch && d.gacc0 && d.cls_ntiles <= 512u && !h->nt && h->fused3_ok) ? 1 : 0;
  if (d.fused3) {
    d.gacc = d.gacc0;
    d.nvb = lpl_validation_blocks(g, d.nv, g.K);
  }
  d.cls_next = (d.lpl && !prm.stoch) ? 1 : 0;
  switch (ph) {
    case SVILS_PHASE_A: {
      if (d.lpl && (!h->cls_valid || prm.stoch)) {
        int rc = classify_now(h, g, d, prm);
        if (rc) return rc;
        h->cls_valid = true;
      }
      if ((h->tmask >> SVILS_KERNEL_PHI) & 1u) h->timed_sweeps.push_back((uint32_t)h->sweeps_issued);
      { Timed t(h, SVILS_KERNEL_PHI); launch_phi(g, d, prm, s); }
      if (!d.fold) { Timed t(h, SVILS_KERNEL_REDUCE_SUM); launch_reduce_a(g, d, s); }
    } break;
    case SVILS_PHASE_B: {
      Timed t(h, SVILS_KERNEL_FINALIZE);
      launch_finalize(g, d, prm, s);
      if (prm.stoch) launch_carry_flags(g, d, s);
    } break;
    case SVILS_PHASE_C: {
      { Timed t(h, SVILS_KERNEL_S3); launch_s3(g, d, prm, s); }
      if (!d.fold && !d.shard_c) { Timed t(h, SVILS_KERNEL_REDUCE_S); launch_reduce_c(g, d, s); }
    } break;
    case SVILS_PHASE_EXPAND: {
      launch_expand(g, d, prm, s);
    } break;
    case SVILS_PHASE_B_LIGHT: {
      if (prm.stoch) return fail(SVILS_ERR_ARG, "SVILS_PHASE_B_LIGHT belongs to whole sweeps, not to mini-batch steps");
      int rc = ensure_blocks(h);
      if (rc) return rc;
      d.gstage = h->d.gstage;
      d.gown = h->d.gown;
      d.light = 1;
      Timed t(h, SVILS_KERNEL_FINALIZE);
      launch_finalize(g, d, prm, s);
    } break;
    case SVILS_PHASE_EXPAND_ALL: {
      int rc = ensure_blocks(h);
      if (rc) return rc;
      d.gstage = h->d.gstage;
      Blocks b = h->blk;
      b.chunk = 0;
      b.nchunks = 1;
      launch_expand_all(g, d, prm, b, s);
    } break;
    case SVILS_PHASE_D: {
      if (!d.fused3) {
        Timed t(h, SVILS_KERNEL_TAIL);
        launch_tail(g, d, prm, s);
        if (h->nt) {   // test_likelihood (src/linksampling.cc:781): the validation kernel over the test pairs, then the row
          DeviceState dt = d;
          dt.vpairs = h->t_pairs; dt.uval = h->t_uval; dt.nv = h->nt;
          launch_validation(g, dt, prm, s);
          launch_test_row(dt, prm, h->t_rows, d.rows_cap, s);
        }
      } else {
        h->v_flush_needed = true;   // the sweep's likelihood row is owed by the next phi launch or by flush_validation()
      }
      if (d.lpl && !d.cls_next) h->cls_valid = false;
      ++h->sweeps_issued;
    } break;
    default:
      return fail(SVILS_ERR_ARG, "unknown phase %d", (int)ph);
  }
  HIPCHK(hipGetLastError());
  // keep the event pools bounded
  for (int i = 0; i < SVILS_KERNEL_COUNT; ++i)
    if (h->pending[i].size() > 8192) return drain_timing(h);
  return 0;
}

int run_phase(svils_handle *h, svils_phase ph, bool fused, bool shard = false) { return run_phase(h, ph, h->geo, h->d, h->prm, fused, shard); }

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

}  // namespace

namespace {
int tiles_create(const svils_config *cfg, svils_handle **out);
int tiles_try_init(svils_handle *h);
}  // namespace

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
  // allocation carries slack for the last row.  SVILS_PACK_ROWS=0 restores the padded stride (A/B).
  {
    const char *e = getenv("SVILS_PACK_ROWS");
    if ((!e || atoi(e)) && use_lpl(cfg->k) && !cfg->k_total) g.ld = (cfg->k + 1u) & ~1u;
  }
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
  if (const char *e = getenv("SVILS_DERIVE_M")) h->derive_ok = atoi(e) != 0;
  if (const char *e = getenv("SVILS_FAULT_INJECT")) h->d.inject_fault = strcmp(e, "cls_handoff") == 0 ? 1 : 0;   // tests only
  const bool whole_graph = g.node_begin == 0 && g.node_end == g.n;
  uint64_t epi_max_mb = (whole_graph && h->derive_ok) ? ~0ull >> 21 : 1536;
  if (const char *e = getenv("SVILS_EPI_MAX_MB")) epi_max_mb = strtoull(e, nullptr, 10);   // A/B knob (profiles/r03*)
  if (d.ksh || (!use_lpl(g.K) && nk * sizeof(double) <= epi_max_mb << 20)) guard(dalloc(h, &d.epi, nk));
  if (d.ksh) {
    guard(dalloc(h, &d.rowx, 3 * (size_t)g.n));
    guard(dalloc(h, &d.q2v, g.Kt));
  }
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
  if (const char *e = getenv("SVILS_GRAPH_AFTER")) h->graph_after = (uint32_t)std::max(0, atoi(e));
  if (const char *e = getenv("SVILS_SHARD_FOLD")) h->shard_fold_ok = atoi(e) != 0;
  *out = h;
  return 0;
}

// ---------------------------------------------------------------- RCCL, bound at run time
namespace {
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  // evidence only (svils_comm_query); an RCCL build without one of them still runs the sweeps
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
};
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

#define NCCLCHK(expr)                                                                          \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess) return fail(SVILS_ERR_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)

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
  if (getenv("SVILS_ONE_COMM") && atoi(getenv("SVILS_ONE_COMM")) != 0) {   // A/B knob: rows share the first communicator
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
}  // namespace

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
  if (const char *e = getenv("SVILS_XCHUNKS")) h->xchunks = (uint32_t)std::max(0, atoi(e));   // chunks of the pipelined row exchange
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

namespace {
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
      // (SVILS_ALLGATHER_ROWS / SVILS_EXACT_ROWS force one form: A/B on real links, and the tests' way to put the grouped
      //  {all-reduce, broadcasts} launch through the real librccl on a world of one)
      const bool padded = !getenv("SVILS_EXACT_ROWS") && ((uint64_t)b.bmax * b.world * 2 <= 3 * (uint64_t)g.n || getenv("SVILS_ALLGATHER_ROWS"));
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
}  // namespace

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
  // SVILS_SHARDED_GRAPHS=0 keeps every sweep eager.  Every rank takes the same decisions (same arguments, same
  // history), so the ranks enqueue the same collectives in the same order whether they replay or launch.
  // (read at every call, not once per process: bench.py times an eager window first and a replayed one after it, so
  // that a first contact with real multi-GPU RCCL that blocks under capture still leaves the eager number behind)
  const char *sg_env = getenv("SVILS_SHARDED_GRAPHS");
  const bool graphs_wanted = !(sg_env && atoi(sg_env) == 0);
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

namespace {
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
}  // namespace

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

// ---------------------------------------------------------------- K-sharded sweeps (svils_ksh.h)
namespace {
int open_step(svils_handle *h);
}
int svils_ksweep_phase(svils_handle *h, svils_kphase phase) {
  NOT_TILED(h, "svils_ksweep_phase");
  if (!h) return fail(SVILS_ERR_ARG, "svils_ksweep_phase: null handle");
  if (!h->d.ksh) return fail(SVILS_ERR_ARG, "svils_ksweep_phase: not a K-sharded handle (svils_config.k_total)");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_ksweep_phase: set graph and state first");
  if ((int)phase < 0 || (int)phase > 7) return fail(SVILS_ERR_ARG, "svils_ksweep_phase: unknown phase %d", (int)phase);
  if (phase == SVILS_KPHASE_DENMAX && !h->d.ksh_log) return fail(SVILS_ERR_ARG, "svils_ksweep_phase: DENMAX belongs to the log-domain mode (svils_ksh_log_domain)");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->stoch && ((int)phase <= 4 || phase == SVILS_KPHASE_DENMAX)) {
    // mini-batch step over the window of nodes every rank shares (open_step: window geometry, item ranges, the
    // factors that turn window sums into estimates, this step's step sizes).  The first phase of a step opens it
    // (DENMAX in the log-domain mode, else DEN), STOP closes it.
    const svils_kphase first = h->d.ksh_log ? SVILS_KPHASE_DENMAX : SVILS_KPHASE_DEN;
    if (phase == first) {
      if (h->step_open) return fail(SVILS_ERR_ARG, "svils_ksweep_phase: the previous step was not closed with phase STOP");
      int rc = open_step(h);
      if (rc) return rc;
    } else if (!h->step_open) {
      return fail(SVILS_ERR_ARG, "svils_ksweep_phase: a mini-batch step starts with phase %s", h->d.ksh_log ? "DENMAX" : "DEN");
    }
    launch_ksh_phase(h->sg, h->sd, h->sp, (int)phase, h->stream);
    HIPCHK(hipGetLastError());
    if (phase == SVILS_KPHASE_STOP) {
      h->step_open = false;
      ++h->steps_done;
      ++h->sweeps_issued;
    }
    return 0;
  }
  launch_ksh_phase(h->geo, h->d, h->prm, (int)phase, h->stream);
  HIPCHK(hipGetLastError());
  if (phase == SVILS_KPHASE_STOP) ++h->sweeps_issued;
  return 0;
}

int svils_ksh_buffer_ptr(svils_handle *h, svils_ksh_buffer which, void **dptr, size_t *ndoubles) {
  NOT_TILED(h, "svils_ksh_buffer_ptr");
  if (!h || !dptr || !ndoubles) return fail(SVILS_ERR_ARG, "svils_ksh_buffer_ptr: null argument");
  if (!h->d.ksh) return fail(SVILS_ERR_ARG, "svils_ksh_buffer_ptr: not a K-sharded handle");
  const DeviceState &d = h->d;
  if (h->stoch && h->step_open) {
    // a mini-batch step: what crosses the ranks is the window's share -- the CSR entries of its rows (one contiguous
    // range of the entry-indexed per-link buffers) and its rows of rowx
    const size_t e0 = (size_t)h->sd.ent_begin, ne = (size_t)(h->sd.ent_end - h->sd.ent_begin);
    const size_t r0 = h->sg.node_begin, nr = h->sg.node_end - h->sg.node_begin;
    switch (which) {
      case SVILS_KSH_DEN: *dptr = d.den + e0; *ndoubles = ne; return 0;
      case SVILS_KSH_DMAX: *dptr = d.dmax + e0; *ndoubles = ne; return 0;
      case SVILS_KSH_EARG: *dptr = d.ksh_lowt ? d.earg + e0 : nullptr; *ndoubles = d.ksh_lowt ? ne : 0; return 0;
      case SVILS_KSH_ROWX: *dptr = d.rowx + 3 * r0; *ndoubles = 3 * nr; return 0;
      default: break;
    }
  }
  switch (which) {
    case SVILS_KSH_DEN: *dptr = d.den; *ndoubles = (size_t)d.nlinks; return 0;
    case SVILS_KSH_ROWX: *dptr = d.rowx; *ndoubles = 3 * (size_t)h->geo.n; return 0;
    case SVILS_KSH_Q2: *dptr = d.q2v; *ndoubles = h->geo.Kt; return 0;
    case SVILS_KSH_VDOT: *dptr = d.vdot; *ndoubles = d.nv; return 0;
    case SVILS_KSH_DMAX: *dptr = d.dmax; *ndoubles = (size_t)d.nlinks; return 0;
    case SVILS_KSH_EARG: *dptr = d.earg; *ndoubles = d.ksh_lowt ? (size_t)d.nlinks : 0; return 0;
  }
  return fail(SVILS_ERR_ARG, "svils_ksh_buffer_ptr: unknown buffer %d", (int)which);
}

int svils_ksh_log_domain(svils_handle *h, int on) {
  NOT_TILED(h, "svils_ksh_log_domain");
  if (!h) return fail(SVILS_ERR_ARG, "svils_ksh_log_domain: null handle");
  if (!h->d.ksh) return fail(SVILS_ERR_ARG, "svils_ksh_log_domain: not a K-sharded handle");
  if (on < 0) return h->d.ksh_log;   // query
  if (!on && h->d.ksh_lowt) return fail(SVILS_ERR_ARG, "svils_ksh_log_domain: link_thresh < 1/2 needs the log-domain exchange (it carries the link's maximum)");
  h->d.ksh_log = on ? 1 : 0;
  return 0;
}

namespace {
int ksh_sum(svils_handle *h, svils_ksh_buffer which) {
  if (!h->comm) return 0;
  void *p = nullptr;
  size_t n = 0;
  int rc = svils_ksh_buffer_ptr(h, which, &p, &n);
  if (rc || n == 0) return rc;
  Timed t(h, SVILS_KERNEL_EXCHANGE);
  NCCLCHK(g_rccl.AllReduce(p, p, n, ncclDouble, which == SVILS_KSH_DMAX ? ncclMax : which == SVILS_KSH_EARG ? ncclMin : ncclSum, h->comm, h->stream));
  return 0;
}
}  // namespace

int svils_ksh_init_state(svils_handle *h) {
  NOT_TILED(h, "svils_ksh_init_state");
  int rc;
  if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_INIT_ROWS))) return rc;
  if ((rc = ksh_sum(h, SVILS_KSH_ROWX))) return rc;
  return svils_ksweep_phase(h, SVILS_KPHASE_INIT_EXPAND);
}

int svils_sweep_ksharded(svils_handle *h, uint32_t nsweeps) {
  NOT_TILED(h, "svils_sweep_ksharded");
  if (!h) return fail(SVILS_ERR_ARG, "svils_sweep_ksharded: null handle");
  if (!h->d.ksh) return fail(SVILS_ERR_ARG, "svils_sweep_ksharded: not a K-sharded handle");
  if (!h->comm && h->geo.K != h->geo.Kt) return fail(SVILS_ERR_ARG, "svils_sweep_ksharded: call svils_comm_init first");
  if (nsweeps > (uint64_t)h->d.rows_cap * h->prm.reportfreq)
    return fail(SVILS_ERR_ARG, "svils_sweep_ksharded: at most %llu sweeps per call",
                (unsigned long long)h->d.rows_cap * h->prm.reportfreq);
  for (uint32_t i = 0; i < nsweeps; ++i) {
    int rc;
    if (h->d.ksh_log) {
      if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_DENMAX))) return rc;
      if ((rc = ksh_sum(h, SVILS_KSH_DMAX))) return rc;   // MAX
    }
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_DEN))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_DEN))) return rc;
    if (h->d.ksh_lowt && (rc = ksh_sum(h, SVILS_KSH_EARG))) return rc;   // MIN
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_PHI))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_ROWX))) return rc;
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_FIN))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_Q2))) return rc;
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_LAMBDA))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_VDOT))) return rc;
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_STOP))) return rc;
  }
  return 0;
}

// Mini-batch (Robbins-Monro) steps on the K-sharded layout: every rank steps through the SAME window of nodes on its own
// column slice; the exchanges are those of a sweep, restricted to the window's share of the buffers.
int svils_step_ksharded(svils_handle *h, uint32_t nsteps) {
  NOT_TILED(h, "svils_step_ksharded");
  if (!h) return fail(SVILS_ERR_ARG, "svils_step_ksharded: null handle");
  if (!h->d.ksh) return fail(SVILS_ERR_ARG, "svils_step_ksharded: not a K-sharded handle");
  if (!h->stoch) return fail(SVILS_ERR_ARG, "svils_step_ksharded: call svils_set_stochastic first");
  if (!h->comm && h->geo.K != h->geo.Kt) return fail(SVILS_ERR_ARG, "svils_step_ksharded: call svils_comm_init first");
  if (nsteps > (uint64_t)h->d.rows_cap * h->prm.reportfreq)
    return fail(SVILS_ERR_ARG, "svils_step_ksharded: at most %llu steps per call",
                (unsigned long long)h->d.rows_cap * h->prm.reportfreq);
  for (uint32_t i = 0; i < nsteps; ++i) {
    int rc;
    if (h->d.ksh_log) {
      if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_DENMAX))) return rc;
      if ((rc = ksh_sum(h, SVILS_KSH_DMAX))) return rc;   // MAX
    }
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_DEN))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_DEN))) return rc;
    if (h->d.ksh_lowt && (rc = ksh_sum(h, SVILS_KSH_EARG))) return rc;   // MIN
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_PHI))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_ROWX))) return rc;
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_FIN))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_Q2))) return rc;
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_LAMBDA))) return rc;
    if ((rc = ksh_sum(h, SVILS_KSH_VDOT))) return rc;
    if ((rc = svils_ksweep_phase(h, SVILS_KPHASE_STOP))) return rc;
  }
  return 0;
}

namespace {
// validation_likelihood (src/linksampling.cc:966-1002) of a K-sharded state between two sweeps: the partial dot
// products of the own columns, summed over the ranks, then the log terms on the host in pair order
// (the order of the reference's map walk).  rowx[3p] holds the full row sum of gamma[p] after
// svils_ksh_init_state and after every sweep.  Collective.
int ksh_validation_row_finish(svils_handle *h, double *row10);
int ksh_validation_row(svils_handle *h, double *row10) {
  if (!h->have_graph) return fail(SVILS_ERR_ARG, "svils_validation_row: a K-sharded handle needs its graph and svils_ksh_init_state first");
  if (!h->comm && h->geo.K != h->geo.Kt) return fail(SVILS_ERR_ARG, "svils_validation_row: call svils_comm_init first");
  launch_ksh_phase(h->geo, h->d, h->prm, 8, h->stream);   // k_vdot_ksh alone
  HIPCHK(hipGetLastError());
  int rc = ksh_sum(h, SVILS_KSH_VDOT);
  if (rc) return rc;
  return ksh_validation_row_finish(h, row10);
}
// the log terms of the summed dot products, on the host in pair order
int ksh_validation_row_finish(svils_handle *h, double *row10) {
  const DeviceState &d = h->d;
  std::vector<double> vdot(d.nv);
  std::vector<uint32_t> vp(3 * (size_t)d.nv);
  DevCtrl c;
  HIPCHK(hipMemcpyAsync(vdot.data(), d.vdot, vdot.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(vp.data(), d.vpairs, vp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&c, d.ctrl, sizeof c, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  double sz = 0.0, so = 0.0;
  uint32_t kz = 0, ko = 0;
  for (uint32_t i = 0; i < d.nv; ++i) {
    const uint32_t y = vp[3 * (size_t)i + 2];
    const double pq = vdot[i];   // (k_vdot_ksh works on the normalised rows)
    double sv = y ? pq : 1.0 - pq;
    if (sv < 1e-30) sv = 1e-30;
    if (y) { so += log(sv); ko++; } else { sz += log(sv); kz++; }
  }
  const double mean0 = sz / kz, mean1 = so / ko;
  row10[0] = (double)c.iter; row10[1] = (sz + so) / d.nv; row10[2] = (double)d.nv;
  row10[3] = mean0; row10[4] = (double)kz; row10[5] = mean1; row10[6] = (double)ko;
  row10[7] = h->prm.zeros_prob * mean0; row10[8] = h->prm.ones_prob * mean1; row10[9] = row10[7] + row10[8];
  return 0;
}
}  // namespace

// ---------------------------------------------------------------- column tiles: k > SVILS_MAX_K on one device
namespace {
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
}  // namespace

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
  d.nb_c = cap((d.nitems_s3 + 3) / 4, 2 * rpw_resident_blocks(g, 1, h->cfg.device));
  // lane-per-link layout for small K: wave-items of 64 consecutive entries of a class list
  // The class lists pack an entry index into 27 bits: graphs of 2^26 training links or more take the
  // row-per-wavefront kernels at small K too (they index with 64 bits and have no such limit; slower per link at
  // K <= 56, but the reference's main use case -- small K on a large graph -- must not be refused).
  // SVILS_LPL_MAX_ENTRIES lowers the switch-over point (tests exercise the fallback on small graphs with it).
  uint64_t lpl_max_entries = 1ull << 27;
  if (const char *e = getenv("SVILS_LPL_MAX_ENTRIES")) lpl_max_entries = std::min<uint64_t>(lpl_max_entries, strtoull(e, nullptr, 10));
  d.lpl = (use_lpl(g.K) && !d.ksh && 2 * nlinks < lpl_max_entries) ? 1 : 0;
  {
    const size_t state_bytes = (size_t)g.n_alloc * g.ld * sizeof(double);
    d.wt = (d.lpl && state_bytes >= ((size_t)1 << 20) && state_bytes <= ((size_t)8 << 20)) ? 1 : 0;    // svils_internal.h: DeviceState::wt
    if (const char *e = getenv("SVILS_WT")) d.wt = (d.lpl && atoi(e) != 0) ? 1 : 0;                    // A/B knob
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
      // CU): both keep the four-launch sweep, whose passes never wait for another workgroup.  SVILS_FUSED3=0 / 1 overrides.
      const uint32_t res = lpl_s3_resident_blocks(g.K, h->cfg.device);
      const bool masked = getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK");
      h->fused3_ok = res >= 2u * 65u && !masked;
      if (const char *e = getenv("SVILS_FUSED3")) h->fused3_ok = atoi(e) != 0;
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
  h->mphi_stale = false;   // the stored rows have nothing to do with the new gamma (nor has the reference's _mphi after load_model)
  HIPCHK(hipSetDevice(h->cfg.device));
  const Geometry &g = h->geo;
  DeviceState &d = h->d;
  HIPCHK(hipMemsetAsync(d.gamma, 0, (size_t)g.n_alloc * g.ld * sizeof(double), h->stream));
  HIPCHK(hipMemcpy2DAsync(d.gamma, g.ld * sizeof(double), gamma, g.K * sizeof(double),
                          g.K * sizeof(double), g.n, hipMemcpyHostToDevice, h->stream));
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

int svils_get_control(svils_handle *h, svils_control *out) {
  if (TILED(h)) return svils_get_control(h->tiles[0], out);   // the loop control is replicated on the tiles
  if (!h || !out) return fail(SVILS_ERR_ARG, "svils_get_control: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->frozen) HIPCHK(hipStreamSynchronize(h->stream));
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

int svils_sweep_phase(svils_handle *h, svils_phase phase) {
  NOT_TILED(h, "svils_sweep_phase");
  if (!h) return fail(SVILS_ERR_ARG, "svils_sweep_phase: null handle");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_sweep_phase: set graph and state first");
  if (h->d.ksh) return fail(SVILS_ERR_ARG, "svils_sweep_phase: a K-sharded handle is driven by svils_ksweep_phase");
  HIPCHK(hipSetDevice(h->cfg.device));
  return run_phase(h, phase, false);
}

namespace {

int eager_sweeps(svils_handle *h, uint32_t nsweeps) {
  for (uint32_t i = 0; i < nsweeps; ++i) {
    int rc;
    if ((rc = run_phase(h, SVILS_PHASE_A, true))) return rc;
    if ((rc = run_phase(h, SVILS_PHASE_B, true))) return rc;
    if ((rc = run_phase(h, SVILS_PHASE_C, true))) return rc;
    if ((rc = run_phase(h, SVILS_PHASE_D, true))) return rc;
  }
  return 0;
}

void drop_graphs(svils_handle *h) {
  if (h->gexec1) { (void)hipGraphExecDestroy(h->gexec1); h->gexec1 = nullptr; }
  if (h->gexecN) { (void)hipGraphExecDestroy(h->gexecN); h->gexecN = nullptr; }
  for (auto &g_ : h->gexecP) if (g_) { (void)hipGraphExecDestroy(g_); g_ = nullptr; }
}

// capture `nsweeps` sweeps of the library's own stream into an executable graph; every kernel
// argument is a by-value snapshot of pointers/sizes that stay fixed after set_graph/set_state
// (all loop state lives in device memory), so the graph can be replayed indefinitely
hipGraphExec_t capture_sweeps(svils_handle *h, uint32_t nsweeps) {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  const uint64_t issued = h->sweeps_issued;
  const bool vf = h->v_flush_needed;
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return nullptr;
  const int rc = eager_sweeps(h, nsweeps);
  const hipError_t e = hipStreamEndCapture(h->stream, &graph);
  h->sweeps_issued = issued;   // nothing ran
  h->v_flush_capture = h->v_flush_needed;
  h->v_flush_needed = vf;
  if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); return nullptr; }
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
  (void)hipGraphDestroy(graph);
  // the first launch of an executable graph otherwise pays for its upload (measured in the drop-in binary's trace: 130 - 250 us
  // in front of the first chunk of every size): done here, where svils_prepare_graphs has the caller still in its set-up
  if (exec && hipGraphUpload(exec, h->stream) != hipSuccess) (void)hipGetLastError();
  return exec;
}

// three-launch sweeps: the held-out likelihood and stop rule of the last sweep enqueued, as a launch of
// its own (inside a run of sweeps they ride on the next phi launch)
int flush_validation(svils_handle *h) {
  if (!h->v_flush_needed) return 0;
  DeviceState d = h->d;
  d.nvb = lpl_validation_blocks(h->geo, d.nv, h->geo.K);
  {
    Timed t(h, SVILS_KERNEL_TAIL);
    launch_validate_lpl(h->geo, d, h->prm, h->stream);
  }
  HIPCHK(hipGetLastError());
  h->v_flush_needed = false;
  return 0;
}

// the captured sweeps assume valid link classes on entry (each sweep leaves them valid for the next)
int ensure_classes(svils_handle *h) {
  if (!h->d.lpl || h->cls_valid) return 0;
  DeviceState d = h->d;
  int rc = classify_now(h, h->geo, d, h->prm);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  h->cls_valid = true;
  return 0;
}

}  // namespace

namespace {

// replay `n` sweeps from the untimed graphs (captured on first use, with event recording off)
int graph_sweeps(svils_handle *h, uint32_t n) {
  int rc = ensure_classes(h);
  if (rc) return rc;
  if (!h->gexec1) {
    const uint32_t saved = h->tmask;
    h->tmask = 0;
    h->gexec1 = capture_sweeps(h, 1);
    h->gexecN = h->gexec1 ? capture_sweeps(h, svils_handle::kGraphSweeps) : nullptr;
    h->tmask = saved;
    if (!h->gexec1 || !h->gexecN) { drop_graphs(h); h->graphs_ok = false; return eager_sweeps(h, n); }
  }
  h->sweeps_issued += n;
  if (n) h->v_flush_needed = h->v_flush_capture;   // what a captured sweep leaves behind
  // ... and what run_phase's bookkeeping would have noted had the sweeps been launched eagerly: whole sweeps in derived
  // form leave the stored mean indicators behind gamma (same condition as d.derive_m there)
  if (n && !h->prm.stoch && !h->d.ksh && !h->d.lpl && h->derive_ok) h->mphi_stale = true;
  // as few replays as possible: powers of two from 2^kGraphMaxLog down (SVILS_GRAPH_POW2=0: 8-sweep graphs + singles)
  static const bool pow2 = !(getenv("SVILS_GRAPH_POW2") && atoi(getenv("SVILS_GRAPH_POW2")) == 0);
  if (pow2) {
    for (int i = (int)svils_handle::kGraphMaxLog; i >= 1; --i) {
      const uint32_t m = 1u << i;
      if (n < m) continue;
      hipGraphExec_t *ge = (m == svils_handle::kGraphSweeps) ? &h->gexecN : &h->gexecP[i];
      if (!*ge) {
        const uint32_t saved = h->tmask;
        h->tmask = 0;
        *ge = capture_sweeps(h, m);
        h->tmask = saved;
        if (!*ge) continue;               // (smaller graphs carry the sweeps)
      }
      for (; n >= m; n -= m) HIPCHK(hipGraphLaunch(*ge, h->stream));
    }
  }
  for (; n >= svils_handle::kGraphSweeps; n -= svils_handle::kGraphSweeps) HIPCHK(hipGraphLaunch(h->gexecN, h->stream));
  for (; n > 0; --n) HIPCHK(hipGraphLaunch(h->gexec1, h->stream));
  return 0;
}

}  // namespace

int svils_sweep(svils_handle *h, uint32_t nsweeps) {
  if (TILED(h)) return tiles_sweep(h, nsweeps);
  if (!h) return fail(SVILS_ERR_ARG, "svils_sweep: null handle");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_sweep: set graph and state first");
  if (h->stoch) return fail(SVILS_ERR_ARG, "svils_sweep: the handle is in mini-batch mode, use svils_step");
  if (h->d.ksh) return fail(SVILS_ERR_ARG, "svils_sweep: a K-sharded handle is driven by svils_ksweep_phase / svils_sweep_ksharded");
  HIPCHK(hipSetDevice(h->cfg.device));
  // likelihood rows go to a ring of rows_cap entries: never enqueue more reports than it holds
  // between two host polls (svils_get_rows)
  const uint64_t max_batch = (uint64_t)h->d.rows_cap * h->prm.reportfreq;
  if (nsweeps > max_batch)
    return fail(SVILS_ERR_ARG, "svils_sweep: at most %llu sweeps per call (likelihood-row ring of %u entries)",
                (unsigned long long)max_batch, h->d.rows_cap);
  int rc = 0;
  // Capturing and instantiating the sweep graphs costs milliseconds (three to five graphs of up to 64 sweeps x 3-4
  // nodes): more than a whole short run -- ca-AstroPh K = 20 with the default flags stops after 31 sweeps, ~2 ms of
  // device time.  Graph replay only removes host launch cost, so it starts paying once a run is long: calls stay
  // eager until the handle has seen graph_after sweeps (128; SVILS_GRAPH_AFTER, read when the handle is created,
  // overrides; 0 = capture at the first call of >= 4 sweeps), unless a single call is itself long.  Results are identical either way (one code path per kernel).
  const bool warm = h->gexec1 != nullptr || h->sweeps_issued + nsweeps >= h->graph_after || nsweeps >= 64;
  // (short calls are not worth a capture -- but once the single-sweep graph exists, svils_prepare_graphs, they replay it:
  // an eager three-launch sweep leaves ~20 us of gaps, a graph launch ~4.5)
  if (!h->graphs_ok || !warm || (nsweeps < 4 && !(h->gexec1 && h->gexecN && h->tmask == 0))) rc = eager_sweeps(h, nsweeps);
  else if (h->tmask == 0) rc = graph_sweeps(h, nsweeps);
  // Per-kernel hipEvent timing needs eager launches: events captured as graph nodes cannot be read
  // with hipEventElapsedTime on this runtime.  With a sampling period P > 1 only every P-th sweep is
  // launched eagerly between events; the P-1 sweeps in between replay the untimed graphs.
  else if (h->tperiod <= 1) rc = eager_sweeps(h, nsweeps);
  else {
    uint32_t left = nsweeps;
    while (left > 0 && !rc) {
      rc = eager_sweeps(h, 1);
      --left;
      const uint32_t n = std::min(left, h->tperiod - 1);
      if (n && !rc) rc = graph_sweeps(h, n);
      left -= n;
    }
  }
  if (rc) return rc;
  return flush_validation(h);
}

// Capture the hipGraphs svils_sweep replays -- 1, 4, 8, 16 ... sweeps up to max_sweeps -- NOW, while the caller is still in
// its set-up, instead of in the middle of the run once the handle has seen 128 sweeps.  A short run (the default ca-AstroPh
// run stops after 31 sweeps) then replays graphs from its first chunk of >= 4 sweeps on: eager launches cost the device
// ~20 us of gaps per three-launch sweep.  The graphs do not depend on the state, only on the buffers: call it after
// svils_set_graph / svils_set_validation / svils_set_state.  Nothing runs except the stand-alone link classification.
int svils_prepare_graphs(svils_handle *h, uint32_t max_sweeps) {
  if (TILED(h)) return 0;   // column tiles launch eagerly (tens of launches of milliseconds each per sweep)
  if (!h) return fail(SVILS_ERR_ARG, "svils_prepare_graphs: null handle");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_prepare_graphs: set graph and state first");
  if (h->stoch || h->d.ksh || !h->graphs_ok) return 0;
  HIPCHK(hipSetDevice(h->cfg.device));
  int rc = ensure_classes(h);
  if (rc) return rc;
  const uint32_t saved = h->tmask;
  h->tmask = 0;
  if (!h->gexec1) h->gexec1 = capture_sweeps(h, 1);
  if (h->gexec1 && !h->gexecN && max_sweeps >= svils_handle::kGraphSweeps) h->gexecN = capture_sweeps(h, svils_handle::kGraphSweeps);
  if (!h->gexecN) h->gexecN = h->gexec1 ? capture_sweeps(h, svils_handle::kGraphSweeps) : nullptr;   // graph_sweeps expects both
  for (int i = 2; i <= (int)svils_handle::kGraphMaxLog && h->gexec1; ++i) {
    const uint32_t m = 1u << i;
    if (m > max_sweeps || m == svils_handle::kGraphSweeps || h->gexecP[i]) continue;
    h->gexecP[i] = capture_sweeps(h, m);
  }
  h->tmask = saved;
  if (!h->gexec1 || !h->gexecN) { drop_graphs(h); h->graphs_ok = false; }
  return 0;
}

int svils_set_timing_period(svils_handle *h, uint32_t period) {
  NOT_TILED(h, "svils_set_timing_period");
  if (!h || period == 0) return fail(SVILS_ERR_ARG, "svils_set_timing_period: bad argument");
  h->tperiod = period;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Mini-batch (Robbins-Monro) steps.  One step = the sweep's four phases restricted to a window of
// consecutive nodes [b, e): phi pass over the window's CSR rows, finalise of the window's rows
// blended into the old gamma with the node's own step size, s3 over the links whose first endpoint
// lies in the window, then lambda blended with rho_lambda, likelihood row and stop rule as in a
// full sweep.  Window sums are scaled to estimates of the full sums (Params::scale_a/scale_c); with
// the window = all nodes and kappa = 0 (rho = 1) a step IS a full sweep.
// ---------------------------------------------------------------------------------------------
void svils_stochastic_default(svils_stochastic *cfg, uint32_t batch_nodes) {
  if (!cfg) return;
  cfg->batch_nodes = batch_nodes;
  cfg->node_tau0 = 1024; cfg->node_kappa = 0.5;   // src/env.hh:405-408
  cfg->tau0 = 1024; cfg->kappa = 0.9;
  cfg->seed = 0;
  cfg->shard_block = 0;
}

int svils_set_stochastic(svils_handle *h, const svils_stochastic *cfg) {
  NOT_TILED(h, "svils_set_stochastic");
  if (!h || !cfg) return fail(SVILS_ERR_ARG, "svils_set_stochastic: null argument");
  if (h->d.ksh && cfg->shard_block) return fail(SVILS_ERR_ARG, "svils_set_stochastic: a K-sharded handle holds every node (shard_block must be 0)");
  if (!(cfg->tau0 >= 1.0) || !(cfg->kappa >= 0.0) || cfg->kappa > 1.0 || !(cfg->node_tau0 >= 1.0) ||
      !(cfg->node_kappa >= 0.0) || cfg->node_kappa > 1.0)
    return fail(SVILS_ERR_ARG, "svils_set_stochastic: need tau0 >= 1 and 0 <= kappa <= 1");
  const Geometry &g0 = h->geo;
  const bool whole = g0.node_begin == 0 && g0.node_end == g0.n;
  if (cfg->shard_block == 0) {
    if (!whole) return fail(SVILS_ERR_ARG, "svils_set_stochastic: a node-block shard needs shard_block");
  } else {
    if (g0.node_begin % cfg->shard_block != 0 || g0.node_end > g0.node_begin + cfg->shard_block ||
        g0.n_alloc % cfg->shard_block != 0)
      return fail(SVILS_ERR_ARG, "svils_set_stochastic: shard_block does not match the handle's node block");
  }
  // the running totals s1/s2 of the mini-batch mode start from mphi == 0: full sweeps first would
  // leave rows it knows nothing about
  if (!h->stoch && h->sweeps_issued > 0)
    return fail(SVILS_ERR_ARG, "svils_set_stochastic: enable the mini-batch mode before the first sweep");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  DeviceState &d = h->d;
  h->cls_valid = false;
  if (!h->stoch) {
    int rc = 0;
    double *gacc = nullptr;
    if ((rc = dalloc(h, &gacc, (size_t)h->geo.n_alloc * h->geo.ld))) return rc;
    if ((rc = dalloc(h, &d.ncnt, h->geo.n_alloc))) return rc;
    if ((rc = dalloc(h, &d.s12run, 2 * (size_t)h->geo.K))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    d.gacc = gacc;
    drop_graphs_of(h);   // captured launches hold gacc == gamma
  }
  h->stoch = true;
  h->scfg = *cfg;
  if (d.ksh) d.ksh_ent = 1;   // per-link exchange buffers by CSR entry from now on (svils_ksh.h)
  return 0;
}

namespace {

// window of step `t` relative to a rank's block, and the per-launch state of this handle for it
void step_window(const svils_handle *h, uint64_t t, uint32_t *b, uint32_t *e) {
  const uint32_t B = h->scfg.shard_block ? h->scfg.shard_block : h->geo.n;
  const uint32_t bn = (h->scfg.batch_nodes == 0 || h->scfg.batch_nodes > B) ? B : h->scfg.batch_nodes;
  const uint32_t nblocks = (B + bn - 1) / bn;
  const uint32_t blk = (uint32_t)((t + h->scfg.seed) % nblocks);   // fixed cyclic order (profiles/HISTORY.md section 6a)
  *b = blk * bn;
  *e = std::min(B, *b + bn);
}

int open_step(svils_handle *h) {
  const uint32_t n = h->geo.n;
  const uint32_t B = h->scfg.shard_block ? h->scfg.shard_block : n;
  const uint32_t world = h->scfg.shard_block ? h->geo.n_alloc / B : 1;
  uint32_t wb, we;
  step_window(h, h->steps_done, &wb, &we);
  h->sw_begin = wb;
  h->sw_end = we;
  Geometry &g = h->sg;
  DeviceState &d = h->sd;
  Params &p = h->sp;
  g = h->geo;
  d = h->d;
  p = h->prm;
  // this handle's rows of the mini-batch
  const uint32_t b = std::min(n, h->geo.node_begin + wb), e = std::min(n, h->geo.node_begin + we);
  g.node_begin = b;
  g.node_end = e;
  d.ent_begin = h->h_rowptr[b];
  d.ent_end = h->h_rowptr[e];
  if (d.lpl) {   // classification tiles covering the window's entries (slot capacity stays the handle's)
    d.cls_tile0 = (uint32_t)(d.ent_begin / d.cls_tile);
    d.cls_ntiles = d.ent_end > d.ent_begin
                       ? (uint32_t)((d.ent_end + d.cls_tile - 1) / d.cls_tile) - d.cls_tile0 : 0u;
  }
  d.link_begin = h->h_linkptr[b];
  d.link_end = h->h_linkptr[e];
  d.item0_phi = h->h_item_phi[b];
  d.nitems_phi = h->h_item_phi[e] - h->h_item_phi[b];
  d.item0_s3 = h->h_item_s3[b];
  d.nitems_s3 = h->h_item_s3[e] - h->h_item_s3[b];
  // grids sized for the window (never larger than the allocation made for full sweeps)
  {
    auto fit = [](uint64_t want, uint32_t lim) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, lim)); };
    const int G = 64 / g.W;
    d.nb_b = fit(((uint64_t)(e - b) + 4 * G - 1) / (4 * G), h->d.nb_b);
    if (d.lpl) {
      const int nw = lpl_phi_waves(g.K);
      const uint64_t items = ((d.ent_end - d.ent_begin + 63) >> 6) + 1;
      d.nb_a = fit((items + nw - 1) / nw, h->d.nb_a);
      const uint32_t fnodes = d.fin_waves * (64u / (uint32_t)lpl_finalize_group(g.K));
      d.nb_b = fit(((uint64_t)(e - b) + fnodes - 1) / fnodes, h->d.nb_b);
      d.nb_c = fit((d.link_end - d.link_begin + d.s3_threads - 1) / d.s3_threads, h->d.nb_c);
    } else {
      d.nb_a = fit(((uint64_t)d.nitems_phi + 3) / 4, h->d.nb_a);
      d.nb_c = fit(((uint64_t)d.nitems_s3 + 3) / 4, h->d.nb_c);
    }
  }
  p.stoch = 1;
  p.tau0 = h->scfg.node_tau0;
  p.kappa = h->scfg.node_kappa;
  p.rho_lambda = std::pow(h->scfg.tau0 + (double)h->steps_done, -h->scfg.kappa);
  // window sums -> estimates of the full sums: the mini-batch is the union of every rank's window
  uint64_t ents = 0, ups = 0;
  for (uint32_t r = 0; r < world; ++r) {
    const uint32_t rb = std::min(n, r * B + wb), re = std::min(n, std::min((r + 1) * B, r * B + we));
    ents += h->h_rowptr[re] - h->h_rowptr[rb];
    ups += h->h_linkptr[re] - h->h_linkptr[rb];
  }
  p.scale_a = ents ? (double)(2 * h->d.nlinks) / (double)ents : 0.0;
  p.scale_c = ups ? (double)h->d.nlinks / (double)ups : 0.0;
  h->step_open = true;
  return 0;
}

}  // namespace

int svils_step_window(svils_handle *h, uint32_t *begin, uint32_t *end) {
  NOT_TILED(h, "svils_step_window");
  if (!h || !begin || !end) return fail(SVILS_ERR_ARG, "svils_step_window: null argument");
  if (!h->stoch) return fail(SVILS_ERR_ARG, "svils_step_window: call svils_set_stochastic first");
  if (h->step_open) { *begin = h->sw_begin; *end = h->sw_end; }
  else step_window(h, h->steps_done, begin, end);
  return 0;
}

namespace {
int step_phase_impl(svils_handle *h, svils_phase phase, bool fused);
}
int svils_step_phase(svils_handle *h, svils_phase phase) { return step_phase_impl(h, phase, false); }

namespace {
int step_phase_impl(svils_handle *h, svils_phase phase, bool fused) {
  if (!h) return fail(SVILS_ERR_ARG, "svils_step_phase: null handle");
  NOT_TILED(h, "svils_step_phase");
  if (!h->stoch) return fail(SVILS_ERR_ARG, "svils_step_phase: call svils_set_stochastic first");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_step_phase: set graph and state first");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (phase == SVILS_PHASE_A) {
    if (h->step_open) return fail(SVILS_ERR_ARG, "svils_step_phase: the previous step was not closed with phase D");
    int rc = open_step(h);
    if (rc) return rc;
  } else if (!h->step_open) {
    return fail(SVILS_ERR_ARG, "svils_step_phase: phase A opens a step");
  }
  int rc;
  if (phase == SVILS_PHASE_EXPAND) {
    if (h->scfg.shard_block) {
      launch_expand_window(h->sg, h->sd, h->sp, h->sw_begin, h->sw_end, h->scfg.shard_block,
                           h->geo.node_begin / h->scfg.shard_block, h->geo.n_alloc / h->scfg.shard_block, h->stream);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  if ((rc = run_phase(h, phase, h->sg, h->sd, h->sp, fused))) return rc;
  if (phase == SVILS_PHASE_D) {
    h->step_open = false;
    ++h->steps_done;
  }
  return 0;
}
}  // namespace

int svils_step(svils_handle *h, uint32_t nsteps) {
  NOT_TILED(h, "svils_step");
  if (!h) return fail(SVILS_ERR_ARG, "svils_step: null handle");
  if (!h->stoch) return fail(SVILS_ERR_ARG, "svils_step: call svils_set_stochastic first");
  if (h->scfg.shard_block) return fail(SVILS_ERR_ARG, "svils_step: a node-block shard is driven with svils_step_phase");
  if (nsteps > (uint64_t)h->d.rows_cap * h->prm.reportfreq)
    return fail(SVILS_ERR_ARG, "svils_step: at most %llu steps per call (likelihood-row ring of %u entries)",
                (unsigned long long)h->d.rows_cap * h->prm.reportfreq, h->d.rows_cap);
  for (uint32_t s = 0; s < nsteps; ++s) {
    int rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_A, true))) return rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_B, true))) return rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_C, true))) return rc;
    if ((rc = step_phase_impl(h, SVILS_PHASE_D, true))) return rc;
  }
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
  if (!h->frozen) HIPCHK(hipStreamSynchronize(h->stream));
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

// ---------------------------------------------------------------- pipelined reports (include/svils.h)
namespace {
void ctrl_out(const DevCtrl &c, svils_control *out) {
  out->iter = c.iter; out->annealing = c.annealing; out->write_comm = c.write_comm; out->nh = c.nh;
  out->prev_h = c.prev_h; out->max_h = c.max_h; out->stopped = c.stopped; out->why = c.why;
  out->sweeps_done = c.sweeps_done; out->rows = c.rows;
  out->links_dense = c.links_dense; out->links_sparse = c.links_sparse; out->links_shortcut = c.links_shortcut;
}
}  // namespace

namespace {
// wait for a report's event: polled for a while (the caller is a host thread that has nothing else to do and the report is
// usually microseconds away; a blocking wait costs a wake-up of tens of microseconds), then the blocking form
hipError_t wait_landed(hipEvent_t ev) {
  for (int i = 0; i < 20000; ++i) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
  }
  return hipEventSynchronize(ev);
}
}  // namespace

int svils_report_enqueue(svils_handle *h, uint32_t row_first, uint32_t row_count, int with_communities, int *ticket) {
  NOT_TILED(h, "svils_report_enqueue");
  if (!h || !ticket) return fail(SVILS_ERR_ARG, "svils_report_enqueue: null argument");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_report_enqueue: set graph and state first");
  const Geometry &g = h->geo;
  if (h->d.ksh || g.node_begin != 0 || g.node_end != g.n)
    return fail(SVILS_ERR_ARG, "svils_report_enqueue: whole-graph handles only (a sharded run gathers its tags collectively)");
  if (row_count > SVILS_REPORT_MAX_ROWS) return fail(SVILS_ERR_ARG, "svils_report_enqueue: at most %d rows per report", SVILS_REPORT_MAX_ROWS);
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t nwords = (size_t)g.n * g.kw;
  if (!h->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    h->rlay.off_rows = 128;   // the control block in front (sizeof(DevCtrl) <= 128)
    static_assert(sizeof(DevCtrl) <= 128, "report layout");
    h->rlay.off_trows = h->rlay.off_rows + (size_t)SVILS_REPORT_MAX_ROWS * 10 * sizeof(double);
    h->rlay.off_member = h->rlay.off_trows + (size_t)SVILS_REPORT_MAX_ROWS * 10 * sizeof(double);
    h->rlay.bytes = h->rlay.off_member + nwords * sizeof(uint64_t);
  }
  int t = -1;
  for (int i = 0; i < SVILS_REPORT_SLOTS; ++i)
    if (!h->rslot[i].busy) { t = i; break; }
  if (t < 0) return fail(SVILS_ERR_ARG, "svils_report_enqueue: %d reports outstanding, fetch one first", SVILS_REPORT_SLOTS);
  svils_handle::ReportSlot &rs = h->rslot[t];
  if (!rs.dev) {
    HIPCHK(hipMalloc((void **)&rs.dev, h->rlay.bytes));
    HIPCHK(hipHostMalloc((void **)&rs.host, h->rlay.bytes, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&rs.packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&rs.landed, hipEventDisableTiming));
  }
  // A small snapshot (up to 1 MB: ca-AstroPh's is 143 KB) is packed straight into the pinned host slot -- the pack launch's
  // stores cross PCIe themselves and the report has landed when that launch has: no second stream, no event hand-over, no
  // SDMA start-up (together ~200 us per report in the drop-in binary's trace, which is what its short default run is made
  // of).  Large ones (config 5: a 64 MB bitmask) keep the device staging + copy stream: the sweeps go on while the copy runs.
  const size_t rbytes = with_communities ? h->rlay.bytes : h->rlay.off_member;
  const bool direct = rbytes <= ((size_t)1 << 20) && !getenv("SVILS_REPORT_STAGED");
  launch_report_pack(h->d.ctrl, sizeof(DevCtrl), h->d.rows, h->nt ? h->t_rows : nullptr, h->d.rows_cap, row_first, row_count,
                     with_communities ? h->d.member : nullptr, with_communities ? nwords : 0, direct ? rs.host : rs.dev, h->rlay, h->stream);
  HIPCHK(hipGetLastError());
  if (direct) {
    HIPCHK(hipEventRecord(rs.landed, h->stream));
  } else {
    HIPCHK(hipEventRecord(rs.packed, h->stream));
    HIPCHK(hipStreamWaitEvent(h->copy_stream, rs.packed, 0));
    HIPCHK(hipMemcpyAsync(rs.host, rs.dev, rbytes, hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(hipEventRecord(rs.landed, h->copy_stream));
  }
  rs.busy = true;
  rs.with_member = with_communities != 0;
  rs.row_first = row_first;
  rs.row_count = row_count;
  *ticket = t;
  return 0;
}

int svils_report_ready(svils_handle *h, int ticket) {
  NOT_TILED(h, "svils_report_ready");
  if (!h || ticket < 0 || ticket >= SVILS_REPORT_SLOTS || !h->rslot[ticket].busy) return fail(SVILS_ERR_ARG, "svils_report_ready: bad ticket");
  const hipError_t e = hipEventQuery(h->rslot[ticket].landed);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return fail(SVILS_ERR_DEVICE, "svils_report_ready: %s", hipGetErrorString(e));
}

int svils_report_test_rows(svils_handle *h, int ticket, double *test_rows, uint32_t *ntest) {
  NOT_TILED(h, "svils_report_test_rows");
  if (!h || ticket < 0 || ticket >= SVILS_REPORT_SLOTS || !h->rslot[ticket].busy) return fail(SVILS_ERR_ARG, "svils_report_test_rows: bad ticket");
  if (!h->nt) return fail(SVILS_ERR_ARG, "svils_report_test_rows: the handle has no test set (svils_set_test)");
  svils_handle::ReportSlot &rs = h->rslot[ticket];
  HIPCHK(wait_landed(rs.landed));
  DevCtrl c;
  memcpy(&c, rs.host, sizeof c);
  if (c.fault) return fault_error(c.fault);
  uint32_t have = c.rows > rs.row_first ? std::min(c.rows - rs.row_first, rs.row_count) : 0u;
  // the stopping sweep recorded its validation row and left before test_likelihood: that row, the last one, has no partner
  if (c.stopped && have && rs.row_first + have == c.rows) --have;
  if (ntest) *ntest = have;
  if (test_rows && have) memcpy(test_rows, rs.host + h->rlay.off_trows, (size_t)have * 10 * sizeof(double));
  return 0;
}

int svils_set_test(svils_handle *h, const uint32_t *pairs_y, uint64_t nt) {
  NOT_TILED(h, "svils_set_test");
  if (!h || (!pairs_y && nt)) return fail(SVILS_ERR_ARG, "svils_set_test: null argument");
  if (h->d.ksh) return fail(SVILS_ERR_UNSUPPORTED, "svils_set_test: not for K-sharded handles");
  if (nt > 0xffffffffull) return fail(SVILS_ERR_ARG, "svils_set_test: too many pairs");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (uint64_t i = 0; i < nt; ++i)
    if (pairs_y[3 * i] >= h->geo.n || pairs_y[3 * i + 1] >= h->geo.n || pairs_y[3 * i] == pairs_y[3 * i + 1])
      return fail(SVILS_ERR_ARG, "svils_set_test: pair %llu names node %u / %u (n = %u)", (unsigned long long)i, pairs_y[3 * i], pairs_y[3 * i + 1], h->geo.n);
  drop_graphs_of(h);            // the captured sweeps do not know about the test launches (or still carry them)
  h->nt = 0;
  if (!nt) return 0;
  int rc;
  if (nt > h->t_cap) {          // a larger set than any before: the old buffers go back (the stream is idle here)
    dfree(h, &h->t_pairs);
    dfree(h, &h->t_uval);
    h->t_cap = 0;
    if ((rc = dalloc(h, &h->t_pairs, 3 * (size_t)nt))) return rc;
    if ((rc = dalloc(h, &h->t_uval, (size_t)nt))) return rc;
    h->t_cap = (uint32_t)nt;
  }
  if (!h->t_rows) {
    if ((rc = dalloc(h, &h->t_rows, (size_t)h->d.rows_cap * 10, false))) return rc;
    // a report without a test row reads as NaN
    HIPCHK(hipMemsetAsync(h->t_rows, 0xff, (size_t)h->d.rows_cap * 10 * sizeof(double), h->stream));
  }
  HIPCHK(hipMemcpyAsync(h->t_pairs, pairs_y, 3 * (size_t)nt * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->nt = (uint32_t)nt;
  return 0;
}

int svils_get_test_rows(svils_handle *h, uint32_t first, uint32_t count, double *rows) {
  NOT_TILED(h, "svils_get_test_rows");
  if (!h || (!rows && count)) return fail(SVILS_ERR_ARG, "svils_get_test_rows: null argument");
  if (!h->nt) return fail(SVILS_ERR_ARG, "svils_get_test_rows: the handle has no test set (svils_set_test)");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->frozen) HIPCHK(hipStreamSynchronize(h->stream));
  DevCtrl c;
  HIPCHK(hipMemcpy(&c, h->d.ctrl, sizeof c, hipMemcpyDeviceToHost));
  if (c.fault) return fault_error(c.fault);
  if ((uint64_t)first + count > c.rows) return fail(SVILS_ERR_ARG, "test rows [%u,%u) not recorded yet (have %u)", first, first + count, c.rows);
  if (c.rows - first > h->d.rows_cap) return fail(SVILS_ERR_ARG, "test row %u already overwritten in the ring", first);
  uint32_t done = 0;
  while (done < count) {
    const uint32_t slot = (first + done) % h->d.rows_cap;
    const uint32_t run = std::min(count - done, h->d.rows_cap - slot);
    HIPCHK(hipMemcpy(rows + (size_t)done * 10, h->t_rows + (size_t)slot * 10, (size_t)run * 10 * sizeof(double), hipMemcpyDeviceToHost));
    done += run;
  }
  return 0;
}

int svils_report_fetch(svils_handle *h, int ticket, svils_control *ctrl, double *rows, uint32_t *nrows, uint8_t *member) {
  NOT_TILED(h, "svils_report_fetch");
  if (!h || ticket < 0 || ticket >= SVILS_REPORT_SLOTS || !h->rslot[ticket].busy) return fail(SVILS_ERR_ARG, "svils_report_fetch: bad ticket");
  svils_handle::ReportSlot &rs = h->rslot[ticket];
  if (member && !rs.with_member) return fail(SVILS_ERR_ARG, "svils_report_fetch: this report was enqueued without communities");
  HIPCHK(wait_landed(rs.landed));
  rs.busy = false;
  DevCtrl c;
  memcpy(&c, rs.host, sizeof c);
  if (c.fault) return fault_error(c.fault);
  if (c.stopped) h->frozen = true;   // (svils_handle::frozen: the getters need not wait for the no-op sweeps behind the stop)
  if (ctrl) ctrl_out(c, ctrl);
  const uint32_t have = c.rows > rs.row_first ? std::min(c.rows - rs.row_first, rs.row_count) : 0u;
  if (nrows) *nrows = have;
  if (rows && have) memcpy(rows, rs.host + h->rlay.off_rows, (size_t)have * 10 * sizeof(double));
  if (member) {
    const Geometry &g = h->geo;
    const uint64_t *bits = (const uint64_t *)(rs.host + h->rlay.off_member);
    memset(member, 0, (size_t)g.n * g.K);
    for (uint32_t p = 0; p < g.n; ++p)
      for (int v = 0; v < g.V; ++v) {
        uint64_t b = bits[(size_t)p * g.kw + v];
        while (b) {
          const int lw = __builtin_ctzll(b);
          b &= b - 1;
          const uint32_t k = kmap_host(g.W, g.V, lw, v);
          if (k < g.K) member[(size_t)p * g.K + k] = 1;
        }
      }
  }
  return 0;
}

namespace {
// (node, community) pairs of a lane-layout community bitmask [n][kw]; counts them all, writes at most `cap`
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
}  // namespace

int svils_report_tag_count(svils_handle *h, int ticket, uint64_t *ntags) {
  NOT_TILED(h, "svils_report_tag_count");
  if (!h || !ntags || ticket < 0 || ticket >= SVILS_REPORT_SLOTS || !h->rslot[ticket].busy) return fail(SVILS_ERR_ARG, "svils_report_tag_count: bad ticket");
  svils_handle::ReportSlot &rs = h->rslot[ticket];
  if (!rs.with_member) return fail(SVILS_ERR_ARG, "svils_report_tag_count: this report was enqueued without communities");
  HIPCHK(wait_landed(rs.landed));
  *ntags = tags_of_bits(h->geo, (const uint64_t *)(rs.host + h->rlay.off_member), nullptr, 0);
  return 0;
}

int svils_report_fetch_tags(svils_handle *h, int ticket, svils_control *ctrl, double *rows, uint32_t *nrows,
                            uint32_t *tags, uint64_t cap, uint64_t *ntags) {
  NOT_TILED(h, "svils_report_fetch_tags");
  if (!h || !ntags || (!tags && cap)) return fail(SVILS_ERR_ARG, "svils_report_fetch_tags: null argument");
  if (ticket < 0 || ticket >= SVILS_REPORT_SLOTS || !h->rslot[ticket].busy) return fail(SVILS_ERR_ARG, "svils_report_fetch_tags: bad ticket");
  if (!h->rslot[ticket].with_member) return fail(SVILS_ERR_ARG, "svils_report_fetch_tags: this report was enqueued without communities");
  HIPCHK(wait_landed(h->rslot[ticket].landed));
  const uint64_t cnt = tags_of_bits(h->geo, (const uint64_t *)(h->rslot[ticket].host + h->rlay.off_member), tags, cap);
  *ntags = cnt;
  if (cnt > cap) return fail(SVILS_ERR_ARG, "svils_report_fetch_tags: %llu tags, room for %llu (svils_report_tag_count says how many); the slot is kept",
                             (unsigned long long)cnt, (unsigned long long)cap);
  return svils_report_fetch(h, ticket, ctrl, rows, nrows, nullptr);
}

int svils_get_community_tags(svils_handle *h, uint32_t *tags, uint64_t cap, uint64_t *ntags) {
  if (TILED(h)) return ntags && (tags || !cap) ? tiles_get_community_tags(h, tags, cap, ntags) : fail(SVILS_ERR_ARG, "svils_get_community_tags: null argument");
  if (!h || !ntags || (!tags && cap)) return fail(SVILS_ERR_ARG, "svils_get_community_tags: null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->frozen) HIPCHK(hipStreamSynchronize(h->stream));
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
  if (!h->frozen) HIPCHK(hipStreamSynchronize(h->stream));
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
  if (!h->frozen) HIPCHK(hipStreamSynchronize(h->stream));
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
    case 0:
      HIPCHK(hipMemcpy2D(out, g.K * sizeof(double), h->d.elogpi, g.ld * sizeof(double), g.K * sizeof(double), g.n, hipMemcpyDeviceToHost));
      return 0;
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
    case SVILS_BUF_ELOGPI: *dptr = d.elogpi; *row_bytes = g.ld * sizeof(double); *bytes = *row_bytes * g.n_alloc; return 0;
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
