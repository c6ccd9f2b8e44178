// svils_handle.h -- what the translation units of the C ABI share: the handle itself, error plumbing, device allocation,
// the hipEvent brackets, the run-time binding of RCCL, and the entry points one unit calls in another.
//
//   svils_api.hip         create / destroy, graph + state upload, getters, timing           (the handle's life cycle)
//   svils_sweep.hip       phases of a sweep, hipGraph capture and replay, svils_sweep       (src/linksampling.cc:571-789)
//   svils_comm.hip        RCCL binding, node blocks, node-block sweeps and steps, gathers   
//   svils_kshard.hip      K-sharded sweeps and steps over a communicator
//   svils_tiles.hip       column-tiled handles (k > SVILS_MAX_K on one device)
//   svils_stoch.hip       mini-batch (Robbins-Monro) steps
//   svils_report_api.hip  pipelined reports, test set
//   svils_init.hip        init_gamma2 on the device (svils_init_gamma: MT19937 streams, per-link draws, rows in link order)
//   svils_options.hip     the option table (svils_set_option, SVILS_* environment defaults)
//
// Host-side work in all of them is plumbing only: argument checks, CSR construction, uploads/downloads, launch
// sequencing and hipEvent timing.  There is no CPU compute path: without a HIP device svils_create() fails.
#pragma once
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
#include "svils_options.h"
#include "svils_report.h"

using namespace svils;

namespace svils_impl {

int fail(int code, const char *fmt, ...);   // sets svils_last_error(), returns code

struct EvPair {
  hipEvent_t a, b;
};
}  // namespace svils_impl

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return svils_impl::fail(e_ == hipErrorOutOfMemory ? SVILS_ERR_NOMEM : SVILS_ERR_DEVICE, \
                              "%s failed: %s", #expr, hipGetErrorString(e_));            \
  } while (0)

// column-tiled handles (k > SVILS_MAX_K, svils_handle::tiles)
#define TILED(h) ((h) && !(h)->tiles.empty())
#define NOT_TILED(h, name) \
  do { if (TILED(h)) return svils_impl::fail(SVILS_ERR_UNSUPPORTED, name ": not available on a column-tiled handle (k > SVILS_MAX_K = %d)", SVILS_MAX_K); } while (0)

using namespace svils_impl;

struct svils_handle {
  static constexpr uint32_t kGraphMaxLog = 6;
  svils_config cfg;
  Geometry geo;
  DeviceState d;
  Params prm;
  Options opt;                    // svils_options.h: the environment defaults as svils_create found them, then svils_set_option
  hipStream_t stream = nullptr;
  bool have_graph = false, have_state = false;
  // lane-per-link layout: the link classes on the device describe the sweep about to run (k_s3_lpl
  // refreshes them for the next sweep); cleared whenever flags / _iter / the window change under them
  bool cls_valid = false;
  // three-launch sweeps hand work between workgroups INSIDE a launch (classification role blocks of the s3 launch): only
  // where the device provably holds all of them at once -- decided when the graph is set (svils_set_graph)
  bool fused3_ok = true;
  bool shard_fold_ok = true;      // option shard_fold = 0: node-block sweeps keep the k_colreduce launches (A/B knob)
  bool cflag_dirty = true;        // the host wrote converged flags (or nothing has yet): rebuild cflag[] before classifying
  bool derive_ok = true;          // option derive_m = 0 keeps the stored mean indicators everywhere (A/B knob)
  bool mphi_stale = false;        // whole sweeps (derive_m) left the stored mean indicators behind gamma: k_mphi_from_gamma on demand
  // The host has SEEN the stop (a fetched report or control block said `stopped`): every launch from the stopping sweep
  // on returns at once without touching the state, so the getters below read it without waiting for the no-op sweeps a
  // pipelined caller still has in flight behind the stop (two chunks of 16 sweeps in the drop-in binary: ~0.2 ms).
  bool frozen = false;
  // ... unless something enqueued since then WRITES state the getters read (svils_gather_communities: the grouped broadcasts
  // of the other blocks' community rows): set by whoever enqueues such work, cleared by the getter that has waited for it
  bool writes_in_flight = false;
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
  uint32_t xchunks = 0;                        // 0: chosen from the payload (option xchunks overrides)
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
  double *elogpi_view = nullptr;   // DeviceState::skip_elogpi: where svils_get_aux(0) / SVILS_BUF_ELOGPI get their Elogpi rows computed
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

namespace svils_impl {

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

// The getters' wait: for the handle's stream -- unless the host has SEEN the stop (svils_handle::frozen: what is still in flight
// are launches that return at once) and nothing enqueued since writes state (svils_handle::writes_in_flight).
inline int settle(svils_handle *h) {
  if (h->frozen && !h->writes_in_flight) return 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  h->writes_in_flight = false;
  return 0;
}

// ---- svils_api.hip
int drain_timing(svils_handle *h);
int fault_error(uint32_t code);
void drop_graphs_of(svils_handle *h);
int elogpi_rows(svils_handle *h, double **rows);
int state_arrived(svils_handle *h, const double *lambda, const uint32_t *converged);   // the tail of svils_set_state   // the Elogpi rows of the state as it stands (computed now where a handle does not store them)
void chunk_row(std::vector<Item> &items, uint32_t p, uint32_t off, uint32_t len, uint32_t ch,
               int32_t *next_slot, int32_t *first_slot, uint32_t *nsplit);
// (node, community) pairs of a lane-layout community bitmask [n][kw]; counts them all, writes at most `cap`
uint64_t tags_of_bits(const Geometry &g, const uint64_t *bits, uint32_t *tags, uint64_t cap);
// ---- svils_sweep.hip
int classify_now(svils_handle *h, const Geometry &g, const DeviceState &d, const Params &prm);
int run_phase(svils_handle *h, svils_phase ph, const Geometry &g, const DeviceState &d0, const Params &prm,
              bool fused, bool shard = false);
int run_phase(svils_handle *h, svils_phase ph, bool fused, bool shard = false);
int ensure_classes(svils_handle *h);
int eager_sweeps(svils_handle *h, uint32_t nsweeps);
int flush_validation(svils_handle *h);
// ---- svils_comm.hip
int apply_s3_split(svils_handle *h);
int apply_blocks(svils_handle *h, int rank, int world, const uint32_t *bounds, bool explicit_bounds);
int ensure_blocks(svils_handle *h);
void comm_destroy(svils_handle *h);
// ---- svils_stoch.hip
int open_step(svils_handle *h);
int step_phase_impl(svils_handle *h, svils_phase phase, bool fused);
// ---- svils_kshard.hip
int ksh_validation_row(svils_handle *h, double *row10);
int ksh_validation_row_finish(svils_handle *h, double *row10);
// ---- svils_tiles.hip
int tiles_create(const svils_config *cfg, svils_handle **out);
int tiles_try_init(svils_handle *h);
int tiles_set_state(svils_handle *h, const double *gamma, const double *lambda, const uint32_t *converged);
int tiles_get_state(svils_handle *h, double *gamma, double *lambda, uint32_t *converged);
int tiles_sweep(svils_handle *h, uint32_t nsweeps);
int tiles_validation_row(svils_handle *h, double *row10);
int tiles_get_communities(svils_handle *h, uint8_t *member);
int tiles_get_community_tags(svils_handle *h, uint32_t *tags, uint64_t cap, uint64_t *ntags);

// ---------------------------------------------------------------- RCCL, bound at run time (svils_comm.hip: rccl_load)
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
extern Rccl g_rccl;
int rccl_load();

#define NCCLCHK(expr)                                                                          \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess) return svils_impl::fail(SVILS_ERR_DEVICE, "%s failed: %s", #expr, svils_impl::g_rccl.GetErrorString(r_)); \
  } while (0)

}  // namespace svils_impl
