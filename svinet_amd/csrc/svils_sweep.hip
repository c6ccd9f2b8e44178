// svils_sweep.hip -- one sweep of LinkSampling::infer() (src/linksampling.cc:571-789) as a sequence of launches: the phases
// (run_phase), whole sweeps enqueued eagerly or replayed from captured hipGraphs (svils_sweep, svils_prepare_graphs) and the
// caller-driven form (svils_sweep_phase).
#include "svils_handle.h"

namespace svils_impl {

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
              bool fused, bool shard) {
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

int run_phase(svils_handle *h, svils_phase ph, bool fused, bool shard) { return run_phase(h, ph, h->geo, h->d, h->prm, fused, shard); }

}  // namespace svils_impl

extern "C" {

int svils_sweep_phase(svils_handle *h, svils_phase phase) {
  NOT_TILED(h, "svils_sweep_phase");
  if (!h) return fail(SVILS_ERR_ARG, "svils_sweep_phase: null handle");
  if (!h->have_graph || !h->have_state) return fail(SVILS_ERR_ARG, "svils_sweep_phase: set graph and state first");
  if (h->d.ksh) return fail(SVILS_ERR_ARG, "svils_sweep_phase: a K-sharded handle is driven by svils_ksweep_phase");
  HIPCHK(hipSetDevice(h->cfg.device));
  return run_phase(h, phase, false);
}

}  // extern "C"
namespace svils_impl {

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

}  // namespace svils_impl
namespace svils_impl {

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
  // as few replays as possible: powers of two from 2^kGraphMaxLog down (option graph_pow2 = 0: 8-sweep graphs + singles)
  if (h->opt.graph_pow2) {
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

}  // namespace svils_impl
extern "C" {

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

}  // extern "C"
