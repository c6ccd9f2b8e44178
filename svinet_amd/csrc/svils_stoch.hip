// svils_stoch.hip -- mini-batch (Robbins-Monro) steps: svils_set_stochastic, the window of a step, svils_step / svils_step_phase.
#include "svils_handle.h"

extern "C" {

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

}  // extern "C"
namespace svils_impl {

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

}  // namespace svils_impl
extern "C" {

int svils_step_window(svils_handle *h, uint32_t *begin, uint32_t *end) {
  NOT_TILED(h, "svils_step_window");
  if (!h || !begin || !end) return fail(SVILS_ERR_ARG, "svils_step_window: null argument");
  if (!h->stoch) return fail(SVILS_ERR_ARG, "svils_step_window: call svils_set_stochastic first");
  if (h->step_open) { *begin = h->sw_begin; *end = h->sw_end; }
  else step_window(h, h->steps_done, begin, end);
  return 0;
}

int svils_step_phase(svils_handle *h, svils_phase phase) { return step_phase_impl(h, phase, false); }

}  // extern "C"
namespace svils_impl {
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
}  // namespace svils_impl
extern "C" {

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

}  // extern "C"
