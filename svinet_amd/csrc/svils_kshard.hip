// svils_kshard.hip -- multi-GPU, K-sharded (every rank a column slice of all rows; kernels in svils_ksh.h): the phases of a
// sweep / mini-batch step with the four all-reduces between them, and the likelihood row of a sharded state.
#include "svils_handle.h"

extern "C" {

// ---------------------------------------------------------------- K-sharded sweeps (svils_ksh.h)
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
  if (on && !h->d.ksh_log && h->sweeps_issued > 0) {
    // the product form keeps no Elogpi rows (k_fin1_ksh writes the exp(Elogpi) rows of the next sweep itself): bring them up
    // to date from gamma and the summed row sums of the last sweep before the log-domain kernels read them
    HIPCHK(hipSetDevice(h->cfg.device));
    launch_ksh_phase(h->geo, h->d, h->prm, 6, h->stream);
    HIPCHK(hipGetLastError());
  }
  h->d.ksh_log = on ? 1 : 0;
  return 0;
}

}  // extern "C"
namespace svils_impl {
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
}  // namespace svils_impl
extern "C" {

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

}  // extern "C"
namespace svils_impl {
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
}  // namespace svils_impl
