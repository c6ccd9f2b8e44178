// svils_report_api.hip -- host side of the pipelined reports (svils_report_*; kernels in svils_report.hip) and of the test set
// (svils_set_test / svils_get_test_rows).
#include "svils_handle.h"

extern "C" {

// ---------------------------------------------------------------- pipelined reports (include/svils.h)
}  // extern "C"
namespace svils_impl {
void ctrl_out(const DevCtrl &c, svils_control *out) {
  out->iter = c.iter; out->annealing = c.annealing; out->write_comm = c.write_comm; out->nh = c.nh;
  out->prev_h = c.prev_h; out->max_h = c.max_h; out->stopped = c.stopped; out->why = c.why;
  out->sweeps_done = c.sweeps_done; out->rows = c.rows;
  out->links_dense = c.links_dense; out->links_sparse = c.links_sparse; out->links_shortcut = c.links_shortcut;
}
}  // namespace svils_impl
namespace svils_impl {
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
}  // namespace svils_impl
extern "C" {

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
  const bool direct = rbytes <= ((size_t)1 << 20) && !h->opt.report_staged;
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
  if (int rc_ = settle(h)) return rc_;
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

}  // extern "C"
