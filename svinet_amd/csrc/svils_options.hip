// svils_options.hip -- the option table (svils_options.h; include/svils.h: svils_set_option / svils_get_option /
// svils_option_table).  One row per tunable: key, the environment variable that sets its default for handles created
// afterwards, the built-in default, until when it may still be changed on a handle, and what it does.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "svils_handle.h"

namespace svils_impl {

namespace {
enum Until { ANY = 0, GRAPH = 1, CREATE = 2 };   // may be set: at any time / before svils_set_graph / only through the environment
enum Kind { I32, U32, I64, U64 };
struct Row {
  const char *key, *env, *dflt;
  Until until;
  Kind kind;
  size_t off;
  long long lo, hi;
  const char *doc;
};
#define OFF(f) offsetof(Options, f)
const Row kRows[] = {
    {"pack_rows", "SVILS_PACK_ROWS", "1", CREATE, I32, OFF(pack_rows), 0, 1,
     "K <= 56: rows packed at ld = round_up(K, 2) (160-byte rows at K = 20) instead of a 128-byte-aligned stride"},
    {"derive_m", "SVILS_DERIVE_M", "1", CREATE, I32, OFF(derive_m), 0, 1,
     "K > 56, whole sweeps: mean indicators derived from gamma in the s3 pass instead of stored by the finalise pass"},
    {"epi_max_mb", "SVILS_EPI_MAX_MB", "-1", CREATE, I64, OFF(epi_max_mb), -1, 1ll << 40,
     "largest n-by-k array (MB) for which exp(Elogpi) rows are kept for the product form of the phi pass; -1: always on "
     "whole-graph handles, 1536 on node-block handles"},
    {"skip_elogpi", "SVILS_SKIP_ELOGPI", "-1", CREATE, I32, OFF(skip_elogpi), -1, 1,
     "57 <= K <= 512, link_thresh >= 1/2: the finalise / expand passes store no Elogpi rows (nothing in a sweep reads them; the phi pass's "
     "underflow fall-back becomes a second launch that returns at once unless a link underflowed): -1 where the n-by-k state is >= 256 MB, 0 / 1 forced"},
    {"graph_after", "SVILS_GRAPH_AFTER", "128", ANY, U32, OFF(graph_after), 0, 0xffffffffll,
     "sweeps a handle runs eagerly before svils_sweep / svils_sweep_sharded capture hipGraphs (0: at the first call of >= 4 sweeps)"},
    {"shard_fold", "SVILS_SHARD_FOLD", "1", ANY, I32, OFF(shard_fold), 0, 1,
     "node-block sweeps at K <= 32: the kernels leave sum / s1 / s2 / s3 themselves; 0 keeps the k_colreduce launches"},
    {"graph_pow2", "SVILS_GRAPH_POW2", "1", ANY, I32, OFF(graph_pow2), 0, 1,
     "replay runs of sweeps as powers of two up to 64 per graph; 0: 8-sweep graphs and singles"},
    {"lpl_max_entries", "SVILS_LPL_MAX_ENTRIES", "134217728", GRAPH, U64, OFF(lpl_max_entries), 0, 1ll << 27,
     "CSR entries (2 x links) below which K <= 56 takes the lane-per-link kernels (their class lists pack an entry index into 27 bits)"},
    {"wt", "SVILS_WT", "-1", GRAPH, I32, OFF(wt), -1, 1,
     "write-through row stores of the lane-per-link finalise / phi passes: -1 when the n-by-k state is 1 - 8 MB, 0 / 1 forced"},
    {"fused3", "SVILS_FUSED3", "-1", GRAPH, I32, OFF(fused3), -1, 1,
     "three-launch sweeps (in-launch hand-offs between role blocks): -1 by the static co-residency check, 0 / 1 forced"},
    {"one_comm", "SVILS_ONE_COMM", "0", ANY, I32, OFF(one_comm), 0, 1,
     "1: the chunked row exchange of node-block sweeps shares the first communicator instead of forming a second one"},
    {"xchunks", "SVILS_XCHUNKS", "0", ANY, U32, OFF(xchunks), 0, 64,
     "chunks of the pipelined row exchange; 0: one below 256 MB of rows, then one per 128 MB, at most 8"},
    {"row_exchange", "SVILS_ROW_EXCHANGE", "0", ANY, I32, OFF(row_exchange), 0, 2,
     "unchunked row exchange: 0 all-gather of padded slices while world x largest block <= 1.5 n, else exact-count broadcasts; "
     "1 / 2 force the all-gather / the broadcasts (SVILS_ALLGATHER_ROWS / SVILS_EXACT_ROWS set 1 / 2 too)"},
    {"sharded_graphs", "SVILS_SHARDED_GRAPHS", "1", ANY, I32, OFF(sharded_graphs), 0, 1,
     "0: node-block sweeps are enqueued launch by launch, never replayed as hipGraphs with their collectives"},
    {"report_staged", "SVILS_REPORT_STAGED", "0", ANY, I32, OFF(report_staged), 0, 1,
     "1: reports of any size go through the device staging slot and the copy stream (default: up to 1 MB straight into pinned memory)"},
#ifdef SVILS_TESTING
    {"fault_inject", "SVILS_FAULT_INJECT", "0", GRAPH, I32, OFF(fault_inject), 0, 1,
     "TESTING BUILD ONLY: one classification worker never publishes its tile (the bounded in-launch wait must fire)"},
    {"assume_cus", "SVILS_ASSUME_CUS", "0", GRAPH, I32, OFF(assume_cus), 0, 4096,
     "TESTING BUILD ONLY: the co-residency check pretends the device has this many CUs"},
#endif
};
#undef OFF
constexpr size_t kNRows = sizeof kRows / sizeof kRows[0];

bool parse(const Row &r, const char *v, Options &o) {
  if (!v || !*v) return false;
  char *end = nullptr;
  long long x = 0;
  if (r.kind == U64) {
    const unsigned long long u = strtoull(v, &end, 10);
    if (end == v || *end) return false;
    x = (long long)std::min<unsigned long long>(u, (unsigned long long)r.hi);
  } else {
    x = strtoll(v, &end, 10);
    if (end == v || *end) return false;
  }
  if (x < r.lo) return false;
  if (x > r.hi) { if (r.kind == U64) x = r.hi; else return false; }
  char *p = reinterpret_cast<char *>(&o) + r.off;
  switch (r.kind) {
    case I32: *reinterpret_cast<int *>(p) = (int)x; break;
    case U32: *reinterpret_cast<uint32_t *>(p) = (uint32_t)x; break;
    case I64: *reinterpret_cast<int64_t *>(p) = (int64_t)x; break;
    case U64: *reinterpret_cast<uint64_t *>(p) = (uint64_t)x; break;
  }
  return true;
}
const Row *find(const char *key) {
  for (const Row &r : kRows)
    if (key && strcmp(r.key, key) == 0) return &r;
  return nullptr;
}
}  // namespace

Options options_from_env() {
  Options o;
  for (const Row &r : kRows) {
    const char *e = getenv(r.env);
    if (!e) continue;
#ifdef SVILS_TESTING
    if (strcmp(r.key, "fault_inject") == 0) { o.fault_inject = strcmp(e, "cls_handoff") == 0 ? 1 : 0; continue; }
#endif
    if (strcmp(r.key, "graph_after") == 0) { o.graph_after = (uint32_t)std::max(0, atoi(e)); continue; }   // (negative: 0)
    (void)parse(r, e, o);   // a value that does not parse leaves the default
  }
  // the two one-word forms of row_exchange (rounds 5's names, kept)
  if (getenv("SVILS_ALLGATHER_ROWS")) o.row_exchange = 1;
  if (getenv("SVILS_EXACT_ROWS")) o.row_exchange = 2;
  return o;
}

int option_set(Options &o, const char *key, const char *value, bool created, bool have_graph) {
  const Row *r = find(key);
  if (!r) return -1;
  if ((r->until == CREATE && created) || (r->until == GRAPH && have_graph)) return -3;
  return parse(*r, value, o) ? 0 : -2;
}

int option_get(const Options &o, const char *key, char *buf, size_t cap) {
  const Row *r = find(key);
  if (!r) return -1;
  const char *p = reinterpret_cast<const char *>(&o) + r->off;
  long long x = 0;
  switch (r->kind) {
    case I32: x = *reinterpret_cast<const int *>(p); break;
    case U32: x = *reinterpret_cast<const uint32_t *>(p); break;
    case I64: x = *reinterpret_cast<const int64_t *>(p); break;
    case U64: x = (long long)*reinterpret_cast<const uint64_t *>(p); break;
  }
  if (buf && cap) snprintf(buf, cap, "%lld", x);
  return 0;
}

const char *option_table_text() {
  static const std::string text = [] {
    std::string s = "key\tenvironment\tdefault\tsettable\tmeaning\n";
    for (const Row &r : kRows) {
      s += r.key; s += '\t'; s += r.env; s += '\t'; s += r.dflt; s += '\t';
      s += r.until == ANY ? "any time" : r.until == GRAPH ? "before svils_set_graph" : "environment only (read by svils_create)";
      s += '\t'; s += r.doc; s += '\n';
    }
    s += "-\tSVILS_RCCL_LIBRARY\tlibrccl.so.1\tenvironment only (read at the first svils_comm_init of the process)\t"
         "the RCCL build the multi-GPU driver binds with dlopen; a named library that does not load is an error\n";
    return s;
  }();
  return text.c_str();
}

}  // namespace svils_impl

extern "C" {

int svils_set_option(svils_handle *h, const char *key, const char *value) {
  if (!h || !key || !value) return fail(SVILS_ERR_ARG, "svils_set_option: null argument");
  if (TILED(h)) {   // the tiles are handles of their own
    for (svils_handle *t : h->tiles) { const int rc = svils_set_option(t, key, value); if (rc) return rc; }
    return 0;
  }
  const int rc = option_set(h->opt, key, value, true, h->have_graph);
  if (rc == -1) return fail(SVILS_ERR_ARG, "svils_set_option: unknown option \"%s\" (svils_option_table lists them)", key);
  if (rc == -2) return fail(SVILS_ERR_ARG, "svils_set_option: \"%s\" is not a value of option %s", value, key);
  if (rc == -3) return fail(SVILS_ERR_ARG, "svils_set_option: %s was consumed when the handle %s; set its environment variable before svils_create",
                            key, h->have_graph ? "got its graph" : "was created");
  if (strcmp(key, "graph_after") == 0) h->graph_after = h->opt.graph_after;
  if (strcmp(key, "shard_fold") == 0) h->shard_fold_ok = h->opt.shard_fold != 0;
  if (strcmp(key, "xchunks") == 0) h->xchunks = h->opt.xchunks;
  if (strcmp(key, "shard_fold") == 0 || strcmp(key, "xchunks") == 0 || strcmp(key, "row_exchange") == 0 || strcmp(key, "one_comm") == 0) {
    // captured node-block sweeps took the other form: nothing of them may be in flight when they go
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs_of(h);
  }
  return 0;
}

int svils_get_option(svils_handle *h, const char *key, char *value, size_t cap) {
  if (!h || !key || !value || !cap) return fail(SVILS_ERR_ARG, "svils_get_option: null argument");
  const svils_handle *src = TILED(h) ? h->tiles[0] : h;
  if (option_get(src->opt, key, value, cap)) return fail(SVILS_ERR_ARG, "svils_get_option: unknown option \"%s\"", key);
  return 0;
}

const char *svils_option_table(void) { return option_table_text(); }

}  // extern "C"
