// capi.cc -- C entry points over the host-side C++ classes, for Python
// (bench.py, tests) to obtain the product's own inputs for the C ABI of
// include/svils.h: graph reading, held-out sampling, gamma/lambda
// initialisation and the training-link list.  No device is touched here.
#include <chrono>
#include "fixedfmt.hh"
#include <cstring>
#include <memory>

#include "env.hh"
#include "linksampling.hh"
#include "mmsbbatch.hh"
#include "network.hh"
#include "nmi.hh"

using namespace svinet;

extern "C" {

typedef struct {
  uint32_t n, k;
  double seed;
  double heldout_ratio;
  double link_thresh;
  uint32_t lt_min_deg;
  int32_t eta_type;        // 0 uniform, 1 fromdata, 2 sparse, 3 dense
  int32_t accuracy;
  int32_t defer_gamma;     // 1: no host init_gamma2 (the caller draws on the device: svih_init_links / svih_init_streams -> svils_init_gamma)
} svih_options;

struct svih_setup {
  std::unique_ptr<Env> env;
  std::unique_ptr<Network> net;
  std::unique_ptr<LinkSampling> ls;
};

static svih_setup *make_setup(const svih_options *o, const char *path, const int32_t *pairs,
                              uint64_t nlines) {
  static const char *eta_names[] = {"uniform", "fromdata", "sparse", "dense"};
  Env::Args a;
  a.n = o->n;
  a.k = o->k;
  a.link_sampling = true;
  a.rand_seed = o->seed;
  a.hol_ratio = o->heldout_ratio;
  a.link_thresh = o->link_thresh;
  a.lt_min_deg = o->lt_min_deg;
  a.eta_type = eta_names[(o->eta_type >= 0 && o->eta_type < 4) ? o->eta_type : 0];
  a.accuracy = o->accuracy != 0;
  a.defer_init_gamma = o->defer_gamma != 0;
  a.write_files = false;
  svih_setup *s = new svih_setup();
  s->env.reset(new Env(a));
  s->net.reset(new Network(*s->env));
  if (path) {
    if (s->net->read(path) < 0) { delete s; return nullptr; }
  } else {
    s->net->read_pairs(pairs, nlines);
  }
  s->env->n = s->net->n() - s->net->singles();   // src/main.cc:291
  s->ls.reset(new LinkSampling(*s->env, *s->net, /*attach_device=*/false));
  return s;
}

void svih_options_default(svih_options *o, uint32_t n, uint32_t k) {
  memset(o, 0, sizeof *o);
  o->n = n; o->k = k; o->seed = 0; o->heldout_ratio = 0.01; o->link_thresh = 0.5;
  o->lt_min_deg = 0; o->eta_type = 0; o->accuracy = 0; o->defer_gamma = 0;
}
svih_setup *svih_setup_from_file(const char *path, const svih_options *o) { return make_setup(o, path, nullptr, 0); }
svih_setup *svih_setup_from_pairs(const int32_t *pairs, uint64_t nlines, const svih_options *o) {
  return make_setup(o, nullptr, pairs, nlines);
}
void svih_setup_free(svih_setup *s) { delete s; }

uint32_t svih_n(const svih_setup *s) { return s->ls->n(); }
uint32_t svih_k(const svih_setup *s) { return s->ls->k(); }
uint32_t svih_ones(const svih_setup *s) { return s->net->ones(); }
uint32_t svih_singles(const svih_setup *s) { return s->net->singles(); }
double svih_total_pairs(const svih_setup *s) { return s->ls->total_pairs(); }
double svih_ones_prob(const svih_setup *s) { return s->ls->ones_prob(); }
double svih_eta0(const svih_setup *s) { return s->env->eta0; }
double svih_eta1(const svih_setup *s) { return s->env->eta1; }
const uint32_t *svih_seq2id(const svih_setup *s) { return s->net->seq2id().data(); }
const double *svih_gamma(const svih_setup *s) { return s->ls->gamma().empty() ? nullptr : s->ls->gamma().data(); }   // null: defer_gamma
const double *svih_lambda(const svih_setup *s) { return s->ls->lambda().data(); }
uint64_t svih_nvalidation(const svih_setup *s) { return s->ls->validation_sorted().size() / 3; }
const uint32_t *svih_validation_sorted(const svih_setup *s) { return s->ls->validation_sorted().data(); }
const uint32_t *svih_validation_accept(const svih_setup *s) { return s->ls->validation_accept().data(); }
uint64_t svih_nlinks(svih_setup *s) { return s->ls->training_links().size() / 2; }
const uint32_t *svih_links(svih_setup *s) { return s->ls->training_links().data(); }
const uint32_t *svih_edges(const svih_setup *s) { return &s->net->edges()[0].first; }
// what svils_init_gamma takes (tests/test_gpu_init.py): the links in drawing order, the generator's states by jump-ahead
uint64_t svih_init_links(const svih_setup *s, uint32_t *out /* [ones][2] or null */) {
  std::vector<uint32_t> e;
  s->ls->init_links(&e);
  if (out) memcpy(out, e.data(), e.size() * sizeof(uint32_t));
  return e.size() / 2;
}
uint64_t svih_init_offset(const svih_setup *s) { return s->ls->init_offset(); }   // outputs drawn before init_gamma2 began
int svih_init_streams(const svih_setup *s, uint64_t nstreams, uint64_t per_stream, uint32_t *out /* [nstreams][624] */) {
  std::vector<uint32_t> st;
  if (!s->ls->init_streams(nstreams, per_stream, &st)) return -1;
  memcpy(out, st.data(), st.size() * sizeof(uint32_t));
  return 0;
}
uint32_t svih_deg(const svih_setup *s, uint32_t p) { return s->net->deg(p); }

// normalised mutual information of a communities.txt-style file against a ground-truth file in the
// "node<TAB>community ..." format of -nmi; < 0 when a file cannot be read
// jump-ahead of the random stream (rng.hh / mtjump.hh) against drawing: the `count` outputs from position `pos` on, once by
// GslMt19937::at(pos) and once by `pos` calls of get().  0: equal; 1: different; -1: the jump machinery is unavailable.
// out_ms (may be null): [0] time of the jump, [1] time of drawing up to pos.
int svih_mt_jump_check(unsigned long seed, uint64_t pos, uint32_t count, double *out_ms) {
  using clk = std::chrono::steady_clock;
  svinet::GslMt19937 a(seed), b(seed), j;
  const auto t0 = clk::now();
  if (!a.at(pos, &j)) return -1;
  const auto t1 = clk::now();
  for (uint64_t i = 0; i < pos; ++i) (void)b.get();
  const auto t2 = clk::now();
  if (out_ms) {
    out_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    out_ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
  }
  if (j.position() != pos || b.position() != pos) return 1;
  for (uint32_t i = 0; i < count; ++i)
    if (j.get() != b.get()) return 1;
  return j.position() == pos + count ? 0 : 1;
}

double svih_nmi(const char *communities_path, const char *ground_truth_path) {
  Cover x, y;
  if (!read_cover_lines(communities_path, &x) || !read_cover_memberships(ground_truth_path, &y)) return -1.0;
  return lfk_nmi(y, x);
}

// ---- the -batch engine (host CPU), for the tests ----
struct svih_batch {
  std::unique_ptr<Env> env;
  std::unique_ptr<Network> net;
  std::unique_ptr<MMSBBatch> eng;
};

svih_batch *svih_batch_from_file(const char *path, const svih_options *o) {
  static const char *eta_names[] = {"uniform", "fromdata", "sparse", "dense"};
  Env::Args a;
  a.n = o->n;
  a.k = o->k;
  a.batch = true;
  a.rand_seed = o->seed;
  a.hol_ratio = o->heldout_ratio;
  a.eta_type = eta_names[(o->eta_type >= 0 && o->eta_type < 4) ? o->eta_type : 0];
  a.write_files = false;
  svih_batch *b = new svih_batch();
  b->env.reset(new Env(a));
  b->net.reset(new Network(*b->env));
  if (b->net->read(path) < 0) { delete b; return nullptr; }
  b->env->n = b->net->n() - b->net->singles();
  b->eng.reset(new MMSBBatch(*b->env, *b->net));
  return b;
}
void svih_batch_free(svih_batch *b) { delete b; }
uint32_t svih_batch_n(const svih_batch *b) { return b->eng->n(); }
uint32_t svih_batch_iter(const svih_batch *b) { return b->eng->iter(); }
double *svih_batch_gamma(svih_batch *b) { return b->eng->gamma().data(); }
double *svih_batch_lambda(svih_batch *b) { return b->eng->lambda().data(); }
uint64_t svih_batch_nheldout(const svih_batch *b) { return b->eng->heldout_edges().size() / 2; }
const uint32_t *svih_batch_heldout(const svih_batch *b) { return b->eng->heldout_edges().data(); }
uint64_t svih_batch_nvalidation(const svih_batch *b) { return b->eng->validation_edges().size() / 2; }
const uint32_t *svih_batch_validation(const svih_batch *b) { return b->eng->validation_edges().data(); }
uint64_t svih_batch_nrows(const svih_batch *b) { return b->eng->heldout_rows().size() / 10; }
const double *svih_batch_rows(const svih_batch *b) { return b->eng->heldout_rows().data(); }
void svih_batch_sweep(svih_batch *b) { b->eng->sweep(); }
int svih_batch_report(svih_batch *b) { return b->eng->report() ? 1 : 0; }
double svih_batch_eta0(const svih_batch *b) { return b->env->eta0; }
double svih_batch_eta1(const svih_batch *b) { return b->env->eta1; }
double svih_batch_ones_prob(const svih_batch *b) { return b->env->ones_prob; }
const uint32_t *svih_batch_edges(const svih_batch *b) { return &b->net->edges()[0].first; }
uint32_t svih_batch_ones(const svih_batch *b) { return b->net->ones(); }

// tests: how many of `count` values differ between the writers' fixed-point formatter (fixedfmt.hh) and printf's
// "%.5f" / "%.3f" -- the values come as they are, the test chooses them
uint64_t svih_fixed_format_mismatches(const double *v, uint64_t count) {
  uint64_t bad = 0;
  std::string a;
  char tmp[400];
  for (uint64_t i = 0; i < count; ++i) {
    a.clear();
    append_fixed<5>(a, v[i], '\t');
    snprintf(tmp, sizeof tmp, "%.5f\t", v[i]);
    bad += a != tmp;
    a.clear();
    append_fixed<3>(a, v[i], '\n');
    snprintf(tmp, sizeof tmp, "%.3f\n", v[i]);
    bad += a != tmp;
  }
  return bad;
}

}  // extern "C"
