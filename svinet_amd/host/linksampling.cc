#include "linksampling.hh"

#include <array>
#include <random>
#include <thread>

#include <algorithm>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "fixedfmt.hh"
#include "svils.h"

namespace svinet {

namespace {
void die_svils(const char *what) {
  fprintf(stderr, "error: %s: %s\n", what, svils_last_error());
  exit(-1);
}
FILE *open_or_die(const std::string &path, const char *what) {
  FILE *f = fopen(path.c_str(), "w");
  if (!f) {
    printf("cannot open %s file:%s\n", what, strerror(errno));
    exit(-1);
  }
  return f;
}
// CPUs this process may really use: the affinity mask and the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us), not the
// machine's core count -- a container with a 16-CPU quota on a 256-core host gains nothing from 64 threads
unsigned usable_cpus() {
  unsigned n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n ? n : 1u, (unsigned)CPU_COUNT(&set));
  double quota = 0.0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[64];
    long per = 0;
    if (fscanf(f, "%63s %ld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) quota = atof(a) / (double)per;
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    long q = -1, per = 100000;
    if (fscanf(g, "%ld", &q) != 1) q = -1;
    fclose(g);
    if (FILE *h2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h2, "%ld", &per) != 1) per = 100000; fclose(h2); }
    if (q > 0 && per > 0) quota = (double)q / (double)per;
  }
  if (quota >= 1.0) n = std::min<unsigned>(n, (unsigned)(quota + 0.5));
  return std::max(1u, n);
}
double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
}  // namespace

// LinkSampling::LinkSampling, src/linksampling.cc:5-155
LinkSampling::LinkSampling(Env &env, Network &network, bool attach_device)
    : env_(env), network_(network), n_(env.n), k_(env.k),
      rng_(env.seed ? (unsigned long)env.seed : 0ul),   // :70-75
      start_time_(time(0)) {
  const double t_ctor = now_s();
  const bool trace_ctor = getenv("SVINET_TRACE_LOOP") != nullptr;
  auto mark = [&](const char *what) { if (trace_ctor) fprintf(stderr, "[ctor] +%.3f s: %s\n", now_s() - t_ctor, what); };
  // `_n * (_n - 1) / 2` in 32-bit unsigned arithmetic (:36-37, quirk Q5)
  total_pairs_ = (double)((uint32_t)(n_ * (n_ - 1u)) / 2u);
  Env::plog("inference n", n_);
  Env::plog("total pairs", total_pairs_);
  ones_prob_ = double(network_.ones()) / total_pairs_;
  zeros_prob_ = 1 - ones_prob_;
  Env::plog("ones_prob", ones_prob_);
  Env::plog("zeros_prob", zeros_prob_);
  uint32_t maxd;
  double avgd;
  network_.deg_stats(maxd, avgd);
  Env::plog("avg degree", avgd);
  Env::plog("max degree", maxd);
  {
    std::ostringstream sa;  // D2Array::s(), src/matrix.hh:898-923 (row/col limits 512/32)
    sa << "\n[ ";
    for (uint32_t i = 0; i < std::min<uint32_t>(k_, 512); ++i) {
      if (i > 0) sa << "  ";
      sa << env_.eta0 << " " << env_.eta1 << " " << "\n";
    }
    sa << "]";
    Env::plog("eta", sa.str());
  }

  FILE *vef = nullptr;
  if (env_.write_files) {
    vef = open_or_die(Env::file_str("/validation-edges.txt"), "validation edges");
    if (!env_.load_test) fclose(open_or_die(Env::file_str("/test-edges.txt"), "test edges"));   // load_test() writes it
  }
  if (!env_.load_heldout) {
    Env::plog("load validation from file:", false);
    init_validation();
    Env::plog("heldout ratio", env_.heldout_ratio);
    Env::plog("validation pairs (1s and 0s)", (uint32_t)validation_map_.size());
  } else {
    Env::plog("load validation from file:", true);
    load_validation();
  }
  if (vef) {
    fprintf(vef, "%s\n", edgelist_s(val_accept_).c_str());
    fclose(vef);
  }
  if (env_.load_test) {                          // src/linksampling.cc:105-108: after the validation sample, so the
    Env::plog("load test from file:", true);     // sampler never saw these pairs
    load_test();
  }

  if (env_.nmi) {   // Network::load_ground_truth / write_gt_communities (src/network.cc:252-307,508-525)
    if (!read_cover_memberships(env_.ground_truth_fname, &ground_truth_)) {
      fprintf(stderr, "error: cannot read ground truth file %s; check path; skipping file\n", env_.ground_truth_fname.c_str());
    } else if (env_.write_files) {
      printf("+ Done loading ground truth\n+ Writing ground truth communities\n");
      FILE *f = open_or_die(Env::file_str("/ground_truth.txt"), "ground truth");
      FILE *g = open_or_die(Env::file_str("/ground_truth_community_sizes.txt"), "ground truth sizes");
      uint32_t c = 0;
      for (const auto &v : ground_truth_) {
        fprintf(g, "%d\t%ld\n", c++, (long)v.size());
        for (uint32_t id : v) fprintf(f, "%d ", id);
        fprintf(f, "\n");
      }
      fclose(f);
      fclose(g);
    }
  }

  mark("held-out sets done");
  lambda_.assign(2 * (size_t)k_, 0.0);
  defer_init_ = !env_.model_load && !env_.use_init_communities && (device_init_wanted(attach_device) || (!attach_device && env_.defer_init_gamma));
  if (!defer_init_) gamma_.assign((size_t)n_ * k_, 0.0);   // (left to the device: no host copy until the final fetch)
  if (env_.model_load) {
    if (load_model() < 0) exit(-1);
  } else if (env_.use_init_communities) {        // src/linksampling.cc:112-115 (nolambda is never set on this path)
    init_gamma_external();
    init_lambda();
  } else {
    init_o0_ = rng_.position();
    if (!defer_init_) init_gamma2();
    init_lambda();
  }

  mark(defer_init_ ? "lambda initialised (gamma: on the device, below)" : "gamma / lambda initialised");
  if (env_.write_files) {
    tf_ = open_or_die(Env::file_str("/test.txt"), "test");
    vf_ = open_or_die(Env::file_str("/validation.txt"), "validation");
    fclose(open_or_die(Env::file_str("/logl.txt"), "logl"));
  }
  Env::plog("network ones", network_.ones());
  Env::plog("network singles", network_.singles());

  // std::map<Edge,bool> iteration order for the likelihood loop (:974-992)
  val_sorted_.reserve(validation_map_.size() * 3);
  for (const auto &kv : validation_map_) {
    val_sorted_.push_back(kv.first.first);
    val_sorted_.push_back(kv.first.second);
    val_sorted_.push_back(network_.y(kv.first.first, kv.first.second) ? 1u : 0u);
  }

  // ... and of test_likelihood's loop over _test_map (:1154-1172); y is asked of the network there
  for (const auto &kv : test_map_) {
    test_sorted_.push_back(kv.first.first);
    test_sorted_.push_back(kv.first.second);
    test_sorted_.push_back(network_.y(kv.first.first, kv.first.second) ? 1u : 0u);
  }

  if (env_.write_files) rank_external_ids();
  mark("sorted pair lists, id ranks");
  if (attach_device) {
    attach();
    mark("device attached (handle, communicator, validation set, state upload)");
    if (!env_.accuracy && !val_sorted_.empty()) {
      // constructor-time validation_likelihood (:149-150): row "iter 0"
      if (svils_validation_row(h_, row0_)) die_svils("svils_validation_row");
      have_row0_ = true;
      if (vf_) {
        write_validation_row(row0_, vf_);
        write_max(row0_, -1, -2147483647.0);
      }
    }
  }
  start_time_ = time(0);
  timing_.ctor_s = now_s() - t_ctor;
}

bool LinkSampling::pipelined_reports() const {
  if (env_.sharded || env_.kshard || env_.minibatch || env_.gpus > 1) return false;
  if (k_ > SVILS_MAX_K) return false;   // a column-tiled handle (svils.h): no report slots, the synchronous loop
  const char *e = getenv("SVINET_SYNC_REPORTS");
  return !(e && atoi(e) != 0);
}

LinkSampling::~LinkSampling() {
  if (h_) svils_destroy(h_);
  if (vf_) fclose(vf_);
  if (tf_) fclose(tf_);
}

void LinkSampling::attach() {
  const double t_attach = now_s();
  const bool trace_attach = getenv("SVINET_TRACE_LOOP") != nullptr;
  auto amark = [&](const char *what) { if (trace_attach) fprintf(stderr, "[attach] +%.3f s: %s\n", now_s() - t_attach, what); };
  svils_config cfg;
  svils_config_default(&cfg, n_, k_);
  cfg.ones = network_.ones();
  cfg.alpha = env_.alpha;
  cfg.eta0 = env_.eta0;
  cfg.eta1 = env_.eta1;
  cfg.epsilon = env_.epsilon;
  cfg.link_thresh = env_.link_thresh;
  cfg.lt_min_deg = env_.lt_min_deg;
  cfg.reportfreq = env_.reportfreq;
  cfg.use_validation_stop = env_.use_validation_stop ? 1 : 0;
  cfg.ones_prob = ones_prob_;
  cfg.zeros_prob = zeros_prob_;
  cfg.device = env_.device;
  cfg.sparse_after_iter = env_.sparse_after;
  if (env_.kshard) {     // -gpus N -kshard: this process owns the columns [k0_, k1_) of every row
    k0_ = (uint32_t)((uint64_t)k_ * (uint32_t)env_.rank / (uint32_t)env_.gpus);
    k1_ = (uint32_t)((uint64_t)k_ * ((uint32_t)env_.rank + 1) / (uint32_t)env_.gpus);
    cfg.k = k1_ - k0_;
    cfg.k_begin = k0_;
    cfg.k_total = k_;
  } else if (env_.sharded && env_.minibatch) {   // -gpus N -minibatch m: equal node blocks (every rank steps through windows of its own)
    const uint32_t B = (n_ + (uint32_t)env_.gpus - 1) / (uint32_t)env_.gpus;
    cfg.node_begin = std::min(n_, (uint32_t)env_.rank * B);
    cfg.node_end = std::min(n_, ((uint32_t)env_.rank + 1) * B);
    cfg.n_alloc = B * (uint32_t)env_.gpus;
  } else if (env_.sharded) {   // -gpus N: this process owns the node block of its rank, blocks balanced by work
    // the reference numbers nodes by first appearance (src/network.cc:10-116): hubs come first, equal-count blocks would
    // give rank 0 of 8 on ca-AstroPh three times the mean number of links to evaluate.  Every rank cuts the same bounds
    // from the same training links.
    const std::vector<uint32_t> &L = training_links();
    blocks_.assign((size_t)env_.gpus + 1, 0);
    if (svils_balance_node_blocks(L.data(), L.size() / 2, n_, env_.gpus, -1.0, blocks_.data())) die_svils("svils_balance_node_blocks");
    cfg.node_begin = blocks_[(size_t)env_.rank];
    cfg.node_end = blocks_[(size_t)env_.rank + 1];
  }
  if (svils_create(&cfg, &h_)) die_svils("svils_create");
  if (!blocks_.empty() && svils_set_node_blocks(h_, env_.rank, env_.gpus, blocks_.data())) die_svils("svils_set_node_blocks");
  if (env_.gpus > 1 || env_.sharded) {
    // rank 0 makes the ncclUniqueId and writes it into the pipes main() made before the fork; the other
    // ranks read it from theirs (a rank 0 that died closes the pipe: the read fails, nobody waits for ever).
    // Then the collective communicator init.
    unsigned char id[SVILS_COMM_ID_BYTES];
    if (env_.gpus == 1) {   // -sharded with one GPU: a communicator of one rank, nobody to tell
      if (svils_comm_unique_id(id)) die_svils("svils_comm_unique_id");
    } else if (env_.rank == 0) {
      if (svils_comm_unique_id(id)) die_svils("svils_comm_unique_id");
      for (int fd : env_.comm_wfds) {
        if (write(fd, id, sizeof id) != (ssize_t)sizeof id) { perror("rank 0: communicator id"); exit(-1); }
        close(fd);
      }
    } else {
      size_t got = 0;
      while (got < sizeof id) {
        const ssize_t r = read(env_.comm_rfd, id + got, sizeof id - got);
        if (r <= 0) { fprintf(stderr, "rank %d: no communicator id from rank 0\n", env_.rank); exit(-1); }
        got += (size_t)r;
      }
      close(env_.comm_rfd);
    }
    if (svils_comm_init(h_, id, env_.rank, env_.gpus)) die_svils("svils_comm_init");
  }
  if (env_.minibatch) {
    // random relabelling (own generator: the GSL stream of the samplers / init is not disturbed)
    std::mt19937_64 eng(0x5eed5eedull + (uint64_t)env_.seed);
    dev_of_.resize(n_);
    for (uint32_t i = 0; i < n_; ++i) dev_of_[i] = i;
    for (uint32_t i = n_; i > 1; --i) std::swap(dev_of_[i - 1], dev_of_[eng() % i]);
    seq_of_.resize(n_);
    for (uint32_t i = 0; i < n_; ++i) seq_of_[dev_of_[i]] = i;
    svils_stochastic sc;
    svils_stochastic_default(&sc, env_.minibatch);
    sc.tau0 = env_.tau0; sc.kappa = env_.kappa; sc.node_tau0 = env_.nodetau0; sc.node_kappa = env_.nodekappa;
    if (env_.sharded) sc.shard_block = (n_ + (uint32_t)env_.gpus - 1) / (uint32_t)env_.gpus;   // every rank steps through its own block
    if (svils_set_stochastic(h_, &sc)) die_svils("svils_set_stochastic");
  }
  amark("handle created (device arrays), communicator / mini-batch set up");
  if (env_.kshard || k_ > SVILS_MAX_K) send_graph();   // the K-sharded (and column-tiled) initial state needs the link list (row sums cross slices)
  // with -accuracy validation_likelihood() returns at once (:969-970)
  if (!env_.accuracy && !val_sorted_.empty()) {
    std::vector<uint32_t> v(val_sorted_);
    if (!dev_of_.empty())
      for (size_t i = 0; i < v.size(); i += 3) { v[i] = dev_of_[v[i]]; v[i + 1] = dev_of_[v[i + 1]]; }
    if (svils_set_validation(h_, v.data(), v.size() / 3)) die_svils("svils_set_validation");
  }
  if (!env_.accuracy && !test_sorted_.empty()) {      // test_likelihood returns at once with -accuracy (:1150-1151)
    if (env_.kshard) { fprintf(stderr, "error: -load-test is not available with -kshard\n"); exit(-1); }
    std::vector<uint32_t> v(test_sorted_);
    if (!dev_of_.empty())
      for (size_t i = 0; i < v.size(); i += 3) { v[i] = dev_of_[v[i]]; v[i + 1] = dev_of_[v[i + 1]]; }
    if (svils_set_test(h_, v.data(), v.size() / 3)) die_svils("svils_set_test");
  }
  if (env_.kshard) {
    const uint32_t w = k1_ - k0_;
    std::vector<double> g((size_t)n_ * w);
    for (uint32_t i = 0; i < n_; ++i)   // (-minibatch: rows in the relabelled order)
      std::copy(&gamma_[(size_t)i * k_ + k0_], &gamma_[(size_t)i * k_ + k0_] + w, &g[(size_t)(dev_of_.empty() ? i : dev_of_[i]) * w]);
    if (svils_set_state(h_, g.data(), &lambda_[2 * (size_t)k0_], nullptr)) die_svils("svils_set_state");
    if (svils_ksh_init_state(h_)) die_svils("svils_ksh_init_state");
  } else if (dev_of_.empty()) {
    if (!(defer_init_ && init_gamma2_on_device())) {
      if (defer_init_) {                         // (the device path was not taken after all: draw on the host as always)
        gamma_.assign((size_t)n_ * k_, 0.0);
        init_gamma2();
      }
      if (svils_set_state(h_, gamma_.data(), lambda_.data(), nullptr)) die_svils("svils_set_state");
    }
  } else {
    std::vector<double> g((size_t)n_ * k_);
    for (uint32_t i = 0; i < n_; ++i)
      std::copy(&gamma_[(size_t)i * k_], &gamma_[(size_t)(i + 1) * k_], &g[(size_t)dev_of_[i] * k_]);
    if (svils_set_state(h_, g.data(), lambda_.data(), nullptr)) die_svils("svils_set_state");
  }
  amark("held-out sets and state on the device");
  // the pipelined loop issues chunks of 1, 2, 4, 8, 16, 16 ... sweeps: their graphs are captured here, in the set-up,
  // not inside a run that may be over after 31 sweeps (svils.h: svils_prepare_graphs)
  if (pipelined_reports() && !getenv("SVILS_GRAPH_AFTER")) {
    send_graph();
    amark("training links on the device (svils_set_graph)");
    if (svils_prepare_graphs(h_, env_.sweep_batch ? std::min<uint32_t>(env_.sweep_batch, 64) : 16)) die_svils("svils_prepare_graphs");
    amark("hipGraphs of the chunks captured");
  }
}

// ---------------------------------------------------------------- validation set
bool LinkSampling::edge_ok(const Edge &e) const {       // src/linksampling.hh:296-326
  if (e.first == e.second) return false;
  if (test_map_.find(e) != test_map_.end()) return false;
  return validation_map_.find(e) == validation_map_.end();
}

void LinkSampling::get_random_edge(bool link, Edge &e) {  // src/linksampling.hh:328-349
  if (!link) {
    do {
      uint32_t a = rng_.uniform_int(n_);
      uint32_t b = rng_.uniform_int(n_);
      e = a < b ? Edge(a, b) : Edge(b, a);
    } while (!edge_ok(e));
  } else {
    do {
      e = network_.edges()[rng_.uniform_int(network_.ones())];
    } while (!edge_ok(e));
  }
}

void LinkSampling::accept_pair(const Edge &e, bool y) {
  val_accept_.push_back(e.first);
  val_accept_.push_back(e.second);
  val_accept_.push_back(y ? 1u : 0u);
  validation_map_[e] = true;
}

void LinkSampling::set_validation_sample(int s) {         // src/linksampling.cc:281-309
  int c0 = 0, c1 = 0;
  const int p = s / 2;
  while (c0 < p || c1 < p) {
    Edge e;
    get_random_edge(c0 == p, e);   // non-links first; a sampled pair that is a link counts as one
    const bool y = network_.y(e.first, e.second);
    if (!y && c0 < p) { c0++; accept_pair(e, false); }
    if (y && c1 < p) { c1++; accept_pair(e, true); }
  }
}

void LinkSampling::init_validation() {                    // src/linksampling.cc:164-188
  const int s1 = env_.heldout_ratio * network_.ones();
  set_validation_sample(s1);
}

void LinkSampling::load_validation() {                    // src/linksampling.cc:1382-1414
  FILE *f = fopen(env_.load_heldout_fname.c_str(), "r");
  if (!f) {
    fprintf(stderr, "error: cannot read test validation file %s\n", env_.load_heldout_fname.c_str());
    exit(-1);
  }
  int a, b;
  uint32_t cnt = 0;
  while (fscanf(f, "%d %d", &a, &b) == 2) {
    uint32_t p, q;
    if (!network_.id2seq((uint32_t)a, &p) || !network_.id2seq((uint32_t)b, &q)) {
      fprintf(stderr, "error: id %d or id %d not found in original network\n", a, b);
      exit(-1);
    }
    Edge e = p < q ? Edge(p, q) : Edge(q, p);
    accept_pair(e, network_.y(p, q));
    ++cnt;
  }
  fclose(f);
  Env::plog("link sampling: loaded validation heldout pairs:", cnt);
}

// LinkSampling::load_test, src/linksampling.cc:1417-1450: "id<TAB>id" lines of external ids; every pair is ordered and
// entered into _test_map (a std::map: a repeated pair collapses) and appended to _test_pairs (test-edges.txt lists
// every line, with the network's y)
void LinkSampling::load_test() {
  FILE *f = fopen(env_.load_test_fname.c_str(), "r");
  if (!f) {
    fprintf(stderr, "error: cannot read test test file %s\n", env_.load_test_fname.c_str());
    exit(-1);
  }
  int a, b;
  uint32_t cnt = 0;
  std::vector<uint32_t> listed;
  while (fscanf(f, "%d %d", &a, &b) == 2) {
    uint32_t p, q;
    if (!network_.id2seq((uint32_t)a, &p) || !network_.id2seq((uint32_t)b, &q)) {
      fprintf(stderr, "error: id %d or id %d not found in original network\n", a, b);
      exit(-1);
    }
    const Edge e = p < q ? Edge(p, q) : Edge(q, p);
    test_map_[e] = true;
    listed.push_back(e.first);
    listed.push_back(e.second);
    listed.push_back(network_.y(p, q) ? 1u : 0u);
    ++cnt;
  }
  fclose(f);
  Env::plog("link sampling: loaded test heldout pairs:", cnt);
  if (env_.write_files) {
    FILE *tef = open_or_die(Env::file_str("/test-edges.txt"), "test edges");
    fprintf(tef, "%s\n", edgelist_s(listed).c_str());
    fclose(tef);
  }
}

std::string LinkSampling::edgelist_s(const std::vector<uint32_t> &t) const {  // :190-206
  std::ostringstream sa;
  const std::vector<uint32_t> &s2i = network_.seq2id();
  for (size_t i = 0; i + 2 < t.size(); i += 3)
    sa << s2i[t[i]] << "\t" << s2i[t[i + 1]] << "\t" << (int)t[i + 2] << "\n";
  return sa.str();
}

// ------------------------------------------------------------------ initialisation
// init_gamma2 (src/linksampling.cc:374-401): for every link (held-out ones included), in the order
// "for p, for q in adj[p] with p < q", draw K uniforms, normalise them to sum 1 and add the vector
// to gamma[p] and gamma[q].  The uniforms come from ONE sequential stream, so they are drawn by this
// thread, chunk by chunk; normalising a chunk and adding it into gamma is done by worker threads
// while the next chunk is being drawn.  Every node's row is owned by one worker and receives its
// links' vectors in link order, so the result is bit-identical to the sequential loop.
// Network::load_init_communities (src/network.cc:374-440) + LinkSampling::init_gamma_external (src/linksampling.cc:405-453).
// The file holds one community per line (external ids, whitespace separated); line c is community c.  gamma starts at
// alpha; for every adjacency entry of p (every link of p, held-out ones included) the vector phi = alpha everywhere,
// + n / |c(p)| on each community that lists p, normalised to sum 1, is added to gamma[p] -- deg(p) additions of one
// vector, done as repeated additions as the reference does them.  No random draw.  init_memberships.txt is written
// as the reference writes it.
void LinkSampling::init_gamma_external() {
  FILE *f = fopen(env_.init_communities_fname.c_str(), "r");
  if (!f) {
    fprintf(stderr, "error: cannot read init communities file %s\n", env_.init_communities_fname.c_str());
    exit(-1);
  }
  printf("+ Loading init communities from %s\n", env_.init_communities_fname.c_str());
  std::vector<std::vector<uint32_t>> of_node(n_);   // _init_communities_seq: node -> communities in line order
  char *line = nullptr;
  size_t cap = 0;
  uint32_t cid = 0;
  ssize_t got;
  std::string prev;     // the reference's scratch buffer `s` (src/network.cc:382,391)
  while ((got = getline(&line, &cap, f)) > 0) {
    // The reference copies the line into `s` with sscanf("%[^\n]") and skips it only when that returns < 0.  On an EMPTY
    // line the directive matches nothing and sscanf returns 0: `s` keeps the PREVIOUS line, whose members are parsed
    // again as community `cid` (:391-416).  Reproduced; a blank FIRST line reads the reference's uninitialised buffer --
    // here it is an empty community.
    if (line[0] != '\n') prev.assign(line, strcspn(line, "\n"));
    const char *p = prev.c_str();
    for (;;) {
      char *e = nullptr;
      const long u = strtol(p, &e, 10);
      if (e == p) break;
      p = e;
      uint32_t seq;
      if (!network_.id2seq((uint32_t)u, &seq)) {
        fprintf(stderr, "error: id %ld of the init communities file is not in the network\n", u);
        exit(-1);
      }
      of_node[seq].push_back(cid);
    }
    cid++;
  }
  free(line);
  fclose(f);
  printf("+ Loaded %d init communities\n", cid);
  if (env_.write_files) {
    FILE *g = open_or_die(Env::file_str("/init_memberships.txt"), "init memberships");
    const std::vector<uint32_t> &s2i = network_.seq2id();
    for (uint32_t i = 0; i < n_; ++i) {
      fprintf(g, "%d\t", s2i[i]);
      for (uint32_t c : of_node[i]) fprintf(g, "%d\t", c);
      fprintf(g, "\n");
    }
    fclose(g);
  }
  std::fill(gamma_.begin(), gamma_.end(), env_.alpha);
  std::vector<double> phi(k_);
  for (uint32_t p = 0; p < n_; ++p) {
    const std::vector<uint32_t> &r = of_node[p];
    for (uint32_t c : r)
      if (c >= k_) {   // the reference logs it and then writes phi[c] out of bounds
        fprintf(stderr, "error: the init communities file has more than k = %u lines (node %u is on line %u)\n", k_, network_.seq2id()[p], c);
        exit(-1);
      }
    std::fill(phi.begin(), phi.end(), env_.alpha);
    for (uint32_t c : r) phi[c] += (double)n_ / (double)r.size();
    double s = 0.0;
    for (uint32_t k = 0; k < k_; ++k) s += phi[k];
    for (uint32_t k = 0; k < k_; ++k) phi[k] = phi[k] / s;
    double *g = &gamma_[(size_t)p * k_];
    const size_t deg = network_.get_edges(p).size();
    for (size_t e = 0; e < deg; ++e)
      for (uint32_t k = 0; k < k_; ++k) g[k] += phi[k];
  }
}

// ---- init_gamma2 on the device (svils_init_gamma, csrc/svils_init.hip)
// Taken by the binary's one-GPU whole-graph run when the draws are worth it (E k >= 2^24 uniforms: below that the host loop is
// milliseconds); SVINET_INIT_DEVICE=1 / 0 forces / forbids it (tests: the two paths leave the same files, bit for bit).
bool LinkSampling::device_init_wanted(bool attach_device) const {
  if (!attach_device || env_.kshard || env_.sharded || env_.gpus > 1 || env_.minibatch || k_ > SVILS_MAX_K) return false;
  if (const char *e = getenv("SVINET_INIT_DEVICE")) return atoi(e) != 0;
  return (uint64_t)network_.ones() * k_ >= (1ull << 24);
}

void LinkSampling::init_links(std::vector<uint32_t> *edges) const {
  edges->clear();
  edges->reserve(2 * (size_t)network_.ones());
  for (uint32_t p = 0; p < n_; ++p)
    for (uint32_t q : network_.get_edges(p))
      if (p < q) { edges->push_back(p); edges->push_back(q); }   // all links, held-out ones included (src/linksampling.cc:378-386)
}

bool LinkSampling::init_streams(uint64_t nstreams, uint64_t per_stream, std::vector<uint32_t> *states) const {
  states->assign((size_t)nstreams * 624, 0u);
  MtJump stride;
  if (nstreams > 1 && !stride.make(per_stream)) return false;
  // thread t: the state at its first stream by a jump from the seed, then along the chain with the one stride polynomial
  const unsigned T = (unsigned)std::min<uint64_t>(std::min(usable_cpus(), 64u), nstreams);
  std::vector<char> ok(T, 1);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      const uint64_t s0 = nstreams * t / T, s1 = nstreams * (t + 1) / T;
      if (s0 >= s1) return;
      GslMt19937 g;
      uint32_t *st = states->data() + (size_t)s0 * 624;
      if (!rng_.at(init_o0_ + s0 * per_stream, &g) || !g.canonical_state(st)) { ok[t] = 0; return; }
      for (uint64_t s = s0 + 1; s < s1; ++s) {
        uint32_t *nx = states->data() + (size_t)s * 624;
        memcpy(nx, nx - 624, 624 * sizeof(uint32_t));
        stride.apply(nx);
      }
    });
  for (auto &x : th) x.join();
  for (char c : ok) if (!c) return false;
  return true;
}

bool LinkSampling::init_gamma2_on_device() {
  const double t0 = now_s();
  std::vector<uint32_t> edges, states;
  init_links(&edges);
  const uint64_t E = edges.size() / 2, total = E * (uint64_t)k_;
  if (!total) return false;
  // streams: enough wavefronts to fill the device (one per stream), each long enough to amortise its state (>= 256 twists),
  // few enough that the host's jump chain stays short (~1 ms per state and thread)
  const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>(2048, total / (624ull * 256ull)));
  const uint64_t per = ((total + want - 1) / want + 623) / 624 * 624;
  const uint64_t ns = (total + per - 1) / per;
  if (!init_streams(ns, per, &states)) return false;
  const double t1 = now_s();
  if (svils_init_gamma(h_, edges.data(), E, states.data(), ns, per, lambda_.data())) {
    fprintf(stderr, "note: init_gamma2 on the device not taken (%s); drawing on the host\n", svils_last_error());
    return false;
  }
  if (getenv("SVINET_TRACE_LOOP"))
    fprintf(stderr, "[ctor] init_gamma2 on the device: %llu links x %u, %llu streams of %llu outputs: host %.3f s (links + states), device call %.3f s\n",
            (unsigned long long)E, k_, (unsigned long long)ns, (unsigned long long)per, t1 - t0, now_s() - t1);
  return true;
}

void LinkSampling::init_gamma2() {
  std::vector<uint32_t> lp, lq;
  lp.reserve(network_.ones());
  lq.reserve(network_.ones());
  for (uint32_t p = 0; p < n_; ++p)
    for (uint32_t q : network_.get_edges(p))
      if (p < q) { lp.push_back(p); lq.push_back(q); }   // all links, held-out ones included
  const size_t E = lp.size(), K = k_;
  unsigned T = std::min(usable_cpus(), 64u);
  if (E * K < (1u << 22)) T = 1;   // small problems: threads cost more than they save
  // links per chunk: 32 MB of uniforms, 16 MB with more than 32 threads (two buffers per thread)
  size_t C = std::max<size_t>(256, ((size_t)(T > 32 ? 16 : 32) << 20) / (K * sizeof(double)));
  // (tests reach the threaded paths on small graphs with these two)
  if (const char *e = getenv("SVINET_INIT_THREADS")) T = (unsigned)std::max(1, atoi(e));
  if (const char *e = getenv("SVINET_INIT_CHUNK_LINKS")) C = (size_t)std::max(1, atoi(e));
  std::vector<double> buf[2];

  // normalise rows [b, e) of a chunk in place: sequential sum over k, then the division (:392-396)
  auto normalise = [&](double *v, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      double *u = v + i * K, s = .0;
      for (size_t k = 0; k < K; ++k) s += u[k];
      for (size_t k = 0; k < K; ++k) u[k] = u[k] / s;
    }
  };
  // add the chunk's vectors into the rows of the nodes in [nb, ne), in link order
  auto accumulate = [&](const double *v, size_t l0, size_t cnt, uint32_t nb, uint32_t ne) {
    for (size_t i = 0; i < cnt; ++i) {
      const uint32_t p = lp[l0 + i], q = lq[l0 + i];
      const double *u = v + i * K;
      if (p >= nb && p < ne) { double *g = &gamma_[(size_t)p * K]; for (size_t k = 0; k < K; ++k) g[k] += u[k]; }
      if (q >= nb && q < ne) { double *g = &gamma_[(size_t)q * K]; for (size_t k = 0; k < K; ++k) g[k] += u[k]; }
    }
  };
  auto process = [&](double *v, size_t l0, size_t cnt) {
    if (T == 1) { normalise(v, 0, cnt); accumulate(v, l0, cnt, 0, n_); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back(normalise, v, cnt * t / T, cnt * (t + 1) / T);
    for (auto &x : th) x.join();
    th.clear();
    for (unsigned t = 0; t < T; ++t)
      th.emplace_back(accumulate, v, l0, cnt, (uint32_t)((uint64_t)n_ * t / T), (uint32_t)((uint64_t)n_ * (t + 1) / T));
    for (auto &x : th) x.join();
  };

  // Large problems: the draws themselves are spread over the threads.  Every link consumes exactly K outputs, so chunk c
  // (links [c C, (c + 1) C)) starts at output o0 + K C c of the stream: thread t jumps there (mtjump.hh) and draws its chunk
  // on its own; in round r the T threads hold chunks r T .. r T + T - 1, which the node-range owners add into gamma in
  // chunk order while round r + 1 is being drawn -- the same stream, the same per-row order of additions, the same bits.
  // One polynomial (x^(T C K) mod phi) carries every thread from its chunk of round r to its chunk of round r + 1.
  if (T > 1 && E > 4 * C) {
    const uint64_t o0 = rng_.position();
    MtJump stride;
    std::vector<std::array<uint32_t, 624>> st(T);
    std::vector<char> ok(T, 1);
    bool have = stride.make((uint64_t)T * C * K);
    if (have) {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          GslMt19937 g;
          if (!rng_.at(o0 + (uint64_t)K * C * t, &g) || !g.canonical_state(st[t].data())) ok[t] = 0;
        });
      for (auto &x : th) x.join();
      for (unsigned t = 0; t < T; ++t) have = have && ok[t];
    }
    if (have) {
      const size_t W = (size_t)T * C, rounds = (E + W - 1) / W;
      std::vector<std::vector<double>> bufs(2 * (size_t)T);
      for (auto &b : bufs) b.resize(C * K);
      auto draw = [&](size_t r, unsigned t) {
        const size_t l0 = r * W + (size_t)t * C;
        if (l0 < E) {
          const size_t cnt = std::min(C, E - l0);
          double *v = bufs[(r & 1) * T + t].data();
          GslMt19937 g(st[t].data(), rng_.seed_state(), o0 + (uint64_t)K * l0);
          for (size_t i = 0; i < cnt; ++i) {        // drawn and normalised while the row sits in L1: one pass over memory
            g.fill_uniform(v + i * K, K);
            normalise(v, i, i + 1);
          }
        }
        stride.apply(st[t].data());
      };
      // Round r's vectors into gamma, keeping every row's additions in link order.  Row x receives its q-side
      // additions (links (p' < x, x)) before its p-side ones (links (x, q)): all links with first endpoint below x come
      // before those with first endpoint x.  So per round: first every q-side addition, rows split over the threads by
      // node range (each scans the round's chunks in order); then the p-side ones, split by CHUNK -- links with the same p
      // are consecutive, so thread t takes the runs of equal p that START in chunk t and follows a run into the next
      // chunks if it straddles a boundary (adding by node range here would leave all p-side work of a round, whose links
      // cover a few thousand consecutive nodes, to one or two threads: 4 of the 5 s this function took at n = 1e6).
      auto chunk_of = [&](size_t r, unsigned t, size_t *l0, size_t *cnt) {
        *l0 = r * W + (size_t)t * C;
        *cnt = *l0 < E ? std::min(C, E - *l0) : 0;
        return bufs[(r & 1) * T + t].data();
      };
      auto gather_q = [&](size_t r, unsigned w) {
        const uint32_t nb = (uint32_t)((uint64_t)n_ * w / T), ne = (uint32_t)((uint64_t)n_ * (w + 1) / T);
        for (unsigned t = 0; t < T; ++t) {
          size_t l0, cnt;
          const double *v = chunk_of(r, t, &l0, &cnt);
          for (size_t i = 0; i < cnt; ++i) {
            const uint32_t q = lq[l0 + i];
            if (q >= nb && q < ne) { double *g = &gamma_[(size_t)q * K]; const double *u = v + i * K; for (size_t k = 0; k < K; ++k) g[k] += u[k]; }
          }
        }
      };
      auto gather_p = [&](size_t r, unsigned t) {
        size_t l0, cnt;
        (void)chunk_of(r, t, &l0, &cnt);
        if (!cnt) return;
        const size_t round_end = std::min(E, (r + 1) * W);
        size_t i = l0;
        if (t > 0) {                                          // the run that came in from the chunk before belongs to its starter
          const uint32_t pin = lp[l0 - 1];
          while (i < l0 + cnt && lp[i] == pin) ++i;
        }
        const size_t own_end = l0 + cnt;
        if (i >= own_end) return;                             // the whole chunk lies inside a run an earlier thread follows
        while (i < round_end && (i < own_end || lp[i] == lp[own_end - 1])) {
          const size_t tt = (i - r * W) / C;                  // the chunk this link sits in
          const double *u = bufs[(r & 1) * T + tt].data() + (i - (r * W + tt * C)) * K;
          double *g = &gamma_[(size_t)lp[i] * K];
          for (size_t k = 0; k < K; ++k) g[k] += u[k];
          ++i;
        }
      };
      {   // prime the pipeline: round 0
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(draw, (size_t)0, t);
        for (auto &x : th) x.join();
      }
      for (size_t r = 0; r < rounds; ++r) {
        std::vector<std::thread> drawing, th;
        // round r + 1 is drawn into the other set of buffers (free since round r - 1 was added) while round r is added
        if (r + 1 < rounds) for (unsigned t = 0; t < T; ++t) drawing.emplace_back(draw, r + 1, t);
        for (unsigned w = 0; w < T; ++w) th.emplace_back(gather_q, r, w);
        for (auto &x : th) x.join();
        th.clear();
        for (unsigned t = 0; t < T; ++t) th.emplace_back(gather_p, r, t);
        for (auto &x : th) x.join();
        for (auto &x : drawing) x.join();
      }
      GslMt19937 after;
      if (!rng_.at(o0 + (uint64_t)E * K, &after)) { fprintf(stderr, "error: random stream jump failed\n"); exit(-1); }
      rng_ = after;
      return;
    }
  }

  buf[0].resize(std::min(C, std::max<size_t>(E, 1)) * K);
  if (T > 1) buf[1].resize(buf[0].size());
  std::thread pending;
  int cur = 0;
  for (size_t l0 = 0; l0 < E; l0 += C) {
    const size_t cnt = std::min(C, E - l0);
    double *v = buf[cur].data();
    rng_.fill_uniform(v, cnt * K);
    if (pending.joinable()) pending.join();
    if (T == 1) {
      process(v, l0, cnt);
    } else {
      pending = std::thread(process, v, l0, cnt);
      cur ^= 1;
    }
  }
  if (pending.joinable()) pending.join();
}

int LinkSampling::init_lambda() {                         // src/linksampling.cc:364-372
  for (uint32_t k = 0; k < k_; ++k) {
    lambda_[2 * k] = env_.eta0;
    lambda_[2 * k + 1] = env_.eta1;
  }
  return 0;
}

// gamma.txt: 2 leading columns skipped; lambda.txt: 1 (src/linksampling.cc:1266-1352).
// The path is gamma_location + "gamma.txt" with no separator added (src/env.hh:277-282).
int LinkSampling::load_model() {
  auto parse = [&](const std::string &path, uint32_t skip, uint32_t cols, uint32_t rows,
                   double *out) -> int {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) { fprintf(stderr, "no %s found\n", path.c_str()); return -1; }
    std::vector<char> line(32 * (size_t)k_ + 64);
    uint32_t r = 0;
    while (fgets(line.data(), (int)line.size(), f)) {
      if (r >= rows) { r++; break; }
      char *p = line.data();
      uint32_t c = 0;
      for (;;) {
        char *q = nullptr;
        double d = strtod(p, &q);
        if (q == p) break;
        p = q;
        if (c >= skip && c - skip < cols) out[(size_t)r * cols + (c - skip)] = d;
        c++;
      }
      if (c < skip + cols) { fprintf(stderr, "error parsing %s\n", path.c_str()); fclose(f); return -1; }
      r++;
    }
    fclose(f);
    if (r != rows) { fprintf(stderr, "%s: expected %u rows, read %u\n", path.c_str(), rows, r); return -1; }
    return 0;
  };
  if (parse(env_.gamma_location + "gamma.txt", 2, k_, n_, gamma_.data()) < 0) return -1;
  if (parse(env_.gamma_location + "lambda.txt", 1, 2, k_, lambda_.data()) < 0) return -1;
  return 0;
}

// assign_training_links, src/linksampling.cc:493-523.  (p,q) with p<q in
// (p, adjacency order) order; _training_links[] = 2 * degree is derived by the
// device library from this list.
const std::vector<uint32_t> &LinkSampling::training_links() {
  if (links_done_) return links_;
  links_.clear();
  for (uint32_t p = 0; p < n_; ++p)
    for (uint32_t q : network_.get_edges(p)) {
      if (p >= q) continue;
      if (!env_.accuracy && !edge_ok(Edge(p, q))) continue;   // held out
      links_.push_back(p);
      links_.push_back(q);
    }
  links_done_ = true;
  return links_;
}

// ----------------------------------------------------------------------- outputs
void LinkSampling::write_validation_row(const double *r, FILE *f) const {   // :996-1002
  fprintf(f, "%d\t%d\t%.9f\t%d\t%.9f\t%d\t%.9f\t%d\t%.9f\t%.9f\t%.9f\n", (int)r[0], duration(), r[1],
          (int)r[2], r[3], (int)r[4], r[5], (int)r[6], r[7], r[8], r[9]);
  fflush(f);
}

void LinkSampling::write_max(const double *r, int why, double max_h) const {   // :1030-1034
  FILE *f = fopen(Env::file_str("/max.txt").c_str(), "w");
  if (!f) return;
  fprintf(f, "%d\t%d\t%.5f\t%.5f\t%.5f\t%d\n", (int)r[0], duration(), r[9], 0.0, max_h, why);
  fclose(f);
}

namespace {
// Rows of a text matrix (gamma.txt, groups.txt: n rows of k numbers -- 5 GB and 3 GB at n = 1e6, k = 512) formatted by
// worker threads in blocks of rows; the numbers go through append_fixed (fixedfmt.hh: printf's own bytes, six times
// faster).  The sequential fprintf loop was 101 s of a 124 s run at that size (tools/cli_config5.py).  A wave of T blocks
// is formatted while the previous wave goes to the file -- not through write(): buffered writes to ONE file are
// serialised by the inode lock (8 threads of pwrite: no faster than one), so the wave's range of the file is mapped and
// the T blocks are copied into the mapping by T threads (page-cache pages are faulted in concurrently).  Small matrices
// keep one thread and plain writes; a file system that cannot map the file falls back to pwrite.
template <class RowFn>
void write_rows(const std::string &path, const char *what, uint32_t n, uint32_t k, size_t bytes_per_number, RowFn row) {
  const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) {
    printf("cannot open %s file:%s\n", what, strerror(errno));
    exit(-1);
  }
  unsigned T = std::min(usable_cpus(), 64u);   // formatting is the bound where the file system is memory-fast (measured: 14 GB/s of write())
  if ((uint64_t)n * k < (1u << 22)) T = 1;
  if (const char *e = getenv("SVINET_WRITE_THREADS")) T = (unsigned)std::max(1, atoi(e));   // (tests: the threaded path on small files)
  if (getenv("SVINET_TRACE_LOOP")) fprintf(stderr, "[final] %s: %u formatting threads (usable cpus %u)\n", what, T, usable_cpus());
  // how a wave reaches the file: plain positional writes from this thread -- 14 GB/s on the memory-backed file system of
  // the GPU box -- unless they turn out slow (0.2 GB/s on a disk-backed VM, where copies into a mapping from T threads
  // reached 1.6 GB/s): then the rest of the file goes through mappings.  SVINET_WRITE_MMAP=0 / 1 fixes the choice.
  int use_map = 0;   // 0 undecided (writes, timed), 1 mappings, -1 writes for good
  if (const char *e = getenv("SVINET_WRITE_MMAP")) use_map = atoi(e) ? 1 : -1;
  double write_s = 0.0;
  uint64_t write_bytes = 0;
  const uint32_t B = (uint32_t)std::max<size_t>(16, ((size_t)8 << 20) / ((size_t)k * bytes_per_number + 24));   // ~8 MB of text per block
  std::vector<std::string> buf[2] = {std::vector<std::string>(T), std::vector<std::string>(T)};
  uint64_t off = 0;
  auto write_all = [&](const char *p, size_t len, uint64_t at) {
    while (len) {
      const ssize_t w = pwrite(fd, p, len, (off_t)at);
      if (w <= 0) { printf("cannot write %s file:%s\n", what, strerror(errno)); exit(-1); }
      p += w; len -= (size_t)w; at += (uint64_t)w;
    }
  };
  auto flush = [&](std::vector<std::string> &bs) {
    uint64_t total = 0;
    std::vector<uint64_t> at(bs.size());
    for (size_t t = 0; t < bs.size(); ++t) { at[t] = off + total; total += bs[t].size(); }
    if (!total) return;
    char *m = (char *)MAP_FAILED;
    const uint64_t a0 = off & ~(uint64_t)4095;
    // (posix_fallocate, not ftruncate: a full disk or a quota must surface HERE as an error code -- stores into a mapping of
    //  a sparse range would turn it into SIGBUS; a file system without fallocate falls back to positional writes)
    if (use_map == 1 && T > 1 && posix_fallocate(fd, (off_t)off, (off_t)total) == 0)
      m = (char *)mmap(nullptr, (size_t)(off + total - a0), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)a0);
    if (m == (char *)MAP_FAILED) {
      if (use_map == 1) use_map = -1;          // this file system cannot map the file
      const double t0 = now_s();
      for (size_t t = 0; t < bs.size(); ++t) write_all(bs[t].data(), bs[t].size(), at[t]);
      write_s += now_s() - t0;
      write_bytes += total;
      if (use_map == 0 && write_bytes >= ((uint64_t)128 << 20)) use_map = (write_bytes / write_s < 1.0e9 && T > 1) ? 1 : -1;
    } else {
      std::vector<std::thread> th;
      for (size_t t = 0; t < bs.size(); ++t)
        if (!bs[t].empty()) th.emplace_back([&, t] { memcpy(m + (at[t] - a0), bs[t].data(), bs[t].size()); });
      for (auto &x : th) x.join();
      munmap(m, (size_t)(off + total - a0));
    }
    off += total;
    for (std::string &b : bs) b.clear();
  };
  int cur = 0;
  double t_fmt = 0.0, t_wait = 0.0, t_flush = 0.0;
  for (uint32_t base = 0; base < n; base += T * B, cur ^= 1) {
    auto work = [&, base, cur](unsigned t) {
      std::string &o = buf[cur][t];
      const uint32_t b = (uint32_t)std::min<uint64_t>(n, (uint64_t)base + (uint64_t)t * B), e = (uint32_t)std::min<uint64_t>(n, (uint64_t)b + B);
      if (e > b && o.capacity() == 0) o.reserve((size_t)(e - b) * ((size_t)k * bytes_per_number + 24));
      RowOut out(o);                         // (fixedfmt.hh: the thread's own scratch in front of the block's string)
      for (uint32_t i = b; i < e; ++i) row(i, out);
    };
    if (T == 1) { work(0); flush(buf[cur]); continue; }
    std::vector<std::thread> th;
    const double w0 = now_s();
    for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t);
    flush(buf[cur ^ 1]);                      // the previous wave goes to the file while this one is formatted
    const double w1 = now_s();
    for (auto &x : th) x.join();
    const double w2 = now_s();
    t_flush += w1 - w0; t_wait += w2 - w1; t_fmt += w2 - w0;
  }
  flush(buf[cur ^ 1]);
  // a deferred write error (mapped pages, delayed allocation, NFS) is reported at fsync / close: the reference's fclose
  // would have lost it too, but a silently short gamma.txt is the one thing this writer must not leave behind
  if (fsync(fd) != 0 && errno != EINVAL && errno != EROFS) { printf("cannot write %s file:%s\n", what, strerror(errno)); exit(-1); }
  if (close(fd) != 0) { printf("cannot write %s file:%s\n", what, strerror(errno)); exit(-1); }
  if (getenv("SVINET_TRACE_LOOP")) fprintf(stderr, "[final] %s: waves %.3f s, of which writing the previous wave %.3f s, then waiting for the formatters %.3f s\n", what, t_fmt, t_flush, t_wait);
}
}  // namespace

void LinkSampling::save_model() {                          // src/linksampling.cc:804-837
  // the state comes back into the host copies the constructor filled (n k doubles: allocating and zeroing a second 4 GB
  // array at n = 1e6, k = 512 cost a second); only the layouts that have to be re-ordered take a scratch array
  const bool direct = !env_.kshard && dev_of_.empty();
  DVec g(direct ? 0 : (size_t)n_ * k_);
  std::vector<double> l(2 * (size_t)k_);
  const double tf = now_s();
  if (direct) {
    if (gamma_.size() != (size_t)n_ * k_) gamma_.resize((size_t)n_ * k_);   // (init_gamma2 ran on the device: first host copy, no zero fill)
    if (svils_get_state(h_, gamma_.data(), l.data(), nullptr)) die_svils("svils_get_state");
    g.swap(gamma_);
  } else if (env_.kshard) fetch_state_ksharded(g, l);
  else if (svils_get_state(h_, g.data(), l.data(), nullptr)) die_svils("svils_get_state");
  if (getenv("SVINET_TRACE_LOOP")) fprintf(stderr, "[final] state fetched in %.3f s\n", now_s() - tf);
  if (!dev_of_.empty()) {   // back to sequence-id order
    DVec t((size_t)n_ * k_);
    for (uint32_t i = 0; i < n_; ++i)
      std::copy(&g[(size_t)dev_of_[i] * k_], &g[(size_t)(dev_of_[i] + 1) * k_], &t[(size_t)i * k_]);
    g.swap(t);
  }
  const std::vector<uint32_t> &s2i = network_.seq2id();
  write_rows(Env::file_str("/gamma.txt"), "gamma", n_, k_, 10, [&](uint32_t i, RowOut &o) {     // "%d\t%d\t" then "%.5f\t" ... "%.5f\n"
    o.integer((long)(int)i, '\t');
    o.integer((long)(int)s2i[i], '\t');
    const double *row = &g[(size_t)i * k_];
    for (uint32_t k = 0; k < k_; ++k) o.fixed<5>(row[k], k == k_ - 1 ? '\n' : '\t');   // %.5f, byte for byte (fixedfmt.hh)
  });
  FILE *lf = open_or_die(Env::file_str("/lambda.txt"), "lambda");
  for (uint32_t k = 0; k < k_; ++k) fprintf(lf, "%d\t%.5f\t%.5f\n", k, l[2 * k], l[2 * k + 1]);
  fclose(lf);
  gamma_.swap(g);
  lambda_.swap(l);
}

// -kshard: every rank holds the columns [k0_, k1_); the slices are padded to the widest one, gathered
// (svils_comm_allgather_host) and put back side by side.  Collective: every rank calls it.
void LinkSampling::fetch_state_ksharded(DVec &g, std::vector<double> &l) {
  const uint32_t G = (uint32_t)env_.gpus, w = k1_ - k0_, wmax = (k_ + G - 1) / G;
  std::vector<double> mine((size_t)n_ * w + 2 * (size_t)w), send(((size_t)n_ + 2) * wmax, 0.0);
  if (svils_get_state(h_, mine.data(), mine.data() + (size_t)n_ * w, nullptr)) die_svils("svils_get_state");
  for (uint32_t i = 0; i < n_; ++i) std::copy(&mine[(size_t)i * w], &mine[(size_t)i * w] + w, &send[(size_t)i * wmax]);
  for (uint32_t c = 0; c < w; ++c) {   // lambda as two rows of wmax behind the gamma rows
    send[(size_t)n_ * wmax + c] = mine[(size_t)n_ * w + 2 * c];
    send[((size_t)n_ + 1) * wmax + c] = mine[(size_t)n_ * w + 2 * c + 1];
  }
  std::vector<double> all(send.size() * G);
  if (svils_comm_allgather_host(h_, send.data(), all.data(), send.size() * sizeof(double))) die_svils("svils_comm_allgather_host");
  for (uint32_t r = 0; r < G; ++r) {
    const uint32_t a = (uint32_t)((uint64_t)k_ * r / G), b = (uint32_t)((uint64_t)k_ * (r + 1) / G);
    const double *src = &all[(size_t)r * send.size()];
    for (uint32_t i = 0; i < n_; ++i) std::copy(src + (size_t)i * wmax, src + (size_t)i * wmax + (b - a), &g[(size_t)i * k_ + a]);
    for (uint32_t c = a; c < b; ++c) {
      l[2 * (size_t)c] = src[(size_t)n_ * wmax + (c - a)];
      l[2 * (size_t)c + 1] = src[((size_t)n_ + 1) * wmax + (c - a)];
    }
  }
}

void LinkSampling::fetch_communities_ksharded() {
  const uint32_t G = (uint32_t)env_.gpus, w = k1_ - k0_, wmax = (k_ + G - 1) / G;
  std::vector<uint8_t> mine((size_t)n_ * w), send((size_t)n_ * wmax, 0), all((size_t)n_ * wmax * G);
  if (svils_get_communities(h_, mine.data())) die_svils("svils_get_communities");
  for (uint32_t i = 0; i < n_; ++i) std::copy(&mine[(size_t)i * w], &mine[(size_t)i * w] + w, &send[(size_t)i * wmax]);
  if (svils_comm_allgather_host(h_, send.data(), all.data(), send.size())) die_svils("svils_comm_allgather_host");
  if (!env_.write_files) return;   // the other ranks only take part in the collective; rank 0 writes
  // the ranks' column slices -> (node, community) pairs in node order
  tags_.clear();
  for (uint32_t i = 0; i < n_; ++i)
    for (uint32_t r = 0; r < G; ++r) {
      const uint32_t a = (uint32_t)((uint64_t)k_ * r / G), b = (uint32_t)((uint64_t)k_ * (r + 1) / G);
      const uint8_t *src = &all[(size_t)r * send.size() + (size_t)i * wmax];
      for (uint32_t c = 0; c < b - a; ++c)
        if (src[c]) { tags_.push_back(i); tags_.push_back(a + c); }
    }
}

void LinkSampling::write_groups() {                        // src/linksampling.cc:1452-1476
  const std::vector<uint32_t> &s2i = network_.seq2id();
  write_rows(Env::file_str("/groups.txt"), "groups", n_, k_, 6, [&](uint32_t i, RowOut &o) {
    const double *g = &gamma_[(size_t)i * k_];
    double s = .0;
    for (uint32_t k = 0; k < k_; ++k) s += g[k];
    o.integer((long)(int)i, '\t');
    o.integer((long)(int)s2i[i], '\t');
    for (uint32_t k = 0; k < k_; ++k) o.fixed<3>(g[k] / s, k == k_ - 1 ? '\n' : '\t');   // %.3f
  });
}

void LinkSampling::log_communities() {                     // :839-852, :882-917
  if (!env_.kshard) {   // -kshard: fetch_communities_ksharded() has filled tags_ on rank 0
    uint64_t nt = 0;
    if (svils_get_community_tags(h_, nullptr, 0, &nt)) die_svils("svils_get_community_tags");
    tags_.resize(2 * (size_t)nt);
    if (svils_get_community_tags(h_, tags_.data(), nt, &nt)) die_svils("svils_get_community_tags");
  }
  write_communities_file();
}

void LinkSampling::do_on_stop() {                          // src/linksampling.cc:792-802
  const double t0 = now_s();
  do_on_stop_impl();
  timing_.final_files_s += now_s() - t0;
}

void LinkSampling::do_on_stop_impl() {
  const double tf0 = now_s();
  const bool trace = getenv("SVINET_TRACE_LOOP") != nullptr;
  auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[final] +%.3f s: %s\n", now_s() - tf0, what); };
  // -gpus N: every rank takes part in the gather of the community bitmasks, rank 0 writes
  if (env_.kshard) {
    fetch_communities_ksharded();
    if (!env_.write_files) {   // the other ranks take part in the gather of the model, rank 0 writes it
      DVec g((size_t)n_ * k_);
      std::vector<double> l(2 * (size_t)k_);
      fetch_state_ksharded(g, l);
      return;
    }
  } else if (env_.sharded && svils_gather_communities(h_)) die_svils("svils_gather_communities");
  if (!env_.write_files) return;
  log_communities();
  mark("communities.txt");
  save_model();
  mark("gamma.txt, lambda.txt (state fetched from the device first)");
  write_groups();
  mark("groups.txt");
}

// true when the batch reached a report (the reference's `_iter % reportfreq == 0` block, :777-785): new likelihood
// rows.  The row count lives in the replicated control block, so every rank of a -gpus N run sees the same answer.
bool LinkSampling::fetch_and_log_rows() {
  svils_control c;
  if (svils_get_control(h_, &c)) die_svils("svils_get_control");
  const bool reported = c.rows > rows_logged_ || env_.accuracy || val_sorted_.empty();
  if (c.rows > rows_logged_) {
    std::vector<double> rows((size_t)(c.rows - rows_logged_) * 10);
    const uint32_t cnt = c.rows - rows_logged_;
    if (svils_get_rows(h_, rows_logged_, cnt, rows.data())) die_svils("svils_get_rows");
    const bool with_test = !test_sorted_.empty() && !env_.accuracy;
    std::vector<double> trows;
    if (with_test) {
      trows.resize((size_t)cnt * 10);
      if (svils_get_test_rows(h_, rows_logged_, cnt, trows.data())) die_svils("svils_get_test_rows");
    }
    log_rows(rows.data(), cnt, c.why, c.max_h, with_test ? trows.data() : nullptr, c.stopped ? cnt - 1 : cnt);
    rows_logged_ = c.rows;
  }
  return reported;
}

// LinkSampling::infer, src/linksampling.cc:556-790.  The loop body lives on
// the device; the host decides how many sweeps to enqueue, polls the
// device-resident control block, and writes the per-report files.
int LinkSampling::infer() {
  if (!h_) {
    fprintf(stderr, "error: LinkSampling::infer() without a device (attach_device=false)\n");
    exit(-1);
  }
  const double t0 = now_s();
  send_graph();
  if (svils_synchronize(h_)) die_svils("svils_synchronize");
  timing_.graph_upload_s = now_s() - t0;
  return sweep_loop();
}

void LinkSampling::send_graph() {
  if (!graph_sent_) {
    const std::vector<uint32_t> &L = training_links();
    if (dev_of_.empty()) {
      if (svils_set_graph(h_, L.data(), L.size() / 2)) die_svils("svils_set_graph");
    } else {
      // relabelled links, p < q, sorted by (p, q): packed as p << 32 | q for the sort, then flat [L][2]
      std::vector<uint64_t> R(L.size() / 2);
      for (size_t i = 0; i < R.size(); ++i) {
        const uint32_t a = dev_of_[L[2 * i]], b = dev_of_[L[2 * i + 1]];
        R[i] = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
      }
      std::sort(R.begin(), R.end());
      std::vector<uint32_t> flat(2 * R.size() + 2, 0);
      for (size_t i = 0; i < R.size(); ++i) { flat[2 * i] = (uint32_t)(R[i] >> 32); flat[2 * i + 1] = (uint32_t)R[i]; }
      if (svils_set_graph(h_, flat.data(), R.size())) die_svils("svils_set_graph");
    }
    graph_sent_ = true;
  }
}

// communities.txt of a report (src/linksampling.cc:839-852,882-917) from tags already on the host
// rank of every node in the order of the external ids (communities.txt lists members by ascending external id): once per
// run, in the constructor when files are written -- not inside the first report of the sweep loop
void LinkSampling::rank_external_ids() {
  const std::vector<uint32_t> &s2i = network_.seq2id();
  std::vector<uint32_t> order(n_);
  for (uint32_t i = 0; i < n_; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return s2i[a] < s2i[b]; });
  ext_rank_.resize(n_);
  for (uint32_t r = 0; r < n_; ++r) ext_rank_[order[r]] = r;
}

void LinkSampling::write_communities_file() {
  const std::vector<uint32_t> &s2i = network_.seq2id();
  // one pass over the (device row, community) pairs -- a node tags one or two communities, so this is O(n), not the
  // O(n k) of a pass over the tag matrix (n = 1e6, k = 512: 512 MB per report) -- then every community sorted by external id
  // ... visited in the order of the external ids (a counting sort of the pairs by the node's rank in that order, the ranks
  // computed once per run), so that every community's member list comes out sorted and no per-report sort is needed
  if (ext_rank_.empty()) rank_external_ids();
  const size_t nt = tags_.size() / 2;
  std::vector<uint32_t> start((size_t)n_ + 1, 0), slot(2 * nt);
  auto seq_of_tag = [&](size_t i) { return dev_of_.empty() ? tags_[2 * i] : seq_of_[tags_[2 * i]]; };   // device rows back to sequence ids
  for (size_t i = 0; i < nt; ++i) start[ext_rank_[seq_of_tag(i)] + 1]++;
  for (uint32_t r = 0; r < n_; ++r) start[r + 1] += start[r];
  for (size_t i = 0; i < nt; ++i) {
    const uint32_t p = seq_of_tag(i);
    const uint32_t at = start[ext_rank_[p]]++;
    slot[2 * (size_t)at] = s2i[p];
    slot[2 * (size_t)at + 1] = tags_[2 * i + 1];
  }
  std::vector<std::vector<uint32_t>> ids(k_);
  for (size_t i = 0; i < nt; ++i) ids[slot[2 * i + 1]].push_back(slot[2 * i]);
  std::string out;
  Cover found;
  char buf[16];
  for (uint32_t c = 0; c < k_; ++c) {
    if (ids[c].empty()) continue;            // empty communities have no map entry => no line
    for (uint32_t id : ids[c]) {             // "%d " per member: the id goes through (int), as printf("%d ") sees it
      char *e = buf + sizeof buf;
      char *b = e;
      *--b = ' ';
      const int32_t sv = (int32_t)id;
      uint32_t v = sv < 0 ? 0u - (uint32_t)sv : (uint32_t)sv;
      do { *--b = (char)('0' + v % 10); v /= 10; } while (v);
      if (sv < 0) *--b = '-';
      out.append(b, (size_t)(e - b));
    }
    out.push_back('\n');
    if (env_.nmi) found.push_back(ids[c]);
  }
  FILE *f = open_or_die(Env::file_str("/communities.txt"), "communities");
  fwrite(out.data(), 1, out.size(), f);
  fclose(f);
  if (env_.nmi && !ground_truth_.empty()) {
    // the reference runs `/usr/local/bin/mutual ground_truth.txt communities.txt >> mutual.txt` here (:843-851)
    FILE *mf = fopen(Env::file_str("/mutual.txt").c_str(), "a");
    if (mf) {
      fprintf(mf, "mutual3:\t%g\n", lfk_nmi(ground_truth_, found));
      fclose(mf);
    }
  }
}

// validation.txt / test.txt / max.txt of `count` reports.  test: [ntest][10] rows of the test set (svils_set_test), or
// null: test_likelihood over the empty test map prints 0/0 ratios (:1147-1182).  ntest < count only when the last
// report is the one that ends the run: the reference leaves through do_on_stop + exit before test_likelihood (:777-781).
void LinkSampling::log_rows(const double *rows, uint32_t count, int why, double max_h, const double *test, uint32_t ntest) {
  if (!vf_ || !count) return;
  for (uint32_t i = 0; i < count; ++i) {
    write_validation_row(&rows[(size_t)i * 10], vf_);
    if (i < ntest) {
      if (test) write_validation_row(&test[(size_t)i * 10], tf_);
      else fprintf(tf_, "%d\t%d\t-nan\t0\t-nan\t0\t-nan\t0\t-nan\t-nan\t-nan\n", (int)rows[(size_t)i * 10], duration());
      fflush(tf_);
    }
  }
  write_max(&rows[(size_t)(count - 1) * 10], why, max_h);
}

// The loop of infer() with the report block of every sweep (:777-786) taken OFF the device's critical path: the host
// enqueues chunks of sweeps (hipGraph replay inside the library) each followed by a report snapshot
// (svils_report_enqueue), keeps up to three chunks in flight and writes the files of report t while the device is
// already past it.  What reaches the files is what the synchronous loop writes: every likelihood row in order,
// max.txt, communities.txt -- except that a communities.txt which the NEXT landed report would overwrite at once is
// not written (the file is rewritten from scratch by every report; only its latest content is observable), and the
// final files always come from do_on_stop().  -sweep-batch B fixes the chunk at B sweeps; the default (0) is 1, 2, 4, 8,
// then 16 sweeps per report.
int LinkSampling::sweep_loop_pipelined() {
  const uint64_t nlinks = links_.size() / 2;
  svils_control c;
  if (svils_get_control(h_, &c)) die_svils("svils_get_control");
  if (env_.max_iterations == 1 && !c.write_comm) {              // :581-582
    c.write_comm = 1;
    if (svils_set_control(h_, &c)) die_svils("svils_set_control");
  }
  struct Flight { int ticket; uint32_t row_count; bool with_comm; };
  std::vector<Flight> flight;             // oldest first
  uint32_t issued_iter = c.iter;          // _iter after the sweeps enqueued so far (if nothing stops them)
  uint32_t rows_issued = rows_logged_;
  uint32_t chunk = env_.sweep_batch ? env_.sweep_batch : 1;
  // -nmi scores the communities of EVERY report (one mutual.txt line per report, :843-851): one report per rfreq sweeps
  const bool fixed_chunk = env_.sweep_batch != 0 || env_.nmi;
  const uint32_t rf = std::max<uint32_t>(1, env_.reportfreq);
  if (env_.nmi && !env_.sweep_batch) chunk = rf;
  const bool always_report = env_.accuracy || val_sorted_.empty();
  std::vector<double> rows((size_t)SVILS_REPORT_MAX_ROWS * 10), trows((size_t)SVILS_REPORT_MAX_ROWS * 10);
  bool quit_max = false;
  timing_.pipelined = true;
  {   // the report slots (device staging + pinned host memory, ~1 ms of allocations) exist before the first sweep
    int t[SVILS_REPORT_SLOTS];
    for (int i = 0; i < SVILS_REPORT_SLOTS; ++i)
      if (svils_report_enqueue(h_, 0, 0, 1, &t[i])) die_svils("svils_report_enqueue");
    for (int i = 0; i < SVILS_REPORT_SLOTS; ++i) {
      svils_control cc;
      uint32_t hv = 0;
      if (svils_report_fetch(h_, t[i], &cc, rows.data(), &hv, nullptr)) die_svils("svils_report_fetch");
    }
  }
  timing_.sweeps_t0 = now_s();
  const bool trace = getenv("SVINET_TRACE_LOOP") != nullptr;
  // ---- keep the device busy: up to SVILS_REPORT_SLOTS - 1 chunks ahead of the host
  auto issue_more = [&]() {
    while (!quit_max && flight.size() + 1 < (size_t)SVILS_REPORT_SLOTS) {
      uint32_t batch = std::min<uint32_t>(chunk, (uint32_t)SVILS_REPORT_MAX_ROWS * rf);
      if (env_.max_iterations) {
        if (issued_iter > env_.max_iterations) { quit_max = true; return; }      // :573-579, seen at issue time
        batch = std::min<uint32_t>(batch, env_.max_iterations + 1 - issued_iter);
      }
      printf("\riteration %d: processing %d links", issued_iter, (int)nlinks);
      fflush(stdout);
      const double tr0 = trace ? now_s() : 0.0;
      if (svils_sweep(h_, batch)) die_svils("svils_sweep");
      if (trace) fprintf(stderr, "[loop] +%.3f ms: issued %u sweeps from iter %u (host %.3f ms)\n", (now_s() - timing_.sweeps_t0) * 1e3, batch, issued_iter, (now_s() - tr0) * 1e3);
      // the device records a row for every sweep whose pre-increment _iter is a multiple of rf (k_tail; _iter starts at
      // 0, so sweep 0 records row 0): the multiples of rf in [iter, iter + batch)
      const uint32_t new_rows = (issued_iter + batch + rf - 1) / rf - (issued_iter + rf - 1) / rf;
      issued_iter += batch;
      Flight f{-1, new_rows, new_rows > 0 || always_report};
      if (svils_report_enqueue(h_, rows_issued, new_rows, f.with_comm ? 1 : 0, &f.ticket)) die_svils("svils_report_enqueue");
      rows_issued += new_rows;
      flight.push_back(f);
      timing_.chunks++;
      // automatic chunks: 1, 2, 4, 8, 16, 16, ... sweeps per report -- the first reports come at once, later ones every 16
      // sweeps (a report costs the host ~0.5 ms of file writing, a sweep the device tens of microseconds)
      if (!fixed_chunk && chunk < 16) chunk *= 2;
    }
  };
  for (;;) {
    issue_more();
    if (flight.empty()) break;              // everything issued and reported: -max-iterations reached
    // ---- the oldest report: blocks until it has landed
    const Flight f = flight.front();
    flight.erase(flight.begin());
    uint32_t have = 0;
    // is a newer report already on the host?  then this one's communities.txt would be overwritten at once
    // (-nmi scores every report's communities -- one mutual.txt line per report, :843-851 -- so nothing is skipped then)
    const bool superseded = !env_.nmi && !flight.empty() && flight.front().with_comm && svils_report_ready(h_, flight.front().ticket) == 1;
    const bool want_comm = f.with_comm && env_.write_files && !superseded;
    uint32_t have_t = 0;
    const bool with_test = !test_sorted_.empty() && !env_.accuracy;
    if (with_test && svils_report_test_rows(h_, f.ticket, trows.data(), &have_t)) die_svils("svils_report_test_rows");
    if (want_comm) {
      uint64_t nt = 0;
      if (svils_report_tag_count(h_, f.ticket, &nt)) die_svils("svils_report_tag_count");
      tags_.resize(2 * (size_t)nt);
      if (svils_report_fetch_tags(h_, f.ticket, &c, rows.data(), &have, tags_.data(), nt, &nt)) die_svils("svils_report_fetch_tags");
    } else if (svils_report_fetch(h_, f.ticket, &c, rows.data(), &have, nullptr)) die_svils("svils_report_fetch");
    // the slot is free again and the run goes on: the next chunk is enqueued BEFORE this report's files are written -- a
    // report costs the host 0.3 - 1 ms of file work, the early chunks only tens of microseconds of device time, and a
    // device that waits for the host to finish writing is what made the default run 1.7x the library sweep
    if (!c.stopped) issue_more();
    const double t0 = now_s();
    if (trace) fprintf(stderr, "[loop] +%.3f ms: report landed: iter %u rows %u stopped %d (%zu in flight)\n", (t0 - timing_.sweeps_t0) * 1e3, c.iter, have, c.stopped, flight.size());
    // (without a test set every report that does not end the run still gets its row of 0/0 ratios)
    if (!with_test) have_t = (c.stopped && have && rows_logged_ + have == c.rows) ? have - 1 : have;
    log_rows(rows.data(), have, c.why, c.max_h, with_test ? trows.data() : nullptr, have_t);
    rows_logged_ += have;
    if (!c.stopped && want_comm) { write_communities_file(); timing_.communities_written++; }
    timing_.reports++;
    timing_.report_host_s += now_s() - t0;
    if (c.stopped) {                                              // :1044-1048; the sweeps behind the stop were no-ops
      // The chunks still in flight behind the stop are launches that return at once (the state is frozen where the
      // reference's do_on_stop() saves it) and their reports repeat this one: nobody waits for them -- the library's
      // getters read a state whose stop the host has seen without synchronising with the stream (svils.h, "After the stop").
      // every recorded row reaches the files: whatever the landed reports did not carry (none, if the row bookkeeping
      // above is right) is read from the ring before the final files are written
      if (c.rows > rows_logged_) fetch_and_log_rows();
      timing_.sweeps_t1 = now_s();
      if (trace) fprintf(stderr, "[loop] +%.3f ms: drained\n", (timing_.sweeps_t1 - timing_.sweeps_t0) * 1e3);
      timing_.sweeps = c.sweeps_done;
      do_on_stop();
      return 1;
    }
    if (env_.terminate) {                                         // :763-766 (SIGTERM: save the model and go on)
      do_on_stop();                                               // synchronises; the reports in flight stay valid
      env_.terminate = 0;
    }
  }
  timing_.sweeps_t1 = now_s();
  timing_.sweeps = c.sweeps_done;
  printf("+ Quitting: reached max iterations.\n");
  Env::plog("maxiterations reached", true);
  env_.terminate = 1;
  do_on_stop();
  return 0;
}

int LinkSampling::sweep_loop() {
  if (pipelined_reports()) return sweep_loop_pipelined();
  const uint64_t nlinks = links_.size() / 2;
  svils_control c;
  if (svils_get_control(h_, &c)) die_svils("svils_get_control");
  timing_.sweeps_t0 = now_s();
  for (;;) {
    if (env_.max_iterations && c.iter > env_.max_iterations) {     // :573-579
      timing_.sweeps_t1 = now_s();
      timing_.sweeps = c.sweeps_done;
      printf("+ Quitting: reached max iterations.\n");
      Env::plog("maxiterations reached", true);
      env_.terminate = 1;
      do_on_stop();
      return 0;
    }
    if (env_.max_iterations == 1 && !c.write_comm) {              // :581-582
      c.write_comm = 1;
      if (svils_set_control(h_, &c)) die_svils("svils_set_control");
    }
    uint32_t batch = env_.sweep_batch ? env_.sweep_batch : 1;
    if (env_.max_iterations) batch = std::min<uint32_t>(batch, env_.max_iterations + 1 - c.iter);
    printf("\riteration %d: processing %d links", c.iter, (int)nlinks);
    fflush(stdout);
    if (env_.minibatch && env_.kshard) {
      if (svils_step_ksharded(h_, batch)) die_svils("svils_step_ksharded");
    } else if (env_.minibatch && env_.sharded) {
      if (svils_step_sharded(h_, batch)) die_svils("svils_step_sharded");
    } else if (env_.minibatch) {
      if (svils_step(h_, batch)) die_svils("svils_step");
    } else if (env_.kshard) {
      if (svils_sweep_ksharded(h_, batch)) die_svils("svils_sweep_ksharded");
    } else if (env_.sharded) {
      if (svils_sweep_sharded(h_, batch)) die_svils("svils_sweep_sharded");
    } else if (svils_sweep(h_, batch)) {
      die_svils("svils_sweep");
    }
    const bool reported = fetch_and_log_rows();
    if (svils_get_control(h_, &c)) die_svils("svils_get_control");
    if (!c.stopped && reported) {                                 // :777-785 (the control block is replicated: same branch on every rank)
      if (env_.kshard) fetch_communities_ksharded();
      else if (env_.sharded && svils_gather_communities(h_)) die_svils("svils_gather_communities");
      if (env_.write_files) log_communities();
    }
    timing_.reports += reported ? 1 : 0;
    if (c.stopped) {                                              // :1044-1048
      timing_.sweeps_t1 = now_s();
      timing_.sweeps = c.sweeps_done;
      do_on_stop();
      return 1;
    }
    // :763-766 (SIGTERM: save the model and go on).  -gpus N: the signal reaches the ranks at different sweeps, and
    // do_on_stop() is collective there, so the ranks agree first: one byte per rank, gathered at every poll
    // (a blocking H2D + all-gather + D2H + sync: asked for at every 16th poll, not at every sweep -- a signal waits
    // for at most 16 polls, the sweeps pay nothing in between; the poll counter runs in step on all ranks)
    bool term = env_.terminate != 0;
    if (env_.gpus > 1 && (++term_polls_ & 15u) != 0) term = false;
    else if (env_.gpus > 1) {
      const unsigned char mine = term ? 1 : 0;
      std::vector<unsigned char> all((size_t)env_.gpus, 0);
      if (svils_comm_allgather_host(h_, &mine, all.data(), 1)) die_svils("svils_comm_allgather_host");
      term = std::any_of(all.begin(), all.end(), [](unsigned char b) { return b != 0; });
    }
    if (term) {
      do_on_stop();
      env_.terminate = 0;
    }
  }
}

}  // namespace svinet
