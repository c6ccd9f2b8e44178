#include "network.hh"

#include <algorithm>
#include <cstdio>
#include <vector>
#include <string>

#include "env.hh"

namespace svinet {

Network::Network(Env &env) : env_(env), declared_n_(env.n), adj_(env.n) {
  id2seq_.reserve(env.n * 2 + 16);
}

bool Network::id2seq(uint32_t id, uint32_t *seq) const { return id2seq_.find(id, seq); }

// Network::add (src/network.hh:134-148): refuse new ids once n are known
bool Network::intern(uint32_t id, uint32_t *seq) {
  if (id2seq_.find(id, seq)) return true;
  if (seq2id_.size() >= declared_n_) return false;
  *seq = (uint32_t)seq2id_.size();
  id2seq_.emplace(id, *seq);
  seq2id_.push_back(id);
  return true;
}

static inline uint64_t pair_key(uint32_t a, uint32_t b) {
  return a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
}

bool Network::y(uint32_t a, uint32_t b) const { return pair_set_.contains(pair_key(a, b)); }

bool Network::add_line(uint32_t id1, uint32_t id2) {
  uint32_t p, q;
  if (!intern(id1, &p)) return false;   // note: id1 stays interned even if id2 is refused
  if (!intern(id2, &q)) return false;
  if (p == q) return false;
  if (!pair_set_.insert(pair_key(p, q))) return false;  // both directions listed / duplicates
  edges_.push_back(p < q ? Edge(p, q) : Edge(q, p));
  adj_[p].push_back(q);
  adj_[q].push_back(p);
  return true;
}

int Network::read(const std::string &path) {
  const bool chat = env_.write_files;
  if (chat) fprintf(stdout, "+ Reading network from %s\n", path.c_str());
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  int a, b;
  if (env_.strid) {
    char b1[512], b2[512];
    while (fscanf(f, "%511s %511s", b1, b2) == 2) {
      uint32_t id[2];
      const char *tok[2] = {b1, b2};
      for (int i = 0; i < 2; ++i) {
        auto it = str2id_.find(tok[i]);
        if (it == str2id_.end()) it = str2id_.emplace(tok[i], (uint32_t)str2id_.size()).first;
        id[i] = it->second;
      }
      add_line(id[0], id[1]);
    }
  } else {
    // the reference's format string is "%d\t%d\n": any white space separates
    // fields, so CRLF files (example/assort-75-4.txt) parse as well.  The file is parsed from one buffer with the same
    // rules as fscanf("%d %d") -- white space skipped, an optional sign, digits; the loop ends at the first token that
    // is not an integer -- and the link containers are sized from its line count first: 12 M lines took 11 s through
    // fscanf and rehashing sets (tools/cli_config5.py), the device's five sweeps half a second.
    std::string buf;
    {
      char chunk[1 << 16];
      size_t got;
      while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, got);
    }
    size_t lines = 0;
    for (char ch : buf) lines += ch == '\n';
    pair_set_.reserve(lines + 1);
    edges_.reserve(lines + 1);
    const char *c = buf.data(), *end = c + buf.size();
    auto next_int = [&](int *out) {
      while (c < end && (*c == ' ' || (*c >= '\t' && *c <= '\r'))) ++c;      // isspace in the C locale
      const char *t = c;
      bool neg = false;
      if (t < end && (*t == '-' || *t == '+')) { neg = *t == '-'; ++t; }
      if (t >= end || *t < '0' || *t > '9') return false;
      long long v = 0;
      while (t < end && *t >= '0' && *t <= '9') { v = v * 10 + (*t - '0'); if (v > 0x7fffffffLL + 1) v = 0x7fffffffLL + 1; ++t; }
      *out = (int)(neg ? -v : v);
      c = t;
      return true;
    };
    // Pass 1: the numbers.  Pass 2: the reference's loop over the lines (Network::add / y(), src/network.hh:134-175), as
    // a software pipeline -- every line costs three random probes into tables far larger than the caches (two ids, one
    // pair key), so the slots of the lines ahead are prefetched while this one is processed: ids of line i + 16, pair
    // key of line i + 8 (its ids are interned then, in line order: interning never depends on the pair set).  Pass 3:
    // the neighbour lists, sized exactly and filled in link order (= the order the reference pushes them).
    std::vector<int> vals;
    vals.reserve(2 * lines + 2);
    while (next_int(&a) && next_int(&b)) { vals.push_back(a); vals.push_back(b); }
    { std::string().swap(buf); }
    const size_t nl = vals.size() / 2;
    constexpr size_t DA = 16, DB = 8;
    struct Pending { uint32_t p, q; bool ok; };
    Pending ring[DB];
    auto intern_line = [&](size_t i) {
      Pending x{0, 0, false};
      if (i + DA < nl) { id2seq_.prefetch((uint32_t)vals[2 * (i + DA)]); id2seq_.prefetch((uint32_t)vals[2 * (i + DA) + 1]); }
      // id1 stays interned even if id2 is refused (as in add_line)
      if (intern((uint32_t)vals[2 * i], &x.p) && intern((uint32_t)vals[2 * i + 1], &x.q) && x.p != x.q) {
        x.ok = true;
        pair_set_.prefetch(pair_key(x.p, x.q));
      }
      return x;
    };
    for (size_t i = 0; i < std::min(nl, DB); ++i) ring[i] = intern_line(i);
    std::vector<uint32_t> degree(declared_n_, 0);
    for (size_t i = 0; i < nl; ++i) {
      const Pending x = ring[i % DB];
      if (i + DB < nl) ring[i % DB] = intern_line(i + DB);
      if (!x.ok || !pair_set_.insert(pair_key(x.p, x.q))) continue;   // both directions listed / duplicates
      edges_.push_back(x.p < x.q ? Edge(x.p, x.q) : Edge(x.q, x.p));
      degree[x.p]++;
      degree[x.q]++;
      if (chat && ones() % 10000 == 0) {                // the reference's cadence (src/network.cc:94)
        printf("\r+ %d entries", ones());
        fflush(stdout);
      }
    }
    // which endpoint came first on the line decides nothing here: adj[p] gets q and adj[q] gets p at the same moment, so
    // walking the links in order and appending both ways reproduces the reference's lists
    for (uint32_t v = 0; v < declared_n_; ++v) adj_[v].reserve(degree[v]);
    for (const Edge &e : edges_) { adj_[e.first].push_back(e.second); adj_[e.second].push_back(e.first); }
  }
  fclose(f);
  if (env_.strid && env_.write_files) {
    FILE *sf = fopen(Env::file_str("/str2id.txt").c_str(), "w");
    if (sf) {
      for (const auto &kv : str2id_) fprintf(sf, "%s\t%d\n", kv.first.c_str(), kv.second);
      fclose(sf);
    }
    fprintf(stdout, "+ Total nodes read = %d\n", (int)str2id_.size());
  }
  if (chat && singles())
    printf("n = %d, curr_seq = %d\n+ Creating ids for %d single nodes\n", declared_n_, nodes_seen(), singles());
  if (chat) fprintf(stdout, "\n+ Done reading network\n");
  fflush(stdout);
  set_env_variables();
  return 0;
}

void Network::read_pairs(const int32_t *pairs, uint64_t nlines) {
  pair_set_.reserve(nlines);
  edges_.reserve(nlines);
  for (uint64_t i = 0; i < nlines; ++i) add_line((uint32_t)pairs[2 * i], (uint32_t)pairs[2 * i + 1]);
  set_env_variables();
}

void Network::deg_stats(uint32_t &max, double &avg) const {
  max = 0;
  uint32_t s = 0;
  for (uint32_t i = 0; i < declared_n_; ++i) {
    if (deg(i) > max) max = deg(i);
    s += deg(i);
  }
  avg = (double)s / declared_n_;
}

void Network::set_env_variables() {
  // `_env.n * (_env.n - 1) / 2` is evaluated in 32-bit unsigned arithmetic in
  // the reference and wraps for n >= 65537; kept for parity.
  const uint32_t n = env_.n;
  env_.total_pairs = (uint32_t)(n * (n - 1u)) / 2u;
  if ((uint64_t)n * (n - 1) / 2 != env_.total_pairs)
    fprintf(stderr, "warning: n(n-1)/2 wraps in 32 bits (%u nodes): total_pairs=%llu as in the reference\n",
            n, (unsigned long long)env_.total_pairs);
  Env::plog("total pairs", env_.total_pairs);
  env_.ones_prob = (double)ones() / env_.total_pairs;
  env_.zeros_prob = 1 - env_.ones_prob;
  Env::plog("ones_prob", env_.ones_prob);
  Env::plog("zeros_prob", env_.zeros_prob);
  if (env_.eta_type == "fromdata") {
    env_.eta0 = env_.total_pairs * env_.ones_prob / env_.k;
    env_.eta1 = env_.total_pairs * 1.0 / (env_.k * env_.k) - env_.eta0;
    if (env_.eta1 <= 0) env_.eta1 = 1.0;
  } else if (env_.eta_type == "uniform") {
    env_.eta0 = 1;
    env_.eta1 = 1;
  } else if (env_.eta_type == "sparse") {
    env_.eta0 = env_.eta0_sparse;
    env_.eta1 = env_.eta1_sparse;
  } else if (env_.eta_type == "dense") {
    env_.eta0 = env_.eta0_dense;
    env_.eta1 = env_.eta1_dense;
  } else {
    fprintf(stderr, "unknown eta_type %s\n", env_.eta_type.c_str());
    exit(-1);
  }
}

}  // namespace svinet
