#include "mmsbbatch.hh"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace svinet {

namespace {

const double kNegInit = -2147483647.0;     // _max_t/_max_h/_max_v/_prev_h, src/mmsbinfer.cc:33-38
const uint32_t kOnlineIterations = 50;     // src/env.hh:415
const double kMeanChangeThresh = 0.00001;  // src/env.hh:337

FILE *open_or_die(const std::string &path, const char *what, const char *mode = "w") {
  FILE *f = fopen(path.c_str(), mode);
  if (!f) {
    printf("cannot open %s file:%s\n", what, strerror(errno));
    exit(-1);
  }
  return f;
}

// psi(x), x > 0: recurrence up to x >= 6, then the asymptotic series (double accurate)
double digamma(double x) {
  double r = 0;
  while (x < 6) { r -= 1 / x; x += 1; }
  const double f = 1 / (x * x);
  return r + log(x) - 0.5 / x -
         f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
}

}  // namespace

MMSBBatch::MMSBBatch(Env &env, Network &network)
    : env_(env), network_(network), n_(env.n), k_(env.k),
      ones_prob_(env.ones_prob), zeros_prob_(env.zeros_prob),
      rng_((unsigned long)env.seed),
      gamma_((size_t)n_ * k_), gammanext_((size_t)n_ * k_, env.alpha),
      lambda_(2 * (size_t)k_), lambdanext_(2 * (size_t)k_),
      elogpi_((size_t)n_ * k_), elogbeta_(2 * (size_t)k_),
      max_t_(kNegInit), max_h_(kNegInit), max_v_(kNegInit), prev_h_(kNegInit),
      start_time_(time(0)) {
  fprintf(stdout, "+ initialization begin\n+ running inference on %d nodes\n", n_);
  Env::plog("inference n", n_);
  init_heldout();
  printf("+ heldout sets created\n");
  init_gamma();
  for (uint32_t k = 0; k < k_; ++k) {                       // init_lambda, :389-397
    lambda_[2 * k] = lambdanext_[2 * k] = env.eta0;
    lambda_[2 * k + 1] = lambdanext_[2 * k + 1] = env.eta1;
  }
  set_dir_exp();
  if (env_.write_files) {
    // files the reference's constructor opens (:62-166); the ones this path never writes stay empty
    for (const char *f : {"/stats.txt", "/time.txt", "/convergence.txt", "/cmap.txt", "/training.txt",
                          "/training-edges.txt", "/logl.txt", "/modularity.txt"})
      fclose(open_or_die(Env::file_str(f), f + 1));
    hf_ = open_or_die(Env::file_str("/heldout.txt"), "heldout");
    vf_ = open_or_die(Env::file_str("/validation.txt"), "validation");
  }
  Env::plog("network ones", network_.ones());
  Env::plog("network singles", network_.singles());
  heldout_likelihood();
  validation_likelihood(nullptr);
  fprintf(stdout, "+ initialization end\n");
  fflush(stdout);
  start_time_ = time(0);
}

MMSBBatch::~MMSBBatch() {
  if (hf_) fclose(hf_);
  if (vf_) fclose(vf_);
}

// ---------------------------------------------------------------------------
// held-out / validation sets (src/mmsbinfer.cc:205-329, src/mmsbinfer.hh:690-748)
// ---------------------------------------------------------------------------
bool MMSBBatch::edge_ok(const Edge &e, bool heldout_flag, bool stratified, int family) const {
  if (e.first == e.second) return false;
  if (heldout_flag) return true;                            // the held-out draw only rejects self pairs
  if (heldout_map_.count(e) || validation_map_.count(e)) return false;
  if (stratified && (int)network_.y(e.first, e.second) != family) return false;
  return true;
}

void MMSBBatch::get_random_edge(bool heldout_flag, bool stratified, int family, Edge &e) {
  if (!stratified || heldout_flag || family == 0) {
    do {
      uint32_t a = rng_.uniform_int(n_), b = rng_.uniform_int(n_);
      e = a < b ? Edge(a, b) : Edge(b, a);
    } while (!edge_ok(e, heldout_flag, stratified, family));
  } else {
    const std::vector<Edge> &edges = network_.edges();
    do {
      e = edges[rng_.uniform_int(network_.ones())];
    } while (!edge_ok(e, heldout_flag, stratified, family));
  }
}

void MMSBBatch::set_sample(int s, bool heldout) {
  if (env_.accuracy) return;
  std::map<Edge, bool> &map = heldout ? heldout_map_ : validation_map_;
  std::vector<uint32_t> &list = heldout ? heldout_edges_ : validation_edges_;
  int c0 = 0, c1 = 0;
  const int p = s / 2;
  while (c0 < p || c1 < p) {
    Edge e;
    if (c0 == p) get_random_edge(false, true, 1, e);        // enough non-links: draw links only
    else get_random_edge(heldout, false, 0, e);
    const bool y = network_.y(e.first, e.second);
    if ((!y && c0 < p) || (y && c1 < p)) {
      (y ? c1 : c0)++;
      list.push_back(e.first);
      list.push_back(e.second);
      map[e] = true;
    }
  }
}

std::string MMSBBatch::edgelist_s(const std::vector<uint32_t> &pairs) const {
  std::ostringstream sa;
  const std::vector<uint32_t> &s2i = network_.seq2id();
  for (size_t i = 0; i + 1 < pairs.size(); i += 2) sa << s2i[pairs[i]] << "\t" << s2i[pairs[i + 1]] << "\n";
  return sa.str();
}

void MMSBBatch::init_heldout() {
  const int s = (int)(env_.heldout_ratio * network_.ones());
  set_sample(s, true);
  set_sample(s, false);
  Env::plog("heldout ratio", env_.heldout_ratio);
  Env::plog("heldout edges (1s and 0s)", (uint32_t)heldout_map_.size());
  if (!env_.write_files) return;
  FILE *f = open_or_die(Env::file_str("/heldout-edges.txt"), "heldout edges");
  fprintf(f, "%s\n", edgelist_s(heldout_edges_).c_str());
  fclose(f);
  f = open_or_die(Env::file_str("/validation-edges.txt"), "validation edges");
  fprintf(f, "%s\n", edgelist_s(validation_edges_).c_str());
  fclose(f);
}

// ---------------------------------------------------------------------------
// gamma initialisation: gsl_ran_gamma(r, 100, 1/100) per cell (src/mmsbinfer.cc:372-387).
// GSL's sampler is Marsaglia-Tsang driven by its ziggurat Gaussian, whose tables are internal to
// GSL; this is the same Marsaglia-Tsang recipe over the same MT19937 stream but with a polar
// Box-Muller Gaussian, so the distribution matches and the stream does not (parity unpinned,
// SURVEY 8c).
// ---------------------------------------------------------------------------
double MMSBBatch::ran_gaussian() {
  if (have_spare_) { have_spare_ = false; return spare_; }
  double u, v, s;
  do {
    u = 2 * rng_.uniform() - 1;
    v = 2 * rng_.uniform() - 1;
    s = u * u + v * v;
  } while (s >= 1 || s == 0);
  const double m = sqrt(-2 * log(s) / s);
  spare_ = v * m;
  have_spare_ = true;
  return u * m;
}

double MMSBBatch::ran_gamma(double a, double b) {           // a >= 1
  const double d = a - 1.0 / 3.0, c = (1.0 / 3.0) / sqrt(d);
  for (;;) {
    double x, v;
    do { x = ran_gaussian(); v = 1.0 + c * x; } while (v <= 0);
    v = v * v * v;
    double u;
    do u = rng_.uniform(); while (u == 0);
    if (u < 1 - 0.0331 * x * x * x * x) return b * d * v;
    if (log(u) < 0.5 * x * x + d * (1 - v + log(v))) return b * d * v;
  }
}

void MMSBBatch::init_gamma() {
  for (size_t i = 0; i < gamma_.size(); ++i) gamma_[i] = ran_gamma(100, 1. / 100);
}

void MMSBBatch::set_dir_exp() {                             // src/mmsbinfer.hh:563-580 (same as A5)
  for (uint32_t i = 0; i < n_; ++i) {
    const double *g = &gamma_[(size_t)i * k_];
    double s = 0;
    for (uint32_t k = 0; k < k_; ++k) s += g[k];
    const double ps = digamma(s);
    for (uint32_t k = 0; k < k_; ++k) elogpi_[(size_t)i * k_ + k] = digamma(g[k]) - ps;
  }
  for (uint32_t k = 0; k < k_; ++k) {
    const double ps = digamma(lambda_[2 * k] + lambda_[2 * k + 1]);
    elogbeta_[2 * k] = digamma(lambda_[2 * k]) - ps;
    elogbeta_[2 * k + 1] = digamma(lambda_[2 * k + 1]) - ps;
  }
}

// ---------------------------------------------------------------------------
// PhiComp (src/mmsbinfer.hh:104-203): Jacobi-style fixed point of the two indicator vectors of
// one pair, at most 50 rounds, convergence tested on odd rounds against the vectors of two rounds ago
// ---------------------------------------------------------------------------
void MMSBBatch::phis(uint32_t p, uint32_t q, int y, double *phi1, double *phi2) const {
  const uint32_t K = k_;
  const double logeps = log(env_.epsilon);
  std::vector<double> buf(5 * (size_t)K);
  double *elogf = &buf[0], *old1 = &buf[K], *old2 = &buf[2 * K], *nx1 = &buf[3 * K], *nx2 = &buf[4 * K];
  for (uint32_t k = 0; k < K; ++k) {
    phi1[k] = phi2[k] = 1. / K;
    elogf[k] = elogbeta_[2 * k] * y + elogbeta_[2 * k + 1] * (1 - y);
    old1[k] = old2[k] = 0;
  }
  auto update = [&](const double *b, uint32_t c, double *next) {
    double s = 0;
    for (uint32_t k = 0; k < K; ++k) {
      const double u = y == 1 ? (1 - b[k]) * logeps : .0;
      next[k] = exp(elogpi_[(size_t)c * K + k] + elogf[k] * b[k] + u);
      s += next[k];
    }
    if (!(s > .0)) {
      fprintf(stderr, "error: phi normaliser underflow for pair (%u,%u)\n", p, q);   // reference: assert(s > .0)
      exit(-1);
    }
    for (uint32_t k = 0; k < K; ++k) next[k] /= s;
  };
  for (uint32_t i = 0; i < kOnlineIterations; ++i) {
    if (i % 2 == 0)
      for (uint32_t k = 0; k < K; ++k) { old1[k] = phi1[k]; old2[k] = phi2[k]; }
    update(phi2, p, nx1);
    update(phi1, q, nx2);
    double m1 = 0, m2 = 0;
    for (uint32_t k = 0; k < K; ++k) {
      m1 += fabs(nx1[k] - old1[k]);
      m2 += fabs(nx2[k] - old2[k]);
      phi1[k] = nx1[k];
      phi2[k] = nx2[k];
    }
    if (i % 2 == 0) continue;
    if (m1 / K < kMeanChangeThresh && m2 / K < kMeanChangeThresh) break;
  }
}

void MMSBBatch::sweep() {                                   // src/mmsbinfer.cc:846-889
  set_dir_exp();
  std::vector<double> phi1(k_), phi2(k_);
  for (uint32_t p = 0; p < n_; ++p)
    for (uint32_t q = p + 1; q < n_; ++q) {
      const Edge e(p, q);
      if (heldout_map_.count(e) || validation_map_.count(e)) continue;
      const int y = network_.y(p, q) ? 1 : 0;
      phis(p, q, y, phi1.data(), phi2.data());
      for (uint32_t k = 0; k < k_; ++k) {
        gammanext_[(size_t)p * k_ + k] += phi1[k];
        gammanext_[(size_t)q * k_ + k] += phi2[k];
        lambdanext_[2 * k + (y ? 0 : 1)] += phi1[k] * phi2[k];
      }
    }
  gamma_ = gammanext_;
  gammanext_.assign(gammanext_.size(), env_.alpha);
  lambda_ = lambdanext_;
  for (uint32_t k = 0; k < k_; ++k) { lambdanext_[2 * k] = env_.eta0; lambdanext_[2 * k + 1] = env_.eta1; }
  iter_++;
}

// ---------------------------------------------------------------------------
// likelihoods and the stop rule
// ---------------------------------------------------------------------------
double MMSBBatch::edge_likelihood(uint32_t p, uint32_t q, int y) const {
  const double *gp = &gamma_[(size_t)p * k_], *gq = &gamma_[(size_t)q * k_];
  double sp = 0, sq = 0;
  for (uint32_t k = 0; k < k_; ++k) { sp += gp[k]; sq += gq[k]; }
  double s = 0;
  if (y == 1) {
    for (uint32_t z = 0; z < k_; ++z)
      s += (gp[z] / sp) * (gq[z] / sq) * (lambda_[2 * z] / (lambda_[2 * z] + lambda_[2 * z + 1]));
  } else {
    for (uint32_t zp = 0; zp < k_; ++zp)
      for (uint32_t zq = 0; zq < k_; ++zq) {
        const double brate = zp == zq ? lambda_[2 * zp] / (lambda_[2 * zp] + lambda_[2 * zp + 1]) : env_.epsilon;
        s += (gp[zp] / sp) * (gq[zq] / sq) * (1 - brate);
      }
  }
  if (s < 1e-30) s = 1e-30;
  return log(s);
}

bool MMSBBatch::heldout_likelihood() {                      // src/mmsbinfer.cc:2086-2174
  if (env_.accuracy) return false;
  uint32_t k = 0, kzeros = 0, kones = 0;
  double s = 0, szeros = 0, sones = 0;
  for (const auto &it : heldout_map_) {
    const int y = network_.y(it.first.first, it.first.second) ? 1 : 0;
    const double u = edge_likelihood(it.first.first, it.first.second, y);
    s += u;
    k++;
    if (y) { sones += u; kones++; } else { szeros += u; kzeros++; }
  }
  const double nshol = zeros_prob_ * (szeros / kzeros) + ones_prob_ * (sones / kones);
  const double row[10] = {(double)iter_, s / k, (double)k, szeros / kzeros, (double)kzeros, sones / kones,
                          (double)kones, zeros_prob_ * (szeros / kzeros), ones_prob_ * (sones / kones), nshol};
  rows_.insert(rows_.end(), row, row + 10);
  if (hf_) {
    fprintf(hf_, "%d\t%d\t%.9f\t%d\t%.9f\t%d\t%.9f\t%d\t%.9f\t%.9f\t%.9f\n", iter_, duration(), s / k, k,
            szeros / kzeros, kzeros, sones / kones, kones, row[7], row[8], nshol);
    fflush(hf_);
  }
  const double a = nshol;
  bool stop = false;
  int why = -1;
  if (iter_ > n_ || iter_ > 5000) {
    if (a > prev_h_ && prev_h_ != 0 && fabs((a - prev_h_) / prev_h_) < 0.00001) {
      stop = true;
      why = 0;
    } else if (a < prev_h_) {
      nh_++;
    } else if (a > prev_h_) {
      nh_ = 0;
    }
    if (a > max_h_) {
      double av = 0;
      validation_likelihood(&av);
      max_h_ = a;
      max_v_ = av;
      max_t_ = 0;
    }
    if (nh_ > 2) { why = 1; stop = true; }
  }
  prev_h_ = nshol;
  if (env_.write_files) {
    FILE *f = open_or_die(Env::file_str("/max.txt"), "max");
    fprintf(f, "%d\t%d\t%.5f\t%.5f\t%.5f\t%.5f\t%d\n", iter_, duration(), a, max_t_, max_h_, max_v_, why);
    fclose(f);
  }
  return env_.use_validation_stop && stop;
}

void MMSBBatch::validation_likelihood(double *av) {         // src/mmsbinfer.cc:2177-2224
  if (env_.accuracy) return;
  uint32_t k = 0, kzeros = 0, kones = 0;
  double s = 0, szeros = 0, sones = 0;
  for (const auto &it : validation_map_) {
    const int y = network_.y(it.first.first, it.first.second) ? 1 : 0;
    const double u = edge_likelihood(it.first.first, it.first.second, y);
    s += u;
    k++;
    if (y) { sones += u; kones++; } else { szeros += u; kzeros++; }
  }
  if (vf_) {
    fprintf(vf_, "%d\t%d\t%.5f\t%d\t%.5f\t%d\t%.5f\t%d\n", iter_, duration(), s / k, k, szeros / kzeros, kzeros,
            sones / kones, kones);
    fflush(vf_);
  }
  if (av) *av = s / k;
}

bool MMSBBatch::report() {                                  // src/mmsbinfer.cc:895-905
  set_dir_exp();
  if (heldout_likelihood()) return true;
  validation_likelihood(nullptr);
  return false;
}

int MMSBBatch::batch_infer() {
  for (;;) {
    if (env_.max_iterations && iter_ > env_.max_iterations) {
      printf("+ Quitting: reached max iterations.\n");
      Env::plog("maxiterations reached", true);
      // the reference exits here without saving anything (src/mmsbinfer.cc:840-844); the model is
      // written so that a bounded run leaves usable output
      do_on_stop();
      return 0;
    }
    sweep();
    if (iter_ % env_.reportfreq == 0) {
      if (report()) {
        do_on_stop();
        return 1;
      }
      printf("\riteration = %d took %d secs", iter_, duration());
      fflush(stdout);
      if (env_.terminate) {
        do_on_stop();
        env_.terminate = 0;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// writers (src/mmsbinfer.cc:743-752, 932-1062)
// ---------------------------------------------------------------------------
void MMSBBatch::do_on_stop() {
  if (!env_.write_files) return;
  save_model();
  compute_and_log_groups();
}

void MMSBBatch::save_model() {
  FILE *gf = open_or_die(Env::file_str("/gamma.txt"), "gamma");
  const std::vector<uint32_t> &s2i = network_.seq2id();
  for (uint32_t i = 0; i < n_; ++i) {
    fprintf(gf, "%d\t%d\t", i, s2i[i]);
    for (uint32_t k = 0; k < k_; ++k) fprintf(gf, k == k_ - 1 ? "%.5f\n" : "%.5f\t", gamma_[(size_t)i * k_ + k]);
  }
  fclose(gf);
  FILE *lf = open_or_die(Env::file_str("/lambda.txt"), "lambda");
  for (uint32_t k = 0; k < k_; ++k) fprintf(lf, "%d\t%.5f\t%.5f\n", k, lambda_[2 * k], lambda_[2 * k + 1]);
  fclose(lf);
}

void MMSBBatch::compute_and_log_groups() {
  FILE *groupsf = open_or_die(Env::file_str("/groups.txt"), "groups");
  FILE *summaryf = open_or_die(Env::file_str("/summary.txt"), "summary", "a");
  FILE *commf = open_or_die(Env::file_str("/communities.txt"), "communities");
  const std::vector<uint32_t> &s2i = network_.seq2id();
  std::vector<double> epi((size_t)n_ * k_), beta(k_);
  for (uint32_t i = 0; i < n_; ++i) {
    double s = 0;
    for (uint32_t k = 0; k < k_; ++k) s += gamma_[(size_t)i * k_ + k];
    for (uint32_t k = 0; k < k_; ++k) epi[(size_t)i * k_ + k] = gamma_[(size_t)i * k_ + k] / s;
  }
  for (uint32_t k = 0; k < k_; ++k) beta[k] = lambda_[2 * k] / (lambda_[2 * k] + lambda_[2 * k + 1]);
  std::map<uint32_t, std::vector<uint32_t> > communities;
  std::vector<uint32_t> groups(n_, 0);
  uint32_t unlikely = 0;
  for (uint32_t i = 0; i < n_; ++i) {
    const double *pi = &epi[(size_t)i * k_];
    fprintf(groupsf, "%d\t%d\t", i, s2i[i]);
    double max = .0;
    for (uint32_t j = 0; j < k_; ++j) {
      fprintf(groupsf, "%.3f\t", pi[j]);
      if (pi[j] > max) { max = pi[j]; groups[i] = j; }
    }
    fprintf(groupsf, "%d\n", groups[i]);
    for (uint32_t m = i + 1; m < n_; ++m) {
      if (!network_.y(i, m)) continue;
      const double *pm = &epi[(size_t)m * k_];
      // inner_prod_max (src/matrix.hh:459-476): largest term of sum_k pi_i pi_m beta over the sum
      double u = .0, s = .0;
      uint32_t max_k = 0;
      for (uint32_t k = 0; k < k_; ++k) {
        const double v = pi[k] * pm[k] * beta[k];
        s += v;
        if (v > u) { u = v; max_k = k; }
      }
      if (u / s < 0.5) { unlikely++; continue; }
      communities[max_k].push_back(i);
      communities[max_k].push_back(m);
    }
  }
  printf("unlikely = %d\n", unlikely);
  std::vector<int> sizes(k_, 0);
  for (uint32_t i = 0; i < n_; ++i) sizes[groups[i]]++;
  for (uint32_t k = 0; k < k_; ++k) fprintf(summaryf, "%d\t", sizes[k]);
  fprintf(summaryf, ":%d\n\n", unlikely);
  for (const auto &c : communities) {
    std::map<uint32_t, bool> uniq;
    for (uint32_t p : c.second)
      if (!uniq.count(p)) {
        fprintf(commf, "%d ", s2i[p]);
        uniq[p] = true;
      }
    fprintf(commf, "\n");
  }
  fclose(groupsf);
  fclose(summaryf);
  fclose(commf);
}

}  // namespace svinet
