#include "env.hh"

#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace svinet {

std::string Env::prefix;
FILE *Env::plogf_ = nullptr;

void Env::plog(const std::string &s, const std::string &v) {
  if (!plogf_) return;
  fprintf(plogf_, "%s: %s\n", s.c_str(), v.c_str());
  fflush(plogf_);
}
void Env::plog(const std::string &s, double v) {
  if (!plogf_) return;
  fprintf(plogf_, "%s: %.9f\n", s.c_str(), v);
  fflush(plogf_);
}
void Env::plog(const std::string &s, bool v) {
  if (!plogf_) return;
  fprintf(plogf_, "%s: %s\n", s.c_str(), v ? "True" : "False");
  fflush(plogf_);
}
void Env::plog(const std::string &s, int v) {
  if (!plogf_) return;
  fprintf(plogf_, "%s: %d\n", s.c_str(), v);
  fflush(plogf_);
}
void Env::plog(const std::string &s, uint32_t v) {
  if (!plogf_) return;
  fprintf(plogf_, "%s: %d\n", s.c_str(), v);
  fflush(plogf_);
}
void Env::plog(const std::string &s, uint64_t v) {
  if (!plogf_) return;
  fprintf(plogf_, "%s: %" PRIu64 "\n", s.c_str(), v);
  fflush(plogf_);
}

Env::Env(const Args &a)
    : n(a.n), k(a.k), t(2),
      alpha((double)1 / a.k),
      heldout_ratio(a.hol_ratio),
      eta0(0), eta1(0),
      eta0_dense(4700.59), eta1_dense(0.77), eta0_sparse(0.97), eta1_sparse(6.33),
      reportfreq(a.rfreq),
      epsilon(1e-30),
      max_iterations(a.max_iterations),
      seed(a.rand_seed),
      eta_type(a.eta_type),
      use_validation_stop(a.use_validation_stop),
      accuracy(a.accuracy),
      defer_init_gamma(a.defer_init_gamma),
      link_thresh(a.link_thresh), lt_min_deg(a.lt_min_deg),
      model_load(a.load), gamma_location(a.location),
      load_heldout(a.val_load), load_heldout_fname(a.val_file_location),
      load_test(a.test_load), load_test_fname(a.test_file_location),
      use_init_communities(a.init_comm), init_communities_fname(a.init_comm_fname),
      nmi(a.nmi), ground_truth_fname(a.ground_truth_fname),
      datfname(a.datfname), label(a.label), gpus(a.gpus), rank(a.rank), kshard(a.kshard), sharded((a.sharded || a.gpus > 1) && !a.kshard), comm_rfd(a.comm_rfd), comm_wfds(a.comm_wfds),
      batch_mode(a.batch), link_sampling(a.link_sampling), strid(a.strid),
      terminate(0), total_pairs(0), ones_prob(0), zeros_prob(1),
      device(a.device), sweep_batch(a.sweep_batch), write_files(a.write_files),
      minibatch(a.minibatch), tau0(a.tau0), kappa(a.kappa), nodetau0(a.nodetau0), nodekappa(a.nodekappa),
      sparse_after(a.sparse_after) {
  if (!write_files) {
    if (plogf_) { fclose(plogf_); plogf_ = nullptr; }
    prefix.clear();
    return;
  }
  // output directory name, src/env.hh:503-551
  std::ostringstream sa;
  sa << "n" << n << "-" << "k" << k;
  if (label != "") sa << "-" << label;
  else if (datfname.length() > 3 && datfname.find("mmsb_gen.dat") == std::string::npos) {
    std::string q = datfname.substr(0, 2);
    if (q == "..") q = "xx";
    sa << "-" << q;
  }
  if (seed) sa << "-seed" << seed;
  if (batch_mode) { sa << "-batch"; reportfreq = 1; }
  else if (link_sampling) sa << "-linksampling";
  if (a.nthreads > 0) sa << "-T" << a.nthreads;
  prefix = a.outdir_root.empty() ? sa.str() : a.outdir_root + "/" + sa.str();

  fprintf(stdout, "+ Output directory: %s\n", prefix.c_str());
  fflush(stdout);
  // Logger::setup_log_dir + setup_logfd (src/log.cc:86-127): create the dir, empty infer.log
  struct stat st;
  if (stat(prefix.c_str(), &st) != 0) {
    mkdir(prefix.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH);
    if (stat(prefix.c_str(), &st) != 0) {
      fprintf(stderr, "Warning: could not create dir %s\n", prefix.c_str());
      exit(-1);
    }
  }
  FILE *lf = fopen(file_str("/infer.log").c_str(), "w");
  if (lf) { fprintf(stdout, "+ Writing log to %s\n", file_str("/infer.log").c_str()); fclose(lf); }
  if (plogf_) fclose(plogf_);
  plogf_ = fopen(file_str("/param.txt").c_str(), "w");
  if (!plogf_) {
    printf("cannot open param file:%s\n", strerror(errno));
    exit(-1);
  }
  plog("nodes", n);
  plog("groups", k);
  plog("t", t);
  plog("minibatch (rpair or stratified rpair options only)", (uint32_t)(n / 2));
  plog("mbsize", (uint32_t)1);
  plog("alpha", alpha);
  plog("sbm_alpha", alpha);
  plog("heldout_ratio", heldout_ratio);
  plog("precision_ratio", 0.001);
  plog("stratified", false);
  plog("delaylearn", false);
  plog("nolambda", false);
  plog("randomnode", false);
  plog("gen", false);
  plog("undirected", true);
  plog("gap", false);
  plog("nthreads", a.nthreads);
  plog("stopthresh", 0.00001);
  plog("infthresh", 0.0);
  plog("randzeros", false);
  plog("benchmark", false);
  plog("max iterations", max_iterations);
  plog("seed", seed);
  plog("use validation stop", use_validation_stop);
  plog("gamma location", gamma_location);
  plog("link_thresh", link_thresh);
  plog("lt_min_deg", lt_min_deg);
  plog("epsilon", epsilon);
  plog("sets_mini_batch", (uint32_t)(n / 100));
  plog("use_init_communities", use_init_communities);
  plog("load_test_sets", false);
  plog("val_load", load_heldout);
  plog("val_file_location", load_heldout_fname);
  plog("test_load", load_test);
  plog("test_file_location", load_test_fname);
  plog("reportfreq", reportfreq);
  plog("eta_type", eta_type);
  if (minibatch) {   // keys of this build (mini-batch mode), after the reference's
    plog("link_sampling_minibatch_nodes", minibatch);
    plog("tau0", tau0);
    plog("kappa", kappa);
    plog("nodetau0", nodetau0);
    plog("nodekappa", nodekappa);
  }
  // network.dat symlink, src/env.hh:621-625
  std::string nd = file_str("/network.dat");
  unlink(nd.c_str());
  if (symlink(datfname.c_str(), nd.c_str()) < 0)
    fprintf(stderr, "warning: cannot symlink %s: %s\n", nd.c_str(), strerror(errno));
  unlink(file_str("/mutual.txt").c_str());
}

Env::~Env() {
  if (plogf_) { fclose(plogf_); plogf_ = nullptr; }
}

}  // namespace svinet
