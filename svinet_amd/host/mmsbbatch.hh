// mmsbbatch.hh -- the reference's `-batch` engine (BASELINE config 1).
//
// `svinet -file F -n N -k K -batch` in the reference constructs MMSBInfer and runs
// MMSBInfer::batch_infer() (src/main.cc:354-358): coordinate-ascent variational inference
// over ALL n(n-1)/2 pairs, on the CPU, single-threaded.  It is a different engine from the
// link-sampling hot path this repo accelerates; it is provided here as plumbing only (same
// flags, same output directory and file formats) and, like the reference's, it runs on the
// host.  It is NOT a fallback for `-link-sampling`: that path has no CPU route in this repo.
//
// Seam mirrored:   MMSBInfer mmsb(env, network);  mmsb.batch_infer();
//
// Where the reference reads members it never initialises (`_iter`, `_ones_prob`,
// `_zeros_prob`: absent from the constructor, src/mmsbinfer.cc:8-41) this class uses 0 and
// the values Network::set_env_variables computes (the ones the authors' shipped run logged,
// example/n75-k4-mmsb-batch.tgz:param.txt).
#pragma once
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <map>
#include <string>
#include <vector>

#include "env.hh"
#include "network.hh"
#include "rng.hh"

namespace svinet {

class MMSBBatch {
 public:
  MMSBBatch(Env &env, Network &network);
  ~MMSBBatch();

  // Runs until the reference would exit: returns 0 when -max-iterations was reached, 1 when the
  // held-out stop rule fired (src/mmsbinfer.cc:2086-2174).
  int batch_infer();
  // one pass of the body of batch_infer()'s loop, without the report step (src/mmsbinfer.cc:846-889)
  void sweep();
  // the report step (:895-905); returns true when the stop rule fired
  bool report();
  void do_on_stop();                                      // :743-752

  uint32_t n() const { return n_; }
  uint32_t k() const { return k_; }
  uint32_t iter() const { return iter_; }
  std::vector<double> &gamma() { return gamma_; }         // [n][k]
  std::vector<double> &lambda() { return lambda_; }       // [k][2]
  const std::vector<uint32_t> &heldout_edges() const { return heldout_edges_; }         // [H][2], acceptance order
  const std::vector<uint32_t> &validation_edges() const { return validation_edges_; }   // [V][2]
  const std::vector<double> &heldout_rows() const { return rows_; }                     // [rows][10]
  // per-pair fixed point, PhiComp::update_phis_until_conv (src/mmsbinfer.hh:159-203)
  void phis(uint32_t p, uint32_t q, int y, double *phi1, double *phi2) const;
  double edge_likelihood(uint32_t p, uint32_t q, int y) const;                          // src/mmsbinfer.hh:634-668

 private:
  void init_heldout();
  void set_sample(int s, bool heldout);
  void get_random_edge(bool heldout_flag, bool stratified, int family, Edge &e);
  bool edge_ok(const Edge &e, bool heldout_flag, bool stratified, int family) const;
  void init_gamma();
  double ran_gamma(double a, double b);
  double ran_gaussian();
  void set_dir_exp();
  bool heldout_likelihood();
  void validation_likelihood(double *av);
  void save_model();
  void compute_and_log_groups();
  std::string edgelist_s(const std::vector<uint32_t> &pairs) const;
  uint32_t duration() const { return (uint32_t)(time(0) - start_time_); }

  Env &env_;
  Network &network_;
  uint32_t n_, k_;
  uint32_t iter_ = 0;
  double ones_prob_, zeros_prob_;
  GslMt19937 rng_;
  std::map<Edge, bool> heldout_map_, validation_map_;
  std::vector<uint32_t> heldout_edges_, validation_edges_;
  std::vector<double> gamma_, gammanext_, lambda_, lambdanext_, elogpi_, elogbeta_;
  std::vector<double> rows_;
  double max_t_, max_h_, max_v_, prev_h_;
  uint32_t nh_ = 0;
  bool have_spare_ = false;
  double spare_ = 0;
  time_t start_time_;
  FILE *hf_ = nullptr, *vf_ = nullptr;
};

}  // namespace svinet
