// linksampling.hh -- host side of the `-link-sampling` engine.
//
// Same seam as the reference (src/linksampling.hh:21-31, used at
// src/main.cc:337-342):
//     LinkSampling ls(env, network);   ls.infer();
// The constructor does what the reference's does on the host (held-out
// sampling, gamma/lambda initialisation, output files, constructor-time
// likelihood row); infer() drives the device-resident sweep through the C ABI
// of include/svils.h and writes the reference's output files.
#pragma once
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <map>
#include <memory>
#include <utility>
#include <string>
#include <vector>

#include "env.hh"
#include "network.hh"
#include "nmi.hh"
#include "rng.hh"

struct svils_handle;

namespace svinet {

// n x k doubles WITHOUT a zero fill when resized (4.1 GB at n = 1e6, k = 512: 0.8 s of page zeroing the state fetched from the
// device overwrites anyway); assign(count, value) still fills
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_allocator<U>; };
  template <class U, class... A>
  void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
    else ::new ((void *)p) U(std::forward<A>(a)...);
  }
};
using DVec = std::vector<double, default_init_allocator<double>>;

class LinkSampling {
 public:
  // attach_device = false builds the host-side state only (no HIP device is
  // touched): used by the CPU test-suite and by callers that drive the C ABI
  // themselves (bench.py).
  LinkSampling(Env &env, Network &network, bool attach_device = true);
  ~LinkSampling();

  // Runs until the reference would call exit(0): returns 0 when
  // -max-iterations was reached, 1 when the validation stop rule fired.
  int infer();
  void save_model();
  void do_on_stop();

  // ---- host-side state (flat row-major) ----
  uint32_t n() const { return n_; }
  uint32_t k() const { return k_; }
  const DVec &gamma() const { return gamma_; }
  const std::vector<double> &lambda() const { return lambda_; }
  const std::vector<uint32_t> &validation_accept() const { return val_accept_; }   // [V][3]
  const std::vector<uint32_t> &validation_sorted() const { return val_sorted_; }   // [V][3]
  const std::vector<uint32_t> &test_sorted() const { return test_sorted_; }        // [T][3]
  // assign_training_links (src/linksampling.cc:493-523); idempotent
  const std::vector<uint32_t> &training_links();                                   // [L][2]
  double total_pairs() const { return total_pairs_; }
  double ones_prob() const { return ones_prob_; }
  double zeros_prob() const { return zeros_prob_; }
  const double *row0() const { return have_row0_ ? row0_ : nullptr; }
  // init_gamma2 as svils_init_gamma takes it: every link (held-out ones included) in the order the reference draws for them
  // ([E][2], p < q), and `nstreams` MT19937 states, `per_stream` outputs apart, the first one standing where the generator stood
  // when init_gamma2 began (mtjump.hh) -- false if the jump machinery is unavailable
  uint64_t init_offset() const { return init_o0_; }
  void init_links(std::vector<uint32_t> *edges) const;
  bool init_streams(uint64_t nstreams, uint64_t per_stream, std::vector<uint32_t> *states) const;

 private:
  void init_validation();
  void load_validation();
  void load_test();                            // -load-test
  void init_gamma_external();                  // -init-communities
  void set_validation_sample(int s);
  void get_random_edge(bool link, Edge &e);
  bool edge_ok(const Edge &e) const;
  void accept_pair(const Edge &e, bool y);
  std::string edgelist_s(const std::vector<uint32_t> &triples) const;
  void init_gamma2();
  bool init_gamma2_on_device();                // svils_init_gamma instead of init_gamma2 + the upload of its result; false: not taken
  bool device_init_wanted(bool attach_device) const;
  int init_lambda();
  int load_model();
  void attach();
  void write_validation_row(const double *row, FILE *f) const;
  void write_max(const double *row, int why, double max_h) const;
  void do_on_stop_impl();
  void log_communities();
  void write_communities_file();               // communities.txt (+ mutual.txt) from tags_
  void log_rows(const double *rows, uint32_t count, int why, double max_h, const double *test, uint32_t ntest);   // validation.txt, test.txt, max.txt
  int sweep_loop_pipelined();                  // reports taken off the device's critical path (svils_report_*)
  // one whole-graph engine driving full sweeps: the pipelined loop; SVINET_SYNC_REPORTS=1 keeps the per-batch synchronous one
  bool pipelined_reports() const;
  void send_graph();                           // training links to the device (once)
  int sweep_loop();                            // the body of infer()
  void fetch_state_ksharded(DVec &g, std::vector<double> &l);   // -kshard: merged gamma / lambda (collective)
  void fetch_communities_ksharded();           // -kshard: merged tags_ (collective)
  void write_groups();
  uint32_t duration() const { return (uint32_t)(time(0) - start_time_); }
  bool fetch_and_log_rows();

  Env &env_;
  Network &network_;
  uint32_t n_, k_;
  double total_pairs_, ones_prob_, zeros_prob_;
  GslMt19937 rng_;
  std::map<Edge, bool> validation_map_;        // std::map: the likelihood loop runs in key order
  std::vector<uint32_t> val_accept_, val_sorted_;
  std::map<Edge, bool> test_map_;              // -load-test
  std::vector<uint32_t> test_sorted_;          // [T][3] p, q, y in map order
  DVec gamma_;                                 // (empty while init_gamma2 is left to the device: defer_init_)
  std::vector<double> lambda_;
  uint64_t init_o0_ = 0;                       // outputs the generator had produced when init_gamma2 began
  bool defer_init_ = false;                    // init_gamma2 runs on the device, in attach()
  std::vector<uint32_t> links_;
  bool links_done_ = false;
  uint32_t k0_ = 0, k1_ = 0;                   // -kshard: this rank's columns
  std::vector<uint32_t> tags_;                 // last downloaded communities: (device row, community) pairs (svils_get_community_tags)
  // mini-batch mode: nodes are handed to the device under a random relabelling so that a window of
  // consecutive device ids is a uniform random subset; dev_of_[seq] / seq_of_[dev], empty otherwise
  std::vector<uint32_t> dev_of_, seq_of_;
  void rank_external_ids();
  std::vector<uint32_t> blocks_;   // -gpus N (whole sweeps): bounds[N + 1] of the work-balanced node blocks, the same on every rank
  std::vector<uint32_t> ext_rank_;             // rank of a sequence id in the order of the external ids (communities.txt lists members by external id)
  Cover ground_truth_;                         // -nmi: the reference cover (external ids)
  svils_handle *h_ = nullptr;
  bool graph_sent_ = false;
  uint32_t rows_logged_ = 0;
  double row0_[10];
  bool have_row0_ = false;
  time_t start_time_;
  FILE *vf_ = nullptr, *tf_ = nullptr;
  uint32_t term_polls_ = 0;

 public:
  // where the wall time of a run went (SVINET_TIMING_FILE=path makes the CLI write it as JSON; bench.py's
  // cli_end_to_end record)
  struct Timing {
    double ctor_s = 0, graph_upload_s = 0, sweeps_t0 = 0, sweeps_t1 = 0, report_host_s = 0, final_files_s = 0;
    uint32_t sweeps = 0, chunks = 0, reports = 0, communities_written = 0;
    bool pipelined = false;
  };
  const Timing &timing() const { return timing_; }

 private:
  Timing timing_;
};

}  // namespace svinet
