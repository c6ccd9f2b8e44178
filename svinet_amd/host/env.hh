// env.hh -- run configuration, output directory and param.txt, for the flags
// the link-sampling path reads.  Mirrors the reference's Env (src/env.hh):
// same defaults (ctor init list :305-483), same output-directory naming
// (:503-568), same param.txt keys (:577-619) and the network.dat symlink.
#pragma once
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace svinet {

class Env {
 public:
  struct Args {
    uint32_t n = 0, k = 0;
    std::string datfname = "network.dat";
    std::string label = "mmsb";
    bool batch = false, link_sampling = false;
    bool load = false;
    std::string location;
    bool val_load = false;
    std::string val_file_location;
    bool test_load = false;
    std::string test_file_location;
    bool init_comm = false;             // -init-communities <file>
    std::string init_comm_fname;
    double hol_ratio = 0.01;
    std::string eta_type = "uniform";
    uint32_t rfreq = 1;
    bool accuracy = false;
    bool defer_init_gamma = false;   // library callers (host_api.Setup(host_gamma=False)): init_gamma2 is left to svils_init_gamma
    uint32_t max_iterations = 0;
    bool use_validation_stop = true;
    double rand_seed = 0;
    double link_thresh = 0.5;
    uint32_t lt_min_deg = 0;
    bool nmi = false;
    std::string ground_truth_fname;
    uint32_t nthreads = 0;
    bool strid = false;
    // extensions of this build (not in the reference)
    int device = 0;
    uint32_t sweep_batch = 0;   // sweeps enqueued per report chunk / between host polls; 0 = automatic (1, 2, 4, 8, then 16)
    std::string outdir_root;    // directory in which the output dir is created ("" = cwd)
    bool write_files = true;    // false: library use (bench / tests), nothing touches the disk
    // mini-batch mode of -link-sampling (include/svils.h, svils_step): 0 = full sweeps (the reference's loop)
    uint32_t minibatch = 0;     // nodes per mini-batch
    double tau0 = 1024, kappa = 0.9, nodetau0 = 1024, nodekappa = 0.5;   // src/env.hh:405-408
    int32_t sparse_after = 1000;   // the active-set branch needs _iter > this (src/linksampling.cc:634)
    // -gpus N: one process per GPU (forked by main), node-block sharding with RCCL exchanges inside
    // the device library (svils_sweep_sharded); rank 0 writes the files
    int gpus = 1, rank = 0;
    bool sharded = false;       // node-block sharding: set by -gpus N > 1, or by -sharded with one GPU (a communicator of one rank: the same code path)
    bool kshard = false;        // -kshard: shard the K columns over the ranks instead of the nodes
    // the ncclUniqueId travels from rank 0 to the others through pipes made before the fork: rank 0 holds the write
    // ends, rank r > 0 the read end of its own
    int comm_rfd = -1;
    std::vector<int> comm_wfds;
    std::vector<int> device_list;   // -device-list d0,d1,..: the HIP ordinal of every rank (default: device, device+1, ..)
  };

  explicit Env(const Args &a);
  ~Env();

  uint32_t n, k, t;
  double alpha;
  double heldout_ratio;
  double eta0, eta1;
  const double eta0_dense, eta1_dense, eta0_sparse, eta1_sparse;
  uint32_t reportfreq;
  double epsilon;
  uint32_t max_iterations;
  double seed;
  std::string eta_type;
  bool use_validation_stop;
  bool accuracy;
  bool defer_init_gamma;
  double link_thresh;
  uint32_t lt_min_deg;
  bool model_load;
  std::string gamma_location;
  bool load_heldout;
  std::string load_heldout_fname;
  bool load_test;
  std::string load_test_fname;
  bool use_init_communities;
  std::string init_communities_fname;
  bool nmi;
  std::string ground_truth_fname;
  std::string datfname, label;
  int gpus, rank;
  bool kshard, sharded;
  int comm_rfd;
  std::vector<int> comm_wfds;
  bool batch_mode, link_sampling;
  bool strid;
  volatile int terminate;
  // set by Network::set_env_variables
  uint64_t total_pairs;
  double ones_prob, zeros_prob;
  // extensions
  int device;
  uint32_t sweep_batch;
  bool write_files;
  uint32_t minibatch;
  double tau0, kappa, nodetau0, nodekappa;
  int32_t sparse_after;

  static std::string prefix;
  static std::string file_str(const std::string &fname) { return prefix + fname; }
  static void plog(const std::string &s, const std::string &v);
  static void plog(const std::string &s, const char *v) { plog(s, std::string(v)); }
  static void plog(const std::string &s, double v);
  static void plog(const std::string &s, bool v);
  static void plog(const std::string &s, int v);
  static void plog(const std::string &s, uint32_t v);
  static void plog(const std::string &s, uint64_t v);

 private:
  static FILE *plogf_;
};

}  // namespace svinet
