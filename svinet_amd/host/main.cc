// svinet -- command line of the MI355X link-sampling build.
//
// Accepts the reference's flag set (src/main.cc:114-242, same spelling, same
// positional/order-sensitive semantics: e.g. -link-sampling resets rfreq to 1,
// so -rfreq must follow it).  Engines: -link-sampling (the MI355X path) and the
// reference's small all-pairs CPU engine -batch (plumbing only); the
// flags that select the reference's other engines are recognised and rejected
// with a message instead of being silently ignored.
#include <csignal>
#include <cerrno>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "env.hh"
#include "linksampling.hh"
#include "mmsbbatch.hh"
#include "network.hh"

using namespace svinet;

static Env *env_global = nullptr;

static void term_handler(int sig) {   // src/main.cc:29-40
  if (env_global) {
    printf("\nGot signal. Saving model and groups.\n");
    fflush(stdout);
    env_global->terminate = 1;
  } else {
    signal(sig, SIG_DFL);
    raise(sig);
  }
}

// -gpus N, parent process: the ranks it forked (-1 once reaped); signals are passed on to them
static std::vector<pid_t> g_kids;
static void forward_handler(int sig) {
  for (pid_t k : g_kids)
    if (k > 0) kill(k, sig);
}

static void usage() {
  fprintf(stdout,
          "\nSVINET (MI355X link-sampling build): stochastic variational inference of undirected networks\n"
          "svinet [OPTIONS]\n"
          "\t-help\t\tusage\n\n"
          "\t-file <name>\tinput tab-separated file with a list of undirected links\n\n"
          "\t-n <N>\t\tnumber of nodes in network\n\n"
          "\t-k <K>\t\tnumber of communities\n\n"
          "\t-link-sampling\tinference using link sampling (the MI355X engine of this build)\n\n"
          "\t-batch\t\trun batch variational inference over all pairs (host CPU, small graphs)\n\n"
          "\t-load-validation <fname>\tuse the pairs in the file as the validation set for convergence\n\n"
          "\t-load <dir>\tresume from <dir>gamma.txt / <dir>lambda.txt\n\n"
          "\t-label\t\ttag output directory\n\n"
          "\t-rfreq\t\tset the frequency at which convergence is estimated (give it after -link-sampling)\n\n"
          "\t-max-iterations\tmaximum number of iterations (use with -no-stop to avoid stopping earlier)\n\n"
          "\t-no-stop\tdisable stopping criteria\n\n"
          "\t-seed\t\tset random generator seed\n\n"
          "\t-heldout-ratio, -link-thresh, -lt-min-deg, -eta-type, -accuracy\tas in the reference\n\n"
          "\t-strid\t\tnode names are strings (writes str2id.txt)\n\n"
          "\t-nmi <file>\tground-truth communities (\"node<TAB>community ...\" per line): the normalised mutual\n"
          "\t\t\tinformation of every communities.txt against it is appended to mutual.txt\n\n"
          "\t-device <d>\tHIP device ordinal (default 0)\n\n"
          "\t-gpus <N>\t-link-sampling over N GPUs of this node: one process per GPU (devices d .. d+N-1), node-block\n"
          "\t\t\tsharding, RCCL all-reduce / all-gather over xGMI between the phases of a sweep\n\n"
          "\t\t\twith -minibatch <m>: every GPU steps through windows of m nodes of its own block; per step an all-reduce\n"
          "\t\t\tof the K-vectors and broadcasts of the touched gamma rows (svils_step_sharded)\n\n"
          "\t-device-list <d0,d1,..>\tthe HIP device ordinal of every rank of a -gpus N run (default: d, d+1, ..)\n\n"
          "\t-sharded\ttake the -gpus N code path with one GPU as well (a communicator of one rank)\n\n"
          "\t-kshard\t\twith -gpus N: shard the K communities over the N GPUs (every GPU holds K/N columns of all rows;\n"
          "\t\t\tper sweep four all-reduces of O(links) doubles instead of an all-gather of the rows) -- the\n"
          "\t\t\tlayout for large K (-link-thresh < 0.5 adds two exchanges of one double per link: the arg-max tagging rule);\n"
          "\t\t\twith -minibatch <m> every GPU steps through the same windows of m nodes on its own columns\n\n"
          "\t-sweep-batch <b>\tsweeps per report chunk (default 0 = automatic: 1, 2, 4, 8, then 16 sweeps per report;\n\t\t\t\treports are written while the device sweeps on)\n\n"
          "\t-sparse-after <i>\tthe active-set branch of the phi pass is used once the iteration count exceeds i\n"
          "\t\t\t(default 1000, the reference's constant)\n\n"
          "\t-minibatch <m>\tmini-batch mode of -link-sampling: one step = the links of m randomly chosen nodes,\n"
          "\t\t\tRobbins-Monro step sizes (-tau0 -kappa -nodetau0 -nodekappa; defaults 1024 0.9 1024 0.5);\n"
          "\t\t\tgive -rfreq <steps> after -link-sampling to evaluate the stop rule every <steps> steps\n\n");
  fflush(stdout);
}

int main(int argc, char **argv) {
  signal(SIGTERM, term_handler);
  Env::Args a;
  bool unsupported = false;
  std::string unsupported_flag;
  if (argc == 1) {
    usage();
    exit(-1);
  }
  auto need = [&](int i) {
    if (i + 1 > argc - 1) {
      fprintf(stderr, "+ insufficient arguments!\n");
      exit(-1);
    }
  };
  for (int i = 1; i <= argc - 1; ++i) {
    const char *f = argv[i];
    auto is = [&](const char *s) { return strcmp(f, s) == 0; };
    if (is("-help")) { usage(); exit(0); }
    else if (is("-force") || is("-online") || is("-nodelay")) {}
    else if (is("-file")) { need(i); a.datfname = argv[++i]; }
    else if (is("-batch")) { a.batch = true; a.link_sampling = false; a.rfreq = 1; }
    else if (is("-link-sampling")) { a.link_sampling = true; a.batch = false; a.rfreq = 1; }
    else if (is("-load")) { need(i); a.load = true; a.location = argv[++i]; }
    else if (is("-load-validation")) { need(i); a.val_load = true; a.val_file_location = argv[++i]; }
    else if (is("-load-test")) { need(i); a.test_load = true; a.test_file_location = argv[++i]; }
    else if (is("-n")) { need(i); a.n = atoi(argv[++i]); }
    else if (is("-k")) { need(i); a.k = atoi(argv[++i]); }
    else if (is("-label")) { need(i); a.label = argv[++i]; }
    else if (is("-nthreads")) { need(i); a.nthreads = atoi(argv[++i]); }
    else if (is("-eta-type")) { need(i); a.eta_type = argv[++i]; }
    else if (is("-nmi")) { need(i); a.ground_truth_fname = argv[++i]; a.nmi = true; }
    else if (is("-rfreq")) { need(i); a.rfreq = atoi(argv[++i]); }
    else if (is("-accuracy")) { a.accuracy = true; }
    else if (is("-max-iterations")) { need(i); a.max_iterations = atoi(argv[++i]); }
    else if (is("-no-stop")) { a.use_validation_stop = false; }
    else if (is("-seed")) { need(i); a.rand_seed = atof(argv[++i]); }
    else if (is("-heldout-ratio")) { need(i); a.hol_ratio = atof(argv[++i]); }
    else if (is("-link-thresh")) { need(i); a.link_thresh = atof(argv[++i]); }
    else if (is("-lt-min-deg")) { need(i); a.lt_min_deg = atof(argv[++i]); }
    else if (is("-strid")) { a.strid = true; }
    else if (is("-device")) { need(i); a.device = atoi(argv[++i]); }
    else if (is("-gpus")) { need(i); a.gpus = atoi(argv[++i]); }
    else if (is("-device-list")) {
      need(i);
      a.device_list.clear();
      for (const char *p = argv[++i]; *p;) {
        char *e = nullptr;
        a.device_list.push_back((int)strtol(p, &e, 10));
        if (e == p) { fprintf(stderr, "error: -device-list wants d0,d1,...\n"); return -1; }
        p = *e == ',' ? e + 1 : e;
      }
    }
    else if (is("-kshard")) { a.kshard = true; }
    else if (is("-sharded")) { a.sharded = true; }
    else if (is("-sweep-batch")) { need(i); a.sweep_batch = atoi(argv[++i]); }
    else if (is("-outdir")) { need(i); a.outdir_root = argv[++i]; }
    else if (is("-sparse-after")) { need(i); a.sparse_after = atoi(argv[++i]); }
    else if (is("-minibatch")) { need(i); a.minibatch = atoi(argv[++i]); }
    else if (is("-tau0")) { need(i); a.tau0 = atof(argv[++i]); }
    else if (is("-kappa")) { need(i); a.kappa = atof(argv[++i]); }
    else if (is("-nodetau0")) { need(i); a.nodetau0 = atof(argv[++i]); }
    else if (is("-nodekappa")) { need(i); a.nodekappa = atof(argv[++i]); }
    else if (is("-init-communities")) { need(i); a.init_comm = true; a.init_comm_fname = argv[++i]; }   // src/main.cc:237-239
    else if (is("-stopthresh") || is("-inf") || is("-scale") || is("-itype") || is("-groups-file")) {
      need(i); ++i;   // value flags of other engines: consumed, no effect on this path
    }
    else if (is("-gen") || is("-ppc") || is("-lcstats") || is("-gml") || is("-findk") || is("-stratified") ||
             is("-rnode") || is("-rpair") || is("-orig") || is("-infset") || is("-single") ||
             is("-preprocess") || is("-gp") || is("-adamic-adar") || is("-disjoint") ||
             is("-load-test-sets")) {
      unsupported = true;
      unsupported_flag = f;
    }
    // unknown flags are ignored, as in the reference
  }
  if (unsupported || !(a.batch || a.link_sampling)) {
    fprintf(stderr,
            "svinet (MI355X build): only the -link-sampling and -batch engines are implemented here%s%s.\n"
            "Use the reference build for the other engines.\n",
            unsupported ? "; unsupported option " : "", unsupported ? unsupported_flag.c_str() : "");
    return 2;
  }
  if (a.kshard && !a.link_sampling) {
    fprintf(stderr, "error: -kshard belongs to -link-sampling runs\n");
    return -1;
  }
  if (a.n == 0 || a.k == 0) {
    fprintf(stderr, "error: -n and -k are required\n");
    return -1;
  }

  if (a.gpus < 1) {
    fprintf(stderr, "error: -gpus needs a positive count\n");
    return -1;
  }
  if (a.kshard && (uint32_t)a.gpus > a.k) {
    fprintf(stderr, "error: -kshard needs at least one community per GPU\n");
    return -1;
  }
  if (!a.device_list.empty() && (int)a.device_list.size() != a.gpus) {
    fprintf(stderr, "error: -device-list names %d devices for -gpus %d\n", (int)a.device_list.size(), a.gpus);
    return -1;
  }
  if (a.gpus == 1 && !a.device_list.empty()) a.device = a.device_list[0];

  // -gpus N: fork one process per GPU before anything touches the HIP runtime.  Every rank reads the
  // graph and runs the (seeded, deterministic) host-side initialisation itself; rank 0 owns the output
  // directory and the files, the others compute their node block (or column slice) only.  The communicator
  // id goes from rank 0 to rank r through a pipe made here.
  if (a.link_sampling && a.gpus > 1) {
    // RCCL shares device buffers between the ranks' processes: the host driver of this GPU pool only supports dmabuf
    // IPC (without it: hipIpcGetMemHandle: invalid argument).  The runtime reads the variable when HIP is first
    // touched -- in the ranks, after the fork below; a value the user exported stays.
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    std::vector<int> rfds((size_t)a.gpus, -1), wfds((size_t)a.gpus, -1);
    for (int r = 1; r < a.gpus; ++r) {
      int fd[2];
      if (pipe(fd)) { perror("pipe"); return -1; }
      rfds[r] = fd[0];
      wfds[r] = fd[1];
    }
    const int dev0 = a.device;
    for (int r = 0; r < a.gpus; ++r) {
      const pid_t pid = fork();
      if (pid < 0) {
        perror("fork");
        for (pid_t k : g_kids) kill(k, SIGKILL);
        return -1;
      }
      if (pid == 0) {
        g_kids.clear();
        a.rank = r;
        a.device = a.device_list.empty() ? dev0 + r : a.device_list[r];
        if (r > 0) a.write_files = false;
        for (int q = 1; q < a.gpus; ++q) {   // keep only the ends this rank uses
          if (r == 0) a.comm_wfds.push_back(wfds[q]); else close(wfds[q]);
          if (q == r) a.comm_rfd = rfds[q]; else close(rfds[q]);
        }
        goto run;
      }
      g_kids.push_back(pid);
    }
    for (int q = 1; q < a.gpus; ++q) { close(rfds[q]); close(wfds[q]); }
    // SIGTERM (the reference's "save the model" signal) and SIGINT go on to the ranks; they agree among themselves
    // at their next poll (LinkSampling::sweep_loop)
    signal(SIGTERM, forward_handler);
    signal(SIGINT, forward_handler);
    // reap in the order the ranks finish: the first one that fails takes the others with it, because they would
    // wait in a collective for ever; only pids not reaped yet are signalled
    int rc = 0;
    size_t left = g_kids.size();
    while (left) {
      int st = 0;
      const pid_t pid = waitpid(-1, &st, 0);
      if (pid < 0) {
        if (errno == EINTR) continue;
        break;
      }
      bool ours = false;
      for (pid_t &k : g_kids)
        if (k == pid) { k = -1; ours = true; }
      if (!ours) continue;
      --left;
      const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
      if (code != 0 && rc == 0) {
        rc = code;
        for (pid_t k : g_kids)
          if (k > 0) kill(k, SIGKILL);
      }
    }
    return rc;
  }
run:
  Env env(a);
  env_global = &env;
  Network network(env);
  timespec t_read0, t_read1;
  clock_gettime(CLOCK_MONOTONIC, &t_read0);
  if (network.read(a.datfname) < 0) {
    fprintf(stderr, "error reading %s; quitting\n", a.datfname.c_str());
    return -1;
  }
  clock_gettime(CLOCK_MONOTONIC, &t_read1);
  env.n = network.n() - network.singles();   // src/main.cc:291
  if (network.ones() == 0 || env.n < 2) {
    fprintf(stderr, "error: no links read from %s; quitting\n", a.datfname.c_str());
    return -1;
  }
  if (a.batch) {                             // src/main.cc:354-358
    printf("+ running mmsb batch inference\n");
    MMSBBatch mmsb(env, network);
    mmsb.batch_infer();
    exit(0);
  }
  LinkSampling ls(env, network);
  const int how = ls.infer();
  if (const char *tf = getenv("SVINET_TIMING_FILE")) {   // where the wall time went (bench.py's cli_end_to_end record)
    if (FILE *f = fopen(tf, "w")) {
      const LinkSampling::Timing &t = ls.timing();
      fprintf(f, "{\"read_s\": %.6f, \"ctor_s\": %.6f, \"graph_upload_s\": %.6f, \"sweeps_s\": %.6f, \"sweeps\": %u, \"chunks\": %u, "
                 "\"reports\": %u, \"communities_written\": %u, \"report_host_s\": %.6f, \"final_files_s\": %.6f, \"pipelined\": %s, "
                 "\"ended_by\": \"%s\"}\n",
              (double)(t_read1.tv_sec - t_read0.tv_sec) + 1e-9 * (double)(t_read1.tv_nsec - t_read0.tv_nsec), t.ctor_s, t.graph_upload_s,
              t.sweeps_t1 - t.sweeps_t0, t.sweeps, t.chunks, t.reports, t.communities_written, t.report_host_s, t.final_files_s,
              t.pipelined ? "true" : "false", how == 1 ? "stop rule" : "max iterations");
      fclose(f);
    }
  }
  exit(0);
}
