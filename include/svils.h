/*
 * svils.h -- C ABI of the MI355X link-sampling sweep engine (libsvils.so).
 *
 * The reference (premgopalan/svinet) has no plugin/FFI boundary: its only seam
 * for this path is the C++ class used at src/main.cc:337-342
 *
 *     LinkSampling ls(env, network);   // src/linksampling.cc:5-155
 *     ls.infer();                      // src/linksampling.cc:556-790
 *
 * This header is the boundary a maintainer would bind in its place: the host
 * keeps parsing, RNG, initialisation and file formats; the library owns the
 * device-resident state and runs the body of infer()'s `while (1)` loop
 * (phi pass :605-725, compute_mean_indicators :526-545, s3 pass :731-746,
 * lambda/swap/set_dir_exp/prune :748-761, validation_likelihood :966-1050
 * including its stop rule and annealing switch) as HIP kernels for gfx950.
 *
 * Conventions: plain C, no exceptions cross the boundary, every function
 * returns 0 on success or a negative svils_error; svils_last_error() gives
 * the text for the calling thread.  A handle is not thread-safe.  The caller
 * keeps ownership of every host buffer (copied in/out).  Indices are
 * uint32_t, reals are IEEE double, matrices are flat row-major with leading
 * dimension = number of columns.  There is NO CPU fallback: creating a handle
 * without a usable HIP device fails with SVILS_ERR_DEVICE.
 */
#ifndef SVILS_H
#define SVILS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVILS_ABI_VERSION 8   /* 3: svils_config gained k_begin / k_total (K-sharded handles); 4: svils_comm_info; 5: community tags;
                                 6: work-balanced node blocks (svils_balance_node_blocks / svils_set_node_blocks), the node-block sweep
                                    with ONE row exchange (SVILS_PHASE_B_LIGHT / SVILS_PHASE_EXPAND_ALL, SVILS_BUF_GSTAGE);
                                 7: k up to SVILS_MAX_K_TOTAL (column-tiled handles above SVILS_MAX_K; K-sharded k_total up to it),
                                    getters that do not wait behind a stop the caller has seen ("After the stop");
                                 8: svils_init_gamma (init_gamma2 on the device); svils_set_option / svils_get_option / svils_option_table (every tunable in one documented table; nothing
                                    on a sweep path reads the environment); svils_gather_communities ends the no-wait window of "After the stop" */

typedef enum {
  SVILS_OK = 0,
  SVILS_ERR_ARG = -1,      /* bad argument / call order                     */
  SVILS_ERR_DEVICE = -2,   /* no HIP device, HIP runtime error              */
  SVILS_ERR_NOMEM = -3,    /* host or device allocation failed              */
  SVILS_ERR_UNSUPPORTED = -4 /* e.g. k > SVILS_MAX_K_TOTAL                   */
} svils_error;

#define SVILS_MAX_K 2048         /* columns ONE set of kernels holds: a whole-row handle up to here, a K-sharded slice up to here */
/* k above SVILS_MAX_K (the reference has no limit short of its 16-bit community ids, src/linksampling.cc:635): svils_create
 * builds a COLUMN-TILED handle -- ceil(k / SVILS_MAX_K) slices of every row on the one device, the K-sharded layout with all
 * its "ranks" on one stream and their four exchanges summed in place.  Same results as a whole-row handle to rounding.  What
 * such a handle offers: svils_set_graph / set_validation / set_state (graph before the first sweep or likelihood row) /
 * sweep / synchronize / get_control / set_control / validation_row / get_rows / get_state / get_communities /
 * get_community_tags / get_sweep_stats / destroy; everything else (reports, test set, mini-batch steps, node blocks,
 * timing) answers SVILS_ERR_UNSUPPORTED.  Across GPUs the same k goes through K-sharded handles (k_total up to
 * SVILS_MAX_K_TOTAL, at most SVILS_MAX_K columns per rank). */
#define SVILS_MAX_K_TOTAL 65535
#define SVILS_MAX_TILES 32

typedef struct svils_handle svils_handle;

/* Everything the sweep needs from Env / Network / the LinkSampling ctor.
 * Replaces the Env& / Network& arguments of LinkSampling::LinkSampling
 * (src/linksampling.cc:5-33) for the fields the loop actually reads. */
typedef struct {
  uint32_t n;             /* env.n after singleton removal, src/main.cc:291          */
  uint32_t k;             /* env.k                                                   */
  uint64_t ones;          /* network.ones(): ALL links incl. held-out (anneal scale,
                             src/linksampling.cc:541-542)                            */
  double alpha;           /* env.alpha = 1/k, src/env.hh:344                         */
  double eta0, eta1;      /* Network::set_env_variables, src/network.cc:222-251      */
  double epsilon;         /* env.epsilon = 1e-30, src/env.hh:395                     */
  double link_thresh;     /* -link-thresh, src/linksampling.cc:672,708               */
  uint32_t lt_min_deg;    /* -lt-min-deg,  src/linksampling.cc:676-679,712-715       */
  uint32_t reportfreq;    /* env.reportfreq (1 under -link-sampling)                 */
  int32_t use_validation_stop; /* 0 with -no-stop, src/linksampling.cc:1044-1048     */
  double ones_prob;       /* _ones_prob,  src/linksampling.cc:49                     */
  double zeros_prob;      /* _zeros_prob, src/linksampling.cc:50                     */
  int32_t device;         /* HIP device ordinal                                      */
  /* node-block ownership for multi-GPU runs (one process per GPU): this
   * handle computes rows [node_begin, node_end); 0,n for a single GPU.     */
  uint32_t node_begin, node_end;
  /* rows allocated for the replicated n-by-k arrays (>= n; 0 = n).  A caller
   * that all-gathers equal node blocks sets world_size * block_rows.        */
  uint32_t n_alloc;
  /* the active-set branch is taken when _iter > sparse_after_iter: the constant 1000 of
   * src/linksampling.cc:634 (svils_config_default).  The revision that produced the runs shipped
   * under example/ behaved like 0, which is how the tests reproduce them. */
  int32_t sparse_after_iter;
  /* K-sharded handles (multi-GPU layout of DESIGN.md section 6): k_total != 0 means this handle holds the
   * columns [k_begin, k_begin + k) of k_total communities for ALL n nodes; alpha stays 1/k_total,
   * gamma / lambda are passed and returned as that column slice.  0: the handle holds every column. */
  uint32_t k_begin, k_total;
} svils_config;

/* fills the reference's defaults for a given n,k (alpha=1/k, eta=1,1, ...) */
int svils_config_default(svils_config *cfg, uint32_t n, uint32_t k);

int svils_create(const svils_config *cfg, svils_handle **out);
int svils_destroy(svils_handle *h);

/* Training links exactly as LinkSampling::assign_training_links leaves them
 * (src/linksampling.cc:493-523): [nlinks][2], p < q, sorted by p then by
 * adjacency order.  _training_links[p] (= 2 * training degree, quirk Q3) is
 * derived from the list.  Builds the device CSR. */
int svils_set_graph(svils_handle *h, const uint32_t *links, uint64_t nlinks);

/* Optional: capture the hipGraphs svils_sweep replays (1, 4, 8, 16 ... sweeps, up to max_sweeps) during set-up instead of
 * in the middle of the run (by itself svils_sweep launches eagerly until a handle has run 128 sweeps: a capture costs more
 * than a short run).  After svils_set_graph / svils_set_validation / svils_set_state.  The drop-in binary calls it from the
 * LinkSampling constructor, so that its default run -- 31 sweeps on ca-AstroPh -- replays graphs from the third chunk on. */
int svils_prepare_graphs(svils_handle *h, uint32_t max_sweeps);

/* Held-out pairs in std::map<Edge,bool> order (src/linksampling.cc:974-992):
 * [nv][3] = (p, q, y).  nv == 0 disables the likelihood/stop rule. */
int svils_set_validation(svils_handle *h, const uint32_t *pairs_y, uint64_t nv);

/* Test pairs of -load-test in std::map<Edge,bool> order (LinkSampling::load_test, src/linksampling.cc:1417-1450;
 * test_likelihood, :1147-1182): [nt][3] = (p, q, y), y = network.y(p, q).  Every report that does not end the run
 * (the stopping sweep leaves validation_likelihood through do_on_stop + exit, before test_likelihood is reached,
 * :777-781) also records a TEST row with the columns of the validation row, under the same row number; fetch them
 * with svils_get_test_rows or a report.  The caller has already taken the pairs out of the training links
 * (edge_ok, src/linksampling.hh:296-305).  Sweeps of a handle with a test set run as four launches (the three-launch
 * form of K <= 32 defers the stop rule to the next launch, too late to decide whether the test row exists).
 * nt == 0 removes the set.  Not for K-sharded handles.  Test rows share the row numbers of the validation rows: a handle
 * without a validation set (svils_set_validation with nv == 0) records neither (the reference always has one on this
 * path: its constructor samples it before anything else, src/linksampling.cc:84-109).  Calling it again re-uses the
 * device buffers (they grow when a larger set arrives; nothing accumulates). */
int svils_set_test(svils_handle *h, const uint32_t *pairs_y, uint64_t nt);
/* out[count][10]: the test rows of reports [first, first + count) (numbered like the validation rows); a report that
 * recorded none (the stopping sweep) reads as NaN.  Synchronises. */
int svils_get_test_rows(svils_handle *h, uint32_t first, uint32_t count, double *rows);

/* gamma [n][k], lambda [k][2] after init_gamma2/init_lambda (or -load);
 * converged [n] or NULL (= all 0, as at the top of infer(), :559).
 * Computes Elogpi / Elogbeta (set_dir_exp, src/linksampling.hh:170-187). */
int svils_set_state(svils_handle *h, const double *gamma, const double *lambda,
                    const uint32_t *converged);

/* init_gamma2 (src/linksampling.cc:374-401) ON THE DEVICE, bit for bit: instead of drawing the E x k uniforms on the host, adding
 * them into an n x k array and uploading it (svils_set_state), the caller hands over where its MT19937 stream stands and the
 * library regenerates the draws, normalises them per link and adds them into the gamma rows in the reference's order.
 *   edges      [nedges][2], p < q, EVERY link (held-out ones included) in the order the reference's loop visits them: p ascending,
 *              then the order of p's adjacency list -- link j consumes outputs [j k, (j + 1) k) of the stream
 *   mt_states  [nstreams][624] MT19937 states in canonical form (the next output is the first word of the next twist of the
 *              state; gsl_rng's mt[] with mti == 624): state s stands outputs_per_stream * s outputs behind state 0, which stands
 *              where the reference's generator stands when init_gamma2 starts (after the validation sampler's draws).
 *              host/mtjump.hh computes them by jump-ahead; streams x outputs_per_stream must cover nedges * k exactly once.
 *   lambda     [k][2] as for svils_set_state (init_lambda, src/linksampling.cc:364-372); the converged flags are zeroed.
 * Any handle: a node-block handle holds every row anyway; a K-sharded one (k_total columns drawn per link, its slice kept,
 * lambda = the slice's rows; svils_ksh_init_state follows as after svils_set_state) regenerates the whole stream on every
 * rank -- what a rank saves is the host's 2.9 s and its n x k_total array.  Needs 4 * nedges * k_total bytes of device scratch
 * for the duration of the call (24 GB at n = 1e6, k = 512; SVILS_ERR_NOMEM if it is not there: upload the state instead).
 * Replaces 2.9 s of host work + a 4.1 GB upload by ~0.1 s of host work (the states) + ~30 ms of device work at that size. */
int svils_init_gamma(svils_handle *h, const uint32_t *edges, uint64_t nedges, const uint32_t *mt_states, uint64_t nstreams,
                     uint64_t outputs_per_stream, const double *lambda);

/* Loop-carried scalars of infer()/validation_likelihood(). */
typedef struct {
  uint32_t iter;          /* _iter                                               */
  int32_t annealing;      /* _annealing_phase                                    */
  int32_t write_comm;     /* write_comm for the NEXT sweep                       */
  int32_t nh;             /* _nh                                                 */
  double prev_h;          /* _prev_h                                             */
  double max_h;           /* _max_h                                              */
  int32_t stopped;        /* 1 once the stop rule fired with use_validation_stop;
                             further sweeps are no-ops and the state is the one
                             do_on_stop() would have saved                       */
  int32_t why;            /* last `why` of validation_likelihood (:1007-1027)    */
  uint32_t sweeps_done;   /* sweeps executed since create                        */
  uint32_t rows;          /* validation rows recorded since create               */
  uint64_t links_dense, links_sparse, links_shortcut; /* c, d and the one-converged
                             count of the last sweep (:602,:726)                 */
} svils_control;

/* After the stop: once a control block or a report that says `stopped` has reached the caller (svils_get_control,
 * svils_report_fetch*), svils_get_control / get_state / get_rows / get_test_rows / get_communities / get_community_tags
 * no longer wait for the stream: whatever a pipelined caller still has in flight behind the stopping sweep are launches
 * that return at once, and the state they would wait for is already final. */
int svils_get_control(svils_handle *h, svils_control *out);   /* synchronises (but see "After the stop") */
/* only iter, annealing, write_comm, nh, prev_h, max_h are taken from `in` */
int svils_set_control(svils_handle *h, const svils_control *in);

/* validation_likelihood() on the current state WITHOUT the stop rule: the row
 * the constructor writes (src/linksampling.cc:149-150).
 * row[10] = iter, s/k, k, mean0, k0, mean1, k1, zeros_prob*mean0,
 *           ones_prob*mean1, a   (the columns of :996-1001 minus duration). */
int svils_validation_row(svils_handle *h, double *row10);

/* Enqueue `nsweeps` iterations of the loop body; asynchronous.  The stop
 * rule, the annealing switch and _iter++ run on the device, so no host
 * round trip is needed between sweeps.  At most 65536 * reportfreq sweeps per
 * call: the likelihood rows go to a ring of 65536 entries that the host drains
 * with svils_get_rows between calls. */
int svils_sweep(svils_handle *h, uint32_t nsweeps);
int svils_synchronize(svils_handle *h);

/* phases of a sweep / mini-batch step between exchange points (see "multi-GPU hooks" below) */
typedef enum { SVILS_PHASE_A = 0, SVILS_PHASE_B, SVILS_PHASE_C, SVILS_PHASE_D, SVILS_PHASE_EXPAND,
               SVILS_PHASE_B_LIGHT = 5, SVILS_PHASE_EXPAND_ALL = 6 } svils_phase;

/* ---- mini-batch mode (an ADDITION of this build; SURVEY 8f N4, BASELINE north_star) ----
 * The reference revision's -link-sampling loop is a deterministic full sweep with step size 1
 * (src/linksampling.cc:556-790); its stochastic engines (MMSBInfer::infer,
 * src/mmsbinfer.cc:564-641) sample node/pair mini-batches and blend with a Robbins-Monro step
 * (tau0 + t)^-kappa, per node and for lambda.  svils_step() is that scheme applied to the
 * link-sampling updates: one step processes the links of a WINDOW of `batch_nodes` consecutive
 * nodes (windows taken in cyclic order -- relabel the nodes randomly for unbiased batches),
 * scales the window sums to estimates of the full sums, and blends gamma rows of the window
 * (per-node step size from the node's own update count) and lambda (step size from the step
 * number).  With batch_nodes = 0 (all nodes) and kappa = 0 a step equals a full sweep.
 * Not a parity mode: the reference has no counterpart to compare against. */
typedef struct {
  uint32_t batch_nodes;   /* nodes per mini-batch; 0 = all nodes */
  /* step sizes, the reference's four constants (src/env.hh:399-408: 1024, 0.5, 1024, 0.9):
   * a node updated c times so far moves by (node_tau0 + c)^-node_kappa, lambda at step t by
   * (tau0 + t)^-kappa; tau >= 1, kappa in [0, 1], kappa = 0 = no damping */
  double node_tau0, node_kappa;
  double tau0, kappa;
  uint64_t seed;          /* offset of the first window in the cyclic order */
  /* node-block shards (svils_config node_begin/node_end): nodes per rank block (= n_alloc / world);
   * 0 on a single handle.  Every rank then takes the window at the SAME offset inside its own block,
   * so a global mini-batch is world x batch_nodes nodes; drive it with svils_step_phase(). */
  uint32_t shard_block;
} svils_stochastic;
void svils_stochastic_default(svils_stochastic *cfg, uint32_t batch_nodes);
/* Call after svils_create; allocates the extra accumulator. */
int svils_set_stochastic(svils_handle *h, const svils_stochastic *cfg);
/* Enqueue `nsteps` mini-batch steps; asynchronous; every step advances _iter, writes a
 * likelihood row when _iter % reportfreq == 0 and runs the stop rule, exactly as a sweep does. */
int svils_step(svils_handle *h, uint32_t nsteps);
/* One mini-batch step split at its exchange points, for node-block shards (one process per GPU):
 * A -> all-reduce SVILS_BUF_KVEC_A -> B -> all-gather, for every rank's window, the rows of
 * SVILS_BUF_GAMMA and SVILS_BUF_MPHI and the flag buffers -> EXPAND (Elogpi of the other ranks' window
 * rows from the gathered gamma) -> C -> all-reduce SVILS_BUF_KVEC_C -> D.  Phase A opens the step,
 * phase D closes it.  svils_step_window gives the window of the open (or next) step relative to a
 * rank's block: rows [r*shard_block + begin, r*shard_block + end) for every rank r. */
int svils_step_phase(svils_handle *h, svils_phase phase);
int svils_step_window(svils_handle *h, uint32_t *begin, uint32_t *end);

/* ---- pipelined reports --------------------------------------------------------------------------
 * The reference's loop reports after every sweep under -link-sampling (rfreq = 1, src/main.cc:149-153): a likelihood
 * row, max.txt, communities.txt (src/linksampling.cc:777-786).  Fetching that with svils_get_control / svils_get_rows /
 * svils_get_communities costs a stream synchronisation per report, i.e. per sweep.  A REPORT is the same data as a
 * snapshot taken in stream order: svils_report_enqueue() puts, behind the sweeps enqueued so far, one launch that packs
 * the control block, the likelihood rows [row_first, row_first + row_count) and (with_communities != 0) the community
 * bitmask of the last tagging sweep into a staging slot; a copy stream takes the slot to pinned host memory while the
 * compute stream goes on with the next sweeps.  The host collects it later (svils_report_fetch blocks until it has
 * landed; svils_report_ready does not block).  The caller names the rows because it knows them without asking the
 * device: every sweep whose _iter becomes a multiple of reportfreq records one (fewer only after the stop rule fired --
 * the fetched control block says how many really exist).  At most SVILS_REPORT_SLOTS reports may be outstanding, each
 * of at most SVILS_REPORT_MAX_ROWS rows.  Whole-graph handles only (a sharded run gathers its tags collectively). */
#define SVILS_REPORT_SLOTS 4
#define SVILS_REPORT_MAX_ROWS 64
int svils_report_enqueue(svils_handle *h, uint32_t row_first, uint32_t row_count, int with_communities, int *ticket);
int svils_report_ready(svils_handle *h, int ticket);     /* 1 = landed, 0 = not yet, < 0 = error */
/* ctrl: as svils_get_control at the snapshot; rows: [row_count][10], *nrows = how many of them exist
 * (min(ctrl.rows, row_first + row_count) - row_first); member: [n][k] as svils_get_communities, or NULL.  Frees the slot. */
int svils_report_fetch(svils_handle *h, int ticket, svils_control *ctrl, double *rows, uint32_t *nrows, uint8_t *member);
/* The communities of a report as (node, community) PAIRS -- what communities.txt is made of (src/linksampling.cc:882-917
 * walks the map community -> members) -- instead of the n x k byte matrix (n = 1e6, k = 512: 512 MB and a pass over it per
 * report, against ~1e6 pairs).  tags[2i] = node, tags[2i+1] = community, by ascending node (a node's communities in no
 * particular order).  svils_report_tag_count blocks until the report has landed and says how many pairs it holds (the slot
 * is kept); svils_report_fetch_tags is svils_report_fetch with the pairs in place of `member`: `cap` = room in pairs,
 * *ntags = pairs that exist; if they do not fit nothing is freed and SVILS_ERR_ARG comes back. */
int svils_report_tag_count(svils_handle *h, int ticket, uint64_t *ntags);
int svils_report_fetch_tags(svils_handle *h, int ticket, svils_control *ctrl, double *rows, uint32_t *nrows,
                            uint32_t *tags, uint64_t cap, uint64_t *ntags);
/* the TEST rows of the same report (svils_set_test): test_rows [row_count][10]; call it BEFORE svils_report_fetch (which
 * frees the slot); blocks until the report has landed.  *ntest = rows that exist: one per validation row, minus the one
 * of the stopping sweep. */
int svils_report_test_rows(svils_handle *h, int ticket, double *test_rows, uint32_t *ntest);

/* rows recorded by the in-loop validation_likelihood(): copies rows
 * [first, first+count) (as numbered since create) into out[count][10]. */
int svils_get_rows(svils_handle *h, uint32_t first, uint32_t count, double *rows);

/* save_model() inputs (src/linksampling.cc:804-837): gamma [n][k],
 * lambda [k][2]; any pointer may be NULL. */
int svils_get_state(svils_handle *h, double *gamma, double *lambda, uint32_t *converged);

/* communities of the last tagging sweep (src/linksampling.cc:704-717,882-917):
 * member[n][k] bytes, 1 = node p belongs on line k of communities.txt. */
int svils_get_communities(svils_handle *h, uint8_t *member);
/* the same as (node, community) pairs (see svils_report_fetch_tags); tags == NULL (cap 0): only the count */
int svils_get_community_tags(svils_handle *h, uint32_t *tags, uint64_t cap, uint64_t *ntags);

/* Derived device arrays for parity tests and integrators:
 * which = 0 Elogpi [n][k], 1 Elogbeta [k][2], 2 mphi [n][k],
 *         3 active_comms [n] (uint32), 4 training_links [n] (double). */
int svils_get_aux(svils_handle *h, int which, void *out);

/* Evaluate the kernels' own special functions on a plain array (unit tests):
 * which = 0 digamma(x) [stands for gsl_sf_psi], 1 exp(x) for x <= 0, 2 1/x, 3 ln(x) for x >= 1. */
int svils_debug_eval(svils_handle *h, int which, const double *in, double *out, uint32_t n);

/* ---- measurement --------------------------------------------------------- */
enum {
  SVILS_KERNEL_PHI = 0, SVILS_KERNEL_REDUCE_SUM, SVILS_KERNEL_FINALIZE, SVILS_KERNEL_S3,
  SVILS_KERNEL_VALIDATION /* part of the tail kernel since ABI 2: never timed */,
  SVILS_KERNEL_REDUCE_S, SVILS_KERNEL_TAIL /* likelihood + lambda + stop rule */,
  SVILS_KERNEL_CLASSIFY /* stand-alone link classification (k <= 32; normally fused into the s3 launch) */,
  SVILS_KERNEL_EXCHANGE /* the RCCL collectives of svils_sweep_sharded */,
  SVILS_KERNEL_COUNT
};
/* mask: bit i set = bracket kernel i with hipEvents on the library's stream */
int svils_enable_timing(svils_handle *h, uint32_t kernel_mask);
/* sample only every `period`-th sweep of a svils_sweep() call (default 1 = every sweep): the
 * bracketed sweeps are launched eagerly, the others replay hipGraphs */
int svils_set_timing_period(svils_handle *h, uint32_t period);
/* synchronises; ms[i] = summed duration, launches[i] = launches timed since
 * the last svils_enable_timing call.  Arrays of SVILS_KERNEL_COUNT. */
int svils_get_timing(svils_handle *h, double *ms, uint64_t *launches);
const char *svils_kernel_name(int kernel);
/* How the links of sweeps [first, first+count) (numbered since create) were evaluated
 * (src/linksampling.cc:622-719; the c / d counters of :602,:726): out[count][3] =
 * full softmax, active-set softmax, O(1) shortcut.  The device keeps the last 4096 sweeps. */
int svils_get_sweep_stats(svils_handle *h, uint32_t first, uint32_t count, uint64_t *out);
/* The same three counts summed over exactly those sweeps whose phi launch was bracketed with
 * hipEvents since the last svils_enable_timing call: what the timed launches processed. */
int svils_get_timed_links(svils_handle *h, uint64_t *out3);

/* ---- multi-GPU hooks (one process per GPU; collectives stay with the caller) */
/* The sweep split at its exchange points.  Node-block ownership, whole sweeps (what svils_sweep_sharded issues itself):
 *   A            phi pass over the owned rows                  -> partial `sum[k]` in SVILS_BUF_KVEC_A
 *   B_LIGHT      mean indicators, s1 / s2 partials, tags and the UNSCALED new row of every owned node, written to this
 *                rank's slice of SVILS_BUF_GSTAGE ([world][bmax][ld]: slice r = the rows of rank r's block)
 *   -- exchange 1: all-reduce(SUM) KVEC_A; all-gather of the GSTAGE slices (in place: the send block IS slice `rank`)
 *   EXPAND_ALL   every row, owned or not: annealing scale ones / sum[k], gamma, Elogpi (digamma), the mean indicators of
 *                the other blocks m = (row - alpha) / (n - 1), prune() flags -- computed by every rank from the same
 *                bytes, so flags never cross the links and ONE n-by-k array does
 *   C            s3 pass over this rank's share of the links   -> SVILS_BUF_KVEC_C (s1, s2, s3)
 *   -- exchange 2: all-reduce(SUM) KVEC_C
 *   D            lambda, likelihood, stop rule, annealing switch (replicated)
 * Two exchange points in both phases of a run: the annealing scale is applied behind the row exchange (until ABI 5
 * `sum` had an exchange point of its own while annealing).  Mini-batch steps (svils_step_phase) keep the older split
 * A | B | EXPAND | C | D with the gamma / mphi / packed-flag rows of the windows exchanged in place.
 * svils_sweep() == A,B,C,D with no exchange. */
int svils_sweep_phase(svils_handle *h, svils_phase phase);

/* Node blocks.  Rank r of `world` owns the nodes [bounds[r], bounds[r + 1]); the handle of rank r is created with
 * svils_config.node_begin / node_end = that range.  svils_balance_node_blocks cuts [0, n) so that every block carries
 * the same WORK, not the same number of nodes ("balanced by sum of degree"): cost of a node = its CSR
 * entries (2 per training link it touches) + node_weight (per-node share of the finalise pass in units of one entry;
 * < 0 = the default 0.5).  The reference numbers nodes by first appearance (src/network.cc:10-116), so on its own
 * example graphs the hubs sit at the low ids: equal-count blocks give rank 0 of 8 on ca-AstroPh 2.96 x the mean number
 * of entries.  Pure host code (no device needed); every rank computes the same bounds from the same link list.
 * svils_set_node_blocks declares the blocks of ALL ranks to a handle (bounds[world + 1]; NULL = equal blocks of
 * ceil(n / world) nodes, what svils_comm_init assumes by itself), before or after svils_set_graph, before
 * svils_comm_init.  With caller-given bounds the s3 pass is cut by link count (equal runs of the link list) instead of
 * by node block, and mini-batch steps (which need equal blocks) are refused.  At most 64 ranks. */
int svils_balance_node_blocks(const uint32_t *links, uint64_t nlinks, uint32_t n, int world, double node_weight, uint32_t *bounds);
int svils_set_node_blocks(svils_handle *h, int rank, int world, const uint32_t *bounds);

typedef enum {
  SVILS_BUF_KVEC_A = 0,   /* double[k]   : sum (phase A -> all-reduce SUM)          */
  SVILS_BUF_KVEC_C,       /* double[3k]  : s1,s2,s3 (phase C -> all-reduce SUM)          */
  SVILS_BUF_GAMMA,        /* double[n_pad][ld] rows, all-gather by node block      */
  SVILS_BUF_ELOGPI,       /* double[n_pad][ld]  (re-derived by EXPAND; exchange optional) */
  SVILS_BUF_MPHI,         /* double[n_pad][ld]  (re-derived by EXPAND; exchange optional) */
  SVILS_BUF_CONV,         /* uint32[n_pad] new converged flags                      */
  SVILS_BUF_ACTIVE,       /* uint32[n_pad] active_comms                             */
  SVILS_BUF_AMASK,        /* uint64[n_pad][kw] active-set bitmask                   */
  SVILS_BUF_MEMBER,       /* uint64[n_pad][kw] community bitmask                    */
  SVILS_BUF_XFLAGS,       /* uint32[n_pad][2+2kw] new converged flag, active_comms and active-set
                             bitmask of every row, packed by phase B: all-gather THIS by node block
                             (instead of CONV/ACTIVE/AMASK); PHASE_EXPAND unpacks the others' rows
                             (mini-batch steps only since ABI 6) */
  SVILS_BUF_GSTAGE        /* double[world * bmax][ld]: staging of the row exchange of whole node-block sweeps; slice r
                             (bmax rows, bmax = the largest block) holds the unscaled new rows of rank r's block.
                             row_bytes = ld * 8, bytes = the whole buffer.  Exists once the blocks are declared. */
} svils_buffer;
/* device pointer + geometry of an exchange buffer (valid until destroy) */
int svils_device_buffer(svils_handle *h, svils_buffer which, void **dptr,
                        size_t *bytes, size_t *row_bytes);
/* the hipStream_t the library launches on, as void* */
int svils_stream(svils_handle *h, void **stream);

/* ---- native multi-GPU driver: one process per GPU, RCCL over xGMI ------------------------------
 * The reference has no distributed path; its one reduce analogue is the in-process sum of per-thread
 * partials in MMSBInfer::multithreaded_process (src/mmsbinfer.cc:1770-1827).  Here every process
 * creates its handle on its own GPU with the node block of its rank (svils_config node_begin / node_end =
 * [bounds[rank], bounds[rank + 1]) of svils_balance_node_blocks + svils_set_node_blocks; or, without them, the equal
 * blocks [rank*B, min(n, (rank+1)*B)), B = ceil(n/world) -- mini-batch steps also need n_alloc = world*B) and the
 * library issues the exchanges itself, on the handle's stream, between the phases of a sweep:
 *   phi pass -> light finalise -> {all-reduce(sum), all-gather(unscaled rows)} -> expand (scale, Elogpi, flags of every
 *   row) -> s3 pass -> all-reduce(s1,s2,s3) -> tail (replicated):
 * two exchange points per sweep in both phases of a run, nothing on the host looks at the control block, and runs of
 * sweeps -- collectives included -- replay as hipGraphs under svils_sweep's rule (eager until the handle has run 128
 * sweeps; option sharded_graphs = 0 keeps them eager; a capture that fails once leaves the handle eager).
 * The K-vector all-reduces run on the handle's stream.  When the n-by-k payload is large enough to be pipelined
 * (chunks on a communication stream of their own, each expanded while the next one travels) the row chunks use a
 * SECOND communicator, formed by the same ranks the first time it is needed (rank 0 draws another unique id and
 * broadcasts it over the first one): RCCL serialises the operations of one communicator, so sharing it would put
 * every K-vector all-reduce behind the broadcasts still in flight.
 * librccl is loaded at run time (dlopen) the first time one of these entry points is used. */
#define SVILS_COMM_ID_BYTES 128
/* rank 0: a fresh ncclUniqueId to hand to every rank (any transport: pipe, file, MPI, torch store) */
int svils_comm_unique_id(void *id128);
/* collective over all ranks: ncclCommInitRank on the handle's device */
int svils_comm_init(svils_handle *h, const void *id128, int rank, int world);
/* What the communicator of this handle looks like FROM RCCL'S SIDE (ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice / ncclGetVersion asked of the library that was bound, not echoed from svils_comm_init's
 * arguments): the evidence a launcher prints to show that N ranks on N devices really formed one communicator. */
typedef struct {
  int32_t nranks;        /* ncclCommCount                                                             */
  int32_t rank;          /* ncclCommUserRank                                                          */
  int32_t device;        /* ncclCommCuDevice: the HIP ordinal RCCL bound the communicator to          */
  int32_t version;       /* ncclGetVersion (e.g. 22203), -1 if the bound library has no such entry    */
  int32_t row_comm;      /* 1 once the second communicator (pipelined row exchange) exists            */
  char pci_bus_id[32];   /* hipDeviceGetPCIBusId of `device`                                          */
  char library[256];     /* path of the shared object the nccl* entry points were bound from          */
} svils_comm_info;
int svils_comm_query(svils_handle *h, svils_comm_info *out);
/* Enqueue `nsweeps` sharded sweeps (asynchronous, like svils_sweep).  Collective. */
int svils_sweep_sharded(svils_handle *h, uint32_t nsweeps);
/* Mini-batch steps (svils_set_stochastic with shard_block = B) over the node-block shards, the exchanges issued
 * here: per step all-reduce(sum) -> broadcasts of every rank's window rows (gamma, mphi, packed flags: world x
 * batch_nodes rows, not n) -> expand -> all-reduce(s1,s2,s3) -- "the K-vector lambda and touched gamma rows at
 * the global step".  svils_comm_init may come before or after svils_set_stochastic.  Collective, asynchronous. */
int svils_step_sharded(svils_handle *h, uint32_t nsteps);
/* all-gather of the community bitmasks before svils_get_communities on a sharded handle.  Collective. */
int svils_gather_communities(svils_handle *h);

/* ---- K-sharded sweeps: one process per GPU, every rank holds a column slice of all rows --------
 * The columns of a row are coupled in four places; each is a buffer of partials
 * that must be SUMmed over the ranks between two phases (tests/test_ksharded_protocol.py is the protocol):
 *   svils_set_state (slices) -> phase KINIT_ROWS -> SUM KSH_ROWX -> phase KINIT_EXPAND        (once)
 *   per sweep: phase KDEN -> SUM KSH_DEN -> phase KPHI -> SUM KSH_ROWX -> phase KFIN -> SUM KSH_Q2
 *              -> phase KLAMBDA -> SUM KSH_VDOT -> phase KSTOP
 * svils_sweep_ksharded does exactly that with RCCL all-reduces on the handle's stream (svils_comm_init
 * first; the node block of a K-sharded handle is [0, n)). */
typedef enum {
  SVILS_KPHASE_DEN = 0, SVILS_KPHASE_PHI = 1, SVILS_KPHASE_FIN = 2, SVILS_KPHASE_LAMBDA = 3, SVILS_KPHASE_STOP = 4,
  SVILS_KPHASE_INIT_ROWS = 5, SVILS_KPHASE_INIT_EXPAND = 6,
  SVILS_KPHASE_DENMAX = 7   /* log-domain mode only: before DEN, followed by a MAX (not SUM) of SVILS_KSH_DMAX */
} svils_kphase;
typedef enum { SVILS_KSH_DEN = 0, SVILS_KSH_ROWX = 1, SVILS_KSH_Q2 = 2, SVILS_KSH_VDOT = 3, SVILS_KSH_DMAX = 4,
               /* link_thresh < 1/2 only (argmax tagging, src/linksampling.cc:704-717, src/matrix.hh:521-532): the lowest
                * column attaining the link's maximum, as a double; exchanged with MIN right after SVILS_KSH_DEN.  Such
                * handles always run the log-domain exchange (the maximum is what SVILS_KSH_DMAX carries). */
               SVILS_KSH_EARG = 5 } svils_ksh_buffer;
int svils_ksweep_phase(svils_handle *h, svils_kphase phase);
/* device pointer and length (doubles) of an exchange buffer */
int svils_ksh_buffer_ptr(svils_handle *h, svils_ksh_buffer which, void **dptr, size_t *ndoubles);
/* Log-domain denominators: the product form of the DEN / PHI phases underflows when max_k x_k < -745 for a link (memberships
 * concentrated on different communities at k_total >~ 740); this mode exchanges the per-link max first (phase DENMAX, MAX of
 * SVILS_KSH_DMAX) and sums e^(x - max).  On by default for k_total > 700.  on = 1 / 0 sets it, on < 0 returns the setting. */
int svils_ksh_log_domain(svils_handle *h, int on);
int svils_ksh_init_state(svils_handle *h);           /* collective: the two INIT phases around their all-reduce */
/* On a K-sharded handle svils_validation_row is collective too (partial dot products of the own columns, SUM of
 * SVILS_KSH_VDOT, the log terms in pair order); call it after svils_ksh_init_state or between two sweeps. */
/* save_model / communities of a sharded run (src/linksampling.cc:804-837,882-917): every rank hands `bytes`
 * bytes of host memory (its slice, padded to a common size) and receives world * bytes, rank by rank.
 * Staged through device memory, ncclAllGather on the handle's stream; synchronises.  Collective.
 * A handle without a communicator (world of one) copies send to recv. */
int svils_comm_allgather_host(svils_handle *h, const void *send, void *recv, size_t bytes);
int svils_sweep_ksharded(svils_handle *h, uint32_t nsweeps);   /* collective, asynchronous */
/* Mini-batch steps on a K-sharded handle (svils_set_stochastic with shard_block = 0: every rank holds all nodes and steps
 * through the SAME windows on its own columns).  The phases and exchanges are those of a sweep restricted to the window:
 * while a step is open svils_ksh_buffer_ptr returns the window's share of SVILS_KSH_DEN / DMAX / EARG (indexed by CSR entry
 * in this mode: the entries of the window's rows are one contiguous range) and of SVILS_KSH_ROWX.  With your own
 * collectives: svils_ksweep_phase in the order of a sweep; the first phase opens the step, SVILS_KPHASE_STOP closes it. */
int svils_step_ksharded(svils_handle *h, uint32_t nsteps);      /* collective, asynchronous */

/* ---- options -------------------------------------------------------------------------------------------------------
 * Every tunable of the library is a row of ONE table: key, the SVILS_* environment variable that sets its default, the
 * built-in default, until when it may be changed on a handle, and what it does -- svils_option_table() returns it as
 * tab-separated text (one row per line, a header line first; DESIGN.md section 9 prints it).  The environment is read
 * when svils_create runs and at no other time (SVILS_RCCL_LIBRARY: at the first svils_comm_init of the process);
 * svils_set_option changes one option of one handle afterwards -- options consumed by svils_create ("environment only")
 * or by svils_set_graph ("before svils_set_graph") answer SVILS_ERR_ARG once it is too late.  Values are decimal
 * integers as text.  The reference has one knob of this kind, its command line (src/env.hh:13-110); none of these changes
 * a result beyond rounding (tests/test_gpu_parity.py runs the A/B forms against the oracle). */
int svils_set_option(svils_handle *h, const char *key, const char *value);
int svils_get_option(svils_handle *h, const char *key, char *value, size_t cap);
const char *svils_option_table(void);

const char *svils_last_error(void);
int svils_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
