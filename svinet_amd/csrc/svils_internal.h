// svils_internal.h -- types shared by the C-ABI layer and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svils.h"

namespace svils {

// One unit of work of the phi pass: a contiguous chunk of a node's CSR row.
// Rows longer than the chunk limit are split so a hub (ca-AstroPh: degree 504
// vs mean 22) does not serialise a wavefront; partial accumulators of split
// rows go to `slot` and are combined in fixed order by the finalise kernel.
struct Item {
  uint32_t node;
  uint32_t off;   // offset inside the node's row
  uint32_t len;   // neighbours in this chunk
  int32_t slot;   // -1: row not split, result goes straight to gamma[node]
};

// Device-resident loop state of LinkSampling::infer() /
// validation_likelihood() (src/linksampling.cc:556-790, :966-1050).
struct DevCtrl {
  uint32_t iter;
  int32_t annealing;
  int32_t write_comm;
  int32_t nh;
  double prev_h;
  double max_h;
  int32_t stopped;
  int32_t why;
  uint32_t sweeps_done;
  uint32_t rows;
  unsigned long long links_dense, links_sparse, links_shortcut;       // of the last sweep
  uint32_t parity;  // conv[parity] is the current _converged, conv[parity^1] receives prune()'s
  uint32_t cls_par; // lane-per-link layout: ltot/shist[cls_par] describe the link classes of the CURRENT sweep
  // three-launch sweeps (fused small-K path): the held-out likelihood + stop rule of sweep v_iter has
  // been handed to the next launch (a role of the next phi launch, or k_validate_lpl)
  uint32_t v_pending, v_iter;
  // set (with `stopped`) by a kernel whose bounded in-launch wait ran out -- only possible if the role
  // blocks of a launch are not co-resident (a partitioned or masked device); every host entry that reads
  // the control block turns it into SVILS_ERR_DEVICE
  uint32_t fault;
  // Handles that store no Elogpi (DeviceState::skip_elogpi): the fast phi launch carries no log-domain fall-back; when a link's row
  // product underflows it writes sweeps_done + 1 here, and the launch behind it -- the same pass with the fall-back compiled in,
  // which otherwise returns at once -- redoes the whole pass (gamma is intact: such handles accumulate beside it)
  uint32_t phi_redo;
};

struct Geometry {
  uint32_t n, n_alloc, K, ld, kw, k10;
  uint32_t K0, Kt;   // K-sharded handles: this rank holds columns [K0, K0 + K) of Kt; otherwise 0, K
  uint32_t node_begin, node_end;
  int W, V;  // lanes per row group, doubles per lane (K <= W*V)
};

struct DeviceState {
  // graph
  uint64_t *rowptr;     // [n+1] symmetric CSR of training links
  uint32_t *col;        // row x = {p < x ascending} ++ {q > x in adjacency order}
  uint32_t *upper;      // [n] offset of the first q > x inside row x
  Item *items_phi;      // chunks over owned rows
  uint32_t nitems_phi;
  Item *items_s3;       // chunks over the upper part of owned rows
  uint32_t nitems_s3;
  uint32_t item0_phi, item0_s3;   // first item of the current node window (0 outside mini-batch steps)
  int32_t *split_first; // [n] first slot of a split row or -1
  uint32_t *split_cnt;  // [n]
  uint32_t nslots;
  double *parts;        // [nslots][ld] partial accumulators of split rows
  uint32_t *part_cnt;   // [nslots][ld] partial fmap counts of split rows
  // lane-per-link layout (K <= 32): wave-item w = CSR entries [64w, 64w+64)
  int lpl;              // 1: k_phi_lpl / k_s3_lpl are used
  uint32_t *erow;       // [2L] row (node) of every CSR entry
  uint32_t *links;      // [L][2] the training-link list itself (p<q), for the s3 pass
  uint64_t nlinks;
  uint64_t ent_begin, ent_end;   // owned CSR entries = [rowptr[node_begin], rowptr[node_end])
  uint64_t link_begin, link_end; // owned links (first endpoint in the node block)
  uint64_t lpl_w0;      // first wave-item (ent_begin / 64)
  uint32_t lpl_nitems;  // capacity: wave-items per class list
  // Where k_phi_lpl leaves the pieces of a node's gammanext row (per class list), by the lanes [a, b] the piece covers in
  // its wave-item: a > 0, b < 63 (interior) -> gacc / gacc1 [node]; a > 0, b = 63 (the run goes on in the next item, or
  // ends exactly there) -> ghead[list][node]; a = 0, b < 63 (the rest of a run, or one that starts exactly there) ->
  // gtail[list][node]; a = 0, b = 63 (a whole item of a hub's run) -> slot_f[list][item].  A node has one run per list,
  // hence at most one head and one tail piece: three of the four kinds sit at addresses that depend on the node alone,
  // so k_finalize_lpl requests them before it has seen the run boundaries (one dependent miss less per launch); until
  // the last session of round 3 head and tail pieces lived in per-item slots (slot_f / slot_l).
  double *slot_f;       // [2][lpl_nitems][ld] whole-item pieces
  double *ghead;        // [2][n_alloc][ld]
  double *gtail;        // [2][n_alloc][ld]
  double *gacc1;        // [n_alloc][ld] interior runs of class list 1 (list 0 writes gacc)
  // Per-sweep link classes (k_classify, src/linksampling.cc:622-634): every owned CSR entry is
  //   0 dense  (full softmax), 1 sparse (active-set softmax, _iter > 1000), 2 shortcut (exactly one
  //   endpoint converged, O(1)); the entries of each class are stream-compacted in CSR order.
  uint32_t *cp[2], *cq[2];        // [2L] (p, q) of the class-0 / class-1 entries
  uint16_t *scol;                 // [2L] community column (pc or qc, minus 1) of the class-2 entries
  uint32_t *npos[3];              // [n_alloc+1] class-l entries before row p (exclusive prefix at row starts)
  unsigned long long *tcnt;       // [ntiles] class-0 << 32 | class-1 entry counts per classification tile
  uint32_t *cls_args;             // [8] conv half, sparse flag, ltot/shist half, epoch of the classification in flight,
                                  //     [4] forced (the active-set regime switches), [5] it runs (count pass's verdict)
  // Work-proportional classification: the classes of a link depend on cflag[] of its endpoints and on the sparse
  // regime only.  The finalise pass stores the epoch (sweeps_done + 1) here whenever it changes a node's cflag word;
  // the passes that classify the NEXT sweep's links run only if that epoch is the current one (or the regime switches),
  // otherwise the lists, totals and shortcut histogram of the current sweep simply stay current (cls_par does not flip).
  // Late in a run few sweeps change a flag; at n = 1e6, k = 20 the two passes are ~0.5 ms of a 2.8 ms sweep.
  uint32_t *cls_epoch;            // [1]
  int inject_fault;               // test hook (libsvils_testing.so, option fault_inject): one worker never publishes its tile
  uint32_t *ltot;                 // [2][8] per cls_par: entries of class 0,1,2; entries with q > p of class 0,1,2
  unsigned long long *shist;      // [2][K] per cls_par: class-2 entries per community column
  // `sum[k]` of whole sweeps driven by this library (fold): per-XCD fixed-point accumulators, [2][8][64] per cls_par.
  // Every phi block adds its column sums with ONE 64-bit integer atomic per column (integers: associative, so the
  // total is bit-reproducible whatever the arrival order); the finalise blocks read 8 x K words instead of folding
  // up to 512 partial rows each.  sum[k] <= 2L, scaled by 2^fx_shift so that it fits 62 bits.
  long long *sumfx;
  double fx_scale, fx_inv;
  uint32_t cls_tile;              // raw entries per classification tile (a multiple of 1024)
  uint32_t cls_tile0, cls_ntiles; // tiles covering the owned entries
  uint64_t ent_pad;               // erow / col are padded to this many entries (0xffffffff)
  uint32_t s3_threads;  // block size of k_s3_lpl (1024, or 512 for K > 32)
  int fold;             // 1: consumers sum the producers' per-block partial rows themselves (no k_colreduce)
  int cls_next;         // the s3 / tail launches carry the two classification passes for the NEXT sweep
  // fused3: a sweep is THREE launches -- phi (+ roles: held-out likelihood and stop rule of the previous sweep),
  // finalise, s3 (+ roles: both classification passes; last block: lambda, Elogbeta, loop control)
  int fused3;
  uint32_t nvb;                   // validation-role blocks appended to the phi launch
  uint32_t *s3_ctl;               // [4] arrival ticket of the s3 blocks
  uint32_t *cls_sync;             // [4] [0] arrival ticket of the classification role blocks, [1] prefix-ready epoch
  unsigned long long *tbase;      // [ntiles] exclusive class-0 << 32 | class-1 prefix per tile (fused3 scatter)
  unsigned long long *tpoll;      // [ntiles] epoch << 32 | class-1 << 16 | class-0 counts of 1024-entry tiles (ticket-free hand-off)
  double *gacc0;                  // [n_alloc][ld] phi accumulator of the fused3 path (gamma stays intact until finalise)
  unsigned long long *sweep_stats;  // [sweep_stats_cap][4] ring: dense, sparse, shortcut links and index of each sweep
  uint32_t sweep_stats_cap;
  unsigned long long *stamps;   // [4][1024][8] wall-clock stamps of blocks (builds with -DSVILS_STAMPS only)
  uint32_t *tail_ctl;   // [4] arrival ticket of k_tail's blocks
  double *tail_part;    // [nb_t][4] per-block held-out partial sums of k_tail (nb_t <= SVILS_TAIL_BLOCKS)
  uint32_t nb_t;
  unsigned long long *member_acc; // [n_alloc] tag bits OR-ed during the phi pass (lt_min_deg == 0)
  uint32_t *fcnt;       // [n_alloc][ld] tag counts (lt_min_deg > 0), else null
  // state
  double *gamma;        // [n_alloc][ld]; doubles as gammanext-accumulator inside a sweep
  // Node-block sweeps (svils_sweep_sharded): the staging of the row exchange, [world][bmax][ld] -- the light finalise pass
  // writes the UNSCALED new rows of the owned block into slice `rank` (gown = its first row), the all-gather (or the
  // chunked broadcasts) fills the other slices, k_expand_all turns every row into gamma / Elogpi / flags
  double *gstage, *gown;
  int light;            // 1: this launch of the finalise pass is the light one
  // Lane-per-link layout: the row stores of the finalise pass and the piece stores of the phi pass go out write-through
  // (agent-scope relaxed atomic stores = `global_store_dwordx2 ... sc1`) when the n-by-k state is between 1 and 8 MB:
  // what a launch leaves DIRTY in the XCD L2s is written back at its end before the next launch may read it from
  // another XCD (~0.6 us per MB); written through, the lines are clean by then.  Measured per sweep, one box, three
  // alternating repetitions (profiles/r05r_ab_write_through_gated.txt): ca-AstroPh K=8 / 20 / 32 (1.1 / 2.9 / 4.6 MB per
  // array) -3.3 / -3.3 / -3.0 %; LFR K=28 (0.2 MB) +0.4 %; n=1e5 K=20 (16 MB) +0.3 %, n=4e5 K=20 (64 MB) +1.1 %: hence
  // the two bounds.  Same values stored either way: results are bit-identical.
  int wt;
  uint32_t fin_waves;   // lane-per-link finalise launch: waves per block (lpl_finalize_waves: 12 selects the 768-thread variant at NC = 4)
  int shard_c;          // 1 (node-block sweeps, K <= 32): the last s3 block materialises this rank's s3 in kvec_c (no k_colreduce)
  double *gacc;         // where the phi pass accumulates gammanext: == gamma for full sweeps, a separate
                        // [n_alloc][ld] buffer in mini-batch mode (the old gamma row is blended in)
  uint32_t *ncnt;       // [n_alloc] mini-batch mode: number of updates each node has received
  double *s12run;       // [2K]     mini-batch mode: running s1, s2 over the stored mphi rows
  double *elogpi;       // [n_alloc][ld]
  double *epi;          // [n_alloc][ld] exp(Elogpi), K > 56 only (k_phi<V, false, true>); null otherwise
  // 57 <= K <= 512 with exp(Elogpi) rows and link_thresh >= 1/2: NOTHING in a sweep reads Elogpi -- the phi pass multiplies
  // exp(Elogpi) rows, the likelihood and the s3 pass read gamma -- so the finalise / expand passes do not store it (one n-by-k
  // write less per sweep: 4.1 of 16.4 GB of the finalise launch at n = 1e6, k = 512).  The phi pass's underflow fall-back
  // (a row product below 1e-280: reachable at K <= 512 only in degenerate annealing states, psi(1 / 2K) = -1024) re-derives the two
  // rows from gamma -- NOT in the launch every sweep runs (the mere presence of that code, inline or as a call, costs the phi kernel
  // 24 - 61 spilled VGPRs and a stack: ca-AstroPh K = 200 phi 88 -> 130 us, profiles/r07j): the fast launch only raises
  // DevCtrl::phi_redo, and a second launch with the fall-back compiled in, which otherwise returns at once, redoes the pass.
  // That is possible because such a handle accumulates gammanext in the buffer Elogpi used to occupy (gacc = elogpi) instead of
  // in place: gamma stays intact during the phi pass.  Taken where the n-by-k state is at least 256 MB (below, the state is
  // cache-resident and the second launch's ~2 us outweigh the write saved; option skip_elogpi forces either way).
  // svils_get_aux(0) / SVILS_BUF_ELOGPI compute a view on demand (k_dir_exp).
  int skip_elogpi;
  // K-sharded sweeps (svils_ksh.h): what crosses ranks, each buffer summed over the ranks between two phases
  int ksh;              // 1: the handle holds a column slice
  uint32_t *elink;      // [2L]  training-link index of every CSR entry
  double *den;          // [L]   softmax denominators of the links (partial -> SUM -> total)
  int ksh_log;          // 1: log-domain denominators (max, then shifted sum): two L-sized exchanges, no underflow
  double *dmax;         // [L]   log-domain mode: max_k x_k of the links (partial -> MAX -> total)
  int ksh_ent;          // 1 (mini-batch steps): den / dmax / earg are indexed by CSR entry, every entry of the window computes its own
  int ksh_lowt;         // 1: link_thresh < 1/2 -- tags go to the first strict maximum of phi (argmax over ALL columns)
  double *earg;         // [L]   ksh_lowt: lowest column (global index) attaining the link's max (partial -> MIN -> total)
  double *rowx;         // [n][3] row sum of the new gamma, active-community count, sum of (community + 1) over them
  double *q2v;          // [Kt]  quirk Q2 contributions that belong to another rank's column
  double *vdot;         // [nv]  partial sum_k gamma_p gamma_q beta_k of the held-out pairs
  double *part_q2;      // [nb_c] per-block partial of this rank's outgoing Q2 contribution
  double *mphi;         // [n_alloc][ld]
  // Whole sweeps driven by this library do not STORE the mean indicators: m = acc / tl is a function of the row the
  // finalise pass writes anyway, gamma = (alpha + acc (n-1)/tl) * scale  =>  m = (gamma * iscale - alpha) / (n-1)
  // (as k_expand derives it for rows another rank owns), so the s3 pass reads gamma rows and one n-by-k write per
  // sweep disappears (the launch boundary behind the finalise pass costs ~0.6 us per MB it leaves dirty at small
  // sizes; 4.1 GB of HBM writes per sweep at n = 1e6, k = 512).  iscale[k] = sum[k] / ones while annealing, else 1:
  // written by the finalise pass of the sweep, so it always matches the gamma in memory.  The array is brought up
  // to date on demand (k_mphi_from_gamma) when something wants the stored form (mini-batch steps, phase-split
  // sweeps, svils_get_aux).
  int derive_m;
  double *iscale;       // [K]
  uint32_t *conv;       // [2][n_alloc]
  uint32_t *active_cnt; // [n_alloc]
  // What the link classification needs of an endpoint, in ONE BYTE: the latest converged flag (what prune() wrote last;
  // 0 or community + 1 <= 56 where a classification exists) | 0x80 when active_cnt < K / 10 (cflag_pack).  Written next to
  // conv / active_cnt by whoever writes those; a second random gather per endpoint (active_cnt) cost the classification
  // passes 4 us per sweep on ca-AstroPh once _iter > 1000.  One byte, not a word: the passes gather it at random for
  // every CSR entry, and a table of n bytes (1 MB at n = 1e6) stays in every XCD's L2 next to the s3 / tail launch's own
  // row traffic where a table of words does not.
  uint8_t *cflag;       // [n_alloc]
  uint64_t *amask;      // [n_alloc][kw] lane-layout bitmask of _active_k
  uint64_t *member;     // [n_alloc][kw] lane-layout bitmask of communities
  uint32_t *xflags;     // [n_alloc][xf_ld] conv (new), active_cnt, amask words of every row, packed: ONE buffer to
                        // all-gather after phase B; PHASE_EXPAND unpacks the other ranks' rows
  uint32_t xf_ld;       // 2 + 2 * kw
  double *lambda;       // [K][2]
  double *elogbeta;     // [K][2]
  // K-vectors / partials
  double *part_a;       // [nb_a][K]      per-block partial of `sum`
  double *part_b;       // [nb_b][2K]     per-block partials of s1,s2
  double *part_c;       // [nb_c][K]      per-block partial of s3
  double *kvec_a;       // [K]            sum
  double *kvec_c;       // [3K+4]         s1,s2,s3
  uint32_t nb_a, nb_b, nb_c;
  unsigned long long *part_links;  // [nb_a][3] per-block dense/sparse/shortcut link counts
  // validation
  uint32_t *vpairs;     // [nv][3]
  double *uval;         // [nv]
  uint32_t nv;
  double *rows;         // [rows_cap][10]
  uint32_t rows_cap;
  DevCtrl *ctrl;
  const double *logtab;      // [128][2] {1/c_i, ln c_i} for log_tab()
};

// The node blocks of all ranks of a node-block run: rank r owns nodes [bounds[r], bounds[r + 1]), balanced by work
// (svils_balance_node_blocks), not equal in size; passed by value to the kernels that walk the staged rows.
constexpr int SVILS_MAX_WORLD = 64;
struct Blocks {
  uint32_t world, bmax;                 // ranks; rows of the largest block = rows of a staging slice
  uint32_t chunk, nchunks;              // this launch covers chunk `chunk` of `nchunks` of EVERY block (pipelined exchange)
  uint32_t bounds[SVILS_MAX_WORLD + 1];
};
// rows [lo, hi) of a block of `size` rows that chunk c of C covers
__host__ __device__ inline void chunk_range(uint32_t size, uint32_t c, uint32_t C, uint32_t *lo, uint32_t *hi) {
  *lo = (uint32_t)((uint64_t)size * c / C);
  *hi = (uint32_t)((uint64_t)size * (c + 1) / C);
}

struct Params {
  uint64_t ones;
  double alpha, eta0, eta1, epsilon, link_thresh;
  uint32_t lt_min_deg, reportfreq;
  int32_t use_validation_stop;
  double ones_prob, zeros_prob;
  int32_t sparse_after;   // active-set branch when _iter > sparse_after (1000, src/linksampling.cc:634)
  // mini-batch (Robbins-Monro) steps, svils_step(); all neutral for full sweeps
  int32_t stoch;          // 1: this launch is a mini-batch step over the node window [node_begin, node_end)
  double tau0, kappa;     // step size of a node that has been updated c times: (tau0 + c)^-kappa (node_tau0/node_kappa)
  double rho_lambda;      // step size of the global lambda update of this step
  double scale_a;         // 2L / (CSR entries of the window): window `sum` -> estimate of the full `sum`
  double scale_c;         // L / (links whose first endpoint is in the window): window s3 -> estimate of s3
};

// launchers (svils_device.hip); all asynchronous on `s`
constexpr uint32_t SVILS_FOLD_ROWS = 512;   // most per-block partial rows a consumer folds itself
constexpr uint32_t SVILS_TAIL_BLOCKS = 1024;   // most blocks of k_tail (held-out pairs); its last block adds their partials
bool use_lpl(uint32_t K);
int lpl_phi_waves(uint32_t K);
uint32_t lpl_phi_resident_blocks(uint32_t K, int device);
void launch_classify(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_finalize_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
uint32_t rpw_resident_blocks(const Geometry &g, int which /*0 phi, 1 s3, 2 finalize*/, int device);
void launch_phi_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_s3_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_phi(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_reduce_a(const Geometry &g, const DeviceState &d, hipStream_t s);
void launch_finalize(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_s3(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_reduce_c(const Geometry &g, const DeviceState &d, hipStream_t s);
void launch_validation(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_tail(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
uint32_t tail_blocks(const Geometry &g, uint32_t nv);
uint32_t lpl_cls_blocks(const DeviceState &d);
void launch_validate_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
uint32_t lpl_validation_blocks(const Geometry &g, uint32_t nv, uint32_t K);
uint32_t lpl_s3_threads(uint32_t K, uint64_t nlinks);   // threads per block of the s3 launch for this many links
uint32_t lpl_finalize_waves(uint32_t K, uint64_t nodes, uint32_t cus);   // waves per block of the finalise launch (8 or 12)
int lpl_finalize_group(uint32_t K);
uint32_t lpl_finalize_resident_blocks(uint32_t K, int device);
uint32_t lpl_scatter_blocks(const DeviceState &d);
uint32_t lpl_s3_resident_blocks(uint32_t K, int device, uint32_t threads, int assume_cus);
void launch_carry_flags(const Geometry &g, const DeviceState &d, hipStream_t s);
void launch_expand_window(const Geometry &g, const DeviceState &d, const Params &p, uint32_t wb, uint32_t we,
                          uint32_t block, uint32_t my_rank, uint32_t world, hipStream_t s);
void launch_expand(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_expand_all(const Geometry &g, const DeviceState &d, const Params &p, const Blocks &b, hipStream_t s);
void launch_expand_chunk(const Geometry &g, const DeviceState &d, const Params &p, uint32_t xb, uint32_t xe, uint32_t block,
                         hipStream_t s);
void launch_dir_exp(const Geometry &g, const DeviceState &d, hipStream_t s);
void launch_lambda_exp(const Geometry &g, const DeviceState &d, hipStream_t s);
void launch_debug_eval(const DeviceState &d, int which, const double *in, double *out, uint32_t n, hipStream_t s);
// the byte of cflag[]: flag values beyond 127 only occur for K > 127, where nothing reads the table
__host__ __device__ inline uint8_t cflag_pack(uint32_t conv, bool few_active) {
  return (uint8_t)((conv < 127u ? conv : 127u) | (few_active ? 0x80u : 0u));
}
void launch_cflag_rebuild(const Geometry &g, const DeviceState &d, hipStream_t s);
void launch_mphi_from_gamma(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s);
void launch_row_only(const Geometry &g, const DeviceState &d, const Params &p, double *row_out,
                     hipStream_t s);
bool pick_layout(uint32_t K, int *W, int *V);
void launch_ksh_phase(const Geometry &g, const DeviceState &d, const Params &p, int phase, hipStream_t s);   // svils_ksh.h
// k owned by (lane-in-group lw, register v) for layout (W,V); host copy of the device mapping
inline uint32_t kmap_host(int W, int V, int lw, int v) {
  return V == 1 ? (uint32_t)lw : (uint32_t)(2 * ((v >> 1) * W + lw) + (v & 1));
}

}  // namespace svils
