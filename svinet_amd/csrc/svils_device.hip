// svils_device.hip -- gfx950 (MI355X, CDNA4) kernels for one sweep of svinet's
// link-sampling inference loop.  fp64 throughout; no atomics on floating-point
// data (every reduction has a fixed order => bit-reproducible run to run).
//
// The reference loop (src/linksampling.cc:600-761) is push-style: for each
// link (p,q) add phi to gammanext[p] AND gammanext[q].  Here it is pull-style
// over a symmetric CSR: the wavefront that owns node x walks x's row, computes
// phi for each incident link and accumulates only gammanext[x] in registers.
// Every undirected link is evaluated twice, nothing is scattered, node
// ownership makes multi-GPU sharding trivial, and HBM traffic per link drops
// from (2 row reads + 2 row RMWs) to (2 row reads).
//
// Register layout of a K-vector: a "group" of W lanes (W = 8..64, 64/W groups
// per wavefront) holds one row; each lane keeps V doubles.  V == 1: lane l
// owns k = l.  V >= 2: lane l owns the double2 chunks c = j*W + l, i.e.
// k = 2c, 2c+1, so each global load instruction moves W*16 contiguous bytes.
#include <hip/hip_runtime.h>
#include <math.h>

#include "svils_cls.h"
#include "svils_devutil.h"

namespace svils {

// block-level reduction of per-lane K-vector partials into one row of `out`:
// out[k] = sum over the block's 4 wavefronts and 64/W groups, fixed order.
template <int W, int V, int NV>
__device__ __forceinline__ void block_reduce_store(double (&part)[NV][V], double *__restrict__ out,
                                                   uint32_t K, double *lds /*[NV][V][64]*/) {
  constexpr int G = 64 / W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wavefronts add their partials one after the other (order 0,1,2,3)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const int i = (a * V + v) * 64 + lane;
          lds[i] = (w == 0) ? part[a][v] : lds[i] + part[a][v];
        }
    }
    __syncthreads();
  }
  // then the 64/W groups (order 0..G-1); thread t handles (a, v, lw) triples
  for (int idx = threadIdx.x; idx < NV * V * W; idx += blockDim.x) {
    const int lw = idx % W, v = (idx / W) % V, a = idx / (W * V);
    const int k = kmap<W, V>(lw, v);
    if ((uint32_t)k < K) {
      double s = 0.0;
      for (int g = 0; g < G; ++g) s += lds[(a * V + v) * 64 + g * W + lw];
      out[(size_t)a * K + k] = s;
    }
  }
}

// exp(Elogpi) next to Elogpi for the row-per-wavefront phi kernel (K > 56): exp(a + b + c) = e^a e^b e^c
// turns the K exps of every (link, direction) into K multiplies of per-node values; all three exponents
// are <= 0, so the factors lie in (0, 1] and the product underflows exactly when the exp of the sum would.
// Padding columns hold 0.  (Null for K <= 56: the lane-per-link kernels do not use it.)
template <int W, int V>
__device__ __forceinline__ void store_epi(const DeviceState &d, uint32_t p, int lw, uint32_t ld, uint32_t K,
                                          const double (&el)[V]) {
  if (!d.epi) return;
  double e[V];
#pragma unroll
  for (int v = 0; v < V; ++v) e[v] = (uint32_t)kmap<W, V>(lw, v) < K ? exp_neg(el[v]) : 0.0;
  store_row<W, V>(d.epi + (size_t)p * ld, lw, ld, e);
}

// The rare path of the product form on a handle that stores no Elogpi (DeviceState::skip_elogpi): x_k = Elogpi[p][k] +
// Elogbeta[k][0] + Elogpi[q][k] of this lane's columns, re-derived from the gamma rows (psi(gamma) - psi(row sum), the row sums
// added in the order k_finalize adds them: the values it would have stored) into the wavefront's LDS scratch xl[v][lane].
// Only the REDO launch (k_phi<V, false, true, 2>) contains it: in the launch every sweep runs its mere presence -- inline or as a
// call -- cost 24 - 61 spilled VGPRs and a stack (ca-AstroPh K = 200: phi 88 -> 130 us with the code never executed).
template <int W, int V>
__device__ __noinline__ void elogpi_pair_from_gamma(const double *__restrict__ gamma, const double *__restrict__ elogbeta, uint32_t p,
                                                    uint32_t q, uint32_t ld, uint32_t K, int lw, const double2 *logtab, double *xl) {
  double sp = 0.0, sq = 0.0;
#pragma unroll 1
  for (int v = 0; v < V; ++v) {
    const int k = kmap<W, V>(lw, v);
    if ((uint32_t)k < K) { sp += gamma[(size_t)p * ld + k]; sq += gamma[(size_t)q * ld + k]; }
  }
  const double psp = digamma(group_sum<W>(sp), logtab), psq = digamma(group_sum<W>(sq), logtab);
#pragma unroll 1
  for (int v = 0; v < V; ++v) {
    const int k = kmap<W, V>(lw, v);
    double t = NEG_INF;
    if ((uint32_t)k < K)
      t = ((digamma(gamma[(size_t)p * ld + k], logtab) - psp) + elogbeta[2 * k]) + (digamma(gamma[(size_t)q * ld + k], logtab) - psq);
    xl[v * 64 + lw] = t;
  }
}

// ============================================================== phi pass (A6)
// src/linksampling.cc:605-725, pull-style, K > 32 (K <= 32 uses k_phi_lpl).  One
// wavefront per Item (a chunk of <= 32 neighbours of one node), all 64 lanes on one
// row, V doubles per lane.  The chunk's column indices and converged flags are
// fetched with one coalesced load each, and the next neighbour's Elogpi row is in
// flight while the current one is reduced (two rows per wave in flight).
// FB (product form only): what a link does whose row product underflows -- 0: the log-domain rows from the stored Elogpi; handles
// that store none (DeviceState::skip_elogpi): 1 = the launch every sweep runs: raise DevCtrl::phi_redo, nothing else; 2 = the redo
// launch behind it: returns at once unless the flag is this sweep's, else the whole pass again with the rows re-derived from gamma.
template <int V, bool LOWT, bool EPI, int FB = 0>
// waves per SIMD asked of the compiler.  V = 16 (K = 513..1024): two -- the kernel then keeps 256 VGPRs and spills ~90 to
// scratch, which still beats one 512-register wave per SIMD (ca-AstroPh K=640 phi 576 -> 397 us, K=1024 574 -> 503 us; n=2e5
// K=640 5.5 -> 4.6 ms, n=1e5 K=1024 unchanged; three waves per SIMD: 4x slower) -- profiles/r02_ubench_and_rejected_variants.txt
__global__ __launch_bounds__(256, (V <= 2 ? 6 : V == 4 ? 5 : V == 8 ? 3 : V <= 16 ? 2 : 1)) void k_phi(Geometry geo, DeviceState d, Params prm) {
  constexpr int W = 64;
  constexpr bool PROD = EPI && !LOWT;   // product form on exp(Elogpi) rows, else exps of sums of Elogpi rows
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  if constexpr (FB == 2) {
    if (ctrl->phi_redo != ctrl->sweeps_done + 1u) return;   // no link of this sweep's fast launch underflowed: nothing to redo
  }
  __shared__ double lds[V * 64];
  // (the redo launch evaluates psi itself and parks the re-derived x_k in LDS)
  __shared__ double2 logtab[FB == 2 ? 128 : 1];
  __shared__ double xlds[FB == 2 ? 4 * V * 64 : 1];
  if constexpr (FB == 2) { load_logtab(logtab, d.logtab); __syncthreads(); }
  // wave-uniform by construction; said so explicitly, or the compiler keeps the item loop and the neighbour loop
  // under exec masks with vector compares (it cannot see that threadIdx.x >> 6 is the same in all 64 lanes)
  // (measured: V = 4: phi -10 % and 96 instead of 111 VGPRs; V = 8: neutral; V <= 2: +7 %, left alone)
  constexpr bool UNI = V >= 4;
  auto uni = [](int x) { return UNI ? __builtin_amdgcn_readfirstlane(x) : x; };
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int lw = lane;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool write_comm = ctrl->write_comm != 0;
  const bool sparse_iter = (long long)ctrl->iter > (long long)prm.sparse_after;  // _iter > 1000, src/linksampling.cc:634
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  const double *__restrict__ elogpi = PROD ? d.epi : d.elogpi;

  int kidx[V];
  bool kval[V];
  double eb[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    kidx[v] = kmap<W, V>(lw, v);
    kval[v] = (uint32_t)kidx[v] < K;
    eb[v] = kval[v] ? d.elogbeta[2 * kidx[v]] : NEG_INF;   // -inf masks the padding columns
    if constexpr (PROD) eb[v] = exp_neg(eb[v]);             // ... 0 in the product form
  }
  double csum[1][V];
#pragma unroll
  for (int v = 0; v < V; ++v) csum[0][v] = 0.0;
  unsigned long long n_dense = 0, n_sparse = 0, n_short = 0;

  for (uint32_t it = blockIdx.x * 4 + wave; it < d.nitems_phi; it += gridDim.x * 4) {
    const Item item_ = d.items_phi[d.item0_phi + it];
    Item item;   // scalar registers
    item.node = (uint32_t)uni((int)item_.node);
    item.off = (uint32_t)uni((int)item_.off);
    item.len = (uint32_t)uni((int)item_.len);
    item.slot = uni(item_.slot);
    const uint32_t p = item.node;
    const uint64_t base = d.rowptr[p] + item.off;
    const uint32_t len = item.len;   // <= 64 (chunk limit 32)
    // one coalesced load of the chunk's neighbours and one gather of their flags
    uint32_t mycol = 0, myconv = 0;
    if ((uint32_t)lane < len) {
      mycol = d.col[base + lane];
      myconv = conv[mycol];
    }
    const uint32_t pc = (uint32_t)uni((int)conv[p]);
    // Elogpi[p][k] + Elogbeta[k][0], once per item; x_k = this + Elogpi[q][k] (the reference adds the
    // two Elogpi terms first, :686 -- a different rounding of the same sum, one add per column saved)
    double ap[V];
    load_row<W, V>(elogpi + (size_t)p * ld, lw, ld, ap);
#pragma unroll
    for (int v = 0; v < V; ++v) ap[v] = PROD ? ap[v] * eb[v] : ap[v] + eb[v];   // padding columns -> 0 / -inf
    double acc[V];
    uint32_t cnt[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { acc[v] = 0.0; cnt[v] = 0; }
    uint32_t p_active = 0;
    if (sparse_iter) p_active = d.active_cnt[p];

    double rcur[V], rnext[V];
    {
      const uint32_t q0 = __builtin_amdgcn_readlane(mycol, 0);
      const uint32_t qc0 = __builtin_amdgcn_readlane(myconv, 0);
      if (len > 0 && ((pc != 0) == (qc0 != 0))) load_row<W, V>(elogpi + (size_t)q0 * ld, lw, ld, rcur);
    }
    for (uint32_t j = 0; j < len; ++j) {
      const uint32_t q = __builtin_amdgcn_readlane(mycol, j);
      const uint32_t qc = __builtin_amdgcn_readlane(myconv, j);
      // prefetch the next neighbour's row (skipped when that link takes the O(1) shortcut)
      if (j + 1 < len) {
        const uint32_t q1 = __builtin_amdgcn_readlane(mycol, j + 1);
        const uint32_t qc1 = __builtin_amdgcn_readlane(myconv, j + 1);
        if ((pc != 0) == (qc1 != 0)) load_row<W, V>(elogpi + (size_t)q1 * ld, lw, ld, rnext);
      }
      const bool count_me = q > p;  // count each undirected link once
      if ((pc != 0) != (qc != 0)) {
        // exactly one endpoint converged: src/linksampling.cc:622-631
        const int c = (int)(pc ? pc : qc) - 1;
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (kidx[v] == c) acc[v] += 1.0;
        if (count_me && lane == 0) n_short++;
      } else {
        bool sparse = false, empty_union = false;
        if (sparse_iter) {
          const uint32_t q_active = d.active_cnt[q];
          sparse = p_active < geo.k10 && q_active < geo.k10;
          empty_union = sparse && (p_active | q_active) == 0u;   // (both bitmasks are the nodes' exact active sets on this branch)
        }
        // x_k = Elogpi[p][k] + Elogpi[q][k] + Elogbeta[k][0]; padding columns and, on the active-set
        // path, columns outside the union -> -inf
        auto xk = [&](int v) {
          double t = ap[v] + rcur[v];
          if (sparse) {
            const uint64_t um = d.amask[(size_t)p * geo.kw + v] | d.amask[(size_t)q * geo.kw + v];
            t = ((um >> lw) & 1ull) ? t : NEG_INF;
          }
          return t;
        };
        if constexpr (PROD) {
          // exp(x_k) = e^Elogpi[p][k] e^Elogbeta[k][0] * e^Elogpi[q][k]: one multiply per column (every
          // exponent is <= 0, so nothing overflows and no max shift is needed): exp(x_k) / sum_j exp(x_j)
          // with one cross-lane reduction.  Only when the whole row underflows (sum below 1e-280; also the
          // empty active-set union) is it redone in the log domain with the shift.
          double s = 0.0;
          double e[V];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            e[v] = ap[v] * rcur[v];
            if (sparse) {
              const uint64_t um = d.amask[(size_t)p * geo.kw + v] | d.amask[(size_t)q * geo.kw + v];
              e[v] = ((um >> lw) & 1ull) ? e[v] : 0.0;
            }
            s += e[v];
          }
          s = group_sum<W>(s);
          bool live = true;
          if (FB == 1 && empty_union) {
            live = false;       // an empty active-set union contributes nothing (:642-664): a plain zero, not an underflow
          } else if (FB == 1 && s < 1e-280) {
            if (lane == 0) ctrl->phi_redo = ctrl->sweeps_done + 1u;   // the launch behind this one redoes the pass (see FB)
            live = false;
          } else if (FB != 1 && s < 1e-280) {   // wave-uniform, rare: the log-domain rows, column by column to
                                                 // keep the registers of the common path (two passes: max, then exp)
            double *xl = xlds + (size_t)wave * V * 64;
            if constexpr (FB == 2) elogpi_pair_from_gamma<W, V>(d.gamma, d.elogbeta, p, q, ld, K, lw, logtab, xl);
            auto xlog = [&](int v) {
              double t = NEG_INF;
              if (kval[v]) {
                if constexpr (FB == 2) t = xl[v * 64 + lw];
                else t = (d.elogpi[(size_t)p * ld + kidx[v]] + d.elogbeta[2 * kidx[v]]) + d.elogpi[(size_t)q * ld + kidx[v]];
              }
              if (sparse) {
                const uint64_t um = d.amask[(size_t)p * geo.kw + v] | d.amask[(size_t)q * geo.kw + v];
                t = ((um >> lw) & 1ull) ? t : NEG_INF;
              }
              return t;
            };
            double m = NEG_INF;
#pragma unroll 1
            for (int v = 0; v < V; ++v) m = fmax(m, xlog(v));
            m = group_max<W>(m);
            live = (m != NEG_INF);   // an empty active-set union contributes nothing (:642-664)
            s = 0.0;
            if (live) {
#pragma unroll
              for (int v = 0; v < V; ++v) {
                e[v] = exp_neg(xlog(v) - m);
                s += e[v];
              }
              s = group_sum<W>(s);
            }
          }
          if (live) {
            const double inv = fast_rcp(s);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fma(e[v], inv, acc[v]);
            // community tagging, src/linksampling.cc:668-681,704-717: tag the first strict maximum
            // of phi if it exceeds link_thresh.  With link_thresh >= 1/2 (this instantiation) a phi
            // above the threshold IS the strict maximum and is unique, so no argmax is needed.
            if (write_comm) {
              const double ts = prm.link_thresh * s;
#pragma unroll
              for (int v = 0; v < V; ++v) cnt[v] += (e[v] > ts) ? 1u : 0u;
            }
          }
        } else if constexpr (!LOWT) {
          // Every x_k is <= 0 (Elogpi and Elogbeta are expectations of logs of probabilities), so the
          // softmax needs no max shift to stay finite: exp(x_k) / sum_j exp(x_j) directly -- one
          // cross-lane reduction instead of two.  Only when the whole row underflows (sum below
          // 1e-280; also the empty active-set union) is it redone with the shift.
          double s = 0.0;
          double e[V];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            e[v] = exp_neg(xk(v));   // exp_neg(-inf) == 0
            s += e[v];
          }
          s = group_sum<W>(s);
          bool live = true;
          if (s < 1e-280) {   // wave-uniform
            double m = NEG_INF;
#pragma unroll
            for (int v = 0; v < V; ++v) m = fmax(m, xk(v));
            m = group_max<W>(m);
            live = (m != NEG_INF);   // an empty active-set union contributes nothing (:642-664)
            s = 0.0;
            if (live) {
#pragma unroll
              for (int v = 0; v < V; ++v) {
                e[v] = exp_neg(xk(v) - m);
                s += e[v];
              }
              s = group_sum<W>(s);
            }
          }
          if (live) {
            const double inv = fast_rcp(s);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fma(e[v], inv, acc[v]);
            // community tagging, src/linksampling.cc:668-681,704-717: tag the first strict maximum
            // of phi if it exceeds link_thresh.  With link_thresh >= 1/2 (this instantiation) a phi
            // above the threshold IS the strict maximum and is unique, so no argmax is needed.
            if (write_comm) {
              const double ts = prm.link_thresh * s;
#pragma unroll
              for (int v = 0; v < V; ++v) cnt[v] += (e[v] > ts) ? 1u : 0u;
            }
          }
        } else {
          double x[V];
          double m = NEG_INF;
#pragma unroll
          for (int v = 0; v < V; ++v) { x[v] = xk(v); m = fmax(m, x[v]); }
          m = group_max<W>(m);
          if (m != NEG_INF) {  // an empty active-set union contributes nothing (:642-664)
            double s = 0.0;
            bool ismax[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
              ismax[v] = (x[v] == m);
              x[v] = exp_neg(x[v] - m);   // exp_neg(-inf) == 0
              s += x[v];
            }
            s = group_sum<W>(s);
            const double inv = fast_rcp(s);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fma(x[v], inv, acc[v]);
            // link_thresh < 1/2: several phi may exceed it; the first strict maximum of phi is the
            // first k with x_k == max, and its phi is 1/s
            if (write_comm && inv > prm.link_thresh) {
              int best = 0x7fffffff;
#pragma unroll
              for (int v = 0; v < V; ++v)
                if (ismax[v]) best = min(best, kidx[v]);
#pragma unroll
              for (int o = 1; o < W; o <<= 1) best = min(best, __shfl_xor(best, o, 64));
#pragma unroll
              for (int v = 0; v < V; ++v)
                if (kidx[v] == best) cnt[v]++;
            }
          }
        }
        if (count_me && lane == 0) { if (sparse) n_sparse++; else n_dense++; }
      }
#pragma unroll
      for (int v = 0; v < V; ++v) rcur[v] = rnext[v];
    }

#pragma unroll
    for (int v = 0; v < V; ++v) csum[0][v] += acc[v];
    if (item.slot < 0) {
      store_row<W, V>(d.gacc + (size_t)p * ld, lw, ld, acc);
      if (write_comm) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const unsigned long long b = __ballot(kval[v] && cnt[v] > prm.lt_min_deg);
          if (lane == 0) d.member[(size_t)p * geo.kw + v] = b;
        }
      }
    } else {
      store_row<W, V>(d.parts + (size_t)item.slot * ld, lw, ld, acc);
      if (write_comm) {
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (kval[v]) d.part_cnt[(size_t)item.slot * ld + kidx[v]] = cnt[v];
      }
    }
  }

  // per-block partial of `sum` (src/linksampling.cc:625,630,663,700 summed per node)
  block_reduce_store<W, V, 1>(csum, d.part_a + (size_t)blockIdx.x * K, K, lds);
  // link statistics (integers: order-free)
  __shared__ unsigned long long lcnt[3 * 4];
  block_store_link_counts(n_dense, n_sparse, n_short, d.part_links, lcnt, 4);
}

// ===================================================== column reduce of partials
// out[c] = sum_b part[b][c] in a fixed order: 4 columns x 64 row-segments per block.
// Up to two jobs per launch (blockIdx.x < nblk0 -> job 0).
struct ReduceJob {
  const double *part;
  double *out;
  uint32_t nb, ncols;
};
__global__ __launch_bounds__(256) void k_colreduce(ReduceJob j0, ReduceJob j1, uint32_t nblk0,
                                                   const DevCtrl *ctrl) {
  // the stop flag is requested beside the partial rows and looked at before the store (a test first was a round trip of
  // its own in front of the rows': these launches are two round trips long)
  const uint32_t stopped = ctrl->stopped;
  __shared__ double lds[64][5];
  const ReduceJob j = blockIdx.x < nblk0 ? j0 : j1;
  const uint32_t blk = blockIdx.x < nblk0 ? blockIdx.x : blockIdx.x - nblk0;
  const int cl = threadIdx.x & 3, seg = threadIdx.x >> 2;
  const uint32_t c = blk * 4 + cl;
  double s = 0.0;
  if (c < j.ncols) {
    // 32 rows per batch in flight (was 8: at 1280 partial rows a thread's 20 loads were three dependent round trips of
    // cold misses -- the rows were written by the launch before); the additions keep their order, so the bits do not change
    uint32_t b = seg;
    for (; b + 31u * 64u < j.nb; b += 32u * 64u) {
      double v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = j.part[(size_t)(b + 64u * i) * j.ncols + c];
#pragma unroll
      for (int i = 0; i < 32; ++i) s += v[i];
    }
    {
      double v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = (b + 64u * i < j.nb) ? j.part[(size_t)(b + 64u * i) * j.ncols + c] : 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i)
        if (b + 64u * i < j.nb) s += v[i];
    }
  }
  lds[seg][cl] = s;
  __syncthreads();
  // fixed-order tree over the 64 row segments
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    if (seg < o) lds[seg][cl] += lds[seg + o][cl];
    __syncthreads();
  }
  if (seg == 0 && c < j.ncols && !stopped) j.out[c] = lds[0][cl];
}

// ======================================= node finalise (A7 + swap + A5 + A9)
// compute_mean_indicators (src/linksampling.cc:526-545), the gamma swap/reset
// (:751-755), set_dir_exp (src/linksampling.hh:170-187) and prune (:455-491),
// one group per owned node.
// LIGHT (node-block sweeps, svils_sweep_sharded): only what does not need the all-reduced `sum` -- the mean indicators, their
// s1 / s2 partials, the community tags and the UNSCALED new row, written to this rank's slice of the exchange staging
// (d.gown) instead of gamma; the annealing scale, Elogpi and prune() follow for every row, owned or not, in k_expand_all
// once the rows and `sum` have crossed the ranks (one exchange point instead of two).
template <int W, int V, bool STOCH, bool LIGHT = false>
// V = 16: two waves per SIMD asked for (256 VGPRs with spills instead of 277 + one wave): ca-AstroPh K=1024 354 -> 273 us,
// n=1e5 K=1024 1.22 -> 1.04 ms
#ifndef FIN_OCC8   // waves per SIMD asked of the compiler at V = 8 (K = 257..512)
#define FIN_OCC8 1
#endif
__global__ __launch_bounds__(256, (V == 16 || V == 12 ? 2 : V == 8 ? FIN_OCC8 : 1)) void k_finalize(Geometry geo, DeviceState d, Params prm) {
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  constexpr int G = 64 / W;
  __shared__ double lds[2 * V * 64];
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  // W == 64: one node per wavefront, its index is wave-uniform (said so: scalar loop control, see k_phi)
  const int lane = threadIdx.x & 63, wave = (W == 64) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool annealing = !LIGHT && ctrl->annealing != 0;
  const bool write_comm = ctrl->write_comm != 0;
  const uint32_t *__restrict__ conv_old = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  uint32_t *__restrict__ conv_new = d.conv + (size_t)(ctrl->parity ^ 1u) * geo.n_alloc;

  int kidx[V];
  bool kval[V];
  double scale[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    kidx[v] = kmap<W, V>(lw, v);
    kval[v] = (uint32_t)kidx[v] < K;
    // _network.ones() / _sum[k], src/linksampling.cc:542
    // (mini-batch step: kvec_a is the window's sum, scaled to an estimate of the full one)
    scale[v] = (annealing && kval[v])
                   ? (double)prm.ones / (STOCH ? d.kvec_a[kidx[v]] * prm.scale_a : d.kvec_a[kidx[v]])
                   : 1.0;
  }
  double s12[2][V];
#pragma unroll
  for (int v = 0; v < V; ++v) { s12[0][v] = 0.0; s12[1][v] = 0.0; }
  // 1 / scale of this sweep, for whoever derives the mean indicators from the gamma rows written below
  if (!LIGHT && blockIdx.x == 0 && threadIdx.x < W) {
#pragma unroll
    for (int v = 0; v < V; ++v)
      if (kval[v]) d.iscale[kidx[v]] = annealing ? (STOCH ? d.kvec_a[kidx[v]] * prm.scale_a : d.kvec_a[kidx[v]]) / (double)prm.ones : 1.0;
  }

  const uint32_t nown = geo.node_end - geo.node_begin;
  // (prefetching the next node's index words and accumulator row was measured at no gain -- ca-AstroPh K=200 40.5 -> 41.4 us,
  //  n=1e6 K=512 3.78 -> 4.12 ms with the occupancy it costs at V = 8: the launch is bound by the rows it writes)
  for (uint32_t i = (blockIdx.x * 4 + wave) * G + g; i < nown; i += gridDim.x * 4 * G) {
    const uint32_t p = geo.node_begin + i;
    const double tl = 2.0 * (double)(d.rowptr[p + 1] - d.rowptr[p]);  // quirk Q3
    double acc[V];
    const int32_t sf = d.split_first[p];
    if (sf < 0) {
      load_row<W, V>(d.gacc + (size_t)p * ld, lw, ld, acc);
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = 0.0;
      const uint32_t sc = d.split_cnt[p];
      for (uint32_t t = 0; t < sc; ++t) {
        double part[V];
        load_row<W, V>(d.parts + (size_t)(sf + t) * ld, lw, ld, part);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += part[v];
      }
      if (write_comm) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          uint32_t c = 0;
          if (kval[v])
            for (uint32_t t = 0; t < sc; ++t) c += d.part_cnt[(size_t)(sf + t) * ld + kidx[v]];
          const unsigned long long b = __ballot(kval[v] && c > prm.lt_min_deg);
          if (lw == 0) d.member[(size_t)p * geo.kw + v] = (b >> (g * W)) & (W == 64 ? ~0ull : ((1ull << (W & 63)) - 1ull));
        }
      }
    }
    double gn[V];
    if (tl > 0.0) {
      double m[V];
      const double rtl = 1.0 / tl;   // ONE division per node (an IEEE division per column was 8 % of this launch's instructions at V = 8)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const double g0 = prm.alpha + acc[v];
        m[v] = (g0 - prm.alpha) * rtl;
        gn[v] = g0 + ((double)geo.n - tl - 1.0) * m[v];
        if (annealing) gn[v] *= scale[v];
        if (kval[v]) { s12[0][v] += m[v]; s12[1][v] += m[v] * m[v]; }
        else { m[v] = 0.0; gn[v] = 0.0; }
      }
      if constexpr (STOCH) {
        // Robbins-Monro step of this node: gamma <- (1 - rho) gamma + rho gamma_hat with
        // rho = (tau0 + c)^-kappa, c = updates the node has had; s1/s2 are kept as running sums
        // over the stored mphi rows, so this row contributes (new - old)
        const uint32_t c = d.ncnt[p];
        const double rho = exp_neg(-prm.kappa * log_tab(prm.tau0 + (double)c, logtab));
        double gold[V], mold[V];
        load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gold);
        load_row<W, V>(d.mphi + (size_t)p * ld, lw, ld, mold);
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (kval[v]) {
            gn[v] = (1.0 - rho) * gold[v] + rho * gn[v];
            s12[0][v] -= mold[v];
            s12[1][v] -= mold[v] * mold[v];
          }
        if (lw == 0) d.ncnt[p] = c + 1u;
      }
      if (STOCH || LIGHT || !d.derive_m) store_row<W, V>(d.mphi + (size_t)p * ld, lw, ld, m);
    } else {
      // no training link: gammanext stays alpha, mphi row stays stale (:532-533)
#pragma unroll
      for (int v = 0; v < V; ++v) gn[v] = kval[v] ? prm.alpha : 0.0;
    }
    if constexpr (LIGHT) {
      store_row<W, V>(d.gown + (size_t)i * ld, lw, ld, gn);
      continue;
    }
    store_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
    double rs = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) rs += gn[v];
    rs = group_sum<W>(rs);
    double el[V];
    if (V == 1 && K < (uint32_t)W) {
      // lane K of the group is idle: let it evaluate psi(row sum) in the same digamma call
      const double arg = ((uint32_t)lw == K) ? rs : (kval[0] ? gn[0] : 1.0);
      const double ps = digamma(arg, logtab);
      const double psi_rs = __shfl(ps, g * W + (int)K, 64);
      el[0] = kval[0] ? ps - psi_rs : 0.0;
    } else {
      const double psi_rs = digamma(rs, logtab);
#pragma unroll
      for (int v = 0; v < V; ++v) el[v] = kval[v] ? digamma(gn[v], logtab) - psi_rs : 0.0;
    }
    if (!d.skip_elogpi) store_row<W, V>(d.elogpi + (size_t)p * ld, lw, ld, el);
    store_epi<W, V>(d, p, lw, ld, K, el);
    // prune / check_and_set_converged, src/linksampling.cc:455-475
    uint32_t active = 0;
    int last_k = -1;
    unsigned long long bits[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const bool a = kval[v] && (gn[v] - prm.alpha >= 1.0);
      const unsigned long long b =
          (__ballot(a) >> (g * W)) & (W == 64 ? ~0ull : ((1ull << (W & 63)) - 1ull));
      bits[v] = b;
      active += (uint32_t)__popcll(b);
      if (a) last_k = max(last_k, kidx[v]);
    }
#pragma unroll
    for (int o = 1; o < W; o <<= 1) last_k = max(last_k, __shfl_xor(last_k, o, 64));
    if (lw == 0) {
      const uint32_t cnew = (active == 1) ? (uint32_t)last_k + 1u : conv_old[p];
      conv_new[p] = cnew;
      d.active_cnt[p] = active;
      uint32_t *xf = d.xflags + (size_t)p * d.xf_ld;   // the same flags, packed for the node-block exchange
      xf[0] = cnew;
      xf[1] = active;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const unsigned long long am = (active <= geo.k10) ? bits[v] : 0ull;
        d.amask[(size_t)p * geo.kw + v] = am;
        xf[2 + 2 * v] = (uint32_t)am;
        xf[3 + 2 * v] = (uint32_t)(am >> 32);
      }
    }
  }
  block_reduce_store<W, V, 2>(s12, d.part_b + (size_t)blockIdx.x * 2 * K, K, lds);
}

// Elogpi from gamma only (LinkSampling ctor / top of infer(), src/linksampling.cc:123,561)
template <int W, int V>
__global__ __launch_bounds__(256) void k_dir_exp(Geometry geo, DeviceState d) {
  constexpr int G = 64 / W;
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  for (uint32_t p = (blockIdx.x * 4 + wave) * G + g; p < geo.n; p += gridDim.x * 4 * G) {
    double gn[V];
    load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
    double rs = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if ((uint32_t)kmap<W, V>(lw, v) >= K) gn[v] = 0.0;
      rs += gn[v];
    }
    rs = group_sum<W>(rs);
    const double psi_rs = digamma(rs, logtab);
    double el[V];
#pragma unroll
    for (int v = 0; v < V; ++v) el[v] = (uint32_t)kmap<W, V>(lw, v) < K ? digamma(gn[v], logtab) - psi_rs : 0.0;
    store_row<W, V>(d.elogpi + (size_t)p * ld, lw, ld, el);
    store_epi<W, V>(d, p, lw, ld, K, el);
  }
}

// Multi-GPU: the converged / active flags of a row another rank owns, from its packed xflags row
// (gathered after phase B) into the arrays the kernels read
template <int V>
__device__ __forceinline__ void unpack_flags(const Geometry &geo, const DeviceState &d, const DevCtrl *ctrl, uint32_t p) {
  const uint32_t *xf = d.xflags + (size_t)p * d.xf_ld;
  d.conv[(size_t)(ctrl->parity ^ 1u) * geo.n_alloc + p] = xf[0];
  d.active_cnt[p] = xf[1];
  const uint8_t cf = cflag_pack(xf[0], xf[1] < geo.k10);
  if (d.cflag[p] != cf) { d.cflag[p] = cf; d.cls_epoch[0] = ctrl->sweeps_done + 1u; }   // see k_finalize_lpl
#pragma unroll
  for (int v = 0; v < V; ++v)
    d.amask[(size_t)p * geo.kw + v] = (unsigned long long)xf[2 + 2 * v] | ((unsigned long long)xf[3 + 2 * v] << 32);
}

// Multi-GPU: Elogpi and mphi of the rows this handle does not own, re-derived from the
// all-gathered gamma instead of being exchanged.  gamma_new = (alpha + acc (n-1)/tl) * scale with
// m = acc/tl (compute_mean_indicators, src/linksampling.cc:536-542)  =>  m = (gamma/scale - alpha)/(n-1).
// Rows without a training link (gamma == alpha, unscaled) never enter the s3 pass; their mphi is written as 0.
// [xb, xe): the rows, relative to the start of every OTHER rank's node block of `xblock` rows, that this launch expands
// (the whole block, or one chunk of a pipelined exchange: svils_sweep_sharded expands chunk c while chunk c + 1 travels).
template <int W, int V>
__global__ __launch_bounds__(256) void k_expand(Geometry geo, DeviceState d, Params prm, uint32_t xb, uint32_t xe, uint32_t xblock) {
  const DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  constexpr int G = 64 / W;
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool annealing = ctrl->annealing != 0;
  bool kval[V];
  double iscale[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int k = kmap<W, V>(lw, v);
    kval[v] = (uint32_t)k < K;
    iscale[v] = (annealing && kval[v]) ? d.kvec_a[k] / (double)prm.ones : 1.0;   // 1 / (ones / sum[k])
  }
  const double inv_nm1 = 1.0 / ((double)geo.n - 1.0);
  // xblock == 0: every row this handle does not own
  const uint32_t nown = geo.node_end - geo.node_begin;
  const uint32_t xlen = xe - xb, nblocks = xblock ? (geo.n_alloc + xblock - 1) / xblock : 0u, myblock = xblock ? geo.node_begin / xblock : 0u;
  const uint32_t total = xblock ? xlen * (nblocks - 1u) : geo.n - nown;
  for (uint32_t i = (blockIdx.x * 4 + wave) * G + g; i < total; i += gridDim.x * 4 * G) {
    uint32_t p;
    if (xblock) {
      uint32_t r = i / xlen;
      if (r >= myblock) ++r;
      p = r * xblock + xb + i % xlen;
      if (p >= geo.n || p >= (r + 1u) * xblock) continue;
    } else {
      p = i < geo.node_begin ? i : i + nown;
    }
    double gn[V];
    load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
    double rs = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) { if (!kval[v]) gn[v] = 0.0; rs += gn[v]; }
    rs = group_sum<W>(rs);
    const double psi_rs = digamma(rs, logtab);
    double el[V], m[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      el[v] = kval[v] ? digamma(gn[v], logtab) - psi_rs : 0.0;
      m[v] = kval[v] ? (gn[v] * iscale[v] - prm.alpha) * inv_nm1 : 0.0;
    }
    if (d.rowptr[p + 1] == d.rowptr[p]) {
#pragma unroll
      for (int v = 0; v < V; ++v) m[v] = 0.0;
    }
    if (!d.skip_elogpi) store_row<W, V>(d.elogpi + (size_t)p * ld, lw, ld, el);
    store_epi<W, V>(d, p, lw, ld, K, el);
    store_row<W, V>(d.mphi + (size_t)p * ld, lw, ld, m);
    if (lw == 0) unpack_flags<V>(geo, d, ctrl, p);
  }
}

// Node-block sweeps, second half of the node finalise for EVERY row (owned or not), after the exchange: the staged row
// is the unscaled gammanext of compute_mean_indicators (src/linksampling.cc:536-540); here the annealing scale
// ones / sum[k] (:541-542, `sum` all-reduced by now), the swap into gamma (:751), set_dir_exp (src/linksampling.hh:170-187)
// and prune (:455-491).  Every rank computes every row's flags from the same bytes with the same instructions, so the
// flags are replicated without being exchanged.  The mean indicators of rows another rank owns are re-derived from the
// staged row, m = (row - alpha) / (n - 1) (gammanext = alpha + acc + (n - tl - 1) acc / tl = alpha + acc (n - 1) / tl);
// the owner stored its own in the light finalise pass.
template <int W, int V>
__global__ __launch_bounds__(256) void k_expand_all(Geometry geo, DeviceState d, Params prm, Blocks blk) {
  const DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  constexpr int G = 64 / W;
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool annealing = ctrl->annealing != 0;
  const uint32_t *__restrict__ conv_old = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  uint32_t *__restrict__ conv_new = d.conv + (size_t)(ctrl->parity ^ 1u) * geo.n_alloc;
  int kidx[V];
  bool kval[V];
  double scale[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    kidx[v] = kmap<W, V>(lw, v);
    kval[v] = (uint32_t)kidx[v] < K;
    scale[v] = (annealing && kval[v]) ? (double)prm.ones / d.kvec_a[kidx[v]] : 1.0;   // _network.ones() / _sum[k]
  }
  if (blockIdx.x == 0 && threadIdx.x < W && blk.chunk == 0) {
#pragma unroll
    for (int v = 0; v < V; ++v)
      if (kval[v]) d.iscale[kidx[v]] = annealing ? d.kvec_a[kidx[v]] / (double)prm.ones : 1.0;
  }
  const double inv_nm1 = 1.0 / ((double)geo.n - 1.0);
  // rows of a slice this launch may touch: the whole slice, or one chunk of it (+1: chunk boundaries are rounded per block)
  const uint32_t clen = blk.nchunks > 1 ? (blk.bmax + blk.nchunks - 1) / blk.nchunks + 1u : blk.bmax;
  const uint64_t total = (uint64_t)blk.world * clen;
  for (uint64_t i = (uint64_t)(blockIdx.x * 4 + wave) * G + g; i < total; i += (uint64_t)gridDim.x * 4 * G) {
    const uint32_t r = (uint32_t)(i / clen), jj = (uint32_t)(i % clen);
    uint32_t lo, hi;
    chunk_range(blk.bounds[r + 1] - blk.bounds[r], blk.chunk, blk.nchunks, &lo, &hi);
    const uint32_t j = lo + jj;
    if (j >= hi) continue;
    const uint32_t p = blk.bounds[r] + j;
    const bool own = p >= geo.node_begin && p < geo.node_end;
    const bool has = d.rowptr[p + 1] != d.rowptr[p];
    double gn[V];
    load_row<W, V>(d.gstage + ((size_t)r * blk.bmax + j) * ld, lw, ld, gn);
    double m[V];
    double rs = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      m[v] = (kval[v] && has) ? (gn[v] - prm.alpha) * inv_nm1 : 0.0;
      if (has) gn[v] *= scale[v];       // rows without a training link stay alpha, unscaled (:532-533)
      if (!kval[v]) gn[v] = 0.0;
      rs += gn[v];
    }
    store_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
    if (!own) store_row<W, V>(d.mphi + (size_t)p * ld, lw, ld, m);
    rs = group_sum<W>(rs);
    const double psi_rs = digamma(rs, logtab);
    double el[V];
#pragma unroll
    for (int v = 0; v < V; ++v) el[v] = kval[v] ? digamma(gn[v], logtab) - psi_rs : 0.0;
    if (!d.skip_elogpi) store_row<W, V>(d.elogpi + (size_t)p * ld, lw, ld, el);
    store_epi<W, V>(d, p, lw, ld, K, el);
    // prune / check_and_set_converged, src/linksampling.cc:455-475
    uint32_t active = 0;
    int last_k = -1;
    unsigned long long bits[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const bool a = kval[v] && (gn[v] - prm.alpha >= 1.0);
      const unsigned long long b = (__ballot(a) >> (g * W)) & (W == 64 ? ~0ull : ((1ull << (W & 63)) - 1ull));
      bits[v] = b;
      active += (uint32_t)__popcll(b);
      if (a) last_k = max(last_k, kidx[v]);
    }
#pragma unroll
    for (int o = 1; o < W; o <<= 1) last_k = max(last_k, __shfl_xor(last_k, o, 64));
    if (lw == 0) {
      const uint32_t cnew = (active == 1) ? (uint32_t)last_k + 1u : conv_old[p];
      conv_new[p] = cnew;
      d.active_cnt[p] = active;
      const uint8_t cf = cflag_pack(cnew, active < geo.k10);
      if (d.cflag[p] != cf) { d.cflag[p] = cf; d.cls_epoch[0] = ctrl->sweeps_done + 1u; }   // see k_finalize_lpl
#pragma unroll
      for (int v = 0; v < V; ++v) d.amask[(size_t)p * geo.kw + v] = (active <= geo.k10) ? bits[v] : 0ull;
    }
  }
}

// cflag[] from conv[parity] and active_cnt[] as they stand (after the host replaced the flags)
__global__ __launch_bounds__(256) void k_cflag_rebuild(Geometry geo, DeviceState d) {
  const uint32_t *__restrict__ conv = d.conv + (size_t)d.ctrl->parity * geo.n_alloc;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < geo.n; p += gridDim.x * blockDim.x)
    d.cflag[p] = cflag_pack(conv[p], d.active_cnt[p] < geo.k10);
}
void launch_cflag_rebuild(const Geometry &g, const DeviceState &d, hipStream_t s) {
  const uint32_t nb = std::min<uint32_t>((g.n + 255u) / 256u, 1024u);
  hipLaunchKernelGGL(k_cflag_rebuild, dim3(nb ? nb : 1), dim3(256), 0, s, g, d);
}

// The stored form of the mean indicators, brought up to date from the gamma rows of the last whole sweep (derive_m):
// every row with a training link; the others keep their stale row (src/linksampling.cc:532-533).
__global__ __launch_bounds__(256) void k_mphi_from_gamma(Geometry geo, DeviceState d, Params prm) {
  const uint32_t K = geo.K, ld = geo.ld;
  const double inv_nm1 = 1.0 / ((double)geo.n - 1.0);
  const uint64_t total = (uint64_t)geo.n * ld;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t p = (uint32_t)(i / ld), k = (uint32_t)(i % ld);
    if (k < K && d.rowptr[p + 1] != d.rowptr[p]) d.mphi[i] = (d.gamma[i] * d.iscale[k] - prm.alpha) * inv_nm1;
  }
}
void launch_mphi_from_gamma(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  const uint64_t total = (uint64_t)g.n * g.ld;
  const uint32_t nb = (uint32_t)std::min<uint64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_mphi_from_gamma, dim3(nb ? nb : 1), dim3(256), 0, s, g, d, p);
}

// ================================================================ s3 pass (A8)
// src/linksampling.cc:731-746 over the upper half (q > p) of each owned row,
// including quirk Q2 (mphi[q][pc], one past the converged community).
#ifndef S3_PIPE_MAXV   // largest V whose k_s3 keeps the next neighbour's row in flight (V more doubles per lane)
#define S3_PIPE_MAXV 4
#endif
template <int W, int V, bool DERIVE>
__global__ __launch_bounds__(256) void k_s3(Geometry geo, DeviceState d, Params prm) {
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  constexpr int G = 64 / W;
  __shared__ double lds[V * 64];
  // wave-uniform values said to be so (see k_phi): scalar loop control instead of exec masks
  constexpr bool UNI = (W == 64);
  auto uni = [](int x) { return UNI ? __builtin_amdgcn_readfirstlane(x) : x; };
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  // derive_m: the mean indicators come from the gamma rows, m = (gamma * iscale - alpha) / (n - 1)
  constexpr bool derive = DERIVE;
  const double *__restrict__ mphi = derive ? d.gamma : d.mphi;
  const double alpha = prm.alpha, inv_nm1 = 1.0 / ((double)geo.n - 1.0);
  int kidx[V];
  double isc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    kidx[v] = kmap<W, V>(lw, v);
    isc[v] = (derive && (uint32_t)kidx[v] < K) ? d.iscale[kidx[v]] : 0.0;
  }
  double s3[1][V];
#pragma unroll
  for (int v = 0; v < V; ++v) s3[0][v] = 0.0;

  for (uint32_t it = blockIdx.x * 4 + wave; it < d.nitems_s3; it += gridDim.x * 4) {
    const Item item_ = d.items_s3[d.item0_s3 + it];
    Item item;
    item.node = (uint32_t)uni((int)item_.node);
    item.off = (uint32_t)uni((int)item_.off);
    item.len = (uint32_t)uni((int)item_.len);
    item.slot = item_.slot;
    const uint32_t p = item.node;
    const uint64_t base = d.rowptr[p] + item.off;
    const uint32_t pc = (uint32_t)uni((int)conv[p]);
    double mp[V];
    load_row<W, V>(mphi + (size_t)p * ld, lw, ld, mp);
    if (derive) {
#pragma unroll
      for (int v = 0; v < V; ++v) mp[v] = (uint32_t)kidx[v] < K ? (mp[v] * isc[v] - alpha) * inv_nm1 : 0.0;
    }
    if constexpr (W == 64 && V <= S3_PIPE_MAXV) {
      // One neighbour at a time was a chain of three dependent misses (column id -> converged flag -> row).  An item has
      // at most 32 neighbours: their ids and flags arrive with ONE coalesced load each (lane j <-> neighbour j), and the
      // row of neighbour j + 1 travels while neighbour j is multiplied (as k_phi does it).
      uint32_t qv = 0, qcv = 0;
      if ((uint32_t)lane < item.len) { qv = d.col[base + lane]; qcv = conv[qv]; }
      auto wants_row = [&](uint32_t qc) { return (pc != 0) == (qc != 0); };   // wave-uniform: pc, qc are
      double mqn[V];
#pragma unroll
      for (int v = 0; v < V; ++v) mqn[v] = 0.0;
      if (item.len > 0) {
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)qv, 0), qc0 = (uint32_t)__builtin_amdgcn_readlane((int)qcv, 0);
        if (wants_row(qc0)) load_row<W, V>(mphi + (size_t)q0 * ld, lw, ld, mqn);
      }
      for (uint32_t j = 0; j < item.len; ++j) {
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)qv, (int)j);
        const uint32_t qc = (uint32_t)__builtin_amdgcn_readlane((int)qcv, (int)j);
        double mq[V];
#pragma unroll
        for (int v = 0; v < V; ++v) mq[v] = mqn[v];
        if (j + 1 < item.len) {
          const uint32_t qn = (uint32_t)__builtin_amdgcn_readlane((int)qv, (int)j + 1);
          const uint32_t qcn = (uint32_t)__builtin_amdgcn_readlane((int)qcv, (int)j + 1);
          if (wants_row(qcn)) load_row<W, V>(mphi + (size_t)qn * ld, lw, ld, mqn);
        }
        if (pc && !qc) {
          double val = pc < K ? mphi[(size_t)q * ld + pc] : 0.0;
          if (derive && pc < K) val = (val * d.iscale[pc] - alpha) * inv_nm1;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kidx[v] == (int)pc - 1) s3[0][v] += val;
        } else if (!pc && qc) {
          double val = qc < K ? mphi[(size_t)p * ld + qc] : 0.0;
          if (derive && qc < K) val = (val * d.iscale[qc] - alpha) * inv_nm1;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kidx[v] == (int)qc - 1) s3[0][v] += val;
        } else {
          if (derive) {
#pragma unroll
            for (int v = 0; v < V; ++v) mq[v] = (mq[v] * isc[v] - alpha) * inv_nm1;   // padding columns: mp is 0 there
          }
#pragma unroll
          for (int v = 0; v < V; ++v) s3[0][v] += mp[v] * mq[v];
        }
      }
    } else {
      for (uint32_t j = g; j < item.len; j += G) {
        const uint32_t q = d.col[base + j];
        const uint32_t qc = conv[q];
        if (pc && !qc) {
          double val = pc < K ? mphi[(size_t)q * ld + pc] : 0.0;
          if (derive && pc < K) val = (val * d.iscale[pc] - alpha) * inv_nm1;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kidx[v] == (int)pc - 1) s3[0][v] += val;
        } else if (!pc && qc) {
          double val = qc < K ? mphi[(size_t)p * ld + qc] : 0.0;
          if (derive && qc < K) val = (val * d.iscale[qc] - alpha) * inv_nm1;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kidx[v] == (int)qc - 1) s3[0][v] += val;
        } else {
          double mq[V];
          load_row<W, V>(mphi + (size_t)q * ld, lw, ld, mq);
          if (derive) {
#pragma unroll
            for (int v = 0; v < V; ++v) mq[v] = (mq[v] * isc[v] - alpha) * inv_nm1;   // padding columns: mp is 0 there
          }
#pragma unroll
          for (int v = 0; v < V; ++v) s3[0][v] += mp[v] * mq[v];
        }
      }
    }
  }
  block_reduce_store<W, V, 1>(s3, d.part_c + (size_t)blockIdx.x * K, K, lds);
}

// lambda of the sweep being finished, from the reduced K-vectors (src/linksampling.cc:748-754).
// Full sweep: lambda[k] = (eta0 + sum[k], eta1 + s1^2 - s2 - s3).  Mini-batch step: sum and s3 are
// window sums scaled to estimates of the full ones, s1/s2 are running totals (previous total + this
// window's change), and the result is blended into the old lambda with the step size rho_lambda.
// Returns the running s1, s2 through s1r/s2r (what k_tail stores back in mini-batch mode).
template <bool STOCH>
__device__ __forceinline__ void lambda_from(const DeviceState &d, const Params &prm, uint32_t K, uint32_t k, double sum,
                                            double s1, double s2, double s3, double &l0, double &l1, double &s1r, double &s2r) {
  if constexpr (!STOCH) {
    l0 = prm.eta0 + sum;
    l1 = prm.eta1 + (s1 * s1 - s2 - s3);
  } else {
    s1 += d.s12run[k];
    s2 += d.s12run[K + k];
    s3 *= prm.scale_c;
    const double h0 = prm.eta0 + sum * prm.scale_a;
    const double h1 = prm.eta1 + (s1 * s1 - s2 - s3);
    l0 = (1.0 - prm.rho_lambda) * d.lambda[2 * k] + prm.rho_lambda * h0;
    l1 = (1.0 - prm.rho_lambda) * d.lambda[2 * k + 1] + prm.rho_lambda * h1;
  }
  s1r = s1;
  s2r = s2;
}
template <bool STOCH>
__device__ __forceinline__ void lambda_of_sweep(const DeviceState &d, const Params &prm, uint32_t K, uint32_t k,
                                                const double *s3v, double &l0, double &l1, double &s1r, double &s2r) {
  lambda_from<STOCH>(d, prm, K, k, d.kvec_a[k], d.kvec_c[k], d.kvec_c[K + k], s3v[k], l0, l1, s1r, s2r);
}

// ================================================== validation likelihood (A10)
// edge_likelihood (src/linksampling.hh:258-292) of one held-out pair, by a group of W lanes.
// The non-link K^2 double loop collapses exactly:
//   sum_{z,z'} pi_p[z] pi_q[z'] (1 - [z==z'] beta_z - [z!=z'] eps), 1 - eps == 1.0 in double
//   = (sum pi_p)(sum pi_q) - sum_z pi_p[z] pi_q[z] beta_z .
template <int W, int V>
__device__ __forceinline__ double pair_loglik(const DeviceState &d, uint32_t ld, int lw, uint32_t i,
                                              const double (&beta)[V], uint32_t *y_out) {
  const uint32_t p = d.vpairs[3 * (size_t)i], q = d.vpairs[3 * (size_t)i + 1];
  const uint32_t y = d.vpairs[3 * (size_t)i + 2];
  double gp[V], gq[V];
  load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gp);
  load_row<W, V>(d.gamma + (size_t)q * ld, lw, ld, gq);
  double sp = 0.0, sq = 0.0, dot = 0.0;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    sp += gp[v];
    sq += gq[v];
    dot += gp[v] * gq[v] * beta[v];
  }
  sp = group_sum<W>(sp);
  sq = group_sum<W>(sq);
  dot = group_sum<W>(dot);
  const double pq = dot / (sp * sq);
  double s = y ? pq : 1.0 - pq;
  if (s < 1e-30) s = 1e-30;
  *y_out = y;
  return log(s);
}

// the constructor's call (src/linksampling.cc:149-150): lambda as it stands, results to uval[]
template <int W, int V>
__global__ __launch_bounds__(256) void k_validation(Geometry geo, DeviceState d) {
  constexpr int G = 64 / W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  double beta[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int k = kmap<W, V>(lw, v);
    beta[v] = 0.0;
    if ((uint32_t)k < K) {
      const double l0 = d.lambda[2 * k], l1 = d.lambda[2 * k + 1];
      beta[v] = l0 / (l0 + l1);  // estimate_bernoulli_rate, src/linksampling.hh:216-225
    }
  }
  for (uint32_t i = (blockIdx.x * 4 + wave) * G + g; i < d.nv; i += gridDim.x * 4 * G) {
    uint32_t y;
    const double u = pair_loglik<W, V>(d, ld, lw, i, beta, &y);
    if (lw == 0) d.uval[i] = u;
  }
}

// Sharded mini-batch step: Elogpi of the OTHER ranks' window rows, from the gamma rows just gathered
// (their mphi rows are gathered as they are: a blended gamma no longer determines them).
template <int W, int V>
__global__ __launch_bounds__(256) void k_expand_window(Geometry geo, DeviceState d, uint32_t wb, uint32_t we,
                                                       uint32_t block, uint32_t my_rank, uint32_t world) {
  const DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  constexpr int G = 64 / W;
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  const uint32_t wlen = we - wb, total = wlen * (world - 1);
  for (uint32_t i = (blockIdx.x * 4 + wave) * G + g; i < total; i += gridDim.x * 4 * G) {
    uint32_t r = i / wlen;
    if (r >= my_rank) ++r;
    const uint32_t p = r * block + wb + i % wlen;
    if (p >= geo.n || p >= (r + 1) * block) continue;
    double gn[V];
    load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
    double rs = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) { if ((uint32_t)kmap<W, V>(lw, v) >= K) gn[v] = 0.0; rs += gn[v]; }
    rs = group_sum<W>(rs);
    const double psi_rs = digamma(rs, logtab);
    double el[V];
#pragma unroll
    for (int v = 0; v < V; ++v) el[v] = (uint32_t)kmap<W, V>(lw, v) < K ? digamma(gn[v], logtab) - psi_rs : 0.0;
    if (!d.skip_elogpi) store_row<W, V>(d.elogpi + (size_t)p * ld, lw, ld, el);
    store_epi<W, V>(d, p, lw, ld, K, el);
    if (lw == 0) unpack_flags<V>(geo, d, ctrl, p);
  }
}

// Mini-batch step over a node window: prune() wrote conv[parity^1] for the window's rows only;
// every other row carries its flag over so that the parity flip in k_tail keeps it.
__global__ __launch_bounds__(256) void k_carry_flags(Geometry geo, DeviceState d) {
  const DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  const uint32_t *__restrict__ conv_old = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  uint32_t *__restrict__ conv_new = d.conv + (size_t)(ctrl->parity ^ 1u) * geo.n_alloc;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < geo.n; p += gridDim.x * blockDim.x)
    if (p < geo.node_begin || p >= geo.node_end) conv_new[p] = conv_old[p];
}

// =============================================================== tail (A8/A10/A11)
// One launch for everything after the s3 pass: lambda of this sweep and the held-out likelihood
// of validation_likelihood() (src/linksampling.cc:966-1002) by every block over its share of the
// pairs; the LAST block to arrive (ticket + agent-scope atomics, no L2 write-back) then adds the
// blocks' partial sums in block order and runs the serial part: lambda update + set_dir_exp(lambda)
// (:748-759), the likelihood row, stop rule and annealing switch (:994-1049), write_comm for the
// next sweep (:768-774) and _iter++ (:787).
// The control block as the tail of a sweep leaves it: the fields a tail changes (bytes 0 .. 79), one by one.  A whole-struct
// `*d.ctrl = c` keeps the 16 bytes a tail never touches in a private 16-byte object per lane; the compiler then either spills
// that to scratch, or -- when the block's LDS is small enough -- promotes it to LDS and indexes it by the work-group
// size, which it READS FROM THE DISPATCH PACKET: host memory, once per wavefront.  246 blocks of four waves doing that in
// a row made the launch 20 us longer (15.5 -> 36 us, profiles/r07v_ab_tail_dispatch_packet_read.txt).
__device__ __forceinline__ void store_ctrl_of_tail(DevCtrl *dst, const DevCtrl &c) {
  dst->iter = c.iter; dst->annealing = c.annealing; dst->write_comm = c.write_comm; dst->nh = c.nh;
  dst->prev_h = c.prev_h; dst->max_h = c.max_h;
  dst->stopped = c.stopped; dst->why = c.why; dst->sweeps_done = c.sweeps_done; dst->rows = c.rows;
  dst->links_dense = c.links_dense; dst->links_sparse = c.links_sparse; dst->links_shortcut = c.links_shortcut;
  dst->parity = c.parity; dst->cls_par = c.cls_par;
}

// Sums of N per-thread values over a 256-thread block, every thread ends with the totals: DPP inside the wavefronts,
// the four wave totals through LDS, added in wave order -- a fixed order, and 0.3 us where a shared-memory tree of eight
// barrier levels over six arrays took 1.2 (profiles/r07t_tail_timeline_astroph_k200_first.txt).  Ends with no barrier: a
// second call on the same `wl` needs one in between (k_tail has the ticket's).
template <int N>
__device__ __forceinline__ void block_sums(double (&x)[N], double (*wl)[6]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < N; ++j) x[j] = group_sum<64>(x[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) wl[wave][j] = x[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) x[j] = ((wl[0][j] + wl[1][j]) + wl[2][j]) + wl[3][j];
}

#ifndef TAIL_BETAL   // K up to here: the rates of lambda go through LDS, once per block (0: every lane loads and divides its own)
#define TAIL_BETAL 256
#endif
template <int W, int V, bool STOCH>
__global__ __launch_bounds__(256) void k_tail(Geometry geo, DeviceState d, Params prm) {
  if (blockIdx.x >= d.nb_t) {
    // extra workgroups (K <= 32, full sweeps): scatter pass of the NEXT sweep's link classes.  They
    // take everything from cls_args, never from the control block this launch is about to advance.
    __shared__ ClsWork csh[1];
    STAMP(3, 0);
    if (!d.ctrl->stopped) cls_scatter_tiles<1>(geo, d, csh, blockIdx.x - d.nb_t, gridDim.x - d.nb_t);
    STAMP(3, 7);
    return;
  }
  STAMP(3, 0);
  constexpr int G = 64 / W;
  constexpr int NPRE = V <= 4 ? 2 : 1;   // held-out pairs whose rows are fetched before lambda is known (a group has ~2 pairs at 2000 held-out links: V = 4, K = 129..256, took its second pair in a dependent round of its own until round 4)
  __shared__ double wsum[4][6];
  __shared__ double2 logtab[128];
  __shared__ double s3l[32];
  __shared__ double ftmp[8 * 32];
#if TAIL_BETAL > 0
  __shared__ double betal[TAIL_BETAL];
#endif
  __shared__ uint32_t lastflag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  // Every dependent global access of this launch costs a cold miss, so the independent chains are
  // started together, the one with a second level first: (1) the held-out pairs, whose gamma rows
  // (level 2) do not need lambda -- their products wait in registers; (2) the log table the last
  // block will need; (3) the s3 pass's partial rows to fold.  What does not depend on the control
  // block -- the pairs' indices, the log table, the reduced K-vectors of the serial part's first
  // round of columns -- is requested BESIDE it, not behind it.
  const uint32_t i0 = (blockIdx.x * 4 + wave) * G + g, istride = d.nb_t * 4 * G;
  // (a load under a per-lane condition is waited for where the branches join: two pairs, each with its rows behind it,
  // were four round trips in a row.  Indices past the end are clamped instead and their results dropped at the sums.)
  uint32_t pp[NPRE], qq[NPRE], yy[NPRE];
#pragma unroll
  for (int t = 0; t < NPRE; ++t) { pp[t] = 0; qq[t] = 0; yy[t] = 0; }
  if (d.nv) {
#pragma unroll
    for (int t = 0; t < NPRE; ++t) {
      const uint32_t i = i0 + t * istride, ic = i < d.nv ? i : d.nv - 1u;
      pp[t] = d.vpairs[3 * (size_t)ic]; qq[t] = d.vpairs[3 * (size_t)ic + 1]; yy[t] = d.vpairs[3 * (size_t)ic + 2];
    }
  }
  double2 ltv = make_double2(0.0, 0.0);
  if (threadIdx.x < 128) ltv = make_double2(d.logtab[2 * threadIdx.x], d.logtab[2 * threadIdx.x + 1]);
  double pre_sum = 0.0, pre_s1 = 0.0, pre_s2 = 0.0, pre_s3 = 0.0;
  if (threadIdx.x < K) {
    pre_sum = d.kvec_a[threadIdx.x]; pre_s1 = d.kvec_c[threadIdx.x]; pre_s2 = d.kvec_c[K + threadIdx.x];
    if (!d.fold) pre_s3 = d.kvec_c[2 * (size_t)K + threadIdx.x];
  }
  DevCtrl c = *d.ctrl;
  if (c.stopped) return;
  STAMP(3, 1);
  const uint32_t iter = c.iter;
  const bool do_val = d.nv > 0 && (iter % prm.reportfreq == 0);
  // A sweep without a likelihood row (nine of ten at the default report frequency) leaves the other blocks nothing to
  // do: block 0 runs the serial part alone -- no partial sums, no ticket (nb_t arrivals at one address in a row).
  if (!do_val && blockIdx.x != 0u) return;
  FoldRows<32, 256> fold;
  if (d.fold) fold.issue(d.part_c, d.nb_c, K);
  double prod[NPRE][V], spq[NPRE];
#pragma unroll
  for (int t = 0; t < NPRE; ++t) {
    spq[t] = 1.0;
#pragma unroll
    for (int v = 0; v < V; ++v) prod[t][v] = 0.0;
  }
  if (do_val) {
    double gp[NPRE][V], gq[NPRE][V];
#pragma unroll
    for (int t = 0; t < NPRE; ++t) {   // every row of every pair in flight before anything is waited for
      load_row<W, V>(d.gamma + (size_t)pp[t] * ld, lw, ld, gp[t]);
      load_row<W, V>(d.gamma + (size_t)qq[t] * ld, lw, ld, gq[t]);
    }
#pragma unroll
    for (int t = 0; t < NPRE; ++t) {
      double sp = 0.0, sq = 0.0;
#pragma unroll
      for (int v = 0; v < V; ++v) { sp += gp[t][v]; sq += gq[t][v]; prod[t][v] = gp[t][v] * gq[t][v]; }
      spq[t] = group_sum<W>(sp) * group_sum<W>(sq);
    }
  }
  if (threadIdx.x < 128) logtab[threadIdx.x] = ltv;
  // s3 of this sweep: folded from the s3 pass's per-block partial rows, or the reduced vector
  const double *s3v = d.kvec_c + 2 * (size_t)K;
  if (d.fold) {
    fold.finish(ftmp, s3l);
    s3v = s3l;
  }
  STAMP(3, 2);
  double sz = 0.0, so = 0.0;
  unsigned long long kz = 0;
  if (do_val) {
    double beta[V];
#if TAIL_BETAL > 0
    // the rates of this sweep's lambda, once per block into LDS from the K-vectors requested at the top
    if (K <= (uint32_t)TAIL_BETAL) {
      for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) {
        double l0, l1, s1r, s2r;
        if (k == threadIdx.x) lambda_from<STOCH>(d, prm, K, k, pre_sum, pre_s1, pre_s2, d.fold ? s3v[k] : pre_s3, l0, l1, s1r, s2r);
        else lambda_of_sweep<STOCH>(d, prm, K, k, s3v, l0, l1, s1r, s2r);
        betal[k] = l0 / (l0 + l1);  // estimate_bernoulli_rate, src/linksampling.hh:216-225
      }
      __syncthreads();
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int k = kmap<W, V>(lw, v);
        beta[v] = (uint32_t)k < K ? betal[k] : 0.0;
      }
    } else
#endif
    {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int k = kmap<W, V>(lw, v);
        beta[v] = 0.0;
        if ((uint32_t)k < K) {
          double l0, l1, s1r, s2r;
          lambda_of_sweep<STOCH>(d, prm, K, (uint32_t)k, s3v, l0, l1, s1r, s2r);
          beta[v] = l0 / (l0 + l1);  // estimate_bernoulli_rate, src/linksampling.hh:216-225
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NPRE; ++t) {
      const uint32_t i = i0 + t * istride;
      double dot = 0.0;
#pragma unroll
      for (int v = 0; v < V; ++v) dot += prod[t][v] * beta[v];
      dot = group_sum<W>(dot);
      if (i < d.nv) {
        const double pq = dot / spq[t];
        double sv = yy[t] ? pq : 1.0 - pq;
        if (sv < 1e-30) sv = 1e-30;
        const double u = log(sv);
        if (lw == 0) { if (yy[t]) so += u; else { sz += u; kz++; } }
      }
    }
    // the remaining pairs, NB at a time per group: all their indices, then all their rows in flight
    constexpr int NB = V <= 2 ? 4 : (V <= 8 ? 2 : 1);
    for (uint32_t ib = i0 + NPRE * istride; ib < d.nv; ib += NB * istride) {
      uint32_t bp[NB], bq[NB], by[NB];
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const uint32_t i = ib + t * istride, ic = i < d.nv ? i : d.nv - 1u;   // clamped, not branched: see the top
        bp[t] = d.vpairs[3 * (size_t)ic]; bq[t] = d.vpairs[3 * (size_t)ic + 1]; by[t] = d.vpairs[3 * (size_t)ic + 2];
      }
      double gp[NB][V], gq[NB][V];
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        load_row<W, V>(d.gamma + (size_t)bp[t] * ld, lw, ld, gp[t]);
        load_row<W, V>(d.gamma + (size_t)bq[t] * ld, lw, ld, gq[t]);
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        double sp = 0.0, sq = 0.0, dot = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) { sp += gp[t][v]; sq += gq[t][v]; dot += gp[t][v] * gq[t][v] * beta[v]; }
        sp = group_sum<W>(sp); sq = group_sum<W>(sq); dot = group_sum<W>(dot);
        if (ib + t * istride < d.nv) {
          const double pq = dot / (sp * sq);
          double sv = by[t] ? pq : 1.0 - pq;
          if (sv < 1e-30) sv = 1e-30;
          const double u = log(sv);
          if (lw == 0) { if (by[t]) so += u; else { sz += u; kz++; } }
        }
      }
    }
  }
  STAMP(3, 3);
  double kzd = 0.0;
  if (do_val) {
    // the block's partial sums (fixed order; the count is an integer far below 2^53: exact as a double), published for
    // the last block
    double bp3[3] = {sz, so, (double)kz};
    block_sums<3>(bp3, wsum);
    if (threadIdx.x == 0) {
      st_agent(d.tail_part + (size_t)blockIdx.x * 4, bp3[0]);
      st_agent(d.tail_part + (size_t)blockIdx.x * 4 + 1, bp3[1]);
      st_agent(d.tail_part + (size_t)blockIdx.x * 4 + 2, bp3[2]);
    }
    STAMP(3, 4);
    if (!last_block_arrives(d.tail_ctl, d.nb_t, &lastflag)) return;
    STAMP(3, 5);
    sz = 0.0; so = 0.0;
  } else {
    __syncthreads();   // block 0 alone: the log table is in LDS for every wave
  }
  // ---- the last block (a sweep with a likelihood row) or block 0 (a sweep without) ----
  // the blocks' partial sums: thread t adds blocks t, t + 256, ..., then a fixed-order tree.  The first round is only
  // REQUESTED here (clamped index, dropped below): the link counts that follow travel beside it, not behind it.
  double pz = 0.0, po = 0.0, pk = 0.0;
  if (do_val) {
    const uint32_t bc = threadIdx.x < d.nb_t ? threadIdx.x : 0u;
    pz = ld_agent(d.tail_part + (size_t)bc * 4);
    po = ld_agent(d.tail_part + (size_t)bc * 4 + 1);
    pk = ld_agent(d.tail_part + (size_t)bc * 4 + 2);
  }
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  const uint32_t cpar0 = c.cls_par;
  uint32_t *ltot = d.lpl ? d.ltot + cpar0 * 8u : nullptr;
  if (!d.lpl) {
    // link statistics of the phi pass: eight blocks' counts per thread requested at once (one block per round was a cold
    // miss per round on the serial part's critical path: five in a row at 1 280 phi blocks; integers, so any order)
    for (uint32_t b0 = threadIdx.x; b0 < d.nb_a; b0 += 8u * blockDim.x) {
      unsigned long long w[8][3];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t b = b0 + (uint32_t)j * blockDim.x, bc = b < d.nb_a ? b : b0;
#pragma unroll
        for (int e = 0; e < 3; ++e) w[j][e] = d.part_links[(size_t)bc * 3 + e];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (b0 + (uint32_t)j * blockDim.x < d.nb_a) { t0 += w[j][0]; t1 += w[j][1]; t2 += w[j][2]; }
    }
  }
  if (do_val) {
    if (threadIdx.x < d.nb_t) { sz += pz; so += po; kzd += pk; }
    for (uint32_t b = threadIdx.x + 256u; b < d.nb_t; b += 256u) {   // (one round up to 256 blocks)
      sz += ld_agent(d.tail_part + (size_t)b * 4);
      so += ld_agent(d.tail_part + (size_t)b * 4 + 1);
      kzd += ld_agent(d.tail_part + (size_t)b * 4 + 2);
    }
  }
  // (link counts: integers below 2^53, exact as doubles)
  double tot[6] = {sz, so, kzd, (double)t0, (double)t1, (double)t2};
  block_sums<6>(tot, wsum);
  // lambda update + set_dir_exp(lambda), src/linksampling.cc:748-759
  for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) {
    double l0, l1, s1r, s2r;
    if (k == threadIdx.x) lambda_from<STOCH>(d, prm, K, k, pre_sum, pre_s1, pre_s2, d.fold ? s3v[k] : pre_s3, l0, l1, s1r, s2r);
    else lambda_of_sweep<STOCH>(d, prm, K, k, s3v, l0, l1, s1r, s2r);
    d.lambda[2 * k] = l0;
    d.lambda[2 * k + 1] = l1;
    if constexpr (STOCH) { d.s12run[k] = s1r; d.s12run[K + k] = s2r; }
    const double ps = digamma(l0 + l1, logtab);
    d.elogbeta[2 * k] = digamma(l0, logtab) - ps;
    d.elogbeta[2 * k + 1] = digamma(l1, logtab) - ps;
  }
  if (threadIdx.x == 0) {
    sz = tot[0]; so = tot[1]; kzd = tot[2];
    c.parity ^= 1u;  // prune()'s flags become current
    if (d.lpl) { c.links_dense = ltot[3]; c.links_sparse = ltot[4]; c.links_shortcut = ltot[5]; }
    else { c.links_dense = (unsigned long long)tot[3]; c.links_sparse = (unsigned long long)tot[4]; c.links_shortcut = (unsigned long long)tot[5]; }
    if (d.sweep_stats) {
      unsigned long long *st = d.sweep_stats + (size_t)(c.sweeps_done % d.sweep_stats_cap) * 4;
      st[0] = c.links_dense; st[1] = c.links_sparse; st[2] = c.links_shortcut; st[3] = c.sweeps_done;
    }
    c.sweeps_done++;
    // (mini-batch steps tag on every step: a window is only visited once per pass over the nodes)
    c.write_comm = (STOCH || iter % prm.reportfreq == prm.reportfreq - 1) ? 1 : 0;
    bool exit_now = false;
    if (do_val) {
      const double szeros = sz, sones = so;
      const uint32_t kzeros = (uint32_t)kzd, kones = d.nv - kzeros;
      const double mean0 = szeros / kzeros, mean1 = sones / kones;
      const double a = prm.zeros_prob * mean0 + prm.ones_prob * mean1;
      double *row = d.rows + (size_t)(c.rows % d.rows_cap) * 10;
      row[0] = (double)iter; row[1] = (szeros + sones) / d.nv; row[2] = (double)d.nv;
      row[3] = mean0; row[4] = (double)kzeros; row[5] = mean1; row[6] = (double)kones;
      row[7] = prm.zeros_prob * mean0; row[8] = prm.ones_prob * mean1; row[9] = a;
      c.rows++;
      bool stop = false;
      int why = -1;
      if (iter > 10) {
        const double prev = c.prev_h;
        if (a > prev && prev != 0 && fabs((a - prev) / prev) < 0.00001) { stop = true; why = 100; }
        else if (a < prev) c.nh++;
        else if (a > prev) c.nh = 0;
        if (a > c.max_h) c.max_h = a;
        if (c.nh > 2) { why = 1; stop = true; }
      }
      c.prev_h = a;
      if (c.annealing && stop) {
        c.annealing = 0; c.nh = 0; c.prev_h = 0;  // max.txt keeps the pre-switch `why`
      } else if (!c.annealing && stop) {
        if (prm.use_validation_stop) exit_now = true;
      }
      c.why = why;
    }
    if (exit_now) c.stopped = 1;  // do_on_stop(); exit(0): _iter is not advanced
    else c.iter = iter + 1;
    // the classes the count / scatter passes computed for the next sweep become current -- unless they did not run
    // because no flag changed (cls_args[5], the count pass's verdict): then the current ones stay
    if (!(d.lpl && d.cls_next && d.cls_args[5] == 0u)) c.cls_par ^= 1u;
    store_ctrl_of_tail(d.ctrl, c);
  }
  STAMP(3, 6);
  // the link counts / shortcut histogram of the finished sweep are consumed: clear them for the
  // classification two sweeps ahead
  __syncthreads();
  if (d.lpl) {
    if (!(d.cls_next && d.cls_args[5] == 0u)) {
      if (threadIdx.x < 8) ltot[threadIdx.x] = 0;
      for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) d.shist[(size_t)cpar0 * K + k] = 0ull;
    }
    for (uint32_t k = threadIdx.x; k < 1024u; k += blockDim.x) d.sumfx[(size_t)cpar0 * 1024 + k] = 0;
  }
}

// likelihood row without the stop rule (the constructor's call, :149-150)
__global__ __launch_bounds__(256) void k_row_only(DeviceState d, Params prm, double *row_out) {
  __shared__ double red[2][256];
  __shared__ unsigned int cntz[256];
  double sz = 0.0, so = 0.0;
  unsigned int kz = 0;
  const uint32_t per = (d.nv + blockDim.x - 1) / blockDim.x;
  const uint32_t b = threadIdx.x * per, e = min(d.nv, b + per);
  for (uint32_t i = b; i < e; ++i) {
    const double u = d.uval[i];
    if (d.vpairs[3 * (size_t)i + 2]) so += u; else { sz += u; kz++; }
  }
  red[0][threadIdx.x] = sz; red[1][threadIdx.x] = so; cntz[threadIdx.x] = kz;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      cntz[threadIdx.x] += cntz[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double szeros = red[0][0], sones = red[1][0];
    const uint32_t kzeros = cntz[0], kones = d.nv - cntz[0];
    const double mean0 = szeros / kzeros, mean1 = sones / kones;
    row_out[0] = (double)d.ctrl->iter; row_out[1] = (szeros + sones) / d.nv; row_out[2] = (double)d.nv;
    row_out[3] = mean0; row_out[4] = (double)kzeros; row_out[5] = mean1; row_out[6] = (double)kones;
    row_out[7] = prm.zeros_prob * mean0; row_out[8] = prm.ones_prob * mean1;
    row_out[9] = prm.zeros_prob * mean0 + prm.ones_prob * mean1;
  }
}

// Elogbeta from lambda (set_dir_exp(_lambda, _Elogbeta), src/linksampling.cc:124,563)
__global__ __launch_bounds__(256) void k_lambda_exp(Geometry geo, DeviceState d) {
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < geo.K; k += gridDim.x * blockDim.x) {
    const double l0 = d.lambda[2 * k], l1 = d.lambda[2 * k + 1];
    const double ps = digamma(l0 + l1, logtab);
    d.elogbeta[2 * k] = digamma(l0, logtab) - ps;
    d.elogbeta[2 * k + 1] = digamma(l1, logtab) - ps;
  }
}

// device special functions on plain arrays, for the unit tests (which: 0 digamma, 1 exp_neg,
// 2 fast_rcp, 3 log_tab)
__global__ __launch_bounds__(256) void k_debug_eval(DeviceState d, int which, const double *in, double *out,
                                                    uint32_t n) {
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double x = in[i];
    out[i] = which == 0 ? digamma(x, logtab) : which == 1 ? exp_neg(x) : which == 2 ? fast_rcp(x) : log_tab(x, logtab);
  }
}

// ------------------------------------------------------------------ launchers
bool pick_layout(uint32_t K, int *W, int *V) {
  if (K == 0 || K > SVILS_MAX_K) return false;
  if (K <= 8) { *W = 8; *V = 1; }
  else if (K <= 16) { *W = 16; *V = 1; }
  else if (K <= 32) { *W = 32; *V = 1; }
  else if (K <= 64) { *W = 64; *V = 1; }
  else if (K <= 128) { *W = 64; *V = 2; }
  else if (K <= 256) { *W = 64; *V = 4; }
  else if (K <= 512) { *W = 64; *V = 8; }
  else if (K <= 768) { *W = 64; *V = 12; }    // six 16-byte chunks per lane: K = 513..768 does not pay for 1024 columns
  else if (K <= 1024) { *W = 64; *V = 16; }
  else { *W = 64; *V = 32; }
  return true;
}

#define SVILS_DISPATCH(geo, CALL)                                              \
  do {                                                                         \
    if ((geo).W == 8) { CALL(8, 1); }                                          \
    else if ((geo).W == 16) { CALL(16, 1); }                                   \
    else if ((geo).W == 32) { CALL(32, 1); }                                   \
    else if ((geo).V == 1) { CALL(64, 1); }                                    \
    else if ((geo).V == 2) { CALL(64, 2); }                                    \
    else if ((geo).V == 4) { CALL(64, 4); }                                    \
    else if ((geo).V == 8) { CALL(64, 8); }                                    \
    else if ((geo).V == 12) { CALL(64, 12); }                                  \
    else if ((geo).V == 16) { CALL(64, 16); }                                  \
    else { CALL(64, 32); }                                                     \
  } while (0)

// K > 32: one row per wavefront, W == 64
#define SVILS_DISPATCH_V(geo, CALL)                                            \
  do {                                                                         \
    if ((geo).V == 1) { CALL(64, 1); }                                         \
    else if ((geo).V == 2) { CALL(64, 2); }                                    \
    else if ((geo).V == 4) { CALL(64, 4); }                                    \
    else if ((geo).V == 8) { CALL(64, 8); }                                    \
    else if ((geo).V == 12) { CALL(64, 12); }                                  \
    else if ((geo).V == 16) { CALL(64, 16); }                                  \
    else { CALL(64, 32); }                                                     \
  } while (0)

void launch_phi(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  if (d.lpl) { launch_phi_lpl(g, d, p, s); return; }
  // link_thresh < 1/2 needs the argmax form of the tagging rule (k_phi<V, true>)
#define PHI(V_)                                                                                    \
  do {                                                                                             \
    if (p.link_thresh < 0.5) hipLaunchKernelGGL((k_phi<V_, true, false>), dim3(d.nb_a), dim3(256), 0, s, g, d, p); \
    else if (d.epi && d.skip_elogpi && V_ <= 8) {                                                                 \
      hipLaunchKernelGGL((k_phi<(V_ <= 8 ? V_ : 8), false, true, 1>), dim3(d.nb_a), dim3(256), 0, s, g, d, p);   \
      hipLaunchKernelGGL((k_phi<(V_ <= 8 ? V_ : 8), false, true, 2>), dim3(d.nb_a), dim3(256), 0, s, g, d, p);   \
    } else if (d.epi) hipLaunchKernelGGL((k_phi<V_, false, true>), dim3(d.nb_a), dim3(256), 0, s, g, d, p);       \
    else hipLaunchKernelGGL((k_phi<V_, false, false>), dim3(d.nb_a), dim3(256), 0, s, g, d, p);                  \
  } while (0)
  switch (g.V) {   // K > 32 => W == 64
    case 1: PHI(1); break;
    case 2: PHI(2); break;
    case 4: PHI(4); break;
    case 8: PHI(8); break;
    case 12: PHI(12); break;
    case 16: PHI(16); break;
    default: PHI(32); break;
  }
#undef PHI
}
// Blocks of the row-per-wavefront phi / s3 kernels resident on the device at once.  Grids are
// made a whole multiple of this: with variable-length items more blocks balance better, but a
// last partial round of blocks idles most of the chip (measured: 2048 blocks at 1536 resident
// cost +12 % on k_s3<64,8>).
uint32_t rpw_resident_blocks(const Geometry &g, int which, int device) {
  int per_cu = 0, cus = 0;
#define OCC(KERNEL) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 256, 0)
  if (which == 0) {
    switch (g.V) {
      case 1: OCC((k_phi<1, false, false>)); break;
      case 2: OCC((k_phi<2, false, false>)); break;
      case 4: OCC((k_phi<4, false, false>)); break;
      case 8: OCC((k_phi<8, false, false>)); break;
      case 12: OCC((k_phi<12, false, false>)); break;
      case 16: OCC((k_phi<16, false, false>)); break;
      default: OCC((k_phi<32, false, false>)); break;
    }
  } else if (which == 2) {
#define CALL(W_, V_) OCC((k_finalize<W_, V_, false>))
    SVILS_DISPATCH_V(g, CALL);
#undef CALL
  } else {
    switch (g.V) {
      case 1: OCC((k_s3<64, 1, false>)); break;
      case 2: OCC((k_s3<64, 2, false>)); break;
      case 4: OCC((k_s3<64, 4, false>)); break;
      case 8: OCC((k_s3<64, 8, false>)); break;
      case 12: OCC((k_s3<64, 12, false>)); break;
      case 16: OCC((k_s3<64, 16, false>)); break;
      default: OCC((k_s3<64, 32, false>)); break;
    }
  }
#undef OCC
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (per_cu <= 0 || cus <= 0) return 1024;
  return (uint32_t)per_cu * (uint32_t)cus;
}
void launch_reduce_a(const Geometry &g, const DeviceState &d, hipStream_t s) {
  const ReduceJob j0{d.part_a, d.kvec_a, d.nb_a, g.K};
  const uint32_t nblk0 = (g.K + 3) / 4;
  hipLaunchKernelGGL(k_colreduce, dim3(nblk0), dim3(256), 0, s, j0, j0, nblk0, d.ctrl);
}
void launch_finalize(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  if (d.lpl) { launch_finalize_lpl(g, d, p, s); return; }
#define CALL(W_, V_)                                                                                     \
  do {                                                                                                   \
    if (p.stoch) hipLaunchKernelGGL((k_finalize<W_, V_, true>), dim3(d.nb_b), dim3(256), 0, s, g, d, p); \
    else if (d.light) hipLaunchKernelGGL((k_finalize<W_, V_, false, true>), dim3(d.nb_b), dim3(256), 0, s, g, d, p); \
    else hipLaunchKernelGGL((k_finalize<W_, V_, false>), dim3(d.nb_b), dim3(256), 0, s, g, d, p);        \
  } while (0)
  SVILS_DISPATCH_V(g, CALL);
#undef CALL
}
void launch_s3(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  if (d.lpl) { launch_s3_lpl(g, d, p, s); return; }
#define CALL(V_)                                                                                         \
  do {                                                                                                   \
    if (d.derive_m) hipLaunchKernelGGL((k_s3<64, V_, true>), dim3(d.nb_c), dim3(256), 0, s, g, d, p);    \
    else hipLaunchKernelGGL((k_s3<64, V_, false>), dim3(d.nb_c), dim3(256), 0, s, g, d, p);              \
  } while (0)
  switch (g.V) {   // row-per-wavefront layout: W == 64 whatever K
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    case 12: CALL(12); break;
    case 16: CALL(16); break;
    default: CALL(32); break;
  }
#undef CALL
}
void launch_reduce_c(const Geometry &g, const DeviceState &d, hipStream_t s) {
  const ReduceJob j0{d.part_b, d.kvec_c, d.nb_b, 2 * g.K};
  const ReduceJob j1{d.part_c, d.kvec_c + 2 * (size_t)g.K, d.nb_c, g.K};
  const uint32_t nblk0 = (2 * g.K + 3) / 4, nblk1 = (g.K + 3) / 4;
  hipLaunchKernelGGL(k_colreduce, dim3(nblk0 + nblk1), dim3(256), 0, s, j0, j1, nblk0, d.ctrl);
}
void launch_validation(const Geometry &g, const DeviceState &d, const Params &, hipStream_t s) {
  if (d.nv == 0) return;
  const int G = 64 / g.W;
  uint32_t nb = (d.nv + 4 * G - 1) / (4 * G);
  if (nb > 2048) nb = 2048;
#define CALL(W_, V_) hipLaunchKernelGGL((k_validation<W_, V_>), dim3(nb), dim3(256), 0, s, g, d)
  SVILS_DISPATCH(g, CALL);
#undef CALL
}
// blocks of k_tail: enough groups for one or two passes over the held-out pairs, few enough that
// the last block's in-order sum of the block partials stays short
uint32_t tail_blocks(const Geometry &g, uint32_t nv) {
  const uint32_t per_block = 4u * (uint32_t)(64 / g.W);
  uint32_t nb = (nv + 2 * per_block - 1) / (2 * per_block);
  if (nb < 1) nb = 1;
  // Up to 256 blocks: one thread of the last block per block partial.  A large held-out set (n = 1e6: ~2.4e5 pairs) is a
  // chain of dependent row gathers per group, so it gets up to SVILS_TAIL_BLOCKS blocks (four per CU) and the last
  // block's threads add four partials each, in block order.
  if (nb > 256) nb = nb > SVILS_TAIL_BLOCKS ? SVILS_TAIL_BLOCKS : (nb + 255u) / 256u * 256u;
  return nb;
}
void launch_tail(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
#define CALL(W_, V_)                                                                                   \
  do {                                                                                                 \
    if (p.stoch) hipLaunchKernelGGL((k_tail<W_, V_, true>), dim3(d.nb_t), dim3(256), 0, s, g, d, p);   \
    else hipLaunchKernelGGL((k_tail<W_, V_, false>), dim3(d.nb_t + lpl_scatter_blocks(d)), dim3(256), 0, s, g, d, p); \
  } while (0)
  SVILS_DISPATCH(g, CALL);
#undef CALL
}
void launch_carry_flags(const Geometry &g, const DeviceState &d, hipStream_t s) {
  if (g.node_end - g.node_begin >= g.n) return;
  uint32_t nb = (g.n + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_carry_flags, dim3(nb), dim3(256), 0, s, g, d);
}
void launch_expand_window(const Geometry &g, const DeviceState &d, const Params &, uint32_t wb, uint32_t we,
                          uint32_t block, uint32_t my_rank, uint32_t world, hipStream_t s) {
  if (world <= 1 || we <= wb) return;
  const uint32_t total = (we - wb) * (world - 1);
  const int G = 64 / g.W;
  uint32_t nb = (total + 4 * G - 1) / (4 * G);
  if (nb > 2048) nb = 2048;
#define CALL(W_, V_) hipLaunchKernelGGL((k_expand_window<W_, V_>), dim3(nb), dim3(256), 0, s, g, d, wb, we, block, my_rank, world)
  SVILS_DISPATCH(g, CALL);
#undef CALL
}
// rows [xb, xe) of every other rank's node block of `block` rows (n_alloc = block * world, the owned block starts at a
// multiple of it); block == 0: every row the handle does not own
void launch_expand_chunk(const Geometry &g, const DeviceState &d, const Params &p, uint32_t xb, uint32_t xe, uint32_t block,
                         hipStream_t s) {
  const uint32_t nown = g.node_end - g.node_begin;
  if (nown >= g.n) return;
  uint64_t total;
  if (block) {
    if (xe <= xb) return;
    total = (uint64_t)(xe - xb) * ((g.n_alloc + block - 1) / block - 1u);
  } else {
    total = g.n - nown;
  }
  if (total == 0) return;
  const int G = 64 / g.W;
  uint32_t nb = (uint32_t)std::min<uint64_t>((total + 4 * G - 1) / (4 * G), 2048);
#define CALL(W_, V_) hipLaunchKernelGGL((k_expand<W_, V_>), dim3(nb), dim3(256), 0, s, g, d, p, xb, xe, block)
  SVILS_DISPATCH(g, CALL);
#undef CALL
}
void launch_expand(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  launch_expand_chunk(g, d, p, 0, 0, 0, s);
}
void launch_expand_all(const Geometry &g, const DeviceState &d, const Params &p, const Blocks &b, hipStream_t s) {
  const uint32_t clen = b.nchunks > 1 ? (b.bmax + b.nchunks - 1) / b.nchunks + 1u : b.bmax;
  const uint64_t total = (uint64_t)b.world * clen;
  if (total == 0) return;
  const int G = 64 / g.W;
  const uint32_t nb = (uint32_t)std::min<uint64_t>((total + 4 * G - 1) / (4 * G), 2048);
#define CALL(W_, V_) hipLaunchKernelGGL((k_expand_all<W_, V_>), dim3(nb), dim3(256), 0, s, g, d, p, b)
  SVILS_DISPATCH(g, CALL);
#undef CALL
}
void launch_dir_exp(const Geometry &g, const DeviceState &d, hipStream_t s) {
  const int G = 64 / g.W;
  uint32_t nb = (g.n + 4 * G - 1) / (4 * G);
  if (nb > 2048) nb = 2048;
#define CALL(W_, V_) hipLaunchKernelGGL((k_dir_exp<W_, V_>), dim3(nb), dim3(256), 0, s, g, d)
  SVILS_DISPATCH(g, CALL);
#undef CALL
}
void launch_lambda_exp(const Geometry &g, const DeviceState &d, hipStream_t s) {
  hipLaunchKernelGGL(k_lambda_exp, dim3((g.K + 255) / 256), dim3(256), 0, s, g, d);
}
void launch_debug_eval(const DeviceState &d, int which, const double *in, double *out, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_eval, dim3((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), dim3(256), 0, s, d, which, in, out, n);
}
void launch_row_only(const Geometry &g, const DeviceState &d, const Params &p, double *row_out,
                     hipStream_t s) {
  (void)g;
  hipLaunchKernelGGL(k_row_only, dim3(1), dim3(256), 0, s, d, p, row_out);
}

}  // namespace svils

#include "svils_ksh.h"
