// svils_ksh.h -- K-sharded sweeps (included once, at the end of svils_device.hip, whose helpers it uses): every rank keeps the columns [K0, K0 + K) of all n rows.
//
// The sweep of src/linksampling.cc:556-790 couples the columns of a row in four places only: the softmax
// denominator of a link, the row sum inside Elogpi = psi(gamma) - psi(sum_k gamma), the active-community
// count of prune(), and the dot products of the held-out likelihood (plus quirk Q2, where mphi[q][pc] feeds
// s3[pc - 1]).  Each becomes a buffer of partials that the caller (or svils_sweep_ksharded over RCCL) SUMs
// over the ranks between two phases; everything else runs on the rank's own columns with the row-per-
// wavefront layout and the product form of the phi pass (exp(Elogpi) rows).  tests/test_ksharded_protocol.py
// is the same protocol in numpy against the oracle.
//
//   DEN   k_phi_ksh<V,1>   den[link] = sum over own columns of e^x_k             -> SUM den        (L doubles)
//   PHI   k_phi_ksh<V,2>   gammanext rows from e^x_k / den, `sum`, tags; k_colreduce; k_fin1_ksh:
//                          mean indicators, gamma, partial row sums / active counts -> SUM rowx    (3n doubles)
//                          and -- product form -- the exp(Elogpi) rows of the NEXT sweep: the softmax of a link does not see
//                          a constant added to a row, so psi(gamma) is shifted by psi of the row sum the ranks ALREADY share
//                          (last sweep's, in rowx) instead of waiting for this sweep's: no second pass over the rows
//   FIN2  k_flags_ksh      prune() flags from the summed rowx (O(n); log-domain mode: k_fin2_ksh, Elogpi rows shifted by THIS
//                          sweep's row sum); k_s3_ksh + k_colreduce (Q2 across the slice edge)  -> SUM q2v       (Kt doubles)
//   LAM   k_lam_ksh        lambda, Elogbeta of own columns; k_vdot_ksh partial dot products -> SUM vdot (nv doubles)
//   STOP  k_stop_ksh       likelihood row, stop rule, annealing switch, loop control (replicated)
namespace svils {

// ---------------------------------------------------------------- phi pass, K-sharded
// LOG: the log-domain form for models whose denominators can underflow (max_k x_k < -745: concentrated memberships at
// K >~ 740): MODE 0 writes the link's max over the own columns (-> MAX over the ranks), MODE 1 sums e^(x_k - max),
// MODE 2 accumulates e^(x_k - max) / den -- two L-sized exchanges and K exps per link and pass instead of one exchange
// and multiplies, which is why it is a mode (svils_ksh_log_domain; on by default above K = 700).
template <int V, int MODE, bool LOG>
__global__ __launch_bounds__(256) void k_phi_ksh(Geometry geo, DeviceState d, Params prm) {
  constexpr int W = 64;
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  __shared__ double lds[V * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lw = lane;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool write_comm = ctrl->write_comm != 0;
  const bool sparse_iter = (long long)ctrl->iter > (long long)prm.sparse_after;
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  const double *__restrict__ epi = LOG ? d.elogpi : d.epi;   // LOG: rows of Elogpi
  int kidx[V];
  bool kval[V];
  double eb[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    kidx[v] = kmap<W, V>(lw, v);
    kval[v] = (uint32_t)kidx[v] < K;
    eb[v] = kval[v] ? (LOG ? d.elogbeta[2 * kidx[v]] : exp_neg(d.elogbeta[2 * kidx[v]])) : (LOG ? NEG_INF : 0.0);
  }
  double csum[1][V];
#pragma unroll
  for (int v = 0; v < V; ++v) csum[0][v] = 0.0;
  unsigned long long n_dense = 0, n_sparse = 0, n_short = 0;

  for (uint32_t it = blockIdx.x * 4 + wave; it < d.nitems_phi; it += gridDim.x * 4) {
    const Item item_ = d.items_phi[d.item0_phi + it];
    const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.node);
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.off);
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.len);
    const int32_t slot = __builtin_amdgcn_readfirstlane(item_.slot);
    const uint64_t base = d.rowptr[p] + off;
    uint32_t mycol = 0, myconv = 0, myel = 0;
    if ((uint32_t)lane < len) {
      mycol = d.col[base + lane];
      myconv = conv[mycol];
      // mini-batch steps (ksh_ent): the per-link buffers are indexed by CSR ENTRY and every entry of the window's rows
      // computes its own value -- a link whose other endpoint lies outside the window has no lower-endpoint row in
      // this step, and the window's entries are one contiguous range to exchange
      myel = d.ksh_ent ? (uint32_t)(base + lane) : d.elink[base + lane];
    }
    const uint32_t pc = (uint32_t)__builtin_amdgcn_readfirstlane((int)conv[p]);
    double ap[V];
    load_row<W, V>(epi + (size_t)p * ld, lw, ld, ap);
#pragma unroll
    for (int v = 0; v < V; ++v) ap[v] = LOG ? ap[v] + eb[v] : ap[v] * eb[v];
    double acc[V];
    uint32_t cnt[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { acc[v] = 0.0; cnt[v] = 0; }
    uint32_t p_active = 0;
    if (sparse_iter) p_active = d.active_cnt[p];

    // a neighbour needs its row when its link takes the softmax branch and (MODE 1) is counted here; the row of
    // the next one that does is in flight while the current one is reduced
    auto needs_row = [&](uint32_t jj) {
      const uint32_t qq = __builtin_amdgcn_readlane(mycol, jj), cc = __builtin_amdgcn_readlane(myconv, jj);
      return ((pc != 0) == (cc != 0)) && (MODE == 2 || qq > p || d.ksh_ent);   // MODE 0 and 1: one value per undirected link
    };
    double r[V], rnext[V];
    // (MODE 2: the link's summed denominator travels with its row -- requested one neighbour ahead, like the row)
    double dcur = 1.0, dnext = 1.0;
    if (len > 0 && needs_row(0)) {
      load_row<W, V>(epi + (size_t)__builtin_amdgcn_readlane(mycol, 0) * ld, lw, ld, r);
      if constexpr (MODE == 2) dcur = d.den[__builtin_amdgcn_readlane(myel, 0)];
    }
    for (uint32_t j = 0; j < len; ++j) {
      const uint32_t q = __builtin_amdgcn_readlane(mycol, j);
      const uint32_t qc = __builtin_amdgcn_readlane(myconv, j);
      const uint32_t el = __builtin_amdgcn_readlane(myel, j);
      if (j + 1 < len && needs_row(j + 1)) {
        load_row<W, V>(epi + (size_t)__builtin_amdgcn_readlane(mycol, j + 1) * ld, lw, ld, rnext);
        if constexpr (MODE == 2) dnext = d.den[__builtin_amdgcn_readlane(myel, j + 1)];
      }
      const bool count_me = q > p;
      bool handled = false;
      if ((pc != 0) != (qc != 0)) {
        if constexpr (MODE == 2) {
          // exactly one endpoint converged: +1 at the converged community, on the rank that holds it (:622-631)
          const int c = (int)(pc ? pc : qc) - 1 - (int)geo.K0;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kidx[v] == c) acc[v] += 1.0;
          if (count_me && lane == 0) n_short++;
        }
        handled = true;
      } else if (MODE != 2 && !count_me && !d.ksh_ent) {
        handled = true;   // one denominator per undirected link
      }
      if (!handled) {   // (body kept at loop depth)
      bool sparse = false;
      if (sparse_iter) sparse = p_active < geo.k10 && d.active_cnt[q] < geo.k10;
      double e[V];
      double s = 0.0;
      if constexpr (LOG) {
        double m = NEG_INF;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          double t = ap[v] + r[v];   // padding columns: -inf
          if (sparse) {
            const uint64_t um = d.amask[(size_t)p * geo.kw + v] | d.amask[(size_t)q * geo.kw + v];
            t = ((um >> lw) & 1ull) ? t : NEG_INF;
          }
          e[v] = t;
          m = fmax(m, t);
        }
        if constexpr (MODE == 0) {
          m = group_max<W>(m);
          if (lane == 0) d.dmax[el] = m;
        } else {
          const double M = d.dmax[el];   // over ALL columns; -inf: empty active-set union
          if (d.ksh_lowt) {
            // the lowest own column attaining the maximum (global index; none: +inf) -- MODE 1 publishes it for the
            // MIN over the ranks, MODE 2 compares its own columns with the result
            double cand = __builtin_huge_val();
#pragma unroll
            for (int v = 0; v < V; ++v)
              if (kval[v] && M != NEG_INF && e[v] == M) cand = fmin(cand, (double)(geo.K0 + (uint32_t)kidx[v]));
            if constexpr (MODE == 1) {
              cand = -group_max<W>(-cand);
              if (lane == 0) d.earg[el] = cand;
            }
          }
#pragma unroll
          for (int v = 0; v < V; ++v) { e[v] = (M == NEG_INF) ? 0.0 : exp_neg(e[v] - M); s += e[v]; }
        }
      } else {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          double t = ap[v] * r[v];
          if (sparse) {
            const uint64_t um = d.amask[(size_t)p * geo.kw + v] | d.amask[(size_t)q * geo.kw + v];
            t = ((um >> lw) & 1ull) ? t : 0.0;
          }
          e[v] = t;
          s += t;
        }
      }
      if constexpr (MODE == 0) {
        // (LOG only) nothing else to do for this link
      } else if constexpr (MODE == 1) {
        s = group_sum<W>(s);
        if (lane == 0) d.den[el] = s;
      } else {
        s = dcur;        // the link's denominator over ALL columns (den[el], requested with the row)
        // a denominator that underflowed (possible only for rows of disjoint support at very large K): the product form
        // has no way back -- say so instead of dropping the link (the log-domain mode cannot get here: its sum is >= 1)
        if (!LOG && s < 1e-280 && !(sparse && s == 0.0)) ctrl->fault = 2u;
        if (s > 0.0) {   // 0: empty active-set union, contributes nothing (:642-664)
          const double inv = fast_rcp(s);
          const double ts = prm.link_thresh * s;
          // link_thresh < 1/2: the tag goes to the first strict maximum (phi there is e^0 / s = 1 / s)
          const double amax = (LOG && d.ksh_lowt) ? d.earg[el] : -1.0;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            acc[v] = fma(e[v], inv, acc[v]);
            if (write_comm) {
              if (LOG && d.ksh_lowt) cnt[v] += (kval[v] && (double)(geo.K0 + (uint32_t)kidx[v]) == amax && 1.0 > ts) ? 1u : 0u;
              else cnt[v] += (e[v] > ts) ? 1u : 0u;
            }
          }
        }
        if (count_me && lane == 0) { if (sparse) n_sparse++; else n_dense++; }
      }
      }
#pragma unroll
      for (int v = 0; v < V; ++v) r[v] = rnext[v];
      dcur = dnext;
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int v = 0; v < V; ++v) csum[0][v] += acc[v];
      if (slot < 0) {
        store_row<W, V>(d.gacc + (size_t)p * ld, lw, ld, acc);
        if (write_comm) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const unsigned long long b = __ballot(kval[v] && cnt[v] > prm.lt_min_deg);
            if (lane == 0) d.member[(size_t)p * geo.kw + v] = b;
          }
        }
      } else {
        store_row<W, V>(d.parts + (size_t)slot * ld, lw, ld, acc);
        if (write_comm) {
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kval[v]) d.part_cnt[(size_t)slot * ld + kidx[v]] = cnt[v];
        }
      }
    }
  }
  if constexpr (MODE == 2) {
    block_reduce_store<W, V, 1>(csum, d.part_a + (size_t)blockIdx.x * K, K, lds);
    __shared__ unsigned long long lcnt[3 * 4];
    block_store_link_counts(n_dense, n_sparse, n_short, d.part_links, lcnt, 4);
  }
}

// ---------------------------------------------------------------- phi pass, K-sharded, slices of <= 64 columns
// A 64-column slice is one double per lane in k_phi_ksh<1,.>: every neighbour pays the fixed per-row cost (read-lanes,
// a six-step reduction, reciprocal) for 512 bytes.  Here a row is spread over 16 lanes x 4 doubles and the four
// 16-lane groups of a wavefront take four DIFFERENT neighbours of the node per step: the reduction is four DPP steps
// inside a row of lanes, shared by the four neighbours, and the bookkeeping is issued once per four.  The handle's
// bitmask arrays (amask in, member out) are one word per node here (bit k = column k of the slice).
template <int MODE>
__global__ __launch_bounds__(256) void k_phi_ksh16(Geometry geo, DeviceState d, Params prm) {
  constexpr int W = 16, V = 4;
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  __shared__ double lds[V * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, lw = lane & 15;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool write_comm = ctrl->write_comm != 0;
  const bool sparse_iter = (long long)ctrl->iter > (long long)prm.sparse_after;
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  const double *__restrict__ epi = d.epi;
  int kidx[V];
  bool kval[V];
  double eb[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    kidx[v] = kmap<W, V>(lw, v);
    kval[v] = (uint32_t)kidx[v] < K;
    eb[v] = kval[v] ? exp_neg(d.elogbeta[2 * kidx[v]]) : 0.0;
  }
  double csum[1][V];
#pragma unroll
  for (int v = 0; v < V; ++v) csum[0][v] = 0.0;
  uint32_t n_dense = 0, n_sparse = 0, n_short = 0;   // wave-uniform

  for (uint32_t it = blockIdx.x * 4 + wave; it < d.nitems_phi; it += gridDim.x * 4) {
    const Item item_ = d.items_phi[d.item0_phi + it];
    const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.node);
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.off);
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.len);
    const int32_t slot = __builtin_amdgcn_readfirstlane(item_.slot);
    const uint64_t base = d.rowptr[p] + off;
    uint32_t mycol = 0, myconv = 0, myel = 0;
    if ((uint32_t)lane < len) {
      mycol = d.col[base + lane];
      myconv = conv[mycol];
      myel = d.ksh_ent ? (uint32_t)(base + lane) : d.elink[base + lane];   // see k_phi_ksh
    }
    const uint32_t pc = (uint32_t)__builtin_amdgcn_readfirstlane((int)conv[p]);
    double ap[V];
    load_row<W, V>(epi + (size_t)p * ld, lw, ld, ap);
#pragma unroll
    for (int v = 0; v < V; ++v) ap[v] *= eb[v];
    double acc[V];
    uint32_t cnt[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { acc[v] = 0.0; cnt[v] = 0; }
    uint32_t p_active = 0;
    unsigned long long pmask = 0ull;
    if (sparse_iter) { p_active = d.active_cnt[p]; pmask = d.amask[p]; }

    // this group's neighbour of the four starting at j, and whether it needs its row
    auto mine = [&](uint32_t j, uint32_t &q, uint32_t &qc, uint32_t &el, bool &need) {
      const uint32_t jj = j + (uint32_t)g;
      const int src = (int)(jj < len ? jj : 0u);
      q = (uint32_t)__shfl((int)mycol, src, 64);
      qc = (uint32_t)__shfl((int)myconv, src, 64);
      el = (uint32_t)__shfl((int)myel, src, 64);
      const bool valid = jj < len;
      need = valid && ((pc != 0) == (qc != 0)) && (MODE == 2 || q > p || d.ksh_ent);
      return valid;
    };
    double r[V], rnext[V];
    uint32_t q, qc, el;
    bool need;
    bool valid = mine(0, q, qc, el, need);
    // (MODE 2: the link's summed denominator is requested with its row, four neighbours ahead of its use)
    double dcur = 1.0, dnext = 1.0;
    if (need) {
      load_row<W, V>(epi + (size_t)q * ld, lw, ld, r);
      if constexpr (MODE == 2) dcur = d.den[el];
    }
    for (uint32_t j = 0; j < len; j += 4) {
      uint32_t q1 = 0, qc1 = 0, el1 = 0;
      bool need1 = false, valid1 = false;
      dnext = 1.0;
      if (j + 4 < len) {
        valid1 = mine(j + 4, q1, qc1, el1, need1);
        if (need1) {
          load_row<W, V>(epi + (size_t)q1 * ld, lw, ld, rnext);
          if constexpr (MODE == 2) dnext = d.den[el1];
        }
      }
      const bool count_me = valid && lw == 0 && q > p;
      if constexpr (MODE == 2) {
        const bool shortc = valid && ((pc != 0) != (qc != 0));
        if (__any(shortc)) {
          // exactly one endpoint converged: +1 at the converged community, on the rank that holds it (:622-631)
          const int c = (int)(pc ? pc : qc) - 1 - (int)geo.K0;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (shortc && kidx[v] == c) acc[v] += 1.0;
          n_short += (uint32_t)__popcll(__ballot(shortc && count_me));
        }
      }
      if (__any(need)) {
        bool sparse = false;
        unsigned long long um = ~0ull;
        if (sparse_iter && need) {
          sparse = p_active < geo.k10 && d.active_cnt[q] < geo.k10;
          if (sparse) um = pmask | d.amask[q];
        }
        double e[V];
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          double t = need ? ap[v] * r[v] : 0.0;
          if (sparse) t = ((um >> kidx[v]) & 1ull) ? t : 0.0;
          e[v] = t;
          s += t;
        }
        if constexpr (MODE == 1) {
          s = group_sum<W>(s);
          if (need && lw == 0) d.den[el] = s;
        } else {
          s = need ? dcur : 1.0;        // the link's denominator over ALL columns (den[el], requested with the row)
          if (need && s < 1e-280 && !(sparse && s == 0.0)) ctrl->fault = 2u;   // see k_phi_ksh
          const bool live = need && s > 0.0;
          const double inv = live ? fast_rcp(s) : 0.0;
          const double ts = prm.link_thresh * s;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            acc[v] = fma(e[v], inv, acc[v]);
            if (write_comm) cnt[v] += (live && e[v] > ts) ? 1u : 0u;
          }
          n_sparse += (uint32_t)__popcll(__ballot(need && sparse && count_me));
          n_dense += (uint32_t)__popcll(__ballot(need && !sparse && count_me));
        }
      }
      q = q1; qc = qc1; el = el1; need = need1; valid = valid1;
      dcur = dnext;
#pragma unroll
      for (int v = 0; v < V; ++v) r[v] = rnext[v];
    }
    if constexpr (MODE == 2) {
      // the four groups' shares of the node
#pragma unroll
      for (int v = 0; v < V; ++v) {
        acc[v] += __shfl_xor(acc[v], 16, 64);
        acc[v] += __shfl_xor(acc[v], 32, 64);
        cnt[v] += (uint32_t)__shfl_xor((int)cnt[v], 16, 64);
        cnt[v] += (uint32_t)__shfl_xor((int)cnt[v], 32, 64);
      }
      if (g == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) csum[0][v] += acc[v];
      }
      if (slot < 0) {
        if (g == 0) store_row<W, V>(d.gacc + (size_t)p * ld, lw, ld, acc);
        if (write_comm) {
          unsigned long long w = 0ull;   // this lane's columns that are tagged, as bits of the node's word
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (g == 0 && kval[v] && cnt[v] > prm.lt_min_deg) w |= 1ull << kidx[v];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) w |= (unsigned long long)__shfl_xor((long long)w, o, 64);
          if (lane == 0) d.member[p] = w;
        }
      } else {
        if (g == 0) store_row<W, V>(d.parts + (size_t)slot * ld, lw, ld, acc);
        if (write_comm && g == 0) {
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kval[v]) d.part_cnt[(size_t)slot * ld + kidx[v]] = cnt[v];
        }
      }
    }
  }
  if constexpr (MODE == 2) {
    block_reduce_store<W, V, 1>(csum, d.part_a + (size_t)blockIdx.x * K, K, lds);
    __shared__ unsigned long long lcnt[3 * 4];
    block_store_link_counts(lane == 0 ? n_dense : 0ull, lane == 0 ? n_sparse : 0ull, lane == 0 ? n_short : 0ull,
                            d.part_links, lcnt, 4);
  }
}

// ---------------------------------------------- node finalise, first half: up to the new gamma
// compute_mean_indicators + swap (src/linksampling.cc:526-545,751-755) on the own columns; what set_dir_exp and
// prune need from the WHOLE row goes to rowx[p] as this rank's partial: sum_k gamma, |{k: gamma - alpha >= 1}|,
// sum of (community + 1) over that set (the community itself when the set has one element).
// W = 64: one node per wavefront, V columns per lane.  W = 16, V = 4 (slices of <= 64 columns, see k_phi_ksh16): FOUR
// nodes per wavefront, one per 16-lane row, so that a 512-byte row does not pay a wavefront's fixed costs alone and
// every lane carries four independent chains.
// fuse (product form, sweeps and steps): this kernel also leaves what k_fin2_ksh would -- the exp(Elogpi) row of the new gamma,
// shifted by psi of the row sum every rank already holds (rowx[3p] still carries the SUMMED row sum of the previous state here;
// any per-row constant cancels in a link's softmax, and this one is the same on every rank) -- and the row's active-set
// candidate bits; what needs the summed rowx of THIS state (the flags) follows in k_flags_ksh, O(n).
#ifndef KSH_FIN_PREFETCH   // 1: (four-node wavefronts) the next turn's loads are requested before this turn's arithmetic
#define KSH_FIN_PREFETCH 1
#endif
template <int W, int V, bool STOCH>
__global__ __launch_bounds__(256) void k_fin1_ksh(Geometry geo, DeviceState d, Params prm, int init, int fuse) {
  constexpr int G = 64 / W;
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  __shared__ double lds[2 * V * 64];
  __shared__ double2 logtab[128];
  if (STOCH || fuse) { load_logtab(logtab, d.logtab); __syncthreads(); }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool annealing = ctrl->annealing != 0;
  const bool write_comm = ctrl->write_comm != 0;
  int kidx[V];
  bool kval[V];
  double scale[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    kidx[v] = kmap<W, V>(lw, v);
    kval[v] = (uint32_t)kidx[v] < K;
    // (mini-batch step: kvec_a is the window's sum, scaled to an estimate of the full one)
    scale[v] = (!init && annealing && kval[v]) ? (double)prm.ones / (STOCH ? d.kvec_a[kidx[v]] * prm.scale_a : d.kvec_a[kidx[v]]) : 1.0;
  }
  double s12[2][V];
#pragma unroll
  for (int v = 0; v < V; ++v) { s12[0][v] = 0.0; s12[1][v] = 0.0; }
  // full sweeps: every node; mini-batch steps: the window [node_begin, node_end)
  // Everything a node's turn reads sits at an address the node number alone decides -- the two row pointers, the split marker,
  // the accumulator row (taken whether or not the node was split: a split node's row is ignored), the row sum the ranks
  // share -- so it is requested in one go, and (PREF) the NEXT turn's before this turn's arithmetic: the loop was three dependent
  // round trips per node (pointers + marker, then the row behind the branch on the marker, then the row sum behind the
  // stores it might alias), with 60 % of the launch's wave cycles parked on memory (profiles/r07p_kshard_sq.txt).
  struct Turn { uint64_t r0, r1; int32_t sf; double acc[V]; double rsum; };
  const uint32_t stride = gridDim.x * 4 * G;
  auto request = [&](uint32_t q0, Turn &t) {
    const uint32_t q = (q0 + (uint32_t)g < geo.node_end) ? q0 + (uint32_t)g : (q0 < geo.node_end ? q0 : geo.node_begin);
    t.r0 = d.rowptr[q]; t.r1 = d.rowptr[q + 1];
    t.sf = d.split_first[q];
    load_row<W, V>(d.gacc + (size_t)q * ld, lw, ld, t.acc);
    t.rsum = fuse ? d.rowx[3 * (size_t)q] : 1.0;
  };
  // (the run-ahead pays where a turn is short -- four 512-byte rows per wavefront: 597 -> 551 us at 64 columns per rank --, not where
  //  a wavefront owns one wide row: 1 033 -> 1 050 us at 128 columns, 1 837 -> 1 850 at 256; profiles/r07zc_ab_fin1_ksh_by_world.txt)
  constexpr bool PREF = KSH_FIN_PREFETCH && W == 16;
  const uint32_t pfirst = geo.node_begin + (blockIdx.x * 4 + wave) * G;
  Turn nxt;
  if constexpr (PREF) { if (!init && pfirst < geo.node_end) request(pfirst, nxt); }
  for (uint32_t p0 = pfirst; p0 < geo.node_end; p0 += stride) {
    const bool ok = p0 + (uint32_t)g < geo.node_end;       // (G > 1: the last wavefront's rows past the end idle on p0's node)
    const uint32_t p = ok ? p0 + (uint32_t)g : p0;
    double gn[V];
    if (init) {
      load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
#pragma unroll
      for (int v = 0; v < V; ++v) if (!kval[v]) gn[v] = 0.0;
    } else {
      Turn cur;
      if constexpr (PREF) { cur = nxt; if (p0 + stride < geo.node_end) request(p0 + stride, nxt); }
      else request(p0, cur);
      const double tl = 2.0 * (double)(cur.r1 - cur.r0);  // quirk Q3
      double acc[V];
      const int32_t sf = cur.sf;
      if (sf < 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = cur.acc[v];
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
      }
      // tags of a node whose entries were split over several wave-items (the others were tagged by the phi pass)
      if constexpr (W == 64) {
        if (sf >= 0 && write_comm) {
          const uint32_t sc = d.split_cnt[p];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            uint32_t c = 0;
            if (kval[v])
              for (uint32_t t = 0; t < sc; ++t) c += d.part_cnt[(size_t)(sf + t) * ld + kidx[v]];
            const unsigned long long b = __ballot(kval[v] && c > prm.lt_min_deg);
            if (lw == 0) d.member[(size_t)p * geo.kw + v] = b;
          }
        }
      } else if (write_comm) {          // one word per node (bit k = column k of the slice); wave-uniform branch
        unsigned long long w = 0ull;
        if (sf >= 0) {
          const uint32_t sc = d.split_cnt[p];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            uint32_t c = 0;
            if (kval[v])
              for (uint32_t t = 0; t < sc; ++t) c += d.part_cnt[(size_t)(sf + t) * ld + kidx[v]];
            if (kval[v] && c > prm.lt_min_deg) w |= 1ull << kidx[v];
          }
        }
#pragma unroll
        for (int o = 1; o < W; o <<= 1) w |= (unsigned long long)__shfl_xor((long long)w, o, 64);
        if (sf >= 0 && ok && lw == 0) d.member[p] = w;
      }
      if (tl > 0.0) {
        double m[V];
        const double rtl = 1.0 / tl;   // (one division per node, as k_finalize)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const double g0 = prm.alpha + acc[v];
          m[v] = (g0 - prm.alpha) * rtl;
          gn[v] = g0 + ((double)geo.n - tl - 1.0) * m[v];
          if (annealing) gn[v] *= scale[v];
          if (kval[v]) { if (ok) { s12[0][v] += m[v]; s12[1][v] += m[v] * m[v]; } }
          else { m[v] = 0.0; gn[v] = 0.0; }
        }
        if constexpr (STOCH) {
          // Robbins-Monro step of this node (k_finalize<W,V,true>): gamma <- (1 - rho) gamma + rho gamma_hat with
          // rho = (tau0 + c)^-kappa; s1/s2 are running sums over the stored mphi rows: this row contributes (new - old)
          const uint32_t c = d.ncnt[p];
          const double rho = exp_neg(-prm.kappa * log_tab(prm.tau0 + (double)c, logtab));
          double gold[V], mold[V];
          load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gold);
          load_row<W, V>(d.mphi + (size_t)p * ld, lw, ld, mold);
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kval[v]) {
              gn[v] = (1.0 - rho) * gold[v] + rho * gn[v];
              if (ok) { s12[0][v] -= mold[v]; s12[1][v] -= mold[v] * mold[v]; }
            }
          if (lw == 0 && ok) d.ncnt[p] = c + 1u;
        }
        if (ok) store_row<W, V>(d.mphi + (size_t)p * ld, lw, ld, m);
      } else {
#pragma unroll
        for (int v = 0; v < V; ++v) gn[v] = kval[v] ? prm.alpha : 0.0;
      }
      if (ok) store_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
      if (fuse) {
        const double psi_prev = digamma(cur.rsum, logtab);
        double ep[V];
#pragma unroll
        for (int v = 0; v < V; ++v) ep[v] = kval[v] ? exp_neg(digamma(gn[v], logtab) - psi_prev) : 0.0;
        if (ok) store_row<W, V>(d.epi + (size_t)p * ld, lw, ld, ep);
        // the active-set candidates of the row (k_flags_ksh clears them where the whole row has more than K / 10)
        if constexpr (W == 64) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const unsigned long long bits = __ballot(kval[v] && (gn[v] - prm.alpha >= 1.0));
            if (lw == 0) d.amask[(size_t)p * geo.kw + v] = bits;
          }
        } else {                          // one word per node: bit k = column k of the slice
          unsigned long long w = 0ull;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (kval[v] && (gn[v] - prm.alpha >= 1.0)) w |= 1ull << kidx[v];
#pragma unroll
          for (int o = 1; o < W; o <<= 1) w |= (unsigned long long)__shfl_xor((long long)w, o, 64);
          if (lw == 0 && ok) d.amask[p] = w;
        }
      }
    }
    double rs = 0.0, na = 0.0, ix = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      rs += gn[v];
      const bool act = kval[v] && (gn[v] - prm.alpha >= 1.0);
      na += act ? 1.0 : 0.0;
      ix += act ? (double)(geo.K0 + (uint32_t)kidx[v] + 1u) : 0.0;
    }
    rs = group_sum<W>(rs); na = group_sum<W>(na); ix = group_sum<W>(ix);
    if (lw == 0 && ok) { d.rowx[3 * (size_t)p] = rs; d.rowx[3 * (size_t)p + 1] = na; d.rowx[3 * (size_t)p + 2] = ix; }
  }
  if (!init) block_reduce_store<W, V, 2>(s12, d.part_b + (size_t)blockIdx.x * 2 * K, K, lds);
}

// ---------------------------------------------- node finalise, second half: from the summed rowx
// set_dir_exp (src/linksampling.hh:170-187) and prune (src/linksampling.cc:455-491) with the row sum and
// the active set of the whole row; the flags come out identical on every rank.  (W, V) as in k_fin1_ksh.
template <int W, int V>
__global__ __launch_bounds__(256) void k_fin2_ksh(Geometry geo, DeviceState d, Params prm, int init) {
  constexpr int G = 64 / W;
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  const uint32_t *__restrict__ conv_old = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  uint32_t *__restrict__ conv_new = d.conv + (size_t)(ctrl->parity ^ 1u) * geo.n_alloc;
  for (uint32_t p0 = geo.node_begin + (blockIdx.x * 4 + wave) * G; p0 < geo.node_end; p0 += gridDim.x * 4 * G) {
    const bool ok = p0 + (uint32_t)g < geo.node_end;
    const uint32_t p = ok ? p0 + (uint32_t)g : p0;
    double gn[V];
    load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gn);
    const double rs = d.rowx[3 * (size_t)p];
    const double psi_rs = digamma(rs, logtab);
    double el[V], ep[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const bool kv = (uint32_t)kmap<W, V>(lw, v) < K;
      el[v] = kv ? digamma(gn[v], logtab) - psi_rs : 0.0;
      ep[v] = kv ? exp_neg(el[v]) : 0.0;
    }
    if (ok) {
      store_row<W, V>(d.elogpi + (size_t)p * ld, lw, ld, el);
      if (d.epi) store_row<W, V>(d.epi + (size_t)p * ld, lw, ld, ep);
    }
    if (init) continue;
    const uint32_t active = (uint32_t)d.rowx[3 * (size_t)p + 1];
    const uint32_t idx = (uint32_t)d.rowx[3 * (size_t)p + 2];
    if constexpr (W == 64) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const bool act = (uint32_t)kmap<W, V>(lw, v) < K && (gn[v] - prm.alpha >= 1.0);
        const unsigned long long bits = __ballot(act);
        if (lw == 0) d.amask[(size_t)p * geo.kw + v] = (active <= geo.k10) ? bits : 0ull;
      }
    } else {                            // one word per node: bit k = column k of the slice
      unsigned long long w = 0ull;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int k = kmap<W, V>(lw, v);
        if ((uint32_t)k < K && (gn[v] - prm.alpha >= 1.0)) w |= 1ull << k;
      }
#pragma unroll
      for (int o = 1; o < W; o <<= 1) w |= (unsigned long long)__shfl_xor((long long)w, o, 64);
      if (lw == 0 && ok) d.amask[p] = (active <= geo.k10) ? w : 0ull;
    }
    if (lw == 0 && ok) {
      conv_new[p] = (active == 1u) ? idx : conv_old[p];
      d.active_cnt[p] = active;
    }
  }
}

// prune (src/linksampling.cc:455-491) from the summed rowx alone -- the product form's second half of the finalise pass
// (k_fin1_ksh left the exp(Elogpi) rows and the active-set candidates): one thread per node.
__global__ __launch_bounds__(256) void k_flags_ksh(Geometry geo, DeviceState d) {
  const DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  const uint32_t *__restrict__ conv_old = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  uint32_t *__restrict__ conv_new = d.conv + (size_t)(ctrl->parity ^ 1u) * geo.n_alloc;
  for (uint32_t p = geo.node_begin + blockIdx.x * blockDim.x + threadIdx.x; p < geo.node_end; p += gridDim.x * blockDim.x) {
    const uint32_t active = (uint32_t)d.rowx[3 * (size_t)p + 1];
    const uint32_t idx = (uint32_t)d.rowx[3 * (size_t)p + 2];
    conv_new[p] = (active == 1u) ? idx : conv_old[p];
    d.active_cnt[p] = active;
    if (active > geo.k10)
      for (uint32_t v = 0; v < geo.kw; ++v) d.amask[(size_t)p * geo.kw + v] = 0ull;
  }
}

// ---------------------------------------------------------------- s3 pass, K-sharded
// src/linksampling.cc:731-746 on the own columns.  Quirk Q2 reads column pc (one past the converged community
// pc - 1) and adds into column pc - 1: the rank that holds column pc does it; when pc is its first column the
// target belongs to the rank on its left and travels through q2v[K0 - 1].
template <int V>
__global__ __launch_bounds__(256) void k_s3_ksh(Geometry geo, DeviceState d) {
  constexpr int W = 64;
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  __shared__ double lds[V * 64];
  __shared__ double xq[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lw = lane;
  const uint32_t K = geo.K, ld = geo.ld, K0 = geo.K0;
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  const double *__restrict__ mphi = d.mphi;
  int kidx[V];
#pragma unroll
  for (int v = 0; v < V; ++v) kidx[v] = kmap<W, V>(lw, v);
  double s3[1][V];
#pragma unroll
  for (int v = 0; v < V; ++v) s3[0][v] = 0.0;
  double out = 0.0;   // wave-uniform: this rank's contribution to column K0 - 1
  for (uint32_t it = blockIdx.x * 4 + wave; it < d.nitems_s3; it += gridDim.x * 4) {
    const Item item_ = d.items_s3[d.item0_s3 + it];
    const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.node);
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.off);
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.len);
    const uint64_t base = d.rowptr[p] + off;
    const uint32_t pc = (uint32_t)__builtin_amdgcn_readfirstlane((int)conv[p]);
    double mp[V];
    load_row<W, V>(mphi + (size_t)p * ld, lw, ld, mp);
    for (uint32_t j = 0; j < len; ++j) {
      const uint32_t q = d.col[base + j];
      const uint32_t qc = conv[q];
      if ((pc != 0) != (qc != 0)) {
        const uint32_t cc = pc ? pc : qc;           // column read (global), target cc - 1
        const uint32_t other = pc ? q : p;
        if (cc < geo.Kt && cc >= K0 && cc < K0 + K) {
          const double val = mphi[(size_t)other * ld + (cc - K0)];
          if (cc == K0) out += val;
          else {
#pragma unroll
            for (int v = 0; v < V; ++v)
              if (kidx[v] == (int)(cc - K0) - 1) s3[0][v] += val;
          }
        }
      } else {
        double mq[V];
        load_row<W, V>(mphi + (size_t)q * ld, lw, ld, mq);
#pragma unroll
        for (int v = 0; v < V; ++v) s3[0][v] += mp[v] * mq[v];
      }
    }
  }
  block_reduce_store<W, V, 1>(s3, d.part_c + (size_t)blockIdx.x * K, K, lds);
  if (lane == 0) xq[wave] = out;
  __syncthreads();
  if (threadIdx.x == 0) d.part_q2[blockIdx.x] = ((xq[0] + xq[1]) + xq[2]) + xq[3];
}
// the same for slices of <= 64 columns: a row over 16 lanes x 4 doubles, four neighbours of the node per step (see k_phi_ksh16)
__global__ __launch_bounds__(256) void k_s3_ksh16(Geometry geo, DeviceState d) {
  constexpr int W = 16, V = 4;
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  __shared__ double lds[V * 64];
  __shared__ double xq[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 4, lw = lane & 15;
  const uint32_t K = geo.K, ld = geo.ld, K0 = geo.K0;
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  const double *__restrict__ mphi = d.mphi;
  int kidx[V];
#pragma unroll
  for (int v = 0; v < V; ++v) kidx[v] = kmap<W, V>(lw, v);
  double s3[1][V];
#pragma unroll
  for (int v = 0; v < V; ++v) s3[0][v] = 0.0;
  double out = 0.0;   // per group (its lanes agree): this rank's contribution to column K0 - 1
  for (uint32_t it = blockIdx.x * 4 + wave; it < d.nitems_s3; it += gridDim.x * 4) {
    const Item item_ = d.items_s3[d.item0_s3 + it];
    const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.node);
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.off);
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_.len);
    const uint64_t base = d.rowptr[p] + off;
    const uint32_t pc = (uint32_t)__builtin_amdgcn_readfirstlane((int)conv[p]);
    double mp[V];
    load_row<W, V>(mphi + (size_t)p * ld, lw, ld, mp);
    for (uint32_t j = (uint32_t)g; j < len; j += 4) {   // each group its own neighbours
      const uint32_t q = d.col[base + j];
      const uint32_t qc = conv[q];
      if ((pc != 0) != (qc != 0)) {
        const uint32_t cc = pc ? pc : qc;           // column read (global), target cc - 1
        const uint32_t other = pc ? q : p;
        if (cc < geo.Kt && cc >= K0 && cc < K0 + K) {
          const double val = mphi[(size_t)other * ld + (cc - K0)];
          if (cc == K0) out += val;
          else {
#pragma unroll
            for (int v = 0; v < V; ++v)
              if (kidx[v] == (int)(cc - K0) - 1) s3[0][v] += val;
          }
        }
      } else {
        double mq[V];
        load_row<W, V>(mphi + (size_t)q * ld, lw, ld, mq);
#pragma unroll
        for (int v = 0; v < V; ++v) s3[0][v] += mp[v] * mq[v];
      }
    }
  }
  block_reduce_store<W, V, 1>(s3, d.part_c + (size_t)blockIdx.x * K, K, lds);
  // the four groups of the wave, in group order, then the waves
  const double o1 = __shfl(out, 16, 64), o2 = __shfl(out, 32, 64), o3 = __shfl(out, 48, 64);
  if (lane == 0) xq[wave] = ((out + o1) + o2) + o3;
  __syncthreads();
  if (threadIdx.x == 0) d.part_q2[blockIdx.x] = ((xq[0] + xq[1]) + xq[2]) + xq[3];
}
// this rank's outgoing Q2 share, summed over the s3 blocks in block order, into q2v (zero elsewhere)
__global__ __launch_bounds__(256) void k_q2_ksh(Geometry geo, DeviceState d) {
  if (d.ctrl->stopped) return;
  for (uint32_t k = threadIdx.x; k < geo.Kt; k += blockDim.x) d.q2v[k] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0 && geo.K0 > 0) {
    double t = 0.0;
    for (uint32_t b = 0; b < d.nb_c; ++b) t += d.part_q2[b];
    d.q2v[geo.K0 - 1] = t;
  }
}

// ---------------------------------------------------------------- lambda of the own columns (one block)
__global__ __launch_bounds__(256) void k_lam_ksh(Geometry geo, DeviceState d, Params prm) {
  if (d.ctrl->stopped) return;
  __shared__ double2 logtab[128];
  load_logtab(logtab, d.logtab);
  __syncthreads();
  const uint32_t K = geo.K;
  for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) {
    double s1 = d.kvec_c[k], s2 = d.kvec_c[K + k];
    double s3 = d.kvec_c[2 * (size_t)K + k] + d.q2v[geo.K0 + k];   // the summed q2v: what the other ranks found for this column
    double l0 = prm.eta0 + d.kvec_a[k];
    double l1;
    if (prm.stoch) {
      // mini-batch step (lambda_of_sweep<true>): sum and s3 are window sums scaled to estimates of the full ones, s1/s2
      // running totals (previous total + this window's change), the result blended into the old lambda
      s1 += d.s12run[k];
      s2 += d.s12run[K + k];
      s3 *= prm.scale_c;
      const double h0 = prm.eta0 + d.kvec_a[k] * prm.scale_a, h1 = prm.eta1 + (s1 * s1 - s2 - s3);
      l0 = (1.0 - prm.rho_lambda) * d.lambda[2 * k] + prm.rho_lambda * h0;
      l1 = (1.0 - prm.rho_lambda) * d.lambda[2 * k + 1] + prm.rho_lambda * h1;
      d.s12run[k] = s1;
      d.s12run[K + k] = s2;
    } else {
      l1 = prm.eta1 + (s1 * s1 - s2 - s3);
    }
    d.lambda[2 * k] = l0;
    d.lambda[2 * k + 1] = l1;
    const double ps = digamma(l0 + l1, logtab);
    d.elogbeta[2 * k] = digamma(l0, logtab) - ps;
    d.elogbeta[2 * k + 1] = digamma(l1, logtab) - ps;
  }
}

// ---------------------------------------------------------------- held-out pairs: partial dot products
template <int V>
__global__ __launch_bounds__(256) void k_vdot_ksh(Geometry geo, DeviceState d) {
  constexpr int W = 64;
  if (d.ctrl->stopped) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lw = lane;
  const uint32_t K = geo.K, ld = geo.ld;
  double beta[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int k = kmap<W, V>(lw, v);
    beta[v] = 0.0;
    if ((uint32_t)k < K) {
      const double l0 = d.lambda[2 * k], l1 = d.lambda[2 * k + 1];
      beta[v] = l0 / (l0 + l1);
    }
  }
  for (uint32_t i = blockIdx.x * 4 + wave; i < d.nv; i += gridDim.x * 4) {
    const uint32_t p = d.vpairs[3 * (size_t)i], q = d.vpairs[3 * (size_t)i + 1];
    double gp[V], gq[V];
    load_row<W, V>(d.gamma + (size_t)p * ld, lw, ld, gp);
    load_row<W, V>(d.gamma + (size_t)q * ld, lw, ld, gq);
    // on normalised rows (the summed row sums are in rowx since the exchange behind the phi pass): the raw product
    // overflows once gamma reaches 1e154, which the annealing scale does on tiny graphs with thousands of communities
    const double rp = 1.0 / d.rowx[3 * (size_t)p], rq = 1.0 / d.rowx[3 * (size_t)q];
    double dot = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) dot += (gp[v] * rp) * (gq[v] * rq) * beta[v];
    dot = group_sum<W>(dot);
    if (lane == 0) d.vdot[i] = dot;
  }
}

// ---------------------------------------------------------------- likelihood row, stop rule, loop control
// validation_likelihood + the tail of the loop body (src/linksampling.cc:966-1050,763-787) from the summed
// vdot and rowx.  k_vsum_ksh: the log terms of the held-out pairs, one fixed-order partial per block;
// k_stop_ksh: one block adds the partials in block order -- the same numbers on every rank.
__global__ __launch_bounds__(256) void k_vsum_ksh(Geometry geo, DeviceState d, Params prm) {
  const DevCtrl *c = d.ctrl;
  if (c->stopped) return;
  __shared__ double red[3][256];
  double sz = 0.0, so = 0.0, kz = 0.0;
  if (d.nv > 0 && (c->iter % prm.reportfreq == 0))
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < d.nv; i += gridDim.x * blockDim.x) {
      const uint32_t y = d.vpairs[3 * (size_t)i + 2];
      const double pq = d.vdot[i];   // (k_vdot_ksh: partial dot products of the normalised rows, summed over the ranks)
      double sv = y ? pq : 1.0 - pq;
      if (sv < 1e-30) sv = 1e-30;
      const double u = log(sv);
      if (y) so += u; else { sz += u; kz += 1.0; }
    }
  red[0][threadIdx.x] = sz; red[1][threadIdx.x] = so; red[2][threadIdx.x] = kz;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int a = 0; a < 3; ++a) red[a][threadIdx.x] += red[a][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) d.tail_part[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ __launch_bounds__(256) void k_stop_ksh(Geometry geo, DeviceState d, Params prm, uint32_t nvb) {
  DevCtrl c = *d.ctrl;
  if (c.stopped) return;
  __shared__ unsigned long long cred[3][256];
  const uint32_t iter = c.iter;
  const bool do_val = d.nv > 0 && (iter % prm.reportfreq == 0);
  // the k_vsum_ksh blocks' partial sums (nvb <= 256): one per thread into LDS, requested beside the link counts -- thread 0
  // adding them straight from memory in block order was a cold miss per block, 32 us at 118 blocks
  __shared__ double pv[3][256];
  {
    const uint32_t bc = threadIdx.x < nvb ? threadIdx.x : 0u;
    const double v0 = d.tail_part[(size_t)bc * 4], v1 = d.tail_part[(size_t)bc * 4 + 1], v2 = d.tail_part[(size_t)bc * 4 + 2];
    pv[0][threadIdx.x] = v0; pv[1][threadIdx.x] = v1; pv[2][threadIdx.x] = v2;
  }
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  for (uint32_t b0 = threadIdx.x; b0 < d.nb_a; b0 += 8u * blockDim.x) {   // eight blocks' counts per thread in flight (integers: any order)
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
  cred[0][threadIdx.x] = t0; cred[1][threadIdx.x] = t1; cred[2][threadIdx.x] = t2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int a = 0; a < 3; ++a) cred[a][threadIdx.x] += cred[a][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  double szeros = 0.0, sones = 0.0, kzd = 0.0;
  for (uint32_t b = 0; b < nvb; ++b) {   // block order
    szeros += pv[0][b]; sones += pv[1][b]; kzd += pv[2][b];
  }
  c.parity ^= 1u;
  c.links_dense = cred[0][0]; c.links_sparse = cred[1][0]; c.links_shortcut = cred[2][0];
  if (d.sweep_stats) {
    unsigned long long *st = d.sweep_stats + (size_t)(c.sweeps_done % d.sweep_stats_cap) * 4;
    st[0] = c.links_dense; st[1] = c.links_sparse; st[2] = c.links_shortcut; st[3] = c.sweeps_done;
  }
  c.sweeps_done++;
  // (mini-batch steps tag on every step: a window is only visited once per pass over the nodes)
  c.write_comm = (prm.stoch || iter % prm.reportfreq == prm.reportfreq - 1) ? 1 : 0;
  bool exit_now = false;
  if (do_val) {
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
    if (iter > 10) {     // src/linksampling.cc:1008-1027
      if (a > c.prev_h && c.prev_h != 0 && fabs((a - c.prev_h) / c.prev_h) < 0.00001) { stop = true; why = 100; }
      else if (a < c.prev_h) c.nh++;
      else if (a > c.prev_h) c.nh = 0;
      if (a > c.max_h) c.max_h = a;
      if (c.nh > 2) { why = 1; stop = true; }
    }
    c.prev_h = a;
    if (c.annealing && stop) { c.annealing = 0; c.nh = 0; c.prev_h = 0; }
    else if (!c.annealing && stop && prm.use_validation_stop) exit_now = true;
    c.why = why;
  }
  if (exit_now) c.stopped = 1;
  else c.iter = iter + 1;
  store_ctrl_of_tail(d.ctrl, c);   // field by field (svils_device.hip): a whole-struct copy made this kernel read the dispatch packet
}

// ------------------------------------------------------------------ launchers
#ifndef KSH_NARROW
#define KSH_NARROW 1
#endif
#define KSH_DISPATCH(g, CALL)                   \
  do {                                          \
    if ((g).V == 1) { CALL(1); }                \
    else if ((g).V == 2) { CALL(2); }           \
    else if ((g).V == 4) { CALL(4); }           \
    else if ((g).V == 8) { CALL(8); }           \
    else if ((g).V == 12) { CALL(12); }         \
    else if ((g).V == 16) { CALL(16); }         \
    else { CALL(32); }                          \
  } while (0)

void launch_ksh_phase(const Geometry &g, const DeviceState &d, const Params &p, int phase, hipStream_t s) {
  const uint32_t nodes = g.node_end - g.node_begin;                   // every node, or the window of a mini-batch step
  const bool narrow = g.V == 1 && KSH_NARROW;                         // slices of <= 64 columns: rows over 16 lanes x 4 doubles
  const uint32_t npb = narrow ? 16u : 4u;                             // nodes per block of the node loops
  const uint32_t nbn = std::max(1u, (nodes + npb - 1) / npb > 2048 ? 2048 : (nodes + npb - 1) / npb);
  switch (phase) {
    case 0: {   // DEN
      if (g.V == 1 && KSH_NARROW && !d.ksh_log) { hipLaunchKernelGGL((k_phi_ksh16<1>), dim3(d.nb_a), dim3(256), 0, s, g, d, p); break; }
      if (d.ksh_log) {
#define CALL(V_) hipLaunchKernelGGL((k_phi_ksh<V_, 1, true>), dim3(d.nb_a), dim3(256), 0, s, g, d, p)
        KSH_DISPATCH(g, CALL);
#undef CALL
        break;
      }
#define CALL(V_) hipLaunchKernelGGL((k_phi_ksh<V_, 1, false>), dim3(d.nb_a), dim3(256), 0, s, g, d, p)
      KSH_DISPATCH(g, CALL);
#undef CALL
    } break;
    case 1: {   // PHI + sum + first half of the finalise pass
      if (d.ksh_log) {
#define CALL(V_) hipLaunchKernelGGL((k_phi_ksh<V_, 2, true>), dim3(d.nb_a), dim3(256), 0, s, g, d, p)
        KSH_DISPATCH(g, CALL);
#undef CALL
      } else if (g.V == 1 && KSH_NARROW) hipLaunchKernelGGL((k_phi_ksh16<2>), dim3(d.nb_a), dim3(256), 0, s, g, d, p);
      else {
#define CALL(V_) hipLaunchKernelGGL((k_phi_ksh<V_, 2, false>), dim3(d.nb_a), dim3(256), 0, s, g, d, p)
        KSH_DISPATCH(g, CALL);
#undef CALL
      }
      launch_reduce_a(g, d, s);
      const int fuse = d.ksh_log ? 0 : 1;   // product form: the exp(Elogpi) rows come out of this pass (see k_fin1_ksh)
#define CALL(V_)                                                                                            \
  do {                                                                                                      \
    if (p.stoch) hipLaunchKernelGGL((k_fin1_ksh<64, V_, true>), dim3(d.nb_b), dim3(256), 0, s, g, d, p, 0, fuse);  \
    else hipLaunchKernelGGL((k_fin1_ksh<64, V_, false>), dim3(d.nb_b), dim3(256), 0, s, g, d, p, 0, fuse);        \
  } while (0)
      if (narrow) {
        if (p.stoch) hipLaunchKernelGGL((k_fin1_ksh<16, 4, true>), dim3(d.nb_b), dim3(256), 0, s, g, d, p, 0, fuse);
        else hipLaunchKernelGGL((k_fin1_ksh<16, 4, false>), dim3(d.nb_b), dim3(256), 0, s, g, d, p, 0, fuse);
      } else KSH_DISPATCH(g, CALL);
#undef CALL
    } break;
    case 2: {   // second half of the finalise pass, s3
      if (!d.ksh_log) {
        const uint32_t nbf = std::max(1u, std::min(1024u, (nodes + 255u) / 256u));
        hipLaunchKernelGGL(k_flags_ksh, dim3(nbf), dim3(256), 0, s, g, d);
      } else {
#define CALL(V_) hipLaunchKernelGGL((k_fin2_ksh<64, V_>), dim3(nbn), dim3(256), 0, s, g, d, p, 0)
        if (narrow) hipLaunchKernelGGL((k_fin2_ksh<16, 4>), dim3(nbn), dim3(256), 0, s, g, d, p, 0);
        else KSH_DISPATCH(g, CALL);
#undef CALL
      }
      if (p.stoch) launch_carry_flags(g, d, s);   // rows outside the window keep their converged flag across the parity flip
      if (g.V == 1 && KSH_NARROW) hipLaunchKernelGGL(k_s3_ksh16, dim3(d.nb_c), dim3(256), 0, s, g, d);
      else {
#define CALL(V_) hipLaunchKernelGGL((k_s3_ksh<V_>), dim3(d.nb_c), dim3(256), 0, s, g, d)
        KSH_DISPATCH(g, CALL);
#undef CALL
      }
      launch_reduce_c(g, d, s);
      hipLaunchKernelGGL(k_q2_ksh, dim3(1), dim3(256), 0, s, g, d);
    } break;
    case 3: {   // lambda, partial dot products
      hipLaunchKernelGGL(k_lam_ksh, dim3(1), dim3(256), 0, s, g, d, p);
      if (d.nv) {
        const uint32_t nb = (d.nv + 3) / 4 > 1024 ? 1024 : (d.nv + 3) / 4;
#define CALL(V_) hipLaunchKernelGGL((k_vdot_ksh<V_>), dim3(nb), dim3(256), 0, s, g, d)
        KSH_DISPATCH(g, CALL);
#undef CALL
      }
    } break;
    case 4: {   // likelihood row, stop rule, loop control
      uint32_t nvb = (d.nv + 1023) / 1024;   // four pairs per thread; tail_part holds 256 block partials
      nvb = nvb < 1 ? 1 : nvb > 256 ? 256 : nvb;
      hipLaunchKernelGGL(k_vsum_ksh, dim3(nvb), dim3(256), 0, s, g, d, p);
      hipLaunchKernelGGL(k_stop_ksh, dim3(1), dim3(256), 0, s, g, d, p, nvb);
    } break;
    case 5: {   // initial state: partial row sums of the gamma just set
#define CALL(V_) hipLaunchKernelGGL((k_fin1_ksh<64, V_, false>), dim3(nbn), dim3(256), 0, s, g, d, p, 1, 0)
      if (narrow) hipLaunchKernelGGL((k_fin1_ksh<16, 4, false>), dim3(nbn), dim3(256), 0, s, g, d, p, 1, 0);
      else KSH_DISPATCH(g, CALL);
#undef CALL
    } break;
    case 6: {   // initial state: Elogpi from the summed row sums
#define CALL(V_) hipLaunchKernelGGL((k_fin2_ksh<64, V_>), dim3(nbn), dim3(256), 0, s, g, d, p, 1)
      if (narrow) hipLaunchKernelGGL((k_fin2_ksh<16, 4>), dim3(nbn), dim3(256), 0, s, g, d, p, 1);
      else KSH_DISPATCH(g, CALL);
#undef CALL
    } break;
    case 7: {   // log-domain mode: per-link max over the own columns (then MAX over the ranks, then DEN)
#define CALL(V_) hipLaunchKernelGGL((k_phi_ksh<V_, 0, true>), dim3(d.nb_a), dim3(256), 0, s, g, d, p)
      KSH_DISPATCH(g, CALL);
#undef CALL
    } break;
    case 8: {   // the partial dot products alone (svils_validation_row between two sweeps)
      if (d.nv) {
        const uint32_t nb = (d.nv + 3) / 4 > 1024 ? 1024 : (d.nv + 3) / 4;
#define CALL(V_) hipLaunchKernelGGL((k_vdot_ksh<V_>), dim3(nb), dim3(256), 0, s, g, d)
        KSH_DISPATCH(g, CALL);
#undef CALL
      }
    } break;
    default: break;
  }
}

}  // namespace svils
