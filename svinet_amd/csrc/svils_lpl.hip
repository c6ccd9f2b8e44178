// svils_lpl.hip -- lane-per-link kernels for small K (K <= 56) on gfx950.
//
// With K = 20..28 a group-per-row layout leaves 37 % of a wavefront's lanes idle
// and spends more time in cross-lane softmax reductions than in exp().  Here one
// LANE owns one directed link, loops over k in registers (no cross-lane reduction
// for the softmax at all), and the per-node sums of gammanext are formed by
// staging the 64 phi rows of a wave-item in LDS and letting lane k walk column k
// over the rows IN ENTRY ORDER (the order in which the reference's link loop adds
// to gammanext[x], src/linksampling.cc:696-701).
//
// The cost of a sweep follows the work the reference's loop does
// (src/linksampling.cc:622-681): once per sweep k_classify sorts every CSR entry
// into one of three classes -- 0: full softmax, 1: active-set softmax (_iter > 1000),
// 2: exactly one endpoint converged (O(1) shortcut) -- and stream-compacts each
// class in CSR order (one pass, decoupled look-back scan).  k_phi_lpl then runs
// only over the compacted class-0 and class-1 lists (a wave-item = 64 consecutive
// entries of ONE list, so every lane does the same amount of work), and the
// shortcut entries cost one 2-byte column id each, added per node by
// k_finalize_lpl.  Because compaction is stable, the entries of a node stay
// contiguous in every list: a node's run can straddle wave-items; interior pieces
// go straight to the node's accumulator row, the piece that ends at lane 63 of an
// item to ghead[list][node], the piece that starts at lane 0 to gtail[list][node],
// a whole item of a hub's run to slot_f[list][item] (svils_internal.h);
// k_finalize_lpl re-derives the same rule from npos[list][] and adds the pieces in
// item order.  No floating-point atomics, bit-reproducible, perfectly balanced.
#include "svils_cls.h"
#include "svils_devutil.h"

namespace svils {

// a row element to global memory, write-through when DeviceState::wt says so (uniform per launch)
__device__ __forceinline__ void row_store(double *p, double v, bool wt) {
  if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// LDS row stride in doubles: an odd number of 16-byte chunks, so the 8-lane groups
// of ds_write_b128 hit distinct slots
template <int KC>
struct LplCfg {
  static constexpr int KR = 2 * KC;
  static constexpr int SROW = 2 * (KC | 1);
};

template <int KC>
__device__ __forceinline__ void load_row_lane(const double *__restrict__ row, double (&x)[2 * KC]) {
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const double2 t = *reinterpret_cast<const double2 *>(row + 2 * c);
    x[2 * c] = t.x;
    x[2 * c + 1] = t.y;
  }
}

// stand-alone classification of the sweep about to run (first sweep, mini-batch steps, after the
// host changed flags or _iter)
__global__ __launch_bounds__(1024) void k_cls_count(Geometry geo, DeviceState d, Params prm) {
  if (d.ctrl->stopped) return;
  __shared__ ClsWork shw[4];
  cls_count_tiles<4>(geo, d, prm, shw, blockIdx.x, gridDim.x, false);
}
__global__ __launch_bounds__(1024) void k_cls_scatter(Geometry geo, DeviceState d) {
  if (d.ctrl->stopped) return;
  __shared__ ClsWork shw[4];
  cls_scatter_tiles<4>(geo, d, shw, blockIdx.x, gridDim.x);
}

// =========================== held-out likelihood + stop rule as a role (three-launch sweeps)
// validation_likelihood() of sweep v_iter (src/linksampling.cc:966-1050) for K <= 32, by `nrb` role
// blocks of any size: one group of W lanes per held-out pair (W = 8 / 16 / 32, lane = community),
// block partials published with agent-scope stores, the last block to arrive adds them in block
// order and runs the stop rule / annealing switch.  lambda and gamma of the finished sweep are in
// memory (the s3 launch's last block wrote lambda); nothing here touches what the phi blocks of
// the same launch read, except `stopped`, which they may see either way.
__device__ __forceinline__ void lpl_validation_role(const Geometry &geo, const DeviceState &d, const Params &prm,
                                                    uint32_t rb, uint32_t nrb, double (*red)[3], uint32_t *flag_lds) {
  if (d.ctrl->stopped || !d.ctrl->v_pending) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int W = geo.W, G = 64 / W;
  const int g = lane / W, lw = lane % W;
  const uint32_t K = geo.K, ld = geo.ld;
  double beta = 0.0;
  if ((uint32_t)lw < K) {
    const double l0 = d.lambda[2 * lw], l1 = d.lambda[2 * lw + 1];
    beta = l0 / (l0 + l1);  // estimate_bernoulli_rate, src/linksampling.hh:216-225
  }
  double sz = 0.0, so = 0.0, kz = 0.0;
  const uint32_t stride = nrb * nw * G;
  // two pairs per group in flight: indices first, then the rows
  for (uint32_t ib = (rb * nw + wave) * G + g; ib < d.nv; ib += 2 * stride) {
    uint32_t pp[2], qq[2], yy[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t i = ib + t * stride;
      pp[t] = 0; qq[t] = 0; yy[t] = 0;
      if (i < d.nv) { pp[t] = d.vpairs[3 * (size_t)i]; qq[t] = d.vpairs[3 * (size_t)i + 1]; yy[t] = d.vpairs[3 * (size_t)i + 2]; }
    }
    double gp[2], gq[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      gp[t] = (uint32_t)lw < K ? d.gamma[(size_t)pp[t] * ld + lw] : 0.0;
      gq[t] = (uint32_t)lw < K ? d.gamma[(size_t)qq[t] * ld + lw] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      double sp = gp[t], sq = gq[t], dot = gp[t] * gq[t] * beta;
      for (int o = 1; o < W; o <<= 1) {   // the group's lanes are contiguous: xor shuffles stay inside it
        sp += __shfl_xor(sp, o, 64);
        sq += __shfl_xor(sq, o, 64);
        dot += __shfl_xor(dot, o, 64);
      }
      if (ib + t * stride < d.nv && lw == 0) {
        // non-links: the K^2 double loop collapses exactly to (sum pi_p)(sum pi_q) - sum pi_p pi_q beta (k_tail)
        const double pq = dot / (sp * sq);
        double sv = yy[t] ? pq : 1.0 - pq;
        if (sv < 1e-30) sv = 1e-30;
        const double u = log(sv);
        if (yy[t]) so += u; else { sz += u; kz += 1.0; }
      }
    }
  }
  // block partial: lanes, then waves, in a fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sz += __shfl_xor(sz, o, 64); so += __shfl_xor(so, o, 64); kz += __shfl_xor(kz, o, 64);
  }
  if (lane == 0) { red[wave][0] = sz; red[wave][1] = so; red[wave][2] = kz; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, cc = 0.0;
    for (int w = 0; w < nw; ++w) { a += red[w][0]; b += red[w][1]; cc += red[w][2]; }
    st_agent(d.tail_part + (size_t)rb * 4, a);
    st_agent(d.tail_part + (size_t)rb * 4 + 1, b);
    st_agent(d.tail_part + (size_t)rb * 4 + 2, cc);
  }
  if (!last_block_arrives(d.tail_ctl, nrb, flag_lds)) return;
  // the blocks' partials: one lane per block (nrb <= 64), then a fixed butterfly
  double szeros = 0.0, sones = 0.0, kzd = 0.0;
  if (threadIdx.x < 64) {
    if (threadIdx.x < nrb) {
      szeros = ld_agent(d.tail_part + (size_t)threadIdx.x * 4);
      sones = ld_agent(d.tail_part + (size_t)threadIdx.x * 4 + 1);
      kzd = ld_agent(d.tail_part + (size_t)threadIdx.x * 4 + 2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      szeros += __shfl_xor(szeros, o, 64); sones += __shfl_xor(sones, o, 64); kzd += __shfl_xor(kzd, o, 64);
    }
  }
  if (threadIdx.x == 0) {
    DevCtrl *c = d.ctrl;   // field by field: a whole-struct copy goes through scratch
    const uint32_t kzeros = (uint32_t)kzd, kones = d.nv - kzeros;
    const double mean0 = szeros / kzeros, mean1 = sones / kones;
    const double a = prm.zeros_prob * mean0 + prm.ones_prob * mean1;
    const uint32_t iter = c->v_iter, nrows = c->rows;
    double *row = d.rows + (size_t)(nrows % d.rows_cap) * 10;
    row[0] = (double)iter; row[1] = (szeros + sones) / d.nv; row[2] = (double)d.nv;
    row[3] = mean0; row[4] = (double)kzeros; row[5] = mean1; row[6] = (double)kones;
    row[7] = prm.zeros_prob * mean0; row[8] = prm.ones_prob * mean1; row[9] = a;
    c->rows = nrows + 1u;
    bool stop = false;
    int why = -1;
    int nh = c->nh;
    const int annealing = c->annealing;
    if (iter > 10) {     // src/linksampling.cc:1008-1027
      const double prev = c->prev_h;
      if (a > prev && prev != 0 && fabs((a - prev) / prev) < 0.00001) { stop = true; why = 100; }
      else if (a < prev) nh++;
      else if (a > prev) nh = 0;
      if (a > c->max_h) c->max_h = a;
      if (nh > 2) { why = 1; stop = true; }
    }
    double prev_new = a;
    if (annealing && stop) {
      c->annealing = 0; nh = 0; prev_new = 0;  // max.txt keeps the pre-switch `why`
    } else if (!annealing && stop && prm.use_validation_stop) {
      c->stopped = 1;      // do_on_stop(); exit(0): _iter is not advanced (the s3 launch had advanced it)
      c->iter = iter;
    }
    c->nh = nh;
    c->prev_h = prev_new;
    c->why = why;
    c->v_pending = 0;
  }
}

// the same as its own launch: after the last sweep of a svils_sweep() call
__global__ __launch_bounds__(256) void k_validate_lpl(Geometry geo, DeviceState d, Params prm) {
  __shared__ double red[4][3];
  __shared__ uint32_t flag;
  lpl_validation_role(geo, d, prm, blockIdx.x, gridDim.x, red, &flag);
}

// ============================================================== phi pass (A6)
// PIPE (K <= 32): few fat waves (two per SIMD) walk several wave-items each, software-pipelined --
// the index pair of item i+2 and the 2*KC row chunks of item i+1 are in flight while item i runs its
// exps, its LDS staging and its column walk, so the gather latency (every row is an L2 miss: the
// finalise launch has just rewritten Elogpi) is paid once per wave, not once per item, and the time
// of the launch follows the number of items.  All 64 phi rows go through LDS in one pass.
// !PIPE (K = 33..64): the row pair no longer fits beside the phi row; one item at a time, the 64
// rows staged in two passes of 32 (half the LDS per wavefront).
// K = 25..32 (KC = 14, 16): waves per block and waves per SIMD.  Round 2 ran them as 6-wave blocks at 3 waves/SIMD (153-168
// VGPRs) and K = 28 / 32 were slower than K = 33; at 4 waves per block and 2 per SIMD like KC >= 18 the phi launch is 13-15 %
// faster (profiles/r03_small_k_latency_chain.txt: ca-AstroPh K=28 37.8 -> 32.7 us, K=32 41.9 -> 35.5, LFR K=28 14.7 -> 12.7)
#ifndef LPL_MID_NW
#define LPL_MID_NW 4
#endif
#ifndef LPL_MID_OCC
#define LPL_MID_OCC 2
#endif
// K = 21..24 (KC = 12) takes the same shape since round 5: as an 8-wave block it sat on the 128-register cap and spilled 24
// VGPRs (profiles/r06a_ab_kc12.txt: ca-AstroPh K=22 phi 35.0 -> 29.1 us, K=24 31.7 -> 28.7)
#ifndef LPL_MID_KC   // smallest KC that takes the LPL_MID_* shape
#define LPL_MID_KC 12
#endif
template <int KC, int NW, bool PIPE>
__global__ __launch_bounds__(64 * NW, (PIPE || KC >= 18 ? 2 : KC >= LPL_MID_KC ? LPL_MID_OCC : 4)) void k_phi_lpl(Geometry geo, DeviceState d, Params prm) {
  STAMP(0, 0);
  DevCtrl *ctrl = d.ctrl;
  constexpr int KR = LplCfg<KC>::KR, SROW = LplCfg<KC>::SROW;
  constexpr int NPASS = PIPE ? 1 : 2, RP = 64 / NPASS;   // staging passes, rows per pass
  __shared__ __attribute__((aligned(16))) double lds[NW][RP * SROW];
  __shared__ double red[NW][64];
  if (blockIdx.x >= d.nb_a) {
    // extra workgroups of a three-launch sweep: likelihood + stop rule of the PREVIOUS sweep, on CUs the
    // phi blocks leave idle.  The phi blocks run speculatively next to them: they accumulate into gacc0,
    // so a stop decided here leaves the state exactly as the previous sweep left it.
    __shared__ uint32_t vflag;
    lpl_validation_role(geo, d, prm, blockIdx.x - d.nb_a, gridDim.x - d.nb_a, reinterpret_cast<double (*)[3]>(&red[0][0]), &vflag);
    return;
  }
  // everything the first item needs from memory is at rest during this launch: fetch it in one go
  // (both parities of the class totals, so that nothing waits for cls_par first)
  const uint32_t stopped = ctrl->stopped, wcomm = ctrl->write_comm, cpar = ctrl->cls_par;
  const uint32_t t00 = d.ltot[0], t01 = d.ltot[1], t10 = d.ltot[8], t11 = d.ltot[9];
  // Elogbeta[.][0], wave-uniform but kept in LDS/VGPRs on purpose: as KR scalar pairs it made the
  // SGPR file spill through v_writelane/v_readlane inside the hot loop.  Padding columns
  // (k >= K) get -inf, which masks them in the softmax without any select.
  __shared__ double eb[KR];
  const uint32_t K = geo.K, ld = geo.ld;
  double ebv = NEG_INF;
  if (threadIdx.x < K) ebv = d.elogbeta[2 * threadIdx.x];
  if (stopped) return;
  STAMP(0, 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool write_comm = wcomm != 0;
  const uint32_t tot0 = cpar ? t10 : t00, tot1 = cpar ? t11 : t01;
  const uint32_t n0 = (tot0 + 63u) >> 6, n1 = (tot1 + 63u) >> 6, nit = n0 + n1;
  const double *__restrict__ elogpi = d.elogpi;
  double *mylds = lds[wave];
  const uint32_t step = d.nb_a * NW;
  uint32_t it = blockIdx.x * NW + wave;

  // index pair of this lane's entry of wave-item i (p = ~0: none)
  auto fetch_idx = [&](uint32_t i, uint32_t &pp, uint32_t &qq) {
    pp = 0xffffffffu; qq = 0;
    if (i < nit) {
      const uint32_t l = i >= n0 ? 1u : 0u;
      const uint32_t g = (l ? i - n0 : i) * 64u + lane;
      if (g < (l ? tot1 : tot0)) { pp = d.cp[l][g]; qq = d.cq[l][g]; }
    }
  };
  auto fetch_rows = [&](uint32_t pp, uint32_t qq, double2 (&ra)[PIPE ? KC : 1], double2 (&rb)[PIPE ? KC : 1]) {
    if (PIPE && pp != 0xffffffffu) {
      const double *rp = elogpi + (size_t)pp * ld, *rq = elogpi + (size_t)qq * ld;
#pragma unroll
      for (int c = 0; c < (PIPE ? KC : 1); ++c) {
        ra[c] = *reinterpret_cast<const double2 *>(rp + 2 * c);
        rb[c] = *reinterpret_cast<const double2 *>(rq + 2 * c);
      }
    }
  };
  uint32_t p, q, pn = 0xffffffffu, qn = 0;
  double2 ra[PIPE ? KC : 1], rb[PIPE ? KC : 1];
  fetch_idx(it, p, q);
  if constexpr (PIPE) {
    fetch_rows(p, q, ra, rb);
    fetch_idx(it + step, pn, qn);
  }
  if (threadIdx.x < KR) eb[threadIdx.x] = ebv;
  __syncthreads();
  double csum = 0.0;  // lane k: partial of sum[k]
  STAMP(0, 2);

#pragma unroll 1
  for (; it < nit; it += step) {
    const uint32_t list = it >= n0 ? 1u : 0u;            // wave-uniform
    const uint32_t w = list ? it - n0 : it;
    const bool valid = p != 0xffffffffu;
    double phi[KR];
    double *mine = mylds + (lane % RP) * SROW;   // this lane's staged phi row
    int tagk = -1;   // community this link tags (src/linksampling.cc:668-681,704-717), -1: none
    bool dense_row = false;
    // active-set path (:634-681): a list of its own, so the branch is wave-uniform.  Only the columns of the union of
    // the two active sets exist for such a link -- at most NU = 2 * (K / 10) of them (4 of 20 at K = 20) -- and only
    // those are fetched (8 bytes each from the two rows), exponentiated and written into the otherwise zero staged row.
    constexpr int NU = 2 * (KR / 10) > 0 ? 2 * (KR / 10) : 1;
    if (list) {
      double su[NU];
      int ku[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) { su[u] = NEG_INF; ku[u] = -1; }
      if (valid) {
        unsigned long long um = d.amask[p] | d.amask[q];   // kw == 1 for K <= 64; <= 2 * k10 bits
        const double *rp = elogpi + (size_t)p * ld, *rq = elogpi + (size_t)q * ld;
        double xa[NU], xb[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          xa[u] = 0.0; xb[u] = 0.0;
          if (um) {
            const int k = __builtin_ctzll(um);
            um &= um - 1ull;
            ku[u] = k;
            xa[u] = rp[k];
            xb[u] = rq[k];
          }
        }
        double m = NEG_INF;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (ku[u] >= 0) su[u] = (xa[u] + xb[u]) + eb[ku[u]];   // the reference's order (:686)
          m = max_f64(m, su[u]);
        }
        if (m != NEG_INF) {
          dense_row = true;
          int best = 0;
#pragma unroll
          for (int u = NU - 1; u >= 0; --u) best = (su[u] == m) ? ku[u] : best;   // first strict maximum (ascending k)
          double ssum = 0.0;
#pragma unroll
          for (int u = 0; u < NU; ++u) su[u] -= m;
          if constexpr (NU % 2 == 0) {
#pragma unroll
            for (int u0 = 0; u0 < NU; u0 += 2) {
              double t[2] = {su[u0], su[u0 + 1]};
              exp_neg_n<2>(t);   // exp_neg(-inf) == 0 for the unused slots
              su[u0] = t[0]; su[u0 + 1] = t[1];
            }
          } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) su[u] = exp_neg(su[u]);
          }
#pragma unroll
          for (int u = 0; u < NU; ++u) ssum += su[u];
          const double inv = fast_rcp(ssum);
#pragma unroll
          for (int u = 0; u < NU; ++u) su[u] *= inv;
          if (write_comm && inv > prm.link_thresh) tagk = best;
        }
      }
      // kept in the phi registers until the row is staged (values, then their column ids as bit patterns): the two
      // paths share one set of live registers
      static_assert(KR >= 2 * NU, "phi row too short for the packed active-set form");
#pragma unroll
      for (int u = 0; u < NU; ++u) { phi[u] = su[u]; phi[NU + u] = __hiloint2double(0, ku[u]); }
    } else if (valid) {
      // x_k = (Elogpi[p][k] + Elogpi[q][k]) + Elogbeta[k][0], the reference's order (:686)
      if constexpr (PIPE) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          phi[2 * c] = (ra[c].x + rb[c].x) + eb[2 * c];
          phi[2 * c + 1] = (ra[c].y + rb[c].y) + eb[2 * c + 1];
        }
      } else {
        const double *rp = elogpi + (size_t)p * ld, *rq = elogpi + (size_t)q * ld;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const double2 a = *reinterpret_cast<const double2 *>(rp + 2 * c);
          const double2 b = *reinterpret_cast<const double2 *>(rq + 2 * c);
          phi[2 * c] = (a.x + b.x) + eb[2 * c];
          phi[2 * c + 1] = (a.y + b.y) + eb[2 * c + 1];
        }
      }
    }
    if constexpr (PIPE) {
      // the row registers are free again: rows of the next item, index pair of the one after it
      fetch_rows(pn, qn, ra, rb);
    }
    uint32_t pnn = 0xffffffffu, qnn = 0;
    if constexpr (PIPE) fetch_idx(it + 2 * step, pnn, qnn);
    if (valid && !list) {
      dense_row = true;
      // branch-free from here so the KR independent exp chains interleave
      double m = NEG_INF;
#pragma unroll
      for (int k = 0; k < KR; ++k) m = max_f64(m, phi[k]);   // padding columns are -inf via eb[]
      if (m != NEG_INF) {
        STAMP(0, 5);
        int best = 0;
#pragma unroll
        for (int k = KR - 1; k >= 0; --k) best = (phi[k] == m) ? k : best;   // first strict maximum
        double s = 0.0;
        // KR is even: exps in interleaved groups of 4 (or 2 for the tail)
#pragma unroll
        for (int k0 = 0; k0 + 4 <= KR; k0 += 4) {
          double t[4] = {phi[k0] - m, phi[k0 + 1] - m, phi[k0 + 2] - m, phi[k0 + 3] - m};
          exp_neg_n<4>(t);   // exp_neg(-inf) == 0 for masked / padding columns
#pragma unroll
          for (int j = 0; j < 4; ++j) { phi[k0 + j] = t[j]; s += t[j]; }
        }
        if constexpr (KR % 4 != 0) {
          double t[2] = {phi[KR - 2] - m, phi[KR - 1] - m};
          exp_neg_n<2>(t);
          phi[KR - 2] = t[0]; phi[KR - 1] = t[1];
          s += t[0]; s += t[1];
        }
        const double inv = fast_rcp(s);
#pragma unroll
        for (int k = 0; k < KR; ++k) phi[k] *= inv;
        // community tagging: the first strict maximum of phi is 1/s
        if (write_comm && inv > prm.link_thresh) tagk = best;
        STAMP(0, 6);
      } else {
        dense_row = false;  // cannot happen on the dense path (every x_k is finite); kept as the guard it was
      }
    }
    // Stage the phi rows in LDS and let lane k sum column k over them in entry order, flushing at
    // node boundaries.
    // heads of the node runs: bit r set <=> row r starts a new node
    const uint32_t pprev = __shfl_up((int)p, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || p != pprev);
    const unsigned long long vmask = __ballot(p != 0xffffffffu);
    const bool any_tag = __ballot(tagk >= 0) != 0ull;
    // lane k: bit r set <=> row r tags community k (KR ballots instead of 64 x 3 VALU ops in the row loop)
    unsigned long long tmask = 0ull;
    if (any_tag) {
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const unsigned long long mk = __ballot(tagk == k);
        tmask = (lane == k) ? mk : tmask;
      }
    }
    // where a piece goes by the lanes it covers (svils_internal.h): whole item -> slot of the item, else by node
    double *const slotf = d.slot_f + ((size_t)list * d.lpl_nitems + w) * ld;
    double *const head = d.ghead + (size_t)list * geo.n_alloc * ld;
    double *const tail = d.gtail + (size_t)list * geo.n_alloc * ld;
    double *const direct = list ? d.gacc1 : d.gacc;
    double acc = 0.0;
    int a = 0;
    // one run [a, b] of node `cur` is complete: store its partial gammanext row and its tags.
    // Tags are pre-reduced per run so that a node costs one atomic per wave-item, not one per link.
#define LPL_FLUSH(DST, B)                                                                   \
    do {                                                                                    \
      if ((vmask >> a) & 1ull) {                                                            \
        const uint32_t cur = __builtin_amdgcn_readlane(p, a);                               \
        double *dst = (DST);                                                                \
        row_store(&dst[lane], acc, d.wt != 0);                                              \
        csum += acc;                                                                        \
        if (any_tag) {                                                                      \
          const unsigned long long runmask = (((B) >= 63) ? ~0ull : ((2ull << (B)) - 1ull)) & ~((1ull << a) - 1ull); \
          const uint32_t cnt = (uint32_t)__popcll(tmask & runmask);                         \
          if (d.fcnt) { if (cnt) atomicAdd(&d.fcnt[(size_t)cur * ld + lane], cnt); }        \
          else {                                                                            \
            const unsigned long long bits = __ballot(cnt > 0);                              \
            if (bits && lane == 0) atomicOr(&d.member_acc[cur], bits);                      \
          }                                                                                 \
        }                                                                                   \
      }                                                                                     \
      acc = 0.0;                                                                            \
    } while (0)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      if (NPASS == 1 || (lane >> 5) == pass) {
        if (dense_row && !list) {
#pragma unroll
          for (int c = 0; c < KC; ++c) *reinterpret_cast<double2 *>(mine + 2 * c) = make_double2(phi[2 * c], phi[2 * c + 1]);
        } else {
#pragma unroll
          for (int c = 0; c < KC; ++c) *reinterpret_cast<double2 *>(mine + 2 * c) = make_double2(0.0, 0.0);
          if (list && dense_row) {   // the union's columns into the zero row (same lane, LDS in order)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
              const int k = __double2loint(phi[NU + u]);
              if (k >= 0) mine[k] = phi[u];
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if ((uint32_t)lane < K) {
#pragma unroll 1
        for (int rb0 = 0; rb0 < RP; rb0 += 16) {
          double v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = mylds[(rb0 + j) * SROW + lane];
          // a chunk without a run boundary (about half of them at an average degree of 22) is 16 plain adds
          uint32_t hb = (uint32_t)(heads >> (pass * RP + rb0)) & 0xffffu;
          if (pass * RP + rb0 == 0) hb &= ~1u;
          if (hb == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int r = pass * RP + rb0 + j;
              if (r > 0 && ((heads >> r) & 1ull)) {      // run [a, r-1] ends (wave-uniform branch)
                LPL_FLUSH(((a == 0) ? tail : direct) + (size_t)cur * ld, r - 1);
                a = r;
              }
              acc += v[j];
            }
          }
        }
        // last run ends at lane 63
        if (pass == NPASS - 1) LPL_FLUSH((a == 0) ? slotf : head + (size_t)cur * ld, 63);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#undef LPL_FLUSH
    STAMP(0, 7);
    if constexpr (PIPE) { p = pn; q = qn; pn = pnn; qn = qnn; }
    else fetch_idx(it + step, p, q);
  }

  // per-block partial of `sum` (src/linksampling.cc:625,630,663,700 summed per node): waves in
  // order; block 0 adds the shortcut entries' share (+1 per directed entry at its column, :625,:630)
  STAMP(0, 3);
  red[wave][lane] = csum;
  __syncthreads();
  STAMP(0, 4);
  if (threadIdx.x < K) {
    double t = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w][threadIdx.x];
    if (d.fold) {
      // whole sweeps driven by the library: one fixed-point integer atomic per column into this XCD's accumulator
      // (k_finalize_lpl adds the eight of them and the shortcut histogram)
      const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID, bits [3:0]
      long long *w = &d.sumfx[(((size_t)cpar * 8 + xcc) * 2) * 64 + threadIdx.x];   // [parity][xcc][hi | lo][64]
      fx_add(w, w + 64, t, d.fx_scale);
    } else {
      if (blockIdx.x == 0) t += (double)d.shist[(size_t)cpar * K + threadIdx.x];
      d.part_a[(size_t)blockIdx.x * K + threadIdx.x] = t;
    }
  }
}

// ======================================= node finalise (A7 + swap + A5 + A9), K <= 56
// compute_mean_indicators (src/linksampling.cc:526-545), the gamma swap/reset (:751-755),
// set_dir_exp (src/linksampling.hh:170-187) and prune (:455-491).  gammanext[p] = the pieces k_phi_lpl left
// for the node in both class lists (item order) + 1.0 per shortcut entry at its column.
//
// One group of FW lanes per owned node, every lane NC communities (lw, lw + FW, ...; FW * NC >= K), so a
// wavefront finalises 64 / FW nodes at once -- EIGHT for K <= 32 (FW = 8, NC = ceil(K / 8): 3 at K = 20, 4 at
// K = 28).  Round 2 gave a node 32 lanes with one community each (12 of 32 idle at K = 20): on ca-AstroPh 17 903
// nodes met 12 288 resident groups, so 46 % of the groups ran a second node behind the first one's chain of
// dependent misses (index words -> pieces -> stores).  With eight nodes per wavefront the whole launch is ONE
// resident round of 12-wave blocks, one per CU (so every block also folds `sum` once, not twice per CU), 20 of
// 24 lane slots work at K = 20, and the NC digamma chains of a lane interleave.
// The launch is a chain of cold misses (everything it reads was written by the launch before, on another XCD),
// so accesses that do not depend on each other are issued together: the first node's index words go out before
// anything else, its pieces before the fold of `sum` is consumed.
template <int W>
__device__ __forceinline__ unsigned long long group_mask() { return W == 64 ? ~0ull : ((1ull << (W & 63)) - 1ull); }

// blocks at three waves per SIMD (<= 168 VGPRs); four communities per lane (K = 25..32, 49..56) need a few more
// than that: 8-wave blocks at two per SIMD
// ... which is one block per CU of 64 nodes (K <= 32): a graph of more than 64 x CUs nodes then needs a second round of
// blocks.  The four-community variants therefore ALSO exist as 12-wave blocks (96 nodes; 9 VGPRs spilled): ca-AstroPh
// (17 903 nodes) K=28 finalise 22.0 -> 18.3 us, K=32 21.7 -> 18.2, K=56 25.6 -> 23.9; on a small graph the spill costs
// (LFR K=28 13.5 -> 15.3) -- lpl_finalize_waves picks by the node count (profiles/r06e_ab_fin768.txt).
// Up to three communities per lane (K <= 24, 33..48): 9-wave blocks (72 nodes at K <= 32): ca-AstroPh's 17 903 nodes then make 249
// blocks, one per CU in one round on 249 of the 256 CUs; as 12-wave blocks (96 nodes) they made 187 -- the same round on fewer
// CUs with more waves each (K=20 51.0 -> 50.5 us per sweep, profiles/r06h_ab_fin3.txt; 10 waves: 50.8; LFR unchanged)
#ifndef LPL_FIN3_THREADS
#define LPL_FIN3_THREADS 576
#endif
constexpr int fin_threads(int nc) { return nc >= 4 ? 512 : LPL_FIN3_THREADS; }

struct FinIdx {
  uint32_t r[3][2];   // npos[l][p], npos[l][p + 1]
  uint64_t rp0, rp1;  // rowptr[p], rowptr[p + 1]
};
__device__ __forceinline__ FinIdx fin_load_idx(const DeviceState &d, uint32_t p, bool ok) {
  FinIdx x;
#pragma unroll
  for (int l = 0; l < 3; ++l) { x.r[l][0] = ok ? d.npos[l][p] : 0u; x.r[l][1] = ok ? d.npos[l][p + 1] : 0u; }
  x.rp0 = ok ? d.rowptr[p] : 0ull;
  x.rp1 = ok ? d.rowptr[p + 1] : 0ull;
  return x;
}

// LIGHT: the node-block form (see k_finalize in svils_device.hip): mean indicators, s1 / s2, tags and the unscaled row into
// the exchange staging; scale, Elogpi and prune() follow in k_expand_all behind the exchange.
template <int FW, int NC, bool STOCH, bool LIGHT = false, int NTH = fin_threads(NC)>
__global__ __launch_bounds__(NTH, (NTH >= 768 ? 3 : 2)) void k_finalize_lpl(Geometry geo, DeviceState d, Params prm) {
  STAMP(1, 0);
  DevCtrl *ctrl = d.ctrl;
  constexpr int G = 64 / FW;
  __shared__ double2 logtab[128];
  __shared__ double ksum[64];
  constexpr int FIN_WAVES = NTH / 64;
  __shared__ double s12l[FIN_WAVES][FW][2 * NC];
  __shared__ uint32_t shh[FIN_WAVES][G * FW * NC];   // per-group histogram of shortcut columns
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int g = lane / FW, lw = lane % FW;
  const uint32_t K = geo.K, ld = geo.ld;
  const uint32_t nown = geo.node_end - geo.node_begin;
  const uint32_t stride = gridDim.x * nw * G;
  uint32_t i = (blockIdx.x * nw + wave) * G + g;
  // nothing below depends on the control block until `scale`: the first node's index words go first
  FinIdx ix = fin_load_idx(d, geo.node_begin + (i < nown ? i : 0u), i < nown);
  const uint32_t stopped = ctrl->stopped, c_ann = (uint32_t)ctrl->annealing, c_wc = ctrl->write_comm, c_par = ctrl->parity;
  // `sum` of a whole sweep (fold): the eight per-XCD fixed-point accumulators + the shortcut histogram, both halves
  // requested now (the half in use is known once the control block has landed)
  // (summed per half right away: four live words instead of 32 until the control block says which half it is)
  long long ah0 = 0, al0 = 0, ah1 = 0, al1 = 0;
  unsigned long long sh0 = 0ull, sh1 = 0ull;
  if (d.fold && threadIdx.x < 64) {
#pragma unroll
    for (int x = 0; x < 8; ++x) {   // [parity][xcc][hi | lo][64]
      ah0 += d.sumfx[(size_t)(2 * x) * 64 + threadIdx.x];
      al0 += d.sumfx[(size_t)(2 * x + 1) * 64 + threadIdx.x];
      ah1 += d.sumfx[(size_t)(16 + 2 * x) * 64 + threadIdx.x];
      al1 += d.sumfx[(size_t)(16 + 2 * x + 1) * 64 + threadIdx.x];
    }
    if (threadIdx.x < K) { sh0 = d.shist[threadIdx.x]; sh1 = d.shist[(size_t)K + threadIdx.x]; }
  }
  const uint32_t c_cpar = ctrl->cls_par, c_epoch = ctrl->sweeps_done + 1u;
  load_logtab(logtab, d.logtab);
  if (stopped) return;
  STAMP(1, 1);
  const bool annealing = !LIGHT && c_ann != 0;
  const bool write_comm = c_wc != 0;
  const uint32_t *__restrict__ conv_old = d.conv + (size_t)c_par * geo.n_alloc;
  uint32_t *__restrict__ conv_new = d.conv + (size_t)(c_par ^ 1u) * geo.n_alloc;
  // three-launch sweeps: the s3 launch classifies the next sweep's links AND advances the control block,
  // so what the classification works from is recorded here, where the control block is at rest
  if (d.fused3 && blockIdx.x == 0 && threadIdx.x == 0) cls_record_args(d, prm, true);
  // `sum`: folded from the phi pass's per-block partial rows (block 0 also publishes it), or the
  // reduced / all-reduced vector when the caller splits the sweep at its exchange points
  bool kv[NC];
#define ST(j) (kv[j] && ok)
#pragma unroll
  for (int j = 0; j < NC; ++j) kv[j] = (uint32_t)(lw + j * FW) < K;
  double s1[NC], s2[NC], scale[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) { s1[j] = 0.0; s2[j] = 0.0; scale[j] = 1.0; }

  // The first round is taken by every lane (a group without a node runs it predicated off: the block's barriers
  // and the fold need all threads); further rounds only exist on graphs of more nodes than resident groups.
  for (bool first = true;; i += stride) {
    const bool ok = i < nown;
    if (!first && !__any(ok)) break;
    const uint32_t p = geo.node_begin + (ok ? i : 0u);
    const double tl = 2.0 * (double)(ix.rp1 - ix.rp0);  // quirk Q3
    const size_t rowoff = (size_t)p * ld + lw;
    // The by-node pieces of list 0 (interior, head, tail: svils_internal.h) sit at addresses that depend on the node alone:
    // requested before the run boundaries are looked at, they travel with the index words instead of behind them.
    double sp_d[NC], sp_h[NC], sp_t[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const bool on = kv[j] && ok;
      sp_d[j] = on ? d.gacc[rowoff + j * FW] : 0.0;
      sp_h[j] = on ? d.ghead[rowoff + j * FW] : 0.0;
      sp_t[j] = on ? d.gtail[rowoff + j * FW] : 0.0;
    }
    double acc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[j] = 0.0;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const uint32_t r0 = ix.r[l][0], r1 = ix.r[l][1];
      if (r1 > r0) {
        const double *sf = d.slot_f + (size_t)l * d.lpl_nitems * ld;
        const double *hd = d.ghead + (size_t)l * geo.n_alloc * ld, *tp = d.gtail + (size_t)l * geo.n_alloc * ld;
        const double *direct = l ? d.gacc1 : d.gacc;
        const uint32_t w0 = r0 >> 6, w1 = (r1 - 1u) >> 6;
        const bool a0 = (r0 & 63u) == 0u, b63 = ((r1 - 1u) & 63u) == 63u;   // the run starts at lane 0 / ends at lane 63
        if (w0 == w1) {
          if (l == 0 && !(a0 && b63)) {
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[j] += a0 ? sp_t[j] : b63 ? sp_h[j] : sp_d[j];
          } else {
            const double *src = (a0 && b63) ? sf + (size_t)w0 * ld
                                : a0 ? tp + (size_t)p * ld
                                : b63 ? hd + (size_t)p * ld
                                      : direct + (size_t)p * ld;
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[j] += kv[j] ? src[lw + j * FW] : 0.0;
          }
        } else {
          // the run spans items: its first piece (head, or a whole item), whole items of a hub in between, its last
          // piece (tail, or a whole item) -- added in item order
          double fp[NC], lp[NC];
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            fp[j] = a0 ? (kv[j] ? sf[(size_t)w0 * ld + lw + j * FW] : 0.0)
                       : (l == 0 ? sp_h[j] : (kv[j] ? hd[rowoff + j * FW] : 0.0));
            lp[j] = b63 ? (kv[j] ? sf[(size_t)w1 * ld + lw + j * FW] : 0.0)
                        : (l == 0 ? sp_t[j] : (kv[j] ? tp[rowoff + j * FW] : 0.0));
          }
#pragma unroll
          for (int j = 0; j < NC; ++j) acc[j] += fp[j];
          constexpr uint32_t HP = NC >= 4 ? 1 : 2;   // whole-item pieces in flight at a time
#pragma unroll 1
          for (uint32_t wb = w0 + 1u; wb < w1; wb += HP) {
            double pv[HP][NC];
#pragma unroll
            for (uint32_t t = 0; t < HP; ++t) {
              const uint32_t w = wb + t;
#pragma unroll
              for (int j = 0; j < NC; ++j) pv[t][j] = (w < w1 && kv[j]) ? sf[(size_t)w * ld + lw + j * FW] : 0.0;
            }
#pragma unroll
            for (uint32_t t = 0; t < HP; ++t)
#pragma unroll
              for (int j = 0; j < NC; ++j) acc[j] += pv[t][j];
          }
#pragma unroll
          for (int j = 0; j < NC; ++j) acc[j] += lp[j];
        }
      }
    }
    {   // exactly-one-converged links: +1 at the converged community (:622-631)
      const uint32_t e0 = ix.r[2][0], e1 = ix.r[2][1];
      if (__any(e1 > e0)) {
        // FW entries at a time per group, counted with integer LDS atomics (order-free)
        uint32_t *hh = &shh[wave][g * FW * NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) hh[lw + j * FW] = 0;
        for (uint32_t jb = e0 + (uint32_t)lw; jb < e1; jb += 4 * FW) {   // four loads in flight per lane
          uint32_t cv[4];
#pragma unroll
          for (uint32_t t = 0; t < 4; ++t) cv[t] = (jb + t * FW < e1) ? (uint32_t)d.scol[jb + t * FW] : 0xffffu;
#pragma unroll
          for (uint32_t t = 0; t < 4; ++t)
            if (cv[t] != 0xffffu) atomicAdd(&hh[cv[t] % (FW * NC)], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[j] += (double)hh[lw + j * FW];
      }
    }
    const uint32_t cf_old = (!LIGHT && ok && lw == 0) ? (uint32_t)d.cflag[p] : 0u;
    // graphs of more nodes than resident groups: the NEXT node's index words travel while this one is finalised (a
    // round is a chain of dependent misses -- index words -> pieces -> stores -- and at n = 1e6 a wave runs ~40 of them)
    const bool ok_next = (uint64_t)i + stride < nown;
    const FinIdx ix_next = fin_load_idx(d, geo.node_begin + (ok_next ? i + stride : 0u), ok_next);
    unsigned long long memb = 0ull;
    uint32_t fc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) fc[j] = 0;
    if (write_comm && ok) {
      if (d.fcnt) {
#pragma unroll
        for (int j = 0; j < NC; ++j)
          if (kv[j]) fc[j] = d.fcnt[rowoff + j * FW];
      } else {
        memb = d.member_acc[p];
      }
    }
    double sold[STOCH ? NC : 1], gold[STOCH ? NC : 1];
    uint32_t ncnt = 0;
    if constexpr (STOCH) {
#pragma unroll
      for (int j = 0; j < NC; ++j) { sold[j] = 0.0; gold[j] = 0.0; }
      if (tl > 0.0 && ok) {
        ncnt = d.ncnt[p];
#pragma unroll
        for (int j = 0; j < NC; ++j)
          if (kv[j]) { gold[j] = d.gamma[rowoff + j * FW]; sold[j] = d.mphi[rowoff + j * FW]; }
      }
    }
    if (first) {
      // everything the first node needs is in flight; now `sum`
      if (LIGHT && d.fold && blockIdx.x == 0 && threadIdx.x < K) {
        // node-block sweeps: this rank's share of `sum`, published for the all-reduce that follows the launch
        d.kvec_a[threadIdx.x] = (c_cpar & 1u) ? fx_value(ah1, al1, d.fx_inv) + (double)sh1 : fx_value(ah0, al0, d.fx_inv) + (double)sh0;
      }
      if (!LIGHT && threadIdx.x < 64) {
        double t = 1.0;
        if (d.fold) {
          t = (c_cpar & 1u) ? fx_value(ah1, al1, d.fx_inv) + (double)sh1 : fx_value(ah0, al0, d.fx_inv) + (double)sh0;
          if (blockIdx.x == 0 && threadIdx.x < K) d.kvec_a[threadIdx.x] = t;
        } else if (threadIdx.x < K) {
          t = d.kvec_a[threadIdx.x];
        }
        ksum[threadIdx.x] = threadIdx.x < K ? t : 1.0;
        // 1 / scale of this sweep, for whoever derives the mean indicators from the gamma rows written below
        if (blockIdx.x == 0 && threadIdx.x < K) d.iscale[threadIdx.x] = annealing ? (STOCH ? t * prm.scale_a : t) / (double)prm.ones : 1.0;
      }
      __syncthreads();   // ksum, the log table
      STAMP(1, 2);
      // _network.ones() / _sum[k], src/linksampling.cc:542
      // (mini-batch step: the window's sum, scaled to an estimate of the full one)
      if (annealing) {
#pragma unroll
        for (int j = 0; j < NC; ++j)
          if (kv[j]) scale[j] = (double)prm.ones / (STOCH ? ksum[lw + j * FW] * prm.scale_a : ksum[lw + j * FW]);
      }
      first = false;
    }
    if (write_comm) {
      unsigned long long b = memb;
      if (d.fcnt) {
        b = 0ull;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          if (kv[j] && ok) d.fcnt[rowoff + j * FW] = 0;
          b |= ((__ballot(kv[j] && fc[j] > prm.lt_min_deg) >> (g * FW)) & group_mask<FW>()) << (j * FW);
        }
      }
      if (lw == 0 && ok) {
        d.member[(size_t)p * geo.kw] = b;
        if (!d.fcnt) d.member_acc[p] = 0ull;
      }
    }
    double gn[NC], m[NC];
    if (tl > 0.0) {
      const double nl = (double)geo.n - tl - 1.0;
      double rho = 0.0;
      if constexpr (STOCH) {
        // Robbins-Monro step of this node: gamma <- (1 - rho) gamma + rho gamma_hat with
        // rho = (tau0 + c)^-kappa, c = updates the node has had; s1/s2 are kept as running sums
        // over the stored mphi rows, so this row contributes (new - old)
        rho = exp_neg(-prm.kappa * log_tab(prm.tau0 + (double)ncnt, logtab));
        if (lw == 0 && ok) d.ncnt[p] = ncnt + 1u;
      }
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const double g0 = prm.alpha + acc[j];
        m[j] = (g0 - prm.alpha) / tl;
        gn[j] = g0 + nl * m[j];
        if (annealing) gn[j] *= scale[j];
        if (!kv[j]) { m[j] = 0.0; gn[j] = 0.0; }
        if (kv[j] && ok) { s1[j] += m[j]; s2[j] += m[j] * m[j]; }
        if constexpr (STOCH) {
          if (kv[j]) gn[j] = (1.0 - rho) * gold[j] + rho * gn[j];
          if (kv[j] && ok) { s1[j] -= sold[j]; s2[j] -= sold[j] * sold[j]; }
        }
        if ((STOCH || !d.derive_m) && ST(j)) row_store(&d.mphi[rowoff + j * FW], m[j], d.wt != 0);
      }
    } else {
      // no training link: gammanext stays alpha, mphi row stays stale (:532-533)
#pragma unroll
      for (int j = 0; j < NC; ++j) { gn[j] = kv[j] ? prm.alpha : 0.0; m[j] = 0.0; }
    }
    if constexpr (LIGHT) {
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (ST(j)) d.gown[(size_t)i * ld + lw + j * FW] = gn[j];
      ix = ix_next;
      continue;
    }
    double rsl = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (ST(j)) row_store(&d.gamma[rowoff + j * FW], gn[j], d.wt != 0);   // padding columns stay 0
      rsl += gn[j];
    }
    const double rs = group_sum<FW>(rsl);
    // A slot above K is idle when K < FW * NC: lane FW - 1 evaluates psi(row sum) in its last slot, in the same
    // digamma call as the last community of the other lanes; a full row (K == FW * NC) takes one more call
    double ps[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      double arg = kv[j] ? gn[j] : 1.0;
      if (j == NC - 1 && lw == FW - 1 && K < (uint32_t)(FW * NC)) arg = rs;
      ps[j] = digamma(arg, logtab);
    }
    double psi_rs;
    if (K < (uint32_t)(FW * NC)) psi_rs = __shfl(ps[NC - 1], g * FW + FW - 1, 64);
    else psi_rs = digamma(rs, logtab);
    unsigned long long bits = 0ull;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (ST(j)) row_store(&d.elogpi[rowoff + j * FW], kv[j] ? ps[j] - psi_rs : 0.0, d.wt != 0);
      // prune / check_and_set_converged, src/linksampling.cc:455-475
      bits |= ((__ballot(kv[j] && (gn[j] - prm.alpha >= 1.0)) >> (g * FW)) & group_mask<FW>()) << (j * FW);
    }
    const uint32_t active = (uint32_t)__popcll(bits);
    if (lw == 0 && ok) {
      const uint32_t cnew = (active == 1) ? (uint32_t)(63 - __builtin_clzll(bits)) + 1u : conv_old[p];
      const unsigned long long am = (active <= geo.k10) ? bits : 0ull;
      conv_new[p] = cnew;
      d.active_cnt[p] = active;
      const uint32_t cf_new = cflag_pack(cnew, active < geo.k10);
      d.cflag[p] = (uint8_t)cf_new;
      // a classification-relevant word changed: the link classes of the next sweep have to be rebuilt (every writer
      // stores the same epoch; nobody reads it during this launch)
      if (cf_new != cf_old) d.cls_epoch[0] = c_epoch;
      d.amask[(size_t)p * geo.kw] = am;
      if (nown < geo.n) {   // node-block handles only: the same flags, packed for the exchange (nobody reads them otherwise)
        uint32_t *xf = d.xflags + (size_t)p * d.xf_ld;
        xf[0] = cnew; xf[1] = active; xf[2] = (uint32_t)am; xf[3] = (uint32_t)(am >> 32);
      }
    }
    ix = ix_next;
  }
  STAMP(1, 3);
  // per-block partials of s1, s2: the groups of a wave (butterfly over the group index), then the waves in order
#pragma unroll
  for (int j = 0; j < NC; ++j) {
#pragma unroll
    for (int o = FW; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    if (lane < FW) { s12l[wave][lane][j] = s1[j]; s12l[wave][lane][NC + j] = s2[j]; }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const uint32_t which = threadIdx.x >> 6, k = threadIdx.x & 63u;
    if (k < K) {
      const int kl = (int)k % FW, ks = (int)k / FW;
      double t = 0.0;
      for (int w = 0; w < nw; ++w) t += s12l[w][kl][which * NC + ks];
      d.part_b[(size_t)blockIdx.x * 2 * K + which * K + k] = t;
    }
  }
}

// ================================================================ s3 pass (A8)
// One lane per training link (the reference's own list, p < q): per-lane accumulators over
// all of a lane's links, then the same two-pass LDS column walk as the phi kernel.  Block 0 also
// publishes s1, s2 (folded from k_finalize_lpl's partial rows).  When cls_next is set the launch
// carries extra blocks after the nb_c s3 blocks: they classify the links for the NEXT sweep at the
// same time on other CUs (prune() of this sweep is complete: the launch follows k_finalize_lpl);
// the scatter pass follows on the tail launch.
// threads per block of k_s3_lpl: up to K = 20 the KR accumulators fit the 128 registers of a 16-wave
// block; beyond that 8 waves (two per SIMD) share the register file
// ... K = 21..32 (KC = 12, 14, 16) have a second shape, 12-wave blocks (768 threads, 168 VGPRs = three waves per SIMD; one block per
// CU either way: the launch is kept co-resident).  KC = 12 fits it without a spill, KC = 14 / 16 spill 8 / 44 VGPRs there.  It pays
// where the 8-wave shape needs two links per lane, i.e. beyond 192 x 512 links, and costs a small graph 1.5 - 3 us otherwise:
// ca-AstroPh K=22 s3 28.8 -> 21.4 us, K=28 29.2 -> 22.9, K=32 29.6 -> 26.0; LFR K=28 13.7 -> 16.7, n=1000 K=24 13.4 -> 15.0
// (profiles/r06a_*, r06b_*, r06c_*) -- lpl_s3_threads decides by the link count.
constexpr int s3_threads(int kc) { return kc >= 12 ? 512 : 1024; }

template <int KC, int NTH = s3_threads(KC)>
__global__ __launch_bounds__(NTH) void k_s3_lpl(Geometry geo, DeviceState d, Params prm) {
  DevCtrl *ctrl = d.ctrl;
  constexpr int KR = LplCfg<KC>::KR, SROW = LplCfg<KC>::SROW;
  constexpr int NWV = NTH / 64, NWORK = NTH / 256;
  constexpr bool PEEL = KC >= 12 && KC <= 20;   // see below
  if constexpr (!PEEL) {
    if (ctrl->stopped) return;   // (the instantiations without the first-link form keep the order of accesses they had)
  }
  // three-launch sweeps: the very last block only folds s1, s2 from k_finalize_lpl's partial rows, from the
  // start of the launch, so that nobody on the critical path has to (it arrives on the s3 ticket like an s3 block)
  const bool fold_role = d.fused3 && blockIdx.x == gridDim.x - 1u;
  // The s3 blocks' path through this launch is a chain of dependent misses: control block -> link -> converged flags ->
  // rows -> (block partial, ticket) -> the last block's serial stage.  On three-launch sweeps (graphs of a few hundred
  // thousand links: about ONE link per lane) the first link of a lane, whose address depends on nothing, is requested
  // together with the control block (a relaxed wavefront-scope atomic = a plain load the compiler leaves in place), and
  // its two rows are requested together with the flags instead of behind them (below): two hops less.
  // (where the registers allow: 512-thread blocks, K = 21..40.  At K <= 20 the 1024-thread block leaves 128 registers per
  //  lane, the peeled form spills, and it measured slower -- ca-AstroPh K=20 52.1 -> 53.0 us per sweep -- while LFR K=28 gains
  //  32.2 -> 31.5 us; K > 40 spills as well.  profiles/r03q_s3_first_link.txt.  The other instantiations compile to what
  //  they were.)
  bool peel = false;
  unsigned long long first_link = 0ull;
  uint32_t c_parity0 = 0;
  if constexpr (PEEL) {
    const uint64_t nl0 = (fold_role || blockIdx.x >= d.nb_c) ? 0 : d.link_end - d.link_begin;
    const uint64_t i0 = (uint64_t)blockIdx.x * NTH + threadIdx.x;
    // ... and only when the launch really has at most one link per lane: with two (ca-AstroPh at K > 20, where the block
    // count is capped) the peeled form measured slower (K = 32: 84.2 -> 87.8 us per sweep)
    peel = d.fused3 != 0 && i0 < nl0 && nl0 <= (uint64_t)d.nb_c * NTH;
    if (peel)
      first_link = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(d.links) + (d.link_begin + i0),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const uint32_t c_stopped0 = ctrl->stopped;
    c_parity0 = ctrl->parity;
    if (c_stopped0) return;
  }
  __shared__ __attribute__((aligned(16))) double lds[NWV][32 * SROW];
  __shared__ double red[NWV][64];
  __shared__ ClsWork shw[NWORK];
  STAMP(2, 0);
  __shared__ uint32_t hflag;
  if (blockIdx.x >= d.nb_c && !fold_role) {   // the blocks after the s3 blocks: link classes of the NEXT sweep
    const uint32_t rb = blockIdx.x - d.nb_c, nrb = gridDim.x - d.nb_c - (d.fused3 ? 1u : 0u);
    if (d.fused3) {
      // three-launch sweeps: both passes here, handed over inside the launch (the scatter pass has no
      // later launch to ride on before the next phi pass needs its lists)
      __shared__ unsigned long long scan_lds[NWV + 1];
      cls_classify_in_launch<NWORK, NTH>(geo, d, prm, shw, rb, nrb, scan_lds, &hflag);
    } else {
      cls_count_tiles<NWORK>(geo, d, prm, shw, rb, nrb, true);   // count pass; the scatter pass rides on k_tail
    }
    STAMP(2, 7);
    return;
  }
  // the log table for the serial stage of a three-launch sweep: fetched ahead of its use, used by the last block only
  double2 ltv = make_double2(0.0, 0.0);
  double suma = 0.0;   // sum[k] of this sweep (k_finalize_lpl's block 0 left it in kvec_a)
  // ... and what its thread 0 needs from the control block and the class totals: all of it is at rest
  // until that block itself writes, so every block fetches it and only the last one uses it
  uint32_t c_iter = 0, c_sd = 0, c_cpar = 0, c_par = 0;
  unsigned long long c_l0 = 0, c_l1 = 0, c_l2 = 0;
  auto load_tail_inputs = [&]() {
    if (d.fused3) {
      if (threadIdx.x < 128) ltv = make_double2(d.logtab[2 * threadIdx.x], d.logtab[2 * threadIdx.x + 1]);
      if (threadIdx.x < geo.K) suma = d.kvec_a[threadIdx.x];
      c_cpar = ctrl->cls_par;
      if (threadIdx.x == 0) {
        c_iter = ctrl->iter; c_sd = ctrl->sweeps_done; c_par = ctrl->parity;
        const uint32_t *lt = d.ltot + c_cpar * 8u;
        c_l0 = lt[3]; c_l1 = lt[4]; c_l2 = lt[5];
      }
    }
  };
  if constexpr (!PEEL) load_tail_inputs();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t K = geo.K, ld = geo.ld;
  const uint32_t *__restrict__ conv = d.conv + (size_t)(PEEL ? c_parity0 : ctrl->parity) * geo.n_alloc;
  const double *__restrict__ mphi = d.mphi;
  double s3[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) s3[k] = 0.0;
  if (peel) {
    // the lane's first link, rows requested with the flags; a link with exactly one converged endpoint picks its one
    // element out of the row registers (src/linksampling.cc:739-742, quirk Q2: element index pc, column pc - 1; an
    // index of K reads nothing).  Products straight into the accumulators: nothing else is live yet.
    const uint32_t p = (uint32_t)first_link, q = (uint32_t)(first_link >> 32);   // links[2 l], links[2 l + 1]
    const double *rp = mphi + (size_t)p * ld, *rq = mphi + (size_t)q * ld;
    const uint32_t pc = conv[p], qc = conv[q];
    const bool dense = (pc != 0) == (qc != 0);
    const uint32_t one = pc ? pc : qc;                   // the converged endpoint's flag (dense: unused)
    const int sel = (!dense && one < K) ? (int)one : -1;
    const int tgt = dense ? -1 : (int)one - 1;
    const bool from_q = pc != 0;                         // pc && !qc: the element comes from q's row
    double val = 0.0;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const double2 a = *reinterpret_cast<const double2 *>(rp + 2 * c);
      const double2 b = *reinterpret_cast<const double2 *>(rq + 2 * c);
      s3[2 * c] = dense ? a.x * b.x : 0.0;
      s3[2 * c + 1] = dense ? a.y * b.y : 0.0;
      const double e0 = from_q ? b.x : a.x, e1 = from_q ? b.y : a.y;
      val = (sel == 2 * c) ? e0 : val;
      val = (sel == 2 * c + 1) ? e1 : val;
    }
    if (__any(!dense)) {
#pragma unroll
      for (int k = 0; k < KR; ++k) s3[k] += (k == tgt) ? val : 0.0;
    }
  }
  if constexpr (PEEL) load_tail_inputs();   // behind the peeled link, whose row registers they would otherwise share
  const uint64_t nl = fold_role ? 0 : d.link_end - d.link_begin;
  for (uint64_t i = (uint64_t)blockIdx.x * NTH + threadIdx.x + (peel ? (uint64_t)d.nb_c * NTH : 0ull); i < nl; i += (uint64_t)d.nb_c * NTH) {
    const uint64_t l = d.link_begin + i;
    const uint32_t p = d.links[2 * l], q = d.links[2 * l + 1];
    const uint32_t pc = conv[p], qc = conv[q];
    if (pc && !qc) {          // src/linksampling.cc:739-740, quirk Q2 (index pc, not pc-1)
      const double val = pc < K ? mphi[(size_t)q * ld + pc] : 0.0;
#pragma unroll
      for (int k = 0; k < KR; ++k) s3[k] += (k == (int)pc - 1) ? val : 0.0;
    } else if (!pc && qc) {   // :741-742
      const double val = qc < K ? mphi[(size_t)p * ld + qc] : 0.0;
#pragma unroll
      for (int k = 0; k < KR; ++k) s3[k] += (k == (int)qc - 1) ? val : 0.0;
    } else {
      // chunk by chunk: only the accumulators stay live across the row
      const double *rp = mphi + (size_t)p * ld, *rq = mphi + (size_t)q * ld;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const double2 a = *reinterpret_cast<const double2 *>(rp + 2 * c);
        const double2 b = *reinterpret_cast<const double2 *>(rq + 2 * c);
        s3[2 * c] += a.x * b.x;
        s3[2 * c + 1] += a.y * b.y;
      }
    }
  }
  STAMP(2, 1);
  // wave total per column: the 64 rows go through LDS in two passes of 32; lane (part, k) adds the
  // rows r = part, part + NP, ... of column k (NP = 64 / KR lanes share a column), then the parts
  {
    constexpr int NP = 64 / KR;
    double *mylds = lds[wave];
    double *mine = mylds + (lane & 31) * SROW;
    const int kcol = lane % KR, part = lane / KR;
    double acc = 0.0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((lane >> 5) == half) {
#pragma unroll
        for (int c = 0; c < KC; ++c) *reinterpret_cast<double2 *>(mine + 2 * c) = make_double2(s3[2 * c], s3[2 * c + 1]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (part < NP) {
        double v[(32 + NP - 1) / NP];
#pragma unroll
        for (int i = 0; i < (32 + NP - 1) / NP; ++i) {
          const int r = part + i * NP;
          v[i] = r < 32 ? mylds[r * SROW + kcol] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < (32 + NP - 1) / NP; ++i) acc += v[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // parts in order
    double tot = 0.0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) tot += __shfl(acc, pp * KR + kcol, 64);
    if (lane < KR) red[wave][lane] = tot;
  }
  __syncthreads();
  if (threadIdx.x < K && !fold_role) {
    double t = 0.0;
    for (int w = 0; w < NWV; ++w) t += red[w][threadIdx.x];
    if (d.fused3 || d.shard_c) st_agent(&d.part_c[(size_t)blockIdx.x * K + threadIdx.x], t);   // read by the last block of THIS launch
    else d.part_c[(size_t)blockIdx.x * K + threadIdx.x] = t;
  }
  STAMP(2, 2);
  if constexpr (KC <= 16) {
    if (d.fold && (d.fused3 ? fold_role : blockIdx.x == 0)) {   // s1, s2 of this sweep (folded from k_finalize_lpl's partial rows)
      __syncthreads();
      double *tmp = &lds[0][0];        // 16 row groups x 64 columns; the staging area is free again
      __shared__ double out64[64];
      fold_rows<64, NTH>(d.part_b, d.nb_b, 2 * K, tmp, out64);
      if (threadIdx.x < 2 * K) {
        if (d.fused3) st_agent(&d.kvec_c[threadIdx.x], out64[threadIdx.x]);
        else d.kvec_c[threadIdx.x] = out64[threadIdx.x];
      }
    }
    if (d.shard_c) {
      // ---- node-block sweeps: the last s3 block to arrive adds the blocks' partial rows in block order and leaves this
      // rank's share of s3 next to s1, s2 (block 0, above) for the all-reduce that follows the launch
      if (!last_block_arrives(d.s3_ctl, d.nb_c, &hflag)) return;
      double *tmp = &lds[0][0];
      {   // s1, s2: the finalise launch's partial rows (<= SVILS_FOLD_ROWS of 2K <= 64 columns), row groups of NTH / 64 in order
        const uint32_t c = threadIdx.x & 63u, r0 = threadIdx.x >> 6;
        double t = 0.0;
        if (c < 2 * K)
          for (uint32_t r = r0; r < d.nb_b; r += NTH / 64) t += d.part_b[(size_t)r * 2 * K + c];
        tmp[r0 * 64 + c] = t;
        __syncthreads();
        if (threadIdx.x < 2 * K) {
          double u = 0.0;
#pragma unroll
          for (uint32_t i = 0; i < NTH / 64; ++i) u += tmp[i * 64 + threadIdx.x];
          d.kvec_c[threadIdx.x] = u;
        }
        __syncthreads();
      }
      {
        constexpr uint32_t RG = NTH / 32, NL = 256 / RG;   // row groups x 32 columns; nb_c <= 192 rows
        const uint32_t c = threadIdx.x & 31u, r0 = threadIdx.x >> 5;
        double v[NL];
#pragma unroll
        for (uint32_t i = 0; i < NL; ++i) {
          const uint32_t r = r0 + RG * i;
          v[i] = (c < K && r < d.nb_c) ? ld_agent(&d.part_c[(size_t)r * K + c]) : 0.0;
        }
        double t = 0.0;
#pragma unroll
        for (uint32_t i = 0; i < NL; ++i) t += v[i];
        tmp[r0 * 32 + c] = t;
      }
      __syncthreads();
      if (threadIdx.x < K) {
        double t = 0.0;
#pragma unroll
        for (uint32_t i = 0; i < NTH / 32; ++i) t += tmp[i * 32 + threadIdx.x];
        d.kvec_c[2 * K + threadIdx.x] = t;
      }
      return;
    }
    if (d.fused3) {
      // ---- three-launch sweeps: the last s3 block to arrive closes the sweep -------------------------
      // s3 = its blocks' partial rows in block order; lambda update + set_dir_exp(lambda)
      // (src/linksampling.cc:748-759); write_comm for the next sweep (:768-774), _iter++ (:787).  The
      // likelihood row, stop rule and annealing switch of this sweep follow as a role of the next launch.
      STAMP(2, 3);
      if (!last_block_arrives(d.s3_ctl, d.nb_c + 1u, &hflag)) return;
      STAMP(2, 4);
      __shared__ double2 logtab[128];
      __shared__ double s3tot[32];
      double *tmp = &lds[0][0];
      if (threadIdx.x < 128) logtab[threadIdx.x] = ltv;
      double s1 = 0.0, s2 = 0.0;
      {
        constexpr uint32_t RG = NTH / 32, NL = 256 / RG;   // row groups x 32 columns; nb_c <= 192 rows
        const uint32_t c = threadIdx.x & 31u, r0 = threadIdx.x >> 5;
        double v[NL];
#pragma unroll
        for (uint32_t i = 0; i < NL; ++i) {
          const uint32_t r = r0 + RG * i;
          v[i] = (c < K && r < d.nb_c) ? ld_agent(&d.part_c[(size_t)r * K + c]) : 0.0;
        }
        if (threadIdx.x < K) { s1 = ld_agent(&d.kvec_c[threadIdx.x]); s2 = ld_agent(&d.kvec_c[K + threadIdx.x]); }   // same wait
        double t = 0.0;
#pragma unroll
        for (uint32_t i = 0; i < NL; ++i) t += v[i];
        tmp[r0 * 32 + c] = t;
      }
      __syncthreads();
      if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (uint32_t i = 0; i < NTH / 32; ++i) t += tmp[i * 32 + threadIdx.x];
        s3tot[threadIdx.x] = t;
      }
      __syncthreads();
      if (threadIdx.x < K) {
        const uint32_t k = threadIdx.x;
        const double l0 = prm.eta0 + suma;
        const double l1 = prm.eta1 + (s1 * s1 - s2 - s3tot[k]);
        d.lambda[2 * k] = l0;
        d.lambda[2 * k + 1] = l1;
        d.kvec_c[2 * K + k] = s3tot[k];
        const double ps = digamma(l0 + l1, logtab);
        d.elogbeta[2 * k] = digamma(l0, logtab) - ps;
        d.elogbeta[2 * k + 1] = digamma(l1, logtab) - ps;
      }
      DevCtrl *c = d.ctrl;   // field by field: a whole-struct copy goes through scratch
      const uint32_t cpar0 = c_cpar;
      uint32_t *ltot = d.ltot + cpar0 * 8u;
      // did this launch's roles classify the next sweep's links?  If no flag changed they did not, and the current
      // lists, totals and shortcut histogram stay current
      const bool reclassified = cls_next_needed_from_args(d);
      if (threadIdx.x == 0) {
        const uint32_t iter = c_iter, sd = c_sd;
        c->parity = c_par ^ 1u;  // prune()'s flags become current
        c->links_dense = c_l0; c->links_sparse = c_l1; c->links_shortcut = c_l2;
        if (d.sweep_stats) {
          unsigned long long *st = d.sweep_stats + (size_t)(sd % d.sweep_stats_cap) * 4;
          st[0] = c_l0; st[1] = c_l1; st[2] = c_l2; st[3] = sd;
        }
        c->sweeps_done = sd + 1u;
        c->write_comm = (iter % prm.reportfreq == prm.reportfreq - 1) ? 1 : 0;
        c->v_pending = (d.nv > 0 && iter % prm.reportfreq == 0) ? 1u : 0u;
        c->v_iter = iter;
        c->iter = iter + 1;
        if (reclassified) c->cls_par = cpar0 ^ 1u;   // the classes this launch's roles computed for the next sweep become current
      }
      // the link counts / shortcut histogram of the finished sweep are consumed: clear them for the
      // classification two sweeps ahead (unless they stay in use); the sweep's column sums are consumed either way
      __syncthreads();
      if (reclassified) {
        if (threadIdx.x < 8) ltot[threadIdx.x] = 0;
        if (threadIdx.x < K) d.shist[(size_t)cpar0 * K + threadIdx.x] = 0ull;
      }
      for (uint32_t i = threadIdx.x; i < 1024u; i += NTH) d.sumfx[(size_t)cpar0 * 1024 + i] = 0;   // (blocks of 512 or 1024 threads)
      STAMP(2, 5);
    }
  }
}

// ------------------------------------------------------------------ launchers
// K = 57..64 would need a phi row of 64 doubles per lane (256 VGPRs and ~1 KB of scratch per lane): measured slower than
// the row-per-wavefront kernels on every size (ca-AstroPh K=64: 0.161 vs 0.156 ms per sweep, n=1e6: 9.2 vs 6.5 ms), so the
// lane-per-link layout ends at K = 56 (there it still wins: 0.124 vs 0.156 ms, 6.0 vs 6.6 ms)
bool use_lpl(uint32_t K) { return K <= 56; }
// waves per block of k_phi_lpl, two blocks per CU either way (a grid of two blocks per CU leaves at
// most SVILS_FOLD_ROWS partial rows of `sum` for the consumers to fold).  K <= 32: the pipelined
// kernel, LPL_PIPE_WAVES / 2 waves per SIMD with up to 256 VGPRs each.  K = 33..64: a phi row of up
// to 128 VGPRs, two waves per SIMD, blocks of four waves.
#ifndef LPL_PIPE_WAVES
#define LPL_PIPE_WAVES 4
#endif
#ifndef LPL_PIPE
#define LPL_PIPE 0
#endif
constexpr bool lpl_pipe(int kc) { return LPL_PIPE && kc <= 16; }
constexpr int lpl_waves(int kc) { return lpl_pipe(kc) ? LPL_PIPE_WAVES : kc >= 18 ? 4 : kc >= LPL_MID_KC ? LPL_MID_NW : 8; }
int lpl_phi_waves(uint32_t K) { return LPL_PIPE && K <= 32 ? LPL_PIPE_WAVES : K > 32 ? 4 : K > (LPL_MID_KC == 12 ? 20 : 24) ? LPL_MID_NW : 8; }

#define LPL_DISPATCH(K_, CALL)                 \
  do {                                         \
    if ((K_) <= 8) { CALL(4); }                \
    else if ((K_) <= 16) { CALL(8); }          \
    else if ((K_) <= 20) { CALL(10); }         \
    else if ((K_) <= 24) { CALL(12); }         \
    else if ((K_) <= 28) { CALL(14); }         \
    else if ((K_) <= 32) { CALL(16); }         \
    else if ((K_) <= 36) { CALL(18); }         \
    else if ((K_) <= 40) { CALL(20); }         \
    else if ((K_) <= 48) { CALL(24); }         \
    else { CALL(28); }                         \
  } while (0)

// blocks of k_phi_lpl that fit on the device at once (registers and LDS of the instantiation
// chosen for K): one grid of this size keeps every wavefront slot busy with no second round
uint32_t lpl_phi_resident_blocks(uint32_t K, int device) {
  int per_cu = 0, cus = 0;
#define CALL(KC_) \
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_phi_lpl<KC_, lpl_waves(KC_), lpl_pipe(KC_)>, 64 * lpl_waves(KC_), 0)
  LPL_DISPATCH(K, CALL);
#undef CALL
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (per_cu <= 0 || cus <= 0) return SVILS_FOLD_ROWS;
  return (uint32_t)per_cu * (uint32_t)cus;
}

void launch_classify(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  uint32_t nb = (d.cls_ntiles + 3u) / 4u;
  if (nb > 256u) nb = 256u;
  if (nb == 0) nb = 1;
  hipLaunchKernelGGL(k_cls_count, dim3(nb), dim3(1024), 0, s, g, d, p);
  hipLaunchKernelGGL(k_cls_scatter, dim3(nb), dim3(1024), 0, s, g, d);
}
uint32_t lpl_s3_threads(uint32_t K, uint64_t nlinks) {
  if (K <= 20) return 1024u;
  return (K <= 32 && nlinks > 192ull * 512ull) ? 768u : 512u;   // (see s3_threads)
}
// validation-role blocks: two pairs per group and pass, at most 64 blocks (the last one adds the
// partials serially)
uint32_t lpl_validation_blocks(const Geometry &g, uint32_t nv, uint32_t K) {
  if (nv == 0) return 1;
  const uint32_t per_block = (uint32_t)lpl_phi_waves(K) * (uint32_t)(64 / g.W) * 2u;
  uint32_t nb = (nv + per_block - 1) / per_block;
  return nb > 64u ? 64u : nb;
}
void launch_validate_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  hipLaunchKernelGGL(k_validate_lpl, dim3(d.nvb ? d.nvb : 1u), dim3(256), 0, s, g, d, p);
}
uint32_t lpl_finalize_waves(uint32_t K, uint64_t nodes, uint32_t cus) {
  const int nc = (K > 24 && K <= 32) || K > 48 ? 4 : 3;
  const uint32_t w = (uint32_t)fin_threads(nc) / 64u;
  // four communities per lane: 12-wave blocks when 8-wave ones would not hold the graph in one round (see fin_threads)
  if (nc == 4 && nodes > (uint64_t)w * (64u / (uint32_t)lpl_finalize_group(K)) * cus) return 12u;
  return w;
}
// classification blocks riding on the s3 launch (one worker per 256 threads, ideally one tile each)
uint32_t lpl_cls_blocks(const DeviceState &d) {
  if (!d.cls_next) return 0;
  const uint32_t wpb = d.s3_threads / 256u;
  // one tile per worker when the s3 blocks leave enough CUs for that many blocks, else two (three-launch
  // sweeps keep up to two tiles per worker in registers), never more than 64 blocks
  uint32_t nb = (d.cls_ntiles + wpb - 1u) / wpb;
  if (!d.fused3) {   // spin-free count pass (the scatter pass rides on the tail launch): no co-residency to respect
    if (nb > 512u) nb = 512u;
    return nb ? nb : 1u;
  }
  if (nb + d.nb_c > 240u) nb = (d.cls_ntiles + 2u * wpb - 1u) / (2u * wpb);
  if (nb > 64u) nb = 64u;
  return nb ? nb : 1u;
}
// scatter-pass blocks (one worker each) riding on the tail launch
uint32_t lpl_scatter_blocks(const DeviceState &d) {
  if (!d.cls_next) return 0;
  // (a worker walks its tiles' 1024-entry sub-tiles one after the other, each a chain of dependent accesses: at most
  //  2048 tiles exist, and with one block per tile the pass is 4x shorter at n = 1e6 than with 512 blocks)
  uint32_t nb = d.cls_ntiles < 2048u ? d.cls_ntiles : 2048u;
  return nb ? nb : 1u;
}
void launch_phi_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
#define CALL(KC_)                                                                                   \
  hipLaunchKernelGGL((k_phi_lpl<KC_, lpl_waves(KC_), lpl_pipe(KC_)>), dim3(d.nb_a + (d.fused3 ? d.nvb : 0u)),      \
                     dim3(64 * lpl_waves(KC_)), 0, s, g, d, p)
  LPL_DISPATCH(g.K, CALL);
#undef CALL
}
// lanes per node in k_finalize_lpl (every lane ceil(K / lanes) communities) -> nodes per wavefront
int lpl_finalize_group(uint32_t K) { return K <= 32 ? 8 : 16; }
#define FIN_NC3_MAXK 24
#define FIN_DISPATCH(K_, FIN)        \
  do {                               \
    if ((K_) <= 8) FIN(8, 1);        \
    else if ((K_) <= 16) FIN(8, 2);  \
    else if ((K_) <= FIN_NC3_MAXK) FIN(8, 3);  \
    else if ((K_) <= 32) FIN(8, 4);  \
    else if ((K_) <= 48) FIN(16, 3); \
    else FIN(16, 4);                 \
  } while (0)
// blocks of k_finalize_lpl the device holds at once (the full-sweep instantiation)
uint32_t lpl_finalize_resident_blocks(uint32_t K, int device) {
  int per_cu = 0, cus = 0;
#define FIN(W_, NC_) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_finalize_lpl<W_, NC_, false>, fin_threads(NC_), 0)
  FIN_DISPATCH(K, FIN);
#undef FIN
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (per_cu <= 0 || cus <= 0) return 256u;
  return (uint32_t)per_cu * (uint32_t)cus;
}
void launch_finalize_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
#define FIN(W_, NC_)                                                                                \
  do {                                                                                              \
    if (p.stoch) hipLaunchKernelGGL((k_finalize_lpl<W_, NC_, true>), dim3(d.nb_b), dim3(fin_threads(NC_)), 0, s, g, d, p); \
    else if (d.light && NC_ == 4 && d.fin_waves == 12u)                                             \
      hipLaunchKernelGGL((k_finalize_lpl<W_, NC_, false, true, (NC_ == 4 ? 768 : fin_threads(NC_))>), dim3(d.nb_b), dim3(768), 0, s, g, d, p); \
    else if (d.light) hipLaunchKernelGGL((k_finalize_lpl<W_, NC_, false, true>), dim3(d.nb_b), dim3(fin_threads(NC_)), 0, s, g, d, p); \
    else if (NC_ == 4 && d.fin_waves == 12u)                                                        \
      hipLaunchKernelGGL((k_finalize_lpl<W_, NC_, false, false, (NC_ == 4 ? 768 : fin_threads(NC_))>), dim3(d.nb_b), dim3(768), 0, s, g, d, p); \
    else hipLaunchKernelGGL((k_finalize_lpl<W_, NC_, false>), dim3(d.nb_b), dim3(fin_threads(NC_)), 0, s, g, d, p);  \
  } while (0)
  FIN_DISPATCH(g.K, FIN);
#undef FIN
}
// blocks of k_s3_lpl the device holds at once.  The three-launch sweep needs its classification role blocks (<= 64, + 1)
// co-resident: they hand tile counts to each other inside the launch.  A whole MI355X holds hundreds; a CPX partition
// (32 CUs) or a masked device may not, and then the handle keeps the four-launch sweep (spin-free passes).
uint32_t lpl_s3_resident_blocks(uint32_t K, int device, uint32_t threads, int assume_cus) {
  int per_cu = 0, cus = 0;
  if (threads == 768u && K > 20 && K <= 32) {   // the 12-wave shape launch_s3_lpl takes for this graph (lpl_s3_threads)
    if (K <= 24) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_s3_lpl<12, 768>, 768, 0);
    else if (K <= 28) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_s3_lpl<14, 768>, 768, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_s3_lpl<16, 768>, 768, 0);
  } else {
#define CALL(KC_) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_s3_lpl<KC_>, s3_threads(KC_), 0)
    LPL_DISPATCH(K, CALL);
#undef CALL
  }
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (assume_cus > 0) cus = assume_cus;   // (libsvils_testing.so: pretend to be a partition of this many CUs)
  if (per_cu <= 0 || cus <= 0) return 0;
  return (uint32_t)per_cu * (uint32_t)cus;
}
void launch_s3_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
  const dim3 grid(d.nb_c + lpl_cls_blocks(d) + (d.fused3 ? 1u : 0u));
  if (d.s3_threads == 768u && g.K > 20 && g.K <= 32) {   // the 12-wave shape of KC = 12 / 14 / 16 (lpl_s3_threads)
    if (g.K <= 24) hipLaunchKernelGGL((k_s3_lpl<12, 768>), grid, dim3(768), 0, s, g, d, p);
    else if (g.K <= 28) hipLaunchKernelGGL((k_s3_lpl<14, 768>), grid, dim3(768), 0, s, g, d, p);
    else hipLaunchKernelGGL((k_s3_lpl<16, 768>), grid, dim3(768), 0, s, g, d, p);
    return;
  }
#define CALL(KC_) hipLaunchKernelGGL((k_s3_lpl<KC_>), grid, dim3(s3_threads(KC_)), 0, s, g, d, p)
  LPL_DISPATCH(g.K, CALL);
#undef CALL
}

}  // namespace svils
