// svils_lpl.hip -- lane-per-link kernels for small K (K <= 32) on gfx950.
//
// With K = 20..28 a group-per-row layout leaves 37 % of a wavefront's lanes idle
// and spends more time in cross-lane softmax reductions than in exp().  Here one
// LANE owns one directed link: wave-item w = the 64 consecutive entries
// [64w, 64w+64) of the symmetric CSR, each lane loops over k in registers (no
// cross-lane reduction for the softmax at all), and the per-node sums of
// gammanext are formed by staging the 64 phi rows in LDS and letting lane k walk
// column k over the 64 rows IN ENTRY ORDER -- which is exactly the order in which
// the reference's link loop adds to gammanext[x] (src/linksampling.cc:696-701).
// A node's run of entries can straddle wave-items; the piece that starts at lane
// 0 of an item goes to slot_f[item], a piece that ends at lane 63 to
// slot_l[item], interior pieces straight to gamma[node]; the finalise kernel
// re-derives the same rule from rowptr and adds the pieces in item order.  No
// floating-point atomics, bit-reproducible, perfectly balanced (hubs included).
#include "svils_devutil.h"

namespace svils {

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

// ============================================================== phi pass (A6)
template <int KC, int NW>
__global__ __launch_bounds__(64 * NW, (KC >= 16 ? 3 : 1)) void k_phi_lpl(Geometry geo, DeviceState d, Params prm) {
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  constexpr int KR = LplCfg<KC>::KR, SROW = LplCfg<KC>::SROW;
  __shared__ __attribute__((aligned(16))) double lds[NW][32 * SROW];
  __shared__ double red[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t K = geo.K, ld = geo.ld;
  const bool write_comm = ctrl->write_comm != 0;
  const bool sparse_iter = (long long)ctrl->iter > (long long)prm.sparse_after;  // _iter > 1000, src/linksampling.cc:634
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  const double *__restrict__ elogpi = d.elogpi;
  double *mylds = lds[wave];

  // Elogbeta[.][0], wave-uniform but kept in VGPRs on purpose: as KR scalar pairs it made the
  // SGPR file spill through v_writelane/v_readlane inside the hot loop.  Padding columns
  // (k >= K) get -inf, which masks them in the softmax without any select.
  __shared__ double eb[KR];
  if (threadIdx.x < KR) eb[threadIdx.x] = (threadIdx.x < K) ? d.elogbeta[2 * threadIdx.x] : NEG_INF;
  __syncthreads();
  double csum = 0.0;  // lane k: partial of sum[k]
  unsigned long long n_dense = 0, n_sparse = 0, n_short = 0;

  for (uint32_t it = blockIdx.x * NW + wave; it < d.lpl_nitems; it += gridDim.x * NW) {
    const uint64_t e = (d.lpl_w0 + it) * 64 + lane;
    const bool valid = e >= d.ent_begin && e < d.ent_end;
    uint32_t p = 0xffffffffu, q = 0;
    if (valid) {
      p = d.erow[e];
      q = d.col[e];
    }
    double phi[KR];
    double *mine = mylds + (lane & 31) * SROW;   // this lane's staged phi row (two passes of 32 rows)
    int tagk = -1;   // community this link tags (src/linksampling.cc:668-681,704-717), -1: none
    int one_at = -1; // >= 0: the row is the unit vector e_c (converged shortcut), stored straight to LDS
    bool dense_row = false;
    if (valid) {
      const uint32_t pc = conv[p], qc = conv[q];
      const bool count_me = q > p;
      if ((pc != 0) != (qc != 0)) {
        // exactly one endpoint converged: src/linksampling.cc:622-631
        one_at = (int)(pc ? pc : qc) - 1;
        n_short += count_me;
      } else {
        dense_row = true;
        // x_k = (Elogpi[p][k] + Elogpi[q][k]) + Elogbeta[k][0], the reference's order (:686)
        {
          const double *rp = elogpi + (size_t)p * ld, *rq = elogpi + (size_t)q * ld;
#pragma unroll
          for (int c = 0; c < KC; ++c) {
            const double2 a = *reinterpret_cast<const double2 *>(rp + 2 * c);
            const double2 b = *reinterpret_cast<const double2 *>(rq + 2 * c);
            phi[2 * c] = (a.x + b.x) + eb[2 * c];
            phi[2 * c + 1] = (a.y + b.y) + eb[2 * c + 1];
          }
        }
        bool sparse = false;
        if (sparse_iter) {
          sparse = d.active_cnt[p] < geo.k10 && d.active_cnt[q] < geo.k10;
          if (sparse) {
            const unsigned long long inmask = d.amask[p] | d.amask[q];   // kw == 1 for K <= 64
#pragma unroll
            for (int k = 0; k < KR; ++k) phi[k] = ((inmask >> k) & 1ull) ? phi[k] : NEG_INF;
          }
        }
        // branch-free from here so the KR independent exp chains interleave
        double m = NEG_INF;
#pragma unroll
        for (int k = 0; k < KR; ++k) m = max_f64(m, phi[k]);   // padding columns are -inf via eb[]
        if (m != NEG_INF) {
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
        } else {
          dense_row = false;  // empty active-set union (:642-664): the row is zero
        }
        if (count_me) { if (sparse) n_sparse++; else n_dense++; }
      }
    }
    // Stage the phi rows in LDS and let lane k sum column k over them in entry order, flushing at
    // node boundaries.  The 64 rows go through LDS in two passes of 32 (lanes 0-31, then 32-63):
    // half the LDS per wavefront, so that the register file, not LDS, sets the occupancy.
    // Dense rows come from registers; shortcut / invalid / empty rows are zeros (+ a single 1.0)
    // without ever being materialised in registers.
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
    double acc = 0.0;
    int a = 0;
    // one run [a, b] of node `cur` is complete: store its partial gammanext row and its tags.
    // Tags are pre-reduced per run so that a node costs one atomic per wave-item, not one per link.
#define LPL_FLUSH(DST, B)                                                                   \
    do {                                                                                    \
      if ((vmask >> a) & 1ull) {                                                            \
        const uint32_t cur = __builtin_amdgcn_readlane(p, a);                               \
        double *dst = (DST);                                                                \
        dst[lane] = acc;                                                                    \
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
    for (int half = 0; half < 2; ++half) {
      if ((lane >> 5) == half) {
        if (dense_row) {
#pragma unroll
          for (int c = 0; c < KC; ++c) *reinterpret_cast<double2 *>(mine + 2 * c) = make_double2(phi[2 * c], phi[2 * c + 1]);
        } else {
#pragma unroll
          for (int c = 0; c < KC; ++c) *reinterpret_cast<double2 *>(mine + 2 * c) = make_double2(0.0, 0.0);
          if (one_at >= 0) mine[one_at] = 1.0;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if ((uint32_t)lane < K) {
#pragma unroll 1
        for (int rb = 0; rb < 32; rb += 16) {
          double v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = mylds[(rb + j) * SROW + lane];
          // a chunk without a run boundary (about half of them at an average degree of 22) is 16 plain adds
          uint32_t hb = (uint32_t)(heads >> (half * 32 + rb)) & 0xffffu;
          if (half * 32 + rb == 0) hb &= ~1u;
          if (hb == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int r = half * 32 + rb + j;
              if (r > 0 && ((heads >> r) & 1ull)) {      // run [a, r-1] ends (wave-uniform branch)
                LPL_FLUSH((a == 0) ? d.slot_f + (size_t)it * ld : d.gacc + (size_t)cur * ld, r - 1);
                a = r;
              }
              acc += v[j];
            }
          }
        }
        // last run ends at lane 63
        if (half == 1) LPL_FLUSH((a == 0) ? d.slot_f + (size_t)it * ld : d.slot_l + (size_t)it * ld, 63);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#undef LPL_FLUSH
  }

  // per-block partial of `sum`: waves in order
  red[wave][lane] = csum;
  __syncthreads();
  if (threadIdx.x < K) {
    double t = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w][threadIdx.x];
    d.part_a[(size_t)blockIdx.x * K + threadIdx.x] = t;
  }
  __shared__ unsigned long long lcnt[3 * NW];
  block_store_link_counts(n_dense, n_sparse, n_short, d.part_links, lcnt, NW);
}

// ================================================================ s3 pass (A8)
// One lane per training link (the reference's own list, p < q): per-lane
// accumulators over all of a lane's links, one LDS transpose per wave at the end.
template <int KC>
__global__ __launch_bounds__(256) void k_s3_lpl(Geometry geo, DeviceState d) {
  DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;
  constexpr int KR = LplCfg<KC>::KR, SROW = LplCfg<KC>::SROW;
  __shared__ __attribute__((aligned(16))) double lds[256 * SROW];
  __shared__ double red[8][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t K = geo.K, ld = geo.ld;
  const uint32_t *__restrict__ conv = d.conv + (size_t)ctrl->parity * geo.n_alloc;
  const double *__restrict__ mphi = d.mphi;
  double s3[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) s3[k] = 0.0;
  const uint64_t nl = d.link_end - d.link_begin;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nl; i += (uint64_t)gridDim.x * 256) {
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
      double mp[KR], mq[KR];
      load_row_lane<KC>(mphi + (size_t)p * ld, mp);
      load_row_lane<KC>(mphi + (size_t)q * ld, mq);
#pragma unroll
      for (int k = 0; k < KR; ++k) s3[k] += mp[k] * mq[k];
    }
  }
  // block total in a fixed order: 8 groups of 32 rows per column, then the 8 partials
  {
    double *mine = lds + (wave * 64 + lane) * SROW;
#pragma unroll
    for (int c = 0; c < KC; ++c) *reinterpret_cast<double2 *>(mine + 2 * c) = make_double2(s3[2 * c], s3[2 * c + 1]);
  }
  __syncthreads();
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  double part = 0.0;
  if (col < KR)
    for (int r = 0; r < 32; ++r) part += lds[(grp * 32 + r) * SROW + col];
  red[grp][col] = part;
  __syncthreads();
  if (threadIdx.x < K) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
    d.part_c[(size_t)blockIdx.x * K + threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------ launchers
bool use_lpl(uint32_t K) { return K <= 32; }
// waves per block of k_phi_lpl (with the 32-row staging area every KC fits four)
constexpr int lpl_waves(int) { return 4; }
int lpl_phi_waves(uint32_t) { return 4; }

#define LPL_DISPATCH(K_, CALL)                 \
  do {                                         \
    if ((K_) <= 8) { CALL(4); }                \
    else if ((K_) <= 16) { CALL(8); }          \
    else if ((K_) <= 20) { CALL(10); }         \
    else if ((K_) <= 24) { CALL(12); }         \
    else if ((K_) <= 28) { CALL(14); }         \
    else { CALL(16); }                         \
  } while (0)

// blocks of k_phi_lpl that fit on the device at once (registers and LDS of the instantiation
// chosen for K): one grid of this size keeps every wavefront slot busy with no second round
uint32_t lpl_phi_resident_blocks(uint32_t K, int device) {
  int per_cu = 0, cus = 0;
#define CALL(KC_) \
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_phi_lpl<KC_, lpl_waves(KC_)>, 64 * lpl_waves(KC_), 0)
  LPL_DISPATCH(K, CALL);
#undef CALL
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (per_cu <= 0 || cus <= 0) return 768;
  return (uint32_t)per_cu * (uint32_t)cus;
}

void launch_phi_lpl(const Geometry &g, const DeviceState &d, const Params &p, hipStream_t s) {
#define CALL(KC_)                                                                                   \
  hipLaunchKernelGGL((k_phi_lpl<KC_, lpl_waves(KC_)>), dim3(d.nb_a), dim3(64 * lpl_waves(KC_)), 0, s, \
                     g, d, p)
  LPL_DISPATCH(g.K, CALL);
#undef CALL
}
void launch_s3_lpl(const Geometry &g, const DeviceState &d, hipStream_t s) {
#define CALL(KC_) hipLaunchKernelGGL((k_s3_lpl<KC_>), dim3(d.nb_c), dim3(256), 0, s, g, d)
  LPL_DISPATCH(g.K, CALL);
#undef CALL
}

}  // namespace svils
