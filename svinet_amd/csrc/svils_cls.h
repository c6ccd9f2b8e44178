// svils_cls.h -- per-sweep link classification for the lane-per-link layout (K <= 32), shared by
// the kernels that carry its two passes as extra workgroups (k_s3_lpl, k_tail) and by the
// stand-alone k_cls_count / k_cls_scatter.
//
// src/linksampling.cc:622-634 evaluated once per sweep for every owned CSR entry, then a stable
// three-way partition of the entries: 0 full softmax, 1 active-set softmax (_iter > 1000),
// 2 exactly one endpoint converged (O(1) shortcut).  Two plain data-parallel passes over tiles of
// `cls_tile` raw entries (a multiple of 1024, chosen so that a graph has at most a few thousand
// tiles), with no communication between workgroups inside a launch:
//   count   : classify, count the class-0 / class-1 entries of the tile -> tcnt[tile]
//   scatter : (next launch) every worker adds up tcnt[] below its tile, classifies again, scans
//             locally and writes the compacted lists, the per-row prefixes npos[], the totals.
// A worker is 256 threads (four wavefronts); a 1024-thread block hosts four of them.  Inside a
// full sweep the two passes ride on the s3 and the tail launch as extra workgroups, on CUs those
// launches leave idle, so the classification is off the sweep's critical path.
#pragma once
#include "svils_devutil.h"

namespace svils {

struct ClsWork {
  unsigned long long wred[4];
  uint32_t wsum[4];
  uint32_t hist[64];
  uint32_t upper[4];
};

// class of the CSR entry (p, q): 0 softmax, 1 active-set softmax, 2 exactly one endpoint converged
__device__ __forceinline__ int classify_entry(const uint8_t *__restrict__ cflag, bool sparse_iter, uint32_t p, uint32_t q,
                                              uint32_t *col2) {
  const uint32_t fp = cflag[p], fq = cflag[q];   // converged flag | "active_cnt < K / 10" << 7 (svils_internal.h)
  const uint32_t pc = fp & 0x7fu, qc = fq & 0x7fu;
  if ((pc != 0) != (qc != 0)) {          // :622-631
    *col2 = (pc ? pc : qc) - 1u;
    return 2;
  }
  *col2 = 0;
  return (sparse_iter && ((fp & fq) >> 7)) ? 1 : 0;   // :634
}

// next = false: classes of the sweep about to run (flags conv[parity], _iter);
// next = true : classes of the FOLLOWING sweep, computed after prune() of the current one
//               (flags conv[parity ^ 1], _iter + 1), written to the other ltot/shist half.
// Role block `rb` of `nrb`, NWORK workers per block.  Worker 0 also records the arguments for the
// scatter pass, which must not read the control block: it shares its launch with the kernel that
// advances it.
// what a classification works from: which half of conv[], whether the active-set class exists, and
// the ltot/shist half it fills.  FROM_ARGS: recorded by the previous launch (cls_record_args), because
// the launch that carries the passes also advances the control block.
__device__ __forceinline__ void cls_record_args(const DeviceState &d, const Params &prm, bool next) {
  const DevCtrl *ctrl = d.ctrl;
  d.cls_args[0] = next ? (ctrl->parity ^ 1u) : ctrl->parity;
  d.cls_args[1] = (((long long)ctrl->iter + (next ? 1 : 0)) > (long long)prm.sparse_after) ? 1u : 0u;
  d.cls_args[2] = next ? (ctrl->cls_par ^ 1u) : ctrl->cls_par;
  d.cls_args[3] = ctrl->sweeps_done + 1u;   // epoch of the in-launch prefix hand-off
  // [4]: the classes change whatever the flags did -- a stand-alone classification, or the active-set regime switching
  // on between this sweep and the next (:634)
  const bool s_now = (long long)ctrl->iter > (long long)prm.sparse_after, s_next = (long long)ctrl->iter + 1 > (long long)prm.sparse_after;
  d.cls_args[4] = (!next || s_now != s_next) ? 1u : 0u;
}

// does the classification of the NEXT sweep have to run?  (from the recorded arguments: launches that advance the
// control block, or follow the one that did)
__device__ __forceinline__ bool cls_next_needed_from_args(const DeviceState &d) {
  return d.cls_args[4] != 0u || ld_agent(d.cls_epoch) == d.cls_args[3];
}

template <int NWORK, bool FROM_ARGS = false>
__device__ __forceinline__ void cls_count_tiles(const Geometry &geo, const DeviceState &d, const Params &prm,
                                                ClsWork (&shw)[NWORK], uint32_t rb, uint32_t nrb, bool next) {
  uint32_t conv_idx, par;
  bool sparse_iter, needed;
  if (FROM_ARGS) {
    conv_idx = d.cls_args[0]; sparse_iter = d.cls_args[1] != 0u; par = d.cls_args[2];
    needed = cls_next_needed_from_args(d);
  } else {
    // the control block is at rest during this launch: every block reaches the same verdict, worker 0 of block 0
    // records it ([5]) for the scatter pass and for the launch that advances the control block
    const DevCtrl *ctrl = d.ctrl;
    conv_idx = next ? (ctrl->parity ^ 1u) : ctrl->parity;
    sparse_iter = ((long long)ctrl->iter + (next ? 1 : 0)) > (long long)prm.sparse_after;
    par = next ? (ctrl->cls_par ^ 1u) : ctrl->cls_par;
    const bool s_now = (long long)ctrl->iter > (long long)prm.sparse_after;
    needed = !next || s_now != sparse_iter || ld_agent(d.cls_epoch) == ctrl->sweeps_done + 1u;
    if (rb == 0 && threadIdx.x == 0) {
      cls_record_args(d, prm, next);
      d.cls_args[5] = needed ? 1u : 0u;
    }
  }
  if (!needed) return;
  uint32_t *ltot = d.ltot + par * 8u;
  unsigned long long *shist = d.shist + (size_t)par * geo.K;
  const uint32_t *__restrict__ conv = d.conv + (size_t)conv_idx * geo.n_alloc;
  const uint32_t wk = threadIdx.x >> 8, tid = threadIdx.x & 255u;
  const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
  ClsWork &sh = shw[wk];
  const uint32_t rw = rb * NWORK + wk, nrw = nrb * NWORK;
  const uint64_t eb = d.ent_begin, ee = d.ent_end;
  const uint32_t subs = d.cls_tile >> 10;
  const uint32_t iters = (d.cls_ntiles + nrw - 1) / nrw;
  if (tid < 64) sh.hist[tid] = 0;
  if (tid < 4) sh.upper[tid] = 0;
  uint32_t up[3] = {0, 0, 0};
  __syncthreads();
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t tile = rw + it * nrw;
    const bool act = tile < d.cls_ntiles;
    unsigned long long n01 = 0;   // class-0 count << 32 | class-1 count
    if (act) {
      for (uint32_t sub = 0; sub < subs; ++sub) {
        const uint64_t e0 = (uint64_t)(d.cls_tile0 + tile) * d.cls_tile + 1024u * sub + 4u * tid;
        const uint4 pr = *reinterpret_cast<const uint4 *>(d.erow + e0);
        const uint4 qr = *reinterpret_cast<const uint4 *>(d.col + e0);
        const uint32_t pp[4] = {pr.x, pr.y, pr.z, pr.w}, qq[4] = {qr.x, qr.y, qr.z, qr.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint64_t e = e0 + j;
          if (e >= eb && e < ee) {
            uint32_t c2;
            const int c = classify_entry(d.cflag, sparse_iter, pp[j], qq[j], &c2);
            n01 += (c == 0 ? (1ull << 32) : 0ull) + (c == 1 ? 1ull : 0ull);
            if (qq[j] > pp[j]) up[c]++;
            if (c == 2) atomicAdd(&sh.hist[c2 & 63u], 1u);
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n01 += (unsigned long long)__shfl_xor((long long)n01, o, 64);
    __syncthreads();
    if (lane == 0) sh.wred[wv] = n01;
    __syncthreads();
    if (act && tid == 0) {
      const unsigned long long t = sh.wred[0] + sh.wred[1] + sh.wred[2] + sh.wred[3];
      if (FROM_ARGS) st_agent(&d.tcnt[tile], t);   // read by another workgroup of the SAME launch
      else d.tcnt[tile] = t;
    }
  }
  // this block's statistics row: shortcut entries per community column (-> `sum`), then the links
  // (q > p) per class (-> the c, d counters of src/linksampling.cc:726), over all its tiles
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint32_t u = up[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) u += (uint32_t)__shfl_xor((int)u, o, 64);
    if (lane == 0 && u) atomicAdd(&sh.upper[c], u);
  }
  __syncthreads();
  // one integer atomic per block and word (the words were cleared by k_tail two sweeps ago, or by the
  // host before a stand-alone classification); nobody waits for them inside this launch
  if (threadIdx.x < 64 + 3) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < NWORK; ++w) t += threadIdx.x < 64 ? shw[w].hist[threadIdx.x] : shw[w].upper[threadIdx.x - 64];
    if (t) {
      if (threadIdx.x < 64) { if (threadIdx.x < geo.K) atomicAdd(&shist[threadIdx.x], (unsigned long long)t); }
      else atomicAdd(&ltot[3 + (threadIdx.x - 64)], t);
    }
  }
}

// BASES: the exclusive per-tile prefixes are in tbase[] (cls_prefix_handoff) instead of being added
// up from tcnt[] by every worker
template <int NWORK, bool BASES = false>
__device__ __forceinline__ void cls_scatter_tiles(const Geometry &geo, const DeviceState &d, ClsWork (&shw)[NWORK],
                                                  uint32_t rb, uint32_t nrb) {
  const uint32_t conv_idx = d.cls_args[0];
  const bool sparse_iter = d.cls_args[1] != 0u;
  const uint32_t par = d.cls_args[2];
  // BASES: second half of an in-launch classification (the caller has checked); otherwise the count pass's verdict
  if (!BASES && d.cls_args[5] == 0u) return;
  const uint32_t *__restrict__ conv = d.conv + (size_t)conv_idx * geo.n_alloc;
  uint32_t *ltot = d.ltot + par * 8u;
  unsigned long long *shist = d.shist + (size_t)par * geo.K;
  const uint32_t wk = threadIdx.x >> 8, tid = threadIdx.x & 255u;
  const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
  ClsWork &sh = shw[wk];
  const uint32_t rw = rb * NWORK + wk, nrw = nrb * NWORK;
  const uint64_t eb = d.ent_begin, ee = d.ent_end;
  if (d.cls_ntiles == 0) {
    // no owned entry: every owned row is empty in every list, every total is zero
    if (rb == 0) {
      for (uint32_t x = geo.node_begin + threadIdx.x; x <= geo.node_end; x += blockDim.x) {
        d.npos[0][x] = 0; d.npos[1][x] = 0; d.npos[2][x] = 0;
      }
      if (threadIdx.x < 3) ltot[threadIdx.x] = 0;
    }
    return;
  }
  const uint32_t subs = d.cls_tile >> 10;
  const uint32_t iters = (d.cls_ntiles + nrw - 1) / nrw;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t tile = rw + it * nrw;
    const bool act = tile < d.cls_ntiles;
    // class-0 / class-1 entries in the tiles below this one
    unsigned long long pre = 0ull;
    if (BASES) {
      if (act && tid == 0) pre = ld_agent(&d.tbase[tile]);
    } else if (act) {
      for (uint32_t i = tid; i < tile; i += 256u) pre += d.tcnt[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += (unsigned long long)__shfl_xor((long long)pre, o, 64);
    __syncthreads();
    if (lane == 0) sh.wred[wv] = pre;
    __syncthreads();
    pre = sh.wred[0] + sh.wred[1] + sh.wred[2] + sh.wred[3];
    uint32_t run0 = (uint32_t)(pre >> 32), run1 = (uint32_t)pre;   // running bases over the sub-tiles
    for (uint32_t sub = 0; sub < subs; ++sub) {
      const uint64_t e0 = (uint64_t)(d.cls_tile0 + tile) * d.cls_tile + 1024u * sub + 4u * tid;
      uint32_t pp[4] = {0, 0, 0, 0}, qq[4] = {0, 0, 0, 0}, pprev = 0xffffffffu;
      int cls[4] = {3, 3, 3, 3};
      uint32_t c2[4] = {0, 0, 0, 0};
      uint32_t n01 = 0;   // class-0 | class-1 << 16 of this thread's four entries
      if (act) {
        const uint4 pr = *reinterpret_cast<const uint4 *>(d.erow + e0);
        const uint4 qr = *reinterpret_cast<const uint4 *>(d.col + e0);
        pp[0] = pr.x; pp[1] = pr.y; pp[2] = pr.z; pp[3] = pr.w;
        qq[0] = qr.x; qq[1] = qr.y; qq[2] = qr.z; qq[3] = qr.w;
        if (e0 > 0) pprev = d.erow[e0 - 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint64_t e = e0 + j;
          if (e >= eb && e < ee) {
            cls[j] = classify_entry(d.cflag, sparse_iter, pp[j], qq[j], &c2[j]);
            n01 += (cls[j] == 0 ? 1u : 0u) + (cls[j] == 1 ? 0x10000u : 0u);
          }
        }
      }
      // exclusive scan of n01 over the worker
      uint32_t inc = n01;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += t;
      }
      __syncthreads();
      if (lane == 63) sh.wsum[wv] = inc;
      __syncthreads();
      uint32_t woff = 0, ttot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t t = sh.wsum[w];
        if (w < wv) woff += t;
        ttot += t;
      }
      const uint32_t excl = woff + inc - n01;
      uint32_t pos0 = run0 + (excl & 0xffffu), pos1 = run1 + (excl >> 16);
      run0 += ttot & 0xffffu;
      run1 += ttot >> 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t e = e0 + j;
        if (cls[j] != 3) {
          const uint32_t p = pp[j], q = qq[j];
          const uint32_t pos2 = (uint32_t)(e - eb) - pos0 - pos1;
          // first entry of a row: class prefixes of this node and of the empty rows just before it
          const uint32_t prev = j == 0 ? pprev : pp[j - 1];
          if (e == eb || p != prev) {
            for (uint32_t x = (e == eb) ? geo.node_begin : prev + 1u; x <= p; ++x) {
              d.npos[0][x] = pos0; d.npos[1][x] = pos1; d.npos[2][x] = pos2;
            }
          }
          uint32_t a0 = pos0, a1 = pos1, a2 = pos2;
          if (cls[j] == 0) { d.cp[0][pos0] = p; d.cq[0][pos0] = q; a0 = ++pos0; }
          else if (cls[j] == 1) { d.cp[1][pos1] = p; d.cq[1][pos1] = q; a1 = ++pos1; }
          else { d.scol[pos2] = (uint16_t)c2[j]; a2 = pos2 + 1u; }
          if (e == ee - 1) {   // last owned entry: totals, and the empty rows after it
            ltot[0] = a0; ltot[1] = a1; ltot[2] = a2;
            for (uint32_t x = p + 1u; x <= geo.node_end; ++x) {
              d.npos[0][x] = a0; d.npos[1][x] = a1; d.npos[2][x] = a2;
            }
          }
        }
      }
    }
  }
}

// In-launch hand-off between the two passes (three-launch sweeps): every role block arrives on a
// ticket after its count pass; the last one scans tcnt[] into exclusive prefixes tbase[] and
// raises the epoch word; the others wait for it (bounded spin, a few dozen blocks, all of them
// resident next to the s3 blocks: only role blocks ever wait, never for a block that cannot start).
template <int NTH>
__device__ __forceinline__ void cls_prefix_handoff(const DeviceState &d, uint32_t nrb, unsigned long long *scan_lds /*[NTH/64 + 1]*/,
                                                   uint32_t *flag_lds) {
  const uint32_t epoch = d.cls_args[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (last_block_arrives(&d.cls_sync[0], nrb, flag_lds)) {
    unsigned long long run = 0ull;
    for (uint32_t base = 0; base < d.cls_ntiles; base += NTH) {
      const uint32_t i = base + threadIdx.x;
      const unsigned long long v = i < d.cls_ntiles ? ld_agent(&d.tcnt[i]) : 0ull;
      unsigned long long inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = (unsigned long long)__shfl_up((long long)inc, o, 64);
        if (lane >= o) inc += t;
      }
      __syncthreads();
      if (lane == 63) scan_lds[wave] = inc;
      __syncthreads();
      unsigned long long woff = 0ull, tot = 0ull;
      for (int w = 0; w < NTH / 64; ++w) {
        const unsigned long long t = scan_lds[w];
        if (w < wave) woff += t;
        tot += t;
      }
      if (i < d.cls_ntiles) st_agent(&d.tbase[i], run + woff + inc - v);
      run += tot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&d.cls_sync[1], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (threadIdx.x == 0) {
      uint32_t spins = 0;
      while (ld_agent(&d.cls_sync[1]) != epoch) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 22)) { d.ctrl->fault = 1u; d.ctrl->stopped = 1; break; }   // never spin unbounded: flag the error
      }
    }
    __syncthreads();
  }
}

// Both passes by the same worker inside ONE launch (three-launch sweeps).  When every worker has at
// most two 1024-entry tiles -- graphs up to a few hundred thousand links, where each dependent global
// access is a visible share of the sweep -- the entries and their classes stay in registers across
// the hand-off, so the scatter half touches no input again, and the hand-off itself is ticket-free:
// a worker publishes {epoch, class-0 count, class-1 count} of its tiles with agent-scope stores and
// every worker adds up the entries below its own tiles, re-reading an entry until it carries this
// launch's epoch (only role blocks ever wait, all of them resident next to the s3 blocks; bounded
// spin).  Otherwise the two generic passes run back to back around cls_prefix_handoff.
template <int NWORK, int NTH>
__device__ __forceinline__ void cls_classify_in_launch(const Geometry &geo, const DeviceState &d, const Params &prm,
                                                       ClsWork (&shw)[NWORK], uint32_t rb, uint32_t nrb,
                                                       unsigned long long *scan_lds, uint32_t *flag_lds) {
  constexpr int T = 2;   // tiles per worker kept in registers
  if (!cls_next_needed_from_args(d)) return;   // no flag changed in this sweep: the current lists stay current
  const uint32_t nrw = nrb * NWORK;
  if (d.cls_ntiles > T * nrw || d.cls_tile != 1024u || d.cls_ntiles == 0u) {
    cls_count_tiles<NWORK, true>(geo, d, prm, shw, rb, nrb, true);
    cls_prefix_handoff<NTH>(d, nrb, scan_lds, flag_lds);
    cls_scatter_tiles<NWORK, true>(geo, d, shw, rb, nrb);
    return;
  }
  const uint32_t conv_idx = d.cls_args[0];
  const bool sparse_iter = d.cls_args[1] != 0u;
  const uint32_t par = d.cls_args[2], epoch = d.cls_args[3];
  const uint32_t *__restrict__ conv = d.conv + (size_t)conv_idx * geo.n_alloc;
  uint32_t *ltot = d.ltot + par * 8u;
  unsigned long long *shist = d.shist + (size_t)par * geo.K;
  const uint32_t wk = threadIdx.x >> 8, tid = threadIdx.x & 255u;
  const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
  ClsWork &sh = shw[wk];
  const uint32_t rw = rb * NWORK + wk;
  const uint64_t eb = d.ent_begin, ee = d.ent_end;
  if (tid < 64) sh.hist[tid] = 0;
  if (tid < 4) sh.upper[tid] = 0;
  __syncthreads();
  uint32_t pp[T][4], qq[T][4], pprev[T], n01[T], excl[T];
  uint32_t cc[T][4];   // class (3 = not an owned entry) | shortcut column << 2
  uint32_t up[3] = {0, 0, 0};
  // all inputs of both tiles first (entries, then flags), so that they are one latency, not two
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint32_t tile = rw + t * nrw;
    const uint64_t e0 = (uint64_t)(d.cls_tile0 + tile) * 1024u + 4u * tid;
    pprev[t] = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < 4; ++j) { pp[t][j] = 0; qq[t][j] = 0; cc[t][j] = 3u; }
    if (tile < d.cls_ntiles) {
      const uint4 pr = *reinterpret_cast<const uint4 *>(d.erow + e0);
      const uint4 qr = *reinterpret_cast<const uint4 *>(d.col + e0);
      pp[t][0] = pr.x; pp[t][1] = pr.y; pp[t][2] = pr.z; pp[t][3] = pr.w;
      qq[t][0] = qr.x; qq[t][1] = qr.y; qq[t][2] = qr.z; qq[t][3] = qr.w;
      if (e0 > 0) pprev[t] = d.erow[e0 - 1];
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint32_t tile = rw + t * nrw;
    const uint64_t e0 = (uint64_t)(d.cls_tile0 + tile) * 1024u + 4u * tid;
    n01[t] = 0;
    if (tile < d.cls_ntiles) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t e = e0 + j;
        if (e >= eb && e < ee) {
          uint32_t c2;
          const int cl = classify_entry(d.cflag, sparse_iter, pp[t][j], qq[t][j], &c2);
          cc[t][j] = (uint32_t)cl | (c2 << 2);
          n01[t] += (cl == 0 ? 1u : 0u) + (cl == 1 ? 0x10000u : 0u);
          if (qq[t][j] > pp[t][j]) up[cl]++;
          if (cl == 2) atomicAdd(&sh.hist[c2 & 63u], 1u);
        }
      }
    }
  }
  // exclusive scans over the worker; a tile's total is its count
#pragma unroll
  for (int t = 0; t < T; ++t) {
    uint32_t inc = n01[t];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
      if (lane >= o) inc += v;
    }
    __syncthreads();
    if (lane == 63) sh.wsum[wv] = inc;
    __syncthreads();
    uint32_t woff = 0, ttot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t v = sh.wsum[w];
      if (w < wv) woff += v;
      ttot += v;
    }
    excl[t] = woff + inc - n01[t];
    const uint32_t tile = rw + t * nrw;
    // (inject_fault: tile 0 is never published, so every worker above it runs into the bound of its wait -- the test
    //  of the time-out path: the run freezes and surfaces as SVILS_ERR_DEVICE)
#ifdef SVILS_TESTING
    const bool withheld = d.inject_fault && tile == 0u;
#else
    constexpr bool withheld = false;
#endif
    if (tile < d.cls_ntiles && tid == 0 && !withheld)
      st_agent(&d.tpoll[tile], ((unsigned long long)epoch << 32) | (unsigned long long)ttot);   // n0 | n1 << 16, <= 1024 each
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint32_t u = up[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) u += (uint32_t)__shfl_xor((int)u, o, 64);
    if (lane == 0 && u) atomicAdd(&sh.upper[c], u);
  }
  __syncthreads();
  if (threadIdx.x < 64 + 3) {   // one integer atomic per block and word, nobody waits for them
    uint32_t v = 0;
#pragma unroll
    for (int w = 0; w < NWORK; ++w) v += threadIdx.x < 64 ? shw[w].hist[threadIdx.x] : shw[w].upper[threadIdx.x - 64];
    if (v) {
      if (threadIdx.x < 64) { if (threadIdx.x < geo.K) atomicAdd(&shist[threadIdx.x], (unsigned long long)v); }
      else atomicAdd(&ltot[3 + (threadIdx.x - 64)], v);
    }
  }
  // ---- hand-off: class-0 / class-1 entries in the tiles below this worker's tiles ----
  unsigned long long pre[T];
  {
    uint32_t acc0[T], acc1[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { acc0[t] = 0; acc1[t] = 0; }
    const uint32_t top = rw + (T - 1) * nrw < d.cls_ntiles ? rw + (T - 1) * nrw : (rw < d.cls_ntiles ? rw : 0u);
    for (uint32_t i = tid; i < top; i += 256u) {
      unsigned long long v;
      uint32_t spins = 0;
      while ((uint32_t)((v = ld_agent(&d.tpoll[i])) >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 20)) { d.ctrl->fault = 1u; d.ctrl->stopped = 1; break; }   // never spin unbounded: flag the error
      }
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (i < rw + t * nrw) { acc0[t] += (uint32_t)v & 0xffffu; acc1[t] += ((uint32_t)v >> 16) & 0xffffu; }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      unsigned long long w = ((unsigned long long)acc0[t] << 32) | acc1[t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) w += (unsigned long long)__shfl_xor((long long)w, o, 64);
      __syncthreads();
      if (lane == 0) sh.wred[wv] = w;
      __syncthreads();
      pre[t] = sh.wred[0] + sh.wred[1] + sh.wred[2] + sh.wred[3];
    }
  }
  (void)scan_lds; (void)flag_lds;
  // (no block barrier below: workers and tiles finish on their own)
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint32_t tile = rw + t * nrw;
    if (tile >= d.cls_ntiles) continue;
    const uint64_t e0 = (uint64_t)(d.cls_tile0 + tile) * 1024u + 4u * tid;
    uint32_t pos0 = (uint32_t)(pre[t] >> 32) + (excl[t] & 0xffffu), pos1 = (uint32_t)pre[t] + (excl[t] >> 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t e = e0 + j;
      const uint32_t cl = cc[t][j] & 3u;
      if (cl != 3u) {
        const uint32_t p = pp[t][j], q = qq[t][j];
        const uint32_t pos2 = (uint32_t)(e - eb) - pos0 - pos1;
        const uint32_t prev = j == 0 ? pprev[t] : pp[t][j - 1];
        if (e == eb || p != prev) {
          for (uint32_t x = (e == eb) ? geo.node_begin : prev + 1u; x <= p; ++x) {
            d.npos[0][x] = pos0; d.npos[1][x] = pos1; d.npos[2][x] = pos2;
          }
        }
        uint32_t a0 = pos0, a1 = pos1, a2 = pos2;
        if (cl == 0u) { d.cp[0][pos0] = p; d.cq[0][pos0] = q; a0 = ++pos0; }
        else if (cl == 1u) { d.cp[1][pos1] = p; d.cq[1][pos1] = q; a1 = ++pos1; }
        else { d.scol[pos2] = (uint16_t)(cc[t][j] >> 2); a2 = pos2 + 1u; }
        if (e == ee - 1) {
          ltot[0] = a0; ltot[1] = a1; ltot[2] = a2;
          for (uint32_t x = p + 1u; x <= geo.node_end; ++x) {
            d.npos[0][x] = a0; d.npos[1][x] = a1; d.npos[2][x] = a2;
          }
        }
      }
    }
  }
}

}  // namespace svils
