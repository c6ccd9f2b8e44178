// svils_init.hip -- init_gamma2 (src/linksampling.cc:374-401) on the device, bit for bit.
//
// The reference starts gamma at 0 and, for every link (p < q, held-out ones included) in the order p ascending /
// adjacency order, draws K gsl_rng_uniform values from ONE sequential MT19937 stream, divides them by their sum and adds
// the vector to gamma[p] and gamma[q].  At n = 1e6, k = 512 that is 6.1e9 draws and 98 GB of read-modify-write: 2.9 s on
// sixteen host threads (host/linksampling.cc: init_gamma2, jump-ahead per chunk), the largest single piece of the drop-in
// binary's wall time at that size.  Here:
//
//   k_mt_generate   the caller hands over MT19937 states at equally spaced positions of the stream (host/mtjump.hh: one
//                   polynomial for the stride, applied along a chain per host thread); one wavefront per state regenerates
//                   its stretch of RAW 32-bit outputs into device memory (E*K words: 24 GB at config-5 size).  The three
//                   dependency-free loops of the twist (host/rng.hh: refill) become 64-lane steps over the state in LDS.
//   k_init_rows     pull-style like every other pass here: the wavefront of node x walks x's row of the ALL-links CSR --
//                   {p < x ascending} ++ {q > x in adjacency order}: the order in which the reference's loop adds to
//                   gamma[x] -- reads the link's K words (each link is read by both its endpoints), forms u = w / 2^32
//                   (exact), the link's sum as a 64-bit INTEGER (every u is a multiple of 2^-32 below 1, so the
//                   reference's sequential double sum is exact whatever its order) and adds u / sum (IEEE division, the
//                   compiler's correctly rounded sequence) column by column in link order.  No atomics, no scatter.
//
// Then the tail of svils_set_state (lambda, flags, expectations).  Same bits as the host path: tests/test_gpu_init.py.
// Node-block handles (the state is replicated) take the same call; a K-sharded handle draws the whole stream too and keeps
// its column slice of every link's vector, divided by the sum over ALL columns (k_link_sums).
#include "svils_handle.h"

#include <time.h>

namespace svils_impl {

namespace {
constexpr int MT_N = 624, MT_M = 397;

// one wavefront (= one block of 64 threads) per stream
__global__ __launch_bounds__(64) void k_mt_generate(const uint32_t *__restrict__ states, uint64_t nstreams, uint64_t per_stream,
                                                    uint64_t total, uint32_t *__restrict__ out) {
  __shared__ uint32_t x[MT_N];
  const uint64_t s = blockIdx.x;
  if (s >= nstreams) return;
  const int lane = threadIdx.x;
  for (int i = lane; i < MT_N; i += 64) x[i] = states[s * MT_N + i];
  __syncthreads();
  const uint64_t begin = s * per_stream, end = begin + per_stream < total ? begin + per_stream : total;
  for (uint64_t pos = begin; pos < end; pos += MT_N) {
    // x[k] <- x[k + M] ^ twist(x[k], x[k + 1]), k ascending: inside a 64-lane step every lane reads before any lane writes
    // (x[k + 1] must be the OLD word for k < 623 and the NEW x[0] for k = 623; x[k + M - N] the NEW word for k >= N - M: both
    // were written at least one step earlier, 64 < N - M = 227)
    for (int base = 0; base < MT_N; base += 64) {
      const int k = base + lane;
      uint32_t v = 0;
      if (k < MT_N) {
        const uint32_t a = x[k], b = x[k + 1 == MT_N ? 0 : k + 1], c = x[k + MT_M >= MT_N ? k + MT_M - MT_N : k + MT_M];
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        v = c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
      }
      __syncthreads();
      if (k < MT_N) x[k] = v;
      __syncthreads();
    }
    for (int base = 0; base < MT_N; base += 64) {   // tempering, coalesced stores
      const int k = base + lane;
      if (k < MT_N && pos + (uint64_t)k < end) {
        uint32_t y = x[k];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[pos + (uint64_t)k] = y;
      }
    }
  }
}

// the links' sums as 64-bit integers (K-sharded handles: a rank divides its COLUMN SLICE of every link's vector by the sum over
// ALL k_total draws): one wavefront per link
__global__ __launch_bounds__(256) void k_link_sums(uint64_t nedges, uint32_t Kt, const uint32_t *__restrict__ raw,
                                                   unsigned long long *__restrict__ sums) {
  const int lane = threadIdx.x & 63;
  for (uint64_t l = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); l < nedges; l += (uint64_t)gridDim.x * 4) {
    unsigned long long tot = 0;
    for (uint32_t k = (uint32_t)lane; k < Kt; k += 64u) tot += raw[l * Kt + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += (unsigned long long)__shfl_xor((long long)tot, o, 64);
    if (lane == 0) sums[l] = tot;
  }
}

// one wavefront per node; lane t holds columns t, t + 64, ... (J of them) of the handle's K columns, which are the columns
// [K0, K0 + K) of the Kt every link draws (whole-row handles: K0 = 0, K = Kt, and the link's sum is formed here from the words
// the wavefront holds anyway; column slices read it from `sums`)
template <int J>
__global__ __launch_bounds__(256) void k_init_rows(uint32_t n, uint32_t K, uint32_t K0, uint32_t Kt, uint32_t ld,
                                                   const uint64_t *__restrict__ rowptr, const uint32_t *__restrict__ elink,
                                                   const uint32_t *__restrict__ raw, const unsigned long long *__restrict__ sums,
                                                   double *__restrict__ gamma) {
  const int lane = threadIdx.x & 63;
  const uint32_t x = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (x >= n) return;
  double acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = 0.0;
  const uint64_t b = rowptr[x], e = rowptr[x + 1];
  uint32_t wnext[J];
  unsigned long long snext = 0;
  auto fetch = [&](uint64_t ent, uint32_t (&w)[J], unsigned long long &sl) {
    const uint32_t l = elink[ent];
    const uint64_t base = (uint64_t)l * Kt + K0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint32_t k = (uint32_t)lane + 64u * (uint32_t)j;
      w[j] = k < K ? raw[base + k] : 0u;
    }
    if (sums) sl = sums[l];
  };
  if (b < e) fetch(b, wnext, snext);
  for (uint64_t ent = b; ent < e; ++ent) {
    uint32_t w[J];
#pragma unroll
    for (int j = 0; j < J; ++j) w[j] = wnext[j];
    unsigned long long tot = snext;
    if (ent + 1 < e) fetch(ent + 1, wnext, snext);   // the next link's words are in flight while this one is divided
    if (!sums) {
      tot = 0;
#pragma unroll
      for (int j = 0; j < J; ++j) tot += w[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += (unsigned long long)__shfl_xor((long long)tot, o, 64);
    }
    const double s = (double)tot * 2.3283064365386963e-10;   // 2^-32: the sum of the link's uniforms, exact
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const double u = (double)w[j] * 2.3283064365386963e-10;   // gsl_rng_uniform: w / 4294967296.0
      acc[j] += u / s;                                          // phi.normalize(): _data[i] / s (src/matrix.hh:341-346)
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const uint32_t k = (uint32_t)lane + 64u * (uint32_t)j;
    if (k < K) gamma[(size_t)x * ld + k] = acc[j];
  }
}
}  // namespace

}  // namespace svils_impl

extern "C" {

int svils_init_gamma(svils_handle *h, const uint32_t *edges, uint64_t nedges, const uint32_t *mt_states, uint64_t nstreams,
                     uint64_t outputs_per_stream, const double *lambda) {
  NOT_TILED(h, "svils_init_gamma");
  if (!h || (!edges && nedges) || !mt_states || !lambda || nstreams == 0 || outputs_per_stream == 0)
    return fail(SVILS_ERR_ARG, "svils_init_gamma: null argument");
  const Geometry &g = h->geo;
  // (node-block handles hold every row of the state; K-sharded ones their column slice of every row: both are drawn here in full)
  const uint64_t total = nedges * (uint64_t)g.Kt;
  if (nedges >= (1ull << 32)) return fail(SVILS_ERR_UNSUPPORTED, "svils_init_gamma: links are indexed with 32 bits");
  if (nstreams > (1ull << 24) || (total && (nstreams - 1) * outputs_per_stream >= total) || nstreams * outputs_per_stream < total)
    return fail(SVILS_ERR_ARG, "svils_init_gamma: %llu streams of %llu outputs do not cover the %llu x %u uniforms exactly once",
                (unsigned long long)nstreams, (unsigned long long)outputs_per_stream, (unsigned long long)nedges, g.Kt);
  HIPCHK(hipSetDevice(h->cfg.device));
#ifdef SVILS_TESTING
  const auto tclock = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
  const double tt0 = tclock();
  double tt1 = tt0, tt2 = tt0, tt3 = tt0;
#endif
  const uint32_t n = g.n;
  // scratch of this call alone (the raw words are 4 E K bytes: 24 GB at n = 1e6, k = 512): freed before returning
  uint32_t *d_raw = nullptr, *d_states = nullptr, *d_elink = nullptr;
  uint64_t *d_rowptr = nullptr;
  unsigned long long *d_sums = nullptr;
  auto release = [&] {
    if (d_sums) (void)hipFree(d_sums);
    if (d_raw) (void)hipFree(d_raw);
    if (d_states) (void)hipFree(d_states);
    if (d_elink) (void)hipFree(d_elink);
    if (d_rowptr) (void)hipFree(d_rowptr);
  };
  auto chk = [&](hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    release();
    return fail(e == hipErrorOutOfMemory ? SVILS_ERR_NOMEM : SVILS_ERR_DEVICE, "svils_init_gamma: %s failed: %s", what, hipGetErrorString(e));
  };
  int rc;
  // the draws first: they need nothing but the states, and run while the host builds the row lists below
  if ((rc = chk(hipMalloc((void **)&d_raw, std::max<uint64_t>(total, 1) * sizeof(uint32_t) + 512), "hipMalloc (raw MT19937 outputs)"))) return rc;
  if ((rc = chk(hipMalloc((void **)&d_states, nstreams * MT_N * sizeof(uint32_t)), "hipMalloc"))) return rc;
  if ((rc = chk(hipMemcpyAsync(d_states, mt_states, nstreams * MT_N * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream), "upload"))) return rc;
  hipLaunchKernelGGL(k_mt_generate, dim3((uint32_t)nstreams), dim3(64), 0, h->stream, d_states, nstreams, outputs_per_stream, total, d_raw);
  if ((rc = chk(hipMemsetAsync(h->d.gamma, 0, (size_t)g.n_alloc * g.ld * sizeof(double), h->stream), "memset"))) return rc;
#ifdef SVILS_TESTING
  tt1 = tclock();
#endif
  // the ALL-links CSR in the order of the reference's additions: row x = {p < x, ascending} ++ {q > x in link order}
  std::vector<uint64_t> rowptr((size_t)n + 1, 0);
  for (uint64_t l = 0; l < nedges; ++l) {
    const uint32_t p = edges[2 * l], q = edges[2 * l + 1];
    if (p >= q || q >= n) { (void)hipStreamSynchronize(h->stream); release(); return fail(SVILS_ERR_ARG, "svils_init_gamma: link %llu = (%u,%u): need p < q < n", (unsigned long long)l, p, q); }
    if (l && edges[2 * l - 2] > p) { (void)hipStreamSynchronize(h->stream); release(); return fail(SVILS_ERR_ARG, "svils_init_gamma: links must come in the order they are drawn (sorted by first endpoint; link %llu)", (unsigned long long)l); }
    rowptr[p + 1]++;
    rowptr[q + 1]++;
  }
  for (uint32_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
  std::vector<uint32_t> elink(std::max<uint64_t>(2 * nedges, 1));
  {
    std::vector<uint64_t> fill(rowptr.begin(), rowptr.end() - 1);
    for (uint64_t l = 0; l < nedges; ++l) elink[fill[edges[2 * l + 1]]++] = (uint32_t)l;   // lower parts: ascending p
    // upper parts: the links (x, .) are consecutive in link order -- row x's upper part is simply its run of link indices
    uint64_t l = 0;
    for (uint32_t x = 0; x < n && l < nedges; ++x) {
      uint64_t f = fill[x];
      while (l < nedges && edges[2 * l] == x) elink[f++] = (uint32_t)l++;
    }
  }
#ifdef SVILS_TESTING
  tt2 = tclock();
#endif
  if ((rc = chk(hipMalloc((void **)&d_elink, elink.size() * sizeof(uint32_t)), "hipMalloc"))) return rc;
  if ((rc = chk(hipMalloc((void **)&d_rowptr, rowptr.size() * sizeof(uint64_t)), "hipMalloc"))) return rc;
  if ((rc = chk(hipMemcpyAsync(d_elink, elink.data(), elink.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream), "upload"))) return rc;
  if ((rc = chk(hipMemcpyAsync(d_rowptr, rowptr.data(), rowptr.size() * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream), "upload"))) return rc;
  if (g.K != g.Kt) {   // a column slice: the links' sums over all k_total draws first
    if ((rc = chk(hipMalloc((void **)&d_sums, std::max<uint64_t>(nedges, 1) * sizeof(unsigned long long)), "hipMalloc"))) return rc;
    const uint32_t nbs = (uint32_t)std::min<uint64_t>((nedges + 3) / 4 + 1, 1u << 16);
    hipLaunchKernelGGL(k_link_sums, dim3(nbs), dim3(256), 0, h->stream, nedges, g.Kt, d_raw, d_sums);
  }
  const uint32_t nb = (n + 3) / 4;
  const int J = (int)((g.K + 63) / 64);
#define ROWS(J_) hipLaunchKernelGGL((k_init_rows<J_>), dim3(nb), dim3(256), 0, h->stream, n, g.K, g.K0, g.Kt, g.ld, d_rowptr, d_elink, d_raw, d_sums, h->d.gamma)
  if (J <= 1) ROWS(1);
  else if (J <= 2) ROWS(2);
  else if (J <= 4) ROWS(4);
  else if (J <= 8) ROWS(8);
  else if (J <= 16) ROWS(16);
  else ROWS(32);
#undef ROWS
  if ((rc = chk(hipGetLastError(), "launch"))) return rc;
  if ((rc = chk(hipStreamSynchronize(h->stream), "the init kernels"))) return rc;
#ifdef SVILS_TESTING
  tt3 = tclock();
#endif
  release();
  rc = state_arrived(h, lambda, nullptr);
#ifdef SVILS_TESTING
  fprintf(stderr, "[svils_init_gamma] scratch + states + generate launched %.3f s | row lists on the host %.3f s | uploads + kernels %.3f s | free + expectations %.3f s\n",
          tt1 - tt0, tt2 - tt1, tt3 - tt2, tclock() - tt3);
#endif
  return rc;
}

}  // extern "C"
