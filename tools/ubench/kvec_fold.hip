// micro-benchmark: how a launch of nb 256-thread blocks should leave the column sums of its per-block K-vector partials.
//   mode 0  every block stores its row; a second launch (k_colreduce shape: 4 columns x 64 row segments per block) adds them
//   mode 1  fixed-point integer atomics into per-XCD accumulators (hi | lo), arrival ticket, the last block converts
//   mode 2  the same into ONE accumulator
//   mode 3  rows published with agent-scope stores; the last block of every group of 32 folds its group's rows, the last
//           group to finish folds the group rows (two-level tickets, fixed order: no atomics on data)
// The blocks do ~WORK us of dependent arithmetic first, of slightly different length per block, so that arrivals are
// spread the way a real pass spreads them.  hipcc --offload-arch=gfx950 -O3 -o /tmp/kvec_fold tools/ubench/kvec_fold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld_agent(const long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool last_arrives(uint32_t *ticket, uint32_t n, uint32_t *flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t last = t == n - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0u;
}
constexpr double FXLO = 1099511627776.0;

template <int MODE>
__global__ __launch_bounds__(256) void k(double *rows, long long *acc, double *grows, uint32_t *tick, double *out, int K, int Kp, int work,
                                         double scale, double inv) {
  __shared__ double part[2048];
  __shared__ uint32_t flag;
  // dependent arithmetic: ~4 cycles per fma and wave
  double x = 1.0 + 1e-9 * threadIdx.x;
  const int iters = work + (int)((blockIdx.x * 2654435761u) >> 27);   // + 0..31
  for (int i = 0; i < iters * 64; ++i) x = x * 1.0000001 + 1e-12;
  for (int c = threadIdx.x; c < K; c += 256) part[c] = x * (1.0 + c) * 1e-3;
  __syncthreads();
  if (MODE == 0) {
    for (int c = threadIdx.x; c < K; c += 256) rows[(size_t)blockIdx.x * K + c] = part[c];
  } else if (MODE == 1 || MODE == 2) {
    const uint32_t xcc = MODE == 1 ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0u;
    for (int c = threadIdx.x; c < K; c += 256) {
      const double t = part[c] * scale, q = rint(t);
      const long long qh = (long long)q, ql = (long long)rint((t - q) * FXLO);
      long long *w = acc + ((size_t)xcc * 2) * Kp + c;
      if (qh) __hip_atomic_fetch_add(w, qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ql) __hip_atomic_fetch_add(w + Kp, ql, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!last_arrives(tick, gridDim.x, &flag)) return;
    for (int c = threadIdx.x; c < K; c += 256) {
      long long h = 0, l = 0;
      for (int xx = 0; xx < (MODE == 1 ? 8 : 1); ++xx) {
        h += ld_agent(acc + ((size_t)xx * 2) * Kp + c);
        l += ld_agent(acc + ((size_t)xx * 2 + 1) * Kp + c);
      }
      out[c] = ((double)h + (double)l * (1.0 / FXLO)) * inv;
      for (int xx = 0; xx < (MODE == 1 ? 8 : 1); ++xx) { acc[((size_t)xx * 2) * Kp + c] = 0; acc[((size_t)xx * 2 + 1) * Kp + c] = 0; }
    }
  } else {
    for (int c = threadIdx.x; c < K; c += 256) st_agent(&rows[(size_t)blockIdx.x * K + c], part[c]);
    const uint32_t ngroups = (gridDim.x + 31) / 32, grp = blockIdx.x / 32;
    const uint32_t gsize = min(32u, gridDim.x - grp * 32);
    if (!last_arrives(tick + 1 + grp, gsize, &flag)) return;
    for (int c = threadIdx.x; c < K; c += 256) {
      double v[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) v[r] = (uint32_t)r < gsize ? ld_agent(&rows[((size_t)grp * 32 + r) * K + c]) : 0.0;
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 32; ++r) s += v[r];
      st_agent(&grows[(size_t)grp * K + c], s);
    }
    if (!last_arrives(tick, ngroups, &flag)) return;
    for (int c = threadIdx.x; c < K; c += 256) {
      double s = 0.0;
      for (uint32_t g0 = 0; g0 < ngroups; g0 += 16) {
        double v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = g0 + r < ngroups ? ld_agent(&grows[(size_t)(g0 + r) * K + c]) : 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[r];
      }
      out[c] = s;
    }
  }
}

__global__ __launch_bounds__(256) void k_colreduce(const double *part, double *out, uint32_t nb, uint32_t ncols) {
  __shared__ double lds[64][5];
  const int cl = threadIdx.x & 3, seg = threadIdx.x >> 2;
  const uint32_t c = blockIdx.x * 4 + cl;
  double s = 0.0;
  if (c < ncols) {
#pragma unroll 8
    for (uint32_t b = seg; b < nb; b += 64) s += part[(size_t)b * ncols + c];
  }
  lds[seg][cl] = s;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (seg < o) lds[seg][cl] += lds[seg + o][cl];
    __syncthreads();
  }
  if (seg == 0 && c < ncols) out[c] = lds[0][cl];
}
__global__ void k_consume(const double *v, double *o, int K) {   // the dependent next launch
  if ((int)threadIdx.x < K) o[threadIdx.x + blockIdx.x] = v[threadIdx.x] + 1.0;
}

template <int MODE>
int run(int nb, int K, int work) {
  const int Kp = (K + 63) / 64 * 64;
  double *rows, *grows, *out, *sink; long long *acc; uint32_t *tick;
  CHK(hipMalloc(&rows, (size_t)nb * K * 8)); CHK(hipMalloc(&grows, (size_t)(nb / 32 + 1) * K * 8));
  CHK(hipMalloc(&out, K * 8)); CHK(hipMalloc(&sink, (K + 4096) * 8));
  CHK(hipMalloc(&acc, (size_t)16 * Kp * 8)); CHK(hipMemset(acc, 0, (size_t)16 * Kp * 8));
  CHK(hipMalloc(&tick, 4096 * 4)); CHK(hipMemset(tick, 0, 4096 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const double scale = 1.0 / 1024.0 * 1e15, inv = 1.0 / scale;
  const int reps = 200;
  float best = 1e30f, tot = 0;
  for (int pass = 0; pass < 3; ++pass) {
    CHK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, rows, acc, grows, tick, out, K, Kp, work, scale, inv);
      if (MODE == 0) hipLaunchKernelGGL(k_colreduce, dim3((K + 3) / 4), dim3(256), 0, 0, rows, out, (uint32_t)nb, (uint32_t)K);
      hipLaunchKernelGGL(k_consume, dim3(256), dim3(256), 0, 0, out, sink, K);
    }
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) { tot += ms; if (ms < best) best = ms; }
  }
  std::vector<double> h(K);
  CHK(hipMemcpy(h.data(), out, K * 8, hipMemcpyDeviceToHost));
  printf("mode %d nb=%5d K=%4d work=%3d : %.2f us per (pass + consumer)   out[1]=%.6g\n", MODE, nb, K, work, best / reps * 1e3, h[1]);
  hipFree(rows); hipFree(grows); hipFree(out); hipFree(sink); hipFree(acc); hipFree(tick);
  return 0;
}

int main() {
  const int cfg[][3] = {{1280, 200, 20}, {1280, 200, 100}, {1280, 100, 20}, {768, 512, 20}, {3072, 512, 100}, {512, 2048, 100}};
  for (auto &c : cfg) {
    if (run<0>(c[0], c[1], c[2])) return 1;
    if (run<1>(c[0], c[1], c[2])) return 1;
    if (run<2>(c[0], c[1], c[2])) return 1;
    if (run<3>(c[0], c[1], c[2])) return 1;
  }
  return 0;
}
