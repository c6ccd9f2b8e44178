// micro-benchmark: HBM rate of random 4 KiB row gathers (K = 512 doubles per row, one row per wavefront
// step, 16 B per lane and instruction) against the number of rows a wave keeps in flight and the
// waves per SIMD.  Table of 2^20 rows (4 GiB), far beyond the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROWD = 512;   // doubles per row
template <int DEPTH, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const double *__restrict__ tab, const uint32_t *__restrict__ idx, double *out, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t *my = idx + (size_t)gw * rows_per_wave;
  double2 buf[DEPTH][4];
  double acc = 0.0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const double *r = tab + (size_t)my[d] * ROWD;
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[d][j] = *reinterpret_cast<const double2 *>(r + 2 * (j * 64 + lane));
  }
  for (int i = 0; i < rows_per_wave; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      double2 cur[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[j] = buf[d][j];
      if (i + DEPTH + d < rows_per_wave) {
        const double *r = tab + (size_t)my[i + DEPTH + d] * ROWD;
#pragma unroll
        for (int j = 0; j < 4; ++j) buf[d][j] = *reinterpret_cast<const double2 *>(r + 2 * (j * 64 + lane));
      }
      // ~250 VALU instructions of dependent-ish work per row, like the phi pass
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { t += cur[j].x; t += cur[j].y; }
#pragma unroll
      for (int u = 0; u < 30; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { cur[j].x = fma(cur[j].x, 0.999, t); cur[j].y = fma(cur[j].y, 1.001, -t); }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += cur[j].x + cur[j].y;
    }
  }
  out[(size_t)gw * 64 + lane] = acc;
}
template <int DEPTH, int OCC>
int run(const double *dt, const uint32_t *di, double *dout, int rpw) {
  const int blocks = 256 * OCC;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<DEPTH, OCC>), dim3(blocks), dim3(256), 0, 0, dt, di, dout, rpw);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes = (double)blocks * 4 * rpw * ROWD * 8;
  printf("depth %d, %d waves/SIMD: %.3f ms, %.2f TB/s\n", DEPTH, OCC, best, bytes / best * 1e-9);
  return 0;
}
int main() {
  const size_t nrows = 1u << 20;
  const int rpw = 600;   // rows per wave
  double *dt, *dout; uint32_t *di;
  CHK(hipMalloc(&dt, nrows * ROWD * 8));
  CHK(hipMemset(dt, 0, nrows * ROWD * 8));
  std::vector<uint32_t> idx((size_t)256 * 4 * 4 * rpw);
  std::mt19937 rng(7);
  for (auto &x : idx) x = rng() & (nrows - 1);
  CHK(hipMalloc(&di, idx.size() * 4)); CHK(hipMemcpy(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMalloc(&dout, (size_t)256 * 4 * 256 * 8));
  if (run<1, 3>(dt, di, dout, rpw)) return 1;
  if (run<2, 3>(dt, di, dout, rpw)) return 1;
  if (run<3, 3>(dt, di, dout, rpw)) return 1;
  if (run<4, 3>(dt, di, dout, rpw)) return 1;
  if (run<1, 4>(dt, di, dout, rpw)) return 1;
  if (run<2, 4>(dt, di, dout, rpw)) return 1;
  if (run<3, 4>(dt, di, dout, rpw)) return 1;
  if (run<2, 2>(dt, di, dout, rpw)) return 1;
  if (run<4, 2>(dt, di, dout, rpw)) return 1;
  return 0;
}
