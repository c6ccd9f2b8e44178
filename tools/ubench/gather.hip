// micro-benchmark: how fast can a CU gather 160-byte rows (K = 20 doubles, 256-byte row stride)?
//  mode 0: lane-per-row, 10 x global_load_dwordx4 per lane (each instruction touches 64 rows)
//  mode 1: 10 lanes per row, plain loads to VGPRs (each instruction touches 6.4 rows, contiguous 160 B each)
//  mode 2: as 1 but global_load_lds_dwordx4 (no VGPR round trip), then per-lane ds_read_b128 of the own row
//  mode 3: as 0 but every lane of a wave reads the same 4 rows (duplicates)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int LD = 32, KC = 10;
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const double *__restrict__ tab, const uint32_t *__restrict__ idx, double *out, int items_per_wave) {
  __shared__ __attribute__((aligned(16))) double lds[4][64 * 2 * KC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  double acc = 0.0;
  uint32_t qn = idx[(size_t)gw * 64 + lane];
  for (int i = 0; i < items_per_wave; ++i) {
    const uint32_t q = qn;
    if (i + 1 < items_per_wave) qn = idx[(size_t)(gw + (i + 1) * nw) * 64 + lane];
    if (MODE == 4) {
      double2 v[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint32_t qq = __shfl((int)q, c * 4 + (lane >> 4), 64);
        v[c] = *reinterpret_cast<const double2 *>(tab + (size_t)qq * LD + 2 * (lane & 15));
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) acc += v[c].x + v[c].y;
    } else if (MODE == 5) {
      double v[20];
#pragma unroll
      for (int c = 0; c < 20; ++c) {
        const int j = c * 64 + lane;          // 8-byte element id 0..1279: row j / 20, element j % 20
        const uint32_t qq = __shfl((int)q, j / 20, 64);
        v[c] = tab[(size_t)qq * LD + (j % 20)];
      }
#pragma unroll
      for (int c = 0; c < 20; ++c) acc += v[c];
    } else if (MODE == 0 || MODE == 3) {
      const double *r = tab + (size_t)(MODE == 3 ? (q & 3u) : q) * LD;
      double2 v[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) v[c] = *reinterpret_cast<const double2 *>(r + 2 * c);
#pragma unroll
      for (int c = 0; c < KC; ++c) acc += v[c].x + v[c].y;
    } else {
      double *my = lds[wave];
      double2 v[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int j = c * 64 + lane;          // chunk id 0..639: row j / 10, chunk j % 10
        const int row = j / KC, ch = j % KC;
        const uint32_t qq = __shfl((int)q, row, 64);
        const double *src = tab + (size_t)qq * LD + 2 * ch;
        if (MODE == 1) v[c] = *reinterpret_cast<const double2 *>(src);
        else __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(my + 2 * 64 * c), 16, 0, 0);
      }
      if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < KC; ++c) *reinterpret_cast<double2 *>(my + 2 * (c * 64 + lane)) = v[c];
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const double2 t = *reinterpret_cast<const double2 *>(my + 2 * (lane * KC + c));
        acc += t.x + t.y;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  out[(size_t)gw * 64 + lane] = acc;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 17903, items = 6126 * 2 * (argc > 2 ? atoi(argv[2]) : 1);   // rows gathered = items * 64 (p and q rows of 6126 wave-items)
  const int blocks = 1024, ipw = (items + blocks * 4 - 1) / (blocks * 4);
  std::vector<double> tab((size_t)n * LD);
  for (size_t i = 0; i < tab.size(); ++i) tab[i] = (double)(i % 97) * 0.01;
  std::vector<uint32_t> idx((size_t)blocks * 4 * ipw * 64);
  std::mt19937 rng(1);
  for (auto &x : idx) x = rng() % n;
  double *dt, *dout; uint32_t *di;
  CHK(hipMalloc(&dt, tab.size() * 8)); CHK(hipMalloc(&di, idx.size() * 4)); CHK(hipMalloc(&dout, (size_t)blocks * 256 * 8));
  CHK(hipMemcpy(dt, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemcpy(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  std::vector<double> ref, got((size_t)blocks * 256);
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
      CHK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, dt, di, dout, ipw);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, dt, di, dout, ipw);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, dt, di, dout, ipw);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, dt, di, dout, ipw);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, dt, di, dout, ipw);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, dt, di, dout, ipw);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CHK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
    if (mode == 0) ref = got;
    double md = 0; for (size_t i = 0; i < got.size(); ++i) md = fmax(md, fabs(got[i] - ref[i]));
    printf("mode %d: %.2f us for %d row gathers (%d items/wave), %.2f clk/row/CU at 2.4 GHz, maxdiff vs mode0 %.3g\n", mode, best * 1e3,
           blocks * 4 * ipw * 64, ipw, best * 1e-3 * 2.4e9 / (blocks * 4.0 * ipw * 64 / 256), md);
  }
  return 0;
}
