// svils_devutil.h -- device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "svils_internal.h"

namespace svils {

#define NEG_INF (-__builtin_huge_val())

// ---------------------------------------------------------------- lane maps
template <int W, int V>
__device__ __forceinline__ int kmap(int lw, int v) {
  return V == 1 ? lw : 2 * ((v >> 1) * W + lw) + (v & 1);
}

// ------------------------------------------------------- group reductions
// All-reduce over the W lanes of a group.  Steps 1,2,4,8 are DPP moves inside a
// 16-lane row (quad_perm / row_half_mirror / row_mirror: ~8 cycles each instead of a
// ~100-cycle ds_bpermute round trip); steps 16 and 32 go through ds_swizzle /
// ds_bpermute.  The mirrors work as xor-4 / xor-8 because after the previous steps
// every lane of a quad (8-group) already holds the same partial.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swz16_f64(double x) {   // lane i <-> i ^ 16
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
  return __hiloint2double(hi, lo);
}
#define SVILS_GROUP_REDUCE(NAME, OP)                                            \
  template <int W>                                                              \
  __device__ __forceinline__ double NAME(double x) {                            \
    if (W >= 2) { const double t = dpp_f64<0xB1>(x); x = OP(x, t); }            \
    if (W >= 4) { const double t = dpp_f64<0x4E>(x); x = OP(x, t); }            \
    if (W >= 8) { const double t = dpp_f64<0x141>(x); x = OP(x, t); }           \
    if (W >= 16) { const double t = dpp_f64<0x140>(x); x = OP(x, t); }          \
    if (W >= 32) { const double t = swz16_f64(x); x = OP(x, t); }               \
    if (W >= 64) { const double t = __shfl_xor(x, 32, 64); x = OP(x, t); }      \
    return x;                                                                   \
  }
__device__ __forceinline__ double svils_add(double a, double b) { return a + b; }
__device__ __forceinline__ double svils_max(double a, double b) { return fmax(a, b); }
SVILS_GROUP_REDUCE(group_sum, svils_add)
SVILS_GROUP_REDUCE(group_max, svils_max)
// sum across the 64/W groups of a wavefront (lane lw of every group ends with the total)
template <int W>
__device__ __forceinline__ double cross_group_sum(double x) {
#pragma unroll
  for (int o = W; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
  return x;
}
template <int W>
__device__ __forceinline__ uint32_t cross_group_sum_u32(uint32_t x) {
#pragma unroll
  for (int o = W; o < 64; o <<= 1) x += __shfl_xor((int)x, o, 64);
  return x;
}

// --------------------------------------------------------------- row loads
template <int W, int V>
__device__ __forceinline__ void load_row(const double *__restrict__ row, int lw, uint32_t ld,
                                         double (&x)[V]) {
  if constexpr (V == 1) {
    x[0] = (uint32_t)lw < ld ? row[lw] : 0.0;
  } else {
#pragma unroll
    for (int j = 0; j < V / 2; ++j) {
      const uint32_t k0 = 2u * (uint32_t)(j * W + lw);
      double2 t = make_double2(0.0, 0.0);
      if (k0 < ld) t = *reinterpret_cast<const double2 *>(row + k0);
      x[2 * j] = t.x;
      x[2 * j + 1] = t.y;
    }
  }
}
template <int W, int V>
__device__ __forceinline__ void store_row(double *__restrict__ row, int lw, uint32_t ld,
                                          const double (&x)[V]) {
  if constexpr (V == 1) {
    if ((uint32_t)lw < ld) row[lw] = x[0];
  } else {
#pragma unroll
    for (int j = 0; j < V / 2; ++j) {
      const uint32_t k0 = 2u * (uint32_t)(j * W + lw);
      if (k0 < ld) *reinterpret_cast<double2 *>(row + k0) = make_double2(x[2 * j], x[2 * j + 1]);
    }
  }
}

// ------------------------------------------------------------------- exp(t), t <= 0
// The softmax only ever needs exp(x - max) with a non-positive argument.  18
// instructions instead of libm's ~40: clamp, n = rint(t*log2e), two-step Cody-Waite
// reduction, degree-11 near-minimax polynomial on [-ln2/2, ln2/2] (Chebyshev
// interpolant, 1 ulp measured against libm over [-745, 0]), ldexp.  exp_neg(0) == 1
// exactly, exp_neg(-inf) == 0, results below 2^-1022 flush through ldexp.
__device__ __forceinline__ double exp_neg(double t) {
  t = fmax(t, -750.0);
  const double n = rint(t * 1.4426950408889634);
  double r = fma(n, -6.93147180369123816490e-01, t);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 2.51100376059637769e-08;
  p = fma(p, r, 2.76326396390410286e-07);
  p = fma(p, r, 2.75572409185789696e-06);
  p = fma(p, r, 2.48014854823284939e-05);
  p = fma(p, r, 1.98412698900471131e-04);
  p = fma(p, r, 1.38888889523147751e-03);
  p = fma(p, r, 8.33333333331960115e-03);
  p = fma(p, r, 4.16666666664880989e-02);
  p = fma(p, r, 1.66666666666666796e-01);
  p = fma(p, r, 5.00000000000001887e-01);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// ----------------------------------------------------------------- digamma
// psi(x), x > 0, double-accurate (stands where the reference calls gsl_sf_psi,
// src/linksampling.hh:181,184).  x < 10 is shifted by 10 with ONE division:
// sum_{i<10} 1/(x+i) = Q'(x)/Q(x), Q = prod (x+i); then the asymptotic series at
// y = x+10 >= 10 (error < 4e-17).  Max error vs mpmath 1.5e-15 relative.
__device__ __forceinline__ double digamma(double x) {
  double shift = 0.0, y = x, xi;
  if (x < 10.0) {
    double Q = x, Qd = 1.0;
#pragma unroll
    for (int i = 1; i < 10; ++i) {
      const double t = x + (double)i;
      Qd = fma(Qd, t, Q);
      Q *= t;
    }
    y = x + 10.0;
    const double r = 1.0 / (Q * y);
    shift = Qd * y * r;
    xi = Q * r;
  } else {
    xi = 1.0 / y;
  }
  const double xi2 = xi * xi;
  const double ser =
      xi2 * (1.0 / 12.0 -
             xi2 * (1.0 / 120.0 -
                    xi2 * (1.0 / 252.0 -
                           xi2 * (1.0 / 240.0 -
                                  xi2 * (1.0 / 132.0 - xi2 * (691.0 / 32760.0 - xi2 * (1.0 / 12.0)))))));
  return log(y) - 0.5 * xi - ser - shift;
}

// per-block link statistics without atomics: every wave's counts go through LDS,
// thread 0 writes the block's three totals (summed by k_tail).  A same-address
// atomicAdd per wave costs ~11 ns each on MI355X and serialises thousands of waves.
__device__ __forceinline__ void block_store_link_counts(unsigned long long n_dense, unsigned long long n_sparse,
                                                        unsigned long long n_short, unsigned long long *out,
                                                        unsigned long long *lds /*[3][nwaves]*/, int nwaves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n_dense += __shfl_xor((long long)n_dense, o, 64);
    n_sparse += __shfl_xor((long long)n_sparse, o, 64);
    n_short += __shfl_xor((long long)n_short, o, 64);
  }
  if (lane == 0) {
    lds[0 * nwaves + wave] = n_dense;
    lds[1 * nwaves + wave] = n_sparse;
    lds[2 * nwaves + wave] = n_short;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long t = 0;
    for (int w = 0; w < nwaves; ++w) t += lds[threadIdx.x * nwaves + w];
    out[(size_t)blockIdx.x * 3 + threadIdx.x] = t;
  }
}

}  // namespace svils
