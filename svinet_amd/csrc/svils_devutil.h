// svils_devutil.h -- device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "svils_internal.h"

namespace svils {

#define NEG_INF (-__builtin_huge_val())

// ------------------------------------------------------------ profiling stamps
// STAMP(kernel, slot): thread 0 of the block records the 100 MHz wall clock.  Compiled in only with
// -DSVILS_STAMPS (tools/stamps.py); the product build carries none of it.
#ifdef SVILS_STAMPS
#define STAMP(KERNEL, SLOT)                                                                         \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x < 1024)                                                      \
      d.stamps[((size_t)(KERNEL) * 1024 + blockIdx.x) * 8 + (SLOT)] = wall_clock64();               \
  } while (0)
#else
#define STAMP(KERNEL, SLOT) do { } while (0)
#endif

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
// row_bcast15 / row_bcast31 under a row mask: the masked-out rows receive `ident`
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_rows_f64(double x, double ident) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(x), CTRL, ROWMASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(x), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// W == 64: after the four steps inside the 16-lane rows every lane holds its row's partial; rows 1 and 3 then
// take lane 15 of the row below (row_bcast15), rows 2 and 3 take lane 31 (row_bcast31), and lane 63's total comes
// back to everybody through two v_readlane -- no LDS round trip (ds_swizzle / ds_bpermute cost ~100 cycles each on
// the dependent chain of every neighbour row), and the result is wave-uniform.
#define SVILS_GROUP_REDUCE(NAME, OP, IDENT)                                     \
  template <int W>                                                              \
  __device__ __forceinline__ double NAME(double x) {                            \
    if (W >= 2) { const double t = dpp_f64<0xB1>(x); x = OP(x, t); }            \
    if (W >= 4) { const double t = dpp_f64<0x4E>(x); x = OP(x, t); }            \
    if (W >= 8) { const double t = dpp_f64<0x141>(x); x = OP(x, t); }           \
    if (W >= 16) { const double t = dpp_f64<0x140>(x); x = OP(x, t); }          \
    if constexpr (W == 64 && SVILS_ROW_BCAST) {                                 \
      { const double t = dpp_rows_f64<0x142, 0xa>(x, IDENT); x = OP(x, t); }    \
      { const double t = dpp_rows_f64<0x143, 0xc>(x, IDENT); x = OP(x, t); }    \
      return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), \
                              __builtin_amdgcn_readlane(__double2loint(x), 63)); \
    }                                                                           \
    if (W >= 32) { const double t = swz16_f64(x); x = OP(x, t); }               \
    if (W >= 64) { const double t = __shfl_xor(x, 32, 64); x = OP(x, t); }      \
    return x;                                                                   \
  }
#ifndef SVILS_ROW_BCAST
#define SVILS_ROW_BCAST 1
#endif
__device__ __forceinline__ double svils_add(double a, double b) { return a + b; }
__device__ __forceinline__ double svils_max(double a, double b) { return fmax(a, b); }
SVILS_GROUP_REDUCE(group_sum, svils_add, 0.0)
SVILS_GROUP_REDUCE(group_max, svils_max, NEG_INF)
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

// one v_max_f64 (fmax() adds two canonicalising v_max per call; the inputs here are never NaN)
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
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

// N independent exp_neg chains written step-by-step so the Horner recurrences interleave
// (a single chain is 15 dependent fp64 ops; hipcc schedules separate calls back to back)
template <int N>
__device__ __forceinline__ void exp_neg_n(double (&x)[N]) {
  double n[N], r[N], p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { x[i] = fmax(x[i], -750.0); n[i] = rint(x[i] * 1.4426950408889634); }
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = fma(n[i], -6.93147180369123816490e-01, x[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = fma(n[i], -1.90821492927058770002e-10, r[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = fma(2.51100376059637769e-08, r[i], 2.76326396390410286e-07);
#define SVILS_EXP_STEP(C)              \
  _Pragma("unroll") for (int i = 0; i < N; ++i) p[i] = fma(p[i], r[i], C);
  SVILS_EXP_STEP(2.75572409185789696e-06)
  SVILS_EXP_STEP(2.48014854823284939e-05)
  SVILS_EXP_STEP(1.98412698900471131e-04)
  SVILS_EXP_STEP(1.38888889523147751e-03)
  SVILS_EXP_STEP(8.33333333331960115e-03)
  SVILS_EXP_STEP(4.16666666664880989e-02)
  SVILS_EXP_STEP(1.66666666666666796e-01)
  SVILS_EXP_STEP(5.00000000000001887e-01)
  SVILS_EXP_STEP(1.0)
  SVILS_EXP_STEP(1.0)
#undef SVILS_EXP_STEP
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = ldexp(p[i], (int)n[i]);
}

// ----------------------------------------------------------------- 1/x and ln(y)
// v_rcp_f64 seed + two Newton steps (5 instructions, <= 1-2 ulp) instead of the IEEE
// division sequence (~25 instructions: div_scale x2, rcp, 5 fma, div_fmas, div_fixup)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// ln(y) for normal y > 0 (digamma only calls it with y >= 10): split y = 2^e * m, m in [1,2);
// 128-entry table {1/c_i, ln c_i} at the interval centres, r = m/c_i - 1 (|r| < 2^-8), degree-7
// log1p series.  ~20 instructions against ~150 for libm's extended-precision log; 1 ulp
// (2.2e-16 max relative error vs mpmath on [10, 1e12]).  `tab` lives in LDS (load_logtab).
__device__ __forceinline__ double log_tab(double y, const double2 *tab) {
  const int hi = __double2hiint(y), lo = __double2loint(y);
  const int e = (hi >> 20) - 1023;
  const int i = (hi >> 13) & 127;
  const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
  const double2 t = tab[i];
  const double r = fma(m, t.x, -1.0);
  double p = 1.0 / 7.0;
  p = fma(p, r, -1.0 / 6.0);
  p = fma(p, r, 0.2);
  p = fma(p, r, -0.25);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  p = fma(p, r, 1.0);
  const double ed = (double)e;
  return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, fma(p, r, t.y)));
}
// copy the 2 KiB table from global memory into LDS (call once per block, then __syncthreads())
__device__ __forceinline__ void load_logtab(double2 *lds_tab, const double *gtab) {
  for (int i = threadIdx.x; i < 128; i += blockDim.x) lds_tab[i] = make_double2(gtab[2 * i], gtab[2 * i + 1]);
}

// ----------------------------------------------------------------- digamma
// psi(x), x > 0, double-accurate (stands where the reference calls gsl_sf_psi,
// src/linksampling.hh:181,184).  x < 10 is shifted by 10 with ONE reciprocal:
// sum_{i<10} 1/(x+i) = Q'(x)/Q(x), Q = prod (x+i); then the asymptotic series at
// y = x+10 >= 10 (error < 4e-17).  ~65 instructions; max error vs mpmath 2e-15 relative.
__device__ __forceinline__ double digamma(double x, const double2 *tab) {
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
    const double r = fast_rcp(Q * y);
    shift = Qd * y * r;
    xi = Q * r;
  } else {
    xi = fast_rcp(y);
  }
  const double xi2 = xi * xi;
  const double ser =
      xi2 * (1.0 / 12.0 -
             xi2 * (1.0 / 120.0 -
                    xi2 * (1.0 / 252.0 -
                           xi2 * (1.0 / 240.0 -
                                  xi2 * (1.0 / 132.0 - xi2 * (691.0 / 32760.0 - xi2 * (1.0 / 12.0)))))));
  return log_tab(y, tab) - 0.5 * xi - ser - shift;
}

// ------------------------------------------------ cross-workgroup hand-off inside one launch
// 8-byte agent-scope relaxed atomics on both sides (global_store/global_load ... sc1): the
// stores are written through, the loads bypass the CU's L1; no L2 write-back or invalidate is
// needed (MI355X_MICROARCH.md, "Valid forms": {8-B agent atomics both sides}).
__device__ __forceinline__ void st_agent(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Arrival ticket: every thread of the block has issued its st_agent() stores; returns true in
// exactly one block of the grid -- the last one to arrive -- after which that block may ld_agent()
// what the others published.  The counter is reset by the last arriver (the launch is over for
// everybody else), so the same word serves every launch.
__device__ __forceinline__ bool last_block_arrives(uint32_t *ticket, uint32_t nblocks, uint32_t *lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t last = (t == nblocks - 1u) ? 1u : 0u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0u;
}

// ------------------------------------------ fixed-point accumulation of block partials (integer atomics: order-free)
// A double s with |s| * scale < 2^61 (scale a power of two) is split into hi = rint(s * scale) and
// lo = rint((s * scale - hi) * 2^40): both exact except for bits of s below 2^-40 / scale, so the sum of the pairs over
// any number of blocks, in any order, is the correctly rounded sum of the block partials to ~1e-12 ulp-of-scale --
// tiny columns (a dead community's sum[k]) keep their relative accuracy.  |lo| <= 2^39 per block.
constexpr double SVILS_FX_LO = 1099511627776.0;   // 2^40
__device__ __forceinline__ void fx_add(long long *hi, long long *lo, double s, double scale) {
  const double t = s * scale, q = rint(t);
  const long long qh = (long long)q, ql = (long long)rint((t - q) * SVILS_FX_LO);
  if (qh) __hip_atomic_fetch_add(hi, qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ql) __hip_atomic_fetch_add(lo, ql, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fx_value(long long hi, long long lo, double inv) {
  return ((double)hi + (double)lo * (1.0 / SVILS_FX_LO)) * inv;
}

// ------------------------------------------ column sums of a few per-block partial rows
// out[c] = sum_r part[r][c], r < nrows (<= SVILS_FOLD_ROWS), c < ncols <= CW, by the whole block in
// a fixed order (row groups of NT/CW, NT = blockDim.x, then the groups in order): every block that folds the
// same rows gets the same bits.  `tmp` holds NT doubles.  Ends with __syncthreads().
template <int CW, int NT>
struct FoldRows {
  static constexpr uint32_t R = NT / CW;
  static constexpr uint32_t MAXL = (SVILS_FOLD_ROWS + R - 1) / R;   // loads per thread, all in flight at once
  double v[MAXL];
  // first half: only issues the loads, so that the caller can start other independent accesses
  // before anything waits (every dependent access after a kernel boundary is a cold miss)
  __device__ __forceinline__ void issue(const double *__restrict__ part, uint32_t nrows, uint32_t ncols) {
    const uint32_t c = threadIdx.x % CW, r0 = threadIdx.x / CW;
#pragma unroll
    for (uint32_t i = 0; i < MAXL; ++i) {
      const uint32_t r = r0 + i * R;
      v[i] = (c < ncols && r < nrows) ? part[(size_t)r * ncols + c] : 0.0;
    }
  }
  __device__ __forceinline__ void finish(double *tmp, double *out) {
    const uint32_t c = threadIdx.x % CW, r0 = threadIdx.x / CW;
    double s = 0.0;
#pragma unroll
    for (uint32_t i = 0; i < MAXL; ++i) s += v[i];
    tmp[r0 * CW + c] = s;
    __syncthreads();
    if (threadIdx.x < CW) {
      double w[R];
#pragma unroll
      for (uint32_t i = 0; i < R; ++i) w[i] = tmp[i * CW + threadIdx.x];
      double t = 0.0;
#pragma unroll
      for (uint32_t i = 0; i < R; ++i) t += w[i];
      out[threadIdx.x] = t;
    }
    __syncthreads();
  }
};
template <int CW, int NT>
__device__ __forceinline__ void fold_rows(const double *__restrict__ part, uint32_t nrows, uint32_t ncols,
                                          double *tmp, double *out) {
  FoldRows<CW, NT> f;
  f.issue(part, nrows, ncols);
  f.finish(tmp, out);
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
