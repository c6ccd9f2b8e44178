// "%.Df" (D = 3 or 5) of a double, byte for byte what printf writes: the number scaled by 10^D and rounded in integer
// arithmetic where that is provably printf's own rounding, snprintf otherwise.  printf rounds the EXACT binary value to D
// decimals; y = v * 10^D carries at most half an ulp of error (< 2e-6 for y < 2^34), so the nearest integer to y is the
// nearest integer to the exact product unless y lies within 4e-6 of a half-integer -- those, negative numbers, -0.0,
// large values, infinities and NaNs take the snprintf path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
namespace svinet {
template <int D>
inline void append_fixed(std::string &o, double v, char sep) {
  static_assert(D == 3 || D == 5, "scales 1e3 and 1e5 only");
  constexpr double S = D == 5 ? 1e5 : 1e3;
  constexpr uint64_t P = D == 5 ? 100000ull : 1000ull;
  if (!std::signbit(v) && v * S < 17179869184.0 /* 2^34 */) {
    const double y = v * S, fl = std::floor(y), fr = y - fl;
    if (std::fabs(fr - 0.5) > 4e-6) {
      const uint64_t r = (uint64_t)fl + (fr > 0.5 ? 1u : 0u);
      uint64_t q = r / P, fq = r % P;
      char tmp[40];
      char *e = tmp + sizeof tmp, *b = e;
      *--b = sep;
      for (int i = 0; i < D; ++i) { *--b = (char)('0' + fq % 10); fq /= 10; }
      *--b = '.';
      do { *--b = (char)('0' + q % 10); q /= 10; } while (q);
      o.append(b, (size_t)(e - b));
      return;
    }
  }
  char tmp[400];
  const int len = snprintf(tmp, sizeof tmp, D == 5 ? "%.5f%c" : "%.3f%c", v, sep);
  o.append(tmp, (size_t)len);
}
}  // namespace svinet
