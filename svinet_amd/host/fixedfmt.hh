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
      uint64_t q = r / P;
      uint32_t fq = (uint32_t)(r % P);
      char tmp[40];
      char *e = tmp + sizeof tmp, *b = e;
      *--b = sep;
      // the D fraction digits, two at a time from a table (D = 5: one single digit first)
      static const char *const dd =
          "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
          "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
      if (D == 5) {
        const uint32_t lo = fq % 100, mid = (fq / 100) % 100, hi = fq / 10000;
        b -= 2; b[0] = dd[2 * lo]; b[1] = dd[2 * lo + 1];
        b -= 2; b[0] = dd[2 * mid]; b[1] = dd[2 * mid + 1];
        *--b = (char)('0' + hi);
      } else {
        const uint32_t lo = fq % 100, hi = fq / 100;
        b -= 2; b[0] = dd[2 * lo]; b[1] = dd[2 * lo + 1];
        *--b = (char)('0' + hi);
      }
      *--b = '.';
      if (q < 10) *--b = (char)('0' + q);           // the common case: values below 10
      else do { *--b = (char)('0' + q % 10); q /= 10; } while (q);
      // (the caller reserves a block's worth of text: this append never reallocates inside a row)
      o.append(b, (size_t)(e - b));
      return;
    }
  }
  char tmp[400];
  const int len = snprintf(tmp, sizeof tmp, D == 5 ? "%.5f%c" : "%.3f%c", v, sep);
  o.append(tmp, (size_t)len);
}
}  // namespace svinet
