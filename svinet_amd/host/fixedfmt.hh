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
// writes "%.Df<sep>" of v at p and returns the end; the caller guarantees 400 bytes of room (the snprintf path of a huge
// value: 309 digits, the point, D decimals, the separator)
template <int D>
inline char *fmt_fixed(char *p, double v, char sep) {
  static_assert(D == 3 || D == 5, "scales 1e3 and 1e5 only");
  constexpr double S = D == 5 ? 1e5 : 1e3;
  constexpr uint64_t P = D == 5 ? 100000ull : 1000ull;
  const double y = v * S;
  if (!std::signbit(v) && y < 17179869184.0 /* 2^34 */) {
    const double fl = std::floor(y), fr = y - fl;
    if (std::fabs(fr - 0.5) > 4e-6) {
      const uint64_t r = (uint64_t)fl + (fr > 0.5 ? 1u : 0u);
      uint64_t q = r / P;
      const uint32_t fq = (uint32_t)(r % P);
      if (q < 10) *p++ = (char)('0' + q);           // the common case: values below 10
      else {
        char t[24];
        int n = 0;
        do { t[n++] = (char)('0' + q % 10); q /= 10; } while (q);
        while (n) *p++ = t[--n];
      }
      *p++ = '.';
      // the D fraction digits, two at a time from a table (one single digit first)
      static const char *const dd =
          "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
          "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
      if (D == 5) {
        const uint32_t lo = fq % 100, mid = (fq / 100) % 100, hi = fq / 10000;
        p[0] = (char)('0' + hi);
        p[1] = dd[2 * mid]; p[2] = dd[2 * mid + 1];
        p[3] = dd[2 * lo]; p[4] = dd[2 * lo + 1];
        p += 5;
      } else {
        const uint32_t lo = fq % 100, hi = fq / 100;
        p[0] = (char)('0' + hi);
        p[1] = dd[2 * lo]; p[2] = dd[2 * lo + 1];
        p += 3;
      }
      *p++ = sep;
      return p;
    }
  }
  return p + snprintf(p, 400, D == 5 ? "%.5f%c" : "%.3f%c", v, sep);
}
template <int D>
inline void append_fixed(std::string &o, double v, char sep) {
  char tmp[400];
  o.append(tmp, (size_t)(fmt_fixed<D>(tmp, v, sep) - tmp));
}
// "%ld<sep>"
inline char *fmt_int(char *p, long v, char sep) {
  unsigned long u = v < 0 ? 0ul - (unsigned long)v : (unsigned long)v;
  char t[24];
  int n = 0;
  do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) *p++ = '-';
  while (n) *p++ = t[--n];
  *p++ = sep;
  return p;
}
// The text of a block of rows: numbers are formatted into a scratch the writing thread owns (its cursor lives in a
// register) and reach the block's string 60 KB at a time.  Appending every number to the std::string itself cost 7.6 ns per
// number on one thread and 28 - 54 ns on 4 - 16: the strings of the threads sit next to each other in one vector and
// every append stores its size field -- two threads per cache line (profiles/archive/r05h_fmt_bench_before.txt).
class RowOut {
 public:
  explicit RowOut(std::string &o) : o_(o), p_(buf_) {}
  ~RowOut() { flush(); }
  template <int D> void fixed(double v, char sep) { room(); p_ = fmt_fixed<D>(p_, v, sep); }
  void integer(long v, char sep) { room(); p_ = fmt_int(p_, v, sep); }
  void flush() { o_.append(buf_, (size_t)(p_ - buf_)); p_ = buf_; }
 private:
  void room() { if (buf_ + sizeof buf_ - p_ < 400) flush(); }
  std::string &o_;
  char *p_;
  char buf_[61440];
};
}  // namespace svinet
