// mtjump.hh -- jump-ahead for MT19937: the state J outputs further down the stream without drawing them.
//
// Why: init_gamma2 (src/linksampling.cc:374-401) draws K uniforms per link from ONE sequential gsl_rng stream -- 6.1e9
// draws at n = 1e6, k = 512, all on one thread while fifteen others wait for their chunk.  Every link consumes exactly K
// draws, so the state at the start of link l is the seed state advanced by o0 + K l outputs: with a jump the links can be
// drawn by many threads at once and the stream stays the reference's, bit for bit.
//
// How (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number
// generators", 2008): one output step is a linear map F on the 19937-bit state; with phi its characteristic polynomial,
// F^J = g(F) for g = x^J mod phi, a polynomial of degree < 19937, evaluated on a state with Horner's rule (19 936
// single steps and ~10 000 state additions: about a millisecond).  phi is not written down here: it is the minimal
// polynomial of any output bit sequence (it is irreducible of degree 19937), found once per process by Berlekamp-Massey
// over GF(2) on 2 x 19937 bits.  x^J mod phi: square-and-multiply on 312-word bit vectors.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

namespace svinet {
namespace mtjump {

constexpr int N = 624, M = 397, MEXP = 19937;
constexpr int PW = (MEXP + 63) / 64 + 1;   // words of a polynomial of degree <= MEXP (+1 word of slack for shifted copies)

// word-at-a-time form of the generator: the next output is temper(w[p]) AFTER w[p] has been updated
struct WState {
  uint32_t w[N];
  int p;
};
inline void step(WState &s) {
  const int p = s.p, p1 = p + 1 == N ? 0 : p + 1, pm = p + M >= N ? p + M - N : p + M;
  const uint32_t y = (s.w[p] & 0x80000000u) | (s.w[p1] & 0x7fffffffu);
  s.w[p] = s.w[pm] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
  s.p = p1;
}
// a += b as 19937-bit states (the words are aligned at the pointers; the 31 low bits of the word at a pointer are not
// part of the state and may hold anything)
inline void add(WState &a, const WState &b) {
  int ia = a.p, ib = b.p;
  for (int left = N; left > 0;) {
    const int run = std::min(left, std::min(N - ia, N - ib));
    uint32_t *x = a.w + ia;
    const uint32_t *y = b.w + ib;
    for (int i = 0; i < run; ++i) x[i] ^= y[i];
    ia = ia + run == N ? 0 : ia + run;
    ib = ib + run == N ? 0 : ib + run;
    left -= run;
  }
}

struct Poly {
  uint64_t b[2 * PW];   // room for a square before its reduction
  Poly() { memset(b, 0, sizeof b); }
  bool bit(int i) const { return (b[i >> 6] >> (i & 63)) & 1u; }
  void set(int i) { b[i >> 6] |= 1ull << (i & 63); }
};

// phi and its 64 shifted copies (so that a reduction step is a plain xor of word runs)
struct Phi {
  bool ok = false;
  uint64_t sh[64][PW + 1];
};

inline const Phi &phi() {
  static Phi P;
  static std::once_flag once;
  std::call_once(once, [] {
    // an output bit sequence of 2 * MEXP bits from an arbitrary non-zero state
    WState s;
    s.w[0] = 19650218u;
    for (int i = 1; i < N; ++i) s.w[i] = 1812433253u * (s.w[i - 1] ^ (s.w[i - 1] >> 30)) + (uint32_t)i;
    s.p = 0;
    const int len = 2 * MEXP;
    // Berlekamp-Massey over GF(2).  C, B: connection polynomials (bit i = coefficient of x^i); R: the sequence reversed,
    // bit j = s[n - j], so that the discrepancy is the parity of C AND R.
    constexpr int W = (2 * MEXP + 64 + 63) / 64;
    std::vector<uint64_t> C(W, 0), B(W, 0), T(W, 0), R(W, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < len; ++n) {
      const int p = s.p;
      step(s);
      const uint64_t bitn = s.w[p] & 1u;
      // R <<= 1; R[0] = bitn  (only the low L + 1 bits are ever read; keep ceil((L + 2) / 64) words current)
      const int rw = std::min(W, (L + 2) / 64 + 2);
      for (int i = rw - 1; i > 0; --i) R[i] = (R[i] << 1) | (R[i - 1] >> 63);
      R[0] = (R[0] << 1) | bitn;
      uint64_t acc = 0;
      const int cw = L / 64 + 1;
      for (int i = 0; i < cw; ++i) acc ^= C[i] & R[i];
      const int d = __builtin_parityll(acc);
      if (!d) { ++m; continue; }
      const int ws = m >> 6, bs = m & 63;
      const int bw = std::min(W - ws, (n + 2) / 64 + 2);
      if (2 * L <= n) {
        T = C;
        for (int i = 0; i < bw; ++i) {
          C[i + ws] ^= B[i] << bs;
          if (bs && i + ws + 1 < W) C[i + ws + 1] ^= B[i] >> (64 - bs);
        }
        L = n + 1 - L;
        B.swap(T);
        m = 1;
      } else {
        for (int i = 0; i < bw; ++i) {
          C[i + ws] ^= B[i] << bs;
          if (bs && i + ws + 1 < W) C[i + ws + 1] ^= B[i] >> (64 - bs);
        }
        ++m;
      }
    }
    if (L != MEXP) return;   // cannot happen for MT19937; the caller stays sequential if it does
    // characteristic polynomial = reciprocal of the connection polynomial: phi_i = C_{L - i}
    uint64_t f[PW + 1];
    memset(f, 0, sizeof f);
    for (int i = 0; i <= MEXP; ++i)
      if ((C[(MEXP - i) >> 6] >> ((MEXP - i) & 63)) & 1u) f[i >> 6] |= 1ull << (i & 63);
    for (int s2 = 0; s2 < 64; ++s2) {
      memset(P.sh[s2], 0, sizeof P.sh[s2]);
      for (int i = 0; i < PW; ++i) {
        P.sh[s2][i] ^= f[i] << s2;
        if (s2) P.sh[s2][i + 1] ^= f[i] >> (64 - s2);
      }
    }
    P.ok = true;
  });
  return P;
}

// r mod phi, for r of degree < 2 * MEXP
inline void reduce(Poly &r, const Phi &P) {
  for (int i = 2 * MEXP - 1; i >= MEXP; --i) {
    if (!r.bit(i)) continue;
    const int off = i - MEXP, w0 = off >> 6;
    const uint64_t *f = P.sh[off & 63];
    uint64_t *x = r.b + w0;
    for (int j = 0; j <= PW; ++j) x[j] ^= f[j];
  }
}

// g = x^steps mod phi
inline bool power(uint64_t steps, Poly &g) {
  const Phi &P = phi();
  if (!P.ok) return false;
  g = Poly();
  g.set(0);
  static const uint16_t *spread = [] {   // byte -> its bits at the even positions of 16
    static uint16_t t[256];
    for (int v = 0; v < 256; ++v) {
      uint16_t o = 0;
      for (int k = 0; k < 8; ++k) o |= (uint16_t)(((v >> k) & 1) << (2 * k));
      t[v] = o;
    }
    return t;
  }();
  for (int bitno = 63; bitno >= 0; --bitno) {
    // square
    Poly q;
    const int words = (MEXP + 63) / 64;
    for (int i = 0; i < words; ++i) {
      const uint64_t v = g.b[i];
      uint64_t lo = 0, hi = 0;
      for (int k = 0; k < 4; ++k) {
        lo |= (uint64_t)spread[(v >> (8 * k)) & 0xff] << (16 * k);
        hi |= (uint64_t)spread[(v >> (32 + 8 * k)) & 0xff] << (16 * k);
      }
      q.b[2 * i] = lo;
      q.b[2 * i + 1] = hi;
    }
    reduce(q, P);
    g = q;
    if ((steps >> bitno) & 1u) {   // times x
      for (int i = words; i > 0; --i) g.b[i] = (g.b[i] << 1) | (g.b[i - 1] >> 63);
      g.b[0] <<= 1;
      if (g.bit(MEXP)) {
        const uint64_t *f = P.sh[0];
        for (int j = 0; j <= PW; ++j) g.b[j] ^= f[j];
      }
    }
  }
  return true;
}

}  // namespace mtjump

// A jump of a fixed number of outputs, applicable to any number of states.
class MtJump {
 public:
  // false: the polynomial machinery is unavailable (never on a conforming build); callers fall back to drawing
  bool make(uint64_t steps) { return mtjump::power(steps, g_); }
  // w: a generator state in canonical form (624 words, the next output comes from updating w[0]); advanced in place
  void apply(uint32_t w[mtjump::N]) const {
    using namespace mtjump;
    WState s;
    memcpy(s.w, w, sizeof s.w);
    s.p = 0;
    int i = MEXP - 1;
    while (i >= 0 && !g_.bit(i)) --i;
    WState t;
    if (i < 0) { memset(t.w, 0, sizeof t.w); t.p = 0; }
    else {
      t = s;                                  // leading coefficient
      for (--i; i >= 0; --i) {
        step(t);
        if (g_.bit(i)) add(t, s);
      }
    }
    for (int k = 0; k < N; ++k) w[k] = t.w[(t.p + k) % N];
  }

 private:
  mtjump::Poly g_;
};

}  // namespace svinet
