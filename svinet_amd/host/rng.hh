// rng.hh -- the random stream the reference takes from GSL.
//
// svinet calls gsl_rng_alloc(gsl_rng_default) [+ gsl_rng_set(seed) when -seed
// is non-zero] (src/linksampling.cc:70-75), then gsl_rng_uniform_int
// (src/linksampling.hh:336-337,344) and gsl_rng_uniform (src/linksampling.cc:392).
// GSL's default generator is MT19937 (Matsumoto & Nishimura, 2002 seeding) with
// default seed 0, which it maps to 4357.  The generator is written out here
// (rather than taken from <random>) so that a whole block of 624 outputs is
// produced by three dependency-free loops the compiler can vectorise: the
// initialisation of gamma draws E*K uniforms from this one sequential stream.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "mtjump.hh"

namespace svinet {

class GslMt19937 {
 public:
  explicit GslMt19937(unsigned long seed = 0) {
    uint32_t s = seed == 0 ? 4357u : (uint32_t)seed;
    mt_[0] = s;
    for (int i = 1; i < N; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
    memcpy(seed_, mt_, sizeof seed_);
    idx_ = N;
  }
  // a generator whose next output is the first one of `state` (canonical form: the next word comes from updating
  // state[0]) and which says it stands at output number `position` of the stream that started at `seed_state`
  GslMt19937(const uint32_t state[624], const uint32_t seed_state[624], uint64_t position) {
    memcpy(mt_, state, sizeof mt_);
    memcpy(seed_, seed_state, sizeof seed_);
    idx_ = N;
    blocks_ = 0;
    base_ = position;
  }
  // outputs drawn since the seed
  uint64_t position() const { return base_ + (blocks_ ? (blocks_ - 1) * (uint64_t)N + (uint64_t)idx_ : 0); }
  const uint32_t *seed_state() const { return seed_; }
  // the generator `steps` outputs after the SEED (not after the current position), by jump-ahead (mtjump.hh: ~20 ms for
  // the polynomial, ~1 ms for its application); false if the jump machinery is unavailable
  bool at(uint64_t pos, GslMt19937 *out) const {
    MtJump j;
    if (!j.make(pos)) return false;
    uint32_t st[N];
    memcpy(st, seed_, sizeof st);
    j.apply(st);
    *out = GslMt19937(st, seed_, pos);
    return true;
  }
  // this generator's state in canonical form, valid only at a block boundary of its own (freshly built or jumped)
  bool canonical_state(uint32_t st[624]) const {
    if (idx_ != N) return false;
    memcpy(st, mt_, sizeof mt_);
    return true;
  }
  uint32_t get() {
    if (idx_ >= N) refill();
    return out_[idx_++];
  }
  // gsl_rng_uniform: [0,1) with 32 random bits
  double uniform() { return get() / 4294967296.0; }
  // n consecutive gsl_rng_uniform() values
  void fill_uniform(double *dst, size_t n) {
    size_t i = 0;
    while (i < n) {
      if (idx_ >= N) refill();
      const size_t take = (size_t)(N - idx_) < n - i ? (size_t)(N - idx_) : n - i;
      const uint32_t *src = out_ + idx_;
      for (size_t j = 0; j < take; ++j) dst[i + j] = src[j] / 4294967296.0;
      idx_ += (int)take;
      i += take;
    }
  }
  // gsl_rng_uniform_int: rejection sampling on range = max - min = 0xffffffff
  uint32_t uniform_int(uint32_t n) {
    const uint32_t scale = 0xffffffffu / n;
    uint32_t k;
    do k = get() / scale; while (k >= n);
    return k;
  }

 private:
  static constexpr int N = 624, M = 397;
  static uint32_t twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
  }
  // next 624 words of the recurrence x[k+N] = x[k+M] ^ twist(x[k], x[k+1]); inside each of the three
  // loops every iteration reads only words no iteration of that loop writes
  void refill() {
    uint32_t *x = mt_;
    for (int k = 0; k < N - M; ++k) x[k] = x[k + M] ^ twist(x[k], x[k + 1]);
    for (int k = N - M; k < 2 * (N - M); ++k) x[k] = x[k + M - N] ^ twist(x[k], x[k + 1]);
    for (int k = 2 * (N - M); k < N - 1; ++k) x[k] = x[k + M - N] ^ twist(x[k], x[k + 1]);
    x[N - 1] = x[M - 1] ^ twist(x[N - 1], x[0]);
    for (int k = 0; k < N; ++k) {   // tempering
      uint32_t y = x[k];
      y ^= y >> 11;
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= y >> 18;
      out_[k] = y;
    }
    idx_ = 0;
    ++blocks_;
  }
  uint32_t mt_[N];
  uint32_t out_[N];
  uint32_t seed_[N];       // the state the stream started from (jump-ahead works from here)
  int idx_;
  uint64_t blocks_ = 0;    // refills since construction
  uint64_t base_ = 0;      // position of the state this object was built from
};

}  // namespace svinet
