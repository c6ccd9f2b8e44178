// rng.hh -- the random stream the reference takes from GSL.
//
// svinet calls gsl_rng_alloc(gsl_rng_default) [+ gsl_rng_set(seed) when -seed
// is non-zero] (src/linksampling.cc:70-75), then gsl_rng_uniform_int
// (src/linksampling.hh:336-337,344) and gsl_rng_uniform (src/linksampling.cc:392).
// GSL's default generator is MT19937 with default seed 0, which it maps to
// 4357; std::mt19937 implements the same recurrence, tempering and (2002)
// seeding, so it is used as the engine and only GSL's integer/real mappings
// are written out here.
#pragma once
#include <cstdint>
#include <random>

namespace svinet {

class GslMt19937 {
 public:
  explicit GslMt19937(unsigned long seed = 0) : eng_(seed == 0 ? 4357u : (uint32_t)seed) {}
  uint32_t get() { return (uint32_t)eng_(); }
  // gsl_rng_uniform: [0,1) with 32 random bits
  double uniform() { return get() / 4294967296.0; }
  // gsl_rng_uniform_int: rejection sampling on range = max - min = 0xffffffff
  uint32_t uniform_int(uint32_t n) {
    const uint32_t scale = 0xffffffffu / n;
    uint32_t k;
    do k = get() / scale; while (k >= n);
    return k;
  }

 private:
  std::mt19937 eng_;
};

}  // namespace svinet
