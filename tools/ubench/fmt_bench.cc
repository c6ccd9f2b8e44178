// Where do the writers of gamma.txt spend their time on the GPU box?  The formatting loop of write_rows
// (svinet_amd/host/linksampling.cc) in isolation: T threads, blocks of rows into reused std::string buffers, no file.
//   g++ -O3 -std=c++17 -pthread tools/ubench/fmt_bench.cc -o /tmp/fmt_bench && /tmp/fmt_bench [n] [k]
#include "../../svinet_amd/host/fixedfmt.hh"
#include <chrono>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 200000, k = argc > 2 ? atoi(argv[2]) : 512;
  std::vector<double> g((size_t)n * k);
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < 16; ++t) th.emplace_back([&, t] {
      std::mt19937_64 e(t);
      std::uniform_real_distribution<double> u(0, 1);
      for (size_t i = (size_t)n * k * t / 16; i < (size_t)n * k * (t + 1) / 16; ++i) g[i] = 1.0 / 512 + (u(e) < 0.01 ? u(e) * 50 : u(e) * 0.003);
    });
    for (auto &x : th) x.join();
  }
  for (int mode = 0; mode < 2; ++mode)
  for (unsigned T : {1u, 4u, 8u, 16u, 32u}) {
    const uint32_t B = (uint32_t)(((size_t)8 << 20) / ((size_t)k * 10 + 24));
    std::vector<std::string> buf[2] = {std::vector<std::string>(T), std::vector<std::string>(T)};
    int cur = 0;
    size_t bytes = 0;
    const double t0 = now();
    double first2 = 0;
    int wave = 0;
    for (uint32_t base = 0; base < n; base += T * B, cur ^= 1, ++wave) {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t, base, cur] {
        std::string &o = buf[cur][t];
        o.clear();
        const uint32_t b = (uint32_t)std::min<uint64_t>(n, (uint64_t)base + (uint64_t)t * B), e = (uint32_t)std::min<uint64_t>(n, (uint64_t)b + B);
        if (e > b && o.capacity() == 0) o.reserve((size_t)(e - b) * ((size_t)k * 10 + 24));
        if (mode == 0) {
          for (uint32_t i = b; i < e; ++i) {
            const double *row = &g[(size_t)i * k];
            for (uint32_t c = 0; c < k; ++c) svinet::append_fixed<5>(o, row[c], c == k - 1 ? '\n' : '\t');
          }
        } else {
          svinet::RowOut out(o);
          for (uint32_t i = b; i < e; ++i) {
            const double *row = &g[(size_t)i * k];
            for (uint32_t c = 0; c < k; ++c) out.fixed<5>(row[c], c == k - 1 ? '\n' : '\t');
          }
        }
      });
      for (auto &x : th) x.join();
      for (auto &s : buf[cur]) bytes += s.size();
      if (wave == 1) first2 = now() - t0;
    }
    const double s = now() - t0;
    printf("%s T=%2u: %.2f s for %.0f M numbers = %.1f ns per number per thread, %.2f GB/s of text (first two waves %.2f s: page faults of the buffers)\n",
           mode ? "RowOut (scratch per thread)  " : "append to the block's string", T, s, (double)n * k / 1e6, s * T / ((double)n * k) * 1e9, bytes / s / 1e9, first2);
  }
  return 0;
}
