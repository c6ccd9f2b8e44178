#include "nmi.hh"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <unordered_map>

namespace svinet {

namespace {

bool read_lines(const std::string &path, std::vector<std::vector<long> > *rows) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return false;
  char *line = nullptr;
  size_t cap = 0;
  while (getline(&line, &cap, f) > 0) {
    std::vector<long> v;
    char *p = line, *e;
    for (;;) {
      const long u = strtol(p, &e, 10);
      if (p == e) break;
      v.push_back(u);
      p = e;
    }
    if (!v.empty()) rows->push_back(v);
  }
  free(line);
  fclose(f);
  return true;
}

inline double h(double w, double n) { return w > 0 ? -(w / n) * std::log2(w / n) : 0.0; }

// H(X|Y) normalised; `yidx` maps a node to the communities of Y that contain it
double h_cond(const Cover &x, const Cover &y, const std::unordered_map<uint32_t, std::vector<uint32_t> > &yidx, double n) {
  if (x.empty()) return 0.0;
  double tot = 0.0;
  std::vector<uint32_t> cnt(y.size(), 0), touched;
  for (const auto &xk : x) {
    const double sx = (double)xk.size();
    const double hx = h(sx, n) + h(n - sx, n);
    touched.clear();
    for (uint32_t node : xk) {
      auto it = yidx.find(node);
      if (it == yidx.end()) continue;
      for (uint32_t l : it->second) { if (cnt[l]++ == 0) touched.push_back(l); }
    }
    // communities of Y that share no node with X_k never beat the ones that do unless all fail the
    // constraint; they are still candidates (a = 0), so every l is visited
    double best = -1.0;
    for (uint32_t l = 0; l < y.size(); ++l) {
      const double a = cnt[l], b = sx - a, c = (double)y[l].size() - a, d = n - a - b - c;
      if (h(a, n) + h(d, n) >= h(b, n) + h(c, n)) {
        const double hy = h((double)y[l].size(), n) + h(n - (double)y[l].size(), n);
        const double v = h(a, n) + h(b, n) + h(c, n) + h(d, n) - hy;
        if (best < 0 || v < best) best = v;
      }
    }
    for (uint32_t l : touched) cnt[l] = 0;
    if (best < 0) best = hx;
    tot += hx > 0 ? best / hx : 0.0;
  }
  return tot / (double)x.size();
}

std::unordered_map<uint32_t, std::vector<uint32_t> > index_of(const Cover &c) {
  std::unordered_map<uint32_t, std::vector<uint32_t> > idx;
  for (uint32_t l = 0; l < c.size(); ++l)
    for (uint32_t node : c[l]) idx[node].push_back(l);
  return idx;
}

Cover dedup(Cover c) {
  for (auto &v : c) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  }
  return c;
}

}  // namespace

bool read_cover_lines(const std::string &path, Cover *out) {
  std::vector<std::vector<long> > rows;
  if (!read_lines(path, &rows)) return false;
  out->clear();
  for (const auto &r : rows) out->emplace_back(r.begin(), r.end());
  return true;
}

bool read_cover_memberships(const std::string &path, Cover *out) {
  std::vector<std::vector<long> > rows;
  if (!read_lines(path, &rows)) return false;
  std::map<long, std::vector<uint32_t> > comm;
  for (const auto &r : rows)
    for (size_t i = 1; i < r.size(); ++i) comm[r[i]].push_back((uint32_t)r[0]);
  out->clear();
  for (auto &kv : comm) out->push_back(kv.second);
  return true;
}

double lfk_nmi(const Cover &x0, const Cover &y0) {
  const Cover x = dedup(x0), y = dedup(y0);
  const auto xi = index_of(x), yi = index_of(y);
  size_t n = xi.size();
  for (const auto &kv : yi) if (!xi.count(kv.first)) ++n;   // nodes of the union of both covers
  if (n == 0 || x.empty() || y.empty()) return 0.0;
  return 1.0 - 0.5 * (h_cond(x, y, yi, (double)n) + h_cond(y, x, xi, (double)n));
}

}  // namespace svinet
