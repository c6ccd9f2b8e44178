// flathash.hh -- open-addressing tables for the edge-list reader (Network): a set of 64-bit pair keys and a map from
// external ids to sequence ids.  Linear probing in a power-of-two table at load <= 1/2; the node-based std containers
// cost the reader ~0.5 us per line of a 12 M-line file (a cache miss per probe plus an allocation per insert).
#pragma once
#include <cstdint>
#include <vector>

namespace svinet {

inline uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// keys are never 0 (a pair key holds two different node numbers, the smaller one in the high word)
class FlatPairSet {
 public:
  void reserve(size_t n) { if (2 * n + 2 > tab_.size()) rehash(2 * n + 2); }
  size_t size() const { return size_; }
  bool insert(uint64_t key) {            // true: new
    if (2 * (size_ + 1) > tab_.size()) rehash(tab_.empty() ? 1024 : 2 * tab_.size());
    const size_t m = tab_.size() - 1;
    for (size_t i = mix64(key) & m;; i = (i + 1) & m) {
      if (tab_[i] == key) return false;
      if (tab_[i] == 0) { tab_[i] = key; ++size_; return true; }
    }
  }
  void prefetch(uint64_t key) const {    // the slot a later insert / contains of `key` starts at
    if (!tab_.empty()) __builtin_prefetch(&tab_[mix64(key) & (tab_.size() - 1)], 1, 1);
  }
  bool contains(uint64_t key) const {
    if (tab_.empty()) return false;
    const size_t m = tab_.size() - 1;
    for (size_t i = mix64(key) & m;; i = (i + 1) & m) {
      if (tab_[i] == key) return true;
      if (tab_[i] == 0) return false;
    }
  }

 private:
  void rehash(size_t want) {
    size_t cap = 1024;
    while (cap < want) cap <<= 1;
    std::vector<uint64_t> old;
    old.swap(tab_);
    tab_.assign(cap, 0);
    size_ = 0;
    for (uint64_t k : old)
      if (k) insert(k);
  }
  std::vector<uint64_t> tab_;
  size_t size_ = 0;
};

// external id (any 32-bit value) -> sequence id; a slot is free while its value is 0xffffffff
class FlatIdMap {
 public:
  void reserve(size_t n) { if (2 * n + 2 > key_.size()) rehash(2 * n + 2); }
  bool find(uint32_t id, uint32_t *seq) const {
    if (key_.empty()) return false;
    const size_t m = key_.size() - 1;
    for (size_t i = mix64(id) & m;; i = (i + 1) & m) {
      if (val_[i] == kFree) return false;
      if (key_[i] == id) { *seq = val_[i]; return true; }
    }
  }
  void prefetch(uint32_t id) const {
    if (key_.empty()) return;
    const size_t i = mix64(id) & (key_.size() - 1);
    __builtin_prefetch(&val_[i], 0, 1);
    __builtin_prefetch(&key_[i], 0, 1);
  }
  void emplace(uint32_t id, uint32_t seq) {   // id must be absent
    if (2 * (size_ + 1) > key_.size()) rehash(key_.empty() ? 1024 : 2 * key_.size());
    const size_t m = key_.size() - 1;
    size_t i = mix64(id) & m;
    while (val_[i] != kFree) i = (i + 1) & m;
    key_[i] = id; val_[i] = seq; ++size_;
  }

 private:
  static constexpr uint32_t kFree = 0xffffffffu;
  void rehash(size_t want) {
    size_t cap = 1024;
    while (cap < want) cap <<= 1;
    std::vector<uint32_t> ok, ov;
    ok.swap(key_); ov.swap(val_);
    key_.assign(cap, 0); val_.assign(cap, kFree);
    size_ = 0;
    for (size_t i = 0; i < ok.size(); ++i)
      if (ov[i] != kFree) emplace(ok[i], ov[i]);
  }
  std::vector<uint32_t> key_, val_;
  size_t size_ = 0;
};

}  // namespace svinet
