// network.hh -- undirected edge-list graph with the reference's node numbering.
// Mirrors the public surface of the reference's Network (src/network.hh:20-101)
// that the link-sampling path uses: read(), n(), ones(), singles(), get_edges(),
// edges(), y(), deg(), seq2id()/id2seq(), deg_stats().
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "flathash.hh"

namespace svinet {

class Env;
typedef std::pair<uint32_t, uint32_t> Edge;

class Network {
 public:
  explicit Network(Env &env);
  // Parses "id <ws> id" lines.  Sequence ids are handed out in order of first
  // appearance (first column before second); once env.n ids exist, lines that
  // name a new id are dropped; self loops and repeated pairs are dropped
  // (src/network.cc:22-104).  Returns 0, or -1 if the file cannot be opened.
  int read(const std::string &path);
  // -strid: node names are arbitrary tokens; they are numbered in order of first appearance and the
  // table is written to <outdir>/str2id.txt sorted by name (src/network.cc:25-45,131-142)
  const std::map<std::string, uint32_t> &str2id() const { return str2id_; }
  // same semantics from memory (used for synthetic graphs)
  void read_pairs(const int32_t *pairs, uint64_t nlines);

  uint32_t n() const { return declared_n_; }             // env.n at construction
  uint32_t nodes_seen() const { return (uint32_t)seq2id_.size(); }
  uint32_t singles() const { return declared_n_ - nodes_seen(); }
  uint32_t ones() const { return (uint32_t)edges_.size(); }
  uint32_t deg(uint32_t p) const { return (uint32_t)adj_[p].size(); }
  const std::vector<uint32_t> &get_edges(uint32_t p) const { return adj_[p]; }
  const std::vector<Edge> &edges() const { return edges_; }
  const std::vector<uint32_t> &seq2id() const { return seq2id_; }
  bool id2seq(uint32_t id, uint32_t *seq) const;
  bool y(uint32_t a, uint32_t b) const;
  void deg_stats(uint32_t &max, double &avg) const;
  // Network::set_env_variables (src/network.cc:222-251): total_pairs (uint32
  // arithmetic), ones_prob/zeros_prob and eta0/eta1 from -eta-type
  void set_env_variables();

 private:
  bool add_line(uint32_t id1, uint32_t id2);
  bool intern(uint32_t id, uint32_t *seq);

  Env &env_;
  uint32_t declared_n_;
  std::vector<std::vector<uint32_t> > adj_;
  std::vector<Edge> edges_;
  std::vector<uint32_t> seq2id_;
  FlatIdMap id2seq_;          // external id -> sequence id
  FlatPairSet pair_set_;      // the links read so far (both directions of a pair are one key)
  std::map<std::string, uint32_t> str2id_;
};

}  // namespace svinet
