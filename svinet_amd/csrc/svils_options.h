// svils_options.h -- every tunable of the library in ONE table (include/svils.h: svils_set_option / svils_get_option /
// svils_option_table).  A handle's options start from the table's defaults, overridden by the SVILS_* environment
// variables as they stand when svils_create() runs (read there, once per handle -- nothing on a sweep path reads the
// environment); svils_set_option changes one of them for one handle afterwards.  The two hooks that exist only for the
// tests (fault injection, a pretended CU count) are compiled in with -DSVILS_TESTING alone (libsvils_testing.so).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace svils_impl {

struct Options {
  // ---- read by svils_create
  int pack_rows = 1;            // K <= 56: rows packed at ld = round_up(K, 2) instead of a 128-byte stride
  int derive_m = 1;             // whole sweeps derive the mean indicators from gamma instead of storing them (K > 56)
  int64_t epi_max_mb = -1;      // largest n-by-k array for which exp(Elogpi) is kept; -1: automatic (svils_create)
  uint32_t graph_after = 128;   // sweeps a handle runs eagerly before svils_sweep captures hipGraphs (0: at once)
  int skip_elogpi = -1;         // 57 <= K <= 512: Elogpi not stored (DeviceState::skip_elogpi): -1 where the n-by-k state is >= 256 MB, 0 / 1 forced
  int shard_fold = 1;           // node-block sweeps, K <= 32: the kernels leave the K-vectors themselves (no k_colreduce)
  int graph_pow2 = 1;           // replay as few graphs as possible (powers of two up to 64 sweeps); 0: 8-sweep graphs + singles
  // ---- read by svils_set_graph
  uint64_t lpl_max_entries = 1ull << 27;   // CSR entries up to which K <= 56 takes the lane-per-link kernels
  int wt = -1;                  // write-through row stores: -1 by the size of the state (1 - 8 MB), 0 / 1 forced
  int fused3 = -1;              // three-launch sweeps: -1 by the co-residency check, 0 / 1 forced
  // ---- node-block runs (read at every svils_sweep_sharded from the handle, never from the environment)
  int one_comm = 0;             // 1: the chunked row exchange shares the first communicator
  uint32_t xchunks = 0;         // chunks of the pipelined row exchange; 0: by payload (one below 256 MB, then one per 128 MB, <= 8)
  int row_exchange = 0;         // 0 automatic, 1 all-gather of padded slices, 2 broadcasts with exact counts
  int sharded_graphs = 1;       // 0: node-block sweeps are never captured into hipGraphs
  // ---- reports
  int report_staged = 0;        // 1: every report goes through the device staging slot + copy stream
#ifdef SVILS_TESTING
  int fault_inject = 0;         // 1: one classification worker never publishes its tile (tests/test_gpu_parity.py)
  int assume_cus = 0;           // > 0: the co-residency check pretends the device has this many CUs
#endif
};

Options options_from_env();
// -> 0, or -1 unknown key, -2 bad value, -3 too late for this handle (the option was consumed by svils_create / svils_set_graph)
int option_set(Options &o, const char *key, const char *value, bool created, bool have_graph);
int option_get(const Options &o, const char *key, char *buf, size_t cap);
const char *option_table_text();

}  // namespace svils_impl
