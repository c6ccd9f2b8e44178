// TEST INFRASTRUCTURE -- never part of the product.
//
// A stand-in for librccl that lets SEVERAL PROCESSES SHARING ONE GPU form a communicator, so that the
// library's native multi-rank drivers (svils_sweep_sharded, svils_step_sharded, svils_sweep_ksharded,
// svils_comm_allgather_host, svils_gather_communities and `svinet -gpus N`) run with rank > 0 on the
// one-GPU test box (real RCCL refuses two ranks on one device).  libsvils binds RCCL through dlopen; the
// environment variable SVILS_RCCL_LIBRARY names the library, and only tests/ ever point it here.
//
// Transport: a POSIX shared-memory segment named after the 128-byte unique id, one staging slot per rank.
// Sums are taken in rank order on every rank, so the result is bit-identical everywhere (as RCCL's all-reduce
// is).  A rank that does not show up within FAKERCCL_TIMEOUT_S (default 120) fails the collective on the others.
//
// Two modes:
//   synchronous (default): every collective is executed at the call (or at ncclGroupEnd for grouped calls, in
//     issue order): stream synchronise, device -> slot, barrier, combine from the slots in rank order, -> device,
//     barrier.  Stricter than RCCL's stream ordering, never weaker: it proves offsets, roots, counts and the
//     protocol, but it HIDES a missing event edge between the streams of the caller.
//   asynchronous (FAKERCCL_ASYNC=1): RCCL's contract and nothing more.  The call only ENQUEUES on the op's stream
//     (copy to pinned staging, a host function that meets the peers and combines, copy back) and returns; the data
//     is read when the stream gets there and the result exists only for work ordered after the op on that stream.
//     A producer that was not ordered before the op (missing hipStreamWaitEvent on the communication stream) is read
//     too early, a consumer on another stream that does not wait for the op reads the old bytes -- wrong answers
//     instead of hidden bugs.  FAKERCCL_DELAY_US stretches every collective (the stream stays blocked that long after
//     the peers met), which turns "usually fast enough" races into certain ones.  The operations of ONE communicator
//     are executed in issue order whatever their streams, as RCCL serialises them.
//   Stream capture (asynchronous mode only): a collective issued on a CAPTURING stream becomes graph nodes (copy to a
//     staging buffer that lives as long as the communicator, host node, copy back) and runs at every replay of the graph,
//     as RCCL's captured collectives do.  Captured operations take their turn in stream order (every communicator of the
//     library is used from one stream); the synchronous mode refuses a capturing stream with ncclInvalidUsage, which is
//     how the tests reach the library's "capture failed, stay eager" path.
//   FAKERCCL_EXECUTED=<file>: rank 0 writes "<collectives executed> <of which from captured graphs>" at exit.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t kHeader = 4096;
constexpr size_t kSlot = 8u << 20;   // bytes staged per rank and round

struct Header {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> failed;
  std::atomic<uint64_t> calls;   // collectives executed (all ranks count the same ones): test evidence
};
std::atomic<uint64_t> g_executed{0}, g_executed_captured{0};   // this process: collectives that ran / ran from a graph replay

struct AsyncOp;
struct Comm {
  int rank = 0, world = 1, device = 0;
  Header *hdr = nullptr;
  unsigned char *slots = nullptr;
  size_t bytes = 0;
  char name[64] = {0};
  unsigned long long calls = 0, moved = 0;   // collectives this rank executed, payload bytes it contributed
  // asynchronous mode
  uint64_t issued = 0;                        // operations enqueued (host order)
  std::atomic<uint64_t> completed{0};         // operations whose host function has run: ops of one communicator go in issue order
  std::vector<AsyncOp *> inflight;            // staging of operations that may still be running, reaped at later calls
  std::vector<AsyncOp *> graph_ops;           // operations captured into graphs: alive until the communicator goes
  std::mutex graph_mu;                        // captured operations of one communicator run one at a time
  std::vector<std::pair<unsigned char *, size_t>> pool;   // pinned buffers free for reuse
};

struct Op {
  int kind;   // 0 all-reduce, 1 all-gather, 2 broadcast
  const void *send;
  void *recv;
  size_t count;
  ncclDataType_t dt;
  ncclRedOp_t red;
  int root;
  Comm *comm;
  hipStream_t stream;
};

std::vector<Comm *> g_live;   // communicators not destroyed yet: their statistics are written at process exit

void write_stats(const Comm *c) {   // test evidence: "<rank> <world> <collectives> <bytes> <communicator>" per communicator
  if (const char *path = getenv("FAKERCCL_STATS")) {
    if (FILE *f = fopen(path, "a")) {
      fprintf(f, "%d %d %llu %llu %s\n", c->rank, c->world, c->calls, c->moved, c->name + 10);
      fclose(f);
    }
  }
}

void at_exit() {
  if (const char *path = getenv("FAKERCCL_EXECUTED")) {
    bool rank0 = false;
    for (const Comm *c : g_live) rank0 = rank0 || c->rank == 0;
    static bool written = false;
    if ((rank0 || g_live.empty()) && !written) {
      if (FILE *f = fopen(path, "w")) {
        fprintf(f, "%llu %llu\n", (unsigned long long)g_executed.load(), (unsigned long long)g_executed_captured.load());
        fclose(f);
        written = true;
      }
    }
  }
  for (const Comm *c : g_live) write_stats(c);
  g_live.clear();
}

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double timeout_s() {
  const char *e = getenv("FAKERCCL_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}

size_t dtype_size(ncclDataType_t dt) {
  switch (dt) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

bool barrier(Comm *c) {
  Header *h = c->hdr;
  if (h->failed.load()) return false;
  const uint32_t gen = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == (uint32_t)c->world) {
    h->arrived.store(0);
    h->generation.fetch_add(1);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const double lim = timeout_s();
  unsigned spins = 0;
  while (h->generation.load() == gen) {
    if (h->failed.load()) return false;
    if ((++spins & 1023u) == 0) {
      // a short spin, then sleep: eight ranks of spinning host threads exhaust a 16-core CPU quota and get throttled
      // for the rest of every scheduler period (a collective then takes ~80 ms)
      if (spins > 8192u) usleep(50); else sched_yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim) {
        h->failed.store(1);
        fprintf(stderr, "fakerccl: rank %d waited %.0f s at a barrier (a peer is missing)\n", c->rank, lim);
        return false;
      }
    }
  }
  return true;
}

template <class T>
void combine(T *acc, const T *x, size_t n, ncclRedOp_t red) {
  if (red == ncclSum) for (size_t i = 0; i < n; ++i) acc[i] += x[i];
  else if (red == ncclMax) for (size_t i = 0; i < n; ++i) acc[i] = x[i] > acc[i] ? x[i] : acc[i];
  else for (size_t i = 0; i < n; ++i) acc[i] = x[i] < acc[i] ? x[i] : acc[i];   // ncclMin
}

#define HIPOK(e) do { if ((e) != hipSuccess) return ncclUnhandledCudaError; } while (0)

ncclResult_t run(const Op &o) {
  Comm *c = o.comm;
  const size_t es = dtype_size(o.dt);
  if (!es) return ncclInvalidArgument;
  if (o.kind == 0 && !(o.red == ncclSum || o.red == ncclMax || o.red == ncclMin)) return ncclInvalidArgument;
  if (o.kind == 0 && !(o.dt == ncclFloat64 || o.dt == ncclUint32 || o.dt == ncclUint64 || o.dt == ncclInt32 || o.dt == ncclInt64))
    return ncclInvalidArgument;
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(o.stream, &cap);
    if (cap == hipStreamCaptureStatusActive) return ncclInvalidUsage;   // the synchronous mode cannot be captured
  }
  HIPOK(hipStreamSynchronize(o.stream));
  const size_t per = (kSlot / es) * es / 8 * 8;   // elements' bytes per round, multiple of 8
  const size_t total = o.count * es;
  std::vector<unsigned char> acc;
  for (size_t off = 0; off < total || (total == 0 && off == 0); off += per) {
    const size_t nb = total - off < per ? total - off : per;
    unsigned char *mine = c->slots + (size_t)c->rank * kSlot;
    if (o.kind != 2 || c->rank == o.root) {
      if (nb) HIPOK(hipMemcpy(mine, (const unsigned char *)o.send + off, nb, hipMemcpyDeviceToHost));
    }
    if (!barrier(c)) return ncclSystemError;
    if (o.kind == 0) {
      acc.assign(c->slots, c->slots + nb);
      for (int r = 1; r < c->world; ++r) {
        const unsigned char *x = c->slots + (size_t)r * kSlot;
        switch (o.dt) {
          case ncclFloat64: combine((double *)acc.data(), (const double *)x, nb / 8, o.red); break;
          case ncclUint64: combine((uint64_t *)acc.data(), (const uint64_t *)x, nb / 8, o.red); break;
          case ncclInt64: combine((int64_t *)acc.data(), (const int64_t *)x, nb / 8, o.red); break;
          case ncclUint32: combine((uint32_t *)acc.data(), (const uint32_t *)x, nb / 4, o.red); break;
          default: combine((int32_t *)acc.data(), (const int32_t *)x, nb / 4, o.red); break;
        }
      }
      if (nb) HIPOK(hipMemcpy((unsigned char *)o.recv + off, acc.data(), nb, hipMemcpyHostToDevice));
    } else if (o.kind == 1) {
      for (int r = 0; r < c->world; ++r)
        if (nb) HIPOK(hipMemcpy((unsigned char *)o.recv + (size_t)r * total + off, c->slots + (size_t)r * kSlot, nb, hipMemcpyHostToDevice));
    } else if (c->rank != o.root || o.recv != o.send) {
      if (nb) HIPOK(hipMemcpy((unsigned char *)o.recv + off, c->slots + (size_t)o.root * kSlot, nb, hipMemcpyHostToDevice));
    }
    if (!barrier(c)) return ncclSystemError;
    if (total == 0) break;
  }
  if (c->rank == 0) c->hdr->calls.fetch_add(1);
  g_executed.fetch_add(1);
  c->calls++;
  c->moved += total;
  return ncclSuccess;
}

// ---------------------------------------------------------------- asynchronous mode
bool async_mode() {
  static const bool on = getenv("FAKERCCL_ASYNC") && atoi(getenv("FAKERCCL_ASYNC")) != 0;
  return on;
}
long delay_us() {
  static const long us = getenv("FAKERCCL_DELAY_US") ? atol(getenv("FAKERCCL_DELAY_US")) : 0;
  return us;
}

struct AsyncOp {
  Op o;
  bool captured = false;                         // a graph node: runs at every replay, takes its turn in stream order
  uint64_t seq = 0;
  unsigned char *in = nullptr, *out = nullptr;   // pinned staging
  size_t in_cap = 0, out_cap = 0;
  hipEvent_t done = nullptr;                     // recorded behind the copy back
};

unsigned char *pinned(Comm *c, size_t need, size_t *cap) {
  need = need ? need : 8;
  for (size_t i = 0; i < c->pool.size(); ++i)
    if (c->pool[i].second >= need) {
      unsigned char *p = c->pool[i].first;
      *cap = c->pool[i].second;
      c->pool.erase(c->pool.begin() + (long)i);
      return p;
    }
  void *p = nullptr;
  if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess) return nullptr;
  *cap = need;
  return (unsigned char *)p;
}

void reap(Comm *c, bool wait) {
  for (size_t i = 0; i < c->inflight.size();) {
    AsyncOp *a = c->inflight[i];
    if (wait) (void)hipEventSynchronize(a->done);
    if (wait || hipEventQuery(a->done) == hipSuccess) {
      if (a->in) c->pool.emplace_back(a->in, a->in_cap);
      if (a->out) c->pool.emplace_back(a->out, a->out_cap);
      (void)hipEventDestroy(a->done);
      delete a;
      c->inflight.erase(c->inflight.begin() + (long)i);
    } else {
      ++i;
    }
  }
}

// runs on a runtime thread when the op's stream reaches it: no HIP calls in here
void host_exchange(void *arg) {
  AsyncOp *a = (AsyncOp *)arg;
  const Op &o = a->o;
  Comm *c = o.comm;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  std::unique_lock<std::mutex> turn(c->graph_mu, std::defer_lock);
  if (a->captured) turn.lock();
  while (!a->captured && c->completed.load(std::memory_order_acquire) != a->seq) {   // RCCL serialises the ops of one communicator
    if ((++spins & 1023u) == 0) {
      if (spins > 8192u) usleep(50); else sched_yield();
      if (c->hdr->failed.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
        c->hdr->failed.store(1);
        fprintf(stderr, "fakerccl: rank %d op %llu never got its turn (the ops of one communicator are serialised)\n", c->rank,
                (unsigned long long)a->seq);
        break;
      }
    }
  }
  const size_t es = dtype_size(o.dt);
  const size_t per = (kSlot / es) * es / 8 * 8;
  const size_t total = o.count * es;
  bool ok = !c->hdr->failed.load();
  for (size_t off = 0; ok && (off < total || (total == 0 && off == 0)); off += per) {
    const size_t nb = total - off < per ? total - off : per;
    unsigned char *mine = c->slots + (size_t)c->rank * kSlot;
    if ((o.kind != 2 || c->rank == o.root) && nb) memcpy(mine, a->in + off, nb);
    if (!barrier(c)) { ok = false; break; }
    if (o.kind == 0) {
      unsigned char *acc = a->out + off;
      if (nb) memcpy(acc, c->slots, nb);
      for (int r = 1; r < c->world; ++r) {
        const unsigned char *x = c->slots + (size_t)r * kSlot;
        switch (o.dt) {
          case ncclFloat64: combine((double *)acc, (const double *)x, nb / 8, o.red); break;
          case ncclUint64: combine((uint64_t *)acc, (const uint64_t *)x, nb / 8, o.red); break;
          case ncclInt64: combine((int64_t *)acc, (const int64_t *)x, nb / 8, o.red); break;
          case ncclUint32: combine((uint32_t *)acc, (const uint32_t *)x, nb / 4, o.red); break;
          default: combine((int32_t *)acc, (const int32_t *)x, nb / 4, o.red); break;
        }
      }
    } else if (o.kind == 1) {
      for (int r = 0; r < c->world; ++r)
        if (nb) memcpy(a->out + (size_t)r * total + off, c->slots + (size_t)r * kSlot, nb);
    } else if (nb) {
      memcpy(a->out + off, c->slots + (size_t)o.root * kSlot, nb);
    }
    if (!barrier(c)) { ok = false; break; }
    if (total == 0) break;
  }
  if (!ok) fprintf(stderr, "fakerccl: rank %d: asynchronous op %llu failed (a peer is missing)\n", c->rank, (unsigned long long)a->seq);
  if (const long us = delay_us()) usleep((useconds_t)us);   // the collective "takes" this long: the stream stays blocked
  if (c->rank == 0) c->hdr->calls.fetch_add(1);
  g_executed.fetch_add(1);
  if (a->captured) g_executed_captured.fetch_add(1);
  else c->completed.store(a->seq + 1, std::memory_order_release);
}

ncclResult_t enqueue(const Op &o) {
  Comm *c = o.comm;
  const size_t es = dtype_size(o.dt);
  if (!es) return ncclInvalidArgument;
  if (o.kind == 0 && !(o.red == ncclSum || o.red == ncclMax || o.red == ncclMin)) return ncclInvalidArgument;
  if (o.kind == 0 && !(o.dt == ncclFloat64 || o.dt == ncclUint32 || o.dt == ncclUint64 || o.dt == ncclInt32 || o.dt == ncclInt64))
    return ncclInvalidArgument;
  if (c->hdr->failed.load()) return ncclSystemError;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(o.stream, &cap);
  const bool capturing = cap == hipStreamCaptureStatusActive;
  if (!capturing) reap(c, false);
  const size_t total = o.count * es;
  AsyncOp *a = new AsyncOp;
  a->o = o;
  a->captured = capturing;
  if (!capturing) a->seq = c->issued++;
  const bool contributes = o.kind != 2 || c->rank == o.root;
  const size_t out_bytes = o.kind == 1 ? total * (size_t)c->world : total;
  if (contributes) a->in = pinned(c, total, &a->in_cap);
  a->out = pinned(c, out_bytes, &a->out_cap);
  if ((contributes && !a->in) || !a->out) return ncclUnhandledCudaError;
  if (!capturing) HIPOK(hipEventCreateWithFlags(&a->done, hipEventDisableTiming));
  if (contributes && total) HIPOK(hipMemcpyAsync(a->in, o.send, total, hipMemcpyDeviceToHost, o.stream));
  HIPOK(hipLaunchHostFunc(o.stream, host_exchange, a));
  if (out_bytes && !(o.kind == 2 && c->rank == o.root && o.recv == o.send))
    HIPOK(hipMemcpyAsync(o.recv, a->out, out_bytes, hipMemcpyHostToDevice, o.stream));
  if (capturing) {
    c->graph_ops.push_back(a);     // the graph may be replayed until the communicator is destroyed
  } else {
    HIPOK(hipEventRecord(a->done, o.stream));
    c->inflight.push_back(a);
  }
  c->calls++;
  c->moved += total;
  return ncclSuccess;
}

ncclResult_t submit(const Op &o) {
  if (!o.comm) return ncclInvalidArgument;
  if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
  return async_mode() ? enqueue(o) : run(o);
}

uint64_t fnv(const void *p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char *)p)[i]) * 1099511628211ull;
  return h;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  uint64_t v[4] = {(uint64_t)getpid(), (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(), 0, 0};
  FILE *f = fopen("/dev/urandom", "rb");
  if (f) { if (fread(&v[2], 8, 2, f) != 2) v[2] = v[0] * 0x9e3779b97f4a7c15ull; fclose(f); }
  snprintf(id->internal, sizeof id->internal, "fakerccl:%016llx%016llx%016llx", (unsigned long long)(v[0] ^ v[1]),
           (unsigned long long)v[2], (unsigned long long)v[3]);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  Comm *c = new Comm;
  c->rank = rank;
  c->world = world;
  (void)hipGetDevice(&c->device);
  snprintf(c->name, sizeof c->name, "/fakerccl-%016llx", (unsigned long long)fnv(&id, sizeof id));
  c->bytes = kHeader + (size_t)world * kSlot;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { perror("fakerccl: shm_open"); delete c; return ncclSystemError; }
  if (ftruncate(fd, (off_t)c->bytes) != 0) { perror("fakerccl: ftruncate"); close(fd); delete c; return ncclSystemError; }
  void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { perror("fakerccl: mmap"); delete c; return ncclSystemError; }
  c->hdr = (Header *)p;           // a new segment is zero-filled: counters start at 0 on whoever comes first
  c->slots = (unsigned char *)p + kHeader;
  const bool ok = barrier(c);     // everybody has mapped it
  if (rank == 0) shm_unlink(c->name);
  if (!ok) { munmap(p, c->bytes); delete c; return ncclSystemError; }
  static bool registered = false;
  if (!registered) { atexit(at_exit); registered = true; }
  g_live.push_back(c);
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = (Comm *)comm;
  if (!c) return ncclSuccess;
  reap(c, true);                                     // asynchronous mode: whatever is still enqueued completes first
  (void)hipDeviceSynchronize();                      // ... and no graph holding a captured operation is still running
  for (AsyncOp *a : c->graph_ops) {
    if (a->in) (void)hipHostFree(a->in);
    if (a->out) (void)hipHostFree(a->out);
    delete a;
  }
  c->graph_ops.clear();
  for (auto &b : c->pool) (void)hipHostFree(b.first);
  c->pool.clear();
  write_stats(c);
  for (size_t i = 0; i < g_live.size(); ++i)
    if (g_live[i] == c) { g_live.erase(g_live.begin() + (long)i); break; }
  munmap((void *)c->hdr, c->bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  return submit(Op{0, send, recv, count, dt, op, 0, (Comm *)comm, stream});
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
  return submit(Op{1, send, recv, count, dt, ncclSum, 0, (Comm *)comm, stream});
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t stream) {
  Comm *c = (Comm *)comm;
  if (c && (root < 0 || root >= c->world)) return ncclInvalidArgument;
  return submit(Op{2, send, recv, count, dt, ncclSum, root, c, stream});
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  for (const Op &o : ops) {
    const ncclResult_t r = async_mode() ? enqueue(o) : run(o);
    if (r != ncclSuccess) return r;
  }
  return ncclSuccess;
}

// what svils_comm_query asks (evidence for launchers): answered from the communicator itself
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = ((const Comm *)comm)->world;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
  if (!comm || !rank) return ncclInvalidArgument;
  *rank = ((const Comm *)comm)->rank;
  return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int *device) {
  if (!comm || !device) return ncclInvalidArgument;
  *device = ((const Comm *)comm)->device;
  return ncclSuccess;
}
ncclResult_t ncclGetVersion(int *version) {
  if (!version) return ncclInvalidArgument;
  *version = 0;   // not an RCCL release: the tests' transport
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fakerccl: HIP call failed";
    case ncclSystemError: return "fakerccl: a peer is missing (barrier timeout) or shared memory failed";
    case ncclInvalidArgument: return "fakerccl: invalid argument";
    case ncclInvalidUsage: return "fakerccl: invalid usage";
    default: return "fakerccl: error";
  }
}

// test evidence: how many collectives the communicator has executed
unsigned long long fakercclCalls(ncclComm_t comm) {
  Comm *c = (Comm *)comm;
  return c ? (unsigned long long)c->hdr->calls.load() : 0ull;
}
}
