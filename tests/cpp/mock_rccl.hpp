// tests/cpp/mock_rccl.hpp -- TEST INFRASTRUCTURE.  A stand-in for the seven RCCL calls blockcg_amd/csrc/comm_rccl.cpp makes,
// so that the NATIVE transport (its message posting order, offsets, validation, the begin/end stream choreography) can be
// run by several processes that SHARE one GPU -- RCCL itself refuses two ranks on one device, and the test box has one.
// Built only into blockcg_amd/_build/libblockcg_rccl_mock.so (make -C blockcg_amd/csrc mock), never into the product.
//
// Semantics kept from NCCL: point-to-point messages between a pair of ranks match in posting order (no tags); everything
// inside ncclGroupStart/End is posted together (sends cannot block on the peer's receives); the all-reduce gives every rank
// the same bits (contributions are added in rank order); and every call is ASYNCHRONOUS and stream-ordered: it enqueues
// copies and a host function on the stream it was given and returns at once (see "stream-ordered execution" below), so
// that missing stream / event ordering in the caller shows up as stale data, as it would over RCCL.
// Transport: files under /dev/shm/<id>/ (host-staged through pinned buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclChar = 0, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;

struct mockComm {
  int rank = 0, world = 1;
  std::string dir;
  std::map<int, uint64_t> sent, received;  // per peer: messages posted so far (matching order)
  uint64_t collectives = 0;
  int splits = 0;                          // ncclCommSplit calls so far (collective: the same count on every rank)
};
typedef mockComm* ncclComm_t;
typedef struct { int unused; } ncclConfig_t;

namespace mock_rccl {
struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; };
// group state is per THREAD, as in NCCL (tests/dist_threads_worker.py runs several ranks as threads of one process)
inline std::vector<Op>& group() { static thread_local std::vector<Op> g; return g; }
inline int& depth() { static thread_local int d = 0; return d; }
// BCG_MOCK_SYNC=1: every call waits for its stream, moves the bytes on the calling thread and returns when they have
// arrived -- for ranks that are threads of ONE process, where blocking host functions of several streams could end up
// on one runtime callback thread and wait for each other.  The asynchronous form below is what the multi-process tests run.
inline bool sync_mode() { static const bool s = std::getenv("BCG_MOCK_SYNC") && std::atoi(std::getenv("BCG_MOCK_SYNC")) != 0; return s; }
inline bool write_file(const std::string& path, const void* data, size_t n) {
  const std::string tmp = path + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(data, 1, n, f) == n;
  std::fclose(f);
  return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}
inline bool read_file(const std::string& path, void* data, size_t n, double timeout_s = 120.0) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (FILE* f = std::fopen(path.c_str(), "rb")) {
      const size_t got = std::fread(data, 1, n, f);
      std::fclose(f);
      if (got == n) return true;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}

// ---- stream-ordered execution, like RCCL's ---------------------------------------------------------------------------
// A call only ENQUEUES on the stream it was given and returns: device -> pinned copies of what is sent, a host function
// (hipLaunchHostFunc: runs when the stream gets there, and holds the stream until it returns) that moves the bytes through
// the files, pinned -> device copies of what was received.  So the begin/end choreography of the transport -- events
// between the context's stream and the transfer stream, kernels queued behind an exchange that has not happened yet -- is
// really exercised: work the caller failed to order behind the exchange runs BEFORE the data arrives and reads stale bytes.
struct Piece { bool send; std::string path; void* host; size_t bytes; };
struct Job {
  mockComm* comm;
  std::vector<Piece> pieces;                 // p2p: sends first, then receives
  bool reduce = false;                       // all-reduce job: pieces[0] = own contribution (in/out, doubles)
  std::string ar_base; int world = 1, rank = 0; ncclRedOp_t op = ncclSum; std::string ar_remove;
};
inline std::atomic<bool>& failed() { static std::atomic<bool> f{false}; return f; }
struct Retired { hipEvent_t done; std::vector<void*> hosts; };
inline std::vector<Retired>& retired() { static thread_local std::vector<Retired> r; return r; }
inline hipError_t enqueue_host(hipStream_t s, Job* j);
inline void collect(bool all) {  // free pinned buffers whose stream work has finished
  auto& r = retired();
  for (size_t i = 0; i < r.size();) {
    if (all) (void)hipEventSynchronize(r[i].done);
    if (all || hipEventQuery(r[i].done) == hipSuccess) {
      for (void* h : r[i].hosts) (void)hipHostFree(h);
      (void)hipEventDestroy(r[i].done);
      r.erase(r.begin() + i);
    } else ++i;
  }
}
inline void host_fn(void* arg) {  // no HIP calls in here
  Job* j = static_cast<Job*>(arg);
  if (!j->reduce) {
    for (const Piece& p : j->pieces)
      if (p.send && !write_file(p.path, p.host, p.bytes)) failed() = true;
    for (const Piece& p : j->pieces) {
      if (p.send) continue;
      if (!read_file(p.path, p.host, p.bytes)) { failed() = true; continue; }
      std::remove(p.path.c_str());
    }
  } else {
    const size_t count = j->pieces[0].bytes / 8;
    double* mine = static_cast<double*>(j->pieces[0].host);
    std::vector<double> other(count), acc(count);
    if (!write_file(j->ar_base + std::to_string(j->rank), mine, count * 8)) failed() = true;
    for (int r = 0; r < j->world; ++r) {  // rank order: identical bits on every rank
      if (!read_file(j->ar_base + std::to_string(r), other.data(), count * 8)) { failed() = true; break; }
      for (size_t i = 0; i < count; ++i)
        acc[i] = r == 0 ? other[i] : (j->op == ncclMax ? (other[i] > acc[i] ? other[i] : acc[i]) : acc[i] + other[i]);
    }
    std::memcpy(mine, acc.data(), count * 8);
    if (!j->ar_remove.empty()) std::remove(j->ar_remove.c_str());
  }
  delete j;
}
inline hipError_t enqueue_host(hipStream_t s, Job* j) {
  if (!sync_mode()) return hipLaunchHostFunc(s, host_fn, j);
  const hipError_t e = hipStreamSynchronize(s);  // the device -> pinned copies of what is sent have landed
  if (e != hipSuccess) { delete j; return e; }
  host_fn(j);
  return hipSuccess;
}
inline ncclResult_t finish(hipStream_t stream, std::vector<void*> hosts) {
  Retired r;
  if (hipEventCreateWithFlags(&r.done, hipEventDisableTiming) != hipSuccess || hipEventRecord(r.done, stream) != hipSuccess)
    return ncclSystemError;
  r.hosts = std::move(hosts);
  retired().push_back(std::move(r));
  return ncclSuccess;
}
inline ncclResult_t run(const std::vector<Op>& ops) {
  if (failed()) return ncclSystemError;
  collect(false);
  // one job per stream, in posting order (the transport posts a group on one stream)
  std::vector<hipStream_t> streams;
  for (const Op& o : ops) {
    bool seen = false;
    for (hipStream_t s : streams) seen = seen || s == o.stream;
    if (!seen) streams.push_back(o.stream);
  }
  for (hipStream_t s : streams) {
    Job* j = new Job();
    std::vector<void*> hosts;
    std::vector<std::pair<void*, const Op*>> recvs;
    for (int pass = 0; pass < 2; ++pass)
      for (const Op& o : ops) {
        if (o.stream != s || o.send != (pass == 0)) continue;
        void* h = nullptr;
        if (hipHostMalloc(&h, o.bytes ? o.bytes : 1, hipHostMallocDefault) != hipSuccess) return ncclSystemError;
        hosts.push_back(h);
        j->comm = o.comm;
        if (o.send) {
          if (hipMemcpyAsync(h, o.buf, o.bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclSystemError;
          const uint64_t seq = o.comm->sent[o.peer]++;
          j->pieces.push_back(Piece{true, o.comm->dir + "/p2p_" + std::to_string(o.comm->rank) + "_" + std::to_string(o.peer) + "_" +
                                              std::to_string(seq), h, o.bytes});
        } else {
          const uint64_t seq = o.comm->received[o.peer]++;
          j->pieces.push_back(Piece{false, o.comm->dir + "/p2p_" + std::to_string(o.peer) + "_" + std::to_string(o.comm->rank) + "_" +
                                               std::to_string(seq), h, o.bytes});
          recvs.emplace_back(h, &o);
        }
      }
    if (enqueue_host(s, j) != hipSuccess) return ncclSystemError;
    for (auto& r : recvs)
      if (hipMemcpyAsync(r.second->buf, r.first, r.second->bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclSystemError;
    if (finish(s, std::move(hosts)) != ncclSuccess) return ncclSystemError;
  }
  return ncclSuccess;
}
}  // namespace mock_rccl

inline const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "mock transport error"; }
inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "bcg_mock_%ld_%lld", static_cast<long>(getpid()),
                static_cast<long long>(std::chrono::steady_clock::now().time_since_epoch().count()));
  return ncclSuccess;
}
inline ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
  mockComm* c = new mockComm();
  c->rank = rank;
  c->world = world;
  id.internal[NCCL_UNIQUE_ID_BYTES - 1] = 0;
  c->dir = std::string("/dev/shm/") + id.internal;
  mkdir(c->dir.c_str(), 0700);
  *comm = c;
  return ncclSuccess;
}
// One colour only (what the transport asks for): a duplicate communicator with its own message and collective sequence.
inline ncclResult_t ncclCommSplit(ncclComm_t parent, int color, int key, ncclComm_t* out, ncclConfig_t*) {
  if (!parent || !out || color != 0 || key != parent->rank) return ncclInvalidArgument;
  mockComm* c = new mockComm();
  c->rank = parent->rank;
  c->world = parent->world;
  c->dir = parent->dir + "/split" + std::to_string(parent->splits++);
  mkdir(c->dir.c_str(), 0700);
  *out = c;
  return ncclSuccess;
}
inline ncclResult_t ncclCommDestroy(ncclComm_t c) {
  mock_rccl::collect(true);  // every enqueued transfer of this process has run
  if (c && c->rank == 0) {  // best effort: the directory is left for the other ranks to finish with and removed if empty
    rmdir(c->dir.c_str());
  }
  delete c;
  return ncclSuccess;
}
inline ncclResult_t ncclGroupStart() { ++mock_rccl::depth(); return ncclSuccess; }
inline ncclResult_t ncclGroupEnd() {
  if (--mock_rccl::depth() > 0) return ncclSuccess;
  std::vector<mock_rccl::Op> ops;
  ops.swap(mock_rccl::group());
  return mock_rccl::run(ops);
}
inline ncclResult_t mock_p2p(bool send, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
  const size_t bytes = count * (t == ncclDouble ? 8 : 1);
  mock_rccl::Op o{send, buf, bytes, peer, c, s};
  if (mock_rccl::depth() > 0) { mock_rccl::group().push_back(o); return ncclSuccess; }
  return mock_rccl::run({o});
}
inline ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  return mock_p2p(true, const_cast<void*>(buf), count, t, peer, c, s);
}
inline ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  return mock_p2p(false, buf, count, t, peer, c, s);
}
inline ncclResult_t ncclAllReduce(const void* in, void* out, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                                  hipStream_t s) {
  if (t != ncclDouble) return ncclInvalidArgument;
  if (mock_rccl::failed()) return ncclSystemError;
  mock_rccl::collect(false);
  void* h = nullptr;
  if (hipHostMalloc(&h, count * 8, hipHostMallocDefault) != hipSuccess) return ncclSystemError;
  if (hipMemcpyAsync(h, in, count * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclSystemError;
  const uint64_t seq = c->collectives++;
  mock_rccl::Job* j = new mock_rccl::Job();
  j->comm = c;
  j->reduce = true;
  j->pieces.push_back(mock_rccl::Piece{true, "", h, count * 8});
  j->ar_base = c->dir + "/ar_" + std::to_string(seq) + "_";
  j->world = c->world;
  j->rank = c->rank;
  j->op = op;
  // the files of collective seq are removed when everyone has surely read them: at collective seq + 2
  if (seq >= 2) j->ar_remove = c->dir + "/ar_" + std::to_string(seq - 2) + "_" + std::to_string(c->rank);
  if (mock_rccl::enqueue_host(s, j) != hipSuccess) return ncclSystemError;
  if (hipMemcpyAsync(out, h, count * 8, hipMemcpyHostToDevice, s) != hipSuccess) return ncclSystemError;
  return mock_rccl::finish(s, std::vector<void*>{h});
}
