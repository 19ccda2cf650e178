// tests/cpp/mock_rccl.hpp -- TEST INFRASTRUCTURE.  A stand-in for the seven RCCL calls blockcg_amd/csrc/comm_rccl.cpp makes,
// so that the NATIVE transport (its message posting order, offsets, validation, the begin/end stream choreography) can be
// run by several processes that SHARE one GPU -- RCCL itself refuses two ranks on one device, and the test box has one.
// Built only into blockcg_amd/_build/libblockcg_rccl_mock.so (make -C blockcg_amd/csrc mock), never into the product.
//
// Semantics kept from NCCL: point-to-point messages between a pair of ranks match in posting order (no tags); everything
// inside ncclGroupStart/End is posted together (sends cannot block on the peer's receives); the all-reduce gives every rank
// the same bits (contributions are added in rank order).  Transport: files under /dev/shm/<id>/ (host-staged), each call
// synchronises the stream it was given first and completes before it returns, so later work on that stream sees the data.
#pragma once
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
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
};
typedef mockComm* ncclComm_t;

namespace mock_rccl {
struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; };
inline std::vector<Op>& group() { static std::vector<Op> g; return g; }
inline int& depth() { static int d = 0; return d; }
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
inline ncclResult_t run(const std::vector<Op>& ops) {
  std::vector<char> host;
  for (const Op& o : ops)  // everything the streams were given before this call is done before any byte moves
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclSystemError;
  for (const Op& o : ops) {  // all sends first: a group never blocks on the peer's receives
    if (!o.send) continue;
    host.resize(o.bytes);
    if (hipMemcpy(host.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
    const uint64_t seq = o.comm->sent[o.peer]++;
    if (!write_file(o.comm->dir + "/p2p_" + std::to_string(o.comm->rank) + "_" + std::to_string(o.peer) + "_" + std::to_string(seq),
                    host.data(), o.bytes))
      return ncclSystemError;
  }
  for (const Op& o : ops) {
    if (o.send) continue;
    host.resize(o.bytes);
    const uint64_t seq = o.comm->received[o.peer]++;
    const std::string path = o.comm->dir + "/p2p_" + std::to_string(o.peer) + "_" + std::to_string(o.comm->rank) + "_" + std::to_string(seq);
    if (!read_file(path, host.data(), o.bytes)) return ncclSystemError;
    std::remove(path.c_str());
    if (hipMemcpy(o.buf, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
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
inline ncclResult_t ncclCommDestroy(ncclComm_t c) {
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
  if (hipStreamSynchronize(s) != hipSuccess) return ncclSystemError;
  std::vector<double> mine(count), other(count), acc(count);
  if (hipMemcpy(mine.data(), in, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
  const uint64_t seq = c->collectives++;
  const std::string base = c->dir + "/ar_" + std::to_string(seq) + "_";
  if (!mock_rccl::write_file(base + std::to_string(c->rank), mine.data(), count * 8)) return ncclSystemError;
  for (int r = 0; r < c->world; ++r) {  // rank order: identical bits on every rank
    if (!mock_rccl::read_file(base + std::to_string(r), other.data(), count * 8)) return ncclSystemError;
    for (size_t i = 0; i < count; ++i) acc[i] = r == 0 ? other[i] : (op == ncclMax ? (other[i] > acc[i] ? other[i] : acc[i]) : acc[i] + other[i]);
  }
  // the files of collective seq are removed when everyone has surely read them: at collective seq + 2
  if (seq >= 2) std::remove((c->dir + "/ar_" + std::to_string(seq - 2) + "_" + std::to_string(c->rank)).c_str());
  if (hipMemcpy(out, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
  return ncclSuccess;
}
