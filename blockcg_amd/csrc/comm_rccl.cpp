// libblockcg_rccl.so: bcg_comm on RCCL (include/blockcg_rccl.h).  Host code only; the transfers are RCCL's kernels.
#include <hip/hip_runtime.h>
#ifndef BCG_RCCL_MOCK  // the test-only twin (make mock) force-includes a stand-in for these declarations instead
#include <rccl/rccl.h>
#endif
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "../../include/blockcg_rccl.h"

static_assert(BCG_RCCL_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "unique id size");

struct bcg_rccl_comm {
  bcg_context* ctx = nullptr;
  ncclComm_t comm = nullptr;       // all-reduce, barrier / max and the blocking exchange: only ever used on ctx_stream
  ncclComm_t halo_comm = nullptr;  // the split exchange: only ever used on xfer_stream (a communicator of its own, see create)
  bool halo_comm_own = false;
  int rank = 0, world = 1, device = 0;
  hipStream_t ctx_stream = nullptr;   // the context's stream (not owned)
  hipStream_t xfer_stream = nullptr;  // split exchange: higher priority than the compute stream
  hipEvent_t packed = nullptr;
  hipEvent_t arrived[2] = {nullptr, nullptr};  // up to two exchanges outstanding, ended in the order they began
  unsigned begun = 0, ended = 0;
  double* scratch = nullptr;          // one device double for barrier / max
  bcg_comm table{};
  std::string err;
};

namespace {

std::string g_err;

bool fail(bcg_rccl_comm* c, const std::string& msg) {
  if (c) c->err = msg;
  else g_err = msg;
  return false;
}
#define RCCL_OK(c, call)                                                                     \
  do {                                                                                       \
    ncclResult_t r_ = (call);                                                                \
    if (r_ != ncclSuccess) { fail(c, std::string(#call) + ": " + ncclGetErrorString(r_)); return 1; } \
  } while (0)
#define HIP_OK(c, call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) { fail(c, std::string(#call) + ": " + hipGetErrorString(e_)); return 1; } \
  } while (0)

// All messages of one exchange as ONE RCCL group on `stream`: every rank posts its sends and receives together, so
// the point-to-point kernels pair up whatever order the peers appear in.  Two messages to the same peer (a direction
// split over two ranks: the plus and the minus neighbour coincide) are matched by posting order, which is the same
// ascending-direction, low-face-first order on both sides (bcg_halo_plan).
int post_group(bcg_rccl_comm* c, ncclComm_t comm, hipStream_t stream, int n, const int* peer_send, const int* peer_recv,
               const size_t* off_s, const size_t* off_r, const size_t* nbytes) {
  void *send = nullptr, *recv = nullptr;
  size_t each = 0;
  if (bcg_halo_buffers(c->ctx, &send, &recv, &each) != BCG_OK) { fail(c, "bcg_halo_buffers failed"); return 1; }
  for (int k = 0; k < n; ++k) {  // validate before the group opens: an error inside it would leave it open
    if (off_s[k] + nbytes[k] > each || off_r[k] + nbytes[k] > each) { fail(c, "halo message outside the halo buffers"); return 1; }
    if (peer_send[k] < 0 || peer_send[k] >= c->world || peer_recv[k] < 0 || peer_recv[k] >= c->world) {
      fail(c, "halo peer rank outside the communicator (process grid and world size disagree)");
      return 1;
    }
  }
  RCCL_OK(c, ncclGroupStart());
  ncclResult_t bad = ncclSuccess;
  const char* what = "";
  for (int k = 0; k < n && bad == ncclSuccess; ++k) {
    bad = ncclSend(static_cast<const char*>(send) + off_s[k], nbytes[k], ncclChar, peer_send[k], comm, stream);
    what = "ncclSend";
    if (bad != ncclSuccess) break;
    bad = ncclRecv(static_cast<char*>(recv) + off_r[k], nbytes[k], ncclChar, peer_recv[k], comm, stream);
    what = "ncclRecv";
  }
  // the group is closed on EVERY path: a group left open on this thread would swallow each later RCCL call (the
  // all-reduce, the next exchange, the barrier) without launching it, and the peers would hang instead of seeing an error
  const ncclResult_t end = ncclGroupEnd();
  if (bad != ncclSuccess) { fail(c, std::string(what) + ": " + ncclGetErrorString(bad)); return 1; }
  if (end != ncclSuccess) { fail(c, std::string("ncclGroupEnd: ") + ncclGetErrorString(end)); return 1; }
  return 0;
}

int cb_halo(void* user, int n, const int* ps, const int* pr, const size_t* os, const size_t* orr, const size_t* nb) {
  bcg_rccl_comm* c = static_cast<bcg_rccl_comm*>(user);
  return post_group(c, c->comm, c->ctx_stream, n, ps, pr, os, orr, nb);
}
int cb_halo_begin(void* user, int n, const int* ps, const int* pr, const size_t* os, const size_t* orr, const size_t* nb) {
  bcg_rccl_comm* c = static_cast<bcg_rccl_comm*>(user);
  HIP_OK(c, hipEventRecord(c->packed, c->ctx_stream));            // the faces are packed once this fires
  HIP_OK(c, hipStreamWaitEvent(c->xfer_stream, c->packed, 0));
  if (c->begun - c->ended >= 2) { fail(c, "halo_exchange_begin: two exchanges are already outstanding"); return 1; }
  if (post_group(c, c->halo_comm, c->xfer_stream, n, ps, pr, os, orr, nb) != 0) return 1;
  HIP_OK(c, hipEventRecord(c->arrived[c->begun & 1], c->xfer_stream));
  c->begun += 1;
  return 0;
}
int cb_halo_end(void* user) {
  bcg_rccl_comm* c = static_cast<bcg_rccl_comm*>(user);
  if (c->ended == c->begun) { fail(c, "halo_exchange_end without an exchange outstanding"); return 1; }
  HIP_OK(c, hipStreamWaitEvent(c->ctx_stream, c->arrived[c->ended & 1], 0));  // the oldest outstanding exchange: its ghosts may be read after this
  c->ended += 1;
  return 0;
}
int cb_allreduce(void* user, void* buf, size_t count) {
  bcg_rccl_comm* c = static_cast<bcg_rccl_comm*>(user);
  RCCL_OK(c, ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, c->ctx_stream));
  return 0;
}

// The rendezvous file of THIS launch: `path`, or `path.<BCG_RUN_TOKEN>` when the launcher exports a per-launch token
// (tools/launch_ranks.sh does), so that ranks of a new launch can never pick up the id a previous launch left behind.
std::string id_file_of(const char* path) {
  const char* tok = std::getenv("BCG_RUN_TOKEN");
  return tok && *tok ? std::string(path) + "." + tok : std::string(path);
}

}  // namespace

extern "C" {

const char* bcg_rccl_last_error(const bcg_rccl_comm* c) { return c ? c->err.c_str() : g_err.c_str(); }

int bcg_rccl_get_unique_id(void* out) {
  if (!out) return BCG_ERR_INVALID;
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) { g_err = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return BCG_ERR_COMM; }
  std::memcpy(out, &id, sizeof id);
  return BCG_OK;
}

int bcg_rccl_unique_id_via_file(const char* path_arg, int rank, double timeout_s, void* out) {
  if (!path_arg || !out) return BCG_ERR_INVALID;
  const std::string path_s = id_file_of(path_arg);
  const char* path = path_s.c_str();
  if (rank == 0) {
    if (bcg_rccl_get_unique_id(out) != BCG_OK) return BCG_ERR_COMM;
    const std::string tmp = std::string(path) + ".tmp." + std::to_string(static_cast<long>(getpid()));
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(out, 1, BCG_RCCL_UNIQUE_ID_BYTES, f) != BCG_RCCL_UNIQUE_ID_BYTES) {
      if (f) std::fclose(f);
      g_err = "cannot write the unique id file " + tmp;
      return BCG_ERR_COMM;
    }
    std::fclose(f);
    if (std::rename(tmp.c_str(), path) != 0) { g_err = "cannot rename the unique id file"; return BCG_ERR_COMM; }
    return BCG_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (FILE* f = std::fopen(path, "rb")) {
      const size_t n = std::fread(out, 1, BCG_RCCL_UNIQUE_ID_BYTES, f);
      std::fclose(f);
      if (n == BCG_RCCL_UNIQUE_ID_BYTES) return BCG_OK;  // rename() made it appear complete
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
      g_err = std::string("timed out waiting for the unique id file ") + path;
      return BCG_ERR_COMM;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

int bcg_rccl_unique_id_file_done(const char* path, int rank, bcg_rccl_comm* comm) {
  if (!path || !comm) return BCG_ERR_INVALID;
  const int rc = bcg_rccl_barrier(comm);  // every rank holds the communicator, so every rank has read the file
  if (rc != BCG_OK) return rc;
  if (rank == 0) (void)std::remove(id_file_of(path).c_str());
  return BCG_OK;
}

int bcg_comm_rccl_create(bcg_context* ctx, const void* id_bytes, int rank, int world, bcg_rccl_comm** out) {
  if (!ctx || !id_bytes || !out || world < 1 || rank < 0 || rank >= world) {
    g_err = "bcg_comm_rccl_create: bad arguments";
    return BCG_ERR_INVALID;
  }
  struct RestoreDevice {  // the caller's current device is its own business
    int prev = -1;
    RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~RestoreDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
  } restore_device;
  bcg_rccl_comm* c = new bcg_rccl_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  void* s = nullptr;
  auto bail = [&](const std::string& m, int code) {
    g_err = m + (c->err.empty() ? "" : ": " + c->err);
    bcg_comm_rccl_destroy(c);
    return code;
  };
  if (bcg_context_stream(ctx, &s, &c->device) != BCG_OK) return bail("bcg_context_stream failed", BCG_ERR_INVALID);
  c->ctx_stream = static_cast<hipStream_t>(s);
  if (hipSetDevice(c->device) != hipSuccess) return bail("hipSetDevice failed", BCG_ERR_HIP);
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof id);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) return bail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r), BCG_ERR_COMM);
  int lo = 0, hi = 0;  // numerically lower = higher priority
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
  if (hipStreamCreateWithPriority(&c->xfer_stream, hipStreamNonBlocking, hi) != hipSuccess ||
      hipEventCreateWithFlags(&c->packed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->arrived[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->arrived[1], hipEventDisableTiming) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->scratch), sizeof(double)) != hipSuccess)
    return bail("stream/event/scratch creation failed", BCG_ERR_HIP);
  // One communicator per stream.  The split exchange lives on xfer_stream while the Gram all-reduce is enqueued on the
  // context's stream; with ONE communicator RCCL would order the two launches against each other with an event of its
  // own -- the next chunk's exchange silently queued behind a pending all-reduce, or the reverse.  ncclCommSplit with one
  // colour duplicates the communicator (collective over all ranks, no second unique id to distribute).
  // BCG_RCCL_SINGLE_COMM=1 keeps the round-3 behaviour (A/B on hardware).
  const char* single = std::getenv("BCG_RCCL_SINGLE_COMM");
  if (single && std::atoi(single) != 0) {
    c->halo_comm = c->comm;
  } else {
    r = ncclCommSplit(c->comm, 0, rank, &c->halo_comm, nullptr);
    if (r == ncclSuccess) c->halo_comm_own = true;
    else std::fprintf(stderr, "blockcg_rccl: rank %d: ncclCommSplit failed (%s)\n", rank, ncclGetErrorString(r));
    // The ranks AGREE on the outcome over the first communicator: a split that failed on one rank only (a local
    // allocation, say) must not leave that rank posting its split exchanges on `comm` while its peers post theirs on
    // their `halo_comm` -- the sends and receives would never meet and the job would hang at the first overlapped
    // exchange.  If any rank failed, every rank drops its second communicator: the run goes on with ONE communicator for
    // both streams (correct; RCCL then orders the two streams' launches itself), says so on stderr, and
    // bcg_comm_rccl_communicators reports 1 -- bench.py prints it in its line.
    double failed = c->halo_comm_own ? 0.0 : 1.0;
    if (bcg_rccl_sum_double(c, &failed) != BCG_OK) return bail("agreeing on the second communicator failed", BCG_ERR_COMM);
    if (failed > 0.0) {
      if (c->halo_comm_own && c->halo_comm) (void)ncclCommDestroy(c->halo_comm);
      c->halo_comm_own = false;
      c->halo_comm = c->comm;
      if (rank == 0)
        std::fprintf(stderr, "blockcg_rccl: ncclCommSplit failed on %d of %d ranks: one communicator serves both streams\n",
                     static_cast<int>(failed), world);
    }
  }
  c->table.user = c;
  c->table.halo_exchange = cb_halo;
  c->table.allreduce_sum = cb_allreduce;
  c->table.halo_exchange_begin = cb_halo_begin;
  c->table.halo_exchange_end = cb_halo_end;
  if (bcg_context_set_comm(ctx, &c->table) != BCG_OK) return bail("bcg_context_set_comm failed", BCG_ERR_INVALID);
  *out = c;
  return BCG_OK;
}

const bcg_comm* bcg_comm_rccl_callbacks(const bcg_rccl_comm* c) { return c ? &c->table : nullptr; }

int bcg_comm_rccl_communicators(const bcg_rccl_comm* c) { return !c ? 0 : (c->halo_comm_own ? 2 : 1); }

namespace {
int reduce_double(bcg_rccl_comm* c, double* v, ncclRedOp_t op) {
  if (!c || !v) return BCG_ERR_INVALID;
  if (hipMemcpyAsync(c->scratch, v, sizeof(double), hipMemcpyHostToDevice, c->ctx_stream) != hipSuccess) return BCG_ERR_HIP;
  ncclResult_t r = ncclAllReduce(c->scratch, c->scratch, 1, ncclDouble, op, c->comm, c->ctx_stream);
  if (r != ncclSuccess) { c->err = std::string("ncclAllReduce: ") + ncclGetErrorString(r); return BCG_ERR_COMM; }
  if (hipMemcpyAsync(v, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->ctx_stream) != hipSuccess ||
      hipStreamSynchronize(c->ctx_stream) != hipSuccess)
    return BCG_ERR_HIP;
  return BCG_OK;
}
}  // namespace

int bcg_rccl_max_double(bcg_rccl_comm* c, double* v) { return reduce_double(c, v, ncclMax); }
int bcg_rccl_sum_double(bcg_rccl_comm* c, double* v) { return reduce_double(c, v, ncclSum); }

// Every peer of the face exchange is contacted once on each communicator, with the Gram all-reduce in between: RCCL sets
// up a peer's channel (and its device buffers) at the first transfer to it, not at ncclCommInitRank, so only after this
// does the free device memory a launcher reads include what the transport takes.
int bcg_rccl_warm_up(bcg_rccl_comm* c, int n_msgs, const int* peer_send, const int* peer_recv) {
  if (!c || n_msgs < 0 || (n_msgs > 0 && (!peer_send || !peer_recv))) return BCG_ERR_INVALID;
  struct RestoreDevice {
    int prev = -1;
    RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~RestoreDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
  } restore_device;
  if (hipSetDevice(c->device) != hipSuccess) return BCG_ERR_HIP;
  for (int k = 0; k < n_msgs; ++k)
    if (peer_send[k] < 0 || peer_send[k] >= c->world || peer_recv[k] < 0 || peer_recv[k] >= c->world) {
      c->err = "bcg_rccl_warm_up: peer outside the communicator";
      return BCG_ERR_INVALID;
    }
  char* buf = nullptr;  // [n_msgs] words out, [n_msgs] words in
  const size_t word = 256;
  if (n_msgs > 0 && hipMalloc(reinterpret_cast<void**>(&buf), 2 * word * n_msgs) != hipSuccess) return BCG_ERR_HIP;
  int rc = BCG_OK;
  ncclComm_t comms[2] = {c->comm, c->halo_comm};
  hipStream_t streams[2] = {c->ctx_stream, c->xfer_stream};
  for (int which = 0; which < (c->halo_comm_own ? 2 : 1) && rc == BCG_OK && n_msgs > 0; ++which) {
    ncclResult_t bad = ncclGroupStart();
    for (int k = 0; k < n_msgs && bad == ncclSuccess; ++k) {  // one word per message of the plan, in the plan's posting order
      bad = ncclSend(buf + word * k, word, ncclChar, peer_send[k], comms[which], streams[which]);
      if (bad == ncclSuccess) bad = ncclRecv(buf + word * (n_msgs + k), word, ncclChar, peer_recv[k], comms[which], streams[which]);
    }
    const ncclResult_t end = ncclGroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) {
      c->err = std::string("bcg_rccl_warm_up: ") + ncclGetErrorString(bad != ncclSuccess ? bad : end);
      rc = BCG_ERR_COMM;
    }
    if (rc == BCG_OK && hipStreamSynchronize(streams[which]) != hipSuccess) rc = BCG_ERR_HIP;
  }
  double one = 1.0;
  if (rc == BCG_OK) rc = reduce_double(c, &one, ncclSum);
  if (buf) (void)hipFree(buf);
  return rc;
}

int bcg_rccl_barrier(bcg_rccl_comm* c) {
  double one = 1.0;
  return bcg_rccl_max_double(c, &one);
}

int bcg_comm_rccl_destroy(bcg_rccl_comm* c) {
  if (!c) return BCG_OK;
  if (c->ctx) (void)bcg_context_set_comm(c->ctx, nullptr);
  if (c->ctx_stream) (void)hipStreamSynchronize(c->ctx_stream);
  if (c->xfer_stream) {
    (void)hipStreamSynchronize(c->xfer_stream);
    (void)hipStreamDestroy(c->xfer_stream);
  }
  if (c->packed) (void)hipEventDestroy(c->packed);
  for (hipEvent_t e : c->arrived)
    if (e) (void)hipEventDestroy(e);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->halo_comm_own && c->halo_comm) (void)ncclCommDestroy(c->halo_comm);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return BCG_OK;
}

}  // extern "C"
