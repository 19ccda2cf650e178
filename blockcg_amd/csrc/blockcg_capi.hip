// Implementation of include/blockcg_hip.h: context, fields, operator and the SBCGrQ driver.
//
// Host side of the hot path: this file holds the reference's control flow
// (inc/block_solvers.hpp:91-185) and its m x m coefficient algebra; every loop over lattice sites
// is a HIP kernel (kernels_generic.hip, kernels_mfma.hip, kernels_stencil.hip).  There is no CPU fallback: without a
// gfx950 device bcg_context_create fails with BCG_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <thread>

#include "context.hpp"
#include "kernels_mfma.hpp"

using bcg::CMat;
using bcg::cd;

namespace {

std::string g_create_error;

#define BCG_FAIL(ctx, code, msg) \
  do {                           \
    (ctx)->err = (msg);          \
    return (code);               \
  } while (0)

#define HIP_TRY(ctx, call)                                                                         \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return BCG_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

#define BCG_TRY(call)            \
  do {                           \
    int rc_ = (call);            \
    if (rc_ != BCG_OK) return rc_; \
  } while (0)

// Every entry point that allocates or launches runs with the context's device current and restores the caller's
// afterwards: a host with several contexts (or one that switches devices between calls, as torch does) must not get
// fields on the wrong GPU.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(const bcg_context* c) {
    if (!c) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

// ---- profiling: HIP events on the context's stream around each kernel class ------------------
struct ProfScope {
  bcg_context* c;
  bcg::ProfEntry* e = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  // alg_bytes: the ALGORITHMIC HBM bytes of what is launched inside the scope (per-site figure x the
  // sites this launch processes); summed per kernel class so that bench.py's roofline is right for split launches
  // (phase C in two launches, capacity-mode windows) too.
  // alg_flops: the fp64 flops of the same launches (row kernels: 8 m^2 per row and m x m product on the matrix pipe;
  // stencil: 576 per site and right-hand side on the VALU) -- the second roofline of the grouped phase C
  ProfScope(bcg_context* ctx, const char* name, double alg_bytes = 0.0, double alg_flops = 0.0) : c(ctx) {
    if (!c->profiling) return;
    e = &c->prof[name];
    e->bytes += alg_bytes;
    e->flops += alg_flops;
    a = take();
    b = take();
    (void)hipEventRecord(a, c->stream);
  }
  ~ProfScope() {
    if (!e) return;
    (void)hipEventRecord(b, c->stream);
    e->pending.emplace_back(a, b);
  }
  hipEvent_t take() {
    if (!c->event_pool.empty()) {
      hipEvent_t ev = c->event_pool.back();
      c->event_pool.pop_back();
      return ev;
    }
    hipEvent_t ev;
    (void)hipEventCreate(&ev);
    return ev;
  }
};

void resolve_profile(bcg_context* c) {
  for (auto& kv : c->prof) {
    for (auto& pr : kv.second.pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        kv.second.ms += ms;
        kv.second.count += 1;
      }
      c->event_pool.push_back(pr.first);
      c->event_pool.push_back(pr.second);
    }
    kv.second.pending.clear();
  }
}

int stream_sync(bcg_context* c) {
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->mat_in_flight = 0;
  if (c->profiling) resolve_profile(c);
  return BCG_OK;
}

int check_launch(bcg_context* c, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return BCG_ERR_HIP;
  }
  return BCG_OK;
}

// ---- scratch management -----------------------------------------------------------------------
int ensure_halo(bcg_context* c, size_t bytes) {
  if (bytes <= c->halo_bytes) return BCG_OK;
  BCG_TRY(stream_sync(c));
  if (c->halo_send) (void)hipFree(c->halo_send);
  if (c->halo_recv) (void)hipFree(c->halo_recv);
  c->halo_send = c->halo_recv = nullptr;
  c->halo_bytes = 0;
  HIP_TRY(c, hipMalloc(&c->halo_send, bytes));
  HIP_TRY(c, hipMalloc(&c->halo_recv, bytes));
  c->halo_bytes = bytes;
  return BCG_OK;
}

constexpr int kMaxGramBlocks = 2048;  // also covers interior + boundary stencil launches (2 x 1024)
constexpr size_t kMatSlotBytes = 32 * 32 * sizeof(double2);
constexpr int kMatSlots = 96;

int ensure_scratch(bcg_context* c) {
  if (!c->partials) {
    HIP_TRY(c, hipMalloc(&c->partials, static_cast<size_t>(kMaxGramBlocks) * 32 * 32 * sizeof(double2)));
    c->partials_bytes = static_cast<size_t>(kMaxGramBlocks) * 32 * 32 * sizeof(double2);
  }
  if (!c->dev_mats) {
    c->mat_slot_bytes = kMatSlotBytes;
    c->mat_slots = kMatSlots;
    HIP_TRY(c, hipMalloc(&c->dev_mats, kMatSlotBytes * kMatSlots));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->pin_mats), kMatSlotBytes * kMatSlots, hipHostMallocDefault));
  }
  if (!c->hop_tune.sync.counters) {  // pacing counters of the specialised stencil (HopSync)
    constexpr int kSyncStride = 8192;
    HIP_TRY(c, hipMalloc(&c->hop_tune.sync.counters, sizeof(unsigned) * 8 * kSyncStride));
    c->hop_tune.sync.stride = kSyncStride;
  }
  if (!c->fold_tickets) {
    HIP_TRY(c, hipMalloc(&c->fold_tickets, 16 * sizeof(unsigned)));
    HIP_TRY(c, hipMemsetAsync(c->fold_tickets, 0, 16 * sizeof(unsigned), c->stream));
  }
  if (!c->dev_gram) {
    HIP_TRY(c, hipMalloc(&c->dev_gram, kMatSlotBytes));
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->pin_gram), kMatSlotBytes, hipHostMallocDefault));
  }
  return BCG_OK;
}

// Copy n coefficient matrices (m x m each) to consecutive device slots; returns the device pointer
// of the first.  Slots are recycled only after a stream synchronization.
int upload_mats(bcg_context* c, int m, const CMat* const* mats, int n, const double2** dev_out) {
  BCG_TRY(ensure_scratch(c));
  const size_t each = static_cast<size_t>(m) * m * sizeof(double2);
  const size_t total = each * n;
  const int need = static_cast<int>((total + c->mat_slot_bytes - 1) / c->mat_slot_bytes);
  if (need > c->mat_slots) BCG_FAIL(c, BCG_ERR_INVALID, "too many coefficient matrices in one upload");
  if (c->mat_next + need > c->mat_slots) {
    c->mat_in_flight += c->mat_slots - c->mat_next;
    c->mat_next = 0;
  }
  if (c->mat_in_flight + need > c->mat_slots) BCG_TRY(stream_sync(c));
  char* hp = reinterpret_cast<char*>(c->pin_mats) + c->mat_next * c->mat_slot_bytes;
  char* dp = reinterpret_cast<char*>(c->dev_mats) + c->mat_next * c->mat_slot_bytes;
  for (int k = 0; k < n; ++k) std::memcpy(hp + k * each, mats[k]->data(), each);
  HIP_TRY(c, hipMemcpyAsync(dp, hp, total, hipMemcpyHostToDevice, c->stream));
  c->mat_next += need;
  c->mat_in_flight += need;
  *dev_out = reinterpret_cast<const double2*>(dp);
  return BCG_OK;
}
int upload_mat(bcg_context* c, const CMat& M, const double2** dev_out) {
  const CMat* p = &M;
  return upload_mats(c, M.dim(), &p, 1, dev_out);
}

int ensure_staging(bcg_context* c, size_t bytes) {
  if (bytes <= c->staging_bytes) return BCG_OK;
  BCG_TRY(stream_sync(c));  // a conversion kernel of an earlier call may still read the old buffer
  if (c->staging) (void)hipFree(c->staging);
  c->staging = nullptr;
  c->staging_bytes = 0;
  HIP_TRY(c, hipMalloc(&c->staging, bytes));
  c->staging_bytes = bytes;
  return BCG_OK;
}

// ---- host <-> device transfer pipeline (bcg_field_upload / bcg_field_download) ---------------------------------------------
// The reference keeps its fields in host memory and reads elements there (benchmark.cpp:61-63); the device layout is
// [site][colour][rhs], the host layout [site][rhs][colour], so every transfer passes a conversion kernel.  Chunks of
// kXferChunk bytes alternate between two streams, each with a device staging buffer: the conversion kernel of one chunk
// runs while the other chunk is on the bus.  Host memory the runtime knows as pinned (bcg_host_alloc, hipHostMalloc,
// hipHostRegister) is the DMA's source / target directly; pageable memory goes through two pinned buffers that host threads
// fill or drain while the other chunk is in flight (one memcpy thread cannot keep up with the bus).
constexpr size_t kXferChunk = static_cast<size_t>(64) << 20;

int ensure_xfer(bcg_context* c, bool need_pinned) {
  if (!c->xfer_stream[0]) {
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(c, hipStreamCreateWithFlags(&c->xfer_stream[k], hipStreamNonBlocking));
      HIP_TRY(c, hipEventCreateWithFlags(&c->xfer_done[k], hipEventDisableTiming));
      HIP_TRY(c, hipMalloc(&c->xfer_dev[k], kXferChunk));
    }
    c->xfer_bytes = kXferChunk;
  }
  if (need_pinned && !c->xfer_pin[0])
    for (int k = 0; k < 2; ++k) HIP_TRY(c, hipHostMalloc(&c->xfer_pin[k], kXferChunk, hipHostMallocDefault));
  return BCG_OK;
}

void par_memcpy(void* dst, const void* src, size_t n) {
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = static_cast<int>(std::max(1u, std::min(8u, hw ? hw / 2 : 1u)));
  if (nt == 1 || n < (static_cast<size_t>(4) << 20)) {
    std::memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> th;
  const size_t each = ((n / nt) + 4095) & ~static_cast<size_t>(4095);
  for (int t = 1; t < nt; ++t) {
    const size_t o = each * t;
    if (o >= n) break;
    th.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, std::min(each, n - o)); });
  }
  std::memcpy(dst, src, std::min(each, n));
  for (auto& t : th) t.join();
}

bool host_is_pinned(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary malloc'ed pointer is "invalid value" to the runtime: not an error here
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

int transfer_field(bcg_context* c, bcg_field* f, double* host, bool to_device) {
  const size_t site_bytes = static_cast<size_t>(3) * f->m * sizeof(double2);
  const bool direct = host_is_pinned(host);
  BCG_TRY(ensure_xfer(c, !direct));
  BCG_TRY(stream_sync(c));  // order against everything enqueued on the context's stream
  const int64_t chunk = static_cast<int64_t>(c->xfer_bytes / site_bytes);
  const int64_t V = f->sites;
  const int64_t nchunks = (V + chunk - 1) / chunk;
  char* const hb = reinterpret_cast<char*>(host);
  auto sites_of = [&](int64_t i) { return std::min<int64_t>(chunk, V - i * chunk); };
  for (int64_t i = 0; i < nchunks + 2; ++i) {
    const int k = static_cast<int>(i & 1);
    if (i >= 2) {  // chunk i - 2 used the same stream and buffers
      HIP_TRY(c, hipEventSynchronize(c->xfer_done[k]));
      if (!to_device && !direct) par_memcpy(hb + (i - 2) * chunk * site_bytes, c->xfer_pin[k], sites_of(i - 2) * site_bytes);
    }
    if (i >= nchunks) continue;
    const int64_t n = sites_of(i);
    hipStream_t s = c->xfer_stream[k];
    char* const hchunk = hb + i * chunk * site_bytes;
    double2* const dchunk = f->d + i * chunk * 3 * f->m;
    if (to_device) {
      const void* src = hchunk;
      if (!direct) {
        par_memcpy(c->xfer_pin[k], hchunk, n * site_bytes);
        src = c->xfer_pin[k];
      }
      HIP_TRY(c, hipMemcpyAsync(c->xfer_dev[k], src, n * site_bytes, hipMemcpyHostToDevice, s));
      bcg::launch_host_to_dev(s, f->m, c->xfer_dev[k], dchunk, n);
      BCG_TRY(check_launch(c, "host_to_dev"));
    } else {
      bcg::launch_dev_to_host(s, f->m, dchunk, c->xfer_dev[k], n);
      BCG_TRY(check_launch(c, "dev_to_host"));
      HIP_TRY(c, hipMemcpyAsync(direct ? static_cast<void*>(hchunk) : c->xfer_pin[k], c->xfer_dev[k], n * site_bytes,
                                hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipEventRecord(c->xfer_done[k], s));
  }
  return BCG_OK;
}

inline int64_t rows_of(const bcg_field* f) { return f->sites * 3; }
// algorithmic bytes: `fields` passes over a width-m field (s = 48 m bytes per site) plus `links` passes over the gauge
// links (g = 144 ndim bytes per site), over the fraction num/den of the local volume
inline double alg_bytes(const bcg_context* c, int m, double fields, double links = 0.0, int64_t num = 1, int64_t den = 1) {
  return static_cast<double>(c->lat.V) * (fields * 48.0 * m + links * 144.0 * c->ndim) * static_cast<double>(num) /
         static_cast<double>(den);
}
// the same for `fields` passes over the rows of one field (half-volume fields have half the rows), and the flops of
// `products` right-multiplications by (or Gram products with) m x m complex matrices over those rows
inline double row_bytes(const bcg_field* f, double fields) { return static_cast<double>(f->sites) * 48.0 * f->m * fields; }
inline double product_flops(const bcg_field* f, double products) {
  return static_cast<double>(f->sites) * 3.0 * f->m * f->m * 8.0 * products;
}
inline double hop_flops(const bcg_context* c, int m, bool gram, int64_t num = 1, int64_t den = 1) {
  return static_cast<double>(c->lat.V) * m * (72.0 * 2 * c->ndim + (gram ? 24.0 * m : 0.0)) * static_cast<double>(num) /
         static_cast<double>(den);
}
inline size_t field_bytes(const bcg_context* c, int m) { return static_cast<size_t>(c->lat.V) * 3 * m * sizeof(double2); }
inline size_t field_bytes(const bcg_field* f) { return static_cast<size_t>(f->sites) * 3 * f->m * sizeof(double2); }

// ---- halo exchange -----------------------------------------------------------------------------
// rank = lexicographic index of grid coordinates, direction 0 fastest
int rank_of_grid(const int* grid, const int* xyz) {
  int r = 0, st = 1;
  for (int mu = 0; mu < 4; ++mu) {
    r += xyz[mu] * st;
    st *= grid[mu];
  }
  return r;
}

// The message plan of one halo exchange (pure host arithmetic, shared by bcg_halo_plan and the
// context).  Ghost/send buffers hold, per split direction in ascending mu, [minus face][plus face]
// (send buffer: [low face x_mu = 0][high face x_mu = L-1]); a face has V_local / L_mu sites.
//   message 2k  : low face  -> minus neighbour (becomes its plus ghost); my plus ghost  <- plus neighbour
//   message 2k+1: high face -> plus neighbour  (becomes its minus ghost); my minus ghost <- minus neighbour
int halo_plan(int ndim, const int* gdims, const int* grid, const int* coords, size_t site_bytes, int* peer_s, int* peer_r,
              size_t* off_s, size_t* off_r, size_t* nb, int64_t* ghost_sites) {
  int g4[4] = {1, 1, 1, 1}, c4[4] = {0, 0, 0, 0}, L[4] = {1, 1, 1, 1};
  int64_t V = 1;
  for (int mu = 0; mu < ndim; ++mu) {
    g4[mu] = grid ? grid[mu] : 1;
    c4[mu] = coords ? coords[mu] : 0;
    if (g4[mu] < 1 || gdims[mu] < 1 || gdims[mu] % g4[mu] != 0 || c4[mu] < 0 || c4[mu] >= g4[mu]) return -1;
    L[mu] = gdims[mu] / g4[mu];
    V *= L[mu];
  }
  int n = 0;
  int64_t ghost = 0;
  for (int mu = 0; mu < ndim; ++mu) {
    if (g4[mu] == 1) continue;
    int xm[4], xp[4];
    for (int nu = 0; nu < 4; ++nu) xm[nu] = xp[nu] = c4[nu];
    xm[mu] = (c4[mu] - 1 + g4[mu]) % g4[mu];
    xp[mu] = (c4[mu] + 1) % g4[mu];
    const int rm = rank_of_grid(g4, xm), rp = rank_of_grid(g4, xp);
    const int64_t face_sites = V / L[mu];
    const size_t face = static_cast<size_t>(face_sites) * site_bytes;
    const size_t base = static_cast<size_t>(ghost) * site_bytes;
    peer_s[n] = rm; peer_r[n] = rp; off_s[n] = base; off_r[n] = base + face; nb[n] = face; ++n;
    peer_s[n] = rp; peer_r[n] = rm; off_s[n] = base + face; off_r[n] = base; nb[n] = face; ++n;
    ghost += 2 * face_sites;
  }
  if (ghost_sites) *ghost_sites = ghost;
  return n;
}

// Post the face messages for `site_bytes` bytes per site (fields: 3*m*16; gauge: 9*16).  split = true uses the
// begin half of the optional split form (the caller then issues exchange_end after the interior tiles).
// x3_n > 0 (direction 3 undivided): only the slices [x3_lo, x3_lo + x3_n), a contiguous sub-range of every face.
// x3b_n > 0: a second range of slices in the same exchange (the messages of the first range, then those of the second)
int exchange_faces(bcg_context* c, size_t site_bytes, bool split = false, int x3_lo = 0, int x3_n = 0, int x3b_lo = 0, int x3b_n = 0) {
  if (!c->have_comm || !c->comm.halo_exchange) BCG_FAIL(c, BCG_ERR_COMM, "lattice is split over ranks but no bcg_comm was set");
  int peer_s[16], peer_r[16];
  size_t off_s[16], off_r[16], nb[16];
  int n = halo_plan(c->ndim, c->gdims, c->grid, c->coords, site_bytes, peer_s, peer_r, off_s, off_r, nb, nullptr);
  if (n < 0) BCG_FAIL(c, BCG_ERR_INVALID, "halo plan");
  if (x3_n > 0) {
    for (int k = 0; k < n; ++k) {
      const size_t slice = nb[k] / c->lat.L[3];
      if (x3b_n > 0) {
        peer_s[n + k] = peer_s[k];
        peer_r[n + k] = peer_r[k];
        off_s[n + k] = off_s[k] + slice * x3b_lo;
        off_r[n + k] = off_r[k] + slice * x3b_lo;
        nb[n + k] = slice * x3b_n;
      }
      off_s[k] += slice * x3_lo;
      off_r[k] += slice * x3_lo;
      nb[k] = slice * x3_n;
    }
    if (x3b_n > 0) n *= 2;
  }
  auto fn = split ? c->comm.halo_exchange_begin : c->comm.halo_exchange;
  if (fn(c->comm.user, n, peer_s, peer_r, off_s, off_r, nb) != 0) BCG_FAIL(c, BCG_ERR_COMM, "halo_exchange callback failed");
  return BCG_OK;
}
inline bool can_overlap(const bcg_context* c) {
  return c->distributed && c->have_comm && c->comm.halo_exchange_begin && c->comm.halo_exchange_end;
}
int exchange_end(bcg_context* c) {
  if (c->comm.halo_exchange_end(c->comm.user) != 0) BCG_FAIL(c, BCG_ERR_COMM, "halo_exchange_end callback failed");
  return BCG_OK;
}

int halo_field(bcg_context* c, const bcg_field* f, bool split = false) {
  if (!c->distributed) return BCG_OK;
  const size_t site_bytes = static_cast<size_t>(3) * f->m * sizeof(double2);
  BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
  if (f->parity >= 0) {
    // a half-volume field: every face holds half its sites (kernels_generic.hip: k_pack_faces_half), at half the offsets
    // of the full plan -- the same messages with half the bytes per site (face sizes are even: every extent is)
    {
      ProfScope ps(c, "pack_faces");
      bcg::launch_pack_faces_half(c->stream, f->m, c->lat, f->parity, f->d, c->halo_send);
    }
    BCG_TRY(check_launch(c, "pack_faces"));
    ProfScope ps(c, split ? "halo_exchange_begin" : "halo_exchange");
    return exchange_faces(c, site_bytes / 2, split);
  }
  {
    ProfScope ps(c, "pack_faces");
    bcg::launch_pack_faces(c->stream, f->m, c->lat, f->d, c->halo_send);
  }
  BCG_TRY(check_launch(c, "pack_faces"));
  ProfScope ps(c, split ? "halo_exchange_begin" : "halo_exchange");
  return exchange_faces(c, site_bytes, split);
}

// Faces of the x3 slices [x3_lo, x3_lo + x3_n) only; `d` is a whole field (ring = 0) or a ring of slices (capacity mode).
// The other slices' ranges of the ghost buffer keep what they held.
// x3b_n > 0: and those of a second range of slices, in the same exchange
// parity >= 0: `d` is a half-volume field of that parity (whole, ring = 0): half faces, half the bytes per site (halo_field)
int halo_window(bcg_context* c, int m, const double2* d, int x3_lo, int x3_n, int ring, bool split = false, int x3b_lo = 0,
                int x3b_n = 0, int parity = -1) {
  if (!c->distributed) return BCG_OK;
  const size_t site_bytes = static_cast<size_t>(3) * m * sizeof(double2);
  BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
  {
    ProfScope ps(c, "pack_faces");
    if (parity >= 0) {
      bcg::launch_pack_faces_half(c->stream, m, c->lat, parity, d, c->halo_send, x3_lo, x3_n);
      if (x3b_n > 0) bcg::launch_pack_faces_half(c->stream, m, c->lat, parity, d, c->halo_send, x3b_lo, x3b_n);
    } else {
      bcg::launch_pack_faces(c->stream, m, c->lat, d, c->halo_send, x3_lo, x3_n, ring);
      if (x3b_n > 0) bcg::launch_pack_faces(c->stream, m, c->lat, d, c->halo_send, x3b_lo, x3b_n, ring);
    }
  }
  BCG_TRY(check_launch(c, "pack_faces"));
  ProfScope ps(c, split ? "halo_exchange_begin" : "halo_exchange");
  return exchange_faces(c, parity >= 0 ? site_bytes / 2 : site_bytes, split, x3_lo, x3_n, x3b_lo, x3b_n);
}

// Capacity mode with overlapped exchanges: the received faces of slice x3 = 0 of every split direction, saved aside
// (save) or put back (!save).  The ghost ranges of slice 0 are re-used for the faces of `tmp` while the source's faces of
// that slice are needed once more at the end of the sweep (apply_shifted_ring).
int ensure_halo_save(bcg_context* c, size_t total) {
  if (total <= c->halo_save_bytes) return BCG_OK;
  BCG_TRY(stream_sync(c));
  if (c->halo_save) (void)hipFree(c->halo_save);
  c->halo_save = nullptr;
  c->halo_save_bytes = 0;
  HIP_TRY(c, hipMalloc(&c->halo_save, total));
  c->halo_save_bytes = total;
  return BCG_OK;
}
int slice0_faces(bcg_context* c, size_t site_bytes, bool save) {
  int peer_s[8], peer_r[8];
  size_t off_s[8], off_r[8], nb[8];
  const int n = halo_plan(c->ndim, c->gdims, c->grid, c->coords, site_bytes, peer_s, peer_r, off_s, off_r, nb, nullptr);
  if (n < 0) BCG_FAIL(c, BCG_ERR_INVALID, "halo plan");
  size_t total = 0;
  for (int k = 0; k < n; ++k) total += nb[k] / c->lat.L[3];
  BCG_TRY(ensure_halo_save(c, total));
  size_t at = 0;
  for (int k = 0; k < n; ++k) {
    const size_t each = nb[k] / c->lat.L[3];
    char* const ghost = reinterpret_cast<char*>(c->halo_recv) + off_r[k];
    char* const keep = reinterpret_cast<char*>(c->halo_save) + at;
    HIP_TRY(c, hipMemcpyAsync(save ? keep : ghost, save ? ghost : keep, each, hipMemcpyDeviceToDevice, c->stream));
    at += each;
  }
  return BCG_OK;
}

int halo_gauge(bcg_context* c, bcg_gauge* g) {
  if (!c->distributed || g->ghost_valid) return BCG_OK;
  const size_t site_bytes = 9 * sizeof(double2);
  BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
  bcg::launch_pack_gauge_faces(c->stream, c->lat, g->U, c->halo_send);
  BCG_TRY(check_launch(c, "pack_gauge_faces"));
  BCG_TRY(exchange_faces(c, site_bytes));
  HIP_TRY(c, hipMemcpyAsync(g->Ughost, c->halo_recv, static_cast<size_t>(c->ghost_sites) * site_bytes,
                            hipMemcpyDeviceToDevice, c->stream));
  BCG_TRY(stream_sync(c));
  g->ghost_valid = true;
  return BCG_OK;
}

// ---- building blocks ---------------------------------------------------------------------------
bool same_shape(const bcg_field* a, const bcg_field* b) {
  return a && b && a->ctx == b->ctx && a->m == b->m && a->parity == b->parity;
}

inline bool fast_rows(const bcg_context* c, int m) { return !c->force_generic && bcg::mfma_width(m); }       // + Gram, phase B
inline bool fast_rmul(const bcg_context* c, int m) { return !c->force_generic && bcg::mfma_rows_width(m); }  // products, phase C
inline bool fast_hop(const bcg_context* c, int m) { return !c->force_generic && bcg::hop_fast_width(m); }
constexpr int kFastBlocks = 1024;  // persistent-style grids: 4 blocks per CU

// The boundary tiles (a site of the tile has a neighbour in a ghost face) of the tiling with `spb` sites per tile, in
// lexicographic order; built once per tile length.  The boundary launch deals them to its blocks round-robin.
int boundary_tile_list(bcg_context* c, int spb, const int** list, int* n) {
  auto it = c->boundary_tiles.find(spb);
  if (it == c->boundary_tiles.end()) {
    const bcg::LatticeDev& L = c->lat;
    std::vector<int> tiles;
    for (int x3 = 0; x3 < L.L[3]; ++x3)
      for (int x2 = 0; x2 < L.L[2]; ++x2)
        for (int x1 = 0; x1 < L.L[1]; ++x1) {
          const bool b123 = (L.split[1] && (x1 == 0 || x1 == L.L[1] - 1)) || (L.split[2] && (x2 == 0 || x2 == L.L[2] - 1)) ||
                            (L.split[3] && (x3 == 0 || x3 == L.L[3] - 1));
          for (int x0b = 0; x0b < L.L[0]; x0b += spb)
            if (b123 || (L.split[0] && (x0b == 0 || x0b + spb == L.L[0])))
              tiles.push_back(x0b + L.L[0] * (x1 + L.L[1] * (x2 + L.L[2] * x3)));
        }
    int* dev = nullptr;
    if (!tiles.empty()) {
      HIP_TRY(c, hipMalloc(&dev, tiles.size() * sizeof(int)));
      HIP_TRY(c, hipMemcpy(dev, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    it = c->boundary_tiles.emplace(spb, std::make_pair(dev, static_cast<int>(tiles.size()))).first;
  }
  *list = it->second.first;
  *n = it->second.second;
  return BCG_OK;
}

// Profiling only: count the launches of each form of the stencil kernel ("stencil_form_k_hop4c" ...), so that tests
// and tuning runs can tell which one a lattice shape gets.
void note_stencil_form(bcg_context* c, int m, int tile_class, const bcg::HopWindow& win, bool plain = false) {
  if (!c->profiling) return;
  static const char* names[] = {"stencil_form_general", "stencil_form_k_hop4", "stencil_form_k_hop4c", "stencil_form_k_hop4b"};
  int form = bcg::hop_kernel_form(m, c->lat, kFastBlocks, c->hop_tune, tile_class, win);
  if (form == 2 && bcg::hop_uses_bundle(m, c->lat, kFastBlocks, c->hop_tune, tile_class, win, plain)) form = 3;
  if (form >= 0 && form <= 3) c->prof[names[form]].count += 1;
}

// out = D in  (HOP_PLAIN)  or  out = c0*p - D in  (HOP_SHIFTED).  With gram_blocks != nullptr (m = 16 fast
// path, HOP_SHIFTED) the kernel also leaves block partials of p^dagger out in c->partials.
int hop(bcg_context* c, const bcg_gauge* g, bcg_field* out, const bcg_field* in, bcg::HopMode mode, const bcg_field* p,
        double c0, int* gram_blocks = nullptr, bool* gram_folded = nullptr) {
  if (in->parity >= 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "D alone maps a half-volume field to the other parity: use bcg_dirac_hop_half");
  BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
  const int m = in->m;
  if (gram_blocks) *gram_blocks = 0;
  if (gram_folded) *gram_folded = false;
  const bool fast = fast_hop(c, m);
  // fused Gram product: m = 16 in every form of the specialised stencil; m = 8 in the column-sweep kernel, one launch
  const bool split_path = fast && bcg::hop_can_split_tiles(m, c->lat) && can_overlap(c);
  const bool gram = fast && gram_blocks && mode == bcg::HOP_SHIFTED &&
                    (m == 16 || (m == 8 && !split_path &&
                                 bcg::hop_kernel_form(m, c->lat, kFastBlocks, c->hop_tune, 0, bcg::HopWindow()) == 2));
  const char* name = gram ? "hop_shifted_gram" : (mode == bcg::HOP_PLAIN ? "hop" : "hop_shifted");
  if (fast) BCG_TRY(ensure_scratch(c));
  // BCG_FORCE_TILE_CLASSES=1 (tuning aid): take the two-launch path on an undivided lattice too, where every tile is
  // an interior one, to time the interior-class kernel on one GPU
  const bool force_classes = c->force_tile_classes;
  if (fast && bcg::hop_can_split_tiles(m, c->lat) && (can_overlap(c) || (force_classes && !c->distributed))) {
    // pack -> post the exchange -> interior tiles (no ghost reads) -> wait for the exchange -> boundary tiles
    if (c->distributed) BCG_TRY(halo_field(c, in, /*split=*/true));
    bcg::HopTuning tune = c->hop_tune;
    tune.blocks = tune.blocks_overlap;  // leave some CUs to the transport's kernels while it runs
    int nb1, nb2;
    note_stencil_form(c, m, 1, bcg::HopWindow());
    {
      ProfScope ps(c, name, alg_bytes(c, m, mode == bcg::HOP_PLAIN ? 2 : 3, 1), hop_flops(c, m, gram));  // both tile classes: counted here
      nb1 = bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                                 p ? p->d : nullptr, c0, c->partials, gram, kFastBlocks, tune, /*interior*/ 1);
    }
    BCG_TRY(check_launch(c, name));
    if (c->distributed) {
      ProfScope ps(c, "halo_exchange_end");
      BCG_TRY(exchange_end(c));
    }
    {
      bcg::HopTuning tb = c->hop_tune;
      BCG_TRY(boundary_tile_list(c, 4 * (64 / m), &tb.boundary_list, &tb.boundary_n));
      if (tb.boundary_n == 0) tb.boundary_list = nullptr;  // nothing to do: fall through to an empty class launch
      ProfScope ps(c, "hop_boundary");
      nb2 = tb.boundary_n == 0 ? 0
                               : bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                                                      p ? p->d : nullptr, c0,
                                                      gram ? c->partials + static_cast<size_t>(nb1) * m * m : c->partials, gram,
                                                      kFastBlocks, tb, /*boundary*/ 2);
    }
    if (gram) *gram_blocks = nb1 + nb2;
    return check_launch(c, "hop_boundary");
  }
  BCG_TRY(halo_field(c, in));
  if (fast) {
    note_stencil_form(c, m, 0, bcg::HopWindow(), mode == bcg::HOP_PLAIN);
    bcg::HopTuning tune = c->hop_tune;
    // one whole launch of a column form: the kernel's last blocks sum the Gram partials themselves (no reduction launch)
    const bool fold = gram && gram_folded && bcg::hop_folds_gram(m, c->lat, kFastBlocks, tune, bcg::HopWindow());
    if (fold) tune.fold = bcg::GramFold{c->dev_gram, c->fold_tickets};
    ProfScope ps(c, name, alg_bytes(c, m, mode == bcg::HOP_PLAIN ? 2 : 3, 1), hop_flops(c, m, gram));
    const int nb = bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                                        p ? p->d : nullptr, c0, c->partials, gram, kFastBlocks, tune, 0);
    if (gram) *gram_blocks = nb;
    if (fold) *gram_folded = true;
  } else {
    ProfScope ps(c, name, alg_bytes(c, m, mode == bcg::HOP_PLAIN ? 2 : 3, 1), hop_flops(c, m, gram));
    bcg::launch_hop_generic(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                            p ? p->d : nullptr, c0);
  }
  return check_launch(c, "hop");
}

// Block partials in c->partials -> G (m x m), summed over blocks in a fixed order and over ranks,
// Hermitian-mirrored exactly as inc/fields.hpp:115-120.
// folded: the producing kernel has already summed them into c->dev_gram (bcg::GramFold)
int finish_gram(bcg_context* c, int m, int nblocks, CMat& G, bool mirror, bool folded = false) {
  if (!folded) {
    ProfScope ps(c, "reduce_partials");
    bcg::launch_reduce_partials(c->stream, m * m, nblocks, c->partials, c->dev_gram);
  }
  BCG_TRY(check_launch(c, "reduce_partials"));
  if (c->distributed) {
    if (!c->have_comm || !c->comm.allreduce_sum) BCG_FAIL(c, BCG_ERR_COMM, "lattice is split over ranks but no bcg_comm was set");
    ProfScope ps(c, "allreduce");
    if (c->comm.allreduce_sum(c->comm.user, c->dev_gram, static_cast<size_t>(2) * m * m) != 0)
      BCG_FAIL(c, BCG_ERR_COMM, "allreduce_sum callback failed");
  }
  HIP_TRY(c, hipMemcpyAsync(c->pin_gram, c->dev_gram, static_cast<size_t>(m) * m * sizeof(double2),
                            hipMemcpyDeviceToHost, c->stream));
  BCG_TRY(stream_sync(c));
  G = CMat(m, c->pin_gram);
  if (mirror)
    for (int i = 1; i < m; ++i)
      for (int j = 0; j < i; ++j) G(j, i) = std::conj(G(i, j));
  return BCG_OK;
}

// G = a^dagger b
int gram(bcg_context* c, const bcg_field* a, const bcg_field* b, CMat& G, bool mirror = true) {
  BCG_TRY(ensure_scratch(c));
  const int m = a->m;
  int nblocks;
  {
    ProfScope ps(c, a == b ? "gram_self" : "gram_pair", alg_bytes(c, m, a == b ? 1 : 2));
    if (fast_rows(c, m)) nblocks = bcg::launch_gram_mfma(c->stream, m, rows_of(a), a->d, b->d, c->partials, kFastBlocks);
    else nblocks = bcg::launch_gram_generic(c->stream, m, rows_of(a), a->d, b->d, c->partials, kMaxGramBlocks);
  }
  BCG_TRY(check_launch(c, "gram"));
  return finish_gram(c, m, nblocks, G, mirror);
}

int rmul(bcg_context* c, bcg_field* y, const bcg_field* x, const CMat& M, double b, bcg::RmulMode mode, const char* name) {
  const double2* Md;
  BCG_TRY(upload_mat(c, M, &Md));
  {
    ProfScope ps(c, name, alg_bytes(c, y->m, (x && x != y) ? 3 : 2));
    if (fast_rmul(c, y->m)) bcg::launch_rmul_mfma(c->stream, y->m, rows_of(y), y->d, x ? x->d : nullptr, Md, b, mode, kFastBlocks);
    else bcg::launch_rmul_generic(c->stream, y->m, rows_of(y), y->d, x ? x->d : nullptr, Md, b, mode);
  }
  return check_launch(c, name);
}

int trisolve(bcg_context* c, bcg_field* y, const CMat& R) {
  const double2* Rd;
  BCG_TRY(upload_mat(c, R, &Rd));
  {
    ProfScope ps(c, "trisolve", alg_bytes(c, y->m, 2));
    bcg::launch_trisolve_generic(c->stream, y->m, rows_of(y), y->d, Rd);
  }
  return check_launch(c, "trisolve");
}

int axpby(bcg_context* c, bcg_field* y, double a, const bcg_field* x, double b, const char* name) {
  {
    ProfScope ps(c, name);
    bcg::launch_axpby(c->stream, y->d, a, x->d, b, rows_of(y) * y->m);
  }
  return check_launch(c, name);
}

int get_tmp(bcg_context* c, int m, bcg_field** out) {
  auto it = c->tmp_field.find(m);
  if (it != c->tmp_field.end()) {
    *out = it->second;
    return BCG_OK;
  }
  bcg_field* f = nullptr;
  BCG_TRY(bcg_field_create(c, m, &f));
  c->tmp_field[m] = f;
  *out = f;
  return BCG_OK;
}
// a new field of the width, parity and site count of `like`
int create_like(bcg_context* c, const bcg_field* like, bcg_field** out) {
  return like->parity >= 0 ? bcg_field_create_half(c, like->m, like->parity, out) : bcg_field_create(c, like->m, out);
}
// the half-volume `tmp` of parity `parity` (= D applied to a field of the other parity)
int get_tmp_half(bcg_context* c, int m, int parity, bcg_field** out) {
  const int key = m + 1000 * (1 + parity);
  auto it = c->tmp_field.find(key);
  if (it != c->tmp_field.end()) {
    *out = it->second;
    return BCG_OK;
  }
  bcg_field* f = nullptr;
  BCG_TRY(bcg_field_create_half(c, m, parity, &f));
  c->tmp_field[key] = f;
  *out = f;
  return BCG_OK;
}

// Capacity mode (bcg_capacity_mode): the same T = (mass^2 + sigma0) P - D(D(P)), with tmp = D P held as a ring of R x3
// slices instead of a whole field.  Direction 3 is undivided, so a slice of T needs the slices x3-1, x3, x3+1 of tmp and
// nothing else of it: the first stencil runs C = R - 2 slices ahead of the second.
//   tmp[L3-1]; then per chunk [lo, hi) of C slices: tmp[.. hi] (slice L3 = slice 0 again), faces of tmp[lo, hi) to the
//   neighbours, T[lo, hi).  Writing slice s of tmp replaces slice s - R, which no later chunk reads.
// Slices L3-1 and 0 of tmp are computed twice (2/L3 more work in the first stencil).  The ghost buffer is shared: the faces of
// tmp[lo, hi) land on the range that held the faces of P[lo, hi), which the first stencil no longer reads -- except
// slice 0 at the very end, whose P faces are exchanged again (serial form) or restored from a copy (overlapped form, below).
bool capacity_path(const bcg_context* c, int m) {
  return c->tmp_ring > 0 && fast_hop(c, m) && bcg::hop_can_split_tiles(m, c->lat);
}
// chunk length of the capacity-mode sweep and whether its exchanges are overlapped (apply_shifted_ring)
inline bool ring_overlapped(const bcg_context* c) { return c->ring_overlap && can_overlap(c) && (c->tmp_ring - 2) / 2 >= 1; }
inline int ring_chunk(const bcg_context* c) {
  const int most = ring_overlapped(c) ? (c->tmp_ring - 2) / 2 : c->tmp_ring - 2;
  // BCG_RING_CHUNK (tests, tuning): shorter chunks than the ring allows -- e.g. the overlapped form's 15-slice windows of
  // ring 32 on a single rank, where the serial form would sweep 30 slices at a time
  return c->ring_chunk_override > 0 && c->ring_chunk_override < most ? c->ring_chunk_override : most;
}
// Everything capacity mode allocates for width m: the ring, the block partials of all chunks side by side (the stencil
// grid stays the tuned one: a smaller grid loses the x3 walk), the face buffers and the copy of the slice-0 faces.
int ensure_ring_scratch(bcg_context* c, int m) {
  const int R = c->tmp_ring, L3 = c->lat.L[3], C = ring_chunk(c);
  double2*& ring = c->tmp_ring_buf[m];
  if (!ring) HIP_TRY(c, hipMalloc(&ring, static_cast<size_t>(R) * c->lat.stride[3] * 3 * m * sizeof(double2)));
  BCG_TRY(ensure_scratch(c));
  const int chunks = (L3 + C - 1) / C;
  const size_t need = static_cast<size_t>(c->hop_tune.blocks > 0 ? c->hop_tune.blocks : kFastBlocks) * chunks * m * m * sizeof(double2);
  if (m == 16 && need > c->partials_bytes) {
    BCG_TRY(stream_sync(c));
    (void)hipFree(c->partials);
    c->partials = nullptr;
    c->partials_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->partials, need));
    c->partials_bytes = need;
  }
  if (c->distributed) {
    const size_t site_bytes = static_cast<size_t>(3) * m * sizeof(double2);
    BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
    if (ring_overlapped(c)) BCG_TRY(ensure_halo_save(c, static_cast<size_t>(c->ghost_sites) / L3 * site_bytes));
  }
  return BCG_OK;
}
// Half-volume fields on a lattice divided over ranks, direction 3 undivided: the same sweep in chunks of x3 slices on a WHOLE
// tmp (half field; no ring), for the sake of its overlapped exchanges -- the faces of the source in two windows, those of tmp
// chunk by chunk, each travelling while the neighbouring chunks are computed (apply_shifted_ring with half_tmp set).
inline int half_chunk(const bcg_context* c) { return c->half_chunk_override > 0 ? c->half_chunk_override : 16; }
inline bool half_chunked_path(const bcg_context* c) {
  // (half_chunk_force: BCG_HALF_CHUNK_FORCE=1, a tuning aid -- the chunked sweep on one GPU, to time what the chunks cost)
  return ((c->distributed && can_overlap(c)) || c->half_chunk_force) && c->ndim == 4 && !c->lat.split[3] && !c->lat.split[0] &&
         c->lat.L[3] > half_chunk(c);
}
int ensure_half_chunk_scratch(bcg_context* c, int m) {
  BCG_TRY(ensure_scratch(c));
  const int C = half_chunk(c), chunks = (c->lat.L[3] + C - 1) / C;
  const size_t need = static_cast<size_t>(c->hop_tune.blocks > 0 ? c->hop_tune.blocks : kFastBlocks) * chunks * m * m * sizeof(double2);
  if (m == 16 && need > c->partials_bytes) {  // the block partials of all chunks side by side, as in capacity mode
    BCG_TRY(stream_sync(c));
    (void)hipFree(c->partials);
    c->partials = nullptr;
    c->partials_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->partials, need));
    c->partials_bytes = need;
  }
  return ensure_halo(c, static_cast<size_t>(c->ghost_sites) * 3 * m * sizeof(double2));
}
int apply_shifted_ring(bcg_context* c, const bcg_gauge* g, double mass, double sigma0, bcg_field* T, const bcg_field* P,
                       int* gram_blocks, bcg_field* half_tmp = nullptr) {
  const bool half = half_tmp != nullptr;  // P, T: half fields of one parity, half_tmp: the whole tmp of the other
  const int m = P->m, L3 = c->lat.L[3], R = half ? L3 : c->tmp_ring;
  // Overlapped form (ranks that exchange faces, split callbacks present, ring of at least 2 C + 2 slices): the exchange of
  // chunk k's tmp faces runs while the first stencil works on chunk k + 1 and the second one on chunk k - 1, so the ring
  // holds two chunks and the two boundary slices.  Otherwise C = R - 2 and every exchange is waited for where it is posted.
  const bool overlap = half ? can_overlap(c) : ring_overlapped(c);
  const int C = half ? half_chunk(c) : ring_chunk(c);
  BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
  if (half) BCG_TRY(ensure_half_chunk_scratch(c, m));
  else BCG_TRY(ensure_ring_scratch(c, m));
  double2* const ring = half ? half_tmp->d : c->tmp_ring_buf[m];
  if (gram_blocks) *gram_blocks = 0;
  const bool gram = gram_blocks && m == 16;
  const bcg::HopTuning& tune = c->hop_tune;
  const size_t site_bytes = static_cast<size_t>(3) * m * sizeof(double2) / (half ? 2 : 1);  // (of the face messages)
  // half fields: the compact lattice with the half ghost faces' offsets (apply_shifted), windows without ring addressing
  bcg::LatticeDev lat = c->lat;
  if (half) {
    lat.L[0] /= 2;
    lat.V /= 2;
    for (int mu = 1; mu < 4; ++mu) lat.stride[mu] /= 2;
    for (int mu = 0; mu < 4; ++mu) {
      lat.face_sites[mu] /= 2;
      lat.ghost_off[mu][0] /= 2;
      lat.ghost_off[mu][1] /= 2;
    }
  }
  const int par_p = half ? P->parity : -1, par_t = half ? half_tmp->parity : -1;
  const int vden = half ? 2 * L3 : L3;  // a window's share of the full local volume
  auto window = [&](int lo, int n, int parity_out) {
    bcg::HopWindow w;
    w.x3_lo = lo;
    w.x3_n = n;
    w.ring = half ? 0 : R;
    w.cb = half ? 1 : 0;
    w.cb_parity = half ? parity_out : 0;
    return w;
  };
  // The source's faces.  Serial form: one blocking exchange of the whole field.  Overlapped form: nothing blocks -- the
  // faces of the slices the first launches read (the wrap slice L3 - 1 and slices 0 .. C, tmp up to one slice past the
  // first chunk) go first, the rest behind them as a second outstanding exchange that travels while those launches run and is ended
  // in front of the first launch that reads it (the transport ends exchanges in the order they began).
  bool p_rest_pending = false;
  // Split exchanges begun and not yet ended.  The transports keep FIFO state per begin (comm_rccl.cpp: begun / ended and
  // the `arrived` events; TorchDistComm: its pending list), so an error return between a begin and its end must not leave
  // an entry behind -- the next exchange on this context would pop the stale one and read ghosts before they arrive.
  // Every error exit of this function therefore ends what it began (the context and its transport stay usable).
  struct OutstandingExchanges {
    bcg_context* c;
    int n = 0;
    ~OutstandingExchanges() {
      const std::string why = c->err;
      for (; n > 0; --n) (void)c->comm.halo_exchange_end(c->comm.user);
      c->err = why;
    }
  } outstanding{c};
  auto begin_window = [&](const double2* d, int lo, int n, int ring_slots, int b_lo, int b_n, int parity) -> int {
    BCG_TRY(halo_window(c, m, d, lo, n, ring_slots, /*split=*/true, b_lo, b_n, parity));
    if (c->distributed) outstanding.n += 1;
    return BCG_OK;
  };
  auto end_oldest = [&]() -> int {
    ProfScope ps(c, "halo_exchange_end");
    BCG_TRY(exchange_end(c));
    outstanding.n -= 1;
    return BCG_OK;
  };
  if (overlap && c->distributed) {
    const int n1 = (C + 1 < L3 - 1) ? C + 1 : L3 - 1;  // slices [0, n1) and slice L3 - 1
    BCG_TRY(begin_window(P->d, 0, n1, 0, L3 - 1, 1, par_p));
    if (n1 < L3 - 1) {
      BCG_TRY(begin_window(P->d, n1, L3 - 1 - n1, 0, 0, 0, par_p));
      p_rest_pending = true;
    }
    BCG_TRY(end_oldest());
  } else {
    BCG_TRY(halo_field(c, P));
  }
  // (a whole tmp keeps its slice 0: nothing is computed twice at the end of the sweep, no faces to put back)
  if (overlap && !half) BCG_TRY(slice0_faces(c, site_bytes, /*save=*/true));
  auto first = [&](int lo, int n) -> int {  // tmp[lo, lo+n) = D P
    if (half) {
      if (c->profiling) c->prof["stencil_form_k_hop4b_checkerboard"].count += 1;
    } else {
      note_stencil_form(c, m, 0, window(lo, n, 0), /*plain=*/true);
    }
    ProfScope ps(c, half ? "hop_half" : "hop_ring", alg_bytes(c, m, 2, 1, n, vden), hop_flops(c, m, false, n, vden));
    const int nb = bcg::launch_hop_fast(c->stream, m, lat, g->U, g->Ughost, P->d, c->halo_recv, ring, bcg::HOP_PLAIN, nullptr,
                                        0.0, c->partials, false, kFastBlocks, tune, 0, window(lo, n, par_t));
    if (nb < 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "capacity mode: stencil window rejected");
    return check_launch(c, "hop_ring");
  };
  const double c0 = mass * mass + sigma0;
  int total = 0;
  auto second = [&](int lo, int hi) -> int {  // T[lo, hi) from tmp[lo - 1, hi]
    {
      if (half && c->profiling) c->prof["stencil_form_k_hop4b_checkerboard"].count += 1;
      ProfScope ps(c, half ? (gram ? "hop_half_shifted_gram" : "hop_half_shifted") : (gram ? "hop_shifted_gram_ring" : "hop_shifted_ring"),
                   alg_bytes(c, m, 3, 1, hi - lo, vden), hop_flops(c, m, gram, hi - lo, vden));
      const int nb = bcg::launch_hop_fast(c->stream, m, lat, g->U, g->Ughost, ring, c->halo_recv, T->d, bcg::HOP_SHIFTED, P->d,
                                          c0, c->partials + static_cast<size_t>(total) * m * m, gram, kFastBlocks, tune, 0,
                                          window(lo, hi - lo, par_p));
      if (nb < 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "capacity mode: stencil window rejected");
      total += nb;
    }
    return check_launch(c, "hop_shifted_ring");
  };
  BCG_TRY(first(L3 - 1, 1));
  int next = 0;  // first slice of tmp not yet computed in order (L3 stands for slice 0 again)
  // tmp up to one slice past chunk [lo, hi); at the end of the sweep slice L3 = slice 0 again, from the source's faces of
  // slice 0: exchanged again (serial form) or put back from their copy (overlapped form: no second exchange in flight)
  auto stage_first = [&](int lo) -> int {
    const int hi = lo + C < L3 ? lo + C : L3;
    const int last = hi < L3 ? hi : L3 - 1;
    if (next <= last) BCG_TRY(first(next, last - next + 1));
    if (hi == L3 && !half) {
      if (overlap) BCG_TRY(slice0_faces(c, site_bytes, /*save=*/false));
      else BCG_TRY(halo_window(c, m, P->d, 0, 1, 0));  // its P faces were replaced by tmp faces of the first chunk
      BCG_TRY(first(0, 1));
    }
    next = hi + 1;
    return BCG_OK;
  };
  if (!overlap) {
    for (int lo = 0; lo < L3; lo += C) {
      const int hi = lo + C < L3 ? lo + C : L3;
      BCG_TRY(stage_first(lo));
      BCG_TRY(halo_window(c, m, ring, lo, hi - lo, half ? 0 : R, false, 0, 0, par_t));
      BCG_TRY(second(lo, hi));
    }
  } else {
    BCG_TRY(stage_first(0));
    BCG_TRY(begin_window(ring, 0, (C < L3 ? C : L3), half ? 0 : R, 0, 0, par_t));
    for (int lo = 0; lo < L3; lo += C) {
      const int hi = lo + C < L3 ? lo + C : L3;
      if (p_rest_pending) {  // the rest of the source's faces: posted before chunk 0's tmp faces, so ended before them
        BCG_TRY(end_oldest());
        p_rest_pending = false;
      }
      if (hi < L3) BCG_TRY(stage_first(hi));  // chunk k + 1's slices of tmp, while chunk k's faces are on the links
      BCG_TRY(end_oldest());
      if (hi < L3) BCG_TRY(begin_window(ring, hi, (hi + C < L3 ? C : L3 - hi), half ? 0 : R, 0, 0, par_t));
      BCG_TRY(second(lo, hi));  // ... and chunk k + 1's faces fly while chunk k's T is computed
    }
  }
  if (gram) *gram_blocks = total;
  return BCG_OK;
}

// The device memory apply_shifted needs for operands shaped like `like`, allocated now rather than at the first call
int reserve_operator_scratch(bcg_context* c, const bcg_field* like) {
  const int m = like->m;
  bcg_field* tmp;
  if (like->parity >= 0) {
    BCG_TRY(get_tmp_half(c, m, 1 - like->parity, &tmp));
    if (c->distributed) BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * 3 * m * sizeof(double2)));
    if (half_chunked_path(c) && fast_hop(c, m)) BCG_TRY(ensure_half_chunk_scratch(c, m));
    return BCG_OK;
  }
  if (fast_hop(c, m)) BCG_TRY(ensure_scratch(c));
  if (capacity_path(c, m)) return ensure_ring_scratch(c, m);
  BCG_TRY(get_tmp(c, m, &tmp));
  if (c->distributed) BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * 3 * m * sizeof(double2)));
  return BCG_OK;
}

// T = (mass^2 + sigma0) P - D(D(P))   [op + add(P, sigma0), inc/block_solvers.hpp:134-136]
int apply_shifted(bcg_context* c, const bcg_gauge* g, double mass, double sigma0, bcg_field* T, const bcg_field* P,
                  int* gram_blocks = nullptr, bool* gram_folded = nullptr) {
  if (gram_folded) *gram_folded = false;
  if (P->parity >= 0) {  // A restricted to one parity: tmp (other parity) = D P, T = (mass^2 + sigma0) P - D tmp
    if (gram_blocks) *gram_blocks = 0;
    if (T->parity != P->parity) BCG_FAIL(c, BCG_ERR_INVALID, "half-volume operator: result and argument must have the same parity");
    const int m = P->m;
    bcg_field* tmp;
    BCG_TRY(get_tmp_half(c, m, 1 - P->parity, &tmp));
    BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
    // the bundle sweep in its checkerboard form (m = 16, compact row a multiple of the tile, patch walk), else the generic kernel
    bcg::LatticeDev latc = c->lat;
    latc.L[0] /= 2;
    latc.V /= 2;
    for (int mu = 1; mu < 4; ++mu) latc.stride[mu] /= 2;
    for (int mu = 0; mu < 4; ++mu) {  // half ghost faces: half the sites at half the offsets, compact in x0 like the field
      latc.face_sites[mu] /= 2;
      latc.ghost_off[mu][0] /= 2;
      latc.ghost_off[mu][1] /= 2;
    }
    // (direction 0 divided over ranks: the compact row's end sites would need the ghost face in one row parity only -- generic kernel)
    const bool fast = fast_hop(c, m) && (m == 16 || m == 32) && c->ndim == 4 && latc.L[0] > 0 && !c->lat.split[0] &&
                      bcg::hop_can_split_tiles(m, latc);
    // direction 3 whole, split exchange available: the sweep in x3 chunks with every exchange overlapped -- provided the
    // checkerboard bundle sweep takes EVERY window the chunked sweep launches (1, C and C + 1 slices and the last, shorter
    // chunk).  The chunked sweep has no generic fallback once its first exchange is posted; a tuning that switches the
    // bundle walk off (BCG_HOP_BUNDLE=0, an odd BCG_HOP_PATCH) or a slice too small for the grid lands in the blocking
    // path below, which falls back to k_hop_half.
    bool chunk_windows_ok = fast && half_chunked_path(c);
    if (chunk_windows_ok) {
      const int C = half_chunk(c), L3 = c->lat.L[3];
      for (int n = 1; n <= std::min(C + 1, L3) && chunk_windows_ok; ++n) {
        bcg::HopWindow w;
        w.x3_lo = 0;
        w.x3_n = n;
        w.cb = 1;
        chunk_windows_ok = bcg::hop_uses_bundle(m, latc, kFastBlocks, c->hop_tune, 0, w, /*plain=*/true);
      }
    }
    if (chunk_windows_ok) {
      int nb = 0;
      BCG_TRY(apply_shifted_ring(c, g, mass, sigma0, T, P, gram_blocks ? &nb : nullptr, tmp));
      if (gram_blocks) *gram_blocks = nb;
      return BCG_OK;
    }
    BCG_TRY(halo_field(c, P));  // (a lattice divided over ranks: the half faces of the source, then below those of tmp)
    int nb1 = -1, nb2 = -1;
    if (fast) {
      BCG_TRY(ensure_scratch(c));
      bcg::HopWindow w;
      w.cb = 1;
      w.cb_parity = tmp->parity;
      {
        ProfScope ps(c, "hop_half", alg_bytes(c, m, 2, 1, 1, 2), hop_flops(c, m, false, 1, 2));
        nb1 = bcg::launch_hop_fast(c->stream, m, latc, g->U, g->Ughost, P->d, c->halo_recv, tmp->d, bcg::HOP_PLAIN, nullptr, 0.0,
                                   c->partials, false, kFastBlocks, c->hop_tune, 0, w);
      }
      if (nb1 >= 0) {
        BCG_TRY(check_launch(c, "hop_half"));
        BCG_TRY(halo_field(c, tmp));
        const bool gram = gram_blocks != nullptr && m == 16;  // the fused product exists at m = 16 (as in the full-volume sweep)
        bcg::HopTuning tune = c->hop_tune;
        const bool fold = gram && gram_folded;
        if (fold) tune.fold = bcg::GramFold{c->dev_gram, c->fold_tickets};
        w.cb_parity = T->parity;
        {
          ProfScope ps(c, gram ? "hop_half_shifted_gram" : "hop_half_shifted", alg_bytes(c, m, 3, 1, 1, 2), hop_flops(c, m, gram, 1, 2));
          nb2 = bcg::launch_hop_fast(c->stream, m, latc, g->U, g->Ughost, tmp->d, c->halo_recv, T->d, bcg::HOP_SHIFTED, P->d,
                                     mass * mass + sigma0, c->partials, gram, kFastBlocks, tune, 0, w);
        }
        if (nb2 < 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "half-volume operator: second stencil rejected after the first ran");
        BCG_TRY(check_launch(c, "hop_half_shifted"));
        if (gram) {
          *gram_blocks = nb2;
          if (fold) *gram_folded = true;
        }
        if (c->profiling) c->prof["stencil_form_k_hop4b_checkerboard"].count += 2;
        return BCG_OK;
      }
    }
    {
      ProfScope ps(c, "hop_half", alg_bytes(c, m, 2, 1, 1, 2), hop_flops(c, m, false, 1, 2));
      bcg::launch_hop_half(c->stream, m, c->lat, tmp->parity, g->U, g->Ughost, P->d, c->halo_recv, tmp->d, bcg::HOP_PLAIN, nullptr, 0.0);
    }
    BCG_TRY(check_launch(c, "hop_half"));
    BCG_TRY(halo_field(c, tmp));
    {
      ProfScope ps(c, "hop_half_shifted", alg_bytes(c, m, 3, 1, 1, 2), hop_flops(c, m, false, 1, 2));
      bcg::launch_hop_half(c->stream, m, c->lat, T->parity, g->U, g->Ughost, tmp->d, c->halo_recv, T->d, bcg::HOP_SHIFTED, P->d,
                           mass * mass + sigma0);
    }
    return check_launch(c, "hop_half_shifted");
  }
  if (capacity_path(c, P->m)) return apply_shifted_ring(c, g, mass, sigma0, T, P, gram_blocks);
  bcg_field* tmp;
  BCG_TRY(get_tmp(c, P->m, &tmp));
  BCG_TRY(hop(c, g, tmp, P, bcg::HOP_PLAIN, nullptr, 0.0));
  return hop(c, g, T, tmp, bcg::HOP_SHIFTED, P, mass * mass + sigma0, gram_blocks, gram_folded);
}

// Phase A of an iteration: T = (A + sigma0) P ; G = P^dagger T   (:134-140)
int phase_A(bcg_context* c, const bcg_gauge* g, double mass, double sigma0, bcg_field* T, const bcg_field* P, CMat& G) {
  int nb = 0;
  bool folded = false;
  BCG_TRY(apply_shifted(c, g, mass, sigma0, T, P, &nb, &folded));
  if (nb > 0) return finish_gram(c, P->m, nb, G, true, folded);
  return gram(c, P, T, G);
}

// Phase B: Q -= T alpha ; G2 = Q^dagger Q   (:148 and the Gram half of :152)
int rmul(bcg_context* c, bcg_field* y, const bcg_field* x, const CMat& M, double b, bcg::RmulMode mode, const char* name);
// Deferred normalisation of Q (widths with both fused row kernels and room for a second matrix in phase B's LDS: m = 8,
// 16).  The reference stores Q rho^-1 (:152, multiply_upper_triangular_inverse_RHS) and reads it back twice: for the P
// updates (:158, :177) and for the next iteration's Q -= T alpha (:148).  Here phase C forms Q rho^-1 in registers for
// the P updates and does NOT write it; the un-normalised Q stays in memory and the next phase B multiplies it by the same
// rho^-1 (same kernel arithmetic, same order: bit-identical iterates) before subtracting T alpha.  One field pass less per
// iteration: (1 + 4 S) s in phase C instead of (2 + 4 S) s.  BCG_LAZY_Q=0 switches it off.
bool lazy_q_width(const bcg_context* c, int m) {
  return c->lazy_q && fast_rows(c, m) && fast_rmul(c, m) && (m == 8 || m == 16 || (m == 32 && c->lazy_q > 1));
}

// rinv_prev: the stored Q is the previous iteration's un-normalised block, to be multiplied by this first (nullptr: Q as it is)
// Qout (fused kernel only): the new Q is written there and Q keeps the old block (pair_shifts below)
int phase_B(bcg_context* c, bcg_field* Q, const bcg_field* T, const CMat& alpha, CMat& G2, const CMat* rinv_prev = nullptr,
            bcg_field* Qout = nullptr) {
  const int m = Q->m;
  if (!fast_rows(c, m)) {
    if (Qout) BCG_FAIL(c, BCG_ERR_INVALID, "phase B: a separate output needs the fused kernel");
    BCG_TRY(rmul(c, Q, T, -alpha, 0.0, bcg::RMUL_ADD, "block_axpy"));
    return gram(c, Q, Q, G2);
  }
  const CMat na = -alpha;
  const CMat* two[2] = {&na, rinv_prev};
  const double2* Md;
  BCG_TRY(upload_mats(c, m, two, rinv_prev ? 2 : 1, &Md));
  int nb;
  {
    ProfScope ps(c, "phaseB", row_bytes(Q, 3), product_flops(Q, rinv_prev ? 3 : 2));  // [rho^-1,] alpha, Gram
    nb = bcg::launch_phaseB(c->stream, m, rows_of(Q), Q->d, T->d, Md, c->partials, c->row_blocks_B,
                            bcg::GramFold{c->dev_gram, c->fold_tickets},
                            rinv_prev ? Md + static_cast<size_t>(m) * m : nullptr, Qout ? Qout->d : nullptr);
  }
  BCG_TRY(check_launch(c, "phaseB"));
  return finish_gram(c, m, nb, G2, true, /*folded=*/true);
}

// Phase C: Q <- Q rho^{-1} ; X_s += P_s A_s ; P_s <- P_s B_s + Q for the n active shifts
// (:152 second half, :145, :158, :175, :177)
int trisolve(bcg_context* c, bcg_field* y, const CMat& R);
// rinv_out != nullptr (lazy_q_width): Q rho^-1 is used but not stored; *rinv_out = rho^-1 for the next phase B
int phase_C(bcg_context* c, bcg_field* Q, const CMat& rho, bcg_field* const* X, bcg_field* const* P, int n,
            const std::vector<CMat>& A, const std::vector<CMat>& Bm, CMat* rinv_out = nullptr) {
  const int m = Q->m;
  if (!fast_rmul(c, m)) {
    BCG_TRY(trisolve(c, Q, rho));
    for (int s = 0; s < n; ++s) {
      BCG_TRY(rmul(c, X[s], P[s], A[s], 0.0, bcg::RMUL_ADD, "block_axpy"));
      BCG_TRY(rmul(c, P[s], Q, Bm[s], 1.0, bcg::RMUL_XPAY, "block_xpay"));
    }
    return BCG_OK;
  }
  const CMat Rinv = bcg::upper_triangular_inverse(rho);
  if (rinv_out) *rinv_out = Rinv;
  for (int s0 = 0, first = 1, per = 0; first || s0 < n; s0 += per, first = 0) {
    per = bcg::phaseC_max_shifts(m, first != 0);
    const int ns = std::min(per, n - s0);
    std::vector<const CMat*> mats;
    mats.push_back(&Rinv);
    double2* Xp[8];
    double2* Pp[8];
    for (int k = 0; k < ns; ++k) {
      mats.push_back(&A[s0 + k]);
      mats.push_back(&Bm[s0 + k]);
      Xp[k] = X[s0 + k]->d;
      Pp[k] = P[s0 + k]->d;
    }
    const double2* Md;
    BCG_TRY(upload_mats(c, m, mats.data(), static_cast<int>(mats.size()), &Md));
    {
      // the launch that applies rho^-1 reads and writes Q; a later launch of the same iteration (m = 32) re-reads it
      ProfScope ps(c, "phaseC", row_bytes(Q, (first && !rinv_out ? 2 : 1) + 4 * ns),
                   product_flops(Q, (rinv_out || first ? 1 : 0) + 2 * ns));
      bcg::launch_phaseC(c->stream, m, rows_of(Q), Q->d, Xp, Pp, ns, Md, rinv_out ? 2 : first, c->row_blocks_C);
    }
    BCG_TRY(check_launch(c, "phaseC"));
  }
  return BCG_OK;
}

// Several iterations of the shifted systems in one pass (SBCGrQ below; kernels_mfma.hip: k_phaseC_multi).  The reference
// updates X_s and P_s of every active shift in every iteration (:175, :177), but only P_0 is read by the rest of the
// iteration (:135).  So an iteration that is certain to be followed by another one updates shift 0 only and keeps its
// un-normalised residual block: phase B of the next iteration writes the new block into another buffer, and the phase C
// that ends the group (the `depth`-th iteration, or the last one before the loop can stop) applies every deferred
// iteration's updates and its own to the shifts >= 1 with X_s, P_s read and written once.  Same kernel arithmetic on the
// same values in the same order: the fields the caller sees after any number of iterations are bit-identical.  Per group
// of D iterations phase C moves (5 (D-1) + D + 4 S) s instead of D (1 + 4 S) s.
// Memory: D - 2 further fields.  The phase B of the iteration that closes a full group writes the new residual block over
// T, which it reads tile by tile just before (T is dead from there to the next operator application), and one of the
// residual buffers the group releases becomes the next T.  So D = 2 costs no memory at all and is what capacity mode
// runs.  BCG_PAIR_SHIFTS=<depth> (0 or 1: off; default 4, the largest instantiated).  Measured at 64^4,
// m = 16, 4 shifts: 67.0 ms per iteration without, 55.5-56.1 at depth 2, 54.3 at 3, 53.4-53.7 at 4 (profiles/r03_group_depth.txt).
int pair_shifts_depth(const bcg_context* c, int m, int n_shifts) {
  if (c->pair_shifts < 2 || n_shifts < 2 || !fast_rows(c, m) || !fast_rmul(c, m)) return 1;
  if (!lazy_q_width(c, m) && m != 32) return 1;  // m = 8, 16 group the un-normalised blocks; m = 32 the stored ones, in pairs
  int d = std::min(c->pair_shifts, capacity_path(c, m) ? 2 : 4);
  while (d >= 2 && !bcg::phaseC_multi_fits(m, d, n_shifts)) --d;
  return d;
}

// An iteration whose updates of the shifts >= 1 wait for a later phase C
struct DeferredIteration {
  bcg_field* Q = nullptr;        // its un-normalised residual block
  CMat rinv;                     // its rho^-1
  int n_active = 0;              // shifts 1 .. n_active-1 were to be updated (:161)
  std::vector<CMat> A, B;        // their coefficients, by shift
};

// The deferred iterations' updates and the current one's (coefficients A0/B0 for shift 0, Anew/Bnew by shift for the rest).
// rinv_out != nullptr (deferred normalisation, m = 8, 16): the blocks are un-normalised and one launch does everything.
// nullptr (m = 32): the blocks are stored normalised -- the current one by the ordinary phase C launch that also updates
// shift 0.  Either way a launch takes as many shifts as have room for their matrices in LDS (each launch reads the
// residual blocks again, and normalises them again if they are stored un-normalised).
// flush_rinv != nullptr: the "current" iteration is itself a deferred one whose shift 0 has been updated already (error
// paths, sbcgrq_flush_pending): only the shifts >= 1 are touched, *flush_rinv is its rho^-1 and rho_new, A0, B0 are unused.
int phase_C_multi(bcg_context* c, const std::vector<DeferredIteration>& pend, bcg_field* Qnew, const CMat& rho_new,
                  bcg_field* const* X, bcg_field* const* P, const CMat& A0, const CMat& B0, int n_active_new,
                  const std::vector<CMat>& Anew, const std::vector<CMat>& Bnew, CMat* rinv_out, bool lazy,
                  const CMat* flush_rinv = nullptr) {
  const int m = Qnew->m, ns = static_cast<int>(pend.size()) + 1;
  const CMat rinv_new = flush_rinv ? *flush_rinv : (lazy ? bcg::upper_triangular_inverse(rho_new) : CMat());
  const double2* Qd[4];
  for (int j = 0; j + 1 < ns; ++j) Qd[j] = pend[j].Q->d;
  Qd[ns - 1] = Qnew->d;
  struct Entry {
    int shift, first, last;
    std::vector<const CMat*> mats;
  };
  std::vector<Entry> entries;
  if (flush_rinv) {
    // nothing for shift 0
  } else if (lazy) {
    entries.push_back(Entry{0, ns - 1, ns, {&A0, &B0}});
  } else {
    const std::vector<CMat> a0(1, A0), b0(1, B0);
    BCG_TRY(phase_C(c, Qnew, rho_new, X, P, 1, a0, b0, nullptr));  // Q <- Q rho^-1 stored; shift 0
  }
  const int n_first = ns > 1 ? pend[0].n_active : n_active_new;
  for (int s = 1; s < n_first; ++s) {  // the active set only shrinks: a shift takes a prefix of the steps
    Entry e{s, 0, 0, {}};
    for (int j = 0; j < ns; ++j) {
      const bool on = s < (j + 1 < ns ? pend[j].n_active : n_active_new);
      if (!on) break;
      e.mats.push_back(j + 1 < ns ? &pend[j].A[s] : &Anew[s]);
      e.mats.push_back(j + 1 < ns ? &pend[j].B[s] : &Bnew[s]);
      ++e.last;
    }
    entries.push_back(e);
  }
  const int per_launch = bcg::phaseC_multi_max_entries(m, ns, lazy);  // m = 16: all of 4 shifts at any depth, 8 at depth 2
  static const char* const names[5] = {"", "", "phaseC_multi2", "phaseC_multi3", "phaseC_multi4"};
  for (size_t e0 = 0; e0 < entries.size(); e0 += per_launch) {
    const int n = static_cast<int>(std::min(entries.size() - e0, static_cast<size_t>(per_launch)));
    std::vector<const CMat*> mats;
    if (lazy) {
      for (int j = 0; j + 1 < ns; ++j) mats.push_back(&pend[j].rinv);
      mats.push_back(&rinv_new);
    }
    double2* Xp[8];
    double2* Pp[8];
    int first[8], last[8];
    double products = lazy ? ns : 0;
    for (int k = 0; k < n; ++k) {
      const Entry& e = entries[e0 + k];
      Xp[k] = X[e.shift]->d;
      Pp[k] = P[e.shift]->d;
      first[k] = e.first;
      last[k] = e.last;
      mats.insert(mats.end(), e.mats.begin(), e.mats.end());
      products += static_cast<double>(e.mats.size());
    }
    const double2* Md;
    BCG_TRY(upload_mats(c, m, mats.data(), static_cast<int>(mats.size()), &Md));
    {
      // one profile entry per group size: each is its own kernel instantiation (k_phaseC_multi<m, waves, ns>)
      ProfScope ps(c, names[ns], row_bytes(Qnew, ns + 4 * n), product_flops(Qnew, products));
      bcg::launch_phaseC_multi(c->stream, m, rows_of(Qnew), ns, Qd, Xp, Pp, n, first, last, Md, c->row_blocks_C, lazy);
    }
    BCG_TRY(check_launch(c, "phaseC_multi"));
  }
  if (rinv_out) *rinv_out = rinv_new;
  return BCG_OK;
}

// thinQR (inc/fields.hpp:140-146)
int thin_qr(bcg_context* c, bcg_field* y, CMat& R) {
  CMat G;
  BCG_TRY(gram(c, y, y, G));
  if (!G.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not finite");
  if (!bcg::cholesky_upper(G, R)) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not positive definite");
  return trisolve(c, y, R);
}

double max_ratio(const std::vector<double>& num, const std::vector<double>& den) {
  double r = 0.0;
  for (size_t i = 0; i < num.size(); ++i) r = std::max(r, num[i] / den[i]);
  return r;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* bcg_last_error(const bcg_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int bcg_context_create(bcg_context** out, int device, void* stream, int ndim, const int* global_dims, const int* grid,
                       const int* coords) {
  if (!out || !global_dims || ndim < 1 || ndim > 4) {
    g_create_error = "bcg_context_create: bad arguments";
    return BCG_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_create_error = "bcg_context_create: no usable HIP device (this library has no CPU fallback)";
    return BCG_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    g_create_error = "bcg_context_create: hipGetDeviceProperties failed";
    return BCG_ERR_NO_DEVICE;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("bcg_context_create: device is not gfx950 (MI355X): ") + prop.gcnArchName;
    return BCG_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_create_error = "bcg_context_create: hipSetDevice failed";
    return BCG_ERR_HIP;
  }
  bcg_context* c = new bcg_context();
  c->device = device;
  c->ndim = ndim;
  int64_t V = 1;
  int64_t ghost = 0;
  for (int mu = 0; mu < 4; ++mu) {
    c->gdims[mu] = mu < ndim ? global_dims[mu] : 1;
    c->grid[mu] = (mu < ndim && grid) ? grid[mu] : 1;
    c->coords[mu] = (mu < ndim && coords) ? coords[mu] : 0;
    if (c->gdims[mu] < 1 || c->grid[mu] < 1 || c->gdims[mu] % c->grid[mu] != 0 || c->coords[mu] < 0 ||
        c->coords[mu] >= c->grid[mu]) {
      g_create_error = "bcg_context_create: lattice extents must be positive multiples of the process grid";
      delete c;
      return BCG_ERR_INVALID;
    }
    c->lat.L[mu] = c->gdims[mu] / c->grid[mu];
    c->lat.origin[mu] = c->coords[mu] * c->lat.L[mu];
    c->lat.split[mu] = c->grid[mu] > 1 ? 1 : 0;
    c->lat.stride[mu] = V;
    V *= c->lat.L[mu];
    if (c->lat.split[mu]) c->distributed = true;
  }
  c->lat.ndim = ndim;
  c->lat.V = V;
  for (int mu = 0; mu < 4; ++mu) {
    c->lat.face_sites[mu] = V / c->lat.L[mu];
    c->lat.ghost_off[mu][0] = c->lat.ghost_off[mu][1] = 0;
    if (c->lat.split[mu]) {
      c->lat.ghost_off[mu][0] = ghost;
      c->lat.ghost_off[mu][1] = ghost + c->lat.face_sites[mu];
      ghost += 2 * c->lat.face_sites[mu];
    }
  }
  c->ghost_sites = ghost;
  // tuning overrides for experiments (tools/hop_sweep.py); defaults in kernels_mfma.hpp
  if (const char* e = std::getenv("BCG_ROW_BLOCKS_B")) c->row_blocks_B = std::atoi(e);
  if (const char* e = std::getenv("BCG_ROW_BLOCKS_C")) c->row_blocks_C = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_WALK")) c->hop_tune.patch_walk = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HOP_BLOCKS")) c->hop_tune.blocks = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_BLOCKS_OVERLAP")) c->hop_tune.blocks_overlap = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_SYNC")) c->hop_tune.sync.window = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_SYNC_LIMIT")) c->hop_tune.sync.limit_ticks = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_COLUMN")) c->hop_tune.sync.column_walk = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HOP_BUNDLE")) c->hop_tune.sync.bundle_walk = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_BUNDLE_SYNC")) c->hop_tune.sync.bundle_window = std::atoi(e);
  if (const char* e = std::getenv("BCG_LAZY_Q")) c->lazy_q = std::atoi(e);  // 2: at m = 32 too (tuning)
  if (const char* e = std::getenv("BCG_PAIR_SHIFTS")) c->pair_shifts = std::atoi(e);  // depth (pair_shifts_depth)
  if (const char* e = std::getenv("BCG_FIELD_STAGGER")) c->field_stagger = static_cast<size_t>(std::atol(e)) & ~static_cast<size_t>(255);
  if (const char* e = std::getenv("BCG_RING_CHUNK")) c->ring_chunk_override = std::atoi(e);
  if (const char* e = std::getenv("BCG_DEBUG_FIELD_BUDGET")) c->debug_field_budget = static_cast<size_t>(std::atoll(e));
  if (const char* e = std::getenv("BCG_DEBUG_FAIL_ITER")) c->debug_fail_iter = std::atoi(e);
  if (const char* e = std::getenv("BCG_RING_OVERLAP")) c->ring_overlap = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HALF_CHUNK_FORCE")) c->half_chunk_force = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HALF_CHUNK")) c->half_chunk_override = std::atoi(e);  // x3 chunk of the half-volume sweep (tests, tuning)
  if (const char* e = std::getenv("BCG_FORCE_TILE_CLASSES")) c->force_tile_classes = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HOP_FLAGS")) c->hop_tune.nontemporal = (std::atoi(e) & 1) != 0;
  if (const char* e = std::getenv("BCG_HOP_PATCH")) std::sscanf(e, "%d,%d,%d", &c->hop_tune.patch[0], &c->hop_tune.patch[1], &c->hop_tune.patch[2]);
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      g_create_error = "bcg_context_create: hipStreamCreate failed";
      delete c;
      return BCG_ERR_HIP;
    }
    c->own_stream = true;
  }
  *out = c;
  return BCG_OK;
}

int bcg_context_destroy(bcg_context* c) {
  DeviceScope on_device(c);
  if (!c) return BCG_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& kv : c->tmp_field) {
    (void)hipFree(kv.second->base);
    delete kv.second;
  }
  for (auto& kv : c->tmp_ring_buf)
    if (kv.second) (void)hipFree(kv.second);
  for (auto& kv : c->boundary_tiles)
    if (kv.second.first) (void)hipFree(kv.second.first);
  if (c->halo_send) (void)hipFree(c->halo_send);
  if (c->halo_recv) (void)hipFree(c->halo_recv);
  if (c->halo_save) (void)hipFree(c->halo_save);
  if (c->partials) (void)hipFree(c->partials);
  if (c->hop_tune.sync.counters) (void)hipFree(c->hop_tune.sync.counters);
  if (c->dev_mats) (void)hipFree(c->dev_mats);
  if (c->pin_mats) (void)hipHostFree(c->pin_mats);
  if (c->dev_gram) (void)hipFree(c->dev_gram);
  if (c->fold_tickets) (void)hipFree(c->fold_tickets);
  if (c->pin_gram) (void)hipHostFree(c->pin_gram);
  if (c->staging) (void)hipFree(c->staging);
  for (int k = 0; k < 2; ++k) {  // the upload / download pipeline (ensure_xfer)
    if (c->xfer_stream[k]) (void)hipStreamSynchronize(c->xfer_stream[k]);
    if (c->xfer_dev[k]) (void)hipFree(c->xfer_dev[k]);
    if (c->xfer_pin[k]) (void)hipHostFree(c->xfer_pin[k]);
    if (c->xfer_done[k]) (void)hipEventDestroy(c->xfer_done[k]);
    if (c->xfer_stream[k]) (void)hipStreamDestroy(c->xfer_stream[k]);
    c->xfer_dev[k] = nullptr;
    c->xfer_pin[k] = nullptr;
    c->xfer_done[k] = nullptr;
    c->xfer_stream[k] = nullptr;
  }
  for (auto& kv : c->prof)
    for (auto& pr : kv.second.pending) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  for (hipEvent_t ev : c->event_pool) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return BCG_OK;
}

int bcg_context_set_comm(bcg_context* c, const bcg_comm* comm) {
  if (!c) return BCG_ERR_INVALID;
  if (comm) {
    c->comm = *comm;
    c->have_comm = true;
  } else {
    c->have_comm = false;
  }
  return BCG_OK;
}

int64_t bcg_local_volume(const bcg_context* c) { return c ? c->lat.V : -1; }

int bcg_local_dims(const bcg_context* c, int* dims4, int* origin4) {
  if (!c) return BCG_ERR_INVALID;
  for (int mu = 0; mu < 4; ++mu) {
    if (dims4) dims4[mu] = c->lat.L[mu];
    if (origin4) origin4[mu] = c->lat.origin[mu];
  }
  return BCG_OK;
}

int bcg_halo_plan(int ndim, const int* global_dims, const int* grid, const int* coords, size_t site_bytes, int* peer_send,
                  int* peer_recv, size_t* send_offset, size_t* recv_offset, size_t* nbytes, int64_t* ghost_sites) {
  if (ndim < 1 || ndim > 4 || !global_dims || !peer_send || !peer_recv || !send_offset || !recv_offset || !nbytes) return -1;
  return halo_plan(ndim, global_dims, grid, coords, site_bytes, peer_send, peer_recv, send_offset, recv_offset, nbytes,
                   ghost_sites);
}

int bcg_halo_buffers(bcg_context* c, void** send, void** recv, size_t* bytes_each) {
  DeviceScope on_device(c);
  if (!c) return BCG_ERR_INVALID;
  if (send) *send = c->halo_send;
  if (recv) *recv = c->halo_recv;
  if (bytes_each) *bytes_each = c->halo_bytes;
  return BCG_OK;
}

// Tuning aid, not part of the interface (no declaration in include/): copy the Gram scratch buffer to the host.
// Builds with -DBCG_HOP4_TRACE leave per-tile time stamps of the plain stencil there (tools/hop_drift.py).
int bcg_debug_read_scratch(bcg_context* c, void* host, size_t bytes) {
  DeviceScope on_device(c);
  if (!c || !host || !c->partials || bytes > c->partials_bytes) return BCG_ERR_INVALID;
  BCG_TRY(stream_sync(c));
  HIP_TRY(c, hipMemcpy(host, c->partials, bytes, hipMemcpyDeviceToHost));
  return BCG_OK;
}

int bcg_context_stream(const bcg_context* c, void** stream_out, int* device_out) {
  if (!c) return BCG_ERR_INVALID;
  if (stream_out) *stream_out = c->stream;
  if (device_out) *device_out = c->device;
  return BCG_OK;
}

int bcg_overlap_tuning(bcg_context* c, int interior_blocks) {
  if (!c || interior_blocks < 0 || interior_blocks > kMaxGramBlocks / 2) return BCG_ERR_INVALID;
  if (interior_blocks > 0) c->hop_tune.blocks_overlap = interior_blocks;
  return BCG_OK;
}

int bcg_synchronize(bcg_context* c) {
  DeviceScope on_device(c);
  if (!c) return BCG_ERR_INVALID;
  return stream_sync(c);
}

int bcg_profiling(bcg_context* c, int enable) {
  if (!c) return BCG_ERR_INVALID;
  BCG_TRY(stream_sync(c));
  c->profiling = enable != 0;
  return BCG_OK;
}

int bcg_profile_reset(bcg_context* c) {
  DeviceScope on_device(c);
  if (!c) return BCG_ERR_INVALID;
  BCG_TRY(stream_sync(c));
  for (auto& kv : c->prof) {
    kv.second.ms = 0;
    kv.second.count = 0;
    kv.second.bytes = 0;
    kv.second.flops = 0;
  }
  return BCG_OK;
}

const char* bcg_profile_json(bcg_context* c) {
  DeviceScope on_device(c);
  if (!c) return "{}";
  (void)stream_sync(c);
  std::ostringstream os;
  os.precision(15);
  os << "{";
  bool first = true;
  for (auto& kv : c->prof) {
    if (kv.second.count == 0) continue;
    if (!first) os << ", ";
    first = false;
    os << "\"" << kv.first << "\": {\"ms\": " << kv.second.ms << ", \"count\": " << kv.second.count
       << ", \"bytes\": " << kv.second.bytes << ", \"flops\": " << kv.second.flops << "}";
  }
  os << "}";
  c->prof_json = os.str();
  return c->prof_json.c_str();
}

int bcg_force_generic(bcg_context* c, int enable) {
  if (!c) return BCG_ERR_INVALID;
  c->force_generic = enable != 0;
  return BCG_OK;
}

int bcg_capacity_mode(bcg_context* c, int ring_slices) {
  DeviceScope on_device(c);
  if (!c) return BCG_ERR_INVALID;
  if (ring_slices != 0) {
    if (c->lat.ndim != 4 || c->lat.split[3])
      BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "capacity mode needs a 4-D lattice whose last direction is not divided over ranks");
    if (ring_slices < 3 || ring_slices > c->lat.L[3] || c->lat.L[3] % ring_slices != 0)
      BCG_FAIL(c, BCG_ERR_INVALID, "capacity mode: ring_slices must be >= 3 and divide the local extent of direction 3");
  }
  if (ring_slices != c->tmp_ring) {  // drop scratch of the other mode
    BCG_TRY(stream_sync(c));
    for (auto& kv : c->tmp_ring_buf)
      if (kv.second) (void)hipFree(kv.second);
    c->tmp_ring_buf.clear();
    if (ring_slices != 0) {
      for (auto& kv : c->tmp_field) {
        (void)hipFree(kv.second->base);
        c->field_bytes_live -= field_bytes(kv.second);
        delete kv.second;
      }
      c->tmp_field.clear();
    }
  }
  c->tmp_ring = ring_slices;
  return BCG_OK;
}

// Pure host arithmetic (no context, no device): what one rank of `grid` allocates for an SBCGrQ solve of width m.
int bcg_sbcgrq_plan_bytes(int ndim, const int* global_dims, const int* grid, int m, int n_shifts, int consume_B, int ring_slices,
                          int ring_overlapped, int group_depth, size_t* bytes_out) {
  if (!global_dims || !bytes_out || ndim < 1 || ndim > 4 || n_shifts < 1 || !bcg::width_supported(m)) return BCG_ERR_INVALID;
  int64_t V = 1, ghost = 0;
  int L[4] = {1, 1, 1, 1};
  for (int mu = 0; mu < ndim; ++mu) {
    const int g = grid ? grid[mu] : 1;
    if (g < 1 || global_dims[mu] < 1 || global_dims[mu] % g != 0) return BCG_ERR_INVALID;
    L[mu] = global_dims[mu] / g;
    V *= L[mu];
  }
  for (int mu = 0; mu < ndim; ++mu)
    if (grid && grid[mu] > 1) ghost += 2 * (V / L[mu]);
  if (ring_slices != 0 && (ndim != 4 || ring_slices < 3 || L[3] % ring_slices != 0)) return BCG_ERR_INVALID;
  const size_t field = static_cast<size_t>(V) * 3 * m * sizeof(double2);
  size_t total = field * (2 * static_cast<size_t>(n_shifts) + 2 + (consume_B ? 0 : 1));  // X_s, P_s, Q, T (+ B)
  total += field * std::max(0, group_depth - 2);                                          // further residual buffers
  total += ring_slices > 0 ? field / L[3] * ring_slices : field;                          // tmp of dirac_op::op
  total += static_cast<size_t>(V) * ndim * 9 * sizeof(double2);                          // links
  total += static_cast<size_t>(ghost) * (2 * 3 * m + 9) * sizeof(double2);                // send + receive faces, ghost links
  if (ring_slices > 0 && (ring_slices - 2) / 2 < 1) ring_overlapped = 0;
  if (ring_slices > 0 && ring_overlapped) total += static_cast<size_t>(ghost) / L[3] * 3 * m * sizeof(double2);  // saved slice-0 faces
  size_t partials = static_cast<size_t>(kMaxGramBlocks) * 32 * 32 * sizeof(double2);
  // (a ring of 3 slices cannot hold two chunks: the library then runs the serial form whatever the callbacks offer --
  //  ring_overlapped(c) -- and so does this plan; the partials term assumes the default stencil grid and no BCG_RING_CHUNK)
  if (ring_slices > 0 && (ring_slices - 2) / 2 < 1) ring_overlapped = 0;
  if (ring_slices > 0 && m == 16) {  // capacity mode: the block partials of all chunks side by side (ensure_ring_scratch)
    const int C = ring_overlapped ? (ring_slices - 2) / 2 : ring_slices - 2;
    const int chunks = (L[3] + C - 1) / C;
    partials = std::max(partials, static_cast<size_t>(kFastBlocks) * chunks * m * m * sizeof(double2));
  }
  total += partials + kMatSlotBytes * (kMatSlots + 1);
  *bytes_out = total;
  return BCG_OK;
}

int bcg_sbcgrq_device_bytes(const bcg_context* c, int m, int n_shifts, int consume_B, size_t* bytes_out) {
  DeviceScope on_device(c);
  if (!c || !bytes_out || n_shifts < 1 || !bcg::width_supported(m)) return BCG_ERR_INVALID;
  const bool cap = capacity_path(c, m);
  return bcg_sbcgrq_plan_bytes(c->ndim, c->gdims, c->grid, m, n_shifts, consume_B, cap ? c->tmp_ring : 0,
                               cap && c->distributed && ring_overlapped(c) ? 1 : 0, pair_shifts_depth(c, m, n_shifts), bytes_out);
}

// The same plan for ONE half-volume solve (bcg_field_create_half: every work field holds V/2 sites; links stay full)
int bcg_sbcgrq_device_bytes_half(const bcg_context* c, int m, int n_shifts, int consume_B, size_t* bytes_out) {
  DeviceScope on_device(c);
  if (!c || !bytes_out || n_shifts < 1 || !bcg::width_supported(m)) return BCG_ERR_INVALID;
  const size_t half = static_cast<size_t>(c->lat.V / 2) * 3 * m * sizeof(double2);
  size_t total = half * (2 * static_cast<size_t>(n_shifts) + 2 + (consume_B ? 0 : 1) + 1);  // X_s, P_s, Q, T (+ B), tmp
  total += half * std::max(0, pair_shifts_depth(c, m, n_shifts) - 2);                       // further residual buffers
  total += static_cast<size_t>(c->lat.V) * c->ndim * 9 * sizeof(double2);                  // links
  if (c->distributed)  // send + receive faces (allocated at the full-field size: the same buffers serve full fields), ghost links
    total += static_cast<size_t>(c->ghost_sites) * (2 * 3 * m + 9) * sizeof(double2);
  total += static_cast<size_t>(kMaxGramBlocks) * 32 * 32 * sizeof(double2) + kMatSlotBytes * (kMatSlots + 1);
  *bytes_out = total;
  return BCG_OK;
}

// ---- fields ------------------------------------------------------------------------------------
namespace {
// parity -1: all local sites; 0 / 1: the parity-compact half (kernels_generic.hip, "Half-volume fields")
int create_field(bcg_context* c, int m, int parity, bcg_field** out) {
  if (!bcg::width_supported(m)) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "block width out of range (supported: 1 <= m <= 32)");
  if (parity >= 0) {
    if (c->distributed && c->ndim < 2) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "half-volume fields on a lattice divided over ranks: two dimensions or more");
    for (int mu = 0; mu < c->ndim; ++mu)
      if (c->lat.L[mu] % 2 != 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "half-volume fields: every lattice extent must be even");
  }
  bcg_field* f = new bcg_field{c, m, nullptr, nullptr, parity, parity >= 0 ? c->lat.V / 2 : c->lat.V};
  // Fields of the lattices that matter have power-of-two sizes (64^4 sites x 768 B = 12 GiB), allocated back to back, so the
  // streaming kernels read the same offset of up to nine of them at once with identical low address bits.  A per-field
  // stagger (a multiple of 256 B, so alignment is kept) spreads those accesses over the memory channels.
  const size_t lead = c->field_stagger * static_cast<size_t>(c->fields_created % 16);
  hipError_t e = (c->debug_field_budget && c->field_bytes_live + field_bytes(f) > c->debug_field_budget)
                     ? hipErrorOutOfMemory  // test aid: a deterministic stand-in for a full device (tests/test_robustness.py)
                     : hipMalloc(&f->base, field_bytes(f) + c->field_stagger * 16);
  if (e != hipSuccess) {
    delete f;
    c->err = std::string("bcg_field_create: hipMalloc: ") + hipGetErrorString(e);
    return BCG_ERR_HIP;
  }
  f->d = reinterpret_cast<double2*>(static_cast<char*>(f->base) + lead);
  c->fields_created += 1;
  c->field_bytes_live += field_bytes(f);
  *out = f;
  return BCG_OK;
}
}  // namespace

int bcg_field_create(bcg_context* c, int m, bcg_field** out) {
  DeviceScope on_device(c);
  if (!c || !out) return BCG_ERR_INVALID;
  return create_field(c, m, -1, out);
}
int bcg_field_create_half(bcg_context* c, int m, int parity, bcg_field** out) {
  DeviceScope on_device(c);
  if (!c || !out || (parity != 0 && parity != 1)) return BCG_ERR_INVALID;
  return create_field(c, m, parity, out);
}
int bcg_field_parity(const bcg_field* f) { return f ? f->parity : -2; }
int64_t bcg_field_sites(const bcg_field* f) { return f ? f->sites : -1; }
// half <- the sites of its parity of full, or the reverse
int bcg_field_parity_copy(bcg_field* full, bcg_field* half, int to_half) {
  DeviceScope on_device(full ? full->ctx : nullptr);
  if (!full || !half || full->ctx != half->ctx || full->m != half->m || full->parity != -1 || half->parity < 0) return BCG_ERR_INVALID;
  bcg_context* c = full->ctx;
  bcg::launch_parity_copy(c->stream, full->m, c->lat, half->parity, full->d, half->d, to_half != 0);
  return check_launch(c, "parity_copy");
}

int bcg_field_destroy(bcg_field* f) {
  DeviceScope on_device(f ? f->ctx : nullptr);
  if (!f) return BCG_OK;
  (void)hipStreamSynchronize(f->ctx->stream);
  (void)hipFree(f->base);
  f->ctx->field_bytes_live -= field_bytes(f);
  delete f;
  return BCG_OK;
}

int bcg_field_width(const bcg_field* f) { return f ? f->m : -1; }

int bcg_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return BCG_ERR_INVALID;
  *out = nullptr;
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? BCG_OK : BCG_ERR_HIP;
}
int bcg_host_free(void* p) { return !p || hipHostFree(p) == hipSuccess ? BCG_OK : BCG_ERR_HIP; }

int bcg_field_upload(bcg_field* f, const double* host) {
  DeviceScope on_device(f ? f->ctx : nullptr);
  if (!f || !host) return BCG_ERR_INVALID;
  return transfer_field(f->ctx, f, const_cast<double*>(host), /*to_device=*/true);
}

int bcg_field_download(const bcg_field* f, double* host) {
  DeviceScope on_device(f ? f->ctx : nullptr);
  if (!f || !host) return BCG_ERR_INVALID;
  return transfer_field(f->ctx, const_cast<bcg_field*>(f), host, /*to_device=*/false);
}

int bcg_field_download_sites(const bcg_field* f, int64_t n, const int64_t* sites, double* host) {
  DeviceScope on_device(f ? f->ctx : nullptr);
  if (!f || n < 0 || (n > 0 && (!sites || !host))) return BCG_ERR_INVALID;
  bcg_context* c = f->ctx;
  for (int64_t k = 0; k < n; ++k)
    if (sites[k] < 0 || sites[k] >= f->sites) BCG_FAIL(c, BCG_ERR_INVALID, "bcg_field_download_sites: site out of range");
  const size_t site_bytes = static_cast<size_t>(3) * f->m * sizeof(double2);
  const int64_t chunk = 4096;
  BCG_TRY(ensure_staging(c, static_cast<size_t>(chunk) * site_bytes));
  for (int64_t k0 = 0; k0 < n; k0 += chunk) {
    const int64_t nk = std::min<int64_t>(chunk, n - k0);
    for (int64_t k = 0; k < nk; ++k)  // one tile each: the layout conversion of bcg_field_download on a single site
      bcg::launch_dev_to_host(c->stream, f->m, f->d + sites[k0 + k] * 3 * f->m, c->staging + k * 3 * f->m, 1);
    BCG_TRY(check_launch(c, "dev_to_host"));
    HIP_TRY(c, hipMemcpyAsync(reinterpret_cast<char*>(host) + k0 * site_bytes, c->staging, nk * site_bytes,
                              hipMemcpyDeviceToHost, c->stream));
    BCG_TRY(stream_sync(c));
  }
  return BCG_OK;
}

int bcg_field_copy(bcg_field* dst, const bcg_field* src) {
  DeviceScope on_device(dst ? dst->ctx : nullptr);
  if (!same_shape(dst, src)) return BCG_ERR_INVALID;
  bcg_context* c = dst->ctx;
  ProfScope ps(c, "copy");
  HIP_TRY(c, hipMemcpyAsync(dst->d, src->d, field_bytes(dst), hipMemcpyDeviceToDevice, c->stream));
  return BCG_OK;
}

int bcg_field_set_zero(bcg_field* f) {
  DeviceScope on_device(f ? f->ctx : nullptr);
  if (!f) return BCG_ERR_INVALID;
  bcg_context* c = f->ctx;
  ProfScope ps(c, "set_zero");
  HIP_TRY(c, hipMemsetAsync(f->d, 0, field_bytes(f), c->stream));
  return BCG_OK;
}

int bcg_field_fill_random(bcg_field* f, uint64_t seed) {
  DeviceScope on_device(f ? f->ctx : nullptr);
  if (!f) return BCG_ERR_INVALID;
  bcg_context* c = f->ctx;
  if (f->parity >= 0) bcg::launch_fill_field_half(c->stream, f->m, c->lat, c->gdims, f->parity, f->d, seed);
  else bcg::launch_fill_field(c->stream, f->m, c->lat, c->gdims, f->d, seed);
  return check_launch(c, "fill_field");
}

int bcg_field_add_assign(bcg_field* y, const bcg_field* x) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!same_shape(y, x)) return BCG_ERR_INVALID;
  return axpby(y->ctx, y, 1.0, x, 1.0, "axpby");
}
int bcg_field_sub_assign(bcg_field* y, const bcg_field* x) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!same_shape(y, x)) return BCG_ERR_INVALID;
  return axpby(y->ctx, y, 1.0, x, -1.0, "axpby");
}
int bcg_field_add_scalar(bcg_field* y, const bcg_field* x, double a) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!same_shape(y, x)) return BCG_ERR_INVALID;
  return axpby(y->ctx, y, 1.0, x, a, "axpby");
}
int bcg_field_rescale_add_scalar(bcg_field* y, double a, const bcg_field* x, double b) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!same_shape(y, x)) return BCG_ERR_INVALID;
  return axpby(y->ctx, y, a, x, b, "axpby");
}
int bcg_field_add_matrix(bcg_field* y, const bcg_field* x, const double* M) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!same_shape(y, x) || !M || y == x) return BCG_ERR_INVALID;
  return rmul(y->ctx, y, x, CMat(y->m, M), 0.0, bcg::RMUL_ADD, "block_axpy");
}
int bcg_field_rescale_add_matrix(bcg_field* y, const double* M, const bcg_field* x, double b) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!same_shape(y, x) || !M) return BCG_ERR_INVALID;
  return rmul(y->ctx, y, x, CMat(y->m, M), b, bcg::RMUL_XPAY, "block_xpay");
}

int bcg_field_hermitian_dot(const bcg_field* a, const bcg_field* b, double* out) {
  DeviceScope on_device(a ? a->ctx : nullptr);
  if (!same_shape(a, b) || !out) return BCG_ERR_INVALID;
  CMat G;
  BCG_TRY(gram(a->ctx, a, b, G));
  G.store(out);
  return BCG_OK;
}

int bcg_field_real_dot(const bcg_field* a, const bcg_field* b, double* out) {
  DeviceScope on_device(a ? a->ctx : nullptr);
  if (!same_shape(a, b) || !out) return BCG_ERR_INVALID;
  if (a->m != 1) BCG_FAIL(a->ctx, BCG_ERR_INVALID, "real_dot is defined for N_rhs = 1 (inc/fields.hpp:93)");
  CMat G;
  BCG_TRY(gram(a->ctx, a, b, G, false));
  *out = G(0, 0).real();
  return BCG_OK;
}

int bcg_field_tri_solve_rhs(bcg_field* y, const double* R) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!y || !R) return BCG_ERR_INVALID;
  return trisolve(y->ctx, y, CMat(y->m, R));
}

int bcg_field_thin_qr(bcg_field* y, double* R_out) {
  DeviceScope on_device(y ? y->ctx : nullptr);
  if (!y || !R_out) return BCG_ERR_INVALID;
  CMat R;
  BCG_TRY(thin_qr(y->ctx, y, R));
  R.store(R_out);
  return BCG_OK;
}

// ---- operator ----------------------------------------------------------------------------------
int bcg_gauge_create(bcg_context* c, bcg_gauge** out) {
  DeviceScope on_device(c);
  if (!c || !out) return BCG_ERR_INVALID;
  bcg_gauge* g = new bcg_gauge{c, nullptr, nullptr, false};
  const size_t u_bytes = static_cast<size_t>(c->lat.V) * c->ndim * 9 * sizeof(double2);
  // experiment switch: links in memory the L2 does not cache (they are streamed; the L2 is for the field slices)
  const char* unc = std::getenv("BCG_U_UNCACHED");
  hipError_t e = (unc && std::atoi(unc) != 0)
                     ? hipExtMallocWithFlags(reinterpret_cast<void**>(&g->U), u_bytes, hipDeviceMallocUncached)
                     : hipMalloc(&g->U, u_bytes);
  if (e == hipSuccess && c->ghost_sites > 0) e = hipMalloc(&g->Ughost, static_cast<size_t>(c->ghost_sites) * 9 * sizeof(double2));
  if (e != hipSuccess) {
    if (g->U) (void)hipFree(g->U);
    delete g;
    c->err = std::string("bcg_gauge_create: hipMalloc: ") + hipGetErrorString(e);
    return BCG_ERR_HIP;
  }
  *out = g;
  return BCG_OK;
}

int bcg_gauge_destroy(bcg_gauge* g) {
  DeviceScope on_device(g ? g->ctx : nullptr);
  if (!g) return BCG_OK;
  (void)hipStreamSynchronize(g->ctx->stream);
  (void)hipFree(g->U);
  if (g->Ughost) (void)hipFree(g->Ughost);
  delete g;
  return BCG_OK;
}

int bcg_gauge_upload(bcg_gauge* g, const double* host) {
  DeviceScope on_device(g ? g->ctx : nullptr);
  if (!g || !host) return BCG_ERR_INVALID;
  bcg_context* c = g->ctx;
  HIP_TRY(c, hipMemcpyAsync(g->U, host, static_cast<size_t>(c->lat.V) * c->ndim * 9 * sizeof(double2),
                            hipMemcpyHostToDevice, c->stream));
  g->ghost_valid = false;
  return stream_sync(c);
}

int bcg_gauge_fill_random(bcg_gauge* g, uint64_t seed) {
  DeviceScope on_device(g ? g->ctx : nullptr);
  if (!g) return BCG_ERR_INVALID;
  bcg_context* c = g->ctx;
  bcg::launch_fill_gauge(c->stream, c->lat, c->gdims, g->U, seed);
  g->ghost_valid = false;
  return check_launch(c, "fill_gauge");
}

int bcg_dirac_hop(bcg_context* c, const bcg_gauge* g, bcg_field* out, const bcg_field* in) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(out, in) || out == in || g->ctx != c || in->ctx != c) return BCG_ERR_INVALID;
  return hop(c, g, out, in, bcg::HOP_PLAIN, nullptr, 0.0);
}

// out (parity p) = D in (parity 1 - p): the two off-diagonal blocks of D in the parity basis
int bcg_dirac_hop_half(bcg_context* c, const bcg_gauge* g, bcg_field* out, const bcg_field* in) {
  DeviceScope on_device(c);
  if (!c || !g || !out || !in || out == in || g->ctx != c || in->ctx != c || out->ctx != c || out->m != in->m || in->parity < 0 ||
      out->parity != 1 - in->parity)
    return BCG_ERR_INVALID;
  BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
  BCG_TRY(halo_field(c, in));
  bcg::launch_hop_half(c->stream, in->m, c->lat, out->parity, g->U, g->Ughost, in->d, c->halo_recv, out->d, bcg::HOP_PLAIN, nullptr, 0.0);
  return check_launch(c, "hop_half");
}

int bcg_dirac_apply(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* out, const bcg_field* in) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(out, in) || out == in || g->ctx != c || in->ctx != c) return BCG_ERR_INVALID;
  return apply_shifted(c, g, mass, 0.0, out, in);
}

// ---- the callers either side of the hot path, on the same kernels ------------
}  // extern "C"

namespace {

// Re(a^dagger b) for N_rhs = 1 fields: real_dot (inc/fields.hpp:93-99)
int real_dot(bcg_context* c, const bcg_field* a, const bcg_field* b, double& out) {
  CMat G;
  BCG_TRY(gram(c, a, b, G, false));
  out = G(0, 0).real();
  return BCG_OK;
}

struct FieldPool {  // work fields of one solver call, released together
  bcg_context* c;
  std::vector<bcg_field*> f;
  int parity = -1;  // of the fields made without an `init` to copy: set it to the parity of the solve's source
  explicit FieldPool(bcg_context* ctx) : c(ctx) {}
  ~FieldPool() {
    for (bcg_field* p : f) bcg_field_destroy(p);
  }
  int make(int m, bcg_field** out, const bcg_field* init = nullptr) {
    bcg_field* p = nullptr;
    const int par = init ? init->parity : parity;
    if (par >= 0) BCG_TRY(bcg_field_create_half(c, m, par, &p));
    else BCG_TRY(bcg_field_create(c, m, &p));
    f.push_back(p);
    if (init) BCG_TRY(bcg_field_copy(p, init));
    *out = p;
    return BCG_OK;
  }
};

}  // namespace

extern "C" {

// True relative residuals exactly as the reference's tests and benchmark measure them
// (test/solvers.cpp:104-116, benchmark.cpp:93-103): AX = op(X_s) + sigma_s X_s - B ;
// res[s][i] = sqrt( (AX^dagger AX)_ii / (B^dagger B)_ii ).
int bcg_true_residuals(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* X, const bcg_field* B,
                       int n_shifts, const double* sigma, double* res_out) {
  DeviceScope on_device(c);
  if (!c || !g || !X || !B || !sigma || !res_out || n_shifts < 1 || g->ctx != c || B->ctx != c) return BCG_ERR_INVALID;
  const int m = B->m;
  FieldPool pool(c);
  pool.parity = B->parity;
  bcg_field* AX = nullptr;  // only the unfused path needs it
  CMat b2, r2;
  BCG_TRY(gram(c, B, B, b2));
  for (int s = 0; s < n_shifts; ++s) {
    if (!same_shape(X[s], B)) return BCG_ERR_INVALID;
    // One pass where the bundle stencil applies (m = 16, whole-field tmp): tmp = D X_s, then the second stencil
    // forms (mass^2 + sigma_s) X_s - D tmp - B in registers and accumulates its Gram product; AX is never written
    // (5 field passes + 2 link passes instead of 9 + 2).
    if (m == 16 && B->parity < 0 && fast_hop(c, m) && !capacity_path(c, m) &&
        bcg::hop_uses_bundle(m, c->lat, kFastBlocks, c->hop_tune, 0, bcg::HopWindow())) {
      bcg_field* tmp;
      BCG_TRY(get_tmp(c, m, &tmp));
      BCG_TRY(hop(c, g, tmp, X[s], bcg::HOP_PLAIN, nullptr, 0.0));
      BCG_TRY(halo_field(c, tmp));
      BCG_TRY(ensure_scratch(c));
      int nb;
      {
        ProfScope ps(c, "hop_residual", alg_bytes(c, m, 3, 1));  // reads tmp, X_s, B and the links; writes nothing
        nb = bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, tmp->d, c->halo_recv, const_cast<double2*>(B->d),
                                  bcg::HOP_RESID, X[s]->d, mass * mass + sigma[s], c->partials, true, kFastBlocks, c->hop_tune, 0);
      }
      if (nb > 0) {
        BCG_TRY(check_launch(c, "hop_residual"));
        BCG_TRY(finish_gram(c, m, nb, r2, true));
        for (int i = 0; i < m; ++i) res_out[s * m + i] = std::sqrt(r2(i, i).real() / b2(i, i).real());
        continue;
      }
    }
    if (!AX) BCG_TRY(pool.make(m, &AX));
    BCG_TRY(apply_shifted(c, g, mass, sigma[s], AX, X[s]));  // op + add(X_s, sigma_s) in one pass
    BCG_TRY(axpby(c, AX, 1.0, B, -1.0, "axpby"));
    BCG_TRY(gram(c, AX, AX, r2));
    for (int i = 0; i < m; ++i) res_out[s * m + i] = std::sqrt(r2(i, i).real() / b2(i, i).real());
  }
  return BCG_OK;
}

// CG (src/standard_solvers.cpp:3-32): single right-hand side, scalar coefficients.
int bcg_cg_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* x, const bcg_field* b, double eps,
                 int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(x, b) || x == b || g->ctx != c || b->ctx != c) return BCG_ERR_INVALID;
  if (b->m != 1) BCG_FAIL(c, BCG_ERR_INVALID, "CG takes fermion_field arguments (N_rhs = 1)");
  FieldPool pool(c);
  bcg_field *t, *p, *r;
  BCG_TRY(bcg_field_set_zero(x));  // :5
  pool.parity = b->parity;
  BCG_TRY(pool.make(1, &t));
  BCG_TRY(pool.make(1, &p, b));    // :7
  BCG_TRY(pool.make(1, &r, b));    // :8
  double rr, pt;
  BCG_TRY(real_dot(c, r, r, rr));  // :9
  int iter = 0;
  const double stop = eps * std::sqrt(rr);  // :11
  while (std::sqrt(rr) > stop && iter < max_iterations) {  // :13
    BCG_TRY(apply_shifted(c, g, mass, 0.0, t, p));          // :15
    ++iter;
    BCG_TRY(real_dot(c, p, t, pt));
    const double alpha = rr / pt;                           // :18
    BCG_TRY(axpby(c, r, 1.0, t, -alpha, "axpby"));          // :20
    const double rr_old = rr;
    BCG_TRY(real_dot(c, r, r, rr));                         // :23
    const double beta = rr / rr_old;                        // :24
    BCG_TRY(axpby(c, x, 1.0, p, alpha, "axpby"));           // :26
    BCG_TRY(axpby(c, p, beta, r, 1.0, "axpby"));            // :28
  }
  BCG_TRY(stream_sync(c));
  if (iterations_out) *iterations_out = iter;
  return BCG_OK;
}

// SCG (src/standard_solvers.cpp:34-95): multi-shift CG, scalar zeta/theta recurrences.
int bcg_scg_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* x, const bcg_field* b, int n_shifts,
                  const double* sigma, double eps, double eps_shifts, int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  if (!c || !g || !x || !b || !sigma || n_shifts < 1 || g->ctx != c || b->ctx != c) return BCG_ERR_INVALID;
  if (b->m != 1) BCG_FAIL(c, BCG_ERR_INVALID, "SCG takes fermion_field arguments (N_rhs = 1)");
  for (int s = 0; s < n_shifts; ++s)
    if (!same_shape(x[s], b) || x[s] == b) return BCG_ERR_INVALID;
  if (sigma[0] < 0.0) BCG_FAIL(c, BCG_ERR_INVALID, "SCG: shifts must be zero or positive");              // :40
  if (!std::is_sorted(sigma, sigma + n_shifts)) BCG_FAIL(c, BCG_ERR_INVALID, "SCG: shifts must be ascending");  // :41-42
  int active = n_shifts;                       // :45
  double alpha = 1.0, beta = 0.0;              // :46-47
  std::vector<double> zeta(n_shifts, 1.0), theta(n_shifts, 1.0);  // :48-49
  FieldPool pool(c);
  std::vector<bcg_field*> p(n_shifts);
  for (int s = 0; s < n_shifts; ++s) {
    BCG_TRY(bcg_field_set_zero(x[s]));         // :50-52
    BCG_TRY(pool.make(1, &p[s], b));           // :53
  }
  bcg_field *t, *r;
  pool.parity = b->parity;
  BCG_TRY(pool.make(1, &t));
  BCG_TRY(pool.make(1, &r, b));                // :54
  double rr, pt;
  BCG_TRY(real_dot(c, r, r, rr));              // :55
  int iter = 0;
  const double stop = eps * std::sqrt(rr);     // :57
  while (std::sqrt(rr) > stop && iter < max_iterations) {  // :58
    BCG_TRY(apply_shifted(c, g, mass, sigma[0], t, p[0]));  // :60-61
    ++iter;
    const double alpha_old = alpha;
    BCG_TRY(real_dot(c, p[0], t, pt));
    alpha = rr / pt;                                        // :65
    BCG_TRY(axpby(c, r, 1.0, t, -alpha, "axpby"));          // :67
    const double rr_old = rr;
    BCG_TRY(real_dot(c, r, r, rr));                         // :69
    const double beta_old = beta;
    beta = rr / rr_old;                                     // :71
    // :73-87 -- the updates of all active shifts as ONE pass over r, x_s, p_s (k_scg_update; same expressions as the
    // axpys, so the same iterates): coefficients first, then one launch
    std::vector<double> a_s(active), b_s(active), z_s(active);
    std::vector<double2*> xs(active), ps(active);
    a_s[0] = alpha; b_s[0] = beta; z_s[0] = 1.0;            // :73, :75
    for (int s = active - 1; s > 0; --s) {                  // :76
      double inv_theta = 1.0 + (sigma[s] - sigma[0]) * alpha;              // :78
      inv_theta += beta_old * (alpha / alpha_old) * (1.0 - theta[s]);      // :79
      theta[s] = 1.0 / inv_theta;                                          // :80
      zeta[s] *= theta[s];                                                 // :81
      a_s[s] = alpha * theta[s];                                           // :82  x_s += alpha_s p_s  (:85)
      b_s[s] = beta * theta[s] * theta[s];                                 // :83  p_s = beta_s p_s + zeta_s r  (:87)
      z_s[s] = zeta[s];
    }
    for (int s = 0; s < active; ++s) { xs[s] = x[s]->d; ps[s] = p[s]->d; }
    {
      ProfScope ps_(c, "scg_update");
      bcg::launch_scg_update(c->stream, r->d, active, xs.data(), ps.data(), a_s.data(), b_s.data(), z_s.data(), rows_of(r));
    }
    BCG_TRY(check_launch(c, "scg_update"));
    if (std::sqrt(rr) * zeta[active - 1] < eps_shifts) --active;           // :90-92
  }
  BCG_TRY(stream_sync(c));
  if (iterations_out) *iterations_out = iter;
  return BCG_OK;
}

// BCG (inc/block_solvers.hpp:10-45): block CG without the QR stabilisation.
int bcg_bcg_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* X, const bcg_field* B, double eps,
                  int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(X, B) || X == B || g->ctx != c || B->ctx != c) return BCG_ERR_INVALID;
  const int m = B->m;
  FieldPool pool(c);
  bcg_field *T, *P, *R;
  BCG_TRY(bcg_field_set_zero(X));   // :14
  pool.parity = B->parity;
  BCG_TRY(pool.make(m, &T));
  BCG_TRY(pool.make(m, &P, B));     // :16
  BCG_TRY(pool.make(m, &R, B));
  CMat r2, r2_old, pt;
  BCG_TRY(gram(c, R, R, r2));       // :17
  std::vector<double> norm0(m);
  for (int i = 0; i < m; ++i) norm0[i] = std::sqrt(r2(i, i).real());  // :19-20
  double residual = 1.0;
  int iter = 0;
  while (residual > eps && iter < max_iterations) {          // :25
    BCG_TRY(apply_shifted(c, g, mass, 0.0, T, P));            // :27
    ++iter;
    BCG_TRY(gram(c, P, T, pt));
    if (!pt.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "BCG: P^dagger A P is not finite");
    const CMat alpha = bcg::inverse_full_pivot(pt) * r2;      // :31  (P.T)^-1 (R.R)
    BCG_TRY(rmul(c, R, T, -alpha, 0.0, bcg::RMUL_ADD, "block_axpy"));  // :33
    r2_old = r2;
    BCG_TRY(gram(c, R, R, r2));                               // :35
    const CMat beta = bcg::inverse_full_pivot(r2_old) * r2;   // :36
    BCG_TRY(rmul(c, X, P, alpha, 0.0, bcg::RMUL_ADD, "block_axpy"));   // :38
    BCG_TRY(rmul(c, P, R, beta, 1.0, bcg::RMUL_XPAY, "block_xpay"));   // :40
    residual = 0.0;                                           // :41-42
    for (int i = 0; i < m; ++i) residual = std::max(residual, std::sqrt(r2(i, i).real()) / norm0[i]);
  }
  BCG_TRY(stream_sync(c));
  if (iterations_out) *iterations_out = iter;
  return BCG_OK;
}

// BCGrQ (inc/block_solvers.hpp:50-86) is SBCGrQ with the single shift sigma = 0: same statements in the same
// order (X += P alpha delta_old, Q -= T alpha, thinQR, P = P rho^dagger + Q, delta = rho delta).
int bcg_bcgrq_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* X, const bcg_field* B, double eps,
                    int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  const double zero = 0.0;
  bcg_field* Xs[1] = {X};
  return bcg_sbcgrq_solve(c, g, mass, Xs, const_cast<bcg_field*>(B), 1, &zero, eps, 0.0, max_iterations, 0, iterations_out,
                          nullptr, nullptr);
}

double bcg_sbcgrq_bytes_per_iteration(const bcg_context* c, int m, int n_shifts) {
  if (!c) return 0.0;
  const double s = 48.0 * m, gl = 144.0 * c->ndim;
  return static_cast<double>(c->lat.V) * ((14.0 + 4.0 * (n_shifts - 1)) * s + 2.0 * gl);
}

// ---- SBCGrQ (inc/block_solvers.hpp:91-185) -----------------------------------------------------
// The solver is a resumable state machine so that a caller can run (and time) an exact number of
// iterations: begin = everything before the while loop (:97-131), iterate = loop bodies, end =
// release of the work fields.  bcg_sbcgrq_solve is begin + iterate(max_iterations) + end.
}  // extern "C"

struct bcg_sbcgrq_state {
  bcg_context* c = nullptr;
  const bcg_gauge* g = nullptr;
  double mass = 0.0;
  int m = 0, n_shifts = 0;
  std::vector<double> sigma;
  double eps = 0.0, eps_shifts = 0.0;
  std::vector<bcg_field*> X;
  bcg_field* B = nullptr;
  bcg_field* T = nullptr;
  bcg_field* Q = nullptr;
  std::vector<bcg_field*> P;
  int n_unconverged = 0;
  CMat alpha, rho, delta, alpha_inv, alpha_inv_old, rho_old;
  std::vector<CMat> alpha_s, beta_s;
  std::vector<double> b_norm;
  double residual = 1.0;
  int iter = 0;
  CMat q_rinv;          // deferred normalisation (lazy_q_width): the stored Q times this is the reference's Q
  bool q_lazy = false;
  // pair_shifts_depth > 1: iterations whose updates of the shifts >= 1 wait for a later phase C, oldest first
  std::vector<DeferredIteration> pending;
  std::vector<bcg_field*> Qfree;         // residual buffers not in use (depth - 2 of them when nothing is pending)
  int depth = 1;
  bool failed = false;                   // an iteration returned an error: no further iterations on this state
};

namespace {

void sbcgrq_release(bcg_sbcgrq_state* st) {
  if (st->T && st->T != st->B) bcg_field_destroy(st->T);  // T, Q and the spare buffers rotate: any of them may be B's storage
  if (st->Q && st->Q != st->B) bcg_field_destroy(st->Q);
  for (bcg_field* q : st->Qfree)
    if (q != st->B) bcg_field_destroy(q);
  for (const DeferredIteration& d : st->pending)
    if (d.Q != st->B && d.Q != st->Q) bcg_field_destroy(d.Q);
  st->Qfree.clear();
  st->pending.clear();
  for (bcg_field* p : st->P)
    if (p) bcg_field_destroy(p);
  st->T = st->Q = nullptr;
  st->P.clear();
}

// One pass of the loop body, inc/block_solvers.hpp:133-182.
// more_follow: the caller will run at least one more iteration if this one leaves the residual above eps
int sbcgrq_iteration(bcg_sbcgrq_state* st, bcg_sbcgrq_trace* trace, bool more_follow) {
  bcg_context* c = st->c;
  const int m = st->m, n_shifts = st->n_shifts;
  const std::vector<double>& sigma = st->sigma;
  const CMat Identity = CMat::identity(m);
  // T = (A + sigma_0) P_0 ; alpha_inv = P_0^dagger T                          :134-140
  ++st->iter;                                            // :137
  st->alpha_inv_old = st->alpha_inv;                     // :139
  BCG_TRY(phase_A(c, st->g, st->mass, sigma[0], st->T, st->P[0], st->alpha_inv));  // global reduction #1
  if (!st->alpha_inv.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "SBCGrQ: P^dagger A P is not finite");
  st->alpha = bcg::inverse_full_pivot(st->alpha_inv);    // :142
  const CMat alpha_delta = st->alpha * st->delta;        // :145 uses delta of the previous iteration
  // Q -= T alpha ; Gram matrix of the new Q                                  :148, :152
  CMat G2;
  if (!st->pending.empty()) {  // the old block is needed by a later phase C: the new one goes to another buffer
    // ... a spare one, or, in the iteration that closes a full group, T itself: phase B reads each tile of T just before
    // it writes the same tile of the new Q, and T is not read again before the next operator application rewrites it
    const bool over_T = st->Qfree.empty();
    bcg_field* out = over_T ? st->T : st->Qfree.back();
    BCG_TRY(phase_B(c, st->Q, st->T, st->alpha, G2, st->q_lazy ? &st->q_rinv : nullptr, out));  // global reduction #2
    if (over_T) st->T = nullptr;  // one of the group's buffers takes its place below
    else st->Qfree.pop_back();
    st->Q = out;  // the old buffer stays with pending.back()
  } else {
    BCG_TRY(phase_B(c, st->Q, st->T, st->alpha, G2, st->q_lazy ? &st->q_rinv : nullptr));  // global reduction #2
  }
  st->rho_old = st->rho;                                 // :150
  if (c->debug_fail_iter > 0 && st->iter == c->debug_fail_iter) G2(0, 0) = cd(std::nan(""), 0.0);  // test aid: BCG_DEBUG_FAIL_ITER
  if (!G2.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not finite");
  if (!bcg::cholesky_upper(G2, st->rho)) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not positive definite");
  st->delta = st->rho * st->delta;                       // :153
  st->residual = max_ratio(st->delta.row_norms(), st->b_norm);  // :155

  const bool tracing = trace && trace->recorded < trace->capacity;
  double* tm = nullptr;
  double* tr = nullptr;
  const size_t mm2 = static_cast<size_t>(m) * m * 2;
  if (tracing) {
    tm = trace->mats ? trace->mats + static_cast<size_t>(trace->recorded) * (3 + 2 * n_shifts) * mm2 : nullptr;
    tr = trace->res ? trace->res + static_cast<size_t>(trace->recorded) * (1 + n_shifts) : nullptr;
    if (tm) {
      std::memset(tm, 0, sizeof(double) * (3 + 2 * n_shifts) * mm2);
      st->alpha.store(tm);
      st->rho.store(tm + mm2);
      st->delta.store(tm + 2 * mm2);
    }
    if (tr) {
      tr[0] = st->residual;
      for (int s = 0; s < n_shifts; ++s) tr[1 + s] = -1.0;
    }
  }
  // Coefficients of every active shift (host, m x m), then ONE pass over the fields (phase C):
  //   Q <- Q rho^{-1} (:152) ; X_0 += P_0 alpha delta_old (:145) ; P_0 <- P_0 rho^dagger + Q (:158)
  //   X_s += P_s alpha_s (:175) ; P_s <- P_s beta_s rho^dagger + Q (:177)
  const CMat rho_dag = st->rho.adjoint();
  std::vector<CMat> Acoef(1, alpha_delta), Bcoef(1, rho_dag);
  std::vector<bcg_field*> Xa(1, st->X[0]), Pa(1, st->P[0]);
  std::vector<CMat> A_by_shift(n_shifts), B_by_shift(n_shifts);
  const int n_active = st->n_unconverged;
  for (int s = n_active - 1; s > 0; --s) {  // :161
    const CMat beta_s_inv = Identity + (sigma[s] - sigma[0]) * st->alpha +
                            st->alpha * st->rho_old * st->alpha_inv_old * (Identity - st->beta_s[s]) *
                                st->rho_old.adjoint();                                                   // :163-165
    st->beta_s[s] = bcg::inverse_full_pivot(beta_s_inv);                                                 // :166
    st->alpha_s[s] = st->beta_s[s] * st->alpha * st->rho_old * st->alpha_inv_old * st->alpha_s[s];       // :167-168
    const double residual_shift = max_ratio((st->rho * st->alpha_inv * st->alpha_s[s]).row_norms(), st->b_norm);  // :169-172
    Acoef.push_back(st->alpha_s[s]);                                                                     // :175
    Bcoef.push_back(st->beta_s[s] * rho_dag);                                                            // :177
    A_by_shift[s] = Acoef.back();
    B_by_shift[s] = Bcoef.back();
    Xa.push_back(st->X[s]);
    Pa.push_back(st->P[s]);
    if (tm) {
      st->alpha_s[s].store(tm + (3 + s) * mm2);
      st->beta_s[s].store(tm + (3 + n_shifts + s) * mm2);
    }
    if (tr) tr[1 + s] = residual_shift;
    if (residual_shift < st->eps_shifts) --st->n_unconverged;  // :179-181
  }
  const bool lazy = lazy_q_width(c, m);
  const bool next_certain = more_follow && st->residual > st->eps;
  if (static_cast<int>(st->pending.size()) + 1 < st->depth && next_certain && n_active >= 2) {
    // shift 0 now, the others in a later iteration's pass (phase_C_multi)
    BCG_TRY(phase_C(c, st->Q, st->rho, Xa.data(), Pa.data(), 1, Acoef, Bcoef, lazy ? &st->q_rinv : nullptr));
    DeferredIteration d;
    d.Q = st->Q;
    d.rinv = st->q_rinv;
    d.n_active = n_active;
    d.A = A_by_shift;
    d.B = B_by_shift;
    st->pending.push_back(d);
  } else if (!st->pending.empty()) {
    std::vector<DeferredIteration> pend;
    pend.swap(st->pending);  // whatever happens below, these updates are not applied a second time (sbcgrq_flush_pending)
    const int rc = phase_C_multi(c, pend, st->Q, st->rho, st->X.data(), st->P.data(), alpha_delta, rho_dag, n_active,
                                 A_by_shift, B_by_shift, lazy ? &st->q_rinv : nullptr, lazy);
    for (const DeferredIteration& d : pend) {
      if (!st->T) st->T = d.Q;
      else st->Qfree.push_back(d.Q);
    }
    BCG_TRY(rc);
  } else {
    BCG_TRY(phase_C(c, st->Q, st->rho, Xa.data(), Pa.data(), static_cast<int>(Xa.size()), Acoef, Bcoef,
                    lazy ? &st->q_rinv : nullptr));
  }
  st->q_lazy = lazy;  // from now on the stored Q is un-normalised: Q_true = Q q_rinv
  if (tracing) trace->recorded += 1;
  return BCG_OK;
}

// An iteration failed (thinQR breakdown, a non-finite Gram matrix, a HIP or communication error) while earlier
// iterations' updates of the shifts >= 1 were still waiting for the pass that closes their group: apply them now, so that
// the X_s the caller keeps are those of the last completed iteration for every shift, as in the reference, where every
// shift is current at any point an error could surface (inc/block_solvers.hpp:161-181 run in every iteration).
int sbcgrq_flush_pending(bcg_sbcgrq_state* st) {
  if (st->pending.empty()) return BCG_OK;
  bcg_context* c = st->c;
  const int m = st->m;
  std::vector<DeferredIteration> pend;
  pend.swap(st->pending);
  const DeferredIteration last = pend.back();
  pend.pop_back();
  const bool lazy = lazy_q_width(c, m);
  int rc = BCG_OK;
  if (pend.empty()) {  // one iteration: the ordinary phase C kernel over the shifts >= 1, its rho^-1 applied in registers
    for (int s0 = 1; s0 < last.n_active && rc == BCG_OK;) {
      const int ns = std::min(bcg::phaseC_max_shifts(m, false), last.n_active - s0);
      const CMat unused = CMat::identity(m);  // slot 0 is skipped by the kernel when the block is stored normalised
      std::vector<const CMat*> mats(1, lazy ? &last.rinv : &unused);
      double2* Xp[8];
      double2* Pp[8];
      for (int k = 0; k < ns; ++k) {
        mats.push_back(&last.A[s0 + k]);
        mats.push_back(&last.B[s0 + k]);
        Xp[k] = st->X[s0 + k]->d;
        Pp[k] = st->P[s0 + k]->d;
      }
      const double2* Md;
      rc = upload_mats(c, m, mats.data(), static_cast<int>(mats.size()), &Md);
      if (rc != BCG_OK) break;
      bcg::launch_phaseC(c->stream, m, rows_of(last.Q), last.Q->d, Xp, Pp, ns, Md, lazy ? 2 : 0, c->row_blocks_C);
      rc = check_launch(c, "phaseC");
      s0 += ns;
    }
  } else {
    const CMat none;
    rc = phase_C_multi(c, pend, last.Q, none, st->X.data(), st->P.data(), none, none, last.n_active, last.A, last.B, nullptr,
                       lazy, &last.rinv);
  }
  pend.push_back(last);
  for (const DeferredIteration& d : pend) {
    if (d.Q == st->Q || d.Q == st->T) continue;
    if (!st->T) st->T = d.Q;
    else st->Qfree.push_back(d.Q);
  }
  if (rc == BCG_OK) rc = stream_sync(c);
  return rc;
}

}  // namespace

extern "C" {

int bcg_sbcgrq_begin(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* X, bcg_field* B, int n_shifts,
                     const double* sigma, double eps, double eps_shifts, int consume_B, bcg_sbcgrq_state** out) {
  DeviceScope on_device(c);
  if (!c || !g || !X || !B || !sigma || !out || n_shifts < 1 || g->ctx != c || B->ctx != c) return BCG_ERR_INVALID;
  const int m = B->m;
  for (int s = 0; s < n_shifts; ++s)
    if (!same_shape(X[s], B) || X[s] == B) BCG_FAIL(c, BCG_ERR_INVALID, "SBCGrQ: X[s] must be distinct fields of B's width");
  // :97-101
  if (sigma[0] < 0.0) BCG_FAIL(c, BCG_ERR_INVALID, "SBCGrQ: shifts must be zero or positive");
  if (!std::is_sorted(sigma, sigma + n_shifts)) BCG_FAIL(c, BCG_ERR_INVALID, "SBCGrQ: shifts must be in ascending order");
  BCG_TRY(ensure_scratch(c));
  bcg_sbcgrq_state* st = new bcg_sbcgrq_state();
  st->c = c;
  st->g = g;
  st->mass = mass;
  st->m = m;
  st->n_shifts = n_shifts;
  st->sigma.assign(sigma, sigma + n_shifts);
  st->eps = eps;
  st->eps_shifts = eps_shifts;
  st->X.assign(X, X + n_shifts);
  st->B = B;
  st->n_unconverged = n_shifts;                  // :104
  const CMat Identity = CMat::identity(m);       // :106
  st->alpha = st->rho = st->delta = CMat(m);     // :107
  st->alpha_inv = Identity;                      // :108
  st->alpha_inv_old = st->rho_old = CMat(m);
  st->P.assign(n_shifts, nullptr);
#define BEGIN_TRY(call)          \
  do {                           \
    int rc_ = (call);            \
    if (rc_ != BCG_OK) {         \
      sbcgrq_release(st);        \
      delete st;                 \
      return rc_;                \
    }                            \
  } while (0)
  // Every allocation the solve cannot do without comes FIRST and in one stretch -- T, Q (:109), the P_s (:117), then what
  // the operator needs (tmp or the ring, face buffers, partials; the first iteration would otherwise allocate it lazily,
  // and it must come before the optional residual buffers below: a solve that fits without them must not fail because
  // they took the room) -- so that on a lattice divided over ranks the ranks can AGREE on the outcome before the first
  // collective of the solve (the Gram all-reduce inside thinQR, :115): a rank that ran out of memory would otherwise leave
  // its peers waiting in that all-reduce for ever.  With the agreement every rank returns from a failed begin, with
  // nothing allocated, and the caller can try again with a smaller plan (bench.py steps its ladder down this way).
  int alloc_rc = create_like(c, B, &st->T);  // T is overwritten before it is read, so it is not initialised from B
  if (alloc_rc == BCG_OK) {
    if (consume_B) st->Q = B;
    else alloc_rc = create_like(c, B, &st->Q);
  }
  for (int s = 0; s < n_shifts && alloc_rc == BCG_OK; ++s) alloc_rc = create_like(c, B, &st->P[s]);
  if (alloc_rc == BCG_OK) alloc_rc = reserve_operator_scratch(c, B);
  if (c->distributed && c->have_comm && c->comm.allreduce_sum) {
    const std::string why = c->err;
    (void)hipGetLastError();
    double failed_ranks = alloc_rc == BCG_OK ? 0.0 : 1.0;
    int rc_ = BCG_OK;
    *reinterpret_cast<double*>(c->pin_gram) = failed_ranks;
    if (hipMemcpyAsync(c->dev_gram, c->pin_gram, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc_ = BCG_ERR_HIP;
    if (rc_ == BCG_OK && c->comm.allreduce_sum(c->comm.user, c->dev_gram, 1) != 0) rc_ = BCG_ERR_COMM;
    if (rc_ == BCG_OK && (hipMemcpyAsync(c->pin_gram, c->dev_gram, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                          hipStreamSynchronize(c->stream) != hipSuccess))
      rc_ = BCG_ERR_HIP;
    if (rc_ == BCG_OK) failed_ranks = *reinterpret_cast<const double*>(c->pin_gram);
    if (alloc_rc == BCG_OK && rc_ != BCG_OK) {
      alloc_rc = rc_;
      c->err = "SBCGrQ: the ranks could not agree on the outcome of their allocations (all-reduce failed)";
    } else if (alloc_rc == BCG_OK && failed_ranks > 0.0) {
      alloc_rc = BCG_ERR_HIP;
      c->err = "SBCGrQ: another rank of the process grid could not allocate the solve's fields (hipErrorOutOfMemory there)";
    } else {
      c->err = why;
    }
  }
  if (alloc_rc != BCG_OK) {
    sbcgrq_release(st);
    delete st;
    return alloc_rc;
  }
  if (!consume_B) BEGIN_TRY(bcg_field_copy(st->Q, B));
  for (int s = 0; s < n_shifts; ++s) BEGIN_TRY(bcg_field_set_zero(X[s]));  // :111-113
  BEGIN_TRY(thin_qr(c, st->Q, st->delta));                                  // :115
  st->rho = st->delta;                                                      // :116
  for (int s = 0; s < n_shifts; ++s) BEGIN_TRY(bcg_field_copy(st->P[s], st->Q));  // :117
#undef BEGIN_TRY
  st->alpha_s.assign(n_shifts, Identity);  // :122
  st->beta_s.assign(n_shifts, Identity);   // :123
  st->depth = pair_shifts_depth(c, m, n_shifts);
  for (int k = 2; k < st->depth; ++k) {  // depth 2 needs none (T doubles as the second residual buffer)
    bcg_field* q = nullptr;
    if (create_like(c, B, &q) != BCG_OK) {  // no room for another residual buffer: a smaller depth
      (void)hipGetLastError();
      c->err.clear();
      st->depth = k;
      break;
    }
    st->Qfree.push_back(q);
  }
  st->iter = 0;                            // :126
  st->b_norm = st->delta.row_norms();      // :130
  st->residual = 1.0;                      // :131
  *out = st;
  return BCG_OK;
}

int bcg_sbcgrq_iterate(bcg_sbcgrq_state* st, int max_new_iterations, int* iterations_total, double* residual_out,
                       bcg_sbcgrq_trace* trace) {
  DeviceScope on_device(st ? st->c : nullptr);
  if (!st) return BCG_ERR_INVALID;
  if (st->failed) BCG_FAIL(st->c, BCG_ERR_INVALID, "SBCGrQ: an earlier iteration on this state returned an error");
  int done = 0;
  while (st->residual > st->eps && done < max_new_iterations) {  // :132
    const int rc = sbcgrq_iteration(st, trace, done + 1 < max_new_iterations);
    if (rc != BCG_OK) {
      st->failed = true;
      const std::string why = st->c->err;
      (void)hipGetLastError();
      if (sbcgrq_flush_pending(st) != BCG_OK) st->c->err = why + " (and the deferred updates of the shifted systems could not be applied)";
      else st->c->err = why;
      return rc;
    }
    ++done;
  }
  BCG_TRY(stream_sync(st->c));
  if (iterations_total) *iterations_total = st->iter;
  if (residual_out) *residual_out = st->residual;
  return BCG_OK;
}

int bcg_sbcgrq_end(bcg_sbcgrq_state* st) {
  DeviceScope on_device(st ? st->c : nullptr);
  if (!st) return BCG_OK;
  (void)stream_sync(st->c);
  sbcgrq_release(st);
  delete st;
  return BCG_OK;
}

int bcg_sbcgrq_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* X, bcg_field* B, int n_shifts,
                     const double* sigma, double eps, double eps_shifts, int max_iterations, int consume_B,
                     int* iterations_out, double* residual_out, bcg_sbcgrq_trace* trace) {
  DeviceScope on_device(c);
  bcg_sbcgrq_state* st = nullptr;
  if (trace) trace->recorded = 0;
  BCG_TRY(bcg_sbcgrq_begin(c, g, mass, X, B, n_shifts, sigma, eps, eps_shifts, consume_B, &st));
  const int rc = bcg_sbcgrq_iterate(st, max_iterations, iterations_out, residual_out, trace);  // :184
  bcg_sbcgrq_end(st);
  return rc;
}

}  // extern "C"
