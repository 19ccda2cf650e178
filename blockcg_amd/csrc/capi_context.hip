// include/blockcg_hip.h, part 1 of 3: context, scratch and profiling, host <-> device transfers, the field primitives
// (block_fermion_field<N_rhs>, inc/fields.hpp:25-147) and memory planning.  There is no CPU fallback: without a gfx950 device
// bcg_context_create fails with BCG_ERR_NO_DEVICE.
#include "capi_internal.hpp"

namespace bcg_impl {

std::string g_create_error;

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

// ---- building blocks ---------------------------------------------------------------------------
bool same_shape(const bcg_field* a, const bcg_field* b) {
  return a && b && a->ctx == b->ctx && a->m == b->m && a->parity == b->parity;
}


// Block partials in c->partials -> G (m x m), summed over blocks in a fixed order and over ranks,
// Hermitian-mirrored exactly as inc/fields.hpp:115-120.
// folded: the producing kernel has already summed them into c->dev_gram (bcg::GramFold)
int finish_gram(bcg_context* c, int m, int nblocks, CMat& G, bool mirror, bool folded) {
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
int gram(bcg_context* c, const bcg_field* a, const bcg_field* b, CMat& G, bool mirror) {
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

// a new field of the width, parity and site count of `like`
int create_like(bcg_context* c, const bcg_field* like, bcg_field** out) {
  return like->parity >= 0 ? bcg_field_create_half(c, like->m, like->parity, out) : bcg_field_create(c, like->m, out);
}

}  // namespace bcg_impl

using namespace bcg_impl;

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
  if (const char* e = std::getenv("BCG_HOP_SUPER")) c->hop_tune.super_patch = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_SYNC")) c->hop_tune.sync.window = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_SYNC_LIMIT")) c->hop_tune.sync.limit_ticks = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_COLUMN")) c->hop_tune.sync.column_walk = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HOP_BUNDLE")) c->hop_tune.sync.bundle_walk = std::atoi(e);
  if (const char* e = std::getenv("BCG_HOP_BUNDLE_SYNC")) c->hop_tune.sync.bundle_window = std::atoi(e);
  if (const char* e = std::getenv("BCG_LAZY_Q")) c->lazy_q = std::atoi(e);  // 2: at m = 32 too (tuning)
  if (const char* e = std::getenv("BCG_PAIR_SHIFTS")) c->pair_shifts = std::atoi(e);  // depth (pair_shifts_depth)
  if (const char* e = std::getenv("BCG_DEFER_X0")) c->defer_x0 = std::atoi(e) != 0;   // deferred update of X_0 (DeferredX0)
  if (const char* e = std::getenv("BCG_DEBUG_X0_COND_LIMIT")) c->x0_cond_limit = std::atof(e);  // test aid: 0 = the guard always refuses
  if (const char* e = std::getenv("BCG_RING_CHUNK")) c->ring_chunk_override = std::atoi(e);
  if (const char* e = std::getenv("BCG_DEBUG_FIELD_BUDGET")) c->debug_field_budget = static_cast<size_t>(std::atoll(e));
  if (const char* e = std::getenv("BCG_DEBUG_FAIL_ITER")) c->debug_fail_iter = std::atoi(e);
  if (const char* e = std::getenv("BCG_RING_OVERLAP")) c->ring_overlap = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HALF_CHUNK_FORCE")) c->half_chunk_force = std::atoi(e) != 0;
  if (const char* e = std::getenv("BCG_HALF_CHUNK")) c->half_chunk_override = std::atoi(e);  // x3 chunk of the half-volume sweep (tests, tuning)
  if (const char* e = std::getenv("BCG_FORCE_TILE_CLASSES")) c->force_tile_classes = std::atoi(e) != 0;
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
  if (group_depth >= 2 && ring_slices == 0 && (m == 8 || m == 16)) total += field;        // the spare P_0 of the deferred X_0 update
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
  if (pair_shifts_depth(c, m, n_shifts) >= 2 && c->defer_x0 && (m == 8 || m == 16)) total += half;  // the spare P_0 (DeferredX0)
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
  // (Fields of the lattices that matter have power-of-two sizes -- 64^4 sites x 768 B = 12 GiB -- and come back to back from
  //  the driver; a per-field stagger of the virtual addresses was tried against the placement sensitivity of the streaming
  //  kernels and made it worse on average: profiles/r04_phaseC_placement.txt.  Fields are plain allocations.)
  hipError_t e = (c->debug_field_budget && c->field_bytes_live + field_bytes(f) > c->debug_field_budget)
                     ? hipErrorOutOfMemory  // test aid: a deterministic stand-in for a full device (tests/test_robustness.py)
                     : hipMalloc(&f->base, field_bytes(f));
  if (e != hipSuccess) {
    delete f;
    c->err = std::string("bcg_field_create: hipMalloc: ") + hipGetErrorString(e);
    return BCG_ERR_HIP;
  }
  f->d = static_cast<double2*>(f->base);
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

}  // extern "C"
