// Internals shared by the three files that implement include/blockcg_hip.h (round 5: blockcg_capi.hip split by subsystem):
//   capi_context.hip    context, scratch and profiling, host <-> device transfers, field primitives, memory planning
//   capi_operator.hip   halo exchange, the stencil launches, dirac_op::op (whole tmp, capacity ring, half-volume), gauge API
//   capi_solvers.hip    SBCGrQ (phases A / B / C, grouped and deferred updates), CG, SCG, BCG, BCGrQ, true residuals
// Host code only; every loop over lattice sites is a HIP kernel (kernels_generic.hip, kernels_mfma.hip, kernels_stencil.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "context.hpp"
#include "kernels_mfma.hpp"

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

namespace bcg_impl {

using bcg::CMat;
using bcg::cd;

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

constexpr int kMaxGramBlocks = 2048;  // also covers interior + boundary stencil launches (2 x 1024)
constexpr size_t kMatSlotBytes = 32 * 32 * sizeof(double2);
constexpr int kMatSlots = 96;

constexpr int kFastBlocks = 1024;  // persistent-style grids: 4 blocks per CU

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


// ---- capi_context.hip ------------------------------------------------------------------------------
void resolve_profile(bcg_context* c);
int stream_sync(bcg_context* c);
int check_launch(bcg_context* c, const char* what);
int ensure_halo(bcg_context* c, size_t bytes);
int ensure_scratch(bcg_context* c);
// Copy n coefficient matrices (m x m each) to consecutive device slots; slots are recycled only after a stream synchronization
int upload_mats(bcg_context* c, int m, const CMat* const* mats, int n, const double2** dev_out);
int upload_mat(bcg_context* c, const CMat& M, const double2** dev_out);
bool same_shape(const bcg_field* a, const bcg_field* b);
inline bool fast_rows(const bcg_context* c, int m) { return !c->force_generic && bcg::mfma_width(m); }       // + Gram, phase B
inline bool fast_rmul(const bcg_context* c, int m) { return !c->force_generic && bcg::mfma_rows_width(m); }  // products, phase C
// (the specialised stencil kernels: 4-D lattices whose L0 is a multiple of the tile; everything else is k_hop_generic)
inline bool fast_hop(const bcg_context* c, int m) { return !c->force_generic && bcg::hop_fast_width(m) && bcg::hop_can_split_tiles(m, c->lat); }
// Block partials in c->partials -> G (m x m), summed over blocks in a fixed order and over ranks, Hermitian-mirrored
int finish_gram(bcg_context* c, int m, int nblocks, CMat& G, bool mirror, bool folded = false);
int gram(bcg_context* c, const bcg_field* a, const bcg_field* b, CMat& G, bool mirror = true);  // G = a^dagger b
int rmul(bcg_context* c, bcg_field* y, const bcg_field* x, const CMat& M, double b, bcg::RmulMode mode, const char* name);
int trisolve(bcg_context* c, bcg_field* y, const CMat& R);
int axpby(bcg_context* c, bcg_field* y, double a, const bcg_field* x, double b, const char* name);
int create_like(bcg_context* c, const bcg_field* like, bcg_field** out);  // a new field of the width, parity and site count of `like`

// ---- capi_operator.hip -----------------------------------------------------------------------------
int halo_plan(int ndim, const int* gdims, const int* grid, const int* coords, size_t site_bytes, int* peer_s, int* peer_r,
              size_t* off_s, size_t* off_r, size_t* nb, int64_t* ghost_sites);
inline bool can_overlap(const bcg_context* c) {
  return c->distributed && c->have_comm && c->comm.halo_exchange_begin && c->comm.halo_exchange_end;
}
int halo_field(bcg_context* c, const bcg_field* f, bool split = false);
// out = D in (HOP_PLAIN) or c0 * p - D in (HOP_SHIFTED); gram_blocks: also block partials of p^dagger out in c->partials
int hop(bcg_context* c, const bcg_gauge* g, bcg_field* out, const bcg_field* in, bcg::HopMode mode, const bcg_field* p, double c0,
        int* gram_blocks = nullptr, bool* gram_folded = nullptr);
int get_tmp(bcg_context* c, int m, bcg_field** out);
bool capacity_path(const bcg_context* c, int m);
// chunk length of the capacity-mode sweep and whether its exchanges are overlapped (apply_shifted_ring)
inline bool ring_overlapped(const bcg_context* c) { return c->ring_overlap && can_overlap(c) && (c->tmp_ring - 2) / 2 >= 1; }
// The device memory apply_shifted needs for operands shaped like `like`, allocated now rather than at the first call
int reserve_operator_scratch(bcg_context* c, const bcg_field* like);
// T = (mass^2 + sigma0) P - D(D(P))   [op + add(P, sigma0), inc/block_solvers.hpp:134-136]
int apply_shifted(bcg_context* c, const bcg_gauge* g, double mass, double sigma0, bcg_field* T, const bcg_field* P,
                  int* gram_blocks = nullptr, bool* gram_folded = nullptr);

// ---- capi_solvers.hip ------------------------------------------------------------------------------
int pair_shifts_depth(const bcg_context* c, int m, int n_shifts);
int thin_qr(bcg_context* c, bcg_field* y, CMat& R);  // thinQR (inc/fields.hpp:140-146)

}  // namespace bcg_impl
