// Context, field and gauge objects behind the opaque handles of include/blockcg_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/blockcg_hip.h"
#include "kernels.hpp"
#include "kernels_mfma.hpp"
#include "small_matrix.hpp"

struct bcg_field {
  bcg_context* ctx;
  int m;
  double2* d;  // [V_local*3][m]
  void* base;  // the allocation d points into (d = base + a stagger, see bcg_field_create)
  int parity = -1;       // -1: all sites of the local lattice; 0 / 1: the sites of that parity only (bcg_field_create_half)
  int64_t sites = 0;     // sites it holds: V_local, or V_local / 2
};

struct bcg_gauge {
  bcg_context* ctx;
  double2* U;       // [V_local][ndim][9]
  double2* Ughost;  // [ghost sites][9], only the minus faces are filled
  bool ghost_valid;
};

namespace bcg {

struct ProfEntry {
  double ms = 0.0;
  long count = 0;
  double bytes = 0.0;  // algorithmic HBM bytes of the launches timed under this name
  double flops = 0.0;  // their fp64 flops (0 where not accounted)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace bcg

struct bcg_context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int ndim = 1;
  int gdims[4] = {1, 1, 1, 1};
  int grid[4] = {1, 1, 1, 1};
  int coords[4] = {0, 0, 0, 0};
  bcg::LatticeDev lat{};
  int64_t ghost_sites = 0;  // total ghost sites over all split directions (both faces)
  bool distributed = false;
  bcg_comm comm{};
  bool have_comm = false;
  bool force_generic = false;
  bool force_tile_classes = false;  // tuning/test aid: interior + boundary stencil launches on an undivided lattice too
  int row_blocks_B = 1024, row_blocks_C = 1024;  // persistent grids of the fused row kernels (phase B, phase C)
  bcg::HopTuning hop_tune;  // specialised stencil: tile walk, patch shape, grid, streaming hints

  // scratch
  std::map<int, bcg_field*> tmp_field;   // per width: the `tmp` of dirac_op::op (inc/dirac_op.hpp:39); key m + 1000 (1 + parity)
                                         // for the half-volume ones
  int tmp_ring = 0;                      // capacity mode: `tmp` kept as a ring of this many x3 slices (0 = whole field)
  std::map<int, double2*> tmp_ring_buf;  // per width: the ring, tmp_ring * stride[3] * 3m complex
  std::map<int, std::pair<int*, int>> boundary_tiles;  // per tile length: device list of the boundary tiles' first sites
  double2* halo_send = nullptr;
  double2* halo_recv = nullptr;
  size_t halo_bytes = 0;
  void* halo_save = nullptr;             // capacity mode, overlapped exchanges: the source's received faces of slice x3 = 0
  size_t halo_save_bytes = 0;
  int lazy_q = 1;                        // SBCGrQ: deferred normalisation of Q (phase_B in capi_solvers.hip; BCG_LAZY_Q)
  int pair_shifts = 4;                   // SBCGrQ: shifts >= 1 updated this many iterations at a time (pair_shifts_depth; BCG_PAIR_SHIFTS)
  bool defer_x0 = true;                  // SBCGrQ: X_0's updates wait for the pass that closes a group too (DeferredX0; BCG_DEFER_X0)
  double x0_cond_limit = 64.0;           // ... spare-less form: taken while ||rho||_F ||rho^-1||_F <= limit * m (BCG_DEBUG_X0_COND_LIMIT: test aid)
  size_t debug_field_budget = 0;         // test aid (BCG_DEBUG_FIELD_BUDGET, bytes): field allocations beyond it fail like an out-of-memory
  size_t field_bytes_live = 0;           // bytes of the fields this context holds (incl. tmp and the solver's work fields)
  int debug_fail_iter = 0;               // test aid (BCG_DEBUG_FAIL_ITER): SBCGrQ iteration whose Gram matrix after phase B is made non-finite
  int ring_chunk_override = 0;           // capacity mode: chunk length in slices when smaller than the ring allows (BCG_RING_CHUNK)
  bool half_chunk_force = false;         // BCG_HALF_CHUNK_FORCE: take the chunked half-volume sweep on an undivided lattice too (timing aid)
  int half_chunk_override = 0;           // > 0: x3 slices per chunk of the overlapped half-volume operator (BCG_HALF_CHUNK; default 16)
  bool ring_overlap = true;              // capacity mode: overlap the per-chunk exchanges when the callbacks allow (BCG_RING_OVERLAP)
  double2* partials = nullptr;           // block partials of Gram products
  size_t partials_bytes = 0;
  double2* dev_mats = nullptr;           // ring of coefficient-matrix slots in device memory
  double* pin_mats = nullptr;            // pinned host mirror of the ring
  size_t mat_slot_bytes = 0;
  int mat_slots = 0, mat_next = 0, mat_in_flight = 0;
  unsigned* fold_tickets = nullptr;      // 9 words: arrival counters of the in-kernel Gram fold (bcg::GramFold)
  double2* dev_gram = nullptr;           // reduced Gram matrix (device), all-reduced in place
  double* pin_gram = nullptr;            // pinned host copy
  double2* staging = nullptr;            // layout-conversion staging of bcg_field_download_sites
  size_t staging_bytes = 0;
  // host <-> device pipeline of bcg_field_upload / bcg_field_download: two chunks in flight, each with its own stream,
  // device staging buffer (layout conversion) and pinned host buffer (only used when the caller's memory is pageable)
  hipStream_t xfer_stream[2] = {nullptr, nullptr};
  hipEvent_t xfer_done[2] = {nullptr, nullptr};
  double2* xfer_dev[2] = {nullptr, nullptr};
  void* xfer_pin[2] = {nullptr, nullptr};
  size_t xfer_bytes = 0;

  // profiling
  bool profiling = false;
  std::map<std::string, bcg::ProfEntry> prof;
  std::vector<hipEvent_t> event_pool;
  std::string prof_json;

  mutable std::string err;
};
