// Launchers of the MFMA fast path (kernels_mfma.hip: row kernels, with the data ownership scheme; kernels_stencil.hip: the stencil).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace bcg {

bool mfma_rows_width(int m);    // widths whose right-multiplications (phase C, K5, K6) run on MFMA (8, 16, 32)
bool hop_fast_width(int m);     // widths served by the LDS-staged stencil kernel (8, 16, 32)
int phaseC_max_shifts(int m, bool applies_rinv);  // shifts one phase-C launch can take (LDS budget; Rinv takes a slot)

// Gram partials folded inside the producing kernel (mfma_common.hpp: gram_fold): the last block to finish sums the block
// partials in a fixed order into `out`, so no reduction launch follows.  out = nullptr: off.  The partials buffer needs
// room for 8 more entries than blocks; tickets: 9 words, zero before the first use (the kernel leaves them zero).
struct GramFold {
  double2* out = nullptr;
  unsigned* tickets = nullptr;
};

// Phase B: [Q <- Q * rinv if rinv] ; Q += T * negalpha ; partials of (new Q)^dagger (new Q).  Returns blocks used.
int launch_phaseB(hipStream_t s, int m, int64_t rows, double2* Q, const double2* T, const double2* negalpha,
                  double2* partials, int max_blocks, GramFold fold = GramFold(), const double2* rinv = nullptr,
                  double2* Qout = nullptr);  // Qout: the new Q goes there and Q is left as it was (nullptr: in place)
// Phase C: q = Q*mats[0] if apply_rinv (1: stored back to Q; 2: used, not stored); for k < nshift:
// X[k] += P[k]*mats[1+2k]; P[k] <- P[k]*mats[2+2k] + q.
void launch_phaseC(hipStream_t s, int m, int64_t rows, double2* Q, double2* const* X, double2* const* P, int nshift,
                   const double2* mats, int apply_rinv, int max_blocks);
// Phase C of nsteps = 2 .. 4 consecutive iterations in one pass (kernels_mfma.hip: k_phaseC_multi).  Q[j]: the
// residual block of step j -- un-normalised with normalise = true (m = 8, 16: mats starts with rinv_0 .. rinv_{nsteps-1}),
// as stored otherwise (m = 32) -- entry e takes the steps first[e] <= j < last[e]; then per entry and step A, B.
bool phaseC_multi_fits(int m, int nsteps, int n_shifts);  // the grouping is available for this width and depth
// coefficient matrices one launch has LDS room for (150 KB of a CU's 160): the rinv_j, then per entry its step matrices
// (and entry 0's composed ones, xacc below); at most 8 entries per launch
int phaseC_multi_capacity(int m);
// xacc > 0 (normalise only; deferred update of X_0): before entry 0's steps, X[0] += p1 C_0 + q_0 C_1 + ... + q_{xacc-2} C_{xacc-1},
// the xacc composed matrices following entry 0's step matrices in `mats`
void launch_phaseC_multi(hipStream_t s, int m, int64_t rows, int nsteps, const double2* const* Q, double2* const* X,
                         double2* const* P, int nent, const int* first, const int* last, const double2* mats, int max_blocks,
                         bool normalise = true, int xacc = 0, const double2* p1 = nullptr);
// Shift 0's phase C with the update of X_0 deferred (m = 8, 16): Pout = P mats[1] + Q mats[0]; mats = [rinv, B]
void launch_phaseC_p0(hipStream_t s, int m, int64_t rows, const double2* Q, const double2* P, double2* Pout, const double2* mats,
                      int max_blocks);
void launch_rmul_mfma(hipStream_t s, int m, int64_t rows, double2* y, const double2* x, const double2* Cd, double b,
                      RmulMode mode, int max_blocks);
int launch_gram_mfma(hipStream_t s, int m, int64_t rows, const double2* a, const double2* b, double2* partials,
                     int max_blocks);
// Pacing counters of the specialised stencil (kernels_stencil.hip, HopWalk::sync); owned by the context.
struct HopSync {
  unsigned* counters = nullptr;  // 8 * stride, device memory
  int stride = 0;                // tiles per block the buffer has room for
  int window = 4;                // slices a block may run ahead of the slowest block of its XCD class (0 = no pacing);
                                 // measured at 64^4: 2 -> 13.3 ms, 3 -> 11.0, 4..8 -> 10.9 (unpaced 13.4)
  int limit_ticks = 5000;        // 100 MHz ticks (50 us) a block waits before it gives up pacing
  bool column_walk = true;       // use k_hop4c (scalar row pointers, column sweep) where the walk allows it
  int bundle_walk = 1;           // k_hop4b (2 x 2 column bundles sharing rows through LDS) for whole launches: 0 off, 1 at
                                 // m = 16 and 32 and for the plain hop at m = 8, 2 for every launch at m = 8 too
  int bundle_window = 4;         // pacing window of k_hop4b (0 = unpaced; < 0: |value|, also where bundle_paced() would not pace).  Round 3, links by LDS-DMA: windows 4..12 take the
                                 // same time as the unpaced sweep (10.50 vs 10.46 ms, 12.00 vs 12.06 ms at 64^4) and move
                                 // 20 % fewer bytes past the L2 (45.7 vs 57.2 GB, 59.0 vs 72.0 GB); window 3 costs 2 %
};

// Tuning of the specialised 4-D stencil (defaults chosen by measurement at 64^4, m = 16).
struct HopTuning {
  bool patch_walk = true;        // per-XCD patches swept along x3 (false: lexicographic tile order)
  int patch[3] = {0, 8, 8};      // patch extents in x0, x1, x2; 0 in x0 = one tile (16 sites at m = 16)
  int blocks = 512;              // persistent grid: 2 blocks per CU at the kernel's register budget
  HopSync sync;                  // pacing of the blocks of an XCD along x3
  const int* boundary_list = nullptr;  // device list of the boundary tiles' first sites (set per launch by the context)
  int boundary_n = 0;
  bool nontemporal = true;       // stream `out` (and p) past L2 (the only form instantiated)
  GramFold fold;                 // set per launch by the context: fold the fused Gram partials in the kernel (column forms)
  int super_patch = 0;           // k_hop4b: the eight XCD classes' concurrent patches form a 2 x 2 x 2 super-patch (HopWalk::super;
                                 // BCG_HOP_SUPER; measured: profiles/r05_stencil_super_patch.txt)
  int blocks_overlap = 512;      // grid of the interior launch while a halo exchange is in flight.  Measured: any grid whose
                                 // per-XCD share differs from the 64 tiles of a patch slice loses the x3 walk (480 blocks: +3 ms),
                                 // so CUs are not vacated for the transport; its kernels co-reside where registers allow
};

// Restriction of one stencil launch to the x3 slices [x3_lo, x3_lo + x3_n) and, with ring > 0, the capacity-mode
// addressing of the intermediate field of dirac_op::op (inc/dirac_op.hpp:39, `tmp`): slice x3 of that field lives in slot
// x3 % ring of a buffer of `ring` slices.  HOP_PLAIN writes its output there, HOP_SHIFTED reads its input from
// there (U, p, the HOP_SHIFTED output and the ghost faces keep the whole-lattice addressing).  Specialised 4-D kernel only.
struct HopWindow {
  int x3_lo = 0, x3_n = 0;  // x3_n = 0: all slices
  int ring = 0;
  // Checkerboard form (half-volume fields): cb = 1, `lat` is the COMPACT lattice (L[0] halved), `out`/`p` hold the sites of
  // parity cb_parity, `in` those of the other parity.  k_hop4b, whole launches on an undivided lattice only.
  int cb = 0, cb_parity = 0;
};

// Stencil; with gram (m = 16, HOP_SHIFTED) also writes partials of p^dagger out.  Returns blocks used.
int launch_hop_fast(hipStream_t s, int m, const LatticeDev& lat, const double2* U, const double2* Ughost,
                    const double2* in, const double2* ghost, double2* out, HopMode mode, const double2* p, double c0,
                    double2* partials, bool gram, int max_blocks, const HopTuning& tune, int tile_class,
                    const HopWindow& win = HopWindow());
// true when launch_hop_fast can process interior (tile_class 1) and boundary (2) tiles in separate launches
bool hop_can_split_tiles(int m, const LatticeDev& lat);
// Which kernel launch_hop_fast will use: 0 none (the caller runs k_hop_generic), 1 k_hop4 (tile counter), 2 k_hop4c (column sweep), -1 rejected
int hop_kernel_form(int m, const LatticeDev& lat, int max_blocks, const HopTuning& tune, int tile_class, const HopWindow& win);
// true when a whole launch (tile class 0) with the fused Gram product folds its partials itself (HopTuning::fold honoured)
bool hop_folds_gram(int m, const LatticeDev& lat, int max_blocks, const HopTuning& tune, const HopWindow& win);
// true when form 2 is served by k_hop4b (2 x 2 column bundles) rather than k_hop4c
bool hop_uses_bundle(int m, const LatticeDev& lat, int max_blocks, const HopTuning& tune, int tile_class, const HopWindow& win,
                     bool plain = false);

}  // namespace bcg
