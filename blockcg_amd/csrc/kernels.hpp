// Launcher declarations for the HIP kernels of the SBCGrQ hot path (gfx950 only).
//
// Device layout of a block field of width m over V sites (private to the library):
//     double2 f[(site*3 + colour)*m + rhs]           "row" r = site*3 + colour, m complex per row
// so a field is a tall row-major real matrix of 3V rows x 2m columns with (re,im) interleaved.
// The reference's host layout is [site][rhs][colour] (inc/fields.hpp:19-20); conversion happens
// only in upload/download.
// Gauge links keep the reference's layout [site][mu][3x3 column-major] (inc/dirac_op.hpp:10-11).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bcg {

// Geometry of this rank's sub-lattice, passed by value to kernels.
struct LatticeDev {
  int ndim;
  int L[4];              // local extents (1 beyond ndim)
  int origin[4];         // global coordinate of local (0,0,0,0)
  int split[4];          // 1 if the direction is divided over ranks (neighbours live in ghost faces)
  int64_t V;             // local sites
  int64_t stride[4];     // site-index stride of direction mu
  int64_t ghost_off[4][2];  // first ghost SITE of direction mu: [0] = minus face (x_mu = -1), [1] = plus face (x_mu = L)
  int64_t face_sites[4];    // V / L[mu]
};

bool width_supported(int m);
bool mfma_width(int m);  // widths with an MFMA fast path

// ---- elementwise / layout --------------------------------------------------------------------
// y = a*y + b*x over n complex elements (K2, K3, K9, operator+=)
void launch_axpby(hipStream_t s, double2* y, double a, const double2* x, double b, int64_t n);
// SCG: x_s += alpha_s p_s ; p_s = beta_s p_s + zeta_s r for all listed shifts in one pass (bit-identical to the axpys)
void launch_scg_update(hipStream_t s, const double2* r, int nshift, double2* const* x, double2* const* p,
                       const double* alpha, const double* beta, const double* zeta, int64_t n);
void launch_host_to_dev(hipStream_t s, int m, const double2* host_layout, double2* dev_layout, int64_t nsites);
void launch_dev_to_host(hipStream_t s, int m, const double2* dev_layout, double2* host_layout, int64_t nsites);
void launch_fill_field(hipStream_t s, int m, const LatticeDev& lat, const int* gdims, double2* f, uint64_t seed);
void launch_fill_gauge(hipStream_t s, const LatticeDev& lat, const int* gdims, double2* U, uint64_t seed);

// ---- halo -------------------------------------------------------------------------------------
// Gather the low (x_mu = 0) and high (x_mu = L-1) faces of every split direction into `send`:
// per split mu, [low face][high face], each face_sites[mu] * 3m complex.
// x3_n > 0 restricts the faces to the slices [x3_lo, x3_lo + x3_n) of an undivided direction 3 (a contiguous range of
// every face); ring > 0: f is a ring of `ring` x3-slices, slice x3 in slot x3 % ring (capacity mode).
void launch_pack_faces(hipStream_t s, int m, const LatticeDev& lat, const double2* f, double2* send, int x3_lo = 0,
                       int x3_n = 0, int ring = 0);
// Same for the gauge links U_mu of direction mu only (9 complex per site).
void launch_pack_gauge_faces(hipStream_t s, const LatticeDev& lat, const double2* U, double2* send);

// ---- generic (any supported m) kernels ---------------------------------------------------------
enum HopMode { HOP_PLAIN = 0, HOP_SHIFTED = 1, HOP_RESID = 2 };
// HOP_PLAIN  : out = D in                          (K1, inc/dirac_op.hpp:14-21)
// HOP_SHIFTED: out = c0 * p - D in                 (second D of op fused with K2 and K3)
// HOP_RESID  : nothing is written; r = c0 * p - D in - b with b passed in `out`'s place, and the block partials of
//              r^dagger r are left like a fused Gram product (the reference's residual check, test/solvers.cpp:105-111,
//              as one pass; specialised bundle kernel at m = 16 only, launch_hop_fast returns -1 otherwise)
void launch_hop_generic(hipStream_t s, int m, const LatticeDev& lat, const double2* U, const double2* Ughost,
                        const double2* in, const double2* ghost, double2* out, HopMode mode, const double2* p,
                        double c0);
// Half-volume (parity-compact) fields, lattices with even extents: out (parity) = D in (1 - parity), or c0 * p - D in
// (ghost: the half ghost faces of `in`, packed by launch_pack_faces_half -- per split mu [low face][high face], each
// face_sites[mu] / 2 sites, face site f of the full numbering at f >> 1; Ughost: the gauge ghost, full numbering);
// the copy between a full field and a half; the counter generator restricted to one parity.
void launch_pack_faces_half(hipStream_t s, int m, const LatticeDev& lat, int parity, const double2* f, double2* send,
                            int x3_lo = 0, int x3_n = 0);  // x3_n > 0: those slices only, as launch_pack_faces
void launch_hop_half(hipStream_t s, int m, const LatticeDev& lat, int parity, const double2* U, const double2* Ughost,
                     const double2* in, const double2* ghost, double2* out, HopMode mode, const double2* p, double c0);
void launch_parity_copy(hipStream_t s, int m, const LatticeDev& lat, int parity, double2* full, double2* half, bool to_half);
void launch_fill_field_half(hipStream_t s, int m, const LatticeDev& lat, const int* gdims, int parity, double2* f, uint64_t seed);

enum RmulMode { RMUL_ADD = 0, RMUL_XPAY = 1, RMUL_MUL = 2 };
// RMUL_ADD : y += x * M                (K5)
// RMUL_XPAY: y  = y * M + b * x        (K6)
// RMUL_MUL : y  = y * M
// Md: m x m complex column-major in device memory.
void launch_rmul_generic(hipStream_t s, int m, int64_t rows, double2* y, const double2* x, const double2* Md,
                         double b, RmulMode mode);
// y <- y R^{-1} by forward substitution over columns (K7, inc/fields.hpp:125-136)
void launch_trisolve_generic(hipStream_t s, int m, int64_t rows, double2* y, const double2* Rd);
// partials[block][m*m] of a^dagger b (column-major); returns the number of blocks used.
int launch_gram_generic(hipStream_t s, int m, int64_t rows, const double2* a, const double2* b, double2* partials,
                        int max_blocks);
// out[p] = sum_b partials[b][p], fixed order (deterministic)
void launch_reduce_partials(hipStream_t s, int n_values, int n_blocks, const double2* partials, double2* out);

}  // namespace bcg
