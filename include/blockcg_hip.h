/* blockcg_hip.h -- C ABI of libblockcg_hip.so: the MI355X (gfx950) SBCGrQ hot path.
 *
 * This is the drop-in boundary for the iteration hot path of lkeegan/blockCG.  The reference has no
 * FFI: its boundary is the header-level C++ API of inc/fields.hpp, inc/dirac_op.hpp and
 * inc/block_solvers.hpp.  Every entry point below names the reference interface it replaces
 * (file:line relative to the reference root).  The C++ classes with the reference's names
 * (blockcg_amd/include/blockcg/{fields,dirac_op,block_solvers}.hpp) are thin wrappers over these
 * calls; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; no C++ or torch types.  Every call returns an int status
 *    (BCG_OK = 0); the reference's assert-only error convention (inc/block_solvers.hpp:97-101)
 *    cannot cross a C ABI.  bcg_last_error() returns a message for the last failure on the context.
 *  - Scalars are complex<double> stored as interleaved (re, im) doubles.
 *  - HOST layouts at this boundary are the reference's in-memory layouts:
 *      block field  [site][rhs j][colour c]      48*m bytes per site   inc/fields.hpp:19-20,28-30
 *      m x m matrix column-major (i,j) at j*m+i                         inc/fields.hpp:22-23
 *      gauge links  [site][mu < ndim][3x3 column-major]  144 B per link inc/dirac_op.hpp:10-11
 *    Device layouts are private to the library: fields are [site][colour][rhs] so that
 *    one (site, colour) row of m complex numbers is contiguous.
 *  - Site counts and byte offsets are 64-bit: 128^4 sites fit in int, 128^4 * 768 B does not.
 *  - Block width m (the reference's template parameter N_rhs, an arbitrary int: inc/fields.hpp:19-26) is a run-time
 *    argument: any 1 <= m <= 32 (BCG_ERR_UNSUPPORTED otherwise).  m = 8, 16, 32 run the MFMA row kernels and the
 *    specialised stencil; every other width the generic kernels (one thread per output element, LDS-staged operands).
 *  - Lattice: up to 4 dimensions, lexicographic site order with x0 fastest, periodic.  ndim = 1,
 *    dims = {V} is exactly the reference's 1-D operator (inc/dirac_op.hpp:14-21); the n-D operator
 *    is  (D psi)(x) = 1/2 sum_mu eta_mu(x) [U_mu(x) psi(x+mu) - U_mu(x-mu)^dagger psi(x-mu)],
 *    eta_mu(x) = (-1)^(x_0+...+x_{mu-1})  (global coordinates).
 *  - One context = one GPU = one rank of a process grid.  With grid = {1,1,1,1} nothing
 *    communicates.  With more ranks the caller supplies two callbacks (bcg_comm) that move halo
 *    faces and sum m x m partials.  libblockcg_rccl.so (include/blockcg_rccl.h) implements them natively on RCCL
 *    (grouped ncclSend/ncclRecv, ncclAllReduce) -- what bench.py and examples/multi_gpu_solver.cpp use;
 *    blockcg_amd/comm.py is a torch.distributed implementation kept for rehearsals (gloo: ranks sharing one GPU).
 *  - All work is enqueued on the context's HIP stream; calls that return host data synchronize it.
 */
#ifndef BLOCKCG_HIP_H
#define BLOCKCG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCG_OK 0
#define BCG_ERR_INVALID 1     /* bad argument (null pointer, size mismatch, unsorted shifts ...) */
#define BCG_ERR_UNSUPPORTED 2 /* block width or lattice shape not instantiated */
#define BCG_ERR_HIP 3         /* a HIP runtime call failed; see bcg_last_error */
#define BCG_ERR_NO_DEVICE 4   /* no usable gfx950 device */
#define BCG_ERR_COMM 5        /* a communication callback failed or is missing */
#define BCG_ERR_NUMERIC 6     /* non-finite or non-positive-definite Gram matrix (CholQR breakdown) */

typedef struct bcg_context bcg_context;
typedef struct bcg_field bcg_field; /* device block field: V_local sites x 3 colours x m columns */
typedef struct bcg_gauge bcg_gauge; /* device gauge links of a dirac_op */

/* ---- communication callbacks (multi-GPU only) ------------------------------------------------
 * halo_exchange: the library has packed 2 faces per split direction into the send buffer and now
 *   needs the matching faces of the neighbour ranks in the receive buffer.  Buffers are device
 *   memory owned by the library (bcg_halo_buffers).  n_msgs messages; message k sends
 *   send_bytes[k] bytes at send_offset[k] to rank peer_send[k] and receives the same number of bytes
 *   at recv_offset[k] from rank peer_recv[k]; messages to one peer must match in posting order.
 *   The exchange must be ordered after prior work on the context's stream and complete (or be
 *   stream-ordered) before later work on it.
 * allreduce_sum: sum `count` doubles at device pointer `buf` over all ranks, in place, identically
 *   on every rank, stream-ordered as above. */
typedef struct bcg_comm {
  void* user;
  int (*halo_exchange)(void* user, int n_msgs, const int* peer_send, const int* peer_recv, const size_t* send_offset,
                       const size_t* recv_offset, const size_t* nbytes);
  int (*allreduce_sum)(void* user, void* buf, size_t count);
  /* Optional split form of halo_exchange (both NULL or both set).  begin posts the same messages, ordered after the
   * work already enqueued on the context's stream, and returns without making the stream wait; end makes the stream
   * wait for their completion.  Between the two the library enqueues the stencil over the interior tiles, which read
   * no ghost site, so the exchange overlaps that arithmetic; the boundary tiles follow end.
   * Up to TWO exchanges may be outstanding (capacity mode posts the faces of the source in two windows and the first
   * chunk's `tmp` faces behind them); halo_exchange_end ends the OLDEST one.  Their send and receive ranges are disjoint;
   * a transport may run them one after the other.  n_msgs is at most 16 (a window may consist of two x3 ranges). */
  int (*halo_exchange_begin)(void* user, int n_msgs, const int* peer_send, const int* peer_recv, const size_t* send_offset,
                             const size_t* recv_offset, const size_t* nbytes);
  int (*halo_exchange_end)(void* user);
} bcg_comm;

/* ---- context ------------------------------------------------------------------------------- */
/* device: HIP device ordinal.  stream: a hipStream_t to enqueue on, or NULL for a private stream.
 * ndim, global_dims[ndim]: the whole lattice.  grid[ndim]: ranks per direction (NULL = all 1);
 * coords[ndim]: this rank's position (NULL = all 0).  global_dims[mu] % grid[mu] must be 0. */
int bcg_context_create(bcg_context** ctx, int device, void* stream, int ndim, const int* global_dims, const int* grid,
                       const int* coords);
int bcg_context_destroy(bcg_context* ctx);
const char* bcg_last_error(const bcg_context* ctx); /* ctx may be NULL: last creation error */
int bcg_context_set_comm(bcg_context* ctx, const bcg_comm* comm);
int64_t bcg_local_volume(const bcg_context* ctx);
int bcg_local_dims(const bcg_context* ctx, int* dims4_out, int* origin4_out);
/* Pure host helper, needs no device: the message plan one halo exchange of this rank would pass to
 * bcg_comm.halo_exchange for `site_bytes` bytes per site (48*m for a field, 144 for links).  Arrays need
 * capacity 8 (2 messages per split direction).  Buffers hold, per split direction in ascending mu,
 * [minus face][plus face], each V_local/L_mu sites in lexicographic order of the other coordinates.
 * Message 2k sends the low face (x_mu = 0) to the minus neighbour and receives the plus ghost from the
 * plus neighbour; message 2k+1 sends the high face to the plus neighbour and receives the minus ghost.
 * Returns the number of messages, or a negative value for an invalid decomposition. */
int bcg_halo_plan(int ndim, const int* global_dims, const int* grid, const int* coords, size_t site_bytes, int* peer_send,
                  int* peer_recv, size_t* send_offset, size_t* recv_offset, size_t* nbytes, int64_t* ghost_sites);
/* Device halo buffers (valid after the first exchange was sized); for wrapping as communicator-visible
 * tensors. */
int bcg_halo_buffers(bcg_context* ctx, void** send, void** recv, size_t* bytes_each);
int bcg_synchronize(bcg_context* ctx);
/* The hipStream_t every call on this context enqueues on and its HIP device ordinal (for a transport that must order its
 * transfers with the library's kernels: include/blockcg_rccl.h). */
int bcg_context_stream(const bcg_context* ctx, void** stream_out, int* device_out);
/* Overlap tuning of the split halo exchange (bcg_comm.halo_exchange_begin/end).  interior_blocks: workgroups of the
 * stencil launch over the interior tiles, which runs while the exchange is in flight (default 512 = 2 per CU; a smaller
 * grid leaves compute units to the transport's kernels, at the price of a slower interior sweep; 0 keeps the current
 * value).  Results do not depend on it.  Also settable with BCG_HOP_BLOCKS_OVERLAP at context creation. */
int bcg_overlap_tuning(bcg_context* ctx, int interior_blocks);
/* Per-kernel timing with HIP events on the context's stream (off by default).  Names and
 * accumulated milliseconds / launch counts are returned as a JSON string owned by the context. */
int bcg_profiling(bcg_context* ctx, int enable);
const char* bcg_profile_json(bcg_context* ctx);
int bcg_profile_reset(bcg_context* ctx);
/* Force the generic (any-m, VALU) kernels even where an MFMA fast path exists; for parity tests. */
int bcg_force_generic(bcg_context* ctx, int enable);
/* Capacity mode.  dirac_op::op (inc/dirac_op.hpp:36-43) holds a whole field `tmp` = D P between its two stencil
 * applications.  With ring_slices = R > 0 the library keeps only R slices of `tmp` (slices of the last direction x3) and
 * runs the second application R-2 slices behind the first, so the solver's working set shrinks by (1 - R/L3) of a
 * field: what lets 128^4 sites x 16 right-hand sides x 4 shifts fit 8 x 288 GiB.  Results equal the default mode's up to
 * the summation order of the fused Gram product.  Requires a 4-D lattice whose direction 3 is not divided over ranks,
 * R >= 3 and R | L3 (else BCG_ERR_UNSUPPORTED / BCG_ERR_INVALID); applies to the widths of the specialised stencil
 * (8, 16, 32 with L0 a multiple of the tile length), other widths keep the whole `tmp`.  Halo exchanges are issued per
 * chunk of R-2 slices; when the bcg_comm offers the split form and R >= 4 they overlap the stencil work on the neighbouring
 * chunks (chunks of (R-2)/2 slices).  0 switches back. */
int bcg_capacity_mode(bcg_context* ctx, int ring_slices);
/* Device memory one SBCGrQ solve of width m with n_shifts shifts occupies on this rank in the current mode: X_s, P_s,
 * Q, T (+ the caller's B unless consume_B), tmp or its ring, links, halo buffers, scratch -- and, outside capacity mode
 * at m = 8 and 16, up to two further residual buffers and one spare field: the solver updates X_s, P_s of the
 * shifts s >= 1 (inc/block_solvers.hpp:161-181) up to four iterations at a time, which needs the residual block of
 * each deferred iteration (two at a time, as in capacity mode and at m = 32, need none: T doubles as the second buffer),
 * and X_0 (:145) with them, which needs the group's first P_0 kept in the spare field while P_0 moves on.  A single
 * system (n_shifts = 1) groups for X_0's sake alone, in threes or fours or not at all.  X_s, s >= 1, the residuals and
 * the iteration count are bit-identical to the ungrouped solver, X_0 agrees to rounding (1e-13); capacity mode (groups of
 * two) defers X_0 as well, in a form that needs no field; a solve that cannot
 * allocate the buffers runs with fewer; BCG_PAIR_SHIFTS=0 at context creation switches the grouping off, BCG_DEFER_X0=0
 * the deferred X_0 update.  Host arithmetic only. */
int bcg_sbcgrq_device_bytes(const bcg_context* ctx, int m, int n_shifts, int consume_B, size_t* bytes_out);
/* The same plan without a context or a device, for one rank of a process grid (a launcher sizing a run before it starts
 * its ranks): ring_slices = 0 for a whole `tmp`; ring_overlapped = the per-chunk exchanges overlap (split callbacks
 * present, ring >= 4); group_depth = iterations the shift updates are grouped over (1 = off, at most 4). */
int bcg_sbcgrq_plan_bytes(int ndim, const int* global_dims, const int* grid, int m, int n_shifts, int consume_B, int ring_slices,
                          int ring_overlapped, int group_depth, size_t* bytes_out);
/* ... and one HALF-VOLUME solve (bcg_field_create_half below: all work fields hold V/2 sites, links stay full-volume) */
int bcg_sbcgrq_device_bytes_half(const bcg_context* ctx, int m, int n_shifts, int consume_B, size_t* bytes_out);

/* ---- fields: block_fermion_field<N_rhs> (inc/fields.hpp:25-147) --------------------------- */
int bcg_field_create(bcg_context* ctx, int m, bcg_field** f);          /* explicit ctor :35 (contents undefined) */
int bcg_field_destroy(bcg_field* f);
int bcg_field_width(const bcg_field* f);
/* Half-volume (parity-compact) fields -- SURVEY.md section 8f-4 / Appendix D.  dirac_op::D couples nearest neighbours only
 * (inc/dirac_op.hpp:14-21), i.e. sites of opposite parity x_0+x_1+x_2+x_3 (mod 2), so dirac_op::op = mass^2 - D^2
 * (inc/dirac_op.hpp:36-43) never mixes the two parities and (op + sigma) X = B splits into two independent solves on V/2
 * sites each.  A half field holds the V/2 local sites of one parity (0 or 1) in the order of the full lattice; every
 * field primitive, dirac_apply and every solver accept half fields (all operands of a call of the SAME parity; the
 * solver's work fields are then half fields too: half the device memory of a full-volume solve, twice).
 * Even extents only (BCG_ERR_UNSUPPORTED otherwise).  Links stay full-volume.  On a lattice divided over ranks a half
 * field's ghost faces hold half the sites: the library posts the same message plan as for a full field (bcg_halo_plan) with
 * HALF the bytes per site, i.e. every offset and size halves; no capacity ring for half fields. */
int bcg_field_create_half(bcg_context* ctx, int m, int parity, bcg_field** f);
int bcg_field_parity(const bcg_field* f);   /* -1: full field; 0 / 1 */
int64_t bcg_field_sites(const bcg_field* f); /* sites held: V_local or V_local / 2 */
/* to_half != 0: half = the sites of its parity of `full`; else: those sites of `full` = half (the rest of `full` is kept) */
int bcg_field_parity_copy(bcg_field* full, bcg_field* half, int to_half);
/* The reference's fields live in host memory in the layout [site][rhs][colour] (inc/fields.hpp:19-20,28-30) and its
 * drivers read elements there (benchmark.cpp:61-63).  Both transfers convert the layout on the device and pipeline two
 * chunks (conversion kernel of one behind the bus transfer of the other).  Host memory the HIP runtime knows as pinned --
 * bcg_host_alloc below, hipHostMalloc, hipHostRegister -- is read / written by the DMA engine directly; pageable memory
 * goes through two pinned staging buffers that several host threads fill / drain. */
int bcg_field_upload(bcg_field* f, const double* host);                 /* host -> device */
int bcg_field_download(const bcg_field* f, double* host);               /* device -> host */
/* Pinned host memory for such mirrors (std::vector's allocator in the reference: Eigen::aligned_allocator,
 * inc/fields.hpp:28-30); no context needed. */
int bcg_host_alloc(size_t bytes, void** out);
int bcg_host_free(void* p);
/* operator[](int) :37-38 read access without moving the whole field: the tiles of n chosen LOCAL sites,
 * host[k] = f[sites[k]] in the host layout ([rhs][colour], 48*m bytes each).  Sites out of range: BCG_ERR_INVALID. */
int bcg_field_download_sites(const bcg_field* f, int64_t n, const int64_t* sites, double* host);
int bcg_field_copy(bcg_field* dst, const bcg_field* src);               /* copy-construct / assignment */
int bcg_field_set_zero(bcg_field* f);                                   /* setZero :57-61 */
/* i.i.d. uniform [-1,1) per real component from the counter-based generator shared with the
 * oracle (value depends on seed and GLOBAL element index only); stands in for setRandom :62-66 */
int bcg_field_fill_random(bcg_field* f, uint64_t seed);
int bcg_field_add_assign(bcg_field* y, const bcg_field* x);             /* operator+= :40-46 */
int bcg_field_sub_assign(bcg_field* y, const bcg_field* x);             /* operator-= :47-53  (K9) */
int bcg_field_add_scalar(bcg_field* y, const bcg_field* x, double a);   /* add(rhs, double) :70-77  (K3) */
int bcg_field_add_matrix(bcg_field* y, const bcg_field* x, const double* M); /* add(rhs, m x m) :70-77  (K5) */
/* rescale_add :79-90:  y = y*a + x*b  (K2) ;  y = y*M + x*b  (K6) */
int bcg_field_rescale_add_scalar(bcg_field* y, double a, const bcg_field* x, double b);
int bcg_field_rescale_add_matrix(bcg_field* y, const double* M, const bcg_field* x, double b);
/* hermitian_dot :103-122: out = a^dagger b (m x m, column-major), lower triangle mirrored so the
 * result is exactly Hermitian; summed over all ranks.  (K4) */
int bcg_field_hermitian_dot(const bcg_field* a, const bcg_field* b, double* out);
int bcg_field_real_dot(const bcg_field* a, const bcg_field* b, double* out); /* real_dot :93-99, m = 1 */
/* multiply_upper_triangular_inverse_RHS :125-136:  y <- y R^{-1}, R upper triangular  (K7) */
int bcg_field_tri_solve_rhs(bcg_field* y, const double* R);
/* thinQR :140-146:  R = chol(y^dagger y)^dagger ; y <- y R^{-1}.  BCG_ERR_NUMERIC if the Gram
 * matrix is not positive definite (the reference would silently produce NaN). */
int bcg_field_thin_qr(bcg_field* y, double* R_out);

/* ---- operator: dirac_op (inc/dirac_op.hpp:8-44) --------------------------------------------- */
int bcg_gauge_create(bcg_context* ctx, bcg_gauge** g);
int bcg_gauge_destroy(bcg_gauge* g);
int bcg_gauge_upload(bcg_gauge* g, const double* host);      /* local sites, [site][mu][3x3 col-major] */
int bcg_gauge_fill_random(bcg_gauge* g, uint64_t seed);      /* stands in for the ctor's setRandom :27-32 */
/* private D :14-21 (exposed for tests):  out = D in */
int bcg_dirac_hop(bcg_context* ctx, const bcg_gauge* g, bcg_field* out, const bcg_field* in);
/* op :36-43:  out = mass^2 in - D(D(in)).  The reference allocates a temporary per call (:39);
 * here the context owns one scratch field per width. */
/* out (parity p) = D in (parity 1 - p), half fields: the two off-diagonal blocks of dirac_op::D in the parity basis */
int bcg_dirac_hop_half(bcg_context* ctx, const bcg_gauge* g, bcg_field* out, const bcg_field* in);
int bcg_dirac_apply(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* out, const bcg_field* in);

/* ---- solver: SBCGrQ (inc/block_solvers.hpp:91-185) ------------------------------------------ */
typedef struct bcg_sbcgrq_trace {
  /* Optional per-iteration record for the first `capacity` iterations; arrays owned by caller.
   * mats: [capacity][3 + 2*n_shifts][m*m] complex column-major = alpha, rho, delta, alpha_s[], beta_s[]
   * res : [capacity][1 + n_shifts] doubles = residual, residual_shift[] (-1 = shift not visited) */
  int capacity;
  int recorded;
  double* mats;
  double* res;
} bcg_sbcgrq_trace;

/* X[n_shifts] must be fields of B's width; they are overwritten (zeroed first, :111-113).
 * sigma must be non-negative and ascending (:99-101) else BCG_ERR_INVALID.
 * eps = eps_shifts = 0 with a finite max_iterations is the fixed-work benchmark mode.
 * consume_B != 0 lets the solver use B's storage as its residual block Q (B is destroyed
 * logically; saves one field of memory at 128^4).  *iterations_out = operator applications (:184). */
int bcg_sbcgrq_solve(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* const* X, bcg_field* B, int n_shifts,
                     const double* sigma, double eps, double eps_shifts, int max_iterations, int consume_B,
                     int* iterations_out, double* residual_out, bcg_sbcgrq_trace* trace);

/* The same solver as a resumable state machine, so a caller can run (and time) an exact number of
 * iterations: begin = everything before the loop (:97-131); iterate = at most max_new_iterations
 * passes of the loop body (:132-182), stopping early when residual <= eps; end releases the work
 * fields.  X and B must outlive the state. */
typedef struct bcg_sbcgrq_state bcg_sbcgrq_state;
int bcg_sbcgrq_begin(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* const* X, bcg_field* B, int n_shifts,
                     const double* sigma, double eps, double eps_shifts, int consume_B, bcg_sbcgrq_state** state);
int bcg_sbcgrq_iterate(bcg_sbcgrq_state* state, int max_new_iterations, int* iterations_total, double* residual_out,
                       bcg_sbcgrq_trace* trace);
int bcg_sbcgrq_end(bcg_sbcgrq_state* state);

/* ---- the callers either side of the hot path, on the same kernels ---------- */
/* True relative residuals as the reference's tests and benchmark measure them (test/solvers.cpp:104-116,
 * benchmark.cpp:93-103): res_out[s*m + i] = |(A + sigma_s) X_s - B|_i / |B|_i. */
int bcg_true_residuals(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* const* X, const bcg_field* B,
                       int n_shifts, const double* sigma, double* res_out);
/* CG  src/standard_solvers.cpp:3-32   and   SCG  src/standard_solvers.cpp:34-95   (fields of width 1) */
int bcg_cg_solve(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* x, const bcg_field* b, double eps,
                 int max_iterations, int* iterations_out);
int bcg_scg_solve(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* const* x, const bcg_field* b, int n_shifts,
                  const double* sigma, double eps, double eps_shifts, int max_iterations, int* iterations_out);
/* BCG  inc/block_solvers.hpp:10-45   and   BCGrQ  inc/block_solvers.hpp:50-86 */
int bcg_bcg_solve(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* X, const bcg_field* B, double eps,
                  int max_iterations, int* iterations_out);
int bcg_bcgrq_solve(bcg_context* ctx, const bcg_gauge* g, double mass, bcg_field* X, const bcg_field* B, double eps,
                    int max_iterations, int* iterations_out);

/* Algorithmic HBM bytes of one SBCGrQ iteration on this rank's sub-lattice:
 *   V_local * [ (14 + 4*(S-1)) * 48*m + 2 * 144*ndim ] */
double bcg_sbcgrq_bytes_per_iteration(const bcg_context* ctx, int m, int n_shifts);

#ifdef __cplusplus
}
#endif
#endif /* BLOCKCG_HIP_H */
