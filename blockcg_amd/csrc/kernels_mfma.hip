// MFMA fast path for gfx950 (MI355X): fused SBCGrQ phase kernels for block widths m = 16 and 32
// (the stencil is in kernels_stencil.hip).  Wave = 64 lanes throughout; no other target is supported.
//
// A block field is a tall real matrix of 3V rows x 2m columns, (re,im) interleaved, rows contiguous.
// The tall-skinny products  out = in * C  (C complex m x m) are done by v_mfma_f64_16x16x4_f64 in
// the transposed form  out^T (2m x 16 rows) = Cr^T (2m x 2m) * in^T (2m x 16 rows):
//   - a wave owns a tile of 16 consecutive rows; lane l = (r = l&15, kq = l>>4) holds the complex
//     elements j = 4s + kq (s = 0..m/4-1) of row r  -> its registers ARE the MFMA B operands
//     (B[k = l>>4][n = l&15]) with no data movement, and the accumulator fragment
//     (D[(l>>4) + 4 reg][l&15]) lands in the same (row, column) ownership, so y += x*C, y = y*C + x
//     chain in registers;
//   - the coefficient matrix is the A operand, read per MFMA from LDS (one ds_read_b64 per lane).
//     Cr = [[Re C, Im C], [-Im C, Re C]] is never materialised: the lane reads Re or Im of C(j_in,
//     j_out) and the minus sign is the MFMA's NEG modifier on A.
// Block inner products (contraction over rows) need lane = (row = l>>4, column = l&15): the stencil
// kernel has that ownership natively (lane = (site, rhs)); phase B transposes its 16 x m tile
// through LDS.  Gram partials are reduced in a fixed order (deterministic).
//
// fp64 MFMA and fp64 VALU have the same peak on gfx950; MFMA is used so that the VALU and LDS stay
// free for addressing and the loads/stores, and the kernels remain HBM-bound.
#include "mfma_common.hpp"

namespace bcg {

namespace {

// ---------------------------------------------------------------------------------------------------
// Phase B:  Q -= T*alpha  (matrix passed as -alpha),  accumulate Q^dagger Q of the NEW Q.
// ---------------------------------------------------------------------------------------------------
template <int M>
// rinv != nullptr (deferred normalisation, see phase_B in capi_solvers.hip): the stored Q is the previous iteration's
// un-normalised block; it is multiplied by rinv = rho_prev^-1 first -- the arithmetic phase C used to do before storing it.
// Qout: where the new Q goes (== Q: in place; another buffer when the old block must survive, see k_phaseC_multi; that
// buffer may be T itself: every wave has read its tile of T when it writes that tile, and no other wave touches it).
__global__ void __launch_bounds__(256) k_phaseB(int64_t rows, const double2* Q, const double2* T,
                                                const double2* __restrict__ negalpha, double2* __restrict__ partials,
                                                GramFold gf, const double2* __restrict__ rinv, double2* Qout) {
  constexpr int NW = 4;
  constexpr int TLD = M * 2 + 2;  // doubles per transposition row (M*16 + 16 bytes)
  constexpr int JB = M / 16;
  constexpr int RED = NW * 8 * 64;
  constexpr int TRN = NW * 16 * TLD;
  constexpr int MDs = (MatLds<M>::DOUBLES + 1) & ~1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ml = smem;                                   // MatLds<M>::DOUBLES
  double* Mr = smem + MDs;                             // rinv, when given
  double* scratch = smem + MDs * (rinv ? 2 : 1);       // max(RED, TRN) doubles, 16-B aligned
  (void)RED; (void)TRN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_matrix<M>(Ml, negalpha, tid, 256);
  if (rinv) stage_matrix<M>(Mr, rinv, tid, 256);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  double* tw = scratch + wave * 16 * TLD;
  GramAcc<M> G;
  gram_zero(G);
  const int64_t ntiles = (rows + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * NW;
  int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave;
  // Two tuning builds of rounds 2-3 are gone with their switches (measured, profiles/r03_phaseB_ab.txt and
  // r03_contiguous_tile_moves.txt): the next tile's loads in flight during the products (7.3 ms in some processes, 8.1 ms
  // in others, against 7.16-7.25 ms in all of them without), and T / Q moved as contiguous memory through the wave's
  // ownership buffer (7.47-7.88 ms against 7.1-7.5).
  Tile<M> t, q;
  for (; tile < ntiles; tile += stride) {
    const int64_t row = tile * 16 + r;
    const bool ok = BCG_ROW_OK(row, rows);
    tile_load<M>(t, T, row, kq, ok);
    tile_load<M>(q, Q, row, kq, ok);
    if (rinv) {  // q <- q rho_prev^-1: exactly what phase C computes for its own use
      Acc<M> A0;
      acc_zero<M>(A0);
      rmul_acc<M>(A0, q, Mr, lane);
      tile_from_acc<M>(q, A0);
    }
    Acc<M> A;
    acc_from_tile<M>(A, q);
    rmul_acc<M>(A, t, Ml, lane);
    tile_from_acc<M>(q, A);
    tile_store<M>(q, Qout, row, kq, ok);
    // transpose the new tile through LDS: write (r, j = 4s+kq), read (row = 4g + (l>>4), j = l&15 + 16 jb)
#pragma unroll
    for (int s = 0; s < M / 4; ++s) *reinterpret_cast<double2*>(tw + r * TLD + 2 * (4 * s + kq)) = q.v[s];
    // same wave wrote and reads: no barrier needed, only LDS ordering (ds ops of one wave are in order)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double2 a[JB];
#pragma unroll
      for (int jb = 0; jb < JB; ++jb)
        a[jb] = *reinterpret_cast<const double2*>(tw + (4 * g + (lane >> 4)) * TLD + 2 * (16 * jb + (lane & 15)));
      gram_step<M>(G, a, a);
    }
  }
  gram_block_store<M, NW>(G, scratch, partials, tid, gf.out != nullptr);
  gram_fold<M * M>(gf, partials, tid, NW * 64);
}

// Phase B with its stores BATCHED (round 5; tools/microbench/rw_phased.hip, and k_phaseC_p0_batched below for the form): a
// block of 8 waves owns chunks of N consecutive tiles, wave w takes tiles w, w + 8, ... of the chunk with the next one's
// loads in flight, leaves each new tile in LDS in the field's own layout (rows padded by one element: the same buffer is
// the transposition source of the fused Gram product), and after a barrier the block writes the chunk, contiguous.  The
// first tiles of the block's next chunk are in flight during the stores.  In place (Qout == Q) and over T are safe: a block
// has read its whole chunk of both before it writes.  Same products as k_phaseB; the Gram partial sums run over other
// tiles per wave and block (a different, equally valid summation order).  rows must be a multiple of 16 N.
template <int M, int N>
__global__ void __launch_bounds__(512) k_phaseB_batched(int64_t rows, const double2* Q, const double2* T,
                                                        const double2* __restrict__ negalpha, double2* __restrict__ partials,
                                                        GramFold gf, const double2* __restrict__ rinv, double2* Qout) {
  static_assert(M == 16, "the batched phase B is instantiated for m = 16");
  constexpr int NW = 8;
  constexpr int RS = M + 1;  // double2 per staged row (= k_phaseB's transposition row, M * 2 + 2 doubles)
  constexpr int MDs = (MatLds<M>::DOUBLES + 1) & ~1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ml = smem;
  double* Mr = smem + MDs;
  double2* const stage = reinterpret_cast<double2*>(smem + 2 * MDs);  // N tiles of 16 rows x RS; the Gram reduction's buffer at the end
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_matrix<M>(Ml, negalpha, tid, NW * 64);
  if (rinv) stage_matrix<M>(Mr, rinv, tid, NW * 64);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  GramAcc<M> G;
  gram_zero(G);
  const int64_t nchunks = rows / (16 * N);
  constexpr int PER = N / NW;
  Tile<M> t, q;
  if (blockIdx.x < nchunks) {
    tile_load<M>(t, T, (static_cast<int64_t>(blockIdx.x) * N + wave) * 16 + r, kq, true);
    tile_load<M>(q, Q, (static_cast<int64_t>(blockIdx.x) * N + wave) * 16 + r, kq, true);
  }
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t t0 = c * N;
    const bool more = c + gridDim.x < nchunks;  // block-uniform
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = wave + NW * j;
      Tile<M> tn, qn;
      if (j + 1 < PER) {
        tile_load<M>(tn, T, (t0 + i + NW) * 16 + r, kq, true);
        tile_load<M>(qn, Q, (t0 + i + NW) * 16 + r, kq, true);
      } else if (more) {
        tile_load<M>(tn, T, ((c + gridDim.x) * N + wave) * 16 + r, kq, true);
        tile_load<M>(qn, Q, ((c + gridDim.x) * N + wave) * 16 + r, kq, true);
      }
      if (rinv) {
        Acc<M> A0;
        acc_zero<M>(A0);
        rmul_acc<M>(A0, q, Mr, lane);
        tile_from_acc<M>(q, A0);
      }
      Acc<M> A;
      acc_from_tile<M>(A, q);
      rmul_acc<M>(A, t, Ml, lane);
      tile_from_acc<M>(q, A);
      double2* const mine = stage + (i * 16) * RS;
#pragma unroll
      for (int s = 0; s < M / 4; ++s) mine[r * RS + 4 * s + kq] = q.v[s];
      // same wave wrote and reads (LDS operations of one wave are in order): rows 4g + (lane >> 4), column lane & 15
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double2 a = mine[(4 * g + (lane >> 4)) * RS + (lane & 15)];
        gram_step<M>(G, &a, &a);
      }
      if (j + 1 < PER || more) {
        t = tn;
        q = qn;
      }
    }
    __syncthreads();
    double2* const out = Qout + t0 * 16 * M;
    constexpr int ELEMS = N * 16 * M;
#pragma unroll 4
    for (int e = tid; e < ELEMS; e += NW * 64) {
      const double2 v = stage[(e / M) * RS + (e % M)];
      dv2 w;
      w.x = v.x;
      w.y = v.y;
      __builtin_nontemporal_store(w, reinterpret_cast<dv2*>(out + e));
    }
    __syncthreads();
  }
  gram_block_store<M, NW>(G, reinterpret_cast<double*>(stage), partials, tid, gf.out != nullptr);
  gram_fold<M * M>(gf, partials, tid, NW * 64);
}

// ---------------------------------------------------------------------------------------------------
// Phase C:  Q <- Q*Rinv ; for each active shift s:  X_s += P_s*A_s ;  P_s <- P_s*B_s + Q.
// mats: [Rinv, A_0, B_0, A_1, B_1, ...] complex column-major, consecutive in device memory.
// ---------------------------------------------------------------------------------------------------
struct ShiftPtrs {
  double2* X[8];
  double2* P[8];
};

// PREFETCH: load the next shift's P/X tiles while the current shift's MFMAs run (m = 16).  At m = 32 a tile is
// 32 VGPRs, the prefetch would push the kernel to one wave per SIMD, and one launch takes a single shift anyway.
// NW: waves per block.  At m = 32 a coefficient matrix is 16.6 KB of LDS; one block of 8 waves per CU (instead of two
// of 4) shares 9 matrices -- Rinv and four shifts -- so that 8 shifts are two launches and Q is read twice, not five times.
template <int M, bool PREFETCH, int NW = 4>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1)))
k_phaseC(int64_t rows, double2* __restrict__ Q, ShiftPtrs sp, int nshift,
                                                    const double2* __restrict__ mats, int apply_rinv) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int MD = (MatLds<M>::DOUBLES + 1) & ~1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // mats = [Rinv, A_0, B_0, A_1, B_1 ...]; Rinv takes an LDS slot only in the launch that applies it (at m = 32 a
  // matrix is 16.6 KB: without it a launch that does not apply Rinv fits two shifts at two blocks per CU)
  const int off = apply_rinv ? 0 : 1;
  const int nmat = 1 + 2 * nshift - off;
  for (int k = 0; k < nmat; ++k) stage_matrix<M>(smem + k * MD, mats + static_cast<int64_t>(k + off) * M * M, tid, NW * 64);
  __syncthreads();
  const double* const smat = smem - off * MD;  // slot of mats[i] = smat + i * MD
  const int r = lane & 15, kq = lane >> 4;
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave; tile < ntiles; tile += static_cast<int64_t>(gridDim.x) * NW) {
    const int64_t row = tile * 16 + r;
    const bool ok = BCG_ROW_OK(row, rows);
    auto load = [&](Tile<M>& t, const double2* f) __attribute__((always_inline)) { tile_load<M>(t, f, row, kq, ok); };
    auto store = [&](Tile<M>& t, double2* f) __attribute__((always_inline)) { tile_store<M>(t, f, row, kq, ok); };
    Tile<M> q;
    load(q, Q);
    Tile<M> p, x;
    if (nshift > 0) {
      load(p, sp.P[0]);
      load(x, sp.X[0]);
    }
    if (apply_rinv) {
      Acc<M> A;
      acc_zero<M>(A);
      rmul_acc<M>(A, q, smem, lane);
      tile_from_acc<M>(q, A);
      if (apply_rinv == 1) {  // 2: deferred normalisation -- Q stays un-normalised in memory, the next phase B applies rho^-1
        Tile<M> qs = q;
        store(qs, Q);
      }
    }
    for (int s = 0; s < nshift; ++s) {
      Tile<M> pn, xn;
      if (PREFETCH && s + 1 < nshift) {  // prefetch the next shift's tiles while this one computes
        load(pn, sp.P[s + 1]);
        load(xn, sp.X[s + 1]);
      }
      Acc<M> AX, AP;
      acc_from_tile<M>(AX, x);
      acc_from_tile<M>(AP, q);
      rmul_acc2<M>(AX, smat + (1 + 2 * s) * MD, AP, smat + (2 + 2 * s) * MD, p, lane);
      tile_from_acc<M>(x, AX);
      store(x, sp.X[s]);
      tile_from_acc<M>(p, AP);
      store(p, sp.P[s]);
      if (s + 1 < nshift) {
        if (PREFETCH) {
          p = pn;
          x = xn;
        } else {
          load(p, sp.P[s + 1]);
          load(x, sp.X[s + 1]);
        }
      }
    }
  }
}

// Phase C of shift 0 with the update of X_0 DEFERRED (SBCGrQ, DeferredX0 in capi_solvers.hip):  q = Q rinv (in registers, not
// stored);  Pout = P B + q.  X_0 is not touched -- the pass that closes the group of iterations adds the group's updates to
// it at once (k_phaseC_multi, XACC) -- so this pass moves three fields instead of five.  Pout may be P (in place: a wave
// reads its tile before it writes it) or another buffer (the first iteration of a group: the group's first P_0 must
// survive).  The product sequence for P is k_phaseC's (rmul_acc and rmul_acc2 run the same chain per accumulator): the
// P_0 the operator sees, and with it every coefficient of the iteration, is bit-identical to the undeferred form.
// mats: [rinv, B].
template <int M, bool AHEAD>
__global__ void __launch_bounds__(256) k_phaseC_p0(int64_t rows, const double2* __restrict__ Q, const double2* P, double2* Pout,
                                                   const double2* __restrict__ mats) {
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int MD = (MatLds<M>::DOUBLES + 1) & ~1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_matrix<M>(smem, mats, tid, NW * 64);
  stage_matrix<M>(smem + MD, mats + static_cast<int64_t>(M) * M, tid, NW * 64);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  const int64_t ntiles = (rows + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * NW;
  int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave;
  Tile<M> q, p;
  if (AHEAD && tile < ntiles) {  // the next tile's loads are in flight while this one is multiplied
    tile_load<M>(q, Q, tile * 16 + r, kq, tile * 16 + r < rows);
    tile_load<M>(p, P, tile * 16 + r, kq, tile * 16 + r < rows);
  }
  for (; tile < ntiles; tile += stride) {
    const int64_t row = tile * 16 + r;
    const bool ok = BCG_ROW_OK(row, rows);
    if (!AHEAD) {
      tile_load<M>(q, Q, row, kq, ok);
      tile_load<M>(p, P, row, kq, ok);
    }
    Tile<M> qn, pn;
    const int64_t nrow = (tile + stride) * 16 + r;
    if (AHEAD && tile + stride < ntiles) {
      tile_load<M>(qn, Q, nrow, kq, nrow < rows);
      tile_load<M>(pn, P, nrow, kq, nrow < rows);
    }
    Acc<M> A;
    acc_zero<M>(A);
    rmul_acc<M>(A, q, smem, lane);
    tile_from_acc<M>(q, A);
    Acc<M> AP;
    acc_from_tile<M>(AP, q);
    rmul_acc<M>(AP, p, smem + MD, lane);
    tile_from_acc<M>(p, AP);
    tile_store<M>(p, Pout, row, kq, ok);
    if (AHEAD && tile + stride < ntiles) {
      q = qn;
      p = pn;
    }
  }
}

// The same pass with its stores BATCHED (experiment of round 5, tools/microbench/rw_phased.hip: a streaming kernel with two
// arrays read and one written gains 10-15 % when every block reads a contiguous chunk, keeps the results in LDS and writes
// them out together).  A block of 8 waves owns chunks of N consecutive tiles: wave w multiplies tiles w, w + 8, ... of the
// chunk (the next one's loads in flight), leaves each result in LDS in the field's own layout (rows padded by one element
// against bank conflicts), and after a barrier the block writes the chunk -- N x 4 KB (m = 16), contiguous, 1 KB per wave
// instruction.  In place is safe: a block has read its whole chunk of P before it writes it.  Same products on the same
// values: bit-identical to k_phaseC_p0.  rows must be a multiple of 16 N.
template <int M, int N>
__global__ void __launch_bounds__(512) k_phaseC_p0_batched(int64_t rows, const double2* __restrict__ Q, const double2* P, double2* Pout,
                                                           const double2* __restrict__ mats) {
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int MD = (MatLds<M>::DOUBLES + 1) & ~1;
  constexpr int RS = M + 1;                       // row stride of the staged tiles, in double2
  double2* const stage = reinterpret_cast<double2*>(smem + 2 * MD);  // N tiles of 16 rows x RS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_matrix<M>(smem, mats, tid, NW * 64);
  stage_matrix<M>(smem + MD, mats + static_cast<int64_t>(M) * M, tid, NW * 64);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  const int64_t nchunks = rows / (16 * N);
  constexpr int PER = N / NW;
  Tile<M> q, p;
  if (blockIdx.x < nchunks) {
    tile_load<M>(q, Q, (static_cast<int64_t>(blockIdx.x) * N + wave) * 16 + r, kq, true);
    tile_load<M>(p, P, (static_cast<int64_t>(blockIdx.x) * N + wave) * 16 + r, kq, true);
  }
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t t0 = c * N;
    const bool more = c + gridDim.x < nchunks;  // block-uniform: the next chunk's first tiles are in flight during the stores
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = wave + NW * j;
      Tile<M> qn, pn;
      if (j + 1 < PER) {
        tile_load<M>(qn, Q, (t0 + i + NW) * 16 + r, kq, true);
        tile_load<M>(pn, P, (t0 + i + NW) * 16 + r, kq, true);
      } else if (more) {
        tile_load<M>(qn, Q, ((c + gridDim.x) * N + wave) * 16 + r, kq, true);
        tile_load<M>(pn, P, ((c + gridDim.x) * N + wave) * 16 + r, kq, true);
      }
      Acc<M> A;
      acc_zero<M>(A);
      rmul_acc<M>(A, q, smem, lane);
      tile_from_acc<M>(q, A);
      Acc<M> AP;
      acc_from_tile<M>(AP, q);
      rmul_acc<M>(AP, p, smem + MD, lane);
      tile_from_acc<M>(p, AP);
      double2* const dst = stage + (i * 16 + r) * RS + kq;
#pragma unroll
      for (int s = 0; s < M / 4; ++s) dst[4 * s] = p.v[s];
      if (j + 1 < PER || more) {
        q = qn;
        p = pn;
      }
    }
    __syncthreads();
    // the chunk, contiguous: element e of the chunk = row e / M (of N x 16), column e % M
    double2* const out = Pout + t0 * 16 * M;
    constexpr int ELEMS = N * 16 * M;
#pragma unroll 4
    for (int e = tid; e < ELEMS; e += NW * 64) {
      const double2 v = stage[(e / M) * RS + (e % M)];
      dv2 w;
      w.x = v.x;
      w.y = v.y;
      __builtin_nontemporal_store(w, reinterpret_cast<dv2*>(out + e));
    }
    __syncthreads();
  }
}

// Phase C of NS = 2 .. 4 consecutive iterations in one pass over the fields of the shifted systems (m = 8, 16).
// Only P_0 feeds the operator, so the updates of the shifts s >= 1 of an iteration can wait for a later one as long as
// that iteration's un-normalised residual block is kept (phase B of the following iteration then writes its result to
// another buffer instead of in place).  Step j = 0 .. NS-1 stands for the j-th of the iterations, the last being the
// current one:
//   q_j = Q_j rinv_j                                   (not stored: deferred normalisation)
//   entry e, for its steps j = first[e] .. last[e]-1:   X_e += P_e A_ej ;  P_e <- P_e B_ej + q_j
// with the intermediate X, P in registers.  Shift 0 was updated in every iteration and takes the last step only; a shift
// that left the active set on the way takes the steps before that.  Every product is the instruction sequence of
// k_phaseC on the same fp64 values (a store and a load between two steps would not change them), so the fields are
// bit-identical to NS k_phaseC passes; X_s and P_s are read and written once instead of NS times.
// mats: [rinv_0 .. rinv_{NS-1}, then per entry and per step of it A, B], consecutive.
struct MultiQ {
  const double2* q[4];
};
struct MultiSteps {
  int first[8], last[8];
  // XACC (deferred X_0, entry 0 only): before its steps, X_0 += P1 C_0 + q_0 C_1 + ... + q_{xacc-2} C_{xacc-1} -- the X_0 updates of
  // the group's earlier iterations, composed on the host onto the group's first P_0 (`p1`) and the normalised residual
  // blocks the kernel holds anyway.  xacc = number of those matrices (0: off); they follow entry 0's step matrices.
  // p1 == nullptr (the spare-less form of a group of two, DeferredX0 in capi_solvers.hip): no P1 term, C_0 is not read.
  int xacc;
  const double2* p1;
};
// Blocks: one per CU whatever the matrices take of its LDS.  m = 16: 8 waves, 2 per SIMD -- NS residual tiles, the
// entry's and the next entry's P and X tiles, two accumulators and the LDS operands in flight are 180 .. 250 registers
// (12 waves at 168 registers spilled and were 1-2 % slower at NS = 2: 56.7 against 55.5-56.1 ms per iteration; with the
// unpredicated body below NS = 2 needs 152 and 12 waves fit, but run the same: 39.8 vs 40.0 ms at the 64^3 x 128 share).  With the
// fields read once per NS iterations the kernel is bound by the fp64 matrix pipe as much as by HBM (NS = 4, 4 shifts:
// 54 TFLOP/s of the 78.6 this chip issues, tools/microbench/mfma_f64_rate.hip, next to 4.5 TB/s).
// NORM = false (m = 32, where the residual block is stored normalised): the Q_j are used as they are and mats holds no
// rinv_j.  PRE = false: no prefetch of the next entry's tiles (m = 32: a tile is 32 registers, and an entry's two steps
// are 128 MFMAs per load).
template <int M, int NW, int NS, bool NORM = true, bool PRE = true>
__global__ void __launch_bounds__(NW * 64)
k_phaseC_multi(int64_t rows, MultiQ qs, ShiftPtrs sp, int nent, MultiSteps steps, int nmat, const double2* __restrict__ mats) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int MD = (MatLds<M>::DOUBLES + 1) & ~1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = 0; k < nmat; ++k) stage_matrix<M>(smem + k * MD, mats + static_cast<int64_t>(k) * M * M, tid, NW * 64);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  const int64_t ntiles = (rows + 15) / 16;
  // One body, instantiated for full tiles (no predication on its loads and stores: straight-line code, 2.5 % faster at
  // 64^4) and for the field's last tile when the row count is no multiple of 16.
  auto body = [&](int64_t tile, auto full) __attribute__((always_inline)) {
    const int64_t row = tile * 16 + r;
    const bool ok = decltype(full)::value || BCG_ROW_OK(row, rows);
    Tile<M> q[NS], p, x;
#pragma unroll
    for (int j = 0; j < NS; ++j) tile_load<M>(q[j], qs.q[j], row, kq, ok);
    if (nent > 0) {
      tile_load<M>(p, sp.P[0], row, kq, ok);
      tile_load<M>(x, sp.X[0], row, kq, ok);
    }
    Tile<M> p1;  // XACC: the group's first P_0, in flight while the residual blocks are normalised
    if (NORM && steps.xacc > 0 && steps.p1 != nullptr) tile_load<M>(p1, steps.p1, row, kq, ok);
    if (NORM) {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        Acc<M> A;
        acc_zero<M>(A);
        rmul_acc<M>(A, q[j], smem + j * MD, lane);
        tile_from_acc<M>(q[j], A);
      }
    }
    const double* mat = smem + (NORM ? NS : 0) * MD;
    if (NORM && steps.xacc > 0) {  // wave-uniform: the deferred X_0 updates of the group's earlier iterations
      const double* cm = mat + 2 * (steps.last[0] - steps.first[0]) * MD;
      Acc<M> AX;
      acc_from_tile<M>(AX, x);
      if (steps.p1 != nullptr) rmul_acc<M>(AX, p1, cm, lane);
#pragma unroll
      for (int j = 0; j + 1 < NS; ++j)
        if (j + 1 < steps.xacc) rmul_acc<M>(AX, q[j], cm + (j + 1) * MD, lane);
      tile_from_acc<M>(x, AX);
    }
    for (int e = 0; e < nent; ++e) {
      Tile<M> pn, xn;
      if (PRE && e + 1 < nent) {  // the next entry's tiles are in flight while this one is multiplied
        tile_load<M>(pn, sp.P[e + 1], row, kq, ok);
        tile_load<M>(xn, sp.X[e + 1], row, kq, ok);
      }
      const int first = steps.first[e], last = steps.last[e];  // wave-uniform
      if (NORM && e == 1 && steps.xacc > 0) mat += steps.xacc * MD;  // entry 0's composed matrices sit behind its step matrices
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j >= first && j < last) {
          Acc<M> AX, AP;
          acc_from_tile<M>(AX, x);
          acc_from_tile<M>(AP, q[j]);
          rmul_acc2<M>(AX, mat, AP, mat + MD, p, lane);
          tile_from_acc<M>(x, AX);
          tile_from_acc<M>(p, AP);
          mat += 2 * MD;
        }
      }
      tile_store<M>(x, sp.X[e], row, kq, ok);
      tile_store<M>(p, sp.P[e], row, kq, ok);
      if (e + 1 < nent) {
        if (PRE) {
          p = pn;
          x = xn;
        } else {
          tile_load<M>(p, sp.P[e + 1], row, kq, ok);
          tile_load<M>(x, sp.X[e + 1], row, kq, ok);
        }
      }
    }
  };
  // (one contiguous run of tiles per block instead of this interleaved map: no gain at depth 2 or 4, profiles/r05_batched_stores.txt)
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave; tile < ntiles; tile += static_cast<int64_t>(gridDim.x) * NW) {
    if ((tile + 1) * 16 <= rows) body(tile, std::true_type{});
    else body(tile, std::false_type{});
  }
}

// ---------------------------------------------------------------------------------------------------
// Stand-alone right-multiplications (K5, K6) on the MFMA path, for the field-level API.
// ---------------------------------------------------------------------------------------------------
template <int M, int MODE>
__global__ void __launch_bounds__(256) k_rmul_mfma(int64_t rows, double2* __restrict__ y, const double2* __restrict__ x,
                                                   const double2* __restrict__ Cg, double b) {
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_matrix<M>(smem, Cg, tid, 256);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave; tile < ntiles; tile += static_cast<int64_t>(gridDim.x) * NW) {
    const int64_t row = tile * 16 + r;
    const bool ok = BCG_ROW_OK(row, rows);
    Tile<M> ty, tx;
    tile_load<M>(ty, y, row, kq, ok);
    Acc<M> A;
    if (MODE == RMUL_ADD) {          // y += x*C
      tile_load<M>(tx, x, row, kq, ok);
      acc_from_tile<M>(A, ty);
      rmul_acc<M>(A, tx, smem, lane);
    } else if (MODE == RMUL_XPAY) {  // y = y*C + b*x
      tile_load<M>(tx, x, row, kq, ok);
#pragma unroll
      for (int s = 0; s < M / 4; ++s) tx.v[s] = make_double2(b * tx.v[s].x, b * tx.v[s].y);
      acc_from_tile<M>(A, tx);
      rmul_acc<M>(A, ty, smem, lane);
    } else {                         // y = y*C
      acc_zero<M>(A);
      rmul_acc<M>(A, ty, smem, lane);
    }
    tile_from_acc<M>(ty, A);
    tile_store<M>(ty, y, row, kq, ok);
  }
}

// ... with the stores batched per chunk of N tiles through LDS (m = 16; the form of k_phaseC_p0_batched: 8 waves, the next
// tile's loads in flight, the first tiles of the block's next chunk in flight during the stores).  y is updated in place: a
// block has read its whole chunk before it writes it.  Same products: bit-identical to k_rmul_mfma.  rows % (16 N) == 0.
template <int M, int MODE, int N>
__global__ void __launch_bounds__(512) k_rmul_mfma_batched(int64_t rows, double2* y, const double2* __restrict__ x,
                                                           const double2* __restrict__ Cg, double b) {
  constexpr int NW = 8;
  constexpr int MD = (MatLds<M>::DOUBLES + 1) & ~1;
  constexpr int RS = M + 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double2* const stage = reinterpret_cast<double2*>(smem + MD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_matrix<M>(smem, Cg, tid, NW * 64);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  const int64_t nchunks = rows / (16 * N);
  constexpr int PER = N / NW;
  Tile<M> ty, tx;
  if (blockIdx.x < nchunks) {
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * N + wave) * 16 + r;
    tile_load<M>(ty, y, row, kq, true);
    if (MODE != RMUL_MUL) tile_load<M>(tx, x, row, kq, true);
  }
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t t0 = c * N;
    const bool more = c + gridDim.x < nchunks;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = wave + NW * j;
      Tile<M> tyn, txn;
      if (j + 1 < PER || more) {
        const int64_t nrow = (j + 1 < PER ? t0 + i + NW : (c + gridDim.x) * N + wave) * 16 + r;
        tile_load<M>(tyn, y, nrow, kq, true);
        if (MODE != RMUL_MUL) tile_load<M>(txn, x, nrow, kq, true);
      }
      Acc<M> A;
      if (MODE == RMUL_ADD) {
        acc_from_tile<M>(A, ty);
        rmul_acc<M>(A, tx, smem, lane);
      } else if (MODE == RMUL_XPAY) {
#pragma unroll
        for (int s = 0; s < M / 4; ++s) tx.v[s] = make_double2(b * tx.v[s].x, b * tx.v[s].y);
        acc_from_tile<M>(A, tx);
        rmul_acc<M>(A, ty, smem, lane);
      } else {
        acc_zero<M>(A);
        rmul_acc<M>(A, ty, smem, lane);
      }
      tile_from_acc<M>(ty, A);
      double2* const dst = stage + (i * 16 + r) * RS + kq;
#pragma unroll
      for (int s = 0; s < M / 4; ++s) dst[4 * s] = ty.v[s];
      if (j + 1 < PER || more) {
        ty = tyn;
        if (MODE != RMUL_MUL) tx = txn;
      }
    }
    __syncthreads();
    double2* const out = y + t0 * 16 * M;
    constexpr int ELEMS = N * 16 * M;
#pragma unroll 4
    for (int e = tid; e < ELEMS; e += NW * 64) {
      const double2 v = stage[(e / M) * RS + (e % M)];
      dv2 w;
      w.x = v.x;
      w.y = v.y;
      __builtin_nontemporal_store(w, reinterpret_cast<dv2*>(out + e));
    }
    __syncthreads();
  }
}

// Stand-alone Gram product a^dagger b: coalesced loads already have (row = l>>4, col = l&15) ownership.
template <int M>
__global__ void __launch_bounds__(256) k_gram_mfma(int64_t rows, const double2* __restrict__ a, const double2* __restrict__ b,
                                                   double2* __restrict__ partials) {
  constexpr int NW = 4;
  constexpr int JB = M / 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  GramAcc<M> G;
  gram_zero(G);
  const int64_t nquads = (rows + 3) / 4;  // 4 rows per MFMA step
  for (int64_t qd = static_cast<int64_t>(blockIdx.x) * NW + wave; qd < nquads; qd += static_cast<int64_t>(gridDim.x) * NW) {
    const int64_t row = qd * 4 + (lane >> 4);
    const bool ok = BCG_ROW_OK(row, rows);
    double2 av[JB], bv[JB];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      av[jb] = ok ? a[row * M + 16 * jb + (lane & 15)] : make_double2(0.0, 0.0);
      bv[jb] = ok ? b[row * M + 16 * jb + (lane & 15)] : make_double2(0.0, 0.0);
    }
    gram_step<M>(G, av, bv);
  }
  gram_block_store<M, NW>(G, smem, partials, tid);
}

// Gram product at m = 8 (see gram_block_store_fold8): 8 rows of 8 columns per MFMA step, 16 B per lane, 1 KB per wave.
__global__ void __launch_bounds__(256) k_gram_mfma8(int64_t rows, const double2* __restrict__ a, const double2* __restrict__ b,
                                                    double2* __restrict__ partials) {
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  GramAcc<16> G;
  gram_zero(G);
  const int64_t nsteps = (rows + 7) / 8;
  for (int64_t st = static_cast<int64_t>(blockIdx.x) * NW + wave; st < nsteps; st += static_cast<int64_t>(gridDim.x) * NW) {
    const int64_t row = st * 8 + (lane >> 3);
    const bool ok = BCG_ROW_OK(row, rows);
    const double2 av = ok ? a[row * 8 + (lane & 7)] : make_double2(0.0, 0.0);
    const double2 bv = ok ? b[row * 8 + (lane & 7)] : make_double2(0.0, 0.0);
    gram_step<16>(G, &av, &bv);
  }
  gram_block_store_fold8<NW>(G, smem, partials, tid);
}

// Phase B at m = 8: Q += T * negalpha through the m = 8 product tile, then the new 16 x 8 tile is re-read from a per-wave
// LDS buffer in (row, column) ownership for the folded Gram product.
__global__ void __launch_bounds__(256) k_phaseB8(GramFold gf, int64_t rows, const double2* Q, const double2* T,
                                                 const double2* __restrict__ negalpha, double2* __restrict__ partials,
                                                 const double2* __restrict__ rinv, double2* Qout) {
  constexpr int M = 8, NW = 4;
  constexpr int TLD = M * 2 + 2;  // doubles per transposition row
  constexpr int MDs = (MatLds<M>::DOUBLES + 1) & ~1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ml = smem;
  double* Mr = smem + MDs;                        // rinv, when given (see k_phaseB)
  double* scratch = smem + MDs * (rinv ? 2 : 1);  // max(NW*16*TLD, NW*8*64) doubles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_matrix<M>(Ml, negalpha, tid, 256);
  if (rinv) stage_matrix<M>(Mr, rinv, tid, 256);
  __syncthreads();
  const int r = lane & 15, kq = lane >> 4;
  double* tw = scratch + wave * 16 * TLD;
  GramAcc<16> G;
  gram_zero(G);
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave; tile < ntiles; tile += static_cast<int64_t>(gridDim.x) * NW) {
    const int64_t row = tile * 16 + r;
    const bool ok = BCG_ROW_OK(row, rows);
    Tile<M> t, q;
    tile_load<M>(t, T, row, kq, ok);
    tile_load<M>(q, Q, row, kq, ok);
    if (rinv) {
      Acc<M> A0;
      acc_zero<M>(A0);
      rmul_acc<M>(A0, q, Mr, lane);
      tile_from_acc<M>(q, A0);
    }
    Acc<M> A;
    acc_from_tile<M>(A, q);
    rmul_acc<M>(A, t, Ml, lane);
    tile_from_acc<M>(q, A);
    tile_store<M>(q, Qout, row, kq, ok);
#pragma unroll
    for (int sx = 0; sx < M / 4; ++sx) *reinterpret_cast<double2*>(tw + r * TLD + 2 * (4 * sx + kq)) = ok ? q.v[sx] : make_double2(0.0, 0.0);
    // same wave wrote and reads (LDS operations of one wave are in order): rows 8g .. 8g+7, one per 8 lanes
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const double2 av = *reinterpret_cast<const double2*>(tw + (8 * g + (lane >> 3)) * TLD + 2 * (lane & 7));
      gram_step<16>(G, &av, &av);
    }
  }
  gram_block_store_fold8<NW>(G, scratch, partials, tid, gf.out != nullptr);
  gram_fold<64>(gf, partials, tid, NW * 64);
}

}  // namespace

// The streaming row kernels at m = 16 (phase B, k_phaseC_p0) batch their stores per chunk of tiles through LDS (k_phaseB_batched).
// BCG_ROW_BATCHED=0 keeps the plain kernels: the A/B of record.
static bool row_batched() {  // (read per launch, not cached: the tests switch it inside one process)
  const char* e = std::getenv("BCG_ROW_BATCHED");
  return e ? std::atoi(e) != 0 : true;
}

bool mfma_width(int m) { return m == 8 || m == 16 || m == 32; }  // declared in kernels.hpp
bool mfma_rows_width(int m) { return m == 8 || m == 16 || m == 32; }  // right-multiply kernels (phase C, K5, K6)
int phaseC_max_shifts(int m, bool applies_rinv) {
  (void)applies_rinv;  // m = 32: 9 matrices (Rinv + 4 shifts) are 149.8 KB of the 160 KB of a CU, one 8-wave block each
  return (m == 16 || m == 8) ? 8 : (m == 32 ? 4 : 0);
}

int launch_phaseB(hipStream_t s, int m, int64_t rows, double2* Q, const double2* T, const double2* negalpha,
                  double2* partials, int max_blocks, GramFold gf, const double2* rinv, double2* Qout) {
  if (!Qout) Qout = Q;
  const int grid = grid_tiles((rows + 15) / 16, 4, max_blocks);
  if (m == 8) {
    const size_t lds = sizeof(double) * (((MatLds<8>::DOUBLES + 1) & ~1) * (rinv ? 2 : 1) + 4 * 8 * 64);  // RED 2048 >= TRN 4*16*18
    hipLaunchKernelGGL(k_phaseB8, dim3(grid), dim3(256), lds, s, gf, rows, Q, T, negalpha, partials, rinv, Qout);
  } else if (m == 16) {
    constexpr int M = 16;
    // stores batched per chunk of 32 tiles: 6.68 against 7.42 ms at 64^4 (profiles/r05_batched_stores.txt); BCG_ROW_BATCHED=0: the plain kernel
    if (row_batched() && rows % (16 * 32) == 0 && max_blocks >= 8) {
      constexpr int N = 32;
      const size_t ldsb = sizeof(double) * ((MatLds<M>::DOUBLES + 1) & ~1) * 2 + sizeof(double2) * N * 16 * (M + 1);
      const int grid8 = static_cast<int>(std::min<int64_t>(rows / (16 * N), std::min(256, max_blocks)));
      allow_lds(k_phaseB_batched<M, N>, ldsb);
      hipLaunchKernelGGL((k_phaseB_batched<M, N>), dim3(grid8), dim3(512), ldsb, s, rows, Q, T, negalpha, partials, gf, rinv, Qout);
      return grid8;
    }
    const size_t lds = sizeof(double) * (((MatLds<M>::DOUBLES + 1) & ~1) * (rinv ? 2 : 1) + 4 * 16 * (2 * M + 2));  // TRN 2176 >= RED 2048
    hipLaunchKernelGGL((k_phaseB<M>), dim3(grid), dim3(256), lds, s, rows, Q, T, negalpha, partials, gf, rinv, Qout);
  } else {
    constexpr int M = 32;
    const size_t lds = sizeof(double) * (((MatLds<M>::DOUBLES + 1) & ~1) * (rinv ? 2 : 1) + 4 * 16 * 66);  // TRN 4224 >= RED 2048: 50 KB, 3 blocks per CU
    allow_lds(k_phaseB<M>, lds);
    hipLaunchKernelGGL((k_phaseB<M>), dim3(grid), dim3(256), lds, s, rows, Q, T, negalpha, partials, gf, rinv, Qout);
  }
  return grid;
}

void launch_phaseC(hipStream_t s, int m, int64_t rows, double2* Q, double2* const* X, double2* const* P, int nshift,
                   const double2* mats, int apply_rinv, int max_blocks) {
  ShiftPtrs sp{};
  for (int k = 0; k < nshift && k < 8; ++k) {
    sp.X[k] = X[k];
    sp.P[k] = P[k];
  }
  const int grid = grid_tiles((rows + 15) / 16, 4, max_blocks);
  const int nmat = 1 + 2 * nshift - (apply_rinv ? 0 : 1);
  if (m == 8) {
    constexpr int M = 8;
    const size_t lds = sizeof(double) * ((MatLds<M>::DOUBLES + 1) & ~1) * nmat;
    hipLaunchKernelGGL((k_phaseC<M, true>), dim3(grid), dim3(256), lds, s, rows, Q, sp, nshift, mats, apply_rinv);
  } else if (m == 16) {
    constexpr int M = 16;
    // (contiguous tile moves through a per-wave LDS buffer -- round 3's BCG_PHASEC_LIN -- and through the matrix pipe -- round 5,
    //  profiles/r05_phaseC_p0.txt -- measured no gain in the row kernels and are gone: the access shape does not bound them)
    const size_t lds = sizeof(double) * ((MatLds<M>::DOUBLES + 1) & ~1) * nmat;
    allow_lds(k_phaseC<M, true>, lds);
    hipLaunchKernelGGL((k_phaseC<M, true>), dim3(grid), dim3(256), lds, s, rows, Q, sp, nshift, mats, apply_rinv);
  } else {
    constexpr int M = 32;
    const size_t lds = sizeof(double) * ((MatLds<M>::DOUBLES + 1) & ~1) * nmat;
    if (nmat > 4) {  // more matrices than two 4-wave blocks per CU can hold: one 8-wave block per CU
      const int grid8 = grid_tiles((rows + 15) / 16, 8, max_blocks / 4 > 0 ? max_blocks / 4 : 1);  // one resident block per CU
      allow_lds(k_phaseC<M, true, 8>, lds);  // 2 waves per SIMD whatever the registers (LDS-bound): room for the tile prefetch
      hipLaunchKernelGGL((k_phaseC<M, true, 8>), dim3(grid8), dim3(512), lds, s, rows, Q, sp, nshift, mats, apply_rinv);
    } else {
      allow_lds(k_phaseC<M, false>, lds);
      hipLaunchKernelGGL((k_phaseC<M, false>), dim3(grid), dim3(256), lds, s, rows, Q, sp, nshift, mats, apply_rinv);
    }
  }
}

void launch_phaseC_p0(hipStream_t s, int m, int64_t rows, const double2* Q, const double2* P, double2* Pout, const double2* mats,
                      int max_blocks) {
  // profiles/r05_phaseC_p0.txt (64^4, m = 16): the next tile's loads in flight during the products and two blocks per CU,
  // 7.26 ms per launch; without the prefetch 7.56-7.63 ms at 1024, 1536 or 2048 blocks, with it at 1024 blocks 7.84
  // stores batched per chunk of 32 tiles, one 8-wave block per CU: 6.73 against 7.25 ms at 64^4 (chunks of 16, two blocks
  // per CU: no gain; profiles/r05_batched_stores.txt); BCG_ROW_BATCHED=0: the plain kernel
  // (m = 8 at 32^4, config 1: the batched form is slower, 0.179 against 0.169 ms -- profiles/r05_batched_stores.txt)
  if (m == 16 && row_batched() && rows % (16 * 32) == 0 && max_blocks >= 8) {
    constexpr int M = 16, N = 32;
    const size_t lds = sizeof(double) * ((MatLds<M>::DOUBLES + 1) & ~1) * 2 + sizeof(double2) * N * 16 * (M + 1);
    const int grid8 = static_cast<int>(std::min<int64_t>(rows / (16 * N), std::min(256, max_blocks)));
    allow_lds(k_phaseC_p0_batched<M, N>, lds);
    hipLaunchKernelGGL((k_phaseC_p0_batched<M, N>), dim3(grid8), dim3(512), lds, s, rows, Q, P, Pout, mats);
    return;
  }
  const int grid = grid_tiles((rows + 15) / 16, 4, max_blocks < 512 ? max_blocks : 512);
  if (m == 8) {
    const size_t lds = sizeof(double) * ((MatLds<8>::DOUBLES + 1) & ~1) * 2;
    hipLaunchKernelGGL((k_phaseC_p0<8, true>), dim3(grid), dim3(256), lds, s, rows, Q, P, Pout, mats);
  } else {
    const size_t lds = sizeof(double) * ((MatLds<16>::DOUBLES + 1) & ~1) * 2;
    hipLaunchKernelGGL((k_phaseC_p0<16, true>), dim3(grid), dim3(256), lds, s, rows, Q, P, Pout, mats);
  }
}

int phaseC_multi_matrices(int nsteps, int nent, const int* first, const int* last, bool normalise) {
  int n = normalise ? nsteps : 0;
  for (int e = 0; e < nent; ++e) n += 2 * (last[e] - first[e]);
  return n;
}
static size_t mat_lds_bytes(int m) {
  return sizeof(double) * (m == 8 ? ((MatLds<8>::DOUBLES + 1) & ~1) : m == 16 ? ((MatLds<16>::DOUBLES + 1) & ~1) : ((MatLds<32>::DOUBLES + 1) & ~1));
}
bool phaseC_multi_fits(int m, int nsteps, int n_shifts) {
  if (nsteps < 2 || nsteps > 4 || n_shifts < 1 || n_shifts > 8) return false;
  if (m == 32) return nsteps == 2;  // un-normalised blocks are not kept at m = 32
  return m == 8 || m == 16;         // more shifts than one launch has LDS room for: several launches (phaseC_multi_max_entries)
}
int phaseC_multi_capacity(int m) { return static_cast<int>(150 * 1024 / mat_lds_bytes(m)); }

void launch_phaseC_multi(hipStream_t s, int m, int64_t rows, int nsteps, const double2* const* Q, double2* const* X,
                         double2* const* P, int nent, const int* first, const int* last, const double2* mats, int max_blocks,
                         bool normalise, int xacc, const double2* p1) {
  ShiftPtrs sp{};
  MultiSteps st{};
  MultiQ qs{};
  for (int k = 0; k < nent && k < 8; ++k) {
    sp.X[k] = X[k];
    sp.P[k] = P[k];
    st.first[k] = first[k];
    st.last[k] = last[k];
  }
  for (int j = 0; j < 4; ++j) qs.q[j] = Q[j < nsteps ? j : nsteps - 1];
  st.xacc = (normalise && nent > 0) ? xacc : 0;
  st.p1 = p1;
  const int nmat = phaseC_multi_matrices(nsteps, nent, first, last, normalise) + st.xacc;
  const int cus = max_blocks / 4 > 0 ? max_blocks / 4 : 1;  // one block per CU
#define BCG_MULTI(MM, NW, NS, NORM, PRE)                                                                           \
  {                                                                                                                \
    const size_t lds = mat_lds_bytes(MM) * nmat;                                                                   \
    const int grid = grid_tiles((rows + 15) / 16, NW, cus);                                                        \
    allow_lds(k_phaseC_multi<MM, NW, NS, NORM, PRE>, lds);                                                         \
    hipLaunchKernelGGL((k_phaseC_multi<MM, NW, NS, NORM, PRE>), dim3(grid), dim3(NW * 64), lds, s, rows, qs, sp, nent, st, nmat, mats);  \
  }
  if (m == 8) {
    if (nsteps == 2) BCG_MULTI(8, 12, 2, true, true) else if (nsteps == 3) BCG_MULTI(8, 12, 3, true, true) else BCG_MULTI(8, 12, 4, true, true)
  } else if (m == 16) {
    if (nsteps == 2) BCG_MULTI(16, 8, 2, true, true) else if (nsteps == 3) BCG_MULTI(16, 8, 3, true, true) else BCG_MULTI(16, 8, 4, true, true)
  } else {
    BCG_MULTI(32, 8, 2, false, false)
  }
#undef BCG_MULTI
}

void launch_rmul_mfma(hipStream_t s, int m, int64_t rows, double2* y, const double2* x, const double2* Cd, double b,
                      RmulMode mode, int max_blocks) {
  const int grid = grid_tiles((rows + 15) / 16, 4, max_blocks);
#define BCG_RMUL(MM)                                                                                             \
  {                                                                                                              \
    constexpr int M = MM;                                                                                        \
    const size_t lds = sizeof(double) * ((MatLds<M>::DOUBLES + 1) & ~1);                                         \
    if (mode == RMUL_ADD) hipLaunchKernelGGL((k_rmul_mfma<M, RMUL_ADD>), dim3(grid), dim3(256), lds, s, rows, y, x, Cd, b);       \
    else if (mode == RMUL_XPAY) hipLaunchKernelGGL((k_rmul_mfma<M, RMUL_XPAY>), dim3(grid), dim3(256), lds, s, rows, y, x, Cd, b); \
    else hipLaunchKernelGGL((k_rmul_mfma<M, RMUL_MUL>), dim3(grid), dim3(256), lds, s, rows, y, x, Cd, b);       \
  }
  // m = 16: stores batched per chunk of 32 tiles (k_phaseC_p0_batched's form; BCG_ROW_BATCHED=0: the plain kernel)
  if (m == 16 && row_batched() && rows % (16 * 32) == 0 && max_blocks >= 8) {
    constexpr int M = 16, N = 32;
    const size_t ldsb = sizeof(double) * ((MatLds<M>::DOUBLES + 1) & ~1) + sizeof(double2) * N * 16 * (M + 1);
    const int grid8 = static_cast<int>(std::min<int64_t>(rows / (16 * N), std::min(256, max_blocks)));
    if (mode == RMUL_ADD) {
      allow_lds(k_rmul_mfma_batched<M, RMUL_ADD, N>, ldsb);
      hipLaunchKernelGGL((k_rmul_mfma_batched<M, RMUL_ADD, N>), dim3(grid8), dim3(512), ldsb, s, rows, y, x, Cd, b);
    } else if (mode == RMUL_XPAY) {
      allow_lds(k_rmul_mfma_batched<M, RMUL_XPAY, N>, ldsb);
      hipLaunchKernelGGL((k_rmul_mfma_batched<M, RMUL_XPAY, N>), dim3(grid8), dim3(512), ldsb, s, rows, y, x, Cd, b);
    } else {
      allow_lds(k_rmul_mfma_batched<M, RMUL_MUL, N>, ldsb);
      hipLaunchKernelGGL((k_rmul_mfma_batched<M, RMUL_MUL, N>), dim3(grid8), dim3(512), ldsb, s, rows, y, x, Cd, b);
    }
    return;
  }
  if (m == 8) BCG_RMUL(8) else if (m == 16) BCG_RMUL(16) else BCG_RMUL(32)
#undef BCG_RMUL
}

int launch_gram_mfma(hipStream_t s, int m, int64_t rows, const double2* a, const double2* b, double2* partials,
                     int max_blocks) {
  const int grid = grid_tiles((rows + 3) / 4, 4 * 16, max_blocks);
  if (m == 8) {
    const size_t lds = sizeof(double) * 4 * 8 * 64;
    hipLaunchKernelGGL(k_gram_mfma8, dim3(grid), dim3(256), lds, s, rows, a, b, partials);
  } else if (m == 16) {
    const size_t lds = sizeof(double) * 4 * 8 * 64;
    hipLaunchKernelGGL((k_gram_mfma<16>), dim3(grid), dim3(256), lds, s, rows, a, b, partials);
  } else {
    const size_t lds = sizeof(double) * 4 * 8 * 64;
    hipLaunchKernelGGL((k_gram_mfma<32>), dim3(grid), dim3(256), lds, s, rows, a, b, partials);
  }
  return grid;
}

}  // namespace bcg
