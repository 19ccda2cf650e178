// Shared device helpers of the MFMA fast path (kernels_mfma.hip: the row kernels; kernels_stencil.hip: the stencil):
// the MFMA wrappers, coefficient matrices in LDS, the 16-row tile and its products, Gram accumulation and the in-kernel
// fold of Gram partials.  Everything lives in an anonymous namespace: each translation unit has its own copy.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.hpp"
#include "kernels_mfma.hpp"

#include <type_traits>

namespace bcg {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double dv2 __attribute__((ext_vector_type(2)));  // native vector: promotes to registers where HIP's double2 struct does not

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ d4 mfma_nega(double a, double b, d4 c) {  // c + (-a) * b
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
}

__device__ __forceinline__ double2 ld_nt(const double2* p) {
  const dv2 v = __builtin_nontemporal_load(reinterpret_cast<const dv2*>(p));
  return make_double2(v.x, v.y);
}
// write-through store (sc1): visible device-wide without a release fence -- the hand-off of Gram partials to the folding block
__device__ __forceinline__ void st_sc1(double2* p, double2 v) {
  dv2 w;
  w.x = v.x;
  w.y = v.y;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(w));
}

// ---- coefficient matrices in LDS ------------------------------------------------------------------
// Layout: Ml[j_in * LD + 2*j_out + comp], LD = 2M + 1 doubles (row stride M*16 + 8 bytes keeps the
// two 16-lane halves of a ds_read_b64 group on disjoint banks).
template <int M>
struct MatLds {
  static constexpr int LD = 2 * M + 1;
  static constexpr int DOUBLES = M * LD;
};

// global: complex column-major, element (i,j) at j*M + i.
template <int M>
__device__ __forceinline__ void stage_matrix(double* Ml, const double2* __restrict__ Cg, int tid, int nthreads) {
  for (int e = tid; e < M * M; e += nthreads) {
    const int i = e % M, jo = e / M;
    const double2 v = Cg[e];
    Ml[i * MatLds<M>::LD + 2 * jo] = v.x;
    Ml[i * MatLds<M>::LD + 2 * jo + 1] = v.y;
  }
}

// ---- 16-row tile in (r = l&15, kq = l>>4) ownership -------------------------------------------------
template <int M>
struct Tile {
  double2 v[M / 4];
};

// ok: this lane's row exists.  Callers pass a wave-uniform `true` for every tile but the field's last one when the row count
// is no multiple of 16 (BCG_ROWS_FULL below): the loads and stores of a full tile are then straight-line code instead of
// one predicated block per instruction.
template <int M>
__device__ __forceinline__ void tile_load(Tile<M>& t, const double2* __restrict__ f, int64_t row, int kq, bool ok) {
  const double2* p = f + row * M + kq;
#pragma unroll
  for (int s = 0; s < M / 4; ++s) t.v[s] = ok ? p[4 * s] : make_double2(0.0, 0.0);
}
template <int M>
__device__ __forceinline__ void tile_store(const Tile<M>& t, double2* __restrict__ f, int64_t row, int kq, bool ok) {
  double2* p = f + row * M + kq;
  if (ok) {
#pragma unroll
    for (int s = 0; s < M / 4; ++s) p[4 * s] = t.v[s];
  }
}
#ifdef BCG_ROWS_ALWAYS_FULL  // timing build: valid only for row counts that are multiples of 16
#define BCG_ROW_OK(row, rows) true
#else
#define BCG_ROW_OK(row, rows) ((row) < (rows))
#endif


// accumulator fragments <-> tile.  Output element (s_o, ri_o) of the lane sits in acc[q>>2][q&3],
// q = ri_o*(M/4) + s_o.
template <int M>
struct Acc {
  d4 a[M / 8];
};
template <int M>
__device__ __forceinline__ void acc_from_tile(Acc<M>& A, const Tile<M>& t) {
#pragma unroll
  for (int s = 0; s < M / 4; ++s) {
    A.a[s >> 2][s & 3] = t.v[s].x;
    A.a[(M / 4 + s) >> 2][(M / 4 + s) & 3] = t.v[s].y;
  }
}
template <int M>
__device__ __forceinline__ void acc_zero(Acc<M>& A) {
#pragma unroll
  for (int q = 0; q < M / 8; ++q) A.a[q] = d4{0.0, 0.0, 0.0, 0.0};
}
template <int M>
__device__ __forceinline__ void tile_from_acc(Tile<M>& t, const Acc<M>& A) {
#pragma unroll
  for (int s = 0; s < M / 4; ++s) {
    t.v[s].x = A.a[s >> 2][s & 3];
    t.v[s].y = A.a[(M / 4 + s) >> 2][(M / 4 + s) & 3];
  }
}

// acc += in * C   (C staged in LDS at Ml).  For M >= 16 the re/im block of an output tile is the same for all
// lanes, so the choice between Re C and Im C and the sign are compile-time (the minus is the MFMA's NEG modifier).
// For M = 8 one 16 x 16 tile holds both blocks: the A-operand lane picks its coefficient by its own output slot.
template <int M>
__device__ __forceinline__ void rmul_acc(Acc<M>& A, const Tile<M>& in, const double* Ml, int lane) {
  static_assert(M == 8 || M == 16 || M == 32, "MFMA right-multiply is instantiated for m = 8, 16, 32");
  constexpr int LD = MatLds<M>::LD;
  const int kq = lane >> 4;          // which of the 4 k-slots of a step this lane feeds (B operand)
  const int ar = lane & 15;          // A-operand row: output slot dr = kq_o + 4*reg_o
#pragma unroll
  for (int T = 0; T < (M + 7) / 8; ++T) {
    const int q_o = 4 * T + (ar >> 2);                           // output slot index: ri_o*(M/4) + s_o
    const int ri_o = q_o / (M / 4);                              // lane independent for M >= 16
    const int s_o = q_o % (M / 4);
    const int j_o = 4 * s_o + (ar & 3);
    const double* base = Ml + kq * LD + 2 * j_o;                 // + s_i*4*LD + comp
#pragma unroll
    for (int s_i = 0; s_i < M / 4; ++s_i) {
      const double a_same = base[s_i * 4 * LD + 0];              // Re C(j_i, j_o)
      const double a_cross = base[s_i * 4 * LD + 1];             // Im C(j_i, j_o)
      if (M >= 16) {
        if ((4 * T) / (M / 4) == 0) {
          // out_re += in_re * Re C - in_im * Im C
          A.a[T] = mfma(a_same, in.v[s_i].x, A.a[T]);
          A.a[T] = mfma_nega(a_cross, in.v[s_i].y, A.a[T]);
        } else {
          // out_im += in_re * Im C + in_im * Re C
          A.a[T] = mfma(a_cross, in.v[s_i].x, A.a[T]);
          A.a[T] = mfma(a_same, in.v[s_i].y, A.a[T]);
        }
      } else {
        const double c_re = ri_o ? a_cross : a_same;             // coefficient of in_re for this lane's output slot
        const double c_im = ri_o ? a_same : -a_cross;            // coefficient of in_im
        A.a[T] = mfma(c_re, in.v[s_i].x, A.a[T]);
        A.a[T] = mfma(c_im, in.v[s_i].y, A.a[T]);
      }
    }
  }
}

// Two products sharing the B operand (the P_s tile): X += P*Ca and Pn += P*Cb.
template <int M>
__device__ __forceinline__ void rmul_acc2(Acc<M>& A1, const double* Ml1, Acc<M>& A2, const double* Ml2, const Tile<M>& in,
                                          int lane) {
  constexpr int LD = MatLds<M>::LD;
  const int kq = lane >> 4;
  const int ar = lane & 15;
#pragma unroll
  for (int T = 0; T < (M + 7) / 8; ++T) {
    const int q_o = 4 * T + (ar >> 2);
    const int ri_o = q_o / (M / 4);
    const int s_o = q_o % (M / 4);
    const int j_o = 4 * s_o + (ar & 3);
    const int off = kq * LD + 2 * j_o;
#pragma unroll
    for (int s_i = 0; s_i < M / 4; ++s_i) {
      const double a1s = Ml1[off + s_i * 4 * LD], a1c = Ml1[off + s_i * 4 * LD + 1];
      const double a2s = Ml2[off + s_i * 4 * LD], a2c = Ml2[off + s_i * 4 * LD + 1];
      if (M >= 16) {
        if ((4 * T) / (M / 4) == 0) {
          A1.a[T] = mfma(a1s, in.v[s_i].x, A1.a[T]);
          A2.a[T] = mfma(a2s, in.v[s_i].x, A2.a[T]);
          A1.a[T] = mfma_nega(a1c, in.v[s_i].y, A1.a[T]);
          A2.a[T] = mfma_nega(a2c, in.v[s_i].y, A2.a[T]);
        } else {
          A1.a[T] = mfma(a1c, in.v[s_i].x, A1.a[T]);
          A2.a[T] = mfma(a2c, in.v[s_i].x, A2.a[T]);
          A1.a[T] = mfma(a1s, in.v[s_i].y, A1.a[T]);
          A2.a[T] = mfma(a2s, in.v[s_i].y, A2.a[T]);
        }
      } else {
        const double c1r = ri_o ? a1c : a1s, c1i = ri_o ? a1s : -a1c;
        const double c2r = ri_o ? a2c : a2s, c2i = ri_o ? a2s : -a2c;
        A1.a[T] = mfma(c1r, in.v[s_i].x, A1.a[T]);
        A2.a[T] = mfma(c2r, in.v[s_i].x, A2.a[T]);
        A1.a[T] = mfma(c1i, in.v[s_i].y, A1.a[T]);
        A2.a[T] = mfma(c2i, in.v[s_i].y, A2.a[T]);
      }
    }
  }
}

// ---- Gram accumulation ---------------------------------------------------------------------------
// Operands in (row k = l>>4, column n = l&15 [+16 jb]) ownership.  Per j-block pair (ja, jb):
//   re(ja,jb) += a_re*b_re + a_im*b_im ;  im(ja,jb) += a_re*b_im - a_im*b_re      (conj(a) * b)
template <int M>
struct GramAcc {
  d4 re[(M / 16) * (M / 16)];
  d4 im[(M / 16) * (M / 16)];
};
template <int M>
__device__ __forceinline__ void gram_zero(GramAcc<M>& G) {
#pragma unroll
  for (int q = 0; q < (M / 16) * (M / 16); ++q) {
    G.re[q] = d4{0.0, 0.0, 0.0, 0.0};
    G.im[q] = d4{0.0, 0.0, 0.0, 0.0};
  }
}
// a[jb], b[jb]: the lane's complex element of column 16*jb + (l&15) for the wave's 4 rows of this step
template <int M>
__device__ __forceinline__ void gram_step(GramAcc<M>& G, const double2* a, const double2* b) {
  constexpr int JB = M / 16;
#pragma unroll
  for (int ja = 0; ja < JB; ++ja)
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      const int q = ja * JB + jb;
      G.re[q] = mfma(a[ja].x, b[jb].x, G.re[q]);
      G.re[q] = mfma(a[ja].y, b[jb].y, G.re[q]);
      G.im[q] = mfma(a[ja].x, b[jb].y, G.im[q]);
      G.im[q] = mfma_nega(a[ja].y, b[jb].x, G.im[q]);
    }
}

// Sum the per-wave fragments of a block in wave order and write partials[block][j*M + i].
// red: LDS scratch of NW * JB*JB * 2 * 4 * 64 doubles.
// wt: write-through (sc1) stores -- the partials are handed to another workgroup inside this launch (gram_fold)
__device__ __forceinline__ void st_partial(double2* p, double2 v, bool wt) {
  if (wt) st_sc1(p, v);
  else *p = v;
}
template <int M, int NW>
__device__ __forceinline__ void gram_block_store(const GramAcc<M>& G, double* red, double2* __restrict__ partials, int tid,
                                                 bool wt = false) {
  // one 16 x 16 block of the Gram matrix at a time through a buffer of NW * 8 * 64 doubles (16 KB at four waves): at
  // m = 32 a buffer for all four blocks was 64 KB and left phase B a single block per CU
  constexpr int JB = M / 16;
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int q = 0; q < JB * JB; ++q) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[(wave * 8 + r) * 64 + lane] = G.re[q][r];
      red[(wave * 8 + 4 + r) * 64 + lane] = G.im[q][r];
    }
    __syncthreads();
    // element e = (r, lane): i = 16*ja + (lane>>4) + 4r, j = 16*jb + (lane&15)
    for (int e = tid; e < 4 * 64; e += NW * 64) {
      const int l = e & 63, r = (e >> 6) & 3;
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        sr += red[(w * 8 + r) * 64 + l];
        si += red[(w * 8 + 4 + r) * 64 + l];
      }
      const int i = 16 * (q / JB) + (l >> 4) + 4 * r, j = 16 * (q % JB) + (l & 15);
      st_partial(partials + static_cast<int64_t>(blockIdx.x) * (M * M) + j * M + i, make_double2(sr, si), wt);
    }
  }
}

// m = 8 on the 16x16 MFMA tile.  With lane = (row pair k = l>>4, row parity s = (l>>3)&1, column c = l&7) holding the
// element (row 2k+s, column c) of both operands, gram_step<16> accumulates C[(s,i)][(s',j)] = sum_k conj(a(2k+s,i)) b(2k+s',j):
// the two diagonal blocks s = s' are Gram contributions (even and odd rows), the off-diagonal ones are discarded.  So
// the whole m = 16 machinery applies unchanged and only this final store differs: G(i,j) = C[i][j] + C[8+i][8+j].
template <int NW>
__device__ __forceinline__ void gram_block_store_fold8(const GramAcc<16>& G, double* red, double2* __restrict__ partials, int tid,
                                                       bool wt = false) {
  constexpr int FR = 8;  // doubles per lane
  const int wave = tid >> 6, lane = tid & 63;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[((wave * FR) + r) * 64 + lane] = G.re[0][r];
    red[((wave * FR) + 4 + r) * 64 + lane] = G.im[0][r];
  }
  __syncthreads();
  // fragment element (r, l): i16 = (l>>4) + 4r, j16 = l&15.  Block (0,0): r < 2, (l&15) < 8; its partner in block (1,1): (r+2, l+8)
  for (int e = tid; e < 4 * 64; e += NW * 64) {
    const int l = e & 63, r = e >> 6;
    if (r >= 2 || (l & 15) >= 8) continue;
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      sr += red[((w * FR) + r) * 64 + l] + red[((w * FR) + r + 2) * 64 + l + 8];
      si += red[((w * FR) + 4 + r) * 64 + l] + red[((w * FR) + 4 + r + 2) * 64 + l + 8];
    }
    const int i = (l >> 4) + 4 * r, j = l & 7;
    st_partial(partials + static_cast<int64_t>(blockIdx.x) * 64 + j * 8 + i, make_double2(sr, si), wt);
  }
}


// ---- Gram partials folded inside the producing kernel ------------------------------------------------------------
// After a block has written partials[blockIdx.x][NV], the LAST block to finish of each of 8 groups (blocks g, g + 8, ...)
// sums its group's partials in block order into partials[gridDim.x + g], and the last of the 8 groups to finish sums those
// in group order into `out`: a fixed order whatever the arrival order (bitwise reproducible), no separate reduction
// launch.  The hand-off between workgroups uses no cache-wide fence (a release would write back every dirty line these
// kernels have just produced -- measured: +80 us on a 0.22 ms phase B): every handed-off byte is stored write-through
// (sc1; the callers pass wt = true to gram_block_store), every storing wave drains its stores, a block barrier, ONE lane
// takes an agent-scope ticket, and the block whose ticket is the last reads the bytes with sc1 loads (the L1 is bypassed,
// the L2 is the point of coherence for write-through data).
// tickets: 9 words, zero before the first launch; the final block leaves them zero again.  All threads of the block call it.
// Sum of n values p[0], p[stride], ... read with sc1 loads, in index order; eight loads in flight at a time (one load and
// its wait at a time made the fold's tail 128 serial L2 round trips).
__device__ __forceinline__ double2 sum_sc1(const double2* p, int64_t stride, int n) {
  double sr = 0.0, si = 0.0;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    dv2 r0, r1, r2, r3, r4, r5, r6, r7;
    const double2* q = p + static_cast<int64_t>(k) * stride;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r0) : "v"(q));
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r1) : "v"(q + stride));
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r2) : "v"(q + 2 * stride));
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r3) : "v"(q + 3 * stride));
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r4) : "v"(q + 4 * stride));
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r5) : "v"(q + 5 * stride));
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r6) : "v"(q + 6 * stride));
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r7) : "v"(q + 7 * stride));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : : "memory");
    sr += r0.x; si += r0.y; sr += r1.x; si += r1.y; sr += r2.x; si += r2.y; sr += r3.x; si += r3.y;
    sr += r4.x; si += r4.y; sr += r5.x; si += r5.y; sr += r6.x; si += r6.y; sr += r7.x; si += r7.y;
  }
  for (; k < n; ++k) {
    dv2 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p + static_cast<int64_t>(k) * stride) : "memory");
    sr += r.x;
    si += r.y;
  }
  return make_double2(sr, si);
}
template <int NV>
__device__ __forceinline__ void gram_fold(const GramFold& gf, double2* __restrict__ partials, int tid, int nthreads) {
  if (gf.out == nullptr) return;
  __shared__ int s_role;
  const int nb = gridDim.x, g = blockIdx.x & 7;
  const int in_group = (nb - g + 7) >> 3;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through partial stores have reached the L2
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(gf.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_role = t == static_cast<unsigned>(in_group - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_role) return;
  double2* const level2 = partials + static_cast<int64_t>(nb) * NV;
  for (int v = tid; v < NV; v += nthreads)
    st_sc1(level2 + g * NV + v, sum_sc1(partials + static_cast<int64_t>(g) * NV + v, static_cast<int64_t>(8) * NV, in_group));
  const int groups = nb < 8 ? nb : 8;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(gf.tickets + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_role = t == static_cast<unsigned>(groups - 1) ? 2 : 0;
  }
  __syncthreads();
  if (s_role != 2) return;
  for (int v = tid; v < NV; v += nthreads) gf.out[v] = sum_sc1(level2 + v, NV, groups);
  if (tid < 9) gf.tickets[tid] = 0u;  // ready for the next launch (stream order makes this visible to it)
}

inline int grid_tiles(int64_t ntiles, int per_block, int cap) {
  int64_t g = (ntiles + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

template <typename K>
void allow_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

}  // namespace
}  // namespace bcg
