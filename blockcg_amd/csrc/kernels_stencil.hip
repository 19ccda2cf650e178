// The stencil of the MFMA fast path for gfx950 (MI355X): dirac_op::D (inc/dirac_op.hpp:14-21) and its fused forms, block
// widths m = 8, 16, 32 -- k_hop4 / k_hop4c (4-D tile and column forms, tile classes of the split
// halo exchange) and k_hop4b (2 x 2 column bundles: the software-pipelined, broadcast-link step; ring windows; the
// checkerboard form for half-volume fields) -- with their launchers.  Shared helpers: mfma_common.hpp.
#include "mfma_common.hpp"

namespace bcg {

namespace {

// LDS-DMA (global_load_lds_dwordx4): 16 bytes per active lane from the lane's own global address straight into LDS at
// `lds_dst` + 16 * lane (wave-uniform base in M0; no VGPR destination).  Written as asm so that hipcc does not see it:
// it would otherwise wait vmcnt(0) at the next use of ANY ordinary load while one is in flight.  Consequences the callers
// rely on: vector-memory operations complete in issue order, so once the wave has consumed an ordinary load issued AFTER
// the DMA, the DMA has landed; nothing else orders a later ds_read behind it.
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_char_t*)p));
}
// Settled in earlier rounds and no longer switches (profiles/r03_stencil_incremental_addresses.txt, r04_stencil_spread.txt):
// the step's addresses are carried along a column instead of recomputed (INCR), the wrap row of a column's last slice too
// in the forms without the fused Gram product (plain hop 10.05-10.23 vs 10.31-10.49 ms; with the product 12.05-12.37 vs
// 11.90-12.16), and the plain form issues its vector-memory instructions grouped (SPREAD 0; the shifted forms: SPREAD below).
// PIPE: the software-pipelined schedule of the bundle sweep (m = 16, 32; full-lattice form): every global access of a step is
// an operation hipcc does not see, issued where it pays and retired by hand-counted s_waitcnt -- see "PIPE" in hop4b_body.
// -DBCG_HOP4B_PIPE=0 is the A/B build: the step of rounds 1-3 (still what m = 8 and the residual form run).
// The tuning builds of round 3 (the +x3 row by LDS-DMA, rows one step ahead, scheduling-barrier masks, write-through stores,
// non-temporal link DMAs, the ablations of profiles/r03_stencil_ablation.txt, the carried-address check) have been removed:
// their results are recorded in profiles/r03_stencil_*.txt and DESIGN_HISTORY.md, and PIPE supersedes what they explored.
#ifndef BCG_HOP4B_PIPE
#define BCG_HOP4B_PIPE 1
#endif
// SPREAD (bit mask per form): parts of the step's vector-memory instructions issued one or two at a time behind the 24-FMA
// units instead of in a group behind a direction -- 1: the row DMAs, 2: the link DMAs, 4: the second group of next-step
// rows behind direction 2 (the waits only need rows first, p second; the rest may come in any order).  Measured at 64^4,
// three alternating runs per build (profiles/r04_stencil_spread.txt): fused form 11.41 ms grouped, 11.27 with 6, 11.31 with 7,
// 11.52 with 3; plain form 8.69 grouped, 8.67 with 1 or 4, 8.94 with 2 -- i.e. the queueing of grouped issue is worth 1 %.
#ifndef BCG_HOP4B_SPREAD
#define BCG_HOP4B_SPREAD 6
#endif
// BCAST: a link entry is read from the image ONCE per site -- the site's 72 entries of a step spread over the 16 lanes of a
// row, 4.5 apiece, six ds_read_b128 per lane at the top of the step -- and reaches the FMAs of all 16 right-hand sides as the
// broadcast operand of v_fmac_f64_dpp (row_newbcast:n = lane n of the lane's row of 16), instead of every lane reading every
// entry (72 ds_read_b128 per lane and step: with the rows' 24 that was 3072 LDS-array cycles per CU and step against 2477 of
// fp64 pipe -- the directions ran at the LDS array's rate, tools/microbench/fma_f64_issue.hip).  Same products, same order
// along every accumulation chain: bit-identical.  -DBCG_HOP4B_BCAST=0: the per-lane reads (A/B build).
#ifndef BCG_HOP4B_BCAST
#define BCG_HOP4B_BCAST 1
#endif
__device__ __forceinline__ void glds16_link(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst));
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst));  // no "memory" clobber: it would force the callers' captured state into scratch;
}                                            // the block barriers around every use order it against the compiler's LDS accesses
// The same with the source given as base + per-lane offset.  -DBCG_HOP4B_GLDS_SBASE=1: the scalar-base encoding of the
// instruction (`global_load_lds_dwordx4 v_off, s[base:base+1]`), which saves the 64-bit vector address arithmetic in front
// of every DMA.  Round 4 tried it and the kernel aborted at its first launch (a memory fault), with and without an immediate
// offset.  Cause (round 5, from the hazard table the asm guide gives for operands INSIDE an asm string): the "s" operand is
// materialised by v_readfirstlane where hipcc keeps the wave-uniform pointer in VGPRs, and a VALU write of an SGPR needs
// five wait states before a vector-memory instruction reads it as its base; hipcc pads hazards only for instructions it
// emits itself, and the string's two s_mov and one s_nop in front of the load are three.  With `s_nop 4` opening the string
// the form runs and passes the parity suites (profiles/r05_glds_scalar_base.txt has the A/B).
#ifndef BCG_HOP4B_GLDS_SBASE
#define BCG_HOP4B_GLDS_SBASE 0
#endif
#if BCG_HOP4B_GLDS_SBASE
__device__ __forceinline__ void glds16_s(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  // (17 of the call sites hold their wave-uniform base in VGPRs: an "s" operand on it does not assemble -- hipcc hands
  //  the asm a VGPR pair -- so the halves are read into SGPRs here, which is where the hazard comes from)
  const unsigned long long a = reinterpret_cast<unsigned long long>(sbase);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
  const unsigned long long sb = (static_cast<unsigned long long>(hi) << 32) | lo;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sb), "s"(lds_dst));
}
#else
__device__ __forceinline__ void glds16_s(const char* sbase, unsigned voff, unsigned lds_dst) { glds16_link(sbase + voff, lds_dst); }
#endif
// One row element per lane (16 bytes at sbase + voff + IMM) by a load hipcc does not see: its result counts as available at
// once, so the CALLER waits (s_waitcnt vmcnt) before the first use.  For values loaded one loop iteration ahead: hipcc's
// own bookkeeping loses count across the loop's back edge and waits for every load of the NEW iteration at their use.
template <int IMM>
__device__ __forceinline__ dv2 ld_sv_async(const char* sbase, unsigned voff) {
  dv2 r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM));
  return r;
}
// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the immediate is part of the instruction): until at most
// the n YOUNGEST vector-memory operations of the wave are outstanding.  n above the table waits for everything (stricter).
// acc += (+/-) u(lane LANE of this lane's row of 16) * p
template <int LANE, bool NEG>
__device__ __forceinline__ void fmac_bcast(double& acc, double u, double p) {
  if (NEG) asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(p), "n"(LANE));
  else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(p), "n"(LANE));
}

__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
#define BCG_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    BCG_W(1) BCG_W(2) BCG_W(3) BCG_W(4) BCG_W(5) BCG_W(6) BCG_W(7) BCG_W(8) BCG_W(9) BCG_W(10) BCG_W(11) BCG_W(12)
    BCG_W(13) BCG_W(14) BCG_W(15) BCG_W(16) BCG_W(17) BCG_W(18) BCG_W(19) BCG_W(20)
#undef BCG_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// link loads of the stencil (non-temporal loads were tried here: 13 % fewer L2 misses, no time gained)
__device__ __forceinline__ dv2 ld_link(const dv2* p) { return *p; }
__device__ __forceinline__ void st_nt(double2* p, double2 v) {
  dv2 w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<dv2*>(p));
}

// ---------------------------------------------------------------------------------------------------
// (The general form of the stencil for m = 8, 16, 32 -- k_hop_fast: any number of dimensions and extents, links staged per
// tile in LDS -- was retired in round 5: lattices the specialised 4-D kernels below do not take (fewer than four dimensions,
// L0 not a multiple of the tile) run k_hop_generic like every other width; measured in profiles/r05_retired_general_stencil.txt.)
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
struct TileGeom {
  int x0b, x1, x2, x3;   // coordinates of the tile's first site
  int site0;             // its local site index
  int edge;              // bit 0/1: the +x1/-x1 neighbour tile lies outside the tile's patch; bit 2/3: +x2/-x2
};

// Tile order: mixed-radix counter, fastest digit first:
//   (x0 tile in patch, x1 in patch, x2 in patch, x3, patch x0, patch x1, patch x2)
// Lexicographic order is the special case patch = whole (x0,x1,x2) volume.  With xcd_split the
// sequence is cut into 8 contiguous ranges, one per block class b%8 (blocks b and b+8 share an XCD
// and its L2 under round-robin dispatch): each XCD sweeps x3 over one compact patch at a time, so
// three x3-slices of a patch can live in its 4 MiB L2.  Performance only: every tile is visited once.
struct HopWalk {
  int p0, p1, p2;        // patch extents (sites) in x0, x1, x2
  int xcd_split;
  // Pacing of the blocks that share an XCD (k_hop4c only; nullptr = off).  The L2 re-use of the x3 walk needs those
  // blocks on neighbouring slices, but nothing couples them and they spread over 7-15 slices (measured with
  // -DBCG_HOP4_TRACE), so the slices they re-read have left the L2.  sync[class * stride + n] counts the blocks of a
  // class that finished their n-th tile; a block starts tile n only when all of them finished tile n - window (bounded
  // wait: after `sync_limit` ticks of the 100 MHz clock it stops pacing, so a block that is not resident cannot hang the rest).
  unsigned* sync;
  int sync_window, sync_stride, sync_limit;
  // k_hop4 only: an explicit list of tiles (first site of each), dealt to the blocks round-robin, instead of the counter.
  // Used for the boundary class of the split halo exchange: under the patch walk its tiles sit in a few columns, i.e. on
  // a few blocks (all x2-edge columns belong to two XCD classes), and that launch is on the critical path.
  const int* tile_list;
  int list_n;
  // k_hop4c / k_hop4b with the fused Gram product: fold the block partials inside the kernel (gram_fold); out = nullptr: off
  GramFold fold;
  // k_hop4b: the patches the eight XCD classes work on at the same time form a 2 x 2 x 2 SUPER-PATCH (class bit 0 / 1 / 2 =
  // the patch's position in x0 / x1 / x2 inside it) instead of lying a quarter of the lattice apart: the rows that cross
  // a face between two of them are then read by two XCDs within a few steps of each other, and the second read can be
  // served by the memory-side cache (HopTuning::super_patch).  Tile order only: every tile is still visited once.
  int super;
};

// Pacing counters are read with the same read-modify-write unit that increments them (an add of 0): the blocks of a
// class share an XCD and so an L2, where those atomics execute; a device-scope LOAD instead goes past the L2 and took
// microseconds per tile (measured: the stencil ran 2x slower).
// `zero` is a run-time 0: a literal would let the compiler turn the add into exactly that load.
__device__ __forceinline__ unsigned read_counter(unsigned* ctr, unsigned zero) {
  return __hip_atomic_fetch_add(ctr, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Digits {
  int d0, d1, d2, d3, d4, d5, d6;
};

__device__ __forceinline__ Digits digits_of(unsigned v, int r0, int r1, int r2, int r3, int r4, int r5) {
  Digits g;
  g.d0 = v % r0; v /= r0;
  g.d1 = v % r1; v /= r1;
  g.d2 = v % r2; v /= r2;
  g.d3 = v % r3; v /= r3;
  g.d4 = v % r4; v /= r4;
  g.d5 = v % r5; v /= r5;
  g.d6 = v;
  return g;
}
#define BCG_ADD_DIGIT(D, S, R)            \
  {                                       \
    D += S + carry;                       \
    carry = 0;                            \
    if (D >= R) { D -= R; carry = 1; }    \
  }
__device__ __forceinline__ void digits_add(Digits& a, const Digits& s, int r0, int r1, int r2, int r3, int r4, int r5) {
  int carry = 0;
  BCG_ADD_DIGIT(a.d0, s.d0, r0)
  BCG_ADD_DIGIT(a.d1, s.d1, r1)
  BCG_ADD_DIGIT(a.d2, s.d2, r2)
  BCG_ADD_DIGIT(a.d3, s.d3, r3)
  BCG_ADD_DIGIT(a.d4, s.d4, r4)
  BCG_ADD_DIGIT(a.d5, s.d5, r5)
  a.d6 += s.d6 + carry;
}
#undef BCG_ADD_DIGIT

template <int SPB>
__device__ __forceinline__ TileGeom geom_of(const Digits& d, int r0, int p1, int p2, int L0, int L1, int L2, int x3_lo) {
  TileGeom g;
  g.x0b = (d.d4 * r0 + d.d0) * SPB;
  g.x1 = d.d5 * p1 + d.d1;
  g.x2 = d.d6 * p2 + d.d2;
  g.x3 = x3_lo + d.d3;
  g.site0 = g.x0b + L0 * (g.x1 + L1 * (g.x2 + L2 * g.x3));
  g.edge = (d.d1 == p1 - 1 ? 1 : 0) | (d.d1 == 0 ? 2 : 0) | (d.d2 == p2 - 1 ? 4 : 0) | (d.d2 == 0 ? 8 : 0);
  return g;
}

// Forward links of a tile: SPB*36 contiguous complex numbers, coalesced over the 256 threads.
template <int SPB, int RF>
__device__ __forceinline__ void fetch_fwd(const double2* __restrict__ fsrc, int tid, dv2 (&rf)[RF]) {
#pragma unroll
  for (int k = 0; k < RF; ++k) {
    const int e = tid + 256 * k;
    if (e < SPB * 36) rf[k] = ld_link(reinterpret_cast<const dv2*>(fsrc + e));
  }
}

// Backward links U_mu(x - mu) of one direction mu >= 1 for the tile's SPB sites (compile-time mu: a
// run-time mu would make the compiler index a scratch copy of the geometry).  Thread t < SPB*9 (+256 k)
// fetches element c9 of site s.
template <int SPB, int RBM>
__device__ __forceinline__ void fetch_back(int mu, int xm, int Lm, int Sm, int spm, int64_t gm, int fi0, int site0,
                                           const double2* __restrict__ U, const double2* __restrict__ Ughost, int tid,
                                           dv2 (&rb)[RBM]) {
#pragma unroll
  for (int k = 0; k < RBM; ++k) {
    const int e = tid + 256 * k;
    if (e < SPB * 9) {
      const int s = e / 9, c9 = e - s * 9;
      const double2* src;
      if (xm > 0) src = U + ((static_cast<int64_t>(site0) + s - Sm) * 4 + mu) * 9;
      else if (!spm) src = U + ((static_cast<int64_t>(site0) + s + static_cast<int64_t>(Lm - 1) * Sm) * 4 + mu) * 9;
      else src = Ughost + (gm + fi0 + s) * 9;
      rb[k] = ld_link(reinterpret_cast<const dv2*>(src + c9));
    }
  }
}

// Measured on one device (tools/ab_bench.sh): carrying the -x3 neighbours / U_3(x-3) across steps pays for the
// fused-Gram variant (15.6 -> 13.8 ms at 64^4) but not for the plain hop (11.7 -> 13.0 ms), and asking for a minimum
// of 2 waves per SIMD in __launch_bounds__ costs 1.5 ms on both, so: CARRY = GRAM, plain __launch_bounds__(256).
// CLS selects the tiles a launch processes: 0 all, 1 interior only (no site of the tile reads a ghost), 2 boundary only.
// Interior and boundary launches bracket the halo exchange so that it overlaps the interior arithmetic.
template <int M, int MODE, bool GRAM, bool NT, int CLS, bool RING>
__global__ void __launch_bounds__(256) k_hop4(LatticeDev lat, const double2* __restrict__ U,
                                              const double2* __restrict__ Ughost, const double2* __restrict__ in,
                                              const double2* __restrict__ ghost, double2* __restrict__ out,
                                              const double2* __restrict__ p, double c0,
                                              double2* __restrict__ partials, int ntiles, HopWalk hw, HopWindow win) {
  static_assert(!GRAM || M == 16, "fused Gram accumulation needs lane&15 == rhs index");
  // capacity mode (HopWindow::ring): the intermediate field is a ring of x3 slices, written by HOP_PLAIN, read by HOP_SHIFTED
  constexpr bool RING_OUT = RING && MODE == HOP_PLAIN;
  constexpr bool RING_IN = RING && MODE == HOP_SHIFTED;
  constexpr bool CARRY = GRAM;
  constexpr bool CARRY_B3 = CARRY;   // -x3 neighbours from the register history
  constexpr bool CARRY_U3 = CARRY;   // U_3(x-3) from the previous link image
  constexpr int STAGES = CARRY_U3 ? 3 : 2;
  // run-time variants in this kernel cost registers (256 VGPRs with one extra branch): none kept
  constexpr int SPW = 64 / M;
  constexpr int SPB = 4 * SPW;
  constexpr int NW = 4;
  constexpr int NF = (SPB + 1) * 36;   // forward links of sites -1 .. SPB-1 (all 4 directions)
  constexpr int NB = 3 * SPB * 9;      // backward links of directions 1..3
  constexpr int STAGE = NF + NB;       // double2 per LDS stage
  constexpr int RF = (SPB * 36 + 255) / 256;  // register slots per thread for the forward run
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double2* Ls = reinterpret_cast<double2*>(smem);  // [3][STAGE] link images
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sl = wave * SPW + lane / M;
  const int j = lane % M;
  // x3 carry: under the patch walk a block visits (tile, x3), (tile, x3+1), ... so the -x3 neighbour of step n is
  // the +x3 neighbour this lane loaded at step n-2 (kept in registers), and U_3(x-3) is the forward link
  // of step n-1, still in the other link stage.  Both would otherwise be re-fetched past L2.
  dv2 h1[3], h2[3];  // the lane's +x3 neighbour values of the previous two steps (registers)
#pragma unroll
  for (int k = 0; k < 3; ++k) h1[k] = h2[k] = dv2{0.0, 0.0};
  // scalar copies of the geometry
  const int L0 = lat.L[0], L1 = lat.L[1], L2 = lat.L[2], L3 = lat.L[3];
  const int S1 = L0, S2 = L0 * L1, S3 = L0 * L1 * L2;
  const int sp0 = lat.split[0], sp1 = lat.split[1], sp2 = lat.split[2], sp3 = lat.split[3];
  const int64_t gm0 = lat.ghost_off[0][0], gp0 = lat.ghost_off[0][1];
  const int64_t gm1 = lat.ghost_off[1][0], gp1 = lat.ghost_off[1][1];
  const int64_t gm2 = lat.ghost_off[2][0], gp2 = lat.ghost_off[2][1];
  const int64_t gm3 = lat.ghost_off[3][0], gp3 = lat.ghost_off[3][1];
  const int og0 = lat.origin[0], og1 = lat.origin[1], og2 = lat.origin[2];
  GramAcc<16> G;
  if (GRAM) gram_zero(G);

  // ---- tile sequence of this block
  const int r0 = hw.p0 / SPB, r1 = hw.p1, r2 = hw.p2, r3 = win.x3_n, r4 = L0 / hw.p0, r5 = L1 / hw.p1;
  const int x3_lo = win.x3_lo;
  unsigned start = blockIdx.x, step = gridDim.x, left;
  if (hw.xcd_split) {
    const unsigned cnt = ntiles >> 3, cls = blockIdx.x & 7, idx = blockIdx.x >> 3;
    step = gridDim.x >> 3;
    start = cls * cnt + idx;
    left = idx < cnt ? (cnt - idx + step - 1) / step : 0;
  } else {
    left = start < static_cast<unsigned>(ntiles) ? (ntiles - start + step - 1) / step : 0;
  }
  Digits dg = digits_of(start, r0, r1, r2, r3, r4, r5);
  const Digits ds = digits_of(step, r0, r1, r2, r3, r4, r5);
  // ---- link registers: forward run, the one extra link U_0(x0b - 1), backward links of directions 1..3
  constexpr int RBM = (SPB * 9 + 255) / 256;
  dv2 rf[RF], rx, rb1[RBM], rb2[RBM], rb3[RBM];
  rx = dv2{0.0, 0.0};
#define BCG_FETCH_LINKS(g, SKIP3)                                                                                      \
  {                                                                                                               \
    fetch_fwd<SPB, RF>(U + static_cast<int64_t>((g).site0) * 36, tid, rf);                                        \
    if (tid < 9) {                                                                                                \
      const double2* src;                                                                                         \
      if ((g).x0b > 0) src = U + (static_cast<int64_t>((g).site0) - 1) * 36;                                      \
      else if (!sp0) src = U + (static_cast<int64_t>((g).site0) + L0 - 1) * 36;                                   \
      else src = Ughost + (gm0 + ((g).x1 + L1 * ((g).x2 + L2 * (g).x3))) * 9;                                     \
      rx = ld_link(reinterpret_cast<const dv2*>(src + tid));                                                      \
    }                                                                                                             \
    fetch_back<SPB, RBM>(1, (g).x1, L1, S1, sp1, gm1, (g).x0b + L0 * ((g).x2 + L2 * (g).x3), (g).site0, U, Ughost, tid, rb1); \
    fetch_back<SPB, RBM>(2, (g).x2, L2, S2, sp2, gm2, (g).x0b + L0 * ((g).x1 + L1 * (g).x3), (g).site0, U, Ughost, tid, rb2); \
    if (!(SKIP3))                                                                                                 \
      fetch_back<SPB, RBM>(3, (g).x3, L3, S3, sp3, gm3, (g).x0b + L0 * ((g).x1 + L1 * (g).x2), (g).site0, U, Ughost, tid, rb3); \
  }

  // tile classes: a tile is "boundary" when one of its sites has a neighbour in a ghost face
  auto wanted = [&](const TileGeom& t) -> bool {
    if (CLS == 0) return true;
    const bool bnd = (sp0 && (t.x0b == 0 || t.x0b + SPB == L0)) || (sp1 && (t.x1 == 0 || t.x1 == L1 - 1)) ||
                     (sp2 && (t.x2 == 0 || t.x2 == L2 - 1)) || (sp3 && (t.x3 == 0 || t.x3 == L3 - 1));
    return bnd == (CLS == 2);
  };
  const int* const tlist = hw.tile_list;
  // list mode: position in the list, dealt round-robin over all blocks (cutting the list into one contiguous part per XCD
  // class instead was 2-5x slower in the two-rank rehearsal: the blocks of a class then sit on neighbouring addresses)
  unsigned lpos = blockIdx.x;
  auto geom_of_site = [&](int site0) -> TileGeom {
    TileGeom t;
    int q = site0 / L0;
    t.x0b = site0 - q * L0;
    const int q2 = q / L1;
    t.x1 = q - q2 * L1;
    t.x3 = q2 / L2;
    t.x2 = q2 - t.x3 * L2;
    t.site0 = site0;
    t.edge = 0;
    return t;
  };
  // advance to the next wanted tile; false when the block's sequence is exhausted
  auto next_tile = [&](TileGeom& t) -> bool {
    if (tlist != nullptr) {
      lpos += gridDim.x;
      if (lpos >= static_cast<unsigned>(hw.list_n)) return false;
      t = geom_of_site(tlist[lpos]);
      return true;
    }
    while (left > 0) {
      digits_add(dg, ds, r0, r1, r2, r3, r4, r5);
      --left;
      t = geom_of<SPB>(dg, r0, hw.p1, hw.p2, L0, L1, L2, x3_lo);
      if (wanted(t)) return true;
    }
    return false;
  };
  TileGeom g = geom_of<SPB>(dg, r0, hw.p1, hw.p2, L0, L1, L2, x3_lo);
  bool have = left > 0;
  if (tlist != nullptr) {
    have = lpos < static_cast<unsigned>(hw.list_n);
    if (have) g = geom_of_site(tlist[lpos]);
  } else if (have) {
    --left;  // `left` now counts the positions after the current one
    if (!wanted(g)) have = next_tile(g);
  }
  if (have) BCG_FETCH_LINKS(g, false)
  int stage = 0;
#ifdef BCG_HOP4_TRACE
  int trace_n = 0;
#endif
  int site_m1 = -1;          // site0 of the previous tile of this block
  int fsite_m1 = -1, fsite_m2 = -1;  // first site of the +x3 neighbour tile loaded 1 and 2 steps ago (-1: ghost)
  bool carry_u3 = false;     // this tile's U_3(x-3) is the previous tile's forward link
  while (have) {
    {  // park the links fetched for this tile in the current LDS stage
      dv2* Lf = reinterpret_cast<dv2*>(Ls + stage * STAGE);
      dv2* Lb = Lf + NF;
#pragma unroll
      for (int k = 0; k < RF; ++k) {
        const int e = tid + 256 * k;
        if (e < SPB * 36) Lf[36 + e] = rf[k];
      }
      if (tid < 9) Lf[tid] = rx;  // slot of "site -1", direction 0
#pragma unroll
      for (int k = 0; k < RBM; ++k) {
        const int e = tid + 256 * k;
        if (e < SPB * 9) {
          Lb[e] = rb1[k];
          Lb[SPB * 9 + e] = rb2[k];
          if (!carry_u3) Lb[2 * SPB * 9 + e] = rb3[k];
        }
      }
    }
    __syncthreads();
#ifdef BCG_HOP4_TRACE
    // drift study: time stamp (100 MHz) and x3 of every tile-step of every block, in the Gram scratch buffer
    if (tid == 0 && !GRAM && partials) {
      long long* tr = reinterpret_cast<long long*>(partials) + static_cast<int64_t>(blockIdx.x) * 4096;
      const int n = trace_n++;
      if (n < 2048) {
        tr[2 * n] = wall_clock64();
        tr[2 * n + 1] = g.site0;
      }
    }
#endif
    const TileGeom cur = g;
    const bool cur_carry_u3 = carry_u3;
    have = next_tile(g);
    if (have) {  // prefetch the next tile's links; they land while this tile computes
      carry_u3 = CARRY_U3 && g.x3 > 0 && g.site0 - S3 == cur.site0;
      BCG_FETCH_LINKS(g, carry_u3)
    }
    // With the carry three stages, not two: the previous tile's image is READ here (U_3 carry) while a faster wave of this
    // block may already be parking the next tile; with three stages that park goes to the third image, and an
    // image is only re-written after the barrier that follows every wave's last read of it.
    const double2* Lf = Ls + stage * STAGE;
    const double2* Lb = Lf + NF;
    const double2* Lf_prev = Ls + (stage == 0 ? STAGES - 1 : stage - 1) * STAGE;  // the previous tile's links
    stage = stage == STAGES - 1 ? 0 : stage + 1;
    // ---- neighbour tiles: uniform over the block except direction 0
    const int64_t site0 = cur.site0;
    const int x0 = cur.x0b + sl;
    const int64_t me = site0 + sl;
    const int f0 = cur.x1 + L1 * (cur.x2 + L2 * cur.x3);          // face index of direction 0
    const int f1 = cur.x0b + L0 * (cur.x2 + L2 * cur.x3);          // directions 1..3: index of the tile's first site
    const int f2 = cur.x0b + L0 * (cur.x1 + L1 * cur.x3);
    const int f3 = cur.x0b + L0 * (cur.x1 + L1 * cur.x2);
    // ring addressing: slot of this tile's slice; direction 3 is undivided and ring | L3, so the neighbour slices are
    // the neighbour slots (with wrap-around)
    const int slot = RING ? cur.x3 % win.ring : 0;
    const int64_t base = me - static_cast<int64_t>(cur.x3) * S3;   // site within the slice
    const int64_t mi = RING_IN ? base + static_cast<int64_t>(slot) * S3 : me;  // this site in the input field
    const double2 *nf0, *nb0, *nf1, *nb1, *nf2, *nb2, *nf3, *nb3;
    if (x0 + 1 < L0) nf0 = in + (mi + 1) * 3 * M;
    else if (!sp0) nf0 = in + (mi + 1 - L0) * 3 * M;
    else nf0 = ghost + (gp0 + f0) * 3 * M;
    if (x0 > 0) nb0 = in + (mi - 1) * 3 * M;
    else if (!sp0) nb0 = in + (mi - 1 + L0) * 3 * M;
    else nb0 = ghost + (gm0 + f0) * 3 * M;
#define BCG_NB(MU, XM, LM, SM, SPM, GM, GP, FI, NF_, NB_)                                   \
  {                                                                                         \
    if ((XM) + 1 < (LM)) NF_ = in + (mi + (SM)) * 3 * M;                                    \
    else if (!(SPM)) NF_ = in + (mi - static_cast<int64_t>((LM) - 1) * (SM)) * 3 * M;       \
    else NF_ = ghost + ((GP) + (FI) + sl) * 3 * M;                                          \
    if ((XM) > 0) NB_ = in + (mi - (SM)) * 3 * M;                                           \
    else if (!(SPM)) NB_ = in + (mi + static_cast<int64_t>((LM) - 1) * (SM)) * 3 * M;       \
    else NB_ = ghost + ((GM) + (FI) + sl) * 3 * M;                                          \
  }
    BCG_NB(1, cur.x1, L1, S1, sp1, gm1, gp1, f1, nf1, nb1)
    BCG_NB(2, cur.x2, L2, S2, sp2, gm2, gp2, f2, nf2, nb2)
    if (RING_IN) {
      nf3 = in + (base + static_cast<int64_t>(slot + 1 == win.ring ? 0 : slot + 1) * S3) * 3 * M;
      nb3 = in + (base + static_cast<int64_t>(slot == 0 ? win.ring - 1 : slot - 1) * S3) * 3 * M;
    } else {
      BCG_NB(3, cur.x3, L3, S3, sp3, gm3, gp3, f3, nf3, nb3)
    }
#undef BCG_NB
    // ---- all 24 neighbour loads first, then the arithmetic
    const bool carry_b3 = CARRY_B3 && cur.x3 > 0 && fsite_m2 == cur.site0 - S3;
    const int fsite_now = (cur.x3 + 1 < L3) ? cur.site0 + S3 : (sp3 ? -1 : cur.site0 - (L3 - 1) * S3);
    double2 f[4][3], bk[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      f[0][k] = nf0[k * M + j]; bk[0][k] = nb0[k * M + j];
      f[1][k] = nf1[k * M + j]; bk[1][k] = nb1[k * M + j];
      f[2][k] = nf2[k * M + j]; bk[2][k] = nb2[k * M + j];
      f[3][k] = nf3[k * M + j];
    }
    if (carry_b3) {
#pragma unroll
      for (int k = 0; k < 3; ++k) bk[3][k] = make_double2(h2[k].x, h2[k].y);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) bk[3][k] = nb3[k * M + j];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      h2[k] = h1[k];
      h1[k].x = f[3][k].x;
      h1[k].y = f[3][k].y;
    }
    fsite_m2 = fsite_m1;
    fsite_m1 = fsite_now;
    site_m1 = cur.site0;
    (void)site_m1;
    double2 pv[3];
    const int64_t o0 = me * 3 * M + j;
    const int64_t oo = RING_OUT ? (base + static_cast<int64_t>(slot) * S3) * 3 * M + j : o0;
    if (MODE == HOP_SHIFTED) {
#pragma unroll
      for (int r = 0; r < 3; ++r) pv[r] = NT ? ld_nt(p + o0 + r * M) : p[o0 + r * M];
    }
    double2 acc[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
    const int par1 = x0 + og0, par2 = par1 + cur.x1 + og1, par3 = par2 + cur.x2 + og2;
#pragma unroll
    for (int mu = 0; mu < 4; ++mu) {
      const int par = mu == 0 ? 0 : (mu == 1 ? par1 : (mu == 2 ? par2 : par3));  // x_0 + ... + x_{mu-1}, global
      const double eta = (par & 1) ? -1.0 : 1.0;
      const double2* uf = Lf + (sl + 1) * 36 + mu * 9;
      const double2* ub = mu == 0 ? Lf + sl * 36
                                  : ((mu == 3 && cur_carry_u3) ? Lf_prev + (sl + 1) * 36 + 27 : Lb + ((mu - 1) * SPB + sl) * 9);
      double2 t[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double2 u = uf[k * 3 + r];   // U(r,k)
          t[r].x = fma(u.x, f[mu][k].x, t[r].x); t[r].x = fma(-u.y, f[mu][k].y, t[r].x);
          t[r].y = fma(u.x, f[mu][k].y, t[r].y); t[r].y = fma(u.y, f[mu][k].x, t[r].y);
          const double2 v = ub[r * 3 + k];   // U_b(k,r); subtract conj(v) * psi_b(k)
          t[r].x = fma(-v.x, bk[mu][k].x, t[r].x); t[r].x = fma(-v.y, bk[mu][k].y, t[r].x);
          t[r].y = fma(-v.x, bk[mu][k].y, t[r].y); t[r].y = fma(v.y, bk[mu][k].x, t[r].y);
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        acc[r].x = fma(eta, t[r].x, acc[r].x);
        acc[r].y = fma(eta, t[r].y, acc[r].y);
      }
    }
    double2 tv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (MODE == HOP_PLAIN) tv[r] = make_double2(0.5 * acc[r].x, 0.5 * acc[r].y);
      else tv[r] = make_double2(fma(c0, pv[r].x, -0.5 * acc[r].x), fma(c0, pv[r].y, -0.5 * acc[r].y));
      if (NT) st_nt(out + oo + r * M, tv[r]);  // streamed once: keep it from displacing the patch slices in L2
      else out[oo + r * M] = tv[r];
    }
    if (GRAM) {
#pragma unroll
      for (int r = 0; r < 3; ++r) gram_step<16>(G, &pv[r], &tv[r]);
    }
  }
#undef BCG_FETCH_LINKS
  if (GRAM) gram_block_store<16, NW>(G, smem, partials, tid);
}

// ---------------------------------------------------------------------------------------------------
// k_hop4c: the patch walk of k_hop4 written as what it is -- every block owns one (x0 tile, x1, x2) column of its
// patch and sweeps x3 -- so that nothing per tile needs vector address arithmetic or the mixed-radix counter:
//   * the eight neighbour rows of a tile are wave-uniform byte pointers (SGPR pairs) recomputed from per-column
//     constants by a few scalar instructions; a lane adds one constant 32-bit offset (site-in-tile, rhs), i.e.
//     `global_load_dwordx4 v, v_off, s[base:base+1] offset:colour` with no VALU in front of it;
//   * the x0 neighbours use the same row pointer shifted by one site; at the ends of a row the edge lane's offset
//     is bent to the periodic image (a per-column choice of the offset register), or patched from the ghost face;
//   * the blocks of an XCD are paced along x3 (HopWalk::sync) so that the slices they share stay in its L2.
// Requirements (checked by the launcher, else k_hop4 runs): patch walk with one block per tile of a patch slice.
// The per-tile instruction count drops from ~1100 to ~600 per wave; a wave issues at most one instruction every
// four cycles, so at 2 waves per SIMD that count, not the arithmetic, was the floor of k_hop4.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 ld_sv(const char* __restrict__ sbase, unsigned voff, int imm) {
  return *reinterpret_cast<const double2*>(sbase + voff + imm);
}

template <int M, int MODE, bool GRAM, int CLS, bool RING>
__device__ __forceinline__ void hop4c_body(const LatticeDev& lat, const double2* __restrict__ U,
                                           const double2* __restrict__ Ughost, const double2* __restrict__ in,
                                           const double2* __restrict__ ghost, double2* __restrict__ out,
                                           const double2* __restrict__ p, double c0, double2* __restrict__ partials,
                                           const HopWalk& hw, const HopWindow& win) {
  static_assert(!GRAM || M == 16 || M == 8, "fused Gram accumulation: lane&15 = rhs index (m = 16) or (site parity, rhs) (m = 8)");
  constexpr bool RING_OUT = RING && MODE == HOP_PLAIN;
  constexpr bool RING_IN = RING && MODE == HOP_SHIFTED;
  constexpr int SPW = 64 / M;
  constexpr int SPB = 4 * SPW;
  constexpr int NW = 4;
  constexpr int NF = (SPB + 1) * 36;
  constexpr int NB = 3 * SPB * 9;
  constexpr int STAGE = NF + NB;
  constexpr int RF = (SPB * 36 + 255) / 256;
  constexpr int RBM = (SPB * 9 + 255) / 256;
  constexpr int RB = 3 * M * 16;  // bytes of one site row of a field
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double2* Ls = reinterpret_cast<double2*>(smem);  // [2][STAGE] link images
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sl = wave * SPW + lane / M;
  const int j = lane % M;
  const unsigned voff = static_cast<unsigned>((sl * 3 * M + j) * 16);  // this lane's byte offset inside a tile row
  const int L0 = lat.L[0], L1 = lat.L[1], L2 = lat.L[2], L3 = lat.L[3];
  const int S1 = L0, S2 = L0 * L1, S3 = L0 * L1 * L2;
  const int sp0 = lat.split[0], sp1 = lat.split[1], sp2 = lat.split[2], sp3 = lat.split[3];
  const int gm0 = static_cast<int>(lat.ghost_off[0][0]), gp0 = static_cast<int>(lat.ghost_off[0][1]);
  const int gm1 = static_cast<int>(lat.ghost_off[1][0]), gp1 = static_cast<int>(lat.ghost_off[1][1]);
  const int gm2 = static_cast<int>(lat.ghost_off[2][0]), gp2 = static_cast<int>(lat.ghost_off[2][1]);
  const int gm3 = static_cast<int>(lat.ghost_off[3][0]), gp3 = static_cast<int>(lat.ghost_off[3][1]);
  const int og0 = lat.origin[0], og1 = lat.origin[1], og2 = lat.origin[2];
  const char* const inb = reinterpret_cast<const char*>(in);
  const char* const ghb = reinterpret_cast<const char*>(ghost);
  GramAcc<16> G;
  if (GRAM) gram_zero(G);

  // ---- this block's column inside a patch, and the patches of its XCD class
  const int r0 = hw.p0 / SPB, r1 = hw.p1, r4 = L0 / hw.p0, r5 = L1 / hw.p1, r6 = L2 / hw.p2;
  const int cls = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int d0 = idx % r0, d1 = (idx / r0) % r1, d2 = idx / (r0 * r1);
  const int ppc = (r4 * r5 * r6) >> 3;
  const int x3_end = win.x3_lo + win.x3_n;

  dv2 rf[RF], rx, rb1[RBM], rb2[RBM], rb3[RBM];
  rx = dv2{0.0, 0.0};
#define BCG_FETCH_LINKS(g)                                                                                        \
  {                                                                                                               \
    fetch_fwd<SPB, RF>(U + static_cast<int64_t>((g).site0) * 36, tid, rf);                                        \
    if (tid < 9) {                                                                                                \
      const double2* src;                                                                                         \
      if ((g).x0b > 0) src = U + (static_cast<int64_t>((g).site0) - 1) * 36;                                      \
      else if (!sp0) src = U + (static_cast<int64_t>((g).site0) + L0 - 1) * 36;                                   \
      else src = Ughost + (static_cast<int64_t>(gm0) + ((g).x1 + L1 * ((g).x2 + L2 * (g).x3))) * 9;               \
      rx = ld_link(reinterpret_cast<const dv2*>(src + tid));                                                      \
    }                                                                                                             \
    fetch_back<SPB, RBM>(1, (g).x1, L1, S1, sp1, gm1, (g).x0b + L0 * ((g).x2 + L2 * (g).x3), (g).site0, U, Ughost, tid, rb1); \
    fetch_back<SPB, RBM>(2, (g).x2, L2, S2, sp2, gm2, (g).x0b + L0 * ((g).x1 + L1 * (g).x3), (g).site0, U, Ughost, tid, rb2); \
    fetch_back<SPB, RBM>(3, (g).x3, L3, S3, sp3, gm3, (g).x0b + L0 * ((g).x1 + L1 * (g).x2), (g).site0, U, Ughost, tid, rb3); \
  }
#ifdef BCG_HOP4C_STAMPS  // diagnostic build (tools/hop_stamps.py): where a tile's cycles go, summed per block by wave 0
  long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_amdgcn_s_memtime();
#define BCG_STAMP(i)                                   \
  {                                                    \
    __builtin_amdgcn_sched_barrier(0);                 \
    const long long t_ = __builtin_amdgcn_s_memtime(); \
    seg[i] += t_ - tlast;                              \
    tlast = t_;                                        \
    __builtin_amdgcn_sched_barrier(0);                 \
  }
#else
#define BCG_STAMP(i)
#endif
  int stage = 0;
  bool pace = true;   // thread 0: still pacing against the other blocks of the XCD class
  const unsigned zero_rt = static_cast<unsigned>(hw.sync_window) >> 30;  // 0, unknown to the compiler (read_counter)
  unsigned seen1 = 0, seen2 = 0;   // thread 0: counters read one and two processed tiles ago ...
  int seen1_idx = -1, seen2_idx = -1;  // ... and which tile numbers they belong to
  const unsigned per = gridDim.x >> 3;
  // Tile classes (CLS 1: interior only, 2: boundary only; see k_hop4).  Every block still walks every tile number of
  // its class's sequence and counts the skipped ones as done, so the pacing counters keep their meaning.

  for (int pk = 0; pk < ppc; ++pk) {
    const int pi = cls * ppc + pk;
    const int d4 = pi % r4, d5 = (pi / r4) % r5, d6 = pi / (r4 * r5);
    const int x0b = (d4 * r0 + d0) * SPB, x1 = d5 * hw.p1 + d1, x2 = d6 * hw.p2 + d2;
    const int col = x0b + L0 * (x1 + L1 * x2);  // site of the column at x3 = 0
    const bool col_bnd = (sp0 && (x0b == 0 || x0b + SPB == L0)) || (sp1 && (x1 == 0 || x1 == L1 - 1)) ||
                         (sp2 && (x2 == 0 || x2 == L2 - 1));
    auto wanted = [&](int x3) -> bool {
      if (CLS == 0) return true;
      return (col_bnd || (sp3 && (x3 == 0 || x3 == L3 - 1))) == (CLS == 2);
    };
    const int vs0 = pk * win.x3_n - win.x3_lo;  // tile number of slice x3 in this block's sequence: vs0 + x3
    // ---- per-column constants of the six in-slice neighbour rows: site at x3 = 0, site stride per slice, buffer.
    // Field rows move by S3 per slice (by ring slots in capacity mode), ghost rows by the face's x3 stride.
#define BCG_COL_DIR(XM, LM, SM, SPM, GMN, GPL, FIDX, FSTR, AF, SF, KF, AB, SB, KB)                         \
  int AF, SF, AB, SB;                                                                                     \
  bool KF, KB;                                                                                            \
  if ((XM) + 1 < (LM)) { AF = col + (SM); SF = S3; KF = false; }                                          \
  else if (!(SPM)) { AF = col - ((LM) - 1) * (SM); SF = S3; KF = false; }                                 \
  else { AF = (GPL) + (FIDX); SF = (FSTR); KF = true; }                                                   \
  if ((XM) > 0) { AB = col - (SM); SB = S3; KB = false; }                                                 \
  else if (!(SPM)) { AB = col + ((LM) - 1) * (SM); SB = S3; KB = false; }                                 \
  else { AB = (GMN) + (FIDX); SB = (FSTR); KB = true; }
    BCG_COL_DIR(x1, L1, S1, sp1, gm1, gp1, x0b + L0 * x2, L0 * L2, a_f1, s_f1, k_f1, a_b1, s_b1, k_b1)
    BCG_COL_DIR(x2, L2, S2, sp2, gm2, gp2, x0b + L0 * x1, L0 * L1, a_f2, s_f2, k_f2, a_b2, s_b2, k_b2)
#undef BCG_COL_DIR
    // direction 0: the row itself, shifted by one site; at the row ends the edge lane is bent to the periodic image
    const bool row_end = x0b + SPB == L0, row_start = x0b == 0;
    const bool gh0p = row_end && sp0, gh0m = row_start && sp0;
    int shift_p = 1, shift_m = -1;  // sites
    unsigned voff_p = voff, voff_m = voff;
    if (row_end) {
      if (!sp0) { shift_p = 1 - L0; voff_p = (sl == SPB - 1) ? voff : voff + static_cast<unsigned>(L0) * RB; }
      else voff_p = (sl == SPB - 1) ? voff - RB : voff;   // edge lane: any valid row, replaced from the ghost face
    }
    if (row_start) {
      if (!sp0) voff_m = (sl == 0) ? voff + static_cast<unsigned>(L0) * RB : voff;
      else voff_m = (sl == 0) ? voff + RB : voff;
    }
    int slot = RING ? win.x3_lo % win.ring : 0;
    TileGeom g;
    g.x0b = x0b; g.x1 = x1; g.x2 = x2; g.x3 = win.x3_lo; g.site0 = col + win.x3_lo * S3; g.edge = 0;
    int links_for = -1;  // slice whose links are in the prefetch registers
    for (int x3 = win.x3_lo; x3 < x3_end; ++x3) {
      const int step_n = vs0 + x3;
      if (!wanted(x3)) {  // not this launch's tile: count it as done
        if (hw.sync != nullptr && tid == 0 && step_n < hw.sync_stride)
          __hip_atomic_fetch_add(hw.sync + cls * hw.sync_stride + step_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (RING) slot = slot + 1 == win.ring ? 0 : slot + 1;
        continue;
      }
      if (links_for != x3) {  // first tile of a run: not overlapped
        g.x3 = x3;
        g.site0 = col + x3 * S3;
        BCG_FETCH_LINKS(g)
      }
      {  // park the links fetched for this tile in the current LDS stage
        dv2* Lf = reinterpret_cast<dv2*>(Ls + stage * STAGE);
        dv2* Lb = Lf + NF;
#pragma unroll
        for (int k = 0; k < RF; ++k) {
          const int e = tid + 256 * k;
          if (e < SPB * 36) Lf[36 + e] = rf[k];
        }
        if (tid < 9) Lf[tid] = rx;
#pragma unroll
        for (int k = 0; k < RBM; ++k) {
          const int e = tid + 256 * k;
          if (e < SPB * 9) {
            Lb[e] = rb1[k];
            Lb[SPB * 9 + e] = rb2[k];
            Lb[2 * SPB * 9 + e] = rb3[k];
          }
        }
      }
      if (hw.sync != nullptr && tid == 0 && pace) {  // pacing: all blocks of this XCD class within `sync_window` slices
        const int need = step_n - hw.sync_window;
        const unsigned known = seen2_idx == need ? seen2 : (seen1_idx == need ? seen1 : 0u);
        if (need >= 0 && known < per) {  // the count read ahead was short (or there is none): poll
          unsigned* ctr = hw.sync + cls * hw.sync_stride + need;
          const long long t0 = wall_clock64();
          while (read_counter(ctr, zero_rt) < per) {
            if (wall_clock64() - t0 > hw.sync_limit) {
              pace = false;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
        }
      }
      BCG_STAMP(0)  // links parked, pacing wait
      __syncthreads();
      BCG_STAMP(1)  // barrier
#define BCG_PREFETCH_LINKS                                                                        \
  {                                                                                               \
    int nx = x3 + 1; /* the next tile of this launch in the column: its links are parked first thing there */ \
    while (CLS != 0 && nx < x3_end && !wanted(nx)) ++nx;                                          \
    if (nx < x3_end) {                                                                            \
      g.x3 = nx;                                                                                  \
      g.site0 = col + nx * S3;                                                                    \
      BCG_FETCH_LINKS(g)                                                                          \
      links_for = nx;                                                                             \
    }                                                                                             \
  }
      BCG_PREFETCH_LINKS
      const double2* Lf = Ls + stage * STAGE;
      const double2* Lb = Lf + NF;
      stage ^= 1;
      // ---- the eight neighbour rows of this tile: scalar pointers
      const int xf = RING_IN ? slot : x3;                       // where slice x3 of `in` lives
      const int xfp = RING_IN ? (slot + 1 == win.ring ? 0 : slot + 1) : x3 + 1;
      const int xfm = RING_IN ? (slot == 0 ? win.ring - 1 : slot - 1) : x3 - 1;
#define BCG_ROW(A, S, K) ((K) ? ghb + static_cast<int64_t>((A) + x3 * (S)) * RB : inb + static_cast<int64_t>((A) + xf * (S)) * RB)
      const char* const crow = inb + static_cast<int64_t>(col + xf * S3) * RB;
      const char* const q_f0 = crow + static_cast<int64_t>(shift_p) * RB;
      const char* const q_b0 = crow + static_cast<int64_t>(shift_m) * RB;
      const char* const q_f1 = BCG_ROW(a_f1, s_f1, k_f1);
      const char* const q_b1 = BCG_ROW(a_b1, s_b1, k_b1);
      const char* const q_f2 = BCG_ROW(a_f2, s_f2, k_f2);
      const char* const q_b2 = BCG_ROW(a_b2, s_b2, k_b2);
#undef BCG_ROW
      const char* q_f3;
      const char* q_b3;
      if (x3 + 1 < L3) q_f3 = inb + static_cast<int64_t>(col + xfp * S3) * RB;
      else if (!sp3) q_f3 = inb + static_cast<int64_t>(col) * RB;  // slice 0 (slot 0: ring | L3)
      else q_f3 = ghb + static_cast<int64_t>(gp3 + col) * RB;
      if (x3 > 0) q_b3 = inb + static_cast<int64_t>(col + xfm * S3) * RB;
      else if (!sp3) q_b3 = inb + static_cast<int64_t>(col + (RING_IN ? win.ring - 1 : L3 - 1) * S3) * RB;
      else q_b3 = ghb + static_cast<int64_t>(gm3 + col) * RB;
      double2 f[4][3], bk[4][3];
      // loads and arithmetic are issued direction by direction, one direction ahead: at most two directions' rows
      // (12 loads) are in flight per wave, which keeps the L2 working set of the resident blocks small
      const char* const qf[4] = {q_f0, q_f1, q_f2, q_f3};
      const char* const qb[4] = {q_b0, q_b1, q_b2, q_b3};
#define BCG_LOAD_DIR(MU)                                                                   \
  _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                          \
    f[MU][k] = ld_sv(qf[MU], (MU) == 0 ? voff_p : voff, k * M * 16);                       \
    bk[MU][k] = ld_sv(qb[MU], (MU) == 0 ? voff_m : voff, k * M * 16);                      \
  }
      BCG_LOAD_DIR(0)
      if (gh0p || gh0m) {  // direction 0 divided over ranks: the edge site of an end-of-row tile reads the ghost face
        const int64_t f0 = x1 + L1 * (x2 + L2 * x3);
        if (gh0p && sl == SPB - 1) {
#pragma unroll
          for (int k = 0; k < 3; ++k) f[0][k] = ghost[(gp0 + f0) * 3 * M + k * M + j];
        }
        if (gh0m && sl == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) bk[0][k] = ghost[(gm0 + f0) * 3 * M + k * M + j];
        }
      }
      const int64_t crow_site = static_cast<int64_t>(col) + static_cast<int64_t>(x3) * S3;
      const char* const prow = reinterpret_cast<const char*>(p) + crow_site * RB;
      char* const orow = reinterpret_cast<char*>(out) + (RING_OUT ? static_cast<int64_t>(col + slot * S3) : crow_site) * RB;
      double2 pv[3];
      // pacing: read now the counter the tile after next is checked against (see k_hop4)
      seen2 = seen1;
      seen2_idx = seen1_idx;
      if (hw.sync != nullptr && tid == 0 && pace && step_n + 2 >= hw.sync_window) {
        seen1_idx = step_n + 2 - hw.sync_window;
        seen1 = read_counter(hw.sync + cls * hw.sync_stride + seen1_idx, zero_rt);
      }
      BCG_STAMP(2)  // link prefetch and direction-0 loads issued
      double2 acc[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
      const int x0 = x0b + sl;
      const int par1 = x0 + og0, par2 = par1 + x1 + og1, par3 = par2 + x2 + og2;
#pragma unroll
      for (int mu = 0; mu < 4; ++mu) {
        if (mu == 0) { BCG_LOAD_DIR(1) }
        if (mu == 1) { BCG_LOAD_DIR(2) }
        if (mu == 2) {
          // Vector memory returns in issue order, so what misses the L2 goes last: the new slice (+x3), then p.
          // (The next tile's links also miss, but issued this late they are not back when that tile parks them: 15.4 ms.)
          BCG_LOAD_DIR(3)
          if (MODE == HOP_SHIFTED) {
#pragma unroll
            for (int r = 0; r < 3; ++r) pv[r] = ld_nt(reinterpret_cast<const double2*>(prow + voff + r * M * 16));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int par = mu == 0 ? 0 : (mu == 1 ? par1 : (mu == 2 ? par2 : par3));
        const double eta = (par & 1) ? -1.0 : 1.0;
        const double2* uf = Lf + (sl + 1) * 36 + mu * 9;
        const double2* ub = mu == 0 ? Lf + sl * 36 : Lb + ((mu - 1) * SPB + sl) * 9;
        double2 t[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const double2 u = uf[k * 3 + r];
            t[r].x = fma(u.x, f[mu][k].x, t[r].x); t[r].x = fma(-u.y, f[mu][k].y, t[r].x);
            t[r].y = fma(u.x, f[mu][k].y, t[r].y); t[r].y = fma(u.y, f[mu][k].x, t[r].y);
            const double2 v = ub[r * 3 + k];
            t[r].x = fma(-v.x, bk[mu][k].x, t[r].x); t[r].x = fma(-v.y, bk[mu][k].y, t[r].x);
            t[r].y = fma(-v.x, bk[mu][k].y, t[r].y); t[r].y = fma(v.y, bk[mu][k].x, t[r].y);
          }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          acc[r].x = fma(eta, t[r].x, acc[r].x);
          acc[r].y = fma(eta, t[r].y, acc[r].y);
        }
        BCG_STAMP(3 + mu)  // direction mu: wait for its rows + arithmetic (+ issue of the next direction's loads)
      }
      double2 tv[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (MODE == HOP_PLAIN) tv[r] = make_double2(0.5 * acc[r].x, 0.5 * acc[r].y);
        else tv[r] = make_double2(fma(c0, pv[r].x, -0.5 * acc[r].x), fma(c0, pv[r].y, -0.5 * acc[r].y));
        st_nt(reinterpret_cast<double2*>(orow + voff + r * M * 16), tv[r]);
      }
      if (GRAM) {
#pragma unroll
        for (int r = 0; r < 3; ++r) gram_step<16>(G, &pv[r], &tv[r]);
      }
      if (hw.sync != nullptr && tid == 0 && step_n < hw.sync_stride)
        __hip_atomic_fetch_add(hw.sync + cls * hw.sync_stride + step_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (RING) slot = slot + 1 == win.ring ? 0 : slot + 1;
      BCG_STAMP(7)  // p wait, stores, Gram step, counter
    }
  }
#undef BCG_FETCH_LINKS
#ifdef BCG_HOP4C_STAMPS
  if (!GRAM && (tid & 63) == 0) {
    double* o = reinterpret_cast<double*>(partials) + (static_cast<int64_t>(blockIdx.x) * 4 + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = static_cast<double>(seg[i]);
  }
#endif
#undef BCG_STAMP
  if (GRAM) {
    if (M == 8) gram_block_store_fold8<NW>(G, smem, partials, tid, hw.fold.out != nullptr);  // two sites per 16-lane row: see the fold
    else gram_block_store<16, NW>(G, smem, partials, tid, hw.fold.out != nullptr);
    gram_fold<M * M>(hw.fold, partials, tid, NW * 64);
  }
}

// Entry points.  The interior-class fused-Gram variant (and those of m = 8) need 260-280 VGPRs as scheduled by default,
// which would leave one wave per SIMD; they are compiled for two.  The same cap on the other variants changes their instruction schedule and
// costs 3 ms at 64^4 (measured), so it is applied only there.
template <int M, int MODE, bool GRAM, int CLS, bool RING>
__global__ void __launch_bounds__(256) k_hop4c(LatticeDev lat, const double2* __restrict__ U,
                                               const double2* __restrict__ Ughost, const double2* __restrict__ in,
                                               const double2* __restrict__ ghost, double2* __restrict__ out,
                                               const double2* __restrict__ p, double c0,
                                               double2* __restrict__ partials, HopWalk hw, HopWindow win) {
  hop4c_body<M, MODE, GRAM, CLS, RING>(lat, U, Ughost, in, ghost, out, p, c0, partials, hw, win);
}
template <int M, int MODE, bool GRAM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_hop4c_interior(LatticeDev lat, const double2* __restrict__ U, const double2* __restrict__ Ughost,
                 const double2* __restrict__ in, const double2* __restrict__ ghost, double2* __restrict__ out,
                 const double2* __restrict__ p, double c0, double2* __restrict__ partials, HopWalk hw, HopWindow win) {
  hop4c_body<M, MODE, GRAM, 1, false>(lat, U, Ughost, in, ghost, out, p, c0, partials, hw, win);
}


// ---------------------------------------------------------------------------------------------------
// k_hop4b: the column sweep over BUNDLES of 2 x 2 columns.
//
// Counters of k_hop4c at 64^4, m = 16 (tools/pmc_tcp_sq.sh): 138 GB pass through the L1s per launch, 58 % of it served
// there and 61 % of the rest by the L2; the waves are parked on memory for 63 % of their cycles; and the time does not
// follow the traffic (an x3 carry through LDS cut the fabric reads by 15 % and changed nothing).  With one row of
// SPB consecutive x0 sites per block, every site costs 8 neighbour-row loads and all re-use is left to the caches.
// Here a block's tile is SPW x 2 x 2 sites -- wave w owns the SPW consecutive x0 sites of row (x1 + (w & 1), x2 + (w >> 1))
// -- and the rows a tile's sites share are loaded once and exchanged through LDS:
//   * a wave loads ONE new row per step, its own +x3 row (with one halo site either side), uses it as the +x3
//     neighbour and parks it in an LDS slot, where it serves at the next step as the wave's own x0 neighbours (the same
//     row shifted by a site) and as the x1 / x2 neighbours of the other three waves, and at the step after that as the
//     wave's own -x3 neighbour (read back just before the slot is overwritten);
//   * only the two rows that leave the bundle (one in x1, one in x2) are fetched besides: 2 + 2/SPW + 1 + 1 row loads per
//     site instead of 8 (4.5 at m = 16), 19 vector-memory instructions per wave and tile instead of 31;
//   * links are staged per wave for its own sites and parked at the END of the step, behind the last use of loaded
//     data and in front of the output stores (see there);
//   * backward links U_mu(x - mu) are NOT fetched again where the bundle already holds them: U_3(x - 3) is the wave's own
//     forward link of the previous step and is carried from the old link image into the new one when the links are parked
//     (fetched only in a column's prologue); U_1(x - 1) / U_2(x - 2) of a neighbour INSIDE the bundle are read from the
//     partner wave's forward image (SHARE: two images per wave, written one step ahead, so that a faster partner never
//     overwrites the image a slower wave still reads); only the rows that leave the bundle come from global memory;
// Tile order, XCD patches, pacing, x3 windows and ring addressing (capacity mode) and the fused Gram product are those
// of k_hop4c; the interior / boundary tile classes of the split halo exchange stay with k_hop4c.  Same arithmetic per site
// and the same order of the four directions, so results are bit-identical to k_hop4c.
// ---------------------------------------------------------------------------------------------------
// Two link images per wave (partner waves read each other's forward links) where two blocks per CU still fit the 160 KB
// of LDS: m = 16 (73.7 KB per block) and m = 32 (70 KB); at m = 8 (two images: 100 KB) one image, own links only.
__host__ __device__ constexpr bool hop4b_share_images(int m) { return m >= 16; }

template <int M, int MODE, bool GRAM, bool RING, bool CB = false>
__device__ __forceinline__ void hop4b_body(const LatticeDev& lat, const double2* __restrict__ U,
                                           const double2* __restrict__ Ughost, const double2* __restrict__ in,
                                           const double2* __restrict__ ghost, double2* __restrict__ out,
                                           const double2* __restrict__ p, double c0, double2* __restrict__ partials,
                                           const HopWalk& hw, const HopWindow& win) {
  static_assert(!GRAM || M == 16 || M == 8, "fused Gram accumulation: lane&15 = rhs index (m = 16) or (site parity, rhs) (m = 8)");
  constexpr bool RING_OUT = RING && MODE == HOP_PLAIN;
  constexpr bool RING_IN = RING && MODE == HOP_SHIFTED;
  constexpr bool RESID = MODE == HOP_RESID;  // no output: the Gram product of (c0 p - D in - b) with itself, b passed as `out`
  static_assert(!RESID || (GRAM && !RING), "the residual form accumulates a Gram product and is not ring-addressed");
  // CB (checkerboard, half-volume fields -- kernels_generic.hip "Half-volume fields"): `in` holds the sites of one parity,
  // `out` / `p` those of the other, both in the compact order, and `lat` is the COMPACT lattice (L[0] = half the row).  A
  // row (x1, x2, x3) of the output has r = (x1 + x2 + x3 + parity of out) & 1 and its compact site k is x0 = 2 k + r; the
  // input row at the same (x1, x2, x3) holds x0 = 2 k + 1 - r.  Neighbours in directions 1..3 keep k; in direction 0 the
  // forward one is input site k + r, the backward one k - 1 + r.  Links stay in the full-lattice layout: a wave's sites are
  // every other site of a full row, no backward link is another output site's forward link (so all four directions'
  // backward links are gathered, none carried or shared), and U_0(x - 0) is gathered like the others.  Undivided lattices.
  static_assert(!CB || (!RING && !RESID), "checkerboard form: whole-field launches only");
  constexpr int SPW = 64 / M;              // sites per wave = tile extent in x0
  constexpr int NW = 4;
  constexpr int CS = (SPW + 2) * 3 * M;    // one wave's row slot: halo site, SPW sites, halo site (complex numbers)
  constexpr int NFW = (SPW + 1) * 36;      // link image of a wave: slot of the site to the left (U_0 only), forward links
  constexpr int NBW = (CB ? 4 : 3) * SPW * 9;  // ... backward links, directions 1..3 (CB: and direction 0, behind them)
  constexpr int LSTAGE = NFW + NBW;
  constexpr int RFW = (SPW * 36 + 63) / 64;
  constexpr int RBK = (SPW * 9 + 63) / 64;
  constexpr bool SHARE = hop4b_share_images(M);   // two link images per wave (LDS-DMA into the one not being read)
  constexpr bool PARTNER = SHARE && !CB;           // in-bundle backward links from the partner waves, U_3(x - 3) carried
  constexpr bool PIPE = BCG_HOP4B_PIPE != 0 && hop4b_share_images(M) && !RESID;  // the software-pipelined step (below; the checkerboard form too)
  constexpr int HB = 3 * M * 16;                  // bytes of one site = of one halo site
  constexpr int NHD = (HB + 1023) / 1024;         // DMA instructions per halo site
  constexpr int RB = 3 * M * 16;           // bytes of one site of a field
  constexpr int NH = (2 * M + 63) / 64;    // halo loads per lane and colour (1)
  static_assert(NH == 1, "halo sites fit one wave instruction per colour");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int e1 = wave & 1, e2 = wave >> 1;
  dv2* const Cbase = reinterpret_cast<dv2*>(smem);                       // [2 slots][4 waves][CS]
  // Link images.  !SHARE: one per wave -- the next step's links wait in registers and are parked at the end of the step,
  // behind the wave's last read of the image (LDS operations of a wave are in order).  SHARE: image x3 & 1 is read in
  // step x3 by its wave AND by the two partner waves; the links of step x3 + 1 go into the other image, whose last
  // readers (step x3 - 1) are behind the barrier at the top of step x3.
  dv2* const Lbase = Cbase + 2 * NW * CS;
  auto image = [&](int x3, int w) __attribute__((always_inline)) { return Lbase + ((SHARE ? (x3 & 1) : 0) * NW + w) * LSTAGE; };
  const int sw = lane / M, j = lane % M;
  const unsigned voff = static_cast<unsigned>((sw * 3 * M + j) * 16);   // byte offset of (site sw, colour 0, rhs j) in a row
  const int co = (sw + 1) * 3 * M + j;                                   // the same element in a row slot (colour c: + c*M)
  // halo element of this lane (lanes < 2M): site hs (0 left, 1 right), rhs hj
  const bool halo_lane = lane < 2 * M;
  const int hs = (lane / M) & 1, hj = lane % M;
  const int ho = (hs ? (SPW + 1) * 3 * M : 0) + hj;
  const int L0 = lat.L[0], L1 = lat.L[1], L2 = lat.L[2], L3 = lat.L[3];
  const int S1 = L0, S2 = L0 * L1, S3 = L0 * L1 * L2;
  const int sp0 = lat.split[0], sp1 = lat.split[1], sp2 = lat.split[2], sp3 = lat.split[3];
  const int gm0 = static_cast<int>(lat.ghost_off[0][0]), gp0 = static_cast<int>(lat.ghost_off[0][1]);
  const int gm1 = static_cast<int>(lat.ghost_off[1][0]), gp1 = static_cast<int>(lat.ghost_off[1][1]);
  const int gm2 = static_cast<int>(lat.ghost_off[2][0]), gp2 = static_cast<int>(lat.ghost_off[2][1]);
  const int gm3 = static_cast<int>(lat.ghost_off[3][0]), gp3 = static_cast<int>(lat.ghost_off[3][1]);
  const int og0 = lat.origin[0], og1 = lat.origin[1], og2 = lat.origin[2];
  const char* const inb = reinterpret_cast<const char*>(in);
  const char* const ghb = reinterpret_cast<const char*>(ghost);
  GramAcc<16> G;
  if (GRAM) gram_zero(G);

  // ---- this block's tile inside a patch, and the patches of its XCD class
  const int r0 = hw.p0 / SPW, r1 = hw.p1 / 2, r4 = L0 / hw.p0, r5 = L1 / hw.p1, r6 = L2 / hw.p2;
  const int cls = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int d0 = idx % r0, d1 = (idx / r0) % r1, d2 = idx / (r0 * r1);
  const int ppc = (r4 * r5 * r6) >> 3;
  const int x3_end = win.x3_lo + win.x3_n;

  dv2 rf[RFW], rx, rb1[RBK], rb2[RBK], rb3[RBK];
  rx = dv2{0.0, 0.0};
  const unsigned fo = static_cast<unsigned>(lane) * 16;
  // byte offset of backward-link element e = lane + 64 k (site e / 9, entry e % 9) in a row of 36-entry site records
  // (bo_f) and in a packed ghost face (bo_g).  Scalars, not arrays: a wave-uniform choice between two array elements made
  // the compiler keep both arrays in scratch and select between their ADDRESSES.
  static_assert(RBK <= 2, "backward-link elements per lane");
  const unsigned bo_f0 = static_cast<unsigned>((lane / 9) * 36 + lane % 9) * 16;
  const unsigned bo_f1 = static_cast<unsigned>(((lane + 64) / 9) * 36 + (lane + 64) % 9) * 16;
  const unsigned bo_g0 = static_cast<unsigned>(lane) * 16, bo_g1 = static_cast<unsigned>(lane + 64) * 16;
  const unsigned bo_d0 = bo_g0 - bo_f0, bo_d1 = bo_g1 - bo_f1;
#define BO_F(k) ((k) == 0 ? bo_f0 : bo_f1)
  // offset for a ghost face (G true) or a field row: written as "bo_f + (G ? delta : 0)" -- a select between the two
  // variables themselves is turned into a select between their addresses, which keeps them in scratch
#define BO_SEL(G, k) (BO_F(k) + ((G) ? ((k) == 0 ? bo_d0 : bo_d1) : 0u))
#ifdef BCG_HOP4B_STAMPS  // diagnostic build (tools/hop_stamps.py 4b): where a step's cycles go, summed per wave
  long long seg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_amdgcn_s_memtime();
#define BCG_STAMPB(i)                                  \
  {                                                    \
    __builtin_amdgcn_sched_barrier(0);                 \
    const long long t_ = __builtin_amdgcn_s_memtime(); \
    seg[i] += t_ - tlast;                              \
    tlast = t_;                                        \
    __builtin_amdgcn_sched_barrier(0);                 \
  }
#else
#define BCG_STAMPB(i)
#endif
  bool pace = true;   // thread 0: still pacing against the other blocks of the XCD class
  const unsigned zero_rt = static_cast<unsigned>(hw.sync_window) >> 30;  // 0, unknown to the compiler (read_counter)
  unsigned seen1 = 0, seen2 = 0;
  int seen1_idx = -1, seen2_idx = -1;
  const unsigned per = gridDim.x >> 3;

  for (int pk = 0; pk < ppc; ++pk) {
    int d4, d5, d6;
    if (hw.super) {  // super-patch pk, this class's corner of it
      const int h4 = r4 >> 1, h5 = r5 >> 1;
      d4 = 2 * (pk % h4) + (cls & 1);
      d5 = 2 * ((pk / h4) % h5) + ((cls >> 1) & 1);
      d6 = 2 * (pk / (h4 * h5)) + (cls >> 2);
    } else {
      const int pi = cls * ppc + pk;
      d4 = pi % r4;
      d5 = (pi / r4) % r5;
      d6 = pi / (r4 * r5);
    }
    const int x0b = (d4 * r0 + d0) * SPW, x1 = d5 * hw.p1 + d1 * 2 + e1, x2 = d6 * hw.p2 + d2 * 2 + e2;
    const int col = x0b + L0 * (x1 + L1 * x2);  // this wave's first site at x3 = 0
    const int vs0 = pk * win.x3_n - win.x3_lo;  // tile number of slice x3 in this block's sequence: vs0 + x3
    // ---- the row of this wave that leaves the bundle in direction 1 (forward if e1, else backward) and in direction 2:
    // site at x3 = 0, site stride per slice, field (false) or ghost face (true) -- as in k_hop4c
#define BCG_OUT_DIR(FWD, XM, LM, SM, SPM, GMN, GPL, FIDX, FSTR, AO, SO, KO)                                \
  int AO, SO;                                                                                             \
  bool KO;                                                                                                \
  if (FWD) {                                                                                              \
    if ((XM) + 1 < (LM)) { AO = col + (SM); SO = S3; KO = false; }                                        \
    else if (!(SPM)) { AO = col - ((LM) - 1) * (SM); SO = S3; KO = false; }                               \
    else { AO = (GPL) + (FIDX); SO = (FSTR); KO = true; }                                                 \
  } else {                                                                                                \
    if ((XM) > 0) { AO = col - (SM); SO = S3; KO = false; }                                               \
    else if (!(SPM)) { AO = col + ((LM) - 1) * (SM); SO = S3; KO = false; }                               \
    else { AO = (GMN) + (FIDX); SO = (FSTR); KO = true; }                                                 \
  }
    BCG_OUT_DIR(e1, x1, L1, S1, sp1, gm1, gp1, x0b + L0 * x2, L0 * L2, a_o1, s_o1, k_o1)
    BCG_OUT_DIR(e2, x2, L2, S2, sp2, gm2, gp2, x0b + L0 * x1, L0 * L1, a_o2, s_o2, k_o2)
#undef BCG_OUT_DIR
    // backward LINK rows of directions 1, 2 (U_mu(x - mu) lives at the backward neighbour site, whichever way the
    // psi row of that direction comes in)
#define BCG_BACK_DIR(XM, LM, SM, SPM, GMN, FIDX, FSTR, AB, SB, KB)                                         \
  int AB, SB;                                                                                             \
  bool KB;                                                                                                \
  if ((XM) > 0) { AB = col - (SM); SB = S3; KB = false; }                                                 \
  else if (!(SPM)) { AB = col + ((LM) - 1) * (SM); SB = S3; KB = false; }                                 \
  else { AB = (GMN) + (FIDX); SB = (FSTR); KB = true; }
    BCG_BACK_DIR(x1, L1, S1, sp1, gm1, x0b + L0 * x2, L0 * L2, a_b1, s_b1, k_b1)
    BCG_BACK_DIR(x2, L2, S2, sp2, gm2, x0b + L0 * x1, L0 * L1, a_b2, s_b2, k_b2)
#undef BCG_BACK_DIR
    const bool row_end = x0b + SPW == L0, row_start = x0b == 0;

    auto fetch_links = [&](int x3, bool first) __attribute__((always_inline)) {
      const int64_t sw0 = static_cast<int64_t>(col) + static_cast<int64_t>(x3) * S3;  // the wave's first site
      const char* const fsrc = reinterpret_cast<const char*>(U) + sw0 * (36 * 16);
      const char* lsrc;  // U_0 of the site to the left of the wave's first site
      if (!row_start) lsrc = reinterpret_cast<const char*>(U) + (sw0 - 1) * (36 * 16);
      else if (!sp0) lsrc = reinterpret_cast<const char*>(U) + (sw0 + L0 - 1) * (36 * 16);
      else lsrc = reinterpret_cast<const char*>(Ughost) + (static_cast<int64_t>(gm0) + (x1 + L1 * (x2 + L2 * x3))) * (9 * 16);
#pragma unroll
      for (int k = 0; k < RFW; ++k)
        if (lane + 64 * k < SPW * 36) rf[k] = ld_link(reinterpret_cast<const dv2*>(fsrc + fo + k * 1024));
      if (lane < 9) rx = ld_link(reinterpret_cast<const dv2*>(lsrc + fo));
      const char* const ub_ = reinterpret_cast<const char*>(U);
      const char* const ug_ = reinterpret_cast<const char*>(Ughost);
      const int64_t n1 = static_cast<int64_t>(a_b1) + static_cast<int64_t>(x3) * s_b1;
      const int64_t n2 = static_cast<int64_t>(a_b2) + static_cast<int64_t>(x3) * s_b2;
      const char* const q1 = k_b1 ? ug_ + n1 * (9 * 16) : ub_ + (n1 * 4 + 1) * (9 * 16);
      const char* const q2 = k_b2 ? ug_ + n2 * (9 * 16) : ub_ + (n2 * 4 + 2) * (9 * 16);
      const char* q3;
      bool k_b3 = false;
      if (x3 > 0) q3 = ub_ + ((sw0 - S3) * 4 + 3) * (9 * 16);
      else if (!sp3) q3 = ub_ + ((sw0 + static_cast<int64_t>(L3 - 1) * S3) * 4 + 3) * (9 * 16);
      else { q3 = ug_ + (static_cast<int64_t>(gm3) + col) * (9 * 16); k_b3 = true; }
#pragma unroll
      for (int k = 0; k < RBK; ++k)
        if (lane + 64 * k < SPW * 9) {
          if (!SHARE || !e1) rb1[k] = ld_link(reinterpret_cast<const dv2*>(q1 + BO_SEL(k_b1, k)));
          if (!SHARE || !e2) rb2[k] = ld_link(reinterpret_cast<const dv2*>(q2 + BO_SEL(k_b2, k)));
          if (first) rb3[k] = ld_link(reinterpret_cast<const dv2*>(q3 + BO_SEL(k_b3, k)));
        }
    };
    // Park the fetched links of slice x3 in image(x3).  first: a column's prologue (U_3(x - 3) fetched); otherwise U_3(x - 3)
    // is U_3 of the wave's own sites in the image of slice x3 - 1, read here before that image (!SHARE: the same one) is
    // overwritten -- LDS operations of a wave execute in order.
    auto park_links = [&](int x3, bool first) __attribute__((always_inline)) {
      dv2* const Lf = image(x3, wave);
      if (!first) {
        const dv2* const Lo = image(x3 - 1, wave);
#pragma unroll
        for (int k = 0; k < RBK; ++k)
          if (lane + 64 * k < SPW * 9) rb3[k] = Lo[36 + 27 + (BO_F(k) >> 4)];
      }
#pragma unroll
      for (int k = 0; k < RFW; ++k)
        if (lane + 64 * k < SPW * 36) Lf[36 + lane + 64 * k] = rf[k];
      if (lane < 9) Lf[lane] = rx;
#pragma unroll
      for (int k = 0; k < RBK; ++k)
        if (lane + 64 * k < SPW * 9) {
          if (!SHARE || !e1) Lf[NFW + lane + 64 * k] = rb1[k];
          if (!SHARE || !e2) Lf[NFW + SPW * 9 + lane + 64 * k] = rb2[k];
          Lf[NFW + 2 * SPW * 9 + lane + 64 * k] = rb3[k];
        }
    };
    // SHARE: the links of slice x3 by LDS-DMA straight into image(x3) -- no staging registers, no ds_write.  Issued at the
    // top of step x3 - 1 (the image's last readers, step x3 - 2, are behind that step's barrier), in FRONT of the step's
    // ordinary loads: those are consumed inside the step, so the DMAs have landed before the wave reaches the next barrier.
    // U_3(x - 3) is not fetched (first: it is, in a column's prologue): park_u3 carries it over.
    auto dma_links = [&](int x3, bool first) __attribute__((always_inline)) {
      const int64_t sw0 = static_cast<int64_t>(col) + static_cast<int64_t>(x3) * S3;
      const char* const fsrc = reinterpret_cast<const char*>(U) + sw0 * (36 * 16);
      const char* lsrc;
      if (!row_start) lsrc = reinterpret_cast<const char*>(U) + (sw0 - 1) * (36 * 16);
      else if (!sp0) lsrc = reinterpret_cast<const char*>(U) + (sw0 + L0 - 1) * (36 * 16);
      else lsrc = reinterpret_cast<const char*>(Ughost) + (static_cast<int64_t>(gm0) + (x1 + L1 * (x2 + L2 * x3))) * (9 * 16);
      const unsigned img = __builtin_amdgcn_readfirstlane(lds_addr_of(image(x3, wave)));
#pragma unroll
      for (int k = 0; k < RFW; ++k)
        if (lane + 64 * k < SPW * 36) glds16_link(fsrc + fo + k * 1024, img + (36 + 64 * k) * 16);
      if (lane < 9) glds16_link(lsrc + fo, img);
      const char* const ub_ = reinterpret_cast<const char*>(U);
      const char* const ug_ = reinterpret_cast<const char*>(Ughost);
      if (!e1) {
        const int64_t n1 = static_cast<int64_t>(a_b1) + static_cast<int64_t>(x3) * s_b1;
        const char* const q1 = k_b1 ? ug_ + n1 * (9 * 16) : ub_ + (n1 * 4 + 1) * (9 * 16);
#pragma unroll
        for (int k = 0; k < RBK; ++k)
          if (lane + 64 * k < SPW * 9) glds16_link(q1 + BO_SEL(k_b1, k), img + (NFW + 64 * k) * 16);
      }
      if (!e2) {
        const int64_t n2 = static_cast<int64_t>(a_b2) + static_cast<int64_t>(x3) * s_b2;
        const char* const q2 = k_b2 ? ug_ + n2 * (9 * 16) : ub_ + (n2 * 4 + 2) * (9 * 16);
#pragma unroll
        for (int k = 0; k < RBK; ++k)
          if (lane + 64 * k < SPW * 9) glds16_link(q2 + BO_SEL(k_b2, k), img + (NFW + SPW * 9 + 64 * k) * 16);
      }
      if (first) {
        const char* q3;
        bool k_b3 = false;
        if (x3 > 0) q3 = ub_ + ((sw0 - S3) * 4 + 3) * (9 * 16);
        else if (!sp3) q3 = ub_ + ((sw0 + static_cast<int64_t>(L3 - 1) * S3) * 4 + 3) * (9 * 16);
        else { q3 = ug_ + (static_cast<int64_t>(gm3) + col) * (9 * 16); k_b3 = true; }
#pragma unroll
        for (int k = 0; k < RBK; ++k)
          if (lane + 64 * k < SPW * 9) glds16_link(q3 + BO_SEL(k_b3, k), img + (NFW + 2 * SPW * 9 + 64 * k) * 16);
      }
    };
    // the same (first = false) from addresses the caller carries along the column instead of recomputing them (INCR below)
    auto dma_links_at = [&](int x3, const char* fsrc, const char* lsrc, const char* q1, const char* q2) __attribute__((always_inline)) {
      const unsigned img = __builtin_amdgcn_readfirstlane(lds_addr_of(image(x3, wave)));
#pragma unroll
      for (int k = 0; k < RFW; ++k)
        if (lane + 64 * k < SPW * 36) glds16_link(fsrc + fo + k * 1024, img + (36 + 64 * k) * 16);
      if (lane < 9) glds16_link(lsrc + fo, img);
      if (!e1) {
#pragma unroll
        for (int k = 0; k < RBK; ++k)
          if (lane + 64 * k < SPW * 9) glds16_link(q1 + BO_SEL(k_b1, k), img + (NFW + 64 * k) * 16);
      }
      if (!e2) {
#pragma unroll
        for (int k = 0; k < RBK; ++k)
          if (lane + 64 * k < SPW * 9) glds16_link(q2 + BO_SEL(k_b2, k), img + (NFW + SPW * 9 + 64 * k) * 16);
      }
    };
    // ... and with the scalar-base form of the DMA (PIPE)
    auto dma_links_at_s = [&](int x3, const char* fsrc, const char* lsrc, const char* q1, const char* q2) __attribute__((always_inline)) {
      const unsigned img = __builtin_amdgcn_readfirstlane(lds_addr_of(image(x3, wave)));
      static_assert(!PIPE || (RFW <= 3 && RBK <= 2), "link DMA instructions per image");
      if (lane < SPW * 36) glds16_s(fsrc, fo, img + 36 * 16);
      if (RFW > 1 && lane + 64 < SPW * 36) glds16_s(fsrc, fo + 1024, img + (36 + 64) * 16);
      if (RFW > 2 && lane + 128 < SPW * 36) glds16_s(fsrc, fo + 2048, img + (36 + 128) * 16);
      if (lane < 9) glds16_s(lsrc, fo, img);
      if (!e1) {
        if (lane < SPW * 9) glds16_s(q1, BO_SEL(k_b1, 0), img + NFW * 16);
        if (RBK > 1 && lane + 64 < SPW * 9) glds16_s(q1, BO_SEL(k_b1, 1), img + (NFW + 64) * 16);
      }
      if (!e2) {
        if (lane < SPW * 9) glds16_s(q2, BO_SEL(k_b2, 0), img + (NFW + SPW * 9) * 16);
        if (RBK > 1 && lane + 64 < SPW * 9) glds16_s(q2, BO_SEL(k_b2, 1), img + (NFW + SPW * 9 + 64) * 16);
      }
    };
    // CB: the links of the wave's SPW output sites of slice x3 -- every other site of a full-lattice row -- by LDS-DMA into
    // image(x3): forward links (one 36-entry record per site, the records 2 apart), then the backward links of all four
    // directions, each U_mu of the full-lattice site x - mu (periodic, or across a divided direction 1..3 from the gauge
    // ghost, which keeps the full-lattice face numbering: its offsets are twice the half faces' that `lat` carries).
    auto dma_links_cb = [&](int x3) __attribute__((always_inline)) {
      const int L0f = 2 * L0;
      const int rr = (x1 + x2 + x3 + win.cb_parity) & 1;           // x0 = 2 k + rr on this row
      const int64_t rowf = static_cast<int64_t>(L0f) * (x1 + static_cast<int64_t>(L1) * (x2 + static_cast<int64_t>(L2) * x3));
      const int xf0 = 2 * x0b + rr;                                  // full x0 of the wave's first site
      const char* const ub_ = reinterpret_cast<const char*>(U);
      const unsigned img = __builtin_amdgcn_readfirstlane(lds_addr_of(image(x3, wave)));
      // forward: element e = lane + 64 k -> site e / 36, entry e % 36
#pragma unroll
      for (int k = 0; k < RFW; ++k) {
        const int e = lane + 64 * k;
        if (e < SPW * 36)
          glds16_link(ub_ + (rowf + xf0 + 2 * (e / 36)) * (36 * 16) + (e % 36) * 16, img + (36 + 64 * k) * 16);
      }
      // backward, direction mu: element e -> site s = e / 9, entry e % 9 of U_mu at the full site of x - mu
      const int x1m = x1 > 0 ? x1 - 1 : L1 - 1, x2m = x2 > 0 ? x2 - 1 : L2 - 1, x3m = x3 > 0 ? x3 - 1 : L3 - 1;
      const int64_t row1 = static_cast<int64_t>(L0f) * (x1m + static_cast<int64_t>(L1) * (x2 + static_cast<int64_t>(L2) * x3));
      const int64_t row2 = static_cast<int64_t>(L0f) * (x1 + static_cast<int64_t>(L1) * (x2m + static_cast<int64_t>(L2) * x3));
      const int64_t row3 = static_cast<int64_t>(L0f) * (x1 + static_cast<int64_t>(L1) * (x2 + static_cast<int64_t>(L2) * x3m));
      // per direction: base of the row of 9-entry records the backward links come from, and the record stride in entries
      // (a field row: 36-entry site records, U_mu at entry 9 mu; a ghost row: 9-entry records) -- wave-uniform
      const char* const ug_ = reinterpret_cast<const char*>(Ughost);
      const bool g1 = sp1 && x1 == 0, g2 = sp2 && x2 == 0, g3 = sp3 && x3 == 0;
      const char* const b1 = g1 ? ug_ + (2 * static_cast<int64_t>(gm1) + static_cast<int64_t>(L0f) * (x2 + static_cast<int64_t>(L2) * x3)) * (9 * 16)
                                : ub_ + (row1 * 4 + 1) * (9 * 16);
      const char* const b2 = g2 ? ug_ + (2 * static_cast<int64_t>(gm2) + static_cast<int64_t>(L0f) * (x1 + static_cast<int64_t>(L1) * x3)) * (9 * 16)
                                : ub_ + (row2 * 4 + 2) * (9 * 16);
      const char* const b3 = g3 ? ug_ + (2 * static_cast<int64_t>(gm3) + static_cast<int64_t>(L0f) * (x1 + static_cast<int64_t>(L1) * x2)) * (9 * 16)
                                : ub_ + (row3 * 4 + 3) * (9 * 16);
      const int t1 = g1 ? 9 * 16 : 36 * 16, t2 = g2 ? 9 * 16 : 36 * 16, t3 = g3 ? 9 * 16 : 36 * 16;
#pragma unroll
      for (int k = 0; k < RBK; ++k) {
        const int e = lane + 64 * k;
        if (e < SPW * 9) {
          const int xs = xf0 + 2 * (e / 9);
          const unsigned eo = (e % 9) * 16;
          glds16_link(b1 + static_cast<int64_t>(xs) * t1 + eo, img + (NFW + 64 * k) * 16);
          glds16_link(b2 + static_cast<int64_t>(xs) * t2 + eo, img + (NFW + SPW * 9 + 64 * k) * 16);
          glds16_link(b3 + static_cast<int64_t>(xs) * t3 + eo, img + (NFW + 2 * SPW * 9 + 64 * k) * 16);
          const int xm = xs > 0 ? xs - 1 : L0f - 1;
          glds16_link(ub_ + ((rowf + xm) * 4 + 0) * (9 * 16) + eo, img + (NFW + 3 * SPW * 9 + 64 * k) * 16);
        }
      }
    };
    // U_3(x - 3) of slice x3 = U_3 of the wave's own sites in image(x3 - 1): copied into image(x3)
    auto park_u3 = [&](int x3) __attribute__((always_inline)) {
      const dv2* const Lo = image(x3 - 1, wave);
      dv2* const Ln = image(x3, wave);
#pragma unroll
      for (int k = 0; k < RBK; ++k)
        if (lane + 64 * k < SPW * 9) Ln[NFW + 2 * SPW * 9 + lane + 64 * k] = Lo[36 + 27 + (BO_F(k) >> 4)];
    };
    // Row `xs` of this wave's column as stored (slice index, or ring slot), with its two halo sites, `gx3` the slice's true
    // index (ghost faces keep whole-lattice indexing): pointers of the own sites and, per lane, of the halo site.
    // kind: 0 a slice of `in`, 1 the +x3 ghost face, 2 the -x3 ghost face (own sites only; no halo is needed from those).
    auto row_ptrs = [&](int kind, int xs, int gx3, const char*& own, const char*& hal) __attribute__((always_inline)) {
      if (kind == 0) {
        own = inb + (static_cast<int64_t>(col) + static_cast<int64_t>(xs) * S3) * RB;
        const int64_t f0i = x1 + L1 * (x2 + static_cast<int64_t>(L2) * gx3);
        const char* lft;
        const char* rgt;
        if (!row_start) lft = own - RB;
        else if (!sp0) lft = own + static_cast<int64_t>(L0 - 1) * RB;
        else lft = ghb + (static_cast<int64_t>(gm0) + f0i) * RB;
        if (!row_end) rgt = own + static_cast<int64_t>(SPW) * RB;
        else if (!sp0) rgt = own - static_cast<int64_t>(L0 - SPW) * RB;
        else rgt = ghb + (static_cast<int64_t>(gp0) + f0i) * RB;
        hal = (hs ? rgt : lft) + hj * 16;
      } else {
        own = ghb + (static_cast<int64_t>(kind == 1 ? gp3 : gm3) + col) * RB;
        hal = own + hj * 16;  // any valid address: this row is never used as a centre row
      }
    };
    // the same row for the DMA form: base pointers of the own sites and of the left / right halo site
    auto row_ptrs3 = [&](int kind, int xs, int gx3, const char*& own, const char*& lft, const char*& rgt) __attribute__((always_inline)) {
      if (kind == 0) {
        own = inb + (static_cast<int64_t>(col) + static_cast<int64_t>(xs) * S3) * RB;
        const int64_t f0i = x1 + L1 * (x2 + static_cast<int64_t>(L2) * gx3);
        if (!row_start) lft = own - RB;
        else if (!sp0) lft = own + static_cast<int64_t>(L0 - 1) * RB;
        else lft = ghb + (static_cast<int64_t>(gm0) + f0i) * RB;
        if (!row_end) rgt = own + static_cast<int64_t>(SPW) * RB;
        else if (!sp0) rgt = own - static_cast<int64_t>(L0 - SPW) * RB;
        else rgt = ghb + (static_cast<int64_t>(gp0) + f0i) * RB;
      } else {
        own = ghb + (static_cast<int64_t>(kind == 1 ? gp3 : gm3) + col) * RB;
        lft = rgt = own;  // any valid address: this row is never used as a centre row
      }
    };
    // which stored slice / ghost face is slice g of the column (g may be -1 or L3: periodic image or ghost)
    auto slice_of = [&](int g, int slot_g, int& kind, int& xs, int& gx3) __attribute__((always_inline)) {
      if (g >= 0 && g < L3) { kind = 0; xs = RING_IN ? slot_g : g; gx3 = g; }
      else if (g < 0) {
        if (!sp3) { kind = 0; xs = RING_IN ? win.ring - 1 : L3 - 1; gx3 = L3 - 1; }
        else { kind = 2; xs = 0; gx3 = 0; }
      } else {
        if (!sp3) { kind = 0; xs = 0; gx3 = 0; }  // slice 0 (ring slot 0: ring | L3)
        else { kind = 1; xs = 0; gx3 = 0; }
      }
    };

    // ---- column prologue: slices x3_lo (centre of the first step) and x3_lo - 1 (its -x3 neighbour) into the row slots
    __syncthreads();  // every wave has left the previous column (its last step reads both slots)
    int slot = RING ? win.x3_lo % win.ring : 0;
    {
      const int lo = win.x3_lo;
      int kind, xs, gx3;
      const char* own;
      const char* hal;
      slice_of(lo, slot, kind, xs, gx3);
      row_ptrs(kind, xs, gx3, own, hal);
      dv2* const Cc = Cbase + ((lo & 1) * NW + wave) * CS;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Cc[co + c * M] = *reinterpret_cast<const dv2*>(own + voff + c * M * 16);
        if (halo_lane) Cc[ho + c * M] = *reinterpret_cast<const dv2*>(hal + c * M * 16);
      }
      slice_of(lo - 1, RING ? (slot == 0 ? win.ring - 1 : slot - 1) : 0, kind, xs, gx3);
      row_ptrs(kind, xs, gx3, own, hal);
      dv2* const Cm = Cbase + (((lo + 1) & 1) * NW + wave) * CS;
#pragma unroll
      for (int c = 0; c < 3; ++c) Cm[co + c * M] = *reinterpret_cast<const dv2*>(own + voff + c * M * 16);
      if (CB) {
        dma_links_cb(lo);
        asm volatile("s_waitcnt vmcnt(0)");
      } else if (SHARE) {
        dma_links(lo, true);
        asm volatile("s_waitcnt vmcnt(0)");  // landed before the first step's barrier (hipcc does not count them)
      } else {
        fetch_links(lo, true);
        park_links(lo, true);
      }
    }
    auto row_o = [&](bool k_o, int a_o, int s_o, int x3v, int slotv) __attribute__((always_inline)) {
      return k_o ? ghb + (static_cast<int64_t>(a_o) + static_cast<int64_t>(x3v) * s_o) * RB
                 : inb + (static_cast<int64_t>(a_o) + static_cast<int64_t>(RING_IN ? slotv : x3v) * s_o) * RB;
    };
    // INCR: every address a step needs is linear in x3 along a column (whole-lattice addressing, no ring, no checkerboard):
    // the link rows of slice x3 + 1, the +x3 row with its halo sites (until the column's last slice, which wraps or is a
    // ghost face), the two rows that leave the bundle, p and out.  They are set up once per column and advanced by their
    // strides; recomputing them cost ~250 scalar instructions (64-bit multiplies, wave-uniform branches, SGPR reloads) per
    // step, as many as the step has FMAs.
    // In capacity mode the ring-addressed side (the output of the plain hop, the input rows of the shifted one) keeps the
    // closed form; links, p and the other side are carried.
    constexpr bool INCR = SHARE && !CB;  // (PIPE below builds on it)
    constexpr bool INCR_IN = INCR && !RING_IN, INCR_OUT = INCR && !RING_OUT;
    const int64_t id_f = static_cast<int64_t>(S3) * (36 * 16), id_row = static_cast<int64_t>(S3) * RB;
    const char* ik_f = nullptr; const char* ik_l = nullptr; const char* ik_1 = nullptr; const char* ik_2 = nullptr;
    const char* ir_own = nullptr; const char* ir_lft = nullptr; const char* ir_rgt = nullptr;
    const char* iw_own = nullptr; const char* iw_lft = nullptr; const char* iw_rgt = nullptr;
    const char* io_1 = nullptr; const char* io_2 = nullptr; const char* ip_p = nullptr; char* ip_o = nullptr;
    int64_t id_l = 0, id_1 = 0, id_2 = 0, id_lft = 0, id_rgt = 0, id_o1 = 0, id_o2 = 0;
    if (INCR) {
      const int lo = win.x3_lo, xn = lo + 1;
      const char* const ub_ = reinterpret_cast<const char*>(U);
      const char* const ug_ = reinterpret_cast<const char*>(Ughost);
      const int64_t sw0 = static_cast<int64_t>(col) + static_cast<int64_t>(xn) * S3;
      const int64_t f0i = x1 + L1 * (x2 + static_cast<int64_t>(L2) * xn);
      const int64_t face = static_cast<int64_t>(L1) * L2;
      ik_f = ub_ + sw0 * (36 * 16);
      if (!row_start) { ik_l = ik_f - 36 * 16; id_l = id_f; }
      else if (!sp0) { ik_l = ik_f + static_cast<int64_t>(L0 - 1) * (36 * 16); id_l = id_f; }
      else { ik_l = ug_ + (static_cast<int64_t>(gm0) + f0i) * (9 * 16); id_l = face * (9 * 16); }
      {
        const int64_t n1 = static_cast<int64_t>(a_b1) + static_cast<int64_t>(xn) * s_b1;
        ik_1 = k_b1 ? ug_ + n1 * (9 * 16) : ub_ + (n1 * 4 + 1) * (9 * 16);
        id_1 = static_cast<int64_t>(s_b1) * (k_b1 ? 9 * 16 : 36 * 16);
        const int64_t n2 = static_cast<int64_t>(a_b2) + static_cast<int64_t>(xn) * s_b2;
        ik_2 = k_b2 ? ug_ + n2 * (9 * 16) : ub_ + (n2 * 4 + 2) * (9 * 16);
        id_2 = static_cast<int64_t>(s_b2) * (k_b2 ? 9 * 16 : 36 * 16);
      }
      if (INCR_IN) ir_own = inb + sw0 * RB;
      if (!INCR_IN) {}
      else if (!row_start) { ir_lft = ir_own - RB; id_lft = id_row; }
      else if (!sp0) { ir_lft = ir_own + static_cast<int64_t>(L0 - 1) * RB; id_lft = id_row; }
      else { ir_lft = ghb + (static_cast<int64_t>(gm0) + f0i) * RB; id_lft = face * RB; }
      if (!INCR_IN) {}
      else if (!row_end) { ir_rgt = ir_own + static_cast<int64_t>(SPW) * RB; id_rgt = id_row; }
      else if (!sp0) { ir_rgt = ir_own - static_cast<int64_t>(L0 - SPW) * RB; id_rgt = id_row; }
      else { ir_rgt = ghb + (static_cast<int64_t>(gp0) + f0i) * RB; id_rgt = face * RB; }
      if (INCR_IN) {  // the +x3 row of the column's last slice (slice 0 again, or the +x3 ghost face): set up here so that the
        int kind, xs, gx3;  // sweep itself holds none of the lattice's extents, splits and ghost offsets
        slice_of(L3, 0, kind, xs, gx3);
        row_ptrs3(kind, xs, gx3, iw_own, iw_lft, iw_rgt);
      }
      if (INCR_IN) {
        io_1 = row_o(k_o1, a_o1, s_o1, lo, 0);
        io_2 = row_o(k_o2, a_o2, s_o2, lo, 0);
      }
      id_o1 = static_cast<int64_t>(s_o1) * RB;
      id_o2 = static_cast<int64_t>(s_o2) * RB;
      const int64_t c0s = (static_cast<int64_t>(col) + static_cast<int64_t>(lo) * S3) * RB;
      ip_p = reinterpret_cast<const char*>(p) + c0s;
      if (INCR_OUT) ip_o = reinterpret_cast<char*>(out) + c0s;
    }
    double2 o1[3], o2[3];
    // ---- PIPE: the software-pipelined step -------------------------------------------------------------------------------
    // What bounded the form below (profiles/r03_stencil_*): every wave of a block issued its ~21 vector-memory instructions
    // -- 6 of them LDS-DMAs, the costliest to issue -- in one burst right behind the step's barrier (2.6 k of a step's 9.7 k
    // cycles), waited for the first of them (1.9 k) and only then computed, with nothing in flight during the arithmetic.
    // Here no global access is seen by hipcc (which would wait for ALL of them at the first use of ANY): they are issued in
    // four small groups between the directions' arithmetic, in the order they are needed, and retired by hand-counted
    // s_waitcnt (vector-memory operations complete in issue order; vmcnt(n) = all but the n youngest are done):
    //   top      B  the +x3 row and its halo sites by LDS-DMA into the row slot (needed by direction 3 and by the next step);
    //               the slot's old contents, the -x3 neighbour, are read into registers first
    //   dir 0 |  E  p (shifted form; needed in the tail)    C  the links of slice x3 + 1 by LDS-DMA (needed after the next barrier)
    //   dir 1 |  D  the two rows that leave the bundle, for step x3 + 1, into registers (needed in the next step)
    //   dir 2
    //   wait B (younger: E, C, D) | dir 3 | wait E (younger: C, D) | output | stores S | wait all but S | barrier
    // so a step's loads have between one and three directions' arithmetic (and the partner wave's) to arrive in.
    // Bit-identical to the other form: the same operands reach the same FMAs in the same order.
    // tools/check_async_regs.py checks in the device code that nothing touches a destination register between issue and wait.
    if constexpr (PIPE) {
      dv2 q1[3], q2[3];  // rows that leave the bundle: this step's (complete)
      {
        const char* const a1 = INCR_IN ? io_1 : row_o(k_o1, a_o1, s_o1, win.x3_lo, slot);
        const char* const a2 = INCR_IN ? io_2 : row_o(k_o2, a_o2, s_o2, win.x3_lo, slot);
        if (INCR_IN) { io_1 += id_o1; io_2 += id_o2; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double2 v1 = ld_sv(a1, voff, c * M * 16), v2 = ld_sv(a2, voff, c * M * 16);
          q1[c] = dv2{v1.x, v1.y};
          q2[c] = dv2{v2.x, v2.y};
        }
        // retired HERE (a load pending at the loop's entry would be waited for at its use in every iteration)
        asm volatile("" : "+v"(q1[0]), "+v"(q1[1]), "+v"(q1[2]), "+v"(q2[0]), "+v"(q2[1]), "+v"(q2[2]));
      }
      constexpr int nB = 3 + 2 * NHD;                               // row DMAs
      constexpr int nE = MODE == HOP_PLAIN ? 0 : 3;                 // p
      // link DMAs of this wave (checkerboard form: the forward links and all four directions' backward links, every wave)
      const int nC = CB ? RFW + 4 * RBK : RFW + 1 + (e1 ? 0 : RBK) + (e2 ? 0 : RBK);
      constexpr int nD = 6, nS = 3;
      constexpr int SPREAD = MODE == HOP_PLAIN ? 0 : BCG_HOP4B_SPREAD;  // 1: rows, 2: links, 4: second next-row group
      constexpr bool SP_B = (SPREAD & 1) != 0, SP_C = (SPREAD & 2) != 0, SP_D = (SPREAD & 4) != 0;
      (void)nB;
      for (int x3 = win.x3_lo; x3 < x3_end; ++x3) {
        const bool more = x3 + 1 < x3_end;
        const int slot_n = RING ? (slot + 1 == win.ring ? 0 : slot + 1) : 0;
        // pacing (thread 0; hipcc sees these few operations and waits for them itself -- at most the previous step's stores
        // are outstanding here, and the counter traffic of the tail is issued behind the end-of-step wait, so none of the
        // hand-counted waits below has one of them among the operations it counts)
        const int step_n = vs0 + x3;
        if (hw.sync != nullptr && tid == 0 && pace) {
          const int need = step_n - hw.sync_window;
          const unsigned known = seen2_idx == need ? seen2 : (seen1_idx == need ? seen1 : 0u);
          if (need >= 0 && known < per) {
            unsigned* ctr = hw.sync + cls * hw.sync_stride + need;
            const long long t0 = wall_clock64();
            while (read_counter(ctr, zero_rt) < per) {
              if (wall_clock64() - t0 > hw.sync_limit) {
                pace = false;
                break;
              }
              __builtin_amdgcn_s_sleep(2);
            }
          }
        }
        BCG_STAMPB(0)   // loop overhead, address bookkeeping, pacing wait of thread 0
        __syncthreads();  // row slot x3 & 1 and link image x3 & 1 (filled during the previous step) are complete
        BCG_STAMPB(1)   // barrier
        const dv2* const Lf = image(x3, wave);
        const dv2* const Lb = Lf + NFW;
        const dv2* const ub1 = (PARTNER && e1) ? image(x3, wave ^ 1) + (sw + 1) * 36 + 9 : Lb + sw * 9;
        const dv2* const ub2 = (PARTNER && e2) ? image(x3, wave ^ 2) + (sw + 1) * 36 + 18 : Lb + (SPW + sw) * 9;
        const int rr = CB ? (x1 + x2 + x3 + win.cb_parity) & 1 : 0;  // CB: this row's x0 = 2 k + rr
        const dv2* const Cc = Cbase + (x3 & 1) * NW * CS;
        dv2* const Cn = Cbase + (((x3 + 1) & 1) * NW + wave) * CS;
        dv2 f0[3], b0[3], b3[3], f3[3], lp1[3], lp2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) b3[c] = Cn[co + c * M];  // -x3 neighbour: the slot's old contents
        asm volatile("" : "+v"(b3[0]), "+v"(b3[1]), "+v"(b3[2]));  // in registers before the DMA below replaces them
        // ---- B: slice x3 + 1 (own sites, halo sites) -> Cn
        const char* b_own;
        const char* b_lft;
        const char* b_rgt;
        {
          const char* own;
          const char* lft;
          const char* rgt;
          if (INCR_IN && x3 + 1 < L3) {
            own = ir_own; lft = ir_lft; rgt = ir_rgt;
            ir_own += id_row; ir_lft += id_lft; ir_rgt += id_rgt;
          } else if (INCR_IN) {
            own = iw_own; lft = iw_lft; rgt = iw_rgt;
          } else {
            int kind, xs, gx3;
            slice_of(x3 + 1, slot_n, kind, xs, gx3);
            row_ptrs3(kind, xs, gx3, own, lft, rgt);
          }
          b_own = own; b_lft = lft; b_rgt = rgt;
        }
        const unsigned cn = __builtin_amdgcn_readfirstlane(lds_addr_of(Cn));
        static_assert(NHD <= 2, "halo site: at most two DMA instructions");
        // the row's DMAs in issue order: own sites (3 KB at every width), then the halo sites
#define BCG_ROW_PIECE(I)                                                                 \
  {                                                                                      \
    if ((I) == 0) glds16_s(b_own, fo, cn + HB);                                          \
    if ((I) == 1) glds16_s(b_own, fo + 1024, cn + HB + 1024);                            \
    if ((I) == 2) glds16_s(b_own, fo + 2048, cn + HB + 2048);                            \
    if ((I) == 3) {                                                                      \
      if (lane * 16 < HB) {                                                              \
        glds16_s(b_lft, fo, cn);                                                         \
        glds16_s(b_rgt, fo, cn + (SPW + 1) * HB);                                        \
      }                                                                                  \
      if (NHD > 1 && lane * 16 + 1024 < HB) {                                            \
        glds16_s(b_lft, fo + 1024, cn + 1024);                                           \
        glds16_s(b_rgt, fo + 1024, cn + (SPW + 1) * HB + 1024);                          \
      }                                                                                  \
    }                                                                                    \
  }
        BCG_ROW_PIECE(0)
        BCG_ROW_PIECE(1)
        if (!SP_B) { BCG_ROW_PIECE(2) BCG_ROW_PIECE(3) }
        BCG_STAMPB(2)   // -x3 read back, row DMAs issued
        // neighbours inside the bundle, from the row slots of this slice
        const dv2* const Cown = Cc + wave * CS;
        const dv2* const Cp1 = Cc + (wave ^ 1) * CS;
        const dv2* const Cp2 = Cc + (wave ^ 2) * CS;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          f0[c] = Cown[co + (CB ? rr : 1) * 3 * M + c * M];      // CB: input sites k + rr and k - 1 + rr
          b0[c] = Cown[co + (CB ? rr - 1 : -1) * 3 * M + c * M];
        }
        const int64_t crow_site = static_cast<int64_t>(col) + static_cast<int64_t>(x3) * S3;
        const char* const prow = INCR ? ip_p : reinterpret_cast<const char*>(p) + crow_site * RB;
        char* const orow = INCR_OUT ? ip_o : reinterpret_cast<char*>(out) + (RING_OUT ? static_cast<int64_t>(col) + static_cast<int64_t>(slot) * S3 : crow_site) * RB;
        if (INCR) ip_p += id_row;
        if (INCR_OUT) ip_o += id_row;
        dv2 pv[3], n1[3], n2[3];
        double2 acc[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
        const int x0 = CB ? 2 * (x0b + sw) + rr : x0b + sw;
        const int par1 = x0 + og0, par2 = par1 + x1 + og1, par3 = par2 + x2 + og2;
// -DBCG_HOP4B_BCAST=0 (A/B build): every lane reads every link entry from the images, the six a unit (mu, k) of 24 FMAs needs
// right in front of it.  (Round 4's first pipelined build read them a unit ahead in the plain form -- 48 registers of staging;
// the broadcast form below needs neither.)  The order of the FMAs along each accumulation chain is the same in all forms:
// bit-identical results.
#define BCG_LD(K, UF, UB)                                                                                    \
  {                                                                                                          \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                          \
      LU[r] = (UF)[(K) * 3 + r];                                                                             \
      LU[3 + r] = (UB)[r * 3 + (K)];                                                                         \
    }                                                                                                        \
  }
// BCAST: unit (MU, K), output colour R -- forward entry 9 MU + 3 K + R of the site's 36, backward entry 9 MU + 3 R + K (LQ below)
#define BCG_FM1(MU, K, R, F, B)                                                                              \
  {                                                                                                          \
    constexpr int ef_ = 9 * (MU) + 3 * (K) + (R), eb_ = 9 * (MU) + 3 * (R) + (K);                            \
    fmac_bcast<ef_ % 16, false>(t[R].x, LQ[ef_ / 16].x, F[K].x);                                             \
    fmac_bcast<ef_ % 16, true>(t[R].x, LQ[ef_ / 16].y, F[K].y);                                              \
    fmac_bcast<ef_ % 16, false>(t[R].y, LQ[ef_ / 16].x, F[K].y);                                             \
    fmac_bcast<ef_ % 16, false>(t[R].y, LQ[ef_ / 16].y, F[K].x);                                             \
    fmac_bcast<eb_ % 16, true>(t[R].x, LQ[3 + eb_ / 16].x, B[K].x);                                          \
    fmac_bcast<eb_ % 16, true>(t[R].x, LQ[3 + eb_ / 16].y, B[K].y);                                          \
    fmac_bcast<eb_ % 16, true>(t[R].y, LQ[3 + eb_ / 16].x, B[K].y);                                          \
    fmac_bcast<eb_ % 16, false>(t[R].y, LQ[3 + eb_ / 16].y, B[K].x);                                         \
  }
#define BCG_FMB(MU, K, F, B) { BCG_FM1(MU, K, 0, F, B) BCG_FM1(MU, K, 1, F, B) BCG_FM1(MU, K, 2, F, B) }
#define BCG_FM(K, F, B)                                                                                      \
  {                                                                                                          \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                          \
      const dv2 u = LU[r];                                                                                   \
      t[r].x = fma(u.x, F[K].x, t[r].x); t[r].x = fma(-u.y, F[K].y, t[r].x);                                 \
      t[r].y = fma(u.x, F[K].y, t[r].y); t[r].y = fma(u.y, F[K].x, t[r].y);                                  \
      const dv2 v = LU[3 + r];                                                                               \
      t[r].x = fma(-v.x, B[K].x, t[r].x); t[r].x = fma(-v.y, B[K].y, t[r].x);                                \
      t[r].y = fma(-v.x, B[K].y, t[r].y); t[r].y = fma(v.y, B[K].x, t[r].y);                                 \
    }                                                                                                        \
  }
// direction MU with forward neighbour F, backward neighbour B, link pointers UF / UB (per-lane reads only); EXTRA: LDS
// reads the next direction needs, issued under this one's last unit; V0, V1: vector-memory instructions issued behind the
// first and the second unit (SPREAD), pinned by scheduling barriers
#define BCG_PIPE_DIR(MU, F, B, UF, UB, EXTRA, V0, V1)                                                        \
  {                                                                                                          \
    const int par = (MU) == 0 ? 0 : ((MU) == 1 ? par1 : ((MU) == 2 ? par2 : par3));                         \
    const double eta = (par & 1) ? -1.0 : 1.0;                                                               \
    double2 t[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};                             \
    if (BCAST) {                                                                                             \
      BCG_FMB(MU, 0, F, B)                                                                                   \
      BCG_PIPE_SLOT(V0)                                                                                      \
      BCG_FMB(MU, 1, F, B)                                                                                   \
      BCG_PIPE_SLOT(V1)                                                                                      \
      EXTRA                                                                                                  \
      BCG_FMB(MU, 2, F, B)                                                                                   \
    } else {                                                                                                 \
      BCG_LD(0, UF, UB) BCG_FM(0, F, B)                                                                      \
      BCG_PIPE_SLOT(V0)                                                                                      \
      BCG_LD(1, UF, UB) BCG_FM(1, F, B)                                                                      \
      BCG_PIPE_SLOT(V1)                                                                                      \
      EXTRA                                                                                                  \
      BCG_LD(2, UF, UB) BCG_FM(2, F, B)                                                                      \
    }                                                                                                        \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                          \
      acc[r].x = fma(eta, t[r].x, acc[r].x);                                                                 \
      acc[r].y = fma(eta, t[r].y, acc[r].y);                                                                 \
    }                                                                                                        \
  }
#define BCG_PIPE_SLOT(V)                     \
  if (SPREAD) {                              \
    __builtin_amdgcn_sched_barrier(0);       \
    V                                        \
    __builtin_amdgcn_sched_barrier(0);       \
  }
// the partner waves' rows of this slice (x1 / x2 neighbours inside the bundle): read a direction ahead of their use
#define BCG_LD_LP(LP, CP) { _Pragma("unroll") for (int c = 0; c < 3; ++c) LP[c] = (CP)[co + c * M]; }
#define BCG_PIPE_PIN asm volatile("" : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y))
#define BCG_PIPE_APART(text) asm volatile("; " text : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y))
        // (the shifted forms have p in flight besides: with the links a unit ahead they need 256 registers and spill 2-100 -- and a
        //  spilled destination of a hand-waited load is read before it arrives; tools/check_async_regs.py, run by the Makefile)
        constexpr bool BCAST = BCG_HOP4B_BCAST != 0;
        dv2 LU[6];
        const dv2* const uf0 = Lf + (sw + 1) * 36;          // U_mu(x): + 9 mu
        const dv2* const ub0 = CB ? Lb + (3 * SPW + sw) * 9 : Lf + sw * 36;  // U_0(x - 0): the left neighbour's forward link (CB: gathered)
        const dv2* const ub3 = Lb + (2 * SPW + sw) * 9;
        // BCAST: this lane's share of its site's links -- forward entries l + 16 q of the site's 36 (q = 0, 1, 2), backward
        // entries l + 16 q of the 36 = 9 per direction (U_mu(x - mu), mu = 0 .. 3, wherever this step finds them), l = lane & 15;
        // entries past the 36th repeat the last one (never used)
        dv2 LQ[6];
        if (BCAST) {
          const int l16 = lane & 15;
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int e = l16 + 16 * q < 36 ? l16 + 16 * q : 35;
            LQ[q] = uf0[e];
            const int d = e / 9, i = e - 9 * d;
            const dv2* const ub = d == 0 ? ub0 : (d == 1 ? ub1 : (d == 2 ? ub2 : ub3));
            LQ[3 + q] = ub[i];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        BCG_PIPE_DIR(0, f0, b0, uf0, ub0, BCG_LD_LP(lp1, Cp1), if (SP_B) BCG_ROW_PIECE(2), if (SP_B) BCG_ROW_PIECE(3))
        BCG_PIPE_PIN;
        __builtin_amdgcn_sched_barrier(0);
#undef BCG_ROW_PIECE
        BCG_STAMPB(3)   // direction 0
        // ---- E: p;  C: links of slice x3 + 1 -> the other image (the INCR form in three pieces: forward 1, forward 2, the rest)
        if (MODE != HOP_PLAIN) {
          asm volatile("; ASYNC_ISSUE p");
          pv[0] = ld_sv_async<0>(prow, voff);
          pv[1] = ld_sv_async<M * 16>(prow, voff);
          pv[2] = ld_sv_async<2 * M * 16>(prow, voff);
          asm volatile("; ASYNC_ISSUED p");
        }
        constexpr bool C_PIECES = SP_C && INCR && !CB;  // (closed-form / checkerboard link DMAs stay one group)
        const unsigned imgn = __builtin_amdgcn_readfirstlane(lds_addr_of(image(x3 + 1, wave)));
#define BCG_LINK_PIECE(I)                                                                                    \
  if (more) {                                                                                                \
    if ((I) == 0) {                                                                                          \
      if (lane < SPW * 36) glds16_s(ik_f, fo, imgn + 36 * 16);                                               \
    }                                                                                                        \
    if ((I) == 1) {                                                                                          \
      if (RFW > 1 && lane + 64 < SPW * 36) glds16_s(ik_f, fo + 1024, imgn + (36 + 64) * 16);                 \
    }                                                                                                        \
    if ((I) == 2) {                                                                                          \
      if (RFW > 2 && lane + 128 < SPW * 36) glds16_s(ik_f, fo + 2048, imgn + (36 + 128) * 16);               \
      if (lane < 9) glds16_s(ik_l, fo, imgn);                                                                \
    }                                                                                                        \
    if ((I) == 3) {                                                                                          \
      if (!e1) {                                                                                             \
        if (lane < SPW * 9) glds16_s(ik_1, BO_SEL(k_b1, 0), imgn + NFW * 16);                                \
        if (RBK > 1 && lane + 64 < SPW * 9) glds16_s(ik_1, BO_SEL(k_b1, 1), imgn + (NFW + 64) * 16);         \
      }                                                                                                      \
      if (!e2) {                                                                                             \
        if (lane < SPW * 9) glds16_s(ik_2, BO_SEL(k_b2, 0), imgn + (NFW + SPW * 9) * 16);                    \
        if (RBK > 1 && lane + 64 < SPW * 9) glds16_s(ik_2, BO_SEL(k_b2, 1), imgn + (NFW + SPW * 9 + 64) * 16); \
      }                                                                                                      \
      ik_f += id_f; ik_l += id_l; ik_1 += id_1; ik_2 += id_2;                                                \
    }                                                                                                        \
  }
        if (C_PIECES) {
          // (behind the units of directions 1 and 2)
        } else if (more) {
          if (CB) {
            dma_links_cb(x3 + 1);
          } else if (INCR) {
            dma_links_at_s(x3 + 1, ik_f, ik_l, ik_1, ik_2);
            ik_f += id_f; ik_l += id_l; ik_1 += id_1; ik_2 += id_2;
          } else {
            dma_links(x3 + 1, false);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        BCG_STAMPB(4)   // p loads and link DMAs issued
#define BCG_C1 if (C_PIECES) { BCG_LINK_PIECE(0) }
#define BCG_C2 if (C_PIECES) { BCG_LINK_PIECE(1) }
        if (e1) { BCG_PIPE_DIR(1, q1, lp1, uf0 + 9, ub1, BCG_LD_LP(lp2, Cp2), BCG_C1, BCG_C2) BCG_PIPE_APART("direction 1, forward row outside the bundle"); }
        else { BCG_PIPE_DIR(1, lp1, q1, uf0 + 9, ub1, BCG_LD_LP(lp2, Cp2), BCG_C1, BCG_C2) BCG_PIPE_APART("direction 1, backward row outside the bundle"); }
        BCG_PIPE_PIN;
        __builtin_amdgcn_sched_barrier(0);
#undef BCG_C1
#undef BCG_C2
        BCG_STAMPB(5)   // direction 1
        // ---- D: the rows that leave the bundle, for the next step (two groups of three loads; with SPREAD the second one
        // behind direction 2 -- not inside it: the direction's two code paths would each get their own registers for it and
        // a copy where they join, in front of the wait -- and the rest of the link DMAs behind direction 2's units.  The waits
        // only need the rows first and p second: E, C and D may come in any order)
        const char* const a_n1 = INCR_IN ? io_1 : row_o(k_o1, a_o1, s_o1, x3 + 1, slot_n);
        const char* const a_n2 = INCR_IN ? io_2 : row_o(k_o2, a_o2, s_o2, x3 + 1, slot_n);
        if (INCR_IN) { io_1 += id_o1; io_2 += id_o2; }
#define BCG_NEXT_ROW(TAG, N, A)                          \
  if (more) {                                            \
    asm volatile("; ASYNC_ISSUE " TAG);                  \
    N[0] = ld_sv_async<0>(A, voff);                      \
    N[1] = ld_sv_async<M * 16>(A, voff);                 \
    N[2] = ld_sv_async<2 * M * 16>(A, voff);             \
    asm volatile("; ASYNC_ISSUED " TAG);                 \
  }
        BCG_NEXT_ROW("n1", n1, a_n1)
        if (!SP_D) { BCG_NEXT_ROW("n2", n2, a_n2) }
        __builtin_amdgcn_sched_barrier(0);
        BCG_STAMPB(6)   // next rows issued
#define BCG_C3 if (C_PIECES) { BCG_LINK_PIECE(2) }
#define BCG_C4 if (C_PIECES) { BCG_LINK_PIECE(3) }
        if (e2) { BCG_PIPE_DIR(2, q2, lp2, uf0 + 18, ub2, , BCG_C3, BCG_C4) BCG_PIPE_APART("direction 2, forward row outside the bundle"); }
        else { BCG_PIPE_DIR(2, lp2, q2, uf0 + 18, ub2, , BCG_C3, BCG_C4) BCG_PIPE_APART("direction 2, backward row outside the bundle"); }
        BCG_PIPE_PIN;
        __builtin_amdgcn_sched_barrier(0);
        if (SP_D) { BCG_NEXT_ROW("n2", n2, a_n2) }
        __builtin_amdgcn_sched_barrier(0);
#undef BCG_C3
#undef BCG_C4
#undef BCG_LINK_PIECE
#undef BCG_NEXT_ROW
        BCG_STAMPB(7)   // direction 2
        // ---- the +x3 row has landed in Cn once at most E, C and D (and the touches) are outstanding
        wait_vmcnt(nE + (more ? nC + nD : 0));
        BCG_STAMPB(8)   // wait for the +x3 row
#pragma unroll
        for (int c = 0; c < 3; ++c) f3[c] = Cn[co + c * M];
        BCG_PIPE_DIR(3, f3, b3, uf0 + 27, ub3, , , )
        BCG_PIPE_PIN;
        __builtin_amdgcn_sched_barrier(0);
#undef BCG_LD
#undef BCG_FM
#undef BCG_FM1
#undef BCG_FMB
#undef BCG_PIPE_DIR
#undef BCG_PIPE_SLOT
#undef BCG_LD_LP
#undef BCG_PIPE_PIN
#undef BCG_PIPE_APART
        BCG_STAMPB(9)   // direction 3
        if (more && !CB) park_u3(x3 + 1);  // U_3(x - 3) of the next slice = U_3 of this one (ds_read / ds_write; the DMAs fill the rest)
        double2 tv[3], pw[3];
        if (MODE != HOP_PLAIN) {
          wait_vmcnt(more ? nC + nD : 0);  // p
          // The values are handed to hipcc as NEW registers written behind the wait (plain inputs, early-clobber outputs): with
          // the loaded registers as in-out operands of an empty asm the allocator may pick other registers for the operand
          // and copy -- i.e. read -- the loaded ones in FRONT of the wait (it did, for the rows below).
          dv2 r0, r1, r2;
          asm volatile("; ASYNC_RETIRE p\n\tv_mov_b64 %0, %6\n\tv_mov_b64 %1, %7\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %9\n\tv_mov_b64 %4, %10\n\tv_mov_b64 %5, %11"
                       : "=&v"(r0.x), "=&v"(r0.y), "=&v"(r1.x), "=&v"(r1.y), "=&v"(r2.x), "=&v"(r2.y)
                       : "v"(pv[0].x), "v"(pv[0].y), "v"(pv[1].x), "v"(pv[1].y), "v"(pv[2].x), "v"(pv[2].y));
          pw[0] = make_double2(r0.x, r0.y);
          pw[1] = make_double2(r1.x, r1.y);
          pw[2] = make_double2(r2.x, r2.y);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (MODE == HOP_PLAIN) tv[r] = make_double2(0.5 * acc[r].x, 0.5 * acc[r].y);
          else tv[r] = make_double2(fma(c0, pw[r].x, -0.5 * acc[r].x), fma(c0, pw[r].y, -0.5 * acc[r].y));
          st_nt(reinterpret_cast<double2*>(orow + voff + r * M * 16), tv[r]);
        }
        if (GRAM) {
#pragma unroll
          for (int r = 0; r < 3; ++r) gram_step<16>(G, &pw[r], &tv[r]);
        }
        BCG_STAMPB(10)  // U_3 carried, wait for p, output, stores
        // everything but the stores: the links and the next rows have landed before this wave reaches the barrier
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          asm volatile("s_waitcnt vmcnt(%24) ; ASYNC_RETIRE n1 ASYNC_RETIRE n2\n\t"
                       "v_mov_b64 %0, %12\n\tv_mov_b64 %1, %13\n\tv_mov_b64 %2, %14\n\tv_mov_b64 %3, %15\n\tv_mov_b64 %4, %16\n\tv_mov_b64 %5, %17\n\t"
                       "v_mov_b64 %6, %18\n\tv_mov_b64 %7, %19\n\tv_mov_b64 %8, %20\n\tv_mov_b64 %9, %21\n\tv_mov_b64 %10, %22\n\tv_mov_b64 %11, %23"
                       : "=&v"(q1[0].x), "=&v"(q1[0].y), "=&v"(q1[1].x), "=&v"(q1[1].y), "=&v"(q1[2].x), "=&v"(q1[2].y),
                         "=&v"(q2[0].x), "=&v"(q2[0].y), "=&v"(q2[1].x), "=&v"(q2[1].y), "=&v"(q2[2].x), "=&v"(q2[2].y)
                       : "v"(n1[0].x), "v"(n1[0].y), "v"(n1[1].x), "v"(n1[1].y), "v"(n1[2].x), "v"(n1[2].y),
                         "v"(n2[0].x), "v"(n2[0].y), "v"(n2[1].x), "v"(n2[1].y), "v"(n2[2].x), "v"(n2[2].y), "n"(nS)
                       : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" : : "n"(nS) : "memory");
        }
        if (hw.sync != nullptr && tid == 0 && step_n < hw.sync_stride)
          __hip_atomic_fetch_add(hw.sync + cls * hw.sync_stride + step_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen2 = seen1;
        seen2_idx = seen1_idx;
        if (hw.sync != nullptr && tid == 0 && pace && step_n + 2 >= hw.sync_window) {
          seen1_idx = step_n + 2 - hw.sync_window;
          seen1 = read_counter(hw.sync + cls * hw.sync_stride + seen1_idx, zero_rt);
        }
        if (RING) slot = slot_n;
        BCG_STAMPB(11)  // end-of-step wait (links, next rows), pacing counters
      }
    } else
    for (int x3 = win.x3_lo; x3 < x3_end; ++x3) {
      const int step_n = vs0 + x3;
      if (hw.sync != nullptr && tid == 0 && pace) {  // pacing: all blocks of this XCD class within `sync_window` slices
        const int need = step_n - hw.sync_window;
        const unsigned known = seen2_idx == need ? seen2 : (seen1_idx == need ? seen1 : 0u);
        if (need >= 0 && known < per) {
          unsigned* ctr = hw.sync + cls * hw.sync_stride + need;
          const long long t0 = wall_clock64();
          while (read_counter(ctr, zero_rt) < per) {
            if (wall_clock64() - t0 > hw.sync_limit) {
              pace = false;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
        }
      }
      BCG_STAMPB(0)   // pacing wait of thread 0 (the other waves' share of it shows up in the barrier)
      __syncthreads();  // row slot x3 & 1 (written in the previous step) is complete
      BCG_STAMPB(1)   // barrier
      if (x3 + 1 < x3_end) {
        if (CB) dma_links_cb(x3 + 1);
        else if (SHARE && INCR) {
          dma_links_at(x3 + 1, ik_f, ik_l, ik_1, ik_2);
          ik_f += id_f; ik_l += id_l; ik_1 += id_1; ik_2 += id_2;
        }
        else if (SHARE) dma_links(x3 + 1, false);   // into the other image, in front of this step's ordinary loads
        else fetch_links(x3 + 1, false);       // parked at the end of this step
      }
      const dv2* const Lf = image(x3, wave);
      const dv2* const Lb = Lf + NFW;
      // backward links of directions 1, 2: own image (row outside the bundle) or the partner wave's forward links
      const dv2* const ub1 = (PARTNER && e1) ? image(x3, wave ^ 1) + (sw + 1) * 36 + 9 : Lb + sw * 9;
      const dv2* const ub2 = (PARTNER && e2) ? image(x3, wave ^ 2) + (sw + 1) * 36 + 18 : Lb + (SPW + sw) * 9;
      const int rr = CB ? (x1 + x2 + x3 + win.cb_parity) & 1 : 0;  // CB: this row's x0 = 2 k + rr
      const dv2* const Cc = Cbase + (x3 & 1) * NW * CS;       // centre rows of the four waves (this slice)
      dv2* const Cn = Cbase + (((x3 + 1) & 1) * NW + wave) * CS;  // this wave's slot for slice x3 + 1; holds slice x3 - 1
      double2 f[4][3], bk[4][3];
      // -x3 neighbour: this wave's own row of slice x3 - 1, read back before the slot is overwritten below
#pragma unroll
      for (int c = 0; c < 3; ++c) { const dv2 v = Cn[co + c * M]; bk[3][c] = make_double2(v.x, v.y); }
      // ---- global loads of the step: the two rows that leave the bundle, the +x3 row with its halo, p
      {
        const char* const q_o1 = INCR_IN ? io_1 : row_o(k_o1, a_o1, s_o1, x3, slot);
        const char* const q_o2 = INCR_IN ? io_2 : row_o(k_o2, a_o2, s_o2, x3, slot);
        if (INCR_IN) { io_1 += id_o1; io_2 += id_o2; }
#pragma unroll
        for (int c = 0; c < 3; ++c) o1[c] = ld_sv(q_o1, voff, c * M * 16);
#pragma unroll
        for (int c = 0; c < 3; ++c) o2[c] = ld_sv(q_o2, voff, c * M * 16);
      }
      dv2 hv[3];
      {
        int kind, xs, gx3;
        const char* own;
        const char* hal;
        if (INCR_IN && x3 + 1 < L3) {  // a slice of `in`: carried along; the column's last +x3 row wraps or is a ghost face
          own = ir_own;
          hal = (hs ? ir_rgt : ir_lft) + hj * 16;
          ir_own += id_row; ir_lft += id_lft; ir_rgt += id_rgt;
        } else if (INCR_IN && !GRAM) {
          own = iw_own;
          hal = (hs ? iw_rgt : iw_lft) + hj * 16;
        } else {
          slice_of(x3 + 1, RING ? (slot + 1 == win.ring ? 0 : slot + 1) : 0, kind, xs, gx3);
          row_ptrs(kind, xs, gx3, own, hal);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) f[3][c] = ld_sv(own, voff, c * M * 16);
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (halo_lane) hv[c] = *reinterpret_cast<const dv2*>(hal + c * M * 16);
      }
      const int64_t crow_site = static_cast<int64_t>(col) + static_cast<int64_t>(x3) * S3;
      const char* const prow = INCR ? ip_p : reinterpret_cast<const char*>(p) + crow_site * RB;
      char* const orow = INCR_OUT ? ip_o : reinterpret_cast<char*>(out) + (RING_OUT ? static_cast<int64_t>(col) + static_cast<int64_t>(slot) * S3 : crow_site) * RB;
      if (INCR) ip_p += id_row;
      if (INCR_OUT) ip_o += id_row;
      double2 pv[3], bv[3];
      if (MODE != HOP_PLAIN) {
#pragma unroll
        for (int r = 0; r < 3; ++r) pv[r] = ld_nt(reinterpret_cast<const double2*>(prow + voff + r * M * 16));
      }
      if (RESID) {
#pragma unroll
        for (int r = 0; r < 3; ++r) bv[r] = ld_nt(reinterpret_cast<const double2*>(orow + voff + r * M * 16));
      }
      BCG_STAMPB(2)   // issue of the step's DMAs and loads
      // ---- neighbours inside the bundle, from the row slots: x0 (own row shifted by a site), x1 and x2 (partner waves)
      const dv2* const Cown = Cc + wave * CS;
      const dv2* const Cp1 = Cc + (wave ^ 1) * CS;
      const dv2* const Cp2 = Cc + (wave ^ 2) * CS;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // full lattice: sites k + 1 and k - 1 of the own row; CB: input sites k + rr and k - 1 + rr
        const dv2 a = Cown[co + (CB ? rr : 1) * 3 * M + c * M], b = Cown[co + (CB ? rr - 1 : -1) * 3 * M + c * M];
        f[0][c] = make_double2(a.x, a.y);
        bk[0][c] = make_double2(b.x, b.y);
      }
      // Directions 1, 2: one neighbour is the partner wave's row (LDS), the other the row that leaves the bundle (o1 / o2);
      // which is forward depends on the wave (e1, e2).  The two cases are two calls of the direction's arithmetic below,
      // not a per-lane select of 24 registers per step (the compiler turned the former `f[1][c] = e1 ? o1[c] : partner`
      // into 48 v_cndmask per step: e1 is wave-uniform, but a select was cheaper than a branch around three moves).
      // Measured (profiles/r03_stencil_incremental_addresses.txt): plain hop 10.1 -> 9.6 ms; the form with the fused Gram
      // product, whose scalar registers are tighter, 11.7 -> 12.7 with it (100 more SGPR reloads per step), so that form
      // keeps the selects.
      constexpr bool DIRBRANCH = !GRAM;
      double2 lp1[3], lp2[3];
      if (DIRBRANCH) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const dv2 v1 = Cp1[co + c * M], v2 = Cp2[co + c * M];
          lp1[c] = make_double2(v1.x, v1.y);
          lp2[c] = make_double2(v2.x, v2.y);
        }
      } else {
        if (e1) {  // forward row of direction 1 leaves the bundle, backward is the partner's
#pragma unroll
          for (int c = 0; c < 3; ++c) { const dv2 v = Cp1[co + c * M]; f[1][c] = o1[c]; bk[1][c] = make_double2(v.x, v.y); }
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) { const dv2 v = Cp1[co + c * M]; f[1][c] = make_double2(v.x, v.y); bk[1][c] = o1[c]; }
        }
        if (e2) {
#pragma unroll
          for (int c = 0; c < 3; ++c) { const dv2 v = Cp2[co + c * M]; f[2][c] = o2[c]; bk[2][c] = make_double2(v.x, v.y); }
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) { const dv2 v = Cp2[co + c * M]; f[2][c] = make_double2(v.x, v.y); bk[2][c] = o2[c]; }
        }
      }
      double2 acc[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
      const int x0 = CB ? 2 * (x0b + sw) + rr : x0b + sw;
      const int par1 = x0 + og0, par2 = par1 + x1 + og1, par3 = par2 + x2 + og2;
#pragma unroll
      for (int mu = 0; mu < 4; ++mu) {
        __builtin_amdgcn_sched_barrier(0);
        const int par = mu == 0 ? 0 : (mu == 1 ? par1 : (mu == 2 ? par2 : par3));
        const double eta = (par & 1) ? -1.0 : 1.0;
        const dv2* uf = Lf + (sw + 1) * 36 + mu * 9;
        const dv2* ub = mu == 0 ? (CB ? Lb + (3 * SPW + sw) * 9 : Lf + sw * 36)
                                : (mu == 1 ? ub1 : (mu == 2 ? ub2 : Lb + (2 * SPW + sw) * 9));
        // acc += eta (U_mu(x) F - U_mu(x - mu)^dagger B): F the forward neighbour, B the backward one.  A macro, not a lambda:
        // with a lambda the form with the fused product reloaded 130 spilled SGPRs per step instead of 40.
#define BCG_LINK_F(k, r) uf[(k) * 3 + (r)]
#define BCG_LINK_B(k, r, u) ub[(r) * 3 + (k)]
#define BCG_DIR_TERM(F, B)                                                                                   \
  {                                                                                                          \
    double2 t[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};                             \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                          \
      _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                        \
        const dv2 u = BCG_LINK_F(k, r);                                                                      \
        t[r].x = fma(u.x, F[k].x, t[r].x); t[r].x = fma(-u.y, F[k].y, t[r].x);                               \
        t[r].y = fma(u.x, F[k].y, t[r].y); t[r].y = fma(u.y, F[k].x, t[r].y);                                \
        const dv2 v = BCG_LINK_B(k, r, u);                                                                   \
        t[r].x = fma(-v.x, B[k].x, t[r].x); t[r].x = fma(-v.y, B[k].y, t[r].x);                              \
        t[r].y = fma(-v.x, B[k].y, t[r].y); t[r].y = fma(v.y, B[k].x, t[r].y);                               \
      }                                                                                                      \
    }                                                                                                        \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                          \
      acc[r].x = fma(eta, t[r].x, acc[r].x);                                                                 \
      acc[r].y = fma(eta, t[r].y, acc[r].y);                                                                 \
    }                                                                                                        \
  }
        // (the two asm comments differ on purpose: identical tails would be merged again, with selects on the operands)
#define BCG_KEEP_APART(text) asm volatile("; " text : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y))
        if (DIRBRANCH && mu == 1) {
          if (e1) { BCG_DIR_TERM(o1, lp1) BCG_KEEP_APART("direction 1, forward row outside the bundle"); }
          else { BCG_DIR_TERM(lp1, o1) BCG_KEEP_APART("direction 1, backward row outside the bundle"); }
        } else if (DIRBRANCH && mu == 2) {
          if (e2) { BCG_DIR_TERM(o2, lp2) BCG_KEEP_APART("direction 2, forward row outside the bundle"); }
          else { BCG_DIR_TERM(lp2, o2) BCG_KEEP_APART("direction 2, backward row outside the bundle"); }
        } else {
          BCG_DIR_TERM(f[mu], bk[mu])
        }
#undef BCG_DIR_TERM
#undef BCG_LINK_F
#undef BCG_LINK_B
#undef BCG_KEEP_APART
        // pin this direction's arithmetic here: the compiler otherwise sinks FMAs past the branches below, towards the
        // stores, and the link entries they read stay live across them
        asm volatile("" : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y));
#ifdef BCG_HOP4B_STAMPS
        if (mu == 0) BCG_STAMPB(3) else if (mu == 1) BCG_STAMPB(4) else if (mu == 2) BCG_STAMPB(5) else BCG_STAMPB(6)
#endif
      }
      // Every load of this step has been consumed, so parking the next step's links (an `s_waitcnt vmcnt(0)` in front of
      // the LDS writes: the compiler cannot count across the step's branches) drains nothing.  The output stores and the
      // pacing atomics are issued behind it and are never waited for inside the step: k_hop4c parks at the top of the
      // next tile and drains them there, 13 % of its time.
      if (x3 + 1 < x3_end) {
        if (CB) {}  // nothing to carry: every backward link was gathered
        else if (SHARE) park_u3(x3 + 1);
        else park_links(x3 + 1, false);
      }
      // park the +x3 row (own sites and halo) as the next step's centre row
      {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dv2 v;
          v.x = f[3][c].x;
          v.y = f[3][c].y;
          Cn[co + c * M] = v;
          if (halo_lane) Cn[ho + c * M] = hv[c];
        }
      }
      double2 tv[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (MODE == HOP_PLAIN) tv[r] = make_double2(0.5 * acc[r].x, 0.5 * acc[r].y);
        else tv[r] = make_double2(fma(c0, pv[r].x, -0.5 * acc[r].x), fma(c0, pv[r].y, -0.5 * acc[r].y));
        if (RESID) tv[r] = make_double2(tv[r].x - bv[r].x, tv[r].y - bv[r].y);  // AX -= B (test/solvers.cpp:109)
        else st_nt(reinterpret_cast<double2*>(orow + voff + r * M * 16), tv[r]);
      }
      if (GRAM) {
#pragma unroll
        for (int r = 0; r < 3; ++r) gram_step<16>(G, RESID ? &tv[r] : &pv[r], &tv[r]);
      }
      if (hw.sync != nullptr && tid == 0 && step_n < hw.sync_stride)
        __hip_atomic_fetch_add(hw.sync + cls * hw.sync_stride + step_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // pacing: read now the counter the tile after next is checked against (see k_hop4)
      seen2 = seen1;
      seen2_idx = seen1_idx;
      if (hw.sync != nullptr && tid == 0 && pace && step_n + 2 >= hw.sync_window) {
        seen1_idx = step_n + 2 - hw.sync_window;
        seen1 = read_counter(hw.sync + cls * hw.sync_stride + seen1_idx, zero_rt);
      }
      if (RING) slot = slot + 1 == win.ring ? 0 : slot + 1;
      BCG_STAMPB(7)   // tail: links/row parked, p, stores, pacing counters
    }
  }
#ifdef BCG_HOP4B_STAMPS
  if (!GRAM && lane == 0) {
    double* o = reinterpret_cast<double*>(partials) + (static_cast<int64_t>(blockIdx.x) * 4 + wave) * 16;
    for (int i = 0; i < 16; ++i) o[i] = static_cast<double>(seg[i]);
  }
#endif
#undef BCG_STAMPB
  if (GRAM) {
    if (M == 8) gram_block_store_fold8<NW>(G, smem, partials, tid, hw.fold.out != nullptr);  // two sites per 16-lane row: see the fold
    else gram_block_store<16, NW>(G, smem, partials, tid, hw.fold.out != nullptr);
    gram_fold<M * M>(hw.fold, partials, tid, NW * 64);
  }
#undef BO_F
#undef BO_SEL
}

template <int M, int MODE, bool GRAM, bool RING, bool CB = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_hop4b(
    LatticeDev lat, const double2* __restrict__ U, const double2* __restrict__ Ughost, const double2* __restrict__ in,
    const double2* __restrict__ ghost, double2* __restrict__ out, const double2* __restrict__ p, double c0,
    double2* __restrict__ partials, HopWalk hw, HopWindow win) {
  hop4b_body<M, MODE, GRAM, RING, CB>(lat, U, Ughost, in, ghost, out, p, c0, partials, hw, win);
}

#ifdef BCG_PROBE  // tuning aid: compile only the probed stencil instantiations (seconds instead of minutes)
template __global__ void k_hop4b<16, HOP_PLAIN, false, false>(LatticeDev, const double2*, const double2*, const double2*,
                                                               const double2*, double2*, const double2*, double, double2*,
                                                               HopWalk, HopWindow);
template __global__ void k_hop4b<16, HOP_SHIFTED, true, false>(LatticeDev, const double2*, const double2*, const double2*,
                                                                const double2*, double2*, const double2*, double, double2*,
                                                                HopWalk, HopWindow);
template __global__ void k_hop4c<16, HOP_PLAIN, false, 0, false>(LatticeDev, const double2*, const double2*, const double2*,
                                                                  const double2*, double2*, const double2*, double, double2*,
                                                                  HopWalk, HopWindow);
template __global__ void k_hop4c<16, HOP_SHIFTED, true, 0, false>(LatticeDev, const double2*, const double2*, const double2*,
                                                                   const double2*, double2*, const double2*, double, double2*,
                                                                   HopWalk, HopWindow);
}  // namespace
}  // namespace bcg
#else
}  // namespace

bool hop_fast_width(int m) { return m == 8 || m == 16 || m == 32; }
bool hop_can_split_tiles(int m, const LatticeDev& lat) {
  const int spb = 4 * (64 / m);
  return hop_fast_width(m) && lat.ndim == 4 && lat.L[0] % spb == 0 && lat.V < (int64_t(1) << 31) - 2 * lat.stride[3];
}
// What one launch of the specialised stencil will do (shared by the launcher and hop_kernel_form).
struct HopPlan {
  bool valid = false;
  bool column = false;  // k_hop4c (column sweep, scalar row pointers) instead of k_hop4
  bool list = false;    // k_hop4 over an explicit tile list (boundary class)
  int ntiles = 0, grid = 0;
  HopWalk hw{};
  HopWindow win{};
};
static HopPlan plan_hop4(int m, const LatticeDev& lat, int max_blocks, const HopTuning& tune, int cls, HopWindow win) {
  HopPlan pl;
  const int SPB = 4 * (64 / m);
  // patch extent in x0: one tile unless set (a patch slice then has p1*p2 = 64 tiles = the blocks of an XCD at every width)
  const int walk = tune.patch_walk ? 3 : 0, p0 = tune.patch[0] > 0 ? tune.patch[0] : SPB, p1 = tune.patch[1], p2 = tune.patch[2];
  if (win.x3_n <= 0) {
    win.x3_lo = 0;
    win.x3_n = lat.L[3];
  }
  if (win.x3_lo < 0 || win.x3_lo + win.x3_n > lat.L[3]) return pl;
  // ring addressing: whole tiles only (no interior/boundary split), direction 3 undivided, ring | L3
  if (win.ring > 0 && (cls != 0 || lat.split[3] || win.ring < 3 || lat.L[3] % win.ring != 0)) return pl;
  pl.win = win;
  pl.ntiles = static_cast<int>(lat.V / lat.L[3] * win.x3_n / SPB);
  const bool ok3 = walk == 3 && p0 > 0 && p1 > 0 && p2 > 0 && p0 % SPB == 0 && lat.L[0] % p0 == 0 && lat.L[1] % p1 == 0 &&
                   lat.L[2] % p2 == 0 && pl.ntiles % 8 == 0 && max_blocks % 8 == 0 && pl.ntiles / 8 >= max_blocks / 8;
  pl.hw = HopWalk{lat.L[0], lat.L[1], lat.L[2], 0, nullptr, 0, 0, 0, nullptr, 0, GramFold{}};  // lexicographic = one patch
  if (ok3) pl.hw = HopWalk{p0, p1, p2, 1, nullptr, 0, 0, 0, nullptr, 0, GramFold{}};
  if (ok3 && tune.super_patch && (lat.L[0] / p0) % 2 == 0 && (lat.L[1] / p1) % 2 == 0 && (lat.L[2] / p2) % 2 == 0) pl.hw.super = 1;
  if (cls == 2 && tune.boundary_list != nullptr) {  // the boundary class from its tile list, round-robin
    pl.hw = HopWalk{lat.L[0], lat.L[1], lat.L[2], 0, nullptr, 0, 0, 0, tune.boundary_list, tune.boundary_n, GramFold{}};
    pl.grid = tune.boundary_n < max_blocks ? tune.boundary_n : max_blocks;
    pl.column = false;
    pl.list = true;
    pl.valid = true;
    return pl;
  }
  pl.grid = pl.ntiles < max_blocks ? pl.ntiles : max_blocks;
  if (pl.hw.xcd_split) pl.grid &= ~7;
  // column form of the walk: one block per tile of a patch slice, whole patches per XCD class
  pl.column = tune.sync.column_walk && pl.hw.xcd_split && pl.grid / 8 == (p0 / SPB) * p1 * p2 &&
              ((lat.L[0] / p0) * (lat.L[1] / p1) * (lat.L[2] / p2)) % 8 == 0;
  if (pl.column && tune.sync.window > 0 && tune.sync.counters != nullptr) {
    const int steps = pl.ntiles / pl.grid;  // tiles per block (every block has the same number)
    if (steps <= tune.sync.stride) {
      pl.hw.sync = tune.sync.counters;
      pl.hw.sync_window = tune.sync.window;
      pl.hw.sync_stride = tune.sync.stride;
      pl.hw.sync_limit = tune.sync.limit_ticks;
    }
  }
  pl.valid = true;
  return pl;
}

// k_hop4b (2 x 2 column bundles) serves whole launches of the column form whose patches are made of whole bundle tiles
// (at m = 8 the plain form only unless bundle_walk = 2: measured at 32^4, 0.36 vs 0.41 ms plain, 0.50 vs 0.44 ms with the Gram product)
static bool bundle_ok(int m, const LatticeDev& lat, const HopTuning& tune, const HopPlan& pl, int cls, bool plain) {
  const int spw = 64 / m;
  // short x3 windows (capacity ring 8: 6 slices) do not repay the bundle's column prologue -- two row loads before the
  // first step -- (measured at 64^3 x 128: ring 8 144.6 vs 142.6 ms per iteration, ring 16 140.4 vs 141.1 ms)
  if (pl.win.ring > 0 && pl.win.x3_n < 10 && tune.sync.bundle_walk < 2) return false;
  return pl.valid && pl.column && tune.sync.bundle_walk && cls == 0 && (m != 8 || plain || tune.sync.bundle_walk > 1) &&
         lat.L[1] % 2 == 0 && lat.L[2] % 2 == 0 && pl.hw.p1 % 2 == 0 && pl.hw.p2 % 2 == 0 && pl.hw.p0 % spw == 0 &&
         (pl.hw.p0 / spw) * (pl.hw.p1 / 2) * (pl.hw.p2 / 2) == pl.grid / 8;
}

// The bundle sweep is paced only where it is free: whole-field launches of at least 256 steps per block.  Measured
// (profiles/r03_pacing_ab_other_shapes.txt): at 64^4 (2048 steps) paced = unpaced in time with 20 % less fabric traffic; at
// 32^4, m = 8 (64 steps) the paced plain hop takes 0.376 vs 0.360 ms; capacity-mode windows (15-30 slices) 22.3 vs 21.6 ms.
// BCG_HOP_BUNDLE_SYNC < 0 forces pacing with window |value| everywhere (tests).
// Round 4: with the software-pipelined step (m = 16, 32) the capacity-mode windows gain from pacing too -- 64^3 x 128 share, ring
// 32: hop_ring 18.2 vs 18.8 ms, with the Gram product 23.8 vs 24.4, 107.6-108.1 vs 108.8 ms per iteration
// (profiles/r04_cap128_pacing_ab.txt) -- so there only the step count decides.
static bool bundle_paced(int m, int ntiles, int grid, const HopWindow& win) {
  const bool pipelined = BCG_HOP4B_PIPE != 0 && hop4b_share_images(m) && win.cb == 0;
  return (win.ring == 0 || pipelined) && grid > 0 && ntiles / grid >= 256;
}

template <int M>
static int launch_hop4(hipStream_t s, const LatticeDev& lat, const double2* U, const double2* Ughost, const double2* in,
                       const double2* ghost, double2* out, HopMode mode, const double2* p, double c0, double2* partials,
                       bool gram, int max_blocks, const HopTuning& tune, int cls, const HopWindow& win_in) {
  constexpr int SPB = 4 * (64 / M);
  const HopPlan pl = plan_hop4(M, lat, max_blocks, tune, cls, win_in);
  if (!pl.valid) return -1;
  const int grid = pl.grid, ntiles = pl.ntiles;
  HopWalk hw = pl.hw;
  if (gram && pl.column && cls == 0) hw.fold = tune.fold;  // whole launches of the column forms fold their Gram partials
  const HopWindow win = pl.win;
  const size_t lds_g = gram ? sizeof(double) * 4 * 8 * 64 : 0;
  if (pl.list && grid == 0) return 0;  // no boundary tiles
  const int cls_t = pl.list ? 0 : cls;  // the list holds exactly the launch's tiles: no class filter in the kernel
  // checkerboard form (half-volume fields): the bundle sweep at m = 16, 32 on a lattice whose direction 0 is not divided
  // over ranks (lat: the compact lattice, with the half ghost faces' offsets), or nothing (the caller then runs the generic
  // half-volume kernel)
  if (win.cb) {
    // m = 16 and 32: the widths with two link images per wave (the DMA needs the one that is not being read)
    // (direction 0 divided: a compact row's end sites would read the ghost face in one row parity only -- not built)
    if (!hop4b_share_images(M) || (gram && M != 16) || mode == HOP_RESID || win.ring > 0 || cls != 0 || lat.split[0] ||
        !bundle_ok(M, lat, tune, pl, cls, true))
      return -1;
    constexpr int MC = hop4b_share_images(M) ? M : 16;  // (never launched for the other widths)
    HopWalk hwb = hw;
    if (tune.sync.bundle_window != 0 && hw.sync && (tune.sync.bundle_window < 0 || bundle_paced(M, ntiles, grid, win)))
      hwb.sync_window = tune.sync.bundle_window < 0 ? -tune.sync.bundle_window : tune.sync.bundle_window;
    else hwb.sync = nullptr;
    if (hwb.sync) (void)hipMemsetAsync(hwb.sync, 0, sizeof(unsigned) * 8 * hwb.sync_stride, s);
    constexpr int SPWc = 64 / MC;
    const size_t lds_u = sizeof(double2) * (2 * 4 * ((SPWc + 2) * 3 * MC) + 2 * 4 * ((SPWc + 1) * 36 + 4 * SPWc * 9));
    const size_t lds = lds_u > lds_g ? lds_u : lds_g;
#define BCG_LAUNCH4B_CB(MM, MD, GR)                                                                                      \
  do {                                                                                                              \
    allow_lds(k_hop4b<MM, MD, GR, false, true>, lds);                                                               \
    hipLaunchKernelGGL((k_hop4b<MM, MD, GR, false, true>), dim3(grid), dim3(256), lds, s, lat, U, Ughost, in, ghost, out, p, \
                       c0, partials, hwb, win);                                                                     \
  } while (0)
    if (mode == HOP_PLAIN) BCG_LAUNCH4B_CB(MC, HOP_PLAIN, false);
    else if (gram) BCG_LAUNCH4B_CB(16, HOP_SHIFTED, true);
    else BCG_LAUNCH4B_CB(MC, HOP_SHIFTED, false);
#undef BCG_LAUNCH4B_CB
    return grid;
  }
  // k_hop4b: the column sweep over 2 x 2 bundles (whole launches only: the tile classes stay with k_hop4c)
  if (bundle_ok(M, lat, tune, pl, cls, mode == HOP_PLAIN)) {
    HopWalk hwb = hw;  // pacing of the bundle sweep: its own window, long whole-field sweeps only (bundle_paced)
    if (tune.sync.bundle_window != 0 && hw.sync && (tune.sync.bundle_window < 0 || bundle_paced(M, ntiles, grid, win)))
      hwb.sync_window = tune.sync.bundle_window < 0 ? -tune.sync.bundle_window : tune.sync.bundle_window;
    else hwb.sync = nullptr;
    if (hwb.sync) (void)hipMemsetAsync(hwb.sync, 0, sizeof(unsigned) * 8 * hwb.sync_stride, s);
    constexpr int SPW = 64 / M;
    const size_t lds_u = sizeof(double2) * (2 * 4 * ((SPW + 2) * 3 * M) +
                                            (hop4b_share_images(M) ? 2 : 1) * 4 * ((SPW + 1) * 36 + 3 * SPW * 9));
    const size_t lds = lds_u > lds_g ? lds_u : lds_g;
#define BCG_LAUNCH4B(MM, MD, GR, RG)                                                                                \
  do {                                                                                                             \
    allow_lds(k_hop4b<MM, MD, GR, RG>, lds);                                                                       \
    hipLaunchKernelGGL((k_hop4b<MM, MD, GR, RG>), dim3(grid), dim3(256), lds, s, lat, U, Ughost, in, ghost, out, p, \
                       c0, partials, hwb, win);                                                                     \
  } while (0)
#define BCG_LAUNCH4B_R(MM, MD, GR)                      \
  do {                                                  \
    if (win.ring > 0) BCG_LAUNCH4B(MM, MD, GR, true);   \
    else BCG_LAUNCH4B(MM, MD, GR, false);               \
  } while (0)
    if (mode == HOP_RESID) {
      if (M != 16 || !gram || win.ring > 0) return -1;
      BCG_LAUNCH4B(16, HOP_RESID, true, false);
    } else if (gram && M == 16 && mode == HOP_SHIFTED) BCG_LAUNCH4B_R(16, HOP_SHIFTED, true);
    else if (gram && M == 8 && mode == HOP_SHIFTED) BCG_LAUNCH4B_R(8, HOP_SHIFTED, true);
    else if (mode == HOP_PLAIN) BCG_LAUNCH4B_R(M, HOP_PLAIN, false);
    else BCG_LAUNCH4B_R(M, HOP_SHIFTED, false);
#undef BCG_LAUNCH4B_R
#undef BCG_LAUNCH4B
    return grid;
  }
  if (mode == HOP_RESID) return -1;  // the fused residual form exists in the bundle sweep only
  if (pl.column) {
    if (hw.sync) (void)hipMemsetAsync(hw.sync, 0, sizeof(unsigned) * 8 * hw.sync_stride, s);
    const size_t lds_u = sizeof(double2) * 2 * ((SPB + 1) * 36 + 3 * SPB * 9);
    const size_t lds = lds_u > lds_g ? lds_u : lds_g;
#define BCG_LAUNCH4C(MM, MD, GR, CL, RG)                                                                                \
  do {                                                                                                                 \
    allow_lds(k_hop4c<MM, MD, GR, CL, RG>, lds);                                                                       \
    hipLaunchKernelGGL((k_hop4c<MM, MD, GR, CL, RG>), dim3(grid), dim3(256), lds, s, lat, U, Ughost, in, ghost, out, p, \
                       c0, partials, hw, win);                                                                         \
  } while (0)
#define BCG_LAUNCH4C_R(MM, MD, GR)                          \
  do {                                                      \
    if (win.ring > 0) BCG_LAUNCH4C(MM, MD, GR, 0, true);    \
    else if (cls == 1 && (GR || MM == 8)) { /* interior variants over 256 VGPRs: register-capped entry point */ \
      allow_lds(k_hop4c_interior<MM, MD, GR>, lds);         \
      hipLaunchKernelGGL((k_hop4c_interior<MM, MD, GR>), dim3(grid), dim3(256), lds, s, lat, U, Ughost, in, ghost, out, p, \
                         c0, partials, hw, win);            \
    } else if (cls == 1) BCG_LAUNCH4C(MM, MD, GR, 1, false); \
    else if (cls == 2) BCG_LAUNCH4C(MM, MD, GR, 2, false);  \
    else BCG_LAUNCH4C(MM, MD, GR, 0, false);                \
  } while (0)
    if (gram && M == 16 && mode == HOP_SHIFTED) BCG_LAUNCH4C_R(16, HOP_SHIFTED, true);
    else if (gram && M == 8 && mode == HOP_SHIFTED) BCG_LAUNCH4C_R(8, HOP_SHIFTED, true);
    else if (mode == HOP_PLAIN) BCG_LAUNCH4C_R(M, HOP_PLAIN, false);
    else BCG_LAUNCH4C_R(M, HOP_SHIFTED, false);
#undef BCG_LAUNCH4C_R
#undef BCG_LAUNCH4C
    return grid;
  }
  // link images: three for the fused-Gram variant (x3 carry reads the previous one), two otherwise
  const size_t lds_u = sizeof(double2) * ((gram && M == 16 && mode == HOP_SHIFTED) ? 3 : 2) * ((SPB + 1) * 36 + 3 * SPB * 9);
  const size_t lds = lds_u > lds_g ? lds_u : lds_g;
  // the streaming (non-temporal) form is the only one instantiated
#define BCG_LAUNCH4(MM, MD, GR, CL, RG)                                                                                   \
  do {                                                                                                                   \
    allow_lds(k_hop4<MM, MD, GR, true, CL, RG>, lds);                                                                    \
    hipLaunchKernelGGL((k_hop4<MM, MD, GR, true, CL, RG>), dim3(grid), dim3(256), lds, s, lat, U, Ughost, in, ghost, out, \
                       p, c0, partials, ntiles, hw, win);                                                                \
  } while (0)
#define BCG_LAUNCH4_CLS(MM, MD, GR)                          \
  do {                                                       \
    if (win.ring > 0) BCG_LAUNCH4(MM, MD, GR, 0, true);      \
    else if (cls_t == 1) BCG_LAUNCH4(MM, MD, GR, 1, false);  \
    else if (cls_t == 2) BCG_LAUNCH4(MM, MD, GR, 2, false);  \
    else BCG_LAUNCH4(MM, MD, GR, 0, false);                  \
  } while (0)
  if (gram && M == 16 && mode == HOP_SHIFTED) BCG_LAUNCH4_CLS(16, HOP_SHIFTED, true);
  else if (mode == HOP_PLAIN) BCG_LAUNCH4_CLS(M, HOP_PLAIN, false);
  else BCG_LAUNCH4_CLS(M, HOP_SHIFTED, false);
#undef BCG_LAUNCH4_CLS
#undef BCG_LAUNCH4
  return grid;
}

int hop_kernel_form(int m, const LatticeDev& lat, int max_blocks, const HopTuning& tune, int tile_class, const HopWindow& win) {
  if (!hop_can_split_tiles(m, lat)) return 0;
  const int mb = tune.blocks > 0 ? tune.blocks : max_blocks;
  const HopPlan pl = plan_hop4(m, lat, mb, tune, tile_class, win);
  return !pl.valid ? -1 : (pl.column ? 2 : 1);
}

bool hop_folds_gram(int m, const LatticeDev& lat, int max_blocks, const HopTuning& tune, const HopWindow& win) {
  if (!hop_can_split_tiles(m, lat) || (m != 16 && m != 8)) return false;
  const int mb = tune.blocks > 0 ? tune.blocks : max_blocks;
  const HopPlan pl = plan_hop4(m, lat, mb, tune, 0, win);
  return pl.valid && pl.column;
}

bool hop_uses_bundle(int m, const LatticeDev& lat, int max_blocks, const HopTuning& tune, int tile_class, const HopWindow& win,
                     bool plain) {
  if (!hop_can_split_tiles(m, lat)) return false;
  const int mb = tune.blocks > 0 ? tune.blocks : max_blocks;
  return bundle_ok(m, lat, tune, plan_hop4(m, lat, mb, tune, tile_class, win), tile_class, plain);
}

int launch_hop_fast(hipStream_t s, int m, const LatticeDev& lat, const double2* U, const double2* Ughost,
                    const double2* in, const double2* ghost, double2* out, HopMode mode, const double2* p, double c0,
                    double2* partials, bool gram, int max_blocks, const HopTuning& tune, int tile_class,
                    const HopWindow& win) {
  const int spb = 4 * (64 / m);
  // specialised 4-D kernel: tile = spb consecutive x0 sites of one row, 32-bit site arithmetic
  if (hop_can_split_tiles(m, lat)) {
    const int mb = tune.blocks > 0 ? tune.blocks : max_blocks;
    if (m == 8) return launch_hop4<8>(s, lat, U, Ughost, in, ghost, out, mode, p, c0, partials, gram, mb, tune, tile_class, win);
    if (m == 16) return launch_hop4<16>(s, lat, U, Ughost, in, ghost, out, mode, p, c0, partials, gram, mb, tune, tile_class, win);
    return launch_hop4<32>(s, lat, U, Ughost, in, ghost, out, mode, p, c0, partials, gram, mb, tune, tile_class, win);
  }
  (void)spb;
  return -1;  // no specialised form for this lattice: the caller runs k_hop_generic
}

}  // namespace bcg
#endif  // BCG_PROBE
