// Generic (any block width 1 <= m <= 32) HIP kernels for gfx950: the complete hot path in plain VALU
// form.  They are the product path for widths without an MFMA fast path (every m but 8, 16, 32) and the
// cross-check for the MFMA kernels (kernels_mfma.hip, kernels_stencil.hip) at m = 8,16,32.
//
// One thread per OUTPUT complex element everywhere, so global loads/stores are 16 B per lane and
// consecutive lanes touch consecutive addresses (the device layout keeps a (site,colour) row of m
// complex numbers contiguous).  Operands that every lane of a row needs (the input row, the m x m
// coefficient matrix) are staged in LDS.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace bcg {

namespace {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// acc += a*b
__device__ __forceinline__ void cfma(double2& acc, double2 a, double2 b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
__device__ __forceinline__ void cfma_conj(double2& acc, double2 a, double2 b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(-a.y, b.x, acc.y);
}
// acc -= a*b
__device__ __forceinline__ void cfms(double2& acc, double2 a, double2 b) {
  acc.x = fma(-a.x, b.x, acc.x);
  acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(-a.x, b.y, acc.y);
  acc.y = fma(-a.y, b.x, acc.y);
}
__device__ __forceinline__ double2 cdiv(double2 a, double2 b) {
  const double d = b.x * b.x + b.y * b.y;
  return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

// splitmix64-based counter generator; bit-identical to oracle::uniform_pm1 (oracle/oracle.hpp) --
// restated here because the product may not depend on oracle/.
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double uniform_pm1(uint64_t seed_mixed, uint64_t counter) {
  const uint64_t h = splitmix64(seed_mixed ^ (counter * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
  return static_cast<double>(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
inline uint64_t splitmix64_host(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ void site_coords(const LatticeDev& lat, int64_t site, int x[4]) {
  x[0] = static_cast<int>(site % lat.L[0]); site /= lat.L[0];
  x[1] = static_cast<int>(site % lat.L[1]); site /= lat.L[1];
  x[2] = static_cast<int>(site % lat.L[2]); site /= lat.L[2];
  x[3] = static_cast<int>(site);
}
// lexicographic index of x over all directions except mu
__device__ __forceinline__ int64_t face_index(const LatticeDev& lat, const int x[4], int mu) {
  int64_t f = 0, st = 1;
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    if (nu == mu) continue;
    f += x[nu] * st;
    st *= lat.L[nu];
  }
  return f;
}
// inverse of face_index with x[mu] = xmu
__device__ __forceinline__ int64_t face_to_site(const LatticeDev& lat, int64_t f, int mu, int xmu) {
  int64_t site = 0;
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    int c;
    if (nu == mu) {
      c = xmu;
    } else {
      c = static_cast<int>(f % lat.L[nu]);
      f /= lat.L[nu];
    }
    site += c * lat.stride[nu];
  }
  return site;
}
__device__ __forceinline__ int64_t global_site(const LatticeDev& lat, const int* gdims, int64_t site) {
  int x[4];
  site_coords(lat, site, x);
  int64_t g = 0, st = 1;
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    g += (x[nu] + lat.origin[nu]) * st;
    st *= gdims[nu];
  }
  return g;
}

struct GDims {
  int d[4];
};

// ---------------------------------------------------------------------------------------------
// y == x is allowed (add(rhs = *this, a), inc/fields.hpp:70-77 accepts it): each element is read before it is written and
// no other thread touches it, so the pointers carry no __restrict__.
__global__ void k_axpby(double2* y, double a, const double2* x, double b, int64_t n) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const double2 yv = y[i], xv = x[i];
    y[i] = make_double2(a * yv.x + b * xv.x, a * yv.y + b * xv.y);
  }
}

// SCG's per-shift updates of one iteration (src/standard_solvers.cpp:73-75,85-87) as one pass: r is read once, every x_s
// and p_s once, instead of three passes per axpy (1 + 4 S field passes instead of 6 S).  The expressions are k_axpby's, so
// the iterates are those of the unfused sequence:  x_s += alpha_s p_s ;  p_s = beta_s p_s + zeta_s r.
struct ScgShiftArgs {
  double2* x[16];
  double2* p[16];
  double alpha[16], beta[16], zeta[16];
  double one;  // 1.0 at run time: keeps the expression (and its FMA contraction) the one k_axpby compiles to
};
__global__ void k_scg_update(const double2* __restrict__ r, ScgShiftArgs a, int nshift, int64_t n) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const double2 rv = r[i];
    for (int s = 0; s < nshift; ++s) {
      const double2 pv = a.p[s][i], xv = a.x[s][i];
      a.x[s][i] = make_double2(a.one * xv.x + a.alpha[s] * pv.x, a.one * xv.y + a.alpha[s] * pv.y);
      a.p[s][i] = make_double2(a.beta[s] * pv.x + a.zeta[s] * rv.x, a.beta[s] * pv.y + a.zeta[s] * rv.y);
    }
  }
}

// dev[(site*3+c)*m + j] = host[(site*m + j)*3 + c]
__global__ void k_host_to_dev(int m, const double2* __restrict__ h, double2* __restrict__ d, int64_t n) {
  const int row = 3 * m;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t site = i / row;
    const int e = static_cast<int>(i - site * row);
    const int c = e / m, j = e - c * m;
    d[i] = h[site * row + j * 3 + c];
  }
}
__global__ void k_dev_to_host(int m, const double2* __restrict__ d, double2* __restrict__ h, int64_t n) {
  const int row = 3 * m;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t site = i / row;
    const int e = static_cast<int>(i - site * row);
    const int j = e / 3, c = e - j * 3;
    h[i] = d[site * row + c * m + j];
  }
}

__global__ void k_fill_field(int m, LatticeDev lat, GDims g, double2* __restrict__ f, uint64_t seed_mixed) {
  const int row = 3 * m;
  const int64_t n = lat.V * row;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t site = i / row;
    const int e = static_cast<int>(i - site * row);
    const int c = e / m, j = e - c * m;
    const uint64_t gx = static_cast<uint64_t>(global_site(lat, g.d, site));
    const uint64_t cnt = ((gx * m + j) * 3 + c) * 2;  // oracle::field_counter
    f[i] = make_double2(uniform_pm1(seed_mixed, cnt), uniform_pm1(seed_mixed, cnt + 1));
  }
}
__global__ void k_fill_gauge(LatticeDev lat, GDims g, double2* __restrict__ U, uint64_t seed_mixed) {
  const int per_site = lat.ndim * 9;
  const int64_t n = lat.V * per_site;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t site = i / per_site;
    const int e = static_cast<int>(i - site * per_site);  // = (mu*3 + k)*3 + r
    const uint64_t gx = static_cast<uint64_t>(global_site(lat, g.d, site));
    const uint64_t cnt = (gx * per_site + e) * 2;  // oracle::gauge_counter
    U[i] = make_double2(uniform_pm1(seed_mixed, cnt), uniform_pm1(seed_mixed, cnt + 1));
  }
}

// one thread per complex element of one face; grid.y = 2*k + side for the k-th split direction
__global__ void k_pack_faces(int m, LatticeDev lat, const double2* __restrict__ f, double2* __restrict__ send, int x3_lo,
                             int x3_n, int ring) {
  int mu = -1, k = blockIdx.y >> 1;
  const int side = blockIdx.y & 1;
  int64_t base_sites = 0;
  for (int nu = 0; nu < lat.ndim; ++nu) {
    if (!lat.split[nu]) continue;
    if (k == 0) { mu = nu; break; }
    --k;
    base_sites += 2 * lat.face_sites[nu];
  }
  if (mu < 0) return;
  const int row = 3 * m;
  // x3 window (direction 3 undivided, so mu < 3 and x3 is the slowest face coordinate): a contiguous range of the face
  const int64_t slice = x3_n > 0 ? lat.face_sites[mu] / lat.L[3] : 0;
  const int64_t first = x3_n > 0 ? x3_lo * slice : 0;
  const int64_t n = (x3_n > 0 ? x3_n * slice : lat.face_sites[mu]) * row;
  double2* dst = send + (base_sites + side * lat.face_sites[mu] + first) * row;
  const int xmu = side ? lat.L[mu] - 1 : 0;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t fs = i / row;
    const int e = static_cast<int>(i - fs * row);
    int64_t site = face_to_site(lat, first + fs, mu, xmu);
    if (ring > 0) {  // slice x3 of the field lives in slot x3 % ring
      const int64_t x3 = site / lat.stride[3];
      site += (x3 % ring - x3) * lat.stride[3];
    }
    dst[i] = f[site * row + e];
  }
}
// The faces of a half-volume field (sites of `parity` only; helpers below, "Half-volume fields"): face site f of the full
// numbering (face_index) goes to f >> 1 -- the face's fastest coordinate (x0; x1 on the faces of direction 0) has an even
// extent and alternates in parity -- so a half face is the first half of the full face's range and every offset of the
// full message plan halves (the host posts the plan with half the bytes per site).
__device__ __forceinline__ int64_t half_index(const LatticeDev& lat, const int x[4]);
// x3_n > 0 (direction 3 undivided): the slices [x3_lo, x3_lo + x3_n) only -- x3 is the slowest face coordinate, so a
// contiguous range of every half face, as for full fields.
__global__ void k_pack_faces_half(int m, LatticeDev lat, int parity, const double2* __restrict__ f, double2* __restrict__ send,
                                  int x3_lo, int x3_n) {
  int mu = -1, k = blockIdx.y >> 1;
  const int side = blockIdx.y & 1;
  int64_t base_sites = 0;
  for (int nu = 0; nu < lat.ndim; ++nu) {
    if (!lat.split[nu]) continue;
    if (k == 0) { mu = nu; break; }
    --k;
    base_sites += 2 * lat.face_sites[nu];
  }
  if (mu < 0) return;
  const int row = 3 * m;
  const int64_t half_face = lat.face_sites[mu] >> 1;
  const int64_t slice = x3_n > 0 ? half_face / lat.L[3] : 0;
  const int64_t first = x3_n > 0 ? x3_lo * slice : 0;
  const int64_t n = (x3_n > 0 ? x3_n * slice : half_face) * row;
  double2* dst = send + ((base_sites >> 1) + side * half_face + first) * row;
  const int cn = mu == 0 ? 1 : 0;  // the coordinate the half face is compact in
  const int o = lat.origin[0] + lat.origin[1] + lat.origin[2] + lat.origin[3];
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t hf = i / row;
    const int e = static_cast<int>(i - hf * row);
    int x[4];
    int64_t fl = 2 * (first + hf);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      if (nu == mu) {
        x[nu] = side ? lat.L[mu] - 1 : 0;
      } else {
        x[nu] = static_cast<int>(fl % lat.L[nu]);
        fl /= lat.L[nu];
      }
    }
    x[cn] += (x[0] + x[1] + x[2] + x[3] + parity + o) & 1;  // x[cn] is even here: the site of the pair that has the parity
    dst[i] = f[half_index(lat, x) * row + e];
  }
}
// gauge: only U_mu of the split direction mu itself, 9 complex per site
__global__ void k_pack_gauge_faces(LatticeDev lat, const double2* __restrict__ U, double2* __restrict__ send) {
  int mu = -1, k = blockIdx.y >> 1;
  const int side = blockIdx.y & 1;
  int64_t base_sites = 0;
  for (int nu = 0; nu < lat.ndim; ++nu) {
    if (!lat.split[nu]) continue;
    if (k == 0) { mu = nu; break; }
    --k;
    base_sites += 2 * lat.face_sites[nu];
  }
  if (mu < 0) return;
  const int64_t n = lat.face_sites[mu] * 9;
  double2* dst = send + (base_sites + side * lat.face_sites[mu]) * 9;
  const int xmu = side ? lat.L[mu] - 1 : 0;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t fs = i / 9;
    const int e = static_cast<int>(i - fs * 9);
    dst[i] = U[(face_to_site(lat, fs, mu, xmu) * lat.ndim + mu) * 9 + e];
  }
}

// ---------------------------------------------------------------------------------------------
// K1.  One thread per (site, rhs j): 3 colour outputs.  U is read by all m lanes of a site at the
// same address (one fetch per site).  Neighbours across a split direction come from the ghost
// buffer filled by the halo exchange; U_mu(x-mu) across it from the gauge ghost.
// ---------------------------------------------------------------------------------------------
template <int M, int MODE>
__global__ void __launch_bounds__(256) k_hop_generic(LatticeDev lat, const double2* __restrict__ U,
                                                     const double2* __restrict__ Ughost,
                                                     const double2* __restrict__ in,
                                                     const double2* __restrict__ ghost, double2* __restrict__ out,
                                                     const double2* __restrict__ p, double c0) {
  constexpr int SPB = 256 / M;  // sites per block
  const int sl = threadIdx.x / M;
  const int j = threadIdx.x - sl * M;
  const int64_t site = static_cast<int64_t>(blockIdx.x) * SPB + sl;
  if (sl >= SPB || site >= lat.V) return;
  int x[4];
  site_coords(lat, site, x);
  double2 acc[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
  int parity = 0;  // x_0 + ... + x_{mu-1} (global)
  for (int mu = 0; mu < lat.ndim; ++mu) {
    const double eta = (parity & 1) ? -1.0 : 1.0;
    // forward neighbour
    const double2* pf;
    if (x[mu] + 1 < lat.L[mu]) pf = in + (site + lat.stride[mu]) * 3 * M;
    else if (!lat.split[mu]) pf = in + (site - (lat.L[mu] - 1) * lat.stride[mu]) * 3 * M;
    else pf = ghost + (lat.ghost_off[mu][1] + face_index(lat, x, mu)) * 3 * M;
    // backward neighbour and its link
    const double2* pb;
    const double2* ub;
    if (x[mu] > 0) {
      const int64_t xb = site - lat.stride[mu];
      pb = in + xb * 3 * M;
      ub = U + (xb * lat.ndim + mu) * 9;
    } else if (!lat.split[mu]) {
      const int64_t xb = site + (lat.L[mu] - 1) * lat.stride[mu];
      pb = in + xb * 3 * M;
      ub = U + (xb * lat.ndim + mu) * 9;
    } else {
      const int64_t fi = face_index(lat, x, mu);
      pb = ghost + (lat.ghost_off[mu][0] + fi) * 3 * M;
      ub = Ughost + (lat.ghost_off[mu][0] + fi) * 9;  // gauge ghost shares the face numbering
    }
    const double2* uf = U + (site * lat.ndim + mu) * 9;
    double2 t[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double2 psf = pf[k * M + j];
      const double2 psb = pb[k * M + j];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        cfma(t[r], uf[k * 3 + r], psf);                                                     // + U(r,k) psi_f(k)
        const double2 v = ub[r * 3 + k];                                                    // U_b(k,r)
        t[r].x = fma(-v.x, psb.x, t[r].x); t[r].x = fma(-v.y, psb.y, t[r].x);             // - conj(U_b(k,r)) psi_b(k)
        t[r].y = fma(-v.x, psb.y, t[r].y); t[r].y = fma(v.y, psb.x, t[r].y);
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      acc[r].x = fma(eta, t[r].x, acc[r].x);
      acc[r].y = fma(eta, t[r].y, acc[r].y);
    }
    parity += x[mu] + lat.origin[mu];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int64_t o = (site * 3 + r) * M + j;
    if (MODE == HOP_PLAIN) {
      out[o] = make_double2(0.5 * acc[r].x, 0.5 * acc[r].y);
    } else {
      const double2 pv = p[o];
      out[o] = make_double2(fma(c0, pv.x, -0.5 * acc[r].x), fma(c0, pv.y, -0.5 * acc[r].y));
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Half-volume (parity-compact) fields.  D couples sites of opposite parity only (inc/dirac_op.hpp:14-21: nearest
// neighbours), so A = mass^2 - D^2 is block diagonal in the parity x_0 + x_1 + x_2 + x_3 (mod 2) and the solve splits into
// two half-volume solves.  A field of parity p holds the V/2 sites of that parity: half site
// h = k + (L0/2) * (x1 + L1 * (x2 + L2 * x3)) is the full-lattice site with x0 = 2 k + ((x1 + x2 + x3 + p + o) & 1), o the
// parity of the local origin.  All extents even.  Links keep the full-lattice layout.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void half_coords(const LatticeDev& lat, int64_t h, int parity, int x[4]) {
  const int h0 = lat.L[0] >> 1;
  const int k = static_cast<int>(h % h0); h /= h0;
  x[1] = static_cast<int>(h % lat.L[1]); h /= lat.L[1];
  x[2] = static_cast<int>(h % lat.L[2]); h /= lat.L[2];
  x[3] = static_cast<int>(h);
  const int o = lat.origin[0] + lat.origin[1] + lat.origin[2] + lat.origin[3];
  x[0] = 2 * k + ((x[1] + x[2] + x[3] + parity + o) & 1);
}
__device__ __forceinline__ int64_t half_index(const LatticeDev& lat, const int x[4]) {
  return (x[0] >> 1) + static_cast<int64_t>(lat.L[0] >> 1) * (x[1] + static_cast<int64_t>(lat.L[1]) * (x[2] + static_cast<int64_t>(lat.L[2]) * x[3]));
}
__device__ __forceinline__ int64_t full_index(const LatticeDev& lat, const int x[4]) {
  return x[0] + static_cast<int64_t>(lat.L[0]) * (x[1] + static_cast<int64_t>(lat.L[1]) * (x[2] + static_cast<int64_t>(lat.L[2]) * x[3]));
}

// out (parity p) = D in (parity 1 - p)   [HOP_PLAIN]   or   out = c0 * pfield - D in   [HOP_SHIFTED; pfield of parity p].
// The arithmetic per site is k_hop_generic's, term for term (same results as the full-volume operator on that site).
// Across a split direction the neighbour comes from the HALF ghost face (k_pack_faces_half: face site f of the full
// numbering lives at f >> 1, the ghost offsets are half the full ones), its link from the gauge ghost (full numbering).
template <int M, int MODE>
__global__ void __launch_bounds__(256) k_hop_half(LatticeDev lat, int parity, const double2* __restrict__ U,
                                                  const double2* __restrict__ Ughost, const double2* __restrict__ in,
                                                  const double2* __restrict__ ghost, double2* __restrict__ out,
                                                  const double2* __restrict__ p, double c0) {
  constexpr int SPB = 256 / M;
  const int sl = threadIdx.x / M;
  const int j = threadIdx.x - sl * M;
  const int64_t h = static_cast<int64_t>(blockIdx.x) * SPB + sl;
  if (sl >= SPB || h >= lat.V / 2) return;
  int x[4];
  half_coords(lat, h, parity, x);
  const int64_t site = full_index(lat, x);
  double2 acc[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
  int par = 0;  // x_0 + ... + x_{mu-1} (global)
  for (int mu = 0; mu < lat.ndim; ++mu) {
    const double eta = (par & 1) ? -1.0 : 1.0;
    int xf[4] = {x[0], x[1], x[2], x[3]}, xb[4] = {x[0], x[1], x[2], x[3]};
    xf[mu] = x[mu] + 1 < lat.L[mu] ? x[mu] + 1 : 0;
    xb[mu] = x[mu] > 0 ? x[mu] - 1 : lat.L[mu] - 1;
    const double2* pf = in + half_index(lat, xf) * 3 * M;
    const double2* pb = in + half_index(lat, xb) * 3 * M;
    const double2* ub = U + (full_index(lat, xb) * lat.ndim + mu) * 9;
    if (lat.split[mu] && (x[mu] + 1 == lat.L[mu] || x[mu] == 0)) {
      const int64_t fi = face_index(lat, x, mu);
      if (x[mu] + 1 == lat.L[mu]) pf = ghost + ((lat.ghost_off[mu][1] >> 1) + (fi >> 1)) * 3 * M;
      if (x[mu] == 0) {
        pb = ghost + ((lat.ghost_off[mu][0] >> 1) + (fi >> 1)) * 3 * M;
        ub = Ughost + (lat.ghost_off[mu][0] + fi) * 9;
      }
    }
    const double2* uf = U + (site * lat.ndim + mu) * 9;
    double2 t[3] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double2 psf = pf[k * M + j];
      const double2 psb = pb[k * M + j];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        cfma(t[r], uf[k * 3 + r], psf);
        const double2 v = ub[r * 3 + k];
        t[r].x = fma(-v.x, psb.x, t[r].x); t[r].x = fma(-v.y, psb.y, t[r].x);
        t[r].y = fma(-v.x, psb.y, t[r].y); t[r].y = fma(v.y, psb.x, t[r].y);
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      acc[r].x = fma(eta, t[r].x, acc[r].x);
      acc[r].y = fma(eta, t[r].y, acc[r].y);
    }
    par += x[mu] + lat.origin[mu];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int64_t o = (h * 3 + r) * M + j;
    if (MODE == HOP_PLAIN) {
      out[o] = make_double2(0.5 * acc[r].x, 0.5 * acc[r].y);
    } else {
      const double2 pv = p[o];
      out[o] = make_double2(fma(c0, pv.x, -0.5 * acc[r].x), fma(c0, pv.y, -0.5 * acc[r].y));
    }
  }
}

// full field <-> its two parity-compact halves (to_half: half = full restricted; else: full sites of that parity = half)
__global__ void k_parity_copy(int m, LatticeDev lat, int parity, double2* __restrict__ full, double2* __restrict__ half, int to_half) {
  const int row = 3 * m;
  const int64_t n = lat.V / 2 * row;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t h = i / row;
    const int e = static_cast<int>(i - h * row);
    int x[4];
    half_coords(lat, h, parity, x);
    const int64_t fi = full_index(lat, x) * row + e;
    if (to_half) half[i] = full[fi];
    else full[fi] = half[i];
  }
}
// the counter-based generator on a half field: the values the full field has at those sites
__global__ void k_fill_field_half(int m, LatticeDev lat, GDims g, int parity, double2* __restrict__ f, uint64_t seed_mixed) {
  const int row = 3 * m;
  const int64_t n = lat.V / 2 * row;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t h = i / row;
    const int e = static_cast<int>(i - h * row);
    const int c = e / m, j = e - c * m;
    int x[4];
    half_coords(lat, h, parity, x);
    const uint64_t gx = static_cast<uint64_t>(global_site(lat, g.d, full_index(lat, x)));
    const uint64_t cnt = ((gx * m + j) * 3 + c) * 2;
    f[i] = make_double2(uniform_pm1(seed_mixed, cnt), uniform_pm1(seed_mixed, cnt + 1));
  }
}

// ---------------------------------------------------------------------------------------------
// K5 / K6: right-multiplication of every row by an m x m matrix.  Thread (row, j) computes one
// output element; the input rows of the tile and the matrix (stored transposed so lanes j read
// consecutive LDS words) sit in LDS.
// ---------------------------------------------------------------------------------------------
template <int M, int MODE>
__global__ void __launch_bounds__(256) k_rmul_generic(int64_t rows, double2* __restrict__ y,
                                                      const double2* __restrict__ x,
                                                      const double2* __restrict__ Md, double b) {
  constexpr int RPB = 256 / M;
  __shared__ double2 Mt[M * M];      // Mt[k*M + j] = M(k,j)
  __shared__ double2 xs[RPB * M];
  for (int e = threadIdx.x; e < M * M; e += blockDim.x) {
    const int k = e % M, jj = e / M;  // Md column-major: element (k,jj) at jj*M+k
    Mt[k * M + jj] = Md[e];
  }
  const int rl = threadIdx.x / M;
  const int j = threadIdx.x - rl * M;
  const bool lane_ok = rl < RPB;
  const int64_t ntiles = (rows + RPB - 1) / RPB;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row = tile * RPB + rl;
    const bool ok = lane_ok && row < rows;
    const int64_t e = row * M + j;
    __syncthreads();
    if (ok) xs[rl * M + j] = (MODE == RMUL_ADD) ? x[e] : y[e];
    __syncthreads();
    if (ok) {
      double2 acc = make_double2(0, 0);
#pragma unroll
      for (int k = 0; k < M; ++k) cfma(acc, xs[rl * M + k], Mt[k * M + j]);
      if (MODE == RMUL_ADD) {
        const double2 yv = y[e];
        y[e] = make_double2(yv.x + acc.x, yv.y + acc.y);
      } else if (MODE == RMUL_XPAY) {
        const double2 xv = x[e];
        y[e] = make_double2(fma(b, xv.x, acc.x), fma(b, xv.y, acc.y));
      } else {
        y[e] = acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K7: column-by-column forward substitution, one thread per row, tile staged through LDS
// (padded by one complex per row against bank conflicts).
// ---------------------------------------------------------------------------------------------
template <int M>
__global__ void __launch_bounds__((M >= 16) ? 64 : 128) k_trisolve_generic(int64_t rows, double2* __restrict__ y,
                                                          const double2* __restrict__ Rd) {
  constexpr int TR = (M >= 16) ? 64 : 128;
  constexpr int LD = M + 1;
  __shared__ double2 ts[TR * LD];
  __shared__ double2 Rs[M * M];  // column-major, R(j,i) at i*M+j
  for (int e = threadIdx.x; e < M * M; e += blockDim.x) Rs[e] = Rd[e];
  const int64_t ntiles = (rows + TR - 1) / TR;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    const int nr = static_cast<int>(rows - row0 < TR ? rows - row0 : TR);
    __syncthreads();
    for (int e = threadIdx.x; e < nr * M; e += TR) ts[(e / M) * LD + (e % M)] = y[row0 * M + e];
    __syncthreads();
    if (threadIdx.x < nr) {
      double2* r = ts + threadIdx.x * LD;
#pragma unroll 1
      for (int i = 0; i < M; ++i) {
        double2 v = r[i];
        for (int jj = 0; jj < i; ++jj) cfms(v, Rs[i * M + jj], r[jj]);
        r[i] = cdiv(v, Rs[i * M + i]);
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nr * M; e += TR) y[row0 * M + e] = ts[(e / M) * LD + (e % M)];
  }
}

// ---------------------------------------------------------------------------------------------
// K4: G(i,j) = sum_rows conj(a[row][i]) b[row][j].  Tiles of TR rows of a and b in LDS; each thread
// owns pair(s) (i,j) and, when m*m < 256, a subset of the rows.  Block partials are written out and
// summed in a fixed order by k_reduce_partials (bitwise reproducible run to run).
// ---------------------------------------------------------------------------------------------
template <int M>
__global__ void __launch_bounds__(256) k_gram_generic(int64_t rows, const double2* __restrict__ a,
                                                      const double2* __restrict__ b,
                                                      double2* __restrict__ partials) {
  constexpr int P = M * M;
  constexpr int TR = (M > 16) ? 32 : (P < 64 ? 256 : 64);  // two tiles of TR rows in LDS: 32 KB at m = 32
  constexpr int NG = (P >= 256) ? 1 : 256 / P;       // row groups
  constexpr int NPT = (P + 255) / 256;               // pairs per thread when P > 256
  __shared__ double2 as[TR * M];
  __shared__ double2 bs[TR * M];
  double2 acc[NPT];
#pragma unroll
  for (int q = 0; q < NPT; ++q) acc[q] = make_double2(0, 0);
  const int g = (P >= 256) ? 0 : threadIdx.x / P;
  const int p0 = (P >= 256) ? threadIdx.x : threadIdx.x - g * P;
  const bool active = g < NG;
  const int64_t ntiles = (rows + TR - 1) / TR;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    const int nr = static_cast<int>(rows - row0 < TR ? rows - row0 : TR);
    __syncthreads();
    for (int e = threadIdx.x; e < nr * M; e += 256) {
      as[e] = a[row0 * M + e];
      bs[e] = b[row0 * M + e];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int q = 0; q < NPT; ++q) {
        const int p = p0 + q * 256;
        if (p < P) {
          const int i = p % M, jj = p / M;
          for (int r = g; r < nr; r += NG) cfma_conj(acc[q], as[r * M + i], bs[r * M + jj]);
        }
      }
    }
  }
  if (NG > 1) {
    // combine the row groups in group order
    __syncthreads();
    __shared__ double2 red[256];
    if (active) red[g * P + p0] = acc[0];
    __syncthreads();
    if (threadIdx.x < P) {
      double2 s = red[threadIdx.x];
      for (int gg = 1; gg < NG; ++gg) {
        s.x += red[gg * P + threadIdx.x].x;
        s.y += red[gg * P + threadIdx.x].y;
      }
      partials[static_cast<int64_t>(blockIdx.x) * P + threadIdx.x] = s;
    }
  } else if (active) {
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      const int p = p0 + q * 256;
      if (p < P) partials[static_cast<int64_t>(blockIdx.x) * P + p] = acc[q];
    }
  }
}

// out[p] = sum over blocks in a fixed order (deterministic): W lanes cooperate on one value (strided partial sums,
// then a fixed-order combine).  W = 8, or 64 when there are thousands of blocks (capacity mode's chunked stencil).
template <int W>
__global__ void __launch_bounds__(256) k_reduce_partials(int n_values, int n_blocks,
                                                         const double2* __restrict__ partials,
                                                         double2* __restrict__ out) {
  __shared__ double2 sh[256];
  const int v = blockIdx.x * (256 / W) + threadIdx.x / W;
  const int w = threadIdx.x % W;
  double2 s = make_double2(0, 0);
  if (v < n_values)
    for (int bidx = w; bidx < n_blocks; bidx += W) {
      const double2 t = partials[static_cast<int64_t>(bidx) * n_values + v];
      s.x += t.x;
      s.y += t.y;
    }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (w == 0 && v < n_values) {
    for (int k = 1; k < W; ++k) {
      s.x += sh[threadIdx.x + k].x;
      s.y += sh[threadIdx.x + k].y;
    }
    out[v] = s;
  }
}

inline int grid_for(int64_t n, int block, int cap) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

}  // namespace

// Any block width 1 <= m <= 32 (the reference's N_rhs is an arbitrary template int, inc/fields.hpp:19-26): every generic
// kernel is instantiated for each of them; the MFMA row kernels and the specialised stencil exist at 8, 16 and 32.
bool width_supported(int m) { return m >= 1 && m <= 32; }

#define BCG_DISPATCH_M(m, CALL)                          \
  switch (m) {                                           \
    case 1: { constexpr int M = 1; CALL; } break;       \
    case 2: { constexpr int M = 2; CALL; } break;       \
    case 3: { constexpr int M = 3; CALL; } break;       \
    case 4: { constexpr int M = 4; CALL; } break;       \
    case 5: { constexpr int M = 5; CALL; } break;       \
    case 6: { constexpr int M = 6; CALL; } break;       \
    case 7: { constexpr int M = 7; CALL; } break;       \
    case 8: { constexpr int M = 8; CALL; } break;       \
    case 9: { constexpr int M = 9; CALL; } break;       \
    case 10: { constexpr int M = 10; CALL; } break;      \
    case 11: { constexpr int M = 11; CALL; } break;      \
    case 12: { constexpr int M = 12; CALL; } break;      \
    case 13: { constexpr int M = 13; CALL; } break;      \
    case 14: { constexpr int M = 14; CALL; } break;      \
    case 15: { constexpr int M = 15; CALL; } break;      \
    case 16: { constexpr int M = 16; CALL; } break;      \
    case 17: { constexpr int M = 17; CALL; } break;      \
    case 18: { constexpr int M = 18; CALL; } break;      \
    case 19: { constexpr int M = 19; CALL; } break;      \
    case 20: { constexpr int M = 20; CALL; } break;      \
    case 21: { constexpr int M = 21; CALL; } break;      \
    case 22: { constexpr int M = 22; CALL; } break;      \
    case 23: { constexpr int M = 23; CALL; } break;      \
    case 24: { constexpr int M = 24; CALL; } break;      \
    case 25: { constexpr int M = 25; CALL; } break;      \
    case 26: { constexpr int M = 26; CALL; } break;      \
    case 27: { constexpr int M = 27; CALL; } break;      \
    case 28: { constexpr int M = 28; CALL; } break;      \
    case 29: { constexpr int M = 29; CALL; } break;      \
    case 30: { constexpr int M = 30; CALL; } break;      \
    case 31: { constexpr int M = 31; CALL; } break;      \
    case 32: { constexpr int M = 32; CALL; } break;      \
    default: break;                                      \
  }

void launch_axpby(hipStream_t s, double2* y, double a, const double2* x, double b, int64_t n) {
  hipLaunchKernelGGL(k_axpby, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, y, a, x, b, n);
}
void launch_scg_update(hipStream_t s, const double2* r, int nshift, double2* const* x, double2* const* p,
                       const double* alpha, const double* beta, const double* zeta, int64_t n) {
  for (int s0 = 0; s0 < nshift; s0 += 16) {  // 16 shifts per launch (kernel-argument space)
    ScgShiftArgs a{};
    a.one = 1.0;
    const int ns = nshift - s0 < 16 ? nshift - s0 : 16;
    for (int k = 0; k < ns; ++k) {
      a.x[k] = x[s0 + k];
      a.p[k] = p[s0 + k];
      a.alpha[k] = alpha[s0 + k];
      a.beta[k] = beta[s0 + k];
      a.zeta[k] = zeta[s0 + k];
    }
    hipLaunchKernelGGL(k_scg_update, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, r, a, ns, n);
  }
}
void launch_host_to_dev(hipStream_t s, int m, const double2* h, double2* d, int64_t nsites) {
  const int64_t n = nsites * 3 * m;
  hipLaunchKernelGGL(k_host_to_dev, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, m, h, d, n);
}
void launch_dev_to_host(hipStream_t s, int m, const double2* d, double2* h, int64_t nsites) {
  const int64_t n = nsites * 3 * m;
  hipLaunchKernelGGL(k_dev_to_host, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, m, d, h, n);
}
void launch_fill_field(hipStream_t s, int m, const LatticeDev& lat, const int* gdims, double2* f, uint64_t seed) {
  GDims g{{gdims[0], gdims[1], gdims[2], gdims[3]}};
  hipLaunchKernelGGL(k_fill_field, dim3(grid_for(lat.V * 3 * m, 256, 8192)), dim3(256), 0, s, m, lat, g, f,
                     splitmix64_host(seed));
}
void launch_fill_gauge(hipStream_t s, const LatticeDev& lat, const int* gdims, double2* U, uint64_t seed) {
  GDims g{{gdims[0], gdims[1], gdims[2], gdims[3]}};
  hipLaunchKernelGGL(k_fill_gauge, dim3(grid_for(lat.V * lat.ndim * 9, 256, 8192)), dim3(256), 0, s, lat, g, U,
                     splitmix64_host(seed));
}
static int n_split(const LatticeDev& lat) {
  int n = 0;
  for (int mu = 0; mu < lat.ndim; ++mu) n += lat.split[mu] ? 1 : 0;
  return n;
}
static int64_t max_face(const LatticeDev& lat) {
  int64_t f = 0;
  for (int mu = 0; mu < lat.ndim; ++mu)
    if (lat.split[mu] && lat.face_sites[mu] > f) f = lat.face_sites[mu];
  return f;
}
void launch_pack_faces(hipStream_t s, int m, const LatticeDev& lat, const double2* f, double2* send, int x3_lo, int x3_n,
                       int ring) {
  const int ns = n_split(lat);
  if (ns == 0) return;
  int64_t work = max_face(lat) * 3 * m;
  if (x3_n > 0) work = work / lat.L[3] * x3_n;
  hipLaunchKernelGGL(k_pack_faces, dim3(grid_for(work, 256, 4096), 2 * ns), dim3(256), 0, s, m, lat, f, send, x3_lo, x3_n,
                     ring);
}
void launch_pack_gauge_faces(hipStream_t s, const LatticeDev& lat, const double2* U, double2* send) {
  const int ns = n_split(lat);
  if (ns == 0) return;
  hipLaunchKernelGGL(k_pack_gauge_faces, dim3(grid_for(max_face(lat) * 9, 256, 4096), 2 * ns), dim3(256), 0, s, lat, U,
                     send);
}

void launch_hop_generic(hipStream_t s, int m, const LatticeDev& lat, const double2* U, const double2* Ughost,
                        const double2* in, const double2* ghost, double2* out, HopMode mode, const double2* p,
                        double c0) {
  BCG_DISPATCH_M(m, {
    constexpr int SPB = 256 / M;
    const unsigned grid = static_cast<unsigned>((lat.V + SPB - 1) / SPB);
    if (mode == HOP_PLAIN)
      hipLaunchKernelGGL((k_hop_generic<M, HOP_PLAIN>), dim3(grid), dim3(SPB * M), 0, s, lat, U, Ughost, in, ghost, out,
                         p, c0);
    else
      hipLaunchKernelGGL((k_hop_generic<M, HOP_SHIFTED>), dim3(grid), dim3(SPB * M), 0, s, lat, U, Ughost, in, ghost,
                         out, p, c0);
  });
}

void launch_pack_faces_half(hipStream_t s, int m, const LatticeDev& lat, int parity, const double2* f, double2* send, int x3_lo,
                            int x3_n) {
  const int ns = n_split(lat);
  if (ns == 0) return;
  int64_t work = max_face(lat) / 2 * 3 * m;
  if (x3_n > 0) work = work / lat.L[3] * x3_n;
  hipLaunchKernelGGL(k_pack_faces_half, dim3(grid_for(work, 256, 4096), 2 * ns), dim3(256), 0, s, m, lat, parity, f, send, x3_lo,
                     x3_n);
}

void launch_hop_half(hipStream_t s, int m, const LatticeDev& lat, int parity, const double2* U, const double2* Ughost,
                     const double2* in, const double2* ghost, double2* out, HopMode mode, const double2* p, double c0) {
  BCG_DISPATCH_M(m, {
    constexpr int SPB = 256 / M;
    const unsigned grid = static_cast<unsigned>((lat.V / 2 + SPB - 1) / SPB);
    if (mode == HOP_PLAIN)
      hipLaunchKernelGGL((k_hop_half<M, HOP_PLAIN>), dim3(grid), dim3(SPB * M), 0, s, lat, parity, U, Ughost, in, ghost, out, p, c0);
    else
      hipLaunchKernelGGL((k_hop_half<M, HOP_SHIFTED>), dim3(grid), dim3(SPB * M), 0, s, lat, parity, U, Ughost, in, ghost, out, p, c0);
  });
}
void launch_parity_copy(hipStream_t s, int m, const LatticeDev& lat, int parity, double2* full, double2* half, bool to_half) {
  hipLaunchKernelGGL(k_parity_copy, dim3(grid_for(lat.V / 2 * 3 * m, 256, 8192)), dim3(256), 0, s, m, lat, parity, full, half,
                     to_half ? 1 : 0);
}
void launch_fill_field_half(hipStream_t s, int m, const LatticeDev& lat, const int* gdims, int parity, double2* f, uint64_t seed) {
  GDims g{{gdims[0], gdims[1], gdims[2], gdims[3]}};
  hipLaunchKernelGGL(k_fill_field_half, dim3(grid_for(lat.V / 2 * 3 * m, 256, 8192)), dim3(256), 0, s, m, lat, g, parity, f,
                     splitmix64_host(seed));
}

void launch_rmul_generic(hipStream_t s, int m, int64_t rows, double2* y, const double2* x, const double2* Md, double b,
                         RmulMode mode) {
  BCG_DISPATCH_M(m, {
    constexpr int RPB = 256 / M;
    const int grid = grid_for(rows, RPB, 256 * 16);
    const dim3 blk(RPB * M);
    if (mode == RMUL_ADD)
      hipLaunchKernelGGL((k_rmul_generic<M, RMUL_ADD>), dim3(grid), blk, 0, s, rows, y, x, Md, b);
    else if (mode == RMUL_XPAY)
      hipLaunchKernelGGL((k_rmul_generic<M, RMUL_XPAY>), dim3(grid), blk, 0, s, rows, y, x, Md, b);
    else
      hipLaunchKernelGGL((k_rmul_generic<M, RMUL_MUL>), dim3(grid), blk, 0, s, rows, y, x, Md, b);
  });
}

void launch_trisolve_generic(hipStream_t s, int m, int64_t rows, double2* y, const double2* Rd) {
  BCG_DISPATCH_M(m, {
    constexpr int TR = (M >= 16) ? 64 : 128;
    hipLaunchKernelGGL((k_trisolve_generic<M>), dim3(grid_for(rows, TR, 256 * 8)), dim3(TR), 0, s, rows, y, Rd);
  });
}

int launch_gram_generic(hipStream_t s, int m, int64_t rows, const double2* a, const double2* b, double2* partials,
                        int max_blocks) {
  const int grid = grid_for(rows, 64, max_blocks);
  BCG_DISPATCH_M(m, { hipLaunchKernelGGL((k_gram_generic<M>), dim3(grid), dim3(256), 0, s, rows, a, b, partials); });
  return grid;
}

void launch_reduce_partials(hipStream_t s, int n_values, int n_blocks, const double2* partials, double2* out) {
  if (n_blocks > 2048)
    hipLaunchKernelGGL(k_reduce_partials<64>, dim3((n_values + 3) / 4), dim3(256), 0, s, n_values, n_blocks, partials, out);
  else
    hipLaunchKernelGGL(k_reduce_partials<8>, dim3((n_values + 31) / 32), dim3(256), 0, s, n_values, n_blocks, partials, out);
}

}  // namespace bcg
