// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// extern "C" entry points over oracle/oracle.hpp so that tests/ and bench.py's cpu_baseline leg can
// call the CPU restatement through ctypes.  Built by oracle/Makefile into oracle/_build/liboracle.so.
// All arrays are in the reference's host layout (see oracle/oracle.hpp header), passed as
// interleaved (re,im) doubles.
#include <chrono>
#include <cstdint>
#include <cstring>

#include "oracle.hpp"

using namespace oracle;

namespace {

template <int M>
Field<M> load_field(int64_t V, const double* p) {
  Field<M> f(V);
  std::memcpy(static_cast<void*>(f.d.data()), p, sizeof(cplx) * f.d.size());
  return f;
}
template <int M>
void store_field(const Field<M>& f, double* p) {
  std::memcpy(p, f.d.data(), sizeof(cplx) * f.d.size());
}
Mat load_mat(int m, const double* p) {
  Mat r(m);
  std::memcpy(static_cast<void*>(r.a.data()), p, sizeof(cplx) * r.a.size());
  return r;
}
void store_mat(const Mat& r, double* p) { std::memcpy(p, r.a.data(), sizeof(cplx) * r.a.size()); }
Gauge load_gauge(int ndim, const int* dims, const double* U) {
  Gauge g{Lattice(ndim, dims)};
  std::memcpy(static_cast<void*>(g.u.data()), U, sizeof(cplx) * g.u.size());
  return g;
}

#define ORC_DISPATCH(m, CALL)          \
  switch (m) {                         \
    case 1: { constexpr int M = 1; CALL; } break; \
    case 2: { constexpr int M = 2; CALL; } break; \
    case 3: { constexpr int M = 3; CALL; } break; \
    case 4: { constexpr int M = 4; CALL; } break; \
    case 5: { constexpr int M = 5; CALL; } break; \
    case 6: { constexpr int M = 6; CALL; } break; \
    case 7: { constexpr int M = 7; CALL; } break; \
    case 8: { constexpr int M = 8; CALL; } break; \
    case 9: { constexpr int M = 9; CALL; } break; \
    case 10: { constexpr int M = 10; CALL; } break; \
    case 11: { constexpr int M = 11; CALL; } break; \
    case 12: { constexpr int M = 12; CALL; } break; \
    case 13: { constexpr int M = 13; CALL; } break; \
    case 14: { constexpr int M = 14; CALL; } break; \
    case 15: { constexpr int M = 15; CALL; } break; \
    case 16: { constexpr int M = 16; CALL; } break; \
    case 17: { constexpr int M = 17; CALL; } break; \
    case 18: { constexpr int M = 18; CALL; } break; \
    case 19: { constexpr int M = 19; CALL; } break; \
    case 20: { constexpr int M = 20; CALL; } break; \
    case 21: { constexpr int M = 21; CALL; } break; \
    case 22: { constexpr int M = 22; CALL; } break; \
    case 23: { constexpr int M = 23; CALL; } break; \
    case 24: { constexpr int M = 24; CALL; } break; \
    case 25: { constexpr int M = 25; CALL; } break; \
    case 26: { constexpr int M = 26; CALL; } break; \
    case 27: { constexpr int M = 27; CALL; } break; \
    case 28: { constexpr int M = 28; CALL; } break; \
    case 29: { constexpr int M = 29; CALL; } break; \
    case 30: { constexpr int M = 30; CALL; } break; \
    case 31: { constexpr int M = 31; CALL; } break; \
    case 32: { constexpr int M = 32; CALL; } break; \
    default: return -1;                \
  }

}  // namespace

extern "C" {

int orc_fill_field(int m, int64_t nsites, int64_t first_global_site, uint64_t seed, double* out) {
  // sites [first_global_site, first_global_site + nsites) of a field whose global element
  // counters are given by oracle::field_counter
  for (int64_t s = 0; s < nsites; ++s)
    for (int j = 0; j < m; ++j)
      for (int c = 0; c < NC; ++c)
        for (int ri = 0; ri < 2; ++ri)
          out[((s * m + j) * NC + c) * 2 + ri] = uniform_pm1(seed, field_counter(first_global_site + s, m, c, j, ri));
  return 0;
}

int orc_fill_gauge(int ndim, const int* dims, uint64_t seed, double* out) {
  Gauge g{Lattice(ndim, dims)};
  g.fill_random(seed);
  std::memcpy(out, g.u.data(), sizeof(cplx) * g.u.size());
  return 0;
}

// Host threads for the site loops (oracle::threads()); 1 = the reference's sequential order.
int orc_set_threads(int n) {
  threads() = n < 1 ? 1 : n;
  return threads();
}

// Summation order of the Gram products (oracle::gram_arith()): 0 reference, 1 pairwise (device-like), 2 long double.
int orc_set_gram_arith(int mode) {
  gram_arith() = mode;
  return gram_arith();
}

// Sampled evaluator (oracle.hpp): out[k] = (D psi)(sites[k]) resp. (A psi)(sites[k]), [nsites][m][3] complex, with U and
// psi from the counter-based generator (seed_U, seed_psi); nothing of size V is allocated.
int orc_hop_sampled(int m, int ndim, const int* dims, uint64_t seed_U, uint64_t seed_psi, int64_t nsites,
                    const int64_t* sites, double* out) {
  const Lattice lat(ndim, dims);
  ORC_DISPATCH(m, {
    BLOCKCG_ORACLE_PARALLEL_FOR
    for (int64_t k = 0; k < nsites; ++k)
      hop_sampled_site<M>(reinterpret_cast<cplx*>(out) + k * M * NC, lat, seed_U, seed_psi, sites[k]);
  });
  return 0;
}
int orc_apply_sampled(int m, int ndim, const int* dims, uint64_t seed_U, uint64_t seed_psi, double mass, int64_t nsites,
                      const int64_t* sites, double* out) {
  const Lattice lat(ndim, dims);
  ORC_DISPATCH(m, {
    BLOCKCG_ORACLE_PARALLEL_FOR
    for (int64_t k = 0; k < nsites; ++k)
      apply_sampled_site<M>(reinterpret_cast<cplx*>(out) + k * M * NC, lat, seed_U, seed_psi, mass, sites[k]);
  });
  return 0;
}

// a^dagger b (m x m, column-major) of two GENERATED fields of `nsites` sites, accumulated in chunks of 4096 sites whose sums
// are added in chunk order: the block inner product at sizes too large to hold (64^4 and up), inc/fields.hpp:103-122.
int orc_gram_generated(int m, int64_t nsites, uint64_t seed_a, uint64_t seed_b, double* out) {
  ORC_DISPATCH(m, {
    constexpr int64_t kChunk = 4096;
    const int64_t nchunk = (nsites + kChunk - 1) / kChunk;
    std::vector<Mat> part(static_cast<size_t>(nchunk), Mat(M));
    BLOCKCG_ORACLE_PARALLEL_FOR
    for (int64_t ch = 0; ch < nchunk; ++ch) {
      const int64_t x0 = ch * kChunk;
      const int64_t n = std::min<int64_t>(kChunk, nsites - x0);
      Field<M> fa(n);
      Field<M> fb(n);
      for (int64_t s = 0; s < n; ++s) {
        generated_tile<M>(fa.site(s), seed_a, x0 + s);
        generated_tile<M>(fb.site(s), seed_b, x0 + s);
      }
      hermitian_dot_lower_range(part[ch], fa, fb, 0, n);
    }
    Mat R(M);
    for (int64_t ch = 0; ch < nchunk; ++ch)
      for (size_t e = 0; e < R.a.size(); ++e) R.a[e] += part[ch].a[e];
    for (int i = 1; i < M; ++i)
      for (int j = 0; j < i; ++j) R(j, i) = std::conj(R(i, j));
    store_mat(R, out);
  });
  return 0;
}

// SBCGrQ on generated inputs (gauge seed_U, source seed_B) without passing lattice-sized arrays through the caller: the
// m x m coefficients of the first `max_iterations` iterations at sizes such as 64^4 (with orc_set_threads > 1).
// Trace layout as orc_sbcgrq.  X is not returned.
int orc_sbcgrq_generated(int m, int ndim, const int* dims, uint64_t seed_U, uint64_t seed_B, double mass, int nshift,
                         const double* sigma, int max_iterations, double* tr_mats, double* tr_res, int* iters_out) {
  Gauge g{Lattice(ndim, dims)};
  g.fill_random(seed_U);
  const int64_t V = g.lat.V;
  std::vector<double> sig(sigma, sigma + nshift);
  ORC_DISPATCH(m, {
    Field<M> fB(V);
    fB.fill_random(seed_B);
    std::vector<Field<M>> X(nshift, Field<M>(V));
    std::vector<IterTrace> trace;
    const int it = SBCGrQ<M>(X, fB, g, mass, sig, 0.0, 0.0, max_iterations, &trace, max_iterations);
    if (iters_out) *iters_out = it;
    const size_t mm2 = static_cast<size_t>(m) * m * 2;
    for (size_t k = 0; k < trace.size(); ++k) {
      double* p = tr_mats + k * (3 + 2 * nshift) * mm2;
      store_mat(trace[k].alpha, p);
      store_mat(trace[k].rho, p + mm2);
      store_mat(trace[k].delta, p + 2 * mm2);
      for (int s = 0; s < nshift; ++s) {
        store_mat(trace[k].alpha_s[s], p + (3 + s) * mm2);
        store_mat(trace[k].beta_s[s], p + (3 + nshift + s) * mm2);
      }
      if (tr_res) {
        double* q = tr_res + k * (1 + nshift);
        q[0] = trace[k].residual;
        for (int s = 0; s < nshift; ++s) q[1 + s] = trace[k].residual_shift[s];
      }
    }
  });
  return 0;
}


int orc_hop(int m, int ndim, const int* dims, const double* U, const double* in, double* out) {
  Gauge g = load_gauge(ndim, dims, U);
  ORC_DISPATCH(m, {
    Field<M> fi = load_field<M>(g.lat.V, in);
    Field<M> fo(g.lat.V);
    hop(fo, fi, g);
    store_field(fo, out);
  });
  return 0;
}

int orc_dirac_apply(int m, int ndim, const int* dims, const double* U, double mass, const double* in, double* out) {
  Gauge g = load_gauge(ndim, dims, U);
  ORC_DISPATCH(m, {
    Field<M> fi = load_field<M>(g.lat.V, in);
    Field<M> fo(g.lat.V);
    dirac_apply(fo, fi, g, mass);
    store_field(fo, out);
  });
  return 0;
}

int orc_add_scalar(int m, int64_t V, double* y, const double* x, double a) {
  ORC_DISPATCH(m, {
    Field<M> fy = load_field<M>(V, y);
    Field<M> fx = load_field<M>(V, x);
    add_scalar(fy, fx, a);
    store_field(fy, y);
  });
  return 0;
}

int orc_rescale_add_scalar(int m, int64_t V, double* y, double a, const double* x, double b) {
  ORC_DISPATCH(m, {
    Field<M> fy = load_field<M>(V, y);
    Field<M> fx = load_field<M>(V, x);
    rescale_add_scalar(fy, a, fx, b);
    store_field(fy, y);
  });
  return 0;
}

int orc_add_matrix(int m, int64_t V, double* y, const double* x, const double* Mx) {
  ORC_DISPATCH(m, {
    Field<M> fy = load_field<M>(V, y);
    Field<M> fx = load_field<M>(V, x);
    add_matrix(fy, fx, load_mat(m, Mx));
    store_field(fy, y);
  });
  return 0;
}

int orc_rescale_add_matrix(int m, int64_t V, double* y, const double* Mx, const double* x, double b) {
  ORC_DISPATCH(m, {
    Field<M> fy = load_field<M>(V, y);
    Field<M> fx = load_field<M>(V, x);
    rescale_add_matrix(fy, load_mat(m, Mx), fx, b);
    store_field(fy, y);
  });
  return 0;
}

int orc_hermitian_dot(int m, int64_t V, const double* a, const double* b, double* out) {
  ORC_DISPATCH(m, {
    Field<M> fa = load_field<M>(V, a);
    Field<M> fb = load_field<M>(V, b);
    store_mat(hermitian_dot(fa, fb), out);
  });
  return 0;
}

int orc_tri_solve_rhs(int m, int64_t V, double* y, const double* R) {
  ORC_DISPATCH(m, {
    Field<M> fy = load_field<M>(V, y);
    tri_solve_rhs(fy, load_mat(m, R));
    store_field(fy, y);
  });
  return 0;
}

int orc_thin_qr(int m, int64_t V, double* y, double* R_out) {
  ORC_DISPATCH(m, {
    Field<M> fy = load_field<M>(V, y);
    Mat R(m);
    thinQR(fy, R);
    store_field(fy, y);
    store_mat(R, R_out);
  });
  return 0;
}

int orc_sub(int m, int64_t V, double* y, const double* x) {
  ORC_DISPATCH(m, {
    Field<M> fy = load_field<M>(V, y);
    Field<M> fx = load_field<M>(V, x);
    sub(fy, fx);
    store_field(fy, y);
  });
  return 0;
}

int orc_cholesky_upper(int m, const double* G, double* R) {
  store_mat(cholesky_upper(load_mat(m, G)), R);
  return 0;
}
int orc_inverse(int m, const double* A, double* Ainv) {
  store_mat(inverse_full_piv_lu(load_mat(m, A)), Ainv);
  return 0;
}

// Trace layout (all optional, pass NULL to skip), for iterations 1..trace_limit:
//   tr_mats : [iter][3 + 2*nshift][m*m] complex  = alpha, rho, delta, alpha_s[0..S), beta_s[0..S)
//   tr_res  : [iter][1 + nshift] doubles          = residual, residual_shift[0..S) (-1 = not visited)
int orc_sbcgrq(int m, int ndim, const int* dims, const double* U, double mass, const double* B, int nshift,
               const double* sigma, double eps, double eps_shifts, int max_iterations, double* X_out, int* iters_out,
               int trace_limit, double* tr_mats, double* tr_res, double* seconds_out) {
  Gauge g = load_gauge(ndim, dims, U);
  const int64_t V = g.lat.V;
  std::vector<double> sig(sigma, sigma + nshift);
  ORC_DISPATCH(m, {
    Field<M> fB = load_field<M>(V, B);
    std::vector<Field<M>> X(nshift, Field<M>(V));
    std::vector<IterTrace> trace;
    const auto t0 = std::chrono::steady_clock::now();
    const int it = SBCGrQ<M>(X, fB, g, mass, sig, eps, eps_shifts, max_iterations, trace_limit > 0 ? &trace : nullptr,
                             trace_limit);
    const auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    if (iters_out) *iters_out = it;
    if (X_out)
      for (int s = 0; s < nshift; ++s) store_field(X[s], X_out + static_cast<size_t>(s) * V * m * NC * 2);
    const size_t mm2 = static_cast<size_t>(m) * m * 2;
    for (size_t k = 0; k < trace.size(); ++k) {
      if (tr_mats) {
        double* p = tr_mats + k * (3 + 2 * nshift) * mm2;
        store_mat(trace[k].alpha, p);
        store_mat(trace[k].rho, p + mm2);
        store_mat(trace[k].delta, p + 2 * mm2);
        for (int s = 0; s < nshift; ++s) {
          store_mat(trace[k].alpha_s[s], p + (3 + s) * mm2);
          store_mat(trace[k].beta_s[s], p + (3 + nshift + s) * mm2);
        }
      }
      if (tr_res) {
        double* p = tr_res + k * (1 + nshift);
        p[0] = trace[k].residual;
        for (int s = 0; s < nshift; ++s) p[1 + s] = trace[k].residual_shift[s];
      }
    }
  });
  return 0;
}

int orc_cg(int ndim, const int* dims, const double* U, double mass, const double* b, double eps, int max_iterations,
           double* x_out, int* iters_out) {
  Gauge g = load_gauge(ndim, dims, U);
  Field<1> fb = load_field<1>(g.lat.V, b);
  Field<1> x(g.lat.V);
  const int it = CG(x, fb, g, mass, eps, max_iterations);
  store_field(x, x_out);
  if (iters_out) *iters_out = it;
  return 0;
}

int orc_scg(int ndim, const int* dims, const double* U, double mass, const double* b, int nshift, const double* sigma,
            double eps, double eps_shifts, int max_iterations, double* x_out, int* iters_out) {
  Gauge g = load_gauge(ndim, dims, U);
  const int64_t V = g.lat.V;
  Field<1> fb = load_field<1>(V, b);
  std::vector<Field<1>> x(nshift, Field<1>(V));
  const int it = SCG(x, fb, g, mass, std::vector<double>(sigma, sigma + nshift), eps, eps_shifts, max_iterations);
  for (int s = 0; s < nshift; ++s) store_field(x[s], x_out + static_cast<size_t>(s) * V * NC * 2);
  if (iters_out) *iters_out = it;
  return 0;
}

int orc_bcg(int m, int ndim, const int* dims, const double* U, double mass, const double* B, double eps, int max_iterations,
            int with_qr, double* X_out, int* iters_out) {
  Gauge g = load_gauge(ndim, dims, U);
  ORC_DISPATCH(m, {
    Field<M> fB = load_field<M>(g.lat.V, B);
    Field<M> X(g.lat.V);
    const int it = with_qr ? BCGrQ<M>(X, fB, g, mass, eps, max_iterations) : BCG<M>(X, fB, g, mass, eps, max_iterations);
    store_field(X, X_out);
    if (iters_out) *iters_out = it;
  });
  return 0;
}

// res_out[nshift][m]
int orc_true_residuals(int m, int ndim, const int* dims, const double* U, double mass, const double* B, int nshift,
                       const double* sigma, const double* X, double* res_out) {
  Gauge g = load_gauge(ndim, dims, U);
  const int64_t V = g.lat.V;
  ORC_DISPATCH(m, {
    Field<M> fB = load_field<M>(V, B);
    for (int s = 0; s < nshift; ++s) {
      Field<M> fX = load_field<M>(V, X + static_cast<size_t>(s) * V * m * NC * 2);
      const std::vector<double> r = true_residuals(fX, fB, g, mass, sigma[s]);
      for (int i = 0; i < m; ++i) res_out[s * m + i] = r[i];
    }
  });
  return 0;
}

// CPU baseline: synthetic lattice + RHS from the counter-based generator, fixed number of
// iterations (eps = eps_shifts = 0), one thread.  Returns seconds for `iterations` iterations
// (setup excluded is not possible without touching the solver, so setup -- one thinQR and S
// copies -- is included and reported separately by timing a 0-iteration call).
int orc_bench_sbcgrq(int m, int ndim, const int* dims, double mass, int nshift, const double* sigma, int iterations,
                     uint64_t seed, double* seconds_total, double* seconds_setup) {
  Gauge g{Lattice(ndim, dims)};
  g.fill_random(seed);
  const int64_t V = g.lat.V;
  std::vector<double> sig(sigma, sigma + nshift);
  ORC_DISPATCH(m, {
    Field<M> fB(V);
    fB.fill_random(seed + 1);
    std::vector<Field<M>> X(nshift, Field<M>(V));
    auto t0 = std::chrono::steady_clock::now();
    SBCGrQ<M>(X, fB, g, mass, sig, 0.0, 0.0, 0);
    auto t1 = std::chrono::steady_clock::now();
    SBCGrQ<M>(X, fB, g, mass, sig, 0.0, 0.0, iterations);
    auto t2 = std::chrono::steady_clock::now();
    if (seconds_setup) *seconds_setup = std::chrono::duration<double>(t1 - t0).count();
    if (seconds_total) *seconds_total = std::chrono::duration<double>(t2 - t1).count();
  });
  return 0;
}

}  // extern "C"
