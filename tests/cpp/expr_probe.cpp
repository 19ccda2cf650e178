// Host-only probe of the Eigen-style expression members of blockcg::cmatrix that the reference's drivers use
// (benchmark.cpp:100-101, inc/block_solvers.hpp:20,42,62,83).  Exit code 0 = all identities hold.
#include <cmath>
#include <cstdio>

#include "blockcg/small_matrix.hpp"

int main() {
  constexpr int N = 5;
  blockcg::cmatrix<N, N> r2, b2;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      r2(i, j) = blockcg::cplx(1.0 + i * j, 0.25 * (i - j));
      b2(i, j) = blockcg::cplx(2.0 + i + j, -0.5 * (i - j));
    }
  double want = 0.0, want_sqrt = 0.0, want_row = 0.0;
  for (int i = 0; i < N; ++i) {
    want = std::fmax(want, r2(i, i).real() / b2(i, i).real());
    want_sqrt = std::fmax(want_sqrt, std::sqrt(r2(i, i).real()) / std::sqrt(b2(i, i).real()));
    double s = 0.0, t = 0.0;
    for (int j = 0; j < N; ++j) {
      s += std::norm(r2(i, j));
      t += std::norm(b2(i, j));
    }
    want_row = std::fmax(want_row, std::sqrt(s) / std::sqrt(t));
  }
  const double res2 = (r2.diagonal().real().array() / b2.diagonal().array().real()).maxCoeff();   // benchmark.cpp:100-101
  const blockcg::rarray<N> norms = b2.diagonal().real().array().sqrt();                           // block_solvers.hpp:19-20
  const double res = (r2.diagonal().real().array().sqrt() / norms).maxCoeff();                    // :41-42
  const blockcg::rarray<N> b_norm = b2.rowwise().norm().array();                                  // :62, :130
  const double res_row = (r2.rowwise().norm().array() / b_norm).maxCoeff();                       // :83, :155
  const bool ok = res2 == want && res == want_sqrt && std::fabs(res_row - want_row) < 1e-15;
  std::printf("%s %.17g %.17g %.17g\n", ok ? "ok" : "MISMATCH", res2, res, res_row);
  return ok ? 0 : 1;
}
