// The reference's OWN solver templates -- /root/reference/inc/block_solvers.hpp: BCG, BCGrQ, SBCGrQ, unmodified, compiled
// where they lie -- instantiated over the drop-in field, matrix and operator types, so that every field primitive they call
// (add, rescale_add, hermitian_dot, thinQR, D.op ...) is a C-ABI call into libblockcg_hip.so and every m x m expression
// (fullPivLu().solve(...), rowwise().norm().array(), Eigen::Array, Identity(), products and sums) is blockcg::cmatrix.
// This is what "a user solver written in the reference's style" looks like to these headers.
//
// How: the reference header includes "dirac_op.hpp" and "fields.hpp" by bare name, which from its own directory would find
// the reference's files; their include guards are defined here first (the same trick, the other way round, by which
// oracle/ref_harness.cpp runs the reference's solver over a substitute operator), so those includes are skipped and the
// names resolve to the drop-in types included above.  Built by `make -C oracle dropin` (needs /root/reference at build
// time); nothing of the reference is copied.
//
// usage: ref_solver_templates_probe          -> prints one line per solver; exit code 0 = all checks hold
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "blockcg/fields.hpp"
#include "blockcg/dirac_op.hpp"
#include "blockcg/eigen_compat.hpp"

#define LKEEGAN_BLOCKCG_FIELDS_H
#define LKEEGAN_BLOCKCG_DIRAC_OP_H
#include REFERENCE_BLOCK_SOLVERS  // -DREFERENCE_BLOCK_SOLVERS='"/root/reference/inc/block_solvers.hpp"'

namespace lib {  // the library's own solvers (one C-ABI call each), for the comparison: the same names in another namespace
int sbcgrq(std::vector<bcg_field*>& X, bcg_field* B, const dirac_op& D, std::vector<double>& sigma, double eps, double eps_shifts) {
  int it = 0;
  double res = 0.0;
  blockcg::check(bcg_sbcgrq_solve(D.lat().ctx(), D.handle(), D.mass, X.data(), B, static_cast<int>(X.size()), sigma.data(), eps,
                                  eps_shifts, 1000000, 0, &it, &res, nullptr),
                 D.lat().ctx(), "bcg_sbcgrq_solve");
  return it;
}
}  // namespace lib

template <int N>
static double worst_true_residual(const block_fermion_field<N>& X, const block_fermion_field<N>& B, const dirac_op& D, double sigma) {
  block_fermion_field<N> AX(B);
  D.op(AX, X);       // test/solvers.cpp:105-111
  AX.add(X, sigma);
  AX -= B;
  const block_matrix<N> r2 = AX.hermitian_dot(AX), b2 = B.hermitian_dot(B);
  return std::sqrt((r2.diagonal().real().array() / b2.diagonal().real().array()).maxCoeff());
}

int main() {
  constexpr int V = 128, N = 3;  // the reference's test configuration, test/solvers.cpp:8-17
  const double mass = 0.5, eps = 1e-10;
  std::vector<double> shifts = {0.0, 0.01, 0.10, 0.20, 0.9};
  srand(1);
  dirac_op D(V, mass);
  dirac_op D_copy(D);  // the reference's operator is copyable (links in a std::vector): so is this one
  block_fermion_field<N> B(V);
  B.setRandom();
  int bad = 0;

  {  // SBCGrQ, the reference's template (inc/block_solvers.hpp:91-185) against the library's solver on the same inputs
    std::vector<block_fermion_field<N>> X(shifts.size(), B), Xl(shifts.size(), B);
    const int it = SBCGrQ(X, B, D_copy, shifts, eps, eps);
    std::vector<bcg_field*> h;
    for (auto& x : Xl) h.push_back(x.handle());
    const int it_lib = lib::sbcgrq(h, B.handle(), D, shifts, eps, eps);
    for (auto& x : Xl) x.device_written();
    double worst = 0.0, diff = 0.0;
    for (size_t s = 0; s < shifts.size(); ++s) {
      worst = std::fmax(worst, worst_true_residual(X[s], B, D, shifts[s]));
      block_fermion_field<N> d(X[s]);
      d -= Xl[s];
      const block_matrix<N> d2 = d.hermitian_dot(d), x2 = Xl[s].hermitian_dot(Xl[s]);
      diff = std::fmax(diff, std::sqrt((d2.diagonal().real().array() / x2.diagonal().real().array()).maxCoeff()));
    }
    const bool ok = worst < 2 * eps && std::abs(it - it_lib) <= 1 && diff < 1e-8 && it >= 41 && it <= 44;
    std::printf("SBCGrQ template: iterations %d (library %d), worst true residual %.3e, |X - X_library| %.3e %s\n", it, it_lib, worst,
                diff, ok ? "ok" : "FAILED");
    bad += !ok;
  }
  {  // BCGrQ (:50-86) and BCG (:10-45)
    block_fermion_field<N> X(B), Y(B);
    const int it1 = BCGrQ(X, B, D, eps), it2 = BCG(Y, B, D, eps);
    const double r1 = worst_true_residual(X, B, D, 0.0), r2 = worst_true_residual(Y, B, D, 0.0);
    const bool ok = r1 < 2 * eps && r2 < 2 * eps && it1 >= 41 && it1 <= 44 && it2 >= 41 && it2 <= 44;
    std::printf("BCGrQ template: iterations %d residual %.3e; BCG template: iterations %d residual %.3e %s\n", it1, r1, it2, r2,
                ok ? "ok" : "FAILED");
    bad += !ok;
  }
  {  // the decompositions by themselves: A A^-1 = 1, L L^dagger = G
    block_matrix<N> A;
    A.setRandom();
    const block_matrix<N> I = block_matrix<N>::Identity();
    const block_matrix<N> Ai = A.fullPivLu().solve(I), G = A.adjoint() * A;
    const block_matrix<N> L = G.llt().matrixL(), R = G.llt().matrixL().adjoint();  // inc/fields.hpp:142
    const double e1 = (A * Ai - I).norm(), e2 = (L * L.adjoint() - G).norm() / G.norm(), e3 = (R.adjoint() * R - G).norm() / G.norm();
    const bool ok = e1 < 1e-13 && e2 < 1e-14 && e3 < 1e-14;
    std::printf("fullPivLu / llt: |A A^-1 - 1| %.2e, |L L^+ - G| %.2e %s\n", e1, e2, ok ? "ok" : "FAILED");
    bad += !ok;
  }
  return bad;
}
