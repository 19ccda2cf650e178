// GPU probe of the drop-in headers' half-volume solve (blockcg::SBCGrQ_half_volume): the reference's acceptance test
// (test/solvers.cpp:104-116 -- true residual of every shift and right-hand side below 2 eps), computed with the FULL-volume
// operator on the merged solution.  Exit code 0 = passed.
#include <cmath>
#include <cstdio>

#include "blockcg/block_solvers.hpp"

int main() {
  constexpr int N = 16;
  std::vector<int> dims = {8, 4, 4, 4};
  blockcg::lattice lat(dims);
  dirac_op D(lat, 0.2, /*seed=*/21);
  block_fermion_field<N> B(lat);
  B.setRandomDevice(22);
  std::vector<double> shifts = {0.0, 1e-3, 5e-2};
  const double eps = 1e-10;
  std::vector<block_fermion_field<N>> X;
  for (size_t s = 0; s < shifts.size(); ++s) X.emplace_back(lat);
  const std::pair<int, int> its = blockcg::SBCGrQ_half_volume(X, B, D, shifts, eps, eps);
  block_fermion_field<N> AX(lat);
  const block_matrix<N> b2 = B.hermitian_dot(B);
  double worst = 0.0;
  for (size_t s = 0; s < shifts.size(); ++s) {
    D.op(AX, X[s]);
    AX.add(X[s], shifts[s]);
    AX -= B;
    const block_matrix<N> r2 = AX.hermitian_dot(AX);
    worst = std::fmax(worst, std::sqrt((r2.diagonal().real().array() / b2.diagonal().array().real()).maxCoeff()));
  }
  std::printf("half-volume solve: %d + %d operator applications, worst true residual %.3e\n", its.first, its.second, worst);
  std::printf("%s\n", worst < 2 * eps ? "HALF_VOLUME_OK" : "HALF_VOLUME_FAILED");
  return worst < 2 * eps ? 0 : 1;
}
