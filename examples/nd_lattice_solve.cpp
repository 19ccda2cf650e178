// examples/nd_lattice_solve.cpp -- what the drop-in headers add to the reference's interface: lattices of up to four
// dimensions (blockcg::lattice), device-side sources, any block width, host element access on a device field, and the
// half-volume form of the solve.  (The reference's own 1-D test configuration is run by the reference's own files:
// `make -C oracle dropin` builds its test/solvers.cpp, benchmark.cpp and inc/block_solvers.hpp against these headers.)
// Prints one line per case; exit code 0 = every true residual is below 2 x the stopping criterion.
#include <cmath>
#include <cstdio>
#include <vector>

#include "blockcg/block_solvers.hpp"

namespace {
const double kMass = 0.5, kEps = 1e-10;
const std::vector<double> kShifts = {0.0, 0.05, 0.4};

// max over shifts and right-hand sides of |(A + sigma_s) X_s - B|_i / |B|_i, with the operator of the lattice
template <int N>
double worst_residual(std::vector<block_fermion_field<N>>& X, const block_fermion_field<N>& B, const dirac_op& D) {
  block_fermion_field<N> R(B);
  const block_matrix<N> b2 = B.hermitian_dot(B);
  double worst = 0.0;
  for (size_t s = 0; s < kShifts.size(); ++s) {
    D.op(R, X[s]);
    R.add(X[s], kShifts[s]);
    R -= B;
    const block_matrix<N> r2 = R.hermitian_dot(R);
    worst = std::fmax(worst, std::sqrt((r2.diagonal().real().array() / b2.diagonal().real().array()).maxCoeff()));
  }
  return worst;
}

template <int N>
int solve_on(blockcg::lattice& lat, const char* what, bool half_volume) {
  dirac_op D(lat, kMass, /*seed=*/11ull);
  block_fermion_field<N> B(lat);
  B.setRandomDevice(12ull);
  std::vector<block_fermion_field<N>> X(kShifts.size(), B);
  std::vector<double> sigma = kShifts;
  int it = 0, it_odd = 0;
  if (half_volume) {
    const std::pair<int, int> its = blockcg::SBCGrQ_half_volume(X, B, D, sigma, kEps, kEps);
    it = its.first;
    it_odd = its.second;
  } else {
    it = SBCGrQ(X, B, D, sigma, kEps, kEps);
  }
  const double worst = worst_residual<N>(X, B, D);
  const bool ok = worst < 2 * kEps && it > 0;
  std::printf("%s N_rhs=%d%s: iterations %d%s, worst true residual %.3e %s\n", what, N, half_volume ? " (two half-volume solves)" : "",
              it, half_volume ? (std::string(" + ") + std::to_string(it_odd)).c_str() : "", worst, ok ? "passed" : "FAILED");
  return ok ? 0 : 1;
}
}  // namespace

int main() {
  int failures = 0;
  blockcg::lattice four_d({16, 4, 4, 4}), three_d({6, 4, 10});
  failures += solve_on<16>(four_d, "4-D 16x4x4x4", false);  // MFMA row kernels, specialised stencil
  failures += solve_on<5>(four_d, "4-D 16x4x4x4", false);   // an odd width: generic kernels
  failures += solve_on<16>(four_d, "4-D 16x4x4x4", true);   // the parity-decoupled form of the same solve
  failures += solve_on<7>(three_d, "3-D 6x4x10", false);
  {  // host element access on a device-resident field: read, write, and the write reaches a copy made afterwards
    block_fermion_field<3> F(four_d);
    F.setRandomDevice(5ull);
    const std::complex<double> z = F[37](1, 2);
    F[37](1, 2) = z + 1.0;
    const block_fermion_field<3> G(F);
    const bool ok = std::abs(G[37](1, 2) - (z + 1.0)) == 0.0 && std::abs(z) > 0.0;
    std::printf("host element access %s\n", ok ? "passed" : "FAILED");
    failures += ok ? 0 : 1;
  }
  return failures ? 1 : 0;
}
