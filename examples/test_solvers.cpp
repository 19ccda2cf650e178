// examples/test_solvers.cpp -- the reference's SBCGrQ unit test (test/solvers.cpp:93-119) against the
// drop-in headers, with the reference's parameters (:8-17) and acceptance criterion (:116).  Also runs a
// 4-D lattice through the same code.  Exit code 0 = every assertion holds.
#include <cmath>
#include <iostream>

#include "blockcg/block_solvers.hpp"

int V = 128;                               // test/solvers.cpp:8
double mass = 0.5;                         // :10
double stopping_criterion = 1.e-10;        // :12
constexpr int N_rhs = 3;                   // :14
std::vector<double> shifts = {0.0, 0.01, 0.10, 0.20, 0.9};  // :16
int N_shifts = static_cast<int>(shifts.size());

template <int N>
int check(const char* name, std::vector<block_fermion_field<N>>& X, block_fermion_field<N>& B, block_fermion_field<N>& AX,
          const dirac_op& D, int iterations) {
  int failures = 0;
  block_matrix<N> b2 = B.hermitian_dot(B);
  for (int i_shift = 0; i_shift < N_shifts; ++i_shift) {
    double shift = shifts[i_shift];
    D.op(AX, X[i_shift]);
    AX.add(X[i_shift], shift);
    AX -= B;
    block_matrix<N> r2 = AX.hermitian_dot(AX);
    for (int i_rhs = 0; i_rhs < N; ++i_rhs) {
      double residual = sqrt(r2(i_rhs, i_rhs).real() / b2(i_rhs, i_rhs).real());
      if (!(residual < 2 * stopping_criterion)) {  // REQUIRE(residual < 2 * stopping_criterion), :116
        ++failures;
        std::cout << "FAILED " << name << " shift " << shift << " rhs " << i_rhs << " residual " << residual << std::endl;
      }
    }
  }
  std::cout << name << ": iterations " << iterations << ", " << (failures ? "FAILED" : "passed") << std::endl;
  return failures;
}

int main() {
  int failures = 0;
  {  // TEST_CASE("SBCGrQ"), the reference's 1-D lattice
    block_fermion_field<N_rhs> B(V), AX(V);
    dirac_op D(V, mass);
    std::vector<block_fermion_field<N_rhs>> X(N_shifts, B);
    B.setRandom();
    int iterations = SBCGrQ(X, B, D, shifts, stopping_criterion);
    failures += check<N_rhs>("SBCGrQ 1-D V=128 N_rhs=3", X, B, AX, D, iterations);
    // element access through operator[] (benchmark.cpp:61-63)
    std::complex<double> z = B[5](1, 2);
    B[5](1, 2) = z + 1.0;
    block_fermion_field<N_rhs> C(B);
    if (std::abs(C[5](1, 2) - (z + 1.0)) > 0) {
      ++failures;
      std::cout << "FAILED host element access" << std::endl;
    }
  }
  {  // same test on a 4-D lattice, block width 16 (MFMA path)
    blockcg::lattice lat({16, 4, 4, 4});
    dirac_op D(lat, mass, 11ull);
    block_fermion_field<16> B(lat), AX(lat);
    B.setRandomDevice(12ull);
    std::vector<block_fermion_field<16>> X(N_shifts, B);
    int iterations = SBCGrQ(X, B, D, shifts, stopping_criterion);
    failures += check<16>("SBCGrQ 4-D 16x4x4x4 N_rhs=16", X, B, AX, D, iterations);
  }
  return failures ? 1 : 0;
}
