// examples/benchmark.cpp -- the SBCGrQ half of the reference's benchmark.cpp against the drop-in headers.
//
// Written like benchmark.cpp:10-110 (same arguments, same N_rhs = 12 and nine shifts, same residual
// measurement with op / add / -= / hermitian_dot), including the per-column SCG comparison (:55-84).
// Build (host compiler only, links the C ABI):
//   g++ -std=c++14 -O2 -I blockcg_amd/include examples/benchmark.cpp -L blockcg_amd/_build -lblockcg_hip
//       -Wl,-rpath,$PWD/blockcg_amd/_build -o benchmark
#include <cmath>
#include <iostream>

#include "blockcg/block_solvers.hpp"
#include "blockcg/standard_solvers.hpp"

constexpr int N_rhs = 12;  // benchmark.cpp:8

int main(int argc, char* argv[]) {
  std::vector<double> shifts = {0, 0, 1e-10, 1e-8, 1e-6, 1e-5, 1e-4, 1e-2, 1e-1};  // :12-13
  int N_shifts = static_cast<int>(shifts.size());
  constexpr int n_args = 3;
  if (argc - 1 < n_args) {
    std::cout << "This program requires at least " << n_args << " arguments:" << std::endl;
    std::cout << "Lattice volume, Dirac operator mass, solver stopping criterion, [solver shifts stopping criterion = 1e-15] "
              << std::endl;
    std::cout << "e.g. ./benchmark 1024 0.01 1e-12" << std::endl;
    return 1;
  }
  int V = static_cast<int>(atof(argv[1]));
  double mass = static_cast<double>(atof(argv[2]));
  double stopping_criterion = static_cast<double>(atof(argv[3]));
  double stopping_criterion_shifts = 1.e-15;
  if (argc - 1 == 4) stopping_criterion_shifts = static_cast<double>(atof(argv[4]));

  dirac_op D(V, mass);                 // :36
  block_fermion_field<N_rhs> B(V);     // :39
  B.setRandom();                       // :40
  std::cout << "# Benchmark of SBCGrQ solver (MI355X): V = " << V << ", N_rhs = " << N_rhs << ", mass = " << mass
            << ", eps = " << stopping_criterion << ", eps_shifts = " << stopping_criterion_shifts << std::endl
            << std::endl;

  // do SCG solve for each RHS of B separately (:55-84)
  std::vector<double> resSCG(N_shifts, 0.0);
  fermion_field b(V), Ax(V);
  std::vector<fermion_field> x(N_shifts, b);
  int iterSCG = 0;
  for (int i_rhs = 0; i_rhs < N_rhs; ++i_rhs) {
    for (int i_x = 0; i_x < V; ++i_x) b[i_x] = B[i_x].col(i_rhs);  // :61-63 (host element access)
    iterSCG += SCG(x, b, D, shifts, stopping_criterion, stopping_criterion_shifts);
    for (int i_shift = 0; i_shift < N_shifts; ++i_shift) {
      D.op(Ax, x[i_shift]);
      Ax.add(x[i_shift], shifts[i_shift]);
      Ax -= b;
      double residual = sqrt(Ax.real_dot(Ax) / b.real_dot(b));
      if (residual > resSCG[i_shift]) resSCG[i_shift] = residual;
    }
  }
  std::cout << "# SCG residuals:\t";
  for (int i_shift = 0; i_shift < N_shifts; ++i_shift) std::cout << std::scientific << resSCG[i_shift] << "\t";
  std::cout << std::endl;

  block_fermion_field<N_rhs> AX(V);                                  // :87
  std::vector<block_fermion_field<N_rhs>> X(N_shifts, B);            // :88
  int iterSBCGrQ = N_rhs * SBCGrQ(X, B, D, shifts, stopping_criterion, stopping_criterion_shifts);  // :89-90
  std::cout << "# SBCGrQ residuals:\t";
  block_matrix<N_rhs> b2 = B.hermitian_dot(B);                       // :93
  double worst = 0.0;
  for (int i_shift = 0; i_shift < N_shifts; ++i_shift) {
    double shift = shifts[i_shift];
    D.op(AX, X[i_shift]);                                            // :96
    AX.add(X[i_shift], shift);                                       // :97
    AX -= B;                                                         // :98
    block_matrix<N_rhs> r2 = AX.hermitian_dot(AX);                   // :99
    double res2 = 0.0;                                               // :100-101
    for (int i = 0; i < N_rhs; ++i) res2 = std::max(res2, r2(i, i).real() / b2(i, i).real());
    std::cout << std::scientific << sqrt(res2) << "\t";
    if (i_shift == 0) worst = sqrt(res2);
  }
  std::cout << std::endl << std::endl;
  std::cout << "# SCG_iterations:\t" << iterSCG << std::endl;        // :107
  std::cout << "# SBCGrQ_iterations:\t" << iterSBCGrQ << std::endl;  // :108
  return worst < 2 * stopping_criterion ? 0 : 2;
}
