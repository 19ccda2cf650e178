// examples/solver_comparison.cpp -- block solver against per-column multi-shift CG on the MI355X path.
//
// Exercises the drop-in headers the way the reference's driver exercises its own (reference benchmark.cpp): a random
// 1-D lattice operator and a block of N_RHS random sources drawn from std::rand(), every column solved with SCG, then
// all columns at once with SBCGrQ, and the true residual |(A + sigma) x - b| / |b| measured for every shift with the
// field primitives (op, add, -=, real_dot / hermitian_dot).  With the default rand() seed the lattice and sources are
// the reference's, so the counts can be compared with its output for the same arguments.
//
//   usage: solver_comparison <volume> <mass> <tolerance> [<tolerance for the shifted systems>]
//
// Build: g++ -std=c++14 -O2 -I blockcg_amd/include examples/solver_comparison.cpp -L blockcg_amd/_build -lblockcg_hip
//            -Wl,-rpath,$PWD/blockcg_amd/_build -o solver_comparison
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "blockcg/block_solvers.hpp"
#include "blockcg/standard_solvers.hpp"

namespace {

constexpr int N_RHS = 12;  // block width of the reference's driver

struct Options {
  int volume = 0;
  double mass = 0.0, tol = 0.0, tol_shifted = 1e-15;
};

bool parse(int argc, char** argv, Options& o) {
  if (argc < 4 || argc > 5) return false;
  o.volume = static_cast<int>(std::atof(argv[1]));
  o.mass = std::atof(argv[2]);
  o.tol = std::atof(argv[3]);
  if (argc == 5) o.tol_shifted = std::atof(argv[4]);
  return o.volume > 0;
}

void print_row(const char* label, const std::vector<double>& v) {
  std::printf("# %s:\t", label);
  for (double x : v) std::printf("%.6e\t", x);
  std::printf("\n");
}

// worst relative residual over the columns of one block solution, for one shift
template <int N>
double worst_block_residual(const dirac_op& A, const block_fermion_field<N>& x, const block_fermion_field<N>& b,
                            const block_matrix<N>& b_norm2, double sigma, block_fermion_field<N>& work) {
  A.op(work, x);
  work.add(x, sigma);
  work -= b;
  const block_matrix<N> r2 = work.hermitian_dot(work);
  double worst = 0.0;
  for (int i = 0; i < N; ++i) worst = std::max(worst, r2(i, i).real() / b_norm2(i, i).real());
  return std::sqrt(worst);
}

}  // namespace

int main(int argc, char** argv) {
  Options opt;
  if (!parse(argc, argv, opt)) {
    std::printf("usage: %s <volume> <mass> <tolerance> [<tolerance for the shifted systems>]\n", argv[0]);
    return 1;
  }
  std::vector<double> sigma = {0, 0, 1e-10, 1e-8, 1e-6, 1e-5, 1e-4, 1e-2, 1e-1};  // the reference driver's shift list
  const int n_sigma = static_cast<int>(sigma.size());

  dirac_op A(opt.volume, opt.mass);        // links from std::rand(), drawn before the sources as in the reference
  block_fermion_field<N_RHS> B(opt.volume);
  B.setRandom();
  std::printf("# SCG per column vs SBCGrQ on MI355X: V = %d, N_rhs = %d, mass = %g, eps = %g, eps_shifts = %g\n\n", opt.volume,
              N_RHS, opt.mass, opt.tol, opt.tol_shifted);
  print_row("Shifts\t\t", sigma);
  std::printf("\n");

  // ---- one multi-shift CG per column; host element access copies the column out of the block
  long scg_calls = 0;
  std::vector<double> scg_worst(n_sigma, 0.0);
  {
    fermion_field column(opt.volume), work(opt.volume);
    std::vector<fermion_field> x(n_sigma, column);
    for (int j = 0; j < N_RHS; ++j) {
      for (int site = 0; site < opt.volume; ++site) column[site] = B[site].col(j);
      scg_calls += SCG(x, column, A, sigma, opt.tol, opt.tol_shifted);
      const double norm2 = column.real_dot(column);
      for (int s = 0; s < n_sigma; ++s) {
        A.op(work, x[s]);
        work.add(x[s], sigma[s]);
        work -= column;
        scg_worst[s] = std::max(scg_worst[s], std::sqrt(work.real_dot(work) / norm2));
      }
    }
  }
  print_row("SCG residuals", scg_worst);

  // ---- all columns at once
  std::vector<double> block_worst(n_sigma, 0.0);
  long block_calls = 0;
  {
    block_fermion_field<N_RHS> work(opt.volume);
    std::vector<block_fermion_field<N_RHS>> X(n_sigma, B);
    block_calls = static_cast<long>(N_RHS) * SBCGrQ(X, B, A, sigma, opt.tol, opt.tol_shifted);
    const block_matrix<N_RHS> b_norm2 = B.hermitian_dot(B);
    for (int s = 0; s < n_sigma; ++s) block_worst[s] = worst_block_residual<N_RHS>(A, X[s], B, b_norm2, sigma[s], work);
  }
  print_row("SBCGrQ residuals", block_worst);
  std::printf("\n# SCG_iterations:\t%ld\n# SBCGrQ_iterations:\t%ld\n", scg_calls, block_calls);
  return (block_worst[0] < 2 * opt.tol && scg_worst[0] < 2 * opt.tol) ? 0 : 2;
}
