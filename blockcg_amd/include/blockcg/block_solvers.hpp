// blockcg/block_solvers.hpp -- drop-in for SBCGrQ of the reference's inc/block_solvers.hpp:91-185.
//
// Same signature and return value (number of operator applications).  The whole iteration runs inside
// libblockcg_hip.so (host control flow + m x m algebra in C++, every loop over sites a HIP kernel).
// The reference's asserts (:97-101) become exceptions; they are not compiled out.
#ifndef BLOCKCG_BLOCK_SOLVERS_HPP
#define BLOCKCG_BLOCK_SOLVERS_HPP
#include <algorithm>
#include <utility>

#include "dirac_op.hpp"
#include "fields.hpp"

// SBCGrQ inversion of (D + sigma_j) X^{sigma_j} = B
template <int N_rhs>
int SBCGrQ(std::vector<block_fermion_field<N_rhs>>& X, const block_fermion_field<N_rhs>& B, const dirac_op& D,
           std::vector<double>& sigma, double eps = 1.e-15, double eps_shifts = 1.e-15, int max_iterations = 1e6) {
  if (sigma.size() != X.size()) throw std::invalid_argument("number of shifts does not match number of solution vectors");
  std::vector<bcg_field*> Xh(X.size());
  for (size_t s = 0; s < X.size(); ++s) Xh[s] = X[s].handle();
  B.flush();
  int iterations = 0;
  blockcg::check(bcg_sbcgrq_solve(D.lat().ctx(), D.handle(), D.mass, Xh.data(), B.handle(), static_cast<int>(X.size()),
                                  sigma.data(), eps, eps_shifts, max_iterations, /*consume_B=*/0, &iterations, nullptr, nullptr),
                 D.lat().ctx(), "SBCGrQ");
  for (auto& x : X) x.device_written();
  return iterations;
}

namespace blockcg {
// The same solver, but B's storage becomes the residual block Q (the reference copies B into Q, :109): one field less
// in HBM, which is what lets the 64^3 x 128 share of a 128^4 lattice fit one MI355X.  B's contents are destroyed.
template <int N_rhs>
int SBCGrQ_consuming_source(std::vector<block_fermion_field<N_rhs>>& X, block_fermion_field<N_rhs>& B, const dirac_op& D,
                            std::vector<double>& sigma, double eps = 1.e-15, double eps_shifts = 1.e-15,
                            int max_iterations = 1e6) {
  if (sigma.size() != X.size()) throw std::invalid_argument("number of shifts does not match number of solution vectors");
  std::vector<bcg_field*> Xh(X.size());
  for (size_t s = 0; s < X.size(); ++s) Xh[s] = X[s].handle();
  B.flush();
  int iterations = 0;
  check(bcg_sbcgrq_solve(D.lat().ctx(), D.handle(), D.mass, Xh.data(), B.handle(), static_cast<int>(X.size()), sigma.data(),
                         eps, eps_shifts, max_iterations, /*consume_B=*/1, &iterations, nullptr, nullptr),
        D.lat().ctx(), "SBCGrQ");
  for (auto& x : X) x.device_written();
  B.device_written();
  return iterations;
}
// Half-volume (parity-decoupled) solve, SURVEY.md section 8f-4: dirac_op::D couples opposite site parities only
// (inc/dirac_op.hpp:14-21), so op + sigma is block diagonal in the parity and the solve splits into two on V / 2 sites,
// each the reference's SBCGrQ unchanged on half fields (half the work fields' memory).  X, B: full fields of a 4-D lattice
// with even extents on one GPU.  Returns the operator applications of the even and of the odd solve.
template <int N_rhs>
std::pair<int, int> SBCGrQ_half_volume(std::vector<block_fermion_field<N_rhs>>& X, const block_fermion_field<N_rhs>& B,
                                       const dirac_op& D, std::vector<double>& sigma, double eps = 1.e-15,
                                       double eps_shifts = 1.e-15, int max_iterations = 1e6) {
  int its[2] = {0, 0};
  for (int par = 0; par < 2; ++par) {
    block_fermion_field<N_rhs> Bp(D.lat(), par);
    Bp.restrict_from(B);
    std::vector<block_fermion_field<N_rhs>> Xp;
    Xp.reserve(X.size());
    for (size_t s = 0; s < X.size(); ++s) Xp.emplace_back(D.lat(), par);
    its[par] = SBCGrQ_consuming_source(Xp, Bp, D, sigma, eps, eps_shifts, max_iterations);
    for (size_t s = 0; s < X.size(); ++s) X[s].insert(Xp[s]);
  }
  return std::make_pair(its[0], its[1]);
}
}  // namespace blockcg

// BCG inversion of D X = B (inc/block_solvers.hpp:10-12); returns the number of operator applications
template <int N_rhs>
int BCG(block_fermion_field<N_rhs>& X, const block_fermion_field<N_rhs>& B, const dirac_op& D, double eps = 1.e-15,
        int max_iterations = 1e6) {
  B.flush();
  int iterations = 0;
  blockcg::check(bcg_bcg_solve(D.lat().ctx(), D.handle(), D.mass, X.handle(), B.handle(), eps, max_iterations, &iterations),
                 D.lat().ctx(), "BCG");
  X.device_written();
  return iterations;
}

// BCGrQ inversion of D X = B (inc/block_solvers.hpp:50-52)
template <int N_rhs>
int BCGrQ(block_fermion_field<N_rhs>& X, const block_fermion_field<N_rhs>& B, const dirac_op& D, double eps = 1.e-15,
          int max_iterations = 1e6) {
  B.flush();
  int iterations = 0;
  blockcg::check(bcg_bcgrq_solve(D.lat().ctx(), D.handle(), D.mass, X.handle(), B.handle(), eps, max_iterations, &iterations),
                 D.lat().ctx(), "BCGrQ");
  X.device_written();
  return iterations;
}

#endif
