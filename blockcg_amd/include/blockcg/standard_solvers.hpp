// blockcg/standard_solvers.hpp -- drop-in for the reference's inc/standard_solvers.hpp (CG, SCG) on MI355X.
// Same signatures (inc/standard_solvers.hpp:10-11,18-20); the loops run in libblockcg_hip.so on the
// width-1 kernels (src/standard_solvers.cpp:3-95 restated in blockcg_amd/csrc/capi_solvers.hip).
#ifndef BLOCKCG_STANDARD_SOLVERS_HPP
#define BLOCKCG_STANDARD_SOLVERS_HPP
#include "dirac_op.hpp"
#include "fields.hpp"

// CG inversion of D x = b; stops when |Dx - b| / |b| < eps; returns the number of operator applications
inline int CG(fermion_field& x, const fermion_field& b, const dirac_op& D, double eps = 1.e-15, int max_iterations = 1e6) {
  b.flush();
  int iterations = 0;
  blockcg::check(bcg_cg_solve(D.lat().ctx(), D.handle(), D.mass, x.handle(), b.handle(), eps, max_iterations, &iterations),
                 D.lat().ctx(), "CG");
  x.device_written();
  return iterations;
}

// SCG inversion of (D + sigma_i) x^sigma_i = b; shifts non-negative and ascending
inline int SCG(std::vector<fermion_field>& x, const fermion_field& b, const dirac_op& D, std::vector<double>& sigma,
               double eps = 1.e-15, double eps_shifts = 1.e-15, int max_iterations = 1e6) {
  if (sigma.size() != x.size()) throw std::invalid_argument("number of shifts does not match number of solution vectors");
  std::vector<bcg_field*> xh(x.size());
  for (size_t s = 0; s < x.size(); ++s) xh[s] = x[s].handle();
  b.flush();
  int iterations = 0;
  blockcg::check(bcg_scg_solve(D.lat().ctx(), D.handle(), D.mass, xh.data(), b.handle(), static_cast<int>(x.size()),
                               sigma.data(), eps, eps_shifts, max_iterations, &iterations),
                 D.lat().ctx(), "SCG");
  for (auto& v : x) v.device_written();
  return iterations;
}

#endif
