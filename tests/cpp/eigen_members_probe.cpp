// Host-only probe of the Eigen-style decomposition members of the drop-in matrix type (blockcg/small_matrix.hpp):
//   A.fullPivLu().solve(B), A.inverse()          (inc/block_solvers.hpp:31,36,73,142,166)
//   G.llt().matrixL().adjoint()                  (inc/fields.hpp:142)
// usage: eigen_members_probe N < [A, B, G as raw column-major complex doubles]  > [solve, inverse, R]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "blockcg/eigen_compat.hpp"

template <int N>
int run() {
  using M = blockcg::cmatrix<N, N>;
  M A, B, G;
  const size_t n = 2 * N * N;
  if (fread(A.data(), sizeof(double), n, stdin) != n || fread(B.data(), sizeof(double), n, stdin) != n ||
      fread(G.data(), sizeof(double), n, stdin) != n)
    return 1;
  const M X = A.fullPivLu().solve(B), Ai = A.inverse(), R = G.llt().matrixL().adjoint();
  Eigen::Array<double, N, 1> rn = A.rowwise().norm().array();  // the compat alias, as inc/block_solvers.hpp:130 spells it
  fwrite(X.data(), sizeof(double), n, stdout);
  fwrite(Ai.data(), sizeof(double), n, stdout);
  fwrite(R.data(), sizeof(double), n, stdout);
  fwrite(&rn(0), sizeof(double), N, stdout);
  return 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 5;
  if (n == 3) return run<3>();
  if (n == 5) return run<5>();
  if (n == 12) return run<12>();
  return 2;
}
