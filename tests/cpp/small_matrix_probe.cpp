// Host-only probe of the product's m x m algebra (blockcg_amd/csrc/small_matrix.hpp): reads a complex m x m
// matrix A (column-major, raw doubles) from stdin and writes, as raw doubles,
//   chol_upper(A^dagger A + m I), inverse_full_pivot(A), upper_triangular_inverse(chol), (A * A^dagger), row norms of A.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "small_matrix.hpp"

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 4;
  std::vector<double> buf(2 * m * m);
  if (fread(buf.data(), sizeof(double), buf.size(), stdin) != buf.size()) return 1;
  bcg::CMat A(m, buf.data());
  bcg::CMat G = A.adjoint() * A + static_cast<double>(m) * bcg::CMat::identity(m);
  bcg::CMat R;
  const bool ok = bcg::cholesky_upper(G, R);
  const bcg::CMat inv = bcg::inverse_full_pivot(A);
  const bcg::CMat rinv = bcg::upper_triangular_inverse(R);
  const bcg::CMat prod = A * A.adjoint();
  const bcg::CMat* outs[4] = {&R, &inv, &rinv, &prod};
  for (const bcg::CMat* M : outs) {
    M->store(buf.data());
    fwrite(buf.data(), sizeof(double), buf.size(), stdout);
  }
  const std::vector<double> rn = A.row_norms();
  fwrite(rn.data(), sizeof(double), rn.size(), stdout);
  const double okd = ok ? 1.0 : 0.0;
  fwrite(&okd, sizeof(double), 1, stdout);
  return 0;
}
