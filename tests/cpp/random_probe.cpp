// Prints, as raw doubles, what the drop-in headers draw from std::rand() after srand(seed):
// V 3x3 links (dirac_op's constructor order) followed by a block_fermion<N> field of V sites.
#include <cstdio>
#include <cstdlib>

#include "blockcg/small_matrix.hpp"

int main(int argc, char** argv) {
  const unsigned seed = argc > 1 ? atoi(argv[1]) : 1;
  const int V = argc > 2 ? atoi(argv[2]) : 8;
  std::srand(seed);
  for (int ix = 0; ix < V; ++ix) {
    blockcg::cmatrix<3, 3> u;
    u.setRandom();
    fwrite(u.data(), sizeof(double), 18, stdout);
  }
  for (int ix = 0; ix < V; ++ix) {
    blockcg::cmatrix<3, 4> b;
    b.setRandom();
    fwrite(b.data(), sizeof(double), 24, stdout);
  }
  return 0;
}
