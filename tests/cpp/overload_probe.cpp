// GPU probe of the drop-in field type's multiplier overloads (the reference templates both multipliers of add / rescale_add,
// inc/fields.hpp:69-90): every (double | N x N) combination against the same expression evaluated site by site on the host
// through operator[] (the lazily synchronised host mirror).  Exit code 0 = all within 1e-13.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "blockcg/fields.hpp"

namespace {
constexpr int N = 4, V = 48;

double max_diff(const block_fermion_field<N>& got, const std::vector<block_fermion<N>>& want) {
  double d = 0.0;
  for (int x = 0; x < V; ++x)
    for (int k = 0; k < 3 * N; ++k) d = std::fmax(d, std::abs(got[x].data()[k] - want[x].data()[k]));
  return d;
}
}  // namespace

int main() {
  std::srand(7);
  block_fermion_field<N> Y(V), B(V);
  Y.setRandom();
  B.setRandom();
  block_matrix<N> M1, M2;
  M1.setRandom();
  M2.setRandom();
  std::vector<block_fermion<N>> y0(V), b0(V);
  for (int x = 0; x < V; ++x) { y0[x] = Y[x]; b0[x] = B[x]; }
  double worst = 0.0;
  auto check = [&](const char* what, const block_fermion_field<N>& got, const std::vector<block_fermion<N>>& want) {
    const double d = max_diff(got, want);
    std::printf("%-34s max |diff| = %.3e\n", what, d);
    worst = std::fmax(worst, d);
  };
  std::vector<block_fermion<N>> want(V);
  {  // add(rhs, double)
    block_fermion_field<N> T(Y);
    T.add(B, 0.3);
    for (int x = 0; x < V; ++x) want[x] = y0[x] + 0.3 * b0[x];
    check("add(rhs, double)", T, want);
  }
  {  // add(rhs, N x N)
    block_fermion_field<N> T(Y);
    T.add(B, M1);
    for (int x = 0; x < V; ++x) want[x] = y0[x] + b0[x] * M1;
    check("add(rhs, matrix)", T, want);
  }
  {  // rescale_add(double, rhs, double)
    block_fermion_field<N> T(Y);
    T.rescale_add(-0.7, B, 0.25);
    for (int x = 0; x < V; ++x) want[x] = -0.7 * y0[x] + 0.25 * b0[x];
    check("rescale_add(double, rhs, double)", T, want);
  }
  {  // rescale_add(N x N, rhs, double)
    block_fermion_field<N> T(Y);
    T.rescale_add(M1, B, 1.0);
    for (int x = 0; x < V; ++x) want[x] = y0[x] * M1 + b0[x];
    check("rescale_add(matrix, rhs, double)", T, want);
  }
  {  // rescale_add(double, rhs, N x N)
    block_fermion_field<N> T(Y);
    T.rescale_add(0.6, B, M2);
    for (int x = 0; x < V; ++x) want[x] = 0.6 * y0[x] + b0[x] * M2;
    check("rescale_add(double, rhs, matrix)", T, want);
  }
  {  // rescale_add(N x N, rhs, N x N)
    block_fermion_field<N> T(Y);
    T.rescale_add(M1, B, M2);
    for (int x = 0; x < V; ++x) want[x] = y0[x] * M1 + b0[x] * M2;
    check("rescale_add(matrix, rhs, matrix)", T, want);
  }
  {  // rhs == *this, every multiplier combination (the reference's templates accept it: inc/fields.hpp:70-90)
    block_fermion_field<N> T(Y);
    T.add(T, 0.3);
    for (int x = 0; x < V; ++x) want[x] = y0[x] + 0.3 * y0[x];
    check("add(*this, double)", T, want);
    block_fermion_field<N> T2(Y);
    T2.add(T2, M1);
    for (int x = 0; x < V; ++x) want[x] = y0[x] + y0[x] * M1;
    check("add(*this, matrix)", T2, want);
    block_fermion_field<N> T3(Y);
    T3.rescale_add(M1, T3, 0.5);
    for (int x = 0; x < V; ++x) want[x] = y0[x] * M1 + 0.5 * y0[x];
    check("rescale_add(matrix, *this, double)", T3, want);
    block_fermion_field<N> T4(Y);
    T4.rescale_add(0.6, T4, M2);
    for (int x = 0; x < V; ++x) want[x] = 0.6 * y0[x] + y0[x] * M2;
    check("rescale_add(double, *this, matrix)", T4, want);
    block_fermion_field<N> T5(Y);
    T5.rescale_add(M1, T5, M2);
    for (int x = 0; x < V; ++x) want[x] = y0[x] * M1 + y0[x] * M2;
    check("rescale_add(matrix, *this, matrix)", T5, want);
  }
  {  // element write through operator[] reaches the device before the next device operation
    block_fermion_field<N> T(Y);
    T[5](1, 2) = blockcg::cplx(3.0, -4.0);
    T += B;
    for (int x = 0; x < V; ++x) want[x] = y0[x] + b0[x];
    want[5](1, 2) = blockcg::cplx(3.0, -4.0) + b0[5](1, 2);
    check("operator[] write, then +=", T, want);
  }
  std::printf("%s\n", worst < 1e-13 ? "OVERLOADS_OK" : "OVERLOADS_MISMATCH");
  return worst < 1e-13 ? 0 : 1;
}
