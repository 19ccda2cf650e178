// blockcg/eigen_compat.hpp -- the Eigen names the reference's own solver templates spell out, for code written in the
// reference's style (inc/block_solvers.hpp:19,61,121-123,130) that is compiled against these drop-in headers WITHOUT Eigen:
//     Eigen::Array<double, N_rhs, 1> b_norm = delta.rowwise().norm().array();
//     using bm_alloc = Eigen::aligned_allocator<block_matrix<N_rhs>>;
// Opt-in (not included by fields.hpp): a translation unit that also includes the real Eigen must not see these.
// With it, the reference's inc/block_solvers.hpp itself -- BCG, BCGrQ, SBCGrQ, unmodified, where it lies -- compiles against
// blockcg/fields.hpp + blockcg/dirac_op.hpp and runs every field primitive on the GPU (tests/cpp/ref_solver_templates_probe.cpp).
#ifndef BLOCKCG_EIGEN_COMPAT_HPP
#define BLOCKCG_EIGEN_COMPAT_HPP
#include <cassert>  // the reference's solvers call assert() and get the declaration through Eigen's headers
#include <memory>

#include "small_matrix.hpp"

namespace Eigen {
template <class Scalar, int Rows, int Cols>
using Array = blockcg::rarray<Rows>;  // the reference only ever forms real column arrays (Cols = 1)
template <class T>
using aligned_allocator = std::allocator<T>;
}  // namespace Eigen
#endif
