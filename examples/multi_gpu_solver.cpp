// examples/multi_gpu_solver.cpp -- the reference driver's calling shape (benchmark.cpp:36-40,87-103: operator, random
// block source, SBCGrQ, true residual of every shift with the field primitives) on a 4-D lattice that is
// domain-decomposed over several MI355X, ONE PROCESS PER GPU, with no Python anywhere: the drop-in headers over
// libblockcg_hip.so, halo faces and the m x m all-reduce over libblockcg_rccl.so (RCCL, xGMI).
//
//   usage: multi_gpu_solver <idfile> L0 L1 L2 L3 G0 G1 G2 G3 [mass=0.1] [eps=1e-10] [capacity_ring=0] [half_volume=0]
//     L = GLOBAL lattice extents, G = process grid (prod G = number of ranks).  Rank, world size and the local device
//     come from RANK / WORLD_SIZE / LOCAL_RANK (set by tools/launch_ranks.sh, mpirun, srun or torch.distributed.run);
//     <idfile> is a path all ranks can see, used once to hand RCCL's 128-byte unique id from rank 0 to the others.
//   e.g. 8 GPUs, the BASELINE headline shape:
//     tools/launch_ranks.sh 8 examples/_build/multi_gpu_solver /tmp/bcg.id 128 128 128 128 2 2 2 1 0.1 1e-10 16
//   half_volume = 1: the same system as two half-volume solves, one per site parity (blockcg::SBCGrQ_half_volume; 192 GB
//   per GPU at that shape without a ring -- keep directions 0 and 3 whole, e.g. grid 1 2 4 1: the exchanges then overlap)
//
// Build: see tests/test_cpp_dropin.py::test_multi_gpu_driver_builds (g++, -lblockcg_rccl -lblockcg_hip).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/blockcg_rccl.h"
#include "blockcg/block_solvers.hpp"

namespace {
constexpr int N_RHS = 16;  // BASELINE configs 2 and 3

int env_int(const char* name, int fallback) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : fallback;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 10) {
    std::fprintf(stderr, "usage: %s <idfile> L0 L1 L2 L3 G0 G1 G2 G3 [mass] [eps] [capacity_ring] [half_volume]\n", argv[0]);
    return 2;
  }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", 0);
  std::vector<int> dims(4), grid(4), coords(4);
  int nranks = 1;
  for (int mu = 0; mu < 4; ++mu) {
    dims[mu] = std::atoi(argv[2 + mu]);
    grid[mu] = std::atoi(argv[6 + mu]);
    nranks *= grid[mu];
  }
  if (nranks != world) {
    std::fprintf(stderr, "process grid has %d ranks, WORLD_SIZE is %d\n", nranks, world);
    return 2;
  }
  for (int mu = 0, r = rank; mu < 4; ++mu) {  // rank = lexicographic index of the coordinates, direction 0 fastest
    coords[mu] = r % grid[mu];
    r /= grid[mu];
  }
  const double mass = argc > 10 ? std::atof(argv[10]) : 0.1;
  const double eps = argc > 11 ? std::atof(argv[11]) : 1e-10;
  const int ring = argc > 12 ? std::atoi(argv[12]) : 0;
  const bool half_volume = argc > 13 && std::atoi(argv[13]) != 0;
  std::vector<double> shifts = {0.0, 1e-6, 1e-4, 1e-2};

  try {
    blockcg::lattice lat(dims, device, grid, coords);
    unsigned char id[BCG_RCCL_UNIQUE_ID_BYTES];
    bcg_rccl_comm* comm = nullptr;
    if (bcg_rccl_unique_id_via_file(argv[1], rank, 300.0, id) != BCG_OK ||
        bcg_comm_rccl_create(lat.ctx(), id, rank, world, &comm) != BCG_OK) {
      std::fprintf(stderr, "rank %d: %s\n", rank, bcg_rccl_last_error(nullptr));
      return 1;
    }
    bcg_rccl_unique_id_file_done(argv[1], rank, comm);  // the rendezvous file is single-use: removed once all ranks are in
    if (ring > 0) blockcg::check(bcg_capacity_mode(lat.ctx(), ring), lat.ctx(), "bcg_capacity_mode");

    dirac_op D(lat, mass, /*seed=*/1);                         // benchmark.cpp:36
    block_fermion_field<N_RHS> B(lat);                         // :39-40
    B.setRandomDevice(2);
    std::vector<block_fermion_field<N_RHS>> X;                 // :88 (the reference copies B; the contents are overwritten)
    X.reserve(shifts.size());
    for (size_t s = 0; s < shifts.size(); ++s) X.emplace_back(lat);
    bcg_rccl_barrier(comm);
    const auto t0 = std::chrono::steady_clock::now();
    // :89-90.  In capacity mode the source's storage becomes the residual block (one field less in HBM) ...
    int iterations;
    if (half_volume) {  // D couples opposite site parities only (inc/dirac_op.hpp:14-21): one solve per parity on half fields
      const std::pair<int, int> its = blockcg::SBCGrQ_half_volume(X, B, D, shifts, eps, eps);
      iterations = its.first > its.second ? its.first : its.second;
    } else {
      iterations = ring > 0 ? blockcg::SBCGrQ_consuming_source(X, B, D, shifts, eps, eps) : SBCGrQ(X, B, D, shifts, eps, eps);
    }
    double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bcg_rccl_max_double(comm, &seconds);
    if (ring > 0) B.setRandomDevice(2);  // ... so draw the same source again for the residual check

    // true residuals, :93-103 (hermitian_dot is summed over all ranks inside the library)
    block_fermion_field<N_RHS> AX(lat);
    const block_matrix<N_RHS> b2 = B.hermitian_dot(B);
    double worst = 0.0;
    if (rank == 0) std::printf("# SBCGrQ residuals:\t");
    for (size_t s = 0; s < shifts.size(); ++s) {
      D.op(AX, X[s]);
      AX.add(X[s], shifts[s]);
      AX -= B;
      const block_matrix<N_RHS> r2 = AX.hermitian_dot(AX);
      const double res = std::sqrt((r2.diagonal().real().array() / b2.diagonal().array().real()).maxCoeff());
      worst = std::fmax(worst, res);
      if (rank == 0) std::printf("%.6e\t", res);
    }
    if (rank == 0) {
      std::printf("\n# lattice %dx%dx%dx%d on a %dx%dx%dx%d process grid, N_rhs = %d, %zu shifts\n", dims[0], dims[1], dims[2],
                  dims[3], grid[0], grid[1], grid[2], grid[3], N_RHS, shifts.size());
      std::printf("# SBCGrQ_iterations:\t%d\n# seconds:\t%.3f\n# iterations_per_second:\t%.3f\n", iterations, seconds,
                  iterations / seconds);
    }
    const bool ok = worst < 2.0 * eps;
    bcg_comm_rccl_destroy(comm);
    return ok ? 0 : 3;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
    return 1;
  }
}
