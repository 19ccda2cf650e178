"""blockcg_amd -- MI355X-native SBCGrQ block-CG hot path (HIP kernels behind a C ABI).

Python here is plumbing for tests and bench.py: ctypes bindings over include/blockcg_hip.h whose
classes mirror the reference's interface names (block_fermion_field, dirac_op, SBCGrQ).  The C++
drop-in headers live in blockcg_amd/include/blockcg/.
"""
from ._lib import build, load, LIB_PATH  # noqa: F401
from .api import (BlockCGError, Context, block_fermion_field, dirac_op, SBCGrQ, SBCGrQState, SUPPORTED_WIDTHS, true_residuals,
                  CG, SCG, BCG, BCGrQ, SBCGrQ_half_volume)  # noqa: F401
