#!/bin/bash
# tools/launch_ranks.sh N program [args...] -- start N copies of a native (non-Python) driver, one per GPU of this node,
# with RANK / WORLD_SIZE / LOCAL_RANK set; exits non-zero if any rank fails.  For examples/multi_gpu_solver.
set -u
N=$1; shift
pids=()
for ((r = 0; r < N; ++r)); do
  RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r HSA_ENABLE_IPC_MODE_LEGACY=0 "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
