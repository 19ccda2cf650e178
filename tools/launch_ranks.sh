#!/bin/bash
# tools/launch_ranks.sh N program [args...] -- start N copies of a native (non-Python) driver, one per GPU of this node,
# with RANK / WORLD_SIZE / LOCAL_RANK set.  For examples/multi_gpu_solver.
#   * every launch gets its own BCG_RUN_TOKEN: the library's file rendezvous (bcg_rccl_unique_id_via_file) then uses
#     <idfile>.<token>, so the id a previous launch left behind at the same path is never read;
#   * as soon as one rank exits non-zero the others are killed (they would otherwise wait in RCCL for the dead peer)
#     and the script exits with that rank's code.
set -u
N=$1; shift
export BCG_RUN_TOKEN="$(date +%s%N).$$"
pids=()
for ((r = 0; r < N; ++r)); do
  RANK=$r WORLD_SIZE=$N LOCAL_RANK=${BCG_LOCAL_RANK_OVERRIDE:-$r} HSA_ENABLE_IPC_MODE_LEGACY=0 "$@" &
  pids+=($!)
done
rc=0
left=$N
while ((left > 0)); do
  wait -n -p done_pid "${pids[@]}"; code=$?
  left=$((left - 1))
  for i in "${!pids[@]}"; do [[ "${pids[$i]}" == "${done_pid:-}" ]] && unset 'pids[i]'; done
  if ((code != 0)); then
    rc=$code
    for p in "${pids[@]}"; do kill "$p" 2>/dev/null; done      # exactly the ranks started above
    sleep 2
    for p in "${pids[@]}"; do kill -9 "$p" 2>/dev/null; done
    wait 2>/dev/null
    break
  fi
done
exit $rc
