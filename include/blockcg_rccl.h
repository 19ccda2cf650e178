/* blockcg_rccl.h -- C ABI of libblockcg_rccl.so: the bcg_comm callbacks of include/blockcg_hip.h implemented
 * directly on RCCL, for hosts that are not Python (the reference's host is C++: benchmark.cpp:36-40,87-103 is the
 * calling shape; SURVEY.md section 8e: pairwise halo exchange = ncclSend/ncclRecv in a group, block dot-products
 * finished with an all-reduce of 2 m^2 doubles).  No torch, no Python: one process per GPU links libblockcg_hip.so,
 * this library and librccl.so.
 *
 * Usage, one process per GPU (examples/multi_gpu_solver.cpp):
 *     rank 0:  bcg_rccl_get_unique_id(id)            and hands the 128 bytes to every rank (file, socket, MPI ...)
 *     all   :  bcg_context_create(&ctx, device, NULL, ndim, global_dims, grid, coords)
 *              bcg_comm_rccl_create(ctx, id, rank, world, &comm)   -- installs the callbacks on ctx
 *              ... solve ...
 *              bcg_comm_rccl_destroy(comm)  before  bcg_context_destroy(ctx)
 * Ranks are the lexicographic index of the process-grid coordinates with direction 0 fastest, the numbering
 * bcg_halo_plan uses for its peers.
 *
 * Stream order: the all-reduce and the blocking halo exchange are enqueued on the context's stream.  The split
 * exchange (halo_exchange_begin/end) runs on a second, higher-priority stream of this library: begin makes that
 * stream wait for the packed faces (an event on the context's stream) and posts the grouped sends/receives there;
 * end makes the context's stream wait for their completion.  The host never blocks in a callback.
 * Each stream has a communicator of its own (the second one is ncclCommSplit of the first with a single colour, so one
 * unique id still suffices): RCCL serialises launches of ONE communicator across streams with its own events, which
 * would order an in-flight face exchange against the Gram all-reduce behind the caller's back.
 */
#ifndef BLOCKCG_RCCL_H
#define BLOCKCG_RCCL_H

#include "blockcg_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BCG_RCCL_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct bcg_rccl_comm bcg_rccl_comm;

/* ncclGetUniqueId: call on ONE rank, distribute the bytes to all. */
int bcg_rccl_get_unique_id(void* id_bytes_out);
/* File rendezvous for hosts without another channel: rank 0 creates the id and writes it to `path` (atomically, via a
 * temporary name); the other ranks poll for the file (up to timeout_s seconds).  All ranks return the same id. */
int bcg_rccl_unique_id_via_file(const char* path, int rank, double timeout_s, void* id_bytes_out);
/* The rendezvous is SINGLE-USE: once bcg_comm_rccl_create has returned on every rank, call this on every rank; it
 * barriers over the communicator and rank 0 removes the file, so that the same path can serve the next launch.  A launch
 * that died before this point leaves the file behind; a launcher therefore also exports a per-launch BCG_RUN_TOKEN
 * (tools/launch_ranks.sh does): when that variable is set both functions use `path.<token>` instead of `path`, and a
 * stale file of another launch is never read. */
int bcg_rccl_unique_id_file_done(const char* path, int rank, bcg_rccl_comm* comm);
/* ncclCommInitRank on the context's device, then bcg_context_set_comm(ctx, the RCCL callbacks).  Collective over all
 * `world` ranks.  world == 1 is allowed (the callbacks then only ever see messages to self). */
int bcg_comm_rccl_create(bcg_context* ctx, const void* unique_id_bytes, int rank, int world, bcg_rccl_comm** out);
/* The callback table that was installed (for callers that want to wrap or inspect it). */
const bcg_comm* bcg_comm_rccl_callbacks(const bcg_rccl_comm* comm);
/* How many RCCL communicators this transport drives: 2 (one per stream, the default) or 1 (BCG_RCCL_SINGLE_COMM=1). */
int bcg_comm_rccl_communicators(const bcg_rccl_comm* comm);
/* Host-side helpers a multi-process driver needs around its timed region: a barrier (all-reduce of one element +
 * stream synchronize) and the maximum of a host double over all ranks. */
int bcg_rccl_barrier(bcg_rccl_comm* comm);
int bcg_rccl_max_double(bcg_rccl_comm* comm, double* value_inout);
int bcg_rccl_sum_double(bcg_rccl_comm* comm, double* value_inout);
/* Contact every peer of the face exchange once on each communicator and run one all-reduce, so that RCCL sets up its
 * per-peer channels and device buffers NOW (it does so at the first transfer to a peer, not at ncclCommInitRank).  A
 * launcher that sizes its run by the free device memory (bench.py's ladder) calls this first, with peer_send / peer_recv
 * of bcg_halo_plan (one 256-byte word per message of the plan, in the plan's posting order).  Collective. */
int bcg_rccl_warm_up(bcg_rccl_comm* comm, int n_msgs, const int* peer_send, const int* peer_recv);
int bcg_comm_rccl_destroy(bcg_rccl_comm* comm);
const char* bcg_rccl_last_error(const bcg_rccl_comm* comm); /* comm may be NULL: last creation error */

#ifdef __cplusplus
}
#endif
#endif /* BLOCKCG_RCCL_H */
