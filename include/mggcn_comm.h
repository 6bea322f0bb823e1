/*
 * mggcn_comm.h -- C ABI of libmggcn_comm.so: the collectives of the single-process,
 * P-GPU form of the 1D row partition, over RCCL (xGMI).
 *
 * Replaces the reference's NCCL call sites (citations relative to the reference tree):
 *   ncclCommInitAll                      src/dist_matrix.hpp:26-31
 *   dist_row_dn_matrix::bcast            src/dist_matrix.hpp:458-467  (group of P ncclBroadcast)
 *   repl_dn_matrix::allreduce            src/dist_matrix.hpp:587-592  (group of P ncclAllReduce)
 *   repl_dn_matrix ctor / init broadcast src/dist_matrix.hpp:573-580, :601-609
 * plus the exchange the MI355X build prefers: ONE all-gather of the row shards instead of P
 * broadcasts (every GPU pushes to its 7 xGMI peers at once).
 *
 * Every function takes one buffer and one stream PER GPU (arrays of length P, index = device
 * ordinal in the communicator).  Enqueue-only; fail-fast (message + exit) like CHECK_NCCL
 * (src/mg_gcn.hpp:60-68).  fp32 payloads only.
 *
 * Two transports (picked at init, see mggcn_comm_transport): "rccl" -- one communicator per GPU,
 * the reference's pattern; "p2p" -- device-to-device copies (hipMemcpyPeerAsync over xGMI: copy
 * engines, no compute unit is taken from the SpMM running meanwhile; same-device copies when ranks
 * share a GPU) ordered by events PER PAIR: a receiver pulls a sender's piece as soon as that
 * sender's stream has produced it, every pull on its own per-peer stream; sums formed in rank order
 * on every GPU.  p2p is selected when two ranks share a GPU (RCCL refuses that: this is how the
 * P > 1 schedules run on a one-GPU box) or with MGGCN_COMM_TRANSPORT=p2p.  MGGCN_P2P_PUSH=1 turns
 * the copies round ("p2p-push"): a sender writes its piece into the receiver's buffer as soon as
 * that buffer is free, receivers only wait; what a sender may overwrite then depends on its own
 * copies only (mggcn_comm_release waits for nobody else).
 *
 * Two families of entry points over the same transports:
 *   all ranks   (the reference's one-host-thread model) one call takes one buffer and one stream PER
 *               GPU and issues the P per-communicator calls inside one ncclGroupStart/End;
 *   one rank    (`_rank_`) the same arrays plus `rank`: only that rank's share is issued, from that
 *               rank's enqueue thread (host/enqueue.hpp: one thread per GPU).  All P ranks must make
 *               the same sequence of calls; a rank's call may block on the host until its peers have
 *               issued theirs (p2p: until their `ready` events are recorded), never on the device.
 *
 * Kept in its own library so that a process that already hosts an RCCL (e.g. PyTorch's
 * bundled one) never loads a second copy: the one-process-per-GPU host layer uses
 * torch.distributed instead and does not link this file.
 */
#ifndef MGGCN_COMM_H_
#define MGGCN_COMM_H_

#include <stddef.h>

#include "mggcn.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mggcn_comm mggcn_comm;

/* ncclCommInitAll over devices 0..P-1 (devices == NULL) or the given ordinals. */
mggcn_comm *mggcn_comm_init_all(int P, const int *devices);
void mggcn_comm_destroy(mggcn_comm *comm);
int mggcn_comm_size(const mggcn_comm *comm);
/* "rccl", "p2p" (receivers read their pieces) or "p2p-push" (MGGCN_P2P_PUSH=1: senders write them) */
const char *mggcn_comm_transport(const mggcn_comm *comm);

/* Exchange flags (p2p transport; rccl ignores them), mggcn_comm_set_exchange_flags:
 *   MGGCN_COMM_DEFER_RELEASE  an exchange returns without making the streams wait for the READERS of what this rank
 *                             sent (default: it waits, which is NCCL's contract -- the send buffer may be overwritten
 *                             by later work on the stream).  The caller then calls mggcn_comm_release[_rank] on the
 *                             stream that will overwrite the send buffers, before that work.  All-reduce never defers.
 *   MGGCN_COMM_SKIP_SELF      all-gather: a rank's own piece is not copied into its receive buffer (the host layer's
 *                             remote blocks never read it; one copy kernel fewer next to the SpMM) */
#define MGGCN_COMM_DEFER_RELEASE 1u
#define MGGCN_COMM_SKIP_SELF 2u
void mggcn_comm_set_exchange_flags(mggcn_comm *comm, unsigned flags);
/* streams[j] (resp. `stream` of `rank`) waits until every peer has read what rank j sent in the exchanges so far.
 * No-op on the rccl transport and when nothing is outstanding. */
void mggcn_comm_release(mggcn_comm *comm, const mggcn_stream_t *streams);
void mggcn_comm_release_rank(mggcn_comm *comm, int rank, mggcn_stream_t stream);

/* recv[j] <- send_root (count floats) on every GPU j; send_root lives on GPU `root`. */
void mggcn_comm_broadcast_f32(mggcn_comm *comm, const float *send_root, float *const *recv, size_t count,
                              int root, const mggcn_stream_t *streams);
/* recv[j][i*count .. (i+1)*count) <- send[i] for all i, on every GPU j. */
void mggcn_comm_allgather_f32(mggcn_comm *comm, const float *const *send, float *const *recv, size_t count,
                              const mggcn_stream_t *streams);
/* Variable-size exchange (the halo form of the shard exchange; the reference only
 * analyses these volumes offline, test/data/prep.py:232-272): counts[j*P + k] floats go from GPU j to GPU k;
 * send[j] holds rank j's outgoing pieces in destination order, recv[k] receives in source order. */
void mggcn_comm_alltoallv_f32(mggcn_comm *comm, const float *const *send, float *const *recv,
                              const size_t *counts, const mggcn_stream_t *streams);
/* The displacement tables mggcn_comm_alltoallv_f32 derives from `counts` (host arithmetic only, no GPU; exported so
 * that a CPU test can pin them): sdis[j*P + k] = offset in send[j] of the piece for GPU k, rdis[j*P + k] = offset in
 * recv[j] of the piece from GPU k.  All three arrays hold P*P entries. */
void mggcn_comm_alltoallv_displacements(int P, const size_t *counts, size_t *sdis, size_t *rdis);
/* bufs[j] <- sum_i bufs[i], in place, on every GPU j. */
void mggcn_comm_allreduce_sum_f32(mggcn_comm *comm, float *const *bufs, size_t count,
                                  const mggcn_stream_t *streams);


/* One rank's share of the collectives above (same arrays, all P entries valid; `stream` is that rank's stream). */
void mggcn_comm_broadcast_rank_f32(mggcn_comm *comm, int rank, const float *send_root, float *const *recv,
                                   size_t count, int root, mggcn_stream_t stream);
void mggcn_comm_allgather_rank_f32(mggcn_comm *comm, int rank, const float *const *send, float *const *recv,
                                   size_t count, mggcn_stream_t stream);
void mggcn_comm_alltoallv_rank_f32(mggcn_comm *comm, int rank, const float *const *send, float *const *recv,
                                   const size_t *counts, mggcn_stream_t stream);
void mggcn_comm_allreduce_sum_rank_f32(mggcn_comm *comm, int rank, float *const *bufs, size_t count,
                                       mggcn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MGGCN_COMM_H_ */
