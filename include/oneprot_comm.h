/*
 * oneprot_comm.h -- C ABI of liboneprot_comm.so: thin RCCL wrappers for the one exchange step of the OneProt contrastive hot path
 * ("comm_init / all_gather / reduce_scatter / all_reduce / destroy taking an opaque communicator handle").
 *
 * What they replace in the reference (all reached through torch.distributed / NCCL there):
 *   oneprot_comm_all_gather      torch.distributed.nn.all_gather / dist.all_gather of the features   ref src/models/components/loss.py:31-44
 *   oneprot_comm_reduce_scatter  the autograd backward of that gather (sum of the slice gradients)    ref loss.py:31-33 (torch.distributed.nn)
 *   oneprot_comm_all_reduce      Lightning DDP's gradient all-reduce (mean)                           ref configs/trainer/ddp.yaml:12
 *   oneprot_comm_send_recv       the SigLIP neighbour exchange (isend + irecv pairs)                 ref loss.py:116-154 (batch_isend_irecv)
 *   oneprot_comm_init / _destroy process-group bootstrap                                             ref src/distributed.py:41-60 + Lightning
 *
 * One communicator per process (one process per GPU); the 128-byte unique id is created on rank 0 (oneprot_comm_unique_id) and handed to the
 * other ranks by the host (file, env, torch.distributed store ...).  Every call only enqueues on `stream`; return 0 on success, -1 invalid
 * argument, -3 RCCL error.  The product's default transport is torch.distributed (backend "nccl" = the same RCCL); these entry points let a host
 * that has no torch.distributed (or wants its own stream placement) run the same exchange -- see oneprot_amd/comm.py and INTEGRATION.md.
 */
#ifndef ONEPROT_COMM_H
#define ONEPROT_COMM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ONEPROT_COMM_F32 = 0, ONEPROT_COMM_BF16 = 1 };
enum { ONEPROT_COMM_SUM = 0, ONEPROT_COMM_AVG = 1 };
#define ONEPROT_COMM_ID_BYTES 128

int oneprot_comm_unique_id(void* id_out /* ONEPROT_COMM_ID_BYTES */);
/* the calling thread's current HIP device is the rank's GPU */
int oneprot_comm_init(void** comm_out, int nranks, int rank, const void* unique_id);
int oneprot_comm_destroy(void* comm);
int oneprot_comm_nranks(void* comm);
/* recv[r * count .. (r+1) * count) = send of rank r */
int oneprot_comm_all_gather(void* comm, const void* send, void* recv, size_t count_per_rank, int dtype, void* stream);
/* recv[0 .. count) = sum over ranks of send[rank * count .. (rank+1) * count) */
int oneprot_comm_reduce_scatter(void* comm, const void* send, void* recv, size_t recv_count, int dtype, void* stream);
/* in place */
int oneprot_comm_all_reduce(void* comm, void* buf, size_t count, int dtype, int op, void* stream);
/* Point-to-point exchange over one xGMI link pair: `count` elements of `send` go to rank `to_rank` while `count` elements from rank `from_rank`
   land in `recv`, as ONE grouped RCCL operation (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd), so that every rank may call it with its own
   (to, from) pair without deadlock.  send == NULL or recv == NULL skips that half; to_rank / from_rank may be the caller itself.  Several calls
   bracketed by oneprot_comm_group_begin / _end form one group (the bidirectional SigLIP step: two sends and two receives in flight together). */
int oneprot_comm_send_recv(void* comm, const void* send, int to_rank, void* recv, int from_rank, size_t count, int dtype, void* stream);
int oneprot_comm_group_begin(void);
int oneprot_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif
