// RCCL wrappers behind the C ABI of include/oneprot_comm.h (one communicator per process, one process per GPU, collectives over xGMI).
#include "../../include/oneprot_comm.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#define COMM_OK 0
#define COMM_EINVAL (-1)
#define COMM_ERCCL (-3)

static_assert(sizeof(ncclUniqueId) == ONEPROT_COMM_ID_BYTES, "unique id size");

static inline bool to_nccl_type(int dtype, ncclDataType_t* t) {
  if (dtype == ONEPROT_COMM_F32) { *t = ncclFloat32; return true; }
  if (dtype == ONEPROT_COMM_BF16) { *t = ncclBfloat16; return true; }
  return false;
}

extern "C" int oneprot_comm_unique_id(void* id_out) {
  if (!id_out) return COMM_EINVAL;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return COMM_ERCCL;
  memcpy(id_out, &id, sizeof(id));
  return COMM_OK;
}

extern "C" int oneprot_comm_init(void** comm_out, int nranks, int rank, const void* unique_id) {
  if (!comm_out || !unique_id || nranks <= 0 || rank < 0 || rank >= nranks) return COMM_EINVAL;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t c;
  if (ncclCommInitRank(&c, nranks, id, rank) != ncclSuccess) return COMM_ERCCL;
  *comm_out = (void*)c;
  return COMM_OK;
}

extern "C" int oneprot_comm_destroy(void* comm) {
  if (!comm) return COMM_EINVAL;
  return ncclCommDestroy((ncclComm_t)comm) == ncclSuccess ? COMM_OK : COMM_ERCCL;
}

extern "C" int oneprot_comm_nranks(void* comm) {
  int n = 0;
  if (!comm || ncclCommCount((ncclComm_t)comm, &n) != ncclSuccess) return COMM_EINVAL;
  return n;
}

extern "C" int oneprot_comm_all_gather(void* comm, const void* send, void* recv, size_t count_per_rank, int dtype, void* stream) {
  ncclDataType_t t;
  if (!comm || !send || !recv || count_per_rank == 0 || !to_nccl_type(dtype, &t)) return COMM_EINVAL;
  return ncclAllGather(send, recv, count_per_rank, t, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess ? COMM_OK : COMM_ERCCL;
}

extern "C" int oneprot_comm_reduce_scatter(void* comm, const void* send, void* recv, size_t recv_count, int dtype, void* stream) {
  ncclDataType_t t;
  if (!comm || !send || !recv || recv_count == 0 || !to_nccl_type(dtype, &t)) return COMM_EINVAL;
  return ncclReduceScatter(send, recv, recv_count, t, ncclSum, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess ? COMM_OK : COMM_ERCCL;
}

extern "C" int oneprot_comm_all_reduce(void* comm, void* buf, size_t count, int dtype, int op, void* stream) {
  ncclDataType_t t;
  if (!comm || !buf || count == 0 || !to_nccl_type(dtype, &t) || (op != ONEPROT_COMM_SUM && op != ONEPROT_COMM_AVG)) return COMM_EINVAL;
  return ncclAllReduce(buf, buf, count, t, op == ONEPROT_COMM_AVG ? ncclAvg : ncclSum, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess ? COMM_OK : COMM_ERCCL;
}

extern "C" int oneprot_comm_group_begin(void) { return ncclGroupStart() == ncclSuccess ? COMM_OK : COMM_ERCCL; }
extern "C" int oneprot_comm_group_end(void) { return ncclGroupEnd() == ncclSuccess ? COMM_OK : COMM_ERCCL; }

extern "C" int oneprot_comm_send_recv(void* comm, const void* send, int to_rank, void* recv, int from_rank, size_t count, int dtype, void* stream) {
  ncclDataType_t t;
  if (!comm || (!send && !recv) || count == 0 || !to_nccl_type(dtype, &t)) return COMM_EINVAL;
  int n = 0;
  if (ncclCommCount((ncclComm_t)comm, &n) != ncclSuccess) return COMM_ERCCL;
  if ((send && (to_rank < 0 || to_rank >= n)) || (recv && (from_rank < 0 || from_rank >= n))) return COMM_EINVAL;
  if (ncclGroupStart() != ncclSuccess) return COMM_ERCCL;
  bool ok = true;
  if (send) ok = ok && ncclSend(send, count, t, to_rank, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess;
  if (recv) ok = ok && ncclRecv(recv, count, t, from_rank, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess;
  ok = (ncclGroupEnd() == ncclSuccess) && ok;
  return ok ? COMM_OK : COMM_ERCCL;
}
