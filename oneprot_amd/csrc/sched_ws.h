// The "sched workspace": a small block of caller-provided device memory through which the persistent one-work-group-per-CU kernels hand out their work
// at run time and keep per-launch bookkeeping ON THE DEVICE (include/oneprot_hip.h: oneprot_sched_workspace_bytes / _init, oneprot_alloc_uncached).
//
//   words   0 .. 255   eight per-XCD queue heads, one 128-byte line each (HEAD(x)): the next undrawn tile / slab of XCD x's share of a launch
//   word  256          DONE: work-groups that have left the current launch.  The LAST one to leave resets the heads and DONE to zero and adds one to
//   word  288          EPOCH: launches completed on this workspace.  A kernel reads it once when it starts: epoch + 1 is the launch's tag (the tagged
//                      partial statistics of epilogue_resid_ln).  Because the counter lives on the device a captured graph gets a fresh tag on every replay.
//   word  320          LN_ERR: sticky; a bounded wait of epilogue_resid_ln ran out (its rows were written as NaN)
//   bytes 32768 ..     ln_part: [rows][8] 16-byte entries {tag, mean, M2, ~tag} (gemm_epi8.h)
//
// Draws, the arrival count and the resets are agent-scope atomics: they execute at the memory side, so it does not matter which XCD's L2 a work-group
// sits behind (MI355X_MICROARCH.md, "dequeue": 0.3 us idle / 1.1-1.3 us under streaming load with eight per-XCD heads and 256 pullers).
// One workspace serves ONE stream: launches that overlap in time must not share it (kernels of one stream never overlap).
#pragma once
#include <hip/hip_runtime.h>
#define SW_HEAD(x) ((x) * 32)
#define SW_DONE 256
#define SW_EPOCH 288
#define SW_LN_ERR 320
#define SW_HEADER_BYTES 32768

__device__ __forceinline__ unsigned sw_draw(unsigned* head) { return __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned sw_epoch(const unsigned* sched) { return __hip_atomic_load(sched + SW_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ONE lane per work-group, after the work-group's last draw has returned; every work-group of the launch calls it exactly once (also those that found no work)
__device__ __forceinline__ void sw_leave(unsigned* sched, unsigned n_wg) {
  const unsigned before = __hip_atomic_fetch_add(sched + SW_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (before == n_wg - 1u) {                               // everybody else has left: nobody draws any more, nobody reads the epoch any more
#pragma unroll
    for (int x = 0; x < 8; ++x) __hip_atomic_store(sched + SW_HEAD(x), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sched + SW_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sched + SW_EPOCH, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
