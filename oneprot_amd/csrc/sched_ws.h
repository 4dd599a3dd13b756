// The "sched workspace": a small block of caller-provided device memory through which the persistent one-work-group-per-CU kernels hand out their work
// at run time and keep per-launch bookkeeping ON THE DEVICE (include/oneprot_hip.h: oneprot_sched_workspace_bytes / _init, oneprot_alloc_uncached).
//
//   words   0 .. 255   eight per-XCD queue heads, one 128-byte line each (HEAD(x)): the next undrawn tile / slab of XCD x's share of a launch
//   words 256-257      DONE (64 bits: low word = work-groups that have left the current launch, high word = tiles / slabs they computed).  The LAST one to leave resets the heads and DONE to zero and adds one to
//   word  288          EPOCH: launches completed on this workspace.  A kernel reads it once when it starts: epoch + 1 is the launch's tag (the tagged
//                      partial statistics of epilogue_resid_ln).  Because the counter lives on the device a captured graph gets a fresh tag on every replay.
//   word  320          LN_ERR: sticky; 1 = a bounded wait of epilogue_resid_ln ran out (its rows were written as NaN), 2 = a launch did not compute exactly the
//                      tiles it was given (sw_leave): either way oneprot_clip_coef turns the step's gradient norm into NaN and the host raises
//   word  352          LATE_DRAWS: diagnostic; ticket draws of k_gemm8 that had not returned behind the counted wait that should cover them (gemm_nt8.hip: draw_result)
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
#define SW_LATE_DRAWS 352
#define SW_HEADER_BYTES 32768

__device__ __forceinline__ unsigned sw_draw(unsigned* head) { return __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned sw_epoch(const unsigned* sched) { return __hip_atomic_load(sched + SW_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ONE lane per work-group, after the work-group's last draw has returned; every work-group of the launch calls it exactly once (also those that found no work).
// `units` = tiles / slabs this work-group computed, `expected` = what the whole launch must compute: arrival count and unit count travel in ONE 64-bit atomic
// (low / high word), so the last work-group to leave sees the launch's total -- a ticket handed out twice, lost, or a queue that did not start at zero (a
// previous launch on this workspace that never finished) sets the sticky error word (value 2) instead of leaving output tiles silently unwritten.
__device__ __forceinline__ void sw_leave(unsigned* sched, unsigned n_wg, unsigned units, unsigned expected) {
  unsigned long long* done = reinterpret_cast<unsigned long long*>(sched + SW_DONE);
  const unsigned long long before = __hip_atomic_fetch_add(done, 1ull | ((unsigned long long)units << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((unsigned)before == n_wg - 1u) {                     // everybody else has left: nobody draws any more, nobody reads the epoch any more
    if ((unsigned)(before >> 32) + units != expected) __hip_atomic_store(sched + SW_LN_ERR, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int x = 0; x < 8; ++x) __hip_atomic_store(sched + SW_HEAD(x), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sched + SW_EPOCH, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- asynchronous draws (kernels whose vector-memory waits are counted by hand: a compiler-inserted vmcnt(0) would drain their LDS-DMA prefetch)
// A draw from a queue head whose result is NOT waited for by the compiler (inline asm: hipcc's own wait would be vmcnt(0), i.e. a drain of the LDS-DMA
// prefetch it cannot see); the caller looks at the register behind a counted wait that covers it.  `lanes` = the EXEC mask of the atomic: 1 in the one
// wave that draws, 0 elsewhere (the instruction then does nothing and the register keeps 0x7fffffff) -- no branch, so no merge of two definitions of the
// register that is written behind the compiler's back.
__device__ __forceinline__ unsigned draw_async(const unsigned* head, int lanes) {
  unsigned v, zero, one;
  unsigned long long keep;
  const int lo = __builtin_amdgcn_readfirstlane(lanes);
  // Scalar base + a zero offset register: no address pair in vector registers (kept over the K loop it was spilled, and its reload -- behind a vmcnt(0) --
  // drained the operand prefetch in every iteration).  The drawing lane's destination starts as 0xffffffff, which no ticket is: draw_result() below tells
  // "not returned yet" from a ticket and never hands out a value the atomic has not delivered.
  asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 1\n\ts_mov_b64 %3, exec\n\ts_mov_b32 exec_lo, %5\n\ts_mov_b32 exec_hi, 0\n\ts_cbranch_execz 1f\n\t"
               "v_mov_b32 %0, -1\n\tglobal_atomic_add %0, %1, %2, %4 sc0\n1:\n\ts_mov_b64 exec, %3"
               : "=&v"(v), "=&v"(zero), "=&v"(one), "=&s"(keep) : "s"(head), "s"(lo) : "memory");
  return v;
}
// The drawn ticket, wave-uniform.  The caller places this behind a counted wait that covers the draw if vector-memory operations retire in order (they
// do in every run observed); should the value not have arrived all the same -- lane 0 still holds 0xffffffff -- the statement waits for everything in flight
// and reads again, inside ONE asm statement, so that hipcc cannot hand the register to another value before the atomic has written it.  (Round 6: a rare
// memory fault under a co-resident kernel pointed at a destination register written AFTER its reuse; with this form a late return costs a drained
// prefetch, not a corrupted register.)
__device__ __forceinline__ int draw_result(unsigned v, int& late) {
  int s;
  asm volatile("s_mov_b32 %1, 0\n\tv_readfirstlane_b32 %0, %2 ; DRAWN\n\ts_cmp_eq_u32 %0, -1\n\ts_cbranch_scc0 2f\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %0, %2\n\ts_mov_b32 %1, 1\n2:"
               : "=&s"(s), "=&s"(late) : "v"(v) : "memory", "scc");
  return s;
}
// the ticket relay word in LDS: plain ds instructions from inline asm (through a volatile C++ access hipcc emitted FLAT instructions with a vmcnt(0) behind
// them -- a drain of the operand prefetch, or of the epilogue's store burst); the reader's wait is the K-tile's own lgkmcnt(0)
__device__ __forceinline__ unsigned lds_read32(unsigned addr) { unsigned v = addr; asm volatile("ds_read_b32 %0, %0" : "+v"(v) : : "memory"); return v; }      // (address and result in ONE register)
__device__ __forceinline__ void lds_write32(unsigned addr, unsigned val) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(val) : "memory"); }
__device__ __forceinline__ int first_lane(unsigned v) { int s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(v)); return s; }
