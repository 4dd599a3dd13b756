// Test / development probe (GPU): a kernel that HOLDS k compute units for a fixed wall time -- a stand-in for the RCCL channels of an overlapped gradient
// all-reduce (one work-group per channel, resident for the length of the collective).  Every work-group declares 100 KB of LDS, so no two of them
// share a CU and none of the product's 160 KB persistent work-groups can co-reside with one: the CU is taken.  The spin ends on the wall clock
// (s_memrealtime, 100 MHz), a condition every wave reaches.
// build: __graft_entry__.build() -> tests/libcu_spin.so (used by tests/test_kernels_gpu.py and tools/ab/cu_occupier.py)
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(64) k_cu_spin(long long ticks, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = (int)threadIdx.x;
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  int acc = 0;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) { acc += lds[(threadIdx.x + acc) & 63]; __builtin_amdgcn_s_sleep(32); }
  if (acc == 0x7fffffff) sink[0] = acc;
}
// holds `cus` compute units for `microseconds` on `stream`; returns 0 / -1
extern "C" int cu_spin_launch(int cus, long long microseconds, int* sink, void* stream) {
  if (cus <= 0) return 0;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k_cu_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess) return -1;
    configured = true;
  }
  hipLaunchKernelGGL(k_cu_spin, dim3(cus), dim3(64), 100 * 1024, (hipStream_t)stream, microseconds * 100, sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
