// Development probe: a VALU-only kernel (independent v_fma_f32 chains, <= 32 VGPRs, no memory traffic) to run NEXT TO a GEMM on another stream:
// does VALU work of one wave issue under the MFMA stream of co-resident waves?  (tools/ab/coissue.py)
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) k_valu_spin(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 1.0000001f, c = 1e-7f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
      a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
    }
  }
  const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 12345.678f) out[0] = s;
}
extern "C" int valu_spin(float* out, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(k_valu_spin, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
