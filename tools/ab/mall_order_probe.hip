// Development probe (GPU): does a streaming consumer gain from walking its input in the REVERSE of the order its producer wrote it?  The Infinity
// Cache (256 MiB, memory side) keeps the most recently touched lines: a consumer that starts where the producer began finds nothing of a > 256 MiB
// tensor, one that starts where the producer ended finds the tail still on die.
//   producer: writes `bytes` of fp32 (float4 per lane, blocks ascending)          consumer: reads them (ascending or descending blocks) and writes half
//   as many bytes of output (the LayerNorm-forward pattern: fp32 in, bf16 out)
// build: hipcc --offload-arch=gfx950 -O3 tools/ab/mall_order_probe.hip -o tools/ab/mall_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(256) k_produce(float4* __restrict__ dst, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = make_float4(v, v + 1, v + 2, v + 3);
}
// block b handles the contiguous chunk b (ascending) or nblocks-1-b (descending); 4 float4 per thread in flight
__global__ void __launch_bounds__(256) k_consume(const float4* __restrict__ src, float2* __restrict__ dst, size_t n4, int reverse) {
  const size_t per = 256 * 8;                                // float4 per block
  const size_t nb = n4 / per;
  for (size_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const size_t c = reverse ? nb - 1 - b : b;
    const float4* s = src + c * per + threadIdx.x;
    float2* d = dst + c * per + threadIdx.x;
    float4 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = s[k * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k * 256] = make_float2(r[k].x + r[k].y, r[k].z + r[k].w);
  }
}
int main() {
  const size_t sizes_mb[] = {168, 335, 503, 671};
  float4* buf; float2* out;
  hipMalloc(&buf, (size_t)1024 << 20); hipMalloc(&out, (size_t)512 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t mb : sizes_mb) {
    const size_t n4 = (mb << 20) / 16 / 2048 * 2048;
    for (int grid : {2048, 65536}) {                          // persistent-ish (8 blocks per CU) and one block per chunk (dispatch order = chunk order)
      for (int reverse = 0; reverse < 2; ++reverse) {
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
          hipLaunchKernelGGL(k_produce, dim3(2048), dim3(256), 0, 0, buf, n4, (float)rep);
          hipEventRecord(e0, 0);
          const int g = grid == 65536 ? (int)(n4 / 2048) : grid;
          hipLaunchKernelGGL(k_consume, dim3(g), dim3(256), 0, 0, buf, out, n4, reverse);
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double bytes = (double)n4 * 16 * 1.5;
        printf("%4zu MB produced, consumer grid %-14s %s: %7.1f us  (%.2f TB/s of its %.0f MB)\n", mb, grid == 65536 ? "one per chunk" : "2048 persistent", reverse ? "DESCENDING" : "ascending ",
               t[3] * 1e3, bytes / t[3] / 1e9, bytes / 1e6);
      }
    }
  }
  return 0;
}
