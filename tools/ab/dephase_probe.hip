// Development probe (GPU): does de-phasing persistent work-groups shorten their store bursts?  Every work-group (8 waves) repeats REPS times
// { busy-wait MAIN cycles (stand-in for a K loop) ; store burst of NST x 1 KiB per wave in row layout (16 rows x 64 B per instruction) } and
// work-groups with (blockIdx / 8) % G != 0 start ((blockIdx / 8) % G) * OFFSET cycles late.  Reports the slowest work-group's total and the
// median burst length.   usage: dephase_probe [G] [OFFSET cycles] [MAIN cycles] [NST]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__global__ void __launch_bounds__(512) k(unsigned char* out, size_t pitch, int nst, int reps, int G, int offset, int main_cycles, unsigned long long* total, unsigned long long* burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane >> 2, cb = (lane & 3) * 16;
  const unsigned long long t_start = now();
  const int late = ((blockIdx.x >> 3) % G) * offset;
  while (now() - t_start < (unsigned long long)late) __builtin_amdgcn_s_sleep(8);
  unsigned long long tb = 0;
  for (int rep = 0; rep < reps; ++rep) {
    const unsigned long long t0 = now();
    while (now() - t0 < (unsigned long long)main_cycles) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    const unsigned long long t1 = now();
    unsigned char* base = out + ((size_t)(blockIdx.x * reps + rep) * 256 + wave * 32) * pitch;
    const u32x4 v = {(unsigned)lane, (unsigned)rep, 3u, 4u};
    for (int k = 0; k < nst; ++k) {
      const int rg = (k / 16) % 2, seg = k % 16;
      *reinterpret_cast<u32x4*>(base + (size_t)(rg * 16 + r) * pitch + seg * 64 + cb) = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tb += now() - t1;
  }
  if (threadIdx.x == 0) { total[blockIdx.x] = now() - t_start - late; burst[blockIdx.x] = tb / reps; }
}
int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 1, offset = argc > 2 ? atoi(argv[2]) : 0, main_cycles = argc > 3 ? atoi(argv[3]) : 37000, nst = argc > 4 ? atoi(argv[4]) : 20;
  const int wgs = 256, reps = 16; const size_t pitch = 5120;
  unsigned char* out; unsigned long long *tt, *tb;
  hipMalloc(&out, (size_t)wgs * reps * 256 * pitch + (1 << 20)); hipMalloc(&tt, wgs * 8); hipMalloc(&tb, wgs * 8);
  std::vector<unsigned long long> ht(wgs), hb(wgs);
  for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, out, pitch, nst, reps, G, offset, main_cycles, tt, tb); hipDeviceSynchronize(); }
  hipMemcpy(ht.data(), tt, wgs * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), tb, wgs * 8, hipMemcpyDeviceToHost);
  std::sort(ht.begin(), ht.end()); std::sort(hb.begin(), hb.end());
  printf("G=%d offset=%d main=%d nst=%d: per work-group total (excl. its start delay) median %llu max %llu cycles = %.0f per repetition; burst median %llu max %llu\n",
         G, offset, main_cycles, nst, ht[wgs / 2], ht[wgs - 1], (double)ht[wgs / 2] / reps, hb[wgs / 2], hb[wgs - 1]);
  return 0;
}
