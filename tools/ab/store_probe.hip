// Development probe (GPU): how fast can ONE work-group (8 waves) push an output tile's stores into the memory system, by access shape?
// Each wave issues NST global_store_dwordx4 (1 KiB per wave-instruction) and we time (s_memtime) the issue of all of them and their completion
// (s_waitcnt vmcnt(0)).  Shapes:  0 = 16 rows x 64 B per instruction (the pair-map epilogue of the GEMMs: row pitch = N * 2 bytes)
//                                 1 =  8 rows x 128 B      2 = 4 rows x 256 B      3 = 1 KiB contiguous
// build: hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe ; run: ./store_probe [work-groups] [row pitch bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int SHAPE>
__global__ void __launch_bounds__(512) k_store(unsigned char* out, size_t pitch, int nst, int reps, unsigned long long* t_issue, unsigned long long* t_done) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows per instruction and bytes per row
  constexpr int BPR = SHAPE == 0 ? 64 : SHAPE == 1 ? 128 : SHAPE == 2 ? 256 : 1024;
  constexpr int RPI = 1024 / BPR;
  const int r = lane / (BPR / 16), cb = (lane % (BPR / 16)) * 16;
  unsigned long long ti = 0, td = 0;
  for (int rep = 0; rep < reps; ++rep) {
    // tile of this work-group and repetition: 256 rows; wave w owns rows 32w .. 32w+31 (shape 0: 2 row groups of 16)
    unsigned char* base = out + ((size_t)(blockIdx.x * reps + rep) * 256 + wave * 32) * pitch;
    __syncthreads();
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    const u32x4 v = {(unsigned)lane, (unsigned)rep, 3u, 4u};
    for (int k = 0; k < nst; ++k) {
      // the wave's region: 32 rows x 1024 bytes; successive instructions walk along the rows first (next BPR bytes), then to the next RPI rows
      constexpr int SEGS = 1024 / BPR;
      const int rg = (k / SEGS) % (32 / RPI), seg = k % SEGS;
      unsigned char* p = base + (size_t)(rg * RPI + r) * pitch + seg * BPR + cb;
      *reinterpret_cast<u32x4*>(p) = v;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
    ti += t1 - t0; td += t2 - t0;
  }
  if (lane == 0) { t_issue[blockIdx.x * 8 + wave] = ti / reps; t_done[blockIdx.x * 8 + wave] = td / reps; }
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;
  const size_t pitch = argc > 2 ? atol(argv[2]) : 5120;
  const int reps = 16, nst = 32;
  unsigned char* out; unsigned long long *ti, *td;
  const size_t bytes = (size_t)wgs * reps * 256 * pitch + (1 << 20);
  hipMalloc(&out, bytes); hipMalloc(&ti, wgs * 8 * 8); hipMalloc(&td, wgs * 8 * 8);
  std::vector<unsigned long long> hi(wgs * 8), hd(wgs * 8);
  for (int shape = 0; shape < 4; ++shape) {
    for (int it = 0; it < 2; ++it) {
      switch (shape) {
        case 0: hipLaunchKernelGGL(k_store<0>, dim3(wgs), dim3(512), 0, 0, out, pitch, nst, reps, ti, td); break;
        case 1: hipLaunchKernelGGL(k_store<1>, dim3(wgs), dim3(512), 0, 0, out, pitch, nst, reps, ti, td); break;
        case 2: hipLaunchKernelGGL(k_store<2>, dim3(wgs), dim3(512), 0, 0, out, pitch, nst, reps, ti, td); break;
        default: hipLaunchKernelGGL(k_store<3>, dim3(wgs), dim3(512), 0, 0, out, pitch, nst, reps, ti, td); break;
      }
      hipDeviceSynchronize();
    }
    hipMemcpy(hi.data(), ti, wgs * 8 * 8, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), td, wgs * 8 * 8, hipMemcpyDeviceToHost);
    std::sort(hi.begin(), hi.end()); std::sort(hd.begin(), hd.end());
    printf("shape %d (%4d B per row per instruction), %3d work-groups, pitch %zu: %d stores/wave issued in %6llu cycles (median wave), done in %6llu -> %.1f B/clk per CU\n",
           shape, shape == 0 ? 64 : shape == 1 ? 128 : shape == 2 ? 256 : 1024, wgs, pitch, nst, hi[hi.size() / 2], hd[hd.size() / 2], 8.0 * nst * 1024 / hd[hd.size() / 2]);
  }
  return 0;
}
