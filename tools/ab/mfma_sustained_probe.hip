// Development probe (GPU): what a bf16 MFMA loop that does NOTHING ELSE delivers on this device once it runs long enough for the clock governor to
// act -- on all-zero and on random operands.  Every CU runs one work-group of 4 or 8 waves (1 or 2 per SIMD); a wave keeps 40 accumulator tiles
// (160 registers, the GEMMs' pattern) and 14 operand fragments in registers and issues 80 v_mfma_f32_16x16x32_bf16 per iteration: no LDS, no
// memory traffic, no barriers.  Launches are repeated for ~2 s before the measured one.  Prints shader cycles per MFMA (s_memtime), the in-kernel
// clock (s_memtime / s_memrealtime x 100 MHz) and the chip-wide TFLOP/s by wall clock: the practical ceiling the GEMM K loops are compared with
// in NOTEBOOK.md section 6d (MI355X_MICROARCH.md, "DVFS give-back").  Reading the output: cycles and clock are those of wave 0 of every work-group; with
// two waves per SIMD the older wave takes the pipe first (16.8 cycles per MFMA = the pipe's rate) and the younger one follows, so the chip-wide
// figure is the one by wall clock.  The one-wave-per-SIMD rows (256 threads) are a code-generation artefact: hipcc then places the accumulators
// in AGPRs and copies them around every MFMA (27.9 cycles).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ab/mfma_sustained_probe.hip -o tools/ab/mfma_sustained_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_probe(const bf8_t* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
  extern __shared__ char smem[];      // forces one work-group per CU
  const int lane = threadIdx.x & 63;
  bf8_t fa[4], fb[10];
  for (int i = 0; i < 4; ++i) fa[i] = in[lane + 64 * i];
  for (int j = 0; j < 10; ++j) fb[j] = in[lane + 64 * (4 + j)];
  f32x4 acc[4][10];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    if ((it & 1023) == 1023) {        // keep the sums finite on random data: the accumulators are rescaled now and then (160 multiplies per 80 k MFMAs)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] *= 1.0e-3f;
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) sum += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int WAVES>
static void run(const char* name, const bf8_t* in, float* out, unsigned long long* clk, int n_cu) {
  hipFuncSetAttribute((const void*)k_probe<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 200000 / (WAVES / 4);           // ~50-100 ms per launch
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f, warm = 0.f;
  while (warm < 2000.f) {                           // let the governor settle
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<WAVES>), dim3(n_cu), dim3(WAVES * 64), 100 * 1024, 0, in, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    warm += ms;
  }
  std::vector<unsigned long long> h(2 * n_cu);
  hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> ghz(n_cu), cyc(n_cu);
  for (int i = 0; i < n_cu; ++i) { ghz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10e-9) / 1e9; cyc[i] = (double)h[2 * i] / ((double)iters * 80.0); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double flops = (double)n_cu * WAVES * iters * 80.0 * 16 * 16 * 32 * 2;
  printf("%-34s %d wave(s)/SIMD: %.1f cycles per MFMA per wave (median CU), in-kernel clock %.2f GHz (median; %.2f-%.2f), %.0f TFLOP/s by wall (%.1f ms)\n", name, WAVES / 4,
         cyc[n_cu / 2], ghz[n_cu / 2], ghz[0], ghz[n_cu - 1], flops / (ms * 1e-3) / 1e12, ms);
}

int main() {
  int dev = 0; hipDeviceProp_t prop; hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
  const int n_cu = prop.multiProcessorCount;
  bf8_t* in; float* out; unsigned long long* clk;
  hipMalloc(&in, 64 * 14 * sizeof(bf8_t)); hipMalloc(&out, (size_t)n_cu * 512 * sizeof(float)); hipMalloc(&clk, 2 * n_cu * sizeof(unsigned long long));
  std::vector<unsigned short> h(64 * 14 * 8);
  for (int pass = 0; pass < 2; ++pass) {
    unsigned s = 12345u;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      // bf16 of a value in (-2, 2) with a random mantissa: sign | exponent 125..127 | 7 mantissa bits
      v = pass == 0 ? 0 : (unsigned short)(((s >> 31) << 15) | ((125 + ((s >> 8) % 3)) << 7) | ((s >> 16) & 0x7f));
    }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<4>(pass == 0 ? "all-zero operands," : "random operands,", in, out, clk, n_cu);
    run<8>(pass == 0 ? "all-zero operands," : "random operands,", in, out, clk, n_cu);
  }
  return 0;
}
