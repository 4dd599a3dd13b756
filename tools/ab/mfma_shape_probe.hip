// Development probe (GPU): the two bf16 MFMA shapes in a loop that does nothing else, SUSTAINED (2 s of launches before the measured one, so the board sits at
// its power cap and the governor has set the clock): v_mfma_f32_16x16x32_bf16 (the GEMMs' instruction; 40 accumulator tiles of 4 registers) against
// v_mfma_f32_32x32x16_bf16 (10 tiles of 16 registers: the same 64 x 160 wave tile, the same 160 accumulator registers, the same FLOPs per iteration, HALF the
// operand-register reads per FLOP).  Two waves per SIMD, one work-group per CU, random bf16 operands in registers: no LDS, no memory traffic.  The question
// ("energy per FLOP"): does the wider shape deliver more FLOP/s once power, not issue, is the limit?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ab/mfma_shape_probe.hip -o tools/ab/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int SHAPE>      // 16: 16x16x32, 32: 32x32x16
__global__ void __launch_bounds__(512, 1) k_probe(const bf8_t* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
  extern __shared__ char smem[];      // forces one work-group per CU
  const int lane = threadIdx.x & 63;
  unsigned long long t0, t1, r0, r1;
  float sum = 0.f;
  if constexpr (SHAPE == 16) {
    bf8_t fa[4], fb[10];
    for (int i = 0; i < 4; ++i) fa[i] = in[lane + 64 * i];
    for (int j = 0; j < 10; ++j) fb[j] = in[lane + 64 * (4 + j)];
    f32x4 acc[4][10];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
      if ((it & 1023) == 1023)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 10; ++j) acc[i][j] *= 1.0e-3f;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) sum += acc[i][j][0] + acc[i][j][3];
  } else {
    bf8_t fa[2][2], fb[5][2];          // [block][k-step of 16]
    for (int i = 0; i < 2; ++i) for (int s = 0; s < 2; ++s) fa[i][s] = in[lane + 64 * (2 * i + s)];
    for (int j = 0; j < 5; ++j) for (int s = 0; s < 2; ++s) fb[j][s] = in[lane + 64 * (4 + 2 * j + s)];
    f32x16 acc[2][5];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 5; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][s], fa[i][s], acc[i][j], 0, 0, 0);
      if ((it & 1023) == 1023)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[i][j] *= 1.0e-3f;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 5; ++j) sum += acc[i][j][0] + acc[i][j][7];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const char* name, const bf8_t* in, float* out, unsigned long long* clk, int n_cu) {
  hipFuncSetAttribute((const void*)k_probe<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 100000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f, warm = 0.f;
  while (warm < 3000.f) {                           // the board reaches its power cap, the governor settles
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<SHAPE>), dim3(n_cu), dim3(512), 100 * 1024, 0, in, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    warm += ms;
  }
  std::vector<unsigned long long> h(2 * n_cu);
  hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> ghz(n_cu);
  for (int i = 0; i < n_cu; ++i) ghz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10e-9) / 1e9;
  std::sort(ghz.begin(), ghz.end());
  const double flops = (double)n_cu * 8 * iters * 80.0 * 16 * 16 * 32 * 2;      // both shapes: 64 x 160 x 64 MACs per wave and iteration
  printf("%-44s in-kernel clock %.2f GHz (median CU; %.2f-%.2f), %.0f TFLOP/s by wall clock (%.1f ms per launch)\n", name, ghz[n_cu / 2], ghz[0], ghz[n_cu - 1],
         flops / (ms * 1e-3) / 1e12, ms);
}

int main() {
  int dev = 0; hipDeviceProp_t prop; hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
  const int n_cu = prop.multiProcessorCount;
  bf8_t* in; float* out; unsigned long long* clk;
  hipMalloc(&in, 64 * 14 * sizeof(bf8_t)); hipMalloc(&out, (size_t)n_cu * 512 * sizeof(float)); hipMalloc(&clk, 2 * n_cu * sizeof(unsigned long long));
  std::vector<unsigned short> h(64 * 14 * 8);
  unsigned s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (unsigned short)(((s >> 31) << 15) | ((125 + ((s >> 8) % 3)) << 7) | ((s >> 16) & 0x7f));      // random bf16 in (-2, 2)
  }
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {
    run<16>("v_mfma_f32_16x16x32_bf16, random operands:", in, out, clk, n_cu);
    run<32>("v_mfma_f32_32x32x16_bf16, random operands:", in, out, clk, n_cu);
  }
  return 0;
}
