// Development probe (GPU): issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD, (a) with one accumulator tile and constant operands,
// (b) with the register pattern of the 8-phase GEMM's K-tile: 40 accumulator tiles (160 registers), 10 + 4 operand fragments, 80 MFMAs per
// iteration in quadrant order.  No LDS, no memory traffic, no barriers.  Prints shader cycles (s_memtime) per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ab/mfma_rate_probe.hip -o tools/ab/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_probe(const bf8_t* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ char smem[];      // forces one work-group per CU
  const int lane = threadIdx.x;
  bf8_t fa[4], fb[10];
  for (int i = 0; i < 4; ++i) fa[i] = in[lane + 64 * i];
  for (int j = 0; j < 10; ++j) fb[j] = in[lane + 64 * (4 + j)];
  f32x4 acc[4][10];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int n = 0; n < 80; ++n) acc[0][n & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0], acc[0][n & 1], 0, 0, 0);
    } else if (MODE == 3) {      // 40 accumulator tiles, constant operands
#pragma unroll
      for (int n = 0; n < 80; ++n) acc[(n % 40) / 10][n % 10] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0], acc[(n % 40) / 10][n % 10], 0, 0, 0);
    } else if (MODE == 4) {      // 2 accumulator tiles, 14 operand fragments in the GEMM's order
#pragma unroll
      for (int n = 0; n < 80; ++n) acc[0][n & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[n % 10], fa[(n / 10) % 4], acc[0][n & 1], 0, 0, 0);
    } else if (MODE == 6) {      // 40 accumulator tiles in VGPRs, in place (inline asm: vdst == srcC), GEMM quadrant order, 14 fragments
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int xh = ph >> 1, yh = (ph == 1 || ph == 2) ? 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[yh * 2 + mt][xh * 5 + nt]) : "v"(fb[xh * 5 + nt]), "v"(fa[yh * 2 + mt]));
      }
    } else if (MODE == 7) {      // the same with the accumulators in AGPRs
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int xh = ph >> 1, yh = (ph == 1 || ph == 2) ? 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[yh * 2 + mt][xh * 5 + nt]) : "v"(fb[xh * 5 + nt]), "v"(fa[yh * 2 + mt]));
      }
    } else if (MODE == 5) {      // 8 accumulator tiles round robin, constant operands
#pragma unroll
      for (int n = 0; n < 80; ++n) acc[0][n & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0], acc[0][n & 7], 0, 0, 0);
    } else {
      // quadrants (xh, yh) = (0,0) (0,1) (1,1) (1,0); X = W halves (5 tiles), Y = A halves (2 tiles); MODE 2: fragments made opaque per phase
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int xh = ph >> 1, yh = (ph == 1 || ph == 2) ? 1 : 0;
        if (MODE == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
          for (int j = 0; j < 10; ++j) asm volatile("" : "+v"(fb[j]));
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
              acc[yh * 2 + mt][xh * 5 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[xh * 5 + nt], fa[yh * 2 + mt], acc[yh * 2 + mt][xh * 5 + nt], 0, 0, 0);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE, int WAVES>
static void run(const char* name, const bf8_t* in, float* out, unsigned long long* cyc, int iters) {
  hipFuncSetAttribute((const void*)k_probe<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 100 * 1024, 0, in, out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 100 * 1024, 0, in, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double mf = 80.0 * iters;
  printf("%-58s waves/SIMD %d  cycles/MFMA (wave 0) %6.2f   wall %.3f ms  -> %.0f TFLOP/s chip-wide\n", name, WAVES / 4, (double)h[0] / mf, ms,
         256.0 * WAVES * mf * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
  bf8_t* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, 64 * 14 * sizeof(bf8_t)); hipMemset(in, 0, 64 * 14 * sizeof(bf8_t));
  hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8 * sizeof(unsigned long long));
  const int iters = 20000;
  run<0, 4>("constant operands, 2 accumulator tiles", in, out, cyc, iters);
  run<1, 4>("GEMM K-tile pattern: 40 acc tiles, 14 fragments", in, out, cyc, iters);
  run<2, 4>("same, fragments opaque per phase", in, out, cyc, iters);
  run<3, 4>("40 accumulator tiles, constant operands", in, out, cyc, iters);
  run<4, 4>("2 accumulator tiles, 14 fragments in GEMM order", in, out, cyc, iters);
  run<5, 4>("8 accumulator tiles round robin, constant operands", in, out, cyc, iters);
  run<6, 4>("GEMM pattern, 40 acc tiles IN PLACE in VGPRs (asm)", in, out, cyc, iters);
  run<7, 4>("GEMM pattern, 40 acc tiles IN PLACE in AGPRs (asm)", in, out, cyc, iters);
  run<0, 8>("constant operands, 2 accumulator tiles", in, out, cyc, iters);
  run<1, 8>("GEMM K-tile pattern: 40 acc tiles, 14 fragments", in, out, cyc, iters);
  run<2, 8>("same, fragments opaque per phase", in, out, cyc, iters);
  return 0;
}
