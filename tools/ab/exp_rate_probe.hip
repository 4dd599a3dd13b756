// Development probe (GPU): what a v_exp_f32 costs on gfx950, alone and beside other work, with 1 / 2 / 4 waves per SIMD.
// No LDS, no memory traffic, no barriers inside the timed loop; random (non-zero) operands.  Prints shader cycles (s_memtime) per
// instruction group for wave 0 and the wall time of the launch (256 work-groups, one per CU).
//   mode 0  v_exp_f32 only                  (16 independent registers)
//   mode 1  v_fma_f32 only                  (16 independent registers)
//   mode 2  v_exp_f32 and v_fma_f32 alternating 1:1 in ONE wave                 -> do the issue costs add?
//   mode 3  v_mfma_f32_32x32x16_bf16 only   (2 accumulators)
//   mode 4  1 MFMA + E v_exp_f32 (E = 1..4) per group in ONE wave               -> exponentials hidden per MFMA gap
//   mode 5  waves 0..W/2-1 run mode 3 (MFMA only), waves W/2..W-1 run mode 0 (exp only): partner waves share a SIMD
//   mode 6  waves 0..W/2-1 run mode 1 (fma only), the others mode 0             -> is the transcendental unit a separate issue port?
//   mode 7  attention-forward tile mix in ONE wave: 6 MFMA + 16 v_exp_f32 + 8 v_cvt_pk_bf16_f32 + 4 plain VALU, exps spread 3/3/3/3/2/2
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ab/exp_rate_probe.hip -o tools/ab/exp_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define EXP(x) asm volatile("v_exp_f32 %0, -%0" : "+v"(x))       // x <- 2^-x: iterates to the fixed point 0.641, never inf / denormal
#define FMA(x, a, b) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))

template <int MODE, int E>
__device__ __forceinline__ void body(float (&x)[16], f32x16 (&acc)[2], const bf8_t& fa, const bf8_t& fb, float ca, float cb, int role) {
  if (MODE == 0 || ((MODE == 5 || MODE == 6) && role == 1)) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) EXP(x[i]);
  } else if (MODE == 1 || (MODE == 6 && role == 0)) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) FMA(x[i], ca, cb);
  } else if (MODE == 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) { EXP(x[i]); FMA(x[8 + i], ca, cb); }
  } else if (MODE == 3 || (MODE == 5 && role == 0)) {
#pragma unroll
    for (int n = 0; n < 16; ++n) MFMA(acc[n & 1], fa, fb);
  } else if (MODE == 4) {
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      MFMA(acc[n & 1], fa, fb);
#pragma unroll
      for (int e = 0; e < E; ++e) EXP(x[(n * E + e) & 15]);
    }
  } else if (MODE == 7) {
    // two tiles per call; per tile: MFMA, 3 exp | MFMA, 3 exp | MFMA, 3 exp, 2 cvt | MFMA, 3 exp, 2 cvt | MFMA, 2 exp, 2 cvt, 2 fma | MFMA, 2 exp, 2 cvt, 2 fma
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float d;
      MFMA(acc[0], fa, fb); EXP(x[0]); EXP(x[1]); EXP(x[2]);
      MFMA(acc[1], fa, fb); EXP(x[3]); EXP(x[4]); EXP(x[5]);
      MFMA(acc[0], fa, fb); EXP(x[6]); EXP(x[7]); EXP(x[8]); CVT(d, x[0], x[1]); CVT(d, x[2], x[3]);
      MFMA(acc[1], fa, fb); EXP(x[9]); EXP(x[10]); EXP(x[11]); CVT(d, x[4], x[5]); CVT(d, x[6], x[7]);
      MFMA(acc[0], fa, fb); EXP(x[12]); EXP(x[13]); CVT(d, x[8], x[9]); CVT(d, x[10], x[11]); FMA(x[14], ca, cb); FMA(x[15], ca, cb);
      MFMA(acc[1], fa, fb); EXP(x[14]); EXP(x[15]); CVT(d, x[12], x[13]); CVT(d, x[14], x[15]); FMA(x[0], ca, cb); FMA(x[1], ca, cb);
    }
  }
}

template <int MODE, int E, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters, int zero) {
  extern __shared__ char smem[];      // 100 KB dynamic: one work-group per CU
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // waves go to SIMDs round robin, so waves w and w + 4 share a SIMD: role by wave >= WAVES / 2 puts one wave of each role on every SIMD
  const int role = wave >= WAVES / 2 ? 1 : 0;
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = in[lane + 64 * i];
  const float ca = 0.999f, cb = in[lane] * 1e-6f;
  // MFMA operands: well-formed bf16 values in (-1, 1) (ZERO = 1: all zeros, the case that clocks highest)
  bf8_t fa, fb;
  { unsigned wa[4], wb[4];
    for (int j = 0; j < 4; ++j) {
      const float a0 = in[(lane * 8 + 2 * j) & 1023] - 0.75f, a1 = in[(lane * 8 + 2 * j + 1) & 1023] - 0.75f;
      const float b0 = in[(lane * 8 + 2 * j + 512) & 1023] - 0.75f, b1 = in[(lane * 8 + 2 * j + 513) & 1023] - 0.75f;
      wa[j] = zero ? 0u : ((__builtin_bit_cast(unsigned, a0) >> 16) | (__builtin_bit_cast(unsigned, a1) & 0xffff0000u));
      wb[j] = zero ? 0u : ((__builtin_bit_cast(unsigned, b0) >> 16) | (__builtin_bit_cast(unsigned, b1) & 0xffff0000u));
    }
    typedef __attribute__((__vector_size__(16))) unsigned u4;
    u4 ua = {wa[0], wa[1], wa[2], wa[3]}, ub = {wb[0], wb[1], wb[2], wb[3]};
    fa = __builtin_bit_cast(bf8_t, ua); fb = __builtin_bit_cast(bf8_t, ub); }
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    if (role == 0) body<MODE, E>(x, acc, fa, fb, ca, cb, 0);
    else body<MODE, E>(x, acc, fa, fb, ca, cb, 1);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i] + acc[0][i] + acc[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) { cyc[wave] = t1 - t0; cyc[16 + wave] = r1 - r0; }
}

template <int MODE, int E, int WAVES>
static void run(const char* name, double per_iter_a, const char* unit_a, double per_iter_b, const char* unit_b, const float* in, float* out,
                unsigned long long* cyc, int iters, int zero = 0) {
  hipFuncSetAttribute((const void*)k_probe<MODE, E, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<MODE, E, WAVES>), dim3(256), dim3(WAVES * 64), 100 * 1024, 0, in, out, cyc, 100, zero);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<MODE, E, WAVES>), dim3(256), dim3(WAVES * 64), 100 * 1024, 0, in, out, cyc, iters, zero);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[32]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double c0 = (double)h[0] / iters, c1 = (double)h[WAVES - 1] / iters;
  printf("%-64s waves/SIMD %d | wave 0: %7.1f cyc/iter = %6.2f cyc/%s", name, WAVES / 4, c0, c0 / per_iter_a, unit_a);
  if (per_iter_b > 0) printf(" | last wave: %7.1f cyc/iter = %6.2f cyc/%s", c1, c1 / per_iter_b, unit_b);
  // SIMD-level throughput: cycles of SIMD time per unit = wave cycles / waves per SIMD (when all waves run the same body)
  // shader clock inside the loop = s_memtime ticks per s_memrealtime tick (100 MHz); SIMD-level cost = wave cycles / waves per SIMD when all waves run the same body
  const double ghz = (double)h[0] / ((double)h[16] * 10e-9) / 1e9;
  // wall-based SIMD cost of one unit of the first role (waves of a SIMD do NOT share it evenly: the oldest wave issues first, so per-wave cycles understate it)
  const double n_a = (MODE == 5 || MODE == 6) ? WAVES / 8.0 : WAVES / 4.0;
  printf(" | wall %.3f ms = %6.2f SIMD-cyc/%s at the in-kernel clock %.2f GHz%s\n", ms, ms * 1e-3 * ghz * 1e9 / (n_a * per_iter_a * iters), unit_a, ghz, zero ? " [zero operands]" : "");
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 64 * 16 * sizeof(float)); hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 32 * sizeof(unsigned long long));
  float h[1024];
  unsigned s = 12345u;
  for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f + 0.25f; }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int it = 20000;
#define RUN3(MODE, E, name, pa, ua, pb, ub)                         \
  run<MODE, E, 4>(name, pa, ua, pb, ub, in, out, cyc, it);          \
  run<MODE, E, 8>(name, pa, ua, pb, ub, in, out, cyc, it);          \
  run<MODE, E, 16>(name, pa, ua, pb, ub, in, out, cyc, it);
  RUN3(0, 0, "v_exp_f32 only (64 per iter)", 64, "exp", 0, "")
  RUN3(1, 0, "v_fma_f32 only (64 per iter)", 64, "fma", 0, "")
  RUN3(2, 0, "v_exp_f32 + v_fma_f32 alternating in one wave (32 + 32)", 32, "pair", 0, "")
  RUN3(3, 0, "v_mfma_f32_32x32x16_bf16 only (16 per iter)", 16, "mfma", 0, "")
  run<3, 0, 4>("v_mfma_f32_32x32x16_bf16 only, ZERO operands", 16, "mfma", 0, "", in, out, cyc, it, 1);
  run<3, 0, 8>("v_mfma_f32_32x32x16_bf16 only, ZERO operands", 16, "mfma", 0, "", in, out, cyc, it, 1);
  RUN3(4, 1, "1 MFMA + 1 v_exp per group (16 groups)", 16, "group", 0, "")
  RUN3(4, 2, "1 MFMA + 2 v_exp per group", 16, "group", 0, "")
  RUN3(4, 3, "1 MFMA + 3 v_exp per group", 16, "group", 0, "")
  RUN3(4, 4, "1 MFMA + 4 v_exp per group", 16, "group", 0, "")
  run<5, 0, 8>("partners: waves 0-3 MFMA only (16), waves 4-7 exp only (64)", 16, "mfma", 64, "exp", in, out, cyc, it);
  run<5, 0, 16>("partners: waves 0-7 MFMA only (16), waves 8-15 exp only (64)", 16, "mfma", 64, "exp", in, out, cyc, it);
  run<6, 0, 8>("partners: waves 0-3 fma only (64), waves 4-7 exp only (64)", 64, "fma", 64, "exp", in, out, cyc, it);
  run<6, 0, 16>("partners: waves 0-7 fma only (64), waves 8-15 exp only (64)", 64, "fma", 64, "exp", in, out, cyc, it);
  RUN3(7, 0, "attention tile mix: 6 MFMA + 16 exp + 8 cvt_pk + 4 fma (2 tiles)", 2, "tile", 0, "")
  return 0;
}
