// Development probe (GPU): the attention-forward tile's REAL dataflow (score MFMA chain -> 16 exponentials on its result -> 8 packs -> row-sum
// and P.V MFMA chains) in registers only -- no LDS, no memory -- to find the instruction ORDER that lets the matrix pipe and the vector pipe
// overlap, before the kernel is restructured around it.  One work-group of W waves per CU, every wave runs `iters` tiles.
//   V0  one tile after the other, as k_attn_fwd3 is written:   QK(t) | exp(t) pack(t) | sum/PV(t)
//   V1  software pipelined: the score chain of tile t+1 is issued before the exponentials of tile t
//   V2  V1 with s_setprio 1 around every MFMA group
//   V3  two independent query blocks per wave, interleaved (QK_a QK_b | exp_a, sumPV_a interleaved with exp_b ...): run with half the waves
//   V4  V0 with the row sum on the vector pipe (16 v_add) instead of two MFMAs
//   V5  V1 with the row sum on the vector pipe
// Prints wall-clock SIMD cycles per 32 x 32 tile (= the figure to compare with ~540 measured in k_attn_fwd3, profiles/r04_fwd3_stamps.txt).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ab/attn_mix_probe.hip -o tools/ab/attn_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

#define MFMA(c, a, b) (c) = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x16 zero16() { f32x16 z; for (int i = 0; i < 16; ++i) z[i] = 0.f; return z; }
__device__ __forceinline__ unsigned cvtpk(float a, float b) { unsigned d; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ void exp16(f32x16& s) {
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
}
__device__ __forceinline__ bf8_t pack(const f32x16& s, int sb) {
  u32x4 w = {cvtpk(s[8 * sb], s[8 * sb + 1]), cvtpk(s[8 * sb + 2], s[8 * sb + 3]), cvtpk(s[8 * sb + 4], s[8 * sb + 5]), cvtpk(s[8 * sb + 6], s[8 * sb + 7])};
  return __builtin_bit_cast(bf8_t, w);
}
#define PRIO(n) __builtin_amdgcn_s_setprio(n)

template <int V, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf8_t kf[2], qa[2], qb[2], vf[2], ones;
  {
    unsigned w[4];
    auto mk = [&](int o) -> bf8_t {
      for (int j = 0; j < 4; ++j) {
        const float a = (in[(lane * 8 + 2 * j + o) & 1023] - 0.75f) * 0.5f, b = (in[(lane * 8 + 2 * j + 1 + o) & 1023] - 0.75f) * 0.5f;
        w[j] = (__builtin_bit_cast(unsigned, a) >> 16) | (__builtin_bit_cast(unsigned, b) & 0xffff0000u);
      }
      u32x4 u = {w[0], w[1], w[2], w[3]};
      return __builtin_bit_cast(bf8_t, u);
    };
    kf[0] = mk(0); kf[1] = mk(64); qa[0] = mk(128); qa[1] = mk(192); qb[0] = mk(256); qb[1] = mk(320); vf[0] = mk(384); vf[1] = mk(448);
    u32x4 o = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    ones = __builtin_bit_cast(bf8_t, o);
  }
  f32x16 accA = zero16(), laccA = zero16(), accB = zero16(), laccB = zero16();
  float lA = 0.f, lB = 0.f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  auto qk = [&](const bf8_t (&q)[2]) -> f32x16 {
    f32x16 s = zero16();
    MFMA(s, kf[0], q[0]); MFMA(s, kf[1], q[1]);
    return s;
  };
  auto sumpv = [&](f32x16& s, f32x16& acc, f32x16& lacc, float& l, bool vsum) {
    if (vsum) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += s[i];
      l += t;
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const bf8_t p = pack(s, sb);
      if (!vsum) MFMA(lacc, ones, p);
      MFMA(acc, vf[sb], p);
    }
  };
  if (V == 0 || V == 4) {
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; ++it) {
      asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(vf[0]), "+v"(vf[1]));      // fragments "re-read" every tile
      f32x16 s = qk(qa);
      exp16(s);
      sumpv(s, accA, laccA, lA, V == 4);
    }
  } else if (V == 1 || V == 2 || V == 5) {
    f32x16 s = qk(qa);
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; ++it) {
      asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(vf[0]), "+v"(vf[1]));
      if (V == 2) PRIO(1);
      f32x16 sn = qk(qa);
      if (V == 2) PRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      exp16(s);
      if (V == 2) PRIO(1);
      sumpv(s, accA, laccA, lA, V == 5);
      if (V == 2) PRIO(0);
      s = sn;
    }
    accA[0] += s[0];
  } else if (V == 3) {
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; ++it) {
      asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(vf[0]), "+v"(vf[1]));
      f32x16 sa = qk(qa);
      f32x16 sb_ = qk(qb);
      exp16(sa);
      sumpv(sa, accA, laccA, lA, false);
      exp16(sb_);
      sumpv(sb_, accB, laccB, lB, false);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
  float sum = lA + lB;
  for (int i = 0; i < 16; ++i) sum += accA[i] + laccA[i] + accB[i] + laccB[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (blockIdx.x == 0 && lane == 0) { cyc[wave] = t1 - t0; cyc[16 + wave] = r1 - r0; }
}

template <int V, int WAVES>
static void run(const char* name, int tiles_per_iter, const float* in, float* out, unsigned long long* cyc, int iters) {
  hipFuncSetAttribute((const void*)k_probe<V, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<V, WAVES>), dim3(256), dim3(WAVES * 64), 100 * 1024, 0, in, out, cyc, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<V, WAVES>), dim3(256), dim3(WAVES * 64), 100 * 1024, 0, in, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[32]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[16] * 10e-9) / 1e9;
  const double tiles_per_simd = (double)(WAVES / 4) * tiles_per_iter * iters;
  printf("%-78s waves/SIMD %d | wave 0 %7.1f, last wave %7.1f cyc/tile | wall %.3f ms = %6.1f SIMD-cyc/tile at %.2f GHz\n", name, WAVES / 4,
         (double)h[0] / iters / tiles_per_iter, (double)h[WAVES - 1] / iters / tiles_per_iter, ms, ms * 1e-3 * ghz * 1e9 / tiles_per_simd, ghz);
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 1024 * sizeof(float)); hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 32 * sizeof(unsigned long long));
  float h[1024];
  unsigned s = 12345u;
  for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f + 0.25f; }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int it = 20000;
#define RUNW(V, name, tp)                            \
  run<V, 4>(name, tp, in, out, cyc, it);             \
  run<V, 8>(name, tp, in, out, cyc, it);             \
  run<V, 16>(name, tp, in, out, cyc, it);
  RUNW(0, "V0 tile after tile (QK | exp pack | sum+PV)", 1)
  RUNW(1, "V1 score chain of tile t+1 before the exponentials of tile t", 1)
  RUNW(2, "V2 = V1 + s_setprio 1 around the MFMA groups", 1)
  run<3, 4>("V3 two query blocks per wave, interleaved", 2, in, out, cyc, it);
  run<3, 8>("V3 two query blocks per wave, interleaved", 2, in, out, cyc, it);
  RUNW(4, "V4 = V0 with the row sum on the vector pipe (16 v_add, no ones-MFMA)", 1)
  RUNW(5, "V5 = V1 with the row sum on the vector pipe", 1)
  return 0;
}
