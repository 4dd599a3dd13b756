// Development probe (GPU box only, not part of the product): how do MFMA work and VALU / transcendental work share one SIMD on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o coissue2 coissue2.hip && ./coissue2
// Every test launches 256 work-groups (one per CU) and reports in-kernel cycles (s_memtime) of the slowest wave class plus wall time.
//   A  mfma only, 4 waves (1 / SIMD)          B  mfma only, 8 waves (2 / SIMD)
//   C  valu only, 4 waves                     D  valu only, 8 waves
//   E  8 waves: waves 0-3 mfma, 4-7 valu (one of each per SIMD)      -> max(A, C) if the pipes overlap across waves, A + C if they do not
//   F  same wave interleaves 1 mfma + k valu (k = 0..8), 4 waves      -> free VALU slots per MFMA, in-wave
//   G  as E with "gelu-like" work (rcp + exp2 + fma chain) instead of plain fma
// MFMA shape selectable: 16x16x32 (MF=0) or 32x32x16 (MF=1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define NACC 8

template <int MF> struct Acc;
template <> struct Acc<0> { typedef f32x4 T; };
template <> struct Acc<1> { typedef f32x16 T; };

// inline asm (volatile) pins the issue ORDER of the probe streams and keeps hipcc from SLP-packing the fma into v_pk_fma_f32
template <int MF> __device__ __forceinline__ void mfma(bf8_t a, bf8_t b, typename Acc<MF>::T& c) {
  if constexpr (MF == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void vfma(float& v, float m, float c) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c)); }

__device__ __forceinline__ float gelu_like(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752f);
  float poly = fmaf(0.5307027145f, t, -0.7265760135f);
  poly = fmaf(poly, t, 0.7107068705f);
  poly = fmaf(poly, t, -0.142248368f);
  poly = fmaf(poly, t, 0.127414796f);
  const float h = poly * t * e;
  return fmaf(-fabsf(x), h, fmaxf(x, 0.f));
}

// mode: 0 all waves mfma; 1 all waves valu; 2 waves < 4 mfma, waves >= 4 valu; 3 in-wave interleave (k valu per mfma); 4 as 2 with gelu work;
//       5 all waves gelu work
template <int MF, int KV>
__global__ void __launch_bounds__(512, 2) k_probe(int mode, int iters, float seed, float* sink, unsigned long long* cyc) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  typename Acc<MF>::T acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < (MF ? 16 : 4); ++r) acc[i][r] = seed * (i + r);
  bf8_t a, b;
  for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(seed + lane * 0.001f + r); b[r] = (__bf16)(seed * 0.5f - lane * 0.002f + r); }
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + i * 0.25f + lane * 0.01f;
  const float c1 = 1.0001f * seed, c2 = 0.5f * seed;
  const bool do_mfma = mode == 0 || mode == 3 || ((mode == 2 || mode == 4) && wave < 4);
  const bool do_valu = mode == 1 || ((mode == 2) && wave >= 4);
  const bool do_gelu = mode == 5 || (mode == 4 && wave >= 4);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        mfma<MF>(a, b, acc[i]);
#pragma unroll
        for (int k = 0; k < KV; ++k) vfma(v[(i * KV + k) & 15], c1, c2);
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) mfma<MF>(a, b, acc[i]);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) vfma(v[i], c1, c2);        // 32 fma per iteration, 16 independent chains
    }
  } else if (do_gelu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = gelu_like(v[i]) + 0.75f;             // 16 gelu per iteration (~15 ops each)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < (MF ? 16 : 4); ++r) s += acc[i][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MF, int KV>
static void run(const char* name, int mode, int waves, int iters, unsigned long long* dcyc, float* dsink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(dcyc, 0, 256 * 8 * sizeof(unsigned long long));
  k_probe<MF, KV><<<256, waves * 64>>>(mode, iters, 1.0f, dsink, dcyc);      // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_probe<MF, KV><<<256, waves * 64>>>(mode, iters, 1.0f, dsink, dcyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), dcyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  // median over work-groups of the per-class cycle counts (waves 0-3, waves 4-7)
  std::vector<unsigned long long> lo, hi;
  for (int b = 0; b < 256; ++b) {
    unsigned long long ml = 0, mh = 0;
    for (int w = 0; w < waves; ++w) { if (w < 4) ml = std::max(ml, h[b * 8 + w]); else mh = std::max(mh, h[b * 8 + w]); }
    lo.push_back(ml); hi.push_back(mh);
  }
  std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
  printf("%-44s waves %d  cycles(w0-3) %9llu  cycles(w4-7) %9llu  per-iter %8.1f / %8.1f   wall %.3f ms\n", name, waves, lo[128], hi[128],
         (double)lo[128] / iters, (double)hi[128] / iters, ms);
}

int main() {
  unsigned long long* dcyc; float* dsink;
  hipMalloc(&dcyc, 256 * 8 * sizeof(unsigned long long));
  hipMalloc(&dsink, 64);
  const int IT = 20000;
  printf("cycles = s_memtime ticks (shader clock); per-iter: mfma loops = %d MFMAs, valu loop = 32 fma, gelu loop = 16 gelu\n", NACC);
  printf("---- v_mfma_f32_16x16x32_bf16\n");
  run<0, 0>("A  mfma only", 0, 4, IT, dcyc, dsink);
  run<0, 0>("B  mfma only", 0, 8, IT, dcyc, dsink);
  run<0, 0>("C  valu only (32 fma/iter)", 1, 4, IT, dcyc, dsink);
  run<0, 0>("D  valu only", 1, 8, IT, dcyc, dsink);
  run<0, 0>("E  w0-3 mfma | w4-7 valu", 2, 8, IT, dcyc, dsink);
  run<0, 0>("G5 gelu only (16 gelu/iter)", 5, 4, IT, dcyc, dsink);
  run<0, 0>("G5 gelu only", 5, 8, IT, dcyc, dsink);
  run<0, 0>("G  w0-3 mfma | w4-7 gelu", 4, 8, IT, dcyc, dsink);
  run<0, 0>("F0 in-wave 1 mfma + 0 fma", 3, 4, IT, dcyc, dsink);
  run<0, 1>("F1 in-wave 1 mfma + 1 fma", 3, 4, IT, dcyc, dsink);
  run<0, 2>("F2 in-wave 1 mfma + 2 fma", 3, 4, IT, dcyc, dsink);
  run<0, 3>("F3 in-wave 1 mfma + 3 fma", 3, 4, IT, dcyc, dsink);
  run<0, 4>("F4 in-wave 1 mfma + 4 fma", 3, 4, IT, dcyc, dsink);
  run<0, 6>("F6 in-wave 1 mfma + 6 fma", 3, 4, IT, dcyc, dsink);
  run<0, 2>("F2 in-wave 1 mfma + 2 fma, 2 waves/SIMD", 3, 8, IT, dcyc, dsink);
  run<0, 4>("F4 in-wave 1 mfma + 4 fma, 2 waves/SIMD", 3, 8, IT, dcyc, dsink);
  printf("---- v_mfma_f32_32x32x16_bf16\n");
  run<1, 0>("A  mfma only", 0, 4, IT, dcyc, dsink);
  run<1, 0>("E  w0-3 mfma | w4-7 valu", 2, 8, IT, dcyc, dsink);
  run<1, 0>("G  w0-3 mfma | w4-7 gelu", 4, 8, IT, dcyc, dsink);
  run<1, 0>("F0 in-wave 1 mfma + 0 fma", 3, 4, IT, dcyc, dsink);
  run<1, 2>("F2 in-wave 1 mfma + 2 fma", 3, 4, IT, dcyc, dsink);
  run<1, 4>("F4 in-wave 1 mfma + 4 fma", 3, 4, IT, dcyc, dsink);
  run<1, 6>("F6 in-wave 1 mfma + 6 fma", 3, 4, IT, dcyc, dsink);
  run<1, 8>("F8 in-wave 1 mfma + 8 fma", 3, 4, IT, dcyc, dsink);
  run<1, 6>("F6 in-wave 1 mfma + 6 fma, 2 waves/SIMD", 3, 8, IT, dcyc, dsink);
  return 0;
}
