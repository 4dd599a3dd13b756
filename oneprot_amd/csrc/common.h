// Shared device helpers for the gfx950 kernels (wave64, bf16 storage / fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;      // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#define OP_OK 0
#define OP_EINVAL (-1)     // bad shape / null pointer / unsupported configuration
#define OP_ELAUNCH (-2)    // hipGetLastError() after the launch was not hipSuccess

// NOTE: element extraction from __bf16 ext-vectors miscompiles on ROCm 7.2 (observed: every lane element read
// element 0), so all packing/unpacking goes through integer bit operations.
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }   // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
// two floats -> packed bf16 pair (lo in bits 0..15): whole-vector convert so hipcc emits one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2_t));
}
__device__ __forceinline__ float bflo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// erf-GELU and its derivative from ONE exponential: erf(y), y = x/sqrt2, by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 on erf,
// far below the bf16 rounding of the stored results); exp(-y^2) = exp(-x^2/2) is also the Gaussian pdf factor of the derivative.
__device__ __forceinline__ void gelu_fwd_and_grad(float x, float& g, float& dg) {
  // written for instruction count (the GEMM epilogues that inline this are issue-bound): constants folded so that |x| and x^2 feed the
  // rational argument and the exponential directly; h = 0.5 * erfc(|x| / sqrt2) comes out of the polynomial with the 0.5 folded in
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));          // 0.3275911 / sqrt2
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752f);                        // exp(-x^2 / 2) = 2^(-x^2 * log2(e) / 2)
  float poly = fmaf(0.5307027145f, t, -0.7265760135f);
  poly = fmaf(poly, t, 0.7107068705f);
  poly = fmaf(poly, t, -0.142248368f);
  poly = fmaf(poly, t, 0.127414796f);
  const float h = poly * t * e;                                                        // 0.5 * erfc(|x|/sqrt2)
  const float cdf = 0.5f + copysignf(0.5f - h, x);                                     // Phi(x)
  g = x * cdf;
  dg = fmaf(x * e, 0.3989422804014327f, cdf);
}

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> four 32-bit words.  The dropout masks of this library are pure functions of
// (seed, stream id, element index) through this generator and are never stored (featops.hip: oneprot_dropout_*; attention.hip: the BERT tower's
// attention-probability dropout).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline int launch_status() { return hipGetLastError() == hipSuccess ? OP_OK : OP_ELAUNCH; }
