// Small fp32 GEMM for the projection head and the contrastive logits (ref base_encoder.py:155-164, loss.py:91-99).
// These contractions are <0.01 % of the step's FLOPs but feed a softmax with logits scaled by 1/0.07, so they stay
// in exact fp32 (fmaf chains), not bf16 MFMA.  64x64 tile, 256 threads, 4x4 outputs per thread, 64-deep K-slices through LDS.
#include "common.h"
#include "../../include/oneprot_hip.h"

// A element (m,k): transA ? A[k*M + m] : A[m*K + k];   B element (k,n): b_is_kn ? B[k*N + n] : B[n*K + k]
// These launches are latency-bound (M = 256 rows: 50-130 work-groups, K <= 1024): a K-slice of 64 per iteration with all of its 32 loads per
// thread in flight together, and the next slice requested before the current one is multiplied (16-deep slices, loaded one after the other,
// took ~4x longer on the head shapes).
#define SG_BK 64
__global__ void __launch_bounds__(256) k_sgemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                               int transA, int b_is_kn, float alpha, int accumulate) {
  __shared__ float sA[SG_BK][64 + 4];
  __shared__ float sB[SG_BK][64 + 4];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  constexpr int NIT = SG_BK * 64 / 256;
  float ra[NIT], rb[NIT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;          // 0 .. SG_BK*64-1
      int kk, mm;
      if (transA) { kk = idx >> 6; mm = idx & 63; } else { mm = idx / SG_BK; kk = idx % SG_BK; }
      const int gm = m0 + mm, gk = k0 + kk;
      ra[it] = (gm < M && gk < K) ? (transA ? A[(size_t)gk * M + gm] : A[(size_t)gm * K + gk]) : 0.f;
      int kb, nn;
      if (b_is_kn) { kb = idx >> 6; nn = idx & 63; } else { nn = idx / SG_BK; kb = idx % SG_BK; }
      const int gn = n0 + nn, gkb = k0 + kb;
      rb[it] = (gn < N && gkb < K) ? (b_is_kn ? B[(size_t)gkb * N + gn] : B[(size_t)gn * K + gkb]) : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      int kk, mm;
      if (transA) { kk = idx >> 6; mm = idx & 63; } else { mm = idx / SG_BK; kk = idx % SG_BK; }
      sA[kk][mm] = ra[it];
      int kb, nn;
      if (b_is_kn) { kb = idx >> 6; nn = idx & 63; } else { nn = idx / SG_BK; kb = idx % SG_BK; }
      sB[kb][nn] = rb[it];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += SG_BK) {
    commit();
    __syncthreads();
    if (k0 + SG_BK < K) fetch(k0 + SG_BK);          // in flight during the multiply
#pragma unroll 8
    for (int kk = 0; kk < SG_BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sA[kk][ty * 4 + i]; b[i] = sB[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn >= N) continue;
      const size_t o = (size_t)gm * N + gn;
      const float v = alpha * acc[i][j];
      C[o] = accumulate ? C[o] + v : v;
    }
  }
}

extern "C" int oneprot_sgemm(const float* A, const float* B, float* C, int M, int N, int K, int transA, int b_is_kn, float alpha, int accumulate, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_sgemm, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, transA, b_is_kn, alpha, accumulate);
  return launch_status();
}
