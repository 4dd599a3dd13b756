// Small fp32 GEMM for the projection head and the contrastive logits (ref base_encoder.py:155-164, loss.py:91-99).
// These contractions are <0.01 % of the step's FLOPs but feed a softmax with logits scaled by 1/0.07, so they stay
// in exact fp32 (fmaf chains), not bf16 MFMA.  64x64 tile, 256 threads, 4x4 outputs per thread, BK=16 through LDS.
#include "common.h"
#include "../../include/oneprot_hip.h"

// A element (m,k): transA ? A[k*M + m] : A[m*K + k];   B element (k,n): b_is_kn ? B[k*N + n] : B[n*K + k]
__global__ void __launch_bounds__(256) k_sgemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                               int transA, int b_is_kn, float alpha, int accumulate) {
  __shared__ float sA[16][64 + 4];
  __shared__ float sB[16][64 + 4];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256;          // 0..1023
      int kk, mm;
      if (transA) { kk = idx >> 6; mm = idx & 63; } else { mm = idx >> 4; kk = idx & 15; }
      const int gm = m0 + mm, gk = k0 + kk;
      float v = 0.f;
      if (gm < M && gk < K) v = transA ? A[(size_t)gk * M + gm] : A[(size_t)gm * K + gk];
      sA[kk][mm] = v;
      int kb, nn;
      if (b_is_kn) { kb = idx >> 6; nn = idx & 63; } else { nn = idx >> 4; kb = idx & 15; }
      const int gn = n0 + nn, gkb = k0 + kb;
      float w = 0.f;
      if (gn < N && gkb < K) w = b_is_kn ? B[(size_t)gkb * N + gn] : B[(size_t)gn * K + gkb];
      sB[kb][nn] = w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
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
