// Development probe (GPU box only): what bounds the L2 -> LDS fill rate of a CU for the operand pattern of the NT GEMMs?
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip && ./fill_probe
// One 512-thread work-group per CU walks tiles exactly like the GEMM kernels (A rows [M, K] of a 256-row panel, W rows of a 128- or 256-row
// block, K-slices of BK), fills an LDS ring by LDS-DMA and does nothing else (optionally the other waves read the ring back).  Reported:
// bytes per clock per CU, over variants of: rows per piece (BK 32 = 16 half lines, BK 64 = 8 full lines), waves issuing (4 or 8), stages in
// flight, cache policy (default / nt), A only / W only / both, panel sharing between CUs (same tiles_n neighbours as the GEMM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BK: K-slice (32 or 64 bf16), LW: loader waves (4 or 8), NB: ring buffers, FL: stages kept in flight, POL: 0 default, 2 nt,
// WHAT: 1 A only, 2 W only, 3 both;  BN: W rows per stage (128 or 256)
template <int BK, int LW, int NB, int FL, int POL, int WHAT, int BN>
__global__ void __launch_bounds__(512, 2) k_fill(const unsigned short* A, const unsigned short* W, int K, int tiles_n, int tiles_per_wg, unsigned long long* cyc) {
  constexpr int ROWB = BK * 2, RPI = 1024 / ROWB;
  constexpr int A_P = 256 / RPI, B_P = BN / RPI;                 // pieces per stage
  constexpr int A_IPW = (WHAT & 1) ? A_P / LW : 0, B_IPW = (WHAT & 2) ? B_P / LW : 0;
  constexpr int LPS = A_IPW + B_IPW;
  constexpr int STAGE = (256 + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NK = K / BK;
  const int x = blockIdx.x & 7, w = blockIdx.x >> 3, g8 = gridDim.x >> 3;
  constexpr int CH = BK / 8;
  const int srow = lane / CH, schunk = lane % CH;
  const unsigned rel = (unsigned)srow * (unsigned)K * 2u + (unsigned)(schunk ^ (srow & (CH - 1))) * 16u;
  const size_t piece = (size_t)RPI * K * 2;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (wave < LW) {
    t0 = __builtin_amdgcn_s_memtime();
    int buf = 0;
    for (int q = 0; q < tiles_per_wg; ++q) {
      const int u = q * g8 + w;
      const int pl = u / tiles_n, tn = u - pl * tiles_n;
      const unsigned char* at = (const unsigned char*)(A + (size_t)((pl * 8 + x) * 256) * K) + (size_t)(wave * A_IPW) * piece;
      const unsigned char* bt = (const unsigned char*)(W + (size_t)(tn * BN) * K) + (size_t)(wave * B_IPW) * piece;
      for (int t = 0; t < NK; ++t) {
        unsigned char* sA = smem + buf * STAGE;
        unsigned char* sB = sA + 256 * ROWB;
#pragma unroll
        for (int i = 0; i < A_IPW; ++i) __builtin_amdgcn_global_load_lds(GLB_PTR(at + (size_t)t * ROWB + i * piece + rel), LDS_PTR(sA + (wave * A_IPW + i) * 1024), 16, 0, POL);
#pragma unroll
        for (int i = 0; i < B_IPW; ++i) __builtin_amdgcn_global_load_lds(GLB_PTR(bt + (size_t)t * ROWB + i * piece + rel), LDS_PTR(sB + (wave * B_IPW + i) * 1024), 16, 0, POL);
        wait_vmcnt<FL * LPS>();
        buf = buf + 1 == NB ? 0 : buf + 1;
      }
    }
    wait_vmcnt<0>();
    t1 = __builtin_amdgcn_s_memtime();
  }
  if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BK, int LW, int NB, int FL, int POL, int WHAT, int BN>
static void run(const char* name, const unsigned short* A, const unsigned short* W, int K, int N, unsigned long long* dcyc) {
  constexpr int STAGE = (256 + BN) * BK * 2;
  const int tiles_n = N / BN, tiles_m = 131072 / 256;
  const int tiles_per_wg = tiles_m * tiles_n / 256;
  hipFuncSetAttribute((const void*)k_fill<BK, LW, NB, FL, POL, WHAT, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, NB * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_fill<BK, LW, NB, FL, POL, WHAT, BN><<<256, 512, NB * STAGE>>>(A, W, K, tiles_n, tiles_per_wg, dcyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_fill<BK, LW, NB, FL, POL, WHAT, BN><<<256, 512, NB * STAGE>>>(A, W, K, tiles_n, tiles_per_wg, dcyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dcyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double bytes_per_wg = (double)tiles_per_wg * (K / BK) * (((WHAT & 1) ? 256 : 0) + ((WHAT & 2) ? BN : 0)) * BK * 2;
  printf("%-64s %7.1f B/clk/CU (median WG)  wall %.3f ms = %6.1f GB/s/CU, %5.2f TB/s chip   [%d KB/stage, %d in flight]\n", name, bytes_per_wg / (double)h[128], ms,
         bytes_per_wg / (ms * 1e-3) / 1e9, bytes_per_wg * 256 / (ms * 1e-3) / 1e12, STAGE / 1024, FL);
}

int main() {
  const int K = 640, N = 2560;
  unsigned short *A, *W; unsigned long long* dcyc;
  hipMalloc(&A, (size_t)131072 * 2560 * 2); hipMalloc(&W, (size_t)2560 * 2560 * 2); hipMalloc(&dcyc, 256 * 8);
  hipMemset(A, 1, (size_t)131072 * 2560 * 2); hipMemset(W, 1, (size_t)2560 * 2560 * 2);
  printf("operands as in FFN-1: A [131072 x 640], W [2560 x 640]; 256 work-groups, tiles n-fastest per XCD\n");
  run<32, 4, 6, 3, 0, 3, 128>("BK32 4 waves 3 in flight  A+W (256+128 rows)  [ping-pong today]", A, W, K, N, dcyc);
  run<32, 8, 6, 3, 0, 3, 128>("BK32 8 waves 3 in flight  A+W (256+128 rows)", A, W, K, N, dcyc);
  run<64, 4, 3, 1, 0, 3, 128>("BK64 4 waves 1 in flight  A+W (256+128 rows)", A, W, K, N, dcyc);
  run<64, 4, 3, 2, 0, 3, 128>("BK64 4 waves 2 in flight  A+W (256+128 rows)", A, W, K, N, dcyc);
  run<64, 8, 3, 2, 0, 3, 128>("BK64 8 waves 2 in flight  A+W (256+128 rows)", A, W, K, N, dcyc);
  run<64, 8, 2, 1, 0, 3, 256>("BK64 8 waves 1 in flight  A+W (256+256 rows)  [classic 256x256]", A, W, K, N, dcyc);
  run<32, 8, 5, 3, 0, 3, 256>("BK32 8 waves 3 in flight  A+W (256+256 rows)", A, W, K, N, dcyc);
  run<32, 8, 5, 4, 0, 3, 256>("BK32 8 waves 4 in flight  A+W (256+256 rows)", A, W, K, N, dcyc);
  run<64, 8, 2, 1, 2, 3, 256>("BK64 8 waves 1 in flight  A+W (256+256 rows)  nt", A, W, K, N, dcyc);
  run<32, 8, 5, 4, 2, 3, 256>("BK32 8 waves 4 in flight  A+W (256+256 rows)  nt", A, W, K, N, dcyc);
  run<64, 8, 3, 2, 0, 1, 128>("BK64 8 waves 2 in flight  A only", A, W, K, N, dcyc);
  run<64, 8, 3, 2, 0, 2, 128>("BK64 8 waves 2 in flight  W only (128 rows)", A, W, K, N, dcyc);
  run<64, 8, 2, 1, 0, 2, 256>("BK64 8 waves 1 in flight  W only (256 rows)", A, W, K, N, dcyc);
  run<32, 8, 6, 4, 0, 1, 128>("BK32 8 waves 4 in flight  A only", A, W, K, N, dcyc);
  printf("---- K = 2560 operands (FFN-2 / dgrad): A [131072 x 2560], W [640 x 2560]\n");
  run<64, 8, 2, 1, 0, 3, 256>("BK64 8 waves 1 in flight  A+W (256+256 rows)  [classic 256x256]", A, W, 2560, 512, dcyc);
  run<32, 8, 5, 4, 0, 3, 256>("BK32 8 waves 4 in flight  A+W (256+256 rows)", A, W, 2560, 512, dcyc);
  run<64, 4, 3, 2, 0, 3, 128>("BK64 4 waves 2 in flight  A+W (256+128 rows)", A, W, 2560, 640, dcyc);
  return 0;
}
