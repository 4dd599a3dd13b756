// Weight-gradient GEMM on MFMA: dW[N,K] (+)= dY[M,N]^T * X[M,K], bf16 operands, fp32 accumulation.  gfx950 only.
//
// The contraction runs over the M = B*L token rows, along which BOTH operands are strided in memory.  The tiles are
// therefore staged exactly as they lie in HBM (64 token rows x 128 columns, coalesced global_load_lds), and the MFMA
// operand fragments (8 consecutive tokens for one output row / column) are gathered with ds_read_b64_tr_b16, the
// gfx950 LDS transpose read -- no transposed copy of any activation is ever written to HBM.
//   * output tile 128 (n) x 128 (k), 4 waves 2x2, 4x4 v_mfma_f32_16x16x32_bf16 accumulators per wave;
//   * the token range is split over S workgroups per output tile (S*tiles ~ 2 waves of 256 CUs); each writes an fp32
//     slab, a second kernel sums the slabs in fixed order (deterministic; no float atomics) into dW (= or +=);
//   * LDS rows are 256 B; 16-byte chunks are XOR-swizzled with 2*((row&3)|((row>>3)&1)<<2) so the 8 rows a half-wave
//     transposed read touches fall in 8 different 32-byte bank slots (source-side swizzle for the LDS-DMA).
#include "common.h"
#include "../../include/oneprot_hip.h"

#define TN_MAX_SLAB_TILES 512

static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0, 0, 0, 0};

__device__ __forceinline__ int tn_f(int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); }

__device__ __forceinline__ bf8_t tn_frag(const unsigned char* tile, int mb, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int r0 = mb + 8 * g + (i >> 2), r1 = r0 + 4;
  const int col = c0 + 4 * (i & 3);
  const int chunk = col >> 3, within = (col & 7) * 2;
  const unsigned char* p0 = tile + r0 * 256 + ((chunk ^ tn_f(r0)) << 4) + within;
  const unsigned char* p1 = tile + r1 * 256 + ((chunk ^ tn_f(r1)) << 4) + within;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  s16x8 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  return __builtin_bit_cast(bf8_t, o);
}

template <int N> __device__ __forceinline__ void tn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BT tokens per LDS stage (32 or 64), NSTAGE-deep ring filled by LDS-DMA with NSTAGE-1 stages in flight (counted vmcnt + raw barrier)
template <int BT, int NSTAGE, int MINW, bool RS>
__global__ void __launch_bounds__(256, MINW) k_gemm_tn(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, int M, int N, int K, int ldy, int ldx,
                                                       float* __restrict__ slab, float* __restrict__ bias_slab, int tiles_k, int tiles_all, int S, int m_per_split) {
  constexpr int TILE_BYTES = BT * 256;              // one operand tile: BT token rows x 128 columns
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int IPW = BT / 4 / 4;                   // global_load_lds instructions per wave per operand per stage (4 rows each)
  constexpr int LPS = 2 * IPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // XCD-aware decode: workgroups b, b+8, ... share an XCD (round-robin dispatch).  The (split, tile) work items are numbered split-major and
  // each XCD takes a contiguous eighth, so the tiles of one token split run on one XCD (two at a boundary) and every dY / X row is pulled
  // through the fabric into one or two L2s; with the plain (tile, split) grid each of the 8 L2s fetched nearly all of both operands.
  const int per_xcd = gridDim.x >> 3;                 // grid = 8 * ceil(tiles * S / 8); work items in split-major order, a contiguous run per XCD
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int w = xcd * per_xcd + seq;
  if (w >= tiles_all * S) return;
  const int split = w / tiles_all, tile = w - split * tiles_all;
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int mbeg = split * m_per_split;
  const int mend = min(M, mbeg + m_per_split);
  const int nsteps = (mend - mbeg + BT - 1) / BT;

  // staging: one instruction covers 4 token rows x 256 B
  const int srow = lane >> 4, schunk = lane & 15;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);
  int rows[IPW], ycol_ok[IPW], xcol_ok[IPW];
  size_t yoff[IPW], xoff[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    rows[i] = (wave * IPW + i) * 4 + srow;
    const int sc = schunk ^ tn_f(rows[i]);
    ycol_ok[i] = (n0 + sc * 8) < N;
    xcol_ok[i] = (k0 + sc * 8) < K;
    yoff[i] = (size_t)(n0 + sc * 8);
    xoff[i] = (size_t)(k0 + sc * 8);
  }
  auto stage = [&](int t, int buf) {
    unsigned char* sY = smem + buf * STAGE_BYTES;
    unsigned char* sX = sY + TILE_BYTES;
    const int mb = mbeg + t * BT;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int m = mb + rows[i];
      const bool mok = m < mend;
      const unsigned char* gy = (mok && ycol_ok[i]) ? reinterpret_cast<const unsigned char*>(dY + (size_t)m * ldy + yoff[i]) : zero;
      const unsigned char* gx = (mok && xcol_ok[i]) ? reinterpret_cast<const unsigned char*>(X + (size_t)m * ldx + xoff[i]) : zero;
      __builtin_amdgcn_global_load_lds(GLB_PTR(gy), LDS_PTR(sY + (wave * IPW + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(gx), LDS_PTR(sX + (wave * IPW + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // bias gradient db[n] = sum_m dY[m,n]: the k-tile-0 column of workgroups multiplies the dY fragments with an all-ones B operand
  // (every column of the 16x16 result then holds the row sums) -- 4 extra MFMAs per k-substep instead of a second pass over dY.
  const bool do_bias = bias_slab != nullptr && tk == 0 && wc == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bf8_t ones = __builtin_bit_cast(bf8_t, ones_u);

  if constexpr (RS) {
    // register-staged fill (global_load_dwordx4 -> VGPR -> ds_write_b128): an LDS-DMA piece moves only 1 KiB here (4 token rows) and costs
    // 60-185 issue cycles inside an MFMA phase (MI355X_MICROARCH.md), 8 pieces per 32 MFMAs; the plain pair costs ~20.  Two LDS buffers, the
    // loads of stage t+2 are issued after stage t+1 is committed and land during step t+1.
    u32x4 gy[IPW], gx[IPW];
    // Fast path (wave-uniform test): the work-group's tile lies inside [N, K] and its token range is whole stages -- every address is then a
    // uniform base (advanced per stage on the scalar unit) plus a constant 32-bit per-lane offset, no selects, no 64-bit multiplies: the
    // kernel is bound by instruction issue (32 MFMAs + 32 transpose reads + 16 fill instructions per stage), so this is not cosmetic.
    const bool fast = (n0 + 128 <= N) && (k0 + 128 <= K) && (mbeg + nsteps * BT <= mend) && ((size_t)BT * ldy * 2 < (1u << 31)) && ((size_t)BT * ldx * 2 < (1u << 31));
    unsigned yrel[IPW], xrel[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int sc = schunk ^ tn_f(rows[i]);
      yrel[i] = (unsigned)rows[i] * (unsigned)ldy * 2u + (unsigned)sc * 16u;
      xrel[i] = (unsigned)rows[i] * (unsigned)ldx * 2u + (unsigned)sc * 16u;
    }
    const unsigned char* ybase = reinterpret_cast<const unsigned char*>(dY + (size_t)mbeg * ldy + n0);
    const unsigned char* xbase = reinterpret_cast<const unsigned char*>(X + (size_t)mbeg * ldx + k0);
    auto fetch = [&](int t) {
      if (fast) {
        const unsigned char* yt = ybase + (size_t)t * BT * ldy * 2;
        const unsigned char* xt = xbase + (size_t)t * BT * ldx * 2;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
          gy[i] = *reinterpret_cast<const u32x4*>(yt + yrel[i]);
          gx[i] = *reinterpret_cast<const u32x4*>(xt + xrel[i]);
        }
        return;
      }
      const int mb = mbeg + t * BT;
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        const int m = mb + rows[i];
        const bool mok = m < mend;
        const unsigned char* py = (mok && ycol_ok[i]) ? reinterpret_cast<const unsigned char*>(dY + (size_t)m * ldy + yoff[i]) : zero;
        const unsigned char* px = (mok && xcol_ok[i]) ? reinterpret_cast<const unsigned char*>(X + (size_t)m * ldx + xoff[i]) : zero;
        gy[i] = *reinterpret_cast<const u32x4*>(py);
        gx[i] = *reinterpret_cast<const u32x4*>(px);
      }
    };
    auto commit = [&](int buf) {
      unsigned char* sY = smem + buf * STAGE_BYTES + lane * 16;
      unsigned char* sX = sY + TILE_BYTES;
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        *reinterpret_cast<u32x4*>(sY + (wave * IPW + i) * 1024) = gy[i];
        *reinterpret_cast<u32x4*>(sX + (wave * IPW + i) * 1024) = gx[i];
      }
    };
    if (nsteps > 0) { fetch(0); commit(0); }
    if (nsteps > 1) fetch(1);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      const unsigned char* sY = smem + (t & 1) * STAGE_BYTES;
      const unsigned char* sX = sY + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < BT / 32; ++kk) {
      bf8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = tn_frag(sY, kk * 32, wr * 64 + i * 16, lane);
        b[i] = tn_frag(sX, kk * 32, wc * 64 + i * 16, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
      }
    }
      if (t + 1 < nsteps) {
        commit((t + 1) & 1);
        if (t + 2 < nsteps) fetch(t + 2);
        __syncthreads();
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nsteps) stage(s, s);
    int buf = 0, nbuf = NSTAGE - 1;
    for (int t = 0; t < nsteps; ++t) {
      if (t + NSTAGE - 2 < nsteps) tn_wait_vmcnt<(NSTAGE - 2) * LPS>(); else tn_wait_vmcnt<0>();
      asm volatile("s_barrier" ::: "memory");
      if (t + NSTAGE - 1 < nsteps) stage(t + NSTAGE - 1, nbuf);
      const unsigned char* sY = smem + buf * STAGE_BYTES;
      const unsigned char* sX = sY + TILE_BYTES;
  #pragma unroll
      for (int kk = 0; kk < BT / 32; ++kk) {
        bf8_t a[4], b[4];
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = tn_frag(sY, kk * 32, wr * 64 + i * 16, lane);
          b[i] = tn_frag(sX, kk * 32, wc * 64 + i * 16, lane);
        }
  #pragma unroll
        for (int i = 0; i < 4; ++i)
  #pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
  #pragma unroll
          for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
        }
      }
      buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
      nbuf = (nbuf + 1 == NSTAGE) ? 0 : nbuf + 1;
    }
}
  float* out = slab + (size_t)split * N * K;
  const int fq = lane >> 4, fr = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wr * 64 + i * 16 + fq * 4 + r;
      if (n >= N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wc * 64 + j * 16 + fr;
        if (k < K) out[(size_t)n * K + k] = acc[i][j][r];
      }
    }
  if (do_bias && fr == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 64 + i * 16 + fq * 4 + r;
        if (n < N) bias_slab[(size_t)split * N + n] = accb[i][r];
      }
  }
}

__global__ void __launch_bounds__(256) k_tn_bias_reduce(const float* __restrict__ bias_slab, float* __restrict__ db, int N, int S, int accumulate) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int p = 0; p < S; ++p) s += bias_slab[(size_t)p * N + n];
  db[n] = accumulate ? db[n] + s : s;
}

__global__ void __launch_bounds__(256) k_tn_reduce(const float* __restrict__ slab, float* __restrict__ dW, size_t n4, size_t stride4, int S, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 s = reinterpret_cast<const float4*>(slab)[i];
    for (int p = 1; p < S; ++p) {
      const float4 v = reinterpret_cast<const float4*>(slab)[i + (size_t)p * stride4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) { const float4 o = reinterpret_cast<const float4*>(dW)[i]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    reinterpret_cast<float4*>(dW)[i] = s;
  }
}

// -1 = auto (= 2).  Measured in-process (tools/tn_ab.py, cfg-2 shapes, after the XCD-contiguous work mapping): register-staged fill (2) 492 / 174 /
// 590 / 568 us on the QKV / out / FFN-1 / FFN-2 weight gradients, LDS-DMA ring (0) 513 / 186 / 625 / 613, 32-token x3 ring (1) ~25 % behind;
// a 256 x 256-tile version (twice the FLOPs per staged byte) was built and ran 20-30 % slower than either and was dropped.
static int g_tn_variant = -1;
extern "C" void oneprot_gemm_tn_variant(int v) { g_tn_variant = v; }

// number of token splits: about TN_MAX_SLAB_TILES workgroups in total (two per CU), at least 256 tokens per split, at most 64
static inline int tn_splits(int64_t M, int tiles) {
  int S = TN_MAX_SLAB_TILES / tiles;
  if (S < 1) S = 1;
  const int64_t max_by_m = (M + 255) / 256;
  if (M > 0 && S > max_by_m) S = (int)max_by_m;
  if (S > 64) S = 64;
  return S;
}

extern "C" size_t oneprot_gemm_bf16_tn_workspace(int N, int K) {
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  const int S = tn_splits(0, tiles);
  return (size_t)S * N * K * sizeof(float) + (size_t)S * N * sizeof(float);
}

extern "C" int oneprot_gemm_bf16_tn(const void* dY, const void* X, int64_t M, int N, int K, int ldy, int ldx, float* dW, float* dbias, void* workspace,
                                    size_t workspace_bytes, int accumulate, void* stream) {
  if (!dY || !X || !dW || !workspace || M <= 0 || N <= 0 || K <= 0 || M > 0x7fffffff) return OP_EINVAL;
  if ((N & 7) || (K & 7) || (ldy & 7) || (ldx & 7) || ldy < N || ldx < K || ((N * (int64_t)K) & 3)) return OP_EINVAL;
  if (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)workspace) & 15) return OP_EINVAL;
  const int variant = g_tn_variant >= 0 ? g_tn_variant : 2;
  const int tiles_n = (N + 127) / 128, tiles_k = (K + 127) / 128, tiles = tiles_n * tiles_k;
  const int S = tn_splits(M, tiles);
  if (workspace_bytes < (size_t)S * N * K * sizeof(float) + (dbias ? (size_t)S * N * sizeof(float) : 0)) return OP_EINVAL;      // slabs would overrun
  hipStream_t s = (hipStream_t)stream;
  float* bias_slab = dbias ? (float*)workspace + (size_t)S * N * K : nullptr;
  // variant 0: 64-token stages, 2-stage LDS-DMA ring (64 KB, 2 workgroups/CU); 1: 32-token stages, 3-stage ring (48 KB, 3 workgroups/CU);
  // 2: as 0 with register-staged fill
  const int BT = variant == 1 ? 32 : 64;
  int m_per = (int)((M + S - 1) / S);
  m_per = ((m_per + BT - 1) / BT) * BT;
  if (variant == 0) {
    static bool c0 = false;
    if (!c0) { if (hipFuncSetAttribute((const void*)k_gemm_tn<64, 2, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 256) != hipSuccess) return OP_ELAUNCH; c0 = true; }
    hipLaunchKernelGGL((k_gemm_tn<64, 2, 2, false>), dim3(8 * ((tiles * S + 7) / 8)), dim3(256), 2 * 2 * 64 * 256, s, (const bf16_t*)dY, (const bf16_t*)X, (int)M, N, K, ldy, ldx,
                       (float*)workspace, bias_slab, tiles_k, tiles, S, m_per);
  } else if (variant == 2) {
    static bool c2 = false;
    if (!c2) { if (hipFuncSetAttribute((const void*)k_gemm_tn<64, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 256) != hipSuccess) return OP_ELAUNCH; c2 = true; }
    hipLaunchKernelGGL((k_gemm_tn<64, 2, 2, true>), dim3(8 * ((tiles * S + 7) / 8)), dim3(256), 2 * 2 * 64 * 256, s, (const bf16_t*)dY, (const bf16_t*)X, (int)M, N, K, ldy, ldx,
                       (float*)workspace, bias_slab, tiles_k, tiles, S, m_per);
  } else {
    static bool c1 = false;
    if (!c1) { if (hipFuncSetAttribute((const void*)k_gemm_tn<32, 3, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 32 * 256) != hipSuccess) return OP_ELAUNCH; c1 = true; }
    hipLaunchKernelGGL((k_gemm_tn<32, 3, 3, false>), dim3(8 * ((tiles * S + 7) / 8)), dim3(256), 3 * 2 * 32 * 256, s, (const bf16_t*)dY, (const bf16_t*)X, (int)M, N, K, ldy, ldx,
                       (float*)workspace, bias_slab, tiles_k, tiles, S, m_per);
  }
  if (dbias) hipLaunchKernelGGL(k_tn_bias_reduce, dim3((N + 255) / 256), dim3(256), 0, s, (const float*)bias_slab, dbias, N, S, accumulate);
  const size_t n4 = ((size_t)N * K) >> 2;
  size_t blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_tn_reduce, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)workspace, dW, n4, n4, S, accumulate);
  return launch_status();
}
