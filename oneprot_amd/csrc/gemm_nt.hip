// bf16 MFMA GEMM, C[M,N] = A[M,K] * B[N,K]^T (+ fused epilogue), fp32 accumulation.  gfx950 only.
//
// Structure:
//   * workgroup = 256 threads = 4 waves (2x2), block tile 128x128, BK = 64; each wave owns a 64x64 sub-tile as
//     4x4 v_mfma_f32_16x16x32_bf16 accumulators (64 VGPRs);
//   * A and B K-slices go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR staging), double buffered,
//     one barrier per K-step; the LDS image is lane-linear (128-byte rows) and the bank-conflict XOR swizzle
//     (chunk ^= (row>>1)&7) is applied on the per-lane SOURCE address and again on the ds_read_b128 address;
//   * rows/cols beyond M/N/K read from a 16-byte zero page (per-lane source select), so any M, any N%8==0,
//     any K%8==0 works without a tail path;
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs take interleaved row panels and sweep N fastest, so an
//     A panel is fetched from HBM once per XCD and the weight matrix stays L2-resident;
//   * epilogue: each wave parks its 64x64 fp32 tile in its own padded LDS region and re-reads it row-wise, 8 columns
//     per lane -> 16-byte coalesced stores in whatever layout the consumer wants (bias, erf-GELU, fp32 residual,
//     q-scale + RoPE + head-major q/k/v, GELU').
#include "common.h"
#include "../../include/oneprot_hip.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (BM * BK * 2 + BN * BK * 2)     // 32 KiB
#define EPI_LD 68                                   // fp32 row pitch of a wave's 64x64 epilogue tile
#define EPI_BYTES (4 * 64 * EPI_LD * 4)             // 69632
#define LDS_BYTES (EPI_BYTES > 2 * STAGE_BYTES ? EPI_BYTES : 2 * STAGE_BYTES)

static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0, 0, 0, 0};

struct GemmArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb;
  const float* bias;
  void* out0; void* out1; void* out2;
  const void* aux;
  const float* cos; const float* sin;
  float q_scale;
  int L, H, hd;
  int tiles_m, tiles_n;
};

template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_nt(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- XCD-aware tile assignment
  const int bid = blockIdx.x;
  const int xcd = bid & 7, seq = bid >> 3;
  const int tm = (seq / p.tiles_n) * 8 + xcd;
  const int tn = seq % p.tiles_n;
  if (tm >= p.tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- per-lane staging addresses: wave w issues instructions i=0..3 for A and for B, each covering 8 rows x 128 B
  const int srow = lane >> 3;                 // row within the 8-row group
  const int schunk = lane & 7;                // LDS chunk (16 B) within the 128-byte row
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);
  const unsigned char* a_src[4]; const unsigned char* b_src[4];
  int a_ok[4], b_ok[4], src_chunk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + srow;             // 0..127
    src_chunk[i] = schunk ^ ((row >> 1) & 7);
    a_ok[i] = (m0 + row) < p.M;
    b_ok[i] = (n0 + row) < p.N;
    a_src[i] = reinterpret_cast<const unsigned char*>(p.A + (size_t)(a_ok[i] ? m0 + row : 0) * p.lda + src_chunk[i] * 8);
    b_src[i] = reinterpret_cast<const unsigned char*>(p.B + (size_t)(b_ok[i] ? n0 + row : 0) * p.ldb + src_chunk[i] * 8);
  }
  const int nk = (p.K + BK - 1) / BK;

  auto stage = [&](int t, int buf) {
    unsigned char* sA = smem + buf * STAGE_BYTES;
    unsigned char* sB = sA + BM * BK * 2;
    const int k0 = t * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool kin = (k0 + src_chunk[i] * 8) < p.K;
      const unsigned char* ga = (a_ok[i] && kin) ? a_src[i] + (size_t)k0 * 2 : zero;
      const unsigned char* gb = (b_ok[i] && kin) ? b_src[i] + (size_t)k0 * 2 : zero;
      __builtin_amdgcn_global_load_lds(GLB_PTR(ga), LDS_PTR(sA + (wave * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(gb), LDS_PTR(sB + (wave * 4 + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes within a stage's A or B image)
  const int frow = lane & 15, fq = lane >> 4;
  int a_off[4][2], b_off[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wr * 64 + i * 16 + frow, rb = wc * 64 + i * 16 + frow;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fq;
      a_off[i][kk] = ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4);
      b_off[i][kk] = rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4);
    }
  }

  stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
    const unsigned char* sA = smem + (t & 1) * STAGE_BYTES;
    const unsigned char* sB = sA + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf8_t*>(sA + a_off[i][kk]);
        b[i] = *reinterpret_cast<const bf8_t*>(sB + b_off[i][kk]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();          // everyone done with the staging buffers before they are reused as epilogue tiles

  // ---- epilogue: wave-private 64x64 fp32 tile in LDS
  float* et = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) et[(i * 16 + fq * 4 + r) * EPI_LD + j * 16 + frow] = acc[i][j][r];
  // (same wave wrote and reads: LDS ops of one wave complete in order; the compiler inserts the lgkmcnt wait)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int er = lane >> 3, ec = (lane & 7) * 8;
  const int gn = n0 + wc * 64 + ec;
  if (gn >= p.N) return;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if (p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + gn), b1 = *reinterpret_cast<const float4*>(p.bias + gn + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  // QKV/RoPE constants for this lane's 8 columns
  int sec = 0, head = 0, j0 = 0, pc = 0; float pbias8[8]; float sgn = 0.f;
  if (EPI == ONEPROT_EPI_QKV_ROPE) {
    const int dm = p.H * p.hd;
    sec = gn / dm;
    const int within = gn - sec * dm;
    head = within / p.hd; j0 = within - head * p.hd;
    const int half = p.hd >> 1;
    const bool lo = j0 < half;
    pc = ec + (lo ? half : -half);          // partner columns inside the wave tile
    sgn = lo ? -1.f : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) pbias8[e] = 0.f;
    if (p.bias && sec < 2) {
      const int pg = gn + (lo ? half : -half);
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + pg), b1 = *reinterpret_cast<const float4*>(p.bias + pg + 4);
      pbias8[0] = b0.x; pbias8[1] = b0.y; pbias8[2] = b0.z; pbias8[3] = b0.w; pbias8[4] = b1.x; pbias8[5] = b1.y; pbias8[6] = b1.z; pbias8[7] = b1.w;
    }
  }
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int r = pass * 8 + er;
    const int gm = m0 + wr * 64 + r;
    if (gm >= p.M) continue;
    float v[8];
    {
      const float4 v0 = *reinterpret_cast<const float4*>(et + r * EPI_LD + ec), v1 = *reinterpret_cast<const float4*>(et + r * EPI_LD + ec + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bias8[e];
    const size_t o = (size_t)gm * p.N + gn;
    if (EPI == ONEPROT_EPI_BF16) {
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      *reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o) = w;
    } else if (EPI == ONEPROT_EPI_F32) {
      float* c = (float*)p.out0 + o;
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else if (EPI == ONEPROT_EPI_BIAS_GELU) {
      if (p.out1) {
        u32x4 z; z.x = pack2bf(v[0], v[1]); z.y = pack2bf(v[2], v[3]); z.z = pack2bf(v[4], v[5]); z.w = pack2bf(v[6], v[7]);
        *reinterpret_cast<u32x4*>((bf16_t*)p.out1 + o) = z;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      *reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o) = w;
    } else if (EPI == ONEPROT_EPI_BIAS_RESID) {
      const float* rs = (const float*)p.aux + o;
      const float4 r0 = *reinterpret_cast<const float4*>(rs), r1 = *reinterpret_cast<const float4*>(rs + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      float* c = (float*)p.out0 + o;
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
      if (p.out1) {
        u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
        *reinterpret_cast<u32x4*>((bf16_t*)p.out1 + o) = w;
      }
    } else if (EPI == ONEPROT_EPI_GELU_BWD) {
      const u32x4 z = *reinterpret_cast<const u32x4*>((const bf16_t*)p.aux + o);
      v[0] *= gelu_erf_grad(bflo(z.x)); v[1] *= gelu_erf_grad(bfhi(z.x)); v[2] *= gelu_erf_grad(bflo(z.y)); v[3] *= gelu_erf_grad(bfhi(z.y));
      v[4] *= gelu_erf_grad(bflo(z.z)); v[5] *= gelu_erf_grad(bfhi(z.z)); v[6] *= gelu_erf_grad(bflo(z.w)); v[7] *= gelu_erf_grad(bfhi(z.w));
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      *reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o) = w;
    } else if (EPI == ONEPROT_EPI_QKV_ROPE) {
      const int b = gm / p.L, l = gm - b * p.L;
      if (sec < 2) {
        float pv[8];
        const float4 q0 = *reinterpret_cast<const float4*>(et + r * EPI_LD + pc), q1 = *reinterpret_cast<const float4*>(et + r * EPI_LD + pc + 4);
        pv[0] = q0.x; pv[1] = q0.y; pv[2] = q0.z; pv[3] = q0.w; pv[4] = q1.x; pv[5] = q1.y; pv[6] = q1.z; pv[7] = q1.w;
        const int half = p.hd >> 1;
        const int jj = j0 < half ? j0 : j0 - half;
        const float* cs = p.cos + (size_t)l * half + jj;
        const float* sn = p.sin + (size_t)l * half + jj;
        const float sc = sec == 0 ? p.q_scale : 1.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = v[e] * sc, xp = (pv[e] + pbias8[e]) * sc;
          v[e] = x * cs[e] + sgn * xp * sn[e];
        }
      }
      bf16_t* dst = (bf16_t*)(sec == 0 ? p.out0 : (sec == 1 ? p.out1 : p.out2));
      const size_t oo = (((size_t)b * p.H + head) * p.L + l) * p.hd + j0;
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      *reinterpret_cast<u32x4*>(dst + oo) = w;
    }
  }
}

template <int EPI>
static int launch_gemm(const GemmArgs& a, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k_gemm_nt<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return OP_ELAUNCH;
    configured = true;
  }
  const int grid = ((a.tiles_m + 7) / 8) * 8 * a.tiles_n;
  hipLaunchKernelGGL(k_gemm_nt<EPI>, dim3(grid), dim3(256), LDS_BYTES, s, a);
  return launch_status();
}

extern "C" int oneprot_gemm_bf16_nt(const void* A, const void* Bw, int64_t M, int N, int K, int lda, int ldb, int epilogue, const float* bias,
                                    void* out0, void* out1, void* out2, const void* aux, const float* rope_cos, const float* rope_sin, float q_scale,
                                    int L, int H, int hd, void* stream) {
  if (!A || !Bw || !out0 || M <= 0 || N <= 0 || K <= 0 || M > 0x7fffffff) return OP_EINVAL;
  if ((N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) return OP_EINVAL;
  if (((uintptr_t)A | (uintptr_t)Bw | (uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)out2 | (uintptr_t)aux | (uintptr_t)bias) & 15) return OP_EINVAL;
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)Bw; a.M = (int)M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.bias = bias;
  a.out0 = out0; a.out1 = out1; a.out2 = out2; a.aux = aux; a.cos = rope_cos; a.sin = rope_sin; a.q_scale = q_scale; a.L = L; a.H = H; a.hd = hd;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (N + BN - 1) / BN;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case ONEPROT_EPI_BF16: return launch_gemm<ONEPROT_EPI_BF16>(a, s);
    case ONEPROT_EPI_F32: return launch_gemm<ONEPROT_EPI_F32>(a, s);
    case ONEPROT_EPI_BIAS_GELU: if (!bias) return OP_EINVAL; return launch_gemm<ONEPROT_EPI_BIAS_GELU>(a, s);
    case ONEPROT_EPI_BIAS_RESID: if (!aux) return OP_EINVAL; return launch_gemm<ONEPROT_EPI_BIAS_RESID>(a, s);
    case ONEPROT_EPI_GELU_BWD: if (!aux) return OP_EINVAL; return launch_gemm<ONEPROT_EPI_GELU_BWD>(a, s);
    case ONEPROT_EPI_QKV_ROPE:
      if (!out1 || !out2 || !rope_cos || !rope_sin || L <= 0 || H <= 0 || (hd != 16 && hd != 32 && hd != 64) || N != 3 * H * hd || (M % L) != 0 || ((H * hd) & 63))
        return OP_EINVAL;
      return launch_gemm<ONEPROT_EPI_QKV_ROPE>(a, s);
    default: return OP_EINVAL;
  }
}
