// bf16 MFMA GEMM, C[M,N] = A[M,K] * B[N,K]^T (+ fused epilogue), fp32 accumulation.  gfx950 only.
//
// Structure:
//   * three block shapes from one template <WM x WN waves, MT 16-row tiles per wave>, BK = 64, v_mfma_f32_16x16x32_bf16:
//       256x256 (8 waves 2x4, wave tile 128x64, 2 LDS stages)   large N: half the L2->LDS bytes per FLOP of a 128^2 tile
//       256x128 (8 waves 4x2, wave tile  64x64, 3 LDS stages)   N = 640-class outputs
//       128x128 (4 waves 2x2, wave tile  64x64, 2 LDS stages, 2 blocks/CU)   small problems
//     (measured: the 128^2 / 2-stage form is bound by bytes-in-flight / load latency at K = 640 -- 64 KB in flight per CU
//      sustains ~12 TB/s of L2->LDS traffic -- so the larger shapes raise FLOPs per staged byte and the 3-stage form keeps
//      two K-slices in flight behind a COUNTED s_waitcnt vmcnt(N) and a raw s_barrier);
//   * A and B K-slices go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR staging); the LDS image is lane-linear
//     (128-byte rows) and the bank-conflict XOR swizzle (chunk ^= (row>>1)&7) is applied on the per-lane SOURCE address
//     and again on the ds_read_b128 address;
//   * rows/cols beyond M/N/K read from a 16-byte zero page (per-lane source select), so any M, any N%8==0,
//     any K%8==0 works without a tail path;
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs take interleaved row panels and sweep N fastest, so an
//     A panel is fetched from HBM once per XCD and the weight matrix stays L2-resident;
//   * epilogue: each wave parks its 64x64 fp32 tile in its own padded LDS region and re-reads it row-wise, 8 columns
//     per lane -> 16-byte coalesced stores in whatever layout the consumer wants (bias, erf-GELU, fp32 residual,
//     q-scale + RoPE + head-major q/k/v, GELU').
#include "common.h"
#include "../../include/oneprot_hip.h"
#include <type_traits>

#define EPI_LD 68                                   // fp32 row pitch of a wave's epilogue tile (64 columns + pad)

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

// WM x WN waves, MT x NTW 16x16 accumulator tiles per wave (wave tile = MT*16 x NTW*16), BKT = K-slice per LDS stage (32 or 64), NSTAGE ring depth,
// EPH = rows per epilogue staging pass (wave-private LDS tile EPH x 64 fp32)
template <int WM, int WN, int MT, int NTW, int BKT, int NSTAGE, int EPH> struct Shape {
  static constexpr int NW = WM * WN;
  static constexpr int BM_ = WM * MT * 16;
  static constexpr int BN_ = WN * NTW * 16;
  static constexpr int ROWB = BKT * 2;              // LDS row pitch in bytes
  static constexpr int RPI = 1024 / ROWB;           // rows covered by one global_load_lds wave-instruction
  static constexpr int A_IPW = BM_ / RPI / NW;      // instructions per wave per stage
  static constexpr int B_IPW = BN_ / RPI / NW;
  static constexpr int LPS = A_IPW + B_IPW;
  static constexpr int STAGE = (BM_ + BN_) * ROWB;
  static constexpr int EPI_WAVE = EPH * EPI_LD * 4;
  static constexpr int LDS = (NSTAGE * STAGE > NW * EPI_WAVE) ? NSTAGE * STAGE : NW * EPI_WAVE;
};

// 16-byte-chunk XOR swizzle keeping ds_read_b128 fragment reads conflict-free:
//   128-B rows (BK 64): chunk ^ ((row>>1)&7);   64-B rows (BK 32): chunk ^ G[(row>>2)&3], G = {0,2,3,1}
template <int BKT> __device__ __forceinline__ int swz(int row, int chunk) {
  return BKT == 64 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((0x1320 >> (((row >> 2) & 3) << 2)) & 3));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- epilogue: the wave parks EPH rows x 64 columns of its fp32 accumulators in its own padded LDS tile `et` and re-reads them row-wise,
// 8 columns per lane -> 16-byte coalesced stores in whatever layout the consumer wants.  FULL = the tile lies entirely inside [M, N] (no masks).
// (A persistent variant of the kernel -- LDS ring running continuously across tiles, epilogue stores never waited for, bias in LDS, store-count-
// exact vmcnt -- was built on this function, passed the tests and ran the FFN-1 launch in the same 0.70 ms as the plain form; dropped.)
template <int EPI, int MT, int NTW, int EPH, bool FULL>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[MT][NTW], float* et, int m0, int n0, int wr, int wc, int lane) {
  const int frow = lane & 15, fq = lane >> 4;
  const int er = lane >> 3, ec = (lane & 7) * 8;
#pragma unroll
  for (int nh = 0; nh < NTW / 4; ++nh) {       // 64-column halves of the wave tile
  const int gn = n0 + wc * (NTW * 16) + nh * 64 + ec;
  // QKV/RoPE constants for this lane's 8 columns
  int sec = 0, head = 0, j0 = 0, pc = 0; float sgn = 0.f;
  if (EPI == ONEPROT_EPI_QKV_ROPE && (FULL || gn < p.N)) {
    const int dm = p.H * p.hd;
    sec = gn / dm;
    const int within = gn - sec * dm;
    head = within / p.hd; j0 = within - head * p.hd;
    const int half = p.hd >> 1;
    const bool lo = j0 < half;
    pc = ec + (lo ? half : -half);          // partner columns inside the wave tile
    sgn = lo ? -1.f : 1.f;
    sec = __builtin_amdgcn_readfirstlane(sec);          // 64-column groups never straddle the q / k / v sections (H*hd % 64 == 0): scalar branches below
  }
  // QKV/RoPE: (sequence b, position l) of this lane's row and the head-major output offset, advanced by 8 rows per pass (no division,
  // no 32-bit multiplies inside the pass loop)
  int rp_l = 0; size_t rp_off = 0; const float* rp_cos = nullptr; const float* rp_sin = nullptr;
  if (EPI == ONEPROT_EPI_QKV_ROPE) {
    const int gm0 = m0 + wr * (MT * 16) + er;
    const int b0 = gm0 / p.L;
    rp_l = gm0 - b0 * p.L;
    rp_off = (((size_t)b0 * p.H + head) * p.L + rp_l) * p.hd + j0;
    const int hh = p.hd >> 1;
    const int jj = j0 < hh ? j0 : j0 - hh;
    rp_cos = p.cos + (size_t)rp_l * hh + jj;
    rp_sin = p.sin + (size_t)rp_l * hh + jj;
  }
#pragma unroll
  for (int half = 0; half < (MT * 16) / EPH; ++half) {
#pragma unroll
    for (int i = 0; i < EPH / 16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) et[(i * 16 + fq * 4 + r) * EPI_LD + j * 16 + frow] = acc[half * (EPH / 16) + i][nh * 4 + j][r];
    // same wave writes and reads: LDS operations of one wave complete in order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (FULL || gn < p.N) {
#pragma unroll 2
    for (int pass = 0; pass < EPH / 8; ++pass) {
    const int r = pass * 8 + er;
    const int gm = m0 + wr * (MT * 16) + half * EPH + r;
    if (EPI == ONEPROT_EPI_QKV_ROPE) {
      if (half > 0 || pass > 0) {            // 8 rows further than the previous pass
        const int hh = p.hd >> 1;
        rp_l += 8; rp_off += (size_t)8 * p.hd; rp_cos += 8 * hh; rp_sin += 8 * hh;
        if (rp_l >= p.L) { rp_l -= p.L; rp_off += (size_t)(p.H - 1) * p.L * p.hd; rp_cos -= (size_t)p.L * hh; rp_sin -= (size_t)p.L * hh; }
      }
    }
    if (!FULL && gm >= p.M) continue;
    float v[8];
    {
      const float4 v0 = *reinterpret_cast<const float4*>(et + r * EPI_LD + ec), v1 = *reinterpret_cast<const float4*>(et + r * EPI_LD + ec + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
    const size_t o = (size_t)gm * p.N + gn;
    if (EPI == ONEPROT_EPI_BF16) {
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      *reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o) = w;
    } else if (EPI == ONEPROT_EPI_F32) {
      float* c = (float*)p.out0 + o;
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else if (EPI == ONEPROT_EPI_BIAS_GELU) {
      float dg[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gelu_fwd_and_grad(v[e], v[e], dg[e]);
      if (p.out1) {          // gelu'(z), consumed by the GELU_BWD epilogue of the dgrad GEMM
        u32x4 z; z.x = pack2bf(dg[0], dg[1]); z.y = pack2bf(dg[2], dg[3]); z.z = pack2bf(dg[4], dg[5]); z.w = pack2bf(dg[6], dg[7]);
        *reinterpret_cast<u32x4*>((bf16_t*)p.out1 + o) = z;
      }
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
      const u32x4 z = *reinterpret_cast<const u32x4*>((const bf16_t*)p.aux + o);     // aux = gelu'(z) saved by the forward epilogue
      v[0] *= bflo(z.x); v[1] *= bfhi(z.x); v[2] *= bflo(z.y); v[3] *= bfhi(z.y);
      v[4] *= bflo(z.z); v[5] *= bfhi(z.z); v[6] *= bflo(z.w); v[7] *= bfhi(z.w);
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      *reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o) = w;
    } else if (EPI == ONEPROT_EPI_QKV_ROPE) {
      if (sec < 2) {
        float pv[8];
        const float4 q0 = *reinterpret_cast<const float4*>(et + r * EPI_LD + pc), q1 = *reinterpret_cast<const float4*>(et + r * EPI_LD + pc + 4);
        pv[0] = q0.x; pv[1] = q0.y; pv[2] = q0.z; pv[3] = q0.w; pv[4] = q1.x; pv[5] = q1.y; pv[6] = q1.z; pv[7] = q1.w;
        const float4 c0 = *reinterpret_cast<const float4*>(rp_cos), c1 = *reinterpret_cast<const float4*>(rp_cos + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(rp_sin), s1 = *reinterpret_cast<const float4*>(rp_sin + 4);
        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sc = sec == 0 ? p.q_scale : 1.0f;
        const float sp = sgn * sc;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] * sc) * cs[e] + (pv[e] * sp) * sn[e];
      }
      bf16_t* dst = (bf16_t*)(sec == 0 ? p.out0 : (sec == 1 ? p.out1 : p.out2));
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      *reinterpret_cast<u32x4*>(dst + rp_off) = w;
    }
  }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads of this pass done before the next pass overwrites the tile
  }
  }
}

template <int EPI, int WM, int WN, int MT, int NTW, int BKT, int NSTAGE, int EPH, int MINW, bool PIPE, bool FULLT>
__global__ void __launch_bounds__(WM * WN * 64, MINW) k_gemm_nt(const GemmArgs p) {
  typedef Shape<WM, WN, MT, NTW, BKT, NSTAGE, EPH> S;
  constexpr int CH = BKT / 8;                      // chunks per row
  constexpr int KK = BKT / 32;                     // MFMA k-substeps per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- XCD-aware, L2-blocked tile assignment.  Blocks b and b+8 share an XCD (round-robin dispatch), so XCD x walks the row panels
  // x, x+8, ... ; inside an XCD the order is super-tiles of SUP_M panels x SUP_N column tiles (n fastest) so that the A panels and
  // the W slices of one super-tile (~3 MB) stay resident in the XCD's 4 MB L2 while its ~40 tiles run (measured with FETCH_SIZE:
  // 1.65 GB -> 1.1 GB of fabric reads for the FFN-1 launch, whose operand set is 171 MB).
  const int bid = blockIdx.x;
  const int xcd = bid & 7, seq = bid >> 3;
  constexpr int SUP_M = 4, SUP_N = 10;
  const int pm_total = (p.tiles_m + 7) >> 3;
  const int per_mgroup = SUP_M * SUP_N * ((p.tiles_n + SUP_N - 1) / SUP_N);      // slots per m-group (partial groups leave idle slots)
  const int mg = seq / per_mgroup;
  int r_ = seq - mg * per_mgroup;
  const int mb_here = min(SUP_M, pm_total - mg * SUP_M);
  if (mb_here <= 0) return;
  const int ng = r_ / (mb_here * SUP_N);
  r_ -= ng * mb_here * SUP_N;
  const int nb_here = min(SUP_N, p.tiles_n - ng * SUP_N);
  if (nb_here <= 0 || r_ >= mb_here * nb_here) return;
  const int m_in = r_ / nb_here, n_in = r_ - m_in * nb_here;
  const int tm = (mg * SUP_M + m_in) * 8 + xcd;
  const int tn = ng * SUP_N + n_in;
  if (tm >= p.tiles_m) return;
  const int m0 = tm * S::BM_, n0 = tn * S::BN_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keeps LDS-DMA destinations and tile bases on the scalar unit
  const int wr = wave / WN, wc = wave % WN;

  // ---- per-lane staging addresses: one global_load_lds instruction covers RPI rows of ROWB bytes (1 KiB, lane-linear)
  const int srow = lane / CH;
  const int schunk = lane % CH;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);
  const unsigned char* a_src[S::A_IPW]; const unsigned char* b_src[S::B_IPW];
  int a_chunk[S::A_IPW], b_chunk[S::B_IPW];
  bool a_ok[S::A_IPW], b_ok[S::B_IPW];
#pragma unroll
  for (int i = 0; i < S::A_IPW; ++i) {
    const int row = (wave * S::A_IPW + i) * S::RPI + srow;
    a_chunk[i] = swz<BKT>(row, schunk);
    a_ok[i] = (m0 + row) < p.M;
    a_src[i] = reinterpret_cast<const unsigned char*>(p.A + (size_t)(a_ok[i] ? m0 + row : 0) * p.lda + a_chunk[i] * 8);
  }
#pragma unroll
  for (int i = 0; i < S::B_IPW; ++i) {
    const int row = (wave * S::B_IPW + i) * S::RPI + srow;
    b_chunk[i] = swz<BKT>(row, schunk);
    b_ok[i] = (n0 + row) < p.N;
    b_src[i] = reinterpret_cast<const unsigned char*>(p.B + (size_t)(b_ok[i] ? n0 + row : 0) * p.ldb + b_chunk[i] * 8);
  }
  const int nk = (p.K + BKT - 1) / BKT;

  // FULLT (every tile inside [M, N], K a multiple of the K-slice -- chosen by the host): no zero-page selects, and each source address is
  // a wave-uniform base (tile origin + K offset, kept and advanced on the scalar unit) plus a constant 32-bit per-lane offset, i.e. the
  // saddr + voffset form of global_load_lds: the main loop's vector ALU work per K-step drops from ~55 instructions to a handful.
  unsigned a_rel[S::A_IPW], b_rel[S::B_IPW];
#pragma unroll
  for (int i = 0; i < S::A_IPW; ++i) a_rel[i] = (unsigned)((wave * S::A_IPW + i) * S::RPI + srow) * (unsigned)p.lda * 2u + (unsigned)a_chunk[i] * 16u;
#pragma unroll
  for (int i = 0; i < S::B_IPW; ++i) b_rel[i] = (unsigned)((wave * S::B_IPW + i) * S::RPI + srow) * (unsigned)p.ldb * 2u + (unsigned)b_chunk[i] * 16u;
  const unsigned char* a_tile = reinterpret_cast<const unsigned char*>(p.A + (size_t)m0 * p.lda);
  const unsigned char* b_tile = reinterpret_cast<const unsigned char*>(p.B + (size_t)n0 * p.ldb);

  auto stage = [&](int t, int buf) {
    unsigned char* sA = smem + buf * S::STAGE;
    unsigned char* sB = sA + S::BM_ * S::ROWB;
    const int k0 = t * BKT;
    if constexpr (FULLT) {
      const unsigned char* ak = a_tile + (size_t)k0 * 2;
      const unsigned char* bk = b_tile + (size_t)k0 * 2;
#pragma unroll
      for (int i = 0; i < S::A_IPW; ++i) __builtin_amdgcn_global_load_lds(GLB_PTR(ak + a_rel[i]), LDS_PTR(sA + (wave * S::A_IPW + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < S::B_IPW; ++i) __builtin_amdgcn_global_load_lds(GLB_PTR(bk + b_rel[i]), LDS_PTR(sB + (wave * S::B_IPW + i) * 1024), 16, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < S::A_IPW; ++i) {
      const unsigned char* ga = (a_ok[i] && (k0 + a_chunk[i] * 8) < p.K) ? a_src[i] + (size_t)k0 * 2 : zero;
      __builtin_amdgcn_global_load_lds(GLB_PTR(ga), LDS_PTR(sA + (wave * S::A_IPW + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < S::B_IPW; ++i) {
      const unsigned char* gb = (b_ok[i] && (k0 + b_chunk[i] * 8) < p.K) ? b_src[i] + (size_t)k0 * 2 : zero;
      __builtin_amdgcn_global_load_lds(GLB_PTR(gb), LDS_PTR(sB + (wave * S::B_IPW + i) * 1024), 16, 0, 0);
    }
  };

  // accumulators start at the bias of their column (C layout: column = lane & 15 within each 16-wide tile): the epilogues then have no bias
  // add and, for QKV/RoPE, the rotation partner read back from the staging tile already carries its own bias
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int gc = n0 + wc * (NTW * 16) + j * 16 + (lane & 15);
    const float bj = (p.bias && gc < p.N) ? p.bias[gc] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][j] = (f32x4){bj, bj, bj, bj};
  }

  // fragment read offsets (bytes within a stage's A or B image) for k-substep 0; substep 1 (BK 64) flips chunk bit 2
  const int frow = lane & 15, fq = lane >> 4;
  int a_off[MT], b_off[NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int ra = wr * (MT * 16) + i * 16 + frow;
    a_off[i] = ra * S::ROWB + (swz<BKT>(ra, fq) << 4);
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int rb = wc * (NTW * 16) + j * 16 + frow;
    b_off[j] = rb * S::ROWB + (swz<BKT>(rb, fq) << 4);
  }

  if constexpr (PIPE) {
    // ---- K loop.  NSTAGE-deep LDS ring filled by LDS-DMA, counted vmcnt + raw s_barrier (never a vmcnt(0) drain mid-loop), and the MFMA
    // operand fragments are software-pipelined in registers: while the MFMAs of sub-step (t,kk) run, the ds_read_b128 of the next sub-step
    // are already in flight (two fragment sets, selected by compile-time parity; the t loop is unrolled by two so no register copies).
    //   end of K-step t:  own fragment reads retired (lgkmcnt 0) -> stage t+1 landed (counted vmcnt) -> barrier -> the buffer of stage t
    //   is free for everyone -> refill it with stage t+NSTAGE -> read the first fragments of stage t+1 -> MFMAs of (t, last kk).
    bf8_t fa[2][MT], fb[2][NTW];
    auto load_frags = [&](int set, const unsigned char* sA, const unsigned char* sB, int kk) {
  #pragma unroll
      for (int j = 0; j < NTW; ++j) fb[set][j] = *reinterpret_cast<const bf8_t*>(sB + (b_off[j] ^ (kk << 6)));
  #pragma unroll
      for (int i = 0; i < MT; ++i) fa[set][i] = *reinterpret_cast<const bf8_t*>(sA + (a_off[i] ^ (kk << 6)));
    };
  #pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
      if (s < nk) stage(s, s);
    if (nk >= NSTAGE) wait_vmcnt<(NSTAGE - 1) * S::LPS>(); else wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    load_frags(0, smem, smem + S::BM_ * S::ROWB, 0);
    int buf = 0;
    auto kstep = [&](int t, auto parity) {
      constexpr int P = decltype(parity)::value;
  #pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        constexpr int dummy = 0; (void)dummy;
        const int cur = (P * KK + kk) & 1;
        const int nxt = cur ^ 1;
        if (kk + 1 < KK) {
          const unsigned char* sA = smem + buf * S::STAGE;
          load_frags(nxt, sA, sA + S::BM_ * S::ROWB, kk + 1);
        } else if (t + 1 < nk) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (t + NSTAGE - 1 < nk) wait_vmcnt<(NSTAGE - 2) * S::LPS>(); else wait_vmcnt<0>();
          asm volatile("s_barrier" ::: "memory");
          if (t + NSTAGE < nk) stage(t + NSTAGE, buf);
          const int nb = (buf + 1 == NSTAGE) ? 0 : buf + 1;
          const unsigned char* sA = smem + nb * S::STAGE;
          load_frags(nxt, sA, sA + S::BM_ * S::ROWB, 0);
        }
  #pragma unroll
        for (int i = 0; i < MT; ++i)
  #pragma unroll
          for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      }
      buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    };
    for (int t = 0; t < nk; t += 2) {
      kstep(t, std::integral_constant<int, 0>{});
      if (t + 1 < nk) kstep(t + 1, std::integral_constant<int, 1>{});
    }
  } else {
    // ---- K loop (plain form, used where the second fragment set would cost a resident workgroup): NSTAGE-deep LDS ring, NSTAGE-1
    // K-slices in flight, counted vmcnt + raw barrier (never a vmcnt(0) drain mid-loop)
  #pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nk) stage(s, s);
    int buf = 0, nbuf = NSTAGE - 1;
    for (int t = 0; t < nk; ++t) {
      if (t + NSTAGE - 2 < nk) wait_vmcnt<(NSTAGE - 2) * S::LPS>(); else wait_vmcnt<0>();
      asm volatile("s_barrier" ::: "memory");
      if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1, nbuf);
      const unsigned char* sA = smem + buf * S::STAGE;
      const unsigned char* sB = sA + S::BM_ * S::ROWB;
  #pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        bf8_t a[MT], b[NTW];
  #pragma unroll
        for (int j = 0; j < NTW; ++j) b[j] = *reinterpret_cast<const bf8_t*>(sB + (b_off[j] ^ (kk << 6)));
  #pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf8_t*>(sA + (a_off[i] ^ (kk << 6)));
  #pragma unroll
        for (int i = 0; i < MT; ++i)
  #pragma unroll
          for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
      nbuf = (nbuf + 1 == NSTAGE) ? 0 : nbuf + 1;
    }
  }
  __syncthreads();          // everyone done with the staging ring before it is reused as epilogue tiles

  // ---- epilogue (staging tiles reuse the ring memory)
  gemm_epilogue<EPI, MT, NTW, EPH, FULLT>(p, acc, reinterpret_cast<float*>(smem) + wave * EPH * EPI_LD, m0, n0, wr, wc, lane);
}

template <int EPI, int WM, int WN, int MT, int NTW, int BKT, int NSTAGE, int EPH, int MINW, bool PIPE, bool FULLT>
static int launch_shape_full(GemmArgs a, hipStream_t s) {
  typedef Shape<WM, WN, MT, NTW, BKT, NSTAGE, EPH> S;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k_gemm_nt<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, FULLT>, hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS) != hipSuccess)
      return OP_ELAUNCH;
    configured = true;
  }
  a.tiles_m = (a.M + S::BM_ - 1) / S::BM_;
  a.tiles_n = (a.N + S::BN_ - 1) / S::BN_;
  const int pm_total = (a.tiles_m + 7) / 8;
  const int grid = ((pm_total + 3) / 4) * (4 * 10 * ((a.tiles_n + 9) / 10)) * 8;       // super-tile slots (SUP_M=4, SUP_N=10); surplus blocks exit at once
  hipLaunchKernelGGL((k_gemm_nt<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, FULLT>), dim3(grid), dim3(S::NW * 64), S::LDS, s, a);
  return launch_status();
}

// FULL-tile specialisation only for the shapes the heuristic picks (keeps the number of kernel instantiations down)
template <int EPI, int WM, int WN, int MT, int NTW, int BKT, int NSTAGE, int EPH, int MINW, bool PIPE, bool TRY_FULL = false>
static int launch_shape(GemmArgs a, hipStream_t s) {
  typedef Shape<WM, WN, MT, NTW, BKT, NSTAGE, EPH> S;
  if constexpr (TRY_FULL) {
    const bool full = a.M % S::BM_ == 0 && a.N % S::BN_ == 0 && a.K % BKT == 0 && (size_t)S::BM_ * a.lda * 2 < (1ull << 31) && (size_t)S::BN_ * a.ldb * 2 < (1ull << 31);
    if (full) return launch_shape_full<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, true>(a, s);
  }
  return launch_shape_full<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, false>(a, s);
}

static int g_force_shape = -1;     // test / tuning hook, see launch_gemm
extern "C" void oneprot_gemm_force_shape(int shape) { g_force_shape = shape; }

// shapes: 0 = 128x128 (BK32, 3 stages, pipelined fragments, 48 KB LDS -> 3 blocks/CU)     1 = 256x128 (BK32, 3 stages, 72 KB -> 2 blocks/CU)
//         2 = 256x256 (BK32, 4 stages, pipelined fragments, 128 KB)   3 = 128x128 BK64 2 stages (first version; still the best for long K)
//         4 = 256x256 BK64 2 stages, pipelined fragments               5 = 256x256 BK32 4 stages, plain loop
// Heuristic from in-process A/B on the training shapes (tools/gemm_ab.py; all shapes lie within ~10 % of each other, vendor hipBLASLt
// runs the same plain shapes at 650-1030 TFLOP/s): long K -> 3, wide N with short K -> 4, otherwise 1; small problems -> 0.
template <int EPI>
static int launch_gemm(const GemmArgs& a, hipStream_t s) {
  int shape;
  if (g_force_shape >= 0) shape = g_force_shape;
  else if (a.M < 2048) shape = 0;
  else if (a.K >= 1024) shape = 3;
  else if (a.N >= 2048) shape = 4;
  else shape = 1;
  switch (shape) {
    case 7: return launch_shape<EPI, 2, 2, 8, 8, 32, 4, 32, 1, false>(a, s);      // 256x256, 4 waves x (128x128), BK32 x4, one wave per SIMD
    case 6: return launch_shape<EPI, 2, 2, 8, 8, 64, 2, 32, 1, false>(a, s);      // 256x256, 4 waves x (128x128), BK64 x2, one wave per SIMD
    case 5: return launch_shape<EPI, 2, 4, 8, 4, 32, 4, 32, 2, false>(a, s);      // 256x256 without fragment pipelining (A/B runs)
    case 4: return launch_shape<EPI, 2, 4, 8, 4, 64, 2, 32, 2, true, true>(a, s);       // 256x256, BK64, 2 stages, pipelined fragments
    case 3: return launch_shape<EPI, 2, 2, 4, 4, 64, 2, 64, 2, false, true>(a, s);
    case 2: return launch_shape<EPI, 2, 4, 8, 4, 32, 4, 32, 2, true>(a, s);
    case 1: return launch_shape<EPI, 4, 2, 4, 4, 32, 3, 32, 4, false, true>(a, s);
    default: return launch_shape<EPI, 2, 2, 4, 4, 32, 3, 32, 3, true>(a, s);
  }
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
  a.tiles_m = 0; a.tiles_n = 0;
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
