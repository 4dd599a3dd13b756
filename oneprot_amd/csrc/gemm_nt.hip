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
#include <utility>
#include <stdlib.h>
#include "gemm_epi.h"
#include "sched_ws.h"

#define EPI_LD 68                                   // fp32 row pitch of a wave's epilogue tile (64 columns + pad)

static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0, 0, 0, 0};


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


static int g_sup_m = 4, g_sup_n = 10;      // L2 super-tile (tuning hook oneprot_gemm_tune)
static int g_nt_store = 0;                 // output store policy (A/B hook: sup_m = 256 * (1 + policy) + sup_m)
extern "C" void oneprot_gemm_tune(int sup_m, int sup_n) {
  if (sup_m >= 256) { g_nt_store = (sup_m >> 8) - 1; sup_m &= 255; }
  if (sup_m > 0) g_sup_m = sup_m;
  if (sup_n > 0) g_sup_n = sup_n;
}

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
      gst(reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o), w, p.nt_store);
    } else if (EPI == ONEPROT_EPI_F32) {
      float* c = (float*)p.out0 + o;
      gst(c, v[0], v[1], v[2], v[3], p.nt_store);
      gst(c + 4, v[4], v[5], v[6], v[7], p.nt_store);
    } else if (EPI == ONEPROT_EPI_BIAS_GELU) {
      if (p.out1) {          // gelu'(z) codes, consumed by the GELU_BWD epilogue of the dgrad GEMM
        gst(reinterpret_cast<u32x2*>((unsigned char*)p.out1 + o), gelu_fwd_and_code8(v), p.nt_store);
      } else gelu_fwd_only8(v);
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      gst(reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o), w, p.nt_store);
    } else if (EPI == ONEPROT_EPI_BIAS_RESID) {
      const float* rs = (const float*)p.aux + o;
      const float4 r0 = *reinterpret_cast<const float4*>(rs), r1 = *reinterpret_cast<const float4*>(rs + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      float* c = (float*)p.out0 + o;
      gst(c, v[0], v[1], v[2], v[3], p.nt_store);
      gst(c + 4, v[4], v[5], v[6], v[7], p.nt_store);
      if (p.out1) {
        u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
        gst(reinterpret_cast<u32x4*>((bf16_t*)p.out1 + o), w, p.nt_store);
      }
    } else if (EPI == ONEPROT_EPI_GELU_BWD) {
      const u32x2 z = *reinterpret_cast<const u32x2*>((const unsigned char*)p.aux + o);     // aux = gelu'(z) codes saved by the forward epilogue (gemm_epi.h)
      gelu_grad_apply8(v, z.x, z.y);
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      gst(reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o), w, p.nt_store);
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
      gst(reinterpret_cast<u32x4*>(dst + rp_off), w, p.nt_store);
    }
  }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads of this pass done before the next pass overwrites the tile
  }
  }
}


template <int EPI, int WM, int WN, int MT, int NTW, int BKT, int NSTAGE, int EPH, int MINW, bool PIPE, bool FULLT, bool DIRECT = false>
__global__ void __launch_bounds__(WM * WN * 64, MINW) k_gemm_nt(const GemmArgs p) {
  static_assert(!DIRECT || FULLT, "the direct-store form is built for full tiles only");
  static_assert(!DIRECT || NTW % 2 == 0, "pair map needs an even number of column tiles per wave");
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
  const int SUP_M = p.sup_m, SUP_N = p.sup_n;
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
  for (int i = 0; i < S::B_IPW; ++i) {
    const int slot = (wave * S::B_IPW + i) * S::RPI + srow;          // LDS row slot of the weight tile this lane fills
    int grow = slot;                                                 // global weight row (relative to the tile) that goes there
    if constexpr (DIRECT) {
      const int blk = slot / (NTW * 16), in = slot - blk * (NTW * 16);
      grow = blk * (NTW * 16) + direct_nmap<DirectMap<EPI>::PAIR>(in >> 4, in & 15);
    }
    b_rel[i] = (unsigned)grow * (unsigned)p.ldb * 2u + (unsigned)b_chunk[i] * 16u;
  }
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
  // residual tile of the lane: held in registers across the prologue when the accumulator leaves room for it (<= 16 tiles), else added up front
  constexpr bool RESID_ACC = DIRECT && EPI == ONEPROT_EPI_BIAS_RESID, RESID_DEFER = RESID_ACC && MT * NTW <= 16;
  float4 resid[RESID_DEFER ? MT : 1][RESID_DEFER ? NTW : 1];
  if constexpr (RESID_DEFER) direct_resid_load<EPI, MT, NTW>(p, resid, m0, n0, wr, wc, lane);
  f32x4 acc[MT][NTW];
  if constexpr (DIRECT) {          // lane (c, q) holds 4 consecutive columns of its token row in every tile: bias (and residual) are float4 per tile
    direct_init_acc<EPI, MT, NTW>(p, acc, m0, n0, wr, wc, lane);
    if constexpr (RESID_ACC && !RESID_DEFER) {   // one tile row at a time
      const int c = lane & 15, q = lane >> 4;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float* rrow = (const float*)p.aux + (size_t)(m0 + wr * (MT * 16) + i * 16 + c) * p.N + n0 + wc * (NTW * 16) + q * 4;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const float4 t = *reinterpret_cast<const float4*>(rrow + j * 16);
          acc[i][j][0] += t.x; acc[i][j][1] += t.y; acc[i][j][2] += t.z; acc[i][j][3] += t.w;
        }
      }
    }
  } else {
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int gc = n0 + wc * (NTW * 16) + j * 16 + (lane & 15);
    const float bj = (p.bias && gc < p.N) ? p.bias[gc] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][j] = (f32x4){bj, bj, bj, bj};
  }
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
    if constexpr (RESID_DEFER) direct_resid_add<EPI, MT, NTW>(acc, resid);
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
          for (int j = 0; j < NTW; ++j)
            acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
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
    if constexpr (RESID_DEFER) direct_resid_add<EPI, MT, NTW>(acc, resid);
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
          for (int j = 0; j < NTW; ++j)
            acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
      nbuf = (nbuf + 1 == NSTAGE) ? 0 : nbuf + 1;
    }
  }
  if constexpr (DIRECT) {
    gemm_epilogue_direct<EPI, MT, NTW>(p, acc, m0, n0, wr, wc, lane);      // straight from the accumulators: the LDS ring is not touched again
  } else {
    __syncthreads();          // everyone done with the staging ring before it is reused as epilogue tiles
    // ---- epilogue (staging tiles reuse the ring memory)
    gemm_epilogue<EPI, MT, NTW, EPH, FULLT>(p, acc, reinterpret_cast<float*>(smem) + wave * EPH * EPI_LD, m0, n0, wr, wc, lane);
  }
}

template <int EPI, int WM, int WN, int MT, int NTW, int BKT, int NSTAGE, int EPH, int MINW, bool PIPE, bool FULLT, bool DIRECT = false>
static int launch_shape_full(GemmArgs a, hipStream_t s) {
  typedef Shape<WM, WN, MT, NTW, BKT, NSTAGE, EPH> S;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k_gemm_nt<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, FULLT, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS) != hipSuccess)
      return OP_ELAUNCH;
    configured = true;
  }
  a.tiles_m = (a.M + S::BM_ - 1) / S::BM_;
  a.tiles_n = (a.N + S::BN_ - 1) / S::BN_;
  const int pm_total = (a.tiles_m + 7) / 8;
  a.sup_m = g_sup_m; a.sup_n = g_sup_n;
  const int grid = ((pm_total + a.sup_m - 1) / a.sup_m) * (a.sup_m * a.sup_n * ((a.tiles_n + a.sup_n - 1) / a.sup_n)) * 8;       // super-tile slots; surplus blocks exit at once
  hipLaunchKernelGGL((k_gemm_nt<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, FULLT, DIRECT>), dim3(grid), dim3(S::NW * 64), S::LDS, s, a);
  return launch_status();
}

// FULL-tile specialisation only for the shapes the heuristic picks (keeps the number of kernel instantiations down)
template <int EPI, int WM, int WN, int MT, int NTW, int BKT, int NSTAGE, int EPH, int MINW, bool PIPE, bool TRY_FULL = false, bool TRY_DIRECT = false>
static int launch_shape(GemmArgs a, hipStream_t s) {
  typedef Shape<WM, WN, MT, NTW, BKT, NSTAGE, EPH> S;
  if constexpr (TRY_FULL) {
    const bool full = a.M % S::BM_ == 0 && a.N % S::BN_ == 0 && a.K % BKT == 0 && (size_t)S::BM_ * a.lda * 2 < (1ull << 31) && (size_t)S::BN_ * a.ldb * 2 < (1ull << 31);
    if constexpr (TRY_DIRECT) {
      // direct-store form: additionally every row of the tile lies in one sequence-aligned 16-row group for QKV/RoPE (L % 16 == 0) and the head
      // dim has the RoPE partner in the same lane (32 / 64); anything else takes the staged epilogue
      const bool rope_ok = EPI != ONEPROT_EPI_QKV_ROPE || (a.hd == 32 || a.hd == 64);
      if (full && rope_ok) return launch_shape_full<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, true, true>(a, s);
    }
    if (full) return launch_shape_full<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, true>(a, s);
  }
  return launch_shape_full<EPI, WM, WN, MT, NTW, BKT, NSTAGE, EPH, MINW, PIPE, false>(a, s);
}

// =====================================================================================================================================
// Register-staged form (whole 256 x 256 tiles, direct-store epilogue).  Every NT-GEMM form above is bound by the same thing (round-2 probes,
// DESIGN.md section 6): a K-slice can only be requested once an LDS buffer is free, 160 KB of LDS hold at most two 64 KB slices of a 256 x 256
// tile, so ONE slice is in flight and a K-step costs max(MFMA time, fill latency + transfer) = ~3200 cycles against 2304 of MFMA work.  Here the
// slices travel through REGISTERS instead: each wave loads its 8 KB share of a slice with eight plain global_load_dwordx4 (32 VGPRs), keeps TWO
// slices in flight that way (64 VGPRs: a load now has two K-steps to arrive, 128 KB in flight per CU on top of the 128 KB resident in LDS), and
// commits a slice to its LDS buffer (ds_write_b128, lane-linear = the same image the LDS-DMA form builds) one K-step before it is consumed.
// The loads are inline asm with hand-counted vmcnt waits: hipcc's waitcnt pass loses the count of the OTHER register set's loads across the loop
// back edge and drains the queue (vmcnt 7..0) before every commit, which puts one slice in flight again.  rs_wait<N>() is also what tells
// the compiler that a register set is (re)defined at that point and not at the load.
__device__ __forceinline__ void rs_load16(u32x4& d, const unsigned char* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void rs_wait(u32x4 (&r)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N) : "memory");
}
template <int EPI>
__global__ void __launch_bounds__(512, 2) k_gemm_nt_rs(const GemmArgs p) {
  constexpr int WM = 2, WN = 4, MT = 8, NTW = 4, BKT = 64, KK = 2;
  typedef Shape<WM, WN, MT, NTW, BKT, 2, 32> S;        // 256 x 256 tile, 128-byte LDS rows, 64 KB per stage, 2 buffers
  static_assert(S::A_IPW == 4 && S::B_IPW == 4, "eight 1 KB pieces per wave and stage");
  constexpr bool PAIR = DirectMap<EPI>::PAIR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- XCD-aware, L2-blocked tile assignment (as k_gemm_nt)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, seq = bid >> 3;
  const int SUP_M = p.sup_m, SUP_N = p.sup_n;
  const int pm_total = (p.tiles_m + 7) >> 3;
  const int per_mgroup = SUP_M * SUP_N * ((p.tiles_n + SUP_N - 1) / SUP_N);
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int nk = p.K / BKT;

  // ---- source offsets.  Piece = 8 rows x 128 B; swizzled source chunk = chunk ^ ((row >> 1) & 7) = (chunk ^ (srow >> 1)) ^ ((piece & 1) << 2)
  const int srow = lane >> 3, schunk = lane & 7;
  const unsigned a_rel_e = (unsigned)srow * (unsigned)p.lda * 2u + (unsigned)(schunk ^ (srow >> 1)) * 16u;      // even pieces; odd: ^ 64
  unsigned b_rel[S::B_IPW];
#pragma unroll
  for (int i = 0; i < S::B_IPW; ++i) {
    const int slot = (wave * S::B_IPW + i) * S::RPI + srow;
    const int blk = slot / (NTW * 16), in = slot - blk * (NTW * 16);
    const int grow = blk * (NTW * 16) + direct_nmap<PAIR>(in >> 4, in & 15);
    b_rel[i] = (unsigned)grow * (unsigned)p.ldb * 2u + (unsigned)swz<BKT>(slot, schunk) * 16u;
  }
  const size_t a_piece = (size_t)S::RPI * p.lda * 2;
  const unsigned char* a_tile = reinterpret_cast<const unsigned char*>(p.A + (size_t)m0 * p.lda) + (size_t)(wave * S::A_IPW) * a_piece;
  const unsigned char* b_tile = reinterpret_cast<const unsigned char*>(p.B + (size_t)n0 * p.ldb);
  const int lane16 = lane * 16;

  u32x4 st[2][8];                                   // two K-slices in flight in registers
  auto gload = [&](int t, auto setc) {
    constexpr int set = decltype(setc)::value;
    const unsigned char* ak = a_tile + (size_t)t * (BKT * 2);
    const unsigned char* bk = b_tile + (size_t)t * (BKT * 2);
#pragma unroll
    for (int i = 0; i < S::A_IPW; ++i) st[set][i] = *reinterpret_cast<const u32x4*>(ak + i * a_piece + (a_rel_e ^ ((i & 1) << 6)));
#pragma unroll
    for (int i = 0; i < S::B_IPW; ++i) st[set][S::A_IPW + i] = *reinterpret_cast<const u32x4*>(bk + b_rel[i]);
  };
  auto commit = [&](int buf, auto setc) {
    constexpr int set = decltype(setc)::value;
    unsigned char* sA = smem + buf * S::STAGE + lane16;
    unsigned char* sB = sA + S::BM_ * S::ROWB;
#pragma unroll
    for (int i = 0; i < S::A_IPW; ++i) *reinterpret_cast<u32x4*>(sA + (wave * S::A_IPW + i) * 1024) = st[set][i];
#pragma unroll
    for (int i = 0; i < S::B_IPW; ++i) *reinterpret_cast<u32x4*>(sB + (wave * S::B_IPW + i) * 1024) = st[set][S::A_IPW + i];
  };

  // accumulators start at the bias of their columns (+ the residual tile)
  f32x4 acc[MT][NTW];
  direct_init_acc<EPI, MT, NTW>(p, acc, m0, n0, wr, wc, lane);
  if constexpr (EPI == ONEPROT_EPI_BIAS_RESID) {      // (no spare registers for a deferred add next to two staging sets: the residual is added up front)
    float4 resid[MT][NTW];
    direct_resid_load<EPI, MT, NTW>(p, resid, m0, n0, wr, wc, lane);
    direct_resid_add<EPI, MT, NTW>(acc, resid);
  }
  const int frow = lane & 15, fq = lane >> 4;
  const int a_off0 = (wr * (MT * 16) + frow) * S::ROWB + (swz<BKT>(frow, fq) << 4);
  const int b_off0 = S::BM_ * S::ROWB + (wc * (NTW * 16) + frow) * S::ROWB + (swz<BKT>(frow, fq) << 4);

  // ---- prologue: slices 0 and 1 requested, slice 0 committed, slice 2 requested into the freed registers
  gload(0, std::integral_constant<int, 0>{});
  if (nk > 1) gload(1, std::integral_constant<int, 1>{});
  commit(0, std::integral_constant<int, 0>{});
  if (nk > 2) gload(2, std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");

  // STEADY: slices t+1 and t+3 exist, nothing is conditional -- hipcc's waitcnt pass then counts the eight younger loads of the other register
  // set exactly (vmcnt(8) before the commit); with the guards inside the loop it merges the paths and drains the queue (vmcnt 7..0), which puts
  // one slice in flight again.  The guarded form only runs the last K-steps of a tile.
  auto kstep = [&](int t, auto parity, auto steadyc) {
    constexpr int P = decltype(parity)::value;      // slice t+1 sits in register set P ^ 1 (slice s uses set s & 1)
    constexpr bool STEADY = decltype(steadyc)::value;
    const unsigned char* base = smem + (t & 1) * S::STAGE;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const unsigned char* sa = base + (a_off0 ^ (kk << 6));
      const unsigned char* sb = base + (b_off0 ^ (kk << 6));
      bf8_t b[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) b[j] = *reinterpret_cast<const bf8_t*>(sb + j * (16 * S::ROWB));
#pragma unroll
      for (int h = 0; h < 2; ++h) {               // activation fragments in two batches of four (register budget)
        bf8_t a[MT / 2];
#pragma unroll
        for (int i = 0; i < MT / 2; ++i) a[i] = *reinterpret_cast<const bf8_t*>(sa + (h * (MT / 2) + i) * (16 * S::ROWB));
#pragma unroll
        for (int i = 0; i < MT / 2; ++i)
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[h * (MT / 2) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[h * (MT / 2) + i][j], 0, 0, 0);
      }
      if (kk == 0) {
        // slice t+1 (requested two K-steps ago) goes into the buffer that slice t-1 vacated at the last barrier; its registers then take slice t+3
        if (STEADY || t + 1 < nk) commit((t + 1) & 1, std::integral_constant<int, P ^ 1>{});
        if (STEADY || t + 3 < nk) gload(t + 3, std::integral_constant<int, P ^ 1>{});
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  };
  int t = 0;
  for (; t + 4 < nk; t += 2) {
    kstep(t, std::integral_constant<int, 0>{}, std::true_type{});
    kstep(t + 1, std::integral_constant<int, 1>{}, std::true_type{});
  }
  for (; t < nk; t += 2) {
    kstep(t, std::integral_constant<int, 0>{}, std::false_type{});
    if (t + 1 < nk) kstep(t + 1, std::integral_constant<int, 1>{}, std::false_type{});
  }
  gemm_epilogue_direct<EPI, MT, NTW>(p, acc, m0, n0, wr, wc, lane);
}

template <int EPI>
static int launch_rs(GemmArgs a, hipStream_t s) {
  constexpr int LDS = 2 * 512 * 128;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k_gemm_nt_rs<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return OP_ELAUNCH;
    configured = true;
  }
  a.tiles_m = a.M / 256; a.tiles_n = a.N / 256;
  a.sup_m = g_sup_m; a.sup_n = g_sup_n;
  const int pm_total = (a.tiles_m + 7) / 8;
  const int grid = ((pm_total + a.sup_m - 1) / a.sup_m) * (a.sup_m * a.sup_n * ((a.tiles_n + a.sup_n - 1) / a.sup_n)) * 8;
  hipLaunchKernelGGL((k_gemm_nt_rs<EPI>), dim3(grid), dim3(512), LDS, s, a);
  return launch_status();
}
template <int EPI>
static bool rs_eligible(const GemmArgs& a) {
  if (a.M % 256 || a.N % 256 || a.K % 64 || a.lda % 64) return false;
  if ((size_t)256 * a.lda * 2 >= (1ull << 31) || (size_t)256 * a.ldb * 2 >= (1ull << 31)) return false;
  if (EPI == ONEPROT_EPI_QKV_ROPE && a.hd != 32 && a.hd != 64) return false;
  return true;
}

// =====================================================================================================================================
// Ping-pong form: ONE 512-thread work-group per CU, persistent over tiles, made of two 4-wave groups (one wave of each per SIMD) that run in
// ANTI-PHASE on different 256 x 128 output tiles: while group g issues the MFMAs of its tile's K loop, group g^1 runs the epilogue of the tile
// it has just finished (GELU / RoPE arithmetic, the fp32 residual read-modify-write, the stores) and then stages the first K-slices of its next
// tile.  Measured on gfx950 (tools/ab/coissue2.hip, profiles/r02_coissue_probe.txt): an MFMA-only wave and a GELU-arithmetic wave on the same SIMD
// overlap almost completely (MFMA stream unimpeded, the VALU stream at 88 % of its stand-alone rate), whereas two waves that are both in their
// epilogue gain nothing from each other (the VALU pipe is already full) -- so with every wave of a work-group in the same phase the epilogue time
// simply ADDS to the main loop (the round-1 finding), and de-phasing independent work-groups cannot be enforced.  Here the phase relation is
// enforced by construction: the work-group barrier is the common clock, every "slot" (one 32-deep K-step of the main-loop group) contains exactly
// one s_barrier in both roles, a main-loop phase is NK = K/32 slots, and the epilogue role spreads its chunks (8 output elements per lane each)
// over the first NK-3 slots of the partner's main loop and issues its own next tile's first three K-slices in the remaining ones.
//   * per group: wave tile 128 x 64 (8 x 4 accumulator tiles), private 3-buffer LDS ring of 24 KB stages (2 x 72 KB per work-group), LDS-DMA
//     fills with counted vmcnt; the operand fragments of K-step t+1 are read while the MFMAs of step t issue (two register sets), so a stage's
//     buffer is free one slot earlier and 3 buffers keep two K-slices in flight;
//   * direct-store epilogue (operand roles swapped, weight rows permuted at staging: see gemm_epilogue_direct);
//   * tiles are dealt per XCD (work-groups b, b+8, ... share an XCD: they walk that XCD's row panels, n fastest), two per work-group per round.
// Built for whole tiles with K % 64 == 0 and K >= 128; everything else takes the per-tile kernels above.
// One epilogue chunk of the ping-pong kernel.  Addresses are formed as (wave-uniform row/column base, on the scalar unit) + ONE 32-bit per-lane
// byte offset (row-in-tile * N + column-in-quad): per-lane 64-bit addresses per chunk would be hoisted by the compiler and pin ~50 VGPRs for the
// whole kernel (the register file is the scarce resource here: 128 accumulators + 64 fragment registers).
template <int EPI, int C, int MT, int NTW>
__device__ __forceinline__ void pp_chunk(const GemmArgs& p, f32x4 (&acc)[MT][NTW], int um0, int un0, unsigned lane_off, const u32x4& aux16, const float4& res4) {
  static_assert(C >= 0 && C < (DirectMap<EPI>::PAIR ? MT * NTW / 2 : MT * NTW), "chunk index");
  if constexpr (DirectMap<EPI>::PAIR) {
    constexpr int i = C / (NTW / 2), pp = C % (NTW / 2);
    const size_t uo = (size_t)(um0 + i * 16) * p.N + un0 + pp * 32;          // uniform element offset of the chunk's row group / column block
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * pp][r]; v[4 + r] = acc[i][2 * pp + 1][r]; }
    if (EPI == ONEPROT_EPI_BIAS_GELU) {
      if (p.out1 != nullptr) {
        *reinterpret_cast<u32x2*>((unsigned char*)p.out1 + uo + (lane_off >> 1)) = gelu_fwd_and_code8(v);      // one byte per element: half the bf16 lane offset
      } else {
        gelu_fwd_only8(v);
      }
    } else if (EPI == ONEPROT_EPI_GELU_BWD) {
      gelu_grad_apply8(v, aux16.x, aux16.y);
    }
    u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>((bf16_t*)p.out0 + uo) + lane_off) = w;
  } else {
    constexpr int i = C / NTW, j = C % NTW;
    const size_t uo = (size_t)(um0 + i * 16) * p.N + un0 + j * 16;
    float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    if (EPI == ONEPROT_EPI_BIAS_RESID) { v.x += res4.x; v.y += res4.y; v.z += res4.z; v.w += res4.w; }
    *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>((float*)p.out0 + uo) + lane_off) = v;
    if (EPI == ONEPROT_EPI_BIAS_RESID && p.out1) {
      u32x2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
      *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>((bf16_t*)p.out1 + uo) + (lane_off >> 1)) = w;
    }
  }
}

// one (i, j) tile of the QKV / RoPE epilogue (natural map; see rope_store_direct)
template <int HD, int C, int MT, int NTW>
__device__ __forceinline__ void pp_chunk_rope(const GemmArgs& p, f32x4 (&acc)[MT][NTW], int ncol0, int q, int b, int l) {
  constexpr int i = C / NTW, j = C % NTW;
  constexpr int HALF = HD / 2, JP = HALF / 16;
  constexpr int jt = (j * 16) % HD;
  constexpr bool lo = jt < HALF;
  const int dm = p.H * HD;
  const int sec = __builtin_amdgcn_readfirstlane(ncol0 / dm);
  const int head = (ncol0 - sec * dm) / HD + (j * 16) / HD;
  bf16_t* dst = (bf16_t*)(sec == 0 ? p.out0 : (sec == 1 ? p.out1 : p.out2));
  const float sc = sec == 0 ? p.q_scale : 1.0f;
  float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
  if (sec < 2) {
    const f32x4 pv = acc[i][lo ? j + JP : j - JP];
    constexpr int jj = lo ? jt : jt - HALF;
    const float4 cs = *reinterpret_cast<const float4*>(p.cos + (size_t)l * HALF + q * 4 + jj);
    const float4 sn = *reinterpret_cast<const float4*>(p.sin + (size_t)l * HALF + q * 4 + jj);
    const float sp = lo ? -sc : sc;
    v[0] = (v[0] * sc) * cs.x + (pv[0] * sp) * sn.x; v[1] = (v[1] * sc) * cs.y + (pv[1] * sp) * sn.y;
    v[2] = (v[2] * sc) * cs.z + (pv[2] * sp) * sn.z; v[3] = (v[3] * sc) * cs.w + (pv[3] * sp) * sn.w;
  }
  u32x2 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<u32x2*>(dst + (((size_t)b * p.H + head) * p.L + l) * HD + jt + q * 4) = w;
}

#define PP_BARRIER() asm volatile("s_barrier" ::: "memory")
template <class F, int... Cs> __device__ __forceinline__ void pp_unroll(F&& f, std::integer_sequence<int, Cs...>) { (f(std::integral_constant<int, Cs>{}), ...); }

// Roles per phase q (= one tile's K loop, NK slots of one 32-deep K-step each; every slot opens with the work-group barrier):
//   consumer = group q & 1, on tile q:   barrier -> ds_read the operand fragments of K-step t+1 -> 32 MFMAs of K-step t.  No vector-memory
//              instruction at all: an LDS-DMA piece costs its issuing wave 60-180 cycles in order, which a lone wave per SIMD cannot hide
//              (first version of this kernel: 1500 cycles per slot against 576 of MFMA work).
//   loader   = the other group = owner of tiles q-1 and q+1:  barrier -> one sixteenth of the epilogue of tile q-1 (slots 0..15) -> the six
//              LDS-DMA pieces of the stage four K-steps ahead of the consumer (crossing into tile q+1, its own next tile, at the end of the
//              phase) -> counted vmcnt that leaves exactly its two youngest stages in flight.
// ONE ring of five 24 KB stage buffers is shared: global stage G = q * NK + t lives in buffer G % 5; the loader writes stage G+4 into the buffer
// whose fragments the consumer fetched two slots earlier.  A stage is published by its issuing wave's vmcnt wait followed by the next slot's
// barrier; the four stages that the new consumer issued itself while it was still loader are covered by its own waits in slots 0 and 1.
// wait until at most `halves` of this wave's youngest half-stage issues (HP LDS-DMA pieces each) are still in flight
template <int HP> __device__ __forceinline__ void pp_wait_halves(int halves) {
  if (halves >= 2) wait_vmcnt<2 * HP>(); else if (halves == 1) wait_vmcnt<HP>(); else wait_vmcnt<0>();
}

// Geometry of the ping-pong kernel.  A STAGE is a 64-deep K-slice of the group tile (256 activation rows + 128 weight rows, 128-byte LDS rows:
// every 1 KB LDS-DMA piece covers 8 rows x one whole 128-byte line -- measured (tools/ab/fill_probe.hip, profiles/r02_fill_probe.txt) 93-100 GB/s
// per CU against 59 GB/s for pieces of 16 half lines (32-deep slices), which capped the first version of this kernel at 22 B/clk); the ring
// holds three stages (144 KB).  TIME is counted in SUB-SLOTS of one 32-deep K-step (= 32 MFMAs per consumer wave), each opened by the
// work-group barrier.  Sub-slot U = 2t + kk works on half kk of stage t.
//   consumer: MFMAs of (t, kk) from registers; fetches the fragments of the next sub-step -- (t, 1) from the same buffer, or (t+1, 0) from the
//             next one -- in two halves behind the MFMAs that free the registers (one activation set, two weight sets: 192 + 64 registers).
//   loader:   one sixteenth of its previous tile's epilogue (sub-slots 0..15), then HALF a stage (6 of the 12 pieces per wave): at U = 2t+1 the
//             first half of stage t+3 into the buffer of stage t (its last fragments were fetched during sub-slot 2t, and the consumer retires
//             its LDS reads before the barrier that opens 2t+1), at U = 2t+2 the second half.  I.e. half-stage h = U + 5 is issued in sub-slot
//             U; afterwards the wave waits until at most its two youngest half-stages are in flight, so that at the END of sub-slot U every
//             half-stage <= U + 3 has landed -- the consumer touches stage t+1 = halves 2t+2, 2t+3 first in sub-slot 2t+1.
//   role change: the new consumer issued half-stages up to U'+4 itself (U' = its first sub-slot); U'+3 must have landed at the end of U'
//             (vmcnt(6)), U'+4 at the end of U'+1 (vmcnt(0)); afterwards its queue is empty.
template <int EPI, bool GRAD, int NBX>
__global__ void __launch_bounds__(512, 2) k_gemm_nt_pp(const GemmArgs p, int g8) {
  constexpr int MT = 8, NTW = 4, BKT = 64, NB = 3, CSLOTS = 16, HLA = 5;
  typedef Shape<2, 2, MT, NTW, BKT, NB, 32> S;          // per GROUP: 4 waves, 256 x 128 tile, 128-byte LDS rows, 48 KB per stage
  constexpr bool PAIR = DirectMap<EPI>::PAIR;
  constexpr int NCH = PAIR ? MT * NTW / 2 : MT * NTW;   // epilogue chunks per wave tile (8 resp. 4 elements per lane each)
  constexpr int CPS = NCH / CSLOTS;                     // chunks per chunk-carrying sub-slot
  constexpr int HP = S::LPS / 2;                        // LDS-DMA pieces per wave per half-stage (6)
  static_assert(S::A_IPW == 8 && S::B_IPW == 4 && HP == 6, "half-stage split below assumes 8 + 4 pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2, gw = wave & 3;
  const int wr = gw >> 1, wc = gw & 1;
  const int NK = p.K / BKT;                             // stages per tile
  const int NU = 2 * NK;                                // sub-slots per phase

  // ---- this work-group's tile sequence (per XCD: row panels x, x+8, ...; n fastest)
  const int x = blockIdx.x & 7, w = blockIdx.x >> 3;
  const int panels_x = p.tiles_m > x ? (p.tiles_m - x + 7) >> 3 : 0;
  const int Tx = panels_x * p.tiles_n;
  const int Q = Tx > w ? (Tx - w + g8 - 1) / g8 : 0;
  if (Q == 0) return;
  auto tile_origin = [&](int qi, int& m0, int& n0) {
    const int u = qi * g8 + w;
    const int pl = u / p.tiles_n, tn = u - pl * p.tiles_n;
    m0 = (pl * 8 + x) * S::BM_; n0 = tn * S::BN_;
  };

  // ---- staging constants.  Piece = 8 rows x 128 B, lane (srow, schunk); the source chunk carries the XOR swizzle chunk ^ ((row >> 1) & 7), which
  // for row = 8 * piece + srow is ((piece & 1) * 4 + (srow >> 1)): even and odd pieces differ by chunk bit 2 = byte offset bit 6.
  // (These per-lane constants are only needed by the loader role and the fragment offsets below only by the consumer role: with 192 registers
  // in accumulators and fragments there is no room to keep both sets alive, so each role recomputes its own from the lane id at phase entry --
  // `role_lane()` hides the lane id behind an empty asm so that the compiler does not hoist the results out of the phase loop and spill them.)
  auto role_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  unsigned a_rel_e = 0, b_rel[S::B_IPW] = {0, 0, 0, 0};
  auto loader_consts = [&]() {
    const int l = role_lane();
    const int srow = l >> 3, schunk = l & 7;
    a_rel_e = (unsigned)srow * (unsigned)p.lda * 2u + (unsigned)(schunk ^ (srow >> 1)) * 16u;      // even pieces; odd: ^ 64
#pragma unroll
    for (int i = 0; i < S::B_IPW; ++i) {
      const int slot = (gw * S::B_IPW + i) * S::RPI + srow;
      const int blk = slot / (NTW * 16), in = slot - blk * (NTW * 16);
      const int grow = blk * (NTW * 16) + direct_nmap<PAIR>(in >> 4, in & 15);
      b_rel[i] = (unsigned)grow * (unsigned)p.ldb * 2u + (unsigned)swz<BKT>(slot, schunk) * 16u;
    }
  };
  const size_t a_piece = (size_t)S::RPI * p.lda * 2;          // bytes between consecutive A pieces (wave-uniform)
  // half 0: A pieces 0..5 of this wave; half 1: A pieces 6, 7 and the four W pieces
  auto issue_half = [&](const unsigned char* a_tile, const unsigned char* b_tile, int t, int half, int buf) {
#if defined(PP_ABL_NODMA)
    return;
#endif
    unsigned char* sA = smem + buf * S::STAGE;
    unsigned char* sB = sA + S::BM_ * S::ROWB;
    const unsigned char* ak = a_tile + (size_t)t * (BKT * 2) + (size_t)(gw * S::A_IPW) * a_piece;
    const unsigned char* bk = b_tile + (size_t)t * (BKT * 2);
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(ak + i * a_piece + (a_rel_e ^ ((i & 1) << 6))), LDS_PTR(sA + (gw * S::A_IPW + i) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 6; i < 8; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(ak + i * a_piece + (a_rel_e ^ ((i & 1) << 6))), LDS_PTR(sA + (gw * S::A_IPW + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < S::B_IPW; ++i) __builtin_amdgcn_global_load_lds(GLB_PTR(bk + b_rel[i]), LDS_PTR(sB + (gw * S::B_IPW + i) * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets inside a stage image for k-half 0 (k-half 1: ^ 64); tile i / j is a constant 2 KB step = an immediate offset
  const int frow = lane & 15, fq = lane >> 4;
  int a_off0 = 0, b_off0 = 0;
  auto consumer_consts = [&]() {
    const int l = role_lane();
    const int fr = l & 15, fqq = l >> 4;
    a_off0 = (wr * (MT * 16) + fr) * S::ROWB + (swz<BKT>(fr, fqq) << 4);
    b_off0 = S::BM_ * S::ROWB + (wc * (NTW * 16) + fr) * S::ROWB + (swz<BKT>(fr, fqq) << 4);
  };

  // Register plan (256 per lane at two waves per SIMD): 128 accumulators + ONE set of activation fragments (32) + two sets of weight fragments
  // (32) = 192.  The activation fragments of the next sub-step are fetched in two halves, each right after the MFMAs that consumed the registers
  // it overwrites, so the reads still run half a sub-step ahead of their use.
  f32x4 acc[MT][NTW];
  bf8_t fa[MT], fb[2][NTW];
  auto load_fa = [&](auto halfc, int buf, int kk) {
    constexpr int half = decltype(halfc)::value;
#if defined(PP_ABL_NOREAD)
    return;
#endif
    const unsigned char* sa = smem + buf * S::STAGE + (a_off0 ^ (kk << 6));
#pragma unroll
    for (int i = half * (MT / 2); i < (half + 1) * (MT / 2); ++i) fa[i] = *reinterpret_cast<const bf8_t*>(sa + i * (16 * S::ROWB));
  };
  auto load_fb = [&](auto setc, int buf, int kk) {
    constexpr int set = decltype(setc)::value;
#if defined(PP_ABL_NOREAD)
    return;
#endif
    const unsigned char* sb = smem + buf * S::STAGE + (b_off0 ^ (kk << 6));
#pragma unroll
    for (int j = 0; j < NTW; ++j) fb[set][j] = *reinterpret_cast<const bf8_t*>(sb + j * (16 * S::ROWB));
  };
  auto init_acc = [&](int n0) {          // accumulators start at the bias of their columns (4 consecutive columns per lane and tile)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      f32x4 bj = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + n0 + wc * (NTW * 16) + direct_nmap<PAIR>(j, fq * 4)); bj = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = bj;
    }
  };

  // ---- cursor of the next half-stage to issue (the same in both groups, whoever is loader), and the consumer's stage buffer
  int iq = 0, it = 0, ih = 0, ibuf = 0;         // tile / stage / half / ring buffer of the next half-stage to issue
  const unsigned char *ia_tile, *ib_tile;
  { int m0, n0; tile_origin(0, m0, n0); ia_tile = reinterpret_cast<const unsigned char*>(p.A + (size_t)m0 * p.lda); ib_tile = reinterpret_cast<const unsigned char*>(p.B + (size_t)n0 * p.ldb); }
  auto advance_issue = [&]() {
    if (ih == 0) { ih = 1; return; }
    ih = 0;
    ibuf = ibuf + 1 == NB ? 0 : ibuf + 1;
    if (++it == NK) {
      it = 0; ++iq;
      if (iq < Q) { int m0, n0; tile_origin(iq, m0, n0); ia_tile = reinterpret_cast<const unsigned char*>(p.A + (size_t)m0 * p.lda); ib_tile = reinterpret_cast<const unsigned char*>(p.B + (size_t)n0 * p.ldb); }
    }
  };
  int rbuf = 0;                                 // ring buffer of the stage whose MFMAs run in the current sub-slot
  // per-lane byte offset of this lane's first output element inside a chunk's (row group, column block): see pp_chunk
  const unsigned lane_off = PAIR ? ((unsigned)frow * (unsigned)p.N + fq * 8u) * 2u : ((unsigned)frow * (unsigned)p.N + fq * 4u) * 4u;

  // ---- start-up: the first consumer (group 0) issues half-stages 0..4 of tile 0 itself; 0..2 must have landed when sub-slot 0 opens
  if (group == 0) {
    int m0, n0; tile_origin(0, m0, n0);
    init_acc(n0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    loader_consts();
  }
#pragma unroll
  for (int k = 0; k < HLA; ++k) { if (group == 0) issue_half(ia_tile, ib_tile, it, ih, ibuf); advance_issue(); }
  if (group == 0) pp_wait_halves<HP>(2);

#if defined(PP_ABL_NOMFMA)
#define PP_MFMA(d, a, b) asm volatile("" :: "v"(a), "v"(b))
#else
#define PP_MFMA(d, a, b) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0)
#endif
  for (int ph = 0; ph < Q; ++ph) {
    if ((ph & 1) == group) {
      // ================= consumer of tile ph
      // sub(t, kk, P): P = weight-fragment set in use (alternates every sub-slot)
      auto sub = [&](int t, auto kkc, auto parity) {
        constexpr int kk = decltype(kkc)::value, P = decltype(parity)::value;
        if (!(t == 0 && kk == 0)) PP_BARRIER();   // (the first sub-slot's barrier and fragment reads sit in front of the loop)
        const int nxt = rbuf + 1 == NB ? 0 : rbuf + 1;
        const bool more = kk == 0 || t + 1 < NK;  // is there a next sub-step in this tile?
        const int nbuf = kk == 0 ? rbuf : nxt;    // ... it lives in the same stage (k-half 1) or in the next one (k-half 0)
#pragma unroll
        for (int i = 0; i < MT / 2; ++i)
#pragma unroll
          for (int j = 0; j < NTW; ++j) PP_MFMA(acc[i][j], fb[P][j], fa[i]);
        if (more) { load_fb(std::integral_constant<int, P ^ 1>{}, nbuf, kk ^ 1); load_fa(std::integral_constant<int, 0>{}, nbuf, kk ^ 1); }
#pragma unroll
        for (int i = MT / 2; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTW; ++j) PP_MFMA(acc[i][j], fb[P][j], fa[i]);
        if (more) load_fa(std::integral_constant<int, 1>{}, nbuf, kk ^ 1);
        if (kk == 1) rbuf = nxt;
        advance_issue();                           // (cursor only: the loader issues)
        // kk == 0: the reads just issued are the LAST ones of this stage, whose buffer the loader refills right after the next barrier
        if (kk == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // half-stages this group issued while it was loader: U'+3 lands by the end of its first sub-slot, U'+4 by the end of the second
        if (t == 0 && kk == 0) wait_vmcnt<HP>(); else wait_vmcnt<0>();
      };
      consumer_consts();
      PP_BARRIER();
      load_fb(std::integral_constant<int, 0>{}, rbuf, 0);
      load_fa(std::integral_constant<int, 0>{}, rbuf, 0);
      load_fa(std::integral_constant<int, 1>{}, rbuf, 0);
      for (int t = 0; t < NK; ++t) {
        sub(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        sub(t, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      }
    } else {
      // ================= loader for tile ph's K loop + epilogue of this group's previous tile (ph-1) + first stages of its next tile (ph+1)
      const bool has_prev = ph >= 1;
      loader_consts();
      int em0 = 0, en0 = 0;
      if (has_prev) tile_origin(ph - 1, em0, en0);
      const int um0 = em0 + wr * (MT * 16), un0 = en0 + wc * (NTW * 16);
      const u32x4 no16 = {0, 0, 0, 0}; const float4 no4 = make_float4(0.f, 0.f, 0.f, 0.f);
      // end of a loader sub-slot: issue the half-stage HLA sub-steps ahead of the consumer, then wait until at most the two youngest half-stages
      // are in flight (this sub-slot's stores are older than its DMA pieces, so they never count among the youngest).  When the sequence has
      // run out, fewer may stay.
      int since_issue = 0;
      auto loader_tail = [&](int u) {
        if (iq < Q) { issue_half(ia_tile, ib_tile, it, ih, ibuf); since_issue = 0; } else ++since_issue;
        advance_issue();
        if (u & 1) rbuf = rbuf + 1 == NB ? 0 : rbuf + 1;      // the consumer's stage cursor advances in BOTH groups
        pp_wait_halves<HP>(2 - since_issue);
      };
      auto cslot = [&](auto sc) {                  // sub-slots 0..15: CPS epilogue chunks each
        constexpr int sidx = decltype(sc)::value;
        PP_BARRIER();
        if (has_prev) {
          pp_unroll([&](auto cc) { pp_chunk<EPI, sidx * CPS + decltype(cc)::value, MT, NTW>(p, acc, um0, un0, lane_off, no16, no4); }, std::make_integer_sequence<int, CPS>{});
        }
        if (sidx == CSLOTS - 1) {                  // accumulators are free: the next tile's bias goes in now (an ordinary load: drains the queue once per tile).
          // Unconditional on purpose (a tile that does not exist re-reads the bias of the last one): a conditional re-definition keeps the
          // old accumulator values alive beside the new ones and costs 50 VGPRs of spills
          int m0, n0; tile_origin(ph + 1 < Q ? ph + 1 : Q - 1, m0, n0);
          init_acc(n0);
        }
        loader_tail(sidx);
      };
      pp_unroll(cslot, std::make_integer_sequence<int, CSLOTS>{});
      for (int u = CSLOTS; u < NU; ++u) {
        PP_BARRIER();
        loader_tail(u);
      }
    }
  }
  // ================= closing phase: the owner of the last tile finishes its epilogue (no partner K loop left: no barriers)
  if (((Q - 1) & 1) == group) {
    int em0, en0; tile_origin(Q - 1, em0, en0);
    const int um0 = em0 + wr * (MT * 16), un0 = en0 + wc * (NTW * 16);
    const u32x4 no16 = {0, 0, 0, 0}; const float4 no4 = make_float4(0.f, 0.f, 0.f, 0.f);
    pp_unroll([&](auto cc) { pp_chunk<EPI, decltype(cc)::value, MT, NTW>(p, acc, um0, un0, lane_off, no16, no4); }, std::make_integer_sequence<int, NCH>{});
  }
}

static int g_pp_ring = 6;            // tuning hook (oneprot_gemm_force_shape(32 + 64 * ring))
template <int EPI, bool GRAD, int NB>
static int launch_pp3(GemmArgs a, hipStream_t s) {
  constexpr int LDS = 3 * (256 + 128) * 128;       // three 48 KB stage buffers (NB is a spare template slot for A/B variants)
  static bool configured = false;
  static int n_cu = 0;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k_gemm_nt_pp<EPI, GRAD, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return OP_ELAUNCH;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return OP_ELAUNCH;
    n_cu = prop.multiProcessorCount;
    configured = true;
  }
  a.tiles_m = a.M / 256; a.tiles_n = a.N / 128;
  const long tiles = (long)a.tiles_m * a.tiles_n;
  int g8 = n_cu / 8;                                   // work-groups per XCD
  static const int g8_env = [] { const char* e = getenv("ONEPROT_PP_G8"); return e ? atoi(e) : 0; }();      // experiment hook (tools/ab/cu_scaling.py): fewer CUs at work; read once
  if (g8_env > 0 && g8_env < g8) g8 = g8_env;
  const long per_xcd = (tiles + 7) / 8;
  if (g8 > per_xcd) g8 = (int)per_xcd;
  if (g8 < 1) g8 = 1;
  hipLaunchKernelGGL((k_gemm_nt_pp<EPI, GRAD, NB>), dim3(g8 * 8), dim3(512), LDS, s, a, g8);
  return launch_status();
}
template <int EPI, bool GRAD>
static int launch_pp2(GemmArgs a, hipStream_t s) {
  return launch_pp3<EPI, GRAD, 3>(a, s);
}
template <int EPI>
static int launch_pp(GemmArgs a, hipStream_t s) {
  if (EPI == ONEPROT_EPI_BIAS_GELU && a.out1 != nullptr) return launch_pp2<EPI, true>(a, s);
  return launch_pp2<EPI, false>(a, s);
}

// whole 256 x 128 tiles, K a multiple of 64 and at least 16 K-steps (the epilogue is spread over 16 slots of the partner's K loop), 31-bit
// tile-relative byte offsets; epilogues that read operands from memory (residual, GELU', RoPE tables) are not built in this form yet
template <int EPI>
static bool pp_eligible(const GemmArgs& a) {
  if (EPI != ONEPROT_EPI_BF16 && EPI != ONEPROT_EPI_F32 && EPI != ONEPROT_EPI_BIAS_GELU) return false;
  if (a.M % 256 || a.N % 128 || a.K % 64 || a.K < 512 || a.lda % 64) return false;      // (lda: odd LDS-DMA pieces address their rows as even ^ 64 bytes)
  if ((size_t)256 * a.lda * 2 >= (1ull << 31) || (size_t)128 * a.ldb * 2 >= (1ull << 31)) return false;
  return true;
}

static int g_force_shape = -1;     // test / tuning hook, see launch_gemm
extern "C" void oneprot_gemm_force_shape(int shape) { g_force_shape = shape; }

// shapes: 0 = 128x128 (BK32, 3 stages, pipelined fragments, 48 KB LDS -> 3 blocks/CU)     1 = 256x128 (BK32, 3 stages, 72 KB -> 2 blocks/CU)
//         2 = 256x256 (BK32, 4 stages, pipelined fragments, 128 KB)   3 = 128x128 BK64 2 stages (first version; still the best for long K)
//         4 = 256x256 BK64 2 stages, pipelined fragments               5 = 256x256 BK32 4 stages, plain loop
//         16 + s = shape s with the direct-store epilogue (17, 19, 20);  32 = persistent ping-pong form (opt-in: see k_gemm_nt_pp)
// Heuristic from in-process A/B on the training shapes (tools/gemm_ab.py; all shapes lie within ~10 % of each other, vendor hipBLASLt
// runs the same plain shapes at 650-1030 TFLOP/s): long K -> 3, wide N with short K -> 4 (20 for the single-output GELU), otherwise 1; small
// problems -> 0.  The direct-store and ping-pong forms tie with these elsewhere (round-2 A/B: every form is bound by the same L2 -> LDS fill
// latency), so they are only selected where they measured faster.
template <int EPI>
static int launch_gemm(const GemmArgs& a, hipStream_t s) {
  int shape;
  if (g_force_shape < 0) {
    // Large whole-tile problems (every GEMM of the BASELINE configurations): the persistent 8-phase forms, 256 x 320 tiles where N allows
    // (d = 320 / 640 / 1280 encoders: N = d, 3d, 4d), else 256 x 256 (BERT-base d = 768; head_dim-64 QKV whose heads straddle 160-column wave
    // blocks).  Round-3 A/B on the cfg-2 shapes (tools/ab/g8_ab.py): 10-34 % less time than the per-tile kernels on every launch.
    for (int cfg = 1; cfg >= 0; --cfg) {
      const int rc = launch_gemm8(EPI, a, cfg, 192, s);
      if (rc != G8_NOT_ELIGIBLE) return rc;
    }
  }
  if (g_force_shape >= 0) shape = g_force_shape;
  else if (a.M < 2048) shape = 0;
  else if (EPI == ONEPROT_EPI_BIAS_RESID) shape = 19;   // residual folded into the accumulator's initial value, direct fp32 stores: -6 % (out-proj), -2.5 % (FFN-2)
  else if (a.K >= 1024) shape = 3;
  else if (a.N >= 2048) shape = (EPI == ONEPROT_EPI_BIAS_GELU && a.out1 == nullptr) ? 20 : 4;      // forward-only GELU (frozen tower): direct-store form, -6 %
  else shape = 1;
  if (shape >= 32 && (shape & 63) == 32) {            // ping-pong form; 32 + 64 * ring selects the ring depth (4..6) for A/B runs
    if (shape >> 6) g_pp_ring = shape >> 6;
    shape = 32;
  }
  if (shape >= 40 && shape <= 42) {      // 8-phase form (gemm_nt8.hip): 256 x 256 / 256 x 320 tiles; anything it is not built for takes the per-tile forms
    const int rc = launch_gemm8(EPI, a, shape - 40, 0, s);
    if (rc != G8_NOT_ELIGIBLE) return rc;
    shape = (EPI == ONEPROT_EPI_BIAS_RESID) ? 19 : (a.K >= 1024 ? 3 : (a.N >= 2048 ? 4 : 1));
  }
  if (shape == 8) {             // register-staged 256 x 256 form (falls back to the direct-store LDS-DMA form)
    if (rs_eligible<EPI>(a)) return launch_rs<EPI>(a, s);
    shape = 20;
  }
  if (shape == 32) {            // ping-pong form (falls back to the direct-store 256 x 128 form when the problem is not made of whole tiles)
    if (pp_eligible<EPI>(a)) return launch_pp<EPI>(a, s);
    shape = 17;
  }
  switch (shape) {
    // 16 + s: shape s with the direct-store epilogue (full tiles; falls back to s otherwise)
    case 20: return launch_shape<EPI, 2, 4, 8, 4, 64, 2, 32, 2, true, true, true>(a, s);
    case 19: return launch_shape<EPI, 2, 2, 4, 4, 64, 2, 64, 2, false, true, true>(a, s);
    case 17: return launch_shape<EPI, 4, 2, 4, 4, 32, 3, 32, 4, false, true, true>(a, s);
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

// x_out = resid + A W^T + bias and h = LayerNorm(x_out) in ONE launch of the 8-phase GEMM (gemm_epi8.h: epilogue_resid_ln): the FFN-2 GEMM of a pre-LN layer with
// the LayerNorm that follows it (hf modeling_esm.py:442-463 -> :429 of the next layer / emb_layer_norm_after).  Whole 256 x 320 tiles, N in {320, 640, 1280}.
extern "C" int oneprot_gemm_resid_ln8_eligible(int64_t M, int N, int K) { return gemm8_ln_eligible((long)M, N, K); }
// ---- the sched workspace (sched_ws.h): allocation helpers (host calls, never on a launch path) and its one-time initialisation
extern "C" size_t oneprot_sched_workspace_bytes(int64_t M_max) { return sched_workspace_bytes((long)M_max); }
extern "C" int oneprot_alloc_uncached(void** out, size_t bytes) {
  if (!out || bytes == 0) return OP_EINVAL;
  *out = nullptr;
  return hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached) == hipSuccess ? 0 : OP_ELAUNCH;
}
extern "C" int oneprot_free_uncached(void* p) { return (!p || hipFree(p) == hipSuccess) ? 0 : OP_ELAUNCH; }
extern "C" int oneprot_sched_workspace_init(void* ws, size_t bytes, void* stream) {
  if (!ws || bytes < SW_HEADER_BYTES || ((uintptr_t)ws & 127)) return OP_EINVAL;
  return hipMemsetAsync(ws, 0, bytes, (hipStream_t)stream) == hipSuccess ? 0 : OP_ELAUNCH;
}
// host-synchronous reads of the sticky flag (tests, end-of-epoch checks): 1 after a launch in which a bounded wait ran out (its rows are NaN)
extern "C" int oneprot_gemm_resid_ln8_error(const void* sched_ws) {
  unsigned e = 0;
  if (!sched_ws) return 0;
  if (hipMemcpy(&e, (const unsigned*)sched_ws + SW_LN_ERR, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}
// diagnostic (host-synchronous): ticket draws of the persistent GEMMs that had not returned where the kernel first looked (each cost one drained prefetch)
extern "C" int oneprot_sched_late_draws(const void* sched_ws) {
  unsigned e = 0;
  if (!sched_ws) return 0;
  if (hipMemcpy(&e, (const unsigned*)sched_ws + SW_LATE_DRAWS, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}
// diagnostic (host-synchronous): launches that have completed on this workspace (the device-side epoch the tags come from)
extern "C" int64_t oneprot_sched_epoch(const void* sched_ws) {
  unsigned e = 0;
  if (!sched_ws || hipMemcpy(&e, (const unsigned*)sched_ws + SW_EPOCH, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)e;
}
extern "C" int oneprot_gemm_resid_ln8_error_clear(void* sched_ws, void* stream) {
  if (!sched_ws) return OP_EINVAL;
  return hipMemsetAsync((unsigned*)sched_ws + SW_LN_ERR, 0, 4, (hipStream_t)stream) == hipSuccess ? 0 : OP_ELAUNCH;
}
extern "C" int oneprot_gemm_bf16_nt_resid_ln8(const void* A, const void* Bw, int64_t M, int N, int K, int lda, int ldb, const float* bias, const float* resid,
                                              float* x_out, const float* gamma, const float* beta, float eps, void* h_bf16, float* stats, void* sched_ws,
                                              size_t sched_ws_bytes, void* stream) {
  if (!A || !Bw || !resid || !x_out || !gamma || !beta || !h_bf16 || M <= 0 || N <= 0 || K <= 0 || M > 0x7fffffff) return OP_EINVAL;
  if ((N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) return OP_EINVAL;
  if (((uintptr_t)A | (uintptr_t)Bw | (uintptr_t)resid | (uintptr_t)x_out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)h_bf16 | (uintptr_t)bias | (uintptr_t)stats) & 15) return OP_EINVAL;
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)Bw; a.M = (int)M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.bias = bias;
  a.out0 = x_out; a.out1 = h_bf16; a.out2 = stats; a.aux = resid; a.cos = gamma; a.sin = beta; a.q_scale = eps; a.L = 0; a.H = 0; a.hd = 0;
  a.tiles_m = 0; a.tiles_n = 0; a.sup_m = g_sup_m; a.sup_n = g_sup_n; a.nt_store = 0;
  a.ln_part = nullptr; a.ln_slots = 0; a.ln_poll_max = 0; a.sched = nullptr; a.dyn = 0;
  const int rc = launch_gemm8_ln(a, sched_ws, sched_ws_bytes, (hipStream_t)stream);
  return rc == G8_NOT_ELIGIBLE ? OP_EINVAL : rc;
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
  a.tiles_m = 0; a.tiles_n = 0; a.sup_m = g_sup_m; a.sup_n = g_sup_n; a.nt_store = g_nt_store;
  a.ln_part = nullptr; a.ln_slots = 0; a.ln_poll_max = 0; a.sched = nullptr; a.dyn = 0;
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
