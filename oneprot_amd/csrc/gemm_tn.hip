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

// Fixed-order sum of the S slabs into dW; the last `bias_blocks` work-groups of the grid do the same for the bias slabs (one launch instead of
// two: 120 weight gradients per sub-step, and a 7 us launch of 3 work-groups between every two of them)
__global__ void __launch_bounds__(256) k_tn_reduce(const float* __restrict__ slab, float* __restrict__ dW, size_t n4, size_t stride4, int S, int accumulate,
                                                   const float* __restrict__ bias_slab, float* __restrict__ db, int N, int bias_blocks) {
  const int wblocks = (int)gridDim.x - bias_blocks;
  if ((int)blockIdx.x >= wblocks) {
    const int n = ((int)blockIdx.x - wblocks) * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int p = 0; p < S; ++p) s += bias_slab[(size_t)p * N + n];
    db[n] = accumulate ? db[n] + s : s;
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)wblocks * 256) {
    float4 s = reinterpret_cast<const float4*>(slab)[i];
    for (int p = 1; p < S; ++p) {
      const float4 v = reinterpret_cast<const float4*>(slab)[i + (size_t)p * stride4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) { const float4 o = reinterpret_cast<const float4*>(dW)[i]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    reinterpret_cast<float4*>(dW)[i] = s;
  }
}

// =====================================================================================================================================
// 8-PHASE form (round 3; the structure of gemm_nt8.hip carried over to the weight gradient).
// The 128 x 128 kernels above run at ~740 TFLOP/s because they are bound by the L2 -> LDS fill rate a CU sustains (~45 GB/s with every CU
// streaming) at 64 FLOP per staged byte; this form stages 145 FLOP per byte: output tile 256 (n) x 320 (k) or 320 x 256, one 512-thread work-group
// per (split, tile) work item, eight waves 4 (n) x 2 (k) with 160 accumulator registers each, in two groups of four waves that run half a phase
// apart: while one group issues its 40 MFMAs the other gathers its next fragments and issues its share of the LDS-DMA prefetch.
//   * a PHASE consumes one UNIT = 32 tokens of both operands as they lie in HBM (36 KB); four units form a ring (144 KB); the unit three phases
//     ahead is requested while the current one is read: two units in flight behind ONE counted vmcnt per phase;
//   * fragments (8 consecutive tokens of one output row / column) are gathered by ds_read_b64_tr_b16; a phase's gathers are retired (lgkmcnt 0)
//     before its first barrier -- they run under the partner group's MFMAs -- so the unit's slot can be refilled one phase later;
//   * the 16-column blocks of a slab are dealt to the waves round-robin (block 4 i + wn of dY, 2 j + wk of X), so that consecutive tiles of one
//     wave are a constant 128 / 64 bytes apart and the bank swizzle (XOR of the 16-byte chunk index with a function of the token row) only
//     touches the low bits of the tile index: a gather address is one of 1-4 per-lane bases plus an immediate offset, no arithmetic per gather;
//   * swizzles put the 8 token rows a half-wave gather touches in 8 different 32-byte bank slots: rows of 512 B with tn_f, rows of 640 B
//     (2.5 bank rows: odd rows start half a bank row in) with tn_f640; applied on the LDS-DMA source address;
//   * dY columns past N (N = 1920, 640 are not multiples of 256: the last n tile is partial) are fetched from 256 columns further left
//     instead -- their products land in accumulator rows that are never stored.
// Needs K % 320 == 0 with N >= 256, or N % 320 == 0 and K % 256 == 0; M a multiple of 32.  Everything else takes the kernels above.
__device__ __forceinline__ int tn_f640(int r) { return 2 * (((r >> 1) & 1) | (((r >> 3) & 1) << 1)); }
template <int PITCH> __device__ __forceinline__ int tn8_swz(int r) { return PITCH == 640 ? tn_f640(r) : tn_f(r); }

__device__ __forceinline__ void tn8_glds16(unsigned voff, const unsigned char* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ bf8_t tn8_frag(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  s16x8 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  return __builtin_bit_cast(bf8_t, o);
}

#ifdef TN8_STAMP      // diagnostic build only (tools/ab/tn8_stamps.py): s_memtime at six points of the first 64 phases, first wave of each group of work-group 0
static __device__ unsigned long long g_tn8_stamps[2 * 64 * 8];
extern "C" int oneprot_tn8_debug_read(unsigned long long* dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tn8_stamps), sizeof(g_tn8_stamps)) == hipSuccess ? 0 : -2; }
#define TN8_ST(i) do { if (blockIdx.x == 0 && (wave & 3) == 0 && p < 64) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0) g_tn8_stamps[(grp * 64 + p) * 8 + (i)] = t_; } } while (0)
#else
#define TN8_ST(i) do { } while (0)
#endif
#define TN8_RING 4

// YC x XC = output tile (dY columns x X columns): 256 x 320 or 320 x 256
template <int YC, int XC>
__global__ void __launch_bounds__(512, 2) k_gemm_tn8(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, int M, int N, int K, int ldy, int ldx,
                                                     float* __restrict__ slab, float* __restrict__ bias_slab, int tiles_k, int tiles_all, int S) {
  constexpr int MT = YC / 64, NT = XC / 32;                 // 16 x 16 accumulator tiles per wave: 4 x 10 or 5 x 8
  constexpr int PY = YC * 2, PX = XC * 2;                   // LDS row pitch of the slabs (bytes)
  constexpr int UY = 32 * PY, UX = 32 * PX, UNIT = UY + UX; // 36 KB
  constexpr int YP = UY / 1024, NP = UNIT / 1024;           // LDS-DMA pieces: dY slab, whole unit (36)
  constexpr int IPW = (NP + 7) / 8;                         // instructions per wave (the last one only in waves 0..3)
  constexpr int NBY = PY == 640 ? 1 : 2, NBX = PX == 640 ? 2 : 4;      // gather bases per read (see the header comment)
  static_assert(MT * NT == 40 && NP == 36 && IPW == 5, "tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int per_xcd = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int w = xcd * per_xcd + seq;                        // split-major work items, a contiguous run per XCD (see k_gemm_tn)
  if (w >= tiles_all * S) return;
  const int split = w / tiles_all, tile = w - split * tiles_all;
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * YC, k0 = tk * XC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wn = wave >> 1, wk = wave & 1;
  // token units of this split: the M / 32 units are dealt as evenly as possible
  const int units = M >> 5, ubase = units / S, urem = units - ubase * S;
  const int P = ubase + (split < urem ? 1 : 0);
  const int u0 = split * ubase + (split < urem ? split : urem);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);

  // ---- LDS-DMA pieces of a unit; piece i * 8 + wave is this wave's instruction i
  unsigned voff[IPW]; unsigned pdst[IPW]; bool isy[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int piece = i * 8 + wave;
    isy[i] = piece < YP;
    if (isy[i]) {
      const int idx = piece * 64 + lane, row = idx / (PY / 16), pos = idx - row * (PY / 16);
      int col = n0 + (pos ^ tn8_swz<PY>(row)) * 8;
      if (col >= N) col -= 256;                              // partial last n tile: any valid columns of the same row (their output rows are not stored)
      voff[i] = (unsigned)row * (unsigned)ldy * 2u + (unsigned)col * 2u;
      pdst[i] = piece * 1024;
    } else {
      const int idx = (piece - YP) * 64 + lane, row = idx / (PX / 16), pos = idx - row * (PX / 16);
      voff[i] = (unsigned)row * (unsigned)ldx * 2u + (unsigned)(k0 + (pos ^ tn8_swz<PX>(row)) * 8) * 2u;
      pdst[i] = UY + (piece - YP) * 1024;
    }
  }
  const unsigned char* ybase = reinterpret_cast<const unsigned char*>(dY + (size_t)u0 * 32 * ldy);
  const unsigned char* xbase = reinterpret_cast<const unsigned char*>(X + (size_t)u0 * 32 * ldx);
  const size_t ystep = (size_t)32 * ldy * 2, xstep = (size_t)32 * ldx * 2;
  auto issue = [&](int u, int slot) {
    const unsigned char* yb = ybase + (size_t)u * ystep;
    const unsigned char* xb = xbase + (size_t)u * xstep;
    const unsigned dst = lds0 + slot * UNIT;
#pragma unroll
    for (int i = 0; i < IPW; ++i)
      if (i + 1 < IPW || wave < 4) tn8_glds16(voff[i], isy[i] ? yb : xb, dst + pdst[i]);
  };

  // ---- gather bases.  Lane (g = lane >> 4, i16 = lane & 15) of a fragment reads tokens 8 g + (i16 >> 2) [+ 4 for the second read], columns
  // 4 (i16 & 3) .. + 3 of its 16-column block.  Tile t of the wave is block 4 t + wn (dY) / 2 t + wk (X): chunk 8 t + Ly / 4 t + Lx.
  const int g = lane >> 4, i16 = lane & 15;
  const int rr[2] = {8 * g + (i16 >> 2), 8 * g + (i16 >> 2) + 4};
  const int within = (i16 & 1) * 8, cb = (i16 & 3) >> 1;
  unsigned yb_[NBY][2], xb_[NBX][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
    for (int k = 0; k < NBY; ++k)
      yb_[k][rd] = rr[rd] * PY + (((k * 8 + 2 * wn + cb) ^ tn8_swz<PY>(rr[rd])) << 4) + within - k * 128;
#pragma unroll
    for (int k = 0; k < NBX; ++k)
      xb_[k][rd] = UY + rr[rd] * PX + (((k * 4 + 2 * wk + cb) ^ tn8_swz<PX>(rr[rd])) << 4) + within - k * 64;
  }

  f32x4 acc[MT][NT], accb[MT];
  {
    float z = 0.f;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      accb[i] = (f32x4){z, z, z, z};
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){z, z, z, z};
    }
  }
  const bool do_bias = bias_slab != nullptr && tk == 0 && wk == 0;      // wave-uniform
  const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bf8_t ones = __builtin_bit_cast(bf8_t, ones_u);

  // ---- prologue: units 0..2 requested, unit 0 landed
#pragma unroll
  for (int u = 0; u < 3; ++u)
    if (u < P) issue(u, u);
  if (P > 2) { if (grp == 0) tn_wait_vmcnt<2 * IPW>(); else tn_wait_vmcnt<2 * (IPW - 1)>(); }
  else tn_wait_vmcnt<0>();
  asm volatile("s_barrier" ::: "memory");
  if (grp == 1) asm volatile("s_barrier" ::: "memory");     // group 1 runs one barrier behind

  int slot = 0, nslot = 3;
#pragma clang loop unroll(disable)
  for (int p = 0; p < P; ++p) {
    const unsigned char* ub = smem + slot * UNIT;
    bf8_t fa[MT], fb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = tn8_frag(ub + yb_[i % NBY][0] + i * 128, ub + yb_[i % NBY][1] + i * 128);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = tn8_frag(ub + xb_[j % NBX][0] + j * 64, ub + xb_[j % NBX][1] + j * 64);
    __builtin_amdgcn_sched_barrier(0);
    TN8_ST(2);
    const bool more = p + 3 < P;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the gathers of this unit are done: its slot is refilled in the NEXT phase
    TN8_ST(4);
    // unit p+1 landed before this phase's first barrier (it is read from the next phase on); unit p+2 may stay in flight (p+3 is requested
    // below, from the MFMA slot: the ~80 issue cycles per LDS-DMA piece are hidden there, the gather slot is the longer one)
    if (p + 2 < P) { if (grp == 0) tn_wait_vmcnt<IPW>(); else tn_wait_vmcnt<IPW - 1>(); }
    else tn_wait_vmcnt<0>();
    TN8_ST(5);
    asm volatile("s_barrier" ::: "memory");
    TN8_ST(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      if (i == 0) {                                         // DMA pieces of unit p+3 behind the first ten MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(p + 3, nslot);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    TN8_ST(3);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < MT; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    TN8_ST(6);
    asm volatile("s_barrier" ::: "memory");
    TN8_ST(1);
    slot = slot + 1 == TN8_RING ? 0 : slot + 1;
    nslot = nslot + 1 == TN8_RING ? 0 : nslot + 1;
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");     // pairs with group 1's last barrier

  // ---- slab: D[row n][column k], lane (fr = column, fq * 4 + r = row) of every tile; tile (i, j) = dY block 4 i + wn, X block 2 j + wk
  float* out = slab + (size_t)split * N * K;
  const int fq = lane >> 4, fr = lane & 15;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + (4 * i + wn) * 16 + fq * 4 + r;
      if (n < N) {
#pragma unroll
        for (int j = 0; j < NT; ++j) out[(size_t)n * K + k0 + (2 * j + wk) * 16 + fr] = acc[i][j][r];
      }
    }
  if (do_bias && fr == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (4 * i + wn) * 16 + fq * 4 + r;
        if (n < N) bias_slab[(size_t)split * N + n] = accb[i][r];
      }
  }
}

// configuration of the 8-phase form for an (N, K) weight gradient: 1 = 256 x 320 tiles, 2 = 320 x 256 tiles, 0 = not eligible
static int tn8_config(int N, int K) {
  if (K % 320 == 0 && N % 256 == 0) return 1;
  if (N % 320 == 0 && K % 256 == 0) return 2;
  if (K % 320 == 0 && N >= 256 && N % 64 == 0) return 1;       // partial last n tile
  return 0;
}
static int tn8_tiles(int cfg, int N, int K) { return cfg == 1 ? ((N + 255) / 256) * (K / 320) : (N / 320) * (K / 256); }
// oneprot_cu_reserve: CUs left to co-resident kernels (RCCL channels of an overlapped all-reduce) when the one-shot work items of the weight-gradient GEMM are
// counted: its grid is at most one work-group per CU, and a work-group that finds no CU free runs AFTER the others -- twice the launch time whatever the
// number of CUs held.  With a reserve the token range is cut into fewer, slightly longer splits instead (another summation order than with reserve 0:
// still fixed, still deterministic).  Process-wide; 0 by default (single-GPU runs).
static int g_cu_reserve = 0;
extern "C" void oneprot_cu_reserve(int cus) { g_cu_reserve = cus < 0 ? 0 : cus; }
static int tn8_cus_all() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0; hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n_cu;
}
static int tn8_cus() {
  const int n = tn8_cus_all() - g_cu_reserve;
  return n < 64 ? 64 : n;
}
// token splits: one work item per CU at most, at most 64, at least 8 units (256 tokens) per split; 0 = leave it to the 128 x 128 kernels
static int tn8_splits(int64_t M, int tiles, int n_cu, int64_t max_by_workspace) {
  if (M % 32) return 0;
  int S = n_cu / tiles;
  if (S > 64) S = 64;
  if (S > M / 256) S = (int)(M / 256);
  if (S > max_by_workspace) S = (int)max_by_workspace;
  if (S < 1 || (int64_t)S * tiles * 2 < n_cu) return 0;       // would leave more than half the chip idle
  return S;
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
  int S = tn_splits(0, tiles);
  const int cfg = tn8_config(N, K);
  if (cfg) {                                                  // the 8-phase form may split finer (fewer, larger tiles)
    int S8 = tn8_cus_all() / tn8_tiles(cfg, N, K);
    if (S8 > 64) S8 = 64;
    if (S8 > S) S = S8;
  }
  return (size_t)S * N * K * sizeof(float) + (size_t)S * N * sizeof(float);
}

extern "C" int oneprot_gemm_bf16_tn(const void* dY, const void* X, int64_t M, int N, int K, int ldy, int ldx, float* dW, float* dbias, void* workspace,
                                    size_t workspace_bytes, int accumulate, void* stream) {
  if (!dY || !X || !dW || !workspace || M <= 0 || N <= 0 || K <= 0 || M > 0x7fffffff) return OP_EINVAL;
  if ((N & 7) || (K & 7) || (ldy & 7) || (ldx & 7) || ldy < N || ldx < K || ((N * (int64_t)K) & 3)) return OP_EINVAL;
  if (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)workspace) & 15) return OP_EINVAL;
  const int variant = (g_tn_variant < 0 || g_tn_variant == 3) ? 2 : g_tn_variant;      // what runs when the 8-phase form does not take the problem
  hipStream_t s = (hipStream_t)stream;
  if (g_tn_variant < 0 || g_tn_variant == 3) {            // 8-phase form where the problem fits its tiles (every weight gradient of a d = 640 / 1280 / 320 encoder)
    const int cfg = tn8_config(N, K);
    if (cfg && (size_t)32 * ldy * 2 < (1u << 31) && (size_t)32 * ldx * 2 < (1u << 31)) {
      const int tiles8 = tn8_tiles(cfg, N, K);
      const size_t per_split = (size_t)N * K * sizeof(float) + (dbias ? (size_t)N * sizeof(float) : 0);
      const int S8 = tn8_splits(M, tiles8, tn8_cus(), 64);
      if (S8 > 0 && workspace_bytes < (size_t)S8 * per_split) return OP_EINVAL;      // smaller than oneprot_gemm_bf16_tn_workspace(N, K): slabs would overrun
      if (S8 > 0) {
        constexpr int LDS8 = TN8_RING * 36 * 1024;
        static bool c8 = false;
        if (!c8) {
          if (hipFuncSetAttribute((const void*)k_gemm_tn8<256, 320>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8) != hipSuccess) return OP_ELAUNCH;
          if (hipFuncSetAttribute((const void*)k_gemm_tn8<320, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8) != hipSuccess) return OP_ELAUNCH;
          c8 = true;
        }
        const int tiles_k8 = cfg == 1 ? K / 320 : K / 256;
        float* bias_slab8 = dbias ? (float*)workspace + (size_t)S8 * N * K : nullptr;
        const dim3 grid(8 * ((tiles8 * S8 + 7) / 8));
        if (cfg == 1) hipLaunchKernelGGL((k_gemm_tn8<256, 320>), grid, dim3(512), LDS8, s, (const bf16_t*)dY, (const bf16_t*)X, (int)M, N, K, ldy, ldx, (float*)workspace, bias_slab8, tiles_k8, tiles8, S8);
        else hipLaunchKernelGGL((k_gemm_tn8<320, 256>), grid, dim3(512), LDS8, s, (const bf16_t*)dY, (const bf16_t*)X, (int)M, N, K, ldy, ldx, (float*)workspace, bias_slab8, tiles_k8, tiles8, S8);
        const size_t n4_8 = ((size_t)N * K) >> 2;
        size_t blocks8 = (n4_8 + 255) / 256; if (blocks8 > 4096) blocks8 = 4096;
        const int bb8 = dbias ? (N + 255) / 256 : 0;
        hipLaunchKernelGGL(k_tn_reduce, dim3((unsigned)blocks8 + bb8), dim3(256), 0, s, (const float*)workspace, dW, n4_8, n4_8, S8, accumulate, (const float*)bias_slab8, dbias, N, bb8);
        return launch_status();
      }
    }
  }
  const int tiles_n = (N + 127) / 128, tiles_k = (K + 127) / 128, tiles = tiles_n * tiles_k;
  const int S = tn_splits(M, tiles);
  if (workspace_bytes < (size_t)S * N * K * sizeof(float) + (dbias ? (size_t)S * N * sizeof(float) : 0)) return OP_EINVAL;      // slabs would overrun
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
  const size_t n4 = ((size_t)N * K) >> 2;
  size_t blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096;
  const int bb = dbias ? (N + 255) / 256 : 0;
  hipLaunchKernelGGL(k_tn_reduce, dim3((unsigned)blocks + bb), dim3(256), 0, s, (const float*)workspace, dW, n4, n4, S, accumulate, (const float*)bias_slab, dbias, N, bb);
  return launch_status();
}
