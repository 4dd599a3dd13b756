// Flash-style multi-head attention for the ESM-2 / BERT encoders on gfx950 (hf modeling_esm.py:292-317, 340-395).
// Scores never touch HBM; the backward recomputes P from q, k and the saved log-sum-exp.
//
// Data layout: q (already scaled by hd^-1/2 * log2(e) and rotated -- scores come out in log2 units, so P = exp2(S - m) needs no multiply),
// k (rotated), v are bf16 [B, H, L, hd] (head-major, written by
// the QKV GEMM epilogue), so one (b, h) slab is a contiguous L*hd*2-byte run.  ctx / dctx are bf16 [B*L, H*hd]
// (token-major, the A operand of the out-projection GEMM).  key_bias is the additive key-padding mask [B, L] fp32.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16, wave64; lane maps verified by csrc/probe_gfx950.hip):
//   forward, per wave = 32 queries, per 32-key tile:
//     S^T[key][query] = K * Q^T      (A = K rows from LDS, B = Q rows in registers)  -> query on the lane, keys in the 16
//                                     accumulator registers: the softmax row-reduce is in-lane + one lane^32 exchange;
//     O^T[d][query] += V^T * P       (P straight from the accumulator registers as the B operand -- no LDS round trip;
//                                     A = V^T gathered with ds_read_b64_tr_b16 from the row-major V tile)
//   backward dQ kernel mirrors the forward (+ dP^T = V dO^T, dQ^T += K^T dS^T) and forms delta = rowsum(dO * O) for both backward kernels;
//   backward dK/dV kernel puts the key on the lane: S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
// Tiles in LDS are row-major with a 16-byte-chunk XOR swizzle that keeps the ds_read_b128 row reads conflict-free.
#include "common.h"
#include "../../include/oneprot_hip.h"
#include <float.h>

#define LOG2E 1.4426950408889634f
#define KC 256        // keys (or queries) staged per LDS chunk
#define RESCALE_THR 10.0f

template <int HD> struct Cfg {
  static constexpr int HDP = HD < 32 ? 32 : HD;      // LDS row pitch in elements (hd=16 rows are zero-padded to 32)
  static constexpr int NCH = HDP / 8;                // 16-byte chunks per row
  static constexpr int KSTEPS = HD / 16;             // 16-deep contraction steps over the head dim
  static constexpr int DBLK = HDP / 32;              // 32-row blocks of the head dim on the MFMA M axis
  static constexpr int ROWB = HDP * 2;               // row pitch in bytes
};

template <int HDP> __device__ __forceinline__ int swz(int row, int chunk) {
  return HDP == 32 ? (chunk ^ ((row >> 2) & 3)) : (chunk ^ ((row >> 1) & 7));
}

// cooperative [nrows x HD] bf16 tile load (global row pitch `gpitch` elements, nrows <= KC) into a swizzled, zero-padded LDS tile.
// All 16-byte global loads of a tile are issued back to back into registers and committed to LDS afterwards: one exposed HBM/L2
// latency per tile instead of one per item (measured: attention forward 466 -> 358 us at the cfg-2 shape).
template <int HD> struct TileRegs { u32x4 v[KC * Cfg<HD>::NCH / 256]; };
template <int HD>
__device__ __forceinline__ void tile_fetch(TileRegs<HD>& r, const bf16_t* g, size_t gpitch, int rows_valid) {
  typedef Cfg<HD> C;
#pragma unroll
  for (int it = 0; it < KC * C::NCH / 256; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int row = idx / C::NCH, ch = idx - row * C::NCH;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < rows_valid && ch < HD / 8) v = *reinterpret_cast<const u32x4*>(g + (size_t)row * gpitch + ch * 8);
    r.v[it] = v;
  }
}
template <int HD>
__device__ __forceinline__ void tile_commit(const TileRegs<HD>& r, unsigned char* lds, int nrows) {
  typedef Cfg<HD> C;
#pragma unroll
  for (int it = 0; it < KC * C::NCH / 256; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int row = idx / C::NCH, ch = idx - row * C::NCH;
    if (row < nrows) *reinterpret_cast<u32x4*>(lds + row * C::ROWB + (swz<C::HDP>(row, ch) << 4)) = r.v[it];
  }
}
template <int HD>
__device__ __forceinline__ void load_tile(unsigned char* lds, const bf16_t* g, size_t gpitch, int rows_valid, int nrows) {
  TileRegs<HD> r;
  tile_fetch<HD>(r, g, gpitch, rows_valid);
  tile_commit<HD>(r, lds, nrows);
}
// two tiles with the same row count; for 64-byte rows both tiles' loads are in flight together (32 VGPRs), wider rows go one after the other
template <int HD>
__device__ __forceinline__ void load_tile_pair(unsigned char* lds0, const bf16_t* g0, size_t pitch0, unsigned char* lds1, const bf16_t* g1, size_t pitch1,
                                               int rows_valid, int nrows) {
  if (Cfg<HD>::NCH == 4) {
    TileRegs<HD> r0, r1;
    tile_fetch<HD>(r0, g0, pitch0, rows_valid);
    tile_fetch<HD>(r1, g1, pitch1, rows_valid);
    tile_commit<HD>(r0, lds0, nrows);
    tile_commit<HD>(r1, lds1, nrows);
  } else {
    load_tile<HD>(lds0, g0, pitch0, rows_valid, nrows);
    load_tile<HD>(lds1, g1, pitch1, rows_valid, nrows);
  }
}

// row fragment: 8 consecutive head-dim elements (16*step + 8*h ...) of `row`
template <int HD>
__device__ __forceinline__ bf8_t rd_row(const unsigned char* lds, int row, int step, int h) {
  typedef Cfg<HD> C;
  return *reinterpret_cast<const bf8_t*>(lds + row * C::ROWB + (swz<C::HDP>(row, 2 * step + h) << 4));
}

// transposed fragment for the 32-row tile starting at row kb: element j of lane (r, h) = T[kb + 16 s + 8 (j>>2) + 4 h + (j&3)][32 db + r]
template <int HD>
__device__ __forceinline__ bf8_t rd_tr(const unsigned char* lds, int kb, int s, int db, int lane) {
  typedef Cfg<HD> C;
  const int g = lane >> 4, i = lane & 15, h = lane >> 5;
  const int col = 32 * db + 16 * (g & 1) + 4 * (i & 3);
  const int r0 = kb + 16 * s + 4 * h + (i >> 2);
  const int r1 = r0 + 8;
  const unsigned char* p0 = lds + r0 * C::ROWB + (swz<C::HDP>(r0, col >> 3) << 4) + (col & 7) * 2;
  const unsigned char* p1 = lds + r1 * C::ROWB + (swz<C::HDP>(r1, col >> 3) << 4) + (col & 7) * 2;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  s16x8 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  return __builtin_bit_cast(bf8_t, o);
}

__device__ __forceinline__ bf8_t pack8(const f32x16& x, int s) {
  u32x4 w;
  w.x = pack2bf(x[8 * s + 0], x[8 * s + 1]); w.y = pack2bf(x[8 * s + 2], x[8 * s + 3]);
  w.z = pack2bf(x[8 * s + 4], x[8 * s + 5]); w.w = pack2bf(x[8 * s + 6], x[8 * s + 7]);
  return __builtin_bit_cast(bf8_t, w);
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// x ~= bf16(w0) + bf16(w1) + bf16(w2) (24 mantissa bits): row constants enter the score chains through an extra MFMA k-step whose operands
// are bf16, so they are split.  Returns w0 | w1 << 16 and w2 (low half).  Non-finite x (padding: -inf) stays in w0 alone.
__device__ __forceinline__ void split3_bf16(float x, unsigned& w01, unsigned& w2) {
  const unsigned a = pack2bf(x, 0.f) & 0xffffu;
  const float r1 = (fabsf(x) < 3.0e38f) ? x - bflo(a) : 0.f;
  const unsigned b = pack2bf(r1, 0.f) & 0xffffu;
  const float r2 = r1 - bflo(b);
  w01 = a | (b << 16);
  w2 = pack2bf(r2, 0.f) & 0xffffu;
}

// XCD-aware decode: blocks with the same (b,h) land on one XCD (they share K/V through its L2)
__device__ __forceinline__ void decode_block(int nblk_per_bh, int nbh, int& bh, int& blk) {
  const int id = blockIdx.x, xcd = id & 7, seq = id >> 3;
  bh = (seq / nblk_per_bh) * 8 + xcd;
  blk = seq % nblk_per_bh;
  (void)nbh;
}

// =========================================================================================================
// forward
// =========================================================================================================
template <int HD>
__global__ void __launch_bounds__(256, 2) k_attn_fwd(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                  const float* __restrict__ key_bias, bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int B, int H,
                                                  int L, int nqb) {
  typedef Cfg<HD> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = sK + KC * C::ROWB;
  u32x4* sE = reinterpret_cast<u32x4*>(sV + KC * C::ROWB);       // per key: bf16 [1, 1, 1, bias, 0, 0, 0, 0]; slot KC = zeros
  int bh, qb;
  decode_block(nqb, B * H, bh, qb);
  if (bh >= B * H) return;
  const int b = bh / H, head = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int q0 = qb * 128 + wave * 32;
  const int qidx = q0 + (lane & 31);
  const int qrow = qidx < L ? qidx : L - 1;
  const bf16_t* qbase = q + (size_t)bh * L * HD;
  const bf16_t* kbase = k + (size_t)bh * L * HD;
  const bf16_t* vbase = v + (size_t)bh * L * HD;
  bf8_t qf[C::KSTEPS];
#pragma unroll
  for (int st = 0; st < C::KSTEPS; ++st) qf[st] = *reinterpret_cast<const bf8_t*>(qbase + (size_t)qrow * HD + 16 * st + 8 * h);
  // q arrives pre-multiplied by hd^-1/2 * log2(e): scores are in log2 units.  The running maximum m (log2 units) is subtracted INSIDE
  // the score MFMA chain by one extra k-step: K side [1, 1, 1, bias_key], Q side [-m split into three bf16, 1]  => p = exp2(S') directly.
  float m = 0.f, l = 0.f;
  u32x4 qe = {0u, 0u, 0u, 0u};
  if (h == 0) { qe.x = 0u; qe.y = 0x3F800000u; }        // -m = 0, slot 3 = 1.0 (picks up the key bias)
  const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  f32x16 acc[C::DBLK], lacc = zero16();
#pragma unroll
  for (int d = 0; d < C::DBLK; ++d) acc[d] = zero16();
  bool first = true;

  for (int kc0 = 0; kc0 < L; kc0 += KC) {
    const int nkeys = min(KC, L - kc0);
    const int nrows = (nkeys + 31) & ~31;
    __syncthreads();
    load_tile_pair<HD>(sK, kbase + (size_t)kc0 * HD, HD, sV, vbase + (size_t)kc0 * HD, HD, nkeys, nrows);
    for (int i = threadIdx.x; i <= KC; i += 256) {
      u32x4 e = {0u, 0u, 0u, 0u};
      if (i < nrows) {
        const float bv = i < nkeys ? (key_bias ? key_bias[(size_t)b * L + kc0 + i] : 0.f) : -INFINITY;
        e.x = 0x3F803F80u; e.y = 0x3F80u | (pack2bf(bv, 0.f) << 16);
      }
      sE[i] = e;
    }
    __syncthreads();
    for (int t = 0; t < nrows / 32; ++t) {
      const u32x4 ke = sE[h ? KC : t * 32 + (lane & 31)];
      f32x16 s = MFMA32(__builtin_bit_cast(bf8_t, ke), __builtin_bit_cast(bf8_t, qe), zero16());
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) s = MFMA32(rd_row<HD>(sK, t * 32 + (lane & 31), st, h), qf[st], s);
      float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
      mx = fmaxf(mx, s[15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // deferred rescale (s is already relative to m): the running max moves when a tile exceeds it by more than the threshold, and
      // unconditionally on the very first tile (m starts at 0, not at the row maximum)
      if (first || __any(mx > RESCALE_THR * LOG2E)) {
        float dlt = first ? mx : fmaxf(mx, 0.f);
        if (!(dlt > -1e30f)) dlt = 0.f;                    // fully masked so far: keep m
        const float alpha = __builtin_amdgcn_exp2f(-dlt);
        l = (l + lacc[0]) * alpha;
        lacc = zero16();
#pragma unroll
        for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] -= dlt;
        m += dlt;
        if (h == 0) {
          const float nm = -m;
          const unsigned w0 = pack2bf(nm, 0.f); const float r1 = nm - bflo(w0);
          const unsigned w1 = pack2bf(r1, 0.f); const float r2 = r1 - bflo(w1);
          qe.x = (w0 & 0xffffu) | (w1 << 16); qe.y = pack2bf(r2, 1.0f);
        }
        first = false;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const bf8_t pf = pack8(s, sb);
        lacc = MFMA32(__builtin_bit_cast(bf8_t, ones), pf, lacc);
#pragma unroll
        for (int d = 0; d < C::DBLK; ++d) acc[d] = MFMA32(rd_tr<HD>(sV, t * 32, sb, d, lane), pf, acc[d]);
      }
    }
  }
  const float lt = l + lacc[0];
  const float inv = lt > 0.f ? 1.0f / lt : 0.f;
  if (qidx < L) {
    bf16_t* dst = ctx + ((size_t)b * L + qidx) * (H * HD) + head * HD;
#pragma unroll
    for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * d + 8 * g + 4 * h;
        if (dd < HD) {
          u32x2 w; w.x = pack2bf(acc[d][4 * g] * inv, acc[d][4 * g + 1] * inv); w.y = pack2bf(acc[d][4 * g + 2] * inv, acc[d][4 * g + 3] * inv);
          *reinterpret_cast<u32x2*>(dst + dd) = w;
        }
      }
    if (lse_out && h == 0) lse_out[(size_t)bh * L + qidx] = (m + __log2f(lt)) * 0.6931471805599453f;
  }
}

template <int HD> static size_t fwd_lds() { return (size_t)2 * KC * Cfg<HD>::ROWB + (KC + 1) * 16; }

template <int HD>
static int launch_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, hipStream_t s) {
  const int nqb = (L + 127) / 128;
  const int nbh8 = ((B * H + 7) / 8) * 8;
  hipLaunchKernelGGL(k_attn_fwd<HD>, dim3(nbh8 * nqb), dim3(256), fwd_lds<HD>(), s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (bf16_t*)ctx, lse, B, H, L, nqb);
  return launch_status();
}

extern "C" int oneprot_attn_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, int hd,
                                void* stream) {
  if (!q || !k || !v || !ctx || B <= 0 || H <= 0 || L <= 0) return OP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (hd) {
    case 16: return launch_fwd<16>(q, k, v, key_bias, ctx, lse, B, H, L, s);
    case 32: return launch_fwd<32>(q, k, v, key_bias, ctx, lse, B, H, L, s);
    case 64: return launch_fwd<64>(q, k, v, key_bias, ctx, lse, B, H, L, s);
    default: return OP_EINVAL;
  }
}

// =========================================================================================================
// backward
// =========================================================================================================
// inverse rotary on a gradient held as O^T-style accumulators: lane owns position `pos`, registers hold head-dim rows
// d = 32 db + 8 g + 4 h + e.  dx1 = dy1 c + dy2 s ; dx2 = dy2 c - dy1 s   (transpose of hf modeling_esm.py:48-79), then * scale.
template <int HD>
__device__ __forceinline__ void unrope_store(f32x16 (&acc)[Cfg<HD>::DBLK], const float* __restrict__ cosT, const float* __restrict__ sinT, int pos, int h,
                                             float scale, bool rope, bf16_t* dst) {
  typedef Cfg<HD> C;
  constexpr int HALF = HD / 2;
#pragma unroll
  for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = 32 * d + 8 * g + 4 * h;
      if (dd >= HD) continue;
      float o[4];
      if (rope) {
        const bool lo = dd < HALF;
        const int jj = lo ? dd : dd - HALF;
        const float4 c = *reinterpret_cast<const float4*>(cosT + (size_t)pos * HALF + jj);
        const float4 sn = *reinterpret_cast<const float4*>(sinT + (size_t)pos * HALF + jj);
        float own[4], par[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) own[e] = acc[d][4 * g + e];
        // partner rows: +-HALF in the head dim
        if (HD == 64) {
#pragma unroll
          for (int e = 0; e < 4; ++e) par[e] = acc[C::DBLK - 1 - d][4 * g + e];
        } else if (HD == 32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) par[e] = acc[0][4 * (g ^ 2) + e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) par[e] = acc[0][4 * (g ^ 1) + e];
        }
        const float cv[4] = {c.x, c.y, c.z, c.w}, sv[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (own[e] * cv[e] + (lo ? par[e] : -par[e]) * sv[e]) * scale;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[d][4 * g + e] * scale;
      }
      u32x2 w; w.x = pack2bf(o[0], o[1]); w.y = pack2bf(o[2], o[3]);
      *reinterpret_cast<u32x2*>(dst + dd) = w;
    }
}

// ---- dQ: one wave = 32 queries, loops over all keys -------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(256, 2) k_attn_bwd_dq(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                     const float* __restrict__ key_bias, const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx,
                                                     const float* __restrict__ lse, float* __restrict__ delta, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                     float q_scale, bf16_t* __restrict__ dqkv, int B, int H, int L, int nqb) {
  typedef Cfg<HD> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = sK + KC * C::ROWB;
  u32x4* sE = reinterpret_cast<u32x4*>(sV + KC * C::ROWB);       // per key: bf16 [1, 1, 1, bias, 0, 0, 0, 0]; slot KC = zeros (as in the forward)
  int bh, qb;
  decode_block(nqb, B * H, bh, qb);
  if (bh >= B * H) return;
  const int b = bh / H, head = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int qidx = qb * 128 + wave * 32 + (lane & 31);
  const int qrow = qidx < L ? qidx : L - 1;
  const int dm = H * HD;
  const bf16_t* kbase = k + (size_t)bh * L * HD;
  const bf16_t* vbase = v + (size_t)bh * L * HD;
  bf8_t qf[C::KSTEPS], dof[C::KSTEPS];
#pragma unroll
  for (int st = 0; st < C::KSTEPS; ++st) {
    qf[st] = *reinterpret_cast<const bf8_t*>(q + ((size_t)bh * L + qrow) * HD + 16 * st + 8 * h);
    dof[st] = *reinterpret_cast<const bf8_t*>(dctx + ((size_t)b * L + qrow) * dm + head * HD + 16 * st + 8 * h);
  }
  const float lse_q = lse[(size_t)bh * L + qrow] * LOG2E;      // scores are in log2 units (q stored x log2 e)
  // delta[query] = sum_d dO[query, d] * O[query, d]: each lane holds 8 of every 16 head-dim columns of its query's dO; formed here (one
  // lane^32 exchange) and published for the dK/dV kernel, which runs after this one on the same stream
  float delta_q = 0.f;
#pragma unroll
  for (int st = 0; st < C::KSTEPS; ++st) {
    const u32x4 x = *reinterpret_cast<const u32x4*>(ctx + ((size_t)b * L + qrow) * dm + head * HD + 16 * st + 8 * h), y = __builtin_bit_cast(u32x4, dof[st]);
    delta_q += bflo(x.x) * bflo(y.x) + bfhi(x.x) * bfhi(y.x) + bflo(x.y) * bflo(y.y) + bfhi(x.y) * bfhi(y.y) + bflo(x.z) * bflo(y.z) + bfhi(x.z) * bfhi(y.z) +
               bflo(x.w) * bflo(y.w) + bfhi(x.w) * bfhi(y.w);
  }
  delta_q += __shfl_xor(delta_q, 32, 64);
  if (h == 0 && qidx < L) delta[(size_t)bh * L + qidx] = delta_q;
  // row constants through one extra MFMA k-step each (no per-element VALU): S' = K Q^T + [1,1,1,bias_key] . [-lse split, 1],
  // dP' = V dO^T + [1,1,1,0] . [-delta split, 0]
  u32x4 qe = {0u, 0u, 0u, 0u}, de = {0u, 0u, 0u, 0u}, ones3 = {0u, 0u, 0u, 0u};
  if (h == 0) {
    unsigned w01, w2;
    split3_bf16(-lse_q, w01, w2); qe.x = w01; qe.y = w2 | 0x3F800000u;
    split3_bf16(-delta_q, w01, w2); de.x = w01; de.y = w2;
    ones3.x = 0x3F803F80u; ones3.y = 0x00003F80u;
  }
  f32x16 acc[C::DBLK];
#pragma unroll
  for (int d = 0; d < C::DBLK; ++d) acc[d] = zero16();
  for (int kc0 = 0; kc0 < L; kc0 += KC) {
    const int nkeys = min(KC, L - kc0);
    const int nrows = (nkeys + 31) & ~31;
    __syncthreads();
    load_tile_pair<HD>(sK, kbase + (size_t)kc0 * HD, HD, sV, vbase + (size_t)kc0 * HD, HD, nkeys, nrows);
    for (int i = threadIdx.x; i <= KC; i += 256) {
      u32x4 e = {0u, 0u, 0u, 0u};
      if (i < nrows) {
        const float bv = i < nkeys ? (key_bias ? key_bias[(size_t)b * L + kc0 + i] : 0.f) : -INFINITY;
        e.x = 0x3F803F80u; e.y = 0x3F80u | (pack2bf(bv, 0.f) << 16);
      }
      sE[i] = e;
    }
    __syncthreads();
    for (int t = 0; t < nrows / 32; ++t) {
      const u32x4 ke = sE[h ? KC : t * 32 + (lane & 31)];
      f32x16 s = MFMA32(__builtin_bit_cast(bf8_t, ke), __builtin_bit_cast(bf8_t, qe), zero16());          // bias[key] - lse[query]
      f32x16 dp = MFMA32(__builtin_bit_cast(bf8_t, ones3), __builtin_bit_cast(bf8_t, de), zero16());      // -delta[query]
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) {
        s = MFMA32(rd_row<HD>(sK, t * 32 + (lane & 31), st, h), qf[st], s);
        dp = MFMA32(rd_row<HD>(sV, t * 32 + (lane & 31), st, h), dof[st], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]) * dp[r];          // dS^T = P * (dP - delta)
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const bf8_t dsf = pack8(s, sb);
#pragma unroll
        for (int d = 0; d < C::DBLK; ++d) acc[d] = MFMA32(rd_tr<HD>(sK, t * 32, sb, d, lane), dsf, acc[d]);
      }
    }
  }
  if (qidx < L) unrope_store<HD>(acc, cosT, sinT, qidx, h, q_scale, cosT != nullptr, dqkv + ((size_t)b * L + qidx) * (3 * dm) + head * HD);
}

// ---- dK, dV: one wave = 32 keys, loops over all queries -----------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(256, 2) k_attn_bwd_dkv(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                      const float* __restrict__ key_bias, const bf16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                      const float* __restrict__ delta, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                      bf16_t* __restrict__ dqkv, int B, int H, int L, int nkb) {
  typedef Cfg<HD> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;
  unsigned char* sdO = sQ + KC * C::ROWB;
  u32x4* sQE = reinterpret_cast<u32x4*>(sdO + KC * C::ROWB);    // per query: bf16 [-lse split in 3, 1, -delta split in 3, 0]; slot KC = zeros
  int bh, kb;
  decode_block(nkb, B * H, bh, kb);
  if (bh >= B * H) return;
  const int b = bh / H, head = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int kidx = kb * 128 + wave * 32 + (lane & 31);
  const int krow = kidx < L ? kidx : L - 1;
  const int dm = H * HD;
  bf8_t kf[C::KSTEPS], vf[C::KSTEPS];
#pragma unroll
  for (int st = 0; st < C::KSTEPS; ++st) {
    kf[st] = *reinterpret_cast<const bf8_t*>(k + ((size_t)bh * L + krow) * HD + 16 * st + 8 * h);
    vf[st] = *reinterpret_cast<const bf8_t*>(v + ((size_t)bh * L + krow) * HD + 16 * st + 8 * h);
  }
  const float bias_k = kidx < L ? (key_bias ? key_bias[(size_t)b * L + kidx] : 0.f) : -INFINITY;
  // key-side operands of the extra k-step: [1,1,1,bias_key,0,0,0,0] picks (-lse + bias) for S, [0,0,0,0,1,1,1,0] picks -delta for dP
  u32x4 kS = {0u, 0u, 0u, 0u}, kD = {0u, 0u, 0u, 0u};
  if (h == 0) { kS.x = 0x3F803F80u; kS.y = 0x3F80u | (pack2bf(bias_k, 0.f) << 16); kD.z = 0x3F803F80u; kD.w = 0x00003F80u; }
  f32x16 adk[C::DBLK], adv[C::DBLK];
#pragma unroll
  for (int d = 0; d < C::DBLK; ++d) { adk[d] = zero16(); adv[d] = zero16(); }
  for (int qc0 = 0; qc0 < L; qc0 += KC) {
    const int nq = min(KC, L - qc0);
    const int nrows = (nq + 31) & ~31;
    __syncthreads();
    load_tile_pair<HD>(sQ, q + ((size_t)bh * L + qc0) * HD, HD, sdO, dctx + ((size_t)b * L + qc0) * dm + head * HD, dm, nq, nrows);
    for (int i = threadIdx.x; i <= KC; i += 256) {
      u32x4 e = {0u, 0u, 0u, 0u};
      if (i < nrows) {
        unsigned w01, w2;
        split3_bf16(i < nq ? -lse[(size_t)bh * L + qc0 + i] * LOG2E : -1.0e30f, w01, w2);       // log2 units; finite for padding rows (-inf x 0 in the dP pick would be NaN)
        e.x = w01; e.y = w2 | 0x3F800000u;
        split3_bf16(i < nq ? -delta[(size_t)bh * L + qc0 + i] : 0.f, w01, w2);
        e.z = w01; e.w = w2;
      }
      sQE[i] = e;
    }
    __syncthreads();
    for (int t = 0; t < nrows / 32; ++t) {
      const bf8_t qe_row = __builtin_bit_cast(bf8_t, sQE[h ? KC : t * 32 + (lane & 31)]);
      f32x16 s = MFMA32(qe_row, __builtin_bit_cast(bf8_t, kS), zero16());            // bias[key] - lse[query]
      f32x16 dp = MFMA32(qe_row, __builtin_bit_cast(bf8_t, kD), zero16());           // -delta[query]
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) {
        s = MFMA32(rd_row<HD>(sQ, t * 32 + (lane & 31), st, h), kf[st], s);          // S[query][key] + bias - lse
        dp = MFMA32(rd_row<HD>(sdO, t * 32 + (lane & 31), st, h), vf[st], dp);       // dP[query][key] - delta
      }
      f32x16 p;
#pragma unroll
      for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s[r]); s[r] = p[r] * dp[r]; }      // P, dS
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const bf8_t pf = pack8(p, sb), dsf = pack8(s, sb);
#pragma unroll
        for (int d = 0; d < C::DBLK; ++d) {
          adv[d] = MFMA32(rd_tr<HD>(sdO, t * 32, sb, d, lane), pf, adv[d]);           // dV^T += dO^T P
          adk[d] = MFMA32(rd_tr<HD>(sQ, t * 32, sb, d, lane), dsf, adk[d]);           // dK^T += Q^T dS
        }
      }
    }
  }
  if (kidx < L) {
    bf16_t* row = dqkv + ((size_t)b * L + kidx) * (3 * dm) + head * HD;
    unrope_store<HD>(adk, cosT, sinT, kidx, h, 0.6931471805599453f, cosT != nullptr, row + dm);      // q is stored x log2(e): dK = ln2 * dS^T q
    unrope_store<HD>(adv, cosT, sinT, kidx, h, 1.0f, false, row + 2 * dm);
  }
}

// ---- fused backward for short sequences (L <= 512, hd <= 32): one 1024-thread work-group per (b, h) ---------------------------------
// Wave w owns keys [32 w, 32 w + 32): K_j / V_j fragments and the dK_j / dV_j accumulators stay in its registers for the whole launch, so S, P,
// dP and dS of every (query block, key block) tile are formed ONCE (the split kernels above form them twice: 2 x 16 exponentials, 8 extra
// MFMAs per tile).  All Q and dO rows of the (b, h) slab, and the [-lse | -delta] row constants, are put in LDS once, behind the only barrier
// of the kernel.  dQ needs the contraction over keys, which sit on the lanes here: the wave transposes its dS tile through a private 2 KB LDS
// slab (4 ds_write_b64, 4 ds_read_b64_tr_b16), forms dQ_i^T += K_j^T dS^T with two MFMAs and adds the 32 x 32 fp32 result into block i's
// accumulator in LDS (8 block buffers, rows of 36 floats: conflict-free 16-byte accesses).  The waves walk the query blocks from different
// starting blocks (blocks 0-7, then 8-15 in the same buffers) and never meet at a barrier, so their MFMA and exponential phases stay
// interleaved; the read-add-write on a block is serialised by a per-buffer ticket counter in LDS: wave w's visit to block i at its step t has
// ticket (visits to i by earlier steps) + w / nblk, so every block receives its contributions in one fixed order -- deterministic like the
// split kernels, no float atomics (ds_add_f32 measured ~6 x slower for the whole launch than this read-add-write).  The first visitor writes
// instead of adding; the last one keeps the sums in registers, applies the inverse rotary + q-scale and stores the block's dQ rows.
#define FQB 8                                          // dQ block buffers (query blocks of 32 in flight)
template <int HD> struct FusedLds {
  static constexpr int ROWS = 512;
  static constexpr int TILE = ROWS * 64;                // all Q (or dO) rows, 64-byte swizzled rows (hd = 16 zero-padded)
  static constexpr int QE = (ROWS + 1) * 16 + 48;       // row constants + FQB ticket counters
  static constexpr int DQP = 36;                        // dQ row pitch in floats: 16-byte aligned rows, conflict-free 16-byte accesses
  static constexpr int DQ = FQB * 32 * DQP * 4;
  static constexpr int SLAB = 32 * 64;
  static constexpr int TOTAL = 2 * TILE + QE + DQ + 16 * SLAB;
};
typedef __attribute__((address_space(3))) volatile int lds_vint;      // (a generic volatile pointer polls with flat loads, which also wait for the wave's global stores)
__device__ __forceinline__ float dot8_bf16(const u32x4 x, const u32x4 y) {
  return (bflo(x.x) * bflo(y.x) + bfhi(x.x) * bfhi(y.x)) + (bflo(x.y) * bflo(y.y) + bfhi(x.y) * bfhi(y.y)) + (bflo(x.z) * bflo(y.z) + bfhi(x.z) * bfhi(y.z)) +
         (bflo(x.w) * bflo(y.w) + bfhi(x.w) * bfhi(y.w));
}

template <int HD>
__global__ void __launch_bounds__(1024, 1) k_attn_bwd_fused(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                         const float* __restrict__ key_bias, const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx,
                                                         const float* __restrict__ lse, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                         float q_scale, bf16_t* __restrict__ dqkv, int B, int H, int L, int abl) {
  typedef Cfg<HD> C;
  typedef FusedLds<HD> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;
  unsigned char* sdO = sQ + F::TILE;
  u32x4* sQE = reinterpret_cast<u32x4*>(sdO + F::TILE);       // per query: bf16 [-lse split in 3, 1, -delta split in 3, 0]; slot ROWS = zeros
  lds_vint* sTicket = (lds_vint*)LDS_PTR(sdO + F::TILE + (F::ROWS + 1) * 16);
  float* sdQ = reinterpret_cast<float*>(sdO + F::TILE + F::QE);   // [buffer][32 queries][36] fp32
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, r = lane & 31;
  unsigned char* sT = reinterpret_cast<unsigned char*>(sdQ) + F::DQ + wave * F::SLAB;      // this wave's transpose slab
  const int dm = H * HD;
  // consecutive work-groups go to consecutive XCDs: XCD x takes the x-th eighth of the (b, h) slabs, so the heads of one batch element --
  // which share the 128-byte lines of dctx / ctx rows and of the dqkv rows they write 64 bytes of -- meet in one L2
  const int nbh = B * H, per_xcd = (nbh + 7) >> 3;
  const int bh = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (bh >= nbh) return;
  const int b = bh / H, head = bh - b * H;
  const int key0 = wave * 32;
  const bool active = key0 < L;
  const int kidx = key0 + r;

  // the wave's own K_j / V_j rows and key bias are requested first, the (b, h) slab's rows right behind them: one exposed round trip
  u32x4 kr[2], vr[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
    const u32x4 z = {0u, 0u, 0u, 0u};
    kr[it] = z; vr[it] = z;
    if (key0 + row < L && ch < HD / 8) {
      kr[it] = *reinterpret_cast<const u32x4*>(k + ((size_t)bh * L + key0 + row) * HD + ch * 8);
      vr[it] = *reinterpret_cast<const u32x4*>(v + ((size_t)bh * L + key0 + row) * HD + ch * 8);
    }
  }
  const float bias_k = kidx < L ? (key_bias ? key_bias[(size_t)b * L + kidx] : 0.f) : -INFINITY;
  // ---- the slab's rows: Q, dO images and the row constants (delta = rowsum(dO * O)), two (row, 16-byte chunk) items per thread
  {
    const bf16_t* qrow0 = q + (size_t)bh * L * HD;
    const bf16_t* dorow0 = dctx + (size_t)b * L * dm + head * HD;
    const bf16_t* orow0 = ctx + (size_t)b * L * dm + head * HD;
    u32x4 rq[2], rdo[2], ro[2];
    float rl[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = threadIdx.x + 1024 * it, row = idx >> 2, ch = idx & 3;
      const u32x4 z = {0u, 0u, 0u, 0u};
      rq[it] = z; rdo[it] = z; ro[it] = z; rl[it] = 0.f;
      if (row < L) {
        if (ch < HD / 8) {
          rq[it] = *reinterpret_cast<const u32x4*>(qrow0 + (size_t)row * HD + ch * 8);
          rdo[it] = *reinterpret_cast<const u32x4*>(dorow0 + (size_t)row * dm + ch * 8);
          ro[it] = *reinterpret_cast<const u32x4*>(orow0 + (size_t)row * dm + ch * 8);
        }
        if (ch == 0) rl[it] = lse[(size_t)bh * L + row];
      }
    }
    const int nrows = (L + 31) & ~31;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = threadIdx.x + 1024 * it, row = idx >> 2, ch = idx & 3;
      float part = dot8_bf16(rdo[it], ro[it]);
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      if (row < nrows) {
        const int off = row * 64 + (swz<32>(row, ch) << 4);
        *reinterpret_cast<u32x4*>(sQ + off) = rq[it];
        *reinterpret_cast<u32x4*>(sdO + off) = rdo[it];
        if (ch == 0) {
          u32x4 e;
          unsigned w01, w2;
          split3_bf16(row < L ? -rl[it] * LOG2E : -1.0e30f, w01, w2);      // log2 units; finite for padding rows (-inf x 0 in the dP pick would be NaN)
          e.x = w01; e.y = w2 | 0x3F800000u;
          split3_bf16(row < L ? -part : 0.f, w01, w2);
          e.z = w01; e.w = w2;
          sQE[row] = e;
        }
      }
    }
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (threadIdx.x == 0) sQE[F::ROWS] = z;
    if (threadIdx.x < FQB) sTicket[threadIdx.x] = 0;
  }

  // ---- the wave's own K_j / V_j through its slab: row fragments, and K_j^T fragments for dQ
  bf8_t kf[C::KSTEPS], vf[C::KSTEPS], ktf[2];
  {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
      *reinterpret_cast<u32x4*>(sT + row * 64 + (swz<32>(row, ch) << 4)) = kr[it];
    }
#pragma unroll
    for (int st = 0; st < C::KSTEPS; ++st) kf[st] = rd_row<HD>(sT, r, st, h);
    ktf[0] = rd_tr<HD>(sT, 0, 0, 0, lane);
    ktf[1] = rd_tr<HD>(sT, 0, 1, 0, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
      *reinterpret_cast<u32x4*>(sT + row * 64 + (swz<32>(row, ch) << 4)) = vr[it];
    }
#pragma unroll
    for (int st = 0; st < C::KSTEPS; ++st) vf[st] = rd_row<HD>(sT, r, st, h);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  u32x4 kS = {0u, 0u, 0u, 0u}, kD = {0u, 0u, 0u, 0u};
  if (h == 0) { kS.x = 0x3F803F80u; kS.y = 0x3F80u | (pack2bf(bias_k, 0.f) << 16); kD.z = 0x3F803F80u; kD.w = 0x00003F80u; }
  f32x16 adk = zero16(), adv = zero16();

  // per-lane byte offsets inside a 32-row block of a swizzled 64-byte-row image
  const int sw = (r >> 2) & 3;
  int row_off[C::KSTEPS];
#pragma unroll
  for (int st = 0; st < C::KSTEPS; ++st) row_off[st] = r * 64 + (((2 * st + h) ^ sw) << 4);
  int tr_off[2][2];
  {
    const int g = lane >> 4, i = lane & 15;
    const int col = 16 * (g & 1) + 4 * (i & 3);
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int r0 = 16 * sb + 4 * h + (i >> 2), r1 = r0 + 8;
      tr_off[sb][0] = r0 * 64 + (swz<32>(r0, col >> 3) << 4) + (col & 7) * 2;
      tr_off[sb][1] = r1 * 64 + (swz<32>(r1, col >> 3) << 4) + (col & 7) * 2;
    }
  }
  const int wr_off = r * 64 + 8 * h;        // + ((g ^ sw) << 4) for query group g: row = key, 4 consecutive queries (8 g + 4 h ...)
  auto tr_frag = [&](const unsigned char* base, int sb) -> bf8_t {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_off[sb][0]));
    const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_off[sb][1]));
    s16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = c[0]; o[5] = c[1]; o[6] = c[2]; o[7] = c[3];
    return __builtin_bit_cast(bf8_t, o);
  };
  __syncthreads();

  if (active) {
    const int nwaves = (L + 31) >> 5;                       // waves that own keys = query blocks = visits per block
    int tbase = 0;                                          // tickets already used on a buffer by the earlier group of blocks
    for (int blk0 = 0; blk0 < nwaves; blk0 += FQB) {
      const int nblk = min(FQB, nwaves - blk0);
      auto visits = [&](int blk) { return blk < nwaves ? (nwaves - blk + nblk - 1) / nblk : 0; };      // waves that start their walk at block blk
      int i = wave % nblk;
      int ticket = wave / nblk;
      for (int t = 0; t < ((abl & 4) ? 0 : nblk); ++t) {
        const int qblk = blk0 + i;
        const unsigned char* tQ = sQ + qblk * 2048;
        const unsigned char* tdO = sdO + qblk * 2048;
        const bf8_t qe_row = __builtin_bit_cast(bf8_t, sQE[h ? F::ROWS : qblk * 32 + r]);
        f32x16 s = MFMA32(qe_row, __builtin_bit_cast(bf8_t, kS), zero16());            // bias[key] - lse[query]
        f32x16 dp = MFMA32(qe_row, __builtin_bit_cast(bf8_t, kD), zero16());           // -delta[query]
#pragma unroll
        for (int st = 0; st < C::KSTEPS; ++st) {
          s = MFMA32(*reinterpret_cast<const bf8_t*>(tQ + row_off[st]), kf[st], s);          // S[query][key] + bias - lse
          dp = MFMA32(*reinterpret_cast<const bf8_t*>(tdO + row_off[st]), vf[st], dp);       // dP[query][key] - delta
        }
        f32x16 p;
#pragma unroll
        for (int e = 0; e < 16; ++e) { p[e] = __builtin_amdgcn_exp2f(s[e]); s[e] = p[e] * dp[e]; }      // P, dS
        bf8_t dsf[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          const bf8_t pf = pack8(p, sb);
          dsf[sb] = pack8(s, sb);
          adv = MFMA32(tr_frag(tdO, sb), pf, adv);           // dV^T += dO^T P
          adk = MFMA32(tr_frag(tQ, sb), dsf[sb], adk);       // dK^T += Q^T dS
        }
        // dS^T through the wave's slab: lane (key, h) holds queries 8 g + 4 h .. + 3 of group g as one 8-byte piece
        {
          const u32x4 w0 = __builtin_bit_cast(u32x4, dsf[0]), w1 = __builtin_bit_cast(u32x4, dsf[1]);
          u32x2 g0 = {w0.x, w0.y}, g1 = {w0.z, w0.w}, g2 = {w1.x, w1.y}, g3 = {w1.z, w1.w};
          *reinterpret_cast<u32x2*>(sT + wr_off + ((0 ^ sw) << 4)) = g0;
          *reinterpret_cast<u32x2*>(sT + wr_off + ((1 ^ sw) << 4)) = g1;
          *reinterpret_cast<u32x2*>(sT + wr_off + ((2 ^ sw) << 4)) = g2;
          *reinterpret_cast<u32x2*>(sT + wr_off + ((3 ^ sw) << 4)) = g3;
        }
        asm volatile("" ::: "memory");
        f32x16 dq = MFMA32(ktf[0], tr_frag(sT, 0), zero16());       // dQ^T[d][query] = K^T dS^T
        dq = MFMA32(ktf[1], tr_frag(sT, 1), dq);
        // this wave's turn on the block's buffer (every wave reaches every ticket it waits for: a visit only waits for visits of earlier or
        // equal steps, and for the earlier group of blocks to have left the buffer)
        const int turn = tbase + ticket;
        if (lane == 0 && !(abl & 1))
          while (sTicket[i] != turn) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        float4* drow = reinterpret_cast<float4*>(sdQ + (i * 32 + r) * F::DQP + 4 * h);
        if (ticket != 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 o = drow[2 * g];
            dq[4 * g] += o.x; dq[4 * g + 1] += o.y; dq[4 * g + 2] += o.z; dq[4 * g + 3] += o.w;
          }
        }
        if (ticket != nwaves - 1) {
#pragma unroll
          for (int g = 0; g < 4; ++g) drow[2 * g] = make_float4(dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]);
        }
        asm volatile("" ::: "memory");
        if (lane == 0) sTicket[i] = turn + 1;          // LDS operations of a wave complete in order: the sums land before the counter moves
        if (ticket == nwaves - 1) {                    // last visitor: the block's dQ rows leave from registers
          const int qidx = qblk * 32 + r;
          f32x16 a0[1] = {dq};
          if (qidx < L && !(abl & 16)) unrope_store<HD>(a0, cosT, sinT, qidx, h, q_scale, cosT != nullptr, dqkv + ((size_t)b * L + qidx) * (3 * dm) + head * HD);
        }
        i = i + 1 == nblk ? 0 : i + 1;
        ticket += visits(i);                             // visits to the next block by all earlier steps (+ this wave's rank among the waves of its own step)
      }
      tbase += nwaves;
    }
  }
  // dK, dV: inverse rotary in registers, then through the wave's slab ([key][64 B dK | 64 B dV]... two 2 KB images one after the other) so
  // that every key's row leaves as whole 16-byte chunks, 64 contiguous bytes per 4 lanes
  if (active && !(abl & 32)) {
    f32x16 a1[1] = {adk}, a2[1] = {adv};
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      // unrope_store writes 8-byte pieces at dst + dd (dd = 8 g + 4 h): aimed at the lane's slab row it builds the row-major bf16 image
      bf16_t* img = reinterpret_cast<bf16_t*>(sT + r * 64);
      if (which == 0) unrope_store<HD>(a1, cosT, sinT, kidx < L ? kidx : 0, h, 0.6931471805599453f, cosT != nullptr, img);      // q is stored x log2(e): dK = ln2 * dS^T q
      else unrope_store<HD>(a2, cosT, sinT, 0, h, 1.0f, false, img);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
        const u32x4 w = *reinterpret_cast<const u32x4*>(sT + row * 64 + ch * 16);
        if (key0 + row < L && ch < HD / 8)
          *reinterpret_cast<u32x4*>(dqkv + ((size_t)b * L + key0 + row) * (3 * dm) + (which + 1) * dm + head * HD + ch * 8) = w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

static int g_attn_bwd_path = -1;      // -1 automatic, 0 split kernels, 1 fused where eligible (A/B runs and tests)
extern "C" void oneprot_attn_force_bwd_path(int path) { g_attn_bwd_path = path < 0 ? -1 : (path ? 1 : 0); }
#ifdef ONEPROT_ATTN_ABLATE
static int g_attn_bwd_ablate = 0;      // timing builds only (tools/attn_only.py with a library built -DONEPROT_ATTN_ABLATE): skips parts of the fused kernel, results are wrong by construction
extern "C" void oneprot_attn_debug_ablate(int mask) { g_attn_bwd_ablate = mask; }
#else
static constexpr int g_attn_bwd_ablate = 0;
#endif

template <int HD>
static int launch_bwd(const void* q, const void* k, const void* v, const float* key_bias, const void* ctx, const void* dctx, const float* lse, float* delta,
                      const float* cosT, const float* sinT, float q_scale, void* dqkv, int B, int H, int L, hipStream_t s) {
  if constexpr (HD <= 32) {
    if (L <= 512 && g_attn_bwd_path != 0) {
      // the fused kernel needs the 143 KB dynamic-LDS opt-in; a device / driver that refuses it takes the split kernels from then on (forced
      // fused path: the refusal is the caller's error)
      static int fused_ok = -1;
      if (fused_ok < 0)
        fused_ok = hipFuncSetAttribute((const void*)k_attn_bwd_fused<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, FusedLds<HD>::TOTAL) == hipSuccess ? 1 : 0;
      if (!fused_ok) { (void)hipGetLastError(); if (g_attn_bwd_path > 0) return OP_EINVAL; }
      else {
      hipLaunchKernelGGL(k_attn_bwd_fused<HD>, dim3(((B * H + 7) / 8) * 8), dim3(1024), FusedLds<HD>::TOTAL, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                         (const bf16_t*)ctx, (const bf16_t*)dctx, lse, cosT, sinT, q_scale, (bf16_t*)dqkv, B, H, L, g_attn_bwd_ablate);
      return launch_status();
      }
    }
  }
  const int nb = (L + 127) / 128;
  const int nbh8 = ((B * H + 7) / 8) * 8;
  const size_t lds_q = (size_t)2 * KC * Cfg<HD>::ROWB + (KC + 1) * 16;
  const size_t lds_kv = (size_t)2 * KC * Cfg<HD>::ROWB + (KC + 1) * 16;
  hipLaunchKernelGGL(k_attn_bwd_dq<HD>, dim3(nbh8 * nb), dim3(256), lds_q, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (const bf16_t*)ctx, (const bf16_t*)dctx, lse, delta, cosT, sinT, q_scale, (bf16_t*)dqkv, B, H, L, nb);
  hipLaunchKernelGGL(k_attn_bwd_dkv<HD>, dim3(nbh8 * nb), dim3(256), lds_kv, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (const bf16_t*)dctx, lse, (const float*)delta, cosT, sinT, (bf16_t*)dqkv, B, H, L, nb);
  return launch_status();
}

extern "C" size_t oneprot_attn_bwd_workspace(int B, int H, int L) { return (size_t)B * H * L * sizeof(float); }

extern "C" int oneprot_attn_bwd(const void* q, const void* k, const void* v, const float* key_bias, const void* ctx, const void* dctx, const float* lse,
                                const float* rope_cos, const float* rope_sin, float q_scale, void* dqkv, void* workspace, int B, int H, int L, int hd,
                                void* stream) {
  if (!q || !k || !v || !ctx || !dctx || !lse || !dqkv || !workspace || B <= 0 || H <= 0 || L <= 0) return OP_EINVAL;
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return OP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)workspace;
  switch (hd) {
    case 16: return launch_bwd<16>(q, k, v, key_bias, ctx, dctx, lse, delta, rope_cos, rope_sin, q_scale, dqkv, B, H, L, s);
    case 32: return launch_bwd<32>(q, k, v, key_bias, ctx, dctx, lse, delta, rope_cos, rope_sin, q_scale, dqkv, B, H, L, s);
    case 64: return launch_bwd<64>(q, k, v, key_bias, ctx, dctx, lse, delta, rope_cos, rope_sin, q_scale, dqkv, B, H, L, s);
    default: return OP_EINVAL;
  }
}
