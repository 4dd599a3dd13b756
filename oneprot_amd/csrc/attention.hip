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
#include "sched_ws.h"
#include "../../include/oneprot_hip.h"
#include <float.h>
#include <atomic>
void* dynamic_tiles_workspace();      // gemm_nt8.hip: what oneprot_dynamic_tiles was given (null: static work lists)

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
// DROP: attention-probability dropout (hf modeling_bert.py BertSelfAttention: softmax -> dropout -> @ V; active in the reference whenever the text
// tower is in train mode, text_encoder.py:59).  keep(b, h, q, k) is a pure function of (seed, stream, b*H+h, q, k), one 32-bit integer hash per
// ELEMENT (lowbias32 finaliser over a multiplicative mix of the indices), so that the forward (a lane holds 1 query x 16 keys), the dQ kernel
// (the same) and the dK / dV kernel (a lane holds 16 queries x 1 key) regenerate the same mask from their own layouts; kept iff the upper 16 bits
// >= thr16 = round(p * 65536), kept probabilities scaled by 65536 / (65536 - thr16).  The row sum, the log-sum-exp and delta = rowsum(dO * O) are
// untouched by the mask.  oneprot_attn_dropout_keep writes the mask out for tests.  (The hidden-state dropouts use Philox, featops.hip; here a
// Philox call per element would cost ten times the tile's own vector work.)
struct AttnDrop { unsigned thr16, s0, s1; float scale; };
__device__ __forceinline__ bool attn_keep(unsigned qi, unsigned ki, unsigned bh, const AttnDrop& dr) {
  unsigned x = (qi * 0x9E3779B1u) ^ (ki * 0x85EBCA77u + dr.s0) ^ (bh * 0xC2B2AE3Du + dr.s1);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (x >> 16) >= dr.thr16;
}
// the 16 elements of a 32 x 32 accumulator tile a lane holds: index 8 * (e >> 2) + 4 * h + (e & 3) along the tile's row axis
__device__ __forceinline__ unsigned attn_keep_bits_q(int qidx, int key0, int h, int bh, const AttnDrop& dr) {      // lane = query, rows = keys
  unsigned bits = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) bits |= (attn_keep((unsigned)qidx, (unsigned)(key0 + 8 * (e >> 2) + 4 * h + (e & 3)), (unsigned)bh, dr) ? 1u : 0u) << e;
  return bits;
}
__device__ __forceinline__ unsigned attn_keep_bits_k(int kidx, int q0, int h, int bh, const AttnDrop& dr) {        // lane = key, rows = queries
  unsigned bits = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) bits |= (attn_keep((unsigned)(q0 + 8 * (e >> 2) + 4 * h + (e & 3)), (unsigned)kidx, (unsigned)bh, dr) ? 1u : 0u) << e;
  return bits;
}
static int attn_drop_make(float p, uint64_t seed, uint64_t stream_id, AttnDrop& dr) {
  if (!(p >= 0.f) || !(p < 1.f)) return OP_EINVAL;
  dr.thr16 = (unsigned)(p * 65536.f + 0.5f);
  if (dr.thr16 >= 65536u) return OP_EINVAL;
  dr.scale = 65536.f / (float)(65536u - dr.thr16);
  const uint64_t m = (seed ^ (stream_id * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
  dr.s0 = (unsigned)m; dr.s1 = (unsigned)(m >> 32) ^ (unsigned)(stream_id * 0x2545F491u);
  return OP_OK;
}

template <int HD, bool DROP = false>
__global__ void __launch_bounds__(256, 2) k_attn_fwd(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                  const float* __restrict__ key_bias, bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int B, int H,
                                                  int L, int nqb, const AttnDrop dr = AttnDrop{0u, 0u, 0u, 1.0f}) {
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
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-dlt);      // (nothing accumulated yet on the first tile; 2^-dlt may be inf there)
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
      f32x16 sd = s;
      if constexpr (DROP) {
        const unsigned bits = attn_keep_bits_q(qidx, kc0 + t * 32, h, bh, dr);
#pragma unroll
        for (int r = 0; r < 16; ++r) sd[r] = ((bits >> r) & 1u) ? s[r] : 0.f;
      }
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const bf8_t pf = pack8(s, sb);
        lacc = MFMA32(__builtin_bit_cast(bf8_t, ones), pf, lacc);
        const bf8_t pv = DROP ? pack8(sd, sb) : pf;
#pragma unroll
        for (int d = 0; d < C::DBLK; ++d) acc[d] = MFMA32(rd_tr<HD>(sV, t * 32, sb, d, lane), pv, acc[d]);
      }
    }
  }
  const float lt = l + lacc[0];
  float inv = lt > 0.f ? 1.0f / lt : 0.f;
  if constexpr (DROP) inv *= dr.scale;
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

template <int HD>
static int launch_fwd_dropout(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, const AttnDrop& dr,
                              hipStream_t s) {
  const int nqb = (L + 127) / 128;
  const int nbh8 = ((B * H + 7) / 8) * 8;
  hipLaunchKernelGGL((k_attn_fwd<HD, true>), dim3(nbh8 * nqb), dim3(256), fwd_lds<HD>(), s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (bf16_t*)ctx, lse, B, H, L, nqb, dr);
  return launch_status();
}
// keep[b][h][q][k] (one byte each, 0 / 1) of the mask the DROP kernels apply: for tests and for an oracle that is handed the mask
__global__ void __launch_bounds__(256) k_attn_dropout_keep(unsigned char* __restrict__ keep, int BH, int L, const AttnDrop dr) {
  const size_t n = (size_t)BH * L * L;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned kk = (unsigned)(i % L), qq = (unsigned)((i / L) % L), bh = (unsigned)(i / ((size_t)L * L));
    keep[i] = attn_keep(qq, kk, bh, dr) ? 1 : 0;
  }
}
extern "C" int oneprot_attn_dropout_keep(void* keep, int B, int H, int L, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  AttnDrop dr;
  if (!keep || B <= 0 || H <= 0 || L <= 0 || attn_drop_make(p, seed, stream_id, dr) != OP_OK) return OP_EINVAL;
  const size_t n = (size_t)B * H * L * L;
  size_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_attn_dropout_keep, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)keep, B * H, L, dr);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// forward, round 4: one pass over the keys WITHOUT a row maximum, K/V of a whole (b, h) staged once.
//
// The per-tile work of the kernel above is bounded by the vector pipe and by the length of each wave's dependent chain, not by the MFMAs: 16
// v_exp_f32 (measured 10.6 SIMD-cycles each, profiles/r04_exp_rate_probe.txt) + 8 packs are the arithmetic that has to happen; the 14 v_max, the
// lane^32 exchange through LDS, the compare-and-branch and the bookkeeping MFMA only serve the running maximum.  exp2 needs no maximum for
// PRECISION (floating point: the relative error of p = 2^s is the same at every magnitude, and P is rounded to bf16, which has the fp32 exponent
// range); the maximum is only there against overflow / underflow.  So the fast pass computes p = 2^s as it stands and looks at the row sums once,
// at the end: a sum within [2^-60, 2^60] proves that no exponential overflowed and that everything that was flushed to zero (< 2^-126) is below
// 2^-66 of the result.  Any other sum (inf, NaN, tiny: a row whose scores all sit below -60, a fully masked row) sends the rows to the exact
// pass = the algorithm of k_attn_fwd with the deferred running maximum.  Scores of a trained or randomly initialised encoder sit within a few
// units of zero, so the second pass is a correctness net, not a working point; tests/test_kernels_gpu.py drives it with scores of +-400.
// Key tiles whose 32 keys are all padding (bias <= -1e30) are skipped (their P is exactly 0), tiles without any bias take no bookkeeping k-step.
//
// Two kernels share the tile steps:
//   k_attn_fwd3  (hd <= 32, L <= 512: every shipped configuration) -- ONE persistent 8-wave work-group per CU walks the (b, h) slabs; the
//                K / V / bias images of slab n+1 arrive by LDS-DMA (global_load_lds, no registers) in the second half of the LDS while slab n
//                is computed, the q fragments of slab n+1 are requested a slab ahead, the outputs of slab n leave after the barrier of slab
//                n+1: one barrier per slab and no exposed load, stage or store phase (stamps of the per-slab work-group form, k_attn_fwd2:
//                16 k of its 31 k cycles were entry / staging / bias / barrier / store phases, profiles/r04_fwd2_stamps.txt).  A wave owns
//                TWO 32-query blocks and runs them interleaved: the K / V fragment reads, their address arithmetic and the loop control are
//                shared, and each block's exponentials sit in the shadow of the other block's MFMA chain -- with one block per wave and four
//                waves per SIMD the hardware serves the oldest wave first and the younger ones only fill its stalls (profiles/r04_fwd3_stamps.txt:
//                waves 0-3 took 18 k cycles per slab, waves 12-15 34 k, 540 SIMD-cycles per tile against 273 for the same dataflow in registers,
//                profiles/r04_attn_mix_probe.txt);
//   k_attn_fwd2  (any hd, any L) -- one 8-wave work-group per 256 queries, keys staged through registers in chunks of 512 (256 at hd 64),
//                the exact pass repeated by the whole work-group (the chunk loop has barriers).
#define FWD2_BIG 1.0e18f            // ~2^60
#define FWD2_TINY 8.7e-19f          // ~2^-60

// Row sum of the probabilities.  It has to be the sum of the bf16-ROUNDED values that enter the P.V product: then the rounding errors of the weights
// cancel in acc / l (the output stays a convex combination); normalising by the fp32 sum of the unrounded values (16 v_add per tile, tried) leaves
// a common-mode error of ~2^-9 / sqrt(n) on every context row: the loss of the hd-32 reference fixture moved by 1.1e-3.  So the sum is taken by an
// all-ones MFMA over the packed fragments, as in k_attn_fwd: no vector instructions, 16 accumulator registers per block; same kernel time as the
// vector form (301 against 293 us, tools/ab/attn_ab.py).
// Row sum of one packed fragment, two forms.  MFMA: an all-ones MFMA per fragment (no vector instructions, a 16-register accumulator tile whose
// registers all hold the sum).  DOT2: v_dot2c_f32_bf16 against (1, 1), two ROUNDED probabilities per vector instruction, one register, no matrix-pipe
// time; the lane then holds the sum over ITS 16 keys of every 32 (the other 16 sit in lane ^ 32) and the halves meet once, in finish_sum.  The hd-64
// kernels are matrix-bound and short of registers and have taken DOT2 since round 5.  hd = 32: in kernel-only loops the two forms tie (round 5; round 6:
// -4.7 % without padding, +2.9 % ragged), but INSIDE the training step -- which runs at the board's power cap, DESIGN section 6 -- the form without the
// four all-ones MFMAs per key tile (16 K multiply-adds each, to add up 32 numbers) is worth 1.4 ms of a 215.6 ms step, four runs of four
// (tools/ab/step_libs_ab.sh, profiles/r06_step_ab.txt): DOT2 from hd = 32 up since round 6; hd = 16 (the padded 8M-model heads of cfg-1) keeps the MFMA form.
// (Inline asm: hipcc 7.2 compiles __builtin_amdgcn_fdot2_f32_bf16 on the four words of a fragment into four instructions that all read word 0.)
template <bool DOT2> struct RowSum;
template <> struct RowSum<false> {
  f32x16 lacc;
  __device__ __forceinline__ void zero() { lacc = zero16(); }
  __device__ __forceinline__ void add(const bf8_t& pf) {
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    lacc = MFMA32(__builtin_bit_cast(bf8_t, ones), pf, lacc);
  }
  __device__ __forceinline__ float take() { const float x = lacc[0]; lacc = zero16(); return x; }
  static __device__ __forceinline__ float whole(float l) { return l; }
};
template <> struct RowSum<true> {
  float lsum;
  __device__ __forceinline__ void zero() { lsum = 0.f; }
  __device__ __forceinline__ void add(const bf8_t& pf) {
    const u32x4 w = __builtin_bit_cast(u32x4, pf);
    const unsigned one = 0x3F803F80u;
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(lsum) : "s"(one), "v"(w.x));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(lsum) : "s"(one), "v"(w.y));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(lsum) : "s"(one), "v"(w.z));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(lsum) : "s"(one), "v"(w.w));
  }
  __device__ __forceinline__ float take() { const float x = lsum; lsum = 0.f; return x; }
  static __device__ __forceinline__ float whole(float l) { return l + __shfl_xor(l, 32, 64); }
};
#ifndef ROWSUM_DOT2_MIN_HD
#define ROWSUM_DOT2_MIN_HD 32
#endif
template <int HD> struct RowState {
  f32x16 acc[Cfg<HD>::DBLK];
  RowSum<(HD >= ROWSUM_DOT2_MIN_HD)> rs;      // sum over the keys so far of the lane's query since the last scale_sum / finish_sum
  float m, l;                       // (DOT2: l is a half sum until finish_sum)
  u32x4 qe;                         // exact pass, query side of the bookkeeping k-step: bf16 [-m split in three, 1, 0, 0, 0, 0] in the lanes with h == 0
  bool first;
  __device__ __forceinline__ void reset(int h) {
    m = 0.f; l = 0.f; first = true;
    rs.zero();
#pragma unroll
    for (int d = 0; d < Cfg<HD>::DBLK; ++d) acc[d] = zero16();
    const u32x4 z = {0u, 0u, 0u, 0u};
    qe = z;
    if (h == 0) qe.y = 0x3F800000u;      // -m = 0, slot 3 = 1.0 (picks up the key bias)
  }
  // accumulate the row sum of one packed fragment (8 bf16 probabilities)
  __device__ __forceinline__ void add_frag(const bf8_t& pf) { rs.add(pf); }
  __device__ __forceinline__ void scale_sum(float alpha) { l = (l + rs.take()) * alpha; }      // (alpha is the same in lane ^ 32: half sums stay half sums)
  __device__ __forceinline__ void finish_sum() { l += rs.take(); l = rs.whole(l); }
  __device__ __forceinline__ bool sums_ok() const { return l >= FWD2_TINY && l <= FWD2_BIG; }
};

// classes of the key tiles of a chunk from its fp32 bias image in LDS (keys >= nkeys count as masked): bit masks with 4 bits per tile,
// NZ = some key of the group of 8 has a bias, DEAD = all 8 keys of the group are masked.  class(t): 0 no bias (no bookkeeping k-step),
// 2 every key masked (tile skipped), 1 mixed.
// `ck` = floats in the bias image (the chunk size): lanes whose group of 8 lies beyond it (256-key chunks of the hd-64 kernel) never touch LDS.
__device__ __forceinline__ void tile_class_masks(const float* sBias, int ck, int nkeys, bool have_bias, int lane, unsigned long long& NZ, unsigned long long& DEAD) {
  bool nz = false, dead = true;
  const int k0 = lane * 8;
  if (have_bias && k0 < ck) {
    const float4 a = *reinterpret_cast<const float4*>(sBias + k0), b = *reinterpret_cast<const float4*>(sBias + k0 + 4);
    const float bv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool in = k0 + j < nkeys;
      nz = nz || !in || bv[j] != 0.f;
      dead = dead && (!in || bv[j] <= -1.0e30f);
    }
  } else {
    nz = k0 + 8 > nkeys;
    dead = k0 >= nkeys;
  }
  NZ = __ballot(nz); DEAD = __ballot(dead);
}
__device__ __forceinline__ int tile_class(unsigned long long NZ, unsigned long long DEAD, int t) {
  const unsigned nzb = (unsigned)(NZ >> (4 * t)) & 15u, ddb = (unsigned)(DEAD >> (4 * t)) & 15u;
  return nzb == 0u ? 0 : (ddb == 15u ? 2 : 1);
}
// K side of the bookkeeping k-step for key tile t: bf16 [1, 1, 1, bias[key], 0, 0, 0, 0] in the lanes with h == 0, zeros in the others
__device__ __forceinline__ bf8_t key_entry(const float* sBias, bool have_bias, int nkeys, int t, int lane) {
  const int key = t * 32 + (lane & 31);
  const float bv = key < nkeys ? (have_bias ? sBias[key] : 0.f) : -INFINITY;
  u32x4 ke = {0u, 0u, 0u, 0u};
  if ((lane >> 5) == 0) { ke.x = 0x3F803F80u; ke.y = 0x3F80u | (pack2bf(bv, 0.f) << 16); }
  return __builtin_bit_cast(bf8_t, ke);
}

// per-lane byte offsets of the K row fragments and the V^T transposed fragments inside a 32-key tile of the swizzled images (the swizzle of a row
// depends on the row modulo 32 at most, so the offsets are the same in every tile: tile t adds the wave-uniform t * 32 * ROWB).  Formed once per
// wave; left to the tile step, hipcc re-derives the swizzle arithmetic in every tile (14 vector instructions per tile, now 6 additions).
template <int HD> struct FragOff {
  int k[Cfg<HD>::KSTEPS];
  int v[2][Cfg<HD>::DBLK][2];
  __device__ __forceinline__ void init(int lane) {
    typedef Cfg<HD> C;
    const int h = lane >> 5, r = lane & 31, g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int st = 0; st < C::KSTEPS; ++st) k[st] = r * C::ROWB + (swz<C::HDP>(r, 2 * st + h) << 4);
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int db = 0; db < C::DBLK; ++db) {
        const int col = 32 * db + 16 * (g & 1) + 4 * (i & 3);
        const int r0 = 16 * sb + 4 * h + (i >> 2), r1 = r0 + 8;
        v[sb][db][0] = r0 * C::ROWB + (swz<C::HDP>(r0, col >> 3) << 4) + (col & 7) * 2;
        v[sb][db][1] = r1 * C::ROWB + (swz<C::HDP>(r1, col >> 3) << 4) + (col & 7) * 2;
      }
  }
  __device__ __forceinline__ int koff(int st) const { return k[st]; }
  __device__ __forceinline__ bf8_t vt(const unsigned char* tile, int sb, int db) const {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + v[sb][db][0]));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + v[sb][db][1]));
    s16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return __builtin_bit_cast(bf8_t, o);
  }
};
// the same offsets for 128-byte rows (hd 64) from TWO registers instead of twelve: with x(row) = (row >> 1) & 7 the row fragment of k-step st sits at
// k0 ^ (st << 5) (chunk 2 st + h = 2 st ^ h), and the transposed fragment (sb, db, j) at (v0 + 2048 sb + 1024 j) ^ ((db ^ j) << 6) (row 16 sb + 8 j + r0:
// x = 4 j ^ (r0 >> 1); chunk 4 db ^ (col0 >> 3)).  A few more vector instructions per tile, which this matrix-bound kernel has to spare.
struct FragOffW {
  int k0, v0;
  __device__ __forceinline__ void init(int lane) {
    const int h = lane >> 5, r = lane & 31, g = lane >> 4, i = lane & 15;
    k0 = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
    const int col0 = 16 * (g & 1) + 4 * (i & 3), r0 = 4 * h + (i >> 2);
    v0 = r0 * 128 + (((col0 >> 3) ^ (r0 >> 1)) << 4) + (col0 & 7) * 2;
  }
  __device__ __forceinline__ int koff(int st) const { return k0 ^ (st << 5); }
  __device__ __forceinline__ bf8_t vt(const unsigned char* tile, int sb, int db) const {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + ((v0 + 2048 * sb) ^ (db << 6))));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + ((v0 + 2048 * sb + 1024) ^ ((db ^ 1) << 6))));
    s16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return __builtin_bit_cast(bf8_t, o);
  }
};

// the K row fragments and V^T fragments of one 32-key tile
template <int HD> struct TileFrags {
  bf8_t k[Cfg<HD>::KSTEPS];
  bf8_t v[2][Cfg<HD>::DBLK];
  template <class FO>
  __device__ __forceinline__ void load(const unsigned char* sK, const unsigned char* sV, int t, const FO& fo) {
    typedef Cfg<HD> C;
    const unsigned char* kt = sK + t * 32 * C::ROWB;
    const unsigned char* vt = sV + t * 32 * C::ROWB;
#pragma unroll
    for (int st = 0; st < C::KSTEPS; ++st) k[st] = *reinterpret_cast<const bf8_t*>(kt + fo.koff(st));
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int d = 0; d < C::DBLK; ++d) v[sb][d] = fo.vt(vt, sb, d);
  }
};

// fast pass, one 32-key tile for NB blocks of 32 queries held by the wave: p = 2^s, no maximum.  BOOK: the tile has key biases (class 1).
// `cur` holds the tile's fragments; when `nxt` is given, the fragments of tile t_next are requested right behind the score chains, so that they
// land under this tile's exponentials instead of in front of the next tile's first MFMA (two waves per SIMD do not cover an LDS round trip).
// FWD3_PRIO: s_setprio around the phases of a tile.  3 (default, round 6): the score chains and each block's packs + row-sum / P.V MFMAs at priority 1, the
// exponentials at 0 -- the partner wave of the SIMD gets its MFMAs out while this one is in its exponentials: -2 % per launch (hd 32), bit-identical.
// 1: score chains only (-1.5 %); 2: the vector phase raised instead (+0.5 %); 0: none.
#ifndef FWD3_PRIO
#define FWD3_PRIO 3
#endif
template <int HD, int NB, bool BOOK, class Side>
__device__ __forceinline__ void fwd_tile_fast(const unsigned char* sK, const unsigned char* sV, const float* sBias, bool have_bias, int nkeys, int t,
                                              const bf8_t (&qf)[NB][Cfg<HD>::KSTEPS], RowState<HD> (&st)[NB], int lane, const FragOff<HD>& fo,
                                              const TileFrags<HD>& cur, TileFrags<HD>* nxt, int t_next, Side&& side) {
  typedef Cfg<HD> C;
  const int h = lane >> 5;
  const bf8_t (&kf)[C::KSTEPS] = cur.k;
  f32x16 s[NB];
#if FWD3_PRIO == 1 || FWD3_PRIO == 3
  __builtin_amdgcn_s_setprio(1);                             // the score chains ahead of the partner wave's instructions
#elif FWD3_PRIO == 2
  __builtin_amdgcn_s_setprio(0);
#endif
#ifdef FWD2_ABL_NOQK
#pragma unroll
  for (int b = 0; b < NB; ++b) { s[b] = zero16(); asm volatile("" : "+v"(s[b])); }
#else
  if constexpr (BOOK) {
    const bf8_t ke = key_entry(sBias, have_bias, nkeys, t, lane);
    u32x4 q1 = {0u, 0u, 0u, 0u};
    if (h == 0) q1.y = 0x3F800000u;                                    // [0, 0, 0, 1]: picks up the key bias
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = MFMA32(ke, __builtin_bit_cast(bf8_t, q1), zero16());
#pragma unroll
    for (int k = 0; k < C::KSTEPS; ++k)
#pragma unroll
      for (int b = 0; b < NB; ++b) s[b] = MFMA32(kf[k], qf[b][k], s[b]);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = MFMA32(kf[0], qf[b][0], zero16());
#pragma unroll
    for (int k = 1; k < C::KSTEPS; ++k)
#pragma unroll
      for (int b = 0; b < NB; ++b) s[b] = MFMA32(kf[k], qf[b][k], s[b]);
  }
#endif
  const bf8_t (&vf)[2][C::DBLK] = cur.v;
  if (nxt) nxt->load(sK, sV, t_next, fo);
  // all score chains are issued before the first exponential: left alone hipcc reuses one register tile for the blocks' scores and sinks the second
  // block's chain behind the first block's exponentials (one exposed MFMA latency per block and tile)
  __builtin_amdgcn_sched_barrier(0);
#if FWD3_PRIO == 1 || FWD3_PRIO == 3
  __builtin_amdgcn_s_setprio(0);
#elif FWD3_PRIO == 2
  __builtin_amdgcn_s_setprio(1);                             // (variant 2, measured slower) the exponentials / packs ahead of the partner wave's instructions
#endif
  side();      // the caller's per-tile share of memory instructions (k_attn_fwd3: LDS-DMA of the next slab, stores of the previous one)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#if FWD3_PRIO == 3
    __builtin_amdgcn_s_setprio(0);
#endif
#ifndef FWD2_ABL_NOEXP
#pragma unroll
    for (int e = 0; e < 16; ++e) s[b][e] = __builtin_amdgcn_exp2f(s[b][e]);
#endif
#if FWD3_PRIO == 3
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const bf8_t pf = pack8(s[b], sb);
      st[b].add_frag(pf);
#ifndef FWD2_ABL_NOPV
#pragma unroll
      for (int d = 0; d < C::DBLK; ++d) st[b].acc[d] = MFMA32(vf[sb][d], pf, st[b].acc[d]);
#else
      st[b].acc[0][sb] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, pf).x);
#endif
    }
  }
}

// exact pass, one 32-key tile for one block of 32 queries: deferred running maximum, subtracted inside the score chain (k_attn_fwd's tile step)
template <int HD>
__device__ __forceinline__ void fwd_tile_exact(const unsigned char* sK, const unsigned char* sV, const float* sBias, bool have_bias, int nkeys, int t,
                                               const bf8_t (&qf)[Cfg<HD>::KSTEPS], RowState<HD>& st, int lane) {
  typedef Cfg<HD> C;
  const int h = lane >> 5, r = lane & 31;
  const unsigned char* kt = sK + t * 32 * C::ROWB;
  f32x16 s = MFMA32(key_entry(sBias, have_bias, nkeys, t, lane), __builtin_bit_cast(bf8_t, st.qe), zero16());      // bias[key] - m[query]
#pragma unroll
  for (int k = 0; k < C::KSTEPS; ++k) s = MFMA32(rd_row<HD>(kt, r, k, h), qf[k], s);
  float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
  for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
  mx = fmaxf(mx, s[15]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  // deferred rescale (s is already relative to m): the running max moves when a tile exceeds it by more than the threshold, and
  // unconditionally on the very first tile (m starts at 0, not at the row maximum)
  if (st.first || __any(mx > RESCALE_THR * LOG2E)) {
    float dlt = st.first ? mx : fmaxf(mx, 0.f);
    if (!(dlt > -1e30f)) dlt = 0.f;                    // fully masked so far: keep m
    const float alpha = st.first ? 1.0f : __builtin_amdgcn_exp2f(-dlt);      // (nothing accumulated yet on the first tile; 2^-dlt may be inf there)
    st.scale_sum(alpha);
#pragma unroll
    for (int d = 0; d < C::DBLK; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) st.acc[d][e] *= alpha;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] -= dlt;
    st.m += dlt;
    if (h == 0) {
      unsigned w01, w2;
      split3_bf16(-st.m, w01, w2);
      st.qe.x = w01; st.qe.y = w2 | 0x3F800000u;
    }
    st.first = false;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(s[e]);
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const bf8_t pf = pack8(s, sb);
    st.add_frag(pf);
#pragma unroll
    for (int d = 0; d < C::DBLK; ++d) st.acc[d] = MFMA32(rd_tr<HD>(sV, t * 32, sb, d, lane), pf, st.acc[d]);
  }
}

// all key tiles of a chunk.  Runs of bias-free tiles go through a loop of their own that contains nothing else (with both tile forms in one
// loop body hipcc keeps the accumulators of the two forms in different registers and copies all of them every tile).
template <int HD, int NB, class Side>
__device__ __forceinline__ void fwd_chunk_fast(const unsigned char* sK, const unsigned char* sV, const float* sBias, bool have_bias, int nkeys, int nt,
                                               unsigned long long NZ, unsigned long long DEAD, const bf8_t (&qf)[NB][Cfg<HD>::KSTEPS], RowState<HD> (&st)[NB], int lane,
                                               Side&& side) {
  FragOff<HD> fo;
  fo.init(lane);
  unsigned long long special = (NZ | (NZ >> 1) | (NZ >> 2) | (NZ >> 3)) & 0x1111111111111111ull;      // bit 4 t: tile t has a bias somewhere
  if (nt < 16) special |= ~0ull << (4 * nt);
  int t = 0;
  while (t < nt) {
    const unsigned long long rest = special >> (4 * t);
    const int run_end = rest ? t + (__builtin_ctzll(rest) >> 2) : nt;                                  // first tile >= t that is not bias-free
    if (t < run_end) {                                       // a run of bias-free tiles: fragments one tile ahead, two register sets in turn
      TileFrags<HD> fa, fb;
      fa.load(sK, sV, t, fo);
      for (; t + 1 < run_end; t += 2) {
        fwd_tile_fast<HD, NB, false>(sK, sV, sBias, have_bias, nkeys, t, qf, st, lane, fo, fa, &fb, t + 1, side);
        fwd_tile_fast<HD, NB, false>(sK, sV, sBias, have_bias, nkeys, t + 1, qf, st, lane, fo, fb, &fa, t + 2, side);      // (t + 2 <= 16: inside the LDS allocation)
      }
      if (t < run_end) { fwd_tile_fast<HD, NB, false>(sK, sV, sBias, have_bias, nkeys, t, qf, st, lane, fo, fa, nullptr, 0, side); ++t; }
    }
    if (t < nt) {
      if (tile_class(NZ, DEAD, t) == 1) {
        TileFrags<HD> f;
        f.load(sK, sV, t, fo);
        fwd_tile_fast<HD, NB, true>(sK, sV, sBias, have_bias, nkeys, t, qf, st, lane, fo, f, nullptr, 0, side);
      }
      ++t;
    }
  }
}
template <int HD>
__device__ __forceinline__ void fwd_chunk_exact(const unsigned char* sK, const unsigned char* sV, const float* sBias, bool have_bias, int nkeys, int nt,
                                                unsigned long long NZ, unsigned long long DEAD, const bf8_t (&qf)[Cfg<HD>::KSTEPS], RowState<HD>& st, int lane) {
  for (int t = 0; t < nt; ++t)
    if (tile_class(NZ, DEAD, t) != 2) fwd_tile_exact<HD>(sK, sV, sBias, have_bias, nkeys, t, qf, st, lane);
}

// normalised outputs of a block's 32 queries: packed context pieces and the natural-log LSE.  A lane's accumulator registers hold 4 head-dim
// elements per group g (dd = 8 g + 4 h ...); v_permlane32_swap pairs the groups g, g + 1 across lane ^ 32 so that every lane owns 8 consecutive
// elements = one 16-byte store (half the store instructions: the per-CU memory pipe, not bandwidth, is what the stores cost here).
template <int HD> struct RowOut {
  u32x4 w[Cfg<HD>::DBLK * 2];        // piece (d, pr): head-dim elements 32 d + 16 pr + 8 h ... + 7
  float lse;
  __device__ __forceinline__ void from(const RowState<HD>& st) {
    const float lt = st.l;
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
#pragma unroll
    for (int d = 0; d < Cfg<HD>::DBLK; ++d)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int g = 2 * pr;
        unsigned ax = pack2bf(st.acc[d][4 * g] * inv, st.acc[d][4 * g + 1] * inv), ay = pack2bf(st.acc[d][4 * g + 2] * inv, st.acc[d][4 * g + 3] * inv);
        unsigned bx = pack2bf(st.acc[d][4 * g + 4] * inv, st.acc[d][4 * g + 5] * inv), by = pack2bf(st.acc[d][4 * g + 6] * inv, st.acc[d][4 * g + 7] * inv);
        // swap(a, b): a's upper half <-> b's lower half.  Lanes h = 0 then hold group g whole (own 4 | lane^32's 4), lanes h = 1 group g + 1
        const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        w[2 * d + pr].x = sx[0]; w[2 * d + pr].y = sy[0]; w[2 * d + pr].z = sx[1]; w[2 * d + pr].w = sy[1];
      }
    lse = (st.m + __log2f(lt)) * 0.6931471805599453f;
  }
  static constexpr int NPIECE = (HD + 15) / 16;      // 16-byte pieces per lane that exist (hd 16: one, hd 32: two, hd 64: four)
  __device__ __forceinline__ void store_piece(int i, bf16_t* dst /* ctx row of the query + head offset */, int h) const {
    *reinterpret_cast<u32x4*>(dst + 16 * i + 8 * h) = w[i];
  }
  __device__ __forceinline__ void store(bf16_t* dst, float* lse_dst, int h) const {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) store_piece(i, dst, h);
    if (lse_dst && h == 0) *lse_dst = lse;
  }
};

#ifdef FWD2_STAMP      // diagnostic build (tools/ab/fwd2_stamps.py): s_memtime at the phase boundaries
__device__ unsigned long long g_fwd2_stamps[64 * 16 * 8];
#define FWD2_T(i)                                                                                              \
  do {                                                                                                         \
    unsigned long long t_;                                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 64 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 5)) \
      g_fwd2_stamps[((blockIdx.x / 37) * 2 + ((threadIdx.x >> 6) == 5)) * 8 + (i)] = t_;                       \
  } while (0)
#define FWD3_T(i)                                                                                              \
  do {                                                                                                         \
    unsigned long long t_;                                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.x % 4 == 0 && slab_no == 3)                                        \
      g_fwd2_stamps[((blockIdx.x / 4) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = t_;                              \
  } while (0)
extern "C" int oneprot_attn_debug_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fwd2_stamps), sizeof(unsigned long long) * 64 * 16 * 8) == hipSuccess ? 0 : -1;
}
#else
#define FWD2_T(i)
#define FWD3_T(i)
#endif

// ---- k_attn_fwd2: one work-group per 256 queries, keys in chunks through registers -----------------------------------------------------
template <int HD> struct Fwd2 {
  typedef Cfg<HD> C;
  static constexpr int CK = HD <= 32 ? 512 : 256;                  // keys per LDS chunk
  static constexpr int TILE = CK * C::ROWB;
  static constexpr int BIAS_OFF = 2 * TILE;                        // fp32 key bias of the chunk
  static constexpr int FLAG_OFF = BIAS_OFF + CK * 4;
  static constexpr int TOTAL = FLAG_OFF + 16;
};

// one pass over all keys for the wave's 32 queries; EXACT = with the running maximum.  Every wave of the work-group calls it (barriers inside).
template <int HD, bool EXACT>
__device__ __forceinline__ void fwd2_pass(unsigned char* smem, const bf16_t* __restrict__ kbase, const bf16_t* __restrict__ vbase, const float* __restrict__ bias_row,
                                          int L, bool active, const bf8_t (&qf)[1][Cfg<HD>::KSTEPS], RowState<HD> (&st)[1]) {
  typedef Cfg<HD> C;
  typedef Fwd2<HD> F;
  unsigned char* sK = smem;
  unsigned char* sV = smem + F::TILE;
  float* sBias = reinterpret_cast<float*>(smem + F::BIAS_OFF);
  const int nthr = blockDim.x;
  const int lane = threadIdx.x & 63;
  st[0].reset(lane >> 5);
  for (int kc0 = 0; kc0 < L; kc0 += F::CK) {
    const int nkeys = min(F::CK, L - kc0);
    const int nrows = (nkeys + 31) & ~31;
    __syncthreads();
    if (!EXACT) FWD2_T(1);
    // ---- stage the chunk: all loads of a round in flight together, then committed to the swizzled images
    {
#ifdef FWD2_ABL_NOLOAD
      const int total = 0;
#else
      const int total = nrows * C::NCH;
#endif
      for (int base = 0; base < total; base += 4 * nthr) {
        u32x4 rk[4], rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = base + j * nthr + (int)threadIdx.x;
          const int row = idx / C::NCH, ch = idx - row * C::NCH;
          const u32x4 z = {0u, 0u, 0u, 0u};
          rk[j] = z; rv[j] = z;
          if (idx < total && row < nkeys && ch < HD / 8) {
            rk[j] = *reinterpret_cast<const u32x4*>(kbase + (size_t)(kc0 + row) * HD + ch * 8);
            rv[j] = *reinterpret_cast<const u32x4*>(vbase + (size_t)(kc0 + row) * HD + ch * 8);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = base + j * nthr + (int)threadIdx.x;
          const int row = idx / C::NCH, ch = idx - row * C::NCH;
          if (idx < total) {
            const int off = row * C::ROWB + (swz<C::HDP>(row, ch) << 4);
            *reinterpret_cast<u32x4*>(sK + off) = rk[j];
            *reinterpret_cast<u32x4*>(sV + off) = rv[j];
          }
        }
      }
      if (!EXACT) FWD2_T(2);
      if (bias_row)
        for (int i = threadIdx.x; i < F::CK; i += nthr) sBias[i] = i < nkeys ? bias_row[kc0 + i] : 0.f;
    }
    if (!EXACT) FWD2_T(3);
    __syncthreads();
    if (!EXACT) FWD2_T(4);
    if (!active) continue;
    unsigned long long NZ, DEAD;
    tile_class_masks(sBias, F::CK, nkeys, bias_row != nullptr, lane, NZ, DEAD);
    if constexpr (EXACT) fwd_chunk_exact<HD>(sK, sV, sBias, bias_row != nullptr, nkeys, nrows >> 5, NZ, DEAD, qf[0], st[0], lane);
    else fwd_chunk_fast<HD, 1>(sK, sV, sBias, bias_row != nullptr, nkeys, nrows >> 5, NZ, DEAD, qf, st, lane, [] {});
  }
}

template <int HD>
__device__ __forceinline__ void fwd2_body(unsigned char* smem, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                          const float* __restrict__ key_bias, bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int H, int L, int bh, int qb) {
  typedef Cfg<HD> C;
  typedef Fwd2<HD> F;
  int* sFlag = reinterpret_cast<int*>(smem + F::FLAG_OFF);
  const int b = bh / H, head = bh - b * H;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, r = lane & 31;
  const int q0 = qb * 256 + wave * 32;
  const bool active = q0 < L;
  const int qidx = q0 + r;
  const int qrow = qidx < L ? qidx : L - 1;
  const bf16_t* kbase = k + (size_t)bh * L * HD;
  const bf16_t* vbase = v + (size_t)bh * L * HD;
  const float* bias_row = key_bias ? key_bias + (size_t)b * L : nullptr;
  bf8_t qf[1][C::KSTEPS];
#pragma unroll
  for (int st = 0; st < C::KSTEPS; ++st) qf[0][st] = *reinterpret_cast<const bf8_t*>(q + ((size_t)bh * L + qrow) * HD + 16 * st + 8 * h);
  if (threadIdx.x == 0) *sFlag = 0;
  FWD2_T(0);
  RowState<HD> st[1];
  fwd2_pass<HD, false>(smem, kbase, vbase, bias_row, L, active, qf, st);
  st[0].finish_sum();
  FWD2_T(5);
  {
#ifdef FWD2_NOFALLBACK      // timing builds of ablated kernels (wrong sums by construction)
    const bool bad = false;
#else
    const bool bad = active && __any(!st[0].sums_ok());
#endif
    if (bad && lane == 0) *sFlag = 1;
    __syncthreads();
    if (*sFlag != 0) {      // the whole work-group repeats with the running maximum
      fwd2_pass<HD, true>(smem, kbase, vbase, bias_row, L, active, qf, st);
      st[0].finish_sum();
    }
  }
  FWD2_T(6);
  if (!active) return;
  RowOut<HD> out;
  out.from(st[0]);
  if (qidx < L) out.store(ctx + ((size_t)b * L + qidx) * (H * HD) + head * HD, lse_out ? lse_out + (size_t)bh * L + qidx : nullptr, h);
#ifdef FWD2_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  FWD2_T(7);
}

template <int HD>
__global__ void __launch_bounds__(512, HD <= 32 ? 4 : 2) k_attn_fwd2(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  const float* __restrict__ key_bias, bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int B,
                                                                  int H, int L, int nqb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int bh, qb;
  decode_block(nqb, B * H, bh, qb);
  if (bh >= B * H) return;
  fwd2_body<HD>(smem, q, k, v, key_bias, ctx, lse_out, H, L, bh, qb);
}

// the slabs that k_attn_fwd3w marked (sums out of range), once more through k_attn_fwd2's path (fast pass, then the work-group-wide exact pass).  Every
// work-group first reads the launch's one word: nothing marked (the normal case) and it leaves.
template <int HD>
__global__ void __launch_bounds__(512, 2) k_attn_fwd2_redo(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                        const float* __restrict__ key_bias, bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int B, int H, int L,
                                                        int nqb, const unsigned* __restrict__ redo_any, const unsigned* __restrict__ redo_slab, unsigned epoch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (*redo_any != epoch) return;
  const int items = B * H * nqb;
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int bh = item / nqb, qb = item - bh * nqb;
    if (redo_slab[bh] != epoch) continue;
    fwd2_body<HD>(smem, q, k, v, key_bias, ctx, lse_out, H, L, bh, qb);
    __syncthreads();
  }
}

// ---- k_attn_fwd3: persistent, LDS-DMA double buffered, two query blocks per wave (hd <= 32, L <= 512) ------------------------------------
static __device__ __attribute__((aligned(16))) unsigned int g_attn_zero_page[4] = {0, 0, 0, 0};
// One LDS-DMA piece: 64 lanes x 16 (or 4) bytes from per-lane global addresses to LDS bytes [lds_dst, +1024) (or +256), lane-linear.  M0 (the
// LDS destination) is written in the statement that uses it and restored afterwards; these loads are invisible to hipcc's s_waitcnt bookkeeping:
// the kernel waits for them itself (vmcnt(0) before the slab barrier).
__device__ __forceinline__ void glds16_addr(const void* src, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4_addr(const void* src, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}

// the same with a wave-uniform 64-bit base in SGPRs + one 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

struct Fwd3 {
  static constexpr int ROWS = 512;
  static constexpr int TILE = ROWS * 64;                // 64-byte swizzled rows (hd = 16 zero-padded)
  static constexpr int BUF = 2 * TILE + ROWS * 4;       // K image, V image, fp32 bias
  static constexpr int TOTAL = 2 * BUF;                 // 132 KB
};

template <int HD>
__global__ void __launch_bounds__(512, 2) k_attn_fwd3(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                   const float* __restrict__ key_bias, bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int B, int H,
                                                   int L, unsigned* __restrict__ sched) {
  typedef Cfg<HD> C;
  static_assert(C::ROWB == 64, "k_attn_fwd3 stages 64-byte rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, r = lane & 31;
  const int nw = blockDim.x >> 6;
  // consecutive work-groups go to consecutive XCDs: XCD x takes the x-th eighth of the (b, h) slabs, its work-groups walk that range with a
  // stride of (work-groups per XCD), so the heads of one batch element -- which share the 128-byte lines of the ctx rows -- meet in one L2
  const int nbh = B * H, per_xcd = (nbh + 7) >> 3, nslot = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int bh_end = min(nbh, (xcd + 1) * per_xcd);
  // Which slabs.  Static (sched == null): the work-group's own, stride (work-groups per XCD).  Dynamic: tickets drawn from XCD x's queue head of the sched
  // workspace (sched_ws.h) -- two at the start, then one per slab, drawn (one lane, a returning agent-scope atomic) right behind the slab's barrier and
  // looked at behind the NEXT slab's vmcnt(0), where it costs nothing; it reaches the other waves through a word of the LDS' spare room across that
  // barrier.  A work-group that could not start with the others finds the queue empty and leaves; the running ones share the slabs.
  unsigned* const relay = reinterpret_cast<unsigned*>(smem + Fwd3::TOTAL);      // 2 words (the launch asks for TOTAL + 16 bytes)
  int bh, bh_next;
  if (sched) {
    if (threadIdx.x == 0) {
      unsigned* head = sched + SW_HEAD(xcd);
      const unsigned t0 = sw_draw(head), t1 = sw_draw(head);
      relay[0] = t0 < 0x3fffffffu ? t0 : 0x3fffffffu; relay[1] = t1 < 0x3fffffffu ? t1 : 0x3fffffffu;
    }
    __syncthreads();
    bh = xcd * per_xcd + __builtin_amdgcn_readfirstlane((int)relay[0]);
    bh_next = xcd * per_xcd + __builtin_amdgcn_readfirstlane((int)relay[1]);
    __syncthreads();
  } else {
    bh = xcd * per_xcd + (int)(blockIdx.x >> 3);
    bh_next = bh + nslot;
  }
  // the wave's two query blocks: rows [64 w, 64 w + 32) and [64 w + 32, 64 w + 64); the second one may lie beyond L (its rows then repeat row L - 1
  // and are not stored)
  const int q0 = wave * 64;
  const bool active = q0 < L;
  int qidx[2], qrow[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) { qidx[b] = q0 + 32 * b + r; qrow[b] = qidx[b] < L ? qidx[b] : L - 1; }
  const int nrows = (L + 31) & ~31;
  const int npieces = nrows >> 4;                           // 1 KiB pieces (16 rows) per image
  const int dm = H * HD;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_attn_zero_page);

  // One memory instruction of the slab's list: LDS-DMA pieces of slab `item` (K and V alternating, then the bias piece), its q fragments, the
  // output stores of slab `pitem`.  The list is worked off one or two entries per key tile INSIDE the tile loop: issued in one burst after the
  // barrier, the ~180 memory instructions of the 8 waves queued on the CU's one memory pipe for 7-12 k cycles during which no wave computed
  // (profiles/r04_fwd3_stamps.txt).
  const int my_pieces = npieces > wave ? (npieces - wave + nw - 1) / nw : 0;
  const int n_biasp = (nrows + 63) >> 6;
  // A piece that lies wholly inside the slab's L rows is one instruction on a wave-uniform base (slab + piece) and ONE per-lane offset that is the
  // same for every piece (the swizzle of row 16 p + j depends on j only; the zero-padded chunks of hd = 16 re-read chunk 0: nothing reads their
  // LDS columns for a value that is kept); only the piece that straddles L forms per-lane 64-bit addresses with the zero page behind rows >= L.
  const unsigned kv_lane_off = (unsigned)(lane >> 2) * (HD * 2) + (unsigned)((((lane & 3) ^ ((lane >> 4) & 3)) < HD / 8) ? ((lane & 3) ^ ((lane >> 4) & 3)) : 0) * 16u;
  auto dma_piece = [&](int item, int buf, int i) {           // i < 2 * my_pieces: K (even) / V (odd) piece i / 2 of this wave; i == 2 * my_pieces: bias
    const unsigned dst0 = lds0 + buf * Fwd3::BUF;
    if (i < 2 * my_pieces) {
      const int p = wave + nw * (i >> 1);
      const unsigned char* base = reinterpret_cast<const unsigned char*>(((i & 1) ? v : k) + (size_t)item * L * HD);
      const unsigned dst = dst0 + ((i & 1) ? Fwd3::TILE : 0) + p * 1024;
      if (16 * p + 16 <= L) glds16_s(kv_lane_off, base + (size_t)p * (16 * HD * 2), dst);
      else {
        const int row = 16 * p + (lane >> 2);
        const int ch = (lane & 3) ^ ((row >> 2) & 3);       // the chunk that lives at this lane's place of the swizzled image
        const bool ok = row < L && ch < HD / 8;
        glds16_addr(ok ? base + (size_t)row * (HD * 2) + ch * 16 : zero, dst);
      }
    } else {
      const float* br = key_bias + (size_t)(item / H) * L + 64 * wave;
      if (64 * wave + 64 <= L) glds4_s((unsigned)lane * 4u, br, dst0 + 2 * Fwd3::TILE + wave * 256);
      else glds4_addr(64 * wave + lane < L ? (const void*)(br + lane) : (const void*)zero, dst0 + 2 * Fwd3::TILE + wave * 256);
    }
  };
  const int n_dma = 2 * my_pieces + ((key_bias && wave < n_biasp) ? 1 : 0);
  constexpr int NQ = 2 * C::KSTEPS;                          // q fragment loads per slab
  constexpr int NST = 2 * (RowOut<HD>::NPIECE + 1);          // output stores per slab (context pieces + lse, two blocks)
  auto load_q1 = [&](int item, bf8_t (&dstq)[2][C::KSTEPS], int i) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st)
        if (i == b * C::KSTEPS + st) dstq[b][st] = *reinterpret_cast<const bf8_t*>(q + ((size_t)item * L + qrow[b]) * HD + 16 * st + 8 * h);
  };
  auto store1 = [&](const RowOut<HD> (&o)[2], int item, int i) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      bf16_t* dst = ctx + ((size_t)(item / H) * L + qidx[b]) * dm + (item % H) * HD;
#pragma unroll
      for (int pc = 0; pc < RowOut<HD>::NPIECE; ++pc)
        if (i == b * (RowOut<HD>::NPIECE + 1) + pc && qidx[b] < L) o[b].store_piece(pc, dst, h);
      if (i == b * (RowOut<HD>::NPIECE + 1) + RowOut<HD>::NPIECE && lse_out && h == 0 && qidx[b] < L) lse_out[(size_t)item * L + qidx[b]] = o[b].lse;
    }
  };

  bf8_t qf[2][C::KSTEPS], qn[2][C::KSTEPS];
  if (bh < bh_end) {
    for (int i = 0; i < n_dma; ++i) dma_piece(bh, 0, i);
    for (int i = 0; i < NQ; ++i) load_q1(bh, qf, i);
  }
  RowOut<HD> pend[2];
  int pend_bh = -1, buf = 0;
  int slab_no = 0;
  // (wave 0, lane 0) the ticket drawn during the previous slab -- for the slab AFTER `bh_next`.  The draw is an inline-asm atomic (sched_ws.h: through the
  // builtin hipcc waits for it, vmcnt(0), right where it is issued: wave 0 then starts every slab a memory round trip late, +100 us per launch); its register
  // is read behind the next slab's own vmcnt(0), where it has returned whatever the order of returns is (check_async_regs.py: nothing else touches it).
  unsigned drawn_v = 0;
  const unsigned relay_lds = lds0 + Fwd3::TOTAL;
  const int draw_lane = (sched != nullptr && wave == 0) ? 1 : 0;
  while (bh < bh_end) {
    FWD3_T(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of slab `bh` have landed (and its q fragments, and older stores -- and the draw)
    FWD3_T(1);
    if (sched && slab_no > 0) {
      int late;
      const int t = draw_result(drawn_v, late);
      if (wave == 0) lds_write32(relay_lds + 4u * (slab_no & 1), (unsigned)(t < 0 || t > 0x3fffffff ? 0x3fffffff : t));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                           // ... everybody's have; everybody has left the other half of the LDS
    FWD3_T(2);
    if (sched && slab_no > 0) {
      const unsigned rv = lds_read32(relay_lds + 4u * (slab_no & 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bh_next = xcd * per_xcd + first_lane(rv);
    }
    drawn_v = draw_async(sched ? sched + SW_HEAD(xcd) : nullptr, draw_lane & (int)((unsigned)(bh_next - bh_end) >> 31));      // (EXEC mask empty unless this wave draws and a next slab exists)
#ifdef FWD3_DEPHASE
    if (wave >= 4) __builtin_amdgcn_s_sleep(FWD3_DEPHASE);    // (ablation) the second wave of every SIMD starts the slab FWD3_DEPHASE x 64 cycles behind the first
#endif
    // hipcc does not see the wait above: it would put its own vmcnt(0) in front of the first use of the q fragments -- behind the LDS-DMA of the
    // NEXT slab, i.e. the compute would start only after that transfer.  Consuming the fragments here puts its wait where it is free.
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) asm volatile("" : "+v"(qf[b][st]));
    const int nxt = bh_next;
    const bool more = nxt < bh_end;
    // the q fragments of the next slab in one go (a load whose result registers are written under a condition inside the tile loop makes hipcc wait
    // for vmcnt(0) -- i.e. for the LDS-DMA too -- in every iteration)
#ifndef FWD3_ABL_NOQ
    if (more)
      for (int i = 0; i < NQ; ++i) load_q1(nxt, qn, i);
#else
    for (int b = 0; b < 2; ++b) for (int st = 0; st < C::KSTEPS; ++st) qn[b][st] = qf[b][st];
#endif
    // The slab's memory instructions, spread over its key tiles (in one burst behind the barrier they queue on the CU's one memory pipe while nobody
    // computes: NOTEBOOK 6d).  Round 6: a SHORT list with a cheap step -- slot j < n_pairs: K piece j and V piece j of this wave for the next slab (two
    // instructions, addresses = a per-slab scalar base + p * constant); then the bias piece; then all stores of block 0; then all stores of block 1.
    // The first form was a generic list of single instructions behind a ladder of compares (which of ~17 entries is entry i?): ~30 scalar
    // instructions per step, two steps per tile, in every wave.
    const int n_pairs = more ? my_pieces : 0;
    const bool bias_piece = more && key_bias && wave < n_biasp;
    const bool have_stores = pend_bh >= 0 && active;
    const int n_slots = n_pairs + 3;
    const unsigned char* const kbase = reinterpret_cast<const unsigned char*>(k + (size_t)nxt * L * HD);
    const unsigned char* const vbase = reinterpret_cast<const unsigned char*>(v + (size_t)nxt * L * HD);
    const unsigned dst_next = lds0 + (buf ^ 1) * Fwd3::BUF;
    int vm_i = 0;
    auto vm_step = [&]() {
      const int i = vm_i;
      if (i >= n_slots) return;
      vm_i = i + 1;
      if (i < n_pairs) {
#ifndef FWD3_ABL_NODMA
        const int p = wave + nw * i;
        const unsigned dk = dst_next + p * 1024;
        if (16 * p + 16 <= L) {
          glds16_s(kv_lane_off, kbase + (size_t)p * (16 * HD * 2), dk);
          glds16_s(kv_lane_off, vbase + (size_t)p * (16 * HD * 2), dk + Fwd3::TILE);
        } else {                                            // the piece that straddles L: per-lane addresses, the zero page behind rows >= L
          const int row = 16 * p + (lane >> 2);
          const int ch = (lane & 3) ^ ((row >> 2) & 3);
          const bool ok = row < L && ch < HD / 8;
          glds16_addr(ok ? kbase + (size_t)row * (HD * 2) + ch * 16 : zero, dk);
          glds16_addr(ok ? vbase + (size_t)row * (HD * 2) + ch * 16 : zero, dk + Fwd3::TILE);
        }
#endif
      } else if (i == n_pairs) {
#ifndef FWD3_ABL_NODMA
        if (bias_piece) {
          const float* br = key_bias + (size_t)(nxt / H) * L + 64 * wave;
          if (64 * wave + 64 <= L) glds4_s((unsigned)lane * 4u, br, dst_next + 2 * Fwd3::TILE + wave * 256);
          else glds4_addr(64 * wave + lane < L ? (const void*)(br + lane) : (const void*)zero, dst_next + 2 * Fwd3::TILE + wave * 256);
        }
#endif
      } else if (have_stores) {
#ifndef FWD3_ABL_NOSTORE
        if (i == n_pairs + 1) {
#pragma unroll
          for (int j = 0; j < NST / 2; ++j) store1(pend, pend_bh, j);
        } else {
#pragma unroll
          for (int j = NST / 2; j < NST; ++j) store1(pend, pend_bh, j);
        }
#endif
      }
    };
    const int e_all = n_slots;
    FWD3_T(3);
    if (active) {
      const unsigned char* sK = smem + buf * Fwd3::BUF;
      const unsigned char* sV = sK + Fwd3::TILE;
      const float* sBias = reinterpret_cast<const float*>(sK + 2 * Fwd3::TILE);
      const bool have_bias = key_bias != nullptr;
      unsigned long long NZ, DEAD;
      tile_class_masks(sBias, Fwd3::ROWS, L, have_bias, lane, NZ, DEAD);
      const int nt = nrows >> 5;
      RowState<HD> st[2];
      st[0].reset(h); st[1].reset(h);
      fwd_chunk_fast<HD, 2>(sK, sV, sBias, have_bias, L, nt, NZ, DEAD, qf, st, lane, [&] { vm_step(); });      // one list entry per key tile (7 entries at L = 512, 16 tiles)
      while (vm_i < e_all) vm_step();                       // (short sequences / skipped tiles: whatever is left of the list)
      st[0].finish_sum(); st[1].finish_sum();
      FWD3_T(4);
#ifndef FWD2_NOFALLBACK
      // the whole slab is resident: a block whose sums are out of range repeats its own rows with the running maximum, no barrier involved
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (__any(!st[b].sums_ok())) {
          st[b].reset(h);
          fwd_chunk_exact<HD>(sK, sV, sBias, have_bias, L, nt, NZ, DEAD, qf[b], st[b], lane);
          st[b].finish_sum();
        }
#endif
      pend[0].from(st[0]); pend[1].from(st[1]);
    } else {
      while (vm_i < e_all) vm_step();                       // a wave without queries still moves its share of the next slab
    }
    FWD3_T(5);
    ++slab_no;
    pend_bh = bh;
    bh = nxt; buf ^= 1;
    if (!sched) bh_next = bh + nslot;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) qf[b][st] = qn[b][st];
  }
  if (pend_bh >= 0 && active)
    for (int i = 0; i < NST; ++i) store1(pend, pend_bh, i);
  if (sched && threadIdx.x == 0) sw_leave(sched, gridDim.x, (unsigned)slab_no, (unsigned)nbh);
}

// ---- k_attn_fwd3w: the persistent form for 128-byte rows (hd = 64: BERT-base, ESM-2-650M), L <= 512 ------------------------------------------
// Same skeleton as k_attn_fwd3 -- one persistent 8-wave work-group per CU, two 32-query blocks per wave, K / V / bias by LDS-DMA into the other half
// of the LDS while the current half is computed, one barrier per buffer -- but a buffer holds a CHUNK of 256 keys (2 x 32 KB images), so a slab of
// L > 256 is two chunks and the stream of buffers runs across slab boundaries: [slab n chunk 0][slab n chunk 1][slab n+1 chunk 0] ...  What differs
// from the 64-byte-row kernel, all for registers (two blocks of hd 64 hold 64 output accumulators and 32 q-fragment registers per wave):
//   * ONE set of K / V^T tile fragments, reloaded in place (K behind the score chains, V^T behind the P.V MFMAs) instead of two sets in turn;
//   * the q fragments of the next slab are requested at the start of the slab's LAST chunk, not a whole slab ahead;
//   * the outputs leave at the end of the slab's last chunk (not from the next slab's tile loop: no registers to hold them);
//   * a slab's first chunk is gone from the LDS when the row sums are checked, and an exact pass inside this kernel cost the fast path 47-88 spilled
//     registers wherever it was put: a slab with a sum out of range is MARKED (redo_slab[bh] = epoch of this launch, one word redo_any for the launch)
//     and k_attn_fwd2_redo, launched right behind, repeats the marked slabs with k_attn_fwd2's work-group-wide exact pass.  In the normal case that
//     kernel is 256 work-groups that read one word and leave.  A correctness net for scores beyond +-60, not a working point (tests drive it with +-400).
#define REDO_SLOTS 8
#define REDO_SLABS 32768
__device__ unsigned g_fwd3w_redo_any[REDO_SLOTS];
__device__ unsigned g_fwd3w_redo_slab[REDO_SLOTS * REDO_SLABS];
template <int HD> struct Fwd3W {
  static constexpr int CK = 256;                           // keys per buffer
  static constexpr int TILE = CK * Cfg<HD>::ROWB;          // 32 KB at hd 64
  static constexpr int BUF = 2 * TILE + CK * 4;            // K image, V image, fp32 bias
  static constexpr int TOTAL = 2 * BUF + 4096;             // (+ one tile: the fragment prefetch of the tile behind a chunk's last one stays inside the allocation)
};

// fast-pass tile with ONE fragment set: NEXT reloads it in place for tile t + 1
template <int HD, int NB, bool BOOK, bool NEXT, class Side>
__device__ __forceinline__ void fwd_tile_fast_ip(const unsigned char* sK, const unsigned char* sV, const float* sBias, bool have_bias, int nkeys, int t,
                                                 const bf8_t (&qf)[NB][Cfg<HD>::KSTEPS], RowState<HD> (&st)[NB], int lane, const FragOffW& fo,
                                                 TileFrags<HD>& f, Side&& side) {
  typedef Cfg<HD> C;
  const int h = lane >> 5;
  f32x16 s[NB];
  if constexpr (BOOK) {
    const bf8_t ke = key_entry(sBias, have_bias, nkeys, t, lane);
    u32x4 q1 = {0u, 0u, 0u, 0u};
    if (h == 0) q1.y = 0x3F800000u;
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = MFMA32(ke, __builtin_bit_cast(bf8_t, q1), zero16());
#pragma unroll
    for (int k = 0; k < C::KSTEPS; ++k)
#pragma unroll
      for (int b = 0; b < NB; ++b) s[b] = MFMA32(f.k[k], qf[b][k], s[b]);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = MFMA32(f.k[0], qf[b][0], zero16());
#pragma unroll
    for (int k = 1; k < C::KSTEPS; ++k)
#pragma unroll
      for (int b = 0; b < NB; ++b) s[b] = MFMA32(f.k[k], qf[b][k], s[b]);
  }
  if constexpr (NEXT) {
    const unsigned char* kt = sK + (t + 1) * 32 * C::ROWB;
#pragma unroll
    for (int st_ = 0; st_ < C::KSTEPS; ++st_) f.k[st_] = *reinterpret_cast<const bf8_t*>(kt + fo.koff(st_));
  }
  __builtin_amdgcn_sched_barrier(0);
  side();
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s[b][e] = __builtin_amdgcn_exp2f(s[b][e]);
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const bf8_t pf = pack8(s[b], sb);
      st[b].add_frag(pf);
#pragma unroll
      for (int d = 0; d < C::DBLK; ++d) st[b].acc[d] = MFMA32(f.v[sb][d], pf, st[b].acc[d]);
    }
  }
  if constexpr (NEXT) {
    const unsigned char* vt = sV + (t + 1) * 32 * C::ROWB;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int d = 0; d < C::DBLK; ++d) f.v[sb][d] = fo.vt(vt, sb, d);
  }
}

template <int HD, int NB, class Side>
__device__ __forceinline__ void fwd_chunk_fast_ip(const unsigned char* sK, const unsigned char* sV, const float* sBias, bool have_bias, int nkeys, int nt,
                                                  unsigned long long NZ, unsigned long long DEAD, const bf8_t (&qf)[NB][Cfg<HD>::KSTEPS], RowState<HD> (&st)[NB],
                                                  int lane, const FragOffW& fo, Side&& side) {
  unsigned long long special = (NZ | (NZ >> 1) | (NZ >> 2) | (NZ >> 3)) & 0x1111111111111111ull;
  if (nt < 16) special |= ~0ull << (4 * nt);
  int t = 0;
  while (t < nt) {
    const unsigned long long rest = special >> (4 * t);
    const int run_end = rest ? t + (__builtin_ctzll(rest) >> 2) : nt;
    if (t < run_end) {
      TileFrags<HD> f;
      f.load(sK, sV, t, fo);
      for (; t + 1 < run_end; ++t) fwd_tile_fast_ip<HD, NB, false, true>(sK, sV, sBias, have_bias, nkeys, t, qf, st, lane, fo, f, side);
      fwd_tile_fast_ip<HD, NB, false, false>(sK, sV, sBias, have_bias, nkeys, t, qf, st, lane, fo, f, side);
      ++t;
    }
    if (t < nt) {
      if (tile_class(NZ, DEAD, t) == 1) {
        TileFrags<HD> f;
        f.load(sK, sV, t, fo);
        fwd_tile_fast_ip<HD, NB, true, false>(sK, sV, sBias, have_bias, nkeys, t, qf, st, lane, fo, f, side);
      }
      ++t;
    }
  }
}

template <int HD>
__global__ void __launch_bounds__(512, 2) k_attn_fwd3w(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                    const float* __restrict__ key_bias, bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int B, int H,
                                                    int L, unsigned* __restrict__ redo_any, unsigned* __restrict__ redo_slab, unsigned epoch) {
  typedef Cfg<HD> C;
  typedef Fwd3W<HD> F;
  static_assert(C::ROWB == 128, "k_attn_fwd3w stages 128-byte rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, r = lane & 31;
  const int nw = blockDim.x >> 6;
  const int nbh = B * H, per_xcd = (nbh + 7) >> 3, nslot = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int bh_end = min(nbh, (xcd + 1) * per_xcd);
  int bh = xcd * per_xcd + (int)(blockIdx.x >> 3);
  const int q0 = wave * 64;
  const bool active = q0 < L;
  int qidx[2], qrow[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) { qidx[b] = q0 + 32 * b + r; qrow[b] = qidx[b] < L ? qidx[b] : L - 1; }
  const int nrows = (L + 31) & ~31;
  const int nch = (nrows + F::CK - 1) / F::CK;              // chunks per slab (1 or 2)
  const int dm = H * HD;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_attn_zero_page);
  const bool have_bias = key_bias != nullptr;

  // One LDS-DMA piece = 8 rows x 128 B, lane-linear: lane (j = lane >> 3, slot = lane & 7) fills row 8 p + j, slot `slot` of the swizzled image with the
  // source chunk slot ^ ((row >> 1) & 7) = slot ^ (j >> 1) ^ 4 (p & 1): one per-lane offset, bit 6 flipped for odd pieces.
  const unsigned kv_lane_off = (unsigned)(lane >> 3) * 128u + (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) << 4);
  auto rows_of = [&](int c) { return min(F::CK, nrows - c * F::CK); };
  auto n_dma_of = [&](int c) {
    const int np = rows_of(c) >> 3;
    const int mine = np > wave ? (np - wave + nw - 1) / nw : 0;
    return 2 * mine + ((have_bias && 64 * wave < rows_of(c)) ? 1 : 0);
  };
  auto dma_piece = [&](int item, int c, int buf, int i) {      // i-th memory instruction of this wave for chunk c of slab `item`: K / V pieces alternating, then the bias
    const int np = rows_of(c) >> 3;
    const int mine = np > wave ? (np - wave + nw - 1) / nw : 0;
    const unsigned dst0 = lds0 + buf * F::BUF;
    if (i < 2 * mine) {
      const int p = wave + nw * (i >> 1);
      const int row0 = c * F::CK + 8 * p;
      const unsigned char* base = reinterpret_cast<const unsigned char*>(((i & 1) ? v : k) + (size_t)item * L * HD);
      const unsigned dst = dst0 + ((i & 1) ? F::TILE : 0) + p * 1024;
      if (row0 + 8 <= L) glds16_s(kv_lane_off ^ ((unsigned)(p & 1) << 6), base + (size_t)row0 * (HD * 2), dst);
      else {
        const int row = row0 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        glds16_addr(row < L ? base + (size_t)row * (HD * 2) + ch * 16 : zero, dst);
      }
    } else {
      const int k0 = c * F::CK + 64 * wave;
      const float* br = key_bias + (size_t)(item / H) * L + k0;
      if (k0 + 64 <= L) glds4_s((unsigned)lane * 4u, br, dst0 + 2 * F::TILE + wave * 256);
      else glds4_addr(k0 + lane < L ? (const void*)(br + lane) : (const void*)zero, dst0 + 2 * F::TILE + wave * 256);
    }
  };
  auto load_q = [&](int item, bf8_t (&dstq)[2][C::KSTEPS]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) dstq[b][st] = *reinterpret_cast<const bf8_t*>(q + ((size_t)item * L + qrow[b]) * HD + 16 * st + 8 * h);
  };

  bf8_t qf[2][C::KSTEPS], qn[2][C::KSTEPS];
  if (bh < bh_end) {
    const int n0 = n_dma_of(0);
    for (int i = 0; i < n0; ++i) dma_piece(bh, 0, 0, i);
    load_q(bh, qf);
  }
  FragOffW fo;
  fo.init(lane);
  RowState<HD> st[2];
  int buf = 0, c = 0;
  while (bh < bh_end) {
    // this wave's pieces of the buffer have landed (and its q fragments, and the output stores of the slab before: a COUNTED wait that leaves exactly
    // those stores in flight was tried -- hipcc's own wait for the q fragments, merged over the paths with and without stores, then waits for them anyway)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool last = c + 1 == nch;
    const int nbh_ = last ? bh + nslot : bh, nc = last ? 0 : c + 1;      // the buffer after this one
    const bool more = nbh_ < bh_end;
    // hipcc's own wait for the q fragments goes here, where it is free -- in EVERY iteration: consumed under `c == 0` only, the other path carried
    // "loads pending" into the tile loop and every score MFMA waited for vmcnt(8 .. 1), i.e. for the LDS-DMA pieces issued a tile earlier
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s_ = 0; s_ < C::KSTEPS; ++s_) asm volatile("" : "+v"(qf[b][s_]));
    if (c == 0) { st[0].reset(h); st[1].reset(h); }
    if (last && more) load_q(nbh_, qn);
    const int e_all = more ? n_dma_of(nc) : 0;
    int vm_i = 0;
    auto vm_step = [&]() {
      const int i = vm_i;
      if (i >= e_all) return;
      vm_i = i + 1;
      dma_piece(nbh_, nc, buf ^ 1, i);
    };
    if (active) {
      const unsigned char* sK = smem + buf * F::BUF;
      const unsigned char* sV = sK + F::TILE;
      const float* sBias = reinterpret_cast<const float*>(sK + 2 * F::TILE);
      const int nkeys = min(F::CK, L - c * F::CK);
      unsigned long long NZ, DEAD;
      tile_class_masks(sBias, F::CK, nkeys, have_bias, lane, NZ, DEAD);
      fwd_chunk_fast_ip<HD, 2>(sK, sV, sBias, have_bias, nkeys, rows_of(c) >> 5, NZ, DEAD, qf, st, lane, fo, [&] { vm_step(); vm_step(); });
    }
    while (vm_i < e_all) vm_step();
    if (last) {
      if (active) {
        st[0].finish_sum(); st[1].finish_sum();
        // a block whose sums are out of range: the slab is marked and k_attn_fwd2_redo (launched behind this kernel) repeats it with the running maximum
        const bool bad = __any(!st[0].sums_ok() || !st[1].sums_ok());
        if (bad && lane == 0) { redo_slab[bh] = epoch; *redo_any = epoch; }
        bf16_t* dst0 = ctx + ((size_t)(bh / H) * L) * dm + (bh % H) * HD;
        float* lse0 = lse_out ? lse_out + (size_t)bh * L : nullptr;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          RowOut<HD> o;
          o.from(st[b]);
          if (qidx[b] < L) o.store(dst0 + (size_t)qidx[b] * dm, lse0 ? lse0 + qidx[b] : nullptr, h);
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s_ = 0; s_ < C::KSTEPS; ++s_) qf[b][s_] = qn[b][s_];
    }
    bh = nbh_; c = nc; buf ^= 1;
  }
}

static int g_attn_fwd_path = -1;      // -1 automatic, 0 k_attn_fwd (per-tile maximum), 1 no-maximum kernels (fwd3 where eligible, else fwd2), 2 k_attn_fwd2 (A/B runs and tests)
extern "C" void oneprot_attn_force_fwd_path(int path) { g_attn_fwd_path = path < 0 ? -1 : (path > 2 ? 1 : path); }

template <int HD>
static int launch_fwd2(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, hipStream_t s) {
  static int ok = -1;
  if (ok < 0) ok = hipFuncSetAttribute((const void*)k_attn_fwd2<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, Fwd2<HD>::TOTAL) == hipSuccess ? 1 : 0;
  if (!ok) { (void)hipGetLastError(); return OP_EINVAL; }
  const int nqb = (L + 255) / 256;
  const int nbh8 = ((B * H + 7) / 8) * 8;
  const int waves = L >= 256 ? 8 : (L + 31) / 32;
  hipLaunchKernelGGL(k_attn_fwd2<HD>, dim3(nbh8 * nqb), dim3(64 * waves), Fwd2<HD>::TOTAL, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (bf16_t*)ctx, lse, B, H, L, nqb);
  return launch_status();
}

template <int HD>
static int launch_fwd3(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, hipStream_t s) {
  static int ok = -1, n_cu = 0;
  if (ok < 0) {
    ok = hipFuncSetAttribute((const void*)k_attn_fwd3<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, Fwd3::TOTAL + 16) == hipSuccess ? 1 : 0;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) ok = 0; else n_cu = prop.multiProcessorCount;
  }
  if (!ok) { (void)hipGetLastError(); return launch_fwd2<HD>(q, k, v, key_bias, ctx, lse, B, H, L, s); }
  const int waves = (L + 63) / 64;                        // <= 8: a wave owns 64 queries
  const int per_xcd = (B * H + 7) / 8;
  int nslot = n_cu / 8;                                   // one persistent work-group per CU
  if (nslot < 1) nslot = 1;
  if (nslot > per_xcd) nslot = per_xcd;
  hipLaunchKernelGGL(k_attn_fwd3<HD>, dim3(8 * nslot), dim3(64 * waves), Fwd3::TOTAL + 16, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (bf16_t*)ctx, lse, B, H, L, (unsigned*)dynamic_tiles_workspace());
  return launch_status();
}

template <int HD>
static int launch_fwd3w(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, hipStream_t s) {
  static int ok = -1, n_cu = 0;
  if (ok < 0) {
    ok = hipFuncSetAttribute((const void*)k_attn_fwd3w<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, Fwd3W<HD>::TOTAL) == hipSuccess ? 1 : 0;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) ok = 0; else n_cu = prop.multiProcessorCount;
  }
  if (!ok) { (void)hipGetLastError(); return launch_fwd2<HD>(q, k, v, key_bias, ctx, lse, B, H, L, s); }
  const int waves = (L + 63) / 64;                        // <= 8: a wave owns 64 queries
  const int per_xcd = (B * H + 7) / 8;
  int nslot = n_cu / 8;                                   // one persistent work-group per CU
  if (nslot < 1) nslot = 1;
  if (nslot > per_xcd) nslot = per_xcd;
  // marks of this launch: a fresh epoch (no word is ever cleared) in one of REDO_SLOTS sets, so launches in flight on other streams do not share a set
  static std::atomic<unsigned> launches{0};
  const unsigned epoch = launches.fetch_add(1u) + 1u;
  static unsigned* any0 = nullptr; static unsigned* slab0 = nullptr;      // (addresses of the marks, looked up once)
  if (!any0 || !slab0) {
    if (hipGetSymbolAddress((void**)&any0, HIP_SYMBOL(g_fwd3w_redo_any)) != hipSuccess || hipGetSymbolAddress((void**)&slab0, HIP_SYMBOL(g_fwd3w_redo_slab)) != hipSuccess)
      return OP_ELAUNCH;
  }
  unsigned* any_w = any0 + epoch % REDO_SLOTS; unsigned* slab_w = slab0 + (size_t)(epoch % REDO_SLOTS) * REDO_SLABS;
  hipLaunchKernelGGL(k_attn_fwd3w<HD>, dim3(8 * nslot), dim3(64 * waves), Fwd3W<HD>::TOTAL, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (bf16_t*)ctx, lse, B, H, L, any_w, slab_w, epoch);
  if (launch_status() != OP_OK) return OP_ELAUNCH;
  static int ok2 = -1;
  if (ok2 < 0) ok2 = hipFuncSetAttribute((const void*)k_attn_fwd2_redo<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, Fwd2<HD>::TOTAL) == hipSuccess ? 1 : 0;
  if (!ok2) return OP_ELAUNCH;
  const int nqb = (L + 255) / 256;
  hipLaunchKernelGGL(k_attn_fwd2_redo<HD>, dim3(n_cu), dim3(512), Fwd2<HD>::TOTAL, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias, (bf16_t*)ctx, lse,
                     B, H, L, nqb, (const unsigned*)any_w, (const unsigned*)slab_w, epoch);
  return launch_status();
}

template <int HD>
static int launch_fwd_nomax(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, hipStream_t s) {
  if constexpr (HD <= 32) {
    if (L <= Fwd3::ROWS && g_attn_fwd_path != 2) return launch_fwd3<HD>(q, k, v, key_bias, ctx, lse, B, H, L, s);
  } else {
    if (L <= 512 && (long)B * H <= REDO_SLABS && g_attn_fwd_path != 2) return launch_fwd3w<HD>(q, k, v, key_bias, ctx, lse, B, H, L, s);
  }
  return launch_fwd2<HD>(q, k, v, key_bias, ctx, lse, B, H, L, s);
}

extern "C" int oneprot_attn_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, int hd,
                                void* stream) {
  if (!q || !k || !v || !ctx || B <= 0 || H <= 0 || L <= 0) return OP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (g_attn_fwd_path != 0) {
    switch (hd) {
      case 16: return launch_fwd_nomax<16>(q, k, v, key_bias, ctx, lse, B, H, L, s);
      case 32: return launch_fwd_nomax<32>(q, k, v, key_bias, ctx, lse, B, H, L, s);
      case 64: return launch_fwd_nomax<64>(q, k, v, key_bias, ctx, lse, B, H, L, s);
      default: return OP_EINVAL;
    }
  }
  switch (hd) {
    case 16: return launch_fwd<16>(q, k, v, key_bias, ctx, lse, B, H, L, s);
    case 32: return launch_fwd<32>(q, k, v, key_bias, ctx, lse, B, H, L, s);
    case 64: return launch_fwd<64>(q, k, v, key_bias, ctx, lse, B, H, L, s);
    default: return OP_EINVAL;
  }
}

extern "C" int oneprot_attn_fwd_dropout(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, int hd, float p,
                                        uint64_t seed, uint64_t stream_id, void* stream) {
  AttnDrop dr;
  if (!q || !k || !v || !ctx || B <= 0 || H <= 0 || L <= 0 || attn_drop_make(p, seed, stream_id, dr) != OP_OK) return OP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (hd) {
    case 16: return launch_fwd_dropout<16>(q, k, v, key_bias, ctx, lse, B, H, L, dr, s);
    case 32: return launch_fwd_dropout<32>(q, k, v, key_bias, ctx, lse, B, H, L, dr, s);
    case 64: return launch_fwd_dropout<64>(q, k, v, key_bias, ctx, lse, B, H, L, dr, s);
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
// DROP (both split kernels): the forward ran with probability dropout, O = (keep * P / keep_prob) V.  Then dV = (keep * P / keep_prob)^T dO and
// dS = P * (keep * dP / keep_prob - delta) with the SAME delta = rowsum(dO * O); the mask is regenerated from (seed, stream, b*H+h, q, k).
template <int HD, bool DROP = false>
__global__ void __launch_bounds__(256, 2) k_attn_bwd_dq(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                     const float* __restrict__ key_bias, const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx,
                                                     const float* __restrict__ lse, float* __restrict__ delta, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                     float q_scale, bf16_t* __restrict__ dqkv, int B, int H, int L, int nqb, const AttnDrop dr = AttnDrop{0u, 0u, 0u, 1.0f}) {
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
      f32x16 dp = DROP ? zero16() : MFMA32(__builtin_bit_cast(bf8_t, ones3), __builtin_bit_cast(bf8_t, de), zero16());      // -delta[query] (DROP: subtracted after the mask)
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) {
        s = MFMA32(rd_row<HD>(sK, t * 32 + (lane & 31), st, h), qf[st], s);
        dp = MFMA32(rd_row<HD>(sV, t * 32 + (lane & 31), st, h), dof[st], dp);
      }
      if constexpr (DROP) {
        const unsigned bits = attn_keep_bits_q(qidx, kc0 + t * 32, h, bh, dr);
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = (((bits >> r) & 1u) ? dp[r] * dr.scale : 0.f) - delta_q;
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
template <int HD, bool DROP = false>
__global__ void __launch_bounds__(256, 2) k_attn_bwd_dkv(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                      const float* __restrict__ key_bias, const bf16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                      const float* __restrict__ delta, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                      bf16_t* __restrict__ dqkv, int B, int H, int L, int nkb, const AttnDrop dr = AttnDrop{0u, 0u, 0u, 1.0f}) {
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
      f32x16 dpr = zero16();                                                          // (DROP) the raw dP, masked before -delta joins it
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) {
        s = MFMA32(rd_row<HD>(sQ, t * 32 + (lane & 31), st, h), kf[st], s);          // S[query][key] + bias - lse
        if constexpr (DROP) dpr = MFMA32(rd_row<HD>(sdO, t * 32 + (lane & 31), st, h), vf[st], dpr);
        else dp = MFMA32(rd_row<HD>(sdO, t * 32 + (lane & 31), st, h), vf[st], dp);  // dP[query][key] - delta
      }
      f32x16 p, pm;
      if constexpr (DROP) {
        const unsigned bits = attn_keep_bits_k(kidx, qc0 + t * 32, h, bh, dr);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool kp = (bits >> r) & 1u;
          p[r] = __builtin_amdgcn_exp2f(s[r]);
          pm[r] = kp ? p[r] * dr.scale : 0.f;                                          // what multiplied V in the forward
          s[r] = p[r] * ((kp ? dpr[r] * dr.scale : 0.f) + dp[r]);                      // dS = P * (keep dP / keep_prob - delta)
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s[r]); s[r] = p[r] * dp[r]; }      // P, dS
      }
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const bf8_t pf = pack8(DROP ? pm : p, sb), dsf = pack8(s, sb);
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

#ifndef BWD64_PRIO
#define BWD64_PRIO 1
#endif
#ifdef BWD64_STAMP      // diagnostic build (tools/ab/bwd64_stamps.py): s_memtime at the stage boundaries of every step of one work-group's eight waves
__device__ unsigned long long g_bwd64_stamps[8 * 16 * 8];
extern "C" int oneprot_attn_debug_bwd64_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bwd64_stamps), sizeof(g_bwd64_stamps)) == hipSuccess ? 0 : -2;
}
#define BWD64_T(i)                                                                                   \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    unsigned long long t_;                                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    if (lane == 0 && blockIdx.x == 2053) g_bwd64_stamps[(wave * 16 + (blk0 + t)) * 8 + (i)] = t_;   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#else
#define BWD64_T(i) do { } while (0)
#endif

// ---- the fused backward with TWO key blocks per wave (round 4): eight waves of 64 keys ----------------------------------------------
// The 16-wave kernel above spends, per (query block, key block) tile, about as long as its MFMAs (12 x 32 cycles), its vector work (16
// exponentials + ~100 others), its LDS reads (15 KB) and its LDS stores (6 KB) take ONE AFTER THE OTHER (5 760 cycles per round of sixteen
// tiles per CU: profiles/r04_*): a wave's tile is one dependent chain and the ticket order keeps the sixteen waves in step.  Here a wave owns
// 64 keys = two key blocks and walks the same query blocks: (i) the Q / dO row fragments and transposed fragments of a query block are read
// once for two tiles; (ii) the dQ contributions of both key blocks accumulate in ONE 32 x 32 result -- one read-add-write of the block's LDS
// buffer and one ticket per TWO tiles; (iii) the two tiles are independent chains inside one wave, so the exponentials of one run under the
// MFMAs of the other without relying on the other waves of the SIMD (sched_group_barrier patterns in the step body); (iv) the dQ half of a
// step (dS^T back from the slabs, four dependent MFMAs, ticket, read-add-write: latency, no arithmetic) is deferred into the next step, behind
// its first chains.  Arithmetic per tile is that of the 16-wave kernel; the sums into dK / dV / dQ run in another fixed order (other start
// blocks of the walks, dQ adds its two key blocks first), so the results differ from it by fp32 rounding only and repeat bit for bit.
template <int HD>
__global__ void __launch_bounds__(512, 1) k_attn_bwd_fused64(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                           const float* __restrict__ key_bias, const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx,
                                                           const float* __restrict__ lse, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                           float q_scale, bf16_t* __restrict__ dqkv, int B, int H, int L, int dph) {
  typedef Cfg<HD> C;
  typedef FusedLds<HD> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;
  unsigned char* sdO = sQ + F::TILE;
  u32x4* sQE = reinterpret_cast<u32x4*>(sdO + F::TILE);
  lds_vint* sTicket = (lds_vint*)LDS_PTR(sdO + F::TILE + (F::ROWS + 1) * 16);
  float* sdQ = reinterpret_cast<float*>(sdO + F::TILE + F::QE);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, r = lane & 31;
  unsigned char* sT = reinterpret_cast<unsigned char*>(sdQ) + F::DQ + wave * 2 * F::SLAB;      // this wave's two transpose slabs (64 rows of 64 B)
  const int dm = H * HD;
  const int nbh = B * H, per_xcd = (nbh + 7) >> 3;
  const int bh = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (bh >= nbh) return;
  const int b = bh / H, head = bh - b * H;
  const int key0 = wave * 64;
  const bool active = key0 < L;

  u32x4 kr[4], vr[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
    const u32x4 z = {0u, 0u, 0u, 0u};
    kr[it] = z; vr[it] = z;
    if (key0 + row < L && ch < HD / 8) {
      kr[it] = *reinterpret_cast<const u32x4*>(k + ((size_t)bh * L + key0 + row) * HD + ch * 8);
      vr[it] = *reinterpret_cast<const u32x4*>(v + ((size_t)bh * L + key0 + row) * HD + ch * 8);
    }
  }
  float bias_k[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int kidx = key0 + 32 * kb + r;
    bias_k[kb] = kidx < L ? (key_bias ? key_bias[(size_t)b * L + kidx] : 0.f) : -INFINITY;
  }
  {
    const bf16_t* qrow0 = q + (size_t)bh * L * HD;
    const bf16_t* dorow0 = dctx + (size_t)b * L * dm + head * HD;
    const bf16_t* orow0 = ctx + (size_t)b * L * dm + head * HD;
    const int nrows = (L + 31) & ~31;
    {                                                       // four (row, chunk) items per thread, all requested before the first is used: one exposed round trip
      u32x4 rq[4], rdo[4], ro[4];
      float rl[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = threadIdx.x + 512 * it, row = idx >> 2, ch = idx & 3;
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
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = threadIdx.x + 512 * it, row = idx >> 2, ch = idx & 3;
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
            split3_bf16(row < L ? -rl[it] * LOG2E : -1.0e30f, w01, w2);
            e.x = w01; e.y = w2 | 0x3F800000u;
            split3_bf16(row < L ? -part : 0.f, w01, w2);
            e.z = w01; e.w = w2;
            sQE[row] = e;
          }
        }
      }
    }
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (threadIdx.x == 0) sQE[F::ROWS] = z;
    if (threadIdx.x < FQB) sTicket[threadIdx.x] = 0;
  }

  // ---- the wave's own K / V rows through its slabs (row fragments; K^T fragments for dQ)
  bf8_t kf[2][C::KSTEPS], vf[2][C::KSTEPS], ktf[2][2];
  {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
      *reinterpret_cast<u32x4*>(sT + row * 64 + (swz<32>(row, ch) << 4)) = kr[it];
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) kf[kb][st] = rd_row<HD>(sT + kb * F::SLAB, r, st, h);
      ktf[kb][0] = rd_tr<HD>(sT + kb * F::SLAB, 0, 0, 0, lane);
      ktf[kb][1] = rd_tr<HD>(sT + kb * F::SLAB, 0, 1, 0, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
      *reinterpret_cast<u32x4*>(sT + row * 64 + (swz<32>(row, ch) << 4)) = vr[it];
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < C::KSTEPS; ++st) vf[kb][st] = rd_row<HD>(sT + kb * F::SLAB, r, st, h);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // bookkeeping k-step operands (bias[key] - lse[query] into the scores, -delta[query] into dP): rebuilt from three registers where they are used
  // rather than held as two 4-register fragments next to the accumulators
  unsigned ky[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) ky[kb] = h == 0 ? (0x3F80u | (pack2bf(bias_k[kb], 0.f) << 16)) : 0u;
  const unsigned hm = h == 0 ? 0x3F803F80u : 0u;
  f32x16 adk[2] = {zero16(), zero16()}, adv[2] = {zero16(), zero16()};

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
  const int wr_off = r * 64 + 8 * h;
  auto tr_frag = [&](const unsigned char* base, int sb) -> bf8_t {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_off[sb][0]));
    const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_off[sb][1]));
    s16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = c[0]; o[5] = c[1]; o[6] = c[2]; o[7] = c[3];
    return __builtin_bit_cast(bf8_t, o);
  };
  __syncthreads();

  // (experiment hook, ablation builds only -- dph is 0 in the product: the second wave of every SIMD starts dph x 64 cycles late.  Measured: 0 is best,
  // the waves drift apart by themselves; tools/attn_only.py ATTN_DPH)
  if (wave >= 4 && dph < 900) for (int i = 0; i < dph; ++i) __builtin_amdgcn_s_sleep(1);
  if (active) {
    const int nkw = (L + 63) >> 6;                          // waves that own keys = visits per query block
    const int nqb = (L + 31) >> 5;                          // query blocks
    // The dQ half of a step -- dS^T fragments back from the slabs, the four dQ MFMAs, the ticketed read-add-write -- is DEFERRED into the next
    // step: its MFMAs are issued behind the next step's score / dP chains and its LDS round trips (ticket, read, add, write: ~450 cycles of
    // latency, no arithmetic) run while those chains execute, instead of standing between two steps with the matrix pipe idle.
    int p_i = -1, p_turn = 0, p_ticket = 0, p_qblk = 0;     // the deferred step: buffer, turn, rank among the block's visitors, query block
    auto dq_mfmas = [&]() -> f32x16 {
      f32x16 dq = MFMA32(ktf[0][0], tr_frag(sT, 0), zero16());       // dQ^T[d][query] = K^T dS^T, both key blocks
      dq = MFMA32(ktf[1][0], tr_frag(sT + F::SLAB, 0), dq);
      dq = MFMA32(ktf[0][1], tr_frag(sT, 1), dq);
      dq = MFMA32(ktf[1][1], tr_frag(sT + F::SLAB, 1), dq);
      return dq;
    };
    auto dq_commit = [&](f32x16& dq) {
#ifdef ONEPROT_ATTN_ABLATE      // timing only: 900 = no ticket wait, no read-add-write (what a dQ without the chain could save at most); results wrong
      if (dph == 900 || dph == 901) { asm volatile("" :: "v"(dq[0]), "v"(dq[5]), "v"(dq[10]), "v"(dq[15])); return; }
#endif
      if (lane == 0)
        while (sTicket[p_i] != p_turn) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      float4* drow = reinterpret_cast<float4*>(sdQ + (p_i * 32 + r) * F::DQP + 4 * h);
      if (p_ticket != 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 o = drow[2 * g];
          dq[4 * g] += o.x; dq[4 * g + 1] += o.y; dq[4 * g + 2] += o.z; dq[4 * g + 3] += o.w;
        }
      }
      if (p_ticket != nkw - 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) drow[2 * g] = make_float4(dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]);
      }
      asm volatile("" ::: "memory");
      if (lane == 0) sTicket[p_i] = p_turn + 1;          // LDS operations of a wave complete in order: the sums land before the counter moves
      if (p_ticket == nkw - 1) {                         // last visitor: the block's dQ rows leave from registers
        const int qidx = p_qblk * 32 + r;
        f32x16 a0[1] = {dq};
        if (qidx < L) unrope_store<HD>(a0, cosT, sinT, qidx, h, q_scale, cosT != nullptr, dqkv + ((size_t)b * L + qidx) * (3 * dm) + head * HD);
      }
    };
    int tbase = 0;
    for (int blk0 = 0; blk0 < nqb; blk0 += FQB) {
      const int nblk = min(FQB, nqb - blk0);
      const int vq = nkw / nblk, vr_ = nkw - vq * nblk;
      auto visits = [&](int blk) { return vq + (blk < vr_ ? 1 : 0); };      // waves that start their walk at block blk (< nblk): #{w < nkw : w % nblk == blk}
      int ticket = wave / nblk;
      int i = wave - ticket * nblk;
      for (int t = 0; t < (dph >= 1000 ? 0 : nblk); ++t) {      // (dph >= 1000: timing-only runs without the walk, ablation builds)
        const int qblk = blk0 + i;
        BWD64_T(0);
        const unsigned char* tQ = sQ + qblk * 2048;
        const unsigned char* tdO = sdO + qblk * 2048;
        const bf8_t qe_row = __builtin_bit_cast(bf8_t, sQE[h ? F::ROWS : qblk * 32 + r]);
        bf8_t qf[C::KSTEPS], dof[C::KSTEPS];
#pragma unroll
        for (int st = 0; st < C::KSTEPS; ++st) {
          qf[st] = *reinterpret_cast<const bf8_t*>(tQ + row_off[st]);
          dof[st] = *reinterpret_cast<const bf8_t*>(tdO + row_off[st]);
        }
        // Order of a step (one wave, in-order issue): chains of key block 0 | dQ chain of the PREVIOUS step | its ticketed read-add-write |
        // chains of key block 1 dealt between the exponentials / products / packs of key block 0 | dV / dK MFMAs of key block 0 dealt between the
        // vector work of key block 1 | dV / dK MFMAs of key block 1: the vector pipe works while the matrix pipe drains the chains.
        f32x16 sc[2], dp[2];
        const u32x4 kD = {0u, 0u, hm, hm >> 16};
        auto chains = [&](int kb) {
          const u32x4 kS = {hm, ky[kb], 0u, 0u};
          sc[kb] = MFMA32(qe_row, __builtin_bit_cast(bf8_t, kS), zero16());           // bias[key] - lse[query]
          dp[kb] = MFMA32(qe_row, __builtin_bit_cast(bf8_t, kD), zero16());           // -delta[query]
#pragma unroll
          for (int st = 0; st < C::KSTEPS; ++st) {
            sc[kb] = MFMA32(qf[st], kf[kb][st], sc[kb]);
            dp[kb] = MFMA32(dof[st], vf[kb][st], dp[kb]);
          }
        };
        // BWD64_PRIO 1 (default, round 6): the MFMA-only head of the step (key block 0's chains, the deferred dQ chain and its read-add-write) at priority 1, the
        // mixed vector / MFMA part at 0: -0.7 % per launch, bit-identical; 2 (the reverse): nothing; 0: none
#if BWD64_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#elif BWD64_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        chains(0);
        BWD64_T(1);
        f32x16 dqp = zero16();
#ifdef ONEPROT_ATTN_ABLATE      // 901: no dQ MFMAs either (and no dS^T fragments back from the slabs)
        if (p_i >= 0 && dph != 901) dqp = dq_mfmas();
#else
        if (p_i >= 0) dqp = dq_mfmas();
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (p_i >= 0) dq_commit(dqp);
        BWD64_T(2);
        __builtin_amdgcn_sched_barrier(0);
#if BWD64_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif BWD64_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        const bf8_t tdo0 = tr_frag(tdO, 0), tdo1 = tr_frag(tdO, 1), tq0 = tr_frag(tQ, 0), tq1 = tr_frag(tQ, 1);      // (requested here, not with the row fragments: 16 registers less across the deferred dQ)
        bf8_t pf[2], dsf[2];
        auto vector_part = [&](int kb) {      // P = 2^s, dS = P * dP', both packed to bf16 fragments
          f32x16 p;
#pragma unroll
          for (int e = 0; e < 16; ++e) { p[e] = __builtin_amdgcn_exp2f(sc[kb][e]); sc[kb][e] = p[e] * dp[kb][e]; }
          pf[0] = pack8(p, 0); pf[1] = pack8(p, 1);
          dsf[0] = pack8(sc[kb], 0); dsf[1] = pack8(sc[kb], 1);
        };
        auto matrix_part = [&](int kb) {      // dV^T += dO^T P, dK^T += Q^T dS; dS^T through the wave's slab kb
          adv[kb] = MFMA32(tdo0, pf[0], adv[kb]);
          adk[kb] = MFMA32(tq0, dsf[0], adk[kb]);
          adv[kb] = MFMA32(tdo1, pf[1], adv[kb]);
          adk[kb] = MFMA32(tq1, dsf[1], adk[kb]);
          const u32x4 w0 = __builtin_bit_cast(u32x4, dsf[0]), w1 = __builtin_bit_cast(u32x4, dsf[1]);
          u32x2 g0 = {w0.x, w0.y}, g1 = {w0.z, w0.w}, g2 = {w1.x, w1.y}, g3 = {w1.z, w1.w};
          unsigned char* slab = sT + kb * F::SLAB;
          *reinterpret_cast<u32x2*>(slab + wr_off + ((0 ^ sw) << 4)) = g0;
          *reinterpret_cast<u32x2*>(slab + wr_off + ((1 ^ sw) << 4)) = g1;
          *reinterpret_cast<u32x2*>(slab + wr_off + ((2 ^ sw) << 4)) = g2;
          *reinterpret_cast<u32x2*>(slab + wr_off + ((3 ^ sw) << 4)) = g3;
        };
        // key block 1's chains (6 MFMAs) between key block 0's vector instructions (~56)
        chains(1);
        vector_part(0);
#pragma unroll
        for (int n = 0; n < 6; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // key block 0's dV / dK MFMAs (4) between key block 1's vector instructions
        matrix_part(0);
        vector_part(1);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x002, 14, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        matrix_part(1);
        asm volatile("" ::: "memory");
        BWD64_T(3);
        p_i = i; p_turn = tbase + ticket; p_ticket = ticket; p_qblk = qblk;
        i = i + 1 == nblk ? 0 : i + 1;
        ticket += visits(i);
      }
      tbase += nkw;
    }
    if (p_i >= 0) {
      f32x16 dql = dq_mfmas();
      dq_commit(dql);
    }
  }
  if (active) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int kidx = key0 + 32 * kb + r;
      f32x16 a1[1] = {adk[kb]}, a2[1] = {adv[kb]};
      unsigned char* slab = sT + kb * F::SLAB;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        bf16_t* img = reinterpret_cast<bf16_t*>(slab + r * 64);
        if (which == 0) unrope_store<HD>(a1, cosT, sinT, kidx < L ? kidx : 0, h, 0.6931471805599453f, cosT != nullptr, img);
        else unrope_store<HD>(a2, cosT, sinT, 0, h, 1.0f, false, img);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int idx = lane + 64 * it, row = idx >> 2, ch = idx & 3;
          const u32x4 w = *reinterpret_cast<const u32x4*>(slab + row * 64 + ch * 16);
          if (key0 + 32 * kb + row < L && ch < HD / 8)
            *reinterpret_cast<u32x4*>(dqkv + ((size_t)b * L + key0 + 32 * kb + row) * (3 * dm) + (which + 1) * dm + head * HD + ch * 8) = w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
}

static int g_attn_bwd_path = -1;      // -1 automatic, 0 split kernels, 1 fused 16-wave kernel, 2 fused 8-wave kernel (two key blocks per wave) where eligible (A/B runs and tests)
extern "C" void oneprot_attn_force_bwd_path(int path) { g_attn_bwd_path = path < 0 ? -1 : (path > 2 ? 2 : path); }
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
    // automatic choice (tools/attn_only.py with ATTN_L, 256 x 20 heads, us per launch: L = 128 / 256 / 320 / 384 / 448 / 512 -- split 151 / 354 / 555 /
    // 642 / 876 / 1004, fused 16 waves 174 / 366 / 510 / 619 / 769 / 885, fused 8 waves 240 / 429 / 547 / 638 / 748 / 845): the fused kernels pay
    // once their work-group has a key block for most of its waves
    const int path = g_attn_bwd_path >= 0 ? g_attn_bwd_path : (L <= 288 ? 0 : (L <= 416 ? 1 : 2));
    if (L <= 512 && path != 0) {
      // the fused kernel needs the 143 KB dynamic-LDS opt-in; a device / driver that refuses it takes the split kernels from then on (forced
      // fused path: the refusal is the caller's error)
      static int fused_ok = -1;
      if (fused_ok < 0)
        fused_ok = hipFuncSetAttribute((const void*)k_attn_bwd_fused<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, FusedLds<HD>::TOTAL) == hipSuccess ? 1 : 0;
      if (!fused_ok) { (void)hipGetLastError(); if (g_attn_bwd_path > 0) return OP_EINVAL; }
      else if (path == 2) {
        static int f64_ok = -1;
        if (f64_ok < 0)
          f64_ok = hipFuncSetAttribute((const void*)k_attn_bwd_fused64<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, FusedLds<HD>::TOTAL) == hipSuccess ? 1 : 0;
        if (!f64_ok) { (void)hipGetLastError(); if (g_attn_bwd_path > 0) return OP_EINVAL; }
        else {
        hipLaunchKernelGGL(k_attn_bwd_fused64<HD>, dim3(((B * H + 7) / 8) * 8), dim3(512), FusedLds<HD>::TOTAL, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                           (const bf16_t*)ctx, (const bf16_t*)dctx, lse, cosT, sinT, q_scale, (bf16_t*)dqkv, B, H, L, g_attn_bwd_ablate);
        return launch_status();
        }
      } else {
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

template <int HD>
static int launch_bwd_dropout(const void* q, const void* k, const void* v, const float* key_bias, const void* ctx, const void* dctx, const float* lse, float* delta,
                              const float* cosT, const float* sinT, float q_scale, void* dqkv, int B, int H, int L, const AttnDrop& dr, hipStream_t s) {
  const int nb = (L + 127) / 128;
  const int nbh8 = ((B * H + 7) / 8) * 8;
  const size_t lds = (size_t)2 * KC * Cfg<HD>::ROWB + (KC + 1) * 16;
  hipLaunchKernelGGL((k_attn_bwd_dq<HD, true>), dim3(nbh8 * nb), dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (const bf16_t*)ctx, (const bf16_t*)dctx, lse, delta, cosT, sinT, q_scale, (bf16_t*)dqkv, B, H, L, nb, dr);
  hipLaunchKernelGGL((k_attn_bwd_dkv<HD, true>), dim3(nbh8 * nb), dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_bias,
                     (const bf16_t*)dctx, lse, (const float*)delta, cosT, sinT, (bf16_t*)dqkv, B, H, L, nb, dr);
  return launch_status();
}
extern "C" int oneprot_attn_bwd_dropout(const void* q, const void* k, const void* v, const float* key_bias, const void* ctx, const void* dctx, const float* lse,
                                        const float* rope_cos, const float* rope_sin, float q_scale, void* dqkv, void* workspace, int B, int H, int L, int hd,
                                        float p, uint64_t seed, uint64_t stream_id, void* stream) {
  AttnDrop dr;
  if (!q || !k || !v || !ctx || !dctx || !lse || !dqkv || !workspace || B <= 0 || H <= 0 || L <= 0 || attn_drop_make(p, seed, stream_id, dr) != OP_OK) return OP_EINVAL;
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return OP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)workspace;
  switch (hd) {
    case 16: return launch_bwd_dropout<16>(q, k, v, key_bias, ctx, dctx, lse, delta, rope_cos, rope_sin, q_scale, dqkv, B, H, L, dr, s);
    case 32: return launch_bwd_dropout<32>(q, k, v, key_bias, ctx, dctx, lse, delta, rope_cos, rope_sin, q_scale, dqkv, B, H, L, dr, s);
    case 64: return launch_bwd_dropout<64>(q, k, v, key_bias, ctx, dctx, lse, delta, rope_cos, rope_sin, q_scale, dqkv, B, H, L, dr, s);
    default: return OP_EINVAL;
  }
}

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
