// Shared by the NT-GEMM translation units (gemm_nt.hip, gemm_nt8.hip): launch arguments, store helpers and the DIRECT-store epilogues.
#pragma once
#include "common.h"
#include "../../include/oneprot_hip.h"
#include <type_traits>
#include <utility>

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
  int csplit;                   // 8-phase kernels: XCDs that share a set of row panels and divide its column tiles between them (1, 2 or 4)
  int sup_m, sup_n;             // L2 super-tile of the per-tile kernels: sup_m row panels x sup_n column tiles per XCD at a time
  int nt_store;                 // output stores non-temporal (streamed past the L2 instead of displacing the operand panels and W)
  // G8_EPI_RESID_LN (8-phase kernels only): out0 = x_out fp32 = acc + bias + aux; out1 = bf16 LayerNorm(x_out) with gamma = cos, beta = sin, eps = q_scale;
  // out2 = fp32 [2][M]: mean | rstd (or null).  A row's statistics are completed across the column tiles through ln_part (gemm_epi8.h).
  void* ln_part;                // [M][8] x 16 bytes {tag, mean, M2, ~tag}: per row, one partial per wave column of every column tile (uncached memory, inside the sched workspace)
  int ln_slots;                 // partials per row = column tiles x wave columns
  int ln_poll_max;              // bound of the wait for the other column tiles' partials, in polls
  // Work queues and launch bookkeeping of the persistent kernels: the caller's "sched workspace" (sched_ws.h; include/oneprot_hip.h: oneprot_sched_workspace_*), or null.
  // With it the last work-group to leave a launch resets the queues and advances the launch epoch ON THE DEVICE (a replayed graph gets fresh tags);
  // `dyn`: tiles are drawn from the per-XCD queues instead of the static list u = q * g8n + w.
  unsigned* sched;
  int dyn;
};
#define G8_EPI_RESID_LN 6      // internal to gemm_nt8.hip / gemm_epi8.h (entry point oneprot_gemm_bf16_nt_resid_ln8), not part of the public epilogue enum
// 16- / 8-byte output stores with the launch's cache policy (wave-uniform branch)
__device__ __forceinline__ void gst(u32x4* p, u32x4 v, int nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void gst(u32x2* p, u32x2 v, int nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void gst(float* p, float a, float b, float c, float d, int nt) {
  const f32x4 v = {a, b, c, d};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = v;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- DIRECT form (full tiles only).  The MFMA is issued with the operand roles swapped -- a = weight fragment, b = activation fragment -- so
// the accumulator of lane (c = lane & 15, q = lane >> 4) is C[token c][feature slot q*4 + r]: four consecutive slots of ONE token row per
// register quad.  Which logical column a slot is, is decided when the weight tile is staged: LDS row slot s of the tile receives global weight
// row nmap(s) (the LDS-DMA source address is per lane, so the permutation costs nothing in the loop and leaves the bank pattern of the fragment
// reads untouched).  bf16 outputs use the PAIR map (slots of tiles 2p / 2p+1 interleaved in runs of four: a lane owns 8 consecutive columns
// = one 16-byte store, four lanes cover 64 contiguous bytes of a row); fp32 outputs and QKV/RoPE use the NATURAL map (4 consecutive fp32 = 16 bytes
// per lane; the RoPE partner column +-hd/2 is the same register of tile j^1 (hd 32) / j^2 (hd 64) of the SAME lane).  No LDS staging of the
// accumulators, no waits on the LDS queue, no reuse of the operand ring: 1 ds_write_b32 + 1/4 ds_read_b128 per element less than the staged form.
template <int EPI> struct DirectMap { static constexpr bool PAIR = (EPI == ONEPROT_EPI_BF16 || EPI == ONEPROT_EPI_BIAS_GELU || EPI == ONEPROT_EPI_GELU_BWD); };
// logical column (within the wave's NTW*16-column block) of slot rho of 16-slot tile j
template <bool PAIR> __device__ __forceinline__ constexpr int direct_nmap(int j, int rho) {
  return PAIR ? ((j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3)) : (j * 16 + rho);
}

// erf-GELU with ONE output (frozen towers, validation): gelu(x) = x Phi(x) = max(x, 0) - |x| h(|x|), h(a) = 0.5 erfc(a / sqrt2) = Phi(-a).
// log2 h is smooth (-a^2/2 log2 e plus a slowly varying term), so h = 2^q(a) with q a polynomial in a (weighted minimax fits on [0, 6], the weight
// following |x| h, i.e. what the error does to the output; beyond 6 q keeps falling -- negative leading coefficient -- and h -> 0):
//   degree 5 (shipped): relative error of h <= 1.8e-4 for a < 3, |gelu error| <= 1.3e-5 everywhere = 1/11 of the bf16 half-ulp of the stored value;
//   degree 7: 3.3e-6 / 3.5e-7, the level of the Abramowitz-Stegun form of the two-output epilogue (-DGELU_FWD_FORM=7; 0 = that rational form).
// One transcendental and 5 (7) fma + max + fma per element against 12 plain + 2 transcendental instructions: the epilogues that inline this are
// bound by vector-instruction issue.  Eight values at once (one lane's 16-byte store) with the polynomial on v_pk_fma_f32: a vector
// instruction of a wave costs the SIMD 4 cycles whether it carries one fp32 per lane or two (every plain instruction removed per element is worth
// 9-11 us of a 335 M-element launch; the packed form of the degree-7 polynomial took FFN-1 from 465 to 434 us; hipcc packs the bias add and the
// final fma by itself but not a Horner chain).  The two-output form (gelu_fwd_and_code8 below) keeps the rational form, whose exponential is
// shared with the derivative.
#ifndef GELU_FWD_FORM
#define GELU_FWD_FORM 5
#endif
__device__ __forceinline__ void gelu_fwd_only8(float (&v)[8]) {
#if GELU_FWD_FORM == 0
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = v[e];
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));
    const float ex = __builtin_amdgcn_exp2f(x * x * -0.72134752f);
    float poly = fmaf(0.5307027145f, t, -0.7265760135f);
    poly = fmaf(poly, t, 0.7107068705f);
    poly = fmaf(poly, t, -0.142248368f);
    poly = fmaf(poly, t, 0.127414796f);
    v[e] = x * (0.5f + copysignf(0.5f - poly * t * ex, x));
  }
#else
#if GELU_FWD_FORM == 7
  constexpr int DEG = 7;
  constexpr float C[8] = {-1.3735314885e-06f, 5.3708949533e-05f, -8.7965580671e-04f, 8.3529787465e-03f, -5.3730911583e-02f, -4.5861420912e-01f, -1.1512197648e+00f, -9.9999524375e-01f};
#else
  constexpr int DEG = 5;
  constexpr float C[6] = {-2.9390228453e-04f, 5.6289777323e-03f, -4.7738369713e-02f, -4.6464447721e-01f, -1.1489399185e+00f, -1.0001595425e+00f};
#endif
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2_t x = {v[e], v[e + 1]};
    const f32x2_t a = {fabsf(v[e]), fabsf(v[e + 1])};
    f32x2_t q = __builtin_elementwise_fma((f32x2_t){C[0], C[0]}, a, (f32x2_t){C[1], C[1]});
#pragma unroll
    for (int k = 2; k <= DEG; ++k) q = __builtin_elementwise_fma(q, a, (f32x2_t){C[k], C[k]});
    const f32x2_t h = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    float r0, r1;
    asm("v_max_f32 %0, 0, %1" : "=v"(r0) : "v"(x.x));       // (plain max: fmaxf adds a canonicalising v_max of x with itself)
    asm("v_max_f32 %0, 0, %1" : "=v"(r1) : "v"(x.y));
    const f32x2_t g = __builtin_elementwise_fma(-a, h, (f32x2_t){r0, r1});
    v[e] = g.x; v[e + 1] = g.y;
  }
#endif
}

// gelu'(z) travels from the FFN-1 forward epilogue (ONEPROT_EPI_BIAS_GELU, out1) to the FFN-2 data-gradient epilogue (ONEPROT_EPI_GELU_BWD, aux) as
// ONE BYTE per element (round 5; it was a bf16 tensor: 671 MB written and read back per trainable layer at cfg-2, now 335 MB each way).  The
// derivative is bounded, gelu' in [-0.1290, 1.1290], and enters the backward only as an elementwise factor, so a uniform code is the better use of
// the bits: c = rint(192 g' + 25) in [0, 242], g' = (c - 25) / 192 -- 0 and 1 are exact (saturated units carry no bias), the error is uniform with
// |e| <= 1/384 = 0.0026 (rms 0.0015), where bf16 has |e| <= 0.0039 for the units near 1 that carry the gradient and spends its bits near 0.
constexpr float GELU_GRAD_CODE_SCALE = 192.0f, GELU_GRAD_CODE_ZERO = 25.0f;
// The two-output form in one go: v[e] <- gelu(v[e]), returns the eight gelu' codes.  Abramowitz-Stegun erfc (common.h: gelu_fwd_and_grad; its
// exponential is also the Gaussian factor of the derivative) written on packed fp32 instructions, two elements per instruction, with the code's affine
// map folded into the derivative's last fma: 10 plain + 2 transcendental instructions per element where hipcc's own packing of the scalar source left
// 17.5 (these epilogues are bound by vector-instruction issue: 4 cycles per instruction and SIMD whatever it carries).
__device__ __forceinline__ u32x2 gelu_fwd_and_code8(float (&v)[8]) {
  unsigned code[2] = {0u, 0u};
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2_t x = {v[e], v[e + 1]};
    const f32x2_t t = {__builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x.x), 1.0f)), __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x.y), 1.0f))};
    const f32x2_t ea = (x * x) * (f32x2_t){-0.72134752f, -0.72134752f};
    const f32x2_t ex = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};             // exp(-x^2 / 2)
    f32x2_t poly = __builtin_elementwise_fma((f32x2_t){0.5307027145f, 0.5307027145f}, t, (f32x2_t){-0.7265760135f, -0.7265760135f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2_t){0.7107068705f, 0.7107068705f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2_t){-0.142248368f, -0.142248368f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2_t){0.127414796f, 0.127414796f});
    const f32x2_t h = (poly * t) * ex;                                                           // 0.5 erfc(|x| / sqrt2)
    const f32x2_t s = (f32x2_t){0.5f, 0.5f} - h;
    const f32x2_t cdf = (f32x2_t){0.5f, 0.5f} + (f32x2_t){copysignf(s.x, x.x), copysignf(s.y, x.y)};      // Phi(x)
    const f32x2_t g = x * cdf;
    // 192 gelu'(x) + 25 = 192 Phi + 25 + (192 / sqrt(2 pi)) x exp(-x^2 / 2)
    const f32x2_t cf = __builtin_elementwise_fma(x * ex, (f32x2_t){76.596919837f, 76.596919837f},
                                                 __builtin_elementwise_fma(cdf, (f32x2_t){GELU_GRAD_CODE_SCALE, GELU_GRAD_CODE_SCALE}, (f32x2_t){GELU_GRAD_CODE_ZERO, GELU_GRAD_CODE_ZERO}));
    v[e] = g.x; v[e + 1] = g.y;
    code[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(cf.x, e & 3, code[e >> 2]);
    code[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(cf.y, (e & 3) + 1, code[e >> 2]);
  }
  return (u32x2){code[0], code[1]};
}

// v[e] *= gelu'(element e) for the eight codes of a lane
__device__ __forceinline__ void gelu_grad_apply8(float (&v)[8], unsigned lo, unsigned hi) {
  constexpr float S = 1.0f / GELU_GRAD_CODE_SCALE, Z = -GELU_GRAD_CODE_ZERO / GELU_GRAD_CODE_SCALE;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] *= fmaf((float)((lo >> (8 * e)) & 0xffu), S, Z);
    v[4 + e] *= fmaf((float)((hi >> (8 * e)) & 0xffu), S, Z);
  }
}

// QKV/RoPE tail of the direct epilogue (natural map: lane owns columns [j*16 + q*4, +4) of tile j).  H*hd is a multiple of 64 = the wave's
// column block, so the whole wave sits in one section (q / k / v); with HD in {32, 64} a head is 2 or 4 tiles of the block, the position of
// tile j inside its head and its rotation partner (tile j ^ (HD/32)) are compile-time constants -- the partner value is a register of the SAME lane.
template <int HD, int MT, int NTW>
__device__ __forceinline__ void rope_store_direct(const GemmArgs& p, f32x4 (&acc)[MT][NTW], int mrow0, int ncol0, int q) {
  static_assert((NTW * 16) % HD == 0, "a head must not straddle the wave's column block");
  constexpr int HALF = HD / 2, JP = HALF / 16;
  const int dm = p.H * HD;
  const int sec = __builtin_amdgcn_readfirstlane(ncol0 / dm);
  const int head0 = (ncol0 - sec * dm) / HD;
  bf16_t* dst = (bf16_t*)(sec == 0 ? p.out0 : (sec == 1 ? p.out1 : p.out2));
  const float sc = sec == 0 ? p.q_scale : 1.0f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gm = mrow0 + i * 16;
    const int b = gm / p.L, l = gm - b * p.L;
    const float* cs_row = p.cos + (size_t)l * HALF + q * 4;
    const float* sn_row = p.sin + (size_t)l * HALF + q * 4;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int jt = (j * 16) % HD;                    // column of the tile inside its head (compile-time after unrolling)
      const bool lo = jt < HALF;
      const int head = head0 + (j * 16) / HD;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (sec < 2) {
        const f32x4 pv = acc[i][lo ? j + JP : j - JP];
        const int jj = lo ? jt : jt - HALF;
        const float4 cs = *reinterpret_cast<const float4*>(cs_row + jj);
        const float4 sn = *reinterpret_cast<const float4*>(sn_row + jj);
        const float sp = lo ? -sc : sc;
        v[0] = (v[0] * sc) * cs.x + (pv[0] * sp) * sn.x; v[1] = (v[1] * sc) * cs.y + (pv[1] * sp) * sn.y;
        v[2] = (v[2] * sc) * cs.z + (pv[2] * sp) * sn.z; v[3] = (v[3] * sc) * cs.w + (pv[3] * sp) * sn.w;
      }
      u32x2 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<u32x2*>(dst + (((size_t)b * p.H + head) * p.L + l) * HD + jt + q * 4) = w;
    }
  }
}

// Starting values of the direct-form accumulators: the bias of the lane's four consecutive columns per tile and, for the residual epilogue, the
// fp32 residual tile itself (acc = resid + bias + sum of products: one rounding order among equals; `out0` may alias the residual because every
// lane reads exactly the elements it later writes).
template <int EPI, int MT, int NTW>
__device__ __forceinline__ void direct_init_acc(const GemmArgs& p, f32x4 (&acc)[MT][NTW], int m0, int n0, int wr, int wc, int lane) {
  constexpr bool PAIR = DirectMap<EPI>::PAIR;
  const int c = lane & 15, q = lane >> 4;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int gc = n0 + wc * (NTW * 16) + direct_nmap<PAIR>(j, q * 4);
    f32x4 bj = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + gc); bj = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][j] = bj;
  }
}
// the residual tile of the lane: requested BEFORE the first K-slices are staged, added AFTER their requests have been issued, so that both are in
// flight together (the compiler waits for ordinary loads where their results are first used)
template <int EPI, int MT, int NTW>
__device__ __forceinline__ void direct_resid_load(const GemmArgs& p, float4 (&rs)[MT][NTW], int m0, int n0, int wr, int wc, int lane) {
  if constexpr (EPI == ONEPROT_EPI_BIAS_RESID) {
    const int c = lane & 15, q = lane >> 4;
    const int mrow0 = m0 + wr * (MT * 16) + c, ncol0 = n0 + wc * (NTW * 16);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float* rrow = (const float*)p.aux + (size_t)(mrow0 + i * 16) * p.N + ncol0 + q * 4;
#pragma unroll
      for (int j = 0; j < NTW; ++j) rs[i][j] = *reinterpret_cast<const float4*>(rrow + j * 16);
    }
  }
}
template <int EPI, int MT, int NTW>
__device__ __forceinline__ void direct_resid_add(f32x4 (&acc)[MT][NTW], const float4 (&rs)[MT][NTW]) {
  if constexpr (EPI == ONEPROT_EPI_BIAS_RESID) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) { acc[i][j][0] += rs[i][j].x; acc[i][j][1] += rs[i][j].y; acc[i][j][2] += rs[i][j].z; acc[i][j][3] += rs[i][j].w; }
  }
}

template <int EPI, int MT, int NTW>
__device__ __forceinline__ void gemm_epilogue_direct(const GemmArgs& p, f32x4 (&acc)[MT][NTW], int m0, int n0, int wr, int wc, int lane) {
  const int c = lane & 15, q = lane >> 4;
  const int mrow0 = m0 + wr * (MT * 16) + c;                 // + i*16
  const int ncol0 = n0 + wc * (NTW * 16);
  if constexpr (DirectMap<EPI>::PAIR) {
    const bool with_grad = (EPI == ONEPROT_EPI_BIAS_GELU) && p.out1 != nullptr;      // wave-uniform
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const size_t rowoff = (size_t)(mrow0 + i * 16) * p.N + ncol0 + q * 8;
#pragma unroll
      for (int pp = 0; pp < NTW / 2; ++pp) {
        const size_t o = rowoff + pp * 32;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * pp][r]; v[4 + r] = acc[i][2 * pp + 1][r]; }
        if (EPI == ONEPROT_EPI_BIAS_GELU) {
          if (with_grad) {
            gst(reinterpret_cast<u32x2*>((unsigned char*)p.out1 + o), gelu_fwd_and_code8(v), p.nt_store);
          } else {
            gelu_fwd_only8(v);
          }
        } else if (EPI == ONEPROT_EPI_GELU_BWD) {
          const u32x2 z = *reinterpret_cast<const u32x2*>((const unsigned char*)p.aux + o);
          gelu_grad_apply8(v, z.x, z.y);
        }
        u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
        gst(reinterpret_cast<u32x4*>((bf16_t*)p.out0 + o), w, p.nt_store);
      }
    }
  } else if constexpr (EPI == ONEPROT_EPI_QKV_ROPE) {
    if (p.hd == 32) rope_store_direct<32, MT, NTW>(p, acc, mrow0, ncol0, q);
    else if constexpr ((NTW * 16) % 64 == 0) rope_store_direct<64, MT, NTW>(p, acc, mrow0, ncol0, q);      // (hosts only pick wave blocks that hold whole heads)
  } else {      // fp32 outputs: ONEPROT_EPI_F32, ONEPROT_EPI_BIAS_RESID (natural map, 16 bytes per lane per tile)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const size_t rowoff = (size_t)(mrow0 + i * 16) * p.N + ncol0 + q * 4;
      // (BIAS_RESID: the residual tile was added to the accumulators' starting values by direct_init_acc -- its read overlaps the first K-slices'
      // flight instead of sitting, latency exposed, in front of the stores)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        gst((float*)p.out0 + rowoff + j * 16, v.x, v.y, v.z, v.w, p.nt_store);
        if (EPI == ONEPROT_EPI_BIAS_RESID && p.out1) {
          u32x2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
          gst(reinterpret_cast<u32x2*>((bf16_t*)p.out1 + rowoff + j * 16), w, p.nt_store);
        }
      }
    }
  }
}

// 8-phase form (gemm_nt8.hip).  cfg 0: 256 x 256 tiles, cfg 1: 256 x 320 tiles; G8_NOT_ELIGIBLE when the problem is not made of whole tiles.
#define G8_NOT_ELIGIBLE (-100)
int launch_gemm8(int epi, const GemmArgs& a, int cfg, long min_tiles, hipStream_t s);
int gemm8_ln_eligible(long M, int N, int K);
int launch_gemm8_ln(GemmArgs a, void* sched_ws, size_t sched_ws_bytes, hipStream_t s);      // G8_EPI_RESID_LN on 256 x 320 tiles
size_t sched_workspace_bytes(long M_max);
void* dynamic_tiles_workspace();                      // what oneprot_dynamic_tiles was given (null: static tile lists)
