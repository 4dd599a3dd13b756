// HBM-bound row kernels: embedding gather (+token-dropout rescale, pad zeroing), LayerNorm fwd/bwd,
// fused final-LayerNorm + pooling fwd/bwd, column reductions, casts.
//
// Layout rules: residual stream fp32 [T, d]; normalised activations bf16 [T, d];
// one wave64 owns one row, lane l owns float4 columns l, l+64, ... so every access is a 16-byte (fp32) or
// 8-byte (bf16) coalesced vector access and all row statistics are wave reductions (no LDS, no barriers).
#include "common.h"
#include "../../include/oneprot_hip.h"

#define MAXV 8                 // float4s per lane -> rows up to 2048 wide
#define ROWS_PER_BLOCK 4       // 256 threads = 4 waves = 4 rows in flight per block

// --------------------------------------------------------------------------------------------------------
// Embedding forward  (hf modeling_esm.py:224-271; ref call site sequence_encoder.py:78)
// x[b,l,:] = W[id] * 0.88 / (1 - n_mask_b / n_valid_b)  ; 0 where id==mask ; 0 where id==pad
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_embed_fwd(const long long* __restrict__ ids, const float* __restrict__ W, float* __restrict__ x,
                                                   float* __restrict__ row_scale, int L, int d, int vocab, int pad_id, int mask_id,
                                                   int token_dropout, int tok_per_block) {
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  __shared__ float s_cnt[2][4];
  __shared__ float s_scale;
  float nvalid = 0.f, nmask = 0.f;
  for (int l = tid; l < L; l += 256) {
    const long long id = ids[(size_t)b * L + l];
    nvalid += (id != pad_id);
    nmask += (id == mask_id);
  }
  nvalid = wave_sum(nvalid); nmask = wave_sum(nmask);
  if ((tid & 63) == 0) { s_cnt[0][tid >> 6] = nvalid; s_cnt[1][tid >> 6] = nmask; }
  __syncthreads();
  if (tid == 0) {
    const float nv = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
    const float nm = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
    float sc = 1.0f;
    if (token_dropout) sc = (1.0f - 0.15f * 0.8f) / (1.0f - nm / nv);
    s_scale = sc;
    if (chunk == 0 && row_scale) row_scale[b] = sc;
  }
  __syncthreads();
  const float sc = s_scale;
  const int nv4 = d >> 2;
  const int l0 = chunk * tok_per_block;
  const int l1 = min(L, l0 + tok_per_block);
  const int wave = tid >> 6, lane = tid & 63;
  for (int l = l0 + wave; l < l1; l += 4) {
    const long long id = ids[(size_t)b * L + l];
    const bool zero = (id == pad_id) || (token_dropout && id == mask_id) || id < 0 || id >= vocab;
    const float4* src = reinterpret_cast<const float4*>(W + (size_t)(zero ? 0 : id) * d);
    float4* dst = reinterpret_cast<float4*>(x + ((size_t)b * L + l) * d);
    for (int v = lane; v < nv4; v += 64) {
      float4 e = src[v];
      if (zero) e = make_float4(0.f, 0.f, 0.f, 0.f);
      else { e.x *= sc; e.y *= sc; e.z *= sc; e.w *= sc; }
      dst[v] = e;
    }
  }
}

extern "C" int oneprot_esm_embed_fwd(const int64_t* ids, const float* table, float* x, float* row_scale, int B, int L, int d, int vocab,
                                     int pad_id, int mask_id, int token_dropout, void* stream) {
  if (!ids || !table || !x || B <= 0 || L <= 0 || d <= 0 || (d & 3)) return OP_EINVAL;
  const int tpb = 32;
  dim3 grid((L + tpb - 1) / tpb, B);
  hipLaunchKernelGGL(k_embed_fwd, grid, dim3(256), 0, (hipStream_t)stream, (const long long*)ids, table, x, row_scale, L, d, vocab, pad_id,
                     mask_id, token_dropout, tpb);
  return launch_status();
}

// Embedding backward for small vocabularies (ESM: 33 / 54 rows): per-block private accumulators in LDS
// (one thread per column => no atomics), partial tables to a workspace, then k_colsum_partials reduces.
// dW[id] += row_scale[b] * dx[b,l,:]  for valid, non-mask tokens.
__global__ void __launch_bounds__(256) k_embed_bwd_small(const long long* __restrict__ ids, const float* __restrict__ dx,
                                                         const float* __restrict__ row_scale, float* __restrict__ partial, int T, int L, int d,
                                                         int vocab, int pad_id, int mask_id, int token_dropout, int tok_per_block) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];   // [vocab][256]
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y;
  for (int v = 0; v < vocab; ++v) s_acc[v * 256 + threadIdx.x] = 0.f;
  const int t0 = chunk * tok_per_block, t1 = min(T, t0 + tok_per_block);
  if (col < d) {
    // eight tokens' loads are in flight before the first LDS add (one token per iteration was one exposed memory latency per token: 607 us at cfg-2)
    int t = t0;
    for (; t + 7 < t1; t += 8) {
      long long id[8]; float v[8], sc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        id[j] = ids[t + j];
        v[j] = dx[(size_t)(t + j) * d + col];
        sc[j] = row_scale ? row_scale[(t + j) / L] : 1.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (id[j] == pad_id || (token_dropout && id[j] == mask_id) || id[j] < 0 || id[j] >= vocab) continue;   // block-uniform branch
        s_acc[(int)id[j] * 256 + threadIdx.x] += sc[j] * v[j];
      }
    }
    for (; t < t1; ++t) {
      const long long id = ids[t];
      if (id == pad_id || (token_dropout && id == mask_id) || id < 0 || id >= vocab) continue;
      const float sc = row_scale ? row_scale[t / L] : 1.0f;
      s_acc[(int)id * 256 + threadIdx.x] += sc * dx[(size_t)t * d + col];
    }
    for (int v = 0; v < vocab; ++v) partial[((size_t)chunk * vocab + v) * d + col] = s_acc[v * 256 + threadIdx.x];
  }
}

// out[j] (+)= sum_p partial[p][j]   for j in [0, n)
__global__ void __launch_bounds__(256) k_reduce_partials(const float* __restrict__ partial, float* __restrict__ out, int nparts, size_t n, int accumulate) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * n + j];
  out[j] = accumulate ? out[j] + s : s;
}

#define EMBED_BWD_CHUNKS 512      // token chunks = blocks per 256-column slab (128 left the 256 CUs with 1.5 blocks each, every block a serial walk over 1 024 tokens)
// the same for many partials: 8 threads per output, each sums every 8th partial (64 loads in flight per output instead of one dependent walk over
// all of them), then a fixed-order combination through LDS -- deterministic
__global__ void __launch_bounds__(256) k_reduce_partials_wide(const float* __restrict__ partial, float* __restrict__ out, int nparts, size_t n, int accumulate) {
  __shared__ float sh[8][32];
  const int jj = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const size_t j = (size_t)blockIdx.x * 32 + jj;
  float s = 0.f;
  if (j < n)
    for (int p = pg; p < nparts; p += 8) s += partial[(size_t)p * n + j];
  sh[pg][jj] = s;
  __syncthreads();
  if (pg == 0 && j < n) {
    const float t = ((sh[0][jj] + sh[1][jj]) + (sh[2][jj] + sh[3][jj])) + ((sh[4][jj] + sh[5][jj]) + (sh[6][jj] + sh[7][jj]));
    out[j] = accumulate ? out[j] + t : t;
  }
}

extern "C" size_t oneprot_esm_embed_bwd_workspace(int T, int d, int vocab) {
  const int chunks = EMBED_BWD_CHUNKS;
  (void)T;
  return (size_t)chunks * vocab * d * sizeof(float);
}

extern "C" int oneprot_esm_embed_bwd(const int64_t* ids, const float* dx, const float* row_scale, float* dtable, void* workspace, int B, int L, int d,
                                     int vocab, int pad_id, int mask_id, int token_dropout, int accumulate, void* stream) {
  if (!ids || !dx || !dtable || !workspace || (d & 3) || vocab <= 0 || vocab > 160) return OP_EINVAL;   // 160*256*4 = 160 KiB of LDS
  const int T = B * L, chunks = EMBED_BWD_CHUNKS;
  const int tpb = (T + chunks - 1) / chunks;
  dim3 grid((d + 255) / 256, chunks);
  const size_t lds = (size_t)vocab * 256 * sizeof(float);
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_embed_bwd_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_embed_bwd_small, grid, dim3(256), lds, (hipStream_t)stream, (const long long*)ids, dx, row_scale, (float*)workspace, T, L, d,
                     vocab, pad_id, mask_id, token_dropout, tpb);
  const size_t n = (size_t)vocab * d;
  hipLaunchKernelGGL(k_reduce_partials_wide, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dtable, chunks, n, accumulate);
  return launch_status();
}

// --------------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma + beta   (hf modeling_esm.py:429,518,552; nn.LayerNorm)
// IN_BF16: input is bf16 instead of fp32.  Outputs: optional bf16 copy and/or fp32 copy, optional mean/rstd.
// --------------------------------------------------------------------------------------------------------
template <int IN_BF16, int NV>
__global__ void __launch_bounds__(256) k_layernorm_fwd(const void* __restrict__ xin, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16_t* __restrict__ y_bf16, float* __restrict__ y_f32, float* __restrict__ mean_out,
                                                       float* __restrict__ rstd_out, int T, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int nv4 = d >> 2;
  const float inv_d = 1.0f / (float)d;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < T; row += gridDim.x * ROWS_PER_BLOCK) {
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        if (IN_BF16) {
          const u32x2 p = reinterpret_cast<const u32x2*>((const bf16_t*)xin + (size_t)row * d)[c];
          v[i] = make_float4(bflo(p.x), bfhi(p.x), bflo(p.y), bfhi(p.y));
        } else {
          v[i] = reinterpret_cast<const float4*>((const float*)xin + (size_t)row * d)[c];
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + dd * dd);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c];
        const float4 bb = reinterpret_cast<const float4*>(beta)[c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + bb.x;
        o.y = (v[i].y - mean) * rstd * g.y + bb.y;
        o.z = (v[i].z - mean) * rstd * g.z + bb.z;
        o.w = (v[i].w - mean) * rstd * g.w + bb.w;
        if (y_f32) reinterpret_cast<float4*>(y_f32 + (size_t)row * d)[c] = o;
        if (y_bf16) {
          u32x2 p; p.x = pack2bf(o.x, o.y); p.y = pack2bf(o.z, o.w);
          reinterpret_cast<u32x2*>(y_bf16 + (size_t)row * d)[c] = p;
        }
      }
    }
  }
}

static inline int ln_grid(int T) {
  const int blocks = (T + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return blocks < 256 * 16 ? blocks : 256 * 16;
}

extern "C" int oneprot_layernorm_fwd(const void* x, int x_is_bf16, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean,
                                     float* rstd, int64_t T, int d, float eps, void* stream) {
  if (!x || !gamma || !beta || (!y_bf16 && !y_f32) || T <= 0 || d <= 0 || (d & 3) || d > MAXV * 256) return OP_EINVAL;
  const int nv = (d / 4 + 63) / 64;
#define LNF(B16, N) hipLaunchKernelGGL((k_layernorm_fwd<B16, N>), dim3(ln_grid((int)T)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd, (int)T, d, eps)
#define LNF_NV(B16) do { if (nv <= 1) LNF(B16, 1); else if (nv == 2) LNF(B16, 2); else if (nv == 3) LNF(B16, 3); else if (nv == 4) LNF(B16, 4); else if (nv == 5) LNF(B16, 5); else LNF(B16, 8); } while (0)
  if (x_is_bf16) LNF_NV(1); else LNF_NV(0);
#undef LNF_NV
#undef LNF
  return launch_status();
}

// --------------------------------------------------------------------------------------------------------
// LayerNorm backward.
//   g = dy * gamma ; dx = rstd * (g - mean_d(g) - xhat * mean_d(g * xhat)) ; dgamma += dy * xhat ; dbeta += dy
// DY_MODE 0: dy bf16 [T,d]   1: dy fp32 [T,d]   2: dy[t,:] = dpool[t / L, :] * wrow[t]  (pooled-gradient broadcast)
// dx_out = (add_to ? add_to[t] : 0) + dx   (add_to may alias dx_out: residual-gradient accumulation in place)
// dgamma/dbeta: per-wave register partials over the rows the wave visits -> partial[(block*4+wave)][2][d]
// --------------------------------------------------------------------------------------------------------
template <int DY_MODE, int X_BF16, int NV>
__global__ void __launch_bounds__(256) k_layernorm_bwd(const void* __restrict__ dy, const float* __restrict__ wrow, int L, const void* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                       const float* add_to, float* dx_out, bf16_t* __restrict__ dx_bf16, float* __restrict__ partial, int T, int d) {
  extern __shared__ __attribute__((aligned(16))) float s_part[];     // [4 waves][2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv4 = d >> 2;
  const float inv_d = 1.0f / (float)d;
  float4 dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dg[i] = make_float4(0, 0, 0, 0);
    db[i] = make_float4(0, 0, 0, 0);
  }
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < T; row += gridDim.x * ROWS_PER_BLOCK) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float w = 1.0f;
    if (DY_MODE == 2) w = wrow[row];
    float4 xh[NV], g[NV];
#ifndef LN_BWD_LATE_ADD
    // the residual-gradient rows are requested together with x and dy: one exposed round trip per row instead of two (they are only needed after
    // the two wave reductions, but a load issued there starts a second latency that the 4-5 resident waves per SIMD do not cover)
    float4 av[NV];
    if (add_to) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv4) av[i] = reinterpret_cast<const float4*>(add_to + (size_t)row * d)[c];
      }
    }
#endif
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        float4 xv;
        if (X_BF16) {
          const u32x2 p = reinterpret_cast<const u32x2*>((const bf16_t*)x + (size_t)row * d)[c];
          xv = make_float4(bflo(p.x), bfhi(p.x), bflo(p.y), bfhi(p.y));
        } else {
          xv = reinterpret_cast<const float4*>((const float*)x + (size_t)row * d)[c];
        }
        float4 dyv;
        if (DY_MODE == 0) {
          const u32x2 p = reinterpret_cast<const u32x2*>((const bf16_t*)dy + (size_t)row * d)[c];
          dyv = make_float4(bflo(p.x), bfhi(p.x), bflo(p.y), bfhi(p.y));
        } else if (DY_MODE == 1) {
          dyv = reinterpret_cast<const float4*>((const float*)dy + (size_t)row * d)[c];
        } else {
          dyv = reinterpret_cast<const float4*>((const float*)dy + (size_t)(row / L) * d)[c];
          dyv.x *= w; dyv.y *= w; dyv.z *= w; dyv.w *= w;
        }
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        const float4 gm = reinterpret_cast<const float4*>(gamma)[c];          // (re-read per row from L1: 12 registers fewer -> one more wave per SIMD)
        g[i] = make_float4(dyv.x * gm.x, dyv.y * gm.y, dyv.z * gm.z, dyv.w * gm.w);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        dg[i].x += dyv.x * xh[i].x; dg[i].y += dyv.y * xh[i].y; dg[i].z += dyv.z * xh[i].z; dg[i].w += dyv.w * xh[i].w;
        db[i].x += dyv.x; db[i].y += dyv.y; db[i].z += dyv.z; db[i].w += dyv.w;
      }
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        float4 o;
        o.x = rstd * (g[i].x - m1 - xh[i].x * m2);
        o.y = rstd * (g[i].y - m1 - xh[i].y * m2);
        o.z = rstd * (g[i].z - m1 - xh[i].z * m2);
        o.w = rstd * (g[i].w - m1 - xh[i].w * m2);
        if (add_to) {
#ifdef LN_BWD_LATE_ADD
          const float4 a = reinterpret_cast<const float4*>(add_to + (size_t)row * d)[c];
#else
          const float4 a = av[i];
#endif
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        reinterpret_cast<float4*>(dx_out + (size_t)row * d)[c] = o;
        if (dx_bf16) { u32x2 pk; pk.x = pack2bf(o.x, o.y); pk.y = pack2bf(o.z, o.w); reinterpret_cast<u32x2*>(dx_bf16 + (size_t)row * d)[c] = pk; }
      }
    }
  }
  if (partial) {
    float* pw = s_part + (size_t)wave * 2 * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        reinterpret_cast<float4*>(pw)[c] = dg[i];
        reinterpret_cast<float4*>(pw + d)[c] = db[i];
      }
    }
    __syncthreads();
    float* pg = partial + (size_t)blockIdx.x * 2 * d;
    for (int j = threadIdx.x; j < 2 * d; j += 256)
      pg[j] = (s_part[j] + s_part[2 * d + j]) + (s_part[4 * d + j] + s_part[6 * d + j]);
  }
}

#define LN_BWD_BLOCKS 1024      // 4 blocks per CU: 259 us against 261-273 with 2048 (half the partial matrix for k_ln_reduce), 272 with 768, 324 with 512 (tools/ab/ln_time.py)
extern "C" size_t oneprot_layernorm_bwd_workspace(int d) { return (size_t)LN_BWD_BLOCKS * 2 * d * sizeof(float); }

// dgamma_dbeta: [2][d] laid out as dgamma then dbeta (the two may be non-adjacent: pass both pointers)
// 16 columns per block, 64 part-slices per column (1024 threads: the 2048 x 2d partial matrix is 10 MB and the kernel is bound by loads in
// flight, not bandwidth; fixed summation order => deterministic)
__global__ void __launch_bounds__(1024) k_ln_reduce(const float* __restrict__ partial, float* __restrict__ dgamma, float* __restrict__ dbeta, int nparts, int d, int accumulate) {
  __shared__ float s_sl[64][17];
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float s = 0.f;
  if (j < 2 * d)
    for (int p = sl; p < nparts; p += 64) s += partial[(size_t)p * 2 * d + j];
  s_sl[sl][c] = s;
  __syncthreads();
  if (sl < 4) {                       // 4 x 16 partial sums, then 4 -> 1
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s_sl[sl * 16 + q][c];
    s_sl[sl][c] = t;
  }
  __syncthreads();
  if (sl == 0 && j < 2 * d) {
    const float t = (s_sl[0][c] + s_sl[1][c]) + (s_sl[2][c] + s_sl[3][c]);
    float* out = j < d ? dgamma + j : dbeta + (j - d);
    *out = accumulate ? *out + t : t;
  }
}

extern "C" int oneprot_layernorm_bwd(const void* dy, int dy_mode, const float* wrow, int L, const void* x, int x_is_bf16, const float* gamma,
                                     const float* mean, const float* rstd, const float* add_to, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                                     void* workspace, int64_t T, int d, int accumulate_param_grads, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || T <= 0 || (d & 3) || d > MAXV * 256) return OP_EINVAL;
  if (dy_mode < 0 || dy_mode > 2 || (dy_mode == 2 && (!wrow || L <= 0))) return OP_EINVAL;
  if ((dgamma || dbeta) && !(dgamma && dbeta && workspace)) return OP_EINVAL;
  int blocks = (int)((T + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  if (blocks > LN_BWD_BLOCKS) blocks = LN_BWD_BLOCKS;
  float* partial = dgamma ? (float*)workspace : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const int nv = (d / 4 + 63) / 64;
  const size_t lds = partial ? (size_t)4 * 2 * d * sizeof(float) : 0;
  if (lds > 64 * 1024) return OP_EINVAL;
#define LAUNCH_LNB(M, XB, N) hipLaunchKernelGGL((k_layernorm_bwd<M, XB, N>), dim3(blocks), dim3(256), lds, s, dy, wrow, L, x, gamma, mean, rstd, add_to, dx, (bf16_t*)dx_bf16, partial, (int)T, d)
#define LNB_NV(M, XB) do { if (nv <= 1) LAUNCH_LNB(M, XB, 1); else if (nv == 2) LAUNCH_LNB(M, XB, 2); else if (nv == 3) LAUNCH_LNB(M, XB, 3); else if (nv == 4) LAUNCH_LNB(M, XB, 4); else if (nv == 5) LAUNCH_LNB(M, XB, 5); else LAUNCH_LNB(M, XB, 8); } while (0)
  if (x_is_bf16) { if (dy_mode == 0) LNB_NV(0, 1); else if (dy_mode == 1) LNB_NV(1, 1); else LNB_NV(2, 1); }
  else { if (dy_mode == 0) LNB_NV(0, 0); else if (dy_mode == 1) LNB_NV(1, 0); else LNB_NV(2, 0); }
#undef LNB_NV
#undef LAUNCH_LNB
  if (dgamma)
    hipLaunchKernelGGL(k_ln_reduce, dim3((2 * d + 15) / 16), dim3(1024), 0, s, (const float*)workspace, dgamma, dbeta, blocks, d, accumulate_param_grads);
  return launch_status();
}

// --------------------------------------------------------------------------------------------------------
// Fused final LayerNorm + pooling forward (hf modeling_esm.py:552 + ref base_encoder.py:109-126).
//   mode 0 (mean): pooled[b] = sum_l valid[b,l] * LN(x[b,l]) / n_valid[b]   -- CLS/EOS included
//   mode 1 (cls):  pooled[b] = LN(x[b,0])
// One 512-thread block per sequence; wave w handles rows w, w+8, ...; lane owns float4 columns; per-wave register
// accumulators are combined through LDS.  Also emits mean/rstd per token and wrow[t] = valid/n_valid (mode 0) or [l==0]
// (mode 1), which is exactly the pooled-gradient broadcast weight used by oneprot_layernorm_bwd(dy_mode=2).
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_lnpool_fwd(const float* __restrict__ x, const long long* __restrict__ ids, int pad_id, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ pooled, float* __restrict__ mean_out,
                                                    float* __restrict__ rstd_out, float* __restrict__ wrow, bf16_t* __restrict__ hidden_bf16,
                                                    float* __restrict__ hidden_f32, int L, int d, float eps, int mode) {
  extern __shared__ __attribute__((aligned(16))) float s_pool[];    // [8][d]
  __shared__ float s_n[8];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv4 = d >> 2;
  const float inv_d = 1.0f / (float)d;
  float cnt = 0.f;
  for (int l = threadIdx.x; l < L; l += 512) cnt += (ids[(size_t)b * L + l] != pad_id);
  cnt = wave_sum(cnt);
  if (lane == 0) s_n[wave] = cnt;
  __syncthreads();
  float nvalid = 0.f;
  for (int w = 0; w < 8; ++w) nvalid += s_n[w];
  const float inv_n = 1.0f / nvalid;
  float4 acc[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) acc[i] = make_float4(0, 0, 0, 0);
  const int lend = (mode == 1 && !hidden_bf16 && !hidden_f32 && !mean_out) ? 1 : L;
  // the NEXT row of the wave is requested before this row's two wave reductions (a wave walks 64 rows of its sequence one after the other: without
  // the prefetch every row paid a memory round trip on top of the reductions -- 260 us per launch at cfg-2)
  float4 vn[MAXV];
  bool validn = false;
  auto request = [&](int l) {
    if (l < lend) {
      const size_t row = (size_t)b * L + l;
      validn = ids[row] != pad_id;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv4) vn[i] = reinterpret_cast<const float4*>(x + row * d)[c];
      }
    }
  };
  request(wave);
  for (int l = wave; l < lend; l += 8) {
    const size_t row = (size_t)b * L + l;
    const bool valid = validn;
    const float wt = (mode == 0) ? (valid ? inv_n : 0.f) : (l == 0 ? 1.f : 0.f);
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) { v[i] = vn[i]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
    }
    request(l + 8);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) { const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean; q += (a * a + bb * bb) + (cc * cc + dd * dd); }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
      if (wrow) wrow[row] = wt;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c];
        const float4 be = reinterpret_cast<const float4*>(beta)[c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + be.x; o.y = (v[i].y - mean) * rstd * g.y + be.y;
        o.z = (v[i].z - mean) * rstd * g.z + be.z; o.w = (v[i].w - mean) * rstd * g.w + be.w;
        if (hidden_f32) reinterpret_cast<float4*>(hidden_f32 + row * d)[c] = o;
        if (hidden_bf16) { u32x2 p; p.x = pack2bf(o.x, o.y); p.y = pack2bf(o.z, o.w); reinterpret_cast<u32x2*>(hidden_bf16 + row * d)[c] = p; }
        acc[i].x += wt * o.x; acc[i].y += wt * o.y; acc[i].z += wt * o.z; acc[i].w += wt * o.w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) reinterpret_cast<float4*>(s_pool + (size_t)wave * d)[c] = acc[i];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < d; j += 512) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += s_pool[(size_t)w * d + j];
    pooled[(size_t)b * d + j] = s;
  }
}

extern "C" int oneprot_lnpool_fwd(const float* x, const int64_t* ids, int pad_id, const float* gamma, const float* beta, float* pooled, float* mean,
                                  float* rstd, float* wrow, void* hidden_bf16, float* hidden_f32, int B, int L, int d, float eps, int mode, void* stream) {
  if (!x || !ids || !gamma || !beta || !pooled || (d & 3) || d > MAXV * 256 || mode < 0 || mode > 1) return OP_EINVAL;
  hipLaunchKernelGGL(k_lnpool_fwd, dim3(B), dim3(512), (size_t)8 * d * sizeof(float), (hipStream_t)stream, x, (const long long*)ids, pad_id, gamma, beta, pooled,
                     mean, rstd, wrow, (bf16_t*)hidden_bf16, hidden_f32, L, d, eps, mode);
  return launch_status();
}

// --------------------------------------------------------------------------------------------------------
// BERT embeddings: x = LayerNorm(word[id] + pos[l] + type[0])   (hf modeling_bert.py:53-108; eval mode, dropout off)
// one wave per token, lane owns float4 columns (same scheme as LayerNorm)
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bert_embed(const long long* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                                    const float* __restrict__ type0, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ x_f32, bf16_t* __restrict__ x_bf16, int T, int L, int d, int vocab, float eps) {
  const int lane = threadIdx.x & 63;
  const int nv4 = d >> 2;
  const float inv_d = 1.0f / (float)d;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < T; row += gridDim.x * ROWS_PER_BLOCK) {
    long long id = ids[row];
    if (id < 0 || id >= vocab) id = 0;
    const int l = row % L;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        const float4 a = reinterpret_cast<const float4*>(word + (size_t)id * d)[c];
        const float4 b = reinterpret_cast<const float4*>(pos + (size_t)l * d)[c];
        const float4 t = reinterpret_cast<const float4*>(type0)[c];
        v[i] = make_float4(a.x + t.x + b.x, a.y + t.y + b.y, a.z + t.z + b.z, a.w + t.w + b.w);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) { const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean; q += (a * a + b * b) + (cc * cc + dd * dd); }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c];
        const float4 be = reinterpret_cast<const float4*>(beta)[c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + be.x; o.y = (v[i].y - mean) * rstd * g.y + be.y;
        o.z = (v[i].z - mean) * rstd * g.z + be.z; o.w = (v[i].w - mean) * rstd * g.w + be.w;
        if (x_f32) reinterpret_cast<float4*>(x_f32 + (size_t)row * d)[c] = o;
        if (x_bf16) { u32x2 pk; pk.x = pack2bf(o.x, o.y); pk.y = pack2bf(o.z, o.w); reinterpret_cast<u32x2*>(x_bf16 + (size_t)row * d)[c] = pk; }
      }
    }
  }
}
extern "C" int oneprot_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma, const float* beta,
                                      float* x_f32, void* x_bf16, int B, int L, int d, int vocab, float eps, void* stream) {
  if (!ids || !word || !pos || !type0 || !gamma || !beta || (!x_f32 && !x_bf16) || B <= 0 || L <= 0 || (d & 3) || d > MAXV * 256) return OP_EINVAL;
  hipLaunchKernelGGL(k_bert_embed, dim3(ln_grid(B * L)), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, word, pos, type0, gamma, beta, x_f32,
                     (bf16_t*)x_bf16, B * L, L, d, vocab, eps);
  return launch_status();
}

// Pooling without a LayerNorm in front (BERT's last layer output is already normalised): mode 0 masked mean, 1 CLS (ref base_encoder.py:109-126)
__global__ void __launch_bounds__(256) k_pool_fwd(const float* __restrict__ x, const long long* __restrict__ ids, int pad_id, float* __restrict__ pooled, int L, int d, int mode) {
  const int b = blockIdx.x;
  __shared__ float s_n[4];
  float cnt = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) cnt += (ids[(size_t)b * L + l] != pad_id);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = cnt;
  __syncthreads();
  const float inv_n = 1.0f / (s_n[0] + s_n[1] + s_n[2] + s_n[3]);
  for (int j = threadIdx.x; j < d; j += 256) {
    float s = 0.f;
    if (mode == 1) s = x[(size_t)b * L * d + j];
    else {
      for (int l = 0; l < L; ++l)
        if (ids[(size_t)b * L + l] != pad_id) s += x[((size_t)b * L + l) * d + j];
      s *= inv_n;
    }
    pooled[(size_t)b * d + j] = s;
  }
}
extern "C" int oneprot_pool_fwd(const float* x, const int64_t* ids, int pad_id, float* pooled, int B, int L, int d, int mode, void* stream) {
  if (!x || !ids || !pooled || B <= 0 || L <= 0 || d <= 0 || mode < 0 || mode > 1) return OP_EINVAL;
  hipLaunchKernelGGL(k_pool_fwd, dim3(B), dim3(256), 0, (hipStream_t)stream, x, (const long long*)ids, pad_id, pooled, L, d, mode);
  return launch_status();
}

// backward of k_pool_fwd: g[b,l,:] = dpooled[b,:] / n_b on non-pad tokens (mean) or dpooled[b,:] at l = 0 (CLS), 0 elsewhere; fp32 + bf16 copy
__global__ void __launch_bounds__(256) k_pool_bwd(const float* __restrict__ dpooled, const long long* __restrict__ ids, int pad_id, float* __restrict__ g,
                                                  bf16_t* __restrict__ g16, int L, int d, int mode) {
  const int b = blockIdx.x;
  __shared__ float s_n[4];
  float cnt = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) cnt += (ids[(size_t)b * L + l] != pad_id);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = cnt;
  __syncthreads();
  const float inv_n = 1.0f / (s_n[0] + s_n[1] + s_n[2] + s_n[3]);
  const int nv4 = d >> 2;
  for (int idx = threadIdx.x; idx < L * nv4; idx += 256) {
    const int l = idx / nv4, c = idx - l * nv4;
    float w;
    if (mode == 1) w = (l == 0) ? 1.0f : 0.0f;
    else w = (ids[(size_t)b * L + l] != pad_id) ? inv_n : 0.0f;
    const float4 dp = reinterpret_cast<const float4*>(dpooled + (size_t)b * d)[c];
    const float4 o = make_float4(dp.x * w, dp.y * w, dp.z * w, dp.w * w);
    const size_t off = ((size_t)b * L + l) * d + 4 * c;
    *reinterpret_cast<float4*>(g + off) = o;
    if (g16) { u32x2 pk; pk.x = pack2bf(o.x, o.y); pk.y = pack2bf(o.z, o.w); *reinterpret_cast<u32x2*>(g16 + off) = pk; }
  }
}
extern "C" int oneprot_pool_bwd(const float* dpooled, const int64_t* ids, int pad_id, float* g, void* g_bf16, int B, int L, int d, int mode, void* stream) {
  if (!dpooled || !ids || !g || B <= 0 || L <= 0 || d <= 0 || (d & 3) || mode < 0 || mode > 1) return OP_EINVAL;
  hipLaunchKernelGGL(k_pool_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, dpooled, (const long long*)ids, pad_id, g, (bf16_t*)g_bf16, L, d, mode);
  return launch_status();
}

// --------------------------------------------------------------------------------------------------------
// Embedding-table gradients for a large vocabulary (BERT: 30522 rows): rows of dx that share a token id are summed in a fixed order.
// The host sorts the token ids (stable) and passes the permutation plus the start of every run of equal ids; one block per run adds
// its rows in sorted order -> deterministic, no float atomics.  Rows of the table that no token used are left untouched (caller zeroes).
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_embed_scatter_sorted(const float* __restrict__ dx, const long long* __restrict__ perm, const long long* __restrict__ seg_start,
                                                              const long long* __restrict__ seg_row, long long n_tokens, int n_seg, int d, int skip_row,
                                                              float* __restrict__ dtable) {
  const int s = blockIdx.x;
  const long long row = seg_row[s];
  if (row == skip_row) return;                      // padding_idx row receives no gradient (nn.Embedding semantics)
  const long long j0 = seg_start[s], j1 = (s + 1 < n_seg) ? seg_start[s + 1] : n_tokens;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (long long j = j0; j < j1; ++j) acc += dx[(size_t)perm[j] * d + c];
    dtable[(size_t)row * d + c] = acc;
  }
}
extern "C" int oneprot_embed_scatter_sorted(const float* dx, const int64_t* perm, const int64_t* seg_start, const int64_t* seg_row, int64_t n_tokens, int n_seg,
                                            int d, int skip_row, float* dtable, void* stream) {
  if (!dx || !perm || !seg_start || !seg_row || !dtable || n_tokens <= 0 || n_seg <= 0 || d <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_embed_scatter_sorted, dim3(n_seg), dim3(256), 0, (hipStream_t)stream, dx, (const long long*)perm, (const long long*)seg_start,
                     (const long long*)seg_row, (long long)n_tokens, n_seg, d, skip_row, dtable);
  return launch_status();
}

// out[j] = sum_r x[r, j] for an fp32 [R, n] matrix (position-embedding gradient = sum over the batch; fixed order, one thread per column)
__global__ void __launch_bounds__(256) k_rowsum_f32(const float* __restrict__ x, float* __restrict__ out, int R, long long n) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) acc += x[(size_t)r * n + j];
  out[j] = acc;
}
extern "C" int oneprot_rowsum_f32(const float* x, float* out, int R, int64_t n, void* stream) {
  if (!x || !out || R <= 0 || n <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_rowsum_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, R, (long long)n);
  return launch_status();
}

// --------------------------------------------------------------------------------------------------------
// Attention1dPooling (ref base_encoder.py:88-103; MaskedConv1d with kernel 1 = one dot product per token):
//   s_l = x_l . w + b ; s_l = -inf where the token is padding ; a = softmax_l(s) ; pooled = sum_l a_l x_l
// One 16-wave block per sequence; scores and weights live in LDS ([L] floats); x (fp32) is read twice, 16 bytes per lane:
//   pass 1  a wave per row (two rows in flight): the row's dot product with a [d] vector
//   pass 2  a wave per (64 float4 columns, row group): sum_l weight_l x_l over its rows, groups combined through LDS in a fixed order
// (round 6: the 4-wave form read 4 bytes per lane in pass 2 and kept one row per wave in flight: 700 / 870 us for 128 x 512 x 1280 fwd / bwd)
// --------------------------------------------------------------------------------------------------------
constexpr int AP_WAVES = 16;
// pass 1: out_l = x_l . v (+ bias) for the rows of one sequence; lane 0 of the row's wave hands (l, value) to `put`
template <class Put>
__device__ __forceinline__ void ap_row_dots(const float* __restrict__ xb, const float* __restrict__ v, int L, int d, int lane, int wave, Put put) {
  const int nv4 = d >> 2;
  const float4* v4 = reinterpret_cast<const float4*>(v);
  for (int l = wave; l < L; l += 2 * AP_WAVES) {
    const int l1 = l + AP_WAVES;
    const bool two = l1 < L;
    const float4* r0 = reinterpret_cast<const float4*>(xb + (size_t)l * d);
    const float4* r1 = reinterpret_cast<const float4*>(xb + (size_t)(two ? l1 : l) * d);
    float s0 = 0.f, s1 = 0.f;
    for (int c = lane; c < nv4; c += 64) {
      const float4 a = r0[c], bq = r1[c], wv = v4[c];
      s0 += (a.x * wv.x + a.y * wv.y) + (a.z * wv.z + a.w * wv.w);
      s1 += (bq.x * wv.x + bq.y * wv.y) + (bq.z * wv.z + bq.w * wv.w);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { put(l, s0); if (two) put(l1, s1); }
  }
}
// pass 2: out[j] = sum_l wt[l] x[l][j] (wt in LDS); with DX also dx[l][j] = at[l] g[j] + wt[l] w[j].  s_part: [groups][d] floats of LDS.
template <bool DX>
__device__ __forceinline__ void ap_weighted_sum(const float* __restrict__ xb, const float* s_wt, float* s_part, float* __restrict__ out, int L, int d, int lane, int wave,
                                                const float* __restrict__ at, const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dxb) {
  const int nv4 = d >> 2;
  const int nch = (nv4 + 63) >> 6;                                   // chunks of 64 float4 columns
  const int G = nch >= AP_WAVES ? 1 : AP_WAVES / nch;                // row groups that work on a chunk side by side
  for (int c0 = 0; c0 < nch; c0 += AP_WAVES) {                       // (one trip unless d > 4096)
    const int c = c0 + (G > 1 ? wave % nch : wave), grp = G > 1 ? wave / nch : 0;
    const int j4 = c * 64 + lane;
    const bool live = c < nch && grp < G && j4 < nv4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      float4 gv = {0.f, 0.f, 0.f, 0.f}, wv = gv;
      if constexpr (DX) { gv = reinterpret_cast<const float4*>(g)[j4]; wv = reinterpret_cast<const float4*>(w)[j4]; }
      const float4* col = reinterpret_cast<const float4*>(xb) + j4;
      int l = grp;
#pragma unroll 1
      for (; l + 3 * G < L; l += 4 * G) {                            // four rows in flight
        const float4 x0 = col[(size_t)l * nv4], x1 = col[(size_t)(l + G) * nv4], x2 = col[(size_t)(l + 2 * G) * nv4], x3 = col[(size_t)(l + 3 * G) * nv4];
        const float a0 = s_wt[l], a1 = s_wt[l + G], a2 = s_wt[l + 2 * G], a3 = s_wt[l + 3 * G];
        acc.x += a0 * x0.x; acc.y += a0 * x0.y; acc.z += a0 * x0.z; acc.w += a0 * x0.w;
        acc.x += a1 * x1.x; acc.y += a1 * x1.y; acc.z += a1 * x1.z; acc.w += a1 * x1.w;
        acc.x += a2 * x2.x; acc.y += a2 * x2.y; acc.z += a2 * x2.z; acc.w += a2 * x2.w;
        acc.x += a3 * x3.x; acc.y += a3 * x3.y; acc.z += a3 * x3.z; acc.w += a3 * x3.w;
        if constexpr (DX) {
          float4* dcol = reinterpret_cast<float4*>(dxb) + j4;
          const float p0 = at[l], p1 = at[l + G], p2 = at[l + 2 * G], p3 = at[l + 3 * G];
          dcol[(size_t)l * nv4] = float4{p0 * gv.x + a0 * wv.x, p0 * gv.y + a0 * wv.y, p0 * gv.z + a0 * wv.z, p0 * gv.w + a0 * wv.w};
          dcol[(size_t)(l + G) * nv4] = float4{p1 * gv.x + a1 * wv.x, p1 * gv.y + a1 * wv.y, p1 * gv.z + a1 * wv.z, p1 * gv.w + a1 * wv.w};
          dcol[(size_t)(l + 2 * G) * nv4] = float4{p2 * gv.x + a2 * wv.x, p2 * gv.y + a2 * wv.y, p2 * gv.z + a2 * wv.z, p2 * gv.w + a2 * wv.w};
          dcol[(size_t)(l + 3 * G) * nv4] = float4{p3 * gv.x + a3 * wv.x, p3 * gv.y + a3 * wv.y, p3 * gv.z + a3 * wv.z, p3 * gv.w + a3 * wv.w};
        }
      }
      for (; l < L; l += G) {
        const float4 x0 = col[(size_t)l * nv4];
        const float a0 = s_wt[l];
        acc.x += a0 * x0.x; acc.y += a0 * x0.y; acc.z += a0 * x0.z; acc.w += a0 * x0.w;
        if constexpr (DX) {
          const float p0 = at[l];
          (reinterpret_cast<float4*>(dxb) + j4)[(size_t)l * nv4] = float4{p0 * gv.x + a0 * wv.x, p0 * gv.y + a0 * wv.y, p0 * gv.z + a0 * wv.z, p0 * gv.w + a0 * wv.w};
        }
      }
      if (G > 1) reinterpret_cast<float4*>(s_part + (size_t)grp * d)[j4] = acc;
    }
    if (G > 1) {
      __syncthreads();
      if (live && grp == 0) {
        for (int q = 1; q < G; ++q) {
          const float4 o = reinterpret_cast<const float4*>(s_part + (size_t)q * d)[j4];
          acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
      }
      __syncthreads();
    }
    if (live && grp == 0) reinterpret_cast<float4*>(out)[j4] = acc;
  }
}
static size_t ap_lds_bytes(int L, int d) {
  const int nch = ((d >> 2) + 63) >> 6, G = nch >= AP_WAVES ? 1 : AP_WAVES / nch;
  return ((size_t)((L + 3) & ~3) + (G > 1 ? (size_t)G * d : 0)) * sizeof(float);
}
__device__ __forceinline__ float ap_block_sum(float v, float* s_red, int lane, int wave) {      // every thread gets the total; s_red: [AP_WAVES]
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int q = 0; q < AP_WAVES; ++q) t += s_red[q];
  return t;
}

__global__ void __launch_bounds__(AP_WAVES * 64) k_attnpool_fwd(const float* __restrict__ x, const long long* __restrict__ ids, int pad_id, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ pooled, float* __restrict__ attn, int L, int d) {
  extern __shared__ __attribute__((aligned(16))) float s_a[];      // [L rounded to 4] weights, then [groups][d] partial sums
  __shared__ float s_red[AP_WAVES];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xb = x + (size_t)b * L * d;
  const long long* idb = ids + (size_t)b * L;
  const float bs = bias[0];
  ap_row_dots(xb, w, L, d, lane, wave, [&](int l, float s) { s_a[l] = (idb[l] != pad_id) ? s + bs : -INFINITY; });
  __syncthreads();
  float mx = -INFINITY;
  for (int l = threadIdx.x; l < L; l += AP_WAVES * 64) mx = fmaxf(mx, s_a[l]);
  mx = wave_max(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = s_red[0];
  for (int q = 1; q < AP_WAVES; ++q) mx = fmaxf(mx, s_red[q]);
  float se = 0.f;
  for (int l = threadIdx.x; l < L; l += AP_WAVES * 64) { const float e = __expf(s_a[l] - mx); s_a[l] = e; se += e; }
  const float inv = 1.0f / ap_block_sum(se, s_red, lane, wave);
  for (int l = threadIdx.x; l < L; l += AP_WAVES * 64) { const float a = s_a[l] * inv; s_a[l] = a; if (attn) attn[(size_t)b * L + l] = a; }
  __syncthreads();
  ap_weighted_sum<false>(xb, s_a, s_a + ((L + 3) & ~3), pooled + (size_t)b * d, L, d, lane, wave, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int oneprot_attnpool_fwd(const float* x, const int64_t* ids, int pad_id, const float* w, const float* bias, float* pooled, float* attn, int B, int L,
                                    int d, void* stream) {
  if (!x || !ids || !w || !bias || !pooled || B <= 0 || L <= 0 || (d & 3) || ap_lds_bytes(L, d) > 60 * 1024) return OP_EINVAL;
  hipLaunchKernelGGL(k_attnpool_fwd, dim3(B), dim3(AP_WAVES * 64), ap_lds_bytes(L, d), (hipStream_t)stream, x, (const long long*)ids, pad_id, w, bias, pooled, attn, L, d);
  return launch_status();
}
// backward: da_l = x_l . dp ; ds = a * (da - sum a da) ; dw_partial[b] = sum_l ds_l x_l ; db_partial[b] = sum_l ds_l ;
//           dx_l = a_l dp + ds_l w (optional)
__global__ void __launch_bounds__(AP_WAVES * 64) k_attnpool_bwd(const float* __restrict__ x, const float* __restrict__ attn, const float* __restrict__ w,
                                                                const float* __restrict__ dpooled, float* __restrict__ dw_part, float* __restrict__ db_part,
                                                                float* __restrict__ dx, int L, int d) {
  extern __shared__ __attribute__((aligned(16))) float s_ds[];     // [L rounded to 4], then [groups][d] partial sums
  __shared__ float s_red[AP_WAVES];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xb = x + (size_t)b * L * d;
  const float* dp = dpooled + (size_t)b * d;
  const float* ab = attn + (size_t)b * L;
  ap_row_dots(xb, dp, L, d, lane, wave, [&](int l, float s) { s_ds[l] = s; });
  __syncthreads();
  float dot = 0.f;
  for (int l = threadIdx.x; l < L; l += AP_WAVES * 64) dot += ab[l] * s_ds[l];
  const float tot = ap_block_sum(dot, s_red, lane, wave);
  float dbs = 0.f;
  for (int l = threadIdx.x; l < L; l += AP_WAVES * 64) { const float v = ab[l] * (s_ds[l] - tot); s_ds[l] = v; dbs += v; }
  dbs = ap_block_sum(dbs, s_red, lane, wave);                       // (its barriers also publish s_ds)
  if (threadIdx.x == 0) db_part[b] = dbs;
  float* part = s_ds + ((L + 3) & ~3);
  if (dx) ap_weighted_sum<true>(xb, s_ds, part, dw_part + (size_t)b * d, L, d, lane, wave, ab, dp, w, dx + (size_t)b * L * d);
  else    ap_weighted_sum<false>(xb, s_ds, part, dw_part + (size_t)b * d, L, d, lane, wave, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int oneprot_attnpool_bwd(const float* x, const float* attn, const float* w, const float* dpooled, float* dw, float* db, float* dx, void* workspace,
                                    int B, int L, int d, void* stream) {
  if (!x || !attn || !w || !dpooled || !dw || !db || !workspace || B <= 0 || L <= 0 || (d & 3) || ap_lds_bytes(L, d) > 60 * 1024) return OP_EINVAL;
  float* part = (float*)workspace;              // [B][d] dw partials then [B] db partials
  hipLaunchKernelGGL(k_attnpool_bwd, dim3(B), dim3(AP_WAVES * 64), ap_lds_bytes(L, d), (hipStream_t)stream, x, attn, w, dpooled, part, part + (size_t)B * d, dx, L, d);
  hipLaunchKernelGGL(k_reduce_partials, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)part, dw, B, (size_t)d, 0);
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)(part + (size_t)B * d), db, B, (size_t)1, 0);
  return launch_status();
}
extern "C" size_t oneprot_attnpool_bwd_workspace(int B, int d) { return ((size_t)B * d + B) * sizeof(float); }

// --------------------------------------------------------------------------------------------------------
// casts / fills / column sums
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cast_f32_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    u32x2 p; p.x = pack2bf(v.x, v.y); p.y = pack2bf(v.z, v.w);
    reinterpret_cast<u32x2*>(dst)[i] = p;
  }
}
extern "C" int oneprot_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0 || (n & 3)) return OP_EINVAL;
  const size_t n4 = (size_t)n >> 2;
  size_t blocks = (n4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_cast_f32_bf16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n4);
  return launch_status();
}

// dst[c][r] = bf16(src[r][c]) : 64x64 tiles through LDS (weights only: a few MB per optimizer step)
__global__ void __launch_bounds__(256) k_transpose_cast(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int C, int64_t src_stride, int64_t dst_stride) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  src += blockIdx.z * src_stride; dst += blockIdx.z * dst_stride;      // matrix z of a batch (the same weight of every layer)
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? src[(size_t)(r0 + r) * C + c0 + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < C) dst[(size_t)(c0 + c) * R + r0 + r] = f2bf(tile[r][c]);
  }
}
extern "C" int oneprot_transpose_cast_f32_to_bf16(const float* src, void* dst, int R, int C, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_transpose_cast, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, C, (int64_t)0, (int64_t)0);
  return launch_status();
}
// `count` matrices src + z * src_stride -> dst + z * dst_stride (strides in elements) in one launch: the same weight of every encoder layer
extern "C" int oneprot_transpose_cast_f32_to_bf16_batched(const float* src, void* dst, int R, int C, int64_t src_stride, int64_t dst_stride, int count, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || count <= 0 || count > 65535 || src_stride < 0 || dst_stride < (int64_t)R * C) return OP_EINVAL;
  hipLaunchKernelGGL(k_transpose_cast, dim3((C + 63) / 64, (R + 63) / 64, count), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, C, src_stride, dst_stride);
  return launch_status();
}

// Column sums of a bf16 [M, N] matrix (bias gradients): partial[blockIdx.y][n] then reduce.
__global__ void __launch_bounds__(256) k_colsum_bf16(const bf16_t* __restrict__ a, float* __restrict__ partial, int M, int N, int rows_per_block) {
  const int c2 = blockIdx.x * 256 + threadIdx.x;      // pair of columns
  if (c2 * 2 >= N) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f;
  for (int r = r0; r < r1; ++r) {
    const unsigned p = reinterpret_cast<const unsigned*>(a + (size_t)r * N)[c2];
    s0 += bflo(p); s1 += bfhi(p);
  }
  partial[(size_t)blockIdx.y * N + 2 * c2] = s0;
  partial[(size_t)blockIdx.y * N + 2 * c2 + 1] = s1;
}
#define COLSUM_PARTS 256
extern "C" size_t oneprot_colsum_workspace(int N) { return (size_t)COLSUM_PARTS * N * sizeof(float); }
extern "C" int oneprot_colsum_bf16(const void* a, float* out, void* workspace, int64_t M, int N, int accumulate, void* stream) {
  if (!a || !out || !workspace || M <= 0 || N <= 0 || (N & 1)) return OP_EINVAL;
  const int rpb = (int)((M + COLSUM_PARTS - 1) / COLSUM_PARTS);
  const int parts = (int)((M + rpb - 1) / rpb);
  hipLaunchKernelGGL(k_colsum_bf16, dim3((N / 2 + 255) / 256, parts), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (float*)workspace, (int)M, N, rpb);
  hipLaunchKernelGGL(k_reduce_partials, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, parts, (size_t)N, accumulate);
  return launch_status();
}
