// Small fp32 kernels around the feature vectors and the optimiser:
//   GELU (proj head), L2-normalise (+logit scale), softmax cross-entropy of the contrastive logits (fwd+bwd fused),
//   L1 feature penalty, global grad-norm (two-stage deterministic reduction), clip coefficient, fused Adam.
// All HBM-bound or latency-bound; reductions are deterministic (no float atomics).
#include "common.h"
#include "sched_ws.h"
#include "../../include/oneprot_hip.h"
#include <float.h>

#define RED_BLOCKS 1024

// block-wide sum for 256 threads; result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* s4) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return s4[0] + s4[1] + s4[2] + s4[3];
}

// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gelu(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = gelu_erf(x[i]);
}
__global__ void __launch_bounds__(256) k_gelu_bwd(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = dy[i] * gelu_erf_grad(x[i]);
}
static inline unsigned ew_grid(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b > 4096 ? 4096 : b); }
extern "C" int oneprot_gelu_f32(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_gelu, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n);
  return launch_status();
}
extern "C" int oneprot_gelu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  if (!x || !dy || !dx || n <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_gelu_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (size_t)n);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// y = scale * x / max(||x||, 1e-12)   (ref base_encoder.py:11-12, 32-33); one wave per row
__global__ void __launch_bounds__(256) k_l2norm_fwd(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm, int R, int D, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float s = 0.f;
  for (int j = lane; j < D; j += 64) { const float v = x[(size_t)row * D + j]; s += v * v; }
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
  for (int j = lane; j < D; j += 64) y[(size_t)row * D + j] = x[(size_t)row * D + j] * inv * scale;
}
// g = dy + l1_coef * sign(y);  dx = scale*inv*(g - xhat * (xhat . g)),  xhat = y / scale
__global__ void __launch_bounds__(256) k_l2norm_bwd(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                    int R, int D, float scale, float l1_coef) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float rs = 1.0f / scale;
  float dot = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float yv = y[(size_t)row * D + j];
    const float g = dy[(size_t)row * D + j] + l1_coef * (yv > 0.f ? 1.f : (yv < 0.f ? -1.f : 0.f));
    dot += yv * rs * g;
  }
  dot = wave_sum(dot);
  const float k = scale * inv_norm[row];
  for (int j = lane; j < D; j += 64) {
    const float yv = y[(size_t)row * D + j];
    const float g = dy[(size_t)row * D + j] + l1_coef * (yv > 0.f ? 1.f : (yv < 0.f ? -1.f : 0.f));
    dx[(size_t)row * D + j] = k * (g - yv * rs * dot);
  }
}
extern "C" int oneprot_l2norm_fwd(const float* x, float* y, float* inv_norm, int R, int D, float scale, void* stream) {
  if (!x || !y || R <= 0 || D <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_l2norm_fwd, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, R, D, scale);
  return launch_status();
}
extern "C" int oneprot_l2norm_bwd(const float* y, const float* dy, const float* inv_norm, float* dx, int R, int D, float scale, float l1_coef, void* stream) {
  if (!y || !dy || !inv_norm || !dx || R <= 0 || D <= 0 || scale == 0.f) return OP_EINVAL;
  hipLaunchKernelGGL(k_l2norm_bwd, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, dy, inv_norm, dx, R, D, scale, l1_coef);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// softmax cross-entropy, label[r] = r + label_offset (ref loss.py:72-83, 108-112).  One block per row.
// row_loss[r] = (lse - logit[label]) * row_weight ; logits <- (softmax - onehot) * row_weight
__global__ void __launch_bounds__(256) k_ce_fwd_bwd(float* __restrict__ logits, float* __restrict__ row_loss, int C, int label_offset, float row_weight) {
  __shared__ float s4[4];
  const int r = blockIdx.x;
  float* row = logits + (size_t)r * C;
  float mx = -FLT_MAX;
  for (int j = threadIdx.x; j < C; j += 256) mx = fmaxf(mx, row[j]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
  __syncthreads();
  float se = 0.f;
  for (int j = threadIdx.x; j < C; j += 256) se += __expf(row[j] - mx);
  se = block_sum_256(se, s4);
  const float lse = mx + logf(se);
  const int label = r + label_offset;
  if (threadIdx.x == 0) row_loss[r] = (lse - row[label]) * row_weight;
  __syncthreads();
  const float inv = 1.0f / se;
  for (int j = threadIdx.x; j < C; j += 256) {
    const float p = __expf(row[j] - mx) * inv;
    row[j] = (p - (j == label ? 1.f : 0.f)) * row_weight;
  }
}
// out[0] += sum_i v[i]  (single block, fixed order)
__global__ void __launch_bounds__(256) k_final_sum(const float* __restrict__ v, int n, float* __restrict__ out, float coef) {
  __shared__ float s4[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = block_sum_256(s, s4);
  if (threadIdx.x == 0) out[0] += coef * s;
}
extern "C" int oneprot_ce_fwd_bwd(float* logits, float* loss_sum, float* row_loss_ws, int R, int C, int label_offset, float row_weight, void* stream) {
  if (!logits || !loss_sum || !row_loss_ws || R <= 0 || C <= 0 || label_offset < 0 || label_offset + R > C) return OP_EINVAL;
  hipLaunchKernelGGL(k_ce_fwd_bwd, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, row_loss_ws, C, label_offset, row_weight);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)row_loss_ws, R, loss_sum, 1.0f);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// deterministic reductions: MODE 0 sum x^2, MODE 1 sum |x|, MODE 2 sum x*y
template <int MODE>
__global__ void __launch_bounds__(256) k_reduce_stage1(const float* __restrict__ x, size_t n, float* __restrict__ partial, const float* __restrict__ y = nullptr) {
  __shared__ float s4[4];
  float s = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    if (MODE == 0) s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    else if (MODE == 1) s += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
    else { const float4 u = reinterpret_cast<const float4*>(y)[i]; s += (v.x * u.x + v.y * u.y) + (v.z * u.z + v.w * u.w); }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float t = x[(n4 << 2) + threadIdx.x];
    s += MODE == 0 ? t * t : MODE == 1 ? fabsf(t) : t * y[(n4 << 2) + threadIdx.x];
  }
  s = block_sum_256(s, s4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
extern "C" size_t oneprot_sumsq_workspace(void) { return RED_BLOCKS * sizeof(float); }
static int reduce_launch(int mode, const float* x, int64_t n, float* out, void* ws, float coef, hipStream_t s, const float* y = nullptr) {
  if (!x || !out || !ws || n <= 0 || ((uintptr_t)x & 15)) return OP_EINVAL;
  if (mode == 2 && (!y || ((uintptr_t)y & 15))) return OP_EINVAL;
  size_t blocks = ((size_t)n / 4 + 255) / 256; if (blocks > RED_BLOCKS) blocks = RED_BLOCKS; if (blocks == 0) blocks = 1;
  if (mode == 0) hipLaunchKernelGGL(k_reduce_stage1<0>, dim3((unsigned)blocks), dim3(256), 0, s, x, (size_t)n, (float*)ws);
  else if (mode == 1) hipLaunchKernelGGL(k_reduce_stage1<1>, dim3((unsigned)blocks), dim3(256), 0, s, x, (size_t)n, (float*)ws);
  else hipLaunchKernelGGL(k_reduce_stage1<2>, dim3((unsigned)blocks), dim3(256), 0, s, x, (size_t)n, (float*)ws, y);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, s, (const float*)ws, (int)blocks, out, coef);
  return launch_status();
}
extern "C" int oneprot_sumsq(const float* x, int64_t n, float* sumsq, void* workspace, void* stream) { return reduce_launch(0, x, n, sumsq, workspace, 1.0f, (hipStream_t)stream); }
extern "C" int oneprot_abs_sum(const float* x, float* out_sum, void* workspace, int64_t n, float coef, void* stream) { return reduce_launch(1, x, n, out_sum, workspace, coef, (hipStream_t)stream); }
extern "C" int oneprot_dot_f32(const float* x, const float* y, float* out_sum, void* workspace, int64_t n, float coef, void* stream) { return reduce_launch(2, x, n, out_sum, workspace, coef, (hipStream_t)stream, y); }

// `poison` (optional): the sticky LN_ERR word of a sched workspace (sched_ws.h).  Non-zero = some launch of this step wrote NaN rows because a bounded wait
// ran out: the gradient norm and the clip coefficient of THIS step become NaN as well (a device read: no host synchronisation), so that the failure is on the
// loss, on the logged gradient norm and on every updated parameter in the step it happened in, whatever the tower's NaN rows fed into.
__global__ void k_clip_coef(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef, float* __restrict__ norm_out, const unsigned* __restrict__ poison) {
  float nrm = sqrtf(sumsq[0]);
  if (poison && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) nrm = __builtin_nanf("");
  if (norm_out) norm_out[0] = nrm;
  coef[0] = nrm != nrm ? nrm : fminf(1.0f, max_norm / (nrm + 1e-6f));      // torch.nn.utils.clip_grad_norm_ semantics (fminf would drop a NaN)
}
extern "C" int oneprot_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, const void* sched_ws, void* stream) {
  if (!sumsq || !coef) return OP_EINVAL;
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef, norm_out, sched_ws ? (const unsigned*)sched_ws + SW_LN_ERR : nullptr);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// torch.optim.Adam (no amsgrad, L2-style weight_decay added to the gradient), 28 B/param of HBM traffic
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n4, float lr,
                                              float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, const float* __restrict__ grad_scale) {
  const float gs = grad_scale ? grad_scale[0] : 1.0f;
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w}, me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = ge[e] * gs;
      if (wd != 0.f) gg += wd * pe[e];
      me[e] = b1 * me[e] + (1.f - b1) * gg;
      ve[e] = b2 * ve[e] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(ve[e]) / bc2_sqrt + eps;
      pe[e] -= step_size * me[e] / denom;
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
  }
}
extern "C" int oneprot_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 int step, const float* grad_scale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || (n & 3) || step < 1) return OP_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const size_t n4 = (size_t)n >> 2;
  size_t blocks = (n4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2), grad_scale);
  return launch_status();
}

// dx (+)= coef * upstream[0] * sign(x)   (gradient of coef * sum|x|; upstream = device scalar dL/dloss, NULL => 1)
__global__ void __launch_bounds__(256) k_l1_bwd(const float* __restrict__ x, float* __restrict__ dx, size_t n, float coef, const float* __restrict__ upstream, int accumulate) {
  const float c = coef * (upstream ? upstream[0] : 1.0f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i];
    const float g = c * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
    dx[i] = accumulate ? dx[i] + g : g;
  }
}
extern "C" int oneprot_l1_bwd(const float* x, float* dx, int64_t n, float coef, const float* upstream, int accumulate, void* stream) {
  if (!x || !dx || n <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_l1_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dx, (size_t)n, coef, upstream, accumulate);
  return launch_status();
}
// x *= s[0]
__global__ void __launch_bounds__(256) k_scale_dev(float* __restrict__ x, size_t n, const float* __restrict__ s) {
  const float c = s[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= c;
}
extern "C" int oneprot_scale_by_device_scalar(float* x, int64_t n, const float* s, void* stream) {
  if (!x || !s || n <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_scale_dev, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, s);
  return launch_status();
}
// additive key-padding mask: bias[i] = ids[i] == pad ? finfo(float32).min : 0   (hf create_bidirectional_mask)
__global__ void __launch_bounds__(256) k_key_bias(const long long* __restrict__ ids, float* __restrict__ bias, size_t n, int pad_id) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bias[i] = ids[i] == pad_id ? -FLT_MAX : 0.f;
}
extern "C" int oneprot_key_padding_bias(const int64_t* ids, float* bias, int64_t n, int pad_id, void* stream) {
  if (!ids || !bias || n <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_key_bias, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, bias, (size_t)n, pad_id);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Dropout of a bf16 operand (peft 0.5.0 lora.Linear: the adapter branch sees dropout(x); ref sequence_encoder.py:61-74, lora_dropout).  The mask is
// never stored: element e keeps iff the 16-bit slice e & 7 of Philox4x32-10(counter = (e >> 3, stream), key = seed) is >= thr = round(p * 65536),
// so the backward kernel regenerates the forward's mask from (seed, stream).  Kept values are scaled by 65536 / (65536 - thr), the inverse of
// the keep probability actually used.  torch's own Philox stream is not reproduced (a different generator of the same distribution).
// MODE 0: y = dropout(x);  MODE 1: y += dropout'(x) = mask * scale * x  (y, x bf16; the adapter branch's input gradient added to the direct one)
template <int MODE>
__global__ void __launch_bounds__(256) k_dropout_bf16(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t n8, unsigned thr, float scale, unsigned long long seed,
                                                     unsigned long long stream_id) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    unsigned rnd[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32), (unsigned)seed, (unsigned)(seed >> 32), rnd);
    const u32x4 v = x[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (MODE == 1) acc = y[i];
    const unsigned a[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = ((rnd[j] & 0xffffu) >= thr) ? bflo(w[j]) * scale : 0.f;
      float hi = ((rnd[j] >> 16) >= thr) ? bfhi(w[j]) * scale : 0.f;
      if (MODE == 1) { lo += bflo(a[j]); hi += bfhi(a[j]); }
      o[j] = pack2bf(lo, hi);
    }
    const u32x4 r = {o[0], o[1], o[2], o[3]};
    y[i] = r;
  }
}
// the same into an fp32 gradient: dx[e] += mask * scale * dy[e]  (post-LN towers keep the layer-input gradient in fp32)
__global__ void __launch_bounds__(256) k_dropout_bwd_add_f32(const u32x4* __restrict__ dy, float4* __restrict__ dx, size_t n8, unsigned thr, float scale,
                                                             unsigned long long seed, unsigned long long stream_id) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    unsigned rnd[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32), (unsigned)seed, (unsigned)(seed >> 32), rnd);
    const u32x4 v = dy[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float4 a = dx[2 * i], b = dx[2 * i + 1];
    float* o[8] = {&a.x, &a.y, &a.z, &a.w, &b.x, &b.y, &b.z, &b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((rnd[j] & 0xffffu) >= thr) *o[2 * j] += bflo(w[j]) * scale;
      if ((rnd[j] >> 16) >= thr) *o[2 * j + 1] += bfhi(w[j]) * scale;
    }
    dx[2 * i] = a; dx[2 * i + 1] = b;
  }
}
// y = dropout(x) [+ resid] on fp32 (hidden-state dropout of the BERT tower: hf modeling_bert.py BertEmbeddings / BertSelfOutput / BertOutput -- the two
// dense outputs are added to the residual stream right after their dropout), in place allowed
__global__ void __launch_bounds__(256) k_dropout_f32(const float4* __restrict__ x, const float4* __restrict__ resid, float4* __restrict__ y, size_t n8, unsigned thr, float scale,
                                                     unsigned long long seed, unsigned long long stream_id) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    unsigned rnd[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32), (unsigned)seed, (unsigned)(seed >> 32), rnd);
    float4 a = x[2 * i], b = x[2 * i + 1];
    float* o[8] = {&a.x, &a.y, &a.z, &a.w, &b.x, &b.y, &b.z, &b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *o[2 * j] = ((rnd[j] & 0xffffu) >= thr) ? *o[2 * j] * scale : 0.f;
      *o[2 * j + 1] = ((rnd[j] >> 16) >= thr) ? *o[2 * j + 1] * scale : 0.f;
    }
    if (resid) {
      const float4 ra = resid[2 * i], rb = resid[2 * i + 1];
      a.x += ra.x; a.y += ra.y; a.z += ra.z; a.w += ra.w; b.x += rb.x; b.y += rb.y; b.z += rb.z; b.w += rb.w;
    }
    y[2 * i] = a; y[2 * i + 1] = b;
  }
}
static int launch_dropout(int mode, const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream, const void* resid = nullptr) {
  if (!x || !y || n <= 0 || (n & 7) || !(p >= 0.f) || !(p < 1.f)) return OP_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)resid) & 15) return OP_EINVAL;
  const unsigned thr = (unsigned)(p * 65536.f + 0.5f);
  if (thr >= 65536u) return OP_EINVAL;
  const float scale = 65536.f / (float)(65536u - thr);
  const size_t n8 = (size_t)n >> 3;
  if (mode == 3) hipLaunchKernelGGL(k_dropout_f32, dim3(ew_grid(n8)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)resid, (float4*)y, n8, thr, scale, (unsigned long long)seed, (unsigned long long)stream_id);
  else if (mode == 2) hipLaunchKernelGGL(k_dropout_bwd_add_f32, dim3(ew_grid(n8)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (float4*)y, n8, thr, scale, (unsigned long long)seed, (unsigned long long)stream_id);
  else if (mode == 0) hipLaunchKernelGGL(k_dropout_bf16<0>, dim3(ew_grid(n8)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (u32x4*)y, n8, thr, scale, (unsigned long long)seed, (unsigned long long)stream_id);
  else hipLaunchKernelGGL(k_dropout_bf16<1>, dim3(ew_grid(n8)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (u32x4*)y, n8, thr, scale, (unsigned long long)seed, (unsigned long long)stream_id);
  return launch_status();
}
extern "C" int oneprot_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  return launch_dropout(0, x, y, n, p, seed, stream_id, stream);
}
extern "C" int oneprot_dropout_bwd_add_bf16(const void* dy, void* dx, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  return launch_dropout(1, dy, dx, n, p, seed, stream_id, stream);
}
extern "C" int oneprot_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  return launch_dropout(3, x, y, n, p, seed, stream_id, stream);
}
extern "C" int oneprot_dropout_add_f32(const float* x, const float* resid, float* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  if (!resid) return OP_EINVAL;
  return launch_dropout(3, x, y, n, p, seed, stream_id, stream, resid);
}
extern "C" int oneprot_dropout_bwd_add_f32(const void* dy, float* dx, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  return launch_dropout(2, dy, dx, n, p, seed, stream_id, stream);
}

// s = resid + dropout(x); LayerNorm(s) in ONE pass over the rows (hf modeling_bert.py BertSelfOutput / BertOutput: dense -> dropout -> LayerNorm(. + input)):
// the separate kernels moved the sum through HBM twice (written by the dropout + add, read again by the LayerNorm).  Same mask as oneprot_dropout_f32 /
// oneprot_dropout_add_f32 (a lane owns the 8 consecutive elements of one Philox call), same LayerNorm arithmetic as k_layernorm_fwd with eight
// elements per lane and step instead of four.  s_out (the pre-LayerNorm sum the backward needs) is optional: a frozen tower does not write it.
template <int NV>
__global__ void __launch_bounds__(256) k_dropout_add_ln_fwd(const float* __restrict__ x, const float* resid, float* s_out,      // (y_f32 / s_out may alias resid: a lane reads what it writes, first)
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y_bf16,
                                                            float* y_f32, float* __restrict__ mean_out, float* __restrict__ rstd_out, int T, int d, float eps,
                                                            unsigned thr, float scale, unsigned long long seed, unsigned long long stream_id) {
  const int lane = threadIdx.x & 63;
  const int nv8 = d >> 3;
  const float inv_d = 1.0f / (float)d;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < T; row += gridDim.x * 4) {
    float v[NV][8];
    float sum = 0.f;
    const size_t row8 = (size_t)row * nv8;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv8) {
        const size_t e8 = row8 + c;
        unsigned rnd[4];
        philox4x32_10((unsigned)e8, (unsigned)(e8 >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32), (unsigned)seed, (unsigned)(seed >> 32), rnd);
        const float4 a = reinterpret_cast<const float4*>(x)[2 * e8], b = reinterpret_cast<const float4*>(x)[2 * e8 + 1];
        const float4 ra = reinterpret_cast<const float4*>(resid)[2 * e8], rb = reinterpret_cast<const float4*>(resid)[2 * e8 + 1];
        const float xin[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const float rin[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[i][2 * j] = (((rnd[j] & 0xffffu) >= thr) ? xin[2 * j] * scale : 0.f) + rin[2 * j];
          v[i][2 * j + 1] = (((rnd[j] >> 16) >= thr) ? xin[2 * j + 1] * scale : 0.f) + rin[2 * j + 1];
        }
        if (s_out) {
          reinterpret_cast<float4*>(s_out)[2 * e8] = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
          reinterpret_cast<float4*>(s_out)[2 * e8 + 1] = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
        }
        sum += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
      }
    }
    const float mean = wave_sum(sum) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nv8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = v[i][j] - mean; q += t * t; }
      }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv8) {
        const float4 g0 = reinterpret_cast<const float4*>(gamma)[2 * c], g1 = reinterpret_cast<const float4*>(gamma)[2 * c + 1];
        const float4 b0 = reinterpret_cast<const float4*>(beta)[2 * c], b1 = reinterpret_cast<const float4*>(beta)[2 * c + 1];
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
        const size_t e8 = row8 + c;
        if (y_f32) {
          reinterpret_cast<float4*>(y_f32)[2 * e8] = make_float4(o[0], o[1], o[2], o[3]);
          reinterpret_cast<float4*>(y_f32)[2 * e8 + 1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (y_bf16) {
          u32x4 w; w.x = pack2bf(o[0], o[1]); w.y = pack2bf(o[2], o[3]); w.z = pack2bf(o[4], o[5]); w.w = pack2bf(o[6], o[7]);
          reinterpret_cast<u32x4*>(y_bf16)[e8] = w;
        }
      }
    }
  }
}
extern "C" int oneprot_dropout_add_layernorm_fwd(const float* x, const float* resid, float* s_out, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                                 float* mean, float* rstd, int64_t T, int d, float eps, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  if (!x || !resid || !gamma || !beta || (!y_bf16 && !y_f32) || T <= 0 || d <= 0 || (d & 7) || d > 4 * 512 || !(p >= 0.f) || !(p < 1.f)) return OP_EINVAL;
  if (((uintptr_t)x | (uintptr_t)resid | (uintptr_t)s_out | (uintptr_t)y_bf16 | (uintptr_t)y_f32 | (uintptr_t)gamma | (uintptr_t)beta) & 15) return OP_EINVAL;
  const unsigned thr = (unsigned)(p * 65536.f + 0.5f);
  if (thr >= 65536u) return OP_EINVAL;
  const float scale = 65536.f / (float)(65536u - thr);
  const int nv = (d / 8 + 63) / 64;
  const int blocks = (int)((T + 3) / 4 < 4096 ? (T + 3) / 4 : 4096);
#define DLN(N) hipLaunchKernelGGL((k_dropout_add_ln_fwd<N>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, resid, s_out, gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd, (int)T, d, eps, thr, scale, (unsigned long long)seed, (unsigned long long)stream_id)
  if (nv <= 1) DLN(1); else if (nv == 2) DLN(2); else if (nv == 3) DLN(3); else DLN(4);
#undef DLN
  return launch_status();
}

// SigLIP block (ref loss.py:229-255): logits [B,B] (already scale*m@s^T) -> loss_sum += -sum logsigmoid(label*(logit+bias)) * inv_b,
// logits <- dloss/dlogit = -label * sigmoid(-label*(logit+bias)) * inv_b;  label = +1 on the diagonal (unless negative_only), else -1.
__global__ void __launch_bounds__(256) k_siglip_fwd_bwd(float* __restrict__ logits, float* __restrict__ row_loss, int B, float bias, const float* __restrict__ bias_dev,
                                                        int negative_only, float inv_b) {
  __shared__ float s4[4];
  const int r = blockIdx.x;
  if (bias_dev) bias = bias_dev[0];      // a learnable bias stays on the device (ref loss.py:243-245 adds the tensor into the logits)
  float acc = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float label = (!negative_only && j == r) ? 1.f : -1.f;
    const float z = label * (logits[(size_t)r * B + j] + bias);
    // logsigmoid(z) = min(z,0) - log1p(exp(-|z|))
    acc -= fminf(z, 0.f) - log1pf(__expf(-fabsf(z)));
    const float sig_neg = 1.0f / (1.0f + __expf(z));           // sigmoid(-z)
    logits[(size_t)r * B + j] = -label * sig_neg * inv_b;
  }
  acc = block_sum_256(acc, s4);
  if (threadIdx.x == 0) row_loss[r] = acc * inv_b;
}
extern "C" int oneprot_siglip_fwd_bwd(float* logits, float* loss_sum, float* row_loss_ws, int B, float logit_bias, int negative_only, void* stream) {
  if (!logits || !loss_sum || !row_loss_ws || B <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_siglip_fwd_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, row_loss_ws, B, logit_bias, (const float*)nullptr, negative_only, 1.0f / (float)B);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)row_loss_ws, B, loss_sum, 1.0f);
  return launch_status();
}
extern "C" int oneprot_siglip_fwd_bwd_dev(float* logits, float* loss_sum, float* row_loss_ws, int B, const float* logit_bias_dev, int negative_only, void* stream) {
  if (!logits || !loss_sum || !row_loss_ws || B <= 0) return OP_EINVAL;
  hipLaunchKernelGGL(k_siglip_fwd_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, row_loss_ws, B, 0.f, logit_bias_dev, negative_only, 1.0f / (float)B);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)row_loss_ws, B, loss_sum, 1.0f);
  return launch_status();
}

// Retrieval ranks (ref retrieval_metric.py:83-102): for logits = S @ M^T [N,N], the rank of the matching pair is the number of entries that
// beat the diagonal -- counted directly (O(N^2)) instead of the reference's full argsort (O(N^2 log N) on the CPU).
//   rank_row[i] = #{ j : logits[i][j] > logits[i][i] }   (sequence -> modality)      rank_col[i] = #{ j : logits[j][i] > logits[i][i] }
__global__ void __launch_bounds__(256) k_diag_rank(const float* __restrict__ logits, int* __restrict__ rank_row, int* __restrict__ rank_col, int N) {
  __shared__ float s4[4];
  const int i = blockIdx.x;
  const float dg = logits[(size_t)i * N + i];
  float cr = 0.f, cc = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    cr += logits[(size_t)i * N + j] > dg ? 1.f : 0.f;
    cc += logits[(size_t)j * N + i] > dg ? 1.f : 0.f;
  }
  cr = block_sum_256(cr, s4);
  __syncthreads();
  cc = block_sum_256(cc, s4);
  if (threadIdx.x == 0) { rank_row[i] = (int)cr; rank_col[i] = (int)cc; }
}
extern "C" int oneprot_diag_rank(const float* logits, int* rank_row, int* rank_col, int N, void* stream) {
  if (!logits || !rank_row || !rank_col || N <= 0 || N >= (1 << 24)) return OP_EINVAL;
  hipLaunchKernelGGL(k_diag_rank, dim3(N), dim3(256), 0, (hipStream_t)stream, logits, rank_row, rank_col, N);
  return launch_status();
}

extern "C" int oneprot_abi_version(void) { return 7; }      // 7: sched workspace (oneprot_sched_workspace_bytes / _init, oneprot_alloc_uncached / _free_uncached, oneprot_dynamic_tiles), oneprot_gemm_bf16_nt_resid_ln8 / _error take it, oneprot_clip_coef reads its error flag; 6: oneprot_dropout_add_layernorm_fwd, oneprot_gemm_bf16_nt_resid_ln8 (+ _eligible, _error); 5: gelu' travels as one-byte codes (out1 of ONEPROT_EPI_BIAS_GELU / aux of ONEPROT_EPI_GELU_BWD are u8 tensors), oneprot_gemm_ln_form, oneprot_dropout_add_f32; 2: oneprot_gemm_bf16_tn takes workspace_bytes; 3: fused GEMM + LayerNorm entry points, oneprot_dot_f32; 4: oneprot_siglip_fwd_bwd_dev, oneprot_attn_force_fwd_path, oneprot_transpose_cast_f32_to_bf16_batched, oneprot_dropout_bf16 / _bwd_add_bf16 / _bwd_add_f32
