// Development probe (GPU): ONE STEP of k_attn_bwd_fused64's dataflow -- a wave's two key blocks (64 keys) against one block of 32 queries, head_dim 32 -- with
// every operand a register that is "re-read" each step: no global memory, no Q / dO rows in LDS, no tickets, no barrier.  What stays from the kernel:
//   per key block: S^T = K Q^T (2 MFMAs 32x32x16 + 1 bookkeeping k-step), dP^T = V dO^T (2 + 1), P = 2^S (16 v_exp_f32), dS = P dP (16 v_mul), two packs of 8
//                  v_cvt_pk each, dV^T += dO^T-frag x P (2 MFMAs), dK^T += Q^T-frag x dS (2 MFMAs), the dS tile through the wave's private LDS slab
//                  (4 ds_write_b64, 4 ds_read_b64_tr_b16: its transpose), dQ^T += K^T-frag x dS^T (2 MFMAs into ONE accumulator for both key blocks);
//   per step:      the dQ tile read-add-written into an fp32 LDS buffer (4 ds_read_b128, 16 v_add, 4 ds_write_b128) -- un-ticketed here.
// = 24 MFMAs, 32 exponentials, ~120 other vector instructions, 16 LDS writes and 12 LDS reads per step.
// Orders:  V0  key block 0 whole, then key block 1, then the dQ read-add-write (everything in dependence order)
//          V1  all four score / dP chains first, then the vector work of both blocks, then the dV / dK / dQ MFMAs (what a schedule that overlaps the pipes needs)
//          V2  V1 with the dQ half (transposed reads, dQ MFMAs, read-add-write) of step s issued in step s + 1 behind its chains: the kernel's deferral
// Run with one wave per SIMD (4 waves) and two (8 waves: the kernel's occupancy).  Prints SIMD cycles per step (two 32 x 32 tiles); the shipped kernel takes a wave
// 2 800-3 500 cycles per step at two waves per SIMD (profiles/r04_bwd64_stamps.txt), i.e. 1 400-1 750 SIMD cycles per step.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ab/attn_bwd_mix_probe.hip -o tools/ab/attn_bwd_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA(c, a, b) (c) = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x16 zero16() { f32x16 z; for (int i = 0; i < 16; ++i) z[i] = 0.f; return z; }
__device__ __forceinline__ unsigned cvtpk(float a, float b) { unsigned d; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ bf8_t pack(const f32x16& s, int sb) {
  u32x4 w = {cvtpk(s[8 * sb], s[8 * sb + 1]), cvtpk(s[8 * sb + 2], s[8 * sb + 3]), cvtpk(s[8 * sb + 4], s[8 * sb + 5]), cvtpk(s[8 * sb + 6], s[8 * sb + 7])};
  return __builtin_bit_cast(bf8_t, w);
}
__device__ __forceinline__ bf8_t rd_tr(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  s16x8 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  return __builtin_bit_cast(bf8_t, o);
}

template <int V, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* slab = smem + wave * 4096;                       // the wave's private dS slab: two 2 KB tiles (one per key block)
  float* dqbuf = reinterpret_cast<float*>(smem + 16 * 4096) + wave * (32 * 36);      // a 32 x 32 fp32 dQ block (36-float rows), private here
  bf8_t kf[2][2], vf[2][2], ktf[2][2], qf[2], dof[2], qtf[2], dotf[2], cst;
  {
    unsigned w[4];
    auto mk = [&](int o) -> bf8_t {
      for (int j = 0; j < 4; ++j) {
        const float a = (in[(lane * 8 + 2 * j + o) & 1023] - 0.75f) * 0.25f, b = (in[(lane * 8 + 2 * j + 1 + o) & 1023] - 0.75f) * 0.25f;
        w[j] = (__builtin_bit_cast(unsigned, a) >> 16) | (__builtin_bit_cast(unsigned, b) & 0xffff0000u);
      }
      u32x4 u = {w[0], w[1], w[2], w[3]};
      return __builtin_bit_cast(bf8_t, u);
    };
    int o = 0;
    for (int kb = 0; kb < 2; ++kb) for (int s = 0; s < 2; ++s) { kf[kb][s] = mk(o); vf[kb][s] = mk(o + 32); ktf[kb][s] = mk(o + 64); o += 96; }
    for (int s = 0; s < 2; ++s) { qf[s] = mk(o); dof[s] = mk(o + 32); qtf[s] = mk(o + 64); dotf[s] = mk(o + 96); o += 128; }
    cst = mk(o);
  }
  f32x16 dV[2] = {zero16(), zero16()}, dK[2] = {zero16(), zero16()};
  // lane addresses of the slab: the row layout written (lane owns 4 consecutive bf16 of 4 rows) and the transposed gather read back
  const unsigned wr_off = (unsigned)(lane & 31) * 64u + (unsigned)(lane >> 5) * 8u;
  const unsigned tr_off = (unsigned)(lane & 15) * 64u + (unsigned)(lane >> 4) * 16u;
  const unsigned dq_off = (unsigned)(lane & 31) * 36u + (unsigned)(lane >> 5) * 16u;
  for (int i = lane; i < 32 * 36; i += 64) dqbuf[i] = 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");

  auto chains = [&](int kb, f32x16& s, f32x16& dp) {
    s = zero16(); dp = zero16();
    MFMA(s, cst, qf[0]);                                          // bookkeeping k-step: bias[key] - lse[query]
    MFMA(s, kf[kb][0], qf[0]); MFMA(s, kf[kb][1], qf[1]);
    MFMA(dp, cst, dof[0]);                                        // bookkeeping k-step: -delta[query]
    MFMA(dp, vf[kb][0], dof[0]); MFMA(dp, vf[kb][1], dof[1]);
  };
  auto vector_work = [&](f32x16& s, const f32x16& dp, bf8_t (&pp)[2], bf8_t (&dsp)[2]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
    pp[0] = pack(s, 0); pp[1] = pack(s, 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = s[i] * dp[i];
    dsp[0] = pack(s, 0); dsp[1] = pack(s, 1);
  };
  auto slab_write = [&](int kb, const bf8_t (&dsp)[2]) {
    unsigned char* t = slab + kb * 2048;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const u32x4 w = __builtin_bit_cast(u32x4, dsp[sb]);
      *reinterpret_cast<u32x2*>(t + wr_off + sb * 1024) = (u32x2){w[0], w[1]};
      *reinterpret_cast<u32x2*>(t + wr_off + sb * 1024 + 32) = (u32x2){w[2], w[3]};
    }
  };
  auto dvdk = [&](int kb, const bf8_t (&pp)[2], const bf8_t (&dsp)[2]) {
    MFMA(dV[kb], dotf[0], pp[0]); MFMA(dV[kb], dotf[1], pp[1]);
    MFMA(dK[kb], qtf[0], dsp[0]); MFMA(dK[kb], qtf[1], dsp[1]);
  };
  auto dq_half = [&](f32x16& dq) {                                // transposed dS fragments of both key blocks, the dQ MFMAs, the read-add-write of the block
    dq = zero16();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const unsigned char* t = slab + kb * 2048;
      const bf8_t d0 = rd_tr(t + tr_off, t + tr_off + 8), d1 = rd_tr(t + tr_off + 1024, t + tr_off + 1032);
      MFMA(dq, ktf[kb][0], d0); MFMA(dq, ktf[kb][1], d1);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 o = *reinterpret_cast<f32x4*>(dqbuf + dq_off + g * 4 + (g >> 1) * 0);
      o[0] += dq[4 * g]; o[1] += dq[4 * g + 1]; o[2] += dq[4 * g + 2]; o[3] += dq[4 * g + 3];
      *reinterpret_cast<f32x4*>(dqbuf + dq_off + g * 4) = o;
    }
  };

#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(dof[0]), "+v"(dof[1]), "+v"(qtf[0]), "+v"(qtf[1]), "+v"(dotf[0]), "+v"(dotf[1]));      // Q / dO fragments "re-read" every step
    f32x16 s0, dp0, s1, dp1, dq;
    bf8_t pp0[2], ds0[2], pp1[2], ds1[2];
    if (V == 0) {
      chains(0, s0, dp0); vector_work(s0, dp0, pp0, ds0); slab_write(0, ds0); dvdk(0, pp0, ds0);
      chains(1, s1, dp1); vector_work(s1, dp1, pp1, ds1); slab_write(1, ds1); dvdk(1, pp1, ds1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dq_half(dq);
    } else if (V == 1) {
      chains(0, s0, dp0); chains(1, s1, dp1);
      vector_work(s0, dp0, pp0, ds0); slab_write(0, ds0);
      dvdk(0, pp0, ds0);
      vector_work(s1, dp1, pp1, ds1); slab_write(1, ds1);
      dvdk(1, pp1, ds1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dq_half(dq);
    } else {
      chains(0, s0, dp0); chains(1, s1, dp1);
      if (it > 0) dq_half(dq);                                    // the PREVIOUS step's dQ half, behind this step's chains (its slabs are rewritten below)
      vector_work(s0, dp0, pp0, ds0);
      dvdk(0, pp0, ds0);
      vector_work(s1, dp1, pp1, ds1);
      dvdk(1, pp1, ds1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slab_write(0, ds0); slab_write(1, ds1);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
  float sum = dqbuf[lane];
  for (int i = 0; i < 16; ++i) sum += dV[0][i] + dV[1][i] + dK[0][i] + dK[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (blockIdx.x == 0 && lane == 0) { cyc[wave] = t1 - t0; cyc[16 + wave] = r1 - r0; }
}

template <int V, int WAVES>
static void run(const char* name, const float* in, float* out, unsigned long long* cyc, int iters) {
  const int lds = 16 * 4096 + 16 * 32 * 36 * 4;
  hipFuncSetAttribute((const void*)k_probe<V, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<V, WAVES>), dim3(256), dim3(WAVES * 64), lds, 0, in, out, cyc, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<V, WAVES>), dim3(256), dim3(WAVES * 64), lds, 0, in, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[32]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[16] * 10e-9) / 1e9;
  const double steps_per_simd = (double)(WAVES / 4) * iters;
  printf("%-74s waves/SIMD %d | wave 0 %7.1f, last wave %7.1f cyc/step | wall %.3f ms = %7.1f SIMD-cyc/step at %.2f GHz\n", name, WAVES / 4,
         (double)h[0] / iters, (double)h[WAVES - 1] / iters, ms, ms * 1e-3 * ghz * 1e9 / steps_per_simd, ghz);
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 1024 * sizeof(float)); hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 32 * sizeof(unsigned long long));
  float h[1024];
  unsigned s = 12345u;
  for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f + 0.25f; }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int it = 10000;
  run<0, 4>("V0 key block 0, key block 1, dQ half: dependence order", in, out, cyc, it);
  run<0, 8>("V0 key block 0, key block 1, dQ half: dependence order", in, out, cyc, it);
  run<1, 4>("V1 all four chains first, then vector work / dV dK per block, dQ half", in, out, cyc, it);
  run<1, 8>("V1 all four chains first, then vector work / dV dK per block, dQ half", in, out, cyc, it);
  run<2, 4>("V2 = V1 with the dQ half of step s issued in step s + 1 behind its chains", in, out, cyc, it);
  run<2, 8>("V2 = V1 with the dQ half of step s issued in step s + 1 behind its chains", in, out, cyc, it);
  printf("per step: 24 MFMA 32x32x16 (bare: 24 x 32 = 768 pipe cycles at one MFMA per 32; 39 measured per MFMA beside other work), 32 v_exp_f32 (10.6 each at four waves per\n"
         "SIMD = 340), ~120 plain vector instructions (~4 each = 480), 16 LDS writes + 12 reads.  Shipped kernel: 2 800-3 500 wave-cycles per step at two waves per SIMD\n"
         "= 1 400-1 750 SIMD-cycles per step (profiles/r04_bwd64_stamps.txt).\n");
  return 0;
}
