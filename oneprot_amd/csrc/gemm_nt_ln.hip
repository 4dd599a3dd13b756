// Full-row NT GEMM with the residual add AND the following LayerNorm in its epilogue.  gfx950 only.
//
//   x_out[M, 640] (fp32) = A[M, K] (bf16) * W[640, K]^T + bias + resid        (hf modeling_esm.py:399-409 out-projection, :455-463 FFN output)
//   h_out[M, 640] (bf16) = LayerNorm(x_out; gamma, beta, eps),  mean / rstd [M]  (hf modeling_esm.py:429,518: the NEXT sub-block's pre-LayerNorm)
//
// Why: after round 2 the fp32 residual stream was written by the N = 640 GEMMs (an HBM-bound epilogue) and read straight back by a separate
// LayerNorm launch -- 9.8 % of the step spent re-reading what the producing kernel still had in registers.  A work-group here owns WHOLE rows
// (128 x 640 tile), so the row statistics are formed from the accumulators and the normalised bf16 operand of the next GEMM leaves with x.
//
// Structure = the 8-phase form of gemm_nt8.hip (two groups of four waves half a phase apart, LDS-DMA stream continuous across the work-group's
// tiles, counted vmcnt, DIRECT accumulator layout), re-cut for a 640-wide tile that leaves no room for two 80 KB weight slices:
//   * waves 2 (M) x 4 (N), wave tile 64 x 160 (160 accumulator registers);
//   * activations are staged per 64-deep K-tile (128 rows x 128 B, whole lines, two buffers); the weight streams through a ring of SIX units of
//     20 KB = one 32-deep K-slab x one column half, pre-packed on the host side (oneprot_gemm_ln_pack_weight) in exactly the order and bank
//     swizzle the kernel consumes, so a unit is 20 contiguous KB of memory and one phase reads exactly one unit;
//   * phase = (K-slab, column half): 5 (+4 activation) fragment reads, 20 MFMAs; the unit five phases ahead is requested while the current one is
//     read (4 weight units + 1 activation tile in flight behind one counted vmcnt per phase).
// Epilogue (both wave groups aligned): pass 1 walks the 16-column tiles with bias / residual loads running two tiles ahead of the stores
// (gemm_epi8.h), moves each accumulator quad into the row layout (4 neighbouring lanes = 64 contiguous bytes of a row), adds the residual, stores
// x and keeps it in the registers; row statistics: two-pass inside the wave (160 columns), Chan's combination of the four column waves through 4 KB
// of LDS -- the same arithmetic order class as the stand-alone kernel (mean, then centred squares); pass 2 normalises from the registers.
// N = 640 only (d = 640 encoders), M % 128 == 0, K % 64 == 0; everything else keeps the GEMM + LayerNorm pair.
#include "gemm_epi8.h"
#ifndef GLN_X_NT
#define GLN_X_NT 0      // non-temporal stores of the fp32 residual stream (A/B builds: measured no better, NOTEBOOK 6e)
#endif

namespace gln {
using g8::gload16; using g8::lane_perm; using g8::as_f; using g8::to_rows_addr; using g8::run_groups;

constexpr int BM = 128, BN = 640, MT = 4, NT = 10, NH = 5;
constexpr int A_UNIT = BM * 128, W_UNIT = 320 * 64;       // 16 KB, 20 KB
constexpr int NA = 2, NW = 6;
constexpr int OFF_W = NA * A_UNIT, OFF_S = OFF_W + NW * W_UNIT, LDS = OFF_S + 4096;      // 156 KB
constexpr int A_IPW = 2;                                   // LDS-DMA pieces per wave per activation tile (16 pieces)
// weight unit: 20 pieces; waves 0-3 issue 3, waves 4-7 issue 2
static_assert(LDS <= 160 * 1024, "LDS");

struct LnArgs {
  const bf16_t* A; const unsigned char* Wp;
  int M, K, lda;
  const float* bias; const float* resid; float* x_out;
  const float* gamma; const float* beta; float eps;
  bf16_t* h_out; float* mean; float* rstd;
  int tiles;
  int late_ticks;
};

__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void glds16(unsigned voff, const unsigned char* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// 16-byte load, wave-uniform 64-bit base in SGPRs + 32-bit per-lane offset + immediate: no 64-bit address VGPRs (hipcc hoists those out of the
// tile loop and spills them; a spill reload is a vector-memory load, whose wait would queue behind the epilogue's stores)
template <int IMM> __device__ __forceinline__ void gload16s(u32x4& d, unsigned voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}

// Host-side companion: W bf16 [640, K] row-major -> units [K/32][2 halves][320 rows][4 chunks of 16 B], row hr of half nh = weight row
// (hr / 80) * 160 + nh * 80 + hr % 80 (the 80 rows wave column block hr/80 reads in that half), chunk c stored at position c ^ ((hr >> 1) & 3).
__global__ void __launch_bounds__(256) k_pack_w(const bf16_t* __restrict__ W, int K, u32x4* __restrict__ Wp) {
  const int units = K / 32 * 2;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < units * 320 * 4; idx += gridDim.x * 256) {
    const int pos = idx & 3, hr = (idx >> 2) % 320, u = idx / (320 * 4);
    const int k32 = u >> 1, nh = u & 1;
    const int c = pos ^ ((hr >> 1) & 3);
    const int n = (hr / 80) * 160 + nh * 80 + hr % 80;
    Wp[idx] = *reinterpret_cast<const u32x4*>(W + (size_t)n * K + k32 * 32 + c * 8);
  }
}

__global__ void __launch_bounds__(512, 2) k_gemm_ln(const LnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 2, wc = wave & 3;                  // wave tile: rows 64 wm .. +63, columns 160 wc .. +159
  const int nk = p.K >> 6;                                  // 64-deep activation K-tiles per output tile
  const int PH = 4 * nk;                                    // phases (= weight units) per output tile
  const int G = gridDim.x;
  const int Q = p.tiles > (int)blockIdx.x ? (p.tiles - (int)blockIdx.x + G - 1) / G : 0;
  if (Q == 0) return;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);

  // ---- LDS-DMA source offsets
  const int sr = lane >> 3, sc = lane & 7;
  unsigned a_voff[A_IPW];
#pragma unroll
  for (int i = 0; i < A_IPW; ++i) {
    const int row = (i * 8 + wave) * 8 + sr;                // piece i * 8 + wave covers rows 8 piece .. + 7 of the 128-row tile
    a_voff[i] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)(sc ^ ((row >> 1) & 7)) * 16u;
  }
  const unsigned w_voff = (unsigned)lane * 16u;             // a weight unit is 20 contiguous 1 KiB pieces; piece i * 8 + wave
  auto issue_a = [&](const unsigned char* abase, int slot) {
#pragma unroll
    for (int i = 0; i < A_IPW; ++i) glds16(a_voff[i], abase, lds0 + slot * A_UNIT + (i * 8 + wave) * 1024);
  };
  auto issue_w = [&](const unsigned char* wbase, int slot) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || wave < 4) glds16(w_voff, wbase + (i * 8 + wave) * 1024, lds0 + OFF_W + slot * W_UNIT + (i * 8 + wave) * 1024);
  };

  // ---- stream cursors (scalar): next weight unit / activation K-tile to request
  int wq = 0, wu = 0, wslot = 0;                            // weight: tile index in this work-group's list, unit inside the tile, ring slot
  int aq = 0, as = 0, aslot = 0;
  auto a_ptr = [&](int q, int s) { return reinterpret_cast<const unsigned char*>(p.A + (size_t)(((int)blockIdx.x + q * G) * BM) * p.lda) + (size_t)s * 128; };
  auto next_w = [&]() { if (wq < Q) issue_w(p.Wp + (size_t)wu * W_UNIT, wslot); if (++wu == PH) { wu = 0; ++wq; } wslot = wslot + 1 == NW ? 0 : wslot + 1; };
  auto next_a = [&]() { if (aq < Q) issue_a(a_ptr(aq, as), aslot); if (++as == nk) { as = 0; ++aq; } aslot ^= 1; };

  // ---- fragment read addresses
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned a_rd = (unsigned)(wm * 64 + fr) * 128u + (unsigned)((fq ^ (fr >> 1)) << 4);      // + mt * 2048, ^ 64 for the second K-slab of the tile
  const unsigned w_rd = OFF_W + (unsigned)(wc * 80 + fr) * 64u + (unsigned)((fq ^ ((fr >> 1) & 3)) << 4);      // + nt * 1024

  f32x4 acc[MT][NT];
  auto zero_acc = [&]() {
    float z = 0.f;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){z, z, z, z};
  };
  zero_acc();
  bf8_t fa[MT], fb[NH];

  // in flight behind the per-phase wait: four weight units and one activation tile
  constexpr int N0 = 4 * 3 + A_IPW, N1 = 4 * 2 + A_IPW;
  int rslot = 0;                                            // ring slot of the weight unit the current phase reads
  // one phase.  KK: K-slab inside the activation tile (0 / 1), HALF: column half; reads the activation fragments when HALF == 0
  auto phase = [&](auto kkc, auto halfc, int abuf, const bool stores_behind, const bool more) {
    constexpr int KK = decltype(kkc)::value, HALF = decltype(halfc)::value;
    if constexpr (HALF == 0) {
      const unsigned char* ab = smem + abuf * A_UNIT;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const bf8_t*>(ab + ((a_rd + mt * 2048) ^ (KK << 6)));
      __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned char* wb = smem + rslot * W_UNIT;
#pragma unroll
    for (int nt = 0; nt < NH; ++nt) fb[nt] = *reinterpret_cast<const bf8_t*>(wb + w_rd + nt * 1024);
    __builtin_amdgcn_sched_barrier(0);
    next_w();                                               // weight unit five phases ahead, into the slot read one phase ago
    if constexpr (KK == 1 && HALF == 1) next_a();           // activation tile two K-tiles ahead, into the buffer read one phase ago
    wait_lgkm<0>();                                         // this phase's reads retired before its first barrier (see the hazard notes in gemm_nt8.hip)
    if (!more) wait_vmcnt<0>();
    else if (stores_behind) wait_vmcnt<63>();               // the operands needed next were requested before the epilogue's ~84 stores: any count <= 63 covers them
    else if (grp == 0) wait_vmcnt<N0>(); else wait_vmcnt<N1>();
    bar();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NH; ++nt)
        acc[mt][HALF * NH + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][HALF * NH + nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    rslot = rslot + 1 == NW ? 0 : rslot + 1;
  };

  // ---- prologue: A(0), W(0..4), A(1) requested; A(0) and W(0) landed
  next_a();
  for (int i = 0; i < 5; ++i) next_w();
  next_a();
  // the counted wait assumes that all seven requests were issued: a work-group whose whole stream is ONE K-tile (K == 64, one tile) issued only
  // A(0) and W(0..3) -- fewer pieces than N0 / N1 allow in flight, the wait would return at once and phase 0 would read LDS before anything landed
  if (Q * nk < 2) wait_vmcnt<0>();
  else if (grp == 0) wait_vmcnt<N0>(); else wait_vmcnt<N1>();
  bar();

  const int c = lane & 15, q4 = lane >> 4;                  // accumulator layout
  const int rsr = lane >> 2, rsq = lane & 3;                // row layout
  const int pa = to_rows_addr(lane);
  float* scratch = reinterpret_cast<float*>(smem + OFF_S);  // [128 rows][4 column waves][2]
  int abuf = 0;
  long done_phases = 0;
  const long total_phases = (long)Q * PH;
#pragma clang loop unroll(disable)
  for (int q = 0; q < Q; ++q) {
    if (grp == 1) bar();                                    // group 1 drops one barrier behind
#pragma clang loop unroll(disable)
    for (int s = 0; s < nk; ++s) {
      const bool sb = q > 0 && s == 0;
      // `more`: will anything requested AFTER this phase's wait still be needed -- else drain (the last phases of the stream)
      phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, abuf, sb, done_phases + 6 <= total_phases); ++done_phases;
      phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, abuf, sb, done_phases + 6 <= total_phases); ++done_phases;
      phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, abuf, sb, done_phases + 6 <= total_phases); ++done_phases;
      phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, abuf, sb, done_phases + 6 <= total_phases); ++done_phases;
      abuf ^= 1;
    }
    if (grp == 0) bar();                                    // both groups aligned for the epilogue

    // =========================== epilogue: x = acc + bias + resid ; h = LayerNorm(x) ===========================
    const int m0 = ((int)blockIdx.x + q * G) * BM;
    const size_t o0 = (size_t)(m0 + wm * 64 + rsr) * BN + wc * 160 + rsq * 4;        // row layout: + i * 16 * BN + j * 16
    const size_t rstep = (size_t)16 * BN;
    const unsigned bias_off = (unsigned)(wc * 160 + q4 * 4) * 4u;                      // accumulator layout: bias columns of this lane, bytes
    const unsigned row_off = (unsigned)((wm * 64 + rsr) * BN + wc * 160 + rsq * 4) * 4u;     // row layout: this lane's first element inside the tile, bytes (fp32)
    const float* res_t = p.resid + (size_t)m0 * BN;                                    // wave-uniform tile bases
    float* xo = p.x_out + o0;
    float s1[MT] = {0.f, 0.f, 0.f, 0.f};
    {
      // group = one 128-byte line of x (tiles 2a, 2a+1) x two row blocks; the stores of the two halves leave back to back (gemm_epi8.h: half
      // lines written a group apart are evicted half-written from the L2 during the burst and cost 17 % extra write traffic here)
      constexpr int GL = 2 + 4, GS = 4, NG = (NT / 2) * (MT / 2);
      auto load = [&](auto gc, u32x4 (&r)[GL]) {
        constexpr int g = decltype(gc)::value, a2 = 2 * (g / (MT / 2)), i2 = 2 * (g % (MT / 2));
        gload16s<a2 * 64>(r[0], bias_off, p.bias);
        gload16s<(a2 + 1) * 64>(r[1], bias_off, p.bias);
        gload16s<a2 * 64>(r[2], row_off, res_t + i2 * rstep);
        gload16s<(a2 + 1) * 64>(r[3], row_off, res_t + i2 * rstep);
        gload16s<a2 * 64>(r[4], row_off, res_t + (i2 + 1) * rstep);
        gload16s<(a2 + 1) * 64>(r[5], row_off, res_t + (i2 + 1) * rstep);
      };
      auto finish = [&](auto gc, const u32x4 (&r)[GL]) {
        constexpr int g = decltype(gc)::value, a2 = 2 * (g / (MT / 2)), i2 = 2 * (g % (MT / 2));
        float h0, h1, h2, h3;
        g8::static_for([&](auto uc) {
          constexpr int u = decltype(uc)::value, j = a2 + (u & 1), i = i2 + (u >> 1);
          u32x4 a;
          a.x = __builtin_bit_cast(unsigned, acc[i][j][0] + as_f(r[u & 1].x)); a.y = __builtin_bit_cast(unsigned, acc[i][j][1] + as_f(r[u & 1].y));
          a.z = __builtin_bit_cast(unsigned, acc[i][j][2] + as_f(r[u & 1].z)); a.w = __builtin_bit_cast(unsigned, acc[i][j][3] + as_f(r[u & 1].w));
          a = lane_perm(pa, a);                              // row layout from here on
          const u32x4 t = r[2 + u];
          const float x0 = as_f(a.x) + as_f(t.x), x1 = as_f(a.y) + as_f(t.y), x2 = as_f(a.z) + as_f(t.z), x3 = as_f(a.w) + as_f(t.w);
          acc[i][j] = (f32x4){x0, x1, x2, x3};
          s1[i] += (x0 + x1) + (x2 + x3);
          if constexpr ((u & 1) == 0) { h0 = x0; h1 = x1; h2 = x2; h3 = x3; }
          else {
            __builtin_amdgcn_sched_barrier(0);
            gst(xo + i * rstep + (j - 1) * 16, h0, h1, h2, h3, GLN_X_NT);
            gst(xo + i * rstep + j * 16, x0, x1, x2, x3, GLN_X_NT);
            __builtin_amdgcn_sched_barrier(0);
          }
        }, std::make_integer_sequence<int, 4>{});
      };
      run_groups<NG, GL, GS>(load, finish);
    }
    // row statistics over this wave's 160 columns: mean, then centred squares (4 lanes per row -> quad reduction)
    float mw[MT], m2[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s = s1[i];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
      mw[i] = s * (1.0f / 160.0f);
      float qv = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float a = acc[i][j][0] - mw[i], b = acc[i][j][1] - mw[i], cc = acc[i][j][2] - mw[i], d = acc[i][j][3] - mw[i];
        qv += (a * a + b * b) + (cc * cc + d * d);
      }
      qv += __shfl_xor(qv, 1, 64); qv += __shfl_xor(qv, 2, 64);
      m2[i] = qv;
      if (rsq == 0) *reinterpret_cast<float2*>(scratch + ((wm * 64 + i * 16 + rsr) * 4 + wc) * 2) = make_float2(mw[i], qv);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();                                                  // all eight waves (the groups are aligned here)
    float mean[MT], rstd[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float4 p01 = *reinterpret_cast<const float4*>(scratch + (wm * 64 + i * 16 + rsr) * 8);
      const float4 p23 = *reinterpret_cast<const float4*>(scratch + (wm * 64 + i * 16 + rsr) * 8 + 4);
      const float mu = ((p01.x + p01.z) + (p23.x + p23.z)) * 0.25f;
      const float d0 = p01.x - mu, d1 = p01.z - mu, d2 = p23.x - mu, d3 = p23.z - mu;
      const float M2 = ((p01.y + p01.w) + (p23.y + p23.w)) + 160.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));      // Chan: equal counts
      mean[i] = mu;
      rstd[i] = rsqrtf(M2 * (1.0f / 640.0f) + p.eps);
      if (wc == 0 && rsq == 0) {
        if (p.mean) p.mean[m0 + wm * 64 + i * 16 + rsr] = mu;
        if (p.rstd) p.rstd[m0 + wm * 64 + i * 16 + rsr] = rstd[i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();                                                  // scratch free again before the next tile's epilogue can write it
    {
      // pass 2: h = (x - mean) * rstd * gamma + beta, bf16.  In the row layout a lane holds 4 consecutive columns of every tile = 8 bytes of h;
      // neighbouring lanes swap halves of a tile PAIR (even lane: both halves of tile 2a, odd lane: of tile 2a+1) so that every lane stores
      // 16 bytes and an instruction covers 64 contiguous bytes per row.  A wave block is 2.5 lines of h: the tile pair that shares its line with
      // the neighbouring wave goes first, the pairs of a whole line follow one another.
      constexpr int GL = 4, GS = MT, NG = NT / 2;
      const unsigned gb_off = (unsigned)(wc * 160 + rsq * 4) * 4u;
      bf16_t* ho = p.h_out + (size_t)(m0 + wm * 64 + rsr) * BN + wc * 160 + (rsq & 1) * 16 + (rsq >> 1) * 8;
      const bool odd = (rsq & 1) != 0;
      auto pass2 = [&](auto oddc) {
        constexpr bool ODDW = decltype(oddc)::value;
        auto load = [&](auto gc, u32x4 (&r)[GL]) {
          constexpr int g = decltype(gc)::value, j = 2 * (ODDW ? g : (g + NG - 1) % NG);
          gload16s<j * 64>(r[0], gb_off, p.gamma); gload16s<j * 64>(r[1], gb_off, p.beta);
          gload16s<(j + 1) * 64>(r[2], gb_off, p.gamma); gload16s<(j + 1) * 64>(r[3], gb_off, p.beta);
        };
        auto finish = [&](auto gc, const u32x4 (&r)[GL]) {
          constexpr int g = decltype(gc)::value, a = ODDW ? g : (g + NG - 1) % NG, j = 2 * a;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            u32x2 w[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const u32x4 ga = r[2 * t], be = r[2 * t + 1];
              const float h0 = (acc[i][j + t][0] - mean[i]) * rstd[i] * as_f(ga.x) + as_f(be.x), h1 = (acc[i][j + t][1] - mean[i]) * rstd[i] * as_f(ga.y) + as_f(be.y);
              const float h2 = (acc[i][j + t][2] - mean[i]) * rstd[i] * as_f(ga.z) + as_f(be.z), h3 = (acc[i][j + t][3] - mean[i]) * rstd[i] * as_f(ga.w) + as_f(be.w);
              w[t].x = pack2bf(h0, h1); w[t].y = pack2bf(h2, h3);
            }
            const unsigned sx = odd ? w[0].x : w[1].x, sy = odd ? w[0].y : w[1].y;              // what the neighbour needs
            const unsigned rx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xF, 0xF, false);      // quad_perm [1, 0, 3, 2]
            const unsigned ry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xF, 0xF, false);
            u32x4 o;
            o.x = odd ? rx : w[0].x; o.y = odd ? ry : w[0].y; o.z = odd ? w[1].x : rx; o.w = odd ? w[1].y : ry;
            gst(reinterpret_cast<u32x4*>(ho + i * rstep + a * 32), o, 0);
          }
        };
        run_groups<NG, GL, GS>(load, finish);
      };
      if (wc & 1) pass2(std::true_type{}); else pass2(std::false_type{});
    }
    zero_acc();
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// FOUR-WAVE form, two work-groups per CU (round 5).  The launch is an HBM kernel (1.01 GB against 107 GFLOP at K = 640): the eight-wave kernel above
// spends 70 us in its K loops with the HBM idle and 183 us in its epilogues with the matrix pipe idle, and every CU reaches both together.  Here a
// work-group is ONE wave group (4 waves, one per SIMD, the same 64 x 160 wave tile) on a 64 x 640 tile with half the LDS (76 KB: three weight
// units + two activation tiles), so two independent work-groups share a CU and one's K loop runs under the other's epilogue -- they drift apart
// by themselves (a work-group waiting on its stores does not hold the other back), and chip-wide 512 work-groups keep the memory system busy.
// The K loop is not software-pipelined (read -> wait -> 20 MFMAs -> counted vmcnt -> barrier): it has the partner work-group's wave to fill the SIMD
// and only a quarter of the launch's time to account for.  Same packed weight, same epilogue arithmetic (bit-identical results).
namespace g1 {
constexpr int BM = 64, NWU = 3;
constexpr int A_UNIT = BM * 128;                                                           // 8 KB: 8 pieces, 2 per wave
constexpr int OFF_W = NA * A_UNIT, OFF_S = OFF_W + NWU * W_UNIT, LDS = OFF_S + 2048;       // 16 + 60 + 2 KB
static_assert(2 * LDS <= 160 * 1024, "two work-groups per CU");
}

__global__ void __launch_bounds__(256, 2) k_gemm_ln4(const LnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave;                                      // wave tile: all 64 rows, columns 160 wc .. +159
  const int nk = p.K >> 6;
  const int PH = 4 * nk;                                    // phases (= weight units) per output tile
  const int G = gridDim.x;
  const int Q = p.tiles > (int)blockIdx.x ? (p.tiles - (int)blockIdx.x + G - 1) / G : 0;
  if (Q == 0) return;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  // de-phasing: the work-groups of the grid's second half (the second work-group of every CU under in-order dispatch) start `late_ticks` of the
  // 100 MHz wall clock late, so that their epilogues fall on the first half's K loops
  if (p.late_ticks > 0 && (int)blockIdx.x >= (G >> 1)) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < p.late_ticks) __builtin_amdgcn_s_sleep(16);
  }

  const int sr = lane >> 3, sc = lane & 7;
  unsigned a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 4 + wave) * 8 + sr;                // piece i * 4 + wave covers rows 8 piece .. + 7 of the 64-row tile
    a_voff[i] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)(sc ^ ((row >> 1) & 7)) * 16u;
  }
  const unsigned w_voff = (unsigned)lane * 16u;
  auto issue_a = [&](const unsigned char* abase, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_voff[i], abase, lds0 + slot * g1::A_UNIT + (i * 4 + wave) * 1024);
  };
  auto issue_w = [&](const unsigned char* wbase, int slot) {
#pragma unroll
    for (int i = 0; i < 5; ++i) glds16(w_voff, wbase + (i * 4 + wave) * 1024, lds0 + g1::OFF_W + slot * W_UNIT + (i * 4 + wave) * 1024);
  };
  int wq = 0, wu = 0, wslot = 0;
  int aq = 0, as = 0, aslot = 0;
  auto a_ptr = [&](int q, int s) { return reinterpret_cast<const unsigned char*>(p.A + (size_t)(((int)blockIdx.x + q * G) * g1::BM) * p.lda) + (size_t)s * 128; };
  auto next_w = [&]() { if (wq < Q) issue_w(p.Wp + (size_t)wu * W_UNIT, wslot); if (++wu == PH) { wu = 0; ++wq; } wslot = wslot + 1 == g1::NWU ? 0 : wslot + 1; };
  auto next_a = [&]() { if (aq < Q) issue_a(a_ptr(aq, as), aslot); if (++as == nk) { as = 0; ++aq; } aslot ^= 1; };

  const int fr = lane & 15, fq = lane >> 4;
  const unsigned a_rd = (unsigned)fr * 128u + (unsigned)((fq ^ (fr >> 1)) << 4);
  const unsigned w_rd = g1::OFF_W + (unsigned)(wc * 80 + fr) * 64u + (unsigned)((fq ^ ((fr >> 1) & 3)) << 4);

  f32x4 acc[MT][NT];
  auto zero_acc = [&]() {
    float z = 0.f;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){z, z, z, z};
  };
  zero_acc();
  bf8_t fa[MT], fb[NH];

  // One phase = (K-slab KK of the activation tile, column half HALF) = one weight unit.  Entered behind a barrier that every wave passed after it
  // had (i) waited for its share of this phase's operands and (ii) retired its reads of the previous phase: the unit two phases ahead goes into the
  // slot read in the previous phase.  `tail`: 0 counted wait (the next unit was requested one phase ago: only this phase's requests are younger),
  // 1 the next unit was requested BEFORE the epilogue (its >= 63 younger stores and loads: any count <= 63 covers it), 2 drain.
  int rslot = 0;
  bool skip_w = false;                                      // the unit this phase would request went out ahead of the last epilogue
  auto phase = [&](auto kkc, auto halfc, int abuf, const int tail, const bool a_too) {
    constexpr int KK = decltype(kkc)::value, HALF = decltype(halfc)::value;
    if constexpr (HALF == 0) {
      const unsigned char* ab = smem + abuf * g1::A_UNIT;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const bf8_t*>(ab + ((a_rd + mt * 2048) ^ (KK << 6)));
    }
    const unsigned char* wb = smem + rslot * W_UNIT;
#pragma unroll
    for (int nt = 0; nt < NH; ++nt) fb[nt] = *reinterpret_cast<const bf8_t*>(wb + w_rd + nt * 1024);
    __builtin_amdgcn_sched_barrier(0);
    if (!skip_w) next_w();                                  // two phases ahead, into the slot read one phase ago
    skip_w = false;
    if (a_too) next_a();                                    // (KK, HALF) == (1, 1): the activation tile two K-tiles ahead, into the buffer whose last read was one phase ago
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NH; ++nt)
        acc[mt][HALF * NH + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][HALF * NH + nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // younger than the next phase's unit: this phase's weight request (5 pieces per wave) and, in the phases (1, 1) / (0, 0), the activation tile
    // requested in / one phase before this one (2 pieces; it is read four / three phases from now)
    if (tail == 2) wait_vmcnt<0>();
    else if (tail == 1) wait_vmcnt<63>();
    else if (KK == HALF) wait_vmcnt<5 + 2>(); else wait_vmcnt<5>();
    bar();
    rslot = rslot + 1 == g1::NWU ? 0 : rslot + 1;
  };

  // ---- prologue: A(0), W(0), W(1), A(1) requested; A(0), W(0) landed
  next_a(); next_w(); next_w(); next_a();
  if (Q * nk < 2) wait_vmcnt<0>(); else wait_vmcnt<5 + 2>();
  bar();

  const int c = lane & 15, q4 = lane >> 4;
  const int rsr = lane >> 2, rsq = lane & 3;
  const int pa = to_rows_addr(lane);
  float* scratch = reinterpret_cast<float*>(smem + g1::OFF_S);      // [64 rows][4 column waves][2]
  int abuf = 0;
  long done = 0;
  const long total = (long)Q * PH;
  int behind = 0;                                           // phases whose NEXT unit was requested before the last epilogue
#pragma clang loop unroll(disable)
  for (int q = 0; q < Q; ++q) {
#pragma clang loop unroll(disable)
    for (int s = 0; s < nk; ++s) {
      // tail of phase d: the unit of phase d + 1 must have landed; nothing is requested beyond the stream's end
      // (the counted waits assume that every request of the steady state was really issued: over the stream's last two K-tiles, where the activation
      // and then the weight requests run out, the waits drain instead)
      auto tl = [&]() { const int t = done + 8 > total ? 2 : (behind > 0 ? 1 : 0); if (behind > 0) --behind; ++done; return t; };
      { const int t = tl(); phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, abuf, t, false); }
      { const int t = tl(); phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, abuf, t, false); }
      { const int t = tl(); phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, abuf, t, false); }
      { const int t = tl(); phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, abuf, t, true); }
      abuf ^= 1;
    }
    // the slot of the tile's last unit is free (its reads retired before the closing barrier): the third unit of the next tile is requested ahead of the
    // epilogue, so that the first two phases of the next tile wait for nothing issued behind the stores
    if (q + 1 < Q) { next_w(); skip_w = true; behind = 2; }

    // =========================== epilogue: x = acc + bias + resid ; h = LayerNorm(x) (the arithmetic of k_gemm_ln) ===========================
    const int m0 = ((int)blockIdx.x + q * G) * g1::BM;
    const size_t o0 = (size_t)(m0 + rsr) * BN + wc * 160 + rsq * 4;
    const size_t rstep = (size_t)16 * BN;
    const unsigned bias_off = (unsigned)(wc * 160 + q4 * 4) * 4u;
    const unsigned row_off = (unsigned)(rsr * BN + wc * 160 + rsq * 4) * 4u;
    const float* res_t = p.resid + (size_t)m0 * BN;
    float* xo = p.x_out + o0;
    float s1[MT] = {0.f, 0.f, 0.f, 0.f};
    {
      constexpr int GL = 2 + 4, GS = 4, NG = (NT / 2) * (MT / 2);
      auto load = [&](auto gc, u32x4 (&r)[GL]) {
        constexpr int g = decltype(gc)::value, a2 = 2 * (g / (MT / 2)), i2 = 2 * (g % (MT / 2));
        gload16s<a2 * 64>(r[0], bias_off, p.bias);
        gload16s<(a2 + 1) * 64>(r[1], bias_off, p.bias);
        gload16s<a2 * 64>(r[2], row_off, res_t + i2 * rstep);
        gload16s<(a2 + 1) * 64>(r[3], row_off, res_t + i2 * rstep);
        gload16s<a2 * 64>(r[4], row_off, res_t + (i2 + 1) * rstep);
        gload16s<(a2 + 1) * 64>(r[5], row_off, res_t + (i2 + 1) * rstep);
      };
      auto finish = [&](auto gc, const u32x4 (&r)[GL]) {
        constexpr int g = decltype(gc)::value, a2 = 2 * (g / (MT / 2)), i2 = 2 * (g % (MT / 2));
        float h0, h1, h2, h3;
        g8::static_for([&](auto uc) {
          constexpr int u = decltype(uc)::value, j = a2 + (u & 1), i = i2 + (u >> 1);
          u32x4 a;
          a.x = __builtin_bit_cast(unsigned, acc[i][j][0] + as_f(r[u & 1].x)); a.y = __builtin_bit_cast(unsigned, acc[i][j][1] + as_f(r[u & 1].y));
          a.z = __builtin_bit_cast(unsigned, acc[i][j][2] + as_f(r[u & 1].z)); a.w = __builtin_bit_cast(unsigned, acc[i][j][3] + as_f(r[u & 1].w));
          a = lane_perm(pa, a);
          const u32x4 t = r[2 + u];
          const float x0 = as_f(a.x) + as_f(t.x), x1 = as_f(a.y) + as_f(t.y), x2 = as_f(a.z) + as_f(t.z), x3 = as_f(a.w) + as_f(t.w);
          acc[i][j] = (f32x4){x0, x1, x2, x3};
          s1[i] += (x0 + x1) + (x2 + x3);
          if constexpr ((u & 1) == 0) { h0 = x0; h1 = x1; h2 = x2; h3 = x3; }
          else {
            __builtin_amdgcn_sched_barrier(0);
            gst(xo + i * rstep + (j - 1) * 16, h0, h1, h2, h3, GLN_X_NT);
            gst(xo + i * rstep + j * 16, x0, x1, x2, x3, GLN_X_NT);
            __builtin_amdgcn_sched_barrier(0);
          }
        }, std::make_integer_sequence<int, 4>{});
      };
      run_groups<NG, GL, GS>(load, finish);
    }
    float mw[MT], m2[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s = s1[i];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
      mw[i] = s * (1.0f / 160.0f);
      float qv = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float a = acc[i][j][0] - mw[i], b = acc[i][j][1] - mw[i], cc = acc[i][j][2] - mw[i], d = acc[i][j][3] - mw[i];
        qv += (a * a + b * b) + (cc * cc + d * d);
      }
      qv += __shfl_xor(qv, 1, 64); qv += __shfl_xor(qv, 2, 64);
      m2[i] = qv;
      if (rsq == 0) *reinterpret_cast<float2*>(scratch + ((i * 16 + rsr) * 4 + wc) * 2) = make_float2(mw[i], qv);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    float mean[MT], rstd[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float4 p01 = *reinterpret_cast<const float4*>(scratch + (i * 16 + rsr) * 8);
      const float4 p23 = *reinterpret_cast<const float4*>(scratch + (i * 16 + rsr) * 8 + 4);
      const float mu = ((p01.x + p01.z) + (p23.x + p23.z)) * 0.25f;
      const float d0 = p01.x - mu, d1 = p01.z - mu, d2 = p23.x - mu, d3 = p23.z - mu;
      const float M2 = ((p01.y + p01.w) + (p23.y + p23.w)) + 160.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
      mean[i] = mu;
      rstd[i] = rsqrtf(M2 * (1.0f / 640.0f) + p.eps);
      if (wc == 0 && rsq == 0) {
        if (p.mean) p.mean[m0 + i * 16 + rsr] = mu;
        if (p.rstd) p.rstd[m0 + i * 16 + rsr] = rstd[i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    {
      constexpr int GL = 4, GS = MT, NG = NT / 2;
      const unsigned gb_off = (unsigned)(wc * 160 + rsq * 4) * 4u;
      bf16_t* ho = p.h_out + (size_t)(m0 + rsr) * BN + wc * 160 + (rsq & 1) * 16 + (rsq >> 1) * 8;
      const bool odd = (rsq & 1) != 0;
      auto pass2 = [&](auto oddc) {
        constexpr bool ODDW = decltype(oddc)::value;
        auto load = [&](auto gc, u32x4 (&r)[GL]) {
          constexpr int g = decltype(gc)::value, j = 2 * (ODDW ? g : (g + NG - 1) % NG);
          gload16s<j * 64>(r[0], gb_off, p.gamma); gload16s<j * 64>(r[1], gb_off, p.beta);
          gload16s<(j + 1) * 64>(r[2], gb_off, p.gamma); gload16s<(j + 1) * 64>(r[3], gb_off, p.beta);
        };
        auto finish = [&](auto gc, const u32x4 (&r)[GL]) {
          constexpr int g = decltype(gc)::value, a = ODDW ? g : (g + NG - 1) % NG, j = 2 * a;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            u32x2 w[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const u32x4 ga = r[2 * t], be = r[2 * t + 1];
              const float h0 = (acc[i][j + t][0] - mean[i]) * rstd[i] * as_f(ga.x) + as_f(be.x), h1 = (acc[i][j + t][1] - mean[i]) * rstd[i] * as_f(ga.y) + as_f(be.y);
              const float h2 = (acc[i][j + t][2] - mean[i]) * rstd[i] * as_f(ga.z) + as_f(be.z), h3 = (acc[i][j + t][3] - mean[i]) * rstd[i] * as_f(ga.w) + as_f(be.w);
              w[t].x = pack2bf(h0, h1); w[t].y = pack2bf(h2, h3);
            }
            const unsigned sx = odd ? w[0].x : w[1].x, sy = odd ? w[0].y : w[1].y;
            const unsigned rx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xF, 0xF, false);
            const unsigned ry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xF, 0xF, false);
            u32x4 o;
            o.x = odd ? rx : w[0].x; o.y = odd ? ry : w[0].y; o.z = odd ? w[1].x : rx; o.w = odd ? w[1].y : ry;
            gst(reinterpret_cast<u32x4*>(ho + i * rstep + a * 32), o, 0);
          }
        };
        run_groups<NG, GL, GS>(load, finish);
      };
      if (wc & 1) pass2(std::true_type{}); else pass2(std::false_type{});
    }
    zero_acc();
  }
}

}  // namespace gln

// W bf16 [640, K] -> packed units for oneprot_gemm_bf16_nt_resid_ln (K * 640 * 2 bytes)
extern "C" int oneprot_gemm_ln_pack_weight(const void* W, void* Wp, int N, int K, void* stream) {
  if (!W || !Wp || N != gln::BN || K <= 0 || (K & 63) || (((uintptr_t)W | (uintptr_t)Wp) & 15)) return OP_EINVAL;
  const int n = K / 32 * 2 * 320 * 4;
  int blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gln::k_pack_w, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, K, (u32x4*)Wp);
  return launch_status();
}

// test / experiment hook: 0 = eight-wave kernel (128-row tiles, one work-group per CU; default), 1 = four-wave kernel (64-row tiles, two per CU)
static int g_gln_form = 0;
extern "C" void oneprot_gemm_ln_form(int form) { g_gln_form = form; }
extern "C" int oneprot_gemm_ln_form_get(void) { return g_gln_form; }

extern "C" int oneprot_gemm_bf16_nt_resid_ln(const void* A, const void* Wp, int64_t M, int N, int K, int lda, const float* bias, const float* resid, float* x_out,
                                             const float* gamma, const float* beta, float eps, void* h_out, float* mean, float* rstd, void* stream) {
  if (!A || !Wp || !bias || !resid || !x_out || !gamma || !beta || !h_out || M <= 0 || M > 0x7fffffff) return OP_EINVAL;
  if (N != gln::BN || (M % gln::BM) || K <= 0 || (K & 63) || (lda & 7) || lda < K || (size_t)gln::BM * lda * 2 >= (1ull << 31)) return OP_EINVAL;
  if (((uintptr_t)A | (uintptr_t)Wp | (uintptr_t)bias | (uintptr_t)resid | (uintptr_t)x_out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)h_out) & 15) return OP_EINVAL;
  static bool configured = false;
  static int n_cu = 0;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)gln::k_gemm_ln, hipFuncAttributeMaxDynamicSharedMemorySize, gln::LDS) != hipSuccess) return OP_ELAUNCH;
    if (hipFuncSetAttribute((const void*)gln::k_gemm_ln4, hipFuncAttributeMaxDynamicSharedMemorySize, gln::g1::LDS) != hipSuccess) return OP_ELAUNCH;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return OP_ELAUNCH;
    n_cu = prop.multiProcessorCount;
    configured = true;
  }
  gln::LnArgs a;
  a.A = (const bf16_t*)A; a.Wp = (const unsigned char*)Wp; a.M = (int)M; a.K = K; a.lda = lda; a.bias = bias; a.resid = resid; a.x_out = x_out;
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.h_out = (bf16_t*)h_out; a.mean = mean; a.rstd = rstd;
  a.late_ticks = (g_gln_form >> 8) * 100;                       // (experiment: form | delay in microseconds << 8)
  if ((g_gln_form & 255) == 1) {
    a.tiles = (int)(M / gln::g1::BM);
    const int grid = a.tiles < 2 * n_cu ? a.tiles : 2 * n_cu;
    hipLaunchKernelGGL(gln::k_gemm_ln4, dim3(grid), dim3(256), gln::g1::LDS, (hipStream_t)stream, a);
  } else {
    a.tiles = (int)(M / gln::BM);
    const int grid = a.tiles < n_cu ? a.tiles : n_cu;
    hipLaunchKernelGGL(gln::k_gemm_ln, dim3(grid), dim3(512), gln::LDS, (hipStream_t)stream, a);
  }
  return launch_status();
}
