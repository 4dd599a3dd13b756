// Epilogues of the persistent 8-phase NT GEMM (gemm_nt8.hip), DIRECT accumulator layout of gemm_epi.h (lane (c = lane & 15, q = lane >> 4) owns
// C[token c][slots q*4 .. q*4+3] of every 16 x 16 tile; pair map for bf16 outputs, natural map for fp32 outputs and QKV / RoPE).
//
// What is different from gemm_epilogue_direct: the kernel is persistent, and the stores of one tile must be able to drain while the next tile's K
// loop runs.  vmcnt retires in issue order, so a wait for ANY load issued after a store also waits for that store (thousands of cycles until the
// write is acknowledged).  Therefore
//   * the accumulators start at zero and the bias is added here, so that no operand load follows the stores of the previous tile;
//   * every operand of the epilogue (bias, GELU', residual, RoPE tables) is fetched in GROUPS that run two groups ahead of the stores: the
//     loads of group g+2 are issued before the stores of group g, by inline asm (hipcc would put its own wait at the first use, behind the
//     stores), and each group is waited for with a hand-counted vmcnt that leaves exactly the younger stores and loads in flight;
//   * groups are column blocks (8 columns x all rows of the wave tile for the pair map, one 16-column tile x all rows for the natural map),
//     so a group's bias is 2 (1) float4 per lane and the register cost of running ahead stays at 16-80 registers.
//   * LANE TRANSPOSE before every store / after every operand load.  In the accumulator layout the 16 lanes of a lane group hold 16 DIFFERENT rows
//     (16 bytes each) and the four 16-byte pieces of one row's 64 bytes sit 16 lanes apart: the memory pipeline coalesces neighbouring lanes
//     only, so such a store leaves the CU as 64 requests of 16 bytes -- measured 13 B/clk per CU however few CUs are active, against 48-57 B/clk for
//     the same bytes with the pieces of a row in neighbouring lanes (tools/ab/store_probe.hip).  ds_bpermute_b32 (LDS crossbar, no LDS memory)
//     moves lane (q*16 + c) to lane (c*4 + q): four per 16-byte store, and the store becomes 16 requests of 64 bytes.
#pragma once
#include "gemm_epi.h"
#include "sched_ws.h"

namespace g8 {

#ifdef G8_STAMP      // diagnostic build only (tools/ab/g8_stamps.py): s_memtime after every group of the pair-map epilogue, first wave of each wave group of work-group 0
#define G8_ESTAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    reinterpret_cast<unsigned long long*>(p.out2)[768 + (threadIdx.x >> 8) * 16 + (k)] = t_; } } while (0)
#else
#define G8_ESTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void gload16(u32x4& d, const void* ptr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(ptr) : "memory"); }
// saddr forms: wave-uniform 64-bit base in SGPRs + one 32-bit per-lane byte offset + an immediate.  The per-lane 64-bit pointers of the plain forms
// (one VGPR pair per output / operand tensor and per row block) were what the two-output GELU and the QKV + RoPE epilogues spilled next to 160
// accumulators -- and a spill reload is a vector-memory load that retires behind the epilogue's own stores.  (The cache policy of the
// stores is a compile-time property of the epilogue kind, see gst16_s.)
template <int IMM> __device__ __forceinline__ void gload16_s(u32x4& d, const void* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// NTS: non-temporal (`nt`) -- the output is streamed past the L2 instead of displacing W and the activation panels.  Measured in one process
// (tools/ab/libs_ab.py, profiles/r05_libs_ab.txt (8)): -4 .. -7 % on the launches with many column tiles and a short K (QKV + RoPE 303 -> 285 us, FFN-1 428 ->
// 404, two outputs 530 -> 505, BERT / 650M QKV -4 / -2 %), +3 .. +4 % on the N = 640 data gradients with K = 1920 / 2560, nothing on the fp32 outputs:
// it is a property of the epilogue kind (QKV_ROPE, BIAS_GELU, GELU_BWD: always many column tiles over a short K), not a run-time switch.
#ifdef G8_NO_NT      // A/B builds: every store with the default policy
constexpr bool G8_NT = false;
#else
constexpr bool G8_NT = true;
#endif
template <int IMM, bool NTS = false> __device__ __forceinline__ void gst16_s(const void* sbase, unsigned voff, const u32x4& v) {
  // (s_nop 1: a store of more than 8 bytes reads its data registers late; hipcc pads its own stores against the next writer of those registers, not an asm one)
  if constexpr (NTS && G8_NT) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
template <int IMM, bool NTS = false> __device__ __forceinline__ void gst8_s(const void* sbase, unsigned voff, const u32x2& v) {
  if constexpr (NTS && G8_NT) asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 nt" ::"v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
// 8-byte load into the low half of a 16-byte register group (the one-byte gelu' codes of GELU_BWD: 8 elements per lane)
template <int IMM> __device__ __forceinline__ void gload8_s(u32x4& d, const void* sbase, unsigned voff) {
  u32x2 t;
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(t) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  d.x = t.x; d.y = t.y;
}
// wait until at most N vector-memory operations are in flight; the loaded registers are defined from here on
template <int N, int G> __device__ __forceinline__ void wait_vm_pin(u32x4 (&r)[G]) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[0]) : "n"(N) : "memory");
#pragma unroll
  for (int k = 1; k < G; ++k) asm volatile("" : "+v"(r[k]));
}
__device__ __forceinline__ float as_f(unsigned u) { return __builtin_bit_cast(float, u); }
// accumulator layout -> row layout: lane l receives the value of lane (l & 3) * 16 + (l >> 2); and back: lane l receives that of lane (l & 15) * 4 + (l >> 4)
__device__ __forceinline__ int to_rows_addr(int lane) { return ((((lane & 3) << 4) | (lane >> 2)) << 2); }
__device__ __forceinline__ int to_acc_addr(int lane) { return ((((lane & 15) << 2) | (lane >> 4)) << 2); }
__device__ __forceinline__ u32x4 lane_perm(int addr, u32x4 v) {
  u32x4 r;
  r.x = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.x); r.y = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.y);
  r.z = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.z); r.w = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.w);
  return r;
}

__device__ __forceinline__ u32x2 lane_perm2(int addr, u32x2 v) {
  u32x2 r;
  r.x = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.x); r.y = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.y);
  return r;
}

template <class F, int... Is> __device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }

// Runs NG groups: load(g, regs) issues GL loads, finish(g, regs) consumes them and issues GS stores.  Two groups of loads in flight.
template <int NG, int GL, int GS, class LoadF, class FinF>
__device__ __forceinline__ void run_groups(LoadF&& load, FinF&& finish) {
  if constexpr (GL == 0) {
    static_for([&](auto gc) { u32x4 none[1]; finish(gc, none); }, std::make_integer_sequence<int, NG>{});
  } else {
    u32x4 r[2][GL];
    load(std::integral_constant<int, 0>{}, r[0]);
    if constexpr (NG > 1) load(std::integral_constant<int, 1>{}, r[1]);
    static_for([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      // younger than the loads of group g: the stores of group g-1 and the loads of group g+1
      constexpr int younger = (g >= 1 ? GS : 0) + (g + 1 < NG ? GL : 0);
      wait_vm_pin<(younger > 63 ? 63 : younger)>(r[g & 1]);
      finish(gc, r[g & 1]);
      if constexpr (g + 2 < NG) load(std::integral_constant<int, g + 2>{}, r[g & 1]);
    }, std::make_integer_sequence<int, NG>{});
  }
}

// The same with per-group operation counts: Cnt::gl(g) loads and Cnt::gs(g) stores in group g (the one-byte gelu' tensor is moved in 16-byte pieces
// where two column units of a row block pair up and in 8-byte pieces for the lone half line of a 160-column wave block).
template <int NG, int MAXGL, class Cnt, class LoadF, class FinF>
__device__ __forceinline__ void run_groups_v(LoadF&& load, FinF&& finish) {
  u32x4 r[2][MAXGL > 0 ? MAXGL : 1];
  if constexpr (Cnt::gl(0) > 0) load(std::integral_constant<int, 0>{}, r[0]);
  if constexpr (NG > 1) { if constexpr (Cnt::gl(1) > 0) load(std::integral_constant<int, 1>{}, r[1]); }
  static_for([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    constexpr int younger = (g >= 1 ? Cnt::gs(g - 1) : 0) + (g + 1 < NG ? Cnt::gl(g + 1) : 0);
    if constexpr (Cnt::gl(g) > 0) wait_vm_pin<(younger > 63 ? 63 : younger)>(r[g & 1]);
    finish(gc, r[g & 1]);
    if constexpr (g + 2 < NG) { if constexpr (Cnt::gl(g + 2) > 0) load(std::integral_constant<int, g + 2>{}, r[g & 1]); }
  }, std::make_integer_sequence<int, NG>{});
}

// ---- store order.  One store instruction of the row layout covers 16 rows x 64 bytes: half of a 128-byte line per row.  When the other half
// follows a whole group later (thousands of cycles, while all CUs pour 40 MB of outputs through 32 MB of L2) the L2 has often evicted the
// half-written line in between and WRITE_SIZE shows 12-23 % more bytes than the tensor has (measured, round 3; the launches are bound by that
// write burst: FFN-1 with both outputs 0.57 -> 0.53 ms, plain bf16 0.40 -> 0.36 ms once fixed).  So a group is made of 4 units = {the two halves
// of a line} x {two row blocks}, and the stores of the two halves leave back to back (the first unit's packed result waits in registers for the
// second; not in the two-output GELU form, which has no registers left for that and keeps them one unit apart).  A wave block of 160 bf16
// columns is 2.5 lines: the half line it shares with the neighbouring wave (its last column pair for an even wave column, its first for an odd
// one) is a group of its own (4 row blocks) and goes FIRST, so both waves write their halves right after the barrier that precedes the epilogue.
// (Groups of 2 units with half the registers in flight were measured too: the shallower store queue costs more than the registers gain.)
template <int MT, int NP, bool ODD> struct LinePlan {      // NP column units of half a line each per wave block
  static constexpr bool LONE = (NP & 1) != 0;
  static constexpr int NSG = LONE ? MT / 4 : 0;             // groups of the lone half line
  static constexpr int S = ODD ? 0 : NP - 1, F0 = LONE && ODD ? 1 : 0;
  static constexpr int NG = NSG + (NP / 2) * (MT / 2);
  static constexpr int col(int g, int u) { return g < NSG ? S : F0 + 2 * ((g - NSG) / (MT / 2)) + (u & 1); }
  static constexpr int row(int g, int u) { return g < NSG ? g * 4 + u : 2 * ((g - NSG) % (MT / 2)) + (u >> 1); }
  static_assert(MT % 2 == 0 && (!LONE || MT % 4 == 0), "row blocks per wave");
};

// EPI in {BF16, BIAS_GELU, GELU_BWD}: pair map.  HB: bias present.  DUAL: BIAS_GELU writes gelu'(z) to out1 as well.
template <int EPI, bool HB, bool DUAL, int MT, int NT, bool ODD>
__device__ __forceinline__ void epilogue_pair_body(const GemmArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wr, int wc, int lane) {
  constexpr bool AUX = EPI == ONEPROT_EPI_GELU_BWD;
  constexpr bool NTS = EPI != ONEPROT_EPI_BF16;            // non-temporal output stores (gst16_s)
#ifndef G8_HOLD_DUAL
#define G8_HOLD_DUAL 1
#endif
  constexpr bool HOLD = !DUAL || G8_HOLD_DUAL;
  using Plan = LinePlan<MT, NT / 2, ODD>;
  constexpr int NG = Plan::NG, GL = (HB ? 4 : 0) + (AUX ? 4 : 0);
  // the one-byte gelu' tensor (out1 of DUAL, aux of GELU_BWD): a lane owns 8 bytes of a unit; in the groups made of two neighbouring column units x two
  // row blocks the lanes of a pair swap halves so that every lane moves 16 contiguous bytes (even lane: unit pp, odd lane: unit pp + 1) -- 8-byte
  // pieces ran the two-output FFN-1 launch 4 % slower than the bf16 tensor had; the lone-half-line groups (one column unit x four row blocks) keep them
  struct Cnt {
    static constexpr bool lone(int g) { return g < Plan::NSG; }
    static constexpr int gl(int g) { return (HB ? 4 : 0) + (AUX ? (lone(g) ? 4 : 2) : 0); }
    static constexpr int gs(int g) { return 4 + (DUAL ? (lone(g) ? 4 : 2) : 0); }
  };
  const int q = lane >> 4;                                  // accumulator layout: lane (c, q) owns columns q*8 .. q*8+7 of the pair, row c
  const int sr = lane >> 2, sq = lane & 3;                  // row layout (stores, GELU' loads): lane owns row sr, columns sq*8 .. sq*8+7
  const int pa = to_rows_addr(lane), pb = to_acc_addr(lane);
  // wave-uniform tile origin (element offset of the wave block's first row / column) + ONE per-lane byte offset shared by out0, out1 and aux:
  // row layout, lane owns row sr, columns sq*8 .. sq*8+7; row block i adds i * 16 * N elements to the scalar base, column unit pp is an immediate
  const size_t t0 = (size_t)(m0 + wr * (MT * 16)) * p.N + n0 + wc * (NT * 16);
  const unsigned voff = (unsigned)(sr * p.N + sq * 8) * 2u;
  const size_t rstep = (size_t)16 * p.N;
  const float* bbase = HB ? p.bias + n0 + wc * (NT * 16) : nullptr;     // accumulator layout: lane (c, q) owns columns q*8 .. q*8+7 of the pair
  const unsigned boff = (unsigned)q * 32u;
  // gelu' travels as one byte per element (gemm_epi.h: gelu_grad_encode8): out1 of the two-output GELU and aux of GELU_BWD are byte tensors, the
  // same lane owns the same 8 columns -- 8 bytes -- and `voff8` is its byte offset there
  const unsigned char* aux0 = AUX ? (const unsigned char*)p.aux + t0 : nullptr;
  const bf16_t* out0 = (const bf16_t*)p.out0 + t0;
  const unsigned char* out1 = DUAL ? (const unsigned char*)p.out1 + t0 : nullptr;
  const unsigned voff8 = (unsigned)(sr * p.N + sq * 8);
  const bool odd = (sq & 1) != 0;
  const unsigned voff16 = (unsigned)(sr * p.N + (odd ? 32 + (sq - 1) * 8 : sq * 8));      // relative to the EVEN unit of the pair
  auto load = [&](auto gc, u32x4 (&r)[GL > 0 ? GL : 1]) {
    constexpr int g = decltype(gc)::value;
    if constexpr (HB) {                                     // (a lone-half group fetches its bias twice: the count per group stays uniform)
      gload16_s<Plan::col(g, 0) * 128>(r[0], bbase, boff); gload16_s<Plan::col(g, 0) * 128 + 16>(r[1], bbase, boff);
      gload16_s<Plan::col(g, 1) * 128>(r[2], bbase, boff); gload16_s<Plan::col(g, 1) * 128 + 16>(r[3], bbase, boff);
    }
    if constexpr (AUX) {
      if constexpr (Cnt::lone(g)) {
        static_for([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          gload8_s<Plan::col(g, u) * 32>(r[(HB ? 4 : 0) + u], aux0 + Plan::row(g, u) * rstep, voff8);
        }, std::make_integer_sequence<int, 4>{});
      } else {                                              // units (0, 1) and (2, 3): neighbouring column units of one row block
        gload16_s<Plan::col(g, 0) * 32>(r[(HB ? 4 : 0) + 0], aux0 + Plan::row(g, 0) * rstep, voff16);
        gload16_s<Plan::col(g, 2) * 32>(r[(HB ? 4 : 0) + 1], aux0 + Plan::row(g, 2) * rstep, voff16);
      }
    }
  };
  auto finish = [&](auto gc, const u32x4 (&r)[GL > 0 ? GL : 1]) {
    constexpr int g = decltype(gc)::value;
    u32x4 hw; u32x2 hz;                                     // HOLD: unit 2k waits, packed, for unit 2k+1
    u32x2 cnext;                                            // GELU_BWD, paired groups: the codes of unit 2k+1, separated when unit 2k was
    static_for([&](auto uc) {
      constexpr int u = decltype(uc)::value, pp = Plan::col(g, u), i = Plan::row(g, u), bs = 2 * (u & 1);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * pp][e]; v[4 + e] = acc[i][2 * pp + 1][e]; }
      if constexpr (HB) {
        v[0] += as_f(r[bs].x); v[1] += as_f(r[bs].y); v[2] += as_f(r[bs].z); v[3] += as_f(r[bs].w);
        v[4] += as_f(r[bs + 1].x); v[5] += as_f(r[bs + 1].y); v[6] += as_f(r[bs + 1].z); v[7] += as_f(r[bs + 1].w);
      }
      if constexpr (EPI == ONEPROT_EPI_BIAS_GELU) {
        if constexpr (DUAL) {
          const u32x2 z = lane_perm2(pa, gelu_fwd_and_code8(v));
          if constexpr (Cnt::lone(g)) gst8_s<pp * 32, NTS>(out1 + i * rstep, voff8, z);
          else if constexpr ((u & 1) == 0) hz = z;
          else {                                            // hz = unit pp - 1, z = unit pp of the same rows: swap halves, one 16-byte store per lane
            const unsigned sx = odd ? hz.x : z.x, sy = odd ? hz.y : z.y;
            const unsigned rx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xF, 0xF, false);      // quad_perm [1, 0, 3, 2]
            const unsigned ry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xF, 0xF, false);
            u32x4 o;
            o.x = odd ? rx : hz.x; o.y = odd ? ry : hz.y; o.z = odd ? z.x : rx; o.w = odd ? z.y : ry;
            gst16_s<Plan::col(g, u - 1) * 32, NTS>(out1 + i * rstep, voff16, o);
          }
        } else {
          gelu_fwd_only8(v);
        }
      } else if constexpr (AUX) {
        u32x2 code;
        if constexpr (Cnt::lone(g)) { const u32x4& ra = r[(HB ? 4 : 0) + u]; code = (u32x2){ra.x, ra.y}; }
        else if constexpr ((u & 1) == 0) {                  // 16 bytes = this lane's half of units u and u + 1 and its neighbour's: swap back
          const u32x4& ra = r[(HB ? 4 : 0) + (u >> 1)];
          const unsigned sx = odd ? ra.x : ra.z, sy = odd ? ra.y : ra.w;
          const unsigned rx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xF, 0xF, false);
          const unsigned ry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xF, 0xF, false);
          code = odd ? (u32x2){rx, ry} : (u32x2){ra.x, ra.y};
          cnext = odd ? (u32x2){ra.z, ra.w} : (u32x2){rx, ry};
        } else code = cnext;
        const u32x2 t = lane_perm2(pb, code);
        gelu_grad_apply8(v, t.x, t.y);
      }
      u32x4 w; w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
      w = lane_perm(pa, w);
      if constexpr (!HOLD) gst16_s<pp * 64, NTS>(out0 + i * rstep, voff, w);
      else if constexpr ((u & 1) == 0) hw = w;
      else {
        constexpr int pp0 = Plan::col(g, u - 1), i0 = Plan::row(g, u - 1);
        __builtin_amdgcn_sched_barrier(0);
        gst16_s<pp0 * 64, NTS>(out0 + i0 * rstep, voff, hw);
        gst16_s<pp * 64, NTS>(out0 + i * rstep, voff, w);
        __builtin_amdgcn_sched_barrier(0);
      }
    }, std::make_integer_sequence<int, 4>{});
    G8_ESTAMP(g + 1);
  };
  G8_ESTAMP(0);
  run_groups_v<NG, GL, Cnt>(load, finish);
}
template <int EPI, bool HB, bool DUAL, int MT, int NT>
__device__ __forceinline__ void epilogue_pair(const GemmArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wr, int wc, int lane) {
  if constexpr (((NT / 2) & 1) != 0) {
    if (wc & 1) epilogue_pair_body<EPI, HB, DUAL, MT, NT, true>(p, acc, m0, n0, wr, wc, lane);
    else epilogue_pair_body<EPI, HB, DUAL, MT, NT, false>(p, acc, m0, n0, wr, wc, lane);
  } else epilogue_pair_body<EPI, HB, DUAL, MT, NT, false>(p, acc, m0, n0, wr, wc, lane);
}

// EPI in {F32, BIAS_RESID}: natural map, fp32 output (one 16-column tile = half a line).  DUAL: BIAS_RESID writes a bf16 copy to out1.  out0 may
// alias the residual: a lane reads exactly the elements it writes, and reads them first.
template <int EPI, bool HB, bool DUAL, int MT, int NT>
__device__ __forceinline__ void epilogue_f32(const GemmArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wr, int wc, int lane) {
  constexpr bool RES = EPI == ONEPROT_EPI_BIAS_RESID;
  static_assert(NT % 2 == 0, "whole lines per wave block");
  using Plan = LinePlan<MT, NT, false>;
  static_assert(!Plan::LONE);
  constexpr int NG = Plan::NG, GL = (HB ? 2 : 0) + (RES ? 4 : 0), GS = 4 * (DUAL ? 2 : 1);
  const int q = lane >> 4;                                  // accumulator layout: columns q*4 .. q*4+3 of tile j, row c
  const int sr = lane >> 2, sq = lane & 3;                  // row layout (stores, residual loads)
  const int pa = to_rows_addr(lane);
  const size_t o0 = (size_t)(m0 + wr * (MT * 16) + sr) * p.N + n0 + wc * (NT * 16) + sq * 4;     // + i * 16 * N + j * 16
  const size_t rstep = (size_t)16 * p.N;
  const float* bcol = HB ? p.bias + n0 + wc * (NT * 16) + q * 4 : nullptr;
  const float* res = RES ? (const float*)p.aux + o0 : nullptr;
  float* out0 = (float*)p.out0 + o0;
  bf16_t* out1 = DUAL ? (bf16_t*)p.out1 + o0 : nullptr;
  auto load = [&](auto gc, u32x4 (&r)[GL > 0 ? GL : 1]) {
    constexpr int g = decltype(gc)::value;
    if constexpr (HB) { gload16(r[0], bcol + Plan::col(g, 0) * 16); gload16(r[1], bcol + Plan::col(g, 1) * 16); }
    if constexpr (RES) {
      static_for([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        gload16(r[(HB ? 2 : 0) + u], res + Plan::row(g, u) * rstep + Plan::col(g, u) * 16);
      }, std::make_integer_sequence<int, 4>{});
    }
  };
  auto finish = [&](auto gc, const u32x4 (&r)[GL > 0 ? GL : 1]) {
    constexpr int g = decltype(gc)::value;
    float h0, h1, h2, h3;
    static_for([&](auto uc) {
      constexpr int u = decltype(uc)::value, j = Plan::col(g, u), i = Plan::row(g, u);
      float x0 = acc[i][j][0], x1 = acc[i][j][1], x2 = acc[i][j][2], x3 = acc[i][j][3];
      if constexpr (HB) { const u32x4 t = r[u & 1]; x0 += as_f(t.x); x1 += as_f(t.y); x2 += as_f(t.z); x3 += as_f(t.w); }
      // the residual stays in the row layout (as loaded, coalesced); the accumulator quad goes there too, and the sum is formed and stored there
      u32x4 a; a.x = __builtin_bit_cast(unsigned, x0); a.y = __builtin_bit_cast(unsigned, x1); a.z = __builtin_bit_cast(unsigned, x2); a.w = __builtin_bit_cast(unsigned, x3);
      a = lane_perm(pa, a);
      x0 = as_f(a.x); x1 = as_f(a.y); x2 = as_f(a.z); x3 = as_f(a.w);
      if constexpr (RES) { const u32x4 t = r[(HB ? 2 : 0) + u]; x0 += as_f(t.x); x1 += as_f(t.y); x2 += as_f(t.z); x3 += as_f(t.w); }
      if constexpr ((u & 1) == 0) { h0 = x0; h1 = x1; h2 = x2; h3 = x3; }
      else {                                                // the two halves of a line leave back to back
        constexpr int j0 = Plan::col(g, u - 1), i0 = Plan::row(g, u - 1);
        __builtin_amdgcn_sched_barrier(0);
        gst(out0 + i0 * rstep + j0 * 16, h0, h1, h2, h3, p.nt_store);
        gst(out0 + i * rstep + j * 16, x0, x1, x2, x3, p.nt_store);
        if constexpr (DUAL) {
          u32x2 w; w.x = pack2bf(h0, h1); w.y = pack2bf(h2, h3); gst(reinterpret_cast<u32x2*>(out1 + i0 * rstep + j0 * 16), w, p.nt_store);
          w.x = pack2bf(x0, x1); w.y = pack2bf(x2, x3); gst(reinterpret_cast<u32x2*>(out1 + i * rstep + j * 16), w, p.nt_store);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }, std::make_integer_sequence<int, 4>{});
  };
  run_groups<NG, GL, GS>(load, finish);
}

// G8_EPI_RESID_LN: x_out = acc + bias + residual (fp32, as BIAS_RESID) AND h = LayerNorm(x_out) in bf16 -- the LayerNorm that follows the FFN-2 GEMM of a
// pre-LN layer (hf modeling_esm.py:482-521 -> the next layer's :429, or emb_layer_norm_after), which used to be a kernel of its own reading x_out back (335 MB per launch at
// the cfg-2 shape, 92 us, no matrix work).  A row of N columns is spread over N / 320 work-groups x 2 wave columns; nobody holds it whole.  So:
//   A  the BIAS_RESID epilogue as it stands (loads two groups ahead, 16-byte fp32 stores), with the finished sums written BACK into the accumulator registers
//      (row layout) and a running row sum per row block;
//   B  per lane: mean and M2 = sum (s - mean)^2 over its 40 values of each of its 4 rows, in registers; the four lanes of a row (a quad) combine theirs by
//      Chan's rule (DPP quad_perm): every lane then holds (mean, M2) of its wave block's 160 columns;
//   C  lanes sq == 0 publish them -- ln_part[row][slot], slot = column tile x 2 + wave column, in UNCACHED device memory, as ONE tagged 16-byte store;
//   D  lane sq polls entry sq (and sq + 4) of its rows until all of them carry this launch's tag (the other wave blocks run the same tile at the same time:
//      the column tiles of one row panel are handed out next to each other and run at the same time), bounded: a wait that runs out sets the sticky LN_ERR
//      flag of the sched workspace and the wave writes NaN for its rows (h, mean, rstd) -- and every later wait of the launch gives up at its first miss;
//   E  the quad combines again: every lane has the row's statistics over all N columns;
//   F  h = (s - mean) rstd gamma + beta from the registers, two column tiles at a time (gamma / beta two groups ahead like every epilogue operand),
//      lane pairs swapping halves so that a lane stores 16 contiguous bytes; mean / rstd leave from slot 0.
// The statistics are exact two-pass ones per wave block (the values are in registers) and Chan's combination is exact in exact arithmetic: the result
// agrees with k_layernorm_fwd's two-pass form to fp32 rounding.  Progress: a wave waits for the waves that run the other column tiles of its row panel -- with
// the static tile list (this epilogue never runs with drawn tickets: gemm_nt8.hip) those are work-groups of the same launch slot.  If one of them has no CU yet
// (a co-resident kernel holds it) its neighbours wait -- once per tile of theirs -- until a work-group that has finished makes room, or until the bound.
// The bound ends it either way: NaN + LN_ERR, never a hang and never a stale statistic.
// 16-byte load / store that miss every cache on their way (system scope): the partial statistics of the wave blocks of a row
template <int OFF> __device__ __forceinline__ void gload16_uc(u32x4& d, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1" : "=v"(d) : "v"(ptr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void gst16_uc(const void* ptr, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc0 sc1\n\ts_nop 1" ::"v"(ptr), "v"(v), "n"(OFF) : "memory");
}
// polls MT x NS tagged entries (row block i: 2 KB apart; second set: 4 entries on; `pl` two row blocks in) until every lane of the wave sees its own carry
// the tag.  Bounded: after `poll_max` polls without them -- or as soon as ANOTHER wave of the launch has given up (the sticky flag is read in the slow path
// only) -- the wave sets the flag and returns false: the caller then writes NaN for its rows (a starved launch ends quickly and loudly, never plausibly wrong).
template <int MT, int NS>
__device__ __forceinline__ bool poll_entries(const u32x4* pl, unsigned tag, u32x4 (&e)[NS * MT], unsigned* err, int poll_max, int lane) {
  int spins = 0;
  for (;;) {
    static_for([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      gload16_uc<(i % MT) * 2048 - 4096 + (i / MT) * 64>(e[i], pl);
    }, std::make_integer_sequence<int, NS * MT>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NS * MT; ++i) { asm volatile("" : "+v"(e[i])); ok = ok && e[i].x == tag && e[i].w == ~tag; }
    if (__all(ok)) return true;
    const unsigned gave_up = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gave_up != 0u || ++spins > poll_max) {
      if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}
__device__ __forceinline__ float dpp_quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)); }
__device__ __forceinline__ float dpp_quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)); }
// (mean, M2) of two equally large sets of n values each -> of their union
__device__ __forceinline__ void chan_merge(float& mean, float& m2, float mo, float m2o, float n) {
  const float d = mo - mean;
  mean = 0.5f * (mean + mo);
  m2 = m2 + m2o + d * d * (0.5f * n);
}
template <bool HB, int MT, int NT>
__device__ __forceinline__ void epilogue_resid_ln(const GemmArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wr, int wc, int lane, const unsigned tag) {
  static_assert(NT % 2 == 0, "whole lines per wave block");
  using Plan = LinePlan<MT, NT, false>;
  static_assert(!Plan::LONE);
  constexpr int NG = Plan::NG, GL = (HB ? 2 : 0) + 4, GS = 4;
  const int q = lane >> 4;
  const int sr = lane >> 2, sq = lane & 3;
  const int pa = to_rows_addr(lane);
  const int row_w = m0 + wr * (MT * 16);                    // first row of the wave block
  const int col_w = n0 + wc * (NT * 16);                    // first column
  const size_t o0 = (size_t)(row_w + sr) * p.N + col_w + sq * 4;
  const size_t rstep = (size_t)16 * p.N;
  const float* bcol = HB ? p.bias + col_w + q * 4 : nullptr;
  const float* res = (const float*)p.aux + o0;
  float* out0 = (float*)p.out0 + o0;
  float rsum[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) rsum[i] = 0.f;
  // ---- A
  {
    auto load = [&](auto gc, u32x4 (&r)[GL]) {
      constexpr int g = decltype(gc)::value;
      if constexpr (HB) { gload16(r[0], bcol + Plan::col(g, 0) * 16); gload16(r[1], bcol + Plan::col(g, 1) * 16); }
      static_for([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        gload16(r[(HB ? 2 : 0) + u], res + Plan::row(g, u) * rstep + Plan::col(g, u) * 16);
      }, std::make_integer_sequence<int, 4>{});
    };
    auto finish = [&](auto gc, const u32x4 (&r)[GL]) {
      constexpr int g = decltype(gc)::value;
      float h0, h1, h2, h3;
      static_for([&](auto uc) {
        constexpr int u = decltype(uc)::value, j = Plan::col(g, u), i = Plan::row(g, u);
        float x0 = acc[i][j][0], x1 = acc[i][j][1], x2 = acc[i][j][2], x3 = acc[i][j][3];
        if constexpr (HB) { const u32x4 t = r[u & 1]; x0 += as_f(t.x); x1 += as_f(t.y); x2 += as_f(t.z); x3 += as_f(t.w); }
        u32x4 a; a.x = __builtin_bit_cast(unsigned, x0); a.y = __builtin_bit_cast(unsigned, x1); a.z = __builtin_bit_cast(unsigned, x2); a.w = __builtin_bit_cast(unsigned, x3);
        a = lane_perm(pa, a);
        x0 = as_f(a.x); x1 = as_f(a.y); x2 = as_f(a.z); x3 = as_f(a.w);
        { const u32x4 t = r[(HB ? 2 : 0) + u]; x0 += as_f(t.x); x1 += as_f(t.y); x2 += as_f(t.z); x3 += as_f(t.w); }
        acc[i][j][0] = x0; acc[i][j][1] = x1; acc[i][j][2] = x2; acc[i][j][3] = x3;      // row layout from here on: row sr of block i, columns j*16 + sq*4 ..
        rsum[i] += (x0 + x1) + (x2 + x3);
        if constexpr ((u & 1) == 0) { h0 = x0; h1 = x1; h2 = x2; h3 = x3; }
        else {
          constexpr int j0 = Plan::col(g, u - 1), i0 = Plan::row(g, u - 1);
          __builtin_amdgcn_sched_barrier(0);
          gst(out0 + i0 * rstep + j0 * 16, h0, h1, h2, h3, 0);
          gst(out0 + i * rstep + j * 16, x0, x1, x2, x3, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }, std::make_integer_sequence<int, 4>{});
    };
    run_groups<NG, GL, GS>(load, finish);
  }
  // ---- B
  constexpr float NL = (float)(NT * 4);                     // values per lane and row
  float mean[MT], m2[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    mean[i] = rsum[i] * (1.0f / NL);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = acc[i][j][e] - mean[i]; s2 += t * t; }
    m2[i] = s2;
    chan_merge(mean[i], m2[i], dpp_quad_xor1(mean[i]), dpp_quad_xor1(m2[i]), NL);
    chan_merge(mean[i], m2[i], dpp_quad_xor2(mean[i]), dpp_quad_xor2(m2[i]), 2.0f * NL);
  }
  // ---- C  (ln_part is UNCACHED device memory, gemm_nt8.hip: plain stores and loads reach it, coherent across the XCDs' L2s; an agent-scope release /
  // acquire on ordinary memory writes back and invalidates a whole L2 per fence on this part -- measured: the launch 360 us longer).  An entry is 16 bytes,
  // {tag, mean, M2, ~tag} with tag = this launch's number, written by ONE 16-byte store: a reader that finds both ends current has the middle too, entries of
  // older launches never match, and nothing has to be cleared or counted (the first form -- partials, a wait for them to land, an arrival counter, polls of
  // the counter, then the loads -- was three round trips to uncached memory per tile and a memset per launch; this is the store and one or two polls).
  const int slot = (n0 / (2 * NT * 16)) * 2 + wc;          // column tile x 2 + wave column (a tile is two wave columns wide)
  const u32x4* ent = reinterpret_cast<const u32x4*>(p.ln_part) + (size_t)(row_w + sr) * 8;
  if (sq == 0) {
    static_for([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      u32x4 e; e.x = tag; e.y = __builtin_bit_cast(unsigned, mean[i]); e.z = __builtin_bit_cast(unsigned, m2[i]); e.w = ~tag;
      gst16_uc<i * 2048 - 4096>(ent + slot + 256, e);      // (13-bit signed immediates: the base sits two row blocks in)
    }, std::make_integer_sequence<int, MT>{});
  }
  // ---- D / E  every lane polls the entries it needs (slot sq -- and sq + 4 with four column tiles -- of its four rows) until all of them carry the tag
  const float nw = 4.0f * NL;                               // columns of a wave block
  float rstd[MT];
  {
    const int s0 = p.ln_slots >= 4 ? sq : (sq & 1);
    float cnt = nw;
    unsigned* const errp = p.sched + SW_LN_ERR;
    bool got;
    if (p.ln_slots == 8) {                                    // four column tiles: entries sq and sq + 4, all eight loads of a poll in flight together
      u32x4 e[2 * MT];
      got = poll_entries<MT, 2>(ent + s0 + 256, tag, e, errp, p.ln_poll_max, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        mean[i] = as_f(e[i].y); m2[i] = as_f(e[i].z);
        chan_merge(mean[i], m2[i], as_f(e[MT + i].y), as_f(e[MT + i].z), cnt);
      }
      cnt *= 2.0f;
    } else {
      u32x4 e[MT];
      got = poll_entries<MT, 1>(ent + s0 + 256, tag, e, errp, p.ln_poll_max, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i) { mean[i] = as_f(e[i].y); m2[i] = as_f(e[i].z); }
    }
    // a wait that ran out: the statistics of these rows are unknown -> NaN (h, mean and rstd of the wave block's rows), never a stale entry's numbers
    const float poison = got ? 0.0f : __builtin_nanf("");
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float mu = mean[i], mm = m2[i], c = cnt;
      chan_merge(mu, mm, dpp_quad_xor1(mu), dpp_quad_xor1(mm), c); c *= 2.0f;
      if (p.ln_slots >= 4) { chan_merge(mu, mm, dpp_quad_xor2(mu), dpp_quad_xor2(mm), c); c *= 2.0f; }
      mean[i] = mu + poison;
      rstd[i] = rsqrtf(mm / (float)p.N + p.q_scale) + poison;
    }
  }
  if (slot == 0 && sq == 0 && p.out2) {
    float* st = (float*)p.out2 + row_w + sr;
#pragma unroll
    for (int i = 0; i < MT; ++i) { st[i * 16] = mean[i]; st[(size_t)p.M + i * 16] = rstd[i]; }
  }
  // ---- F
  {
    constexpr int NP = NT / 2, GLF = 4, GSF = MT;
    const float* gcol = p.cos + col_w + sq * 4;
    const float* bcolf = p.sin + col_w + sq * 4;
    const bool odd = (sq & 1) != 0;
    const bf16_t* hbase = (const bf16_t*)p.out1 + (size_t)row_w * p.N + col_w;
    const unsigned voffh = (unsigned)(sr * p.N + (odd ? 16 + (sq - 1) * 4 : sq * 4)) * 2u;
    auto load = [&](auto gc, u32x4 (&r)[GLF]) {
      constexpr int jp = decltype(gc)::value;
      gload16(r[0], gcol + (2 * jp) * 16); gload16(r[1], gcol + (2 * jp + 1) * 16);
      gload16(r[2], bcolf + (2 * jp) * 16); gload16(r[3], bcolf + (2 * jp + 1) * 16);
    };
    auto finish = [&](auto gc, const u32x4 (&r)[GLF]) {
      constexpr int jp = decltype(gc)::value;
      static_for([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const float mu = mean[i], rs = rstd[i];
        const f32x4 sa = acc[i][2 * jp], sb = acc[i][2 * jp + 1];
        const float a0 = (sa[0] - mu) * rs * as_f(r[0].x) + as_f(r[2].x), a1 = (sa[1] - mu) * rs * as_f(r[0].y) + as_f(r[2].y);
        const float a2 = (sa[2] - mu) * rs * as_f(r[0].z) + as_f(r[2].z), a3 = (sa[3] - mu) * rs * as_f(r[0].w) + as_f(r[2].w);
        const float b0 = (sb[0] - mu) * rs * as_f(r[1].x) + as_f(r[3].x), b1 = (sb[1] - mu) * rs * as_f(r[1].y) + as_f(r[3].y);
        const float b2 = (sb[2] - mu) * rs * as_f(r[1].z) + as_f(r[3].z), b3 = (sb[3] - mu) * rs * as_f(r[1].w) + as_f(r[3].w);
        const unsigned ax = pack2bf(a0, a1), ay = pack2bf(a2, a3), bx = pack2bf(b0, b1), by = pack2bf(b2, b3);
        // even lane: its four columns of tile 2jp, then the odd neighbour's; odd lane: the even neighbour's four columns of tile 2jp+1, then its own
        const unsigned sx = odd ? ax : bx, sy = odd ? ay : by;
        const unsigned rx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xF, 0xF, false);
        const unsigned ry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xF, 0xF, false);
        u32x4 o;
        o.x = odd ? rx : ax; o.y = odd ? ry : ay; o.z = odd ? bx : rx; o.w = odd ? by : ry;
        gst16_s<jp * 64>(hbase + (size_t)i * 16 * p.N, voffh, o);
      }, std::make_integer_sequence<int, MT>{});
    };
    run_groups<NP, GLF, GSF>(load, finish);
  }
}

// QKV / RoPE, head_dim 32 (natural map; a head = tiles 2h, 2h+1 of the wave's column block, the rotation partner of a column is the same
// register of the other tile).  Group = one head.  The RoPE table rows of the wave's MT row groups are fetched once, before any store.
template <bool HB, int MT, int NT>
__device__ __forceinline__ void epilogue_rope32(const GemmArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wr, int wc, int lane) {
  constexpr int HD = 32, HALF = 16, NG = NT / 2, GL = HB ? 2 : 0, GS = MT;
  const int c = lane & 15, q = lane >> 4;
  const int mrow0 = m0 + wr * (MT * 16) + c, ncol0 = n0 + wc * (NT * 16);
  const int dm = p.H * HD;
  const int sec = __builtin_amdgcn_readfirstlane(ncol0 / dm);
  const int head0 = (ncol0 - sec * dm) / HD;
  const float sc = sec == 0 ? p.q_scale : 1.0f;
  const float* bbase = HB ? p.bias + ncol0 : nullptr;       // + q * 4 floats per lane (boff), + h * 32 (+ 16) floats as an immediate
  const unsigned boff = (unsigned)q * 16u;
  float4 cs[MT], sn[MT];
  // row layout (stores): the head-major output is addressed as a wave-uniform base -- (batch element of the wave block's first row, head0) -- plus a
  // 32-bit per-lane byte offset per row block (a wave block of MT*16 rows spans a few batch elements at most), head h adds h * L * HD to the base
  const int row0 = m0 + wr * (MT * 16);
  const int b0 = __builtin_amdgcn_readfirstlane(row0 / p.L);
  const bf16_t* dst = (const bf16_t*)(sec == 0 ? p.out0 : (sec == 1 ? p.out1 : p.out2)) + ((size_t)b0 * p.H + head0) * p.L * HD;
  unsigned roff[MT];
  const int sr = lane >> 2, sq = lane & 3, pa = to_rows_addr(lane);
  const bool odd = (sq & 1) != 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gs = row0 + i * 16 + sr;
    const int bs = gs / p.L, ls = gs - bs * p.L;
    // one 16-byte store per lane: neighbouring lanes swap halves below, the even lane of a pair then holds columns sq*4 .. sq*4+7 of the head's first
    // half, the odd lane columns 16 + (sq-1)*4 .. +7 of its second half (two 8-byte stores per lane ran the store path at 4.2 TB/s)
    roff[i] = (unsigned)((((bs - b0) * p.H) * p.L + ls) * HD + ((sq & 1) ? HALF + (sq - 1) * 4 : sq * 4)) * 2u;
    const int gm = mrow0 + i * 16;                          // accumulator layout (rotation arithmetic): row c
    const int l = gm % p.L;
    if (sec < 2) {
      cs[i] = *reinterpret_cast<const float4*>(p.cos + (size_t)l * HALF + q * 4);
      sn[i] = *reinterpret_cast<const float4*>(p.sin + (size_t)l * HALF + q * 4);
    } else { cs[i] = make_float4(1.f, 1.f, 1.f, 1.f); sn[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(cs[i].x), "+v"(cs[i].y), "+v"(cs[i].z), "+v"(cs[i].w), "+v"(sn[i].x), "+v"(sn[i].y), "+v"(sn[i].z), "+v"(sn[i].w));
  const size_t hstep = (size_t)p.L * HD;                    // next head of the same token
  auto load = [&](auto gc, u32x4 (&r)[GL > 0 ? GL : 1]) {
    constexpr int h = decltype(gc)::value;
    if constexpr (HB) { gload16_s<h * 128>(r[0], bbase, boff); gload16_s<h * 128 + 64>(r[1], bbase, boff); }
  };
  auto finish = [&](auto gc, const u32x4 (&r)[GL > 0 ? GL : 1]) {
    constexpr int h = decltype(gc)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float lo[4] = {acc[i][2 * h][0], acc[i][2 * h][1], acc[i][2 * h][2], acc[i][2 * h][3]};
      float hi[4] = {acc[i][2 * h + 1][0], acc[i][2 * h + 1][1], acc[i][2 * h + 1][2], acc[i][2 * h + 1][3]};
      if constexpr (HB) {
        lo[0] += as_f(r[0].x); lo[1] += as_f(r[0].y); lo[2] += as_f(r[0].z); lo[3] += as_f(r[0].w);
        hi[0] += as_f(r[1].x); hi[1] += as_f(r[1].y); hi[2] += as_f(r[1].z); hi[3] += as_f(r[1].w);
      }
      const float cv[4] = {cs[i].x, cs[i].y, cs[i].z, cs[i].w}, sv[4] = {sn[i].x, sn[i].y, sn[i].z, sn[i].w};
      float ol[4], oh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {                         // v: sec 2 -> cos = 1, sin = 0, scale = 1: the value itself
        ol[e] = (lo[e] * sc) * cv[e] - (hi[e] * sc) * sv[e];
        oh[e] = (hi[e] * sc) * cv[e] + (lo[e] * sc) * sv[e];
      }
      const bf16_t* d = dst + h * hstep;                    // wave-uniform
      u32x4 w; w.x = pack2bf(ol[0], ol[1]); w.y = pack2bf(ol[2], ol[3]); w.z = pack2bf(oh[0], oh[1]); w.w = pack2bf(oh[2], oh[3]);
      w = lane_perm(pa, w);                                 // row layout: four neighbouring lanes hold the 32 + 32 bytes of one (token, head) row
      const unsigned sx = odd ? w.x : w.z, sy = odd ? w.y : w.w;                                        // what the neighbour stores
      const unsigned rx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xF, 0xF, false);      // quad_perm [1, 0, 3, 2]
      const unsigned ry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xF, 0xF, false);
      u32x4 o;
      o.x = odd ? rx : w.x; o.y = odd ? ry : w.y; o.z = odd ? w.z : rx; o.w = odd ? w.w : ry;
      gst16_s<0, true>(d, roff[i], o);
    }
  };
  run_groups<NG, GL, GS>(load, finish);
}

// QKV / RoPE, head_dim 64 (BERT-base, ESM-2-650M; natural map, wave column block = whole heads of four tiles: tiles 4h, 4h+1 hold the first half of
// the head, 4h+2, 4h+3 the second, the rotation partner of a column is the same register of tile j ^ 2).  Round 5: this shape used to run the
// generic tail (bias into the accumulators, then rope_store_direct: per-element divisions, table loads waited for one by one, 8-byte stores) --
// 584 us for the 650M QKV launch of the cfg-5 shape, a fifth of that step.  Group = (head, row block): bias, cos and sin pieces of the group run
// two groups ahead of the stores; after the lane transpose neighbouring lanes swap halves, so a (token, head) row of 128 bytes leaves as
// 2 x 4 lanes x 16 bytes.
template <bool HB, int MT, int NT>
__device__ __forceinline__ void epilogue_rope64(const GemmArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wr, int wc, int lane) {
  static_assert(NT % 4 == 0, "whole 64-wide heads per wave column block");
  constexpr int HD = 64, HALF = 32, NHB = NT / 4, NG = NHB * MT, GL = (HB ? 4 : 0) + 4, GS = 2;
  const int c = lane & 15, q = lane >> 4;
  const int mrow0 = m0 + wr * (MT * 16) + c, ncol0 = n0 + wc * (NT * 16);
  const int dm = p.H * HD;
  const int sec = __builtin_amdgcn_readfirstlane(ncol0 / dm);
  const int head0 = (ncol0 - sec * dm) / HD;
  const float sc = sec == 0 ? p.q_scale : 1.0f;
  const bool rot = sec < 2;                                 // v: no rotation
  const float* bbase = HB ? p.bias + ncol0 : nullptr;
  const unsigned boff = (unsigned)q * 16u;
  const int row0 = m0 + wr * (MT * 16);
  const int b0 = __builtin_amdgcn_readfirstlane(row0 / p.L);
  const bf16_t* dst = (const bf16_t*)(sec == 0 ? p.out0 : (sec == 1 ? p.out1 : p.out2)) + ((size_t)b0 * p.H + head0) * p.L * HD;
  const int sr = lane >> 2, sq = lane & 3, pa = to_rows_addr(lane);
  const bool odd = (sq & 1) != 0;
  unsigned roff[MT];                                        // row layout: byte offset of the lane's 16-byte piece inside the first 32 columns of its (token, head) row
  unsigned toff[MT];                                        // accumulator layout: byte offset of the lane's 4 table entries (row l of the cos / sin tables)
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gs = row0 + i * 16 + sr;
    const int bs = gs / p.L, ls = gs - bs * p.L;
    roff[i] = (unsigned)((((bs - b0) * p.H) * p.L + ls) * HD + (odd ? 16 + (sq - 1) * 4 : sq * 4)) * 2u;
    toff[i] = (unsigned)(((mrow0 + i * 16) % p.L) * HALF + q * 4) * 4u;
  }
  const size_t hstep = (size_t)p.L * HD;
  auto load = [&](auto gc, u32x4 (&r)[GL]) {
    constexpr int g = decltype(gc)::value, hh = g / MT, i = g % MT;
    gload16_s<0>(r[0], p.cos, toff[i]); gload16_s<64>(r[1], p.cos, toff[i]);
    gload16_s<0>(r[2], p.sin, toff[i]); gload16_s<64>(r[3], p.sin, toff[i]);
    if constexpr (HB) {
      gload16_s<hh * 256>(r[4], bbase, boff); gload16_s<hh * 256 + 64>(r[5], bbase, boff);
      gload16_s<hh * 256 + 128>(r[6], bbase, boff); gload16_s<hh * 256 + 192>(r[7], bbase, boff);
    }
  };
  auto finish = [&](auto gc, const u32x4 (&r)[GL]) {
    constexpr int g = decltype(gc)::value, hh = g / MT, i = g % MT, j0 = 4 * hh;
    float v[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[t][0] = acc[i][j0 + t][0]; v[t][1] = acc[i][j0 + t][1]; v[t][2] = acc[i][j0 + t][2]; v[t][3] = acc[i][j0 + t][3];
      if constexpr (HB) { v[t][0] += as_f(r[4 + t].x); v[t][1] += as_f(r[4 + t].y); v[t][2] += as_f(r[4 + t].z); v[t][3] += as_f(r[4 + t].w); }
    }
    u32x4 w[2];                                             // [half of the head]: {tile 2 half, tile 2 half + 1} packed
#pragma unroll
    for (int t = 0; t < 2; ++t) {                           // column tile inside a half: table entries 16 t + q * 4 ..
      const float cv[4] = {as_f(r[t].x), as_f(r[t].y), as_f(r[t].z), as_f(r[t].w)}, sv[4] = {as_f(r[2 + t].x), as_f(r[2 + t].y), as_f(r[2 + t].z), as_f(r[2 + t].w)};
      float ol[4], oh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = v[t][e] * sc, b = v[2 + t][e] * sc;
        ol[e] = rot ? a * cv[e] - b * sv[e] : a;
        oh[e] = rot ? b * cv[e] + a * sv[e] : b;
      }
      if (t == 0) { w[0].x = pack2bf(ol[0], ol[1]); w[0].y = pack2bf(ol[2], ol[3]); w[1].x = pack2bf(oh[0], oh[1]); w[1].y = pack2bf(oh[2], oh[3]); }
      else { w[0].z = pack2bf(ol[0], ol[1]); w[0].w = pack2bf(ol[2], ol[3]); w[1].z = pack2bf(oh[0], oh[1]); w[1].w = pack2bf(oh[2], oh[3]); }
    }
    const bf16_t* d = dst + hh * hstep;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const u32x4 t = lane_perm(pa, w[hf]);                 // row layout: lane (sr, sq) holds columns sq*4.. of tile 2 hf (x, y) and of tile 2 hf + 1 (z, w)
      const unsigned sx = odd ? t.x : t.z, sy = odd ? t.y : t.w;
      const unsigned rx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xF, 0xF, false);      // quad_perm [1, 0, 3, 2]
      const unsigned ry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xF, 0xF, false);
      u32x4 o;
      o.x = odd ? rx : t.x; o.y = odd ? ry : t.y; o.z = odd ? t.z : rx; o.w = odd ? t.w : ry;
      if (hf == 0) gst16_s<0, true>(d, roff[i], o); else gst16_s<HALF * 2, true>(d, roff[i], o);
    }
  };
  run_groups<NG, GL, GS>(load, finish);
}

// vector-memory stores one wave issues per tile (for the store-tolerant wait that follows the epilogue)
template <int EPI, bool DUAL, int MT, int NT> constexpr int epilogue_stores() {
  if (EPI == G8_EPI_RESID_LN) return MT * NT / 2;           // the bf16 stores behind the epilogue's own full wait (everything older has landed by then)
  if (EPI == ONEPROT_EPI_BIAS_GELU && DUAL) {               // out0 in 16-byte pieces; the one-byte out1 in 16-byte pieces for paired units, 8-byte for the lone half line
    const int lone_units = (((NT / 2) & 1) != 0) ? MT : 0;
    return MT * NT / 2 + lone_units + (MT * NT / 2 - lone_units) / 2;
  }
  if (EPI == ONEPROT_EPI_BF16 || EPI == ONEPROT_EPI_GELU_BWD || EPI == ONEPROT_EPI_BIAS_GELU) return MT * NT / 2;
  if (EPI == ONEPROT_EPI_QKV_ROPE) return MT * NT / 2;      // head_dim 32: one 16-byte store per head and row block (head_dim 64 issues twice as many: the smaller count is the safe one)
  return MT * NT * (DUAL ? 2 : 1);      // F32, BIAS_RESID (fp32 + optional bf16 copy)
}

}  // namespace g8
