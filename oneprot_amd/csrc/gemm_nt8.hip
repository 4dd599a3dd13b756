// bf16 MFMA GEMM, C[M,N] = A[M,K] * B[N,K]^T (+ fused epilogue) -- the 8-PHASE form.  gfx950 only.
//
// One persistent 512-thread work-group per CU.  Its eight waves are two groups of four (one wave of each group per SIMD) that run the SAME
// instruction stream half a phase apart: while group 0 issues the MFMAs of phase p, group 1 reads the LDS fragments of phase p and issues its
// share of the LDS-DMA prefetch, and the other way round one barrier later.  The matrix pipe of a SIMD therefore always has exactly one wave
// feeding it, and that wave's LDS reads and DMA issue were done while its partner computed (cdna_hip_programming.md section 5, "The 256^2
// 8-phase template"; MI355X_MICROARCH.md "Two waves per SIMD").
//
//   * a K-tile is 64 deep (128-byte LDS rows, every LDS-DMA piece covers whole 128-byte lines); the wave tile is split in 2 x 2 quadrants and
//     a PHASE is one quadrant x K = 64 (16 or 20 MFMAs): { ds_read the quadrant's fragments | issue one unit of LDS-DMA -> s_barrier ->
//     lgkmcnt(0) -> MFMAs -> s_barrier }.  Four phases per K-tile, the loop body is two K-tiles = 8 phases (LDS buffers are compile-time);
//   * the operand slice of a K-tile is staged as four UNITS (two row halves of the A tile, two column halves of the W tile), laid out so that
//     a unit holds exactly the rows ONE phase reads: the "major" operand X (the one with the larger fragment set per quadrant) is read once per
//     unit and held in registers for two phases, the "minor" operand Y keeps both its fragment sets.  Quadrant order (X0,Y0) (X0,Y1) (X1,Y1)
//     (X1,Y0): reads are X0+Y0 | Y1 | X1 | none, so a unit's LDS is dead one phase after it was needed and can be refilled for the K-tile after
//     next: the DMA stream runs 7 units (1.75 K-tiles, up to 126 KB) ahead of the reads behind ONE counted vmcnt per K-tile;
//   * the unit stream is continuous across the output tiles of the work-group: the first K-tiles of the next tile arrive while the epilogue
//     of the current one runs, and the epilogue works straight from the accumulators (gemm_epi8.h: lane transpose by ds_bpermute, whole
//     128-byte lines per pair of stores), never touching LDS memory;
//   * hazards (slot = the interval between two consecutive barriers; group 0 reads phase p in slot 2p and computes in 2p+1, group 1 one slot
//     later): a DMA may overwrite what phase p read from phase p+2 on -- or from p+1 on when the reads were retired (counted lgkmcnt) before
//     the first barrier of phase p, which is how X0 is recycled; data waited for (counted vmcnt, before a phase's first barrier) in phase p is
//     read from phase p+1 on.
//
// Configurations:  256 x 256 (waves 2 x 4, wave tile 128 x 64, X = A)  -- N % 256 == 0; also with MERGED phases (two phases of two quadrants per
//                                                                          K-tile, `ktile2`; forced shape 42: measured equal, kept as a tested variant)
//                  256 x 320 (waves 4 x 2, wave tile  64 x 160, X = W) -- every GEMM of a d = 640 / 1280 / 320 encoder (N = d, 3d, 4d); ONE
//                                                                          activation fragment set, its first half re-read from a 3-slot ring (Y3)
// Whole tiles only, K % 128 == 0; everything else takes the per-tile kernels of gemm_nt.hip.
#include "gemm_epi8.h"
#include <atomic>

namespace g8 {

template <int WM_, int WN_, int MTW_, int NTW_, bool XA_, bool Y3_, bool M2_ = false> struct Cfg {
  static constexpr int WM = WM_, WN = WN_, MTW = MTW_, NTW = NTW_;
  static constexpr bool XA = XA_;                           // major operand: A (activations) or W
  static constexpr bool Y3 = Y3_;                           // ONE register set for the minor operand: Y0 is read again in phase 4 and therefore lives in a 3-slot ring
  static constexpr bool M2 = M2_;                           // merged phases: two phases of two quadrants per K-tile instead of four of one (both minor-operand sets held; no Y0 ring)
  static_assert(!(M2 && Y3), "merged phases hold both minor sets");
  static constexpr int MH = MTW / 2, NH = NTW / 2;          // 16-row tiles per quadrant side
  static constexpr int BM = WM * MTW * 16, BN = WN * NTW * 16;
  static constexpr int A_ROWS = BM / 2, B_ROWS = BN / 2;    // rows per unit
  static constexpr int A_UNIT = A_ROWS * 128, B_UNIT = B_ROWS * 128;
  static constexpr int A_PIECES = A_ROWS / 8, B_PIECES = B_ROWS / 8;       // 1 KiB LDS-DMA pieces (8 rows x 128 B) per unit
  static constexpr int A_IPW = (A_PIECES + 7) / 8, B_IPW = (B_PIECES + 7) / 8;      // instructions per wave (the last one only in waves < *_REM)
  static constexpr int A_REM = A_PIECES % 8, B_REM = B_PIECES % 8;                // 0: every wave issues *_IPW
  static constexpr int Y_UNIT = XA ? B_UNIT : A_UNIT, X_UNIT = XA ? A_UNIT : B_UNIT;
  // LDS: two buffers of [X0 | X1 | Y1 (| Y0)], and with Y3 the three Y0 slots behind them
  static constexpr int BUF = 2 * X_UNIT + Y_UNIT + (Y3 ? 0 : Y_UNIT);
  static constexpr int OFF_X0 = 0, OFF_X1 = X_UNIT, OFF_Y1 = 2 * X_UNIT, OFF_Y0 = 2 * X_UNIT + Y_UNIT;      // OFF_Y0: inside the buffer (!Y3)
  static constexpr int RING_Y0 = 2 * BUF;                                                                   // Y3: slot k at RING_Y0 + k * Y_UNIT
  static constexpr int LDS = 2 * BUF + (Y3 ? 3 * Y_UNIT : 0);
  static_assert(MTW % 2 == 0 && NTW % 2 == 0, "quadrants");
  static_assert(A_REM == 0 || A_REM == 4, "uneven units: group 0 carries the extra piece");
  static_assert(B_REM == 0 || B_REM == 4, "uneven units: group 0 carries the extra piece");
  static_assert(LDS <= 160 * 1024, "LDS");
  // LDS-DMA instructions of wave group g per unit
  static constexpr int a_cnt(int g) { return A_IPW - ((A_REM != 0 && g != 0) ? 1 : 0); }
  static constexpr int b_cnt(int g) { return B_IPW - ((B_REM != 0 && g != 0) ? 1 : 0); }
  static constexpr int x_cnt(int g) { return XA ? a_cnt(g) : b_cnt(g); }
  static constexpr int y_cnt(int g) { return XA ? b_cnt(g) : a_cnt(g); }
};

__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }
// Diagnostic build (-DG8_STAMP, tools/ab/g8_stamps.py): the first wave of each group of work-group 0 records s_memtime after every barrier of its
// first 32 K-tiles into the buffer passed as `out2` -- never compiled into the product library.
#ifdef G8_STAMP
#define STAMP(i) do { if (blockIdx.x == 0 && (wave & 3) == 0 && s < 32) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0) reinterpret_cast<unsigned long long*>(p.out2)[(grp * 32 + s) * 8 + (i)] = t_; } } while (0)
#define STAMP_E(i) do { if (blockIdx.x == 0 && (wave & 3) == 0 && q < 8) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0) reinterpret_cast<unsigned long long*>(p.out2)[512 + (grp * 8 + q) * 8 + (i)] = t_; } } while (0)
#define STAMP_M(ph, i) do { if (blockIdx.x == 0 && (wave & 3) == 0 && stamp_s < 32) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0) reinterpret_cast<unsigned long long*>(p.out2)[1024 + (grp * 32 + stamp_s) * 8 + (ph) * 2 + (i)] = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP_E(i) do { } while (0)
#define STAMP_M(ph, i) do { } while (0)
#endif
// One LDS-DMA piece: 64 lanes x 16 bytes from `sbase + voff` (wave-uniform base in SGPRs + 32-bit per-lane offset) to LDS bytes [lds_dst, +1024).
// Inline asm for two reasons: (i) the saddr + 32-bit-voffset form -- through the builtin hipcc keeps a zero-extended 64-bit offset PAIR per piece
// alive (16 VGPRs that the 160-accumulator configuration does not have: it spilled them and reloaded behind a vmcnt(0), draining the prefetch);
// (ii) none of these loads enters hipcc's own s_waitcnt bookkeeping, which is what the hand-counted vmcnt waits below assume.  M0 (the LDS
// destination) is written in the statement that uses it and restored afterwards (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(unsigned voff, const unsigned char* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// the lane id, recomputed where it is needed (two instructions) rather than kept over the K loop: a volatile statement, so that hipcc neither hoists it out
// of the tile loop nor keeps its result (the 160-accumulator configurations have no register for it: kept, it is spilled and its reload -- behind a
// vmcnt(0) -- drains the operand prefetch once per tile)
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// logical column (inside the wave's NTW*16-column block) held by LDS row slot `rho` of the wave's tile jt -- see gemm_epi.h
template <bool PAIR> __device__ __forceinline__ int slot_col(int jt, int rho) { return direct_nmap<PAIR>(jt, rho); }

template <class C, int EPI, bool HB, bool DUAL, bool DYN>
__global__ void __launch_bounds__(512, 2) k_gemm8(const GemmArgs p, const int g8n, const int dph_groups, const int dph_sleeps) {
  constexpr int MT = C::MTW, NT = C::NTW, MH = C::MH, NH = C::NH;
  constexpr bool XA = C::XA;
  constexpr bool PAIR = DirectMap<EPI>::PAIR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                               // waves w and w + 4 share a SIMD: one of each group
  const int wr = wave / C::WN, wc = wave % C::WN;
  const int nk = p.K >> 6;

  // ---- this work-group's tiles: XCD x (= blockIdx & 7 under round-robin dispatch; speed only).  csplit = 1: XCD x walks row panels x, x+8, ... with n
  // fastest.  csplit = c: the XCDs form 8 / c sets; set xg walks row panels xg, xg + 8/c, ... and each of its c XCDs takes 1/c of the column tiles, so an XCD
  // keeps 1/c of W in its L2 for the whole launch (N = 2560, K = 640: 3.3 MB of W next to four activation panels do not fit 4 MB) and reads every
  // activation panel c times less often than W used to be re-read.
  const int x = blockIdx.x & 7, w = blockIdx.x >> 3;
  const int cs = p.csplit, ng = 8 / cs, xg = x / cs, cg = x - xg * cs, tn_x = p.tiles_n / cs;
  const int panels_x = p.tiles_m > xg ? (p.tiles_m - xg + ng - 1) / ng : 0;
  const int Tx = panels_x * tn_x;                          // tiles of this XCD's share, numbered 0 .. Tx-1 ("tickets")
  // ticket -> tile: column tile fastest (an XCD works on a few row panels at a time: they and W stay in its L2)
  auto tile_origin = [&](int u, int& m0, int& n0) __attribute__((always_inline)) {
    const int pl = u / tn_x, tn = u - pl * tn_x;
    m0 = (pl * ng + xg) * C::BM; n0 = (cg * tn_x + tn) * C::BN;
  };
  // ---- which tiles.  Static (p.dyn == 0): tickets w, w + g8n, w + 2 g8n, ...  Dynamic: tickets are drawn from XCD x's queue head in the sched workspace
  // (sched_ws.h), two at the start and one per tile after that, always for the tile AFTER the next one (the operand stream runs into the next tile two
  // K-tiles before the current one ends, so the next tile must be known by then).  A work-group that finds the queue empty leaves at once: work-groups
  // that could not start with the others -- a co-resident kernel holds their CU -- cost nothing, the running ones share the tiles.
  // RESID_LN is never dynamic: its work-groups WAIT for the other column tiles of their row panel, and a work-group that holds drawn tickets ahead of its
  // current tile can be waited on, through such a ticket, by the very work-group it waits for (two work-groups that start together and draw alternately:
  // observed as a bounded wait that ran out, round 6).  With the static list the partners are work-groups w, w + 1 of the same launch slot.  It is a
  // forward-pass kernel: nothing is co-resident with it in a training step (the overlapped gradient all-reduce runs under the backward).
  // DYN is a template parameter: the static instantiation carries none of the draw / relay code in its K loop (0.5-1 % of a launch in round 6's first form)
  static_assert(!(DYN && EPI == G8_EPI_RESID_LN), "RESID_LN runs on static lists");
  constexpr bool dyn = DYN;
  int tk_cur, tk_nxt;
  unsigned ln_tag = 0;
  if constexpr (dyn) {
    unsigned* sm = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) {
      unsigned* head = p.sched + SW_HEAD(x);
      const unsigned t0 = sw_draw(head), t1 = sw_draw(head);
      sm[0] = t0 < 0x7fffffffu ? t0 : 0x7fffffffu; sm[1] = t1 < 0x7fffffffu ? t1 : 0x7fffffffu; sm[2] = sw_epoch(p.sched);
    }
    __syncthreads();
    tk_cur = __builtin_amdgcn_readfirstlane((int)sm[0]); tk_nxt = __builtin_amdgcn_readfirstlane((int)sm[1]);
    ln_tag = (unsigned)__builtin_amdgcn_readfirstlane((int)sm[2]) + 1u;
    __syncthreads();
  } else {
    tk_cur = w; tk_nxt = w + g8n;
    if (EPI == G8_EPI_RESID_LN) ln_tag = (unsigned)__builtin_amdgcn_readfirstlane((int)sw_epoch(p.sched)) + 1u;
  }
  const unsigned tiles_expected = (unsigned)(p.tiles_m * p.tiles_n);
  if (tk_cur >= Tx) {
    if (p.sched && tid == 0) sw_leave(p.sched, gridDim.x, 0u, tiles_expected);
    return;
  }
  // ---- de-phasing.  Every work-group runs the same number of equally long tiles, so without it all 256 CUs reach their epilogues together and
  // the output bursts (160-320 KB per CU) queue on the HBM write path while nothing computes.  Phase group (w mod G) starts (w mod G) * dph_sleeps
  // sleeps of 1024 cycles late; the offsets persist, and the stores of one group drain under the K loops of the others.
  if (dph_groups > 1) {
    const int late = (w % dph_groups) * dph_sleeps;
    for (int i = 0; i < late; ++i) __builtin_amdgcn_s_sleep(16);
  }

  // ---- per-lane LDS-DMA source offsets.  Piece P (8 rows x 128 B) of a unit is issued by wave P & 7 as its instruction P >> 3; the lane
  // (sr = lane >> 3, sc = lane & 7) fills LDS row P*8 + sr, chunk slot sc, with source chunk sc ^ ((row >> 1) & 7) (bank swizzle on the source
  // address, undone by the same XOR on the fragment read address).
  // These per-lane constants (and the fragment read addresses below) are RE-DERIVED from the lane id after every epilogue rather than kept: the
  // epilogue of the 160-accumulator configuration has no registers for them, and a spilled value is reloaded behind the whole store burst
  // (vector-memory operations retire in order), which would stall the next tile's first K-tiles.
  constexpr int BH = PAIR ? 2 : 1;                        // sets of W source offsets
  static_assert(64 % (MH * 16) == 0, "a wave's A pieces are whole row-block groups apart");
  constexpr int A_STEP = (64 / (MH * 16)) * (MT * 16);   // tile rows between a wave's consecutive A pieces
  unsigned a_voff, b_voff[BH][C::B_IPW];
  unsigned a_rd, b_rd;
  auto lane_consts = [&]() __attribute__((always_inline)) {
    int ln = lane_now();
    asm volatile("" : "+v"(ln));                            // opaque: not hoisted out of the tile loop
    const int sr = ln >> 3, sc = ln & 7;
    {
      // (a wave's piece i lies 64 LDS rows = A_STEP tile rows behind its piece 0, with the same swizzle: a wave-uniform distance that issue_a adds to the
      // scalar base -- one offset register)
      const int hr = wave * 8 + sr;                                   // LDS row inside the unit (piece `wave`)
      const int trow = (hr / (MH * 16)) * (MT * 16) + (hr % (MH * 16));   // tile row of half 0 (half 1: + MH*16)
      a_voff = (unsigned)trow * (unsigned)p.lda * 2u + (unsigned)(sc ^ ((hr >> 1) & 7)) * 16u;
    }
    // (natural column map: the second half's columns are the first half's + NH * 16, a wave-uniform distance that issue_b adds to the scalar base instead --
    // one set of offsets; the pair-interleaved map of the bf16 epilogues has no such constant)
#pragma unroll
    for (int h = 0; h < BH; ++h)
#pragma unroll
      for (int i = 0; i < C::B_IPW; ++i) {
        const int hr = (i * 8 + wave) * 8 + sr;
        const int wcs = hr / (NH * 16), in = hr % (NH * 16);            // wave column block, slot inside the block's half
        const int jt = h * NH + (in >> 4);
        const int col = wcs * (NT * 16) + slot_col<PAIR>(jt, in & 15);
        b_voff[h][i] = (unsigned)col * (unsigned)p.ldb * 2u + (unsigned)(sc ^ ((hr >> 1) & 7)) * 16u;
      }
    // fragment read addresses: lane (fr = lane & 15, fq = lane >> 4) reads row fr of a 16-row tile, chunk (kk*4 + fq) ^ ((row >> 1) & 7);
    // rows of different tiles differ by multiples of 16, so the XOR term is (fr >> 1) and kk flips byte-offset bit 6
    const int fr = ln & 15, fq = ln >> 4;
    const unsigned rd_lane = (unsigned)fr * 128u + (unsigned)((fq ^ (fr >> 1)) << 4);
    a_rd = (unsigned)(wr * (MH * 16)) * 128u + rd_lane;                        // + unit offset + mt * 2048, ^ 64 for kk = 1
    b_rd = (unsigned)(wc * (NH * 16)) * 128u + rd_lane;                        // + unit offset + nt * 2048
  };
  lane_consts();

  // unit U of a K-tile: 0 = X0, 1 = Y0, 2 = Y1, 3 = X1 (the order in which the phases need them)
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  auto issue_a = [&](const unsigned char* abase, int half, unsigned dst) __attribute__((always_inline)) {
    const unsigned char* src = abase + (size_t)half * (MH * 16) * p.lda * 2;
#pragma unroll
    for (int i = 0; i < C::A_IPW; ++i)
      if (C::A_REM == 0 || i + 1 < C::A_IPW || wave < C::A_REM) glds16(a_voff, src + (size_t)i * A_STEP * p.lda * 2, lds0 + dst + (i * 8 + wave) * 1024);
  };
  auto issue_b = [&](const unsigned char* bbase, int half, unsigned dst) __attribute__((always_inline)) {
    const unsigned char* src = PAIR ? bbase : bbase + (size_t)half * (NH * 16) * p.ldb * 2;
#pragma unroll
    for (int i = 0; i < C::B_IPW; ++i)
      if (C::B_REM == 0 || i + 1 < C::B_IPW || wave < C::B_REM) glds16((PAIR && half) ? b_voff[BH - 1][i] : b_voff[0][i], src, lds0 + dst + (i * 8 + wave) * 1024);
  };
  // unit U of K-tile n goes to buffer n & 1 (Y0 under Y3: ring slot n % 3, passed as `y0slot`)
  auto issue_unit = [&](auto uc, int buf, int y0slot, const unsigned char* abase, const unsigned char* bbase) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    constexpr bool isx = (U == 0 || U == 3);
    constexpr int half = (U == 0 || U == 1) ? 0 : 1;
    constexpr bool isa = (isx == XA);
    unsigned dst;
    if constexpr (U == 1 && C::Y3) dst = C::RING_Y0 + y0slot * C::Y_UNIT;
    else dst = buf * C::BUF + (U == 0 ? C::OFF_X0 : (U == 3 ? C::OFF_X1 : (U == 2 ? C::OFF_Y1 : C::OFF_Y0)));
    if constexpr (isa) issue_a(abase, half, dst); else issue_b(bbase, half, dst);
  };

  // ---- stream cursors (all on the scalar unit): operand bases of K-tiles s+1 and s+2 of the stream, where s is the K-tile being computed
  struct Cur { const unsigned char* a; const unsigned char* b; int t; };
  auto cur_origin = [&](Cur& c, int tk) __attribute__((always_inline)) {
    int m0, n0; tile_origin(tk, m0, n0);
    c.a = reinterpret_cast<const unsigned char*>(p.A + (size_t)m0 * p.lda);
    c.b = reinterpret_cast<const unsigned char*>(p.B + (size_t)n0 * p.ldb);
  };
  auto cur_next = [&](Cur& c) __attribute__((always_inline)) {                            // (a cursor runs at most two K-tiles ahead: it crosses into the NEXT tile, once per tile)
    c.a += 128; c.b += 128;
    if (++c.t == nk) { c.t = 0; cur_origin(c, tk_nxt < Tx ? tk_nxt : tk_cur); }
  };
  Cur c1, c2;                                              // K-tiles s+1, s+2
  c1.t = 0; cur_origin(c1, tk_cur);

  f32x4 acc[MT][NT];
  constexpr int YSETS = C::Y3 ? 1 : 2;
  bf8_t fa[XA ? 1 : YSETS][MH][2], fb[XA ? YSETS : 1][NH][2];       // [set][tile][kk]; the major operand has one set
  auto read_a = [&](auto setc, const unsigned char* unit) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int mt = 0; mt < MH; ++mt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[set][mt][kk] = *reinterpret_cast<const bf8_t*>(unit + ((a_rd + mt * 2048) ^ (kk << 6)));
  };
  auto read_b = [&](auto setc, const unsigned char* unit) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int nt = 0; nt < NH; ++nt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb[set][nt][kk] = *reinterpret_cast<const bf8_t*>(unit + ((b_rd + nt * 2048) ^ (kk << 6)));
  };
  // X unit h -> set 0; Y unit h -> set h (set 0 under Y3)
  auto read_x = [&](const unsigned char* unit) __attribute__((always_inline)) { if constexpr (XA) read_a(std::integral_constant<int, 0>{}, unit); else read_b(std::integral_constant<int, 0>{}, unit); };
  auto read_y = [&](auto hc, const unsigned char* unit) __attribute__((always_inline)) {
    constexpr int set = C::Y3 ? 0 : decltype(hc)::value;
    if constexpr (XA) read_b(std::integral_constant<int, set>{}, unit); else read_a(std::integral_constant<int, set>{}, unit);
  };
  // MFMAs of quadrant (X half xh, Y half yh), operand roles swapped (a = weight fragment, b = activation fragment): lane (c, q) owns
  // C[token c][slots q*4 .. q*4+3] of every tile (gemm_epi.h)
  int stamp_s = 0, stamp_ph = 0;                           // (diagnostic builds: K-tile / phase the MFMA-run stamps belong to)
#ifdef G8_DUMMY_VALU      // experiment (tools/ab): G8_DUMMY_VALU plain + G8_DUMMY_EXP transcendental vector instructions in every READ slot -- how much epilogue work hides there?
  float dv[4] = {0.1f, 0.2f, 0.3f, 0.4f};
  asm volatile("" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(dv[3]));
  auto dummy_valu = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < G8_DUMMY_VALU; ++i) asm volatile("v_fma_f32 %0, %0, 0.5, 0.5" : "+v"(dv[i & 3]));
#pragma unroll
    for (int i = 0; i < G8_DUMMY_EXP; ++i) asm volatile("v_exp_f32 %0, -%0" : "+v"(dv[i & 3]));
  };
#else
  auto dummy_valu = [&]() __attribute__((always_inline)) {};
#endif
  auto quadrant = [&](auto xhc, auto yhc) __attribute__((always_inline)) {
    constexpr int xh = decltype(xhc)::value, yh = decltype(yhc)::value;
    constexpr int mh = XA ? xh : yh, nh = XA ? yh : xh;
    constexpr int yset = C::Y3 ? 0 : yh, aset = XA ? 0 : yset, bset = XA ? yset : 0;
    STAMP_M(stamp_ph, 0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < MH; ++mt)
#pragma unroll
        for (int nt = 0; nt < NH; ++nt)
          acc[mh * MH + mt][nh * NH + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[bset][nt][kk], fa[aset][mt][kk], acc[mh * MH + mt][nh * NH + nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    STAMP_M(stamp_ph, 1);
    stamp_ph = (stamp_ph + 1) & 3;
  };
  constexpr int NY = XA ? NH * 2 : MH * 2;                // ds_reads of one Y fragment set

  // ---- one K-tile = four phases.  s = stream index of the K-tile (buffer BUFI = s & 1, compile-time through the two-fold unrolling);
  // r3 = s % 3 (Y0 ring slot under Y3).
  int r3 = 0;
  constexpr int ST = epilogue_stores<EPI, DUAL, MT, NT>();
  // the QKV + RoPE kernel with a bias on 256 x 320 tiles sits at 256 registers: the ticket is read at the head of P1's MFMA slot there (the form that fits;
  // ~1 % of that launch), everywhere else in P2's read slot
  constexpr bool RELAY_IN_P1 = C::Y3 && EPI == ONEPROT_EPI_QKV_ROPE && HB;
  unsigned relay_addr = 0;                                  // LDS byte address of the ticket relay word of the tile being entered
  int lim = tk_nxt < Tx ? 0x7fffffff : nk;                 // K-tiles the operand stream may run ahead to, counted from the current tile's first
  constexpr unsigned RELAY = C::Y3 ? 0u : (unsigned)C::LDS;      // byte offset of the ticket relay word (Y3: inside the Y0 ring slot that is dead at that time)
  auto ktile = [&](auto bufc, int s, int t, const bool after_epi, const int relay_flag) __attribute__((always_inline)) {
    constexpr int BUFI = decltype(bufc)::value;
    stamp_s = s; stamp_ph = 0;
    const unsigned char* bufp = smem + BUFI * C::BUF;
    const unsigned char* y0p = C::Y3 ? smem + C::RING_Y0 + r3 * C::Y_UNIT : bufp + C::OFF_Y0;
    const int r3n2 = r3 == 0 ? 2 : r3 - 1;                 // (s + 2) % 3
    // K-tiles s+1 / s+2 of the stream exist: inside this tile, or in the next one (t = K-tile inside the tile; lim = nk, or "no end" when a next tile is
    // known).  With drawn tickets the next ticket arrives in P1 of the first K-tile after an epilogue (relay_now) and lim with it; nk >= 4 there
    const bool more1 = t + 1 < lim;
    // P1 (X0, Y0): X0 reads first and retired before the barrier -- its LDS is refilled in the NEXT phase
    read_x(bufp + C::OFF_X0);
    __builtin_amdgcn_sched_barrier(0);
    read_y(std::integral_constant<int, 0>{}, y0p);
    __builtin_amdgcn_sched_barrier(0);
    if (more1 && !after_epi) issue_unit(std::integral_constant<int, 3>{}, BUFI ^ 1, 0, c1.a, c1.b);          // X1 of K-tile s+1 (issued ahead of the epilogue at a tile start)
    dummy_valu();
    wait_lgkm<NY>();
    bar();
    STAMP(0);
    if constexpr (RELAY_IN_P1 && DYN) {                     // (see P2: this instantiation has no register to spare there)
      unsigned relay = 0;
      if (relay_flag != 0) relay = lds_read32(relay_addr);
      wait_lgkm<0>();
      if (relay_flag != 0) { tk_nxt = first_lane(relay); lim = tk_nxt < Tx ? 0x7fffffff : nk; }
    } else {
      wait_lgkm<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    bar();
    STAMP(1);
    // P2 (X0, Y1).  First K-tile after an epilogue with drawn tickets: the next tile's ticket arrives here, in a READ slot (wave 0 wrote the word before
    // P1's first barrier; under Y3 the slot it lies in is refilled from P3 on) -- not at the head of an MFMA slot, where every instruction delays the run
    if constexpr (!RELAY_IN_P1 && DYN) {
      if (relay_flag != 0) {
        const unsigned relay = lds_read32(relay_addr);
        wait_lgkm<0>();
        tk_nxt = first_lane(relay);
        lim = tk_nxt < Tx ? 0x7fffffff : nk;
      }
    }
    const bool more2 = t + 2 < lim;
    c2 = c1; cur_next(c2);
    read_y(std::integral_constant<int, 1>{}, bufp + C::OFF_Y1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) issue_unit(std::integral_constant<int, 0>{}, BUFI, 0, c2.a, c2.b);              // X0 of K-tile s+2
    dummy_valu();
    bar();
    STAMP(2);
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    bar();
    STAMP(3);
    // P3 (X1, Y1)
    read_x(bufp + C::OFF_X1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) issue_unit(std::integral_constant<int, 1>{}, BUFI, r3n2, c2.a, c2.b);           // Y0 of K-tile s+2
    dummy_valu();
    bar();
    STAMP(4);
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    bar();
    STAMP(5);
    // P4 (X1, Y0): Y0 again from LDS under Y3, else still in registers; K-tile s+1 must have landed when this phase's first barrier opens
    // (it is read from the next phase on)
    if constexpr (C::Y3) { read_y(std::integral_constant<int, 0>{}, y0p); __builtin_amdgcn_sched_barrier(0); }
    if (more2) {
      issue_unit(std::integral_constant<int, 2>{}, BUFI, 0, c2.a, c2.b);                       // Y1 of K-tile s+2
      // after_epi (first K-tile after an epilogue): K-tile s+1 was requested BEFORE the epilogue's ST stores, so that many more operations may
      // stay in flight -- the stores drain under the first two K-tiles of the new tile instead of being waited for here
      constexpr int N0 = C::x_cnt(0) + 2 * C::y_cnt(0), N1 = C::x_cnt(1) + 2 * C::y_cnt(1);
      if (!after_epi) { if (grp == 0) wait_vmcnt<N0>(); else wait_vmcnt<N1>(); }
      else { if (grp == 0) wait_vmcnt<(N0 + ST > 63 ? 63 : N0 + ST)>(); else wait_vmcnt<(N1 + ST > 63 ? 63 : N1 + ST)>(); }
    } else {
      wait_vmcnt<0>();
    }
    dummy_valu();
    bar();
    STAMP(6);
    if constexpr (C::Y3) wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    bar();
    STAMP(7);
    c1 = c2;
    r3 = r3 == 2 ? 0 : r3 + 1;
  };

  // ---- merged schedule (C::M2): one K-tile = TWO phases of two quadrants each (twice the MFMAs behind one pair of barriers: the barrier pair and
  // the wake-up around it cost ~145 cycles per phase however long the MFMA run is -- ablation stamps, NOTEBOOK 6c).
  //   phase A: read X0, Y0, Y1 | issue X1 of K-tile s+1                 | wait: X1 of K-tile s landed            | MFMAs X0 x (Y0, Y1)
  //   phase B: read X1         | issue X0, Y0, Y1 of K-tile s+2         | wait: those units of K-tile s+1 landed | MFMAs X1 x (Y1, Y0)
  // Every read is retired before the phase's first barrier, so the LDS a phase read may be refilled from the next phase on.  Each wait leaves
  // exactly the units of one K-tile younger than its target in flight: 2 * (x_cnt + y_cnt) operations.
  int ew = 0;                                               // waits that still have the last epilogue's stores among their younger operations
  bool x1_ahead = false;                                    // X1 of K-tile s+1 was requested before the epilogue
  auto vwait2 = [&](bool counted) __attribute__((always_inline)) {
    constexpr int N0 = 2 * (C::x_cnt(0) + C::y_cnt(0)), N1 = 2 * (C::x_cnt(1) + C::y_cnt(1));
    if (!counted) wait_vmcnt<0>();
    else if (ew > 0) { --ew; if (grp == 0) wait_vmcnt<(N0 + ST > 63 ? 63 : N0 + ST)>(); else wait_vmcnt<(N1 + ST > 63 ? 63 : N1 + ST)>(); }
    else { if (grp == 0) wait_vmcnt<N0>(); else wait_vmcnt<N1>(); }
  };
  auto ktile2 = [&](auto bufc, int s, int t) __attribute__((always_inline)) {
    constexpr int BUFI = decltype(bufc)::value;
    const unsigned char* bufp = smem + BUFI * C::BUF;
    const bool more1 = t + 1 < lim;
    const bool first = x1_ahead;                            // first K-tile after an epilogue
    // phase A
    read_x(bufp + C::OFF_X0);
    read_y(std::integral_constant<int, 0>{}, bufp + C::OFF_Y0);
    read_y(std::integral_constant<int, 1>{}, bufp + C::OFF_Y1);
    __builtin_amdgcn_sched_barrier(0);
    if (more1 && !x1_ahead) issue_unit(std::integral_constant<int, 3>{}, BUFI ^ 1, 0, c1.a, c1.b);
    x1_ahead = false;
    vwait2(more1);
    wait_lgkm<0>();
    bar();
    STAMP(0);
    __builtin_amdgcn_sched_barrier(0);
    quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    bar();
    STAMP(1);
    if constexpr (DYN) if (first) {                         // (the next tile's ticket: see ktile)
      const unsigned relay = lds_read32(relay_addr);
      wait_lgkm<0>();
      tk_nxt = first_lane(relay);
      lim = tk_nxt < Tx ? 0x7fffffff : nk;
    }
    const bool more2 = t + 2 < lim;
    // phase B
    c2 = c1; cur_next(c2);
    read_x(bufp + C::OFF_X1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      issue_unit(std::integral_constant<int, 0>{}, BUFI, 0, c2.a, c2.b);
      issue_unit(std::integral_constant<int, 1>{}, BUFI, 0, c2.a, c2.b);
      issue_unit(std::integral_constant<int, 2>{}, BUFI, 0, c2.a, c2.b);
    }
    vwait2(more2);
    wait_lgkm<0>();
    bar();
    STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    bar();
    STAMP(3);
    c1 = c2;
  };

  // ---- prologue: units 0..6 of the stream (K-tile 0 whole, X0 Y0 Y1 of K-tile 1); K-tile 0 landed before the first phase reads it.
  // Accumulators start at zero: the bias is added by the epilogue (gemm_epi8.h).
  // (an opaque zero: with a literal 0 hipcc peels the first K-tiles of every tile off the loop to fold the constant into the first MFMAs, and
  // the peeled copy -- 256 registers, like the loop -- spills accumulators and reloads them behind vmcnt(0))
  auto zero_acc = [&]() __attribute__((always_inline)) {
    float z = 0.f;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){z, z, z, z};
  };
  zero_acc();
  issue_unit(std::integral_constant<int, 0>{}, 0, 0, c1.a, c1.b);
  issue_unit(std::integral_constant<int, 1>{}, 0, 0, c1.a, c1.b);
  issue_unit(std::integral_constant<int, 2>{}, 0, 0, c1.a, c1.b);
  issue_unit(std::integral_constant<int, 3>{}, 0, 0, c1.a, c1.b);
  cur_next(c1);                                            // c1 = K-tile 1 (S >= 2: K % 128 == 0)
  issue_unit(std::integral_constant<int, 0>{}, 1, 1, c1.a, c1.b);
  issue_unit(std::integral_constant<int, 1>{}, 1, 1, c1.a, c1.b);
  issue_unit(std::integral_constant<int, 2>{}, 1, 1, c1.a, c1.b);
  if constexpr (C::M2) vwait2(true);                       // X0, Y0, Y1 of K-tile 0 landed; X1 of it and three units of K-tile 1 may be in flight
  else { if (grp == 0) wait_vmcnt<C::x_cnt(0) + 2 * C::y_cnt(0)>(); else wait_vmcnt<C::x_cnt(1) + 2 * C::y_cnt(1)>(); }
  bar();
  // Per output tile: group 1 drops one barrier behind (stagger), the K loop runs, group 0 waits one barrier (both groups aligned again) and
  // the two groups run the epilogue TOGETHER -- two waves per SIMD issue vector instructions at twice the rate of one, so epilogues run one
  // group after the other would take twice as long -- while the first K-tiles of the next tile keep arriving.
  // Order at a tile boundary: X1 of the next tile's K-tile 1 (now BOTH of its first K-tiles are requested) -> epilogue (bias / residual / GELU'
  // loads pipelined ahead of its stores, gemm_epi8.h) -> per-lane constants re-derived -> accumulators = 0.  Nothing after the stores waits for
  // them until the K-tile-2 wait, two K-tiles later.
  int s = 0;
  bool after_epi = false;
  const int draw_lane = (dyn && wave == 0) ? 1 : 0;         // EXEC mask of the draw: lane 0 of wave 0
  const unsigned* const headp = dyn ? p.sched + SW_HEAD(x) : nullptr;
  int tiles_mine = 0;                                       // output tiles this work-group has written (sw_leave checks the launch's total)
  int relay_flag = 0;                                       // 1 in the first K-tile of a tile whose successor's ticket was drawn (it arrives there)
  // One LDS-DMA-free way to hand a drawn ticket to all eight waves: wave 0 draws (a returning agent-scope atomic, issued ahead of the tile's LAST two K-tiles
  // and looked at behind their last counted wait: loads return in order, so it has returned when that wait is over), keeps the value in a scalar register over
  // the epilogue and writes it into LDS behind it -- 4 bytes that are free at that time: spare bytes behind the buffers, or under Y3 (all 160 KB taken) the Y0
  // ring slot of the K-tile after next, which nobody reads before its refill is issued in P3 of the next K-tile; everybody reads the word behind the first
  // barrier of that K-tile (ktile / ktile2).
  auto kpair = [&](int t) __attribute__((always_inline)) {
    if constexpr (C::M2) {
      ktile2(std::integral_constant<int, 0>{}, s, t);
      ktile2(std::integral_constant<int, 1>{}, s + 1, t + 1);
    } else {
      ktile(std::integral_constant<int, 0>{}, s, t, after_epi, relay_flag);
      ktile(std::integral_constant<int, 1>{}, s + 1, t + 1, false, 0);
      relay_flag = 0;
      after_epi = false;
    }
    s += 2;
  };
#pragma clang loop unroll(disable)
  for (;;) {
    if (grp == 1) bar();
    // The draw for the tile after next goes out ahead of the tile's LAST two K-tiles (one wave, one lane: the statement is in every iteration with an empty
    // EXEC mask elsewhere, so that the register it writes behind the compiler's back is defined in the iteration it is used after and never merged with another
    // definition) and is looked at behind their last counted wait: at most the three youngest units are in flight then, or nothing -- loads return in order,
    // so the draw has returned.  oneprot_amd/csrc/check_async_regs.py verifies in the ISA that nothing copies or reuses the register in between.
    unsigned drawn_v;                                        // (defined in the last iteration, used behind the loop)
    int t = 0;
#pragma clang loop unroll(disable)
    do {                                                   // (nk >= 2)
      // (tk_nxt: with drawn tickets it arrives in the tile's first K-tile -- it is looked at in the last pair only, nk >= 4)
      if constexpr (DYN) if (t + 2 >= nk) drawn_v = draw_async(headp, draw_lane & (int)((unsigned)(tk_nxt - Tx) >> 31));      // (all scalar: tk_nxt < Tx <=> the difference is negative)
      kpair(t);
      t += 2;
    } while (t < nk);
    const bool has_next = tk_nxt < Tx;
    int drawn = 0x7fffffff;
    if constexpr (DYN) {
      int late;
      drawn = draw_result(drawn_v, late);                    // (its ; DRAWN marker is what check_async_regs.py looks for)
      if (late && lane_now() == 0) __hip_atomic_fetch_add(p.sched + SW_LATE_DRAWS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // diagnostic counter: draws that had not returned behind the counted wait
    }
    STAMP_E(0);
    if (grp == 0) bar();
    STAMP_E(1);
    int m0, n0; tile_origin(tk_cur, m0, n0);
    if (has_next) {
      issue_unit(std::integral_constant<int, 3>{}, 1, 0, c1.a, c1.b);        // X1 of the next tile's K-tile 1 (c1; s is even: buffer 1)
      after_epi = true;
      x1_ahead = true; ew = 3;                                               // (merged schedule: the next three waits have the stores behind their targets)
    }
    STAMP_E(2);
    if (p.nt_store != 77) {                                                  // (77: timing-only runs of the main loop, tools/ab)
      int lane_e = lane_now();
      asm volatile("" : "+v"(lane_e));                                        // the epilogue's per-lane addresses are formed here, per tile: hoisted out of the tile loop they are spilled
      if constexpr (DirectMap<EPI>::PAIR) epilogue_pair<EPI, HB, DUAL, MT, NT>(p, acc, m0, n0, wr, wc, lane_e);
      else if constexpr (EPI == ONEPROT_EPI_QKV_ROPE) {
        if (p.hd == 32) epilogue_rope32<HB, MT, NT>(p, acc, m0, n0, wr, wc, lane_e);
        else if constexpr ((NT * 16) % 64 == 0) epilogue_rope64<HB, MT, NT>(p, acc, m0, n0, wr, wc, lane_e);      // head_dim 64 (whole heads per wave column block)
      } else if constexpr (EPI == G8_EPI_RESID_LN) epilogue_resid_ln<HB, MT, NT>(p, acc, m0, n0, wr, wc, lane_e, ln_tag);
      else epilogue_f32<EPI, HB, DUAL, MT, NT>(p, acc, m0, n0, wr, wc, lane_e);
    }
    STAMP_E(3);
    ++tiles_mine;
    if (!has_next) break;
    if constexpr (dyn) {
      // (under Y3 the word lies in the Y0 ring slot of K-tile s+2 = slot (r3 + 2) % 3, r3 being the slot of the new tile's first K-tile)
      relay_addr = lds0 + (C::Y3 ? C::RING_Y0 + (r3 == 0 ? 2 : r3 - 1) * C::Y_UNIT : RELAY);
      if (wave == 0) lds_write32(relay_addr, (unsigned)drawn);      // (every lane of the wave: the same word, the same value)
      tk_cur = tk_nxt;                                                       // (tk_nxt: read back by every wave in the first K-tile of the new tile)
      tk_nxt = 0x7fffffff; lim = nk; relay_flag = 1;
    } else {
      tk_cur = tk_nxt; tk_nxt += g8n;
      lim = tk_nxt < Tx ? 0x7fffffff : nk;
    }
    lane_consts();
    zero_acc();
    STAMP_E(4);
  }
  if (p.sched && wave == 0 && lane_now() == 0) sw_leave(p.sched, gridDim.x, (unsigned)tiles_mine, tiles_expected);
#ifdef G8_DUMMY_VALU
  asm volatile("" :: "v"(dv[0]), "v"(dv[1]), "v"(dv[2]), "v"(dv[3]));
#endif
}

static int g_dph_groups = 1, g_dph_sleeps = 0, g_g8n_cap = 0;
static unsigned* g_dyn_sched = nullptr;      // oneprot_dynamic_tiles: the workspace whose queues hand out the tiles, or null (static tile lists)
#ifndef G8_CSPLIT
#define G8_CSPLIT 0
#endif
static int g_csplit = G8_CSPLIT;      // 0 automatic; 1 / 2 / 4 forced (A/B builds)
template <class C, int EPI, bool HB, bool DUAL>
static int launch_cfg(GemmArgs a, hipStream_t s) {
  static bool configured = false;
  static int n_cu = 0;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k_gemm8<C, EPI, HB, DUAL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS + (C::Y3 ? 0 : 64)) != hipSuccess) return OP_ELAUNCH;
    if constexpr (EPI != G8_EPI_RESID_LN)
      if (hipFuncSetAttribute((const void*)k_gemm8<C, EPI, HB, DUAL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS + (C::Y3 ? 0 : 64)) != hipSuccess) return OP_ELAUNCH;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return OP_ELAUNCH;
    n_cu = prop.multiProcessorCount;
    configured = true;
  }
  a.tiles_m = a.M / C::BM; a.tiles_n = a.N / C::BN;
  // column split: when W (N x K bf16) next to one round of activation panels overflows an XCD's 4 MB L2 and the activation panels are short (K small), halve W per XCD
  int cs = 1;
  if (g_csplit > 0) cs = g_csplit;
  else if ((size_t)a.N * a.K * 2 > (size_t)(5 << 19) && a.K <= 1024) cs = 2;
  while (cs > 1 && (a.tiles_n % cs || a.tiles_m < 8 / cs)) cs >>= 1;
  if (EPI == G8_EPI_RESID_LN) cs = 1;                       // the column tiles of a row panel must run side by side (gemm_epi8.h: epilogue_resid_ln)
  a.csplit = cs;
  const long per_xcd = (long)((a.tiles_m + 8 / cs - 1) / (8 / cs)) * (a.tiles_n / cs);
  int g8n = n_cu / 8;
  if (g8n > per_xcd) g8n = (int)per_xcd;
  if (g8n < 1) g8n = 1;
  if (g_g8n_cap > 0 && g8n > g_g8n_cap) g8n = g_g8n_cap;      // experiment hook: fewer work-groups per XCD
  if (EPI == G8_EPI_RESID_LN) {                              // tile u = q * g8n + w: the tiles_n tiles of a panel are in flight together iff g8n is a multiple of tiles_n
    g8n -= g8n % a.tiles_n;
    if (g8n < a.tiles_n) return G8_NOT_ELIGIBLE;
  }
  // tiles drawn from the work queues of the sched workspace (oneprot_dynamic_tiles): needs the workspace and four K-tiles per tile
  a.dyn = (EPI != G8_EPI_RESID_LN && g_dyn_sched != nullptr && a.K >= 256) ? 1 : 0;
  if (a.dyn) a.sched = g_dyn_sched;
  else if (EPI != G8_EPI_RESID_LN) a.sched = nullptr;
  if constexpr (EPI != G8_EPI_RESID_LN) {
    if (a.dyn) {
      hipLaunchKernelGGL((k_gemm8<C, EPI, HB, DUAL, true>), dim3(g8n * 8), dim3(512), C::LDS + (C::Y3 ? 0 : 64), s, a, g8n, g_dph_groups, g_dph_sleeps);      // (+ the ticket relay word where the buffers leave room)
      return launch_status();
    }
  }
  hipLaunchKernelGGL((k_gemm8<C, EPI, HB, DUAL, false>), dim3(g8n * 8), dim3(512), C::LDS + (C::Y3 ? 0 : 64), s, a, g8n, g_dph_groups, g_dph_sleeps);
  return launch_status();
}

typedef Cfg<2, 4, 8, 4, true, false, true> C256M;      // 256 x 256 with merged phases (32 MFMAs per barrier pair)
typedef Cfg<2, 4, 8, 4, true, false> C256;       // 256 x 256, wave tile 128 x 64, X = A (32 registers per quadrant), both W sets held
typedef Cfg<4, 2, 4, 10, false, true> C320;      // 256 x 320, wave tile 64 x 160, X = W (40 registers), ONE activation set, A0 in a 3-slot ring (160 KB of LDS)

template <class C>
static bool eligible(const GemmArgs& a, int epi) {
  if (a.M % C::BM || a.N % C::BN || a.K % 128 || a.lda % 8 || a.ldb % 8) return false;
  if ((size_t)C::BM * a.lda * 2 >= (1ull << 31) || (size_t)C::BN * a.ldb * 2 >= (1ull << 31)) return false;      // 32-bit per-lane source offsets
  if (epi == ONEPROT_EPI_GELU_BWD && a.bias) return false;
  if (epi == ONEPROT_EPI_QKV_ROPE) {
    if (a.hd != 32 && a.hd != 64) return false;
    if ((C::NTW * 16) % a.hd) return false;                // a head must not straddle the wave's column block
    if ((a.H * a.hd) % (C::NTW * 16)) return false;        // nor a wave block the q / k / v sections
    if (a.L % 16) return false;
  }
  return true;
}

template <class C>
static int launch_epi(int epi, const GemmArgs& a, hipStream_t s) {
  const bool hb = a.bias != nullptr, dual = a.out1 != nullptr;
  switch (epi) {
    case ONEPROT_EPI_BF16: return hb ? launch_cfg<C, ONEPROT_EPI_BF16, true, false>(a, s) : launch_cfg<C, ONEPROT_EPI_BF16, false, false>(a, s);
    case ONEPROT_EPI_F32: return hb ? launch_cfg<C, ONEPROT_EPI_F32, true, false>(a, s) : launch_cfg<C, ONEPROT_EPI_F32, false, false>(a, s);
    case ONEPROT_EPI_BIAS_GELU: return dual ? launch_cfg<C, ONEPROT_EPI_BIAS_GELU, true, true>(a, s) : launch_cfg<C, ONEPROT_EPI_BIAS_GELU, true, false>(a, s);
    case ONEPROT_EPI_BIAS_RESID:
      if (hb) return dual ? launch_cfg<C, ONEPROT_EPI_BIAS_RESID, true, true>(a, s) : launch_cfg<C, ONEPROT_EPI_BIAS_RESID, true, false>(a, s);
      return dual ? launch_cfg<C, ONEPROT_EPI_BIAS_RESID, false, true>(a, s) : launch_cfg<C, ONEPROT_EPI_BIAS_RESID, false, false>(a, s);
    case ONEPROT_EPI_GELU_BWD: return launch_cfg<C, ONEPROT_EPI_GELU_BWD, false, false>(a, s);
    case ONEPROT_EPI_QKV_ROPE: return hb ? launch_cfg<C, ONEPROT_EPI_QKV_ROPE, true, false>(a, s) : launch_cfg<C, ONEPROT_EPI_QKV_ROPE, false, false>(a, s);
    default: return OP_EINVAL;
  }
}

}  // namespace g8

// FFN-2 + bias + residual + the LayerNorm that follows, 256 x 320 tiles only (N = 320, 640 or 1280): see epilogue_resid_ln.  The partial statistics, the launch
// epoch their tags come from and the sticky error flag live in the caller's sched workspace (sched_ws.h): nothing is allocated, cleared or counted on the host.
static int g_ln_poll_max = 1 << 20;                          // ~1 s of polling: far beyond any wait that ends by itself (a partner's remaining tile time)
// 0: the shape is not made of whole tiles this form serves; 1: it is, but with fewer than 192 tiles (a persistent work-group per CU does not pay: callers
// keep the pair of launches unless told otherwise -- the parity tests against the oracle run small batches through it); 2: it is and it pays
int gemm8_ln_eligible(long M, int N, int K) {
  typedef g8::C320 C;
  if (M <= 0 || M % C::BM || N % C::BN || K % 128 || (N / C::BN != 1 && N / C::BN != 2 && N / C::BN != 4)) return 0;
  if ((M / C::BM + 7) / 8 * (N / C::BN) < N / C::BN) return 0;
  return (M / C::BM) * (long)(N / C::BN) >= 192 ? 2 : 1;
}
size_t sched_workspace_bytes(long M_max) { return (size_t)SW_HEADER_BYTES + (size_t)(M_max > 0 ? M_max : 0) * 8 * 16; }
int launch_gemm8_ln(GemmArgs a, void* ws, size_t ws_bytes, hipStream_t s) {
  typedef g8::C320 C;
  if (gemm8_ln_eligible(a.M, a.N, a.K) == 0 || !g8::eligible<C>(a, ONEPROT_EPI_BIAS_RESID)) return G8_NOT_ELIGIBLE;
  if (!ws || ((uintptr_t)ws & 127) || ws_bytes < sched_workspace_bytes(a.M)) return OP_EINVAL;
  a.sched = (unsigned*)ws;
  a.ln_part = (unsigned char*)ws + SW_HEADER_BYTES;
  a.ln_slots = 2 * (a.N / C::BN);
  a.ln_poll_max = g_ln_poll_max;
  return a.bias ? g8::launch_cfg<C, G8_EPI_RESID_LN, true, false>(a, s) : g8::launch_cfg<C, G8_EPI_RESID_LN, false, false>(a, s);
}
extern "C" void oneprot_gemm_resid_ln8_poll_bound(int polls) { g_ln_poll_max = polls < 0 ? (1 << 20) : polls; }
extern "C" void oneprot_dynamic_tiles(void* sched_ws, size_t bytes) { g8::g_dyn_sched = (sched_ws && bytes >= SW_HEADER_BYTES && !((uintptr_t)sched_ws & 127)) ? (unsigned*)sched_ws : nullptr; }
void* dynamic_tiles_workspace() { return g8::g_dyn_sched; }

// experiment hook (tools/ab/g8_ab.py): de-phasing of the persistent work-groups
extern "C" void oneprot_gemm8_dephase(int groups, int sleeps) { g8::g_dph_groups = groups & 0xffff; g8::g_dph_sleeps = sleeps; g8::g_g8n_cap = groups >> 16; }

// cfg 0: 256 x 256, cfg 1: 256 x 320, cfg 2: 256 x 256 with merged phases.  Returns G8_NOT_ELIGIBLE when the problem is not made of whole tiles of that configuration or has fewer
// than `min_tiles` of them (a persistent work-group per CU only pays when most CUs get a tile).
int launch_gemm8(int epi, const GemmArgs& a, int cfg, long min_tiles, hipStream_t s) {
  if (cfg == 0 || cfg == 2) {
    if (!g8::eligible<g8::C256>(a, epi) || (long)(a.M / g8::C256::BM) * (a.N / g8::C256::BN) < min_tiles) return G8_NOT_ELIGIBLE;
    return cfg == 2 ? g8::launch_epi<g8::C256M>(epi, a, s) : g8::launch_epi<g8::C256>(epi, a, s);
  }
  if (!g8::eligible<g8::C320>(a, epi) || (long)(a.M / g8::C320::BM) * (a.N / g8::C320::BN) < min_tiles) return G8_NOT_ELIGIBLE;
  return g8::launch_epi<g8::C320>(epi, a, s);
}
