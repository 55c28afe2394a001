// tapgemm.hip — the MFMA contraction kernel behind every Linear / 1x1 / 3x3 / (3,1,1) conv of
// the VGen UNet and VAE (see include/vgen_hip.h for the reference call sites).
//
// Design (gfx950 / CDNA4):
//   * implicit GEMM over "taps": out[m,:] = sum_tap A[src(m,tap), :] @ W[:, tap, :]^T.  With the
//     channels-last row layout every tap's K-slab is one contiguous C-length row of a shifted
//     pixel / frame, so a conv A-tile is a row GATHER of 128-byte pieces — no im2col buffer,
//     no torch.cat, no F.interpolate, no rearrange.
//   * one kernel template, three block shapes chosen per launch by make_plan() (table further down):
//     "pp" 256 x BN, BK 64, 8 waves, one block per CU, ping-pong wave groups; "dual" 256 x BN, BK 32,
//     4 waves with 128-row wave tiles, two blocks per CU; "pp128" 128 x BN for under-filled launches.
//     BN = 128 (N % 128 == 0), 160 (N = 320 / 960 ...: exact tiles instead of 2.5 x 128) or 64
//     (small / odd N); 16x16x32 MFMA fragments.  256-row tiles halve the L2->LDS bytes per flop of a
//     128 x 128 tile (85-98 flop/B).
//   * operands swapped on the matrix core: D[i = n][j = m] = sum_k W[n,k] * A[m,k].  The
//     C/D fragment then holds 4 CONSECUTIVE n for one m per lane -> bias / residual / output
//     move as one 16-byte (fp32) or 8-byte (16-bit) vector per fragment.
//   * global -> LDS staging by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB = 8 tile rows per
//     wave-instruction): no staging VGPRs, no ds_write pass.  THREE-stage ring (up to 156 KiB of
//     the 160 KiB LDS), raw s_barriers, COUNTED vmcnt: while tile t is multiplied
//     the DMAs of tiles t+1 and t+2 stay in flight across the barrier (~100 KiB in flight per CU,
//     what Little's law asks for at ~1 us of loaded L2/MALL latency).  v1 (register staging,
//     2 stages) was LDS-write bound, v2 (DMA, 2 stages, vmcnt(0) + __syncthreads) spent 40-50 %
//     of its wave cycles parked at the wait (profiles/).
//     Out-of-range rows (conv padding, M/N tails) read a 16-byte zero line instead.
//   * LDS tiles are [rows][BK] 16-bit (128 or 64 B / row).  The DMA destination is lane-linear, so
//     the bank swizzle (swz_key below) is applied on the per-lane SOURCE address and again on the
//     fragment reads: ds_read_b128 of a fragment (16 rows x one chunk per 16-lane group) is
//     bank-conflict free (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * XCD-aware tile order: block b runs on XCD b % 8; tiles are renumbered so that each XCD works
//     on a contiguous run of tiles (same A rows, consecutive W panels) -> its private L2 sees reuse.
//   * split-K for launches with too few tiles to fill 256 CUs (the 4x7 / 8x14 levels of the UNet):
//     partial fp32 tiles go to a caller workspace and a small second kernel reduces them in a
//     fixed order (deterministic) and applies the epilogue.
//   * fp32 accumulate; epilogue in fp32: + bias + per-image row-bias (time embedding)
//     + fp32 residual, optional GEGLU gate, fp32 or 16-bit store, optional per-64-row-slab column
//     statistics for the GroupNorm that consumes the output.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

// panelgemm.hip: the W-panel-resident shape of the short-K (K = 320) linears; vgen_panel_bn() = its column-panel width for
// a launch it takes, 0 for every other launch
int vgen_panel_bn(const vgen_tapgemm_args& a);
int vgen_panel_launch(const vgen_tapgemm_args& a, hipStream_t s);

namespace {

// Two block shapes of ONE kernel template (BM x BN tile, BK K-elements per stage, WM x WN waves):
//   "pp"   256 x BN, BK = 64, 4x2 waves (512 threads), 3-stage ring, ONE block per CU, ping-pong wave
//          groups (below) — the long-K shape: convs, big linears.
//   "dual" 256 x BN, BK = 32, 2x2 waves (256 threads, wave tile 128 x BN/2), 3-stage ring of 24-26 KiB
//          stages, TWO blocks per CU (<= 80 KiB LDS, <= 256 VGPRs each).  The two blocks are not
//          barrier-coupled: one block's prologue (index math, first DMA latency), fragment reads and
//          epilogue (GELU, residual traffic, stores) run under the other's MFMAs.  The short-K
//          linears of the transformer blocks (5-20 K-steps per tile) spend 60-80 % of a "pp" tile in
//          exactly those phases (ablation in profiles/).  The 128-row wave tile also reads 25 % fewer
//          LDS bytes per MFMA than the 64 x 64 one.
struct RowState {
  int base;  // LINEAR/TEMPORAL: source row m; CONV: img * Hi * Wi
  int a;     // CONV: iy0 ; TEMPORAL: frame index
  int b;     // CONV: ix0
};

constexpr int INVALID = -(1 << 24);

// Source of padding / out-of-range rows: a zero REGION long enough for a pointer to walk a whole K extent through it
// (K * 2 bytes — twice that on the W side of a dual-W launch — <= 256 KiB - 128, checked on the host), so every DMA
// source pointer advances by the same ROW_BYTES per K-tile whether its row is live or not — no per-piece increment
// registers (7 VGPRs on the dual shape, which sat at 256 VGPRs with 3 spills).  256 KiB: K <= 131008 for a plain
// launch, <= 65504 with dual-W — the VAE's P.V product over a 90 x 160 latent (K = 14400) and the widest decoder
// conv of the UNet (K = 9 * 2560, dual-W) fit with room (r02's 64 KiB bound did not: ADVICE r02).
constexpr int ZERO_BYTES = 262144;
__device__ __attribute__((aligned(128))) unsigned char g_zeros[ZERO_BYTES];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
  // 64 lanes x 16 B -> LDS [lds_wave_base + lane*16]; the LDS base must be wave-uniform
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}


template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS rows are BK 16-bit elements = CPR 16-byte chunks.  The DMA destination is lane-linear, so the
// bank swizzle is applied to the per-lane SOURCE chunk and again on the fragment reads.
//   CPR = 8 (128-B rows): chunk ^ (row & 7)
//   CPR = 4 ( 64-B rows): chunk ^ f[(row >> 2) & 3], f = {0, 3, 2, 1} — with ds_read_b128's 16-lane
//           groups {0-3, 12-15, 20-27}, ... every group then covers all 64 banks exactly once.
template <int CPR>
__device__ __forceinline__ int swz_key(int row) {
  if constexpr (CPR == 8) {
    return row & 7;
  } else {
    const int q = (row >> 2) & 3;
    return (4 - q) & 3;
  }
}

// Phase-ablation switches (skip the K loop / the stores / force tile (0,0)) exist only in the TUNING build of this
// file (-DVGEN_TUNING, tools/ build a separate libvgen_hip_tuning.so): in the product library `ablate` is the
// constant 0, the branches below fold away and no environment variable can make a launch skip its work.
#ifdef VGEN_TUNING
#define VGEN_ABLATE_ARG(x) (x)
#else
#define VGEN_ABLATE_ARG(x) 0
#endif

// DW ("dual-W", vgen_tapgemm_args.dualw): the weight operand carries, for every 64-element K-tile, the tile of W_hi
// followed by the tile of W_lo = round16(W - W_hi); the launch computes A . (W_hi + W_lo)^T — the high-precision mode
// of the models — with every A K-tile staged and read from LDS ONCE: an even K-step is an ordinary ping-pong step
// (A + W_hi), the odd step that follows stages and reads only the W_lo tile and multiplies it with the A fragments
// still sitting in registers.  r02 ran this product as a K-doubled launch ([A | A] x [W_hi | W_lo], every A tile
// gathered, DMA'd and ds_read twice) or, for tap gathers, as two launches through an fp32 temporary.
template <typename T, int BM, int BN, int BK, int WM, int WN, int STAGES, bool PP, bool DW>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN == 4) ? 2 : 1) void tapgemm_kernel(
    const vgen_tapgemm_args p, const int splitk, float* __restrict__ ws, const int ablate_arg, const int stagger,
    const int first_round) {
  static_assert(!PP || (WM * WN == 8 && STAGES == 3), "ping-pong needs 8 waves and a 3-stage ring");
  static_assert(!DW || (PP && BK == 64), "dual-W K-steps are built on the ping-pong schedule, 64-element K-tiles");
  const int ablate = VGEN_ABLATE_ARG(ablate_arg);
  constexpr int WPA = DW ? 2 : 1;               // W K-tiles per A K-tile
  constexpr int NT = WM * WN * 64;              // threads
  constexpr int ROW_BYTES = BK * 2;             // bytes per LDS tile row
  constexpr int CPR = ROW_BYTES / 16;           // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;                 // tile rows written by one DMA wave-instruction (1 KiB)
  constexpr int RPP = NT / CPR;                 // tile rows covered by one DMA pass of the block
  constexpr int KS = BK / 32;                   // MFMA k-steps per K-tile
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int MF = WTM / 16, NF = WTN / 16;   // fragments per wave
  constexpr int RA = BM / RPP;                  // A DMA passes per tile (4)
  constexpr int RBF = BN / RPP;                 // full W passes
  constexpr int RBT = BN % RPP;                 // tail rows of W: only waves with wave*RPI < RBT issue
  constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
  constexpr int LPT = RA + RBF;                 // piece slots per wave per tile without the W tail (+1 with it)
  static_assert(BM % RPP == 0 && RBT % RPI == 0, "tile rows must split into whole DMA instructions");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int lr = lane & 15;  // row within a 16-row fragment
  const int lq = lane >> 4;  // 16-lane group: k-chunk (operands) / 4-row group (C/D)
  const bool w_tail = RBT > 0 && wave * RPI < RBT;

  // ---- start stagger (r06) -------------------------------------------------------------------------------------------
  // Every block of a launch has the same work, so all CUs reach their epilogues together: the chip's write path (5 - 6.7 TB/s
  // for all 256 CUs, tools/probes/vmem_probe.hip) sees the launch's stores as one burst per round of tiles while the matrix
  // pipes idle, then nothing for a K loop (profiles/r02_tapgemm_ablation.json: a launch without its stores is 20 - 37 %
  // shorter).  Half of the FIRST-round blocks therefore start `stagger` x 1024 cycles late — later rounds inherit the phase
  // of the slot they fill — so that one half's stores drain under the other half's MFMAs: blocks on odd CUs for the
  // one-block-per-CU shapes, the block in the odd wave slot of its SIMD for the two-blocks-per-CU ones (HW_REG_HW_ID:
  // wave_id [3:0], cu_id [11:8]).  Wave 0 decides and sleeps; the others meet it at the first barrier.
  if (stagger > 0 && (int)(blockIdx.x + gridDim.x * blockIdx.y) < first_round && wave == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | ((WM * WN == 8 ? 8 : 0) << 6) | 4);   // 4 bits of cu_id / wave_id
    if (hw & 1u)
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }

  // ---- XCD-aware tile renumbering (bijective for any grid size) ----------------------------
  const int tiles_n = (p.N + BN - 1) / BN;
  int64_t tile;
  {
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int64_t m0 = (tile / tiles_n) * BM;
  const int n0 = (int)(tile % tiles_n) * BN;
  // `ablate` bit 3 (tuning switch, see below): every block stages tile (0, 0)'s operands — all DMA traffic hits the L2,
  // the K loop is otherwise unchanged: separates fabric / L2-miss bandwidth from what the CU itself can sustain
  const int64_t lm0 = (ablate & 8) ? 0 : m0;
  const int ln0 = (ablate & 8) ? 0 : n0;

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ A2 = (const uint16_t*)p.A2;
  const uint16_t* __restrict__ W = (const uint16_t*)p.W;
  const int64_t ldw = p.ldw ? p.ldw : ((int64_t)p.taps * p.C1 + p.C2) * WPA;

  // ---- per-thread DMA assignment: LDS chunk position ld_c of tile rows ld_r + RPP*i -----------
  const int ld_c = tid % CPR;
  const int ld_r = tid / CPR;

  RowState rs[RA];
  unsigned mvalid = 0;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    // M < 2^31 is enforced by the host wrapper: 32-bit index math (64-bit divides cost ~1 us/block)
    const unsigned m = (unsigned)lm0 + ld_r + RPP * i;
    const bool ok = m < (unsigned)p.M;
    if (ok) mvalid |= 1u << i;
    if (p.mode == VGEN_TAP_CONV3X3) {
      const unsigned hw = p.Ho * p.Wo;
      const unsigned img = m / hw;
      const unsigned rem = m - img * hw;
      const unsigned oy = rem / (unsigned)p.Wo;
      const unsigned ox = rem - oy * p.Wo;
      rs[i].base = img * p.Hi * p.Wi;
      rs[i].a = ok ? (int)oy * p.stride - p.pad_t : INVALID;
      rs[i].b = (int)ox * p.stride - p.pad_l;
    } else if (p.mode == VGEN_TAP_TEMPORAL3) {
      const unsigned fs = m / (unsigned)p.S;  // global frame index b*F + f
      rs[i].base = (int)m;
      rs[i].a = ok ? (int)(fs % (unsigned)p.F) : INVALID;
      rs[i].b = 0;
    } else {
      rs[i].base = ok ? (int)m : -1;
      rs[i].a = 0;
      rs[i].b = 0;
    }
  }

  const int cpt1 = p.C1 / BK;           // K-tiles per tap of segment 1
  const int T1 = p.taps * cpt1;
  const int KT = T1 + p.C2 / BK;        // K-tiles of the A side (the W side has WPA per A tile)
  // this block's K-tile range (split-K: blockIdx.y)
  const int split = blockIdx.y;
  const int kt_begin = (int)(((int64_t)KT * split) / splitk);
  const int kt_end = (int)(((int64_t)KT * (split + 1)) / splitk);
  // `ablate` (tuning switch VGEN_TAPGEMM_ABLATE, 0 in production): bit 0 skips the K loop, bit 1 the epilogue's
  // stores — the phase decomposition of a launch (profiles/r02_tapgemm_ablation.json); bit 2 takes the 8-byte
  // store path for 16-bit outputs (A/B of the paired 16-byte stores); bit 3 see lm0 above
  // (profiles/r02_tapgemm_l2_ablation.json: the K loop is not fabric-bound)
  const int nk = (ablate & 1) ? 0 : (kt_end - kt_begin) * WPA;   // K-steps = W K-tiles

  // ---- incremental per-lane DMA source pointers ---------------------------------------------
  // pc[j] = source of piece j for the NEXT K-tile to issue; consecutive K-tiles inside one tap /
  // K segment just advance by ROW_BYTES (0 for rows that read the zero line), so the steady-state
  // cost is one 64-bit add per piece; the row gather is recomputed only when the tap changes.
  // (v3a recomputed rows, bounds and 64-bit products every K-tile: ~700 ALU instructions per wave
  // per K-tile against 32 MFMAs — the loop was issue-bound, not memory- or MFMA-bound.)
  constexpr int NP = LPT + (RBT > 0 ? 1 : 0);
  const char* pc[NP];
  const char* const zline = (const char*)g_zeros;
  const int src_cb = (ld_c ^ swz_key<CPR>(ld_r)) * 16;   // byte offset of this lane's source chunk
  const int wave_row0 = wave * RPI;              // first tile row written by this wave's DMA (+RPP*i)
  int kt_next = kt_begin;                        // K-tile the pointers currently describe
  int left;                                      // K-tiles until the A pointers must be regathered

  // Branch-free regather (r03).  The first version compiled `ok ? A + row * lda : zline` with 64-bit row / stride
  // products into a diamond per piece (350 instructions in ~25 basic blocks, exec-mask branches): the s_memtime probe
  // (r03's in-kernel s_memtime probe, profiles/r03l_kstep_stamps_base.json) put the pointer step of a 3x3 conv at ~300 cycles per
  // K-step averaged over the tap's K-tiles — on the matrix-phase side of the ping-pong, i.e. on its critical path.  Now:
  // 32-bit source row (validated on the host), one v_mad_u64_u32 per piece, predication instead of branches (~150
  // instructions): whole step -1.2 / -1.3 % (mixed / single-pass) in a same-box A/B, profiles/r03m_ab_gather.jsonl.
  const unsigned lda_b = (unsigned)p.lda * 2u, lda2_b = (unsigned)p.lda2 * 2u;     // row strides in bytes (< 2^31)
  const char* const zl = zline + src_cb;
  auto gather_a = [&](int kt) __attribute__((always_inline)) {   // (re)compute pc[0..RA) for K-tile kt
    if (kt < T1) {
      const int tap = kt / cpt1;
      const int cch = kt - tap * cpt1;
      left = cpt1 - cch;
      const char* const a_cb = (const char*)A + (cch * BK * 2 + src_cb);
      if (p.mode == VGEN_TAP_CONV3X3) {
        const int d0 = tap / 3, d1 = tap - 3 * d0;
        const unsigned Hv = (unsigned)((p.Hi << p.ups) - 2 * p.crop_t), Wv = (unsigned)(p.Wi << p.ups);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const int iy = rs[i].a + d0, ix = rs[i].b + d1;
          const bool ok = ((unsigned)iy < Hv) & ((unsigned)ix < Wv);
          const unsigned row = (unsigned)rs[i].base + (unsigned)((iy + p.crop_t) >> p.ups) * (unsigned)p.Wi + (unsigned)(ix >> p.ups);
          const char* const real = a_cb + (uint64_t)row * lda_b;
          pc[i] = ok ? real : zl;
        }
      } else if (p.mode == VGEN_TAP_TEMPORAL3) {
        const int dt_ = tap - 1;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool ok = (unsigned)(rs[i].a + dt_) < (unsigned)p.F;
          const unsigned row = (unsigned)(rs[i].base + dt_ * p.S);
          const char* const real = a_cb + (uint64_t)row * lda_b;
          pc[i] = ok ? real : zl;
        }
      } else {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool ok = rs[i].base >= 0;
          const char* const real = a_cb + (uint64_t)(unsigned)rs[i].base * lda_b;
          pc[i] = ok ? real : zl;
        }
      }
    } else {
      left = KT - kt + 1;
      const char* const a_cb = (const char*)A2 + ((kt - T1) * BK * 2 + src_cb);
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const bool ok = (mvalid >> i) & 1u;
        const char* const real = a_cb + (uint64_t)((unsigned)lm0 + ld_r + RPP * i) * lda2_b;
        pc[i] = ok ? real : zl;
      }
    }
  };
  gather_a(kt_begin);
#pragma unroll
  for (int i = 0; i < NP - RA; ++i) {
    const int n = ln0 + ld_r + RPP * i;
    const bool ok = n < p.N;
    pc[RA + i] = (ok ? (const char*)(W + (int64_t)n * ldw + (int64_t)kt_begin * WPA * BK) : zline) + src_cb;
  }
  auto advance_a = [&]() __attribute__((always_inline)) {   // A pointers -> next A K-tile
    ++kt_next;
    if (--left == 0) {
      if (kt_next < KT) gather_a(kt_next);
    } else {
#pragma unroll
      for (int i = 0; i < RA; ++i) pc[i] += ROW_BYTES;
    }
  };
  auto advance_w = [&]() __attribute__((always_inline)) {   // W pointers -> next W K-tile
#pragma unroll
    for (int i = RA; i < NP; ++i) pc[i] += ROW_BYTES;
  };
  auto advance = [&]() __attribute__((always_inline)) {
    advance_a();
    advance_w();
  };
  // LDS destination (wave-uniform) of piece j in `stage`
  auto piece_dst = [&](int stage, int j) __attribute__((always_inline)) -> unsigned char* {
    unsigned char* const base = smem + stage * STAGE_BYTES + wave_row0 * ROW_BYTES;
    return j < RA ? base + j * RPP * ROW_BYTES : base + (BM + (j - RA) * RPP) * ROW_BYTES;
  };
  auto dma = [&](int j, int stage) __attribute__((always_inline)) {   // piece j of the next K-tile -> `stage`
    glds16(pc[j], piece_dst(stage, j));
  };
  auto load_tile = [&](int stage) __attribute__((always_inline)) {   // all pieces of the next K-tile back to back
#pragma unroll
    for (int j = 0; j < LPT; ++j)
      dma(j, stage);
    if (RBT > 0 && w_tail) dma(NP - 1, stage);
    advance();
  };
  auto load_tile_w = [&](int stage) __attribute__((always_inline)) {   // dual-W odd K-tile: the W_lo pieces only
#pragma unroll
    for (int j = RA; j < LPT; ++j) dma(j, stage);
    if (RBT > 0 && w_tail) dma(NP - 1, stage);
    advance_w();
  };

  // Accumulators start from the fp32 RESIDUAL tile instead of zero: the residual loads (one HBM / MALL round trip
  // per row fragment, 4-8 of them back to back in the old epilogue because hoisting them all would need > 100
  // VGPRs) are issued here, land straight in the accumulator registers while the first operand tiles are in
  // flight, and the epilogue is left with arithmetic and stores only.  VMEM loads return in order, so the counted
  // vmcnt that publishes K-tile 0 also covers them.  (Not with split-K — the reducer adds the residual — nor with
  // GEGLU, whose residual is added after the gate.)
  const bool geglu = p.epilogue == VGEN_EPI_GEGLU;
  const int n_out = geglu ? p.N / 2 : p.N;
  const bool vec = ((n_out & 3) == 0) && ((p.ldo & 3) == 0) &&
                   (p.residual == nullptr || (p.ldr & 3) == 0) &&
                   (p.rowbias == nullptr || (p.rowbias_ld & 3) == 0);
  const bool res_folded = p.residual != nullptr && splitk == 1 && vec && !geglu;
  f32x4 acc[NF][MF];
#pragma unroll
  for (int ni = 0; ni < NF; ++ni)
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (res_folded) {
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const int64_t m = m0 + wm * WTM + mi * 16 + lr;
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int n = n0 + wn * WTN + ni * 16 + lq * 4;
        if (m < p.M && n < p.N) acc[ni][mi] = *(const f32x4*)(p.residual + m * p.ldr + n);
      }
    }
  }

  // fragment read offsets: row (frag*16 + lr), chunk (ks*4 + lq) ^ key(lr)
  const int rd_row = lr * ROW_BYTES;
  const int sw = swz_key<CPR>(lr);

  // A fragments are read in MH passes of MFH row fragments (the 128-row wave tile of the dual shape
  // would otherwise hold 32 + 16 operand registers on top of 128 accumulators and spill)
  constexpr int MH = MF > 4 ? MF / 4 : 1;
  constexpr int MFH = MF / MH;
  u32x4 wf[KS][NF], xf[KS][MFH];
  auto read_w = [&](int stage) __attribute__((always_inline)) {
    const unsigned char* b = smem + stage * STAGE_BYTES + BM * ROW_BYTES + wn * WTN * ROW_BYTES + rd_row;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int co = ((ks * 4 + lq) ^ sw) << 4;
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) wf[ks][ni] = *(const u32x4*)(b + ni * 16 * ROW_BYTES + co);
    }
  };
  auto read_x = [&](int stage, int mh) __attribute__((always_inline)) {
    const unsigned char* a = smem + stage * STAGE_BYTES + (wm * WTM + mh * MFH * 16) * ROW_BYTES + rd_row;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int co = ((ks * 4 + lq) ^ sw) << 4;
#pragma unroll
      for (int mi = 0; mi < MFH; ++mi) xf[ks][mi] = *(const u32x4*)(a + mi * 16 * ROW_BYTES + co);
    }
  };

  // Lock-step K-step (dual shape): multiply one staged K-tile and, interleaved with the MFMAs, issue
  // the DMA pieces of the tile two steps ahead.  An LDS-DMA instruction occupies the CU's single
  // texture-address path for ~16-25 cycles (1 KiB at <= 64 B/clk) and stalls the issuing wave ~100
  // cycles; issued in one burst right after the barrier (v3a) the DMA phase and the MFMA phase
  // simply added up.  Spread one piece per MFMA group, the other waves on the SIMD keep the matrix
  // pipe busy while this wave waits on the address path.
  auto compute = [&](int stage, auto prefetch_tag, int stage_pf) __attribute__((always_inline)) {
    constexpr bool prefetch = decltype(prefetch_tag)::value;   // compile-time: branch-free K-step body
    constexpr int NMF = KS * NF * MF;                // MFMAs per K-tile per wave
    constexpr int NMH = NMF / MH;
    constexpr int GRP = NMF / NP;                    // MFMAs between two DMA pieces (>= NP slots)
    static_assert(GRP >= 1 && (NMF + GRP - 1) / GRP >= NP, "not enough MFMA groups for the DMA pieces");
    read_w(stage);
    int piece = 0;
#pragma unroll
    for (int mh = 0; mh < MH; ++mh) {
      read_x(stage, mh);
#pragma unroll
      for (int j = 0; j < NMH; ++j) {
        const int i = mh * NMH + j;
        const int ks = j / (NF * MFH), r = j % (NF * MFH), ni = r / MFH, mi = r % MFH;
        if (i % GRP == 0) {
          if (prefetch) {
            if (piece < LPT) {
              dma(piece, stage_pf);
            } else if (RBT > 0 && piece == NP - 1 && w_tail) {
              dma(NP - 1, stage_pf);
            }
          }
          ++piece;
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[ni][mh * MFH + mi] = T::mfma32(wf[ks][ni], xf[ks][mi], acc[ni][mh * MFH + mi]);
      }
      if (MH > 1) __builtin_amdgcn_sched_barrier(0);   // the next pass reuses xf: keep its reads below
    }
    if (prefetch) advance();
  };

  // ---- ping-pong schedule (PP) -----------------------------------------------------------------
  // Waves w and w+4 share a SIMD.  With one barrier per K-tile both sit in the same phase: they read
  // fragments together (matrix pipe idle), then fight for the pipe together — SQ counters of the
  // lock-step 8-wave loop: 38 % of wave cycles parked at s_waitcnt/s_barrier, MFMA pipe 48 % busy.
  // Here every K-tile is split into an R phase (16 fragment ds_reads + the 6 DMA pieces of tile t+2
  // + the counted vmcnt for tile t+1) and an M phase (32 back-to-back MFMAs out of registers), each
  // closed by a barrier, and waves 4-7 run ONE PHASE BEHIND waves 0-3 (one extra barrier up front,
  // one at the end for the leaders): on every SIMD one wave multiplies while its partner reads.
  //   hazards: DMA(t+2) overwrites the stage of tile t-1, last read in R(t-1) of both groups, i.e.
  //   before the barrier that precedes the earlier group's R(t);  tile t+1 is complete once every
  //   wave passed the vmcnt at the end of its R(t), which for both groups is before any R(t+1).
  // `odd_tag` (dual-W only): the odd K-step of a pair reads the W_lo fragments and issues the W pieces of the next odd
  // tile — the A fragments of the even step stay in their registers, the A region of the odd stages stays unused.
  auto read_phase = [&](int stage, auto prefetch_tag, int stage_pf, auto odd_tag) __attribute__((always_inline)) {
    constexpr bool prefetch = decltype(prefetch_tag)::value;
    constexpr bool odd = decltype(odd_tag)::value;
    static_assert(!PP || MH == 1, "ping-pong keeps a whole K-tile of fragments in registers");
    // fragment reads and DMA issues interleaved: a DMA instruction parks the wave on the CU's texture-address path
    // while the LDS serves the reads issued just before it (issued as two blocks, reads then DMAs, the two phases
    // simply added up: profiles/r02_tapgemm_rphase_ablation.json).  conv 57344x320x2880 144.8 -> 135.7 us, linear
    // 57344x2560x2560 781 -> 740 us; whole step -0.8 % in same-box A/Bs (34.35 -> 34.05 ms, 30.83 -> 30.59 ms);
    // 2 or 3 reads per DMA, DMA leading or trailing its group: all within 0.2 %.
    const unsigned char* bw = smem + stage * STAGE_BYTES + BM * ROW_BYTES + wn * WTN * ROW_BYTES + rd_row;
    const unsigned char* bx = smem + stage * STAGE_BYTES + (wm * WTM) * ROW_BYTES + rd_row;
    constexpr int P0 = odd ? RA : 0;                   // first DMA piece this phase issues
    constexpr int NPI = NP - P0;
    constexpr int NRF = odd ? NF : NF + MF;            // fragment reads per k-step
    constexpr int NR = KS * NRF;                       // fragment reads per wave and K-tile
    constexpr int EVERY = (NR + NPI) / (NPI + 1) > 0 ? (NR + NPI) / (NPI + 1) : 1;   // reads between two DMA issues (3)
    int piece = P0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int ks = r / NRF, q = r % NRF;
      const int co = ((ks * 4 + lq) ^ sw) << 4;
      if (q < NF) wf[ks][q] = *(const u32x4*)(bw + q * 16 * ROW_BYTES + co);
      else xf[ks][q - NF] = *(const u32x4*)(bx + (q - NF) * 16 * ROW_BYTES + co);
      if (prefetch && (r % EVERY) == EVERY - 1 && piece < NP) {
        __builtin_amdgcn_sched_barrier(0);
        if (piece < LPT) {
          dma(piece, stage_pf);
        } else if (RBT > 0 && w_tail) {
          dma(NP - 1, stage_pf);
        }
        ++piece;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (prefetch) {
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (j >= piece) {
          if (j < LPT) {
            dma(j, stage_pf);
          } else if (RBT > 0 && w_tail) {
            dma(NP - 1, stage_pf);
          }
        }
    }
  };
  auto mfma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < MF; ++mi) acc[ni][mi] = T::mfma32(wf[ks][ni], xf[ks][mi], acc[ni][mi]);
  };

  // ---- main loop: STAGES-deep DMA ring, counted vmcnt -------------------------------------------
  // Two sequential loops — `nk - AHEAD` K-steps that also issue the DMA of tile it+AHEAD, then the
  // last AHEAD steps without — rather than one loop with an if/else around two bodies: the diamond
  // made hipcc give every accumulator a second register for the join (2 x 128 on the dual shape =
  // spills to scratch inside the K loop).
  constexpr int AHEAD = STAGES - 1;
  if constexpr (DW) {
    if (nk > 0) {
      load_tile(0);
      load_tile_w(1);
    }
  } else {
    for (int i = 0; i < AHEAD; ++i)
      if (nk > i) load_tile(i);
  }
  int st_c = 0, st_l = AHEAD;   // stage to compute / stage to load into
  const int nk_main = nk > AHEAD ? nk - AHEAD : 0;
  auto rotate = [&]() __attribute__((always_inline)) {
    st_c = st_c == STAGES - 1 ? 0 : st_c + 1;
    st_l = st_l == STAGES - 1 ? 0 : st_l + 1;
  };
  auto wait_next = [&]() __attribute__((always_inline)) {   // all but the newest tile's pieces have landed
    if (w_tail) wait_vmcnt<LPT + 1>();                      // pieces this wave issues per tile (+ the W tail it is in)
    else wait_vmcnt<LPT>();
  };
  auto wait_next_w = [&]() __attribute__((always_inline)) {   // ... when the newest tile is a dual-W odd one (W pieces only)
    if (w_tail) wait_vmcnt<LPT - RA + 1>();
    else wait_vmcnt<LPT - RA>();
  };
  if constexpr (PP && DW) {
    // K-steps come in (even, odd) pairs: nk is even.  Tile t+2 has the parity of tile t, so an even step issues (and
    // leaves in flight) a full tile, an odd step the W pieces only; each waits for the tile of the OTHER parity.
    const bool follower = wave >= 4;
    if (nk > 0) {
      wait_next_w();                                    // tile 0 landed, tile 1 (W_lo of the first pair) may fly
      __builtin_amdgcn_s_barrier();
      if (follower) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      auto dw_step = [&](auto prefetch_tag, auto odd_tag) __attribute__((always_inline)) {
        constexpr bool pf = decltype(prefetch_tag)::value;
        constexpr bool odd = decltype(odd_tag)::value;
        read_phase(st_c, prefetch_tag, st_l, odd_tag);
        if constexpr (pf) {
          if constexpr (odd) wait_next_w();             // tile it+1 (even: full) landed, it+2 (odd: W only) may fly
          else wait_next();                             // tile it+1 (odd) landed, it+2 (even: full) may fly
        } else {
          wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (pf) {
          if constexpr (!odd) advance_a();
          advance_w();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rotate();
      };
      for (int it = 0; it + 2 < nk; it += 2) {
        dw_step(std::true_type{}, std::false_type{});
        dw_step(std::true_type{}, std::true_type{});
      }
      dw_step(std::false_type{}, std::false_type{});
      dw_step(std::false_type{}, std::true_type{});
      if (!follower) __builtin_amdgcn_s_barrier();
    }
  } else if constexpr (PP) {
    const bool follower = wave >= 4;
    if (nk > 1) wait_next();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                       // tile 0 published
    if (follower) __builtin_amdgcn_s_barrier();         // run one phase behind
    asm volatile("" ::: "memory");
    auto pp_step = [&](auto prefetch_tag, bool more) __attribute__((always_inline)) {
      read_phase(st_c, prefetch_tag, st_l, std::false_type{});
      // tile it+1 (issued one iteration ago) must have landed before the NEXT read phase of anyone
      if (more) wait_next();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mfma_phase();
      __builtin_amdgcn_sched_barrier(0);
      // the pointer step of this wave's next DMA issue runs here, behind the queued MFMAs: a K-step lasts two READ
      // phases (the matrix phase of one wave group hides under the read phase of the other), so VALU work moved
      // out of the read phase shortens the step twice over
      if (decltype(prefetch_tag)::value) advance();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      rotate();
    };
    for (int it = 0; it < nk_main; ++it) pp_step(std::true_type{}, true);
    for (int it = nk_main; it < nk; ++it) pp_step(std::false_type{}, false);
    if (!follower) __builtin_amdgcn_s_barrier();
  } else {
    // Iteration `it`: wait until tile `it` has landed (the DMA of tile it+1 may stay in flight),
    // barrier (publishes tile `it` of every wave AND proves every wave finished reading the stage
    // that the next DMA overwrites, last read in iteration it-1), issue tile it+AHEAD interleaved
    // with the MFMAs of tile it.
    for (int it = 0; it < nk_main; ++it) {
      if (STAGES == 3) wait_next();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      compute(st_c, std::true_type{}, st_l);
      rotate();
    }
    for (int it = nk_main; it < nk; ++it) {
      if (STAGES == 3 && it + 1 < nk) wait_next();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      compute(st_c, std::false_type{}, st_l);
      rotate();
    }
  }

  if (splitk > 1) {   // raw fp32 partial tile -> workspace [split][M][N]; epilogue in the reducer
    float* const wsp = ws + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const int64_t m = m0 + wm * WTM + mi * 16 + lr;
      if (m >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int n = n0 + wn * WTN + ni * 16 + lq * 4;
        if (n < p.N) *(f32x4*)(wsp + m * p.N + n) = acc[ni][mi];
      }
    }
    return;
  }

  // ---- epilogue -------------------------------------------------------------------------
  if (ablate & 2) {   // keep the accumulators live (a store that never happens), skip everything else
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
      for (int mi = 0; mi < MF; ++mi) t += acc[ni][mi];
    if (t.x + t.y + t.z + t.w == 1.2345e30f) ((float*)p.out)[0] = t.x;
    return;
  }
  float* const of = (float*)p.out;
  uint16_t* const oh = (uint16_t*)p.out;

  // bias vectors of this lane's columns: loaded once per tile, not once per row fragment
  f32x4 bv[NF];
  if (vec || geglu) {
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
      const int n = n0 + wn * WTN + ni * 16 + lq * 4;
      bv[ni] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- 16-bit outputs: their own epilogue loop (no column statistics live here: the dual shape has no registers
  // to spare), two column fragments per store ----------------------------------------------------------------
  if (vec && p.out_dtype != VGEN_F32 && (p.ldo & 7) == 0 && (!geglu || NF % 4 == 0) && !(ablate & 4)) {
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      __builtin_amdgcn_sched_barrier(0);
      const int64_t m = m0 + wm * WTM + mi * 16 + lr;
      if (m >= p.M) continue;
      const float* rbp = p.rowbias ? p.rowbias + (m / p.rows_per_rb) * p.rowbias_ld : nullptr;
    // 16-bit outputs, two column fragments at a time.  A lane's C fragment is 4 consecutive n = 8 bytes: stored
    // directly, a wave-level store covers 16 rows x 32 contiguous bytes per fragment and the write path runs at half
    // its rate (ablation, profiles/r02_tapgemm_ablation.json: the epilogue of the 57344 x 960 x 320 qkv GEMM — 110 MB
    // of stores — took 47 us = 2.3 TB/s and did not overlap the K loop; the fp32-output o-proj stores 16 B per lane
    // at 4.7 TB/s).  v_permlane16_swap_b32 (gfx950) exchanges the odd 16-lane rows of one register with the even
    // rows of another: after swapping the packed dwords of fragments (A, B) lane-row pairs, even rows hold 8
    // consecutive n of fragment A and odd rows 8 consecutive n of fragment B — one 16-byte store per lane, 64
    // contiguous bytes per output row per instruction.  (r01 staged such stores through LDS: the extra barrier + LDS
    // round trip cost more than it won; this exchange is two register-only instructions per fragment pair.)
    auto store_pair16 = [&](int frag0_col, bool pair_ok, const f32x4& va, const f32x4& vb, int64_t m_) __attribute__((always_inline)) {
      // frag0_col: column of fragment A's n = 0; fragment B starts 16 columns later
      u32x2 pa = pack4<T>(va.x, va.y, va.z, va.w);
      u32x2 pb = pack4<T>(vb.x, vb.y, vb.z, vb.w);
      if (pair_ok) {
        const auto sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
        const u32x4 q = {sx[0], sy[0], sx[1], sy[1]};
        *(u32x4*)(oh + m_ * p.ldo + frag0_col + (lq & 1) * 16 + (lq >> 1) * 8) = q;
      } else {
        if (frag0_col + lq * 4 < n_out) *(u32x2*)(oh + m_ * p.ldo + frag0_col + lq * 4) = pa;
        if (frag0_col + 16 + lq * 4 < n_out) *(u32x2*)(oh + m_ * p.ldo + frag0_col + 16 + lq * 4) = pb;
      }
    };
      if (!geglu) {
        auto final_val = [&](int ni) __attribute__((always_inline)) -> f32x4 {      // bias / row-bias / residual
          const int n = n0 + wn * WTN + ni * 16 + lq * 4;
          f32x4 v = acc[ni][mi];
          if (n < p.N) {
            v += bv[ni];
            if (rbp) v += *(const f32x4*)(rbp + n);
            if (p.residual && !res_folded) v += *(const f32x4*)(p.residual + m * p.ldr + n);
          }
          return v;
        };
        // split_out (two-term activation rows, vgen_tapgemm_args.split_out): column n holds hi = round16(v), column
        // N + n holds round16(v - hi) — the value a separate vgen_cast_split pass over an fp32 output would have written
        auto lo_of = [&](const f32x4& v) __attribute__((always_inline)) -> f32x4 {
          const u32x2 h = pack4<T>(v.x, v.y, v.z, v.w);
          return f32x4{v.x - T::to_f32((uint16_t)(h.x & 0xffffu)), v.y - T::to_f32((uint16_t)(h.x >> 16)),
                       v.z - T::to_f32((uint16_t)(h.y & 0xffffu)), v.w - T::to_f32((uint16_t)(h.y >> 16))};
        };
#pragma unroll
        for (int ni = 0; ni + 1 < NF; ni += 2) {
          const int c0 = n0 + wn * WTN + ni * 16;
          const f32x4 va = final_val(ni), vb = final_val(ni + 1);
          store_pair16(c0, c0 + 32 <= p.N, va, vb, m);
          if (p.split_out) store_pair16(p.N + c0, c0 + 32 <= p.N, lo_of(va), lo_of(vb), m);
        }
        if constexpr (NF % 2 == 1) {
          const int n = n0 + wn * WTN + (NF - 1) * 16 + lq * 4;
          if (n < p.N) {
            const f32x4 v = final_val(NF - 1);
            *(u32x2*)(oh + m * p.ldo + n) = pack4<T>(v.x, v.y, v.z, v.w);
            if (p.split_out) {
              const f32x4 l = lo_of(v);
              *(u32x2*)(oh + m * p.ldo + p.N + n) = pack4<T>(l.x, l.y, l.z, l.w);
            }
          }
        }
      } else if constexpr (NF % 4 == 0) {
        auto gated = [&](int np) __attribute__((always_inline)) -> f32x4 {
          const int pn = n0 + wn * WTN + np * 32 + lq * 4;  // packed index of the value lanes
          const int j = (n0 + wn * WTN) / 2 + np * 16 + lq * 4;
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          if (pn < p.N) {
            const f32x4 val = acc[2 * np][mi] + bv[2 * np];
            const f32x4 gat = acc[2 * np + 1][mi] + bv[2 * np + 1];
            o = geglu4(val, gat);
            if (p.residual) o += *(const f32x4*)(p.residual + m * p.ldr + j);
          }
          return o;
        };
#pragma unroll
        for (int np = 0; np < NF / 2; np += 2) {
          const int j0 = (n0 + wn * WTN) / 2 + np * 16;
          store_pair16(j0, j0 + 32 <= n_out, gated(np), gated(np + 1), m);
        }
      }
    }
    return;
  }
  // "pp256" (NF = 8, 64 x 128 wave tiles): planned for 16-bit outputs only (legal() below) — the fp32 / column-statistics
  // epilogue is not instantiated for it (its cs_s / cs_q / bias registers next to 128 accumulators spilled 320 VGPRs)
  if constexpr (NF > 5) return;
  // column statistics of the final fp32 values per 64-row slab (vgen_tapgemm_args.colstats): the wave
  // tile is 64 ("pp") or 128 ("dual") rows = 1 or 2 whole slabs, so no cross-wave step is needed.
  // Each lane keeps partials over the 4 row fragments of a slab; the 16 lanes that hold different rows
  // of the same 4 columns are then folded through the (now idle) stage LDS: [result][16 lanes] floats
  // per wave, every lane sums one result row in a fixed order and stores it — 8 ds_write_b32 per
  // fragment and 4 ds_read_b128 per result instead of a 4-step butterfly over 8 x NF registers
  // (the ds_bpermute butterfly cost as much as the statistics pass it replaces).
  const bool do_cs = p.colstats != nullptr && vec && !geglu;
  f32x4 cs_s[NF], cs_q[NF];
#pragma unroll
  for (int ni = 0; ni < NF; ++ni) cs_s[ni] = cs_q[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (do_cs) __syncthreads();   // every wave is done reading the operand stages
  auto flush_cs = [&](int slab) __attribute__((always_inline)) {
    constexpr int NC = NF * 16;                                        // columns of this wave
    constexpr int NRES = 2 * NC;                                       // sums | sums of squares
    float* const cb = (float*)smem + wave * (NRES * 16);
    const int64_t srow = (m0 + wm * WTM) / 64 + slab;                  // global slab index
    const bool slab_live = srow * 64 < p.M;
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cw = ni * 16 + lq * 4 + r;
        cb[cw * 16 + lr] = cs_s[ni][r];
        cb[(NC + cw) * 16 + lr] = cs_q[ni][r];
      }
      cs_s[ni] = cs_q[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // LDS operations of one wave execute in order: the reads below see the writes above
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int res = lane; res < NRES; res += 64) {
      const f32x4* q = (const f32x4*)(cb + res * 16);
      const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
      const float t = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) +
                      (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
      const int plane = res >= NC ? 1 : 0;
      const int n = n0 + wn * WTN + (res - plane * NC);
      if (slab_live && n < p.N) p.colstats[(srow * 2 + plane) * (int64_t)p.N + n] = t;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next slab overwrites cb
  };
#pragma unroll
  for (int mi = 0; mi < MF; ++mi) {
    // keep the residual / row-bias loads of one row fragment from being hoisted above the stores of
    // the previous one: with 8 row fragments the hoisted loads alone would need > 100 VGPRs
    __builtin_amdgcn_sched_barrier(0);
    const int64_t m = m0 + wm * WTM + mi * 16 + lr;
    if constexpr (MF % 4 == 0) {
      if (do_cs && mi % 4 == 0 && mi > 0) flush_cs(mi / 4 - 1);
    }
    if (m >= p.M) continue;
    const float* rbp = p.rowbias ? p.rowbias + (m / p.rows_per_rb) * p.rowbias_ld : nullptr;
    if (!geglu) {
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int n = n0 + wn * WTN + ni * 16 + lq * 4;
        if (n >= p.N) continue;
        f32x4 v = acc[ni][mi];
        if (vec) {
          v += bv[ni];
          if (rbp) v += *(const f32x4*)(rbp + n);
          if (p.residual && !res_folded) v += *(const f32x4*)(p.residual + m * p.ldr + n);
          if (do_cs) {
            cs_s[ni] += v;
            cs_q[ni] += v * v;
          }
          if (p.out_dtype == VGEN_F32) {
            *(f32x4*)(of + m * p.ldo + n) = v;
          } else {
            *(u32x2*)(oh + m * p.ldo + n) = pack4<T>(v.x, v.y, v.z, v.w);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) break;
            float s = v[r];
            if (p.bias) s += p.bias[n + r];
            if (rbp) s += rbp[n + r];
            if (p.residual) s += p.residual[m * p.ldr + n + r];
            if (p.out_dtype == VGEN_F32) of[m * p.ldo + n + r] = s;
            else oh[m * p.ldo + n + r] = T::from_f32(s);
          }
        }
      }
    } else if constexpr (NF % 2 == 0) {
      // packed columns: fragment pairs (value, gate) = (ni even, ni odd)
      auto gated = [&](int np) __attribute__((always_inline)) -> f32x4 {
        const int pn = n0 + wn * WTN + np * 32 + lq * 4;  // packed index of the value lanes
        const int j = (n0 + wn * WTN) / 2 + np * 16 + lq * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (pn < p.N) {
          const f32x4 val = acc[2 * np][mi] + bv[2 * np];
          const f32x4 gat = acc[2 * np + 1][mi] + bv[2 * np + 1];
          o = geglu4(val, gat);
          if (p.residual) o += *(const f32x4*)(p.residual + m * p.ldr + j);
        }
        return o;
      };
#pragma unroll
      for (int np = 0; np < NF / 2; ++np) {
        const int pn = n0 + wn * WTN + np * 32 + lq * 4;
        if (pn >= p.N) continue;
        const int j = (n0 + wn * WTN) / 2 + np * 16 + lq * 4;
        const f32x4 o = gated(np);
        if (p.out_dtype == VGEN_F32) {
          *(f32x4*)(of + m * p.ldo + j) = o;
        } else {
          *(u32x2*)(oh + m * p.ldo + j) = pack4<T>(o.x, o.y, o.z, o.w);
        }
      }
    }
  }
  if constexpr (MF % 4 == 0) {
    if (do_cs) flush_cs(MF / 4 - 1);
  }
}

// split-K reducer: fixed summation order over the splits, then the same epilogue as the main kernel.
// Why this is still a second launch (56 per denoise step, ~7 us each) — r05 built the reduction INTO the main kernel
// twice ("a split-K whose last-arriving block reduces in-kernel"; commits 4c202c3 and the one after):
//   v1  every thread __threadfence()s its partial stores, the block draws one agent-scope ticket per tile, the block
//       with the last ticket __threadfence()s again, sums the partials in split order and runs the ordinary epilogue.
//       Correct (150 tap-GEMM parity cases, forced dual-shape splits, 12-round repeat test, model-level tests all green,
//       profiles/r05m_pytest_tapgemm.log) and 13 % SLOWER on the whole step, 33.4 -> 37.7 ms
//       (profiles/r05m_ab_splitk_fences.jsonl): on gfx950 an agent-scope release / acquire is `buffer_wbl2 sc1` +
//       `buffer_inv sc1` — a write-back and an invalidate of the XCD's WHOLE 4 MB L2 — issued by every wave of every
//       block; the blocks still in their K loops lose their operand lines each time.
//   v2  no fences: partials stored / loaded by inline-asm `global_store/load_dwordx4 ... sc0 sc1` (or `sc1`), s_waitcnt
//       vmcnt(0) + barrier before a relaxed ticket.  WRONG — 15 of 16 split-K cases off by 3e-3 .. 7e-3 (stale or
//       not-yet-visible partial lines: the scope bits alone do not order a 16-byte data store against another XCD's
//       later load on this part; profiles/r05m2_pytest_splitk_scope_bits_FAILED.log) — and still +1.4 % on the step
//       (33.38 -> 33.85 ms, profiles/r05m2_ab_splitk_scope_bits.jsonl): one block per tile walking `splitk` dependent
//       round trips at the launch's tail costs more than a chip-wide reducer launch behind a kernel boundary, which
//       does the cache maintenance ONCE.
// Both removed; the tests they were run on stayed (tests/test_gpu_kernels.py::test_splitk_launches_are_repeatable).
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const vgen_tapgemm_args p, const int splitk,
                                                            const float* __restrict__ ws) {
  const bool geglu = p.epilogue == VGEN_EPI_GEGLU;
  const int n_out = geglu ? p.N / 2 : p.N;
  const int ng = n_out >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.M * ng) return;
  const int64_t m = idx / ng;
  const int j = (int)(idx - m * ng) * 4;
  const int64_t plane = p.M * (int64_t)p.N;
  const int pn = geglu ? 32 * (j >> 4) + (j & 15) : j;
  f32x4 v = {0, 0, 0, 0}, gt = {0, 0, 0, 0};
  for (int s = 0; s < splitk; ++s) {
    v += *(const f32x4*)(ws + s * plane + m * p.N + pn);
    if (geglu) gt += *(const f32x4*)(ws + s * plane + m * p.N + pn + 16);
  }
  if (p.bias) {
    v += *(const f32x4*)(p.bias + pn);
    if (geglu) gt += *(const f32x4*)(p.bias + pn + 16);
  }
  if (geglu) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] * gelu_erf_f(gt[r]);
  } else if (p.rowbias) {
    v += *(const f32x4*)(p.rowbias + (m / p.rows_per_rb) * p.rowbias_ld + j);
  }
  if (p.residual) v += *(const f32x4*)(p.residual + m * p.ldr + j);
  if (p.out_dtype == VGEN_F32) *(f32x4*)((float*)p.out + m * p.ldo + j) = v;
  else *(u32x2*)((uint16_t*)p.out + m * p.ldo + j) = pack4<T>(v.x, v.y, v.z, v.w);
}

// ---- launch planning ---------------------------------------------------------------------------
// 256-row tiles make tile-count quantisation expensive (280 tiles on 256 CUs = 2 rounds at 55 %
// fill), so the block shape, the column tile BN and the split-K factor are chosen together from a
// small cost model (microseconds; constants fitted to profiles/r01_*_tapgemm_shapes.json):
//   cost = rounds(tiles * s / slots) * (ceil(KT / s) * t_ktile + t_tile) + [s > 1] * reduce(s)
// with KT in 64-element K-steps, slots = 256 ("pp", one block per CU) or 512 ("dual").
// r06 shapes (never proposed by the cost model: reached through the measured plan table / forced plans only):
//   "pp256" 256 x 256, BK 32, 8 waves (64 x 128 wave tiles), ping-pong, 3 x 32 KiB ring: 128 FLOP per staged byte instead of
//           85-98 — for the N % 256 == 0 launches with 16-bit outputs (GEGLU up-projections, the K = 1280 q/k/v); 214 VGPRs.
//   "q128"  128 x BN, BK 32, 4 waves (64 x BN/2 wave tiles), lock-step K-steps, 3 x (128 + BN) x 64 B ring (48-55 KiB):
//           2 INDEPENDENT blocks per CU — for the under-filled 8 x 14 / 4 x 7 levels, where one 8-wave block per CU
//           leaves every stall of its short K loop uncovered.  (Bounded to three blocks per CU the 128 x 128 instance
//           spilled 16 VGPRs in its epilogue: two, like the dual shape.)
enum Shape { SHAPE_PP = 0, SHAPE_DUAL = 1, SHAPE_PP128 = 2, SHAPE_PANEL = 3, SHAPE_PP256 = 4, SHAPE_Q128 = 5 };
// r04: 224-row tiles (every row count of the t2v UNet is 7 * 2^k, so 256-row tiles fill the last round over the CUs at
// most 87.5 %) were built as a dual 224 x BN shape and a 224 x 320 ping-pong shape, passed every parity case they are legal
// for, and measured +0.5 % (mixed) / +1.4 % (single-pass) SLOWER on the whole step in a same-box A/B
// (profiles/r04a_ab_libs.jsonl); so was the buffer-resource LDS-DMA (+0.1 / +0.6 %).  Both were removed again (history:
// commits 34443ca, 6b9d39a).

// start stagger of the streaming shapes (see tapgemm_kernel): per cent of one tile's estimated time, by blocks per CU, and the
// fewest rounds of tiles a launch must have; measured on the whole step (profiles/r06b_ab_stagger.jsonl)
// r06 call B, one process, 8 settings x 3 interleaved rounds (profiles/r06b_ab_stagger.jsonl, r06b_ab_stagger_shapes.json):
// whole step 31.50 -> 31.35 ms (-0.5 %) at 50 / 50 / 2; per launch the 3584 x 10240 x 1280 GEGLU -7 %, the 57344 x 320 x 2880
// conv -5 %, everything else within +-1 %; 100 % of a tile +0.1 %.  Small, free, kept — and a measured answer to "are the
// epilogue stores a chip-wide burst?": mostly not (a launch without its stores is 20 - 37 % shorter, a launch whose CUs
// store out of phase 0.5 %): the store path binds per CU, under its own block's idle matrix pipe.
constexpr int STAGGER_PCT_PP = 50;
constexpr int STAGGER_PCT_DUAL = 50;
constexpr int STAGGER_MIN_ROUNDS = 2;

struct Plan {
  int shape;
  int bn;
  int splitk;
};

#ifdef VGEN_TUNING
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
#endif

// Measured plans for the launches of the reference's t2v UNet at its benchmark shape (tools/autotune_gemm.py
// times every (shape, BN, split-K) candidate per distinct launch signature on the GPU and writes this table):
// consulted before the cost model, which stays the rule for every other shape.
struct PlanEntry {
  int mode;
  int64_t M;
  int N, C1, C2, taps, epilogue, out_dtype, flags;   // flags: residual | rowbias << 1 | colstats << 2
  int shape, bn, splitk;
};
#include "tapgemm_plans.inc"

// the active table: the compiled-in one, or whatever vgen_tapgemm_set_plans installed (tools/autotune_gemm.py
// A/B-tests a candidate table inside one process before it is baked into tapgemm_plans.inc)
PlanEntry* g_plans = nullptr;
int g_nplans = -1;

Plan make_plan(const vgen_tapgemm_args& a, bool* from_table = nullptr) {
  if (from_table) *from_table = false;
  const bool geglu = a.epilogue == VGEN_EPI_GEGLU;
  const int KT = a.taps * (a.C1 / 64) + a.C2 / 64;
  const int n_out = geglu ? a.N / 2 : a.N;
  const bool vec = (a.N % 4 == 0) && (n_out % 4 == 0) && (a.ldo % 4 == 0) &&
                   (a.residual == nullptr || a.ldr % 4 == 0) &&
                   (a.rowbias == nullptr || a.rowbias_ld % 4 == 0);
  int cands[2], nc = 0;
  if (a.N % 128 == 0) cands[nc++] = 128;
  if (a.N % 160 == 0 && !geglu) cands[nc++] = 160;
  if (nc == 0) cands[nc++] = 64;
  // tuning build only (not part of the ABI): VGEN_TAPGEMM_SHAPE = 0 (pp) / 1 (dual) / 2 (pp128) forces a shape
#ifdef VGEN_TUNING
  static const int force_shape = env_int("VGEN_TAPGEMM_SHAPE", -1);
#else
  constexpr int force_shape = -1;
#endif
  const int smax = (vec && a.colstats == nullptr && !a.split_out) ? (KT / 4 < 32 ? KT / 4 : 32) : 1;
  // HBM time of the epilogue traffic (output + fp32 residual), not hidden behind MFMAs when every CU
  // runs one block in the same phase ("pp"); about half hidden with two independent blocks per CU
  const double epi_us = (double)a.M * n_out * ((a.out_dtype == VGEN_F32 ? 4 : 2) + (a.residual ? 4 : 0)) / 4.5e6;
  auto legal = [&](int shape, int bn, int sk) {
    // BN = 64 is legal for any N as a forced / tabled plan (small-M levels: more, smaller tiles instead of split-K);
    // the cost model itself only proposes it when neither 128 nor 160 divides N
    if (sk < 1 || sk > (smax < 1 ? 1 : smax)) return false;
    if (shape == SHAPE_PP256)   // its own epilogue: 16-bit outputs through the paired 16-byte stores only; with split-K the
                                // reducer launch holds the epilogue, so any output the reducer takes is legal
      return bn == 256 && a.N % 256 == 0 && vec && !a.colstats && !a.dualw && !a.split_out &&
             (sk > 1 || (a.out_dtype != VGEN_F32 && a.ldo % 8 == 0));
    bool ok = bn == 64 && a.N % 64 == 0 && (!geglu || a.N % 64 == 0);
    for (int c = 0; c < nc; ++c) ok |= cands[c] == bn;
    if (shape == SHAPE_Q128) return ok && !a.dualw;
    return ok && shape >= SHAPE_PP && shape <= SHAPE_PP128 &&
           !(a.colstats && shape == SHAPE_PP128) && !(a.dualw && shape == SHAPE_DUAL);
  };
#ifdef VGEN_TUNING
  // tuning build only: VGEN_TAPGEMM_PLAN="shape,bn,splitk" forces one plan (read on every call: the autotuner flips
  // it between launches); VGEN_TAPGEMM_TABLE=0 ignores the measured table
  if (const char* fp = getenv("VGEN_TAPGEMM_PLAN")) {
    int sh = -1, bn = 0, sk = 0;
    if (sscanf(fp, "%d,%d,%d", &sh, &bn, &sk) == 3 && legal(sh, bn, sk)) return Plan{sh, bn, sk};
  }
  static const bool use_table = env_int("VGEN_TAPGEMM_TABLE", 1) != 0;
#else
  constexpr bool use_table = true;
#endif
  if (use_table && force_shape < 0 && !a.dualw) {
    const int flags = (a.residual ? 1 : 0) | (a.rowbias ? 2 : 0) | (a.colstats ? 4 : 0);
    const PlanEntry* tab = g_nplans >= 0 ? g_plans : kPlans;
    const int ntab = g_nplans >= 0 ? g_nplans : (int)(sizeof(kPlans) / sizeof(kPlans[0]));
    for (int i = 0; i < ntab; ++i) {
      const PlanEntry& e = tab[i];
      if (e.mode == a.mode && e.M == a.M && e.N == a.N && e.C1 == a.C1 && e.C2 == a.C2 && e.taps == a.taps &&
          e.epilogue == a.epilogue && (e.out_dtype == VGEN_F32) == (a.out_dtype == VGEN_F32) && e.flags == flags &&
          legal(e.shape, e.bn, e.splitk)) {   // out_dtype: fp32 vs 16-bit (bf16 and fp16 launches share an entry)
        if (from_table) *from_table = true;
        return Plan{e.shape, e.bn, e.splitk};
      }
    }
  }
  Plan best{SHAPE_PP, cands[0], 1};
  double best_cost = 1e30;
  for (int shape = SHAPE_PP; shape <= SHAPE_PP128; ++shape) {
    if (force_shape >= 0 && shape != force_shape && !(a.colstats && force_shape == SHAPE_PP128)) continue;
    if (a.colstats && shape == SHAPE_PP128) continue;   // 32-row wave tiles: a slab would span two waves
    if (a.dualw && shape == SHAPE_DUAL) continue;       // dual-W K-steps exist on the ping-pong shapes only
    const int bm = shape == SHAPE_PP128 ? 128 : 256;
    const int64_t tiles_m = (a.M + bm - 1) / bm;
    for (int c = 0; c < nc; ++c) {
      const int bn = cands[c];
      const int bi = bn == 128 ? 0 : (bn == 160 ? 1 : 2);
      // us per 64-element K-step of one block: pp alone on its CU; dual alone / sharing the CU
      static const double t_pp[3] = {0.85, 1.15, 0.60};
      static const double t_d1[3] = {1.10, 1.30, 0.75};
      static const double t_d2[3] = {2.20, 2.50, 1.40};
      static const double t_p128[3] = {0.62, 0.80, 0.45};
      const int64_t tiles = tiles_m * ((a.N + bn - 1) / bn);
      for (int s = 1; s <= (smax < 1 ? 1 : smax); ++s) {
        const int64_t blocks = tiles * s;
        const int kts = (KT + s - 1) / s;
        // a dual-W pair = an ordinary K-step + an odd step that reads 10 of 18 fragments and issues 3 of 7 DMA pieces
        const double dwf = a.dualw ? 1.6 : 1.0;
        double cost;
        if (shape == SHAPE_PP) {
          cost = (double)((blocks + 255) / 256) * (kts * dwf * t_pp[bi] + 8.0) + epi_us;
        } else if (shape == SHAPE_PP128) {
          cost = (double)((blocks + 255) / 256) * (kts * dwf * t_p128[bi] + 6.0) + epi_us;
        } else if (blocks <= 256) {
          cost = kts * t_d1[bi] + 10.0 + epi_us;
        } else {
          cost = (double)((blocks + 511) / 512) * (kts * t_d2[bi] + 9.0) + 0.5 * epi_us + 1.0;
        }
        if (s > 1) cost += 5.0 + (double)(s + 1) * a.M * a.N * 4.0 / 3.0e6;   // partials at ~3 TB/s
        if (cost < best_cost - 1e-9) {
          best_cost = cost;
          best = Plan{shape, bn, s};
        }
      }
    }
  }
  return best;
}

template <typename T, int BM, int BN, int BK, int WM, int WN, int STAGES, bool PP, bool DW = false>
int launch(const vgen_tapgemm_args& a, int splitk, hipStream_t stream) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  static bool attr_done[VGEN_MAX_DEVICES] = {false};   // the opt-in is per device
  const int dev = vgen_device_slot();
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)tapgemm_kernel<T, BM, BN, BK, WM, WN, STAGES, PP, DW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vgen_set_error("tapgemm: hipFuncSetAttribute(%zu B LDS) failed: %s", lds,
                     hipGetErrorString(e));
      return (int)e;
    }
    attr_done[dev] = true;
  }
  const int64_t tiles_m = (a.M + BM - 1) / BM;
  const int64_t tiles_n = (a.N + BN - 1) / BN;
  const int64_t grid = tiles_m * tiles_n;
  if (grid <= 0) return 0;
  if (grid > 0x7fffffffLL) {
    vgen_set_error("tapgemm: grid too large");
    return VGEN_E_BADARG;
  }
  if (splitk > 1 && (a.ws == nullptr || a.ws_bytes < (size_t)splitk * a.M * a.N * sizeof(float)))
    splitk = 1;   // caller did not provide the workspace: still correct, just fewer blocks
#ifdef VGEN_TUNING
  const char* ab = getenv("VGEN_TAPGEMM_ABLATE");
  const int ablate = ab ? atoi(ab) : 0;
#else
  const int ablate = 0;
#endif
  // start stagger (see the kernel): a fraction of the estimated time of one tile (the cost model's constants), only for
  // launches with enough rounds of tiles to win it back
  constexpr int BPC = (WM * WN == 4) ? 2 : 1;      // co-resident blocks per CU
  const int cus = vgen_device_cus();
  int stagger = 0;
  {
    int pct = BPC == 1 ? STAGGER_PCT_PP : STAGGER_PCT_DUAL, min_rounds = STAGGER_MIN_ROUNDS;
#ifdef VGEN_TUNING
    if (const char* e = getenv("VGEN_TAPGEMM_STAGGER")) {
      int a1 = 0, a2 = 0, a3 = min_rounds;
      if (sscanf(e, "%d,%d,%d", &a1, &a2, &a3) >= 2) {
        pct = BPC == 1 ? a1 : a2;
        min_rounds = a3;
      }
    }
#endif
    const int64_t blocks = grid * splitk;
    const double rounds = (double)blocks / ((double)cus * BPC);
    if (pct > 0 && rounds >= (double)min_rounds) {
      const int KT = a.taps * (a.C1 / 64) + a.C2 / 64;
      const int kts = (KT + splitk - 1) / splitk;
      const double tile_us = kts * (a.dualw ? 1.6 : 1.0) * (BM == 256 ? (BPC == 1 ? 0.0060 * BN : 0.0150 * BN) : 0.0050 * BN) + 7.0;
      stagger = (int)(tile_us * pct / 100.0 * 2400.0 / 1024.0 + 0.5);
    }
  }
  hipLaunchKernelGGL((tapgemm_kernel<T, BM, BN, BK, WM, WN, STAGES, PP, DW>), dim3((unsigned)grid, (unsigned)splitk),
                     dim3(WM * WN * 64), lds, stream, a, splitk, (float*)a.ws, ablate, stagger, cus * BPC);
  int rc = vgen_check_launch("tapgemm");
  if (rc || splitk == 1) return rc;
  const int n_out = a.epilogue == VGEN_EPI_GEGLU ? a.N / 2 : a.N;
  const int64_t threads = a.M * (n_out / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     stream, a, splitk, (const float*)a.ws);
  return vgen_check_launch("tapgemm(splitk reduce)");
}

// the panel shape is asked first: a launch it takes is never planned on a streaming shape (tuning build:
// VGEN_TAPGEMM_PANEL=0 sends everything to the streaming shapes — the same-box A/B of the two)
int panel_bn(const vgen_tapgemm_args& a) {
#ifdef VGEN_TUNING
  if (env_int("VGEN_TAPGEMM_PANEL", 1) == 0) return 0;
#endif
  return vgen_panel_bn(a);
}

// The plan of a launch: a measured table entry first (r06: it may also take a K = 320 / 640 linear AWAY from the panel shape),
// then the panel shape for the launches it takes, then the cost model.
Plan full_plan(const vgen_tapgemm_args& a) {
  bool tabled = false;
  const Plan pl = make_plan(a, &tabled);
  if (!tabled)
    if (const int bn = panel_bn(a)) return Plan{SHAPE_PANEL, bn, 1};
  return pl;
}

template <typename T>
int dispatch(const vgen_tapgemm_args& a, hipStream_t s) {
  const Plan pl = full_plan(a);
  if (pl.shape == SHAPE_PANEL) return vgen_panel_launch(a, s);
  if (a.dualw) {
    if (pl.shape == SHAPE_PP128) {
      switch (pl.bn) {
        case 128: return launch<T, 128, 128, 64, 4, 2, 3, true, true>(a, pl.splitk, s);
        case 160: return launch<T, 128, 160, 64, 4, 2, 3, true, true>(a, pl.splitk, s);
        default: return launch<T, 128, 64, 64, 4, 2, 3, true, true>(a, pl.splitk, s);
      }
    }
    switch (pl.bn) {
      case 128: return launch<T, 256, 128, 64, 4, 2, 3, true, true>(a, pl.splitk, s);
      case 160: return launch<T, 256, 160, 64, 4, 2, 3, true, true>(a, pl.splitk, s);
      default: return launch<T, 256, 64, 64, 4, 2, 3, true, true>(a, pl.splitk, s);
    }
  }
  if (pl.shape == SHAPE_PP256) return launch<T, 256, 256, 32, 4, 2, 3, true>(a, pl.splitk, s);
  if (pl.shape == SHAPE_Q128) {
    switch (pl.bn) {
      case 128: return launch<T, 128, 128, 32, 2, 2, 3, false>(a, pl.splitk, s);
      case 160: return launch<T, 128, 160, 32, 2, 2, 3, false>(a, pl.splitk, s);
      default: return launch<T, 128, 64, 32, 2, 2, 3, false>(a, pl.splitk, s);
    }
  }
  if (pl.shape == SHAPE_PP) {
    switch (pl.bn) {
      case 128: return launch<T, 256, 128, 64, 4, 2, 3, true>(a, pl.splitk, s);
      case 160: return launch<T, 256, 160, 64, 4, 2, 3, true>(a, pl.splitk, s);
      default: return launch<T, 256, 64, 64, 4, 2, 3, true>(a, pl.splitk, s);
    }
  }
  if (pl.shape == SHAPE_PP128) {
    switch (pl.bn) {
      case 128: return launch<T, 128, 128, 64, 4, 2, 3, true>(a, pl.splitk, s);
      case 160: return launch<T, 128, 160, 64, 4, 2, 3, true>(a, pl.splitk, s);
      default: return launch<T, 128, 64, 64, 4, 2, 3, true>(a, pl.splitk, s);
    }
  }
  switch (pl.bn) {
    case 128: return launch<T, 256, 128, 32, 2, 2, 3, false>(a, pl.splitk, s);
    case 160: return launch<T, 256, 160, 32, 2, 2, 3, false>(a, pl.splitk, s);
    default: return launch<T, 256, 64, 32, 2, 2, 3, false>(a, pl.splitk, s);
  }
}

}  // namespace

extern "C" int vgen_tapgemm_query_plan(const vgen_tapgemm_args* args, int32_t* out3) {
  if (!args || !out3 || args->N <= 0 || args->M <= 0 || args->C1 <= 0 || args->C1 % 64 || args->C2 % 64) return VGEN_E_BADARG;
  const Plan pl = full_plan(*args);
  out3[0] = pl.shape;
  out3[1] = pl.bn;
  out3[2] = pl.splitk;
  return 0;
}

extern "C" int vgen_tapgemm_set_plans(const int64_t* rows, int32_t n) {
  if (n < 0) {            // back to the compiled-in table
    free(g_plans);
    g_plans = nullptr;
    g_nplans = -1;
    return 0;
  }
  if (n > 0 && !rows) return VGEN_E_BADARG;
  PlanEntry* t = (PlanEntry*)malloc(sizeof(PlanEntry) * (n > 0 ? n : 1));
  if (!t) return VGEN_E_BADARG;
  for (int i = 0; i < n; ++i) {
    const int64_t* r = rows + 12 * i;
    t[i] = PlanEntry{(int)r[0], r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], (int)r[6], (int)r[7], (int)r[8],
                     (int)r[9], (int)r[10], (int)r[11]};
  }
  free(g_plans);
  g_plans = t;
  g_nplans = n;
  return 0;
}

extern "C" size_t vgen_tapgemm_ws_bytes(const vgen_tapgemm_args* args) {
  if (!args || args->N <= 0 || args->M <= 0 || args->C1 <= 0 || args->C1 % 64 || args->C2 % 64) return 0;
  const int s = full_plan(*args).splitk;
  return s > 1 ? (size_t)s * args->M * args->N * sizeof(float) : 0;
}

extern "C" int vgen_tapgemm(const vgen_tapgemm_args* args, void* stream) {
  if (!args) {
    vgen_set_error("tapgemm: null args");
    return VGEN_E_BADARG;
  }
  const vgen_tapgemm_args& a = *args;
  VGEN_REQUIRE(a.dtype == VGEN_BF16 || a.dtype == VGEN_F16, "tapgemm: dtype must be bf16/f16");
  VGEN_REQUIRE(a.M >= 0 && a.N > 0, "tapgemm: bad M/N");
  VGEN_REQUIRE(a.C1 > 0 && a.C1 % 64 == 0 && a.C2 >= 0 && a.C2 % 64 == 0,
               "tapgemm: C1=%d / C2=%d must be multiples of 64", a.C1, a.C2);
  VGEN_REQUIRE(a.dualw == 0 || a.dualw == 1, "tapgemm: dualw must be 0 or 1");
  VGEN_REQUIRE(a.lda % 8 == 0 && (a.C2 == 0 || a.lda2 % 8 == 0) && a.ldw % 8 == 0 &&
                   (a.ldw == 0 || a.ldw >= ((int64_t)a.taps * a.C1 + a.C2) * (a.dualw ? 2 : 1)),
               "tapgemm: lda/lda2/ldw must be multiples of 8 (ldw >= K, 2 K with dualw)");
  VGEN_REQUIRE(vgen_aligned16(a.A) && vgen_aligned16(a.W) && vgen_aligned16(a.out) &&
                   (a.C2 == 0 || (a.A2 && vgen_aligned16(a.A2))),
               "tapgemm: pointers must be 16-byte aligned");
  VGEN_REQUIRE(a.bias == nullptr || vgen_aligned16(a.bias), "tapgemm: bias alignment");
  VGEN_REQUIRE(a.residual == nullptr || vgen_aligned16(a.residual), "tapgemm: residual alignment");
  VGEN_REQUIRE(a.rowbias == nullptr || (vgen_aligned16(a.rowbias) && a.rows_per_rb > 0),
               "tapgemm: rowbias alignment / rows_per_rb");
  VGEN_REQUIRE(a.out_dtype == VGEN_F32 || a.out_dtype == a.dtype, "tapgemm: out_dtype");
  VGEN_REQUIRE(a.ws == nullptr || vgen_aligned16(a.ws), "tapgemm: workspace alignment");
  switch (a.mode) {
    case VGEN_TAP_LINEAR:
      VGEN_REQUIRE(a.taps == 1, "tapgemm: linear mode needs taps == 1");
      break;
    case VGEN_TAP_CONV3X3:
      VGEN_REQUIRE(a.taps == 9 && a.Hi > 0 && a.Wi > 0 && a.Ho > 0 && a.Wo > 0 &&
                       (a.stride == 1 || a.stride == 2) && (a.ups == 0 || a.ups == 1) && a.crop_t >= 0 &&
                       (a.crop_t == 0 || a.ups == 1),
                   "tapgemm: bad conv3x3 geometry");
      VGEN_REQUIRE(a.M % ((int64_t)a.Ho * a.Wo) == 0, "tapgemm: M not a multiple of Ho*Wo");
      VGEN_REQUIRE((a.M / ((int64_t)a.Ho * a.Wo)) * a.Hi * a.Wi < (1LL << 31),
                   "tapgemm: source row index overflows int32");
      break;
    case VGEN_TAP_TEMPORAL3:
      VGEN_REQUIRE(a.taps == 3 && a.F > 0 && a.S > 0 && a.M % (a.S * a.F) == 0,
                   "tapgemm: bad temporal geometry");
      break;
    default:
      vgen_set_error("tapgemm: unknown mode %d", a.mode);
      return VGEN_E_BADARG;
  }
  VGEN_REQUIRE(a.M + 256 < (1LL << 31), "tapgemm: M overflows int32 row index");
  VGEN_REQUIRE(a.lda >= 0 && a.lda < (1LL << 30) && a.lda2 >= 0 && a.lda2 < (1LL << 30),
               "tapgemm: lda / lda2 must be in [0, 2^30)");
  VGEN_REQUIRE(((int64_t)a.taps * a.C1 + a.C2) * (a.dualw ? 4 : 2) <= ZERO_BYTES - 128,
               "tapgemm: K = %lld too long (<= 131008; <= 65504 with dualw)",
               (long long)((int64_t)a.taps * a.C1 + a.C2));
  if (a.epilogue == VGEN_EPI_GEGLU) {
    VGEN_REQUIRE(a.N % 64 == 0 && a.rowbias == nullptr && (a.ldo % 4 == 0) &&
                     (a.residual == nullptr || a.ldr % 4 == 0),
                 "tapgemm: GEGLU needs N %% 64 == 0, no rowbias, ldo/ldr %% 4 == 0");
  } else {
    VGEN_REQUIRE(a.epilogue == VGEN_EPI_NONE, "tapgemm: unknown epilogue");
  }
  if (a.colstats) {
    VGEN_REQUIRE(a.out_dtype == VGEN_F32 && a.epilogue == VGEN_EPI_NONE && a.N % 4 == 0 && a.ldo % 4 == 0 &&
                     (a.residual == nullptr || a.ldr % 4 == 0) && (a.rowbias == nullptr || a.rowbias_ld % 4 == 0) &&
                     vgen_aligned16(a.colstats),
                 "tapgemm: colstats needs fp32 output, no GEGLU, N/ldo/ldr/rowbias_ld %% 4 == 0");
  }
  if (a.split_out) {
    VGEN_REQUIRE(a.split_out == 1 && a.out_dtype != VGEN_F32 && a.epilogue == VGEN_EPI_NONE && a.colstats == nullptr &&
                     a.N % 32 == 0 && a.ldo % 8 == 0 && a.ldo >= 2 * (int64_t)a.N && (a.residual == nullptr || a.ldr % 4 == 0) &&
                     (a.rowbias == nullptr || a.rowbias_ld % 4 == 0),
                 "tapgemm: split_out needs a 16-bit output [M, >= 2 N], no GEGLU / colstats, N %% 32 == 0, ldo %% 8 == 0");
  }
  hipStream_t s = (hipStream_t)stream;
  return a.dtype == VGEN_BF16 ? dispatch<BF16>(a, s) : dispatch<F16>(a, s);
}


