// tapgemm.hip — the MFMA contraction kernel behind every Linear / 1x1 / 3x3 / (3,1,1) conv of
// the VGen UNet and VAE (see include/vgen_hip.h for the reference call sites).
//
// Design (gfx950 / CDNA4):
//   * implicit GEMM over "taps": out[m,:] = sum_tap A[src(m,tap), :] @ W[:, tap, :]^T.  With the
//     channels-last row layout every tap's K-slab is one contiguous C-length row of a shifted
//     pixel / frame, so a conv A-tile is a row GATHER of 128-byte pieces — no im2col buffer,
//     no torch.cat, no F.interpolate, no rearrange.
//   * 512 threads = 8 waves (4 along M x 2 along N), ONE block per CU.  Block tile 256 x BN with
//     BK = 64; BN = 128 (N % 128 == 0), 160 (N = 320 / 960 ...: exact tiles instead of 2.5 x 128)
//     or 64 (small / odd N).  Each wave owns a 64 x (BN/2) sub-tile of 16x16x32 MFMA fragments.
//     256-row tiles halve the L2->LDS bytes per flop of a 128x128 tile (85-98 flop/B).
//   * operands swapped on the matrix core: D[i = n][j = m] = sum_k W[n,k] * A[m,k].  The
//     C/D fragment then holds 4 CONSECUTIVE n for one m per lane -> bias / residual / output
//     move as one 16-byte (fp32) or 8-byte (16-bit) vector per fragment.
//   * global -> LDS staging by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB = 8 tile rows per
//     wave-instruction): no staging VGPRs, no ds_write pass.  THREE-stage ring (up to 156 KiB of
//     the 160 KiB LDS), one raw s_barrier per K-tile, COUNTED vmcnt: while tile t is multiplied
//     the DMAs of tiles t+1 and t+2 stay in flight across the barrier (~100 KiB in flight per CU,
//     what Little's law asks for at ~1 us of loaded L2/MALL latency).  v1 (register staging,
//     2 stages) was LDS-write bound, v2 (DMA, 2 stages, vmcnt(0) + __syncthreads) spent 40-50 %
//     of its wave cycles parked at the wait (profiles/).
//     Out-of-range rows (conv padding, M/N tails) read a 16-byte zero line instead.
//   * LDS tiles are [rows][64] 16-bit (128 B / row).  The DMA destination is lane-linear, so the
//     XOR swizzle (chunk ^ (row & 7)) is applied on the per-lane SOURCE address and again on the
//     fragment reads: ds_read_b128 of a fragment (16 rows x one chunk per 16-lane group) is
//     bank-conflict free (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * XCD-aware tile order: block b runs on XCD b % 8; tiles are renumbered so that each XCD works
//     on a contiguous run of tiles (same A rows, consecutive W panels) -> its private L2 sees reuse.
//   * split-K for launches with too few tiles to fill 256 CUs (the 4x7 / 8x14 levels of the UNet):
//     partial fp32 tiles go to a caller workspace and a small second kernel reduces them in a
//     fixed order (deterministic) and applies the epilogue.
//   * fp32 accumulate; epilogue in fp32: + bias + per-image row-bias (time embedding)
//     + fp32 residual, optional GEGLU gate, fp32 or 16-bit store.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 64;          // K elements per tile (128 bytes per row)
constexpr int ROW_BYTES = 128;  // BK * 2
// Two block shapes (template parameters BM / WM / STAGES of the kernel):
//   "big"   BM = 256, 4x2 waves (512 threads), 3-stage ring, one block per CU  — long-K GEMMs
//   "small" BM = 128, 2x2 waves (256 threads), 2-stage ring, two blocks per CU — the two blocks are
//           not barrier-coupled, so one block's MFMAs cover the other's fragment-read / epilogue
//           bubbles, and 512 tile slots halve the tile-quantisation loss of short launches.
constexpr int WN = 2;

struct RowState {
  int base;  // LINEAR/TEMPORAL: source row m; CONV: img * Hi * Wi
  int a;     // CONV: iy0 ; TEMPORAL: frame index
  int b;     // CONV: ix0
};

constexpr int INVALID = -(1 << 24);

__device__ u32x4 g_zero16 = {0u, 0u, 0u, 0u};   // source of padding / out-of-range rows

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

template <typename T, int BM, int BN, int STAGES>
__global__ __launch_bounds__(BM * 2) void tapgemm_kernel(const vgen_tapgemm_args p, const int splitk,
                                                         float* __restrict__ ws) {
  constexpr int WM = BM / 64;                   // waves along M (wave tile is 64 rows)
  constexpr int NT = WM * WN * 64;              // threads (= 2 * BM)
  constexpr int RPP = NT / 8;                   // tile rows covered by one DMA pass of the block
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile 64 x {64, 80, 32}
  constexpr int MF = WTM / 16, NF = WTN / 16;   // fragments per wave
  constexpr int RA = BM / RPP;                  // A DMA passes per tile (4)
  constexpr int RBF = BN / RPP;                 // full W passes
  constexpr int RBT = BN % RPP;                 // tail rows of W: only waves with wave*8 < RBT issue
  constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
  constexpr int LPT = RA + RBF;                 // DMA instructions per wave per tile (+1 with tail)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int lr = lane & 15;  // row within a 16-row fragment
  const int lq = lane >> 4;  // 16-lane group: k-chunk (operands) / 4-row group (C/D)
  const bool w_tail = RBT > 0 && wave * 8 < RBT;

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

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ A2 = (const uint16_t*)p.A2;
  const uint16_t* __restrict__ W = (const uint16_t*)p.W;
  const int64_t ldw = p.ldw ? p.ldw : (int64_t)p.taps * p.C1 + p.C2;

  // ---- per-thread DMA assignment: LDS chunk position ld_c of tile rows ld_r + 64*i -----------
  const int ld_c = tid & 7;
  const int ld_r = tid >> 3;

  RowState rs[RA];
  unsigned mvalid = 0;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    // M < 2^31 is enforced by the host wrapper: 32-bit index math (64-bit divides cost ~1 us/block)
    const unsigned m = (unsigned)m0 + ld_r + RPP * i;
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
  const int KT = T1 + p.C2 / BK;
  // this block's K-tile range (split-K: blockIdx.y)
  const int split = blockIdx.y;
  const int kt_begin = (int)(((int64_t)KT * split) / splitk);
  const int kt_end = (int)(((int64_t)KT * (split + 1)) / splitk);
  const int nk = kt_end - kt_begin;

  // ---- incremental per-lane DMA source pointers ---------------------------------------------
  // pc[j] = source of piece j for the NEXT K-tile to issue; consecutive K-tiles inside one tap /
  // K segment just advance by 128 bytes (0 for rows that read the zero line), so the steady-state
  // cost is one 64-bit add per piece; the row gather is recomputed only when the tap changes.
  // (v3a recomputed rows, bounds and 64-bit products every K-tile: ~700 ALU instructions per wave
  // per K-tile against 32 MFMAs — the loop was issue-bound, not memory- or MFMA-bound.)
  constexpr int NP = LPT + (RBT > 0 ? 1 : 0);
  const char* pc[NP];
  int inc[NP];
  const char* const zline = (const char*)&g_zero16;
  const int src_cb = (ld_c ^ (ld_r & 7)) * 16;   // byte offset of this lane's source chunk
  const int wave_row0 = wave * 8;                // first tile row written by this wave's DMA (+64*i)
  int kt_next = kt_begin;                        // K-tile the pointers currently describe
  int left;                                      // K-tiles until the A pointers must be regathered

  auto gather_a = [&](int kt) {                  // (re)compute pc[0..RA) for K-tile kt
    if (kt < T1) {
      const int tap = kt / cpt1;
      const int cch = kt - tap * cpt1;
      left = cpt1 - cch;
      int d0 = 0, d1 = 0;
      if (p.mode == VGEN_TAP_CONV3X3) {
        d0 = tap / 3;
        d1 = tap - 3 * d0;
      }
      const int Hv = (p.Hi << p.ups) - 2 * p.crop_t, Wv = p.Wi << p.ups;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int64_t row;
        bool ok;
        if (p.mode == VGEN_TAP_CONV3X3) {
          const int iy = rs[i].a + d0;
          const int ix = rs[i].b + d1;
          ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv);
          row = (int64_t)rs[i].base + (int64_t)((iy + p.crop_t) >> p.ups) * p.Wi + (ix >> p.ups);
        } else if (p.mode == VGEN_TAP_TEMPORAL3) {
          const int f2 = rs[i].a + tap - 1;
          ok = (f2 >= 0) & (f2 < p.F);
          row = (int64_t)rs[i].base + (int64_t)(tap - 1) * p.S;
        } else {
          ok = rs[i].base >= 0;
          row = rs[i].base;
        }
        pc[i] = ok ? (const char*)(A + row * p.lda + cch * BK) + src_cb : zline;
        inc[i] = ok ? ROW_BYTES : 0;
      }
    } else {
      left = KT - kt + 1;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const bool ok = (mvalid >> i) & 1u;
        pc[i] = ok ? (const char*)(A2 + (int64_t)((unsigned)m0 + ld_r + RPP * i) * p.lda2 + (kt - T1) * BK) + src_cb
                   : zline;
        inc[i] = ok ? ROW_BYTES : 0;
      }
    }
  };
  gather_a(kt_begin);
#pragma unroll
  for (int i = 0; i < NP - RA; ++i) {
    const int n = n0 + ld_r + RPP * i;
    const bool ok = n < p.N;
    pc[RA + i] = ok ? (const char*)(W + (int64_t)n * ldw + (int64_t)kt_begin * BK) + src_cb : zline;
    inc[RA + i] = ok ? ROW_BYTES : 0;
  }
  auto advance = [&]() {                         // pointers -> next K-tile
    ++kt_next;
    if (--left == 0) {
      if (kt_next < KT) gather_a(kt_next);
    } else {
#pragma unroll
      for (int i = 0; i < RA; ++i) pc[i] += inc[i];
    }
#pragma unroll
    for (int i = RA; i < NP; ++i) pc[i] += inc[i];
  };
  // LDS destination (wave-uniform) of piece j in `stage`
  auto piece_dst = [&](int stage, int j) -> unsigned char* {
    unsigned char* const base = smem + stage * STAGE_BYTES + wave_row0 * ROW_BYTES;
    return j < RA ? base + j * RPP * ROW_BYTES : base + (BM + (j - RA) * RPP) * ROW_BYTES;
  };
  auto load_tile = [&](int stage) {   // prologue: issue all pieces of the next K-tile back to back
#pragma unroll
    for (int j = 0; j < LPT; ++j) glds16(pc[j], piece_dst(stage, j));
    if (RBT > 0 && w_tail) glds16(pc[NP - 1], piece_dst(stage, NP - 1));
    advance();
  };

  f32x4 acc[NF][MF];
#pragma unroll
  for (int ni = 0; ni < NF; ++ni)
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row (frag*16 + lr), chunk (ks*4 + lq) ^ (lr & 7)
  const int rd_row = lr * ROW_BYTES;
  const int sw = lr & 7;

  // Multiply one staged K-tile and, interleaved with the MFMAs, issue the DMA pieces of the tile
  // two steps ahead.  An LDS-DMA instruction occupies the CU's single texture-address path for
  // ~16-25 cycles (1 KiB at <= 64 B/clk) and stalls the issuing wave ~100 cycles; issued in one
  // burst by all 8 waves right after the barrier (v3a) the DMA phase and the MFMA phase simply
  // added up (ablation: 142 us compute-only + 116 us DMA-only - 41 us fixed = 207 us measured at
  // 4096^3).  Spread one piece per MFMA group, the partner wave on the SIMD keeps the matrix pipe
  // busy while this wave waits on the address path.
  auto compute = [&](int stage, auto prefetch_tag, int stage_pf) {
    constexpr bool prefetch = decltype(prefetch_tag)::value;   // compile-time: branch-free K-step body
    const unsigned char* a = smem + stage * STAGE_BYTES + wm * WTM * ROW_BYTES + rd_row;
    const unsigned char* b = smem + stage * STAGE_BYTES + BM * ROW_BYTES + wn * WTN * ROW_BYTES + rd_row;
    u32x4 wf[2][NF], xf[2][MF];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ((ks * 4 + lq) ^ sw) << 4;
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) wf[ks][ni] = *(const u32x4*)(b + ni * 16 * ROW_BYTES + co);
#pragma unroll
      for (int mi = 0; mi < MF; ++mi) xf[ks][mi] = *(const u32x4*)(a + mi * 16 * ROW_BYTES + co);
    }
    constexpr int NMF = 2 * NF * MF;                 // MFMAs per K-tile per wave
    constexpr int GRP = NMF / NP;                    // MFMAs between two DMA pieces (>= NP slots)
    static_assert(GRP >= 1 && (NMF + GRP - 1) / GRP >= NP, "not enough MFMA groups for the DMA pieces");
    int piece = 0;
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      const int ks = i / (NF * MF), r = i % (NF * MF), ni = r / MF, mi = r % MF;
      if (i % GRP == 0) {
        if (prefetch) {
          if (piece < LPT) glds16(pc[piece], piece_dst(stage_pf, piece));
          else if (RBT > 0 && piece == NP - 1 && w_tail) glds16(pc[NP - 1], piece_dst(stage_pf, NP - 1));
        }
        ++piece;
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[ni][mi] = T::mfma32(wf[ks][ni], xf[ks][mi], acc[ni][mi]);
    }
    if (prefetch) advance();
  };

  // ---- main loop: STAGES-deep DMA ring, one barrier per K-tile, counted vmcnt ------------------
  // Iteration `it`: wait until tile `it` has landed (with 3 stages the DMA of tile it+1 may stay in
  // flight), barrier (publishes tile `it` of every wave AND proves every wave finished reading the
  // stage that the next DMA overwrites, last read in iteration it-1), issue tile it+STAGES-1
  // interleaved with the MFMAs of tile it.
  constexpr int AHEAD = STAGES - 1;
  for (int i = 0; i < AHEAD; ++i)
    if (nk > i) load_tile(i);
  int st_c = 0, st_l = AHEAD;   // stage to compute / stage to load into
  for (int it = 0; it < nk; ++it) {
    if (STAGES == 3 && it + 1 < nk) {
      if (w_tail) wait_vmcnt<LPT + 1>();
      else wait_vmcnt<LPT>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // two straight-line bodies (with / without prefetch) instead of a per-piece branch: hipcc can
    // then count lgkmcnt for the fragment reads instead of draining all 16 before the first MFMA
    if (it + AHEAD < nk) compute(st_c, std::true_type{}, st_l);
    else compute(st_c, std::false_type{}, st_l);
    st_c = st_c == STAGES - 1 ? 0 : st_c + 1;
    st_l = st_l == STAGES - 1 ? 0 : st_l + 1;
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
  const bool geglu = p.epilogue == VGEN_EPI_GEGLU;
  const int n_out = geglu ? p.N / 2 : p.N;
  const bool vec = ((n_out & 3) == 0) && ((p.ldo & 3) == 0) &&
                   (p.residual == nullptr || (p.ldr & 3) == 0) &&
                   (p.rowbias == nullptr || (p.rowbias_ld & 3) == 0);
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
#pragma unroll
  for (int mi = 0; mi < MF; ++mi) {
    const int64_t m = m0 + wm * WTM + mi * 16 + lr;
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
          if (p.residual) v += *(const f32x4*)(p.residual + m * p.ldr + n);
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
#pragma unroll
      for (int np = 0; np < NF / 2; ++np) {
        const int pn = n0 + wn * WTN + np * 32 + lq * 4;  // packed index of the value lanes
        if (pn >= p.N) continue;
        const int j = (n0 + wn * WTN) / 2 + np * 16 + lq * 4;
        const f32x4 val = acc[2 * np][mi] + bv[2 * np];
        const f32x4 gat = acc[2 * np + 1][mi] + bv[2 * np + 1];
        f32x4 o = geglu4(val, gat);
        if (p.residual) o += *(const f32x4*)(p.residual + m * p.ldr + j);
        if (p.out_dtype == VGEN_F32) {
          *(f32x4*)(of + m * p.ldo + j) = o;
        } else {
          *(u32x2*)(oh + m * p.ldo + j) = pack4<T>(o.x, o.y, o.z, o.w);
        }
      }
    }
  }
}

// split-K reducer: fixed summation order over the splits, then the same epilogue as the main kernel
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
// One block per CU and 256-row tiles make tile-count quantisation expensive (280 tiles = 2 rounds
// at 55 % fill), so the column tile BN and the split-K factor are chosen together from a small
// cost model (microseconds; constants fitted to profiles/r01_v3_tapgemm_shapes.json):
//   cost = rounds(tiles * s) * (ceil(KT / s) * t_ktile(BN) + t_tile) + [s > 1] * reduce(s)
// relative K-step time and per-tile fixed cost of the small (128-row, 2 blocks / CU) shape
constexpr double kSmallRel = 1.6;
constexpr double kSmallFixed = 6.0;

struct Plan {
  int bm;
  int bn;
  int splitk;
};

Plan make_plan(const vgen_tapgemm_args& a) {
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
  static const int force_bm = getenv("VGEN_TAPGEMM_FORCE_BM") ? atoi(getenv("VGEN_TAPGEMM_FORCE_BM")) : 0;
  const int smax = vec ? (KT / 4 < 32 ? KT / 4 : 32) : 1;
  Plan best{256, cands[0], 1};
  double best_cost = 1e30;
  for (int bm = 256; bm >= 128; bm -= 128) {
    if (force_bm && bm != force_bm) continue;
    const int64_t tiles_m = (a.M + bm - 1) / bm;
    const int slots = bm == 256 ? 256 : 512;          // co-resident blocks on the chip
    for (int c = 0; c < nc; ++c) {
      const int bn = cands[c];
      // time of one K-step of one block (us); a small block shares its CU with a second one
      const double t_ktile = (bn == 160 ? 1.45 : (bn == 128 ? 1.2 : 0.75)) * (bm == 256 ? 1.0 : kSmallRel);
      const double t_tile = bm == 256 ? 6.0 : kSmallFixed;
      const int64_t tiles = tiles_m * ((a.N + bn - 1) / bn);
      for (int s = 1; s <= (smax < 1 ? 1 : smax); ++s) {
        const int64_t rounds = (tiles * s + slots - 1) / slots;
        double cost = (double)rounds * (((KT + s - 1) / s) * t_ktile + t_tile);
        if (s > 1) cost += 5.0 + (double)(s + 1) * a.M * a.N * 4.0 / 3.0e6;   // partials at ~3 TB/s
        if (cost < best_cost - 1e-9) {
          best_cost = cost;
          best = Plan{bm, bn, s};
        }
      }
    }
  }
  return best;
}

template <typename T, int BM, int BN, int STAGES>
int launch(const vgen_tapgemm_args& a, int splitk, hipStream_t stream) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * ROW_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)tapgemm_kernel<T, BM, BN, STAGES>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vgen_set_error("tapgemm: hipFuncSetAttribute(%zu B LDS) failed: %s", lds,
                     hipGetErrorString(e));
      return (int)e;
    }
    attr_done = true;
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
  hipLaunchKernelGGL((tapgemm_kernel<T, BM, BN, STAGES>), dim3((unsigned)grid, (unsigned)splitk),
                     dim3(BM * 2), lds, stream, a, splitk, (float*)a.ws);
  int rc = vgen_check_launch("tapgemm");
  if (rc || splitk == 1) return rc;
  const int n_out = a.epilogue == VGEN_EPI_GEGLU ? a.N / 2 : a.N;
  const int64_t threads = a.M * (n_out / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     stream, a, splitk, (const float*)a.ws);
  return vgen_check_launch("tapgemm(splitk reduce)");
}

template <typename T>
int dispatch(const vgen_tapgemm_args& a, hipStream_t s) {
  const Plan pl = make_plan(a);
  if (pl.bm == 256) {
    switch (pl.bn) {
      case 128: return launch<T, 256, 128, 3>(a, pl.splitk, s);
      case 160: return launch<T, 256, 160, 3>(a, pl.splitk, s);
      default: return launch<T, 256, 64, 3>(a, pl.splitk, s);
    }
  }
  switch (pl.bn) {
    case 128: return launch<T, 128, 128, 2>(a, pl.splitk, s);
    case 160: return launch<T, 128, 160, 2>(a, pl.splitk, s);
    default: return launch<T, 128, 64, 2>(a, pl.splitk, s);
  }
}

}  // namespace

extern "C" size_t vgen_tapgemm_ws_bytes(const vgen_tapgemm_args* args) {
  if (!args || args->N <= 0 || args->M <= 0 || args->C1 <= 0 || args->C1 % 64 || args->C2 % 64) return 0;
  const int s = make_plan(*args).splitk;
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
  VGEN_REQUIRE(a.lda % 8 == 0 && (a.C2 == 0 || a.lda2 % 8 == 0) && a.ldw % 8 == 0 &&
                   (a.ldw == 0 || a.ldw >= (int64_t)a.taps * a.C1 + a.C2),
               "tapgemm: lda/lda2/ldw must be multiples of 8 (ldw >= K)");
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
  if (a.epilogue == VGEN_EPI_GEGLU) {
    VGEN_REQUIRE(a.N % 64 == 0 && a.rowbias == nullptr && (a.ldo % 4 == 0) &&
                     (a.residual == nullptr || a.ldr % 4 == 0),
                 "tapgemm: GEGLU needs N %% 64 == 0, no rowbias, ldo/ldr %% 4 == 0");
  } else {
    VGEN_REQUIRE(a.epilogue == VGEN_EPI_NONE, "tapgemm: unknown epilogue");
  }
  hipStream_t s = (hipStream_t)stream;
  return a.dtype == VGEN_BF16 ? dispatch<BF16>(a, s) : dispatch<F16>(a, s);
}
