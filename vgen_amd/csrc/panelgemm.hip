// panelgemm.hip — the W-panel-resident, barrier-free tap-GEMM shape for the short-K linears of the UNet's full-resolution
// level (K = C = 320: q/k/v, the attention out-projections, proj_in, the GEGLU up-projection of every Spatial- /
// TemporalTransformer at 32 x 56 — util.py:224-228, 337, 710, 1213 —; r05, VERDICT r04 "next" #1).
//
// Why another shape.  A K = 320 launch on the streaming shapes of tapgemm.hip is 5-10 K-steps per tile: the tile's fixed
// cost (operand prologue, two barriers per K-step, an epilogue that overlaps nothing inside its block) is ~9 K-steps'
// worth of time, i.e. most of the launch (NOTES §8: 2-3 x the floor on every level-0 linear).  Here:
//   * one block per CU, persistent.  It owns ONE panel of 160 weight rows (all of K: 100 KiB) and stages it into LDS ONCE
//     (LDS-DMA, same bank swizzle as tapgemm.hip's stages); the 8 waves only READ it afterwards, so the main loop has no
//     s_barrier: every wave walks its own 32-row slices (w, w + 8, ... of the block's row range), 200 MFMAs per slice, one
//     ds_read_b128 of W per TWO MFMAs (128 B/clk of the LDS's 256), the fragment reads running 8 ahead of their MFMAs.
//   * the A operand reaches the wave through its PRIVATE LDS ring (3 chunks of [32 rows x 32 k], 6 KiB per wave), filled
//     by LDS-DMA whose lanes read quad-contiguously (lane l -> row l >> 2, 16 bytes at chunk l & 3: 47 B/clk/CU from L2).
//     The first version loaded the fragments straight into MFMA layout (lane (lr, lq) = row lr, chunk lq: every QUAD of
//     lanes touches four different rows) — the texture addresser splits such a load into 64 requests and caps it at
//     17.5 B/clk/CU wherever the data sits (tools/probes/vmem_probe.hip, profiles/r05d / r05f / r05h): the kernel was bound by
//     VMEM issue, not by the matrix pipe.  Only the wave's own vmcnt / lgkmcnt order ring and reads — no barrier.
//   * waves are independent, so one wave's epilogue (bias, GEGLU gate, conversion, stores — as long as its MFMA loop:
//     the store path takes ~10 B/clk/CU) runs under its SIMD partner's MFMAs — provided the two are out of phase: waves
//     4-7 start PANEL_STAGGER x 64 cycles late.
//   * dual-W (vgen_tapgemm_args.dualw): the panel is 80 output columns, LDS rows 0-79 = W_hi, 80-159 = W_lo, both
//     accumulate into the same 80-column accumulators; A is staged once.  (GEGLU + dual-W: 64 columns.)
//   * blocks of one row range (all column panels) sit on the same XCD (block b -> XCD b % 8 renumbering), so the A rows
//     every panel re-reads come out of that XCD's L2: HBM sees A once.
// Same operand swap and C/D layout as tapgemm.hip (D[n][m]: a lane holds 4 consecutive n of one row m), same epilogue
// arithmetic order (bias, then residual through the accumulator's initial value; GEGLU gate polynomial; paired 16-byte
// 16-bit stores): single-pass results are BIT-IDENTICAL to the streaming shapes' (one accumulator chain per output in the
// same k order), dual-W ones differ by summation order (5e-6).
// Measured (MI355X, same process, operands not cache-resident; profiles/r05e_panel_probe_dma.json, r05g): GEGLU
// 57344 x 2560 x 320 175 -> 118 us, q/k/v 57344 x 960 x 320 85 -> 64 us (dual-W 114 -> 86), out-projection with fp32
// residual 55 -> 49 us (HBM-bound: 183 MB), whole t2v step -4 % in same-box A/Bs.
#include "common.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Tuning build only (-DVGEN_TUNING): when the launch carries a workspace pointer, every wave sums s_memtime differences
// per segment of its slice loop into it ([block * 8 + wave][8] int64: issue of the residual loads | wait until every
// outstanding load / store has landed | the MFMA loop | issue of the next slice's A loads | epilogue | slices) —
// tools/panel_probe.py --stamps.  The product build has neither the parameter's use nor the waits it adds.
#ifdef VGEN_TUNING
#define PANEL_STAMP(i)                                              \
  if (stamps) {                                                     \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
    seg[i] += now_ - tprev;                                         \
    tprev = now_;                                                   \
  }
#else
#define PANEL_STAMP(i)
#endif

// The A-chunk DMAs of the slice loop are issued through inline asm: hipcc books a `global_load_lds` as an access that may
// touch LDS through the flat path, and from then on degrades every counted lgkmcnt of the loop to lgkmcnt(0) — the W
// fragment ring (8 reads ahead) would drain twice per chunk.  The asm is invisible to that bookkeeping (its extra VMEM
// operations only make the compiler's own vmcnt waits stricter); the waits that order these DMAs against the fragment
// reads of their slot are the explicit ones in the loop.  m0 <- LDS byte address; one wait state before its use.
__device__ __forceinline__ void glds16_asm(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  (`#pragma unroll` is not enough
// for the main loop: with inline asm in its body hipcc keeps a run-time loop and indexes the register arrays through
// s_set_gpr_idx — 30 instructions between two MFMAs.)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

constexpr int PANEL_WAVES = 8;
constexpr int SLICE_ROWS = 32;
// x 64 cycles: head start of waves 0-3 over their SIMD partners 4-7.  Scanned 0 ... 128 per shape on the GPU
// (profiles/r05g_panel_stagger.json): GEGLU 138 -> 118 us at 64, q / qkv -4 ... -5 %, the HBM-bound fp32 launches +-1 %.
constexpr int PANEL_STAGGER = 64;
constexpr bool PANEL_K640 = true;     // K = 640 launches (80-column single-pass panels) take this shape too

enum { EPI_F32 = 0, EPI_16 = 1, EPI_GEGLU16 = 2 };   // fp32 store | 16-bit store | GEGLU gate + 16-bit store

template <typename T, int KS, int BN, bool DW, int EPI>
__global__ __launch_bounds__(PANEL_WAVES * 64) void panel_kernel(const vgen_tapgemm_args p, const int P, const int Cn,
                                                                  const int stagger) {
  constexpr int KT = KS / 2;                  // 64-element K-tiles of the LDS panel
  constexpr int LROWS = DW ? 2 * BN : BN;     // LDS rows: [W_hi rows | W_lo rows] with dual-W
  constexpr int NFL = LROWS / 16;             // W fragments per 32-element k-step
  constexpr int NF = BN / 16;                 // accumulator column fragments
  constexpr int MF = SLICE_ROWS / 16;         // accumulator row fragments (2)
  constexpr int TILE_BYTES = LROWS * 128;
  constexpr int W_BYTES = KT * TILE_BYTES;
  static_assert(KS % 2 == 0 && LROWS % 16 == 0 && BN % 16 == 0, "panel geometry");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const bias_lds = (float*)(smem + W_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15;   // row within a 16-row fragment
  const int lq = lane >> 4;   // k-chunk (operands) / 4-column group (C/D)

  // logical block id: XCD b % 8 gets a contiguous run (all panels of a row range share one L2)
  unsigned L;
  {
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int panel = (int)(L % (unsigned)P);
  const int chunk = (int)(L / (unsigned)P);
  const int n0 = panel * BN;
  const int M = (int)p.M;                                   // < 2^31 (host check)
  const int nslices = (M + SLICE_ROWS - 1) / SLICE_ROWS;
  const int s_begin = (int)(((int64_t)nslices * chunk) / Cn);
  const int s_end = (int)(((int64_t)nslices * (chunk + 1)) / Cn);

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ W = (const uint16_t*)p.W;
  const int64_t ldw = p.ldw ? p.ldw : (int64_t)KS * 32 * (DW ? 2 : 1);

  // ---- the A operand: per-WAVE staging ring in LDS, filled by LDS-DMA ---------------------------------------------------
  // r05 call C / D (profiles/r05c_panel_stamps.json, r05d_vmem_probe.txt): loading the fragments straight from row-major
  // memory into MFMA layout (lane (lr, lq) = row lr, 16 bytes at chunk lq: every quad of lanes touches 4 different rows)
  // runs at 17.5 B/clk/CU wherever the data sits — the texture addresser splits each quad that is not 64 contiguous
  // bytes — and made the first version of this kernel VMEM-issue-bound (20 such loads per slice took 1.7 - 5.6 k cycles
  // to issue against a 3.2 - 4 k cycle MFMA loop).  LDS-DMA with lane-linear sources moves the same bytes at 25 - 45
  // B/clk/CU.  Each wave owns a ring of ARING chunks of [32 rows x 32 k] (2 KiB = 2 DMA instructions of 16 rows x 64 B,
  // quad = one 64-byte piece); nobody else touches it, so the only synchronisation is the wave's own vmcnt / lgkmcnt:
  // chunk g + 3 is issued into the slot chunk g was read out of (its two fragments sit in registers by then), chunk
  // g + 1 is read into the other half of a register double buffer while chunk g multiplies.
  constexpr int ACH = 2048, ARING = 3;
  unsigned char* const abuf = smem + W_BYTES + BN * sizeof(float) + wave * (ARING * ACH);
  const int a_r16 = lane >> 2, a_c = lane & 3;
  const int a_src_chunk = (a_c ^ ((4 - ((a_r16 >> 2) & 3)) & 3)) << 3;             // swz_key<4> of tapgemm.hip, in elements
  const unsigned char* const ard = abuf + lr * 64 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 4);
  auto a_ptrs = [&](int sl, const uint16_t* (&ap)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row = sl * SLICE_ROWS + j * 16 + a_r16;
      row = row < M ? row : M - 1;                           // tail rows re-read the last row; never stored
      ap[j] = A + (int64_t)row * p.lda + a_src_chunk;
    }
  };
  const unsigned abuf_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)abuf;
  auto a_dma = [&](const uint16_t* const (&ap)[2], int kk, int slot_off) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_asm(ap[j] + kk * 32, abuf_lds + slot_off + j * 1024);
  };
  int s = s_begin + wave;
  const uint16_t* ap_cur[2];
  if (s < s_end) {
    a_ptrs(s, ap_cur);
#pragma unroll
    for (int kk = 0; kk < ARING; ++kk) a_dma(ap_cur, kk, kk * ACH);      // in flight while the panel is staged
  }

  // ---- stage the W panel: KT tiles of [LROWS][64] 16-bit, chunk c of row r at ((c ^ (r & 7)) << 4) ---------------
  {
    const int rsub = lane >> 3, c = lane & 7;
    constexpr int R8 = LROWS / 8;                            // 8-row groups (one 1-KiB DMA instruction each) per tile
    for (int i = wave; i < KT * R8; i += PANEL_WAVES) {
      const int kt = i / R8, r8 = i - kt * R8;
      const int lrow = r8 * 8 + rsub;
      const int lo = DW ? (lrow >= BN ? 1 : 0) : 0;
      const int n = n0 + lrow - lo * BN;
      const uint16_t* src = W + (int64_t)n * ldw + (DW ? (kt * 2 + lo) * 64 : kt * 64) + ((c ^ rsub) << 3);
      glds16(src, smem + kt * TILE_BYTES + r8 * 1024);
    }
    if (tid < BN) bias_lds[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // Waves w and w + 4 share a SIMD, leave the barrier together and have identical work: left alone they run their MFMA
  // loops at the same time (sharing the matrix pipe) and their epilogues at the same time (pipe idle) — the stamps of
  // r05 call E show an MFMA loop of ~2 x its stand-alone length.  The second wave of every SIMD starts `stagger` x 64
  // cycles late, so that one multiplies while the other converts and stores.
  if (wave >= PANEL_WAVES / 2) {
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }

  const unsigned char* const wrd = smem + lr * 128;          // this lane's row inside a 16-row fragment
  const int sw = lr & 7;
  const bool res_folded = EPI != EPI_GEGLU16 && p.residual != nullptr;     // GEGLU adds its residual after the gate

#ifdef VGEN_TUNING
  // the stamp buffer is the caller's workspace, used only when it holds gridDim.x * PANEL_WAVES * 8 int64 (ADVICE r05:
  // vgen_tapgemm_ws_bytes reports 0 for panel launches, so a split-K workspace of another size may be passed in)
  long long* const stamps = p.ws_bytes >= (size_t)gridDim.x * PANEL_WAVES * 8 * sizeof(long long) ? (long long*)p.ws : nullptr;
  long long seg[6] = {0, 0, 0, 0, 0, 0};
  long long tprev = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
#endif
  int ring0 = 0;                                             // ring slot of this slice's chunk 0 (chunk kk: (ring0 + kk) % 3)
  for (; s < s_end; s += PANEL_WAVES) {
    const int mrow0 = s * SLICE_ROWS;
    const bool has_next = s + PANEL_WAVES < s_end;
    int soff[ARING];                                         // LDS offset of the slot of chunk kk: soff[kk % 3]
#pragma unroll
    for (int j = 0; j < ARING; ++j) soff[j] = ((ring0 + j) % ARING) * ACH;
    f32x4 acc[NF][MF];
    // accumulators start from the fp32 residual tile (same as tapgemm.hip: the loads land while the A chunks do); tail
    // rows re-read a valid address and are never stored, so the loads need no per-lane predicate
    if (res_folded) {
#pragma unroll
      for (int mi = 0; mi < MF; ++mi) {
        int m = mrow0 + mi * 16 + lr;
        m = m < M ? m : M - 1;
        const float* const rp = p.residual + (int64_t)m * p.ldr + n0 + lq * 4;
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) acc[ni][mi] = *(const f32x4*)(rp + ni * 16);
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    PANEL_STAMP(0)
    // everything this wave has in flight lands here: chunks 0 - 2 of this slice (issued three chunks ago), the residual
    // tile, the previous slice's stores.  (One vmcnt serves loads and stores, so a counted wait past the stores would
    // have to assume their completion order.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PANEL_STAMP(1)

    // ---- KS chunks x NFL W fragments against the resident panel, two MFMAs each.  The W fragment reads run RING
    // fragments ahead of their MFMAs (left to itself hipcc serialises read -> lgkmcnt(0) -> 2 MFMAs through ONE register)
    constexpr int NT = KS * NFL;
    constexpr int RING = 8;
    u32x4 wring[RING];
    u32x4 xa[2][MF];                                          // A fragments of chunk kk in xa[kk & 1]
    auto wread = [&](int t) __attribute__((always_inline)) -> u32x4 {
      const int kk = t / NFL, nfl = t % NFL;
      return *(const u32x4*)(wrd + (kk >> 1) * TILE_BYTES + ((((kk & 1) * 4 + lq) ^ sw) << 4) + nfl * 16 * 128);
    };
    auto aread = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
      for (int mi = 0; mi < MF; ++mi) xa[kk & 1][mi] = *(const u32x4*)(ard + soff[kk % ARING] + mi * 1024);
    };
    aread(0);
#pragma unroll
    for (int t = 0; t < RING; ++t) wring[t] = wread(t);
    const uint16_t* ap_nxt[2];
    static_for<NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      constexpr int kk = t / NFL, nfl = t % NFL;
      constexpr int ni = DW ? nfl % NF : nfl;
      const u32x4 wv = wring[t % RING];
#pragma unroll
      for (int mi = 0; mi < MF; ++mi) acc[ni][mi] = T::mfma32(wv, xa[kk & 1][mi], acc[ni][mi]);
      if constexpr (t + RING < NT) wring[t % RING] = wread(t + RING);
      // chunk kk + 3 goes into the slot chunk kk came out of (its fragments were waited for by the MFMAs just issued).
      // In chunk 0 it is issued LAST: hipcc waits for the residual loads accumulator by accumulator (vmcnt(19) ... (0)
      // through the chunk's first touches) without knowing that the vmcnt(0) above already covered them — a DMA it cannot
      // see in the queue would turn its last vmcnt(0) into a wait for that DMA.
      if constexpr (nfl == (kk == 0 ? NFL - 1 : 0)) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kk + ARING < KS) {
          a_dma(ap_cur, kk + ARING, soff[kk % ARING]);
        } else {
          if (has_next) {
            if constexpr (kk + ARING == KS) a_ptrs(s + PANEL_WAVES, ap_nxt);
            a_dma(ap_nxt, kk + ARING - KS, soff[kk % ARING]);
          }
        }
      }
      if constexpr (nfl == 2 && kk + 1 < KS) {
        // chunk kk + 1 must have landed; newer than it in the queue: chunks kk + 2, kk + 3 (2 DMA instructions each) —
        // fewer at the tail of the wave's last slice, where nothing follows
        if (has_next || kk + 1 + 2 < KS) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (kk + 1 + 1 < KS) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        aread(kk + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_sched_barrier(0);
    PANEL_STAMP(2)
    if (has_next) {
      ap_cur[0] = ap_nxt[0];
      ap_cur[1] = ap_nxt[1];
    }
    ring0 = (ring0 + KS) % ARING;
    PANEL_STAMP(3)

    // ---- epilogue (N % BN == 0 is an eligibility condition: no column guards) -------------------------------------
    auto bias4 = [&](int ni) __attribute__((always_inline)) -> f32x4 { return *(const f32x4*)(bias_lds + ni * 16 + lq * 4); };
    // two column fragments -> one 16-byte store per lane (v_permlane16_swap, see tapgemm.hip's 16-bit epilogue)
    auto store_pair16 = [&](uint16_t* dst, const f32x4& va, const f32x4& vb) __attribute__((always_inline)) {
      const u32x2 pa = pack4<T>(va.x, va.y, va.z, va.w);
      const u32x2 pb = pack4<T>(vb.x, vb.y, vb.z, vb.w);
      const auto sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
      const u32x4 q = {sx[0], sy[0], sx[1], sy[1]};
      *(u32x4*)(dst + (lq & 1) * 16 + (lq >> 1) * 8) = q;
    };
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const int64_t m = mrow0 + mi * 16 + lr;
      if (m >= M) continue;
      if constexpr (EPI == EPI_F32) {
        float* const orow = (float*)p.out + m * p.ldo + n0 + lq * 4;
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) *(f32x4*)(orow + ni * 16) = acc[ni][mi] + bias4(ni);
      } else if constexpr (EPI == EPI_16) {
        uint16_t* const orow = (uint16_t*)p.out + m * p.ldo + n0;
#pragma unroll
        for (int ni = 0; ni + 1 < NF; ni += 2)
          store_pair16(orow + ni * 16, acc[ni][mi] + bias4(ni), acc[ni + 1][mi] + bias4(ni + 1));
        if constexpr (NF % 2 == 1) {
          const f32x4 v = acc[NF - 1][mi] + bias4(NF - 1);
          *(u32x2*)(orow + (NF - 1) * 16 + lq * 4) = pack4<T>(v.x, v.y, v.z, v.w);
        }
      } else {
        // packed columns [16 value | 16 gate]: fragment pairs (2 np, 2 np + 1) -> 16 output columns n0 / 2 + 16 np ...
        static_assert(EPI != EPI_GEGLU16 || NF % 2 == 0, "GEGLU pairs value / gate fragments");
        constexpr int NP = NF / 2;
        uint16_t* const orow = (uint16_t*)p.out + m * p.ldo + n0 / 2;
        const float* const rrow = p.residual ? p.residual + m * p.ldr + n0 / 2 + lq * 4 : nullptr;
        auto gated = [&](int np) __attribute__((always_inline)) -> f32x4 {
          f32x4 o = geglu4(acc[2 * np][mi] + bias4(2 * np), acc[2 * np + 1][mi] + bias4(2 * np + 1));
          if (rrow) o += *(const f32x4*)(rrow + np * 16);
          return o;
        };
#pragma unroll
        for (int np = 0; np + 1 < NP; np += 2) store_pair16(orow + np * 16, gated(np), gated(np + 1));
        if constexpr (NP % 2 == 1) {
          const f32x4 o = gated(NP - 1);
          *(u32x2*)(orow + (NP - 1) * 16 + lq * 4) = pack4<T>(o.x, o.y, o.z, o.w);
        }
      }
    }
    PANEL_STAMP(4)
#ifdef VGEN_TUNING
    seg[5] += 1;
#endif
  }
#ifdef VGEN_TUNING
  if (stamps && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) stamps[((int64_t)blockIdx.x * PANEL_WAVES + wave) * 8 + i] = seg[i];
  }
#endif
}

template <typename T, int KS, int BN, bool DW, int EPI>
int launch_panel(const vgen_tapgemm_args& a, hipStream_t stream) {
  constexpr int LROWS = DW ? 2 * BN : BN;
  constexpr size_t lds = (size_t)(KS / 2) * LROWS * 128 + BN * sizeof(float) + PANEL_WAVES * 3 * 2048;   // panel | bias | A rings
  static bool attr_done[VGEN_MAX_DEVICES] = {false};   // the opt-in is per device
  const int dev = vgen_device_slot();
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)panel_kernel<T, KS, BN, DW, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vgen_set_error("tapgemm(panel): hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_done[dev] = true;
  }
  const int P = a.N / BN;
  int Cn = vgen_device_cus() / P;                            // one block per CU: row ranges x panels <= the device's CUs
  const int nslices = (int)((a.M + SLICE_ROWS - 1) / SLICE_ROWS);
  if (Cn > nslices) Cn = nslices;
  if (Cn < 1) Cn = 1;
  int stagger = PANEL_STAGGER;
#ifdef VGEN_TUNING
  if (const char* e = getenv("VGEN_PANEL_STAGGER")) stagger = atoi(e);
#endif
  hipLaunchKernelGGL((panel_kernel<T, KS, BN, DW, EPI>), dim3((unsigned)(P * Cn)), dim3(PANEL_WAVES * 64), lds, stream, a, P,
                     Cn, stagger);
  return vgen_check_launch("tapgemm(panel)");
}

}  // namespace

// K = 640 panels: on by default once measured; the tuning build can switch them off (VGEN_PANEL_K640=0) for the A/B
static bool panel640_enabled() {
#ifdef VGEN_TUNING
  if (const char* e = getenv("VGEN_PANEL_K640")) return atoi(e) != 0;
#endif
  return PANEL_K640;
}

// which launches take the panel shape (host side; tapgemm.hip's dispatch asks before it plans a streaming shape):
// the column-panel width, 0 = not this shape
int vgen_panel_bn(const vgen_tapgemm_args& a) {
  const bool geglu = a.epilogue == VGEN_EPI_GEGLU;
  if (a.mode != VGEN_TAP_LINEAR || a.taps != 1 || a.C2 != 0 || (a.C1 != 320 && a.C1 != 640)) return 0;
  if (a.rowbias || a.colstats || a.split_out) return 0;
  if (a.M < 2048) return 0;                                  // a handful of slices per CU: the streaming shapes' split-K wins
  if (a.out_dtype == VGEN_F32 ? (a.ldo % 4 != 0 || geglu) : (a.ldo % 8 != 0)) return 0;
  if (a.residual && a.ldr % 4 != 0) return 0;
  int bn;
  if (a.C1 == 640) {
    // K = 640 (the 16 x 28 level): an 80-row single-pass panel is the 100 KiB; no dual-W, no GEGLU (40 / 64-column panels
    // would re-read A 2-4 x as often as the streaming tiles do)
    if (a.dualw || geglu || !panel640_enabled()) return 0;
    bn = 80;
  } else {
    bn = a.dualw ? (geglu ? 64 : 80) : 160;
  }
  return a.N % bn == 0 ? bn : 0;
}

template <typename T>
static int panel_dispatch(const vgen_tapgemm_args& a, hipStream_t s) {
  const int bn = vgen_panel_bn(a);
  const int epi = a.epilogue == VGEN_EPI_GEGLU ? EPI_GEGLU16 : (a.out_dtype == VGEN_F32 ? EPI_F32 : EPI_16);
  if (a.C1 == 640) {
    if (bn == 80 && epi == EPI_F32) return launch_panel<T, 20, 80, false, EPI_F32>(a, s);
    if (bn == 80 && epi == EPI_16) return launch_panel<T, 20, 80, false, EPI_16>(a, s);
  } else {
    if (bn == 160 && epi == EPI_F32) return launch_panel<T, 10, 160, false, EPI_F32>(a, s);
    if (bn == 160 && epi == EPI_16) return launch_panel<T, 10, 160, false, EPI_16>(a, s);
    if (bn == 160 && epi == EPI_GEGLU16) return launch_panel<T, 10, 160, false, EPI_GEGLU16>(a, s);
    if (bn == 80 && epi == EPI_F32) return launch_panel<T, 10, 80, true, EPI_F32>(a, s);
    if (bn == 80 && epi == EPI_16) return launch_panel<T, 10, 80, true, EPI_16>(a, s);
    if (bn == 64 && epi == EPI_GEGLU16) return launch_panel<T, 10, 64, true, EPI_GEGLU16>(a, s);
  }
  vgen_set_error("tapgemm(panel): launch not eligible");
  return VGEN_E_BADARG;
}

int vgen_panel_launch(const vgen_tapgemm_args& a, hipStream_t s) {
  return a.dtype == VGEN_BF16 ? panel_dispatch<BF16>(a, s) : panel_dispatch<F16>(a, s);
}
