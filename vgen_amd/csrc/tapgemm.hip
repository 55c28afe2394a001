// tapgemm.hip — the MFMA contraction kernel behind every Linear / 1x1 / 3x3 / (3,1,1) conv of
// the VGen UNet and VAE (see include/vgen_hip.h for the reference call sites).
//
// Design (gfx950 / CDNA4):
//   * implicit GEMM over "taps": out[m,:] = sum_tap A[src(m,tap), :] @ W[:, tap, :]^T.  With the
//     channels-last row layout every tap's K-slab is one contiguous C-length row of a shifted
//     pixel / frame, so a conv A-tile is a row GATHER of 128-byte pieces — no im2col buffer,
//     no torch.cat, no F.interpolate, no rearrange.
//   * 256 threads = 4 waves; block tile BM x BN with BK = 64; each wave owns a 64(m) x 64(n)
//     sub-tile = 4x4 MFMA 16x16x32 fragments (64 fp32 accumulators / lane).
//     Two shapes: 128x128 (2x2 waves) for N % 128 == 0, 256x64 (4x1 waves) otherwise
//     (N = 320/960: 5 / 15 exact 64-wide tiles instead of 2.5 / 7.5 128-wide ones).
//   * operands swapped on the matrix core: D[i = n][j = m] = sum_k W[n,k] * A[m,k].  The
//     C/D fragment then holds 4 CONSECUTIVE n for one m per lane -> bias / residual / output
//     move as one 16-byte (fp32) or 8-byte (16-bit) vector per fragment.
//   * global -> LDS staging by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB = 8 tile rows per
//     wave-instruction): no staging VGPRs and no ds_write pass (the v1 register-staged kernel was
//     LDS-write bound: 32 KiB of ds_write_b128 per K-tile at ~79 B/clk/CU).  Double-buffered,
//     one barrier per K-tile: the DMA of tile t+1 is issued before the MFMAs of tile t.
//     Out-of-range rows (conv padding, M/N tails) read a 16-byte zero line instead.
//   * LDS tiles are [rows][64] 16-bit (128 B / row).  The DMA destination is lane-linear, so the
//     XOR swizzle (chunk ^ (row & 7)) is applied on the per-lane SOURCE address and again on the
//     fragment reads: ds_read_b128 of a fragment (16 rows x one chunk per 16-lane group) is
//     bank-conflict free.
//   * split-K for launches with too few tiles to fill 256 CUs (the 4x7 / 8x14 levels of the UNet:
//     M = 896): partial fp32 tiles go to a caller workspace and a small second kernel reduces them
//     in a fixed order (deterministic) and applies the epilogue.
//   * fp32 accumulate; epilogue in fp32: + bias + per-image row-bias (time embedding)
//     + fp32 residual, optional GEGLU gate, fp32 or 16-bit store.
#include "common.h"

namespace {

constexpr int BK = 64;          // K elements per tile (128 bytes per row)
constexpr int ROW_BYTES = 128;  // BK * 2

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

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void tapgemm_kernel(const vgen_tapgemm_args p, const int splitk,
                                                      float* __restrict__ ws) {
  constexpr int WN = BN / 64;
  constexpr int WM = 4 / WN;
  static_assert(WM * 64 == BM, "block tile must be 4 waves of 64x64");
  constexpr int RA = BM / 32;  // A rows staged per thread
  constexpr int RB = BN / 32;  // W rows staged per thread

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA = smem;                          // [2][BM][128 B]
  unsigned char* const sB = smem + 2 * BM * ROW_BYTES;     // [2][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int lr = lane & 15;  // row within a 16-row fragment
  const int lq = lane >> 4;  // 16-lane group: k-chunk (operands) / 4-row group (C/D)

  const int tiles_n = (p.N + BN - 1) / BN;
  const int64_t tile = blockIdx.x;
  const int64_t m0 = (tile / tiles_n) * BM;
  const int n0 = (int)(tile % tiles_n) * BN;

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ A2 = (const uint16_t*)p.A2;
  const uint16_t* __restrict__ W = (const uint16_t*)p.W;
  const int64_t ldw = p.ldw ? p.ldw : (int64_t)p.taps * p.C1 + p.C2;

  // ---- per-thread staging assignment: chunk (16 B) ld_c of rows ld_r + 32*i ------------
  const int ld_c = tid & 7;
  const int ld_r = tid >> 3;

  RowState rs[RA];
  unsigned mvalid = 0;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int64_t m = m0 + ld_r + 32 * i;
    const bool ok = m < p.M;
    if (ok) mvalid |= 1u << i;
    if (p.mode == VGEN_TAP_CONV3X3) {
      const int hw = p.Ho * p.Wo;
      const int img = (int)(m / hw);
      const int rem = (int)(m - (int64_t)img * hw);
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      rs[i].base = img * p.Hi * p.Wi;
      rs[i].a = ok ? oy * p.stride - p.pad_t : INVALID;
      rs[i].b = ox * p.stride - p.pad_l;
    } else if (p.mode == VGEN_TAP_TEMPORAL3) {
      const int64_t fs = m / p.S;  // global frame index b*F + f
      rs[i].base = (int)m;
      rs[i].a = ok ? (int)(fs % p.F) : INVALID;
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

  // running (tap, c0) of the NEXT tile to load
  int nx_tap = kt_begin / cpt1, nx_c = kt_begin - (kt_begin / cpt1) * cpt1;

  // source chunk of this lane: LDS position ld_c holds chunk ld_c ^ (row & 7)
  const int src_c = (ld_c ^ (ld_r & 7)) * 8;
  const int wave_row0 = (tid >> 6) * 8;    // first tile row written by this wave's DMA (+32*i)

  auto load_tile = [&](int kt, int buf) {
    unsigned char* const dA = sA + buf * BM * ROW_BYTES + wave_row0 * ROW_BYTES;
    unsigned char* const dB = sB + buf * BN * ROW_BYTES + wave_row0 * ROW_BYTES;
    // ---- A rows ----
    if (kt < T1) {
      const int tap = nx_tap;
      const int c0 = nx_c * BK + src_c;
      int d0 = 0, d1 = 0;
      if (p.mode == VGEN_TAP_CONV3X3) {
        d0 = tap / 3;
        d1 = tap - 3 * d0;
      }
      const int Hv = p.Hi << p.ups, Wv = p.Wi << p.ups;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int64_t row;
        bool ok;
        if (p.mode == VGEN_TAP_CONV3X3) {
          const int iy = rs[i].a + d0;
          const int ix = rs[i].b + d1;
          ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv);
          row = (int64_t)rs[i].base + (int64_t)(iy >> p.ups) * p.Wi + (ix >> p.ups);
        } else if (p.mode == VGEN_TAP_TEMPORAL3) {
          const int f2 = rs[i].a + tap - 1;
          ok = (f2 >= 0) & (f2 < p.F);
          row = (int64_t)rs[i].base + (int64_t)(tap - 1) * p.S;
        } else {
          ok = rs[i].base >= 0;
          row = rs[i].base;
        }
        const void* src = ok ? (const void*)(A + row * p.lda + c0) : (const void*)&g_zero16;
        glds16(src, dA + i * 32 * ROW_BYTES);
      }
      if (++nx_c == cpt1) {
        nx_c = 0;
        ++nx_tap;
      }
    } else {
      const int c0 = (kt - T1) * BK + src_c;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const void* src = ((mvalid >> i) & 1u)
                              ? (const void*)(A2 + (m0 + ld_r + 32 * i) * p.lda2 + c0)
                              : (const void*)&g_zero16;
        glds16(src, dA + i * 32 * ROW_BYTES);
      }
    }
    // ---- W rows ----
    const int64_t kofs = (int64_t)kt * BK + src_c;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = n0 + ld_r + 32 * i;
      const void* src = n < p.N ? (const void*)(W + (int64_t)n * ldw + kofs) : (const void*)&g_zero16;
      glds16(src, dB + i * 32 * ROW_BYTES);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row (frag*16 + lr), chunk (ks*4 + lq) ^ (lr & 7)
  const int rd_row = lr * ROW_BYTES;
  const int sw = lr & 7;

  auto compute = [&](int buf) {
    const unsigned char* a = sA + buf * BM * ROW_BYTES + wm * 64 * ROW_BYTES + rd_row;
    const unsigned char* b = sB + buf * BN * ROW_BYTES + wn * 64 * ROW_BYTES + rd_row;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ((ks * 4 + lq) ^ sw) << 4;
      u32x4 wf[4], xf[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) wf[ni] = *(const u32x4*)(b + ni * 16 * ROW_BYTES + co);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) xf[mi] = *(const u32x4*)(a + mi * 16 * ROW_BYTES + co);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = T::mfma32(wf[ni], xf[mi], acc[ni][mi]);
    }
  };

  // ---- main loop: one barrier per K-tile ------------------------------------------------
  // The DMA is tracked by vmcnt; the explicit wait + barrier publishes a landed tile to all waves
  // and (WAR) guarantees every wave finished reading the buffer the next DMA overwrites.
  load_tile(kt_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    if (kt + 1 < kt_end) load_tile(kt + 1, buf ^ 1);
    compute(buf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (splitk > 1) {   // raw fp32 partial tile -> workspace [split][M][N]; epilogue in the reducer
    float* const wsp = ws + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int64_t m = m0 + wm * 64 + mi * 16 + lr;
      if (m >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + lq * 4;
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

#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t m = m0 + wm * 64 + mi * 16 + lr;
    if (m >= p.M) continue;
    const float* rbp = p.rowbias ? p.rowbias + (m / p.rows_per_rb) * p.rowbias_ld : nullptr;
    if (!geglu) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + lq * 4;
        if (n >= p.N) continue;
        f32x4 v = acc[ni][mi];
        if (vec) {
          if (p.bias) v += *(const f32x4*)(p.bias + n);
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
    } else {
      // packed columns: fragment pairs (value, gate) = (ni even, ni odd)
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        const int pn = n0 + wn * 64 + np * 32 + lq * 4;  // packed index of the value lanes
        if (pn >= p.N) continue;
        const int j = (n0 + wn * 64) / 2 + np * 16 + lq * 4;
        f32x4 val = acc[2 * np][mi];
        f32x4 gat = acc[2 * np + 1][mi];
        if (p.bias) {
          val += *(const f32x4*)(p.bias + pn);
          gat += *(const f32x4*)(p.bias + pn + 16);
        }
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = val[r] * gelu_erf_f(gat[r]);
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

// Deterministic split-K plan: only for launches that cannot fill the chip and whose epilogue
// operands are 16-byte vectorisable.
int plan_splitk(const vgen_tapgemm_args& a) {
  const bool wide = (a.N % 128 == 0);
  const int BM = wide ? 128 : 256, BN = wide ? 128 : 64;
  const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const int KT = a.taps * (a.C1 / 64) + a.C2 / 64;
  const int n_out = a.epilogue == VGEN_EPI_GEGLU ? a.N / 2 : a.N;
  const bool vec = (a.N % 4 == 0) && (n_out % 4 == 0) && (a.ldo % 4 == 0) &&
                   (a.residual == nullptr || a.ldr % 4 == 0) &&
                   (a.rowbias == nullptr || a.rowbias_ld % 4 == 0);
  if (!vec || tiles >= 160 || KT < 16) return 1;
  int64_t s = (384 + tiles - 1) / tiles;
  if (s > KT / 8) s = KT / 8;
  if (s > 32) s = 32;
  return s < 2 ? 1 : (int)s;
}

template <typename T, int BM, int BN>
int launch(const vgen_tapgemm_args& a, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)(BM + BN) * ROW_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)tapgemm_kernel<T, BM, BN>,
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
  int splitk = plan_splitk(a);
  if (splitk > 1 && (a.ws == nullptr || a.ws_bytes < (size_t)splitk * a.M * a.N * sizeof(float)))
    splitk = 1;   // caller did not provide the workspace: still correct, just fewer blocks
  hipLaunchKernelGGL((tapgemm_kernel<T, BM, BN>), dim3((unsigned)grid, (unsigned)splitk), dim3(256), lds,
                     stream, a, splitk, (float*)a.ws);
  int rc = vgen_check_launch("tapgemm");
  if (rc || splitk == 1) return rc;
  const int n_out = a.epilogue == VGEN_EPI_GEGLU ? a.N / 2 : a.N;
  const int64_t threads = a.M * (n_out / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     stream, a, splitk, (const float*)a.ws);
  return vgen_check_launch("tapgemm(splitk reduce)");
}

}  // namespace

extern "C" size_t vgen_tapgemm_ws_bytes(const vgen_tapgemm_args* args) {
  if (!args || args->N <= 0 || args->M <= 0 || args->C1 <= 0 || args->C1 % 64 || args->C2 % 64) return 0;
  const int s = plan_splitk(*args);
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
                       (a.stride == 1 || a.stride == 2) && (a.ups == 0 || a.ups == 1),
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
  VGEN_REQUIRE(a.M < (1LL << 31), "tapgemm: M overflows int32 row index");
  if (a.epilogue == VGEN_EPI_GEGLU) {
    VGEN_REQUIRE(a.N % 64 == 0 && a.rowbias == nullptr && (a.ldo % 4 == 0) &&
                     (a.residual == nullptr || a.ldr % 4 == 0),
                 "tapgemm: GEGLU needs N %% 64 == 0, no rowbias, ldo/ldr %% 4 == 0");
  } else {
    VGEN_REQUIRE(a.epilogue == VGEN_EPI_NONE, "tapgemm: unknown epilogue");
  }
  hipStream_t s = (hipStream_t)stream;
  const bool wide = (a.N % 128 == 0);
  if (a.dtype == VGEN_BF16) {
    return wide ? launch<BF16, 128, 128>(a, s) : launch<BF16, 256, 64>(a, s);
  } else {
    return wide ? launch<F16, 128, 128>(a, s) : launch<F16, 256, 64>(a, s);
  }
}
