// norms.hip — GroupNorm(+SiLU) and LayerNorm, fp32 in -> 16-bit out (HBM-bound kernels).
//
// GroupNorm runs as three launches over channels-last rows [nb*S, C]:
//   1. gn_stats   : grid (nsplit, nb); every block streams a contiguous slab of rows with
//                   16-byte loads (all channels of a row are contiguous -> fully coalesced),
//                   keeps per-channel partial sums in registers, folds channels -> groups
//                   through LDS in a fixed order and writes (count, mean, M2) per group.
//   2. gn_finalize: grid (nb); Chan-combines the nsplit partials per group -> (mean, rstd).
//   3. gn_apply   : same streaming pattern; y = act(x * scale_c + shift_c) with
//                   scale_c = gamma_c * rstd_g, shift_c = beta_c - mean_g * scale_c held in
//                   registers; 8-byte 16-bit stores.
// The reduction order is fixed (no atomics) so results are run-to-run deterministic.
// The input row is the virtual concat [x1 | x2] (decoder skip connections).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SLOTS = 3;   // float4 slots per thread -> C <= 3072
constexpr int GN_G = 32;          // groups supported per launch (reference always uses 32)

struct GnGeom {
  int C, C1, C2;
  int nslots;       // C / 4
  int tpr;          // threads per row pass
  int rpb;          // rows in flight per block pass
  int spt;          // slots per thread
  int cpg;          // channels per group
};

__device__ __forceinline__ GnGeom gn_geom(int C1, int C2, int groups) {
  GnGeom g;
  g.C1 = C1;
  g.C2 = C2;
  g.C = C1 + C2;
  g.nslots = g.C >> 2;
  if (g.nslots <= GN_THREADS) {
    g.tpr = g.nslots;
    g.rpb = GN_THREADS / g.nslots;
    g.spt = 1;
  } else {
    g.tpr = GN_THREADS;
    g.rpb = 1;
    g.spt = (g.nslots + GN_THREADS - 1) / GN_THREADS;
  }
  g.cpg = g.C / groups;
  return g;
}

__device__ __forceinline__ f32x4 gn_load(const float* x1, const float* x2, const GnGeom& g,
                                         int64_t row, int c) {
  if (c < g.C1) return *(const f32x4*)(x1 + row * g.C1 + c);
  return *(const f32x4*)(x2 + row * g.C2 + (c - g.C1));
}

// ws layout: part[nb][nsplit][groups][3] (count, mean, M2) then stat[nb][C][2]: per-channel
// (scale, shift) = (gamma * rstd, beta - mean * gamma * rstd), written by the finalize kernels so that the
// apply blocks start streaming after two coalesced loads instead of 16 dependent scalar ones
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(
    const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2, int64_t S,
    int groups, int nsplit, float* __restrict__ part) {
  __shared__ float red[2][3072];  // [sum|sumsq][rowlane * C + c]  (rpb * C <= 1024 when rpb > 1)
  const GnGeom g = gn_geom(C1, C2, groups);
  const int tid = threadIdx.x;
  const int split = blockIdx.x;
  const int64_t nb = blockIdx.y;
  const int64_t rows_per = (S + nsplit - 1) / nsplit;
  const int64_t r_begin = (int64_t)split * rows_per;
  const int64_t r_end = min(S, r_begin + rows_per);
  const int rowlane = tid / g.tpr;
  const int slot0 = tid - rowlane * g.tpr;
  const bool active = rowlane < g.rpb;

  f32x4 s[GN_MAX_SLOTS], ss[GN_MAX_SLOTS];
#pragma unroll
  for (int k = 0; k < GN_MAX_SLOTS; ++k) {
    s[k] = f32x4{0, 0, 0, 0};
    ss[k] = f32x4{0, 0, 0, 0};
  }
  if (active) {
#pragma unroll 4
    for (int64_t r = r_begin + rowlane; r < r_end; r += g.rpb) {
      const int64_t row = nb * S + r;
#pragma unroll
      for (int k = 0; k < GN_MAX_SLOTS; ++k) {
        const int slot = slot0 + k * GN_THREADS;
        if (k < g.spt && slot < g.nslots) {
          const f32x4 v = gn_load(x1, x2, g, row, slot * 4);
          s[k] += v;
          ss[k] += v * v;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < GN_MAX_SLOTS; ++k) {
      const int slot = slot0 + k * GN_THREADS;
      if (k < g.spt && slot < g.nslots) {
        const int o = rowlane * g.C + slot * 4;
        *(f32x4*)&red[0][o] = s[k];
        *(f32x4*)&red[1][o] = ss[k];
      }
    }
  }
  __syncthreads();
  if (tid < groups) {
    float a = 0.f, b = 0.f;
    for (int rl = 0; rl < g.rpb; ++rl) {
      const int o = rl * g.C + tid * g.cpg;
      for (int c = 0; c < g.cpg; ++c) {
        a += red[0][o + c];
        b += red[1][o + c];
      }
    }
    const float n = (float)((r_end > r_begin ? (r_end - r_begin) : 0) * g.cpg);
    const float mean = n > 0.f ? a / n : 0.f;
    const float m2 = n > 0.f ? fmaxf(b - a * mean, 0.f) : 0.f;
    float* o = part + ((nb * nsplit + split) * groups + tid) * 3;
    o[0] = n;
    o[1] = mean;
    o[2] = m2;
  }
}

__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part,
                                                         int groups, int nsplit, float eps, int C,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         float* __restrict__ stat) {
  // one wave per (batch, group): lane j Chan-folds splits j, j+64, ... (<= 16 each), then a 6-step
  // butterfly.  (The first version used 8 lanes per group in one block per batch: up to 128
  // dependent steps — 17 us per launch, as slow as the streaming passes it sits between.)
  const int lane = threadIdx.x;
  const int grp = blockIdx.x;
  const int64_t nb = blockIdx.y;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int sp = lane; sp < nsplit; sp += 64) {
    const float* p = part + ((nb * nsplit + sp) * groups + grp) * 3;
    const float nb_ = p[0], mb = p[1], m2b = p[2];
    if (nb_ > 0.f) {
      const float nt = n + nb_;
      const float d = mb - mean;
      mean += d * (nb_ / nt);
      m2 += m2b + d * d * (n * nb_ / nt);
      n = nt;
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float n2 = __shfl_xor(n, o, 64), mean2 = __shfl_xor(mean, o, 64),
                m22 = __shfl_xor(m2, o, 64);
    // symmetric form: both partners compute bit-identical results
    const float nt = n + n2;
    if (nt > 0.f) {
      const float lo_n = (lane & o) ? n2 : n, hi_n = (lane & o) ? n : n2;
      const float lo_m = (lane & o) ? mean2 : mean, hi_m = (lane & o) ? mean : mean2;
      const float d = hi_m - lo_m;
      const float w = hi_n / nt;
      m2 = m2 + m22 + d * d * (lo_n * w);
      mean = lo_m + d * w;
    }
    n = nt;
  }
  // the symmetric butterfly leaves identical (n, mean, m2) in every lane
  const float rstd = 1.0f / sqrtf((n > 0.f ? m2 / n : 0.f) + eps);
  const int cpg = C / groups;
  for (int j = lane; j < cpg; j += 64) {
    const int c = grp * cpg + j;
    const float a = gamma[c] * rstd;
    stat[(nb * C + c) * 2 + 0] = a;
    stat[(nb * C + c) * 2 + 1] = beta[c] - mean * a;
  }
}

// finalize from column partials (vgen_groupnorm_cs): one 256-thread block per (batch, group); items
// are (64-row slab, channel) pairs of the group, each (n = 64, mean, M2) from its (sum, sumsq).
// Threads fold items tid, tid + 256, ... (4 loads in flight), waves fold by butterfly, the 4 wave
// results are merged in index order.  (First version: one wave, one dependent load pair per step —
// 70 serial L2 round trips per launch, as slow as the statistics pass it replaced.)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb_, float mb, float m2b) {
  const float nt = n + nb_;
  if (nt > 0.f) {
    const float d = mb - mean;
    const float w = nb_ / nt;
    m2 = m2 + m2b + d * d * (n * w);
    mean = mean + d * w;
  }
  n = nt;
}

// r04: NT = 1024 for the 5-D norms of the big levels (448 slabs x 10..40 channels = 4.5 K - 18 K items per group: 18 - 70
// dependent L2 round trips per thread at 256 threads were most of this launch's ~10 us; 68 such launches per step).
template <int NT>
__global__ __launch_bounds__(NT) void gn_finalize_cs_kernel(const float* __restrict__ cs1, int C1,
                                                             const float* __restrict__ cs2, int C2,
                                                             int64_t S, int groups, float eps,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ stat) {
  constexpr int NW = NT / 64;
  __shared__ float red[NW][3];
  __shared__ float mr[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int grp = blockIdx.x;
  const int64_t nb = blockIdx.y;
  const int cpg = (C1 + C2) / groups;
  const int slabs = (int)(S / 64);
  const int items = slabs * cpg;
  const int64_t srow0 = nb * slabs;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  auto fetch = [&](int it, float& sm, float& sq) __attribute__((always_inline)) {
    const int sl = it / cpg;
    const int c = grp * cpg + (it - sl * cpg);
    const int64_t srow = srow0 + sl;
    if (c < C1) {
      sm = cs1[(srow * 2) * C1 + c];
      sq = cs1[(srow * 2 + 1) * C1 + c];
    } else {
      sm = cs2[(srow * 2) * C2 + (c - C1)];
      sq = cs2[(srow * 2 + 1) * C2 + (c - C1)];
    }
  };
  int it = tid;
  for (; it + 3 * NT < items; it += 4 * NT) {
    float sm[4], sq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(it + u * NT, sm[u], sq[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float mb = sm[u] * (1.0f / 64.f);
      chan_merge(n, mean, m2, 64.f, mb, fmaxf(sq[u] - sm[u] * mb, 0.f));
    }
  }
  for (; it < items; it += NT) {
    float sm, sq;
    fetch(it, sm, sq);
    const float mb = sm * (1.0f / 64.f);
    chan_merge(n, mean, m2, 64.f, mb, fmaxf(sq - sm * mb, 0.f));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float n2 = __shfl_xor(n, o, 64), mean2 = __shfl_xor(mean, o, 64),
                m22 = __shfl_xor(m2, o, 64);
    // symmetric form: both partners compute bit-identical results
    const float nt = n + n2;
    if (nt > 0.f) {
      const float lo_n = (lane & o) ? n2 : n, hi_n = (lane & o) ? n : n2;
      const float lo_m = (lane & o) ? mean2 : mean, hi_m = (lane & o) ? mean : mean2;
      const float d = hi_m - lo_m;
      const float wgt = hi_n / nt;
      m2 = m2 + m22 + d * d * (lo_n * wgt);
      mean = lo_m + d * wgt;
    }
    n = nt;
  }
  if (lane == 0) {
    red[w][0] = n;
    red[w][1] = mean;
    red[w][2] = m2;
  }
  __syncthreads();
  if (tid == 0) {
    float N = red[0][0], M = red[0][1], Q = red[0][2];
    for (int k = 1; k < NW; ++k) chan_merge(N, M, Q, red[k][0], red[k][1], red[k][2]);
    const float var = N > 0.f ? Q / N : 0.f;
    mr[0] = M;
    mr[1] = 1.0f / sqrtf(var + eps);
  }
  __syncthreads();
  const int C = C1 + C2;
  for (int j = tid; j < cpg; j += NT) {
    const int c = grp * cpg + j;
    const float a = gamma[c] * mr[1];
    stat[(nb * C + c) * 2 + 0] = a;
    stat[(nb * C + c) * 2 + 1] = beta[c] - mr[0] * a;
  }
}

// raw copy of the un-normalised input (the ResBlock's 1x1 skip-conv operand): plain 16-bit, or — raw_lo > 0 — as a
// two-term row [hi | lo] with lo = round16(x - hi) at column offset raw_lo (row stride raw_ld = 2 C)
template <typename T>
__device__ __forceinline__ void store_raw4(uint16_t* raw, int64_t off, int raw_lo, const f32x4& v) {
  const u32x2 h = pack4<T>(v.x, v.y, v.z, v.w);
  *(u32x2*)(raw + off) = h;
  if (raw_lo > 0) {
    *(u32x2*)(raw + off + raw_lo) =
        pack4<T>(v.x - T::to_f32((uint16_t)(h.x & 0xffffu)), v.y - T::to_f32((uint16_t)(h.x >> 16)),
                 v.z - T::to_f32((uint16_t)(h.y & 0xffffu)), v.w - T::to_f32((uint16_t)(h.y >> 16)));
  }
}
template <typename T>
__device__ __forceinline__ void store_raw2(uint16_t* raw, int64_t off, int raw_lo, float a, float b) {
  const uint32_t h = T::pack2(a, b);
  *(uint32_t*)(raw + off) = h;
  if (raw_lo > 0) *(uint32_t*)(raw + off + raw_lo) = T::pack2(a - T::to_f32((uint16_t)(h & 0xffffu)), b - T::to_f32((uint16_t)(h >> 16)));
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(
    const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2, int64_t S,
    int groups, int nsplit, const float* __restrict__ stat, const float* __restrict__ gamma,
    const float* __restrict__ beta, int silu, uint16_t* __restrict__ y,
    uint16_t* __restrict__ raw, int raw_lo) {
  const GnGeom g = gn_geom(C1, C2, groups);
  const int raw_ld = raw_lo > 0 ? 2 * g.C : g.C;
  const int tid = threadIdx.x;
  const int split = blockIdx.x;
  const int64_t nb = blockIdx.y;
  const int64_t rows_per = (S + nsplit - 1) / nsplit;
  const int64_t r_begin = (int64_t)split * rows_per;
  const int64_t r_end = min(S, r_begin + rows_per);
  const int rowlane = tid / g.tpr;
  const int slot0 = tid - rowlane * g.tpr;
  if (rowlane >= g.rpb) return;

  f32x4 sc[GN_MAX_SLOTS], sh[GN_MAX_SLOTS];
#pragma unroll
  for (int k = 0; k < GN_MAX_SLOTS; ++k) {
    const int slot = slot0 + k * GN_THREADS;
    if (k < g.spt && slot < g.nslots) {
      const f32x4 p0 = *(const f32x4*)(stat + (nb * g.C + slot * 4) * 2);       // a0 b0 a1 b1
      const f32x4 p1 = *(const f32x4*)(stat + (nb * g.C + slot * 4) * 2 + 4);   // a2 b2 a3 b3
      sc[k] = f32x4{p0.x, p0.z, p1.x, p1.z};
      sh[k] = f32x4{p0.y, p0.w, p1.y, p1.w};
    }
  }
  if (g.spt == 1) {
    // one 16-byte slot per thread and row (C <= 1024, every UNet width): explicit register double
    // buffer, U rows per thread in flight while the previous U are normalised and stored
    constexpr int U = 4;
    const int c = slot0 * 4;
    const bool from1 = c < g.C1;
    const float* src = from1 ? x1 + c : x2 + (c - g.C1);
    const int64_t ld = from1 ? g.C1 : g.C2;
    f32x4 cur[U], nxt[U];
    int64_t r = r_begin + rowlane;
    auto load = [&](int64_t r0, f32x4* dst) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = r0 + (int64_t)u * g.rpb;
        const int64_t row = nb * S + (rr < r_end ? rr : r_end - 1);     // clamped rows are never stored
        dst[u] = *(const f32x4*)(src + row * ld);
      }
    };
    if (r < r_end) load(r, nxt);
    for (; r < r_end; r += (int64_t)U * g.rpb) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
      if (r + (int64_t)U * g.rpb < r_end) load(r + (int64_t)U * g.rpb, nxt);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = r + (int64_t)u * g.rpb;
        if (rr < r_end) {
          const int64_t row = nb * S + rr;
          f32x4 o = cur[u] * sc[0] + sh[0];
          if (silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = silu_f(o[e]);
          }
          *(u32x2*)(y + row * g.C + c) = pack4<T>(o.x, o.y, o.z, o.w);
          if (raw) store_raw4<T>(raw, row * raw_ld + c, raw_lo, cur[u]);
        }
      }
    }
    return;
  }
#pragma unroll 4
  for (int64_t r = r_begin + rowlane; r < r_end; r += g.rpb) {
    const int64_t row = nb * S + r;
#pragma unroll
    for (int k = 0; k < GN_MAX_SLOTS; ++k) {
      const int slot = slot0 + k * GN_THREADS;
      if (k < g.spt && slot < g.nslots) {
        const f32x4 v = gn_load(x1, x2, g, row, slot * 4);
        f32x4 o = v * sc[k] + sh[k];
        if (silu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = silu_f(o[e]);
        }
        *(u32x2*)(y + row * g.C + slot * 4) = pack4<T>(o.x, o.y, o.z, o.w);
        if (raw) store_raw4<T>(raw, row * raw_ld + slot * 4, raw_lo, v);
      }
    }
  }
}

// ---- single-launch GroupNorm for small tensors ------------------------------------------------
// The three-launch pipeline above costs ~18-28 us on the 4x7 / 8x14 levels of the UNet no matter how
// small the tensor is (three dependent launches, each with its own ramp and tail).  When one
// (batch, group) slice fits 96 KiB of LDS, one block per slice does everything: load once into LDS
// while summing, centred variance from LDS, then normalise + affine + SiLU + 16-bit store (12 us).
// Larger slices stay on the streaming pipeline: a single block per slice is latency-bound (one CU
// pulls ~20 GB/s with 16 KiB in flight; measured 51 us vs 26 us on [2 x 1792 x 1280]).
// Fixed reduction order: wave butterflies, then the wave partials in index order.
constexpr int GNF_THREADS = 512;         // r04 same-box A/B of the whole step: 256 threads +0.3 %, 1024 +0.35 % (profiles/r04f_ab_gnf_threads.jsonl)
constexpr int GNF_LDS_FLOATS = 24576;   // 96 KiB of the CU's 160 KiB

__device__ __forceinline__ float gnf_block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();                       // red[] may still be read from the previous reduction
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < GNF_THREADS / 64; ++i) t += red[i];
  return t;
}

template <typename T>
__global__ __launch_bounds__(GNF_THREADS) void gn_fused_kernel(
    const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2, int64_t S, int groups,
    float eps, const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
    uint16_t* __restrict__ y, uint16_t* __restrict__ raw, int raw_lo) {
  extern __shared__ __attribute__((aligned(16))) float gnf_stage[];
  __shared__ float red[GNF_THREADS / 64];
  const int C = C1 + C2;
  const int cpg = C / groups;
  const int I = cpg >> 1;                       // float2 items per row of this group
  const int rpi = GNF_THREADS / I;              // rows per block iteration
  const int tid = threadIdx.x;
  const int r0 = tid / I;
  const int i = tid - r0 * I;
  const bool active = r0 < rpi;
  const int c = blockIdx.x * cpg + 2 * i;       // this thread's channel pair
  const int64_t row0 = (int64_t)blockIdx.y * S;
  const bool from1 = c < C1;
  const float* src = from1 ? x1 + row0 * C1 + c : x2 + row0 * C2 + (c - C1);
  const int64_t lds = from1 ? C1 : C2;
  const int nit = active ? (int)((S - r0 + rpi - 1) / rpi) : 0;
  f32x2* const st = (f32x2*)gnf_stage + r0 * I + i;   // this thread's column of the staged slice
  const int sstep = rpi * I;

  float a = 0.f;
  {
    const float* p = src + (int64_t)r0 * lds;
    const int64_t step = (int64_t)rpi * lds;
#pragma unroll 8
    for (int k = 0; k < nit; ++k) {
      const f32x2 v = *(const f32x2*)(p + k * step);
      st[k * sstep] = v;
      a += v.x + v.y;
    }
  }
  const float n = (float)S * (float)cpg;
  const float mean = gnf_block_sum(a, red) / n;
  float q = 0.f;
  for (int k = 0; k < nit; ++k) {
    const f32x2 v = st[k * sstep];
    const float d0 = v.x - mean, d1 = v.y - mean;
    q += d0 * d0 + d1 * d1;
  }
  const float rstd = 1.0f / sqrtf(gnf_block_sum(q, red) / n + eps);
  if (!active) return;
  const float sc0 = gamma[c] * rstd, sc1 = gamma[c + 1] * rstd;
  const float sh0 = beta[c] - mean * sc0, sh1 = beta[c + 1] - mean * sc1;
  const int raw_ld = raw_lo > 0 ? 2 * C : C;
  uint16_t* yo = y + (row0 + r0) * C + c;
  uint16_t* ro = raw ? raw + (row0 + r0) * raw_ld + c : nullptr;
  const int64_t ostep = (int64_t)rpi * C, rstep = (int64_t)rpi * raw_ld;
#pragma unroll 4
  for (int k = 0; k < nit; ++k) {
    const f32x2 v = st[k * sstep];
    float o0 = v.x * sc0 + sh0, o1 = v.y * sc1 + sh1;
    if (silu) {
      o0 = silu_f(o0);
      o1 = silu_f(o1);
    }
    *(uint32_t*)(yo + k * ostep) = T::pack2(o0, o1);
    if (ro) store_raw2<T>(ro, k * rstep, raw_lo, v.x, v.y);
  }
}

// ---- single-launch GroupNorm, slice resident in REGISTERS ---------------------------------------------------
// Slices too big for the LDS path above but <= 72 K elements (the 5-D norms of the 8x14 level: [1792 x 40] fp32 =
// 287 KB per (batch, group)) used to take the three-launch streaming pipeline: 36 us for an 18 MB tensor, three
// dependent launches that each re-read it.  A 1024-thread block can hold such a slice in its registers (72 floats
// per thread): load once (the loads ARE the staging), block-reduce the mean, centred variance from the registers,
// normalise + SiLU + store.  One launch, 6 B / element instead of 10.  Fixed reduction order.
constexpr int GNR_THREADS = 1024;
constexpr int GNR_NIT = 36;                       // float2 items per thread

__device__ __forceinline__ float gnr_block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < GNR_THREADS / 64; ++i) t += red[i];
  return t;
}

template <typename T>
__global__ __launch_bounds__(GNR_THREADS) void gn_regs_kernel(
    const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2, int64_t S, int groups,
    float eps, const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
    uint16_t* __restrict__ y, uint16_t* __restrict__ raw, int raw_lo) {
  __shared__ float red[GNR_THREADS / 64];
  const int C = C1 + C2;
  const int cpg = C / groups;
  const int I = cpg >> 1;                       // float2 items per row of this group
  const int rpi = GNR_THREADS / I;              // rows per block iteration
  const int tid = threadIdx.x;
  const int r0 = tid / I;
  const int i = tid - r0 * I;
  const bool active = r0 < rpi;
  const int c = blockIdx.x * cpg + 2 * i;       // this thread's channel pair
  const int64_t row0 = (int64_t)blockIdx.y * S;
  const bool from1 = c < C1;
  const float* src = from1 ? x1 + row0 * C1 + c : x2 + row0 * C2 + (c - C1);
  const int64_t lds = from1 ? C1 : C2;
  const int nit = active ? (int)((S - r0 + rpi - 1) / rpi) : 0;

  f32x2 v[GNR_NIT];
  float a = 0.f;
  {
    const float* p = src + (int64_t)r0 * lds;
    const int64_t step = (int64_t)rpi * lds;
#pragma unroll
    for (int k = 0; k < GNR_NIT; ++k) {
      v[k] = f32x2{0.f, 0.f};
      if (k < nit) v[k] = *(const f32x2*)(p + k * step);
    }
#pragma unroll
    for (int k = 0; k < GNR_NIT; ++k) a += v[k].x + v[k].y;      // rows beyond nit hold zeros
  }
  const float n = (float)S * (float)cpg;
  const float mean = gnr_block_sum(a, red) / n;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < GNR_NIT; ++k) {
    if (k < nit) {
      const float d0 = v[k].x - mean, d1 = v[k].y - mean;
      q += d0 * d0 + d1 * d1;
    }
  }
  const float rstd = 1.0f / sqrtf(gnr_block_sum(q, red) / n + eps);
  if (!active) return;
  const float sc0 = gamma[c] * rstd, sc1 = gamma[c + 1] * rstd;
  const float sh0 = beta[c] - mean * sc0, sh1 = beta[c + 1] - mean * sc1;
  const int raw_ld = raw_lo > 0 ? 2 * C : C;
  uint16_t* yo = y + (row0 + r0) * C + c;
  uint16_t* ro = raw ? raw + (row0 + r0) * raw_ld + c : nullptr;
  const int64_t ostep = (int64_t)rpi * C, rstep = (int64_t)rpi * raw_ld;
#pragma unroll
  for (int k = 0; k < GNR_NIT; ++k) {
    if (k < nit) {
      float o0 = v[k].x * sc0 + sh0, o1 = v[k].y * sc1 + sh1;
      if (silu) {
        o0 = silu_f(o0);
        o1 = silu_f(o1);
      }
      *(uint32_t*)(yo + k * ostep) = T::pack2(o0, o1);
      if (ro) store_raw2<T>(ro, k * rstep, raw_lo, v[k].x, v[k].y);
    }
  }
}

int gn_nsplit(int64_t nb, int64_t S, int C) {
  // ~16 K elements (64 KiB of fp32) per block, but at least ~1024 blocks overall when the tensor
  // allows it (>= 2 rows per block): the 4x7 / 8x14 levels are latency-bound, not bandwidth-bound.
  int64_t rows = 16384 / C;
  if (rows < 2) rows = 2;
  int64_t ns = (S + rows - 1) / rows;
  const int64_t want = (1024 + nb - 1) / nb;
  if (ns < want) ns = want;
  const int64_t cap = (S + 1) / 2;
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  if (ns > 1024) ns = 1024;
  return (int)ns;
}

// ---- LayerNorm: LPR lanes per row (16 / 32 / 64), two-pass from registers ------------------
// d = 320..1280 in the UNet: one 64-lane wave per 1.25 KiB row left most lanes idle and one load in
// flight per wave (1.2 TB/s measured); with 16 lanes per row a wave streams 4 rows at once.
constexpr int LN_MAX_SLOTS = 8;  // float4 slots per lane: d <= 4 * 8 * LPR

template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t M,
                                                        int d, float eps,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        void* __restrict__ y) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int sub = threadIdx.x % LPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  const bool live = row < M;
  const int nslots = d >> 2;
  const float* xr = x + (live ? row : 0) * d;
  f32x4 v[LN_MAX_SLOTS];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX_SLOTS; ++k) {
    const int slot = sub + LPR * k;
    v[k] = f32x4{0, 0, 0, 0};
    if (slot < nslots) {
      v[k] = *(const f32x4*)(xr + slot * 4);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)d;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX_SLOTS; ++k) {
    const int slot = sub + LPR * k;
    if (slot < nslots) {
      const f32x4 c = v[k] - mean;
      q += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)d + eps);
  if (!live) return;
#pragma unroll
  for (int k = 0; k < LN_MAX_SLOTS; ++k) {
    const int slot = sub + LPR * k;
    if (slot < nslots) {
      const f32x4 ga = *(const f32x4*)(gamma + slot * 4);
      const f32x4 be = *(const f32x4*)(beta + slot * 4);
      const f32x4 o = (v[k] - mean) * rstd * ga + be;
      if constexpr (std::is_same<T, F32Out>::value) {
        *(f32x4*)((float*)y + row * d + slot * 4) = o;           // dtype VGEN_F32: unrounded (ln_final of the text tower)
      } else {
        *(u32x2*)((uint16_t*)y + row * d + slot * 4) = pack4<T>(o.x, o.y, o.z, o.w);
      }
    }
  }
}

// Streaming variant for the UNet's own widths (d = NS * 4 * LPR exactly: 320/640/1280 with NS = 5,
// 512/1024/2048 with NS = 8): a fixed grid of blocks walks the row groups with a register
// double buffer — the loads of the next row group are in flight while the current one is reduced
// and stored — and gamma / beta live in registers.  The one-shot kernel above launches one short
// wave per 4 rows (14 K waves at M = 57344): 2.85 TB/s; wave turnover, not bandwidth, was the bound.
template <typename T, int LPR, int NS>
__global__ __launch_bounds__(256) void layernorm_stream_kernel(const float* __restrict__ x, int64_t M,
                                                               float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               uint16_t* __restrict__ y) {
  constexpr int RPB = 256 / LPR;
  constexpr int D = NS * 4 * LPR;
  const int sub = threadIdx.x % LPR;
  const int rl = threadIdx.x / LPR;
  const int64_t ngroups = (M + RPB - 1) / RPB;
  f32x4 ga[NS], be[NS], v[NS], nx[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    ga[k] = *(const f32x4*)(gamma + (sub + LPR * k) * 4);
    be[k] = *(const f32x4*)(beta + (sub + LPR * k) * 4);
  }
  int64_t g = blockIdx.x;
  auto load = [&](int64_t grp, f32x4* dst) __attribute__((always_inline)) {
    const int64_t row = grp * RPB + rl;
    const float* xr = x + (row < M ? row : M - 1) * D;     // clamp: tail rows reload the last row, never stored
#pragma unroll
    for (int k = 0; k < NS; ++k) dst[k] = *(const f32x4*)(xr + (sub + LPR * k) * 4);
  };
  if (g < ngroups) load(g, nx);
  for (; g < ngroups; g += gridDim.x) {
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = nx[k];
    if (g + gridDim.x < ngroups) load(g + gridDim.x, nx);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const f32x4 c = v[k] - mean;
      q += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    const int64_t row = g * RPB + rl;
    if (row < M) {
      uint16_t* yr = y + row * D;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const f32x4 o = (v[k] - mean) * rstd * ga[k] + be[k];
        *(u32x2*)(yr + (sub + LPR * k) * 4) = pack4<T>(o.x, o.y, o.z, o.w);
      }
    }
  }
}

template <typename T, int LPR, int NS>
void launch_ln_stream(const float* x, int64_t M, float eps, const float* gamma, const float* beta,
                      uint16_t* y, hipStream_t s) {
  const int64_t ngroups = (M + 256 / LPR - 1) / (256 / LPR);
  const int64_t grid = ngroups < 2048 ? ngroups : 2048;     // 8 blocks per CU
  hipLaunchKernelGGL((layernorm_stream_kernel<T, LPR, NS>), dim3((unsigned)grid), dim3(256), 0, s, x, M, eps,
                     gamma, beta, y);
}

template <typename T, int LPR>
void launch_ln(const float* x, int64_t M, int d, float eps, const float* gamma, const float* beta,
               void* y, hipStream_t s) {
  const int64_t grid = (M + 256 / LPR - 1) / (256 / LPR);
  hipLaunchKernelGGL((layernorm_kernel<T, LPR>), dim3((unsigned)grid), dim3(256), 0, s, x, M, d, eps,
                     gamma, beta, y);
}

template <typename T>
void dispatch_ln(const float* x, int64_t M, int d, float eps, const float* gamma, const float* beta,
                 uint16_t* y, hipStream_t s) {
  switch (d) {   // exact-width streaming kernels
    case 320: return launch_ln_stream<T, 16, 5>(x, M, eps, gamma, beta, y, s);
    case 512: return launch_ln_stream<T, 16, 8>(x, M, eps, gamma, beta, y, s);
    case 640: return launch_ln_stream<T, 32, 5>(x, M, eps, gamma, beta, y, s);
    case 1024: return launch_ln_stream<T, 32, 8>(x, M, eps, gamma, beta, y, s);
    case 1280: return launch_ln_stream<T, 64, 5>(x, M, eps, gamma, beta, y, s);
    case 2048: return launch_ln_stream<T, 64, 8>(x, M, eps, gamma, beta, y, s);
    default: break;
  }
  if (d <= 512) launch_ln<T, 16>(x, M, d, eps, gamma, beta, y, s);
  else if (d <= 1024) launch_ln<T, 32>(x, M, d, eps, gamma, beta, y, s);
  else launch_ln<T, 64>(x, M, d, eps, gamma, beta, y, s);
}

}  // namespace

extern "C" size_t vgen_groupnorm_ws_bytes(int64_t nb, int64_t S) {
  const int ns = 1024;   // upper bound of gn_nsplit (independent of C so callers need not pass it)
  return (size_t)(nb * ns * GN_G * 3 + nb * 3072 * 2) * sizeof(float);   // partials + per-channel (scale, shift)
}

static int groupnorm_impl(const float* x1, int32_t C1, const float* cs1, const float* x2, int32_t C2,
                          const float* cs2, int64_t nb, int64_t S, int32_t groups, float eps,
                          const float* gamma, const float* beta, int32_t silu, void* y,
                          void* raw, int32_t raw_split, int32_t dtype, float* ws, size_t ws_bytes,
                          void* stream) {
  const int C = C1 + C2;
  const int raw_lo = (raw != nullptr && raw_split) ? C : 0;
  VGEN_REQUIRE(dtype == VGEN_BF16 || dtype == VGEN_F16, "groupnorm: dtype");
  VGEN_REQUIRE(groups > 0 && groups <= GN_G && C % groups == 0, "groupnorm: C=%d groups=%d", C,
               groups);
  VGEN_REQUIRE(C1 > 0 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0 && C <= 3072,
               "groupnorm: C1=%d C2=%d (need %%4, total <= 3072)", C1, C2);
  VGEN_REQUIRE(C2 == 0 || x2 != nullptr, "groupnorm: x2 null with C2 > 0");
  VGEN_REQUIRE(vgen_aligned16(x1) && vgen_aligned16(x2) && vgen_aligned16(y) &&
                   vgen_aligned16(raw) && vgen_aligned16(ws),
               "groupnorm: alignment");
  VGEN_REQUIRE(nb > 0 && S > 0 && nb <= 65535, "groupnorm: nb=%lld S=%lld", (long long)nb,
               (long long)S);
  if (ws_bytes < vgen_groupnorm_ws_bytes(nb, S)) {
    vgen_set_error("groupnorm: workspace %zu < %zu", ws_bytes, vgen_groupnorm_ws_bytes(nb, S));
    return VGEN_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  {
    // tuning switch (not part of the ABI): VGEN_GN_FUSED_MAX_MB (0 keeps everything on the streaming
    // pipeline).  A slice row is only cpg * 4 bytes: at C = 320 (40 B) every 128-byte line is fetched by
    // 3-4 blocks and big tensors lose (76 vs 53 us on [32 x 1792 x 320]); from 80-byte rows on the
    // single launch wins as long as the slice fits the LDS (24 vs 40 us on [32 x 448 x 640]).
    static const int env_max = getenv("VGEN_GN_FUSED_MAX_MB") ? atoi(getenv("VGEN_GN_FUSED_MAX_MB")) : -1;
    const int64_t fused_max = (int64_t)(env_max >= 0 ? env_max : (C / groups >= 16 ? 96 : 24)) << 20;
    const int cpg = C / groups;
    if (nb * S * C * 4 <= fused_max && cpg % 2 == 0 && C1 % 2 == 0 && cpg / 2 <= GNF_THREADS && S * cpg <= GNF_LDS_FLOATS) {
      const size_t lds = (size_t)S * cpg * sizeof(float);
      static bool attr_done_dev[VGEN_MAX_DEVICES] = {false};
      bool& attr_done = attr_done_dev[vgen_device_slot()];
      if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute((const void*)gn_fused_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            GNF_LDS_FLOATS * 4);
        hipError_t e2 = hipFuncSetAttribute((const void*)gn_fused_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            GNF_LDS_FLOATS * 4);
        if (e1 != hipSuccess || e2 != hipSuccess) {
          vgen_set_error("groupnorm: hipFuncSetAttribute(LDS) failed");
          return VGEN_E_BADARG;
        }
        attr_done = true;
      }
      dim3 fgrid((unsigned)groups, (unsigned)nb);
      if (dtype == VGEN_BF16) {
        hipLaunchKernelGGL(gn_fused_kernel<BF16>, fgrid, dim3(GNF_THREADS), lds, s, x1, C1, x2, C2, S, groups, eps,
                           gamma, beta, silu, (uint16_t*)y, (uint16_t*)raw, raw_lo);
      } else {
        hipLaunchKernelGGL(gn_fused_kernel<F16>, fgrid, dim3(GNF_THREADS), lds, s, x1, C1, x2, C2, S, groups, eps,
                           gamma, beta, silu, (uint16_t*)y, (uint16_t*)raw, raw_lo);
      }
      return vgen_check_launch("gn_fused");
    }
  }
  {
    // register-resident single launch: slices that missed the LDS path but fit 72 floats x 1024 threads and are few
    // enough that one block per slice is not the bottleneck (tuning switch: VGEN_GN_REGS=0 disables)
    static const int regs_on = getenv("VGEN_GN_REGS") ? atoi(getenv("VGEN_GN_REGS")) : 1;
    const int cpg = C / groups;
    const int I = cpg / 2;
    if (regs_on && cs1 == nullptr && cpg % 2 == 0 && C1 % 2 == 0 && I > 0 && I <= GNR_THREADS &&
        (S + (GNR_THREADS / I) - 1) / (GNR_THREADS / I) <= GNR_NIT && nb * groups <= 1024 && S * cpg > GNF_LDS_FLOATS) {
      dim3 fgrid((unsigned)groups, (unsigned)nb);
      if (dtype == VGEN_BF16) {
        hipLaunchKernelGGL(gn_regs_kernel<BF16>, fgrid, dim3(GNR_THREADS), 0, s, x1, C1, x2, C2, S, groups, eps, gamma, beta,
                           silu, (uint16_t*)y, (uint16_t*)raw, raw_lo);
      } else {
        hipLaunchKernelGGL(gn_regs_kernel<F16>, fgrid, dim3(GNR_THREADS), 0, s, x1, C1, x2, C2, S, groups, eps, gamma, beta,
                           silu, (uint16_t*)y, (uint16_t*)raw, raw_lo);
      }
      return vgen_check_launch("gn_regs");
    }
  }
  // rpb * C must fit the LDS staging of gn_stats (3072 floats per plane)
  const int nslots = C / 4;
  const int rpb = nslots <= GN_THREADS ? GN_THREADS / nslots : 1;
  VGEN_REQUIRE(rpb * C <= 3072, "groupnorm: internal LDS bound");
  const int ns = gn_nsplit(nb, S, C);
  float* part = ws;
  float* stat = ws + nb * ns * GN_G * 3;
  dim3 grid((unsigned)ns, (unsigned)nb);
  int rc;
  if (cs1 != nullptr) {
    if ((S / 64) * (C / groups) > 2048) {
      hipLaunchKernelGGL(gn_finalize_cs_kernel<1024>, dim3((unsigned)groups, (unsigned)nb), dim3(1024), 0, s, cs1, C1,
                         cs2, C2, S, groups, eps, gamma, beta, stat);
    } else {
      hipLaunchKernelGGL(gn_finalize_cs_kernel<256>, dim3((unsigned)groups, (unsigned)nb), dim3(256), 0, s, cs1, C1,
                         cs2, C2, S, groups, eps, gamma, beta, stat);
    }
    rc = vgen_check_launch("gn_finalize_cs");
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(GN_THREADS), 0, s, x1, C1, x2, C2, S, groups, ns,
                       part);
    rc = vgen_check_launch("gn_stats");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)groups, (unsigned)nb), dim3(64), 0, s, part,
                       groups, ns, eps, C, gamma, beta, stat);
    rc = vgen_check_launch("gn_finalize");
    if (rc) return rc;
  }
  if (dtype == VGEN_BF16) {
    hipLaunchKernelGGL(gn_apply_kernel<BF16>, grid, dim3(GN_THREADS), 0, s, x1, C1, x2, C2, S,
                       groups, ns, stat, gamma, beta, silu, (uint16_t*)y, (uint16_t*)raw, raw_lo);
  } else {
    hipLaunchKernelGGL(gn_apply_kernel<F16>, grid, dim3(GN_THREADS), 0, s, x1, C1, x2, C2, S,
                       groups, ns, stat, gamma, beta, silu, (uint16_t*)y, (uint16_t*)raw, raw_lo);
  }
  return vgen_check_launch("gn_apply");
}

extern "C" int vgen_groupnorm(const float* x1, int32_t C1, const float* x2, int32_t C2,
                              int64_t nb, int64_t S, int32_t groups, float eps,
                              const float* gamma, const float* beta, int32_t silu, void* y,
                              void* raw, int32_t raw_split, int32_t dtype, float* ws, size_t ws_bytes,
                              void* stream) {
  return groupnorm_impl(x1, C1, nullptr, x2, C2, nullptr, nb, S, groups, eps, gamma, beta, silu, y, raw, raw_split,
                        dtype, ws, ws_bytes, stream);
}

extern "C" int vgen_groupnorm_cs(const float* x1, int32_t C1, const float* cs1, const float* x2,
                                 int32_t C2, const float* cs2, int64_t nb, int64_t S, int32_t groups,
                                 float eps, const float* gamma, const float* beta, int32_t silu,
                                 void* y, void* raw, int32_t raw_split, int32_t dtype, float* ws, size_t ws_bytes,
                                 void* stream) {
  VGEN_REQUIRE(cs1 != nullptr && (C2 == 0 || cs2 != nullptr), "groupnorm_cs: missing column statistics");
  VGEN_REQUIRE(S % 64 == 0, "groupnorm_cs: S=%lld must be a multiple of the 64-row slab", (long long)S);
  return groupnorm_impl(x1, C1, cs1, x2, C2, cs2, nb, S, groups, eps, gamma, beta, silu, y, raw, raw_split, dtype,
                        ws, ws_bytes, stream);
}

extern "C" int vgen_layernorm(const float* x, int64_t M, int32_t d, float eps, const float* gamma,
                              const float* beta, void* y, int32_t dtype, void* stream) {
  VGEN_REQUIRE(dtype == VGEN_BF16 || dtype == VGEN_F16 || dtype == VGEN_F32, "layernorm: dtype");
  VGEN_REQUIRE(d > 0 && d % 4 == 0 && d <= 64 * 4 * LN_MAX_SLOTS, "layernorm: d=%d", d);
  VGEN_REQUIRE(vgen_aligned16(x) && vgen_aligned16(y) && vgen_aligned16(gamma) &&
                   vgen_aligned16(beta),
               "layernorm: alignment");
  if (M <= 0) return 0;
  VGEN_REQUIRE(M < (1LL << 32), "layernorm: M too large");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VGEN_F32) {
    if (d <= 512) launch_ln<F32Out, 16>(x, M, d, eps, gamma, beta, y, s);
    else if (d <= 1024) launch_ln<F32Out, 32>(x, M, d, eps, gamma, beta, y, s);
    else launch_ln<F32Out, 64>(x, M, d, eps, gamma, beta, y, s);
    return vgen_check_launch("layernorm");
  }
  if (dtype == VGEN_BF16) dispatch_ln<BF16>(x, M, d, eps, gamma, beta, (uint16_t*)y, s);
  else dispatch_ln<F16>(x, M, d, eps, gamma, beta, (uint16_t*)y, s);
  return vgen_check_launch("layernorm");
}
