// attention.hip — fused softmax(Q K^T * scale) V for head_dim 64 on gfx950 MFMA.
//
// Two kernels (see include/vgen_hip.h for the strided sequence addressing):
//
//  flash_kernel    : general nq x nk (spatial self-attention 1792/448/112/28 tokens, cross
//                    attention over 77 context tokens).  Block = 4 waves x 32 queries, KV tiles
//                    of 64 keys staged through LDS, online softmax in fp32.
//                    Both products are computed TRANSPOSED so that every lane owns exactly one
//                    query column:
//                      S^T[key, q]  = mfma(A = K-frag [16 keys x 32 d], B = Q-frag [32 d x 16 q])
//                      O^T[d, q]   += mfma(A = V^T-frag [16 d x 32 keys], B = P^T-frag)
//                    -> the softmax max/sum are in-lane over 16 scores plus two xor-shuffles,
//                       the rescale factor is one scalar per lane, and the C/D layout of S^T
//                       (4 consecutive keys per lane) IS the B-operand layout of the PV product
//                       (any k-permutation is legal as long as A and B agree), so P never
//                       leaves registers.  V sits in LDS row-major like K; the matching A fragments
//                       (V^T) are two ds_read_b64_tr_b16 each (gfx950's transpose read, r04).
//
//  temporal_kernel : nq, nk <= 16 (attention over the 16 frames of one pixel; batch = B*H*W,
//                    up to 8960 sequences x heads).  One wave per (sequence, head): Q/K
//                    fragments are loaded straight from global in MFMA operand layout, S^T is
//                    2 MFMA 16x16x32, the softmax is in-lane + 2 shuffles, and P V uses the
//                    K=16 MFMA (16x16x16) whose B layout again equals S^T's C/D layout.
#include "common.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "attention.hip uses gfx950-only instructions (ds_read_b64_tr_b16): build with --offload-arch=gfx950"
#endif
#include <stdlib.h>

namespace {

constexpr int HD = 64;  // head dim

// gfx950's LDS transpose read (ds_read_b64_tr_b16), semantics measured with tools/probes/tr16_probe.hip
// (profiles/r04_tr16_probe.txt): inside every 16-lane group, source lane s = 4 j + q hands in the address of 4 consecutive
// 16-bit elements = row j, columns 4 q .. 4 q + 3 of a [4][16] block; destination lane c = 4 q + e receives column c of that
// block, rows 0 .. 3, in its 4 halves.  For V stored ROW-MAJOR [key][d] that is exactly the A fragment of the O^T = V^T P^T
// product (lane -> one d, 4 consecutive keys): the transposition r03 did with DPP moves in the staging pass (8 x ~5 VALU per
// thread and tile in a VALU-bound kernel) is free here.
typedef short v4i16_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 lds_read_tr16(const uint16_t* p) {
  const v4i16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)p);
  u32x2 o;
  __builtin_memcpy(&o, &r, 8);
  return o;
}

__device__ __forceinline__ int64_t seq_off(int64_t bi, int64_t inner, int64_t bo, int64_t bin) {
  return (bi / inner) * bo + (bi % inner) * bin;
}

// =========================================================================================
template <typename T>
__global__ __launch_bounds__(256, 2) void flash_kernel(const vgen_attn_args p, int qtiles) {
  constexpr int BQ = 128, BKV = 64;
  constexpr int VS = 80;  // V row stride in elements (64 d + 16 pad) -> 160 B: the 8 rows x 32 B a half-wave's transpose
                          // read touches then cover all 64 banks exactly once (row offsets 0, 40, 16, 56, 32, 8, 48, 24 words)
  // double-buffered K / V tiles: the global loads of tile t+1 are issued before the MFMAs of
  // tile t and written to the other buffer after them -> one barrier per tile, HBM/L2 latency
  // hidden behind the compute (v1 was single-buffered with two barriers: latency-bound, 250 TF/s).
  __shared__ __attribute__((aligned(16))) unsigned char sK[2][BKV * 128];   // [key][64 d] swizzled
  __shared__ __attribute__((aligned(16))) uint16_t sV[2][BKV * VS];         // [key][d] row-major (r04: was [d][key])

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;

  int64_t bid = blockIdx.x;
  const int qt = (int)(bid % qtiles);
  bid /= qtiles;
  const int h = (int)(bid % p.heads);
  const int64_t bi = bid / p.heads;

  const uint16_t* Q = (const uint16_t*)p.q + seq_off(bi, p.inner, p.q_bo, p.q_bi) + h * HD;
  const uint16_t* K = (const uint16_t*)p.k + seq_off(bi, p.inner, p.k_bo, p.k_bi) + h * HD;
  const uint16_t* V = (const uint16_t*)p.v + seq_off(bi, p.inner, p.v_bo, p.v_bi) + h * HD;
  uint16_t* O = (uint16_t*)p.out + seq_off(bi, p.inner, p.o_bo, p.o_bi) + h * HD;

  const int q0 = qt * BQ + wave * 32;

  // Q fragments (B operand): lane (lq, lr): Q[q0 + qf*16 + lr][32*ks + 8*lq .. +8]
  u32x4 qf[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = q0 + f * 16 + lr;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < p.nq) v = *(const u32x4*)(Q + (int64_t)row * p.q_rs + ks * 32 + lq * 8);
      qf[f][ks] = v;
    }

  f32x4 o_acc[2][4];
  float m_run[2], l_run[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    m_run[f] = -INFINITY;
    l_run[f] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o_acc[f][d] = f32x4{0, 0, 0, 0};
  }
  const float c = p.scale * 1.44269504088896340736f;  // scores -> log2 domain

  // staging assignments
  const int k_row = tid >> 2, k_ch = (tid & 3) * 2;      // K: row, two 16-B chunks
  const int v_key = tid & 63, v_d0 = (tid >> 6) * 16;    // V: key, 16 d values (two 16-B chunks)

  u32x4 rk0, rk1, rv0, rv1;
  auto gload = [&](int kv0) {
    rk0 = rk1 = rv0 = rv1 = u32x4{0u, 0u, 0u, 0u};
    if (kv0 + k_row < p.nk) {
      const uint16_t* src = K + (int64_t)(kv0 + k_row) * p.k_rs + k_ch * 8;
      rk0 = *(const u32x4*)src;
      rk1 = *(const u32x4*)(src + 8);
    }
    if (kv0 + v_key < p.nk) {
      const uint16_t* src = V + (int64_t)(kv0 + v_key) * p.v_rs + v_d0;
      rv0 = *(const u32x4*)src;
      rv1 = *(const u32x4*)(src + 8);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* dst = sK[buf] + k_row * 128;
    *(u32x4*)(dst + (((k_ch) ^ (k_row & 7)) << 4)) = rk0;
    *(u32x4*)(dst + (((k_ch + 1) ^ (k_row & 7)) << 4)) = rk1;
    // V row-major: two 16-byte stores (r03 transposed here: 8 DPP moves + bit merges + 8 ds_write_b32 per thread)
    uint16_t* vd = sV[buf] + v_key * VS + v_d0;
    *(u32x4*)vd = rv0;
    *(u32x4*)(vd + 8) = rv1;
  };

  const int ntile = (p.nk + BKV - 1) / BKV;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int kv0 = t * BKV;
    const int buf = t & 1;
    const bool more = t + 1 < ntile;
    if (more) gload(kv0 + BKV);
    const unsigned char* cK = sK[buf];
    const uint16_t* cV = sV[buf];

    // ---- S^T = K Q^T for both query fragments; K fragments shared ------------------------
    f32x4 s[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) s[f][kf] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ((ks * 4 + lq) ^ (lr & 7)) << 4;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        const u32x4 kfrag = *(const u32x4*)(cK + (kf * 16 + lr) * 128 + co);
        s[0][kf] = T::mfma32(kfrag, qf[0][ks], s[0][kf]);
        s[1][kf] = T::mfma32(kfrag, qf[1][ks], s[1][kf]);
      }
    }

    // ---- online softmax; lane (lq, lr) holds keys kv0 + 16*kf + 4*lq + r of query lr ------
    const bool ragged = kv0 + BKV > p.nk;
    // causal (CLIP text tower): key j is visible to query i iff j <= i.  Only tiles that reach past the wave's
    // first query need the mask (wave-uniform test); key 0 is always visible, so every row keeps a finite maximum.
    const bool diag = p.causal && kv0 + BKV - 1 > q0;
    u32x4 pb[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (ragged) {   // block-uniform: only the last KV tile of a ragged sequence pays for the mask
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s[f][kf][r] = (kv0 + kf * 16 + lq * 4 + r < p.nk) ? s[f][kf][r] : -INFINITY;
      }
      if (diag) {
        const int qi = q0 + f * 16 + lr;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s[f][kf][r] = (kv0 + kf * 16 + lq * 4 + r <= qi) ? s[f][kf][r] : -INFINITY;
      }
      // 16 scores -> 8 three-input maxima (v_max3_f32) instead of 16 two-input ones: the softmax's VALU slots, not the 32
      // MFMAs, bound this kernel
      float mx = fmaxf(s[f][0][0], s[f][0][1]);
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        if (kf > 0) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[f][kf][0]), s[f][kf][1]);
        mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[f][kf][2]), s[f][kf][3]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[f], mx);  // finite: every tile has >= 1 valid key
      // The running maximum settles after the first few KV tiles; when NO lane of the wave saw a larger score the
      // rescale factor is exp2(0) = 1 for every query: skip its exponential and the 18 multiplies (wave-uniform
      // branch; the result is bit-identical either way).
      const bool grew = __builtin_amdgcn_ballot_w64(m_new > m_run[f]) != 0ull;
      const float mc = m_new * c;
      float pv[4][4];
      // scores -> probabilities two at a time: v_pk_fma_f32 / v_pk_add_f32 do the scale-and-shift and the row sum in half
      // the issue slots (the softmax, not the 32 MFMAs, bounds this kernel: ~350 VALU issue slots per KV tile and wave,
      // the 32 quarter-rate v_exp_f32 included).  Same-box A/B, whole step: 42.05 -> 41.81 ms (high), 31.40 -> 31.15 ms
      // (fast): profiles/r03b_ab_libs.jsonl
      const f32x2 c2 = {c, c}, mc2 = {mc, mc};
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        const f32x2 a = f32x2{s[f][kf][0], s[f][kf][1]} * c2 - mc2;
        const f32x2 b = f32x2{s[f][kf][2], s[f][kf][3]} * c2 - mc2;
        const f32x2 ea = {fast_exp2(a.x), fast_exp2(a.y)};
        const f32x2 eb = {fast_exp2(b.x), fast_exp2(b.y)};
        pv[kf][0] = ea.x;
        pv[kf][1] = ea.y;
        pv[kf][2] = eb.x;
        pv[kf][3] = eb.y;
        ps2 += ea;
        ps2 += eb;
      }
      const float psum = ps2.x + ps2.y;
      if (grew) {
        const float alpha = fast_exp2((m_run[f] - m_new) * c);
        m_run[f] = m_new;
        l_run[f] = l_run[f] * alpha + psum;
#pragma unroll
        for (int d = 0; d < 4; ++d) o_acc[f][d] *= alpha;
      } else {
        l_run[f] += psum;
      }
      // B operand of the PV product for key-step st: keys {32st + 4lq + r} U {32st + 16 + 4lq + r}
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        u32x4 tt;
        tt.x = pack2<T>(pv[2 * st][0], pv[2 * st][1]);
        tt.y = pack2<T>(pv[2 * st][2], pv[2 * st][3]);
        tt.z = pack2<T>(pv[2 * st + 1][0], pv[2 * st + 1][1]);
        tt.w = pack2<T>(pv[2 * st + 1][2], pv[2 * st + 1][3]);
        pb[f][st] = tt;
      }
    }

    // ---- O^T += V^T P^T ------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        // A fragment V^T[d = 16 d + lr][keys 32 st + 4 lq + j | 32 st + 16 + 4 lq + j]: two transpose reads of [4 keys][16 d]
        // blocks; as a SOURCE lane this lane supplies row (lr >> 2), columns 4 (lr & 3) .. + 3 of its group's block
        const uint16_t* va = cV + (st * 32 + lq * 4 + (lr >> 2)) * VS + d * 16 + (lr & 3) * 4;
        const u32x2 lo = lds_read_tr16(va);
        const u32x2 hi = lds_read_tr16(va + 16 * VS);
        const u32x4 vfrag = {lo.x, lo.y, hi.x, hi.y};
        o_acc[0][d] = T::mfma32(vfrag, pb[0][st], o_acc[0][d]);
        o_acc[1][d] = T::mfma32(vfrag, pb[1][st], o_acc[1][d]);
      }

    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane (lq, lr) owns O[q = lr][d = 16*dd + 4*lq + r] -------------
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    float l = l_run[f];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = q0 + f * 16 + lr;
    if (row < p.nq) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const f32x4 o = o_acc[f][d] * inv;
        *(u32x2*)(O + (int64_t)row * p.o_rs + d * 16 + lq * 4) = pack4<T>(o.x, o.y, o.z, o.w);
      }
    }
  }
}


// =========================================================================================
// History: r01 saw this kernel drift from run to run under `__launch_bounds__(256, 2)` and kept the default bounds
// without a cause.  Root cause (r02, tools/determinism_probe.py located the first differing launch, the ISA showed
// it): BF16::pack2 was an inline-asm `v_cvt_pk_bf16_f32`; with min-2-blocks bounds the MFMA results stay in VGPRs
// and the asm consumed them straight after `v_mfma_f32_16x16x16_bf16` — the compiler's hazard recogniser does not
// look inside asm blocks, so none of the required wait states were inserted and the conversion read the registers
// before the matrix pipe had written them (stale / NaN values).  With AGPR accumulators the v_accvgpr_read in
// between hid the latency.  pack2 is now a vector conversion the compiler lowers (common.h); both bounds are
// bit-reproducible (profiles/r02_temporal_minb2_*.log).
template <typename T, int MINB>
__global__ __launch_bounds__(256, MINB) void temporal_kernel(const vgen_attn_args p, int64_t npairs) {
  constexpr int VS = 72;
  __shared__ __attribute__((aligned(16))) uint16_t sV[4][16 * VS];  // per wave: [key][64 d + pad]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  const bool live = pair < npairs;
  const int64_t pr = live ? pair : 0;
  const int h = (int)(pr % p.heads);
  const int64_t bi = pr / p.heads;

  const uint16_t* Q = (const uint16_t*)p.q + seq_off(bi, p.inner, p.q_bo, p.q_bi) + h * HD;
  const uint16_t* K = (const uint16_t*)p.k + seq_off(bi, p.inner, p.k_bo, p.k_bi) + h * HD;
  const uint16_t* V = (const uint16_t*)p.v + seq_off(bi, p.inner, p.v_bo, p.v_bi) + h * HD;
  uint16_t* O = (uint16_t*)p.out + seq_off(bi, p.inner, p.o_bo, p.o_bi) + h * HD;

  // operand fragments straight from global: lane (lq, lr): X[lr][32*ks + 8*lq .. +8]
  u32x4 qf[2], kf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
    if (live && lr < p.nq) a = *(const u32x4*)(Q + (int64_t)lr * p.q_rs + ks * 32 + lq * 8);
    if (live && lr < p.nk) b = *(const u32x4*)(K + (int64_t)lr * p.k_rs + ks * 32 + lq * 8);
    qf[ks] = a;
    kf[ks] = b;
  }
  // V tile -> LDS row-major: lane -> key = lane >> 2, chunks 2*(lane&3), +1
  {
    const int key = lane >> 2, ch = (lane & 3) * 2;
    u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
    if (live && key < p.nk) {
      const uint16_t* src = V + (int64_t)key * p.v_rs + ch * 8;
      a = *(const u32x4*)src;
      b = *(const u32x4*)(src + 8);
    }
    uint16_t* dst = &sV[wave][key * VS + ch * 8];
    *(u32x4*)dst = a;
    *(u32x4*)(dst + 8) = b;
  }
  __syncthreads();

  f32x4 s = {0, 0, 0, 0};
  s = T::mfma32(kf[0], qf[0], s);
  s = T::mfma32(kf[1], qf[1], s);  // lane: S^T[key = 4*lq + r][query = lr]

  const float c = p.scale * 1.44269504088896340736f;
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = (lq * 4 + r) < p.nk ? s[r] : -INFINITY;
    s[r] = v;
    mx = fmaxf(mx, v);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float e[4], sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    e[r] = exp2f((s[r] - mx) * c);
    sum += e[r];
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const u32x2 pb = pack4<T>(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);

#pragma unroll
  for (int d = 0; d < 4; ++d) {
    // A operand: V^T[d*16 + lr][key = 4*lq + r] — one transpose read of the [4 keys][16 d] block of this 16-lane group
    // (r03: four ds_read_u16 + two merges per fragment)
    const u32x2 vfrag = lds_read_tr16(&sV[wave][(lq * 4 + (lr >> 2)) * VS + d * 16 + (lr & 3) * 4]);
    f32x4 o = {0, 0, 0, 0};
    o = T::mfma16(vfrag, pb, o);  // O^T[d*16 + 4*lq + r][query = lr]
    if (live && lr < p.nq)
      *(u32x2*)(O + (int64_t)lr * p.o_rs + d * 16 + lq * 4) = pack4<T>(o.x, o.y, o.z, o.w);
  }
}

}  // namespace

extern "C" int vgen_attention(const vgen_attn_args* args, void* stream) {
  if (!args) {
    vgen_set_error("attention: null args");
    return VGEN_E_BADARG;
  }
  const vgen_attn_args& a = *args;
  VGEN_REQUIRE(a.dtype == VGEN_BF16 || a.dtype == VGEN_F16, "attention: dtype");
  VGEN_REQUIRE(a.heads > 0 && a.nq > 0 && a.nk > 0 && a.nbatch > 0 && a.inner > 0,
               "attention: bad sizes");
  VGEN_REQUIRE(vgen_aligned16(a.q) && vgen_aligned16(a.k) && vgen_aligned16(a.v) &&
                   vgen_aligned16(a.out),
               "attention: pointer alignment");
  VGEN_REQUIRE((a.q_rs | a.q_bo | a.q_bi | a.k_rs | a.k_bo | a.k_bi | a.v_rs | a.v_bo | a.v_bi |
                a.o_rs | a.o_bo | a.o_bi) % 8 == 0,
               "attention: strides must be multiples of 8 elements");
  hipStream_t s = (hipStream_t)stream;
  VGEN_REQUIRE(a.causal == 0 || a.causal == 1, "attention: causal flag");
  if (a.nq <= 16 && a.nk <= 16 && !a.causal) {
    const int64_t npairs = a.nbatch * a.heads;
    const int64_t grid = (npairs + 3) / 4;
    VGEN_REQUIRE(grid < (1LL << 31), "attention: grid too large");
    // min-2-blocks launch bounds keep the MFMA results in VGPRs (no v_accvgpr round trip): 34.62 vs 34.72 ms / step.
    // tuning / diagnosis switch (not part of the ABI): VGEN_TEMPORAL_MINB=1 runs the AGPR instantiation
    static const int minb = getenv("VGEN_TEMPORAL_MINB") ? atoi(getenv("VGEN_TEMPORAL_MINB")) : 2;
    if (minb == 2) {
      if (a.dtype == VGEN_BF16)
        hipLaunchKernelGGL((temporal_kernel<BF16, 2>), dim3((unsigned)grid), dim3(256), 0, s, a, npairs);
      else
        hipLaunchKernelGGL((temporal_kernel<F16, 2>), dim3((unsigned)grid), dim3(256), 0, s, a, npairs);
    } else if (a.dtype == VGEN_BF16)
      hipLaunchKernelGGL((temporal_kernel<BF16, 1>), dim3((unsigned)grid), dim3(256), 0, s, a, npairs);
    else
      hipLaunchKernelGGL((temporal_kernel<F16, 1>), dim3((unsigned)grid), dim3(256), 0, s, a, npairs);
    return vgen_check_launch("attention(temporal)");
  }
  const int qtiles = (a.nq + 127) / 128;
  const int64_t grid = a.nbatch * a.heads * qtiles;
  VGEN_REQUIRE(grid < (1LL << 31), "attention: grid too large");
  if (a.dtype == VGEN_BF16)
    hipLaunchKernelGGL(flash_kernel<BF16>, dim3((unsigned)grid), dim3(256), 0, s, a, qtiles);
  else
    hipLaunchKernelGGL(flash_kernel<F16>, dim3((unsigned)grid), dim3(256), 0, s, a, qtiles);
  return vgen_check_launch("attention(flash)");
}

// =========================================================================================
// Row softmax for the VAE's single-head 512-channel attention (scores via tap-GEMM).
// r04: a fused flash kernel for that attention (one head of 512 channels, V^T operand, O accumulators split over the waves
// by channel: vgen_attention_d512, commit d51c7ab) was built, passed its parity cases and measured 57 TFLOP/s — 0.23 ms for
// two 1 792-token frames against ~0.11 ms for this scores-GEMM -> softmax -> PV-GEMM sequence
// (profiles/r04c_vae_shapes_256x448_high_fused_attention.json): a 512-wide output row leaves room for only 64 queries per
// block, so every block re-stages the whole K / V^T stream (64 KB per 32 keys) for 8 MFLOP of work and is bound by the
// load issue, not by the matrix pipe.  The GEMM formulation tiles the same products 256 x 128; the kernel was removed again.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S,
                                                           int cols, int64_t lds, float scale,
                                                           uint16_t* __restrict__ P, int64_t ldp) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float* sr = S + row * lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, sr[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float cs = scale * 1.44269504088896340736f;
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) sum += exp2f((sr[c] - mx) * cs);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / sum;
  uint16_t* pr = P + row * ldp;
  for (int c = tid; c < cols; c += 256) pr[c] = T::from_f32(exp2f((sr[c] - mx) * cs) * inv);
}
}  // namespace

extern "C" int vgen_softmax_rows(const float* S, int64_t rows, int32_t cols, int64_t lds,
                                 float scale, void* P, int64_t ldp, int32_t dtype, void* stream) {
  VGEN_REQUIRE(dtype == VGEN_BF16 || dtype == VGEN_F16, "softmax_rows: dtype");
  VGEN_REQUIRE(rows >= 0 && cols > 0 && rows < (1LL << 31), "softmax_rows: sizes");
  if (rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VGEN_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<BF16>, dim3((unsigned)rows), dim3(256), 0, s, S, cols,
                       lds, scale, (uint16_t*)P, ldp);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<F16>, dim3((unsigned)rows), dim3(256), 0, s, S, cols,
                       lds, scale, (uint16_t*)P, ldp);
  return vgen_check_launch("softmax_rows");
}
