// common.h — shared device helpers for the gfx950 kernels of libvgen_hip.so.
// CDNA4 only: wave64, MFMA 16x16x32 (16-bit in / fp32 accumulate), 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vgen_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- 16-bit storage types -------------------------------------------------------------
struct BF16 {
  static constexpr int kEnum = VGEN_BF16;
  static __device__ __forceinline__ float to_f32(uint16_t b) {
    return __uint_as_float(((uint32_t)b) << 16);
  }
  // round-to-nearest-even, NaN kept quiet
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  // two fp32 -> packed bf16 in ONE instruction (gfx950 v_cvt_pk_bf16_f32, round-to-nearest-even);
  // the bit-twiddling from_f32() above costs ~6 VALU ops per value and made the attention
  // softmax VALU-bound (SQ_ACTIVE_INST_VALU = 43 % of wave cycles at 2 waves / SIMD).
  // Written as a vector conversion the COMPILER lowers to that instruction — not as inline asm: the hazard
  // recogniser does not look inside asm blocks, so an asm v_cvt_pk_bf16_f32 that consumed an MFMA result straight
  // from VGPRs was issued without the required wait states and read the registers before the matrix pipe had
  // written them (r01's "temporal_kernel drifts run to run under __launch_bounds__(256, 2)": with accumulators
  // in AGPRs the v_accvgpr_read in between happened to cover the latency).
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  }
  static __device__ __forceinline__ f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a),
                                                     __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
  }
};

struct F16 {
  static constexpr int kEnum = VGEN_F16;
  static __device__ __forceinline__ float to_f32(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
  // r04: two fp32 -> packed fp16 in ONE instruction (gfx950 v_cvt_pk_f16_f32, round-to-nearest-even) through the vector
  // conversion the compiler lowers itself (see BF16::pack2 for why not inline asm).  The scalar form it replaces compiled
  // to v_cvt_f16_f32 + v_cvt_f16_f32_sdwa + v_or_b32 per pair: 48 of the ~100 VALU instructions of flash_kernel's
  // probability block, and three times the conversion work of every 16-bit tap-GEMM / norm epilogue.  Bit-identical.
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  }
  static __device__ __forceinline__ f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a),
                                                 __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return T::pack2(lo, hi);
}

template <typename T>
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  u32x2 r;
  r.x = pack2<T>(a, b);
  r.y = pack2<T>(c, d);
  return r;
}

// ---- wave64 reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// raw v_exp_f32 (2^x): -inf -> 0, no denormal fix-up sequence (4 extra VALU ops per call in exp2f)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// SiLU on the raw transcendentals: v_exp_f32 + v_rcp_f32 (1 ulp each) and 3 VALU ops.  The libm form
// x / (1 + expf(-x)) expands to ~22 VALU instructions + the same two transcendentals (range fix-ups of expf,
// IEEE division sequence): the GroupNorm apply pass was VALU-bound on it, not HBM-bound.
// x -> -inf: exp2 = +inf, rcp = 0, result -0;  x -> +inf: exp2 = 0, rcp(1) = 1, result x.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}

// erf for the exact-GELU gate of GEGLU (util.py:707-714, F.gelu default): odd minimax polynomial
// x * P(x^2), degree 8 in x^2, on |x| < 3 and exactly +-1 beyond (1 - erf(3) = 2.2e-5; the exact
// saturation matters: with a clamped polynomial gelu(g) -> 1.1e-5 * g instead of 0 for very negative
// gates, an error that grows with |g|).  Max abs error 2.2e-5 — GELU
// relative L2 error 8.5e-6 over N(0, 1.5) gates, two orders below the 16-bit rounding of the output it
// feeds.  No transcendental: the A&S 7.1.26 form used before (v_rcp + v_exp per element, quarter rate)
// made the gate 21 % of the 57344 x 2560 x 320 GEGLU GEMM (201 -> 159 us without it).
#define VGEN_ERF_C0 1.128268426e+00f
#define VGEN_ERF_C1 -3.753148778e-01f
#define VGEN_ERF_C2 1.110793392e-01f
#define VGEN_ERF_C3 -2.510286405e-02f
#define VGEN_ERF_C4 4.235428536e-03f
#define VGEN_ERF_C5 -5.110371224e-04f
#define VGEN_ERF_C6 4.106055754e-05f
#define VGEN_ERF_C7 -1.944825013e-06f
#define VGEN_ERF_C8 4.074217005e-08f

__device__ __forceinline__ float erf_poly(float x0) {
  const float x = fminf(fmaxf(x0, -3.0f), 3.0f);
  const float u = x * x;
  float p = VGEN_ERF_C8;
  p = p * u + VGEN_ERF_C7;
  p = p * u + VGEN_ERF_C6;
  p = p * u + VGEN_ERF_C5;
  p = p * u + VGEN_ERF_C4;
  p = p * u + VGEN_ERF_C3;
  p = p * u + VGEN_ERF_C2;
  p = p * u + VGEN_ERF_C1;
  p = p * u + VGEN_ERF_C0;
  return fabsf(x0) < 3.0f ? p * x : copysignf(1.0f, x0);
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float h = 0.5f * x;
  return h + h * erf_poly(x * 0.70710678118654752440f);
}
// value * gelu(gate) on 4 lanes-worth at once: written on f32x4 so the polynomial lowers to packed
// v_pk_fma_f32 / v_pk_mul_f32 (2 floats per VALU op).  Same operation order as gelu_erf_f (the split-K
// reducer's scalar path).
__device__ __forceinline__ f32x4 geglu4(f32x4 val, f32x4 g) {
  const f32x4 x0 = g * 0.70710678118654752440f;
  f32x4 x;
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = fminf(fmaxf(x0[i], -3.0f), 3.0f);
  const f32x4 u = x * x;
  f32x4 p = u * VGEN_ERF_C8 + VGEN_ERF_C7;
  p = p * u + VGEN_ERF_C6;
  p = p * u + VGEN_ERF_C5;
  p = p * u + VGEN_ERF_C4;
  p = p * u + VGEN_ERF_C3;
  p = p * u + VGEN_ERF_C2;
  p = p * u + VGEN_ERF_C1;
  p = p * u + VGEN_ERF_C0;
  f32x4 e = p * x;
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = fabsf(x0[i]) < 3.0f ? e[i] : copysignf(1.0f, x0[i]);
  const f32x4 h = g * 0.5f;
  return val * (h + h * e);
}

// exact GELU on libm's erff (condition stems: once per prompt, not performance relevant)
__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// gelu(x) on 4 values (exact-erf form, same polynomial as the GEGLU gate): the CLIP text tower's MLP activation
__device__ __forceinline__ f32x4 gelu4(f32x4 g) {
  const f32x4 one = {1.f, 1.f, 1.f, 1.f};
  return geglu4(one, g);
}

// fp32 "storage type" for the kernels that can also emit unrounded values (dtype == VGEN_F32)
struct F32Out {
  static __device__ __forceinline__ float from_f32(float f) { return f; }
};
template <typename T> struct StoreOf { typedef uint16_t type; };
template <> struct StoreOf<F32Out> { typedef float type; };

// ---- host-side error plumbing (defined in cabi.cpp) -------------------------------------
void vgen_set_error(const char* fmt, ...);
int vgen_check_launch(const char* what);

#define VGEN_REQUIRE(cond, ...)       \
  do {                                \
    if (!(cond)) {                    \
      vgen_set_error(__VA_ARGS__);    \
      return VGEN_E_BADARG;           \
    }                                 \
  } while (0)

static inline bool vgen_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Per-DEVICE launch state: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is an attribute of the function ON
// the current device, and the CU count is the device's — a process that drives several GPUs (the in-process unit
// partition) must not reuse device 0's opt-in or its 256 CUs.  `vgen_device_slot()` = the current device's index into a
// kernel's `static bool done[VGEN_MAX_DEVICES]`; `vgen_device_cus()` = its multiprocessor count (cached).
constexpr int VGEN_MAX_DEVICES = 16;
int vgen_device_slot();
int vgen_device_cus();
