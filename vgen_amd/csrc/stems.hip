// stems.hip — the condition stems ahead of the trunk: small-channel convolutions, adaptive average
// pooling and the per-pixel frame transformer of UNetSD_I2VGen (tools/modules/unet/unet_i2vgen.py:116-132, 280-321)
// and of the composer UNets (unet_videolcm.py:294-372, 598-699; unet_tf2tv.py likewise).
//
// They depend on the conditioning images only, so a sampling session evaluates them ONCE per prompt; channel counts
// are 1..64 (one 64 -> 1024 projection at 8x8) — nothing here is MFMA-shaped or bandwidth-critical.  Plain fp32
// kernels, fixed summation order, NCHW frames in, the trunk's [B, C, F, H, W] stem-channel layout out.
#include "common.h"

namespace {

constexpr int CO_T = 8;   // output channels per thread of the direct convolution

// y[n, co, oy, ox] = act(b[co] + sum_{ci, ky, kx} w[co, ci, ky, kx] * x[n, ci, oy*stride + ky - 1, ox*stride + kx - 1])
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const float* __restrict__ x, int Cin, int H, int W,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            int Cout, int stride, int Ho, int Wo, int act,
                                                            float* __restrict__ y, int64_t npix) {
#pragma clang fp contract(off)
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (n, oy, ox)
  if (pix >= npix) return;
  const int co0 = blockIdx.y * CO_T;
  const int ox = (int)(pix % Wo);
  const int oy = (int)((pix / Wo) % Ho);
  const int64_t n = pix / ((int64_t)Wo * Ho);
  float acc[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) acc[j] = (b && co0 + j < Cout) ? b[co0 + j] : 0.f;
  const float* xn = x + n * Cin * (int64_t)H * W;
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xc = xn + (int64_t)ci * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * stride + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * stride + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const float v = xc[(int64_t)iy * W + ix];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) {
          if (co0 + j < Cout) acc[j] = __builtin_fmaf(w[((int64_t)(co0 + j) * Cin + ci) * 9 + ky * 3 + kx], v, acc[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CO_T; ++j) {
    if (co0 + j < Cout) {
      float o = acc[j];
      if (act == 1) o = o / (1.0f + expf(-o));
      y[((n * Cout + co0 + j) * Ho + oy) * (int64_t)Wo + ox] = o;
    }
  }
}

// nn.AdaptiveAvgPool2d: window of output o = [floor(o*H/Ho), ceil((o+1)*H/Ho))
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ x, int H, int W, int Ho, int Wo,
                                                               float* __restrict__ y, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (plane, oy, ox)
  if (i >= total) return;
  const int ox = (int)(i % Wo);
  const int oy = (int)((i / Wo) % Ho);
  const int64_t pl = i / ((int64_t)Wo * Ho);
  const int y0 = (int)(((int64_t)oy * H) / Ho), y1 = (int)((((int64_t)oy + 1) * H + Ho - 1) / Ho);
  const int x0 = (int)(((int64_t)ox * W) / Wo), x1 = (int)((((int64_t)ox + 1) * W + Wo - 1) / Wo);
  const float* p = x + pl * (int64_t)H * W;
  float s = 0.f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) s += p[(int64_t)yy * W + xx];
  y[i] = s / (float)((y1 - y0) * (x1 - x0));
}

// One layer of the reference's TransformerV2 over the FRAME axis of every pixel (util.py:1396-1453):
//   x = to_out(softmax(q k^T / sqrt(dh)) v) + x   with q, k, v = to_qkv(LayerNorm(x)) split into `heads` heads
//   x = W2 gelu(W1 x + b1) + b2 + x
// Tokens: x[b, f, c, pix] (NCHW frames, d = C channels <= 16, F <= 32 frames).  One thread per (sequence, frame); a
// block holds whole sequences and shares their K / V through LDS.  The result goes either back to NCHW frames
// (another layer follows) or, `last`, to the trunk's stem-channel layout out[b, c, f, pix] scaled by `out_scale`
// and optionally accumulated (the composer sums its stems; the reference's I2VGen adds its map twice).
constexpr int FT_MAXD = 16, FT_MAXI = 32, FT_MAXH = 64;

struct FrameTfParams {
  const float *ln_w, *ln_b, *wqkv, *wout, *bout, *w1, *b1, *w2, *b2;
  int d, heads, dh, hidden;
};

__global__ __launch_bounds__(256) void frame_transformer_kernel(const float* __restrict__ x, int64_t nseq, int F, int HW,
                                                                const FrameTfParams P, float* __restrict__ y,
                                                                int last, float out_scale, int accumulate) {
  extern __shared__ float ft_kv[];   // [seq in block][F][2 * inner]
  const int spb = 256 / F;           // sequences per block
  const int ls = threadIdx.x / F, f = threadIdx.x - ls * F;
  const int64_t s = (int64_t)blockIdx.x * spb + ls;
  const bool live = ls < spb && s < nseq;
  const int d = P.d, inner = P.heads * P.dh;
  const int64_t b = live ? s / HW : 0;
  const int pix = live ? (int)(s - b * HW) : 0;
  float xv[FT_MAXD], nv[FT_MAXD];
  float q[FT_MAXI];
  if (live) {
    float mean = 0.f;
    for (int c = 0; c < d; ++c) {
      xv[c] = x[((b * F + f) * d + c) * (int64_t)HW + pix];
      mean += xv[c];
    }
    mean /= (float)d;
    float var = 0.f;
    for (int c = 0; c < d; ++c) var += (xv[c] - mean) * (xv[c] - mean);
    const float rstd = 1.0f / sqrtf(var / (float)d + 1e-5f);
    for (int c = 0; c < d; ++c) nv[c] = (xv[c] - mean) * rstd * P.ln_w[c] + P.ln_b[c];
    float* kv = ft_kv + ((int64_t)ls * F + f) * 2 * inner;
    for (int j = 0; j < 3 * inner; ++j) {
      float a = 0.f;
      for (int c = 0; c < d; ++c) a += P.wqkv[j * d + c] * nv[c];
      if (j < inner) q[j] = a;
      else kv[j - inner] = a;                       // k | v
    }
  }
  __syncthreads();
  if (!live) return;
  float o[FT_MAXI];
  const float scale = 1.0f / sqrtf((float)P.dh);
  const float* kvs = ft_kv + (int64_t)ls * F * 2 * inner;
  for (int h = 0; h < P.heads; ++h) {
    float mx = -INFINITY;
    float sc[32];
    for (int g = 0; g < F; ++g) {
      float a = 0.f;
      for (int e = 0; e < P.dh; ++e) a += q[h * P.dh + e] * kvs[(int64_t)g * 2 * inner + h * P.dh + e];
      sc[g] = a * scale;
      mx = fmaxf(mx, sc[g]);
    }
    float sum = 0.f;
    for (int g = 0; g < F; ++g) {
      sc[g] = expf(sc[g] - mx);
      sum += sc[g];
    }
    for (int e = 0; e < P.dh; ++e) {
      float a = 0.f;
      for (int g = 0; g < F; ++g) a += sc[g] * kvs[(int64_t)g * 2 * inner + inner + h * P.dh + e];
      o[h * P.dh + e] = a / sum;
    }
  }
  for (int c = 0; c < d; ++c) {                     // to_out (+ bias) + residual
    float a = P.bout ? P.bout[c] : 0.f;
    if (P.wout) {
      for (int j = 0; j < inner; ++j) a += P.wout[c * inner + j] * o[j];
    } else {
      a = o[c];                                     // heads == 1 && dh == d: identity projection
    }
    xv[c] += a;
  }
  float hv[FT_MAXH];
  for (int j = 0; j < P.hidden; ++j) {
    float a = P.b1[j];
    for (int c = 0; c < d; ++c) a += P.w1[j * d + c] * xv[c];
    hv[j] = gelu_erf_exact(a);
  }
  for (int c = 0; c < d; ++c) {
    float a = P.b2[c];
    for (int j = 0; j < P.hidden; ++j) a += P.w2[c * P.hidden + j] * hv[j];
    const float r = a + xv[c];
    if (last) {
      float* dst = y + ((b * d + c) * F + f) * (int64_t)HW + pix;
      const float v = r * out_scale;
      *dst = accumulate ? *dst + v : v;
    } else {
      y[((b * F + f) * d + c) * (int64_t)HW + pix] = r;
    }
  }
}

}  // namespace

extern "C" int vgen_conv3x3_small(const float* x, int64_t n, int32_t Cin, int32_t H, int32_t W, const float* w,
                                  const float* b, int32_t Cout, int32_t stride, int32_t act, float* y, void* stream) {
  VGEN_REQUIRE(x && w && y && n > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) &&
                   (act == 0 || act == 1),
               "conv3x3_small: arguments");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const int64_t npix = n * Ho * Wo;
  const int64_t gx = (npix + 255) / 256;
  VGEN_REQUIRE(gx < (1LL << 31) && (Cout + CO_T - 1) / CO_T <= 65535, "conv3x3_small: too large");
  hipLaunchKernelGGL(conv3x3_small_kernel, dim3((unsigned)gx, (unsigned)((Cout + CO_T - 1) / CO_T)), dim3(256), 0,
                     (hipStream_t)stream, x, Cin, H, W, w, b, Cout, stride, Ho, Wo, act, y, npix);
  return vgen_check_launch("conv3x3_small");
}

extern "C" int vgen_adaptive_avgpool2d(const float* x, int64_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                       float* y, void* stream) {
  VGEN_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "adaptive_avgpool2d: arguments");
  const int64_t total = planes * Ho * Wo;
  const int64_t grid = (total + 255) / 256;
  VGEN_REQUIRE(grid < (1LL << 31), "adaptive_avgpool2d: too large");
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, H, W, Ho, Wo, y,
                     total);
  return vgen_check_launch("adaptive_avgpool2d");
}

extern "C" int vgen_frame_transformer(const float* x, int64_t B, int32_t F, int32_t d, int64_t HW, int32_t heads,
                                      int32_t dim_head, int32_t hidden, const float* ln_w, const float* ln_b,
                                      const float* wqkv, const float* wout, const float* bout, const float* w1,
                                      const float* b1, const float* w2, const float* b2, float* y, int32_t last,
                                      float out_scale, int32_t accumulate, void* stream) {
  VGEN_REQUIRE(x && y && ln_w && ln_b && wqkv && w1 && b1 && w2 && b2, "frame_transformer: null parameter");
  VGEN_REQUIRE(B > 0 && HW > 0 && F > 0 && F <= 32 && d > 0 && d <= FT_MAXD && heads > 0 && dim_head > 0 &&
                   heads * dim_head <= FT_MAXI && hidden > 0 && hidden <= FT_MAXH && HW < (1LL << 31),
               "frame_transformer: F=%d d=%d heads=%d dim_head=%d hidden=%d out of range", F, d, heads, dim_head, hidden);
  VGEN_REQUIRE(wout != nullptr || (heads == 1 && dim_head == d), "frame_transformer: missing to_out");
  const int64_t nseq = B * HW;
  const int spb = 256 / F;
  const int64_t grid = (nseq + spb - 1) / spb;
  VGEN_REQUIRE(grid < (1LL << 31), "frame_transformer: too large");
  const FrameTfParams P{ln_w, ln_b, wqkv, wout, bout, w1, b1, w2, b2, d, heads, dim_head, hidden};
  const size_t lds = (size_t)spb * F * 2 * heads * dim_head * sizeof(float);
  hipLaunchKernelGGL(frame_transformer_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, x, nseq, F,
                     (int)HW, P, y, last, out_scale, accumulate);
  return vgen_check_launch("frame_transformer");
}
