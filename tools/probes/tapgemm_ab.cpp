// Stand-alone A/B of two builds of libvgen_hip.so through the C ABI (no torch): the same tap-GEMM launches on both, first
// mismatches printed.   g++ tools/probes/tapgemm_ab.cpp -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ \
//                         -L /opt/rocm/lib -lamdhip64 -ldl -o tools/probes/tapgemm_ab ; ./tapgemm_ab libA.so libB.so
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "vgen_hip.h"
typedef int (*tapgemm_fn)(const vgen_tapgemm_args*, void*);
typedef const char* (*err_fn)(void);

static uint16_t f2h(float f) {   // fp32 -> fp16 (round to nearest even, no denormal care needed here)
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t s = (x >> 16) & 0x8000; int e = (int)((x >> 23) & 0xff) - 127 + 15; uint32_t m = x & 0x7fffff;
  if (e <= 0) return (uint16_t)s;
  if (e >= 31) return (uint16_t)(s | 0x7c00);
  uint32_t h = s | (e << 10) | (m >> 13);
  if ((m & 0x1fff) > 0x1000 || ((m & 0x1fff) == 0x1000 && (h & 1))) h++;
  return (uint16_t)h;
}
static uint32_t rng = 12345;
static float rnd() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f; }
template <class T> static T* dev(const std::vector<T>& h) {
  T* d; hipMalloc((void**)&d, h.size() * sizeof(T) + 256); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d;
}

struct Case { const char* name; int mode; int64_t M; int N, C, Hi, Wi, Ho, Wo, stride, pad, F; int64_t S; };

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  setvbuf(stdout, NULL, _IONBF, 0);
  void* h[2]; tapgemm_fn fn[2]; err_fn ef[2];
  for (int i = 0; i < 2; ++i) {
    h[i] = dlopen(argv[1 + i], RTLD_NOW | RTLD_LOCAL);
    if (!h[i]) { printf("dlopen %s: %s\n", argv[1 + i], dlerror()); return 1; }
    fn[i] = (tapgemm_fn)dlsym(h[i], "vgen_tapgemm"); ef[i] = (err_fn)dlsym(h[i], "vgen_last_error");
  }
  Case cases[] = {
      {"linear 512x128x64", 0, 512, 128, 64, 0, 0, 0, 0, 0, 0, 0, 0},
      {"linear 512x128x256", 0, 512, 128, 256, 0, 0, 0, 0, 0, 0, 0, 0},
      {"linear 3000x160x128 (ragged)", 0, 3000, 160, 128, 0, 0, 0, 0, 0, 0, 0, 0},
      {"temporal 2x8 frames x 100 C64 -> 64", 2, 2 * 8 * 100, 64, 64, 0, 0, 0, 0, 0, 0, 8, 100},
      {"conv 2x16x28 C64 -> 64", 1, 2 * 16 * 28, 64, 64, 16, 28, 16, 28, 1, 1, 0, 0},

  };
  for (const Case& c : cases) {
    const int taps = c.mode == 0 ? 1 : (c.mode == 1 ? 9 : 3);
    const int64_t K = (int64_t)taps * c.C;
    const int64_t rowsA = c.mode == 1 ? (c.M / (c.Ho * c.Wo)) * c.Hi * c.Wi : c.M;
    std::vector<uint16_t> A(rowsA * c.C), W((int64_t)c.N * K);
    std::vector<float> bias(c.N);
    for (auto& v : A) v = f2h(rnd());
    for (auto& v : W) v = f2h(rnd() * 0.25f);
    for (auto& v : bias) v = rnd();
    uint16_t* dA = dev(A); uint16_t* dW = dev(W); float* dB = dev(bias);
    std::vector<float> out[2];
    printf("case %s\n", c.name);
    for (int i = 0; i < 2; ++i) {
      float* dO; hipMalloc((void**)&dO, c.M * c.N * 4); hipMemset(dO, 0xff, c.M * c.N * 4);
      vgen_tapgemm_args a; memset(&a, 0, sizeof(a));
      a.M = c.M; a.N = c.N; a.dtype = VGEN_F16; a.A = dA; a.lda = c.C; a.C1 = c.C; a.taps = taps; a.mode = c.mode;
      a.Hi = c.Hi; a.Wi = c.Wi; a.Ho = c.Ho; a.Wo = c.Wo; a.stride = c.stride; a.pad_t = c.pad; a.pad_l = c.pad;
      a.F = c.F; a.S = c.S; a.W = dW; a.bias = dB; a.out = dO; a.ldo = c.N; a.out_dtype = VGEN_F32;
      const int rc = fn[i](&a, nullptr);
      const hipError_t e = hipDeviceSynchronize();
      if (rc != 0 || e != hipSuccess) printf("  lib %d: rc %d (%s) sync %s\n", i, rc, ef[i] ? ef[i]() : "?", hipGetErrorString(e));
      out[i].resize(c.M * c.N);
      hipMemcpy(out[i].data(), dO, c.M * c.N * 4, hipMemcpyDeviceToHost);
      hipFree(dO);
    }
    typedef int (*dump_fn)(void*);
    dump_fn dump = (dump_fn)dlsym(h[1], "vgen_debug_dump");
    if (dump) {
      unsigned long long* dd; hipMalloc((void**)&dd, 128); dump(dd); hipDeviceSynchronize();
      unsigned long long hd[16]; hipMemcpy(hd, dd, 128, hipMemcpyDeviceToHost); hipFree(dd);
      printf("  self-check: mismatching issues %llu; first: piece %llu tile %llu tid %llu block %llu exp %llx got %llx vo %llx so %llx base %llx mode %llu operand %llx\n",
             hd[15], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], hd[8], hd[9], hd[10], hd[11]);
    }
    int64_t bad = 0, nonfinite = 0; double maxd = 0; int shown = 0;
    int64_t first_m = -1, last_m = -1;
    for (int64_t i = 0; i < c.M * c.N; ++i) {
      if (!isfinite(out[1][i])) nonfinite++;
      const double d = fabs((double)out[0][i] - (double)out[1][i]);
      if (d > 1e-6 || !(d == d)) {
        if (first_m < 0) first_m = i / c.N;
        last_m = i / c.N;
        if (shown < 6) { printf("    m %lld n %lld: %g vs %g\n", (long long)(i / c.N), (long long)(i % c.N), out[0][i], out[1][i]); shown++; }
        bad++;
      }
      if (d == d && d > maxd) maxd = d;
    }
    printf("%-40s mismatches %lld / %lld (rows %lld..%lld) max|d| %.3g nonfinite(B) %lld\n", c.name, (long long)bad,
           (long long)(c.M * c.N), (long long)first_m, (long long)last_m, maxd, (long long)nonfinite);
    hipFree(dA); hipFree(dW); hipFree(dB);
  }
  return 0;
}
