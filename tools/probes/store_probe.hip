// store_probe.hip — torch-free microbenchmark (r06): is the tap-GEMM epilogue's store rate (10 - 12 B/clk/CU with all 256 CUs
// storing, tools/probes/vmem_probe.hip) a limit of the CU's own write path or of the chip's?  And do a block's stores run
// under ANOTHER block's MFMAs on the same CU?
//   (1) `store_kernel` on G = 8 ... 512 blocks (one per CU up to 256): every wave stores fp32 C-fragment rows (16 B / lane,
//       16 rows x 64 B per instruction, row stride 1280 B: the epilogue's pattern) — B/clk per ACTIVE CU as G grows.
//   (2) the same with every wave issuing `mf` MFMAs between two stores (an epilogue hidden inside a K loop).
//   (3) two blocks per CU (256 threads each): block parity 0 only multiplies, parity 1 only stores.
// build: hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip ; run: ./store_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// MODE 0: stores only; 1: `mf` MFMAs between stores in the same wave; 2: odd blocks store, even blocks multiply
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(float* __restrict__ dst, int tiles, int mf, int mfma_only_iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  f32x4 acc[4] = {{1, 2, 3, 4}, {5, 6, 7, 8}, {9, 10, 11, 12}, {13, 14, 15, 16}};
  f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {2, 2, 2, 2, 2, 2, 2, 2};
  const bool storer = MODE != 2 || (blockIdx.x & 1);
  if (!storer) {
    for (int i = 0; i < mfma_only_iters; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
  } else {
    // a "tile" = 256 rows x 160 fp32 per block: wave w stores rows [32 w, 32 w + 32) as 2 x 10 fragments
    float* base = dst + (int64_t)blockIdx.x * tiles * 256 * 320;
    for (int t = 0; t < tiles; ++t) {
      float* tp = base + (int64_t)t * 256 * 320 + (int64_t)(wave * (256 / nw)) * 320;
      for (int mi = 0; mi < 256 / nw / 16; ++mi)
#pragma unroll
        for (int ni = 0; ni < 10; ++ni) {
          *(f32x4*)(tp + (int64_t)(mi * 16 + lr) * 320 + ni * 16 + lq * 4) = acc[ni & 3];
          if (MODE == 1)
            for (int j = 0; j < mf; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j & 3], 0, 0, 0);
        }
    }
  }
  if (acc[0].x == 1.2345f) sink[0] = acc[1].y + acc[2].z + acc[3].w;
}

template <int MODE>
double run(int G, int threads, float* dst, int tiles, int mf, int iters, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(G), dim3(threads), 0, 0, dst, tiles, mf, iters, sink);
  hipEventRecord(e0);
  const int it = 5;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(G), dim3(threads), 0, 0, dst, tiles, mf, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / it;
}

int main() {
  const int tiles = 24;                                  // 24 x 164 KB = 3.9 MB per block
  float *dst, *sink;
  hipMalloc(&dst, (size_t)512 * tiles * 256 * 320 * 4);
  hipMalloc(&sink, 64);
  hipMemset(dst, 0, (size_t)512 * tiles * 256 * 320 * 4);
  const double clk = 2.4e3;                              // cycles per us
  printf("(1) stores only, 8 waves per block, one block per CU up to 256 blocks: fp32 [256 x 160] tiles, %d per block\n", tiles);
  for (int G : {8, 16, 32, 64, 128, 256, 512}) {
    const double us = run<0>(G, 512, dst, tiles, 0, 0, sink);
    const double bytes = (double)G * tiles * 256 * 160 * 4;
    const int cus = G < 256 ? G : 256;
    printf("  G = %3d blocks: %8.1f us  %7.1f GB/s  %5.1f B/clk per active CU  (%5.2f us per tile)\n", G, us, bytes / us / 1e3,
           bytes / us / clk / cus, us / tiles * (G > 256 ? 0.5 : 1.0));
  }
  printf("(2) 256 blocks, every store followed by mf MFMAs (16 cycles each) in the SAME wave\n");
  for (int mf : {0, 1, 2, 4, 8, 16}) {
    const double us = run<1>(256, 512, dst, tiles, mf, 0, sink);
    const double mfma_us = (double)tiles * 20 * 2 * mf * 16 / clk;   // per wave: 20 stores per tile ... 2 waves per SIMD
    printf("  mf = %2d: %8.1f us   (the MFMAs alone: %7.1f us, stores alone: see (1) G = 256)\n", mf, us, mfma_us);
  }
  printf("(3) 512 blocks of 4 waves, two per CU: even blocks only multiply (iters x 4 MFMAs per wave), odd blocks only store\n");
  for (int iters : {0, 2000, 8000, 16000, 32000}) {
    const double us = run<2>(512, 256, dst, tiles, 0, iters, sink);
    printf("  iters = %5d: %8.1f us   (MFMA part alone on its SIMD: %7.1f us)\n", iters, us, (double)iters * 4 * 16 / clk);
  }
  return 0;
}
