// vmem_probe.hip — torch-free microbenchmark (r05): what one CU's vector-memory path sustains for the access patterns of
// the W-panel-resident tap-GEMM (csrc/panelgemm.hip), 8 waves per CU, one block per CU.
//   pattern 0 "row-quad":  lane (lr = l & 15, lq = l >> 4) touches row lr, bytes [16 lq, 16 lq + 16) of a 64-byte piece —
//                          the MFMA operand layout read / written straight from row-major memory: 16 rows x 64 B per
//                          wave instruction (16 half cache lines)
//   pattern 1 "full-line": lane touches row (lr & 7), bytes [16 (lq + 4 (lr >> 3)), ...) of a 128-byte piece: 8 rows x 128 B
//                          per instruction (8 whole cache lines) — needs a DPP half-row swap to become an MFMA operand
//   pattern 2 "linear":    lane l touches bytes [16 l, 16 l + 16) of a 1-KiB piece (what LDS-DMA staging reads)
// modes: loads only | stores only | loads + stores (2 : 1), each over an L2-resident footprint (every wave re-walks
// 20 KiB slices of a 37 MB buffer that 15 other blocks of its XCD walk too) and over a streaming one.
// build: hipcc --offload-arch=gfx950 -O3 -o vmem_probe vmem_probe.hip ; run: ./vmem_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// byte offset of this lane's 16 bytes for instruction i of N over a [32 rows x row_bytes] tile with row stride `stride`
template <int PAT>
__device__ __forceinline__ int64_t lane_off(int lane, int i, int N, int64_t stride, int row_bytes) {
  const int lr = lane & 15, lq = lane >> 4;
  if (PAT == 2) {                                       // linear: consecutive lanes, consecutive 16-byte units of the tile
    const int upr = row_bytes / 16, idx = i * 64 + lane;
    return (int64_t)(idx / upr) * stride + (idx % upr) * 16;
  }
  const int half = i / (N / 2), pi = i % (N / 2);       // rows 0-15 | 16-31
  // pattern 3: the SAME 16 rows x 64 B piece as pattern 0, lanes renumbered so that a quad is one row's 64 bytes
  // (lane l -> row l >> 2, chunk l & 3): what a ds_bpermute of the C fragment would store
  // pattern 4: 8 rows x 128 B per instruction (lane l -> row l >> 3, chunk l & 7): the streaming shapes' BK = 64 DMA piece
  if (PAT == 4) return (int64_t)(half * 16 + (lane >> 3) + 8 * (pi & 1)) * stride + (pi >> 1) * 128 + (lane & 7) * 16;
  if (PAT == 3) return (int64_t)(half * 16 + (lane >> 2)) * stride + pi * 64 + (lane & 3) * 16;
  if (PAT == 0) return (int64_t)(half * 16 + lr) * stride + pi * 64 + lq * 16;
  return (int64_t)(half * 16 + (lr & 7) + 8 * (pi & 1)) * stride + (pi >> 1) * 128 + (lq + 4 * (lr >> 3)) * 16;
}

// every wave: slices of 32 rows; per slice NL loads (16 B / lane) from the source tile and NS stores to the destination tile
template <int PAT, int NL, int NS>
__global__ __launch_bounds__(512) void vmem_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                   int64_t src_stride, int64_t dst_stride, int slices, int share,
                                                   unsigned* sink, int stagger, int wrap) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // `share` blocks (consecutive on one XCD: block b -> XCD b % 8) read the SAME source rows, write distinct rows
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned L = (bid & 7u) * (nwg >> 3) + (bid >> 3);
  const int rdgrp = L / share;
  constexpr int SRB = NL * 1024 / 32, DRB = NS * 1024 / 32;       // tile row bytes
  u32x4 acc = {0, 0, 0, 0};
  for (int s = wave; s < slices; s += 8) {
    // stagger: the `share` blocks of a group walk the same rows in a rotated order (no two of them on the same cache
    // lines at the same time); wrap: the walk repeats over `wrap` slices (a footprint that stays in the 4 MiB L2)
    int sr = (s + stagger * (int)(L % share)) % slices;
    if (wrap) sr %= wrap;
    const unsigned char* sp = src + ((int64_t)rdgrp * slices + sr) * 32 * src_stride;
    unsigned char* dp = dst + ((int64_t)L * slices + s) * 32 * dst_stride;
    u32x4 v[NL > 0 ? NL : 1];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = *(const u32x4*)(sp + lane_off<PAT>(lane, i, NL, src_stride, SRB));
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i];
#pragma unroll
    for (int i = 0; i < NS; ++i) *(u32x4*)(dp + lane_off<PAT>(lane, i, NS, dst_stride, DRB)) = acc;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc.z;
}

// the same 8 rows x 128 B pieces moved by LDS-DMA (global_load_lds_dwordx4) into a per-wave 4-KiB LDS ring instead of
// registers: what the staging path of the streaming tap-GEMM shapes can take
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ __launch_bounds__(512) void dma_kernel(const unsigned char* __restrict__ src, int64_t src_stride, int slices,
                                                  int share, int wrap, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 4096];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned L = (bid & 7u) * (nwg >> 3) + (bid >> 3);
  const int rdgrp = L / share;
  for (int s = wave; s < slices; s += 8) {
    int sr = wrap ? s % wrap : s;
    const unsigned char* sp = src + ((int64_t)rdgrp * slices + sr) * 32 * src_stride;
#pragma unroll
    for (int i = 0; i < 20; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(sp + lane_off<4>(lane, i, 20, src_stride, 640)),
                                       (lptr_t)(lds + wave * 4096 + (i & 3) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lds[threadIdx.x] == 0x5a && lds[threadIdx.x + 512] == 0xa5) sink[1] = 1;
}

void run_dma(const char* name, unsigned char* src, int64_t sstride, int slices, int share, int wrap, unsigned* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(dma_kernel, dim3(256), dim3(512), 0, 0, src, sstride, slices, share, wrap, sink);
  hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(dma_kernel, dim3(256), dim3(512), 0, 0, src, sstride, slices, share, wrap, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it, lb = 256.0 * slices * 20 * 1024;
  printf("%-44s LDS-DMA 8x128B pieces: %8.1f us  %7.1f GB/s (%5.1f B/clk/CU @2.1GHz)\n", name, us, lb / us / 1e3, lb / us / 1e3 / 256 / 2.1);
}

template <int PAT, int NL, int NS>
void run(const char* name, unsigned char* src, unsigned char* dst, int64_t sstride, int64_t dstride, int slices, int share,
         unsigned* sink, int stagger = 0, int wrap = 0) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((vmem_kernel<PAT, NL, NS>), dim3(256), dim3(512), 0, 0, src, dst, sstride, dstride, slices, share, sink, stagger, wrap);
  hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((vmem_kernel<PAT, NL, NS>), dim3(256), dim3(512), 0, 0, src, dst, sstride, dstride, slices, share, sink, stagger, wrap);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
  const double lb = 256.0 * slices * NL * 1024, sb = 256.0 * slices * NS * 1024;
  printf("%-44s pat %d  %2d ld %2d st / slice: %8.1f us  loads %7.1f GB/s (%5.1f B/clk/CU @2.1GHz)  stores %7.1f GB/s (%5.1f)  %5.0f ns per VMEM instr per CU\n",
         name, PAT, NL, NS, us, lb / us / 1e3, lb / us / 1e3 / 256 / 2.1, sb / us / 1e3, sb / us / 1e3 / 256 / 2.1,
         us * 1e3 / (slices * (NL + NS)));
}

int main(int argc, char** argv) {
  const int slices = argc > 1 ? atoi(argv[1]) : 112;   // per block: 112 = the GEGLU launch, 14 = an o-proj launch (73 MB out)
  const bool quick = argc > 2;                         // only the r05-F block
  printf("slices per block: %d\n", slices);
  const int64_t src_bytes = 256LL * slices * 32 * 2560, dst_bytes = 256LL * slices * 32 * 2560;
  unsigned char *src, *dst;
  unsigned* sink;
  hipMalloc(&src, src_bytes);
  hipMalloc(&dst, dst_bytes);
  hipMalloc(&sink, 64);
  hipMemset(src, 1, src_bytes);
  hipMemset(dst, 0, dst_bytes);
#define ALL(NAME, NL, NS, SS, DS, SH)                                \
  run<0, NL, NS>(NAME, src, dst, SS, DS, slices, SH, sink);          \
  run<1, NL, NS>(NAME, src, dst, SS, DS, slices, SH, sink);          \
  run<2, NL, NS>(NAME, src, dst, SS, DS, slices, SH, sink);
#define ALLX(NAME, NL, NS, SS, DS, SH, STG, WR)                      \
  run<0, NL, NS>(NAME, src, dst, SS, DS, slices, SH, sink, STG, WR); \
  run<3, NL, NS>(NAME, src, dst, SS, DS, slices, SH, sink, STG, WR); \
  run<2, NL, NS>(NAME, src, dst, SS, DS, slices, SH, sink, STG, WR);
  // r05 call H: the streaming shapes' DMA piece (8 rows x 128 B) against 16 rows x 64 B, by row stride and sharing
  if (argc > 3) {
#define H4(NAME, SS, SH, WR)                                                   \
  run<4, 20, 0>(NAME, src, dst, SS, SS, slices, SH, sink, 0, WR);              \
  run<3, 20, 0>(NAME, src, dst, SS, SS, slices, SH, sink, 0, WR);
    run_dma("row stride  640 B, 16 share", src, 640, slices, 16, 0, sink);
    run_dma("row stride 2560 B, 8 share", src, 2560, slices, 8, 0, sink);
    run_dma("row stride  640 B, private 4-slice (L2)", src, 640, slices, 1, 4, sink);
    H4("row stride  640 B, 16 share", 640, 16, 0)
    H4("row stride  640 B, 4 share", 640, 4, 0)
    H4("row stride 1280 B, 16 share", 1280, 16, 0)
    H4("row stride 2560 B, 8 share", 2560, 8, 0)
    H4("row stride 2560 B, private 4-slice (L2)", 2560, 1, 4)
    H4("row stride  640 B, private 4-slice (L2)", 640, 1, 4)
    return 0;
  }
  // r05 call F: is quad-contiguity all it takes?  (pattern 3 = pattern 0's pieces with the lanes renumbered)
  ALLX("A loads, 16 share (L2)", 20, 0, 640, 640, 16, 0, 0)
  ALLX("fp32 tile loads, private 4-slice (L2)", 20, 0, 1280, 1280, 1, 0, 4)
  ALLX("fp32 stores [32x160], 18 MB footprint (MALL)", 0, 20, 1280, 1280, 1, 0, 0)
  ALLX("16-bit stores [32x160] stride 1920", 0, 10, 640, 1920, 1, 0, 0)
  ALLX("fp32 loads + fp32 stores (residual + out)", 20, 20, 1280, 1280, 1, 0, 0)
  if (quick) return 0;
  ALLX("A loads, 16 share, staggered by 7 slices", 20, 0, 640, 640, 16, 7, 0)
  ALLX("A loads, 2 share, staggered", 20, 0, 640, 640, 2, 7, 0)
  ALLX("A loads, private 4-slice footprint (L2)", 20, 0, 640, 640, 1, 0, 4)
  ALLX("A loads, private 1-slice footprint (L1?)", 20, 0, 640, 640, 1, 0, 1)
  // A operand: 32 rows x 640 B per slice = 20 loads; 16 blocks of an XCD share the rows (GEGLU), or none do
  ALL("A loads, rows shared by 16 blocks (L2)", 20, 0, 640, 640, 16)
  ALL("A loads, every block its own rows (HBM)", 20, 0, 640, 640, 1)
  // outputs: 16-bit [32 x 160] = 10 stores, fp32 [32 x 160] = 20 stores
  ALL("stores 16-bit [32x160], stride 1920 B", 0, 10, 640, 1920, 1)
  ALL("stores fp32 [32x160], stride 1280 B", 0, 20, 1280, 1280, 1)
  ALL("A loads (6 share) + 16-bit stores (qkv)", 20, 10, 640, 1920, 6)
  ALL("fp32 [32x160] loads + fp32 stores (res)", 20, 20, 1280, 1280, 1)
  return 0;
}
