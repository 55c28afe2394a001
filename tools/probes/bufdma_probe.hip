// Stand-alone probe (no torch): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 — does an out-of-range lane
// write zeros to LDS, is the scalar offset part of the range check, is the per-lane LDS placement lane*16.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/bufdma_probe.hip -o tools/probes/bufdma_probe && ./bufdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void fill(uint32_t* p, size_t nchunks) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < nchunks; i += (size_t)gridDim.x * blockDim.x) {
    p[i * 4 + 0] = (uint32_t)i; p[i * 4 + 1] = 0x11111111u; p[i * 4 + 2] = 0x22222222u; p[i * 4 + 3] = 0x33333333u;
  }
}

// test t: LDS pre-filled with 0xABABABAB, one DMA, dump the first dword of every 16-byte LDS slot
__global__ void probe(const char* base, uint32_t* out, int t) {
  __shared__ __attribute__((aligned(16))) uint32_t smem[64 * 4];
  const int lane = threadIdx.x;
  for (int k = 0; k < 4; ++k) smem[lane * 4 + k] = 0xABABABABu;
  __syncthreads();
  unsigned num = 0x7fffffffu; unsigned vo = lane * 16; int so = 0;
  if (t == 1) vo = (lane & 1) ? 0x80000000u : lane * 16;          // OOB by a huge offset
  if (t == 2) num = 512;                                           // lanes >= 32 out of range
  if (t == 3) { num = 512; so = 256; }                             // is soffset part of the check? (then lanes >= 16 OOB)
  if (t == 4) { vo = lane * 16 + 4096; so = 128; }                 // plain soffset addition
  if (t == 5) vo = (lane & 1) ? 0xfffffff0u : lane * 16;          // OOB just below 2^32
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)num, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)smem, 16, (int)vo, so, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[t * 64 + lane] = smem[lane * 4];
}

int main() {
  const size_t bytes = 3ull << 30;   // 3 GiB: base + 0x80000000 stays inside the allocation (no fault if not range-checked)
  char* buf; uint32_t* out;
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 6 * 64 * 4);
  fill<<<4096, 256>>>((uint32_t*)buf, bytes / 16);
  for (int t = 0; t < 6; ++t) probe<<<1, 64>>>(buf, out, t);
  uint32_t h[6 * 64];
  hipError_t e = hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("memcpy: %s\n", hipGetErrorString(e));
  const char* names[6] = {"t0 in range (expect chunk index = lane)", "t1 odd lanes voffset 0x80000000 (0 = zero-filled, ab.. = skipped, 0x08000000+ = read)",
                          "t2 num_records 512 (lanes >= 32 OOB)", "t3 num_records 512, soffset 256", "t4 voffset lane*16+4096, soffset 128 (expect 264 + lane)",
                          "t5 odd lanes voffset 0xfffffff0"};
  for (int t = 0; t < 6; ++t) {
    printf("%s\n ", names[t]);
    for (int l = 0; l < 64; ++l) printf("%x ", h[t * 64 + l]);
    printf("\n");
  }
  return 0;
}
