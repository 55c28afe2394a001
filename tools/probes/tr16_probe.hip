// Semantics probe for gfx950's LDS transpose read (ds_read_b64_tr_b16) — torch-free:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o tools/probes/tr16_probe && tools/probes/tr16_probe
// LDS element e holds the value e.  Every lane hands the instruction an 8-byte-aligned address of 4 consecutive 16-bit
// elements; the probe prints, per lane, the 4 element indices it received -> (lane, j) -> (source lane, source element).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16* lds_v4;

__global__ void probe(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int e;                                     // first element of this lane's 4
  if (mode == 0) e = lane * 4;                                     // linear: lane l -> elements 4l .. 4l+3
  else if (mode == 1) e = (lane & 15) * 64 + (lane >> 4) * 4;      // rows of 64 elements: row = lane % 16, col group = lane / 16
  else if (mode == 2) e = (lane >> 2) * 64 + (lane & 3) * 4;       // 16 rows x 16 cols blocks: row = lane / 4, 4-col group = lane % 4
  else e = ((lane & 15) >> 2) * 64 + (lane & 3) * 4 + (lane >> 4) * 16;   // per 16-lane group a [4 rows][16 cols] block, groups side by side
  const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + e));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}

int main() {
  unsigned short* d;
  unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: lane: received element indices  [as (source lane, element) for mode 0]\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        if (mode == 0) printf(" %4d(l%2d,e%d)", h[l * 4 + j], h[l * 4 + j] / 4, h[l * 4 + j] % 4);
        else printf(" %4d(r%2d,c%2d)", h[l * 4 + j], h[l * 4 + j] / 64, h[l * 4 + j] % 64);
      }
      printf("\n");
    }
  }
  return 0;
}
