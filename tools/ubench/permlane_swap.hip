// What v_permlane16_swap_b32 does on gfx950 (which 16-lane rows trade places): prints lane -> (a', b') for a = lane, b = 100 + lane.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/permlane_swap.hip -o tools/ubench/permlane_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[threadIdx.x] = r[0];
  p[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d;
  unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int row = 0; row < 4; ++row)
    printf("row %d (lanes %2d..%2d): a' = %3u..%3u   b' = %3u..%3u\n", row, 16 * row, 16 * row + 15, h[16 * row], h[16 * row + 15],
           h[64 + 16 * row], h[64 + 16 * row + 15]);
  return 0;
}
