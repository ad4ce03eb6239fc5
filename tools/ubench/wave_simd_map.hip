// Which SIMD does wavefront i of a workgroup land on?  (HW_REG_HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13.)
// Launches <blocks> workgroups of <waves> wavefronts with <lds> bytes of LDS each (the PNG inflate kernel's shape:
// 880 x 2 x 40 KB) that stay resident for a while, and prints the histogram wave index -> SIMD id.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/wave_simd_map.hip -o tools/ubench/wave_simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  extern __shared__ char lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
  // stay resident so that later workgroups are placed beside earlier ones, as in a long-running decode
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 9999) lds[0] = 1;
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 880, waves = argc > 2 ? atoi(argv[2]) : 2, lds = argc > 3 ? atoi(argv[3]) : 40924;
  unsigned* d;
  std::vector<unsigned> h((size_t)blocks * waves);
  if (hipMalloc(&d, h.size() * 4) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 2;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), lds, 0, d, 200000 /* 2 ms */);
  if (hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  long hist[8][4] = {};
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < waves; ++w) hist[w][(h[(size_t)b * waves + w] >> 4) & 3]++;
  printf("%d workgroups x %d wavefronts, %d B LDS\n", blocks, waves, lds);
  for (int w = 0; w < waves; ++w)
    printf("  wavefront %d on SIMD 0/1/2/3: %ld %ld %ld %ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  // per CU (se, sh, cu): how many wave-0s share the busiest SIMD
  long worst = 0, cus = 0, sum = 0;
  std::vector<int> cnt(1 << 16, 0);
  for (int b = 0; b < blocks; ++b) {
    const unsigned id = h[(size_t)b * waves];
    cnt[((id >> 8) & 0xff) << 2 | ((id >> 4) & 3)]++;
  }
  for (int c = 0; c < (1 << 14); ++c) {
    int m = 0, t = 0;
    for (int s = 0; s < 4; ++s) { m = cnt[c << 2 | s] > m ? cnt[c << 2 | s] : m; t += cnt[c << 2 | s]; }
    if (t) { ++cus; sum += m; worst = m > worst ? m : worst; }
  }
  printf("  (se,sh,cu) ids seen: %ld (XCDs share ids); wavefront 0s on the busiest SIMD of an id: mean %.2f max %ld\n", cus, (double)sum / cus, worst);
  return 0;
}
