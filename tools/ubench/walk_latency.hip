// Micro-benchmark behind csrc/png.hip's serial walk: what one wave pays for dependent scalar chains on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/walk_latency.hip -o /tmp/walk_latency && /tmp/walk_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void k(uint64_t* out, uint32_t seed) {
  const int lane = threadIdx.x;
  uint32_t cur = 7 + ((lane * 2654435761u + seed) >> 29);  // a fake entry: 7..14 bits per symbol, bit 7 clear
  uint64_t t0, t1;
  uint32_t s = seed & 1, p = 0, e = 0, t = 0;
  uint64_t m = 0;
  int n = 0;
  // 0: dependent s_add chain
  t0 = __builtin_readcyclecounter();
  asm volatile(REP64("s_add_u32 %0, %0, 1\n\t") : "+s"(s)::"scc");
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 1: v_readlane -> s_and -> v_readlane (lane select from SALU result)
  p = seed & 3;
  t0 = __builtin_readcyclecounter();
  asm volatile(REP64("v_readlane_b32 %1, %2, %0\n\ts_and_b32 %0, %1, 63\n\t") : "+s"(p), "=&s"(e) : "v"(cur) : "scc");
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 2: v_readlane -> s_and -> s_add -> s_and63 -> v_readlane (4-instruction chain)
  t0 = __builtin_readcyclecounter();
  asm volatile(REP64("v_readlane_b32 %1, %2, %0\n\ts_and_b32 %3, %1, 63\n\ts_add_u32 %0, %0, %3\n\ts_and_b32 %0, %0, 63\n\t")
               : "+s"(p), "=&s"(e), "+v"(cur), "=&s"(t)::"scc");
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 3: as 2 plus a never-taken conditional branch per step
  t0 = __builtin_readcyclecounter();
  asm volatile(REP64("v_readlane_b32 %1, %2, %0\n\ts_bitcmp1_b32 %1, 7\n\ts_cbranch_scc1 .Lx3_%=\n\ts_and_b32 %3, %1, 63\n\ts_add_u32 "
                     "%0, %0, %3\n\ts_and_b32 %0, %0, 63\n\t") ".Lx3_%=:\n\t"
               : "+s"(p), "=&s"(e), "+v"(cur), "=&s"(t)::"scc");
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 4: the walk step of png.hip (8 instructions, 2 never-taken branches), p wrapped instead of leaving
  t0 = __builtin_readcyclecounter();
  asm volatile(REP64("v_readlane_b32 %1, %2, %0\n\ts_bitcmp1_b32 %1, 7\n\ts_cbranch_scc1 .Lx4_%=\n\ts_bitset1_b64 %4, %0\n\ts_and_b32 "
                     "%3, %1, 63\n\ts_add_u32 %0, %0, %3\n\ts_and_b32 %0, %0, 63\n\ts_cmp_gt_u32 %0, 63\n\ts_cbranch_scc1 .Lx4_%=\n\t")
               ".Lx4_%=:\n\t"
               : "+s"(p), "=&s"(e), "+v"(cur), "=&s"(t), "+s"(m)::"scc");
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 5: taken branches: 64 x (s_branch to the next instruction group)
  t0 = __builtin_readcyclecounter();
  asm volatile(REP64("s_branch 1f\n\ts_nop 0\n1:\n\t") ::: "scc");
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 6: independent s_nop 0 x 64 (issue rate)
  t0 = __builtin_readcyclecounter();
  asm volatile(REP64("s_nop 0\n\t"));
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 7: v_readlane -> v_readlane through VALU only: v_readlane, v_mov from sgpr, (lane select must be sgpr) -- skip;
  //    instead: ds_bpermute chain (all-VALU/LDS formulation)
  uint32_t vp = lane & 3;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 64; ++i) vp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(vp << 2), (int)cur) & 63;
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[n] = t1 - t0;
  ++n;
  // 8: s_movrels chain: table in SGPRs (pure SALU)
  {
    uint32_t idx = seed & 7, v;
    t0 = __builtin_readcyclecounter();
    asm volatile(
        "s_mov_b32 s40, 1\n\ts_mov_b32 s41, 2\n\ts_mov_b32 s42, 3\n\ts_mov_b32 s43, 1\n\ts_mov_b32 s44, 2\n\ts_mov_b32 s45, "
        "3\n\ts_mov_b32 s46, 1\n\ts_mov_b32 s47, 2\n\t" REP64(
            "s_mov_b32 m0, %0\n\ts_movrels_b32 %1, s40\n\ts_add_u32 %0, %0, %1\n\ts_and_b32 %0, %0, 7\n\t")
        : "+s"(idx), "=&s"(v)::"scc", "m0", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[n] = t1 - t0 + (idx & 0);
    ++n;
  }
  // 9: LDS table lookup chain: ds_read_b32 (uniform address) -> readfirstlane -> address
  {
    __shared__ uint32_t tab[64];
    tab[lane] = (lane * 7 + 3) & 63;
    __syncthreads();
    uint32_t a = seed & 63;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 64; ++i) a = __builtin_amdgcn_readfirstlane(tab[a]);
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[n] = t1 - t0 + (a & 0);
    ++n;
  }
  if (lane == 0) out[15] = p + e + t + s + (uint32_t)m + vp;
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 16 * 8);
  uint64_t h[16];
  const char* names[] = {"s_add chain", "readlane->s_and->readlane", "readlane+3 salu chain", "  + never-taken branch",
                         "png walk step (8 instr)", "taken s_branch", "s_nop issue", "ds_bpermute chain", "s_movrels chain (4 salu)",
                         "lds read + readfirstlane chain"};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 12345u + rep);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  }
  for (int i = 0; i < 10; ++i) printf("%-34s %6.1f cycles per step (64 steps: %llu)\n", names[i], h[i] / 64.0, (unsigned long long)h[i]);
  return 0;
}
