// Why did "6 add : mad + ashr" run at 1.04 ns per instruction in valu_kinds2 and at 1.87 in valu_mix2?  Same pattern,
// different registers / harness.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
#define PAT(A,B,C,D,E,S) "v_add_u32 " A ", " B ", " E "\n v_sub_u32 " B ", " C ", " E "\n v_add_u32 " C ", " D ", " E "\n v_sub_u32 " D ", " A ", " E "\n v_add_u32 " A ", " B ", " E "\n v_sub_u32 " B ", " C ", " E "\n v_mad_i32_i24 " C ", " D ", " S ", " E "\n v_ashrrev_i32 " D ", 8, " A "\n"
template <int K> __global__ void k(unsigned* out, int iters, int s) {
  unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5, e = 7;
  if (K == 0) for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile("v_add_u32 %0, %1, %4\n v_sub_u32 %1, %2, %4\n v_add_u32 %2, %3, %4\n v_sub_u32 %3, %0, %4\n" "v_add_u32 %0, %1, %4\n v_sub_u32 %1, %2, %4\n v_mad_i32_i24 %2, %3, %5, %4\n v_ashrrev_i32 %3, 8, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(s));)) }
  if (K == 1) for (int i = 0; i < iters; i++) { asm volatile("s_mov_b32 s5, %0\n v_mov_b32 v5, 7" :: "s"(s) : "s5","v5"); REP8(REP8(asm volatile(PAT("v4","v3","v1","v2","v5","s5") ::: "v1","v2","v3","v4","v5","s5");)) }
  if (K == 2) for (int i = 0; i < iters; i++) { asm volatile("s_mov_b32 s5, %0\n v_mov_b32 v5, 7" :: "s"(s) : "s5","v5"); asm volatile(REP8(REP8(PAT("v4","v3","v1","v2","v5","s5"))) ::: "v1","v2","v3","v4","v5","s5"); }
  if (K == 3) for (int i = 0; i < iters; i++) { asm volatile("s_mov_b32 s4, %0\n v_mov_b32 v4, 7" :: "s"(s) : "s4","v4"); asm volatile(REP8(REP8(PAT("v16","v17","v18","v19","v4","s4"))) ::: "v16","v17","v18","v19","v4","s4"); }
  if (K == 4) for (int i = 0; i < iters; i++) { asm volatile("s_mov_b32 s4, %0\n v_mov_b32 v4, 7" :: "s"(s) : "s4","v4"); asm volatile(REP8(REP8(PAT("v16","v17","v18","v19","v4","s4"))) ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","s4"); }
  if (K == 5) for (int i = 0; i < iters; i++) { asm volatile("s_movk_i32 s4, 3\n v_mov_b32 v4, 7" ::: "s4","v4"); asm volatile(REP8(REP8(PAT("v16","v17","v18","v19","v4","s4"))) ::: "v16","v17","v18","v19","v4","s4"); }
  if (K == 6) for (int i = 0; i < iters; i++) { asm volatile("s_movk_i32 s4, 362\n v_mov_b32 v4, 7" ::: "s4","v4"); asm volatile(REP8(REP8(PAT("v16","v17","v18","v19","v4","s4"))) ::: "v16","v17","v18","v19","v4","s4"); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
static int g_waves = 4;
template <int K> void run(const char* name) {
  const int iters = 200, blocks = 256 * 4 * g_waves;
  unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<K><<<blocks, 64>>>(d, 2, 362); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<K><<<blocks, 64>>>(d, iters, 362); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double ins = (double)iters * 64 * 8 * g_waves;
  printf("%-60s %.3f ns per instruction per SIMD\n", name, ms * 1e6 / ins);
  (void)hipFree(d);
}
int main(int argc, char** argv) {
  if (argc > 1) g_waves = atoi(argv[1]);
  printf("waves per SIMD: %d\n", g_waves);
  run<0>("valu_kinds2 K=45 as it was (compiler's registers)"); run<1>("explicit v1-v5/s5, 64 asm statements");
  run<2>("explicit v1-v5/s5, one asm statement"); run<3>("explicit v16-v19/v4/s4 (multiplier 362)"); run<4>("the same + 32-register clobber list");
  run<5>("v16-v19, multiplier 3 (s_movk)"); run<6>("v16-v19, multiplier 362 (s_movk)");
  return 0;
}
