// Cost of a dependent instruction right behind its producer, by producer class and distance.  gfx950, wave64.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","s4"
#define MAD(d, s) "v_mad_i32_i24 " d ", " s ", s4, v10\n"
#define ASHR(d) "v_ashrrev_i32 " d ", 8, " d "\n"
#define SUB(d, a, b) "v_sub_u32 " d ", " a ", " b "\n"
#define ADD(d, a, b) "v_add_u32 " d ", " a ", " b "\n"
template <int K> __global__ void k(unsigned* out, int iters) {
  asm volatile("s_movk_i32 s4, 362\n v_mov_b32 v0, 1\n v_mov_b32 v1, 2\n v_mov_b32 v2, 3\n v_mov_b32 v3, 4\n v_mov_b32 v4, 5\n v_mov_b32 v5, 6\n v_mov_b32 v6, 7\n v_mov_b32 v7, 8\n"
               "v_mov_b32 v8, 1\n v_mov_b32 v9, 2\n v_mov_b32 v10, 128\n v_mov_b32 v11, 4\n v_mov_b32 v12, 5\n v_mov_b32 v13, 6\n v_mov_b32 v14, 7\n v_mov_b32 v15, 8\n" ::: CLOB);
  for (int i = 0; i < iters; i++) {
    // 12 instructions per unit in every variant: 4 x (mad, ashr, sub)
    if (K == 0) asm volatile(REP16(MAD("v16","v0") ASHR("v16") SUB("v20","v16","v5") MAD("v17","v1") ASHR("v17") SUB("v21","v17","v6") MAD("v18","v2") ASHR("v18") SUB("v22","v18","v7") MAD("v19","v3") ASHR("v19") SUB("v23","v19","v4")) ::: CLOB);  // distance 1
    if (K == 1) asm volatile(REP16(MAD("v16","v0") MAD("v17","v1") ASHR("v16") ASHR("v17") SUB("v20","v16","v5") SUB("v21","v17","v6") MAD("v18","v2") MAD("v19","v3") ASHR("v18") ASHR("v19") SUB("v22","v18","v7") SUB("v23","v19","v4")) ::: CLOB);  // distance 2
    if (K == 2) asm volatile(REP16(MAD("v16","v0") MAD("v17","v1") MAD("v18","v2") MAD("v19","v3") ASHR("v16") ASHR("v17") ASHR("v18") ASHR("v19") SUB("v20","v16","v5") SUB("v21","v17","v6") SUB("v22","v18","v7") SUB("v23","v19","v4")) ::: CLOB);  // distance 4
    if (K == 3) asm volatile(REP16(MAD("v16","v0") MAD("v17","v1") MAD("v18","v2") MAD("v19","v3") ASHR("v24") ASHR("v25") ASHR("v26") ASHR("v27") SUB("v20","v8","v5") SUB("v21","v9","v6") SUB("v22","v8","v7") SUB("v23","v9","v4")) ::: CLOB);  // no dependencies at all
    if (K == 4) asm volatile(REP16(MAD("v16","v0") ADD("v24","v1","v5") ASHR("v16") SUB("v20","v16","v5") MAD("v17","v1") ADD("v25","v2","v6") ASHR("v17") SUB("v21","v17","v6") MAD("v18","v2") ADD("v26","v3","v7") ASHR("v18") SUB("v22","v18","v7")) ::: CLOB);  // one independent add between mad and ashr
    if (K == 5) asm volatile(REP16(MAD("v16","v0") ADD("v24","v1","v5") ADD("v25","v2","v6") ASHR("v16") MAD("v17","v1") ADD("v26","v3","v7") ADD("v27","v0","v4") ASHR("v17") MAD("v18","v2") ADD("v24","v1","v5") ADD("v25","v2","v6") ASHR("v18")) ::: CLOB);  // two independent adds between
    if (K == 6) asm volatile(REP16(MAD("v16","v0") ADD("v24","v1","v5") ADD("v25","v2","v6") ADD("v26","v3","v7") MAD("v17","v1") ADD("v27","v0","v4") ADD("v24","v1","v5") ADD("v25","v2","v6") MAD("v18","v2") ADD("v26","v3","v7") ADD("v27","v0","v4") ADD("v24","v1","v5")) ::: CLOB);  // mad + 3 independent adds, no dependent
    // cheap -> cheap dependent at distance 1 (ashr -> sub), no mad
    if (K == 7) asm volatile(REP16(ASHR("v16") SUB("v20","v16","v5") ADD("v24","v1","v5") ASHR("v17") SUB("v21","v17","v6") ADD("v25","v2","v6") ASHR("v18") SUB("v22","v18","v7") ADD("v26","v3","v7") ASHR("v19") SUB("v23","v19","v4") ADD("v27","v0","v4")) ::: CLOB);
    // expensive -> expensive dependent (bfe -> med3), distance 1 and 4
    if (K == 8) asm volatile(REP16("v_bfe_i32 v16, v0, 3, 16\n v_med3_i32 v16, v16, 16, v12\n v_bfe_i32 v17, v1, 3, 16\n v_med3_i32 v17, v17, 16, v12\n v_bfe_i32 v18, v2, 3, 16\n v_med3_i32 v18, v18, 16, v12\n"
                                   "v_bfe_i32 v19, v3, 3, 16\n v_med3_i32 v19, v19, 16, v12\n v_bfe_i32 v20, v4, 3, 16\n v_med3_i32 v20, v20, 16, v12\n v_bfe_i32 v21, v5, 3, 16\n v_med3_i32 v21, v21, 16, v12\n") ::: CLOB);
    if (K == 9) asm volatile(REP16("v_bfe_i32 v16, v0, 3, 16\n v_bfe_i32 v17, v1, 3, 16\n v_bfe_i32 v18, v2, 3, 16\n v_bfe_i32 v19, v3, 3, 16\n v_bfe_i32 v20, v4, 3, 16\n v_bfe_i32 v21, v5, 3, 16\n"
                                   "v_med3_i32 v16, v16, 16, v12\n v_med3_i32 v17, v17, 16, v12\n v_med3_i32 v18, v18, 16, v12\n v_med3_i32 v19, v19, 16, v12\n v_med3_i32 v20, v20, 16, v12\n v_med3_i32 v21, v21, 16, v12\n") ::: CLOB);
    // cheap -> expensive dependent (add -> mad)
    if (K == 10) asm volatile(REP16(ADD("v16","v0","v5") MAD("v20","v16") ADD("v24","v1","v5") ADD("v17","v1","v6") MAD("v21","v17") ADD("v25","v2","v6") ADD("v18","v2","v7") MAD("v22","v18") ADD("v26","v3","v7") ADD("v19","v3","v4") MAD("v23","v19") ADD("v27","v0","v4")) ::: CLOB);
    // ashr(3) -> max_i16 -> min_i16 chains (candidate descale+clamp), distance 1 / interleaved by 4
    if (K == 11) asm volatile(REP16("v_ashrrev_i32 v16, 3, v0\n v_max_i16 v16, 16, v16\n v_min_i16 v16, v16, v12\n v_ashrrev_i32 v17, 3, v1\n v_max_i16 v17, 16, v17\n v_min_i16 v17, v17, v12\n"
                                    "v_ashrrev_i32 v18, 3, v2\n v_max_i16 v18, 16, v18\n v_min_i16 v18, v18, v12\n v_ashrrev_i32 v19, 3, v3\n v_max_i16 v19, 16, v19\n v_min_i16 v19, v19, v12\n") ::: CLOB);
  }
  unsigned r;
  asm volatile("v_add_u32 %0, v16, v17\n v_add_u32 %0, %0, v18\n v_add_u32 %0, %0, v19\n v_add_u32 %0, %0, v20\n v_add_u32 %0, %0, v21\n v_add_u32 %0, %0, v22\n v_add_u32 %0, %0, v23" : "=v"(r)::CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_waves = 4;
template <int K> void run(const char* name) {
  const int iters = 300, blocks = 256 * 4 * g_waves;
  unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<K><<<blocks, 64>>>(d, 2); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<K><<<blocks, 64>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double units = (double)iters * 16 * g_waves;  // 12-instruction units per SIMD
  printf("%-52s %.2f ns per 12 instructions per SIMD\n", name, ms * 1e6 / units);
  (void)hipFree(d);
}
int main(int argc, char** argv) {
  if (argc > 1) g_waves = atoi(argv[1]);
  printf("waves per SIMD: %d\n", g_waves);
  run<3>("4 mad + 8 cheap, no dependencies"); run<0>("mad->ashr->sub, distance 1"); run<1>("mad->ashr->sub, distance 2"); run<2>("mad->ashr->sub, distance 4");
  run<4>("3 x (mad, add, ashr(dep), sub(dep))"); run<5>("3 x (mad, add, add, ashr(dep))"); run<6>("3 x (mad, 3 independent adds)");
  run<7>("4 x (ashr, sub(dep), add): 12 cheap"); run<8>("6 x (bfe, med3(dep))"); run<9>("6 bfe, 6 med3 (dep at distance 6)");
  run<10>("4 x (add, mad(dep), add)"); run<11>("4 x (ashr3, max_i16(dep), min_i16(dep))");
  return 0;
}
