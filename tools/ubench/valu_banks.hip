// Does the issue cost of a plain VALU instruction depend on WHICH registers it names (register-file banks)?
// Explicit register numbers; each kernel runs 64 x 4 instructions per loop iteration.  gfx950, wave64.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define R4(A, B, C, D) A "\n" B "\n" C "\n" D "\n"
#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
template <int K> __global__ void k(unsigned* out, int iters) {
  asm volatile("v_mov_b32 v0, 1\n v_mov_b32 v1, 2\n v_mov_b32 v2, 3\n v_mov_b32 v3, 4\n v_mov_b32 v4, 5\n v_mov_b32 v5, 6\n v_mov_b32 v6, 7\n v_mov_b32 v7, 8\n"
               "v_mov_b32 v8, 1\n v_mov_b32 v9, 2\n v_mov_b32 v10, 3\n v_mov_b32 v11, 4\n v_mov_b32 v12, 5\n v_mov_b32 v13, 6\n v_mov_b32 v14, 7\n v_mov_b32 v15, 8\n" ::: CLOB);
  for (int i = 0; i < iters; i++) {
    // sources v0..v7 never change; destinations v16..v31 (independent instructions)
    if (K == 0) asm volatile(REP16(R4("v_add_u32 v16, v0, v4", "v_add_u32 v17, v1, v5", "v_add_u32 v18, v2, v6", "v_add_u32 v19, v3, v7")) ::: CLOB);  // src0, src1 same bank (mod 4)
    if (K == 1) asm volatile(REP16(R4("v_add_u32 v16, v0, v5", "v_add_u32 v17, v1, v6", "v_add_u32 v18, v2, v7", "v_add_u32 v19, v3, v4")) ::: CLOB);  // different banks, dst = src0 bank
    if (K == 2) asm volatile(REP16(R4("v_add_u32 v18, v0, v5", "v_add_u32 v19, v1, v6", "v_add_u32 v16, v2, v7", "v_add_u32 v17, v3, v4")) ::: CLOB);  // all three different
    if (K == 3) asm volatile(REP16(R4("v_add_u32 v16, v0, v0", "v_add_u32 v17, v1, v1", "v_add_u32 v18, v2, v2", "v_add_u32 v19, v3, v3")) ::: CLOB);  // same register twice
    if (K == 4) asm volatile(REP16(R4("v_add_u32 v16, v0, v8", "v_add_u32 v17, v1, v9", "v_add_u32 v18, v2, v10", "v_add_u32 v19, v3, v11")) ::: CLOB); // same bank mod 4 and mod 8
    if (K == 5) asm volatile(REP16(R4("v_add_u32 v16, v0, v2", "v_add_u32 v17, v1, v3", "v_add_u32 v18, v4, v6", "v_add_u32 v19, v5, v7")) ::: CLOB);  // same parity
    if (K == 6) asm volatile(REP16(R4("v_add_u32 v16, v0, v1", "v_add_u32 v17, v2, v3", "v_add_u32 v18, v4, v5", "v_add_u32 v19, v6, v7")) ::: CLOB);  // neighbours
    // dependent chains (each instruction reads the previous result), 4 chains interleaved / 1 chain
    if (K == 7) asm volatile(REP16(R4("v_add_u32 v16, v16, v1", "v_add_u32 v17, v17, v2", "v_add_u32 v18, v18, v3", "v_add_u32 v19, v19, v4")) ::: CLOB);
    if (K == 8) asm volatile(REP16(R4("v_add_u32 v16, v16, v1", "v_add_u32 v16, v16, v2", "v_add_u32 v16, v16, v3", "v_add_u32 v16, v16, v5")) ::: CLOB);
    if (K == 9) asm volatile(REP16(R4("v_add_u32 v16, v16, v1", "v_add_u32 v17, v17, v2", "v_add_u32 v16, v16, v3", "v_add_u32 v17, v17, v4")) ::: CLOB);  // 2 chains
    // expensive forms, bank variants
    if (K == 10) asm volatile(REP16(R4("v_mad_i32_i24 v16, v0, v4, v8", "v_mad_i32_i24 v17, v1, v5, v9", "v_mad_i32_i24 v18, v2, v6, v10", "v_mad_i32_i24 v19, v3, v7, v11")) ::: CLOB);  // all same bank
    if (K == 11) asm volatile(REP16(R4("v_mad_i32_i24 v16, v0, v5, v10", "v_mad_i32_i24 v17, v1, v6, v11", "v_mad_i32_i24 v18, v2, v7, v8", "v_mad_i32_i24 v19, v3, v4, v9")) ::: CLOB);  // all different
    if (K == 12) asm volatile(REP16(R4("v_mad_i32_i24 v16, v0, s4, v10", "v_mad_i32_i24 v17, v1, s4, v11", "v_mad_i32_i24 v18, v2, s4, v8", "v_mad_i32_i24 v19, v3, s4, v9")) ::: CLOB, "s4");
    // dependent pair as in the butterfly: mad -> ashr -> sub, 4 chains
    if (K == 14) asm volatile(REP16(R4("v_mad_i32_i24 v16, v0, s4, v10\n v_ashrrev_i32 v16, 8, v16\n v_sub_u32 v20, v16, v5", "v_mad_i32_i24 v17, v1, s4, v11\n v_ashrrev_i32 v17, 8, v17\n v_sub_u32 v21, v17, v6",
                                       "v_mad_i32_i24 v18, v2, s4, v8\n v_ashrrev_i32 v18, 8, v18\n v_sub_u32 v22, v18, v7", "v_mad_i32_i24 v19, v3, s4, v9\n v_ashrrev_i32 v19, 8, v19\n v_sub_u32 v23, v19, v4")) ::: CLOB, "s4");
    // v_mul_i32_i24 + add 128 + ashr  vs mad
    if (K == 15) asm volatile(REP16(R4("v_add_u32_sdwa v16, sext(v0), v5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD", "v_add_u32_sdwa v17, sext(v1), v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD",
                                       "v_add_u32_sdwa v18, sext(v2), v7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa v19, sext(v3), v4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")) ::: CLOB);
    if (K == 16) asm volatile(REP16(R4("v_max_i16 v16, v0, v5", "v_max_i16 v17, v1, v6", "v_min_i16 v18, v2, v7", "v_min_i16 v19, v3, v4")) ::: CLOB);
    if (K == 17) asm volatile(REP16(R4("v_min_i16_sdwa v16, v0, v5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD", "v_min_i16_sdwa v17, v1, v6 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD",
                                       "v_min_i16_sdwa v18, v2, v7 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD", "v_min_i16_sdwa v19, v3, v4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")) ::: CLOB);
    if (K == 18) asm volatile(REP16(R4("v_max_i16 v16, 16, v5", "v_max_i16 v17, 16, v6", "v_min_i16 v18, v2, v12", "v_min_i16 v19, v3, v12")) ::: CLOB);  // inline constant operand
    if (K == 19) asm volatile(REP16(R4("v_bfe_i32 v16, v0, 3, 16", "v_med3_i32 v17, v1, 16, v6", "v_bfe_i32 v18, v2, 3, 16", "v_med3_i32 v19, v3, 16, v4")) ::: CLOB);
    if (K == 20) asm volatile(REP16(R4("v_add_u32 v16, v0, v5\n v_mov_b32 v20, v1", "v_sub_u32 v17, v1, v6\n v_mov_b32 v21, v2", "v_add_u32 v18, v2, v7\n v_mov_b32 v22, v3", "v_sub_u32 v19, v3, v4\n v_mov_b32 v23, v0")) ::: CLOB);  // add + mov pairs
    if (K == 21) asm volatile(REP16(R4("v_pk_add_i16 v16, v0, v5", "v_pk_sub_i16 v17, v1, v6", "v_pk_add_i16 v18, v2, v7", "v_pk_sub_i16 v19, v3, v4")) ::: CLOB);
    if (K == 22) asm volatile(REP16(R4("v_add_f32 v16, v0, v5", "v_sub_f32 v17, v1, v6", "v_mul_f32 v18, v2, v7", "v_add_f32 v19, v3, v4")) ::: CLOB);
    if (K == 23) asm volatile(REP16(R4("v_fma_f32 v16, v0, v5, v10", "v_fma_f32 v17, v1, v6, v11", "v_fma_f32 v18, v2, v7, v8", "v_fma_f32 v19, v3, v4, v9")) ::: CLOB);
    if (K == 24) asm volatile(REP16(R4("v_fmac_f32 v16, v0, v5", "v_fmac_f32 v17, v1, v6", "v_fmac_f32 v18, v2, v7", "v_fmac_f32 v19, v3, v4")) ::: CLOB);
  }
  unsigned r;
  asm volatile("v_add_u32 %0, v16, v17\n v_add_u32 %0, %0, v18\n v_add_u32 %0, %0, v19" : "=v"(r)::CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_waves = 4;
template <int K> void run(const char* name, int per4 = 4) {
  const int iters = 400, blocks = 256 * 4 * g_waves;
  unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<K><<<blocks, 64>>>(d, 2); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<K><<<blocks, 64>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double per_simd = (double)iters * 16 * per4 * g_waves;
  printf("%-44s %.3f ns per wave-instruction per SIMD\n", name, ms * 1e6 / per_simd);
  (void)hipFree(d);
}
int main(int argc, char** argv) {
  if (argc > 1) g_waves = atoi(argv[1]);
  printf("waves per SIMD: %d\n", g_waves);
  run<0>("add: src0,src1 same bank (mod 4)"); run<1>("add: different banks, dst=src0 bank"); run<2>("add: all three different banks");
  run<3>("add: same register twice"); run<4>("add: src regs 8 apart"); run<5>("add: src regs 2 apart"); run<6>("add: src neighbours");
  run<7>("add: 4 dependent chains"); run<9>("add: 2 dependent chains"); run<8>("add: 1 dependent chain");
  run<10>("mad: 3 sources same bank"); run<11>("mad: 3 different banks"); run<12>("mad: sgpr multiplier"); 
  run<14>("mad->ashr->sub x4 chains (3 instr)", 12); run<15>("add_sdwa sext word");
  run<16>("max_i16/min_i16 vgpr"); run<18>("max_i16 inline const / min vgpr"); run<17>("min_i16_sdwa -> byte, preserve"); run<19>("bfe_i32 + med3_i32");
  run<20>("add + mov pairs", 8); run<21>("pk_add/sub_i16"); run<22>("add/sub/mul f32"); run<23>("fma_f32"); run<24>("fmac_f32");
  return 0;
}
