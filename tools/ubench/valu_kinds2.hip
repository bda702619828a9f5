// Issue cost of further VALU instruction forms (round 2), wave64 on gfx950; waves per SIMD = argv[1] (default 4).
// Same harness as valu_kinds.hip: 4 rotating registers so that neighbours are independent.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP8(x) x x x x x x x x
#define BODY(INS) \
  for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile(INS : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(s) : "vcc");)) }
#define Q4(OP, TAIL) OP " %0, %1" TAIL "\n " OP " %1, %2" TAIL "\n " OP " %2, %3" TAIL "\n " OP " %3, %0" TAIL
template <int K> __global__ void k(unsigned* out, int iters, int s) {
  unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5, e = 7;
  if (K == 0) BODY(Q4("v_add_u32", ", %4"))
  if (K == 1) BODY(Q4("v_sub_u32", ", %4"))
  if (K == 2) BODY(Q4("v_xor_b32", ", %4"))
  if (K == 3) BODY("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4")
  if (K == 4) BODY(Q4("v_fma_f32", ", %4, %4"))
  if (K == 5) BODY(Q4("v_add_f32", ", %4"))
  if (K == 6) BODY("v_sat_pk_u8_i16 %0, %1\n v_sat_pk_u8_i16 %1, %2\n v_sat_pk_u8_i16 %2, %3\n v_sat_pk_u8_i16 %3, %0")
  if (K == 7) BODY(Q4("v_dot2_i32_i16", ", %4, %4"))
  if (K == 8) BODY(Q4("v_pk_max_i16", ", %4"))
  if (K == 9) BODY(Q4("v_pk_mad_i16", ", %4, %4"))
  if (K == 10) BODY(Q4("v_mad_i32_i16", ", %4, %4"))
  if (K == 11) BODY(Q4("v_pk_add_i16", ", %4"))
  if (K == 12) BODY(Q4("v_pk_ashrrev_i16", ", %4"))
  if (K == 13) BODY(Q4("v_mul_u32_u24", ", %4"))
  if (K == 14) BODY(Q4("v_mad_u32_u24", ", %4, %4"))
  if (K == 15) BODY(Q4("v_and_or_b32", ", %4, %4"))
  if (K == 16) BODY(Q4("v_bfi_b32", ", %4, %4"))
  if (K == 17) BODY(Q4("v_alignbit_b32", ", %4, 8"))
  if (K == 18) BODY(Q4("v_min_i32", ", %4"))
  if (K == 19) BODY(Q4("v_max_i16", ", %4"))
  if (K == 20) BODY(Q4("v_min_f32", ", %4"))
  if (K == 21) BODY(Q4("v_max_f32", ", %4"))
  if (K == 22) BODY("v_cvt_f32_i32 %0, %1\n v_cvt_f32_i32 %1, %2\n v_cvt_f32_i32 %2, %3\n v_cvt_f32_i32 %3, %0")
  if (K == 23) BODY("v_floor_f32 %0, %1\n v_floor_f32 %1, %2\n v_floor_f32 %2, %3\n v_floor_f32 %3, %0")
  if (K == 24) BODY(Q4("v_mul_f32", ", %4"))
  if (K == 25) BODY("v_add_u32 %0, 0x12345, %1\n v_add_u32 %1, 0x12345, %2\n v_add_u32 %2, 0x12345, %3\n v_add_u32 %3, 0x12345, %0")
  if (K == 26) BODY(Q4("v_lshlrev_b32", ", %4") )
  if (K == 27) BODY(Q4("v_or_b32", ", %4"))
  if (K == 28) BODY(Q4("v_mul_i32_i24_sdwa", ", %4 dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:BYTE_1 src1_sel:DWORD"))
  if (K == 29) BODY(Q4("v_add_u32_dpp", ", %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
  if (K == 30) BODY(Q4("v_med3_f32", ", %4, %4"))
  if (K == 31) BODY(Q4("v_cvt_pk_u8_f32", ", 1, %4"))
  if (K == 32) BODY(Q4("v_pk_mul_lo_u16", ", %4"))
  if (K == 33) BODY(Q4("v_add_u16", ", %4"))
  if (K == 34) BODY(Q4("v_ashrrev_i16", ", %4"))
  if (K == 35) BODY(Q4("v_mul_lo_u16", ", %4"))
  if (K == 36) BODY(Q4("v_mad_i16", ", %4, %4"))
  if (K == 37) BODY(Q4("v_pk_fma_f32", ", %4, %4") )   // operates on register pairs? no: 64-bit operands; skipped at run time
  if (K == 38) BODY(Q4("v_add3_u32", ", %4, %4"))
  if (K == 39) BODY(Q4("v_xad_u32", ", %4, %4"))
  if (K == 40) BODY(Q4("v_lshl_or_b32", ", 8, %4"))
  if (K == 41) BODY(Q4("v_perm_b32", ", %4, %4"))
  if (K == 42) BODY("v_subrev_u32 %0, 16, %1\n v_subrev_u32 %1, 16, %2\n v_subrev_u32 %2, 16, %3\n v_subrev_u32 %3, 16, %0")
  if (K == 43) BODY(Q4("v_ashrrev_i32", ", %4"))
  if (K == 44) BODY("v_ashrrev_i32 %0, 8, %1\n v_ashrrev_i32 %1, 8, %2\n v_ashrrev_i32 %2, 8, %3\n v_ashrrev_i32 %3, 8, %0")
  // mixtures as the butterfly has them: 6 plain add/sub per mad+ashr pair
  if (K == 45) BODY("v_add_u32 %0, %1, %4\n v_sub_u32 %1, %2, %4\n v_add_u32 %2, %3, %4\n v_sub_u32 %3, %0, %4\n"
                    "v_add_u32 %0, %1, %4\n v_sub_u32 %1, %2, %4\n v_mad_i32_i24 %2, %3, %5, %4\n v_ashrrev_i32 %3, 8, %0")
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
static int g_waves = 4;
template <int K> void run(const char* name, int per_body = 4) {
  const int iters = 200, blocks = 256 * 4 * g_waves;
  unsigned* d; hipMalloc(&d, (size_t)blocks * 64 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<K><<<blocks, 64>>>(d, 2, 362); hipDeviceSynchronize();
  hipEventRecord(a); k<K><<<blocks, 64>>>(d, iters, 362); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double per_simd = (double)iters * 64 * per_body * g_waves;  // instructions per SIMD
  printf("%-28s %.3f ns per wave-instruction per SIMD\n", name, ms * 1e6 / per_simd);
  hipFree(d);
}
int main(int argc, char** argv) {
  if (argc > 1) g_waves = atoi(argv[1]);
  printf("waves per SIMD: %d\n", g_waves);
  run<0>("v_add_u32"); run<1>("v_sub_u32"); run<2>("v_xor_b32"); run<3>("v_mov_b32"); run<26>("v_lshlrev_b32 vgpr");
  run<27>("v_or_b32"); run<25>("v_add_u32 literal"); run<42>("v_sub_u32 inline const"); run<43>("v_ashrrev_i32 (3, v)");
  run<44>("v_ashrrev_i32 8"); run<33>("v_add_u16"); run<34>("v_ashrrev_i16"); run<35>("v_mul_lo_u16"); run<19>("v_max_i16");
  run<4>("v_fma_f32"); run<5>("v_add_f32"); run<24>("v_mul_f32"); run<20>("v_min_f32"); run<21>("v_max_f32");
  run<22>("v_cvt_f32_i32"); run<23>("v_floor_f32"); run<30>("v_med3_f32"); run<31>("v_cvt_pk_u8_f32");
  run<6>("v_sat_pk_u8_i16"); run<7>("v_dot2_i32_i16"); run<8>("v_pk_max_i16"); run<9>("v_pk_mad_i16");
  run<10>("v_mad_i32_i16"); run<11>("v_pk_add_i16"); run<12>("v_pk_ashrrev_i16"); run<32>("v_pk_mul_lo_u16"); run<36>("v_mad_i16");
  run<13>("v_mul_u32_u24"); run<14>("v_mad_u32_u24"); run<15>("v_and_or_b32"); run<16>("v_bfi_b32");
  run<17>("v_alignbit_b32"); run<18>("v_min_i32"); run<38>("v_add3_u32"); run<39>("v_xad_u32"); run<40>("v_lshl_or_b32"); run<41>("v_perm_b32");
  run<28>("v_mul_i32_i24_sdwa sext16"); run<29>("v_add_u32_dpp quad_perm");
  run<45>("butterfly mix 6 add : mad+ashr", 8);
  return 0;
}
