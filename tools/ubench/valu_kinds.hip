// Issue cost of the VALU instruction forms the decode kernels use, wave64 on gfx950 (8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP8(x) x x x x x x x x
#define BODY(INS) \
  for (int i = 0; i < iters; i++) { REP8(REP8(asm volatile(INS : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(s) : "vcc", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");)) }
template <int K> __global__ void k(unsigned* out, int iters, int s) {
  unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5, e = 7;
  if (K == 0) BODY("v_add_u32 %0, %1, %4\n v_add_u32 %1, %2, %4\n v_add_u32 %2, %3, %4\n v_add_u32 %3, %0, %4")
  if (K == 1) BODY("v_mad_i32_i24 %0, %1, %5, %4\n v_mad_i32_i24 %1, %2, %5, %4\n v_mad_i32_i24 %2, %3, %5, %4\n v_mad_i32_i24 %3, %0, %5, %4")
  if (K == 2) BODY("v_bfe_i32 %0, %1, 3, 16\n v_bfe_i32 %1, %2, 3, 16\n v_bfe_i32 %2, %3, 3, 16\n v_bfe_i32 %3, %0, 3, 16")
  if (K == 3) BODY("v_med3_i32 %0, %1, 16, %4\n v_med3_i32 %1, %2, 16, %4\n v_med3_i32 %2, %3, 16, %4\n v_med3_i32 %3, %0, 16, %4")
  if (K == 4) BODY("v_add_u32_sdwa %0, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
  if (K == 5) BODY("v_lshl_add_u32 %0, %1, 1, %4\n v_lshl_add_u32 %1, %2, 1, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_lshl_add_u32 %3, %0, 1, %4")
  if (K == 6) BODY("v_ashrrev_i32 %0, 8, %1\n v_ashrrev_i32 %1, 8, %2\n v_ashrrev_i32 %2, 8, %3\n v_ashrrev_i32 %3, 8, %0")
  if (K == 7) BODY("v_perm_b32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %4, %5\n v_perm_b32 %2, %3, %4, %5\n v_perm_b32 %3, %0, %4, %5")
  if (K == 8) BODY("v_pk_add_u16 %0, %1, %4\n v_pk_add_u16 %1, %2, %4\n v_pk_add_u16 %2, %3, %4\n v_pk_add_u16 %3, %0, %4")
  if (K == 9) BODY("v_mul_i32_i24 %0, %1, %4\n v_mul_i32_i24 %1, %2, %4\n v_mul_i32_i24 %2, %3, %4\n v_mul_i32_i24 %3, %0, %4")
  if (K == 10) BODY("v_mul_lo_u32 %0, %1, %4\n v_mul_lo_u32 %1, %2, %4\n v_mul_lo_u32 %2, %3, %4\n v_mul_lo_u32 %3, %0, %4")
  if (K == 11) BODY("v_add3_u32 %0, %1, %4, %2\n v_add3_u32 %1, %2, %4, %3\n v_add3_u32 %2, %3, %4, %0\n v_add3_u32 %3, %0, %4, %1")
  if (K == 12) BODY("v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %1, %2, %4, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %0, %4, vcc")
  // three-instruction search-step candidates, 4 chains each (count = 12 instructions per BODY unit)
  if (K == 13) BODY("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, 0, %4, vcc\n v_add_u32 %0, %0, %1\n"
                    "v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, 0, %4, vcc\n v_add_u32 %1, %1, %2\n"
                    "v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, 0, %4, vcc\n v_add_u32 %2, %2, %3\n"
                    "v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %3, 0, %4, vcc\n v_add_u32 %3, %3, %0")
  if (K == 14) BODY("v_sub_u32 %0, %0, %4\n v_bfe_u32 %0, %0, 15, 1\n v_lshl_add_u32 %0, %0, 6, %1\n"
                    "v_sub_u32 %1, %1, %4\n v_bfe_u32 %1, %1, 15, 1\n v_lshl_add_u32 %1, %1, 6, %2\n"
                    "v_sub_u32 %2, %2, %4\n v_bfe_u32 %2, %2, 15, 1\n v_lshl_add_u32 %2, %2, 6, %3\n"
                    "v_sub_u32 %3, %3, %4\n v_bfe_u32 %3, %3, 15, 1\n v_lshl_add_u32 %3, %3, 6, %0")
  if (K == 15) BODY("v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %0, %4")
  if (K == 16) BODY("v_and_b32 %0, 0x8000, %1\n v_and_b32 %1, 0x8000, %2\n v_and_b32 %2, 0x8000, %3\n v_and_b32 %3, 0x8000, %0")
  if (K == 17) BODY("v_and_b32 %0, %5, %1\n v_and_b32 %1, %5, %2\n v_and_b32 %2, %5, %3\n v_and_b32 %3, %5, %0")
  if (K == 18) BODY("v_lshl_or_b32 %0, %1, 8, %4\n v_lshl_or_b32 %1, %2, 8, %4\n v_lshl_or_b32 %2, %3, 8, %4\n v_lshl_or_b32 %3, %0, 8, %4")
  if (K == 19) BODY("v_sub_u16 %0, %1, %4\n v_sub_u16 %1, %2, %4\n v_sub_u16 %2, %3, %4\n v_sub_u16 %3, %0, %4")
  if (K == 20) BODY("v_max_i32 %0, %1, %4\n v_max_i32 %1, %2, %4\n v_max_i32 %2, %3, %4\n v_max_i32 %3, %0, %4")
  if (K == 21) BODY("v_cmp_lt_u32 s[10:11], %0, %4\n v_cndmask_b32 %0, 0, %4, s[10:11]\n v_add_u32 %0, %0, %1\n"
                    "v_cmp_lt_u32 s[12:13], %1, %4\n v_cndmask_b32 %1, 0, %4, s[12:13]\n v_add_u32 %1, %1, %2\n"
                    "v_cmp_lt_u32 s[14:15], %2, %4\n v_cndmask_b32 %2, 0, %4, s[14:15]\n v_add_u32 %2, %2, %3\n"
                    "v_cmp_lt_u32 s[16:17], %3, %4\n v_cndmask_b32 %3, 0, %4, s[16:17]\n v_add_u32 %3, %3, %0")
  if (K == 22) BODY("v_subrev_u32 %0, %4, %0\n v_subrev_u32 %1, %4, %1\n v_xor_b32 %2, %4, %2\n v_lshlrev_b32 %3, 1, %3")
  if (K == 23) BODY("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc")
  if (K == 24) BODY("v_add_u32 %0, %1, %4\n v_bfe_i32 %1, %2, 3, 16\n v_add_u32 %2, %3, %4\n v_bfe_i32 %3, %0, 3, 16")
  if (K == 25) BODY("v_add_u32 %0, %1, %4\n v_add_u32 %1, %2, %4\n v_bfe_i32 %2, %3, 3, 16\n v_bfe_i32 %3, %0, 3, 16")
  if (K == 26) BODY("v_add_u32 %0, %1, %4\n v_add_u32 %1, %2, %4\n v_add_u32 %2, %3, %4\n v_bfe_i32 %3, %0, 3, 16")
  if (K == 27) BODY("v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4")
  if (K == 28) BODY("v_bfe_i32 %0, %0, 3, 16\n v_bfe_i32 %0, %0, 3, 16\n v_bfe_i32 %0, %0, 3, 16\n v_bfe_i32 %0, %0, 3, 16")
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
static int g_waves = 8;
template <int K> void run(const char* name) {
  const int iters = 200, blocks = 256 * 4 * g_waves;
  unsigned* d; hipMalloc(&d, (size_t)blocks * 64 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<K><<<blocks, 64>>>(d, 2, 362); hipDeviceSynchronize();
  hipEventRecord(a); k<K><<<blocks, 64>>>(d, iters, 362); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double per_simd = (double)iters * 64 * 4 * g_waves;  // instructions per SIMD
  printf("%-18s %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  hipFree(d);
}
int main(int argc, char** argv) {
  if (argc > 1) g_waves = atoi(argv[1]);
  printf("waves per SIMD: %d\n", g_waves);
  run<24>("add,bfe alternating"); run<25>("add,add,bfe,bfe"); run<26>("add x3, bfe"); run<27>("add dependent chain"); run<28>("bfe dependent chain");
  run<0>("v_add_u32"); run<6>("v_ashrrev_i32"); run<12>("v_cndmask_b32"); run<9>("v_mul_i32_i24"); run<4>("v_add_u32_sdwa");
  run<1>("v_mad_i32_i24"); run<2>("v_bfe_i32"); run<3>("v_med3_i32"); run<5>("v_lshl_add_u32"); run<11>("v_add3_u32");
  run<7>("v_perm_b32"); run<8>("v_pk_add_u16"); run<10>("v_mul_lo_u32");
  run<15>("v_cmp_lt_u32 e32"); run<16>("v_and_b32 literal"); run<17>("v_and_b32 sgpr"); run<18>("v_lshl_or_b32"); run<19>("v_sub_u16");
  run<20>("v_max_i32"); run<22>("subrev/xor/lshl mix"); run<23>("v_addc_co_u32");
  printf("sequences below: the figure is the cost of one three-instruction search step\n");
  run<13>("cmp+cndmask+add vcc"); run<21>("cmp+cndmask+add sgpr"); run<14>("sub+bfe+lshl_add");
  return 0;
}
