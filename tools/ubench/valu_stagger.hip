// Is the slow issue of plain adds behind expensive instructions a property of the WAVE or of the SIMD?
// Pattern: 5 mads, s_nop 4, 35 adds (valu_cluster.hip: ~46 ns when all waves run it in step, ~68 ns without the s_nop).
// Here the waves of a SIMD are started out of step (each first runs blockIdx-dependent filler), or half of them run a
// different instruction mix altogether.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP8(x) x x x x x x x x
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","s4"
#define PAT "v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n s_nop 4\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n"
#define PATN "v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n"
#define ALLMAD "v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n v_mad_i32_i24 v24, v0, s4, v6\n v_mad_i32_i24 v25, v1, s4, v6\n v_mad_i32_i24 v26, v2, s4, v6\n v_mad_i32_i24 v27, v3, s4, v6\n"
#define ALLADD "v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v5\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v5\n v_add_u32 v20, v0, v4\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v4\n v_add_u32 v23, v3, v5\n"
// mode 0: all waves in step; 1: staggered start (k * 7 adds, k = wave number mod 8); 2: odd waves run only mads (and are
// not counted); 3: odd waves run only adds (not counted); 4: staggered, pattern without the s_nop
template <int MODE> __global__ void k(unsigned* out, int iters) {
  asm volatile("s_movk_i32 s4, 362\n v_mov_b32 v0, 1\n v_mov_b32 v1, 2\n v_mov_b32 v2, 3\n v_mov_b32 v3, 4\n v_mov_b32 v4, 5\n v_mov_b32 v5, 6\n v_mov_b32 v6, 128" ::: CLOB);
  const int w = blockIdx.x;
  if (MODE == 1 || MODE == 4) {
    for (int j = 0; j < (w & 7); j++) asm volatile("v_add_u32 v16, v0, v4\n v_add_u32 v17, v1, v4\n v_add_u32 v18, v2, v4\n v_add_u32 v19, v3, v4\n v_add_u32 v20, v0, v5\n v_add_u32 v21, v1, v5\n v_add_u32 v22, v2, v5" ::: CLOB);
  }
  if ((MODE == 2 || MODE == 3) && (w & 1)) {
    for (int i = 0; i < iters; i++) {
      if (MODE == 2) asm volatile(REP8(ALLMAD) ::: CLOB);
      else asm volatile(REP8(ALLADD) ::: CLOB);
    }
  } else {
    for (int i = 0; i < iters; i++) {
      if (MODE == 4) asm volatile(REP8(PATN) ::: CLOB);
      else asm volatile(REP8(PAT) ::: CLOB);
    }
  }
  unsigned r;
  asm volatile("v_add_u32 %0, v16, v24" : "=v"(r)::CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_waves = 4;
template <int MODE> void run(const char* name, double counted_frac) {
  const int iters = 300, blocks = 256 * 4 * g_waves;
  unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<MODE><<<blocks, 64>>>(d, 2); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<MODE><<<blocks, 64>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-58s %6.2f ns of SIMD time per pattern (40 vector instructions) per wave\n", name, ms * 1e6 / ((double)iters * 8 * g_waves));
  (void)hipFree(d);
}
int main(int argc, char** argv) {
  if (argc > 1) g_waves = atoi(argv[1]);
  printf("waves per SIMD: %d\n", g_waves);
  run<0>("all waves in step", 1); run<1>("waves started out of step", 1); run<4>("out of step, no s_nop", 1);
  run<2>("every other wave runs only mads (40 per unit)", 0.5); run<3>("every other wave runs only adds (40 per unit)", 0.5);
  return 0;
}
