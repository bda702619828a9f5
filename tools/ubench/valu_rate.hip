// Measures wave64 integer VALU issue rate on gfx950: cycles per instruction per SIMD at several occupancies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int KIND>
__global__ void k(unsigned* out, int iters) {
  unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5, e = 7, f = 11, g = 13, h = 17;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (KIND == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }            // v_add_u32
      if (KIND == 1) { a = __builtin_amdgcn_ubfe(a + b, 3, 16); c = __builtin_amdgcn_ubfe(c + d, 3, 16);
                       e = __builtin_amdgcn_ubfe(e + f, 3, 16); g = __builtin_amdgcn_ubfe(g + h, 3, 16);
                       b += a; d += c; f += e; h += g; }                                              // bfe mix
      if (KIND == 2) { int r; asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); a = r;
                       asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(d), "v"(e), "v"(f)); d = r;
                       asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(g), "v"(h), "v"(a)); g = r;
                       asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(c), "v"(d)); b = r;
                       asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(e), "v"(f), "v"(g)); e = r;
                       asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(h), "v"(a), "v"(b)); h = r;
                       asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(c), "v"(d), "v"(e)); c = r;
                       asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(g), "v"(h)); f = r; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int KIND> void run(const char* name, int waves_per_simd) {
  int iters = 2000;
  int blocks = 256 * 4 * waves_per_simd;  // 64-thread blocks: one wave each
  unsigned* d; hipMalloc(&d, (size_t)blocks * 64 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<KIND><<<blocks, 64>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(a); k<KIND><<<blocks, 64>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double insts_per_simd = (double)iters * 16 * 8 * waves_per_simd;
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) run<0>("v_add_u32", w);
  for (int w : {1, 4, 8}) run<1>("add+bfe", w);
  for (int w : {1, 4, 8}) run<2>("mad_i24", w);
  return 0;
}
