// Issue rate of k_decode's transform section on its own: column pass, row pass, descale/clamp/pack, with the
// coefficient scratch in LDS as in the kernel, at a chosen number of waves per SIMD (dynamic LDS caps it).
// Tells how much of k_decode's time per block is the butterflies themselves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../gmerlin-avdecoder_amd/csrc/rtj_decode_kernels.h"
using namespace mirtj;

template <int MODE> __global__ __launch_bounds__(64) void k(uint32_t* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_all[];
  uint4* my = (uint4*)(s_all + threadIdx.x * 144);
  for (int c = 0; c < 8; c++) my[c] = make_uint4(threadIdx.x * 2654435761u + c, blockIdx.x + c * 77u, c * 0x10001u, threadIdx.x);
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint2 rows[8];
    int ws[8][8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint4 q = my[c];
      int x0 = (int)(int16_t)(q.x & 0xFFFFu), x1 = (int)q.x >> 16;
      const int x2 = (int)(int16_t)(q.y & 0xFFFFu), x3 = (int)q.y >> 16;
      const int x4 = (int)(int16_t)(q.z & 0xFFFFu), x5 = (int)q.z >> 16;
      const int x6 = (int)(int16_t)(q.w & 0xFFFFu), x7 = (int)q.w >> 16;
      if (c == 0) x0 += 4;
      int y[8];
      idct8(x0, x1, x2, x3, x4, x5, x6, x7, y);
#pragma unroll
      for (int r = 0; r < 8; r++) ws[r][c] = y[r];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      int y[8];
      idct8(ws[r][0], ws[r][1], ws[r][2], ws[r][3], ws[r][4], ws[r][5], ws[r][6], ws[r][7], y);
      if (MODE == 0) {
        uint2 o;
        o.x = lshl_or(lshl_or(px(y[3]), 8, px(y[2])), 16, lshl_or(px(y[1]), 8, px(y[0])));
        o.y = lshl_or(lshl_or(px(y[7]), 8, px(y[6])), 16, lshl_or(px(y[5]), 8, px(y[4])));
        rows[r] = o;
      } else {  // butterflies only: fold the row cheaply
        rows[r] = make_uint2((uint32_t)(y[0] ^ y[1] ^ y[2] ^ y[3]), (uint32_t)(y[4] ^ y[5] ^ y[6] ^ y[7]));
      }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) ((uint2*)my)[r] = rows[r];  // next iteration's low coefficients
    acc ^= rows[7].x;
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int MODE> void run(const char* name, int waves, int instr) {
  const int iters = 400, blocks = 256 * 4 * waves * 4;
  uint32_t* d; hipMalloc(&d, (size_t)blocks * 64 * 4);
  // LDS per workgroup so that exactly 4*waves workgroups fit a CU (160 KB)
  size_t lds = (160 * 1024 / (4 * waves)) & ~255;
  if (lds < 64 * 144) { printf("too many waves for the scratch\n"); return; }
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 64, lds>>>(d, 2); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 64, lds>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double blk_per_simd = (double)iters * blocks / 1024.0;
  double ns = ms * 1e6 / blk_per_simd;
  printf("%-28s waves/SIMD %d: %.1f ns per block-wave per SIMD = %.0f cycles @2.4 GHz", name, waves, ns, ns * 2.4);
  if (instr) printf(" (%.2f cycles per VALU instruction, %d counted)", ns * 2.4 / instr, instr);
  printf("\n");
  hipFree(d);
}
// The same transform with the coefficients carried in registers (no LDS at all), so that the number of waves
// per SIMD is limited by registers only: what more occupancy would buy the butterflies.
__global__ __launch_bounds__(64) void k_regs(uint32_t* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_all[];
  uint32_t in[32];
#pragma unroll
  for (int i = 0; i < 32; i++) in[i] = threadIdx.x * 2654435761u + i * 0x9E3779B1u + blockIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    int ws[8][8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint32_t qx = in[4 * c], qy = in[4 * c + 1], qz = in[4 * c + 2], qw = in[4 * c + 3];
      int x0 = (int)(int16_t)(qx & 0xFFFFu), x1 = (int)qx >> 16;
      const int x2 = (int)(int16_t)(qy & 0xFFFFu), x3 = (int)qy >> 16;
      const int x4 = (int)(int16_t)(qz & 0xFFFFu), x5 = (int)qz >> 16;
      const int x6 = (int)(int16_t)(qw & 0xFFFFu), x7 = (int)qw >> 16;
      if (c == 0) x0 += 4;
      int y[8];
      idct8(x0, x1, x2, x3, x4, x5, x6, x7, y);
#pragma unroll
      for (int r = 0; r < 8; r++) ws[r][c] = y[r];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      int y[8];
      idct8(ws[r][0], ws[r][1], ws[r][2], ws[r][3], ws[r][4], ws[r][5], ws[r][6], ws[r][7], y);
      const uint32_t ox = lshl_or(lshl_or(px(y[3]), 8, px(y[2])), 16, lshl_or(px(y[1]), 8, px(y[0])));
      const uint32_t oy = lshl_or(lshl_or(px(y[7]), 8, px(y[6])), 16, lshl_or(px(y[5]), 8, px(y[4])));
      in[2 * r] = ox;  // the next iteration's coefficients
      in[2 * r + 1] = oy;
      acc ^= ox + oy;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc + (s_all[0] & 1);
}

void run_regs(int waves, int instr) {
  const int iters = 400, blocks = 256 * 4 * waves * 4;
  uint32_t* d; hipMalloc(&d, (size_t)blocks * 64 * 4);
  size_t lds = (160 * 1024 / (4 * waves)) & ~255;
  hipFuncSetAttribute((const void*)k_regs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_regs<<<blocks, 64, lds>>>(d, 2); hipDeviceSynchronize();
  hipEventRecord(a); k_regs<<<blocks, 64, lds>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ns = ms * 1e6 / ((double)iters * blocks / 1024.0);
  printf("%-28s waves/SIMD %d: %.1f ns per block-wave per SIMD = %.0f cycles @2.4 GHz", "registers only", waves, ns, ns * 2.4);
  if (instr) printf(" (%.2f cycles per VALU instruction, %d counted)", ns * 2.4 / instr, instr);
  printf("\n");
  hipFree(d);
}

int main(int argc, char** argv) {
  int n0 = argc > 1 ? atoi(argv[1]) : 0, n1 = argc > 2 ? atoi(argv[2]) : 0;  // VALU per iteration, from the disassembly
  for (int w : {1, 2, 3, 4, 5}) run<0>("transform+descale+pack", w, n0);
  for (int w : {1, 2, 4}) run<1>("butterflies only", w, n1);
  for (int w : {1, 2, 3, 4, 5, 6}) run_regs(w, argc > 3 ? atoi(argv[3]) : 0);
  return 0;
}
