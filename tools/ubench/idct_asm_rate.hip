// The hand-ordered transform blocks (csrc/rtj_idct_asm.h) against the compiler's ordering of the same arithmetic
// (idct8 / px in rtj_decode_kernels.h): (1) identical output on random and extreme int16 blocks, (2) time per block,
// coefficients in registers, at 1..5 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../gmerlin-avdecoder_amd/csrc/rtj_decode_kernels.h"
#include "../../gmerlin-avdecoder_amd/csrc/rtj_idct_asm.h"
using namespace mirtj;

template <bool kAsm> __device__ __forceinline__ void block(const uint4 (&q)[8], uint2 (&rows)[8], const IdctK& K) {
  int ws[8][8];
  if (kAsm) {
    int y[8];
    idct8_col<true>(q[0], y, K);
#pragma unroll
    for (int r = 0; r < 8; r++) ws[r][0] = y[r];
#pragma unroll
    for (int c = 1; c < 8; c++) {
      idct8_col<false>(q[c], y, K);
#pragma unroll
      for (int r = 0; r < 8; r++) ws[r][c] = y[r];
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = idct8_row_px(ws[r][0], ws[r][1], ws[r][2], ws[r][3], ws[r][4], ws[r][5], ws[r][6], ws[r][7], K);
  } else {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      int x0 = (int)(int16_t)(q[c].x & 0xFFFFu), x1 = (int)q[c].x >> 16;
      const int x2 = (int)(int16_t)(q[c].y & 0xFFFFu), x3 = (int)q[c].y >> 16;
      const int x4 = (int)(int16_t)(q[c].z & 0xFFFFu), x5 = (int)q[c].z >> 16;
      const int x6 = (int)(int16_t)(q[c].w & 0xFFFFu), x7 = (int)q[c].w >> 16;
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
      rows[r].x = lshl_or(lshl_or(px(y[3]), 8, px(y[2])), 16, lshl_or(px(y[1]), 8, px(y[0])));
      rows[r].y = lshl_or(lshl_or(px(y[7]), 8, px(y[6])), 16, lshl_or(px(y[5]), 8, px(y[4])));
    }
  }
}

template <bool kAsm> __global__ __launch_bounds__(64) void k_check(const uint4* in, uint2* out, int nblk) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nblk) return;
  IdctK K{362, 473, -669, 277, 128, 235};
  uint4 q[8];
  for (int c = 0; c < 8; c++) q[c] = in[(size_t)i * 8 + c];
  uint2 rows[8];
  block<kAsm>(q, rows, K);
  for (int r = 0; r < 8; r++) out[(size_t)i * 8 + r] = rows[r];
}

template <bool kAsm> __global__ __launch_bounds__(64) void k_rate(uint32_t* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_all[];
  IdctK K{362, 473, -669, 277, 128, 235};
  uint4 q[8];
#pragma unroll
  for (int c = 0; c < 8; c++) q[c] = make_uint4(threadIdx.x * 2654435761u + c, blockIdx.x + c * 77u, c * 0x10001u, threadIdx.x * 40503u);
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint2 rows[8];
    block<kAsm>(q, rows, K);
#pragma unroll
    for (int r = 0; r < 8; r++) {  // the next iteration's coefficients: this one's pixels (small values, like real blocks)
      q[r].x = rows[r].x & 0x00FF00FFu;
      q[r].y = rows[r].y & 0x000F00FFu;
      q[r].z = rows[r].x >> 28;
      q[r].w = rows[r].y >> 28;
      acc ^= rows[r].x + rows[r].y;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc + (s_all[0] & 1);
}

template <bool kAsm> double rate(int waves) {
  const int iters = 400, blocks = 256 * 4 * waves * 4;
  uint32_t* d; (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
  size_t lds = (160 * 1024 / (4 * waves)) & ~255;
  (void)hipFuncSetAttribute((const void*)k_rate<kAsm>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k_rate<kAsm><<<blocks, 64, lds>>>(d, 2); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k_rate<kAsm><<<blocks, 64, lds>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipFree(d);
  return ms * 1e6 / ((double)iters * blocks / 1024.0);
}

int main() {
  // ---- 1. same output ----
  const int n = 1 << 18;
  std::vector<int16_t> h((size_t)n * 64);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int i = 0; i < n; i++) {
    const int kind = i & 7;
    for (int k = 0; k < 64; k++) {
      int v;
      if (kind == 0) v = (int)(rnd() % 65536) - 32768;                    // anything an int16 can hold
      else if (kind == 1) v = (rnd() & 1) ? 32767 : -32768;               // extremes
      else if (kind == 2) v = (rnd() % 16 == 0) ? (int)(rnd() % 4096) - 2048 : 0;  // sparse
      else if (kind == 3) v = k == 0 ? (int)(rnd() % 2048) : 0;           // DC only
      else v = (int)(rnd() % 512) - 256 + (k == 0 ? 1000 : 0);            // typical
      h[(size_t)i * 64 + k] = (int16_t)v;
    }
  }
  uint4* d_in; uint2 *d_a, *d_b;
  (void)hipMalloc(&d_in, (size_t)n * 128); (void)hipMalloc(&d_a, (size_t)n * 64); (void)hipMalloc(&d_b, (size_t)n * 64);
  (void)hipMemcpy(d_in, h.data(), (size_t)n * 128, hipMemcpyHostToDevice);
  k_check<false><<<n / 64, 64>>>(d_in, d_a, n);
  k_check<true><<<n / 64, 64>>>(d_in, d_b, n);
  std::vector<uint8_t> a((size_t)n * 64), b((size_t)n * 64);
  (void)hipMemcpy(a.data(), d_a, a.size(), hipMemcpyDeviceToHost);
  (void)hipMemcpy(b.data(), d_b, b.size(), hipMemcpyDeviceToHost);
  size_t bad = 0, first = 0;
  for (size_t i = 0; i < a.size(); i++) if (a[i] != b[i]) { if (!bad) first = i; bad++; }
  printf("output of %d blocks: %zu bytes differ%s\n", n, bad, bad ? "" : " (identical)");
  if (bad) printf("  first at block %zu byte %zu: compiler %u asm %u\n", first / 64, first % 64, a[first], b[first]);
  // ---- 2. time ----
  for (int w : {1, 2, 3, 4, 5}) {
    const double c = rate<false>(w), x = rate<true>(w);
    printf("waves/SIMD %d: compiler order %.1f ns per block-wave per SIMD, hand order %.1f ns (%.2fx)\n", w, c, x, c / x);
  }
  return bad ? 1 : 0;
}
