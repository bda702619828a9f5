// rtj_color_kernels.h — YUV 4:2:0 -> packed RGB, gfx950.
// Restates RTjpeg_yuv420rgb32 / bgr32 / rgb24 / bgr24 / rgb16 (lib/RTjpeg.c:3123-3475; constants
// :3071-3075).  Pure streaming integer work, HBM-bound: each thread converts an 8x2 pixel tile
// (two 8-byte luma loads, one 4-byte load per chroma plane) and writes whole 8/12/16-byte pieces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirtj {

enum { kFmtRGB32 = 0, kFmtBGR32 = 1, kFmtRGB24 = 2, kFmtBGR24 = 3, kFmtRGB16 = 4 };

// The empty asm keeps hipcc (ROCm 7.2) from fusing ">> 16, clamp, pack" into gfx950's
// v_ashr_pk_u8_i32: with that instruction the 24-bit formats came out wrong on the MI355X
// (negative values were not clamped to 0); v_med3_i32 on the shifted value is exact.
__device__ __forceinline__ int clamp255(int v) {
  asm volatile("" : "+v"(v));
  return v > 255 ? 255 : (v < 0 ? 0 : v);
}

struct Rgb {
  int r, g, b;
};
__device__ __forceinline__ Rgb yuv2rgb(int y, int cb, int cr) {
  const int yy = (y - 16) * 76284;
  Rgb o;
  o.r = clamp255((yy + (cr - 128) * 76284) >> 16);  // KcrR == Ky in the reference
  o.g = clamp255((yy - (cr - 128) * 53281 - (cb - 128) * 25625) >> 16);
  o.b = clamp255((yy + (cb - 128) * 132252) >> 16);
  return o;
}

template <int FMT>
__global__ __launch_bounds__(256) void k_yuv420_to_rgb(const uint8_t* __restrict__ planes, size_t in_frame_stride,
                                                        uint8_t* __restrict__ rgb, size_t row_pitch,
                                                        size_t out_frame_stride, int w, int h) {
  const uint8_t* f = planes + (size_t)blockIdx.y * in_frame_stride;
  uint8_t* o = rgb + (size_t)blockIdx.y * out_frame_stride;
  const int tiles_x = w >> 3, tiles = tiles_x * (h >> 1);
  const size_t ysz = (size_t)w * h;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < tiles; t += gridDim.x * blockDim.x) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const uint8_t* py = f + (size_t)(2 * ty) * w + 8 * tx;
    const uint2 y0 = *(const uint2*)py, y1 = *(const uint2*)(py + w);
    const uint32_t cb4 = *(const uint32_t*)(f + ysz + (size_t)ty * (w >> 1) + 4 * tx);
    const uint32_t cr4 = *(const uint32_t*)(f + ysz + (ysz >> 2) + (size_t)ty * (w >> 1) + 4 * tx);
#pragma unroll
    for (int row = 0; row < 2; row++) {
      const uint32_t ylo = row ? y1.x : y0.x, yhi = row ? y1.y : y0.y;
      uint8_t* dst = o + (size_t)(2 * ty + row) * row_pitch;
      Rgb px[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int yv = ((i < 4 ? ylo : yhi) >> (8 * (i & 3))) & 0xFF;
        const int cb = (cb4 >> (8 * (i >> 1))) & 0xFF, cr = (cr4 >> (8 * (i >> 1))) & 0xFF;
        px[i] = yuv2rgb(yv, cb, cr);
      }
      if (FMT == kFmtRGB32 || FMT == kFmtBGR32) {
        // the reference never touches the fourth byte of a pixel: merge with what is there
        uint4* d4 = (uint4*)(dst + 32 * tx);
        uint4 a = d4[0], b = d4[1];
        uint32_t* aw = (uint32_t*)&a;
        uint32_t* bw = (uint32_t*)&b;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t c0 = FMT == kFmtRGB32 ? px[i].r : px[i].b, c2 = FMT == kFmtRGB32 ? px[i].b : px[i].r;
          uint32_t& d = i < 4 ? aw[i] : bw[i - 4];
          d = (d & 0xFF000000u) | c0 | ((uint32_t)px[i].g << 8) | (c2 << 16);
        }
        d4[0] = a;
        d4[1] = b;
      } else if (FMT == kFmtRGB24 || FMT == kFmtBGR24) {
        uint32_t wd[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t c[3] = {(uint32_t)(FMT == kFmtRGB24 ? px[i].r : px[i].b), (uint32_t)px[i].g,
                                 (uint32_t)(FMT == kFmtRGB24 ? px[i].b : px[i].r)};
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const int byte = 3 * i + k;
            wd[byte >> 2] |= c[k] << (8 * (byte & 3));
          }
        }
        uint2* d2 = (uint2*)(dst + 24 * tx);
        d2[0] = make_uint2(wd[0], wd[1]);
        d2[1] = make_uint2(wd[2], wd[3]);
        d2[2] = make_uint2(wd[4], wd[5]);
      } else {  // RGB565, little endian
        uint32_t wd[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint32_t p0 = (px[2 * i].b >> 3) | ((px[2 * i].g >> 2) << 5) | ((px[2 * i].r >> 3) << 11);
          const uint32_t p1 = (px[2 * i + 1].b >> 3) | ((px[2 * i + 1].g >> 2) << 5) | ((px[2 * i + 1].r >> 3) << 11);
          wd[i] = p0 | (p1 << 16);
        }
        *(uint4*)(dst + 16 * tx) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
      }
    }
  }
}

// plain streaming copy, 16 bytes per lane: the yardstick for what a pure HBM-bound kernel sustains
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace mirtj
