// rtj_encode_kernels.h — gfx950 kernels of the stream generator:
// synthetic frame content and the intra-only RTjpeg encoder (lib/RTjpeg.c:109-155, 245-252,
// 288-389, 2510-2563, 3488-3524).  Not on the timed path: it exists so that benchmark-size
// streams can be made on the GPU box, and as the mirror image of the decode kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "rtj_common.h"
#include "rtj_decode_kernels.h"  // c_zz

namespace mirtj {

// ---- synthetic content (the tests hold a numpy twin of this generator) ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}

__global__ void k_synth(uint8_t* __restrict__ frames, int w, int h, int first_frame, uint32_t seed, int amp) {
  const uint32_t n = first_frame + blockIdx.y;
  const size_t ysz = (size_t)w * h, csz = ysz >> 2, fsz = ysz + 2 * csz;
  uint8_t* f = frames + (size_t)blockIdx.y * fsz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < fsz; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t plane, idx;
    int a, basev;
    if (i < ysz) {
      plane = 0;
      idx = (uint32_t)i;
      a = amp;
      const uint32_t x = idx % (uint32_t)w, y = idx / (uint32_t)w;
      basev = 16 + (int)(((x + y + 7u * n) % (uint32_t)(w + h)) * 219u / (uint32_t)(w + h));
    } else {
      plane = i < ysz + csz ? 1u : 2u;
      idx = (uint32_t)(i - ysz - (plane == 2u ? csz : 0));
      a = amp / 2;
      basev = 128;
    }
    const uint32_t key = seed * 0x9E3779B1u + n * 0x7FEB352Du + plane * 0x846CA68Bu;
    const int noise = (int)(mix32(idx + key) % (uint32_t)(2 * a + 1)) - a;
    int v = basev + noise;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    f[i] = (uint8_t)v;
  }
}

// ---- forward AAN butterfly (lib/RTjpeg.c:301-336 rows, :341-385 columns) ----
__device__ __forceinline__ void fdct8(const int (&p)[8], int (&r)[8]) {
  const int a0 = p[0] + p[7], a7 = p[0] - p[7], a1 = p[1] + p[6], a6 = p[1] - p[6];
  const int a2 = p[2] + p[5], a5 = p[2] - p[5], a3 = p[3] + p[4], a4 = p[3] - p[4];
  const int b0 = a0 + a3, b3 = a0 - a3, b1 = a1 + a2, b2 = a1 - a2;
  r[0] = b0 + b1;
  r[4] = b0 - b1;
  const int z1 = (b2 + b3) * 181;
  r[2] = (b3 << 8) + z1;
  r[6] = (b3 << 8) - z1;
  const int c0 = a4 + a5, c1 = a5 + a6, c2 = a6 + a7;
  const int z5 = (c0 - c2) * 98;
  const int z2 = c0 * 139 + z5, z4 = c2 * 334 + z5, z3 = c1 * 181;
  const int z11 = (a7 << 8) + z3, z13 = (a7 << 8) - z3;
  r[5] = z13 + z2;
  r[3] = z13 - z2;
  r[1] = z11 + z4;
  r[7] = z11 - z4;
}

// One thread per 8x8 block: DCT + quantise + run-length pack into a private 64-byte slot.
// blocks are numbered in stream order (6 per macroblock) across all frames of the call.
// With `old` (inter mode, RTjpeg_mcompressYUV420 lib/RTjpeg.c:2841-2921 + RTjpeg_bcomp :2827-2838) the
// call covers ONE frame: a block whose quantised coefficients all lie within `mask` of the stored
// previous ones is emitted as the single byte 0xFF and the store is left alone; otherwise the store
// takes the new block.
__global__ void k_encode_blocks(const uint8_t* __restrict__ frames, int w, int h, int nframes,
                                const QTab* __restrict__ qt, uint8_t* __restrict__ slots,
                                uint8_t* __restrict__ lens, int16_t* __restrict__ old, int lmask, int cmask) {
  const uint32_t nmb = (uint32_t)(w / 16) * (h / 16), mbw = w / 16;
  const size_t total = (size_t)nframes * nmb * 6;
  const size_t gb = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gb >= total) return;
  const uint32_t fr = (uint32_t)(gb / (nmb * 6u)), bi = (uint32_t)(gb % (nmb * 6u));
  const uint32_t mb = bi / 6u, k = bi % 6u, mx = mb % mbw, my = mb / mbw;
  const size_t ysz = (size_t)w * h;
  const uint8_t* f = frames + (size_t)fr * (ysz + (ysz >> 1));
  const uint8_t* src;
  int stride;
  if (k < 4) {
    stride = w;
    src = f + (size_t)(16u * my + 8u * (k >> 1)) * w + 16u * mx + 8u * (k & 1u);
  } else {
    stride = w >> 1;
    src = f + ysz + (k == 5 ? ysz >> 2 : 0) + (size_t)(8u * my) * stride + 8u * mx;
  }
  int ws[64];
#pragma unroll
  for (int row = 0; row < 8; row++) {
    int p[8], r[8];
#pragma unroll
    for (int c = 0; c < 8; c++) p[c] = src[(size_t)row * stride + c];
    fdct8(p, r);
    r[0] <<= 8;
    r[4] <<= 8;
#pragma unroll
    for (int c = 0; c < 8; c++) ws[8 * row + c] = r[c];
  }
  const int32_t* q = k < 4 ? qt->lqt : qt->cqt;
  int16_t blk[64];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    int p[8], r[8];
#pragma unroll
    for (int kk = 0; kk < 8; kk++) p[kk] = ws[8 * kk + c];
    fdct8(p, r);
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const int16_t d = (kk == 0 || kk == 4) ? (int16_t)((r[kk] + 128) >> 8) : (int16_t)((r[kk] + 32768) >> 16);
      blk[8 * kk + c] = (int16_t)(((int)d * q[8 * kk + c] + 32767) >> 16);  // RTjpeg_quant
    }
  }
  uint8_t* out = slots + gb * 64;
  if (old) {
    int16_t* o = old + gb * 64;
    const int mask = k < 4 ? lmask : cmask;
    bool same = true;
    for (int i = 0; i < 64; i++) {
      const int d = (int)o[i] - (int)blk[i];
      same = same && (d < 0 ? -d : d) <= mask;
    }
    if (same) {
      out[0] = 0xFF;
      lens[gb] = 1;
      return;
    }
    for (int i = 0; i < 64; i++) o[i] = blk[i];
  }
  // RTjpeg_b2s: DC clamped to 0..254, bt8 full-range bytes, then 7-bit values and zero runs
  const int bt8 = k < 4 ? qt->lb8 : qt->cb8;
  int n = 0, z = 1;
  int v = blk[c_zz[0]];
  out[n++] = (uint8_t)(v > 254 ? 254 : (v < 0 ? 0 : v));
  for (; z <= bt8; z++) {
    v = blk[c_zz[z]];
    out[n++] = (uint8_t)(int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
  }
  while (z < 64) {
    v = blk[c_zz[z]];
    if (v != 0) {
      out[n++] = (uint8_t)(int8_t)(v > 63 ? 63 : (v < -64 ? -64 : v));
      z++;
    } else {
      int run = 0;
      while (z < 64 && blk[c_zz[z]] == 0) {
        z++;
        run++;
      }
      out[n++] = (uint8_t)(63 + run);
    }
  }
  lens[gb] = (uint8_t)n;
}

// One workgroup per frame: exclusive scan of the block lengths -> block offsets (relative to the
// first data byte) and the frame's data size.
__global__ __launch_bounds__(256) void k_encode_scan(const uint8_t* __restrict__ lens, uint32_t nblk,
                                                      uint32_t* __restrict__ offs,
                                                      uint32_t* __restrict__ frame_bytes) {
  __shared__ uint32_t part[256];
  const uint8_t* l = lens + (size_t)blockIdx.x * nblk;
  uint32_t* o = offs + (size_t)blockIdx.x * nblk;
  const uint32_t per = (nblk + 255u) / 256u;
  const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nblk);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += l[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; i++) {
      const uint32_t t = part[i];
      part[i] = run;
      run += t;
    }
    frame_bytes[blockIdx.x] = run;
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (uint32_t i = lo; i < hi; i++) {
    o[i] = run;
    run += l[i];
  }
}

// Compaction: block bytes to their final place, plus the 12-byte frame header
// (RTjpeg_frameheader, include/RTjpeg.h:100-109; written at lib/RTjpeg.c:3516-3522).
__global__ void k_encode_pack(const uint8_t* __restrict__ slots, const uint8_t* __restrict__ lens,
                              const uint32_t* __restrict__ offs, const uint64_t* __restrict__ pkt_off,
                              const uint32_t* __restrict__ frame_bytes, uint32_t nblk, int w, int h, int Q,
                              int key, uint8_t* __restrict__ stream) {
  const uint32_t fr = blockIdx.y;
  uint8_t* pkt = stream + pkt_off[fr];
  if (blockIdx.x == 0 && threadIdx.x < 12) {
    const uint32_t total = frame_bytes[fr] + 12u;
    const uint8_t hdr[12] = {(uint8_t)total, (uint8_t)(total >> 8), (uint8_t)(total >> 16), (uint8_t)(total >> 24),
                             12, 0, (uint8_t)w, (uint8_t)(w >> 8), (uint8_t)h, (uint8_t)(h >> 8), (uint8_t)Q, (uint8_t)key};
    pkt[threadIdx.x] = hdr[threadIdx.x];
  }
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const size_t gb = (size_t)fr * nblk + b;
  const uint8_t* s = slots + gb * 64;
  uint8_t* d = pkt + 12 + offs[gb];
  const int n = lens[gb];
  for (int i = 0; i < n; i++) d[i] = s[i];
}

}  // namespace mirtj
