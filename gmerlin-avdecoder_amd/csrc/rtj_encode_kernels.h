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

// The same pictures with the noise of SURVEY.md section 8d / BASELINE.md section 2: ONE linear congruential sequence
// s <- s * 1664525 + 1013904223 (mod 2^32) from `seed`, one draw per sample in the order frame 0 Y (row major), U, V,
// frame 1 ..., noise = ((s >> 8) mod (2a + 1)) - a with a = amp (luma) / amp / 2 (chroma).  A thread jumps to the
// state of its first sample — s_k = A^k s_0 + C (A^k - 1) / (A - 1), by squaring: 32 steps — and draws kLcgRun samples in
// a row.  (The gradient wraps mod w + h as in k_synth: BASELINE.md's unwrapped formula saturates after a few hundred
// frames.)
constexpr uint32_t kLcgA = 1664525u, kLcgC = 1013904223u;
constexpr int kLcgRun = 256;
__device__ __forceinline__ uint32_t lcg_skip(uint32_t s, uint64_t k) {
  uint32_t a = kLcgA, c = kLcgC, A = 1u, C = 0u;  // x -> A x + C is k steps so far; x -> a x + c is 2^i steps
  for (; k; k >>= 1) {
    if (k & 1u) {
      A = A * a;
      C = C * a + c;
    }
    c = c * (a + 1u);
    a = a * a;
  }
  return A * s + C;
}
__global__ void k_synth_lcg(uint8_t* __restrict__ frames, int w, int h, int first_frame, int nframes, uint32_t seed, int amp) {
  const size_t ysz = (size_t)w * h, csz = ysz >> 2, fsz = ysz + 2 * csz, total = fsz * (size_t)nframes;
  for (size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * kLcgRun; i0 < total; i0 += (size_t)gridDim.x * blockDim.x * kLcgRun) {
    uint32_t s = lcg_skip(seed, (uint64_t)first_frame * fsz + i0);  // the state BEFORE this thread's first draw
    for (size_t i = i0; i < i0 + kLcgRun && i < total; i++) {
      s = s * kLcgA + kLcgC;
      const size_t fr = i / fsz, o = i - fr * fsz;
      const uint32_t n = (uint32_t)first_frame + (uint32_t)fr;
      int a, basev;
      if (o < ysz) {
        const uint32_t x = (uint32_t)(o % (uint32_t)w), y = (uint32_t)(o / (uint32_t)w);
        a = amp;
        basev = 16 + (int)(((x + y + 7u * n) % (uint32_t)(w + h)) * 219u / (uint32_t)(w + h));
      } else {
        a = amp / 2;
        basev = 128;
      }
      int v = basev + (int)((s >> 8) % (uint32_t)(2 * a + 1)) - a;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      frames[i] = (uint8_t)v;
    }
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

// The intra form, one WAVE per 64 horizontally adjacent blocks (the arrangement of k_decode: a macroblock group of 32
// macroblocks has three parts — upper luma, lower luma, chroma — and a lane holds one block of its part), round 4:
//   * a lane reads its block as eight 8-byte pieces of rows that the wave's lanes cover contiguously (k_encode_blocks
//     reads byte by byte, one thread per block in stream order);
//   * both transform passes and the quantiser in registers, every index a compile-time constant;
//   * the run-length pack walks the zig-zag order fully unrolled (the coefficient of a slot is a REGISTER, not an
//     indexed array in scratch memory) and appends bytes to the lane's 64-byte slot in LDS under lane predicates;
//   * the slot leaves as four 16-byte stores.
// Same arithmetic, same bytes (lib/RTjpeg.c:288-389 dctY, :245-252 quant, :109-155 b2s).  grid (3 * groups, frames).
constexpr int kEncSlotStride = 80;  // LDS bytes per lane: a 64-byte slot + padding (16-byte reads without bank conflicts)
__global__ __launch_bounds__(64) void k_encode_wave(const uint8_t* __restrict__ frames, int w, int h,
                                                     const QTab* __restrict__ qt, uint8_t* __restrict__ slots,
                                                     uint8_t* __restrict__ lens) {
  __shared__ __attribute__((aligned(16))) uint8_t s_slot[64 * kEncSlotStride];
  const int lane = threadIdx.x;
  const uint32_t mbw = (uint32_t)w / 16u, nmb = mbw * ((uint32_t)h / 16u);
  const uint32_t grp = blockIdx.x / 3u, part = blockIdx.x - 3u * grp, fr = blockIdx.y;
  const uint32_t dmb = part == 2u ? (uint32_t)(lane & 31) : (uint32_t)(lane >> 1);
  const uint32_t k = part == 2u ? 4u + (uint32_t)(lane >> 5) : 2u * part + (uint32_t)(lane & 1);
  const uint32_t mb = grp * (uint32_t)kMbPerGroup + dmb;
  if (mb >= nmb) return;  // (no barrier below: a wave's lanes only touch their own slots)
  const uint32_t my = mb / mbw, mx = mb - my * mbw;
  const size_t ysz = (size_t)w * h;
  const uint8_t* f = frames + (size_t)fr * (ysz + (ysz >> 1));
  const uint8_t* src;
  uint32_t stride;
  if (k < 4u) {
    stride = (uint32_t)w;
    src = f + (size_t)(16u * my + 8u * (k >> 1)) * w + 16u * mx + 8u * (k & 1u);
  } else {
    stride = (uint32_t)w >> 1;
    src = f + ysz + (k == 5u ? ysz >> 2 : 0) + (size_t)(8u * my) * stride + 8u * mx;
  }
  int ws[64];
#pragma unroll
  for (int row = 0; row < 8; row++) {
    const uint2 d = *(const uint2*)(src + (size_t)row * stride);  // 8-byte aligned: widths are multiples of 16
    int p[8] = {(int)(d.x & 255u), (int)((d.x >> 8) & 255u), (int)((d.x >> 16) & 255u), (int)(d.x >> 24),
                (int)(d.y & 255u), (int)((d.y >> 8) & 255u), (int)((d.y >> 16) & 255u), (int)(d.y >> 24)};
    int r[8];
    fdct8(p, r);
    r[0] <<= 8;
    r[4] <<= 8;
#pragma unroll
    for (int c = 0; c < 8; c++) ws[8 * row + c] = r[c];
  }
  const int32_t* q = k < 4u ? qt->lqt : qt->cqt;
  int blk[64];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    int p[8], r[8];
#pragma unroll
    for (int kk = 0; kk < 8; kk++) p[kk] = ws[8 * kk + c];
    fdct8(p, r);
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const int16_t d = (kk == 0 || kk == 4) ? (int16_t)((r[kk] + 128) >> 8) : (int16_t)((r[kk] + 32768) >> 16);
      blk[8 * kk + c] = (int)(int16_t)(((int)d * q[8 * kk + c] + 32767) >> 16);  // RTjpeg_quant
    }
  }
  // RTjpeg_b2s: DC clamped to 0..254, bt8 full-range bytes, then 7-bit values and zero runs (63 + run)
  constexpr uint8_t zz[64] = MIRTJ_ZZ_INIT;
  const int bt8 = k < 4u ? qt->lb8 : qt->cb8;  // the same for all lanes of a part
  uint8_t* const out = s_slot + lane * kEncSlotStride;
  int n = 0, run = 0;
  {
    const int v = blk[zz[0]];
    out[n++] = (uint8_t)(v > 254 ? 254 : (v < 0 ? 0 : v));
  }
#pragma unroll
  for (int z = 1; z < 64; z++) {
    const int v = blk[zz[z]];
    if (z <= kMaxRawBytes && z <= bt8) {  // (bt8 <= kMaxRawBytes: the host refuses other tables)
      out[n++] = (uint8_t)(int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
    } else if (v != 0) {
      if (run) {
        out[n++] = (uint8_t)(63 + run);
        run = 0;
      }
      out[n++] = (uint8_t)(int8_t)(v > 63 ? 63 : (v < -64 ? -64 : v));
    } else {
      run++;
    }
  }
  if (run) out[n++] = (uint8_t)(63 + run);
  const size_t gb = (size_t)fr * nmb * 6u + 6u * mb + k;
  lens[gb] = (uint8_t)n;
  const uint4* o4 = (const uint4*)out;
  uint4* g4 = (uint4*)(slots + gb * 64);
#pragma unroll
  for (int i = 0; i < 4; i++) g4[i] = o4[i];
}

// Where the packets of a pass go: frame sizes (k_encode_scan) -> packet offsets in the output stream, each aligned to
// `align`, continuing at the stream's cursor; both are kept on the device for all frames of the call, so that the host
// has nothing to wait for between passes (round 3 synchronised twice per 32 frames).  One workgroup.
__global__ __launch_bounds__(256) void k_encode_place(const uint32_t* __restrict__ frame_bytes, int m, uint32_t align,
                                                       uint64_t* __restrict__ cursor, uint64_t* __restrict__ pkt_off,
                                                       uint32_t* __restrict__ pkt_len, uint64_t* __restrict__ pass_off) {
  if (threadIdx.x != 0) return;  // (a pass has at most a few hundred frames: a serial walk of them is microseconds)
  uint64_t cur = *cursor;
  for (int i = 0; i < m; i++) {
    cur = (cur + align - 1u) / align * align;
    const uint32_t len = frame_bytes[i] + 12u;
    pkt_off[i] = cur;
    pass_off[i] = cur;
    pkt_len[i] = len;
    cur += len;
  }
  *cursor = cur;
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
