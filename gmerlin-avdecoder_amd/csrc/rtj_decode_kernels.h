// rtj_decode_kernels.h — gfx950 kernels of the RTjpeg decode path.
//
//   k_index_walk   block-start index of every packet: RTjpeg_s2b's length rule
//                  (lib/RTjpeg.c:157-186) driven by RTjpeg_decompressYUV420's block order
//                  (lib/RTjpeg.c:2688-2749).
//   k_decode       dequantise + 8x8 AAN inverse transform + plane scatter
//                  (lib/RTjpeg.c:157-186, 2209-2332, 2688-2749), one lane per 8x8 block.
//
// Integer only; no MFMA: the transform is a fixed 8-point butterfly network with rounding
// after every product, not a contraction.
#pragma once
#include <hip/hip_runtime.h>

#include "rtj_common.h"

namespace mirtj {

__constant__ uint8_t c_zz[64] = MIRTJ_ZZ_INIT;

// ---------------------------------------------------------------------------------------
// wave64 inclusive prefix sum with DPP row shifts + row broadcasts (no LDS)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  // within each row of 16 lanes: Hillis-Steele with row_shr 1,2,4,8 (out-of-row sources read 0)
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
  // lane 15 of rows 0 and 2 into every lane of rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
  // lane 31 into every lane of rows 2 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
  return x;
}

// number of coefficient slots a stream byte covers when it is read as a token (lib/RTjpeg.c:171-182):
// 64..127 (int8 > 63) is a zero run of b-63, anything else is one coefficient
__device__ __forceinline__ uint32_t token_weight(uint32_t b) {
  return (b - 64u < 64u) ? b - 63u : 1u;
}

// ---------------------------------------------------------------------------------------
// walk_blocks: one wave walks a stretch of the block chain of one packet.
//
// Lane i of the wave holds byte base+i of the stream (two 64-byte windows, "cur" and "nxt")
// and the running sum W of token weights.  A block that starts at p with bt8 raw bytes ends at
// the first byte e whose W reaches W[p+bt8] + (63-bt8): one compare + find-first-set per block.
// The chain itself is serial (a block's length is only known once it has been read), so the
// position lives in scalar registers and the vector unit is used 64 bytes at a time.
// Walks blocks k0..k1-1 (k0 a multiple of 6) starting at byte p_start and writes out[k] = byte
// offset of block k relative to the first data byte; with write_end also out[k1] = end position.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void walk_blocks(const FrameDev& f, const uint8_t* __restrict__ stream,
                                            const QTab* __restrict__ lut, uint32_t p_start, uint32_t k0,
                                            uint32_t k1, bool write_end, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint8_t* g = stream + f.data_off;
  const uint32_t len = f.data_len;
  const uint32_t lb8 = (uint32_t)lut[f.qidx].lb8, cb8 = (uint32_t)lut[f.qidx].cb8;

  auto fetch = [&](uint32_t pos) -> uint32_t {
    const uint32_t i = pos + lane;
    return i < len ? (uint32_t)g[i] : 0u;
  };

  uint32_t base = p_start;
  uint32_t cur = fetch(base), nxt = fetch(base + 64u), pf1 = fetch(base + 128u), pf2 = fetch(base + 192u),
           pf3 = fetch(base + 256u);
  uint32_t Wc = wave_incl_scan(token_weight(cur));
  uint32_t Wn = wave_incl_scan(token_weight(nxt)) + (uint32_t)__builtin_amdgcn_readlane((int)Wc, 63);

  uint32_t p = p_start;  // current block start (uniform)
  uint32_t ph = 0;       // block number within the macroblock, 0..5
  uint32_t acc = 0;      // up to 64 offsets gathered one lane at a time, stored 256 B at once

  for (uint32_t k = k0;; ++k) {
    while (p - base >= 64u) {  // slide the two windows forward
      base += 64u;
      cur = nxt;
      Wc = Wn;
      nxt = pf1;
      pf1 = pf2;
      pf2 = pf3;
      pf3 = fetch(base + 256u);
      Wn = wave_incl_scan(token_weight(nxt)) + (uint32_t)__builtin_amdgcn_readlane((int)Wc, 63);
    }
    const bool last = k == k1;
    if (!last || write_end) acc = (uint32_t)lane == (k & 63u) ? p : acc;
    if ((k & 63u) == 63u || last) {
      const uint32_t idx = (k & ~63u) + (uint32_t)lane;
      const uint32_t top = last && !write_end ? k - 1u : k;  // k1 > k0 >= 0, so no wrap
      if (idx >= k0 && idx <= top) out[idx] = acc;
    }
    if (last) break;

    const uint32_t lp = p - base;
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)lp);
    const uint32_t bt8 = ph >= 4u ? cb8 : lb8;
    ph = ph == 5u ? 0u : ph + 1u;
    if (b0 == 0xFFu) {  // "unchanged" block: a single byte
      p += 1u;
      continue;
    }
    const uint32_t need = 63u - bt8;
    if (need == 0u) {  // every coefficient is a raw byte
      p += 64u;
      continue;
    }
    const uint32_t iq = lp + bt8;  // last non-token byte of the block, 0..126
    const uint32_t Wq = iq < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)Wc, (int)iq)
                                 : (uint32_t)__builtin_amdgcn_readlane((int)Wn, (int)(iq - 64u));
    const uint32_t target = Wq + need;
    const unsigned long long mc = __ballot(Wc >= target);
    uint32_t e;
    if (mc) {
      e = (uint32_t)__builtin_ctzll(mc);
    } else {
      const unsigned long long mn = __ballot(Wn >= target);
      e = 64u + (uint32_t)__builtin_ctzll(mn);  // W grows by >= 1 per byte, so mn != 0
    }
    p = base + e + 1u;
  }
}

// Whole packet by one wave: the simple (serial) index, kept as the A/B baseline of the parallel one.
__global__ __launch_bounds__(64) void k_index_walk(const FrameDev* __restrict__ frames,
                                                    const uint8_t* __restrict__ stream,
                                                    const QTab* __restrict__ lut,
                                                    uint32_t* __restrict__ blkoff) {
  const FrameDev f = frames[blockIdx.x];
  walk_blocks(f, stream, lut, 0u, 0u, f.nmb * 6u, true, blkoff + f.blk_base);
}

// ---------------------------------------------------------------------------------------
// 8-point inverse AAN butterfly (lib/RTjpeg.c:2240-2283 for columns, :2290-2327 for rows;
// constants :1196-1199, MULTIPLY :1206).  Every multiplicand stays below 2^23 in magnitude
// for any int16 input block (tests/test_bounds.py), so the 24-bit multiplier is exact and
// wraps like the reference's 32-bit product.
// ---------------------------------------------------------------------------------------
// v_mad_i32_i24 is spelled out: hipcc's value tracking cannot see the 2^23 bound and would
// otherwise fall back to the quarter-rate 32-bit multiplier for the whole row pass.
__device__ __forceinline__ int mulr8(int x, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(c), "v"(128));
  return r >> 8;
}

__device__ __forceinline__ int mul24(int a, int b) {  // both operands inside the signed 24-bit range
  int r;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void idct8(int x0, int x1, int x2, int x3, int x4, int x5, int x6, int x7,
                                      int (&y)[8]) {
  const int s04 = x0 + x4, d04 = x0 - x4;
  const int s26 = x2 + x6;
  const int r26 = mulr8(x2 - x6, 362) - s26;
  const int e0 = s04 + s26, e3 = s04 - s26, e1 = d04 + r26, e2 = d04 - r26;
  const int s53 = x5 + x3, d53 = x5 - x3, s17 = x1 + x7, d17 = x1 - x7;
  const int o7 = s17 + s53;
  const int m = mulr8(s17 - s53, 362);
  const int z5 = mulr8(d53 + d17, 473);
  const int o6 = mulr8(d53, -669) + z5 - o7;
  const int o5 = m - o6;
  const int o4 = mulr8(d17, 277) - z5 + o5;
  y[0] = e0 + o7; y[7] = e0 - o7;
  y[1] = e1 + o6; y[6] = e1 - o6;
  y[2] = e2 + o5; y[5] = e2 - o5;
  y[4] = e3 + o4; y[3] = e3 - o4;
}

// DESCALE + int16 narrowing + clamp 16..235 (lib/RTjpeg.c:1201-1205).  The +4 rounding term
// was folded into the DC coefficient before the column pass, so only the shift remains:
// bits [18:3] sign-extended == (int16_t)(v >> 3).
__device__ __forceinline__ uint32_t px(int v) {
  int s = (int)((uint32_t)v << 13) >> 16;
  s = s > 235 ? 235 : s;
  s = s < 16 ? 16 : s;
  return (uint32_t)s;
}

constexpr int kDecThreads = 64;
#ifndef MIRTJ_COEF_STRIDE
#define MIRTJ_COEF_STRIDE 72
#endif
#ifndef MIRTJ_DEC_WAVES
#define MIRTJ_DEC_WAVES 1
#endif
#ifndef MIRTJ_DEC_ITERS
#define MIRTJ_DEC_ITERS 4
#endif
constexpr int kCoefStride = MIRTJ_COEF_STRIDE;  // int16 per lane: 64 + 8 pad (144 B, conflict-free b128 reads)
constexpr int kDecIters = MIRTJ_DEC_ITERS;      // macroblock groups a wave works through, one after the other

// ---------------------------------------------------------------------------------------
// k_decode: grid (3 * slots, frames), one wave per workgroup; slots = groups / kDecIters rounded up
// to a multiple of 8.  A group is kMbPerGroup consecutive macroblocks; a wave owns one PART of
// kDecIters groups (slot, slot + slots, ...):
//   part 0: the 64 upper luma blocks (Y0,Y1 of each MB), part 1: the 64 lower ones,
//   part 2: 32 Cb + 32 Cr blocks.
// Lanes of a wave hold horizontally adjacent blocks, so every row store of a wave covers 512
// (luma) or 2x256 (chroma) contiguous bytes.  Each lane pulls its own block's bytes from the
// stream (neighbouring lanes read neighbouring bytes, so the wave's loads stay within a few
// cache lines), parses them into a private LDS scratch (transposed, so a column is one 16-byte
// read), then runs both transform passes entirely in registers.
//
// A wave works through several groups so that it can request the next group's block offset before
// parsing and the next group's stream bytes before transforming: only the first group pays the
// three dependent loads (descriptor -> block offset -> stream bytes).  Measured: 2 % over one
// group per wave; eight groups per wave lose it again (fewer, longer waves balance worse).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kDecThreads, MIRTJ_DEC_WAVES) void k_decode(const FrameDev* __restrict__ frames,
                                                         const uint8_t* __restrict__ stream,
                                                         const QTab* __restrict__ lut,
                                                         const uint32_t* __restrict__ blkoff,
                                                         uint8_t* __restrict__ outbuf) {
  __shared__ __attribute__((aligned(16))) int16_t s_coef[kDecThreads * kCoefStride];
  __shared__ uint32_t s_tab[64];  // per zig-zag slot: (dequantiser << 8) | transposed position

  const FrameDev f = frames[blockIdx.y];
  // part-major numbering with the slot count a multiple of 8: the three waves that share a set
  // of groups get linear ids that differ by a multiple of 8, i.e. (as workgroups are dealt
  // round-robin to the 8 XCDs) they share one L2 and the stream bytes come from HBM once
  const uint32_t slots = gridDim.x / 3u;
  const uint32_t part = blockIdx.x / slots, slot = blockIdx.x - part * slots;
  const uint32_t ngroups = (f.nmb + (uint32_t)kMbPerGroup - 1u) / (uint32_t)kMbPerGroup;
  if (slot >= ngroups) return;
  const int lane = threadIdx.x;
  const uint32_t* off = blkoff + f.blk_base;
  const QTab& qt = lut[f.qidx];
  const int chroma = part == 2u;
  {
    const int nat = c_zz[lane];
    const int q = chroma ? qt.ciqt[nat] : qt.liqt[nat];
    s_tab[lane] = ((uint32_t)q << 8) | (uint32_t)((nat & 7) * 8 + (nat >> 3));
  }
  __syncthreads();  // one wave: orders the table write before the lanes' reads
  const uint32_t bt8 = (uint32_t)(chroma ? qt.cb8 : qt.lb8);
  const uint32_t* tab = s_tab;
  int16_t* my = s_coef + lane * kCoefStride;

  // ---- which block of a group is mine ----
  const uint32_t dmb = chroma ? (uint32_t)(lane & 31) : (uint32_t)(lane >> 1);
  const uint32_t kblk = chroma ? 4u + (uint32_t)(lane >> 5) : 2u * part + (uint32_t)(lane & 1);
  const uint8_t* data = stream + f.data_off;

  // the dword that holds stream byte `p` and the four after it, bytes at or past data_len read as 0
  struct Bytes {
    uint32_t d[5];
  };
  auto fetch = [&](uint32_t p) -> Bytes {
    const uint8_t* g = data + p;
    const uint32_t sh = (uint32_t)((uintptr_t)g & 3u);
    const uint32_t* g4 = (const uint32_t*)(g - sh);
    const long long rel = (long long)p - (long long)sh;  // position of g4[0]'s first byte
    Bytes b;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const long long rem = (long long)f.data_len - (rel + 4ll * k);
      uint32_t v = 0;
      if (rem > 0) {
        v = g4[k];
        if (rem < 4) v &= (1u << (8 * (int)rem)) - 1u;
      }
      b.d[k] = v;
    }
    return b;
  };

  uint32_t grp = slot;
  uint32_t mb = grp * (uint32_t)kMbPerGroup + dmb;
  bool valid = mb < f.nmb;
  uint32_t pos0 = valid ? off[6u * mb + kblk] : 0u;  // block start relative to the first data byte
  Bytes cur = fetch(pos0);

  for (int it = 0; it < kDecIters; it++) {
    // ---- request the next group's block offset before anything else ----
    const uint32_t grp_n = grp + slots;
    const bool have_n = it + 1 < kDecIters && grp_n < ngroups;  // wave-uniform
    const uint32_t mb_n = grp_n * (uint32_t)kMbPerGroup + dmb;
    const bool valid_n = have_n && mb_n < f.nmb;
    const uint32_t pos_n = valid_n ? off[6u * mb_n + kblk] : 0u;

    const uint32_t sh = (uint32_t)((uintptr_t)(data + pos0) & 3u);
    uint32_t d0 = cur.d[0], d1 = cur.d[1], d2 = cur.d[2], d3 = cur.d[3], d4 = cur.d[4];
    // unchanged block: previous pixels stay (lib/RTjpeg.c:2704)
    const bool live_blk = valid && (__builtin_amdgcn_alignbyte(d1, d0, sh) & 0xFFu) != 0xFFu;

    if (live_blk) {
      // ---- stream -> dequantised coefficients, int16, transposed (lib/RTjpeg.c:157-186) ----
      // The block's bytes are consumed 16 at a time from registers (aligned dwords + a byte funnel
      // shift), so the only loop-carried dependency is the coefficient counter; table reads and
      // scratch writes of the byte slots are independent and pipeline through the LDS.
      {
        uint4* z = (uint4*)my;
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = make_uint4(0, 0, 0, 0);
      }
      uint32_t co = 0;      // next coefficient slot (zig-zag index)
      uint32_t jbase = 0;   // index of the round's first byte within the block
      uint32_t pnext = pos0;
      while (true) {
        const uint32_t wd[4] = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                                __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh)};
        // 16 bytes are in registers; they are consumed eight at a time so that short blocks stop early
        // (here luma blocks are 13..24 bytes, chroma 2..5)
        bool more = true;
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if (more) {
#pragma unroll
            for (int t = 8 * half; t < 8 * half + 8; t++) {
              const uint32_t j = jbase + t;
              const uint32_t ub = (wd[t >> 2] >> (8 * (t & 3))) & 0xFFu;
              const int sv = (int)(int8_t)ub;
              const int val = j == 0u ? (int)ub : sv;                // DC is the only unsigned byte
              const bool run = j > bt8 && sv > 63;                   // zero run of sv-63 slots (scratch is already 0)
              const bool live = co < 64u;
              const uint32_t e = tab[co & 63u];
              const int prod = mul24(val, (int)(e >> 8));            // |val| < 2^8, dequantiser < 2^15
              my[(live && !run) ? (e & 63u) : 64u] = (int16_t)prod;  // slot 64 is a write-only dump
              co += run ? (uint32_t)(sv - 63) : 1u;
            }
            more = __any(co < 64u);
          }
        }
        if (!more) break;
        jbase += 16u;
        pnext += 16u;
        const Bytes nb = fetch(pnext);  // same alignment as pos0: nb.d[0] is the old d4
        d0 = nb.d[0];
        d1 = nb.d[1];
        d2 = nb.d[2];
        d3 = nb.d[3];
        d4 = nb.d[4];
      }
    }

    // ---- request the next group's stream bytes: they arrive while this group is transformed ----
    Bytes nxt;
    if (have_n) nxt = fetch(pos_n);

    if (live_blk) {
      // ---- column pass: column c of the block is the c-th 16-byte piece of the scratch ----
      int ws[8][8];
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const uint4 q = ((const uint4*)my)[c];
        int x0 = (int)(int16_t)(q.x & 0xFFFFu), x1 = (int)q.x >> 16;
        const int x2 = (int)(int16_t)(q.y & 0xFFFFu), x3 = (int)q.y >> 16;
        const int x4 = (int)(int16_t)(q.z & 0xFFFFu), x5 = (int)q.z >> 16;
        const int x6 = (int)(int16_t)(q.w & 0xFFFFu), x7 = (int)q.w >> 16;
        if (c == 0) x0 += 4;  // DESCALE's rounding term, carried through both linear DC paths
        int y[8];
        idct8(x0, x1, x2, x3, x4, x5, x6, x7, y);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[r][c] = y[r];
      }

      // ---- row pass + scatter ----
      // macroblock coordinates without a per-lane division: one scalar division for the group's
      // first macroblock, then at most one row wrap per lane when rows are at least a group wide
      uint32_t mx, my_;
      {
        const uint32_t mbw = f.mbw, mb0 = grp * (uint32_t)kMbPerGroup;
        const uint32_t gy = mb0 / mbw, gx = mb0 - gy * mbw;  // uniform
        if (mbw >= (uint32_t)kMbPerGroup) {
          const bool wrap = gx + dmb >= mbw;
          mx = wrap ? gx + dmb - mbw : gx + dmb;
          my_ = wrap ? gy + 1u : gy;
        } else {
          my_ = mb / mbw;
          mx = mb - my_ * mbw;
        }
      }
      uint8_t* dst;
      uint32_t stride;
      if (!chroma) {
        stride = f.w;
        dst = outbuf + f.out_off + (size_t)(16u * my_ + 8u * (kblk >> 1)) * stride + 16u * mx + 8u * (kblk & 1u);
      } else {
        stride = f.w >> 1;
        const size_t ysz = (size_t)f.w * f.h;
        dst = outbuf + f.out_off + ysz + (kblk == 5u ? ysz >> 2 : 0) + (size_t)(8u * my_) * stride + 8u * mx;
      }
#pragma unroll
      for (int r = 0; r < 8; r++) {
        int y[8];
        idct8(ws[r][0], ws[r][1], ws[r][2], ws[r][3], ws[r][4], ws[r][5], ws[r][6], ws[r][7], y);
        uint2 o;
        o.x = px(y[0]) | (px(y[1]) << 8) | (px(y[2]) << 16) | (px(y[3]) << 24);
        o.y = px(y[4]) | (px(y[5]) << 8) | (px(y[6]) << 16) | (px(y[7]) << 24);
        *(uint2*)dst = o;
        dst += stride;
      }
    }
    if (!have_n) break;
    grp = grp_n;
    mb = mb_n;
    valid = valid_n;
    pos0 = pos_n;
    cur = nxt;
  }
}

}  // namespace mirtj
