// dv_decode_kernels.h — gfx950 kernel of the DV25 525/60 video decoder (the arithmetic: DESIGN.md section 9; the
// format: IEC 61834-2 / SMPTE 314M; nothing in the reference to follow — lib/dvframe.c:663-676 passes the DIF frame on).
//
//   k_dv_decode    one wave per two video segments (2 x 5 compressed macroblocks = 60 blocks, one lane each):
//                  the three passes of the variable-length decode, reconstruction, both inverse transforms, placement.
//                  A workgroup is kDvWaves such waves that share the constant tables in LDS and nothing else: the only
//                  workgroup barrier is the one behind the table load; everything after it is ordered inside a wave.
//
// A video segment is the unit nothing crosses: its 30 blocks share their unused bits (pass 2 inside a macroblock,
// pass 3 across the segment), and its five macroblocks land in five different super blocks of the picture.  Integer
// only; no MFMA (variable-length decode and a rounded butterfly).
#pragma once
#include <hip/hip_runtime.h>

#include "dv_common.h"
#include "rtj_idct_pk.h"  // the packed passes of the RTjpeg path: the butterfly is the same

namespace midv {

#ifndef MIDV_LANE_STRIDE
#define MIDV_LANE_STRIDE 144
#endif
#ifndef MIDV_WAVES
#define MIDV_WAVES 1
#endif
// bytes of LDS scratch per lane: 64 int16 coefficients + padding (144: 16-byte reads without bank conflicts)
constexpr int kLaneStride = MIDV_LANE_STRIDE;
static_assert(kLaneStride >= 128 && kLaneStride % 16 == 0, "a lane's scratch: 64 int16, read 16 bytes at a time");
constexpr int kDvWaves = MIDV_WAVES;  // waves per workgroup (2.9 KB of tables per workgroup instead of per wave)
constexpr int kDvLive = 60;       // lanes of a wave that hold a block
constexpr int kDvPairs = kSegments / 2;                             // waves a frame needs
constexpr int kDvGridX = (kDvPairs + kDvWaves - 1) / kDvWaves;      // workgroups per frame
constexpr int kMbufWords = 20;    // a macroblock's free space: at most 6 x 100 bits, + a dword to read past
constexpr int kVbufWords = 86;    // a segment's: at most 2680 bits, + a dword to read past
#ifndef MIDV_SKIP  // timing builds only (wrong pictures): 1 no pass 2 / 3, 2 no transforms and stores, 4 no pass 1
#define MIDV_SKIP 0
#endif

// (x * c + 128) >> 8.  v_mad_i32_i24 is spelled out: every multiplicand of both passes stays below 2^23 in magnitude for
// any block of int16 coefficients (the butterfly is the RTjpeg path's, whose bound tests/test_bounds.py derives; the
// 2-4-8 column pass is its even half plus one addition), so the 24-bit multiplier is exact and wraps like the 32-bit
// product — and the compiler, which cannot see the bound, would use the quarter-rate 32-bit multiplier for the row pass.
__device__ __forceinline__ int dv_mul(int x, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(c), "v"(128));
  return r >> 8;
}
// the scaled 8-point butterfly (lib/RTjpeg.c:2240-2283 as SURVEY.md appendix A.4 states it) and its even half
__device__ __forceinline__ void dv_idct8(const int (&x)[8], int (&y)[8]) {
  const int t10 = x[0] + x[4], t11 = x[0] - x[4], t13 = x[2] + x[6], t12 = dv_mul(x[2] - x[6], 362) - t13;
  const int e0 = t10 + t13, e3 = t10 - t13, e1 = t11 + t12, e2 = t11 - t12;
  const int z13 = x[5] + x[3], z10 = x[5] - x[3], z11 = x[1] + x[7], z12 = x[1] - x[7];
  const int o7 = z11 + z13, m = dv_mul(z11 - z13, 362), z5 = dv_mul(z10 + z12, 473);
  const int t10o = dv_mul(z12, 277) - z5, t12o = dv_mul(z10, -669) + z5;
  const int o6 = t12o - o7, o5 = m - o6, o4 = t10o + o5;
  y[0] = e0 + o7; y[7] = e0 - o7; y[1] = e1 + o6; y[6] = e1 - o6;
  y[2] = e2 + o5; y[5] = e2 - o5; y[4] = e3 + o4; y[3] = e3 - o4;
}
__device__ __forceinline__ void dv_idct4(int x0, int x1, int x2, int x3, int (&a)[4]) {
  const int t10 = x0 + x2, t11 = x0 - x2, t13 = x1 + x3, t12 = dv_mul(x1 - x3, 362) - t13;
  a[0] = t10 + t13; a[3] = t10 - t13; a[1] = t11 + t12; a[2] = t11 - t12;
}

// 16 bits (in the top half of the result) of an LDS bit buffer from bit position bp on, MSB first
__device__ __forceinline__ uint32_t dv_peek(const uint32_t* buf, uint32_t bp) {
  const uint32_t i = bp >> 5, sh = bp & 31u;
  const uint32_t d0 = buf[i], d1 = buf[i + 1];
  return sh ? __builtin_amdgcn_alignbit(d0, d1, 32u - sh) : d0;
}
// OR `n` bits (the top n of v, n <= 32, the rest of v zero) into an LDS bit buffer at bit position bp
__device__ __forceinline__ void dv_or_bits(uint32_t* buf, uint32_t bp, uint32_t v) {
  const uint32_t i = bp >> 5, sh = bp & 31u;
  atomicOr(&buf[i], v >> sh);
  if (sh) atomicOr(&buf[i + 1], v << (32u - sh));
}

// ---- two int16 values to a register (rtj_idct_pk.h: exact while the range test there holds) ----
__device__ __forceinline__ uint32_t dv_pk_add(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t dv_pk_sub(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (x * 362 + 128) >> 8 in both halves, modulo 2^16
__device__ __forceinline__ uint32_t dv_pk_mul362(uint32_t x, const mirtj::IdctPK& K) {
  uint32_t pl, ph, r;
  asm("v_mad_i32_i16 %0, %3, %4, %5\n\t"
      "v_mad_i32_i16 %1, %3, %4, %5 op_sel:[1,0,0,0]\n\t"
      "v_perm_b32 %2, %1, %0, %6"
      : "=&v"(pl), "=&v"(ph), "=v"(r)
      : "v"(x), "s"(K.k362), "v"(K.c128), "s"(K.sel_m));
  return r;
}
// the 2-4-8 column pass on a column pair: x[r] = (row r of the even column | of the odd one << 16); rows 0, 2, 4, 6 are
// the 4-point transform of the fields' sum, rows 1, 3, 5, 7 of their difference (dv_idct4 twice, then sum / difference).
// Every linear form here is one of the 8-8 pass's even half or a sum of two of them over disjoint coefficients, so the
// range test's weights (rtj_idct_pk.h) cover it.
__device__ __forceinline__ void dv_col248_pk(uint32_t (&x)[8], const mirtj::IdctPK& K) {
  uint32_t a[4], b[4];
  auto idct4 = [&](uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t(&o)[4]) {
    const uint32_t t10 = dv_pk_add(p0, p2), t11 = dv_pk_sub(p0, p2), t13 = dv_pk_add(p1, p3);
    const uint32_t t12 = dv_pk_sub(dv_pk_mul362(dv_pk_sub(p1, p3), K), t13);
    o[0] = dv_pk_add(t10, t13);
    o[3] = dv_pk_sub(t10, t13);
    o[1] = dv_pk_add(t11, t12);
    o[2] = dv_pk_sub(t11, t12);
  };
  idct4(x[0], x[2], x[4], x[6], a);
  idct4(x[1], x[3], x[5], x[7], b);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    x[2 * i] = dv_pk_add(a[i], b[i]);
    x[2 * i + 1] = dv_pk_sub(a[i], b[i]);
  }
}

// LDS traffic of one wave is ordered by the hardware (a wave's LDS instructions execute in order); what its lanes need
// between a write and another lane's read is that the compiler keeps the order
__device__ __forceinline__ void dv_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64 * kDvWaves) void k_dv_decode(const uint8_t* __restrict__ frames, uint8_t* __restrict__ pics,
                                                   const Tables* __restrict__ T
#ifdef MIDV_DEBUG
                                                   , int16_t* __restrict__ dbg  // [frame][workgroup][lane][64 coefficients + 8 state words]
#endif
) {
  __shared__ __attribute__((aligned(16))) uint8_t s_coef_all[kDvWaves][kDvLive * kLaneStride];
  __shared__ uint32_t s_lut9[512], s_lut2[64], s_tab[128], s_sh[24];
  __shared__ uint32_t s_mbuf_all[kDvWaves][10][kMbufWords], s_vbuf_all[kDvWaves][2][kVbufWords];
  __shared__ uint32_t s_mpos_all[kDvWaves][10], s_mlen_all[kDvWaves][10], s_vpos_all[kDvWaves][2], s_vlen_all[kDvWaves][2],
      s_excl_all[kDvWaves][64];

  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 512; i += 64 * kDvWaves) s_lut9[i] = T->lut9[i];
  for (int i = tid; i < 64; i += 64 * kDvWaves) s_lut2[i] = T->lut2[i];
  for (int i = tid; i < 128; i += 64 * kDvWaves) s_tab[i] = (&T->tab[0][0])[i];
  if (tid < 24) s_sh[tid] = T->shift4[tid];
  uint8_t* const s_coef = s_coef_all[wv];
  uint32_t(*const s_mbuf)[kMbufWords] = s_mbuf_all[wv];
  uint32_t(*const s_vbuf)[kVbufWords] = s_vbuf_all[wv];
  uint32_t* const s_mpos = s_mpos_all[wv];
  uint32_t* const s_mlen = s_mlen_all[wv];
  uint32_t* const s_vpos = s_vpos_all[wv];
  uint32_t* const s_vlen = s_vlen_all[wv];
  uint32_t* const s_excl = s_excl_all[wv];
  for (int i = lane; i < 10 * kMbufWords; i += 64) (&s_mbuf[0][0])[i] = 0u;
  for (int i = lane; i < 2 * kVbufWords; i += 64) (&s_vbuf[0][0])[i] = 0u;
  __syncthreads();  // the tables are there (the workgroup's only barrier: every wave reaches it, before anything can end one)
  const uint32_t pair = (uint32_t)blockIdx.x * (uint32_t)kDvWaves + wv;  // which two segments of the frame
  if (pair >= (uint32_t)kDvPairs) return;

  // ---- which block this lane has ----
  const bool live = lane < kDvLive;
  const uint32_t seg = (uint32_t)lane / 30u, b30 = (uint32_t)lane - 30u * seg, mbi = b30 / 6u, j = b30 - 6u * mbi;
  const uint32_t mb10 = (uint32_t)lane / 6u;  // macroblock of the wave, 0..9 (10: the idle lanes)
  uint32_t S = 2u * pair + seg;                // video segment of the frame
  if (S >= (uint32_t)kSegments) S = kSegments - 1;  // (never: 270 is even)
  const uint32_t seq = S / 27u, slot = S - 27u * seq;
  const uint32_t v = 5u * slot + mbi;
  const uint8_t* mbp = frames + (size_t)blockIdx.y * kFrameBytes + (size_t)((seq * 150u + 7u + v + v / 15u) * 80u);
  const uint32_t ao = j < 4u ? 4u + 14u * j : 60u + 10u * (j - 4u);  // the block's area inside the compressed macroblock
  const uint32_t A = j < 4u ? 112u : 80u;                            // ... and its bits
  uint32_t W0, W1, W2, W3;  // the area's bits, MSB first, shifted left as they are consumed
  uint32_t qno;
  {
    const uint32_t* p4 = (const uint32_t*)(mbp + (ao & ~3u));
    const uint32_t d0 = p4[0], d1 = p4[1], d2 = p4[2], d3 = j == 5u ? 0u : p4[3];  // (area 5 ends with the DIF block)
    const uint32_t sh = ao & 3u;
    const uint32_t b0 = __builtin_amdgcn_alignbyte(d1, d0, sh), b1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                   b2 = __builtin_amdgcn_alignbyte(d3, d2, sh), b3 = __builtin_amdgcn_alignbyte(0u, d3, sh);
    W0 = __builtin_bswap32(b0); W1 = __builtin_bswap32(b1); W2 = __builtin_bswap32(b2); W3 = __builtin_bswap32(b3);
    qno = ((const uint32_t*)mbp)[0] >> 24 & 15u;  // byte 3: STA | QNO
  }
  const int dc = (int)W0 >> 23;
  const uint32_t mode = (W0 >> 22) & 1u, cls = (W0 >> 20) & 3u;
  // consume n bits (1 <= n <= 31)
  auto shift = [&](uint32_t n) {  // one funnel shift per register: ({hi, lo} >> (32 - n)) is (hi << n) | (lo >> (32 - n))
    const uint32_t r = 32u - n;
    W0 = __builtin_amdgcn_alignbit(W0, W1, r);
    W1 = __builtin_amdgcn_alignbit(W1, W2, r);
    W2 = __builtin_amdgcn_alignbit(W2, W3, r);
    W3 <<= n;
  };
  shift(12u);
  uint32_t p = 12u;
  uint8_t* const my = s_coef + (live ? lane : 0) * kLaneStride;  // (the idle lanes have no scratch and write none)
  if (live) {
    uint2* z = (uint2*)my;  // (pairs of these merge into 16-byte writes where the stride allows)
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = make_uint2(0, 0);
    *(int16_t*)my = (int16_t)(dc * 4 + 1024 + 4);  // level shift and DESCALE's rounding term ride on the DC (both passes are linear in it)
  }
  const uint32_t sh4 = s_sh[qno + (cls == 0u ? 6u : cls == 1u ? 3u : cls == 2u ? 0u : 1u)] + (cls == 3u ? 0x1111u : 0u);
  const uint32_t tab_m = 64u * mode;

  // a code word: its table entry from the next 16 bits (top of w).  lut9 holds 0 for 11111....: the longer words
  auto lookup = [&](uint32_t w) -> uint32_t {
    uint32_t e = s_lut9[w >> 23];
    if (e == 0u) {
      const uint32_t i7 = (w >> 20) & 127u;
      e = i7 < 64u ? s_lut2[i7]
          : i7 < 96u ? vlc_entry(13u, ((w >> 19) & 63u) + 1u, 0u)  // 1111110 rrrrrr: a run of zeros (and one more)
                     : vlc_entry(16u, 1u, (w >> 17) & 255u);       // 1111111 aaaaaaaa s
    }
    return e;
  };
  uint32_t pos = 0;  // scan position of the coefficient decoded last; > 63: the block is finished
  auto apply = [&](uint32_t e, uint32_t w) {  // the word is complete: move on, store the coefficient
    const uint32_t len = e & 31u;
    pos += (e >> 5) & 127u;
    if (pos > 63u) return;
    const uint32_t amp = (e >> 12) & 255u;
    const int level = (w >> (32u - len)) & 1u ? -(int)amp : (int)amp;
    const uint32_t t = s_tab[tab_m + pos];
    const uint32_t s = (sh4 >> (4u * ((t >> 8) & 3u))) & 15u;
    int prod;  // |level| <= 255, the multiplier below 2^22: the 24-bit multiplier is exact (and full rate)
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(prod) : "v"(level), "v"((int)((t >> 16) << s)), "v"(8192));
    *(int16_t*)(my + (t & 255u)) = (int16_t)(prod >> 14);
  };

  // ---- pass 1: every block from its own area ----
  // (every loop of this kernel carries a bound no stream can reach — a code word is at least three bits long — so that a
  // wave always comes to its end whatever the tables or the bytes hold)
  bool act = live;
  for (int guard = 0; guard < ((MIDV_SKIP & 4) ? 0 : 48) && __any(act); guard++) {
    if (act) {
      const uint32_t e = lookup(W0), len = e & 31u;
      if (p + len > A) {
        act = false;  // the word does not end inside the area: its A - p bits stay in W0
      } else {
        apply(e, W0);
        shift(len);
        p += len;
        act = pos <= 63u;
      }
    }
  }
  bool fin = pos > 63u;
  uint32_t part = 0, npart = 0;  // an unfinished block's cut-off word: npart bits at the top of part
  uint32_t rem = A - p;          // bits left in the area: free space of a finished block, the cut-off word of another
  if (!live) rem = 0u;
  if (!fin) {
    npart = rem;
    part = rem ? W0 & ~(0xFFFFFFFFu >> rem) : 0u;
    rem = 0u;
  }
  // ---- the macroblock's free space: what its finished blocks left, in block order ----
  {
    uint32_t incl = rem;  // inclusive scan over the wave (60 values below 101: sums below 2^16)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl += o;
    }
    s_excl[lane] = incl - rem;
    dv_wave_sync();
    const uint32_t base = s_excl[6u * mb10 < 64u ? 6u * mb10 : 63u];
    const uint32_t at = incl - rem - base;  // where this block's bits go in its macroblock's buffer
    if (live && j == 5u) {
      s_mlen[mb10] = at + rem;
      s_mpos[mb10] = 0u;
    }
    if (rem) {
      uint32_t* mb = s_mbuf[mb10];
      uint32_t w[4] = {W0, W1, W2, W3};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (rem > 32u * i) {
          const uint32_t n = rem - 32u * i;
          dv_or_bits(mb, at + 32u * i, n >= 32u ? w[i] : w[i] & ~(0xFFFFFFFFu >> n));
        }
      }
    }
  }
  dv_wave_sync();
  // lanes that want bits take turns, lowest lane of a group first, reading from the group's buffer until their block
  // is finished or the buffer is used up
  auto drain = [&](const unsigned long long gmask, const uint32_t* buf, uint32_t* gpos, const uint32_t glen) {
    for (int turn = 0; turn < ((MIDV_SKIP & 1) ? 0 : 64); turn++) {
      const bool want = live && !fin && *gpos < glen;
      const unsigned long long m = __ballot(want);
      if (m == 0ull) break;
      const unsigned long long mine = m & gmask;
      if (want && (uint32_t)lane == (uint32_t)__builtin_ctzll(mine)) {
        uint32_t bp = *gpos;
        for (int guard = 0; guard < 1024; guard++) {
          const uint32_t avail = glen - bp;
          uint32_t w = dv_peek(buf, bp);
          if (npart) w = part | (w >> npart);
          const uint32_t e = lookup(w), len = e & 31u;
          if (len > npart + avail) {  // cut off again: keep what there is
            const uint32_t have = npart + avail;
            part = have ? w & ~(0xFFFFFFFFu >> have) : 0u;
            npart = have;
            bp = glen;
            break;
          }
          bp += len - npart;
          npart = 0u;
          part = 0u;
          apply(e, w);
          if (pos > 63u) {
            fin = true;
            break;
          }
        }
        *gpos = bp;
      }
      dv_wave_sync();
    }
  };
  // ---- pass 2: inside the macroblock ----
  drain(0x3Full << (6u * mb10 < 60u ? 6u * mb10 : 60u), s_mbuf[mb10 < 10u ? mb10 : 9u], &s_mpos[mb10 < 10u ? mb10 : 9u],
        live ? s_mlen[mb10] : 0u);
  // ---- the segment's free space: what its macroblocks left (those whose blocks are all finished) ----
  {
    const unsigned long long unf = __ballot(live && !fin);
    const bool allfin = (unf & (0x3Full << (6u * (mb10 < 10u ? mb10 : 0u)))) == 0ull;
    const uint32_t left = live && j == 0u && allfin ? s_mlen[mb10] - s_mpos[mb10] : 0u;
    s_excl[lane] = left;
    dv_wave_sync();
    if (live && j == 0u) {
      uint32_t at = 0;
      for (uint32_t k = 0; k < mbi; k++) at += s_excl[30u * seg + 6u * k];
      if (mbi == 4u) {
        s_vlen[seg] = at + left;
        s_vpos[seg] = 0u;
      }
      const uint32_t* mb = s_mbuf[mb10];
      uint32_t from = s_mpos[mb10];
      for (uint32_t done = 0; done < left; done += 32u) {
        const uint32_t n = left - done;
        uint32_t w = dv_peek(mb, from + done);
        if (n < 32u) w &= ~(0xFFFFFFFFu >> n);
        dv_or_bits(s_vbuf[seg], at + done, w);
      }
    }
  }
  dv_wave_sync();
  // ---- pass 3: across the segment ----
  drain(seg ? 0x3FFFFFFFull << 30 : 0x3FFFFFFFull, s_vbuf[seg < 2u ? seg : 1u], &s_vpos[seg < 2u ? seg : 1u],
        live ? s_vlen[seg] : 0u);

#ifdef MIDV_DEBUG
  if (dbg) {
    dv_wave_sync();
    int16_t* o = dbg + (((size_t)blockIdx.y * kDvPairs + pair) * 64 + lane) * 72;
    for (int i = 0; i < 64; i++)  // natural order out of the scratch's (column pair, row) layout
      o[i] = ((const int16_t*)my)[2 * (8 * ((i & 7) >> 1) + (i >> 3)) + (i & 1)];
    o[64] = (int16_t)pos; o[65] = (int16_t)p; o[66] = (int16_t)fin; o[67] = (int16_t)npart; o[68] = (int16_t)mode; o[69] = (int16_t)cls;
    o[70] = (int16_t)qno; o[71] = (int16_t)(live ? (mb10 < 10u ? s_mlen[mb10] : 0) : 0);
  }
#endif
  // ---- inverse transform and placement ----
  // The scratch holds the block as the RTjpeg path's does (dv_tables.cpp): dword (column pair j, row r) at 8 j + r.  A wave
  // whose blocks all pass the 16-bit range test takes the packed passes (two values to a register: rtj_idct_pk.h, and
  // dv_col248_pk for 2-4-8 blocks); any other wave the 32-bit ones below.  Both are the statement's arithmetic exactly.
  mirtj::IdctPK K = mirtj::idct_pk_constants();
  K.c235 = 0x00FF00FFu;  // pixels clamp to 0..255 here
  uint4 q[8];
  {
    const uint4* qq = (const uint4*)my;
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = qq[i];
  }
  const bool pk = __all(!live || mirtj::pk_range_full(q, K));
  if (!live || (MIDV_SKIP & 2)) return;
  // where the block goes (525/60 4:1:1 macroblock shuffling and placement)
  uint32_t x32, y8;
  {
    const uint32_t off = mbi == 0u ? 2u : mbi == 1u ? 6u : mbi == 2u ? 8u : mbi == 3u ? 0u : 4u;
    const uint32_t start = mbi == 0u ? 9u : mbi == 1u ? 4u : mbi == 2u ? 13u : mbi == 3u ? 0u : 18u;
    const uint32_t i = (seq + off) % 10u;
    const uint32_t k = slot + (mbi == 1u || mbi == 2u ? 3u : 0u);
    const uint32_t k6 = k / 6u, km = k - 6u * k6;
    const uint32_t serp = k6 & 1u ? 5u - km : km;
    x32 = start + k6;
    y8 = x32 > 21u ? 2u * serp + 6u * i : serp + 6u * i;
  }
  uint8_t* pic = pics + (size_t)blockIdx.y * kPicBytes;
  typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(4)));
  const bool edge = x32 == 22u;
  uint32_t stride, org;
  if (j < 4u) {
    stride = kW;
    org = edge ? (8u * y8 + 8u * (j >> 1)) * kW + 32u * x32 + 8u * (j & 1u) : 8u * y8 * kW + 32u * x32 + 8u * j;
  } else {
    stride = kCW;
    org = kW * kH + (j == 4u ? kCW * kH : 0u) + 8u * y8 * kCW + 8u * x32;  // block 4 is Cr (third plane), block 5 Cb
  }
  const bool halves = j >= 4u && edge;  // the right-edge chroma block: left half here, right half eight lines below
  auto put = [&](int r, uint32_t lo, uint32_t hi) {
    if (halves) {
      *(uint32_t*)(pic + org + (uint32_t)r * stride) = lo;
      *(uint32_t*)(pic + org + (uint32_t)(r + 8) * stride) = hi;
    } else {
      u32x2a o;
      o.x = lo;
      o.y = hi;
      *(u32x2a*)(pic + org + (uint32_t)r * stride) = o;
    }
  };
  if (pk) {
    uint32_t y[4][8];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      uint32_t x[8] = {q[2 * jj].x, q[2 * jj].y, q[2 * jj].z, q[2 * jj].w, q[2 * jj + 1].x, q[2 * jj + 1].y, q[2 * jj + 1].z, q[2 * jj + 1].w};
      if (mode == 0u)
        mirtj::idct8_pk_col<false>(x, K);
      else
        dv_col248_pk(x, K);
#pragma unroll
      for (int r = 0; r < 8; r++) y[jj][r] = x[r];
    }
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      uint32_t ya[4] = {y[0][r], y[1][r], y[2][r], y[3][r]}, yb[4] = {y[0][r + 1], y[1][r + 1], y[2][r + 1], y[3][r + 1]};
      uint2 a, b;
      mirtj::idct8_pk_row_px<0>(ya, yb, a, b, K);
      put(r, a.x, a.y);
      put(r + 1, b.x, b.y);
    }
    return;
  }
  int c[64];
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const uint4& t = q[2 * jj + (r >> 2)];
      const uint32_t d = (r & 3) == 0 ? t.x : (r & 3) == 1 ? t.y : (r & 3) == 2 ? t.z : t.w;
      c[8 * r + 2 * jj] = (int)(int16_t)(d & 0xFFFFu);
      c[8 * r + 2 * jj + 1] = (int)d >> 16;
    }
  }
  int ws[64];
  if (mode == 0u) {
#pragma unroll
    for (int h = 0; h < 8; h++) {
      int x[8], y[8];
#pragma unroll
      for (int r = 0; r < 8; r++) x[r] = c[8 * r + h];
      dv_idct8(x, y);
#pragma unroll
      for (int r = 0; r < 8; r++) ws[8 * r + h] = y[r];
    }
  } else {  // 2-4-8: rows 2v / 2v + 1 hold the sum / the difference of the two fields
#pragma unroll
    for (int h = 0; h < 8; h++) {
      int a[4], b[4];
      dv_idct4(c[h], c[16 + h], c[32 + h], c[48 + h], a);
      dv_idct4(c[8 + h], c[24 + h], c[40 + h], c[56 + h], b);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ws[8 * (2 * i) + h] = a[i] + b[i];
        ws[8 * (2 * i + 1) + h] = a[i] - b[i];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int y[8];
    const int(&xr)[8] = *(const int(*)[8])(ws + 8 * r);
    dv_idct8(xr, y);
    uint32_t px[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int s = (int)((uint32_t)y[k] << 13) >> 16;  // (int16)(v >> 3): DESCALE with int16 narrowing (the + 4 came in with DC)
      // (the empty asm keeps hipcc (ROCm 7.2) from fusing "shift, clamp, pack" into gfx950's v_ashr_pk_u8_i32, which came
      // out wrong on the MI355X here as it did in the colour stage, rtj_color_kernels.h: third byte of every dword)
      asm volatile("" : "+v"(s));
      s = s < 0 ? 0 : s > 255 ? 255 : s;
      px[k] = (uint32_t)s;
    }
    put(r, px[0] | px[1] << 8 | px[2] << 16 | px[3] << 24, px[4] | px[5] << 8 | px[6] << 16 | px[7] << 24);
  }
}

}  // namespace midv
