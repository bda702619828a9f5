/*
 * dv_oracle.c — see dv_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no DV pixel decoder exists in the
 * reference tree: lib/dvframe.c:663-676 is a memcpy, the pixels are libavcodec's, lib/video_ffmpeg.c:1572-1575).
 *
 * Sources, all from memory of the published format (IEC 61834-2 / SMPTE 314M), section by section:
 *   DIF sequence layout        150 blocks of 80 bytes: header, 2 subcode, 3 VAUX, then 9 x (1 audio + 15 video)
 *   compressed macroblock      3 id bytes, STA | QNO, Y0..Y3 of 14 bytes, Cr and Cb of 10 bytes
 *   block header               DC (9 bits, signed), transform mode (1), class (2), then the AC code words
 *   video segment              5 consecutive compressed macroblocks; AC words that do not fit their block's area
 *                              continue in the unused space of the macroblock, then of the segment (three passes)
 *   macroblock shuffling       525/60 4:1:1: super blocks of 27 macroblocks, 5 columns x 10 rows; segment k of sequence
 *                              i takes macroblock k of the super blocks (i+2, 2), (i+6, 1), (i+8, 3), (i+0, 0), (i+4, 4)
 *   variable-length code       run / amplitude pairs, 2..16 bits, prefix 1111110 = run escape, 1111111 = amplitude escape
 *   quantisation               step = 2^shift by class, quantisation number and area of the scan position
 *   weighting                  w(0)=1, w(1)=CS4/(4 CS7 CS2), w(2)=CS4/(2 CS6), w(3)=1/(2 CS5), w(4)=7/8, w(5)=CS4/CS3,
 *                              w(6)=CS4/CS2, w(7)=CS4/CS1 with CSm = cos(m pi/16); W(h,v) = w(h) w(v) / 2 (8-8) or
 *                              w(h) w(2v) / 2 (2-4-8); DC 1/4
 * The fixed-point arithmetic (reconstruction multipliers with 14 fractional bits, the scaled 8-point butterfly of
 * lib/RTjpeg.c:2209-2332 with constants 362 / 473 / 669 / 277 over 256 for both transform sizes, int16 coefficients)
 * is this repository's own choice — see the header.
 */
#include "dv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- the variable-length code: (run, amplitude) in code order, lengths without the sign bit; 255 = end of block ---- */
static const uint8_t vlc_len[] = {
    2, 3, 4,   4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8,  8, 8, 8, 8, 8, 8, 8,
    8, 8, 8,   8, 8, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11,
    11, 11, 11, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12};
static const uint8_t vlc_run[] = {
    0, 0, 255, 1, 0, 0, 2, 1, 0, 0, 3, 4, 0, 0, 5, 6, 2, 1, 1, 0, 0, 0, 7, 8, 9, 10, 3, 4, 2, 1, 1, 1, 0,
    0, 0, 0,   0, 0, 11, 12, 13, 14, 5, 6, 3, 4, 2, 2, 1, 0, 0, 0, 0, 0, 5, 3, 3, 2, 1, 1, 1, 0, 1, 6, 4, 3,
    1, 1, 1,   2, 3, 4, 5, 7, 8, 9, 10, 7, 8, 4, 3, 2, 2, 2, 2, 2, 1, 1, 1};
static const uint8_t vlc_amp[] = {
    1,  2,  0,  1,  3,  4,  1,  2,  5,  6,  1,  1,  7,  8,  1,  1,  2,  3,  4,  9, 10, 11, 1, 1, 1, 1, 2, 2, 3, 5, 6, 7, 12,
    13, 14, 15, 16, 17, 1,  1,  1,  1,  2,  2,  3,  3,  4,  5,  8,  18, 19, 20, 21, 22, 3, 4, 5, 6, 9, 10, 11, 0, 0, 3, 4, 6,
    12, 13, 14, 0,  0,  0,  0,  2,  2,  2,  2,  3,  3,  5,  7,  7,  8,  9,  10, 11, 15, 16, 17};
enum { NSHORT = (int)sizeof vlc_len };

typedef struct {
  uint8_t len;  /* total bits, sign included */
  uint8_t run;  /* zero coefficients before this one; 255: end of block */
  int16_t level;
} vlc_ent;
static vlc_ent lut16[65536];
/* encoder side: code word and length (sign excluded) of (run, amp), 0 length = no direct word */
static uint16_t enc_code[64][256];
static uint8_t enc_len[64][256];
static uint16_t eob_code;
static uint8_t eob_len;

static const uint8_t zz88[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
/* 2-4-8: rows 2v and 2v + 1 of the array hold vertical frequency v of the sum and of the difference of the two fields */
static const uint8_t zz248[64] = {0,  8,  1,  9,  16, 24, 2,  10, 17, 25, 32, 40, 48, 56, 33, 41, 18, 26, 3,  11, 4,  12,
                                  19, 27, 34, 42, 49, 57, 50, 58, 35, 43, 20, 28, 5,  13, 6,  14, 21, 29, 36, 44, 51, 59,
                                  52, 60, 37, 45, 22, 30, 7,  15, 23, 31, 38, 46, 53, 61, 54, 62, 39, 47, 55, 63};
static const uint8_t quant_shifts[22][4] = {{3, 3, 4, 4}, {3, 3, 4, 4}, {2, 3, 3, 4}, {2, 3, 3, 4}, {2, 2, 3, 3}, {2, 2, 3, 3},
                                            {1, 2, 2, 3}, {1, 2, 2, 3}, {1, 1, 2, 2}, {1, 1, 2, 2}, {0, 1, 1, 2}, {0, 1, 1, 2},
                                            {0, 0, 1, 1}, {0, 0, 1, 1}, {0, 0, 0, 1}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
                                            {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
static const uint8_t quant_offset[4] = {6, 3, 0, 1};
static int area_of(int k) { return k < 6 ? 0 : k < 21 ? 1 : k < 43 ? 2 : 3; }

static int32_t qbase[2][64];
static double fwd[2][64][64]; /* encoder: pixels (minus 128) -> transform inputs, the exact inverse of the decoder's linear part */
static int ready;

static void init_vlc(void) {
  /* canonical code words from the lengths; the lengths must form a complete code */
  uint32_t code = 0;
  int prev = vlc_len[0];
  double kraft = 0;
  for (int r = 0; r < 64; r++)
    for (int a = 0; a < 256; a++) enc_len[r][a] = 0;
  for (int i = 0; i < NSHORT + 64 + 256; i++) {
    int len, run, amp;
    if (i < NSHORT) len = vlc_len[i], run = vlc_run[i], amp = vlc_amp[i];
    else if (i < NSHORT + 64) len = 13, run = i - NSHORT, amp = 0;
    else len = 15, run = 0, amp = i - NSHORT - 64;
    code <<= (len - prev);
    prev = len;
    kraft += ldexp(1.0, -len);
    const int has_sign = run != 255 && (amp != 0 || len == 15); /* the amplitude escape always carries its sign bit */
    const int total = len + has_sign;
    /* every 16-bit window that starts with this word */
    for (uint32_t s = 0; s < (has_sign ? 2u : 1u); s++) {
      const uint32_t w = ((code << has_sign) | s) << (16 - total);
      for (uint32_t rest = 0; rest < (1u << (16 - total)); rest++) {
        vlc_ent *e = &lut16[w | rest];
        e->len = (uint8_t)total;
        e->run = (uint8_t)run;
        e->level = (int16_t)(s ? -amp : amp);
      }
    }
    if (run == 255) eob_code = (uint16_t)code, eob_len = (uint8_t)len;
    else if (enc_len[run][amp] == 0) enc_code[run][amp] = (uint16_t)code, enc_len[run][amp] = (uint8_t)len; /* shortest word first */
    code++;
  }
  if (kraft != 1.0 || code != (1u << 15)) abort(); /* not a complete prefix code: the table above is wrong */
}

static int32_t MUL(int32_t x, int32_t c) { return (x * c + 128) >> 8; }
/* the scaled 8-point butterfly (lib/RTjpeg.c:2240-2283) */
static void idct8(const int32_t x[8], int32_t y[8]) {
  const int32_t t10 = x[0] + x[4], t11 = x[0] - x[4], t13 = x[2] + x[6], t12 = MUL(x[2] - x[6], 362) - t13;
  const int32_t e0 = t10 + t13, e3 = t10 - t13, e1 = t11 + t12, e2 = t11 - t12;
  const int32_t z13 = x[5] + x[3], z10 = x[5] - x[3], z11 = x[1] + x[7], z12 = x[1] - x[7];
  const int32_t o7 = z11 + z13, m = MUL(z11 - z13, 362), z5 = MUL(z10 + z12, 473);
  const int32_t t10o = MUL(z12, 277) - z5, t12o = MUL(z10, -669) + z5;
  const int32_t o6 = t12o - o7, o5 = m - o6, o4 = t10o + o5;
  y[0] = e0 + o7; y[7] = e0 - o7; y[1] = e1 + o6; y[6] = e1 - o6;
  y[2] = e2 + o5; y[5] = e2 - o5; y[4] = e3 + o4; y[3] = e3 - o4;
}
/* its even half alone is a 4-point transform: inputs at the places of x0, x2, x4, x6 */
static void idct4(const int32_t x[4], int32_t a[4]) {
  const int32_t t10 = x[0] + x[2], t11 = x[0] - x[2], t13 = x[1] + x[3], t12 = MUL(x[1] - x[3], 362) - t13;
  a[0] = t10 + t13; a[3] = t10 - t13; a[1] = t11 + t12; a[2] = t11 - t12;
}
/* the same two in real arithmetic, for the encoder's forward transform */
static void idct8d(const double x[8], double y[8]) {
  const double c362 = 362 / 256.0, c473 = 473 / 256.0, c669 = 669 / 256.0, c277 = 277 / 256.0;
  const double t10 = x[0] + x[4], t11 = x[0] - x[4], t13 = x[2] + x[6], t12 = (x[2] - x[6]) * c362 - t13;
  const double e0 = t10 + t13, e3 = t10 - t13, e1 = t11 + t12, e2 = t11 - t12;
  const double z13 = x[5] + x[3], z10 = x[5] - x[3], z11 = x[1] + x[7], z12 = x[1] - x[7];
  const double o7 = z11 + z13, m = (z11 - z13) * c362, z5 = (z10 + z12) * c473;
  const double t10o = z12 * c277 - z5, t12o = -z10 * c669 + z5;
  const double o6 = t12o - o7, o5 = m - o6, o4 = t10o + o5;
  y[0] = e0 + o7; y[7] = e0 - o7; y[1] = e1 + o6; y[6] = e1 - o6;
  y[2] = e2 + o5; y[5] = e2 - o5; y[4] = e3 + o4; y[3] = e3 - o4;
}
static void idct4d(const double x[4], double a[4]) {
  const double t10 = x[0] + x[2], t11 = x[0] - x[2], t13 = x[1] + x[3], t12 = (x[1] - x[3]) * (362 / 256.0) - t13;
  a[0] = t10 + t13; a[3] = t10 - t13; a[1] = t11 + t12; a[2] = t11 - t12;
}

/* coefficients (natural order, row = vertical frequency) -> 64 values before descale; T = int32_t or double */
#define DEFINE_BLOCK_TRANSFORM(NAME, T, I8, I4)                 \
  static void NAME(int mode, const T *c, T *out) {              \
    T ws[64];                                                   \
    for (int h = 0; h < 8; h++) {                               \
      if (!mode) {                                              \
        T x[8], y[8];                                           \
        for (int r = 0; r < 8; r++) x[r] = c[8 * r + h];        \
        I8(x, y);                                               \
        for (int r = 0; r < 8; r++) ws[8 * r + h] = y[r];       \
      } else {                                                  \
        T s[4], d[4], a[4], b[4];                               \
        for (int v = 0; v < 4; v++) {                           \
          s[v] = c[8 * (2 * v) + h];                            \
          d[v] = c[8 * (2 * v + 1) + h];                        \
        }                                                       \
        I4(s, a);                                               \
        I4(d, b);                                               \
        for (int i = 0; i < 4; i++) {                           \
          ws[8 * (2 * i) + h] = a[i] + b[i];                    \
          ws[8 * (2 * i + 1) + h] = a[i] - b[i];                \
        }                                                       \
      }                                                         \
    }                                                           \
    for (int r = 0; r < 8; r++) I8(ws + 8 * r, out + 8 * r);    \
  }
DEFINE_BLOCK_TRANSFORM(block_transform, int32_t, idct8, idct4)
DEFINE_BLOCK_TRANSFORM(block_transform_d, double, idct8d, idct4d)

static void invert64(double a[64][64], double inv[64][64]) {
  static double m[64][128];
  for (int i = 0; i < 64; i++)
    for (int j = 0; j < 64; j++) m[i][j] = a[i][j], m[i][64 + j] = i == j;
  for (int c = 0; c < 64; c++) {
    int p = c;
    for (int r = c + 1; r < 64; r++)
      if (fabs(m[r][c]) > fabs(m[p][c])) p = r;
    if (fabs(m[p][c]) < 1e-9) abort();
    for (int j = 0; j < 128; j++) {
      const double t = m[c][j];
      m[c][j] = m[p][j];
      m[p][j] = t;
    }
    const double d = m[c][c];
    for (int j = 0; j < 128; j++) m[c][j] /= d;
    for (int r = 0; r < 64; r++)
      if (r != c && m[r][c] != 0) {
        const double f = m[r][c];
        for (int j = 0; j < 128; j++) m[r][j] -= f * m[c][j];
      }
  }
  for (int i = 0; i < 64; i++)
    for (int j = 0; j < 64; j++) inv[i][j] = m[i][64 + j];
}

static void init_once(void) {
  if (ready) return;
  init_vlc();
  {
    uint8_t seen[64];
    for (int t = 0; t < 2; t++) {
      memset(seen, 0, sizeof seen);
      for (int k = 0; k < 64; k++) seen[(t ? zz248 : zz88)[k]]++;
      for (int k = 0; k < 64; k++)
        if (seen[k] != 1) abort(); /* a scan order must be a permutation */
    }
  }
  const double pi = 3.14159265358979323846;
  double cs[8], w[8], aan[8];
  for (int m = 0; m < 8; m++) cs[m] = cos(m * pi / 16);
  w[0] = 1; w[1] = cs[4] / (4 * cs[7] * cs[2]); w[2] = cs[4] / (2 * cs[6]); w[3] = 1 / (2 * cs[5]);
  w[4] = 7.0 / 8; w[5] = cs[4] / cs[3]; w[6] = cs[4] / cs[2]; w[7] = cs[4] / cs[1];
  aan[0] = 1;
  for (int m = 1; m < 8; m++) aan[m] = cs[m] * sqrt(2.0);
  for (int mode = 0; mode < 2; mode++)
    for (int k = 0; k < 64; k++) {
      const int nat = (mode ? zz248 : zz88)[k], r = nat >> 3, h = nat & 7;
      const int v = mode ? 2 * (r >> 1) : r; /* the 4-point transform is the even half of the 8-point one */
      qbase[mode][k] = (int32_t)floor(16384.0 * aan[h] * aan[v] / (w[h] * w[v]) + 0.5);
    }
  /* the encoder's forward transforms: invert the decoder's linear map (pixel = transform(input) / 8) */
  static double a[64][64];
  for (int mode = 0; mode < 2; mode++) {
    for (int k = 0; k < 64; k++) {
      double c[64] = {0}, out[64];
      c[k] = 1;
      block_transform_d(mode, c, out);
      for (int p = 0; p < 64; p++) a[p][k] = out[p] / 8;
    }
    invert64(a, fwd[mode]);
  }
  ready = 1;
}

int dvo_shift(int qno, int cls, int area) { return quant_shifts[qno + quant_offset[cls]][area] + 1 + (cls == 3); }
void dvo_qbase(int mode, int32_t out[64]) {
  init_once();
  memcpy(out, qbase[mode], sizeof qbase[mode]);
}
void dvo_scan(int mode, uint8_t out[64]) { memcpy(out, mode ? zz248 : zz88, 64); }
int dvo_vlc_lookup(uint32_t bits16, int *len, int *run, int *level) {
  init_once();
  const vlc_ent *e = &lut16[bits16 & 0xFFFF];
  *len = e->len;
  *run = e->run;
  *level = e->level;
  return e->run == 255;
}

static int16_t recon(int level, int mode, int k, int shift) {
  return (int16_t)(((int32_t)level * (qbase[mode][k] << shift) + 8192) >> 14);
}
static uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static void coef_pixels(int mode, const int16_t coef[64], uint8_t px[64]) {
  int32_t c[64], out[64];
  for (int i = 0; i < 64; i++) c[i] = coef[i];
  block_transform(mode, c, out);
  for (int i = 0; i < 64; i++) px[i] = clamp255((int16_t)((out[i] + 4) >> 3)); /* DESCALE with int16 narrowing, then 0..255 */
}
void dvo_block_pixels(int dc, int mode, int cls, int qno, const int16_t levels[64], uint8_t px[64]) {
  init_once();
  int16_t coef[64] = {0};
  coef[0] = (int16_t)(dc * 4 + 1024);
  for (int k = 1; k < 64; k++)
    if (levels[k]) coef[(mode ? zz248 : zz88)[k]] = recon(levels[k], mode, k, dvo_shift(qno, cls, area_of(k)));
  coef_pixels(mode, coef, px);
}

void dvo_mb_place(int seq, int slot, int m, int *x, int *y) {
  static const uint8_t off[5] = {2, 6, 8, 0, 4}, start[5] = {9, 4, 13, 0, 18};
  const int i = (seq + off[m]) % 10;
  const int k = slot + (m == 1 || m == 2 ? 3 : 0); /* these two super-block columns begin in the middle of a column */
  const int serp = (k / 6) & 1 ? 5 - k % 6 : k % 6;
  *x = start[m] + k / 6;
  *y = *x > 21 ? 2 * serp + 6 * i : serp + 6 * i; /* column 22: 16 x 16 macroblocks, three per super block */
}

/* where the 64 pixels of block j of a macroblock live: base offset in the picture and row stride; the right-edge
 * chroma blocks are split (left half rows 0-7, right half below), reported through `split` */
static size_t block_origin(int x, int y, int j, int *stride, int *split) {
  *split = 0;
  if (j < 4) {
    *stride = DVO_W;
    if (x < 22) return (size_t)(8 * y) * DVO_W + 32 * x + 8 * j;
    return (size_t)(8 * y + 8 * (j >> 1)) * DVO_W + 32 * x + 8 * (j & 1);
  }
  *stride = DVO_CW;
  *split = x == 22;
  /* block 4 is Cr (third plane), block 5 Cb (second plane) */
  return (size_t)DVO_W * DVO_H + (j == 4 ? (size_t)DVO_CW * DVO_H : 0) + (size_t)(8 * y) * DVO_CW + 8 * x;
}

static size_t dif_block(int seq, int v) { return ((size_t)seq * 150 + 7 + v + v / 15) * 80; } /* video block v = 0..134 */
static const int area_off[6] = {4, 18, 32, 46, 60, 70}; /* byte offsets of the six block areas in a compressed macroblock */
static int area_bits(int j) { return j < 4 ? 112 : 80; }

/* ---------------- decoder ---------------- */
typedef struct {
  int pos;       /* scan position of the coefficient decoded last; 64: finished */
  uint32_t part; /* bits of a code word cut off by the end of the space it was read from */
  int npart;
  int mode, cls, qno;
  int16_t coef[64];
} blk_t;
typedef struct {
  uint8_t bit[2800];
  int n, rd;
} bitbuf;
static void bb_put(bitbuf *b, const uint8_t *bytes, int from, int to) {
  for (int i = from; i < to; i++) b->bit[b->n++] = (bytes[i >> 3] >> (7 - (i & 7))) & 1;
}
/* statistics for the design notes (not thread safe): code words completed in pass 1 / 2 / 3, blocks entering pass 2 / 3 */
static long stat_words[3], stat_blocks[3];
static int stat_pass;
void dvo_pass_stats(long words[3], long blocks[3], int reset) {
  for (int i = 0; i < 3; i++) {
    words[i] = stat_words[i];
    blocks[i] = stat_blocks[i];
    if (reset) stat_words[i] = stat_blocks[i] = 0;
  }
}
/* the block reads code words from b until it is finished or b is used up (a cut-off word stays with the block) */
static void decode_ac(blk_t *k, bitbuf *b) {
  stat_blocks[stat_pass]++;
  while (k->pos < 64) {
    const int avail = k->npart + (b->n - b->rd);
    uint32_t w = k->part; /* k->npart bits */
    int have = k->npart;
    for (int i = b->rd; i < b->n && have < 16; i++) w = (w << 1) | b->bit[i], have++;
    const vlc_ent *e = &lut16[(w << (16 - have)) & 0xFFFF];
    if (e->len > avail) { /* the word does not end inside this space: keep what there is of it */
      k->part = w;
      k->npart = have; /* == avail: fewer than 16 bits */
      b->rd = b->n;
      return;
    }
    b->rd += e->len - k->npart;
    k->part = 0;
    k->npart = 0;
    stat_words[stat_pass]++;
    if (e->run == 255) {
      k->pos = 64;
      return;
    }
    k->pos += e->run + 1;
    if (k->pos > 63) { /* a run past the last coefficient ends the block */
      k->pos = 64;
      return;
    }
    k->coef[(k->mode ? zz248 : zz88)[k->pos]] = recon(e->level, k->mode, k->pos, dvo_shift(k->qno, k->cls, area_of(k->pos)));
  }
}

static void decode_segment(const uint8_t *dif, int seq, int slot, uint8_t *pic, int16_t (*coefs)[64]) {
  blk_t blk[5][6];
  bitbuf mbuf[5], vbuf;
  vbuf.n = vbuf.rd = 0;
  for (int m = 0; m < 5; m++) {
    const uint8_t *mb = dif + dif_block(seq, 5 * slot + m);
    const int qno = mb[3] & 15;
    mbuf[m].n = mbuf[m].rd = 0;
    for (int j = 0; j < 6; j++) { /* pass 1: every block from its own area */
      blk_t *k = &blk[m][j];
      const uint8_t *a = mb + area_off[j];
      memset(k, 0, sizeof *k);
      int dc = (a[0] << 1) | (a[1] >> 7);
      if (dc & 256) dc -= 512;
      k->mode = (a[1] >> 6) & 1;
      k->cls = (a[1] >> 4) & 3;
      k->qno = qno;
      k->coef[0] = (int16_t)(dc * 4 + 1024);
      bitbuf own;
      own.n = own.rd = 0;
      bb_put(&own, a, 12, area_bits(j));
      stat_pass = 0;
      decode_ac(k, &own);
      if (k->pos == 64) bb_put(&mbuf[m], a, 12 + own.rd, area_bits(j)); /* what a finished block leaves is the macroblock's */
    }
    int all = 1;
    for (int j = 0; j < 6; j++) { /* pass 2: unfinished blocks, in order, from the macroblock's space */
      stat_pass = 1;
      if (blk[m][j].pos < 64 && mbuf[m].rd < mbuf[m].n) decode_ac(&blk[m][j], &mbuf[m]);
      all = all && blk[m][j].pos == 64;
    }
    if (all) /* what the macroblock leaves is the segment's */
      for (int i = mbuf[m].rd; i < mbuf[m].n; i++) vbuf.bit[vbuf.n++] = mbuf[m].bit[i];
  }
  for (int m = 0; m < 5; m++) /* pass 3 */
    for (int j = 0; j < 6; j++)
      if (blk[m][j].pos < 64 && vbuf.rd < vbuf.n) {
        stat_pass = 2;
        decode_ac(&blk[m][j], &vbuf);
      }
  if (coefs) {
    for (int m = 0; m < 5; m++)
      for (int j = 0; j < 6; j++) memcpy(coefs[6 * m + j], blk[m][j].coef, sizeof blk[m][j].coef);
    return;
  }
  for (int m = 0; m < 5; m++) {
    int x, y;
    dvo_mb_place(seq, slot, m, &x, &y);
    for (int j = 0; j < 6; j++) {
      uint8_t px[64];
      int stride, split;
      coef_pixels(blk[m][j].mode, blk[m][j].coef, px);
      uint8_t *o = pic + block_origin(x, y, j, &stride, &split);
      for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
          if (!split) o[(size_t)r * stride + c] = px[8 * r + c];
          else o[(size_t)(c < 4 ? r : r + 8) * stride + (c & 3)] = px[8 * r + c];
        }
    }
  }
}

void dvo_decode_frame(const uint8_t *dif, uint8_t *pic) {
  init_once();
  for (int seq = 0; seq < 10; seq++)
    for (int slot = 0; slot < 27; slot++) decode_segment(dif, seq, slot, pic, NULL);
}
void dvo_segment_coefs(const uint8_t *dif, int seq, int slot, int16_t coefs[30][64]) {
  init_once();
  decode_segment(dif, seq, slot, NULL, coefs);
}

/* ---------------- encoder (makes the test streams) ---------------- */
typedef struct {
  int dc, mode, cls;
  double cin[64];     /* transform inputs, natural order */
  int16_t level[64];  /* scan order */
  uint8_t bits[1200]; /* the block's AC code words, end-of-block included */
  int nbits;
} eblk_t;

static void put_bits(eblk_t *b, uint32_t v, int n) {
  for (int i = n - 1; i >= 0; i--) b->bits[b->nbits++] = (v >> i) & 1;
}
static void put_pair(eblk_t *b, int run, int level) {
  const int amp = level < 0 ? -level : level;
  if (run < 64 && enc_len[run][amp] && !(run == 0 && amp > 22) && !(amp == 0 && run > 5)) {
    put_bits(b, enc_code[run][amp], enc_len[run][amp]);
    if (amp) put_bits(b, level < 0, 1);
    return;
  }
  if (amp == 0) { /* run escape: run zeros and one more */
    put_bits(b, (0x7Eu << 6) | (unsigned)run, 13);
    return;
  }
  if (run > 0) put_pair(b, run - 1, 0); /* the zeros on their own, then the amplitude with run 0 */
  if (amp <= 22) {
    put_bits(b, enc_code[0][amp], enc_len[0][amp]);
    put_bits(b, level < 0, 1);
  } else {
    put_bits(b, (0x7Fu << 8) | (unsigned)amp, 15);
    put_bits(b, level < 0, 1);
  }
}
static void quantise(eblk_t *b, int qno) {
  b->nbits = 0;
  int run = 0;
  for (int k = 1; k < 64; k++) {
    const int sh = dvo_shift(qno, b->cls, area_of(k));
    const double step = (double)((int64_t)qbase[b->mode][k] << sh) / 16384.0;
    int lv = (int)floor(fabs(b->cin[(b->mode ? zz248 : zz88)[k]]) / step + 0.5);
    if (lv > 255) lv = 255;
    if (b->cin[(b->mode ? zz248 : zz88)[k]] < 0) lv = -lv;
    b->level[k] = (int16_t)lv;
    if (lv == 0) {
      run++;
      continue;
    }
    put_pair(b, run, lv);
    run = 0;
  }
  put_bits(b, eob_code, eob_len);
}

static void encode_segment(const uint8_t *pic, uint8_t *dif, int seq, int slot, int flags) {
  static eblk_t blk[5][6];
  for (int m = 0; m < 5; m++) {
    int x, y;
    dvo_mb_place(seq, slot, m, &x, &y);
    for (int j = 0; j < 6; j++) {
      eblk_t *b = &blk[m][j];
      int stride, split;
      const uint8_t *o = pic + block_origin(x, y, j, &stride, &split);
      double px[64];
      for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++)
          px[8 * r + c] = (split ? o[(size_t)(c < 4 ? r : r + 8) * stride + (c & 3)] : o[(size_t)r * stride + c]) - 128.0;
      /* 2-4-8 when neighbouring lines differ much more than lines two apart (a combed picture) */
      double d1 = 0, d2 = 0;
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 8; c++) d1 += fabs(px[8 * r + c] - px[8 * r + 8 + c]), d2 += fabs(px[8 * r + c] - px[8 * r + 16 + c]);
      b->mode = (flags & 1) && d1 > 2 * d2 + 64;
      double act = 0;
      for (int k = 0; k < 64; k++) {
        double s = 0;
        for (int p = 0; p < 64; p++) s += fwd[b->mode][k][p] * px[p];
        b->cin[k] = s;
        if (k) act += fabs(s);
      }
      int dc = (int)floor(b->cin[0] / 4 + 0.5);
      b->dc = dc < -256 ? -256 : dc > 255 ? 255 : dc;
      b->cls = !(flags & 2) ? 0 : act < 300 ? 0 : act < 1200 ? 1 : act < 5000 ? 2 : 3;
    }
  }
  /* rate control: the finest quantisation number whose code words fit the segment's 2680 AC bits */
  int qno = 15;
  for (;; qno--) {
    int total = 0;
    for (int m = 0; m < 5; m++)
      for (int j = 0; j < 6; j++) {
        quantise(&blk[m][j], qno);
        total += blk[m][j].nbits;
      }
    if (total <= 5 * (4 * 100 + 2 * 68) || qno == 0) break;
  }
  for (;;) { /* still too much at the coarsest step: drop the last coefficient of the longest block until it fits */
    int total = 0;
    eblk_t *big = &blk[0][0];
    for (int m = 0; m < 5; m++)
      for (int j = 0; j < 6; j++) {
        total += blk[m][j].nbits;
        if (blk[m][j].nbits > big->nbits) big = &blk[m][j];
      }
    if (total <= 5 * (4 * 100 + 2 * 68)) break;
    int last = 63;
    while (last > 0 && big->level[last] == 0) last--;
    big->cin[(big->mode ? zz248 : zz88)[last]] = 0;
    quantise(big, qno);
  }
  /* three passes: own area, then the macroblock's free space, then the segment's */
  static uint8_t mfree[5][700], vfree[2800]; /* bit positions (macroblock-relative / segment-relative) that are still free */
  (void)mfree;
  int used[5][6], vpos_n = 0;
  struct { int m, byte_bit; } vslot[2800];
  for (int m = 0; m < 5; m++) {
    uint8_t *mb = dif + dif_block(seq, 5 * slot + m);
    mb[3] = (uint8_t)qno; /* STA = 0 */
    memset(mb + 4, 0, 76);
    int freepos[700], nfree = 0; /* bit offsets inside the compressed macroblock, counted from byte 4 */
    for (int j = 0; j < 6; j++) {
      const eblk_t *b = &blk[m][j];
      uint8_t *a = mb + area_off[j];
      const unsigned hd = ((unsigned)(b->dc & 511) << 3) | ((unsigned)b->mode << 2) | (unsigned)b->cls;
      a[0] = (uint8_t)(hd >> 4);
      a[1] = (uint8_t)((hd & 15) << 4);
      const int cap = area_bits(j) - 12, n = b->nbits < cap ? b->nbits : cap;
      for (int i = 0; i < n; i++)
        if (b->bits[i]) a[(12 + i) >> 3] |= (uint8_t)(0x80 >> ((12 + i) & 7));
      used[m][j] = n;
      for (int i = n; i < cap; i++) freepos[nfree++] = 8 * (area_off[j] - 4) + 12 + i;
    }
    int fp = 0;
    for (int j = 0; j < 6; j++) { /* pass 2 */
      const eblk_t *b = &blk[m][j];
      while (used[m][j] < b->nbits && fp < nfree) {
        if (b->bits[used[m][j]]) mb[4 + (freepos[fp] >> 3)] |= (uint8_t)(0x80 >> (freepos[fp] & 7));
        used[m][j]++;
        fp++;
      }
    }
    for (; fp < nfree; fp++) vslot[vpos_n].m = m, vslot[vpos_n++].byte_bit = freepos[fp];
  }
  int vp = 0;
  for (int m = 0; m < 5; m++) /* pass 3 */
    for (int j = 0; j < 6; j++) {
      const eblk_t *b = &blk[m][j];
      while (used[m][j] < b->nbits && vp < vpos_n) {
        uint8_t *mb = dif + dif_block(seq, 5 * slot + vslot[vp].m);
        if (b->bits[used[m][j]]) mb[4 + (vslot[vp].byte_bit >> 3)] |= (uint8_t)(0x80 >> (vslot[vp].byte_bit & 7));
        used[m][j]++;
        vp++;
      }
    }
  (void)vfree;
}

void dvo_encode_frame(const uint8_t *pic, uint8_t *dif, int flags) {
  init_once();
  memset(dif, 0, DVO_FRAME_BYTES);
  for (int seq = 0; seq < 10; seq++) {
    for (int b = 0; b < 150; b++) { /* block ids: section type, sequence number, block number within the section */
      uint8_t *id = dif + ((size_t)seq * 150 + b) * 80;
      int sct, num;
      if (b == 0) sct = 0, num = 0;
      else if (b < 3) sct = 1, num = b - 1;
      else if (b < 6) sct = 2, num = b - 3;
      else if ((b - 6) % 16 == 0) sct = 3, num = (b - 6) / 16;
      else sct = 4, num = (b - 6) - (b - 6) / 16 - 1;
      id[0] = (uint8_t)((sct << 5) | 0x1F);
      id[1] = (uint8_t)((seq << 4) | 0x07);
      id[2] = (uint8_t)num;
    }
    dif[(size_t)seq * 150 * 80 + 3] = 0x3F; /* header block: DSF = 0 (525/60) */
    for (int slot = 0; slot < 27; slot++) encode_segment(pic, dif, seq, slot, flags);
  }
}

void dvo_synth(uint8_t *pic, int n, uint32_t seed, int amp) {
  for (int y = 0; y < DVO_H; y++)
    for (int x = 0; x < DVO_W; x++) {
      uint32_t h = (uint32_t)x * 0x9E3779B1u ^ (uint32_t)y * 0x85EBCA77u ^ (uint32_t)n * 0xC2B2AE3Du ^ seed * 0x27D4EB2Fu;
      h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
      int xs = x;
      if (y >= 160 && y < 224 && (y & 1)) xs = x + 12; /* a combed band: the odd field has moved */
      int v = 16 + ((xs + y + 7 * n) % 1200) * 219 / 1200;
      if (((xs / 48) + (y / 40)) % 5 == 0) v = 235 - v / 2; /* a few hard edges */
      if (amp) v += (int)(h % (uint32_t)(2 * amp + 1)) - amp;
      pic[(size_t)y * DVO_W + x] = clamp255(v);
      if (x < DVO_CW) {
        const int nz = amp ? (int)((h >> 16) % (uint32_t)(amp + 1)) - amp / 2 : 0;
        pic[(size_t)DVO_W * DVO_H + (size_t)y * DVO_CW + x] = clamp255(128 + (x - 90) / 3 + nz / 2);
        pic[(size_t)DVO_W * DVO_H + (size_t)DVO_CW * DVO_H + (size_t)y * DVO_CW + x] = clamp255(128 - (y - 240) / 8 + nz / 2);
      }
    }
}
