/*
 * rtj_oracle.c — scalar CPU restatement of the RTjpeg hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see rtj_oracle.h).  Parity status: PINNED against
 * the reference's own lib/RTjpeg.c built into oracle/_ref/ (see Makefile) and
 * against tests/golden/.
 *
 * Written from the arithmetic specification in SURVEY.md Appendix A/B; every
 * function names the reference lines it restates.  Build with -fwrapv: the
 * reference relies on two's-complement wrap in a few products.
 */
#include "rtj_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- normative constants of the codec (data, RTjpeg.c:59-107) ---- */

/* zig-zag scan order, transposed w.r.t. JPEG's (RTjpeg.c:59-74) */
static const uint8_t k_zz[64] = {
    0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40,
    33, 26, 19, 12, 5,  6,  13, 20, 27, 34, 41, 48, 56, 49, 42, 35,
    28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30,
    23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63};

/* luminance / chrominance quantiser bases (RTjpeg.c:87-107) */
static const uint8_t k_lum_q[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t k_chr_q[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* AAN scale factors, 32.32 fixed point, natural order (RTjpeg.c:76-85).
 * Normative data: checked against the reference's RTjpeg_get_tables for every Q. */
static const uint64_t k_aan[64] = {
    4294967296ULL, 5957222912ULL, 5611718144ULL, 5050464768ULL,
    4294967296ULL, 3374581504ULL, 2324432128ULL, 1184891264ULL,
    5957222912ULL, 8263040512ULL, 7783580160ULL, 7005009920ULL,
    5957222912ULL, 4680582144ULL, 3224107520ULL, 1643641088ULL,
    5611718144ULL, 7783580160ULL, 7331904512ULL, 6598688768ULL,
    5611718144ULL, 4408998912ULL, 3036936960ULL, 1548224000ULL,
    5050464768ULL, 7005009920ULL, 6598688768ULL, 5938608128ULL,
    5050464768ULL, 3968072960ULL, 2733115392ULL, 1393296000ULL,
    4294967296ULL, 5957222912ULL, 5611718144ULL, 5050464768ULL,
    4294967296ULL, 3374581504ULL, 2324432128ULL, 1184891264ULL,
    3374581504ULL, 4680582144ULL, 4408998912ULL, 3968072960ULL,
    3374581504ULL, 2651326208ULL, 1826357504ULL, 931136000ULL,
    2324432128ULL, 3224107520ULL, 3036936960ULL, 2733115392ULL,
    2324432128ULL, 1826357504ULL, 1258030336ULL, 641204288ULL,
    1184891264ULL, 1643641088ULL, 1548224000ULL, 1393296000ULL,
    1184891264ULL, 931136000ULL,  641204288ULL,  326894240ULL};

static inline uint64_t aan_at(int i) { return k_aan[i]; }

/* ------------------------------------------------------------------------ */
/* tables: RTjpeg_calc_tbls (2344-2369) + RTjpeg_dct_init (277-286) +        */
/*         RTjpeg_idct_init (1208-1217), driven by set_quality (2408-2419)   */
/* ------------------------------------------------------------------------ */
static int lead8(const int32_t *inv_unscaled) {
  /* count zig-zag ACs 1..n whose un-scaled inverse quantiser is <= 8 */
  int n = 0;
  while (n < 63 && inv_unscaled[k_zz[n + 1]] <= 8) n++;
  return n;
}

void rtjo_make_tables(int Q, rtjo_tables *t) {
  int32_t linv[64], cinv[64];
  if (Q < 1) Q = 1;
  if (Q > 255) Q = 255;
  const uint64_t qual = (uint64_t)Q << 25;
  for (int i = 0; i < 64; i++) {
    int32_t l = (int32_t)((qual / ((uint64_t)k_lum_q[i] << 16)) >> 3);
    int32_t c = (int32_t)((qual / ((uint64_t)k_chr_q[i] << 16)) >> 3);
    if (l == 0) l = 1;
    if (c == 0) c = 1;
    linv[i] = 65536 / (l << 3);
    cinv[i] = 65536 / (c << 3);
    /* the forward step is re-derived from the inverse one */
    t->lqt[i] = (65536 / linv[i]) >> 3;
    t->cqt[i] = (65536 / cinv[i]) >> 3;
  }
  t->lb8 = lead8(linv);
  t->cb8 = lead8(cinv);
  for (int i = 0; i < 64; i++) {
    t->lqt[i] = (int32_t)(((uint64_t)t->lqt[i] << 32) / aan_at(i));
    t->cqt[i] = (int32_t)(((uint64_t)t->cqt[i] << 32) / aan_at(i));
    t->liqt[i] = (int32_t)(((uint64_t)linv[i] * aan_at(i)) >> 32);
    t->ciqt[i] = (int32_t)(((uint64_t)cinv[i] * aan_at(i)) >> 32);
  }
}

/* ------------------------------------------------------------------------ */
/* bounded byte reader: bytes at or past the end read as 0                   */
/* ------------------------------------------------------------------------ */
typedef struct {
  const uint8_t *p;
  size_t n;
} rd_t;
static inline uint8_t rd_u8(const rd_t *r, size_t i) { return i < r->n ? r->p[i] : 0; }

/* ------------------------------------------------------------------------ */
/* stream -> block: RTjpeg_s2b (157-186)                                     */
/* ------------------------------------------------------------------------ */
static int s2b_at(const rd_t *r, size_t at, int bt8, const int32_t *q, int16_t coef[64]) {
  size_t ci = at;
  int co;
  /* DC is the only unsigned byte; every product is kept to 16 bits */
  coef[k_zz[0]] = (int16_t)((uint32_t)rd_u8(r, ci++) * (uint32_t)q[k_zz[0]]);
  for (co = 1; co <= bt8; co++)
    coef[k_zz[co]] = (int16_t)((uint32_t)(int32_t)(int8_t)rd_u8(r, ci++) * (uint32_t)q[k_zz[co]]);
  while (co < 64) {
    int v = (int8_t)rd_u8(r, ci++);
    if (v > 63) {
      int stop = co + (v - 63);
      if (stop > 64) stop = 64; /* reference runs off the block here; well-formed streams never do */
      while (co < stop) coef[k_zz[co++]] = 0;
    } else {
      coef[k_zz[co]] = (int16_t)((uint32_t)(int32_t)v * (uint32_t)q[k_zz[co]]);
      co++;
    }
  }
  return (int)(ci - at);
}

int rtjo_s2b(const uint8_t *strm, size_t avail, int bt8, const int32_t *qtbl, int16_t coef[64]) {
  rd_t r = {strm, avail};
  return s2b_at(&r, 0, bt8, qtbl, coef);
}

/* ------------------------------------------------------------------------ */
/* inverse transform: RTjpeg_idct, C path (2209-2332; constants 1196-1206)   */
/* ------------------------------------------------------------------------ */
static inline int32_t mulr8(int32_t x, int32_t c) { return (int32_t)(x * c + 128) >> 8; }

/* one 8-point AAN inverse pass, no scaling; used for both directions */
static void idct8(const int32_t x[8], int32_t y[8]) {
  const int32_t s04 = x[0] + x[4], d04 = x[0] - x[4];
  const int32_t s26 = x[2] + x[6];
  const int32_t r26 = mulr8(x[2] - x[6], 362) - s26;
  const int32_t e0 = s04 + s26, e3 = s04 - s26, e1 = d04 + r26, e2 = d04 - r26;

  const int32_t s53 = x[5] + x[3], d53 = x[5] - x[3];
  const int32_t s17 = x[1] + x[7], d17 = x[1] - x[7];
  const int32_t o7 = s17 + s53;
  const int32_t m = mulr8(s17 - s53, 362);
  const int32_t z5 = mulr8(d53 + d17, 473);
  const int32_t a = mulr8(d17, 277) - z5;
  const int32_t b = mulr8(d53, -669) + z5;
  const int32_t o6 = b - o7, o5 = m - o6, o4 = a + o5;

  y[0] = e0 + o7; y[7] = e0 - o7;
  y[1] = e1 + o6; y[6] = e1 - o6;
  y[2] = e2 + o5; y[5] = e2 - o5;
  y[4] = e3 + o4; y[3] = e3 - o4;
}

static inline uint8_t clamp_px(int32_t v) {
  const int16_t s = (int16_t)((v + 4) >> 3); /* narrowing happens before the clamp */
  return (uint8_t)(s > 235 ? 235 : (s < 16 ? 16 : s));
}

void rtjo_idct(const int16_t coef[64], uint8_t *dst, int stride) {
  int32_t ws[64], in[8], out[8];
  for (int c = 0; c < 8; c++) { /* columns first, unscaled 32-bit workspace */
    for (int r = 0; r < 8; r++) in[r] = coef[8 * r + c];
    idct8(in, out);
    for (int r = 0; r < 8; r++) ws[8 * r + c] = out[r];
  }
  for (int r = 0; r < 8; r++) { /* then rows, descale + clamp 16..235 (chroma too) */
    idct8(&ws[8 * r], out);
    for (int c = 0; c < 8; c++) dst[r * stride + c] = clamp_px(out[c]);
  }
}

/* ------------------------------------------------------------------------ */
/* decoder state + macroblock walk: RTjpeg_decompress (3565-3586),           */
/* RTjpeg_set_size (2427-2453), RTjpeg_decompressYUV420 (2688-2749)          */
/* ------------------------------------------------------------------------ */
struct rtjo_dec {
  int w, h, Q; /* Q==0: tables never built (all zero), as after RTjpeg_init's bzero */
  rtjo_tables t;
};

rtjo_dec *rtjo_dec_new(void) { return (rtjo_dec *)calloc(1, sizeof(rtjo_dec)); }
void rtjo_dec_free(rtjo_dec *d) { free(d); }
int rtjo_dec_quality(const rtjo_dec *d) { return d->Q; }

static int dec_header(rtjo_dec *d, const uint8_t *pkt, size_t len) {
  uint8_t hdr[RTJO_HEADER_SIZE] = {0};
  memcpy(hdr, pkt, len < RTJO_HEADER_SIZE ? len : RTJO_HEADER_SIZE);
  const int w = hdr[6] | (hdr[7] << 8), h = hdr[8] | (hdr[9] << 8), q = hdr[10];
  if (w <= 0 || h <= 0 || (w & 15) || (h & 15)) return -1;
  d->w = w;
  d->h = h;
  if (q != d->Q) { /* a 0 here on a fresh decoder keeps the zero tables */
    d->Q = q < 1 ? 1 : q;
    rtjo_make_tables(d->Q, &d->t);
  }
  return 0;
}

static long walk(rtjo_dec *d, const uint8_t *pkt, size_t len, uint8_t *y, uint8_t *u,
                 uint8_t *v, uint32_t *offs) {
  if (dec_header(d, pkt, len) < 0) return -1;
  const rd_t r = {pkt, len};
  const int w = d->w, cw = w >> 1;
  size_t sp = RTJO_HEADER_SIZE;
  long nblk = 0;
  int16_t coef[64];
  for (int my = 0; my < d->h / 16; my++) {
    for (int mx = 0; mx < w / 16; mx++) {
      for (int k = 0; k < 6; k++) {
        const int chroma = k >= 4;
        if (offs) offs[nblk] = (uint32_t)sp;
        nblk++;
        if (rd_u8(&r, sp) == 0xFF) { /* unchanged block: keep previous pixels */
          sp++;
          continue;
        }
        sp += s2b_at(&r, sp, chroma ? d->t.cb8 : d->t.lb8, chroma ? d->t.ciqt : d->t.liqt, coef);
        if (!y) continue;
        if (!chroma)
          rtjo_idct(coef, y + (size_t)(16 * my + 8 * (k >> 1)) * w + 16 * mx + 8 * (k & 1), w);
        else
          rtjo_idct(coef, (k == 4 ? u : v) + (size_t)(8 * my) * cw + 8 * mx, cw);
      }
    }
  }
  if (offs) offs[nblk] = (uint32_t)sp;
  return y ? (long)sp : nblk;
}

long rtjo_decode(rtjo_dec *d, const uint8_t *pkt, size_t len, uint8_t *y, uint8_t *u,
                 uint8_t *v) {
  return walk(d, pkt, len, y, u, v, NULL);
}

long rtjo_block_offsets(rtjo_dec *d, const uint8_t *pkt, size_t len, uint32_t *offs) {
  return walk(d, pkt, len, NULL, NULL, NULL, offs);
}

/* ------------------------------------------------------------------------ */
/* encoder (stream generator): RTjpeg_dctY (288-389), RTjpeg_quant (245-252),*/
/* RTjpeg_b2s (109-155), RTjpeg_bcomp (2827-2838), compressYUV420 (2510-2563)*/
/* mcompressYUV420 (2841-2921), RTjpeg_compress (3488-3524)                  */
/* ------------------------------------------------------------------------ */
struct rtjo_enc {
  int w, h, Q, key_rate, key_count;
  int lmask, cmask;
  rtjo_tables t;
  int16_t *old; /* previous quantised blocks, 64 per block, stream order */
};

rtjo_enc *rtjo_enc_new(int width, int height, int Q, int key_rate, int lmask, int cmask) {
  if (width <= 0 || height <= 0 || (width & 15) || (height & 15)) return NULL;
  rtjo_enc *e = (rtjo_enc *)calloc(1, sizeof(*e));
  e->w = width;
  e->h = height;
  e->Q = Q < 1 ? 1 : (Q > 255 ? 255 : Q);
  rtjo_make_tables(e->Q, &e->t);
  e->key_rate = key_rate < 0 ? 0 : (key_rate > 255 ? 255 : key_rate);
  e->lmask = lmask < 0 ? 0 : (lmask > 16 ? 16 : lmask);
  e->cmask = cmask < 0 ? 0 : (cmask > 16 ? 16 : cmask);
  if (e->key_rate > 0)
    e->old = (int16_t *)calloc((size_t)(width / 16) * (height / 16) * 6 * 64, sizeof(int16_t));
  return e;
}
void rtjo_enc_free(rtjo_enc *e) {
  if (!e) return;
  free(e->old);
  free(e);
}

/* 8-point forward AAN butterfly; returns unscaled r[0], r[4] and 8-bit-scaled others */
static void fdct8(const int32_t p[8], int32_t r[8]) {
  const int32_t a0 = p[0] + p[7], a7 = p[0] - p[7], a1 = p[1] + p[6], a6 = p[1] - p[6];
  const int32_t a2 = p[2] + p[5], a5 = p[2] - p[5], a3 = p[3] + p[4], a4 = p[3] - p[4];
  const int32_t b0 = a0 + a3, b3 = a0 - a3, b1 = a1 + a2, b2 = a1 - a2;
  r[0] = b0 + b1;
  r[4] = b0 - b1;
  const int32_t z1 = (b2 + b3) * 181;
  r[2] = (b3 << 8) + z1;
  r[6] = (b3 << 8) - z1;
  const int32_t c0 = a4 + a5, c1 = a5 + a6, c2 = a6 + a7;
  const int32_t z5 = (c0 - c2) * 98;
  const int32_t z2 = c0 * 139 + z5, z4 = c2 * 334 + z5, z3 = c1 * 181;
  const int32_t z11 = (a7 << 8) + z3, z13 = (a7 << 8) - z3;
  r[5] = z13 + z2;
  r[3] = z13 - z2;
  r[1] = z11 + z4;
  r[7] = z11 - z4;
}

static void fdct_block(const uint8_t *src, int stride, int16_t blk[64]) {
  int32_t ws[64], in[8], r[8];
  for (int row = 0; row < 8; row++) {
    for (int c = 0; c < 8; c++) in[c] = src[row * stride + c];
    fdct8(in, r);
    r[0] <<= 8;
    r[4] <<= 8;
    memcpy(&ws[8 * row], r, sizeof r);
  }
  for (int c = 0; c < 8; c++) {
    for (int k = 0; k < 8; k++) in[k] = ws[8 * k + c];
    fdct8(in, r);
    for (int k = 0; k < 8; k++)
      blk[8 * k + c] = (k == 0 || k == 4) ? (int16_t)((r[k] + 128) >> 8)
                                          : (int16_t)((r[k] + 32768) >> 16);
  }
}

static void quant_block(int16_t blk[64], const int32_t *q) {
  for (int i = 0; i < 64; i++) blk[i] = (int16_t)((blk[i] * q[i] + 32767) >> 16);
}

static int b2s(const int16_t blk[64], uint8_t *out, int bt8) {
  int n = 0, k;
  int16_t v = blk[k_zz[0]];
  out[n++] = (uint8_t)(v > 254 ? 254 : (v < 0 ? 0 : v));
  for (k = 1; k <= bt8; k++) {
    v = blk[k_zz[k]];
    out[n++] = (uint8_t)(int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
  }
  while (k < 64) {
    v = blk[k_zz[k]];
    if (v != 0) {
      out[n++] = (uint8_t)(int8_t)(v > 63 ? 63 : (v < -64 ? -64 : v));
      k++;
    } else { /* a run of zeros to the next non-zero or the end of the block */
      int run = 0;
      while (k < 64 && blk[k_zz[k]] == 0) { k++; run++; }
      out[n++] = (uint8_t)(63 + run);
    }
  }
  return n;
}

long rtjo_encode(rtjo_enc *e, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                 uint8_t *out) {
  const int w = e->w, cw = w >> 1;
  uint8_t *sp = out + RTJO_HEADER_SIZE;
  int16_t blk[64];
  int16_t *old = e->old;
  const int inter = e->key_rate > 0;
  if (inter && e->key_count == 0)
    memset(e->old, 0, (size_t)(w / 16) * (e->h / 16) * 6 * 64 * sizeof(int16_t));
  for (int my = 0; my < e->h / 16; my++) {
    for (int mx = 0; mx < w / 16; mx++) {
      for (int k = 0; k < 6; k++) {
        const int chroma = k >= 4;
        if (!chroma)
          fdct_block(y + (size_t)(16 * my + 8 * (k >> 1)) * w + 16 * mx + 8 * (k & 1), w, blk);
        else
          fdct_block((k == 4 ? u : v) + (size_t)(8 * my) * cw + 8 * mx, cw, blk);
        quant_block(blk, chroma ? e->t.cqt : e->t.lqt);
        if (inter) {
          const int mask = chroma ? e->cmask : e->lmask;
          int same = 1;
          for (int i = 0; i < 64 && same; i++) same = abs(old[i] - blk[i]) <= mask;
          if (same) {
            *sp++ = 0xFF;
            old += 64;
            continue;
          }
          memcpy(old, blk, sizeof blk);
          old += 64;
        }
        sp += b2s(blk, sp, chroma ? e->t.cb8 : e->t.lb8);
      }
    }
  }
  const uint32_t total = (uint32_t)(sp - out);
  out[0] = (uint8_t)total; out[1] = (uint8_t)(total >> 8);
  out[2] = (uint8_t)(total >> 16); out[3] = (uint8_t)(total >> 24);
  out[4] = RTJO_HEADER_SIZE;
  out[5] = 0;
  out[6] = (uint8_t)w; out[7] = (uint8_t)(w >> 8);
  out[8] = (uint8_t)e->h; out[9] = (uint8_t)(e->h >> 8);
  out[10] = (uint8_t)e->Q;
  out[11] = 0;
  if (inter) {
    out[11] = (uint8_t)e->key_count;
    if (++e->key_count > e->key_rate) e->key_count = 0;
  }
  return (long)total;
}

/* ------------------------------------------------------------------------ */
/* colour stage: RTjpeg_yuv420rgb32 (3123), bgr32 (3192), rgb24 (3261),      */
/* bgr24 (3326), rgb16 (3391); constants 3071-3075.  planes[1] is Cb.        */
/* ------------------------------------------------------------------------ */
static inline int sat8(int32_t v) { return v > 255 ? 255 : (v < 0 ? 0 : v); }

void rtjo_yuv420_to_rgb(int fmt, int w, int h, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                        uint8_t *dst, size_t pitch) {
  for (int row = 0; row < h; row++) {
    uint8_t *o = dst + (size_t)row * pitch;
    for (int x = 0; x < w; x++) {
      const int32_t cb = u[(row >> 1) * (w >> 1) + (x >> 1)] - 128, cr = v[(row >> 1) * (w >> 1) + (x >> 1)] - 128;
      const int32_t yy = (y[row * w + x] - 16) * 76284;
      const int r = sat8((yy + cr * 76284) >> 16);
      const int g = sat8((yy - cr * 53281 - cb * 25625) >> 16);
      const int b = sat8((yy + cb * 132252) >> 16);
      switch (fmt) {
        case 0: o[4 * x] = (uint8_t)r; o[4 * x + 1] = (uint8_t)g; o[4 * x + 2] = (uint8_t)b; break;
        case 1: o[4 * x] = (uint8_t)b; o[4 * x + 1] = (uint8_t)g; o[4 * x + 2] = (uint8_t)r; break;
        case 2: o[3 * x] = (uint8_t)r; o[3 * x + 1] = (uint8_t)g; o[3 * x + 2] = (uint8_t)b; break;
        case 3: o[3 * x] = (uint8_t)b; o[3 * x + 1] = (uint8_t)g; o[3 * x + 2] = (uint8_t)r; break;
        default: {
          const int p = (b >> 3) | ((g >> 2) << 5) | ((r >> 3) << 11);
          o[2 * x] = (uint8_t)(p & 0xff);
          o[2 * x + 1] = (uint8_t)(p >> 8);
        }
      }
    }
  }
}
