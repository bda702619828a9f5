/*
 * qt_rtj0_host.c — minimal QuickTime reader/writer for RTJ0 tracks (include/mi_qtrtj.h).
 * Host C only.  Atom layouts as the reference's parsers read them: stsd video entry
 * lib/qt_stsd.c:262-273,402-428; stts lib/qt_stts.c; stsc lib/qt_stsc.c; stsz lib/qt_stsz.c:44-66;
 * stco/co64 lib/qt_stco.c; stss lib/qt_stss.c; mdhd lib/qt_mdhd.c; hdlr lib/qt_hdlr.c.
 */
#include "mi_qtrtj.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ writer */

struct mi_qt_writer {
  FILE *f;
  int width, height;
  uint32_t timescale, frame_duration;
  uint64_t mdat_pos;      /* file offset of the mdat atom header (16 bytes: extended size) */
  uint64_t *off;          /* per sample */
  uint32_t *size;
  uint8_t *key;
  uint64_t n, cap;
  int failed;
};

typedef struct {
  uint8_t *p;
  size_t n, cap;
  int oom;
} buf_t;

static void put(buf_t *b, const void *src, size_t n) {
  if (b->oom) return;
  if (b->n + n > b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 1024;
    while (nc < b->n + n) nc *= 2;
    uint8_t *q = (uint8_t *)realloc(b->p, nc);
    if (!q) {
      b->oom = 1;
      return;
    }
    b->p = q;
    b->cap = nc;
  }
  memcpy(b->p + b->n, src, n);
  b->n += n;
}
static void put8(buf_t *b, unsigned v) {
  uint8_t x = (uint8_t)v;
  put(b, &x, 1);
}
static void put16(buf_t *b, unsigned v) {
  uint8_t x[2] = {(uint8_t)(v >> 8), (uint8_t)v};
  put(b, x, 2);
}
static void put32(buf_t *b, uint32_t v) {
  uint8_t x[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
  put(b, x, 4);
}
static void put64(buf_t *b, uint64_t v) {
  put32(b, (uint32_t)(v >> 32));
  put32(b, (uint32_t)v);
}
static void putz(buf_t *b, size_t n) {
  while (n--) put8(b, 0);
}
/* atoms are built inside out: begin() reserves the header, end() patches the size */
static size_t begin(buf_t *b, const char *type) {
  size_t at = b->n;
  put32(b, 0);
  put(b, type, 4);
  return at;
}
static void end(buf_t *b, size_t at) {
  if (b->oom) return;
  uint32_t sz = (uint32_t)(b->n - at);
  b->p[at] = (uint8_t)(sz >> 24);
  b->p[at + 1] = (uint8_t)(sz >> 16);
  b->p[at + 2] = (uint8_t)(sz >> 8);
  b->p[at + 3] = (uint8_t)sz;
}
static void matrix(buf_t *b) { /* identity, 16.16 / 2.30 fixed point */
  static const uint32_t m[9] = {0x00010000, 0, 0, 0, 0x00010000, 0, 0, 0, 0x40000000};
  for (int i = 0; i < 9; i++) put32(b, m[i]);
}

mi_qt_writer *mi_qt_writer_open(const char *path, int width, int height, uint32_t timescale, uint32_t frame_duration) {
  if (!path || width <= 0 || height <= 0 || width > 65535 || height > 65535 || !timescale || !frame_duration) return NULL;
  mi_qt_writer *w = (mi_qt_writer *)calloc(1, sizeof(*w));
  if (!w) return NULL;
  w->f = fopen(path, "wb");
  if (!w->f) {
    free(w);
    return NULL;
  }
  w->width = width;
  w->height = height;
  w->timescale = timescale;
  w->frame_duration = frame_duration;
  buf_t b = {0};
  size_t a = begin(&b, "ftyp");
  put(&b, "qt  ", 4);
  put32(&b, 0x20050300);
  put(&b, "qt  ", 4);
  end(&b, a);
  w->mdat_pos = b.n;
  put32(&b, 1); /* size 1: the real size follows the type as 64 bits (patched on close) */
  put(&b, "mdat", 4);
  put64(&b, 16);
  if (b.oom || fwrite(b.p, 1, b.n, w->f) != b.n) {
    fclose(w->f);
    free(b.p);
    free(w);
    return NULL;
  }
  free(b.p);
  return w;
}

int mi_qt_writer_add(mi_qt_writer *w, const uint8_t *pkt, uint32_t len, int keyframe) {
  if (!w || (!pkt && len)) return MI_QT_ERR_ARG;
  if (w->failed) return MI_QT_ERR_IO;
  if (w->n == w->cap) {
    uint64_t nc = w->cap ? w->cap * 2 : 256;
    uint64_t *o = (uint64_t *)realloc(w->off, nc * sizeof(*o));
    if (o) w->off = o;
    uint32_t *s = o ? (uint32_t *)realloc(w->size, nc * sizeof(*s)) : NULL;
    if (s) w->size = s;
    uint8_t *k = s ? (uint8_t *)realloc(w->key, nc) : NULL;
    if (k) w->key = k;
    if (!o || !s || !k) return MI_QT_ERR_NOMEM;
    w->cap = nc;
  }
  long pos = ftell(w->f);
  if (pos < 0 || (len && fwrite(pkt, 1, len, w->f) != len)) {
    w->failed = 1;
    return MI_QT_ERR_IO;
  }
  w->off[w->n] = (uint64_t)pos;
  w->size[w->n] = len;
  w->key[w->n] = keyframe ? 1 : 0;
  w->n++;
  return MI_QT_OK;
}

int mi_qt_writer_close(mi_qt_writer *w) {
  if (!w) return MI_QT_ERR_ARG;
  int rc = w->failed ? MI_QT_ERR_IO : MI_QT_OK;
  long endpos = ftell(w->f);
  if (endpos < 0) rc = MI_QT_ERR_IO;
  buf_t b = {0};
  const uint64_t n = w->n, dur = n * (uint64_t)w->frame_duration;
  uint64_t nkey = 0;
  int big = 0;
  for (uint64_t i = 0; i < n; i++) {
    nkey += w->key[i];
    if (w->off[i] > 0xFFFFFFFFull) big = 1;
  }
  size_t moov = begin(&b, "moov");
  {
    size_t a = begin(&b, "mvhd");
    put32(&b, 0);                     /* version, flags */
    put32(&b, 0), put32(&b, 0);       /* creation, modification */
    put32(&b, w->timescale), put32(&b, (uint32_t)dur);
    put32(&b, 0x00010000), put16(&b, 0x0100), putz(&b, 10); /* rate, volume, reserved */
    matrix(&b);
    putz(&b, 24);                     /* preview, poster, selection, current time */
    put32(&b, 2);                     /* next track id */
    end(&b, a);
  }
  size_t trak = begin(&b, "trak");
  {
    size_t a = begin(&b, "tkhd");
    put32(&b, 0x0000000F);            /* version 0, enabled | in movie | in preview | in poster */
    put32(&b, 0), put32(&b, 0), put32(&b, 1), put32(&b, 0), put32(&b, (uint32_t)dur);
    putz(&b, 8), put16(&b, 0), put16(&b, 0), put16(&b, 0), put16(&b, 0);
    matrix(&b);
    put32(&b, (uint32_t)w->width << 16), put32(&b, (uint32_t)w->height << 16);
    end(&b, a);
  }
  size_t mdia = begin(&b, "mdia");
  {
    size_t a = begin(&b, "mdhd");
    put32(&b, 0), put32(&b, 0), put32(&b, 0), put32(&b, w->timescale), put32(&b, (uint32_t)dur), put16(&b, 0), put16(&b, 0);
    end(&b, a);
    a = begin(&b, "hdlr");
    put32(&b, 0), put(&b, "mhlr", 4), put(&b, "vide", 4), putz(&b, 12), put8(&b, 0); /* empty pascal name */
    end(&b, a);
  }
  size_t minf = begin(&b, "minf");
  {
    size_t a = begin(&b, "vmhd");
    put32(&b, 1), put16(&b, 0x40), put16(&b, 0x8000), put16(&b, 0x8000), put16(&b, 0x8000);
    end(&b, a);
    a = begin(&b, "hdlr");
    put32(&b, 0), put(&b, "dhlr", 4), put(&b, "alis", 4), putz(&b, 12), put8(&b, 0);
    end(&b, a);
    a = begin(&b, "dinf");
    size_t d = begin(&b, "dref");
    put32(&b, 0), put32(&b, 1);
    size_t e = begin(&b, "alis");
    put32(&b, 1); /* self-contained */
    end(&b, e);
    end(&b, d);
    end(&b, a);
  }
  size_t stbl = begin(&b, "stbl");
  {
    size_t a = begin(&b, "stsd");
    put32(&b, 0), put32(&b, 1);
    size_t e = begin(&b, "RTJ0");     /* sample description: size, fourcc, then (lib/qt_stsd.c:262-273) */
    putz(&b, 6), put16(&b, 1);        /* reserved, data reference index */
    put16(&b, 0), put16(&b, 0), put(&b, "mirt", 4);        /* version, revision, vendor */
    put32(&b, 0), put32(&b, 512);     /* temporal, spatial quality (:413-414) */
    put16(&b, (unsigned)w->width), put16(&b, (unsigned)w->height);
    put32(&b, 0x00480000), put32(&b, 0x00480000);          /* 72 dpi */
    put32(&b, 0), put16(&b, 1);       /* data size, frames per sample */
    put8(&b, 6), put(&b, "RTjpeg", 6), putz(&b, 25);        /* 32-byte pascal compressor name */
    put16(&b, 24), put16(&b, 0xFFFF); /* depth, colour table id (-1: none) */
    end(&b, e);
    end(&b, a);

    a = begin(&b, "stts");
    put32(&b, 0);
    if (n) put32(&b, 1), put32(&b, (uint32_t)n), put32(&b, w->frame_duration);
    else put32(&b, 0);
    end(&b, a);

    if (nkey != n) { /* no stss: every sample is a key frame (lib/demux_quicktime.c:1522-1525) */
      a = begin(&b, "stss");
      put32(&b, 0), put32(&b, (uint32_t)nkey);
      for (uint64_t i = 0; i < n; i++)
        if (w->key[i]) put32(&b, (uint32_t)(i + 1));
      end(&b, a);
    }

    a = begin(&b, "stsc"); /* one sample per chunk throughout */
    put32(&b, 0);
    if (n) put32(&b, 1), put32(&b, 1), put32(&b, 1), put32(&b, 1);
    else put32(&b, 0);
    end(&b, a);

    a = begin(&b, "stsz");
    put32(&b, 0), put32(&b, 0), put32(&b, (uint32_t)n);
    for (uint64_t i = 0; i < n; i++) put32(&b, w->size[i]);
    end(&b, a);

    a = begin(&b, big ? "co64" : "stco");
    put32(&b, 0), put32(&b, (uint32_t)n);
    for (uint64_t i = 0; i < n; i++) {
      if (big) put64(&b, w->off[i]);
      else put32(&b, (uint32_t)w->off[i]);
    }
    end(&b, a);
  }
  end(&b, stbl);
  end(&b, minf);
  end(&b, mdia);
  end(&b, trak);
  end(&b, moov);
  if (b.oom) rc = MI_QT_ERR_NOMEM;
  if (rc == MI_QT_OK && fwrite(b.p, 1, b.n, w->f) != b.n) rc = MI_QT_ERR_IO;
  if (rc == MI_QT_OK) { /* the mdat atom spans header .. end of the packets */
    buf_t s = {0};
    put64(&s, (uint64_t)endpos - w->mdat_pos);
    if (s.oom || fseek(w->f, (long)w->mdat_pos + 8, SEEK_SET) != 0 || fwrite(s.p, 1, 8, w->f) != 8) rc = MI_QT_ERR_IO;
    free(s.p);
  }
  if (fclose(w->f) != 0 && rc == MI_QT_OK) rc = MI_QT_ERR_IO;
  free(b.p);
  free(w->off);
  free(w->size);
  free(w->key);
  free(w);
  return rc;
}

/* ------------------------------------------------------------------------------------------ reader */

struct mi_qt_reader {
  FILE *f;
  uint64_t fsize;
  uint32_t fourcc, timescale;
  int width, height;
  uint64_t n;
  mi_qt_sample *s;
};

typedef struct {
  const uint8_t *p;
  size_t n;
} view_t;

static uint32_t rd32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint64_t rd64(const uint8_t *p) { return ((uint64_t)rd32(p) << 32) | rd32(p + 4); }

/* next child atom of `in` at *pos: type, payload view; returns 0 at the end or on a malformed size */
static int next_atom(view_t in, size_t *pos, uint32_t *type, view_t *payload) {
  if (*pos + 8 > in.n) return 0;
  uint64_t size = rd32(in.p + *pos);
  *type = rd32(in.p + *pos + 4);
  size_t hdr = 8;
  if (size == 1) {
    if (*pos + 16 > in.n) return 0;
    size = rd64(in.p + *pos + 8);
    hdr = 16;
  } else if (size == 0) {
    size = in.n - *pos; /* extends to the end of the container */
  }
  if (size < hdr || size > in.n - *pos) return 0;
  payload->p = in.p + *pos + hdr;
  payload->n = (size_t)(size - hdr);
  *pos += (size_t)size;
  return 1;
}
static int find(view_t in, uint32_t want, view_t *out) {
  size_t pos = 0;
  uint32_t t;
  view_t v;
  while (next_atom(in, &pos, &t, &v))
    if (t == want) {
      *out = v;
      return 1;
    }
  return 0;
}
static void seterr(char *err, size_t n, const char *fmt, ...) {
  if (!err || !n) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err, n, fmt, ap);
  va_end(ap);
}

/* sample tables of one trak -> r->s[].  Returns MI_QT_OK, or an error with a message. */
static int build_index(mi_qt_reader *r, view_t stbl, char *err, size_t errlen) {
  view_t stsd, stts, stsc, stsz, stco, stss = {NULL, 0};
  int co64 = 0, has_stss;
  if (!find(stbl, MI_QT_FOURCC('s', 't', 's', 'd'), &stsd) || !find(stbl, MI_QT_FOURCC('s', 't', 't', 's'), &stts) ||
      !find(stbl, MI_QT_FOURCC('s', 't', 's', 'c'), &stsc) || !find(stbl, MI_QT_FOURCC('s', 't', 's', 'z'), &stsz))
    return seterr(err, errlen, "stbl lacks stsd/stts/stsc/stsz"), MI_QT_ERR_FORMAT;
  if (!find(stbl, MI_QT_FOURCC('s', 't', 'c', 'o'), &stco)) {
    if (!find(stbl, MI_QT_FOURCC('c', 'o', '6', '4'), &stco)) return seterr(err, errlen, "stbl lacks stco/co64"), MI_QT_ERR_FORMAT;
    co64 = 1;
  }
  has_stss = find(stbl, MI_QT_FOURCC('s', 't', 's', 's'), &stss);
  /* stsd: version/flags, count, then entries: size, fourcc, 6 reserved, dref index, then the video fields */
  if (stsd.n < 8 + 8 + 8 + 16 + 8 || rd32(stsd.p + 4) < 1) return seterr(err, errlen, "stsd too short"), MI_QT_ERR_FORMAT;
  {
    const uint8_t *e = stsd.p + 8;
    const uint32_t esize = rd32(e);
    if (esize < 16 + 70 || esize > stsd.n - 8) return seterr(err, errlen, "stsd entry of %u bytes", esize), MI_QT_ERR_FORMAT;
    r->fourcc = rd32(e + 4);
    /* e+16: version(2) revision(2) vendor(4) temporal(4) spatial(4) width(2) height(2) */
    r->width = (e[16 + 16] << 8) | e[16 + 17];
    r->height = (e[16 + 18] << 8) | e[16 + 19];
  }
  /* sizes */
  if (stsz.n < 12) return seterr(err, errlen, "stsz too short"), MI_QT_ERR_FORMAT;
  const uint32_t fixed = rd32(stsz.p + 4), nsz = rd32(stsz.p + 8);
  if (!fixed && (uint64_t)nsz * 4 > stsz.n - 12) return seterr(err, errlen, "stsz: %u entries do not fit", nsz), MI_QT_ERR_FORMAT;
  /* chunk offsets */
  if (stco.n < 8) return seterr(err, errlen, "stco too short"), MI_QT_ERR_FORMAT;
  const uint32_t nch = rd32(stco.p + 4);
  if ((uint64_t)nch * (co64 ? 8 : 4) > stco.n - 8) return seterr(err, errlen, "stco: %u entries do not fit", nch), MI_QT_ERR_FORMAT;
  /* sample-to-chunk runs */
  if (stsc.n < 8) return seterr(err, errlen, "stsc too short"), MI_QT_ERR_FORMAT;
  const uint32_t nsc = rd32(stsc.p + 4);
  if ((uint64_t)nsc * 12 > stsc.n - 8) return seterr(err, errlen, "stsc: %u entries do not fit", nsc), MI_QT_ERR_FORMAT;
  /* time-to-sample runs */
  if (stts.n < 8) return seterr(err, errlen, "stts too short"), MI_QT_ERR_FORMAT;
  const uint32_t ntt = rd32(stts.p + 4);
  if ((uint64_t)ntt * 8 > stts.n - 8) return seterr(err, errlen, "stts: %u entries do not fit", ntt), MI_QT_ERR_FORMAT;

  r->n = nsz;
  r->s = (mi_qt_sample *)calloc(nsz ? nsz : 1, sizeof(*r->s));
  if (!r->s) return seterr(err, errlen, "out of memory"), MI_QT_ERR_NOMEM;
  for (uint32_t i = 0; i < nsz; i++) r->s[i].size = fixed ? fixed : rd32(stsz.p + 12 + 4 * (size_t)i);
  /* offsets: walk the chunks, each run of stsc says how many samples its chunks hold */
  uint64_t si = 0;
  for (uint32_t k = 0; k < nsc && si < nsz; k++) {
    const uint32_t first = rd32(stsc.p + 8 + 12 * (size_t)k), per = rd32(stsc.p + 12 + 12 * (size_t)k);
    const uint32_t next = k + 1 < nsc ? rd32(stsc.p + 8 + 12 * (size_t)(k + 1)) : nch + 1;
    if (first < 1 || next < first || !per) return seterr(err, errlen, "stsc run %u is malformed", k), MI_QT_ERR_FORMAT;
    for (uint32_t c = first; c < next && c <= nch && si < nsz; c++) {
      uint64_t o = co64 ? rd64(stco.p + 8 + 8 * (size_t)(c - 1)) : rd32(stco.p + 8 + 4 * (size_t)(c - 1));
      for (uint32_t j = 0; j < per && si < nsz; j++) {
        r->s[si].offset = o;
        o += r->s[si].size;
        si++;
      }
    }
  }
  if (si != nsz) return seterr(err, errlen, "sample tables describe %llu of %u samples", (unsigned long long)si, nsz), MI_QT_ERR_FORMAT;
  /* times */
  int64_t t = 0;
  si = 0;
  for (uint32_t k = 0; k < ntt; k++) {
    const uint32_t cnt = rd32(stts.p + 8 + 8 * (size_t)k), d = rd32(stts.p + 12 + 8 * (size_t)k);
    for (uint32_t j = 0; j < cnt && si < nsz; j++) {
      r->s[si].pts = t;
      r->s[si].duration = d;
      t += d;
      si++;
    }
  }
  if (si != nsz) return seterr(err, errlen, "stts covers %llu of %u samples", (unsigned long long)si, nsz), MI_QT_ERR_FORMAT;
  /* key frames */
  if (!has_stss) {
    for (uint32_t i = 0; i < nsz; i++) r->s[i].keyframe = 1;
  } else {
    if (stss.n < 8 || (uint64_t)rd32(stss.p + 4) * 4 > stss.n - 8) return seterr(err, errlen, "stss malformed"), MI_QT_ERR_FORMAT;
    for (uint32_t k = 0, nk = rd32(stss.p + 4); k < nk; k++) {
      const uint32_t id = rd32(stss.p + 8 + 4 * (size_t)k);
      if (id >= 1 && id <= nsz) r->s[id - 1].keyframe = 1;
    }
  }
  for (uint32_t i = 0; i < nsz; i++)
    if (r->s[i].offset > r->fsize || r->s[i].size > r->fsize - r->s[i].offset)
      return seterr(err, errlen, "sample %u lies outside the file", i), MI_QT_ERR_FORMAT;
  return MI_QT_OK;
}

mi_qt_reader *mi_qt_reader_open(const char *path, char *err, size_t errlen) {
  if (!path) return seterr(err, errlen, "no path"), NULL;
  FILE *f = fopen(path, "rb");
  if (!f) return seterr(err, errlen, "cannot open %s", path), NULL;
  mi_qt_reader *r = (mi_qt_reader *)calloc(1, sizeof(*r));
  uint8_t *moov = NULL;
  if (!r) goto fail_msg;
  r->f = f;
  if (fseek(f, 0, SEEK_END) != 0) goto fail_msg;
  {
    long sz = ftell(f);
    if (sz < 0) goto fail_msg;
    r->fsize = (uint64_t)sz;
  }
  /* top level: find moov without reading mdat */
  uint64_t pos = 0, moov_off = 0, moov_len = 0;
  while (pos + 8 <= r->fsize) {
    uint8_t h[16];
    if (fseek(f, (long)pos, SEEK_SET) != 0 || fread(h, 1, 8, f) != 8) break;
    uint64_t size = rd32(h);
    const uint32_t type = rd32(h + 4);
    uint64_t hdr = 8;
    if (size == 1) {
      if (fread(h + 8, 1, 8, f) != 8) break;
      size = rd64(h + 8);
      hdr = 16;
    } else if (size == 0) {
      size = r->fsize - pos;
    }
    if (size < hdr || size > r->fsize - pos) {
      seterr(err, errlen, "atom at %llu has size %llu", (unsigned long long)pos, (unsigned long long)size);
      goto fail;
    }
    if (type == MI_QT_FOURCC('m', 'o', 'o', 'v')) {
      moov_off = pos + hdr;
      moov_len = size - hdr;
      break;
    }
    pos += size;
  }
  if (!moov_len) {
    seterr(err, errlen, "no moov atom");
    goto fail;
  }
  if (moov_len > (1ull << 31)) {
    seterr(err, errlen, "moov of %llu bytes", (unsigned long long)moov_len);
    goto fail;
  }
  moov = (uint8_t *)malloc((size_t)moov_len);
  if (!moov || fseek(f, (long)moov_off, SEEK_SET) != 0 || fread(moov, 1, (size_t)moov_len, f) != moov_len) goto fail_msg;
  {
    view_t mv = {moov, (size_t)moov_len}, trak, mdia, mdhd, hdlr, minf, stbl;
    size_t p = 0;
    uint32_t t;
    int found = 0;
    while (next_atom(mv, &p, &t, &trak)) {
      if (t != MI_QT_FOURCC('t', 'r', 'a', 'k')) continue;
      if (!find(trak, MI_QT_FOURCC('m', 'd', 'i', 'a'), &mdia) || !find(mdia, MI_QT_FOURCC('h', 'd', 'l', 'r'), &hdlr) ||
          hdlr.n < 12 || rd32(hdlr.p + 8) != MI_QT_FOURCC('v', 'i', 'd', 'e'))
        continue; /* not a video track */
      if (!find(mdia, MI_QT_FOURCC('m', 'd', 'h', 'd'), &mdhd) || mdhd.n < 20 || !find(mdia, MI_QT_FOURCC('m', 'i', 'n', 'f'), &minf) ||
          !find(minf, MI_QT_FOURCC('s', 't', 'b', 'l'), &stbl)) {
        seterr(err, errlen, "video track lacks mdhd/minf/stbl");
        goto fail;
      }
      r->timescale = mdhd.p[0] == 1 ? (mdhd.n >= 32 ? rd32(mdhd.p + 20) : 0) : rd32(mdhd.p + 12);
      if (!r->timescale) {
        seterr(err, errlen, "mdhd time scale is 0");
        goto fail;
      }
      if (build_index(r, stbl, err, errlen) != MI_QT_OK) goto fail;
      found = 1;
      break;
    }
    if (!found) {
      seterr(err, errlen, "no video track");
      goto fail;
    }
  }
  free(moov);
  return r;
fail_msg:
  seterr(err, errlen, "I/O or memory error reading %s", path);
fail:
  free(moov);
  if (r) free(r->s);
  free(r);
  fclose(f);
  return NULL;
}

void mi_qt_reader_close(mi_qt_reader *r) {
  if (!r) return;
  fclose(r->f);
  free(r->s);
  free(r);
}

int mi_qt_reader_info(const mi_qt_reader *r, uint32_t *fourcc, int *width, int *height, uint32_t *timescale, uint64_t *nsamples) {
  if (!r) return MI_QT_ERR_ARG;
  if (fourcc) *fourcc = r->fourcc;
  if (width) *width = r->width;
  if (height) *height = r->height;
  if (timescale) *timescale = r->timescale;
  if (nsamples) *nsamples = r->n;
  return MI_QT_OK;
}

int mi_qt_reader_sample(const mi_qt_reader *r, uint64_t i, mi_qt_sample *s) {
  if (!r || !s || i >= r->n) return MI_QT_ERR_ARG;
  *s = r->s[i];
  return MI_QT_OK;
}

long mi_qt_reader_read(mi_qt_reader *r, uint64_t i, uint8_t *buf, size_t cap) {
  if (!r || !buf || i >= r->n) return MI_QT_ERR_ARG;
  const mi_qt_sample *s = &r->s[i];
  if (s->size > cap) return MI_QT_ERR_ARG;
  if (fseek(r->f, (long)s->offset, SEEK_SET) != 0 || fread(buf, 1, s->size, r->f) != s->size) return MI_QT_ERR_IO;
  return (long)s->size;
}
