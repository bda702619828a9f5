/*
 * plugin_harness.c — drives video_rtjpeg_mi355x.c exactly the way lib/video.c drives a
 * bgav_video_decoder_t (bgav_video_start: lib/video.c:375-463; read_video_copy: :279-312), with
 * the small slice of bgav/gavl it needs implemented here (registry as lib/codecs.c:201-279, packet
 * queue behind bgav_stream_get_packet_read, frame metadata copy as lib/video.c:861-871).
 *
 *   plugin_harness <packets.bin> <image_w> <image_h> <out.bin> [skip_every] [key=value ...]
 *
 *   seek=N:M      after N pictures, seek: the packet source continues at packet M and the decoder's .resync is
 *                 called, as bgav_video_resync does after a seek (lib/video.c:525-565)
 *   skipto=N:T    after N pictures, bgav_video_skipto(T) the way it runs for streams with keyframes: the decoder's
 *                 .skipto if it has one, else decode-and-drop (lib/video.c:633-660)
 *   skippkts=N:T  after N pictures, bgav_video_skipto(T) the way it runs for intra-only streams: packets are
 *                 skipped at the source, s->out_time is set to the first one kept (lib/video.c:596-612)
 *   repeat=R      play the packet list R times (time stamps keep counting); bench=1: write no pictures, print
 *                 {"frames":..,"seconds":..,"fps":..} on stdout (the end-to-end figure of bench.py)
 *
 * packets.bin: repeated { u32 le length, bytes }.  out.bin: for every decoded frame the cropped
 * planes Y (w*h), U, V ((w+1)/2*(h+1)/2 each), tightly packed, followed by 8 bytes pts (le).
 * Exit codes: 0 ok, 3 no decoder accepted the stream (e.g. no GPU), 4 init failed, 1 usage/io.
 */
#include <stdarg.h>
#include <stdio.h>
#include "mi_qtrtj.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <avdec_private.h>
#include <codecs.h>

/* ---- registry (lib/codecs.c:201-215, 246-279) ---- */
static bgav_video_decoder_t *video_decoders = NULL;
void bgav_video_decoder_register(bgav_video_decoder_t *dec) {
  bgav_video_decoder_t **pp = &video_decoders;
  while (*pp) pp = &(*pp)->next;
  *pp = dec;
  dec->next = NULL;
}
static bgav_video_decoder_t *find_video_decoder(uint32_t fourcc, const gavl_dictionary_t *stream) {
  for (bgav_video_decoder_t *cur = video_decoders; cur; cur = cur->next)
    for (int i = 0; cur->fourccs[i]; i++)
      if (cur->fourccs[i] == fourcc && (!cur->probe || !stream || cur->probe(stream))) return cur;
  return NULL;
}

/* ---- packet queue ---- */
typedef struct { gavl_packet_t *pkts; int n, next, repeat, lap; gavl_packet_t cur; } queue_t;
gavl_source_status_t bgav_stream_get_packet_read(bgav_stream_t *s, bgav_packet_t **p) {
  queue_t *q = s->harness;
  if (q->next >= q->n) {
    if (q->lap + 1 >= q->repeat) return GAVL_SOURCE_EOF;
    q->lap++;
    q->next = 0;
  }
  q->cur = q->pkts[q->next++];
  q->cur.pts += (int64_t)q->lap * q->n * 40; /* laps keep the time stamps growing */
  *p = &q->cur;
  return GAVL_SOURCE_OK;
}
void bgav_stream_done_packet_read(bgav_stream_t *s, bgav_packet_t *p) { (void)s; (void)p; }
void bgav_set_video_frame_from_packet(const bgav_packet_t *p, gavl_video_frame_t *f) {
  f->timestamp = p->pts;
  f->duration = p->duration;
  f->timecode = p->timecode;
  f->dst_x = p->dst_x;
  f->dst_y = p->dst_y;
  f->src_rect = p->src_rect;
}
void gavl_dictionary_set_string(gavl_dictionary_t *d, const char *key, const char *val) {
  if (!strcmp(key, GAVL_META_FORMAT)) snprintf(d->format, sizeof d->format, "%s", val);
}
gavl_video_frame_t *gavl_video_frame_create(const gavl_video_format_t *format) {
  (void)format; /* only the NULL-format form (no plane memory) is needed here */
  return calloc(1, sizeof(gavl_video_frame_t));
}
void gavl_video_frame_null(gavl_video_frame_t *f) { memset(f->planes, 0, sizeof f->planes); }
void gavl_video_frame_destroy(gavl_video_frame_t *f) { free(f); }
void gavl_log(int level, const char *domain, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "[%s] %s: ", level == GAVL_LOG_ERROR ? "error" : "info", domain);
  vfprintf(stderr, fmt, ap);
  fputc('\n', stderr);
  va_end(ap);
}

/* packets from a QuickTime file (include/mi_qtrtj.h): the role of lib/demux_quicktime.c in front of the decoder */
static int load_mov(const char *path, queue_t *q, uint32_t *fourcc, int *w, int *h) {
  char err[256];
  mi_qt_reader *r = mi_qt_reader_open(path, err, sizeof err);
  if (!r) return fprintf(stderr, "%s: %s\n", path, err), 0;
  uint64_t n;
  uint32_t ts;
  mi_qt_reader_info(r, fourcc, w, h, &ts, &n);
  q->pkts = calloc(n ? n : 1, sizeof(gavl_packet_t));
  for (uint64_t i = 0; i < n; i++) {
    mi_qt_sample smp;
    mi_qt_reader_sample(r, i, &smp);
    gavl_packet_t *p = &q->pkts[i];
    p->buf.buf = calloc(smp.size + 64, 1);
    p->buf.len = (int)smp.size;
    if (mi_qt_reader_read(r, i, p->buf.buf, smp.size) != (long)smp.size) return fprintf(stderr, "%s: short sample %llu\n", path, (unsigned long long)i), 0;
    p->pts = smp.pts;
    p->duration = smp.duration;
    q->n++;
  }
  mi_qt_reader_close(r);
  return 1;
}

int main(int argc, char **argv) {
  if (argc < 5) return fprintf(stderr, "usage: %s packets.bin|movie.mov w h out.bin [skip_every]  (w h 0 0: from the movie)\n", argv[0]), 1;
  int iw = atoi(argv[2]), ih = atoi(argv[3]);
  int skip_every = 0, seek_at = -1, seek_to = 0, skipto_at = -1, skippkts_at = -1, bench = 0;
  long long skipto_t = 0, skippkts_t = 0;
  queue_t q = {0};
  q.repeat = 1;
  for (int i = 5; i < argc; i++) {
    if (sscanf(argv[i], "seek=%d:%d", &seek_at, &seek_to) == 2) continue;
    if (sscanf(argv[i], "skipto=%d:%lld", &skipto_at, &skipto_t) == 2) continue;
    if (sscanf(argv[i], "skippkts=%d:%lld", &skippkts_at, &skippkts_t) == 2) continue;
    if (sscanf(argv[i], "repeat=%d", &q.repeat) == 1) continue;
    if (sscanf(argv[i], "bench=%d", &bench) == 1) continue;
    skip_every = atoi(argv[i]);
  }
  uint32_t fourcc = BGAV_MK_FOURCC('R', 'T', 'J', '0');
  const size_t plen = strlen(argv[1]);
  if (plen > 4 && strcmp(argv[1] + plen - 4, ".mov") == 0) {
    int mw = 0, mh = 0;
    if (!load_mov(argv[1], &q, &fourcc, &mw, &mh)) return 1;
    if (!iw || !ih) iw = mw, ih = mh;  /* image size from the sample description (lib/demux_quicktime.c:1515-1518) */
  } else {
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return perror(argv[1]), 1;
    for (;;) {
      uint32_t len;
      if (fread(&len, 4, 1, fi) != 1) break;
      q.pkts = realloc(q.pkts, sizeof(gavl_packet_t) * (q.n + 1));
      gavl_packet_t *p = &q.pkts[q.n];
      memset(p, 0, sizeof *p);
      p->buf.buf = calloc(len + 64, 1); /* GAVL_PACKET_PADDING-style zero padding (lib/stream.c:487-492) */
      p->buf.len = (int)len;
      if (fread(p->buf.buf, 1, len, fi) != len) return fprintf(stderr, "short packet file\n"), 1;
      p->pts = 1000 + 40 * (int64_t)q.n;
      p->duration = 40;
      q.n++;
    }
    fclose(fi);
  }

  /* bgav_codecs_init -> bgav_init_video_decoders_rtjpeg (lib/codecs.c:176) */
  bgav_init_video_decoders_rtjpeg();

  gavl_video_format_t fmt = {.image_width = iw, .image_height = ih};
  gavl_dictionary_t meta = {{0}}, info = {{0}};
  bgav_stream_t s = {0};
  s.fourcc = fourcc;
  s.m = &meta;
  s.info = &info;
  s.data.video.format = &fmt;
  s.harness = &q;

  /* bgav_video_start (lib/video.c:383-405) */
  bgav_video_decoder_t *dec = find_video_decoder(s.fourcc, s.info);
  if (!dec) return fprintf(stderr, "no video decoder accepted fourcc %c%c%c%c\n", (int)(fourcc >> 24), (int)(fourcc >> 16) & 255, (int)(fourcc >> 8) & 255, (int)fourcc & 255), 3;
  if (!dec->init(&s)) return fprintf(stderr, "decoder init failed\n"), 4;
  fprintf(stderr, "decoder: %s, format %s, frame %dx%d image %dx%d\n", dec->name, meta.format, fmt.frame_width,
          fmt.frame_height, fmt.image_width, fmt.image_height);

  /* the caller's frame: gavl aligns strides; use a deliberately odd pitch */
  gavl_video_frame_t f;
  memset(&f, 0, sizeof f);
  const int cw = (iw + 1) / 2, ch = (ih + 1) / 2;
  f.strides[0] = ((iw + 63) & ~63) + 64;
  f.strides[1] = f.strides[2] = ((cw + 63) & ~63) + 64;
  f.planes[0] = malloc((size_t)f.strides[0] * ih);
  f.planes[1] = malloc((size_t)f.strides[1] * ch);
  f.planes[2] = malloc((size_t)f.strides[2] * ch);

  FILE *fo = bench ? NULL : fopen(argv[4], "wb");
  if (!fo && !bench) return perror(argv[4]), 1;
  int nframes = 0, k = 0;
  struct timespec t0, t1;
  s.out_time = q.n ? q.pkts[0].pts : 0; /* STREAM_GET_SYNC at start (lib/video.c:527-528) */
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    const gavl_video_frame_t *res;
    if (nframes == seek_at) { /* a seek: the demultiplexer repositions, then bgav_video_resync (lib/video.c:525-565) */
      seek_at = -1;
      q.next = seek_to;
      s.out_time = q.pkts[seek_to].pts;
      if (dec->resync) dec->resync(&s);
    }
    if (nframes == skippkts_at) { /* bgav_video_skipto, intra-only branch (lib/video.c:596-612) */
      skippkts_at = -1;
      while (q.next < q.n && q.pkts[q.next].pts + q.pkts[q.next].duration <= skippkts_t) q.next++;
      if (q.next < q.n) s.out_time = q.pkts[q.next].pts;
    }
    if (nframes == skipto_at) { /* bgav_video_skipto, keyframe branch (lib/video.c:633-660) */
      skipto_at = -1;
      if (dec->skipto) {
        if (!dec->skipto(&s, skipto_t)) break;
      } else {
        for (;;) { /* decode and drop until the picture that ends after the target */
          gavl_packet_t *nx = q.next < q.n ? &q.pkts[q.next] : NULL;
          if (!nx || nx->pts + nx->duration > skipto_t) break;
          if (dec->decode(&s, s.vframe ? NULL : &f) != GAVL_SOURCE_OK) break;
        }
      }
    }
    if (s.vframe) { /* read_video_nocopy (lib/video.c:253-277): decode(s, NULL), picture in s->vframe */
      if (dec->decode(&s, NULL) != GAVL_SOURCE_OK) break;
      res = s.vframe;
    } else {        /* read_video_copy (lib/video.c:279-312) */
      const int skip = skip_every && (++k % skip_every) == 0;
      if (dec->decode(&s, skip ? NULL : &f) != GAVL_SOURCE_OK) break;
      if (skip) continue;
      res = &f;
    }
    s.out_time = res->timestamp + res->duration; /* lib/video.c:274,295 */
    if (fo) {
      for (int y = 0; y < ih; y++) fwrite(res->planes[0] + (size_t)y * res->strides[0], 1, iw, fo);
      for (int pl = 1; pl < 3; pl++)
        for (int y = 0; y < ch; y++) fwrite(res->planes[pl] + (size_t)y * res->strides[pl], 1, cw, fo);
      fwrite(&res->timestamp, 8, 1, fo);
    } else { /* bench: touch the picture like a consumer would (one byte per 4 KiB page of every plane) */
      volatile unsigned acc = 0;
      for (size_t o = 0; o < (size_t)res->strides[0] * ih; o += 4096) acc += res->planes[0][o];
      for (int pl = 1; pl < 3; pl++)
        for (size_t o = 0; o < (size_t)res->strides[pl] * ch; o += 4096) acc += res->planes[pl][o];
      (void)acc;
    }
    nframes++;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (fo) fclose(fo);
  if (bench) {
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("{\"frames\": %d, \"seconds\": %.6f, \"fps\": %.1f, \"decoder\": \"%s\"}\n", nframes, sec, nframes / sec, dec->name);
  }
  dec->close(&s); /* bgav_video_stop (lib/video.c:510-514) */
  fprintf(stderr, "%d frames\n", nframes);
  return 0;
}
