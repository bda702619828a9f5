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
 *   warm=N        (bench) the clock starts after the first N pictures: a session allocates its buffers while its
 *                 first packets come in (like the W untimed warm-up steps of bench.py)
 *   streams=K     (bench) K streams side by side, each with its own decoder instance on its own thread, as an
 *                 application playing K files would (doc/mainpage.incl:54-55: instances may run on different threads);
 *                 the figure printed is the aggregate
 *   opt=KEY:INT   an entry of the stream's options dictionary (s->opt), e.g. opt=mi355x-depth:3
 *   ptsjump=N:D   packets N.. carry time stamps D later: a forward jump that is a property of the stream (an empty
 *                 edit, a fragment gap), nothing is skipped
 *   noci=1        the stream has no compression info (s->ci == NULL)
 *   fourcc=XXXX   the stream's fourcc (default RTJ0; "dvc " with packets of 120,000-byte DIF frames plays DV through
 *                 csrc/video_dv_mi355x.c in the harness built with -DMI_HARNESS_DV)
 *
 * skipto= follows the library: streams whose compression info says GAVL_COMPRESSION_HAS_P_FRAMES go to the decoder's
 * .skipto (or decode-and-drop), the others have their packets skipped at the source; skippkts= forces the latter.
 *
 * packets.bin: repeated { u32 le length, bytes }.  out.bin: for every decoded frame the cropped
 * planes Y (w*h), U, V ((w+1)/2*(h+1)/2 each), tightly packed, followed by 8 bytes pts (le).
 * Exit codes: 0 ok, 3 no decoder accepted the stream (e.g. no GPU), 4 init failed, 1 usage/io.
 */
#include <stdarg.h>
#include <stdio.h>
#include "mi_qtrtj.h"
#include <pthread.h>
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
int gavl_dictionary_get_int(const gavl_dictionary_t *d, const char *key, int *val) {
  for (int i = 0; d && i < d->n_ints; i++)
    if (!strcmp(d->ints[i].key, key)) return *val = d->ints[i].val, 1;
  return 0;
}
gavl_video_frame_t *gavl_video_frame_create(const gavl_video_format_t *format) {
  (void)format; /* only the NULL-format form (no plane memory) is needed here */
  return calloc(1, sizeof(gavl_video_frame_t));
}
void gavl_video_frame_null(gavl_video_frame_t *f) { memset(f->planes, 0, sizeof f->planes); }
void gavl_video_frame_destroy(gavl_video_frame_t *f) { free(f); }
const gavl_video_format_t *gavl_stream_get_video_format(const gavl_dictionary_t *stream) { return stream ? stream->vfmt : NULL; }
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

/* what one stream is asked to do, and what came of it */
typedef struct {
  const gavl_packet_t *pkts;
  int n, repeat, iw, ih, skip_every, seek_at, seek_to, skipto_at, skippkts_at, bench, warm, noci;
  long long skipto_t, skippkts_t;
  uint32_t fourcc;
  const gavl_dictionary_t *opt;
  const char *out_path;
  int nframes, rc;
  double t_start, t_end; /* CLOCK_MONOTONIC seconds: first decode call .. last picture (init and close are not timed) */
  char name[96], format[64];
  int frame_w, frame_h;
} play_t;

static void *play(void *arg) {
  play_t *c = arg;
  queue_t q = {0};
  q.pkts = (gavl_packet_t *)c->pkts;
  q.n = c->n;
  q.repeat = c->repeat;
  const int iw = c->iw, ih = c->ih;
  int seek_at = c->seek_at, skipto_at = c->skipto_at, skippkts_at = c->skippkts_at;

  gavl_video_format_t fmt = {.image_width = iw, .image_height = ih};
  gavl_dictionary_t meta, info;
  memset(&meta, 0, sizeof meta);
  memset(&info, 0, sizeof info);
  info.vfmt = &fmt; /* the demultiplexer knows the picture size before a decoder is chosen */
  gavl_compression_info_t ci = {0}; /* RTJ0 in a QuickTime file whose samples are all sync samples: intra-only
                                       (lib/demux_quicktime.c:1522-1525) until the decoder says otherwise */
  bgav_stream_t s = {0};
  s.fourcc = c->fourcc;
  s.m = &meta;
  s.info = &info;
  s.opt = c->opt;
  s.ci = c->noci ? NULL : &ci;
  s.data.video.format = &fmt;
  s.harness = &q;

  /* bgav_video_start (lib/video.c:383-405) */
  bgav_video_decoder_t *dec = find_video_decoder(s.fourcc, s.info);
  if (!dec) return c->rc = 3, NULL;
  if (!dec->init(&s)) return c->rc = 4, NULL;
  snprintf(c->name, sizeof c->name, "%s", dec->name);
  snprintf(c->format, sizeof c->format, "%s", meta.format);
  c->frame_w = fmt.frame_width;
  c->frame_h = fmt.frame_height;

  /* the caller's frame: gavl aligns strides; use a deliberately odd pitch */
  gavl_video_frame_t f;
  memset(&f, 0, sizeof f);
  /* chroma planes by the pixel format the decoder announced: 4:2:0 (RTjpeg) or 4:1:1 (DV 525/60) */
  const int cw = fmt.pixelformat == GAVL_YUV_411_P ? iw / 4 : (iw + 1) / 2, ch = fmt.pixelformat == GAVL_YUV_411_P ? ih : (ih + 1) / 2;
  f.strides[0] = ((iw + 63) & ~63) + 64;
  f.strides[1] = f.strides[2] = ((cw + 63) & ~63) + 64;
  f.planes[0] = malloc((size_t)f.strides[0] * ih);
  f.planes[1] = malloc((size_t)f.strides[1] * ch);
  f.planes[2] = malloc((size_t)f.strides[2] * ch);

  FILE *fo = c->bench ? NULL : fopen(c->out_path, "wb");
  if (!fo && !c->bench) return perror(c->out_path), c->rc = 1, NULL;
  int nframes = 0, k = 0;
  struct timespec ts;
  s.out_time = q.n ? q.pkts[0].pts : 0; /* STREAM_GET_SYNC at start (lib/video.c:527-528) */
  clock_gettime(CLOCK_MONOTONIC, &ts);
  c->t_start = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
  for (;;) {
    const gavl_video_frame_t *res;
    if (nframes == seek_at) { /* a seek: the demultiplexer repositions, then bgav_video_resync (lib/video.c:525-565) */
      seek_at = -1;
      q.next = c->seek_to;
      s.out_time = q.pkts[c->seek_to].pts;
      if (dec->resync) dec->resync(&s);
    }
    int source_skip = 0;
    long long target = 0;
    if (nframes == skippkts_at) source_skip = 1, target = c->skippkts_t, skippkts_at = -1;
    if (nframes == skipto_at) { /* bgav_video_skipto (lib/video.c:579-660) */
      skipto_at = -1;
      target = c->skipto_t;
      if (!(ci.flags & GAVL_COMPRESSION_HAS_P_FRAMES)) {
        source_skip = 1;
      } else if (dec->skipto) {
        if (!dec->skipto(&s, target)) break;
      } else {
        for (;;) { /* decode and drop until the picture that ends after the target */
          gavl_packet_t *nx = q.next < q.n ? &q.pkts[q.next] : NULL;
          if (!nx || nx->pts + nx->duration > target) break;
          if (dec->decode(&s, s.vframe ? NULL : &f) != GAVL_SOURCE_OK) break;
        }
      }
    }
    if (source_skip) { /* the intra-only branch (lib/video.c:596-612): packets are skipped at the source */
      while (q.next < q.n && q.pkts[q.next].pts + q.pkts[q.next].duration <= target) q.next++;
      if (q.next < q.n) s.out_time = q.pkts[q.next].pts;
    }
    if (s.vframe) { /* read_video_nocopy (lib/video.c:253-277): decode(s, NULL), picture in s->vframe */
      if (dec->decode(&s, NULL) != GAVL_SOURCE_OK) break;
      res = s.vframe;
    } else {        /* read_video_copy (lib/video.c:279-312) */
      const int skip = c->skip_every && (++k % c->skip_every) == 0;
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
    } else if (!getenv("MI_RTJ_HARNESS_NO_TOUCH")) { /* bench: touch the picture like a consumer would (one byte per 4 KiB page of every plane) */
      volatile unsigned acc = 0;
      for (size_t o = 0; o < (size_t)res->strides[0] * ih; o += 4096) acc += res->planes[0][o];
      for (int pl = 1; pl < 3; pl++)
        for (size_t o = 0; o < (size_t)res->strides[pl] * ch; o += 4096) acc += res->planes[pl][o];
      (void)acc;
    }
    nframes++;
    if (c->bench && nframes == c->warm) { /* warm=N: the first N pictures (the session's allocations) are not timed */
      clock_gettime(CLOCK_MONOTONIC, &ts);
      c->t_start = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &ts);
  c->t_end = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
  if (fo) fclose(fo);
  dec->close(&s); /* bgav_video_stop (lib/video.c:510-514) */
  free(f.planes[0]);
  free(f.planes[1]);
  free(f.planes[2]);
  c->nframes = c->bench && nframes > c->warm ? nframes - c->warm : nframes;
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 5) return fprintf(stderr, "usage: %s packets.bin|movie.mov w h out.bin [skip_every]  (w h 0 0: from the movie)\n", argv[0]), 1;
  int iw = atoi(argv[2]), ih = atoi(argv[3]);
  int streams = 1, jump_at = -1;
  uint32_t fourcc_arg = 0;
  long long jump_by = 0;
  play_t cfg = {0};
  gavl_dictionary_t opt;
  memset(&opt, 0, sizeof opt);
  queue_t q = {0};
  cfg.repeat = 1;
  cfg.seek_at = cfg.skipto_at = cfg.skippkts_at = -1;
  for (int i = 5; i < argc; i++) {
    char key[32];
    int val;
    if (sscanf(argv[i], "seek=%d:%d", &cfg.seek_at, &cfg.seek_to) == 2) continue;
    if (sscanf(argv[i], "skipto=%d:%lld", &cfg.skipto_at, &cfg.skipto_t) == 2) continue;
    if (sscanf(argv[i], "skippkts=%d:%lld", &cfg.skippkts_at, &cfg.skippkts_t) == 2) continue;
    if (sscanf(argv[i], "repeat=%d", &cfg.repeat) == 1) continue;
    if (sscanf(argv[i], "bench=%d", &cfg.bench) == 1) continue;
    if (sscanf(argv[i], "warm=%d", &cfg.warm) == 1) continue;
    if (sscanf(argv[i], "streams=%d", &streams) == 1) continue;
    if (sscanf(argv[i], "ptsjump=%d:%lld", &jump_at, &jump_by) == 2) continue;
    if (sscanf(argv[i], "noci=%d", &cfg.noci) == 1) continue;
    if (strncmp(argv[i], "fourcc=", 7) == 0 && strlen(argv[i]) == 11) {
      const unsigned char *c4 = (const unsigned char *)argv[i] + 7;
      fourcc_arg = ((uint32_t)c4[0] << 24) | ((uint32_t)c4[1] << 16) | ((uint32_t)c4[2] << 8) | c4[3];
      continue;
    }
    if (sscanf(argv[i], "opt=%31[^:]:%d", key, &val) == 2 && opt.n_ints < MI_COMPAT_DICT_INTS) {
      snprintf(opt.ints[opt.n_ints].key, sizeof opt.ints[0].key, "%s", key);
      opt.ints[opt.n_ints++].val = val;
      continue;
    }
    cfg.skip_every = atoi(argv[i]);
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
  for (int i = 0; i < q.n; i++)
    if (jump_at >= 0 && i >= jump_at) q.pkts[i].pts += jump_by;
  if (streams < 1 || streams > 16 || (streams > 1 && !cfg.bench)) return fprintf(stderr, "streams=1..16, more than one only with bench=1\n"), 1;

  if (fourcc_arg) fourcc = fourcc_arg;
  /* bgav_codecs_init -> bgav_init_video_decoders_rtjpeg (lib/codecs.c:176) */
#ifdef MI_HARNESS_DV
  bgav_init_video_decoders_dv_mi355x(); /* in front of the library's other decoders */
#endif
  bgav_init_video_decoders_rtjpeg();

  cfg.pkts = q.pkts;
  cfg.n = q.n;
  cfg.iw = iw;
  cfg.ih = ih;
  cfg.fourcc = fourcc;
  cfg.opt = &opt;
  cfg.out_path = argv[4];
  play_t run[16];
  pthread_t th[16];
  for (int i = 0; i < streams; i++) {
    run[i] = cfg;
    if (streams == 1) play(&run[i]);
    else if (pthread_create(&th[i], NULL, play, &run[i])) return fprintf(stderr, "pthread_create failed\n"), 1;
  }
  for (int i = 0; i < streams && streams > 1; i++) pthread_join(th[i], NULL);
  int nframes = 0;
  double t0 = 0, t1 = 0;
  for (int i = 0; i < streams; i++) {
    if (run[i].rc == 3) return fprintf(stderr, "no video decoder accepted fourcc %c%c%c%c\n", (int)(fourcc >> 24), (int)(fourcc >> 16) & 255, (int)(fourcc >> 8) & 255, (int)fourcc & 255), 3;
    if (run[i].rc == 4) return fprintf(stderr, "decoder init failed\n"), 4;
    if (run[i].rc) return run[i].rc;
    nframes += run[i].nframes;
    if (i == 0 || run[i].t_start < t0) t0 = run[i].t_start; /* the aggregate: first decode call of any stream .. */
    if (i == 0 || run[i].t_end > t1) t1 = run[i].t_end;     /* .. last picture of any stream */
  }
  fprintf(stderr, "decoder: %s, format %s, frame %dx%d image %dx%d\n", run[0].name, run[0].format, run[0].frame_w,
          run[0].frame_h, iw, ih);
  if (cfg.bench) {
    const double sec = t1 - t0;
    printf("{\"frames\": %d, \"seconds\": %.6f, \"fps\": %.1f, \"streams\": %d, \"decoder\": \"%s\"}\n", nframes, sec,
           nframes / sec, streams, run[0].name);
  }
  fprintf(stderr, "%d frames\n", nframes);
  return 0;
}
