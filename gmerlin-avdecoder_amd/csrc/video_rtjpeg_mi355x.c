/*
 * video_rtjpeg_mi355x.c — drop-in replacement for gmerlin-avdecoder's lib/video_rtjpeg.c.
 *
 * Same translation-unit contract as the file it replaces: it exports
 * bgav_init_video_decoders_rtjpeg() (include/codecs.h:97, called from lib/codecs.c:176) and
 * registers a bgav_video_decoder_t for fourcc 'RTJ0' (lib/video_rtjpeg.c:103-115).  Nothing in
 * avdec.h, video.c, codecs.c or stream.c changes; bgav_open()/bgav_read_video() callers are
 * untouched.  The arithmetic (RTjpeg_decompress + the frame copy, lib/video_rtjpeg.c:81-83) runs
 * on an MI355X through the C ABI in include/mi_rtjpeg.h.
 *
 * Host code stays C.  Build inside the tree with the real <avdec_private.h>; the compat_lite/
 * headers next to this file exist only so that the test harness can compile it without gavl.
 *
 * Three flavours, one file:
 *   (no flag)            the read-ahead, frame-owning decoder: packets in flight on the device, pictures handed out in
 *                        the instance's pinned host memory (s->vframe, lib/video.c:420-441), .resync and .skipto.  This is
 *                        what a maintainer gets by default and the fastest by far (end to end).
 *   -DMI_RTJ_SYNC_NOCOPY synchronous and frame-owning: one packet in, one picture out per call.
 *   -DMI_RTJ_COPY_MODE   synchronous into the caller's frame, honouring its strides — lib/video_rtjpeg.c's own shape.
 * (-DMI_RTJ_PIPELINE / -DMI_RTJ_NOCOPY, round 2's spellings, still select the first two.)
 *
 * Options (the stream's options dictionary first, s->opt as lib/video_v4l2_m2m.c:66 reads BGAV_OPT_VIDEOBUFFER from it;
 * the environment as fallback): "mi355x-device" / MI_RTJ_DEVICE = HIP device ordinal, "mi355x-depth" / MI_RTJ_DEPTH =
 * packets in flight (2..64, default 12).
 */
#include <stdlib.h>
#include <string.h>

#include <avdec_private.h>
#include <codecs.h>

#include "mi_rtjpeg.h"

#define LOG_DOMAIN "video_rtjpeg_mi355x"

/* .skipto leaves the time of the picture it stopped at in bgav_stream_t::out_time, as the library's own loop does
 * (lib/video.c:636-655); -DMI_RTJ_NO_OUT_TIME for a tree whose stream struct lacks the member */
#if !defined(MI_RTJ_NO_OUT_TIME) && !defined(MI_RTJ_HAVE_OUT_TIME)
#define MI_RTJ_HAVE_OUT_TIME 1
#endif

#define BLOCK_SIZE 16
#define PADD(x) ((((x) + BLOCK_SIZE - 1) / BLOCK_SIZE) * BLOCK_SIZE)

#if defined(MI_RTJ_COPY_MODE)
#undef MI_RTJ_PIPELINE
#undef MI_RTJ_NOCOPY
#elif defined(MI_RTJ_SYNC_NOCOPY) || (defined(MI_RTJ_NOCOPY) && !defined(MI_RTJ_PIPELINE))
#undef MI_RTJ_PIPELINE
#ifndef MI_RTJ_NOCOPY
#define MI_RTJ_NOCOPY 1
#endif
#else /* the default: read ahead, own the frame */
#ifndef MI_RTJ_PIPELINE
#define MI_RTJ_PIPELINE 1
#endif
#ifndef MI_RTJ_NOCOPY
#define MI_RTJ_NOCOPY 1
#endif
#endif
#ifdef MI_RTJ_PIPELINE
#define MI_RTJ_MAX_DEPTH 64
#define MI_RTJ_DEFAULT_DEPTH 12 /* three groups of four: pictures leave (and packets are indexed) four at a time from 12 on */
#endif

#define MI_RTJ_OPT_DEVICE "mi355x-device" /* int: HIP device ordinal */
#define MI_RTJ_OPT_DEPTH "mi355x-depth"   /* int: packets in flight */

typedef struct {
  mi_rtj_ctx *ctx; /* owns the persistent device picture: the role of priv->frame + priv->rtjpeg */
#ifdef MI_RTJ_NOCOPY
  gavl_video_frame_t *vframe; /* plane pointers into the instance's pinned host picture */
#endif
#ifdef MI_RTJ_PIPELINE
  mi_rtj_pipe *pipe;                      /* packets in flight (include/mi_rtjpeg.h, "pipelined session") */
  bgav_packet_t meta[MI_RTJ_MAX_DEPTH];   /* pts, duration, timecode... of the packets in flight, by tag % MI_RTJ_MAX_DEPTH
                                           * (a session never has more in flight, whatever it rounds the depth to) */
  uint64_t next_tag;
  int depth;
  int eof;                                /* the packet source ran dry: hand out what is in flight, then EOF */
  int have_last;                          /* last_end is valid */
  int64_t last_end;                       /* pts + duration of the packet read last: where the next one should begin */
  uint64_t gap_tag;                       /* first packet read after a gap in the time stamps (packets skipped at the
                                           * source): everything in flight before it is stale */
  int have_gap;
#endif
} rtjpeg_hip_priv_t;

/* an integer option: the stream's options dictionary first (include/avdec_private.h:265, filled from bgav_options_t,
 * include/avdec.h:258-267), then the environment, then the default */
static int option_int(const bgav_stream_t *s, const char *key, const char *env, int dflt) {
  int v;
  const char *e;
  if (s->opt && gavl_dictionary_get_int(s->opt, key, &v)) return v;
  e = getenv(env);
  return e ? atoi(e) : dflt;
}

/* .probe (include/avdec_private.h:95): claim the stream only if a gfx950 device is usable, so
 * that a CPU decoder registered after this one still gets it otherwise (first match wins,
 * lib/codecs.c:246-279; same arrangement as lib/video_v4l2_m2m.c). */
static int probe_rtjpeg_hip(const gavl_dictionary_t *stream) {
  (void)stream;
  return mi_rtj_device_count() > 0;
}

static int init_rtjpeg_hip(bgav_stream_t *s) {
  rtjpeg_hip_priv_t *priv = calloc(1, sizeof(*priv));
  if (!priv) return 0;
  priv->ctx = mi_rtj_create(option_int(s, MI_RTJ_OPT_DEVICE, "MI_RTJ_DEVICE", -1)); /* -1: the process's current device */
  if (!priv->ctx) {
    gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Cannot open MI355X decoder: %s", mi_rtj_last_error(NULL));
    free(priv);
    return 0; /* bgav_video_start logs and aborts the start (lib/video.c:397-405) */
  }
  s->decoder_priv = priv;
  /* lib/video_rtjpeg.c:50-56 */
  s->data.video.format->frame_width = PADD(s->data.video.format->image_width);
  s->data.video.format->frame_height = PADD(s->data.video.format->image_height);
  s->data.video.format->pixelformat = GAVL_YUV_420_P;
  gavl_dictionary_set_string(s->m, GAVL_META_FORMAT, "RTjpeg");
#ifdef MI_RTJ_NOCOPY
  /* nocopy mode (lib/video.c:420-429, as lib/video_yuv.c:211 does): the decoder owns the frame, the
   * library hands s->vframe to the application and never copies it — the picture goes from the GPU
   * straight into pinned host memory and stays there until the next packet. */
  priv->vframe = gavl_video_frame_create(NULL);
  s->vframe = priv->vframe;
#endif
#ifdef MI_RTJ_PIPELINE
  {
    priv->depth = option_int(s, MI_RTJ_OPT_DEPTH, "MI_RTJ_DEPTH", MI_RTJ_DEFAULT_DEPTH);
    if (priv->depth < 2) priv->depth = 2;
    if (priv->depth > MI_RTJ_MAX_DEPTH) priv->depth = MI_RTJ_MAX_DEPTH;
    priv->pipe = mi_rtj_pipe_create(priv->ctx, priv->depth, s->data.video.format->frame_width,
                                    s->data.video.format->frame_height);
    if (!priv->pipe) {
      gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Cannot set up the decoder pipeline: %s", mi_rtj_last_error(priv->ctx));
      gavl_video_frame_null(priv->vframe);
      gavl_video_frame_destroy(priv->vframe);
      s->vframe = NULL;
      mi_rtj_destroy(priv->ctx);
      free(priv);
      s->decoder_priv = NULL;
      return 0;
    }
    /* A decoder that reads ahead cannot let the library skip packets behind its back (bgav_video_skipto's intra-only
     * branch, lib/video.c:596-612, does not say where it skipped to, and the target may lie inside what is in flight),
     * and a stream with unchanged (0xFF) blocks is not intra-only anyway: every packet has to pass through the decoder.
     * With this flag the library calls .skipto with the exact target instead (lib/video.c:614-634). */
    if (s->ci) s->ci->flags |= GAVL_COMPRESSION_HAS_P_FRAMES;
  }
#endif
  return 1;
}

#ifdef MI_RTJ_PIPELINE
/* Read ahead: packets are pulled and queued while the pipeline has room.  A packet's bytes are copied by
 * mi_rtj_pipe_submit before bgav_stream_done_packet_read (they are only valid until the next get,
 * lib/stream.c:538-601); what bgav_set_video_frame_from_packet needs later is kept under the packet's tag.
 * Returns the status of the last get (OK while the source delivers). */
static gavl_source_status_t fill_pipeline(bgav_stream_t *s) {
  rtjpeg_hip_priv_t *priv = s->decoder_priv;
  gavl_source_status_t st = GAVL_SOURCE_OK;
  while (!priv->eof && mi_rtj_pipe_room(priv->pipe) > 0) {
    bgav_packet_t *p = NULL;
    if ((st = bgav_stream_get_packet_read(s, &p)) != GAVL_SOURCE_OK) {
      if (st == GAVL_SOURCE_EOF) priv->eof = 1;
      break;
    }
    priv->meta[priv->next_tag % MI_RTJ_MAX_DEPTH] = *p;
    priv->meta[priv->next_tag % MI_RTJ_MAX_DEPTH].buf.buf = NULL; /* the bytes are not ours to keep */
    /* time stamps that jump forward: somebody MAY have skipped packets at the source (see decode_rtjpeg_pipe).  Only
     * armed for a stream without compression info: with one, init marked the stream GAVL_COMPRESSION_HAS_P_FRAMES and
     * bgav_video_skipto never skips at the source (lib/video.c:612-634 calls .skipto instead), so a forward jump is a
     * property of the stream — an empty edit, dropped-frame chunks, a fragment gap — and every picture is shown, as
     * lib/video_rtjpeg.c shows one picture per packet. */
    if (!s->ci && priv->have_last && p->duration > 0 && p->pts > priv->last_end) {
      priv->gap_tag = priv->next_tag;
      priv->have_gap = 1;
    }
    if (p->duration > 0) {
      priv->last_end = p->pts + p->duration;
      priv->have_last = 1;
    }
    if (mi_rtj_pipe_submit(priv->pipe, p->buf.buf, (size_t)p->buf.len, priv->next_tag) != MI_RTJ_OK) {
      gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Decoding failed: %s", mi_rtj_last_error(priv->ctx));
      bgav_stream_done_packet_read(s, p);
      priv->eof = 1; /* errors are EOF + a log line; pictures already in flight are still handed out */
      return GAVL_SOURCE_EOF;
    }
    priv->next_tag++;
    bgav_stream_done_packet_read(s, p);
  }
  return st;
}

static gavl_source_status_t decode_rtjpeg_pipe(bgav_stream_t *s, gavl_video_frame_t *f) {
  rtjpeg_hip_priv_t *priv = s->decoder_priv;
  const uint8_t *planes[3];
  int strides[3], i;
  uint64_t tag;
  gavl_source_status_t st;
  (void)f; /* always NULL for a frame-owning decoder (lib/video.c:262) */
  for (;;) {
    /* Should the library skip packets at the source all the same (its intra-only branch; init asks it not to), what
     * is in flight ends before the skip's target and is stale.  Two things prove that packets were skipped behind us:
     * s->out_time (set to the first packet kept, lib/video.c:607) lies past the end of the last packet read here — then
     * everything in flight goes —, or, for a stream without compression info (the only kind the library can still skip
     * packets of at the source), the time stamps of the packets read jump — then everything before the jump goes.
     * Without either nothing was skipped at the source and nothing is dropped: round 2's rule (drop what ends before
     * s->out_time) threw away pictures the caller was still waiting for when the target lay inside the read-ahead
     * window, where out_time is just the start of the next unread packet. */
#ifdef MI_RTJ_HAVE_OUT_TIME
    while (priv->have_last && s->out_time != GAVL_TIME_UNDEFINED && s->out_time > priv->last_end &&
           mi_rtj_pipe_peek_tag(priv->pipe, &tag) == MI_RTJ_OK)
      mi_rtj_pipe_next(priv->pipe, NULL, NULL, NULL, NULL, NULL);
#endif
    st = fill_pipeline(s);
    if (mi_rtj_pipe_pending(priv->pipe) == 0) return st == GAVL_SOURCE_OK ? GAVL_SOURCE_EOF : st; /* EOF or AGAIN */
    if (priv->have_gap && mi_rtj_pipe_peek_tag(priv->pipe, &tag) == MI_RTJ_OK) {
      if (tag < priv->gap_tag) {
        mi_rtj_pipe_next(priv->pipe, NULL, NULL, NULL, NULL, NULL);
        continue;
      }
      priv->have_gap = 0;
    }
    break;
  }
  if (mi_rtj_pipe_next(priv->pipe, planes, strides, NULL, NULL, &tag) != MI_RTJ_OK) {
    gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Decoding failed: %s", mi_rtj_last_error(priv->ctx));
    return GAVL_SOURCE_EOF;
  }
  for (i = 0; i < 3; i++) {
    priv->vframe->planes[i] = (uint8_t *)planes[i];
    priv->vframe->strides[i] = strides[i];
  }
  bgav_set_video_frame_from_packet(&priv->meta[tag % MI_RTJ_MAX_DEPTH], priv->vframe);
  fill_pipeline(s); /* keep the device busy while the application looks at this picture */
  return GAVL_SOURCE_OK;
}

/* .resync (include/avdec_private.h:110; called after a seek, lib/video.c:561-562): everything read ahead belongs
 * to the old position */
static void resync_rtjpeg_pipe(bgav_stream_t *s) {
  rtjpeg_hip_priv_t *priv = s->decoder_priv;
  mi_rtj_pipe_flush(priv->pipe);
  priv->eof = 0;
  priv->have_last = 0; /* the next packet's time stamp is a new beginning, not a gap */
  priv->have_gap = 0;
}

/* .skipto (include/avdec_private.h:112-115: "only needed for decoders which are not synchronous"; called by
 * bgav_video_skipto for streams with keyframes, lib/video.c:633-634): drop pictures until the next one ends after
 * `dest`; that one stays in flight for the next decode.  Every packet still goes through the decoder, as 0xFF blocks
 * need their predecessors. */
static int skipto_rtjpeg_pipe(bgav_stream_t *s, int64_t dest) {
  rtjpeg_hip_priv_t *priv = s->decoder_priv;
  for (;;) {
    uint64_t tag;
    const bgav_packet_t *m;
    gavl_source_status_t st = fill_pipeline(s);
    if (mi_rtj_pipe_pending(priv->pipe) == 0) return st == GAVL_SOURCE_AGAIN ? 1 : 0;
    mi_rtj_pipe_peek_tag(priv->pipe, &tag);
    m = &priv->meta[tag % MI_RTJ_MAX_DEPTH];
    if (m->pts + m->duration > dest) {
#ifdef MI_RTJ_HAVE_OUT_TIME
      s->out_time = m->pts;
#endif
      return 1;
    }
    mi_rtj_pipe_next(priv->pipe, NULL, NULL, NULL, NULL, NULL);
  }
}
#endif /* MI_RTJ_PIPELINE */

#ifndef MI_RTJ_PIPELINE
static gavl_source_status_t decode_rtjpeg_hip(bgav_stream_t *s, gavl_video_frame_t *f) {
  rtjpeg_hip_priv_t *priv = s->decoder_priv;
  bgav_packet_t *p = NULL;
  gavl_source_status_t st;
  int rc;

  /* We assume one frame per packet (lib/video_rtjpeg.c:69-72) */
  if ((st = bgav_stream_get_packet_read(s, &p)) != GAVL_SOURCE_OK) return st;

#ifdef MI_RTJ_NOCOPY
  /* always called with f == NULL in this mode (lib/video.c:262); picture, timestamp and duration
   * are read back from s->vframe (lib/video.c:270-274) */
  {
    const uint8_t *planes[3];
    int strides[3], i;
    (void)f;
    /* the application was told frame_width x frame_height (init): a packet whose own header says otherwise would
     * hand it planes of another size (and a 65520 x 65520 header would ask for gigabytes) — refused, like any
     * other damaged packet */
    if (p->buf.len >= 12 &&
        ((p->buf.buf[6] | (p->buf.buf[7] << 8)) != s->data.video.format->frame_width ||
         (p->buf.buf[8] | (p->buf.buf[9] << 8)) != s->data.video.format->frame_height)) {
      gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Packet header %dx%d does not match the stream's %dx%d",
               p->buf.buf[6] | (p->buf.buf[7] << 8), p->buf.buf[8] | (p->buf.buf[9] << 8),
               s->data.video.format->frame_width, s->data.video.format->frame_height);
      bgav_stream_done_packet_read(s, p);
      return GAVL_SOURCE_EOF;
    }
    rc = mi_rtj_decode_nocopy(priv->ctx, p->buf.buf, (size_t)p->buf.len, planes, strides);
    if (rc != MI_RTJ_OK) {
      gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Decoding failed: %s", mi_rtj_last_error(priv->ctx));
      bgav_stream_done_packet_read(s, p);
      return GAVL_SOURCE_EOF;
    }
    for (i = 0; i < 3; i++) {
      priv->vframe->planes[i] = (uint8_t *)planes[i];
      priv->vframe->strides[i] = strides[i];
    }
    bgav_set_video_frame_from_packet(p, priv->vframe);
    bgav_stream_done_packet_read(s, p);
    return GAVL_SOURCE_OK;
  }
#else
  /* Skip frame: the packet is consumed, nothing is decoded (lib/video_rtjpeg.c:75-79) */
  if (!f) {
    bgav_stream_done_packet_read(s, p);
    return GAVL_SOURCE_OK;
  }

  /* RTjpeg_decompress into the persistent picture + gavl_video_frame_copy of the
   * image_width x image_height region into the caller's planes/strides, in one call */
  rc = mi_rtj_decode(priv->ctx, p->buf.buf, (size_t)p->buf.len, (uint8_t *const *)f->planes, f->strides,
                     s->data.video.format->image_width, s->data.video.format->image_height);
  if (rc != MI_RTJ_OK) {
    gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Decoding failed: %s", mi_rtj_last_error(priv->ctx));
    bgav_stream_done_packet_read(s, p);
    return GAVL_SOURCE_EOF; /* never abort: errors are EOF + a log line, as everywhere in the library */
  }
  bgav_set_video_frame_from_packet(p, f);
  bgav_stream_done_packet_read(s, p);
  return GAVL_SOURCE_OK;
#endif
}

#endif /* !MI_RTJ_PIPELINE */

static void close_rtjpeg_hip(bgav_stream_t *s) {
  rtjpeg_hip_priv_t *priv = s->decoder_priv;
  if (!priv) return;
#ifdef MI_RTJ_PIPELINE
  mi_rtj_pipe_destroy(priv->pipe);
#endif
#ifdef MI_RTJ_NOCOPY
  gavl_video_frame_null(priv->vframe); /* the planes belong to the instance, not to gavl */
  gavl_video_frame_destroy(priv->vframe);
  s->vframe = NULL;
#endif
  mi_rtj_destroy(priv->ctx);
  free(priv);
  s->decoder_priv = NULL;
}

static bgav_video_decoder_t rtjpeg_hip_decoder = {
    .name = "rtjpeg video decoder (MI355X)",
    .fourccs = (uint32_t[]){BGAV_MK_FOURCC('R', 'T', 'J', '0'), 0x00},
    .probe = probe_rtjpeg_hip,
    .init = init_rtjpeg_hip,
#ifdef MI_RTJ_PIPELINE
    .decode = decode_rtjpeg_pipe,
    .resync = resync_rtjpeg_pipe,
    .skipto = skipto_rtjpeg_pipe,
#else
    .decode = decode_rtjpeg_hip,
    /* synchronous, one packet in / one frame out: no .resync state to flush, no .skipto */
#endif
    .close = close_rtjpeg_hip,
};

#ifdef MI_RTJ_KEEP_CPU_DECODER
/* When the original lib/video_rtjpeg.c is kept as the fallback, rename its registration function
 * to this name (one-line change) and build both files; see INTEGRATION.md. */
void bgav_init_video_decoders_rtjpeg_cpu(void);
#endif

void bgav_init_video_decoders_rtjpeg(void) {
  bgav_video_decoder_register(&rtjpeg_hip_decoder); /* first: wins whenever its probe accepts */
#ifdef MI_RTJ_KEEP_CPU_DECODER
  bgav_init_video_decoders_rtjpeg_cpu();
#endif
}
