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
 */
#include <stdlib.h>
#include <string.h>

#include <avdec_private.h>
#include <codecs.h>

#include "mi_rtjpeg.h"

#define LOG_DOMAIN "video_rtjpeg_mi355x"

#define BLOCK_SIZE 16
#define PADD(x) ((((x) + BLOCK_SIZE - 1) / BLOCK_SIZE) * BLOCK_SIZE)

typedef struct {
  mi_rtj_ctx *ctx; /* owns the persistent device picture: the role of priv->frame + priv->rtjpeg */
#ifdef MI_RTJ_NOCOPY
  gavl_video_frame_t *vframe; /* plane pointers into the instance's pinned host picture */
#endif
} rtjpeg_hip_priv_t;

/* .probe (include/avdec_private.h:95): claim the stream only if a gfx950 device is usable, so
 * that a CPU decoder registered after this one still gets it otherwise (first match wins,
 * lib/codecs.c:246-279; same arrangement as lib/video_v4l2_m2m.c). */
static int probe_rtjpeg_hip(const gavl_dictionary_t *stream) {
  (void)stream;
  return mi_rtj_device_count() > 0;
}

static int init_rtjpeg_hip(bgav_stream_t *s) {
  rtjpeg_hip_priv_t *priv = calloc(1, sizeof(*priv));
  const char *dev = getenv("MI_RTJ_DEVICE"); /* default: the process's current device */
  if (!priv) return 0;
  priv->ctx = mi_rtj_create(dev ? atoi(dev) : -1);
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
  return 1;
}

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

static void close_rtjpeg_hip(bgav_stream_t *s) {
  rtjpeg_hip_priv_t *priv = s->decoder_priv;
  if (!priv) return;
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
    .decode = decode_rtjpeg_hip,
    .close = close_rtjpeg_hip,
    /* synchronous, one packet in / one frame out: no .resync state to flush, no .skipto */
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
