/*
 * video_dv_mi355x.c — a bgav_video_decoder_t for DV25 525/60 (NTSC) video on an MI355X.
 *
 * In gmerlin-avdecoder DV pixels are libavcodec's: lib/dvframe.c:663-676 hands each 120,000-byte DIF frame on as a
 * video packet, and the "FFmpeg DV decoder" entry of lib/video_ffmpeg.c:1572-1575 decodes it for the fourccs of
 * lib/video.c:122-145 (bgav_dv_fourccs).  This file registers a decoder for the same fourccs that is asked FIRST
 * (first match wins, lib/codecs.c:246-279): its .probe accepts a stream only when a gfx950 device is usable and the
 * stream is 720 x 480, so every other DV flavour (625/50, DVCPRO50, DVCPRO HD) and every host without the device still
 * go to the FFmpeg decoder.  The pixels come from include/mi_dv.h.
 *
 * Integration: add this file to lib/Makefile.am, declare
 * bgav_init_video_decoders_dv_mi355x() in include/codecs.h and call it in bgav_codecs_init (lib/codecs.c:160-200) BEFORE
 * bgav_init_video_decoders_ffmpeg().  Host code stays C.
 *
 * Shape: lib/video_rtjpeg.c's own — synchronous, one packet in, one picture out, into the caller's frame with the
 * caller's strides (copy mode; a skipped frame, f == NULL, consumes its packet and decodes nothing).
 *
 * PARITY UNPINNED: see include/mi_dv.h.
 */
#include <stdlib.h>
#include <string.h>

#include <avdec_private.h>
#include <codecs.h>

#include "mi_dv.h"

#define LOG_DOMAIN "video_dv_mi355x"

typedef struct {
  mi_dv_ctx *ctx;
} dv_hip_priv_t;

/* the fourccs of lib/video.c:122-145; inside the tree the library's own array is used */
#ifdef MI_COMPAT_LITE
static const uint32_t dv_fourccs[] = {
    BGAV_MK_FOURCC('d', 'v', 's', 'd'), BGAV_MK_FOURCC('D', 'V', 'S', 'D'), BGAV_MK_FOURCC('d', 'v', 'h', 'd'),
    BGAV_MK_FOURCC('d', 'v', 's', 'l'), BGAV_MK_FOURCC('d', 'v', '2', '5'), BGAV_MK_FOURCC('D', 'V', ' ', ' '),
    BGAV_MK_FOURCC('d', 'v', 'c', 'p'), BGAV_MK_FOURCC('d', 'v', 'c', ' '), BGAV_MK_FOURCC('d', 'v', 'p', 'p'),
    BGAV_MK_FOURCC('A', 'V', 'd', 'v'), BGAV_MK_FOURCC('A', 'V', 'd', '1'), 0x00};
#define DV_FOURCCS dv_fourccs
#else
#define DV_FOURCCS bgav_dv_fourccs /* include/avdec_private.h:1435 */
#endif

static int is_525_60(const gavl_video_format_t *fmt) { return fmt && fmt->image_width == MI_DV_WIDTH && fmt->image_height == MI_DV_HEIGHT; }

/* .probe (include/avdec_private.h:95): only what this decoder can do, so that the FFmpeg decoder registered behind it
 * gets everything else */
static int probe_dv_hip(const gavl_dictionary_t *stream) {
  if (mi_dv_device_count() <= 0) return 0;
  return is_525_60(gavl_stream_get_video_format(stream));
}

static int init_dv_hip(bgav_stream_t *s) {
  dv_hip_priv_t *priv;
  if (!is_525_60(s->data.video.format)) { /* (a caller that skipped .probe) */
    gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Only 525/60 25 Mbit/s DV (720x480) is decoded on the MI355X");
    return 0;
  }
  priv = calloc(1, sizeof(*priv));
  if (!priv) return 0;
  priv->ctx = mi_dv_create(-1);
  if (!priv->ctx) {
    gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Cannot open MI355X DV decoder: %s", mi_dv_last_error(NULL));
    free(priv);
    return 0;
  }
  s->decoder_priv = priv;
  s->data.video.format->frame_width = MI_DV_WIDTH;
  s->data.video.format->frame_height = MI_DV_HEIGHT;
  s->data.video.format->pixelformat = GAVL_YUV_411_P; /* lib/dvframe.c:119: the 525/60 profile's pix_fmt */
  gavl_dictionary_set_string(s->m, GAVL_META_FORMAT, "DV");
  return 1;
}

static gavl_source_status_t decode_dv_hip(bgav_stream_t *s, gavl_video_frame_t *f) {
  dv_hip_priv_t *priv = s->decoder_priv;
  bgav_packet_t *p = NULL;
  gavl_source_status_t st;
  if ((st = bgav_stream_get_packet_read(s, &p)) != GAVL_SOURCE_OK) return st;
  if (!f) { /* skip frame: the packet is consumed, nothing is decoded (every DV frame is a key frame) */
    bgav_stream_done_packet_read(s, p);
    return GAVL_SOURCE_OK;
  }
  if (mi_dv_decode_frame(priv->ctx, p->buf.buf, (size_t)p->buf.len, (uint8_t *const *)f->planes, f->strides) != MI_DV_OK) {
    gavl_log(GAVL_LOG_ERROR, LOG_DOMAIN, "Decoding failed: %s", mi_dv_last_error(priv->ctx));
    bgav_stream_done_packet_read(s, p);
    return GAVL_SOURCE_EOF; /* never abort: errors are EOF + a log line, as everywhere in the library */
  }
  bgav_set_video_frame_from_packet(p, f);
  bgav_stream_done_packet_read(s, p);
  return GAVL_SOURCE_OK;
}

static void close_dv_hip(bgav_stream_t *s) {
  dv_hip_priv_t *priv = s->decoder_priv;
  if (!priv) return;
  mi_dv_destroy(priv->ctx);
  free(priv);
  s->decoder_priv = NULL;
}

static bgav_video_decoder_t dv_hip_decoder = {
    .name = "DV video decoder (MI355X)",
    .fourccs = DV_FOURCCS,
    .probe = probe_dv_hip,
    .init = init_dv_hip,
    .decode = decode_dv_hip,
    .close = close_dv_hip,
};

void bgav_init_video_decoders_dv_mi355x(void) { bgav_video_decoder_register(&dv_hip_decoder); }
