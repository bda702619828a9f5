/*
 * compat_lite/avdec_private.h — TEST-ONLY stand-in for the handful of gavl / bgav declarations
 * that lib/video_rtjpeg.c touches, so that video_rtjpeg_mi355x.c can be compiled
 * and driven by tests/harness/plugin_harness.c in a container that has neither gavl nor the rest
 * of gmerlin-avdecoder.  It is NOT used when the file is built inside the real tree: there the
 * real <avdec_private.h> (include/avdec_private.h) is found first and this directory is not on
 * the include path.  Only the members the RTjpeg wrapper reads or writes exist here; names and
 * meanings follow include/avdec_private.h:90-118, lib/video.c:861-871 and the gavl headers.
 */
#ifndef MI_COMPAT_LITE_AVDEC_PRIVATE_H
#define MI_COMPAT_LITE_AVDEC_PRIVATE_H

#include <stdint.h>
#include <stdlib.h>

#define MI_COMPAT_LITE 1

typedef enum { GAVL_SOURCE_EOF = 0, GAVL_SOURCE_OK = 1, GAVL_SOURCE_AGAIN = 2 } gavl_source_status_t;

#define GAVL_MAX_PLANES 4
#define GAVL_YUV_420_P 0x0501 /* opaque tag here */
#define GAVL_YUV_411_P 0x0505 /* opaque tag here */
#define GAVL_META_FORMAT "Format"
#define GAVL_LOG_ERROR 1
#define GAVL_LOG_INFO 4

typedef struct { int x, y, w, h; } gavl_rectangle_i_t;

typedef struct {
  int frame_width, frame_height;
  int image_width, image_height;
  int pixelformat;
} gavl_video_format_t;

typedef struct {
  uint8_t *planes[GAVL_MAX_PLANES];
  int strides[GAVL_MAX_PLANES];
  int64_t timestamp, duration;
  uint32_t timecode;
  int dst_x, dst_y;
  gavl_rectangle_i_t src_rect;
} gavl_video_frame_t;

typedef struct { uint8_t *buf; int len; } gavl_buffer_t;

typedef struct {
  gavl_buffer_t buf;
  int64_t pts, duration;
  uint32_t timecode;
  int dst_x, dst_y;
  gavl_rectangle_i_t src_rect;
} gavl_packet_t;
typedef gavl_packet_t bgav_packet_t;

/* a dictionary as far as this wrapper looks into one: the "Format" string it sets, and integer options it reads */
#define MI_COMPAT_DICT_INTS 8
typedef struct gavl_dictionary_s {
  char format[64];
  int n_ints;
  struct { char key[32]; int val; } ints[MI_COMPAT_DICT_INTS];
  const gavl_video_format_t *vfmt; /* a stream dictionary's video format (gavl_stream_get_video_format) */
} gavl_dictionary_t;
typedef gavl_dictionary_t bgav_options_t; /* "Options are now passed as dictionary", include/avdec.h:252-254 */

/* gavl/compression.h */
#define GAVL_COMPRESSION_HAS_P_FRAMES (1 << 0)
#define GAVL_COMPRESSION_HAS_B_FRAMES (1 << 1)
typedef struct { int flags; } gavl_compression_info_t;

typedef struct bgav_stream_s bgav_stream_t;
typedef struct bgav_video_decoder_s bgav_video_decoder_t;

#define GAVL_TIME_UNDEFINED ((int64_t)0x8000000000000000LL)
#define MI_RTJ_HAVE_OUT_TIME 1 /* the real bgav_stream_t has it too (include/avdec_private.h; lib/video.c:295) */

struct bgav_stream_s {
  void *decoder_priv;
  int64_t out_time; /* timestamp the next picture is expected to have (lib/video.c:295,527-528,607) */
  gavl_video_frame_t *vframe; /* set by a decoder that owns its output frame: nocopy mode (lib/video.c:420-429) */
  uint32_t fourcc;
  gavl_dictionary_t *m;    /* stream metadata */
  gavl_dictionary_t *info; /* what .probe receives */
  const bgav_options_t *opt;   /* include/avdec_private.h:265; lib/video_v4l2_m2m.c:66 reads BGAV_OPT_VIDEOBUFFER from it */
  gavl_compression_info_t *ci; /* include/avdec_private.h:358; lib/video.c:596 tests GAVL_COMPRESSION_HAS_P_FRAMES */
  struct { struct { gavl_video_format_t *format; } video; } data;
  /* harness side: the packet queue behind bgav_stream_get_packet_read */
  void *harness;
};

/* include/avdec_private.h:90-118 */
struct bgav_video_decoder_s {
  const uint32_t *fourccs;
  const char *name;
  int (*probe)(const gavl_dictionary_t *stream);
  int (*init)(bgav_stream_t *);
  gavl_source_status_t (*decode)(bgav_stream_t *, gavl_video_frame_t *);
  void (*close)(bgav_stream_t *);
  void (*resync)(bgav_stream_t *);
  int (*skipto)(bgav_stream_t *, int64_t dest);
  bgav_video_decoder_t *next;
};

#define BGAV_MK_FOURCC(a, b, c, d) ((a << 24) | (b << 16) | (c << 8) | d) /* include/avdec_private.h:47 */

/* implemented by the harness (lib/stream.c:538-601, lib/video.c:861-871, lib/codecs.c:201-215, gavl) */
gavl_source_status_t bgav_stream_get_packet_read(bgav_stream_t *s, bgav_packet_t **p);
void bgav_stream_done_packet_read(bgav_stream_t *s, bgav_packet_t *p);
void bgav_set_video_frame_from_packet(const bgav_packet_t *p, gavl_video_frame_t *f);
void bgav_video_decoder_register(bgav_video_decoder_t *dec);
void gavl_dictionary_set_string(gavl_dictionary_t *d, const char *key, const char *val);
int gavl_dictionary_get_int(const gavl_dictionary_t *d, const char *key, int *val); /* 1 if the key is there */
gavl_video_frame_t *gavl_video_frame_create(const gavl_video_format_t *format); /* NULL: no plane memory */
void gavl_video_frame_null(gavl_video_frame_t *f);
void gavl_video_frame_destroy(gavl_video_frame_t *f);
void gavl_log(int level, const char *domain, const char *fmt, ...);
const gavl_video_format_t *gavl_stream_get_video_format(const gavl_dictionary_t *stream); /* gavl/metatags.h */

#endif
