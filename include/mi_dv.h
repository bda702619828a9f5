/*
 * mi_dv.h — C ABI of the MI355X (gfx950) DV25 525/60 video decoder: 120,000-byte DIF frames in, 720 x 480 4:1:1
 * pictures out.  Plain C types only; libmi_dv.so.
 *
 * Where it sits in gmerlin-avdecoder.  lib/dvframe.c:663-676 (bgav_dv_dec_get_video_packet) hands every DIF frame on
 * unchanged as a video packet; the pixels are then made by libavcodec's "dvvideo" decoder behind
 * lib/video_ffmpeg.c:556,628 (table entry :1572-1575; the fourccs 'dvc ', 'dvcp', 'dvsd', ... of lib/video.c:122-145).
 * This library replaces that step for the 525/60 25 Mbit/s profile: a decoder
 * registered for those fourccs in front of the FFmpeg one calls mi_dv_decode_frame with
 * gavl_packet_t::buf and the planes / strides of the gavl_video_frame_t (GAVL_YUV_411_P: Y, Cb, Cr).
 *
 * PARITY UNPINNED: the reference holds no DV pixel decoder to compare with and none is reachable from the build
 * container.  The arithmetic is stated in oracle/dv_oracle.c (written from the published format, from memory); the
 * GPU path reproduces THAT bit for bit.  Pictures of a real DV stream will look right only as far as that statement
 * matches the standard's tables.
 */
#ifndef MI_DV_H
#define MI_DV_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MI_DV_OK = 0, MI_DV_ERR_ARG = -1, MI_DV_ERR_HIP = -2, MI_DV_ERR_NOMEM = -3, MI_DV_ERR_FORMAT = -4 };
enum { MI_DV_FRAME_BYTES = 120000, MI_DV_WIDTH = 720, MI_DV_HEIGHT = 480, MI_DV_CHROMA_WIDTH = 180,
       MI_DV_PICTURE_BYTES = 720 * 480 * 3 / 2 };

typedef struct mi_dv_ctx mi_dv_ctx;

int mi_dv_device_count(void);                 /* gfx950 devices usable by this process */
mi_dv_ctx *mi_dv_create(int device);          /* -1: the process's current device; NULL on failure (mi_dv_last_error(NULL)) */
void mi_dv_destroy(mi_dv_ctx *c);
const char *mi_dv_last_error(const mi_dv_ctx *c);

/* device memory of the instance's device (plain device pointers; 256-byte aligned) */
void *mi_dv_dev_alloc(mi_dv_ctx *c, size_t bytes);
void mi_dv_dev_free(mi_dv_ctx *c, void *d);
int mi_dv_h2d(mi_dv_ctx *c, void *d, const void *h, size_t n); /* synchronous */
int mi_dv_d2h(mi_dv_ctx *c, void *h, const void *d, size_t n); /* synchronous */
int mi_dv_sync(mi_dv_ctx *c);

/* The hot path: n DIF frames resident in device memory (back to back, MI_DV_FRAME_BYTES each, d_frames 4-byte aligned)
 * -> n pictures (back to back, MI_DV_PICTURE_BYTES each: Y 720 x 480, Cb 180 x 480, Cr 180 x 480, tightly packed;
 * d_pics 8-byte aligned).  Queued on the instance's stream; pair with mi_dv_sync.  Any 120,000 bytes decode to
 * something (damaged frames are not detected, as in the oracle). */
int mi_dv_decode_batch(mi_dv_ctx *c, const void *d_frames, int n, void *d_pics);
/* Device time of k_dv_decode: every mi_dv_decode_batch brackets its launch with HIP events on the instance's stream;
 * this sums them over the launches since the last call (synchronises, then forgets them). */
int mi_dv_kernel_times(mi_dv_ctx *c, float *total_ms, int *launches);

/* One frame from host memory into the caller's planes (Y, Cb, Cr) with the caller's strides — what a
 * bgav_video_decoder_t::decode does with a packet and a gavl_video_frame_t.  Synchronous.  Checks that the frame
 * announces 525/60 (DSF bit clear, lib/dvframe.c:298-316) and returns MI_DV_ERR_FORMAT otherwise. */
int mi_dv_decode_frame(mi_dv_ctx *c, const uint8_t *frame, size_t len, uint8_t *const planes[3], const int strides[3]);

/* The decoder's constant tables exactly as its kernels read them (csrc/dv_common.h, struct Tables: the variable-length
 * code's look-up tables, reconstruction multipliers / areas / coefficient places for both transform modes, quantiser
 * shifts).  Host only — no device needed: the non-GPU tests compare them with what oracle/dv_oracle.c makes of the same
 * format data.  Returns the size in bytes; copies when `out` has room. */
size_t mi_dv_copy_tables(void *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
