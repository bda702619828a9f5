/*
 * mi_dvframe.h — host-side DV DIF-frame handling, plain C, no gavl.
 *
 * What the reference's lib/dvframe.c does is demultiplexing, not pixel decoding: profile
 * detection, a memcpy of the DIF frame into a video packet, audio PCM de-shuffling and subcode
 * pack extraction.  None of it is data-parallel and none of it runs on the GPU here; it is
 * restated so that the DV rows of the scope table exist next to the RTjpeg path.  DV *pixel*
 * decoding (row D4) is libavcodec's in the reference and is not part of this repository.
 *
 * PARITY UNPINNED: lib/dvframe.c cannot be compiled in the build container (it needs gavl
 * through avdec_private.h) and the reference has no tests or vectors for it; the tests check this
 * file against an independent numpy restatement of the same lines only.
 */
#ifndef MI_DVFRAME_H
#define MI_DVFRAME_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MI_DV_PIX_411 = 0, MI_DV_PIX_420 = 1, MI_DV_PIX_422 = 2 };

/* DVprofile (lib/dvframe.c:46-74, table :106-296), the fields the in-tree code reads */
typedef struct {
  int dsf;            /* byte 3 bit 7 of the header: 0 = 525/60, 1 = 625/50 */
  int video_stype;    /* VAUX source pack stype */
  int frame_size;     /* bytes per DIF frame */
  int difseg_size;    /* DIF sequences per channel */
  int n_difchan;
  int frame_rate, frame_rate_base, ltc_divisor;
  int width, height;
  int sar[2][2];      /* {num, den} for 4:3 and 16:9 */
  int pix_fmt;        /* MI_DV_PIX_* */
  int bpm;            /* blocks per macroblock */
  int audio_stride;
  int audio_min_samples[3]; /* 48, 44.1, 32 kHz */
  const uint16_t (*audio_shuffle)[9];
} mi_dv_profile;

/* D1: dv_frame_profile (lib/dvframe.c:298-316).  frame: at least 480 bytes.  NULL if unknown. */
const mi_dv_profile *mi_dv_frame_profile(const uint8_t *frame);
int mi_dv_num_profiles(void);
const mi_dv_profile *mi_dv_profile_at(int i);
/* bgav_dv_dec_get_pixel_aspect (lib/dvframe.c:464-476) */
void mi_dv_pixel_aspect(const mi_dv_profile *p, const uint8_t *frame, int *num, int *den);

/* D2: bgav_dv_dec_get_video_packet (lib/dvframe.c:663-676): the packet IS the DIF frame.
 * Copies frame_size bytes; returns the length, sets *keyframe = 1 (every DV frame is one). */
int mi_dv_video_packet(const mi_dv_profile *p, const uint8_t *frame, uint8_t *out, int *keyframe);

/* D3: dv_extract_audio (lib/dvframe.c:545-628) with dv_audio_12to16 (:521-543).
 * pcm[0..3]: interleaved-stereo S16 buffers per channel pair (NULL when absent), little endian.
 * Returns samples per channel, 0 when the frame has no audio source pack, -1 for an unsupported
 * quantisation. */
int mi_dv_extract_audio(const mi_dv_profile *p, const uint8_t *frame, uint8_t *pcm[4]);
uint16_t mi_dv_audio_12to16(uint16_t sample);
/* audio stream parameters as bgav_dv_dec_init_audio derives them (lib/dvframe.c:424-462);
 * returns 0 when there is no audio source pack */
int mi_dv_audio_format(const mi_dv_profile *p, const uint8_t *frame, int *samplerate, int *channel_pairs,
                       int *max_samples_per_frame);

/* GetSSYBPack and its users (lib/dvframe.c:678-783): 1 when the pack was found */
int mi_dv_ssyb_pack(const mi_dv_profile *p, const uint8_t *frame, int pack_id, uint8_t pack[5]);
int mi_dv_date(const mi_dv_profile *p, const uint8_t *frame, int *year, int *month, int *day);
int mi_dv_time(const mi_dv_profile *p, const uint8_t *frame, int *hour, int *minute, int *second);
int mi_dv_timecode(const mi_dv_profile *p, const uint8_t *frame, int *hour, int *minute, int *second, int *fr);

#ifdef __cplusplus
}
#endif
#endif
