/* compat_lite/codecs.h — TEST-ONLY stand-in for include/codecs.h:97 (see avdec_private.h here). */
#ifndef MI_COMPAT_LITE_CODECS_H
#define MI_COMPAT_LITE_CODECS_H
void bgav_init_video_decoders_rtjpeg(void);
void bgav_init_video_decoders_dv_mi355x(void); /* the declaration INTEGRATION.md section 6 adds to include/codecs.h */
#endif
