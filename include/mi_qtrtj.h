/*
 * mi_qtrtj.h — minimal QuickTime (.mov) reader and writer for RTjpeg ('RTJ0') video tracks
 * (SURVEY.md §8f row N4), plain C, host only, no gavl.
 *
 * In the reference an RTJ0 stream reaches lib/RTjpeg.c through the QuickTime demultiplexer:
 * lib/demux_quicktime.c builds a video stream from the track's sample description
 * (:1498-1520: fourcc, width and height come from the stsd entry, lib/qt_stsd.c:402-428) and its packet
 * index from the sample tables (stts lib/qt_stts.c, stsc lib/qt_stsc.c, stsz lib/qt_stsz.c:44-66,
 * stco/co64 lib/qt_stco.c, stss lib/qt_stss.c; time scale from mdhd, lib/qt_mdhd.c), and
 * lib/video_rtjpeg.c is selected by that fourcc (lib/video_rtjpeg.c:104-113).  This file is the
 * counterpart a test or tool needs on a box without gavl: it writes files with exactly those atoms and
 * reads the packets back in presentation order.  Nothing here is data-parallel; none of it runs on the GPU.
 *
 * PARITY UNPINNED: the reference's demultiplexer needs gavl (avdec_private.h) and cannot be compiled in
 * the build container, and the reference holds no QuickTime fixtures; the tests check this file against
 * an independent Python restatement of the atom layout only.
 */
#ifndef MI_QTRTJ_H
#define MI_QTRTJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_QT_FOURCC(a, b, c, d) (((uint32_t)(a) << 24) | ((uint32_t)(b) << 16) | ((uint32_t)(c) << 8) | (uint32_t)(d))
#define MI_QT_RTJ0 MI_QT_FOURCC('R', 'T', 'J', '0')

enum { MI_QT_OK = 0, MI_QT_ERR_IO = -1, MI_QT_ERR_FORMAT = -2, MI_QT_ERR_ARG = -3, MI_QT_ERR_NOMEM = -4 };

/* ---- writer: ftyp, mdat (packets back to back), moov (one video track) ---- */
typedef struct mi_qt_writer mi_qt_writer;
/* width/height: what the stsd entry and tkhd carry (the display size; RTjpeg packets carry their own
 * coded size).  timescale: units per second (mdhd); frame_duration: units per frame (one stts run). */
mi_qt_writer *mi_qt_writer_open(const char *path, int width, int height, uint32_t timescale, uint32_t frame_duration);
/* Appends one packet as one sample in a chunk of its own.  keyframe: listed in stss. */
int mi_qt_writer_add(mi_qt_writer *w, const uint8_t *pkt, uint32_t len, int keyframe);
/* Writes the moov atom, patches the mdat size, closes the file and frees the writer. */
int mi_qt_writer_close(mi_qt_writer *w);

/* ---- reader ---- */
typedef struct mi_qt_reader mi_qt_reader;
typedef struct {
  uint64_t offset;   /* file offset of the packet */
  uint32_t size;     /* bytes */
  int64_t pts;       /* in time scale units, from the stts runs */
  uint32_t duration;
  int keyframe;      /* 1 if listed in stss, or if the track has no stss (every sample is one) */
} mi_qt_sample;

/* Opens the file, walks moov and builds the packet index of the first video track.  On failure returns
 * NULL and, when err is given, a message. */
mi_qt_reader *mi_qt_reader_open(const char *path, char *err, size_t errlen);
void mi_qt_reader_close(mi_qt_reader *r);
/* fourcc/width/height of the track's first sample description, mdhd time scale, number of samples */
int mi_qt_reader_info(const mi_qt_reader *r, uint32_t *fourcc, int *width, int *height, uint32_t *timescale,
                      uint64_t *nsamples);
int mi_qt_reader_sample(const mi_qt_reader *r, uint64_t i, mi_qt_sample *s);
/* Reads packet i into buf (cap bytes); returns its size, or a negative error. */
long mi_qt_reader_read(mi_qt_reader *r, uint64_t i, uint8_t *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
