/*
 * dv_oracle.h — CPU statement of the DV25 525/60 (NTSC, 4:1:1) video codec this repository's GPU decoder is
 * checked against.  TEST INFRASTRUCTURE ONLY: nothing in the product may call it.
 *
 * PARITY UNPINNED.  The reference (gmerlin-avdecoder) holds no DV pixel decoder: lib/dvframe.c:663-676 hands the
 * 120,000-byte DIF frame to libavcodec (lib/video_ffmpeg.c:1572-1575, fourccs lib/video.c:122-145), which is not in
 * the tree, not in this container and not reachable.  What follows is written from the published format
 * (IEC 61834-2 / SMPTE 314M: DIF block layout, macroblock shuffling, 9-bit DC + mode + class header, the run/amplitude
 * variable-length code, three-pass bit redistribution inside a video segment, class / quantisation-number / area
 * quantiser shifts, coefficient weighting, 8-8 and 2-4-8 transforms) as the author remembers it — no copy of the
 * standard or of any other decoder is available offline, so tables could differ from the standard's in places and
 * nothing here can tell.  The variable-length code's LENGTHS are checked for completeness (Kraft sum exactly 1) and
 * its code words follow from them canonically.  The arithmetic of dequantisation and of the inverse transforms (fixed
 * point, defined in dv_oracle.c) is this repository's own choice: a real DV decoder is free in it, and pictures of two
 * conforming decoders differ by a level or two.  Within the repository it is normative: the GPU path must reproduce
 * it bit for bit, on streams made by the encoder below and on arbitrary bytes.
 */
#ifndef DV_ORACLE_H
#define DV_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { DVO_W = 720, DVO_H = 480, DVO_CW = 180, DVO_FRAME_BYTES = 120000, DVO_PIC_BYTES = 720 * 480 * 3 / 2 };

/* picture layout everywhere: Y 720x480, then Cb 180x480, then Cr 180x480 (gavl's GAVL_YUV_411_P plane order), tightly packed */

/* one picture -> one DIF frame (header / subcode / VAUX / audio blocks carry ids and zeros, the NTSC profile bits set).
 * flags: bit 0 = blocks with strong line-to-line differences use the 2-4-8 transform; bit 1 = vary the class by block
 * content.  Deterministic. */
void dvo_encode_frame(const uint8_t *pic, uint8_t *dif, int flags);
/* one DIF frame (any 120,000 bytes) -> one picture */
void dvo_decode_frame(const uint8_t *dif, uint8_t *pic);

/* the reconstructed coefficients (natural order, the level shift on DC) of the 30 blocks of one video segment after the
 * three passes */
void dvo_segment_coefs(const uint8_t *dif, int seq, int slot, int16_t coefs[30][64]);

/* pieces, for known-answer tests */
/* the 64 reconstruction multipliers (scan order, 14 fractional bits) of a transform mode: 0 = 8-8, 1 = 2-4-8 */
void dvo_qbase(int mode, int32_t out[64]);
/* scan position -> natural position (8 * row + column of the 8x8 coefficient array the transform reads) */
void dvo_scan(int mode, uint8_t out[64]);
/* quantiser shift of area 0..3 for quantisation number qno (0..15) and class (0..3), the class-3 doubling included */
int dvo_shift(int qno, int cls, int area);
/* (dc, mode, class, qno, levels[64] in scan order, levels[0] unused) -> 64 pixels, row major */
void dvo_block_pixels(int dc, int mode, int cls, int qno, const int16_t levels[64], uint8_t px[64]);
/* macroblock m (0..4) of video segment `slot` (0..26) of DIF sequence `seq` (0..9): x in 32-pixel units (0..22), y in
 * 8-line units (0..59; a macroblock in column 22 is 16 x 16 pixels) */
void dvo_mb_place(int seq, int slot, int m, int *x, int *y);
/* the variable-length code: for the next 16 bits of a stream (MSB first) -> total length in bits (sign included), run,
 * signed level; returns 1 for "end of block" */
int dvo_vlc_lookup(uint32_t bits16, int *len, int *run, int *level);
/* synthetic content: picture n of a seeded sequence (gradient, noise of amplitude amp, a combed band for the 2-4-8 mode) */
void dvo_synth(uint8_t *pic, int n, uint32_t seed, int amp);

/* design-note statistics (not thread safe): code words completed in pass 1 / 2 / 3 and blocks that read in each pass since
 * the last reset */
void dvo_pass_stats(long words[3], long blocks[3], int reset);

#ifdef __cplusplus
}
#endif
#endif
