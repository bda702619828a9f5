/*
 * rtj_oracle.h — CPU oracle for the RTjpeg hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gmerlin-avdecoder_amd/ may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker.
 *
 * This is an independent scalar restatement of the arithmetic in the
 * reference's lib/RTjpeg.c (file:line cited per function in rtj_oracle.c).
 * Parity status: PINNED — tests/test_oracle_vs_reference.py drives the
 * reference's own lib/RTjpeg.c (compiled by oracle/Makefile into
 * oracle/_ref/librtjpeg_ref.so, never copied) against this restatement, and
 * tests/golden/ holds vectors produced by that reference build.
 */
#ifndef RTJ_ORACLE_H
#define RTJ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTJO_HEADER_SIZE 12

/* Dequant/quant tables for one quality value (RTjpeg.c:2344-2369, 277-286, 1208-1217). */
typedef struct {
  int32_t lqt[64], cqt[64];   /* forward (encoder) tables, AAN-scaled   */
  int32_t liqt[64], ciqt[64]; /* inverse (decoder) tables, AAN-scaled   */
  int lb8, cb8;               /* number of leading "8-bit" zig-zag ACs  */
} rtjo_tables;

/* Q is clamped to 1..255 like RTjpeg_set_quality (RTjpeg.c:2408-2419). */
void rtjo_make_tables(int Q, rtjo_tables *t);

/* ---- decoder (RTjpeg_decompress, RTjpeg.c:3565-3586) ---- */
typedef struct rtjo_dec rtjo_dec;
rtjo_dec *rtjo_dec_new(void);
void rtjo_dec_free(rtjo_dec *d);
/*
 * Decode one packet into contiguous planes (Y stride = width, U/V stride =
 * width/2), exactly as RTjpeg_decompress does for format YUV420.  Blocks whose
 * first byte is 0xFF leave their 8x8 destination untouched.  Bytes at or past
 * `len` read as 0 (the reference has no bound; gavl zero-pads packets).
 * Returns the number of packet bytes consumed (header included), or -1 when
 * the header's width/height are not positive multiples of 16 (the reference
 * does not terminate on such input, RTjpeg.c:2701).
 */
long rtjo_decode(rtjo_dec *d, const uint8_t *pkt, size_t len,
                 uint8_t *y, uint8_t *u, uint8_t *v);
/* Byte offset (from packet start) of every block start plus a final
 * end-of-stream entry: offs[0..nblocks].  Uses and updates the same header
 * state as rtjo_decode.  Returns nblocks or -1. */
long rtjo_block_offsets(rtjo_dec *d, const uint8_t *pkt, size_t len, uint32_t *offs);
/* Effective table index the decoder used for the last packet: 0 = the
 * never-initialised all-zero tables (Q field 0 on a fresh decoder), else 1..255. */
int rtjo_dec_quality(const rtjo_dec *d);

/* ---- encoder (RTjpeg_compress, RTjpeg.c:3488-3524) — stream generator ---- */
typedef struct rtjo_enc rtjo_enc;
/* key_rate/lmask/cmask as RTjpeg_set_intra (RTjpeg.c:2455-2488); key_rate 0 = intra only. */
rtjo_enc *rtjo_enc_new(int width, int height, int Q, int key_rate, int lmask, int cmask);
void rtjo_enc_free(rtjo_enc *e);
/* Worst case output: 12 + 64 bytes per block.  Returns packet length. */
long rtjo_encode(rtjo_enc *e, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                 uint8_t *out);

/* ---- single-block known-answer helpers ---- */
/* stream bytes -> dequantised coefficients (RTjpeg_s2b, RTjpeg.c:157-186); returns bytes consumed */
int rtjo_s2b(const uint8_t *strm, size_t avail, int bt8, const int32_t *qtbl, int16_t coef[64]);
/* coefficients -> 8x8 pixels at stride `stride` (RTjpeg_idct C path, RTjpeg.c:2209-2332) */
void rtjo_idct(const int16_t coef[64], uint8_t *dst, int stride);

/* ---- colour stage: RTjpeg_yuv420rgb32/bgr32/rgb24/bgr24/rgb16 (RTjpeg.c:3123-3475) ----
 * fmt 0..4 in that order.  dst rows are `pitch` bytes apart; the 32-bit formats leave byte 3 of every
 * pixel untouched, as the reference does. */
void rtjo_yuv420_to_rgb(int fmt, int w, int h, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                        uint8_t *dst, size_t pitch);

#ifdef __cplusplus
}
#endif
#endif
