// rtj_common.h — structures shared by the host side and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace mirtj {

// One quality's tables as the device sees them (built on the host, rtj_tables.cpp).
// Natural (row-major) order like RTjpeg_t's liqt/ciqt/lqt/cqt (include/RTjpeg.h:40-58).
struct QTab {
  int32_t liqt[64];
  int32_t ciqt[64];
  int32_t lqt[64];
  int32_t cqt[64];
  int32_t lb8;
  int32_t cb8;
  int32_t pad[2];
};
static_assert(sizeof(QTab) == 1040, "QTab layout");

// Index 0 = the all-zero tables of a decoder that never saw a non-zero quality
// (RTjpeg_init bzero's the struct, lib/RTjpeg.c:2495-2502); 1..255 = RTjpeg_set_quality(Q).
constexpr int kNumQTab = 256;

// One packet of a plan, device view.
struct FrameDev {
  uint64_t data_off;  // offset of the first data byte (packet offset + 12) in the stream buffer
  uint64_t out_off;   // offset of the Y plane in the output buffer; U and V follow
  uint32_t data_len;  // bytes after the header; reads at or past it return 0
  uint32_t w, h;      // positive multiples of 16
  uint32_t qidx;      // row of the QTab LUT
  uint32_t blk_base;  // first entry of this frame in the block-offset index (nblk+1 entries)
  uint32_t nmb;       // macroblocks = (w/16)*(h/16)
  uint32_t mbw;       // macroblocks per row
  uint32_t nchunks;   // stream chunks of kChunk bytes covering data_len (at least 1)
  uint32_t chunk_base;  // first entry of this frame in the per-chunk arrays (nchunks+1 entries)
  uint32_t sum_base;    // first chunk of this frame in the summary array (nchunks rows of kEntries)
  uint32_t pad[2];
};
static_assert(sizeof(FrameDev) == 64, "FrameDev layout");

// Parallel block-offset index: the stream of a packet is cut into chunks of kChunk bytes.  A
// macroblock is at most 6*64 bytes, so the first macroblock that starts inside a chunk starts at
// one of kEntries offsets; a chunk's summary maps each of them to (exit offset, macroblock count).
constexpr int kChunk = 3584;
constexpr int kEntries = 384;
constexpr int kTabN = kChunk + kEntries;  // positions that get block-length tables
constexpr int kStageN = 4096;             // bytes staged per chunk (kTabN + 64 rounded up)

constexpr int kMbPerGroup = 32;  // macroblocks per decode workgroup (3 waves: Ytop, Ybottom, chroma)

// zig-zag order of the bitstream, transposed relative to JPEG (lib/RTjpeg.c:59-74)
#define MIRTJ_ZZ_INIT                                                                        \
  {0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40, 33, 26, 19, 12, 5,  6,   \
   13, 20, 27, 34, 41, 48, 56, 49, 42, 35, 28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58,  \
   51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63}

}  // namespace mirtj
