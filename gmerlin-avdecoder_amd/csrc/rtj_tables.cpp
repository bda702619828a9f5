// rtj_tables.cpp — host-side construction of the 256-row quantiser LUT the kernels read.
//
// Follows RTjpeg_calc_tbls (lib/RTjpeg.c:2344-2369), RTjpeg_dct_init (:277-286) and
// RTjpeg_idct_init (:1208-1217) as driven by RTjpeg_set_quality (:2408-2419).  The tables
// are a pure function of Q, so all of them are built once per process and kept in HBM.
#include "rtj_tables.h"

#include <string.h>

namespace mirtj {

namespace {
// quantiser bases, natural order (lib/RTjpeg.c:87-107)
const uint8_t kLum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                          14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                          18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                          49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                          24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                          99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                          99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// AAN scale factors, 32.32 fixed point (lib/RTjpeg.c:76-85); symmetric, so only the upper
// triangle is spelled out and mirrored at start-up.
const uint64_t kAanUpper[36] = {
    4294967296ULL, 5957222912ULL, 5611718144ULL, 5050464768ULL, 4294967296ULL, 3374581504ULL, 2324432128ULL, 1184891264ULL,
                   8263040512ULL, 7783580160ULL, 7005009920ULL, 5957222912ULL, 4680582144ULL, 3224107520ULL, 1643641088ULL,
                                  7331904512ULL, 6598688768ULL, 5611718144ULL, 4408998912ULL, 3036936960ULL, 1548224000ULL,
                                                 5938608128ULL, 5050464768ULL, 3968072960ULL, 2733115392ULL, 1393296000ULL,
                                                                4294967296ULL, 3374581504ULL, 2324432128ULL, 1184891264ULL,
                                                                               2651326208ULL, 1826357504ULL, 931136000ULL,
                                                                                              1258030336ULL, 641204288ULL,
                                                                                                             326894240ULL};
const uint8_t kZZ[64] = MIRTJ_ZZ_INIT;

void aan_full(uint64_t out[64]) {
  int k = 0;
  for (int i = 0; i < 8; i++)
    for (int j = i; j < 8; j++) out[8 * i + j] = out[8 * j + i] = kAanUpper[k++];
}

int32_t step_for(int Q, uint8_t base) {
  const uint64_t qual = (uint64_t)Q << 25;  // "32 bit FP, 255=2, 0=0"
  int32_t s = (int32_t)((qual / ((uint64_t)base << 16)) >> 3);
  return s ? s : 1;
}
}  // namespace

void build_qtab(int Q, QTab* t) {
  memset(t, 0, sizeof(*t));
  if (Q <= 0) return;  // row 0: never-initialised decoder, everything dequantises to 0
  if (Q > 255) Q = 255;
  uint64_t aan[64];
  aan_full(aan);
  int32_t linv[64], cinv[64];
  for (int i = 0; i < 64; i++) {
    linv[i] = 65536 / (step_for(Q, kLum[i]) << 3);
    cinv[i] = 65536 / (step_for(Q, kChr[i]) << 3);
  }
  // number of leading zig-zag ACs carried as full 8-bit values: decided on the un-scaled tables
  while (t->lb8 < 63 && linv[kZZ[t->lb8 + 1]] <= 8) t->lb8++;
  while (t->cb8 < 63 && cinv[kZZ[t->cb8 + 1]] <= 8) t->cb8++;
  for (int i = 0; i < 64; i++) {
    const int32_t lfwd = (65536 / linv[i]) >> 3, cfwd = (65536 / cinv[i]) >> 3;
    t->lqt[i] = (int32_t)(((uint64_t)lfwd << 32) / aan[i]);
    t->cqt[i] = (int32_t)(((uint64_t)cfwd << 32) / aan[i]);
    t->liqt[i] = (int32_t)(((uint64_t)linv[i] * aan[i]) >> 32);
    t->ciqt[i] = (int32_t)(((uint64_t)cinv[i] * aan[i]) >> 32);
  }
}

void build_all_qtabs(QTab* lut) {
  for (int q = 0; q < kNumQTab; q++) build_qtab(q, &lut[q]);
}

}  // namespace mirtj
