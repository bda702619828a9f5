// dv_tables.cpp — the constant tables of the DV25 decoder, built on the host once per instance.
//
// Normative data of the format (IEC 61834-2 / SMPTE 314M, written down from memory: the reference holds no DV pixel
// decoder, lib/dvframe.c:663-676): the (run, amplitude) pairs of the variable-length code in code order with their
// lengths, the two scan orders, the quantiser shifts by (quantisation number + class offset, area), the weights.
// The checker under the test suite states the same data independently; tests/test_dv_oracle.py compares what the two
// make of it.
#include "dv_common.h"

#include <math.h>
#include <string.h>

namespace midv {
namespace {

// lengths without the sign bit; run 255 = end of block
const uint8_t kLen[] = {2, 3, 4, 4,  4,  4,  5,  5,  5,  5,  6,  6,  6,  6,  7,  7,  7,  7,  7,  7,  7,  7,  8,  8,  8,  8,  8,  8,  8,  8,
                        8, 8, 8, 8,  8,  8,  8,  8,  9,  9,  9,  9,  9,  9,  9,  9,  9,  9,  9,  9,  9,  9,  9,  9,  10, 10, 10, 10, 10, 10,
                        10, 11, 11, 11, 11, 11, 11, 11, 11, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12};
const uint8_t kRun[] = {0, 0, 255, 1, 0, 0, 2, 1, 0, 0, 3, 4, 0, 0, 5, 6, 2, 1, 1,  0,  0,  0,  7, 8, 9, 10, 3, 4, 2, 1,
                        1, 1, 0,   0, 0, 0, 0, 0, 11, 12, 13, 14, 5, 6, 3, 4, 2, 2,  1,  0,  0,  0,  0, 0, 5, 3,  3, 2, 1, 1,
                        1, 0, 1,   6, 4, 3, 1, 1, 1, 2, 3, 4, 5, 7, 8, 9, 10, 7, 8,  4,  3,  2,  2,  2, 2, 2, 1,  1, 1};
const uint8_t kAmp[] = {1, 2,  0,  1,  3,  4,  1,  2, 5, 6, 1, 1, 7, 8, 1, 1, 2, 3, 4,  9,  10, 11, 1,  1,  1, 1, 2, 2, 3, 5,
                        6, 7,  12, 13, 14, 15, 16, 17, 1, 1, 1, 1, 2, 2, 3, 3, 4, 5, 8,  18, 19, 20, 21, 22, 3, 4, 5, 6, 9, 10,
                        11, 0, 0,  3,  4,  6,  12, 13, 14, 0, 0, 0, 0, 2, 2, 2, 2, 3, 3,  5,  7,  7,  8,  9,  10, 11, 15, 16, 17};
constexpr int kShort = (int)sizeof kLen;
static_assert(sizeof kRun == sizeof kLen && sizeof kAmp == sizeof kLen, "one entry per code word");

const uint8_t kScan88[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kScan248[64] = {0,  8,  1,  9,  16, 24, 2,  10, 17, 25, 32, 40, 48, 56, 33, 41, 18, 26, 3,  11, 4,  12,
                              19, 27, 34, 42, 49, 57, 50, 58, 35, 43, 20, 28, 5,  13, 6,  14, 21, 29, 36, 44, 51, 59,
                              52, 60, 37, 45, 22, 30, 7,  15, 23, 31, 38, 46, 53, 61, 54, 62, 39, 47, 55, 63};
const uint8_t kQuantShift[22][4] = {{3, 3, 4, 4}, {3, 3, 4, 4}, {2, 3, 3, 4}, {2, 3, 3, 4}, {2, 2, 3, 3}, {2, 2, 3, 3},
                                    {1, 2, 2, 3}, {1, 2, 2, 3}, {1, 1, 2, 2}, {1, 1, 2, 2}, {0, 1, 1, 2}, {0, 1, 1, 2},
                                    {0, 0, 1, 1}, {0, 0, 1, 1}, {0, 0, 0, 1}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
                                    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

}  // namespace

bool build_tables(Tables* t) {
  memset(t, 0, sizeof *t);
  // ---- the variable-length code: canonical code words from the lengths ----
  uint32_t code = 0, prev = kLen[0];
  uint64_t kraft = 0;  // in units of 2^-15
  for (int i = 0; i < kShort; i++) {
    const uint32_t len = kLen[i], run = kRun[i], amp = kAmp[i];
    code <<= (len - prev);
    prev = len;
    kraft += 1ull << (15 - len);
    const uint32_t has_sign = run != 255u && amp != 0u;
    const uint32_t e = vlc_entry(len + has_sign, run == 255u ? 64u : run + 1u, amp);
    if (len <= 9) {
      for (uint32_t r = 0; r < (1u << (9 - len)); r++) t->lut9[(code << (9 - len)) | r] = e;
    } else {  // 11111 0 ...: by the bits behind the five ones
      const uint32_t tail = code & ((1u << (len - 5)) - 1u);
      if (code >> (len - 5) != 31u || (tail << (12 - len)) >= 64u) return false;
      for (uint32_t r = 0; r < (1u << (12 - len)); r++) t->lut2[(tail << (12 - len)) | r] = e;
    }
    code++;
  }
  // then 64 run escapes of 13 bits (1111110 rrrrrr) and 256 amplitude escapes of 15 bits + sign (1111111 aaaaaaaa s):
  // the kernels decode those from their bits
  code <<= 13 - prev;
  if (code != 0x1F80u) return false;
  kraft += 64ull << 2;
  kraft += 256ull;
  if (kraft != 1ull << 15) return false;
  // (the kernels take a zero in the first table for "11111....: look in the second table or at the escape's own bits")
  for (uint32_t i = 0; i < 512u; i++)
    if ((t->lut9[i] == 0u) != (i >> 4 == 31u)) return false;
  // ---- reconstruction: multiplier, area, where the coefficient goes ----
  const double pi = 3.14159265358979323846;
  double cs[8], w[8], aan[8];
  for (int m = 0; m < 8; m++) cs[m] = cos(m * pi / 16);
  w[0] = 1; w[1] = cs[4] / (4 * cs[7] * cs[2]); w[2] = cs[4] / (2 * cs[6]); w[3] = 1 / (2 * cs[5]);
  w[4] = 7.0 / 8; w[5] = cs[4] / cs[3]; w[6] = cs[4] / cs[2]; w[7] = cs[4] / cs[1];
  aan[0] = 1;
  for (int m = 1; m < 8; m++) aan[m] = cs[m] * sqrt(2.0);
  for (int mode = 0; mode < 2; mode++)
    for (int k = 0; k < 64; k++) {
      const int nat = (mode ? kScan248 : kScan88)[k], r = nat >> 3, h = nat & 7;
      const int v = mode ? 2 * (r >> 1) : r;  // 2-4-8: the 4-point transform is the even half of the 8-point one
      const uint32_t q = (uint32_t)floor(16384.0 * aan[h] * aan[v] / (w[h] * w[v]) + 0.5);
      if (q >= 65536u) return false;
      const uint32_t area = k < 6 ? 0u : k < 21 ? 1u : k < 43 ? 2u : 3u;
      // where the coefficient goes in a lane's scratch: columns in pairs, dword (column pair j, row r) at 8 j + r — rows
      // 0-3 / 4-7 of a pair are 16 bytes each, what the packed column pass reads (rtj_idct_pk.h)
      const uint32_t where = 4u * (8u * (uint32_t)(h >> 1) + (uint32_t)r) + 2u * (uint32_t)(h & 1);
      t->tab[mode][k] = (q << 16) | (area << 8) | where;
    }
  for (int i = 0; i < 22; i++)
    t->shift4[i] = (uint32_t)(kQuantShift[i][0] + 1) | (uint32_t)(kQuantShift[i][1] + 1) << 4 |
                   (uint32_t)(kQuantShift[i][2] + 1) << 8 | (uint32_t)(kQuantShift[i][3] + 1) << 12;
  return true;
}

}  // namespace midv
