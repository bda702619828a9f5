// dv_common.h — what the host side and the gfx950 kernels of the DV25 decoder share.
#pragma once
#include <stdint.h>

namespace midv {

// A code word of the variable-length code as the kernels see it:
//   bits 0..4   total length in bits, sign bit included
//   bits 5..11  how far the scan position moves: run + 1; 64 for "end of block" (past the last coefficient)
//   bits 12..19 amplitude (the sign is the word's last bit)
constexpr uint32_t vlc_entry(uint32_t len, uint32_t adv, uint32_t amp) { return len | (adv << 5) | (amp << 12); }

// One reconstruction table entry per (transform mode, scan position):
//   bits 16..31 multiplier with 14 fractional bits (aan(h) aan(v) / (w(h) w(v)))
//   bits 8..9   area of the scan position (picks the quantiser shift)
//   bits 0..7   byte offset of the coefficient in a lane's scratch
struct Tables {
  uint32_t lut9[512];   // by the next 9 bits: words of up to 9 bits (+ sign)
  uint32_t lut2[64];    // words that begin 11111 and go on with 0: by the 6 bits behind that (lengths 10..12)
  uint32_t tab[2][64];  // [mode][scan position]
  uint32_t shift4[24];  // [quantisation number + class offset]: four 4-bit shifts (area 0 in the low nibble), the
                        // "+ 1" of the reconstruction included; class 3 adds one more
};

constexpr int kFrameBytes = 120000, kW = 720, kH = 480, kCW = 180, kPicBytes = kW * kH * 3 / 2;
constexpr int kSegments = 270;  // video segments per frame: 10 DIF sequences of 27

// builds the tables (dv_tables.cpp); false if the code's lengths are not a complete prefix code
bool build_tables(Tables* t);

}  // namespace midv
