// rtj_idct_pk.h — the inverse transform of RTjpeg_idct (lib/RTjpeg.c:2209-2332) on PAIRS of columns / rows, two int16
// values to a register (v_pk_add_i16 / v_pk_sub_i16, products by v_mad_i32_i16), for blocks whose values are small
// enough that 16 bits lose nothing.  k_decode is bound by the NUMBER of vector instructions it issues
// (rtj_idct_asm.h); a pair of 1-D passes costs 44 instructions here against 2 x 39 (columns) or
// 2 x 39 + descale (rows) one value to a register.
//
// Exactness.  The reference computes in int32.  Here additions and subtractions are taken modulo 2^16 — harmless as
// long as every value that is USED for more than further adding is the true value:
//   * the five multiplicands of every 1-D pass (d26, s17 - s53, d53 + d17, d53, d17): v_mad_i32_i16 forms the exact
//     32-bit product x * c + 128 of the 16-bit field, and bits 8..23 of it are MULTIPLY's result modulo 2^16;
//   * the 64 results of the row pass, which are shifted (DESCALE) and clamped.
// Each of these is a linear form of the 64 coefficients (plus rounding: at most 92.1 over both passes, from 0.5 per
// MULTIPLY).  With g = (1, 1, 1, 1.18, 1, 1.77, 2.41, 5.03) the coefficient at (row r, column c) enters none of
// them with a factor above g[r] * g[c] (tests/test_bounds.py derives the factors and the slack), so
//     sum over the block of g[r] * g[c] * |coefficient|  <=  32767 - 93 - 4        (4: DESCALE's rounding term on DC)
// keeps all of them inside int16.  pk_range_* below evaluates a rounded-up form of that sum with v_sad_u16 (two
// |.| and an add per instruction); a wave takes the packed passes when all of its blocks satisfy it, and the
// one-value-per-register passes of rtj_idct_asm.h otherwise.  (In units of DESCALE the budget is 16 times the
// pixel range: blocks of legal pictures pass, which is what the weights g are for — a flat weight of 25.3, the
// largest product, would refuse any block with a bright DC.)
//
// Register layout.  The scratch of a lane holds its block as dwords (coefficient (r, 2j) | coefficient (r, 2j+1) << 16)
// at dword index 8 j + r: `my[2j]` is rows 0-3 and `my[2j+1]` rows 4-7 of the column pair j, ready for the packed
// column pass.  Its results (row r of columns 2j, 2j+1) are regrouped by v_perm_b32 into (row r, row r+1) of one
// column for the packed row pass, whose clamped pixels v_perm_b32 sorts into the two rows' dwords.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtj_idct_asm.h"

namespace mirtj {

#ifndef MIRTJ_PK_IDCT
#define MIRTJ_PK_IDCT 1
#endif

// g[r] * g[c], times 4, rounded up to one of eight classes; one v_sad_u16 accumulator per class
// (tests/test_bounds.py checks class >= 4 g[r] max(g[2j], g[2j+1]) for every dword)
constexpr int kPkBudget = 4 * (32767 - 93 - 4);
constexpr int kPkClassW[8] = {4, 6, 9, 13, 24, 36, 49, 102};
// class of the dword (column pair j, row r)
__host__ __device__ constexpr int pk_class(int j, int r) {
  // g by row: 1 1 1 1.18 1 1.77 2.41 5.03; by column pair (the larger of the two): 1 1.18 1.77 5.03
  constexpr int rowk[8] = {0, 0, 0, 1, 0, 2, 3, 4};  // 1, 1.18, 1.77, 2.41, 5.03
  constexpr int cls[5][4] = {
      {0, 1, 2, 4},  // 1     1.18  1.77  5.03
      {1, 1, 2, 4},  // 1.18  1.39  2.08  5.93
      {2, 2, 3, 5},  // 1.77  2.08  3.12  8.88
      {3, 3, 4, 6},  // 2.41  2.85  4.26  12.14
      {4, 4, 5, 7},  // 5.03  5.93  8.88  25.27
  };
  return cls[rowk[r]][j];
}

struct IdctPK {
  int k362, k473, km669, k277;                 // scalar; the 16-bit multiplier is the low half
  uint32_t sel_m, sel_t, sel_lo, sel_hi, sadk;  // scalar: v_perm_b32 selectors, v_sad_u16's 0x8000 pair
  int c128;                                     // vector
  uint32_t c235, four;                          // vector: 235 | 235 << 16; DESCALE's 4 on the low half (column 0)
};
__device__ __forceinline__ IdctPK idct_pk_constants() {
  // sel_m: bits 8..23 of two products side by side; sel_t: bytes 0 and 2 of two registers interleaved;
  // sel_lo / sel_hi: the low / high halves of two registers side by side
  return IdctPK{362, 473, -669, 277, 0x06050201u, 0x06020400u, 0x05040100u, 0x07060302u, 0x80008000u,
                128, 0x00EB00EBu, 4u};
}

#define MIRTJ_PK_OPERANDS                                                                                        \
  [k362] MIRTJ_KREG(K.k362), [k473] MIRTJ_KREG(K.k473), [km669] MIRTJ_KREG(K.km669), [k277] MIRTJ_KREG(K.k277), [selm] MIRTJ_KREG(K.sel_m),          \
      [c128] "v"(K.c128)

#define MIRTJ_PK_ADD(D, A, B) "v_pk_add_i16 " D ", " A ", " B "\n\t"
#define MIRTJ_PK_SUB(D, A, B) "v_pk_sub_i16 " D ", " A ", " B "\n\t"
// X <- MULTIPLY(X, K) in both halves
#define MIRTJ_PK_M(X, K)                                             \
  "v_mad_i32_i16 %[pl], " X ", " K ", %[c128]\n\t"                   \
  "v_mad_i32_i16 %[ph], " X ", " K ", %[c128] op_sel:[1,0,0,0]\n\t"  \
  "v_perm_b32 " X ", %[ph], %[pl], %[selm]\n\t"
// D <- MULTIPLY(X, K), X kept
#define MIRTJ_PK_M2(D, X, K)                                         \
  "v_mad_i32_i16 %[pl], " X ", " K ", %[c128]\n\t"                   \
  "v_mad_i32_i16 %[ph], " X ", " K ", %[c128] op_sel:[1,0,0,0]\n\t"  \
  "v_perm_b32 " D ", %[ph], %[pl], %[selm]\n\t"

#define MIRTJ_PK_STAGE1(x0, x1, x2, x3, x4, x5, x6, x7, t0, t1) \
  MIRTJ_PK_ADD(t0, x0, x4) /* s04 */                            \
  MIRTJ_PK_SUB(x0, x0, x4) /* d04 */                            \
  MIRTJ_PK_ADD(x4, x2, x6) /* s26 */                            \
  MIRTJ_PK_SUB(x2, x2, x6) /* x2 - x6 */                        \
  MIRTJ_PK_ADD(x6, x5, x3) /* s53 */                            \
  MIRTJ_PK_SUB(x5, x5, x3) /* d53 */                            \
  MIRTJ_PK_ADD(x3, x1, x7) /* s17 */                            \
  MIRTJ_PK_SUB(x1, x1, x7) /* d17 */                            \
  MIRTJ_PK_SUB(x7, x3, x6) /* s17 - s53 */                      \
  MIRTJ_PK_ADD(t1, x5, x1) /* d53 + d17 */
#define MIRTJ_PK_PRODUCTS(x1, x2, x5, x7, t1)                                                           \
  MIRTJ_PK_M(x2, "%[k362]") MIRTJ_PK_M(x7, "%[k362]") MIRTJ_PK_M(t1, "%[k473]") MIRTJ_PK_M(x5, "%[km669]") \
  MIRTJ_PK_M(x1, "%[k277]")
// as MIRTJ_IDCT_TAIL (rtj_idct_asm.h), the products already shifted: on exit y0..y7 = x2 x3 x5 t0 x7 x0 x4 x6
#define MIRTJ_PK_TAIL(x0, x1, x2, x3, x4, x5, x6, x7, t0, t1) \
  MIRTJ_PK_SUB(x2, x2, x4) /* r26 */                          \
  MIRTJ_PK_ADD(x3, x3, x6) /* o7 */                           \
  MIRTJ_PK_ADD(x6, t0, x4) /* e0 */                           \
  MIRTJ_PK_SUB(t0, t0, x4) /* e3 */                           \
  MIRTJ_PK_ADD(x4, x0, x2) /* e1 */                           \
  MIRTJ_PK_SUB(x0, x0, x2) /* e2 */                           \
  MIRTJ_PK_ADD(x5, x5, t1)                                    \
  MIRTJ_PK_SUB(x5, x5, x3) /* o6 */                           \
  MIRTJ_PK_SUB(x7, x7, x5) /* o5 */                           \
  MIRTJ_PK_SUB(x1, x1, t1)                                    \
  MIRTJ_PK_ADD(x1, x1, x7) /* o4 */                           \
  MIRTJ_PK_ADD(x2, x6, x3) /* y0 */                           \
  MIRTJ_PK_SUB(x6, x6, x3) /* y7 */                           \
  MIRTJ_PK_ADD(x3, x4, x5) /* y1 */                           \
  MIRTJ_PK_SUB(x4, x4, x5) /* y6 */                           \
  MIRTJ_PK_ADD(x5, x0, x7) /* y2 */                           \
  MIRTJ_PK_SUB(x0, x0, x7) /* y5 */                           \
  MIRTJ_PK_ADD(x7, t0, x1) /* y4 */                           \
  MIRTJ_PK_SUB(t0, t0, x1) /* y3 */

// DESCALE (the +4 came in with DC), clamp to 16..235 (RTjpeg; the DV decoder passes 0 and 255 | 255 << 16: operands lo,
// c235), and the two rows' pixels sorted into their dwords:
// y0..y7 hold (row r | row r+1 << 16) of pixel columns 0..7; on exit y1, y5 = row r and y3, y7 = row r + 1
#define MIRTJ_PK_PX1(Y)                                      \
  "v_pk_ashrrev_i16 " Y ", 3, " Y " op_sel_hi:[0,1]\n\t"     \
  "v_pk_max_i16 " Y ", %[lo], " Y " op_sel_hi:[0,1]\n\t"     \
  "v_pk_min_i16 " Y ", " Y ", %[c235]\n\t"
#define MIRTJ_PK_PX_PACK(y0, y1, y2, y3, y4, y5, y6, y7)                                                            \
  MIRTJ_PK_PX1(y0) MIRTJ_PK_PX1(y1) MIRTJ_PK_PX1(y2) MIRTJ_PK_PX1(y3) MIRTJ_PK_PX1(y4) MIRTJ_PK_PX1(y5)             \
  MIRTJ_PK_PX1(y6) MIRTJ_PK_PX1(y7)                                                                                 \
  "v_perm_b32 " y0 ", " y1 ", " y0 ", %[selt]\n\t" /* px0 r, px1 r, px0 r+1, px1 r+1 */                             \
  "v_perm_b32 " y2 ", " y3 ", " y2 ", %[selt]\n\t"                                                                  \
  "v_perm_b32 " y4 ", " y5 ", " y4 ", %[selt]\n\t"                                                                  \
  "v_perm_b32 " y6 ", " y7 ", " y6 ", %[selt]\n\t"                                                                  \
  "v_perm_b32 " y1 ", " y2 ", " y0 ", %[sello]\n\t" /* row r, pixels 0-3 */                                         \
  "v_perm_b32 " y3 ", " y2 ", " y0 ", %[selhi]\n\t" /* row r + 1, pixels 0-3 */                                     \
  "v_perm_b32 " y5 ", " y6 ", " y4 ", %[sello]\n\t" /* row r, pixels 4-7 */                                         \
  "v_perm_b32 " y7 ", " y6 ", " y4 ", %[selhi]\n\t" /* row r + 1, pixels 4-7 */

// ---- column pass on a column pair: x[r] = (row r of the even column | row r of the odd one << 16), r = 0..7;
// on return x[r] holds the pass's result for row r.  kDc: the pair (0, 1); column 0 carries DESCALE's +4. ----
template <bool kDc>
__device__ __forceinline__ void idct8_pk_col(uint32_t (&x)[8], const IdctPK& K) {
  uint32_t t0, t1, pl, ph;
  uint32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4], x5 = x[5], x6 = x[6], x7 = x[7];
#define MIRTJ_PK_COL_OPERANDS                                                                                     \
  [t0] "=&v"(t0), [t1] "=&v"(t1), [pl] "=&v"(pl), [ph] "=&v"(ph), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2),     \
      [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6), [x7] "+v"(x7)                                    \
      : MIRTJ_PK_OPERANDS, [four] "v"(K.four)
  if (kDc)
    asm(MIRTJ_PK_STAGE1("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
        MIRTJ_PK_ADD("%[t0]", "%[t0]", "%[four]") MIRTJ_PK_ADD("%[x0]", "%[x0]", "%[four]")
        MIRTJ_PK_PRODUCTS("%[x1]", "%[x2]", "%[x5]", "%[x7]", "%[t1]")
        MIRTJ_PK_TAIL("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
        : MIRTJ_PK_COL_OPERANDS);
  else
    asm(MIRTJ_PK_STAGE1("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
        MIRTJ_PK_PRODUCTS("%[x1]", "%[x2]", "%[x5]", "%[x7]", "%[t1]")
        MIRTJ_PK_TAIL("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
        : MIRTJ_PK_COL_OPERANDS);
  x[0] = x2; x[1] = x3; x[2] = x5; x[3] = t0; x[4] = x7; x[5] = x0; x[6] = x4; x[7] = x6;
}

// ---- row pass on a row pair + descale, clamp, pack.  ya[j], yb[j] = the column pass's results for rows r and r + 1
// of the column pair j.  The block first regroups them into (row r | row r+1 << 16) of one column each — inside the
// block, so that the regrouped values never exist for more than one row pair at a time — and runs in the twelve
// registers it is given.  Returns row r in a and row r + 1 in b. ----
#define MIRTJ_PK_REGROUP(T, YA, YB)                                  \
  "v_perm_b32 " T ", " YB ", " YA ", %[sello]\n\t" /* even column */ \
  "v_perm_b32 " YB ", " YB ", " YA ", %[selhi]\n\t" /* odd column */
template <int kLo = 16>  // the lower clamp (an inline constant); the upper one is K.c235
__device__ __forceinline__ void idct8_pk_row_px(uint32_t (&ya)[4], uint32_t (&yb)[4], uint2& a, uint2& b,
                                                const IdctPK& K) {
  uint32_t x0, x2, x4, x6;
  uint32_t a0 = ya[0], a1 = ya[1], a2 = ya[2], a3 = ya[3], b0 = yb[0], b1 = yb[1], b2 = yb[2], b3 = yb[3];
  // x1 x3 x5 x7 = b0..b3 after the regrouping; t0 t1 pl ph = a0..a3 (free by then)
  asm(MIRTJ_PK_REGROUP("%[x0]", "%[t0]", "%[x1]") MIRTJ_PK_REGROUP("%[x2]", "%[t1]", "%[x3]")
      MIRTJ_PK_REGROUP("%[x4]", "%[pl]", "%[x5]") MIRTJ_PK_REGROUP("%[x6]", "%[ph]", "%[x7]")
      MIRTJ_PK_STAGE1("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
      MIRTJ_PK_PRODUCTS("%[x1]", "%[x2]", "%[x5]", "%[x7]", "%[t1]")
      MIRTJ_PK_TAIL("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
      MIRTJ_PK_PX_PACK("%[x2]", "%[x3]", "%[x5]", "%[t0]", "%[x7]", "%[x0]", "%[x4]", "%[x6]")
      : [x0] "=&v"(x0), [x2] "=&v"(x2), [x4] "=&v"(x4), [x6] "=&v"(x6), [t0] "+v"(a0), [t1] "+v"(a1), [pl] "+v"(a2),
        [ph] "+v"(a3), [x1] "+v"(b0), [x3] "+v"(b1), [x5] "+v"(b2), [x7] "+v"(b3)
      : MIRTJ_PK_OPERANDS, [selt] MIRTJ_KREG(K.sel_t), [sello] MIRTJ_KREG(K.sel_lo), [selhi] MIRTJ_KREG(K.sel_hi), [c235] "v"(K.c235),
        [lo] "n"(kLo));
  // MIRTJ_PK_PX_PACK leaves the rows in the registers of y1 y3 y5 y7 = x3 t0 x0 x6
  a = make_uint2(b1, x0);
  b = make_uint2(a0, x6);
}

// ---- the same when only rows / columns 0, 1, 2 can be non-zero (idct8_lo3) ----
// On entry a, m, z, b = M(x2,362), M(x1,362), M(x1,473), M(x1,277); on exit y0..y7 = e0 e1 e2 a b x1 z x0.
#define MIRTJ_PK_LO3_TAIL(x0, x1, x2, a, m, z, b, e0, e1, e2) \
  MIRTJ_PK_SUB(a, a, x2)   /* r26 */                          \
  MIRTJ_PK_ADD(e0, x0, x2) /* e0 */                           \
  MIRTJ_PK_SUB(x2, x0, x2) /* e3 */                           \
  MIRTJ_PK_ADD(e1, x0, a)  /* e1 */                           \
  MIRTJ_PK_SUB(e2, x0, a)  /* e2 */                           \
  MIRTJ_PK_SUB(a, z, x1)   /* o6 */                           \
  MIRTJ_PK_SUB(m, m, a)    /* o5 */                           \
  MIRTJ_PK_SUB(b, b, z)                                       \
  MIRTJ_PK_ADD(b, b, m)    /* o4 */                           \
  MIRTJ_PK_SUB(x0, e0, x1) /* y7 */                           \
  MIRTJ_PK_ADD(e0, e0, x1) /* y0 */                           \
  MIRTJ_PK_SUB(z, e1, a)   /* y6 */                           \
  MIRTJ_PK_ADD(e1, e1, a)  /* y1 */                           \
  MIRTJ_PK_SUB(x1, e2, m)  /* y5 */                           \
  MIRTJ_PK_ADD(e2, e2, m)  /* y2 */                           \
  MIRTJ_PK_SUB(a, x2, b)   /* y3 */                           \
  MIRTJ_PK_ADD(b, x2, b)   /* y4 */
#define MIRTJ_PK_LO3_PRODUCTS                                                                                   \
  MIRTJ_PK_M2("%[a]", "%[x2]", "%[k362]") MIRTJ_PK_M2("%[m]", "%[x1]", "%[k362]") MIRTJ_PK_M2("%[z]", "%[x1]", "%[k473]") \
  MIRTJ_PK_M2("%[b]", "%[x1]", "%[k277]")

template <bool kDc>
__device__ __forceinline__ void idct8_pk_lo3_col(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t (&y)[8],
                                                 const IdctPK& K) {
  uint32_t a, m, z, b, e0, e1, e2, pl, ph;
#define MIRTJ_PK_LO3_COL_OPERANDS                                                                                  \
  [a] "=&v"(a), [m] "=&v"(m), [z] "=&v"(z), [b] "=&v"(b), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2),          \
      [pl] "=&v"(pl), [ph] "=&v"(ph), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2)                                   \
      : MIRTJ_PK_OPERANDS, [four] "v"(K.four)
  if (kDc)
    asm(MIRTJ_PK_ADD("%[x0]", "%[x0]", "%[four]") MIRTJ_PK_LO3_PRODUCTS
        MIRTJ_PK_LO3_TAIL("%[x0]", "%[x1]", "%[x2]", "%[a]", "%[m]", "%[z]", "%[b]", "%[e0]", "%[e1]", "%[e2]")
        : MIRTJ_PK_LO3_COL_OPERANDS);
  else
    asm(MIRTJ_PK_LO3_PRODUCTS
        MIRTJ_PK_LO3_TAIL("%[x0]", "%[x1]", "%[x2]", "%[a]", "%[m]", "%[z]", "%[b]", "%[e0]", "%[e1]", "%[e2]")
        : MIRTJ_PK_LO3_COL_OPERANDS);
  y[0] = e0; y[1] = e1; y[2] = e2; y[3] = a; y[4] = b; y[5] = x1; y[6] = z; y[7] = x0;
}

// ya0, ya1 = rows r, r + 1 of the column pair (0, 1); yb0, yb1 = of the pair (2, 3)
__device__ __forceinline__ void idct8_pk_lo3_row_px(uint32_t ya0, uint32_t ya1, uint32_t yb0, uint32_t yb1, uint2& ra,
                                                    uint2& rb, const IdctPK& K) {
  uint32_t x0, a, m, z, b, e0, e1, e2;
  // x1 = ya1, x2 = yb0 after the regrouping; pl, ph = ya0, yb1 (free by then)
  asm("v_perm_b32 %[x0], %[x1], %[pl], %[sello]\n\t"
      "v_perm_b32 %[x1], %[x1], %[pl], %[selhi]\n\t"
      "v_perm_b32 %[x2], %[ph], %[x2], %[sello]\n\t"
      MIRTJ_PK_LO3_PRODUCTS
      MIRTJ_PK_LO3_TAIL("%[x0]", "%[x1]", "%[x2]", "%[a]", "%[m]", "%[z]", "%[b]", "%[e0]", "%[e1]", "%[e2]")
      MIRTJ_PK_PX_PACK("%[e0]", "%[e1]", "%[e2]", "%[a]", "%[b]", "%[x1]", "%[z]", "%[x0]")
      : [x0] "=&v"(x0), [a] "=&v"(a), [m] "=&v"(m), [z] "=&v"(z), [b] "=&v"(b), [e0] "=&v"(e0), [e1] "=&v"(e1),
        [e2] "=&v"(e2), [pl] "+v"(ya0), [x1] "+v"(ya1), [x2] "+v"(yb0), [ph] "+v"(yb1)
      : MIRTJ_PK_OPERANDS, [selt] MIRTJ_KREG(K.sel_t), [sello] MIRTJ_KREG(K.sel_lo), [selhi] MIRTJ_KREG(K.sel_hi), [c235] "v"(K.c235),
        [lo] "n"(16));
  // rows in the registers of y1 y3 y5 y7 = e1 a x1 x0
  ra = make_uint2(e1, ya1);
  rb = make_uint2(a, x0);
}

// acc + (32768 - |lo half|) + (32768 - |hi half|), halves read as int16
__device__ __forceinline__ uint32_t pk_sad(uint32_t w, uint32_t acc, const IdctPK& K) {
  uint32_t r;
  asm("v_sad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(w), MIRTJ_KREG(K.sadk), "v"(acc));
  return r;
}

// Is the weighted sum of |coefficients| of this lane's block inside the budget?  q[2j], q[2j+1]: rows 0-3, 4-7 of pair j.
__device__ __forceinline__ bool pk_range_full(const uint4 (&q)[8], const IdctPK& K) {
  uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int halves[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const uint4& v = q[2 * j + (r >> 2)];
      const uint32_t w = (r & 3) == 0 ? v.x : (r & 3) == 1 ? v.y : (r & 3) == 2 ? v.z : v.w;
      const int k = pk_class(j, r);
      acc[k] = pk_sad(w, acc[k], K);
      halves[k] += 2;
    }
  }
  // sum_k W_k * (32768 * halves_k - acc_k) <= budget
  uint32_t s = 0;
  int all = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    s += __umul24((uint32_t)kPkClassW[k], acc[k]);  // acc < 2^21, the sum < 2^26
    all += kPkClassW[k] * halves[k];
  }
  return s >= (uint32_t)all * 32768u - (uint32_t)kPkBudget;
}
// the same for a block with nothing outside rows / columns 0-2 (every weight is 1): a = rows 0-3 of columns (0, 1),
// b = rows 0-3 of columns (2, 3)
__device__ __forceinline__ bool pk_range_lo3(const uint4& a, const uint4& b, const IdctPK& K) {
  uint32_t acc = 0;
  acc = pk_sad(a.x, acc, K);
  acc = pk_sad(a.y, acc, K);
  acc = pk_sad(a.z, acc, K);
  acc = pk_sad(b.x, acc, K);
  acc = pk_sad(b.y, acc, K);
  acc = pk_sad(b.z, acc, K);
  return acc >= 12u * 32768u - (uint32_t)(kPkBudget / 4);
}

}  // namespace mirtj
