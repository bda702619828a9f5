// rtj_idct_asm.h — the 8-point inverse AAN butterfly of RTjpeg_idct (lib/RTjpeg.c:2209-2332) as hand-ordered gfx950
// instruction blocks.  Same arithmetic as idct8 / idct8_lo / idct8_lo3 / px in rtj_decode_kernels.h (which stay as the
// readable statement of it and as the A/B baseline, -DMIRTJ_ASM_IDCT=0); what differs is the ORDER of instructions.
//
// Why the order matters (tools/ubench/valu_snop.hip, valu_mix*.hip; profiles/r02/ubench_*):
//   * plain 32-bit add / sub / shift / logic, and 16-bit add / min / max, issue at ~1.0 ns per wave64 instruction per
//     SIMD ("cheap"); everything else the transform needs — v_mad_i32_i24, SDWA forms, v_bfe, v_med3, v_lshl_or,
//     packed 16-bit — at ~1.8 ns ("expensive");
//   * but once a wave has issued an expensive instruction, its FOLLOWING cheap instructions are also issued at the
//     expensive rate, until the wave issues a scalar instruction: one mad per 64 adds makes all 64 cost 1.65 ns
//     instead of 1.0; an `s_nop 0` behind the mad gives the adds their own rate back (1.08 ns per instruction for
//     one mad per eight adds, against 1.82 without).
// The compiler knows nothing of this and interleaves the two kinds freely (the round-1 transform ran at 1.6-1.9 ns
// per instruction throughout).  Here every 1-D pass is: cheap first stage, the five products as one cluster, ONE
// s_nop, then a tail of cheap instructions only; descale and clamp are cheap 16-bit operations, and the byte
// packing rides on the clamp's SDWA destination select (one expensive cluster per row, one s_nop).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirtj {

#ifndef MIRTJ_ASM_IDCT
#define MIRTJ_ASM_IDCT 1
#endif

// wave-uniform constants the blocks take as operands: FIX_1_414213562 = 362, FIX_1_847759065 = 473,
// -FIX_2_613125930 = -669, FIX_1_082392200 = 277 (lib/RTjpeg.c:1196-1199) in scalar registers; 128 (MULTIPLY's
// rounding term, :1206) and 235 (RL's upper bound, :1205) in vector registers (a VOP3 takes one scalar operand).
// Where the transform's multipliers and byte selectors live.  They are wave-uniform and the compiler keeps them in scalar
// registers ("s").  k_decode is short of those (round 2's kernel paid for 50 spilled scalars in v_writelane / v_readlane;
// the batch instantiation is down to 10 since the launch shape became a template parameter), and it has vector
// registers to spare below the 128 that four waves per SIMD allow, so -DMIRTJ_K_IN_VGPR makes them vector operands:
// measured, that leaves 7 spills instead of 10 and is 0.5 % SLOWER (profiles/r03/ab_spills.txt) — the default stays "s".
#ifdef MIRTJ_K_IN_VGPR
#define MIRTJ_KREG(x) "v"(x)
#else
#define MIRTJ_KREG(x) "s"(x)
#endif
struct IdctK {
  int k362, k473, km669, k277;  // scalar
  int c128, c235;               // vector
};

// wait states behind a cluster of expensive instructions: one per instruction of the cluster is what it takes
// (tools/ubench/valu_cluster.hip: five mads want `s_nop 4`; `s_nop 0` behind them changes nothing)
#ifndef MIRTJ_NOP_ROW
#define MIRTJ_NOP_ROW 4
#endif
#ifndef MIRTJ_NOP_COL
#define MIRTJ_NOP_COL 12
#endif
#ifndef MIRTJ_NOP_PX
#define MIRTJ_NOP_PX 7
#endif
#define MIRTJ_STR2(x) #x
#define MIRTJ_STR(x) MIRTJ_STR2(x)
// MIRTJ_WAIT(n): n + 1 wait states (n up to 31)
#define MIRTJ_WAIT_A(n) "s_nop " MIRTJ_STR(n)
#define MIRTJ_WAIT_B(n) "s_nop 15\n\ts_nop " MIRTJ_STR(n)
#if MIRTJ_NOP_ROW < 16
#define MIRTJ_WAIT_ROW MIRTJ_WAIT_A(MIRTJ_NOP_ROW)
#else
#define MIRTJ_WAIT_ROW MIRTJ_WAIT_B(MIRTJ_NOP_ROW - 16)
#endif
#if MIRTJ_NOP_COL < 16
#define MIRTJ_WAIT_COL MIRTJ_WAIT_A(MIRTJ_NOP_COL)
#else
#define MIRTJ_WAIT_COL MIRTJ_WAIT_B(MIRTJ_NOP_COL - 16)
#endif
#if MIRTJ_NOP_PX < 16
#define MIRTJ_WAIT_PX MIRTJ_WAIT_A(MIRTJ_NOP_PX)
#else
#define MIRTJ_WAIT_PX MIRTJ_WAIT_B(MIRTJ_NOP_PX - 16)
#endif

// MULTIPLY(x, c) = (x * c + 128) >> 8 is split in two: MAD here, the shift in the cheap tail.
#define MIRTJ_MAD(D, X, K) "v_mad_i32_i24 " D ", " X ", " K ", %[c128]\n\t"
#define MIRTJ_ASHR8(D) "v_ashrrev_i32 " D ", 8, " D "\n\t"

// The tail shared by every full pass.  On entry:
//   s04 (t0)  d04 (x0)  s26 (x4)  r26' (x2)  s53 (x6)  M(d53)' (x5)  s17 (x3)  M(d17)' (x1)  m' (x7)  z5' (t1)
// (primed values still carry the factor 256).  On exit y0..y7 = x2 x3 x5 t0 x7 x0 x4 x6.
#define MIRTJ_IDCT_TAIL(x0, x1, x2, x3, x4, x5, x6, x7, t0, t1) \
  MIRTJ_ASHR8(x2) MIRTJ_ASHR8(x7) MIRTJ_ASHR8(t1) MIRTJ_ASHR8(x5) MIRTJ_ASHR8(x1) \
  "v_sub_u32 " x2 ", " x2 ", " x4 "\n\t" /* r26 = M(x2 - x6, 362) - s26 */        \
  "v_add_u32 " x3 ", " x3 ", " x6 "\n\t" /* o7 = s17 + s53 */                     \
  "v_add_u32 " x6 ", " t0 ", " x4 "\n\t" /* e0 = s04 + s26 */                     \
  "v_sub_u32 " t0 ", " t0 ", " x4 "\n\t" /* e3 = s04 - s26 */                     \
  "v_add_u32 " x4 ", " x0 ", " x2 "\n\t" /* e1 = d04 + r26 */                     \
  "v_sub_u32 " x0 ", " x0 ", " x2 "\n\t" /* e2 = d04 - r26 */                     \
  "v_add_u32 " x5 ", " x5 ", " t1 "\n\t" /* M(d53, -669) + z5 */                  \
  "v_sub_u32 " x5 ", " x5 ", " x3 "\n\t" /* o6 = ... - o7 */                      \
  "v_sub_u32 " x7 ", " x7 ", " x5 "\n\t" /* o5 = m - o6 */                        \
  "v_sub_u32 " x1 ", " x1 ", " t1 "\n\t" /* M(d17, 277) - z5 */                   \
  "v_add_u32 " x1 ", " x1 ", " x7 "\n\t" /* o4 = ... + o5 */                      \
  "v_add_u32 " x2 ", " x6 ", " x3 "\n\t" /* y0 = e0 + o7 */                       \
  "v_sub_u32 " x6 ", " x6 ", " x3 "\n\t" /* y7 = e0 - o7 */                       \
  "v_add_u32 " x3 ", " x4 ", " x5 "\n\t" /* y1 = e1 + o6 */                       \
  "v_sub_u32 " x4 ", " x4 ", " x5 "\n\t" /* y6 = e1 - o6 */                       \
  "v_add_u32 " x5 ", " x0 ", " x7 "\n\t" /* y2 = e2 + o5 */                       \
  "v_sub_u32 " x0 ", " x0 ", " x7 "\n\t" /* y5 = e2 - o5 */                       \
  "v_add_u32 " x7 ", " t0 ", " x1 "\n\t" /* y4 = e3 + o4 */                       \
  "v_sub_u32 " t0 ", " t0 ", " x1 "\n\t" /* y3 = e3 - o4 */

// DESCALE + int16 narrowing + clamp 16..235 of eight values (the +4 was folded into the DC coefficient), packed
// into two dwords.  Two forms:
//   MIRTJ_PX_FORM 0: v_bfe_i32 (bits 18:3, sign-extended = int16(v >> 3)), v_med3_i32, and three
//     v_lshl_or_b32 per four pixels: 22 instructions per row, the fewest — and while other waves of the SIMD keep
//     expensive instructions in flight every instruction costs the same (rtj_decode_kernels.h, "what an instruction
//     costs"), so the count is what matters;
//   MIRTJ_PX_FORM 1 (default; the two measure alike in k_decode, 4.54-4.56 against 4.56-4.60 ms): v_ashrrev_i32 3, v_max_i16 16 (both plain: v >> 3 leaves int16(v >> 3) in the low half, which
//     the 16-bit max / min read as signed) and a v_min_i16 whose SDWA destination select writes the byte straight
//     into the packed dword: 24 instructions, 16 of them plain — the faster form when the SIMD's waves run in step.
#ifndef MIRTJ_PX_FORM
#define MIRTJ_PX_FORM 1
#endif
#if MIRTJ_PX_FORM == 1
#define MIRTJ_PX1(Y) "v_ashrrev_i32 " Y ", 3, " Y "\n\tv_max_i16 " Y ", 16, " Y "\n\t"
#define MIRTJ_PACK1(O, Y, SEL, KEEP) \
  "v_min_i16_sdwa " O ", " Y ", %[c235] dst_sel:" SEL " dst_unused:" KEEP " src0_sel:WORD_0 src1_sel:WORD_0\n\t"
#define MIRTJ_PX_PACK(y0, y1, y2, y3, y4, y5, y6, y7)                                                            \
  MIRTJ_PX1(y0) MIRTJ_PX1(y1) MIRTJ_PX1(y2) MIRTJ_PX1(y3) MIRTJ_PX1(y4) MIRTJ_PX1(y5) MIRTJ_PX1(y6) MIRTJ_PX1(y7) \
  MIRTJ_PACK1("%[o0]", y0, "BYTE_0", "UNUSED_PAD") MIRTJ_PACK1("%[o1]", y4, "BYTE_0", "UNUSED_PAD")             \
  MIRTJ_PACK1("%[o0]", y1, "BYTE_1", "UNUSED_PRESERVE") MIRTJ_PACK1("%[o1]", y5, "BYTE_1", "UNUSED_PRESERVE")   \
  MIRTJ_PACK1("%[o0]", y2, "BYTE_2", "UNUSED_PRESERVE") MIRTJ_PACK1("%[o1]", y6, "BYTE_2", "UNUSED_PRESERVE")   \
  MIRTJ_PACK1("%[o0]", y3, "BYTE_3", "UNUSED_PRESERVE") MIRTJ_PACK1("%[o1]", y7, "BYTE_3", "UNUSED_PRESERVE")   \
  MIRTJ_WAIT_PX
#else
#define MIRTJ_PX1(Y) "v_bfe_i32 " Y ", " Y ", 3, 16\n\tv_med3_i32 " Y ", " Y ", 16, %[c235]\n\t"
#define MIRTJ_PX_PACK(y0, y1, y2, y3, y4, y5, y6, y7)                                                            \
  MIRTJ_PX1(y0) MIRTJ_PX1(y1) MIRTJ_PX1(y2) MIRTJ_PX1(y3) MIRTJ_PX1(y4) MIRTJ_PX1(y5) MIRTJ_PX1(y6) MIRTJ_PX1(y7) \
  "v_lshl_or_b32 " y1 ", " y1 ", 8, " y0 "\n\t"                                                                  \
  "v_lshl_or_b32 " y3 ", " y3 ", 8, " y2 "\n\t"                                                                  \
  "v_lshl_or_b32 " y5 ", " y5 ", 8, " y4 "\n\t"                                                                  \
  "v_lshl_or_b32 " y7 ", " y7 ", 8, " y6 "\n\t"                                                                  \
  "v_lshl_or_b32 %[o0], " y3 ", 16, " y1 "\n\t"                                                                  \
  "v_lshl_or_b32 %[o1], " y7 ", 16, " y5 "\n\t"                                                                  \
  MIRTJ_WAIT_PX
#endif

#define MIRTJ_K_OPERANDS \
  [k362] MIRTJ_KREG(K.k362), [k473] MIRTJ_KREG(K.k473), [km669] MIRTJ_KREG(K.km669), [k277] MIRTJ_KREG(K.k277), [c128] "v"(K.c128), [c235] "v"(K.c235)

// ---- row pass on eight int32 values + descale, clamp, pack: one row of the block as two dwords ----
__device__ __forceinline__ uint2 idct8_row_px(int x0, int x1, int x2, int x3, int x4, int x5, int x6, int x7,
                                              const IdctK& K) {
  uint32_t o0, o1;
  int t0, t1;
  asm(  // first stage: cheap
      "v_add_u32 %[t0], %[x0], %[x4]\n\t"  // s04
      "v_sub_u32 %[x0], %[x0], %[x4]\n\t"  // d04
      "v_add_u32 %[x4], %[x2], %[x6]\n\t"  // s26
      "v_sub_u32 %[x2], %[x2], %[x6]\n\t"  // x2 - x6
      "v_add_u32 %[x6], %[x5], %[x3]\n\t"  // s53
      "v_sub_u32 %[x5], %[x5], %[x3]\n\t"  // d53
      "v_add_u32 %[x3], %[x1], %[x7]\n\t"  // s17
      "v_sub_u32 %[x1], %[x1], %[x7]\n\t"  // d17
      "v_sub_u32 %[x7], %[x3], %[x6]\n\t"  // s17 - s53
      "v_add_u32 %[t1], %[x5], %[x1]\n\t"  // d53 + d17
      // the five products, then the scalar instruction that ends the expensive stretch
      MIRTJ_MAD("%[x2]", "%[x2]", "%[k362]") MIRTJ_MAD("%[x7]", "%[x7]", "%[k362]") MIRTJ_MAD("%[t1]", "%[t1]", "%[k473]")
      MIRTJ_MAD("%[x5]", "%[x5]", "%[km669]") MIRTJ_MAD("%[x1]", "%[x1]", "%[k277]")
      MIRTJ_WAIT_ROW "\n\t"
      MIRTJ_IDCT_TAIL("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
      MIRTJ_PX_PACK("%[x2]", "%[x3]", "%[x5]", "%[t0]", "%[x7]", "%[x0]", "%[x4]", "%[x6]")
      : [o0] "=&v"(o0), [o1] "=&v"(o1), [t0] "=&v"(t0), [t1] "=&v"(t1), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2),
        [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6), [x7] "+v"(x7)
      : MIRTJ_K_OPERANDS);
  return make_uint2(o0, o1);
}

// ---- column pass from the scratch: a = rows 0-3, b = rows 4-7 of a column PAIR, one row to a dword, the even
// column in the low half (rtj_decode_kernels.h, coef_byte); kOdd picks the column.  The first stage reads the halves
// directly (SDWA: expensive, so it opens the cluster).  kDc: column 0 carries DESCALE's rounding term, +4 on the
// DC coefficient, through both of its linear paths. ----
#define MIRTJ_SDWA2(OP, D, A, B, SEL) \
  OP " " D ", sext(" A "), sext(" B ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SEL " src1_sel:" SEL "\n\t"
#define MIRTJ_COL_HEAD(SEL)                                                                         \
  MIRTJ_SDWA2("v_add_u32_sdwa", "%[t0]", "%[ax]", "%[bx]", SEL) /* s04 = x0 + x4 */                 \
  MIRTJ_SDWA2("v_sub_u32_sdwa", "%[x0]", "%[ax]", "%[bx]", SEL) /* d04 = x0 - x4 */                 \
  MIRTJ_SDWA2("v_add_u32_sdwa", "%[x4]", "%[az]", "%[bz]", SEL) /* s26 = x2 + x6 */                 \
  MIRTJ_SDWA2("v_sub_u32_sdwa", "%[x2]", "%[az]", "%[bz]", SEL) /* x2 - x6 */                       \
  MIRTJ_SDWA2("v_add_u32_sdwa", "%[x6]", "%[by]", "%[aw]", SEL) /* s53 = x5 + x3 */                 \
  MIRTJ_SDWA2("v_sub_u32_sdwa", "%[x5]", "%[by]", "%[aw]", SEL) /* d53 = x5 - x3 */                 \
  MIRTJ_SDWA2("v_add_u32_sdwa", "%[x3]", "%[ay]", "%[bw]", SEL) /* s17 = x1 + x7 */                 \
  MIRTJ_SDWA2("v_sub_u32_sdwa", "%[x1]", "%[ay]", "%[bw]", SEL) /* d17 = x1 - x7 */                 \
  "v_sub_u32 %[x7], %[x3], %[x6]\n\t"                                          /* s17 - s53 */                \
  "v_add_u32 %[t1], %[x5], %[x1]\n\t"                                          /* d53 + d17 */                \
  MIRTJ_MAD("%[x2]", "%[x2]", "%[k362]") MIRTJ_MAD("%[x7]", "%[x7]", "%[k362]") MIRTJ_MAD("%[t1]", "%[t1]", "%[k473]") \
  MIRTJ_MAD("%[x5]", "%[x5]", "%[km669]") MIRTJ_MAD("%[x1]", "%[x1]", "%[k277]")                               \
  MIRTJ_WAIT_COL "\n\t"
#define MIRTJ_COL_OPERANDS                                                                                              \
  [t0] "=&v"(t0), [t1] "=&v"(t1), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3), [x4] "=&v"(x4),       \
      [x5] "=&v"(x5), [x6] "=&v"(x6), [x7] "=&v"(x7)                                                                    \
      : [ax] "v"(a.x), [ay] "v"(a.y), [az] "v"(a.z), [aw] "v"(a.w), [bx] "v"(b.x), [by] "v"(b.y), [bz] "v"(b.z),      \
        [bw] "v"(b.w), MIRTJ_K_OPERANDS
template <bool kDc, bool kOdd>
__device__ __forceinline__ void idct8_col(const uint4& a, const uint4& b, int (&y)[8], const IdctK& K) {
  int x0, x1, x2, x3, x4, x5, x6, x7, t0, t1;
  if (kDc)  // (column 0: even)
    asm(MIRTJ_COL_HEAD("WORD_0")
        "v_add_u32 %[t0], 4, %[t0]\n\t"
        "v_add_u32 %[x0], 4, %[x0]\n\t"
        MIRTJ_IDCT_TAIL("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
        : MIRTJ_COL_OPERANDS);
  else if (kOdd)
    asm(MIRTJ_COL_HEAD("WORD_1")
        MIRTJ_IDCT_TAIL("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
        : MIRTJ_COL_OPERANDS);
  else
    asm(MIRTJ_COL_HEAD("WORD_0")
        MIRTJ_IDCT_TAIL("%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[x4]", "%[x5]", "%[x6]", "%[x7]", "%[t0]", "%[t1]")
        : MIRTJ_COL_OPERANDS);
  y[0] = x2; y[1] = x3; y[2] = x5; y[3] = t0; y[4] = x7; y[5] = x0; y[6] = x4; y[7] = x6;
}

// ---- the same when only x0, x1, x2 can be non-zero (idct8_lo3: every term the others feed vanishes exactly) ----
#ifndef MIRTJ_NOP_LO3
#define MIRTJ_NOP_LO3 6
#endif
// On entry a' = M(x2,362)', m' = M(x1,362)', z' = M(x1,473)', b' = M(x1,277)' (factor 256 still on), x0 x1 x2 plain.
// On exit y0..y7 = e0 e1 e2 a b x1 z x0 (named after the registers they end up in).
#define MIRTJ_LO3_TAIL(x0, x1, x2, a, m, z, b, e0, e1, e2)                  \
  MIRTJ_ASHR8(a) MIRTJ_ASHR8(m) MIRTJ_ASHR8(z) MIRTJ_ASHR8(b)               \
  "v_sub_u32 " a ", " a ", " x2 "\n\t"   /* r26 = M(x2, 362) - x2 */       \
  "v_add_u32 " e0 ", " x0 ", " x2 "\n\t" /* e0 */                          \
  "v_sub_u32 " x2 ", " x0 ", " x2 "\n\t" /* e3 (in x2) */                  \
  "v_add_u32 " e1 ", " x0 ", " a "\n\t"  /* e1 */                          \
  "v_sub_u32 " e2 ", " x0 ", " a "\n\t"  /* e2 */                          \
  "v_sub_u32 " a ", " z ", " x1 "\n\t"   /* o6 = z5 - x1 (in a) */         \
  "v_sub_u32 " m ", " m ", " a "\n\t"    /* o5 = m - o6 (in m) */          \
  "v_sub_u32 " b ", " b ", " z "\n\t"    /* M(x1, 277) - z5 */             \
  "v_add_u32 " b ", " b ", " m "\n\t"    /* o4 (in b) */                   \
  "v_sub_u32 " x0 ", " e0 ", " x1 "\n\t" /* y7 = e0 - o7, o7 = x1 */       \
  "v_add_u32 " e0 ", " e0 ", " x1 "\n\t" /* y0 */                          \
  "v_sub_u32 " z ", " e1 ", " a "\n\t"   /* y6 = e1 - o6 */                \
  "v_add_u32 " e1 ", " e1 ", " a "\n\t"  /* y1 */                          \
  "v_sub_u32 " x1 ", " e2 ", " m "\n\t"  /* y5 = e2 - o5 */                \
  "v_add_u32 " e2 ", " e2 ", " m "\n\t"  /* y2 */                          \
  "v_sub_u32 " a ", " x2 ", " b "\n\t"   /* y3 = e3 - o4 */                \
  "v_add_u32 " b ", " x2 ", " b "\n\t"   /* y4 = e3 + o4 */

__device__ __forceinline__ uint2 idct8_lo3_row_px(int x0, int x1, int x2, const IdctK& K) {
  uint32_t o0, o1;
  int a, m, z, b, e0, e1, e2;
  asm(MIRTJ_MAD("%[a]", "%[x2]", "%[k362]") MIRTJ_MAD("%[m]", "%[x1]", "%[k362]") MIRTJ_MAD("%[z]", "%[x1]", "%[k473]")
      MIRTJ_MAD("%[b]", "%[x1]", "%[k277]")
      "s_nop " MIRTJ_STR(MIRTJ_NOP_LO3) "\n\t"
      MIRTJ_LO3_TAIL("%[x0]", "%[x1]", "%[x2]", "%[a]", "%[m]", "%[z]", "%[b]", "%[e0]", "%[e1]", "%[e2]")
      MIRTJ_PX_PACK("%[e0]", "%[e1]", "%[e2]", "%[a]", "%[b]", "%[x1]", "%[z]", "%[x0]")
      : [o0] "=&v"(o0), [o1] "=&v"(o1), [a] "=&v"(a), [m] "=&v"(m), [z] "=&v"(z), [b] "=&v"(b), [e0] "=&v"(e0),
        [e1] "=&v"(e1), [e2] "=&v"(e2), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2)
      : MIRTJ_K_OPERANDS);
  return make_uint2(o0, o1);
}

// column pass: q0, q1, q2 = rows 0, 1, 2 of a column pair; kOdd picks the column
template <bool kDc, bool kOdd>
__device__ __forceinline__ void idct8_lo3_col(uint32_t q0, uint32_t q1, uint32_t q2, int (&y)[8], const IdctK& K) {
  int x0, x1, x2, a, m, z, b, e0, e1, e2;
#define MIRTJ_LO3_COL_HEAD_EVEN                                                                               \
  "v_bfe_i32 %[x0], %[q0], 0, 16\n\t"                                                                         \
  "v_bfe_i32 %[x1], %[q1], 0, 16\n\t"                                                                         \
  "v_bfe_i32 %[x2], %[q2], 0, 16\n\t"                                                                         \
  MIRTJ_LO3_COL_PRODUCTS
#define MIRTJ_LO3_COL_HEAD_ODD                                                                                \
  "v_ashrrev_i32 %[x0], 16, %[q0]\n\t"                                                                        \
  "v_ashrrev_i32 %[x1], 16, %[q1]\n\t"                                                                        \
  "v_ashrrev_i32 %[x2], 16, %[q2]\n\t"                                                                        \
  MIRTJ_LO3_COL_PRODUCTS
#define MIRTJ_LO3_COL_PRODUCTS                                                                                \
  MIRTJ_MAD("%[m]", "%[x1]", "%[k362]") MIRTJ_MAD("%[z]", "%[x1]", "%[k473]") MIRTJ_MAD("%[b]", "%[x1]", "%[k277]") \
  MIRTJ_MAD("%[a]", "%[x2]", "%[k362]")                                                                       \
  "s_nop " MIRTJ_STR(MIRTJ_NOP_LO3) "\n\t"
#define MIRTJ_LO3_COL_OPERANDS                                                                                      \
  [a] "=&v"(a), [m] "=&v"(m), [z] "=&v"(z), [b] "=&v"(b), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2),           \
      [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2)                                                                \
      : [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), MIRTJ_K_OPERANDS
  if (kDc)  // (column 0: even)
    asm(MIRTJ_LO3_COL_HEAD_EVEN
        "v_add_u32 %[x0], 4, %[x0]\n\t"
        MIRTJ_LO3_TAIL("%[x0]", "%[x1]", "%[x2]", "%[a]", "%[m]", "%[z]", "%[b]", "%[e0]", "%[e1]", "%[e2]")
        : MIRTJ_LO3_COL_OPERANDS);
  else if (kOdd)
    asm(MIRTJ_LO3_COL_HEAD_ODD
        MIRTJ_LO3_TAIL("%[x0]", "%[x1]", "%[x2]", "%[a]", "%[m]", "%[z]", "%[b]", "%[e0]", "%[e1]", "%[e2]")
        : MIRTJ_LO3_COL_OPERANDS);
  else
    asm(MIRTJ_LO3_COL_HEAD_EVEN
        MIRTJ_LO3_TAIL("%[x0]", "%[x1]", "%[x2]", "%[a]", "%[m]", "%[z]", "%[b]", "%[e0]", "%[e1]", "%[e2]")
        : MIRTJ_LO3_COL_OPERANDS);
  y[0] = e0; y[1] = e1; y[2] = e2; y[3] = a; y[4] = b; y[5] = x1; y[6] = z; y[7] = x0;
}

}  // namespace mirtj