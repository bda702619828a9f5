// rtj_spec_kernels.h — the speculative block index (gfx950).
//
// The exact index (rtj_index_kernels.h) computes, for every 3584-byte chunk, where a macroblock walk
// would leave the chunk from EVERY possible entry offset, because a chunk's true entry is only known
// once all chunks before it are resolved.  On streams as encoders make them that generality is rarely
// needed: a walk started at an arbitrary byte (as if a macroblock began there) falls into step with
// the true block chain within a few hundred bytes — a block ends where its token slots are covered,
// and the final zero run of a block covers them whatever the count was before — and, when luma and
// chroma blocks carry different numbers of raw bytes, into step with the macroblock phase as well.
//
//   k_spec_walk    one LANE per chunk (2048 bytes): starts kSpecLead bytes before its chunk, assuming a macroblock
//                  starts there, and runs RTjpeg_s2b's length rule (lib/RTjpeg.c:157-186, 2704) as a
//                  byte-serial state machine over lead + chunk, recording every block start as ONE BIT per stream
//                  byte (round 3; 16-bit positions through an LDS ring before).
//                  64 chunks advance in lockstep per wave: 15 vector instructions per byte for 64
//                  streams, against ~20 per BLOCK POSITION AND TYPE in k_index_summarize.
//   k_spec_verify  per packet: the last (macro)block start a walker saw in its lead, i.e. before its
//                  chunk, must be the last one its predecessor saw before the end of its span, which is
//                  the same byte (the walkers report both); chunk 0 starts at byte 0.  Two walkers that stand on the same byte in the same
//                  phase go on alike, so by induction every walker is on the true chain from that
//                  hand-over point on: the check is exact, not a heuristic — it either proves the whole
//                  packet's index or rejects it.
//                  Also counts blocks per chunk (prefix sums -> global block numbers) and turns the
//                  recorded starts into the block-offset index k_decode reads.
//
// Packets that fail the check (noisy content whose blocks end without a zero run, adversarial bytes,
// packets that end early, more than kSpecCap blocks in a walker's span) are put on a to-do list; the
// exact kernels run over that list only (a grid of kSpecFallbackRows rows that loops).
#pragma once
#include <hip/hip_runtime.h>

#include "rtj_common.h"
#include "rtj_decode_kernels.h"

#ifndef MIRTJ_SPEC_CHUNK
#define MIRTJ_SPEC_CHUNK 2048
#endif
#ifndef MIRTJ_SPEC_LEAD
#define MIRTJ_SPEC_LEAD 768
#endif

namespace mirtj {

constexpr int kSpecChunk = MIRTJ_SPEC_CHUNK;  // stream bytes a walker owns
constexpr int kSpecLead = MIRTJ_SPEC_LEAD;   // bytes it parses before them, from an assumed macroblock start
#ifndef MIRTJ_SPEC_LEAD_LONG
#define MIRTJ_SPEC_LEAD_LONG 1536
#endif
constexpr int kSpecLeadLong = MIRTJ_SPEC_LEAD_LONG;  // the lead of the second walker form, for content that falls into step late
#ifndef MIRTJ_SPEC_LEAD_VERY
#define MIRTJ_SPEC_LEAD_VERY 6144
#endif
// The third form: three chunks of lead.  Content with noise of +-32 at the highest quality (blocks of 32 bytes, few
// zero runs) needs it: half of the walks that start at an arbitrary byte are in step after 1.2 KB, 99 % after 5 KB
// (tools/analysis/lock_distance.py).  A walker then parses four times its chunk, still a third of what the exact index
// costs.  Walkers of a packet's first chunks, whose lead would start before the packet, start at byte 0 — on the true
// chain — with whatever lead that leaves them.
constexpr int kSpecLeadVery = MIRTJ_SPEC_LEAD_VERY;
constexpr int kSpecSpan = kSpecLeadVery + kSpecChunk;  // the longest span a walker parses
// which lead a plan's walkers use: the value of the policy's state word kSpecStLong
__host__ __device__ constexpr int spec_lead_of_level(uint32_t level) {
  return level == 0u ? kSpecLead : level == 1u ? kSpecLeadLong : kSpecLeadVery;
}
constexpr int kSpecTile = 128;    // bytes per lane fetched at a time (one cache line)
// What a walker records: one START BIT per byte of its span — bit (31 - i) of dword k says that a block starts at
// walker-relative position 32 k + i (the walker's first byte is taken to start one).  The walker shifts the "previous
// byte ended a block" mask into a register with one add-with-carry per byte and stores 16 bytes per 128-byte tile; the
// 64 walkers of a wave share a region laid out [tile][lane] (what a wave stores at a time is 1 KB of contiguous
// memory).  Round 2 recorded 16-bit positions through a 32-entry LDS ring: two more vector instructions and an LDS
// write per byte, flush checks every 16 bytes, 4 KB reserved per walker and a cap of 2048 blocks per span; the bits
// cost k_spec_verify a rank-to-position step per chunk (a popcount scan over at most 112 dwords) and need no cap.
constexpr int kSpecTilesMax = kSpecSpan / kSpecTile;  // tiles of the longest span: a walker's bits are 16 bytes per tile
__host__ __device__ constexpr size_t spec_bits_words(uint64_t walkers) {  // dwords of the bit buffer (+ a spare walker)
  return (size_t)((walkers + 1 + 63) / 64) * (size_t)kSpecTilesMax * 64u * 4u;
}
// dword k (positions 32 k .. 32 k + 31) of a walker's bits, in dwords from the start of the buffer: [wave of walkers]
// [tile][lane][4].  (A layout with each walker's bits contiguous, the walkers transposing four tiles at a time through
// LDS, was tried at the end of round 3 and did not pass the GPU tests; it is not in the tree.)
__host__ __device__ constexpr size_t spec_bits_dword(uint32_t walker, uint32_t k) {
  return (((size_t)(walker >> 6) * (size_t)kSpecTilesMax + (k >> 2)) * 64u + (walker & 63u)) * 4u + (k & 3u);
}
// The layout's bounds, checked where the layout is defined: every (walker, dword) of a
// launch of `walkers` walkers — the idle lanes of the last wave stand in for the spare walker `walkers` — lies inside
// spec_bits_words(walkers), and no two of them share a dword.  Evaluated at compile time for walker counts on both
// sides of a wave boundary; a change of either function that breaks this does not build.
constexpr bool spec_bits_layout_ok(uint32_t walkers) {
  const size_t words = spec_bits_words(walkers);
  const uint32_t rows = ((walkers + 1u + 63u) / 64u) * 64u;  // walker rows the kernels may touch (whole waves)
  const uint32_t kmax = (uint32_t)kSpecTilesMax * 4u;
  // (sampled, to stay inside the compiler's constexpr step limit: every walker row at the first and last dwords of its
  // first, second and last tile; every dword of the first, 64th and last row)
  size_t top = 0;
  for (uint32_t w = 0; w < rows; w++) {
    const uint32_t ks[6] = {0u, 3u, 4u, 7u, kmax - 4u, kmax - 1u};
    for (uint32_t k : ks) {
      const size_t d = spec_bits_dword(w, k);
      if (d >= words) return false;
      if (w + 1u < rows && d == spec_bits_dword(w + 1u, k)) return false;
      top = d > top ? d : top;
    }
  }
  const uint32_t ws[3] = {0u, rows > 64u ? 64u : rows - 1u, rows - 1u};
  for (uint32_t w : ws)
    for (uint32_t k = 0; k < kmax; k++) {
      const size_t d = spec_bits_dword(w, k);
      if (d >= words || (k + 1u < kmax && d == spec_bits_dword(w, k + 1u))) return false;
      top = d > top ? d : top;
    }
  return top + 1u == (size_t)rows * kmax && top + 1u <= words;  // dense: the last dword of the last row is the buffer's last
}
static_assert(spec_bits_layout_ok(1) && spec_bits_layout_ok(63) && spec_bits_layout_ok(64) && spec_bits_layout_ok(65) &&
                  spec_bits_layout_ok(200),
              "a walker's start bits leave the buffer");
constexpr uint64_t kSpecMinWalkers = 40000;  // below this (~100 MB of packets) the exact kernels index a batch faster:
                                            // a walker is one lane and runs ~0.35 ms whatever the batch (host policy)
constexpr int kSpecRingRow = 64 + 16;       // LDS bytes per lane: the start bits of the last four tiles (512 bytes of
                                            // stream: more than a macroblock, 384) + bank padding
static_assert(kSpecSpan % kSpecTile == 0 && kSpecSpan < 65536, "walker span: whole tiles, 16-bit positions");
static_assert(kSpecLead >= 6 * 64 + 64 && kSpecLead % kSpecTile == 0 && kSpecChunk % kSpecTile == 0, "the lead: at least one whole macroblock, whole tiles");
static_assert(kSpecLeadLong >= kSpecLead && kSpecLeadLong % kSpecTile == 0 && kSpecLeadVery >= kSpecLeadLong &&
                  kSpecLeadVery % kSpecTile == 0 && kSpecLeadVery % kSpecChunk == 0,
              "leads: whole tiles, growing; the longest a whole number of chunks (a walker that cannot have it starts at byte 0)");

struct SpecChunkDev {
  uint32_t frame;  // index into the plan's frames
  uint32_t c;      // chunk number within the packet
};
constexpr int kSpecFallbackRows = 64;
constexpr int kSpecRepairGrid = 4096;   // waves of k_spec_repair (they loop over the list of chunks to walk again)
constexpr int kSpecPauseLaunches = 64;  // launches a plan goes without speculation after two in which every packet was refused
// k_spec_policy's state words (per plan, on the device)
enum { kSpecStLost = 0,    // launches in a row that were lost
       kSpecStPause = 1,   // launches left without speculation
       kSpecStLong = 2,    // 1: the walkers use the long lead
       kSpecStQuiet = 3,   // long lead: launches in a row that needed next to no repairs
       kSpecStWords = 4 };
constexpr int kSpecQuietLaunches = 16;  // that many of them in a row and the plan tries the short lead again
#ifndef MIRTJ_SPEC_VER_THREADS
#define MIRTJ_SPEC_VER_THREADS 512
#endif
#ifndef MIRTJ_SPEC_VER_BATCH
#define MIRTJ_SPEC_VER_BATCH 4
#endif
constexpr int kSpecVerWindow = 512;  // ranks a wave of k_spec_verify orders in LDS at a time (a chunk of ordinary content: ~200)
constexpr int kSpecVerBatch = MIRTJ_SPEC_VER_BATCH;  // chunks whose records a wave of k_spec_verify copies side by side
constexpr int kSpecVerThreads = MIRTJ_SPEC_VER_THREADS;  // k_spec_verify: chunks of a packet handled side by side  // grid rows of the exact kernels when they only serve refused packets

// One byte of the walker's state machine, spelled out: 15 vector instructions (the compiler's version of the same C++
// had 21-22: it keeps lane masks as 0/1 integers and splits the counters; round 2's, with 16-bit records, 17 and an LDS
// write).  State: u = units of the current block so far, q = 5 - (block number within the macroblock), rb = DC + raw
// bytes of the current block's type, em = lane mask "the previous byte ended a block" = "a block starts at this byte",
// bits = the start bits of the 32 bytes in hand (shifted in with one add-with-carry).  The byte's token value t = byte - 63
// and max(1, t) arrive precomputed (tm, wtm) and the macro computes them for the NEXT byte (byte NSEL of word wn): that
// is independent work to put into the two wait states gfx950 wants between a vector compare that writes a lane mask
// and a vector instruction that reads it (the assembler pads nothing inside an asm statement).  Checked gaps: vcc (a) ->
// s_and: b, c; ma (b) -> g: c .. f; em (i) -> l: j + s_nop; mx (m) -> p: n, o; my (n) -> q: o, p.
#define MIRTJ_SPEC_STEP(NSEL)                                                                                     \
  asm("v_cmp_eq_i32_e32 vcc, %[kn64], %[tm]\n\t"        /* a: the byte is 0xFF */                                  \
      "v_cmp_lt_i32_e64 %[ma], %[u], %[rb]\n\t"         /* b: a DC or raw byte */                                  \
      "v_addc_co_u32_e64 %[bits], %[mz], %[bits], %[bits], %[em]\n\t" /* c: bits = bits << 1 | starts-here */     \
      "s_and_b64 vcc, vcc, %[em]\n\t"                   /* 0xFF as a block's first byte: one-byte block */        \
      "v_sub_u32_sdwa %[tn], sext(%[wn]), %[k63] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" NSEL             \
      " src1_sel:DWORD\n\t"                             /* e: the next byte's token value */                     \
      "v_cndmask_b32_e64 %[wr], 1, 64, vcc\n\t"         /* f */                                                    \
      "v_cndmask_b32_e64 %[wt], %[wtm], %[wr], %[ma]\n\t" /* g: token weight (lib/RTjpeg.c:171-182), or 1 / 64 */ \
      "v_add_u32_e32 %[u], %[u], %[wt]\n\t"             /* h */                                                    \
      "v_cmp_lt_i32_e64 %[em], 63, %[u]\n\t"            /* i: 64 units: the block ends with this byte */           \
      "v_max_i32_e32 %[wtn], 1, %[tn]\n\t"              /* j: the next byte's token weight */                     \
      "s_nop 0\n\t"                                                                                                \
      "v_subb_co_u32_e64 %[q], vcc, %[q], 0, %[em]\n\t" /* l */                                                    \
      "v_cmp_lt_i32_e64 %[mx], %[q], 0\n\t"             /* m: past the macroblock's last block */                  \
      "v_cmp_lt_u32_e64 %[my], %[q], 2\n\t"             /* n: a chroma block is next */                            \
      "v_cndmask_b32_e64 %[u], %[u], 0, %[em]\n\t"      /* o */                                                    \
      "v_cndmask_b32_e64 %[q], %[q], 5, %[mx]\n\t"      /* p */                                                    \
      "v_cndmask_b32_e64 %[rb], %[lb], %[cb], %[my]"    /* q */                                                    \
      : [u] "+v"(u), [q] "+v"(q), [rb] "+v"(rb), [em] "+s"(em), [bits] "+v"(bits_), [tn] "=&v"(tn_), [wtn] "=&v"(wtn_), \
        [wt] "=&v"(wt_), [wr] "=&v"(wr_), [ma] "=&s"(ma_), [mx] "=&s"(mx_), [my] "=&s"(my_), [mz] "=&s"(mz_)         \
      : [tm] "v"(tm_), [wtm] "v"(wtm_), [wn] "v"(wn_), [k63] "v"(k63), [kn64] "v"(kn64), [lb] "v"(lb), [cb] "v"(cb)  \
      : "vcc")

// The same when every packet of the launch has lb8 == cb8 (two thirds of the qualities): all blocks parse
// alike, no phase to track, 10 vector instructions.
#define MIRTJ_SPEC_STEP1(NSEL)                                                                                    \
  asm("v_cmp_eq_i32_e32 vcc, %[kn64], %[tm]\n\t"                                                                   \
      "v_cmp_lt_i32_e64 %[ma], %[u], %[rb]\n\t"                                                                    \
      "v_addc_co_u32_e64 %[bits], %[mz], %[bits], %[bits], %[em]\n\t"                                              \
      "s_and_b64 vcc, vcc, %[em]\n\t"                                                                              \
      "v_sub_u32_sdwa %[tn], sext(%[wn]), %[k63] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" NSEL             \
      " src1_sel:DWORD\n\t"                                                                                       \
      "v_cndmask_b32_e64 %[wr], 1, 64, vcc\n\t"                                                                    \
      "v_cndmask_b32_e64 %[wt], %[wtm], %[wr], %[ma]\n\t"                                                          \
      "v_add_u32_e32 %[u], %[u], %[wt]\n\t"                                                                        \
      "v_cmp_lt_i32_e64 %[em], 63, %[u]\n\t"                                                                       \
      "v_max_i32_e32 %[wtn], 1, %[tn]\n\t"                                                                         \
      "s_nop 0\n\t"                                                                                                \
      "v_cndmask_b32_e64 %[u], %[u], 0, %[em]"                                                                     \
      : [u] "+v"(u), [em] "+s"(em), [bits] "+v"(bits_), [tn] "=&v"(tn_), [wtn] "=&v"(wtn_), [wt] "=&v"(wt_),       \
        [wr] "=&v"(wr_), [ma] "=&s"(ma_), [mz] "=&s"(mz_)                                                          \
      : [tm] "v"(tm_), [wtm] "v"(wtm_), [wn] "v"(wn_), [k63] "v"(k63), [kn64] "v"(kn64), [rb] "v"(rb)              \
      : "vcc")

// What a launch's walkers zero before anything else of the launch looks at it: the lists k_spec_verify fills (walkers
// to repair, packets refused) and the per-walker "on the repair list" marks.  (Three memsets in front of the walkers
// before: three more dispatches per launch, which a launch of 1024 pictures feels.)
struct SpecReset {
  uint32_t* nfix;
  uint32_t* ntodo;
  uint8_t* fixflag;
};

template <bool PHASE, int LEAD>
__device__ __forceinline__ void spec_walk_body(uint8_t* __restrict__ s_ring, const FrameDev* __restrict__ frames,
                                               const SpecChunkDev* __restrict__ chunks, uint32_t total,
                                               const uint8_t* __restrict__ stream, const QTab* __restrict__ lut,
                                               uint32_t* __restrict__ recbits, uint32_t* __restrict__ nrec,
                                               uint32_t* __restrict__ wstart, uint2* __restrict__ hand,
                                               const SpecReset rs) {
  constexpr int kSpan = LEAD + kSpecChunk;
  static_assert(kSpan % kSpecTile == 0 && kSpan < 65536, "walker span: whole tiles, 16-bit positions");
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x * 64u + (uint32_t)lane;
  const bool act = g < total;
  if (act) rs.fixflag[g] = 0;
  if (g == 0u) {
    *rs.nfix = 0u;
    *rs.ntodo = 0u;
  }
  const SpecChunkDev sc = chunks[act ? g : total - 1u];  // idle lanes shadow the last chunk and store nothing
  const FrameDev f = frames[sc.frame];
  const int lb = lut[f.qidx].lb8 + 1, cb = lut[f.qidx].cb8 + 1;  // DC + raw bytes of a luma / chroma block
  // first byte parsed: LEAD bytes before the chunk, or the packet's first byte where the packet does not reach that
  // far back (then the walker is on the true chain from its first byte)
  const uint32_t cstart = sc.c * (uint32_t)kSpecChunk;
  const uint32_t start = cstart > (uint32_t)LEAD ? cstart - (uint32_t)LEAD : 0u;
  const int take_tile = (int)((cstart - start) / (uint32_t)kSpecTile);  // the tile the walker's own chunk begins with
  const int end_tile = take_tile + kSpecChunk / kSpecTile;             // ... and the first tile past it
  const uint8_t* gp = stream + f.data_off + start;
  const uint32_t sh = (uint32_t)((uintptr_t)gp & 3u);
  const uint32_t* g4 = (const uint32_t*)(gp - sh);
  // bytes of the packet from g4[0] on (start < data_len for every chunk but an empty packet's only one)
  const int avail = (int)f.data_len - (int)start + (int)sh;

  // ---- one tile = 32 dwords + the one after them (for the funnel shift), bytes past the packet read 0 ----
  uint32_t buf[33];
  auto request = [&](int t) {
    const int d0 = t * (kSpecTile / 4);
    if (__all(avail >= 4 * (d0 + 33))) {  // every lane's tile lies inside its packet
#pragma unroll
      for (int k = 0; k < 33; k++) buf[k] = g4[d0 + k];
    } else {
#pragma unroll
      for (int k = 0; k < 33; k++) {
        const int rem = avail - 4 * (d0 + k);
        uint32_t v = 0;
        if (rem > 0) {
          v = g4[d0 + k];
          if (rem < 4) v &= (1u << (8 * rem)) - 1u;
        }
        buf[k] = v;
      }
    }
  };

  // ---- the length rule as a branch-free state machine.  A block is complete after 64 units: DC and the
  //      raw bytes count 1 each, a token its weight.  u = units of the current block so far, rb = its
  //      1 + bt8 DC/raw bytes, q = 5 - its number within the macroblock ----
  int u = 0, rb = lb;
  uint32_t q = 5;        // 5 - block number within the macroblock
  uint64_t em = ~0ull;   // the walker's first byte is taken to be a macroblock's first
  // idle lanes of the last wave own the spare walker after the last one
  const uint32_t gw = act ? g : total;
  uint4* const ring = (uint4*)(s_ring + lane * kSpecRingRow);  // the start bits of the last four tiles
  const int kn64 = -64, k63 = 63;
  uint32_t cnt = 0;  // block starts recorded so far (counted tile by tile)

  // (index << 16 | position) of the last unit-aligned block start below the start of tile `lt` (walker-relative; the
  // unit is the macroblock, or the block when lb8 == cb8: k_spec_verify).  Called between tiles: cnt counts the starts
  // below the limit, the current block has number 5 - q in its macroblock, and if the byte before the limit ended a
  // block (the lane's bit of em), a start AT the limit is pending that is in neither.  The unit start wanted is at
  // most a macroblock (384 bytes) back: inside the four tiles the ring holds.
  const bool by_block = lb == cb;
  auto last_aligned_below = [&](int lt) -> uint32_t {
    const int d = (int)((em >> lane) & 1ull);
    int r = by_block ? 0 : 5 - (int)q - d;  // starts between the one wanted and the limit
    r = r < 0 ? r + 6 : r;
    int left = r;
    for (int back = 1; back <= 4 && back <= lt; back++) {
      const int tile = lt - back;
      const uint4 v = ring[tile & 3];
      const uint32_t dw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        uint32_t m = dw[j];
        const int c = __builtin_popcount(m);
        if (left >= 0 && left < c) {
          for (int k = 0; k < left; k++) m &= m - 1u;  // drop the `left` latest starts of this dword (lowest bits)
          const uint32_t pos = (uint32_t)(tile * kSpecTile + 32 * j + 31 - __builtin_ctz(m));
          return ((cnt - 1u - (uint32_t)r) << 16) | pos;
        }
        left -= c;
      }
    }
    return 0u;  // nothing but the walker's own first byte (record 0, position 0)
  };
  uint32_t take = 0;  // where this walker's chunk takes over from its predecessor (chunk 0: byte 0, record 0)
  uint32_t tail = 0;  // where the next chunk has to take over: the last unit start below the end of this one
  uint32_t cnt_end = 0;  // block starts below the end of the walker's chunk
  uint32_t* const out_bits = recbits + spec_bits_dword(gw, 0);
  uint4 tb = make_uint4(0, 0, 0, 0);  // the tile's start bits

  request(0);
  for (int t = 0; t < kSpan / kSpecTile; t++) {
    // the tile is parsed out of registers (fully unrolled: 128 byte steps); staging it in LDS for a
    // smaller loop body capped the kernel at 11 waves per CU
    // the two hand-over points, per lane (walkers that started at byte 0 reach theirs earlier than the others)
    if (__any(t == take_tile || t == end_tile)) {
      const uint32_t v = last_aligned_below(t);
      if (t == take_tile) take = sc.c ? v : 0u;
      if (t == end_tile) {
        tail = v;
        cnt_end = cnt;
      }
    }
    uint32_t cur[33];
#pragma unroll
    for (int k = 0; k < 33; k++) cur[k] = buf[k];
    // the bits of the tile before leave now, BEFORE the next tile's loads are queued: loads and stores
    // complete in order, so a store queued after those loads would have to be waited for with them
    if (t > 0) *(uint4*)(out_bits + (size_t)(t - 1) * 256u) = tb;
    if (t + 1 < kSpan / kSpecTile) request(t + 1);  // in flight while this tile is parsed
    // the first byte's token value and weight; every step computes them for the byte after it
    int tm_, wtm_;
    {
      const uint32_t w0 = __builtin_amdgcn_alignbyte(cur[1], cur[0], sh);
      tm_ = (int)(int8_t)(w0 & 0xFFu) - 63;
      wtm_ = tm_ > 1 ? tm_ : 1;
    }
    uint32_t bw[4];
#pragma unroll
    for (int i = 0; i < kSpecTile / 16; i++) {
      const uint32_t wd[5] = {__builtin_amdgcn_alignbyte(cur[4 * i + 1], cur[4 * i], sh),
                              __builtin_amdgcn_alignbyte(cur[4 * i + 2], cur[4 * i + 1], sh),
                              __builtin_amdgcn_alignbyte(cur[4 * i + 3], cur[4 * i + 2], sh),
                              __builtin_amdgcn_alignbyte(cur[4 * i + 4], cur[4 * i + 3], sh),
                              // the word after them, for the step that looks one byte ahead (the tile's last byte
                              // looks at a byte that is never used: the next tile starts afresh)
                              i + 1 < kSpecTile / 16 ? __builtin_amdgcn_alignbyte(cur[4 * i + 5], cur[4 * i + 4], sh) : 0u};
      uint32_t bits_ = (i & 1) ? bw[i >> 1] : 0u;  // 32 steps shift every older bit out: no need to clear
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const uint32_t wn_ = wd[(b + 1) >> 2];
        int tn_, wtn_, wt_, wr_;
        uint64_t ma_, mx_, my_, mz_;
        if (PHASE) {
          switch ((b + 1) & 3) {
            case 0: MIRTJ_SPEC_STEP("BYTE_0"); break;
            case 1: MIRTJ_SPEC_STEP("BYTE_1"); break;
            case 2: MIRTJ_SPEC_STEP("BYTE_2"); break;
            default: MIRTJ_SPEC_STEP("BYTE_3"); break;
          }
        } else {
          switch ((b + 1) & 3) {
            case 0: MIRTJ_SPEC_STEP1("BYTE_0"); break;
            case 1: MIRTJ_SPEC_STEP1("BYTE_1"); break;
            case 2: MIRTJ_SPEC_STEP1("BYTE_2"); break;
            default: MIRTJ_SPEC_STEP1("BYTE_3"); break;
          }
        }
        tm_ = tn_;
        wtm_ = wtn_;
      }
      bw[i >> 1] = bits_;
    }
    tb = make_uint4(bw[0], bw[1], bw[2], bw[3]);
    ring[t & 3] = tb;
    cnt += (uint32_t)(__builtin_popcount(bw[0]) + __builtin_popcount(bw[1]) + __builtin_popcount(bw[2]) +
                      __builtin_popcount(bw[3]));
  }
  *(uint4*)(out_bits + (size_t)(kSpan / kSpecTile - 1) * 256u) = tb;
  if (end_tile == kSpan / kSpecTile) {  // (walkers with the whole lead end with the loop)
    tail = last_aligned_below(kSpan / kSpecTile);
    cnt_end = cnt;
  }
  if (act) {
    nrec[g] = cnt_end;  // block starts the walker saw up to the end of its chunk
    wstart[g] = start;
    hand[g] = make_uint2(take, tail);  // .y: what the next chunk must take over from
  }
}

// One form, named by the host (plans without a policy: MI_RTJ_SPEC = 1 / 3 / 4, tests).
template <bool PHASE, int LEAD>
__global__ __launch_bounds__(64) void k_spec_walk(const FrameDev* __restrict__ frames,
                                                   const SpecChunkDev* __restrict__ chunks, uint32_t total,
                                                   const uint8_t* __restrict__ stream,
                                                   const QTab* __restrict__ lut, uint32_t* __restrict__ recbits,
                                                   uint32_t* __restrict__ nrec, uint32_t* __restrict__ wstart,
                                                   uint2* __restrict__ hand, const SpecReset rs) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[64 * kSpecRingRow];
  spec_walk_body<PHASE, LEAD>(s_ring, frames, chunks, total, stream, lut, recbits, nrec, wstart, hand, rs);
}

// The form the plan's policy state names (none while the speculation is paused, k_spec_policy): ONE dispatch whatever
// the lead.  (Round 3 first queued the three forms as three kernels of which two returned at once.)
template <bool PHASE>
__global__ __launch_bounds__(64) void k_spec_walk_any(const FrameDev* __restrict__ frames,
                                                       const SpecChunkDev* __restrict__ chunks, uint32_t total,
                                                       const uint8_t* __restrict__ stream,
                                                       const QTab* __restrict__ lut, uint32_t* __restrict__ recbits,
                                                       uint32_t* __restrict__ nrec, uint32_t* __restrict__ wstart,
                                                       uint2* __restrict__ hand, const SpecReset rs,
                                                       const uint32_t* __restrict__ state) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[64 * kSpecRingRow];
  if (state[kSpecStPause]) return;
  const uint32_t level = state[kSpecStLong];  // wave-uniform
  if (level == 0u)
    spec_walk_body<PHASE, kSpecLead>(s_ring, frames, chunks, total, stream, lut, recbits, nrec, wstart, hand, rs);
  else if (level == 1u)
    spec_walk_body<PHASE, kSpecLeadLong>(s_ring, frames, chunks, total, stream, lut, recbits, nrec, wstart, hand, rs);
  else
    spec_walk_body<PHASE, kSpecLeadVery>(s_ring, frames, chunks, total, stream, lut, recbits, nrec, wstart, hand, rs);
}

// One workgroup per packet: the chain check, the block numbering, and — the numbering is all it needs —
// the block-offset index itself.  A packet that fails goes on the exact kernels' to-do list (they then
// rewrite its index).
__global__ __launch_bounds__(kSpecVerThreads) void k_spec_verify(const FrameDev* __restrict__ frames,
                                                      const uint32_t* __restrict__ spec_base,
                                                      const QTab* __restrict__ lut,
                                                      const uint32_t* __restrict__ recbits,
                                                      const uint32_t* __restrict__ nrec, uint32_t* __restrict__ blkoff,
                                                      uint32_t* __restrict__ ok, uint32_t* __restrict__ todo,
                                                      uint32_t* __restrict__ ntodo, const uint32_t* __restrict__ state,
                                                      const uint32_t* __restrict__ wstart,
                                                      const uint2* __restrict__ hand, uint2* __restrict__ fix,
                                                      uint32_t* __restrict__ nfix, uint8_t* __restrict__ fixflag, int pass) {
  if (state && state[kSpecStPause]) {  // paused (k_spec_policy): nothing was walked, nothing is proven
    if (threadIdx.x == 0) ok[blockIdx.x] = 0;
    return;
  }
  if (pass == 2 && ok[blockIdx.x] != 2u) return;  // the second pass only looks at packets with repaired chunks
  __shared__ uint32_t s_wave[kSpecVerThreads / 64], s_carry[3];  // carry: [1] blocks so far, [2] bad
  __shared__ uint32_t s_i0[kSpecVerThreads], s_base[kSpecVerThreads], s_cnt[kSpecVerThreads];
  __shared__ uint16_t s_pos[kSpecVerThreads / 64][kSpecVerWindow];  // per wave: block offsets in rank order, on their way out
  const FrameDev f = frames[blockIdx.x];
  const uint32_t sc0 = spec_base[blockIdx.x], nsc = spec_base[blockIdx.x + 1] - sc0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // lb8 == cb8: every block parses alike, walkers fall into step with the BLOCK chain but keep whatever
  // macroblock phase they assumed; the unit of the chain check is then the block, and the phase follows
  // from the global block number
  const uint32_t last = 6u * f.nmb;  // the index holds one entry past the last block: the end position
  uint32_t* const out = blkoff + f.blk_base;
  if (tid == 0) {
    s_carry[1] = 0;
    s_carry[2] = 0;
  }
  __syncthreads();
  for (uint32_t c0 = 0; c0 < nsc; c0 += kSpecVerThreads) {
    const uint32_t c = c0 + (uint32_t)tid;
    uint32_t cnt = 0, i0 = 0, bad = 0, soft = 0;
    if (c < nsc) {
      // The walkers left, per chunk, the last unit-aligned record before the chunk (hand.x: where it takes
      // over; record 0, its arbitrary first byte, if it saw no other) and the last one before the end of the
      // span (hand.y: where the next chunk has to take over), each as index << 16 | position.
      const uint32_t n = nrec[sc0 + c];  // block starts in the walker's span (fewer than 2^16: a span is under 4 KB)
      const uint2 mine = hand[sc0 + c];
      i0 = c ? mine.x >> 16 : 0u;
      uint32_t i1 = n;  // the packet's last walker: everything it saw (bytes past the packet read as 0)
      if (c + 1 < nsc) {
        const uint32_t h = wstart[sc0 + c + 1] + (hand[sc0 + c + 1].x & 0xFFFFu);  // where chunk c + 1 took over
        const uint32_t t = wstart[sc0 + c] + (mine.y & 0xFFFFu);                    // where it had to
        i1 = mine.y >> 16;
        if (h != t) {
          if (pass == 1 && !bad) {
            // the next walker had not fallen into step yet: it is re-walked (k_spec_repair) from the last
            // (macro)block start this one saw, and the packet gets a second pass
            soft = 1;
            fix[atomicAdd(nfix, 1u)] = make_uint2(sc0 + c + 1u, t);
            fixflag[sc0 + c + 1u] = 1;  // (k_spec_repair: walkers to repair that follow one another are one wave's job)
          } else {
            bad = 1;
          }
        }
      }
      if (bad || i1 < i0 || i1 > n) bad = 1;
      else cnt = i1 - i0;
    }
    // blocks before this chunk: scan over the tile + carry
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane == 63) s_wave[wv] = incl;
    const uint32_t anybad = (__any(bad) ? 1u : 0u) | (__any(soft) ? 2u : 0u);
    __syncthreads();
    uint32_t before = s_carry[1] + incl - cnt;
    for (int k = 0; k < wv; k++) before += s_wave[k];
    s_i0[tid] = c < nsc && c ? hand[sc0 + c].x & 0xFFFFu : 0u;  // where the chunk's first block starts, walker-relative
    s_base[tid] = before;
    s_cnt[tid] = cnt;
    __syncthreads();
    if (lane == 0 && anybad) atomicOr(&s_carry[2], anybad);
    if (tid == kSpecVerThreads - 1) s_carry[1] = before + cnt;
    // recorded starts -> block offsets, one wave per chunk of the tile (harmless if the packet fails later:
    // counts of refused chunks are 0, everything is clipped to the packet's own index, and the exact
    // kernels rewrite it)
    const uint32_t tile_n = min((uint32_t)kSpecVerThreads, nsc - c0);
    // A chunk's block starts are bits in its walker's record, from the hand-over point (at most a macroblock, 384
    // bytes, before the chunk) to the end of the chunk: at most 77 dwords whatever the walker's lead, two per lane.
    // Rank of a bit = starts between the hand-over point and it (popcounts, one wave scan per half); the ranks
    // 0 .. m - 1 are this chunk's blocks.  A wave works on kSpecVerBatch chunks at a time so that it waits for memory
    // once per batch.
    constexpr uint32_t kWaves = kSpecVerThreads / 64;
    static_assert((kSpecChunk + kEntries) / 32 + 2 + 3 <= 128, "two dwords of start bits per lane");
    // (a wave taking kSpecVerBatch NEIGHBOURING chunks, whose walkers' bits share lines, measured the same:
    // profiles/r03/ab_verify_adjacent_chunks.txt)
    for (uint32_t j0 = (uint32_t)wv; j0 < tile_n; j0 += kSpecVerBatch * kWaves) {
      uint32_t base[kSpecVerBatch], m[kSpecVerBatch], start[kSpecVerBatch], first[kSpecVerBatch];
      uint32_t w0[kSpecVerBatch], w1[kSpecVerBatch];
#pragma unroll
      for (int u = 0; u < kSpecVerBatch; u++) {
        const uint32_t j = min(j0 + (uint32_t)u * kWaves, tile_n - 1u);
        const bool have = j0 + (uint32_t)u * kWaves < tile_n;
        base[u] = s_base[j];
        m[u] = have && base[u] <= last ? min(s_cnt[j], last + 1u - base[u]) : 0u;  // clipped to the packet's own index
        start[u] = wstart[sc0 + c0 + j];
        first[u] = s_i0[j];
        const uint32_t R = sc0 + c0 + j;  // the walker whose starts are copied
#ifdef MIRTJ_EXP_VER_FROM_ZERO  // A/B (short leads only): every bit of the span, ranks counted from the walker's first byte
        w0[u] = m[u] ? recbits[spec_bits_dword(R, (uint32_t)lane)] : 0u;
        w1[u] = m[u] ? recbits[spec_bits_dword(R, (uint32_t)lane + 64u)] : 0u;
        first[u] = c0 + j < nsc && c0 + j ? hand[sc0 + c0 + j].x >> 16 : 0u;
        continue;
#endif
        // dwords from the 16-byte piece that holds the hand-over point on: four lanes then read one piece (the pieces of
        // a walker lie 1 KB apart); started at the hand-over point's own dword, lane quads straddled two pieces and the
        // kernel took 2.04 ms per 16,384 pictures instead of 1.60 (profiles/r03/ab_verify_bit_addressing.txt)
        const uint32_t kt = first[u] >> 5;  // the dword of the hand-over point
        const uint32_t k0 = (kt & ~3u) + (uint32_t)lane, k1 = k0 + 64u;
        w0[u] = m[u] && k0 >= kt && k0 < (uint32_t)(kSpecTilesMax * 4) ? recbits[spec_bits_dword(R, k0)] : 0u;
        w1[u] = m[u] && k1 < (uint32_t)(kSpecTilesMax * 4) ? recbits[spec_bits_dword(R, k1)] : 0u;
        if (k0 == kt) w0[u] &= 0xFFFFFFFFu >> (first[u] & 31u);  // starts before the hand-over point (bit 31 = first byte)
      }
#pragma unroll
      for (int u = 0; u < kSpecVerBatch; u++) {
        if (m[u] == 0u) continue;  // wave-uniform
        const uint32_t c0_ = (uint32_t)__builtin_popcount(w0[u]), c1_ = (uint32_t)__builtin_popcount(w1[u]);
        const uint32_t in0 = wave_incl_scan(c0_);
        const uint32_t tot0 = (uint32_t)__builtin_amdgcn_readlane((int)in0, 63);
        const uint32_t in1 = wave_incl_scan(c1_) + tot0;
        uint32_t* const o = out + base[u];
#ifdef MIRTJ_EXP_VER_FROM_ZERO
        const uint32_t q0 = 32u * (uint32_t)lane, q1 = q0 + 2048u;
#else
        const uint32_t q0 = (first[u] & ~127u) + 32u * (uint32_t)lane, q1 = q0 + 2048u;  // walker-relative position of the dwords' first byte
#endif
        // Ranks are scattered over the lanes (a lane owns the starts of its 32 bytes), the index wants them in order:
        // written straight to memory, a store instruction touched a dozen 64-byte pieces for 64 offsets and the kernel
        // took 2.1 ms per 16384 pictures.  The offsets are therefore put in rank order in LDS (scattered 16-bit writes
        // are cheap there) and leave as whole 256-byte rows, kSpecVerWindow ranks at a time (one window covers a chunk
        // of ordinary content; a chunk of one-byte blocks takes five).
        uint16_t* const sp = s_pos[wv];
        for (uint32_t r0 = 0; r0 < m[u]; r0 += (uint32_t)kSpecVerWindow) {
          const uint32_t span_ = min((uint32_t)kSpecVerWindow, m[u] - r0);
#ifdef MIRTJ_EXP_VER_FROM_ZERO
          int rk0 = (int)(in0 - c0_) - (int)first[u] - (int)r0, rk1 = (int)(in1 - c1_) - (int)first[u] - (int)r0;
#else
          int rk0 = (int)(in0 - c0_) - (int)r0, rk1 = (int)(in1 - c1_) - (int)r0;
#endif
          uint32_t m0 = w0[u], m1 = w1[u];
          // both dwords of a lane in one loop: it runs for the fullest dword of the wave (4-5 starts where a
          // macroblock's chroma blocks lie, 2 in luma), not for the sum of the two
          while (__any((m0 | m1) != 0u)) {
            if (m0) {
              const int z = __builtin_clz(m0);  // bit 31 is the dword's first byte
              if ((uint32_t)rk0 < span_) sp[rk0] = (uint16_t)(q0 + (uint32_t)z);
              rk0++;
              m0 &= ~(0x80000000u >> z);
            }
            if (m1) {
              const int z = __builtin_clz(m1);
              if ((uint32_t)rk1 < span_) sp[rk1] = (uint16_t)(q1 + (uint32_t)z);
              rk1++;
              m1 &= ~(0x80000000u >> z);
            }
          }
          // (one wave, and a wave's LDS operations complete in order: the reads below see the writes above; the
          // compiler is told not to move them across)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          for (uint32_t k = (uint32_t)lane; k < span_; k += 64u) o[r0 + k] = start[u] + (uint32_t)sp[k];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
    __syncthreads();
  }
  // every block start and the end position must be there: indices 0 .. 6 * nmb
  if (tid == 0) {
    // 1 proven, 2 some walkers are being repaired (first pass only), 0 refused
    uint32_t v = (s_carry[2] & 1u) ? 0u : (s_carry[2] & 2u) ? 2u : s_carry[1] >= last + 1u ? 1u : 0u;
    ok[blockIdx.x] = v;
    if (!v) todo[atomicAdd(ntodo, 1u)] = blockIdx.x;
  }
}

// A walker that had not fallen into step by the end of its lead is walked again, this time from a byte that is known
// to start a (macro)block if its predecessor is right (the second k_spec_verify pass checks exactly that): one WAVE per
// such walker, the block-by-block walker of rtj_decode_kernels.h (a lane-serial walker would take as long for one chunk
// as k_spec_walk takes for all).  Content whose walks lock late — noise of +-24 and more at the highest quality —
// needs two things round 2's repair did not do, and without them its packets were refused one and all:
//   * walkers out of step that FOLLOW one another are one wave's job: the byte the second is to start from was read off
//     the first one's old records, which are what is being replaced.  The wave of a run's first walker goes on into the
//     next chunk for as long as that chunk's walker is marked as well, each time from the hand-over point it has just
//     found itself; the waves of the others leave;
//   * a walker that is NOT marked may still be wrong: it agreed with its predecessor's old records because the two
//     walks had found each other before either found the true chain.  So behind a repair the wave looks at the next
//     walker itself and goes on while that one does not take over from the new hand-over point — unless it is the first
//     of another wave's run (that wave started from what this one has just replaced: such a packet fails the second
//     proof pass and goes to the exact kernels; rare).
// The proof stays k_spec_verify's alone: this kernel only rewrites records.  grid: any; waves loop over the list.
__global__ __launch_bounds__(64) void k_spec_repair(const FrameDev* __restrict__ frames,
                                                     const SpecChunkDev* __restrict__ chunks,
                                                     const uint8_t* __restrict__ stream,
                                                     const QTab* __restrict__ lut, uint32_t* __restrict__ recbits,
                                                     uint32_t* __restrict__ nrec, uint32_t* __restrict__ wstart,
                                                     uint2* __restrict__ hand, const uint2* __restrict__ fix,
                                                     uint32_t* __restrict__ nfix,
                                                     const uint8_t* __restrict__ fixflag, uint32_t total,
                                                     const uint32_t* __restrict__ state) {
  __shared__ uint32_t s_bits[kSpecTilesMax * 4];
  if (state && state[kSpecStPause]) return;  // paused: no walker ran, the list is the last unpaused launch's  // the re-walked span's start bits (at most kSpecSpan bytes)
  const uint32_t n = min(*nfix, total);  // (the count also takes the walkers repaired beyond the list: policy's figure)
  uint32_t extra = 0;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint2 e = fix[i];
    // (a list entry's walker is never a packet's first, so walker e.x - 1 belongs to the same packet)
    if (fixflag[e.x - 1u]) continue;  // not the first of its run
    uint32_t w = e.x, from = e.y;
    const uint32_t frame = chunks[w].frame;
    const FrameDev f = frames[frame];
    bool listed = true;  // walker w is on the list (marked)
    for (int guard = 0; guard < 4096; guard++) {
      const SpecChunkDev sc = chunks[w];
      const uint32_t limit = (sc.c + 1u) * (uint32_t)kSpecChunk;  // where the walker's span ends
      uint32_t take, tail;
      for (int k = threadIdx.x; k < kSpecTilesMax * 4; k += 64) s_bits[k] = 0u;
      __syncthreads();
      const uint32_t cnt = walk_record(f, stream, lut, from, sc.c * (uint32_t)kSpecChunk, limit, s_bits,
                                       (uint32_t)(kSpecTilesMax * 128), take, tail);
      __syncthreads();
      for (int k = threadIdx.x; k < kSpecTilesMax * 4; k += 64) recbits[spec_bits_dword(w, (uint32_t)k)] = s_bits[k];
      __syncthreads();
      if (threadIdx.x == 0) {
        nrec[w] = cnt;
        wstart[w] = from;
        hand[w] = make_uint2(take, tail);
      }
      if (!listed) extra++;
      from += tail & 0xFFFFu;  // the last (macro)block start below the end of this chunk: where the next walker takes over
      // wave-uniform from here: does the next walker of this packet need the same?
      if (w + 1u >= total || chunks[w + 1u].frame != frame) break;
      const bool next_listed = fixflag[w + 1u] != 0;
      if (next_listed && !listed) break;  // the first of another wave's run
      if (!next_listed && wstart[w + 1u] + (hand[w + 1u].x & 0xFFFFu) == from) break;  // it takes over where it should
      listed = next_listed;
      w++;
    }
  }
  if (threadIdx.x == 0 && extra) atomicAdd(nfix, extra);
}

// After k_spec_verify, one workgroup.  todo_cnt[0] = packets refused in this launch, todo_cnt[1..] = the list.
// The plan's walkers start with the short lead.  A launch in which more than 1/32 of them had to be walked
// again (or that was lost, see below) moves the plan to the long lead: twice the bytes before each chunk to
// fall into step in, a quarter more to parse.  With the long lead, kSpecQuietLaunches launches in a row that
// needed next to no repairs (under 1/32768 of the walkers: what content that is comfortable with the short
// lead shows; a small batch shows it on any content, hence the run length) move it back;
// two lost launches in a row pause the speculation for kSpecPauseLaunches launches, during which both walkers
// and k_spec_verify return at once and this kernel puts every packet on the exact kernels' list.
__global__ __launch_bounds__(256) void k_spec_policy(uint32_t n, uint32_t walkers, uint32_t* __restrict__ todo_cnt,
                                                      const uint32_t* __restrict__ nfix, uint32_t* __restrict__ state,
                                                      const uint32_t serial_min) {
  const uint32_t pause = state[kSpecStPause];
  if (pause) {
    for (uint32_t i = threadIdx.x; i < n; i += 256) todo_cnt[1 + i] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
      todo_cnt[0] = n;
      state[kSpecStPause] = pause - 1u;
      if (pause == 1u) state[kSpecStLost] = 1u;  // one more lost launch pauses again
    }
  } else if (threadIdx.x == 0) {
    // a lost launch: more than half of the packets refused (they pay for the walkers AND for the exact kernels:
    // noise of +-36 at the highest quality ran at 87 K pictures per second that way, against 144 K with the exact
    // kernels alone), or so many walkers repaired (one wave each) that the exact kernels would have been quicker
    const uint32_t nf = *nfix;
    const bool lost = 2u * (uint64_t)todo_cnt[0] > n || 4u * (uint64_t)nf > walkers;
    const uint32_t level = state[kSpecStLong];  // 0 / 1 / 2: the lead of kSpecLead / kSpecLeadLong / kSpecLeadVery bytes
    if (level < 2u && (lost || 32u * (uint64_t)nf > walkers)) {
      // many walkers lock late: the next longer lead — unless that is the longest and the launch is one the serial
      // walker takes (serial_min): walkers with the longest lead parse four times their chunk, 37 ms per 16,384 packets of
      // 1080p at +-32 where the serial walker takes 30 (profiles/r04/noisy_by_index.txt), so such a plan pauses instead
      if (level == 1u && serial_min != 0u && n >= serial_min) {
        state[kSpecStPause] = (uint32_t)kSpecPauseLaunches;
      } else {
        state[kSpecStLong] = level + 1u;
      }
      state[kSpecStQuiet] = 0u;
      state[kSpecStLost] = 0u;
    } else if (level > 0u) {
      if (level == 2u) {  // nothing longer to try: two lost launches in a row pause the speculation
        const uint32_t streak = lost ? state[kSpecStLost] + 1u : 0u;
        state[kSpecStLost] = streak;
        if (streak >= 2u) state[kSpecStPause] = (uint32_t)kSpecPauseLaunches;
      }
      const uint32_t quiet = !lost && 32768u * (uint64_t)nf < walkers ? state[kSpecStQuiet] + 1u : 0u;
      state[kSpecStQuiet] = quiet;
      if (quiet >= (uint32_t)kSpecQuietLaunches) {  // comfortable for a while: try the next shorter lead again
        state[kSpecStLong] = level - 1u;
        state[kSpecStQuiet] = 0u;
      }
    }
  }
  // Who indexes the packets on the list: from serial_min packets on the serial walker (k_index_walk_todo, one wave per
  // packet — enough of them in flight to beat the chunk-parallel kernels, whose work grows with the bytes), else the
  // exact kernels.  The walker's count lives behind the list.
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t cnt = todo_cnt[0];
    const bool serial = serial_min != 0u && cnt >= serial_min;
    todo_cnt[1 + n] = serial ? cnt : 0u;
    if (serial) todo_cnt[0] = 0u;
  }
}

}  // namespace mirtj
