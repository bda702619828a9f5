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
//                  byte-serial state machine over lead + chunk, recording every block start.
//                  64 chunks advance in lockstep per wave: ~20 vector instructions per byte for 64
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
constexpr int kSpecSpan = kSpecLeadLong + kSpecChunk;  // the longest span a walker parses
constexpr int kSpecTile = 128;    // bytes per lane fetched at a time (one cache line)
constexpr int kSpecCap = 2048;    // block starts a walker can record (16-bit, relative to its first byte):
                                  // enough for blocks of 1.75 bytes on average over its span
// Where a walker's records live.  MIRTJ_SPEC_REC_INTERLEAVE == 0: a row of kSpecCap records per walker (4 KB, of which
// a tenth is used: the 64 lanes of a wave store 16 bytes each into 64 rows 4 KB apart).  == 1: the 64 walkers of a
// wave share one region of 64 * kSpecCap records, laid out [group of 8 records][lane][8]: what a wave stores at a time
// is contiguous, and the region fills from its start.
#ifndef MIRTJ_SPEC_REC_INTERLEAVE
#define MIRTJ_SPEC_REC_INTERLEAVE 1
#endif
constexpr bool kSpecRecInterleave = MIRTJ_SPEC_REC_INTERLEAVE != 0;
__host__ __device__ constexpr size_t spec_rec_base(uint32_t walker) {  // in records, from the start of the buffer
  return kSpecRecInterleave ? (size_t)(walker >> 6) * (64u * (size_t)kSpecCap) : (size_t)walker * (size_t)kSpecCap;
}
__host__ __device__ constexpr size_t spec_rec_index(uint32_t walker, uint32_t i) {
  return spec_rec_base(walker) +
         (kSpecRecInterleave ? ((size_t)(i >> 3) * 64u + (walker & 63u)) * 8u + (i & 7u) : (size_t)i);
}
constexpr uint64_t kSpecMinWalkers = 40000;  // below this (~100 MB of packets) the exact kernels index a batch faster:
                                            // a walker is one lane and runs ~0.35 ms whatever the batch (host policy)
constexpr int kSpecRingRow = 64 + 16;       // LDS bytes per lane of the record ring: 32 records + bank padding
constexpr int kSpecRow = kSpecTile + 16;  // LDS bytes per lane: tile + the dword after it + bank padding
static_assert(kSpecSpan % kSpecTile == 0 && kSpecSpan < 65536, "walker span: whole tiles, 16-bit positions");
static_assert(kSpecLead >= 6 * 64 + 64 && kSpecLead % kSpecTile == 0 && kSpecChunk % kSpecTile == 0, "the lead: at least one whole macroblock, whole tiles");
static_assert(kSpecLeadLong >= kSpecLead && kSpecLeadLong % kSpecTile == 0 && kSpecLeadLong <= kSpecChunk,
              "a walker starts inside the chunk before its own (chunk 1's walker at byte kSpecChunk - lead >= 0)");

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
constexpr int kSpecVerBatch = MIRTJ_SPEC_VER_BATCH;  // chunks whose records a wave of k_spec_verify copies side by side
constexpr int kSpecVerThreads = MIRTJ_SPEC_VER_THREADS;  // k_spec_verify: chunks of a packet handled side by side  // grid rows of the exact kernels when they only serve refused packets

// One byte of the walker's state machine, spelled out: 17 vector instructions (the compiler's version of
// the same C++ had 21-22: it keeps lane masks as 0/1 integers and splits the counters).  State: u = units
// of the current block so far, q = 5 - (block number within the macroblock), rb = DC + raw bytes of the
// current block's type, cnt = records so far, em = lane mask "the previous byte ended a block".  The next
// block's start `val` goes to ring slot cnt % 32 on every byte; cnt only moves on when this byte ends its
// block.  gfx950 wants two instructions between a vector compare and the use of its mask; the order below
// provides them without s_nop.
#define MIRTJ_SPEC_STEP(SEL)                                                                                      \
  asm("v_sub_u32_sdwa %[t], sext(%[w]), %[k63] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SEL               \
      " src1_sel:DWORD\n\t"                                                                                       \
      "v_cmp_eq_i32_e32 vcc, %[kn64], %[t]\n\t"         /* the byte is 0xFF */                                     \
      "v_cmp_lt_i32_e64 %[ma], %[u], %[rb]\n\t"         /* a DC or raw byte */                                     \
      "v_max_i32_e32 %[wt], 1, %[t]\n\t"                /* token weight (lib/RTjpeg.c:171-182) */                  \
      "s_and_b64 vcc, vcc, %[em]\n\t"                   /* 0xFF as a block's first byte: one-byte block */        \
      "v_bfe_u32 %[idx], %[cnt], 0, 5\n\t"                                                                         \
      "v_cndmask_b32_e64 %[wr], 1, 64, vcc\n\t"                                                                    \
      "v_cndmask_b32_e64 %[wt], %[wt], %[wr], %[ma]\n\t"                                                           \
      "v_add_u32_e32 %[u], %[u], %[wt]\n\t"                                                                        \
      "v_cmp_lt_i32_e64 %[em], 63, %[u]\n\t"            /* 64 units: the block ends with this byte */              \
      "v_lshl_add_u32 %[ra], %[idx], 1, %[ring]\n\t"                                                               \
      "ds_write_b16 %[ra], %[val]\n\t"                                                                             \
      "v_subb_co_u32_e64 %[q], vcc, %[q], 0, %[em]\n\t"                                                            \
      "v_cmp_lt_i32_e64 %[mx], %[q], 0\n\t"             /* past the macroblock's last block */                     \
      "v_cmp_lt_u32_e64 %[my], %[q], 2\n\t"             /* a chroma block is next */                               \
      "v_addc_co_u32_e64 %[cnt], vcc, 0, %[cnt], %[em]\n\t"                                                        \
      "v_cndmask_b32_e64 %[u], %[u], 0, %[em]\n\t"                                                                 \
      "v_cndmask_b32_e64 %[q], %[q], 5, %[mx]\n\t"                                                                 \
      "v_cndmask_b32_e64 %[rb], %[lb], %[cb], %[my]"                                                               \
      : [u] "+v"(u), [cnt] "+v"(cnt), [q] "+v"(q), [rb] "+v"(rb), [em] "+s"(em), [t] "=&v"(t_), [wt] "=&v"(wt_),  \
        [wr] "=&v"(wr_), [idx] "=&v"(idx_), [ra] "=&v"(ra_), [ma] "=&s"(ma_), [mx] "=&s"(mx_), [my] "=&s"(my_)     \
      : [w] "v"(w_), [k63] "v"(k63), [kn64] "v"(kn64), [ring] "v"(ring_a), [val] "v"(val_), [lb] "v"(lb), [cb] "v"(cb) \
      : "vcc", "memory")

// The same when every packet of the launch has lb8 == cb8 (two thirds of the qualities): all blocks parse
// alike, no phase to track, 12 vector instructions.
#define MIRTJ_SPEC_STEP1(SEL)                                                                                     \
  asm("v_sub_u32_sdwa %[t], sext(%[w]), %[k63] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SEL               \
      " src1_sel:DWORD\n\t"                                                                                       \
      "v_cmp_eq_i32_e32 vcc, %[kn64], %[t]\n\t"                                                                    \
      "v_cmp_lt_i32_e64 %[ma], %[u], %[rb]\n\t"                                                                    \
      "v_max_i32_e32 %[wt], 1, %[t]\n\t"                                                                           \
      "s_and_b64 vcc, vcc, %[em]\n\t"                                                                              \
      "v_bfe_u32 %[idx], %[cnt], 0, 5\n\t"                                                                         \
      "v_cndmask_b32_e64 %[wr], 1, 64, vcc\n\t"                                                                    \
      "v_cndmask_b32_e64 %[wt], %[wt], %[wr], %[ma]\n\t"                                                           \
      "v_add_u32_e32 %[u], %[u], %[wt]\n\t"                                                                        \
      "v_cmp_lt_i32_e64 %[em], 63, %[u]\n\t"                                                                       \
      "v_lshl_add_u32 %[ra], %[idx], 1, %[ring]\n\t"                                                               \
      "ds_write_b16 %[ra], %[val]\n\t"                                                                             \
      "v_addc_co_u32_e64 %[cnt], vcc, 0, %[cnt], %[em]\n\t"                                                        \
      "v_cndmask_b32_e64 %[u], %[u], 0, %[em]"                                                                     \
      : [u] "+v"(u), [cnt] "+v"(cnt), [em] "+s"(em), [t] "=&v"(t_), [wt] "=&v"(wt_), [wr] "=&v"(wr_),              \
        [idx] "=&v"(idx_), [ra] "=&v"(ra_), [ma] "=&s"(ma_)                                                        \
      : [w] "v"(w_), [k63] "v"(k63), [kn64] "v"(kn64), [ring] "v"(ring_a), [val] "v"(val_), [rb] "v"(rb)           \
      : "vcc", "memory")

template <bool PHASE, int LEAD>
__global__ __launch_bounds__(64) void k_spec_walk(const FrameDev* __restrict__ frames,
                                                   const SpecChunkDev* __restrict__ chunks, uint32_t total,
                                                   const uint8_t* __restrict__ stream,
                                                   const QTab* __restrict__ lut, uint16_t* __restrict__ records,
                                                   uint32_t* __restrict__ nrec, uint32_t* __restrict__ wstart,
                                                   uint2* __restrict__ hand, const uint32_t* __restrict__ state) {
  // both forms are launched; the plan's policy state says which one works (none while paused, k_spec_policy)
  if (state && (state[kSpecStPause] || (state[kSpecStLong] != 0u) != (LEAD != kSpecLead))) return;
  constexpr int kSpan = LEAD + kSpecChunk;
  static_assert(kSpan % kSpecTile == 0 && kSpan < 65536, "walker span: whole tiles, 16-bit positions");
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[64 * kSpecRingRow];
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x * 64u + (uint32_t)lane;
  const bool act = g < total;
  const SpecChunkDev sc = chunks[act ? g : total - 1u];  // idle lanes shadow the last chunk and store nothing
  const FrameDev f = frames[sc.frame];
  const int lb = lut[f.qidx].lb8 + 1, cb = lut[f.qidx].cb8 + 1;  // DC + raw bytes of a luma / chroma block
  const uint32_t start = sc.c ? sc.c * (uint32_t)kSpecChunk - (uint32_t)LEAD : 0u;  // first byte parsed
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
  //      1 + bt8 DC/raw bytes, ph = its number within the macroblock ----
  int u = 0, rb = lb;
  uint32_t q = 5;        // 5 - block number within the macroblock
  uint64_t em = ~0ull;   // the walker's first byte is taken to be a macroblock's first
  // record k = start of the walker's block k (phase k mod 6), 16-bit.  Records are staged in a 32-entry
  // ring per lane in LDS and leave for HBM eight at a time (one 16-byte store): a 2-byte global store
  // per block end was 40 % of the kernel.  Idle lanes of the last wave own the spare row after the last
  // walker's.
  const uint32_t gw = act ? g : total;  // idle lanes of the last wave own the spare walker after the last one
  uint8_t* const rec8 = (uint8_t*)records;
  uint8_t* const ring = s_ring + lane * kSpecRingRow;
  const uint32_t ring_a = lds_address(ring);
  const int kn64 = -64;
  uint32_t cnt = 1, flushed = 0;
  *(uint16_t*)ring = 0;
  const int k63 = 63;
  // pending = cnt - flushed as a signed number: the last, partial group leaves it negative
  auto flush = [&](int least) {
    while (__any((int)(cnt - flushed) >= least)) {
      if ((int)(cnt - flushed) >= least) {
        const uint4 v = *(const uint4*)(ring + ((flushed & 31u) << 1));  // flushed is a multiple of 8
        if (flushed <= (uint32_t)kSpecCap - 8u) *(uint4*)(rec8 + 2u * spec_rec_index(gw, flushed)) = v;
        flushed += 8u;
      }
    }
  };

  // (index << 16 | position) of the last unit-aligned record below `limit` (walker-relative); the unit is the
  // macroblock, or the block when lb8 == cb8 (k_spec_verify).  Called between bytes: at most the newest
  // record can sit AT the limit, and the records wanted are among the last seven, i.e. still in the ring.
  const bool by_block = lb == cb;
  auto last_aligned_below = [&](uint32_t limit) -> uint32_t {
    const uint32_t newest = *(const uint16_t*)(ring + (((cnt - 1u) & 31u) << 1));
    const uint32_t d = newest >= limit ? 1u : 0u, n = cnt - d;
    // block cnt-1 has phase 5-q, so record n-1 has phase (5-q-d) mod 6: no division (one by a lane-dependent
    // value, or even by 6, does not get through the compiler next to the SGPR-mask assembly)
    int r = 5 - (int)q - (int)d;
    r = r < 0 ? r + 6 : r;
    const uint32_t idx = n - 1u - (by_block ? 0u : (uint32_t)r);
    return (idx << 16) | *(const uint16_t*)(ring + ((idx & 31u) << 1));
  };
  uint32_t take = 0;  // where this walker's chunk takes over from its predecessor (chunk 0: byte 0, record 0)
  uint32_t tail0 = 0;

  request(0);
  for (int t = 0; t < kSpan / kSpecTile; t++) {
    // the tile is parsed out of registers (fully unrolled: 128 byte steps); staging it in LDS for a
    // smaller loop body capped the kernel at 11 waves per CU
    if (t == LEAD / kSpecTile) {  // the chunk begins with this tile
      const uint32_t v = last_aligned_below((uint32_t)LEAD);
      take = sc.c ? v : 0u;
    }
    if (t == kSpecChunk / kSpecTile) tail0 = last_aligned_below((uint32_t)kSpecChunk);  // chunk 0 (no lead) ends here
    uint32_t cur[33];
#pragma unroll
    for (int k = 0; k < 33; k++) cur[k] = buf[k];
    // the records of the tile before leave now, BEFORE the next tile's loads are queued: loads and stores
    // complete in order, so a store queued after those loads would have to be waited for with them (-3 %)
    flush(8);
    if (t + 1 < kSpan / kSpecTile) request(t + 1);  // in flight while this tile is parsed
#pragma unroll
    for (int i = 0; i < kSpecTile / 16; i++) {
      const uint32_t wd[4] = {__builtin_amdgcn_alignbyte(cur[4 * i + 1], cur[4 * i], sh),
                              __builtin_amdgcn_alignbyte(cur[4 * i + 2], cur[4 * i + 1], sh),
                              __builtin_amdgcn_alignbyte(cur[4 * i + 3], cur[4 * i + 2], sh),
                              __builtin_amdgcn_alignbyte(cur[4 * i + 4], cur[4 * i + 3], sh)};
      const uint32_t pos = (uint32_t)(t * kSpecTile + 16 * i);  // walker-relative position of wd's first byte
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const uint32_t w_ = wd[b >> 2], val_ = pos + (uint32_t)b + 1u;  // the next block would start at pos + b + 1
        int t_, wt_, wr_;
        uint32_t idx_, ra_;
        uint64_t ma_, mx_, my_;
        if (PHASE) {
          switch (b & 3) {
            case 0: MIRTJ_SPEC_STEP("BYTE_0"); break;
            case 1: MIRTJ_SPEC_STEP("BYTE_1"); break;
            case 2: MIRTJ_SPEC_STEP("BYTE_2"); break;
            default: MIRTJ_SPEC_STEP("BYTE_3"); break;
          }
        } else {
          switch (b & 3) {
            case 0: MIRTJ_SPEC_STEP1("BYTE_0"); break;
            case 1: MIRTJ_SPEC_STEP1("BYTE_1"); break;
            case 2: MIRTJ_SPEC_STEP1("BYTE_2"); break;
            default: MIRTJ_SPEC_STEP1("BYTE_3"); break;
          }
        }
      }
      flush(16);  // inside a tile only when the ring is half full: at most 15 + 16 records are staged at this point
    }
  }
  flush(1);  // the last, partial group (the slots past cnt are never read)
  if (act) {
    nrec[g] = cnt;  // > kSpecCap: the span held more blocks than a walker records
    wstart[g] = start;
    const uint32_t tail = last_aligned_below((uint32_t)kSpan);
    hand[g] = make_uint2(take, sc.c ? tail : tail0);  // .y: what the next chunk must take over from
  }
}

// One workgroup per packet: the chain check, the block numbering, and — the numbering is all it needs —
// the block-offset index itself.  A packet that fails goes on the exact kernels' to-do list (they then
// rewrite its index).
__global__ __launch_bounds__(kSpecVerThreads) void k_spec_verify(const FrameDev* __restrict__ frames,
                                                      const uint32_t* __restrict__ spec_base,
                                                      const QTab* __restrict__ lut,
                                                      const uint16_t* __restrict__ records,
                                                      const uint32_t* __restrict__ nrec, uint32_t* __restrict__ blkoff,
                                                      uint32_t* __restrict__ ok, uint32_t* __restrict__ todo,
                                                      uint32_t* __restrict__ ntodo, const uint32_t* __restrict__ state,
                                                      const uint32_t* __restrict__ wstart,
                                                      const uint2* __restrict__ hand, uint2* __restrict__ fix,
                                                      uint32_t* __restrict__ nfix, int pass) {
  if (state && state[kSpecStPause]) {  // paused (k_spec_policy): nothing was walked, nothing is proven
    if (threadIdx.x == 0) ok[blockIdx.x] = 0;
    return;
  }
  if (pass == 2 && ok[blockIdx.x] != 2u) return;  // the second pass only looks at packets with repaired chunks
  __shared__ uint32_t s_wave[kSpecVerThreads / 64], s_carry[3];  // carry: [1] blocks so far, [2] bad
  __shared__ uint32_t s_i0[kSpecVerThreads], s_base[kSpecVerThreads], s_cnt[kSpecVerThreads];
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
      const uint32_t full = nrec[sc0 + c], n = min(full, (uint32_t)kSpecCap);
      const uint2 mine = hand[sc0 + c];
      bad = full > (uint32_t)kSpecCap;  // the walker ran out of slots
      i0 = c ? mine.x >> 16 : 0u;
      uint32_t i1 = n;  // the packet's last walker: everything it saw (bytes past the packet read as 0)
      if (c + 1 < nsc) {
        if (nrec[sc0 + c + 1] > (uint32_t)kSpecCap) bad = 1;
        const uint32_t h = wstart[sc0 + c + 1] + (hand[sc0 + c + 1].x & 0xFFFFu);  // where chunk c + 1 took over
        const uint32_t t = wstart[sc0 + c] + (mine.y & 0xFFFFu);                    // where it had to
        i1 = mine.y >> 16;
        if (h != t) {
          if (pass == 1 && !bad) {
            // the next walker had not fallen into step yet: it is re-walked (k_spec_repair) from the last
            // (macro)block start this one saw, and the packet gets a second pass
            soft = 1;
            fix[atomicAdd(nfix, 1u)] = make_uint2(sc0 + c + 1u, t);
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
    s_i0[tid] = i0;
    s_base[tid] = before;
    s_cnt[tid] = cnt;
    __syncthreads();
    if (lane == 0 && anybad) atomicOr(&s_carry[2], anybad);
    if (tid == kSpecVerThreads - 1) s_carry[1] = before + cnt;
    // recorded starts -> block offsets, one wave per chunk of the tile (harmless if the packet fails later:
    // counts of refused chunks are 0, everything is clipped to the packet's own index, and the exact
    // kernels rewrite it)
    const uint32_t tile_n = min((uint32_t)kSpecVerThreads, nsc - c0);
    // A chunk holds ~200 blocks, i.e. one round trip of four 2-byte gathers per lane; a wave works on
    // kSpecVerBatch chunks at a time so that it waits for memory once per batch, not once per chunk.
    constexpr uint32_t kWaves = kSpecVerThreads / 64;
    for (uint32_t j0 = (uint32_t)wv; j0 < tile_n; j0 += kSpecVerBatch * kWaves) {
      uint32_t base[kSpecVerBatch], m[kSpecVerBatch], start[kSpecVerBatch], first[kSpecVerBatch], mmax = 0;
      uint32_t R[kSpecVerBatch];  // the walker whose records are copied
#pragma unroll
      for (int u = 0; u < kSpecVerBatch; u++) {
        const uint32_t j = min(j0 + (uint32_t)u * kWaves, tile_n - 1u);
        const bool have = j0 + (uint32_t)u * kWaves < tile_n;
        base[u] = s_base[j];
        m[u] = have && base[u] <= last ? min(s_cnt[j], last + 1u - base[u]) : 0u;  // clipped to the packet's own index
        start[u] = wstart[sc0 + c0 + j];
        R[u] = sc0 + c0 + j;
        first[u] = s_i0[j];
        mmax = max(mmax, m[u]);
      }
      for (uint32_t k0 = 0; k0 < mmax; k0 += 256) {
        uint32_t v[kSpecVerBatch][4];
#pragma unroll
        for (int u = 0; u < kSpecVerBatch; u++) {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const uint32_t k = k0 + 64u * (uint32_t)t + (uint32_t)lane;
            v[u][t] = k < m[u] ? records[spec_rec_index(R[u], first[u] + k)] : 0u;
          }
        }
#pragma unroll
        for (int u = 0; u < kSpecVerBatch; u++) {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const uint32_t k = k0 + 64u * (uint32_t)t + (uint32_t)lane;
            if (k < m[u]) out[base[u] + k] = start[u] + v[u][t];
          }
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

// A walker that had not fallen into step by the end of its lead is walked again, this time from a byte
// that is known to start a (macro)block if its predecessor is right (the second k_spec_verify pass checks
// exactly that): one WAVE per such chunk, the block-by-block walker of rtj_decode_kernels.h (a lane-serial
// walker would take as long for one chunk as k_spec_walk takes for all).  grid: any; loops over the list.
__global__ __launch_bounds__(64) void k_spec_repair(const FrameDev* __restrict__ frames,
                                                     const SpecChunkDev* __restrict__ chunks,
                                                     const uint8_t* __restrict__ stream,
                                                     const QTab* __restrict__ lut, uint16_t* __restrict__ records,
                                                     uint32_t* __restrict__ nrec, uint32_t* __restrict__ wstart,
                                                     uint2* __restrict__ hand, const uint2* __restrict__ fix,
                                                     const uint32_t* __restrict__ nfix) {
  const uint32_t n = *nfix;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint2 e = fix[i];
    const SpecChunkDev sc = chunks[e.x];
    const FrameDev f = frames[sc.frame];
    const uint32_t limit = (sc.c + 1u) * (uint32_t)kSpecChunk;  // where the walker's span ends
    uint32_t take, tail;
    const uint32_t cnt = walk_record(f, stream, lut, e.y, sc.c * (uint32_t)kSpecChunk, limit,
                                     records + spec_rec_base(e.x), (uint32_t)kSpecCap, take, tail, e.x, kSpecRecInterleave);
    if (threadIdx.x == 0) {
      nrec[e.x] = cnt;
      wstart[e.x] = e.y;
      hand[e.x] = make_uint2(take, tail);
    }
  }
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
                                                      const uint32_t* __restrict__ nfix, uint32_t* __restrict__ state) {
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
    // a lost launch: every packet refused, or so many walkers repaired (one wave each) that the exact
    // kernels would have been quicker
    const uint32_t nf = *nfix;
    const bool lost = todo_cnt[0] == n || 4u * (uint64_t)nf > walkers;
    if (!state[kSpecStLong]) {
      if (lost || 32u * (uint64_t)nf > walkers) {
        state[kSpecStLong] = 1u;
        state[kSpecStQuiet] = 0u;
      }
    } else {
      const uint32_t streak = lost ? state[kSpecStLost] + 1u : 0u;
      state[kSpecStLost] = streak;
      if (streak >= 2u) state[kSpecStPause] = (uint32_t)kSpecPauseLaunches;
      const uint32_t quiet = !lost && 32768u * (uint64_t)nf < walkers ? state[kSpecStQuiet] + 1u : 0u;
      state[kSpecStQuiet] = quiet;
      if (quiet >= (uint32_t)kSpecQuietLaunches) state[kSpecStLong] = 0u;
    }
  }
}

}  // namespace mirtj
