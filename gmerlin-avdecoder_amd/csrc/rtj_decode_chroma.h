// rtj_decode_chroma.h — the chroma waves of a batch launch of k_decode: three macroblock groups' chroma parts per round,
// their busy blocks POOLED into one transform round.
//
// Why.  At the qualities RTjpeg streams are made with, the chroma tables have no raw bytes (cb8 == 0) and three of
// four chroma blocks of ordinary content are the two bytes "DC, run of 63": 64 equal pixels,
// clamp16_235(int16((int16(DC * q0) + 4) >> 3)) — the reference's own shortcut (lib/RTjpeg.c:2223-2238); the others are
// 3 to 6 bytes, one or two coefficients next to the DC.  A chroma round of 64 blocks nevertheless costs a transform
// round (~520 vector instructions) in decode_wave, because a vector instruction costs the same for 17 live lanes as for
// 64, and the chroma rounds were 24 % of k_decode's instructions.  Here a wave takes the chroma parts
// of THREE consecutive groups at once:
//   1. classify the 3 x 64 blocks from their lengths (the block-offset index gives them: entry k + 1 minus entry k)
//      and first byte: unchanged (0xFF) / DC only (two bytes) / busy (up to eight bytes);
//   2. the busy blocks (about 3 x 17) are compacted over the lanes: block of rank k hands its first eight stream
//      bytes to lane k through the LDS; lanes 0 .. n-1 parse and transform them with the short forms of the
//      transform (transform_lo: the three- and four-input butterflies) and put the pixel rows back into the LDS;
//   3. the home lanes of each group store their rows — DC-only ones from a register, busy ones from the LDS — as the
//      same whole 256-byte row segments (32 blocks of Cb, 32 of Cr) a plain round stores.  (Round 2 deferred the
//      busy blocks to a second kernel and lost on partial-line writes; nothing is deferred here.)
// A group with a block the short forms do not cover (longer than eight bytes, a coefficient outside the low 4x4, bytes
// past the packet's end) is left, whole, to k_decode_list (DecList, rtj_decode_kernels.h), which runs decode_wave's
// general path on it: noisy content pays ~40 instructions per group for having asked.
//
// Loads.  As in decode_wave nothing the wave waits for may be younger than a row store (one counter, vmcnt, for
// loads and stores, in issue order): the stream bytes of round j + 1 and the block offsets of round j + 2 are requested
// at the top of round j — hand-issued, the compiler does not know they are pending — and waited for at the top of
// round j + 1 with a counted wait that leaves exactly round j's row stores (0, 8, 16 or 24) in flight.
#pragma once
#include "rtj_decode_kernels.h"

namespace mirtj {

constexpr int kPoolGroups = 3;  // groups per pooled round (17 busy blocks per group on the bench content: 51 of 64 lanes)
constexpr int kPoolItersMax = 11;  // pooled rounds per wave at most (33 groups)


#ifdef MIRTJ_EXPERIMENTS  // timing builds: shader-clock ticks per section of a pooling wave, summed over all waves
__device__ unsigned long long g_pool_stamps[16];
#define MIRTJ_PSTAMP(i)                                         \
  do {                                                          \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    pst[i] += t_ - pst_last;                                    \
    pst_last = t_;                                              \
  } while (0)
#else
#define MIRTJ_PSTAMP(i) \
  do {                  \
  } while (0)
#endif

typedef uint32_t mirtj_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t mirtj_u32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ void chroma_pool_wave(uint32_t* __restrict__ s_lds, const FrameDev* __restrict__ frames,
                                                 const uint32_t fidx, const uint32_t slot, const uint32_t slots,
                                                 const uint8_t* __restrict__ stream, const QTab* __restrict__ lut,
                                                 const uint32_t* __restrict__ blkoff, uint8_t* __restrict__ outbuf,
                                                 const DecList list) {
  uint32_t* s_tab = s_lds + kCoefWords;
  const FrameDev f = frames[fidx];
  const uint32_t ngroups = (f.nmb + (uint32_t)kMbPerGroup - 1u) / (uint32_t)kMbPerGroup;
  const uint32_t nsg = (ngroups + (uint32_t)kPoolGroups - 1u) / (uint32_t)kPoolGroups;  // rounds of the frame
  if (slot >= nsg) return;
  const int lane = threadIdx.x & 63;
  const QTab& qt = lut[f.qidx];
  if (qt.cb8 != 0) {  // tables with raw chroma bytes (none of RTjpeg_set_quality's): every group the general way
    if (lane == 0)
      for (uint32_t sg = slot; sg < nsg; sg += slots)
        for (uint32_t q = 0; q < (uint32_t)kPoolGroups; q++)
          if (sg * (uint32_t)kPoolGroups + q < ngroups) declist_push(list, fidx, sg * (uint32_t)kPoolGroups + q, 2u);
    return;
  }
  {  // the chroma slot table (decode_wave's second one): dequantiser << 16 | scratch byte offset, then "finished" entries
    const int nat = c_zz[lane];
    s_tab[kSlotTabN + lane] = ((uint32_t)qt.ciqt[nat] << 16) | (uint32_t)coef_byte(nat);
    if (lane < kSlotTabN - 64) s_tab[kSlotTabN + 64 + lane] = 128u;
  }
  wave_lds_sync();  // orders the table write before the lanes' reads (the table is the wave's own)
  const uint32_t lds0 = lds_address(s_lds);
  const uint32_t my_a = lds0 + (uint32_t)lane * (uint32_t)(kCoefStride * 2);
  const uint4* my = (const uint4*)((const uint8_t*)s_lds + (size_t)lane * (kCoefStride * 2));
  const uint32_t tab_c = lds_address(s_tab) + 4u * (uint32_t)kSlotTabN;
  const int cend = (int)tab_c + 4 * 64;
  const int k63 = 63;
  const uint32_t q0 = (uint32_t)qt.ciqt[0];  // the DC's dequantiser
  const uint32_t* off = uniform_ptr(blkoff + f.blk_base);
  const uint8_t* data = uniform_ptr(stream + f.data_off);
  const uint32_t dsh = (uint32_t)((uintptr_t)data & 3u);
  const uint8_t* const g4b = data - dsh;  // dword-aligned base of the packet's bytes (wave-uniform)
  const uint32_t dmb = (uint32_t)(lane & 31), kblk = 4u + (uint32_t)(lane >> 5);
  const size_t ysz = (size_t)f.w * f.h;
  const uint32_t stride = f.w >> 1;
  uint8_t* const plane0 = outbuf + f.out_off + ysz;  // Cb, then Cr (wave-uniform)
  const IdctK K{362, 473, -669, 277, 128, 235};
  const IdctPK KP = idct_pk_constants();

  // byte offset into `off` of this lane's block in group q of round sg (0: no such block)
  auto off_byte = [&](uint32_t sg, uint32_t q) -> uint32_t {
    const uint32_t mb = (sg * (uint32_t)kPoolGroups + q) * (uint32_t)kMbPerGroup + dmb;
    return mb < f.nmb ? 4u * (6u * mb + kblk) : 0u;
  };
  // what a lane keeps of its block between the arrival of its offsets and its round: length (0: no block, or its
  // group goes to the list) | alignment of its bytes << 8; `gslow`: bit q = group q of the round goes to the list
  struct Meta {
    uint32_t m[kPoolGroups];
    uint32_t gslow;
  };
  // from the offsets (pos, next pos) of a round: the lanes' Meta and where their bytes are loaded from
  auto prepare = [&](uint32_t sg, const mirtj_u32x2 (&o)[kPoolGroups], Meta& mt, uint32_t (&ld)[kPoolGroups]) {
    mt.gslow = 0u;
#pragma unroll
    for (int q = 0; q < kPoolGroups; q++) {
      const uint32_t mb = (sg * (uint32_t)kPoolGroups + (uint32_t)q) * (uint32_t)kMbPerGroup + dmb;
      const bool valid = mb < f.nmb;
      const uint32_t pos = o[q].x, len = o[q].y - o[q].x;
      // the short forms take blocks of up to eight bytes that lie inside the packet (bytes past its end read as 0:
      // the general path's business)
      const bool slow = valid && (len > 8u || len == 0u || pos + len > f.data_len);
      const bool gs = __any(slow);
      mt.gslow |= gs ? 1u << q : 0u;
      const bool use = valid && !gs;
      const uint32_t a = (use ? pos : 0u) + dsh;
      ld[q] = a & ~3u;
      mt.m[q] = use ? len | ((a & 3u) << 8) : 0u;
    }
  };

#ifdef MIRTJ_EXPERIMENTS
  unsigned long long pst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pst_last = __builtin_amdgcn_s_memtime();
#endif
  mirtj_u32x3 B[kPoolGroups];  // stream bytes of the round in hand: 12 bytes from the dword that holds the block's first
  mirtj_u32x2 O[kPoolGroups];  // (block start, next block's start) of the round after it
  Meta cur_m;
  {
    // the first round's offsets and bytes the plain way, the second round's offsets with them
    mirtj_u32x2 o0[kPoolGroups];
#pragma unroll
    for (int q = 0; q < kPoolGroups; q++) {
      const uint32_t* p = (const uint32_t*)((const uint8_t*)off + off_byte(slot, (uint32_t)q));
      o0[q].x = p[0];
      o0[q].y = p[1];
    }
    uint32_t ld[kPoolGroups];
    prepare(slot, o0, cur_m, ld);
#pragma unroll
    for (int q = 0; q < kPoolGroups; q++) {
      const uint32_t* p = (const uint32_t*)(g4b + ld[q]);
      B[q].x = p[0];
      B[q].y = p[1];
      B[q].z = p[2];
      const uint32_t* p2 = (const uint32_t*)((const uint8_t*)off + (slot + slots < nsg ? off_byte(slot + slots, (uint32_t)q) : 0u));
      O[q].x = p2[0];
      O[q].y = p2[1];
    }
    // waited for here, by the compiler, outside the loop (the loop's waits are hand-placed)
    asm volatile("" ::"v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(O[0]), "v"(O[1]), "v"(O[2]));
  }

  MIRTJ_PSTAMP(0);  // prologue
  uint32_t exit_sg = 0u, exit_bailed = 0u;
  uint32_t nstores = 0u;  // row-store instructions of the round before (wave-uniform): what the counted wait leaves in flight
  for (uint32_t sg = slot;; sg += slots) {
    const bool have_n = sg + slots < nsg;  // wave-uniform
    // ---- the bytes of this round, then the registers go to the next round's loads ----
    uint32_t first4[kPoolGroups], second4[kPoolGroups];
#pragma unroll
    for (int q = 0; q < kPoolGroups; q++) {
      const uint32_t sh = cur_m.m[q] >> 8;
      first4[q] = __builtin_amdgcn_alignbyte(B[q].y, B[q].x, sh);
      second4[q] = __builtin_amdgcn_alignbyte(B[q].z, B[q].y, sh);
    }
    if (cur_m.gslow) {  // groups the short forms do not cover (appended before this round's loads are issued: nothing but
                        // row stores may sit between the hand-issued loads and their wait)
#pragma unroll
      for (int q = 0; q < kPoolGroups; q++)
        if (cur_m.gslow >> q & 1u) declist_push(list, fidx, sg * (uint32_t)kPoolGroups + (uint32_t)q, 2u);
    }
    Meta nxt_m;
    nxt_m.gslow = 0u;
#pragma unroll
    for (int q = 0; q < kPoolGroups; q++) nxt_m.m[q] = 0u;
    // the registers the hand-issued loads fill belong to this iteration alone (copied into B / O behind the wait): a
    // loop-carried register would invite the compiler to copy it in front of the wait
    mirtj_u32x3 nB[kPoolGroups];
    mirtj_u32x2 nO[kPoolGroups];
    if (have_n) {
      uint32_t ld[kPoolGroups], ob[kPoolGroups];
      prepare(sg + slots, O, nxt_m, ld);
      const bool have_nn = sg + 2u * slots < nsg;
#pragma unroll
      for (int q = 0; q < kPoolGroups; q++) ob[q] = have_nn ? off_byte(sg + 2u * slots, (uint32_t)q) : 0u;
      // (s_nop 4: a vector-memory instruction must not read a scalar register within five wait states of a vector
      // instruction writing it, and nobody pads that hazard inside an asm block — see decode_wave)
      asm volatile(
          "s_nop 4\n\t"
          "global_load_dwordx3 %0, %6, %12\n\t"
          "global_load_dwordx3 %1, %7, %12\n\t"
          "global_load_dwordx3 %2, %8, %12\n\t"
          "global_load_dwordx2 %3, %9, %13\n\t"
          "global_load_dwordx2 %4, %10, %13\n\t"
          "global_load_dwordx2 %5, %11, %13 ; mirtj chroma pool loads"
          : "=&v"(nB[0]), "=&v"(nB[1]), "=&v"(nB[2]), "=&v"(nO[0]), "=&v"(nO[1]), "=&v"(nO[2])
          : "v"(ld[0]), "v"(ld[1]), "v"(ld[2]), "v"(ob[0]), "v"(ob[1]), "v"(ob[2]), "s"(g4b), "s"(off)
          : "memory");
    }

    MIRTJ_PSTAMP(1);  // bytes of the round, next round's addresses, loads issued
    // ---- classify: 0 nothing to store (no block, unchanged block, group left to the list), 1 DC only, 2 busy ----
    uint32_t info[kPoolGroups];  // pixel | class << 8 | rank among the round's busy blocks << 10
    uint32_t cnt[kPoolGroups];   // busy blocks of the group (wave-uniform)
    uint32_t gstore = 0u;        // bit q: group q has rows to store (wave-uniform)
#pragma unroll
    for (int q = 0; q < kPoolGroups; q++) {
      const uint32_t len = cur_m.m[q] & 0xFFu, b0 = first4[q] & 0xFFu;
      const uint32_t cls = len == 0u || b0 == 0xFFu ? 0u : len == 2u ? 1u : 2u;  // (two bytes: DC and a run that covers the rest)
      const unsigned long long gm = __ballot(cls == 2u);
      cnt[q] = (uint32_t)__popcll(gm);
      const uint32_t rank = (uint32_t)__popcll(gm & ((1ull << lane) - 1ull));
      // the pixel of a DC-only block: what the transform makes of a lone DC (x0 in both passes)
      const int dc = (int)(int16_t)(b0 * q0);
      info[q] = px(dc + 4) | (cls << 8) | (rank << 10);
      gstore |= __ballot(cls != 0u) != 0ull ? 1u << q : 0u;
    }
    MIRTJ_PSTAMP(2);  // classification
    // ---- rounds of pooled groups: as many consecutive groups as have 64 busy blocks or fewer between them ----
    uint32_t stored = 0u, bailed = 0u;  // groups whose rows were stored / that turned out to need the general path (bit q)
    for (uint32_t qa = 0u; qa < (uint32_t)kPoolGroups;) {
      uint32_t qb = qa, tot = 0u, mask = 0u;
#pragma unroll
      for (int q = 0; q < kPoolGroups; q++) {
        if ((uint32_t)q == qb && ((uint32_t)q == qa || tot + cnt[q] <= 64u)) {
          tot += cnt[q];
          mask |= 1u << q;
          qb++;
        }
      }
      qa = qb;
      mask &= gstore;
      if (mask == 0u) continue;
      bool bail = false;
      if (tot != 0u) {
        // busy blocks hand their first eight bytes to lane `rank`: bytes 136..143 of that lane's scratch (the
        // coefficients end at 128, the parse's dump is 128..129)
        uint32_t base = 0u;
#pragma unroll
        for (int q = 0; q < kPoolGroups; q++) {
          if (mask >> q & 1u) {
            if (((info[q] >> 8) & 3u) == 2u) {
              const uint32_t k = base + (info[q] >> 10);
              info[q] = (info[q] & 0x3FFu) | (k << 10);
              const uint32_t ea = lds0 + k * (uint32_t)(kCoefStride * 2) + 136u;
              *(lds_u32_t*)(uintptr_t)ea = first4[q];
              *(lds_u32_t*)(uintptr_t)(ea + 4u) = second4[q];
            }
            base += cnt[q];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (one wave, LDS operations in order: for the compiler)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool vlive = (uint32_t)lane < tot;
        bool bad = false;
        if (vlive) {
          // ---- stream -> dequantised coefficients (lib/RTjpeg.c:157-186): DC, then seven tokens ----
          const uint32_t w0 = *(const lds_u32_t*)(uintptr_t)(my_a + 136u), w1 = *(const lds_u32_t*)(uintptr_t)(my_a + 140u);
          {
            uint4* z = (uint4*)my;
#pragma unroll
            for (int i = 0; i < 8; i++) z[i] = make_uint4(0, 0, 0, 0);
          }
          const uint32_t e0 = *(const lds_u32_t*)(uintptr_t)tab_c;
          *(lds_i16_t*)(uintptr_t)(my_a + (uint32_t)slot_byte(0)) = (int16_t)mul_byte_hi16(w0, e0, 0, false);
          int ca = (int)tab_c + 4;
          int svb[7];
          uint32_t e[7];
#pragma unroll
          for (int k = 1; k < 8; k++) {
            const uint32_t w = k < 4 ? w0 : w1;
            svb[k - 1] = sbyte_minus(w, k & 3, k63);  // token - 63: > 0 for a run, its length
            e[k - 1] = *(const lds_u32_t*)(uintptr_t)(uint32_t)ca;
            ca = med3_i32(ca + 4, (svb[k - 1] << 2) + ca, cend);
          }
#pragma unroll
          for (int k = 1; k < 8; k++) {
            const uint32_t w = k < 4 ? w0 : w1;
            int prod = mul_byte_hi16(w, e[k - 1], k & 3, true);
            prod = svb[k - 1] > 0 ? 0 : prod;
            *(lds_i16_t*)(uintptr_t)(my_a + (e[k - 1] & 0xFFFFu)) = (int16_t)prod;
          }
          bad = ca < cend;  // (cannot be: the block is at most eight bytes long)
          uint32_t hi = 0u;  // anything outside rows 0-3 of columns 0-3?
#pragma unroll
          for (int i = 0; i < 8; i++) {
            if (i == 0 || i == 2) continue;
            const uint4 q = my[i];
            hi |= q.x | q.y | q.z | q.w;
          }
          bad = bad || hi != 0u;
        }
        bail = __any(bad);  // wave-uniform
        MIRTJ_PSTAMP(3);  // hand-over through LDS, parse
        if (bail) {
          bailed |= mask;  // (appended behind the counted wait)
        } else if (vlive) {
          uint32_t vrow = my_a;  // the rows go back into the lane's scratch (its coefficients are in registers by then)
          transform_lo(my, K, KP, [&](uint2 o) {
            *(lds_u32_t*)(uintptr_t)vrow = o.x;
            *(lds_u32_t*)(uintptr_t)(vrow + 4u) = o.y;
            vrow += 8u;
          });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      MIRTJ_PSTAMP(4);  // transform, rows into LDS
      if (bail) continue;
      // ---- the rows of the pooled groups leave: whole row segments, as from a plain round ----
#pragma unroll
      for (int q = 0; q < kPoolGroups; q++) {
        if (mask >> q & 1u) {
          stored++;
          const uint32_t cls = (info[q] >> 8) & 3u;
          if (cls != 0u) {
            // macroblock coordinates: one division for the group's first macroblock, then at most one row wrap per lane
            const uint32_t grp = sg * (uint32_t)kPoolGroups + (uint32_t)q;
            const uint32_t mbw = f.mbw, mb0 = grp * (uint32_t)kMbPerGroup;
            const uint32_t gy = mb0 / mbw, gx = mb0 - gy * mbw;  // wave-uniform
            uint32_t mx, my_;
            if (mbw >= (uint32_t)kMbPerGroup) {
              const bool wrap = gx + dmb >= mbw;
              mx = wrap ? gx + dmb - mbw : gx + dmb;
              my_ = wrap ? gy + 1u : gy;
            } else {
              my_ = (mb0 + dmb) / mbw;
              mx = mb0 + dmb - my_ * mbw;
            }
            const uint32_t o = (kblk == 5u ? (uint32_t)(ysz >> 2) : 0u) + 8u * my_ * stride + 8u * mx;
            uint8_t* plane = plane0;  // wave-uniform; steps from row to row on the scalar side
            const uint32_t fill = (info[q] & 0xFFu) * 0x01010101u;
            uint32_t ra = lds0 + (info[q] >> 10) * (uint32_t)(kCoefStride * 2);
#pragma unroll
            for (int r = 0; r < kRowStores; r++) {
              mirtj_u32x2 ov;
              ov.x = fill;
              ov.y = fill;
              if (cls == 2u) {
                ov.x = *(const lds_u32_t*)(uintptr_t)ra;
                ov.y = *(const lds_u32_t*)(uintptr_t)(ra + 4u);
              }
              // plain stores: as nontemporal ones these 8-byte-per-lane rows reached memory as 22.8 GB per launch for
              // 17.1 GB of chroma planes (WRITE_SIZE; whatever the row alignment), plain ones as 17.2; the time is the
              // same (profiles/r04/chroma_store_traffic.txt)
              *(mirtj_u32x2*)(plane + o) = ov;
              plane += stride;
              ra += 8u;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the scratch is written again by the next round's parse
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      MIRTJ_PSTAMP(5);  // row stores
    }
    auto push_bailed = [&](uint32_t sgx, uint32_t bits) {
#pragma unroll
      for (int q = 0; q < kPoolGroups; q++)
        if (bits >> q & 1u) declist_push(list, fidx, sgx * (uint32_t)kPoolGroups + (uint32_t)q, 2u);
    };
    if (!have_n) {
      exit_sg = sg;
      exit_bailed = bailed;
      break;
    }
    // ---- the counted wait: everything but this round's row stores (8 per stored group) has arrived ----
    nstores = stored;
    asm volatile(
        "s_cmp_lg_u32 %0, 0\n\t"
        "s_cbranch_scc1 .Lmirtj_cw1_%=\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_branch .Lmirtj_cwe_%=\n"
        ".Lmirtj_cw1_%=:\n\t"
        "s_cmp_lg_u32 %0, 1\n\t"
        "s_cbranch_scc1 .Lmirtj_cw2_%=\n\t"
        "s_waitcnt vmcnt(8)\n\t"
        "s_branch .Lmirtj_cwe_%=\n"
        ".Lmirtj_cw2_%=:\n\t"
        "s_cmp_lg_u32 %0, 2\n\t"
        "s_cbranch_scc1 .Lmirtj_cw3_%=\n\t"
        "s_waitcnt vmcnt(16)\n\t"
        "s_branch .Lmirtj_cwe_%=\n"
        ".Lmirtj_cw3_%=:\n\t"
        "s_waitcnt vmcnt(24) ; mirtj chroma pool wait\n"
        ".Lmirtj_cwe_%=:"
        :
        : "s"(nstores)
        : "scc", "memory");
    // copies of our own, BEHIND the wait (volatile asm statements keep their order): tied to the wait block as in/out
    // operands the registers were copied by the compiler IN FRONT of it (tools/check_async_loads.py found exactly that)
#define MIRTJ_COPY(D, S) asm volatile("v_mov_b32 %0, %1" : "=v"(D) : "v"(S))
#pragma unroll
    for (int q = 0; q < kPoolGroups; q++) {
      MIRTJ_COPY(B[q].x, nB[q].x);
      MIRTJ_COPY(B[q].y, nB[q].y);
      MIRTJ_COPY(B[q].z, nB[q].z);
      MIRTJ_COPY(O[q].x, nO[q].x);
      MIRTJ_COPY(O[q].y, nO[q].y);
    }
#undef MIRTJ_COPY
    MIRTJ_PSTAMP(6);  // the counted wait
    if (bailed) push_bailed(sg, bailed);
    cur_m = nxt_m;
  }
#ifdef MIRTJ_EXPERIMENTS
  if (lane == 0) {
    for (int i = 0; i < 7; i++) atomicAdd(&g_pool_stamps[i], pst[i]);
    atomicAdd(&g_pool_stamps[7], 1ull);
  }
#endif
  if (exit_bailed)  // the last round's, behind the loop
    for (int q = 0; q < kPoolGroups; q++)
      if (exit_bailed >> q & 1u) declist_push(list, fidx, exit_sg * (uint32_t)kPoolGroups + (uint32_t)q, 2u);
}

// ---------------------------------------------------------------------------------------
// k_decode_split: the batch form of k_decode since round 4.  grid (8 * (L + C), frames): workgroup w of a frame is wave
// r = w / 8 of XCD x = w % 8 (workgroups are dealt to the eight XCDs round-robin by their linear number, and 8 * (L + C)
// per frame keeps that number's remainder the same for every frame).  An XCD owns the SUPER GROUPS (three consecutive
// macroblock groups: what a pooling chroma wave takes per round) x, x + 8, x + 16, ...: its L luma waves take them in
// turn (decode_wave, the two luma parts of each group; blocks outside the packed passes' budget go to the list), its C
// chroma waves pool their chroma (above).  So the lines of a stretch of the packet and of its block offsets, which luma
// and chroma waves both read, are fetched into ONE L2 — round 4's first version dealt the waves without regard to the
// XCDs and fetched them twice (26.9 GB per launch instead of 13.8: profiles/r04/).  The affinity is for speed only:
// nothing depends on which XCD a workgroup lands on.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kXcds = 8;
#ifndef MIRTJ_SPLIT_LUMA_SG
#define MIRTJ_SPLIT_LUMA_SG 3
#endif
constexpr int kSplitLumaSuperGroups = MIRTJ_SPLIT_LUMA_SG;  // super groups a luma wave takes at most
constexpr uint32_t kSplitXcdRot = 1;  // stripes move on by one XCD per frame (k_decode_split)
// luma / chroma waves per XCD and frame: a luma wave takes up to kSplitLumaSuperGroups super groups (9 groups; round 3's
// waves took 11 with their chroma), a chroma wave up to kPoolItersMax.  How many there are of each decides more than how
// long they run: with L + C EVEN the launch is 2 ... 25 % slower than with the odd counts next to it (1080p x 16,384, 11
// super groups per XCD, C = 1: L = 2 16.4 ms, 3 17.8, 4 16.2, 5 16.6, 7 19.8, 8 16.4, 9 17.0, 10 16.5, 11 21.0, 12 16.4,
// 13 17.4; two chroma waves of half the length 23 ... 24 ms: profiles/r04/split_waves_per_xcd.txt) — an XCD deals its
// workgroups to its 32 CUs in turn, and an even period puts the chroma waves, which run longest, on a fraction of them.
// So L is made to leave L + C odd.
__host__ __device__ constexpr uint32_t split_chroma_waves(uint32_t groups) {
  const uint32_t sg = (groups + (uint32_t)kPoolGroups - 1u) / (uint32_t)kPoolGroups, per_xcd = (sg + kXcds - 1u) / kXcds;
  return per_xcd ? (per_xcd + (uint32_t)kPoolItersMax - 1u) / (uint32_t)kPoolItersMax : 1u;
}
__host__ __device__ constexpr uint32_t split_luma_waves(uint32_t groups) {
  const uint32_t sg = (groups + (uint32_t)kPoolGroups - 1u) / (uint32_t)kPoolGroups, per_xcd = (sg + kXcds - 1u) / kXcds;
  const uint32_t lw = per_xcd ? (per_xcd + (uint32_t)kSplitLumaSuperGroups - 1u) / (uint32_t)kSplitLumaSuperGroups : 1u;
  return ((lw + split_chroma_waves(groups)) & 1u) ? lw : lw + 1u;
}
__global__ __launch_bounds__(kDecThreads, MIRTJ_DEC_WAVES) void k_decode_split(const FrameDev* __restrict__ frames,
                                                               const uint8_t* __restrict__ stream,
                                                               const QTab* __restrict__ lut,
                                                               const uint32_t* __restrict__ blkoff,
                                                               uint8_t* __restrict__ outbuf, const uint32_t luma_waves,
                                                               const uint32_t chroma_waves, const DecList list,
                                                               const uint32_t* __restrict__ only_in_mode,
                                                               const uint32_t xcd_rot) {
  if (only_in_mode && *only_in_mode != kDecModeSplit) return;  // the plan's policy has the classic form run (DecPolicy)
  __shared__ __attribute__((aligned(16))) uint32_t s_lds[kDecLdsWords];
  // (the stripe of super groups an XCD owns moves on by xcd_rot from frame to frame: every XCD gets every stripe in turn)
  const uint32_t x = (blockIdx.x + blockIdx.y * (xcd_rot & 7u)) % kXcds;
  const uint32_t r = blockIdx.x / kXcds;
  if (r < luma_waves)
    decode_wave<true, false, 2, true, true>(s_lds, frames, blockIdx.y, x + kXcds * r, kXcds * luma_waves, 0u, stream, lut,
                                            blkoff, outbuf, nullptr, list);
  else if (chroma_waves)
    chroma_pool_wave(s_lds, frames, blockIdx.y, x + kXcds * (r - luma_waves), kXcds * chroma_waves, stream, lut, blkoff,
                     outbuf, list);
}

// k_decode_list: the parts k_decode_split left over, one wave per entry at a time, decode_wave's general path
// (every form of the transform).  A fixed grid that loops; an empty list costs a launch and one load per wave.
constexpr unsigned kDecListGrid = 4096;
__global__ __launch_bounds__(kDecThreads, MIRTJ_DEC_WAVES) void k_decode_list(const FrameDev* __restrict__ frames,
                                                              const uint8_t* __restrict__ stream,
                                                              const QTab* __restrict__ lut,
                                                              const uint32_t* __restrict__ blkoff,
                                                              uint8_t* __restrict__ outbuf, const DecList list,
                                                              uint32_t* __restrict__ next_count,
                                                              uint32_t* __restrict__ policy, const uint32_t parts_total,
                                                              uint32_t* __restrict__ mode_seen) {
  __shared__ __attribute__((aligned(16))) uint32_t s_lds[kDecLdsWords];
  uint32_t n = *list.count;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (the next launch's counter: nobody reads it during this launch, and the next launch's kernels run behind this one)
    *next_count = 0u;
    if (policy) {  // the books of the plan's decode policy (rtj_decode_kernels.h): nobody else reads them in this kernel
      if (policy[0] == kDecModeSplit) {
        if (n > parts_total / kDecListShare) {
          policy[0] = kDecModeClassic;
          policy[1] = kDecClassicLaunches;
        }
      } else if (--policy[1] == 0u) {
        policy[0] = kDecModeSplit;
      }
      // what the next launch will run, where the host can see it without waiting (pinned host memory; it decides whether
      // the classic form's kernel is enqueued at all: mi_rtjpeg.hip, h_mode_seen)
      if (mode_seen) __hip_atomic_store(mode_seen, policy[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  n = n < list.cap ? n : list.cap;
  for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
    const uint2 it = list.items[e];
    // (wave-uniform, and the compiler has to know: the wave's base addresses live in scalar registers)
    const uint32_t fr = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.x), gp = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.y);
    const uint32_t grp = gp & 0x0FFFFFFFu, part = gp >> 28;
    // slot = the group, slots = "no second group": one part of one group
    decode_wave<false, false, 3, false>(s_lds, frames, fr, grp, 0x04000000u, part, stream, lut, blkoff, outbuf,
                                        nullptr, DecList{nullptr, nullptr, 0u});
    __syncthreads();  // the next entry rewrites the slot tables
  }
}

}  // namespace mirtj
