// rtj_index_kernels.h — parallel block-offset index for RTjpeg packets (gfx950).
//
// A block's length is only known after reading it (RTjpeg_s2b returns it, lib/RTjpeg.c:185), so
// block n's offset depends on every block before it — 48,960 dependent steps per 1080p frame.
// The index is therefore built in three launches that break the chain at fixed byte positions:
//
//   k_index_summarize  per chunk of kChunk stream bytes: length of a block of either type at
//                      EVERY byte position (prefix sums of token weights + a bit-by-bit search
//                      of the 64-byte window), composed into "next macroblock" jumps; every one
//                      of the kEntries possible entry offsets takes one jump, the distinct targets
//                      are walked to the chunk's end -> (exit offset, MB count) per entry.  The
//                      block lengths are left in HBM for k_index_emit.
//   k_index_resolve    per packet: chains the chunk summaries (one short serial walk in LDS) to
//                      the true entry offset and first macroblock number of every chunk.
//   k_index_emit       per chunk: macroblock jumps from the stored lengths, one serial walk over
//                      the chunk's own macroblocks from its true entry, then the byte offset of
//                      every block.  (k_index_emit_walk re-walks the stream bytes instead: A/B.)
//
// Block-length rule (lib/RTjpeg.c:157-186, 2704): first byte 0xFF -> 1 byte; otherwise 1 DC byte,
// bt8 raw bytes, then tokens until 63-bt8 coefficient slots are covered, a token 64..127 covering
// (token-63) slots and any other byte one slot.
#pragma once
#include <hip/hip_runtime.h>

#include "rtj_common.h"
#include "rtj_decode_kernels.h"

namespace mirtj {

constexpr int kSumThreads = 256;
#ifndef MIRTJ_SEARCH_UNROLL
#define MIRTJ_SEARCH_UNROLL 8
#endif
constexpr int kSearchUnroll = MIRTJ_SEARCH_UNROLL;
#ifndef MIRTJ_F_CHAINS
#define MIRTJ_F_CHAINS 7
#endif
#ifndef MIRTJ_SUM_WAVES
#define MIRTJ_SUM_WAVES 7
#endif

// One search step: probe `step` entries ahead; still below the target -> move on by `step`.
// "Below" = bit 15 of the difference (the sums differ by < 2^15 and are kept mod 2^16).  Three
// instructions: 16-bit subtract, shift, shift-add into the byte address; the probe distance is an
// immediate offset of the LDS read.
template <int STEP>
__device__ __forceinline__ uint32_t search_step(const uint16_t* s_w, uint32_t addr, uint32_t tgt) {
  const uint32_t w = *(const uint16_t*)((const uint8_t*)s_w + addr + 2 * STEP);
  uint32_t dlt, na;
  asm("v_sub_u16 %0, %1, %2" : "=v"(dlt) : "v"(w), "v"(tgt));
  asm("v_lshrrev_b32 %0, 15, %1" : "=v"(dlt) : "v"(dlt));
  asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(na) : "v"(dlt), "n"(__builtin_ctz(2 * STEP)), "v"(addr));
  return na;
}

// Block length at every table position: s_nl[i] = length of a luma block that starts at i, s_nc[i] = of
// a chroma block (separate byte arrays: four positions to a bank keeps the macroblock chase's scattered
// reads mostly conflict-free; interleaving the two was measured slower).  A block of a type with bt8 raw bytes that starts at i has its
// last raw byte at p = i + bt8 and ends at p + d, d the smallest distance with W[p + d] >= W[p] + need,
// need = 63 - bt8.  d - 1 is built bit by bit (32, 16, .. 1); searching the full 1..64 window is exact
// because the predicate is monotone.
//   NT == 1 (lb8 == cb8, two thirds of the qualities): one search per position serves both types.
//   NT == 2: the searches are organised by p, not by i.  From one p the type with MORE raw bytes
//   needs FEWER slots, so its end comes first (six steps); the other type needs bA - bB <= 15 slots
//   more, every byte covers at least one, so its end is at most 15 bytes further: four more steps
//   from where the first search stopped.  Ten steps for both instead of twelve; the two results
//   belong to different start positions (p - bA and p - bB).
// kSearchUnroll positions are searched together so that their dependent reads overlap.
// A first byte 0xFF (block of length 1) is patched in afterwards by patch_unchanged_blocks.
template <int NT>
__device__ __forceinline__ void search_lengths(const uint16_t* s_w, uint8_t* s_nl, uint8_t* s_nc, uint32_t lb8, uint32_t cb8, int tid) {
  const bool luma_first = lb8 >= cb8;
  const uint32_t bA = luma_first ? lb8 : cb8, bB = luma_first ? cb8 : lb8;  // bA >= bB
  uint8_t* const s_a = luma_first ? s_nl : s_nc;  // lengths of the type searched first / second
  uint8_t* const s_b2 = luma_first ? s_nc : s_nl;
  const int np = kTabN + (NT == 2 ? (int)bA : 0);  // NT == 1: p runs over the start positions themselves
  for (int p0 = tid; p0 < np; p0 += kSumThreads * kSearchUnroll) {
    uint32_t addr[kSearchUnroll], tgt[kSearchUnroll];
#pragma unroll
    for (int u = 0; u < kSearchUnroll; u++) {
      const uint32_t p = (uint32_t)min(p0 + u * kSumThreads, np - 1) + (NT == 1 ? bA : 0u);
      addr[u] = 2u * p;  // byte address of W[p]
      tgt[u] = (uint32_t)s_w[p] + (63u - bA);
    }
#pragma unroll
    for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<32>(s_w, addr[u], tgt[u]);
#pragma unroll
    for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<16>(s_w, addr[u], tgt[u]);
#pragma unroll
    for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<8>(s_w, addr[u], tgt[u]);
#pragma unroll
    for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<4>(s_w, addr[u], tgt[u]);
#pragma unroll
    for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<2>(s_w, addr[u], tgt[u]);
#pragma unroll
    for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<1>(s_w, addr[u], tgt[u]);
    if (NT == 1) {
#pragma unroll
      for (int u = 0; u < kSearchUnroll; u++) {
        const int i = p0 + u * kSumThreads;
        if (i < kTabN) {
          const uint32_t len = addr[u] / 2u + 2u - (uint32_t)i;  // last byte is W index addr/2 + 1
          s_nl[i] = (uint8_t)len;
          s_nc[i] = (uint8_t)len;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < kSearchUnroll; u++) {
        const int p = p0 + u * kSumThreads;
        if (p < np && p >= (int)bA) s_a[p - (int)bA] = (uint8_t)(addr[u] / 2u + 2u - (uint32_t)(p - (int)bA));
        tgt[u] += bA - bB;
      }
#pragma unroll
      for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<8>(s_w, addr[u], tgt[u]);
#pragma unroll
      for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<4>(s_w, addr[u], tgt[u]);
#pragma unroll
      for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<2>(s_w, addr[u], tgt[u]);
#pragma unroll
      for (int u = 0; u < kSearchUnroll; u++) addr[u] = search_step<1>(s_w, addr[u], tgt[u]);
#pragma unroll
      for (int u = 0; u < kSearchUnroll; u++) {
        const int p = p0 + u * kSumThreads, i = p - (int)bB;
        if (p < np && i >= 0 && i < kTabN) s_b2[i] = (uint8_t)(addr[u] / 2u + 2u - (uint32_t)i);
      }
    }
  }
}

// A block whose first byte is 0xFF is one byte long whatever its type (lib/RTjpeg.c:2704).  Rare (only
// streams with unchanged blocks have them), so it is patched in after the search: every thread looks
// at 16 bytes at once.
__device__ __forceinline__ void patch_unchanged_blocks(const uint32_t* s_b4, uint8_t* s_nl, uint8_t* s_nc, int tid) {
  if (16 * tid >= kTabN) return;
  const uint4 q = ((const uint4*)s_b4)[tid];
  const uint32_t d[4] = {q.x, q.y, q.z, q.w};
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) any |= (~d[k] - 0x01010101u) & d[k] & 0x80808080u;  // a byte of ~d is zero
  if (any == 0u) return;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (((d[k >> 2] >> (8 * (k & 3))) & 0xFFu) == 0xFFu) s_nl[16 * tid + k] = s_nc[16 * tid + k] = 1;
}

template <int NT>
__device__ __forceinline__ void summarize_chunk(const FrameDev* __restrict__ frames, const uint8_t* __restrict__ stream,
                                                const QTab* __restrict__ lut, uint32_t* __restrict__ summary,
                                                uint16_t* __restrict__ lentab, uint32_t frame_index) {
  __shared__ __attribute__((aligned(16))) uint32_t s_b4[kStageN / 4];  // stream bytes
  __shared__ __attribute__((aligned(16))) uint16_t s_w[kStageN];       // inclusive weight sums mod 2^16
  __shared__ __attribute__((aligned(16))) uint8_t s_nl[kTabN];         // block length if luma ...
  __shared__ __attribute__((aligned(16))) uint8_t s_nc[kTabN];         // ... or chroma
  __shared__ uint32_t s_wave[kSumThreads / 64];
  __shared__ uint32_t s_grp[12];
#ifdef MIRTJ_SUM_NOALIAS
  __shared__ uint16_t s_f[kChunk];
  __shared__ uint16_t s_slot[2 * kEntries];
  __shared__ uint16_t s_list[kEntries];
  __shared__ uint32_t s_res[kEntries];
#else
  // the sums and the bytes are dead once the lengths are known (step 3): their space is reused, which
  // brings the workgroup to 20 KB of LDS, i.e. up to eight workgroups per CU instead of five
  static_assert(kChunk <= kStageN && 2 * kEntries + kEntries + 2 * kEntries <= kStageN / 2, "aliased LDS layout");
  uint16_t* const s_f = s_w;                               // macroblock length
  uint16_t* const s_slot = (uint16_t*)s_b4;                // first-hop targets: marker, then index into s_list
  uint16_t* const s_list = s_slot + 2 * kEntries;          // the distinct first-hop targets
  uint32_t* const s_res = (uint32_t*)(s_list + kEntries);  // (macroblocks << 16) | exit offset of each distinct target
#endif

  const FrameDev f = frames[frame_index];
  const uint32_t c = blockIdx.x;
  if (c >= f.nchunks) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t cbase = c * (uint32_t)kChunk;

  // ---- 1. stage kStageN bytes starting at cbase; bytes at or past data_len read as 0 ----
  {
    const uint8_t* g = stream + f.data_off + cbase;
    const uint32_t mis = (uint32_t)((uintptr_t)g & 3u);
    const uint32_t* g4 = (const uint32_t*)(g - mis);
    // bytes of the packet left from the chunk start on (plans keep data_len below 2^31)
    const uint32_t left = f.data_len > cbase ? f.data_len - cbase : 0u;
    if (!kForceGenericPaths && left >= (uint32_t)kStageN + 4u) {
      // the whole window (and the dword after it) lies inside the packet: all chunks but a packet's last two
      for (int j = tid; j < kStageN / 4; j += kSumThreads)
        s_b4[j] = __builtin_amdgcn_alignbyte(g4[j + 1], g4[j], mis);  // mis is uniform: a funnel shift by 0 is a copy
    } else {
      for (int j = tid; j < kStageN / 4; j += kSumThreads) {
        // LDS dword j = stream bytes cbase+4j .. +3 = aligned global dwords j, j+1 shifted by mis;
        // g4[j] starts mis bytes before chunk byte 4j
        const uint32_t b = 4u * (uint32_t)j;
        uint32_t lo = 0, hi = 0;
        if (b < left + mis) lo = g4[j];
        if (mis && b + 4u < left + mis) hi = g4[j + 1];
        uint32_t v = mis ? __builtin_amdgcn_alignbyte(hi, lo, mis) : lo;
        if (b >= left) v = 0;
        else if (left - b < 4u) v &= (1u << (8u * (left - b))) - 1u;
        s_b4[j] = v;
      }
    }
  }
  __syncthreads();

  // ---- 2. inclusive prefix sum of token weights, 16 positions per thread ----
  {
    const uint4 q = ((const uint4*)s_b4)[tid];
    const uint32_t d[4] = {q.x, q.y, q.z, q.w};
    uint32_t loc[16];
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      // token_weight(b) == max((int8)b - 63, 1): the byte select and sign extension ride on the subtract
      run += (uint32_t)max(sbyte_minus(d[i >> 2], i & 3, 63), 1);
      loc[i] = run;
    }
    const uint32_t incl = wave_incl_scan(run);
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint32_t off = incl - run;
    for (int k = 0; k < wv; k++) off += s_wave[k];
    uint32_t pk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) pk[i] = __builtin_amdgcn_perm(loc[2 * i + 1] + off, loc[2 * i] + off, 0x05040100u);  // low halves
    uint4* dst = (uint4*)(s_w + 16 * tid);
    dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
  }
  __syncthreads();

  // ---- 3. block length at every position, for both block types (see search_lengths) ----
  const uint32_t lb8 = (uint32_t)lut[f.qidx].lb8, cb8 = (uint32_t)lut[f.qidx].cb8;
  // the host picks the instantiation per plan: NT == 1 when every packet of the launch has lb8 == cb8
  search_lengths<NT>(s_w, s_nl, s_nc, lb8, cb8, tid);
  __syncthreads();
  patch_unchanged_blocks(s_b4, s_nl, s_nc, tid);
  __syncthreads();

  // both lengths of the chunk's own positions go to HBM for k_index_emit: (luma | chroma << 8)
  {
    uint4* dst = (uint4*)(lentab + (size_t)(f.sum_base + c) * kChunk);
    for (int i = tid; i < kChunk / 8; i += kSumThreads) {
      const uint2 l = ((const uint2*)s_nl)[i], h = ((const uint2*)s_nc)[i];
      uint4 o;
      o.x = __builtin_amdgcn_perm(h.x, l.x, 0x05010400u);  // bytes l0 h0 l1 h1
      o.y = __builtin_amdgcn_perm(h.x, l.x, 0x07030602u);  //       l2 h2 l3 h3
      o.z = __builtin_amdgcn_perm(h.y, l.y, 0x05010400u);
      o.w = __builtin_amdgcn_perm(h.y, l.y, 0x07030602u);
      dst[i] = o;
    }
  }

  // ---- 4. macroblock length: four luma blocks then two chroma blocks ----
  // kFChains positions per thread are chased together: the six lookups of one position depend on each
  // other, those of different positions do not
  {
    constexpr int kFChains = MIRTJ_F_CHAINS;
    static_assert(kChunk % (kSumThreads * kFChains) == 0, "macroblock-length pass covers the chunk in whole rounds");
    for (int p0 = tid; p0 < kChunk; p0 += kSumThreads * kFChains) {
      uint32_t q[kFChains];
#pragma unroll
      for (int u = 0; u < kFChains; u++) q[u] = (uint32_t)(p0 + u * kSumThreads);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        uint32_t l[kFChains];
#pragma unroll
        for (int u = 0; u < kFChains; u++) l[u] = k < 4 ? s_nl[q[u]] : s_nc[q[u]];
#pragma unroll
        for (int u = 0; u < kFChains; u++) q[u] += l[u];
      }
#pragma unroll
      for (int u = 0; u < kFChains; u++) s_f[p0 + u * kSumThreads] = (uint16_t)(q[u] - (uint32_t)(p0 + u * kSumThreads));
    }
  }
  __syncthreads();
  // ---- 5. every possible entry offset walked to the end of the chunk ----
  // The 384 trails merge almost at once (after one macroblock a few dozen distinct positions are
  // left), so each entry takes one hop, the distinct targets are compacted, only those are walked
  // to the end, and every entry picks up the result of its target.
  uint32_t* out = summary + (size_t)(f.sum_base + c) * kEntries;
  constexpr int kHopN = 2 * kEntries;  // a first hop lands below this
  static_assert(kHopN == 3 * kSumThreads && kEntries <= 2 * kSumThreads, "compaction layout");
  const bool two = tid + kSumThreads < kEntries;
  const uint32_t h0 = tid + s_f[tid], h1 = two ? tid + kSumThreads + s_f[tid + kSumThreads] : h0;
  for (int i = tid; i < kHopN; i += kSumThreads) s_slot[i] = 0xFFFFu;
  __syncthreads();
  s_slot[h0] = 0;
  s_slot[h1] = 0;
  __syncthreads();
  {
    // compaction of the marked positions, in position order: group g = k*4 + wave covers 64 positions
    unsigned long long m[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      m[k] = __ballot(s_slot[k * kSumThreads + tid] != 0xFFFFu);
      if (lane == 0) s_grp[k * 4 + wv] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int g = 0; g < 12; g++) {
      const uint32_t n = s_grp[g];
      total += n;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      base = 0;
#pragma unroll
      for (int g = 0; g < 12; g++) base += g < k * 4 + wv ? s_grp[g] : 0u;
      if ((m[k] >> lane) & 1ull) {
        const uint32_t idx = base + (uint32_t)__popcll(m[k] & below);
        s_list[idx] = (uint16_t)(k * kSumThreads + tid);
        s_slot[k * kSumThreads + tid] = (uint16_t)idx;
      }
    }
    __syncthreads();
    for (uint32_t u = tid; u < total; u += kSumThreads) {
      uint32_t p = s_list[u], cnt = 1;  // the hop that led here is counted
      while (p < (uint32_t)kChunk) {
        p += s_f[p];
        cnt++;
      }
      s_res[u] = (cnt << 16) | (p - (uint32_t)kChunk);
    }
    __syncthreads();
  }
  out[tid] = s_res[s_slot[h0]];
  if (two) out[tid + kSumThreads] = s_res[s_slot[h1]];
}

// grid (chunks, frames)
template <int NT>
__global__ __launch_bounds__(kSumThreads, MIRTJ_SUM_WAVES) void k_index_summarize(const FrameDev* __restrict__ frames,
                                                                  const uint8_t* __restrict__ stream,
                                                                  const QTab* __restrict__ lut,
                                                                  uint32_t* __restrict__ summary,
                                                                  uint16_t* __restrict__ lentab) {
  summarize_chunk<NT>(frames, stream, lut, summary, lentab, blockIdx.y);
}

// grid (chunks, rows): the rows loop over the to-do list of packets the speculative index refused
// (rtj_spec_kernels.h).  A kernel of its own: with both forms in one kernel the direct one ran 6 % slower.
template <int NT>
__global__ __launch_bounds__(kSumThreads, MIRTJ_SUM_WAVES) void k_index_summarize_todo(const FrameDev* __restrict__ frames,
                                                                       const uint8_t* __restrict__ stream,
                                                                       const QTab* __restrict__ lut,
                                                                       uint32_t* __restrict__ summary,
                                                                       uint16_t* __restrict__ lentab,
                                                                       const uint32_t* __restrict__ todo,
                                                                       const uint32_t* __restrict__ ntodo) {
  const uint32_t n = *ntodo;
  for (uint32_t i = blockIdx.y; i < n; i += gridDim.y) {
    summarize_chunk<NT>(frames, stream, lut, summary, lentab, todo[i]);
    __syncthreads();  // the next packet's chunk reuses the LDS
  }
}

// One workgroup per packet.  Summaries are pulled through LDS a tile at a time; lane 0 chains them.
constexpr int kResTile = 24;
__device__ __forceinline__ void resolve_packet(const FrameDev* __restrict__ frames, const uint32_t* __restrict__ summary,
                                               uint32_t* __restrict__ chunk_pos, uint32_t* __restrict__ chunk_mb,
                                               uint32_t frame_index) {
  __shared__ uint32_t s_sum[kResTile * kEntries];
  __shared__ uint32_t s_state[2];
  const FrameDev f = frames[frame_index];
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_state[0] = 0;  // entry offset into the current chunk
    s_state[1] = 0;  // macroblocks before it
  }
  // the next tile's summaries are requested into registers before the current tile is chained, so
  // that their latency hides behind the serial part
  constexpr int kPer = kResTile * kEntries / 256;
  static_assert(kPer * 256 == kResTile * kEntries, "tile is a whole number of rounds");
  uint32_t pre[kPer];
  auto request = [&](uint32_t c0) {
    const uint32_t nt = min((uint32_t)kResTile, f.nchunks - c0);
    const uint32_t* src = summary + (size_t)(f.sum_base + c0) * kEntries;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const uint32_t i = (uint32_t)tid + 256u * (uint32_t)k;
      pre[k] = i < nt * kEntries ? src[i] : 0u;
    }
  };
  if (f.nchunks) request(0);
  for (uint32_t c0 = 0; c0 < f.nchunks; c0 += kResTile) {
    const uint32_t nt = min((uint32_t)kResTile, f.nchunks - c0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) s_sum[tid + 256 * k] = pre[k];
    __syncthreads();
    if (c0 + kResTile < f.nchunks) request(c0 + kResTile);
    if (tid == 0) {
      uint32_t e = s_state[0], mb = s_state[1];
      for (uint32_t j = 0; j < nt; j++) {
        chunk_pos[f.chunk_base + c0 + j] = (c0 + j) * (uint32_t)kChunk + e;
        chunk_mb[f.chunk_base + c0 + j] = min(mb, f.nmb);
        const uint32_t v = s_sum[j * kEntries + e];
        e = v & 0xFFFFu;
        mb += v >> 16;
      }
      s_state[0] = e;
      s_state[1] = mb;
    }
  }
  // whatever the stream holds past the last macroblock is ignored; a packet that ends early is
  // continued with zero bytes by the last chunk's walker
  if (tid == 0) {
    chunk_pos[f.chunk_base + f.nchunks] = 0;
    chunk_mb[f.chunk_base + f.nchunks] = f.nmb;
  }
}

__global__ __launch_bounds__(256) void k_index_resolve(const FrameDev* __restrict__ frames,
                                                        const uint32_t* __restrict__ summary,
                                                        uint32_t* __restrict__ chunk_pos,
                                                        uint32_t* __restrict__ chunk_mb,
                                                        const uint32_t* __restrict__ todo,
                                                        const uint32_t* __restrict__ ntodo) {
  if (!todo) {
    resolve_packet(frames, summary, chunk_pos, chunk_mb, blockIdx.x);
    return;
  }
  const uint32_t n = *ntodo;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    resolve_packet(frames, summary, chunk_pos, chunk_mb, todo[i]);
    __syncthreads();
  }
}

// A/B baseline of k_index_emit (MI_RTJ_EMIT=walk): one wave per chunk re-walks the chunk's own
// blocks from the stream bytes.
__global__ __launch_bounds__(64) void k_index_emit_walk(const FrameDev* __restrict__ frames,
                                                    const uint8_t* __restrict__ stream,
                                                    const QTab* __restrict__ lut,
                                                    const uint32_t* __restrict__ chunk_pos,
                                                    const uint32_t* __restrict__ chunk_mb,
                                                    uint32_t* __restrict__ blkoff) {
  const FrameDev f = frames[blockIdx.y];
  const uint32_t c = blockIdx.x;
  if (c >= f.nchunks) return;
  const uint32_t m0 = chunk_mb[f.chunk_base + c];
  const uint32_t m1 = chunk_mb[f.chunk_base + c + 1];  // == nmb for the last chunk
  if (m0 >= m1) return;
  walk_blocks(f, stream, lut, chunk_pos[f.chunk_base + c], 6u * m0, 6u * m1, m1 == f.nmb, blkoff + f.blk_base);
}

// k_index_emit: per chunk, from the block lengths k_index_summarize left in HBM and the true
// entry k_index_resolve found: macroblock lengths in parallel, one short serial walk over the
// chunk's macroblocks, then the six block offsets of every macroblock in parallel.
#ifndef MIRTJ_EMIT_THREADS
#define MIRTJ_EMIT_THREADS 256
#endif
constexpr int kEmitThreads = MIRTJ_EMIT_THREADS;
constexpr int kMaxMbPerChunk = kChunk / 6 + 2;  // a macroblock is at least six 1-byte blocks
__device__ __forceinline__ void emit_chunk(const FrameDev* __restrict__ frames, const uint16_t* __restrict__ lentab,
                                           const uint32_t* __restrict__ chunk_pos, const uint32_t* __restrict__ chunk_mb,
                                           uint32_t* __restrict__ blkoff, uint32_t frame_index) {
  __shared__ __attribute__((aligned(16))) uint16_t s_len[kTabN + 8];  // (luma | chroma << 8) per position
  __shared__ uint16_t s_f[kChunk];   // macroblock length
  __shared__ uint16_t s_mb[kMaxMbPerChunk + 2];
  __shared__ uint32_t s_cnt[2];

  const FrameDev f = frames[frame_index];
  const uint32_t c = blockIdx.x;
  if (c >= f.nchunks) return;
  const int tid = threadIdx.x;
  const uint32_t cbase = c * (uint32_t)kChunk;

  // ---- block lengths of [cbase, cbase + kTabN); past the packet's last chunk every byte is 0,
  //      and a block of zero bytes is 64 bytes long whatever its type.  The kernel is bound by the
  //      latency of its dependent steps, so this load is issued before anything else ----
  constexpr int kPieces = (kTabN + 7) / 8, kPer = (kPieces + kEmitThreads - 1) / kEmitThreads;
  uint4 lens[kPer];
  {
    const size_t have = (size_t)(f.nchunks - c) * kChunk;  // table entries from cbase to the end of the packet
    const uint4* src = (const uint4*)(lentab + (size_t)(f.sum_base + c) * kChunk);
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = tid + k * kEmitThreads;
      lens[k] = make_uint4(0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u);
      if (i < kPieces && (size_t)i * 8 + 8 <= have) lens[k] = src[i];  // kChunk % 8 == 0: no piece straddles the end
    }
  }
  const uint32_t m0 = chunk_mb[f.chunk_base + c];
  const uint32_t m1 = chunk_mb[f.chunk_base + c + 1];  // == nmb for the last chunk
  if (m0 >= m1) return;
  const uint32_t q0 = chunk_pos[f.chunk_base + c] - cbase;  // entry offset, < kEntries
  uint32_t* out = blkoff + f.blk_base;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int i = tid + k * kEmitThreads;
    if (i < kPieces) ((uint4*)s_len)[i] = lens[k];
  }
  __syncthreads();

  // ---- macroblock length at every position: six dependent reads, kEmitChains positions chased together.
  //      The position is kept as a byte address into s_len (luma length at +0, chroma at +1), so that a
  //      step is one byte read and one shift-add ----
  {
    constexpr int kEmitChains = 7;
    static_assert(kChunk % (kEmitThreads * kEmitChains) == 0 || kEmitThreads != 256, "whole rounds");
    const uint8_t* lb = (const uint8_t*)s_len;
    for (int p0 = tid; p0 < kChunk; p0 += kEmitChains * kEmitThreads) {
      uint32_t q[kEmitChains];
#pragma unroll
      for (int u = 0; u < kEmitChains; u++) q[u] = 2u * (uint32_t)min(p0 + u * kEmitThreads, kChunk - 1);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        uint32_t l[kEmitChains];
#pragma unroll
        for (int u = 0; u < kEmitChains; u++) l[u] = lb[q[u] + (k < 4 ? 0u : 1u)];
#pragma unroll
        for (int u = 0; u < kEmitChains; u++) q[u] += 2u * l[u];
      }
#pragma unroll
      for (int u = 0; u < kEmitChains; u++) {
        const int p = p0 + u * kEmitThreads;
        if (p < kChunk) s_f[p] = (uint16_t)(q[u] / 2u - (uint32_t)p);
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t q = q0, n = 0;
    const uint32_t want = min(m1 - m0, (uint32_t)kMaxMbPerChunk);
    while (q < (uint32_t)kChunk && n < want) {
      s_mb[n++] = (uint16_t)q;
      q += s_f[q];
    }
    s_cnt[0] = n;  // macroblocks that start inside the chunk
    s_cnt[1] = q;  // where the next one starts
  }
  __syncthreads();
  const uint32_t n_in = s_cnt[0], q_end = s_cnt[1];
  for (uint32_t i = tid; i < n_in; i += kEmitThreads) {
    uint32_t q = s_mb[i];
    uint32_t* o = out + 6u * (m0 + i);
    o[0] = cbase + q;
    q += s_len[q] & 0xFFu;
    o[1] = cbase + q;
    q += s_len[q] & 0xFFu;
    o[2] = cbase + q;
    q += s_len[q] & 0xFFu;
    o[3] = cbase + q;
    q += s_len[q] & 0xFFu;
    o[4] = cbase + q;
    q += s_len[q] >> 8;
    o[5] = cbase + q;
  }
  // Macroblocks the packet does not hold any more (it ended early): only the last chunk gets
  // here with m0 + n_in < m1, and from q_end on every byte reads as 0: 64-byte blocks.
  const uint32_t n_tail = (m1 - m0) - n_in;
  for (uint32_t i = tid; i < n_tail * 6u; i += kEmitThreads) out[6u * (m0 + n_in) + i] = cbase + q_end + 64u * i;
  if (tid == 0 && m1 == f.nmb) out[6u * f.nmb] = cbase + q_end + 384u * n_tail;
}

__global__ __launch_bounds__(kEmitThreads) void k_index_emit(const FrameDev* __restrict__ frames,
                                                              const uint16_t* __restrict__ lentab,
                                                              const uint32_t* __restrict__ chunk_pos,
                                                              const uint32_t* __restrict__ chunk_mb,
                                                              uint32_t* __restrict__ blkoff,
                                                              const uint32_t* __restrict__ todo,
                                                              const uint32_t* __restrict__ ntodo) {
  if (!todo) {
    emit_chunk(frames, lentab, chunk_pos, chunk_mb, blkoff, blockIdx.y);
    return;
  }
  const uint32_t n = *ntodo;
  for (uint32_t i = blockIdx.y; i < n; i += gridDim.y) {
    emit_chunk(frames, lentab, chunk_pos, chunk_mb, blkoff, todo[i]);
    __syncthreads();
  }
}

}  // namespace mirtj
