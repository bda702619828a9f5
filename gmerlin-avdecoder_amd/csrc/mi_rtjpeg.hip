// mi_rtjpeg.hip — host side of libmi_rtjpeg.so: the C ABI of include/mi_rtjpeg.h over the
// gfx950 kernels.  No CPU decode path exists here: without a usable device every entry point
// fails and reports why.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi_rtjpeg.h"
#include "rtj_color_kernels.h"
#include "rtj_common.h"
#include "rtj_decode_chroma.h"
#include "rtj_decode_kernels.h"
#include "rtj_encode_kernels.h"
#include "rtj_index_kernels.h"
#include "rtj_spec_kernels.h"
#include "rtj_tables.h"

using namespace mirtj;

namespace {

constexpr size_t kAllocPad = 256;  // kernels may read a few bytes past a packet's last dword
constexpr int kMaxPlanFrames = 65535;       // a plan's frames are the y dimension of every grid
constexpr uint64_t kOverlapMaxGroups = 8192ull * 255ull;  // plans from this many macroblock groups on run their kernels back to back

std::mutex g_err_mu;
std::string g_create_err = "";

struct Timed {
  hipEvent_t a = nullptr, b = nullptr;
  bool owns_a = true;  // false: `a` is the previous kernel's `b` (kernels run back to back on one stream)
};

}  // namespace

struct mi_rtj_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  QTab* d_lut = nullptr;
  uint8_t b8[kNumQTab][2] = {};  // host copy of (lb8, cb8) per LUT row
  std::string err;
  // RTjpeg_t's header-driven state (lib/RTjpeg.c:3568-3579)
  int width = 0, height = 0, Q = 0;
  // persistent picture of the one-packet path (priv->frame of lib/video_rtjpeg.c:31-35)
  uint8_t* d_frame = nullptr;
  size_t frame_bytes = 0;
  uint8_t* d_pkt = nullptr;
  size_t pkt_cap = 0;
  uint8_t* h_frame = nullptr;     // pinned host picture of the nocopy path
  size_t h_frame_cap = 0;
  mi_rtj_plan* single = nullptr;  // reusable 1-frame plan
  // Asynchronous producers on the instance's stream (mi_rtj_dev_memset) may have written what a plan's index kernels
  // read on their own stream: the next launch of a plan with an index stream makes that stream wait for this event once
  // (not at every launch — that would put the index of launch k + 1 behind the transform of launch k)
  bool input_dirty = false;
  hipEvent_t e_input = nullptr;
};

struct mi_rtj_plan {
  mi_rtj_ctx* ctx = nullptr;
  int n = 0;
  std::vector<FrameDev> h_frames;
  FrameDev* d_frames = nullptr;
  uint32_t* d_blkoff = nullptr;
  uint32_t* d_summary = nullptr;    // [total chunks][kEntries]
  uint16_t* d_lentab = nullptr;     // [total chunks][kChunk] block lengths (luma | chroma << 8)
  uint32_t* d_chunk_pos = nullptr;  // [total chunks + n]
  uint32_t* d_chunk_mb = nullptr;
  uint64_t n_blocks = 0, n_index = 0, bytes_in = 0, bytes_out = 0;
  uint64_t n_chunks = 0, n_chunk_entries = 0;
  uint32_t max_groups = 0, max_chunks = 0;
  bool profile = false;
  bool serial_index = false;        // MI_RTJ_INDEX=serial: one wave per packet (A/B baseline)
  bool emit_walk = false;           // MI_RTJ_EMIT=walk: per-chunk re-walk instead of the length tables
  bool one_block_type = false;      // every frame's tables have lb8 == cb8: single-search summarize
  // speculative index (rtj_spec_kernels.h): one walker per kSpecChunk bytes, proven per packet afterwards
  uint32_t dec_slots = 0;           // MI_RTJ_DEC_SLOTS (A/B): k_decode waves per part, 0 = decode_slots()
  int spec_mode = -1;               // MI_RTJ_SPEC: 0 never, 1 / 3 always with the short / long lead (no policy), 2 whatever the batch size, otherwise by batch size
  bool spec = false;
  uint64_t n_spec = 0, cap_spec = 0;       // walkers of this plan / allocated
  std::vector<SpecChunkDev> h_spec_chunks;
  std::vector<uint32_t> h_spec_base;       // [n + 1]
  SpecChunkDev* d_spec_chunks = nullptr;
  uint32_t* d_spec_base = nullptr;
  uint32_t* d_spec_rec = nullptr;          // the walkers' start bits: spec_bits_words(walkers) dwords, [wave][tile][lane][4]
  uint32_t* d_spec_nrec = nullptr;
  uint32_t* d_spec_wstart = nullptr;       // [walkers]: first byte each walker parsed (repairs move it)
  uint2* d_spec_hand = nullptr;            // [walkers]: where a chunk takes over / where the next one has to
  uint2* d_spec_fix = nullptr;             // [walkers]: (walker, byte to start from) of the chunks to walk again
  uint8_t* d_spec_flag = nullptr;          // [walkers + 64]: 1 = on that list in this launch
  uint32_t* d_spec_nfix = nullptr;
  uint32_t* d_spec_ok = nullptr;           // [n]: 1 = the packet's index is proven
  uint32_t* d_spec_todo = nullptr;         // [n + 2]: count, then the packets left to the exact kernels; [n + 1]: how many
                                           // of them the serial walker takes instead (k_spec_policy)
  int cap_spec_frames = 0;
  uint32_t* d_spec_state = nullptr;        // [2]: launches in a row that refused every packet, launches left paused
  // batches: what k_decode_split leaves to k_decode_list (rtj_decode_kernels.h, DecList): every part of every group of a
  // launch fits; two counters, launches alternate, a launch's k_decode_list zeroes the other one
  uint2* d_declist = nullptr;
  uint32_t* d_declist_cnt = nullptr;
  // the policy's mode word as the host last saw it: pinned host memory that k_decode_list writes behind its books.  The
  // host reads it without waiting (it may be launches old) for ONE decision: while it says "split form" the classic form's
  // kernel is not enqueued at all (it would return at once: ~90 us of empty workgroups per 16,384 pictures) and
  // k_decode_split is told to run whatever the device's word says — so a stale view costs time on content that just
  // turned noisy, never pictures
  uint32_t* h_mode_seen = nullptr;
  uint32_t* d_mode_seen = nullptr;
  uint64_t declist_cap = 0;
  int declist_flip = 0;
  bool batch_seen = false;                 // a batch launch (a wave takes several parts of its groups, no previous picture) was queued
  uint32_t serial_min = 4096;              // MI_RTJ_SERIAL_MIN: to-do lists from this many packets on go to the serial walker (0: never)
  int split = -1;                          // MI_RTJ_SPLIT (A/B): 0 = batches always run k_decode<true, false> (round 3's form),
                                           // 1 = always k_decode_split; otherwise the plan's policy decides on the device
  int rotate = -1;                         // MI_RTJ_ROTATE: 1 / 0 = a k_decode wave takes all three parts of its groups / one part; -1 = by batch size
  const uint8_t* prev_pic = nullptr;       // sessions: where unchanged blocks of this launch are copied from
  hipStream_t idx_stream = nullptr;        // the stream of the index kernels (sessions: theirs, not owned; batches: own_idx); null = the instance's
  hipEvent_t e_idx = nullptr;              // index done (only used with idx_stream)
  // Batches: the index of launch k + 1 is built while launch k is still being transformed.  The only thing the two
  // halves of a launch share is the block-offset index, so there are two of those (launches alternate) and an event per
  // index that says "k_decode has read it" before the launch after next writes it again.  Both halves are bound by
  // vector-instruction issue, so a long launch gains nothing; a short one (1024 pictures: 6 rounds of resident waves
  // for k_decode, under one for the walkers) fills the other half's idle tails.  MI_RTJ_OVERLAP=0 / 1 overrides.
  bool overlap = false;
  // Long launches (which do not overlap by default) of NOISY content do gain: there the index is the serial walker, heavy
  // on the scalar unit, next to a transform heavy on the vector unit (+-32: 332 K against 297 K pictures per second at
  // 16,384 per launch, +-64: 271 K against 251 K, profiles/r04/overlap_at_16384.txt).  The host knows such content by
  // the decode policy's mode word (h_mode_seen: "classic form"), seen without waiting; while it says so the launches of
  // a plan that may (overlap_dyn) run like an overlapped plan's; the second index and the stream are made with the plan.
  bool overlap_dyn = false;
  bool last_overlapped = false;            // the launch before this one ran its index on own_idx
  hipStream_t own_idx = nullptr;
  uint32_t* d_blkoff_b = nullptr;          // the second block-offset index (d_blkoff is the first)
  hipEvent_t e_read[2] = {nullptr, nullptr};
  int flip = 0;                            // which index the NEXT launch writes
  int last = 0;                            // which one the last launch wrote (mi_rtj_plan_read_index)
  std::vector<Timed> ev[MI_RTJ_NUM_KERNELS];  // one pair per launch while profiling
  int launches = 0;
};

// One packet in flight of a pipelined session (mi_rtj_pipe_*): its own pinned staging, device packet, device
// picture, pinned host picture and one-frame plan, so that the copy in of packet i+1, the kernels of packet i and the
// copy out of picture i-1 run side by side on three streams.
struct PipeSlot {
  mi_rtj_plan* plan = nullptr;
  uint8_t* h_stage = nullptr;  // pinned: FrameDev (64 bytes), then the packet
  uint8_t* d_stage = nullptr;
  size_t stage_cap = 0;
  uint8_t* d_pic = nullptr;    // this packet's picture on the device (starts as a copy of its predecessor's)
  uint8_t* h_pic = nullptr;    // pinned host picture handed to the caller
  size_t pic_cap = 0;
  bool owns_pic = true;        // false: d_pic / h_pic point into the session's group buffers (copies out in groups)
  int out_state = 0;           // 0 nothing to copy, 1 decoded and waiting for its group's copy out, 2 copy out queued
  hipEvent_t out_ev = nullptr; // the event that says this picture is in host memory (e_out of the slot that closed the copy)
  hipEvent_t e_in = nullptr, e_dec = nullptr, e_out = nullptr;
  uint64_t tag = 0;
  int w = 0, h = 0;
  FrameDev fd;                      // (index groups) the packet's descriptor, bases filled in when its group is queued
  int staged = 0;                   // (index groups) 1: copied in, kernels not queued yet
  // hand-over to the session's worker thread (the HIP calls of a packet are made there)
  size_t len = 0;                   // packet bytes in h_stage
  const uint8_t* prev = nullptr;    // predecessor's device picture (or none)
  int issued = 0;                   // 1 once the worker has queued everything up to the copy out (guarded by mu)
  int rc = 0;                       // what queuing it returned
  std::string err;
};

struct mi_rtj_pipe {
  mi_rtj_ctx* ctx = nullptr;
  int depth = 0;
  int max_w = 0, max_h = 0;   // the stream's coded size: packets that announce more are refused
  std::vector<PipeSlot> slot;
  int head = 0, count = 0;    // oldest packet in flight, packets in flight
  int lent = -1;              // slot whose host picture the caller is looking at (until the next mi_rtj_pipe_next)
  const uint8_t* prev_pic = nullptr;  // device picture of the packet submitted last (what 0xFF blocks keep)
  size_t prev_bytes = 0;
  // kernels run on the instance's stream; packets come in on s_in; pictures leave on s_out[slot & 1]: with one stream a
  // session has one picture-sized copy out in flight at a time, and the link gives such a copy (3.1 MB at 1080p) 38 GB/s
  // where two side by side, or larger ones, get past 50 (profiles/r02/pcie_probe.json); MI_RTJ_OUT_STREAMS=1 is the A/B
  hipStream_t s_in = nullptr, s_out[2] = {nullptr, nullptr};
  int n_out = 2;
  // Pictures leave in groups: the slots of a group lie side by side on the device and in pinned host memory, and one
  // copy takes all of them once the group's last packet is decoded (or as many as are decoded when the caller asks for
  // one earlier).  The link moves a 1080p picture (3.1 MB) at 38 GB/s and 6-12 MB at 47-55 (profiles/r02/pcie_probe.json),
  // and two copy streams do NOT overlap two small copies (profiles/r03/e2e_what_bounds_a_session.txt).
  // MI_RTJ_OUT_GROUP = 1 / 2 / 4; sessions without a fixed coded size copy picture by picture.
  int group = 1;
  size_t grp_fsz = 0;                  // bytes of one picture in a group buffer
  std::vector<uint8_t*> d_grp, h_grp;  // [depth / group]
  // The INDEX of the packets is built in groups as well (MI_RTJ_IDX_GROUP = 1 / 2 / 4, default: the copy group): the
  // exact index of ONE 1080p packet is three launches that each run one round of workgroups or one serial chain
  // (k_index_summarize 19 us, k_index_resolve 21, k_index_emit 13, tools/one_packet_kernels.py) and take as long for
  // two or four packets side by side.  A packet is copied in at once and waits ("staged") until its group is complete
  // or the caller asks for its picture; then one index launch covers the group and k_decode follows packet by packet
  // (a packet's unchanged blocks come from its predecessor's picture).  One plan per group of slots; the descriptors
  // of all slots live in one array (pinned + device) so that a group's are contiguous.
  int igroup = 1;
  int nstaged = 0;                     // slots staged and not yet queued: the last `nstaged` submitted
  std::vector<mi_rtj_plan*> gplan;     // [depth / igroup]
  FrameDev* h_desc = nullptr;          // [depth], pinned
  FrameDev* d_desc = nullptr;          // [depth]
  hipStream_t s_idx = nullptr;         // MI_RTJ_IDX_STREAM=1 (A/B, slower): index kernels of the packets on a stream of their own
  int out_kernel = 0;  // MI_RTJ_OUT_KERNEL=1: the picture leaves through a copy kernel that stores into the pinned host
                       // picture, instead of the copy engine (A/B)
  int exp_skip = 0;    // MI_RTJ_EXP_SKIP (experiments, wrong pictures): 1 no copy out, 2 no kernels, 4 no copy in
  uint64_t submitted = 0, returned = 0;
  // MI_RTJ_PIPE_STATS=1: where the submitting thread's time goes (seconds; printed to stderr when the session ends):
  // [0] copying packets into pinned staging, [1] runtime calls that queue work, [2] waiting for a picture's copy out
  bool stats = false;
  int wait_mode = 0;
  double t_stage = 0, t_issue = 0, t_wait = 0, t_d2h_call = 0;  // t_d2h_call: inside hipMemcpyAsync of the copy out (part of t_issue)
  // The thread that submits is what bounds a session (about 50 us of runtime calls per packet next to the 30 us it
  // takes to copy a 1080p packet into pinned staging), so the two halves run on two threads: the caller copies, a
  // worker owned by the session makes the HIP calls, in submission order.
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<int> jobs;      // ring of slot numbers waiting for the worker
  size_t job_head = 0, job_count = 0;
  bool stop = false;
  bool threaded = false;      // MI_RTJ_PIPE_THREAD=1: HIP calls on the worker thread (A/B; off by default)
};

namespace {

// Environment switches.  Tuning knobs of the product (launch sizes, group sizes, which index runs) are read with
// getenv; switches that only exist to take measurements — the ones that leave work out (wrong or no pictures) and the
// A/B forms that were measured slower and are kept for the record — are read through exp_env(), which a product build
// compiles to "not set": they exist in builds with -DMIRTJ_EXPERIMENTS only (tools/ builds those next to the product).
inline const char* exp_env(const char* name) {
#ifdef MIRTJ_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

int fail(mi_rtj_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  else {
    std::lock_guard<std::mutex> l(g_err_mu);
    g_create_err = buf;
  }
  return code;
}

#define HIPCHK(c, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail((c), MI_RTJ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                        \
  } while (0)

bool device_is_gfx950(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

// RTjpeg_decompress's header handling (lib/RTjpeg.c:3568-3579) on the instance state.
// Returns the LUT row to use, or a negative error.
int apply_header(mi_rtj_ctx* c, const uint8_t* hdr, int* w, int* h) {
  const int hw = hdr[6] | (hdr[7] << 8), hh = hdr[8] | (hdr[9] << 8), q = hdr[10];
  if (hw <= 0 || hh <= 0 || (hw & 15) || (hh & 15))
    return fail(c, MI_RTJ_ERR_GEOMETRY, "packet header %dx%d: width and height must be positive multiples of 16", hw, hh);
  c->width = hw;
  c->height = hh;
  if (q != c->Q) c->Q = q < 1 ? 1 : q;  // RTjpeg_set_quality clamps to 1..255 (lib/RTjpeg.c:2410-2411)
  *w = hw;
  *h = hh;
  return c->Q;  // 0 only while no non-zero quality was ever seen: the all-zero tables
}

int fill_frame(mi_rtj_ctx* c, const uint8_t* hdr, uint64_t pkt_off, uint32_t pkt_len, uint64_t out_off,
               uint32_t blk_base, FrameDev* f) {
  int w, h;
  const int q = apply_header(c, hdr, &w, &h);
  if (q < 0) return q;
  if (out_off & 15) return fail(c, MI_RTJ_ERR_ARG, "out_offset %llu is not a multiple of 16", (unsigned long long)out_off);
  memset(f, 0, sizeof(*f));
  f->data_off = pkt_off + MI_RTJ_HEADER_SIZE;
  f->out_off = out_off;
  if (pkt_len >= 0x80000000u) return fail(c, MI_RTJ_ERR_ARG, "packet of %u bytes: packets are limited to 2 GiB", pkt_len);
  f->data_len = pkt_len > MI_RTJ_HEADER_SIZE ? pkt_len - MI_RTJ_HEADER_SIZE : 0;
  f->w = (uint32_t)w;
  f->h = (uint32_t)h;
  f->qidx = (uint32_t)q;
  f->blk_base = blk_base;
  f->mbw = (uint32_t)w / 16;
  f->nmb = f->mbw * ((uint32_t)h / 16);
  if ((uint64_t)f->nmb * 6 * 64 + kAllocPad >= 0xFFFFFFFFull)  // block offsets and the walkers' positions are 32-bit
    return fail(c, MI_RTJ_ERR_GEOMETRY, "picture of %dx%d: its worst-case stream does not fit 32-bit offsets", w, h);
  f->nchunks = (f->data_len + kChunk - 1) / kChunk;
  if (f->nchunks == 0) f->nchunks = 1;
  return MI_RTJ_OK;
}

// per-chunk scratch of a plan (re)sized for its frames; chunk_base / sum_base are set here
int plan_alloc_chunks(mi_rtj_plan* p) {
  mi_rtj_ctx* c = p->ctx;
  uint64_t chunks = 0, entries = 0;
  p->max_chunks = 0;
  p->one_block_type = true;
  for (auto& f : p->h_frames) {
    p->one_block_type = p->one_block_type && c->b8[f.qidx][0] == c->b8[f.qidx][1];
    f.sum_base = (uint32_t)chunks;
    f.chunk_base = (uint32_t)entries;
    chunks += f.nchunks;
    entries += f.nchunks + 1;
    if (f.nchunks > p->max_chunks) p->max_chunks = f.nchunks;
  }
  if (chunks > p->n_chunks) {
    if (p->d_summary) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(p->d_summary);
      (void)hipFree(p->d_lentab);
      p->d_summary = nullptr;
      p->d_lentab = nullptr;
      p->n_chunks = 0;
    }
    HIPCHK(c, hipMalloc((void**)&p->d_summary, sizeof(uint32_t) * kEntries * chunks));
    HIPCHK(c, hipMalloc((void**)&p->d_lentab, sizeof(uint16_t) * kChunk * chunks + 64));
    p->n_chunks = chunks;
  }
  if (entries > p->n_chunk_entries) {
    if (p->d_chunk_pos) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(p->d_chunk_pos);
      (void)hipFree(p->d_chunk_mb);
      p->d_chunk_pos = p->d_chunk_mb = nullptr;
      p->n_chunk_entries = 0;
    }
    HIPCHK(c, hipMalloc((void**)&p->d_chunk_pos, sizeof(uint32_t) * entries));
    HIPCHK(c, hipMalloc((void**)&p->d_chunk_mb, sizeof(uint32_t) * entries));
    p->n_chunk_entries = entries;
  }
  // ---- speculative index: worth it once the batch fills the device (a walker is one lane and runs
  //      for ~0.25 ms whatever the batch; a single packet is indexed faster by the exact kernels) ----
  uint64_t walkers = 0;
  // one walker per kSpecChunk bytes AND one whose chunk begins at or past the packet's last byte: the index's last entry,
  // the end position, is the start of a block that is not there, and a packet whose length is a whole number of chunks
  // has it at the first byte of a chunk of its own (12 of the bench's 16,384 packets were refused for the lack of it)
  auto spec_chunks_of = [](const FrameDev& f) -> uint32_t { return f.data_len / (uint32_t)kSpecChunk + 1u; };
  for (auto& f : p->h_frames) walkers += spec_chunks_of(f);
  p->spec = !p->serial_index && !p->emit_walk && (p->spec_mode >= 1 || (p->spec_mode != 0 && walkers >= kSpecMinWalkers));
  if (p->spec) {
    p->h_spec_chunks.clear();
    p->h_spec_base.assign(1, 0u);
    for (size_t i = 0; i < p->h_frames.size(); i++) {
      const FrameDev& f = p->h_frames[i];
      const uint32_t nsc = spec_chunks_of(f);
      for (uint32_t k = 0; k < nsc; k++) p->h_spec_chunks.push_back(SpecChunkDev{(uint32_t)i, k});
      p->h_spec_base.push_back((uint32_t)p->h_spec_chunks.size());
    }
    p->n_spec = p->h_spec_chunks.size();
    if (p->n_spec >= 0x7FFFFFFFull) p->spec = false;  // walker numbers are 32-bit
  }
  if (p->spec) {
    if (p->n_spec > p->cap_spec) {
      if (p->d_spec_chunks) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(p->d_spec_chunks);
        (void)hipFree(p->d_spec_rec);
        (void)hipFree(p->d_spec_nrec);
        (void)hipFree(p->d_spec_wstart);
        (void)hipFree(p->d_spec_hand);
        (void)hipFree(p->d_spec_fix);
        (void)hipFree(p->d_spec_flag);
        p->d_spec_chunks = nullptr;
        p->d_spec_rec = nullptr;
        p->d_spec_nrec = p->d_spec_wstart = nullptr;
        p->d_spec_hand = nullptr;
        p->d_spec_fix = nullptr;
        p->d_spec_flag = nullptr;
        p->cap_spec = 0;  // a failed hipMalloc below must not leave freed pointers behind
      }
      HIPCHK(c, hipMalloc((void**)&p->d_spec_chunks, sizeof(SpecChunkDev) * p->n_spec));
      HIPCHK(c, hipMalloc((void**)&p->d_spec_rec, sizeof(uint32_t) * spec_bits_words(p->n_spec)));  // whole waves of walkers, + a spare walker for idle lanes
      HIPCHK(c, hipMalloc((void**)&p->d_spec_nrec, sizeof(uint32_t) * p->n_spec));
      HIPCHK(c, hipMalloc((void**)&p->d_spec_wstart, sizeof(uint32_t) * p->n_spec));
      HIPCHK(c, hipMalloc((void**)&p->d_spec_hand, sizeof(uint2) * p->n_spec));
      HIPCHK(c, hipMalloc((void**)&p->d_spec_fix, sizeof(uint2) * p->n_spec));
      HIPCHK(c, hipMalloc((void**)&p->d_spec_flag, p->n_spec + 64));
      HIPCHK(c, hipMemsetAsync(p->d_spec_flag, 0, p->n_spec + 64, c->stream));  // (the walkers zero [0, n_spec) per launch)
      if (!p->d_spec_nfix) {
        HIPCHK(c, hipMalloc((void**)&p->d_spec_nfix, sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(p->d_spec_nfix, 0, sizeof(uint32_t), c->stream));
      }
      p->cap_spec = p->n_spec;
    }
    if ((int)p->h_frames.size() > p->cap_spec_frames) {
      if (p->d_spec_base) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(p->d_spec_base);
        (void)hipFree(p->d_spec_ok);
        (void)hipFree(p->d_spec_todo);
        p->d_spec_base = p->d_spec_ok = p->d_spec_todo = nullptr;
        p->cap_spec_frames = 0;
      }
      HIPCHK(c, hipMalloc((void**)&p->d_spec_base, sizeof(uint32_t) * (p->h_frames.size() + 1)));
      HIPCHK(c, hipMalloc((void**)&p->d_spec_ok, sizeof(uint32_t) * p->h_frames.size()));
      HIPCHK(c, hipMemsetAsync(p->d_spec_ok, 0, sizeof(uint32_t) * p->h_frames.size(), c->stream));  // "nothing proven yet" for mi_rtj_plan_spec_stats
      HIPCHK(c, hipMalloc((void**)&p->d_spec_todo, sizeof(uint32_t) * (p->h_frames.size() + 2)));
      HIPCHK(c, hipMemsetAsync(p->d_spec_todo + p->h_frames.size() + 1, 0, sizeof(uint32_t), c->stream));
      if (!p->d_spec_state) {
        HIPCHK(c, hipMalloc((void**)&p->d_spec_state, sizeof(uint32_t) * kSpecStWords));
        HIPCHK(c, hipMemsetAsync(p->d_spec_state, 0, sizeof(uint32_t) * kSpecStWords, c->stream));
      }
      p->cap_spec_frames = (int)p->h_frames.size();
    }
    HIPCHK(c, hipMemcpyAsync(p->d_spec_chunks, p->h_spec_chunks.data(), sizeof(SpecChunkDev) * p->n_spec, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p->d_spec_base, p->h_spec_base.data(), sizeof(uint32_t) * p->h_spec_base.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the host vectors may be rebuilt by the next call
  }
  return MI_RTJ_OK;
}

int plan_upload(mi_rtj_plan* p) {
  mi_rtj_ctx* c = p->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  if (!p->d_frames) HIPCHK(c, hipMalloc((void**)&p->d_frames, sizeof(FrameDev) * p->n));
  HIPCHK(c, hipMemcpyAsync(p->d_frames, p->h_frames.data(), sizeof(FrameDev) * p->n, hipMemcpyHostToDevice, c->stream));
  return MI_RTJ_OK;
}

// what: the index kernels (kLaunchIndex), k_decode (kLaunchDecode) or both.  A session whose packets are indexed in
// groups queues the index of a group once and then k_decode packet by packet (dframe = the packet's place in the group,
// dcount = 1, d_out = its picture): a packet's unchanged blocks come from its predecessor's picture, so the pictures of
// a group cannot be made by one launch.
enum { kLaunchIndex = 1, kLaunchDecode = 2, kLaunchAll = 3 };
int plan_launch(mi_rtj_plan* p, const void* d_stream, void* d_out, int what = kLaunchAll, uint32_t dframe = 0,
                int dcount = -1) {
  mi_rtj_ctx* c = p->ctx;
  const uint8_t* st = (const uint8_t*)d_stream;
  // The index kernels run on `is`, k_decode on the instance's stream.  Plans have them equal; a session gives its slots
  // an index stream of their own (p->idx_stream), so that the index of packet i + 1 is built while packet i — whose
  // picture the unchanged blocks of i + 1 come from — is still being transformed.
  bool ov = p->overlap;
  if (!ov && p->overlap_dyn && what == kLaunchAll && p->h_mode_seen &&
      *(volatile uint32_t*)p->h_mode_seen == (uint32_t)kDecModeClassic) {
    ov = true;  // (the second index and the stream were made with the plan: nothing is allocated on the launch path)
  }
  hipStream_t const ds = c->stream, is = ov && !p->idx_stream ? p->own_idx : p->idx_stream ? p->idx_stream : c->stream;
  hipStream_t cur = is;
  // (a launch that overlaps behind one that did not: whatever the instance's stream still has queued — the last launch's
  // transform reading the first index, the plan's descriptors on their way — comes first)
  if (is != ds && (c->input_dirty || (ov && !p->overlap && !p->last_overlapped))) {  // something queued on the instance's stream may still be writing the packets
    if (!c->e_input) HIPCHK(c, hipEventCreateWithFlags(&c->e_input, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->e_input, ds));
    HIPCHK(c, hipStreamWaitEvent(is, c->e_input, 0));
  }
  c->input_dirty = false;
  uint32_t* const blk = ov && p->flip ? p->d_blkoff_b : p->d_blkoff;  // the index this launch writes and reads
  if (ov) {
    // the launch before last read this index: its k_decode must be through with it
    if (p->e_read[p->flip]) HIPCHK(c, hipStreamWaitEvent(is, p->e_read[p->flip], 0));
    else HIPCHK(c, hipEventCreateWithFlags(&p->e_read[p->flip], hipEventDisableTiming));
  }
  Timed t[MI_RTJ_NUM_KERNELS];
  // while profiling: one event after every kernel; a kernel's start is its predecessor's end (they run back
  // to back on one stream), so a launch costs one event record per kernel instead of two
  hipEvent_t prev = nullptr;
  auto begin = [&](int k) -> int {
    if (!p->profile) return MI_RTJ_OK;
    if (prev) {
      t[k].a = prev;
      t[k].owns_a = false;
      return MI_RTJ_OK;
    }
    HIPCHK(c, hipEventCreate(&t[k].a));
    HIPCHK(c, hipEventRecord(t[k].a, cur));
    return MI_RTJ_OK;
  };
  auto end = [&](int k) -> int {
    if (!p->profile) return MI_RTJ_OK;
    HIPCHK(c, hipEventCreate(&t[k].b));
    HIPCHK(c, hipEventRecord(t[k].b, cur));
    p->ev[k].push_back(t[k]);
    prev = t[k].b;
    return MI_RTJ_OK;
  };
  int rc;
  if (what & kLaunchIndex) {
    if (p->serial_index) {
      if ((rc = begin(MI_RTJ_K_EMIT)) != MI_RTJ_OK) return rc;
      hipLaunchKernelGGL(k_index_walk, dim3(p->n), dim3(64), 0, is, p->d_frames, st, c->d_lut, blk);
      if ((rc = end(MI_RTJ_K_EMIT)) != MI_RTJ_OK) return rc;
    } else {
      const uint32_t *todo = nullptr, *ntodo = nullptr;
      unsigned rows = (unsigned)p->n;  // grid rows of the exact kernels: one per packet, or a few that loop over the to-do list
      const bool spec = p->spec;
      // noisy content defeats the speculation; a plan that sees every packet refused twice in a row goes
      // without it for kSpecPauseLaunches launches.  The policy lives on the device (k_spec_policy), so it
      // also works when launches are queued faster than they run.
      const bool no_policy = p->spec_mode == 1 || p->spec_mode == 3 || p->spec_mode == 4;  // always speculate: short (1), long (3) or very long (4) lead
      uint32_t* const state = no_policy ? nullptr : p->d_spec_state;
      if (spec) {
        ntodo = p->d_spec_todo;
        todo = p->d_spec_todo + 1;
        rows = std::min<unsigned>(rows, kSpecFallbackRows);
        if ((rc = begin(MI_RTJ_K_SPEC_WALK)) != MI_RTJ_OK) return rc;
        // The walkers zero the launch's lists themselves (SpecReset).  With a policy ONE dispatch runs the form the
        // policy's state names (or returns at once while the speculation is paused); without (MI_RTJ_SPEC = 1 / 3 / 4:
        // tests) the form asked for.
        const dim3 wgrid((unsigned)((p->n_spec + 63) / 64));
        const SpecReset rs{p->d_spec_nfix, p->d_spec_todo, p->d_spec_flag};
#define MIRTJ_WALK_ARGS p->d_frames, p->d_spec_chunks, (uint32_t)p->n_spec, st, c->d_lut, p->d_spec_rec, p->d_spec_nrec, p->d_spec_wstart, p->d_spec_hand, rs
        if (state) {
          if (p->one_block_type) hipLaunchKernelGGL((k_spec_walk_any<false>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS, state);
          else hipLaunchKernelGGL((k_spec_walk_any<true>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS, state);
        } else if (p->one_block_type) {
          if (p->spec_mode == 1) hipLaunchKernelGGL((k_spec_walk<false, kSpecLead>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS);
          if (p->spec_mode == 3) hipLaunchKernelGGL((k_spec_walk<false, kSpecLeadLong>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS);
          if (p->spec_mode == 4) hipLaunchKernelGGL((k_spec_walk<false, kSpecLeadVery>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS);
        } else {
          if (p->spec_mode == 1) hipLaunchKernelGGL((k_spec_walk<true, kSpecLead>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS);
          if (p->spec_mode == 3) hipLaunchKernelGGL((k_spec_walk<true, kSpecLeadLong>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS);
          if (p->spec_mode == 4) hipLaunchKernelGGL((k_spec_walk<true, kSpecLeadVery>), wgrid, dim3(64), 0, is, MIRTJ_WALK_ARGS);
        }
#undef MIRTJ_WALK_ARGS
        if ((rc = end(MI_RTJ_K_SPEC_WALK)) != MI_RTJ_OK) return rc;
        if ((rc = begin(MI_RTJ_K_SPEC_VERIFY)) != MI_RTJ_OK) return rc;
        // first pass; walkers that had not fallen into step are walked again from a known block start; second
        // pass over the packets concerned (both return at once when there is nothing to repair)
        hipLaunchKernelGGL(k_spec_verify, dim3(p->n), dim3(kSpecVerThreads), 0, is, p->d_frames, p->d_spec_base, c->d_lut,
                           p->d_spec_rec, p->d_spec_nrec, blk, p->d_spec_ok, p->d_spec_todo + 1, p->d_spec_todo, state,
                           p->d_spec_wstart, p->d_spec_hand, p->d_spec_fix, p->d_spec_nfix, p->d_spec_flag, 1);
        hipLaunchKernelGGL(k_spec_repair, dim3(kSpecRepairGrid), dim3(64), 0, is, p->d_frames, p->d_spec_chunks, st,
                           c->d_lut, p->d_spec_rec, p->d_spec_nrec, p->d_spec_wstart, p->d_spec_hand, p->d_spec_fix, p->d_spec_nfix,
                           p->d_spec_flag, (uint32_t)p->n_spec, state);
        hipLaunchKernelGGL(k_spec_verify, dim3(p->n), dim3(kSpecVerThreads), 0, is, p->d_frames, p->d_spec_base, c->d_lut,
                           p->d_spec_rec, p->d_spec_nrec, blk, p->d_spec_ok, p->d_spec_todo + 1, p->d_spec_todo, state,
                           p->d_spec_wstart, p->d_spec_hand, p->d_spec_fix, p->d_spec_nfix, p->d_spec_flag, 2);
        if ((rc = end(MI_RTJ_K_SPEC_VERIFY)) != MI_RTJ_OK) return rc;
        if (state) hipLaunchKernelGGL(k_spec_policy, dim3(1), dim3(256), 0, is, (uint32_t)p->n, (uint32_t)p->n_spec, p->d_spec_todo, p->d_spec_nfix, state,
                                      p->serial_min);
      }
      if ((rc = begin(MI_RTJ_K_SUMMARIZE)) != MI_RTJ_OK) return rc;
      // A long to-do list (the speculation paused on noisy content, a big launch) goes to the serial walker, one wave
      // per packet: with thousands of packets in flight it indexes faster than the chunk-parallel kernels, whose work
      // grows with the stream's bytes (profiles/r04/noisy_serial_ab.txt: +15 % at 4096 packets of +-64 noise, +28 % at
      // 16384).  k_spec_policy moved the count to the walker's word; the exact kernels then see an empty list.
      if (todo && state)
        hipLaunchKernelGGL(k_index_walk_todo, dim3(std::min<unsigned>((unsigned)p->n, 16384u)), dim3(64), 0, is, p->d_frames,
                           st, c->d_lut, blk, todo, p->d_spec_todo + 1 + p->n);
      if (todo) {
        if (p->one_block_type)
          hipLaunchKernelGGL(k_index_summarize_todo<1>, dim3(p->max_chunks, rows), dim3(kSumThreads), 0, is,
                             p->d_frames, st, c->d_lut, p->d_summary, p->d_lentab, todo, ntodo);
        else
          hipLaunchKernelGGL(k_index_summarize_todo<2>, dim3(p->max_chunks, rows), dim3(kSumThreads), 0, is,
                             p->d_frames, st, c->d_lut, p->d_summary, p->d_lentab, todo, ntodo);
      } else if (p->one_block_type) {
        hipLaunchKernelGGL(k_index_summarize<1>, dim3(p->max_chunks, rows), dim3(kSumThreads), 0, is, p->d_frames,
                           st, c->d_lut, p->d_summary, p->d_lentab);
      } else {
        hipLaunchKernelGGL(k_index_summarize<2>, dim3(p->max_chunks, rows), dim3(kSumThreads), 0, is, p->d_frames,
                           st, c->d_lut, p->d_summary, p->d_lentab);
      }
      if ((rc = end(MI_RTJ_K_SUMMARIZE)) != MI_RTJ_OK) return rc;
      if ((rc = begin(MI_RTJ_K_RESOLVE)) != MI_RTJ_OK) return rc;
      hipLaunchKernelGGL(k_index_resolve, dim3(todo ? std::min<unsigned>((unsigned)p->n, 1024u) : rows), dim3(256), 0, is, p->d_frames, p->d_summary,
                         p->d_chunk_pos, p->d_chunk_mb, todo, ntodo);
      if ((rc = end(MI_RTJ_K_RESOLVE)) != MI_RTJ_OK) return rc;
      if ((rc = begin(MI_RTJ_K_EMIT)) != MI_RTJ_OK) return rc;
      if (p->emit_walk)
        hipLaunchKernelGGL(k_index_emit_walk, dim3(p->max_chunks, p->n), dim3(64), 0, is, p->d_frames, st,
                           c->d_lut, p->d_chunk_pos, p->d_chunk_mb, blk);
      else
        hipLaunchKernelGGL(k_index_emit, dim3(p->max_chunks, rows), dim3(kEmitThreads), 0, is, p->d_frames,
                           p->d_lentab, p->d_chunk_pos, p->d_chunk_mb, blk, todo, ntodo);
      if ((rc = end(MI_RTJ_K_EMIT)) != MI_RTJ_OK) return rc;
    }
  }
  if (is != ds) {  // k_decode waits for the index (and, through it, for the packet's copy in)
    if (!p->e_idx) HIPCHK(c, hipEventCreateWithFlags(&p->e_idx, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(p->e_idx, is));
    HIPCHK(c, hipStreamWaitEvent(ds, p->e_idx, 0));
    prev = nullptr;  // (profiling) k_decode's start is not the end of a kernel on another stream
  }
  cur = ds;
  if (!(what & kLaunchDecode)) {
    HIPCHK(c, hipGetLastError());
    return MI_RTJ_OK;
  }
  if (what != kLaunchAll) prev = nullptr;  // (profiling) k_decode's start is not the end of the kernel queued before it
  const unsigned drows = dcount < 0 ? (unsigned)p->n : (unsigned)dcount;
  const FrameDev* const dfr = p->d_frames + dframe;
  if ((rc = begin(MI_RTJ_K_DECODE)) != MI_RTJ_OK) return rc;
  // the A/B override is honoured only where it still covers every group
  // a wave per (slot, part) or, in batches that still make enough waves that way, per slot: the wave then takes the
  // three parts of each of its groups in turn and the group's stream bytes cross the fabric once (kernel header)
  const uint32_t span = p->rotate >= 0 ? (p->rotate ? 3u : 1u)
                        : (uint64_t)drows * p->max_groups >= (uint64_t)kDecRotateMinGroups ? 3u : 1u;
  const uint32_t dslots = p->dec_slots && p->dec_slots * (uint32_t)kDecIters >= p->max_groups
                              ? p->dec_slots
                              : decode_slots(p->max_groups, (uint32_t)drows, span);
  {
    const dim3 grid(span == 3u ? dslots : dslots * 3u, drows), block(kDecThreads);
    uint8_t* const out8 = (uint8_t*)d_out;
    // four instantiations: (a wave takes all three parts of its groups | one part) x (unchanged blocks stay | are
    // fetched from the previous packet's picture); a launch runs the one that carries nothing else
    if (span == 3u && !p->prev_pic) p->batch_seen = true;
    if (span == 3u && !p->prev_pic && p->split != 0) {
      // batches: luma waves (two parts of their groups in turn) and chroma waves that pool three groups' busy blocks
      // into one transform round; what neither covers goes to the list and k_decode_list (rtj_decode_chroma.h)
      const uint64_t need = 3ull * drows * p->max_groups;
      if (p->declist_cap < need) {
        HIPCHK(c, hipStreamSynchronize(ds));
        if (p->d_declist) (void)hipFree(p->d_declist);
        p->d_declist = nullptr;
        p->declist_cap = 0;
        if (need > 0xFFFFFFFFull) return fail(c, MI_RTJ_ERR_ARG, "plan too large: %llu group parts", (unsigned long long)need);
        HIPCHK(c, hipMalloc((void**)&p->d_declist, sizeof(uint2) * need));
        p->declist_cap = need;
      }
      if (!p->d_declist_cnt) {  // two list counters, then the policy's two words (mode, launches left in it)
        HIPCHK(c, hipMalloc((void**)&p->d_declist_cnt, 4 * sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(p->d_declist_cnt, 0, 4 * sizeof(uint32_t), ds));
        if (hipHostMalloc((void**)&p->h_mode_seen, sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&p->d_mode_seen, p->h_mode_seen, 0) == hipSuccess) {
          *p->h_mode_seen = 0xFFFFFFFFu;  // nothing seen yet: both forms are enqueued
        } else {  // (no mapped host memory: both forms are enqueued on every launch, as before)
          (void)hipGetLastError();
          if (p->h_mode_seen) (void)hipHostFree(p->h_mode_seen);
          p->h_mode_seen = p->d_mode_seen = nullptr;
        }
      }
      uint32_t* const policy = p->split < 0 ? p->d_declist_cnt + 2 : nullptr;  // (forced split: no policy)
      const bool split_seen = policy && p->h_mode_seen && *(volatile uint32_t*)p->h_mode_seen == (uint32_t)kDecModeSplit;
      const DecList list{p->d_declist_cnt + p->declist_flip, p->d_declist, (uint32_t)p->declist_cap};
      uint32_t lw = split_luma_waves(p->max_groups);
      uint32_t cw = split_chroma_waves(p->max_groups);
      if (const char* e = exp_env("MI_RTJ_LUMA_WAVES")) lw = (uint32_t)atoi(e) ? (uint32_t)atoi(e) : lw;
      if (const char* e = exp_env("MI_RTJ_CHROMA_WAVES")) cw = (uint32_t)atoi(e) ? (uint32_t)atoi(e) : cw;
      // (timing builds only, wrong pictures: MI_RTJ_SPLIT_ONLY = 1 the luma waves alone, 2 the chroma waves alone)
      uint32_t xrot = kSplitXcdRot;
      if (const char* e = exp_env("MI_RTJ_XCD_ROT")) xrot = (uint32_t)atoi(e);
      if (const char* only = exp_env("MI_RTJ_SPLIT_ONLY")) {
        if (atoi(only) == 1)
          hipLaunchKernelGGL(k_decode_split, dim3(kXcds * lw, drows), block, 0, ds, dfr, st, c->d_lut, blk, out8, lw, 0u, list,
                             (const uint32_t*)nullptr, xrot);
        else
          hipLaunchKernelGGL(k_decode_split, dim3(kXcds * cw, drows), block, 0, ds, dfr, st, c->d_lut, blk, out8, 0u, cw, list,
                             (const uint32_t*)nullptr, xrot);
      } else
      hipLaunchKernelGGL(k_decode_split, dim3(kXcds * (lw + cw), drows), block, 0, ds, dfr, st, c->d_lut, blk, out8, lw, cw,
                         list, split_seen ? (const uint32_t*)nullptr : (const uint32_t*)policy, xrot);
      // the classic form: runs while the policy says so (returns at once otherwise), and is not enqueued while the host
      // has seen the policy in its split mode (h_mode_seen above)
      if (policy && !split_seen)
        hipLaunchKernelGGL((k_decode<true, false>), grid, block, 0, ds, dfr, st, c->d_lut, blk, out8,
                           (const uint8_t*)nullptr, (const uint32_t*)policy);
      hipLaunchKernelGGL(k_decode_list, dim3(kDecListGrid), block, 0, ds, dfr, st, c->d_lut, blk, out8, list,
                         p->d_declist_cnt + (p->declist_flip ^ 1), policy, (uint32_t)need, policy ? p->d_mode_seen : nullptr);
      p->declist_flip ^= 1;
    } else if (span == 3u) {
      if (p->prev_pic)
        hipLaunchKernelGGL((k_decode<true, true>), grid, block, 0, ds, dfr, st, c->d_lut, blk, out8,
                           p->prev_pic, (const uint32_t*)nullptr);
      else
        hipLaunchKernelGGL((k_decode<true, false>), grid, block, 0, ds, dfr, st, c->d_lut, blk, out8,
                           (const uint8_t*)nullptr, (const uint32_t*)nullptr);
    } else {
      if (p->prev_pic)
        hipLaunchKernelGGL((k_decode<false, true>), grid, block, 0, ds, dfr, st, c->d_lut, blk, out8,
                           p->prev_pic, (const uint32_t*)nullptr);
      else
        hipLaunchKernelGGL((k_decode<false, false>), grid, block, 0, ds, dfr, st, c->d_lut, blk,
                           out8, (const uint8_t*)nullptr, (const uint32_t*)nullptr);
    }
  }
  if ((rc = end(MI_RTJ_K_DECODE)) != MI_RTJ_OK) return rc;
  HIPCHK(c, hipGetLastError());
  if (ov) {
    HIPCHK(c, hipEventRecord(p->e_read[p->flip], ds));
    p->last = p->flip;
    p->flip ^= 1;
  } else {
    p->last = 0;
  }
  if (what == kLaunchAll) p->last_overlapped = ov && is != ds;
  p->launches++;
  return MI_RTJ_OK;
}

void drop_events(std::vector<Timed>& v) {
  for (auto& t : v) {
    if (t.a && t.owns_a) (void)hipEventDestroy(t.a);
    if (t.b) (void)hipEventDestroy(t.b);
  }
  v.clear();
}

}  // namespace

extern "C" {

int mi_rtj_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; i++) ok += device_is_gfx950(i) ? 1 : 0;
  return ok;
}

mi_rtj_ctx* mi_rtj_create(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    fail(nullptr, MI_RTJ_ERR_NO_DEVICE, "no HIP device: %s — this library has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= n) {
    fail(nullptr, MI_RTJ_ERR_ARG, "device %d out of range (%d devices)", device, n);
    return nullptr;
  }
  if (!device_is_gfx950(device)) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, device);
    fail(nullptr, MI_RTJ_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device, p.gcnArchName);
    return nullptr;
  }
  mi_rtj_ctx* c = new mi_rtj_ctx();
  c->device = device;
  auto bail = [&](const char* what, hipError_t err) -> mi_rtj_ctx* {
    fail(nullptr, MI_RTJ_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
    mi_rtj_destroy(c);
    return nullptr;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  std::vector<QTab> lut(kNumQTab);
  build_all_qtabs(lut.data());
  for (int q = 0; q < kNumQTab; q++) {
    c->b8[q][0] = (uint8_t)lut[q].lb8;
    c->b8[q][1] = (uint8_t)lut[q].cb8;
    // k_decode places DC and the raw bytes inside a block's first 16 bytes (the formula yields 0, 4, 8, 9)
    if (lut[q].lb8 > kMaxRawBytes || lut[q].cb8 > kMaxRawBytes) {
      fail(nullptr, MI_RTJ_ERR_ARG, "quantiser table %d has more than %d raw coefficients", q, kMaxRawBytes);
      mi_rtj_destroy(c);
      return nullptr;
    }
  }
  if ((e = hipMalloc((void**)&c->d_lut, sizeof(QTab) * kNumQTab)) != hipSuccess) return bail("hipMalloc(lut)", e);
  if ((e = hipMemcpy(c->d_lut, lut.data(), sizeof(QTab) * kNumQTab, hipMemcpyHostToDevice)) != hipSuccess)
    return bail("hipMemcpy(lut)", e);
  return c;
}

void mi_rtj_destroy(mi_rtj_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->single) mi_rtj_plan_destroy(c->single);
  if (c->d_frame) (void)hipFree(c->d_frame);
  if (c->d_pkt) (void)hipFree(c->d_pkt);
  if (c->h_frame) (void)hipHostFree(c->h_frame);
  if (c->d_lut) (void)hipFree(c->d_lut);
  if (c->e_input) (void)hipEventDestroy(c->e_input);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* mi_rtj_last_error(const mi_rtj_ctx* c) {
  if (c) return c->err.c_str();
  std::lock_guard<std::mutex> l(g_err_mu);
  static thread_local std::string copy;
  copy = g_create_err;
  return copy.c_str();
}

void mi_rtj_get_state(const mi_rtj_ctx* c, int* w, int* h, int* q) {
  if (!c) return;
  if (w) *w = c->width;
  if (h) *h = c->height;
  if (q) *q = c->Q;
}

void* mi_rtj_dev_alloc(mi_rtj_ctx* c, size_t bytes) {
  if (!c) return nullptr;
  void* p = nullptr;
  if (hipSetDevice(c->device) != hipSuccess) return nullptr;
  hipError_t e = hipMalloc(&p, bytes + kAllocPad);
  if (e != hipSuccess) {
    fail(c, MI_RTJ_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  // the tail pad is read (never used) by dword-granular staging; keep it defined
  (void)hipMemsetAsync((uint8_t*)p + bytes, 0, kAllocPad, c->stream);
  return p;
}

void mi_rtj_dev_free(mi_rtj_ctx* c, void* d) {
  if (!c || !d) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(d);
}

int mi_rtj_h2d(mi_rtj_ctx* c, void* d, const void* s, size_t n) {
  if (!c || (!d && n) || (!s && n)) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_h2d: NULL argument");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MI_RTJ_OK;
}

int mi_rtj_d2h(mi_rtj_ctx* c, void* dst, const void* d, size_t n) {
  if (!c || (!d && n) || (!dst && n)) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_d2h: NULL argument");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(dst, d, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MI_RTJ_OK;
}

int mi_rtj_dev_memset(mi_rtj_ctx* c, void* d, int v, size_t n) {
  if (!c || (!d && n)) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_dev_memset: NULL argument");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemsetAsync(d, v, n, c->stream));
  c->input_dirty = true;  // (plans whose index kernels run on another stream wait for it: plan_launch)
  return MI_RTJ_OK;
}

int mi_rtj_sync(mi_rtj_ctx* c) {
  if (!c) return MI_RTJ_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MI_RTJ_OK;
}

mi_rtj_plan* mi_rtj_plan_create(mi_rtj_ctx* c, int n, const uint8_t* headers, const uint64_t* pkt_offset,
                                const uint32_t* pkt_len, const uint64_t* out_offset) {
  if (!c || n <= 0 || !headers || !pkt_offset || !pkt_len || !out_offset) {
    fail(c, MI_RTJ_ERR_ARG, "mi_rtj_plan_create: bad argument");
    return nullptr;
  }
  if (n > kMaxPlanFrames) {
    fail(c, MI_RTJ_ERR_ARG, "mi_rtj_plan_create: %d packets; a plan takes at most %d (split the batch)", n, kMaxPlanFrames);
    return nullptr;
  }
  mi_rtj_plan* p = new mi_rtj_plan();
  p->ctx = c;
  p->n = n;
  p->h_frames.resize(n);
  uint64_t blk_base = 0;
  for (int i = 0; i < n; i++) {
    if (blk_base > 0xFFFFFFFFull - (1u << 24)) {
      fail(c, MI_RTJ_ERR_ARG, "plan too large: block index exceeds 32 bits");
      delete p;
      return nullptr;
    }
    if (fill_frame(c, headers + (size_t)i * MI_RTJ_HEADER_SIZE, pkt_offset[i], pkt_len[i], out_offset[i],
                   (uint32_t)blk_base, &p->h_frames[i]) != MI_RTJ_OK) {
      delete p;
      return nullptr;
    }
    const FrameDev& f = p->h_frames[i];
    const uint64_t nblk = (uint64_t)f.nmb * 6;
    p->n_blocks += nblk;
    blk_base += (nblk + 1 + 63) & ~63ull;  // keep every frame's index 256-byte aligned
    p->bytes_in += pkt_len[i];
    p->bytes_out += (uint64_t)f.w * f.h * 3 / 2;
    const uint32_t groups = (f.nmb + kMbPerGroup - 1) / kMbPerGroup;
    if (groups > p->max_groups) p->max_groups = groups;
  }
  p->n_index = blk_base;
  {
    const char* mode = getenv("MI_RTJ_INDEX");
    p->serial_index = mode && strcmp(mode, "serial") == 0;
    const char* em = getenv("MI_RTJ_EMIT");
    p->emit_walk = em && strcmp(em, "walk") == 0;
    const char* sp = getenv("MI_RTJ_SPEC");
    p->spec_mode = sp ? atoi(sp) : -1;
    const char* ds = getenv("MI_RTJ_DEC_SLOTS");
    p->dec_slots = ds ? (uint32_t)atoi(ds) : 0u;
    const char* ro = getenv("MI_RTJ_ROTATE");
    p->rotate = ro ? (atoi(ro) != 0) : -1;
    const char* sm = getenv("MI_RTJ_SERIAL_MIN");
    if (sm) p->serial_min = (uint32_t)atoi(sm);
    const char* spl = getenv("MI_RTJ_SPLIT");
    p->split = spl ? (atoi(spl) != 0) : -1;
  }
  if (hipSetDevice(c->device) != hipSuccess || plan_alloc_chunks(p) != MI_RTJ_OK) {
    mi_rtj_plan_destroy(p);
    return nullptr;
  }
  if (
hipMalloc((void**)&p->d_blkoff, sizeof(uint32_t) * p->n_index) != hipSuccess) {
    fail(c, MI_RTJ_ERR_NOMEM, "hipMalloc(block index, %llu entries) failed", (unsigned long long)p->n_index);
    mi_rtj_plan_destroy(p);
    return nullptr;
  }
  {
    const char* ov = getenv("MI_RTJ_OVERLAP");
    // by default for plans of 129 .. 8191 pictures of 1080p: smaller ones do not fill the device either way and pay for
    // the cross-stream waits; longer launches gain under 1 % (profiles/r03/ab_overlap_index_with_transform.txt), and
    // kernels that share the device cannot be timed one by one — the per-kernel figures of bench.py and rocprofv3 are
    // taken on launches that run their kernels back to back
    const uint64_t groups = (uint64_t)p->n * p->max_groups;
    p->overlap = ov ? atoi(ov) == 1 : groups >= (uint64_t)kDecRotateMinGroups && groups < (uint64_t)kOverlapMaxGroups;
    p->overlap_dyn = ov ? atoi(ov) == 2 : !p->overlap && groups >= (uint64_t)kOverlapMaxGroups;  // (2: tests, on small plans)
    if (p->overlap) {
      if (hipMalloc((void**)&p->d_blkoff_b, sizeof(uint32_t) * p->n_index) != hipSuccess ||
          hipStreamCreateWithFlags(&p->own_idx, hipStreamNonBlocking) != hipSuccess) {
        fail(c, MI_RTJ_ERR_NOMEM, "second block index / index stream of the plan");
        mi_rtj_plan_destroy(p);
        return nullptr;
      }
      p->idx_stream = p->own_idx;
    } else if (p->overlap_dyn) {  // wanted, not needed: a plan without room for the second index stays as it was
      if (hipMalloc((void**)&p->d_blkoff_b, sizeof(uint32_t) * p->n_index) != hipSuccess ||
          hipStreamCreateWithFlags(&p->own_idx, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        if (p->d_blkoff_b) (void)hipFree(p->d_blkoff_b);
        p->d_blkoff_b = nullptr;
        if (p->own_idx) (void)hipStreamDestroy(p->own_idx);
        p->own_idx = nullptr;
        p->overlap_dyn = false;
      }
    }
  }
  if (plan_upload(p) != MI_RTJ_OK) {
    mi_rtj_plan_destroy(p);
    return nullptr;
  }
  if (p->overlap && hipStreamSynchronize(c->stream) != hipSuccess) {  // the descriptors are read from the other stream too
    mi_rtj_plan_destroy(p);
    return nullptr;
  }
  return p;
}

void mi_rtj_plan_destroy(mi_rtj_plan* p) {
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  for (auto& v : p->ev) drop_events(v);
  if (p->d_frames) (void)hipFree(p->d_frames);
  if (p->d_blkoff) (void)hipFree(p->d_blkoff);
  if (p->d_summary) (void)hipFree(p->d_summary);
  if (p->d_lentab) (void)hipFree(p->d_lentab);
  if (p->d_chunk_pos) (void)hipFree(p->d_chunk_pos);
  if (p->d_chunk_mb) (void)hipFree(p->d_chunk_mb);
  if (p->d_spec_chunks) (void)hipFree(p->d_spec_chunks);
  if (p->d_spec_base) (void)hipFree(p->d_spec_base);
  if (p->d_spec_rec) (void)hipFree(p->d_spec_rec);
  if (p->d_spec_nrec) (void)hipFree(p->d_spec_nrec);
  if (p->d_spec_wstart) (void)hipFree(p->d_spec_wstart);
  if (p->d_spec_hand) (void)hipFree(p->d_spec_hand);
  if (p->d_spec_fix) (void)hipFree(p->d_spec_fix);
  if (p->d_spec_flag) (void)hipFree(p->d_spec_flag);
  if (p->d_spec_nfix) (void)hipFree(p->d_spec_nfix);
  if (p->d_spec_todo) (void)hipFree(p->d_spec_todo);
  if (p->d_spec_state) (void)hipFree(p->d_spec_state);
  if (p->d_spec_ok) (void)hipFree(p->d_spec_ok);
  if (p->e_idx) (void)hipEventDestroy(p->e_idx);
  if (p->own_idx) {
    (void)hipStreamSynchronize(p->own_idx);
    (void)hipStreamDestroy(p->own_idx);
  }
  if (p->d_blkoff_b) (void)hipFree(p->d_blkoff_b);
  if (p->d_declist) (void)hipFree(p->d_declist);
  if (p->d_declist_cnt) (void)hipFree(p->d_declist_cnt);
  if (p->h_mode_seen) (void)hipHostFree(p->h_mode_seen);
  for (hipEvent_t e : p->e_read)
    if (e) (void)hipEventDestroy(e);
  delete p;
}

int mi_rtj_plan_decode(mi_rtj_plan* p, const void* d_stream, void* d_out) {
  if (!p || !d_stream || !d_out) return fail(p ? p->ctx : nullptr, MI_RTJ_ERR_ARG, "mi_rtj_plan_decode: NULL argument");
  HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
  return plan_launch(p, d_stream, d_out);
}

void mi_rtj_plan_info(const mi_rtj_plan* p, int* n, uint64_t* nb, uint64_t* bi, uint64_t* bo) {
  if (!p) return;
  if (n) *n = p->n;
  if (nb) *nb = p->n_blocks;
  if (bi) *bi = p->bytes_in;
  if (bo) *bo = p->bytes_out;
}

void mi_rtj_plan_profile(mi_rtj_plan* p, int enable) {
  if (!p) return;
  (void)hipStreamSynchronize(p->ctx->stream);
  for (auto& v : p->ev) drop_events(v);
  p->profile = enable != 0;
  p->launches = 0;
}

int mi_rtj_plan_times(mi_rtj_plan* p, float ms[MI_RTJ_NUM_KERNELS], int* launches) {
  if (!p || !ms) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = p->ctx;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < MI_RTJ_NUM_KERNELS; k++) {
    ms[k] = 0.f;
    for (auto& t : p->ev[k]) {
      float x = 0;
      HIPCHK(c, hipEventElapsedTime(&x, t.a, t.b));
      ms[k] += x;
    }
  }
  if (launches) *launches = (int)p->ev[MI_RTJ_K_DECODE].size();
  return MI_RTJ_OK;
}

int mi_rtj_plan_step_times(mi_rtj_plan* p, float* ms, int max_steps, int* steps) {
  if (!p || !ms || !steps || max_steps < 0) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = p->ctx;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int n = (int)p->ev[MI_RTJ_K_DECODE].size();
  *steps = n;
  for (int i = 0; i < n && i < max_steps; i++) {
    // the kernel that owns its begin event is the first of its launch; k_decode's end event closes the launch
    hipEvent_t first = nullptr;
    for (int k = 0; k < MI_RTJ_NUM_KERNELS && !first; k++)
      if ((int)p->ev[k].size() > i && p->ev[k][i].owns_a) first = p->ev[k][i].a;
    float x = 0;
    if (first) HIPCHK(c, hipEventElapsedTime(&x, first, p->ev[MI_RTJ_K_DECODE][i].b));
    ms[i] = x;
  }
  return MI_RTJ_OK;
}

int mi_rtj_plan_spec_lead(mi_rtj_plan* p, int* lead_bytes, int* paused_launches) {
  if (!p || !lead_bytes || !paused_launches) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = p->ctx;
  *lead_bytes = 0;
  *paused_launches = 0;
  if (!p->spec) return MI_RTJ_OK;
  *lead_bytes = p->spec_mode == 4 ? kSpecLeadVery : p->spec_mode == 3 ? kSpecLeadLong : kSpecLead;
  if (p->spec_mode == 1 || p->spec_mode == 3 || p->spec_mode == 4 || !p->d_spec_state) return MI_RTJ_OK;  // no policy
  uint32_t st[kSpecStWords];
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(st, p->d_spec_state, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *lead_bytes = spec_lead_of_level(st[kSpecStLong]);
  *paused_launches = (int)st[kSpecStPause];
  return MI_RTJ_OK;
}

int mi_rtj_plan_overlapped(const mi_rtj_plan* p) { return p && p->last_overlapped ? 1 : 0; }

int mi_rtj_plan_decode_form(mi_rtj_plan* p, int* form, int* classic_launches_left, long long* parts_listed) {
  if (!p || !form || !classic_launches_left || !parts_listed) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = p->ctx;
  *form = -1;  // no batch launch yet, or a plan whose launches are not batches: k_decode's other forms
  *classic_launches_left = 0;
  *parts_listed = 0;
  if (!p->batch_seen) return MI_RTJ_OK;
  if (p->split == 0 || !p->d_declist_cnt) {  // forced: always the classic form
    *form = 1;
    return MI_RTJ_OK;
  }
  uint32_t w[4];
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(w, p->d_declist_cnt, sizeof(w), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *form = p->split == 1 ? 0 : p->split == 0 ? 1 : (int)w[2];
  *classic_launches_left = p->split < 0 ? (int)w[3] : 0;
  *parts_listed = (long long)w[p->declist_flip ^ 1];  // the counter of the launch queued last (the next launch's is zero)
  return MI_RTJ_OK;
}

int mi_rtj_plan_spec_stats(mi_rtj_plan* p, int* proven, long long* walkers, long long* repaired) {
  if (!p || !proven || !walkers || !repaired) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = p->ctx;
  *proven = 0;
  *repaired = 0;
  *walkers = p->spec ? (long long)p->n_spec : 0;
  if (!p->spec) return MI_RTJ_OK;
  std::vector<uint32_t> ok(p->n);
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t nfix = 0;
  HIPCHK(c, hipMemcpyAsync(ok.data(), p->d_spec_ok, sizeof(uint32_t) * p->n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&nfix, p->d_spec_nfix, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  uint32_t st[kSpecStWords] = {};
  if (p->d_spec_state) HIPCHK(c, hipMemcpyAsync(st, p->d_spec_state, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // (while the speculation is paused no walker runs and nobody zeroes the repair count: it is the last unpaused launch's)
  *repaired = st[kSpecStPause] ? 0 : nfix;
  for (uint32_t v : ok) *proven += v == 1u ? 1 : 0;
  return MI_RTJ_OK;
}

#ifdef MIRTJ_EXPERIMENTS
// timing builds only (not declared in include/mi_rtjpeg.h): ticks per section of the pooling chroma waves since the last call
extern "C" int mi_rtj_debug_pool_stamps(unsigned long long out[16]) {
  if (hipDeviceSynchronize() != hipSuccess) return MI_RTJ_ERR_HIP;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pool_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return MI_RTJ_ERR_HIP;
  unsigned long long zero[16] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_pool_stamps), zero, sizeof zero) != hipSuccess) return MI_RTJ_ERR_HIP;
  return MI_RTJ_OK;
}
#endif

int mi_rtj_plan_read_index(mi_rtj_plan* p, uint32_t* dst, size_t max_entries) {
  if (!p || !dst) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = p->ctx;
  std::vector<uint32_t> all(p->n_index);
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (k_decode of the last launch waited for its index)
  HIPCHK(c, hipMemcpyAsync(all.data(), p->last ? p->d_blkoff_b : p->d_blkoff, sizeof(uint32_t) * p->n_index,
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  size_t k = 0;
  for (int i = 0; i < p->n; i++) {
    const FrameDev& f = p->h_frames[i];
    const size_t cnt = (size_t)f.nmb * 6 + 1;
    if (k + cnt > max_entries) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_plan_read_index: destination too small");
    memcpy(dst + k, all.data() + f.blk_base, cnt * sizeof(uint32_t));
    k += cnt;
  }
  return MI_RTJ_OK;
}

}  // extern "C" (reopened below)

namespace {
// Shared front half of the one-packet paths: header, (re)allocation, upload, the four kernels.
// On return the picture is being produced on c->stream into c->d_frame.
int decode_one_launch(mi_rtj_ctx* c, const uint8_t* pkt, size_t len) {
  if (!c || !pkt) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_decode: NULL argument");
  if (len < MI_RTJ_HEADER_SIZE) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_decode: packet shorter than its %d-byte header", MI_RTJ_HEADER_SIZE);
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->single) {
    c->single = new mi_rtj_plan();
    c->single->ctx = c;
    c->single->n = 1;
    c->single->h_frames.resize(1);
    // the A/B switches of the index are read once per decoder, not once per packet
    const char* mode = getenv("MI_RTJ_INDEX");
    c->single->serial_index = mode && strcmp(mode, "serial") == 0;
    const char* em = getenv("MI_RTJ_EMIT");
    c->single->emit_walk = em && strcmp(em, "walk") == 0;
    const char* sp = getenv("MI_RTJ_SPEC");
    c->single->spec_mode = sp ? atoi(sp) : -1;
  }
  mi_rtj_plan* p = c->single;
  const int rc = fill_frame(c, pkt, 0, (uint32_t)len, 0, 0, &p->h_frames[0]);
  if (rc != MI_RTJ_OK) return rc;
  const FrameDev& f = p->h_frames[0];
  const size_t need_frame = (size_t)f.w * f.h * 3 / 2;
  if (need_frame > c->frame_bytes) {  // gavl_video_frame_create in init_rtjpeg (lib/video_rtjpeg.c:54)
    uint8_t* nf = nullptr;
    HIPCHK(c, hipMalloc((void**)&nf, need_frame + kAllocPad));
    HIPCHK(c, hipMemsetAsync(nf, 0, need_frame + kAllocPad, c->stream));
    if (c->d_frame) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(c->d_frame);
    }
    c->d_frame = nf;
    c->frame_bytes = need_frame;
  }
  if (len + kAllocPad > c->pkt_cap) {
    if (c->d_pkt) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(c->d_pkt);
      c->d_pkt = nullptr;
    }
    c->pkt_cap = (len + kAllocPad) * 2;
    HIPCHK(c, hipMalloc((void**)&c->d_pkt, c->pkt_cap));
  }
  const uint64_t nidx = (((uint64_t)f.nmb * 6 + 1) + 63) & ~63ull;
  if (nidx > p->n_index) {
    if (p->d_blkoff) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(p->d_blkoff);
      p->d_blkoff = nullptr;
    }
    HIPCHK(c, hipMalloc((void**)&p->d_blkoff, sizeof(uint32_t) * nidx));
    p->n_index = nidx;
  }
  p->n_blocks = (uint64_t)f.nmb * 6;
  p->max_groups = (f.nmb + kMbPerGroup - 1) / kMbPerGroup;
  {
    const int rc2 = plan_alloc_chunks(p);  // lays the packet's chunks out; allocates only when it is larger than any before it
    if (rc2 != MI_RTJ_OK) return rc2;
  }
  // pageable source: the runtime stages it; a private pinned staging copy measured no faster
  HIPCHK(c, hipMemcpyAsync(c->d_pkt, pkt, len, hipMemcpyHostToDevice, c->stream));
  int r = plan_upload(p);
  if (r != MI_RTJ_OK) return r;
  return plan_launch(p, c->d_pkt, c->d_frame);
}
}  // namespace

extern "C" {

int mi_rtj_decode(mi_rtj_ctx* c, const uint8_t* pkt, size_t len, uint8_t* const dst[3], const int dst_stride[3],
                  int crop_w, int crop_h) {
  const int r = decode_one_launch(c, pkt, len);
  if (r != MI_RTJ_OK) return r;
  const FrameDev& f = c->single->h_frames[0];
  if (dst) {
    // gavl_video_frame_copy(format, f, priv->frame) (lib/video_rtjpeg.c:82): image_width x
    // image_height region, each side's own strides — done by the copy engine on the way out.
    if (!dst[0] || !dst[1] || !dst[2] || !dst_stride) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_decode: NULL plane");
    if (crop_w <= 0 || crop_h <= 0 || crop_w > (int)f.w || crop_h > (int)f.h)
      return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_decode: crop %dx%d outside the coded %ux%u picture", crop_w, crop_h, f.w, f.h);
    const size_t ysz = (size_t)f.w * f.h;
    const int cw = (crop_w + 1) / 2, ch = (crop_h + 1) / 2;
    HIPCHK(c, hipMemcpy2DAsync(dst[0], (size_t)dst_stride[0], c->d_frame, f.w, (size_t)crop_w, (size_t)crop_h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(dst[1], (size_t)dst_stride[1], c->d_frame + ysz, f.w / 2, (size_t)cw, (size_t)ch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(dst[2], (size_t)dst_stride[2], c->d_frame + ysz + ysz / 4, f.w / 2, (size_t)cw, (size_t)ch, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MI_RTJ_OK;
}

int mi_rtj_decode_nocopy(mi_rtj_ctx* c, const uint8_t* pkt, size_t len, const uint8_t* planes[3], int strides[3]) {
  if (!planes || !strides) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_decode_nocopy: NULL argument");
  const int r = decode_one_launch(c, pkt, len);
  if (r != MI_RTJ_OK) return r;
  const FrameDev& f = c->single->h_frames[0];
  const size_t ysz = (size_t)f.w * f.h, fsz = ysz * 3 / 2;
  if (fsz > c->h_frame_cap) {
    if (c->h_frame) (void)hipHostFree(c->h_frame);
    c->h_frame = nullptr;
    HIPCHK(c, hipHostMalloc((void**)&c->h_frame, fsz, hipHostMallocDefault));
    c->h_frame_cap = fsz;
  }
  HIPCHK(c, hipMemcpyAsync(c->h_frame, c->d_frame, fsz, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  planes[0] = c->h_frame;
  planes[1] = c->h_frame + ysz;
  planes[2] = c->h_frame + ysz + ysz / 4;
  strides[0] = (int)f.w;
  strides[1] = strides[2] = (int)f.w / 2;
  return MI_RTJ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined session: the decoder a frame-owning plugin instance reads ahead with (lib/video.c:420-441 nocopy source,
// lib/video_v4l2_m2m.c:43-131 as the in-tree precedent of a decoder with packets in flight).
// ---------------------------------------------------------------------------------------------------------------
namespace {
inline double host_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// everything of one packet that goes to the device: copy in, kernels, copy out (called on the worker thread, or on
// the caller's when the session runs without one)
// how the pinned host pictures are allocated (MI_RTJ_HOST_FLAGS, experiments: 1 non-coherent, 2 coherent, 4 write-combined)
unsigned pic_host_flags() {
  const char* f = exp_env("MI_RTJ_HOST_FLAGS");
  const int v = f ? atoi(f) : 0;
  unsigned fl = hipHostMallocDefault;
  if (v & 1) fl |= hipHostMallocNonCoherent;
  if (v & 2) fl |= hipHostMallocCoherent;
  if (v & 4) fl |= hipHostMallocWriteCombined;
  return fl;
}

// queue the copy out of slot `last` together with the decoded, not yet copied slots of its group right before it
int pipe_copy_out(mi_rtj_pipe* q, int last) {
  mi_rtj_ctx* c = q->ctx;
  int first = last;
  while (first % q->group != 0 && q->slot[first - 1].out_state == 1 && q->slot[first - 1].w == q->slot[last].w &&
         q->slot[first - 1].h == q->slot[last].h)
    first--;
  PipeSlot& a = q->slot[first];
  PipeSlot& z = q->slot[last];
  const size_t fsz = (size_t)z.w * z.h * 3 / 2;
  const size_t bytes = (size_t)(z.d_pic - a.d_pic) + fsz;  // one slot: fsz; a group: its members lie side by side
  hipStream_t so = q->s_out[(last / q->group) % q->n_out];
  // "decoded" is recorded here, once per copy, not behind every packet's k_decode: the kernels of the group's members
  // were queued before this point on the same stream (whatever else was queued behind them merely makes the copy wait
  // a little longer: that only happens when the caller asks for a picture of an incomplete group)
  HIPCHK(c, hipEventRecord(z.e_dec, c->stream));
  HIPCHK(c, hipStreamWaitEvent(so, z.e_dec, 0));
  if (q->exp_skip & 1) {
  } else if (q->out_kernel) {
    hipLaunchKernelGGL(k_copy16, dim3(q->out_kernel), dim3(256), 0, so, (const uint4*)a.d_pic, (uint4*)a.h_pic, bytes / 16);
  } else {
    const double t0 = q->stats ? host_now() : 0.0;
    HIPCHK(c, hipMemcpyAsync(a.h_pic, a.d_pic, bytes, hipMemcpyDeviceToHost, so));
    if (q->stats) q->t_d2h_call += host_now() - t0;
  }
  HIPCHK(c, hipEventRecord(z.e_out, so));
  for (int i = first; i <= last; i++) {
    q->slot[i].out_state = 2;
    q->slot[i].out_ev = z.e_out;
  }
  return MI_RTJ_OK;
}

int pipe_issue(mi_rtj_pipe* q, PipeSlot& sl) {
  mi_rtj_ctx* c = q->ctx;
  mi_rtj_plan* p = sl.plan;
  // (the copy in has a stream of its own: queued between the kernels of successive packets the session ran at
  // 6-10 K pictures per second instead of 12 K, and the deeper the pipeline the slower — profiles/r02/e2e_*.txt)
  if (!(q->exp_skip & 4))
    HIPCHK(c, hipMemcpyAsync(sl.d_stage, sl.h_stage, sizeof(FrameDev) + sl.len, hipMemcpyHostToDevice, q->s_in));
  HIPCHK(c, hipEventRecord(sl.e_in, q->s_in));
  HIPCHK(c, hipStreamWaitEvent(p->idx_stream ? p->idx_stream : c->stream, sl.e_in, 0));
  // kernels, in submission order on the instance's stream.  Unchanged (0xFF) blocks: k_decode fetches them from the
  // predecessor's picture (a packet of another size has no predecessor in that sense: its unchanged blocks keep what
  // the slot's buffer holds, zeros at first)
  p->prev_pic = sl.prev;
  const int rc = (q->exp_skip & 2) ? MI_RTJ_OK : plan_launch(p, sl.d_stage, sl.d_pic);
  if (rc != MI_RTJ_OK) return rc;
  // copy out: queued now if this packet closes its group, else when the group's last packet is decoded or the caller
  // asks for one of its pictures, whichever comes first
  sl.out_state = 1;
  const int idx = (int)(&sl - q->slot.data());
  if (idx % q->group == q->group - 1) return pipe_copy_out(q, idx);
  return MI_RTJ_OK;
}

// ---- index groups: copy in now, kernels when the group is complete ----
int pipe_stage(mi_rtj_pipe* q, PipeSlot& sl) {
  mi_rtj_ctx* c = q->ctx;
  if (!(q->exp_skip & 4))
    HIPCHK(c, hipMemcpyAsync(sl.d_stage, sl.h_stage, sizeof(FrameDev) + sl.len, hipMemcpyHostToDevice, q->s_in));
  sl.staged = 1;
  sl.out_state = 0;
  q->nstaged++;
  return MI_RTJ_OK;
}

// queue the kernels of the `count` staged slots first .. first + count - 1 (one aligned group or a part of one)
int pipe_issue_group(mi_rtj_pipe* q, int first, int count) {
  mi_rtj_ctx* c = q->ctx;
  mi_rtj_plan* p = q->gplan[first / q->igroup];
  p->n = count;
  p->h_frames.resize((size_t)count);
  uint64_t nidx = 0;
  uint32_t max_groups = 0;
  p->n_blocks = 0;
  for (int j = 0; j < count; j++) {
    FrameDev& f = q->slot[first + j].fd;
    f.blk_base = (uint32_t)nidx;
    nidx += (((uint64_t)f.nmb * 6 + 1) + 63) & ~63ull;
    p->n_blocks += (uint64_t)f.nmb * 6;
    max_groups = std::max(max_groups, (f.nmb + kMbPerGroup - 1) / kMbPerGroup);
    p->h_frames[(size_t)j] = f;
  }
  p->max_groups = max_groups;
  if (nidx > p->n_index) {
    if (p->d_blkoff) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(p->d_blkoff);
      p->d_blkoff = nullptr;
      p->n_index = 0;
    }
    const uint64_t cap = nidx / (uint64_t)count * (uint64_t)q->igroup;  // room for a whole group of such pictures
    HIPCHK(c, hipMalloc((void**)&p->d_blkoff, sizeof(uint32_t) * std::max(cap, nidx)));
    p->n_index = std::max(cap, nidx);
  }
  {
    const int rc = plan_alloc_chunks(p);  // chunk bases of the group's packets; allocates only when the group outgrows its buffers
    if (rc != MI_RTJ_OK) return rc;
  }
  // the descriptors follow the packets on the copy-in stream; the kernels wait for both
  for (int j = 0; j < count; j++) q->h_desc[first + j] = p->h_frames[(size_t)j];
  HIPCHK(c, hipMemcpyAsync(q->d_desc + first, q->h_desc + first, sizeof(FrameDev) * (size_t)count, hipMemcpyHostToDevice, q->s_in));
  PipeSlot& lastsl = q->slot[first + count - 1];
  HIPCHK(c, hipEventRecord(lastsl.e_in, q->s_in));
  HIPCHK(c, hipStreamWaitEvent(c->stream, lastsl.e_in, 0));
  p->d_frames = q->d_desc + first;
  // (data_off of these descriptors is the packet's device ADDRESS: the kernels add it to a null stream base)
  if (!(q->exp_skip & 2)) {
    p->prev_pic = nullptr;
    const int rc = plan_launch(p, nullptr, nullptr, kLaunchIndex);
    if (rc != MI_RTJ_OK) return rc;
  }
  for (int j = 0; j < count; j++) {
    PipeSlot& sl = q->slot[first + j];
    if (!(q->exp_skip & 2)) {
      p->prev_pic = sl.prev;
      const int rc = plan_launch(p, nullptr, sl.d_pic, kLaunchDecode, (uint32_t)j, 1);
      if (rc != MI_RTJ_OK) return rc;
    }
    sl.staged = 0;
    sl.out_state = 1;
    q->nstaged--;
    if ((first + j) % q->group == q->group - 1) {
      const int rc = pipe_copy_out(q, first + j);
      if (rc != MI_RTJ_OK) return rc;
    }
  }
  return MI_RTJ_OK;
}

// the staged slots are the last q->nstaged submitted ones; queue their kernels
int pipe_issue_staged(mi_rtj_pipe* q) {
  if (q->nstaged <= 0) return MI_RTJ_OK;
  const int count = q->nstaged;
  const int last = (q->head + q->count - 1 + q->depth) % q->depth;
  const int first = last - (count - 1);  // a group never wraps: the depth is a multiple of the group size
  const int rc = pipe_issue_group(q, first, count);
  if (rc != MI_RTJ_OK) {  // what could not be queued yields no pictures
    for (int j = 0; j < count; j++) {
      PipeSlot& sl = q->slot[first + j];
      if (sl.staged) {
        sl.staged = 0;
        sl.rc = rc;
        sl.err = q->ctx->err;
      }
    }
    q->nstaged = 0;
  }
  return rc;
}

void pipe_worker(mi_rtj_pipe* q) {
  (void)hipSetDevice(q->ctx->device);
  for (;;) {
    int idx;
    {
      std::unique_lock<std::mutex> lk(q->mu);
      q->cv_work.wait(lk, [&] { return q->stop || q->job_count > 0; });
      if (q->job_count == 0) return;  // stop, and nothing left to queue
      idx = q->jobs[q->job_head];
    }
    PipeSlot& sl = q->slot[idx];
    const int rc = pipe_issue(q, sl);
    {
      std::lock_guard<std::mutex> lk(q->mu);
      sl.rc = rc;
      if (rc != MI_RTJ_OK) sl.err = q->ctx->err;
      sl.issued = 1;
      q->job_head = (q->job_head + 1) % q->jobs.size();
      q->job_count--;
    }
    q->cv_done.notify_all();
  }
}

// wait until the worker has nothing left to queue (allocation changes, flush, destroy)
void pipe_drain_jobs(mi_rtj_pipe* q) {
  if (!q->threaded) return;
  std::unique_lock<std::mutex> lk(q->mu);
  q->cv_done.wait(lk, [&] { return q->job_count == 0; });
}
}  // namespace

mi_rtj_pipe* mi_rtj_pipe_create(mi_rtj_ctx* c, int depth, int max_w, int max_h) {
  if (!c || depth < 2 || depth > 64 || max_w < 0 || max_h < 0) {
    fail(c, MI_RTJ_ERR_ARG, "mi_rtj_pipe_create: bad argument (depth 2..64)");
    return nullptr;
  }
  if (hipSetDevice(c->device) != hipSuccess) {
    fail(c, MI_RTJ_ERR_HIP, "mi_rtj_pipe_create: hipSetDevice failed");
    return nullptr;
  }
  mi_rtj_pipe* q = new mi_rtj_pipe();
  q->ctx = c;
  q->depth = depth;
  q->max_w = max_w;
  q->max_h = max_h;
  q->slot.resize(depth);
  {
    const char* os = exp_env("MI_RTJ_OUT_STREAMS");
    q->n_out = os && atoi(os) == 1 ? 1 : 2;
    const char* ok_ = exp_env("MI_RTJ_OUT_KERNEL");
    q->out_kernel = ok_ ? atoi(ok_) : 0;  // the value is the copy kernel's grid (workgroups of 256)
    const char* sk = exp_env("MI_RTJ_EXP_SKIP");
    q->exp_skip = sk ? atoi(sk) : 0;
    const char* ps = getenv("MI_RTJ_PIPE_STATS");
    q->stats = ps && atoi(ps) != 0;
    const char* wm = exp_env("MI_RTJ_WAIT");
    q->wait_mode = wm && strcmp(wm, "query") == 0 ? 1 : 0;
  }
  {
    const char* og = getenv("MI_RTJ_OUT_GROUP");
    // fours from twelve packets in flight on (three groups: one being copied, one queued behind it, one being filled):
    // 16.0-16.2 K pictures per second at 1080p against 15.0 K in pairs, 4.36 K at 4K against 4.26 K (the link gives
    // 4.39 K); with two groups of four the copy engine idles while the caller refills one (9.9 K)
    const int g = og ? atoi(og) : depth >= 12 ? 4 : 2;
    // (groups need every picture to have the session's size, and the caller's thread to be the one that queues copies)
    const char* th0 = exp_env("MI_RTJ_PIPE_THREAD");
    q->group = (max_w > 0 && max_h > 0 && !(th0 && atoi(th0) != 0) && (g == 2 || g == 4)) ? g : 1;
    // the index is built in groups too, by default the same ones (needs the same things: one coded size, the
    // caller's thread queuing the work); MI_RTJ_IDX_GROUP = 1 is the packet-by-packet index of rounds 2 and 3
    const char* ig = getenv("MI_RTJ_IDX_GROUP");
    const int igv = ig ? atoi(ig) : q->group;
    q->igroup = (max_w > 0 && max_h > 0 && !(th0 && atoi(th0) != 0) && (igv == 2 || igv == 4)) ? igv : 1;
    const int unit = std::max(q->group, q->igroup);  // (1, 2 or 4 each: the larger is a multiple of the other)
    if (unit > 1) {
      q->depth = depth = (depth + unit - 1) / unit * unit;
      q->slot.resize(depth);
    }
    if (q->group > 1) q->grp_fsz = (size_t)max_w * max_h * 3 / 2;
  }
  // The runtime multiplexes the streams of a process onto a few hardware queues PER PRIORITY (four by default,
  // GPU_MAX_HW_QUEUES): with every stream at the default priority two sessions' eight streams shared four queues and one
  // session's copy in waited behind the other's copy out (two sessions then aggregated LESS than one).
  // The copy streams of a session are therefore created at the two other priorities the device offers — packets in at
  // the highest, pictures out at the lowest — which puts them into queue pools of their own: kernels, copies in and
  // copies out of several sessions no longer meet in one queue, and the application need not set anything.  (For copies
  // the priority itself changes nothing: they run on the copy engines.)  MI_RTJ_STREAM_PRIO=0: all default (A/B).
  int prio_lo = 0, prio_hi = 0;  // numerically: lowest priority = largest value
  {
    const char* sp = getenv("MI_RTJ_STREAM_PRIO");
    if ((!sp || atoi(sp) != 0) && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    if (sp && atoi(sp) == 0) prio_lo = prio_hi = 0;
  }
  auto make_stream = [&](hipStream_t* st, int prio) {
    return (prio_lo != prio_hi ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio)
                               : hipStreamCreateWithFlags(st, hipStreamNonBlocking)) == hipSuccess;
  };
  bool ok = make_stream(&q->s_in, prio_hi);
  {
    // off unless asked for: with the index on a stream of its own a 1080p session ran at 6,600 pictures per second
    // instead of 12,000 (profiles/r03/e2e_ab.txt) — every packet then crosses streams twice more, and the runtime's
    // cross-stream waits cost more than the overlap of one packet's index with its predecessor's transform gives
    const char* ix = exp_env("MI_RTJ_IDX_STREAM");
    if (ix && atoi(ix) != 0) ok = ok && hipStreamCreateWithFlags(&q->s_idx, hipStreamNonBlocking) == hipSuccess;
  }
  for (int i = 0; i < q->n_out; i++) ok = ok && make_stream(&q->s_out[i], prio_lo);
  const char* mode = getenv("MI_RTJ_INDEX");
  const char* em = getenv("MI_RTJ_EMIT");
  for (auto& sl : q->slot) {
    sl.plan = new mi_rtj_plan();
    sl.plan->ctx = c;
    sl.plan->n = 1;
    sl.plan->spec_mode = 0;  // one packet per launch: the exact index (a lone walker takes longer than all of it)
    sl.plan->serial_index = mode && strcmp(mode, "serial") == 0;
    sl.plan->emit_walk = em && strcmp(em, "walk") == 0;
    sl.plan->h_frames.resize(1);
    sl.plan->idx_stream = q->s_idx;
    ok = ok && hipEventCreateWithFlags(&sl.e_in, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&sl.e_dec, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&sl.e_out, hipEventDisableTiming) == hipSuccess;
  }
  if (ok && q->igroup > 1) {  // one plan per index group, the descriptors of all slots in one array
    ok = hipHostMalloc((void**)&q->h_desc, sizeof(FrameDev) * (size_t)depth, hipHostMallocDefault) == hipSuccess &&
         hipMalloc((void**)&q->d_desc, sizeof(FrameDev) * (size_t)depth) == hipSuccess;
    q->gplan.assign((size_t)(depth / q->igroup), nullptr);
    for (auto& gp : q->gplan) {
      gp = new mi_rtj_plan();
      gp->ctx = c;
      gp->n = q->igroup;
      gp->spec_mode = 0;  // a handful of packets per launch: the exact index
      gp->serial_index = mode && strcmp(mode, "serial") == 0;
      gp->emit_walk = em && strcmp(em, "walk") == 0;
      gp->h_frames.resize((size_t)q->igroup);
    }
  }
  if (ok && q->group > 1) {  // the group buffers: `group` pictures side by side, on the device and in pinned host memory
    const int ng = depth / q->group;
    q->d_grp.assign(ng, nullptr);
    q->h_grp.assign(ng, nullptr);
    const size_t gb = q->grp_fsz * (size_t)q->group;
    for (int g = 0; g < ng && ok; g++) {
      ok = hipMalloc((void**)&q->d_grp[g], gb + kAllocPad) == hipSuccess &&
           hipMemsetAsync(q->d_grp[g], 0, gb + kAllocPad, c->stream) == hipSuccess &&
           hipHostMalloc((void**)&q->h_grp[g], gb, pic_host_flags()) == hipSuccess;
      for (int j = 0; j < q->group && ok; j++) {
        PipeSlot& sl = q->slot[g * q->group + j];
        sl.d_pic = q->d_grp[g] + (size_t)j * q->grp_fsz;
        sl.h_pic = q->h_grp[g] + (size_t)j * q->grp_fsz;
        sl.pic_cap = q->grp_fsz;
        sl.owns_pic = false;
      }
    }
  }
  if (!ok) {
    fail(c, MI_RTJ_ERR_HIP, "mi_rtj_pipe_create: cannot create streams / events / picture buffers");
    mi_rtj_pipe_destroy(q);
    return nullptr;
  }
  // MI_RTJ_PIPE_THREAD=1: the HIP calls of a packet are made by a worker thread of the session instead of the
  // caller's.  It frees the caller (which then only copies the packet into staging) but buys no throughput where it
  // was hoped to: 12,020 against 12,070 pictures per second at 1080p (the copy out bounds the session there, see
  // DESIGN.md), 20,260 against 18,870 at 320x240 (profiles/r02/e2e_worker_thread.txt).  Off by default.
  const char* th = exp_env("MI_RTJ_PIPE_THREAD");
  q->threaded = th && atoi(th) != 0;
  q->jobs.assign((size_t)depth, 0);
  if (q->threaded) q->worker = std::thread(pipe_worker, q);
  return q;
}

void mi_rtj_pipe_destroy(mi_rtj_pipe* q) {
  if (!q) return;
  mi_rtj_ctx* c = q->ctx;
  if (q->stats && q->returned)
    fprintf(stderr, "{\"pipe_stats\": {\"pictures\": %llu, \"us_per_picture\": {\"staging_memcpy\": %.1f, \"queueing_calls\": %.1f, \"of_which_the_d2h_memcpy_call\": %.1f, \"waiting_for_copy_out\": %.1f}}}\n",
            (unsigned long long)q->returned, 1e6 * q->t_stage / q->returned, 1e6 * q->t_issue / q->returned, 1e6 * q->t_d2h_call / q->returned, 1e6 * q->t_wait / q->returned);
  (void)hipSetDevice(c->device);
  if (q->worker.joinable()) {
    {
      std::lock_guard<std::mutex> lk(q->mu);
      q->stop = true;
    }
    q->cv_work.notify_all();
    q->worker.join();  // it queues what it still holds, then leaves
  }
  if (q->s_in) (void)hipStreamSynchronize(q->s_in);
  if (q->s_idx) (void)hipStreamSynchronize(q->s_idx);
  (void)hipStreamSynchronize(c->stream);
  for (hipStream_t so : q->s_out)
    if (so) (void)hipStreamSynchronize(so);
  for (auto& sl : q->slot) {
    if (sl.plan) {
      sl.plan->d_frames = nullptr;  // points into d_stage, not owned by the plan
      mi_rtj_plan_destroy(sl.plan);
    }
    if (sl.h_stage) (void)hipHostFree(sl.h_stage);
    if (sl.d_stage) (void)hipFree(sl.d_stage);
    if (sl.owns_pic && sl.d_pic) (void)hipFree(sl.d_pic);
    if (sl.owns_pic && sl.h_pic) (void)hipHostFree(sl.h_pic);
    if (sl.e_in) (void)hipEventDestroy(sl.e_in);
    if (sl.e_dec) (void)hipEventDestroy(sl.e_dec);
    if (sl.e_out) (void)hipEventDestroy(sl.e_out);
  }
  for (mi_rtj_plan* gp : q->gplan)
    if (gp) {
      gp->d_frames = nullptr;  // points into d_desc, not owned by the plan
      mi_rtj_plan_destroy(gp);
    }
  if (q->h_desc) (void)hipHostFree(q->h_desc);
  if (q->d_desc) (void)hipFree(q->d_desc);
  for (uint8_t* g : q->d_grp)
    if (g) (void)hipFree(g);
  for (uint8_t* g : q->h_grp)
    if (g) (void)hipHostFree(g);
  if (q->s_in) (void)hipStreamDestroy(q->s_in);
  if (q->s_idx) {
    (void)hipStreamSynchronize(q->s_idx);
    (void)hipStreamDestroy(q->s_idx);
  }
  for (hipStream_t so : q->s_out)
    if (so) (void)hipStreamDestroy(so);
  delete q;
}

int mi_rtj_pipe_room(const mi_rtj_pipe* q) {
  if (!q) return 0;
  return q->depth - q->count - (q->lent >= 0 ? 1 : 0);
}

int mi_rtj_pipe_pending(const mi_rtj_pipe* q) { return q ? q->count : 0; }

int mi_rtj_pipe_submit(mi_rtj_pipe* q, const uint8_t* pkt, size_t len, uint64_t tag) {
  if (!q || !pkt) return fail(q ? q->ctx : nullptr, MI_RTJ_ERR_ARG, "mi_rtj_pipe_submit: NULL argument");
  mi_rtj_ctx* c = q->ctx;
  // (an error message is written into the instance, which the worker thread of a threaded session writes too: the
  // worker is idle before any of the refusals below speaks — ADVICE r2)
  if (len < MI_RTJ_HEADER_SIZE || mi_rtj_pipe_room(q) <= 0) pipe_drain_jobs(q);
  if (len < MI_RTJ_HEADER_SIZE) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_pipe_submit: packet shorter than its %d-byte header", MI_RTJ_HEADER_SIZE);
  if (mi_rtj_pipe_room(q) <= 0) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_pipe_submit: %d packets in flight already (take a picture first)", q->count);
  HIPCHK(c, hipSetDevice(c->device));
  // the header against the stream's size BEFORE anything is allocated for it (a 65520 x 65520 header would ask for
  // gigabytes; the reference's frame has the container's size whatever the packet says, lib/video_rtjpeg.c:50-54)
  {
    const int hw = pkt[6] | (pkt[7] << 8), hh = pkt[8] | (pkt[9] << 8);
    if (q->max_w && q->max_h && (hw != q->max_w || hh != q->max_h)) pipe_drain_jobs(q);
    if (q->max_w && q->max_h && (hw != q->max_w || hh != q->max_h))
      return fail(c, MI_RTJ_ERR_GEOMETRY, "packet header %dx%d does not match the stream's coded size %dx%d", hw, hh, q->max_w, q->max_h);
  }
  // the slots form a ring: the lent one (if any) sits just before head, then head .. head+count-1, then the free ones
  const int idx = (q->head + q->count) % q->depth;
  PipeSlot& sl = q->slot[idx];
  mi_rtj_plan* p = sl.plan;
  // (fill_frame and everything below may write the instance's error string and allocate: the worker must be idle
  // whenever that can happen, i.e. on an error or when a buffer has to grow — not in the steady state)
  FrameDev f;
  {
    FrameDev probe;
    const int hw = pkt[6] | (pkt[7] << 8), hh = pkt[8] | (pkt[9] << 8);
    const bool geometry_ok = hw > 0 && hh > 0 && !(hw & 15) && !(hh & 15);
    const size_t fsz0 = geometry_ok ? (size_t)hw * hh * 3 / 2 : 0;
    const uint64_t nidx0 = geometry_ok ? ((((uint64_t)(hw / 16) * (hh / 16)) * 6 + 1) + 63) & ~63ull : 0;
    const uint32_t nchunks0 = (uint32_t)((len - MI_RTJ_HEADER_SIZE + kChunk - 1) / kChunk) + 1u;
    const bool grows = !geometry_ok || fsz0 > sl.pic_cap || sizeof(FrameDev) + len + kAllocPad > sl.stage_cap ||
                       nidx0 > p->n_index || nchunks0 > p->n_chunks || (uint64_t)nchunks0 + 1 > p->n_chunk_entries;
    if (grows) pipe_drain_jobs(q);
    (void)probe;
  }
  const int rc = fill_frame(c, pkt, 0, (uint32_t)len, 0, 0, &f);
  if (rc != MI_RTJ_OK) return rc;
  p->h_frames[0] = f;
  const size_t fsz = (size_t)f.w * f.h * 3 / 2;
  if (fsz > sl.pic_cap && !sl.owns_pic)  // cannot happen: grouped sessions refuse packets of another size above
    return fail(c, MI_RTJ_ERR_GEOMETRY, "mi_rtj_pipe_submit: picture larger than the session's");
  if (fsz > sl.pic_cap) {
    if (sl.d_pic) (void)hipFree(sl.d_pic);
    if (sl.h_pic) (void)hipHostFree(sl.h_pic);
    sl.d_pic = sl.h_pic = nullptr;
    sl.pic_cap = 0;
    HIPCHK(c, hipMalloc((void**)&sl.d_pic, fsz + kAllocPad));
    HIPCHK(c, hipMemsetAsync(sl.d_pic, 0, fsz + kAllocPad, c->stream));
    HIPCHK(c, hipHostMalloc((void**)&sl.h_pic, fsz, pic_host_flags()));
    sl.pic_cap = fsz;
  }
  const size_t need = sizeof(FrameDev) + len + kAllocPad;
  if (need > sl.stage_cap) {
    if (sl.h_stage) (void)hipHostFree(sl.h_stage);
    if (sl.d_stage) (void)hipFree(sl.d_stage);
    sl.h_stage = sl.d_stage = nullptr;
    sl.stage_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&sl.h_stage, need * 2, hipHostMallocDefault));
    HIPCHK(c, hipMalloc((void**)&sl.d_stage, need * 2));
    sl.stage_cap = need * 2;
  }
  const uint64_t nidx = (((uint64_t)f.nmb * 6 + 1) + 63) & ~63ull;
  const bool grouped = q->igroup > 1;  // the index of this packet is built with its group's (pipe_issue_group)
  if (!grouped) {
    if (nidx > p->n_index) {
      if (p->d_blkoff) (void)hipFree(p->d_blkoff);
      p->d_blkoff = nullptr;
      p->n_index = 0;
      HIPCHK(c, hipMalloc((void**)&p->d_blkoff, sizeof(uint32_t) * nidx));
      p->n_index = nidx;
    }
    p->n_blocks = (uint64_t)f.nmb * 6;
    p->max_groups = (f.nmb + kMbPerGroup - 1) / kMbPerGroup;
    const int rc2 = plan_alloc_chunks(p);  // sets the chunk bases in the descriptor; allocates only when it grows
    if (rc2 != MI_RTJ_OK) return rc2;
  }
  // descriptor and packet travel together: one copy in, on its own stream
  FrameDev& fd = p->h_frames[0];  // plan_alloc_chunks has set its chunk bases
  fd.data_off = sizeof(FrameDev) + MI_RTJ_HEADER_SIZE;
  if (grouped) {
    // the kernels of a group get a null stream base: data_off is the device address of the packet's first data byte
    // (the packets of a group lie in their slots' own buffers); block and chunk bases are set when the group is queued
    fd.data_off = (uint64_t)(uintptr_t)sl.d_stage + sizeof(FrameDev) + MI_RTJ_HEADER_SIZE;
    sl.fd = fd;
  }
  const double ts0 = q->stats ? host_now() : 0.0;
  memcpy(sl.h_stage, &fd, sizeof(FrameDev));
  memcpy(sl.h_stage + sizeof(FrameDev), pkt, len);
  if (q->stats) q->t_stage += host_now() - ts0;
  p->d_frames = (FrameDev*)sl.d_stage;
  sl.len = len;
  sl.tag = tag;
  sl.w = (int)fd.w;
  sl.h = (int)fd.h;
  const uint8_t* const was_prev = q->prev_pic;
  const size_t was_bytes = q->prev_bytes;
  sl.prev = q->prev_pic && q->prev_bytes == fsz && q->prev_pic != sl.d_pic ? q->prev_pic : nullptr;
  q->prev_pic = sl.d_pic;
  q->prev_bytes = fsz;
  q->count++;
  q->submitted++;
  if (!q->threaded) {
    const double ti0 = q->stats ? host_now() : 0.0;
    bool group_failed = false;
    if (grouped) {
      sl.rc = pipe_stage(q, sl);
      // the group's last slot: queue the kernels of the whole group.  Should THAT fail, the group's packets are in all
      // the same, this one included: their slots carry the error and yield no pictures, mi_rtj_pipe_next reports it once
      // per packet, and submit returns OK — one behaviour for all members of the group (the last member used
      // to be taken out again and reported twice while its siblings stayed in flight)
      if (sl.rc == MI_RTJ_OK && idx % q->igroup == q->igroup - 1) group_failed = pipe_issue_staged(q) != MI_RTJ_OK;
    } else {
      sl.rc = pipe_issue(q, sl);
    }
    if (q->stats) q->t_issue += host_now() - ti0;
    sl.issued = 1;
    if (group_failed) return MI_RTJ_OK;
    if (sl.rc != MI_RTJ_OK) {  // this packet's own copy in (or, ungrouped, its kernels) could not be queued: undo — it never
                               // went in, its predecessor is still the last picture
      q->count--;
      q->submitted--;
      q->prev_pic = was_prev;
      q->prev_bytes = was_bytes;
      return sl.rc;
    }
    return MI_RTJ_OK;
  }
  {
    std::lock_guard<std::mutex> lk(q->mu);
    sl.issued = 0;
    sl.rc = MI_RTJ_OK;
    q->jobs[(q->job_head + q->job_count) % q->jobs.size()] = idx;
    q->job_count++;
  }
  q->cv_work.notify_one();
  return MI_RTJ_OK;
}

int mi_rtj_pipe_next(mi_rtj_pipe* q, const uint8_t* planes[3], int strides[3], int* w, int* h, uint64_t* tag) {
  if (!q) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = q->ctx;
  q->lent = -1;  // the picture handed out before is the caller's no longer
  if (q->count == 0) return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_pipe_next: nothing in flight");
  HIPCHK(c, hipSetDevice(c->device));
  PipeSlot& sl = q->slot[q->head];
  if (q->threaded) {
    std::unique_lock<std::mutex> lk(q->mu);
    q->cv_done.wait(lk, [&] { return sl.issued != 0; });
  }
  if (sl.staged) {  // its index group is not complete: the kernels of what is staged are queued now
    const double ti0 = q->stats ? host_now() : 0.0;
    (void)pipe_issue_staged(q);  // (a failure is in the slots' rc)
    if (q->stats) q->t_issue += host_now() - ti0;
  }
  if (sl.rc != MI_RTJ_OK) {  // queuing this packet failed (on the worker, or with its group): it yields no picture
    const int rc = sl.rc;
    if (!sl.err.empty()) c->err = sl.err;
    q->head = (q->head + 1) % q->depth;
    q->count--;
    return rc;
  }
  if (sl.out_state == 1) {  // its group is not complete yet: the copy goes out now, with what is decoded behind it
    int last = q->head;
    while ((last + 1) % q->group != 0 && last + 1 < q->depth && q->slot[last + 1].out_state == 1) last++;
    const int rc = pipe_copy_out(q, last);
    if (rc != MI_RTJ_OK) return rc;
  }
  {
    const double tw0 = q->stats ? host_now() : 0.0;
    hipEvent_t const ev = sl.out_ev ? sl.out_ev : sl.e_out;
    if (q->wait_mode == 1) {  // MI_RTJ_WAIT=query (A/B): poll instead of blocking inside the runtime
      hipError_t e;
      while ((e = hipEventQuery(ev)) == hipErrorNotReady) std::this_thread::yield();
      HIPCHK(c, e);
    } else {
      HIPCHK(c, hipEventSynchronize(ev));
    }
    if (q->stats) q->t_wait += host_now() - tw0;
  }
  sl.out_state = 0;
  if (planes) {
    const size_t ysz = (size_t)sl.w * sl.h;
    planes[0] = sl.h_pic;
    planes[1] = sl.h_pic + ysz;
    planes[2] = sl.h_pic + ysz + ysz / 4;
    q->lent = q->head;
  }
  if (strides) {
    strides[0] = sl.w;
    strides[1] = strides[2] = sl.w / 2;
  }
  if (w) *w = sl.w;
  if (h) *h = sl.h;
  if (tag) *tag = sl.tag;
  q->head = (q->head + 1) % q->depth;
  q->count--;
  q->returned++;
  return MI_RTJ_OK;
}

// Kernel times of a session (bench.py's roofline for the in-order workloads): profiling records two events per kernel
// and packet on the submitting thread, so it is switched on for a lap of its own, never inside a timed region.
int mi_rtj_pipe_profile(mi_rtj_pipe* q, int enable) {
  if (!q) return MI_RTJ_ERR_ARG;
  const int rc = mi_rtj_pipe_flush(q);
  if (rc != MI_RTJ_OK) return rc;
  for (auto& sl : q->slot) mi_rtj_plan_profile(sl.plan, enable);
  for (mi_rtj_plan* gp : q->gplan) mi_rtj_plan_profile(gp, enable);
  return MI_RTJ_OK;
}

int mi_rtj_pipe_times(mi_rtj_pipe* q, float ms[MI_RTJ_NUM_KERNELS], int* launches) {
  if (!q || !ms) return MI_RTJ_ERR_ARG;
  pipe_drain_jobs(q);
  int total = 0;
  for (int k = 0; k < MI_RTJ_NUM_KERNELS; k++) ms[k] = 0.f;
  std::vector<mi_rtj_plan*> plans;
  for (auto& sl : q->slot) plans.push_back(sl.plan);
  for (mi_rtj_plan* gp : q->gplan) plans.push_back(gp);
  for (mi_rtj_plan* pl : plans) {  // (launches = k_decode launches = packets: a group's index kernels count once)
    float one[MI_RTJ_NUM_KERNELS];
    int n = 0;
    const int rc = mi_rtj_plan_times(pl, one, &n);
    if (rc != MI_RTJ_OK) return rc;
    for (int k = 0; k < MI_RTJ_NUM_KERNELS; k++) ms[k] += one[k];
    total += n;
  }
  if (launches) *launches = total;
  return MI_RTJ_OK;
}

int mi_rtj_pipe_peek_tag(const mi_rtj_pipe* q, uint64_t* tag) {
  if (!q || !tag || q->count == 0) return MI_RTJ_ERR_ARG;
  *tag = q->slot[q->head].tag;
  return MI_RTJ_OK;
}

int mi_rtj_pipe_flush(mi_rtj_pipe* q) {
  if (!q) return MI_RTJ_ERR_ARG;
  mi_rtj_ctx* c = q->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  // work in flight cannot be recalled; it is waited for and forgotten (a seek is rare, a frame takes microseconds)
  pipe_drain_jobs(q);
  // Packets copied in whose index group was not complete yet are decoded all the same before they are forgotten:
  // q->prev_pic — the picture of the packet submitted last — is then a picture that WAS decoded, and the first packet
  // after the resync takes its unchanged (0xFF) blocks from it, as the reference takes them from priv->frame (ADVICE r3:
  // they used to be dropped unindexed, and prev_pic pointed at a picture up to `depth` packets old or at zeros)
  if (q->nstaged > 0) (void)pipe_issue_staged(q);
  HIPCHK(c, hipStreamSynchronize(q->s_in));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (q->s_idx) HIPCHK(c, hipStreamSynchronize(q->s_idx));
  for (int i = 0; i < q->n_out; i++) HIPCHK(c, hipStreamSynchronize(q->s_out[i]));
  for (auto& sl : q->slot) {
    sl.out_state = 0;
    sl.staged = 0;
  }
  q->nstaged = 0;
  q->head = (q->head + q->count) % q->depth;
  q->count = 0;
  q->lent = -1;
  return MI_RTJ_OK;
}

int mi_rtj_synth_frames(mi_rtj_ctx* c, int w, int h, int first, int n, uint32_t seed, int amp, void* d_frames) {
  if (!c || !d_frames || w <= 0 || h <= 0 || (w & 15) || (h & 15) || n <= 0 || amp < 0)
    return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_synth_frames: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_synth, dim3(1024, n), dim3(256), 0, c->stream, (uint8_t*)d_frames, w, h, first, seed, amp);
  HIPCHK(c, hipGetLastError());
  return MI_RTJ_OK;
}

int mi_rtj_synth_frames_lcg(mi_rtj_ctx* c, int w, int h, int first, int n, uint32_t seed, int amp, void* d_frames) {
  if (!c || !d_frames || w <= 0 || h <= 0 || (w & 15) || (h & 15) || n <= 0 || amp < 0 || first < 0)
    return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_synth_frames_lcg: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_synth_lcg, dim3(4096), dim3(256), 0, c->stream, (uint8_t*)d_frames, w, h, first, n, seed, amp);
  HIPCHK(c, hipGetLastError());
  return MI_RTJ_OK;
}

size_t mi_rtj_encode_bound(int w, int h, int n, int align) {
  if (w <= 0 || h <= 0 || n <= 0 || align < 1) return 0;
  const size_t per = MI_RTJ_HEADER_SIZE + (size_t)(w / 16) * (h / 16) * 6 * 64;
  return (size_t)n * ((per + align - 1) / align * align) + align;
}

}  // extern "C" (reopened below)

namespace {
// shared body of the intra batch encoder and the in-order inter (skip-block) stream encoder
int encode_impl(mi_rtj_ctx* c, int w, int h, int Q, int key_rate, int lmask, int cmask, int n, const void* d_frames,
                void* d_stream, int align, uint64_t* pkt_offset, uint32_t* pkt_len) {
  if (!c || !d_frames || !d_stream || !pkt_offset || !pkt_len || w <= 0 || h <= 0 || (w & 15) || (h & 15) ||
      n <= 0 || align < 1 || (align & (align - 1)))
    return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_encode: bad argument");
  if (Q < 1) Q = 1;
  if (Q > 255) Q = 255;
  // RTjpeg_set_intra's clamps (lib/RTjpeg.c:2459-2468)
  key_rate = key_rate < 0 ? 0 : (key_rate > 255 ? 255 : key_rate);
  lmask = lmask < 0 ? 0 : (lmask > 16 ? 16 : lmask);
  cmask = cmask < 0 ? 0 : (cmask > 16 ? 16 : cmask);
  const bool inter = key_rate > 0;
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t nblk = (uint32_t)(w / 16) * (h / 16) * 6;
  const size_t fsz = (size_t)w * h * 3 / 2;
  // frames per pass: inter frames depend on each other, intra ones do not (256: 0.8 GB of block slots at 1080p)
  const int chunk = inter ? 1 : (int)std::max<size_t>(1, std::min<size_t>(256, ((size_t)1 << 30) / ((size_t)nblk * 64)));
  uint8_t *slots = nullptr, *lens = nullptr;
  uint32_t *offs = nullptr, *fbytes = nullptr;
  uint64_t *d_pktoff = nullptr, *d_alloff = nullptr;
  uint32_t* d_alllen = nullptr;
  int16_t* old = nullptr;
  int rc = MI_RTJ_OK;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(c->stream);
    if (slots) (void)hipFree(slots);
    if (lens) (void)hipFree(lens);
    if (offs) (void)hipFree(offs);
    if (fbytes) (void)hipFree(fbytes);
    if (d_pktoff) (void)hipFree(d_pktoff);
    if (d_alloff) (void)hipFree(d_alloff);
    if (d_alllen) (void)hipFree(d_alllen);
    if (old) (void)hipFree(old);
  };
#define ENC_CHK(call)                                                                                         \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      rc = fail(c, MI_RTJ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      cleanup();                                                                                              \
      return rc;                                                                                              \
    }                                                                                                         \
  } while (0)
  ENC_CHK(hipMalloc((void**)&slots, (size_t)chunk * nblk * 64));
  ENC_CHK(hipMalloc((void**)&lens, (size_t)chunk * nblk));
  ENC_CHK(hipMalloc((void**)&offs, (size_t)chunk * nblk * 4));
  ENC_CHK(hipMalloc((void**)&fbytes, (size_t)chunk * 4));
  ENC_CHK(hipMalloc((void**)&d_pktoff, (size_t)chunk * 8));
  if (inter) ENC_CHK(hipMalloc((void**)&old, (size_t)nblk * 64 * sizeof(int16_t)));
  // packet offsets and lengths of ALL frames stay on the device until the end: no host round trip between passes
  ENC_CHK(hipMalloc((void**)&d_alloff, (size_t)n * 8 + 8));
  ENC_CHK(hipMalloc((void**)&d_alllen, (size_t)n * 4));
  uint64_t* const d_cursor = d_alloff + n;
  ENC_CHK(hipMemsetAsync(d_cursor, 0, 8, c->stream));
  int key_count = 0;
  for (int f0 = 0; f0 < n; f0 += chunk) {
    const int m = n - f0 < chunk ? n - f0 : chunk;
    const size_t tot = (size_t)m * nblk;
    // RTjpeg_compress: the previous-block store is cleared whenever key_count is 0 (lib/RTjpeg.c:3504-3505)
    if (inter && key_count == 0) ENC_CHK(hipMemsetAsync(old, 0, (size_t)nblk * 64 * sizeof(int16_t), c->stream));
    if (inter) {
      hipLaunchKernelGGL(k_encode_blocks, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                         (const uint8_t*)d_frames + (size_t)f0 * fsz, w, h, m, c->d_lut + Q, slots, lens, old, lmask, cmask);
    } else {  // intra: a wave per 64 horizontally adjacent blocks (rtj_encode_kernels.h)
      const unsigned groups = (nblk / 6u + (unsigned)kMbPerGroup - 1u) / (unsigned)kMbPerGroup;
      hipLaunchKernelGGL(k_encode_wave, dim3(3u * groups, (unsigned)m), dim3(64), 0, c->stream,
                         (const uint8_t*)d_frames + (size_t)f0 * fsz, w, h, c->d_lut + Q, slots, lens);
    }
    hipLaunchKernelGGL(k_encode_scan, dim3(m), dim3(256), 0, c->stream, lens, nblk, offs, fbytes);
    hipLaunchKernelGGL(k_encode_place, dim3(1), dim3(256), 0, c->stream, fbytes, m, (uint32_t)align, d_cursor, d_alloff + f0,
                       d_alllen + f0, d_pktoff);
    hipLaunchKernelGGL(k_encode_pack, dim3((nblk + 255) / 256, m), dim3(256), 0, c->stream, slots, lens, offs,
                       d_pktoff, fbytes, nblk, w, h, Q, inter ? key_count : 0, (uint8_t*)d_stream);
    ENC_CHK(hipGetLastError());
    if (inter && ++key_count > key_rate) key_count = 0;  // lib/RTjpeg.c:3512-3514
  }
  // one copy of every packet's place and size at the end (the kernels of all passes are queued by now)
  ENC_CHK(hipMemcpyAsync(pkt_offset, d_alloff, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  ENC_CHK(hipMemcpyAsync(pkt_len, d_alllen, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  ENC_CHK(hipStreamSynchronize(c->stream));
#undef ENC_CHK
  cleanup();
  return MI_RTJ_OK;
}
}  // namespace

extern "C" {

int mi_rtj_encode_frames(mi_rtj_ctx* c, int w, int h, int Q, int n, const void* d_frames, void* d_stream, int align,
                         uint64_t* pkt_offset, uint32_t* pkt_len) {
  return encode_impl(c, w, h, Q, 0, 0, 0, n, d_frames, d_stream, align, pkt_offset, pkt_len);
}

int mi_rtj_encode_stream(mi_rtj_ctx* c, int w, int h, int Q, int key_rate, int lmask, int cmask, int n,
                         const void* d_frames, void* d_stream, int align, uint64_t* pkt_offset, uint32_t* pkt_len) {
  return encode_impl(c, w, h, Q, key_rate, lmask, cmask, n, d_frames, d_stream, align, pkt_offset, pkt_len);
}

int mi_rtj_yuv420_to_rgb(mi_rtj_ctx* c, int fmt, int w, int h, int n, const void* d_planes, size_t in_frame_stride,
                         void* d_rgb, size_t row_pitch, size_t out_frame_stride) {
  static const int bpp[5] = {4, 4, 3, 3, 2};
  if (!c || !d_planes || !d_rgb || fmt < 0 || fmt > 4 || w <= 0 || h <= 0 || (w & 15) || (h & 15) || n <= 0)
    return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_yuv420_to_rgb: bad argument");
  if (row_pitch < (size_t)w * bpp[fmt] || (row_pitch & 15) || ((uintptr_t)d_rgb & 15) || (out_frame_stride & 15) ||
      (in_frame_stride & 15) || ((uintptr_t)d_planes & 15))
    return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_yuv420_to_rgb: buffers, row pitch and frame strides must be 16-byte aligned");
  HIPCHK(c, hipSetDevice(c->device));
  const int tiles = (w / 8) * (h / 2);
  const dim3 grid((unsigned)((tiles + 255) / 256 < 2048 ? (tiles + 255) / 256 : 2048), (unsigned)n), block(256);
  const uint8_t* in = (const uint8_t*)d_planes;
  uint8_t* out = (uint8_t*)d_rgb;
  switch (fmt) {
    case kFmtRGB32: hipLaunchKernelGGL(k_yuv420_to_rgb<kFmtRGB32>, grid, block, 0, c->stream, in, in_frame_stride, out, row_pitch, out_frame_stride, w, h); break;
    case kFmtBGR32: hipLaunchKernelGGL(k_yuv420_to_rgb<kFmtBGR32>, grid, block, 0, c->stream, in, in_frame_stride, out, row_pitch, out_frame_stride, w, h); break;
    case kFmtRGB24: hipLaunchKernelGGL(k_yuv420_to_rgb<kFmtRGB24>, grid, block, 0, c->stream, in, in_frame_stride, out, row_pitch, out_frame_stride, w, h); break;
    case kFmtBGR24: hipLaunchKernelGGL(k_yuv420_to_rgb<kFmtBGR24>, grid, block, 0, c->stream, in, in_frame_stride, out, row_pitch, out_frame_stride, w, h); break;
    default:        hipLaunchKernelGGL(k_yuv420_to_rgb<kFmtRGB16>, grid, block, 0, c->stream, in, in_frame_stride, out, row_pitch, out_frame_stride, w, h); break;
  }
  HIPCHK(c, hipGetLastError());
  return MI_RTJ_OK;
}

int mi_rtj_copy_ceiling(mi_rtj_ctx* c, const void* d_src, void* d_dst, size_t bytes, int reps, double* gbs) {
  if (!c || !d_src || !d_dst || !gbs || bytes < 16 || (bytes & 15) || reps < 1 || ((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 15))
    return fail(c, MI_RTJ_ERR_ARG, "mi_rtj_copy_ceiling: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  const size_t n16 = bytes / 16;
  const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 256u * 32u);
  hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, c->stream, (const uint4*)d_src, (uint4*)d_dst, n16);  // warm-up
  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, c->stream, (const uint4*)d_src, (uint4*)d_dst, n16);
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbs = 2.0 * (double)bytes * reps / ((double)ms * 1e6);
  return MI_RTJ_OK;
}

int mi_rtj_get_tables(int Q, int32_t tables[128], int* lb8, int* cb8) {
  if (!tables || Q < 1 || Q > 255) return MI_RTJ_ERR_ARG;
  QTab t;
  build_qtab(Q, &t);
  memcpy(tables, t.liqt, sizeof t.liqt);
  memcpy(tables + 64, t.ciqt, sizeof t.ciqt);
  if (lb8) *lb8 = t.lb8;
  if (cb8) *cb8 = t.cb8;
  return MI_RTJ_OK;
}

}  // extern "C"
