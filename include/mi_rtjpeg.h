/*
 * mi_rtjpeg.h — C ABI of the MI355X (gfx950) RTjpeg frame-decode path.
 *
 * This is the drop-in boundary for gmerlin-avdecoder's in-tree RTjpeg decoder: plain C
 * types, no HIP or torch types.  The library behind it (libmi_rtjpeg.so) holds only
 * hand-written HIP kernels plus the host logic that launches them; there is no CPU decode
 * path — every entry point fails (and says why through mi_rtj_last_error) when no gfx950
 * device is usable.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * gmerlin-avdecoder tree).  INTEGRATION.md shows the replacement lib/video_rtjpeg.c that
 * binds them into the bgav_video_decoder_t table.
 */
#ifndef MI_RTJPEG_H
#define MI_RTJPEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_RTJ_HEADER_SIZE 12 /* RTJPEG_HEADER_SIZE, include/RTjpeg.h:142 */

enum {
  MI_RTJ_OK = 0,
  MI_RTJ_ERR_NO_DEVICE = -1, /* no HIP device / not gfx950 */
  MI_RTJ_ERR_HIP = -2,       /* a HIP runtime call failed (text in mi_rtj_last_error) */
  MI_RTJ_ERR_ARG = -3,       /* NULL / out-of-range argument */
  MI_RTJ_ERR_GEOMETRY = -4,  /* header width/height not positive multiples of 16: the reference's
                                macroblock loop (lib/RTjpeg.c:2701-2703) never terminates on these */
  MI_RTJ_ERR_NOMEM = -5
};

typedef struct mi_rtj_ctx mi_rtj_ctx;   /* one per decoder instance == one RTjpeg_t (include/RTjpeg.h:40-74) */
typedef struct mi_rtj_plan mi_rtj_plan; /* a batch of packets laid out for the device */

/* ---- device probe: what a .probe callback of bgav_video_decoder_t would ask
 *      (include/avdec_private.h:95; precedent lib/video_v4l2_m2m.c:43-60) ---- */
int mi_rtj_device_count(void); /* number of usable gfx950 devices, 0 when none */

/* ---- instance lifecycle ---- */
/* replaces RTjpeg_init (lib/RTjpeg.c:2495-2502) as called by init_rtjpeg (lib/video_rtjpeg.c:48).
 * device < 0 picks the current device (LOCAL_RANK / HIP_VISIBLE_DEVICES decide).  NULL on failure. */
mi_rtj_ctx *mi_rtj_create(int device);
/* replaces RTjpeg_close (lib/RTjpeg.c:2504-2508) as called by close_rtjpeg (lib/video_rtjpeg.c:98). */
void mi_rtj_destroy(mi_rtj_ctx *ctx);
/* Text of the last failure on this instance (ctx == NULL: last mi_rtj_create failure). */
const char *mi_rtj_last_error(const mi_rtj_ctx *ctx);

/* ---- one packet in, one frame out ----
 * replaces RTjpeg_decompress(rtj, p->buf.buf, priv->frame->planes) followed by
 * gavl_video_frame_copy(format, f, priv->frame)  (lib/video_rtjpeg.c:81-83; lib/RTjpeg.c:3565-3586).
 * The decoded picture persists on the device between calls exactly like priv->frame does, so
 * 0xFF "unchanged" blocks (lib/RTjpeg.c:2704) keep the previous frame's pixels.
 * dst[0..2]/dst_stride[0..2] describe the caller's Y, U, V planes (gavl_video_frame_t planes[]/
 * strides[]); the crop_w x crop_h top-left region (image_width x image_height) is copied.
 * dst == NULL decodes into the persistent frame only.  Returns MI_RTJ_OK or an error. */
int mi_rtj_decode(mi_rtj_ctx *ctx, const uint8_t *pkt, size_t len, uint8_t *const dst[3],
                  const int dst_stride[3], int crop_w, int crop_h);
/* The same decode for a decoder that owns its output frame (the reference's "nocopy" mode: the
 * decoder sets s->vframe in init and decode(s, NULL) leaves the picture there, lib/video.c:253-277,
 * 420-429).  The whole coded picture lands in pinned host memory owned by the instance; planes[]/
 * strides[] describe it (stride = coded width) and stay valid until the next decode call. */
int mi_rtj_decode_nocopy(mi_rtj_ctx *ctx, const uint8_t *pkt, size_t len, const uint8_t *planes[3],
                         int strides[3]);

/* ---- pipelined session: a frame-owning decoder with packets in flight ----
 * What a plugin instance that reads ahead uses instead of mi_rtj_decode (the one-packet calls above finish a picture
 * before they return: copy in, kernels and copy out of successive packets never overlap).  Replaces, for such an
 * instance, RTjpeg_decompress + gavl_video_frame_copy (lib/video_rtjpeg.c:81-82) with: packets go in in stream order
 * (their bytes are copied at once into pinned staging: a bgav packet is only valid until the next
 * bgav_stream_get_packet_read, lib/stream.c:538-601), pictures come out in the same order as pointers into pinned
 * host memory the session owns (the nocopy source of lib/video.c:420-441; lib/video_v4l2_m2m.c:43-131 is the
 * in-tree precedent of a decoder with buffers in flight).  Copy in, kernels and copy out of up to `depth` packets
 * run side by side on three streams; unchanged (0xFF) blocks keep the previous picture of the stream as in the
 * reference (k_decode fetches them from the predecessor's device picture).
 * With a coded size given, packets are indexed on the device and pictures copied out in groups (four of each from a
 * depth of 12 on, else two; the depth is rounded up to whole groups): a packet is copied in at once, its kernels are
 * queued when its group is complete or its picture is asked for, whichever comes first.
 *   depth            packets in flight plus the one picture on loan, 2..64
 *   coded_w, coded_h the stream's coded size (multiples of 16; 0 0 = take it from the packets): a packet whose header
 *                    says otherwise is refused before anything is allocated for it — the reference's frame has the
 *                    container's size whatever a packet claims (lib/video_rtjpeg.c:50-54)
 * One session per stream; not thread safe; the instance's header-driven state (mi_rtj_get_state) is shared with
 * the one-packet calls, which must not be mixed into a running session. */
typedef struct mi_rtj_pipe mi_rtj_pipe;
mi_rtj_pipe *mi_rtj_pipe_create(mi_rtj_ctx *ctx, int depth, int coded_w, int coded_h);
void mi_rtj_pipe_destroy(mi_rtj_pipe *pipe);
/* packets that can still be submitted before a picture has to be taken / packets in flight */
int mi_rtj_pipe_room(const mi_rtj_pipe *pipe);
int mi_rtj_pipe_pending(const mi_rtj_pipe *pipe);
/* Queue one packet (whole packet, header included).  `tag` comes back with its picture (the wrapper keeps pts,
 * duration and timecode under it: bgav_set_video_frame_from_packet, lib/video.c:861-871). */
int mi_rtj_pipe_submit(mi_rtj_pipe *pipe, const uint8_t *pkt, size_t len, uint64_t tag);
/* The oldest picture in flight: waits for it, then hands out the coded picture (planes[0..2], strides = width,
 * width/2) — valid until the next mi_rtj_pipe_next / _flush / _destroy.  planes == NULL: the picture is dropped
 * (bgav's frame skipping, lib/video_rtjpeg.c:75-79; .skipto). */
int mi_rtj_pipe_next(mi_rtj_pipe *pipe, const uint8_t *planes[3], int strides[3], int *w, int *h, uint64_t *tag);
/* tag of the oldest picture in flight without waiting for it (for .skipto) */
int mi_rtj_pipe_peek_tag(const mi_rtj_pipe *pipe, uint64_t *tag);
/* .resync (include/avdec_private.h:110, lib/video.c:561-562): forget everything in flight.  The stream's previous
 * picture stays, as priv->frame does in the reference. */
int mi_rtj_pipe_flush(mi_rtj_pipe *pipe);
/* Measurement only (bench.py): per-kernel device time of the packets a session decoded since profiling was switched on
 * (summed over its slots; indices as MI_RTJ_K_* below).  Switching flushes the session.  Profiling costs the
 * submitting thread two event records per kernel and packet: use a lap of its own, not a timed region. */
int mi_rtj_pipe_profile(mi_rtj_pipe *pipe, int enable);
int mi_rtj_pipe_times(mi_rtj_pipe *pipe, float ms[6], int *launches);

/* Geometry and effective quality the last mi_rtj_decode / plan used (RTjpeg_t width/height/Q). */
void mi_rtj_get_state(const mi_rtj_ctx *ctx, int *width, int *height, int *quality);

/* ---- device memory (HBM) owned by the instance; plain pointers ---- */
void *mi_rtj_dev_alloc(mi_rtj_ctx *ctx, size_t bytes); /* padded so kernels may over-read 64 B */
void mi_rtj_dev_free(mi_rtj_ctx *ctx, void *dptr);
int mi_rtj_h2d(mi_rtj_ctx *ctx, void *dptr, const void *src, size_t bytes);
int mi_rtj_d2h(mi_rtj_ctx *ctx, void *dst, const void *dptr, size_t bytes);
int mi_rtj_dev_memset(mi_rtj_ctx *ctx, void *dptr, int value, size_t bytes);
int mi_rtj_sync(mi_rtj_ctx *ctx); /* wait for everything queued on the instance's stream */

/* ---- batches of packets (independent frames or streams), device resident ----
 * A plan applies RTjpeg_decompress's header logic (size / quality change, lib/RTjpeg.c:3568-3579)
 * to n packets in order, starting from the instance's current state, and fixes where each
 * packet lives in a device stream buffer and where its planes go in a device output buffer
 * (Y at out_offset, then U, then V, contiguous, stride = width as lib/RTjpeg.c:2708,2741 write).
 * headers: n * 12 bytes (host).  pkt_offset/pkt_len: whole packets, header included. */
mi_rtj_plan *mi_rtj_plan_create(mi_rtj_ctx *ctx, int n, const uint8_t *headers,
                                const uint64_t *pkt_offset, const uint32_t *pkt_len,
                                const uint64_t *out_offset);
void mi_rtj_plan_destroy(mi_rtj_plan *plan);
/* Queue the whole hot path for the batch: block-offset index (RTjpeg_s2b's length semantics,
 * lib/RTjpeg.c:157-186) then dequant + IDCT + plane scatter (lib/RTjpeg.c:2209-2332, 2688-2749).
 * Asynchronous; pair with mi_rtj_sync.  Ordering: the transform always runs on the instance's stream, and
 * mi_rtj_sync waits for all of a launch.  Plans of some hundred to some thousand pictures build the index of launch
 * k + 1 on a second stream of their own while launch k is transformed; that stream is ordered behind everything this
 * library queued on the instance's stream before the call (mi_rtj_dev_memset; mi_rtj_h2d and the encoder entry points
 * are synchronous).  A caller that writes d_stream with work of its own must finish it before calling. */
int mi_rtj_plan_decode(mi_rtj_plan *plan, const void *d_stream, void *d_out);
/* Frames in the plan, total blocks, total algorithmic bytes (packet bytes read + plane bytes written). */
void mi_rtj_plan_info(const mi_rtj_plan *plan, int *n_frames, uint64_t *n_blocks,
                      uint64_t *bytes_in, uint64_t *bytes_out);
/* Per-kernel device time: when enabled, mi_rtj_plan_decode brackets each kernel with HIP events
 * on the launch stream.  mi_rtj_plan_times sums them since the last reset (after a sync), one
 * slot per kernel of the path; launches = number of mi_rtj_plan_decode calls summed. */
enum {
  MI_RTJ_K_SUMMARIZE = 0, /* k_index_summarize: per-chunk block-length tables + entry summaries */
  MI_RTJ_K_RESOLVE = 1,   /* k_index_resolve: chain the summaries per packet */
  MI_RTJ_K_EMIT = 2,      /* k_index_emit (or k_index_walk with MI_RTJ_INDEX=serial): block offsets */
  MI_RTJ_K_DECODE = 3,    /* k_decode: dequant + IDCT + plane scatter */
  MI_RTJ_K_SPEC_WALK = 4,   /* k_spec_walk: speculative index, one lane per stream chunk */
  MI_RTJ_K_SPEC_VERIFY = 5, /* k_spec_verify: proves or rejects it per packet, writes the proven block offsets
                             * (kernels 0-2 then only serve the rejected packets) */
  MI_RTJ_NUM_KERNELS = 6
};
void mi_rtj_plan_profile(mi_rtj_plan *plan, int enable);
int mi_rtj_plan_times(mi_rtj_plan *plan, float ms[MI_RTJ_NUM_KERNELS], int *launches);
/* While profiling: device time of every decode since profiling was switched on, first kernel's start to k_decode's
 * end, one value per decode (at most max_steps are written; *steps = how many there were).  For the median step
 * SURVEY.md section 8d asks for.  Synchronises the instance's stream. */
int mi_rtj_plan_step_times(mi_rtj_plan *plan, float *ms, int max_steps, int *steps);
/* After a decode: *walkers = stream chunks the speculative index covered (0: it was not used for this
 * plan — small batch, MI_RTJ_SPEC=0, or an A/B index mode), *proven = packets whose index its proof step
 * accepted (the others were indexed by the exact kernels), *repaired = chunks that were walked a second
 * time before that.  Synchronises the instance's stream. */
int mi_rtj_plan_spec_stats(mi_rtj_plan *plan, int *proven, long long *walkers, long long *repaired);
/* What the plan's device-side policy has decided for the NEXT decode: *lead_bytes = bytes each walker parses
 * before its chunk (the short or the long form; 0: the speculative index is not used for this plan),
 * *paused_launches = decodes left that go straight to the exact kernels.  Synchronises the instance's stream. */
int mi_rtj_plan_spec_lead(mi_rtj_plan *plan, int *lead_bytes, int *paused_launches);
/* Which form of the transform kernel the plan's device-side policy has chosen for the NEXT batch decode: *form = 0
 * k_decode_split (luma waves + chroma waves that pool three groups' busy blocks), 1 k_decode<true, false> (a wave takes
 * the three parts of its groups; what noisy content gets), -1 no batch launch yet / not a batch plan;
 * *classic_launches_left = decodes left before the split form is tried again; *parts_listed = group parts the last
 * decode left to k_decode_list.  Synchronises the instance's stream. */
int mi_rtj_plan_decode_form(mi_rtj_plan *plan, int *form, int *classic_launches_left, long long *parts_listed);
/* 1 if the plan's last decode built its block index on the plan's own stream, next to the transform of the decode before
 * it (plans of 129 .. 8191 pictures of 1080p always do; longer ones while the host has seen the decode policy in its
 * classic mode, i.e. on noisy content; MI_RTJ_OVERLAP=0 / 1 / 2: never / always / by that rule whatever the size), else 0.
 * Does not synchronise. */
int mi_rtj_plan_overlapped(const mi_rtj_plan *plan);
/* Test hook: copy the plan's block-start index (relative to each packet's first data byte,
 * nblocks+1 entries per frame, frames back to back) to the host after a decode. */
int mi_rtj_plan_read_index(mi_rtj_plan *plan, uint32_t *dst, size_t max_entries);

/* ---- stream generator (RTjpeg_compress, lib/RTjpeg.c:3488-3524, intra only) ----
 * Synthetic frames (gradient + hashed noise, the tests hold a numpy twin):
 * n frames numbered first_frame.., each 1.5*w*h bytes, back to back in d_frames. */
int mi_rtj_synth_frames(mi_rtj_ctx *ctx, int w, int h, int first_frame, int n, uint32_t seed,
                        int amp, void *d_frames);
/* The same pictures with the noise generator of SURVEY.md section 8d / BASELINE.md section 2: one linear congruential
 * sequence s <- s * 1664525 + 1013904223 from `seed` (12345 there), one draw per sample in stream order (frame 0: Y row
 * major, U, V; frame 1; ...), noise = ((s >> 8) mod (2a + 1)) - a.  first_frame continues the sequence where frame
 * first_frame begins, so that ranks and passes make disjoint parts of ONE stream.  tests/rtjlib.py holds the numpy twin. */
int mi_rtj_synth_frames_lcg(mi_rtj_ctx *ctx, int w, int h, int first_frame, int n, uint32_t seed,
                            int amp, void *d_frames);
/* Encode n frames (w x h, contiguous planes) at quality Q into d_stream, packets back to back
 * (each start aligned to `align` bytes, a power of two >= 1).  Host arrays pkt_offset/pkt_len
 * (n entries) receive the layout.  d_stream must hold mi_rtj_encode_bound(w,h,n,align) bytes. */
size_t mi_rtj_encode_bound(int w, int h, int n, int align);
int mi_rtj_encode_frames(mi_rtj_ctx *ctx, int w, int h, int Q, int n, const void *d_frames,
                         void *d_stream, int align, uint64_t *pkt_offset, uint32_t *pkt_len);

/* The same for ONE stream whose frames are encoded in order with "unchanged block" detection
 * (RTjpeg_set_intra + RTjpeg_compress with key_rate > 0, lib/RTjpeg.c:2455-2488, 2841-2921, 3500-3514):
 * blocks within lmask/cmask of the previous coded block become the byte 0xFF.  Such packets must be
 * decoded in order by one instance (mi_rtj_decode). */
int mi_rtj_encode_stream(mi_rtj_ctx *ctx, int w, int h, int Q, int key_rate, int lmask, int cmask, int n,
                         const void *d_frames, void *d_stream, int align, uint64_t *pkt_offset,
                         uint32_t *pkt_len);

/* ---- colour stage: RTjpeg_yuv420rgb32 / bgr32 / rgb24 / bgr24 / rgb16
 * (lib/RTjpeg.c:3123-3475), device resident.  n frames of contiguous Y,U,V planes (in_frame_stride
 * bytes apart) to packed pixels: rows row_pitch bytes apart (the reference takes a rows[] pointer
 * array), frames out_frame_stride bytes apart.  Like the reference, the 32-bit formats leave the
 * fourth byte of every pixel as it was.  Asynchronous on the instance's stream. */
enum { MI_RTJ_RGB32 = 0, MI_RTJ_BGR32 = 1, MI_RTJ_RGB24 = 2, MI_RTJ_BGR24 = 3, MI_RTJ_RGB16 = 4 };
int mi_rtj_yuv420_to_rgb(mi_rtj_ctx *ctx, int fmt, int w, int h, int n, const void *d_planes,
                         size_t in_frame_stride, void *d_rgb, size_t row_pitch, size_t out_frame_stride);

/* Measurement aid (SURVEY.md §8d "achievable-copy ceiling"): copies `bytes` (a multiple of 16) from d_src to
 * d_dst `reps` times with a plain 16-byte-per-lane grid-stride kernel and returns the sustained
 * read+write rate in GB/s in *gbs — the HBM rate a pure streaming kernel reaches on this device, the
 * second yardstick next to the nominal 8 TB/s.  Synchronous. */
int mi_rtj_copy_ceiling(mi_rtj_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, int reps, double *gbs);

/* Dequantiser tables the device uses for quality Q (1..255): 64 luma + 64 chroma entries in
 * natural order and the lb8/cb8 counts — the values RTjpeg_get_tables returns
 * (lib/RTjpeg.c:2371-2378) after RTjpeg_set_quality. */
int mi_rtj_get_tables(int Q, int32_t tables[128], int *lb8, int *cb8);

#ifdef __cplusplus
}
#endif
#endif
