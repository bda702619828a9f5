/*
 * dvframe_host.c — DV DIF-frame handling on the host; see include/mi_dvframe.h.
 * Byte shuffling only: stays C on the CPU by design.  PARITY UNPINNED (header).
 */
#include "mi_dvframe.h"

#include <string.h>

/* PCM shuffling tables: DIF sequence x AV sequence -> first sample slot (lib/dvframe.c:76-104).
 * Both follow one rule — slot = (seq*6 + blk*step + chan) with per-AV-sequence offsets — but they
 * are normative data of IEC 61834, so they are tabulated. */
static const uint16_t shuffle525[10][9] = {
    {0, 30, 60, 20, 50, 80, 10, 40, 70},  {6, 36, 66, 26, 56, 86, 16, 46, 76},
    {12, 42, 72, 2, 32, 62, 22, 52, 82},  {18, 48, 78, 8, 38, 68, 28, 58, 88},
    {24, 54, 84, 14, 44, 74, 4, 34, 64},  {1, 31, 61, 21, 51, 81, 11, 41, 71},
    {7, 37, 67, 27, 57, 87, 17, 47, 77},  {13, 43, 73, 3, 33, 63, 23, 53, 83},
    {19, 49, 79, 9, 39, 69, 29, 59, 89},  {25, 55, 85, 15, 45, 75, 5, 35, 65}};
static const uint16_t shuffle625[12][9] = {
    {0, 36, 72, 26, 62, 98, 16, 52, 88},   {6, 42, 78, 32, 68, 104, 22, 58, 94},
    {12, 48, 84, 2, 38, 74, 28, 64, 100},  {18, 54, 90, 8, 44, 80, 34, 70, 106},
    {24, 60, 96, 14, 50, 86, 4, 40, 76},   {30, 66, 102, 20, 56, 92, 10, 46, 82},
    {1, 37, 73, 27, 63, 99, 17, 53, 89},   {7, 43, 79, 33, 69, 105, 23, 59, 95},
    {13, 49, 85, 3, 39, 75, 29, 65, 101},  {19, 55, 91, 9, 45, 81, 35, 71, 107},
    {25, 61, 97, 15, 51, 87, 5, 41, 77},   {31, 67, 103, 21, 57, 93, 11, 47, 83}};

#define NTSC_AUDIO 90, {1580, 1452, 1053}, shuffle525
#define PAL_AUDIO 108, {1896, 1742, 1264}, shuffle625
#define SAR_NTSC {{10, 11}, {40, 33}}
#define SAR_PAL {{59, 54}, {118, 81}}
#define SAR_SQ {{1, 1}, {1, 1}}

/* lib/dvframe.c:106-296, in the reference's order (index 2 is the 625/50 4:1:1 special case) */
static const mi_dv_profile profiles[] = {
    {0, 0x00, 120000, 10, 1, 30000, 1001, 30, 720, 480, SAR_NTSC, MI_DV_PIX_411, 6, NTSC_AUDIO},   /* DV NTSC */
    {1, 0x00, 144000, 12, 1, 25, 1, 25, 720, 576, SAR_PAL, MI_DV_PIX_420, 6, PAL_AUDIO},           /* DV PAL */
    {1, 0x00, 144000, 12, 1, 25, 1, 25, 720, 576, SAR_PAL, MI_DV_PIX_411, 6, PAL_AUDIO},           /* DVCPRO PAL */
    {0, 0x04, 240000, 10, 2, 30000, 1001, 30, 720, 480, SAR_NTSC, MI_DV_PIX_422, 6, NTSC_AUDIO},   /* DVCPRO50 NTSC */
    {1, 0x04, 288000, 12, 2, 25, 1, 25, 720, 576, SAR_PAL, MI_DV_PIX_422, 6, PAL_AUDIO},           /* DVCPRO50 PAL */
    {0, 0x14, 480000, 10, 4, 30000, 1001, 30, 1280, 1080, SAR_SQ, MI_DV_PIX_422, 8, NTSC_AUDIO},   /* 1080i60 */
    {1, 0x14, 576000, 12, 4, 25, 1, 25, 1440, 1080, SAR_SQ, MI_DV_PIX_422, 8, PAL_AUDIO},          /* 1080i50 */
    {0, 0x18, 240000, 10, 2, 60000, 1001, 60, 960, 720, SAR_SQ, MI_DV_PIX_422, 8, NTSC_AUDIO},     /* 720p60 */
    {1, 0x18, 288000, 12, 2, 50, 1, 50, 960, 720, SAR_SQ, MI_DV_PIX_422, 8, NTSC_AUDIO},           /* 720p50: 525 audio layout in the reference */
};

int mi_dv_num_profiles(void) { return (int)(sizeof profiles / sizeof profiles[0]); }
const mi_dv_profile *mi_dv_profile_at(int i) { return i >= 0 && i < mi_dv_num_profiles() ? &profiles[i] : NULL; }

const mi_dv_profile *mi_dv_frame_profile(const uint8_t *frame) {
  const int dsf = (frame[3] & 0x80) >> 7;
  const int stype = frame[80 * 5 + 48 + 3] & 0x1f;
  /* 576i50 25 Mbps 4:1:1 is told apart by the APT bits of the header */
  if (dsf == 1 && stype == 0 && (frame[5] & 0x07)) return &profiles[2];
  for (int i = 0; i < mi_dv_num_profiles(); i++)
    if (profiles[i].dsf == dsf && profiles[i].video_stype == stype) return &profiles[i];
  return NULL;
}

/* packs live at fixed offsets; anything else is "not present" (lib/dvframe.c:333-356) */
enum { PACK_TIMECODE = 0x13, PACK_AUDIO_SOURCE = 0x50, PACK_AUDIO_CONTROL = 0x51, PACK_VIDEO_CONTROL = 0x61,
       PACK_VIDEO_RECDATE = 0x62, PACK_VIDEO_RECTIME = 0x63 };

static const uint8_t *fixed_pack(const uint8_t *frame, int id) {
  int offs;
  switch (id) {
    case PACK_AUDIO_SOURCE: offs = 80 * 6 + 80 * 16 * 3 + 3; break;
    case PACK_AUDIO_CONTROL: offs = 80 * 6 + 80 * 16 * 4 + 3; break;
    case PACK_VIDEO_CONTROL: offs = 80 * 5 + 48 + 5; break;
    default: return NULL;
  }
  return frame[offs] == id ? frame + offs : NULL;
}

void mi_dv_pixel_aspect(const mi_dv_profile *p, const uint8_t *frame, int *num, int *den) {
  const uint8_t *vsc = fixed_pack(frame, PACK_VIDEO_CONTROL);
  const int apt = frame[4] & 0x07;
  const int wide = vsc && ((vsc[2] & 0x07) == 0x02 || (!apt && (vsc[2] & 0x07) == 0x07));
  *num = p->sar[wide][0];
  *den = p->sar[wide][1];
}

int mi_dv_video_packet(const mi_dv_profile *p, const uint8_t *frame, uint8_t *out, int *keyframe) {
  memcpy(out, frame, (size_t)p->frame_size);
  if (keyframe) *keyframe = 1;
  return p->frame_size;
}

/* 12-bit non-linear PCM to 16 bits (IEC 61834-2, the expansion dv_audio_12to16 implements, lib/dvframe.c:521-543).
 * Written from the companding law rather than from the reference's arithmetic: magnitudes are coded in eight
 * segments of 256 codes; segments 0 and 1 are linear, segment k >= 2 covers a range 2^(k-1) times as wide, i.e.
 * value = (code - 256 (k - 1)) << (k - 1).  Negative samples are the one's complement of the positive law
 * (code -> ~code, value -> ~value), which is what makes the two halves of the reference's formula one rule. */
static unsigned expand_magnitude(unsigned m) { /* m: 0 .. 0x7ff */
  const unsigned k = m >> 8;
  return k < 2 ? m : (m - 256u * (k - 1u)) << (k - 1u);
}

uint16_t mi_dv_audio_12to16(uint16_t sample) {
  const unsigned s = sample & 0xfffu;
  if (s & 0x800u) return (uint16_t)~expand_magnitude(~s & 0x7ffu);
  return (uint16_t)expand_magnitude(s);
}

static const int audio_hz[3] = {48000, 44100, 32000};

int mi_dv_audio_format(const mi_dv_profile *p, const uint8_t *frame, int *samplerate, int *channel_pairs,
                       int *max_samples_per_frame) {
  const uint8_t *as = fixed_pack(frame, PACK_AUDIO_SOURCE);
  if (!as || !p) return 0;
  const int freq = (as[4] >> 3) & 0x07, stype = as[3] & 0x1f, quant = as[4] & 0x07;
  if (freq > 2) return 0;
  if (samplerate) *samplerate = audio_hz[freq];
  if (channel_pairs) *channel_pairs = stype == 3 ? 4 : ((stype == 2 || (quant && freq == 2)) ? 2 : 1);
  if (max_samples_per_frame) *max_samples_per_frame = p->audio_min_samples[freq] + 0x3f;
  return 1;
}

/* ---- audio de-shuffle (what dv_extract_audio does, lib/dvframe.c:545-628), written from the DIF layout ----
 * A frame is n_difchan channels of difseg_size DIF sequences of 150 DIF blocks of 80 bytes.  Within a sequence,
 * block 0 is the header, 1-2 subcode, 3-5 VAUX, and then nine times (1 audio block, 15 video blocks): audio block j
 * of sequence q of channel c is DIF block (c * difseg_size + q) * 150 + 6 + 16 j.  An audio block is 3 bytes ID,
 * a 5-byte AAUX pack and 72 bytes of samples:
 *   16-bit mode  36 big-endian samples; sample n goes to slot shuffle[q][j] + n * stride of the channel's pair
 *   12-bit mode  24 byte triples (left high 8, right high 8, low nibbles of both); the sequences of a channel's
 *                first half fill one stereo pair, those of its second half the next one; the left sample of triple
 *                n goes to slot shuffle[q mod half][j] + n * stride, the right one to the slot the sequence `half`
 *                further on would use
 * A slot is a 16-bit little-endian sample in the pair's buffer. */
enum { DIF_BLOCK = 80, DIF_SEQ_BLOCKS = 150, AUDIO_FIRST = 6, AUDIO_EVERY = 16, AUDIO_PER_SEQ = 9, AUDIO_PAYLOAD = 8 };

static void put_le16(uint8_t *pcm, int slot, unsigned v) {
  pcm[2 * slot] = (uint8_t)(v & 0xffu);
  pcm[2 * slot + 1] = (uint8_t)(v >> 8);
}

int mi_dv_extract_audio(const mi_dv_profile *p, const uint8_t *frame, uint8_t *ppcm[4]) {
  const uint8_t *as = fixed_pack(frame, PACK_AUDIO_SOURCE);
  if (!as) return 0; /* no audio */
  const int extra = as[1] & 0x3f;       /* samples in this frame beyond the profile's minimum */
  const int freq = (as[4] >> 3) & 0x07; /* 0: 48 kHz, 1: 44.1 kHz, 2: 32 kHz */
  const int twelve = (as[4] & 0x07) == 1;
  if ((as[4] & 0x07) > 1 || freq > 2) return -1;
  const int slots = (p->audio_min_samples[freq] + extra) * 2; /* 16-bit slots per stereo pair buffer */
  const int half = p->difseg_size / 2;
  /* 720p frames come in halves: the first one carries channel pairs 2 and 3 (as the reference has it) */
  const int first_pair = (p->height == 720 && ((frame[1] >> 2) & 0x3) == 0) ? 2 : 0;

  for (int c = 0; c < p->n_difchan; c++) {
    for (int q = 0; q < p->difseg_size; q++) {
      const int pair = first_pair + (twelve ? 2 * c + (q >= half) : c);
      uint8_t *pcm = pair < 4 ? ppcm[pair] : NULL;
      if (!pcm) return slots / 2; /* the caller wants no more pairs than it gave buffers for */
      for (int j = 0; j < AUDIO_PER_SEQ; j++) {
        const uint8_t *blk =
            frame + ((size_t)(c * p->difseg_size + q) * DIF_SEQ_BLOCKS + AUDIO_FIRST + AUDIO_EVERY * j) * DIF_BLOCK + AUDIO_PAYLOAD;
        if (!twelve) {
          const int base = p->audio_shuffle[q][j];
          for (int n = 0; n < 36; n++) {
            const int slot = base + n * p->audio_stride;
            unsigned v = (unsigned)blk[2 * n] << 8 | blk[2 * n + 1];
            if (slot >= slots) continue;
            if (v == 0x8000u) v = 0; /* the format's "no sample" code plays as silence */
            put_le16(pcm, slot, v);
          }
        } else {
          const int base_l = p->audio_shuffle[q % half][j], base_r = p->audio_shuffle[q % half + half][j];
          for (int n = 0; n < 24; n++) {
            const uint8_t *t = blk + 3 * n;
            const unsigned l = (unsigned)t[0] << 4 | t[2] >> 4, r = (unsigned)t[1] << 4 | (t[2] & 0x0fu);
            const int slot_l = base_l + n * p->audio_stride, slot_r = base_r + n * p->audio_stride;
            if (slot_l >= slots) continue; /* the reference tests the left slot only; the right one lies beside it */
            put_le16(pcm, slot_l, l == 0x800u ? 0u : mi_dv_audio_12to16((uint16_t)l));
            put_le16(pcm, slot_r, r == 0x800u ? 0u : mi_dv_audio_12to16((uint16_t)r));
          }
        }
      }
    }
  }
  return slots / 2;
}

int mi_dv_ssyb_pack(const mi_dv_profile *p, const uint8_t *frame, int pack_id, uint8_t pack[5]) {
  /* 150 DIF blocks of 80 bytes per sequence; subcode blocks are blocks 1 and 2; block and packet
   * have 3-byte headers; six 8-byte packets per block */
  for (int i = 0; i < p->difseg_size; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 6; k++) {
        const uint8_t *s = frame + i * 150 * 80 + 80 + j * 80 + 3 + k * 8 + 3;
        if (s[0] == pack_id) {
          memcpy(pack, s, 5);
          return 1;
        }
      }
  return 0;
}

static int bcd(int v, int hi_mask) { return (v & 0xf) + 10 * ((v >> 4) & hi_mask); }

int mi_dv_date(const mi_dv_profile *p, const uint8_t *frame, int *year, int *month, int *day) {
  uint8_t pk[5];
  if (!mi_dv_ssyb_pack(p, frame, PACK_VIDEO_RECDATE, pk)) return 0;
  if (year) {
    *year = bcd(pk[4], 0xf);
    *year += *year < 25 ? 2000 : 1900;
  }
  if (month) *month = bcd(pk[3], 0x1);
  if (day) *day = bcd(pk[2], 0x3);
  return 1;
}

int mi_dv_time(const mi_dv_profile *p, const uint8_t *frame, int *hour, int *minute, int *second) {
  uint8_t pk[5];
  if (!mi_dv_ssyb_pack(p, frame, PACK_VIDEO_RECTIME, pk)) return 0;
  if (hour) *hour = bcd(pk[4], 0x3);
  if (minute) *minute = bcd(pk[3], 0x7);
  if (second) *second = bcd(pk[2], 0x7);
  return 1;
}

int mi_dv_timecode(const mi_dv_profile *p, const uint8_t *frame, int *hour, int *minute, int *second, int *fr) {
  uint8_t pk[5];
  if (!mi_dv_ssyb_pack(p, frame, PACK_TIMECODE, pk)) return 0;
  if (fr) *fr = bcd(pk[1], 0x3);
  if (second) *second = bcd(pk[2], 0x7);
  if (minute) *minute = bcd(pk[3], 0x7);
  if (hour) *hour = bcd(pk[4], 0x3);
  return 1;
}
