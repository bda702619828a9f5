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

uint16_t mi_dv_audio_12to16(uint16_t sample) {
  uint16_t shift;
  sample = (uint16_t)(sample < 0x800 ? sample : sample | 0xf000);
  shift = (uint16_t)((sample & 0xf00) >> 8);
  if (shift < 0x2 || shift > 0xd) return sample;
  if (shift < 0x8) {
    shift--;
    return (uint16_t)((sample - 256 * shift) << shift);
  }
  shift = (uint16_t)(0xe - shift);
  return (uint16_t)(((sample + (256 * shift + 1)) << shift) - 1);
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

int mi_dv_extract_audio(const mi_dv_profile *p, const uint8_t *frame, uint8_t *ppcm[4]) {
  const uint8_t *as = fixed_pack(frame, PACK_AUDIO_SOURCE);
  if (!as) return 0; /* no audio */
  const int smpls = as[1] & 0x3f;       /* samples in this frame beyond the minimum */
  const int freq = (as[4] >> 3) & 0x07; /* 0: 48 kHz, 1: 44.1 kHz, 2: 32 kHz */
  const int quant = as[4] & 0x07;       /* 0: 16 bit linear, 1: 12 bit non-linear */
  if (quant > 1 || freq > 2) return -1;
  const int size = (p->audio_min_samples[freq] + smpls) * 4; /* 2 channels x 2 bytes */
  const int half_ch = p->difseg_size / 2;
  /* 720p frames come in halves: even ones carry channel pairs 0,1, odd ones 2,3 */
  int ipcm = (p->height == 720 && ((frame[1] >> 2) & 0x3) == 0) ? 2 : 0;
  uint8_t *pcm = ppcm[ipcm++];

  for (int chan = 0; chan < p->n_difchan; chan++) {
    for (int i = 0; i < p->difseg_size; i++) {
      frame += 6 * 80; /* header, subcode and VAUX blocks of the sequence */
      if (quant == 1 && i == half_ch) { /* second stereo pair (12-bit mode only) */
        pcm = ppcm[ipcm++];
        if (!pcm) break;
      }
      for (int j = 0; j < 9; j++) { /* nine audio DIF blocks per sequence */
        for (int d = 8; d < 80; d += 2) {
          if (quant == 0) {
            const int of = p->audio_shuffle[i][j] + (d - 8) / 2 * p->audio_stride;
            if (of * 2 >= size) continue;
            pcm[of * 2] = frame[d + 1]; /* DV PCM is big endian */
            pcm[of * 2 + 1] = frame[d];
            if (pcm[of * 2 + 1] == 0x80 && pcm[of * 2] == 0x00) pcm[of * 2 + 1] = 0; /* error code -> silence */
          } else {
            uint16_t lc = (uint16_t)(((uint16_t)frame[d] << 4) | ((uint16_t)frame[d + 2] >> 4));
            uint16_t rc = (uint16_t)(((uint16_t)frame[d + 1] << 4) | ((uint16_t)frame[d + 2] & 0x0f));
            lc = lc == 0x800 ? 0 : mi_dv_audio_12to16(lc);
            rc = rc == 0x800 ? 0 : mi_dv_audio_12to16(rc);
            int of = p->audio_shuffle[i % half_ch][j] + (d - 8) / 3 * p->audio_stride;
            if (of * 2 >= size) continue;
            pcm[of * 2] = (uint8_t)(lc & 0xff);
            pcm[of * 2 + 1] = (uint8_t)(lc >> 8);
            of = p->audio_shuffle[i % half_ch + half_ch][j] + (d - 8) / 3 * p->audio_stride;
            pcm[of * 2] = (uint8_t)(rc & 0xff);
            pcm[of * 2 + 1] = (uint8_t)(rc >> 8);
            ++d; /* three bytes carry two 12-bit samples */
          }
        }
        frame += 16 * 80; /* 15 video DIF blocks + 1 audio DIF block */
      }
    }
    pcm = ppcm[ipcm++]; /* next pair (50 and 100 Mbps only) */
    if (!pcm) break;
  }
  return size / 4;
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
