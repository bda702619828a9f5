// mi_dv.hip — C ABI (include/mi_dv.h) of the MI355X DV25 525/60 decoder.  No CPU path: without a gfx950 device
// every call fails with a message.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi_dv.h"
#include "dv_common.h"
#include "dv_decode_kernels.h"

using namespace midv;

static_assert(MI_DV_FRAME_BYTES == kFrameBytes && MI_DV_PICTURE_BYTES == kPicBytes, "header and kernels agree");

namespace {
std::mutex g_mu;
std::string g_err;
}  // namespace

struct mi_dv_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  Tables* d_tab = nullptr;
  // one pair of events around every launch since the last mi_dv_kernel_times (recycled there)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
  // the one-frame path's buffers
  uint8_t* d_frame = nullptr;
  uint8_t* d_pic = nullptr;
  uint8_t* h_pic = nullptr;  // pinned
  std::string err;
};

namespace {
int fail(mi_dv_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  else {
    std::lock_guard<std::mutex> l(g_mu);
    g_err = buf;
  }
  return code;
}
#define DVCHK(c, call)                                                                                              \
  do {                                                                                                              \
    hipError_t e_ = (call);                                                                                         \
    if (e_ != hipSuccess)                                                                                           \
      return fail((c), MI_DV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);   \
  } while (0)

bool is_gfx950(int dev) {
  hipDeviceProp_t p;
  return hipGetDeviceProperties(&p, dev) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0;
}
}  // namespace

extern "C" {

int mi_dv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; i++) ok += is_gfx950(i);
  return ok;
}

const char* mi_dv_last_error(const mi_dv_ctx* c) {
  if (c) return c->err.c_str();
  std::lock_guard<std::mutex> l(g_mu);
  return g_err.c_str();
}

mi_dv_ctx* mi_dv_create(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    fail(nullptr, MI_DV_ERR_HIP, "no HIP device: the DV decoder has no CPU path");
    return nullptr;
  }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (device >= n || !is_gfx950(device)) {
    fail(nullptr, MI_DV_ERR_ARG, "device %d is not a gfx950 (MI355X) device", device);
    return nullptr;
  }
  Tables t;
  if (!build_tables(&t)) {
    fail(nullptr, MI_DV_ERR_ARG, "internal: the variable-length code's tables are inconsistent");
    return nullptr;
  }
  mi_dv_ctx* c = new mi_dv_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&c->d_tab, sizeof(Tables)) != hipSuccess ||
      hipMemcpy(c->d_tab, &t, sizeof t, hipMemcpyHostToDevice) != hipSuccess) {
    fail(nullptr, MI_DV_ERR_HIP, "cannot set up device %d: %s", device, hipGetErrorString(hipGetLastError()));
    mi_dv_destroy(c);
    return nullptr;
  }
  return c;
}

void mi_dv_destroy(mi_dv_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->d_tab) (void)hipFree(c->d_tab);
  if (c->d_frame) (void)hipFree(c->d_frame);
  if (c->d_pic) (void)hipFree(c->d_pic);
  if (c->h_pic) (void)hipHostFree(c->h_pic);
  for (auto* v : {&c->ev_used, &c->ev_free})
    for (auto& e : *v) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void* mi_dv_dev_alloc(mi_dv_ctx* c, size_t bytes) {
  if (!c) return nullptr;
  void* d = nullptr;
  if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) {
    fail(c, MI_DV_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return d;
}
void mi_dv_dev_free(mi_dv_ctx* c, void* d) {
  if (!c || !d) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(d);
}
int mi_dv_h2d(mi_dv_ctx* c, void* d, const void* h, size_t n) {
  if (!c || (!d && n) || (!h && n)) return fail(c, MI_DV_ERR_ARG, "mi_dv_h2d: NULL argument");
  DVCHK(c, hipSetDevice(c->device));
  DVCHK(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream));
  DVCHK(c, hipStreamSynchronize(c->stream));
  return MI_DV_OK;
}
int mi_dv_d2h(mi_dv_ctx* c, void* h, const void* d, size_t n) {
  if (!c || (!d && n) || (!h && n)) return fail(c, MI_DV_ERR_ARG, "mi_dv_d2h: NULL argument");
  DVCHK(c, hipSetDevice(c->device));
  DVCHK(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
  DVCHK(c, hipStreamSynchronize(c->stream));
  return MI_DV_OK;
}
int mi_dv_sync(mi_dv_ctx* c) {
  if (!c) return MI_DV_ERR_ARG;
  DVCHK(c, hipSetDevice(c->device));
  DVCHK(c, hipStreamSynchronize(c->stream));
  return MI_DV_OK;
}

#ifdef MIDV_DEBUG
static void* g_dbg = nullptr;
extern "C" void mi_dv_debug_buffer(void* d) { g_dbg = d; }
#endif
int mi_dv_decode_batch(mi_dv_ctx* c, const void* d_frames, int n, void* d_pics) {
  if (!c || !d_frames || !d_pics || n <= 0) return fail(c, MI_DV_ERR_ARG, "mi_dv_decode_batch: bad argument");
  if (n > 65535) return fail(c, MI_DV_ERR_ARG, "mi_dv_decode_batch: %d frames; at most 65535 per call (split the batch)", n);
  if (((uintptr_t)d_frames & 3u) || ((uintptr_t)d_pics & 7u))
    return fail(c, MI_DV_ERR_ARG, "mi_dv_decode_batch: d_frames must be 4-byte and d_pics 8-byte aligned");
  DVCHK(c, hipSetDevice(c->device));
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!c->ev_free.empty()) {
    ev = c->ev_free.back();
    c->ev_free.pop_back();
  } else {
    DVCHK(c, hipEventCreate(&ev.first));
    DVCHK(c, hipEventCreate(&ev.second));
  }
  c->ev_used.push_back(ev);
  if (c->ev_used.size() > 4096) {  // nobody asks for times: keep the newest
    c->ev_free.push_back(c->ev_used.front());
    c->ev_used.erase(c->ev_used.begin());
  }
  DVCHK(c, hipEventRecord(ev.first, c->stream));
  hipLaunchKernelGGL(k_dv_decode, dim3(kDvGridX, (unsigned)n), dim3(64 * kDvWaves), 0, c->stream, (const uint8_t*)d_frames,
                     (uint8_t*)d_pics, c->d_tab
#ifdef MIDV_DEBUG
                     , (int16_t*)g_dbg
#endif
  );
  DVCHK(c, hipGetLastError());
  DVCHK(c, hipEventRecord(ev.second, c->stream));
  return MI_DV_OK;
}

int mi_dv_kernel_times(mi_dv_ctx* c, float* total_ms, int* launches) {
  if (!c || !total_ms || !launches) return MI_DV_ERR_ARG;
  DVCHK(c, hipSetDevice(c->device));
  DVCHK(c, hipStreamSynchronize(c->stream));
  float sum = 0.f;
  for (auto& e : c->ev_used) {
    float ms = 0.f;
    DVCHK(c, hipEventElapsedTime(&ms, e.first, e.second));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (int)c->ev_used.size();
  for (auto& e : c->ev_used) c->ev_free.push_back(e);
  c->ev_used.clear();
  return MI_DV_OK;
}

size_t mi_dv_copy_tables(void* out, size_t cap) {
  Tables t;
  if (!build_tables(&t)) return 0;
  if (out && cap >= sizeof t) memcpy(out, &t, sizeof t);
  return sizeof t;
}

int mi_dv_decode_frame(mi_dv_ctx* c, const uint8_t* frame, size_t len, uint8_t* const planes[3], const int strides[3]) {
  if (!c || !frame || !planes || !strides || !planes[0] || !planes[1] || !planes[2])
    return fail(c, MI_DV_ERR_ARG, "mi_dv_decode_frame: NULL argument");
  if (len < (size_t)kFrameBytes) return fail(c, MI_DV_ERR_FORMAT, "DIF frame of %zu bytes: 525/60 frames have %d", len, kFrameBytes);
  // dv_frame_profile (lib/dvframe.c:298-316): DSF = byte 3 bit 7, stype = byte 80*5+48+3 & 0x1f; only 525/60 25 Mbit/s here
  if ((frame[3] & 0x80) || (frame[80 * 5 + 48 + 3] & 0x1f) != 0)
    return fail(c, MI_DV_ERR_FORMAT, "not a 525/60 25 Mbit/s DV frame (DSF %d, stype 0x%02x)", frame[3] >> 7, frame[80 * 5 + 48 + 3] & 0x1f);
  if (strides[0] < kW || strides[1] < kCW || strides[2] < kCW) return fail(c, MI_DV_ERR_ARG, "strides below the picture's width");
  DVCHK(c, hipSetDevice(c->device));
  if (!c->d_frame) {
    DVCHK(c, hipMalloc((void**)&c->d_frame, kFrameBytes));
    DVCHK(c, hipMalloc((void**)&c->d_pic, kPicBytes));
    DVCHK(c, hipHostMalloc((void**)&c->h_pic, kPicBytes, hipHostMallocDefault));
  }
  DVCHK(c, hipMemcpyAsync(c->d_frame, frame, kFrameBytes, hipMemcpyHostToDevice, c->stream));
  const int rc = mi_dv_decode_batch(c, c->d_frame, 1, c->d_pic);
  if (rc != MI_DV_OK) return rc;
  DVCHK(c, hipMemcpyAsync(c->h_pic, c->d_pic, kPicBytes, hipMemcpyDeviceToHost, c->stream));
  DVCHK(c, hipStreamSynchronize(c->stream));
  const uint8_t* src = c->h_pic;
  for (int pl = 0; pl < 3; pl++) {
    const int w = pl ? kCW : kW;
    for (int y = 0; y < kH; y++) memcpy(planes[pl] + (size_t)y * strides[pl], src + (size_t)y * w, (size_t)w);
    src += (size_t)w * kH;
  }
  return MI_DV_OK;
}

}  // extern "C"
