// d2h_probe.hip — what a session's copy out costs on this host link, one ingredient at a time.
//   hipcc --offload-arch=gfx950 -O2 -o d2h_probe d2h_probe.hip && ./d2h_probe [bytes] [copies]
// Every case copies `bytes` device -> pinned host `copies` times and prints us per copy and GB/s:
//   a  one destination, back to back on one stream (tools/pcie_probe.py's case)
//   b  six destinations in turn
//   c  six destinations, an event recorded behind every copy, the host waits for copy i - 5 before queuing copy i
//      (what a session of depth 6 does)
//   d  c + every copy waits for an event of a compute stream on which a tiny kernel ran (the session's e_dec)
//   e  d with the copies dealt to two streams in turn
//   f  c with destinations from mmap + MADV_HUGEPAGE + hipHostRegister
//   g  c with hipHostMallocNonCoherent destinations
//   h  c with one destination six times as large, copies at six offsets (one registration)
//   i  c + the host copies bytes / 5 (a packet) into pinned staging before queuing each copy
//   j  i + the host reads one byte per 4 KiB of the picture that has arrived (a consumer)
//   k  e (compute-stream event, two copy streams) + i + j: everything a session's thread does
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void k_tick(uint32_t* p) { p[threadIdx.x] += 1u; }

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)3840 * 2160 * 3 / 2;
  const int n = argc > 2 ? atoi(argv[2]) : 300;
  const int D = 6;
  uint8_t* d = nullptr;
  uint32_t* d_t = nullptr;
  CK(hipMalloc((void**)&d, bytes * D));
  CK(hipMalloc((void**)&d_t, 4096));
  CK(hipMemset(d, 1, bytes * D));
  CK(hipMemset(d_t, 0, 4096));
  hipStream_t s[2], sc;
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  std::vector<uint8_t*> h(D), hn(D), hh(D);
  for (int i = 0; i < D; i++) {
    CK(hipHostMalloc((void**)&h[i], bytes, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&hn[i], bytes, hipHostMallocNonCoherent));
    const size_t rounded = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    void* m = mmap(nullptr, rounded + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return 1;
    uint8_t* al = (uint8_t*)(((uintptr_t)m + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    madvise(al, rounded, MADV_HUGEPAGE);
    for (size_t k = 0; k < rounded; k += 4096) al[k] = 0;
    CK(hipHostRegister(al, rounded, hipHostRegisterDefault));
    hh[i] = al;
  }
  uint8_t* big = nullptr;
  CK(hipHostMalloc((void**)&big, bytes * D, hipHostMallocDefault));
  hipEvent_t ev[D], ec[D];
  for (int i = 0; i < D; i++) {
    CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ec[i], hipEventDisableTiming));
  }
  auto report = [&](const char* name, double dt) {
    printf("%-70s %8.1f us per copy  %6.1f GB/s\n", name, dt / n * 1e6, (double)bytes * n / dt / 1e9);
    fflush(stdout);
  };
  // a, b
  for (int rot = 0; rot < 2; rot++) {
    for (int i = 0; i < 10; i++) CK(hipMemcpyAsync(h[rot ? i % D : 0], d, bytes, hipMemcpyDeviceToHost, s[0]));
    CK(hipStreamSynchronize(s[0]));
    const double t0 = now();
    for (int i = 0; i < n; i++) CK(hipMemcpyAsync(h[rot ? i % D : 0], d + (rot ? (size_t)(i % D) * bytes : 0), bytes, hipMemcpyDeviceToHost, s[0]));
    CK(hipStreamSynchronize(s[0]));
    report(rot ? "b  six destinations in turn, back to back" : "a  one destination, back to back", now() - t0);
  }
  // c .. h
  // host-side work of a session: a packet (a fifth of the picture) copied from pageable memory into pinned staging
  const size_t pkt = bytes / 5;
  std::vector<uint8_t> src(pkt * 24, 3);
  uint8_t* stage = nullptr;
  CK(hipHostMalloc((void**)&stage, pkt * D, hipHostMallocDefault));
  volatile unsigned sink = 0;
  auto session = [&](const char* name, std::vector<uint8_t*>& dst, bool wait_compute, int nstreams, bool offsets,
                     bool host_copy = false, bool touch = false) {
    for (int pass = 0; pass < 2; pass++) {  // pass 0 warms up
      const int m = pass ? n : 12;
      const double t0 = now();
      for (int i = 0; i < m; i++) {
        const int k = i % D;
        if (i >= D) {
          CK(hipEventSynchronize(ev[k]));
          if (touch) {
            unsigned acc = 0;
            for (size_t o = 0; o < bytes; o += 4096) acc += dst[k][o];
            sink += acc;
          }
        }
        if (host_copy) memcpy(stage + (size_t)k * pkt, src.data() + (size_t)(i % 24) * pkt, pkt);
        hipStream_t so = s[nstreams == 2 ? (i & 1) : 0];
        if (wait_compute) {
          hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, sc, d_t);
          CK(hipEventRecord(ec[k], sc));
          CK(hipStreamWaitEvent(so, ec[k], 0));
        }
        uint8_t* to = offsets ? big + (size_t)k * bytes : dst[k];
        CK(hipMemcpyAsync(to, d + (size_t)k * bytes, bytes, hipMemcpyDeviceToHost, so));
        CK(hipEventRecord(ev[k], so));
      }
      CK(hipStreamSynchronize(s[0]));
      CK(hipStreamSynchronize(s[1]));
      if (pass) report(name, now() - t0);
    }
  };
  session("c  six destinations, event per copy, host keeps five queued", h, false, 1, false);
  session("d  c + each copy waits for an event of a compute stream", h, true, 1, false);
  session("e  d + copies dealt to two streams", h, true, 2, false);
  session("f  c, destinations mmap + MADV_HUGEPAGE + hipHostRegister", hh, false, 1, false);
  session("g  c, hipHostMallocNonCoherent destinations", hn, false, 1, false);
  session("h  c, one large destination, six offsets", h, false, 1, true);
  session("i  c + host copies a packet into pinned staging per copy", h, false, 1, false, true, false);
  session("j  i + host touches the arrived picture", h, false, 1, false, true, true);
  session("k  e + i + j", h, true, 2, false, true, true);
  session("c  again", h, false, 1, false);
  return 0;
}
