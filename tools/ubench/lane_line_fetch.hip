// lane_line_fetch.hip — what FETCH_SIZE counts for the walker's access shape.
//
// k_spec_walk has one lane per 2048-byte chunk; a lane reads its chunk (and a lead before it) one 128-byte cache line
// at a time with dword loads, so a wave-wide load instruction touches 64 lines 2 KB apart, 4 bytes of each.  The
// guide's gfx950 correction (FETCH_SIZE reports half the bytes) is stated for 16-byte-per-lane coalesced streams.
// This reads a buffer of KNOWN size exactly once in each shape, under `rocprofv3 --pmc FETCH_SIZE`:
//   k_stream16   16 bytes per lane, coalesced (the calibrated shape)
//   k_lane_line  one lane per 2048-byte chunk, 33 dword loads per 128-byte tile as the walker issues them (the 33rd dword
//                belongs to the next tile: read twice by design, + 3 %), no lead
//   k_lane_line_lead  the same with the walker's 768-byte lead before every chunk: 1.375 x the bytes by design
// Build: hipcc --offload-arch=gfx950 -O3 -o lane_line_fetch lane_line_fetch.hip;  run: ./lane_line_fetch [MiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_stream16(const uint4* __restrict__ p, size_t n16, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int LEAD>
__global__ __launch_bounds__(64) void k_lane_line(const uint8_t* __restrict__ p, size_t bytes, uint32_t* __restrict__ sink) {
  const size_t chunk = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t c0 = chunk * 2048;
  if (c0 >= bytes) return;
  const size_t start = c0 >= (size_t)LEAD ? c0 - LEAD : 0;
  const uint32_t* g4 = (const uint32_t*)(p + start);
  const int tiles = (int)((c0 - start) / 128) + 16;
  uint32_t acc = 0;
  for (int t = 0; t < tiles; t++) {
    uint32_t buf[33];
#pragma unroll
    for (int k = 0; k < 33; k++) buf[k] = g4[t * 32 + k];
#pragma unroll
    for (int k = 0; k < 33; k++) acc ^= buf[k];
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 4096;
  const size_t bytes = mib << 20;
  uint8_t* d;
  uint32_t* sink;
  if (hipMalloc((void**)&d, bytes + 4096) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) return 1;
  hipMemset(d, 1, bytes + 4096);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k_stream16, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)d, bytes / 16, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("k_stream16        %zu MiB in %.3f ms = %.0f GB/s\n", mib, ms, bytes / ms / 1e6);
    const unsigned waves = (unsigned)((bytes / 2048 + 63) / 64);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_lane_line<0>), dim3(waves), dim3(64), 0, 0, d, bytes, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("k_lane_line<0>    %zu MiB in %.3f ms = %.0f GB/s\n", mib, ms, bytes / ms / 1e6);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_lane_line<768>), dim3(waves), dim3(64), 0, 0, d, bytes, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("k_lane_line<768>  %zu MiB (x 1.375 by design) in %.3f ms = %.0f GB/s of chunk bytes\n", mib, ms, bytes / ms / 1e6);
  }
  return 0;
}
