// Memory-system probe for gfx950: what a streaming read and a random 16-byte gather reach on this part, by footprint (an XCD's L2,
// the Infinity Cache, HBM), and -- run under `rocprofv3 --pmc FETCH_SIZE` -- what the counter reports for a KNOWN number of bytes /
// requests in these two access patterns (MI355X_MICROARCH.md: calibrate FETCH_SIZE on your own pattern before trusting an absolute).
// The seed lookup (k_lookup_l1) is a random gather of 16-byte tag runs / slots; k_l2_locate is a stream.  Test infrastructure only.
// build: hipcc --offload-arch=gfx950 -O2 mem_rate.hip -o mem_rate.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// every thread reads 16 bytes per step, a wave 1 KiB contiguous, the grid sweeps the buffer `passes` times
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ buf, size_t nVec, int passes, uint32_t* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (int p = 0; p < passes; p++)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nVec; i += stride) { const uint4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) sink[0] = acc;
}
// every thread issues `per` random 16-byte loads, U of them in flight; DEP: a second load whose address comes from the first
template <int U, bool DEP>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ buf, uint64_t mask, int per, uint32_t* __restrict__ sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < per; i += U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = buf[mix(t * 0x9E3779B97F4A7C15ull + (uint64_t)(i + u)) & mask];
    if (DEP) {
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = buf[(mix(t + (uint64_t)(i + u) * 77u) ^ v[u].x) & mask];
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].w;
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const bool once = argc > 1;                       // under the profiler: one launch per case is enough
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device : %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const size_t maxBytes = (size_t)8 << 30;
  uint4* buf; uint32_t* sink;
  CHECK(hipMalloc(&buf, maxBytes)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, maxBytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timed = [&](auto&& launch) { float best = 1e30f; for (int r = 0; r < (once ? 1 : 3); r++) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best; };
  const size_t foot[] = {(size_t)2 << 20, (size_t)128 << 20, (size_t)8 << 30};
  const char* fname[] = {"2 MiB (an XCD's L2)", "128 MiB (Infinity Cache)", "8 GiB (HBM)"};
  for (int f = 0; f < 3; f++) {
    const size_t nVec = foot[f] / 16;
    const int passes = (int)(((size_t)16 << 30) / foot[f] / (f == 2 ? 1 : 1)); // ~16 GiB of reads per case
    const float ms = timed([&] { hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, buf, nVec, passes, sink); });
    printf("stream  %-26s: %8.1f GB/s   (%.3f ms for %.2f GB read)\n", fname[f], (double)foot[f] * passes / ms / 1e6, ms, (double)foot[f] * passes / 1e9);
  }
  for (int f = 0; f < 3; f++) {
    const uint64_t mask = foot[f] / 16 - 1;
    const int per = 256; const int blocks = 256 * 32;             // 2 M threads x 256 loads = 537 M requests
    const double req = (double)blocks * 256 * per;
    float ms = timed([&] { hipLaunchKernelGGL((k_gather<4, false>), dim3(blocks), dim3(256), 0, 0, buf, mask, per, sink); });
    printf("gather  %-26s: %8.2f G loads/s of 16 B, 4 in flight per thread    (%.3f ms for %.0f M loads)\n", fname[f], req / ms / 1e6, ms, req / 1e6);
    ms = timed([&] { hipLaunchKernelGGL((k_gather<8, false>), dim3(blocks), dim3(256), 0, 0, buf, mask, per, sink); });
    printf("gather  %-26s: %8.2f G loads/s of 16 B, 8 in flight per thread\n", fname[f], req / ms / 1e6);
    ms = timed([&] { hipLaunchKernelGGL((k_gather<4, true>), dim3(blocks), dim3(256), 0, 0, buf, mask, per, sink); });
    printf("gather  %-26s: %8.2f G loads/s of 16 B, pairs of dependent loads  (%.0f M loads)\n", fname[f], 2 * req / ms / 1e6, 2 * req / 1e6);
  }
  return 0;
}
