// What does the DEVICE do in the first launches after a pause?  (profiles/r06_first_launches_1024.txt: after any pause of >= 1 ms the
// encoded-row count matrix runs +5 % in its first launch, +25-30 % in its third and fourth, and is back at its sustained time after
// ~20 launches = ~20 ms — whatever the pause was, with or without an upload.)
//
// Two synthetic kernels of ~1 ms, launched 24 times with a host synchronisation after each (the protocol of the measurement above)
// after pauses of 0 / 1 ms / 100 ms / 2 s:
//   stream  every block reads its slice of a 4 GiB buffer once (non-temporal 16-byte loads): HBM-bound, insensitive to the shader clock
//   spin    every wave runs a fixed chain of dependent integer multiply-adds: its time IS 1 / shader clock
// Per launch: duration by HIP events, and the shader clock the kernel itself saw — s_memtime (shader cycles) against s_memrealtime
// (100 MHz, constant) between the first block's start and the last block's end.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/clock_probe.hip -o scripts/clock_probe && scripts/clock_probe > profiles/r06_clock_probe.txt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

typedef unsigned long long u64;

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ in, uint64_t vec_per_block, u64* __restrict__ stamps, uint32_t* __restrict__ sink) {
  const u64 c0 = clock64(), w0 = wall_clock64();
  const uint4* p = in + (uint64_t)blockIdx.x * vec_per_block;
  uint32_t acc = 0;
  for (uint64_t i = threadIdx.x; i < vec_per_block; i += 256) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + i));
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    stamps[4 * blockIdx.x + 0] = c0;
    stamps[4 * blockIdx.x + 1] = w0;
    stamps[4 * blockIdx.x + 2] = clock64();
    stamps[4 * blockIdx.x + 3] = wall_clock64();
  }
}

__global__ void __launch_bounds__(256) k_spin(uint32_t iters, u64* __restrict__ stamps, uint32_t* __restrict__ sink) {
  const u64 c0 = clock64(), w0 = wall_clock64();
  uint32_t a = threadIdx.x * 2654435761u + 1u, b = blockIdx.x | 1u;
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a = a * b + 0x9E3779B9u;
  }
  if (a == 0x12345678u) sink[0] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    stamps[4 * blockIdx.x + 0] = c0;
    stamps[4 * blockIdx.x + 1] = w0;
    stamps[4 * blockIdx.x + 2] = clock64();
    stamps[4 * blockIdx.x + 3] = wall_clock64();
  }
}

int main() {
  const uint64_t bytes = 4ull << 30;
  const uint32_t blocks = 4096;
  uint4* buf;
  u64* stamps;
  uint32_t* sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipMalloc(&stamps, blocks * 4 * sizeof(u64)));
  CHECK(hipMalloc(&sink, 64));
  std::vector<u64> h(blocks * 4);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const uint64_t vec_per_block = bytes / 16 / blocks;
  auto one = [&](int which, float& us, double& mhz) -> int {
    CHECK(hipEventRecord(e0));
    if (which == 0) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, buf, vec_per_block, stamps, sink);
    else hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, 0, 5200u, stamps, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    us = ms * 1e3f;
    CHECK(hipMemcpy(h.data(), stamps, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
    // clocks are per XCD: take the longest-living single block's own ratio (its two counters tick on the same XCD)
    double best = 0, best_ticks = 0;
    for (uint32_t b = 0; b < blocks; ++b) {
      const double cyc = double(h[4 * b + 2] - h[4 * b + 0]), ticks = double(h[4 * b + 3] - h[4 * b + 1]);
      if (ticks > best_ticks) {
        best_ticks = ticks;
        best = cyc / ticks * 100.0;
      }
    }
    mhz = best;
    return 0;
  };
  const char* names[2] = {"stream (HBM-bound, 4 GiB read once)", "spin (dependent integer multiply-adds)"};
  const double pauses_ms[4] = {0, 1, 100, 2000};
  for (int which = 0; which < 2; ++which) {
    float us;
    double mhz;
    for (int i = 0; i < 60; ++i)
      if (one(which, us, mhz)) return 1;  // settle
    std::printf("# %s: sustained %.1f us at %.0f MHz\n", names[which], us, mhz);
    for (double pause : pauses_ms) {
      std::this_thread::sleep_for(std::chrono::microseconds((long long)(pause * 1000)));
      std::printf("%-40s after a pause of %6.0f ms | us:", names[which], pause);
      std::vector<double> clk;
      for (int i = 0; i < 24; ++i) {
        if (one(which, us, mhz)) return 1;
        std::printf(" %.0f", us);
        clk.push_back(mhz);
      }
      std::printf(" | MHz:");
      for (double c : clk) std::printf(" %.0f", c);
      std::printf("\n");
    }
  }
  return 0;
}
