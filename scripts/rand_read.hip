// rand_read.hip — what does HBM deliver for reads of B contiguous bytes at pseudo-random places?
// Every wavefront reads `per_wave` blocks of B bytes (B = 1 KiB .. 64 KiB, 1 KiB = one 64-lane x
// 16-byte load) from a 2 GiB buffer, 8 loads in flight per lane, nt.  Not part of the product: it
// separates "the access pattern of small containers" from "the kernel" for fbk_fold_kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/rand_read.hip -o scripts/rand_read
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                               \
    }                                                        \
  } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// kib = block size in KiB; blocks start at multiples of `align` bytes
__global__ void __launch_bounds__(256) k_rand(const uint8_t* __restrict__ buf, unsigned long long buf_kib, uint32_t kib,
                                             uint32_t per_wave, uint32_t align, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = blockIdx.x * 4ull + (threadIdx.x >> 6);
  unsigned long long acc = 0;
  const unsigned long long slots = (buf_kib * 1024ull - kib * 1024ull) / align;
  for (uint32_t b = 0; b < per_wave; ++b) {
    const unsigned long long start = (mix(wave * 0x9E3779B97F4A7C15ull + b) % slots) * align;
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(buf + start) + lane;
    for (uint32_t c = 0; c < kib; c += 8) {
      ulonglong2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + u < kib) {
          v[u].x = __builtin_nontemporal_load(&p[(c + u) * 64].x);
          v[u].y = __builtin_nontemporal_load(&p[(c + u) * 64].y);
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + u < kib) acc += v[u].x ^ v[u].y;
    }
  }
  if (acc == 0x1234567) *out = acc;
}

int main() {
  const unsigned long long buf_kib = 2ull << 20;  // 2 GiB
  uint8_t* buf;
  unsigned long long* out;
  CK(hipMalloc(&buf, buf_kib * 1024));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(buf, 1, buf_kib * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint32_t blocks = 256 * 8;  // 8 blocks of 4 waves per CU
  for (uint32_t align : {16u, 128u, 4096u}) {
    for (uint32_t kib : {1u, 2u, 4u, 8u, 16u, 32u, 64u}) {
      const uint32_t per_wave = 512 / kib > 0 ? 512 / kib : 1;  // 512 KiB per wave, 4 GiB in total
      for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rand, dim3(blocks), dim3(256), 0, 0, buf, buf_kib, kib, per_wave, align, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 1)
          printf("align %5u  block %3u KiB  %8.1f us  %6.2f TB/s\n", align, kib, ms * 1e3,
                 (double)blocks * 4 * per_wave * kib * 1024 / (ms * 1e-3) * 1e-12);
      }
    }
  }
  CK(hipGetLastError());
  return 0;
}
