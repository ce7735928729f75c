// tune_dense.hip — standalone micro-benchmark used to tune the dense (all-bitmap) kernels.
// Not part of the product: it includes the product kernels and times experimental
// variants next to them with hipEvents, cycling over several resident data sets so the
// 256 MiB Infinity Cache cannot serve the reads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_dense.hip -o /tmp/tune_dense && /tmp/tune_dense
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../featurebase_amd/csrc/fbk_kernels.hip.h"

using fbk::u64;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                    \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void k_fill(u64* p, size_t n, u64 seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    u64 z = (i + seed) * 0x9E3779B97F4A7C15ull;  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    p[i] = z ^ (z >> 31);
  }
}

// ---- reference ceilings -----------------------------------------------------------------
__global__ void __launch_bounds__(256) k_read_sum(const ulonglong2* __restrict__ a, size_t n16, u64* out) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  u64 acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    ulonglong2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
    acc += v0.x ^ v0.y ^ v1.x ^ v1.y ^ v2.x ^ v2.y ^ v3.x ^ v3.y;
  }
  for (; i < n16; i += stride) acc += a[i].x ^ a[i].y;
  if (acc == 0x1234567) *out = acc;
}

__global__ void __launch_bounds__(256) k_copy(const ulonglong2* __restrict__ a, ulonglong2* __restrict__ o, size_t n16) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  for (; i < n16; i += stride) o[i] = a[i];
}

// ---- icount variants ----------------------------------------------------------------------
// V2: one wave per container, 4 per block, atomics per pair
__global__ void __launch_bounds__(256) k_icount_wave(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                    u64* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint32_t wslot = blockIdx.x * 4 + (threadIdx.x >> 6);
  u64 wa[16], wb[16];
  fbk::frag_load_bitmap(A + (size_t)wslot * 8192, lane, wa);
  fbk::frag_load_bitmap(B + (size_t)wslot * 8192, lane, wb);
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) c += __popcll(wa[i] & wb[i]);
  c = fbk::wave_reduce_add(c);
  if (lane == 0) atomicAdd(&out[wslot >> 4], (u64)c);
}

// V3: block per row pair, non-temporal loads, unroll U
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_icount_row(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                   u64* __restrict__ out) {
  const size_t rowBytes = 16 * 8192;
  const ulonglong2* a = reinterpret_cast<const ulonglong2*>(A + blockIdx.x * rowBytes);
  const ulonglong2* b = reinterpret_cast<const ulonglong2*>(B + blockIdx.x * rowBytes);
  uint32_t c = 0;
  for (int i0 = 0; i0 < 32; i0 += U) {
    ulonglong2 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) {
        va[u].x = __builtin_nontemporal_load(&a[(i0 + u) * 256 + threadIdx.x].x);
        va[u].y = __builtin_nontemporal_load(&a[(i0 + u) * 256 + threadIdx.x].y);
      } else {
        va[u] = a[(i0 + u) * 256 + threadIdx.x];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) {
        vb[u].x = __builtin_nontemporal_load(&b[(i0 + u) * 256 + threadIdx.x].x);
        vb[u].y = __builtin_nontemporal_load(&b[(i0 + u) * 256 + threadIdx.x].y);
      } else {
        vb[u] = b[(i0 + u) * 256 + threadIdx.x];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) c += __popcll(va[u].x & vb[u].x) + __popcll(va[u].y & vb[u].y);
  }
  c = fbk::wave_reduce_add(c);
  __shared__ uint32_t part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (u64)part[0] + part[1] + part[2] + part[3];
}

// V4: persistent grid: G blocks stride over row pairs
__global__ void __launch_bounds__(256) k_icount_persist(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                       u64* __restrict__ out, uint32_t n_pairs) {
  const size_t rowBytes = 16 * 8192;
  __shared__ uint32_t part[4];
  for (uint32_t pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
    const ulonglong2* a = reinterpret_cast<const ulonglong2*>(A + pair * rowBytes);
    const ulonglong2* b = reinterpret_cast<const ulonglong2*>(B + pair * rowBytes);
    uint32_t c = 0;
    for (int i0 = 0; i0 < 32; i0 += 8) {
      ulonglong2 va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) va[u] = a[(i0 + u) * 256 + threadIdx.x];
#pragma unroll
      for (int u = 0; u < 8; ++u) vb[u] = b[(i0 + u) * 256 + threadIdx.x];
#pragma unroll
      for (int u = 0; u < 8; ++u) c += __popcll(va[u].x & vb[u].x) + __popcll(va[u].y & vb[u].y);
    }
    c = fbk::wave_reduce_add(c);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) out[pair] = (u64)part[0] + part[1] + part[2] + part[3];
    __syncthreads();
  }
}

// ---- setop variants -------------------------------------------------------------------------
template <bool NTL, bool NT>
__global__ void __launch_bounds__(256) k_and_row(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                uint8_t* __restrict__ O, u64* __restrict__ out) {
  const size_t rowBytes = 16 * 8192;
  const ulonglong2* a = reinterpret_cast<const ulonglong2*>(A + blockIdx.x * rowBytes);
  const ulonglong2* b = reinterpret_cast<const ulonglong2*>(B + blockIdx.x * rowBytes);
  ulonglong2* o = reinterpret_cast<ulonglong2*>(O + blockIdx.x * rowBytes);
  uint32_t c = 0;
  for (int i0 = 0; i0 < 32; i0 += 8) {
    ulonglong2 va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) va[u] = NTL ? fbk::ld_stream(&a[(i0 + u) * 256 + threadIdx.x]) : a[(i0 + u) * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 8; ++u) vb[u] = NTL ? fbk::ld_stream(&b[(i0 + u) * 256 + threadIdx.x]) : b[(i0 + u) * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      va[u].x &= vb[u].x;
      va[u].y &= vb[u].y;
      c += __popcll(va[u].x) + __popcll(va[u].y);
      if (NT) {
        __builtin_nontemporal_store(va[u].x, &o[(i0 + u) * 256 + threadIdx.x].x);
        __builtin_nontemporal_store(va[u].y, &o[(i0 + u) * 256 + threadIdx.x].y);
      } else {
        o[(i0 + u) * 256 + threadIdx.x] = va[u];
      }
    }
  }
  c = fbk::wave_reduce_add(c);
  __shared__ uint32_t part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (u64)part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(256) k_and_wave_nt(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                    uint8_t* __restrict__ O, u64* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint32_t wslot = blockIdx.x * 4 + (threadIdx.x >> 6);
  u64 wa[16], wb[16];
  fbk::frag_load_bitmap(A + (size_t)wslot * 8192, lane, wa);
  fbk::frag_load_bitmap(B + (size_t)wslot * 8192, lane, wb);
  uint32_t c = 0;
  ulonglong2* o = reinterpret_cast<ulonglong2*>(O + (size_t)wslot * 8192);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 v;
    v.x = wa[2 * j] & wb[2 * j];
    v.y = wa[2 * j + 1] & wb[2 * j + 1];
    c += __popcll(v.x) + __popcll(v.y);
    fbk::st_stream(&o[j * 64 + lane], v);
  }
  c = fbk::wave_reduce_add(c);
  if (lane == 0) atomicAdd(&out[wslot >> 4], (u64)c);
}

int main(int argc, char** argv) {
  const uint32_t n = 1024;                       // row pairs per set
  const size_t rowBytes = 16 * 8192;
  const size_t setBytes = n * rowBytes;          // 128 MiB per operand
  const int sets = argc > 1 ? atoi(argv[1]) : 4; // distinct resident data sets
  const int iters = 40 * sets;
  std::vector<uint8_t*> A(sets), B(sets), O(sets);
  u64* out;
  uint32_t* rows;
  fbk::Slot* oslots;
  CK(hipMalloc(&out, n * 8));
  CK(hipMalloc(&rows, n * 4));
  CK(hipMalloc(&oslots, (size_t)n * 16 * sizeof(fbk::Slot)));
  std::vector<uint32_t> hr(n);
  for (uint32_t i = 0; i < n; ++i) hr[i] = i;
  CK(hipMemcpy(rows, hr.data(), n * 4, hipMemcpyHostToDevice));
  for (int s = 0; s < sets; ++s) {
    CK(hipMalloc(&A[s], setBytes));
    CK(hipMalloc(&B[s], setBytes));
    CK(hipMalloc(&O[s], setBytes));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A[s], setBytes / 8, (u64)(2 * s + 1) << 40);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)B[s], setBytes / 8, (u64)(2 * s + 2) << 40);
  }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<u64> ref(n), got(n);

#define TIME(name, bytes, LAUNCH)                                                              \
  do {                                                                                         \
    for (int w = 0; w < 2 * sets; ++w) {                                                       \
      int s = w % sets;                                                                        \
      LAUNCH;                                                                                  \
    }                                                                                          \
    CK(hipDeviceSynchronize());                                                                \
    CK(hipEventRecord(e0, 0));                                                                 \
    for (int it = 0; it < iters; ++it) {                                                       \
      int s = it % sets;                                                                       \
      LAUNCH;                                                                                  \
    }                                                                                          \
    CK(hipEventRecord(e1, 0));                                                                 \
    CK(hipEventSynchronize(e1));                                                               \
    CK(hipGetLastError());                                                                     \
    float ms;                                                                                  \
    CK(hipEventElapsedTime(&ms, e0, e1));                                                      \
    double us = ms * 1e3 / iters;                                                              \
    printf("%-34s %8.2f us  %8.1f GB/s  (%.1f%% of 8 TB/s)\n", name, us, (bytes) / us / 1e3, \
           (bytes) / us / 1e3 / 80.0);                                                         \
  } while (0)

  const double rd = 2.0 * setBytes, rdwr = 3.0 * setBytes;
  printf("sets=%d (working set %.0f MiB read)\n", sets, sets * rd / 1048576.0);
  TIME("read_sum 2x128MiB (ceiling)", rd, {
    hipLaunchKernelGGL(k_read_sum, dim3(2048), dim3(256), 0, 0, (const ulonglong2*)A[s], setBytes / 16, out);
    hipLaunchKernelGGL(k_read_sum, dim3(2048), dim3(256), 0, 0, (const ulonglong2*)B[s], setBytes / 16, out);
  });
  TIME("copy 128MiB->128MiB x1.5 (ceiling)", rdwr, {
    hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const ulonglong2*)A[s], (ulonglong2*)O[s], setBytes / 16);
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, 0, (const ulonglong2*)B[s], (ulonglong2*)O[s], setBytes / 32);
  });
  // reference result
  hipLaunchKernelGGL(fbk::k_icount_dense<16>, dim3(n), dim3(256), 0, 0, A[0], rows, B[0], rows, out, (fbk::u64*)nullptr, (uint32_t*)nullptr, (uint32_t)n, (fbk::u64*)nullptr);
  CK(hipMemcpy(ref.data(), out, n * 8, hipMemcpyDeviceToHost));
  auto check = [&](const char* nm) {
    CK(hipMemcpy(got.data(), out, n * 8, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i)
      if (got[i] != ref[i]) {
        printf("  !! %s mismatch at %u\n", nm, i);
        return;
      }
  };
  TIME("icount product <16>", rd, hipLaunchKernelGGL(fbk::k_icount_dense<16>, dim3(n), dim3(256), 0, 0, A[s], rows, B[s], rows, out, (fbk::u64*)nullptr, (uint32_t*)nullptr, (uint32_t)n, (fbk::u64*)nullptr));
  TIME("icount product <4> +memset", rd, {
    CK(hipMemsetAsync(out, 0, n * 8, 0));
    hipLaunchKernelGGL(fbk::k_icount_dense<4>, dim3(n * 4), dim3(256), 0, 0, A[s], rows, B[s], rows, out, (fbk::u64*)nullptr, (uint32_t*)nullptr, (uint32_t)n, (fbk::u64*)nullptr);
  });
  TIME("icount wave/container +memset", rd, {
    CK(hipMemsetAsync(out, 0, n * 8, 0));
    hipLaunchKernelGGL(k_icount_wave, dim3(n * 4), dim3(256), 0, 0, A[s], B[s], out);
  });
  TIME("icount row U=4", rd, hipLaunchKernelGGL((k_icount_row<4, false>), dim3(n), dim3(256), 0, 0, A[s], B[s], out));
  TIME("icount row U=8", rd, hipLaunchKernelGGL((k_icount_row<8, false>), dim3(n), dim3(256), 0, 0, A[s], B[s], out));
  TIME("icount row U=16", rd, hipLaunchKernelGGL((k_icount_row<16, false>), dim3(n), dim3(256), 0, 0, A[s], B[s], out));
  TIME("icount row U=8 nontemporal", rd, hipLaunchKernelGGL((k_icount_row<8, true>), dim3(n), dim3(256), 0, 0, A[s], B[s], out));
  hipLaunchKernelGGL((k_icount_row<8, true>), dim3(n), dim3(256), 0, 0, A[0], B[0], out);
  check("row nt");
  TIME("icount row U=4 nontemporal", rd, hipLaunchKernelGGL((k_icount_row<4, true>), dim3(n), dim3(256), 0, 0, A[s], B[s], out));
  TIME("icount row U=16 nontemporal", rd, hipLaunchKernelGGL((k_icount_row<16, true>), dim3(n), dim3(256), 0, 0, A[s], B[s], out));
  TIME("icount persistent 512 blocks", rd, hipLaunchKernelGGL(k_icount_persist, dim3(512), dim3(256), 0, 0, A[s], B[s], out, n));
  TIME("icount persistent 768 blocks", rd, hipLaunchKernelGGL(k_icount_persist, dim3(768), dim3(256), 0, 0, A[s], B[s], out, n));
  hipLaunchKernelGGL(k_icount_persist, dim3(512), dim3(256), 0, 0, A[0], B[0], out, n);
  check("persist");

  TIME("AND product wave/container", rdwr, {
    CK(hipMemsetAsync(out, 0, n * 8, 0));
    hipLaunchKernelGGL(fbk::k_setop_dense<0>, dim3(n * 4), dim3(256), 0, 0, A[s], rows, B[s], rows, O[s], oslots, out);
  });
  // the reference counts were computed for data set 0: run the checked launch on THAT set (the timed
  // loop above ends on whichever set the cycle stopped at — comparing that with set 0's reference is
  // what printed "and product mismatch" for sets > 1 in the round-1 log; the kernel was never wrong)
  CK(hipMemsetAsync(out, 0, n * 8, 0));
  hipLaunchKernelGGL(fbk::k_setop_dense<0>, dim3(n * 4), dim3(256), 0, 0, A[0], rows, B[0], rows, O[0], oslots, out);
  check("and product");
  TIME("AND row ld plain st plain", rdwr, hipLaunchKernelGGL((k_and_row<false, false>), dim3(n), dim3(256), 0, 0, A[s], B[s], O[s], out));
  TIME("AND row ld plain st nt", rdwr, hipLaunchKernelGGL((k_and_row<false, true>), dim3(n), dim3(256), 0, 0, A[s], B[s], O[s], out));
  TIME("AND row ld nt st plain", rdwr, hipLaunchKernelGGL((k_and_row<true, false>), dim3(n), dim3(256), 0, 0, A[s], B[s], O[s], out));
  TIME("AND row ld nt st nt", rdwr, hipLaunchKernelGGL((k_and_row<true, true>), dim3(n), dim3(256), 0, 0, A[s], B[s], O[s], out));
  TIME("AND wave/container ld nt st nt", rdwr, {
    CK(hipMemsetAsync(out, 0, n * 8, 0));
    hipLaunchKernelGGL(k_and_wave_nt, dim3(n * 4), dim3(256), 0, 0, A[s], B[s], O[s], out);
  });
  hipLaunchKernelGGL((k_and_row<true, true>), dim3(n), dim3(256), 0, 0, A[0], B[0], O[0], out);
  check("and row nt");
  return 0;
}
