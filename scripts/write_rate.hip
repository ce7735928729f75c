// write_rate.hip — what does HBM take for WRITES of 8 KiB cells, alone and next to reads?
// The materialising kernels whose output is one 8 KiB bitmap cell per wavefront (k_shift, k_flip, k_bsi_add, the plain pair
// set-ops) sit at 0.59-0.73 of the 8 TB/s read peak; this separates "the write path" from "the kernel": one wavefront per
// cell, eight 16-byte stores per lane, nothing else.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/write_rate.hip -o scripts/write_rate
//   scripts/write_rate            (prints one line per shape: bytes read / written per cell, us, TB/s of read + write)
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

typedef unsigned long long u64;

// one wave per cell: RD = KiB read per cell (0, 1, 2, 8, 16), 8 KiB written; NT = nontemporal stores (and loads)
template <int RD, bool NT, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_cells(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, u64 n_cells, u64 salt) {
  const int lane = threadIdx.x & 63;
  const u64 cell = (u64)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (cell >= n_cells) return;
  ulonglong2 acc;
  acc.x = salt + cell;
  acc.y = salt ^ (u64)lane;
  if constexpr (RD > 0) {
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(src + cell * (u64)(RD * 1024)) + lane;
    ulonglong2 v[RD > 0 ? RD : 1];
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      if (NT) {
        v[u].x = __builtin_nontemporal_load(&p[u * 64].x);
        v[u].y = __builtin_nontemporal_load(&p[u * 64].y);
      } else {
        v[u] = p[u * 64];
      }
    }
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      acc.x ^= v[u].x;
      acc.y += v[u].y;
    }
  }
  ulonglong2* q = reinterpret_cast<ulonglong2*>(dst + cell * 8192ull) + lane;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    ulonglong2 w;
    w.x = acc.x + u;
    w.y = acc.y ^ u;
    if (NT) {
      __builtin_nontemporal_store(w.x, &q[u * 64].x);
      __builtin_nontemporal_store(w.y, &q[u * 64].y);
    } else {
      q[u * 64] = w;
    }
  }
}

// read-only reference on the same grid shape: 8 KiB read per wave, one dword written per wave
template <int WPB>
__global__ void __launch_bounds__(64 * WPB) k_read(const uint8_t* __restrict__ src, u64* __restrict__ out, u64 n_cells) {
  const int lane = threadIdx.x & 63;
  const u64 cell = (u64)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (cell >= n_cells) return;
  const ulonglong2* p = reinterpret_cast<const ulonglong2*>(src + cell * 8192ull) + lane;
  ulonglong2 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    v[u].x = __builtin_nontemporal_load(&p[u * 64].x);
    v[u].y = __builtin_nontemporal_load(&p[u * 64].y);
  }
  u64 a = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) a += v[u].x ^ v[u].y;
  if (a == 0x1234567ull) out[0] = a;
}

template <class F>
static void timed(const char* name, double rd_bytes, double wr_bytes, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch(i);
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  const int reps = 10;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    launch(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double us = sum / reps * 1e3;
  printf("%-58s read %7.1f MB written %7.1f MB  avg %8.1f us (min %8.1f)  %5.2f TB/s  (writes alone %5.2f TB/s)\n", name, rd_bytes / 1e6, wr_bytes / 1e6, us,
         best * 1e3, (rd_bytes + wr_bytes) / (us * 1e-6) * 1e-12, wr_bytes / (us * 1e-6) * 1e-12);
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const u64 n_cells = argc > 1 ? strtoull(argv[1], nullptr, 10) : 66560;  // k_shift / k_flip on 64 shards of config 3's rows: 65 rows x 16 x 64
  uint8_t *src, *dst;
  u64* out;
  const u64 src_bytes = n_cells * 16384ull, dst_bytes = n_cells * 8192ull;
  CK(hipMalloc(&src, src_bytes));
  CK(hipMalloc(&dst, dst_bytes * 3));  // three output sets in rotation: a cell is not rewritten while it may still sit in a cache
  CK(hipMalloc(&out, 8));
  CK(hipMemset(src, 1, src_bytes));
  CK(hipMemset(dst, 0, dst_bytes * 3));
  const double W = (double)dst_bytes;
  printf("%llu cells of 8 KiB, one wavefront per cell\n", n_cells);
#define RUN(NAME, RD, NT, WPB)                                                                                                      \
  timed(NAME, (double)n_cells* RD * 1024.0, W, [&](int i) {                                                                          \
    hipLaunchKernelGGL((k_cells<RD, NT, WPB>), dim3((unsigned)((n_cells + WPB - 1) / WPB)), dim3(64 * WPB), 0, 0, src, dst + (u64)(i % 3) * dst_bytes, \
                       n_cells, (u64)i);                                                                                            \
  })
  RUN("write only, nt stores, 4 waves per block", 0, true, 4);
  RUN("write only, plain stores, 4 waves per block", 0, false, 4);
  RUN("write only, nt stores, one-wave blocks", 0, true, 1);
  RUN("read 2 KiB + write 8 KiB per cell (k_shift / k_flip's mix), nt", 2, true, 4);
  RUN("read 2 KiB + write 8 KiB per cell, plain", 2, false, 4);
  RUN("read 8 KiB + write 8 KiB per cell (copy), nt", 8, true, 4);
  RUN("read 16 KiB + write 8 KiB per cell (k_setop_dense's mix), nt", 16, true, 4);
  timed("read 8 KiB per wave, nothing written (reference)", (double)n_cells * 8192.0, 0.0,
        [&](int) { hipLaunchKernelGGL((k_read<4>), dim3((unsigned)((n_cells + 3) / 4)), dim3(256), 0, 0, src, out, n_cells); });
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  return 0;
}
