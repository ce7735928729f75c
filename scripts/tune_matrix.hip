// tune_matrix.hip — standalone timing of the count-matrix kernels (not part of the product):
// 128 shards x (32 A rows + 32 B rows + filter), dense rows of random bits, hipEvent timing of
// k_count_matrix_mfma<HAS_F, WAVES, DEPTH> variants next to k_count_matrix_dense, with a
// cross-check of the two results.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_matrix.hip -o scripts/tune_matrix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../featurebase_amd/csrc/fbk_matrix_kernels.hip.h"
#include "../featurebase_amd/csrc/fbk_matrix_mfma.hip.h"

using fbk::u64;

#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                               \
    }                                                        \
  } while (0)

__global__ void k_fill(u64* p, size_t n, u64 seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    u64 z = (i + seed) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    p[i] = z ^ (z >> 31);
  }
}

struct Problem {
  uint32_t shards, nA, nB;
  uint8_t *A, *B, *F;
  uint32_t *rowsA, *rowsB, *rowsF;
  u64* out;
};

template <typename K>
static float time_kernel(const char* name, K launch, const Problem& p, int iters, std::vector<u64>* result) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t outBytes = (size_t)p.shards * p.nA * p.nB * 8;
  for (int i = 0; i < 3; ++i) {
    CK(hipMemset(p.out, 0, outBytes));
    launch();
  }
  CK(hipDeviceSynchronize());
  float tot = 0;
  for (int i = 0; i < iters; ++i) {
    CK(hipMemsetAsync(p.out, 0, outBytes, 0));
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms;
  }
  CK(hipGetLastError());
  if (result) {
    result->resize(outBytes / 8);
    CK(hipMemcpy(result->data(), p.out, outBytes, hipMemcpyDeviceToHost));
  }
  const double us = tot / iters * 1e3;
  const double bytes = (double)p.shards * (p.nA + p.nB + 1) * 16 * 8192;
  printf("%-44s %8.1f us  %7.2f TB/s\n", name, us, bytes / us * 1e-6);
  return (float)us;
}

int main(int argc, char** argv) {
  Problem p;
  p.shards = argc > 1 ? atoi(argv[1]) : 128;
  p.nA = argc > 2 ? atoi(argv[2]) : 32;
  p.nB = argc > 3 ? atoi(argv[3]) : 32;
  const int iters = 20;
  const size_t rowBytes = 16 * 8192;
  CK(hipMalloc(&p.A, (size_t)p.shards * p.nA * rowBytes));
  CK(hipMalloc(&p.B, (size_t)p.shards * p.nB * rowBytes));
  CK(hipMalloc(&p.F, (size_t)p.shards * rowBytes));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)p.A, (size_t)p.shards * p.nA * rowBytes / 8, 1ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)p.B, (size_t)p.shards * p.nB * rowBytes / 8, 77777777ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)p.F, (size_t)p.shards * rowBytes / 8, 999999999ull);
  std::vector<uint32_t> ra((size_t)p.shards * p.nA), rb((size_t)p.shards * p.nB), rf(p.shards);
  for (size_t i = 0; i < ra.size(); ++i) ra[i] = (uint32_t)i;
  for (size_t i = 0; i < rb.size(); ++i) rb[i] = (uint32_t)i;
  for (size_t i = 0; i < rf.size(); ++i) rf[i] = (uint32_t)i;
  CK(hipMalloc(&p.rowsA, ra.size() * 4));
  CK(hipMalloc(&p.rowsB, rb.size() * 4));
  CK(hipMalloc(&p.rowsF, rf.size() * 4));
  CK(hipMemcpy(p.rowsA, ra.data(), ra.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(p.rowsB, rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(p.rowsF, rf.data(), rf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&p.out, (size_t)p.shards * p.nA * p.nB * 8));
  const uint32_t tiles = ((p.nA + 31) / 32) * ((p.nB + 31) / 32);

  std::vector<u64> ref, got;
  for (uint32_t spb : {16u, 4u}) {
    const uint32_t blocks = p.shards * (16 / spb) * tiles;
    printf("-- spb %u (%u blocks)\n", spb, blocks);
    if (spb == 4)
      time_kernel("valu k_count_matrix_dense", [&] {
        hipLaunchKernelGGL(fbk::k_count_matrix_dense, dim3(blocks), dim3(512), 0, 0, p.A, p.rowsA, p.nA, p.B, p.rowsB, p.nB, p.F,
                           p.rowsF, p.shards, spb, p.out);
      }, p, iters, &ref);
#define VARIANT(W, D, AUX, TM, TN)                                                                                         \
  time_kernel("mfma<F, W=" #W ", D=" #D ", aux=" #AUX ", " #TM "x" #TN ">", [&] {                                           \
    const uint32_t blocks = p.shards * (16 / spb) * ((p.nA + 32 * TM - 1) / (32 * TM)) * ((p.nB + 32 * TN - 1) / (32 * TN)); \
    hipLaunchKernelGGL((fbk::k_count_matrix_mfma<true, W, D, AUX, TM, TN>), dim3(blocks), dim3(W * 64), 0, 0, p.A, p.rowsA, p.nA, p.B, \
                       p.rowsB, p.nB, p.F, p.rowsF, p.shards, spb, p.out);                                                 \
  }, p, iters, &got);                                                                                                      \
  if (!ref.empty() && got != ref) printf("   MISMATCH vs valu kernel\n");
    VARIANT(4, 2, 2, 1, 1)
    VARIANT(4, 3, 2, 1, 1)
    VARIANT(4, 2, 2, 2, 2)
    VARIANT(4, 2, 2, 2, 1)
    VARIANT(4, 2, 2, 1, 2)
    VARIANT(2, 2, 2, 2, 2)
    VARIANT(2, 3, 2, 2, 2)
    time_kernel("mfma<noF, W=4, D=3>", [&] {
      hipLaunchKernelGGL((fbk::k_count_matrix_mfma<false, 4, 3>), dim3(blocks), dim3(256), 0, 0, p.A, p.rowsA, p.nA, p.B, p.rowsB,
                         p.nB, (const uint8_t*)nullptr, (const uint32_t*)nullptr, p.shards, spb, p.out);
    }, p, iters, nullptr);
  }
  return 0;
}
