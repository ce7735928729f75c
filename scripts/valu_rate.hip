// Issue-rate microbenchmark for the VALU ops of the count-matrix inner loop on gfx950:
// how many cycles does one wave64 v_and_b32 / v_bcnt_u32_b32 take on a SIMD?
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  uint32_t a[16], c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = threadIdx.x * 2654435761u + i;
    c[i] = i;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(c[i]) : "v"(a[i]));
        if (MODE == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c[i]) : "v"(a[i]));
        if (MODE == 2) {
          uint32_t t;
          asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(c[(i + 1) & 15]));
          asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c[i]) : "v"(t));
        }
        if (MODE == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(c[i]) : "v"(a[i]));
      }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(const char* name, int waves_per_simd, int ops_per_inner) {
  const int blocks = 256 * waves_per_simd;  // 4 waves per block = 1 per SIMD per block per CU
  uint32_t* d;
  hipMalloc(&d, blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = double(iters) * 128 * ops_per_inner * waves_per_simd;  // wave-instructions per SIMD
  const double ns_per_instr = ms * 1e6 / instr_per_simd;
  std::printf("%-28s waves/SIMD %d: %.3f ms, %.3f ns per wave-instruction per SIMD (= %.2f cycles at 2.2 GHz)\n", name, waves_per_simd, ms,
              ns_per_instr, ns_per_instr * 2.2);
  hipFree(d);
  return ns_per_instr;
}

// the count-matrix kernel's shape: ONE 512-thread block per CU (128 KiB of LDS), i.e. 2 waves per
// SIMD from the same workgroup, chains of and -> bcnt -> bcnt on the same accumulator
template <int CHAIN>
__global__ void __launch_bounds__(512) k512(uint32_t* out, int iters) {
  __shared__ uint32_t big[32768];  // 128 KiB: one block per CU
  uint32_t a[16], c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = threadIdx.x * 2654435761u + i;
    c[i] = i;
  }
  if (iters < 0) big[threadIdx.x] = 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        uint32_t t;
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(a[(i + 1) & 15]));
        if (CHAIN) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c[i & ~1]) : "v"(t));  // two in a row on one accumulator
        else asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c[i]) : "v"(t));
      }
  }
  uint32_t s = big[(threadIdx.x * 7) & 32767];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int CHAIN>
void run512(const char* name) {
  uint32_t* d;
  hipMalloc(&d, 256 * 512 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k512<CHAIN>, dim3(256), dim3(512), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k512<CHAIN>, dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = double(iters) * 128 * 2 * 2;  // 2 instructions per inner step, 2 waves per SIMD
  std::printf("%-40s %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.2 GHz\n", name, ms, ms * 1e6 / instr_per_simd * 2.2);
  hipFree(d);
}

int main() {
  run512<0>("512-thread block, independent accumulators");
  run512<1>("512-thread block, 2 bcnt per accumulator");
  for (int w : {1, 2, 4}) {
    run<0>("v_and_b32", w, 1);
    run<1>("v_bcnt_u32_b32", w, 1);
    run<3>("v_add_u32", w, 1);
    run<2>("v_and + v_bcnt (pair)", w, 2);
  }
  return 0;
}
