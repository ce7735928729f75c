// LDS atomic / store throughput on gfx950, as a function of active lanes and address pattern.
// 8 wavefronts per block (one block per CU) each issue ITER ds ops back to back into an 64 KB LDS array.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_atomic_rate.hip -o scripts/lds_atomic_rate && scripts/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int ITER = 2048;

template <int MODE>  // 5/6/7: the address patterns of the bitmap-stage decode (fbk_matrix_fused.hip.h); 0 ds_or random, 1 ds_or conflict-free (lane -> own bank), 2 ds_write_b32 random, 3 ds_or same dword pairs, 4 ds_or_b64 random
__global__ void __launch_bounds__(512) k(uint32_t* out, uint32_t active_lanes, uint32_t seed) {
  __shared__ uint32_t lds[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 0;
  __syncthreads();
  uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 977u;
  const bool on = (uint32_t)lane < active_lanes;
  for (int it = 0; it < ITER; ++it) {
    x = x * 1664525u + 1013904223u;
    uint32_t a = (x >> 8) & 16383u;
    if (MODE == 1) a = (a & ~63u) | lane;
    if (MODE == 3) a = (a & ~63u) | (lane >> 1);
    if (MODE == 5) a = ((x >> 24) & 63u) * 256u + ((x >> 8) & 255u);             // all 64 lanes inside ONE 1 KiB row (row picked per lane: random rows)
    if (MODE == 6) a = (__builtin_amdgcn_readfirstlane((int)(x >> 24)) & 63u) * 256u + ((x >> 8) & 255u);  // all 64 lanes inside the SAME 1 KiB row
    if (MODE == 7) a = (__builtin_amdgcn_readfirstlane((int)(x >> 24)) & 63u) * 256u + lane * 4u + ((x >> 8) & 3u);  // same row, lane l -> its own 16-byte piece
    if (on) {
      if (MODE == 2) asm volatile("ds_write_b32 %0, %1" ::"v"(a * 4), "v"(x) : "memory");
      else if (MODE == 4) asm volatile("ds_or_b64 %0, %1" ::"v"((a & ~1u) * 4), "v"((unsigned long long)x) : "memory");
      else asm volatile("ds_or_b32 %0, %1" ::"v"(a * 4), "v"(1u << (x & 31)) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[seed & 16383];
}

template <int MODE>
void run(const char* name, uint32_t* d, uint32_t lanes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, lanes, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, lanes, 2u + r);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5;
  const double cyc_per_inst_cu = us * 2400.0 / (8.0 * ITER);  // cycles of the CU's LDS per wave-instruction at 2.4 GHz
  std::printf("%-34s lanes %2u: %8.1f us  -> %6.1f cycles per wave-instruction per CU, %5.2f lane-ops/clk/CU\n", name, lanes, us, cyc_per_inst_cu,
              lanes / cyc_per_inst_cu);
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 4096);
  for (uint32_t lanes : {64u, 32u, 16u, 8u, 1u}) run<0>("ds_or_b32 random dword", d, lanes);
  for (uint32_t lanes : {64u, 16u}) run<1>("ds_or_b32 conflict-free", d, lanes);
  for (uint32_t lanes : {64u, 16u}) run<2>("ds_write_b32 random dword", d, lanes);
  for (uint32_t lanes : {64u}) run<3>("ds_or_b32 pairs share a dword", d, lanes);
  for (uint32_t lanes : {64u, 16u}) run<4>("ds_or_b64 random qword", d, lanes);
  for (uint32_t lanes : {64u, 16u}) run<5>("ds_or_b32 random rows, dword in row", d, lanes);
  for (uint32_t lanes : {64u, 32u, 16u}) run<6>("ds_or_b32 all lanes in ONE 1 KiB row", d, lanes);
  for (uint32_t lanes : {64u}) run<7>("ds_or_b32 one row, lane owns a piece", d, lanes);
  return 0;
}
