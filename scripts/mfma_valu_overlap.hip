// Does ONE wave overlap its own matrix instructions with its own vector instructions on gfx950, and what does a second wave on
// the SIMD change?  The consumer waves of k_count_matrix_fusedq run, per 128-byte octet of K, 40 v_and / v_lshrrev + 4
// v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands) + 3 ds_read_b128 + 2 ds_write_b128; the ablations say a consumer wave alone
// needs ~3000 cycles per stage of 8 octets where the matrix pipe needs 1024 and the vector pipe 1280 (profiles/r05_fused_ablate*).
//
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o scripts/mfma_valu_overlap && scripts/mfma_valu_overlap
//
// Modes (one block per CU, W waves per SIMD, every wave runs `iters` octets):
//   0  4 MFMA per octet, nothing else          1  40 VALU per octet, nothing else
//   2  the consumer's octet: 8 VALU, MFMA, 8 VALU, MFMA, 24 VALU, MFMA, MFMA — operands rebuilt in ONE register set (the compiler's code)
//   3  the same with TWO operand register sets alternating (no write-after-read on the MFMA's sources)
//   4  mode 2 + the octet's LDS traffic (3 ds_read_b128 + 2 ds_write_b128, an octet ahead)
//   5  mode 2 with the masks in scalar registers (4-byte encodings instead of 8-byte ones with a literal)      6  mode 1 likewise
// The second half: ds_write_b128 with all / half / a quarter of the lanes enabled — does a masked lane cost LDS time?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  __shared__ u4 lds[4096];  // 64 KiB
  const int lane = threadIdx.x & 63;
  v16f acc0{}, acc1{}, acc2{}, acc3{};
  u4 a = u4{threadIdx.x * 2654435761u, threadIdx.x * 40503u, 0x12345678u, threadIdx.x};
  u4 b = u4{threadIdx.x * 97u, 0x9abcdef0u, threadIdx.x * 31u, 7u};
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = a;
  __syncthreads();
  const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0] + (threadIdx.x >> 6) * 4096u + lane * 16u;
  u4 zero = u4{0, 0, 0, 0};
  asm volatile("" : "+v"(zero));
  v8i oa{}, ob{}, pa{}, pb{};
  uint32_t m1 = 0x11111111u, m2 = 0x22222222u, m4 = 0x44444444u;
  if (MODE >= 5) asm volatile("" : "+s"(m1), "+s"(m2), "+s"(m4));  // the masks live in scalar registers: 4-byte v_and encodings, no literals
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(a), "+v"(b));
    if (MODE == 4) {
      u4 x, y, z;
      asm volatile("ds_read_b128 %0, %1" : "=&v"(x) : "v"(base));
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=&v"(y) : "v"(base));
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=&v"(z) : "v"(base));
      asm volatile("ds_write_b128 %0, %1" ::"v"(base), "v"(zero) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(base), "v"(zero) : "memory");
      asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
      a = a ^ x ^ z;
      b = b ^ y;
    }
    if (MODE == 0) {
      MFMA(acc0, oa, ob);
      MFMA(acc1, oa, ob);
      MFMA(acc2, oa, ob);
      MFMA(acc3, oa, ob);
    } else if (MODE == 6) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          asm volatile("v_and_b32 %0, %2, %1" : "=v"(oa[d]) : "v"(a[d]), "s"(m1));
          asm volatile("v_and_b32 %0, %2, %1" : "=v"(ob[d]) : "v"(b[d]), "s"(m1));
        }
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          asm volatile("v_and_b32 %0, 0x11111111, %1" : "=v"(oa[d]) : "v"(a[d]));
          asm volatile("v_and_b32 %0, 0x11111111, %1" : "=v"(ob[d]) : "v"(b[d]));
        }
    } else {
      constexpr bool TWO = MODE == 3;
#pragma unroll
      for (int d = 0; d < 4; ++d) oa[d] = (int)(a[d] & m1), ob[d] = (int)(b[d] & m1);
      MFMA(acc0, oa, ob);
      v8i& qa = TWO ? pa : oa;
      v8i& qb = TWO ? pb : ob;
#pragma unroll
      for (int d = 0; d < 4; ++d) qa[d] = (int)(a[d] & m2), qb[d] = (int)(b[d] & m2);
      MFMA(acc1, qa, qb);
#pragma unroll
      for (int d = 0; d < 4; ++d) oa[d] = (int)(a[d] & m4), ob[d] = (int)(b[d] & m4);
      v8i ra{}, rb{};
#pragma unroll
      for (int d = 0; d < 4; ++d) ra[d] = (int)((a[d] >> 3) & m1), rb[d] = (int)((b[d] >> 3) & m1);
      MFMA(acc2, oa, ob);
      MFMA(acc3, ra, rb);
    }
  }
  float s = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q] + acc2[q] + acc3[q];
  s += (float)(oa[0] + ob[1] + pa[2] + pb[3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS writes with part of the lanes enabled: `keep` lanes of every 16 write
__global__ void __launch_bounds__(256) kw(uint32_t* out, int iters, int keep, int contiguous) {
  __shared__ u4 lds[4096];
  const int lane = threadIdx.x & 63;
  const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0] + (threadIdx.x >> 6) * 8192u + lane * 16u;
  u4 zero = u4{1, 2, 3, 4};
  asm volatile("" : "+v"(zero));
  if (contiguous ? lane < 4 * keep : (lane & 15) < keep) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base), "v"(zero), "n"(1024 * r) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x].x;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int threads = 256 * waves_per_simd, iters = 20000;
  float* d;
  (void)hipMalloc(&d, 256 * threads * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double ns_oct = best * 1e6 / iters;  // per octet of ONE wave; the SIMD's octets per ns = waves / that
  std::printf("%-46s waves/SIMD %d: %8.3f ms  %7.1f ns per octet per wave  %7.1f ns per octet of the SIMD (%.0f cycles at 2.3 GHz)\n", name, waves_per_simd, best,
              ns_oct, ns_oct / waves_per_simd, ns_oct / waves_per_simd * 2.3);
  (void)hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("0: 4 MFMA (FP4 32x32x64)", w);
    run<1>("1: 40 VALU", w);
    run<2>("2: consumer octet, one operand set", w);
    run<3>("3: consumer octet, two operand sets", w);
    run<4>("4: consumer octet + 3 ds_read + 2 ds_write b128", w);
    run<5>("5: consumer octet, masks in scalar registers", w);
    run<6>("6: 40 VALU, mask in a scalar register", w);
  }
  uint32_t* d;
  (void)hipMalloc(&d, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int contiguous = 0; contiguous < 2; ++contiguous)
  for (int keep : {16, 8, 4, 1}) {
    const int iters = 20000;
    hipLaunchKernelGGL(kw, dim3(256), dim3(256), 0, 0, d, iters, keep, contiguous);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kw, dim3(256), dim3(256), 0, 0, d, iters, keep, contiguous);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("ds_write_b128, %s, 4 waves per CU: %.3f ms = %.1f cycles per instruction per CU at 2.3 GHz\n",
                (std::to_string(contiguous ? 4 * keep : keep) + (contiguous ? " first lanes of 64 enabled" : " of every 16 lanes enabled")).c_str(), ms,
                ms * 1e6 / (iters * 8.0 * 4) * 2.3);
  }
  return 0;
}
