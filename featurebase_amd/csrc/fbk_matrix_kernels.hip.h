// fbk_matrix_kernels.hip.h — the many-row IntersectionCount matrix (GroupBy / TopN / TopK
// shape, executor.go:8880-8934, 2705-2774) for DENSE batches: every container of the rows
// involved is a bitmap at row*128 KiB + slot*8 KiB, so no descriptor is read at all.
//
// This is the VECTOR-ALU version.  The product path is k_count_matrix_mfma
// (fbk_matrix_mfma.hip.h, 1.9x faster and HBM-bound); this kernel stays selectable with
// FBK_MATRIX_VALU=1 for A/B measurements and as the independent cross-check of
// scripts/tune_matrix.hip.  k_densify_rows at the end of this file serves both.
//
// out[shard][i][j] (+)= sum over the block's slots of |A[shard][i] ∩ F[shard] ∩ B[shard][j]|.
//
// This shape is NOT HBM-bound: nA*nB pairs per slot reuse nA+nB containers.  Per 64-bit word pair
// the work is 2 x v_and_b32 + 2 x v_bcnt_u32_b32; measured on this chip (scripts/valu_rate.hip)
// a SIMD issues one wave64 v_and every 2.2 cycles but one v_bcnt only every 3.9 (half rate), and
// an and->bcnt stream with two wavefronts per SIMD averages 2.9-3.2 cycles per instruction.
// 32 x 32 rows x 16 slots x 128 shards = 2.1 M container pairs x 64 instructions per lane is
// therefore >= 175 us of pure VALU issue, against 168 us for reading every container once at
// 6.5 TB/s.  The kernel is organised around the VALU:
//   * one 512-thread block (8 wavefronts) per (shard, slot group, 32 A rows, 32 B columns);
//     wavefront w owns A rows 4w..4w+3 and all 32 columns;
//   * the container dimension is cut into 4 chunks of 2 KiB.  Per step (slot, chunk) the 32 B
//     chunks are brought into a double-buffered 2 x 64 KiB LDS ring by direct global->LDS DMA
//     (each B byte is fetched from HBM once per block), the wave's 4 A chunks (ANDed with the
//     filter chunk) sit in registers — 4 u64 per lane each — and are prefetched one step
//     ahead, and every lane accumulates its own partial count of all 4 x 32 pairs in 128
//     registers: acc[a][j] += popc(A[a] & B[j]) is v_and + v_bcnt with the add folded into
//     v_bcnt, with NO cross-lane traffic in the loop;
//   * the 64-lane reduction happens once per block, as a transposing butterfly that leaves
//     lane l with the total of pair l (63 shuffles for 64 values instead of 64 x 6).
// The first version (A tile of whole containers in registers, B ring of whole containers,
// a wave reduction per pair) measured 644 us on 128 shards x 32 x 32 rows; this one 350 us,
// i.e. half of the VALU issue rate the microbenchmark reaches with the same instruction mix.
// Ablations on the GPU: without barriers -14 us, without the DMA -50 us, without the A loads
// -58 us, the bare arithmetic + LDS reads 260 us; the schedule of the LDS reads (see below)
// does not matter.  What holds the arithmetic at 1.35x of the microbenchmark is not identified
// yet (VGPR bank conflicts / LDS return traffic on the register file are the candidates).
#pragma once
#include "fbk_kernels.hip.h"

namespace fbk {

constexpr int kMxWaves = 8;     // wavefronts per block
constexpr int kMxTA = 4;        // A rows per wavefront
constexpr int kMxNB = 32;       // B columns per block
constexpr int kMxChunks = 4;    // chunks per container
constexpr int kMxChunkBytes = 2048;
constexpr int kMxCW = 4;        // u64 per lane per chunk

// sum over the 64 lanes of 64 per-lane values; lane l returns the total of v[l]
__device__ __forceinline__ uint32_t transpose_reduce64(uint32_t (&v)[64], int lane) {
#pragma unroll
  for (int lvl = 0; lvl < 6; ++lvl) {
    const int m = 1 << lvl;
    const bool up = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < (32 >> lvl); ++i) {
      const uint32_t keep = up ? v[2 * i + 1] : v[2 * i];
      const uint32_t send = up ? v[2 * i] : v[2 * i + 1];
      v[i] = keep + (uint32_t)__shfl_xor((int)send, m, kWave);
    }
  }
  return v[0];
}

struct MxA {
  u64 w[kMxTA][kMxCW];
};

__global__ void __launch_bounds__(512) k_count_matrix_dense(
    const uint8_t* __restrict__ arenaA, const uint32_t* __restrict__ rowsA, uint32_t nA, const uint8_t* __restrict__ arenaB,
    const uint32_t* __restrict__ rowsB, uint32_t nBtot, const uint8_t* __restrict__ arenaF,
    const uint32_t* __restrict__ rowsF, uint32_t n_shards, uint32_t spb, u64* __restrict__ out_shard) {
  // Two separate LDS objects, not one [2][...] array: the compiler must be able to prove that the
  // global->LDS DMA filling one buffer cannot alias the ds_reads of the other, otherwise it puts
  // s_waitcnt vmcnt(0) in front of every ds_read and the "prefetch" of the next step (DMA and the
  // A loads) is serialised with the arithmetic (seen in the ISA of the first version; that alone
  // cost a third of the kernel time).
  __shared__ u64 ring0[kMxNB][kMxChunkBytes / 8];  // 64 KiB
  __shared__ u64 ring1[kMxNB][kMxChunkBytes / 8];  // 64 KiB
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps row bases in SGPRs
  const uint32_t agroups = (nA + kMxWaves * kMxTA - 1) / (kMxWaves * kMxTA);
  const uint32_t btiles = (nBtot + kMxNB - 1) / kMxNB;
  const uint32_t sgroups = kSlots / spb;
  uint32_t b = blockIdx.x;
  const uint32_t bt = b % btiles;
  b /= btiles;
  const uint32_t ag = b % agroups;
  b /= agroups;
  const uint32_t sg = b % sgroups;
  const uint32_t shard = b / sgroups;
  if (shard >= n_shards) return;
  const uint32_t i0 = ag * kMxWaves * kMxTA + wv * kMxTA;
  const uint32_t j0 = bt * kMxNB;
  const uint32_t nB = min((uint32_t)kMxNB, nBtot - j0);
  const uint64_t rowBytes = (uint64_t)kSlots * 8192;

  // per-wave constant base addresses (slot 0, chunk 0) of its A rows, the filter and the 4 B
  // columns this wave stages
  const uint8_t* pa[kMxTA];
#pragma unroll
  for (int a = 0; a < kMxTA; ++a)
    pa[a] = (i0 + a < nA) ? arenaA + (uint64_t)rowsA[(uint64_t)shard * nA + i0 + a] * rowBytes : nullptr;
  const uint8_t* pf = arenaF ? arenaF + (uint64_t)rowsF[shard] * rowBytes : nullptr;
  constexpr int kPerWave = kMxNB / kMxWaves;  // 4 B columns staged per wave
  const uint8_t* pb[kPerWave];
#pragma unroll
  for (int q = 0; q < kPerWave; ++q) {
    const uint32_t j = wv * kPerWave + q;
    pb[q] = (j < nB) ? arenaB + (uint64_t)rowsB[(uint64_t)shard * nBtot + j0 + j] * rowBytes : nullptr;
  }

  uint32_t acc[kMxTA][kMxNB];
#pragma unroll
  for (int a = 0; a < kMxTA; ++a)
#pragma unroll
    for (int j = 0; j < kMxNB; ++j) acc[a][j] = 0;

  const uint32_t steps = spb * kMxChunks;
  const uint32_t loff = lane * 16;  // the only per-lane part of every address
  auto step_off = [&](uint32_t st) -> uint32_t {  // byte offset of (slot, chunk) inside a row
    return (sg * spb + st / kMxChunks) * 8192u + (st % kMxChunks) * kMxChunkBytes;
  };
  auto stage_b = [&](uint32_t st, u64 (*ring)[kMxChunkBytes / 8]) {  // this wave's 4 B chunks -> ring
    const uint32_t off = step_off(st) + loff;
#pragma unroll
    for (int q = 0; q < kPerWave; ++q) {
      if (pb[q]) {
        uint8_t* l = reinterpret_cast<uint8_t*>(&ring[wv * kPerWave + q][0]);
        __builtin_amdgcn_global_load_lds((gptr_t)(pb[q] + off), (lptr_t)l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(pb[q] + off + 1024), (lptr_t)(l + 1024), 16, 0, 0);
      }
    }
  };
  auto load_a = [&](uint32_t st, MxA& A) {  // A chunks of step st, ANDed with the filter chunk
    const uint32_t off = step_off(st) + loff;
    ulonglong2 f0, f1;
    f0.x = f0.y = f1.x = f1.y = ~0ull;
    if (pf) {
      f0 = *reinterpret_cast<const ulonglong2*>(pf + off);
      f1 = *reinterpret_cast<const ulonglong2*>(pf + off + 1024);
    }
#pragma unroll
    for (int a = 0; a < kMxTA; ++a) {
      ulonglong2 v0, v1;
      v0.x = v0.y = v1.x = v1.y = 0;
      if (pa[a]) {
        v0 = ld_stream(reinterpret_cast<const ulonglong2*>(pa[a] + off));
        v1 = ld_stream(reinterpret_cast<const ulonglong2*>(pa[a] + off + 1024));
      }
      A.w[a][0] = v0.x & f0.x;
      A.w[a][1] = v0.y & f0.y;
      A.w[a][2] = v1.x & f1.x;
      A.w[a][3] = v1.y & f1.y;
    }
  };
  // acc += popcount(a & b): v_and_b32 + v_bcnt_u32_b32 per 32-bit half, the running count being
  // v_bcnt's addend.  Written as asm because the compiler otherwise emits v_bcnt(x, 0) and
  // separate v_add3 (12-25 % more VALU work in a VALU-bound loop).
  auto pc = [](uint32_t& c, u64 a, u64 b) {
    const u64 x = a & b;
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c) : "v"((uint32_t)x));
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c) : "v"((uint32_t)(x >> 32)));
  };
  struct BPair {
    ulonglong2 b0, b1, c0, c1;  // chunk of column j (b) and of column j+1 (c)
  };
  auto compute = [&](const MxA& A, const u64 (*ring)[kMxChunkBytes / 8]) {
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(&ring[0][0]) + lane;
    auto read_pair = [&](int j, BPair& P) {
      const ulonglong2* qn = q + j * (kMxChunkBytes / 16);
      P.b0 = qn[0];
      P.b1 = qn[64];
      P.c0 = qn[128];
      P.c1 = qn[192];
    };
    auto count_half = [&](int j, const BPair& P, int a0) {
#pragma unroll
      for (int a = a0; a < a0 + kMxTA / 2; ++a) {
        pc(acc[a][j], A.w[a][0], P.b0.x);
        pc(acc[a][j + 1], A.w[a][0], P.c0.x);
        pc(acc[a][j], A.w[a][1], P.b0.y);
        pc(acc[a][j + 1], A.w[a][1], P.c0.y);
        pc(acc[a][j], A.w[a][2], P.b1.x);
        pc(acc[a][j + 1], A.w[a][2], P.c1.x);
        pc(acc[a][j], A.w[a][3], P.b1.y);
        pc(acc[a][j + 1], A.w[a][3], P.c1.y);
      }
    };
    // Two register sets alternate; every basic block is closed by a branch on a scalar the
    // compiler cannot see through, which pins the order (instruction selection otherwise hoists
    // all 64 LDS reads of the step above the arithmetic: 256 live registers, 2.5 KB of spills per
    // lane; sched_barrier does not stop that and sched_group_barrier takes minutes to compile).
    // Per column pair: block 1 counts A rows 0..1, block 2 issues the LDS reads of the NEXT pair
    // and counts A rows 2..3 — so the s_waitcnt lgkmcnt(0) that opens the next pair's block 1
    // finds its data 64 VALU instructions old instead of freshly issued.  (Issuing the reads
    // as asm a whole pair ahead with an exact lgkmcnt(4), or halving the number of branches,
    // changes nothing measurable: 350 us either way.)
    BPair P, Q;
    read_pair(0, P);
    auto opaque = [] {
      uint32_t one;
      asm volatile("s_mov_b32 %0, 1" : "=s"(one));
      return one;
    };
#pragma unroll
    for (int j = 0; j < kMxNB; j += 4) {
      if (opaque()) count_half(j, P, 0);
      if (opaque()) {
        read_pair(j + 2, Q);
        count_half(j, P, kMxTA / 2);
      }
      if (opaque()) count_half(j + 2, Q, 0);
      if (opaque()) {
        if (j + 4 < kMxNB) read_pair(j + 4, P);
        count_half(j + 2, Q, kMxTA / 2);
      }
    }
  };

  MxA A0, A1;
  stage_b(0, ring0);
  load_a(0, A0);
  for (uint32_t st = 0; st < steps; st += 2) {
    // even step: compute from ring0 with A0 while ring1 / A1 are being filled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage_b(st + 1, ring1);
    load_a(st + 1, A1);
    compute(A0, ring0);
    // odd step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 2 < steps) {
      stage_b(st + 2, ring0);
      load_a(st + 2, A0);
    }
    compute(A1, ring1);
  }

  // one reduction per block: two passes of 64 values (a = 0,1 then a = 2,3)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    uint32_t v[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) v[q] = acc[2 * p + (q >> 5)][q & 31];
    const uint32_t tot = transpose_reduce64(v, lane);
    const uint32_t a = 2 * p + (lane >> 5), j = lane & 31;
    if (i0 + a < nA && j < nB && tot) {
      u64* dst = &out_shard[((uint64_t)shard * nA + i0 + a) * nBtot + j0 + j];
      if (spb == kSlots) *dst = tot;
      else atomicAdd(dst, (u64)tot);
    }
  }
}

// Rows of any encoding -> temporary dense rows (16 bitmap cells of 8 KiB each), so that the
// matrix kernels can run on array / run containers too: decoding a container once and writing
// 8 KiB is cheap next to the nA x nB pair work that follows.  One wavefront per (row ordinal,
// slot); nil / empty containers become all-zero cells.  Up to three sources (the A rows, the B
// rows, the filter rows of one count-matrix call) share ONE launch.
struct DensifySrc {
  const Slot* slots;
  const uint8_t* arena;
  const uint32_t* rows;
  uint64_t n_rows;
  uint8_t* out;
};
struct DensifyArgs {
  DensifySrc src[3];
};
__global__ void __launch_bounds__(256) k_densify_rows(DensifyArgs args) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  int which = 0;  // wave-uniform
  while (which < 3 && wslot >= args.src[which].n_rows * kSlots) {
    wslot -= args.src[which].n_rows * kSlots;
    ++which;
  }
  if (which == 3) return;
  const DensifySrc& S = args.src[which];
  const uint64_t i = wslot >> 4;
  const uint32_t slot = wslot & 15;
  const Slot s = S.slots[(uint64_t)S.rows[i] * kSlots + slot];
  u64 w[kWordsPerLane];
  if (slot_n(s) == 0) frag_zero(w);
  else frag_load(s, S.arena, lane, lds[wv], w);
  frag_store_bitmap(S.out + wslot * 8192ull, lane, w);
}

// Dense shadows of a batch's heavy containers (fbk.hip heavy_shadow): one wavefront per listed container decodes it into an
// 8 KiB bitmap.  list[k] = index of the container's descriptor; shadow k goes to out + k * 8192.
__global__ void __launch_bounds__(256) k_shadow_build(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena, const uint32_t* __restrict__ list,
                                                     uint64_t n_list, uint8_t* __restrict__ out) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t k = (uint64_t)blockIdx.x * 4 + wv;
  if (k >= n_list) return;
  const Slot s = slots[list[k]];
  u64 w[kWordsPerLane];
  if (slot_n(s) == 0) frag_zero(w);
  else frag_load(s, arena, lane, lds[wv], w);
  frag_store_bitmap(out + k * 8192ull, lane, w);
}

}  // namespace fbk
