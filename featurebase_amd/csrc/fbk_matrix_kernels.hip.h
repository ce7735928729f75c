// fbk_matrix_kernels.hip.h — rows of any encoding -> dense 8 KiB cells: k_densify_rows (temporary dense rows for the
// kernels that take dense operands only — the BSI TopK's plane-major form, fbk_query_api.inc) and k_shadow_build (the dense
// shadows of a batch's heavy containers, fbk.hip heavy_shadow).
//
// (Rounds 1-4 kept the VECTOR-ALU count matrix of round 1 here, k_count_matrix_dense — acc[a][j] += popc(A[a] & B[j]) over a
// double-buffered LDS ring, 350 us for 128 shards x 32 x 32 dense rows where the matrix-core kernel of fbk_matrix_mfma.hip.h
// takes 168-180 — behind option matrix_valu as an A/B and cross-check kernel.  The tests compare every matrix-core result with the
// restated reference per shard; the kernel and its option were removed in round 5.)
#pragma once
#include "fbk_kernels.hip.h"

namespace fbk {

// Rows of any encoding -> temporary dense rows (16 bitmap cells of 8 KiB each).  One wavefront per (row ordinal, slot);
// nil / empty containers become all-zero cells.  Up to three sources share ONE launch.
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
