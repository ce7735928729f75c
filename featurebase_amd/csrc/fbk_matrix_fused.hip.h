// fbk_matrix_fused.hip.h — the many-row IntersectionCount matrix (GroupBy / TopN shape,
// executor.go:8880-8934, 2705-2774) for rows in ANY encoding, on the matrix cores, decoding the
// rows inside the kernel (no decoded row ever goes to HBM): what the kernel (fbk_matrix_fusedq.hip.h)
// and the program it runs share.
//
//   out[shard][i][j] += sum over the block's slots of |A[shard][i] ∩ F[shard] ∩ B[shard][j]|
//
//   * the batch carries a WINDOW INDEX (k_window_index, 16 bytes per container): where each eighth
//     of the value range begins inside an array / run list.  Nothing is searched, walked or
//     re-derived in the kernel;
//   * one block = (shard, slot group, 32 A rows, 32 B rows), 16 wavefronts: 4 CONSUMERS (one per
//     SIMD; bit -> FP4 nibble by ONE v_and per operand dword, v_mfma_scale_f32_32x32x64_f8f6f4, see
//     fbk_matrix_mfma.hip.h) and 12 PRODUCERS that decode;
//   * a stage = the 8192 bit positions [q * 8192, (q + 1) * 8192) of all 65 rows (32 A, 32 B, the
//     filter) as 1 KiB of bitmap per row in LDS (row stride 1040 bytes: the 16 rows of a
//     ds_read_b128 lane group fall into 16 different bank groups); two stages alternate, one
//     barrier per stage;
//   * array bits are OR-ed in with LDS atomics, so any 16-lane group may write any row; the CONSUMERS
//     zero the piece of every row they have just read (the buffer is clean when the producers get it
//     back), which removes the ordering "zero before scatter" between producer waves;
//   * bitmap rows: the q-th KiB of the container, global -> registers -> LDS; run rows: toggles at the
//     clamped start / one past the clamped end, then a parity prefix over the row's 1 KiB
//     (runToBitmap, roaring.go:3792).
//
// THE PROGRAM (round 5).  The work lists of a block depend only on (the batches' descriptors, the query's row lists);
// k_fused_program computes them ONCE per (batch versions, row lists):
//   * FxProg, one per (shard, 32 x 32 tile, container slot): the row table, the bitmap / run / long-array lists and the
//     per-stage item counts — 2384 bytes in global memory that ONE wave copies into LDS with three global->LDS DMA
//     instructions per slot (no registers, no vector instructions);
//   * FxItem, the RESOLVED array items of every stage: {address of the item's first value, number of values, byte offset
//     of its row in a stage buffer}, 16 bytes each, in the order the kernel deals them to its sixteen-lane groups: one
//     16-byte load and ~6 vector instructions per item.
//
// History (the kernels themselves are in the repository's history, their measurements in DESIGN.md sections 6 and 9):
// rounds 2-4, k_count_matrix_fused — every block built the lists in LDS (nine producer waves per slot) and walked
// item -> row table -> address per stage: 863 -> 394 -> 303-315 us on 256 shards of config 3's rows; round 5's first
// program-driven form, k_count_matrix_fusedp — twelve identical producer waves, loads one stage ahead: 282-291 us; the
// kernel of fbk_matrix_fusedq.hip.h: 242-255 us.  Both predecessors were removed when the latter had its parity and
// its A/B (profiles/r05_fused_program_ab.json, r05_fused_ab_final_kernel.json).
#pragma once
#include "fbk_matrix_mfma.hip.h"

namespace fbk {

constexpr int kFxSB = 1024;               // bytes of every row per stage
constexpr int kFxStages = 8192 / kFxSB;   // 8 stages per container slot
constexpr int kFxNR = 65;                 // 32 A rows + 32 B rows + the filter row
constexpr int kFxStride = kFxSB + 16;     // LDS row stride
constexpr int kFxBuf = kFxNR * kFxStride;  // 67 600 bytes per stage buffer
constexpr int kFxConsumers = 4;
constexpr int kFxProducers = 12;
constexpr int kFxWaves = kFxConsumers + kFxProducers;
constexpr int kFxItemArrayMax = 4096;        // arrays up to this length go through the item lists (ArrayMaxSize, roaring.go:46)

// Loads through pointers that went through an integer round trip (row-table entries in LDS) come out of the
// compiler as FLAT loads, and a flat load counts on vmcnt AND lgkmcnt: waiting for its data then waits for every
// LDS atomic the wave has in flight.  These say "global" explicitly.
__device__ __forceinline__ uint32_t fx_ld_global4(const uint8_t* p) {
  return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>((uintptr_t)p);
}

typedef mm_u4 __attribute__((aligned(2))) fx_u4_unaligned;

// (the prefetched 16-byte values stay ext-vector typed end to end: as a struct of four components they are four
// separate registers to the compiler, which then copies parts of a load's result around — waiting for the load first)
__device__ __forceinline__ mm_u4 fx_ld_global16_u(const uint8_t* p) {  // 16 bytes at 2-byte alignment, global address space
  return *reinterpret_cast<const __attribute__((address_space(1))) fx_u4_unaligned*>((uintptr_t)p);
}
__device__ __forceinline__ mm_u4 fx_ld_global16(const uint8_t* p) {  // 16-byte aligned
  return *reinterpret_cast<const __attribute__((address_space(1))) mm_u4*>((uintptr_t)p);
}
__device__ __forceinline__ uint32_t fx_win(const uint4& w, int k) {  // k-th 16-bit entry of a window index
  const uint32_t d = k < 2 ? w.x : k < 4 ? w.y : k < 6 ? w.z : w.w;
  return (k & 1) ? d >> 16 : d & 0xFFFFu;
}
__device__ __forceinline__ uint32_t fx_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct alignas(16) FxItem {
  uint32_t lo, hi;  // address of the item's first value
  uint32_t nv;      // values of the item (1 .. 128); 0: no item
  uint32_t rowoff;  // byte offset of the item's row inside a stage buffer
};

struct alignas(16) FxProg {          // the work lists of one (shard, tile, container slot)
  uint4 row[kFxNR][2];               // [0] = {payload address lo, hi, len, type}; [1] = window index
  uint32_t ibase[kFxStages], icnt[kFxStages];  // this unit's items of stage q: items[ibase[q] .. ibase[q] + icnt[q])
  uint8_t bml[72], runl[72], bigl[72];
  uint32_t nbm, nrun, nbig, active;  // active = 0: the filter has no container in this slot (nothing can intersect)
};
constexpr int kFxProgU4 = (int)(sizeof(FxProg) / 16);  // 149
static_assert(sizeof(FxProg) % 16 == 0 && kFxProgU4 <= 192, "FxProg is copied global -> LDS by three 16-byte DMA instructions of one wave");

// ---- the program of a query: one wave per (shard, tile, slot) ------------------------------------------------
// lane l stands for matrix row l (0..31 = A rows i0.., 32..63 = B rows j0..), the filter row (64) is wave-uniform.
// cursor[0] = the next free item (bump allocation, one atomic per unit), cursor[1] = 1 when `cap` items were not enough
// (cannot happen with the caller's bound; the unit then gets no array items and the flag fails the call).
template <bool HAS_F>
__global__ void __launch_bounds__(256) k_fused_program(
    const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA, const uint4* __restrict__ winA, const uint32_t* __restrict__ rowsA, uint32_t nA,
    const Slot* __restrict__ slotsB, const uint8_t* __restrict__ arenaB, const uint4* __restrict__ winB, const uint32_t* __restrict__ rowsB, uint32_t nBtot,
    const Slot* __restrict__ slotsF, const uint8_t* __restrict__ arenaF, const uint4* __restrict__ winF, const uint32_t* __restrict__ rowsF, uint32_t n_shards,
    FxProg* __restrict__ prog, FxItem* __restrict__ items, u64* __restrict__ cursor, u64 cap) {
  const int lane = threadIdx.x & 63;
  const uint32_t agroups = (nA + 31) / 32, btiles = (nBtot + 31) / 32;
  const uint64_t unit = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= (uint64_t)n_shards * agroups * btiles * kSlots) return;
  const uint32_t slot = (uint32_t)(unit % kSlots);
  uint64_t b = unit / kSlots;
  const uint32_t bt = (uint32_t)(b % btiles);
  b /= btiles;
  const uint32_t ag = (uint32_t)(b % agroups);
  const uint32_t shard = (uint32_t)(b / agroups);
  const uint32_t i0 = ag * 32, j0 = bt * 32;
  FxProg& P = prog[unit];
  const u64 lane_lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  // ---- descriptors ----
  Slot d;
  d.off = 0, d.len = 0, d.tn = 0;
  Slot df = d;
  uint4 w = uint4{0, 0, 0, 0}, wf = w;
  const uint8_t* my_base = lane < 32 ? arenaA : arenaB;
  if (lane < 32) {
    if (i0 + lane < nA) {
      const uint64_t rr = (uint64_t)rowsA[(uint64_t)shard * nA + i0 + lane] * kSlots + slot;
      d = slotsA[rr];
      if (winA) w = winA[rr];
    }
  } else if (j0 + lane - 32 < nBtot) {
    const uint64_t rr = (uint64_t)rowsB[(uint64_t)shard * nBtot + j0 + lane - 32] * kSlots + slot;
    d = slotsB[rr];
    if (winB) w = winB[rr];
  }
  if (HAS_F) {
    const uint64_t rf = (uint64_t)rowsF[shard] * kSlots + slot;
    df = slotsF[rf];
    if (winF) wf = winF[rf];
  }
  if (HAS_F && slot_n(df) == 0) {  // a nil filter container annihilates the slot: the kernel skips it
    if (lane < kFxStages) P.ibase[lane] = 0, P.icnt[lane] = 0;
    if (lane == 0) P.nbm = 0, P.nrun = 0, P.nbig = 0, P.active = 0;
    return;
  }
  // ---- the row table and the bitmap / run / long-array lists ----
  const uint32_t type = slot_n(d) ? slot_type(d) : 0u;
  const uint32_t typeF = (HAS_F && slot_n(df)) ? slot_type(df) : 0u;  // wave-uniform
  const uintptr_t pa = (uintptr_t)my_base + d.off;  // (integer arithmetic: a shadow descriptor's offset reaches into another allocation)
  const uintptr_t pf = HAS_F ? (uintptr_t)arenaF + df.off : 0;
  P.row[lane][0] = uint4{(uint32_t)pa, (uint32_t)(pa >> 32), d.len, type};
  P.row[lane][1] = w;
  if (lane == 0) {
    P.row[64][0] = uint4{(uint32_t)pf, (uint32_t)((u64)pf >> 32), HAS_F ? df.len : 0u, typeF};
    P.row[64][1] = wf;
  }
  const bool isbig = type == kTypeArray && d.len > (uint32_t)kFxItemArrayMax;
  const u64 mb = __ballot(type == kTypeBitmap), mr = __ballot(type == kTypeRun), mg = __ballot(isbig);
  if (type == kTypeBitmap) P.bml[__popcll(mb & lane_lt)] = (uint8_t)lane;
  if (type == kTypeRun) P.runl[__popcll(mr & lane_lt)] = (uint8_t)lane;
  if (isbig) P.bigl[__popcll(mg & lane_lt)] = (uint8_t)lane;
  if (lane == 0) {
    uint32_t nb = __popcll(mb), nr = __popcll(mr), ng = __popcll(mg);
    if (typeF == kTypeBitmap) P.bml[nb++] = 64;
    if (typeF == kTypeRun) P.runl[nr++] = 64;
    if (typeF == kTypeArray && df.len > (uint32_t)kFxItemArrayMax) P.bigl[ng++] = 64;
    P.nbm = nb, P.nrun = nr, P.nbig = ng, P.active = 1;
  }
  // ---- array items: per stage, row after row (the order the kernel deals them to its groups), the filter row's last ----
  const bool isarr = type == kTypeArray && d.len <= (uint32_t)kFxItemArrayMax;
  const bool farr = HAS_F && typeF == kTypeArray && df.len <= (uint32_t)kFxItemArrayMax;
  uint32_t st[kFxStages], cnt[kFxStages], nch[kFxStages], incl[kFxStages], tot[kFxStages], stF[kFxStages], cntF[kFxStages], nchF[kFxStages];
  uint32_t total = 0;
#pragma unroll
  for (int q = 0; q < kFxStages; ++q) {
    st[q] = fx_win(w, q);
    const uint32_t en = q + 1 < kFxStages ? fx_win(w, q + 1) : d.len;
    cnt[q] = (isarr && en > st[q]) ? min(en - st[q], (uint32_t)kFxItemArrayMax) : 0u;
    nch[q] = (cnt[q] + 127u) >> 7;
    incl[q] = wave_incl_scan(nch[q]);
    tot[q] = (uint32_t)__builtin_amdgcn_readlane((int)incl[q], 63);
    stF[q] = cntF[q] = nchF[q] = 0;
    if (farr) {
      stF[q] = fx_win(wf, q);
      const uint32_t enF = q + 1 < kFxStages ? fx_win(wf, q + 1) : df.len;
      cntF[q] = enF > stF[q] ? min(enF - stF[q], (uint32_t)kFxItemArrayMax) : 0u;
      nchF[q] = (cntF[q] + 127u) >> 7;
    }
    total += tot[q] + nchF[q];
  }
  u64 base = 0;
  if (lane == 0 && total) base = atomicAdd(cursor, (u64)total);
  base = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
  const bool fits = base + total <= cap && base + total <= 0xFFFFFFFFull;
  if (!fits && lane == 0) atomicOr(cursor + 1, 1ull);
  uint32_t at = (uint32_t)base;
#pragma unroll
  for (int q = 0; q < kFxStages; ++q) {
    const uint32_t n = fits ? tot[q] + nchF[q] : 0u;
    if (lane == 0) P.ibase[q] = at, P.icnt[q] = n;
    if (fits) {
      const uint32_t mine = at + incl[q] - nch[q];
      for (uint32_t c = 0; c < nch[q]; ++c) {
        const uintptr_t p = pa + 2u * (uintptr_t)(st[q] + 128u * c);
        items[mine + c] = FxItem{(uint32_t)p, (uint32_t)((u64)p >> 32), min(128u, cnt[q] - 128u * c), (uint32_t)lane * (uint32_t)kFxStride};
      }
      if ((uint32_t)lane < nchF[q]) {  // the filter row's items (wave-uniform quantities; lane c writes chunk c: at most 32)
        const uintptr_t p = pf + 2u * (uintptr_t)(stF[q] + 128u * (uint32_t)lane);
        items[at + tot[q] + lane] = FxItem{(uint32_t)p, (uint32_t)((u64)p >> 32), min(128u, cntF[q] - 128u * (uint32_t)lane), 64u * (uint32_t)kFxStride};
      }
    }
    at += n;
  }
}

}  // namespace fbk
