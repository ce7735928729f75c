// fbk_matrix_fused.hip.h — the many-row IntersectionCount matrix (GroupBy / TopN shape,
// executor.go:8880-8934, 2705-2774) for rows in ANY encoding, on the matrix cores, decoding the
// rows inside the kernel (no decoded row ever goes to HBM).
//
//   out[shard][i][j] += sum over the block's slots of |A[shard][i] ∩ F[shard] ∩ B[shard][j]|
//
// What bounds this kernel is instruction ISSUE, not memory: a SIMD issues about one instruction per ~4.9
// cycles over its four waves whatever the number of active lanes (rocprofv3 counters and the kernel's own
// cycle stamps, profiles/r02_pmc_fused2_*.txt, profiles/r02_fused2_*_cycle_stamps.txt).  The first version of
// the in-kernel decode (round 2, 863 us on 256 shards of config 3's rows) walked every array with a cursor,
// re-derived its windows and ran half-empty passes: ~185 instructions per (row, stage).  This one (394 us) is
// built around the instruction count:
//
//   * the batch carries a WINDOW INDEX (k_window_index, 16 bytes per container): where each eighth
//     of the value range begins inside an array / run list.  Nothing is searched, walked or
//     re-derived here;
//   * one block = (shard, slot group, 32 A rows, 32 B rows), 16 wavefronts: 4 CONSUMERS (one per
//     SIMD; bit -> FP4 nibble by ONE v_and per operand dword, v_mfma_scale_f32_32x32x64_f8f6f4, see
//     fbk_matrix_mfma.hip.h) and 12 PRODUCERS that decode;
//   * a stage = the 8192 bit positions [q * 8192, (q + 1) * 8192) of all 65 rows (32 A, 32 B, the
//     filter) as 1 KiB of bitmap per row in LDS (row stride 1040 bytes: the 16 rows of a
//     ds_read_b128 lane group fall into 16 different bank groups); two stages alternate, one
//     barrier per stage;
//   * once per container slot nine producer waves (one per eighth + one for the row table) turn the 65
//     descriptors + window indexes into work lists in LDS: for every stage the list of ARRAY ITEMS (row, first value index, <= 128
//     values), the bitmap rows, the run rows.  Array items of a stage are dealt out to the 48
//     16-lane groups of the producers round robin — a heavy row is spread over several groups, four
//     rows are decoded per wave pass, every lane holds 8 values (one 16-byte load at 2-byte
//     alignment, exactly the stage's values: no window test, only "k < valid");
//   * array bits are OR-ed in with LDS atomics, so any group may write any row; the CONSUMERS zero
//     the piece of every row they have just read (the buffer is clean when the producers get it
//     back), which removes the ordering "zero before scatter" between producer waves;
//   * (round 3, tried and dropped: walking the six address chains of a stage's prefetch level by level — 5 LDS round
//     trips instead of 16 — with clamped, branch-free indexes, and loading 128 runs ahead instead of 64.  418 us against
//     397: the clamps and selects are vector instructions, and issue slots, not LDS latency, are what this kernel is short
//     of.  profiles/r03_fused_prefetch_ab.txt)
//   * every global load is issued a WHOLE stage ahead (two register sets alternate: items, bitmap KiBs
//     and run windows of stage t + 1 go out at the start of stage t), so a stage never waits for HBM;
//   * bitmap rows: the q-th KiB of the container, global -> registers (a stage ahead) -> LDS;
//     run rows (owned by one wave each): toggles at the clamped start / one past the clamped end
//     (issued before the wave's array items), then a parity prefix over the row's 1 KiB (issued after
//     them: the LDS round trip in between is covered) — runToBitmap, roaring.go:3792.
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
constexpr int kFxGroups = kFxProducers * 4;  // 16-lane groups
// (template parameters of the kernel: APREF = array items per 16-lane group and stage that are loaded a stage ahead — 48 groups:
// 96 per stage with 2, longer lists load in place; BPREF = bitmap rows per wave whose KiB is loaded a stage ahead — 12 waves: 24
// rows with 2.  Encoded rows as uploaded: 2 / 2.  Rows whose heavy containers have a dense shadow (fbk.hip heavy_shadow): mostly
// bitmap rows and short arrays, 1 / 3.)
constexpr int kFxRunPref = 2;                // run rows per wave whose first 64 runs are loaded a stage ahead
constexpr int kFxItemArrayMax = 4096;        // arrays up to this length go through the item lists (ArrayMaxSize, roaring.go:46)
constexpr int kFxItemCap = 2624;             // >= 65 rows x (4096 / 128 + 8) items per container slot

struct FxTab {                      // the work lists of one container slot
  uint4 row[kFxNR][2];              // [0] = {payload address lo, hi, len, type}; [1] = window index
  uint32_t pool[kFxItemCap];        // array items of the 8 stages: row | first value << 7 | (values - 1) << 19
  uint32_t ibase[kFxStages], icnt[kFxStages];
  uint8_t bml[72], runl[72], bigl[72];  // rows holding bitmaps / runs / arrays longer than kFxItemArrayMax
  uint32_t nbm, nrun, nbig, pad;
};

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
__device__ __forceinline__ uint32_t fx_win_dyn(const uint4& w, uint32_t k) {
  const uint32_t d = k < 2 ? w.x : k < 4 ? w.y : k < 6 ? w.z : w.w;
  return (k & 1) ? d >> 16 : d & 0xFFFFu;
}
__device__ __forceinline__ uint32_t fx_wave_incl_scan(uint32_t v) { return wave_incl_scan(v); }
__device__ __forceinline__ uint32_t fx_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <bool HAS_F, bool PROF = false, int APREF = 2, int BPREF = 2>
__global__ void __launch_bounds__(kFxWaves * 64) k_count_matrix_fused(
    const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA, const uint4* __restrict__ winA, const uint32_t* __restrict__ rowsA, uint32_t nA,
    const Slot* __restrict__ slotsB, const uint8_t* __restrict__ arenaB, const uint4* __restrict__ winB, const uint32_t* __restrict__ rowsB, uint32_t nBtot,
    const Slot* __restrict__ slotsF, const uint8_t* __restrict__ arenaF, const uint4* __restrict__ winF, const uint32_t* __restrict__ rowsF, uint32_t n_shards,
    uint32_t spb, u64* __restrict__ out_shard, uint32_t ablate, u64* __restrict__ prof = nullptr) {
  // `ablate` (option matrix_fused_ablate, timing experiments only — results are wrong when set):
  // 1 no consumer arithmetic, 2 no array decode, 4 no run decode, 8 no bitmap rows
  // PROF (ablate & 32): the block in the middle of the grid writes cycle stamps, prof[(wave * 24 + stage) * 8 + k]:
  // producers k = 0 stage start, 1 this stage's loads settled and bitmap rows stored, 2 next stage's loads issued,
  // 3 array items done, 4 run rows (and list building) done, 5 past the barrier; consumers k = 0 start,
  // 1 arithmetic done, 5 past the barrier
  constexpr int kFxPref = APREF, kFxBmPref = BPREF;
#ifndef FBK_EXPERIMENTS
  ablate = 0;  // experiment builds only (the option does not exist in the product library): the branches on it fold away
#endif
  const bool traced = PROF && blockIdx.x == (gridDim.x / 2 | 1u);
  auto stamp = [&](uint32_t st, int k) {
    if (PROF && traced && (threadIdx.x & 63) == 0 && st < 24u) prof[((threadIdx.x >> 6) * 24u + st) * 8u + k] = (u64)__builtin_readcyclecounter();
  };
  __shared__ uint4 ring[2 * kFxBuf / 16];  // 135 200 bytes
  __shared__ FxTab tabs[2];                // 2 x 12 896 bytes
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t agroups = (nA + 31) / 32, btiles = (nBtot + 31) / 32, sgroups = kSlots / spb;
  uint32_t b = xcd_swizzle(blockIdx.x, gridDim.x);  // (a shard's slot groups and tiles on one XCD: they share the descriptor lines)
  const uint32_t bt = b % btiles;
  b /= btiles;
  const uint32_t ag = b % agroups;
  b /= agroups;
  const uint32_t sg = b % sgroups;
  const uint32_t shard = b / sgroups;
  if (shard >= n_shards) return;
  const uint32_t i0 = ag * 32, j0 = bt * 32;

  // slots of this block at which anything can intersect (a nil filter container annihilates the slot)
  uint32_t act[kSlots];
  uint32_t n_act = 0;
  for (uint32_t s = sg * spb; s < (sg + 1) * spb; ++s) {
    bool on = true;
    if (HAS_F) on = slot_n(slotsF[(uint64_t)rowsF[shard] * kSlots + s]) != 0;
    if (on) act[n_act++] = s;
  }
  const uint32_t n_stage = n_act * kFxStages;
  uint8_t* const ring8 = reinterpret_cast<uint8_t*>(&ring[0]);
  // both stage buffers start clean (afterwards the consumers clean what they have read)
  for (uint32_t i = threadIdx.x; i < (uint32_t)(2 * kFxBuf / 16); i += kFxWaves * 64) ring[i] = uint4{0, 0, 0, 0};

  if (wv < kFxConsumers) {
    // ============================== consumers ==============================
    const uint32_t r = lane & 31, g = lane >> 5;
    mm_v16f acc0{}, acc1{}, acc2{};
    constexpr uint32_t M4 = 0x11111111u;
    __syncthreads();  // (the producers' two set-up barriers)
    __syncthreads();
    for (uint32_t it = 0; it <= n_stage; ++it) {
      stamp(it, 0);
      if (it >= 1 && !(ablate & 1u)) {
        // rows as uint4 pieces (row stride 65 pieces): this wave's K range is pieces 16 wv .. 16 wv + 15 of
        // every row; pair o = pieces 16 wv + 2 o + g
        uint4* buf = ring + ((it - 1) & 1u) * (uint32_t)(kFxBuf / 16);
        uint4* rowA = buf + r * (uint32_t)(kFxStride / 16) + 16u * (uint32_t)wv + g;
        uint4* rowB = rowA + 32 * (kFxStride / 16);
        uint4* rowF = buf + 64 * (kFxStride / 16) + 16u * (uint32_t)wv;
        auto ld = [&](int o, uint4& va, uint4& vb, uint4& vf) {
          va = rowA[2 * o];
          vb = rowB[2 * o];
          if (HAS_F) vf = rowF[2 * o + g];
          // clean behind the read (LDS operations of one wave execute in order)
          rowA[2 * o] = uint4{0, 0, 0, 0};
          rowB[2 * o] = uint4{0, 0, 0, 0};
        };  // (no branch in here: the optimiser sinks the arithmetic of all eight octets below the last conditional block)
        auto octet = [&](const uint4& va, const uint4& vb, const uint4& vf) {
          uint32_t a[4] = {va.x, va.y, va.z, va.w};
          const uint32_t bb[4] = {vb.x, vb.y, vb.z, vb.w};
          const uint32_t f[4] = {vf.x, vf.y, vf.z, vf.w};
#pragma unroll
          for (int d = 0; d < 4; ++d) a[d] = HAS_F ? (a[d] & f[d]) : a[d];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            mm_v8i oa, ob;  // the instruction reads the first four registers of an FP4 operand
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              oa[d] = (int)(k < 3 ? (a[d] & (M4 << k)) : ((a[d] >> 3) & M4));
              ob[d] = (int)(k < 3 ? (bb[d] & (M4 << k)) : ((bb[d] >> 3) & M4));
            }
            if (k == 0 || k == 3) acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc0, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else if (k == 1) acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc2, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
          }
        };
        uint4 xa, xb, xf = uint4{0, 0, 0, 0}, ya, yb, yf = uint4{0, 0, 0, 0};
        // (the scheduler would otherwise hoist all 24 reads of the stage to the top: 96 registers, spills)
        ld(0, xa, xb, xf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 0; o < 8; o += 2) {
          ld(o + 1, ya, yb, yf);
          __builtin_amdgcn_sched_barrier(0);
          octet(xa, xb, xf);
          __builtin_amdgcn_sched_barrier(0);
          if (o + 2 < 8) ld(o + 2, xa, xb, xf);
          __builtin_amdgcn_sched_barrier(0);
          octet(ya, yb, yf);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the filter row's 16 pieces of this wave's K range
        // (all lanes, four per piece, the same zeros: a condition here would put the arithmetic above behind it)
        if (HAS_F) rowF[lane & 15] = uint4{0, 0, 0, 0};
      }
      stamp(it, 1);
      __syncthreads();
      stamp(it, 5);
    }
    // cross-wave reduction through LDS (the ring is free now): [wave][16 regs][64 lanes]
    uint32_t* red = reinterpret_cast<uint32_t*>(ring8);
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(wv * 16 + q) * 64 + lane] = (uint32_t)(acc0[q] * 4.0f + acc1[q] + acc2[q] * 0.25f + 0.5f);
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < 16 / kFxConsumers; ++qq) {
      const int q = wv * (16 / kFxConsumers) + qq;
      uint32_t tot = 0;
#pragma unroll
      for (int w = 0; w < kFxConsumers; ++w) tot += red[(w * 16 + q) * 64 + lane];
      const uint32_t i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), j = lane & 31;
      if (i0 + i < nA && j0 + j < nBtot && tot) atomicAdd(&out_shard[((uint64_t)shard * nA + i0 + i) * nBtot + j0 + j], (u64)tot);
    }
    return;
  }

  // ============================== producers ==============================
  const uint32_t pw = (uint32_t)wv - kFxConsumers;  // 0..11
  const uint32_t gq = lane >> 4, gl = lane & 15;
  const uint32_t first_group = 4u * pw;  // array items of a stage: item x goes to group x mod 48
  const u64 lane_lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  // ---- descriptor sources of the set-up wave: lane l stands for matrix row l (0..31 A, 32..63 B) ----
  const Slot* my_slots = nullptr;
  const uint4* my_win = nullptr;
  const uint8_t* my_base = lane < 32 ? arenaA : arenaB;
  if (lane < 32) {
    if (i0 + lane < nA) {
      const uint64_t rr = rowsA[(uint64_t)shard * nA + i0 + lane];
      my_slots = slotsA + rr * kSlots;
      my_win = winA ? winA + rr * kSlots : nullptr;
    }
  } else if (j0 + lane - 32 < nBtot) {
    const uint64_t rr = rowsB[(uint64_t)shard * nBtot + j0 + lane - 32];
    my_slots = slotsB + rr * kSlots;
    my_win = winB ? winB + rr * kSlots : nullptr;
  }
  const uint64_t rowF = HAS_F ? (uint64_t)rowsF[shard] * kSlots : 0;
  struct Desc {
    Slot d;
    uint4 w;
    Slot df;
    uint4 wf;
  };
  auto load_desc = [&](uint32_t slot) {
    Desc x;
    x.d.off = 0, x.d.len = 0, x.d.tn = 0;
    x.df = x.d;
    x.w = uint4{0, 0, 0, 0};
    x.wf = x.w;
    if (my_slots) {
      x.d = my_slots[slot];
      if (my_win) x.w = my_win[slot];
    }
    if (HAS_F) {
      x.df = slotsF[rowF + slot];
      if (winF) x.wf = winF[rowF + slot];
    }
    return x;
  };
  // ---- the work lists of one container slot, built by nine waves in three steps a stage apart: (1) the
  //      descriptors are fetched; (2) wave w < 8 counts the items of eighth w, wave 8 writes the row table
  //      and the bitmap / run / long-array lists; (3) wave w adds up the counts before it and writes its
  //      items.  (One wave doing all of it took 9000 cycles — longer than a whole stage of the block.) ----
  struct Build {
    uint32_t st, cnt, nch, incl, tot;  // this lane's row in the wave's eighth: first value, values, items, inclusive scan, wave total
    uint32_t stF, cntF, nchF;          // the filter row (wave-uniform)
  };
  auto build_rows = [&](FxTab& T, const Desc& x) {
    const uint32_t type = slot_n(x.d) ? slot_type(x.d) : 0u;
    const uint32_t typeF = (HAS_F && slot_n(x.df)) ? slot_type(x.df) : 0u;  // wave-uniform
    const uintptr_t pa = (uintptr_t)my_base + x.d.off;  // (integer arithmetic: a shadow descriptor's offset reaches into another allocation, fbk.hip heavy_shadow)
    T.row[lane][0] = uint4{(uint32_t)pa, (uint32_t)(pa >> 32), x.d.len, type};
    T.row[lane][1] = x.w;
    if (HAS_F && lane == 0) {
      const uintptr_t pf = (uintptr_t)arenaF + x.df.off;
      T.row[64][0] = uint4{(uint32_t)pf, (uint32_t)(pf >> 32), x.df.len, typeF};
      T.row[64][1] = x.wf;
    }
    const bool isbig = type == kTypeArray && x.d.len > (uint32_t)kFxItemArrayMax;
    const u64 mb = __ballot(type == kTypeBitmap), mr = __ballot(type == kTypeRun), mg = __ballot(isbig);
    if (type == kTypeBitmap) T.bml[__popcll(mb & lane_lt)] = (uint8_t)lane;
    if (type == kTypeRun) T.runl[__popcll(mr & lane_lt)] = (uint8_t)lane;
    if (isbig) T.bigl[__popcll(mg & lane_lt)] = (uint8_t)lane;
    uint32_t nb = __popcll(mb), nr = __popcll(mr), ng = __popcll(mg);
    if (lane == 0) {
      if (typeF == kTypeBitmap) T.bml[nb++] = 64;
      if (typeF == kTypeRun) T.runl[nr++] = 64;
      if (typeF == kTypeArray && x.df.len > (uint32_t)kFxItemArrayMax) T.bigl[ng++] = 64;
      T.nbm = nb, T.nrun = nr, T.nbig = ng;
    }
  };
  auto build_count = [&](FxTab& T, const Desc& x, uint32_t w, Build& B) {  // w: wave-uniform
    const uint32_t type = slot_n(x.d) ? slot_type(x.d) : 0u;
    const uint32_t typeF = (HAS_F && slot_n(x.df)) ? slot_type(x.df) : 0u;
    const bool isarr = type == kTypeArray && x.d.len <= (uint32_t)kFxItemArrayMax;
    const bool farr = typeF == kTypeArray && x.df.len <= (uint32_t)kFxItemArrayMax;
    B.st = fx_win_dyn(x.w, w);
    const uint32_t en = w + 1 < (uint32_t)kFxStages ? fx_win_dyn(x.w, w + 1) : x.d.len;
    B.cnt = (isarr && en > B.st) ? min(en - B.st, (uint32_t)kFxItemArrayMax) : 0u;
    B.nch = (B.cnt + 127u) >> 7;
    B.incl = fx_wave_incl_scan(B.nch);
    B.tot = (uint32_t)__builtin_amdgcn_readlane((int)B.incl, 63);
    B.stF = B.cntF = B.nchF = 0;
    if (HAS_F && farr) {
      B.stF = fx_win_dyn(x.wf, w);
      const uint32_t enF = w + 1 < (uint32_t)kFxStages ? fx_win_dyn(x.wf, w + 1) : x.df.len;
      B.cntF = enF > B.stF ? min(enF - B.stF, (uint32_t)kFxItemArrayMax) : 0u;
      B.nchF = (B.cntF + 127u) >> 7;
    }
    if (lane == 0) T.icnt[w] = B.tot + B.nchF;
  };
  auto build_items = [&](FxTab& T, uint32_t w, const Build& B) {
    uint32_t base = 0;
    for (uint32_t v = 0; v < w; ++v) base += T.icnt[v];  // (counts of the earlier eighths: written a stage / a barrier ago)
    base = min(base, (uint32_t)kFxItemCap);
    const uint32_t at = base + B.incl - B.nch;
    for (uint32_t c = 0; __ballot(c < B.nch) != 0; ++c)
      if (c < B.nch && at + c < (uint32_t)kFxItemCap) T.pool[at + c] = (uint32_t)lane | ((B.st + 128u * c) << 7) | ((min(128u, B.cnt - 128u * c) - 1u) << 19);
    // the filter row's items (wave-uniform quantities; lane c writes chunk c: at most 32 chunks + 1)
    if (HAS_F && (uint32_t)lane < B.nchF && base + B.tot + lane < (uint32_t)kFxItemCap)
      T.pool[base + B.tot + lane] = 64u | ((B.stF + 128u * lane) << 7) | ((min(128u, B.cntF - 128u * lane) - 1u) << 19);
    // (the lists fit by construction; the clamps only keep a corrupt index inside the pool.  icnt[w] itself stays
    // as counted: the waves of the later eighths are reading it)
    if (lane == 0) T.ibase[w] = base;
  };

  // ---- per-stage state: the loads of a stage, issued a WHOLE stage before they are used ----
  // (Issued at the end of the previous stage they would be exposed on the slowest wave of every stage
  // — the one that reaches the barrier last and starts the next stage at once: measured, the first
  // cut of this kernel spent 2 us per stage that way.)  Two sets alternate.
  // Wave-uniform quantities (row offsets, run ranges, list lengths) stay in VECTOR registers, the same
  // value in every lane, and conditions on them are exec masks: moving them to scalar registers costs a
  // v_readfirstlane each plus v_readlane / v_writelane spills (there are not enough scalar registers
  // for two sets) — the vector ALU is what bounds this kernel, and that bookkeeping was a third of it.
  struct Pre {
    mm_u4 a_w[kFxPref];        // array items: 8 values of this lane
    uint32_t a_nv[kFxPref];    //   how many of them exist (0: this lane has nothing)
    uint32_t a_off[kFxPref];   //   byte offset of the item's row inside a stage buffer
    uint32_t n_items, item_base;  // the stage's item list
    mm_u4 b_w[kFxBmPref];      // bitmap rows: this lane's 16 bytes of the stage's KiB
    uint32_t b_off[kFxBmPref];  //   byte offset of the row; ~0u: none
    uint32_t r_iv[kFxRunPref];  // run rows: run (i0 + lane) of the stage
    uint32_t r_i0[kFxRunPref], r_i1[kFxRunPref];  //   the runs [i0, i1) can intersect the stage (i0 == i1: nothing to do)
    uint32_t r_row[kFxRunPref];
  };
  Pre P0, P1;
  auto clear_pre = [&](Pre& P) {
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) P.a_w[k] = mm_u4{0, 0, 0, 0}, P.a_nv[k] = 0, P.a_off[k] = 0;
#pragma unroll
    for (int k = 0; k < kFxBmPref; ++k) P.b_w[k] = mm_u4{0, 0, 0, 0}, P.b_off[k] = ~0u;
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k) P.r_iv[k] = 0, P.r_i0[k] = 0, P.r_i1[k] = 0, P.r_row[k] = 0;
    P.n_items = P.item_base = 0;
  };
  clear_pre(P0);
  clear_pre(P1);
  const uint32_t gl8 = 8u * gl, gl16 = 16u * gl, lane16 = 16u * (uint32_t)lane;

  // one array item of group (first_group + gq): fetch the lane's 8 values
  auto fetch_item = [&](const FxTab& T, uint32_t idx, uint32_t n, uint32_t ib, mm_u4& w, uint32_t& nv, uint32_t& off) {
    nv = 0;
    if (idx < n) {
      const uint32_t it = T.pool[ib + idx];
      const uint32_t row = it & 127u;
      const uint4 rt = T.row[row][0];
      off = row * (uint32_t)kFxStride;
      const int mine = (int)__builtin_amdgcn_ubfe(it, 19u, 7u) + 1 - (int)gl8;  // values of the item from this lane's first on
      if (mine > 0) {
        nv = (uint32_t)min(mine, 8);
        const uint8_t* p = reinterpret_cast<const uint8_t*>(((uintptr_t)rt.y << 32) | rt.x);
        w = fx_ld_global16_u(p + (((it >> 6) & 0x1FFEu) + gl16));  // 2 * (first value of the item + 8 * lane)
      }
    }
  };
  // 8 values of one lane -> bits of a row of the stage buffer (values are inside the stage by construction).
  // No branches and no exec juggling per value: a slot past the lane's last value ORs a zero mask into the row
  // (4 vector instructions + the LDS atomic per value: bfe + shift-add for the address, bfe + shift for the mask).
  auto scatter8 = [&](const mm_u4& w, uint32_t nv, uint32_t rowaddr) {
    // Lanes without any value are switched off for the whole pass: left on, their zero-mask atomics all
    // go to ONE address and the LDS serialises same-address atomics (measured: SQ_LDS_ADDR_CONFLICT
    // 62.8 M quad-cycles per launch, the LDS 80 % busy, 250 us of 630).
    if (nv == 0) return;
    const uint32_t ww[4] = {w[0], w[1], w[2], w[3]};
    const uint32_t valid = (1u << nv) - 1u;  // nv <= 8
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t d = ww[k >> 1];
      // (value >> 5) & 255 = the dword inside the stage's KiB.  Written as the two instructions it should be:
      // the compiler turns bfe + shift-add into shift + and + add
      uint32_t word, addr;
      if (k & 1) asm("v_bfe_u32 %0, %1, 21, 8" : "=v"(word) : "v"(d));
      else asm("v_bfe_u32 %0, %1, 5, 8" : "=v"(word) : "v"(d));
      asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(addr) : "v"(word), "v"(rowaddr));
      const uint32_t sh = (k & 1) ? d >> 16 : d;  // (the shifter reads the low 5 bits only)
      atomicOr(reinterpret_cast<uint32_t*>(ring8 + addr), __builtin_amdgcn_ubfe(valid, (uint32_t)k, 1u) << (sh & 31u));
    }
  };
  auto row_ptr = [&](const FxTab& T, uint32_t row, uint32_t& len) {  // payload address and length of a row (the same in all lanes)
    const uint4 rt = T.row[row][0];
    len = rt.z;
    return reinterpret_cast<const uint8_t*>(((uintptr_t)rt.y << 32) | rt.x);
  };
  // k-th entry of a row's window index, straight from the work lists
  auto win_of = [&](const FxTab& T, uint32_t row, uint32_t k) {
    return (uint32_t) reinterpret_cast<const uint16_t*>(&T.row[row][1])[k];
  };
  // the runs [i0, i1) of a run row can intersect stage q
  auto run_range = [&](const FxTab& T, uint32_t row, uint32_t q, uint32_t len, uint32_t& i0, uint32_t& i1) {
    i0 = win_of(T, row, q);                                                            // first run whose last value is >= lo
    i1 = q + 1 < (uint32_t)kFxStages ? min(win_of(T, row, q + 1) + 1u, len) : len;       // one past the last run that can start below hi
  };
  // loads of stage `it` (slot si, eighth q): array items, bitmap KiBs, the first runs of the run rows
  auto prefetch = [&](uint32_t it, Pre& P) {
    const uint32_t si = it / kFxStages, q = it % kFxStages;
    const FxTab& T = tabs[si & 1u];
    P.item_base = T.ibase[q];  // (<= kFxItemCap)
    P.n_items = min(T.icnt[q], (uint32_t)kFxItemCap - P.item_base);  // (fits by construction; the clamp keeps a corrupt window index inside the pool)
    const uint32_t nbm = T.nbm, nrun = T.nrun;
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) {
      P.a_nv[k] = 0;
      if (!(ablate & 2u)) fetch_item(T, first_group + gq + (uint32_t)kFxGroups * k, P.n_items, P.item_base, P.a_w[k], P.a_nv[k], P.a_off[k]);
    }
#pragma unroll
    for (int k = 0; k < kFxBmPref; ++k) {
      P.b_off[k] = ~0u;
      const uint32_t e = pw + (uint32_t)kFxProducers * k;
      if (e < nbm && !(ablate & 8u)) {
        const uint32_t row = T.bml[e];
        uint32_t len;
        const uint8_t* p = row_ptr(T, row, len);
        P.b_w[k] = fx_ld_global16(p + (q * (uint32_t)kFxSB + lane16));
        P.b_off[k] = row * (uint32_t)kFxStride;
      }
    }
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k) {
      const uint32_t e = (uint32_t)(kFxProducers - 1) - pw + (uint32_t)kFxProducers * k;
      P.r_i0[k] = P.r_i1[k] = 0;
      if (e < nrun && !(ablate & 4u)) {
        const uint32_t row = T.runl[e];
        uint32_t len;
        const uint8_t* p = row_ptr(T, row, len);
        run_range(T, row, q, len, P.r_i0[k], P.r_i1[k]);
        const uint32_t idx = P.r_i0[k] + (uint32_t)lane;
        P.r_iv[k] = idx < len ? fx_ld_global4(p + 4u * idx) : 0u;
        P.r_row[k] = row;
      }
    }
  };
  // A run row of a stage, in two steps: (1) toggles at the clamped start and one past the clamped end of the runs
  // [i0, i1) (the first 64 of them were loaded a stage ahead); (2) the parity prefix over the row's 1 KiB.  The two
  // steps are issued apart — toggles of all the wave's run rows, then the array items, then the prefixes — so that
  // the LDS round trip between a row's atomics and the read-back of its words is covered by other work.
  auto run_toggles = [&](const FxTab& T, uint32_t row, uint32_t q, uint32_t bufoff, uint32_t i0, uint32_t i1, bool have_first, uint32_t first_iv) {
    const uint32_t lo = q * (uint32_t)(kFxSB * 8), hi = lo + (uint32_t)(kFxSB * 8);
    const uint32_t rowaddr = bufoff + row * (uint32_t)kFxStride;
    auto toggle = [&](uint32_t idx, uint32_t iv) {
      const uint32_t s = iv & 0xFFFFu, l = iv >> 16;
      if (idx < i1 && s < hi && l >= lo) {
        const uint32_t s2 = (s > lo ? s : lo) - lo;           // 0 .. 8191
        const uint32_t e2 = (l + 1u < hi ? l + 1u : hi) - lo;  // 1 .. 8192
        atomicXor(reinterpret_cast<uint32_t*>(ring8 + rowaddr + ((s2 >> 3) & 0x3FCu)), 1u << (s2 & 31u));
        if (e2 < (uint32_t)(kFxSB * 8)) atomicXor(reinterpret_cast<uint32_t*>(ring8 + rowaddr + ((e2 >> 3) & 0x3FCu)), 1u << (e2 & 31u));
      }
    };
    uint32_t base = i0;
    if (have_first) {
      toggle(i0 + (uint32_t)lane, first_iv);
      base += 64u;
    }
    if (base < i1) {  // more than 64 runs inside one eighth of the container (or a row beyond the prefetched two)
      uint32_t len;
      const uint8_t* p = row_ptr(T, row, len);
      for (; base < i1; base += 64u) {
        const uint32_t idx = base + (uint32_t)lane;
        toggle(idx, idx < len ? fx_ld_global4(p + 4u * idx) : 0u);
      }
    }
  };
  auto prefix_of = [&](const uint4& tv) {  // parity prefix over the row, lane j holding its bytes 16 j .. 16 j + 15
    const u64 t0 = ((u64)tv.y << 32) | tv.x, t1 = ((u64)tv.w << 32) | tv.z;
    const uint32_t p0 = __popcll(t0) & 1u, p1 = __popcll(t1) & 1u;
    const u64 mm = __ballot((p0 ^ p1) != 0);
    const uint32_t in = __popcll(mm & lane_lt) & 1u;
    const u64 f0 = prefix_xor64(t0) ^ (in ? ~0ull : 0ull);
    const u64 f1 = prefix_xor64(t1) ^ ((in ^ p0) ? ~0ull : 0ull);
    return uint4{(uint32_t)f0, (uint32_t)(f0 >> 32), (uint32_t)f1, (uint32_t)(f1 >> 32)};
  };
  auto run_prefix = [&](uint32_t row, uint32_t bufoff) {
    uint4* pc = reinterpret_cast<uint4*>(ring8 + (bufoff + row * (uint32_t)kFxStride) + lane16);
    *pc = prefix_of(*pc);
  };
  auto settle = [&](Pre& P) {
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) asm volatile("" : "+v"(P.a_w[k]));
#pragma unroll
    for (int k = 0; k < kFxBmPref; ++k) asm volatile("" : "+v"(P.b_w[k]));
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k) asm volatile("" : "+v"(P.r_iv[k]));
  };
  Desc next_d = {};
  Build bld = {};
  // one stage: `cur` was loaded during the previous stage, `nxt` is loaded now for the next one
  auto stage = [&](uint32_t it, Pre& cur, Pre& nxt) {
    const uint32_t si = it / kFxStages, q = it % kFxStages;
    const FxTab& T = tabs[si & 1u];
    const uint32_t bufoff = (it & 1u) * (uint32_t)kFxBuf;
    stamp(it, 0);
    // ---- 0. this stage's loads (issued a stage ago) have landed: ONE wait for all of them here.  The compiler
    //         cannot count what is in flight across the conditional loads, so wherever it waits it waits for
    //         everything — after the next stage's loads have gone out that would be their full latency ----
    settle(cur);
    // ---- 1. bitmap rows: registers -> LDS ----
#pragma unroll
    for (int k = 0; k < kFxBmPref; ++k)
      if (cur.b_off[k] != ~0u) *reinterpret_cast<mm_u4*>(ring8 + (bufoff + lane16) + cur.b_off[k]) = cur.b_w[k];
    stamp(it, 1);
    // ---- 2. the next stage's loads go out ----
    if (it + 1 < n_stage) prefetch(it + 1, nxt);
    //      a wave's third and later bitmap rows (more than 24 bitmap rows among the 65) are loaded in place,
    //      all of them before the first is stored
    if (cur.b_off[kFxBmPref - 1] != ~0u && !(ablate & 8u)) {
      const uint32_t nbm = T.nbm;
      constexpr int kMore = (kFxNR + kFxProducers - 1) / kFxProducers - kFxBmPref;  // 4
      mm_u4 t[kMore];
      uint32_t toff[kMore];
#pragma unroll
      for (int k = 0; k < kMore; ++k) {
        const uint32_t e = pw + (uint32_t)kFxProducers * (kFxBmPref + k);
        toff[k] = ~0u;
        if (e < nbm) {
          const uint32_t row = T.bml[e];
          uint32_t len;
          const uint8_t* p = row_ptr(T, row, len);
          t[k] = fx_ld_global16(p + (q * (uint32_t)kFxSB + lane16));
          toff[k] = row * (uint32_t)kFxStride;
        }
      }
#pragma unroll
      for (int k = 0; k < kMore; ++k)
        if (toff[k] != ~0u) *reinterpret_cast<mm_u4*>(ring8 + (bufoff + lane16) + toff[k]) = t[k];
    }
    stamp(it, 2);
    // ---- 3a. run rows, step 1 (each run row is owned by one wave, so its parity prefix follows this wave's own toggles) ----
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k)
      if (cur.r_i0[k] < cur.r_i1[k]) run_toggles(T, cur.r_row[k], q, bufoff, cur.r_i0[k], cur.r_i1[k], true, cur.r_iv[k]);
    // ---- 3. array items: the prefetched ones, then (long lists only) the rest ----
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) scatter8(cur.a_w[k], cur.a_nv[k], bufoff + cur.a_off[k]);
    if (first_group + (uint32_t)kFxGroups * kFxPref < cur.n_items && !(ablate & 2u)) {
      const uint32_t n = fx_uniform(cur.n_items), ib = fx_uniform(cur.item_base);
      for (uint32_t x = first_group + (uint32_t)kFxGroups * kFxPref; x < n; x += (uint32_t)kFxGroups) {
        mm_u4 w;
        uint32_t nv, off;
        fetch_item(T, x + gq, n, ib, w, nv, off);
        scatter8(w, nv, bufoff + off);
      }
    }
    // ---- 3b. arrays longer than 4096 values (never produced by optimize(); uploads may hold them): one row per wave pass ----
    if (!(ablate & 2u)) {
      const uint32_t nbig = fx_uniform(T.nbig);
      for (uint32_t e = pw; e < nbig; e += (uint32_t)kFxProducers) {
        const uint32_t row = fx_uniform(T.bigl[e]);
        uint32_t len;
        const uint8_t* p = row_ptr(T, row, len);
        const uint32_t v0 = fx_uniform(win_of(T, row, q)), v1 = q + 1 < (uint32_t)kFxStages ? fx_uniform(min(win_of(T, row, q + 1), len)) : fx_uniform(len);
        for (uint32_t base = v0; base < v1; base += 512u) {
          const uint32_t mine = base + 8u * (uint32_t)lane;
          if (mine < v1) scatter8(fx_ld_global16_u(p + 2u * mine), min(v1 - mine, 8u), bufoff + row * (uint32_t)kFxStride);
        }
      }
    }
    stamp(it, 3);
    // ---- 4. run rows, step 2: the parity prefixes; then (a wave with a second run row may have a third) the rest ----
    wave_lds_sync();
    // (reading both rows back before computing either prefix was tried: 25 us slower)
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k)
      if (cur.r_i0[k] < cur.r_i1[k]) run_prefix(cur.r_row[k], bufoff);
    if (cur.r_i1[kFxRunPref - 1] != 0 && !(ablate & 4u)) {
      const uint32_t nrun = fx_uniform(T.nrun);
      for (uint32_t e = (uint32_t)(kFxProducers - 1) - pw + (uint32_t)kFxProducers * kFxRunPref; e < nrun; e += (uint32_t)kFxProducers) {
        const uint32_t row = fx_uniform(T.runl[e]);
        uint32_t len, i0, i1;
        (void)row_ptr(T, row, len);
        run_range(T, row, q, len, i0, i1);
        if (i0 < i1) {
          run_toggles(T, row, q, bufoff, i0, i1, false, 0u);
          wave_lds_sync();
          run_prefix(row, bufoff);
        }
      }
    }
    // ---- 5. the work lists of slot si + 1 (see build_*): read from the start of stage (si, 7) on ----
    if (si + 1 < n_act && pw <= 8u) {
      FxTab& N = tabs[(si + 1) & 1u];
      if (q == 1) next_d = load_desc(act[si + 1]);
      if (q == 2) {
        if (pw < 8u) build_count(N, next_d, pw, bld);
        else build_rows(N, next_d);
      }
      if (q == 3 && pw < 8u) build_items(N, pw, bld);
    }
    stamp(it, 4);
  };

  // ---- set-up: the work lists of the first slot (the same three steps, a barrier apart), then the stage loop ----
  if (n_stage && pw <= 8u) {
    next_d = load_desc(act[0]);
    if (pw < 8u) build_count(tabs[0], next_d, pw, bld);
    else build_rows(tabs[0], next_d);
  }
  __syncthreads();
  if (n_stage && pw < 8u) build_items(tabs[0], pw, bld);
  __syncthreads();  // the work lists and the clean ring are visible
  if (n_stage) prefetch(0, P0);
  for (uint32_t it = 0; it <= n_stage; it += 2) {  // (n_stage is a multiple of 8)
    if (it < n_stage) stage(it, P0, P1);
    __syncthreads();
    stamp(it, 5);
    if (it + 1 <= n_stage) {
      if (it + 1 < n_stage) stage(it + 1, P1, P0);
      __syncthreads();
      stamp(it + 1, 5);
    }
  }
  __syncthreads();  // the consumers' reduction barrier
}

}  // namespace fbk
