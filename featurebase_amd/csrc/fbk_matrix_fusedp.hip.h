// fbk_matrix_fusedp.hip.h — k_count_matrix_fused (fbk_matrix_fused.hip.h) driven by a PREPARED PROGRAM (round 5).
//
// Round 4's counters said what that kernel is short of: not memory, not one pipe — the vector ALU is 52 % busy, the LDS
// 46 %, the matrix cores 36 % — its waves WAIT 54 % of their cycles: sixteen waves in lock step, and every producer wave
// walks a chain of dependent LDS round trips per stage (item -> row table -> address; bitmap list -> row table ->
// address; list lengths ...) before its loads can even go out, while nine of them rebuild, in every block of every query,
// work lists that depend only on (the batches' descriptors, the query's row lists).  Here those are computed ONCE per
// (batch versions, row lists) by k_fused_program:
//   * FxProg, one per (shard, 32 x 32 tile, container slot): the row table, the bitmap / run / long-array lists and the
//     per-stage item counts — what build_rows / build_count made in LDS, now 2384 bytes in global memory that ONE wave
//     copies into LDS with three global->LDS DMA instructions per slot (no registers, no vector instructions);
//   * FxItem, the RESOLVED array items of every stage: {address of the item's first value, number of values, byte offset
//     of its row in a stage buffer}, 16 bytes each, in the order the kernel deals them to its 48 sixteen-lane groups.  A
//     group's item of stage t + 2 is loaded during stage t (one 16-byte load, the same address in the group's lanes), its
//     values during stage t + 1, scattered in stage t + 2: no LDS read and ~6 vector instructions per item where
//     fetch_item had two dependent LDS reads and ~20;
//   * a wave's first BPREF bitmap rows of a slot are looked up once per slot (wave-uniform registers), a stage adds 1 KiB.
// The consumers, the stage buffers, the scatter, the run / long-array paths are the ones of fbk_matrix_fused.hip.h.
#pragma once
#include "fbk_matrix_fused.hip.h"

namespace fbk {

struct alignas(16) FxItem {
  uint32_t lo, hi;  // address of the item's first value
  uint32_t nv;      // values of the item (1 .. 128); 0: no item
  uint32_t rowoff;  // byte offset of the item's row inside a stage buffer
};

struct alignas(16) FxProg {          // the work lists of one (shard, tile, container slot), see FxTab
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
  // ---- the row table and the bitmap / run / long-array lists (build_rows of fbk_matrix_fused.hip.h) ----
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

// ---- the kernel ------------------------------------------------------------------------------------------------
template <bool HAS_F, int APREF = 2, int BPREF = 2, bool PROF = false>
__global__ void __launch_bounds__(kFxWaves * 64) k_count_matrix_fusedp(const FxProg* __restrict__ prog, const FxItem* __restrict__ items, uint32_t nA, uint32_t nBtot,
                                                                      uint32_t n_shards, uint32_t spb, u64* __restrict__ out_shard, u64* __restrict__ prof = nullptr, uint32_t ablate = 0) {
  // `ablate` (option matrix_fused_ablate, experiment builds only — results are wrong when set): 1 no consumer arithmetic,
  // 2 no array items, 8 no bitmap rows, 16 the producers only keep the barriers
#ifndef FBK_EXPERIMENTS
  ablate = 0;  // (the option does not exist in the product library: the branches on it fold away)
#endif
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int kFxPref = APREF, kFxBmPref = BPREF;
  // PROF (experiment builds, scripts/fused_prof.py): the block in the middle of the grid writes cycle stamps, prof[(wave * 24 + stage) * 8 + k]:
  // producers k = 0 stage start, 1 this stage's loads settled and bitmap rows stored, 2 next stage's loads issued, 3 array items
  // done, 4 run rows done, 5 past the barrier; consumers k = 0 start, 1 arithmetic done, 5 past the barrier
  const bool traced = PROF && blockIdx.x == (gridDim.x / 2 | 1u);
  auto stamp = [&](uint32_t st, int k) {
    if (PROF && traced && (threadIdx.x & 63) == 0 && st < 24u) prof[((threadIdx.x >> 6) * 24u + st) * 8u + k] = (u64)__builtin_readcyclecounter();
  };
  __shared__ uint4 ring[2 * kFxBuf / 16];  // 135 200 bytes
  __shared__ FxProg tabs[2];               // 2 x 2384 bytes
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t agroups = (nA + 31) / 32, btiles = (nBtot + 31) / 32, sgroups = kSlots / spb;
  uint32_t b = xcd_swizzle(blockIdx.x, gridDim.x);  // (a shard's slot groups and tiles on one XCD: they share the program's lines)
  const uint32_t bt = b % btiles;
  b /= btiles;
  const uint32_t ag = b % agroups;
  b /= agroups;
  const uint32_t sg = b % sgroups;
  const uint32_t shard = b / sgroups;
  if (shard >= n_shards) return;
  const uint32_t i0 = ag * 32, j0 = bt * 32;
  const FxProg* const bprog = prog + (((uint64_t)shard * agroups + ag) * btiles + bt) * kSlots;  // the 16 slot programs of this (shard, tile)

  // slots of this block at which anything can intersect, 4 bits each (slot_of(i) = the i-th of them)
  u64 actp = 0;
  uint32_t n_act = 0;
  for (uint32_t s = sg * spb; s < (sg + 1) * spb; ++s)
    if (bprog[s].active) actp |= (u64)s << (4u * n_act++);
  auto slot_of = [&](uint32_t i) { return (uint32_t)(actp >> (4u * i)) & 15u; };
  const uint32_t n_stage = n_act * kFxStages;
  uint8_t* const ring8 = reinterpret_cast<uint8_t*>(&ring[0]);
  // both stage buffers start clean (afterwards the consumers clean what they have read)
  for (uint32_t i = threadIdx.x; i < (uint32_t)(2 * kFxBuf / 16); i += kFxWaves * 64) ring[i] = uint4{0, 0, 0, 0};

  if (wv < kFxConsumers) {
    // ============================== consumers (as in fbk_matrix_fused.hip.h) ==============================
    const uint32_t r = lane & 31, g = lane >> 5;
    mm_v16f acc0{}, acc1{}, acc2{};
    constexpr uint32_t M4 = 0x11111111u;
    __syncthreads();  // (the producers' set-up barrier)
    for (uint32_t it = 0; it <= n_stage; ++it) {
      stamp(it, 0);
      if (it >= 1 && !(ablate & 1u)) {
        uint4* buf = ring + ((it - 1) & 1u) * (uint32_t)(kFxBuf / 16);
        uint4* rowA = buf + r * (uint32_t)(kFxStride / 16) + 16u * (uint32_t)wv + g;
        uint4* rowB = rowA + 32 * (kFxStride / 16);
        uint4* rowF = buf + 64 * (kFxStride / 16) + 16u * (uint32_t)wv;
        auto ld = [&](int o, uint4& va, uint4& vb, uint4& vf) {
          va = rowA[2 * o];
          vb = rowB[2 * o];
          if (HAS_F) vf = rowF[2 * o + g];
          rowA[2 * o] = uint4{0, 0, 0, 0};
          rowB[2 * o] = uint4{0, 0, 0, 0};
        };
        auto octet = [&](const uint4& va, const uint4& vb, const uint4& vf) {
          uint32_t a[4] = {va.x, va.y, va.z, va.w};
          const uint32_t bb[4] = {vb.x, vb.y, vb.z, vb.w};
          const uint32_t f[4] = {vf.x, vf.y, vf.z, vf.w};
#pragma unroll
          for (int d = 0; d < 4; ++d) a[d] = HAS_F ? (a[d] & f[d]) : a[d];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            mm_v8i oa, ob;
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
        if (HAS_F) rowF[lane & 15] = uint4{0, 0, 0, 0};
      }
      stamp(it, 1);
      __syncthreads();
      stamp(it, 5);
    }
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
  (void)lane_lt;
  const uint32_t gl8 = 8u * gl, gl16 = 16u * gl, lane16 = 16u * (uint32_t)lane;

  // the work lists of a slot: global -> LDS, three DMA instructions of ONE wave (the data lands with the wave's later
  // loads: vector memory returns in order, so the `settle` of the wave's next stage has seen it arrive)
  auto dma_table = [&](FxProg& dst, const FxProg* src) {
    const uint8_t* gsrc = reinterpret_cast<const uint8_t*>(src) + lane16;
    uint8_t* l = reinterpret_cast<uint8_t*>(&dst);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (64 * k + lane < kFxProgU4) __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + 1024 * k), (lptr_t)(l + 1024 * k), 16, 0, 0);
  };

  struct Pre {
    mm_u4 a_w[kFxPref];        // array items: 8 values of this lane
    uint32_t a_nv[kFxPref];    //   how many of them exist (0: this lane has nothing)
    uint32_t a_off[kFxPref];   //   byte offset of the item's row inside a stage buffer
    mm_u4 b_w[kFxBmPref];      // bitmap rows: this lane's 16 bytes of the stage's KiB
    uint32_t b_off[kFxBmPref];  //   byte offset of the row; ~0u: none
    uint32_t r_iv[kFxRunPref];  // run rows: run (i0 + lane) of the stage
    uint32_t r_i0[kFxRunPref], r_i1[kFxRunPref];
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
  };
  clear_pre(P0);
  clear_pre(P1);
  // the group's items of the stage after the one whose values are being loaded: {address lo, hi, values, row offset}
  mm_u4 E[kFxPref];
#pragma unroll
  for (int k = 0; k < kFxPref; ++k) E[k] = mm_u4{0, 0, 0, 0};
  // this wave's first BPREF bitmap rows of the slot whose stages are being loaded (wave-uniform)
  uint32_t bm_lo[kFxBmPref], bm_hi[kFxBmPref], bm_off[kFxBmPref];
#pragma unroll
  for (int k = 0; k < kFxBmPref; ++k) bm_lo[k] = bm_hi[k] = 0, bm_off[k] = ~0u;
  uint32_t s_nrun = 0;  // run rows of that slot

  auto scatter8 = [&](const mm_u4& w, uint32_t nv, uint32_t rowaddr) {
    if (nv == 0) return;
    const uint32_t ww[4] = {w[0], w[1], w[2], w[3]};
    const uint32_t valid = (1u << nv) - 1u;  // nv <= 8
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t d = ww[k >> 1];
      uint32_t word, addr;
      if (k & 1) asm("v_bfe_u32 %0, %1, 21, 8" : "=v"(word) : "v"(d));
      else asm("v_bfe_u32 %0, %1, 5, 8" : "=v"(word) : "v"(d));
      asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(addr) : "v"(word), "v"(rowaddr));
      const uint32_t sh = (k & 1) ? d >> 16 : d;
      atomicOr(reinterpret_cast<uint32_t*>(ring8 + addr), __builtin_amdgcn_ubfe(valid, (uint32_t)k, 1u) << (sh & 31u));
    }
  };
  auto row_ptr = [&](const FxProg& T, uint32_t row, uint32_t& len) {
    const uint4 rt = T.row[row][0];
    len = rt.z;
    return reinterpret_cast<const uint8_t*>(((uintptr_t)rt.y << 32) | rt.x);
  };
  auto win_of = [&](const FxProg& T, uint32_t row, uint32_t k) { return (uint32_t) reinterpret_cast<const uint16_t*>(&T.row[row][1])[k]; };
  auto run_range = [&](const FxProg& T, uint32_t row, uint32_t q, uint32_t len, uint32_t& i0, uint32_t& i1) {
    i0 = win_of(T, row, q);
    i1 = q + 1 < (uint32_t)kFxStages ? min(win_of(T, row, q + 1) + 1u, len) : len;
  };
  // E <- the group's items of stage `it` (nv = 0: none)
  auto load_entries = [&](uint32_t it) {
    const uint32_t si = it / kFxStages, q = it % kFxStages;
    const FxProg& T = tabs[si & 1u];
    const uint32_t n = T.icnt[q], ib = T.ibase[q];
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) {
      const uint32_t idx = first_group + gq + (uint32_t)kFxGroups * k;
      E[k] = mm_u4{0, 0, 0, 0};
      if (idx < n && !(ablate & 2u)) E[k] = fx_ld_global16(reinterpret_cast<const uint8_t*>(items + ((u64)ib + idx)));
    }
  };
  // the slot whose first stage is `it`: this wave's first bitmap rows and the slot's run-row count
  auto enter_slot = [&](uint32_t it) {
    const FxProg& T = tabs[(it / kFxStages) & 1u];
    const uint32_t nbm = T.nbm;
    s_nrun = T.nrun;
#pragma unroll
    for (int k = 0; k < kFxBmPref; ++k) {
      const uint32_t e = pw + (uint32_t)kFxProducers * k;
      bm_off[k] = ~0u;
      if (e < nbm) {
        const uint32_t row = T.bml[e];
        const uint4 rt = T.row[row][0];
        bm_lo[k] = rt.x, bm_hi[k] = rt.y, bm_off[k] = row * (uint32_t)kFxStride;
      }
    }
  };
  // loads of stage `it`: the values of the items in E (loaded a stage ago), the bitmap KiBs, the first runs of the run
  // rows; then E <- the items of stage it + 1
  auto prefetch = [&](uint32_t it, Pre& P) {
    const uint32_t si = it / kFxStages, q = it % kFxStages;
    const FxProg& T = tabs[si & 1u];
    if (q == 0) enter_slot(it);
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) {
      const int mine = (int)E[k][2] - (int)gl8;  // values of the item from this lane's first on
      P.a_nv[k] = 0;
      if (mine > 0 && !(ablate & 2u)) {
        P.a_nv[k] = (uint32_t)min(mine, 8);
        P.a_off[k] = E[k][3];
        const uint8_t* p = reinterpret_cast<const uint8_t*>(((uintptr_t)E[k][1] << 32) | E[k][0]);
        P.a_w[k] = fx_ld_global16_u(p + gl16);
      }
    }
#pragma unroll
    for (int k = 0; k < kFxBmPref; ++k) {
      P.b_off[k] = (ablate & 8u) ? ~0u : bm_off[k];
      if (bm_off[k] != ~0u && !(ablate & 8u)) {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(((uintptr_t)bm_hi[k] << 32) | bm_lo[k]);
        P.b_w[k] = fx_ld_global16(p + (q * (uint32_t)kFxSB + lane16));
      }
    }
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k) {
      const uint32_t e = (uint32_t)(kFxProducers - 1) - pw + (uint32_t)kFxProducers * k;
      P.r_i0[k] = P.r_i1[k] = 0;
      if (e < s_nrun) {
        const uint32_t row = T.runl[e];
        uint32_t len;
        const uint8_t* p = row_ptr(T, row, len);
        run_range(T, row, q, len, P.r_i0[k], P.r_i1[k]);
        const uint32_t idx = P.r_i0[k] + (uint32_t)lane;
        P.r_iv[k] = idx < len ? fx_ld_global4(p + 4u * idx) : 0u;
        P.r_row[k] = row;
      }
    }
    if (it + 1 < n_stage) load_entries(it + 1);
    else {
#pragma unroll
      for (int k = 0; k < kFxPref; ++k) E[k] = mm_u4{0, 0, 0, 0};
    }
  };
  auto run_toggles = [&](const FxProg& T, uint32_t row, uint32_t q, uint32_t bufoff, uint32_t i0, uint32_t i1, bool have_first, uint32_t first_iv) {
    const uint32_t lo = q * (uint32_t)(kFxSB * 8), hi = lo + (uint32_t)(kFxSB * 8);
    const uint32_t rowaddr = bufoff + row * (uint32_t)kFxStride;
    auto toggle = [&](uint32_t idx, uint32_t iv) {
      const uint32_t s = iv & 0xFFFFu, l = iv >> 16;
      if (idx < i1 && s < hi && l >= lo) {
        const uint32_t s2 = (s > lo ? s : lo) - lo;
        const uint32_t e2 = (l + 1u < hi ? l + 1u : hi) - lo;
        atomicXor(reinterpret_cast<uint32_t*>(ring8 + rowaddr + ((s2 >> 3) & 0x3FCu)), 1u << (s2 & 31u));
        if (e2 < (uint32_t)(kFxSB * 8)) atomicXor(reinterpret_cast<uint32_t*>(ring8 + rowaddr + ((e2 >> 3) & 0x3FCu)), 1u << (e2 & 31u));
      }
    };
    uint32_t base = i0;
    if (have_first) {
      toggle(i0 + (uint32_t)lane, first_iv);
      base += 64u;
    }
    if (base < i1) {
      uint32_t len;
      const uint8_t* p = row_ptr(T, row, len);
      for (; base < i1; base += 64u) {
        const uint32_t idx = base + (uint32_t)lane;
        toggle(idx, idx < len ? fx_ld_global4(p + 4u * idx) : 0u);
      }
    }
  };
  auto prefix_of = [&](const uint4& tv) {
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
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) asm volatile("" : "+v"(E[k]));
  };
  // one stage: `cur` was loaded during the previous stage, `nxt` is loaded now for the next one
  auto stage = [&](uint32_t it, Pre& cur, Pre& nxt) {
    const uint32_t si = it / kFxStages, q = it % kFxStages;
    const FxProg& T = tabs[si & 1u];
    const uint32_t bufoff = (it & 1u) * (uint32_t)kFxBuf;
    stamp(it, 0);
    // ---- 0. this stage's loads and the next stage's items (all issued a stage ago) have landed ----
    settle(cur);
    // ---- 1. bitmap rows: registers -> LDS ----
#pragma unroll
    for (int k = 0; k < kFxBmPref; ++k)
      if (cur.b_off[k] != ~0u) *reinterpret_cast<mm_u4*>(ring8 + (bufoff + lane16) + cur.b_off[k]) = cur.b_w[k];
    stamp(it, 1);
    // ---- 1b. the work lists of the next slot: DMA at (si, 1) — BEFORE this stage's loads, so that they return after it —,
    //          landed by this wave's settle of (si, 2), visible to the block after that stage's barrier, read from (si, 6) on ----
    if (q == 1 && pw == 8u && si + 1 < n_act) dma_table(tabs[(si + 1) & 1u], bprog + slot_of(si + 1));
    // ---- 2. the next stage's loads go out ----
    if (it + 1 < n_stage) prefetch(it + 1, nxt);
    //      a wave's later bitmap rows (more than 12 x BPREF bitmap rows among the 65) are loaded in place
    if (cur.b_off[kFxBmPref - 1] != ~0u) {
      const uint32_t nbm = T.nbm;
      constexpr int kMore = (kFxNR + kFxProducers - 1) / kFxProducers - kFxBmPref;
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
    // ---- 3a. run rows, step 1 ----
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k)
      if (cur.r_i0[k] < cur.r_i1[k]) run_toggles(T, cur.r_row[k], q, bufoff, cur.r_i0[k], cur.r_i1[k], true, cur.r_iv[k]);
    // ---- 3. array items: the prefetched ones, then (long lists only) the rest, items and values loaded in place ----
#pragma unroll
    for (int k = 0; k < kFxPref; ++k) scatter8(cur.a_w[k], cur.a_nv[k], bufoff + cur.a_off[k]);
    {
      const uint32_t n = (ablate & 2u) ? 0u : fx_uniform(T.icnt[q]);
      if (first_group + (uint32_t)kFxGroups * kFxPref < n) {
        const uint32_t ib = fx_uniform(T.ibase[q]);
        for (uint32_t x = first_group + (uint32_t)kFxGroups * kFxPref; x < n; x += (uint32_t)kFxGroups) {
          if (x + gq < n) {
            const mm_u4 e = fx_ld_global16(reinterpret_cast<const uint8_t*>(items + ((u64)ib + x + gq)));
            const int mine = (int)e[2] - (int)gl8;
            if (mine > 0) {
              const uint8_t* p = reinterpret_cast<const uint8_t*>(((uintptr_t)e[1] << 32) | e[0]);
              scatter8(fx_ld_global16_u(p + gl16), (uint32_t)min(mine, 8), bufoff + e[3]);
            }
          }
        }
      }
    }
    // ---- 3b. arrays longer than 4096 values: one row per wave pass ----
    {
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
    // ---- 4. run rows, step 2: the parity prefixes; then a wave's later run rows ----
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < kFxRunPref; ++k)
      if (cur.r_i0[k] < cur.r_i1[k]) run_prefix(cur.r_row[k], bufoff);
    if (cur.r_i1[kFxRunPref - 1] != 0) {
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
    stamp(it, 4);
  };

  // ---- set-up: the work lists of the first slot, the items of stage 0, then the stage loop ----
  if (n_stage && pw == 8u) {
    dma_table(tabs[0], bprog + slot_of(0));
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the table is in LDS before the barrier publishes it
  }
  __syncthreads();  // the work lists and the clean ring are visible
  if (n_stage) {
    load_entries(0);
    prefetch(0, P0);
  }
  for (uint32_t it = 0; it <= n_stage; it += 2) {  // (n_stage is a multiple of 8)
    if (it < n_stage && !(ablate & 16u)) stage(it, P0, P1);
    __syncthreads();
    stamp(it, 5);
    if (it + 1 <= n_stage) {
      if (it + 1 < n_stage && !(ablate & 16u)) stage(it + 1, P1, P0);
      __syncthreads();
      stamp(it + 1, 5);
    }
  }
  __syncthreads();  // the consumers' reduction barrier
}

}  // namespace fbk
