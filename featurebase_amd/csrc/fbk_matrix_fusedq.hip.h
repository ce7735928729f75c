// fbk_matrix_fusedq.hip.h — the count matrix over encoded rows: the kernel that runs the prepared program of
// fbk_matrix_fused.hip.h, with its producer waves SPECIALISED and their loads issued TWO stages ahead (round 5).
//
// What the ablations of the first program-driven form said (twelve identical producer waves; scripts/fused_ablate.py,
// profiles/r05_fused_ablate.jsonl; config 4 as SURVEY 8d
// writes it, 1024 shards): the whole kernel 1135-1200 us; the consumers ALONE (producers reduced to their barriers) 717 us;
// the producers alone 1056 us, 734 without the array items; the bytes at the achievable HBM rate ~700 us.  Every part fits
// the budget, their sum does not overlap: a stage's loads go out at its start and are waited for at the start of the next one,
// so ONE stage of loads (34 KB per CU) is all that is ever in flight — the memory system idles from the moment they have
// landed until the next barrier — and every one of the twelve producer waves carries the bookkeeping of BOTH kinds of rows
// (two item slots, three bitmap rows, two run rows: 120 registers, nothing left for a deeper pipeline).
//
// Here a producer wave has ONE job:
//   * 5 ARRAY waves (20 sixteen-lane groups): the resolved items of the program, kFqAP = 4 per group and stage loaded ahead
//     (80 per stage; configs 3 and 4 have 50-80), scattered with LDS atomics as before;
//   * 7 BITMAP waves: kFqBP = 6 bitmap rows each (42 per slot) global -> registers -> LDS, the run rows (one per wave loaded
//     ahead, the rest in place), the table DMA of the next slot;
// and with half the state per wave each role keeps THREE register sets: the loads of stage t + 2 go out during stage t (the
// items' entries during stage t - 1), two stages of loads are in flight per CU.  The consumers: FP4 operands by one v_and per dword, hand-counted LDS waits, a
// fourth accumulator (the k = 3 product no longer waits for the k = 0 one of the same octet).
#pragma once
#include "fbk_matrix_fused.hip.h"

namespace fbk {

#ifndef FBK_V_FQNA  // (build variants: scripts/build_variant.sh x -DFBK_V_FQNA=6 -DFBK_V_FQBP=7)
#define FBK_V_FQNA 5
#endif
#ifndef FBK_V_FQBP
#define FBK_V_FQBP 6
#endif
constexpr int kFqNA = FBK_V_FQNA;        // array waves: producer waves 0 .. 4
constexpr int kFqNB = kFxProducers - kFqNA;  // bitmap waves: producer waves 5 .. 11
constexpr int kFqAP = 4;                 // array items per group and stage loaded ahead
constexpr int kFqBP = FBK_V_FQBP;        // bitmap rows per bitmap wave and slot loaded ahead
constexpr int kFqGroups = kFqNA * 4;     // 16-lane groups of the array waves: item x of a stage goes to group x mod 20

template <bool HAS_F, bool PROF = false>
__global__ void __launch_bounds__(kFxWaves * 64) k_count_matrix_fusedq(const FxProg* __restrict__ prog, const FxItem* __restrict__ items, uint32_t nA, uint32_t nBtot,
                                                                      uint32_t n_shards, uint32_t spb, u64* __restrict__ out_shard, u64* __restrict__ prof, uint32_t ablate) {
  // `ablate` (option matrix_fused_ablate, experiment builds only — results are wrong when set): 1 no consumer arithmetic,
  // 2 no array items, 8 no bitmap rows, 16 the producers only keep the barriers
#ifndef FBK_EXPERIMENTS
  ablate = 0;
#endif
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const bool traced = PROF && blockIdx.x == (gridDim.x / 2 | 1u);
  auto stamp = [&](uint32_t st, int k) {
    if (PROF && traced && (threadIdx.x & 63) == 0 && st < 24u) prof[((threadIdx.x >> 6) * 24u + st) * 8u + k] = (u64)__builtin_readcyclecounter();
  };
  __shared__ uint4 ring[2 * kFxBuf / 16];  // 135 200 bytes
  __shared__ FxProg tabs[2];               // 2 x 2384 bytes
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t agroups = (nA + 31) / 32, btiles = (nBtot + 31) / 32, sgroups = kSlots / spb;
  uint32_t b = xcd_swizzle(blockIdx.x, gridDim.x);
  const uint32_t bt = b % btiles;
  b /= btiles;
  const uint32_t ag = b % agroups;
  b /= agroups;
  const uint32_t sg = b % sgroups;
  const uint32_t shard = b / sgroups;
  if (shard >= n_shards) return;
  const uint32_t i0 = ag * 32, j0 = bt * 32;
  const FxProg* const bprog = prog + (((uint64_t)shard * agroups + ag) * btiles + bt) * kSlots;
  u64 actp = 0;  // the block's active slots, 4 bits each
  uint32_t n_act = 0;
  for (uint32_t s = sg * spb; s < (sg + 1) * spb; ++s)
    if (bprog[s].active) actp |= (u64)s << (4u * n_act++);
  auto slot_of = [&](uint32_t i) { return (uint32_t)(actp >> (4u * i)) & 15u; };
  const uint32_t n_stage = n_act * kFxStages;
  uint8_t* const ring8 = reinterpret_cast<uint8_t*>(&ring[0]);
  for (uint32_t i = threadIdx.x; i < (uint32_t)(2 * kFxBuf / 16); i += kFxWaves * 64) ring[i] = uint4{0, 0, 0, 0};

  if (wv < kFxConsumers) {
    // ============================== consumers ==============================
    // The LDS reads of an octet are issued a whole octet ahead by hand (asm volatile, hand-counted s_waitcnt, the idiom of
    // fbk_matrix_mfma.hip.h): written as plain C++ the scheduler moved every read down to its first use, and each of the eight
    // octets of a stage then waited out a full LDS round trip — the consumers alone took 1.4 us per stage, more than twice
    // what their 290 vector and 32 matrix instructions need (scripts/fused_ablate.py, profiles/r05_fused_ablate*.jsonl).
    const uint32_t r = lane & 31, g = lane >> 5;
    mm_v16f acc0{}, acc1{}, acc2{}, acc3{};
    constexpr uint32_t M4 = 0x11111111u;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)&ring[0];
    // rows as 16-byte pieces: this wave's K range is pieces 16 wv .. 16 wv + 15 of every row; octet o = pieces 16 wv + 2 o + g
    const uint32_t addrA = lds0 + r * (uint32_t)kFxStride + (16u * (uint32_t)wv + g) * 16u;  // row r of A; row 32 + r of B is 32 strides on
    const uint32_t addrF = lds0 + 64u * (uint32_t)kFxStride + (16u * (uint32_t)wv + g) * 16u;
    const uint32_t addrFz = lds0 + 64u * (uint32_t)kFxStride + (16u * (uint32_t)wv + ((uint32_t)lane & 15u)) * 16u;
    mm_u4 zero4 = mm_u4{0, 0, 0, 0};
    asm volatile("" : "+v"(zero4));
    struct Oct {
      mm_u4 a, b, f;
    };
    // LDS instructions of one octet's loads: the reads and the two clean-up writes behind them.  (Round 6, profiles/r06_fused_lib_ab_prio_nozb.jsonl:
    // a build WITHOUT the clean-up writes — wrong counts — is 2.5-4 % faster, which bounds what type-aware or producer-side zeroing could
    // bring; a ds_write_b128 costs the same 14.7 cycles whatever lanes are enabled (profiles/r06_mfma_valu_overlap.txt), so masking the
    // lanes of bitmap rows buys nothing; s_setprio 1 / 3 on these waves: +-0.5 %.)
    constexpr int kLdOps = (HAS_F ? 3 : 2) + 2;
    constexpr int kBOff = 32 * kFxStride;        // (33 280: the offset field of a DS instruction is 16 bits)
    auto issue = [&](Oct& o, uint32_t bufoff, int t) {
      const uint32_t aa = addrA + bufoff, ff = addrF + bufoff;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o.a) : "v"(aa), "n"(32 * t));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o.b) : "v"(aa), "n"(kBOff + 32 * t));
      if (HAS_F) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o.f) : "v"(ff), "n"(32 * t));
      // clean behind the read (LDS operations of one wave execute in order): the producers get the buffer back zeroed
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(aa), "v"(zero4), "n"(32 * t) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(aa), "v"(zero4), "n"(kBOff + 32 * t) : "memory");
    };
    auto landed = [&](Oct& o, bool more_behind) {  // o's reads are complete (LDS returns in order)
      if (more_behind) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(kLdOps) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(o.a));
      asm volatile("" : "+v"(o.b));
      if (HAS_F) asm volatile("" : "+v"(o.f));
    };
    auto octet = [&](const Oct& o) {
      uint32_t a[4] = {o.a[0], o.a[1], o.a[2], o.a[3]};
      const uint32_t bb[4] = {o.b[0], o.b[1], o.b[2], o.b[3]};
#pragma unroll
      for (int d = 0; d < 4; ++d) a[d] = HAS_F ? (a[d] & o.f[d]) : a[d];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mm_v8i oa, ob;  // the instruction reads the first four registers of an FP4 operand
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          oa[d] = (int)(k < 3 ? (a[d] & (M4 << k)) : ((a[d] >> 3) & M4));
          ob[d] = (int)(k < 3 ? (bb[d] & (M4 << k)) : ((bb[d] >> 3) & M4));
        }
        if (k == 0) acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc0, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        else if (k == 1) acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        else if (k == 2) acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc2, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        else acc3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc3, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    };
    __syncthreads();  // (the producers' set-up barrier)
    for (uint32_t it = 0; it <= n_stage; ++it) {
      stamp(it, 0);
      if (it >= 1 && !(ablate & 1u)) {
        const uint32_t bufoff = ((it - 1) & 1u) * (uint32_t)kFxBuf;
        Oct X, Y;
        X.f = Y.f = mm_u4{0, 0, 0, 0};
        issue(X, bufoff, 0);
#pragma unroll
        for (int o = 0; o < 8; o += 2) {
          issue(Y, bufoff, o + 1);
          landed(X, true);
          octet(X);
          if (o + 2 < 8) issue(X, bufoff, o + 2);
          landed(Y, o + 2 < 8);
          octet(Y);
        }
        // the filter row's 16 pieces of this wave's K range (all lanes, four per piece, the same zeros)
        if (HAS_F) asm volatile("ds_write_b128 %0, %1" ::"v"(addrFz + bufoff), "v"(zero4) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the compiler does not know of the hand-issued writes: they are done before the barrier)
      }
      stamp(it, 1);
      __syncthreads();
      stamp(it, 5);
    }
    // (bit p of a nibble has the FP4 value 0.5 * 2^p for p < 3; bit 3 is shifted down to bit 0 first: products 0.25, 1, 4, 0.25)
    uint32_t* red = reinterpret_cast<uint32_t*>(ring8);
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(wv * 16 + q) * 64 + lane] = (uint32_t)((acc0[q] + acc3[q]) * 4.0f + acc1[q] + acc2[q] * 0.25f + 0.5f);
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
  uint32_t lane16 = 16u * (uint32_t)lane;  // (laundered per stage, see the array waves)
  const u64 lane_lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  auto row_ptr = [&](const FxProg& T, uint32_t row, uint32_t& len) {
    const uint4 rt = T.row[row][0];
    len = rt.z;
    return reinterpret_cast<const uint8_t*>(((uintptr_t)rt.y << 32) | rt.x);
  };
  auto win_of = [&](const FxProg& T, uint32_t row, uint32_t k) { return (uint32_t) reinterpret_cast<const uint16_t*>(&T.row[row][1])[k]; };
  // 8 values of one lane -> bits of a row of the stage buffer (see fbk_matrix_fused.hip.h scatter8)
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

  if (pw < (uint32_t)kFqNA) {
    // ------------------------------ array waves ------------------------------
    const uint32_t gq = lane >> 4, gl = lane & 15;
    uint32_t gl8 = 8u * gl, gl16 = 16u * gl;  // (laundered through an empty asm per stage: otherwise every per-lane address of every one of the three
                                              // instances of the stage is hoisted out of the loop as "invariant", and the register file spills)
    const uint32_t first_group = 4u * pw;
    struct ASet {
      mm_u4 v[kFqAP];      // 8 values of this lane
      uint32_t m[kFqAP];   // how many of them exist | byte offset of the item's row << 4 (0: nothing)
    };
    ASet S0, S1, S2;
#pragma unroll
    for (int k = 0; k < kFqAP; ++k) {
      S0.v[k] = S1.v[k] = S2.v[k] = mm_u4{0, 0, 0, 0};
      S0.m[k] = S1.m[k] = S2.m[k] = 0;
    }
    mm_u4 E[kFqAP];  // the group's items of the stage whose values go out next
#pragma unroll
    for (int k = 0; k < kFqAP; ++k) E[k] = mm_u4{0, 0, 0, 0};
    uint32_t nI0 = 0, nI1 = 0, nI2 = 0, ibI0 = 0, ibI1 = 0, ibI2 = 0;  // item counts / bases of stages it, it + 1, it + 2 (wave-uniform)
    uint32_t s_nbig = 0;
    // E <- the group's items of stage s; its item count / base
    auto load_entries = [&](uint32_t s, uint32_t& n_out, uint32_t& ib_out) {
      const FxProg& T = tabs[(s / kFxStages) & 1u];
      const uint32_t n = (ablate & 2u) ? 0u : T.icnt[s % kFxStages], ib = T.ibase[s % kFxStages];
#pragma unroll
      for (int k = 0; k < kFqAP; ++k) {
        const uint32_t idx = first_group + gq + (uint32_t)kFqGroups * k;
        E[k] = mm_u4{0, 0, 0, 0};
        if (idx < n) E[k] = fx_ld_global16(reinterpret_cast<const uint8_t*>(items + ((u64)ib + idx)));
      }
      n_out = n, ib_out = ib;
    };
    // the values of the items in E
    auto issue_values = [&](ASet& S) {
#pragma unroll
      for (int k = 0; k < kFqAP; ++k) {
        const int mine = (int)E[k][2] - (int)gl8;  // values of the item from this lane's first on
        S.m[k] = 0;
        if (mine > 0) {
          S.m[k] = (uint32_t)min(mine, 8) | (E[k][3] << 4);
          const uint8_t* p = reinterpret_cast<const uint8_t*>(((uintptr_t)E[k][1] << 32) | E[k][0]);
          S.v[k] = fx_ld_global16_u(p + gl16);
        }
      }
    };
    auto astage = [&](uint32_t it, ASet& cur, ASet& fill) {
      const uint32_t si = it / kFxStages, q = it % kFxStages;
      const FxProg& T = tabs[si & 1u];
      const uint32_t bufoff = (it & 1u) * (uint32_t)kFxBuf;
      asm volatile("" : "+v"(gl8), "+v"(gl16));
      stamp(it, 0);
      // this stage's values (issued two stages ago) and the items of stage it + 2 (issued a stage ago) have landed
#pragma unroll
      for (int k = 0; k < kFqAP; ++k) asm volatile("" : "+v"(cur.v[k]));
#pragma unroll
      for (int k = 0; k < kFqAP; ++k) asm volatile("" : "+v"(E[k]));
      stamp(it, 1);
      if (it + 2 < n_stage) issue_values(fill);
      else {
#pragma unroll
        for (int k = 0; k < kFqAP; ++k) fill.m[k] = 0;
      }
      uint32_t n3 = 0, ib3 = 0;
      if (it + 3 < n_stage) load_entries(it + 3, n3, ib3);
      else {
#pragma unroll
        for (int k = 0; k < kFqAP; ++k) E[k] = mm_u4{0, 0, 0, 0};
      }
      stamp(it, 2);
#pragma unroll
      for (int k = 0; k < kFqAP; ++k) scatter8(cur.v[k], cur.m[k] & 15u, bufoff + (cur.m[k] >> 4));
      // a stage with more than 80 items: the rest, items and values loaded in place
      if (first_group + (uint32_t)kFqGroups * kFqAP < nI0) {
        const uint32_t n = fx_uniform(nI0), ib = fx_uniform(ibI0);
        for (uint32_t x = first_group + (uint32_t)kFqGroups * kFqAP; x < n; x += (uint32_t)kFqGroups) {
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
      // arrays longer than 4096 values (never produced by optimize(); uploads may hold them): one row per wave pass
      if (q == 0) s_nbig = (ablate & 2u) ? 0u : fx_uniform(T.nbig);
      for (uint32_t e = pw; e < s_nbig; e += (uint32_t)kFqNA) {
        const uint32_t row = fx_uniform(T.bigl[e]);
        uint32_t len;
        const uint8_t* p = row_ptr(T, row, len);
        const uint32_t v0 = fx_uniform(win_of(T, row, q)), v1 = q + 1 < (uint32_t)kFxStages ? fx_uniform(min(win_of(T, row, q + 1), len)) : fx_uniform(len);
        for (uint32_t base = v0; base < v1; base += 512u) {
          const uint32_t mine = base + 8u * (uint32_t)lane;
          if (mine < v1) scatter8(fx_ld_global16_u(p + 2u * mine), min(v1 - mine, 8u), bufoff + row * (uint32_t)kFxStride);
        }
      }
      nI0 = nI1, ibI0 = ibI1, nI1 = nI2, ibI1 = ibI2, nI2 = n3, ibI2 = ib3;
      stamp(it, 3);
      stamp(it, 4);
    };
    __syncthreads();  // the first slot's work lists and the clean ring are visible
    if (n_stage && !(ablate & 16u)) {  // (n_stage is a multiple of 8: stages 0, 1, 2 exist)
      load_entries(0, nI0, ibI0);
      issue_values(S0);
      load_entries(1, nI1, ibI1);
      issue_values(S1);
      load_entries(2, nI2, ibI2);
    }
    for (uint32_t it = 0; it <= n_stage; it += 3) {  // (three stages per trip: the register sets rotate by NAME, nothing is copied)
      if (it < n_stage && !(ablate & 16u)) astage(it, S0, S2);
      __syncthreads();
      stamp(it, 5);
      if (it + 1 <= n_stage) {
        if (it + 1 < n_stage && !(ablate & 16u)) astage(it + 1, S1, S0);
        __syncthreads();
        stamp(it + 1, 5);
      }
      if (it + 2 <= n_stage) {
        if (it + 2 < n_stage && !(ablate & 16u)) astage(it + 2, S2, S1);
        __syncthreads();
        stamp(it + 2, 5);
      }
    }
    __syncthreads();  // the consumers' reduction barrier
    return;
  }

  // ------------------------------ bitmap waves ------------------------------
  const uint32_t bw = pw - (uint32_t)kFqNA;  // 0..6
  auto dma_table = [&](FxProg& dst, const FxProg* src) {  // the work lists of a slot: global -> LDS, three DMA instructions
    const uint8_t* gsrc = reinterpret_cast<const uint8_t*>(src) + lane16;
    uint8_t* l = reinterpret_cast<uint8_t*>(&dst);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (64 * k + lane < kFxProgU4) __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + 1024 * k), (lptr_t)(l + 1024 * k), 16, 0, 0);
  };
  auto run_range = [&](const FxProg& T, uint32_t row, uint32_t q, uint32_t len, uint32_t& r0, uint32_t& r1) {
    r0 = win_of(T, row, q);
    r1 = q + 1 < (uint32_t)kFxStages ? min(win_of(T, row, q + 1) + 1u, len) : len;
  };
  auto run_toggles = [&](const FxProg& T, uint32_t row, uint32_t q, uint32_t bufoff, uint32_t r0, uint32_t r1, bool have_first, uint32_t first_iv) {
    const uint32_t lo = q * (uint32_t)(kFxSB * 8), hi = lo + (uint32_t)(kFxSB * 8);
    const uint32_t rowaddr = bufoff + row * (uint32_t)kFxStride;
    auto toggle = [&](uint32_t idx, uint32_t iv) {
      const uint32_t s = iv & 0xFFFFu, l = iv >> 16;
      if (idx < r1 && s < hi && l >= lo) {
        const uint32_t s2 = (s > lo ? s : lo) - lo;
        const uint32_t e2 = (l + 1u < hi ? l + 1u : hi) - lo;
        atomicXor(reinterpret_cast<uint32_t*>(ring8 + rowaddr + ((s2 >> 3) & 0x3FCu)), 1u << (s2 & 31u));
        if (e2 < (uint32_t)(kFxSB * 8)) atomicXor(reinterpret_cast<uint32_t*>(ring8 + rowaddr + ((e2 >> 3) & 0x3FCu)), 1u << (e2 & 31u));
      }
    };
    uint32_t base = r0;
    if (have_first) {
      toggle(r0 + (uint32_t)lane, first_iv);
      base += 64u;
    }
    if (base < r1) {
      uint32_t len;
      const uint8_t* p = row_ptr(T, row, len);
      for (; base < r1; base += 64u) {
        const uint32_t idx = base + (uint32_t)lane;
        toggle(idx, idx < len ? fx_ld_global4(p + 4u * idx) : 0u);
      }
    }
  };
  auto run_prefix = [&](uint32_t row, uint32_t bufoff) {  // parity prefix over the row's 1 KiB: runToBitmap, roaring.go:3792
    uint4* pc = reinterpret_cast<uint4*>(ring8 + (bufoff + row * (uint32_t)kFxStride) + lane16);
    const uint4 tv = *pc;
    const u64 t0 = ((u64)tv.y << 32) | tv.x, t1 = ((u64)tv.w << 32) | tv.z;
    const uint32_t p0 = __popcll(t0) & 1u, p1 = __popcll(t1) & 1u;
    const u64 mm = __ballot((p0 ^ p1) != 0);
    const uint32_t in = __popcll(mm & lane_lt) & 1u;
    const u64 f0 = prefix_xor64(t0) ^ (in ? ~0ull : 0ull);
    const u64 f1 = prefix_xor64(t1) ^ ((in ^ p0) ? ~0ull : 0ull);
    *pc = uint4{(uint32_t)f0, (uint32_t)(f0 >> 32), (uint32_t)f1, (uint32_t)(f1 >> 32)};
  };
  struct BSet {
    mm_u4 v[kFqBP];        // this lane's 16 bytes of the stage's KiB of the wave's k-th bitmap row
    uint32_t r_iv, r_i0, r_i1, r_row;  // the wave's first run row: run (r_i0 + lane) of the stage, the runs [r_i0, r_i1) can intersect it
  };
  BSet B0, B1, B2;
  auto clear_set = [&](BSet& S) {
#pragma unroll
    for (int k = 0; k < kFqBP; ++k) S.v[k] = mm_u4{0, 0, 0, 0};
    S.r_iv = S.r_i0 = S.r_i1 = S.r_row = 0;
  };
  clear_set(B0);
  clear_set(B1);
  clear_set(B2);
  // this wave's first kFqBP bitmap rows (wave-uniform): where the rows of the slot whose stages are being LOADED lie in memory,
  // and the stage-buffer offsets of the rows of the even / odd slots (~0u: none) — a stage's rows are its slot's
  uint32_t bm_lo[kFqBP], bm_hi[kFqBP], bmo0[kFqBP], bmo1[kFqBP];
#pragma unroll
  for (int k = 0; k < kFqBP; ++k) bm_lo[k] = bm_hi[k] = 0, bmo0[k] = bmo1[k] = ~0u;
  uint32_t l_nrun = 0;
  uint32_t c_nbm = 0, c_nrun = 0;  // bitmap / run rows of the slot whose stage is being WRITTEN
  auto enter_slot = [&](uint32_t s) {
    const uint32_t par = (s / kFxStages) & 1u;
    const FxProg& T = tabs[par];
    const uint32_t nbm = fx_uniform(T.nbm);
    l_nrun = fx_uniform(T.nrun);
    // LANE k looks up the wave's k-th row (list entry, then the row's address): two LDS round trips for the whole slot.  Row by row —
    // each entry and each address made wave-uniform before the next is asked for — it was twelve, one after the other: 3 000 cycles
    // of the stage that crosses a slot boundary in the INSTRUMENTED build (profiles/r06_fused_cycle_stamps_*.txt).  The product
    // build is bound elsewhere: same counts, +-1 % against the row-by-row form (profiles/r06_fused_lib_ab_roles_enter.jsonl).
    uint32_t row_l = 0, lo_l = 0, hi_l = 0;
    {
      const uint32_t e_l = bw + (uint32_t)kFqNB * (uint32_t)lane;
      if (lane < kFqBP && e_l < nbm) {
        row_l = T.bml[e_l];
        const uint4 rt = T.row[row_l][0];
        lo_l = rt.x, hi_l = rt.y;
      }
    }
#pragma unroll
    for (int k = 0; k < kFqBP; ++k) {
      const uint32_t e = bw + (uint32_t)kFqNB * k;
      uint32_t off = ~0u;
      if (e < nbm) {
        bm_lo[k] = (uint32_t)__builtin_amdgcn_readlane((int)lo_l, k), bm_hi[k] = (uint32_t)__builtin_amdgcn_readlane((int)hi_l, k);
        off = (uint32_t)__builtin_amdgcn_readlane((int)row_l, k) * (uint32_t)kFxStride;
      }
      if (par) bmo1[k] = off;
      else bmo0[k] = off;
    }
  };
  auto row_off = [&](uint32_t s, int k) { return ((s / kFxStages) & 1u) ? bmo1[k] : bmo0[k]; };
  // the loads of stage s into S
  auto issue = [&](uint32_t s, BSet& S) {
    const uint32_t q = s % kFxStages;
    if (q == 0) enter_slot(s);
#pragma unroll
    for (int k = 0; k < kFqBP; ++k) {
      if (row_off(s, k) != ~0u && !(ablate & 8u)) {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(((uintptr_t)bm_hi[k] << 32) | bm_lo[k]);
        S.v[k] = fx_ld_global16(p + (q * (uint32_t)kFxSB + lane16));
      }
    }
    S.r_i0 = S.r_i1 = 0;
    const uint32_t e = (uint32_t)(kFqNB - 1) - bw;
    if (e < l_nrun) {
      const FxProg& T = tabs[(s / kFxStages) & 1u];
      const uint32_t row = T.runl[e];
      uint32_t len;
      const uint8_t* p = row_ptr(T, row, len);
      run_range(T, row, q, len, S.r_i0, S.r_i1);
      const uint32_t idx = S.r_i0 + (uint32_t)lane;
      S.r_iv = idx < len ? fx_ld_global4(p + 4u * idx) : 0u;
      S.r_row = row;
    }
  };
  bool dma_pending = false;
  auto bstage = [&](uint32_t it, BSet& cur, BSet& fill) {
    const uint32_t si = it / kFxStages, q = it % kFxStages;
    const FxProg& T = tabs[si & 1u];
    const uint32_t bufoff = (it & 1u) * (uint32_t)kFxBuf;
    asm volatile("" : "+v"(lane16));
    stamp(it, 0);
    // the next slot's work lists (DMA of stage (si, 1)) are in LDS before this stage's barrier publishes them: they are read from
    // stage (si, 5) on (the items of the stage three ahead).  Explicit: a wave without rows issues no later load to wait for.
    if (q == 2 && dma_pending) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      dma_pending = false;
    }
    // this stage's loads (issued two stages ago) have landed
#pragma unroll
    for (int k = 0; k < kFqBP; ++k) asm volatile("" : "+v"(cur.v[k]));
    asm volatile("" : "+v"(cur.r_iv));
    // bitmap rows: registers -> LDS
#pragma unroll
    for (int k = 0; k < kFqBP; ++k)
      if (row_off(it, k) != ~0u && !(ablate & 8u)) *reinterpret_cast<mm_u4*>(ring8 + (bufoff + lane16) + row_off(it, k)) = cur.v[k];
    stamp(it, 1);
    if (q == 1 && bw == 0u && si + 1 < n_act) {
      dma_table(tabs[(si + 1) & 1u], bprog + slot_of(si + 1));
      dma_pending = true;
    }
    if (q == 0) c_nbm = fx_uniform(T.nbm), c_nrun = fx_uniform(T.nrun);
    if (it + 2 < n_stage) issue(it + 2, fill);
    else clear_set(fill);
    stamp(it, 2);
    // more than kFqNB x kFqBP bitmap rows among the 65: the rest is loaded in place, two rows at a time
    if (c_nbm > (uint32_t)(kFqNB * kFqBP) && !(ablate & 8u)) {
      for (uint32_t e = bw + (uint32_t)(kFqNB * kFqBP); e < c_nbm; e += 2u * (uint32_t)kFqNB) {
        const uint32_t e2 = e + (uint32_t)kFqNB;
        const uint32_t row = fx_uniform(T.bml[e]), row2 = e2 < c_nbm ? fx_uniform(T.bml[e2]) : row;
        uint32_t len;
        const uint8_t* p = row_ptr(T, row, len);
        const uint8_t* p2 = row_ptr(T, row2, len);
        const mm_u4 t = fx_ld_global16(p + (q * (uint32_t)kFxSB + lane16));
        const mm_u4 t2 = fx_ld_global16(p2 + (q * (uint32_t)kFxSB + lane16));
        *reinterpret_cast<mm_u4*>(ring8 + (bufoff + lane16) + row * (uint32_t)kFxStride) = t;
        if (e2 < c_nbm) *reinterpret_cast<mm_u4*>(ring8 + (bufoff + lane16) + row2 * (uint32_t)kFxStride) = t2;
      }
    }
    stamp(it, 3);
    // run rows (each owned by one wave: its parity prefix follows its own toggles): the one loaded ahead, then the rest in place
    if (c_nrun) {
      if (cur.r_i0 < cur.r_i1) {
        run_toggles(T, cur.r_row, q, bufoff, cur.r_i0, cur.r_i1, true, cur.r_iv);
        wave_lds_sync();
        run_prefix(cur.r_row, bufoff);
      }
      for (uint32_t e = (uint32_t)(kFqNB - 1) - bw + (uint32_t)kFqNB; e < c_nrun; e += (uint32_t)kFqNB) {
        const uint32_t row = fx_uniform(T.runl[e]);
        uint32_t len, r0, r1;
        (void)row_ptr(T, row, len);
        run_range(T, row, q, len, r0, r1);
        if (r0 < r1) {
          run_toggles(T, row, q, bufoff, r0, r1, false, 0u);
          wave_lds_sync();
          run_prefix(row, bufoff);
        }
      }
    }
    stamp(it, 4);
  };
  // ---- set-up: the work lists of the first slot, the loads of stages 0 and 1, then the stage loop ----
  if (n_stage && bw == 0u) {
    dma_table(tabs[0], bprog + slot_of(0));
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the table is in LDS before the barrier publishes it
  }
  __syncthreads();
  if (n_stage && !(ablate & 16u)) {
    issue(0, B0);
    issue(1, B1);
  }
  for (uint32_t it = 0; it <= n_stage; it += 3) {
    if (it < n_stage && !(ablate & 16u)) bstage(it, B0, B2);
    __syncthreads();
    stamp(it, 5);
    if (it + 1 <= n_stage) {
      if (it + 1 < n_stage && !(ablate & 16u)) bstage(it + 1, B1, B0);
      __syncthreads();
      stamp(it + 1, 5);
    }
    if (it + 2 <= n_stage) {
      if (it + 2 < n_stage && !(ablate & 16u)) bstage(it + 2, B2, B1);
      __syncthreads();
      stamp(it + 2, 5);
    }
  }
  __syncthreads();  // the consumers' reduction barrier
}

}  // namespace fbk
