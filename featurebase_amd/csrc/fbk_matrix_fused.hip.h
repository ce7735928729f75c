// fbk_matrix_fused.hip.h — the many-row IntersectionCount matrix (GroupBy / TopN shape,
// executor.go:8880-8934, 2705-2774) for rows in ANY encoding, on the matrix cores, without
// ever writing a decoded row to HBM.
//
//   out[shard][i][j] += sum over the block's slots of |A[shard][i] ∩ F[shard] ∩ B[shard][j]|
//
// Round 1 densified encoded rows into temporary bitmap rows (k_densify_rows) and ran the dense
// matrix-core kernel on those: 291 MB of encoded rows became 1.07 GB written and 1.07 GB read back
// (8.4x the algorithmic traffic, 453 us end to end for 128 shards of config 3's rows).  Here the
// decode happens inside the kernel, one eighth of the value range at a time:
//
//   * one block = (shard, slot group, 32 A rows, 32 B rows) = 12 wavefronts: 4 CONSUMERS (one per
//     SIMD, the matrix-core pipeline of fbk_matrix_mfma.hip.h: bit -> i8 expansion by one v_and per
//     operand dword, v_mfma_i32_32x32x32_i8) and 8 PRODUCERS that decode;
//   * a stage = the 8192 bit positions [q * 8192, (q + 1) * 8192) of all 65 rows (32 A, 32 B, the
//     filter) as 1 KiB of bitmap per row in LDS; two stages (130 KB of the CU's 160 KB) alternate:
//     the producers fill one while the consumers multiply the other, one barrier per stage;
//   * producers own rows (row r belongs to producer r mod 8) and bring a row's share of the stage
//     in from its ENCODED form:
//       bitmap  one global->LDS DMA of the container's q-th KiB (nothing passes through registers)
//       array   values are sorted, so a stage's values are a contiguous piece of the array: a per-row
//               cursor walks it.  Four rows are decoded per pass, 16 lanes x 8 values (one 16-byte
//               load per lane) each, one ds_or_b32 per value
//       run     the runs that intersect the stage are toggled in at their (clamped) start and one
//               past their (clamped) end, then a parity prefix over the row's 128 words fills them
//               — constant work per row whatever the run lengths (runToBitmap, roaring.go:3792)
//     so HBM sees the encoded payload once per 32 x 32 tile (plus re-reads of an array's cache lines
//     by later stages, which hit in L2) and nothing else;
//   * inside a row the 16-byte pieces of every 256-byte group are permuted by XOR with (row mod 16),
//     so that the 16 lanes of a ds_read_b128 pass — 16 different rows at the same logical piece —
//     hit 16 different bank groups (and a value's dword address is ONE xor away from its logical
//     one); the DMA realises the permutation through its per-lane SOURCE addresses (the LDS side of
//     a DMA is linear).  K order is free for a count, as long as all
//     rows use the same logical order, which they do.
#pragma once
#include "fbk_matrix_mfma.hip.h"

namespace fbk {

constexpr int kFxSB = 1024;                     // bytes of every row per stage
constexpr int kFxStagesPerSlot = 8192 / kFxSB;  // 8
constexpr int kFxNR = 65;                       // 32 A rows + 32 B rows + the filter row
constexpr int kFxConsumers = 4;
constexpr int kFxProducers = 12;
constexpr int kFxWaves = kFxConsumers + kFxProducers;
constexpr int kFxRowsPerProducer = (kFxNR + kFxProducers - 1) / kFxProducers;  // 9
constexpr int kFxArrayPasses = (kFxRowsPerProducer + 3) / 4;                  // 3

constexpr int kFxNarrowMax = 640;  // arrays of at most this many values are decoded four rows at a time

// sum over the 16 lanes of a DPP row, left in every lane of the row: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror — no LDS round trip (a ds_bpermute shuffle costs one)
__device__ __forceinline__ uint32_t fx_row16_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
  return v;
}

// Loads through pointers that went through a readlane / shuffle come out of the compiler as FLAT
// loads (the address space is lost in the integer round trip), and a flat load counts on vmcnt AND
// lgkmcnt: waiting for its data then waits for every LDS atomic the wave has in flight (~2000 cycles
// with eight producers firing bursts of them — measured: 2100 cycles per 100-instruction iteration).
// These say "global" explicitly.
__device__ __forceinline__ uint4 fx_ld_global16(const uint8_t* p) {
  const mm_u4 v = *reinterpret_cast<const __attribute__((address_space(1))) mm_u4*>((uintptr_t)p);
  return uint4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ uint32_t fx_ld_global4(const uint8_t* p) {
  return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>((uintptr_t)p);
}

// physical 16-byte piece of logical piece p in a row whose rotation is rot (= row & 15)
__device__ __forceinline__ uint32_t fx_phys_piece(uint32_t p, uint32_t rot) { return p ^ rot; }

template <bool HAS_F, bool PROF = false>
__global__ void __launch_bounds__(kFxWaves * 64) k_count_matrix_fused(
    const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA, const uint32_t* __restrict__ rowsA, uint32_t nA,
    const Slot* __restrict__ slotsB, const uint8_t* __restrict__ arenaB, const uint32_t* __restrict__ rowsB, uint32_t nBtot,
    const Slot* __restrict__ slotsF, const uint8_t* __restrict__ arenaF, const uint32_t* __restrict__ rowsF, uint32_t n_shards,
    uint32_t spb, u64* __restrict__ out_shard, uint32_t ablate, u64* __restrict__ prof = nullptr) {
  // PROF (option matrix_fused_ablate & 32): cycles per phase, summed over all waves into prof[]:
  // 0 zeroing, 1 narrow arrays, 2 wide arrays, 3 runs, 4 DMA issue + slot switch, 5 producer barrier wait,
  // 6 consumer arithmetic, 7 consumer barrier wait, 8 narrow iterations, 9 wide iterations, 10 run rows x stages,
  // 11 producer waves, 12 consumer waves
  bool lane0_flag = false;
  u64 pt[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto clk = [&]() -> u64 { return PROF ? (u64)__builtin_readcyclecounter() : 0ull; };
  u64 t_last = clk();
  auto lap = [&](int k) {
    if (PROF) {
      const u64 t = clk();
      pt[k] += t - t_last;
      t_last = t;
    }
  };
  auto prof_flush = [&]() {
    if (PROF && lane0_flag) {
      for (int k = 0; k < 13; ++k)
        if (pt[k]) atomicAdd(&prof[k], pt[k]);
    }
  };
  // `ablate` (option matrix_fused_ablate, timing experiments only — results are wrong when set):
  // 1 no consumer arithmetic, 2 no array decode, 4 no run decode, 8 no bitmap DMA, 16 no row zeroing
  __shared__ uint4 ring[2][kFxNR * kFxSB / 16];  // 133 120 bytes
  __shared__ uint32_t s_list[kFxProducers][16];  // per producer: lanes (= rows) holding narrow arrays, in order (read at slot set-up only)
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lane0_flag = lane == 0;
  const uint32_t agroups = (nA + 31) / 32, btiles = (nBtot + 31) / 32, sgroups = kSlots / spb;
  uint32_t b = blockIdx.x;
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
  const uint32_t n_stage = n_act * kFxStagesPerSlot;
  uint8_t* const ring8 = reinterpret_cast<uint8_t*>(&ring[0][0]);
  constexpr uint32_t kBufBytes = kFxNR * kFxSB;

  if (wv < kFxConsumers) {
    // ============================== consumers ==============================
    const uint32_t r = lane & 31, g = lane >> 5;
    const uint32_t rot = r & 15u;
    mm_v16i accP0{}, accP1{}, accN{};
    constexpr uint32_t M = 0x01010101u;
    for (uint32_t it = 0; it <= n_stage; ++it) {
      if (it >= 1 && !(ablate & 1u)) {
        const uint8_t* buf = ring8 + ((it - 1) & 1u) * kBufBytes;
        const uint8_t* rowA = buf + r * kFxSB;
        const uint8_t* rowB = buf + (32 + r) * kFxSB;
        const uint8_t* rowF = buf + 64 * kFxSB;
        auto ld = [&](int o, uint4& va, uint4& vb, uint4& vf) {
          const uint32_t p = 16u * (uint32_t)wv + 2u * o + g;  // logical piece of this lane's K block
          const uint32_t pp = fx_phys_piece(p, rot) * 16u;
          va = *reinterpret_cast<const uint4*>(rowA + pp);
          vb = *reinterpret_cast<const uint4*>(rowB + pp);
          if (HAS_F) vf = *reinterpret_cast<const uint4*>(rowF + p * 16u);
        };
        auto octet = [&](const uint4& va, const uint4& vb, const uint4& vf) {
          uint32_t a[4] = {va.x, va.y, va.z, va.w}, bb[4] = {vb.x, vb.y, vb.z, vb.w};
          const uint32_t f[4] = {vf.x, vf.y, vf.z, vf.w};
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            a[d] = __builtin_bswap32(HAS_F ? (a[d] & f[d]) : a[d]);
            bb[d] = __builtin_bitreverse32(bb[d]);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            mm_v4i oa, ob;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              oa[d] = (int)(a[d] & (M << k));
              ob[d] = (int)(bb[d] & (M << (7 - k)));
            }
            if (k == 0 || k == 7) accN = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob, accN, 0, 0, 0);
            else if (k & 1) accP1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob, accP1, 0, 0, 0);
            else accP0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob, accP0, 0, 0, 0);
          }
        };
        // the reads of octet o + 1 are in flight while octet o is multiplied
        uint4 xa, xb, xf = uint4{0, 0, 0, 0}, ya, yb, yf = uint4{0, 0, 0, 0};
        ld(0, xa, xb, xf);
#pragma unroll
        for (int o = 0; o < 8; o += 2) {
          ld(o + 1, ya, yb, yf);
          octet(xa, xb, xf);
          if (o + 2 < 8) ld(o + 2, xa, xb, xf);
          octet(ya, yb, yf);
        }
      }
      lap(6);
      __syncthreads();
      lap(7);
    }
    pt[12] = 1;
    prof_flush();
    // cross-wave reduction through LDS (the ring is free now): [wave][16 regs][64 lanes]
    uint32_t* red = reinterpret_cast<uint32_t*>(ring8);
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(wv * 16 + q) * 64 + lane] = (uint32_t)(accP0[q] + accP1[q] - accN[q]) >> 7;
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
  const uint32_t pw = (uint32_t)wv - kFxConsumers;  // 0..7; owns rows pw, pw + 8, ... (lane j stands for row pw + 8 j)
  const uint32_t gq = lane >> 4, gl = lane & 15;
  // the descriptor table row of this lane's matrix row (fixed for the whole kernel)
  const uint32_t my_row = pw + (uint32_t)kFxProducers * (uint32_t)lane;
  const Slot* my_slots = nullptr;
  const uint8_t* my_base = nullptr;
  if (lane < kFxRowsPerProducer && my_row < (uint32_t)kFxNR) {
    if (my_row < 32) {
      if (i0 + my_row < nA) my_slots = slotsA + (uint64_t)rowsA[(uint64_t)shard * nA + i0 + my_row] * kSlots;
      my_base = arenaA;
    } else if (my_row < 64) {
      if (j0 + my_row - 32 < nBtot) my_slots = slotsB + (uint64_t)rowsB[(uint64_t)shard * nBtot + j0 + my_row - 32] * kSlots;
      my_base = arenaB;
    } else if (HAS_F) {
      my_slots = slotsF + (uint64_t)rowsF[shard] * kSlots;
      my_base = arenaF;
    }
  }
  auto load_desc = [&](uint32_t slot) {
    Slot d;
    d.off = 0;
    d.len = 0;
    d.tn = 0;
    if (my_slots) d = my_slots[slot];
    return d;
  };
  // per-slot state ------------------------------------------------------------------------------
  const uint8_t* d_ptr = nullptr;  // descriptor lanes: payload pointer, length, type (0 nil / empty)
  uint32_t d_len = 0;
  uint32_t n_nar = 0;
  u64 bmask = 0, wmask = 0, rmask = 0;  // descriptor lanes holding bitmaps / wide arrays / runs (wave-uniform)
  uint32_t d_cur = 0;                   // descriptor lanes: cursor of a wide array / run row (readlane / writelane)
  // narrow arrays (<= kFxNarrowMax values): group gq of pass p decodes the (4 p + gq)-th of them,
  // 16 lanes x 8 values = a window of 128 values per stage; the window of the NEXT stage is loaded
  // as soon as the cursor is known, so its latency hides behind the rest of the stage and the barrier
  const uint8_t* a_ptr[kFxArrayPasses];
  uint32_t a_len[kFxArrayPasses], a_row[kFxArrayPasses], a_cur[kFxArrayPasses];
  uint4 a_win[kFxArrayPasses];
  // wide arrays and runs: one row at a time over all 64 lanes (windows of 512 values / 64 runs);
  // the first two rows of each kind have their next window prefetched.  Nothing on these paths READS
  // the LDS (row lists are walked as bit masks, cursors sit in descriptor lanes): an LDS read returns
  // behind every atomic the wave has in flight, and with 8 producers firing bursts of atomics that
  // queue is ~2000 cycles deep — one such read per row was what the first version spent its time on
  uint4 w_win0 = uint4{0, 0, 0, 0}, w_win1 = uint4{0, 0, 0, 0};
  uint32_t r_win0 = 0, r_win1 = 0;

  auto row_state = [&](int src, const uint8_t*& ptr, uint32_t& len, uint32_t& row) {  // src: wave-uniform lane index
    const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)d_ptr, src);
    const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uintptr_t)d_ptr >> 32), src);
    ptr = reinterpret_cast<const uint8_t*>(((uintptr_t)phi << 32) | plo);
    len = (uint32_t)__builtin_amdgcn_readlane((int)d_len, src);
    row = pw + (uint32_t)kFxProducers * (uint32_t)src;
  };
  // slots at and past the end of the array get a value that lies outside the stage starting at lo
  auto fix_tail = [&](uint4& w, uint32_t idx0, uint32_t len, uint32_t lo) {
    if (idx0 + 8u > len) {
      const uint32_t sent = (lo + (uint32_t)(kFxSB * 8)) & 0xFFFFu, sent2 = sent | (sent << 16);
      const uint32_t nv = idx0 < len ? len - idx0 : 0u;  // valid slots of this lane: 0 .. 7
      uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) ww[d] = nv >= 2u * d + 2u ? ww[d] : nv == 2u * d + 1u ? ((ww[d] & 0xFFFFu) | (sent << 16)) : sent2;
      w = uint4{ww[0], ww[1], ww[2], ww[3]};
    }
  };
  auto load_wide = [&](const uint8_t* ptr, uint32_t len, uint32_t cur, uint32_t lo) {  // lo: first value of the stage the window is for
    const uint32_t idx0 = (cur & ~7u) + 8u * (uint32_t)lane;
    uint4 w = uint4{0, 0, 0, 0};
    if (idx0 < len) w = fx_ld_global16(ptr + 2u * idx0);  // payloads are padded to 16 bytes
    fix_tail(w, idx0, len, lo);
    return w;
  };
  auto load_runs = [&](const uint8_t* ptr, uint32_t len, uint32_t cur) {
    const uint32_t idx = cur + (uint32_t)lane;
    return idx < len ? fx_ld_global4(ptr + 4u * idx) : 0u;
  };
  auto load_narrow = [&](int p, uint32_t lo) {
    const uint32_t idx0 = (a_cur[p] & ~7u) + 8u * gl;
    uint4 w = uint4{0, 0, 0, 0};
    if (idx0 < a_len[p]) w = fx_ld_global16(a_ptr[p] + 2u * idx0);
    fix_tail(w, idx0, a_len[p], lo);
    return w;
  };
  // 8 values of one lane -> bits of the stage [lo, lo + 8192) of a row; returns how many were in the stage.
  // The LDS executes only a few lane-atomics per clock (the scatter kernels of round 1 measured ~3):
  // lanes without a value in the stage are switched off, not given a dummy target.  Invalid slots
  // (past the end of the array) were replaced by an out-of-stage sentinel when the window was loaded.
  auto scatter8 = [&](const uint4& w, uint32_t lo, uint32_t rowkey) {  // rowkey = byte address of the row ^ (rot << 4)
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t v = (k & 1) ? (ww[k >> 1] >> 16) : (ww[k >> 1] & 0xFFFFu);
      const uint32_t bit = v - lo;  // values below lo wrap to something huge
      if (bit < (uint32_t)(kFxSB * 8)) {
        // dword (bit >> 5) of the row, pieces permuted: byte address = ((bit >> 5) << 2) ^ rowkey
        atomicOr(reinterpret_cast<uint32_t*>(ring8 + (((bit >> 3) & 0x3FCu) ^ rowkey)), 1u << (bit & 31u));
        ++c;
      }
    }
    return c;
  };
  // make the descriptors `d` (one per descriptor lane) the current slot and load its first windows
  auto setup_slot = [&](const Slot& d) {
    const uint32_t type = slot_n(d) ? slot_type(d) : 0u;
    d_len = d.len;
    d_ptr = my_base + d.off;
    const bool narrow = type == kTypeArray && d.len <= (uint32_t)kFxNarrowMax;
    const bool wide = type == kTypeArray && !narrow;
    const u64 nm = __ballot(narrow);
    wmask = __ballot(wide);
    rmask = __ballot(type == kTypeRun);
    bmask = __ballot(type == kTypeBitmap);
    n_nar = __popcll(nm);
    d_cur = 0;
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (narrow) s_list[pw][__popcll(nm & lt)] = lane;
    wave_lds_sync();
#pragma unroll
    for (int p = 0; p < kFxArrayPasses; ++p) {
      const uint32_t e = 4u * p + gq;
      const uint32_t src = e < n_nar ? s_list[pw][e] : 0u;
      const uint32_t plo = (uint32_t)__shfl((int)(uint32_t)(uintptr_t)d_ptr, (int)src, kWave);
      const uint32_t phi = (uint32_t)__shfl((int)(uint32_t)((uintptr_t)d_ptr >> 32), (int)src, kWave);
      a_ptr[p] = reinterpret_cast<const uint8_t*>(((uintptr_t)phi << 32) | plo);
      a_len[p] = e < n_nar ? (uint32_t)__shfl((int)d_len, (int)src, kWave) : 0u;
      a_row[p] = pw + (uint32_t)kFxProducers * src;
      a_cur[p] = 0;
      a_win[p] = uint4{0, 0, 0, 0};
      if (4u * p < n_nar) a_win[p] = load_narrow(p, 0);
    }
    const uint8_t* ptr;
    uint32_t len, row;
    {
      u64 m = wmask;
      if (m) {
        row_state(__builtin_ctzll(m), ptr, len, row);
        w_win0 = load_wide(ptr, len, 0, 0);
        m &= m - 1;
      }
      if (m) {
        row_state(__builtin_ctzll(m), ptr, len, row);
        w_win1 = load_wide(ptr, len, 0, 0);
      }
      m = rmask;
      if (m) {
        row_state(__builtin_ctzll(m), ptr, len, row);
        r_win0 = load_runs(ptr, len, 0);
        m &= m - 1;
      }
      if (m) {
        row_state(__builtin_ctzll(m), ptr, len, row);
        r_win1 = load_runs(ptr, len, 0);
      }
    }
  };

  Slot next_d;  // descriptors of the next active slot, fetched a few stages ahead
  next_d.off = 0;
  next_d.len = 0;
  next_d.tn = 0;
  if (n_stage) setup_slot(load_desc(act[0]));
  for (uint32_t it = 0; it <= n_stage; ++it) {
    if (it < n_stage) {
      const uint32_t si = it / kFxStagesPerSlot;
      const uint32_t q = it % kFxStagesPerSlot;
      const uint32_t lo = q * (kFxSB * 8u), hi = lo + (uint32_t)(kFxSB * 8);
      const uint32_t bufoff = (it & 1u) * kBufBytes;
      const bool last_q = q + 1 == (uint32_t)kFxStagesPerSlot;
      if (q == 1 && si + 1 < n_act) next_d = load_desc(act[si + 1]);
      // ---- 1. zero the rows that are not bitmaps (arrays, runs, nil): lane j's 16 bytes of each ----
      if (!(ablate & 16u)) {
        const u64 zm = ~bmask;
        for (uint32_t j = 0; j < (uint32_t)kFxRowsPerProducer; ++j) {
          const uint32_t row = pw + (uint32_t)kFxProducers * j;
          if (row < (uint32_t)kFxNR && ((zm >> j) & 1ull))
            *reinterpret_cast<uint4*>(ring8 + bufoff + row * kFxSB + lane * 16) = uint4{0, 0, 0, 0};
        }
      }
      wave_lds_sync();
      lap(0);
      // ---- 2. narrow arrays: four rows per pass ----
#pragma unroll
      for (int p = 0; p < kFxArrayPasses; ++p) {
        if (4u * p < n_nar && !(ablate & 2u)) {  // wave-uniform
          const uint32_t rowkey = (bufoff + a_row[p] * kFxSB) ^ ((a_row[p] & 15u) << 4);
          uint4 w = a_win[p];
          bool live = true;  // this 16-lane group still has values of the stage to read
          for (;;) {
            if (PROF) ++pt[8];
            uint32_t c = 0;
            if (live) c = scatter8(w, lo, rowkey);
            c = fx_row16_sum(c);
            const uint32_t wend = (a_cur[p] & ~7u) + 128u;
            a_cur[p] += c;
            live = live && a_cur[p] == wend && a_cur[p] < a_len[p];  // the window ended inside the stage: read on
            if (__ballot(live) == 0) break;
            w = uint4{0, 0, 0, 0};
            if (live) w = load_narrow(p, lo);
          }
          if (!last_q) a_win[p] = load_narrow(p, hi);  // the next stage's window
        }
      }
      lap(1);
      // ---- 3. wide arrays: one row at a time, all 64 lanes ----
      {
        u64 m = (ablate & 2u) ? 0ull : wmask;
        for (uint32_t e = 0; m; ++e) {
          const int src = __builtin_ctzll(m);
          m &= m - 1;
          const uint8_t* ptr;
          uint32_t len, row;
          row_state(src, ptr, len, row);
          const uint32_t rowkey = (bufoff + row * kFxSB) ^ ((row & 15u) << 4);
          uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)d_cur, src);
          uint4 w = e == 0 ? w_win0 : e == 1 ? w_win1 : load_wide(ptr, len, cur, lo);
          for (;;) {
            if (PROF) ++pt[9];
            const uint32_t c = fx_row16_sum(scatter8(w, lo, rowkey));
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)c, 0) + (uint32_t)__builtin_amdgcn_readlane((int)c, 16) +
                                 (uint32_t)__builtin_amdgcn_readlane((int)c, 32) + (uint32_t)__builtin_amdgcn_readlane((int)c, 48);
            const uint32_t wend = (cur & ~7u) + 512u;
            cur += tot;
            if (!(cur == wend && cur < len)) break;
            w = load_wide(ptr, len, cur, lo);
          }
          d_cur = lane == src ? cur : d_cur;
          if (!last_q) {
            if (e == 0) w_win0 = load_wide(ptr, len, cur, hi);
            if (e == 1) w_win1 = load_wide(ptr, len, cur, hi);
          }
        }
      }
      lap(2);
      // ---- 4. runs: one row at a time, all 64 lanes ----
      {
        const u64 rm = (ablate & 4u) ? 0ull : rmask;
        u64 m = rm;
        for (uint32_t e = 0; m; ++e) {  // (a) toggles of every run row of this wave
          const int src = __builtin_ctzll(m);
          m &= m - 1;
          const uint8_t* ptr;
          uint32_t len, row;
          row_state(src, ptr, len, row);
          const uint32_t rowkey = (bufoff + row * kFxSB) ^ ((row & 15u) << 4);
          uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)d_cur, src);
          if (PROF) ++pt[10];
          uint32_t iv = e == 0 ? r_win0 : e == 1 ? r_win1 : load_runs(ptr, len, cur);
          for (;;) {
            const bool have = cur + (uint32_t)lane < len;
            const uint32_t s = iv & 0xFFFFu, l = iv >> 16;
            const bool inr = have && s < hi;  // (l >= lo: the cursor never rests on a run that ended before lo)
            const uint32_t s2 = (s > lo ? s : lo) - lo;           // 0 .. 8191
            const uint32_t e2 = (l + 1u < hi ? l + 1u : hi) - lo;  // 1 .. 8192
            if (inr) {
              atomicXor(reinterpret_cast<uint32_t*>(ring8 + (((s2 >> 3) & 0x3FCu) ^ rowkey)), 1u << (s2 & 31u));
              if (e2 < (uint32_t)(kFxSB * 8)) atomicXor(reinterpret_cast<uint32_t*>(ring8 + (((e2 >> 3) & 0x3FCu) ^ rowkey)), 1u << (e2 & 31u));
            }
            const uint32_t cnt = (uint32_t)__popcll(__ballot(inr && l < hi));  // runs that end inside this stage
            cur += cnt;
            if (!(cnt == 64u && cur < len)) break;
            iv = load_runs(ptr, len, cur);
          }
          d_cur = lane == src ? cur : d_cur;
          if (!last_q) {
            if (e == 0) r_win0 = load_runs(ptr, len, cur);
            if (e == 1) r_win1 = load_runs(ptr, len, cur);
          }
        }
        wave_lds_sync();
        m = rm;
        while (m) {  // (b) parity prefix over every run row: lane j owns logical piece j (words 2j, 2j + 1)
          const int src = __builtin_ctzll(m);
          m &= m - 1;
          const uint32_t row = pw + (uint32_t)kFxProducers * (uint32_t)src, rot = row & 15u;
          uint4* pc = reinterpret_cast<uint4*>(ring8 + bufoff + row * kFxSB + fx_phys_piece((uint32_t)lane, rot) * 16u);
          const uint4 tv = *pc;
          const u64 t0 = ((u64)tv.y << 32) | tv.x, t1 = ((u64)tv.w << 32) | tv.z;
          const uint32_t p0 = __popcll(t0) & 1u, p1 = __popcll(t1) & 1u;
          const u64 mm = __ballot((p0 ^ p1) != 0);
          const u64 lane_lt = lane ? (~0ull >> (64 - lane)) : 0ull;
          const uint32_t in = __popcll(mm & lane_lt) & 1u;
          const u64 f0 = prefix_xor64(t0) ^ (in ? ~0ull : 0ull);
          const u64 f1 = prefix_xor64(t1) ^ ((in ^ p0) ? ~0ull : 0ull);
          *pc = uint4{(uint32_t)f0, (uint32_t)(f0 >> 32), (uint32_t)f1, (uint32_t)(f1 >> 32)};
        }
      }
      lap(3);
      // ---- 5. bitmaps: the q-th KiB of the container straight into the row (DMA, rotated source) ----
      if (!(ablate & 8u)) {
        u64 bm = bmask;
        while (bm) {
          const int src = __builtin_ctzll(bm);
          bm &= bm - 1;
          const uint8_t* pay;
          uint32_t len, row;
          row_state(src, pay, len, row);
          const uint32_t rot = row & 15u;
          const uint32_t logical = (uint32_t)lane ^ rot;  // the piece that belongs at LDS position `lane` (the permutation is an involution)
          __builtin_amdgcn_global_load_lds((gptr_t)(pay + q * kFxSB + logical * 16u), (lptr_t)(ring8 + bufoff + row * kFxSB), 16, 0, 2);
        }
      }
      // ---- 6. the slot is done: switch to the next one (its descriptors were fetched at q == 1) ----
      if (last_q && si + 1 < n_act) setup_slot(next_d);
      lap(4);
    }
    __syncthreads();  // (waits for this wave's DMA and LDS traffic, then the block barrier)
    lap(5);
  }
  __syncthreads();  // the consumers' reduction barrier
  pt[11] = 1;
  prof_flush();
}

}  // namespace fbk
