// fbk_matrix_mfma.hip.h — the many-row IntersectionCount matrix (GroupBy / TopN / TopK shape,
// executor.go:8880-8934, 2705-2774) for DENSE rows, on the matrix cores.
//
// out[shard][i][j] (+)= sum over the block's slots of |A[shard][i] ∩ F[shard] ∩ B[shard][j]|
//
// is C = A' · Bᵀ over {0,1} with K = 65536 bit positions per container: nA x nB pairs reuse
// nA + nB containers, so unlike every other kernel of this library the shape is arithmetic-bound
// on the vector ALU (round 1's vector-ALU kernel: 2.1 M container pairs x 64 v_and / v_bcnt per lane,
// 350 us for 128 shards x 32 x 32 rows, twice the time it takes to read the rows once).  The
// matrix cores do the pair work instead: v_mfma_i32_32x32x32_i8 multiplies a 32-row x 32-byte A
// operand with a 32-column x 32-byte B operand, and a bit becomes a byte with ONE v_and:
//
//   a = bswap(A & F), b = bitreverse(B)          (per 32-bit word, once)
//   for k in 0..7:  opA = a & (0x01010101 << k)           -> byte value 2^k where the bit is set
//                   opB = b & (0x01010101 << (7 - k))     -> byte value 2^(7-k) at the same source bit
//
// so every matching bit contributes 2^k · 2^(7-k) = 128, except that a byte with bit 7 set is
// -128 in i8: the k = 0 and k = 7 products are -128 and go to a second accumulator; the count is
// (P - N) >> 7, exact in i32 (at most 2^20 bits x 128 per shard).  Which source bit lands on which
// K index is irrelevant as long as A and B use the same mapping, which leaves the lane -> data
// assignment free: lane (r = lane & 31, g = lane >> 5) consumes 16-byte pieces of ITS row r.
//
// Per 32 bytes of every row: 8 MFMAs (256 matrix-core cycles per SIMD) against 76 VALU
// instructions, so a wavefront per SIMD keeps up with HBM: the kernel is organised around
// the memory system again.
//   * one 256-thread block (4 wavefronts, one per SIMD) per (shard, slot group, 32 A rows, 32 B
//     rows); the wavefronts split K — wave w owns the 128-byte pieces 4m + w of every container —
//     and never synchronise until the final reduction;
//   * a wave stages its own step (32 A + 32 B row pieces + the filter piece = 8.3 KB) with
//     global->LDS DMA, 8 lanes per 128-byte line (coalesced), 3 steps deep: 2 steps = 66 KB per CU
//     in flight while the third is consumed;
//   * inside a row's 128 bytes the eight 16-byte pieces are rotated by (row >> 1) on the way in
//     (a permutation inside one cache line, free), so that the 16 lanes of a ds_read_b128 pass hit
//     16 different bank groups.
#pragma once
#include "fbk_kernels.hip.h"

namespace fbk {

// product configuration (scripts/tune_matrix.hip: 4 waves x 2 stages = 66 KB of LDS, two blocks per
// CU; the non-temporal policy on the DMA is worth 10-15 %, as on every other streaming kernel here)
constexpr int kMmWaves = 4;  // wavefronts per block
constexpr int kMmDepth = 2;  // DMA ring depth (steps)
constexpr int kMmAux = 2;    // cache policy of the global->LDS DMA: nt
constexpr int kMmPiece = 128;  // bytes of every row per step

typedef int mm_v4i __attribute__((ext_vector_type(4)));
typedef int mm_v16i __attribute__((ext_vector_type(16)));
typedef uint32_t mm_u4 __attribute__((ext_vector_type(4)));

// TM x TN: 32-row tiles of A and of B per block.  1 x 1 is the HBM-bound shape (a 32 x 32 matrix
// reads every row once); for larger matrices a block computes TM x TN tiles from ONE expansion of
// its TM A operands and TN B operands — 2 x 2 halves the VALU, LDS and DMA work per MFMA, which is
// what bounds a matrix of many tiles (128 x 128 rows x 64 shards: 1256 us with 1 x 1 tiles, where
// the matrix cores alone would need 437 us).
// FP4 = true: the same product on the block-scaled FP4 matrix instruction
// (v_mfma_scale_f32_32x32x64_f8f6f4, formats E2M1 x E2M1, scales 2^0): a bit becomes a NIBBLE —
// (x >> k) & 0x11111111 for k = 0..3 puts bit (4 j + k) of the word into nibble j as code 0001 = 0.5
// — so one instruction covers K = 64 bit positions instead of 32 and every common bit adds
// 0.5 * 0.5 = 0.25 to an f32 accumulator (exact: at most 2^20 * 0.25 per shard, all partial sums
// are multiples of 0.25 below 2^24).  Per 16 bytes of every row: 4 matrix instructions instead of
// 8 and 60 vector-ALU operations instead of 76 — for matrices of several tiles, which the i8
// version leaves matrix-core bound (58-63 % of the dense i8 rate in round 1).
typedef int mm_v8i __attribute__((ext_vector_type(8)));
typedef float mm_v16f __attribute__((ext_vector_type(16)));

// The tiers of a ticketed launch (see TICKETS in the kernel): `resident` = the blocks the device holds at a time (2 per CU).  A tier's
// work has to outlast the raggedness the tier before it leaves behind — one unit of that tier per resident block — so tier k + 1
// gets the shards that `resident` units of tier k amount to; what is left over is tier 0, in units of spb0 slots.  Returns false
// (launch by block id) when the launch is too short for that: fewer than four rounds of tier-0 units.
struct MmTickets {
  uint32_t tier0_shards, tier1_shards, tier_spb, grid;
};
inline bool mm_ticket_plan(uint32_t n_shards, uint32_t spb0, uint32_t resident, MmTickets& t) {
  if (spb0 < 2 || uint64_t(n_shards) * (kSlots / spb0) < 4ull * resident) return false;
  const uint32_t spb1 = spb0 >= 8 ? spb0 / 4 : spb0 / 2, spb2 = 1;
  uint32_t s1 = (resident * spb0 + kSlots - 1) / kSlots;                 // shards of tier 1
  uint32_t s2 = spb1 > 1 ? (resident * spb1 + kSlots - 1) / kSlots : 0;  // shards of tier 2 (none when tier 1 is already single slots)
  if (s1 + s2 > n_shards / 2) return false;
  t.tier0_shards = n_shards - s1 - s2;
  t.tier1_shards = s1;
  t.tier_spb = spb1 | (spb2 << 8);
  const uint64_t units = uint64_t(t.tier0_shards) * (kSlots / spb0) + uint64_t(s1) * (kSlots / spb1) + uint64_t(s2) * (kSlots / spb2);
  t.grid = uint32_t((units + units / 8 + 64 + 7) & ~7ull);  // spare blocks: an XCD may take an eighth more than its share
  return true;
}

// A block's unit by ticket (see TICKETS in k_count_matrix_mfma): the ticket as the block id of the launch-by-id form of the unit's
// tier — its slots per block and slot groups per shard come back in spb / sgroups.  Called by every thread of the block.
__device__ __forceinline__ uint32_t mm_ticket_unit(uint32_t* __restrict__ ticket, uint32_t tier0_shards, uint32_t tier1_shards, uint32_t tier_spb, uint32_t& spb,
                                                   uint32_t& sgroups) {
  __shared__ uint32_t s_ticket;
  if (threadIdx.x == 0) s_ticket = atomicInc(ticket, gridDim.x - 1u);
  __syncthreads();
  uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket);
  const uint32_t units0 = tier0_shards * sgroups;
  if (b >= units0) {
    b -= units0;
    spb = tier_spb & 0xFFu;
    sgroups = kSlots / spb;
    uint32_t first = tier0_shards;  // the tier's first shard
    const uint32_t units1 = tier1_shards * sgroups;
    if (b >= units1) {
      b -= units1;
      spb = (tier_spb >> 8) & 0xFFu;
      sgroups = kSlots / spb;
      first += tier1_shards;
    }
    b += first * sgroups;
  }
  return b;
}

template <bool HAS_F, int WAVES = kMmWaves, int DEPTH = kMmDepth, int AUX = kMmAux, int TM = 1, int TN = 1, bool FP4 = false>
__global__ void __launch_bounds__(WAVES * 64) k_count_matrix_mfma(
    const uint8_t* __restrict__ arenaA, const uint32_t* __restrict__ rowsA, uint32_t nA, const uint8_t* __restrict__ arenaB,
    const uint32_t* __restrict__ rowsB, uint32_t nBtot, const uint8_t* __restrict__ arenaF,
    const uint32_t* __restrict__ rowsF, uint32_t n_shards, uint32_t spb, u64* __restrict__ out_shard,
    uint32_t* __restrict__ ticket = nullptr, uint32_t tier0_shards = 0, uint32_t tier1_shards = 0, uint32_t tier_spb = 0) {
  // stage of one wave and one step: A rows, B rows (128 bytes each), the filter piece
  constexpr uint32_t kABytes = TM * 32 * kMmPiece, kBBytes = TN * 32 * kMmPiece;
  constexpr uint32_t kStageBytes = kABytes + kBBytes + 256;
  // The ring is read with ds_read_b128 written as asm: the compiler's waitcnt pass otherwise puts
  // s_waitcnt vmcnt(0) in front of every LDS read that might alias a pending global->LDS DMA
  // (it cannot count DMA steps across the loop back-edge), which would serialise the prefetch
  // with the arithmetic.  vmcnt / lgkmcnt are managed by hand below.
  __shared__ uint4 ring[DEPTH][WAVES][kStageBytes / 16];
  static_assert(kStageBytes >= TM * TN * 16 * 64 * 4, "the final reduction parks the partial counts in stage 0");
#ifdef FBK_MM_STAMPS
  const unsigned long long stamp_t0 = wall_clock64();
#endif
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t agroups = (nA + 32 * TM - 1) / (32 * TM);
  const uint32_t btiles = (nBtot + 32 * TN - 1) / (32 * TN);
  uint32_t sgroups = kSlots / spb;
  uint32_t b = blockIdx.x;
  // The tiles of one (shard, slot group) read the same rows.  Workgroups go to the 8 XCDs round
  // robin and every XCD has its own L2, so consecutive block ids would put those tiles on 8
  // different L2s; with ids that differ by 8 they share one and the rows come from HBM once.
  if (agroups * btiles > 1 && (gridDim.x & 7u) == 0) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
  // TICKETS (round 6, single-tile matrices = the HBM-bound shape).  Block ids go to the XCDs round robin, an eighth of the grid
  // each whatever their pace — and their paces differ: scripts/matrix_xcd_hist.hip (profiles/r06_matrix_xcd_hist_*.txt) shows the
  // even XCDs through their share 5-8 % before the odd ones, every launch, and the last blocks of a launch running on a half-empty
  // device (4-5 % of a 1024-shard launch is that drain).  With a ticket counter a block's unit is the NEXT one, not its id: the
  // grid is launched with spare blocks (they find no unit and end at once), so an XCD that is ahead simply takes more units; and
  // the units come in three TIERS of shards — the first tier0_shards in units of `spb` slots, the next tier1_shards in units of
  // tier_spb[7:0] slots, the rest in units of tier_spb[15:8] — long units first, so that the end of the launch is ragged by a
  // short unit's time, not by a long one's.  atomicInc wraps at the grid size: every block takes exactly one ticket and the
  // counter is back at zero when the last one has (stream order does the rest).
  if (ticket) b = mm_ticket_unit(ticket, tier0_shards, tier1_shards, tier_spb, spb, sgroups);
  const uint32_t bt = b % btiles;
  b /= btiles;
  const uint32_t ag = b % agroups;
  b /= agroups;
  const uint32_t sg = b % sgroups;
  const uint32_t shard = b / sgroups;
  if (shard >= n_shards) return;
  const uint32_t i0 = ag * 32 * TM, j0 = bt * 32 * TN;
  const uint64_t rowBytes = (uint64_t)kSlots * 8192;

  // DMA side: instruction n covers rows 8n..8n+7, lane -> (row 8n + lane/8, LDS position lane%8),
  // which holds piece (position - row/2) mod 8 of the row's 128 bytes.  Rows past the end of the
  // matrix re-read its last row (their products are never written out).
  const uint8_t* pa[4 * TM];
  const uint8_t* pb[4 * TN];
#pragma unroll
  for (int n = 0; n < 4 * TM; ++n) {
    const uint32_t rr = 8 * n + (lane >> 3);
    const uint32_t piece = ((lane & 7) - (rr >> 1)) & 7;
    pa[n] = arenaA + (uint64_t)rowsA[(uint64_t)shard * nA + min(i0 + rr, nA - 1)] * rowBytes + piece * 16;
  }
#pragma unroll
  for (int n = 0; n < 4 * TN; ++n) {
    const uint32_t rr = 8 * n + (lane >> 3);
    const uint32_t piece = ((lane & 7) - (rr >> 1)) & 7;
    pb[n] = arenaB + (uint64_t)rowsB[(uint64_t)shard * nBtot + min(j0 + rr, nBtot - 1)] * rowBytes + piece * 16;
  }
  const uint8_t* pf = nullptr;
  if (HAS_F) pf = arenaF + (uint64_t)rowsF[shard] * rowBytes + (lane & 7) * 16;

  // consumer side: lane (r, g) reads piece 2t + g of row r (of every tile) in octet t
  const uint32_t r = lane & 31, g = lane >> 5;
  const uint32_t rot = (g + (r >> 1)) & 7;  // (32 rows per tile: the rotation repeats per tile)
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)&ring[0][wv][0];  // LDS byte address of this wave's stage 0
  uint32_t offAB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) offAB[t] = lds0 + r * 128 + ((rot + 2 * t) & 7) * 16;
  const uint32_t offF = lds0 + g * 16;

  // per tile: P collects the +128 products (k = 1..6), N the -128 ones (k = 0, 7); a single tile
  // alternates between two P accumulators so that consecutive MFMAs never chain on one
  constexpr int NP = (TM * TN == 1) ? 2 : 1;
  mm_v16i accP[TM][TN][NP], accN[TM][TN];
  mm_v16f accF[TM][TN][3];  // FP4 only (the compiler drops whichever set is unused)
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) {
#pragma unroll
      for (int h = 0; h < NP; ++h) accP[m][n][h] = mm_v16i{};
#pragma unroll
      for (int h = 0; h < 3; ++h) accF[m][n][h] = mm_v16f{};
      accN[m][n] = mm_v16i{};
    }

  constexpr uint32_t kStepsPerSlot = 8192 / (kMmPiece * WAVES);
  const uint32_t steps = spb * kStepsPerSlot;
  auto stage = [&](uint32_t st) {
    const uint32_t off = (sg * spb + st / kStepsPerSlot) * 8192u + ((st % kStepsPerSlot) * WAVES + wv) * kMmPiece;
    uint8_t* l = reinterpret_cast<uint8_t*>(&ring[st % DEPTH][wv][0]);
#pragma unroll
    for (int n = 0; n < 4 * TM; ++n) __builtin_amdgcn_global_load_lds((gptr_t)(pa[n] + off), (lptr_t)(l + n * 1024), 16, 0, AUX);
#pragma unroll
    for (int n = 0; n < 4 * TN; ++n)
      __builtin_amdgcn_global_load_lds((gptr_t)(pb[n] + off), (lptr_t)(l + kABytes + n * 1024), 16, 0, AUX);
    if (HAS_F) {
      if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(pf + off), (lptr_t)(l + kABytes + kBBytes), 16, 0, AUX);
    }
  };
  constexpr int kOps = 4 * TM + 4 * TN + (HAS_F ? 1 : 0);  // vmem instructions per staged step
  constexpr int kReads = TM + TN + (HAS_F ? 1 : 0);         // LDS reads per octet
  struct Oct {
    mm_u4 A[TM], B[TN], F;
  };
  auto issue = [&](Oct& o, uint32_t sbase, int t) {  // LDS reads of octet t of the stage at byte offset sbase
    const uint32_t ab = offAB[t] + sbase;
#pragma unroll
    for (int m = 0; m < TM; ++m) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o.A[m]) : "v"(ab), "n"(m * 4096));
#pragma unroll
    for (int n = 0; n < TN; ++n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o.B[n]) : "v"(ab), "n"(kABytes + n * 4096));
    if (HAS_F) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o.F) : "v"(offF + sbase), "n"(kABytes + kBBytes + 32 * t));
  };
  auto landed = [&](Oct& o, bool more_behind) {  // o's reads are complete (LDS returns in order)
    if (more_behind) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(kReads) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the values are defined from here on: nothing that uses them may be scheduled above the wait
#pragma unroll
    for (int m = 0; m < TM; ++m) asm volatile("" : "+v"(o.A[m]));
#pragma unroll
    for (int n = 0; n < TN; ++n) asm volatile("" : "+v"(o.B[n]));
    if (HAS_F) asm volatile("" : "+v"(o.F));
  };
  constexpr uint32_t M = 0x01010101u;
  auto octet = [&](const Oct& o) {
    if (FP4) {
      // nibble codes of E2M1: 0001 = 0.5, 0010 = 1.0, 0100 = 2.0 (1000 = -0.0: useless).  Masking the
      // word with 0x11111111 << k (k = 0, 1, 2) leaves bit 4j + k in nibble j with those values — ONE
      // v_and per operand dword, no shift; only k = 3 needs (x >> 3) & 0x11111111.  The products are
      // 0.25 (k = 0 and 3), 1 (k = 1) and 4 (k = 2) per common bit, kept in three accumulators:
      // count = 4 acc0 + acc1 + acc2 / 4, every partial sum exact in f32.
      constexpr uint32_t M4 = 0x11111111u;
      uint32_t a[TM][4];
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int d = 0; d < 4; ++d) a[m][d] = HAS_F ? (o.A[m][d] & o.F[d]) : o.A[m][d];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mm_v8i oa[TM], ob[TN];  // the instruction reads the first four registers of an FP4 operand: the rest stays undefined
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int d = 0; d < 4; ++d) oa[m][d] = (int)(k < 3 ? (a[m][d] & (M4 << k)) : ((a[m][d] >> 3) & M4));
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int d = 0; d < 4; ++d) ob[n][d] = (int)(k < 3 ? (o.B[n][d] & (M4 << k)) : ((o.B[n][d] >> 3) & M4));
        constexpr int kAcc[4] = {0, 1, 2, 0};
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            accF[m][n][kAcc[k]] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa[m], ob[n], accF[m][n][kAcc[k]], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
      return;
    }
    uint32_t a[TM][4], bb[TN][4];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int d = 0; d < 4; ++d) a[m][d] = __builtin_bswap32(HAS_F ? (o.A[m][d] & o.F[d]) : o.A[m][d]);
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int d = 0; d < 4; ++d) bb[n][d] = __builtin_bitreverse32(o.B[n][d]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mm_v4i oa[TM], ob[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int d = 0; d < 4; ++d) oa[m][d] = (int)(a[m][d] & (M << k));
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int d = 0; d < 4; ++d) ob[n][d] = (int)(bb[n][d] & (M << (7 - k)));
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          if (k == 0 || k == 7) accN[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa[m], ob[n], accN[m][n], 0, 0, 0);
          else accP[m][n][k % NP] = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa[m], ob[n], accP[m][n][k % NP], 0, 0, 0);
        }
    }
  };

  // the step about to be read has landed once at most `younger` later steps are still in flight
  auto dma_landed = [&](uint32_t younger) {
    static_assert(DEPTH >= 2 && DEPTH <= 5, "ring depth");
    static_assert((DEPTH - 2) * kOps <= 63, "vmcnt is a 6-bit counter");
    if (DEPTH > 4 && younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH > 4 ? 3 * kOps : 0) : "memory");
    else if (DEPTH > 3 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH > 3 ? 2 * kOps : 0) : "memory");
    else if (DEPTH > 2 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH > 2 ? kOps : 0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  Oct X, Y;
  for (uint32_t st = 0; st + 1 < (uint32_t)DEPTH && st < steps; ++st) stage(st);
  dma_landed(min((uint32_t)DEPTH - 2, steps - 1));
  issue(X, 0, 0);
  for (uint32_t st = 0; st < steps; ++st) {
    if (st + DEPTH - 1 < steps) stage(st + DEPTH - 1);
    const uint32_t sb = (st % DEPTH) * (WAVES * kStageBytes);
    issue(Y, sb, 1);
    landed(X, true);
    octet(X);
    issue(X, sb, 2);
    landed(Y, true);
    octet(Y);
    issue(Y, sb, 3);
    landed(X, true);
    octet(X);
    const bool more = st + 1 < steps;
    if (more) {
      dma_landed(min((uint32_t)DEPTH - 2, steps - 2 - st));
      issue(X, ((st + 1) % DEPTH) * (WAVES * kStageBytes), 0);
    }
    landed(Y, more);
    octet(Y);
  }

  // cross-wave reduction through LDS: every wave parks its partial counts in its own (fully
  // consumed) stage 0, then the waves share out the TM*TN*16 accumulator registers
  uint32_t* red = reinterpret_cast<uint32_t*>(&ring[0][wv][0]);  // [TM*TN*16][64]
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        red[((m * TN + n) * 16 + q) * 64 + lane] =
            FP4 ? (uint32_t)(accF[m][n][0][q] * 4.0f + accF[m][n][1][q] + accF[m][n][2][q] * 0.25f + 0.5f)
                : (uint32_t)(accP[m][n][0][q] + (NP > 1 ? accP[m][n][NP - 1][q] : 0) - accN[m][n][q]) >> 7;
  __syncthreads();
  constexpr int kRegs = TM * TN * 16;
  static_assert(kRegs % WAVES == 0, "accumulator registers are dealt evenly to the waves");
#pragma unroll
  for (int qq = 0; qq < kRegs / WAVES; ++qq) {
    const int x = wv * (kRegs / WAVES) + qq;  // (tile, register): wave-uniform
    const int tile = x >> 4, q = x & 15;
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) tot += reinterpret_cast<const uint32_t*>(&ring[0][w][0])[x * 64 + lane];
    const uint32_t i = (tile / TN) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), j = (tile % TN) * 32 + (lane & 31);
    if (i0 + i < nA && j0 + j < nBtot && tot) {
      u64* dst = &out_shard[((uint64_t)shard * nA + i0 + i) * nBtot + j0 + j];
      if (spb == kSlots) *dst = tot;
      else atomicAdd(dst, (u64)tot);
    }
  }
#ifdef FBK_MM_STAMPS
  __syncthreads();
  if (threadIdx.x == 0 && g_mm_stamps) {
    unsigned long long* st = g_mm_stamps + 4ull * blockIdx.x;
    st[0] = stamp_t0;
    st[1] = wall_clock64();
    st[2] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11));   // HW_REG_XCC_ID[3:0]
    st[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID
  }
#endif
}

}  // namespace fbk
