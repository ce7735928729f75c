// fbk_pair_kernels.hip.h — type-pair specialised row-pair kernels (round 3): k_icount2, k_setop2<OP>.
//
// The round-1/2 pair kernels (fbk_kernels.hip.h: k_icount, k_setop) turn BOTH operands of every
// (pair, slot) into a register fragment through the wave's 8 KiB LDS scratch: clear 8 KiB, scatter,
// read 8 KiB back — twice per pair, whatever the operands hold.  On gfx950 a `ds_write_b128` costs
// ~13 LDS cycles per wave instruction (MI355X_MICROARCH.md, LDS table), so the two clears alone are
// ~210 cycles, and the array scatter (4 consecutive values per lane: neighbouring lanes 4 values
// = ~8 dwords apart for a 1000-value array, 32 banks) ran 8-way bank-conflicted.  With 128 waves per
// CU on config 3's row pairs that is ~35 us of LDS time out of the 70 us the kernel took.
//
// Here the work per pair follows the operand types, as the reference's dispatch does
// (intersectionCount roaring.go:4477-4512 and its six kernels :4514-4614; intersect :4753-4978 ...):
//
//   bitmap x bitmap   both streamed to registers, no LDS                 (intersectionCountBitmapBitmap :4611)
//   array  x array    the shorter array is scattered into ONE cleared 8 KiB table, the longer one
//                     PROBES it (one ds_read_b32 per value, hits counted per lane): 1 clear, no
//                     read-back                                          (intersectionCountArrayArray :4514)
//   array  x bitmap   <= 128 values: probe the bitmap's dwords in global memory; else the bitmap is
//                     copied global -> registers -> LDS (no clear) and the array probes it
//                                                                        (intersectionCountArrayBitmap :4596)
//   run    x any      the pair loader below: ONE clear for both operands (intersectionCountArrayRun :4537,
//                     BitmapRun :4563, RunRun :4573 — all three as toggle decode + AND + popcount)
//
// Pair loader (frag_load_pair), also the front end of the materialising k_setop2: bitmaps stream
// straight to registers; the first sparse operand XORs its raw bits (array: one bit per value; run: a
// toggle at start and at last + 1) into the cleared table and is read back; the second sparse operand
// XORs ITS raw bits ON TOP of the first's and is read back as (first ^ second) ^ first — no second
// clear.  Run toggles become filled runs by the in-register parity prefix (frag_load_run's).
//
// Every sparse payload is read lane-consecutively (element i by lane i mod 64): neighbouring lanes
// then hit neighbouring dwords of the table (2-way conflicts at 1000 values, none from 2000 on)
// and the first batch of BOTH operands is in flight before any LDS work starts.
#pragma once
#include "fbk_kernels.hip.h"

namespace fbk {

constexpr int kPairBatch = 8;  // dwords per lane and operand in flight: 8 x 64 dwords = 1024 array values / 512 runs per batch

// A sparse payload is read in DWORD units, lane-consecutively (unit i by lane i mod 64): an array dword holds two
// values, a run dword one {start, last} interval.  The instruction count per value is what these kernels are made
// of (rocprofv3 --pmc on config 3's row pairs, profiles/r03_pmc_pair_kernels.txt: a SIMD issues about one
// instruction per 4 cycles whatever its kind, and the first version of this file spent ~20 on every array value —
// index, bounds test, address, uint16 load, bit, exec-mask juggling), so a batch whose lanes are ALL valid runs
// without any per-value predicate and only the ragged last batch tests indexes.
__device__ __forceinline__ uint32_t sparse_units(uint32_t type, uint32_t len) { return type == kTypeArray ? (len + 1u) >> 1 : len; }

// batch starting at unit `base`: v[k] = unit base + 64 k + lane (junk — zero — past the end)
__device__ __forceinline__ void sparse_load(const uint8_t* __restrict__ p, uint32_t n_units, uint32_t base, int lane, uint32_t (&v)[kPairBatch]) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p) + base + (uint32_t)lane;
  if (base + kPairBatch * kWave <= n_units) {
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = q[k * kWave];
  } else {
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = (base + (uint32_t)k * kWave + (uint32_t)lane < n_units) ? q[k * kWave] : 0u;
  }
}

// The wave's table as a raw LDS byte offset (8 KiB aligned): the dword of bit b is base | (b[15:5] << 2) — a bit-field
// extract and one v_lshl_or_b32 instead of shift + mask + add;
// shift counts and bit-field offsets use the hardware's own masking of bits [4:0].
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_table_base(u64* table) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)table);
}
// x = the value in bits [15:0] (higher bits: anything)
// (written in asm: the optimiser rewrites the C form of these two into shift + mask + or)
__device__ __forceinline__ lds_u32* table_dword_lo(uint32_t base, uint32_t x) {
  uint32_t i, a;
  asm("v_bfe_u32 %0, %1, 5, 11" : "=v"(i) : "v"(x));
  asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(a) : "v"(i), "s"(base));
  return (lds_u32*)(uintptr_t)a;
}
// x = the value in bits [31:16]
__device__ __forceinline__ lds_u32* table_dword_hi(uint32_t base, uint32_t x) {
  uint32_t i, a;
  asm("v_lshrrev_b32 %0, 21, %1" : "=v"(i) : "v"(x));
  asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(a) : "v"(i), "s"(base));
  return (lds_u32*)(uintptr_t)a;
}
__device__ __forceinline__ void toggle_lo(uint32_t base, uint32_t x) {
  (void)__hip_atomic_fetch_xor(table_dword_lo(base, x), 1u << (x & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void toggle_hi(uint32_t base, uint32_t x) {
  (void)__hip_atomic_fetch_xor(table_dword_hi(base, x), 1u << ((x >> 16) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// XOR the raw bits of one batch into the table (array: one bit per value; run: a toggle at start and at last + 1)
__device__ __forceinline__ void sparse_xor_batch(uint32_t type, uint32_t tb, uint32_t len, uint32_t base, int lane,
                                                 const uint32_t (&v)[kPairBatch]) {
  if (type == kTypeArray) {
    // ONE body, every value under its own test.  (Until round 5 a batch whose values all exist had an unpredicated copy of the loop:
    // with two copies of every eight-row loop in every role of every type pair k_icount2 was 102 KB of code, now 49 KB.  Measured on
    // config 3's 8192 row pairs, old and new library in one call: 159-160 against 161 us — the instruction cache (64 KB per pair of
    // CUs) was NOT what the kernel waits for, profiles/r05_pairs_code_size_ab.txt; the smaller form stays because it is the simpler one.
    // The array that is scattered is the shorter one: it rarely filled a batch anyway.)
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) {
      const uint32_t i2 = (base + (uint32_t)k * kWave + (uint32_t)lane) * 2u;
      if (i2 < len) toggle_lo(tb, v[k]);
      if (i2 + 1u < len) toggle_hi(tb, v[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) {
      const uint32_t i = base + (uint32_t)k * kWave + (uint32_t)lane;
      if (i < len) {
        const uint32_t e = (v[k] >> 16) + 1u;
        toggle_lo(tb, v[k]);
        if (e < 65536u) toggle_lo(tb, e);
      }
    }
  }
}

// the whole container, batch 0 already in v: the next batch is in flight while the current one is applied
__device__ __forceinline__ void sparse_xor_all(uint32_t type, const uint8_t* __restrict__ p, uint32_t len, int lane, uint32_t tb,
                                               uint32_t (&v)[kPairBatch]) {
  const uint32_t n_units = sparse_units(type, len);
  for (uint32_t base = 0;;) {
    const uint32_t nb = base + kPairBatch * kWave;
    uint32_t nv[kPairBatch];
    if (nb < n_units) sparse_load(p, n_units, nb, lane, nv);
    sparse_xor_batch(type, tb, len, base, lane, v);
    if (nb >= n_units) break;
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = nv[k];
    base = nb;
  }
}

// (w: the bit field's width — 1, or 0 for a value that is not there: v_bfe_u32 then yields 0 whatever the dword holds)
__device__ __forceinline__ uint32_t table_bit_lo(uint32_t tb, uint32_t x, uint32_t w = 1u) { return __builtin_amdgcn_ubfe(*table_dword_lo(tb, x), x, w); }
__device__ __forceinline__ uint32_t table_bit_hi(uint32_t tb, uint32_t x, uint32_t w = 1u) { return __builtin_amdgcn_ubfe(*table_dword_hi(tb, x), x >> 16, w); }

// bit (value >> 5) of a 2048-bit map at LDS byte offset mb: one bit per DWORD of the table (the interior map of a run
// container after its parity prefix, see run_fill_batch); value in bits [15:0] / [31:16] of x
__device__ __forceinline__ uint32_t map_bit_lo(uint32_t mb, uint32_t x, uint32_t width = 1u) {
  const uint32_t w = *(const lds_u32*)(uintptr_t)(mb + (__builtin_amdgcn_ubfe(x, 10u, 6u) << 2));
  return __builtin_amdgcn_ubfe(w, x >> 5, width);  // (bit-field offsets use bits [4:0] only)
}
__device__ __forceinline__ uint32_t map_bit_hi(uint32_t mb, uint32_t x, uint32_t width = 1u) {
  const uint32_t w = *(const lds_u32*)(uintptr_t)(mb + ((x >> 26) << 2));
  return __builtin_amdgcn_ubfe(w, x >> 21, width);
}

// ---- the array forms of the COUNT (round 6) ---------------------------------------------------------------------------------------
// An array that is scattered into the table or that probes one (arrays_table_probe, bitmap_table_probe, run_table_probe) is read
// as its floor(len / 2) FULL dwords — `nd` below; item_prefetch asks count_units for the unit count — and an odd last value goes
// by itself (one scalar load, one LDS operation).  What that buys: no value of a dword row needs a test of its own.
//   * the SCATTER predicates a ROW (`lane < dwords left`, one compare per two values; rows past the end are not executed);
//   * the PROBE predicates nothing: a dword past the end was loaded as zero (sparse_load), so each of its two junk values
//     "hits" exactly when value 0 is in the table — the same answer z in every lane — and the wave takes junk x z off its
//     scalar partial instead.
// Round 5's form tested every value (compare + select, a third of the probe's ~6 vector instructions per value), and the
// compiler interleaved its reads and uses two at a time: EIGHT LDS round trips per batch one after the other
// (`s_waitcnt lgkmcnt(0)` behind every second read), where all sixteen reads of a batch can be in flight at once.  The SQ
// counters of the shipped kernel on config 3's 8192 row pairs (profiles/r04_pmc_pairs_shipped.txt; the counters tick in
// QUAD cycles) say what that costs: 49.8 M vector instructions = 85 us of every SIMD's issue slots out of the kernel's 150,
// and waves waiting 61 % of their lives.

// value x (bits [15:0]) is in the table (MAP: or in a dword the run map marks full)
template <bool MAP>
__device__ __forceinline__ uint32_t probe_value(uint32_t tb, uint32_t mb, uint32_t x) {
  uint32_t b = table_bit_lo(tb, x);
  if (MAP) b |= map_bit_lo(mb, x);
  return b;
}

// rows K0 .. K1 - 1 of a batch: every table read is issued before the first one is used (the scheduling barrier)
template <bool MAP, int K0, int K1>
__device__ __forceinline__ uint32_t array_probe_rows(uint32_t tb, uint32_t mb, const uint32_t (&v)[kPairBatch]) {
  uint32_t tl[kPairBatch], th[kPairBatch], ml[kPairBatch], mh[kPairBatch];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    tl[k] = *table_dword_lo(tb, v[k]);
    th[k] = *table_dword_hi(tb, v[k]);
    if (MAP) {
      ml[k] = *(const lds_u32*)(uintptr_t)(mb + (__builtin_amdgcn_ubfe(v[k], 10u, 6u) << 2));
      mh[k] = *(const lds_u32*)(uintptr_t)(mb + ((v[k] >> 26) << 2));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  uint32_t hits = 0;
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    uint32_t b0 = __builtin_amdgcn_ubfe(tl[k], v[k], 1u), b1 = __builtin_amdgcn_ubfe(th[k], v[k] >> 16, 1u);  // (bit-field offsets use bits [4:0] only)
    if (MAP) {
      b0 |= __builtin_amdgcn_ubfe(ml[k], v[k] >> 5, 1u);
      b1 |= __builtin_amdgcn_ubfe(mh[k], v[k] >> 21, 1u);
    }
    hits += b0 + b1;
  }
  return hits;
}

// this lane's RAW hits of the batch at dword `base` (nd > base), junk values included; `junk` counts those (wave-uniform)
template <bool MAP = false>
__device__ __forceinline__ uint32_t array_probe_batch(uint32_t tb, uint32_t nd, uint32_t base, const uint32_t (&v)[kPairBatch], uint32_t mb, uint32_t& junk) {
  const uint32_t rem = nd - base;  // dwords from this batch on
  uint32_t hits = array_probe_rows<MAP, 0, kPairBatch / 2>(tb, mb, v);
  if (rem > (kPairBatch / 2) * kWave) {
    hits += array_probe_rows<MAP, kPairBatch / 2, kPairBatch>(tb, mb, v);
    junk += rem >= kPairBatch * kWave ? 0u : 2u * (kPairBatch * kWave - rem);
  } else {
    junk += 2u * ((kPairBatch / 2) * kWave - rem);
  }
  return hits;
}

// The probing array of an item, ALL of it in flight at once: batch 0 came with the item's prefetch, batches 1..3
// (an array has at most 4095 values = 4 batches unless it is an oversized intermediate result) are requested by
// probe_tail_load BEFORE the table is built, so that an array of any legal size costs one memory round trip, not
// one per batch (a batch is probed in a tenth of the time its loads take to arrive).
struct ProbeTail {
  uint32_t v1[kPairBatch], v2[kPairBatch], v3[kPairBatch];
};
// (nd: the dword units that are read — len >> 1 in the count, (len + 1) >> 1 where the odd last value rides in the last dword)
__device__ __forceinline__ void probe_tail_load(const uint8_t* __restrict__ p, uint32_t nd, int lane, ProbeTail& t) {
  constexpr uint32_t B = kPairBatch * kWave;
  if (nd > B) sparse_load(p, nd, B, lane, t.v1);
  if (nd > 2 * B) sparse_load(p, nd, 2 * B, lane, t.v2);
  if (nd > 3 * B) sparse_load(p, nd, 3 * B, lane, t.v3);
}
// number of this lane's array values that are set in the table; the junk correction and an odd last value go to `spart`
// (wave-uniform, added once per wave)
template <bool MAP = false>
__device__ __forceinline__ uint32_t array_probe_all(const uint8_t* __restrict__ p, uint32_t len, int lane, uint32_t tb,
                                                    const uint32_t (&v0)[kPairBatch], const ProbeTail& t, uint32_t& spart, uint32_t mb = 0) {
  const uint32_t nd = len >> 1;
  constexpr uint32_t B = kPairBatch * kWave;
  uint32_t odd_value = 0;
  if (len & 1u) odd_value = reinterpret_cast<const uint32_t*>(p)[nd] & 0xFFFFu;  // (wave-uniform address: a scalar load)
  uint32_t hits = 0, junk = 0;
  if (nd > 0) hits += array_probe_batch<MAP>(tb, nd, 0, v0, mb, junk);
  if (nd > B) hits += array_probe_batch<MAP>(tb, nd, B, t.v1, mb, junk);
  if (nd > 2 * B) hits += array_probe_batch<MAP>(tb, nd, 2 * B, t.v2, mb, junk);
  if (nd > 3 * B) hits += array_probe_batch<MAP>(tb, nd, 3 * B, t.v3, mb, junk);
  for (uint32_t base = 4 * B; base < nd; base += B) {  // arrays beyond 4096 values (roaring.go:5054)
    uint32_t v[kPairBatch];
    sparse_load(p, nd, base, lane, v);
    hits += array_probe_batch<MAP>(tb, nd, base, v, mb, junk);
  }
  if (junk) spart -= junk * (uint32_t)__builtin_amdgcn_readfirstlane((int)probe_value<MAP>(tb, mb, 0u));
  if (len & 1u) spart += (uint32_t)__builtin_amdgcn_readfirstlane((int)probe_value<MAP>(tb, mb, odd_value));
  return hits;
}

// one batch of an array's full dwords ORed into the table (the values of an array are distinct: OR = the toggle of sparse_xor_batch)
__device__ __forceinline__ void array_or_batch(uint32_t tb, uint32_t nd, uint32_t base, int lane, const uint32_t (&v)[kPairBatch]) {
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) {
    const int32_t left = (int32_t)(nd - base) - k * kWave;  // dwords from this row on (wave-uniform)
    if (left <= 0) break;
    if (lane < left) {
      (void)__hip_atomic_fetch_or(table_dword_lo(tb, v[k]), 1u << (v[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_or(table_dword_hi(tb, v[k]), 1u << ((v[k] >> 16) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
// the whole array (batch 0 in v; an odd last value by one lane)
__device__ __forceinline__ void array_or_all(const uint8_t* __restrict__ p, uint32_t len, int lane, uint32_t tb, uint32_t (&v)[kPairBatch]) {
  const uint32_t nd = len >> 1;
  uint32_t odd_value = 0;
  if (len & 1u) odd_value = reinterpret_cast<const uint32_t*>(p)[nd] & 0xFFFFu;
  for (uint32_t base = 0; base < nd;) {
    const uint32_t nb = base + kPairBatch * kWave;
    uint32_t nv[kPairBatch];
    if (nb < nd) sparse_load(p, nd, nb, lane, nv);
    array_or_batch(tb, nd, base, lane, v);
    if (nb >= nd) break;
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = nv[k];
    base = nb;
  }
  if ((len & 1u) && lane == 0)
    (void)__hip_atomic_fetch_or((lds_u32*)(uintptr_t)(tb + ((odd_value >> 5) << 2)), 1u << (odd_value & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// dword units of the two payloads' batches as the count kernel reads them: floor(len / 2) for an array on one of the table
// paths above (the dispatch of icount_item, restated), sparse_units otherwise
__device__ __forceinline__ void count_units(uint32_t ta, uint32_t la, uint32_t tb, uint32_t lb, uint32_t sparse_paths, uint32_t& ua, uint32_t& ub);

// toggles -> filled runs: the inclusive parity prefix over the 65536 bits of a fragment (see frag_load_run)
__device__ __forceinline__ void frag_parity_prefix(u64 (&w)[kWordsPerLane], int lane) {
  const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t carry = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const u64 t0 = w[2 * j], t1 = w[2 * j + 1];
    const uint32_t p0 = __popcll(t0) & 1u, p1 = __popcll(t1) & 1u;
    const u64 m = __ballot((p0 ^ p1) != 0);
    const uint32_t in = carry ^ (__popcll(m & lane_lt) & 1u);
    w[2 * j] = prefix_xor64(t0) ^ (in ? ~0ull : 0ull);
    w[2 * j + 1] = prefix_xor64(t1) ^ ((in ^ p0) ? ~0ull : 0ull);
    carry ^= __popcll(m) & 1u;
  }
}

// ---- run containers of few runs: boundary masks + a 2048-bit interior map ----------------------------------
// The parity prefix above costs ~400 vector instructions per run container whatever it holds (16 words x a 6-step
// 64-bit shift-xor ladder + the carries), which made the run pairs two thirds of k_icount2's instruction count on
// config 3's rows.  A run [s, l] is instead written as what it is: a partial mask in its first and in its last
// dword (XORed into the table: the runs of one container are disjoint, so XOR is OR — and stays separable from
// another operand's raw bits underneath), and every dword strictly between them is FULL: those are recorded as
// two toggles in a map with ONE BIT PER DWORD (2048 bits = 64 lanes x 32: the "mini table", 256 bytes per
// operand), whose parity prefix is one dword per lane — 5 shift-xor steps and one ballot for the carries.  Each
// lane then ORs 0xFFFFFFFF into the dwords of its fragment whose map bit is set.  ~100 instructions + ~30 per 64
// runs; containers of more than kRunFillMax runs keep the toggle form, which is cheaper per run.
constexpr uint32_t kRunFillMax = 600;
constexpr int kMiniDwords = 64;  // per operand

__device__ __forceinline__ void run_fill_batch(uint32_t tb, uint32_t mb, uint32_t len, uint32_t base, int lane, const uint32_t (&v)[kPairBatch]) {
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) {
    const uint32_t i = base + (uint32_t)k * kWave + (uint32_t)lane;
    if (i < len) {
      const uint32_t s = v[k] & 0xFFFFu, l = v[k] >> 16;
      const uint32_t ds = s >> 5, dl = l >> 5;
      const uint32_t ms = 0xFFFFFFFFu << (s & 31u), ml = 0xFFFFFFFFu >> (~l & 31u);
      lds_u32* first = (lds_u32*)(uintptr_t)(tb + (ds << 2));
      if (ds == dl) {
        (void)__hip_atomic_fetch_xor(first, ms & ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        (void)__hip_atomic_fetch_xor(first, ms, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        (void)__hip_atomic_fetch_xor((lds_u32*)(uintptr_t)(tb + (dl << 2)), ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (dl - ds > 1u) {  // dwords ds + 1 .. dl - 1 are full
          const uint32_t d0 = ds + 1u;
          (void)__hip_atomic_fetch_xor((lds_u32*)(uintptr_t)(mb + ((d0 >> 5) << 2)), 1u << (d0 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_xor((lds_u32*)(uintptr_t)(mb + ((dl >> 5) << 2)), 1u << (dl & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
}

__device__ __forceinline__ void run_fill_all(const uint8_t* __restrict__ p, uint32_t len, int lane, uint32_t tb, uint32_t mb, uint32_t (&v)[kPairBatch]) {
  for (uint32_t base = 0;;) {
    const uint32_t nb = base + kPairBatch * kWave;
    uint32_t nv[kPairBatch];
    if (nb < len) sparse_load(p, len, nb, lane, nv);
    run_fill_batch(tb, mb, len, base, lane, v);
    if (nb >= len) break;
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = nv[k];
    base = nb;
  }
}

// OR the full interior dwords recorded in the mini table at byte offset mb into a fragment of boundary masks
__device__ __forceinline__ void frag_or_interior(u64 (&w)[kWordsPerLane], uint32_t mb, int lane) {
  lds_u32* mine = (lds_u32*)(uintptr_t)(mb + 4u * (uint32_t)lane);
  uint32_t x = *mine;
  x ^= x << 1;
  x ^= x << 2;
  x ^= x << 4;
  x ^= x << 8;
  x ^= x << 16;  // inclusive parity prefix inside the dword; bit 31 = parity of the whole dword
  const u64 odd = __ballot((x >> 31) != 0);
  const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (__popcll(odd & lane_lt) & 1) x = ~x;
  *mine = x;
  wave_lds_sync();
  const lds_u32* row = (const lds_u32*)(uintptr_t)(mb + 4u * ((uint32_t)lane >> 3));  // + 32 j bytes: dword 8 j + lane / 8
  const uint32_t sh = 4u * ((uint32_t)lane & 7u);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t nib = row[8 * j] >> sh;  // bits 0..3: dwords 256 j + 4 lane + 0..3
    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 0, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 1, 1);
    const uint32_t m2 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 2, 1), m3 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 3, 1);
    w[2 * j] |= (u64)m0 | ((u64)m1 << 32);
    w[2 * j + 1] |= (u64)m2 | ((u64)m3 << 32);
  }
  wave_lds_sync();
}

// A run container of <= kRunFillMax intervals (batch 0 in vr) as something an array can PROBE: boundary masks in the cleared
// table, and the map of the dwords that lie entirely inside a run (after its parity prefix: one bit per dword, 2048 bits in the
// first 64 dwords of `mini`).  Value x is in the container iff its table bit or the map bit of its dword is set.
__device__ __forceinline__ void run_table_build(const uint8_t* __restrict__ pr, uint32_t lr, uint32_t (&vr)[kPairBatch], int lane, u64* table, uint32_t* mini,
                                                uint32_t& tbase, uint32_t& mbase) {
  tbase = lds_table_base(table);
  mbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)mini);
  lds_zero(table, lane);
  mini[lane] = 0;
  wave_lds_sync();
  run_fill_all(pr, lr, lane, tbase, mbase, vr);
  wave_lds_sync();
  uint32_t x = mini[lane];  // toggles -> filled: the parity prefix of the 2048-bit map, one dword per lane (as run_finish_init)
  x ^= x << 1;
  x ^= x << 2;
  x ^= x << 4;
  x ^= x << 8;
  x ^= x << 16;
  const u64 odd = __ballot((x >> 31) != 0);
  const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (__popcll(odd & lane_lt) & 1) x = ~x;
  mini[lane] = x;
  wave_lds_sync();
}

// |array ∩ run container| by probing (intersectionCountArrayRun, roaring.go:4537-4557, walks both lists; pair_stream below
// decodes both operands into 8 KiB): this lane's hits.  Round 4.
__device__ __forceinline__ uint32_t run_table_probe(const uint8_t* __restrict__ pr, uint32_t lr, uint32_t (&vr)[kPairBatch], const uint8_t* __restrict__ pp,
                                                    uint32_t lp, const uint32_t (&vp)[kPairBatch], int lane, u64* table, uint32_t* mini, uint32_t& spart) {
  ProbeTail tail;
  probe_tail_load(pp, lp >> 1, lane, tail);
  uint32_t tbase, mbase;
  run_table_build(pr, lr, vr, lane, table, mini, tbase, mbase);
  const uint32_t h = array_probe_all<true>(pp, lp, lane, tbase, vp, tail, spart, mbase);
  wave_lds_sync();
  return h;
}

// Both containers of an item as register fragments, batch 0 of the sparse ones already in va / vb: bitmaps stream to
// registers, the first sparse operand's raw bits go into the cleared table and are read back, the second's go ON
// TOP and come back as (first ^ second) ^ first.  ONE clear of the 8 KiB table per item.  ta, tb: array / bitmap / run.
__device__ __forceinline__ void frag_pair_decode(uint32_t ta, const uint8_t* __restrict__ pa, uint32_t lena, uint32_t tb,
                                                 const uint8_t* __restrict__ pb, uint32_t lenb, int lane, u64* table, uint32_t* mini,
                                                 uint32_t (&va)[kPairBatch], uint32_t (&vb)[kPairBatch], u64 (&wa)[kWordsPerLane],
                                                 u64 (&wb)[kWordsPerLane]) {
  const bool la = ta == kTypeArray || ta == kTypeRun, lb = tb == kTypeArray || tb == kTypeRun;  // wave-uniform
  if (ta == kTypeBitmap) frag_load_bitmap(pa, lane, wa);
  if (tb == kTypeBitmap) frag_load_bitmap(pb, lane, wb);
  if (ta == kTypeNil) frag_zero(wa);
  if (tb == kTypeNil) frag_zero(wb);
  if (!la && !lb) return;
  const bool fa = ta == kTypeRun && lena <= kRunFillMax, fb = tb == kTypeRun && lenb <= kRunFillMax;
  const uint32_t tbase = lds_table_base(table);
  const uint32_t mbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)mini);
  lds_zero(table, lane);
  if (fa || fb) {
    uint2 z;
    z.x = 0;
    z.y = 0;
    reinterpret_cast<uint2*>(mini)[lane] = z;  // both operands' maps: 2 x 64 dwords
  }
  wave_lds_sync();
  if (la) {
    if (fa) run_fill_all(pa, lena, lane, tbase, mbase, va);
    else sparse_xor_all(ta, pa, lena, lane, tbase, va);
    wave_lds_sync();
    lds_read_frag(table, lane, wa);  // raw bits of A: array bits / run toggles / run boundary masks
    wave_lds_sync();
  }
  if (lb) {
    if (fb) run_fill_all(pb, lenb, lane, tbase, mbase + 4u * kMiniDwords, vb);
    else sparse_xor_all(tb, pb, lenb, lane, tbase, vb);  // on top of A's raw bits
    wave_lds_sync();
    lds_read_frag(table, lane, wb);
    wave_lds_sync();
    if (la) {
#pragma unroll
      for (int i = 0; i < kWordsPerLane; ++i) wb[i] ^= wa[i];
    }
  }
  if (ta == kTypeRun) {
    if (fa) frag_or_interior(wa, mbase, lane);
    else frag_parity_prefix(wa, lane);
  }
  if (tb == kTypeRun) {
    if (fb) frag_or_interior(wb, mbase + 4u * kMiniDwords, lane);
    else frag_parity_prefix(wb, lane);
  }
}

// ---- the streaming form of the pair decode: the consumer sees the two containers 1 KiB at a time -----------------
// frag_pair_decode hands back two whole fragments (64 registers).  A consumer that only folds them — the pair count —
// does not need them whole: pair_stream keeps ONE raw fragment (the first sparse operand's, needed to take it back
// out of what the second wrote on top; or the one bitmap operand) and calls f(a0, a1, b0, b1) for the eight pairs of
// words of the lane's fragment as they come out of the table.  32 registers less, which is what lets the next item's
// payload prefetch and the probing array's tail sit in registers without spilling.
struct RunFinish {
  uint32_t mode;   // 0: nothing to do (array / bitmap), 1: toggles -> parity prefix, 2: boundary masks + interior map
  uint32_t carry;  // mode 1: parity of all toggles in earlier words
  const lds_u32* row;
  uint32_t sh;
};

__device__ __forceinline__ void run_finish_init(RunFinish& r, uint32_t type, uint32_t len, uint32_t mb, int lane) {
  r.mode = type != kTypeRun ? 0u : (len <= kRunFillMax ? 2u : 1u);
  r.carry = 0;
  r.row = (const lds_u32*)(uintptr_t)(mb + 4u * ((uint32_t)lane >> 3));
  r.sh = 4u * ((uint32_t)lane & 7u);
  if (r.mode == 2u) {  // parity prefix of the interior map, in place (the caller synchronises before the first step)
    lds_u32* mine = (lds_u32*)(uintptr_t)(mb + 4u * (uint32_t)lane);
    uint32_t x = *mine;
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    const u64 odd = __ballot((x >> 31) != 0);
    const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (__popcll(odd & lane_lt) & 1) x = ~x;
    *mine = x;
  }
}

template <int J>
__device__ __forceinline__ void run_finish_step(RunFinish& r, u64& w0, u64& w1, int lane) {
  if (r.mode == 2u) {
    const uint32_t nib = r.row[8 * J] >> r.sh;
    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 0, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 1, 1);
    const uint32_t m2 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 2, 1), m3 = (uint32_t)__builtin_amdgcn_sbfe((int)nib, 3, 1);
    w0 |= (u64)m0 | ((u64)m1 << 32);
    w1 |= (u64)m2 | ((u64)m3 << 32);
  } else if (r.mode == 1u) {
    const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t p0 = __popcll(w0) & 1u, p1 = __popcll(w1) & 1u;
    const u64 m = __ballot((p0 ^ p1) != 0);
    const uint32_t in = r.carry ^ (__popcll(m & lane_lt) & 1u);
    w0 = prefix_xor64(w0) ^ (in ? ~0ull : 0ull);
    w1 = prefix_xor64(w1) ^ ((in ^ p0) ? ~0ull : 0ull);
    r.carry ^= __popcll(m) & 1u;
  }
}

template <int J, class F>
__device__ __forceinline__ void pair_stream_steps(bool la, bool lb, const u64 (&keep)[kWordsPerLane], const ulonglong2* q, RunFinish& ra, RunFinish& rb,
                                                  int lane, F& f) {
  if constexpr (J < 8) {
    u64 a0, a1, b0, b1;
    if (la && lb) {
      const ulonglong2 x = q[J * kWave + lane];
      a0 = keep[2 * J];
      a1 = keep[2 * J + 1];
      b0 = x.x ^ a0;
      b1 = x.y ^ a1;
    } else if (la) {
      const ulonglong2 x = q[J * kWave + lane];
      a0 = x.x;
      a1 = x.y;
      b0 = keep[2 * J];
      b1 = keep[2 * J + 1];
    } else {
      const ulonglong2 x = q[J * kWave + lane];
      a0 = keep[2 * J];
      a1 = keep[2 * J + 1];
      b0 = x.x;
      b1 = x.y;
    }
    run_finish_step<J>(ra, a0, a1, lane);
    run_finish_step<J>(rb, b0, b1, lane);
    f(a0, a1, b0, b1);
    pair_stream_steps<J + 1>(la, lb, keep, q, ra, rb, lane, f);
  }
}

// ta, tb: array / bitmap / run, at least one of them sparse; batch 0 of the sparse ones in va / vb
template <class F>
__device__ __forceinline__ void pair_stream(uint32_t ta, const uint8_t* __restrict__ pa, uint32_t lena, uint32_t tb, const uint8_t* __restrict__ pb,
                                            uint32_t lenb, int lane, u64* table, uint32_t* mini, uint32_t (&va)[kPairBatch],
                                            uint32_t (&vb)[kPairBatch], F f) {
  const bool la = ta != kTypeBitmap, lb = tb != kTypeBitmap;  // wave-uniform
  u64 keep[kWordsPerLane];  // the bitmap operand, or the raw bits of A when both are sparse
  if (!la) frag_load_bitmap(pa, lane, keep);
  if (!lb) frag_load_bitmap(pb, lane, keep);
  const bool fa = ta == kTypeRun && lena <= kRunFillMax, fb = tb == kTypeRun && lenb <= kRunFillMax;
  const uint32_t tbase = lds_table_base(table);
  const uint32_t mbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)mini);
  lds_zero(table, lane);
  if (fa || fb) {
    uint2 z;
    z.x = 0;
    z.y = 0;
    reinterpret_cast<uint2*>(mini)[lane] = z;
  }
  wave_lds_sync();
  if (la) {
    if (fa) run_fill_all(pa, lena, lane, tbase, mbase, va);
    else sparse_xor_all(ta, pa, lena, lane, tbase, va);
    wave_lds_sync();
    if (lb) {
      lds_read_frag(table, lane, keep);  // raw bits of A, before B goes on top
      wave_lds_sync();
    }
  }
  if (lb) {
    if (fb) run_fill_all(pb, lenb, lane, tbase, mbase + 4u * kMiniDwords, vb);
    else sparse_xor_all(tb, pb, lenb, lane, tbase, vb);
    wave_lds_sync();
  }
  RunFinish ra, rb;
  run_finish_init(ra, ta, lena, mbase, lane);
  run_finish_init(rb, tb, lenb, mbase + 4u * kMiniDwords, lane);
  wave_lds_sync();
  pair_stream_steps<0>(la, lb, keep, reinterpret_cast<const ulonglong2*>(table), ra, rb, lane, f);
  wave_lds_sync();
}

// The same for callers that hold only the descriptors (k_setop2): nil operands come back as zero fragments.
__device__ __forceinline__ void frag_load_pair(const Slot& sa, const uint8_t* __restrict__ arenaA, const Slot& sb,
                                               const uint8_t* __restrict__ arenaB, int lane, u64* table, uint32_t* mini,
                                               u64 (&wa)[kWordsPerLane], u64 (&wb)[kWordsPerLane]) {
  const uint32_t ta = slot_n(sa) ? slot_type(sa) : kTypeNil, tb = slot_n(sb) ? slot_type(sb) : kTypeNil;
  const uint8_t* pa = arenaA + sa.off;
  const uint8_t* pb = arenaB + sb.off;
  uint32_t va[kPairBatch], vb[kPairBatch];
  if (ta == kTypeArray || ta == kTypeRun) sparse_load(pa, sparse_units(ta, sa.len), 0, lane, va);
  if (tb == kTypeArray || tb == kTypeRun) sparse_load(pb, sparse_units(tb, sb.len), 0, lane, vb);
  frag_pair_decode(ta, pa, sa.len, tb, pb, sb.len, lane, table, mini, va, vb, wa, wb);
}

// ---- |A ∩ B| over row pairs, any mix of encodings (intersectionCount, roaring.go:4477-4614) -------------
//
// One WAVE = SPW consecutive slots of one row pair.  Everything that decides control flow is wave-uniform
// and held in SCALAR registers: the wave index comes from v_readfirstlane, so the row indexes and the 2 x SPW
// descriptors are s_loads (the round-2 kernel computed them per lane: four dependent VECTOR round trips —
// row index, cardinality, offset / length, payload — before the first payload byte arrived, ~10 us of a
// wave's life at five waves per SIMD; that chain, not the LDS, was what the 70 us of k_icount were made of)
// and every loop has a uniform trip count.  The first payload batch of slot k + 1 is in flight while slot k
// is worked on, and the wave adds ONE count to out[pair].

// batch 0 of both operands of an item, if the item will be decoded at all
__device__ __forceinline__ void count_units(uint32_t ta, uint32_t la, uint32_t tb, uint32_t lb, uint32_t sparse_paths, uint32_t& ua, uint32_t& ub) {
  ua = sparse_units(ta, la);
  ub = sparse_units(tb, lb);
  bool fl = false;  // wave-uniform
  if (ta == kTypeArray && tb == kTypeArray) fl = !((sparse_paths & 1u) && la <= kSmallArray && lb <= kSmallArray);
  else if (ta == kTypeArray && tb == kTypeBitmap) fl = !((sparse_paths & 1u) && la <= kProbeArray);
  else if (ta == kTypeBitmap && tb == kTypeArray) fl = !((sparse_paths & 1u) && lb <= kProbeArray);
  else if (ta == kTypeArray && tb == kTypeRun) fl = (sparse_paths & 2u) && lb <= kRunFillMax;
  else if (ta == kTypeRun && tb == kTypeArray) fl = (sparse_paths & 2u) && la <= kRunFillMax;
  if (fl) {
    if (ta == kTypeArray) ua = la >> 1;
    if (tb == kTypeArray) ub = lb >> 1;
  }
}

__device__ __forceinline__ void item_prefetch(const Slot& sa, const uint8_t* __restrict__ arenaA, const Slot& sb,
                                              const uint8_t* __restrict__ arenaB, int lane, uint32_t (&va)[kPairBatch],
                                              uint32_t (&vb)[kPairBatch], uint32_t sparse_paths) {
  const uint32_t na = slot_n(sa), nb = slot_n(sb);
  if (na == 0 || nb == 0 || na == 65536u || nb == 65536u) return;
  const uint32_t ta = slot_type(sa), tb = slot_type(sb);
  uint32_t ua, ub;
  count_units(ta, sa.len, tb, sb.len, sparse_paths, ua, ub);
  if (ta == kTypeArray || ta == kTypeRun) sparse_load(arenaA + sa.off, ua, 0, lane, va);
  if (tb == kTypeArray || tb == kTypeRun) sparse_load(arenaB + sb.off, ub, 0, lane, vb);
}

// the two operands' first batches change places (in place: one temporary at a time)
__device__ __forceinline__ void batch_exchange(uint32_t (&a)[kPairBatch], uint32_t (&b)[kPairBatch]) {
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) {
    const uint32_t t = a[k];
    a[k] = b[k];
    b[k] = t;
  }
}

// shorter array -> table, longer array probes it; returns this lane's hits
__device__ __forceinline__ uint32_t arrays_table_probe(const uint8_t* __restrict__ pt, uint32_t lt, uint32_t (&vt)[kPairBatch],
                                                       const uint8_t* __restrict__ pp, uint32_t lp, uint32_t (&vp)[kPairBatch], int lane,
                                                       u64* table, uint32_t& spart) {
  ProbeTail tail;
  probe_tail_load(pp, lp >> 1, lane, tail);
  const uint32_t tbase = lds_table_base(table);
  lds_zero(table, lane);
  wave_lds_sync();
  array_or_all(pt, lt, lane, tbase, vt);
  wave_lds_sync();
  const uint32_t h = array_probe_all(pp, lp, lane, tbase, vp, tail, spart);
  wave_lds_sync();
  return h;
}

// bitmap -> table (no clear), array probes it; returns this lane's hits
__device__ __forceinline__ uint32_t bitmap_table_probe(const uint8_t* __restrict__ pbm, const uint8_t* __restrict__ parr, uint32_t larr,
                                                       uint32_t (&vp)[kPairBatch], int lane, u64* table, uint32_t& spart) {
  u64 wb[kWordsPerLane];
  frag_load_bitmap(pbm, lane, wb);
  ProbeTail tail;
  probe_tail_load(parr, larr >> 1, lane, tail);
  ulonglong2* q = reinterpret_cast<ulonglong2*>(table);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 x;
    x.x = wb[2 * j];
    x.y = wb[2 * j + 1];
    q[j * kWave + lane] = x;  // fragment layout -> natural word order in the table
  }
  wave_lds_sync();
  const uint32_t h = array_probe_all(parr, larr, lane, lds_table_base(table), vp, tail, spart);
  wave_lds_sync();
  return h;
}

// values (<= 128: one dword of two values per lane, in v[0]) of an array that are set in a bitmap container in
// global memory.  (Round 4 tried the same for up to 1024 values — 16 gather loads per lane instead of copying the bitmap's
// 8 KiB through the wave's table: 167.3 / 166.9 / 170.4 us for thresholds 128 / 512 / 1024 on config 3's 8192 row pairs,
// profiles/r04_pairs_probe_max_ab.json — no gain, removed again.)
__device__ __forceinline__ uint32_t array_probe_global(const uint32_t (&v)[kPairBatch], uint32_t len, const uint8_t* __restrict__ pbm, int lane) {
  const uint32_t* bm = reinterpret_cast<const uint32_t*>(pbm);
  const uint32_t i2 = 2u * (uint32_t)lane, lo = v[0] & 0xFFFFu, hi = v[0] >> 16;
  uint32_t h = 0;
  if (i2 < len) h += (bm[lo >> 5] >> (lo & 31u)) & 1u;
  if (i2 + 1u < len) h += (bm[hi >> 5] >> (hi & 31u)) & 1u;
  return h;
}

// one (pair, slot): adds to `part` (per lane) or `spart` (wave-uniform)
__device__ __forceinline__ void icount_item(const Slot& sa, const uint8_t* __restrict__ arenaA, const Slot& sb,
                                            const uint8_t* __restrict__ arenaB, int lane, u64* table, uint32_t* mini, uint32_t (&va)[kPairBatch],
                                            uint32_t (&vb)[kPairBatch], uint32_t sparse_paths, uint32_t& part, uint32_t& spart) {
  // The lane index is laundered through an empty asm per item: otherwise the optimiser hoists every per-lane
  // address of every path (table rows, payload pointers, masks) out of the item loop as "loop invariant",
  // runs out of registers and spills them — one scratch reload per use, in the hot path (measured: 247 spilled
  // registers in the two-slots-per-wave kernel).
  asm volatile("" : "+v"(lane));
  const uint32_t na = slot_n(sa), nb = slot_n(sb);
  const uint32_t ta = slot_type(sa), tb = slot_type(sb);
  // timing experiments (option pair_ablate, WRONG results): 2 = no item is decoded, 8 = items with a run are skipped,
  // 16 = array x array items are skipped, 32 = bitmap x array items are skipped
  if (sparse_paths & 0x200u) return;
  if ((sparse_paths & 0x800u) && (ta == kTypeRun || tb == kTypeRun)) return;
  if ((sparse_paths & 0x1000u) && ta == kTypeArray && tb == kTypeArray) return;
  if ((sparse_paths & 0x2000u) && ((ta == kTypeArray && tb == kTypeBitmap) || (ta == kTypeBitmap && tb == kTypeArray))) return;
  const uint8_t* pa = arenaA + sa.off;
  const uint8_t* pb = arenaB + sb.off;
  if (na == 0 || nb == 0) return;
  if (na == 65536u) {
    spart += nb;
  } else if (nb == 65536u) {
    spart += na;
  } else if (ta == kTypeBitmap && tb == kTypeBitmap) {
    u64 wa[kWordsPerLane], wb[kWordsPerLane];
    frag_load_bitmap(pa, lane, wa);
    frag_load_bitmap(pb, lane, wb);
#pragma unroll
    for (int i = 0; i < kWordsPerLane; ++i) part += __popcll(wa[i] & wb[i]);
  } else if (ta == kTypeArray && tb == kTypeArray) {
    if ((sparse_paths & 1u) && sa.len <= kSmallArray && sb.len <= kSmallArray) {
      // all-pairs compare in registers (two values per lane, lanes 0..31), the shorter array broadcast value by value
      const bool a_long = sa.len >= sb.len;
      const uint32_t lng = a_long ? va[0] : vb[0], sht = a_long ? vb[0] : va[0];
      const uint32_t nl = a_long ? sa.len : sb.len, ns = a_long ? sb.len : sa.len;
      const uint32_t lo = lng & 0xFFFFu, hi = lng >> 16;
      bool m_lo = false, m_hi = false;
      for (uint32_t k = 0; k < ns; ++k) {
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)sht, (int)(k >> 1));
        const uint32_t sv = (k & 1u) ? d >> 16 : d & 0xFFFFu;
        m_lo |= lo == sv;
        m_hi |= hi == sv;
      }
      part += ((2u * (uint32_t)lane < nl && m_lo) ? 1u : 0u) + ((2u * (uint32_t)lane + 1u < nl && m_hi) ? 1u : 0u);
    } else {
      // ONE instance of the table + probe loops for both assignments of the roles: the shorter array's batch 0 is brought into va by an
      // IN-PLACE exchange (no register is added — role-named COPIES had cost 16 registers and 24 spilled ones in round 3, which is why
      // every type pair carried two inlined instances of its loops until round 5: half of the kernel's 102 KB of code)
      const bool a_tab = sa.len <= sb.len;  // wave-uniform
      if (!a_tab) batch_exchange(va, vb);
      part += arrays_table_probe(a_tab ? pa : pb, a_tab ? sa.len : sb.len, va, a_tab ? pb : pa, a_tab ? sb.len : sa.len, vb, lane, table, spart);
    }
  } else if ((ta == kTypeArray && tb == kTypeBitmap) || (ta == kTypeBitmap && tb == kTypeArray)) {
    const bool a_arr = ta == kTypeArray;  // wave-uniform; the array's batch 0 goes to va (the bitmap side has none)
    if (!a_arr) batch_exchange(va, vb);
    const uint8_t* parr = a_arr ? pa : pb;
    const uint8_t* pbm = a_arr ? pb : pa;
    const uint32_t larr = a_arr ? sa.len : sb.len;
    if ((sparse_paths & 1u) && larr <= kProbeArray) part += array_probe_global(va, larr, pbm, lane);
    else part += bitmap_table_probe(pbm, parr, larr, va, lane, table, spart);
  } else if ((sparse_paths & 2u) && ((ta == kTypeArray && tb == kTypeRun && sb.len <= kRunFillMax) || (ta == kTypeRun && tb == kTypeArray && sa.len <= kRunFillMax))) {
    // array x run: the run container becomes the table, the array probes it — its batch 0 in va, the array's in vb
    const bool a_run = ta == kTypeRun;  // wave-uniform
    if (!a_run) batch_exchange(va, vb);
    part += run_table_probe(a_run ? pa : pb, a_run ? sa.len : sb.len, va, a_run ? pb : pa, a_run ? sb.len : sa.len, vb, lane, table, mini, spart);
  } else {
    // a run on at least one side: both operands 1 KiB at a time out of the table, one clear
    uint32_t acc = 0;
    pair_stream(ta, pa, sa.len, tb, pb, sb.len, lane, table, mini, va, vb,
                [&acc](u64 a0, u64 a1, u64 b0, u64 b1) { acc += (uint32_t)__popcll(a0 & b0) + (uint32_t)__popcll(a1 & b1); });
    part += acc;
  }
}

// items K .. SPW - 1 of a wave, the next one's batch 0 in flight while item K is worked on (a template recursion:
// the item body is too large for the unroller, and a real loop would index the descriptor arrays dynamically)
template <int K, int SPW>
__device__ __forceinline__ void icount_items(const Slot (&sa)[SPW], const uint8_t* __restrict__ arenaA, const Slot (&sb)[SPW],
                                             const uint8_t* __restrict__ arenaB, int lane, u64* table, uint32_t* mini, uint32_t (&va)[kPairBatch],
                                             uint32_t (&vb)[kPairBatch], uint32_t sparse_paths, uint32_t& part, uint32_t& spart) {
  if constexpr (K < SPW) {
    uint32_t xa[kPairBatch], xb[kPairBatch];
    if constexpr (K + 1 < SPW) item_prefetch(sa[K + 1], arenaA, sb[K + 1], arenaB, lane, xa, xb, sparse_paths);
    icount_item(sa[K], arenaA, sb[K], arenaB, lane, table, mini, va, vb, sparse_paths, part, spart);
    if constexpr (K + 1 < SPW) icount_items<K + 1, SPW>(sa, arenaA, sb, arenaB, lane, table, mini, xa, xb, sparse_paths, part, spart);
  }
}

// WPB waves per block.  A block's LDS and wave slots are released when its LAST wave ends, and an item with a run
// lives several times as long as an array x array one: with four waves per block 84 % of the blocks of config 3's
// row pairs held a run item and every block lived as long as its slowest wave (skipping the array x array items —
// half of all items — shortened the kernel by 4 us out of 57).  One-wave blocks release each wave's table the
// moment it ends.
template <int SPW, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_icount2(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                                     const uint32_t* __restrict__ rowsA, const Slot* __restrict__ slotsB,
                                                     const uint8_t* __restrict__ arenaB, const uint32_t* __restrict__ rowsB, uint64_t n_pairs,
                                                     u64* __restrict__ out, uint32_t sparse_paths, const Slot* __restrict__ items,
                                                     uint32_t* __restrict__ wave_out) {
  __shared__ u64 lds[WPB][kWords];
  __shared__ uint32_t mini[WPB][2 * kMiniDwords];
#ifndef FBK_EXPERIMENTS
  sparse_paths &= 0xFFu;  // the ablation / cycle-stamp bits (8..23) exist in experiment builds only: every branch on them below folds away
#endif
  const u64 t0 = ((sparse_paths >> 16) & 7u) ? __builtin_readcyclecounter() : 0;
  constexpr int kWavesPerPair = kSlots / SPW;
  const int lane = threadIdx.x & 63;
  const int wv = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and the compiler knows it
  const uint64_t wid = (uint64_t)blockIdx.x * WPB + (uint64_t)wv;
  const uint64_t pair = wid / kWavesPerPair;
  if (pair >= n_pairs) return;
  const uint32_t slot0 = (uint32_t)(wid % kWavesPerPair) * SPW;
  Slot sa[SPW], sb[SPW];
  if (items) {
    // the plan's resolved item records (k_resolve_items): {A's descriptor, B's descriptor} per (pair, slot), in item
    // order — ONE scalar round trip instead of row index -> descriptor, and neighbouring waves share its lines
    const Slot* it = items + (pair * kSlots + slot0) * 2;
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      sa[k] = it[2 * k];
      sb[k] = it[2 * k + 1];
    }
  } else {
    const uint32_t ra = rowsA[pair], rb = rowsB[pair];  // both row indexes in flight together
    const Slot* da = slotsA + (uint64_t)ra * kSlots + slot0;
    const Slot* db = slotsB + (uint64_t)rb * kSlots + slot0;
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      sa[k] = da[k];
      sb[k] = db[k];
    }
  }
  u64* table = lds[wv];
  uint32_t part = 0, spart = 0;
  uint32_t va[kPairBatch], vb[kPairBatch];
  // option pair_stamp (timing experiment, WRONG results): the wave reports shader cycles instead of its count —
  // 1: launch -> descriptors in registers, 2: -> batch 0 of both operands landed, 3: -> decoded, 4: the whole wave
  const uint32_t stamp = (sparse_paths >> 16) & 7u;
  u64 t_mark = 0, t_prev = 0;
  if (stamp) {
    t_prev = t0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t_mark = __builtin_readcyclecounter();
    if (stamp == 1) spart = (uint32_t)(t_mark - t_prev);
    if (stamp != 4) t_prev = t_mark;
  }
  // (Round 6 tried an L2 PREFETCH here — one dword per 128-byte line of the payloads of the item 1536 / 4608 / 9216 waves ahead, so
  // that the wave that takes this wave's slot a generation later finds them in the L2 / Infinity Cache: 162.5 / 171.4 / 174.1 us
  // against 156.5-156.7, profiles/r06_pairs_prefetch_ahead_ab.txt.  More bytes in flight make this access pattern SLOWER.)
  if (!(sparse_paths & 0x400u)) {  // (0x400: timing experiment, descriptors only)
    item_prefetch(sa[0], arenaA, sb[0], arenaB, lane, va, vb, sparse_paths);
    if (stamp) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t_mark = __builtin_readcyclecounter();
      if (stamp == 2) spart = (uint32_t)(t_mark - t_prev);
      if (stamp != 4) t_prev = t_mark;
    }
    icount_items<0, SPW>(sa, arenaA, sb, arenaB, lane, table, mini[wv], va, vb, sparse_paths, part, spart);
    if (stamp >= 3) {
      const uint32_t keep = wave_reduce_add(part);  // (the decode has to finish before the stamp)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      t_mark = __builtin_readcyclecounter();
      spart = (uint32_t)(t_mark - t_prev) + (keep & 0u);
      part = 0;
    } else if (stamp) {
      part = 0;
    }
  } else {
    spart = slot_n(sa[0]) + slot_n(sb[SPW - 1]);
  }
  if (sparse_paths & 0x100u) return;  // (timing experiment: no output)
  const uint32_t c = wave_reduce_add(part) + spart;
  if (wave_out) {
    // one plain store per wave; k_sum_wave_counts adds a pair's waves up.  (The uint64 atomics of the 16 waves of a
    // pair onto out[pair] — from up to 8 XCDs, so executed at the memory side — and the memset they need in front cost
    // ~8 us of a 46 us launch.)
    if (lane == 0) wave_out[wid] = c;
  } else if (lane == 0 && c) {
    atomicAdd(&out[pair], (u64)c);
  }
}

// (Round 4 built a LEAN array x array count here — k_icount_aa: a 4 KiB table for half the value range used twice, both
// arrays in registers, 32 registers, 32 waves per CU, fed by a host pass that sorted a plan's items by class.  Parity-green and
// worth 0-3 % on config 3's row pairs (those items are half of the waves and a seventh of the time; profiles/r04_pairs_lean_ab.json):
// removed in round 5 with its option.)

// out[pair] = the sum of the pair's waves' counts (per = 16 / SPW of them, consecutive)
// (reset: a word this launch puts back to zero — k_icount3's chunk counter — or null)
__global__ void __launch_bounds__(256) k_sum_wave_counts(const uint32_t* __restrict__ wave_counts, uint32_t per, uint64_t n_pairs, u64* __restrict__ out,
                                                        uint32_t* __restrict__ reset) {
  if (reset && blockIdx.x == 0 && threadIdx.x == 0) *reset = 0u;
  const uint64_t pair = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (pair >= n_pairs) return;
  const uint32_t* w = wave_counts + pair * per;
  u64 acc = 0;
  for (uint32_t i = 0; i < per; ++i) acc += w[i];
  out[pair] = acc;
}

// item records of a plan: items[2 i] = A's descriptor, items[2 i + 1] = B's, i = pair * 16 + slot
__global__ void __launch_bounds__(256) k_resolve_items(const Slot* __restrict__ slotsA, const uint32_t* __restrict__ rowsA, const Slot* __restrict__ slotsB,
                                                      const uint32_t* __restrict__ rowsB, uint64_t n_pairs, Slot* __restrict__ items) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pairs * kSlots) return;
  const uint64_t pair = i >> 4, slot = i & 15;
  items[2 * i] = slotsA[(uint64_t)rowsA[pair] * kSlots + slot];
  items[2 * i + 1] = slotsB[(uint64_t)rowsB[pair] * kSlots + slot];
}

// out[pair] = the cardinality of output row `pair`: the sum of the n its 16 cells' descriptors carry.  The materialising
// kernels used to add every wave's count onto out[pair] with a uint64 atomic (16 waves of a pair, from up to 8 XCDs: executed
// at the memory side, plus the memset in front); the descriptors they write anyway hold the same numbers.
__global__ void __launch_bounds__(256) k_sum_slot_n(const Slot* __restrict__ outSlots, uint64_t n_pairs, u64* __restrict__ out) {
  const uint64_t pair = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (pair >= n_pairs) return;
  const Slot* s = outSlots + pair * kSlots;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < kSlots; ++k) acc += slot_n(s[k]);
  out[pair] = acc;
}

// (A PERSISTENT form — a grid as large as the device holds, every wave striding through the items as a four-stage
// software pipeline: row indexes of item u + 3 W, descriptors of u + 2 W, payload batch 0 of u + W in flight while item u is
// decoded — was built and measured this round: 64 us against 46 for the 2048 config-3 pairs (128 registers, static striding
// over items of very different cost).  Removed; the one-wave blocks above give the hardware's dispatcher that job.)

// ---- Intersect / Difference whose result is a subset of an ARRAY operand, by the count kernel's table + probe -------
//
// intersectArrayArray / intersectArrayBitmap / differenceArrayArray / differenceArrayBitmap (roaring.go:4632-4700,
// 5339-5420) walk the array and keep the values found (not found) in the other container.  The materialising kernel did
// that by decoding BOTH operands into 8 KiB fragments whatever their size (frag_load_pair: a table clear, a scatter and a
// read-back per sparse operand), then popcount, run count, and an encode pass through the LDS — ~310 us for config 3's
// 8192 sparse row pairs where counting the same intersections takes 176.  Here the other operand becomes the wave's
// table exactly as in k_icount2 (shorter array scattered into the cleared table / bitmap copied in), the array PROBES it,
// and the survivors are written straight into the output cell as the sorted array they already are: positions from two
// ballots per dword row, the run count of the result (for Container.optimize()'s rule) from each survivor's predecessor.
struct ProbeEmit {                 // (everything wave-uniform: scalar registers)
  uint32_t before = 0;             // survivors written so far
  uint32_t runs = 0;               // runs the survivors so far start
  uint32_t last_val = 0x7FFFFFFFu; // the array element in front of lane 0's of the coming dword row, and whether it
  uint32_t last_kept = 0;          // survived
  bool nostore = false;            // (timing experiments only)
};

__device__ __forceinline__ uint32_t mbcnt64(u64 m, uint32_t add) {  // add + the bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, add));
}

// one batch of the probing array: KEEP = 1 keeps the values set in the table, 0 those that are not; MAP: the table holds a
// run container as boundary masks, its full dwords are in the map at mb.
// The emission is what this kernel costs beyond the count of the same pairs (scripts/bench_pairs.py with pair_ablate,
// profiles/r05_setop_ablate.json: storing nothing changes nothing, the ~40 vector instructions per dword row of the first
// version were the 90 us), so everything that is the same for the 64 lanes is kept in SCALAR registers: the survivor masks are
// the comparisons' own lane masks, "which lanes hold a value" is two scalar bit-field masks per row, positions are two
// v_mbcnt pairs, the run count is mask arithmetic (a survivor starts a run unless its predecessor in the array survived and
// is its value - 1: two adjacency masks per row, the predecessor of a lane's low value through one wave-shift DPP), and a
// row without survivors ends after its two probes.
template <int KEEP, bool MAP>
__device__ __forceinline__ void array_probe_emit_batch(uint32_t tb, uint32_t mb, uint32_t len, uint32_t base, int lane, const uint32_t (&v)[kPairBatch],
                                                       uint16_t* __restrict__ o16, ProbeEmit& e) {
  // every probe of the batch first: 16 (MAP: 32) LDS reads in flight together.  (With the reads inside the row loop every row
  // waited for its own LDS round trip — ~30 dependent round trips per item, which is what the emission cost once its arithmetic
  // was out of the way: profiles/r05_setop_ablate_scalar_masks.json.)  Rows past the array's end hold zeros: they read word 0.
  uint32_t t_lo[kPairBatch], t_hi[kPairBatch];
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) {
    t_lo[k] = table_bit_lo(tb, v[k]);
    t_hi[k] = table_bit_hi(tb, v[k]);
    if (MAP) {
      t_lo[k] |= map_bit_lo(mb, v[k]);
      t_hi[k] |= map_bit_hi(mb, v[k]);
    }
  }
  // (a scheduling barrier here — the count kernel's probe needed one — changes nothing: 229.5 / 230.7 against 229.4 / 230.4 us, round 6)
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) {
    const uint32_t first = (base + (uint32_t)k * kWave) * 2u;  // index of lane 0's low value
    if (first >= len) break;  // (wave-uniform: no value in this dword row or after it)
    const uint32_t rem = len - first;
    const uint32_t n_lo = min((rem + 1u) >> 1, (uint32_t)kWave), n_hi = min(rem >> 1, (uint32_t)kWave);  // lanes whose low / high value exists
    const u64 have_lo = n_lo >= (uint32_t)kWave ? ~0ull : ((1ull << n_lo) - 1ull), have_hi = n_hi >= (uint32_t)kWave ? ~0ull : ((1ull << n_hi) - 1ull);
    const bool c_lo = t_lo[k] == (uint32_t)KEEP, c_hi = t_hi[k] == (uint32_t)KEEP;
    const u64 m_lo = __ballot(c_lo) & have_lo, m_hi = __ballot(c_hi) & have_hi;
    if ((m_lo | m_hi) == 0ull) {  // nothing of this row survives
      e.last_kept = 0;
      continue;
    }
    const uint32_t lo = v[k] & 0xFFFFu, hi = v[k] >> 16;
    const uint32_t pos = mbcnt64(m_hi, mbcnt64(m_lo, e.before));
    const bool k_lo = c_lo && (uint32_t)lane < n_lo, k_hi = c_hi && (uint32_t)lane < n_hi;
    if (!e.nostore) {
      if (k_lo) o16[pos] = (uint16_t)lo;
      if (k_hi) o16[pos + (k_lo ? 1u : 0u)] = (uint16_t)hi;
    }
    // the element in front of this lane's low value is the previous lane's high value (lane 0: the previous row's last)
    const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp((int)e.last_val, (int)hi, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const u64 adj_lo = __ballot(pv + 1u == lo), adj_hi = __ballot(lo + 1u == hi);
    const u64 prev_kept = (m_hi << 1) | (u64)e.last_kept;
    e.runs += (uint32_t)__popcll(m_lo & ~(adj_lo & prev_kept)) + (uint32_t)__popcll(m_hi & ~(adj_hi & m_lo));
    e.before += (uint32_t)__popcll(m_lo) + (uint32_t)__popcll(m_hi);
    e.last_val = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    e.last_kept = (uint32_t)(m_hi >> 63);
  }
}

// the whole probing array of <= 4095 values (batch 0 in v0, batches 1..3 in the tail requested before the table was built)
template <int KEEP, bool MAP = false>
__device__ __forceinline__ void array_probe_emit_all(const uint8_t* __restrict__ /*p*/, uint32_t len, int lane, uint32_t tb, const uint32_t (&v0)[kPairBatch],
                                                     const ProbeTail& t, uint16_t* __restrict__ o16, uint32_t& n_out, uint32_t& runs_out, uint32_t mb = 0, bool nostore = false) {
  const uint32_t n_units = (len + 1u) >> 1;
  constexpr uint32_t B = kPairBatch * kWave;
  ProbeEmit e;
  e.nostore = nostore;
  array_probe_emit_batch<KEEP, MAP>(tb, mb, len, 0, lane, v0, o16, e);
  if (n_units > B) array_probe_emit_batch<KEEP, MAP>(tb, mb, len, B, lane, t.v1, o16, e);
  if (n_units > 2 * B) array_probe_emit_batch<KEEP, MAP>(tb, mb, len, 2 * B, lane, t.v2, o16, e);
  if (n_units > 3 * B) array_probe_emit_batch<KEEP, MAP>(tb, mb, len, 3 * B, lane, t.v3, o16, e);
  // (len <= 4095 — the caller's condition: the survivors must fit the cell — is at most four batches)
  n_out = e.before;
  runs_out = e.runs;
}

// probing array (pp, lp <= 4095 values, batch 0 in vp) against an ARRAY operand (pt, lt, batch 0 in vt)
template <int KEEP>
__device__ __forceinline__ void array_vs_array_emit(const uint8_t* __restrict__ pt, uint32_t lt, uint32_t (&vt)[kPairBatch], const uint8_t* __restrict__ pp,
                                                    uint32_t lp, const uint32_t (&vp)[kPairBatch], int lane, u64* table, uint16_t* __restrict__ o16,
                                                    uint32_t& n_out, uint32_t& runs_out, bool nostore = false) {
  ProbeTail tail;
  probe_tail_load(pp, (lp + 1u) >> 1, lane, tail);
  const uint32_t tbase = lds_table_base(table);
  lds_zero(table, lane);
  wave_lds_sync();
  sparse_xor_all(kTypeArray, pt, lt, lane, tbase, vt);
  wave_lds_sync();
  array_probe_emit_all<KEEP>(pp, lp, lane, tbase, vp, tail, o16, n_out, runs_out, 0, nostore);
  wave_lds_sync();
}

// probing array against a BITMAP operand (copied into the table, no clear)
template <int KEEP>
__device__ __forceinline__ void array_vs_bitmap_emit(const uint8_t* __restrict__ pbm, const uint8_t* __restrict__ pp, uint32_t lp,
                                                     const uint32_t (&vp)[kPairBatch], int lane, u64* table, uint16_t* __restrict__ o16, uint32_t& n_out,
                                                     uint32_t& runs_out, bool nostore = false) {
  u64 wb[kWordsPerLane];
  frag_load_bitmap(pbm, lane, wb);
  ProbeTail tail;
  probe_tail_load(pp, (lp + 1u) >> 1, lane, tail);
  ulonglong2* q = reinterpret_cast<ulonglong2*>(table);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 x;
    x.x = wb[2 * j];
    x.y = wb[2 * j + 1];
    q[j * kWave + lane] = x;  // fragment layout -> natural word order in the table
  }
  wave_lds_sync();
  array_probe_emit_all<KEEP>(pp, lp, lane, lds_table_base(table), vp, tail, o16, n_out, runs_out, 0, nostore);
  wave_lds_sync();
}

// probing array against a RUN operand of <= kRunFillMax intervals (batch 0 in vr): the runs as boundary masks in the
// cleared table + the map of full dwords (run_fill_batch), the array probes both (intersectArrayRun / differenceArrayRun,
// roaring.go:4702-4740, 5421-5470, walk the two lists; here neither list is walked)
template <int KEEP>
__device__ __forceinline__ void array_vs_run_emit(const uint8_t* __restrict__ pr, uint32_t lr, uint32_t (&vr)[kPairBatch], const uint8_t* __restrict__ pp,
                                                  uint32_t lp, const uint32_t (&vp)[kPairBatch], int lane, u64* table, uint32_t* mini,
                                                  uint16_t* __restrict__ o16, uint32_t& n_out, uint32_t& runs_out, bool nostore = false) {
  ProbeTail tail;
  probe_tail_load(pp, (lp + 1u) >> 1, lane, tail);
  uint32_t tbase, mbase;
  run_table_build(pr, lr, vr, lane, table, mini, tbase, mbase);
  array_probe_emit_all<KEEP, true>(pp, lp, lane, tbase, vp, tail, o16, n_out, runs_out, mbase, nostore);
  wave_lds_sync();
}

// Materialising A <op> B, one wave per (pair, slot): k_setop's outputs and right-sized array paths
// (fbk_kernels.hip.h) behind the pair loader above.
// LEAN (round 6): the instance for calls whose results are mostly written by the probe paths (Intersect / Difference with optimize()).
// The kernel's time follows its occupancy (profiles/r06_setop2_occupancy.txt: 16 / 14 / 12 waves per CU = 231 / 269 / 318 us) and 102
// registers hold it at 4 waves per SIMD where the LDS allows 4.5: this instance is compiled for 5 (96 registers), its general path
// — the minority of the items there — decodes one operand after the other (frag_load, the round-2 loader).
template <int OP, int WPB, bool LEAN = false>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(LEAN ? 5 : 4))) k_setop2(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                                    const uint32_t* __restrict__ rowsA, const Slot* __restrict__ slotsB,
                                                    const uint8_t* __restrict__ arenaB, const uint32_t* __restrict__ rowsB, uint64_t n_pairs,
                                                    uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                                    uint32_t direct, const Slot* __restrict__ items) {
  __shared__ u64 lds[WPB][kWords];
  __shared__ uint32_t mini[WPB][2 * kMiniDwords];
  const int lane = threadIdx.x & 63;
  const int wv = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: descriptors become scalar loads (see k_icount2)
  const bool probe = (direct & 0x100u) != 0;  // Intersect / Difference of an array by table + probe
  // timing experiments (option pair_ablate in the experiments build, WRONG results): 8 items with a run are skipped, 16 array x array
  // items, 32 bitmap x array items, 64 the probe paths keep their arithmetic but store no survivor, 128 the general path (both
  // operands decoded into fragments) is skipped, 512 it encodes and writes nothing, 1024 it does not count runs, 2048 probe results that
  // optimize() would store as runs are left as arrays, 256 the longer array probes in Intersect (the first version's roles)
#ifdef FBK_EXPERIMENTS
  const uint32_t abl = direct >> 16;
#else
  constexpr uint32_t abl = 0;
#endif
  direct &= 0xFFu;
  const uint64_t wslot = (uint64_t)blockIdx.x * WPB + (uint64_t)wv;
  const uint64_t pair = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (pair >= n_pairs) return;
  Slot sa, sb;
  if (items) {  // the plan's resolved item records (k_resolve_items): one scalar round trip instead of row index -> descriptor
    sa = items[2 * wslot];
    sb = items[2 * wslot + 1];
  } else {
    const uint32_t ra = rowsA[pair], rb = rowsB[pair];
    sa = slotsA[(uint64_t)ra * kSlots + slot];
    sb = slotsB[(uint64_t)rb * kSlots + slot];
  }
  const uint32_t na = slot_n(sa), nb = slot_n(sb);
  Slot so;
  so.off = wslot * 8192ull;
  so.len = kWords;
  so.tn = 0;
  bool empty;
  if (OP == 0) empty = (na == 0 || nb == 0);
  else if (OP == 3) empty = (na == 0 || nb == 65536u);
  else empty = (na == 0 && nb == 0);
  if (empty) {
    if (lane == 0) {
      outSlots[wslot] = so;
      if (outRuns) outRuns[wslot] = 0;
    }
    return;
  }
  if (abl) {
    const uint32_t xa = slot_type(sa), xb = slot_type(sb);
    const bool skip = ((abl & 8u) && (xa == kTypeRun || xb == kTypeRun)) || ((abl & 16u) && xa == kTypeArray && xb == kTypeArray) ||
                      ((abl & 32u) && ((xa == kTypeArray && xb == kTypeBitmap) || (xa == kTypeBitmap && xb == kTypeArray)));
    if (skip) {  // (wave-uniform)
      if (lane == 0) {
        outSlots[wslot] = so;
        if (outRuns) outRuns[wslot] = 0;
      }
      return;
    }
  }
  if (OP == 0 && direct && (outRuns || direct == 2u)) {
    // intersection with a small array, written as an array (see k_setop)
    const uint32_t ta = slot_type(sa), tb = slot_type(sb);
    u64 mm = 0;
    uint32_t v = 0;
    bool handled = true, al;
    if (ta == kTypeArray && tb == kTypeArray && sa.len <= kSmallArray && sb.len <= kSmallArray)
      mm = small_arrays_match(arenaA + sa.off, sa.len, arenaB + sb.off, sb.len, lane, v, al);
    else if (ta == kTypeArray && tb == kTypeBitmap && sa.len <= kSmallArray)
      mm = array_probe_bitmap(arenaA + sa.off, sa.len, 0, arenaB + sb.off, lane, v);
    else if (tb == kTypeArray && ta == kTypeBitmap && sb.len <= kSmallArray)
      mm = array_probe_bitmap(arenaB + sb.off, sb.len, 0, arenaA + sa.off, lane, v);
    else
      handled = false;
    if (handled) {  // wave-uniform
      const u64 below = lane ? (mm & (~0ull >> (64 - lane))) : 0ull;
      const bool mine = (mm >> lane) & 1ull;
      const int prev = below ? 63 - __builtin_clzll(below) : 0;
      const uint32_t vprev = (uint32_t)__shfl((int)v, prev, kWave);
      const uint32_t r = (uint32_t)__popcll(__ballot(mine && (below == 0 || vprev + 1u != v)));
      const uint32_t c = (uint32_t)__popcll(mm);
      if (!(direct == 2u && c != 0 && r <= c / 2u)) {  // (survivors optimize() would store as runs take the general path)
        if (mine) reinterpret_cast<uint16_t*>(arenaO + so.off)[__popcll(below)] = (uint16_t)v;
        if (lane == 0) {
          so.len = c;
          so.tn = make_tn(c ? kTypeArray : kTypeNil, c);
          outSlots[wslot] = so;
          if (outRuns) outRuns[wslot] = r;
        }
        return;
      }
    }
  }
  if ((OP == 0 || OP == 3) && direct == 2u && probe) {
    // the result is a subset of an array operand: table + probe, survivors written as the array they are (see above).
    // (Arrays beyond 4095 values — oversized intermediates, roaring.go:5054 — could overflow the cell: general path.)
    const uint32_t ta = slot_type(sa), tb = slot_type(sb);
    const uint8_t* pa = arenaA + sa.off;
    const uint8_t* pb = arenaB + sb.off;
    uint16_t* o16 = reinterpret_cast<uint16_t*>(arenaO + so.off);
    const bool nostore = (abl & 64u) != 0;  // (experiment: the probe paths keep their arithmetic and store no survivor)
    uint32_t c = 0, r = 0;
    bool done = false;
    // who probes: the array the result is a subset of (Difference: A; Intersect: the SHORTER array, the longer one is
    // the table; against a bitmap or a run: the array).  Wave-uniform; batch 0 of both payloads is requested after the roles
    // are known, so each type pair has ONE instance of the probe loop.
    const bool a_arr = ta == kTypeArray, b_arr = tb == kTypeArray;
    // (Intersect of two arrays: the SHORTER one probes — a probed dword row costs ~3 x a scattered one once its survivors are
    // emitted, the reverse of the count kernel's choice; experiments: pair_ablate bit 256 restores the longer array)
    const bool a_probes = a_arr && (OP == 3 || !b_arr || ((abl & 256u) ? sa.len > sb.len : sa.len <= sb.len));
    const bool b_probes = OP == 0 && b_arr && !a_probes;
    if (a_probes || b_probes) {
      const uint8_t* pp = a_probes ? pa : pb;   // the probing array
      const uint8_t* pt = a_probes ? pb : pa;   // the operand that becomes the table
      const uint32_t lp = a_probes ? sa.len : sb.len, lt = a_probes ? sb.len : sa.len;
      const uint32_t tt = a_probes ? tb : ta;
      if (lp <= 4095u) {
        uint32_t vp[kPairBatch], vt[kPairBatch];
        if (tt == kTypeArray) {
          sparse_load(pp, (lp + 1u) >> 1, 0, lane, vp);
          sparse_load(pt, (lt + 1u) >> 1, 0, lane, vt);
          array_vs_array_emit<OP == 0>(pt, lt, vt, pp, lp, vp, lane, lds[wv], o16, c, r, nostore);
          done = true;
        } else if (tt == kTypeBitmap) {
          sparse_load(pp, (lp + 1u) >> 1, 0, lane, vp);
          array_vs_bitmap_emit<OP == 0>(pt, pp, lp, vp, lane, lds[wv], o16, c, r, nostore);
          done = true;
        } else if (tt == kTypeRun && lt <= kRunFillMax) {
          sparse_load(pp, (lp + 1u) >> 1, 0, lane, vp);
          sparse_load(pt, lt, 0, lane, vt);
          array_vs_run_emit<OP == 0>(pt, lt, vt, pp, lp, vp, lane, lds[wv], mini[wv], o16, c, r, nostore);
          done = true;
        }
      }
    }
    if (done && (!(c != 0 && r <= c / 2u) || (abl & 2048u))) {  // (survivors optimize() would store as runs take the general path below and overwrite the cell; 2048: experiment, they do not)
      if (lane == 0) {
        so.len = c;
        so.tn = make_tn(c ? kTypeArray : kTypeNil, c);
        outSlots[wslot] = so;
        if (outRuns) outRuns[wslot] = r;
      }
      return;
    }
  }
  if (abl & 128u) {  // (experiment: no item takes the general path)
    if (lane == 0) {
      outSlots[wslot] = so;
      if (outRuns) outRuns[wslot] = 0;
    }
    return;
  }
  u64 wa[kWordsPerLane], wb[kWordsPerLane];
  if (LEAN) {  // one operand after the other through the table (the round-2 loader): the first one's temporaries are dead when the second starts
    if (na) frag_load(sa, arenaA, lane, lds[wv], wa);
    else frag_zero(wa);
    if (nb) frag_load(sb, arenaB, lane, lds[wv], wb);
    else frag_zero(wb);
  } else {
    frag_load_pair(sa, arenaA, sb, arenaB, lane, lds[wv], mini[wv], wa, wb);
  }
#pragma unroll
  for (int i = 0; i < kWordsPerLane; ++i) {
    wa[i] = apply_op<OP>(wa[i], wb[i]);
    if (OP == 2) asm volatile("" : "+v"(wa[i]));  // (as in k_setop: keeps the XOR's result instead of re-deriving it; 141 -> ~100 registers)
  }
  uint32_t c = wave_reduce_add(frag_popcount(wa));
  uint32_t r = 0;
  if ((outRuns || direct == 2u) && !(abl & 1024u)) r = wave_reduce_add(frag_count_runs(wa, lane));  // (1024: experiment, no run count)
  if (direct == 2u && c != 0) {  // Container.optimize() applied here (see k_setop / frag_store_encoded)
    uint32_t t_out = kTypeBitmap, l_out = kWords;
    if (!(abl & 512u)) frag_store_encoded(wa, c, r, lane, lds[wv], arenaO + so.off, t_out, l_out);  // (512: experiment, nothing is encoded or written)
    if (lane == 0) {
      so.len = l_out;
      so.tn = make_tn(t_out, c);
      outSlots[wslot] = so;
      if (outRuns) outRuns[wslot] = r;
    }
    return;
  }
  bool as_array = false;
  if (direct && outRuns && c != 0 && c <= kDirectArrayMax) {
    as_array = true;
    uint16_t* o16 = reinterpret_cast<uint16_t*>(arenaO + so.off);
    uint32_t before = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t mine = (uint32_t)__popcll(wa[2 * j]) + (uint32_t)__popcll(wa[2 * j + 1]);
      const uint32_t incl = wave_incl_scan(mine);
      uint32_t pos = before + incl - mine;
      const uint32_t base = (128u * j + 2u * (uint32_t)lane) * 64u;
      for (u64 x = wa[2 * j]; x; x &= x - 1) o16[pos++] = (uint16_t)(base + (uint32_t)__builtin_ctzll(x));
      for (u64 x = wa[2 * j + 1]; x; x &= x - 1) o16[pos++] = (uint16_t)(base + 64u + (uint32_t)__builtin_ctzll(x));
      before += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  } else if (direct != 2u) {  // (direct == 2 gets here only with an empty result: nil, nothing to write)
    frag_store_bitmap(arenaO + so.off, lane, wa);
  }
  if (lane == 0) {
    if (as_array) so.len = c;
    so.tn = make_tn(c ? (as_array ? kTypeArray : kTypeBitmap) : kTypeNil, c);
    outSlots[wslot] = so;
    if (outRuns) outRuns[wslot] = r;
  }
}

}  // namespace fbk
