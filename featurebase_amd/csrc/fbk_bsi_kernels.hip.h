// fbk_bsi_kernels.hip.h — bit-sliced-integer kernels: Sum, Range (plane-program interpreter),
// Min/Max.  HBM-bound streaming of bit planes; no MFMA.
//
// Sum and Range use the BLOCK layout ("bfrag"): one 256-thread block owns one (shard, slot)
// cell and thread t holds the 16-byte chunks t and t+256 of every 8 KiB container of that
// slot, i.e. container words 2t, 2t+1, 512+2t, 513+2t.  Per-thread state is then only 4 x u64
// per fragment register, so (a) a 96-shard BSI field becomes 1536 blocks = 6144 wavefronts
// that all fit on the chip at once (24 waves per CU) and (b) the loads of several bit planes
// can be issued back to back before the first one is consumed (U planes in flight per
// thread).  The first version of these kernels (one wavefront per cell, 16 x u64 per lane,
// one plane in flight) measured 1.9 TB/s (Range) and 4.5 TB/s (Sum) on 96 shards x 66 rows:
// latency-bound, 1.5 wavefronts per SIMD.
#pragma once
#include "fbk_kernels.hip.h"

namespace fbk {

constexpr int kBW = 4;  // u64 words per thread in the block layout

__device__ __forceinline__ void bfrag_zero(u64 (&w)[kBW]) {
#pragma unroll
  for (int i = 0; i < kBW; ++i) w[i] = 0;
}

__device__ __forceinline__ void bfrag_load_bitmap(const uint8_t* __restrict__ p, int t, u64 (&w)[kBW]) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
  const ulonglong2 v0 = ld_stream(&q[t]), v1 = ld_stream(&q[256 + t]);
  w[0] = v0.x;
  w[1] = v0.y;
  w[2] = v1.x;
  w[3] = v1.y;
}

__device__ __forceinline__ void bfrag_store_bitmap(uint8_t* __restrict__ p, int t, const u64 (&w)[kBW]) {
  ulonglong2* q = reinterpret_cast<ulonglong2*>(p);
  ulonglong2 v0, v1;
  v0.x = w[0];
  v0.y = w[1];
  v1.x = w[2];
  v1.y = w[3];
  st_stream(&q[t], v0);
  st_stream(&q[256 + t], v1);
}

// true if the container can be fetched without barriers (bitmap, nil or empty)
__device__ __forceinline__ bool bfrag_is_fast(const Slot& s) {
  return slot_n(s) == 0 || slot_type(s) == kTypeBitmap || slot_type(s) == kTypeNil;
}

// Bitmap / nil / empty container: two streaming 16-byte loads per thread, no synchronisation —
// the caller may issue several of these before touching the first result.
__device__ __forceinline__ void bfrag_load_fast(const Slot& s, const uint8_t* __restrict__ arena, int t, u64 (&w)[kBW]) {
  if (slot_n(s) == 0 || slot_type(s) != kTypeBitmap) bfrag_zero(w);
  else bfrag_load_bitmap(arena + s.off, t, w);
}

// Any container, block-uniform (every thread of the block passes the same descriptor).
// Array / run containers (rare among bit planes: only the sparse high planes) are decoded by
// wavefront 0 with the wave-level decoder into the 8 KiB LDS scratch and then redistributed;
// that path contains block barriers, so callers keep it out of their pipelined loops.
__device__ __forceinline__ void bfrag_load(const Slot& s, const uint8_t* __restrict__ arena, int t, u64* scratch,
                                           u64 (&w)[kBW]) {
  if (bfrag_is_fast(s)) {
    bfrag_load_fast(s, arena, t, w);
    return;
  }
  if (t < kWave) {
    u64 f[kWordsPerLane];
    frag_load(s, arena, t, scratch, f);
    ulonglong2* q2 = reinterpret_cast<ulonglong2*>(scratch);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ulonglong2 v;
      v.x = f[2 * j];
      v.y = f[2 * j + 1];
      q2[j * kWave + t] = v;  // fragment register j of lane l is chunk 64*j + l: linear layout
    }
  }
  __syncthreads();
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(scratch);
  const ulonglong2 v0 = q[t], v1 = q[256 + t];
  w[0] = v0.x;
  w[1] = v0.y;
  w[2] = v1.x;
  w[3] = v1.y;
  __syncthreads();  // scratch may be reused
}

__device__ __forceinline__ uint32_t bfrag_popcount(const u64 (&w)[kBW]) {
  return __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
}

__device__ __forceinline__ u64 wave_reduce_add_u64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// Sum of one u64 per thread over a 256-thread block; `part` = 4 u64 of LDS.  Result valid in
// every thread.  Contains two barriers.
__device__ __forceinline__ u64 block_reduce_add_u64(u64 v, u64* part) {
  v = wave_reduce_add_u64(v);
  __syncthreads();  // `part` may still be read from a previous reduction
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  return part[0] + part[1] + part[2] + part[3];
}

// The descriptors of one slot of a BSI fragment's rows (exists, sign, planes: at most 66) staged in LDS by one
// wavefront, one parallel round of loads: a plane's descriptor is then an LDS read instead of a global load that the
// plane's own load has to wait for (descriptor -> payload is a dependent chain; it halves the effective prefetch depth).
constexpr int kBsiDescCap = 128;
__device__ __forceinline__ void bsi_stage_descs(const Slot* __restrict__ slots, uint64_t r0, uint32_t slot, uint32_t n_rows, int lane, Slot* tab) {
  for (uint32_t r = (uint32_t)lane; r < n_rows && r < (uint32_t)kBsiDescCap; r += kWave) tab[r] = slots[(r0 + r) * kSlots + slot];
  wave_lds_sync();
}
__device__ __forceinline__ Slot bsi_desc(const Slot* __restrict__ slots, uint64_t r0, uint32_t slot, uint32_t r, const Slot* tab) {
  return r < (uint32_t)kBsiDescCap ? tab[r] : slots[(r0 + r) * kSlots + slot];
}

// ---- BSI Sum ---------------------------------------------------------------------------------
// positive = filter ∩ exists \ sign, negative = filter ∩ exists ∩ sign stay in registers while
// the bit planes stream past once:
//   psum += |positive ∩ plane_i| << i ; nsum += |negative ∩ plane_i| << i   (uint64 wrap-around,
// roaring/filter.go:1157-1160).  Rows of the BSI fragment of shard s are base[s] + {0: exists,
// 1: sign, 2+i: bit i} (fragment.go:62-65).  out3[shard] = {psum, nsum, count}.
// One WAVEFRONT per (shard, slot) (the round-1 form, a 256-thread block per (shard, slot) that loaded 8 planes and waited
// for all of them, was removed in round 3: 141 us against 130):
// the planes of a slot do not depend on each other, so a wavefront keeps positive / negative (16 words per lane each)
// in registers and streams the planes with kAhead of them in flight — 1536 independent streams for 100 M columns,
// no LDS staging for bitmap planes, no barrier, and ONE wave reduction at the very end: a lane adds its own
// (popcount << i) terms in uint64 (wrap-around is linear, so the order of additions does not matter).
__global__ void __launch_bounds__(64) k_bsi_sum_slot(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                    const uint32_t* __restrict__ base, uint32_t n_shards, uint32_t bit_depth,
                                                    const Slot* __restrict__ fslots, const uint8_t* __restrict__ farena,
                                                    const uint32_t* __restrict__ frows, u64* __restrict__ out3) {
  __shared__ u64 lds[kWords];
  __shared__ Slot tab[kBsiDescCap];
  const int lane = threadIdx.x;
  const uint64_t shard = blockIdx.x >> 4;
  const uint32_t slot = blockIdx.x & 15;
  if (shard >= n_shards) return;
  const uint64_t r0 = base[shard];
  bsi_stage_descs(slots, r0, slot, bit_depth + 2, lane, tab);
  const Slot se = tab[0];
  if (slot_n(se) == 0) return;  // no existence bits: positive stays nil (filter.go:1135)
  u64 pos[kWordsPerLane], neg[kWordsPerLane];
  constexpr int kAhead = 3;
  u64 T[kAhead][kWordsPerLane];
  frag_load(se, arena, lane, lds, pos);
  if (fslots) {
    const Slot sf = fslots[(uint64_t)frows[shard] * kSlots + slot];
    if (slot_n(sf) == 0) return;  // ConsiderKey rejects: no filter container here (filter.go:1112)
    frag_load(sf, farena, lane, lds, T[0]);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) pos[q] &= T[0][q];
  }
  const uint32_t cnt = wave_reduce_add(frag_popcount(pos));
  if (cnt == 0) return;  // (wave-uniform) nothing considered in this slot: every term below is zero
  auto load_plane = [&](uint32_t i, u64 (&w)[kWordsPerLane]) {
    const Slot sp = bsi_desc(slots, r0, slot, 2 + i, tab);
    if (slot_n(sp) == 0) frag_zero(w);
    else frag_load(sp, arena, lane, lds, w);
  };
  {
    const Slot ss = tab[1];
    if (slot_n(ss) == 0) frag_zero(T[0]);  // nil sign row => zeros
    else frag_load(ss, arena, lane, lds, T[0]);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) {
      neg[q] = pos[q] & T[0][q];
      pos[q] &= ~T[0][q];
    }
  }
  u64 psum = 0, nsum = 0;
#pragma unroll
  for (int u = 0; u < kAhead; ++u)
    if ((uint32_t)u < bit_depth) load_plane((uint32_t)u, T[u]);
  for (uint32_t i0 = 0; i0 < bit_depth; i0 += kAhead) {
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const uint32_t i = i0 + (uint32_t)u;
      if (i < bit_depth) {  // (wave-uniform)
        uint32_t pc = 0, nc = 0;
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) {
          pc += __popcll(pos[q] & T[u][q]);
          nc += __popcll(neg[q] & T[u][q]);
        }
        psum += (u64)pc << i;
        nsum += (u64)nc << i;
        if (i + kAhead < bit_depth) load_plane(i + kAhead, T[u]);  // this register set is free again
      }
    }
  }
  psum = wave_reduce_add_u64(psum);
  nsum = wave_reduce_add_u64(nsum);
  if (lane == 0) {
    if (psum) atomicAdd(&out3[shard * 3 + 0], psum);
    if (nsum) atomicAdd(&out3[shard * 3 + 1], nsum);
    atomicAdd(&out3[shard * 3 + 2], (u64)cnt);
  }
}

// ---- BSI Range: plane-program interpreter -----------------------------------------------------
// The host walks the reference's control flow (rangeEQ/LT/GT/Between, fragment.go:963-1303)
// ONCE per query and emits a short straight-line program over three fragment registers
// X (remaining / result), M (matched), S (saved); every (shard, slot) block then runs the
// same program on its own containers, each bit plane read at most once per pass.  The loads
// of U consecutive instructions are issued together (they do not depend on X/M/S), the
// register updates are then applied in program order.
enum BsiOp : uint32_t {
  kLoadX = 0,   // X = row[r]
  kAndX = 1,    // X &= row[r]            (Row.Intersect)
  kAndnX = 2,   // X &= ~row[r]           (Row.Difference)
  kMorXA = 3,   // M |= X & row[r]        (matched = matched.Union(remaining.Intersect(row)))
  kMorXAn = 4,  // M |= X & ~row[r]       (matched = matched.Union(remaining.Difference(row)))
  kMZero = 5,   // M = 0                  (NewRow())
  kXFromM = 6,  // X = M
  kZeroX = 7,   // X = 0
  kSaveX = 8,   // S = X
  kOrXS = 9,    // X |= S
  kAndnXS = 10, // X &= ~S
  kNop = 255
};

template <int NW>
__device__ __forceinline__ void bsi_apply(uint32_t op, u64 (&X)[NW], u64 (&M)[NW], u64 (&S)[NW], const u64 (&T)[NW]) {
  switch (op) {
    case kLoadX:
#pragma unroll
      for (int q = 0; q < NW; ++q) X[q] = T[q];
      break;
    case kAndX:
#pragma unroll
      for (int q = 0; q < NW; ++q) X[q] &= T[q];
      break;
    case kAndnX:
#pragma unroll
      for (int q = 0; q < NW; ++q) X[q] &= ~T[q];
      break;
    case kMorXA:
#pragma unroll
      for (int q = 0; q < NW; ++q) M[q] |= X[q] & T[q];
      break;
    case kMorXAn:
#pragma unroll
      for (int q = 0; q < NW; ++q) M[q] |= X[q] & ~T[q];
      break;
    case kMZero:
#pragma unroll
      for (int q = 0; q < NW; ++q) M[q] = 0;
      break;
    case kXFromM:
#pragma unroll
      for (int q = 0; q < NW; ++q) X[q] = M[q];
      break;
    case kZeroX:
#pragma unroll
      for (int q = 0; q < NW; ++q) X[q] = 0;
      break;
    case kSaveX:
#pragma unroll
      for (int q = 0; q < NW; ++q) S[q] = X[q];
      break;
    case kOrXS:
#pragma unroll
      for (int q = 0; q < NW; ++q) X[q] |= S[q];
      break;
    case kAndnXS:
#pragma unroll
      for (int q = 0; q < NW; ++q) X[q] &= ~S[q];
      break;
    default: break;
  }
}

// the instructions that do not touch S (k_bsi_range_slot keeps S in LDS)
template <int NW>
__device__ __forceinline__ void bsi_apply2(uint32_t op, u64 (&X)[NW], u64 (&M)[NW], const u64 (&T)[NW]) {
  if (op <= kAndnX) {
    const u64 inv = op == kAndnX ? ~0ull : 0ull, keep = op == kLoadX ? ~0ull : 0ull;
#pragma unroll
    for (int q = 0; q < NW; ++q) X[q] = (X[q] | keep) & (T[q] ^ inv);
  } else if (op <= kMorXAn) {
    const u64 inv = op == kMorXAn ? ~0ull : 0ull;
#pragma unroll
    for (int q = 0; q < NW; ++q) M[q] |= X[q] & (T[q] ^ inv);
  } else if (op == kMZero) {
#pragma unroll
    for (int q = 0; q < NW; ++q) M[q] = 0;
  } else if (op == kXFromM) {
#pragma unroll
    for (int q = 0; q < NW; ++q) X[q] = M[q];
  } else if (op == kZeroX) {
#pragma unroll
    for (int q = 0; q < NW; ++q) X[q] = 0;
  }
}

// The plane-program interpreter, one WAVEFRONT per (shard, slot): X / M / S are 16 words per lane, the planes of the next
// kAhead instructions are in flight while the current one is applied (a load never depends on X / M / S).  (The round-1
// form — a 256-thread block per (shard, slot) that issued U loads, waited for all of them, applied, and only then read
// the next U descriptors — was removed in round 3: 142 us against 131.)
__global__ void __launch_bounds__(64) k_bsi_range_slot(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                      const uint32_t* __restrict__ base, uint32_t n_shards,
                                                      const uint32_t* __restrict__ prog, uint32_t prog_len, uint32_t n_rows_frag,
                                                      uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots,
                                                      uint32_t* __restrict__ outRuns, u64* __restrict__ out_counts, uint32_t encode) {
  __shared__ u64 lds[kWords];
  __shared__ Slot tab[kBsiDescCap];
  __shared__ u64 save[kWords];  // S lives in LDS (only BETWEEN programs use it): 32 registers fewer, two wavefronts per SIMD
  const int lane = threadIdx.x;
  const uint64_t cell = blockIdx.x;
  const uint64_t shard = cell >> 4;
  const uint32_t slot = cell & 15;
  if (shard >= n_shards) return;
  const uint64_t r0 = base[shard];
  bsi_stage_descs(slots, r0, slot, n_rows_frag, lane, tab);
  constexpr int kAhead = 2;  // (3 would need 300 registers: one wavefront per SIMD, and 1536 wavefronts do not fit 1024 SIMDs in one round)
  u64 X[kWordsPerLane], M[kWordsPerLane], T[kAhead][kWordsPerLane];
  frag_zero(X);
  frag_zero(M);
#pragma unroll
  for (int q = 0; q < kWordsPerLane; ++q) save[q * kWave + lane] = 0;
  auto issue = [&](uint32_t pc, u64 (&w)[kWordsPerLane]) {
    if (pc >= prog_len) return;
    const uint32_t ins = prog[pc];
    if ((ins >> 24) > kMorXAn) return;  // no operand
    const Slot sp = bsi_desc(slots, r0, slot, ins & 0xFFFFFFu, tab);
    if (slot_n(sp) == 0) frag_zero(w);
    else frag_load(sp, arena, lane, lds, w);
  };
#pragma unroll
  for (int u = 0; u < kAhead; ++u) {
    frag_zero(T[u]);
    issue((uint32_t)u, T[u]);
  }
  for (uint32_t pc0 = 0; pc0 < prog_len; pc0 += kAhead) {
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const uint32_t pc = pc0 + (uint32_t)u;
      if (pc < prog_len) {  // (wave-uniform)
        const uint32_t op = prog[pc] >> 24;
        if (op == kSaveX) {
#pragma unroll
          for (int q = 0; q < kWordsPerLane; ++q) save[q * kWave + lane] = X[q];
        } else if (op == kOrXS) {
#pragma unroll
          for (int q = 0; q < kWordsPerLane; ++q) X[q] |= save[q * kWave + lane];
        } else if (op == kAndnXS) {
#pragma unroll
          for (int q = 0; q < kWordsPerLane; ++q) X[q] &= ~save[q * kWave + lane];
        } else {
          bsi_apply2<kWordsPerLane>(op, X, M, T[u]);
        }
        issue(pc + kAhead, T[u]);
      }
    }
  }
  const uint32_t c = wave_reduce_add(frag_popcount(X));
  Slot so;
  so.off = cell * 8192ull;
  so.len = kWords;
  so.tn = make_tn(c ? kTypeBitmap : kTypeNil, c);
  uint32_t rr = 0;
  if (outRuns || encode) rr = wave_reduce_add(frag_count_runs(X, lane));  // bitmapCountRuns (roaring.go:3372-3380)
  if (encode && c) {  // Container.optimize() applied here (frag_store_encoded, fbk_kernels.hip.h): the scratch is free now
    uint32_t t_out, l_out;
    frag_store_encoded(X, c, rr, lane, lds, arenaO + so.off, t_out, l_out);
    so.len = l_out;
    so.tn = make_tn(t_out, c);
  } else if (c) {
    frag_store_bitmap(arenaO + so.off, lane, X);
  }
  if (lane == 0) {
    outSlots[cell] = so;
    if (outRuns) outRuns[cell] = rr;
    if (c && out_counts) atomicAdd(&out_counts[shard], (u64)c);
  }
}

// ---- Sum over a Range of the SAME field, one pass ("fused") ---------------------
// Sum(Row(v op k), field = v): the reference runs the range scan (a Row), then fragment.sum with that Row as the
// filter — every plane is read twice.  The scans of rangeLTUnsigned / rangeGTUnsigned (fragment.go:1070-1208) walk the
// planes MSB -> LSB with `remaining` X and `matched` M, and a column that ENTERS M at plane i has, above i, exactly the
// predicate's bits (it survived every X &= (~)plane_j and was not matched earlier) and at i the opposite bit — so its
// contribution above and at plane i is a constant per plane (vhi[i]), and its bits below i are still to come:
//     sum over M = Σ_i |newly matched at i| · vhi[i]  +  Σ_i 2^i · |M_before_i ∩ plane_i|
// — both terms available while plane i is in registers.  The other sign class, where the signed comparison takes it
// whole (v > k with k < 0 takes every positive column), is a plain sum over the same planes.  uint64 wrap-around is
// linear, so the totals equal the reference's Σ 2^i · count_i.  plan.action[i]: what the reference's loop does at plane
// i (it may stop early; the planes below are still read here, for the sums); the host builds it from the same
// predicate transformations as the plane programs above and falls back to the two-pass path for the special forms.
// one plane of the one-pass Range + Sum; branch-free over the five actions (uniform masks: specialising the body per
// action through a switch costs the register allocator 260+ spills at two wavefronts per SIMD, and the kernel waits for
// memory, not for the vector ALU)
template <bool OTHER>
__device__ __forceinline__ void range_sum_plane(uint32_t act, u64 (&X)[kWordsPerLane], u64 (&M)[kWordsPerLane], const u64 (&O)[OTHER ? kWordsPerLane : 1],
                                                const u64 (&T)[kWordsPerLane], uint32_t& a, uint32_t& o, uint32_t& d) {
  const u64 inv = (act == 2u || act == 4u) ? ~0ull : 0ull;
  const u64 keep_x = (act == 1u || act == 2u) ? 0ull : ~0ull;  // X &= tx only for actions 1 / 2
  const u64 match = act >= 3u ? ~0ull : 0ull;
#pragma unroll
  for (int q = 0; q < kWordsPerLane; ++q) {
    const u64 t = T[q];
    a += __popcll(M[q] & t);
    if (OTHER) o += __popcll(O[OTHER ? q : 0] & t);
    const u64 tx = t ^ inv;
    const u64 nw = X[q] & tx & ~M[q] & match;
    d += __popcll(nw);
    M[q] |= nw;
    X[q] &= tx | keep_x;
  }
}

// ---- planes in flight, counted by hand -------------------------------------------------------------------------
// The one-pass kernels keep kAhead planes in flight per wavefront and refill a register set as soon as its plane has been
// applied.  Written with ordinary loads, the compiler's s_waitcnt insertion has to merge "loaded in the prologue" with
// "refilled in the previous iteration" at the loop header (and, with the refill behind a condition, at every step): the
// ISA then waits for vmcnt(0) before a step — the plane requested a moment ago must land before the oldest one may be
// used, ONE plane in flight whatever kAhead says (round 3: found in the listing of every BSI streaming kernel).  Here the
// loads are `asm volatile` (invisible to that pass) and the kernel states the count itself: a plane's NW / 2 loads have
// landed when at most `PENDING` younger loads are outstanding.  Nothing else in the loop may touch vmcnt (plan values come
// through scalar loads and the masks of plane_codes); scripts/isa_stats.py --vmem lists a kernel's vector-memory
// instructions to check that.  A register set with a load in flight must not be COPIED either: the main loops below keep
// each set in fixed registers (checked in the listing); where the compiler is free to merge code paths — the last planes —
// everything is drained first (planes_drain) and the few planes never requested are read with ordinary loads.
typedef uint32_t bsi_u4 __attribute__((ext_vector_type(4)));

template <int NW>
__device__ __forceinline__ void plane_request(const uint8_t* __restrict__ p, int lane, bsi_u4 (&w)[NW / 2]) {
#pragma unroll
  for (int j = 0; j < NW / 2; ++j) {
    const uint8_t* a = p + (uint32_t)(j * kWave + lane) * 16u;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(w[j]) : "v"(a));
  }
}
template <int NW, int PENDING>
__device__ __forceinline__ void plane_landed(bsi_u4 (&w)[NW / 2]) {
  static_assert(NW == 4 || NW == 8, "a plane is two or four 16-byte loads per lane");
  if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(PENDING));
  else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(PENDING));
}
// everything requested so far has landed; afterwards the register sets hold ordinary values (the compiler may copy them)
template <int NW, int K>
__device__ __forceinline__ void planes_drain(bsi_u4 (&T)[K][NW / 2]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < K; ++u)
#pragma unroll
    for (int j = 0; j < NW / 2; ++j) asm volatile("" : "+v"(T[u][j]));
}
template <int NW>
__device__ __forceinline__ void plane_words(const bsi_u4 (&w)[NW / 2], u64 (&t)[NW]) {
#pragma unroll
  for (int j = 0; j < NW / 2; ++j) {
    t[2 * j] = ((u64)w[j][1] << 32) | w[j][0];
    t[2 * j + 1] = ((u64)w[j][3] << 32) | w[j][2];
  }
}

// The per-plane codes of a plan (one byte per plane, values 0..7) as three 64-bit masks in SCALAR registers: bit i of
// mask k = bit k of codes[i].  Read ONCE, by one vector load + three ballots, before any plane is requested.  Reading
// `codes[i]` inside the plane loop compiles to a global_load_ubyte (the scalar unit has no byte loads) followed by
// s_waitcnt vmcnt(0) — which also waits for EVERY plane in flight: the planes loaded ahead were serialised behind each
// step (found in the ISA in round 3; the one-pass kernels ran 129 / 199 us with it).
struct PlaneCodes {
  u64 b0, b1, b2;
};
__device__ __forceinline__ PlaneCodes plane_codes(const uint8_t* __restrict__ codes, int lane) {
  const uint32_t c = codes[lane];
  PlaneCodes m;
  m.b0 = __ballot((c & 1u) != 0);
  m.b1 = __ballot((c & 2u) != 0);
  m.b2 = __ballot((c & 4u) != 0);
  return m;
}
__device__ __forceinline__ uint32_t plane_code(const PlaneCodes& m, uint32_t i) {  // i: wave-uniform
  return (uint32_t)((m.b0 >> i) & 1ull) | ((uint32_t)((m.b1 >> i) & 1ull) << 1) | ((uint32_t)((m.b2 >> i) & 1ull) << 2);
}

struct RangeSumPlan {
  u64 vhi[64];
  uint8_t action[64];  // 0: sums only; 1: X &= T; 2: X &= ~T; 3: D = X & T & ~M; 4: D = X & ~T & ~M  (D: newly matched)
  uint32_t depth;
  uint32_t scan_positive;  // the scanned class: 1 = exists \ sign, 0 = exists ∩ sign
  uint32_t take_other;     // the other class belongs to the result as a whole
};

// OTHER: the other sign class is part of the result (one more fragment in registers: two planes in flight instead of three)
template <bool OTHER>
__global__ void __launch_bounds__(64) k_bsi_range_sum_slot(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                          const uint32_t* __restrict__ base, uint32_t n_shards, const RangeSumPlan* __restrict__ planp,
                                                          const Slot* __restrict__ fslots, const uint8_t* __restrict__ farena,
                                                          const uint32_t* __restrict__ frows, u64* __restrict__ out4) {
  __shared__ u64 lds[kWords];
  __shared__ Slot tab[kBsiDescCap];
  const int lane = threadIdx.x;
  const uint64_t shard = blockIdx.x >> 4;
  const uint32_t slot = blockIdx.x & 15;
  if (shard >= n_shards) return;
  const uint64_t r0 = base[shard];
  const RangeSumPlan& plan = *planp;  // (uniform: scalar loads)
  const uint32_t depth = plan.depth;
  const PlaneCodes codes = plane_codes(plan.action, lane);
  bsi_stage_descs(slots, r0, slot, depth + 2, lane, tab);
  if (slot_n(tab[0]) == 0) return;
  constexpr int kAhead = OTHER ? 2 : 3;
  u64 X[kWordsPerLane], M[kWordsPerLane], O[OTHER ? kWordsPerLane : 1], T[kAhead][kWordsPerLane];
  frag_load(tab[0], arena, lane, lds, X);  // consider = exists (∩ filter)
  if (fslots) {
    const Slot sf = fslots[(uint64_t)frows[shard] * kSlots + slot];
    if (slot_n(sf) == 0) return;
    frag_load(sf, farena, lane, lds, T[0]);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) X[q] &= T[0][q];
  }
  if (slot_n(tab[1]) == 0) frag_zero(T[0]);
  else frag_load(tab[1], arena, lane, lds, T[0]);
  {
    const u64 sp = plan.scan_positive ? ~0ull : 0ull;
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) {
      const u64 e = X[q], sg = T[0][q];
      X[q] = e & (sg ^ sp);          // scan_positive: e & ~sg; else e & sg
      if (OTHER) O[q] = e & ~(sg ^ sp);  // the other class
      M[q] = 0;
    }
  }
  auto load_plane = [&](uint32_t i, u64 (&w)[kWordsPerLane]) {
    const Slot sp = tab[2 + i];
    if (slot_n(sp) == 0) frag_zero(w);
    else frag_load(sp, arena, lane, lds, w);
  };
  u64 sum_m = 0, sum_o = 0;
  uint32_t cnt_m = 0;
#pragma unroll
  for (int u = 0; u < kAhead; ++u)
    if ((uint32_t)u < depth) load_plane(depth - 1 - (uint32_t)u, T[u]);
  for (uint32_t j0 = 0; j0 < depth; j0 += kAhead) {  // j counts planes from the most significant one down
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const uint32_t j = j0 + (uint32_t)u;
      if (j < depth) {  // (wave-uniform)
        const uint32_t i = depth - 1 - j;
        const uint32_t act = plane_code(codes, i);
        uint32_t a = 0, o = 0, d = 0;
        range_sum_plane<OTHER>(act, X, M, O, T[u], a, o, d);
        sum_m += ((u64)a << i) + (u64)d * plan.vhi[i];
        sum_o += (u64)o << i;
        cnt_m += d;
        if (j + kAhead < depth) load_plane(depth - 1 - (j + kAhead), T[u]);
      }
    }
  }
  uint32_t cnt_o = 0;
  if (OTHER) {
#pragma unroll
    for (int q = 0; q < (OTHER ? kWordsPerLane : 1); ++q) cnt_o += __popcll(O[q]);
    cnt_o = wave_reduce_add(cnt_o);
  }
  cnt_m = wave_reduce_add(cnt_m);
  sum_m = wave_reduce_add_u64(sum_m);
  sum_o = wave_reduce_add_u64(sum_o);
  if (lane == 0) {
    if (sum_m) atomicAdd(&out4[shard * 4 + 0], sum_m);
    if (sum_o) atomicAdd(&out4[shard * 4 + 1], sum_o);
    if (cnt_m) atomicAdd(&out4[shard * 4 + 2], (u64)cnt_m);
    if (cnt_o) atomicAdd(&out4[shard * 4 + 3], (u64)cnt_o);
  }
}

// ---- dense BSI batches: HALF a container per wavefront ---------------------------------------------------------
// One wavefront per (shard, slot) is 1536 wavefronts for 100 M columns: 6 per CU, i.e. two SIMDs of every CU carry two
// and two carry one, and at 200+ registers no more fit.  When the batch is in the dense layout (every container a bitmap
// at (row * 16 + slot) * 8192 — fbk_batch_upload_dense, the bit planes of any field with enough values) the payload of a
// plane needs no descriptor and no decode, and a (shard, slot) splits into two independent halves of 4 KiB per plane:
// 3072 wavefronts, 3 per SIMD everywhere, 8 words per lane, four planes in flight.  Sums and counts are reductions, the
// two halves just add into the same totals.  Used by the one-pass Range + Sum, whose plane step is the heaviest (150 ->
// 136 us on config 5); the plain Sum gains nothing from it (133 vs 136 us, scripts/bsi_bench.py) and stays with one
// wavefront per container, as do the Range OUTPUT rows, whose run count crosses the middle.
constexpr int kHalfWords = 8;

__device__ __forceinline__ void half_load(const uint8_t* __restrict__ p, int lane, u64 (&w)[kHalfWords]) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const ulonglong2 v = ld_stream(&q[j * kWave + lane]);
    w[2 * j] = v.x;
    w[2 * j + 1] = v.y;
  }
}

// half `h` of any container (the filter row may come from anywhere): bitmaps directly, the rest through the full decode
__device__ __forceinline__ void half_load_any(const Slot& s, const uint8_t* __restrict__ arena, int lane, uint32_t h, u64* lds, u64 (&w)[kHalfWords]) {
  if (slot_n(s) == 0) {
#pragma unroll
    for (int q = 0; q < kHalfWords; ++q) w[q] = 0;
  } else if (slot_type(s) == kTypeBitmap) {
    half_load(arena + s.off + h * 4096u, lane, w);
  } else {
    u64 full[kWordsPerLane];
    frag_load(s, arena, lane, lds, full);
#pragma unroll
    for (int q = 0; q < kHalfWords; ++q) w[q] = h ? full[kHalfWords + q] : full[q];
  }
}

template <bool OTHER, int AHEAD = 4>
__global__ void __launch_bounds__(64) k_bsi_range_sum_half(const uint8_t* __restrict__ arena, const uint32_t* __restrict__ base, uint32_t n_shards,
                                                          const RangeSumPlan* __restrict__ planp, const Slot* __restrict__ fslots,
                                                          const uint8_t* __restrict__ farena, const uint32_t* __restrict__ frows, u64* __restrict__ out4) {
  __shared__ u64 lds[kWords];
  const int lane = threadIdx.x;
  const uint32_t h = blockIdx.x & 1u;
  const uint64_t cell = blockIdx.x >> 1;
  const uint64_t shard = cell >> 4;
  const uint32_t slot = cell & 15;
  if (shard >= n_shards) return;
  const RangeSumPlan& plan = *planp;
  const uint32_t depth = plan.depth;
  const PlaneCodes codes = plane_codes(plan.action, lane);
  const uint8_t* const row0 = arena + ((uint64_t)base[shard] * kSlots + slot) * 8192ull + h * 4096u;
  constexpr uint64_t kRow = (uint64_t)kSlots * 8192ull;
  constexpr int kAhead = AHEAD;  // planes in flight per wavefront (option bsi_planes_ahead)
  u64 X[kHalfWords], M[kHalfWords], O[OTHER ? kHalfWords : 1], R[kHalfWords];
  bsi_u4 T[kAhead][kHalfWords / 2] = {};  // planes in flight (see plane_request)
  half_load(row0, lane, X);
  if (fslots) {
    half_load_any(fslots[(uint64_t)frows[shard] * kSlots + slot], farena, lane, h, lds, R);
#pragma unroll
    for (int q = 0; q < kHalfWords; ++q) X[q] &= R[q];
  }
  {
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < kHalfWords; ++q) any |= (uint32_t)(X[q] != 0);
    if (__ballot(any != 0) == 0) return;  // nothing to consider in this half
  }
  half_load(row0 + kRow, lane, R);
  {
    const u64 sp = plan.scan_positive ? ~0ull : 0ull;
#pragma unroll
    for (int q = 0; q < kHalfWords; ++q) {
      const u64 e = X[q], sg = R[q];
      X[q] = e & (sg ^ sp);
      if (OTHER) O[q] = e & ~(sg ^ sp);
      M[q] = 0;
    }
  }
  u64 sum_m = 0, sum_o = 0;
  uint32_t cnt_m = 0;
#pragma unroll
  for (int u = 0; u < kAhead; ++u)
    if ((uint32_t)u < depth) plane_request<kHalfWords>(row0 + kRow * (2u + depth - 1 - (uint32_t)u), lane, T[u]);
  auto step_words = [&](uint32_t j, const u64 (&tw)[kHalfWords]) {
    const uint32_t i = depth - 1 - j;
    const uint32_t act = plane_code(codes, i);
    const u64 inv = (act == 2u || act == 4u) ? ~0ull : 0ull;
    const u64 keep_x = (act == 1u || act == 2u) ? 0ull : ~0ull;
    const u64 match = act >= 3u ? ~0ull : 0ull;
    uint32_t a = 0, o = 0, d = 0;
#pragma unroll
    for (int q = 0; q < kHalfWords; ++q) {
      const u64 t = tw[q];
      a += __popcll(M[q] & t);
      if (OTHER) o += __popcll(O[OTHER ? q : 0] & t);
      const u64 tx = t ^ inv;
      const u64 nw = X[q] & tx & ~M[q] & match;
      d += __popcll(nw);
      M[q] |= nw;
      X[q] &= tx | keep_x;
    }
    sum_m += ((u64)a << i) + (u64)d * plan.vhi[i];
    sum_o += (u64)o << i;
    cnt_m += d;
  };
  auto step = [&](uint32_t j, const bsi_u4 (&w)[kHalfWords / 2]) {
    u64 tw[kHalfWords];
    plane_words<kHalfWords>(w, tw);
    step_words(j, tw);
  };
  // full groups: every step refills its register set, so exactly kAhead - 1 younger planes are in flight at each use
  uint32_t j0 = 0;
  for (; j0 + 2u * kAhead <= depth; j0 += kAhead) {
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      plane_landed<kHalfWords, (kAhead - 1) * (kHalfWords / 2)>(T[u]);
      step(j0 + (uint32_t)u, T[u]);
      plane_request<kHalfWords>(row0 + kRow * (2u + depth - 1 - (j0 + (uint32_t)u + kAhead)), lane, T[u]);
    }
  }
  // the last planes: what is in the register sets (planes j0 .. j0 + kAhead - 1 as far as they exist), then the at
  // most kAhead - 1 planes that were never requested
  planes_drain<kHalfWords, kAhead>(T);
#pragma unroll
  for (int u = 0; u < kAhead; ++u)
    if (j0 + (uint32_t)u < depth) step(j0 + (uint32_t)u, T[u]);
  for (uint32_t j = j0 + kAhead; j < depth; ++j) {
    half_load(row0 + kRow * (2u + depth - 1 - j), lane, R);
    step_words(j, R);
  }
  uint32_t cnt_o = 0;
  if (OTHER) {
#pragma unroll
    for (int q = 0; q < (OTHER ? kHalfWords : 1); ++q) cnt_o += __popcll(O[q]);
    cnt_o = wave_reduce_add(cnt_o);
  }
  cnt_m = wave_reduce_add(cnt_m);
  sum_m = wave_reduce_add_u64(sum_m);
  sum_o = wave_reduce_add_u64(sum_o);
  if (lane == 0) {
    if (sum_m) atomicAdd(&out4[shard * 4 + 0], sum_m);
    if (sum_o) atomicAdd(&out4[shard * 4 + 1], sum_o);
    if (cnt_m) atomicAdd(&out4[shard * 4 + 2], (u64)cnt_m);
    if (cnt_o) atomicAdd(&out4[shard * 4 + 3], (u64)cnt_o);
  }
}

// ---- Sum over lo <= v <= hi of the same field, one pass ---------------------------------------------------------
// Two scan lanes, each with its own remaining set X and matched set M.  lo < 0 <= hi: lane 0 = the positive columns with
// magnitude <= hi, lane 1 = the negative ones with magnitude <= |lo| (two "less or equal" scans from the top plane).
// Bounds of one sign, mn < mx: the planes above the highest bit t where mn and mx differ filter lane 0 down to the columns
// that share the common prefix; plane t SPLITS them — bit 1 (already above mn) goes to lane 1 and goes on as "<= mx", bit
// 0 (already below mx) stays in lane 0 and goes on as ">= mn".  A scan step either drops columns (X &= (~)plane) or
// MATCHES the columns whose bit decides the comparison; a column matched at plane i carries the bound's bits above i and
// the deciding bit (vhi), its lower planes are added as they stream past (|M ∩ plane| << i); what is left in a lane
// after plane 0 equals its bound (vfin).  Unlike the plane programs this is not a transcription of the reference's loops
// (rangeBetweenUnsigned runs its two scans one after the other, fragment.go:1262-1303) but the same sets by the
// definition of the comparison; tests/test_range_sum_plan.py executes the schedule for every pair of bounds at small depths.
struct BetweenSumPlan {
  u64 vhi[2][64];
  u64 vfin[2];
  uint8_t action[2][64];  // 0 none; 1 X &= T; 2 X &= ~T; 3 match X & T (X keeps the rest); 4 match X & ~T
  uint8_t split[64];      // 1: lane 1 takes X0 & T, lane 0 keeps X0 & ~T (before nothing else happens at this plane)
  uint32_t depth;
  uint32_t class_pos[2];  // lane l scans exists \ sign (1) or exists ∩ sign (0)
  uint32_t init_b;        // lane 1 starts as its whole class (two sign classes); else empty until the split
  uint32_t whole[2];      // the lane's class belongs to the result as a whole (bound beyond the bit depth): M = class, X = ∅
};

// NW = words per lane (kHalfWords: dense batches, half a container per wavefront; other batches take the two-pass path —
// four fragments of 16 words plus the planes in flight do not fit two wavefronts per SIMD)
// one lane's step, specialised on its action
template <int ACT, int NW>
__device__ __forceinline__ void between_lane(u64 (&X)[NW], u64 (&M)[NW], const u64 (&T)[NW], uint32_t& low, uint32_t& d) {
#pragma unroll
  for (int q = 0; q < NW; ++q) {
    const u64 t = T[q];
    low += __popcll(M[q] & t);
    if (ACT == 1) X[q] &= t;
    if (ACT == 2) X[q] &= ~t;
    if (ACT == 3 || ACT == 4) {
      const u64 n = ACT == 3 ? (X[q] & t) : (X[q] & ~t);
      d += __popcll(n);
      M[q] |= n;
      X[q] ^= n;
    }
  }
}

template <int NW>
__device__ __forceinline__ void between_lane_any(uint32_t act, u64 (&X)[NW], u64 (&M)[NW], const u64 (&T)[NW], uint32_t& low, uint32_t& d) {
  switch (act) {  // (wave-uniform)
    case 1: between_lane<1, NW>(X, M, T, low, d); break;
    case 2: between_lane<2, NW>(X, M, T, low, d); break;
    case 3: between_lane<3, NW>(X, M, T, low, d); break;
    case 4: between_lane<4, NW>(X, M, T, low, d); break;
    default: between_lane<0, NW>(X, M, T, low, d); break;
  }
}

template <int NW>
__device__ __forceinline__ void between_sum_plane(uint32_t a0, uint32_t a1, uint32_t sp, u64 (&X0)[NW], u64 (&X1)[NW], u64 (&M0)[NW], u64 (&M1)[NW],
                                                  const u64 (&T)[NW], uint32_t (&low)[2], uint32_t (&d)[2]) {
  if (sp) {  // the split plane: nothing else happens at it
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const u64 t = T[q];
      low[0] += __popcll(M0[q] & t);
      low[1] += __popcll(M1[q] & t);
      X1[q] |= X0[q] & t;
      X0[q] &= ~t;
    }
    return;
  }
  between_lane_any<NW>(a0, X0, M0, T, low[0], d[0]);
  between_lane_any<NW>(a1, X1, M1, T, low[1], d[1]);
}

template <int NW, int AHEAD, typename LoadPlane, typename LoadRow>
__device__ __forceinline__ void between_sum_body(const BetweenSumPlan& plan, int lane, LoadRow&& load_row, LoadPlane&& load_plane, bool has_filter,
                                                 u64* __restrict__ out4, uint64_t shard) {
  constexpr int kAhead = AHEAD;  // planes in flight per wavefront (option bsi_planes_ahead)
  const uint32_t depth = plan.depth;
  const PlaneCodes codes0 = plane_codes(plan.action[0], lane), codes1 = plane_codes(plan.action[1], lane), codes_sp = plane_codes(plan.split, lane);
  u64 X0[NW], X1[NW], M0[NW], M1[NW], R[NW];
  bsi_u4 T[kAhead][NW / 2] = {};
  load_row(0u, X0);  // exists
  if (has_filter) {
    load_row(~0u, R);
#pragma unroll
    for (int q = 0; q < NW; ++q) X0[q] &= R[q];
  }
  {
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) any |= (uint32_t)(X0[q] != 0);
    if (__ballot(any != 0) == 0) return;  // nothing to consider here
  }
  load_row(1u, R);  // sign
  {
    const u64 p0 = plan.class_pos[0] ? ~0ull : 0ull, p1 = plan.class_pos[1] ? ~0ull : 0ull;
    const u64 ib = plan.init_b ? ~0ull : 0ull, w0 = plan.whole[0] ? ~0ull : 0ull, w1 = plan.whole[1] ? ~0ull : 0ull;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const u64 e = X0[q], sg = R[q];
      const u64 c0 = e & (sg ^ p0), c1 = e & (sg ^ p1) & ib;
      X0[q] = c0 & ~w0;
      M0[q] = c0 & w0;
      X1[q] = c1 & ~w1;
      M1[q] = c1 & w1;
    }
  }
  u64 sum[2] = {0, 0};
  uint32_t cnt[2] = {0, 0};
#pragma unroll
  for (int q = 0; q < NW; ++q) {  // whole classes are matched from the start
    cnt[0] += __popcll(M0[q]);
    cnt[1] += __popcll(M1[q]);
  }
#pragma unroll
  for (int u = 0; u < kAhead; ++u)
    if ((uint32_t)u < depth) load_plane(depth - 1 - (uint32_t)u, T[u]);
  auto step_words = [&](uint32_t j, const u64 (&t)[NW]) {
    const uint32_t i = depth - 1 - j;
    uint32_t low[2] = {0, 0}, d[2] = {0, 0};
    between_sum_plane<NW>(plane_code(codes0, i), plane_code(codes1, i), (uint32_t)((codes_sp.b0 >> i) & 1ull), X0, X1, M0, M1, t, low, d);
    sum[0] += ((u64)low[0] << i) + (u64)d[0] * plan.vhi[0][i];
    sum[1] += ((u64)low[1] << i) + (u64)d[1] * plan.vhi[1][i];
    cnt[0] += d[0];
    cnt[1] += d[1];
  };
  auto step = [&](uint32_t j, const bsi_u4 (&w)[NW / 2]) {
    u64 t[NW];
    plane_words<NW>(w, t);
    step_words(j, t);
  };
  // full groups: every step refills its register set, so exactly kAhead - 1 younger planes are in flight at each use
  uint32_t j0 = 0;
  for (; j0 + 2u * kAhead <= depth; j0 += kAhead) {
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      plane_landed<NW, (kAhead - 1) * (NW / 2)>(T[u]);
      step(j0 + (uint32_t)u, T[u]);
      load_plane(depth - 1 - (j0 + (uint32_t)u + kAhead), T[u]);
    }
  }
  // the last planes: what is in the register sets (planes j0 .. j0 + kAhead - 1 as far as they exist), then the at
  // most kAhead - 1 planes that were never requested
  planes_drain<NW, kAhead>(T);
#pragma unroll
  for (int u = 0; u < kAhead; ++u)
    if (j0 + (uint32_t)u < depth) step(j0 + (uint32_t)u, T[u]);
  for (uint32_t j = j0 + kAhead; j < depth; ++j) {
    load_row(2u + (depth - 1 - j), R);
    step_words(j, R);
  }
  {  // what is left in a lane equals its bound
    uint32_t r0 = 0, r1 = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      r0 += __popcll(X0[q]);
      r1 += __popcll(X1[q]);
    }
    sum[0] += (u64)r0 * plan.vfin[0];
    sum[1] += (u64)r1 * plan.vfin[1];
    cnt[0] += r0;
    cnt[1] += r1;
  }
  const u64 s0 = wave_reduce_add_u64(sum[0]), s1 = wave_reduce_add_u64(sum[1]);
  const uint32_t c0 = wave_reduce_add(cnt[0]), c1 = wave_reduce_add(cnt[1]);
  if (lane == 0) {
    if (s0) atomicAdd(&out4[shard * 4 + 0], s0);
    if (s1) atomicAdd(&out4[shard * 4 + 1], s1);
    if (c0) atomicAdd(&out4[shard * 4 + 2], (u64)c0);
    if (c1) atomicAdd(&out4[shard * 4 + 3], (u64)c1);
  }
}

// NW words per lane = NW * 512 bytes of every plane per wavefront.  The library instantiates 8 = half a container (four
// 8-word fragments + AHEAD planes in flight: ~205-220 registers, two wavefronts per SIMD; sums and counts are reductions,
// the two halves just add into the same totals).  Round 2 measured 205-215 us for 830 MB and called the step "vector-ALU
// bound"; the listing showed s_waitcnt vmcnt(0) in front of every step (see plane_request): with the planes really in flight
// it is 135-141 us.  NW = 4 (a quarter: ~120 registers, 6144 wavefronts) was tried in round 3 — 199 us before, 143-145 us
// after the same fix, and two of its instantiations failed scripts/check_inflight.py: not instantiated any more.
template <int NW>
__device__ __forceinline__ void part_load(const uint8_t* __restrict__ p, int lane, u64 (&w)[NW]) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
#pragma unroll
  for (int j = 0; j < NW / 2; ++j) {
    const ulonglong2 v = ld_stream(&q[j * kWave + lane]);
    w[2 * j] = v.x;
    w[2 * j + 1] = v.y;
  }
}

// part `part` of any container (the filter row may come from anywhere): bitmaps directly, the rest through the full decode
template <int NW>
__device__ __forceinline__ void part_load_any(const Slot& s, const uint8_t* __restrict__ arena, int lane, uint32_t part, u64* lds, u64 (&w)[NW]) {
  if (slot_n(s) == 0) {
#pragma unroll
    for (int q = 0; q < NW; ++q) w[q] = 0;
  } else if (slot_type(s) == kTypeBitmap) {
    part_load<NW>(arena + s.off + part * (NW * 512u), lane, w);
  } else {
    u64 full[kWordsPerLane];
    frag_load(s, arena, lane, lds, full);
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      u64 v = full[q];
#pragma unroll
      for (int pp = 1; pp < kWordsPerLane / NW; ++pp) v = part == (uint32_t)pp ? full[pp * NW + q] : v;
      w[q] = v;
    }
  }
}

template <int NW, int AHEAD = 4>
__global__ void __launch_bounds__(64) k_bsi_between_sum_part(const uint8_t* __restrict__ arena, const uint32_t* __restrict__ base, uint32_t n_shards,
                                                            const BetweenSumPlan* __restrict__ planp, const Slot* __restrict__ fslots,
                                                            const uint8_t* __restrict__ farena, const uint32_t* __restrict__ frows, u64* __restrict__ out4) {
  __shared__ u64 lds[kWords];
  constexpr uint32_t kParts = kWordsPerLane / NW;
  const int lane = threadIdx.x;
  const uint32_t part = blockIdx.x % kParts;
  const uint64_t cell = blockIdx.x / kParts;
  const uint64_t shard = cell >> 4;
  const uint32_t slot = cell & 15;
  if (shard >= n_shards) return;
  const uint8_t* const row0 = arena + ((uint64_t)base[shard] * kSlots + slot) * 8192ull + part * (NW * 512u);
  constexpr uint64_t kRow = (uint64_t)kSlots * 8192ull;
  Slot sf;
  sf.off = 0, sf.len = 0, sf.tn = 0;
  if (fslots) sf = fslots[(uint64_t)frows[shard] * kSlots + slot];
  auto load_row = [&](uint32_t r, u64 (&w)[NW]) {
    if (r == ~0u) part_load_any<NW>(sf, farena, lane, part, lds, w);
    else part_load<NW>(row0 + kRow * r, lane, w);
  };
  auto load_plane = [&](uint32_t i, bsi_u4 (&w)[NW / 2]) { plane_request<NW>(row0 + kRow * (2u + i), lane, w); };
  between_sum_body<NW, AHEAD>(*planp, lane, load_row, load_plane, fslots != nullptr, out4, shard);
}

// ---- BSI Min / Max ------------------------------------------------------------------------------
// fragment.min / fragment.max / minUnsigned / maxUnsigned (fragment.go:754-853): a scan over the bit planes, MSB -> LSB,
// that keeps the candidate set and at every plane needs the cardinality of candidates ∩ plane (max) or candidates \ plane
// (min).  Done per (shard, SLOT): min / max over a shard = min / max over its 16 slots' own minima / maxima (the count of
// the winning value adds up over the slots that reach it), so the 16 slots need not be scanned in lock step: one wavefront
// owns one (shard, slot) — 1536 independent scans for 100 M columns fill the chip, the next plane's loads are in flight
// while the current plane is counted (they do not depend on the decision), and there is no barrier at all.  (The round-1
// form, one 1024-thread block per shard with a barrier per plane, ran 96 blocks on 256 CUs: 256 us against 121; removed in
// round 3.)  mode 0 = min, 1 = max.  out2[(shard * 16 + slot)] = {magnitude, count | scan flag}; the host folds the 16
// pairs of a shard (fbk_query_api.inc bsi_minmax).
__global__ void __launch_bounds__(64) k_bsi_minmax_slot(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                       const uint32_t* __restrict__ base, uint32_t n_shards, uint32_t bit_depth,
                                                       uint32_t mode, const Slot* __restrict__ fslots, const uint8_t* __restrict__ farena,
                                                       const uint32_t* __restrict__ frows, u64* __restrict__ out2) {
  __shared__ u64 lds[kWords];
  __shared__ Slot tab[kBsiDescCap];
  const int lane = threadIdx.x;
  const uint64_t cell = blockIdx.x;
  const uint64_t shard = cell >> 4;
  const uint32_t slot = cell & 15;
  if (shard >= n_shards) return;
  const uint64_t r0 = base[shard];
  bsi_stage_descs(slots, r0, slot, bit_depth + 2, lane, tab);
  auto load_plane = [&](uint64_t row, u64 (&w)[kWordsPerLane]) {
    const Slot sp = bsi_desc(slots, r0, slot, (uint32_t)(row - r0), tab);
    if (slot_n(sp) == 0) frag_zero(w);
    else frag_load(sp, arena, lane, lds, w);
  };
  u64 F[kWordsPerLane], T[kWordsPerLane], N[kWordsPerLane];
  load_plane(r0 + 0, F);  // consider = exists ∩ filter
  if (fslots) {
    const Slot sf = fslots[(uint64_t)frows[shard] * kSlots + slot];
    if (slot_n(sf) == 0) frag_zero(T);
    else frag_load(sf, farena, lane, lds, T);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) F[q] &= T[q];
  }
  uint32_t cur = wave_reduce_add(frag_popcount(F));
  if (cur == 0) {
    if (lane == 0) {
      out2[cell * 2] = 0;
      out2[cell * 2 + 1] = 0;
    }
    return;
  }
  bool scan_max, negate;
  {
    load_plane(r0 + 1, T);  // sign
    u64 G[kWordsPerLane];
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) G[q] = mode == 0 ? (F[q] & T[q]) : (F[q] & ~T[q]);
    const uint32_t g = wave_reduce_add(frag_popcount(G));
    if (g != 0) {
      scan_max = true;
      negate = (mode == 0);
      cur = g;
#pragma unroll
      for (int q = 0; q < kWordsPerLane; ++q) F[q] = G[q];
    } else {
      scan_max = false;
      negate = (mode == 1);
    }
  }
  u64 val = 0;
  if (bit_depth) load_plane(r0 + 2 + (uint64_t)(bit_depth - 1), T);
  for (int i = (int)bit_depth - 1; i >= 0; --i) {
    if (i > 0) load_plane(r0 + 2 + (uint64_t)(i - 1), N);  // the next plane, whatever this one decides
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) T[q] = scan_max ? (F[q] & T[q]) : (F[q] & ~T[q]);
    const uint32_t c = wave_reduce_add(frag_popcount(T));
    if (c > 0) {
#pragma unroll
      for (int q = 0; q < kWordsPerLane; ++q) F[q] = T[q];
      cur = c;
      if (scan_max) val += 1ull << i;
    } else if (!scan_max) {
      val += 1ull << i;
    }
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) T[q] = N[q];
  }
  (void)negate;
  if (lane == 0) {
    // the unsigned magnitude and, in bit 63 of the count word, which scan produced it: 1 = maxUnsigned
    // over the columns whose sign decides the result (negatives for min, positives for max), 0 =
    // minUnsigned over all considered columns.  The host applies the sign AFTER folding the slots, so
    // that magnitudes are compared unsigned exactly as fragment.min / max do over the whole row.
    out2[cell * 2] = val;
    out2[cell * 2 + 1] = (u64)cur | (scan_max ? (1ull << 63) : 0ull);
  }
}

// ---- BSI adder ---------------------------------------------------------------------------------
// roaring.Add (roaring/add.go:12-849), used by AddBSI (bsi.go:83-175) to merge the per-shard
// TopK count BSIs: z = x + y over unsigned bit-sliced values, plane i of z = x_i ^ y_i ^ carry,
// carry' = majority(x_i, y_i, carry) — a ripple-carry adder evaluated for 65 536 columns at a
// time.  One block per (group, slot) in the block layout: the carry lives in 4 u64 per thread
// while the planes of both operands stream past once and the sum planes stream out; plane D
// (D = max depth) receives the final carry.  Nil planes read as zero.
__global__ void __launch_bounds__(256) k_bsi_add(const Slot* __restrict__ slotsX, const uint8_t* __restrict__ arenaX,
                                                const uint32_t* __restrict__ rowsX, uint32_t dx,
                                                const Slot* __restrict__ slotsY, const uint8_t* __restrict__ arenaY,
                                                const uint32_t* __restrict__ rowsY, uint32_t dy, uint64_t n_groups,
                                                uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots,
                                                uint32_t* __restrict__ outRuns) {
  __shared__ u64 scratch[kWords];
  __shared__ u64 part[4];
  __shared__ uint8_t tops[512];
  // Round 6.  The (group, slot)'s descriptors come into the LDS ONCE, in one parallel round of loads (until round 5 every plane
  // step walked row index -> descriptor -> payload: three dependent round trips per plane), and when every plane is a bitmap (or
  // nil: the usual case) the planes go FOUR AT A TIME — sixteen 16-byte loads per thread in flight where there were four — with no
  // barrier in the loop: a wave leaves its share of each plane's cardinality and run count in the LDS and the block adds them up
  // once at the end.
  __shared__ Slot dsc[2][65];
  __shared__ uint32_t wpc[66][4], wrr[66][4];
  __shared__ uint8_t wedge[66][4][4];  // per plane and wave: bit 0 of the first lane's z0 and z2, top bit of the last lane's z1 and z3
  __shared__ int slow;
  const int t = threadIdx.x;
  const uint64_t g = blockIdx.x >> 4;
  const uint32_t slot = blockIdx.x & 15;
  if (g >= n_groups) return;
  const uint32_t D = max(dx, dy);
  if (t == 0) slow = 0;
  __syncthreads();
  if ((uint32_t)t < dx) {
    const Slot sl = slotsX[(uint64_t)rowsX[g * dx + t] * kSlots + slot];
    dsc[0][t] = sl;
    if (!bfrag_is_fast(sl)) slow = 1;
  } else if (t >= 128 && (uint32_t)(t - 128) < dy) {
    const Slot sl = slotsY[(uint64_t)rowsY[g * dy + (t - 128)] * kSlots + slot];
    dsc[1][t - 128] = sl;
    if (!bfrag_is_fast(sl)) slow = 1;
  }
  __syncthreads();
  u64 c[kBW], x[kBW], y[kBW], z[kBW];
  bfrag_zero(c);
  if (slow) {
    // array / run containers among the planes: the decoding path (block barriers inside bfrag_load), one plane per step
    for (uint32_t i = 0; i <= D; ++i) {
      bfrag_zero(x);
      bfrag_zero(y);
      if (i < dx) bfrag_load(dsc[0][i], arenaX, t, scratch, x);
      if (i < dy) bfrag_load(dsc[1][i], arenaY, t, scratch, y);
#pragma unroll
      for (int q = 0; q < kBW; ++q) {
        z[q] = x[q] ^ y[q] ^ c[q];
        c[q] = (x[q] & y[q]) | (c[q] & (x[q] ^ y[q]));
      }
      const uint64_t cell = (g * (D + 1) + i) * kSlots + slot;
      const uint32_t n = (uint32_t)block_reduce_add_u64(bfrag_popcount(z), part);
      Slot so;
      so.off = cell * 8192ull;
      so.len = kWords;
      so.tn = make_tn(n ? kTypeBitmap : kTypeNil, n);
      if (n) bfrag_store_bitmap(arenaO + so.off, t, z);
      uint32_t rr = 0;
      if (outRuns) {  // bitmapCountRuns for the optimize() pass
        __syncthreads();
        tops[t] = (uint8_t)(z[1] >> 63);
        tops[256 + t] = (uint8_t)(z[3] >> 63);
        __syncthreads();
        const u64 l0 = t ? tops[t - 1] : 0, l1 = tops[255 + t];
        const uint32_t r = __popcll(z[0] & ~((z[0] << 1) | l0)) + __popcll(z[1] & ~((z[1] << 1) | (z[0] >> 63))) +
                           __popcll(z[2] & ~((z[2] << 1) | l1)) + __popcll(z[3] & ~((z[3] << 1) | (z[2] >> 63)));
        rr = (uint32_t)block_reduce_add_u64(r, part);
      }
      if (t == 0) {
        outSlots[cell] = so;
        if (outRuns) outRuns[cell] = rr;
      }
    }
    return;
  }
  const int lane = t & 63, wv = t >> 6;
  constexpr int U = 4;  // planes per round of loads
  const uint64_t cell0 = g * (D + 1) * kSlots + slot;  // cell of plane i: cell0 + i * kSlots
  for (uint32_t i0 = 0; i0 <= D; i0 += U) {
    ulonglong2 xa[U][2], ya[U][2];
    u64 mx[U], my[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // a plane that is not there (past an operand's depth, nil, empty) reads this step's own output cell — valid memory — under a zero mask
      const uint32_t i = min(i0 + (uint32_t)u, D);
      const uint8_t* dummy = arenaO + (cell0 + (uint64_t)i * kSlots) * 8192ull;
      const Slot sx = dsc[0][min(i, 64u)], sy = dsc[1][min(i, 64u)];
      const bool lx = i0 + u < dx && slot_n(sx) != 0 && slot_type(sx) == kTypeBitmap, ly = i0 + u < dy && slot_n(sy) != 0 && slot_type(sy) == kTypeBitmap;
      mx[u] = lx ? ~0ull : 0ull;
      my[u] = ly ? ~0ull : 0ull;
      const ulonglong2* qx = reinterpret_cast<const ulonglong2*>(lx ? arenaX + sx.off : dummy);
      const ulonglong2* qy = reinterpret_cast<const ulonglong2*>(ly ? arenaY + sy.off : dummy);
      xa[u][0] = ld_stream(&qx[t]);
      xa[u][1] = ld_stream(&qx[256 + t]);
      ya[u][0] = ld_stream(&qy[t]);
      ya[u][1] = ld_stream(&qy[256 + t]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + (uint32_t)u;
      if (i > D) break;  // (block-uniform)
      x[0] = xa[u][0].x & mx[u];
      x[1] = xa[u][0].y & mx[u];
      x[2] = xa[u][1].x & mx[u];
      x[3] = xa[u][1].y & mx[u];
      y[0] = ya[u][0].x & my[u];
      y[1] = ya[u][0].y & my[u];
      y[2] = ya[u][1].x & my[u];
      y[3] = ya[u][1].y & my[u];
#pragma unroll
      for (int q = 0; q < kBW; ++q) {
        z[q] = x[q] ^ y[q] ^ c[q];
        c[q] = (x[q] & y[q]) | (c[q] & (x[q] ^ y[q]));
      }
      // always stored (an empty plane's cell is described as nil below: its bytes are never read)
      bfrag_store_bitmap(arenaO + (cell0 + (uint64_t)i * kSlots) * 8192ull, t, z);
      const uint32_t pc = wave_reduce_add(bfrag_popcount(z));
      if (lane == 0) wpc[i][wv] = pc;
      if (outRuns) {
        // run starts with the predecessor bit taken from the lane below (words 2t, 2t + 1 and 512 + 2t, 513 + 2t); a wave's first
        // lane assumes "no predecessor" and the block corrects that from the waves' edge bits at the end
        const uint32_t t1 = (uint32_t)(z[1] >> 63), t3 = (uint32_t)(z[3] >> 63);
        uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t1, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        uint32_t p3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t3, 0x138, 0xF, 0xF, false);
        if (lane == 0) p1 = p3 = 0;
        uint32_t r = __popcll(z[0] & ~((z[0] << 1) | (u64)p1)) + __popcll(z[1] & ~((z[1] << 1) | (z[0] >> 63))) +
                     __popcll(z[2] & ~((z[2] << 1) | (u64)p3)) + __popcll(z[3] & ~((z[3] << 1) | (z[2] >> 63)));
        r = wave_reduce_add(r);
        if (lane == 0) {
          wrr[i][wv] = r;
          wedge[i][wv][0] = (uint8_t)(z[0] & 1ull);
          wedge[i][wv][1] = (uint8_t)(z[2] & 1ull);
        }
        if (lane == 63) {
          wedge[i][wv][2] = (uint8_t)t1;
          wedge[i][wv][3] = (uint8_t)t3;
        }
      }
    }
  }
  __syncthreads();
  if ((uint32_t)t <= D) {
    const uint32_t i = (uint32_t)t;
    const uint32_t n = wpc[i][0] + wpc[i][1] + wpc[i][2] + wpc[i][3];
    const uint64_t cell = cell0 + (uint64_t)i * kSlots;
    Slot so;
    so.off = cell * 8192ull;
    so.len = kWords;
    so.tn = make_tn(n ? kTypeBitmap : kTypeNil, n);
    outSlots[cell] = so;
    if (outRuns) {
      uint32_t rr = wrr[i][0] + wrr[i][1] + wrr[i][2] + wrr[i][3];
      // a wave's first word continues a run that ends in the word before it: words 0..511 — wave w's first is 128 w, its
      // predecessor the last of wave w - 1's z1; words 512..1023 — wave w's first is 512 + 128 w, predecessor wave w - 1's z3
      // (wave 0's: word 511 = wave 3's z1)
      for (int w = 1; w < 4; ++w) rr -= (uint32_t)(wedge[i][w][0] & wedge[i][w - 1][2]) + (uint32_t)(wedge[i][w][1] & wedge[i][w - 1][3]);
      rr -= (uint32_t)(wedge[i][0][1] & wedge[i][3][2]);
      outRuns[cell] = rr;
    }
  }
}

// ---- BSI Distinct: bit planes -> per-column values ------------------------------------------------
// executeDistinctShardBSI (executor.go:2034-2153) walks the exists bitmap bit by bit and gathers
// the value of every column from the bit planes.  Here that is a 64 x 64 bit-matrix transpose in
// registers: lane i of a wavefront holds 64 columns' worth of plane i (one u64), a 6-stage
// butterfly of lane exchanges turns it into lane j holding the 64-bit value of column j, and the
// values of the columns in exists ∩ filter are appended to a global list (sorted and
// de-duplicated afterwards).  Operands must be DENSE rows (k_densify_rows makes them so): lane i
// reads a whole 128-byte line of plane i per round, i.e. 16 word positions per load.
//   rows[shard] = ordinal of the exists row; +1 sign, +2+i plane i.   `split` blocks per (shard, slot), 16 / split rounds of
//   1024 columns per wave each (round 5: it was ONE block per (shard, slot) — 64 blocks for a 4-shard field on 256 CUs,
//   0.03 of the HBM rate; the launch code splits until the grid holds ~2048 blocks).
// ---- the 64 x 64 bit-matrix transpose of a wavefront, in registers (round 6) ----------------------------------------------------
// Lane i holds row i (a u64: bit c = column c); afterwards lane j holds column j (bit i = row i).  Six index-bit swaps (lane bit s
// <-> bit-position bit s), none through the LDS: 32 = ONE v_permlane32_swap between the two halves of the u64; 16 = a
// v_permlane16_swap of the dword and its upper half + one v_perm; 8 = DPP row_ror:8 + v_perm; 4 = two bank-masked DPP row shifts +
// rotate + v_bfi; 2, 1 = DPP quad_perm + rotate + v_bfi: 31 vector instructions per word.  (Until round 5 every stage was two
// ds_bpermute through __shfl_xor and ~8 vector instructions: 12 LDS operations and ~50 vector instructions per word, each stage a
// full LDS round trip behind the one before.)
struct TrConst {
  uint32_t sel8, rot4, m4, rot2, m2, rot1, m1;
};
__device__ __forceinline__ TrConst tr_const(int lane) {
  TrConst c;
  c.sel8 = (lane & 8) ? 0x03070105u : 0x06020400u;
  c.rot4 = (lane & 4) ? 4u : 28u;
  c.m4 = (lane & 4) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
  c.rot2 = (lane & 2) ? 2u : 30u;
  c.m2 = (lane & 2) ? 0xCCCCCCCCu : 0x33333333u;
  c.rot1 = (lane & 1) ? 1u : 31u;
  c.m1 = (lane & 1) ? 0xAAAAAAAAu : 0x55555555u;
  return c;
}
__device__ __forceinline__ uint32_t tr_low_stages(uint32_t d, const TrConst& c) {
  // 16: the dword's halves change places between the lane rows r and r ^ 1
  {
    auto r = __builtin_amdgcn_permlane16_swap(d, d >> 16, false, false);  // r[0] (odd rows) <- the partner's upper half; r[1] (even rows) <- the partner's dword
    d = __builtin_amdgcn_perm(r[1], r[0], 0x05040100u);                   // {r[1].lo16 : r[0].lo16}
  }
  // 8: bytes, partner = lane ^ 8 (row_ror:8 inside the 16-lane row)
  {
    const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x128, 0xF, 0xF, true);
    d = __builtin_amdgcn_perm(x, d, c.sel8);
  }
  // 4: nibbles, partner = lane ^ 4 (lanes of banks 0 / 2 read four lanes up, banks 1 / 3 four lanes down)
  {
    uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x104, 0xF, 0xF, true);  // (every lane written: no initialisation; banks 1 / 3 are replaced next)
    x = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)d, 0x114, 0xF, 0xA, false);
    const uint32_t t = __builtin_amdgcn_alignbit(x, x, c.rot4);
    d = (d & c.m4) | (t & ~c.m4);
  }
  // 2: bit pairs, partner = lane ^ 2
  {
    const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x4E, 0xF, 0xF, true);
    const uint32_t t = __builtin_amdgcn_alignbit(x, x, c.rot2);
    d = (d & c.m2) | (t & ~c.m2);
  }
  // 1: bits, partner = lane ^ 1
  {
    const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0xB1, 0xF, 0xF, true);
    const uint32_t t = __builtin_amdgcn_alignbit(x, x, c.rot1);
    d = (d & c.m1) | (t & ~c.m1);
  }
  return d;
}
__device__ __forceinline__ u64 wave_transpose64(u64 v, const TrConst& c) {
  // 32: lanes 0..31 hand their upper dword to lanes 32..63 and get those lanes' lower dword
  auto r = __builtin_amdgcn_permlane32_swap((uint32_t)v, (uint32_t)(v >> 32), false, false);
  const uint32_t lo = tr_low_stages(r[0], c), hi = tr_low_stages(r[1], c);
  return ((u64)hi << 32) | lo;
}

// columns of exists ∩ filter per (shard, slot) cell: one wavefront per cell (dense rows); the Distinct with a filter needs them
// before it can say where a cell's values go (without a filter the stored cardinalities of the exists row are the counts)
__global__ void __launch_bounds__(256) k_bsi_cell_counts(const uint8_t* __restrict__ arena, const uint32_t* __restrict__ rows, uint32_t n_shards,
                                                        const uint8_t* __restrict__ farena, const uint32_t* __restrict__ frows,
                                                        uint32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= n_shards * kSlots) return;
  const uint32_t shard = cell >> 4, slot = cell & 15;
  const uint64_t rowBytes = (uint64_t)kSlots * 8192;
  const ulonglong2* e = reinterpret_cast<const ulonglong2*>(arena + (uint64_t)rows[shard] * rowBytes + slot * 8192ull);
  const ulonglong2* f = reinterpret_cast<const ulonglong2*>(farena + (uint64_t)frows[shard] * rowBytes + slot * 8192ull);
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const ulonglong2 a = e[j * kWave + lane], b = f[j * kWave + lane];
    c += (uint32_t)__popcll(a.x & b.x) + (uint32_t)__popcll(a.y & b.y);
  }
  c = wave_reduce_add(c);
  if (lane == 0) counts[cell] = c;
}

// cell_base[0 .. n_cells] = *carry + the exclusive prefix of the cells' counts (counts[] if given, else the stored cardinality of
// the exists row's container); *carry moves on by the total.  One block.
__global__ void __launch_bounds__(1024) k_bsi_cell_scan(const Slot* __restrict__ slots, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ counts,
                                                       uint32_t n_cells, u64* __restrict__ cell_base, u64* __restrict__ carry) {
  __shared__ u64 wsum[16];
  __shared__ u64 run;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) run = *carry;
  __syncthreads();
  for (uint32_t c0 = 0; c0 < n_cells; c0 += 1024) {
    const uint32_t c = c0 + (uint32_t)t;
    u64 v = 0;
    if (c < n_cells) v = counts ? (u64)counts[c] : (u64)slot_n(slots[(uint64_t)rows[c >> 4] * kSlots + (c & 15)]);
    u64 incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, o, kWave), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o, kWave);
      if (lane >= o) incl += ((u64)hi << 32) | lo;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u64 before = run;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    if (c < n_cells) cell_base[c] = before + incl - v;
    __syncthreads();
    if (t == 1023) run = before + incl;
    __syncthreads();
  }
  if (t == 0) {
    cell_base[n_cells] = run;
    *carry = run;
  }
}

constexpr int kBsiValStride = 144;  // bytes per plane row of the staging tile (128 + 16)
// Round 6.  `split` blocks per (shard, slot) cell, 16 / split rounds of 1024 columns per wave.  Where a value goes is ARITHMETIC:
// cell_base[cell] (k_bsi_cell_scan) + the columns of the cell's earlier units — parts before this one, then waves before this one,
// then rounds — which the block counts from the exists (∩ filter) words it reads anyway.  (Until round 5 every block reserved its
// space with an atomic on ONE global cursor: 1024 serialised L2 round trips for a 4-shard field, a third of the kernel's 35 us.)
// flag: set if a cell holds another number of columns than cell_base says (a stored cardinality that is wrong).
__global__ void __launch_bounds__(256) k_bsi_values(const uint8_t* __restrict__ arena, const uint32_t* __restrict__ rows,
                                                   uint32_t n_shards, uint32_t depth, const uint8_t* __restrict__ farena,
                                                   const uint32_t* __restrict__ frows, long long* __restrict__ out,
                                                   u64 out_cap, const u64* __restrict__ cell_base, uint32_t* __restrict__ flag, uint32_t split) {
  __shared__ ulonglong2 stage[4][64 * kBsiValStride / 16];  // 4 x 9216 bytes: one 64-plane x 128-byte tile per wave
  __shared__ uint32_t wtot[4], wpre[4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t part = blockIdx.x % split, cell = blockIdx.x / split;  // split: 1, 2, 4, 8 or 16
  const uint32_t shard = cell >> 4, slot = cell & 15;
  if (shard >= n_shards) return;
  const uint32_t R = 16u / split;  // rounds per wave
  const uint64_t rowBytes = (uint64_t)kSlots * 8192;
  const uint8_t* ex = arena + (uint64_t)rows[shard] * rowBytes + slot * 8192ull;
  const uint8_t* sg = ex + rowBytes;
  const uint8_t* fl = farena ? farena + (uint64_t)frows[shard] * rowBytes + slot * 8192ull : nullptr;
  const u64* exw = reinterpret_cast<const u64*>(ex);
  const u64* flw = reinterpret_cast<const u64*>(fl);
  const u64* sgw = reinterpret_cast<const u64*>(sg);
  const u64 cbase = cell_base[cell], cnext = cell_base[cell + 1];
  // columns of this wave's own rounds and of the cell's earlier parts in this wave's 256-word segment
  const uint32_t wave_w0 = (uint32_t)wv * 256u + part * R * 16u;
  // The planes of round 0 are REQUESTED here, before the exists words are even counted (a round without a column has loaded them
  // for nothing: rare, a BSI's exists row is dense): the two memory round trips of a one-round wave — a small field — overlap.
  // Instruction i brings the lines of planes 8 i .. 8 i + 7, eight lanes per 128-byte line.  Later rounds are requested at the END
  // of the round before (a set held across the sixteen transposes would cost 32 registers and an occupancy step); !live — behind
  // a wave's last round — every lane re-reads one line, so that the set is REDEFINED on every path and is not kept allocated.
  ulonglong2 pl[8];
  auto request = [&](uint32_t w0, bool live) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int plane = 8 * i + (lane >> 3), piece = lane & 7;
      const uint8_t* a = live && plane < (int)depth ? ex + (uint64_t)(2 + plane) * rowBytes + (uint64_t)w0 * 8 + (uint32_t)piece * 16u : ex;
      pl[i] = ld_stream(reinterpret_cast<const ulonglong2*>(a));  // (planes past the depth are zeroed where the set is USED: a select here would wait for the load)
    }
  };
  request(wave_w0, true);
  uint32_t mine = 0, pre = 0;
  for (uint32_t w = (uint32_t)lane; w < R * 16u; w += kWave) {
    u64 x = exw[wave_w0 + w];
    if (fl) x &= flw[wave_w0 + w];
    mine += (uint32_t)__popcll(x);
  }
  for (uint32_t w = (uint32_t)lane; w < part * R * 16u; w += kWave) {
    u64 x = exw[(uint32_t)wv * 256u + w];
    if (fl) x &= flw[(uint32_t)wv * 256u + w];
    pre += (uint32_t)__popcll(x);
  }
  pre = wave_reduce_add(pre);
  mine = wave_reduce_add(mine);
  if (lane == 0) {
    wpre[wv] = pre;
    wtot[wv] = mine;
  }
  __syncthreads();
  u64 wave_pos = cbase + wpre[0] + wpre[1] + wpre[2] + wpre[3];
  for (int w = 0; w < wv; ++w) wave_pos += wtot[w];
  if (part == split - 1 && threadIdx.x == 0) {
    const u64 total = (u64)wpre[0] + wpre[1] + wpre[2] + wpre[3] + wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (total != cnext - cbase) atomicOr(flag, 1u);
  }
  if (mine == 0) return;  // (wave-uniform; no barrier follows)
  const TrConst tc = tr_const(lane);
  const uint32_t bit_lo = lane < 32 ? 1u << lane : 0u, bit_hi = lane < 32 ? 0u : 1u << (lane - 32);
  uint8_t* stg = reinterpret_cast<uint8_t*>(&stage[wv][0]);
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t w0 = wave_w0 + r * 16u;  // first word of this round
    // the round's sixteen exists (∩ filter) and sign words: wave-uniform addresses, i.e. SCALAR loads — the per-word masks, their
    // popcounts and the running position stay in scalar registers (until round 6 lanes 0..15 held them and every word cost five
    // v_readlane and a lane-wise prefix)
    u64 ew[16];
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      ew[k] = exw[w0 + k];
      if (fl) ew[k] &= flw[w0 + k];
      total += (uint32_t)__popcll(ew[k]);
    }
    // this lane's plane: 16 words = one 128-byte line, handed to lane `plane` through the wave's LDS staging (row stride 144 B: the
    // eight lanes of a 16-byte-per-lane access fall into eight different bank groups both ways)
    u64 pw[16];
    {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int plane = 8 * i + (lane >> 3), piece = lane & 7;
        ulonglong2 v = pl[i];
        if (plane >= (int)depth) v.x = v.y = 0;
        *reinterpret_cast<ulonglong2*>(stg + plane * kBsiValStride + piece * 16) = v;
      }
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(stg + lane * kBsiValStride + k * 16);
        pw[2 * k] = v.x;
        pw[2 * k + 1] = v.y;
      }
      wave_lds_sync();  // (the next round writes the staging again)
    }
    if (total != 0) {  // (wave-uniform; 0: no column of these 1024 has a value)
    u64 pos = wave_pos;
    wave_pos += total;
#pragma unroll
    for (int k = 0; k < 16; ++k) pw[k] = wave_transpose64(pw[k], tc);  // sixteen independent transposes: straight-line code the scheduler interleaves
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const u64 v = pw[k];
      const u64 swk = sgw[w0 + k];  // (a scalar load per word, next to its use: all sixteen held from the top of the round overflow the scalar file)
      const uint32_t elo = (uint32_t)ew[k], ehi = (uint32_t)(ew[k] >> 32), slo = (uint32_t)swk, shi = (uint32_t)(swk >> 32);
      if (((elo & bit_lo) | (ehi & bit_hi)) != 0u) {
        // value *= -1 for negative columns (int64 wrap-around as in the reference, executor.go:2123)
        const long long val = (((slo & bit_lo) | (shi & bit_hi)) != 0u) ? (long long)(0ull - v) : (long long)v;
        const u64 at = pos + __builtin_amdgcn_mbcnt_hi(ehi, __builtin_amdgcn_mbcnt_lo(elo, 0u));  // + the existing columns below this lane
        if (at < out_cap && at < cnext) out[at] = val;
      }
      pos += (uint32_t)__popcll(ew[k]);
    }
    }
    request(w0 + 16u, r + 1 < R);
  }
}

}  // namespace fbk
