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

constexpr int kPairBatch = 8;  // elements per lane and operand in flight (8 x 64 = 512 values / runs per batch)

// batch `base / 512` of a sparse container, lane-consecutive: v[k] = element base + 64 k + lane
// (array: the uint16 value; run: the {start, last} pair as one dword).  Lanes past the end hold junk.
__device__ __forceinline__ void sparse_load(uint32_t type, const uint8_t* __restrict__ p, uint32_t len, uint32_t base, int lane,
                                            uint32_t (&v)[kPairBatch]) {
  if (type == kTypeArray) {
    const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) {
      const uint32_t i = base + (uint32_t)k * kWave + (uint32_t)lane;
      v[k] = i < len ? (uint32_t)q[i] : 0u;
    }
  } else {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) {
      const uint32_t i = base + (uint32_t)k * kWave + (uint32_t)lane;
      v[k] = i < len ? q[i] : 0u;
    }
  }
}

// XOR the raw bits of one batch into the table
__device__ __forceinline__ void sparse_xor_batch(uint32_t type, uint32_t* s32, uint32_t len, uint32_t base, int lane,
                                                 const uint32_t (&v)[kPairBatch]) {
  if (type == kTypeArray) {
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) {
      const uint32_t i = base + (uint32_t)k * kWave + (uint32_t)lane;
      if (i < len) atomicXor(&s32[v[k] >> 5], 1u << (v[k] & 31u));
    }
  } else {
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) {
      const uint32_t i = base + (uint32_t)k * kWave + (uint32_t)lane;
      if (i < len) {
        const uint32_t s = v[k] & 0xFFFFu, e = (v[k] >> 16) + 1u;
        atomicXor(&s32[s >> 5], 1u << (s & 31u));
        if (e < 65536u) atomicXor(&s32[e >> 5], 1u << (e & 31u));
      }
    }
  }
}

// the whole container, batch 0 already in v: the next batch is in flight while the current one is applied
__device__ __forceinline__ void sparse_xor_all(uint32_t type, const uint8_t* __restrict__ p, uint32_t len, int lane, uint32_t* s32,
                                               uint32_t (&v)[kPairBatch]) {
  for (uint32_t base = 0;;) {
    const uint32_t nb = base + kPairBatch * kWave;
    uint32_t nv[kPairBatch];
    if (nb < len) sparse_load(type, p, len, nb, lane, nv);
    sparse_xor_batch(type, s32, len, base, lane, v);
    if (nb >= len) break;
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = nv[k];
    base = nb;
  }
}

// number of this lane's array values that are set in the table (batch 0 already in v)
__device__ __forceinline__ uint32_t array_probe_all(const uint8_t* __restrict__ p, uint32_t len, int lane, const uint32_t* s32,
                                                    uint32_t (&v)[kPairBatch]) {
  uint32_t hits = 0;
  for (uint32_t base = 0;;) {
    const uint32_t nb = base + kPairBatch * kWave;
    uint32_t nv[kPairBatch];
    if (nb < len) sparse_load(kTypeArray, p, len, nb, lane, nv);
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) {
      const uint32_t i = base + (uint32_t)k * kWave + (uint32_t)lane;
      const uint32_t w = s32[v[k] >> 5];  // (junk lanes read word 0: in bounds)
      hits += (i < len) ? ((w >> (v[k] & 31u)) & 1u) : 0u;
    }
    if (nb >= len) break;
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = nv[k];
    base = nb;
  }
  return hits;
}

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

// Both containers of a (pair, slot) as register fragments with at most ONE clear of the wave's table.
__device__ __forceinline__ void frag_load_pair(const Slot& sa, const uint8_t* __restrict__ arenaA, const Slot& sb,
                                               const uint8_t* __restrict__ arenaB, int lane, u64* scratch, u64 (&wa)[kWordsPerLane],
                                               u64 (&wb)[kWordsPerLane]) {
  const uint32_t ta = slot_n(sa) ? slot_type(sa) : kTypeNil, tb = slot_n(sb) ? slot_type(sb) : kTypeNil;
  const uint8_t* pa = arenaA + sa.off;
  const uint8_t* pb = arenaB + sb.off;
  const bool la = ta == kTypeArray || ta == kTypeRun, lb = tb == kTypeArray || tb == kTypeRun;  // wave-uniform
  uint32_t va[kPairBatch], vb[kPairBatch];
  // everything that comes from global memory first
  if (ta == kTypeBitmap) frag_load_bitmap(pa, lane, wa);
  if (tb == kTypeBitmap) frag_load_bitmap(pb, lane, wb);
  if (la) sparse_load(ta, pa, sa.len, 0, lane, va);
  if (lb) sparse_load(tb, pb, sb.len, 0, lane, vb);
  if (ta == kTypeNil) frag_zero(wa);
  if (tb == kTypeNil) frag_zero(wb);
  if (!la && !lb) return;
  uint32_t* s32 = reinterpret_cast<uint32_t*>(scratch);
  lds_zero(scratch, lane);
  wave_lds_sync();
  if (la) {
    sparse_xor_all(ta, pa, sa.len, lane, s32, va);
    wave_lds_sync();
    lds_read_frag(scratch, lane, wa);  // raw bits of A: array bits / run toggles
    wave_lds_sync();
  }
  if (lb) {
    sparse_xor_all(tb, pb, sb.len, lane, s32, vb);  // on top of A's raw bits
    wave_lds_sync();
    lds_read_frag(scratch, lane, wb);
    wave_lds_sync();
    if (la) {
#pragma unroll
      for (int i = 0; i < kWordsPerLane; ++i) wb[i] ^= wa[i];
    }
  }
  if (ta == kTypeRun) frag_parity_prefix(wa, lane);
  if (tb == kTypeRun) frag_parity_prefix(wb, lane);
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
__device__ __forceinline__ void item_prefetch(const Slot& sa, const uint8_t* __restrict__ arenaA, const Slot& sb,
                                              const uint8_t* __restrict__ arenaB, int lane, uint32_t (&va)[kPairBatch],
                                              uint32_t (&vb)[kPairBatch]) {
  const uint32_t na = slot_n(sa), nb = slot_n(sb);
  if (na == 0 || nb == 0 || na == 65536u || nb == 65536u) return;
  const uint32_t ta = slot_type(sa), tb = slot_type(sb);
  if (ta == kTypeArray || ta == kTypeRun) sparse_load(ta, arenaA + sa.off, sa.len, 0, lane, va);
  if (tb == kTypeArray || tb == kTypeRun) sparse_load(tb, arenaB + sb.off, sb.len, 0, lane, vb);
}

// shorter array -> table, longer array probes it; returns this lane's hits
__device__ __forceinline__ uint32_t arrays_table_probe(const uint8_t* __restrict__ pt, uint32_t lt, uint32_t (&vt)[kPairBatch],
                                                       const uint8_t* __restrict__ pp, uint32_t lp, uint32_t (&vp)[kPairBatch], int lane,
                                                       u64* table) {
  uint32_t* s32 = reinterpret_cast<uint32_t*>(table);
  lds_zero(table, lane);
  wave_lds_sync();
  sparse_xor_all(kTypeArray, pt, lt, lane, s32, vt);
  wave_lds_sync();
  const uint32_t h = array_probe_all(pp, lp, lane, s32, vp);
  wave_lds_sync();
  return h;
}

// bitmap -> table (no clear), array probes it; returns this lane's hits
__device__ __forceinline__ uint32_t bitmap_table_probe(const uint8_t* __restrict__ pbm, const uint8_t* __restrict__ parr, uint32_t larr,
                                                       uint32_t (&vp)[kPairBatch], int lane, u64* table) {
  u64 wb[kWordsPerLane];
  frag_load_bitmap(pbm, lane, wb);
  ulonglong2* q = reinterpret_cast<ulonglong2*>(table);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 x;
    x.x = wb[2 * j];
    x.y = wb[2 * j + 1];
    q[j * kWave + lane] = x;  // fragment layout -> natural word order in the table
  }
  wave_lds_sync();
  const uint32_t h = array_probe_all(parr, larr, lane, reinterpret_cast<const uint32_t*>(table), vp);
  wave_lds_sync();
  return h;
}

// values (<= 128, in v[0..1]) of an array that are set in a bitmap container in global memory
__device__ __forceinline__ uint32_t array_probe_global(const uint32_t (&v)[kPairBatch], uint32_t len, const uint8_t* __restrict__ pbm, int lane) {
  uint32_t h = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t i = (uint32_t)k * kWave + (uint32_t)lane;
    if (i < len) h += (reinterpret_cast<const uint32_t*>(pbm)[v[k] >> 5] >> (v[k] & 31u)) & 1u;
  }
  return h;
}

// one (pair, slot): adds to `part` (per lane) or `spart` (wave-uniform)
__device__ __forceinline__ void icount_item(const Slot& sa, const uint8_t* __restrict__ arenaA, const Slot& sb,
                                            const uint8_t* __restrict__ arenaB, int lane, u64* table, uint32_t (&va)[kPairBatch],
                                            uint32_t (&vb)[kPairBatch], uint32_t sparse_paths, uint32_t& part, uint32_t& spart) {
  const uint32_t na = slot_n(sa), nb = slot_n(sb);
  const uint32_t ta = slot_type(sa), tb = slot_type(sb);
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
    if (sparse_paths && sa.len <= kSmallArray && sb.len <= kSmallArray) {
      // all-pairs compare in registers, the shorter array broadcast value by value
      const uint32_t a = (uint32_t)lane < sa.len ? va[0] : 0xFFFFFFFFu;
      const uint32_t b = (uint32_t)lane < sb.len ? vb[0] : 0xFFFFFFFEu;
      const bool a_long = sa.len >= sb.len;
      const uint32_t lng = a_long ? a : b, sht = a_long ? b : a;
      const uint32_t ns = a_long ? sb.len : sa.len;
      bool m = false;
      for (uint32_t k = 0; k < ns; ++k) m |= lng == (uint32_t)__builtin_amdgcn_readlane((int)sht, (int)k);
      part += m ? 1u : 0u;
    } else if (sa.len <= sb.len) {
      part += arrays_table_probe(pa, sa.len, va, pb, sb.len, vb, lane, table);
    } else {
      part += arrays_table_probe(pb, sb.len, vb, pa, sa.len, va, lane, table);
    }
  } else if (ta == kTypeArray && tb == kTypeBitmap) {
    if (sparse_paths && sa.len <= kProbeArray) part += array_probe_global(va, sa.len, pb, lane);
    else part += bitmap_table_probe(pb, pa, sa.len, va, lane, table);
  } else if (ta == kTypeBitmap && tb == kTypeArray) {
    if (sparse_paths && sb.len <= kProbeArray) part += array_probe_global(vb, sb.len, pa, lane);
    else part += bitmap_table_probe(pa, pb, sb.len, vb, lane, table);
  } else {
    // a run on at least one side: both operands as fragments, one clear (frag_load_pair with batch 0 in hand)
    u64 wa[kWordsPerLane], wb[kWordsPerLane];
    const bool la = ta != kTypeBitmap, lb = tb != kTypeBitmap;
    if (!la) frag_load_bitmap(pa, lane, wa);
    if (!lb) frag_load_bitmap(pb, lane, wb);
    uint32_t* s32 = reinterpret_cast<uint32_t*>(table);
    lds_zero(table, lane);
    wave_lds_sync();
    if (la) {
      sparse_xor_all(ta, pa, sa.len, lane, s32, va);
      wave_lds_sync();
      lds_read_frag(table, lane, wa);
      wave_lds_sync();
    }
    if (lb) {
      sparse_xor_all(tb, pb, sb.len, lane, s32, vb);
      wave_lds_sync();
      lds_read_frag(table, lane, wb);
      wave_lds_sync();
      if (la) {
#pragma unroll
        for (int i = 0; i < kWordsPerLane; ++i) wb[i] ^= wa[i];
      }
    }
    if (ta == kTypeRun) frag_parity_prefix(wa, lane);
    if (tb == kTypeRun) frag_parity_prefix(wb, lane);
#pragma unroll
    for (int i = 0; i < kWordsPerLane; ++i) part += __popcll(wa[i] & wb[i]);
  }
}

template <int SPW>
__global__ void __launch_bounds__(256, 4) k_icount2(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                                   const uint32_t* __restrict__ rowsA, const Slot* __restrict__ slotsB,
                                                   const uint8_t* __restrict__ arenaB, const uint32_t* __restrict__ rowsB, uint64_t n_pairs,
                                                   u64* __restrict__ out, uint32_t sparse_paths) {
  __shared__ u64 lds[4][kWords];
  constexpr int kWavesPerPair = kSlots / SPW;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and the compiler knows it
  const uint64_t wid = (uint64_t)blockIdx.x * 4 + (uint64_t)wv;
  const uint64_t pair = wid / kWavesPerPair;
  if (pair >= n_pairs) return;
  const uint32_t slot0 = (uint32_t)(wid % kWavesPerPair) * SPW;
  const uint32_t ra = rowsA[pair], rb = rowsB[pair];  // both row indexes in flight together
  const Slot* da = slotsA + (uint64_t)ra * kSlots + slot0;
  const Slot* db = slotsB + (uint64_t)rb * kSlots + slot0;
  Slot sa[SPW], sb[SPW];
#pragma unroll
  for (int k = 0; k < SPW; ++k) {
    sa[k] = da[k];
    sb[k] = db[k];
  }
  u64* table = lds[wv];
  uint32_t part = 0, spart = 0;
  uint32_t va[kPairBatch], vb[kPairBatch];
  item_prefetch(sa[0], arenaA, sb[0], arenaB, lane, va, vb);
#pragma unroll
  for (int k = 0; k < SPW; ++k) {
    uint32_t xa[kPairBatch], xb[kPairBatch];
    if (k + 1 < SPW) item_prefetch(sa[k + 1], arenaA, sb[k + 1], arenaB, lane, xa, xb);
    icount_item(sa[k], arenaA, sb[k], arenaB, lane, table, va, vb, sparse_paths, part, spart);
    if (k + 1 < SPW) {
#pragma unroll
      for (int i = 0; i < kPairBatch; ++i) {
        va[i] = xa[i];
        vb[i] = xb[i];
      }
    }
  }
  const uint32_t c = wave_reduce_add(part) + spart;
  if (lane == 0 && c) atomicAdd(&out[pair], (u64)c);
}

// Materialising A <op> B, one wave per (pair, slot): k_setop's outputs and right-sized array paths
// (fbk_kernels.hip.h) behind the pair loader above.
template <int OP>
__global__ void __launch_bounds__(256, 4) k_setop2(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                               const uint32_t* __restrict__ rowsA, const Slot* __restrict__ slotsB,
                                               const uint8_t* __restrict__ arenaB, const uint32_t* __restrict__ rowsB, uint64_t n_pairs,
                                               uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                               u64* __restrict__ out_counts, uint32_t direct) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: descriptors become scalar loads (see k_icount2)
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + (uint64_t)wv;
  const uint64_t pair = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (pair >= n_pairs) return;
  const uint32_t ra = rowsA[pair], rb = rowsB[pair];
  const Slot sa = slotsA[(uint64_t)ra * kSlots + slot];
  const Slot sb = slotsB[(uint64_t)rb * kSlots + slot];
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
  if (OP == 0 && direct && outRuns) {
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
      if (mine) reinterpret_cast<uint16_t*>(arenaO + so.off)[__popcll(below)] = (uint16_t)v;
      const int prev = below ? 63 - __builtin_clzll(below) : 0;
      const uint32_t vprev = (uint32_t)__shfl((int)v, prev, kWave);
      const uint32_t r = (uint32_t)__popcll(__ballot(mine && (below == 0 || vprev + 1u != v)));
      const uint32_t c = (uint32_t)__popcll(mm);
      if (lane == 0) {
        so.len = c;
        so.tn = make_tn(c ? kTypeArray : kTypeNil, c);
        outSlots[wslot] = so;
        outRuns[wslot] = r;
        if (out_counts && c) atomicAdd(&out_counts[pair], (u64)c);
      }
      return;
    }
  }
  u64 wa[kWordsPerLane], wb[kWordsPerLane];
  frag_load_pair(sa, arenaA, sb, arenaB, lane, lds[wv], wa, wb);
#pragma unroll
  for (int i = 0; i < kWordsPerLane; ++i) wa[i] = apply_op<OP>(wa[i], wb[i]);
  uint32_t c = wave_reduce_add(frag_popcount(wa));
  uint32_t r = 0;
  if (outRuns) r = wave_reduce_add(frag_count_runs(wa, lane));
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
  } else {
    frag_store_bitmap(arenaO + so.off, lane, wa);
  }
  if (lane == 0) {
    if (as_array) so.len = c;
    so.tn = make_tn(c ? (as_array ? kTypeArray : kTypeBitmap) : kTypeNil, c);
    outSlots[wslot] = so;
    if (outRuns) outRuns[wslot] = r;
    if (out_counts && c) atomicAdd(&out_counts[pair], (u64)c);
  }
}

}  // namespace fbk
