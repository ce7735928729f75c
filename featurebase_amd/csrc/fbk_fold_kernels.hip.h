// fbk_fold_kernels.hip.h — n-way Union / Xor / Difference of k rows with a SHARED LDS
// accumulator (the n-way Intersect stays on k_fold_n<0>, fbk_query_kernels.hip.h).
//
// One 256-thread block per (group, slot).  The k containers of the slot are dealt to the 4
// wavefronts (wave w takes rows w, w+4, ...); every container is OR-ed (XOR-ed) straight into
// ONE 8 KiB LDS bitmap, in its encoded form:
//   array   one ds_or_b32 per value                          (arrayToBitmap, roaring.go:3756)
//   run     masked ds_or_b32 on the two boundary dwords, whole dwords in between; long runs are
//           filled by all 64 lanes together                  (splatRun, container_stash.go:696-729)
//   bitmap  ds_or_b64 of the words each lane fetched
// so there is no per-container zero / decode / read-back of a scratch bitmap, no register
// accumulator and no cross-wave combine.
//
// The payloads reach the wave as ONE sequence of 1 KiB chunks (64 lanes x 16 bytes) through a
// ring of 6 chunks = 24 registers: 57 VGPRs in all, 8 wavefronts per SIMD.  History, 128 shards
// x 64 mixed rows (291 MB of payload):
//   296 us  decode every container into a per-wave scratch, accumulate in registers (256 VGPRs)
//   107 us  direct scatter, ring of 3 whole containers (96 registers of which 27 % held payload:
//           the average container is 2.2 KB) — 4 waves per SIMD; making that kernel branch-free or
//           a fifth of its code size changed nothing, dummy ORs for the lanes past the end of an
//           array made it 40 % slower (the LDS serialises same-address atomics)
//    70 us  this version (4.2 TB/s; 4.8 TB/s at 384 shards).  Ring depths 4 .. 7 measure the
//           same, 8 .. 16 are slower: occupancy, not bytes in flight per wave, is what counts.
// scripts/rand_read.hip: HBM delivers 6.1 TB/s for random 2 KiB reads, so the access pattern is
// not the limit; what remains is the per-block chain rows -> descriptors -> payload at the start.
//   OP 1 (OR)      r0 | r1 | ...               roaring.go:1455-1560, filter.go:327-334
//   OP 2 (XOR)     r0 ^ r1 ^ ...               executor.go:5513-5552 (values of one array / the
//                  runs of one container are disjoint, so XOR-ing them in one by one is exact)
//   OP 3 (ANDNOT)  r0 & ~(r1 | r2 | ...)       executor.go:2950-2983, roaring.go:1564-1595
#pragma once
#include "fbk_query_kernels.hip.h"

namespace fbk {

template <int OP>
__device__ __forceinline__ void lds_acc32(uint32_t* p, uint32_t m) {
  if (OP == 2) atomicXor(p, m);
  else atomicOr(p, m);
}

// dwords [ws+1, we) of the accumulator become all ones (the interior of a run).  OR / ANDNOT: a
// plain store — whatever other waves OR into the dword concurrently, all ones is the result;
// XOR has to flip.
template <int OP>
__device__ __forceinline__ void lds_fill32(uint32_t* p) {
  if (OP == 2) atomicXor(p, ~0u);
  else *(__attribute__((address_space(3))) uint32_t*)p = ~0u;  // explicit LDS store (a volatile generic store is a flat_store + vmcnt(0))
}

// row j (64 x 16 bytes) of a prefetched payload; j is wave-uniform, so this is a scalar branch
// tree around four v_mov — it lets the loops over rows below stay rolled (the fully unrolled
// version of this file was 16 000 instructions long, three copies of every path)
// One 1 KiB chunk (64 lanes x 16 bytes = d[0..3] per lane) of a payload: row j of container
// (type, len).  Everything but d and lane is wave-uniform.
template <int OP>
__device__ __forceinline__ void scatter_chunk(const uint32_t (&d)[4], uint32_t type, uint32_t len, uint32_t j, int lane,
                                              uint32_t* acc32) {
  if (type == kTypeBitmap) {
    u64* acc64 = reinterpret_cast<u64*>(acc32);
    const uint32_t w = (j * kWave + lane) * 2;
    const u64 x = ((u64)d[1] << 32) | d[0], y = ((u64)d[3] << 32) | d[2];
    if (OP == 2) {
      atomicXor(&acc64[w], x);
      atomicXor(&acc64[w + 1], y);
    } else {
      atomicOr(&acc64[w], x);
      atomicOr(&acc64[w + 1], y);
    }
    return;
  }
  if (type == kTypeArray) {
    const uint32_t row0 = j * (kWave * 8u);  // first element of this row of 16-byte chunks
    if (row0 + kWave * 8u <= len) {          // scalar: all 512 values of the row exist, no predicate at all
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        lds_acc32<OP>(&acc32[(d[q] >> 5) & 0x7FFu], 1u << (d[q] & 31));
        lds_acc32<OP>(&acc32[d[q] >> 21], 1u << ((d[q] >> 16) & 31));
      }
    } else {
      // the ragged last row: lanes past the end sit out as a whole (ONE change of EXEC; a dummy OR would not
      // be free — the LDS serialises same-address atomics, measured).  The one lane that holds the end of the
      // array ORs zero masks for the slots past it (those hold real values: whatever follows the payload)
      const uint32_t e0 = row0 + lane * 8u;
      if (e0 < len) {
        const uint32_t valid = (1u << min(len - e0, 8u)) - 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          lds_acc32<OP>(&acc32[(d[q] >> 5) & 0x7FFu], __builtin_amdgcn_ubfe(valid, 2u * q, 1u) << (d[q] & 31));
          lds_acc32<OP>(&acc32[d[q] >> 21], __builtin_amdgcn_ubfe(valid, 2u * q + 1u, 1u) << ((d[q] >> 16) & 31));
        }
      }
    }
    return;
  }
  // runs: 4 intervals {start u16, last u16} per 16-byte chunk
  const uint32_t i0 = (j * kWave + lane) * 4u;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool on = i0 + t < len;
    const uint32_t s = d[t] & 0xFFFFu, l = d[t] >> 16;
    const uint32_t ws = s >> 5, we = l >> 5;
    const uint32_t ms = ~0u << (s & 31), ml = ~0u >> (31 - (l & 31));
    if (on) {
      if (ws == we) {
        lds_acc32<OP>(&acc32[ws], ms & ml);
      } else {
        lds_acc32<OP>(&acc32[ws], ms);
        lds_acc32<OP>(&acc32[we], ml);
      }
    }
    const uint32_t inner = (on && we > ws + 1) ? we - ws - 1 : 0;
    // short interiors: the owning lane; long ones (> 4 dwords): all 64 lanes together
    if (inner && inner <= 4)
      for (uint32_t q = ws + 1; q < we; ++q) lds_fill32<OP>(&acc32[q]);
    u64 longm = __ballot(inner > 4);
    while (longm) {
      const int src = __builtin_ctzll(longm);  // scalar
      longm &= longm - 1;
      const uint32_t bs = __builtin_amdgcn_readlane(ws, src), be = __builtin_amdgcn_readlane(we, src);
      for (uint32_t q = bs + 1 + lane; q < be; q += kWave) lds_fill32<OP>(&acc32[q]);
    }
  }
}

// a payload larger than 8 KiB (arrays > 4096 values, > 2048 runs: legal, outside roaring policy)
template <int OP>
__device__ __forceinline__ void scatter_big(const uint8_t* __restrict__ p, uint32_t type, uint32_t len, int lane,
                                            uint32_t* acc32) {
  if (type == kTypeArray) {
    const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
    for (uint32_t i = lane; i < len; i += kWave) {
      const uint32_t a = q[i];
      lds_acc32<OP>(&acc32[a >> 5], 1u << (a & 31));
    }
  } else {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    for (uint32_t i = lane; i < len; i += kWave) {
      const uint32_t iv = q[i], s = iv & 0xFFFFu, l = iv >> 16;
      const uint32_t ws = s >> 5, we = l >> 5;
      const uint32_t ms = ~0u << (s & 31), ml = ~0u >> (31 - (l & 31));
      if (ws == we) {
        lds_acc32<OP>(&acc32[ws], ms & ml);
      } else {
        lds_acc32<OP>(&acc32[ws], ms);
        lds_acc32<OP>(&acc32[we], ml);
        for (uint32_t d = ws + 1; d < we; ++d) lds_acc32<OP>(&acc32[d], ~0u);
      }
    }
  }
}

// WRITE: 0 counts only; 1 the result as an 8 KiB bitmap cell (+ its run count for a later optimize pass);
// 2 Container.optimize() in the epilogue (roaring.go:3412-3461): the block knows N and the run count of its
// result, picks the encoding by optimize()'s rule, stages arrays / run lists in the (now free) LDS accumulator
// and writes exactly the encoded bytes into the head of the cell — no second pass over the cells, no scan, no
// compaction, no host round trip (the separate re-encode pass cost 150 of the 270 us of a materialised
// Union-of-64 + optimize() on 256 shards).  Cells stay at their 8 KiB stride in the output arena.
template <int OP, int WRITE>
__global__ void __launch_bounds__(256, 8) k_fold_scatter(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                     const uint32_t* __restrict__ rows, uint64_t n_groups, uint32_t k,
                                                     const Slot* __restrict__ fslots, const uint8_t* __restrict__ farena,
                                                     const uint32_t* __restrict__ frows, uint8_t* __restrict__ arenaO,
                                                     Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                                     u64* __restrict__ out_counts, const Slot* __restrict__ recs) {
  static_assert(OP == 1 || OP == 2 || OP == 3, "scatter fold handles OR, XOR and ANDNOT");
  __shared__ u64 acc[kWords];  // the union / xor of the group's containers at this slot
  __shared__ u64 aux[kWords];  // decode scratch for the filter row and for r0 of a difference
  __shared__ uint32_t s_short;
  __shared__ uint32_t s_cnt[3][4];
  __shared__ uint32_t s_scan[2][4];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint64_t cell = xcd_swizzle(blockIdx.x, gridDim.x);  // (group, slot): the 16 slots of a group on one XCD
  const uint64_t g = cell >> 4;
  const uint32_t slot = cell & 15;
  if (g >= n_groups) return;  // block-uniform
  uint32_t* acc32 = reinterpret_cast<uint32_t*>(acc);
  {
    ulonglong2 z;
    z.x = z.y = 0;
    reinterpret_cast<ulonglong2*>(acc)[t] = z;
    reinterpret_cast<ulonglong2*>(acc)[256 + t] = z;
    if (t == 0) s_short = 0;
  }
  __syncthreads();
  const uint32_t* grow = rows + g * k;
  // a prepared query's resolved row records (k_resolve_rows): the k descriptors of this (group, slot) side by side — one
  // coalesced load instead of row index -> descriptor gathered from k rows
  const Slot* grec = recs ? recs + cell * (uint64_t)k : nullptr;
  // The filter container (and r0 of a difference) is needed only in the epilogue, but its
  // address hangs off a chain of dependent loads (row index -> descriptor -> payload): start
  // it now, in the block layout (thread t: chunks t and t+256), so it has landed by then.
  Slot sf;
  sf.off = 0;
  sf.len = 0;
  sf.tn = 0;
  u64 pf[4] = {0, 0, 0, 0};
  bool f_ready = false;
  if (fslots) {
    sf = fslots[(uint64_t)frows[g] * kSlots + slot];
    if (slot_n(sf) != 0 && slot_type(sf) == kTypeBitmap) {
      const ulonglong2* q = reinterpret_cast<const ulonglong2*>(farena + sf.off);
      const ulonglong2 v0 = ld_stream(&q[t]), v1 = ld_stream(&q[256 + t]);
      pf[0] = v0.x;
      pf[1] = v0.y;
      pf[2] = v1.x;
      pf[3] = v1.y;
      f_ready = true;
    }
  }
  Slot s0;
  s0.off = 0;
  s0.len = 0;
  s0.tn = 0;
  u64 p0[4] = {0, 0, 0, 0};
  bool r0_ready = false;
  if (OP == 3) {
    s0 = grec ? grec[0] : slots[(uint64_t)grow[0] * kSlots + slot];
    if (slot_n(s0) != 0 && slot_type(s0) == kTypeBitmap) {
      const ulonglong2* q = reinterpret_cast<const ulonglong2*>(arena + s0.off);
      const ulonglong2 v0 = ld_stream(&q[t]), v1 = ld_stream(&q[256 + t]);
      p0[0] = v0.x;
      p0[1] = v0.y;
      p0[2] = v1.x;
      p0[3] = v1.y;
      r0_ready = true;
    }
  }
  bool shortcut = false;  // OR / ANDNOT: a full operand saturates the accumulator
  for (uint32_t base = 0; base < k && !shortcut; base += 64) {
    Slot mine;  // lane l holds the descriptor of row base+l of the group
    mine.off = 0;
    mine.len = 0;
    mine.tn = 0;
    if (base + lane < k) mine = grec ? grec[base + lane] : slots[(uint64_t)grow[base + lane] * kSlots + slot];
    if (OP == 3 && base + lane == 0) mine.tn = 0;  // r0 is not one of the subtrahends
    const uint32_t cnt = min(64u, k - base);
    if (OP != 2 && __ballot(slot_n(mine) == 65536u) != 0) {
      shortcut = true;
      break;
    }
    // this wave's containers: i = wv, wv+4, ...
    auto meta = [&](uint32_t i, u64& off, uint32_t& len, uint32_t& tn) {
      // wave-uniform values: pull them into SGPRs so the type dispatch is scalar branching
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mine.off, (int)(i & 63));
      const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(mine.off >> 32), (int)(i & 63));
      off = ((u64)hi << 32) | lo;
      len = __builtin_amdgcn_readlane(mine.len, (int)(i & 63));
      tn = __builtin_amdgcn_readlane(mine.tn, (int)(i & 63));
    };
    // payloads beyond 8 KiB (arrays > 4096 values, > 2048 runs: legal, outside roaring policy) do
    // not fit the ring: they are scattered straight from global memory, dealt to the waves in turn
    {
      u64 bigm = __ballot((mine.tn & 0xFFFFFFu) != 0 && payload_bytes(mine.tn >> 24, mine.len) > 8192u);
      for (uint32_t ord = 0; bigm; ++ord) {
        const uint32_t i = (uint32_t)__builtin_ctzll(bigm);
        bigm &= bigm - 1;
        if ((ord & 3u) == (uint32_t)wv) {
          u64 off;
          uint32_t len, tn;
          meta(i, off, len, tn);
          scatter_big<OP>(arena + off, tn >> 24, len, lane, acc32);
        }
      }
    }
    {
      // ---- chunk-granular ring: the wave's containers as ONE sequence of 1 KiB chunks ----
      constexpr int NCH = 6;
      typedef uint32_t Chunk __attribute__((ext_vector_type(4)));
      Chunk C[NCH];
      // chunks of lane l's container (0: nil / empty / beyond 8 KiB, not part of the sequence)
      const uint32_t my_bytes = payload_bytes(mine.tn >> 24, mine.len);
      const uint32_t nr = ((mine.tn & 0xFFFFFFu) != 0 && my_bytes <= 8192u) ? (my_bytes + 1023u) >> 10 : 0u;
      auto next_valid = [&](uint32_t i) {
        while (i < cnt && __builtin_amdgcn_readlane(nr, (int)(i & 63)) == 0) i += 4;
        return i;
      };
      // producer (next chunk to load) and consumer (next chunk to scatter): position in the
      // sequence plus the descriptor of the current container, held in scalar registers and
      // refreshed only when the container changes (8 v_readlane per chunk otherwise)
      struct Cursor {
        uint32_t i, j, nr, len, tn, bytes;
        u64 off;
      };
      auto fetch = [&](Cursor& k) {
        k.j = 0;
        if (k.i < cnt) {
          meta(k.i, k.off, k.len, k.tn);
          k.bytes = payload_bytes(k.tn >> 24, k.len);
          k.nr = (k.bytes + 1023u) >> 10;
        }
      };
      auto advance = [&](Cursor& k) {
        if (++k.j == k.nr) {
          k.i = next_valid(k.i + 4);
          fetch(k);
        }
      };
      Cursor P, Q;
      P.i = next_valid(wv);
      P.nr = P.len = P.tn = P.bytes = 0;
      P.off = 0;
      fetch(P);
      Q = P;
      // Exactly ONE load instruction per step, written as asm, so that "the chunk issued NCH steps
      // ago has landed" is the constant s_waitcnt vmcnt(NCH - 1): left to the compiler, the
      // conditional loads of a rolled ring end in vmcnt(0) before every use (seen in the ISA), which
      // serialises the ring.  Lanes past the end of a payload re-read its first 16 bytes and an
      // exhausted producer reads the first 16 bytes of the arena: always a valid address, never a
      // change of EXEC around the load.
      const uint32_t lane16 = lane * 16u;
      auto load_chunk = [&](Chunk& c) {
        const uint8_t* p = arena;
        if (P.i < cnt) {
          const uint32_t b0 = P.j * 1024u + lane16;
          p = arena + P.off + (b0 < P.bytes ? b0 : 0u);
          advance(P);
        }
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(c) : "v"(p));
      };
#pragma unroll
      for (int q = 0; q < NCH; ++q) load_chunk(C[q]);
      while (Q.i < cnt) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
          // Every slot of a lap waits and reloads whether or not the sequence still has a chunk for it (an exhausted
          // producer reads the arena's first 16 bytes, see load_chunk): the number of loads behind a slot then does not
          // depend on the path, which is what the constant in the wait assumes — and what scripts/check_inflight.py, which
          // follows the control flow and not the values, can verify.  (With the whole slot behind "if (Q.i < cnt)" it has to
          // assume a slot running after an earlier one was skipped, one load short; at most NCH - 1 idle slots per wave.)
          asm volatile("s_waitcnt vmcnt(%1)" : "+v"(C[q]) : "n"(NCH - 1));
          if (Q.i < cnt) {
            const uint32_t d[4] = {C[q][0], C[q][1], C[q][2], C[q][3]};
            scatter_chunk<OP>(d, Q.tn >> 24, Q.len, Q.j, lane, acc32);
            advance(Q);
          }
          load_chunk(C[q]);
        }
      }
      // loads still in flight target registers the compiler is about to reuse
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (shortcut && lane == 0) s_short = 1;
  __syncthreads();
  const bool sat = s_short != 0;

  // ---- epilogue: thread t owns the 16-byte chunks t and t+256 of the result ----
  u64 w[4];
  {
    const ulonglong2 v0 = reinterpret_cast<const ulonglong2*>(acc)[t], v1 = reinterpret_cast<const ulonglong2*>(acc)[256 + t];
    w[0] = sat ? ~0ull : v0.x;
    w[1] = sat ? ~0ull : v0.y;
    w[2] = sat ? ~0ull : v1.x;
    w[3] = sat ? ~0ull : v1.y;
  }
  // a container in the block layout (bitmap: direct loads; array / run: wave 0 decodes into aux)
  auto load_block = [&](const Slot& s, const uint8_t* __restrict__ ar, u64 (&o)[4]) {
    if (slot_n(s) == 0) {
      o[0] = o[1] = o[2] = o[3] = 0;
      return;
    }
    if (slot_type(s) == kTypeBitmap) {
      const ulonglong2* q = reinterpret_cast<const ulonglong2*>(ar + s.off);
      const ulonglong2 v0 = ld_stream(&q[t]), v1 = ld_stream(&q[256 + t]);
      o[0] = v0.x;
      o[1] = v0.y;
      o[2] = v1.x;
      o[3] = v1.y;
      return;
    }
    __syncthreads();  // aux may still be read
    if (t < kWave) {
      u64 f[kWordsPerLane];
      frag_load(s, ar, t, aux, f);
      lds_write_frag(aux, t, f);
    }
    __syncthreads();
    const ulonglong2 v0 = reinterpret_cast<const ulonglong2*>(aux)[t], v1 = reinterpret_cast<const ulonglong2*>(aux)[256 + t];
    o[0] = v0.x;
    o[1] = v0.y;
    o[2] = v1.x;
    o[3] = v1.y;
  };
  if (OP == 3) {  // r0 \ (r1 | r2 | ...)
    if (!r0_ready) load_block(s0, arena, p0);
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = p0[q] & ~w[q];
  }
  uint32_t cu = __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);  // |result|
  uint32_t cf = cu;                                                                 // |result ∩ filter|
  if (fslots) {
    if (!f_ready) load_block(sf, farena, pf);
    cf = __popcll(w[0] & pf[0]) + __popcll(w[1] & pf[1]) + __popcll(w[2] & pf[2]) + __popcll(w[3] & pf[3]);
  }
  uint32_t rr = 0;
  if (WRITE == 2 || (WRITE && outRuns)) {
    // bitmapCountRuns (roaring.go:3372-3380): predecessor of a chunk's first bit is the top
    // bit of the previous chunk — publish the result and read the neighbour's last word
    __syncthreads();
    {
      ulonglong2 v0, v1;
      v0.x = w[0];
      v0.y = w[1];
      v1.x = w[2];
      v1.y = w[3];
      reinterpret_cast<ulonglong2*>(aux)[t] = v0;
      reinterpret_cast<ulonglong2*>(aux)[256 + t] = v1;
    }
    __syncthreads();
    const u64 l0 = t ? (aux[2 * t - 1] >> 63) : 0ull, l1 = aux[512 + 2 * t - 1] >> 63;
    rr = __popcll(w[0] & ~((w[0] << 1) | l0)) + __popcll(w[1] & ~((w[1] << 1) | (w[0] >> 63))) +
         __popcll(w[2] & ~((w[2] << 1) | l1)) + __popcll(w[3] & ~((w[3] << 1) | (w[2] >> 63)));
  }
  cu = wave_reduce_add(cu);
  cf = wave_reduce_add(cf);
  rr = wave_reduce_add(rr);
  if (lane == 0) {
    s_cnt[0][wv] = cu;
    s_cnt[1][wv] = cf;
    s_cnt[2][wv] = rr;
  }
  __syncthreads();
  const uint32_t tot_u = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
  const uint32_t tot_f = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
  if (WRITE == 2) {
    const uint32_t tot_r = s_cnt[2][0] + s_cnt[2][1] + s_cnt[2][2] + s_cnt[2][3];
    // optimize() (roaring.go:3412-3461): nil when empty; runs when runs <= 2048 and runs <= N / 2; an array when N < 4096; else the bitmap
    const uint32_t enc = tot_u == 0 ? kTypeNil : (tot_r <= 2048u && tot_r <= tot_u / 2u) ? kTypeRun : tot_u < 4096u ? kTypeArray : kTypeBitmap;
    Slot so;
    so.off = enc == kTypeNil ? 0ull : cell * 8192ull;
    so.len = 0;
    so.tn = make_tn(enc, tot_u);
    uint8_t* dst = arenaO + cell * 8192ull;
    if (enc == kTypeBitmap) {
      ulonglong2* q = reinterpret_cast<ulonglong2*>(dst);
      ulonglong2 v0, v1;
      v0.x = w[0];
      v0.y = w[1];
      v1.x = w[2];
      v1.y = w[3];
      st_stream(&q[t], v0);
      st_stream(&q[256 + t], v1);
      so.len = kWords;
    } else if (enc != kTypeNil) {
      // the encoded payload is staged in the accumulator (8 KiB: 4095 values or 2048 intervals fit; every thread read its
      // part of it before the barriers of the count reduction) in value order: the first halves of all threads (words
      // 2t, 2t+1), then the second halves (words 512+2t, 513+2t).  Counts travel as two 16-bit fields of one scan.
      uint16_t* st16 = reinterpret_cast<uint16_t*>(acc);
      const uint32_t v0 = 128u * (uint32_t)t, v1 = 32768u + 128u * (uint32_t)t;  // value of bit 0 of w[0] / w[2]
      u64 a0 = w[0], a1 = w[1], a2 = w[2], a3 = w[3];  // array: the values themselves; runs: the run starts
      u64 e0 = 0, e1 = 0, e2 = 0, e3 = 0;              // runs: the run ends
      if (enc == kTypeRun) {
        // (aux holds the published result: the neighbours' boundary bits)
        const u64 l0 = t ? (aux[2 * t - 1] >> 63) : 0ull, l1 = aux[512 + 2 * t - 1] >> 63;
        const u64 n0 = aux[2 * t + 2] & 1ull, n1 = t < 255 ? (aux[512 + 2 * t + 2] & 1ull) : 0ull;
        a0 = w[0] & ~((w[0] << 1) | l0);
        a1 = w[1] & ~((w[1] << 1) | (w[0] >> 63));
        a2 = w[2] & ~((w[2] << 1) | l1);
        a3 = w[3] & ~((w[3] << 1) | (w[2] >> 63));
        e0 = w[0] & ~((w[0] >> 1) | ((w[1] & 1ull) << 63));
        e1 = w[1] & ~((w[1] >> 1) | (n0 << 63));
        e2 = w[2] & ~((w[2] >> 1) | ((w[3] & 1ull) << 63));
        e3 = w[3] & ~((w[3] >> 1) | (n1 << 63));
      }
      const uint32_t ca = (uint32_t)(__popcll(a0) + __popcll(a1)) | ((uint32_t)(__popcll(a2) + __popcll(a3)) << 16);
      const uint32_t ce = (uint32_t)(__popcll(e0) + __popcll(e1)) | ((uint32_t)(__popcll(e2) + __popcll(e3)) << 16);
      const uint32_t ia = wave_incl_scan(ca), ie = wave_incl_scan(ce);
      if (lane == 63) {
        s_scan[0][wv] = ia;
        s_scan[1][wv] = ie;
      }
      __syncthreads();
      uint32_t wa = 0, we = 0, ta = 0, te = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < wv) wa += s_scan[0][k], we += s_scan[1][k];
        ta += s_scan[0][k], te += s_scan[1][k];
      }
      const uint32_t xa = wa + ia - ca, xe = we + ie - ce;  // exclusive, per field
      uint32_t pa0 = xa & 0xFFFFu, pa1 = (ta & 0xFFFFu) + (xa >> 16);
      uint32_t pe0 = xe & 0xFFFFu, pe1 = (te & 0xFFFFu) + (xe >> 16);
      const uint32_t stride = enc == kTypeRun ? 2u : 1u;  // intervals: start at 2k, last at 2k + 1
      auto emit = [&](u64 bits, uint32_t base, uint32_t& at, uint32_t odd) {
        while (bits) {
          st16[stride * (at++) + odd] = (uint16_t)(base + (uint32_t)__builtin_ctzll(bits));
          bits &= bits - 1;
        }
      };
      emit(a0, v0, pa0, 0u);
      emit(a1, v0 + 64u, pa0, 0u);
      emit(a2, v1, pa1, 0u);
      emit(a3, v1 + 64u, pa1, 0u);
      if (enc == kTypeRun) {
        emit(e0, v0, pe0, 1u);
        emit(e1, v0 + 64u, pe0, 1u);
        emit(e2, v1, pe1, 1u);
        emit(e3, v1 + 64u, pe1, 1u);
      }
      __syncthreads();
      so.len = enc == kTypeRun ? tot_r : tot_u;
      const uint32_t chunks = ((enc == kTypeRun ? 4u * tot_r : 2u * tot_u) + 15u) >> 4;  // <= 512
      const ulonglong2* src = reinterpret_cast<const ulonglong2*>(acc);
      ulonglong2* q = reinterpret_cast<ulonglong2*>(dst);
      if ((uint32_t)t < chunks) st_stream(&q[t], src[t]);
      if (256u + (uint32_t)t < chunks) st_stream(&q[256 + t], src[256 + t]);
    }
    if (t == 0) outSlots[cell] = so;
  } else if (WRITE) {
    Slot so;
    so.off = cell * 8192ull;
    so.len = kWords;
    so.tn = make_tn(tot_u ? kTypeBitmap : kTypeNil, tot_u);
    if (tot_u) {
      ulonglong2* q = reinterpret_cast<ulonglong2*>(arenaO + so.off);
      ulonglong2 v0, v1;
      v0.x = w[0];
      v0.y = w[1];
      v1.x = w[2];
      v1.y = w[3];
      st_stream(&q[t], v0);
      st_stream(&q[256 + t], v1);
    }
    if (t == 0) {
      outSlots[cell] = so;
      if (outRuns) outRuns[cell] = s_cnt[2][0] + s_cnt[2][1] + s_cnt[2][2] + s_cnt[2][3];
    }
  }
  if (t == 0 && tot_f && out_counts) atomicAdd(&out_counts[g], (u64)tot_f);
}

}  // namespace fbk
