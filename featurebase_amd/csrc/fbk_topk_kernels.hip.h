// fbk_topk_kernels.hip.h — many rows x ONE filter row: counts[shard][i] = |A[shard][i] ∩ F[shard]|,
// the inner loop of doTopK (executor.go:2705-2746: every container of a fragment against
// topKFilter[key % 16]) and of fragment.top (fragment.go:1317-1437: filter.intersectionCount(row)
// per candidate row).
//
// One 256-thread block per (shard, slot, chunk of rows).  The filter container is brought into
// LDS ONCE per block as a 1024-word bitmap together with a rank table (popcount of all words
// before word w), and the row containers stream past it in their ENCODED form — they are never
// decoded:
//   array   one LDS bit probe per value          (intersectionCountArrayBitmap, roaring.go:4596-4608)
//   run     rank(last+1) - rank(start) per run: two table lookups + two masked popcounts,
//           whatever the run length              (intersectionCountBitmapRun via BitmapCountRange,
//                                                 roaring.go:4559-4567, 3092-3125)
//   bitmap  16 words per lane AND-ed with the LDS copy (popcountAndSlice, roaring.go:6928)
// Each wavefront walks rows w, w+4, ... and receives their payloads as ONE sequence of 1 KiB
// chunks through a 6-chunk register ring (asm loads, constant vmcnt — same scheme and same
// reasons as fbk_fold_kernels.hip.h: 8 wavefronts per SIMD beat a deep ring of whole containers).
// Pairing every row with the filter through the generic pair kernel (k_icount) re-reads and
// re-decodes the filter container once per row: 193 us on 128 shards x 64 mixed rows; this
// kernel with a ring of 3 whole containers: 114 us.
#pragma once
#include "fbk_query_kernels.hip.h"

namespace fbk {

// Row records of a prepared query: recs[(unit * 16 + slot) * n_per + i] = the descriptor of slot `slot` of row
// rows[unit * n_per + i] (unit = a group of a fold / a shard of a TopN).  Resolved once per version of the batch; the
// scatter kernels then read the descriptors of a (unit, slot) as ONE contiguous piece.
__global__ void __launch_bounds__(256) k_resolve_rows(const Slot* __restrict__ slots, const uint32_t* __restrict__ rows, uint64_t n_units, uint32_t n_per,
                                                     Slot* __restrict__ recs) {
  const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_units * kSlots * n_per) return;
  const uint32_t i = (uint32_t)(idx % n_per);
  const uint64_t us = idx / n_per;
  const uint32_t slot = (uint32_t)(us & 15u);
  const uint64_t unit = us >> 4;
  recs[idx] = slots[(uint64_t)rows[unit * n_per + i] * kSlots + slot];
}

__global__ void __launch_bounds__(256, 8) k_rows_vs_filter(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                                          const uint32_t* __restrict__ rowsA, uint32_t nA,
                                                          const Slot* __restrict__ slotsF, const uint8_t* __restrict__ arenaF,
                                                          const uint32_t* __restrict__ rowsF, uint32_t n_shards,
                                                          uint32_t rows_per_block, u64* __restrict__ out_shard, const Slot* __restrict__ recs) {
  __shared__ u64 F[kWords];         // the filter container as a bitmap
  __shared__ uint32_t rank[kWords + 1];  // rank[w] = popcount(F[0..w))
  __shared__ uint32_t s_part[4];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t chunks = (nA + rows_per_block - 1) / rows_per_block;
  uint32_t b = xcd_swizzle(blockIdx.x, gridDim.x);  // (shard, slot, chunk of rows): a shard's blocks on one XCD
  const uint32_t chunk = b % chunks;
  b /= chunks;
  const uint32_t slot = b & 15u;
  const uint32_t shard = b >> 4;
  if (shard >= n_shards) return;
  // The descriptors of the block's first 64 rows hang off a chain of dependent loads (row index -> descriptor)
  // just like the filter container (row index -> descriptor -> payload): both chains start here, side by side.
  const uint32_t i_begin = chunk * rows_per_block, i_end = min(nA, i_begin + rows_per_block);
  const uint32_t* arow = rowsA + (uint64_t)shard * nA;
  Slot first_mine;
  first_mine.off = 0;
  first_mine.len = 0;
  first_mine.tn = 0;
  // (a prepared query's resolved row records, k_resolve_rows: the nA descriptors of this (shard, slot) side by side)
  const Slot* srec = recs ? recs + ((uint64_t)shard * kSlots + slot) * nA : nullptr;
  if (i_begin + lane < i_end) first_mine = srec ? srec[i_begin + lane] : slotsA[(uint64_t)arow[i_begin + lane] * kSlots + slot];
  const Slot sf = slotsF[(uint64_t)rowsF[shard] * kSlots + slot];
  const uint32_t nf = slot_n(sf);
  if (nf == 0) return;  // nothing can intersect at this slot (block-uniform)
  const bool f_full = nf == 65536u;
  if (!f_full) {
    // ---- filter -> LDS bitmap (wave 0 decodes arrays / runs in place) + rank table ----
    if (slot_type(sf) == kTypeBitmap) {
      const ulonglong2* q = reinterpret_cast<const ulonglong2*>(arenaF + sf.off);
      reinterpret_cast<ulonglong2*>(F)[t] = q[t];
      reinterpret_cast<ulonglong2*>(F)[256 + t] = q[256 + t];
    } else if (t < kWave) {
      u64 f[kWordsPerLane];
      frag_load(sf, arenaF, t, F, f);
      lds_write_frag(F, t, f);
    }
    __syncthreads();
    // exclusive prefix popcount over 1024 words: 4 words per thread, wave scan, wave offsets
    const uint32_t p0 = __popcll(F[4 * t]), p1 = __popcll(F[4 * t + 1]), p2 = __popcll(F[4 * t + 2]), p3 = __popcll(F[4 * t + 3]);
    const uint32_t mine = p0 + p1 + p2 + p3;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += v;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int k = 0; k < wv; ++k) base += s_part[k];
    rank[4 * t] = base;
    rank[4 * t + 1] = base + p0;
    rank[4 * t + 2] = base + p0 + p1;
    rank[4 * t + 3] = base + p0 + p1 + p2;
    if (t == 255) rank[kWords] = base + mine;
    __syncthreads();
  }
  const uint32_t* F32 = reinterpret_cast<const uint32_t*>(F);
  auto rank_of = [&](uint32_t x) -> uint32_t {  // bits of F below position x (0 <= x <= 65536)
    const uint32_t w = x >> 6;
    const uint32_t r = rank[w];
    const uint32_t bpos = x & 63u;
    return bpos ? r + __popcll(F[w] & (~0ull >> (64 - bpos))) : r;
  };

  for (uint32_t base = i_begin; base < i_end; base += 64) {
    Slot mine = first_mine;  // lane l holds the descriptor of row base+l
    if (base != i_begin) {
      mine.off = 0;
      mine.len = 0;
      mine.tn = 0;
      if (base + lane < i_end) mine = srec ? srec[base + lane] : slotsA[(uint64_t)arow[base + lane] * kSlots + slot];
    }
    const uint32_t cnt = min(64u, i_end - base);
    auto meta = [&](uint32_t i, u64& off, uint32_t& len, uint32_t& tn) {
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mine.off, (int)(i & 63));
      const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(mine.off >> 32), (int)(i & 63));
      off = ((u64)hi << 32) | lo;
      len = __builtin_amdgcn_readlane(mine.len, (int)(i & 63));
      tn = __builtin_amdgcn_readlane(mine.tn, (int)(i & 63));
    };
    const uint32_t my_n = mine.tn & 0xFFFFFFu, my_bytes = payload_bytes(mine.tn >> 24, mine.len);
    // intersectionCount's short-circuits on the stored N (roaring.go:4478-4486) need no payload:
    // the lane that holds the descriptor adds the count itself (rows are dealt to waves by i & 3)
    const bool trivial = my_n != 0 && (f_full || my_n == 65536u);
    if (trivial && (uint32_t)(lane & 3) == (uint32_t)wv && base + lane < i_end)
      atomicAdd(&out_shard[(uint64_t)shard * nA + base + lane], (u64)(f_full ? my_n : nf));
    // payloads beyond 8 KiB (arrays > 4096 values / > 2048 runs: outside roaring policy) do not fit
    // the ring: counted straight from global memory, dealt to the waves in turn
    {
      u64 bigm = __ballot(my_n != 0 && !trivial && my_bytes > 8192u);
      for (uint32_t ord = 0; bigm; ++ord) {
        const uint32_t i = (uint32_t)__builtin_ctzll(bigm);
        bigm &= bigm - 1;
        if ((ord & 3u) != (uint32_t)wv) continue;
        u64 off;
        uint32_t len, tn;
        meta(i, off, len, tn);
        const uint8_t* p = arenaA + off;
        uint32_t c = 0;
        if ((tn >> 24) == kTypeArray) {
          const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
          for (uint32_t k = lane; k < len; k += kWave) {
            const uint32_t e = q[k];
            c += (F32[e >> 5] >> (e & 31)) & 1u;
          }
        } else {
          const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
          for (uint32_t k = lane; k < len; k += kWave) {
            const uint32_t iv = q[k];
            c += rank_of((iv >> 16) + 1u) - rank_of(iv & 0xFFFFu);
          }
        }
        c = wave_reduce_add(c);
        if (lane == 0 && c) atomicAdd(&out_shard[(uint64_t)shard * nA + base + i], (u64)c);
      }
    }
    // ---- the other containers: one sequence of 1 KiB chunks through a register ring ----
    constexpr int NCH = 6;
    typedef uint32_t Chunk __attribute__((ext_vector_type(4)));
    Chunk C[NCH];
    const uint32_t nr = (my_n != 0 && !trivial && my_bytes <= 8192u) ? (my_bytes + 1023u) >> 10 : 0u;  // chunks of lane l's container
    auto next_valid = [&](uint32_t i) {
      while (i < cnt && __builtin_amdgcn_readlane(nr, (int)(i & 63)) == 0) i += 4;
      return i;
    };
    // producer (next chunk to load) and consumer (next chunk to count): position in the sequence
    // plus the descriptor of the current container, held in scalar registers and refreshed only
    // when the container changes (8 v_readlane per chunk otherwise)
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
    auto advance = [&](Cursor& k) -> bool {  // true: the container is finished
      if (++k.j == k.nr) {
        k.i = next_valid(k.i + 4);
        fetch(k);
        return true;
      }
      return false;
    };
    Cursor P, Q;
    P.i = next_valid(wv);
    P.nr = P.len = P.tn = P.bytes = 0;
    P.off = 0;
    fetch(P);
    Q = P;
    // exactly one load instruction per step (see fbk_fold_kernels.hip.h): lanes past the end of a
    // payload re-read its first 16 bytes, an exhausted producer the first 16 bytes of the arena
    const uint32_t lane16 = lane * 16u;
    auto load_chunk = [&](Chunk& c) {
      const uint8_t* p = arenaA;
      if (P.i < cnt) {
        const uint32_t b0 = P.j * 1024u + lane16;
        p = arenaA + P.off + (b0 < P.bytes ? b0 : 0u);
        advance(P);
      }
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(c) : "v"(p));
    };
    // |chunk j of container (type, len) ∩ F|, this lane's share
    auto count_chunk = [&](const uint32_t (&d)[4], uint32_t type, uint32_t len, uint32_t j) -> uint32_t {
      uint32_t c = 0;
      if (type == kTypeBitmap) {
        const ulonglong2 f = reinterpret_cast<const ulonglong2*>(F)[j * kWave + lane];
        c = __popcll((((u64)d[1] << 32) | d[0]) & f.x) + __popcll((((u64)d[3] << 32) | d[2]) & f.y);
      } else if (type == kTypeArray) {
        const uint32_t row0 = j * (kWave * 8u);
        if (row0 + kWave * 8u <= len) {  // scalar: a whole row of 512 values
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t a = d[q] & 0xFFFFu, bb = d[q] >> 16;
            c += ((F32[a >> 5] >> (a & 31)) & 1u) + ((F32[bb >> 5] >> (bb & 31)) & 1u);
          }
        } else {
          // the ragged last row, without a branch: slots past the end hold readable values (lanes past the payload
          // re-read its first 16 bytes, the lane with the end reads on into the padding / the next payload) whose
          // probe is simply not counted
          const uint32_t e0 = row0 + lane * 8u;
          const uint32_t valid = e0 < len ? (1u << min(len - e0, 8u)) - 1u : 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t a = d[q] & 0xFFFFu, bb = d[q] >> 16;
            c += ((F32[a >> 5] >> (a & 31)) & (valid >> (2 * q)) & 1u) + ((F32[bb >> 5] >> (bb & 31)) & (valid >> (2 * q + 1)) & 1u);
          }
        }
      } else {  // run: 4 intervals per 16-byte chunk
        // (no branch: a slot past the last run holds some readable interval whose count is masked away)
        const uint32_t i0 = (j * kWave + lane) * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) c += (rank_of((d[q] >> 16) + 1u) - rank_of(d[q] & 0xFFFFu)) & (i0 + q < len ? ~0u : 0u);
      }
      return c;
    };
#pragma unroll
    for (int q = 0; q < NCH; ++q) load_chunk(C[q]);
    uint32_t c = 0;
    while (Q.i < cnt) {
#pragma unroll
      for (int q = 0; q < NCH; ++q) {
        // (every slot of a lap waits and reloads, only the counting is conditional: see k_fold_scatter)
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(C[q]) : "n"(NCH - 1));
        if (Q.i < cnt) {
          const uint32_t d[4] = {C[q][0], C[q][1], C[q][2], C[q][3]};
          c += count_chunk(d, Q.tn >> 24, Q.len, Q.j);
          const uint32_t row = Q.i;
          if (advance(Q)) {
            // the container's count: sums of the four 16-lane rows on the DPP network, then four scalar reads
            // (a shuffle reduction is six dependent LDS round trips on the wave's critical path)
            c = wave_rows_sum(c);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)c, 0) + (uint32_t)__builtin_amdgcn_readlane((int)c, 16) +
                                 (uint32_t)__builtin_amdgcn_readlane((int)c, 32) + (uint32_t)__builtin_amdgcn_readlane((int)c, 48);
            if (lane == 0 && tot) atomicAdd(&out_shard[(uint64_t)shard * nA + base + row], (u64)tot);
            c = 0;
          }
        }
        load_chunk(C[q]);
      }
    }
    // loads still in flight target registers the compiler is about to reuse
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// ---- TopK / TopN selection on the device ------------------------------------------------------
// counts[shard][i] = cardinality of row i of the shard from the stored container counts
// (Row.Count, row.go:446: the sum of the segment counts) — TopK without a filter
__global__ void __launch_bounds__(256) k_row_cardinality(const Slot* __restrict__ slots, const uint32_t* __restrict__ rows,
                                                        uint64_t n, u64* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Slot* s = slots + (uint64_t)rows[i] * kSlots;
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < kSlots; ++k) c += slot_n(s[k]);
  out[i] = c;
}

__global__ void __launch_bounds__(256) k_iota_u32(uint32_t* __restrict__ v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = i;
}

// The ordering of TopK / TopN for a field of up to kRankSortMax rows in ONE launch: row i's position is the number of rows
// that come before it — a larger count, or the same count and a smaller index (count descending, row index ascending:
// BSIData.PivotDescending, bsi.go:18-62) — and it writes itself there.  O(n^2) compares, tiles of 256 counts through LDS.
// *nz (zeroed by the caller) receives the number of rows with a non-zero count.
constexpr uint32_t kRankSortMax = 4096;
__global__ void __launch_bounds__(256) k_rank_sort_desc(const u64* __restrict__ counts, uint32_t n, u64* __restrict__ out_counts, uint32_t* __restrict__ out_index,
                                                       uint32_t* __restrict__ nz) {
  __shared__ u64 tile[256];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const u64 mine = i < n ? counts[i] : 0;
  uint32_t rank = 0;
  for (uint32_t j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    tile[threadIdx.x] = j0 + threadIdx.x < n ? counts[j0 + threadIdx.x] : 0;
    __syncthreads();
    const uint32_t m = min(256u, n - j0);
    for (uint32_t k = 0; k < m; ++k) {
      const u64 c = tile[k];
      rank += (c > mine || (c == mine && j0 + k < i)) ? 1u : 0u;
    }
  }
  if (i < n) {
    out_counts[rank] = mine;
    out_index[rank] = i;
  }
  const u64 any = __ballot(i < n && mine != 0);
  if ((threadIdx.x & 63) == 0 && any) atomicAdd(nz, (uint32_t)__popcll(any));
}

// number of leading non-zero keys of a descending sequence (binary search by one thread)
__global__ void k_count_nonzero_desc(const u64* __restrict__ keys, uint32_t n, uint32_t* __restrict__ out) {
  uint32_t lo = 0, hi = n;  // first index with key == 0
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] != 0) lo = mid + 1;
    else hi = mid;
  }
  *out = lo;
}

}  // namespace fbk
