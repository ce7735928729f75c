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
// Each wavefront walks rows w, w+4, ... with a 3-deep payload prefetch ring.  Pairing every row
// with the filter through the generic pair kernel (k_icount) re-reads and re-decodes the filter
// container once per row: 193 us vs the bytes-once time of the fused union (104 us) on
// 128 shards x 64 mixed rows.
#pragma once
#include "fbk_query_kernels.hip.h"

namespace fbk {

__global__ void __launch_bounds__(256, 4) k_rows_vs_filter(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                                          const uint32_t* __restrict__ rowsA, uint32_t nA,
                                                          const Slot* __restrict__ slotsF, const uint8_t* __restrict__ arenaF,
                                                          const uint32_t* __restrict__ rowsF, uint32_t n_shards,
                                                          uint32_t rows_per_block, u64* __restrict__ out_shard) {
  __shared__ u64 F[kWords];         // the filter container as a bitmap
  __shared__ uint32_t rank[kWords + 1];  // rank[w] = popcount(F[0..w))
  __shared__ uint32_t s_part[4];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t chunks = (nA + rows_per_block - 1) / rows_per_block;
  uint32_t b = blockIdx.x;
  const uint32_t chunk = b % chunks;
  b /= chunks;
  const uint32_t slot = b & 15u;
  const uint32_t shard = b >> 4;
  if (shard >= n_shards) return;
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

  const uint32_t i_begin = chunk * rows_per_block, i_end = min(nA, i_begin + rows_per_block);
  const uint32_t* arow = rowsA + (uint64_t)shard * nA;
  for (uint32_t base = i_begin; base < i_end; base += 64) {
    Slot mine;  // lane l holds the descriptor of row base+l
    mine.off = 0;
    mine.len = 0;
    mine.tn = 0;
    if (base + lane < i_end) mine = slotsA[(uint64_t)arow[base + lane] * kSlots + slot];
    const uint32_t cnt = min(64u, i_end - base);
    constexpr int D = 3;
    Raw R[D];
    auto meta = [&](uint32_t i, u64& off, uint32_t& len, uint32_t& tn) {
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mine.off, (int)(i & 63));
      const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(mine.off >> 32), (int)(i & 63));
      off = ((u64)hi << 32) | lo;
      len = __builtin_amdgcn_readlane(mine.len, (int)(i & 63));
      tn = __builtin_amdgcn_readlane(mine.tn, (int)(i & 63));
    };
    auto needs_payload = [&](uint32_t tn) {  // intersectionCount short-circuits, roaring.go:4478-4486
      const uint32_t n = tn & 0xFFFFFFu;
      return n != 0 && n != 65536u && !f_full;
    };
    auto issue = [&](uint32_t i, Raw& r) {
      if (i < cnt) {
        u64 off;
        uint32_t len, tn;
        meta(i, off, len, tn);
        const uint32_t bytes = payload_bytes(tn >> 24, len);
        if (needs_payload(tn) && bytes <= 8192u) raw_load(arenaA + off, bytes, lane, r);
      }
    };
    auto consume = [&](uint32_t i, const Raw& r) {
      if (i >= cnt) return;
      u64 off;
      uint32_t len, tn;
      meta(i, off, len, tn);
      const uint32_t n = tn & 0xFFFFFFu, type = tn >> 24;
      if (n == 0) return;
      uint32_t c = 0;
      if (f_full) {
        c = (lane == 0) ? n : 0;  // a.N == 65536 -> b.N
      } else if (n == 65536u) {
        c = (lane == 0) ? nf : 0;
      } else if (payload_bytes(type, len) > 8192u) {  // arrays > 4096 values / > 2048 runs: outside roaring policy
        const uint8_t* p = arenaA + off;
        if (type == kTypeArray) {
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
      } else if (type == kTypeBitmap) {
        const ulonglong2* qf = reinterpret_cast<const ulonglong2*>(F);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const ulonglong2 f = qf[j * kWave + lane];
          c += __popcll(r.v[j].x & f.x) + __popcll(r.v[j].y & f.y);
        }
      } else if (type == kTypeArray) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t e0 = (j * kWave + lane) * 8u;
          if (e0 < len) {
            const u64 lo = r.v[j].x, hi = r.v[j].y;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t a = (uint32_t)(lo >> (16 * q)) & 0xFFFFu, bb = (uint32_t)(hi >> (16 * q)) & 0xFFFFu;
              if (e0 + q < len) c += (F32[a >> 5] >> (a & 31)) & 1u;
              if (e0 + 4 + q < len) c += (F32[bb >> 5] >> (bb & 31)) & 1u;
            }
          }
        }
      } else {  // run: 4 intervals per 16-byte chunk
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t i0 = (j * kWave + lane) * 4u;
          if (i0 < len) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const u64 w = (q < 2) ? r.v[j].x : r.v[j].y;
              const uint32_t iv = (uint32_t)(w >> (32 * (q & 1)));
              if (i0 + q < len) c += rank_of((iv >> 16) + 1u) - rank_of(iv & 0xFFFFu);
            }
          }
        }
      }
      c = wave_reduce_add(c);
      if (lane == 0 && c) atomicAdd(&out_shard[(uint64_t)shard * nA + base + i], (u64)c);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(wv + 4 * d, R[d]);
    for (uint32_t i = wv; i < cnt; i += 4 * D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        consume(i + 4 * d, R[d]);
        issue(i + 4 * (d + D), R[d]);
      }
    }
  }
}

}  // namespace fbk
