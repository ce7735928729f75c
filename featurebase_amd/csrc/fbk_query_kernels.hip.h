// fbk_query_kernels.hip.h — query-level CDNA4 kernels built on the wave-owns-a-container
// fragments of fbk_kernels.hip.h:
//   k_union_n        n-way union of rows (+ fused |union ∩ filter|)     roaring.go:1272-1561, filter.go:294-366
//   k_count_matrix   |A_i ∩ B_j (∩ F)| for all i, j per shard           executor.go:8880-8934 (GroupBy), :2705-2746 (TopK)
//   k_bsi_sum        BSI Sum / Count over bit planes                     roaring/filter.go:1097-1218, fragment.go:724-750
//   k_bsi_range      BSI Range via a host-generated plane program        fragment.go:937-1303
//   k_encode_*       Container.optimize() re-encode of bitmap cells      roaring.go:3412-3461, 3687-3928
#pragma once
#include "fbk_kernels.hip.h"

namespace fbk {

__device__ __forceinline__ u64 wave_reduce_add64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

__device__ __forceinline__ void lds_write_frag(u64* slot, int lane, const u64 (&w)[kWordsPerLane]) {
  ulonglong2* q2 = reinterpret_cast<ulonglong2*>(slot);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    ulonglong2 v;
    v.x = w[2 * jj];
    v.y = w[2 * jj + 1];
    q2[jj * kWave + lane] = v;
  }
}

// ---- raw payload prefetch ----------------------------------------------------------------
// A container payload of at most 8 KiB (every bitmap, arrays <= 4096 values, <= 2048 runs —
// i.e. everything roaring policy produces, roaring.go:3036-3044) is fetched as eight 16-byte
// chunks per lane *before* it is needed and decoded later from registers, so that a wave
// walking a list of containers overlaps the HBM latency of container i+1 with the decode of
// container i.  Chunk c = 64*j + lane covers payload bytes [16c, 16c+16).
struct Raw {
  ulonglong2 v[8];
};
__device__ __forceinline__ uint32_t payload_bytes(uint32_t type, uint32_t len) {
  return type == kTypeBitmap ? 8192u : type == kTypeArray ? len * 2u : len * 4u;
}
__device__ __forceinline__ void raw_load(const uint8_t* __restrict__ p, uint32_t bytes, int lane, Raw& r) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t c = j * kWave + lane;
    if (c * 16u < bytes) r.v[j] = ld_stream(&q[c]);  // payloads are padded to 16 bytes in the arena
  }
}
// Decode a prefetched payload (bytes <= 8192) into a fragment.
__device__ __forceinline__ void frag_from_raw(const Raw& r, uint32_t type, uint32_t len, int lane, u64* scratch,
                                              u64 (&w)[kWordsPerLane]) {
  if (type == kTypeBitmap) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      w[2 * j] = r.v[j].x;
      w[2 * j + 1] = r.v[j].y;
    }
    return;
  }
  lds_zero(scratch, lane);
  wave_lds_sync();
  uint32_t* s32 = reinterpret_cast<uint32_t*>(scratch);
  if (type == kTypeArray) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t e0 = (j * kWave + lane) * 8u;
      if (e0 < len) {
        const u64 lo = r.v[j].x, hi = r.v[j].y;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t a = (uint32_t)(lo >> (16 * t)) & 0xFFFFu, bb = (uint32_t)(hi >> (16 * t)) & 0xFFFFu;
          if (e0 + t < len) atomicOr(&s32[a >> 5], 1u << (a & 31));
          if (e0 + 4 + t < len) atomicOr(&s32[bb >> 5], 1u << (bb & 31));
        }
      }
    }
    wave_lds_sync();
    lds_read_frag(scratch, lane, w);
    wave_lds_sync();
    return;
  }
  // run: toggles, then parity prefix scan (as frag_load_run)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t i0 = (j * kWave + lane) * 4u;
    if (i0 < len) {
      const u64 lo = r.v[j].x, hi = r.v[j].y;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t iv = (t < 2) ? (uint32_t)(lo >> (32 * t)) : (uint32_t)(hi >> (32 * (t - 2)));
        if (i0 + t < len) {
          const uint32_t st = iv & 0xFFFFu, e = (iv >> 16) + 1u;
          atomicXor(&s32[st >> 5], 1u << (st & 31));
          if (e < 65536u) atomicXor(&s32[e >> 5], 1u << (e & 31));
        }
      }
    }
  }
  wave_lds_sync();
  lds_read_frag(scratch, lane, w);
  wave_lds_sync();
  const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t carry = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u64 t0 = w[2 * j], t1 = w[2 * j + 1];
    uint32_t p0 = __popcll(t0) & 1u, p1 = __popcll(t1) & 1u;
    u64 m = __ballot((p0 ^ p1) != 0);
    uint32_t in = carry ^ (__popcll(m & lane_lt) & 1u);
    w[2 * j] = prefix_xor64(t0) ^ (in ? ~0ull : 0ull);
    w[2 * j + 1] = prefix_xor64(t1) ^ ((in ^ p0) ? ~0ull : 0ull);
    carry ^= __popcll(m) & 1u;
  }
}

// ---- n-way fold (union / intersect / xor / difference of k rows) ---------------------------
// One 256-thread block per (group, slot): the k containers of the group's rows at that
// slot are split over the 4 wavefronts (wave w takes rows w, w+4, ...), each accumulating
// in registers (the result never touches HBM unless WRITE); the 4 partial results are
// combined through LDS.  Latency is what matters here (many small containers): all
// descriptors of the group are fetched by one vector load (lane i reads row i's slot) and
// handed out by readlane, and the payload of container i+1 is prefetched while container i
// is decoded (raw_load / frag_from_raw).  A single wave walking 64 containers with dependent
// descriptor -> payload loads measured 184 us for 64 shards x 64 rows; see profiles/.
//   OP 1 (OR):  r0 | r1 | ...   BitmapRowsUnion's 16 accumulators (filter.go:327-334) and the
//               n-way Bitmap.unionInPlace per key (roaring.go:1455-1560); any full operand =>
//               full container (roaring.go:1465-1474).
//   OP 0 (AND): r0 & r1 & ...   executeIntersectShard's left fold (executor.go:5357-5380) /
//               Bitmap.IntersectInPlace(others...) (roaring.go:855-925): any empty operand =>
//               empty, full operands are the identity (roaring.go:929-942).
//   OP 2 (XOR): r0 ^ r1 ^ ...   executeXorShard's left fold (executor.go:5513-5552).
//   OP 3 (ANDNOT): r0 \ r1 \ r2 ... = r0 & ~(r1 | r2 | ...)   executeDifferenceShard
//               (executor.go:2950-2983) / Bitmap.Difference(others...) (roaring.go:1564-1595).
// With a filter batch: counts[g] += |result ∩ F.rows_f[g]| (Bitmap.IntersectionCount of
// the result against the filter row), else counts[g] += |result|.
template <int OP, int WRITE>  // WRITE: 0 counts only, 1 the result as a bitmap cell (+ run count), 2 Container.optimize() applied here (frag_store_encoded)
__global__ void __launch_bounds__(256) k_fold_n(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                               const uint32_t* __restrict__ rows, uint64_t n_groups, uint32_t k,
                                               const Slot* __restrict__ fslots, const uint8_t* __restrict__ farena,
                                               const uint32_t* __restrict__ frows, uint8_t* __restrict__ arenaO,
                                               Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                               u64* __restrict__ out_counts) {
  __shared__ u64 lds[4][kWords];
  __shared__ uint32_t s_short;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t cell = blockIdx.x;  // (group, slot)
  const uint64_t g = cell >> 4;
  const uint32_t slot = cell & 15;
  if (g >= n_groups) return;  // block-uniform
  if (threadIdx.x == 0) s_short = 0;
  __syncthreads();
  u64 acc[kWordsPerLane];
#pragma unroll
  for (int q = 0; q < kWordsPerLane; ++q) acc[q] = (OP == 0) ? ~0ull : 0ull;
  const uint32_t* grow = rows + g * k;
  // a container that leaves the accumulator unchanged: empty for OR/XOR/ANDNOT, full for AND
  auto is_identity = [](uint32_t tn) { return (OP == 0) ? (tn & 0xFFFFFFu) == 65536u : (tn & 0xFFFFFFu) == 0u; };
  bool shortcut = false;  // OR/ANDNOT: a full operand saturates; AND: an empty operand annihilates
  for (uint32_t base = 0; base < k && !shortcut; base += 64) {
    // lane l holds the descriptor of row base+l of the group
    Slot mine;
    mine.off = 0;
    mine.len = 0;
    mine.tn = 0;
    const bool valid = base + lane < k;
    if (valid) mine = slots[(uint64_t)grow[base + lane] * kSlots + slot];
    if (OP == 3 && base + lane == 0) mine.tn = 0;  // r0 is not one of the subtrahends
    const uint32_t cnt = min(64u, k - base);
    if (OP == 1 || OP == 3) {
      if (__ballot(slot_n(mine) == 65536u) != 0) shortcut = true;
    } else if (OP == 0) {
      if (__ballot(valid && slot_n(mine) == 0u) != 0) shortcut = true;
    }
    if (shortcut) break;
    Raw cur, nxt;
    uint32_t i = wv;
    // descriptor of this wave's first container
    u64 off = __shfl(mine.off, (int)(i & 63), kWave);
    uint32_t len = __shfl(mine.len, (int)(i & 63), kWave), tn = __shfl(mine.tn, (int)(i & 63), kWave);
    bool have = i < cnt;
    uint32_t bytes = have ? payload_bytes(tn >> 24, len) : 0;
    if (have && !is_identity(tn) && bytes <= 8192u) raw_load(arena + off, bytes, lane, cur);
    while (have) {
      const uint32_t in = i + 4;
      const bool have_n = in < cnt;
      u64 off_n = 0;
      uint32_t len_n = 0, tn_n = 0, bytes_n = 0;
      if (have_n) {
        off_n = __shfl(mine.off, (int)(in & 63), kWave);
        len_n = __shfl(mine.len, (int)(in & 63), kWave);
        tn_n = __shfl(mine.tn, (int)(in & 63), kWave);
        bytes_n = payload_bytes(tn_n >> 24, len_n);
        if (!is_identity(tn_n) && bytes_n <= 8192u) raw_load(arena + off_n, bytes_n, lane, nxt);
      }
      if (!is_identity(tn)) {
        u64 w[kWordsPerLane];
        if (bytes <= 8192u) {
          frag_from_raw(cur, tn >> 24, len, lane, lds[wv], w);
        } else {  // arrays > 4096 values / > 2048 runs: legal but outside roaring policy
          Slot s;
          s.off = off;
          s.len = len;
          s.tn = tn;
          frag_load(s, arena, lane, lds[wv], w);
        }
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) {
          if (OP == 0) acc[q] &= w[q];
          else if (OP == 2) acc[q] ^= w[q];
          else acc[q] |= w[q];
        }
      }
      i = in;
      have = have_n;
      off = off_n;
      len = len_n;
      tn = tn_n;
      bytes = bytes_n;
#pragma unroll
      for (int j = 0; j < 8; ++j) cur.v[j] = nxt.v[j];
    }
  }
  if (shortcut && lane == 0) s_short = 1;
  // publish the partial result of this wave (its scratch is free again)
  lds_write_frag(lds[wv], lane, acc);
  __syncthreads();
  if (wv != 0) return;
  if (s_short) {
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) acc[q] = (OP == 1 || OP == 3) ? ~0ull : 0ull;
  } else {
#pragma unroll
    for (int o = 1; o < 4; ++o) {
      u64 w[kWordsPerLane];
      lds_read_frag(lds[o], lane, w);
#pragma unroll
      for (int q = 0; q < kWordsPerLane; ++q) {
        if (OP == 0) acc[q] &= w[q];
        else if (OP == 2) acc[q] ^= w[q];
        else acc[q] |= w[q];
      }
    }
  }
  if (OP == 3) {  // r0 \ (r1 | r2 | ...)
    const Slot s0 = slots[(uint64_t)grow[0] * kSlots + slot];
    u64 w[kWordsPerLane];
    if (slot_n(s0) == 0 || s_short) frag_zero(w);
    else frag_load(s0, arena, lane, lds[0], w);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) acc[q] = w[q] & ~acc[q];
  }
  uint32_t c;
  if (fslots) {
    const Slot sf = fslots[(uint64_t)frows[g] * kSlots + slot];
    if (slot_n(sf) == 0) {
      c = 0;
    } else {
      u64 w[kWordsPerLane];
      frag_load(sf, farena, lane, lds[0], w);
      uint32_t part = 0;
#pragma unroll
      for (int q = 0; q < kWordsPerLane; ++q) part += __popcll(acc[q] & w[q]);
      c = wave_reduce_add(part);
    }
  } else {
    c = wave_reduce_add(frag_popcount(acc));
  }
  if (WRITE) {
    const uint32_t cu = fslots ? wave_reduce_add(frag_popcount(acc)) : c;
    Slot so;
    so.off = cell * 8192ull;
    so.len = kWords;
    so.tn = make_tn(cu ? kTypeBitmap : kTypeNil, cu);
    uint32_t r = 0;
    if (outRuns || WRITE == 2) r = wave_reduce_add(frag_count_runs(acc, lane));
    if (WRITE == 2 && cu) {  // (wave 0's scratch is free: every partial has been read)
      uint32_t t_out, l_out;
      frag_store_encoded(acc, cu, r, lane, lds[0], arenaO + so.off, t_out, l_out);
      so.len = l_out;
      so.tn = make_tn(t_out, cu);
    } else if (cu) {
      frag_store_bitmap(arenaO + so.off, lane, acc);
    }
    if (lane == 0) {
      outSlots[cell] = so;
      if (outRuns) outRuns[cell] = r;
    }
  }
  if (lane == 0 && c && out_counts) atomicAdd(&out_counts[g], (u64)c);
}

// ---- count matrix (GroupBy / TopK / TopN shape) -----------------------------------------------
// out_shard[(shard*nA + i)*nBtot + j] (+)= sum over the block's slots of
//     |A[shard][i] ∩ B[shard][j] ∩ F[shard]|.
// One 512-thread block (8 wavefronts) per (shard, slot group, group of 8*TA A-rows).
// Wave w keeps TA A-fragments (already ANDed with the filter: rows[0] ∩= filter,
// executor.go:8830) in registers.  The B containers of the slot are fetched from HBM exactly
// ONCE per block: in groups of 8, wave w brings B[8g+w] into slot w of a double-buffered LDS
// ring — bitmap containers by direct global->LDS DMA (global_load_lds_dwordx4: lane l of
// load j lands at 1024*j + 16*l, which IS the fragment layout), array/run containers by
// decoding in place — and after one barrier every wave reads all 8 ring slots (ds_read_b128)
// against its A tile while the next group's DMA is in flight.  A first version that re-read
// B from global memory per A tile was bound by HBM re-reads (787 us for 128 shards x 32x32
// rows: the per-XCD working set is far beyond the 4 MiB L2); keeping the prefetched B
// fragment in registers instead of using the DMA spilled VGPRs.
// Counts are wave-reduced and accumulated lane-distributed (lane j%64 owns column j), so no
// atomics are needed inside the block.
__device__ __forceinline__ void dma_bitmap_to_lds(const uint8_t* __restrict__ g, u64* lds_slot, int lane) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    __builtin_amdgcn_global_load_lds((gptr_t)(g + j * 1024 + lane * 16),
                                     (lptr_t)(reinterpret_cast<uint8_t*>(lds_slot) + j * 1024), 16, 0, 0);
}

// Bring one container into an LDS ring slot as a 1024-word bitmap (async for bitmaps).
__device__ __forceinline__ void ring_fetch(const Slot& s, const uint8_t* __restrict__ arena, int lane, u64* slot) {
  const uint32_t t = slot_type(s);
  if (slot_n(s) == 0 || t == kTypeNil) {
    lds_zero(slot, lane);
  } else if (t == kTypeBitmap) {
    dma_bitmap_to_lds(arena + s.off, slot, lane);
  } else {
    u64 w[kWordsPerLane];
    frag_load(s, arena, lane, slot, w);  // decodes in `slot`, result in registers
    lds_write_frag(slot, lane, w);
  }
}

template <int TA>
__global__ void __launch_bounds__(512) k_count_matrix(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                                     const uint32_t* __restrict__ rowsA, uint32_t nA,
                                                     const Slot* __restrict__ slotsB, const uint8_t* __restrict__ arenaB,
                                                     const uint32_t* __restrict__ rowsB, uint32_t nBtot, uint32_t j0,
                                                     uint32_t nB, const Slot* __restrict__ slotsF,
                                                     const uint8_t* __restrict__ arenaF, const uint32_t* __restrict__ rowsF,
                                                     uint32_t n_shards, uint32_t spb, u64* __restrict__ out_shard) {
  constexpr int kWaves = 8;
  constexpr int kNBC = 4;  // 64-column chunks per launch: nB <= 256
  __shared__ u64 ring[2][kWaves][kWords];  // 128 KiB of the CU's 160 KiB LDS
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint32_t agroups = (nA + kWaves * TA - 1) / (kWaves * TA);
  const uint32_t sgroups = kSlots / spb;
  uint32_t b = blockIdx.x;
  const uint32_t ag = b % agroups;
  b /= agroups;
  const uint32_t sg = b % sgroups;
  const uint32_t shard = b / sgroups;
  if (shard >= n_shards) return;
  const uint32_t i0 = ag * kWaves * TA + wv * TA;
  uint32_t acc[TA][kNBC];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int c = 0; c < kNBC; ++c) acc[a][c] = 0;
  const uint32_t groups = (nB + kWaves - 1) / kWaves;
  const uint32_t* brow = rowsB + (uint64_t)shard * nBtot + j0;
  for (uint32_t slot = sg * spb; slot < (sg + 1) * spb; ++slot) {
    // ---- A tile (and filter) of this wave; ring[0][wv] doubles as decode scratch ----
    u64 fa[TA][kWordsPerLane];
    bool any = false;
    const bool have_f = slotsF != nullptr;
    if (have_f) {
      const Slot sf = slotsF[(uint64_t)rowsF[shard] * kSlots + slot];
      if (slot_n(sf) == 0) continue;  // block-uniform: nothing can intersect at this slot
    }
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      bool present = false;
      if (i0 + a < nA) {
        const Slot sa = slotsA[(uint64_t)rowsA[(uint64_t)shard * nA + i0 + a] * kSlots + slot];
        if (slot_n(sa) != 0) {
          frag_load(sa, arenaA, lane, ring[0][wv], fa[a]);
          present = true;
        }
      }
      if (!present) frag_zero(fa[a]);
      any |= present;
    }
    if (have_f) {
      const Slot sf = slotsF[(uint64_t)rowsF[shard] * kSlots + slot];
      u64 wf[kWordsPerLane];
      frag_load<false>(sf, arenaF, lane, ring[0][wv], wf);
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) fa[a][q] &= wf[q];
    }
    // ---- B pipeline: one barrier per group of 8 containers ----
    if (wv < (int)nB) ring_fetch(slotsB[(uint64_t)brow[wv] * kSlots + slot], arenaB, lane, ring[0][wv]);
    for (uint32_t g = 0; g < groups; ++g) {
      const uint32_t cur = g & 1u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA into ring[cur][wv] has landed
      __syncthreads();
      if (g + 1 < groups) {
        const uint32_t j = (g + 1) * kWaves + wv;
        if (j < nB) ring_fetch(slotsB[(uint64_t)brow[j] * kSlots + slot], arenaB, lane, ring[cur ^ 1u][wv]);
      }
      if (any) {
#pragma unroll 1
        for (int r = 0; r < kWaves; ++r) {  // not unrolled: keeps only one 16-byte B chunk live
          const uint32_t j = g * kWaves + r;
          if (j >= nB) break;
          const ulonglong2* qb = reinterpret_cast<const ulonglong2*>(ring[cur][r]);
          uint32_t p[TA];
#pragma unroll
          for (int a = 0; a < TA; ++a) p[a] = 0;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const ulonglong2 v = qb[jj * kWave + lane];
#pragma unroll
            for (int a = 0; a < TA; ++a) p[a] += __popcll(fa[a][2 * jj] & v.x) + __popcll(fa[a][2 * jj + 1] & v.y);
          }
#pragma unroll
          for (int a = 0; a < TA; ++a) {
            const uint32_t c = wave_reduce_add(p[a]);
            if (lane == (int)(j & 63u)) {
#pragma unroll
              for (int cc = 0; cc < kNBC; ++cc)
                if ((j >> 6) == (uint32_t)cc) acc[a][cc] += c;
            }
          }
        }
      }
    }
    __syncthreads();  // everyone is done with the ring before the next slot reuses it as scratch
  }
#pragma unroll
  for (int a = 0; a < TA; ++a) {
    if (i0 + a >= nA) continue;
#pragma unroll
    for (int cc = 0; cc < kNBC; ++cc) {
      const uint32_t j = cc * 64 + lane;
      if (j < nB && acc[a][cc]) {
        u64* dst = &out_shard[((uint64_t)shard * nA + i0 + a) * nBtot + j0 + j];
        if (spb == kSlots) *dst = acc[a][cc];
        else atomicAdd(dst, (u64)acc[a][cc]);
      }
    }
  }
}

// out[c] += sum over a chunk of shards of in[shard*width + c]   (mergeGroupCounts' arithmetic,
// executor.go:3728).  grid = (width/256, shard chunks); out must be zeroed.
__global__ void __launch_bounds__(256) k_reduce_shards(const u64* __restrict__ in, uint32_t n_shards, uint64_t width,
                                                      u64* __restrict__ out) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  const uint32_t per = (n_shards + gridDim.y - 1) / gridDim.y;
  const uint32_t s0 = blockIdx.y * per, s1 = min(n_shards, s0 + per);
  u64 acc = 0;
  for (uint32_t s = s0; s < s1; ++s) acc += in[(uint64_t)s * width + c];
  if (acc) atomicAdd(&out[c], acc);
}

// ---- Container.optimize() on the device ---------------------------------------------------------
// Phase 1: choose the encoding of every output cell exactly as optimize() does
// (roaring.go:3412-3461) and its byte size (16-byte aligned).
__global__ void __launch_bounds__(256) k_encode_plan(const Slot* __restrict__ cells, const uint32_t* __restrict__ runs,
                                                    uint64_t n_slots, uint32_t* __restrict__ enc_type,
                                                    u64* __restrict__ enc_bytes) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_slots) return;
  const uint32_t n = slot_n(cells[i]);
  uint32_t t = kTypeNil;
  u64 bytes = 0;
  if (n != 0) {
    const uint32_t r = runs[i];
    if (r <= 2048u && r <= n / 2) {
      t = kTypeRun;
      bytes = (u64)r * 4;
    } else if (n < 4096u) {
      t = kTypeArray;
      bytes = (u64)n * 2;
    } else {
      t = kTypeBitmap;
      bytes = 8192;
    }
  }
  enc_type[i] = t;
  enc_bytes[i] = (bytes + 15) & ~15ull;
}

// Phase 2: exclusive prefix sum of the byte sizes, two levels: k_scan_blocks scans chunks of 1024
// cells in parallel (one block each: local exclusive offsets + the chunk's total), k_exclusive_scan
// — a single block, chunked — then only scans the per-chunk totals (n_slots / 1024 values; it used
// to walk all n_slots cells, 3 barriers per 1024 of them: milliseconds at 8 M cells), and
// k_encode_write adds the two.  total[0] receives the arena size.
__global__ void __launch_bounds__(1024) k_scan_blocks(const u64* __restrict__ in, u64* __restrict__ out_local, uint64_t n,
                                                     u64* __restrict__ block_sums) {
  __shared__ u64 wsum[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const u64 v = i < n ? in[i] : 0;
  u64 incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u64 t = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  u64 woff = 0;
  for (int k = 0; k < wv; ++k) woff += wsum[k];
  if (i < n) out_local[i] = woff + incl - v;
  if (threadIdx.x == 1023) block_sums[blockIdx.x] = woff + incl;
}

__global__ void __launch_bounds__(1024) k_exclusive_scan(const u64* __restrict__ in, u64* __restrict__ out, uint64_t n,
                                                        u64* __restrict__ total) {
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n; base += 1024) {
    const uint64_t i = base + threadIdx.x;
    u64 v = i < n ? in[i] : 0;
    u64 incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      u64 t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u64 woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    const u64 carry = carry_s;
    if (i < n) out[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

// ---- compaction of a batch whose containers are ALREADY in their final encodings (the kernels that apply optimize()
// themselves write the encoded bytes into the head of an 8 KiB cell): the payload size of every container as it is,
// the same two-level scan, then one wave per container copies its bytes.  No decode, no re-encode.
__global__ void __launch_bounds__(256) k_compact_plan(const Slot* __restrict__ cells, uint64_t n_slots, u64* __restrict__ bytes) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_slots) return;
  const Slot s = cells[i];
  const uint32_t t = slot_n(s) ? slot_type(s) : kTypeNil;
  const u64 b = t == kTypeArray ? (u64)s.len * 2 : t == kTypeRun ? (u64)s.len * 4 : t == kTypeBitmap ? 8192ull : 0ull;
  bytes[i] = (b + 15) & ~15ull;
}

__global__ void __launch_bounds__(256) k_compact_write(const Slot* __restrict__ cells, const uint8_t* __restrict__ arena, const u64* __restrict__ bytes,
                                                      const u64* __restrict__ off_local, const u64* __restrict__ off_block, uint64_t n_slots,
                                                      uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots) {
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_slots) return;
  Slot s = cells[i];
  const u64 nb = bytes[i];  // a multiple of 16; payloads are 16-byte aligned in both arenas
  const u64 off = off_block[i >> 10] + off_local[i];
  if (nb == 0) {
    s.off = 0, s.len = 0, s.tn = 0;
  } else {
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(arena + s.off);
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(arenaO + off);
    for (u64 k = lane; k < nb / 16; k += kWave) dst[k] = src[k];
    s.off = off;
  }
  if (lane == 0) outSlots[i] = s;
}

// Phase 3: one wave per cell re-encodes its bitmap cell into the compact arena
// (bitmapToArray roaring.go:3687, bitmapToRun :3859) or copies it.
__global__ void __launch_bounds__(256) k_encode_write(const Slot* __restrict__ cells, const uint8_t* __restrict__ cell_arena,
                                                     const uint32_t* __restrict__ enc_type, const u64* __restrict__ enc_off_local,
                                                     const u64* __restrict__ enc_off_block, const uint32_t* __restrict__ runs,
                                                     uint64_t n_slots, uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots) {
  __shared__ u64 lds[4][kWords];  // decode scratch: a cell may already be an array (k_setop's right-sized outputs)
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_slots) return;
  const uint32_t t = enc_type[i];
  Slot so;
  so.off = 0;
  so.len = 0;
  so.tn = 0;
  if (t == kTypeNil) {
    if (lane == 0) outSlots[i] = so;
    return;
  }
  const Slot cell = cells[i];
  const uint32_t n = slot_n(cell);
  const u64 my_off = enc_off_local[i] + enc_off_block[i >> 10];
  uint8_t* dst = arenaO + my_off;
  so.off = my_off;
  so.tn = make_tn(t, n);
  if (t == kTypeArray && slot_type(cell) == kTypeArray) {
    // the cell already holds the array (k_setop's right-sized outputs): the re-encode is a copy of 2 n bytes
    // (cells and encoded payloads are 16-byte aligned and padded)
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(cell_arena + cell.off);
    ulonglong2* d16 = reinterpret_cast<ulonglong2*>(dst);
    for (uint32_t k = lane; k < (2u * n + 15u) / 16u; k += kWave) d16[k] = src[k];
    so.len = n;
    if (lane == 0) outSlots[i] = so;
    return;
  }
  u64 w[kWordsPerLane];
  frag_load(cell, cell_arena, lane, lds[threadIdx.x >> 6], w);
  if (t == kTypeBitmap) {
    frag_store_bitmap(dst, lane, w);
    so.len = kWords;
  } else if (t == kTypeArray) {
    uint16_t* out = reinterpret_cast<uint16_t*>(dst);
    uint32_t basecnt = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      u64 w0 = w[2 * j], w1 = w[2 * j + 1];
      const uint32_t c0 = __popcll(w0), c = c0 + __popcll(w1);
      const uint32_t incl = wave_incl_scan(c);
      uint32_t at = basecnt + incl - c;
      const uint32_t v0 = (128u * j + 2u * lane) * 64u;
      while (w0) {
        out[at++] = (uint16_t)(v0 + __builtin_ctzll(w0));
        w0 &= w0 - 1;
      }
      while (w1) {
        out[at++] = (uint16_t)(v0 + 64u + __builtin_ctzll(w1));
        w1 &= w1 - 1;
      }
      basecnt += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    so.len = n;
  } else {  // run
    uint16_t* out = reinterpret_cast<uint16_t*>(dst);  // {start,last} pairs: start at 2k, last at 2k+1
    uint32_t sbase = 0, ebase = 0;
    uint32_t prev_top = 0;  // top bit of the word before this iteration's first word
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u64 w0 = w[2 * j], w1 = w[2 * j + 1];
      const uint32_t top1 = (uint32_t)(w1 >> 63);
      uint32_t left = __shfl_up(top1, 1, kWave);
      if (lane == 0) left = prev_top;
      // bit 0 of the word after this lane's w1: next lane's w0, or next iteration's lane-0 w0
      uint32_t next0 = (uint32_t)(__shfl_down(w0, 1, kWave) & 1ull);
      const u64 nxt_iter_w0 = (j < 7) ? w[2 * j + 2 > 15 ? 15 : 2 * j + 2] : 0ull;
      const uint32_t n0 = (uint32_t)(__shfl(nxt_iter_w0, 0, kWave) & 1ull);
      if (lane == 63) next0 = (j < 7) ? n0 : 0u;
      u64 st0 = w0 & ~((w0 << 1) | (u64)left);
      u64 st1 = w1 & ~((w1 << 1) | (w0 >> 63));
      u64 en0 = w0 & ~((w0 >> 1) | ((w1 & 1ull) << 63));
      u64 en1 = w1 & ~((w1 >> 1) | ((u64)next0 << 63));
      const uint32_t cs = __popcll(st0) + __popcll(st1), ce = __popcll(en0) + __popcll(en1);
      const uint32_t is = wave_incl_scan(cs), ie = wave_incl_scan(ce);
      uint32_t as = sbase + is - cs, ae = ebase + ie - ce;
      const uint32_t v0 = (128u * j + 2u * lane) * 64u;
      while (st0) {
        out[2 * (as++)] = (uint16_t)(v0 + __builtin_ctzll(st0));
        st0 &= st0 - 1;
      }
      while (st1) {
        out[2 * (as++)] = (uint16_t)(v0 + 64u + __builtin_ctzll(st1));
        st1 &= st1 - 1;
      }
      while (en0) {
        out[2 * (ae++) + 1] = (uint16_t)(v0 + __builtin_ctzll(en0));
        en0 &= en0 - 1;
      }
      while (en1) {
        out[2 * (ae++) + 1] = (uint16_t)(v0 + 64u + __builtin_ctzll(en1));
        en1 &= en1 - 1;
      }
      sbase += (uint32_t)__builtin_amdgcn_readlane((int)is, 63);
      ebase += (uint32_t)__builtin_amdgcn_readlane((int)ie, 63);
      prev_top = __shfl(top1, 63, kWave);
    }
    so.len = runs[i];
  }
  if (lane == 0) outSlots[i] = so;
}

// ---- Shift: every column of a row moves up by one ---------------------------------------------
// Row.Shift / RowSegment.Shift / Bitmap.Shift(1) (row.go:374-396, 613-626; roaring.go:1629-1662,
// shiftArray / shiftBitmap / shiftRun :6184-6257).  The reference shifts container by container
// and carries the bit that leaves value 65535 into value 0 of the next key — including, at the end
// of a shard's segment, into a container keyed one past the segment ("TODO: deal with overflow"),
// which Row.Columns() then reports as column (shard + 1) * ShardWidth.  Observable result: the
// shift of the whole column space; the carry row below is how that bit gets into the next shard.
//
// bit 65535 of a container (its carry out), from the encoded form
__device__ __forceinline__ bool slot_top_bit(const Slot& s, const uint8_t* __restrict__ arena) {
  const uint32_t n = slot_n(s);
  if (n == 0) return false;
  if (n == 65536u) return true;
  const uint8_t* p = arena + s.off;
  if (slot_type(s) == kTypeBitmap) return (reinterpret_cast<const u64*>(p)[kWords - 1] >> 63) != 0;
  if (slot_type(s) == kTypeArray) return reinterpret_cast<const uint16_t*>(p)[s.len - 1] == 0xFFFFu;
  return reinterpret_cast<const uint16_t*>(p)[2 * s.len - 1] == 0xFFFFu;  // `last` of the last interval
}

// One wavefront per (output row i, slot): out = rows[i] << 1, with bit 0 of slot 0 taken from bit
// 65535 of slot 15 of carry_rows[i] (the row of the previous shard); either index may be
// kNoRow (an absent row / no predecessor).
constexpr uint32_t kNoRow = 0xFFFFFFFFu;
__global__ void __launch_bounds__(256) k_shift(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                              const uint32_t* __restrict__ rows, const uint32_t* __restrict__ carry_rows,
                                              uint64_t n_rows, uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots,
                                              uint32_t* __restrict__ outRuns, u64* __restrict__ out_counts, uint32_t encode) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  const uint64_t i = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (i >= n_rows) return;
  const uint32_t row = rows[i], crow = carry_rows ? carry_rows[i] : kNoRow;
  Slot s, sp;
  s.off = sp.off = 0;
  s.len = sp.len = 0;
  s.tn = sp.tn = 0;
  if (row != kNoRow) s = slots[(uint64_t)row * kSlots + slot];
  if (slot > 0) {
    if (row != kNoRow) sp = slots[(uint64_t)row * kSlots + slot - 1];
  } else if (crow != kNoRow) {
    sp = slots[(uint64_t)crow * kSlots + (kSlots - 1)];
  }
  const bool carry_in = slot_top_bit(sp, arena);
  Slot so;
  so.off = wslot * 8192ull;
  so.len = kWords;
  so.tn = 0;
  if (slot_n(s) == 0 && !carry_in) {
    if (lane == 0) {
      outSlots[wslot] = so;
      if (outRuns) outRuns[wslot] = 0;
    }
    return;
  }
  u64 w[kWordsPerLane];
  if (slot_n(s) == 0) frag_zero(w);
  else frag_load(s, arena, lane, lds[wv], w);
  // lane l holds words 128j + 2l (w[2j]) and 128j + 2l + 1 (w[2j+1]); the bit shifted into word k
  // is the top bit of word k - 1
  u64 top_odd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) top_odd[j] = w[2 * j + 1] >> 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u64 in = (u64)__shfl_up((unsigned)top_odd[j], 1, kWave);                 // word 128j + 2l - 1 (lane l - 1)
    const u64 wrap = j ? (u64)__shfl((unsigned)top_odd[j - 1], 63, kWave) : (carry_in ? 1ull : 0ull);  // word 128j - 1
    if (lane == 0) in = wrap;
    const u64 even = w[2 * j];
    w[2 * j] = (even << 1) | in;
    w[2 * j + 1] = (w[2 * j + 1] << 1) | (even >> 63);
  }
  const uint32_t c = wave_reduce_add(frag_popcount(w));
  uint32_t r = 0;
  if (outRuns || encode) r = wave_reduce_add(frag_count_runs(w, lane));
  uint32_t t_out = c ? kTypeBitmap : kTypeNil, l_out = kWords;
  if (encode && c) frag_store_encoded(w, c, r, lane, lds[wv], arenaO + so.off, t_out, l_out);  // optimize() applied here
  else if (c) frag_store_bitmap(arenaO + so.off, lane, w);
  if (lane == 0) {
    so.len = l_out;
    so.tn = make_tn(t_out, c);
    outSlots[wslot] = so;
    if (outRuns) outRuns[wslot] = r;
    if (out_counts && c) atomicAdd(&out_counts[i], (u64)c);
  }
}

// ---- Flip: negate the bits of a row inside [start, end] --------------------------------------
// Bitmap.Flip(start, end) (roaring.go:2769-2799; the container-level flipArray / flipBitmap / flipRun
// of roaring.go:6259-6274 are the case start = 0, end = 65535 of one slot).  One wavefront per
// (row, slot): decode, XOR with the slot's share of the range mask, write a bitmap cell.
__global__ void __launch_bounds__(256) k_flip(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                             const uint32_t* __restrict__ rows, uint64_t n_rows, uint32_t start, uint32_t end_incl,
                                             uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                             u64* __restrict__ out_counts, uint32_t encode) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  const uint64_t i = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (i >= n_rows) return;
  const Slot s = slots[(uint64_t)rows[i] * kSlots + slot];
  const uint32_t base = slot << 16;
  // the part of [start, end_incl] that falls into this slot, as [lo, hi) relative to the slot
  const uint32_t lo = start > base ? min(start - base, 65536u) : 0u;
  const uint32_t hi = end_incl + 1u > base ? min(end_incl + 1u - base, 65536u) : 0u;
  Slot so;
  so.off = wslot * 8192ull;
  so.len = kWords;
  so.tn = 0;
  u64 w[kWordsPerLane];
  if (slot_n(s) == 0) frag_zero(w);
  else frag_load(s, arena, lane, lds[wv], w);
  if (lo < hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t b0 = (128u * j + 2u * lane + h) * 64u;  // first bit of this word
        u64 m = ~0ull;
        if (lo > b0) m = (lo - b0 >= 64u) ? 0ull : (m << (lo - b0));
        if (hi < b0 + 64u) m = (hi <= b0) ? 0ull : (m & (~0ull >> (b0 + 64u - hi)));
        w[2 * j + h] ^= m;
      }
  }
  const uint32_t c = wave_reduce_add(frag_popcount(w));
  uint32_t r = 0;
  if (outRuns || encode) r = wave_reduce_add(frag_count_runs(w, lane));
  uint32_t t_out = c ? kTypeBitmap : kTypeNil, l_out = kWords;
  if (encode && c) frag_store_encoded(w, c, r, lane, lds[wv], arenaO + so.off, t_out, l_out);  // optimize() applied here
  else if (c) frag_store_bitmap(arenaO + so.off, lane, w);
  if (lane == 0) {
    so.len = l_out;
    so.tn = make_tn(t_out, c);
    outSlots[wslot] = so;
    if (outRuns) outRuns[wslot] = r;
    if (out_counts && c) atomicAdd(&out_counts[i], (u64)c);
  }
}

// ---- TopN qualification (fragment.top's threshold rules, fragment.go:1329-1402) ------------------
// counts[s][i] = |row i of shard s ∩ src_s| (the row's cardinality when there is no source row),
// cards[s][i] = the row's cardinality, src_counts[s] = |src_s|.  A row that does not qualify in a
// shard contributes nothing from that shard (its count is zeroed).  Integer forms of the
// reference's float64 comparisons: cnt <= src*T/100 <=> cnt*100 <= src*T; cnt >= src*100/T <=>
// cnt*T >= src*100; ceil(count*100 / (cnt + src - count)) is the integer ceiling division.
__global__ void __launch_bounds__(256) k_topn_filter(u64* __restrict__ counts, const u64* __restrict__ cards,
                                                    const u64* __restrict__ src_counts, uint64_t n, uint32_t n_a, u64 min_threshold,
                                                    u64 tanimoto_threshold) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 cnt = cards[i], count = counts[i];
  bool ok = cnt != 0 && count != 0;
  if (ok) {
    if (tanimoto_threshold > 0 && src_counts) {
      const u64 src = src_counts[i / n_a];
      if (cnt * 100 <= src * tanimoto_threshold || cnt * tanimoto_threshold >= src * 100) ok = false;
      else {
        const u64 den = cnt + src - count;
        ok = (count * 100 + den - 1) / den > tanimoto_threshold;
      }
    } else {
      ok = cnt >= min_threshold && count >= min_threshold;
    }
  }
  if (!ok) counts[i] = 0;
}

// ---- TopN candidates: pass 1 of executeTopN (executor.go:2779-2864), per SHARD ------------------------
// fragment.top (fragment.go:1317-1437) with N = n walks the shard's rows in rank-cache order (cardinality descending; rows
// of one cardinality in row-index order — Go sorts the cache with an unstable sort, the tie order is fixed here): the first n rows that pass
// the thresholds fill the heap ("P"); with a source row every LATER row whose cardinality passes the cnt-level test and
// whose count reaches T = the smallest count in P is pushed as well, without evicting (:1404-1425: the heap only grows,
// so its minimum stays T; "cnt < threshold -> break" skips rows that could not reach T anyway); without a source row the
// walk stops at n (:1398-1401).  Every id a shard returns is a candidate of the second pass (:2812-2818).
//
// One block per shard, no sort: the n-th qualifying row in rank order is found by two-level histogram selection on
// (2^22 - 1 - cnt) and then, among the rows of that cardinality, on the row index (both < 2^22: 2048 x 2048 bins).
// counts[s][i] = |row i ∩ src_s| BEFORE k_topn_filter (= cards without a source row), cards[s][i] = the row's
// cardinality, src_counts[s] = |src_s|.  cand[i] is set to 1 for every candidate row (all shards write the same value).
struct TopnRule {
  u64 min_threshold, tanimoto, src;
  bool tani;
  __device__ __forceinline__ bool cnt_ok(u64 cnt) const {
    if (cnt == 0) return false;
    if (tani) return !(cnt * 100 <= src * tanimoto || cnt * tanimoto >= src * 100);
    return cnt >= min_threshold;
  }
  __device__ __forceinline__ bool count_ok(u64 cnt, u64 count) const {
    if (count == 0) return false;
    if (tani) {
      const u64 den = cnt + src - count;
      return (count * 100 + den - 1) / den > tanimoto;
    }
    return count >= min_threshold;
  }
};

// the k-th smallest (k >= 1) 22-bit key among the rows i < n_a with pred(i); returns the key, *below = the number of
// flagged rows with a smaller key.  Requires k <= the number of flagged rows.  hist: 2048 uint32 of LDS; sel: 4 uint32.
template <typename Pred, typename Key>
__device__ __forceinline__ uint32_t block_kth_smallest(uint32_t n_a, uint32_t k, Pred pred, Key key, uint32_t* hist, uint32_t* sel, uint32_t* below) {
  uint32_t base = 0, prefix = 0;  // rows below the current bin; the key bits fixed so far
  for (int level = 0; level < 2; ++level) {
    for (uint32_t b = threadIdx.x; b < 2048; b += 256) hist[b] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_a; i += 256) {
      if (!pred(i)) continue;
      const uint32_t kk = key(i);
      if (level == 0) atomicAdd(&hist[kk >> 11], 1u);
      else if ((kk >> 11) == prefix) atomicAdd(&hist[kk & 2047], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: lane l owns bins [32 l, 32 l + 32)
      const int lane = threadIdx.x;
      uint32_t mine = 0;
      for (int j = 0; j < 32; ++j) mine += hist[lane * 32 + j];
      uint32_t incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += t;
      }
      const uint32_t excl = incl - mine;
      if (base + excl < k && k <= base + incl) {  // exactly one lane
        uint32_t run = base + excl;
        for (int j = 0; j < 32; ++j) {
          const uint32_t h = hist[lane * 32 + j];
          if (k <= run + h) {
            sel[0] = uint32_t(lane * 32 + j);
            sel[1] = run;
            break;
          }
          run += h;
        }
      }
    }
    __syncthreads();
    const uint32_t bin = sel[0];
    base = sel[1];
    __syncthreads();
    prefix = level == 0 ? bin : ((prefix << 11) | bin);
  }
  *below = base;
  return prefix;
}

__global__ void __launch_bounds__(256) k_topn_candidates(const u64* __restrict__ counts, const u64* __restrict__ cards, const u64* __restrict__ src_counts,
                                                        uint32_t n_a, uint32_t top_n, u64 min_threshold, u64 tanimoto_threshold, int has_src,
                                                        u64* __restrict__ cand) {
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sel[4];
  __shared__ unsigned long long red[4];
  const uint64_t s = blockIdx.x;
  counts += s * n_a;
  cards += s * n_a;
  TopnRule rule;
  rule.min_threshold = min_threshold;
  rule.tanimoto = tanimoto_threshold;
  rule.src = (has_src && src_counts) ? src_counts[s] : 0;
  rule.tani = tanimoto_threshold > 0 && has_src;
  constexpr uint32_t kTop = (1u << 22) - 1;
  auto qualifies = [&](uint32_t i) {
    const u64 cnt = cards[i];
    return rule.cnt_ok(cnt) && rule.count_ok(cnt, counts[i]);
  };
  // ---- how many rows qualify at all
  uint32_t q = 0;
  for (uint32_t i = threadIdx.x; i < n_a; i += 256) q += qualifies(i) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += (uint32_t)__shfl_xor((int)q, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  const uint32_t n_qual = uint32_t(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  if (n_qual < top_n) {  // the heap never fills: every qualifying row is returned, nothing else is looked at
    for (uint32_t i = threadIdx.x; i < n_a; i += 256)
      if (qualifies(i)) cand[i] = 1;
    return;
  }
  // ---- the n-th qualifying row in rank order: cardinality v, and among the rows of cardinality v the row index id_cut
  uint32_t above = 0;
  const uint32_t vkey = block_kth_smallest(n_a, top_n, qualifies, [&](uint32_t i) { return kTop - uint32_t(cards[i]); }, hist, sel, &above);
  const u64 v = kTop - vkey;
  uint32_t dummy = 0;
  const uint32_t id_cut = block_kth_smallest(n_a, top_n - above, [&](uint32_t i) { return cards[i] == v && qualifies(i); }, [&](uint32_t i) { return i; }, hist, sel, &dummy);
  auto in_p = [&](uint32_t i) {
    const u64 cnt = cards[i];
    return (cnt > v || (cnt == v && i <= id_cut)) && qualifies(i);
  };
  // ---- T = the smallest count in P
  unsigned long long t = ~0ull;
  for (uint32_t i = threadIdx.x; i < n_a; i += 256)
    if (in_p(i)) t = counts[i] < t ? counts[i] : t;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long x = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(t >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)t, o, 64);
    t = x < t ? x : t;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  unsigned long long T = red[0];
  for (int w = 1; w < 4; ++w) T = red[w] < T ? red[w] : T;
  const bool later = has_src && T >= min_threshold;  // (:1409: "threshold < MinThreshold -> break")
  for (uint32_t i = threadIdx.x; i < n_a; i += 256) {
    const u64 cnt = cards[i];
    bool c = in_p(i);
    if (!c && later && (cnt < v || (cnt == v && i > id_cut))) c = rule.cnt_ok(cnt) && counts[i] >= T;
    if (c) cand[i] = 1;
  }
}

// totals of the rows that are no candidate of any shard do not exist in the reference's second pass
__global__ void __launch_bounds__(256) k_topn_mask(u64* __restrict__ totals, const u64* __restrict__ cand, uint32_t n_a) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_a && cand[i] == 0) totals[i] = 0;
}

__global__ void __launch_bounds__(256) k_max_u64(const u64* __restrict__ v, uint64_t n, u64* __restrict__ out) {
  u64 m = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) m = v[i] > m ? v[i] : m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 t = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), o, kWave) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)m, o, kWave);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ---- counts -> BSI planes (bsiBuilder.Insert(rowID, count), bsi.go:251-284) ------------------------
// totals[i] = count of row i; plane p (output row p) gets bit i iff bit p of totals[i] is set.  One
// wavefront per 64 row ids: 64 ballots give the 64-bit word of every plane.  Cells are 8 KiB bitmap
// cells (zeroed by the caller); k_cell_stats fills cardinalities and run counts afterwards.
__global__ void __launch_bounds__(256) k_counts_to_bsi(const u64* __restrict__ totals, uint32_t n_a, uint32_t depth,
                                                      uint8_t* __restrict__ arenaO) {
  const int lane = threadIdx.x & 63;
  const uint32_t wordi = blockIdx.x * 4 + (threadIdx.x >> 6);  // which 64 row ids
  const uint32_t i = wordi * 64 + lane;
  if (wordi * 64 >= n_a) return;
  const u64 v = i < n_a ? totals[i] : 0;
  for (uint32_t p = 0; p < depth; ++p) {
    const u64 word = __ballot((v >> p) & 1ull);
    // plane p = output row p; row-relative bit position = row id: slot = i >> 16, word (i & 65535) >> 6
    if (lane == 0 && word)
      reinterpret_cast<u64*>(arenaO + ((uint64_t)p * kSlots + (wordi >> 10)) * 8192ull)[wordi & 1023u] = word;
  }
}

// cardinality and run count of bitmap cells (cell layout: slot s at s * 8 KiB)
__global__ void __launch_bounds__(256) k_cell_stats(uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                                   uint64_t n_slots) {
  const int lane = threadIdx.x & 63;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wslot >= n_slots) return;
  u64 w[kWordsPerLane];
  frag_load_bitmap(arenaO + wslot * 8192ull, lane, w);
  const uint32_t c = wave_reduce_add(frag_popcount(w));
  const uint32_t r = wave_reduce_add(frag_count_runs(w, lane));
  if (lane == 0) {
    Slot so;
    so.off = wslot * 8192ull;
    so.len = kWords;
    so.tn = make_tn(c ? kTypeBitmap : kTypeNil, c);
    outSlots[wslot] = so;
    if (outRuns) outRuns[wslot] = r;
  }
}


// ---- fragment.rows: which rows hold anything / hold a given column ----------------------------
// The reference walks a fragment's containers in key order through the filter protocol of
// roaring/filter.go (BitmapRowFilter over an optional BitmapColumnFilter and BitmapRowLimitFilter,
// ApplyFilterToIterator :1062-1085): a container with n != 0 makes its row a candidate, a column
// filter opens the ONE container of the row that can hold the column and skips the other fifteen by
// key (YesKey / NoKey), a limit filter stops the scan.  On the device every row is looked at at once:
// 16 lanes per row, lane = slot, read the row's 16 descriptors (one 256-byte line) and vote;
// the lane of the column's slot probes its container (bitmap: one word; array / run: a binary search
// in the payload) — the same one container per row that the protocol would open.
//   flags[i] bit 0: the row holds a container with n != 0; bit 1: ... and contains `column`
//   (column == ~0: no column filter, bit 1 = bit 0); bit 2: it holds a container with n != 0 in the column's slot
//   or a later one (what the scan still sees of a row when it arrives by a skip to the column's slot).
__global__ void __launch_bounds__(256) k_rows_flags(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                   const uint32_t* __restrict__ rows, uint64_t n, uint64_t column,
                                                   uint8_t* __restrict__ flags) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t i = gid >> 4;
  const uint32_t slot = gid & 15u;
  const int lane = threadIdx.x & 63;
  Slot s;
  s.off = 0, s.len = 0, s.tn = 0;
  if (i < n) s = slots[(uint64_t)rows[i] * kSlots + slot];
  const uint32_t cnt = slot_n(s);
  bool hit = false;
  if (column != ~0ull && cnt != 0 && slot == (uint32_t)((column >> 16) & 15u)) {
    const uint32_t v = (uint32_t)(column & 0xFFFFu);
    const uint8_t* p = arena + s.off;
    const uint32_t t = slot_type(s);
    if (t == kTypeBitmap) {
      hit = (reinterpret_cast<const u64*>(p)[v >> 6] >> (v & 63)) & 1ull;
    } else if (t == kTypeArray) {  // Container.Contains -> search64 over the sorted values
      const uint16_t* a = reinterpret_cast<const uint16_t*>(p);
      uint32_t lo = 0, hi = s.len;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      hit = lo < s.len && a[lo] == v;
    } else if (t == kTypeRun) {  // first run whose last value is >= v
      const uint32_t* r = reinterpret_cast<const uint32_t*>(p);
      uint32_t lo = 0, hi = s.len;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((r[mid] >> 16) < v) lo = mid + 1;
        else hi = mid;
      }
      hit = lo < s.len && (r[lo] & 0xFFFFu) <= v;
    }
  }
  const uint32_t cslot = column == ~0ull ? 0u : (uint32_t)((column >> 16) & 15u);
  const u64 ne = __ballot(cnt != 0), hm = __ballot(hit);
  const uint32_t sh = (uint32_t)lane & 48u;  // this row's 16 lanes inside the ballot
  const uint32_t nes = (uint32_t)((ne >> sh) & 0xFFFFull);
  const bool nonempty = nes != 0, contains = ((hm >> sh) & 0xFFFFull) != 0, tail = (nes >> cslot) != 0;
  if (slot == 0 && i < n) flags[i] = (uint8_t)((nonempty ? 1u : 0u) | ((nonempty && (column == ~0ull || contains)) ? 2u : 0u) | (tail ? 4u : 0u));
}

// limit filter (BitmapRowLimitFilter, filter.go:471-509, as executeRowsShard appends it AFTER the column filter,
// executor.go:4139-4155).  BitmapRowFilterMultiFilter.ConsiderKey (filter.go:603-622) consults every undecided
// filter for the key the scan is at, so the limit filter spends one of its rows on every row in which the scan
// LOOKS AT a container, whether or not the column filter goes on to match the row.  Which rows are those: after a
// row r whose container in the column's slot c (or a later slot) was looked at, the column filter's answer skips
// the scan to key (r + 1, c) — the containers of row r + 1 in slots below c are never presented.  So a row counts
// unless ALL its containers lie below slot c AND the row with the directly preceding id is a candidate with a
// container in slot >= c.  (Such a row cannot hold the column either.)  Without a column filter every non-empty row
// counts.  rank[i] = number of counted rows before i (exclusive scan of RowCounted).
__global__ void __launch_bounds__(256) k_rows_select(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ rank, uint64_t n,
                                                    uint64_t limit, uint8_t* __restrict__ keep) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keep[i] = (uint8_t)(((flags[i] & 2u) != 0 && (limit == 0 || rank[i] < limit)) ? 1 : 0);
}

struct RowCounted {
  const uint8_t* flags;
  const uint64_t* ids;  // ids of the candidate rows (NULL without a column filter: adjacency does not matter then)
  __host__ __device__ uint32_t operator()(uint32_t i) const {
    const uint32_t f = flags[i];
    if (!(f & 1u)) return 0u;
    if (ids && i > 0 && !(f & 4u) && (flags[i - 1] & 4u) && ids[i - 1] + 1 == ids[i]) return 0u;
    return 1u;
  }
};

}  // namespace fbk
