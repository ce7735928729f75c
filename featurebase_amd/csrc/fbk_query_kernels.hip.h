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

// ---- n-way union ------------------------------------------------------------------------
// One wave per (group, slot): OR-accumulates the k containers of the group's rows at that
// slot in registers (the union never touches HBM unless WRITE), mirroring what
// BitmapRowsUnion does with its 16 accumulators (filter.go:327-334) and what the n-way
// Bitmap.unionInPlace does per key (roaring.go:1455-1560).  Short-circuit: any full
// operand => full container (roaring.go:1465-1474).
// With a filter batch: counts[g] += |union ∩ F.rows_f[g]| (Bitmap.IntersectionCount of
// the union against the filter row), else counts[g] += |union|.
template <bool WRITE>
__global__ void __launch_bounds__(256) k_union_n(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                const uint32_t* __restrict__ rows, uint64_t n_groups, uint32_t k,
                                                const Slot* __restrict__ fslots, const uint8_t* __restrict__ farena,
                                                const uint32_t* __restrict__ frows, uint8_t* __restrict__ arenaO,
                                                Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                                u64* __restrict__ out_counts) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  const uint64_t g = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (g >= n_groups) return;
  u64 acc[kWordsPerLane];
  frag_zero(acc);
  bool full = false;
  const uint32_t* grow = rows + g * k;
  for (uint32_t i = 0; i < k; ++i) {
    const Slot s = slots[(uint64_t)grow[i] * kSlots + slot];
    const uint32_t n = slot_n(s);
    if (n == 0) continue;
    if (n == 65536u) {
      full = true;
      break;
    }
    u64 w[kWordsPerLane];
    frag_load(s, arena, lane, lds[wv], w);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) acc[q] |= w[q];
  }
  if (full) {
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) acc[q] = ~0ull;
  }
  uint32_t c;
  if (fslots) {
    const Slot sf = fslots[(uint64_t)frows[g] * kSlots + slot];
    if (slot_n(sf) == 0) {
      c = 0;
    } else {
      u64 w[kWordsPerLane];
      frag_load(sf, farena, lane, lds[wv], w);
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
    so.off = wslot * 8192ull;
    so.len = kWords;
    so.tn = make_tn(cu ? kTypeBitmap : kTypeNil, cu);
    if (cu) frag_store_bitmap(arenaO + so.off, lane, acc);
    uint32_t r = 0;
    if (outRuns) r = wave_reduce_add(frag_count_runs(acc, lane));
    if (lane == 0) {
      outSlots[wslot] = so;
      if (outRuns) outRuns[wslot] = r;
    }
  }
  if (lane == 0 && c && out_counts) atomicAdd(&out_counts[g], (u64)c);
}

// ---- count matrix (GroupBy / TopK / TopN shape) -----------------------------------------------
// out_shard[(shard*nA + i)*nB + j] = sum over the 16 slots of |A[shard][i] ∩ B[shard][j] ∩ F[shard]|.
// One 256-thread block per (shard, tile of TA A-rows); wave w owns slots w, w+4, w+8, w+12.
// Per slot the wave keeps TA A-fragments (already ANDed with the filter) in registers and
// streams the nB B-containers past them, so every B container is read by nA/TA blocks
// (they are scheduled on the same XCD so the re-reads hit its L2) and every A container
// exactly once.  The block->work mapping keeps all blocks of one shard on one XCD
// (block b runs on XCD b % 8 on MI355X; used for L2 affinity only, never for correctness).
template <int TA>
__global__ void __launch_bounds__(256) k_count_matrix(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                                     const uint32_t* __restrict__ rowsA, uint32_t nA,
                                                     const Slot* __restrict__ slotsB, const uint8_t* __restrict__ arenaB,
                                                     const uint32_t* __restrict__ rowsB, uint32_t nB,
                                                     const Slot* __restrict__ slotsF, const uint8_t* __restrict__ arenaF,
                                                     const uint32_t* __restrict__ rowsF, uint32_t n_shards,
                                                     u64* __restrict__ out_shard) {
  __shared__ u64 lds[4][kWords];
  extern __shared__ uint32_t cnt[];  // TA * nB block-level counters
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint32_t tiles = (nA + TA - 1) / TA;
  // XCD-aware remap: x = b % 8 picks the shard residue class, so all tiles of a shard
  // share b % 8.  Shards beyond the last full group of 8 are handled by the same formula.
  const uint32_t b = blockIdx.x;
  const uint32_t x = b & 7u, t = b >> 3;
  const uint32_t shard = (t / tiles) * 8u + x;
  const uint32_t tile = t % tiles;
  if (shard >= n_shards) return;
  for (uint32_t q = threadIdx.x; q < TA * nB; q += 256) cnt[q] = 0;
  __syncthreads();
  const uint32_t i0 = tile * TA;
  for (uint32_t slot = wv; slot < kSlots; slot += 4) {
    u64 fa[TA][kWordsPerLane];
    bool any = false;
    // filter fragment first (intersected into every A row: rows[0] ∩= filter, executor.go:8830)
    bool have_f = slotsF != nullptr;
    u64 wf[kWordsPerLane];
    if (have_f) {
      const Slot sf = slotsF[(uint64_t)rowsF[shard] * kSlots + slot];
      if (slot_n(sf) == 0) continue;  // nothing can intersect at this slot
      frag_load(sf, arenaF, lane, lds[wv], wf);
    }
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      bool present = false;
      if (i0 + a < nA) {
        const Slot sa = slotsA[(uint64_t)rowsA[(uint64_t)shard * nA + i0 + a] * kSlots + slot];
        if (slot_n(sa) != 0) {
          frag_load(sa, arenaA, lane, lds[wv], fa[a]);
          present = true;
        }
      }
      if (!present) frag_zero(fa[a]);
      if (have_f) {
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) fa[a][q] &= wf[q];
      }
      any |= present;
    }
    if (!any) continue;
    for (uint32_t j = 0; j < nB; ++j) {
      const Slot sb = slotsB[(uint64_t)rowsB[(uint64_t)shard * nB + j] * kSlots + slot];
      if (slot_n(sb) == 0) continue;
      u64 wb[kWordsPerLane];
      frag_load(sb, arenaB, lane, lds[wv], wb);
      uint32_t part[TA];
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        uint32_t p = 0;
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) p += __popcll(fa[a][q] & wb[q]);
        part[a] = p;
      }
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        uint32_t c = wave_reduce_add(part[a]);
        if (lane == 0 && c) atomicAdd(&cnt[a * nB + j], c);
      }
    }
  }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < TA * nB; q += 256) {
    const uint32_t a = q / nB, j = q % nB;
    if (i0 + a < nA) out_shard[((uint64_t)shard * nA + i0 + a) * nB + j] = cnt[q];
  }
}

// out[c] = sum over shards of in[shard*width + c]   (mergeGroupCounts' arithmetic, executor.go:3728)
__global__ void __launch_bounds__(256) k_reduce_shards(const u64* __restrict__ in, uint32_t n_shards, uint64_t width,
                                                      u64* __restrict__ out) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  u64 acc = 0;
  for (uint32_t s = 0; s < n_shards; ++s) acc += in[(uint64_t)s * width + c];
  out[c] = acc;
}

// ---- BSI Sum ---------------------------------------------------------------------------------
// One wave per (shard, slot).  positive = filter ∩ exists \ sign, negative = filter ∩ exists ∩ sign
// stay in registers while the bit planes stream past once:
//   psum += |positive ∩ plane_i| << i ; nsum += |negative ∩ plane_i| << i   (uint64 wrap-around,
// roaring/filter.go:1157-1160).  Rows of the BSI fragment of shard s are base[s] + {0: exists,
// 1: sign, 2+i: bit i} (fragment.go:62-65).  out3[shard] = {psum, nsum, count}.
__global__ void __launch_bounds__(256) k_bsi_sum(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                const uint32_t* __restrict__ base, uint32_t n_shards,
                                                uint32_t bit_depth, const Slot* __restrict__ fslots,
                                                const uint8_t* __restrict__ farena, const uint32_t* __restrict__ frows,
                                                u64* __restrict__ out3) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  const uint64_t shard = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (shard >= n_shards) return;
  const uint64_t r0 = base[shard];
  const Slot se = slots[(r0 + 0) * kSlots + slot];
  if (slot_n(se) == 0) return;  // no existence bits: positive stays nil (filter.go:1135)
  u64 pos[kWordsPerLane], neg[kWordsPerLane], w[kWordsPerLane];
  frag_load(se, arena, lane, lds[wv], pos);
  if (fslots) {
    const Slot sf = fslots[(uint64_t)frows[shard] * kSlots + slot];
    if (slot_n(sf) == 0) return;  // ConsiderKey rejects: no filter container here (filter.go:1112)
    frag_load(sf, farena, lane, lds[wv], w);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) pos[q] &= w[q];
  }
  const uint32_t count = wave_reduce_add(frag_popcount(pos));
  const Slot ss = slots[(r0 + 1) * kSlots + slot];
  if (slot_n(ss) != 0) {
    frag_load(ss, arena, lane, lds[wv], w);
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) {
      neg[q] = pos[q] & w[q];
      pos[q] &= ~w[q];
    }
  } else {
    frag_zero(neg);
  }
  u64 psum = 0, nsum = 0;
  for (uint32_t i = 0; i < bit_depth; ++i) {
    const Slot sp = slots[(r0 + 2 + i) * kSlots + slot];
    if (slot_n(sp) == 0) continue;
    frag_load(sp, arena, lane, lds[wv], w);
    uint32_t pc = 0, nc = 0;
#pragma unroll
    for (int q = 0; q < kWordsPerLane; ++q) {
      pc += __popcll(pos[q] & w[q]);
      nc += __popcll(neg[q] & w[q]);
    }
    psum += (u64)pc << i;
    nsum += (u64)nc << i;
  }
  psum = wave_reduce_add64(psum);
  nsum = wave_reduce_add64(nsum);
  if (lane == 0) {
    if (psum) atomicAdd(&out3[shard * 3 + 0], psum);
    if (nsum) atomicAdd(&out3[shard * 3 + 1], nsum);
    if (count) atomicAdd(&out3[shard * 3 + 2], (u64)count);
  }
}

// ---- BSI Range: plane-program interpreter -----------------------------------------------------
// The host walks the reference's control flow (rangeEQ/LT/GT/Between, fragment.go:963-1303)
// ONCE per query and emits a short straight-line program over three fragment registers
// X (remaining / result), M (matched), S (saved); every (shard, slot) wave then runs the
// same program with its own containers, each bit plane read at most once per pass.
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
  kAndnXS = 10  // X &= ~S
};

__global__ void __launch_bounds__(256) k_bsi_range(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                  const uint32_t* __restrict__ base, uint32_t n_shards,
                                                  const uint32_t* __restrict__ prog, uint32_t prog_len,
                                                  uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots,
                                                  uint32_t* __restrict__ outRuns, u64* __restrict__ out_counts) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  const uint64_t shard = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (shard >= n_shards) return;
  const uint64_t r0 = base[shard];
  u64 X[kWordsPerLane], M[kWordsPerLane], S[kWordsPerLane], T[kWordsPerLane];
  frag_zero(X);
  frag_zero(M);
  frag_zero(S);
  for (uint32_t pc = 0; pc < prog_len; ++pc) {
    const uint32_t ins = prog[pc];
    const uint32_t op = ins >> 24, r = ins & 0xFFFFFFu;
    if (op <= kMorXAn) {
      const Slot s = slots[(r0 + r) * kSlots + slot];
      if (slot_n(s) == 0) frag_zero(T);
      else frag_load(s, arena, lane, lds[wv], T);
    }
    switch (op) {
      case kLoadX:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) X[q] = T[q];
        break;
      case kAndX:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) X[q] &= T[q];
        break;
      case kAndnX:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) X[q] &= ~T[q];
        break;
      case kMorXA:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) M[q] |= X[q] & T[q];
        break;
      case kMorXAn:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) M[q] |= X[q] & ~T[q];
        break;
      case kMZero: frag_zero(M); break;
      case kXFromM:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) X[q] = M[q];
        break;
      case kZeroX: frag_zero(X); break;
      case kSaveX:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) S[q] = X[q];
        break;
      case kOrXS:
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) X[q] |= S[q];
        break;
      default:  // kAndnXS
#pragma unroll
        for (int q = 0; q < kWordsPerLane; ++q) X[q] &= ~S[q];
        break;
    }
  }
  const uint32_t c = wave_reduce_add(frag_popcount(X));
  Slot so;
  so.off = wslot * 8192ull;
  so.len = kWords;
  so.tn = make_tn(c ? kTypeBitmap : kTypeNil, c);
  if (c) frag_store_bitmap(arenaO + so.off, lane, X);
  uint32_t rr = 0;
  if (outRuns) rr = wave_reduce_add(frag_count_runs(X, lane));
  if (lane == 0) {
    outSlots[wslot] = so;
    if (outRuns) outRuns[wslot] = rr;
    if (c && out_counts) atomicAdd(&out_counts[shard], (u64)c);
  }
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

// Phase 2: exclusive prefix sum of the byte sizes (single block, chunked; n_slots is at most
// a few million).  total[0] receives the arena size.
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

// Phase 3: one wave per cell re-encodes its bitmap cell into the compact arena
// (bitmapToArray roaring.go:3687, bitmapToRun :3859) or copies it.
__global__ void __launch_bounds__(256) k_encode_write(const Slot* __restrict__ cells, const uint8_t* __restrict__ cell_arena,
                                                     const uint32_t* __restrict__ enc_type, const u64* __restrict__ enc_off,
                                                     const uint32_t* __restrict__ runs, uint64_t n_slots,
                                                     uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots) {
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
  u64 w[kWordsPerLane];
  frag_load_bitmap(cell_arena + cell.off, lane, w);
  uint8_t* dst = arenaO + enc_off[i];
  so.off = enc_off[i];
  so.tn = make_tn(t, n);
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
      uint32_t incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t tt = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += tt;
      }
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
      basecnt += __shfl(incl, 63, kWave);
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
      uint32_t is = cs, ie = ce;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t ts = __shfl_up(is, o, kWave), te = __shfl_up(ie, o, kWave);
        if (lane >= o) {
          is += ts;
          ie += te;
        }
      }
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
      sbase += __shfl(is, 63, kWave);
      ebase += __shfl(ie, 63, kWave);
      prev_top = __shfl(top1, 63, kWave);
    }
    so.len = runs[i];
  }
  if (lane == 0) outSlots[i] = so;
}

}  // namespace fbk
