// fbk_fold_kernels.hip.h — n-way Union / Xor / Difference of k rows with a SHARED LDS
// accumulator (the n-way Intersect stays on k_fold_n<0>, fbk_query_kernels.hip.h).
//
// One 256-thread block per (group, slot).  The k containers of the slot are dealt to the 4
// wavefronts (wave w takes rows w, w+4, ...); every container is OR-ed (XOR-ed) straight into
// ONE 8 KiB LDS bitmap, in its encoded form:
//   array   one ds_or_b32 per value                          (arrayToBitmap, roaring.go:3756)
//   run     masked ds_or_b32 on the two boundary dwords, whole dwords in between; long runs are
//           filled by all 64 lanes together                  (splatRun, container_stash.go:696-729)
//   bitmap  ds_or_b64 of the 16 words each lane fetched
// so there is no per-container zero / decode / read-back of a scratch bitmap, no register
// accumulator and no cross-wave combine: the only state per lane is the payload prefetch ring
// (3 containers deep), which keeps 12 payloads per block in flight.  The previous kernel
// (decode every container into a per-wave scratch, accumulate in registers, prefetch depth 1,
// 256 VGPRs -> 2 waves per SIMD) measured 296 us on 256 shards x 64 mixed rows (581 MB).
// Unions of sparse rows are dominated by per-container latency, not bytes.
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

// one prefetched payload (<= 8 KiB) into the shared accumulator
template <int OP>
__device__ __forceinline__ void scatter_raw(const Raw& r, uint32_t type, uint32_t len, int lane, uint32_t* acc32) {
  if (type == kTypeBitmap) {
    u64* acc64 = reinterpret_cast<u64*>(acc32);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t w = (j * kWave + lane) * 2;
      if (OP == 2) {
        atomicXor(&acc64[w], r.v[j].x);
        atomicXor(&acc64[w + 1], r.v[j].y);
      } else {
        if (r.v[j].x) atomicOr(&acc64[w], r.v[j].x);
        if (r.v[j].y) atomicOr(&acc64[w + 1], r.v[j].y);
      }
    }
    return;
  }
  if (type == kTypeArray) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t e0 = (j * kWave + lane) * 8u;
      if (e0 < len) {
        const u64 lo = r.v[j].x, hi = r.v[j].y;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t a = (uint32_t)(lo >> (16 * t)) & 0xFFFFu, b = (uint32_t)(hi >> (16 * t)) & 0xFFFFu;
          if (e0 + t < len) lds_acc32<OP>(&acc32[a >> 5], 1u << (a & 31));
          if (e0 + 4 + t < len) lds_acc32<OP>(&acc32[b >> 5], 1u << (b & 31));
        }
      }
    }
    return;
  }
  // runs: 4 intervals per 16-byte chunk
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t i0 = (j * kWave + lane) * 4u;
    if (__ballot(i0 < len) == 0) break;  // wave-uniform: no lane has intervals in this chunk row
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const u64 q = (t < 2) ? r.v[j].x : r.v[j].y;
      const uint32_t iv = (uint32_t)(q >> (32 * (t & 1)));
      const bool on = i0 + t < len;
      const uint32_t s = iv & 0xFFFFu, l = iv >> 16;
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
        for (uint32_t d = ws + 1; d < we; ++d) lds_acc32<OP>(&acc32[d], ~0u);
      u64 longm = __ballot(inner > 4);
      while (longm) {
        const int src = __builtin_ctzll(longm);
        longm &= longm - 1;
        const uint32_t bs = __shfl(ws, src, kWave), be = __shfl(we, src, kWave);
        for (uint32_t d = bs + 1 + lane; d < be; d += kWave) lds_acc32<OP>(&acc32[d], ~0u);
      }
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

template <int OP, bool WRITE>
__global__ void __launch_bounds__(256, 4) k_fold_scatter(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                     const uint32_t* __restrict__ rows, uint64_t n_groups, uint32_t k,
                                                     const Slot* __restrict__ fslots, const uint8_t* __restrict__ farena,
                                                     const uint32_t* __restrict__ frows, uint8_t* __restrict__ arenaO,
                                                     Slot* __restrict__ outSlots, uint32_t* __restrict__ outRuns,
                                                     u64* __restrict__ out_counts) {
  static_assert(OP == 1 || OP == 2 || OP == 3, "scatter fold handles OR, XOR and ANDNOT");
  __shared__ u64 acc[kWords];  // the union / xor of the group's containers at this slot
  __shared__ u64 aux[kWords];  // decode scratch for the filter row and for r0 of a difference
  __shared__ uint32_t s_short;
  __shared__ uint32_t s_cnt[3][4];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint64_t cell = blockIdx.x;  // (group, slot)
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
    s0 = slots[(uint64_t)grow[0] * kSlots + slot];
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
    if (base + lane < k) mine = slots[(uint64_t)grow[base + lane] * kSlots + slot];
    if (OP == 3 && base + lane == 0) mine.tn = 0;  // r0 is not one of the subtrahends
    const uint32_t cnt = min(64u, k - base);
    if (OP != 2 && __ballot(slot_n(mine) == 65536u) != 0) {
      shortcut = true;
      break;
    }
    // this wave's containers: i = wv, wv+4, ...; payload prefetch ring 3 deep
    constexpr int D = 3;
    Raw R[D];
    auto meta = [&](uint32_t i, u64& off, uint32_t& len, uint32_t& tn) {
      // wave-uniform values: pull them into SGPRs so the type dispatch is scalar branching
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mine.off, (int)(i & 63));
      const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(mine.off >> 32), (int)(i & 63));
      off = ((u64)hi << 32) | lo;
      len = __builtin_amdgcn_readlane(mine.len, (int)(i & 63));
      tn = __builtin_amdgcn_readlane(mine.tn, (int)(i & 63));
    };
    auto issue = [&](uint32_t i, Raw& r) {
      if (i < cnt) {
        u64 off;
        uint32_t len, tn;
        meta(i, off, len, tn);
        const uint32_t bytes = payload_bytes(tn >> 24, len);
        if ((tn & 0xFFFFFFu) != 0 && bytes <= 8192u) raw_load(arena + off, bytes, lane, r);
      }
    };
    auto consume = [&](uint32_t i, const Raw& r) {
      if (i < cnt) {
        u64 off;
        uint32_t len, tn;
        meta(i, off, len, tn);
        if ((tn & 0xFFFFFFu) != 0) {
          const uint32_t bytes = payload_bytes(tn >> 24, len);
          if (bytes <= 8192u) scatter_raw<OP>(r, tn >> 24, len, lane, acc32);
          else scatter_big<OP>(arena + off, tn >> 24, len, lane, acc32);
        }
      }
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
  if (WRITE && outRuns) {
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
  if (WRITE) {
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
