// fbk_pair_ring.hip.h — the row-pair count as a PERSISTENT loader / decoder kernel (round 6): k_icount3.
//
// k_icount2 (fbk_pair_kernels.hip.h) is one wave per (pair, slot) item walking launch -> descriptors -> payload -> LDS clear ->
// scatter -> probe -> reduce, each step waited for by that one wave; its own ablations (DESIGN, round 5) put the loads at 97 us and
// the decode at 85 us on config 3's 8192 row pairs — one after the other, 150-168 us in all, whatever the occupancy, the
// instruction mix or the code size.  Here the two halves belong to different waves of a block that stays on its compute unit:
//
//   LOADER wave (wave 0)   walks the plan's resolved item records (k_resolve_items) 32 at a time — the records themselves arrive
//                          by one 1 KiB LDS-DMA — classifies the 32 items lane-parallel (sizes, ring positions by a wave scan,
//                          the short circuits of intersectionCount roaring.go:4478-4486 answered on the spot) and streams every
//                          item's encoded payload into an LDS RING with global_load_lds_dwordx4, allocated by the payload's
//                          ACTUAL bytes (16-byte granules, A then B).  Its vector-memory counter holds nothing but these DMAs, in
//                          order, so "item i has landed" is `s_waitcnt vmcnt(pieces issued after it)` — a counted wait, kLag items
//                          behind the issue point — and the item is PUBLISHED to the block's queue.
//   DECODER waves (1..D)   claim published items (one LDS ticket), decode them out of the ring with their own 8 KiB table and
//                          write ONE count per item; when the last ring byte of an item has been read the entry is RELEASED and
//                          the loader reclaims the space in allocation order.  A decoder never waits for HBM.
//
// Type pairs, as the reference dispatches them (intersectionCount roaring.go:4477-4512):
//   array x array    shorter array scattered into the cleared table, longer one probes      (intersectionCountArrayArray :4514)
//   array x bitmap   the array probes the bitmap WHERE IT LANDED in the ring: no copy        (intersectionCountArrayBitmap :4596)
//   array x run      runs of <= kRunFillMax intervals: boundary masks + interior map, the array probes both  (ArrayRun :4537)
//   bitmap x bitmap  AND + popcount straight out of the ring                                  (BitmapBitmap :4611)
//   run x run / run x bitmap / long run lists: both operands 1 KiB at a time out of the table (RunRun :4573, BitmapRun :4563)
// Items whose payload exceeds kRingItemMax (arrays beyond 4096 values are legal intermediates, roaring.go:5054) and tiny arrays
// against bitmaps (a few gathered dwords beat streaming 8 KiB) keep k_icount2's path: the decoder runs icount_item on them.
//
// Every wait loop is bounded (kSpinLimit polls with s_sleep): a protocol error ends the block with the abort word set instead of
// hanging the device; the host reports it (plan_read).
#pragma once
#include "fbk_pair_kernels.hip.h"

namespace fbk {

constexpr int kRgChunk = 32;            // items per record chunk: 32 x 32 bytes = one 1 KiB DMA
constexpr int kRgQ = 64;                // queue entries (power of two; the loader's per-entry FIFO is one lane each)
constexpr uint32_t kRingItemMax = 16384;  // payload bytes of an item that goes through the ring
constexpr uint32_t kSpinLimit = 1u << 22;

// LDS carve of a block with D decoders and a ring of RING bytes (power of two)
template <int D, int RING>
struct RingLayout {
  static constexpr uint32_t kTab = 0;                           // D x 8 KiB tables, 8 KiB aligned (table_dword_lo ORs the base in)
  static constexpr uint32_t kMini = kTab + D * 8192;            // D x 512 B run maps
  static constexpr uint32_t kRing = kMini + D * 512;
  static constexpr uint32_t kStage = kRing + RING;              // 2 x 1 KiB record chunks (also the slack a ragged last batch reads into)
  static constexpr uint32_t kQueue = kStage + 2048;             // kRgQ x 32 B entries
  static constexpr uint32_t kRel = kQueue + kRgQ * 32;          // kRgQ release words: 0 = held, else the entry's end position + 1
  static constexpr uint32_t kCtl = kRel + kRgQ * 4;             // pub, claim, fin, abort
  static constexpr uint32_t kTotal = kCtl + 16;
  static_assert((RING & (RING - 1)) == 0, "ring size must be a power of two");
  static_assert(kTotal <= 160 * 1024, "LDS carve exceeds a compute unit");
};

// queue entry, 8 dwords: meta (ta | tb << 4 | direct << 8), item, off_a, len_a, off_b, len_b, end position + 1, 0
// (offsets are byte offsets inside the block's LDS carve)

// ---- loader primitives ---------------------------------------------------------------------------------------------------------

// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform; >= 63: nothing to wait for —
// the counter has six bits, so an operation with 63 younger ones issued behind it has retired)
__device__ __forceinline__ void wait_vmcnt_le(uint32_t n) {
#define FBK_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define FBK_W8(k) FBK_W(k) FBK_W(k + 1) FBK_W(k + 2) FBK_W(k + 3) FBK_W(k + 4) FBK_W(k + 5) FBK_W(k + 6) FBK_W(k + 7)
  switch (n) {
    FBK_W8(0) FBK_W8(8) FBK_W8(16) FBK_W8(24) FBK_W8(32) FBK_W8(40) FBK_W8(48)
    FBK_W(56) FBK_W(57) FBK_W(58) FBK_W(59) FBK_W(60) FBK_W(61) FBK_W(62)
    default: break;
  }
#undef FBK_W8
#undef FBK_W
}

// one LDS-DMA piece: lanes [0, nlanes) copy 16 bytes each from gbase + voff to LDS byte address lds_dst + 16 lane.
// (M0 and EXEC are saved and restored inside the statement: the compiler does not model either.)
template <bool NT>
__device__ __forceinline__ void glds_piece_full(uint64_t gbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
template <bool NT>
__device__ __forceinline__ void glds_piece_part(uint64_t gbase, uint32_t voff, uint32_t lds_dst, uint32_t nlanes /* 1..63 */) {
  uint32_t keep;
  u64 keepx;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_bfm_b64 exec, %5, 0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(keepx) : "v"(voff), "s"(gbase), "s"(lds_dst), "s"(nlanes) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_bfm_b64 exec, %5, 0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(keepx) : "v"(voff), "s"(gbase), "s"(lds_dst), "s"(nlanes) : "memory");
}
// bytes16 (a multiple of 16, > 0) from gbase to LDS byte address lds_dst; returns the number of pieces issued
template <bool NT>
__device__ __forceinline__ uint32_t glds_stream(uint64_t gbase, uint32_t bytes16, uint32_t lds_dst, uint32_t voff) {
  const uint32_t nfull = bytes16 >> 10, rem = bytes16 & 1023u;
  for (uint32_t k = 0; k < nfull; ++k) {
    glds_piece_full<NT>(gbase, voff, lds_dst);
    gbase += 1024;
    lds_dst += 1024;
  }
  if (rem) glds_piece_part<NT>(gbase, voff, lds_dst, rem >> 4);
  return nfull + (rem ? 1u : 0u);
}

__device__ __forceinline__ uint32_t rg_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t rg_readlane(uint32_t x, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l); }
__device__ __forceinline__ uint32_t container_bytes(uint32_t type, uint32_t len) {
  return type == kTypeArray ? 2u * len : type == kTypeBitmap ? 8192u : type == kTypeRun ? 4u * len : 0u;
}

// volatile LDS words (the control block and the release words are polled)
__device__ __forceinline__ uint32_t lds_peek(const uint8_t* p) { return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_poke(uint8_t* p, uint32_t v) { __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- decoder primitives: payload batches out of the ring -------------------------------------------------------------------------

// batch starting at unit `base` of the payload at rp (LDS): v[k] = unit base + 64 k + lane.  Units past the payload's end hold
// whatever the ring held (every consumer below masks by index or by bit-field width).
__device__ __forceinline__ void ring_load(const uint8_t* rp, uint32_t base, int lane, uint32_t (&v)[kPairBatch]) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(rp) + base + (uint32_t)lane;
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) v[k] = q[k * kWave];
}

// the dword of bit x of a bitmap at an ARBITRARY LDS byte address (table_dword_lo / _hi OR the base in: 8 KiB aligned tables only)
__device__ __forceinline__ lds_u32* bits_dword_lo(uint32_t base, uint32_t x) {
  uint32_t i, a;
  asm("v_bfe_u32 %0, %1, 5, 11" : "=v"(i) : "v"(x));
  asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(i), "s"(base));
  return (lds_u32*)(uintptr_t)a;
}
__device__ __forceinline__ lds_u32* bits_dword_hi(uint32_t base, uint32_t x) {
  uint32_t i, a;
  asm("v_lshrrev_b32 %0, 21, %1" : "=v"(i) : "v"(x));
  asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(i), "s"(base));
  return (lds_u32*)(uintptr_t)a;
}
// this lane's hits of one batch of array dwords against a bitmap at LDS byte address bb (see array_probe_batch)
__device__ __forceinline__ uint32_t ring_probe_bits_batch(uint32_t bb, uint32_t len, uint32_t base, int lane, const uint32_t (&v)[kPairBatch]) {
  uint32_t hits = 0;
  int32_t left = (int32_t)(len - 2u * base) - 2 * lane;
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) {
    const uint32_t w_lo = (uint32_t)min(max(left, 0), 1), w_hi = (uint32_t)min(max(left - 1, 0), 1);
    hits += __builtin_amdgcn_ubfe(*bits_dword_lo(bb, v[k]), v[k], w_lo) + __builtin_amdgcn_ubfe(*bits_dword_hi(bb, v[k]), v[k] >> 16, w_hi);
    left -= 2 * kWave;
  }
  return hits;
}

// every batch of a sparse payload, the next one's LDS reads issued before the current one is worked on
template <class F>
__device__ __forceinline__ void ring_batches(const uint8_t* rp, uint32_t n_units, int lane, F f) {
  uint32_t v[kPairBatch];
  ring_load(rp, 0, lane, v);
  for (uint32_t base = 0;;) {
    const uint32_t nb = base + kPairBatch * kWave;
    uint32_t nv[kPairBatch];
    if (nb < n_units) ring_load(rp, nb, lane, nv);
    f(base, v);
    if (nb >= n_units) break;
#pragma unroll
    for (int k = 0; k < kPairBatch; ++k) v[k] = nv[k];
    base = nb;
  }
}

__device__ __forceinline__ void ring_xor_all(uint32_t type, const uint8_t* rp, uint32_t len, int lane, uint32_t tb) {
  ring_batches(rp, sparse_units(type, len), lane, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) { sparse_xor_batch(type, tb, len, base, lane, v); });
}
__device__ __forceinline__ void ring_fill_all(const uint8_t* rp, uint32_t len, int lane, uint32_t tb, uint32_t mb) {
  ring_batches(rp, len, lane, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) { run_fill_batch(tb, mb, len, base, lane, v); });
}

// sum over the wave, wave-uniform result (DPP row sums + four lane reads: no LDS round trip)
__device__ __forceinline__ uint32_t wave_sum_uniform(uint32_t v) {
  v = wave_rows_sum(v);
  return rg_readlane(v, 0) + rg_readlane(v, 16) + rg_readlane(v, 32) + rg_readlane(v, 48);
}

// the parity prefix of the 2048-bit interior map of one run operand, one dword per lane, in place (run_table_build's tail)
__device__ __forceinline__ void mini_prefix(uint32_t* mini, int lane) {
  uint32_t x = mini[lane];
  x ^= x << 1;
  x ^= x << 2;
  x ^= x << 4;
  x ^= x << 8;
  x ^= x << 16;
  const u64 odd = __ballot((x >> 31) != 0);
  const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (__popcll(odd & lane_lt) & 1) x = ~x;
  mini[lane] = x;
}

// pair_stream (fbk_pair_kernels.hip.h) with both payloads in the ring: at least one operand sparse, any of them a run
template <class F>
__device__ __forceinline__ void ring_pair_stream(uint32_t ta, const uint8_t* ra, uint32_t lena, uint32_t tb, const uint8_t* rb, uint32_t lenb, int lane,
                                                 u64* table, uint32_t* mini, F f) {
  const bool la = ta != kTypeBitmap, lb = tb != kTypeBitmap;  // wave-uniform
  u64 keep[kWordsPerLane];  // the bitmap operand, or the raw bits of A when both are sparse
  if (!la) lds_read_frag(reinterpret_cast<const u64*>(ra), lane, keep);
  if (!lb) lds_read_frag(reinterpret_cast<const u64*>(rb), lane, keep);
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
    if (fa) ring_fill_all(ra, lena, lane, tbase, mbase);
    else ring_xor_all(ta, ra, lena, lane, tbase);
    wave_lds_sync();
    if (lb) {
      lds_read_frag(table, lane, keep);  // raw bits of A, before B goes on top
      wave_lds_sync();
    }
  }
  if (lb) {
    if (fb) ring_fill_all(rb, lenb, lane, tbase, mbase + 4u * kMiniDwords);
    else ring_xor_all(tb, rb, lenb, lane, tbase);
    wave_lds_sync();
  }
  RunFinish fra, frb;
  run_finish_init(fra, ta, lena, mbase, lane);
  run_finish_init(frb, tb, lenb, mbase + 4u * kMiniDwords, lane);
  wave_lds_sync();
  pair_stream_steps<0>(la, lb, keep, reinterpret_cast<const ulonglong2*>(table), fra, frb, lane, f);
  wave_lds_sync();
}

// one ring item: this lane's part of |A ∩ B|
__device__ __forceinline__ uint32_t ring_icount_item(uint32_t ta, const uint8_t* ra, uint32_t lena, uint32_t tb, const uint8_t* rb, uint32_t lenb, int lane,
                                                     u64* table, uint32_t* mini) {
  asm volatile("" : "+v"(lane));  // (see icount_item: keeps per-lane addresses of every path from being hoisted out of the item loop)
  uint32_t part = 0;
  if (ta == kTypeBitmap && tb == kTypeBitmap) {
    const ulonglong2* qa = reinterpret_cast<const ulonglong2*>(ra);
    const ulonglong2* qb = reinterpret_cast<const ulonglong2*>(rb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const ulonglong2 x = qa[j * kWave + lane], y = qb[j * kWave + lane];
      part += (uint32_t)__popcll(x.x & y.x) + (uint32_t)__popcll(x.y & y.y);
    }
  } else if (ta == kTypeArray && tb == kTypeArray) {
    const bool a_tab = lena <= lenb;  // wave-uniform: the shorter array becomes the table
    const uint8_t* rt = a_tab ? ra : rb;
    const uint8_t* rp = a_tab ? rb : ra;
    const uint32_t lt = a_tab ? lena : lenb, lp = a_tab ? lenb : lena;
    const uint32_t tbase = lds_table_base(table);
    lds_zero(table, lane);
    wave_lds_sync();
    ring_xor_all(kTypeArray, rt, lt, lane, tbase);
    wave_lds_sync();
    ring_batches(rp, (lp + 1u) >> 1, lane, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) { part += array_probe_batch(tbase, lp, base, lane, v); });
    wave_lds_sync();
  } else if ((ta == kTypeArray && tb == kTypeBitmap) || (ta == kTypeBitmap && tb == kTypeArray)) {
    const bool a_arr = ta == kTypeArray;
    const uint8_t* rarr = a_arr ? ra : rb;
    const uint8_t* rbm = a_arr ? rb : ra;
    const uint32_t larr = a_arr ? lena : lenb;
    const uint32_t bb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)rbm);
    ring_batches(rarr, (larr + 1u) >> 1, lane, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) { part += ring_probe_bits_batch(bb, larr, base, lane, v); });
  } else if ((ta == kTypeArray && tb == kTypeRun && lenb <= kRunFillMax) || (ta == kTypeRun && tb == kTypeArray && lena <= kRunFillMax)) {
    const bool a_run = ta == kTypeRun;
    const uint8_t* rr = a_run ? ra : rb;
    const uint8_t* rp = a_run ? rb : ra;
    const uint32_t lr = a_run ? lena : lenb, lp = a_run ? lenb : lena;
    const uint32_t tbase = lds_table_base(table);
    const uint32_t mbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)mini);
    lds_zero(table, lane);
    mini[lane] = 0;
    wave_lds_sync();
    ring_fill_all(rr, lr, lane, tbase, mbase);
    wave_lds_sync();
    mini_prefix(mini, lane);
    wave_lds_sync();
    ring_batches(rp, (lp + 1u) >> 1, lane, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) { part += array_probe_batch<true>(tbase, lp, base, lane, v, mbase); });
    wave_lds_sync();
  } else {
    uint32_t acc = 0;
    ring_pair_stream(ta, ra, lena, tb, rb, lenb, lane, table, mini,
                     [&acc](u64 a0, u64 a1, u64 b0, u64 b1) { acc += (uint32_t)__popcll(a0 & b0) + (uint32_t)__popcll(a1 & b1); });
    part = acc;
  }
  return part;
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------------
//
// grid: any number of blocks (the host launches blocks-per-CU x CUs); block b takes the record chunks b, b + G, b + 2 G, ...
// items: 2 Slots per item (A's descriptor, B's), item = pair * 16 + slot; wave_out[item] = |A_item ∩ B_item|.
// ctl_global[0] is OR-ed with 1 if a block gave up on a wait (never in a correct run).
template <int D, int RING, bool NT>
__global__ void __launch_bounds__(64 * (D + 1)) k_icount3(const Slot* __restrict__ items, const uint8_t* __restrict__ arenaA,
                                                         const uint8_t* __restrict__ arenaB, uint64_t n_items, uint32_t* __restrict__ wave_out,
                                                         uint32_t lag, uint32_t* __restrict__ ctl_global) {
  using L = RingLayout<D, RING>;
  __shared__ __attribute__((aligned(8192))) uint8_t smem[L::kTotal];
  const int lane = threadIdx.x & 63;
  const uint32_t wv = rg_uniform(threadIdx.x >> 6);
  // control block, release words
  for (uint32_t i = threadIdx.x; i < (uint32_t)kRgQ + 4u; i += 64u * (D + 1)) reinterpret_cast<uint32_t*>(smem + L::kRel)[i] = i == (uint32_t)kRgQ + 2u ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  uint8_t* const ctl = smem + L::kCtl;  // +0 pub, +4 claim, +8 fin (0xFFFFFFFF: the loader is still producing), +12 abort

  if (wv != 0) {
    // ------------------------------------------------ decoder ------------------------------------------------
    u64* table = reinterpret_cast<u64*>(smem + L::kTab + (wv - 1u) * 8192u);
    uint32_t* mini = reinterpret_cast<uint32_t*>(smem + L::kMini + (wv - 1u) * 512u);
    for (;;) {
      uint32_t t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(ctl + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      t = rg_uniform(t);
      bool stop = false;
      for (uint32_t spins = 0;; ++spins) {
        const uint32_t pub = rg_uniform(lds_peek(ctl)), fin = rg_uniform(lds_peek(ctl + 8)), ab = rg_uniform(lds_peek(ctl + 12));
        if ((int32_t)(pub - t) > 0) break;
        if (fin != 0xFFFFFFFFu && (int32_t)(t - fin) >= 0) { stop = true; break; }
        if (ab || spins > kSpinLimit) {
          lds_poke(ctl + 12, 1u);
          if (lane == 0) atomicOr(ctl_global, 1u);
          stop = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (stop) break;
      asm volatile("" ::: "memory");  // (the entry and the payload are read after the poll that saw them published)
      const uint32_t e = t & (uint32_t)(kRgQ - 1);
      const uint4 e0 = *reinterpret_cast<const uint4*>(smem + L::kQueue + e * 32u);
      const uint4 e1 = *reinterpret_cast<const uint4*>(smem + L::kQueue + e * 32u + 16u);
      const uint32_t meta = rg_uniform(e0.x), item = rg_uniform(e0.y), offa = rg_uniform(e0.z), lena = rg_uniform(e0.w);
      const uint32_t offb = rg_uniform(e1.x), lenb = rg_uniform(e1.y), endp1 = rg_uniform(e1.z);
      const uint32_t ta = meta & 15u, tb = (meta >> 4) & 15u;
      uint32_t c;
      if (meta & 0x100u) {
        // outside the ring: k_icount2's item, payloads from global memory
        const Slot sa = items[2ull * item], sb = items[2ull * item + 1];
        uint32_t va[kPairBatch], vb[kPairBatch], part = 0, spart = 0;
        item_prefetch(sa, arenaA, sb, arenaB, lane, va, vb);
        icount_item(sa, arenaA, sb, arenaB, lane, table, mini, va, vb, 3u, part, spart);
        c = wave_sum_uniform(part) + spart;
      } else {
        c = wave_sum_uniform(ring_icount_item(ta, smem + offa, lena, tb, smem + offb, lenb, lane, table, mini));
      }
      // every ring byte of the item has been read (LDS operations of a wave retire in order): hand the space back
      asm volatile("" ::: "memory");
      lds_poke(smem + L::kRel + e * 4u, endp1);
      if (lane == 0) wave_out[item] = c;
    }
    return;
  }

  // -------------------------------------------------- loader --------------------------------------------------
  const uint32_t smem_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
  const uint32_t voff = 16u * (uint32_t)lane;
  const uint64_t n_chunks = (n_items + kRgChunk - 1) / kRgChunk;
  uint32_t seq = 0, pub_seq = 0, tail_seq = 0;  // entries created / published / reclaimed
  uint32_t head_pos = 0, tail_pos = 0;          // ring positions (bytes, monotonic mod 2^32; RING divides 2^32)
  uint32_t cum = 0;                             // DMA pieces issued
  uint32_t fifo = 0;                            // lane (seq & 63): `cum` after entry seq's pieces
  bool dead = false;
  lag = lag < 1u ? 1u : (lag > 48u ? 48u : lag);

  auto publish_to = [&](uint32_t upto) {  // entries [pub_seq, upto) have landed
    pub_seq = upto;
    asm volatile("" ::: "memory");  // (the entries' plain LDS stores stay in front: LDS operations of a wave retire in order)
    lds_poke(ctl, upto);
  };
  auto publish_oldest = [&]() {  // the oldest unpublished entry: wait for exactly its pieces
    wait_vmcnt_le(cum - rg_readlane(fifo, pub_seq & 63u));
    publish_to(pub_seq + 1u);
  };
  // released entries, in allocation order: one LDS round trip reclaims up to 64 of them
  auto reclaim = [&]() {
    const uint32_t e = (tail_seq + (uint32_t)lane) & (uint32_t)(kRgQ - 1);
    uint8_t* w = smem + L::kRel + e * 4u;
    const uint32_t r = ((uint32_t)lane < seq - tail_seq) ? lds_peek(w) : 0u;
    const u64 held = ~__ballot(r != 0u);
    const uint32_t k = held ? (uint32_t)__builtin_ctzll(held) : 64u;  // leading released entries
    if (k) {
      if ((uint32_t)lane < k) lds_poke(w, 0u);
      tail_pos = rg_readlane(r, k - 1u) - 1u;
      tail_seq += k;
    }
    return k;
  };
  // until enough(): space comes back only from decoders, and decoders need published entries — so whatever is still unpublished
  // is published while waiting, oldest first, each with its own counted wait (the younger DMAs stay in flight)
  auto wait_reclaim = [&](auto enough) {
    for (uint32_t spins = 0; !enough(); ++spins) {
      if (reclaim()) continue;
      if (pub_seq != seq) {
        publish_oldest();
        continue;
      }
      if (rg_uniform(lds_peek(ctl + 12)) || spins > kSpinLimit) {
        dead = true;
        return;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  };
  auto dma_records = [&](uint64_t chunk, uint32_t buf) {
    const uint64_t first = chunk * kRgChunk;
    const uint32_t n = (uint32_t)min((uint64_t)kRgChunk, n_items - first);
    cum += glds_stream<false>(reinterpret_cast<uint64_t>(items + 2 * first), n * 32u, smem_base + L::kStage + buf * 1024u, voff);
  };

  uint64_t chunk = blockIdx.x;
  uint32_t buf = 0, rec_cum = 0;
  if (chunk < n_chunks) {
    dma_records(chunk, 0);
    rec_cum = cum;
  }
  while (chunk < n_chunks && !dead) {
    const uint64_t next = chunk + gridDim.x;
    uint32_t next_cum = 0;
    if (next < n_chunks) {
      dma_records(next, buf ^ 1u);
      next_cum = cum;
    }
    wait_vmcnt_le(cum - rec_cum);  // this chunk's records have landed
    // ---- 32 items, one per lane ----
    const uint64_t item = chunk * kRgChunk + (uint64_t)lane;
    const bool valid = lane < kRgChunk && item < n_items;
    const uint4* rec = reinterpret_cast<const uint4*>(smem + L::kStage + buf * 1024u + (uint32_t)(lane & (kRgChunk - 1)) * 32u);
    const uint4 ra = rec[0], rb = rec[1];  // Slot {off lo, off hi, len, tn}
    const uint32_t na = ra.w & 0xFFFFFFu, nb = rb.w & 0xFFFFFFu, ta = ra.w >> 24, tb = rb.w >> 24;
    const bool triv = na == 0 || nb == 0 || na == 65536u || nb == 65536u;
    if (valid && triv) wave_out[item] = (na == 0 || nb == 0) ? 0u : (na == 65536u ? nb : na);  // intersectionCount's short circuits, roaring.go:4478-4486
    const bool active = valid && !triv;
    const uint32_t ba = (container_bytes(ta, ra.z) + 15u) & ~15u, bb = (container_bytes(tb, rb.z) + 15u) & ~15u;
    const bool tiny_probe = (ta == kTypeArray && tb == kTypeBitmap && ra.z <= kProbeArray) || (tb == kTypeArray && ta == kTypeBitmap && rb.z <= kProbeArray);
    const bool direct = ba + bb > kRingItemMax || tiny_probe;
    const uint32_t need = (active && !direct) ? ba + bb : 0u;
    const uint32_t pieces = need ? ((ba + 1023u) >> 10) + ((bb + 1023u) >> 10) : 0u;
    const u64 amask = __ballot(active);
    const uint32_t cnt = (uint32_t)__popcll(amask);
    const uint32_t rank = mbcnt64(amask, 0);
    const uint32_t incl = wave_incl_scan(need);
    uint32_t pos = head_pos + incl - need;
    // an item does not straddle the ring's end: the first one that would is moved to the start, everything after it follows
    for (;;) {
      const u64 st = __ballot(need != 0u && (pos & (uint32_t)(RING - 1)) + need > (uint32_t)RING);
      if (!st) break;
      const uint32_t f = (uint32_t)__builtin_ctzll(st);
      const uint32_t bump = (uint32_t)RING - (rg_readlane(pos, f) & (uint32_t)(RING - 1));
      if ((uint32_t)lane >= f) pos += bump;
    }
    const uint32_t endpos = pos + need;
    const uint32_t cum_after = cum + wave_incl_scan(pieces);
    if (cnt) {
      // queue entries for the whole chunk (their slots must have been reclaimed)
      wait_reclaim([&]() { return seq + cnt - tail_seq <= (uint32_t)kRgQ; });
      if (dead) break;
      if (active) {
        const uint32_t phys = L::kRing + (pos & (uint32_t)(RING - 1));
        uint4 q0, q1;
        q0.x = ta | (tb << 4) | (direct ? 0x100u : 0u);
        q0.y = (uint32_t)item;
        q0.z = phys;
        q0.w = ra.z;
        q1.x = phys + ba;
        q1.y = rb.z;
        q1.z = endpos + 1u;
        q1.w = 0;
        uint4* q = reinterpret_cast<uint4*>(smem + L::kQueue + ((seq + rank) & (uint32_t)(kRgQ - 1)) * 32u);
        q[0] = q0;
        q[1] = q1;
      }
      const uint64_t ga = reinterpret_cast<uint64_t>(arenaA) + (((uint64_t)ra.y << 32) | ra.x), gb = reinterpret_cast<uint64_t>(arenaB) + (((uint64_t)rb.y << 32) | rb.x);
      const uint32_t ga_lo = (uint32_t)ga, ga_hi = (uint32_t)(ga >> 32), gb_lo = (uint32_t)gb, gb_hi = (uint32_t)(gb >> 32);
      // ---- the items one after the other: space, DMA, publication kLag items behind ----
      for (u64 m = amask; m && !dead; m &= m - 1) {
        const uint32_t j = (uint32_t)__builtin_ctzll(m);
        const uint32_t need_j = rg_readlane(need, j);
        if (need_j) {
          const uint32_t end_j = rg_readlane(endpos, j);
          wait_reclaim([&]() { return end_j - tail_pos <= (uint32_t)RING; });
          if (dead) break;
          const uint32_t ba_j = rg_readlane(ba, j), dst = smem_base + L::kRing + (rg_readlane(pos, j) & (uint32_t)(RING - 1));
          const uint64_t pa = ((uint64_t)rg_readlane(ga_hi, j) << 32) | rg_readlane(ga_lo, j), pb = ((uint64_t)rg_readlane(gb_hi, j) << 32) | rg_readlane(gb_lo, j);
          glds_stream<NT>(pa, ba_j, dst, voff);
          glds_stream<NT>(pb, need_j - ba_j, dst + ba_j, voff);
        }
        cum = rg_readlane(cum_after, j);
        fifo = (uint32_t)lane == (seq & 63u) ? cum : fifo;
        ++seq;
        if (seq - pub_seq > lag) publish_oldest();
      }
      head_pos = rg_readlane(endpos, 63u - (uint32_t)__builtin_clzll(amask));
    }
    chunk = next;
    buf ^= 1u;
    rec_cum = next_cum;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (dead) {
    lds_poke(ctl + 12, 1u);
    if (lane == 0) atomicOr(ctl_global, 1u);
  }
  publish_to(seq);
  lds_poke(ctl + 8, seq);  // fin: decoders leave once their ticket is past it
}

}  // namespace fbk
