// fbk_pair_ring.hip.h — the row-pair count as a PERSISTENT loader / decoder kernel (round 6): k_icount3.
//
// k_icount2 (fbk_pair_kernels.hip.h) is one wave per (pair, slot) item walking launch -> descriptors -> payload -> LDS clear ->
// scatter -> probe -> reduce, each step waited for by that one wave; its own ablations (DESIGN, round 5) put the loads at 97 us and
// the decode at 85 us on config 3's 8192 row pairs — one after the other, 150-168 us in all, whatever the occupancy, the
// instruction mix or the code size.  Here the two halves belong to different waves of a block that stays on its compute unit:
//
//   PLANNER wave (wave 0)  walks the plan's resolved item records (k_resolve_items) 32 at a time — the records themselves arrive
//                          by one 1 KiB LDS-DMA — and classifies the 32 items LANE-PARALLEL: sizes, the short circuits of
//                          intersectionCount (roaring.go:4478-4486, answered on the spot), and a place in the block's LDS RING for
//                          every payload by a wave scan, allocated by the payload's ACTUAL bytes (16-byte granules, A then B).  It
//                          writes one queue entry per item and reclaims the ring space of released entries in allocation order.
//   DECODER waves (1..D)   claim entries (one LDS ticket each).  A decoder first starts the global -> LDS DMA of the entry it has
//                          just claimed (global_load_lds_dwordx4 into the item's place in the ring: no registers, nothing waits),
//                          THEN decodes the item it claimed one round earlier, whose payload landed meanwhile: its vector-memory
//                          counter holds its own DMAs in issue order, so "my item has landed" is the counted wait `s_waitcnt
//                          vmcnt(pieces issued after it)`.  Decoding works out of the ring with the wave's own 8 KiB table and
//                          ends with ONE count per item; the entry is then RELEASED.  With D decoders a compute unit has D
//                          payloads in flight while D others are decoded, and no wave waits for HBM in front of its own decode.
// (The first version of this file had ONE loader wave issue every DMA and publish landed items: correct, and 3.5 x slower than
// k_icount2 — ~1800 cycles of scalar bookkeeping and DMA issue per item in one instruction stream, decoders idle 70 % of the time,
// profiles/r06_ring_v1_debug.txt.)
//
// Type pairs, as the reference dispatches them (intersectionCount roaring.go:4477-4512):
//   array x array    shorter array scattered into the cleared table, longer one probes      (intersectionCountArrayArray :4514)
//   array x bitmap   the array probes the bitmap WHERE IT LANDED in the ring: no copy        (intersectionCountArrayBitmap :4596)
//   array x run      runs of <= kRunFillMax intervals: boundary masks + interior map, the array probes both  (ArrayRun :4537)
//   bitmap x bitmap  AND + popcount straight out of the ring                                  (BitmapBitmap :4611)
//   run x run / run x bitmap / long run lists: both operands 1 KiB at a time out of the table (RunRun :4573, BitmapRun :4563)
// Batches that may hold an oversized container (arrays beyond 4096 values are legal intermediates, roaring.go:5054) stay with
// k_icount2: the host knows (fbk_batch::ring_regular).
//
// Every wait loop is bounded (kSpinLimit polls with s_sleep): a protocol error ends the block with the abort word set instead of
// hanging the device; the host reports it (plan_read).
#pragma once
#include "fbk_pair_kernels.hip.h"

namespace fbk {

constexpr int kRgChunk = 32;            // items per record chunk: 32 x 32 bytes = one 1 KiB DMA
constexpr int kRgQ = 64;                // queue entries (power of two; one reclaim pass looks at 64 release words, one per lane)
constexpr int kRgEntry = 64;            // bytes per queue entry
constexpr uint32_t kRingItemMax = 16384;  // payload bytes of an item that goes through the ring
constexpr uint32_t kSpinLimit = 1u << 22;

// LDS carve of a block with D decoders and a ring of RING bytes (power of two)
template <int D, int RING>
struct RingLayout {
  static constexpr uint32_t kTab = 0;                           // D x 8 KiB tables, 8 KiB aligned (table_dword_lo ORs the base in)
  static constexpr uint32_t kMini = kTab + D * 8192;            // D x 512 B run maps
  static constexpr uint32_t kRing = kMini + D * 512;
  static constexpr uint32_t kStage = kRing + RING;              // 2 x 1 KiB record chunks (also the slack a ragged last batch reads into)
  static constexpr uint32_t kQueue = kStage + 2048;             // kRgQ x 64 B entries
  static constexpr uint32_t kRel = kQueue + kRgQ * kRgEntry;    // kRgQ release words: 0 = held, 1 = done (the planner reuses queue slots in order)
  static constexpr uint32_t kCtl = kRel + kRgQ * 4;             // pub, claim, fin, abort, -, heartbeat, page mask (8 bytes)
  static constexpr uint32_t kTotal = kCtl + 32;
  static_assert(RING / 1024 <= 64, "one bit per 1 KiB page of the ring");
  static_assert((RING & (RING - 1)) == 0, "ring size must be a power of two");
  static_assert(kTotal <= 160 * 1024, "LDS carve exceeds a compute unit");
};

// queue entry, 12 dwords: meta (ta | tb << 4), item, off_a, len_a | bytes_a (16-byte granules rounded up), len_b, first page,
// need (payload bytes: A then B) | A's global address, B's global address
// (off_a: byte offset inside the block's LDS carve, at a 1 KiB page boundary of the ring; B follows A)
//
// RING SPACE is handed out in 1 KiB pages, one bit each in a 64-bit word of the control block.  The planner gives every item the
// pages FOLLOWING the previous item's (so that consecutive tickets land side by side); a decoder owns them from the moment its
// atomic OR finds them all clear to its release, whatever other waves do meanwhile.  (Until this version the ring was reclaimed IN
// ORDER behind a tail position: one slow item — a run x run decode takes twice the average — held back every payload behind it,
// 40-70 % of the DMAs could not start when their entry was taken and the waves waited for HBM after all:
// profiles/r06_ring_v3_inorder_debug.txt.)

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

// LDS-DMA pieces: lanes [0, nlanes) copy 16 bytes each from gbase + voff (+ 1024 k) to LDS byte address lds_dst + 16 lane (+ 1024 k)
// — the instruction's immediate offset moves BOTH addresses, so up to four pieces share one M0.  (M0 and EXEC are saved and
// restored inside the statement: the compiler does not model either.)
#define FBK_GLDS_ASM(NTS)                                                                                                                  \
  switch (nfull) {                                                                                                                         \
    case 4:                                                                                                                                \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" NTS                                 \
                   "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" NTS "\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" NTS                \
                   "\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" NTS "\n\ts_mov_b32 m0, %0"                                              \
                   : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");                                                        \
      break;                                                                                                                               \
    case 3:                                                                                                                                \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" NTS                                 \
                   "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" NTS "\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" NTS                \
                   "\n\ts_mov_b32 m0, %0"                                                                                                   \
                   : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");                                                        \
      break;                                                                                                                               \
    case 2:                                                                                                                                \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" NTS                                 \
                   "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" NTS "\n\ts_mov_b32 m0, %0"                                              \
                   : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");                                                        \
      break;                                                                                                                               \
    case 1:                                                                                                                                \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" NTS "\n\ts_mov_b32 m0, %0"         \
                   : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");                                                        \
      break;                                                                                                                               \
    default: break;                                                                                                                        \
  }
// nfull (0..4) full pieces
template <bool NT>
__device__ __forceinline__ void glds_full(uint64_t gbase, uint32_t voff, uint32_t lds_dst, uint32_t nfull) {
  uint32_t keep;
  if (NT) {
    FBK_GLDS_ASM(" nt")
  } else {
    FBK_GLDS_ASM("")
  }
}
#undef FBK_GLDS_ASM
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
  const uint32_t pieces = (bytes16 + 1023u) >> 10;
  while (bytes16 >= 4096u) {
    glds_full<NT>(gbase, voff, lds_dst, 4u);
    gbase += 4096;
    lds_dst += 4096;
    bytes16 -= 4096u;
  }
  const uint32_t nfull = bytes16 >> 10, rem = bytes16 & 1023u;
  glds_full<NT>(gbase, voff, lds_dst, nfull);
  if (rem) glds_piece_part<NT>(gbase + 1024u * nfull, voff, lds_dst + 1024u * nfull, rem >> 4);
  return pieces;
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

// ---- a sparse payload WHOLE in registers: up to four batches (an array of 4096 values, 2048 runs) ---------------------------------
// A decoder's time was not instructions but dependent LDS round trips (rocprofv3: vector ALU 39 % busy, LDS 32 %, ~6000 cycles
// per item in a wave whose instructions need ~2500): batch by batch, every load waited for before the next step.  LDS operations
// of one wave execute in order, so everything an item needs can be REQUESTED up front — both payloads, then the table clear —
// and the three phases (scatter, probe, reduce) each wait once.
constexpr uint32_t kRgWholeUnits = 4u * kPairBatch * kWave;  // 2048 dwords
struct Whole {
  uint32_t b0[kPairBatch], b1[kPairBatch], b2[kPairBatch], b3[kPairBatch];
};
__device__ __forceinline__ void whole_load(const uint8_t* rp, uint32_t n_units, int lane, Whole& w) {
  constexpr uint32_t B = kPairBatch * kWave;
  ring_load(rp, 0, lane, w.b0);
  if (n_units > B) ring_load(rp, B, lane, w.b1);
  if (n_units > 2 * B) ring_load(rp, 2 * B, lane, w.b2);
  if (n_units > 3 * B) ring_load(rp, 3 * B, lane, w.b3);
}
// a register of the LAST batch requested
__device__ __forceinline__ uint32_t whole_last(const Whole& w, uint32_t n_units) {
  constexpr uint32_t B = kPairBatch * kWave;
  return n_units > 3 * B ? w.b3[kPairBatch - 1] : n_units > 2 * B ? w.b2[kPairBatch - 1] : n_units > B ? w.b1[kPairBatch - 1] : w.b0[kPairBatch - 1];
}
// f(base, batch) for the batches that exist
template <class F>
__device__ __forceinline__ void whole_each(const Whole& w, uint32_t n_units, F f) {
  constexpr uint32_t B = kPairBatch * kWave;
  f(0u, w.b0);
  if (n_units > B) f(B, w.b1);
  if (n_units > 2 * B) f(2 * B, w.b2);
  if (n_units > 3 * B) f(3 * B, w.b3);
}

// XOR one batch of array dwords into the table, no branch: a value past the array's end toggles nothing (its mask is 0 << bit;
// its dword — whatever the ring held — is still an address inside the table)
__device__ __forceinline__ void array_xor_batch_w(uint32_t tb, uint32_t len, uint32_t base, int lane, const uint32_t (&v)[kPairBatch]) {
  int32_t left = (int32_t)(len - 2u * base) - 2 * lane;
#pragma unroll
  for (int k = 0; k < kPairBatch; ++k) {
    const uint32_t w_lo = (uint32_t)min(max(left, 0), 1), w_hi = (uint32_t)min(max(left - 1, 0), 1);
    (void)__hip_atomic_fetch_xor(table_dword_lo(tb, v[k]), w_lo << (v[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)__hip_atomic_fetch_xor(table_dword_hi(tb, v[k]), w_hi << ((v[k] >> 16) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    left -= 2 * kWave;
  }
}

// ---- row-exact scatter and probe -----------------------------------------------------------------------------------------------
// A wave's time per item is its instruction count (~5.5 cycles per instruction in a stream without waits) plus the LDS round
// trips it cannot overlap.  The batch functions above execute all eight dword rows of a batch whatever the array holds and carry
// every value's bit-field width along (v_cmp + v_cndmask per value); but of an array's rows only the LAST can be ragged.  Here a
// batch executes exactly its rows: the full ones (128 values each, no width arithmetic) through a fall-through switch — one
// branch tree per batch — and the one ragged row, if it lies in the batch, with its widths.
template <class F>
__device__ __forceinline__ void rows_desc(uint32_t n, const uint32_t (&v)[kPairBatch], F f) {  // rows n - 1 .. 0
  switch (n) {
    default: f(7, v[7]); [[fallthrough]];
    case 7: f(6, v[6]); [[fallthrough]];
    case 6: f(5, v[5]); [[fallthrough]];
    case 5: f(4, v[4]); [[fallthrough]];
    case 4: f(3, v[3]); [[fallthrough]];
    case 3: f(2, v[2]); [[fallthrough]];
    case 2: f(1, v[1]); [[fallthrough]];
    case 1: f(0, v[0]); [[fallthrough]];
    case 0: break;
  }
}
template <class F>
__device__ __forceinline__ void row_at(uint32_t k, const uint32_t (&v)[kPairBatch], F f) {  // row k (0..7)
  switch (k) {
    case 0: f(0, v[0]); break;
    case 1: f(1, v[1]); break;
    case 2: f(2, v[2]); break;
    case 3: f(3, v[3]); break;
    case 4: f(4, v[4]); break;
    case 5: f(5, v[5]); break;
    case 6: f(6, v[6]); break;
    default: f(7, v[7]); break;
  }
}
// the rows of an array of len values that lie in the batch starting at dword `base`: full rows [0, nfull), then at most one ragged
// row at index nfull holding `left` (1..127) values
struct BatchRows {
  uint32_t nfull, left;
};
__device__ __forceinline__ BatchRows batch_rows(uint32_t len, uint32_t base) {
  const uint32_t first = 2u * base;                      // the batch's first value
  const uint32_t have = len > first ? len - first : 0u;  // values from there on
  BatchRows r;
  r.nfull = min(have >> 7, (uint32_t)kPairBatch);
  r.left = r.nfull < (uint32_t)kPairBatch ? have - (r.nfull << 7) : 0u;  // (< 128: nfull is the floor)
  return r;
}
// one dword row (two values per lane) into the table at tb
__device__ __forceinline__ void xor_row_full(uint32_t tb, uint32_t x) {
  (void)__hip_atomic_fetch_xor(table_dword_lo(tb, x), 1u << (x & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  (void)__hip_atomic_fetch_xor(table_dword_hi(tb, x), 1u << ((x >> 16) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void xor_row_part(uint32_t tb, uint32_t x, int32_t mine /* values of the row from this lane's low one on */) {
  const uint32_t w_lo = (uint32_t)min(max(mine, 0), 1), w_hi = (uint32_t)min(max(mine - 1, 0), 1);
  (void)__hip_atomic_fetch_xor(table_dword_lo(tb, x), w_lo << (x & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  (void)__hip_atomic_fetch_xor(table_dword_hi(tb, x), w_hi << ((x >> 16) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the whole array (in registers) into the cleared table
__device__ __forceinline__ void whole_xor(const Whole& w, uint32_t len, int lane, uint32_t tb) {
  whole_each(w, (len + 1u) >> 1, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) {
    const BatchRows r = batch_rows(len, base);
    rows_desc(r.nfull, v, [&](int, uint32_t x) { xor_row_full(tb, x); });
    if (r.left) row_at(r.nfull, v, [&](int, uint32_t x) { xor_row_part(tb, x, (int32_t)r.left - 2 * lane); });
  });
}
// this lane's hits of the whole array (in registers) against the bitmap at LDS byte address tb: ALIGNED = an 8 KiB aligned table
// (the base is OR-ed in), else any address (added); MAP: a run container's boundary masks at tb, its full dwords in the map at mb.
// Every table read of a batch is requested before the first one is looked at.
template <bool ALIGNED, bool MAP>
__device__ __forceinline__ uint32_t whole_probe(const Whole& w, uint32_t len, int lane, uint32_t tb, uint32_t mb = 0) {
  uint32_t hits = 0;
  whole_each(w, (len + 1u) >> 1, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) {
    const BatchRows r = batch_rows(len, base);
    const uint32_t nrows = r.nfull + (r.left ? 1u : 0u);
    uint32_t t_lo[kPairBatch], t_hi[kPairBatch], m_lo[kPairBatch], m_hi[kPairBatch];
    rows_desc(nrows, v, [&](int k, uint32_t x) {
      t_lo[k] = ALIGNED ? *table_dword_lo(tb, x) : *bits_dword_lo(tb, x);
      t_hi[k] = ALIGNED ? *table_dword_hi(tb, x) : *bits_dword_hi(tb, x);
      if (MAP) {
        m_lo[k] = *(const lds_u32*)(uintptr_t)(mb + (__builtin_amdgcn_ubfe(x, 10u, 6u) << 2));
        m_hi[k] = *(const lds_u32*)(uintptr_t)(mb + ((x >> 26) << 2));
      }
    });
    rows_desc(r.nfull, v, [&](int k, uint32_t x) {
      uint32_t b0 = __builtin_amdgcn_ubfe(t_lo[k], x, 1u), b1 = __builtin_amdgcn_ubfe(t_hi[k], x >> 16, 1u);
      if (MAP) {
        b0 |= __builtin_amdgcn_ubfe(m_lo[k], x >> 5, 1u);
        b1 |= __builtin_amdgcn_ubfe(m_hi[k], x >> 21, 1u);
      }
      hits += b0 + b1;
    });
    if (r.left)
      row_at(r.nfull, v, [&](int k, uint32_t x) {
        const int32_t mine = (int32_t)r.left - 2 * lane;
        const uint32_t w_lo = (uint32_t)min(max(mine, 0), 1), w_hi = (uint32_t)min(max(mine - 1, 0), 1);
        uint32_t b0 = __builtin_amdgcn_ubfe(t_lo[k], x, w_lo), b1 = __builtin_amdgcn_ubfe(t_hi[k], x >> 16, w_hi);
        if (MAP) {
          b0 |= __builtin_amdgcn_ubfe(m_lo[k], x >> 5, w_lo);
          b1 |= __builtin_amdgcn_ubfe(m_hi[k], x >> 21, w_hi);
        }
        hits += b0 + b1;
      });
  });
  return hits;
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
// release(): called once, as soon as no ring byte of the item will be read again — for the table + probe pairs that is right after
// both payloads have reached registers, BEFORE the decode (the ring then holds what is in flight, not what is being worked on)
// next(): called once — the point where the wave should take its following entry and start that entry's DMA: right after the
// release where the release is early (the space just freed may be what the next payload needs), first thing otherwise.
template <class R, class N>
__device__ __forceinline__ uint32_t ring_icount_item(uint32_t ta, const uint8_t* ra, uint32_t lena, uint32_t tb, const uint8_t* rb, uint32_t lenb, int lane,
                                                     u64* table, uint32_t* mini, R release, N next, uint32_t* ph = nullptr) {
  // the payload's LAST requested dword has arrived, so every earlier one has (LDS operations of a wave return in order): the
  // compiler turns the register dependence into a counted lgkmcnt wait that leaves younger operations (the table clear) in flight
  auto landed = [&](uint32_t last) {
    asm volatile("" ::"v"(last) : "memory");
    release();
  };
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
    landed(part);
    next();
  } else if (ta == kTypeArray && tb == kTypeArray) {
    const bool a_tab = lena <= lenb;  // wave-uniform: the shorter array becomes the table
    const uint8_t* rt = a_tab ? ra : rb;
    const uint8_t* rp = a_tab ? rb : ra;
    const uint32_t lt = a_tab ? lena : lenb, lp = a_tab ? lenb : lena;
    const uint32_t ut = (lt + 1u) >> 1, up = (lp + 1u) >> 1;
    const uint32_t tbase = lds_table_base(table);
    Whole wt, wp;
    const u64 p0 = ph ? __builtin_readcyclecounter() : 0;  // (ring_flags bit 1: cycle stamps per phase of the array x array items; every stamp drains the LDS queue)
    whole_load(rt, ut, lane, wt);
    whole_load(rp, up, lane, wp);
    lds_zero(table, lane);
    landed(whole_last(wp, up));
    const u64 p1 = ph ? __builtin_readcyclecounter() : 0;
    next();
    const u64 p2 = ph ? __builtin_readcyclecounter() : 0;
    wave_lds_sync();
    whole_xor(wt, lt, lane, tbase);
    wave_lds_sync();
    const u64 p3 = ph ? __builtin_readcyclecounter() : 0;
    part += whole_probe<true, false>(wp, lp, lane, tbase);
    wave_lds_sync();
    if (ph) {
      asm volatile("" ::"v"(part));
      const u64 p4 = __builtin_readcyclecounter();
      ph[0] += (uint32_t)(p1 - p0), ph[1] += (uint32_t)(p2 - p1), ph[2] += (uint32_t)(p3 - p2), ph[3] += (uint32_t)(p4 - p3), ph[4] += ut, ph[5] += up;
    }
  } else if ((ta == kTypeArray && tb == kTypeBitmap) || (ta == kTypeBitmap && tb == kTypeArray)) {
    const bool a_arr = ta == kTypeArray;
    const uint8_t* rarr = a_arr ? ra : rb;
    const uint8_t* rbm = a_arr ? rb : ra;
    const uint32_t larr = a_arr ? lena : lenb, ua = (larr + 1u) >> 1;
    const uint32_t bb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)rbm);
    next();
    Whole wa;
    whole_load(rarr, ua, lane, wa);
    part += whole_probe<false, false>(wa, larr, lane, bb);
    landed(part);  // (the bitmap is probed where it lies)
  } else if ((ta == kTypeArray && tb == kTypeRun && lenb <= kRunFillMax) || (ta == kTypeRun && tb == kTypeArray && lena <= kRunFillMax)) {
    const bool a_run = ta == kTypeRun;
    const uint8_t* rr = a_run ? ra : rb;
    const uint8_t* rp = a_run ? rb : ra;
    const uint32_t lr = a_run ? lena : lenb, lp = a_run ? lenb : lena, up = (lp + 1u) >> 1;
    const uint32_t tbase = lds_table_base(table);
    const uint32_t mbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)mini);
    Whole wr, wp;
    whole_load(rr, lr, lane, wr);  // (<= kRunFillMax intervals: two batches)
    whole_load(rp, up, lane, wp);
    lds_zero(table, lane);
    mini[lane] = 0;
    landed(whole_last(wp, up));
    next();
    wave_lds_sync();
    whole_each(wr, lr, [&](uint32_t base, const uint32_t (&v)[kPairBatch]) { run_fill_batch(tbase, mbase, lr, base, lane, v); });
    wave_lds_sync();
    mini_prefix(mini, lane);
    wave_lds_sync();
    part += whole_probe<true, true>(wp, lp, lane, tbase, mbase);
    wave_lds_sync();
  } else {
    next();
    uint32_t acc = 0;
    ring_pair_stream(ta, ra, lena, tb, rb, lenb, lane, table, mini,
                     [&acc](u64 a0, u64 a1, u64 b0, u64 b1) { acc += (uint32_t)__popcll(a0 & b0) + (uint32_t)__popcll(a1 & b1); });
    part = acc;
    landed(part);
  }
  return part;
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------------
//
// grid: any number of blocks (the host launches blocks-per-CU x CUs); block b starts with record chunk b and draws the following
// ones from the counter ctl_global[1] (zero at launch: k_sum_wave_counts, which follows on the stream, puts it back).
// items: 2 Slots per item (A's descriptor, B's), item = pair * 16 + slot; wave_out[item] = |A_item ∩ B_item|.
// ctl_global[0] is OR-ed with 1 if a block gave up on a wait (never in a correct run).  dbg (option ring_debug, else null): 32 words
// per block of cycle counts.

struct RingEntry {  // a queue entry in scalar registers
  uint32_t meta, item, off_a, len_a, bytes_a, len_b, page0, need;
  uint64_t ga, gb;
};

template <int D, int RING, bool NT>
__global__ void __launch_bounds__(64 * (D + 1)) k_icount3(const Slot* __restrict__ items, const uint8_t* __restrict__ arenaA,
                                                         const uint8_t* __restrict__ arenaB, uint64_t n_items, uint32_t* __restrict__ wave_out,
                                                         uint32_t* __restrict__ ctl_global, uint32_t* __restrict__ dbg, uint32_t flags) {
  using L = RingLayout<D, RING>;
  __shared__ __attribute__((aligned(8192))) uint8_t smem[L::kTotal];
  const int lane = threadIdx.x & 63;
  const uint32_t wv = rg_uniform(threadIdx.x >> 6);
  // release words and control block: pub, claim, fin (0xFFFFFFFF: the planner is still producing), abort, tail position
  for (uint32_t i = threadIdx.x; i < (uint32_t)kRgQ + 8u; i += 64u * (D + 1)) reinterpret_cast<uint32_t*>(smem + L::kRel)[i] = i == (uint32_t)kRgQ + 2u ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  uint8_t* const ctl = smem + L::kCtl;  // +0 pub (entries written), +4 claim, +8 fin, +12 abort, +16 tail position
  const uint32_t smem_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
  const uint32_t voff = 16u * (uint32_t)lane;
  auto tick = [&]() { return dbg ? __builtin_readcyclecounter() : (u64)0; };
  auto give_up = [&](uint32_t where, uint32_t x0, uint32_t x1, uint32_t x2) {  // (the first block to give up leaves where and what it saw in ctl_global[2..7])
    const uint32_t already = rg_uniform(lds_peek(ctl + 12));
    lds_poke(ctl + 12, 1u);
    if (lane == 0 && !already && atomicOr(ctl_global, 1u) == 0u) {
      ctl_global[2] = where | (blockIdx.x << 8) | (wv << 24);
      ctl_global[3] = x0, ctl_global[4] = x1, ctl_global[5] = x2;
      ctl_global[6] = lds_peek(ctl), ctl_global[7] = lds_peek(ctl + 20);  // (entries written, the planner's last heartbeat)
    }
  };

  if (wv != 0) {
    // ------------------------------------------------ decoder ------------------------------------------------
    // One round: the payload of the entry in hand has landed -> its sparse operands go to registers and its ring space goes back
    // -> the NEXT entry is taken and its DMA started (into space this very round may just have freed) -> the entry in hand is
    // decoded while that DMA flies.  A wave holds one place in the ring for all but a few hundred cycles of a round.
    u64* table = reinterpret_cast<u64*>(smem + L::kTab + (wv - 1u) * 8192u);
    uint32_t* mini = reinterpret_cast<uint32_t*>(smem + L::kMini + (wv - 1u) * 512u);
    uint32_t d_poll = 0, d_land = 0, d_dec = 0, d_n = 0, d_defer = 0;  // (ring_debug: cycles waiting for an entry / for the payload / in the decode; items; deferred issues)
    uint32_t d_cls[5] = {0, 0, 0, 0, 0}, d_cln[5] = {0, 0, 0, 0, 0};  // (ring_debug: decode cycles / items per class: a x a, a x b, a x run(short), general, b x b)
    uint32_t d_ph[6] = {0, 0, 0, 0, 0, 0};
    RingEntry cur, nxt;
    uint32_t cur_slot = 0, nxt_slot = 0;
    bool cur_issued = false, nxt_issued = false, got = false, bad = false;
    auto issue = [&](const RingEntry& e) {  // the item's payload, A then B, into its place in the ring
      const uint32_t dst = smem_base + e.off_a;
      (void)glds_stream<NT>(e.ga, e.bytes_a, dst, voff);
      (void)glds_stream<NT>(e.gb, e.need - e.bytes_a, dst + e.bytes_a, voff);
    };
    u64* const pages = reinterpret_cast<u64*>(ctl + 24);
    auto pages_of = [&](const RingEntry& e) { return (~0ull >> (64u - ((e.need + 1023u) >> 10))) << e.page0; };  // (1..16 pages)
    auto acquire = [&](const RingEntry& e) {  // all of the item's pages, or none
      const u64 my = pages_of(e);
      u64 old = 0;
      if (lane == 0) old = __hip_atomic_fetch_or(pages, my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = ((u64)rg_uniform((uint32_t)(old >> 32)) << 32) | rg_uniform((uint32_t)old);
      if ((old & my) == 0ull) return true;
      if (lane == 0) (void)__hip_atomic_fetch_and(pages, ~(my & ~old), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (give back the ones that were clear)
      return false;
    };
    auto claim = [&]() {  // a ticket (lane 0's register; read with rg_uniform when it is needed: the LDS round trip hides behind a decode)
      uint32_t t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(ctl + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return t;
    };
    uint32_t ticket = claim();
    uint32_t t_nxt = 0;
    // the entry of ticket t_nxt, if the planner has written it: one LDS round trip (the counters, then — LDS operations of a wave execute
    // in order — the entry's words), then its DMA if its place in the ring is free
    auto take = [&]() {
      const uint32_t pub0 = lds_peek(ctl);
      asm volatile("" ::: "memory");
      const uint4* qe = reinterpret_cast<const uint4*>(smem + L::kQueue + (t_nxt & (uint32_t)(kRgQ - 1)) * (uint32_t)kRgEntry);
      const uint4 e0 = qe[0], e1 = qe[1], e2 = qe[2];
      got = (int32_t)(rg_uniform(pub0) - t_nxt) > 0;
      if (!got) return;
      nxt_slot = t_nxt & (uint32_t)(kRgQ - 1);
      nxt.meta = rg_uniform(e0.x), nxt.item = rg_uniform(e0.y), nxt.off_a = rg_uniform(e0.z), nxt.len_a = rg_uniform(e0.w);
      nxt.bytes_a = rg_uniform(e1.x), nxt.len_b = rg_uniform(e1.y), nxt.page0 = rg_uniform(e1.z), nxt.need = rg_uniform(e1.w);
      nxt.ga = ((uint64_t)rg_uniform(e2.y) << 32) | rg_uniform(e2.x);
      nxt.gb = ((uint64_t)rg_uniform(e2.w) << 32) | rg_uniform(e2.z);
      nxt_issued = acquire(nxt);
      if (nxt_issued) issue(nxt);
    };
    // the same when the wave has nothing in hand: it may wait for the planner (or for the end).  (A wave that still holds an undecoded
    // entry never does: the planner may itself be waiting for queue slots, which only released entries give back — the first version
    // of this loop did, and hung as soon as the decoders caught up.)
    auto take_blocking = [&]() {
      for (uint32_t spins = 0;; ++spins) {
        take();
        if (got) return;
        const uint32_t fin = rg_uniform(lds_peek(ctl + 8)), ab = rg_uniform(lds_peek(ctl + 12));
        if (fin != 0xFFFFFFFFu && (int32_t)(t_nxt - fin) >= 0) return;
        if (ab || spins > kSpinLimit) {
          bad = true;
          give_up(1u, t_nxt, fin, rg_uniform(lds_peek(ctl + 4)));
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    };
    // the first entry
    {
      const u64 c0 = tick();
      t_nxt = rg_uniform(ticket);
      ticket = claim();
      take_blocking();
      d_poll += (uint32_t)(tick() - c0);
    }
    while (got && !bad) {
      cur = nxt;
      cur_slot = nxt_slot;
      cur_issued = nxt_issued;
      got = false;
      t_nxt = rg_uniform(ticket);
      ticket = claim();  // (the ticket after the next one)
      const u64 c0 = tick();
      if (!cur_issued) {
        // its pages were not all free when the entry was taken: their owners are other waves in the middle of a round — this wave
        // holds nothing, so it can wait
        ++d_defer;
        for (uint32_t spins = 0; !acquire(cur); ++spins) {
          if (rg_uniform(lds_peek(ctl + 12)) || spins > kSpinLimit) {
            bad = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (bad) {
          give_up(2u, cur.page0 | (cur.need << 8), lds_peek(ctl + 24), lds_peek(ctl + 28));
          break;
        }
        issue(cur);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the payload has landed (this wave's only vector-memory operations in flight)
      const u64 c1 = tick();
      const uint32_t ta = cur.meta & 15u, tb = (cur.meta >> 4) & 15u;
      const uint8_t* ra = smem + cur.off_a;
      uint8_t* const relw = smem + L::kRel + cur_slot * 4u;
      const u64 my_pages = pages_of(cur);
      bool taken = false;
      const uint32_t c = wave_sum_uniform(ring_icount_item(
          ta, ra, cur.len_a, tb, ra + cur.bytes_a, cur.len_b, lane, table, mini,
          [&]() {  // release: every ring byte of the item has been read (LDS operations of a wave retire in order)
            if (flags & 1u) return;  // (experiment: release after the decode)
            if (lane == 0) (void)__hip_atomic_fetch_and(pages, ~my_pages, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            lds_poke(relw, 1u);
          },
          [&]() {                            // next: take the following entry and start its DMA
            take();
            taken = true;
          },
          (dbg && (flags & 2u)) ? d_ph : nullptr));
      if (flags & 1u) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) (void)__hip_atomic_fetch_and(pages, ~my_pages, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_poke(relw, 1u);
      }
      if (lane == 0) wave_out[cur.item] = c;
      ++d_n;
      const u64 c2 = tick();
      if (!taken) take();
      if (!got) take_blocking();  // (nothing in hand)
      if (dbg) {
        const uint32_t dd = (uint32_t)(c2 - c1);
        d_land += (uint32_t)(c1 - c0);
        d_dec += dd;
        d_poll += (uint32_t)(tick() - c2);
        const uint32_t cls = (ta == kTypeBitmap && tb == kTypeBitmap) ? 4u
                             : (ta == kTypeArray && tb == kTypeArray) ? 0u
                             : ((ta == kTypeArray && tb == kTypeBitmap) || (ta == kTypeBitmap && tb == kTypeArray)) ? 1u
                             : ((ta == kTypeArray && tb == kTypeRun && cur.len_b <= kRunFillMax) || (ta == kTypeRun && tb == kTypeArray && cur.len_a <= kRunFillMax)) ? 2u : 3u;
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k)
          if (cls == k) d_cls[k] += dd, ++d_cln[k];
      }
    }
    if (dbg && lane == 0) {
      uint32_t* o = dbg + 32u * blockIdx.x;
      atomicAdd(o + 8, d_poll);
      atomicAdd(o + 9, d_dec);
      atomicAdd(o + 10, d_n);
      atomicAdd(o + 11, d_land);
      atomicAdd(o + 13, d_defer);
      for (int k = 0; k < 6; ++k) atomicAdd(o + 26 + k, d_ph[k]);
      for (int k = 0; k < 5; ++k) {
        atomicAdd(o + 16 + k, d_cls[k]);
        atomicAdd(o + 21 + k, d_cln[k]);
      }
    }
    return;
  }

  // -------------------------------------------------- planner --------------------------------------------------
  const uint64_t n_chunks = (n_items + kRgChunk - 1) / kRgChunk;
  uint32_t seq = 0, tail_seq = 0;  // entries written / queue slots taken back
  uint32_t head_pos = 0;           // the next item's ring position (bytes, a multiple of 1024, mod 2^32; RING divides 2^32)
  uint32_t l_rec = 0, l_slots = 0, l_polls = 0;
  const u64 l_t0 = tick();
  bool dead = false;
  // queue slots of finished entries, in order: one LDS round trip takes back up to 64 of them
  auto reclaim = [&]() {
    ++l_polls;
    const uint32_t e = (tail_seq + (uint32_t)lane) & (uint32_t)(kRgQ - 1);
    uint8_t* w = smem + L::kRel + e * 4u;
    const uint32_t r = ((uint32_t)lane < seq - tail_seq) ? lds_peek(w) : 0u;
    const u64 held = ~__ballot(r != 0u);
    const uint32_t k = held ? (uint32_t)__builtin_ctzll(held) : 64u;  // leading released entries
    if (k) {
      if ((uint32_t)lane < k) lds_poke(w, 0u);
      tail_seq += k;
    }
    return k;
  };
  auto dma_records = [&](uint64_t chunk, uint32_t buf) {
    const uint64_t first = chunk * kRgChunk;
    const uint32_t n = (uint32_t)min((uint64_t)kRgChunk, n_items - first);
    (void)glds_stream<false>(reinterpret_cast<uint64_t>(items + 2 * first), n * 32u, smem_base + L::kStage + buf * 1024u, voff);
  };

  uint64_t chunk = blockIdx.x;
  uint32_t buf = 0;
  if (chunk < n_chunks) dma_records(chunk, 0);
  while (chunk < n_chunks && !dead) {
    lds_poke(ctl + 20, 0x10000000u | (uint32_t)chunk);
    // the chunk after this one: a ticket from the launch's counter (blocks that draw light chunks draw more of them)
    uint32_t tkt = 0;
    if (lane == 0) tkt = atomicAdd(ctl_global + 1, 1u);
    const uint64_t next = (uint64_t)gridDim.x + rg_uniform(tkt);
    lds_poke(ctl + 20, 0x20000000u | (uint32_t)next);
    {
      const u64 c = tick();
      // this chunk's records have landed (the planner's vector-memory counter: record DMAs in order, plus stores that only make a wait stricter)
      if (next < n_chunks) {
        dma_records(next, buf ^ 1u);
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      l_rec += (uint32_t)(tick() - c);
    }
    lds_poke(ctl + 20, 0x30000000u | (uint32_t)chunk);
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
    // (a sparse payload of more than four batches — an array beyond 4096 values, roaring.go:5054, more than 2048 runs — does not fit
    // a decoder's registers: the host sends batches that may hold one to k_icount2, ring_regular() in fbk.hip; meeting one here is
    // a protocol error)
    const bool long_a = (ta == kTypeArray && ra.z > 2u * kRgWholeUnits) || (ta == kTypeRun && ra.z > kRgWholeUnits);
    const bool long_b = (tb == kTypeArray && rb.z > 2u * kRgWholeUnits) || (tb == kTypeRun && rb.z > kRgWholeUnits);
    if (__ballot(active && (long_a || long_b))) {
      dead = true;
      break;
    }
    const uint32_t need = active ? ba + bb : 0u;
    const uint32_t room = (need + 1023u) & ~1023u;  // whole pages
    const u64 amask = __ballot(active);
    const uint32_t cnt = (uint32_t)__popcll(amask);
    if (cnt) {
      const uint32_t rank = mbcnt64(amask, 0);
      uint32_t pos = head_pos + wave_incl_scan(room) - room;
      // an item does not straddle the ring's end: the first one that would is moved to the start, everything after it follows
      for (;;) {
        const u64 st = __ballot(need != 0u && (pos & (uint32_t)(RING - 1)) + room > (uint32_t)RING);
        if (!st) break;
        const uint32_t f = (uint32_t)__builtin_ctzll(st);
        const uint32_t bump = (uint32_t)RING - (rg_readlane(pos, f) & (uint32_t)(RING - 1));
        if ((uint32_t)lane >= f) pos += bump;
      }
      const uint32_t endpos = pos + room;
      lds_poke(ctl + 20, 0x40000000u | seq);
      // queue slots for the whole chunk
      {
        const u64 c = tick();
        for (uint32_t spins = 0; seq + cnt - tail_seq > (uint32_t)kRgQ; ++spins) {
          if (reclaim()) continue;
          if (rg_uniform(lds_peek(ctl + 12)) || spins > kSpinLimit) {
            dead = true;
            break;
          }
          __builtin_amdgcn_s_sleep(16);
        }
        l_slots += (uint32_t)(tick() - c);
      }
      if (dead) break;
      if (active) {
        const uint64_t ga = reinterpret_cast<uint64_t>(arenaA) + (((uint64_t)ra.y << 32) | ra.x), gb = reinterpret_cast<uint64_t>(arenaB) + (((uint64_t)rb.y << 32) | rb.x);
        uint4 q0, q1, q2;
        q0.x = ta | (tb << 4);
        q0.y = (uint32_t)item;
        q0.z = L::kRing + (pos & (uint32_t)(RING - 1));
        q0.w = ra.z;
        q1.x = ba;
        q1.y = rb.z;
        q1.z = (pos & (uint32_t)(RING - 1)) >> 10;
        q1.w = need;
        q2.x = (uint32_t)ga, q2.y = (uint32_t)(ga >> 32), q2.z = (uint32_t)gb, q2.w = (uint32_t)(gb >> 32);
        uint4* q = reinterpret_cast<uint4*>(smem + L::kQueue + ((seq + rank) & (uint32_t)(kRgQ - 1)) * (uint32_t)kRgEntry);
        q[0] = q0;
        q[1] = q1;
        q[2] = q2;
      }
      seq += cnt;
      head_pos = rg_readlane(endpos, 63u - (uint32_t)__builtin_clzll(amask));
      asm volatile("" ::: "memory");  // (the entries' plain LDS stores stay in front of the counter: LDS operations of a wave retire in order)
      lds_poke(ctl, seq);
    }
    lds_poke(ctl + 20, 0x50000000u | seq);
    (void)reclaim();
    chunk = next;
    buf ^= 1u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_poke(ctl + 8, seq);  // fin: a decoder whose ticket is past it leaves
  if (dead) give_up(3u, seq, tail_seq, head_pos);
  if (dbg && lane == 0) {
    uint32_t* o = dbg + 32u * blockIdx.x;
    o[0] = (uint32_t)(__builtin_readcyclecounter() - l_t0);
    o[1] = l_rec, o[2] = l_slots, o[6] = l_polls, o[7] = seq;
  }
}

}  // namespace fbk
