// fbk_kernels.hip.h — hand-written CDNA4 (gfx950) kernels for the roaring set-op /
// count hot path.  Integer bit-twiddling, HBM-bound: no MFMA.  Written for 64-lane
// wavefronts: ONE WAVEFRONT OWNS ONE CONTAINER SLOT (2^16 bits) and holds it as
// 16 x uint64 per lane (1024 words / 64 lanes), loaded with 8 coalesced 16-byte loads.
//
// Word <-> lane mapping used everywhere ("fragment layout"):
//     register w[2*j + h] of lane l  holds container word  128*j + 2*l + h     (j=0..7, h=0..1)
// so that load j of a wave covers one contiguous KiB of the 8 KiB bitmap container.
//
// Non-bitmap encodings are decoded on the fly into a per-wave 8 KiB LDS scratch:
//   array: zero scratch, ds_or one bit per element           (arrayToBitmap, roaring.go:3756)
//   run  : zero scratch, ds_xor a toggle bit at start and last+1, then a parity
//          prefix-scan (in-word shifts + wave ballot carry) turns toggles into filled
//          runs                                              (runToBitmap,  roaring.go:3792)
// so HBM traffic is the *encoded* payload (2n / 4*runs / 8192 bytes), never more.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fbk {

constexpr int kWave = 64;
constexpr int kSlots = 16;          // containers per shard row (fragment.go:47)
constexpr int kWords = 1024;        // bitmapN (roaring.go:44)
constexpr int kWordsPerLane = 16;   // 1024 / 64
constexpr uint32_t kTypeNil = 0, kTypeArray = 1, kTypeBitmap = 2, kTypeRun = 3;
constexpr uint32_t kDirectArrayMax = 1024;  // set-op results up to this many values are written as arrays when optimize() follows (4095: the peeling of large arrays into 8 KiB-strided cells triples the kernel time, measured)

// Device-side container descriptor: one 16-byte record per (row, slot), loaded with a
// single scalar dwordx4 load because it is wave-uniform.  type==0 means nil container.
struct alignas(16) Slot {
  uint64_t off;   // byte offset into the batch arena (16-byte aligned)
  uint32_t len;   // array: #u16, bitmap: 1024, run: #intervals
  uint32_t tn;    // type << 24 | n   (n <= 65536 needs 17 bits)
};
__host__ __device__ inline uint32_t slot_type(const Slot& s) { return s.tn >> 24; }
__host__ __device__ inline uint32_t slot_n(const Slot& s) { return s.tn & 0xFFFFFFu; }
__host__ __device__ inline uint32_t make_tn(uint32_t type, uint32_t n) { return (type << 24) | n; }

typedef unsigned long long u64;

// -DFBK_MM_STAMPS (scripts/matrix_xcd_hist.hip only, never the library): every block of k_count_matrix_mfma / k_icount_dense records
// when it started and ended (s_memrealtime, 100 MHz) and where it ran (XCC_ID, HW_ID) — four words per block at g_mm_stamps.
#ifdef FBK_MM_STAMPS
__device__ unsigned long long* g_mm_stamps;
#endif

// XCD-aware block index.  The dispatcher places block b on XCD b % 8, each XCD with its own L2: blocks with
// CONSECUTIVE indexes share nothing in cache.  Kernels whose neighbouring blocks read the same lines — the 16
// slots of one row's descriptor table (256 bytes = two lines per row, one block per slot), the rows of one
// shard — take their work item from this remap instead: XCD x gets the contiguous range of items
// [x * nwg / 8, (x + 1) * nwg / 8) (bijective for any nwg).
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t bid, uint32_t nwg) {
  const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS instructions of one wavefront are issued and retired in order; this only stops
  // the compiler from moving LDS accesses of other lanes' data across the point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 16 lanes of each DPP row, left in every lane of the row: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror — no LDS round trip (a ds_bpermute shuffle costs one)
__device__ __forceinline__ uint32_t wave_rows_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
  return v;
}

// inclusive prefix sum over the 64 lanes on the DPP network (no LDS round trips): Hillis-Steele inside
// each row of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15),
// then lane 31 into rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);
  return v;
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// Streaming (non-temporal) 16-byte global accesses.  Container payloads are read once
// per kernel and never reused by the same launch, so they should not displace each other
// in L2 / Infinity Cache: measured on MI355X with a 1 GiB working set, the dense
// |A∩B| kernel goes from 5.69 TB/s (plain loads) to 6.54 TB/s with `nt` loads
// (profiles/tune_dense_r01.txt).
__device__ __forceinline__ ulonglong2 ld_stream(const ulonglong2* p) {
  ulonglong2 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  return v;
}
__device__ __forceinline__ void st_stream(ulonglong2* p, ulonglong2 v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
}

// ---- fragment loaders --------------------------------------------------------------

__device__ __forceinline__ void frag_zero(u64 (&w)[kWordsPerLane]) {
#pragma unroll
  for (int i = 0; i < kWordsPerLane; ++i) w[i] = 0;
}

// bitmap container: 8 x global_load_dwordx4 per lane, each wave-load = 1 KiB contiguous.
__device__ __forceinline__ void frag_load_bitmap(const uint8_t* __restrict__ p, int lane,
                                                 u64 (&w)[kWordsPerLane]) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 v = ld_stream(&q[j * kWave + lane]);
    w[2 * j] = v.x;
    w[2 * j + 1] = v.y;
  }
}

// Same, with the default cache policy: for operands that other workgroups of the same XCD
// re-read soon (the B rows of the count matrix), so that they stay in L2.
__device__ __forceinline__ void frag_load_bitmap_cached(const uint8_t* __restrict__ p, int lane,
                                                        u64 (&w)[kWordsPerLane]) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 v = q[j * kWave + lane];
    w[2 * j] = v.x;
    w[2 * j + 1] = v.y;
  }
}

__device__ __forceinline__ void frag_store_bitmap(uint8_t* __restrict__ p, int lane,
                                                  const u64 (&w)[kWordsPerLane]) {
  ulonglong2* q = reinterpret_cast<ulonglong2*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 v;
    v.x = w[2 * j];
    v.y = w[2 * j + 1];
    st_stream(&q[j * kWave + lane], v);
  }
}

__device__ __forceinline__ void lds_zero(u64* scratch, int lane) {
  ulonglong2* q = reinterpret_cast<ulonglong2*>(scratch);
  ulonglong2 z;
  z.x = 0;
  z.y = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j * kWave + lane] = z;
}

__device__ __forceinline__ void lds_read_frag(const u64* scratch, int lane, u64 (&w)[kWordsPerLane]) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(scratch);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ulonglong2 v = q[j * kWave + lane];
    w[2 * j] = v.x;
    w[2 * j + 1] = v.y;
  }
}

// array container -> fragment.  Elements are sorted and unique, but several lanes can
// hit one dword, hence LDS atomics (ds_or_b32, no return).
__device__ __forceinline__ void frag_load_array(const uint8_t* __restrict__ p, uint32_t len, int lane,
                                                u64* scratch, u64 (&w)[kWordsPerLane]) {
  lds_zero(scratch, lane);
  wave_lds_sync();
  uint32_t* s32 = reinterpret_cast<uint32_t*>(scratch);
  // 4 elements (8 bytes) per lane per iteration; payloads are 16-byte aligned and the
  // arena is padded, but we never *use* elements past len.
  const uint2* q = reinterpret_cast<const uint2*>(p);
  const uint32_t nquad = (len + 3) >> 2;
  for (uint32_t i = lane; i < nquad; i += kWave) {
    uint2 v = q[i];
    uint32_t base = i << 2;
    uint32_t e0 = v.x & 0xFFFFu, e1 = v.x >> 16, e2 = v.y & 0xFFFFu, e3 = v.y >> 16;
    if (base + 0 < len) atomicOr(&s32[e0 >> 5], 1u << (e0 & 31));
    if (base + 1 < len) atomicOr(&s32[e1 >> 5], 1u << (e1 & 31));
    if (base + 2 < len) atomicOr(&s32[e2 >> 5], 1u << (e2 & 31));
    if (base + 3 < len) atomicOr(&s32[e3 >> 5], 1u << (e3 & 31));
  }
  wave_lds_sync();
  lds_read_frag(scratch, lane, w);
  wave_lds_sync();
}

__device__ __forceinline__ u64 prefix_xor64(u64 x) {
  x ^= x << 1;
  x ^= x << 2;
  x ^= x << 4;
  x ^= x << 8;
  x ^= x << 16;
  x ^= x << 32;
  return x;
}

// run container -> fragment.  Each interval [s,l] toggles bit s and bit l+1; the
// inclusive parity prefix of the toggle bitmap is the filled bitmap.  Adjacent runs
// ([0,5],[6,9]) toggle bit 6 twice and merge, as they should.
__device__ __forceinline__ void frag_load_run(const uint8_t* __restrict__ p, uint32_t len, int lane,
                                              u64* scratch, u64 (&w)[kWordsPerLane]) {
  lds_zero(scratch, lane);
  wave_lds_sync();
  uint32_t* s32 = reinterpret_cast<uint32_t*>(scratch);
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
  for (uint32_t i = lane; i < len; i += kWave) {
    uint32_t iv = q[i];
    uint32_t s = iv & 0xFFFFu, e = (iv >> 16) + 1u;
    atomicXor(&s32[s >> 5], 1u << (s & 31));
    if (e < 65536u) atomicXor(&s32[e >> 5], 1u << (e & 31));
  }
  wave_lds_sync();
  lds_read_frag(scratch, lane, w);
  wave_lds_sync();
  const u64 lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t carry = 0;  // parity of all toggles in earlier words
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

// Any container (wave-uniform dispatch on type; no divergence inside a wave).
// STREAM = true: non-temporal loads (payload read once); false: keep it in L2 for re-reads.
template <bool STREAM = true>
__device__ __forceinline__ void frag_load(const Slot& s, const uint8_t* __restrict__ arena, int lane,
                                          u64* scratch, u64 (&w)[kWordsPerLane]) {
  const uint32_t t = slot_type(s);
  const uint8_t* p = arena + s.off;
  if (t == kTypeBitmap) {
    if (STREAM) frag_load_bitmap(p, lane, w);
    else frag_load_bitmap_cached(p, lane, w);
  } else if (t == kTypeArray) {
    frag_load_array(p, s.len, lane, scratch, w);
  } else if (t == kTypeRun) {
    frag_load_run(p, s.len, lane, scratch, w);
  } else {
    frag_zero(w);
  }
}

template <int OP>
__device__ __forceinline__ u64 apply_op(u64 a, u64 b) {
  if (OP == 0) return a & b;   // intersect  (roaring.go:4971)
  if (OP == 1) return a | b;   // union      (roaring.go:5463)
  if (OP == 2) return a ^ b;   // xor        (roaring.go:6170)
  return a & ~b;               // difference (roaring.go:6040)
}

__device__ __forceinline__ uint32_t frag_popcount(const u64 (&w)[kWordsPerLane]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < kWordsPerLane; ++i) c += __popcll(w[i]);
  return c;
}

// Number of runs in a fragment (bitmapCountRuns, roaring.go:3372-3380): a run starts at
// every 1-bit whose predecessor bit is 0.  The predecessor of a word's bit 0 is bit 63
// of the previous container word, which lives in the neighbouring register / lane.
__device__ __forceinline__ uint32_t frag_count_runs(const u64 (&w)[kWordsPerLane], int lane) {
  uint32_t r = 0;
  uint32_t prev_iter_last = 0;  // top bit of word 128*j - 1 (lane 63's w[2j-1])
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u64 w0 = w[2 * j], w1 = w[2 * j + 1];
    uint32_t top1 = (uint32_t)(w1 >> 63);
    uint32_t left = __shfl_up(top1, 1, kWave);  // top bit of previous lane's w1
    if (lane == 0) left = prev_iter_last;
    r += __popcll(w0 & ~((w0 << 1) | (u64)left));
    r += __popcll(w1 & ~((w1 << 1) | (w0 >> 63)));
    prev_iter_last = __shfl(top1, 63, kWave);
  }
  return r;
}

// The wave's result fragment written the way Container.optimize() encodes it (roaring.go:3412-3461: runs when runs <= 2048
// and runs <= n / 2, an array when n < 4096, else the bitmap), exactly the encoded bytes, into `dst` (the head of the
// result's 8 KiB cell).  n and r (bitmapCountRuns of the fragment) are wave-uniform, n != 0.  Arrays and interval lists
// are staged in `stage` (8 KiB of LDS owned by this wave, free at this point) in value order — word 128 j + 2 lane + h sits
// in register 2 j + h, so a lane's first position in group j is the count of the earlier groups plus an exclusive scan
// over the lanes — and leave as coalesced 16-byte stores (peeling straight into global memory with 2-byte stores tripled
// the set-op kernels' time in round 2, which is why their own right-sized path stopped at 1024 values).
__device__ __forceinline__ void frag_store_encoded(const u64 (&w)[kWordsPerLane], uint32_t n, uint32_t r, int lane, u64* stage,
                                                   uint8_t* __restrict__ dst, uint32_t& out_type, uint32_t& out_len) {
  if (!(r <= 2048u && r <= n / 2u) && n >= 4096u) {
    frag_store_bitmap(dst, lane, w);
    out_type = kTypeBitmap;
    out_len = kWords;
    return;
  }
  uint16_t* s16 = reinterpret_cast<uint16_t*>(stage);
  const bool as_run = r <= 2048u && r <= n / 2u;
  if (!as_run) {
    uint32_t before = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t mine = (uint32_t)__popcll(w[2 * j]) + (uint32_t)__popcll(w[2 * j + 1]);
      const uint32_t incl = wave_incl_scan(mine);
      uint32_t pos = before + incl - mine;
      const uint32_t base = (128u * j + 2u * (uint32_t)lane) * 64u;
      for (u64 x = w[2 * j]; x; x &= x - 1) s16[pos++] = (uint16_t)(base + (uint32_t)__builtin_ctzll(x));
      for (u64 x = w[2 * j + 1]; x; x &= x - 1) s16[pos++] = (uint16_t)(base + 64u + (uint32_t)__builtin_ctzll(x));
      before += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  } else {
    // {start, last} pairs: start at 2k, last at 2k + 1 (bitmapToRun, roaring.go:3859)
    uint32_t sbase = 0, ebase = 0;
    uint32_t prev_top = 0;  // top bit of the word before this group's first word
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u64 w0 = w[2 * j], w1 = w[2 * j + 1];
      const uint32_t top1 = (uint32_t)(w1 >> 63);
      uint32_t left = __shfl_up(top1, 1, kWave);
      if (lane == 0) left = prev_top;
      // bit 0 of the word after this lane's w1: the next lane's w0, or the next group's lane-0 w0
      uint32_t next0 = (uint32_t)(__shfl_down((unsigned)(w0 & 1ull), 1, kWave));
      const uint32_t n0 = j < 7 ? (uint32_t)__shfl((unsigned)(w[j < 7 ? 2 * j + 2 : 0] & 1ull), 0, kWave) : 0u;
      if (lane == 63) next0 = n0;
      u64 st0 = w0 & ~((w0 << 1) | (u64)left);
      u64 st1 = w1 & ~((w1 << 1) | (w0 >> 63));
      u64 en0 = w0 & ~((w0 >> 1) | ((w1 & 1ull) << 63));
      u64 en1 = w1 & ~((w1 >> 1) | ((u64)next0 << 63));
      const uint32_t cs = (uint32_t)__popcll(st0) + (uint32_t)__popcll(st1), ce = (uint32_t)__popcll(en0) + (uint32_t)__popcll(en1);
      const uint32_t is = wave_incl_scan(cs), ie = wave_incl_scan(ce);
      uint32_t as = sbase + is - cs, ae = ebase + ie - ce;
      const uint32_t base = (128u * j + 2u * (uint32_t)lane) * 64u;
      for (; st0; st0 &= st0 - 1) s16[2 * (as++)] = (uint16_t)(base + (uint32_t)__builtin_ctzll(st0));
      for (; st1; st1 &= st1 - 1) s16[2 * (as++)] = (uint16_t)(base + 64u + (uint32_t)__builtin_ctzll(st1));
      for (; en0; en0 &= en0 - 1) s16[2 * (ae++) + 1] = (uint16_t)(base + (uint32_t)__builtin_ctzll(en0));
      for (; en1; en1 &= en1 - 1) s16[2 * (ae++) + 1] = (uint16_t)(base + 64u + (uint32_t)__builtin_ctzll(en1));
      sbase += (uint32_t)__builtin_amdgcn_readlane((int)is, 63);
      ebase += (uint32_t)__builtin_amdgcn_readlane((int)ie, 63);
      prev_top = (uint32_t)__shfl((int)top1, 63, kWave);
    }
  }
  wave_lds_sync();
  const uint32_t chunks = ((as_run ? 4u * r : 2u * n) + 15u) >> 4;  // <= 512
  const ulonglong2* src = reinterpret_cast<const ulonglong2*>(stage);
  ulonglong2* q = reinterpret_cast<ulonglong2*>(dst);
  for (uint32_t k = (uint32_t)lane; k < chunks; k += kWave) st_stream(&q[k], src[k]);
  wave_lds_sync();
  out_type = as_run ? kTypeRun : kTypeArray;
  out_len = as_run ? r : n;
}

// ---- kernels -------------------------------------------------------------------------

// Cardinality of every container of a batch whose n is unknown (the analogue of
// Container.count / bitmapRepair, roaring.go:3052,4193).  One wave per slot.
__global__ void __launch_bounds__(256) k_recount(Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                uint64_t n_slots) {
  const int lane = threadIdx.x & 63;
  const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= n_slots) return;
  Slot s = slots[wid];
  const uint32_t t = slot_type(s);
  if (t == kTypeNil) return;
  uint32_t c = 0;
  const uint8_t* p = arena + s.off;
  if (t == kTypeBitmap) {
    u64 w[kWordsPerLane];
    frag_load_bitmap(p, lane, w);
    c = frag_popcount(w);
  } else if (t == kTypeArray) {
    c = (lane == 0) ? s.len : 0;
  } else {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    for (uint32_t i = lane; i < s.len; i += kWave) {
      uint32_t iv = q[i];
      c += (iv >> 16) - (iv & 0xFFFFu) + 1u;  // Interval16.runlen, roaring.go:3047
    }
  }
  c = wave_reduce_add(c);
  if (lane == 0) slots[wid].tn = make_tn(t, c);
}

// Upload-time check of containers that did not pass the host validator (serialised roaring /
// RBF images are unpacked on the device and never parsed value by value on the host): arrays
// must be strictly ascending, runs ordered and non-overlapping (roaring.go:53-58; the XOR-toggle
// run decode and every `n == 65536` shortcut rely on it), and the header's cardinality is not
// trusted — bitmaps and runs are recounted, as bitmapRepair does (roaring.go:4193-4206).  One wave
// per slot; *bad receives bit 0 (array order) / bit 1 (run order).
__global__ void __launch_bounds__(256) k_validate_recount(Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                         uint64_t n_slots, uint32_t* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= n_slots) return;
  const Slot s = slots[wid];
  const uint32_t t = slot_type(s);
  if (t == kTypeNil) return;
  const uint8_t* p = arena + s.off;
  uint32_t c = 0, err = 0;
  if (t == kTypeBitmap) {
    u64 w[kWordsPerLane];
    frag_load_bitmap(p, lane, w);
    c = frag_popcount(w);
  } else if (t == kTypeArray) {
    const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
    for (uint32_t i = lane; i < s.len; i += kWave) {
      if (i + 1 < s.len && q[i + 1] <= q[i]) err = 1u;
      ++c;
    }
  } else {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    for (uint32_t i = lane; i < s.len; i += kWave) {
      const uint32_t iv = q[i], st = iv & 0xFFFFu, la = iv >> 16;
      if (la < st) err = 2u;
      else c += la - st + 1u;
      if (i + 1 < s.len && (q[i + 1] & 0xFFFFu) <= la) err = 2u;
    }
  }
  c = wave_reduce_add(c);
  const u64 anyerr = __ballot(err != 0);
  if (anyerr) {
    if (err) atomicOr(bad, err);
    return;
  }
  if (lane == 0) slots[wid].tn = make_tn(t, c);
}

// Window index of a batch: for every array / run container the position at which each eighth of
// the value range begins — win[w] (w = 0..7, 16 bits each, one uint4 per (row, slot)) = the index of
// the first array value >= w * 8192, or of the first run whose LAST value is >= w * 8192 (win[0] = 0;
// "none" = len).  The count-matrix kernel decodes rows 8192 bit positions at a time
// (fbk_matrix_fused.hip.h); with the index a stage's values of a row are a known piece of the array
// and nothing is searched or walked inside that kernel.  16 bytes per container, built once per batch
// (lazily, on the first count matrix that reads the batch) by streaming the arrays and run lists once.
// One wave per slot.
__global__ void __launch_bounds__(256) k_window_index(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                     uint64_t n_slots, uint4* __restrict__ win) {
  __shared__ uint16_t sw[4][8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t wid = (uint64_t)blockIdx.x * 4 + wv;
  if (wid >= n_slots) return;
  const Slot s = slots[wid];
  const uint32_t t = slot_n(s) ? slot_type(s) : kTypeNil;
  if (t != kTypeArray && t != kTypeRun) {
    if (lane == 0) win[wid] = uint4{0, 0, 0, 0};
    return;
  }
  const uint8_t* p = arena + s.off;
  if (lane < 8) sw[wv][lane] = (uint16_t)s.len;  // (len <= 65535 whenever some window has no value)
  wave_lds_sync();
  const uint32_t shift = t == kTypeArray ? 1u : 2u;  // array: uint16 values; run: {start, last} pairs, the key is `last`
  for (uint32_t i = lane; i < s.len; i += kWave) {
    const int k = (int)(*reinterpret_cast<const uint16_t*>(p + ((uint64_t)i << shift) + (t == kTypeRun ? 2u : 0u)) >> 13);
    const int pk = i ? (int)(*reinterpret_cast<const uint16_t*>(p + ((uint64_t)(i - 1) << shift) + (t == kTypeRun ? 2u : 0u)) >> 13) : -1;
    for (int w = pk + 1; w <= k; ++w) sw[wv][w] = (uint16_t)i;  // sorted input: each w is written by exactly one i
  }
  wave_lds_sync();
  if (lane == 0) {
    const uint16_t* q = sw[wv];
    win[wid] = uint4{(uint32_t)q[0] | ((uint32_t)q[1] << 16), (uint32_t)q[2] | ((uint32_t)q[3] << 16), (uint32_t)q[4] | ((uint32_t)q[5] << 16),
                     (uint32_t)q[6] | ((uint32_t)q[7] << 16)};
  }
}

// Bits of each row in [start, end) (Bitmap.CountRange, roaring.go:573-615): one wave per
// (row, slot).  Containers wholly inside the range contribute their stored n (roaring.go:603),
// the (at most two) boundary containers are loaded and masked (BitmapCountRange :3092,
// ArrayCountRange :3074, RunCountRange :3200 — here all three are "decode, mask, popcount").
__global__ void __launch_bounds__(256) k_count_range(const Slot* __restrict__ slots, const uint8_t* __restrict__ arena,
                                                    const uint32_t* __restrict__ rows, uint64_t n_rows, uint32_t start,
                                                    uint32_t end, u64* __restrict__ out, uint32_t run_quirk) {
  // One wavefront per ROW (round 4; it was one per (row, slot) with an atomic add each: 65 536 waves and as many atomics for
  // 4096 rows, 68 us).  Which slots lie wholly inside [start, end) and which one or two are cut by an end of the range is
  // the same for every row: lanes 0..15 add up the stored n of the interior slots (Bitmap.CountRange, roaring.go:573-621,
  // takes c.N() for those, :603), the whole wave decodes the boundary containers; one plain store per row.
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t r = (uint64_t)blockIdx.x * 4 + wv;
  if (r >= n_rows) return;
  const Slot* rowslots = slots + (uint64_t)rows[r] * kSlots;
  auto cut = [&](uint32_t slot, uint32_t& lo, uint32_t& hi) {
    const uint32_t base = slot << 16;
    lo = start > base ? min(start - base, 65536u) : 0u;
    hi = end > base ? min(end - base, 65536u) : 0u;
  };
  uint32_t total;
  {
    uint32_t c = 0;
    if (lane < kSlots) {
      uint32_t lo, hi;
      cut((uint32_t)lane, lo, hi);
      if (lo == 0 && hi == 65536u) c = slot_n(rowslots[lane]);
    }
    total = wave_reduce_add(c);
  }
  if (start < end) {
    const uint32_t s_lo = start >> 16, s_hi = (end - 1u) >> 16;
    for (uint32_t slot = s_lo;; slot = s_hi) {  // the slot of `start`, then (if another) the slot of `end - 1`
      uint32_t lo, hi;
      cut(slot, lo, hi);
      if (slot < (uint32_t)kSlots && lo < hi && !(lo == 0 && hi == 65536u)) {
        const Slot s = rowslots[slot];
        const uint32_t n = slot_n(s);
        if (n != 0) {
          uint32_t c;
          if (run_quirk && slot_type(s) == kTypeRun) {
            // option count_range_reference_quirk: RunCountRange AS WRITTEN (roaring.go:3200-3232).  Its tests mix an
            // exclusive range end with inclusive run ends: a run whose last value equals `end` is a "subset of the
            // range" (all of it counted, one value past the range) AND, when it starts after `start`, also "overlaps
            // the end" (counted again up to `end`).  Every run's contribution is a closed form of (start, last, lo, hi)
            // — the early `return end - start` is the only contribution when it fires, earlier runs end before `start`
            // — so the runs are summed lane-parallel.
            const uint32_t* q = reinterpret_cast<const uint32_t*>(arena + s.off);
            const int32_t st = (int32_t)lo, en = (int32_t)hi;
            int32_t part = 0;
            for (uint32_t i = lane; i < s.len; i += kWave) {
              const uint32_t iv = q[i];
              const int32_t rs = (int32_t)(iv & 0xFFFFu), rl = (int32_t)(iv >> 16);
              if (rl < st || en < rs) continue;
              if (rs <= st && rl >= en) {
                part += en - st;
                continue;
              }
              if (rs >= st && rl <= en) part += rl - rs + 1;
              if (rs < st && rl < en) part += rl - st + 1;
              if (rs > st && rl >= en) part += en - rs;
            }
            c = wave_reduce_add((uint32_t)part);
          } else {
            u64 w[kWordsPerLane];
            frag_load(s, arena, lane, lds[wv], w);
            uint32_t part = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const uint32_t b0 = (128u * j + 2u * lane + h) * 64u;  // first bit of this word
                // mask of the bits b0+i with lo <= b0+i < hi
                u64 m = ~0ull;
                if (lo > b0) m = (lo - b0 >= 64u) ? 0ull : (m << (lo - b0));
                if (hi < b0 + 64u) m = (hi <= b0) ? 0ull : (m & (~0ull >> (b0 + 64u - hi)));
                part += __popcll(w[2 * j + h] & m);
              }
            c = wave_reduce_add(part);
          }
          total += c;
        }
      }
      if (slot == s_hi) break;
    }
  }
  if (lane == 0) out[r] = (u64)total;
}

// |A ∩ B| for dense rows: every slot a bitmap container and each row one contiguous
// 128 KiB block (config 2, the HBM-roofline case).  One 256-thread block per
// (pair, group of SPB slots); thread t streams 16-byte chunks t, t+256, ... of its
// group from both operands.  No LDS staging (nothing is reused), no descriptors read.
template <int SPB>
__global__ void __launch_bounds__(256) k_icount_dense(const uint8_t* __restrict__ arenaA,
                                                     const uint32_t* __restrict__ rowsA,
                                                     const uint8_t* __restrict__ arenaB,
                                                     const uint32_t* __restrict__ rowsB,
                                                     u64* __restrict__ out, u64* __restrict__ total,
                                                     uint32_t* __restrict__ done, uint32_t n_pairs,
                                                     u64* __restrict__ accum) {
  constexpr int kGroups = kSlots / SPB;
#ifdef FBK_MM_STAMPS
  const unsigned long long stamp_t0 = wall_clock64();
#endif
  const uint32_t pair = blockIdx.x / kGroups;
  const uint32_t grp = blockIdx.x % kGroups;
  const uint64_t rowBytes = (uint64_t)kSlots * 8192;
  const ulonglong2* a = reinterpret_cast<const ulonglong2*>(arenaA + rowsA[pair] * rowBytes + (uint64_t)grp * SPB * 8192);
  const ulonglong2* b = reinterpret_cast<const ulonglong2*>(arenaB + rowsB[pair] * rowBytes + (uint64_t)grp * SPB * 8192);
  constexpr int kIters = SPB * 8192 / 16 / 256;  // 16-byte chunks per thread
  constexpr int kUnroll = kIters < 8 ? kIters : 8;
  uint32_t c = 0;
  // (Round 6 tried a per-block rotation of the 4 KiB chunks — every row is 128 KiB aligned and all blocks start together, so they
  // ask for the same offsets of their rows at the same time: 42.3-42.5 us with or without it, profiles/r06_icount_dense_blocks.txt.
  // The same stamps show what the end of a launch looks like: a CU serves its four blocks oldest first — they end at ~13, ~24, ~32
  // and ~37 us — and the CUs themselves end between 34 and 39 us.)
  for (int i0 = 0; i0 < kIters; i0 += kUnroll) {
    ulonglong2 va[kUnroll], vb[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) va[u] = ld_stream(&a[(i0 + u) * 256 + threadIdx.x]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) vb[u] = ld_stream(&b[(i0 + u) * 256 + threadIdx.x]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) c += __popcll(va[u].x & vb[u].x) + __popcll(va[u].y & vb[u].y);
  }
  c = wave_reduce_add(c);
  __shared__ uint32_t part[4];
  __shared__ uint32_t s_last;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
#ifdef FBK_MM_STAMPS
  if (threadIdx.x == 0 && g_mm_stamps) {
    unsigned long long* st = g_mm_stamps + 4ull * blockIdx.x;
    st[0] = stamp_t0;
    st[1] = wall_clock64();
    st[2] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11));  // HW_REG_XCC_ID[3:0]
    st[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));  // HW_REG_HW_ID
  }
#endif
  if (threadIdx.x == 0) {
    u64 tot = (u64)part[0] + part[1] + part[2] + part[3];
    s_last = 0;
    if (!total) {
      if (kGroups == 1) out[pair] = tot;
      else atomicAdd(&out[pair], tot);
      // per-node reduce by accumulation: every block adds its count to *accum, which the caller
      // zeroed beforehand — no ticket, no final pass, nothing serial behind the last workgroup
      if (accum && tot) atomicAdd(accum, tot);
    } else {
      // Fused per-node reduce (executeCount's reduceFn, executor.go:5880): the block that
      // finishes last sums the per-pair counts, so one step of the hot path is ONE launch.
      // The blocks run on 8 XCDs with separate L2s: the count is published with an agent-scope
      // atomic store (write-through, no L2-wide flush: a release FENCE here costs a full L2
      // write-back per block and doubled the kernel time), completed before the ticket is taken.
      if (kGroups == 1) __hip_atomic_store(&out[pair], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else atomicAdd(&out[pair], tot);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s_last = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1u : 0u;
    }
  }
  if (!total) return;
  __syncthreads();
  if (!s_last) return;
  u64 acc = 0;  // agent-scope loads: straight from the coherence point, no stale L2 lines
  for (uint32_t i = threadIdx.x; i < n_pairs; i += 256) acc += __hip_atomic_load(&out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  __shared__ u64 tpart[4];
  if ((threadIdx.x & 63) == 0) tpart[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    *total = tpart[0] + tpart[1] + tpart[2] + tpart[3];
    *done = 0;  // ready for the next launch (stream ordered)
  }
}

// A <op> B for dense rows, materialised as dense rows, cardinality fused in the same
// pass (roaring.go:4971-4974).  out row `pair` is written at out + pair*128 KiB and
// per-slot cardinalities into outSlots (the pair's cardinality is their sum: k_sum_slot_n).
template <int OP>
__global__ void __launch_bounds__(256) k_setop_dense(const uint8_t* __restrict__ arenaA,
                                                    const uint32_t* __restrict__ rowsA,
                                                    const uint8_t* __restrict__ arenaB,
                                                    const uint32_t* __restrict__ rowsB,
                                                    uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots) {
  // one wave per container slot; 4 slots per block
  const int lane = threadIdx.x & 63;
  const uint32_t wslot = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t pair = wslot >> 4, slot = wslot & 15;
  const uint64_t rowBytes = (uint64_t)kSlots * 8192;
  u64 wa[kWordsPerLane], wb[kWordsPerLane];
  frag_load_bitmap(arenaA + rowsA[pair] * rowBytes + slot * 8192ull, lane, wa);
  frag_load_bitmap(arenaB + rowsB[pair] * rowBytes + slot * 8192ull, lane, wb);
#pragma unroll
  for (int i = 0; i < kWordsPerLane; ++i) wa[i] = apply_op<OP>(wa[i], wb[i]);
  const uint64_t ooff = (uint64_t)pair * rowBytes + slot * 8192ull;
  frag_store_bitmap(arenaO + ooff, lane, wa);
  uint32_t c = wave_reduce_add(frag_popcount(wa));
  if (lane == 0) {
    Slot s;
    s.off = ooff;
    s.len = kWords;
    s.tn = make_tn(c ? kTypeBitmap : kTypeNil, c);
    outSlots[wslot] = s;
  }
}

// ---- small-operand fast paths (option sparse_paths) ------------------------------------------
// The generic pair kernels below decode BOTH operands into an 8 KiB LDS bitmap: zero 8 KiB, scatter,
// read 8 KiB back — per operand, whatever its size.  For the shapes the reference serves with
// intersectionCountArrayArray / intersectionCountArrayBitmap (roaring.go:4514-4536, 4596-4609) that
// is two orders of magnitude more LDS traffic than the operands have bytes; and an array x bitmap
// pair streams the whole 8 KiB bitmap from HBM to test a handful of values.
//   array x array, both <= 64 values   all-pairs compare in registers: the shorter array is broadcast
//                                      value by value (v_readlane), no LDS, no decode
//   array (<= 128 values) x bitmap     each value probes its dword of the bitmap straight from global
//                                      memory: <= 128 sectors instead of the whole container
constexpr uint32_t kSmallArray = 64;
constexpr uint32_t kProbeArray = 128;

// lanes whose value of the longer array also occurs in the shorter one (both <= 64 values)
__device__ __forceinline__ u64 small_arrays_match(const uint8_t* __restrict__ pa, uint32_t la, const uint8_t* __restrict__ pb,
                                                  uint32_t lb, int lane, uint32_t& my_value, bool& a_is_long) {
  const uint32_t a = (uint32_t)lane < la ? reinterpret_cast<const uint16_t*>(pa)[lane] : 0xFFFFFFFFu;
  const uint32_t b = (uint32_t)lane < lb ? reinterpret_cast<const uint16_t*>(pb)[lane] : 0xFFFFFFFEu;
  a_is_long = la >= lb;  // wave-uniform
  const uint32_t lng = a_is_long ? a : b, sht = a_is_long ? b : a;
  const uint32_t ns = a_is_long ? lb : la;
  bool m = false;
  for (uint32_t k = 0; k < ns; ++k) m |= lng == (uint32_t)__builtin_amdgcn_readlane((int)sht, (int)k);
  my_value = lng;
  return __ballot(m);
}

// lanes (of chunk `base`) whose array value is set in the bitmap container at pb
__device__ __forceinline__ u64 array_probe_bitmap(const uint8_t* __restrict__ pa, uint32_t la, uint32_t base,
                                                  const uint8_t* __restrict__ pb, int lane, uint32_t& my_value) {
  const uint32_t i = base + (uint32_t)lane;
  const bool on = i < la;
  const uint32_t a = on ? reinterpret_cast<const uint16_t*>(pa)[i] : 0u;
  const uint32_t w = on ? reinterpret_cast<const uint32_t*>(pb)[a >> 5] : 0u;
  my_value = a;
  return __ballot(on && ((w >> (a & 31u)) & 1u));
}

// Generic |A ∩ B| over row pairs with any mix of array / bitmap / run / nil containers
// (intersectionCount and its six kernels, roaring.go:4477-4614).  One wave per
// (pair, slot).  Short-circuits mirror roaring.go:4478-4486.
__global__ void __launch_bounds__(256) k_icount(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                               const uint32_t* __restrict__ rowsA,
                                               const Slot* __restrict__ slotsB, const uint8_t* __restrict__ arenaB,
                                               const uint32_t* __restrict__ rowsB, uint64_t n_pairs,
                                               u64* __restrict__ out, uint32_t sparse_paths) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  const uint64_t pair = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (pair >= n_pairs) return;
  const Slot sa = slotsA[(uint64_t)rowsA[pair] * kSlots + slot];
  const Slot sb = slotsB[(uint64_t)rowsB[pair] * kSlots + slot];
  const uint32_t na = slot_n(sa), nb = slot_n(sb);
  const uint32_t ta = slot_type(sa), tb = slot_type(sb);
  uint32_t c;
  if (na == 0 || nb == 0) {
    return;  // contributes 0
  } else if (na == 65536u) {
    c = nb;
  } else if (nb == 65536u) {
    c = na;
  } else if (sparse_paths && ta == kTypeArray && tb == kTypeArray && sa.len <= kSmallArray && sb.len <= kSmallArray) {
    uint32_t v;
    bool al;
    c = (uint32_t)__popcll(small_arrays_match(arenaA + sa.off, sa.len, arenaB + sb.off, sb.len, lane, v, al));
  } else if (sparse_paths && ta == kTypeArray && tb == kTypeBitmap && sa.len <= kProbeArray) {
    uint32_t v;
    c = 0;
    for (uint32_t base = 0; base < sa.len; base += kWave) c += (uint32_t)__popcll(array_probe_bitmap(arenaA + sa.off, sa.len, base, arenaB + sb.off, lane, v));
  } else if (sparse_paths && tb == kTypeArray && ta == kTypeBitmap && sb.len <= kProbeArray) {
    uint32_t v;
    c = 0;
    for (uint32_t base = 0; base < sb.len; base += kWave) c += (uint32_t)__popcll(array_probe_bitmap(arenaB + sb.off, sb.len, base, arenaA + sa.off, lane, v));
  } else {
    u64 wa[kWordsPerLane], wb[kWordsPerLane];
    frag_load(sa, arenaA, lane, lds[wv], wa);
    frag_load(sb, arenaB, lane, lds[wv], wb);
    uint32_t part = 0;
#pragma unroll
    for (int i = 0; i < kWordsPerLane; ++i) part += __popcll(wa[i] & wb[i]);
    c = wave_reduce_add(part);
  }
  if (lane == 0 && c) atomicAdd(&out[pair], (u64)c);
}

// Generic materialising A <op> B: result of every (pair, slot) is written as a bitmap
// container at a fixed 8 KiB cell of the output arena, with its cardinality and run
// count (for the optional optimize() re-encode pass).
template <int OP>
__global__ void __launch_bounds__(256) k_setop(const Slot* __restrict__ slotsA, const uint8_t* __restrict__ arenaA,
                                              const uint32_t* __restrict__ rowsA,
                                              const Slot* __restrict__ slotsB, const uint8_t* __restrict__ arenaB,
                                              const uint32_t* __restrict__ rowsB, uint64_t n_pairs,
                                              uint8_t* __restrict__ arenaO, Slot* __restrict__ outSlots,
                                              uint32_t* __restrict__ outRuns, uint32_t direct) {
  __shared__ u64 lds[4][kWords];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint64_t wslot = (uint64_t)blockIdx.x * 4 + wv;
  const uint64_t pair = wslot >> 4;
  const uint32_t slot = wslot & 15;
  if (pair >= n_pairs) return;
  const Slot sa = slotsA[(uint64_t)rowsA[pair] * kSlots + slot];
  const Slot sb = slotsB[(uint64_t)rowsB[pair] * kSlots + slot];
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
  if (OP == 0 && direct && (outRuns || direct == 2u)) {
    // Right-sized output (option setop_direct_encode, only when the caller asked for optimize()):
    // an intersection with a small array is a subset of that array — written as an ARRAY of <= 64
    // values into the cell (<= 128 bytes instead of 8 KiB; intersectArrayArray / intersectArrayBitmap,
    // roaring.go:4778-4830), with the run count optimize() needs taken from the sorted survivors.
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
      // a run starts at every survivor whose predecessor among the survivors is not value - 1
      const int prev = below ? 63 - __builtin_clzll(below) : 0;
      const uint32_t vprev = (uint32_t)__shfl((int)v, prev, kWave);
      const uint32_t r = (uint32_t)__popcll(__ballot(mine && (below == 0 || vprev + 1u != v)));
      const uint32_t c = (uint32_t)__popcll(mm);
      // (with the encoding decided HERE, survivors that optimize() would store as runs — r <= c / 2 — take the general path)
      if (!(direct == 2u && c != 0 && r <= c / 2u)) {
        if (mine) reinterpret_cast<uint16_t*>(arenaO + so.off)[__popcll(below)] = (uint16_t)v;
        if (lane == 0) {
          so.len = c;
          so.tn = make_tn(c ? kTypeArray : kTypeNil, c);
          outSlots[wslot] = so;
          if (outRuns) outRuns[wslot] = r;
        }
        return;
      }
    }
  }
  u64 wa[kWordsPerLane], wb[kWordsPerLane];
  frag_load(sa, arenaA, lane, lds[wv], wa);
  frag_load(sb, arenaB, lane, lds[wv], wb);
#pragma unroll
  for (int i = 0; i < kWordsPerLane; ++i) {
    wa[i] = apply_op<OP>(wa[i], wb[i]);
    // XOR only: without this barrier the optimiser re-derives a ^ b inside the population / run counts and the array peeling
    // below instead of keeping the result, and the kernel needs 146 registers instead of 83 (3 waves per SIMD: Ary4096 x
    // Ary1 73 us where Union takes 59)
    if (OP == 2) asm volatile("" : "+v"(wa[i]));
  }
  uint32_t c = wave_reduce_add(frag_popcount(wa));
  uint32_t r = 0;
  if (outRuns || direct == 2u) r = wave_reduce_add(frag_count_runs(wa, lane));
  if (direct == 2u && c != 0) {
    // option setop_direct_encode = 2 (round 4, the default when optimize() is asked for): Container.optimize() applied here,
    // the encoded container written into the head of the cell — no re-encode pass, no compaction, no host round trip
    uint32_t t_out, l_out;
    frag_store_encoded(wa, c, r, lane, lds[wv], arenaO + so.off, t_out, l_out);
    if (lane == 0) {
      so.len = l_out;
      so.tn = make_tn(t_out, c);
      outSlots[wslot] = so;
      if (outRuns) outRuns[wslot] = r;
    }
    return;
  }
  bool as_array = false;
  if (direct && outRuns && c != 0 && c <= kDirectArrayMax) {
    // Right-sized output for ANY operation (only when the caller asked for optimize()): a result of at most
    // kDirectArrayMax values leaves the kernel as an ARRAY of 2 c bytes in its cell instead of as an 8 KiB bitmap
    // that the re-encode pass would read back only to shrink it.  Values in word order: word 128 j + 2 lane + h
    // sits in this lane's register 2 j + h, so the position of a lane's first value in group j is the count of
    // all earlier groups plus an exclusive scan over the lanes (DPP network), then each lane peels its bits.
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
  } else if (direct != 2u) {  // (direct == 2 gets here only with an empty result: nil, nothing to write)
    frag_store_bitmap(arenaO + so.off, lane, wa);
  }
  if (lane == 0) {
    if (as_array) so.len = c;
    so.tn = make_tn(c ? (as_array ? kTypeArray : kTypeBitmap) : kTypeNil, c);
    outSlots[wslot] = so;
    if (outRuns) outRuns[wslot] = r;
  }
}

// total = sum(counts[0..n)): the per-node half of executeCount's reduceFn
// (executor.go:5880): one block, deterministic order (integer adds anyway).
__global__ void __launch_bounds__(256) k_sum_u64(const u64* __restrict__ counts, uint64_t n, u64* __restrict__ total) {
  u64 acc = 0;
  for (uint64_t i = threadIdx.x; i < n; i += 256) acc += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  __shared__ u64 part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *total = part[0] + part[1] + part[2] + part[3];
}

// *total += sum(counts[0..n)): the accumulate form of the per-node reduce (the caller zeroed *total)
__global__ void __launch_bounds__(256) k_sum_u64_add(const u64* __restrict__ counts, uint64_t n, u64* __restrict__ total) {
  u64 acc = 0;
  for (uint64_t i = threadIdx.x; i < n; i += 256) acc += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  __shared__ u64 part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(total, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace fbk
