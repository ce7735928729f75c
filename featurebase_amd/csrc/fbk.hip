// fbk.hip — C ABI (include/fbk.h) over the CDNA4 kernels in fbk_kernels.hip.h.
// Built with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
// No torch, no CPU fallback: every compute entry point launches a HIP kernel.
#include "../../include/fbk.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>  // device radix sort / unique for fbk_bsi_distinct (plumbing, not the hot path)

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "fbk_kernels.hip.h"
#include "fbk_pair_kernels.hip.h"
#ifdef FBK_EXPERIMENTS
#include "fbk_pair_ring.hip.h"  // k_icount3 (round 6): parity-green and 1.5 x SLOWER than k_icount2 — kept as an experiment, not shipped
#endif
#include "fbk_query_kernels.hip.h"
#include "fbk_fold_kernels.hip.h"
#include "fbk_topk_kernels.hip.h"
#include "fbk_bsi_kernels.hip.h"
#include "fbk_matrix_kernels.hip.h"
#include "fbk_matrix_mfma.hip.h"
#include "fbk_matrix_fused.hip.h"
#include "fbk_matrix_fused.hip.h"
#include "fbk_matrix_fusedq.hip.h"
#include "fbk_wire_kernels.hip.h"

using fbk::Slot;
using fbk::u64;

namespace {

// Error reporting.  Every failing call leaves its message in TWO places: a thread-local string
// (what fbk_last_error(NULL) returns: the only place for failures that have no context, e.g.
// fbk_open) and the context the call was made on (fbk_last_error_r / fbk_last_error(ctx)).  The
// per-context copy is what a cgo caller must read: a goroutine may be moved to another OS thread
// between the failing call and the call that fetches the message, so thread-local state alone is
// not reliable there.
thread_local std::string g_err = "";
thread_local fbk_ctx* g_scope_ctx = nullptr;  // context of the API call running on this thread
thread_local fbk_group* g_scope_group = nullptr;  // group of the fbk_group_* call running on this thread
void ctx_record_error(fbk_ctx* ctx, int32_t code, const std::string& msg);  // defined after fbk_ctx
void group_record_error(fbk_group* g, int32_t code, const std::string& msg);  // fbk_group_api.inc
void comm_release(fbk_ctx* ctx);                                                // fbk_group_api.inc

int32_t fail(int32_t code, const std::string& msg) {
  g_err = msg;
  if (g_scope_ctx) ctx_record_error(g_scope_ctx, code, msg);
  if (g_scope_group) group_record_error(g_scope_group, code, msg);
  return code;
}

// first statement of every entry point that takes a context
struct ApiScope {
  fbk_ctx* prev;
  explicit ApiScope(fbk_ctx* c) : prev(g_scope_ctx) { g_scope_ctx = c; }
  ~ApiScope() { g_scope_ctx = prev; }
};
#define FBK_ENTER(ctx) ApiScope api_scope_(ctx)

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (void)hipGetLastError();                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? FBK_E_NOMEM : FBK_E_HIP,                         \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                          \
    }                                                                                          \
  } while (0)

// Every extern "C" entry point is a function-try-block closed by this handler: the host side of the library uses std::vector /
// std::string / std::thread, and no exception may leave the C ABI (the caller is cgo: an exception unwinding into Go frames ends the
// server).  The locals — locks, FBK_ENTER's scope — are gone when the handler runs, so the context is entered again for the message.
#define FBK_ABI_CATCH(ctx)                                                                   \
  catch (const std::bad_alloc&) {                                                            \
    FBK_ENTER(ctx);                                                                          \
    return fail(FBK_E_NOMEM, std::string(__func__) + ": host allocation failed");            \
  }                                                                                          \
  catch (const std::exception& e_) {                                                         \
    FBK_ENTER(ctx);                                                                          \
    return fail(FBK_E_HIP, std::string(__func__) + ": " + e_.what());                        \
  }                                                                                          \
  catch (...) {                                                                              \
    FBK_ENTER(ctx);                                                                          \
    return fail(FBK_E_HIP, std::string(__func__) + ": unknown exception");                   \
  }

inline uint64_t align16(uint64_t x) { return (x + 15) & ~uint64_t(15); }

}  // namespace

// Tuning / test knobs.  Read from the environment ONCE, in fbk_open (FBK_<NAME>), and changed
// afterwards only through fbk_set_option: no entry point calls getenv.
struct FbkOptions {
  int64_t dense_spb = 16;                // slots per block of k_icount_dense: 1|2|4|8|16
  int64_t matrix_spb = 0;                // slots per block of the dense count matrix; 0 = chosen per launch
  int64_t matrix_tickets = 1;            // dense single-tile count matrix: 1 the blocks take their units from a ticket counter, long units first (the XCDs run at different paces); 0 by block id
  int64_t matrix_pass_kb = 1 << 20;      // per-shard matrices are produced in passes of at most this many KiB
  int64_t matrix_fused = -1;             // count matrix over encoded rows: 1 decode inside the matrix-core kernel, 0 the generic pair kernel, -1 by the matrix size
  int64_t matrix_fp4 = -1;               // dense count matrix on the FP4 matrix instruction: 1 always, 0 never, -1 when it has several tiles
  int64_t time_kernels = 0;              // 1: HIP events around the dominant kernel of a query-level call (count matrix, fold, BSI range / sum)
  int64_t last_kernel_ns = 0;            //    ... read its duration back here (fbk_get_option) after the call
  int64_t matrix_shadow = 1;             // count matrix over encoded rows: dense shadows of the heavy containers, built per batch on first use (heavy_shadow); 0: decode every container in every query
  int64_t matrix_shadow_array = 2048;    //   arrays longer than this are heavy
  int64_t matrix_shadow_run = 0;         //   run containers of more than this many runs are heavy (0: every run container — rounds 3-5)
  int64_t matrix_shadow_max_mb = 16384;  //   no shadow for a batch that would need more than this (nor more than twice its arena, nor a quarter of the free device memory)
  int64_t matrix_shadow_arena_x = 8;     //   ... nor more than this many times the batch's own arena (0: no such rule); fbk_batch_memory reports what a batch got
#ifdef FBK_EXPERIMENTS  // (scripts/ build their own variant with -DFBK_EXPERIMENTS into build_variants/; the product library has neither the options nor the device branches)
  int64_t matrix_fused_ablate = 0;       // timing experiments on the fused kernel (skips parts of it: WRONG results)
#endif
  int64_t topk_device_sort = -1;         // 1 / 0 pins the ordering path of fbk_topk, -1: by field size
  int64_t upload_threads = 0;            // host threads that fill the pinned upload buffers (0: min(8, cores / 2))
  int64_t upload_chunk_mb = 64;          // size of each of the two pinned upload buffers
  int64_t setop_direct_encode = 2;       // pair set-ops with optimize(): 2 the kernel applies Container.optimize() itself (encoded bytes into the head of the cell; no re-encode pass), 1 results of <= 1024 values leave the kernel as arrays and the re-encode pass does the rest (round 2), 0 always 8 KiB cells first (A/B runs, cross-checks)
  int64_t setop_compact = 1;             // one-shot set-ops / folds / BSI ranges / Flip / Shift with optimize() applied inside the kernel: 1 the output batch (owned by the caller) is compacted into a right-sized arena before it is returned (payload sizes + scan + one copy per container), 0 it keeps its 8 KiB cells
  int64_t count_range_reference_quirk = 1;  // 1 (default: identical to the reference): fbk_count_range reproduces RunCountRange's double count of a run ending at `end` (roaring.go:3216-3227); 0: the arithmetically right count
  int64_t topn_semantics = 1;            // fbk_topn / fbk_query_topn / fbk_group_topn / fbk_topn_partials with n > 0: 1 (default) the reference's two passes — candidates = the union over the SHARDS of fragment.top(N = n) (k_topn_candidates), then their exact totals (executeTopN, executor.go:2779-2864); 0: the exact top n of all rows
  int64_t pair_wpb = 0;                  // wavefronts per block of k_icount2 / k_setop2: 1 (a wave's LDS table is released when IT ends) or 4; 0 = by the rows' payload size
#ifdef FBK_EXPERIMENTS
  int64_t pair_stamp = 0;                // timing experiment on k_icount2: waves report shader cycles of a phase instead of counts (WRONG results)
  int64_t pair_ablate = 0;               // timing experiments on k_icount2 (skips parts of it: WRONG results)
  int64_t pair_spw = 1;                  // experiment: container slots per wave of k_icount2 (1 | 2 | 4): the next slot's first payload batch is in flight while the current one is decoded
#endif
  int64_t pair_kernels = 0;              // 2: type-pair specialised k_icount2 / k_setop2 (one LDS clear per pair, probing); 1: the round-2 kernels; 0: by the rows' average payload (use_pair_kernels2); experiments builds only: 3 = the persistent loader / decoder count k_icount3 (fbk_pair_ring.hip.h; set-ops as 2)
#ifdef FBK_EXPERIMENTS
  int64_t ring_geom = 0;                 // k_icount3's block: 0 = 10 decoders + 64 KiB ring, one block per CU; 1 = 8 decoders; 2 = 6 decoders; 3 = 5 decoders + 32 KiB ring, two blocks per CU
  int64_t ring_nt = 0;                   //   1: the payload DMAs carry the non-temporal hint
  int64_t ring_flags = 0;                //   experiments on k_icount3 (bit 0: ring space is released after the decode)
  int64_t ring_debug = 0;                //   1: every block reports the cycles its loader and decoders spent waiting; the averages go to stderr after each launch (which is then synchronous)#endif
#endif
};

struct fbk_ctx {
  int device = 0;
  int n_cu = 0;  // compute units of the device (sizes persistent grids)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // fbk_comm_*: this rank's RCCL communicator (one process per GPU), its stream and the two events that order it against `stream`
  void* comm = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev_in = nullptr, comm_ev_out = nullptr;
  std::mutex mu;
  FbkOptions opt;
  // last failing call on this context (fbk_last_error_r): guarded by err_mu, NOT by mu, so that
  // a failure recorded before / after the call took the context lock never deadlocks
  std::mutex err_mu;
  std::string err_msg;
  int32_t err_code = 0;
  uint64_t err_seq = 0;
  // fbk_ctx_fork: a child context has its own stream, lock, pool and staging (so that concurrent
  // callers overlap on the device) and shares the fragment cache of the root context
  fbk_ctx* root = nullptr;  // nullptr: this IS a root context
  std::atomic<int> children{0};
  // Device-memory pool: every query needs a handful of temporaries (row index arrays, count
  // vectors, the output arena); hipMalloc / hipFree cost 50-200 us each and hipFree
  // synchronises the device, which on a 100-400 us query was most of the wall time
  // (measured: BSI Range 452 us per call with a 142 us kernel).  Freed blocks are kept in
  // size buckets and handed out again; everything on one context runs on one stream, so reuse
  // is ordered by the stream.
  std::mutex pool_mu;
  std::unordered_map<uint64_t, std::vector<void*>> pool_free_lists;
  std::unordered_map<void*, uint64_t> pool_live;  // block -> bucket size
  uint64_t pool_cached_bytes = 0;
  uint64_t pool_cap_bytes = 8ull << 30;
  // Pinned host staging for the small index arrays every query uploads: a copy from pageable
  // memory is staged by the runtime and costs 10-20 us per call; from pinned memory it is one
  // asynchronous DMA.  Bump-allocated, reset at the start of every API call (each call that
  // uses it synchronises the stream before returning).
  uint8_t* h_stage = nullptr;
  uint64_t h_stage_cap = 0, h_stage_used = 0;
  // Bulk uploads (fbk_batch_upload / _dense): two pinned buffers, one being filled by host threads while the other's DMA
  // runs (staged_h2d).  Allocated on the first bulk upload, kept for the context's lifetime.
  uint8_t* up_ring[2] = {nullptr, nullptr};
  hipEvent_t up_ev[2] = {nullptr, nullptr};
  uint64_t up_cap = 0;
  // option time_kernels: events around the dominant kernel of the last query-level call
  hipEvent_t kt0 = nullptr, kt1 = nullptr;
  bool kt_armed = false;
  // ticket counters of the launches whose blocks take their work by ticket (ticket_counter): 64 words, zero between launches
  uint32_t* d_tickets = nullptr;
  // device fragment cache (fbk_cache_api.inc)
  std::unordered_map<std::string, struct fbk_cache_entry*> cache;
  std::unordered_map<const fbk_batch*, struct fbk_cache_entry*> cache_by_batch;  // release() looks entries up by handle
  std::vector<struct fbk_cache_entry*> cache_zombies;  // invalidated while pinned
  uint64_t cache_bytes = 0, cache_cap_bytes = 128ull << 30, cache_clock = 0;
  uint64_t cache_hits = 0, cache_misses = 0, cache_evictions = 0;
};

// Host mirror of one device-resident batch.
struct fbk_batch {
  fbk_ctx* ctx = nullptr;
  uint32_t n_rows = 0;
  uint64_t arena_bytes = 0;
  uint8_t* d_arena = nullptr;  // payloads, each 16-byte aligned and padded
  Slot* d_slots = nullptr;     // n_rows * 16
  bool dense = false;          // every slot a bitmap at (row*16+slot)*8192
  bool slots_stale = false;    // host copy must be refreshed from device before use
  std::vector<Slot> h_slots;   // host copy of the descriptors (types, n, offsets)
  std::vector<uint64_t> h_keys;  // container key per slot (carried through to download)
  // window index (k_window_index): 16 bytes per slot, built on the first count matrix that reads the
  // batch, dropped whenever the containers are rewritten (plan outputs, optimize)
  mutable uint4* d_win = nullptr;
  mutable std::mutex win_mu;
  // dense shadows of the heavy containers (run containers, long arrays), built on the first count matrix over mixed rows
  // (heavy_shadow): a copy of the descriptor table in which those containers are bitmaps in a shadow arena.  The matrix-core
  // kernel with in-kernel decode then streams them like any bitmap row instead of decoding them in every query.  Under win_mu,
  // dropped with the window index.  shadow_state: 0 not looked at, 1 built, 2 nothing heavy / over the memory cap.  The
  // parameters (options matrix_shadow, matrix_shadow_array, the cap) are those of the context whose count matrix touched the
  // batch FIRST: a batch shared between contexts keeps that decision until its containers are rewritten
  mutable uint8_t* d_shadow_arena = nullptr;
  mutable Slot* d_shadow_slots = nullptr;
  mutable uint64_t shadow_bytes = 0;
  mutable int shadow_state = 0;
  std::mutex slots_mu;  // refresh_slots: h_slots / slots_stale
  uint64_t version = 0;  // bumped whenever the device descriptors are rewritten (a plan's resolved item records follow it)
  bool borrowed = false;  // the output of a plan / prepared query: owned by it and rewritten in place (n_rows x 16 cells of 8 KiB) by its next run
  // every sparse container fits a k_icount3 decoder's registers (arrays <= 4096 values, run lists <= 2048 intervals); false = not
  // known: uploads check their descriptors, kernel outputs are 8 KiB cells.  Batches that are not go to k_icount2.
  bool ring_regular = false;
};
inline bool ring_fits(uint32_t type, uint32_t len) { return !((type == fbk::kTypeArray && len > 4096u) || (type == fbk::kTypeRun && len > 2048u)); }

namespace {

void ctx_record_error(fbk_ctx* ctx, int32_t code, const std::string& msg) {
  std::lock_guard<std::mutex> g(ctx->err_mu);
  ctx->err_msg = msg;
  ctx->err_code = code;
  ++ctx->err_seq;
}

uint64_t pool_bucket(uint64_t bytes) {
  if (bytes <= 256) return 256;
  if (bytes <= (1ull << 20)) {  // next power of two
    uint64_t b = 256;
    while (b < bytes) b <<= 1;
    return b;
  }
  return (bytes + (1ull << 21) - 1) & ~((1ull << 21) - 1);  // 2 MiB granules
}

// The context's ticket counters (kernels whose blocks take their work by ticket: the counter wraps back to zero inside the
// launch, so it is allocated and cleared ONCE; everything on a context runs on one stream).  Caller holds ctx->mu.
int32_t ticket_counter(fbk_ctx* ctx, uint32_t** out) {
  if (!ctx->d_tickets) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, 256));
    hipError_t e = hipMemsetAsync(p, 0, 256, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (once per context: the caller may move the context to another stream afterwards)
    if (e != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(p);
      return fail(FBK_E_HIP, std::string("ticket counter: ") + hipGetErrorString(e));
    }
    ctx->d_tickets = static_cast<uint32_t*>(p);
  }
  *out = ctx->d_tickets;
  return FBK_OK;
}

hipError_t ctx_malloc(fbk_ctx* ctx, void** out, uint64_t bytes) {
  // 64 bytes of slack behind every allocation: kernels read array payloads with 16-byte loads at
  // 2-byte alignment (fbk_matrix_fused.hip.h), up to 14 bytes past the padded end of the last container
  const uint64_t bucket = pool_bucket(bytes + 64);
  {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    auto it = ctx->pool_free_lists.find(bucket);
    if (it != ctx->pool_free_lists.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      ctx->pool_cached_bytes -= bucket;
      ctx->pool_live[*out] = bucket;
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bucket);
  if (e == hipErrorOutOfMemory) {  // give the cache back and retry once
    (void)hipGetLastError();
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> g(ctx->pool_mu);
      for (auto& kv : ctx->pool_free_lists) {
        drop.insert(drop.end(), kv.second.begin(), kv.second.end());
        kv.second.clear();
      }
      ctx->pool_cached_bytes = 0;
    }
    for (void* p : drop) (void)hipFree(p);
    e = hipMalloc(out, bucket);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    ctx->pool_live[*out] = bucket;
  }
  return e;
}

void ctx_free(fbk_ctx* ctx, void* p) {
  if (!p) return;
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    auto it = ctx->pool_live.find(p);
    if (it != ctx->pool_live.end()) {
      const uint64_t bucket = it->second;
      ctx->pool_live.erase(it);
      if (bucket <= ctx->pool_cap_bytes) {
        // the block just returned is the likeliest to be asked for again: keep it, and when the cache is over its cap give
        // back the largest blocks cached before it (other sizes first) until it fits
        std::vector<void*> drop;
        while (ctx->pool_cached_bytes + bucket > ctx->pool_cap_bytes) {
          uint64_t victim = 0;
          for (auto& kv : ctx->pool_free_lists)
            if (!kv.second.empty() && kv.first != bucket && kv.first > victim) victim = kv.first;
          if (!victim) {
            auto own = ctx->pool_free_lists.find(bucket);
            if (own == ctx->pool_free_lists.end() || own->second.empty()) break;
            victim = bucket;
          }
          auto& fl = ctx->pool_free_lists[victim];
          drop.push_back(fl.front());
          fl.erase(fl.begin());
          ctx->pool_cached_bytes -= victim;
        }
        ctx->pool_free_lists[bucket].push_back(p);
        ctx->pool_cached_bytes += bucket;
        for (void* d : drop) (void)hipFree(d);
        return;
      }
    }
  }
  (void)hipFree(p);
}

// hand a live block from one context's book-keeping to another's (same device)
void pool_rehome(fbk_ctx* from, fbk_ctx* to, void* p) {
  if (!p || from == to) return;
  uint64_t bucket = 0;
  {
    std::lock_guard<std::mutex> g(from->pool_mu);
    auto it = from->pool_live.find(p);
    if (it == from->pool_live.end()) return;
    bucket = it->second;
    from->pool_live.erase(it);
  }
  std::lock_guard<std::mutex> g(to->pool_mu);
  to->pool_live[p] = bucket;
}

void pool_release_all(fbk_ctx* ctx) {
  std::lock_guard<std::mutex> g(ctx->pool_mu);
  for (auto& kv : ctx->pool_free_lists)
    for (void* p : kv.second) (void)hipFree(p);
  ctx->pool_free_lists.clear();
  ctx->pool_cached_bytes = 0;
}

void cache_release_all(fbk_ctx* ctx);  // fbk_cache_api.inc needs fbk_batch: defined after it

struct DevBuf {
  fbk_ctx* c = nullptr;
  void* p = nullptr;
  uint64_t cap = 0;  // bytes asked for by the last alloc / ensure
  hipError_t alloc(fbk_ctx* ctx, uint64_t bytes) {
    c = ctx;
    const hipError_t e = ctx_malloc(ctx, &p, bytes);
    cap = e == hipSuccess ? bytes : 0;
    return e;
  }
  // a buffer kept between calls (a prepared query's): at least `bytes`, reallocated when a later call needs more.
  // The old block goes back to the pool; everything on a context is ordered by its one stream, so work still
  // queued on it finishes before the block's next user starts.
  hipError_t ensure(fbk_ctx* ctx, uint64_t bytes) {
    if (p && cap >= bytes) return hipSuccess;
    if (p) ctx_free(c, p);
    p = nullptr;
    return alloc(ctx, bytes);
  }
  ~DevBuf() {
    if (p) ctx_free(c, p);
  }
  template <class T>
  T* as() {
    return static_cast<T*>(p);
  }
};

int32_t set_device(fbk_ctx* ctx) {
  g_scope_ctx = ctx;  // entry points that take the context from a batch / plan handle (ApiScope restores the caller's)
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->h_stage_used = 0;
  return FBK_OK;
}

// n bytes of pinned staging valid until the next API call on this context, or nullptr
uint8_t* stage_alloc(fbk_ctx* ctx, uint64_t n) {
  if (!ctx->h_stage) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 8u << 20, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    ctx->h_stage = static_cast<uint8_t*>(p);
    ctx->h_stage_cap = 8u << 20;
  }
  const uint64_t at = (ctx->h_stage_used + 63) & ~uint64_t(63);
  if (at + n > ctx->h_stage_cap) return nullptr;
  ctx->h_stage_used = at + n;
  return ctx->h_stage + at;
}

// Small results back to the host through the pinned staging area: queue, then ONE stream
// synchronisation and plain memcpys (a device->pageable copy is staged by the runtime and
// synchronises by itself, once per call).
struct D2H {
  struct Item {
    void* dst;
    const uint8_t* st;
    uint64_t n;
  };
  fbk_ctx* ctx;
  std::vector<Item> items;
  explicit D2H(fbk_ctx* c) : ctx(c) {}
  hipError_t add(void* dst, const void* dev, uint64_t n) {
    if (n == 0) return hipSuccess;
    if (uint8_t* st = stage_alloc(ctx, n)) {
      items.push_back({dst, st, n});
      return hipMemcpyAsync(st, dev, n, hipMemcpyDeviceToHost, ctx->stream);
    }
    return hipMemcpyAsync(dst, dev, n, hipMemcpyDeviceToHost, ctx->stream);
  }
  hipError_t finish() {
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess)
      for (const Item& it : items) std::memcpy(it.dst, it.st, it.n);
    items.clear();
    return e;
  }
};

// The host copy of a batch's descriptors, refreshed after the device rewrote them (plan outputs, optimize).  Any
// context of the device may read a batch (fbk_ctx_fork): the refresh is serialised per BATCH (two forks reading a
// stale plan output would otherwise race on the host vector) and runs on the CALLING context's stream, whose lock
// the caller holds — never on the owner's stream behind the owner's back.  The owner's stream is drained first when
// the caller is another context, so that the rewrite that made the copy stale has landed.
int32_t refresh_slots(fbk_batch* b) {
  std::lock_guard<std::mutex> g(b->slots_mu);
  if (!b->slots_stale) return FBK_OK;
  fbk_ctx* c = g_scope_ctx && g_scope_ctx->device == b->ctx->device ? g_scope_ctx : b->ctx;
  if (c != b->ctx) HIP_TRY(hipStreamSynchronize(b->ctx->stream));
  HIP_TRY(hipMemcpyAsync(b->h_slots.data(), b->d_slots, b->h_slots.size() * sizeof(Slot), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  b->slots_stale = false;
  return FBK_OK;
}

// The containers of `b` were (re)written on the device: the host descriptors are stale and the
// window index no longer describes the payloads.  (Called with the stream's earlier readers of the
// index behind the rewrite in stream order, or already synchronised.)
void slots_rewritten(fbk_batch* b) {
  b->slots_stale = true;
  ++b->version;
  std::lock_guard<std::mutex> g(b->win_mu);
  if (b->d_win) {
    (void)ctx_free(b->ctx, b->d_win);
    b->d_win = nullptr;
  }
  if (b->d_shadow_arena) (void)ctx_free(b->ctx, b->d_shadow_arena);
  if (b->d_shadow_slots) (void)ctx_free(b->ctx, b->d_shadow_slots);
  b->d_shadow_arena = nullptr;
  b->d_shadow_slots = nullptr;
  b->shadow_bytes = 0;
  b->shadow_state = 0;
}

// The window index of `b`, built on first use (one pass over its arrays and run lists) on the
// calling context's stream and complete when this returns.
int32_t window_index(fbk_ctx* ctx, const fbk_batch* b, const uint4** out) {
  std::lock_guard<std::mutex> g(b->win_mu);
  if (!b->d_win) {
    const uint64_t n_slots = uint64_t(b->n_rows) * fbk::kSlots;
    uint4* w = nullptr;
    // from the CALLING context's pool: a pooled block is only safe to reuse on the stream that freed it (reuse is
    // ordered by the stream), and the kernel below runs on the caller's stream — a block the owner had freed with
    // work still pending on ITS stream would be overwritten under that work.  The finished index then belongs to
    // the batch's owner, which frees it with the batch.
    HIP_TRY(ctx_malloc(ctx, reinterpret_cast<void**>(&w), std::max<uint64_t>(n_slots, 1) * sizeof(uint4)));
    if (n_slots) hipLaunchKernelGGL(fbk::k_window_index, dim3(uint32_t((n_slots + 3) / 4)), dim3(256), 0, ctx->stream, b->d_slots, b->d_arena, n_slots, w);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      (void)ctx_free(ctx, w);
      return fail(FBK_E_HIP, std::string("window index: ") + hipGetErrorString(e));
    }
    pool_rehome(ctx, b->ctx, w);
    b->d_win = w;
  }
  *out = b->d_win;
  return FBK_OK;
}

// The descriptor table the in-kernel-decode count matrix should read for `b`: the batch's own, or — option matrix_shadow — a copy
// in which every HEAVY container (a run container, an array of more than matrix_shadow_array values) is a bitmap in a shadow
// arena built here on first use.  The kernel that decodes rows in place is bound by vector instruction issue, and what it issued them for
// was decoding exactly those containers again in every query (397 us for 581 MB of config 3's rows); as bitmap
// rows they cost one 16-byte load per lane and stage.  The trade is the review's option (a): more bytes per query (8 KiB per
// heavy container instead of its payload) and resident memory for the shadows (capped: matrix_shadow_max_mb), paid once per
// batch like the window index and amortised over every query on a cached fragment.  Shadow descriptors address the shadow arena
// relative to the batch's arena (the kernels form arena + off in 64-bit arithmetic).
int32_t heavy_shadow(fbk_ctx* ctx, const fbk_batch* b, const Slot** out_slots, bool* out_shadowed) {
  *out_slots = b->d_slots;
  *out_shadowed = false;
  if (!ctx->opt.matrix_shadow) return FBK_OK;
  if (int32_t rc = refresh_slots(const_cast<fbk_batch*>(b))) return rc;
  std::lock_guard<std::mutex> g(b->win_mu);
  if (b->shadow_state == 0) {
    const uint64_t n_slots = uint64_t(b->n_rows) * fbk::kSlots;
    const uint32_t thr = uint32_t(ctx->opt.matrix_shadow_array), thr_run = uint32_t(ctx->opt.matrix_shadow_run);
    std::vector<uint32_t> list;
    for (uint64_t i = 0; i < n_slots && i < b->h_slots.size(); ++i) {
      const Slot& s = b->h_slots[i];
      const uint32_t n = s.tn & 0xFFFFFFu, t = s.tn >> 24;
      if (n != 0 && ((t == fbk::kTypeRun && s.len > thr_run) || (t == fbk::kTypeArray && s.len > thr))) list.push_back(uint32_t(i));
    }
    b->shadow_state = 2;
    const uint64_t bytes = uint64_t(list.size()) * 8192ull;
    // no shadow beyond the option's cap (matrix_shadow_max_mb), beyond matrix_shadow_arena_x times the batch's own arena, or beyond a quarter of what the device has
    // free right now (the shadows live outside the fragment cache's accounting until its next get / release: they must
    // never be what makes a later upload fail).  The threshold and this decision are the FIRST caller's: the shadow is
    // built once per batch content.
    uint64_t limit = uint64_t(ctx->opt.matrix_shadow_max_mb) << 20;
    if (ctx->opt.matrix_shadow_arena_x)  // (0: no such rule — run-heavy batches, a few bytes per container and 8 KiB per shadow, are what shadows were built for)
      limit = std::min<uint64_t>(limit, uint64_t(ctx->opt.matrix_shadow_arena_x) * std::max<uint64_t>(b->arena_bytes, 1 << 20));
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) limit = std::min<uint64_t>(limit, (uint64_t(mem_free) + ctx->pool_cached_bytes) / 4);
    else (void)hipGetLastError();
    if (!list.empty() && bytes <= limit) {
      std::vector<Slot> hs(b->h_slots);
      uint8_t* arena = nullptr;
      Slot* dslots = nullptr;
      DevBuf dlist;
      hipError_t e = ctx_malloc(ctx, reinterpret_cast<void**>(&arena), bytes);
      if (e == hipSuccess) e = ctx_malloc(ctx, reinterpret_cast<void**>(&dslots), n_slots * sizeof(Slot));
      if (e == hipSuccess) e = dlist.alloc(ctx, list.size() * 4);
      if (e == hipSuccess) {
        const uint64_t rel = uint64_t(reinterpret_cast<uintptr_t>(arena)) - uint64_t(reinterpret_cast<uintptr_t>(b->d_arena));
        for (size_t k = 0; k < list.size(); ++k) {
          Slot& s = hs[list[k]];
          s.off = rel + uint64_t(k) * 8192ull;
          s.len = fbk::kWords;
          s.tn = fbk::make_tn(fbk::kTypeBitmap, s.tn & 0xFFFFFFu);
        }
        e = hipMemcpyAsync(dlist.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dslots, hs.data(), n_slots * sizeof(Slot), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
          hipLaunchKernelGGL(fbk::k_shadow_build, dim3(uint32_t((list.size() + 3) / 4)), dim3(256), 0, ctx->stream, b->d_slots, b->d_arena,
                             dlist.as<uint32_t>(), uint64_t(list.size()), arena);
          e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (the host vectors above are read by the copies)
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (arena) (void)ctx_free(ctx, arena);
        if (dslots) (void)ctx_free(ctx, dslots);
        if (e != hipErrorOutOfMemory) return fail(FBK_E_HIP, std::string("heavy-row shadow: ") + hipGetErrorString(e));
        // out of memory: the query runs on the encoded rows
      } else {
        pool_rehome(ctx, b->ctx, arena);
        pool_rehome(ctx, b->ctx, dslots);
        b->d_shadow_arena = arena;
        b->d_shadow_slots = dslots;
        b->shadow_bytes = bytes + n_slots * sizeof(Slot);
        b->shadow_state = 1;
      }
    }
  }
  if (b->shadow_state == 1) {
    *out_slots = b->d_shadow_slots;
    *out_shadowed = true;
  }
  return FBK_OK;
}

// Option time_kernels: events on the context's stream right before and after the dominant kernel(s) of a
// query-level call, so that a caller (bench.py's secondary configurations) can separate the kernel from the
// row-index upload, memsets, reduce and download around it.  The last span of a call wins.
struct KernelSpan {
  fbk_ctx* c;
  explicit KernelSpan(fbk_ctx* ctx) : c(ctx->opt.time_kernels ? ctx : nullptr) {
    if (!c) return;
    if (!c->kt0 && (hipEventCreate(&c->kt0) != hipSuccess || hipEventCreate(&c->kt1) != hipSuccess)) {
      c = nullptr;
      return;
    }
    (void)hipEventRecord(c->kt0, c->stream);
  }
  ~KernelSpan() {
    if (!c) return;
    (void)hipEventRecord(c->kt1, c->stream);
    c->kt_armed = true;
  }
  void restart() {  // what was enqueued so far inside the span was preparation (a prepared program built on its first run), not the kernel
    if (c) (void)hipEventRecord(c->kt0, c->stream);
  }
};

int32_t upload_rows(fbk_ctx* ctx, const uint32_t* rows, uint64_t n, uint32_t n_rows_limit, DevBuf& out) {
  for (uint64_t i = 0; n_rows_limit != UINT32_MAX && i < n; ++i)  // UINT32_MAX: the caller has validated the indices
    if (rows[i] >= n_rows_limit)
      return fail(FBK_E_INVALID, "row index " + std::to_string(rows[i]) + " out of range (batch has " +
                                     std::to_string(n_rows_limit) + " rows)");
  HIP_TRY(out.alloc(ctx, std::max<uint64_t>(n, 1) * sizeof(uint32_t)));
  if (n) {
    const void* src = rows;
    if (uint8_t* st = stage_alloc(ctx, n * sizeof(uint32_t))) {
      std::memcpy(st, rows, n * sizeof(uint32_t));
      src = st;
    }
    HIP_TRY(hipMemcpyAsync(out.p, src, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  }
  return FBK_OK;
}

// Several row-index arrays of one call in ONE host->device copy (each copy is a launch of its own:
// ~5 us apiece on the calls that take rows for A, B and the filter).  d[i] points into `out`.
struct RowsArg {
  const uint32_t* rows;
  uint64_t n;
  uint32_t limit;  // UINT32_MAX: validated by the caller
};
int32_t upload_rows_multi(fbk_ctx* ctx, std::initializer_list<RowsArg> args, DevBuf& out, const uint32_t** d) {
  uint64_t total = 0;
  for (const RowsArg& a : args) {
    for (uint64_t i = 0; a.limit != UINT32_MAX && i < a.n; ++i)
      if (a.rows[i] >= a.limit)
        return fail(FBK_E_INVALID, "row index " + std::to_string(a.rows[i]) + " out of range (batch has " +
                                       std::to_string(a.limit) + " rows)");
    total += (a.n + 3) & ~uint64_t(3);  // 16-byte aligned segments
  }
  HIP_TRY(out.alloc(ctx, std::max<uint64_t>(total, 4) * sizeof(uint32_t)));
  uint8_t* st = total ? stage_alloc(ctx, total * sizeof(uint32_t)) : nullptr;
  uint64_t at = 0;
  int k = 0;
  for (const RowsArg& a : args) {
    d[k++] = static_cast<const uint32_t*>(out.p) + at;
    if (a.n) {
      if (st) std::memcpy(st + at * sizeof(uint32_t), a.rows, a.n * sizeof(uint32_t));
      else HIP_TRY(hipMemcpyAsync(static_cast<uint32_t*>(out.p) + at, a.rows, a.n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    at += (a.n + 3) & ~uint64_t(3);
  }
  if (st && total) HIP_TRY(hipMemcpyAsync(out.p, st, total * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  return FBK_OK;
}

bool is_gfx950(const hipDeviceProp_t& p) { return std::strncmp(p.gcnArchName, "gfx950", 6) == 0; }

}  // namespace

extern "C" {

int32_t fbk_abi_version(void) { return FBK_ABI_VERSION; }

const char* fbk_last_error(fbk_ctx* ctx) {
  if (!ctx) return g_err.c_str();
  // the context's last message, copied under its lock into this thread's buffer (the pointer stays
  // valid until this thread's next call into the library)
  thread_local std::string copy;
  {
    std::lock_guard<std::mutex> g(ctx->err_mu);
    copy = ctx->err_msg;
  }
  return copy.c_str();
}

int32_t fbk_last_error_r(fbk_ctx* ctx, char* buf, uint64_t cap, int32_t* out_code) try {
  FBK_ENTER(ctx);
  std::string msg;
  int32_t code = 0;
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->err_mu);
    msg = ctx->err_msg;
    code = ctx->err_code;
  } else {
    msg = g_err;
  }
  if (out_code) *out_code = code;
  if (buf && cap) {
    const uint64_t n = std::min<uint64_t>(cap - 1, msg.size());
    std::memcpy(buf, msg.data(), n);
    buf[n] = 0;
  }
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_device_count(int32_t* out_n) try {
  if (!out_n) return fail(FBK_E_INVALID, "out_n is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *out_n = 0;
    return fail(FBK_E_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *out_n = n;
  return FBK_OK;
} FBK_ABI_CATCH(nullptr)

}  // extern "C"

namespace {

struct OptionDesc {
  const char* name;  // fbk_set_option name; the environment variable is FBK_<NAME in upper case>
  int64_t FbkOptions::*field;
  int64_t lo, hi;
};
const OptionDesc kOptions[] = {
    {"dense_spb", &FbkOptions::dense_spb, 1, 16},
    {"matrix_spb", &FbkOptions::matrix_spb, 0, 16},
    {"matrix_tickets", &FbkOptions::matrix_tickets, 0, 1},
    {"matrix_pass_kb", &FbkOptions::matrix_pass_kb, 1, int64_t(1) << 40},
    {"matrix_fused", &FbkOptions::matrix_fused, -1, 1},
    {"matrix_fp4", &FbkOptions::matrix_fp4, -1, 1},
    {"matrix_shadow", &FbkOptions::matrix_shadow, 0, 1},
    {"matrix_shadow_array", &FbkOptions::matrix_shadow_array, 0, 65536},
    {"matrix_shadow_run", &FbkOptions::matrix_shadow_run, 0, 65536},
    {"matrix_shadow_max_mb", &FbkOptions::matrix_shadow_max_mb, 0, 1 << 20},
    {"matrix_shadow_arena_x", &FbkOptions::matrix_shadow_arena_x, 0, 1 << 20},
#ifdef FBK_EXPERIMENTS
    {"matrix_fused_ablate", &FbkOptions::matrix_fused_ablate, 0, 63},
#endif
    {"time_kernels", &FbkOptions::time_kernels, 0, 1},
    {"last_kernel_ns", &FbkOptions::last_kernel_ns, 0, INT64_MAX},
    {"topk_device_sort", &FbkOptions::topk_device_sort, -1, 1},
    {"upload_threads", &FbkOptions::upload_threads, 0, 64},
    {"upload_chunk_mb", &FbkOptions::upload_chunk_mb, 1, 1024},
    {"setop_direct_encode", &FbkOptions::setop_direct_encode, 0, 2},
#ifndef FBK_EXPERIMENTS
    {"pair_kernels", &FbkOptions::pair_kernels, 0, 2},
#else
    {"pair_kernels", &FbkOptions::pair_kernels, 0, 3},
    {"ring_geom", &FbkOptions::ring_geom, 0, 3},
    {"ring_nt", &FbkOptions::ring_nt, 0, 1},
    {"ring_debug", &FbkOptions::ring_debug, 0, 1},
    {"ring_flags", &FbkOptions::ring_flags, 0, 255},
    {"pair_ablate", &FbkOptions::pair_ablate, 0, 4095},
    {"pair_stamp", &FbkOptions::pair_stamp, 0, 4},
    {"pair_spw", &FbkOptions::pair_spw, 1, 4},
#endif
    {"pair_wpb", &FbkOptions::pair_wpb, 0, 4},
    {"setop_compact", &FbkOptions::setop_compact, 0, 1},
    {"count_range_reference_quirk", &FbkOptions::count_range_reference_quirk, 0, 1},
    {"topn_semantics", &FbkOptions::topn_semantics, 0, 1},
};

int32_t option_set(FbkOptions& o, const char* name, int64_t v) {
  for (const OptionDesc& d : kOptions)
    if (std::strcmp(d.name, name) == 0) {
      if (v < d.lo || v > d.hi) return fail(FBK_E_INVALID, std::string("option ") + name + ": value out of range");
      if ((d.field == &FbkOptions::dense_spb || (d.field == &FbkOptions::matrix_spb && v)) && (v & (v - 1)))
        return fail(FBK_E_INVALID, std::string("option ") + name + ": must be a power of two");
      o.*(d.field) = v;
      return FBK_OK;
    }
  return fail(FBK_E_INVALID, std::string("unknown option ") + name);
}

void options_from_env(FbkOptions& o) {
  for (const OptionDesc& d : kOptions) {
    std::string env = "FBK_";
    for (const char* c = d.name; *c; ++c) env += char(std::toupper(*c));
    if (const char* e = getenv(env.c_str())) (void)option_set(o, d.name, strtoll(e, nullptr, 10));
  }
}

int32_t open_on_device(int32_t device, fbk_ctx* root, fbk_ctx** out_ctx) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(FBK_E_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
  }
  if (device < 0 || device >= n) return fail(FBK_E_INVALID, "device ordinal out of range");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (!is_gfx950(prop))
    return fail(FBK_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  fbk_ctx* ctx = new (std::nothrow) fbk_ctx();
  if (!ctx) return fail(FBK_E_NOMEM, "host allocation failed");
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  hipError_t e2 = hipSetDevice(device);
  if (e2 == hipSuccess) e2 = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e2 != hipSuccess) {
    delete ctx;
    return fail(FBK_E_HIP, std::string("stream create: ") + hipGetErrorString(e2));
  }
  ctx->stream = ctx->own_stream;
  if (root) {
    ctx->root = root;
    ctx->opt = root->opt;
    ctx->pool_cap_bytes = root->pool_cap_bytes;
    root->children.fetch_add(1);
  } else {
    // the only place the environment is read
    // cached (freed, reusable) device blocks: an eighth of the device, at least 8 GiB; given back on an allocation failure
    ctx->pool_cap_bytes = std::max<uint64_t>(ctx->pool_cap_bytes, uint64_t(prop.totalGlobalMem) / 8);
    if (const char* cap = getenv("FBK_POOL_MAX_BYTES")) ctx->pool_cap_bytes = strtoull(cap, nullptr, 10);
    options_from_env(ctx->opt);
  }
  *out_ctx = ctx;
  return FBK_OK;
}

}  // namespace

extern "C" {

int32_t fbk_open(int32_t device, uint32_t /*flags*/, fbk_ctx** out_ctx) try {
  if (!out_ctx) return fail(FBK_E_INVALID, "out_ctx is NULL");
  *out_ctx = nullptr;
  return open_on_device(device, nullptr, out_ctx);
} FBK_ABI_CATCH(nullptr)

int32_t fbk_ctx_fork(fbk_ctx* ctx, fbk_ctx** out_child) try {
  FBK_ENTER(ctx);
  if (!ctx || !out_child) return fail(FBK_E_INVALID, "NULL argument");
  *out_child = nullptr;
  fbk_ctx* root = ctx->root ? ctx->root : ctx;
  return open_on_device(root->device, root, out_child);
} FBK_ABI_CATCH(ctx)

int32_t fbk_close(fbk_ctx* ctx) try {
  FBK_ENTER(ctx);
  if (!ctx) return FBK_OK;
  if (!ctx->root && ctx->children.load() != 0) {
    FBK_ENTER(ctx);
    return fail(FBK_E_INVALID, "fbk_close: forked contexts of this context are still open");
  }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  comm_release(ctx);
  if (!ctx->root) cache_release_all(ctx);
  pool_release_all(ctx);
  if (ctx->d_tickets) (void)hipFree(ctx->d_tickets);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (int k = 0; k < 2; ++k) {
    if (ctx->up_ring[k]) (void)hipHostFree(ctx->up_ring[k]);
    if (ctx->up_ev[k]) (void)hipEventDestroy(ctx->up_ev[k]);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->root) ctx->root->children.fetch_sub(1);
  delete ctx;
  return FBK_OK;
} FBK_ABI_CATCH(nullptr)  // (the context may be gone: nothing is recorded on it)

int32_t fbk_set_option(fbk_ctx* ctx, const char* name, int64_t value) try {
  FBK_ENTER(ctx);
  if (!ctx || !name) return fail(FBK_E_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ctx->mu);
  return option_set(ctx->opt, name, value);
} FBK_ABI_CATCH(ctx)

int32_t fbk_get_option(fbk_ctx* ctx, const char* name, int64_t* out_value) try {
  FBK_ENTER(ctx);
  if (!ctx || !name || !out_value) return fail(FBK_E_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (ctx->kt_armed && std::strcmp(name, "last_kernel_ns") == 0) {
    if (int32_t rc = set_device(ctx)) return rc;
    float ms = 0;
    HIP_TRY(hipEventSynchronize(ctx->kt1));
    HIP_TRY(hipEventElapsedTime(&ms, ctx->kt0, ctx->kt1));
    ctx->opt.last_kernel_ns = int64_t(double(ms) * 1e6);
    ctx->kt_armed = false;
  }
  for (const OptionDesc& d : kOptions)
    if (std::strcmp(d.name, name) == 0) {
      *out_value = ctx->opt.*(d.field);
      return FBK_OK;
    }
  return fail(FBK_E_INVALID, std::string("unknown option ") + name);
} FBK_ABI_CATCH(ctx)

int32_t fbk_set_stream(fbk_ctx* ctx, void* hip_stream) try {
  FBK_ENTER(ctx);
  if (!ctx) return fail(FBK_E_INVALID, "ctx is NULL");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  // pooled blocks freed under the old stream may be reused under the new one: drain it first
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_synchronize(fbk_ctx* ctx) try {
  FBK_ENTER(ctx);
  if (!ctx) return fail(FBK_E_INVALID, "ctx is NULL");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_batch_free(fbk_ctx* ctx, fbk_batch* b) try {
  FBK_ENTER(ctx);
  if (!b) return FBK_OK;
  if (!ctx) ctx = b->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (b->d_arena) (void)ctx_free(b->ctx, b->d_arena);
  if (b->d_slots) (void)ctx_free(b->ctx, b->d_slots);
  if (b->d_win) (void)ctx_free(b->ctx, b->d_win);
  if (b->d_shadow_arena) (void)ctx_free(b->ctx, b->d_shadow_arena);
  if (b->d_shadow_slots) (void)ctx_free(b->ctx, b->d_shadow_slots);
  delete b;
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

// ---- bulk host -> device copies -----------------------------------------------------------------------------------
// `total` bytes of an image the caller describes by fill(off, len, dst): "write bytes [off, off + len) of the image to dst".
// The image goes through two pinned buffers: while the DMA of one runs, a few host threads fill the other (each its own
// slice, so fill must be safe to call concurrently on disjoint ranges).  hipMemcpy from pageable memory stages through the
// runtime's own bounce buffers on ONE thread: 1.3 GB/s for the 581 MB of config 3's rows when the host-side validation and
// the assembly of the arena were counted in, 7.9 GB/s for a plain 256 MiB copy (round-3 bench line).  Leaves the copies
// enqueued on the context's stream; the caller synchronises.
extern "C++" {
template <class Fill>
hipError_t staged_h2d(fbk_ctx* ctx, uint8_t* d_dst, uint64_t total, Fill fill) {
  if (total == 0) return hipSuccess;
  const uint64_t want = uint64_t(ctx->opt.upload_chunk_mb) << 20;
  if (ctx->up_cap != want) {
    for (int k = 0; k < 2; ++k) {
      if (ctx->up_ring[k]) (void)hipHostFree(ctx->up_ring[k]);
      ctx->up_ring[k] = nullptr;
    }
    ctx->up_cap = 0;
    for (int k = 0; k < 2; ++k) {
      void* p = nullptr;
      hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
      if (e == hipSuccess) ctx->up_ring[k] = static_cast<uint8_t*>(p);  // (owned by the context from here on: freed by the next resize or by fbk_close, also when the event below fails)
      if (e == hipSuccess && !ctx->up_ev[k]) e = hipEventCreateWithFlags(&ctx->up_ev[k], hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
    ctx->up_cap = want;
  }
  unsigned nt = ctx->opt.upload_threads ? unsigned(ctx->opt.upload_threads) : std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
  bool used[2] = {false, false};
  uint64_t c = 0;
  for (uint64_t off = 0; off < total; off += ctx->up_cap, ++c) {
    const int k = int(c & 1);
    const uint64_t len = std::min<uint64_t>(ctx->up_cap, total - off);
    if (used[k]) {
      const hipError_t e = hipEventSynchronize(ctx->up_ev[k]);  // the DMA that last read this buffer
      if (e != hipSuccess) return e;
    }
    uint8_t* dst = ctx->up_ring[k];
    const uint64_t per = ((len + nt - 1) / nt + 4095) & ~uint64_t(4095);
    const unsigned parts = unsigned((len + per - 1) / per);
    if (parts <= 1) {
      fill(off, len, dst);
    } else {
      // No exception may cross the C ABI (a cgo caller's process would terminate): a thread that cannot be started
      // (EAGAIN at the process's thread limit -> std::system_error) or a failed reserve leaves its piece, and every later
      // one, to the calling thread.
      std::vector<std::thread> th;
      unsigned started = 1;
      try {
        th.reserve(parts - 1);
        for (unsigned t = 1; t < parts; ++t) {
          const uint64_t a = uint64_t(t) * per, n = std::min<uint64_t>(per, len - a);
          th.emplace_back([&fill, off, a, n, dst] { fill(off + a, n, dst + a); });
          started = t + 1;
        }
      } catch (...) {
      }
      fill(off, std::min<uint64_t>(per, len), dst);
      for (unsigned t = started; t < parts; ++t) {
        const uint64_t a = uint64_t(t) * per;
        fill(off + a, std::min<uint64_t>(per, len - a), dst + a);
      }
      for (std::thread& x : th) x.join();
    }
    hipError_t e = hipMemcpyAsync(d_dst + off, dst, len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->up_ev[k], ctx->stream);
    if (e != hipSuccess) return e;
    used[k] = true;
  }
  return hipSuccess;
}
}  // extern "C++"

static int32_t validate_container(const fbk_container_desc& d, const uint8_t* /*payload*/, uint64_t payload_len,
                                  uint64_t* bytes) {
  uint64_t need;
  switch (d.type) {
    case FBK_TYPE_ARRAY:
      if (d.len > 65536) return fail(FBK_E_INVALID, "array container longer than 65536");
      need = uint64_t(d.len) * 2;
      break;
    case FBK_TYPE_BITMAP:
      if (d.len != FBK_BITMAP_WORDS) return fail(FBK_E_INVALID, "bitmap container must have len 1024");
      need = 8192;
      break;
    case FBK_TYPE_RUN:
      if (d.len > 32768) return fail(FBK_E_INVALID, "run container with more than 32768 intervals");
      need = uint64_t(d.len) * 4;
      break;
    default:
      return fail(FBK_E_INVALID, "unknown container type " + std::to_string(d.type));
  }
  if (d.off > payload_len || need > payload_len - d.off) return fail(FBK_E_INVALID, "container payload out of bounds");
  if (d.n < -1 || d.n > 65536) return fail(FBK_E_INVALID, "container n out of range");
  // The CONTENT rules — arrays strictly ascending (sorted []uint16, roaring.go:53-58), runs ordered and non-overlapping — are
  // checked on the device after the copy (k_validate_recount, one wavefront per container): walking 581 MB value by value
  // on one host thread was most of the 0.44 s the round-3 bench line reports for the upload of config 3's rows.
  if (d.type == FBK_TYPE_ARRAY && d.n >= 0 && uint32_t(d.n) != d.len) return fail(FBK_E_INVALID, "array container n != len");
  *bytes = need;
  return FBK_OK;
}

static int32_t batch_upload_impl(fbk_ctx* ctx, const fbk_container_desc* descs, uint64_t n_desc, uint32_t n_rows, const void* payload, uint64_t payload_len,
                                 fbk_batch** out_batch);

int32_t fbk_batch_upload(fbk_ctx* ctx, const fbk_container_desc* descs, uint64_t n_desc, uint32_t n_rows,
                         const void* payload, uint64_t payload_len, fbk_batch** out_batch) try {
  FBK_ENTER(ctx);
  // the host-side tables of an upload are std::vectors: their allocation failures must not leave the C ABI as exceptions
  try {
    return batch_upload_impl(ctx, descs, n_desc, n_rows, payload, payload_len, out_batch);
  } catch (const std::bad_alloc&) {
    return fail(FBK_E_NOMEM, "batch_upload: host allocation failed");
  } catch (const std::exception& e) {
    return fail(FBK_E_INVALID, std::string("batch_upload: ") + e.what());
  }
} FBK_ABI_CATCH(ctx)

static int32_t batch_upload_impl(fbk_ctx* ctx, const fbk_container_desc* descs, uint64_t n_desc, uint32_t n_rows, const void* payload, uint64_t payload_len,
                                 fbk_batch** out_batch) {
  if (!ctx || !out_batch || (n_desc && (!descs || !payload))) return fail(FBK_E_INVALID, "NULL argument");
  *out_batch = nullptr;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  const uint8_t* pay = static_cast<const uint8_t*>(payload);
  const uint64_t n_slots = uint64_t(n_rows) * fbk::kSlots;

  fbk_batch* b = new (std::nothrow) fbk_batch();
  if (!b) return fail(FBK_E_NOMEM, "host allocation failed");
  b->ctx = ctx;
  b->n_rows = n_rows;
  b->h_slots.assign(n_slots, Slot{0, 0, 0});
  b->h_keys.assign(n_slots, 0);
  std::vector<int64_t> src(n_slots, -1);
  std::vector<uint64_t> nbytes(n_slots, 0);
  bool need_recount = false;
  for (uint64_t i = 0; i < n_desc; ++i) {
    const fbk_container_desc& d = descs[i];
    if (d.row >= n_rows) {
      delete b;
      return fail(FBK_E_INVALID, "container row ordinal out of range");
    }
    const uint64_t s = uint64_t(d.row) * fbk::kSlots + (d.key & 15);
    if (src[s] >= 0) {
      delete b;
      return fail(FBK_E_INVALID, "duplicate container for (row, key&15)");
    }
    uint64_t bytes = 0;
    if (int32_t rc = validate_container(d, pay, payload_len, &bytes)) {
      delete b;
      return rc;
    }
    src[s] = int64_t(i);
    nbytes[s] = bytes;
    b->h_keys[s] = d.key;
  }
  // Lay the arena out in (row, slot) order, 16-byte aligned and padded, so that a dense
  // row is one contiguous 128 KiB block and consecutive wavefronts read consecutive KiBs.
  uint64_t off = 0;
  bool dense = n_slots > 0;
  for (uint64_t s = 0; s < n_slots; ++s) {
    if (src[s] < 0) {
      dense = false;
      continue;
    }
    const fbk_container_desc& d = descs[src[s]];
    if (d.n == 0 || d.len == 0) {  // empty containers are stored as nil (roaring.go:751-752)
      src[s] = -1;
      dense = false;
      continue;
    }
    Slot& hs = b->h_slots[s];
    hs.off = off;
    hs.len = d.len;
    hs.tn = fbk::make_tn(d.type, d.n < 0 ? 0x00FFFFFFu : uint32_t(d.n));
    // the caller's n is not trusted for bitmaps and runs (every `n == 65536` shortcut and every
    // buffer sized from a cardinality depends on it): recounted on the device, as bitmapRepair does
    if (d.n < 0 || d.type != FBK_TYPE_ARRAY) need_recount = true;
    if (d.type != FBK_TYPE_BITMAP || off != s * 8192ull) dense = false;
    off += align16(nbytes[s]);
  }
  b->arena_bytes = off;
  b->dense = dense;
  b->ring_regular = true;
  for (uint64_t s = 0; s < n_slots; ++s)
    if (src[s] >= 0 && !ring_fits(fbk::slot_type(b->h_slots[s]), b->h_slots[s].len)) b->ring_regular = false;
  // the arena image in (row, slot) order: payload bytes, then zeros up to the next 16-byte boundary
  struct Piece {
    uint64_t at, bytes, src;
  };
  std::vector<Piece> pieces;
  pieces.reserve(n_desc);
  for (uint64_t s = 0; s < n_slots; ++s)
    if (src[s] >= 0) pieces.push_back(Piece{b->h_slots[s].off, nbytes[s], descs[src[s]].off});
  auto fill = [&pieces, pay](uint64_t a, uint64_t len, uint8_t* dst) {
    const uint64_t e = a + len;
    size_t i = size_t(std::upper_bound(pieces.begin(), pieces.end(), a, [](uint64_t v, const Piece& p) { return v < p.at; }) - pieces.begin());
    if (i) --i;  // the piece that starts at or before a
    for (; i < pieces.size() && pieces[i].at < e; ++i) {
      const Piece& p = pieces[i];
      const uint64_t pe = p.at + p.bytes, ze = p.at + align16(p.bytes);
      if (pe > a) {
        const uint64_t x0 = std::max(a, p.at), x1 = std::min(e, pe);
        if (x1 > x0) std::memcpy(dst + (x0 - a), pay + p.src + (x0 - p.at), x1 - x0);
      }
      const uint64_t z0 = std::max(a, pe), z1 = std::min(e, ze);
      if (z1 > z0) std::memset(dst + (z0 - a), 0, z1 - z0);
    }
  };

  DevBuf dbad;
  uint32_t h_bad = 0;
  hipError_t e = ctx_malloc(ctx, reinterpret_cast<void**>(&b->d_arena), std::max<uint64_t>(off, 16));
  if (e == hipSuccess) e = ctx_malloc(ctx, reinterpret_cast<void**>(&b->d_slots), std::max<uint64_t>(n_slots, 1) * sizeof(Slot));
  if (e == hipSuccess) e = dbad.alloc(ctx, 16);
  if (e == hipSuccess) e = hipMemsetAsync(dbad.p, 0, 4, ctx->stream);
  if (e == hipSuccess && n_slots)
    e = hipMemcpyAsync(b->d_slots, b->h_slots.data(), n_slots * sizeof(Slot), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = staged_h2d(ctx, b->d_arena, off, fill);
  if (e == hipSuccess && n_slots) {
    // content validation and the cardinalities in one pass over what has just landed (arrays: order; runs: order and n;
    // bitmaps: n — the caller's n is not trusted: every `n == 65536` shortcut and every buffer sized from a cardinality
    // depends on it, bitmapRepair roaring.go:4193-4206)
    const uint32_t blocks = uint32_t((n_slots + 3) / 4);
    hipLaunchKernelGGL(fbk::k_validate_recount, dim3(blocks), dim3(256), 0, ctx->stream, b->d_slots, b->d_arena, n_slots, dbad.as<uint32_t>());
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(b->h_slots.data(), b->d_slots, n_slots * sizeof(Slot), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, dbad.p, 4, hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess || h_bad) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(ctx->stream);
    if (b->d_arena) (void)ctx_free(b->ctx, b->d_arena);
    if (b->d_slots) (void)ctx_free(b->ctx, b->d_slots);
    delete b;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FBK_E_NOMEM : FBK_E_HIP, std::string("batch upload: ") + hipGetErrorString(e));
    return fail(FBK_E_INVALID, (h_bad & 1u) ? "array container not strictly ascending" : "run container intervals overlap or are unordered");
  }
  need_recount = n_slots != 0;
  if (need_recount) {
    // containers that turned out empty become nil on both sides
    bool changed = false;
    for (uint64_t s = 0; s < n_slots; ++s) {
      Slot& hs = b->h_slots[s];
      if (fbk::slot_type(hs) != fbk::kTypeNil && fbk::slot_n(hs) == 0) {
        hs = Slot{0, 0, 0};
        b->dense = false;
        changed = true;
      }
    }
    if (changed) {
      e = hipMemcpy(b->d_slots, b->h_slots.data(), n_slots * sizeof(Slot), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        (void)ctx_free(b->ctx, b->d_arena);
        (void)ctx_free(b->ctx, b->d_slots);
        delete b;
        return fail(FBK_E_HIP, std::string("batch upload: ") + hipGetErrorString(e));
      }
    }
  }
  *out_batch = b;
  return FBK_OK;
}

int32_t fbk_batch_upload_dense(fbk_ctx* ctx, const uint64_t* words, uint32_t n_rows, fbk_batch** out_batch) try {
  FBK_ENTER(ctx);
  if (!ctx || !out_batch || (n_rows && !words)) return fail(FBK_E_INVALID, "NULL argument");
  *out_batch = nullptr;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  const uint64_t n_slots = uint64_t(n_rows) * fbk::kSlots;
  const uint64_t bytes = n_slots * 8192ull;
  fbk_batch* b = new (std::nothrow) fbk_batch();
  if (!b) return fail(FBK_E_NOMEM, "host allocation failed");
  b->ctx = ctx;
  b->n_rows = n_rows;
  b->arena_bytes = bytes;
  b->dense = n_rows > 0;
  b->ring_regular = true;
  b->h_slots.resize(n_slots);
  b->h_keys.resize(n_slots);
  for (uint64_t s = 0; s < n_slots; ++s) {
    b->h_slots[s] = Slot{s * 8192ull, FBK_BITMAP_WORDS, fbk::make_tn(fbk::kTypeBitmap, 0)};
    b->h_keys[s] = s;
  }
  hipError_t e = ctx_malloc(ctx, reinterpret_cast<void**>(&b->d_arena), std::max<uint64_t>(bytes, 16));
  if (e == hipSuccess) e = ctx_malloc(ctx, reinterpret_cast<void**>(&b->d_slots), std::max<uint64_t>(n_slots, 1) * sizeof(Slot));
  if (e == hipSuccess && bytes) {
    hipPointerAttribute_t at;
    const bool on_device = hipPointerGetAttributes(&at, words) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // (an unregistered host pointer is reported as an error by some runtimes)
    if (on_device) {
      e = hipMemcpyAsync(b->d_arena, words, bytes, hipMemcpyDeviceToDevice, ctx->stream);
    } else {
      const uint8_t* src8 = reinterpret_cast<const uint8_t*>(words);
      e = staged_h2d(ctx, b->d_arena, bytes, [src8](uint64_t a, uint64_t len, uint8_t* dst) { std::memcpy(dst, src8 + a, len); });
    }
  }
  if (e == hipSuccess && n_slots) {
    e = hipMemcpyAsync(b->d_slots, b->h_slots.data(), n_slots * sizeof(Slot), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(fbk::k_recount, dim3(uint32_t((n_slots + 3) / 4)), dim3(256), 0, ctx->stream, b->d_slots,
                         b->d_arena, n_slots);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(b->h_slots.data(), b->d_slots, n_slots * sizeof(Slot), hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(ctx->stream);  // (copies out of the pinned buffers / into the arena may still be in flight)
    if (b->d_arena) (void)ctx_free(b->ctx, b->d_arena);
    if (b->d_slots) (void)ctx_free(b->ctx, b->d_slots);
    delete b;
    return fail(e == hipErrorOutOfMemory ? FBK_E_NOMEM : FBK_E_HIP, std::string("dense upload: ") + hipGetErrorString(e));
  }
  // NOTE: an all-zero bitmap keeps type bitmap with n == 0 here (the batch stays
  // `dense`); every kernel treats n == 0 as empty, and download skips it.
  *out_batch = b;
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_batch_info(fbk_ctx* ctx, const fbk_batch* batch, uint32_t* n_rows, uint64_t* n_containers,
                       uint64_t* payload_bytes) try {
  FBK_ENTER(ctx);
  if (!batch) return fail(FBK_E_INVALID, "batch is NULL");
  fbk_batch* b = const_cast<fbk_batch*>(batch);
  if (!ctx) ctx = b->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  if (int32_t rc = refresh_slots(b)) return rc;
  uint64_t nc = 0, pb = 0;
  for (const Slot& s : b->h_slots) {
    const uint32_t t = fbk::slot_type(s);
    if (t == fbk::kTypeNil || fbk::slot_n(s) == 0) continue;
    ++nc;
    pb += t == fbk::kTypeArray ? uint64_t(s.len) * 2 : t == fbk::kTypeRun ? uint64_t(s.len) * 4 : 8192ull;
  }
  if (n_rows) *n_rows = b->n_rows;
  if (n_containers) *n_containers = nc;
  if (payload_bytes) *payload_bytes = pb;
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_batch_download(fbk_ctx* ctx, const fbk_batch* batch, fbk_container_desc* descs_out, uint64_t descs_cap,
                           void* payload_out, uint64_t payload_cap) try {
  FBK_ENTER(ctx);
  if (!batch) return fail(FBK_E_INVALID, "batch is NULL");
  fbk_batch* b = const_cast<fbk_batch*>(batch);
  if (!ctx) ctx = b->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  if (int32_t rc = refresh_slots(b)) return rc;
  // Descriptors come from the host copy of the slot table; the payloads are packed on the device
  // (k_wire_copy: one wavefront per container, arena -> a contiguous image in download order) and
  // cross the bus in ONE device-to-host copy.  (One hipMemcpyAsync per container — 16 384 of them
  // for the result of a 1024-shard set-op — was all launch overhead.)
  uint64_t nc = 0, pb = 0;
  std::vector<fbk::WireDesc> wd;
  for (uint64_t s = 0; s < b->h_slots.size(); ++s) {
    const Slot& hs = b->h_slots[s];
    const uint32_t t = fbk::slot_type(hs);
    if (t == fbk::kTypeNil || fbk::slot_n(hs) == 0) continue;
    const uint64_t bytes = t == fbk::kTypeArray ? uint64_t(hs.len) * 2 : t == fbk::kTypeRun ? uint64_t(hs.len) * 4 : 8192ull;
    if (nc >= descs_cap || pb + bytes > payload_cap || !descs_out || !payload_out)
      return fail(FBK_E_CAPACITY, "download buffers too small (use fbk_batch_info)");
    fbk_container_desc& d = descs_out[nc];
    std::memset(&d, 0, sizeof(d));
    d.key = b->h_keys[s];
    d.off = pb;
    d.row = uint32_t(s / fbk::kSlots);
    d.len = hs.len;
    d.n = int32_t(fbk::slot_n(hs));
    d.type = uint8_t(t);
    wd.push_back(fbk::WireDesc{hs.off, pb, uint32_t(bytes), 0u});
    ++nc;
    pb += bytes;
  }
  if (nc == 0) return FBK_OK;
  DevBuf dpack, ddesc;
  HIP_TRY(dpack.alloc(ctx, std::max<uint64_t>(pb, 16)));
  HIP_TRY(ddesc.alloc(ctx, nc * sizeof(fbk::WireDesc)));
  HIP_TRY(hipMemcpyAsync(ddesc.p, wd.data(), nc * sizeof(fbk::WireDesc), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(fbk::k_wire_copy, dim3(uint32_t((nc + 3) / 4)), dim3(256), 0, ctx->stream, b->d_arena, dpack.as<uint8_t>(),
                     ddesc.as<fbk::WireDesc>(), nc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(payload_out, dpack.p, pb, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // (also keeps `wd` alive until the descriptor copy has been made)
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_count(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, uint64_t n, uint64_t* out_counts) try {
  FBK_ENTER(ctx);
  if (!batch || (n && (!rows || !out_counts))) return fail(FBK_E_INVALID, "NULL argument");
  fbk_batch* b = const_cast<fbk_batch*>(batch);
  if (!ctx) ctx = b->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  if (int32_t rc = refresh_slots(b)) return rc;
  // Container.N is a stored field (container_stash.go:430-440); Bitmap.Count sums it
  // (containers_slice.go:120-126).  The device produced every n (k_recount / fused in
  // the set-op kernels), so Count itself is the same 16 adds the reference does.
  for (uint64_t i = 0; i < n; ++i) {
    if (rows[i] >= b->n_rows) return fail(FBK_E_INVALID, "row index out of range");
    uint64_t c = 0;
    for (int s = 0; s < fbk::kSlots; ++s) c += fbk::slot_n(b->h_slots[uint64_t(rows[i]) * fbk::kSlots + s]);
    out_counts[i] = c;
  }
  return FBK_OK;
} FBK_ABI_CATCH(ctx)


int32_t fbk_count_range(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, uint64_t n, uint64_t start,
                        uint64_t end, uint64_t* out_counts) try {
  FBK_ENTER(ctx);
  if (!batch || (n && (!rows || !out_counts))) return fail(FBK_E_INVALID, "NULL argument");
  const uint64_t width = uint64_t(fbk::kSlots) << 16;
  if (start > end || end > width) return fail(FBK_E_INVALID, "count_range: need 0 <= start <= end <= 2^20");
  if (n > (1ull << 27)) return fail(FBK_E_INVALID, "too many rows in one call");
  if (n == 0) return FBK_OK;
  fbk_batch* b = const_cast<fbk_batch*>(batch);
  if (!ctx) ctx = b->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  DevBuf drows, dcnt;
  if (int32_t rc = upload_rows(ctx, rows, n, b->n_rows, drows)) return rc;
  HIP_TRY(dcnt.alloc(ctx, n * 8));
  hipLaunchKernelGGL(fbk::k_count_range, dim3(uint32_t((n + 3) / 4)), dim3(256), 0, ctx->stream, b->d_slots,
                     b->d_arena, drows.as<uint32_t>(), n, uint32_t(start), uint32_t(end), dcnt.as<u64>(),
                     uint32_t(ctx->opt.count_range_reference_quirk));
  HIP_TRY(hipGetLastError());
  D2H back(ctx);
  HIP_TRY(back.add(out_counts, dcnt.p, n * 8));
  HIP_TRY(back.finish());
  return FBK_OK;
} FBK_ABI_CATCH(ctx)


// ---- plans ---------------------------------------------------------------------------
// A plan is a prepared list of row pairs (A.rows_a[i], B.rows_b[i]) whose index arrays and
// output buffers live on the device, so that the hot path is launch-only: no host
// allocation, no H2D copy, no synchronisation between steps.  It is the unit one
// (query, node) batch call from mapperLocal (executor.go:6742) becomes.

}  // extern "C"

struct fbk_plan {
  fbk_ctx* ctx = nullptr;
  const fbk_batch* a = nullptr;
  const fbk_batch* b = nullptr;
  uint64_t n_pairs = 0;
  uint32_t* d_rows_a = nullptr;
  uint32_t* d_rows_b = nullptr;
  u64* d_counts = nullptr;     // n_pairs (owned unless ext_counts)
  u64* d_total = nullptr;      // 1
  uint32_t* d_done = nullptr;  // ticket counter of the fused count + total kernel (kept at 0 between launches)
  bool ext_counts = false;
  fbk_batch* out = nullptr;    // lazily created by the first set-op enqueue
  uint32_t* d_runs = nullptr;  // per output slot run count (optimize pass)
  std::vector<uint32_t> h_rows_a, h_rows_b;
  // k_icount2: the (pair, slot) item records resolved once per plan (re-resolved when a batch's descriptors were
  // rewritten since) and one count per wave, summed per pair by k_sum_wave_counts
  Slot* d_items = nullptr;
  uint32_t* d_wave_counts = nullptr;
  uint32_t* d_ring_ctl = nullptr;  // k_icount3: word 0 is OR-ed with 1 by a block that gave up on a wait (never in a correct run; fbk_plan_read reports it)
  uint64_t items_va = ~0ull, items_vb = ~0ull;
};

namespace {

int32_t optimize_cells(fbk_ctx* ctx, fbk_batch* o, const uint32_t* d_runs);  // fbk_query_api.inc
int32_t compact_cells(fbk_ctx* ctx, fbk_batch* o);                            // fbk_query_api.inc

// Average encoded payload per container of a batch, in bytes.
uint64_t batch_avg_payload(const fbk_batch* b) {
  const uint64_t slots = uint64_t(b->n_rows) * fbk::kSlots;
  return slots ? b->arena_bytes / slots : 0;
}
// Which generation of the pair kernels a launch uses (option pair_kernels pins it: 1 / 2; 0 decides by the rows).
// Rows of tiny containers on BOTH sides (arrays of a few values, a handful of runs) are served better by the round-2
// kernels: there the launch of a block per item is what a kernel costs, not decode work or latency (BenchmarkCtOps
// matrix, profiles/ctops_r03.txt: Ary16 x Ary16 10.5 us with k_icount, 18 us with k_icount2).
bool use_pair_kernels2(const fbk_ctx* ctx, const fbk_batch* a, const fbk_batch* b, int op /* -1: count */) {
  if (ctx->opt.pair_kernels) return ctx->opt.pair_kernels >= 2;
  const uint64_t lo = std::min(batch_avg_payload(a), batch_avg_payload(b)), both = (a->arena_bytes + b->arena_bytes);
  const uint64_t slots = (uint64_t(a->n_rows) + b->n_rows) * fbk::kSlots;
  if (op >= 0) {
    // materialising operations (8 KiB written per pair whatever the operands): k_setop2 wins where at least one side
    // holds KiB-sized SPARSE containers — arrays from ~1000 values, long run lists — whose decode and load latency it
    // was built around (Ary4096 x Ary1 XOR 73 -> 53 us; config 3's rows 89 -> 69 us); with small arrays on one side and
    // small arrays or bitmaps on the other the round-2 kernel's cheaper per-item path is 10-25 % ahead
    // (BenchmarkCtOps matrix, profiles/ctops_r03.txt)
    const uint64_t big_sparse = std::max(a->dense ? 0 : batch_avg_payload(a), b->dense ? 0 : batch_avg_payload(b));
    return big_sparse >= 1536;
  }
  if (!slots || both < 256 * slots) return false;
  // one side tiny: when the other one is all bitmaps (or there is nothing on one side at all) the items are probes of a
  // few dwords in global memory in either generation, and the round-2 kernel's four-wave blocks launch faster
  // (Ary1 x BM512 7.4 us vs 9.3); against big arrays / run lists the table + probe form wins (Ary4096 x Ary1 48 -> 27 us)
  if (lo < 256 && (a->arena_bytes == 0 || b->arena_bytes == 0 || (batch_avg_payload(a) < 256 ? b->dense : a->dense))) return false;
  return true;
}
// Waves per block of the round-3 pair kernels (option pair_wpb pins it).  One-wave blocks release a wave's LDS table the
// moment IT ends, which is what heterogeneous items (runs next to arrays) need; when one side's containers are tiny the
// items are all alike and short, and four waves per block quarter the number of blocks to launch.
int pair_wpb_for(const fbk_ctx* ctx, const fbk_batch* a, const fbk_batch* b) {
  if (ctx->opt.pair_wpb) return ctx->opt.pair_wpb >= 4 ? 4 : 1;  // (normalised once: 1 or 4.  Two-wave blocks were built and measured in round 4: count 182 against 166-173 us, set-ops equal — profiles/r04_pairs_wpb_ab.json — and removed)
  return std::min(batch_avg_payload(a), batch_avg_payload(b)) < 256 ? 4 : 1;
}

// The plan's item records: {A's descriptor, B's descriptor} per (pair, slot), resolved on the device once per version of the
// two batches (k_resolve_items) — the pair kernels then start with ONE scalar round trip instead of row index -> descriptor.
// Allocated with the per-wave count vector of the count kernel: both buffers or neither (a plan that got only the first, out
// of memory on the second, must not take the resolved path next time).
int32_t plan_resolve_items(fbk_ctx* ctx, fbk_plan* p) {
  const uint64_t n_items = p->n_pairs * fbk::kSlots;
  if (!p->d_items || !p->d_wave_counts) {
    Slot* di = nullptr;
    uint32_t* dw = nullptr;
    HIP_TRY(ctx_malloc(ctx, reinterpret_cast<void**>(&di), std::max<uint64_t>(n_items, 1) * 2 * sizeof(Slot)));
    if (hipError_t e = ctx_malloc(ctx, reinterpret_cast<void**>(&dw), std::max<uint64_t>(n_items, 1) * sizeof(uint32_t)); e != hipSuccess) {
      ctx_free(ctx, di);
      HIP_TRY(e);
    }
    p->d_items = di;
    p->d_wave_counts = dw;
    p->items_va = p->items_vb = ~0ull;
  }
  if (n_items && (p->items_va != p->a->version || p->items_vb != p->b->version)) {
    hipLaunchKernelGGL(fbk::k_resolve_items, dim3(uint32_t((n_items + 255) / 256)), dim3(256), 0, ctx->stream, p->a->d_slots, p->d_rows_a, p->b->d_slots,
                       p->d_rows_b, p->n_pairs, p->d_items);
    HIP_TRY(hipGetLastError());
    p->items_va = p->a->version;
    p->items_vb = p->b->version;
  }
  return FBK_OK;
}

template <int OP>
void launch_setop(bool dense, fbk_plan* p, hipStream_t st, bool want_runs, const Slot* items) {
  const uint32_t blocks = uint32_t(p->n_pairs * fbk::kSlots / 4);
  // (the in-kernel optimize() — mode 2 — only when the caller asked for optimize(): plain set-ops keep their bitmap cells)
  const uint32_t direct = want_runs ? uint32_t(p->ctx->opt.setop_direct_encode) : 0u;
#ifdef FBK_EXPERIMENTS  // (option pair_ablate, timing experiments on k_setop2: item classes skipped, emission without its stores — WRONG results)
  const uint32_t direct2 = direct | 0x100u | (uint32_t(p->ctx->opt.pair_ablate) << 16);
#else
  const uint32_t direct2 = direct | 0x100u;  // k_setop2 only: Intersect / Difference whose result is a subset of an array operand by table + probe (an A/B option until round 5)
#endif
  if (dense)
    hipLaunchKernelGGL(fbk::k_setop_dense<OP>, dim3(blocks), dim3(256), 0, st, p->a->d_arena, p->d_rows_a,
                       p->b->d_arena, p->d_rows_b, p->out->d_arena, p->out->d_slots);
  else if (use_pair_kernels2(p->ctx, p->a, p->b, OP == 0 ? FBK_OP_AND : OP == 1 ? FBK_OP_OR : OP == 2 ? FBK_OP_XOR : FBK_OP_ANDNOT) && pair_wpb_for(p->ctx, p->a, p->b) == 4)
    hipLaunchKernelGGL((fbk::k_setop2<OP, 4>), dim3(blocks), dim3(256), 0, st, p->a->d_slots, p->a->d_arena, p->d_rows_a,
                       p->b->d_slots, p->b->d_arena, p->d_rows_b, p->n_pairs, p->out->d_arena, p->out->d_slots,
                       want_runs ? p->d_runs : nullptr, direct2, (const Slot*)nullptr);
  else if (use_pair_kernels2(p->ctx, p->a, p->b, OP == 0 ? FBK_OP_AND : OP == 1 ? FBK_OP_OR : OP == 2 ? FBK_OP_XOR : FBK_OP_ANDNOT))
  {
    // Intersect / Difference with optimize(): most results come from the probe paths — the register-lean instance (18 instead of 16 waves per CU;
    // setop_direct_encode = 0 / 1 run the common instance + the separate re-encode pass: the byte-for-byte cross-check).  Union / Xor with
    // optimize() send every item down the general path: there the lean instance loses 3 % (509 -> 524 us, profiles/r06_setop2_occupancy.txt)
    constexpr bool kHasProbe = OP == 0 || OP == 3;
    if (kHasProbe && direct == 2u)
      hipLaunchKernelGGL((fbk::k_setop2<OP, 1, kHasProbe>), dim3(uint32_t(p->n_pairs * fbk::kSlots)), dim3(64), 0, st, p->a->d_slots, p->a->d_arena, p->d_rows_a,
                         p->b->d_slots, p->b->d_arena, p->d_rows_b, p->n_pairs, p->out->d_arena, p->out->d_slots,
                         want_runs ? p->d_runs : nullptr, direct2, items);
    else
      hipLaunchKernelGGL((fbk::k_setop2<OP, 1>), dim3(uint32_t(p->n_pairs * fbk::kSlots)), dim3(64), 0, st, p->a->d_slots, p->a->d_arena, p->d_rows_a,
                         p->b->d_slots, p->b->d_arena, p->d_rows_b, p->n_pairs, p->out->d_arena, p->out->d_slots,
                         want_runs ? p->d_runs : nullptr, direct2, items);
  }
  else
    hipLaunchKernelGGL(fbk::k_setop<OP>, dim3(blocks), dim3(256), 0, st, p->a->d_slots, p->a->d_arena, p->d_rows_a,
                       p->b->d_slots, p->b->d_arena, p->d_rows_b, p->n_pairs, p->out->d_arena, p->out->d_slots,
                       want_runs ? p->d_runs : nullptr, direct);
}

void free_batch_storage(fbk_batch* b) {
  if (!b) return;
  if (b->d_arena) (void)ctx_free(b->ctx, b->d_arena);
  if (b->d_slots) (void)ctx_free(b->ctx, b->d_slots);
  if (b->d_win) (void)ctx_free(b->ctx, b->d_win);
  if (b->d_shadow_arena) (void)ctx_free(b->ctx, b->d_shadow_arena);
  if (b->d_shadow_slots) (void)ctx_free(b->ctx, b->d_shadow_slots);
  delete b;
}

void free_plan_storage(fbk_plan* p) {
  if (!p) return;
  if (p->d_rows_a) (void)ctx_free(p->ctx, p->d_rows_a);
  if (p->d_rows_b) (void)ctx_free(p->ctx, p->d_rows_b);
  if (p->d_counts && !p->ext_counts) (void)ctx_free(p->ctx, p->d_counts);
  if (p->d_total) (void)ctx_free(p->ctx, p->d_total);
  if (p->d_done) (void)ctx_free(p->ctx, p->d_done);
  if (p->d_runs) (void)ctx_free(p->ctx, p->d_runs);
  if (p->d_items) (void)ctx_free(p->ctx, p->d_items);
  if (p->d_wave_counts) (void)ctx_free(p->ctx, p->d_wave_counts);
  if (p->d_ring_ctl) (void)ctx_free(p->ctx, p->d_ring_ctl);
  free_batch_storage(p->out);
  delete p;
}

int32_t plan_create_locked(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, const fbk_batch* b,
                           const uint32_t* rows_b, uint64_t n_pairs, void* ext_counts, fbk_plan** out_plan) {
  if (n_pairs > (1ull << 27)) return fail(FBK_E_INVALID, "too many pairs in one plan");
  for (uint64_t i = 0; i < n_pairs; ++i)
    if (rows_a[i] >= a->n_rows || rows_b[i] >= b->n_rows) return fail(FBK_E_INVALID, "row index out of range");
  fbk_plan* p = new (std::nothrow) fbk_plan();
  if (!p) return fail(FBK_E_NOMEM, "host allocation failed");
  p->ctx = ctx;
  p->a = a;
  p->b = b;
  p->n_pairs = n_pairs;
  p->h_rows_a.assign(rows_a, rows_a + n_pairs);
  p->h_rows_b.assign(rows_b, rows_b + n_pairs);
  const uint64_t rb = std::max<uint64_t>(n_pairs, 1) * sizeof(uint32_t);
  hipError_t e = ctx_malloc(ctx, reinterpret_cast<void**>(&p->d_rows_a), rb);
  if (e == hipSuccess) e = ctx_malloc(ctx, reinterpret_cast<void**>(&p->d_rows_b), rb);
  if (e == hipSuccess) e = ctx_malloc(ctx, reinterpret_cast<void**>(&p->d_total), sizeof(u64));
  if (e == hipSuccess) e = ctx_malloc(ctx, reinterpret_cast<void**>(&p->d_done), sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemsetAsync(p->d_done, 0, sizeof(uint32_t), ctx->stream);
  if (e == hipSuccess) {
    if (ext_counts) {
      p->d_counts = static_cast<u64*>(ext_counts);
      p->ext_counts = true;
    } else {
      e = ctx_malloc(ctx, reinterpret_cast<void**>(&p->d_counts), std::max<uint64_t>(n_pairs, 1) * sizeof(u64));
    }
  }
  if (e == hipSuccess && n_pairs) {
    e = hipMemcpyAsync(p->d_rows_a, rows_a, n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->d_rows_b, rows_b, n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    free_plan_storage(p);
    return fail(e == hipErrorOutOfMemory ? FBK_E_NOMEM : FBK_E_HIP, std::string("plan: ") + hipGetErrorString(e));
  }
  *out_plan = p;
  return FBK_OK;
}

int32_t plan_icount_enqueue_locked(fbk_ctx* ctx, fbk_plan* p, u64* fused_total = nullptr, u64* accum = nullptr) {
  if (p->n_pairs == 0) {
    if (fused_total) HIP_TRY(hipMemsetAsync(fused_total, 0, sizeof(u64), ctx->stream));
    return FBK_OK;
  }
  const uint32_t np = uint32_t(p->n_pairs);
  if (p->a->dense && p->b->dense) {
    // all-bitmap rows: pure streaming kernel.  slots-per-block 16 = one block per row
    // pair, plain store; smaller groups = more blocks + one atomicAdd per block.
    const int spb = int(ctx->opt.dense_spb);
    if (spb != 16) HIP_TRY(hipMemsetAsync(p->d_counts, 0, p->n_pairs * sizeof(u64), ctx->stream));
#define FBK_LAUNCH_DENSE(S)                                                                                       \
  hipLaunchKernelGGL(fbk::k_icount_dense<S>, dim3(np*(16 / S)), dim3(256), 0, ctx->stream, p->a->d_arena,         \
                     p->d_rows_a, p->b->d_arena, p->d_rows_b, p->d_counts, fused_total, p->d_done, np, accum)
    switch (spb) {
      case 1: FBK_LAUNCH_DENSE(1); break;
      case 2: FBK_LAUNCH_DENSE(2); break;
      case 4: FBK_LAUNCH_DENSE(4); break;
      case 8: FBK_LAUNCH_DENSE(8); break;
      default: FBK_LAUNCH_DENSE(16); break;
    }
#undef FBK_LAUNCH_DENSE
  } else {
    const bool pk2 = use_pair_kernels2(ctx, p->a, p->b, -1);
#ifdef FBK_EXPERIMENTS
    const bool pk3 = ctx->opt.pair_kernels == 3 && p->a->ring_regular && p->b->ring_regular;
#else
    constexpr bool pk3 = false;
#endif
    // (resolved item records + a count per wave pay for their second launch only where the items are heavy: one-wave blocks)
    const bool resolved = pk3 || (pk2 && pair_wpb_for(ctx, p->a, p->b) == 1);
    if (!resolved) HIP_TRY(hipMemsetAsync(p->d_counts, 0, p->n_pairs * sizeof(u64), ctx->stream));
    if (resolved)
      if (int32_t rc = plan_resolve_items(ctx, p)) return rc;
#ifdef FBK_EXPERIMENTS
    if (pk3) {
      // the persistent loader / decoder kernel: a block per compute unit (or two), every block walks its share of the item records
      if (!p->d_ring_ctl) {
        HIP_TRY(ctx_malloc(ctx, reinterpret_cast<void**>(&p->d_ring_ctl), 32));
        HIP_TRY(hipMemsetAsync(p->d_ring_ctl, 0, 32, ctx->stream));
      }
      const uint64_t n_items = p->n_pairs * fbk::kSlots, n_chunks = (n_items + fbk::kRgChunk - 1) / fbk::kRgChunk;
      const uint32_t cus = uint32_t(ctx->n_cu > 0 ? ctx->n_cu : 256);
#define FBK_LAUNCH_ICOUNT3(D, R, NT, PER_CU)                                                                                                   \
  hipLaunchKernelGGL((fbk::k_icount3<D, R, NT>), dim3(uint32_t(std::min<uint64_t>(n_chunks, uint64_t(cus) * PER_CU))), dim3(64 * (D + 1)), 0, \
                     ctx->stream, p->d_items, p->a->d_arena, p->b->d_arena, n_items, p->d_wave_counts, p->d_ring_ctl, d_dbg, uint32_t(ctx->opt.ring_flags))
      const bool nt = ctx->opt.ring_nt != 0;
      const uint32_t per_cu = ctx->opt.ring_geom == 3 ? 2u : 1u;
      const uint32_t dbg_blocks = uint32_t(std::min<uint64_t>(n_chunks, uint64_t(cus) * per_cu));
      DevBuf dbgbuf;
      uint32_t* d_dbg = nullptr;
      if (ctx->opt.ring_debug) {
        HIP_TRY(dbgbuf.alloc(ctx, uint64_t(dbg_blocks) * 32 * 4));
        HIP_TRY(hipMemsetAsync(dbgbuf.p, 0, uint64_t(dbg_blocks) * 32 * 4, ctx->stream));
        d_dbg = dbgbuf.as<uint32_t>();
      }
      switch (ctx->opt.ring_geom) {
        case 1: if (nt) FBK_LAUNCH_ICOUNT3(8, 65536, true, 1); else FBK_LAUNCH_ICOUNT3(8, 65536, false, 1); break;
        case 2: if (nt) FBK_LAUNCH_ICOUNT3(6, 65536, true, 1); else FBK_LAUNCH_ICOUNT3(6, 65536, false, 1); break;
        case 3: if (nt) FBK_LAUNCH_ICOUNT3(5, 32768, true, 2); else FBK_LAUNCH_ICOUNT3(5, 32768, false, 2); break;
        default: if (nt) FBK_LAUNCH_ICOUNT3(10, 65536, true, 1); else FBK_LAUNCH_ICOUNT3(10, 65536, false, 1); break;
      }
#undef FBK_LAUNCH_ICOUNT3
      if (d_dbg) {
        std::vector<uint32_t> h(uint64_t(dbg_blocks) * 32);
        HIP_TRY(hipMemcpyAsync(h.data(), d_dbg, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        double sum[32] = {0};
        uint32_t tmax = 0, tmin = ~0u;
        for (uint32_t b = 0; b < dbg_blocks; ++b) {
          for (int k = 0; k < 32; ++k) sum[k] += h[uint64_t(b) * 32 + k];
          tmax = std::max(tmax, h[uint64_t(b) * 32]);
          tmin = std::min(tmin, h[uint64_t(b) * 32]);
        }
        std::fprintf(stderr, "[ring_debug] blocks %u planner cycles avg %.0f min %u max %u | wait records %.0f slots %.0f | reclaim polls %.0f entries %.0f | decoders (summed per block): wait entry %.0f issue %.0f wait payload %.0f decode %.0f items %.0f deferred %.0f\n",
                     dbg_blocks, sum[0] / dbg_blocks, tmin, tmax, sum[1] / dbg_blocks, sum[2] / dbg_blocks, sum[6] / dbg_blocks, sum[7] / dbg_blocks, sum[8] / dbg_blocks,
                     sum[12] / dbg_blocks, sum[11] / dbg_blocks, sum[9] / dbg_blocks, sum[10] / dbg_blocks, sum[13] / dbg_blocks);
        if (sum[21] > 0 && sum[26] > 0)
          std::fprintf(stderr, "[ring_debug]   array x array phases, cycles per item: payload -> registers %.0f, take next %.0f, scatter %.0f, probe + reduce %.0f (dwords per item: table side %.0f, probing side %.0f)\n",
                       sum[26] / sum[21], sum[27] / sum[21], sum[28] / sum[21], sum[29] / sum[21], sum[30] / sum[21], sum[31] / sum[21]);
        static const char* kCls[5] = {"array x array", "array x bitmap", "array x run (<= 600)", "general (run x run, run x bitmap, long runs)", "outside the ring / bitmap x bitmap"};
        for (int k = 0; k < 5; ++k)
          if (sum[21 + k] > 0) std::fprintf(stderr, "[ring_debug]   %-48s items %8.0f  decode cycles per item %7.0f\n", kCls[k], sum[21 + k], sum[16 + k] / sum[21 + k]);
      }
      hipLaunchKernelGGL(fbk::k_sum_wave_counts, dim3(uint32_t((p->n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, p->d_wave_counts,
                         uint32_t(fbk::kSlots), p->n_pairs, p->d_counts, p->d_ring_ctl + 1);
    } else
#endif
    if (pk2) {
#define FBK_LAUNCH_ICOUNT2(S, W)                                                                                                   \
  hipLaunchKernelGGL((fbk::k_icount2<S, W>), dim3(uint32_t((p->n_pairs * (fbk::kSlots / S) + W - 1) / W)), dim3(64 * W), 0, ctx->stream, \
                     p->a->d_slots, p->a->d_arena, p->d_rows_a, p->b->d_slots, p->b->d_arena, p->d_rows_b, p->n_pairs,            \
                     p->d_counts, pair_flags, resolved ? p->d_items : (const Slot*)nullptr,  \
                     resolved ? p->d_wave_counts : (uint32_t*)nullptr)
      // One container slot per wave.  (The kernel is written for SPW slots per wave with the next slot's payload in flight while the
      // current one is decoded; SPW = 2 / 4 measured 49.6 / 56 us against 46 on 2048 row pairs in round 3 and 172 / 184 us against 157-162
      // on 8192 in round 5 — profiles/r05_pairs_spw_8192_pairs.json: the same 6-10 % / 17-24 % behind at both sizes, so not a tail effect;
      // a wave that decodes two items one after the other holds its table twice as long, and the second payload, requested before the
      // first item's own later loads, has to land before those can be waited for (the vector memory counter is in order).  Instantiated in
      // the experiments build only, option pair_spw there.)
      const int wpb = pair_wpb_for(ctx, p->a, p->b);
#ifdef FBK_EXPERIMENTS
      const uint32_t pair_flags = 3u | (uint32_t(ctx->opt.pair_ablate & 255) << 8) | (uint32_t(ctx->opt.pair_stamp) << 16);
#else
      constexpr uint32_t pair_flags = 3u;  // bit 0: the small-array / probe paths, bit 1: array x run items probe the run container's table (both were A/B options until round 5)
#endif
      uint32_t spw = 1;
      if (wpb == 4) FBK_LAUNCH_ICOUNT2(1, 4);
#ifdef FBK_EXPERIMENTS
      else if (ctx->opt.pair_spw == 4) { spw = 4; FBK_LAUNCH_ICOUNT2(4, 1); }
      else if (ctx->opt.pair_spw == 2) { spw = 2; FBK_LAUNCH_ICOUNT2(2, 1); }
#endif
      else FBK_LAUNCH_ICOUNT2(1, 1);
#undef FBK_LAUNCH_ICOUNT2
      if (resolved)
        hipLaunchKernelGGL(fbk::k_sum_wave_counts, dim3(uint32_t((p->n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, p->d_wave_counts,
                           uint32_t(fbk::kSlots) / spw, p->n_pairs, p->d_counts, (uint32_t*)nullptr);
    }
    else
      hipLaunchKernelGGL(fbk::k_icount, dim3(np * (fbk::kSlots / 4)), dim3(256), 0, ctx->stream, p->a->d_slots,
                         p->a->d_arena, p->d_rows_a, p->b->d_slots, p->b->d_arena, p->d_rows_b, p->n_pairs, p->d_counts,
                         1u);
    if (fused_total)
      hipLaunchKernelGGL(fbk::k_sum_u64, dim3(1), dim3(256), 0, ctx->stream, p->d_counts, p->n_pairs, fused_total);
    if (accum) hipLaunchKernelGGL(fbk::k_sum_u64_add, dim3(1), dim3(256), 0, ctx->stream, p->d_counts, p->n_pairs, accum);
  }
  HIP_TRY(hipGetLastError());
  return FBK_OK;
}

int32_t plan_setop_enqueue_locked(fbk_ctx* ctx, fbk_plan* p, int32_t op, bool want_runs) {
  const uint64_t n_slots = p->n_pairs * fbk::kSlots;
  if (!p->out) {
    // the output keys are derived from the inputs' host slot tables: refresh them if an input is
    // itself the output of an asynchronous operation
    if (int32_t rc = refresh_slots(const_cast<fbk_batch*>(p->a))) return rc;
    if (int32_t rc = refresh_slots(const_cast<fbk_batch*>(p->b))) return rc;
    fbk_batch* o = new (std::nothrow) fbk_batch();
    if (!o) return fail(FBK_E_NOMEM, "host allocation failed");
    o->ctx = ctx;
    o->n_rows = uint32_t(p->n_pairs);
    o->arena_bytes = n_slots * 8192ull;
    o->ring_regular = true;  // (8 KiB cells)
    o->h_slots.assign(n_slots, Slot{0, 0, 0});
    o->h_keys.assign(n_slots, 0);
    for (uint64_t i = 0; i < p->n_pairs; ++i)
      for (int s = 0; s < fbk::kSlots; ++s) {
        const uint64_t ia = uint64_t(p->h_rows_a[i]) * fbk::kSlots + s, ib = uint64_t(p->h_rows_b[i]) * fbk::kSlots + s;
        // a nil/nil slot pair yields nil and its key is never reported
        const bool has_a = fbk::slot_type(p->a->h_slots[ia]) != fbk::kTypeNil;
        o->h_keys[i * fbk::kSlots + s] = has_a ? p->a->h_keys[ia] : p->b->h_keys[ib];
      }
    hipError_t e = ctx_malloc(ctx, reinterpret_cast<void**>(&o->d_arena), std::max<uint64_t>(o->arena_bytes, 16));
    if (e == hipSuccess) e = ctx_malloc(ctx, reinterpret_cast<void**>(&o->d_slots), std::max<uint64_t>(n_slots, 1) * sizeof(Slot));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      free_batch_storage(o);
      return fail(e == hipErrorOutOfMemory ? FBK_E_NOMEM : FBK_E_HIP, std::string("setop output: ") + hipGetErrorString(e));
    }
    p->out = o;
    o->borrowed = true;
  }
  if (want_runs && !p->d_runs) HIP_TRY(ctx_malloc(ctx, reinterpret_cast<void**>(&p->d_runs), std::max<uint64_t>(n_slots, 1) * 4));
  if (p->n_pairs == 0) return FBK_OK;
  const bool dense = p->a->dense && p->b->dense && !want_runs;
  // the one-wave-block pair kernels start from the plan's resolved item records (as the count does)
  const Slot* items = nullptr;
  if (!dense && use_pair_kernels2(ctx, p->a, p->b, op) && pair_wpb_for(ctx, p->a, p->b) == 1) {
    if (int32_t rc = plan_resolve_items(ctx, p)) return rc;
    items = p->d_items;
  }
  switch (op) {
    case FBK_OP_AND: launch_setop<0>(dense, p, ctx->stream, want_runs, items); break;
    case FBK_OP_OR: launch_setop<1>(dense, p, ctx->stream, want_runs, items); break;
    case FBK_OP_XOR: launch_setop<2>(dense, p, ctx->stream, want_runs, items); break;
    default: launch_setop<3>(dense, p, ctx->stream, want_runs, items); break;
  }
  // the pair's cardinality = the sum of the n its 16 output descriptors carry (rounds 1-3: a uint64 atomic per wave onto a
  // zeroed vector; measured equal within 1 % on config 3's 8192 row pairs, and one launch instead of memset + atomics)
  hipLaunchKernelGGL(fbk::k_sum_slot_n, dim3(uint32_t((p->n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, p->out->d_slots, p->n_pairs, p->d_counts);
  HIP_TRY(hipGetLastError());
  // dense kernels write the dense layout (an all-zero result cell stays an all-zero
  // bitmap in the arena, its slot says nil): the output can feed the dense kernels again
  p->out->dense = dense;
  slots_rewritten(p->out);
  return FBK_OK;
}

}  // namespace

extern "C" {

int32_t fbk_plan_create(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, const fbk_batch* b,
                        const uint32_t* rows_b, uint64_t n_pairs, void* device_counts_or_null, fbk_plan** out_plan) try {
  FBK_ENTER(ctx);
  if (!ctx || !a || !b || !out_plan || (n_pairs && (!rows_a || !rows_b))) return fail(FBK_E_INVALID, "NULL argument");
  *out_plan = nullptr;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  return plan_create_locked(ctx, a, rows_a, b, rows_b, n_pairs, device_counts_or_null, out_plan);
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_free(fbk_ctx* ctx, fbk_plan* plan) try {
  FBK_ENTER(ctx);
  if (!plan) return FBK_OK;
  if (!ctx) ctx = plan->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_plan_storage(plan);
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_intersection_count(fbk_ctx* ctx, fbk_plan* plan) try {
  FBK_ENTER(ctx);
  if (!ctx || !plan) return fail(FBK_E_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  return plan_icount_enqueue_locked(ctx, plan);
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_intersection_count_total(fbk_ctx* ctx, fbk_plan* plan, void* device_total_or_null) try {
  FBK_ENTER(ctx);
  if (!ctx || !plan) return fail(FBK_E_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  u64* dst = device_total_or_null ? static_cast<u64*>(device_total_or_null) : plan->d_total;
  return plan_icount_enqueue_locked(ctx, plan, dst);
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_intersection_count_accumulate(fbk_ctx* ctx, fbk_plan* plan, void* device_accum) try {
  FBK_ENTER(ctx);
  if (!ctx || !plan || !device_accum) return fail(FBK_E_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  return plan_icount_enqueue_locked(ctx, plan, nullptr, static_cast<u64*>(device_accum));
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_setop(fbk_ctx* ctx, fbk_plan* plan, int32_t op, uint32_t flags) try {
  FBK_ENTER(ctx);
  if (!ctx || !plan) return fail(FBK_E_INVALID, "NULL argument");
  if (op < 0 || op > 3) return fail(FBK_E_INVALID, "unknown set operation");
  if (flags & ~FBK_SETOP_OPTIMIZE) return fail(FBK_E_INVALID, "unknown flags");
  const bool opt = (flags & FBK_SETOP_OPTIMIZE) != 0;
  if (opt && ctx->opt.setop_direct_encode != 2)
    return fail(FBK_E_INVALID, "FBK_SETOP_OPTIMIZE on a plan needs option setop_direct_encode = 2 (optimize() inside the kernel); the separate re-encode pass sizes its output on the host: use fbk_setop");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  return plan_setop_enqueue_locked(ctx, plan, op, opt);
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_total(fbk_ctx* ctx, fbk_plan* plan, void* device_total_or_null) try {
  FBK_ENTER(ctx);
  if (!ctx || !plan) return fail(FBK_E_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  u64* dst = device_total_or_null ? static_cast<u64*>(device_total_or_null) : plan->d_total;
  hipLaunchKernelGGL(fbk::k_sum_u64, dim3(1), dim3(256), 0, ctx->stream, plan->d_counts, plan->n_pairs, dst);
  HIP_TRY(hipGetLastError());
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_read(fbk_ctx* ctx, fbk_plan* plan, uint64_t* out_counts, uint64_t* out_total) try {
  FBK_ENTER(ctx);
  if (!ctx || !plan) return fail(FBK_E_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  if (out_counts && plan->n_pairs)
    HIP_TRY(hipMemcpyAsync(out_counts, plan->d_counts, plan->n_pairs * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  if (out_total) HIP_TRY(hipMemcpyAsync(out_total, plan->d_total, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  uint32_t ring_ctl[8] = {0};
  if (plan->d_ring_ctl) HIP_TRY(hipMemcpyAsync(ring_ctl, plan->d_ring_ctl, sizeof(ring_ctl), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ring_ctl[0]) {
    char buf[256];
    std::snprintf(buf, sizeof(buf), "k_icount3: a block gave up on a wait (protocol error): the counts of this plan are not valid [where %u block %u wave %u: %u %u %u | pub %u tail %u]",
                  ring_ctl[2] & 255u, (ring_ctl[2] >> 8) & 0xFFFFu, ring_ctl[2] >> 24, ring_ctl[3], ring_ctl[4], ring_ctl[5], ring_ctl[6], ring_ctl[7]);
    return fail(FBK_E_HIP, buf);
  }
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_output(fbk_ctx* ctx, fbk_plan* plan, fbk_batch** out_batch) try {
  FBK_ENTER(ctx);
  if (!plan || !out_batch) return fail(FBK_E_INVALID, "NULL argument");
  (void)ctx;
  *out_batch = plan->out;
  if (!plan->out) return fail(FBK_E_INVALID, "plan has no set-op output yet");
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

int32_t fbk_plan_detach_output(fbk_ctx* ctx, fbk_plan* plan, fbk_batch** out_batch) try {
  FBK_ENTER(ctx);
  if (!plan || !out_batch) return fail(FBK_E_INVALID, "NULL argument");
  if (!ctx) ctx = plan->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (!plan->out) return fail(FBK_E_INVALID, "plan has no set-op output yet");
  *out_batch = plan->out;
  plan->out = nullptr;
  return FBK_OK;
} FBK_ABI_CATCH(ctx)

// ---- one-shot calls (plan + enqueue + read) --------------------------------------------

int32_t fbk_intersection_count(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, const fbk_batch* b,
                               const uint32_t* rows_b, uint64_t n_pairs, uint64_t* out_counts) try {
  FBK_ENTER(ctx);
  if (!ctx || !a || !b || (n_pairs && (!rows_a || !rows_b || !out_counts))) return fail(FBK_E_INVALID, "NULL argument");
  if (n_pairs == 0) return FBK_OK;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  fbk_plan* p = nullptr;
  if (int32_t rc = plan_create_locked(ctx, a, rows_a, b, rows_b, n_pairs, nullptr, &p)) return rc;
  int32_t rc = plan_icount_enqueue_locked(ctx, p);
  if (!rc) {
    D2H back(ctx);
    hipError_t e = back.add(out_counts, p->d_counts, n_pairs * sizeof(u64));
    if (e == hipSuccess) e = back.finish();
    if (e != hipSuccess) rc = fail(FBK_E_HIP, std::string("intersection_count: ") + hipGetErrorString(e));
  }
  (void)hipStreamSynchronize(ctx->stream);
  free_plan_storage(p);
  return rc;
} FBK_ABI_CATCH(ctx)

int32_t fbk_setop(fbk_ctx* ctx, int32_t op, const fbk_batch* a, const uint32_t* rows_a, const fbk_batch* b,
                  const uint32_t* rows_b, uint64_t n_pairs, uint32_t flags, fbk_batch** out_batch,
                  uint64_t* out_counts) try {
  FBK_ENTER(ctx);
  if (!ctx || !a || !b || !out_batch || (n_pairs && (!rows_a || !rows_b))) return fail(FBK_E_INVALID, "NULL argument");
  if (op < 0 || op > 3) return fail(FBK_E_INVALID, "unknown set operation");
  if (flags & ~FBK_SETOP_OPTIMIZE) return fail(FBK_E_INVALID, "unknown flags");
  *out_batch = nullptr;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (int32_t rc = set_device(ctx)) return rc;
  fbk_plan* p = nullptr;
  if (int32_t rc = plan_create_locked(ctx, a, rows_a, b, rows_b, n_pairs, nullptr, &p)) return rc;
  const bool opt = (flags & FBK_SETOP_OPTIMIZE) != 0;
  int32_t rc = plan_setop_enqueue_locked(ctx, p, op, opt);
  if (!rc && out_counts && n_pairs) {
    hipError_t e = hipMemcpyAsync(out_counts, p->d_counts, n_pairs * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = fail(FBK_E_HIP, std::string("setop: ") + hipGetErrorString(e));
  }
  if (!rc && opt && ctx->opt.setop_direct_encode != 2) rc = optimize_cells(ctx, p->out, p->d_runs);  // (mode 2: the kernel has encoded already)
  else if (!rc && opt && ctx->opt.setop_compact) rc = compact_cells(ctx, p->out);  // ... into the head of 8 KiB cells: the caller owns this batch, it gets a right-sized arena
  if (!rc) rc = refresh_slots(p->out);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (!rc && e != hipSuccess) rc = fail(FBK_E_HIP, std::string("setop: ") + hipGetErrorString(e));
  if (!rc) {
    *out_batch = p->out;
    p->out->borrowed = false;
    p->out = nullptr;
  }
  free_plan_storage(p);
  return rc;
} FBK_ABI_CATCH(ctx)

}  // extern "C"

#include "fbk_query_api.inc"
#include "fbk_prepared_api.inc"
#include "fbk_wire_api.inc"
#include "fbk_cache_api.inc"
#include "fbk_group_api.inc"
