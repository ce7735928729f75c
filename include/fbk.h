/*
 * fbk.h — C ABI of the MI355X roaring-bitmap execution kernel ("fbk" = FeatureBase
 * Kernel).  This is the drop-in boundary for ONE hot path of FeatureBase: the
 * Intersect / Union / Difference / Xor / Count / IntersectionCount loop that the
 * reference runs container-by-container on the CPU in Go.
 *
 * The reference has no FFI for this path (CGO_ENABLED=0, Makefile:34); the narrowest
 * Go choke points each entry point replaces are cited per function below as
 * <file>:<line> relative to the FeatureBase tree.  INTEGRATION.md shows the cgo stub a
 * maintainer would add on the Go side.
 *
 * Conventions
 *   - C linkage, plain pointers and sizes, no exceptions across the boundary: every entry point is a function-try-block
 *     (a host allocation that fails inside the library comes back as FBK_E_NOMEM, anything else a C++ library could throw
 *     as FBK_E_HIP with the message; tests/test_abi.py checks that no definition is left unwrapped).
 *   - Every function returns an int32 status: FBK_OK (0) or a negative FBK_E_* code;
 *     fbk_last_error_r(ctx, ...) returns the context's human readable message (see below).
 *     (roaring set-ops never return errors in Go — invariant breaks panic,
 *     roaring/roaring.go:976,3524 — so every error here is an argument/resource error.)
 *   - The caller owns every input buffer for the duration of the call only: the
 *     library copies/uploads before returning, which matches the lifetime rule of
 *     mmapped containers ("must not retain", roaring/filter.go:179-181, tx.go:66-72).
 *   - Handles (fbk_ctx, fbk_batch) are opaque; one fbk_ctx drives ONE GPU.  A process
 *     that owns several GPUs opens a group (fbk_group_open: one member context per device,
 *     shards partitioned over the members as executor.go:6579 `mapper` partitions shards
 *     over nodes, partial counts reduced inside the library); one process per GPU with
 *     torch.distributed / RCCL above the ABI is the other supported deployment.
 *   - Thread safety: calls on one context are serialised by an internal mutex (cgo pins
 *     one OS thread per call).  Concurrent callers that should overlap on the device
 *     (~NumCPU goroutines, executor.go:6723-6737) each use their own fbk_ctx_fork.
 *   - There is NO CPU fallback: if no gfx950 device is usable, fbk_open fails.
 *
 * Data model (modelled on roaring/containers_slice.go:5-10 — sorted keys + containers —
 * flattened to SoA descriptors over one payload arena):
 *   A *batch* is a set of *rows*.  A row is what fragment.row() returns
 *   (fragment.go:283-333): the <=16 containers of one (fragment,rowID) in one shard,
 *   ShardWidth = 2^20 columns = 16 containers of 2^16 bits (shardwidth/helper.go:14,
 *   fragment.go:47).  Container `key & 15` is the slot inside the row; the library never
 *   interprets the high key bits, it only carries them through to download.
 *   Payload encodings are byte-identical to what arrayWriteTo / bitmapWriteTo /
 *   runWriteTo emit (roaring/roaring.go:4068-4108) minus the 2-byte run-count prefix:
 *     array : len x uint16 ascending
 *     bitmap: 1024 x uint64 little endian, bit v at word v/64, bit v%64
 *     run   : len x {uint16 start, uint16 last}, inclusive, ascending, non-overlapping
 */
#ifndef FBK_H
#define FBK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBK_ABI_VERSION 6
/* ABI history.
 *   6 (round 6): + fbk_comm_unique_id / fbk_comm_init / fbk_comm_all_reduce_u64 / fbk_comm_fence / fbk_comm_close (the
 *      one-process-per-GPU exchange issued by the library itself).  fbk_batch_compact refuses a batch of another context;
 *      fbk_group_topn refuses members whose topn_semantics differ.
 *   5 (round 5): + fbk_topn_partials, options topn_semantics, matrix_shadow_arena_x.  CHANGED: fbk_topn / fbk_query_topn /
 *      fbk_group_topn with 0 < n < n_a return the reference's two-pass answer by default (topn_semantics = 1; = 0 restores
 *      round 4's exact top n of fbk_topn; round 4's per-member candidate rule of fbk_group_topn is gone — it was neither);
 *      count_range_reference_quirk defaults to 1 (the reference's number).
 *   4 (round 4): + fbk_query_bsi_range / _fold / _topn / _output, fbk_group_bsi_sum / _topn.  Also changed in that round
 *      (not said at the time): option pair_spw REMOVED (fbk_set_option fails, FBK_PAIR_SPW is ignored); pair_wpb accepts
 *      0 / 1 / 4 only; the default of matrix_shadow_max_mb went from 65536 to 16384; fbk_plan_setop(FBK_SETOP_OPTIMIZE)
 *      succeeds (when setop_direct_encode == 2) where it was an error.
 *
 * Where the DEFAULT deliberately differs from the reference, and the option that restores identity:
 *   (none since ABI 5.)  Before: CountRange on runs ending at `end` (count_range_reference_quirk = 1), TopN with n > 0
 *   (topn_semantics = 1).  What remains different cannot be restored by an option because the reference itself leaves it
 *   unspecified or depends on state off the path: the order of TopN pairs of EQUAL count (Go's unstable sort over map
 *   order; here row index ascending) and the rank cache's size / staleness (cache.go: at most CacheSize rows, refreshed
 *   lazily; here every row of the field is ranked, as after RecalculateCaches with CacheSize >= the field's rows). */

/* status codes */
#define FBK_OK 0
#define FBK_E_INVALID (-1)   /* bad argument / malformed container */
#define FBK_E_NODEVICE (-2)  /* no usable gfx950 device */
#define FBK_E_HIP (-3)       /* HIP runtime error (message has hipGetErrorString) */
#define FBK_E_NOMEM (-4)     /* host or device allocation failed */
#define FBK_E_CAPACITY (-5)  /* caller-provided output buffer too small */
#define FBK_E_NOTFOUND (-6)  /* cache miss (absent or stale entry) */

/* container type codes — identical to roaring/roaring.go:53-58 */
#define FBK_TYPE_NIL 0
#define FBK_TYPE_ARRAY 1
#define FBK_TYPE_BITMAP 2
#define FBK_TYPE_RUN 3

/* set operations: intersect (roaring.go:4753), union (:4980), xor (:6052),
 * difference a\b (:5692) */
#define FBK_OP_AND 0
#define FBK_OP_OR 1
#define FBK_OP_XOR 2
#define FBK_OP_ANDNOT 3

#define FBK_SLOTS_PER_ROW 16      /* containers per shard row, fragment.go:47 */
#define FBK_BITMAP_WORDS 1024     /* bitmapN, roaring.go:44 */
#define FBK_CONTAINER_BITS 65536

/* fbk_setop flags */
#define FBK_SETOP_KEEP_BITMAP 0u  /* every non-empty output container is a bitmap */
#define FBK_SETOP_OPTIMIZE 1u     /* re-encode outputs with Container.optimize() rules
                                     (roaring.go:3412-3461): run if runs<=2048 && runs<=n/2,
                                     else array if n<4096, else bitmap */

typedef struct fbk_ctx fbk_ctx;
typedef struct fbk_batch fbk_batch;

/* One container of a batch.  Replaces *roaring.Container (container_stash.go:46-53). */
typedef struct fbk_container_desc {
  uint64_t key;  /* roaring container key; slot = key & 15 */
  uint64_t off;  /* byte offset of the payload inside `payload` */
  uint32_t row;  /* batch-local row ordinal, 0 <= row < n_rows */
  uint32_t len;  /* array: #uint16; bitmap: 1024; run: #intervals */
  int32_t n;     /* cardinality (Container.N, container_stash.go:430); -1 = recount on device */
  uint8_t type;  /* FBK_TYPE_* */
  uint8_t pad[3];
} fbk_container_desc;

/* ---- lifetime ---------------------------------------------------------------- */

/* Number of visible HIP devices. */
int32_t fbk_device_count(int32_t* out_n);

/* Open a context on HIP device `device`.  Replaces nothing in Go; called at server
 * start (server/server.go Open). */
int32_t fbk_open(int32_t device, uint32_t flags, fbk_ctx** out_ctx);
int32_t fbk_close(fbk_ctx* ctx);

/* Message of the last failing call (never NULL).
 *   ctx == NULL : the calling THREAD's last failure (the only source for failures that have no
 *                 context yet: fbk_open, fbk_device_count, fbk_rbf_find_root).
 *   ctx != NULL : the CONTEXT's last failure, copied into a buffer owned by the calling thread (valid
 *                 until that thread's next call into the library).
 * fbk_last_error_r copies the context's message (and its status code) into a caller buffer — the
 * form a cgo binding must use: a goroutine can be rescheduled onto another OS thread between the
 * failing call and the call that fetches the message, so nothing thread-local is reliable there.
 * Semantics are sqlite3_errmsg's: with several callers failing concurrently on ONE context the
 * message is that of the most recent failure; callers that need their own message use their own
 * forked context (fbk_ctx_fork). */
const char* fbk_last_error(fbk_ctx* ctx);
int32_t fbk_last_error_r(fbk_ctx* ctx, char* buf, uint64_t cap, int32_t* out_code);

/* A second context on the same device for another calling thread: its own stream, lock, staging
 * area and memory pool, so that concurrent callers overlap on the device instead of serialising on
 * one context mutex — the analogue of the reference's pool of ~NumCPU shard workers
 * (executor.go:6723-6737).  Batches are read-only once uploaded and may be used through any context
 * of the same device; the fragment cache (fbk_cache_*) is shared with the root context.  A fork is
 * closed with fbk_close, before its root. */
int32_t fbk_ctx_fork(fbk_ctx* ctx, fbk_ctx** out_child);

/* Options (22; round 5 removed fifteen A/B switches together with the kernels and paths that had lost their comparison —
 * DESIGN.md section 4 lists them).  The environment variables FBK_<NAME> are read ONCE, by fbk_open; afterwards only these
 * calls change an option.
 *   semantics (DEFAULT 1 = the reference's result, 0 = the arithmetically exact one; see fbk_count_range, fbk_topn):
 *     count_range_reference_quirk, topn_semantics;
 *   memory: matrix_shadow (1: the count matrix over encoded rows reads run containers of more than matrix_shadow_run runs and
 *     arrays of more than matrix_shadow_array values through dense shadows built per batch on first use — up to matrix_shadow_max_mb of device
 *     memory per batch and matrix_shadow_arena_x times its own arena (0: no such rule), fbk_batch_memory reports what a batch
 *     got; 0: every container is decoded in every query), setop_compact (fbk_batch_compact applied to the outputs of one-shot
 *     calls with FBK_SETOP_OPTIMIZE), matrix_pass_kb (per-shard matrices are produced in passes of at most this size),
 *     upload_chunk_mb, upload_threads (the pinned staging buffers of the uploads and the host threads that fill them);
 *   measurement: time_kernels = 1 makes the query-level calls (count matrix, n-way fold, BSI range / sum / min / max) record
 *     HIP events on the context's stream right before and after their dominant kernel; fbk_get_option("last_kernel_ns")
 *     then returns that kernel's duration for the last such call;
 *   kernel selection, each a choice the library makes by itself (the default) that a test or a measurement can pin — BOTH
 *     sides are product paths, chosen by the shape of the call or of the rows: dense_spb and matrix_spb (container slots per
 *     block of the dense count / the count matrices), matrix_tickets (dense single-tile count matrix: 1 its blocks take their units
 *     from a ticket counter, long units first — the device's XCDs run at different paces; 0 by block id), matrix_fused (encoded rows: -1 by the matrix size, 1 the matrix-core
 *     kernel that decodes the rows in place, 0 the generic pair kernel), matrix_fp4 (dense count matrix on the FP4 matrix
 *     instruction: -1 for matrices of several tiles), topk_device_sort (-1 by the field size), pair_kernels (0 by the rows'
 *     payload: the round-2 kernels for tiny containers, the table + probe kernels otherwise), pair_wpb (their waves per
 *     block: 0 by the rows' payload), setop_direct_encode (pair set-ops with FBK_SETOP_OPTIMIZE: 2 Container.optimize() inside
 *     the kernel; 1 / 0 bitmap or small-array cells and a separate re-encode pass — the byte-for-byte cross-check of the
 *     in-kernel encoders, and what a plan's launch-only form refuses).
 * Every value of every kernel-selection option gives the same results (the tests run them against each other). */
int32_t fbk_set_option(fbk_ctx* ctx, const char* name, int64_t value);
int32_t fbk_get_option(fbk_ctx* ctx, const char* name, int64_t* out_value);

int32_t fbk_abi_version(void);

/* Use an externally owned hipStream_t (e.g. the caller's framework stream) for all
 * subsequent launches on this context; NULL restores the context's own stream. */
int32_t fbk_set_stream(fbk_ctx* ctx, void* hip_stream);

/* Block until everything enqueued on the context's stream has finished. */
int32_t fbk_synchronize(fbk_ctx* ctx);

/* ---- residency ----------------------------------------------------------------
 * fbk_batch_upload makes the rows a query touches device resident.  Replaces the
 * per-row materialisation fragment.row / rowFromStorage (fragment.go:283-333) ->
 * Tx.OffsetRange (rbf/tx.go:1586-1638).  `descs` may be in any order; (row, key&15)
 * must be unique.  Containers with n == 0 are legal and stored as nil (the reference
 * stores empty results as nil, roaring.go:751-752). */
int32_t fbk_batch_upload(fbk_ctx* ctx, const fbk_container_desc* descs, uint64_t n_desc,
                         uint32_t n_rows, const void* payload, uint64_t payload_len,
                         fbk_batch** out_batch);

/* Dense upload: n_rows rows of 16 bitmap containers each, `words` = n_rows*16*1024
 * uint64 (row-major).  Cardinalities are recounted on the device (the analogue of
 * bitmapRepair, roaring.go:4193-4206).  Keys are assigned row*16+slot.  `words` may also be a
 * DEVICE pointer on the context's device (rows a device-side decode produced, or a benchmark's
 * on-device generator): the copy is then device to device. */
int32_t fbk_batch_upload_dense(fbk_ctx* ctx, const uint64_t* words, uint32_t n_rows,
                               fbk_batch** out_batch);

int32_t fbk_batch_free(fbk_ctx* ctx, fbk_batch* batch);

/* Device memory a batch holds: its arena, its heavy-row shadows (option matrix_shadow; built on the first count matrix
 * over encoded rows that reads the batch) and the shadow state — 0 not looked at yet, 1 shadowed, 2 nothing heavy or over a
 * limit (matrix_shadow_max_mb of device memory; matrix_shadow_arena_x times the batch's own arena; a quarter of the free
 * device memory): the batch's containers are then decoded in every query. */
int32_t fbk_batch_memory(fbk_ctx* ctx, const fbk_batch* batch, uint64_t* out_arena_bytes, uint64_t* out_shadow_bytes,
                         int32_t* out_shadow_state);

/* Move the containers of a batch the caller owns into a right-sized arena.  Materialising calls with FBK_SETOP_OPTIMIZE apply
 * Container.optimize() inside their kernel, which writes the encoded bytes into the head of an 8 KiB cell per container
 * (arena = n_rows x 16 x 8 KiB); the one-shot calls (fbk_setop, fbk_fold_n / fbk_union_n, fbk_bsi_range*, fbk_flip,
 * fbk_shift) compact their output before returning it (option setop_compact = 1, the default: payload sizes, one scan, one
 * copy per container — nothing is decoded), so a caller that keeps results holds their encoded size.  Outputs of plans and
 * prepared queries (fbk_plan_output, fbk_query_output) are BORROWED and keep the 8 KiB stride — their next run rewrites
 * them in place — and this call refuses them; copy them out (download, or a one-shot call) to keep them.
 * *out_arena_bytes = the arena size afterwards (also what fbk_batch_memory reports). */
int32_t fbk_batch_compact(fbk_ctx* ctx, fbk_batch* batch, uint64_t* out_arena_bytes);

/* Sizes needed to download a batch. */
int32_t fbk_batch_info(fbk_ctx* ctx, const fbk_batch* batch, uint32_t* n_rows,
                       uint64_t* n_containers, uint64_t* payload_bytes);

/* Copy a batch back to the host in the same flattened layout (non-nil containers only,
 * sorted by (row, slot)).  The Go side rebuilds containers with
 * NewContainerArray/BitmapN/RunN (container_stash.go).  Replaces result Row
 * construction (row.go:561-610). */
int32_t fbk_batch_download(fbk_ctx* ctx, const fbk_batch* batch, fbk_container_desc* descs_out,
                           uint64_t descs_cap, void* payload_out, uint64_t payload_cap);

/* ---- serialised roaring (the reference's interchange format) ------------------------------
 * fbk_batch_upload_roaring makes the containers of ONE serialised bitmap device resident:
 * either the Pilosa format Bitmap.WriteTo emits (roaring.go:1730-1817: cookie 12348, 12-byte
 * {key u64, type u16, N-1 u16} headers, u32 offsets, payloads with a u16 count prefix on run
 * containers, :4054-4108) or the official RoaringBitmap format (cookies 12346 / 12347,
 * readOfficialHeader roaring.go:6948-7006, runs stored {start, length-1} and converted on
 * read, :2239-2247).  It replaces Bitmap.UnmarshalBinary / NewRoaringIterator
 * (roaring.go:1945-2262) + the per-container copy into fragment storage
 * (ImportRoaringBits, fragment.go:2038-2165): the blob crosses PCIe once and is unpacked by a
 * device kernel.  Batch row i holds the containers whose key >> 4 equals out_row_ids[i]
 * (ascending; for fragment storage that is the row ID, for a serialised Row the shard number);
 * slot = key & 15.  out_row_ids may be NULL; *out_n_rows is always set.
 * An ops log behind the containers of a Pilosa-format FILE image is replayed as
 * Bitmap.UnmarshalBinary does (unmarshal_binary.go:66-95; op layout and FNV-1a checksum
 * roaring.go:6325-6431): every op's checksum is verified on the host, consecutive add / remove /
 * addN / removeN ops collapse to the last op per position and are applied as one union and one
 * difference on the device, addRoaring / removeRoaring ops upload their nested image from the
 * same blob and fold it in; the containers that come out are Optimize()d.  The row list then
 * also names every row an op touches (such a row may end up empty).  A truncated or corrupt op
 * is FBK_E_INVALID (the reference's FileShouldBeTruncatedError), nothing is uploaded. */
int32_t fbk_batch_upload_roaring(fbk_ctx* ctx, const void* data, uint64_t len, fbk_batch** out_batch,
                                 uint64_t* out_row_ids, uint32_t row_cap, uint32_t* out_n_rows);

/* RBF, the reference's storage file (rbf/rbf.go): make the containers of ONE bitmap b-tree
 * (one fragment: index/field/view/shard, rbf.go rbfName) device resident straight from the file
 * image.  `file` is the database file (or mmap) from page 0; `root_pgno` the bitmap's root page
 * (fbk_rbf_find_root resolves a name through the root records, rbf.go:222-255).  The host walks
 * branch pages and reads cell headers (rbf.go:488-520, 616-626); array / RLE cell payloads and
 * the 8 KiB pages that BitmapPtr cells point to are moved into the arena by the device unpack
 * kernel after ONE H2D copy of the file image.  Replaces Tx.ContainerIterator / OffsetRange ->
 * cursor -> toContainer (rbf/tx.go:1333, 1586-1638, rbf/cursorx.go:230-266).  Row mapping as
 * fbk_batch_upload_roaring: batch row i holds keys with key >> 4 == out_row_ids[i]. */
int32_t fbk_rbf_find_root(const void* file, uint64_t len, const char* name, uint32_t* out_pgno);
int32_t fbk_batch_upload_rbf(fbk_ctx* ctx, const void* file, uint64_t len, uint32_t root_pgno, fbk_batch** out_batch,
                             uint64_t* out_row_ids, uint32_t row_cap, uint32_t* out_n_rows);

/* Serialise a batch in the Pilosa format (what Bitmap.WriteTo / writeToUnoptimized write,
 * roaring.go:1730-1817): non-empty containers in (row, slot) order, whose keys must be
 * strictly ascending.  Containers are written in their current encoding — WriteTo runs
 * Optimize() first, so produce the batch with FBK_SETOP_OPTIMIZE to get byte-identical
 * output.  fbk_batch_roaring_size reports the bytes needed. */
int32_t fbk_batch_roaring_size(fbk_ctx* ctx, const fbk_batch* batch, uint64_t* out_bytes);
int32_t fbk_batch_download_roaring(fbk_ctx* ctx, const fbk_batch* batch, void* out, uint64_t cap, uint64_t* out_len);

/* ---- device fragment cache ----------------------------------------------------------------
 * The reference re-materialises a row from storage on every use (fragment.row,
 * fragment.go:283-333; zero-copy views of mmapped pages valid inside one Tx,
 * rbf/cursorx.go:243-247).  On the GPU the steady state keeps a fragment's rows resident:
 * entries are keyed by the reference's fragment identity ("index/field/view/shard", the RBF
 * bitmap name) and carry a version the host bumps whenever the fragment is written (every
 * write path of fragment.go ends in one Tx commit), so a stale entry is a miss.
 *   put        hands a batch (and the row ids fbk_batch_upload_rbf/_roaring returned) to the
 *              cache, which now owns it; an existing entry under the key is replaced.
 *   get        FBK_OK + the batch pinned for the caller, or FBK_E_NOTFOUND (absent / other
 *              version; a stale entry is dropped).  The pointers stay valid until release.
 *   release    unpins; invalidated or evicted entries are freed at their last release.
 *   invalidate drops every entry whose key starts with key_prefix ("" = everything).
 *   configure  sets the byte budget (default 128 GiB of the 288 GB HBM); least recently used
 *              unpinned entries are evicted beyond it. */
int32_t fbk_cache_put(fbk_ctx* ctx, const char* key, uint64_t version, fbk_batch* batch, const uint64_t* row_ids,
                      uint32_t n_row_ids);
int32_t fbk_cache_get(fbk_ctx* ctx, const char* key, uint64_t version, const fbk_batch** out_batch,
                      const uint64_t** out_row_ids, uint32_t* out_n_row_ids);
int32_t fbk_cache_release(fbk_ctx* ctx, const fbk_batch* batch);
int32_t fbk_cache_invalidate(fbk_ctx* ctx, const char* key_prefix, uint32_t* out_dropped);
int32_t fbk_cache_configure(fbk_ctx* ctx, uint64_t cap_bytes);
int32_t fbk_cache_stats(fbk_ctx* ctx, uint64_t* entries, uint64_t* bytes, uint64_t* hits, uint64_t* misses,
                        uint64_t* evictions);

/* ---- counts -------------------------------------------------------------------- */

/* out[i] = Row.Count of rows[i]: sum of stored container N (roaring.go:542,
 * containers_slice.go:120-126, row.go:446). */
int32_t fbk_count(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, uint64_t n,
                  uint64_t* out_counts);

/* out[i] = number of bits of rows[i] in [start, end), positions relative to the row
 * (0 <= start <= end <= 2^20).  Replaces Bitmap.CountRange (roaring.go:573-615) ->
 * Container.countRange -> ArrayCountRange / BitmapCountRange / RunCountRange
 * (roaring.go:3074-3232), which fragment.go:237,396,444 use to count one row of a fragment.
 * The device counts bits; this equals the reference everywhere except on run containers that
 * hold a run whose last value == end-1+1 (`iv.Last == end`), where RunCountRange's
 * "subset of range" and "overlaps end" branches both fire and over-count (roaring.go:3216-3227)
 * — a case the reference's own callers (container-aligned ranges) never produce.
 * DEFAULT (option "count_range_reference_quirk" = 1): RunCountRange as written, over-count included — the
 * reference's number on the same inputs.  = 0 (fbk_set_option, or FBK_COUNT_RANGE_REFERENCE_QUIRK=0 in the
 * environment of fbk_open): the bit count.  Both modes are tested against the oracle (tests/test_gpu_parity.py). */
int32_t fbk_count_range(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, uint64_t n, uint64_t start,
                        uint64_t end, uint64_t* out_counts);

/* out[i] = |A.rows_a[i] ∩ B.rows_b[i]| without materialising.  Replaces
 * RowSegment.IntersectionCount (row.go:556) -> Bitmap.IntersectionCount
 * (roaring.go:711-733) -> intersectionCount (roaring.go:4477-4614), one call per
 * (query,node) instead of one per container.  a and b may be the same batch. */
int32_t fbk_intersection_count(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a,
                               const fbk_batch* b, const uint32_t* rows_b, uint64_t n_pairs,
                               uint64_t* out_counts);

/* ---- materialising set operations ---------------------------------------------- */

/* out row i = A.rows_a[i] <op> B.rows_b[i]; out_counts[i] (may be NULL) = its cardinality.
 * Replaces RowSegment.{Intersect,Union,Difference,Xor} (row.go:561-610) ->
 * Bitmap.{Intersect,Union,Difference,Xor} (roaring.go:736,1272,1564,1598).  The
 * cardinality is produced in the same pass, as the reference fuses it
 * (roaring.go:4971-4974).  Output keys are taken from operand A (or B where A's slot
 * is nil). */
int32_t fbk_setop(fbk_ctx* ctx, int32_t op, const fbk_batch* a, const uint32_t* rows_a,
                  const fbk_batch* b, const uint32_t* rows_b, uint64_t n_pairs, uint32_t flags,
                  fbk_batch** out_batch, uint64_t* out_counts);

/* ---- plans: launch-only hot path -------------------------------------------------
 * A plan is a prepared list of row pairs whose index arrays and result buffers are
 * device resident, so one step of the hot path is kernel launches only (no host
 * allocation, copy or synchronisation).  It is what ONE batch call per (query, node)
 * from mapperLocal (executor.go:6742-6790) becomes: the per-shard mapFn closure
 * (executor.go:5871) is the pair list, the reduceFn (executor.go:5880) is
 * fbk_plan_total.  `device_counts_or_null`, when given, is a caller-owned device buffer
 * of n_pairs uint64 (e.g. the tensor a collective library will all-reduce). */
typedef struct fbk_plan fbk_plan;

int32_t fbk_plan_create(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, const fbk_batch* b,
                        const uint32_t* rows_b, uint64_t n_pairs, void* device_counts_or_null,
                        fbk_plan** out_plan);
int32_t fbk_plan_free(fbk_ctx* ctx, fbk_plan* plan);

/* Enqueue counts[i] = |A.rows_a[i] ∩ B.rows_b[i]| (Bitmap.IntersectionCount,
 * roaring.go:711-733).  Asynchronous. */
int32_t fbk_plan_intersection_count(fbk_ctx* ctx, fbk_plan* plan);

/* fbk_plan_intersection_count + fbk_plan_total in ONE launch: the per-node sum is fused into
 * the counting kernel (its last block to finish adds up the per-pair counts), i.e. one step of
 * Count(Intersect(Row, Row)) over all local shards — mapFn and reduceFn of executeCount,
 * executor.go:5871-5880 — is a single kernel.  Asynchronous. */
int32_t fbk_plan_intersection_count_total(fbk_ctx* ctx, fbk_plan* plan, void* device_total_or_null);

/* The same counts, with the per-node reduce done by accumulation: every workgroup adds its count
 * to *device_accum (device memory, uint64), which the CALLER has zeroed — e.g. one slot of a
 * vector that is cleared once per N steps, or of an all-reduce bucket.  One launch and nothing
 * serial behind the last workgroup (fbk_plan_intersection_count_total pays ~3 us for its final
 * pass, fbk_plan_intersection_count + fbk_plan_total ~2 us for the second launch). */
int32_t fbk_plan_intersection_count_accumulate(fbk_ctx* ctx, fbk_plan* plan, void* device_accum);

/* Enqueue out row i = A.rows_a[i] <op> B.rows_b[i] and counts[i] = its cardinality
 * (Bitmap.Intersect/Union/Xor/Difference + Count, roaring.go:736,1272,1598,1564;
 * executeCount's mapFn, executor.go:5871-5876).  The output batch is owned by the plan
 * and overwritten by the next enqueue.  Asynchronous.  flags = FBK_SETOP_OPTIMIZE (round 4): the
 * kernel applies Container.optimize() itself and writes the encoded container into the head of each
 * result cell — still launch-only (option setop_direct_encode = 2, the default; with 0 / 1 the
 * re-encode is a separate pass that sizes its output on the host, which only fbk_setop can run). */
int32_t fbk_plan_setop(fbk_ctx* ctx, fbk_plan* plan, int32_t op, uint32_t flags);

/* Enqueue total = sum_i counts[i] (executeCount reduceFn, executor.go:5880) into the
 * plan's total cell or into a caller-owned device uint64.  Asynchronous. */
int32_t fbk_plan_total(fbk_ctx* ctx, fbk_plan* plan, void* device_total_or_null);

/* Synchronise and copy counts (n_pairs, may be NULL) and total (may be NULL) to the host. */
int32_t fbk_plan_read(fbk_ctx* ctx, fbk_plan* plan, uint64_t* out_counts, uint64_t* out_total);

/* Borrow / take ownership of the plan's set-op output batch. */
int32_t fbk_plan_output(fbk_ctx* ctx, fbk_plan* plan, fbk_batch** out_batch);
int32_t fbk_plan_detach_output(fbk_ctx* ctx, fbk_plan* plan, fbk_batch** out_batch);

/* ---- prepared queries: the launch-only form of the query-level calls -------------------------
 * fbk_plan_* above makes the PAIR operations launch-only.  A prepared query does the same for the
 * GroupBy / TopK count matrix (fbk_count_matrix), the n-way fold with its fused count
 * (fbk_fold_n_intersection_count / fbk_union_n_intersection_count) and BSI Sum / the one-pass
 * Sum(Range) (fbk_bsi_sum, fbk_bsi_range_sum): the row lists are validated and uploaded ONCE, the
 * result buffers live on the device, and every execution is memset + kernel(s) on the context's
 * stream — no allocation, no host<->device copy, no synchronisation.  It is what one (query, node)
 * call of mapperLocal (executor.go:6742) becomes when the same query shape runs again on resident
 * fragments, and it leaves the partial result where the multi-GPU reduce wants it (the cell of an
 * all-reduce).  The batches a query was prepared on must outlive it and must not be rewritten; calls on
 * one query are serialised by its context's mutex (run and read from different threads are safe, the
 * read returns the result of whichever run was enqueued last).
 *
 *   fbk_query_count_matrix              arguments as fbk_count_matrix; keep_per_shard != 0 keeps the
 *                                       per-shard matrices readable (fbk_query_read's out1)
 *   fbk_query_fold_intersection_count   arguments as fbk_fold_n_intersection_count
 *   fbk_query_bsi_sum                   op == 0: fbk_bsi_sum; op = FBK_BSI_*: fbk_bsi_range_sum in one
 *                                       pass (FBK_E_INVALID for the predicates only the two-pass form serves)
 *   fbk_query_run(q, device_out, flags) enqueue one execution.  device_out == NULL: the query's own
 *                                       result buffer; otherwise a caller-owned device buffer of the
 *                                       result's size (count matrix: n_a * n_b uint64; fold: n_groups
 *                                       uint64).  FBK_QUERY_ACCUMULATE adds to what the buffer holds
 *                                       instead of overwriting it (count-valued kinds only).
 *   fbk_query_result                    the query's own device result buffer and its size in bytes
 *   fbk_query_read(q, out0, out1)       synchronise; copy the last run's result to the host:
 *                                       count matrix: out0 = total [n_a * n_b] uint64, out1 = per-shard
 *                                       [n_shards][n_a * n_b] (or NULL); fold: out0 = counts [n_groups];
 *                                       BSI: out0 = int64 sums [n_shards], out1 = uint64 counts [n_shards].
 *
 * Queries with a ROW result keep the rows on the device too (round 4: the one-shot forms paid 40-80 us of
 * host work around a 130 us kernel for allocating the output batch, uploading the plane program and
 * downloading descriptors that nobody reads when the rows only feed the next operator):
 *   fbk_query_bsi_range                 Row(v op predicate), fragment.rangeOp (fragment.go:937-1303), arguments
 *                                       as fbk_bsi_range (8 KiB cells).  run = memset + k_bsi_range_slot;
 *                                       read: out0 = uint64 cardinalities [n_shards]
 *   fbk_query_fold                      the materialised n-way fold of any of the four operations (fbk_fold_n; flags =
 *                                       FBK_SETOP_OPTIMIZE: Container.optimize() in the kernel's epilogue).
 *                                       run = memset + ONE launch; read: out0 = uint64 cardinalities [n_groups]
 *   fbk_query_output(q, &batch)         the output batch of the last run, BORROWED: owned by the query, rewritten
 *                                       by its next run, freed with it.  A valid operand of any later call on the
 *                                       same context (the filter of fbk_bsi_sum / fbk_query_bsi_sum, a plan's row).
 *   fbk_query_topn                      fbk_topn with the per-shard counts, the totals AND the ordering (device
 *                                       radix sort, scratch sized at prepare) launch-only; read: out0 = uint32
 *                                       {number of results r, r row indexes} (capacity 1 + min(n, n_a), n = 0:
 *                                       1 + n_a), out1 = uint64 counts [r].  device_out must be NULL. */
typedef struct fbk_query fbk_query;
#define FBK_QUERY_ACCUMULATE 1u
int32_t fbk_query_count_matrix(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, uint32_t n_a, const fbk_batch* b,
                               const uint32_t* rows_b, uint32_t n_b, const fbk_batch* filter, const uint32_t* rows_f,
                               uint32_t n_shards, uint32_t keep_per_shard, fbk_query** out_query);
int32_t fbk_query_fold_intersection_count(fbk_ctx* ctx, int32_t op, const fbk_batch* batch, const uint32_t* rows, uint64_t n_groups,
                                          uint32_t k, const fbk_batch* filter, const uint32_t* rows_f, fbk_query** out_query);
int32_t fbk_query_bsi_sum(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards, uint32_t bit_depth,
                          int32_t op, int64_t predicate, const fbk_batch* filter, const uint32_t* rows_f, fbk_query** out_query);
int32_t fbk_query_bsi_range(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards, int32_t op,
                            uint32_t bit_depth, int64_t predicate, fbk_query** out_query);
int32_t fbk_query_fold(fbk_ctx* ctx, int32_t op, const fbk_batch* batch, const uint32_t* rows, uint64_t n_groups, uint32_t k,
                       uint32_t flags, fbk_query** out_query);
int32_t fbk_query_topn(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, uint32_t n_a, const fbk_batch* filter,
                       const uint32_t* rows_f, uint32_t n_shards, uint32_t n, uint64_t min_threshold, uint64_t tanimoto_threshold,
                       fbk_query** out_query);
int32_t fbk_query_output(fbk_ctx* ctx, fbk_query* query, fbk_batch** out_batch);
int32_t fbk_query_run(fbk_ctx* ctx, fbk_query* query, void* device_out, uint32_t flags);
int32_t fbk_query_result(fbk_ctx* ctx, fbk_query* query, void** out_device_ptr, uint64_t* out_bytes);
int32_t fbk_query_read(fbk_ctx* ctx, fbk_query* query, void* out0, void* out1);
int32_t fbk_query_free(fbk_ctx* ctx, fbk_query* query);

/* ---- n-way union ---------------------------------------------------------------------
 * rows holds n_groups groups of k row ordinals (group g = rows[g*k .. g*k+k)); out row g
 * is the union of the group's rows, out_counts[g] its cardinality.  Replaces
 * Row.Union(others...) (row.go:288) -> RowSegment.Union (:572) -> the n-way
 * Bitmap.unionInPlace (roaring.go:1410-1561) and the streaming BitmapRowsUnion used by
 * UnionRows (roaring/filter.go:294-366, fragment.go:2489): the accumulation happens in
 * registers, the union is written once. */
int32_t fbk_union_n(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, uint64_t n_groups,
                    uint32_t k, uint32_t flags, fbk_batch** out_batch, uint64_t* out_counts);

/* Fused Union-of-k-rows then IntersectionCount against a filter row, without ever
 * materialising the union: out_counts[g] = |(∪ group g) ∩ filter.rows_f[g]|
 * (filter == NULL: |∪ group g|).  executor.go:5382 executeUnionShard followed by
 * Row.intersectionCount (row.go:226). */
int32_t fbk_union_n_intersection_count(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows,
                                       uint64_t n_groups, uint32_t k, const fbk_batch* filter,
                                       const uint32_t* rows_f, uint64_t* out_counts);

/* ---- n-way fold -----------------------------------------------------------------------
 * Generalisation of fbk_union_n to the four set operations, folded left to right over the k
 * rows of each group exactly as the executor's per-shard functions fold a call's children:
 *   FBK_OP_AND     r0 & r1 & ... & r(k-1)    executeIntersectShard  executor.go:5357-5380
 *   FBK_OP_OR      r0 | r1 | ...             executeUnionShard      executor.go:5382-5404
 *   FBK_OP_XOR     r0 ^ r1 ^ ...             executeXorShard        executor.go:5513-5552
 *   FBK_OP_ANDNOT  r0 \ r1 \ ... \ r(k-1)     executeDifferenceShard executor.go:2950-2983
 * (Bitmap.IntersectInPlace / Difference with several others, roaring.go:855-925, 1564-1595.)
 * One launch, the intermediate rows never touch HBM.  AND / ANDNOT need k >= 1. */
int32_t fbk_fold_n(fbk_ctx* ctx, int32_t op, const fbk_batch* batch, const uint32_t* rows, uint64_t n_groups,
                   uint32_t k, uint32_t flags, fbk_batch** out_batch, uint64_t* out_counts);

/* out_counts[g] = |fold(group g) ∩ filter.rows_f[g]| (filter == NULL: |fold(group g)|), never
 * materialised: Count(Intersect(...)) / Count(Union(...)) etc. in one pass
 * (executeCount -> executeBitmapCallShard, executor.go:5839-5892). */
int32_t fbk_fold_n_intersection_count(fbk_ctx* ctx, int32_t op, const fbk_batch* batch, const uint32_t* rows,
                                      uint64_t n_groups, uint32_t k, const fbk_batch* filter,
                                      const uint32_t* rows_f, uint64_t* out_counts);

/* ---- count matrix (GroupBy / TopN / TopK shape) --------------------------------------------
 * For every shard s, out[s][i][j] = |A.rows_a[s*n_a+i] ∩ B.rows_b[s*n_b+j] ∩ F.rows_f[s]|
 * (filter may be NULL).  out_total[i*n_b+j] is the sum over shards (mergeGroupCounts /
 * Pairs.Add arithmetic, executor.go:3728, 2852); out_per_shard (n_shards*n_a*n_b, may be
 * NULL) keeps the per-shard matrices.  Replaces groupByIterator.Next's
 * rows[last].intersectionCount(rows[last-1]) (executor.go:8893) and, with n_b == 1 and
 * B = the filter row, doTopK / fragment.top (executor.go:2705-2746, fragment.go:1317).
 * Limits: n_a, n_b <= 4096 for a matrix; with n_b == 1 (the TopK / TopN shape, served by a
 * dedicated rows-vs-filter kernel) n_a <= 2^22. */
int32_t fbk_count_matrix(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, uint32_t n_a,
                         const fbk_batch* b, const uint32_t* rows_b, uint32_t n_b, const fbk_batch* filter,
                         const uint32_t* rows_f, uint32_t n_shards, uint64_t* out_total,
                         uint64_t* out_per_shard);

/* ---- BSI (bit-sliced integers) ----------------------------------------------------------------
 * A BSI fragment of shard s occupies bit_depth+2 consecutive rows of `batch` starting at
 * base_rows[s]: +0 exists, +1 sign, +2+i magnitude bit i (fragment.go:62-65). */
#define FBK_BSI_EQ 1
#define FBK_BSI_NEQ 2
#define FBK_BSI_LT 3
#define FBK_BSI_LTE 4
#define FBK_BSI_GT 5
#define FBK_BSI_GTE 6

/* Per shard: out_counts[s] = |exists ∩ filter|, out_sums[s] = Σ_i 2^i (|pos ∩ bit_i| − |neg ∩ bit_i|)
 * with uint64 wrap-around exactly as BitmapBSICountFilter (roaring/filter.go:1097-1218,
 * fragment.sum fragment.go:724-750).  filter == NULL means "no filter"; a filter row with no
 * container in a slot contributes nothing for that slot.  The caller adds count*Base
 * (executor.go:2205). */
int32_t fbk_bsi_sum(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards,
                    uint32_t bit_depth, const fbk_batch* filter, const uint32_t* rows_f, int64_t* out_sums,
                    uint64_t* out_counts);

/* Per shard: (out_vals[s], out_counts[s]) = fragment.min / fragment.max (fragment.go:754-853):
 * the smallest / largest stored value among the columns of exists ∩ filter and the number of
 * columns holding it; (0, 0) when there is no such column.  Sign-magnitude planes, MSB -> LSB
 * scan (minUnsigned :781, maxUnsigned :832).  The caller adds Base (field.go MinForShard /
 * MaxForShard) and folds shards with ValCount.Smaller / Larger (executor.go:8446, 8526). */
int32_t fbk_bsi_min(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards,
                    uint32_t bit_depth, const fbk_batch* filter, const uint32_t* rows_f, int64_t* out_vals,
                    uint64_t* out_counts);
int32_t fbk_bsi_max(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards,
                    uint32_t bit_depth, const fbk_batch* filter, const uint32_t* rows_f, int64_t* out_vals,
                    uint64_t* out_counts);

/* Distinct values of a BSI field among the columns of exists ∩ filter over all the shards given:
 * executeDistinctShardBSI (executor.go:2034-2153) — which gathers every column's value from the
 * bit planes — followed by the SignedRow.Union reduce over shards (executor.go:1190-1196).
 * out_values receives the sorted distinct stored values (sign applied, Base NOT added: the
 * caller adds bsiGroup.Base and splits into SignedRow{Neg, Pos}); *out_n their number, also when
 * FBK_E_CAPACITY reports that `cap` was too small.  The planes are transposed on the device
 * (64 x 64 bit-matrix transpose in registers), sorted and de-duplicated there. */
int32_t fbk_bsi_distinct(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards,
                         uint32_t bit_depth, const fbk_batch* filter, const uint32_t* rows_f, int64_t* out_values,
                         uint64_t cap, uint64_t* out_n);

/* TopK / TopN: totals[i] = sum over the shards of |A[shard][i] ∩ F[shard]| (of the stored row
 * cardinalities when filter == NULL), ordered on the device by count descending and row index
 * ascending inside one count, rows with a zero count dropped, at most k results (k = 0: all) —
 * executeTopK / doTopK (executor.go:2705-2774) with the order of BSIData.PivotDescending
 * (bsi.go:18-62); executeTopN / fragment.top (fragment.go:1317-1437) count the same intersections
 * (their rank-cache thresholds are approximations the device does not need).  rows_a is
 * [n_shards][n_a] (row i of every shard = the same field row), rows_f [n_shards].
 * out_index[j] = i, out_count[j] = totals[i]; *out_n = number of results, also when
 * FBK_E_CAPACITY reports that `cap` was too small.  Fields of more than 4096 rows are ordered by
 * a device radix sort and only the k winners cross the bus, not n_a counts (n_a <= 2^22). */
int32_t fbk_topk(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, uint32_t n_a, const fbk_batch* filter,
                 const uint32_t* rows_f, uint32_t n_shards, uint32_t k, uint32_t* out_index, uint64_t* out_count, uint32_t cap,
                 uint32_t* out_n);

/* TopN (executeTopN, executor.go:2779-2864; per shard fragment.top, fragment.go:1317-1437).  Per shard s and row i,
 * cnt = the row's stored cardinality and count = |row ∩ F_s| (count = cnt without a filter = the `Src` row).  The row
 * QUALIFIES in that shard iff cnt != 0, count != 0 and
 *   tanimoto_threshold > 0 and a filter is given:
 *       src*T/100 < cnt < src*100/T   and   ceil(count*100 / (cnt + src - count)) > T     (:1334-1391;
 *       src = |F_s|, T = tanimoto_threshold, a percentage; evaluated in integer arithmetic, which equals
 *       the reference's float64 comparisons for every count a shard can hold)
 *   otherwise:  cnt >= min_threshold and count >= min_threshold                              (:1357, :1394;
 *       the caller passes what executeTopNShard passes: 0 there becomes defaultMinThreshold = 1, the same rule)
 * and totals[i] = the sum of count over the shards where row i qualifies (Pairs.Add, cache.go:463).
 *
 * n = 0 or n >= n_a: every row with a total is returned, ordered count descending / row index ascending.
 * 0 < n < n_a, option topn_semantics = 1 (DEFAULT): the reference's two passes.  Pass 1: per SHARD, fragment.top with
 *   N = n walks the rows in rank order (cnt descending; equal cnt: row index ascending) — the first n qualifying rows,
 *   and with a filter every later row whose cnt passes the cnt-level test and whose count reaches the smallest count of
 *   those n (the heap only grows, :1404-1425) — and the ids of all shards are merged, untrimmed (executeTopNShards
 *   :2829-2864).  Pass 2: the totals of exactly those candidates (:2812-2818), ordered, trimmed to n (:2823-2825).  A row
 *   outside every shard's own list is not returned even if its total would rank: that IS the reference's answer
 *   (TestExecutor_Execute_TopN_fill_small, tests/golden/executor_topn_vectors.json).  One kernel (k_topn_candidates,
 *   one block per shard, histogram selection — no sort) next to the counting the call does anyway.
 * 0 < n < n_a, topn_semantics = 0: the exact top n of totals.
 * The rank cache itself (cache.go: at most CacheSize rows per fragment, refreshed lazily) is not modelled: every row given
 * in rows_a is ranked.  fbk_topk (executeTopK, an exact operator in the reference too) is fbk_topn with both thresholds 0
 * and no candidate pass. */
int32_t fbk_topn(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, uint32_t n_a, const fbk_batch* filter,
                 const uint32_t* rows_f, uint32_t n_shards, uint32_t n, uint64_t min_threshold, uint64_t tanimoto_threshold,
                 uint32_t* out_index, uint64_t* out_count, uint32_t cap, uint32_t* out_n);

/* One member's share of a TopN for the one-process-per-GPU deployment (featurebase_amd/dist.py topn_reduce): totals[i] as
 * fbk_topn computes them over THIS context's shards and candidates[i] != 0 iff some shard here lists row i in pass 1
 * (without a candidate pass — n = 0, n >= n_a, topn_semantics = 0 — every row with a total).  The ranks all-reduce
 * [totals | candidates] (2 n_a words, ONE collective), drop the rows no rank flagged, order and trim: the same answer
 * as fbk_topn over all shards, however they are dealt.  n_shards = 0: zeros.  Both outputs hold n_a uint64. */
int32_t fbk_topn_partials(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, uint32_t n_a, const fbk_batch* filter,
                          const uint32_t* rows_f, uint32_t n_shards, uint32_t n, uint64_t min_threshold, uint64_t tanimoto_threshold,
                          uint64_t* out_totals, uint64_t* out_candidates);

/* The TopK counts in the form executeTopKShard hands to its reducer: BSI planes over the ROW IDS
 * (bsiBuilder.Insert(rowID, count), bsi.go:251-284; merged across shards with AddBSI = fbk_bsi_add,
 * read back with PivotDescending, bsi.go:18-62).  totals[i] as in fbk_topk; out row p (p <
 * *out_depth = bits of the largest total) holds bit i iff bit p of totals[i] is set; row ids are the
 * indices i (n_a <= 2^20), container keys p * 16 + slot. */
int32_t fbk_topk_bsi(fbk_ctx* ctx, const fbk_batch* a, const uint32_t* rows_a, uint32_t n_a, const fbk_batch* filter,
                     const uint32_t* rows_f, uint32_t n_shards, uint32_t flags, fbk_batch** out_batch, uint32_t* out_depth);

/* out row i = rows[i] with the bits of the row-relative positions [start, end] (inclusive, 0 <= start
 * <= end < 2^20) negated: Bitmap.Flip (roaring/roaring.go:2769-2799); the container-level flip of the
 * reference's combination table (flipArray / flipBitmap / flipRun, roaring.go:6259-6274) is the
 * range of one slot.  No PQL call reaches Flip (Not is Difference(existence, row), executor.go:5554);
 * it completes the container algebra.  out_counts[i] = cardinality of out row i. */
int32_t fbk_flip(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, uint64_t n_rows, uint64_t start, uint64_t end,
                 uint32_t flags, fbk_batch** out_batch, uint64_t* out_counts);

/* Shift: out row i = rows[i] with every column moved up by one — Row.Shift (row.go:374-396) /
 * RowSegment.Shift (:613-626) / Bitmap.Shift(1) (roaring/roaring.go:1629-1662; shiftArray,
 * shiftBitmap, shiftRun :6184-6257), which executeShiftShard (executor.go:5818-5836) applies n
 * times.  The reference lets the bit that leaves a shard's last column fall into a container
 * keyed one past the segment, which Row.Columns() reports as the first column of the next shard;
 * here that bit is handed over explicitly: carry_rows[i] (or NULL) names the row of the
 * PREVIOUS shard whose column ShardWidth-1 becomes column 0 of out row i.  Either index may be
 * FBK_NO_ROW: rows[i] absent (a shard that exists only to receive the carried bit), no
 * predecessor.  out_counts[i] = cardinality of out row i (rows left empty are nil containers). */
#define FBK_NO_ROW 0xFFFFFFFFu
int32_t fbk_shift(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, const uint32_t* carry_rows, uint64_t n_rows,
                  uint32_t flags, fbk_batch** out_batch, uint64_t* out_counts);

/* Unsigned BSI addition z = x + y, plane by plane (ripple carry): roaring.Add
 * (roaring/add.go:12-849), which AddBSI (bsi.go:83-175) uses to merge per-shard TopK counts.
 * Group g of x is the depth_x rows rows_x[g*depth_x + i] (plane i = bit i, no exists / sign
 * planes), likewise y; out row g*(D+1) + i is plane i of the sum, D = max(depth_x, depth_y),
 * plane D holding the final carry.  Output container keys are out_row * 16 + slot (the fragment-storage
 * form rowID << 4 | slot with the output row ordinal as row id). */
int32_t fbk_bsi_add(fbk_ctx* ctx, const fbk_batch* x, const uint32_t* rows_x, uint32_t depth_x, const fbk_batch* y,
                    const uint32_t* rows_y, uint32_t depth_y, uint64_t n_groups, uint32_t flags, fbk_batch** out_batch);

/* out row s = columns of shard s whose value satisfies `op predicate` (fragment.rangeOp,
 * fragment.go:937-1208); the container keys of the result are s * 16 + slot. */
int32_t fbk_bsi_range(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards,
                      int32_t op, uint32_t bit_depth, int64_t predicate, uint32_t flags, fbk_batch** out_batch,
                      uint64_t* out_counts);

/* Sum(Row(v op predicate), field = v) — the values of the columns that satisfy a range predicate on the SAME BSI
 * field, summed: executeSumCountShard (executor.go:2155) with the filter that executeRowBSIGroupShard (:5249) builds
 * from fragment.rangeOp.  The reference computes the range as a Row and then runs fragment.sum with that Row as the
 * filter (fragment.go:724-750): every bit plane is read twice.  Here one pass: a column that the MSB -> LSB scan
 * matches at plane i has the predicate's bits above i, so its high part is a constant per plane and its low planes
 * are still to come (k_bsi_range_sum_slot).  `filter` (optional, rows_f[s] per shard) restricts the
 * columns first, as Sum(Intersect(Row(v op k), <filter>), field = v).  out_sums[s] / out_counts[s] equal fbk_bsi_sum
 * with filter = fbk_bsi_range(op, predicate) (∩ filter) bit for bit; the reference's special forms (predicate 0,
 * saturated predicates, EQ / NEQ) run as exactly those two calls. */
int32_t fbk_bsi_range_sum(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards, int32_t op,
                          uint32_t bit_depth, int64_t predicate, const fbk_batch* filter, const uint32_t* rows_f,
                          int64_t* out_sums, uint64_t* out_counts);

/* The per-plane schedule fbk_bsi_range_sum runs for (op, bit_depth, predicate) — host arithmetic only, no device:
 * out_actions[i] (64 entries) = what happens at plane i (0 sums only, 1 remaining &= plane, 2 remaining &= ~plane,
 * 3 matched |= remaining & plane, 4 matched |= remaining & ~plane), out_vhi[i] (64) = value of a column matched at
 * plane i above and at that plane, *out_scan_positive = the scanned sign class, *out_take_other = the other class
 * belongs to the result as a whole.  Returns 1 (not an error) when this predicate takes the two-pass path.  For
 * tests and for callers that want to know which path a query will take. */
int32_t fbk_bsi_range_sum_plan(int32_t op, uint32_t bit_depth, int64_t predicate, uint8_t* out_actions, uint64_t* out_vhi,
                               uint32_t* out_scan_positive, uint32_t* out_take_other);

/* Sum(Row(lo <= v <= hi), field = v): fbk_bsi_range_sum's two-sided form (fragment.rangeBetween + fragment.sum on the same
 * field; `filter` as there).  One pass over the planes for batches in the dense layout (fbk_batch_upload_dense) and
 * lo < hi within the field's range: two scan lanes — the positives up to hi and the negatives up to |lo|, or, for bounds of
 * one sign, the columns sharing the bounds' common prefix split at the highest differing bit into ">= lower" and
 * "<= upper" (k_bsi_between_sum_half); otherwise fbk_bsi_range_between + fbk_bsi_sum.  Totals equal those
 * two calls bit for bit.  fbk_bsi_between_sum_plan: the schedule (host arithmetic only; returns 1 for the two-pass path):
 * out_actions[2][64], out_vhi[2][64], out_split[64], out_vfin[2], out_flags[5] = {lane 0 positive, lane 1 positive,
 * lane 1 starts as its class, lane 0 whole class, lane 1 whole class}. */
int32_t fbk_bsi_range_between_sum(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows, uint32_t n_shards,
                                  uint32_t bit_depth, int64_t lo, int64_t hi, const fbk_batch* filter, const uint32_t* rows_f,
                                  int64_t* out_sums, uint64_t* out_counts);
int32_t fbk_bsi_between_sum_plan(uint32_t bit_depth, int64_t lo, int64_t hi, uint8_t* out_actions, uint64_t* out_vhi,
                                 uint8_t* out_split, uint64_t* out_vfin, uint32_t* out_flags);

/* lo <= value <= hi (fragment.rangeBetween, fragment.go:1213-1303). */
int32_t fbk_bsi_range_between(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* base_rows,
                              uint32_t n_shards, uint32_t bit_depth, int64_t lo, int64_t hi, uint32_t flags,
                              fbk_batch** out_batch, uint64_t* out_counts);


/* fragment.rows(start, filters...) (fragment.go:2465-2486; executeRowsShard, executor.go:4077-4182): which of
 * the rows rows[0 .. n_rows) — the fragment's rows from `start` on, in ascending row-id order, one device row each
 * — hold anything, hold `column`, and fall under `limit`.  The reference streams the fragment's containers in key
 * order through the filter protocol of roaring/filter.go (BitmapRowFilter :371-469 over BitmapColumnFilter :226-249
 * and BitmapRowLimitFilter :471-509, ApplyFilterToIterator :1062-1085: skip-ahead by YesKey / NoKey so that a column
 * filter opens one container per row); here every row is examined at once (16 lanes read a row's 16 descriptors,
 * one lane probes the column's container) and the survivors are compacted in order.
 *   column  FBK_NO_COLUMN, or a column of the shard (0 .. 2^20 - 1): the row must contain it
 *   limit   0 = none (a PQL `limit=0` is NewBitmapRowLimitFilter(0), which ends the scan at once: the caller returns no
 *           rows without calling); else the reference's composition [column filter, limit filter]: the limit filter spends one
 *           of its rows on every row in which the scan looks at a container, matching or not.  Without a column
 *           that is every non-empty row (result: the first `limit` non-empty rows).  With a column the scan leaves a
 *           row r in which it saw the column's slot c (or a later one) by skipping to key (r + 1, c), so a row whose
 *           containers all lie below slot c is not counted when it directly follows (id + 1) such a row — the result
 *           is "the rows among the first `limit` COUNTED rows that contain the column".
 *   row_ids the fragment row ids of rows[0 .. n_rows), strictly ascending; required when a column AND a limit are
 *           given (the rule above needs to know which rows follow each other directly), otherwise may be NULL
 *   out_idx[cap]  positions (into rows[]) of the matching rows, ascending; *out_n = how many (FBK_E_CAPACITY when
 *           cap is too small: *out_n says how many there are). */
#define FBK_NO_COLUMN (~0ull)
int32_t fbk_rows(fbk_ctx* ctx, const fbk_batch* batch, const uint32_t* rows, const uint64_t* row_ids, uint64_t n_rows, uint64_t column,
                 uint64_t limit, uint32_t* out_idx, uint64_t cap, uint64_t* out_n);

/* ---- several GPUs in one process ---------------------------------------------------------------
 * The reference maps per-shard functions on the node that owns each shard and folds count-valued
 * results with an associative reduceFn in the same process (mapReduce / mapperLocal,
 * executor.go:6449-6533, 6742-6790; executeCount's reduceFn :5880; mergeGroupCounts :3728).  A
 * group is that for the GPUs of one node: one context per device (member m), shard s on member
 * s mod G, every member's work enqueued on its own stream so that the devices run concurrently, and
 * the partial counts reduced by one of
 *   FBK_REDUCE_HOST  G device->pinned-host copies of the partials and an add on the host
 *   FBK_REDUCE_PEER  one kernel on member 0 loading the other devices' partials over xGMI (peer access)
 *   FBK_REDUCE_RCCL  ncclAllReduce over single-process communicators (ncclCommInitAll); librccl is
 *                    dlopen'ed on first use, so libfbk.so itself links only the HIP runtime.
 * Bitmap-valued results are not exchanged (they stay per shard, executor.go:1767).  Data is made
 * resident and plans are created through the member contexts (fbk_group_member) with the ordinary
 * calls.  The same device ordinal may appear several times (members share the device): that is how
 * the G > 1 path is exercised on a one-GPU box; FBK_REDUCE_RCCL refuses such a group. */
typedef struct fbk_group fbk_group;
#define FBK_REDUCE_HOST 0
#define FBK_REDUCE_PEER 1
#define FBK_REDUCE_RCCL 2
int32_t fbk_group_open(const int32_t* devices, uint32_t n_devices, uint32_t flags, fbk_group** out_group);
int32_t fbk_group_close(fbk_group* group);
int32_t fbk_group_size(const fbk_group* group, uint32_t* out_n);
int32_t fbk_group_member(fbk_group* group, uint32_t i, fbk_ctx** out_ctx); /* borrowed: closed by fbk_group_close */
int32_t fbk_group_set_reduce(fbk_group* group, int32_t mode);

/* One step of Count(Intersect(Row, Row)) over the shards of all members: plans[m] (created on
 * member m over the shards it owns; NULL = none) runs as one launch per device with the per-device
 * sum fused, all devices concurrently; *out_total = sum over the members (executor.go:5871-5880). */
int32_t fbk_group_plan_intersection_count_total(fbk_group* group, fbk_plan* const* plans, uint64_t* out_total);

/* The count matrix of fbk_count_matrix over the shards of all members: per_member[m] are member m's
 * arguments (n_shards == 0: the member has no shard of this query), out_total[i * n_b + j] the sum
 * over every shard of every member (mergeGroupCounts across nodes, executor.go:3728). */
typedef struct fbk_matrix_args {
  const fbk_batch* a;
  const uint32_t* rows_a; /* [n_shards][n_a] */
  const fbk_batch* b;
  const uint32_t* rows_b; /* [n_shards][n_b] */
  const fbk_batch* filter; /* may be NULL */
  const uint32_t* rows_f; /* [n_shards] */
  uint32_t n_shards;
  uint32_t pad;
} fbk_matrix_args;
int32_t fbk_group_count_matrix(fbk_group* group, const fbk_matrix_args* per_member, uint32_t n_a, uint32_t n_b,
                               uint64_t* out_total);

/* Sum(field = v) over the shards of all members (executeSumCountShard, executor.go:2155, reduced by ValCount.Add
 * :8438): per_member[m] are member m's arguments of fbk_bsi_sum (n_shards == 0: none of its shards), each member
 * folds its shards on its device into {psum, nsum, count} and the three words are reduced over the members.
 * *out_sum = int64(psum) - int64(nsum) with uint64 wrap-around (roaring/filter.go:1103-1108) = the sum of
 * fbk_bsi_sum's out_sums over every shard of every member; *out_count likewise.  The caller adds count * Base. */
typedef struct fbk_bsi_args {
  const fbk_batch* batch;
  const uint32_t* base_rows; /* [n_shards] */
  const fbk_batch* filter;   /* may be NULL */
  const uint32_t* rows_f;    /* [n_shards] */
  uint32_t n_shards;
  uint32_t pad;
} fbk_bsi_args;
int32_t fbk_group_bsi_sum(fbk_group* group, const fbk_bsi_args* per_member, uint32_t bit_depth, int64_t* out_sum,
                          uint64_t* out_count);

/* TopN over the shards of all members: fbk_topn's answer over the union of the members' shards (the same two passes
 * under topn_semantics = 1, exact under 0), independent of how the shards are dealt — in the reference a node returns its
 * shards' merged pairs UNTRIMMED (executeTopNShards :2829-2864), the truncation to n happens per shard inside fragment.top.
 * Every member counts its shards and flags its shards' candidates; ONE reduce of [totals | candidate flags] (2 n_a words;
 * n_a without a candidate pass) over the members, then order and trim on the host.  per_member[m] are member m's
 * arguments of fbk_topn; outputs as fbk_topn.  The first member's topn_semantics decides for the group. */
typedef struct fbk_topn_args {
  const fbk_batch* a;
  const uint32_t* rows_a;  /* [n_shards][n_a] */
  const fbk_batch* filter; /* may be NULL */
  const uint32_t* rows_f;  /* [n_shards] */
  uint32_t n_shards;
  uint32_t pad;
} fbk_topn_args;
int32_t fbk_group_topn(fbk_group* group, const fbk_topn_args* per_member, uint32_t n_a, uint32_t n, uint64_t min_threshold,
                       uint64_t tanimoto_threshold, uint32_t* out_index, uint64_t* out_count, uint32_t cap, uint32_t* out_n);

/* The reduce alone, for count-valued partials the caller produced with the member contexts (BSI
 * sums, TopK counts, fold counts): device_partials[m] = `words` uint64 on member m's device (NULL =
 * zeros), produced on member m's stream. */
int32_t fbk_group_reduce_u64(fbk_group* group, void* const* device_partials, uint64_t words, uint64_t* out_total);

/* ---- one process per GPU: the exchange step issued by the library -------------------------------
 * The deployment of executor.go:6449-6533 on one node is one process per GPU, each with a context over the shards it
 * owns; the only exchange is the sum of count-valued partials (reduceFn).  Issued through PyTorch that all-reduce costs
 * the launching thread ~28 us per call (profiles/r06_collective_host_cost.json) — more than half of a 41 us headline step.
 * These calls put it on the library's side: a communicator per context over RCCL (dlopen'ed, as for fbk_group), created
 * from a unique id that rank 0 draws and the host code hands to the other ranks by whatever channel it has (the Go shim:
 * its cluster RPC; bench.py / featurebase_amd.dist: torch.distributed.broadcast_object_list).
 *   fbk_comm_unique_id      rank 0: 128 bytes (ncclGetUniqueId)
 *   fbk_comm_init           every rank: ncclCommInitRank on the context's device; collective (blocks until all ranks arrive)
 *   fbk_comm_all_reduce_u64 in place sum of n_words uint64 at device_words over the ranks, ASYNCHRONOUS: ordered after what
 *                           the context's stream holds at the call, run on the communicator's own stream, so the kernels
 *                           of the following queries overlap it.  The words must not be touched until a fence.
 *   fbk_comm_fence          the context's stream waits for every all-reduce enqueued so far (one event: no host wait)
 *   fbk_comm_close          destroys the communicator (fbk_close does it too) */
#define FBK_COMM_ID_BYTES 128
int32_t fbk_comm_unique_id(uint8_t* out_id /* FBK_COMM_ID_BYTES */);
int32_t fbk_comm_init(fbk_ctx* ctx, const uint8_t* id /* FBK_COMM_ID_BYTES */, int32_t n_ranks, int32_t rank);
int32_t fbk_comm_all_reduce_u64(fbk_ctx* ctx, void* device_words, uint64_t n_words);
int32_t fbk_comm_fence(fbk_ctx* ctx);
int32_t fbk_comm_close(fbk_ctx* ctx);

/* Message (and status code) of the last failing fbk_group_* call on THIS group, copied into a caller
 * buffer: the form a cgo binding must use (see fbk_last_error_r).  Group-level failures — argument
 * checks, "plan was not created on member m", peer / RCCL set-up, buffer growth — have no member
 * context to be recorded on; failures inside a member's enqueue are recorded on the group AND on that
 * member.  While a group call runs it holds every member's context lock (in member order) until the
 * member streams are synchronised, and a call that fails half-way drains the members it had enqueued. */
int32_t fbk_group_last_error_r(fbk_group* group, char* buf, uint64_t cap, int32_t* out_code);

#ifdef __cplusplus
}
#endif
#endif /* FBK_H */
