/*
 * roaring_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's roaring container algebra for the hot path
 * named in BASELINE.json (FeatureBase roaring/roaring.go, roaring/container_stash.go,
 * roaring/filter.go, fragment.go, executor.go).  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or load this; the product (featurebase_amd/, libfbk.so)
 * never does.
 *
 * Parity status: PINNED — checked against the reference's own golden tables
 * (roaring/roaring_internal_test.go TestContainerCombinations and the per-kernel table
 * tests, extracted by tests/golden/extract_go_tables.py into JSON fixtures under tests/golden).
 * The Go toolchain is absent, so the reference itself cannot be run here.
 */
#ifndef ROARING_ORACLE_H
#define ROARING_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* roaring/roaring.go:53-58 */
#define ORC_NIL 0
#define ORC_ARRAY 1
#define ORC_BITMAP 2
#define ORC_RUN 3

#define ORC_ARRAY_MAX_SIZE 4096 /* roaring.go:3036 */
#define ORC_RUN_MAX_SIZE 2048   /* roaring.go:3039 */
#define ORC_BITMAP_N 1024       /* roaring.go:44 */
#define ORC_MAX_CONTAINER_VAL 0xffff

typedef struct orc_interval16 { /* roaring.go:3041-3044 */
  uint16_t start;
  uint16_t last;
} orc_interval16;

/* roaring/container_stash.go:46-53 (no frozen/mapped/dirty flags: every result here is a
 * fresh allocation, which has the same bit content as Go's shared frozen containers). */
typedef struct orc_container {
  uint8_t typ;
  int32_t n;   /* cardinality */
  int32_t len; /* array: #elements; run: #intervals; bitmap: 1024 */
  void* data;  /* uint16_t[] / uint64_t[1024] / orc_interval16[] */
} orc_container;

/* ---- construction / access (container_stash.go NewContainer*) ---- */
orc_container* orc_new_array(const uint16_t* v, int32_t n);
orc_container* orc_new_bitmap(const uint64_t* words, int32_t n /* -1: count */);
orc_container* orc_new_run(const orc_interval16* r, int32_t len);
orc_container* orc_clone(const orc_container* c);
void orc_free(orc_container* c);
int32_t orc_n(const orc_container* c); /* Container.N, nil -> 0 (container_stash.go:430) */
int32_t orc_typ(const orc_container* c);
int32_t orc_len(const orc_container* c);
const void* orc_data(const orc_container* c);
/* bit content as 1024 words, whatever the encoding */
void orc_to_words(const orc_container* c, uint64_t* out1024);

/* ---- counting (roaring.go:3052-3233, 3368-3408) ---- */
int32_t orc_count(const orc_container* c);
int32_t orc_count_range(const orc_container* c, int32_t start, int32_t end);
int32_t orc_array_count_range(const uint16_t* a, int32_t len, int32_t start, int32_t end);
int32_t orc_words_count_range(const uint64_t* bm, int32_t start, int32_t end);
int32_t orc_run_count_range(const orc_interval16* r, int32_t len, int32_t start, int32_t end);
int32_t orc_count_runs(const orc_container* c);

/* ---- conversions and policy (roaring.go:3412-3461, 3687-4018) ---- */
orc_container* orc_optimize(const orc_container* c); /* NULL when empty */
orc_container* orc_array_to_bitmap(const orc_container* c);
orc_container* orc_bitmap_to_array(const orc_container* c);
orc_container* orc_run_to_bitmap(const orc_container* c);
orc_container* orc_bitmap_to_run(const orc_container* c);
orc_container* orc_array_to_run(const orc_container* c);
orc_container* orc_run_to_array(const orc_container* c);

/* ---- bitmap range helpers (roaring.go:5237-5330); mutate c (a bitmap container) ---- */
void orc_bitmap_set_range(orc_container* c, uint64_t i, uint64_t j);
void orc_bitmap_xor_range(orc_container* c, uint64_t i, uint64_t j);
void orc_bitmap_zero_range(orc_container* c, uint64_t i, uint64_t j);

/* ---- container dispatchers (roaring.go:4477, 4753, 4980, 5692, 6052) ---- */
int32_t orc_intersection_count(const orc_container* a, const orc_container* b);
orc_container* orc_intersect(const orc_container* a, const orc_container* b);
orc_container* orc_union(const orc_container* a, const orc_container* b);
orc_container* orc_difference(const orc_container* a, const orc_container* b);
orc_container* orc_xor(const orc_container* a, const orc_container* b);
/* Container.unionInPlace semantics (roaring.go:3470-3525) returning a fresh container
 * with a valid n (i.e. after Repair) */
orc_container* orc_union_in_place(const orc_container* c, const orc_container* other);
/* BitwiseCompare (roaring.go:5396): 0 when the bit content is equal */
int32_t orc_bitwise_compare(const orc_container* a, const orc_container* b);

/* ---- Bitmap = sorted keys + containers (roaring.go:232-248, containers_slice.go:5-10) ---- */
typedef struct orc_bitmap {
  int32_t len, cap;
  uint64_t* keys;
  orc_container** cs; /* entries may be NULL (nil containers are legal, containers_slice.go:238) */
} orc_bitmap;

orc_bitmap* orc_bitmap_new(void);
void orc_bitmap_free(orc_bitmap* b);
/* append with key > last key; takes ownership of c (may be NULL) */
void orc_bitmap_put(orc_bitmap* b, uint64_t key, orc_container* c);
int32_t orc_bitmap_len(const orc_bitmap* b);
uint64_t orc_bitmap_key(const orc_bitmap* b, int32_t i);
const orc_container* orc_bitmap_container(const orc_bitmap* b, int32_t i);

uint64_t orc_bitmap_count(const orc_bitmap* b);                                 /* roaring.go:542 */
uint64_t orc_bitmap_count_range(const orc_bitmap* b, uint64_t s, uint64_t e);   /* roaring.go:573 */
uint64_t orc_bitmap_intersection_count(const orc_bitmap* a, const orc_bitmap* b); /* :711 */
orc_bitmap* orc_bitmap_intersect(const orc_bitmap* a, const orc_bitmap* b);     /* :736 */
orc_bitmap* orc_bitmap_union(const orc_bitmap* a, const orc_bitmap* const* others, int32_t n_others); /* :1272 */
orc_bitmap* orc_bitmap_difference(const orc_bitmap* a, const orc_bitmap* const* others, int32_t n_others); /* :1564 */
orc_bitmap* orc_bitmap_xor(const orc_bitmap* a, const orc_bitmap* b);           /* :1598 */

/* ---- bulk helper for the CPU baseline: |A∩B| over n_pairs dense rows of 16 bitmap
 * containers each (Bitmap.IntersectionCount -> popcountAndSlice, roaring.go:711,6928).
 * rows are 16*1024 words each; returns the total and writes per-pair counts. */
uint64_t orc_dense_intersection_count(const uint64_t* a, const uint64_t* b, uint64_t n_pairs, uint64_t* out_counts);
/* materialising variant: Bitmap.Intersect -> intersectBitmapBitmap (roaring.go:4960) then
 * Count; writes the 16*1024-word result rows */
uint64_t orc_dense_intersect_count(const uint64_t* a, const uint64_t* b, uint64_t n_pairs, uint64_t* out_rows,
                                   uint64_t* out_counts);

/* ---- BSI / TopK / GroupBy / UnionRows (bsi_oracle.c).  A fragment is an array of rows;
 * row r is an orc_bitmap with container keys 0..15 (NULL = no containers).  BSI layout
 * (fragment.go:62-65): row 0 exists, row 1 sign, row 2+i magnitude bit i. ---- */
#define ORC_EQ 1
#define ORC_NEQ 2
#define ORC_LT 3
#define ORC_LTE 4
#define ORC_GT 5
#define ORC_GTE 6
/* fragment.sum (fragment.go:724) via BitmapBSICountFilter (roaring/filter.go:1097-1218) */
void orc_bsi_sum(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                 int64_t* out_sum, uint64_t* out_count);
/* fragment.rangeOp (fragment.go:937) / rangeBetween (:1213) */
orc_bitmap* orc_bsi_range(const orc_bitmap* const* rows, int32_t n_rows, int32_t op, uint64_t bit_depth,
                          int64_t predicate);
orc_bitmap* orc_bsi_range_between(const orc_bitmap* const* rows, int32_t n_rows, uint64_t bit_depth, int64_t pmin,
                                  int64_t pmax);
orc_bitmap* orc_bsi_range_lt_unsigned(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter,
                                      uint64_t bit_depth, uint64_t predicate, int32_t allow_eq);
orc_bitmap* orc_bsi_range_gt_unsigned(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter,
                                      uint64_t bit_depth, uint64_t predicate, int32_t allow_eq);
orc_bitmap* orc_bsi_range_between_unsigned(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter,
                                           uint64_t bit_depth, uint64_t pmin, uint64_t pmax);
/* doTopK (executor.go:2705): per-row |row ∩ filter| */
void orc_topk_row_counts(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                         uint64_t* out_counts);
/* groupByIterator (executor.go:8880): out[i*nb+j] = |(A_i ∩ F) ∩ B_j| */
void orc_groupby_counts(const orc_bitmap* const* a_rows, int32_t na, const orc_bitmap* const* b_rows, int32_t nb,
                        const orc_bitmap* filter, int32_t has_filter, uint64_t* out);
/* BitmapRowsUnion (roaring/filter.go:294) */
orc_bitmap* orc_union_rows(const orc_bitmap* const* rows, int32_t n_rows);

/* ---- serialisation (wire_oracle.c): Bitmap.WriteTo (roaring.go:1730-1817) and
 * UnmarshalBinary through NewRoaringIterator (:1945-2262), Pilosa and official formats ---- */
uint8_t* orc_roaring_marshal(const orc_bitmap* b, int32_t optimize, uint64_t* out_len);
void orc_wire_free(uint8_t* p);
orc_bitmap* orc_roaring_unmarshal(const uint8_t* data, uint64_t len, int32_t* err);

/* every worker thread makes `passes` passes over its own chunk of row pairs */
uint64_t orc_dense_intersection_count_mt(const uint64_t* a, const uint64_t* b, uint64_t n_pairs, uint64_t* out_counts,
                                         int32_t n_threads, uint64_t passes);

/* ---- test hooks (the reference's per-kernel tests call the type-pair kernels directly) */
void orc_set_n(orc_container* c, int32_t n);
orc_container* orc_flip(const orc_container* a); /* roaring.go:4221 */
int32_t orc_run_append_interval(const orc_interval16* base, int32_t len, orc_interval16 v);
/* name = Go kernel name, e.g. "intersectRunRun" (roaring.go:4835) */
orc_container* orc_kernel(const char* name, const orc_container* a, const orc_container* b);
int32_t orc_count_kernel(const char* name, const orc_container* a, const orc_container* b);

#ifdef __cplusplus
}
#endif
#endif
