/*
 * batch_oracle.c — CPU ORACLE, batch entry points (test infrastructure, NOT product code).
 *
 * The per-container / per-bitmap functions of roaring_oracle.c and bsi_oracle.c restate the
 * reference function by function.  This file adds nothing to the algorithm: it only (1) builds
 * the oracle's Bitmaps straight from the FLATTENED form fbk_batch_upload takes (descriptor table +
 * one payload buffer, include/fbk.h fbk_container_desc) or from dense words, so that no Python
 * object is made per container, and (2) runs the restated reference calls over all shards of a
 * BASELINE.json configuration on a pool of pthreads — the reference itself runs one goroutine per
 * shard over NumCPU pool workers (executor.go:6723-6737, mapperLocal :6742).
 *
 * What each entry point calls, and the reference call it stands for:
 *   orc_batch_intersection_count   orc_bitmap_intersection_count   Bitmap.IntersectionCount   roaring.go:711-733
 *   orc_batch_setop                orc_bitmap_intersect/union/...  Bitmap.Intersect/Union/Difference/Xor :736,1272,1564,1598
 *   orc_batch_union_n(_icount)     orc_bitmap_union(k-1 others)    Bitmap.Union (n-way)       roaring.go:1272-1284, 1410-1561
 *   orc_batch_topk_counts          orc_topk_row_counts             doTopK / topKFilter        executor.go:2705-2774
 *   orc_batch_count_matrix         orc_groupby_counts              groupByIterator.Next       executor.go:8880-8934
 *   orc_batch_bsi_range/_between   orc_bsi_range / _range_between  fragment.rangeOp / rangeBetween  fragment.go:937-1303
 *   orc_batch_bsi_sum              orc_bsi_sum                     fragment.sum, BitmapBSICountFilter  fragment.go:724, filter.go:1097-1218
 *   orc_batch_bsi_minmax           orc_bsi_min / orc_bsi_max       fragment.min / max         fragment.go:754-853
 *
 * Rows of a rowset carry container keys 0..15 (key & 15): fragment.row rebases every row of a
 * shard to the same key range (fragment.go:318), so two rows of one shard always have comparable
 * keys — which rows belong to one shard is the caller's row-index lists, as in the C ABI.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "roaring_oracle.h"

/* bsi_oracle.c (not all of them are in the public header) */
void orc_bsi_min(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                 uint64_t bit_depth, int64_t* out_min, uint64_t* out_count);
void orc_bsi_max(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                 uint64_t bit_depth, int64_t* out_max, uint64_t* out_count);

/* == fbk_container_desc (include/fbk.h), restated here so that the oracle includes nothing of the product */
typedef struct orc_flat_desc {
  uint64_t key;
  uint64_t off;
  uint32_t row;
  uint32_t len;
  int32_t n;
  uint8_t type;
  uint8_t pad[3];
} orc_flat_desc;

typedef struct orc_rowset {
  uint32_t n_rows;
  orc_bitmap** rows; /* never NULL entries: a row without containers is an empty Bitmap */
} orc_rowset;

static void* xm(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) abort();
  return p;
}

static orc_rowset* rowset_new(uint32_t n_rows) {
  orc_rowset* rs = (orc_rowset*)xm(sizeof(*rs));
  rs->n_rows = n_rows;
  rs->rows = (orc_bitmap**)xm((size_t)n_rows * sizeof(orc_bitmap*));
  for (uint32_t i = 0; i < n_rows; i++) rs->rows[i] = NULL;
  return rs;
}

void orc_rowset_free(orc_rowset* rs) {
  if (!rs) return;
  for (uint32_t i = 0; i < rs->n_rows; i++) orc_bitmap_free(rs->rows[i]);
  free(rs->rows);
  free(rs);
}

uint32_t orc_rowset_rows(const orc_rowset* rs) { return rs->n_rows; }

/* ---- a pool of pthreads over [0, n): work items handed out one at a time -------------------- */
typedef void (*item_fn)(uint64_t i, void* arg);
typedef struct {
  uint64_t n;
  uint64_t next;
  item_fn fn;
  void* arg;
} pool_job;

static void* pool_worker(void* p) {
  pool_job* j = (pool_job*)p;
  for (;;) {
    uint64_t i = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
    if (i >= j->n) return NULL;
    j->fn(i, j->arg);
  }
}

static void parallel_for(uint64_t n, int32_t threads, item_fn fn, void* arg) {
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > n) threads = (int32_t)(n ? n : 1);
  pool_job j = {n, 0, fn, arg};
  if (threads == 1) {
    pool_worker(&j);
    return;
  }
  pthread_t* th = (pthread_t*)xm((size_t)threads * sizeof(pthread_t));
  for (int32_t t = 0; t < threads; t++) pthread_create(&th[t], NULL, pool_worker, &j);
  for (int32_t t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th);
}

typedef struct {
  const orc_flat_desc* d;
  const uint8_t* payload;
  orc_container** cell;
  const uint64_t* words;
  orc_rowset* rs;
} build_job;

static void flat_item(uint64_t i, void* a) {
  build_job* j = (build_job*)a;
  const orc_flat_desc* e = &j->d[i];
  uint64_t bytes = e->type == ORC_ARRAY ? 2ull * e->len : e->type == ORC_RUN ? 4ull * e->len : 8192ull;
  void* tmp = xm(bytes); /* payload offsets are only byte aligned in general */
  memcpy(tmp, j->payload + e->off, bytes);
  orc_container* c = e->type == ORC_ARRAY ? orc_new_array((const uint16_t*)tmp, (int32_t)e->len)
                     : e->type == ORC_RUN ? orc_new_run((const orc_interval16*)tmp, (int32_t)e->len)
                                          : orc_new_bitmap((const uint64_t*)tmp, -1);
  free(tmp);
  j->cell[(size_t)e->row * 16 + (e->key & 15u)] = c;
}

/* any descriptor order (fbk_batch_upload places a container by its row and key & 15); returns NULL on a
 * malformed table: row or payload out of bounds, unknown type, two containers for one (row, slot) */
orc_rowset* orc_rowset_from_flat(const orc_flat_desc* d, uint64_t n_desc, uint32_t n_rows, const uint8_t* payload,
                                 uint64_t payload_len, int32_t threads) {
  orc_container** cell = (orc_container**)xm((size_t)n_rows * 16 * sizeof(*cell));
  for (size_t i = 0; i < (size_t)n_rows * 16; i++) cell[i] = NULL;
  for (uint64_t i = 0; i < n_desc; i++) {
    const orc_flat_desc* e = &d[i];
    uint64_t bytes = e->type == ORC_ARRAY ? 2ull * e->len : e->type == ORC_RUN ? 4ull * e->len : 8192ull;
    if (e->row >= n_rows || e->off > payload_len || bytes > payload_len - e->off || e->type < ORC_ARRAY || e->type > ORC_RUN ||
        cell[(size_t)e->row * 16 + (e->key & 15u)]) {
      free(cell);
      return NULL;
    }
    cell[(size_t)e->row * 16 + (e->key & 15u)] = (orc_container*)(uintptr_t)1; /* taken */
  }
  build_job j = {d, payload, cell, NULL, NULL};
  parallel_for(n_desc, threads, flat_item, &j);
  orc_rowset* rs = rowset_new(n_rows);
  for (uint32_t r = 0; r < n_rows; r++) {
    orc_bitmap* b = orc_bitmap_new();
    for (int sl = 0; sl < 16; sl++)
      if (cell[(size_t)r * 16 + sl]) orc_bitmap_put(b, (uint64_t)sl, cell[(size_t)r * 16 + sl]);
    rs->rows[r] = b;
  }
  free(cell);
  return rs;
}

static void dense_item(uint64_t r, void* a) {
  build_job* j = (build_job*)a;
  orc_bitmap* b = orc_bitmap_new();
  for (int s = 0; s < 16; s++) {
    orc_container* c = orc_new_bitmap(j->words + ((size_t)r * 16 + s) * ORC_BITMAP_N, -1);
    /* an all-zero container is stored as nil (roaring.go:751-752) */
    if (orc_n(c) == 0) {
      orc_free(c);
      continue;
    }
    orc_bitmap_put(b, (uint64_t)s, c);
  }
  j->rs->rows[r] = b;
}

/* n_rows rows of 16 bitmap containers (fbk_batch_upload_dense's input); cardinalities counted */
orc_rowset* orc_rowset_from_dense(const uint64_t* words, uint32_t n_rows, int32_t threads) {
  orc_rowset* rs = rowset_new(n_rows);
  build_job j = {NULL, NULL, NULL, words, rs};
  parallel_for(n_rows, threads, dense_item, &j);
  return rs;
}

/* bit content of one row as 16 x 1024 words */
void orc_rowset_row_words(const orc_rowset* rs, uint32_t row, uint64_t* out) {
  memset(out, 0, 16 * ORC_BITMAP_N * 8);
  const orc_bitmap* b = rs->rows[row];
  for (int32_t i = 0; i < b->len; i++)
    if (b->cs[i]) orc_to_words(b->cs[i], out + (b->keys[i] & 15u) * ORC_BITMAP_N);
}

/* all rows of the set written out at once: n_rows x 16 x 1024 words */
typedef struct {
  const orc_rowset* rs;
  uint64_t* out;
} words_job;
static void words_item(uint64_t i, void* a) {
  words_job* j = (words_job*)a;
  orc_rowset_row_words(j->rs, (uint32_t)i, j->out + i * 16 * ORC_BITMAP_N);
}
void orc_rowset_words(const orc_rowset* rs, uint64_t* out, int32_t threads) {
  words_job j = {rs, out};
  parallel_for(rs->n_rows, threads, words_item, &j);
}

/* Bitmap.Count of every listed row (roaring.go:542) */
void orc_rowset_counts(const orc_rowset* rs, const uint32_t* rows, uint64_t n, uint64_t* out) {
  for (uint64_t i = 0; i < n; i++) out[i] = orc_bitmap_count(rs->rows[rows ? rows[i] : i]);
}

/* ---- pair operations ------------------------------------------------------------------------- */
typedef struct {
  const orc_rowset *A, *B;
  const uint32_t *ra, *rb;
  uint64_t* out;
  int32_t op;
  orc_rowset* res;
} pair_job;

static void icount_item(uint64_t i, void* a) {
  pair_job* j = (pair_job*)a;
  j->out[i] = orc_bitmap_intersection_count(j->A->rows[j->ra[i]], j->B->rows[j->rb[i]]);
}

void orc_batch_intersection_count(const orc_rowset* A, const uint32_t* ra, const orc_rowset* B, const uint32_t* rb, uint64_t n_pairs,
                                  uint64_t* out, int32_t threads) {
  pair_job j = {A, B, ra, rb, out, 0, NULL};
  parallel_for(n_pairs, threads, icount_item, &j);
}

/* the same, `passes` times over, on ONE pool of threads (bench.py's CPU leg: thread start-up must not be what is timed);
 * item i is pair i mod n_pairs, so every pass walks every shard once and consecutive items stream through distinct rows */
typedef struct {
  pair_job j;
  uint64_t n_pairs;
} repeat_job;
static void icount_repeat_item(uint64_t i, void* a) {
  repeat_job* r = (repeat_job*)a;
  icount_item(i % r->n_pairs, &r->j);
}
void orc_batch_intersection_count_repeat(const orc_rowset* A, const uint32_t* ra, const orc_rowset* B, const uint32_t* rb, uint64_t n_pairs,
                                         uint64_t* out, int32_t threads, uint64_t passes) {
  repeat_job r = {{A, B, ra, rb, out, 0, NULL}, n_pairs};
  parallel_for(n_pairs * passes, threads, icount_repeat_item, &r);
}

static void setop_item(uint64_t i, void* a) {
  pair_job* j = (pair_job*)a;
  const orc_bitmap* x = j->A->rows[j->ra[i]];
  const orc_bitmap* y = j->B->rows[j->rb[i]];
  const orc_bitmap* others[1] = {y};
  orc_bitmap* r;
  switch (j->op) { /* FBK_OP_* numbering: 0 AND, 1 OR, 2 XOR, 3 ANDNOT */
    case 0: r = orc_bitmap_intersect(x, y); break;
    case 1: r = orc_bitmap_union(x, others, 1); break;
    case 2: r = orc_bitmap_xor(x, y); break;
    default: r = orc_bitmap_difference(x, others, 1); break;
  }
  j->res->rows[i] = r;
  if (j->out) j->out[i] = orc_bitmap_count(r);
}

orc_rowset* orc_batch_setop(int32_t op, const orc_rowset* A, const uint32_t* ra, const orc_rowset* B, const uint32_t* rb, uint64_t n_pairs,
                            uint64_t* out_counts, int32_t threads) {
  orc_rowset* res = rowset_new((uint32_t)n_pairs);
  pair_job j = {A, B, ra, rb, out_counts, op, res};
  parallel_for(n_pairs, threads, setop_item, &j);
  return res;
}

/* ---- n-way union (config 3) ------------------------------------------------------------------ */
typedef struct {
  const orc_rowset *A, *F;
  const uint32_t *groups, *frows;
  uint32_t k;
  uint64_t *out, *out_union_count;
  orc_rowset* res;
} union_job;

static orc_bitmap* union_group(const union_job* j, uint64_t g) {
  const uint32_t* rows = j->groups + g * j->k;
  const orc_bitmap** others = (const orc_bitmap**)xm((size_t)j->k * sizeof(*others));
  for (uint32_t r = 1; r < j->k; r++) others[r - 1] = j->A->rows[rows[r]];
  orc_bitmap* u = orc_bitmap_union(j->A->rows[rows[0]], others, (int32_t)j->k - 1);
  free(others);
  return u;
}

static void union_item(uint64_t g, void* a) {
  union_job* j = (union_job*)a;
  orc_bitmap* u = union_group(j, g);
  if (j->out_union_count) j->out_union_count[g] = orc_bitmap_count(u);
  if (j->F) j->out[g] = orc_bitmap_intersection_count(u, j->F->rows[j->frows[g]]);
  if (j->res) j->res->rows[g] = u;
  else orc_bitmap_free(u);
}

/* out[g] = |(rows[g][0] ∪ ... ∪ rows[g][k-1]) ∩ F[frows[g]]| */
void orc_batch_union_n_icount(const orc_rowset* A, const uint32_t* groups, uint64_t n_groups, uint32_t k, const orc_rowset* F,
                              const uint32_t* frows, uint64_t* out, uint64_t* out_union_count, int32_t threads) {
  union_job j = {A, F, groups, frows, k, out, out_union_count, NULL};
  parallel_for(n_groups, threads, union_item, &j);
}

orc_rowset* orc_batch_union_n(const orc_rowset* A, const uint32_t* groups, uint64_t n_groups, uint32_t k, uint64_t* out_union_count,
                              int32_t threads) {
  orc_rowset* res = rowset_new((uint32_t)n_groups);
  union_job j = {A, NULL, groups, NULL, k, NULL, out_union_count, res};
  parallel_for(n_groups, threads, union_item, &j);
  return res;
}

/* ---- TopK / GroupBy counting (configs 3, 4) -------------------------------------------------- */
typedef struct {
  const orc_rowset *A, *B, *F;
  const uint32_t *ra, *rb, *frows;
  uint32_t na, nb;
  uint64_t* out;
} matrix_job;

static const orc_bitmap** gather(const orc_rowset* rs, const uint32_t* idx, uint32_t n) {
  const orc_bitmap** v = (const orc_bitmap**)xm((size_t)n * sizeof(*v));
  for (uint32_t i = 0; i < n; i++) v[i] = rs->rows[idx[i]];
  return v;
}

static void matrix_item(uint64_t s, void* a) {
  matrix_job* j = (matrix_job*)a;
  const orc_bitmap** ar = gather(j->A, j->ra + s * j->na, j->na);
  const orc_bitmap** br = gather(j->B, j->rb + s * j->nb, j->nb);
  orc_groupby_counts(ar, (int32_t)j->na, br, (int32_t)j->nb, j->F ? j->F->rows[j->frows[s]] : NULL, j->F != NULL,
                     j->out + s * (uint64_t)j->na * j->nb);
  free(ar);
  free(br);
}

/* out[s][i][j] = |(A[ra[s][i]] ∩ F[frows[s]]) ∩ B[rb[s][j]]| for every shard s (F == NULL: no filter) */
void orc_batch_count_matrix(const orc_rowset* A, const uint32_t* ra, uint32_t na, const orc_rowset* B, const uint32_t* rb, uint32_t nb,
                            const orc_rowset* F, const uint32_t* frows, uint64_t n_shards, uint64_t* out, int32_t threads) {
  matrix_job j = {A, B, F, ra, rb, frows, na, nb, out};
  parallel_for(n_shards, threads, matrix_item, &j);
}

static void topk_item(uint64_t s, void* a) {
  matrix_job* j = (matrix_job*)a;
  const orc_bitmap** ar = gather(j->A, j->ra + s * j->na, j->na);
  orc_topk_row_counts(ar, (int32_t)j->na, j->F ? j->F->rows[j->frows[s]] : NULL, j->F != NULL, j->out + s * (uint64_t)j->na);
  free(ar);
}

/* out[s][r] = |A[ra[s][r]] ∩ F[frows[s]]| (doTopK's per-row counts) */
void orc_batch_topk_counts(const orc_rowset* A, const uint32_t* ra, uint32_t k, const orc_rowset* F, const uint32_t* frows,
                           uint64_t n_shards, uint64_t* out, int32_t threads) {
  matrix_job j = {A, NULL, F, ra, NULL, frows, k, 0, out};
  parallel_for(n_shards, threads, topk_item, &j);
}

/* ---- BSI (config 5): the fragment of shard s is rows base[s] .. base[s] + depth + 1 ----------- */
typedef struct {
  const orc_rowset *A, *F;
  const uint32_t *base, *frows;
  uint32_t depth;
  int32_t op;
  int64_t p0, p1;
  int64_t* out_val;
  uint64_t* out_count;
  orc_rowset* res;
} bsi_job;

static void bsi_range_item(uint64_t s, void* a) {
  bsi_job* j = (bsi_job*)a;
  const orc_bitmap* const* rows = (const orc_bitmap* const*)(j->A->rows + j->base[s]);
  int32_t n_rows = (int32_t)j->depth + 2;
  orc_bitmap* r = j->op == 0 ? orc_bsi_range_between(rows, n_rows, j->depth, j->p0, j->p1) : orc_bsi_range(rows, n_rows, j->op, j->depth, j->p0);
  if (!r) r = orc_bitmap_new();
  if (j->out_count) j->out_count[s] = orc_bitmap_count(r);
  j->res->rows[s] = r;
}

/* op = ORC_EQ..ORC_GTE: fragment.rangeOp(op, depth, predicate); op = 0: fragment.rangeBetween(depth, predicate, predicate2) */
orc_rowset* orc_batch_bsi_range(const orc_rowset* A, const uint32_t* base, uint64_t n_shards, uint32_t depth, int32_t op, int64_t predicate,
                                int64_t predicate2, uint64_t* out_count, int32_t threads) {
  orc_rowset* res = rowset_new((uint32_t)n_shards);
  bsi_job j = {A, NULL, base, NULL, depth, op, predicate, predicate2, NULL, out_count, res};
  parallel_for(n_shards, threads, bsi_range_item, &j);
  return res;
}

static void bsi_sum_item(uint64_t s, void* a) {
  bsi_job* j = (bsi_job*)a;
  const orc_bitmap* const* rows = (const orc_bitmap* const*)(j->A->rows + j->base[s]);
  orc_bsi_sum(rows, (int32_t)j->depth + 2, j->F ? j->F->rows[j->frows[s]] : NULL, j->F != NULL, &j->out_val[s], &j->out_count[s]);
}

void orc_batch_bsi_sum(const orc_rowset* A, const uint32_t* base, uint64_t n_shards, uint32_t depth, const orc_rowset* F, const uint32_t* frows,
                       int64_t* out_sum, uint64_t* out_count, int32_t threads) {
  bsi_job j = {A, F, base, frows, depth, 0, 0, 0, out_sum, out_count, NULL};
  parallel_for(n_shards, threads, bsi_sum_item, &j);
}

static void bsi_minmax_item(uint64_t s, void* a) {
  bsi_job* j = (bsi_job*)a;
  const orc_bitmap* const* rows = (const orc_bitmap* const*)(j->A->rows + j->base[s]);
  const orc_bitmap* f = j->F ? j->F->rows[j->frows[s]] : NULL;
  if (j->op) orc_bsi_max(rows, (int32_t)j->depth + 2, f, j->F != NULL, j->depth, &j->out_val[s], &j->out_count[s]);
  else orc_bsi_min(rows, (int32_t)j->depth + 2, f, j->F != NULL, j->depth, &j->out_val[s], &j->out_count[s]);
}

/* is_max = 0: fragment.min, 1: fragment.max, per shard */
void orc_batch_bsi_minmax(const orc_rowset* A, const uint32_t* base, uint64_t n_shards, uint32_t depth, int32_t is_max, const orc_rowset* F,
                          const uint32_t* frows, int64_t* out_val, uint64_t* out_count, int32_t threads) {
  bsi_job j = {A, F, base, frows, depth, is_max, 0, 0, out_val, out_count, NULL};
  parallel_for(n_shards, threads, bsi_minmax_item, &j);
}
