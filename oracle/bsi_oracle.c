/*
 * bsi_oracle.c — CPU ORACLE (test infrastructure, NOT product code): BSI Sum / Range,
 * TopK row counts, GroupBy count matrix and UnionRows, restated from fragment.go,
 * roaring/filter.go and executor.go on top of the container algebra in
 * roaring_oracle.c.  Each function cites the reference lines it follows.
 *
 * A *fragment* here is an array of rows; row r is an orc_bitmap whose container keys are
 * the slot numbers 0..15 (keys are coerced with & 15 exactly as roaring/filter.go:749,1193
 * and executor.go:2767 do).  A NULL row pointer is a row with no containers.
 * BSI layout (fragment.go:62-65): row 0 = exists, row 1 = sign, row 2+i = magnitude bit i.
 */
#include <stdlib.h>
#include <string.h>

#include "roaring_oracle.h"

#define ROW_WIDTH 16 /* containers per shard row: 1<<(shardwidth.Exponent-16), roaring/filter.go:23-28 */

static inline uint64_t go_shl64(uint64_t x, uint64_t s) { return s >= 64 ? 0 : x << s; }
static inline uint64_t go_shr64(uint64_t x, uint64_t s) { return s >= 64 ? 0 : x >> s; }
static inline uint64_t bits_len64(uint64_t x) { return x ? 64 - (uint64_t)__builtin_clzll(x) : 0; }

/* ---- Row helpers (row.go) over one shard segment -------------------------------------------- */

static orc_bitmap* row_new(void) { return orc_bitmap_new(); } /* NewRow(), row.go:36 */

static orc_bitmap* row_clone(const orc_bitmap* r) {
  orc_bitmap* o = orc_bitmap_new();
  if (r)
    for (int32_t i = 0; i < r->len; i++) orc_bitmap_put(o, r->keys[i], orc_clone(r->cs[i]));
  return o;
}
/* Row.Intersect, row.go:242 -> RowSegment.Intersect :561 -> Bitmap.Intersect */
static orc_bitmap* row_intersect(const orc_bitmap* a, const orc_bitmap* b) {
  orc_bitmap *ea = NULL, *eb = NULL;
  if (!a) a = ea = orc_bitmap_new();
  if (!b) b = eb = orc_bitmap_new();
  orc_bitmap* o = orc_bitmap_intersect(a, b);
  orc_bitmap_free(ea);
  orc_bitmap_free(eb);
  return o;
}
/* Row.Difference, row.go:333 -> RowSegment.Difference :585 -> Bitmap.Difference */
static orc_bitmap* row_difference(const orc_bitmap* a, const orc_bitmap* b) {
  orc_bitmap *ea = NULL, *eb = NULL;
  if (!a) a = ea = orc_bitmap_new();
  if (!b) b = eb = orc_bitmap_new();
  const orc_bitmap* others[1] = {b};
  orc_bitmap* o = orc_bitmap_difference(a, others, 1);
  orc_bitmap_free(ea);
  orc_bitmap_free(eb);
  return o;
}
/* Row.Union with one other, row.go:288 -> RowSegment.Union :572 -> Bitmap.Union(1 other) */
static orc_bitmap* row_union(const orc_bitmap* a, const orc_bitmap* b) {
  orc_bitmap *ea = NULL, *eb = NULL;
  if (!a) a = ea = orc_bitmap_new();
  if (!b) b = eb = orc_bitmap_new();
  const orc_bitmap* others[1] = {b};
  orc_bitmap* o = orc_bitmap_union(a, others, 1);
  orc_bitmap_free(ea);
  orc_bitmap_free(eb);
  return o;
}
/* Row.Any, row.go:257 */
static int row_any(const orc_bitmap* r) {
  if (!r) return 0;
  for (int32_t i = 0; i < r->len; i++)
    if (orc_n(r->cs[i]) > 0) return 1;
  return 0;
}

typedef struct {
  const orc_bitmap* const* rows;
  int32_t n_rows;
} frag;

/* fragment.row(tx, rowID), fragment.go:283: a missing row is an empty Row */
static const orc_bitmap* frag_row(const frag* f, uint64_t row) {
  if (row >= (uint64_t)f->n_rows) return NULL;
  return f->rows[row];
}

/* ---- BSI Sum: fragment.sum (fragment.go:724-750) driving BitmapBSICountFilter
 * (roaring/filter.go:1097-1218) through Tx.ApplyFilter, which streams the fragment's
 * non-empty containers in key order = (row, slot) order (roaring/filter.go:1062-1085).
 * has_filter == 0 is "no filter" (every position considered, filter.go:1170-1176);
 * has_filter != 0 with filter == NULL or an empty bitmap is "filter has no contents for
 * this shard" -> (0, 0) (fragment.go:736-738). */
void orc_bsi_sum(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                 int64_t* out_sum, uint64_t* out_count) {
  *out_sum = 0;
  *out_count = 0;
  orc_container* containers[ROW_WIDTH];
  orc_container* positive[ROW_WIDTH];
  orc_container* negative[ROW_WIDTH];
  memset(containers, 0, sizeof(containers));
  memset(positive, 0, sizeof(positive));
  memset(negative, 0, sizeof(negative));
  if (has_filter) {
    if (!filter) return;
    for (int32_t i = 0; i < filter->len; i++)
      if (filter->cs[i]) containers[filter->keys[i] & (ROW_WIDTH - 1)] = orc_clone(filter->cs[i]);
  } else {
    orc_interval16 full = {0, 0xffff};
    for (int i = 0; i < ROW_WIDTH; i++) containers[i] = orc_new_run(&full, 1);
  }
  int32_t count = 0;
  uint64_t psum = 0, nsum = 0;
  frag f = {rows, n_rows};
  for (uint64_t row = 0; row < (uint64_t)n_rows; row++) {
    const orc_bitmap* r = frag_row(&f, row);
    if (!r) continue;
    for (int32_t ci = 0; ci < r->len; ci++) {
      const orc_container* data = r->cs[ci];
      uint64_t pos = r->keys[ci] & (ROW_WIDTH - 1);
      /* ConsiderKey, filter.go:1110-1116: skipped when the filter has no container at
       * this offset or the data container is empty */
      if (containers[pos] == NULL || orc_n(data) == 0) continue;
      if (row == 0) { /* existence bit, filter.go:1135-1141 */
        orc_free(positive[pos]);
        positive[pos] = orc_intersect(containers[pos], data);
        count += orc_n(positive[pos]);
      } else if (row == 1) { /* sign bit, filter.go:1142-1151 */
        orc_free(negative[pos]);
        negative[pos] = orc_intersect(positive[pos], data);
        orc_container* np = orc_difference(positive[pos], data);
        orc_free(positive[pos]);
        positive[pos] = np;
      } else { /* value rows, filter.go:1157-1160: uint64 wrap-around sums */
        uint64_t pc = (uint64_t)orc_intersection_count(positive[pos], data);
        uint64_t nc = (uint64_t)orc_intersection_count(negative[pos], data);
        psum += go_shl64(pc, row - 2);
        nsum += go_shl64(nc, row - 2);
      }
    }
  }
  for (int i = 0; i < ROW_WIDTH; i++) {
    orc_free(containers[i]);
    orc_free(positive[i]);
    orc_free(negative[i]);
  }
  /* Total(), filter.go:1106-1108; fragment.sum returns uint64(c32), fragment.go:747-749 */
  *out_sum = (int64_t)psum - (int64_t)nsum;
  *out_count = (uint64_t)count;
}

/* ---- BSI Range ------------------------------------------------------------------------------ */

#define BSI_EXISTS 0
#define BSI_SIGN 1
#define BSI_OFFSET 2

/* pql tokens used by rangeOp (fragment.go:937-950) */
#define ORC_EQ 1
#define ORC_NEQ 2
#define ORC_LT 3
#define ORC_LTE 4
#define ORC_GT 5
#define ORC_GTE 6

/* absInt64, fragment.go:952-961 */
static uint64_t abs_int64(int64_t v) {
  if (v > 0) return (uint64_t)v;
  if (v == INT64_MIN) return 9223372036854775808ull;
  return (uint64_t)(-v);
}

/* rangeEQ, fragment.go:963-1003 */
static orc_bitmap* range_eq(const frag* f, uint64_t bit_depth, int64_t predicate) {
  orc_bitmap* b = row_clone(frag_row(f, BSI_EXISTS));
  uint64_t up = abs_int64(predicate);
  if (bits_len64(up) > bit_depth) {
    orc_bitmap_free(b);
    return row_new();
  }
  const orc_bitmap* r = frag_row(f, BSI_SIGN);
  orc_bitmap* t = predicate < 0 ? row_intersect(b, r) : row_difference(b, r);
  orc_bitmap_free(b);
  b = t;
  for (int i = (int)(bit_depth - 1); i >= 0; i--) {
    const orc_bitmap* row = frag_row(f, (uint64_t)(BSI_OFFSET + i));
    uint64_t bit = (up >> (unsigned)i) & 1;
    t = bit == 1 ? row_intersect(b, row) : row_difference(b, row);
    orc_bitmap_free(b);
    b = t;
  }
  return b;
}

/* rangeNEQ, fragment.go:1005-1022 */
static orc_bitmap* range_neq(const frag* f, uint64_t bit_depth, int64_t predicate) {
  orc_bitmap* eq = range_eq(f, bit_depth, predicate);
  orc_bitmap* o = row_difference(frag_row(f, BSI_EXISTS), eq);
  orc_bitmap_free(eq);
  return o;
}

/* rangeLTUnsigned, fragment.go:1070-1113.  `filter` is borrowed. */
static orc_bitmap* range_lt_unsigned(const frag* f, const orc_bitmap* filter, uint64_t bit_depth, uint64_t predicate,
                                     int allow_eq) {
  const uint64_t all_ones = go_shl64(1, bit_depth) - 1; /* (1<<bitDepth)-1; 1<<64 == 0 in Go */
  if (bits_len64(predicate) > bit_depth) return row_clone(filter);
  if (predicate == all_ones && allow_eq) return row_clone(filter);
  if (predicate == all_ones && !allow_eq) {
    orc_bitmap* matches = row_new();
    for (uint64_t i = 0; i < bit_depth; i++) {
      orc_bitmap* d = row_difference(filter, frag_row(f, BSI_OFFSET + i));
      orc_bitmap* u = row_union(matches, d);
      orc_bitmap_free(d);
      orc_bitmap_free(matches);
      matches = u;
    }
    return matches;
  }
  if (allow_eq) predicate++;
  orc_bitmap* matched = row_new();
  orc_bitmap* remaining = row_clone(filter);
  for (int i = (int)(bit_depth - 1); i >= 0 && predicate > 0 && row_any(remaining); i--) {
    orc_bitmap* zeroes = row_difference(remaining, frag_row(f, (uint64_t)(BSI_OFFSET + i)));
    if ((predicate >> (unsigned)i) & 1) {
      orc_bitmap* u = row_union(matched, zeroes);
      orc_bitmap_free(matched);
      orc_bitmap_free(zeroes);
      matched = u;
      predicate &= ~(1ull << (unsigned)i);
    } else {
      orc_bitmap_free(remaining);
      remaining = zeroes;
    }
  }
  orc_bitmap_free(remaining);
  return matched;
}

/* rangeGTUnsigned, fragment.go:1157-1208 */
static orc_bitmap* range_gt_unsigned(const frag* f, const orc_bitmap* filter, uint64_t bit_depth, uint64_t predicate,
                                     int allow_eq) {
prep:
  if (predicate == 0 && allow_eq) return row_clone(filter);
  if (predicate == 0 && !allow_eq) {
    orc_bitmap* matches = row_new();
    for (uint64_t i = 0; i < bit_depth; i++) {
      orc_bitmap* d = row_intersect(filter, frag_row(f, BSI_OFFSET + i));
      orc_bitmap* u = row_union(matches, d);
      orc_bitmap_free(d);
      orc_bitmap_free(matches);
      matches = u;
    }
    return matches;
  }
  if (!allow_eq && bits_len64(predicate) > bit_depth) return row_new();
  if (allow_eq) {
    predicate--;
    allow_eq = 0;
    goto prep;
  }
  orc_bitmap* matched = row_new();
  orc_bitmap* remaining = row_clone(filter);
  predicate |= go_shl64(~0ull, bit_depth);
  for (int i = (int)(bit_depth - 1); i >= 0 && predicate < ~0ull && row_any(remaining); i--) {
    orc_bitmap* ones = row_intersect(remaining, frag_row(f, (uint64_t)(BSI_OFFSET + i)));
    if ((predicate >> (unsigned)i) & 1) {
      orc_bitmap_free(remaining);
      remaining = ones;
    } else {
      orc_bitmap* u = row_union(matched, ones);
      orc_bitmap_free(matched);
      orc_bitmap_free(ones);
      matched = u;
      predicate |= 1ull << (unsigned)i;
    }
  }
  orc_bitmap_free(remaining);
  return matched;
}

/* rangeLT, fragment.go:1024-1067 */
static orc_bitmap* range_lt(const frag* f, uint64_t bit_depth, int64_t predicate, int allow_eq) {
  if (predicate == 1 && !allow_eq) {
    predicate = 0;
    allow_eq = 1;
  }
  const orc_bitmap* b = frag_row(f, BSI_EXISTS);
  const orc_bitmap* sign = frag_row(f, BSI_SIGN);
  uint64_t up = abs_int64(predicate);
  if (predicate == 0 && !allow_eq) return row_intersect(b, sign);
  if (predicate == 0 && allow_eq) {
    orc_bitmap* zeroes = range_eq(f, bit_depth, 0);
    orc_bitmap* neg = row_intersect(b, sign);
    orc_bitmap* o = row_union(neg, zeroes);
    orc_bitmap_free(zeroes);
    orc_bitmap_free(neg);
    return o;
  }
  if (predicate < 0) {
    orc_bitmap* neg = row_intersect(b, sign);
    orc_bitmap* o = range_gt_unsigned(f, neg, bit_depth, up, allow_eq);
    orc_bitmap_free(neg);
    return o;
  }
  orc_bitmap* posf = row_difference(b, sign);
  orc_bitmap* pos = range_lt_unsigned(f, posf, bit_depth, up, allow_eq);
  orc_bitmap* neg = row_intersect(b, sign);
  orc_bitmap* o = row_union(pos, neg);
  orc_bitmap_free(posf);
  orc_bitmap_free(pos);
  orc_bitmap_free(neg);
  return o;
}

/* rangeGT, fragment.go:1115-1155 */
static orc_bitmap* range_gt(const frag* f, uint64_t bit_depth, int64_t predicate, int allow_eq) {
  if (predicate == -1 && !allow_eq) {
    predicate = 0;
    allow_eq = 1;
  }
  const orc_bitmap* b = frag_row(f, BSI_EXISTS);
  const orc_bitmap* sign = frag_row(f, BSI_SIGN);
  uint64_t up = abs_int64(predicate);
  if (predicate == 0 && !allow_eq) {
    orc_bitmap* nonzero = range_neq(f, bit_depth, 0);
    orc_bitmap* o = row_difference(nonzero, sign);
    orc_bitmap_free(nonzero);
    return o;
  }
  if (predicate == 0 && allow_eq) return row_difference(b, sign);
  if (predicate >= 0) {
    orc_bitmap* posf = row_difference(b, sign);
    orc_bitmap* o = range_gt_unsigned(f, posf, bit_depth, up, allow_eq);
    orc_bitmap_free(posf);
    return o;
  }
  orc_bitmap* negf = row_intersect(b, sign);
  orc_bitmap* neg = range_lt_unsigned(f, negf, bit_depth, up, allow_eq);
  orc_bitmap* pos = row_difference(b, sign);
  orc_bitmap* o = row_union(pos, neg);
  orc_bitmap_free(negf);
  orc_bitmap_free(neg);
  orc_bitmap_free(pos);
  return o;
}

/* rangeOp, fragment.go:937-950 */
orc_bitmap* orc_bsi_range(const orc_bitmap* const* rows, int32_t n_rows, int32_t op, uint64_t bit_depth,
                          int64_t predicate) {
  frag f = {rows, n_rows};
  switch (op) {
    case ORC_EQ: return range_eq(&f, bit_depth, predicate);
    case ORC_NEQ: return range_neq(&f, bit_depth, predicate);
    case ORC_LT: return range_lt(&f, bit_depth, predicate, 0);
    case ORC_LTE: return range_lt(&f, bit_depth, predicate, 1);
    case ORC_GT: return range_gt(&f, bit_depth, predicate, 0);
    case ORC_GTE: return range_gt(&f, bit_depth, predicate, 1);
  }
  return NULL; /* ErrInvalidRangeOperation */
}

/* rangeBetweenUnsigned, fragment.go:1262-1303 */
static orc_bitmap* range_between_unsigned(const frag* f, const orc_bitmap* filter, uint64_t bit_depth, uint64_t pmin,
                                          uint64_t pmax) {
  if (pmax > go_shl64(1, bit_depth) - 1) return range_gt_unsigned(f, filter, bit_depth, pmin, 1);
  if (pmin == 0) return range_lt_unsigned(f, filter, bit_depth, pmax, 1);
  int diff_len = (int)bits_len64(pmax ^ pmin);
  orc_bitmap* remaining = row_clone(filter);
  for (int i = (int)(bit_depth - 1); i >= diff_len; i--) {
    const orc_bitmap* row = frag_row(f, (uint64_t)(BSI_OFFSET + i));
    orc_bitmap* t = ((pmin >> (unsigned)i) & 1) ? row_intersect(remaining, row) : row_difference(remaining, row);
    orc_bitmap_free(remaining);
    remaining = t;
  }
  uint64_t equal_mask = go_shl64(~0ull, (uint64_t)diff_len);
  pmin &= ~equal_mask;
  pmax &= ~equal_mask;
  orc_bitmap* t = range_gt_unsigned(f, remaining, (uint64_t)diff_len, pmin, 1);
  orc_bitmap_free(remaining);
  remaining = t;
  t = range_lt_unsigned(f, remaining, (uint64_t)diff_len, pmax, 1);
  orc_bitmap_free(remaining);
  return t;
}

/* rangeBetween, fragment.go:1213-1260 */
orc_bitmap* orc_bsi_range_between(const orc_bitmap* const* rows, int32_t n_rows, uint64_t bit_depth, int64_t pmin,
                                  int64_t pmax) {
  frag f = {rows, n_rows};
  const orc_bitmap* b = frag_row(&f, BSI_EXISTS);
  const orc_bitmap* sign = frag_row(&f, BSI_SIGN);
  uint64_t umin = abs_int64(pmin), umax = abs_int64(pmax);
  if (pmin == pmax) return range_eq(&f, bit_depth, pmin);
  if (pmin >= 0) {
    orc_bitmap* flt = row_difference(b, sign);
    orc_bitmap* o = range_between_unsigned(&f, flt, bit_depth, umin, umax);
    orc_bitmap_free(flt);
    return o;
  }
  if (pmax < 0) {
    orc_bitmap* flt = row_intersect(b, sign);
    orc_bitmap* o = range_between_unsigned(&f, flt, bit_depth, umax, umin);
    orc_bitmap_free(flt);
    return o;
  }
  orc_bitmap* posf = row_difference(b, sign);
  orc_bitmap* pos = range_lt_unsigned(&f, posf, bit_depth, umax, 1);
  orc_bitmap* negf = row_intersect(b, sign);
  orc_bitmap* neg = range_lt_unsigned(&f, negf, bit_depth, umin, 1);
  orc_bitmap* o = row_union(pos, neg);
  orc_bitmap_free(posf);
  orc_bitmap_free(pos);
  orc_bitmap_free(negf);
  orc_bitmap_free(neg);
  return o;
}

/* ---- BSI Min / Max: fragment.min / minUnsigned / max / maxUnsigned, fragment.go:754-853.
 * `filter` may be NULL with has_filter == 0 ("no filter"); int64 arithmetic wraps as Go's does
 * (min += 1 << uint(i) with i == 63). ---------------------------------------------------- */

/* minUnsigned, fragment.go:781-801; `filter` is consumed */
static void min_unsigned(const frag* f, orc_bitmap* filter, uint64_t bit_depth, int64_t* out_min, uint64_t* out_count) {
  uint64_t mn = 0;
  uint64_t count = orc_bitmap_count(filter);
  for (int i = (int)bit_depth - 1; i >= 0; i--) {
    orc_bitmap* row = row_difference(filter, frag_row(f, (uint64_t)(BSI_OFFSET + i)));
    count = orc_bitmap_count(row);
    if (count > 0) {
      orc_bitmap_free(filter);
      filter = row;
    } else {
      mn += go_shl64(1, (uint64_t)i);
      if (i == 0) count = orc_bitmap_count(filter);
      orc_bitmap_free(row);
    }
  }
  orc_bitmap_free(filter);
  *out_min = (int64_t)mn;
  *out_count = count;
}

/* maxUnsigned, fragment.go:832-853; `filter` is consumed */
static void max_unsigned(const frag* f, orc_bitmap* filter, uint64_t bit_depth, int64_t* out_max, uint64_t* out_count) {
  uint64_t mx = 0;
  uint64_t count = orc_bitmap_count(filter);
  for (int i = (int)bit_depth - 1; i >= 0; i--) {
    orc_bitmap* row = row_intersect(frag_row(f, (uint64_t)(BSI_OFFSET + i)), filter);
    count = orc_bitmap_count(row);
    if (count > 0) {
      mx += go_shl64(1, (uint64_t)i);
      orc_bitmap_free(filter);
      filter = row;
    } else {
      if (i == 0) count = orc_bitmap_count(filter);
      orc_bitmap_free(row);
    }
  }
  orc_bitmap_free(filter);
  *out_max = (int64_t)mx;
  *out_count = count;
}

/* fragment.min, fragment.go:754-779 */
void orc_bsi_min(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                 uint64_t bit_depth, int64_t* out_min, uint64_t* out_count) {
  frag f = {rows, n_rows};
  *out_min = 0;
  *out_count = 0;
  orc_bitmap* consider = has_filter ? row_intersect(frag_row(&f, BSI_EXISTS), filter) : row_clone(frag_row(&f, BSI_EXISTS));
  if (orc_bitmap_count(consider) == 0) {
    orc_bitmap_free(consider);
    return;
  }
  orc_bitmap* neg = row_intersect(frag_row(&f, BSI_SIGN), consider);
  if (row_any(neg)) {
    int64_t v;
    max_unsigned(&f, neg, bit_depth, &v, out_count);
    *out_min = (int64_t)(0 - (uint64_t)v);
    orc_bitmap_free(consider);
    return;
  }
  orc_bitmap_free(neg);
  min_unsigned(&f, consider, bit_depth, out_min, out_count);
}

/* fragment.max, fragment.go:803-830 */
void orc_bsi_max(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                 uint64_t bit_depth, int64_t* out_max, uint64_t* out_count) {
  frag f = {rows, n_rows};
  *out_max = 0;
  *out_count = 0;
  orc_bitmap* consider = has_filter ? row_intersect(frag_row(&f, BSI_EXISTS), filter) : row_clone(frag_row(&f, BSI_EXISTS));
  if (!row_any(consider)) {
    orc_bitmap_free(consider);
    return;
  }
  orc_bitmap* pos = row_difference(consider, frag_row(&f, BSI_SIGN));
  if (!row_any(pos)) {
    int64_t v;
    orc_bitmap_free(pos);
    min_unsigned(&f, consider, bit_depth, &v, out_count);
    *out_max = (int64_t)(0 - (uint64_t)v);
    return;
  }
  orc_bitmap_free(consider);
  max_unsigned(&f, pos, bit_depth, out_max, out_count);
}

/* ---- TopK row counts: doTopK, executor.go:2705-2746, with topKFilter :2750-2774.
 * out_counts[r] = sum over the containers of row r of |container ∩ filter[slot]|
 * (or container.N() without a filter). */
void orc_topk_row_counts(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter, int32_t has_filter,
                         uint64_t* out_counts) {
  const orc_container* flt[ROW_WIDTH];
  memset(flt, 0, sizeof(flt));
  if (has_filter && filter)
    for (int32_t i = 0; i < filter->len; i++)
      if (filter->cs[i]) flt[filter->keys[i] % ROW_WIDTH] = filter->cs[i]; /* fillIt :2761-2773 */
  for (int32_t r = 0; r < n_rows; r++) {
    uint64_t count = 0;
    const orc_bitmap* row = rows[r];
    if (row)
      for (int32_t i = 0; i < row->len; i++) {
        const orc_container* c = row->cs[i];
        if (!c) continue;
        if (has_filter) {
          const orc_container* fc = flt[row->keys[i] % ROW_WIDTH];
          if (!fc) continue;
          count += (uint64_t)orc_intersection_count(c, fc);
        } else {
          count += (uint64_t)orc_n(c);
        }
      }
    out_counts[r] = count;
  }
}

/* ---- GroupBy over two fields: groupByIterator, executor.go:8617-8934.  rows[0] of the
 * first field are intersected with the filter (:8830) and each group's count is
 * rows[last].intersectionCount(rows[last-1]) (:8893): out[i*nb + j] = |(A_i ∩ F) ∩ B_j|. */
void orc_groupby_counts(const orc_bitmap* const* a_rows, int32_t na, const orc_bitmap* const* b_rows, int32_t nb,
                        const orc_bitmap* filter, int32_t has_filter, uint64_t* out) {
  for (int32_t i = 0; i < na; i++) {
    orc_bitmap* ai = has_filter ? row_intersect(a_rows[i], filter) : row_clone(a_rows[i]);
    for (int32_t j = 0; j < nb; j++) {
      orc_bitmap* eb = NULL;
      const orc_bitmap* bj = b_rows[j];
      if (!bj) bj = eb = orc_bitmap_new();
      out[(size_t)i * nb + j] = orc_bitmap_intersection_count(ai, bj);
      orc_bitmap_free(eb);
    }
    orc_bitmap_free(ai);
  }
}

/* ---- UnionRows: BitmapRowsUnion, roaring/filter.go:294-366: one streaming pass ORs every
 * selected row into 16 accumulators (c[key&15] = c[key&15].UnionInPlace(data), :327-334),
 * Results() repairs them (:341-349). */
orc_bitmap* orc_union_rows(const orc_bitmap* const* rows, int32_t n_rows) {
  orc_container* acc[ROW_WIDTH];
  memset(acc, 0, sizeof(acc));
  for (int32_t r = 0; r < n_rows; r++) {
    const orc_bitmap* row = rows[r];
    if (!row) continue;
    for (int32_t i = 0; i < row->len; i++) {
      const orc_container* c = row->cs[i];
      if (!c || orc_n(c) == 0) continue;
      uint64_t pos = row->keys[i] & (ROW_WIDTH - 1);
      orc_container* t = acc[pos] ? orc_union_in_place(acc[pos], c) : orc_clone(c);
      orc_free(acc[pos]);
      acc[pos] = t;
    }
  }
  orc_bitmap* o = orc_bitmap_new();
  for (int i = 0; i < ROW_WIDTH; i++)
    if (acc[i]) {
      if (orc_n(acc[i]) > 0) orc_bitmap_put(o, (uint64_t)i, acc[i]);
      else orc_free(acc[i]);
    }
  return o;
}

/* ---- test hooks: the reference's regression tests call the unsigned kernels directly
 * (fragment_internal_test.go:758, 841, 859, 816) -------------------------------------------- */
orc_bitmap* orc_bsi_range_lt_unsigned(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter,
                                      uint64_t bit_depth, uint64_t predicate, int32_t allow_eq) {
  frag f = {rows, n_rows};
  return range_lt_unsigned(&f, filter, bit_depth, predicate, allow_eq);
}
orc_bitmap* orc_bsi_range_gt_unsigned(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter,
                                      uint64_t bit_depth, uint64_t predicate, int32_t allow_eq) {
  frag f = {rows, n_rows};
  return range_gt_unsigned(&f, filter, bit_depth, predicate, allow_eq);
}
orc_bitmap* orc_bsi_range_between_unsigned(const orc_bitmap* const* rows, int32_t n_rows, const orc_bitmap* filter,
                                           uint64_t bit_depth, uint64_t pmin, uint64_t pmax) {
  frag f = {rows, n_rows};
  return range_between_unsigned(&f, filter, bit_depth, pmin, pmax);
}
