/*
 * roaring_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See roaring_oracle.h.  Plain C99 restatement of FeatureBase's Go roaring container
 * algebra; every function cites the reference lines it follows (paths relative to the
 * FeatureBase tree).  Go integer semantics that differ from C are made explicit:
 * uint16 arithmetic wraps (casts below), shifts >= 64 yield 0 (go_shl/go_shr).
 */
#include "roaring_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MAXV ORC_MAX_CONTAINER_VAL
#define BN ORC_BITMAP_N

/* ---- Go semantics helpers -------------------------------------------------------- */
static inline uint64_t go_shl(uint64_t x, unsigned s) { return s >= 64 ? 0 : x << s; }
static inline uint64_t go_shr(uint64_t x, unsigned s) { return s >= 64 ? 0 : x >> s; }
static inline int popcnt(uint64_t x) { return __builtin_popcountll(x); } /* bits.OnesCount64, roaring.go:6924 */
static inline int ctz64(uint64_t x) { return x ? __builtin_ctzll(x) : 64; } /* trailingZeroN */

static void* xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) abort();
  return p;
}
static void* xcalloc(size_t n, size_t sz) {
  void* p = calloc(n ? n : 1, sz);
  if (!p) abort();
  return p;
}

/* ---- growable vectors ------------------------------------------------------------- */
typedef struct {
  uint16_t* v;
  int32_t n, cap;
} u16vec;
static void u16_push(u16vec* a, uint16_t x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 64;
    a->v = (uint16_t*)realloc(a->v, (size_t)a->cap * 2);
    if (!a->v) abort();
  }
  a->v[a->n++] = x;
}
typedef struct {
  orc_interval16* v;
  int32_t n, cap;
} ivvec;
static void iv_push(ivvec* a, orc_interval16 x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 64;
    a->v = (orc_interval16*)realloc(a->v, (size_t)a->cap * sizeof(orc_interval16));
    if (!a->v) abort();
  }
  a->v[a->n++] = x;
}

/* ---- construction / access -------------------------------------------------------- */
static orc_container* mk(uint8_t typ, int32_t n, int32_t len, void* data) {
  orc_container* c = (orc_container*)xmalloc(sizeof(*c));
  c->typ = typ;
  c->n = n;
  c->len = len;
  c->data = data;
  return c;
}
static inline const uint16_t* ARR(const orc_container* c) { return (const uint16_t*)c->data; }
static inline uint64_t* BMP(const orc_container* c) { return (uint64_t*)c->data; }
static inline const orc_interval16* RUNS(const orc_container* c) { return (const orc_interval16*)c->data; }

/* Interval16.runlen, roaring.go:3047-3049: the subtraction is done in uint16 */
static inline int32_t runlen(orc_interval16 iv) { return 1 + (int32_t)(uint16_t)(iv.last - iv.start); }

orc_container* orc_new_array(const uint16_t* v, int32_t n) {
  uint16_t* d = (uint16_t*)xmalloc((size_t)n * 2);
  if (n) memcpy(d, v, (size_t)n * 2);
  return mk(ORC_ARRAY, n, n, d);
}
/* takes ownership of the vector's storage */
static orc_container* array_from_vec(u16vec* a) {
  if (!a->v) a->v = (uint16_t*)xmalloc(1);
  return mk(ORC_ARRAY, a->n, a->n, a->v);
}
orc_container* orc_new_bitmap(const uint64_t* words, int32_t n) {
  uint64_t* d = (uint64_t*)xcalloc(BN, 8);
  if (words) memcpy(d, words, BN * 8);
  if (n < 0) {
    n = 0;
    for (int i = 0; i < BN; i++) n += popcnt(d[i]);
  }
  return mk(ORC_BITMAP, n, BN, d);
}
/* NewContainerRun computes N from the intervals (container_stash.go NewContainerRun) */
orc_container* orc_new_run(const orc_interval16* r, int32_t len) {
  orc_interval16* d = (orc_interval16*)xmalloc((size_t)len * sizeof(*d));
  int32_t n = 0;
  for (int32_t i = 0; i < len; i++) {
    d[i] = r[i];
    n += runlen(r[i]);
  }
  return mk(ORC_RUN, n, len, d);
}
static orc_container* run_from_vec(ivvec* a, int32_t n) {
  if (!a->v) a->v = (orc_interval16*)xmalloc(1);
  if (n < 0) {
    n = 0;
    for (int32_t i = 0; i < a->n; i++) n += runlen(a->v[i]);
  }
  return mk(ORC_RUN, n, a->n, a->v);
}
orc_container* orc_clone(const orc_container* c) {
  if (!c) return NULL;
  size_t bytes = c->typ == ORC_ARRAY ? (size_t)c->len * 2 : c->typ == ORC_RUN ? (size_t)c->len * 4 : (size_t)BN * 8;
  void* d = xmalloc(bytes);
  if (bytes) memcpy(d, c->data, bytes);
  return mk(c->typ, c->n, c->len, d);
}
void orc_free(orc_container* c) {
  if (!c) return;
  free(c->data);
  free(c);
}
int32_t orc_n(const orc_container* c) { return c ? c->n : 0; }
int32_t orc_typ(const orc_container* c) { return c ? c->typ : ORC_NIL; }
int32_t orc_len(const orc_container* c) { return c ? c->len : 0; }
const void* orc_data(const orc_container* c) { return c ? c->data : NULL; }

/* fullContainer = run {0,65535}, roaring.go:67 */
static orc_container* full_container(void) {
  orc_interval16 iv = {0, MAXV};
  return orc_new_run(&iv, 1);
}

void orc_to_words(const orc_container* c, uint64_t* out) {
  memset(out, 0, BN * 8);
  if (!c) return;
  if (c->typ == ORC_BITMAP) {
    memcpy(out, c->data, BN * 8);
  } else if (c->typ == ORC_ARRAY) {
    for (int32_t i = 0; i < c->len; i++) out[ARR(c)[i] >> 6] |= 1ull << (ARR(c)[i] & 63);
  } else {
    for (int32_t i = 0; i < c->len; i++)
      for (uint32_t v = RUNS(c)[i].start; v <= RUNS(c)[i].last; v++) out[v >> 6] |= 1ull << (v & 63);
  }
}

/* ---- counting ----------------------------------------------------------------------- */

/* ArrayCountRange, roaring.go:3074-3089 */
int32_t orc_array_count_range(const uint16_t* a, int32_t len, int32_t start, int32_t end) {
  int32_t lo = 0, hi = len; /* sort.Search: first i with a[i] >= start */
  while (lo < hi) {
    int32_t mid = lo + (hi - lo) / 2;
    if ((int32_t)a[mid] >= start) hi = mid;
    else lo = mid + 1;
  }
  int32_t n = 0;
  for (int32_t i = lo; i < len; i++) {
    if ((int32_t)a[i] >= end) break;
    n++;
  }
  return n;
}

/* BitmapCountRange, roaring.go:3092-3125 */
int32_t orc_words_count_range(const uint64_t* bm, int32_t start, int32_t end) {
  uint64_t n = 0;
  int32_t i = start / 64, j = end / 64;
  if (i == j) {
    unsigned offi = (unsigned)(start % 64), offj = (unsigned)(64 - end % 64);
    n += popcnt(go_shl(go_shr(bm[i], offi), offj + offi));
    return (int32_t)n;
  }
  unsigned off = (unsigned)start % 64;
  if (off != 0) {
    n += popcnt(bm[i] >> off);
    i++;
  }
  for (; i < j; i++) n += popcnt(bm[i]);
  if (j < BN) {
    unsigned o2 = 64 - ((unsigned)end % 64);
    n += popcnt(go_shl(bm[j], o2));
  }
  return (int32_t)n;
}

/* RunCountRange, roaring.go:3200-3232 (conditions restated as written, including the
 * inclusive/exclusive mix at the range end) */
int32_t orc_run_count_range(const orc_interval16* r, int32_t len, int32_t start, int32_t end) {
  int32_t n = 0;
  for (int32_t k = 0; k < len; k++) {
    int32_t s = r[k].start, l = r[k].last;
    if (l < start) continue;
    if (end < s) break;
    if (s <= start && l >= end) return end - start;
    if (s >= start && l <= end) n += runlen(r[k]);
    if (s < start && l < end) n += l - start + 1;
    if (s > start && l >= end) n += end - s;
  }
  return n;
}

/* Container.countRange, roaring.go:3058-3072 */
int32_t orc_count_range(const orc_container* c, int32_t start, int32_t end) {
  if (!c) return 0;
  if (c->typ == ORC_ARRAY) return orc_array_count_range(ARR(c), c->len, start, end);
  if (c->typ == ORC_RUN) return orc_run_count_range(RUNS(c), c->len, start, end);
  return orc_words_count_range(BMP(c), start, end);
}
/* Container.count, roaring.go:3052 */
int32_t orc_count(const orc_container* c) { return orc_count_range(c, 0, MAXV + 1); }

/* bitmapCountRuns, roaring.go:3372-3380 */
static int32_t bitmap_count_runs(const uint64_t* bm) {
  int32_t r = 0;
  for (int i = 0; i < BN - 1; i++) {
    uint64_t v = bm[i], v1 = bm[i + 1];
    r += (int32_t)(popcnt((v << 1) & ~v) + ((v >> 63) & ~v1));
  }
  uint64_t vl = bm[BN - 1];
  r += (int32_t)(popcnt((vl << 1) & ~vl) + (vl >> 63));
  return r;
}
/* arrayCountRuns, roaring.go:3382-3392 */
static int32_t array_count_runs(const uint16_t* a, int32_t len) {
  int32_t r = 0, prev = -2;
  for (int32_t i = 0; i < len; i++) {
    if (prev + 1 != (int32_t)a[i]) r++;
    prev = a[i];
  }
  return r;
}
/* Container.countRuns, roaring.go:3398-3409 */
int32_t orc_count_runs(const orc_container* c) {
  if (!c) return 0;
  if (c->typ == ORC_ARRAY) return array_count_runs(ARR(c), c->len);
  if (c->typ == ORC_BITMAP) return bitmap_count_runs(BMP(c));
  if (c->typ == ORC_RUN) return c->len;
  return 0;
}

/* ---- conversions --------------------------------------------------------------------- */

/* bitmapToArray, roaring.go:3687-3753 */
orc_container* orc_bitmap_to_array(const orc_container* c) {
  if (!c) return NULL;
  u16vec out = {0};
  const uint64_t* bm = BMP(c);
  for (int i = 0; i < BN; i++) {
    uint64_t w = bm[i];
    while (w) {
      uint64_t t = w & (~w + 1);
      u16_push(&out, (uint16_t)(i * 64 + popcnt(t - 1)));
      w ^= t;
    }
  }
  return array_from_vec(&out);
}
/* arrayToBitmap, roaring.go:3756-3789 */
orc_container* orc_array_to_bitmap(const orc_container* c) {
  if (!c) return NULL;
  orc_container* o = orc_new_bitmap(NULL, 0);
  for (int32_t i = 0; i < c->len; i++) BMP(o)[ARR(c)[i] / 64] |= 1ull << (ARR(c)[i] % 64);
  o->n = c->n;
  return o;
}
/* runToBitmap, roaring.go:3792-3856 */
orc_container* orc_run_to_bitmap(const orc_container* c) {
  if (!c) return NULL;
  orc_container* o = orc_new_bitmap(NULL, 0);
  uint64_t* bm = BMP(o);
  for (int32_t k = 0; k < c->len; k++) {
    orc_interval16 iv = RUNS(c)[k];
    unsigned w1 = iv.start / 64, w2 = iv.last / 64, b1 = iv.start & 63, b2 = iv.last & 63;
    uint64_t m1 = (1ull << b1) - 1;
    uint64_t m2 = (((1ull << b2) - 1) << 1) | 1;
    if (w1 == w2) {
      bm[w1] |= (m2 & ~m1);
      continue;
    }
    bm[w2] |= m2;
    bm[w1] |= ~m1;
    for (unsigned w = w1 + 1; w < w2; w++) bm[w] = ~0ull;
  }
  o->n = c->n;
  return o;
}
/* bitmapToRun, roaring.go:3859-3928 */
orc_container* orc_bitmap_to_run(const orc_container* c) {
  if (!c) return NULL;
  ivvec runs = {0};
  if (c->n == 0) return run_from_vec(&runs, 0);
  const uint64_t* bm = BMP(c);
  uint64_t cur = bm[0];
  uint16_t i = 0, start, last;
  for (;;) {
    while (cur == 0 && i < BN - 1) {
      i++;
      cur = bm[i];
    }
    if (cur == 0) break;
    uint16_t cs = (uint16_t)ctz64(cur);
    start = (uint16_t)(64 * i + cs);
    cur = cur | (cur - 1);
    while (cur == ~0ull && i < BN - 1) {
      i++;
      cur = bm[i];
    }
    if (cur == ~0ull) {
      orc_interval16 iv = {start, MAXV};
      iv_push(&runs, iv);
      break;
    }
    uint16_t cl = (uint16_t)ctz64(~cur);
    last = (uint16_t)(64 * i + cl);
    orc_interval16 iv = {start, (uint16_t)(last - 1)};
    iv_push(&runs, iv);
    cur = cur & (cur + 1);
  }
  return run_from_vec(&runs, c->n);
}
/* arrayToRun, roaring.go:3931-3974 */
orc_container* orc_array_to_run(const orc_container* c) {
  if (!c) return NULL;
  ivvec runs = {0};
  if (c->n == 0) return run_from_vec(&runs, 0);
  const uint16_t* a = ARR(c);
  uint16_t start = a[0];
  for (int32_t i = 1; i < c->len; i++) {
    if ((uint16_t)(a[i] - a[i - 1]) > 1) {
      orc_interval16 iv = {start, a[i - 1]};
      iv_push(&runs, iv);
      start = a[i];
    }
  }
  orc_interval16 iv = {start, a[c->len - 1]};
  iv_push(&runs, iv);
  return run_from_vec(&runs, c->n);
}
/* runToArray, roaring.go:3977-4018 */
orc_container* orc_run_to_array(const orc_container* c) {
  if (!c) return NULL;
  u16vec out = {0};
  for (int32_t k = 0; k < c->len; k++)
    for (int v = RUNS(c)[k].start; v <= (int)RUNS(c)[k].last; v++) u16_push(&out, (uint16_t)v);
  return array_from_vec(&out);
}

/* Container.optimize, roaring.go:3412-3461 */
orc_container* orc_optimize(const orc_container* c) {
  if (orc_n(c) == 0) return NULL;
  int32_t runs = orc_count_runs(c);
  uint8_t nt;
  if (runs <= ORC_RUN_MAX_SIZE && runs <= c->n / 2) nt = ORC_RUN;
  else if (c->n < ORC_ARRAY_MAX_SIZE) nt = ORC_ARRAY;
  else nt = ORC_BITMAP;
  if (c->typ == ORC_ARRAY) {
    if (nt == ORC_BITMAP) return orc_array_to_bitmap(c);
    if (nt == ORC_RUN) return orc_array_to_run(c);
  } else if (c->typ == ORC_BITMAP) {
    if (nt == ORC_ARRAY) return orc_bitmap_to_array(c);
    if (nt == ORC_RUN) return orc_bitmap_to_run(c);
  } else if (c->typ == ORC_RUN) {
    if (nt == ORC_BITMAP) return orc_run_to_bitmap(c);
    if (nt == ORC_ARRAY) return orc_run_to_array(c);
  }
  return orc_clone(c);
}

/* ---- bitmap range helpers ------------------------------------------------------------- */

/* Container.bitmapSetRange, roaring.go:5237-5261 */
void orc_bitmap_set_range(orc_container* c, uint64_t i, uint64_t j) {
  uint64_t* bm = BMP(c);
  uint64_t x = i >> 6, y = (j - 1) >> 6;
  uint64_t X = ~0ull << (i % 64), Y = ~0ull >> (63 - ((j - 1) % 64));
  int32_t n = c->n;
  if (x == y) {
    n += (int32_t)((j - i) - (uint64_t)popcnt(bm[x] & (X & Y)));
    bm[x] |= (X & Y);
  } else {
    n += popcnt(X) - popcnt(bm[x] & X);
    bm[x] |= X;
    for (uint64_t k = x + 1; k < y; k++) {
      n += 64 - popcnt(bm[k]);
      bm[k] = ~0ull;
    }
    n += popcnt(Y) - popcnt(bm[y] & Y);
    bm[y] |= Y;
  }
  c->n = n;
}
/* Container.bitmapXorRange, roaring.go:5281-5306 */
void orc_bitmap_xor_range(orc_container* c, uint64_t i, uint64_t j) {
  uint64_t* bm = BMP(c);
  uint64_t x = i >> 6, y = (j - 1) >> 6;
  uint64_t X = ~0ull << (i % 64), Y = ~0ull >> (63 - ((j - 1) % 64));
  int32_t n = c->n;
  if (x == y) {
    int cnt = popcnt(bm[x]);
    bm[x] ^= (X & Y);
    n += popcnt(bm[x]) - cnt;
  } else {
    int cnt = popcnt(bm[x]);
    bm[x] ^= X;
    n += popcnt(bm[x]) - cnt;
    for (uint64_t k = x + 1; k < y; k++) {
      cnt = popcnt(bm[k]);
      bm[k] ^= ~0ull;
      n += popcnt(bm[k]) - cnt;
    }
    cnt = popcnt(bm[y]);
    bm[y] ^= Y;
    n += popcnt(bm[y]) - cnt;
  }
  c->n = n;
}
/* Container.bitmapZeroRange, roaring.go:5309-5330 */
void orc_bitmap_zero_range(orc_container* c, uint64_t i, uint64_t j) {
  uint64_t* bm = BMP(c);
  uint64_t x = i >> 6, y = (j - 1) >> 6;
  uint64_t X = ~0ull << (i % 64), Y = ~0ull >> (63 - ((j - 1) % 64));
  int32_t n = c->n;
  if (x == y) {
    n -= popcnt(bm[x] & (X & Y));
    bm[x] &= ~(X & Y);
  } else {
    n -= popcnt(bm[x] & X);
    bm[x] &= ~X;
    for (uint64_t k = x + 1; k < y; k++) {
      n -= popcnt(bm[k]);
      bm[k] = 0;
    }
    n -= popcnt(bm[y] & Y);
    bm[y] &= ~Y;
  }
  c->n = n;
}

static inline int bitmap_contains(const uint64_t* bm, uint16_t v) { return (bm[v / 64] >> (v % 64)) & 1; }

/* ---- intersectionCount ------------------------------------------------------------------ */

/* intersectionCountArrayArray, roaring.go:4514-4535 */
static int32_t icount_array_array(const orc_container* a, const orc_container* b) {
  const uint16_t *ca = ARR(a), *cb = ARR(b);
  int32_t na = a->len, nb = b->len, n = 0;
  if (na > nb) {
    const uint16_t* t = ca;
    ca = cb;
    cb = t;
    int32_t tn = na;
    na = nb;
    nb = tn;
  }
  int32_t j = 0;
  for (int32_t i = 0; i < na; i++) {
    uint16_t va = ca[i];
    while (cb[j] < va) {
      j++;
      if (j >= nb) return n;
    }
    if (cb[j] == va) n++;
  }
  return n;
}
/* intersectionCountArrayRun, roaring.go:4537-4553 */
static int32_t icount_array_run(const orc_container* a, const orc_container* b) {
  int32_t na = a->len, nb = b->len, n = 0;
  for (int32_t i = 0, j = 0; i < na && j < nb;) {
    uint16_t va = ARR(a)[i];
    orc_interval16 vb = RUNS(b)[j];
    if (va < vb.start) i++;
    else if (va >= vb.start && va <= vb.last) {
      i++;
      n++;
    } else if (va > vb.last) j++;
  }
  return n;
}
/* intersectionCountRunRun, roaring.go:4555-4586 */
static int32_t icount_run_run(const orc_container* a, const orc_container* b) {
  int32_t na = a->len, nb = b->len, n = 0;
  for (int32_t i = 0, j = 0; i < na && j < nb;) {
    orc_interval16 va = RUNS(a)[i], vb = RUNS(b)[j];
    if (va.last < vb.start) i++;
    else if (va.start > vb.last) j++;
    else if (va.last > vb.last && va.start >= vb.start) {
      n += 1 + (int32_t)(uint16_t)(vb.last - va.start);
      j++;
    } else if (va.last > vb.last && va.start < vb.start) {
      n += 1 + (int32_t)(uint16_t)(vb.last - vb.start);
      j++;
    } else if (va.last <= vb.last && va.start >= vb.start) {
      n += 1 + (int32_t)(uint16_t)(va.last - va.start);
      i++;
    } else if (va.last <= vb.last && va.start < vb.start) {
      n += 1 + (int32_t)(uint16_t)(va.last - vb.start);
      i++;
    }
  }
  return n;
}
/* intersectionCountBitmapRun, roaring.go:4588-4594 */
static int32_t icount_bitmap_run(const orc_container* a, const orc_container* b) {
  int32_t n = 0;
  for (int32_t k = 0; k < b->len; k++)
    n += orc_words_count_range(BMP(a), (int32_t)RUNS(b)[k].start, (int32_t)RUNS(b)[k].last + 1);
  return n;
}
/* intersectionCountArrayBitmap, roaring.go:4596-4609 */
static int32_t icount_array_bitmap(const orc_container* a, const orc_container* b) {
  int32_t n = 0;
  for (int32_t k = 0; k < a->len; k++) {
    uint16_t val = ARR(a)[k];
    int i = val >> 6;
    if (i >= BN) break;
    n += (int32_t)(BMP(b)[i] >> (val % 64)) & 1;
  }
  return n;
}
/* popcountAndSlice, roaring.go:6928-6939; intersectionCountBitmapBitmap :4611 */
static int32_t icount_bitmap_bitmap(const orc_container* a, const orc_container* b) {
  uint64_t n = 0;
  for (int i = 0; i < BN; i++) n += popcnt(BMP(a)[i] & BMP(b)[i]);
  return (int32_t)n;
}

/* intersectionCount, roaring.go:4477-4512 */
int32_t orc_intersection_count(const orc_container* a, const orc_container* b) {
  if (orc_n(a) == MAXV + 1) return orc_n(b);
  if (orc_n(b) == MAXV + 1) return orc_n(a);
  if (orc_n(a) == 0 || orc_n(b) == 0) return 0;
  if (a->typ == ORC_ARRAY) {
    if (b->typ == ORC_ARRAY) return icount_array_array(a, b);
    if (b->typ == ORC_RUN) return icount_array_run(a, b);
    return icount_array_bitmap(a, b);
  } else if (a->typ == ORC_RUN) {
    if (b->typ == ORC_ARRAY) return icount_array_run(b, a);
    if (b->typ == ORC_RUN) return icount_run_run(a, b);
    return icount_bitmap_run(b, a);
  } else {
    if (b->typ == ORC_ARRAY) return icount_array_bitmap(b, a);
    if (b->typ == ORC_RUN) return icount_bitmap_run(a, b);
    return icount_bitmap_bitmap(a, b);
  }
}

/* ---- intersect ------------------------------------------------------------------------------ */

/* Container.runAppendInterval, roaring.go:5155-5178; returns the cardinality increase */
static int32_t run_append_interval(ivvec* runs, orc_interval16 v) {
  if (runs->n == 0) {
    iv_push(runs, v);
    return (int32_t)(uint16_t)(v.last - v.start) + 1;
  }
  orc_interval16 last = runs->v[runs->n - 1];
  if (last.last == MAXV) return 0;
  if ((uint16_t)(last.last + 1) >= v.start && v.last > last.last) {
    runs->v[runs->n - 1].last = v.last;
    return (int32_t)(uint16_t)(v.last - last.last);
  } else if ((uint16_t)(last.last + 1) < v.start) {
    iv_push(runs, v);
    return (int32_t)(uint16_t)(v.last - v.start) + 1;
  }
  return 0;
}

/* intersectArrayArray, roaring.go:4793-4810 */
static orc_container* intersect_array_array(const orc_container* a, const orc_container* b) {
  u16vec out = {0};
  for (int32_t i = 0, j = 0; i < a->len && j < b->len;) {
    uint16_t va = ARR(a)[i], vb = ARR(b)[j];
    if (va < vb) i++;
    else if (va > vb) j++;
    else {
      u16_push(&out, va);
      i++;
      j++;
    }
  }
  return array_from_vec(&out);
}
/* intersectArrayRun, roaring.go:4815-4832 */
static orc_container* intersect_array_run(const orc_container* a, const orc_container* b) {
  u16vec out = {0};
  for (int32_t i = 0, j = 0; i < a->len && j < b->len;) {
    uint16_t va = ARR(a)[i];
    orc_interval16 vb = RUNS(b)[j];
    if (va < vb.start) i++;
    else if (va > vb.last) j++;
    else {
      u16_push(&out, va);
      i++;
    }
  }
  return array_from_vec(&out);
}
/* intersectRunRun, roaring.go:4835-4875 */
static orc_container* intersect_run_run(const orc_container* a, const orc_container* b) {
  ivvec out = {0};
  int32_t n = 0;
  for (int32_t i = 0, j = 0; i < a->len && j < b->len;) {
    orc_interval16 va = RUNS(a)[i], vb = RUNS(b)[j];
    if (va.last < vb.start) i++;
    else if (vb.last < va.start) j++;
    else if (va.last > vb.last && va.start >= vb.start) {
      orc_interval16 iv = {va.start, vb.last};
      n += run_append_interval(&out, iv);
      j++;
    } else if (va.last > vb.last && va.start < vb.start) {
      n += run_append_interval(&out, vb);
      j++;
    } else if (va.last <= vb.last && va.start >= vb.start) {
      n += run_append_interval(&out, va);
      i++;
    } else if (va.last <= vb.last && va.start < vb.start) {
      orc_interval16 iv = {vb.start, va.last};
      n += run_append_interval(&out, iv);
      i++;
    }
  }
  int32_t nruns = out.n;
  orc_container* o = run_from_vec(&out, n);
  if (n < ORC_ARRAY_MAX_SIZE && nruns > n / 2) {
    orc_container* t = orc_run_to_array(o);
    orc_free(o);
    return t;
  } else if (nruns > ORC_RUN_MAX_SIZE) {
    orc_container* t = orc_run_to_bitmap(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* intersectBitmapRun, roaring.go:4879-4942 */
static orc_container* intersect_bitmap_run(const orc_container* a, const orc_container* b) {
  const orc_interval16* runs = RUNS(b);
  if (b->n <= ORC_ARRAY_MAX_SIZE) {
    u16vec out = {0};
    for (int32_t k = 0; k < b->len; k++)
      for (int i = runs[k].start; i <= (int)runs[k].last; i++)
        if (bitmap_contains(BMP(a), (uint16_t)i)) u16_push(&out, (uint16_t)i);
    return array_from_vec(&out);
  }
  orc_container* o = orc_new_bitmap(NULL, 0);
  uint64_t* bm = BMP(o);
  const uint64_t* ab = BMP(a);
  int32_t n = 0;
  for (int32_t j = 0; j < b->len; j++) {
    orc_interval16 vb = runs[j];
    uint16_t i = vb.start >> 6; /* all of these are uint16 in Go and wrap */
    uint16_t vastart = (uint16_t)(i << 6);
    uint16_t valast = (uint16_t)(vastart + 63);
    while (valast >= vb.start && vastart <= vb.last && i < BN) {
      if (vastart >= vb.start && valast <= vb.last) { /* a within b */
        bm[i] = ab[i];
        n += popcnt(ab[i]);
      } else if (vb.start >= vastart && vb.last <= valast) { /* b within a */
        uint64_t mask = go_shl(go_shl(1, (unsigned)(uint16_t)(vb.last - vb.start + 1)) - 1, (unsigned)(uint16_t)(vb.start - vastart));
        uint64_t bits = ab[i] & mask;
        bm[i] |= bits;
        n += popcnt(bits);
      } else if (vastart < vb.start) { /* a overlaps front of b */
        unsigned off = (unsigned)(uint16_t)(64 - (1 + valast - vb.start));
        uint64_t bits = go_shl(go_shr(ab[i], off), off);
        bm[i] |= bits;
        n += popcnt(bits);
      } else if (vb.start < vastart) { /* b overlaps front of a */
        unsigned off = (unsigned)(uint16_t)(64 - (1 + vb.last - vastart));
        uint64_t bits = go_shr(go_shl(ab[i], off), off);
        bm[i] |= bits;
        n += popcnt(bits);
      }
      i++;
      vastart = (uint16_t)(i << 6);
      valast = (uint16_t)(vastart + 63);
    }
  }
  o->n = n;
  return o;
}
/* intersectArrayBitmap, roaring.go:4944-4958 */
static orc_container* intersect_array_bitmap(const orc_container* a, const orc_container* b) {
  u16vec out = {0};
  for (int32_t k = 0; k < a->len; k++) {
    uint16_t va = ARR(a)[k];
    if (BMP(b)[va / 64] & (1ull << (va % 64))) u16_push(&out, va);
  }
  return array_from_vec(&out);
}
/* intersectBitmapBitmap, roaring.go:4960-4978 (never down-converts) */
static orc_container* intersect_bitmap_bitmap(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_new_bitmap(NULL, 0);
  int32_t n = 0;
  for (int i = 0; i < BN; i++) {
    BMP(o)[i] = BMP(a)[i] & BMP(b)[i];
    n += popcnt(BMP(o)[i]);
  }
  o->n = n;
  return o;
}

/* intersect, roaring.go:4753-4791 */
orc_container* orc_intersect(const orc_container* a, const orc_container* b) {
  if (orc_n(a) == MAXV + 1) return orc_clone(b); /* b.Freeze() */
  if (orc_n(b) == MAXV + 1) return orc_clone(a);
  if (orc_n(a) == 0 || orc_n(b) == 0) return NULL;
  if (a->typ == ORC_ARRAY) {
    if (b->typ == ORC_ARRAY) return intersect_array_array(a, b);
    if (b->typ == ORC_RUN) return intersect_array_run(a, b);
    return intersect_array_bitmap(a, b);
  } else if (a->typ == ORC_RUN) {
    if (b->typ == ORC_ARRAY) return intersect_array_run(b, a);
    if (b->typ == ORC_RUN) return intersect_run_run(a, b);
    return intersect_bitmap_run(b, a);
  } else {
    if (b->typ == ORC_ARRAY) return intersect_array_bitmap(b, a);
    if (b->typ == ORC_RUN) return intersect_bitmap_run(a, b);
    return intersect_bitmap_bitmap(a, b);
  }
}

/* ---- union ------------------------------------------------------------------------------------ */

/* unionArrayArray, roaring.go:5016-5056 (output may exceed 4096 elements, :5054) */
static orc_container* union_array_array(const orc_container* a, const orc_container* b) {
  if (a->n == 0) return orc_clone(b);
  if (b->n == 0) return orc_clone(a);
  const uint16_t *s1 = ARR(a), *s2 = ARR(b);
  int32_t n1 = a->len, n2 = b->len, i = 0, j = 0;
  u16vec out = {0};
  for (;;) {
    uint16_t va = s1[i], vb = s2[j];
    if (va < vb) {
      u16_push(&out, va);
      i++;
    } else if (va > vb) {
      u16_push(&out, vb);
      j++;
    } else {
      u16_push(&out, va);
      i++;
      j++;
    }
    if (j >= n2) {
      for (; i < n1; i++) u16_push(&out, s1[i]);
      break;
    }
    if (i >= n1) {
      for (; j < n2; j++) u16_push(&out, s2[j]);
      break;
    }
  }
  return array_from_vec(&out);
}
/* unionArrayRun, roaring.go:5120-5153 */
static orc_container* union_array_run(const orc_container* a, const orc_container* b) {
  ivvec out = {0};
  int32_t na = a->len, nb = b->len, n = 0;
  orc_interval16 vb = {0, 0};
  uint16_t va = 0;
  for (int32_t i = 0, j = 0; i < na || j < nb;) {
    if (i < na) va = ARR(a)[i];
    if (j < nb) vb = RUNS(b)[j];
    if (i < na && (j >= nb || va < vb.start)) {
      orc_interval16 iv = {va, va};
      n += run_append_interval(&out, iv);
      i++;
    } else {
      n += run_append_interval(&out, vb);
      j++;
    }
  }
  int32_t nruns = out.n;
  orc_container* o = run_from_vec(&out, n);
  if (n < ORC_ARRAY_MAX_SIZE) {
    orc_container* t = orc_run_to_array(o);
    orc_free(o);
    return t;
  } else if (nruns > ORC_RUN_MAX_SIZE) {
    orc_container* t = orc_run_to_bitmap(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* unionRunRun, roaring.go:5182-5209 */
static orc_container* union_run_run(const orc_container* a, const orc_container* b) {
  ivvec out = {0};
  int32_t na = a->len, nb = b->len, n = 0;
  orc_interval16 va = {0, 0}, vb = {0, 0};
  for (int32_t i = 0, j = 0; i < na || j < nb;) {
    if (i < na) va = RUNS(a)[i];
    if (j < nb) vb = RUNS(b)[j];
    if (i < na && (j >= nb || va.start < vb.start)) {
      n += run_append_interval(&out, va);
      i++;
    } else {
      n += run_append_interval(&out, vb);
      j++;
    }
  }
  int32_t nruns = out.n;
  orc_container* o = run_from_vec(&out, n);
  if (nruns > ORC_RUN_MAX_SIZE) {
    orc_container* t = orc_run_to_bitmap(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* unionBitmapRun, roaring.go:5211-5218 */
static orc_container* union_bitmap_run(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_clone(a);
  for (int32_t k = 0; k < b->len; k++) orc_bitmap_set_range(o, RUNS(b)[k].start, (uint64_t)RUNS(b)[k].last + 1);
  return o;
}
/* unionArrayBitmap, roaring.go:5424-5436 */
static orc_container* union_array_bitmap(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_clone(b);
  int32_t n = o->n;
  for (int32_t k = 0; k < a->len; k++) {
    uint16_t v = ARR(a)[k];
    if (!bitmap_contains(BMP(o), v)) {
      BMP(o)[v / 64] |= 1ull << (v % 64);
      n++;
    }
  }
  o->n = n;
  return o;
}
/* unionBitmapBitmap, roaring.go:5450-5470 */
static orc_container* union_bitmap_bitmap(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_new_bitmap(NULL, 0);
  int32_t n = 0;
  for (int i = 0; i < BN; i++) {
    BMP(o)[i] = BMP(a)[i] | BMP(b)[i];
    n += popcnt(BMP(o)[i]);
  }
  o->n = n;
  return o;
}

/* union, roaring.go:4980-5012.  A nil operand never reaches the Go dispatcher (isArray
 * panics on nil, container_stash.go:811); here it is treated as an empty array. */
orc_container* orc_union(const orc_container* a, const orc_container* b) {
  if (orc_n(a) == MAXV + 1 || orc_n(b) == MAXV + 1) return full_container();
  orc_container *ea = NULL, *eb = NULL;
  if (!a) a = ea = orc_new_array(NULL, 0);
  if (!b) b = eb = orc_new_array(NULL, 0);
  orc_container* r;
  if (a->typ == ORC_ARRAY) {
    if (b->typ == ORC_ARRAY) r = union_array_array(a, b);
    else if (b->typ == ORC_RUN) r = union_array_run(a, b);
    else r = union_array_bitmap(a, b);
  } else if (a->typ == ORC_RUN) {
    if (b->typ == ORC_ARRAY) r = union_array_run(b, a);
    else if (b->typ == ORC_RUN) r = union_run_run(a, b);
    else r = union_bitmap_run(b, a);
  } else {
    if (b->typ == ORC_ARRAY) r = union_array_bitmap(b, a);
    else if (b->typ == ORC_RUN) r = union_bitmap_run(a, b);
    else r = union_bitmap_bitmap(a, b);
  }
  orc_free(ea);
  orc_free(eb);
  return r;
}

/* Container.unionInPlace, roaring.go:3470-3525, with the in-place kernels
 * unionBitmapBitmapInPlace :5473, unionBitmapArrayInPlace :5440, unionBitmapRunInPlace
 * :5222, unionArrayArrayInPlace :5061 (merge then optimize()), unionRunRunInPlace :5496
 * (merge; > 2048 runs -> bitmap).  Returned with a valid n, i.e. as after Repair()
 * (roaring.go:4181). */
orc_container* orc_union_in_place(const orc_container* c, const orc_container* other) {
  int32_t cn = orc_n(c), on = orc_n(other);
  if (cn == MAXV + 1) return full_container();
  if (cn == 0) return orc_clone(other);
  if (on == MAXV + 1) return full_container();
  if (on == 0) return orc_clone(c);
  if (c->typ == ORC_ARRAY && other->typ == ORC_ARRAY) {
    orc_container* m = union_array_array(c, other);
    orc_container* o = orc_optimize(m);
    orc_free(m);
    return o;
  }
  if (c->typ == ORC_RUN && other->typ == ORC_RUN) return union_run_run(c, other);
  /* every other pairing first turns c into a bitmap and ORs into it */
  orc_container* t = c->typ == ORC_ARRAY ? orc_array_to_bitmap(c) : c->typ == ORC_RUN ? orc_run_to_bitmap(c) : orc_clone(c);
  uint64_t ow[BN];
  orc_to_words(other, ow);
  int32_t n = 0;
  for (int i = 0; i < BN; i++) {
    BMP(t)[i] |= ow[i];
    n += popcnt(BMP(t)[i]);
  }
  t->n = n; /* Repair */
  return t;
}

/* ---- difference --------------------------------------------------------------------------------- */

/* differenceArrayArray, roaring.go:5730-5754.  output.add() appends in ascending order;
 * arrayAdd converts to a bitmap once the array already holds 4096 values and the value
 * is not a plain append (roaring.go:3248-3277) — with ascending appends the fast path at
 * :3251 stops at N == 4096, after which :3264 converts to a bitmap. */
static orc_container* difference_array_array(const orc_container* a, const orc_container* b) {
  u16vec out = {0};
  orc_container* bm = NULL; /* set once the output overflowed into a bitmap */
  int32_t na = a->len, nb = b->len;
  for (int32_t i = 0, j = 0; i < na;) {
    uint16_t va = ARR(a)[i];
    int keep = 0;
    if (j >= nb) {
      keep = 1;
      i++;
    } else {
      uint16_t vb = ARR(b)[j];
      if (va < vb) {
        keep = 1;
        i++;
      } else if (va > vb) j++;
      else {
        i++;
        j++;
      }
    }
    if (keep) {
      if (bm) {
        if (!bitmap_contains(BMP(bm), va)) {
          BMP(bm)[va / 64] |= 1ull << (va % 64);
          bm->n++;
        }
      } else if (out.n >= ORC_ARRAY_MAX_SIZE) {
        orc_container* t = array_from_vec(&out);
        bm = orc_array_to_bitmap(t);
        orc_free(t);
        memset(&out, 0, sizeof(out));
        BMP(bm)[va / 64] |= 1ull << (va % 64);
        bm->n++;
      } else {
        u16_push(&out, va);
      }
    }
  }
  return bm ? bm : array_from_vec(&out);
}
/* differenceArrayRun, roaring.go:5757-5799 */
static orc_container* difference_array_run(const orc_container* a, const orc_container* b) {
  u16vec out = {0};
  int32_t i = 0, j = 0;
  const uint16_t* aa = ARR(a);
  const orc_interval16* rb = RUNS(b);
  while (i < a->len) {
    if (aa[i] < rb[j].start) {
      u16_push(&out, aa[i]);
      i++;
      continue;
    }
    if (aa[i] >= rb[j].start && aa[i] <= rb[j].last) {
      i++;
      continue;
    }
    if (aa[i] > rb[j].last) {
      j++;
      if (j == b->len) break;
    }
  }
  for (; i < a->len; i++) u16_push(&out, aa[i]);
  return array_from_vec(&out);
}
/* differenceBitmapRun, roaring.go:5802-5809 */
static orc_container* difference_bitmap_run(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_clone(a);
  for (int32_t k = 0; k < b->len; k++) orc_bitmap_zero_range(o, RUNS(b)[k].start, (uint64_t)RUNS(b)[k].last + 1);
  return o;
}
/* differenceRunArray, roaring.go:5813-5863 */
static orc_container* difference_run_array(const orc_container* a, const orc_container* b) {
  const orc_interval16* ra = RUNS(a);
  const uint16_t* ab = ARR(b);
  int32_t nb = b->len;
  ivvec runs = {0};
  int32_t bidx = 0;
  uint16_t vb = ab[bidx];
  for (int32_t r = 0; r < a->len; r++) {
    orc_interval16 run = ra[r];
    uint16_t start = run.start;
    while (vb < run.start) {
      bidx++;
      if (bidx >= nb) break;
      vb = ab[bidx];
    }
    while (vb >= run.start && vb <= run.last) {
      if (vb == start) {
        if (vb == 65535) goto done; /* break RUNLOOP */
        start++;
        bidx++;
        if (bidx >= nb) break;
        vb = ab[bidx];
        continue;
      }
      orc_interval16 iv = {start, (uint16_t)(vb - 1)};
      iv_push(&runs, iv);
      if (vb == 65535) goto done;
      start = (uint16_t)(vb + 1);
      bidx++;
      if (bidx >= nb) break;
      vb = ab[bidx];
    }
    if (start <= run.last) {
      orc_interval16 iv = {start, run.last};
      iv_push(&runs, iv);
    }
  }
done:;
  orc_container* o = run_from_vec(&runs, -1);
  orc_container* t = orc_optimize(o);
  orc_free(o);
  return t;
}
/* differenceBitmapBitmap, roaring.go:6027-6050 */
static orc_container* difference_bitmap_bitmap(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_new_bitmap(NULL, 0);
  int32_t n = 0;
  for (int i = 0; i < BN; i++) {
    BMP(o)[i] = BMP(a)[i] & ~BMP(b)[i];
    n += popcnt(BMP(o)[i]);
  }
  o->n = n;
  if (n < ORC_ARRAY_MAX_SIZE) {
    orc_container* t = orc_bitmap_to_array(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* flipBitmap, roaring.go:4239-4250 */
static orc_container* flip_bitmap(const orc_container* b) {
  orc_container* o = orc_new_bitmap(NULL, 0);
  int32_t n = 0;
  for (int i = 0; i < BN; i++) {
    BMP(o)[i] = ~BMP(b)[i];
    n += popcnt(BMP(o)[i]);
  }
  o->n = n;
  return o;
}
/* differenceRunBitmap, roaring.go:5866-5927 (bit-serial walk over every run) */
static orc_container* difference_run_bitmap(const orc_container* a, const orc_container* b) {
  const orc_interval16* ra = RUNS(a);
  if (a->len > 0 && ra[0].start == 0 && ra[0].last == 65535) return flip_bitmap(b);
  const uint64_t* bb = BMP(b);
  ivvec runs = {0};
  for (int32_t r = 0; r < a->len; r++) {
    orc_interval16 in = ra[r], run = ra[r];
    int add = 1;
    for (uint16_t bit = in.start; bit <= in.last; bit++) {
      if ((bb[bit >> 6] >> (bit & 63)) & 1) {
        if (run.start == bit) {
          if (bit == 65535) add = 0;
          run.start++;
        } else if (bit == run.last) {
          run.last--;
        } else {
          run.last = (uint16_t)(bit - 1);
          if (run.last >= run.start) {
            if (runs.n >= ORC_RUN_MAX_SIZE) {
              free(runs.v);
              orc_container* asb = orc_run_to_bitmap(a);
              orc_container* o = difference_bitmap_bitmap(asb, b);
              orc_free(asb);
              return o;
            }
            iv_push(&runs, run);
          }
          run.start = (uint16_t)(bit + 1);
          run.last = in.last;
        }
        if (run.start > run.last) break;
      }
      if (bit == 65535) break;
    }
    if (run.start <= run.last) {
      if (add) {
        if (runs.n >= ORC_RUN_MAX_SIZE) {
          free(runs.v);
          orc_container* asb = orc_run_to_bitmap(a);
          orc_container* o = difference_bitmap_bitmap(asb, b);
          orc_free(asb);
          return o;
        }
        iv_push(&runs, run);
      }
    }
  }
  int32_t nruns = runs.n;
  orc_container* o = run_from_vec(&runs, -1);
  if (o->n < ORC_ARRAY_MAX_SIZE && nruns > o->n / 2) {
    orc_container* t = orc_run_to_array(o);
    orc_free(o);
    return t;
  } else if (nruns > ORC_RUN_MAX_SIZE) {
    orc_container* t = orc_run_to_bitmap(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* differenceRunRun, roaring.go:5931-5989 */
static orc_container* difference_run_run(const orc_container* a, const orc_container* b) {
  const orc_interval16 *ra = RUNS(a), *rb = RUNS(b);
  int32_t apos = 0, bpos = 0, alen = a->len, blen = b->len;
  uint16_t astart = ra[0].start, alast = ra[0].last, bstart = rb[0].start, blast = rb[0].last;
  ivvec runs = {0};
  while (apos < alen && bpos < blen) {
    if (alast < bstart) {
      orc_interval16 iv = {astart, alast};
      iv_push(&runs, iv);
      apos++;
      if (apos < alen) {
        astart = ra[apos].start;
        alast = ra[apos].last;
      }
    } else if (blast < astart) {
      bpos++;
      if (bpos < blen) {
        bstart = rb[bpos].start;
        blast = rb[bpos].last;
      }
    } else {
      if (astart < bstart) {
        orc_interval16 iv = {astart, (uint16_t)(bstart - 1)};
        iv_push(&runs, iv);
      }
      if (alast > blast) {
        astart = (uint16_t)(blast + 1);
      } else {
        apos++;
        if (apos < alen) {
          astart = ra[apos].start;
          alast = ra[apos].last;
        }
      }
    }
  }
  if (apos < alen) {
    orc_interval16 iv = {astart, alast};
    iv_push(&runs, iv);
    apos++;
    for (; apos < alen; apos++) iv_push(&runs, ra[apos]);
  }
  return run_from_vec(&runs, -1);
}
/* differenceArrayBitmap, roaring.go:5991-6006 */
static orc_container* difference_array_bitmap(const orc_container* a, const orc_container* b) {
  u16vec out = {0};
  for (int32_t k = 0; k < a->len; k++) {
    uint16_t va = ARR(a)[k];
    if ((1ull << (va % 64)) & ~BMP(b)[va / 64]) u16_push(&out, va);
  }
  return array_from_vec(&out);
}
/* differenceBitmapArray, roaring.go:6008-6025 */
static orc_container* difference_bitmap_array(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_clone(a);
  int32_t n = o->n;
  for (int32_t k = 0; k < b->len; k++) {
    uint16_t v = ARR(b)[k];
    if (bitmap_contains(BMP(o), v)) {
      BMP(o)[v / 64] &= ~(1ull << (v % 64));
      n--;
    }
  }
  o->n = n;
  if (n < ORC_ARRAY_MAX_SIZE) {
    orc_container* t = orc_bitmap_to_array(o);
    orc_free(o);
    return t;
  }
  return o;
}

/* difference, roaring.go:5692-5727 */
orc_container* orc_difference(const orc_container* a, const orc_container* b) {
  if (orc_n(a) == 0 || orc_n(b) == MAXV + 1) return NULL;
  if (orc_n(b) == 0) return orc_clone(a); /* a.Freeze() */
  if (a->typ == ORC_ARRAY) {
    if (b->typ == ORC_ARRAY) return difference_array_array(a, b);
    if (b->typ == ORC_RUN) return difference_array_run(a, b);
    return difference_array_bitmap(a, b);
  } else if (a->typ == ORC_RUN) {
    if (b->typ == ORC_ARRAY) return difference_run_array(a, b);
    if (b->typ == ORC_RUN) return difference_run_run(a, b);
    return difference_run_bitmap(a, b);
  } else {
    if (b->typ == ORC_ARRAY) return difference_bitmap_array(a, b);
    if (b->typ == ORC_RUN) return difference_bitmap_run(a, b);
    return difference_bitmap_bitmap(a, b);
  }
}

/* ---- xor -------------------------------------------------------------------------------------------- */

/* xorArrayArray, roaring.go:6089-6133 */
static orc_container* xor_array_array(const orc_container* a, const orc_container* b) {
  const uint16_t *aa = ARR(a), *ab = ARR(b);
  int32_t la = a->len, lb = b->len, i = 0, j = 0;
  u16vec out = {0};
  while (i < la && j < lb) {
    uint16_t va = aa[i], vb = ab[j];
    if (va < vb) {
      while (i < la && aa[i] < vb) u16_push(&out, aa[i++]);
    } else if (va > vb) {
      while (j < lb && ab[j] < va) u16_push(&out, ab[j++]);
    } else {
      i++;
      j++;
    }
  }
  if (i < la) {
    for (; i < la; i++) u16_push(&out, aa[i]);
  } else if (j < lb) {
    for (; j < lb; j++) u16_push(&out, ab[j]);
  }
  return array_from_vec(&out);
}
/* xorArrayBitmap, roaring.go:6135-6153: per element remove/add on a clone of b
 * (bitmapRemove converts to an array the moment N reaches 4096, roaring.go:3632-3636;
 * arrayAdd converts back to a bitmap when it would exceed 4096, :3264).  Only the final
 * representation is observable: the walk below tracks the same type transitions. */
static orc_container* xor_array_bitmap(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_clone(b); /* bitmap */
  for (int32_t k = 0; k < a->len; k++) {
    uint16_t v = ARR(a)[k];
    int in_b = bitmap_contains(BMP(b), v);
    if (o == NULL) { /* remove() returned nil: only when the last bit was removed */
      if (!in_b) { /* add on nil container -> new array {v} (roaring.go:3236) */
        o = orc_new_array(&v, 1);
      }
      continue;
    }
    if (o->typ == ORC_BITMAP) {
      if (in_b) { /* bitmapRemove, roaring.go:3618-3638 */
        if (bitmap_contains(BMP(o), v)) {
          if (o->n == 1) {
            orc_free(o);
            o = NULL;
            continue;
          }
          BMP(o)[v / 64] &= ~(1ull << (v % 64));
          o->n--;
          if (o->n == ORC_ARRAY_MAX_SIZE) {
            orc_container* t = orc_bitmap_to_array(o);
            orc_free(o);
            o = t;
          }
        }
      } else { /* bitmapAdd, roaring.go:3280-3294 */
        if (!bitmap_contains(BMP(o), v)) {
          BMP(o)[v / 64] |= 1ull << (v % 64);
          o->n++;
        }
      }
    } else { /* array container: arrayRemove :3597 / arrayAdd :3248 */
      uint16_t* arr = (uint16_t*)o->data;
      int32_t lo = 0, hi = o->len;
      while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (arr[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      int found = lo < o->len && arr[lo] == v;
      if (in_b) {
        if (found) {
          if (o->n == 1) {
            orc_free(o);
            o = NULL;
            continue;
          }
          memmove(arr + lo, arr + lo + 1, (size_t)(o->len - lo - 1) * 2);
          o->len--;
          o->n--;
        }
      } else if (!found) {
        if (o->n >= ORC_ARRAY_MAX_SIZE) {
          orc_container* t = orc_array_to_bitmap(o);
          orc_free(o);
          o = t;
          BMP(o)[v / 64] |= 1ull << (v % 64);
          o->n++;
        } else {
          arr = (uint16_t*)realloc(arr, (size_t)(o->len + 1) * 2);
          if (!arr) abort();
          memmove(arr + lo + 1, arr + lo, (size_t)(o->len - lo) * 2);
          arr[lo] = v;
          o->data = arr;
          o->len++;
          o->n++;
        }
      }
    }
  }
  if (o && o->typ == ORC_BITMAP && orc_count(o) < ORC_ARRAY_MAX_SIZE) {
    orc_container* t = orc_bitmap_to_array(o);
    orc_free(o);
    o = t;
  }
  return o;
}
/* xorBitmapBitmap, roaring.go:6155-6179 */
static orc_container* xor_bitmap_bitmap(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_new_bitmap(NULL, 0);
  int32_t n = 0;
  for (int i = 0; i < BN; i++) {
    BMP(o)[i] = BMP(a)[i] ^ BMP(b)[i];
    n += popcnt(BMP(o)[i]);
  }
  o->n = n;
  if (n < ORC_ARRAY_MAX_SIZE) {
    orc_container* t = orc_bitmap_to_array(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* xorArrayRun, roaring.go:6607-6671 */
static orc_container* xor_array_run(const orc_container* a, const orc_container* b) {
  ivvec out = {0};
  int32_t na = a->len, nb = b->len, n = 0;
  orc_interval16 vb = {0, 0};
  uint16_t va = 0;
  int32_t lastI = -1, lastJ = -1;
  for (int32_t i = 0, j = 0; i < na || j < nb;) {
    if (i < na && i != lastI) va = ARR(a)[i];
    if (j < nb && j != lastJ) vb = RUNS(b)[j];
    lastI = i;
    lastJ = j;
    if (i < na && (j >= nb || va < vb.start)) { /* before */
      orc_interval16 iv = {va, va};
      n += run_append_interval(&out, iv);
      i++;
    } else if (j < nb && (i >= na || va > vb.last)) { /* after */
      n += run_append_interval(&out, vb);
      j++;
    } else if (va > vb.start) {
      if (va < vb.last) {
        orc_interval16 iv = {vb.start, (uint16_t)(va - 1)};
        n += run_append_interval(&out, iv);
        i++;
        vb.start = (uint16_t)(va + 1);
        if (vb.start > vb.last) j++;
      } else if (va > vb.last) {
        n += run_append_interval(&out, vb);
        j++;
      } else { /* va == vb.last */
        vb.last--;
        if (vb.start <= vb.last) n += run_append_interval(&out, vb);
        j++;
        i++;
      }
    } else { /* va == vb.start */
      if (vb.start == MAXV) {
        j++;
      } else {
        vb.start++;
        if (vb.start > vb.last) j++;
      }
      i++;
    }
  }
  int32_t nruns = out.n;
  orc_container* o = run_from_vec(&out, n);
  if (n < ORC_ARRAY_MAX_SIZE) {
    orc_container* t = orc_run_to_array(o);
    orc_free(o);
    return t;
  } else if (nruns > ORC_RUN_MAX_SIZE) {
    orc_container* t = orc_run_to_bitmap(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* xorstm / xorCompare, roaring.go:6675-6763 */
typedef struct {
  int va_valid, vb_valid;
  orc_interval16 va, vb;
} xorstm;
static int xor_compare(xorstm* x, orc_interval16* r1) {
  int has = 0;
  if (!x->va_valid || !x->vb_valid) {
    if (x->vb_valid) {
      x->vb_valid = 0;
      *r1 = x->vb;
      return 1;
    }
    if (x->va_valid) {
      x->va_valid = 0;
      *r1 = x->va;
      return 1;
    }
    return 0;
  }
  if (x->va.last < x->vb.start) { /* va before */
    x->va_valid = 0;
    *r1 = x->va;
    has = 1;
  } else if (x->vb.last < x->va.start) { /* vb before */
    x->vb_valid = 0;
    *r1 = x->vb;
    has = 1;
  } else if (x->va.start == x->vb.start && x->va.last == x->vb.last) { /* equal */
    x->va_valid = 0;
    x->vb_valid = 0;
  } else if (x->va.start <= x->vb.start && x->va.last >= x->vb.last) { /* vb inside */
    x->vb_valid = 0;
    if (x->va.start != x->vb.start) {
      r1->start = x->va.start;
      r1->last = (uint16_t)(x->vb.start - 1);
      has = 1;
    }
    if (x->vb.last == MAXV) {
      x->va_valid = 0;
    } else {
      x->va.start = (uint16_t)(x->vb.last + 1);
      if (x->va.start > x->va.last) x->va_valid = 0;
    }
  } else if (x->vb.start <= x->va.start && x->vb.last >= x->va.last) { /* va inside */
    x->va_valid = 0;
    if (x->vb.start != x->va.start) {
      r1->start = x->vb.start;
      r1->last = (uint16_t)(x->va.start - 1);
      has = 1;
    }
    if (x->va.last == MAXV) {
      x->vb_valid = 0;
    } else {
      x->vb.start = (uint16_t)(x->va.last + 1);
      if (x->vb.start > x->vb.last) x->vb_valid = 0;
    }
  } else if (x->va.start < x->vb.start && x->va.last <= x->vb.last) { /* va first overlap */
    x->va_valid = 0;
    r1->start = x->va.start;
    r1->last = (uint16_t)(x->vb.start - 1);
    has = 1;
    if (x->va.last == MAXV) {
      x->vb_valid = 0;
    } else {
      x->vb.start = (uint16_t)(x->va.last + 1);
      if (x->vb.start > x->vb.last) x->vb_valid = 0;
    }
  } else if (x->vb.start < x->va.start && x->vb.last <= x->va.last) { /* vb first overlap */
    x->vb_valid = 0;
    r1->start = x->vb.start;
    r1->last = (uint16_t)(x->va.start - 1);
    has = 1;
    if (x->vb.last == MAXV) {
      x->va_valid = 0;
    } else {
      x->va.start = (uint16_t)(x->vb.last + 1);
      if (x->va.start > x->va.last) x->va_valid = 0;
    }
  }
  return has;
}
/* xorRunRun, roaring.go:6769-6813 */
static orc_container* xor_run_run(const orc_container* a, const orc_container* b) {
  int32_t na = a->len, nb = b->len, n = 0;
  ivvec out = {0};
  int32_t lastI = -1, lastJ = -1;
  xorstm st;
  memset(&st, 0, sizeof(st));
  for (int32_t i = 0, j = 0; i < na || j < nb;) {
    if (i < na && lastI != i) {
      st.va = RUNS(a)[i];
      st.va_valid = 1;
    }
    if (j < nb && lastJ != j) {
      st.vb = RUNS(b)[j];
      st.vb_valid = 1;
    }
    lastI = i;
    lastJ = j;
    orc_interval16 r1 = {0, 0};
    if (xor_compare(&st, &r1)) n += run_append_interval(&out, r1);
    if (!st.va_valid) i++;
    if (!st.vb_valid) j++;
  }
  int32_t l = out.n;
  orc_container* o = run_from_vec(&out, n);
  if (n < ORC_ARRAY_MAX_SIZE && l > n / 2) {
    orc_container* t = orc_run_to_array(o);
    orc_free(o);
    return t;
  } else if (l > ORC_RUN_MAX_SIZE) {
    orc_container* t = orc_run_to_bitmap(o);
    orc_free(o);
    return t;
  }
  return o;
}
/* xorBitmapRun, roaring.go:6816-6825 */
static orc_container* xor_bitmap_run(const orc_container* a, const orc_container* b) {
  orc_container* o = orc_clone(a);
  for (int32_t k = 0; k < b->len; k++) orc_bitmap_xor_range(o, RUNS(b)[k].start, (uint64_t)RUNS(b)[k].last + 1);
  return o;
}

/* xor, roaring.go:6052-6087 */
orc_container* orc_xor(const orc_container* a, const orc_container* b) {
  if (orc_n(a) == 0) return orc_clone(b); /* b.Freeze() */
  if (orc_n(b) == 0) return orc_clone(a);
  if (a->typ == ORC_ARRAY) {
    if (b->typ == ORC_ARRAY) return xor_array_array(a, b);
    if (b->typ == ORC_RUN) return xor_array_run(a, b);
    return xor_array_bitmap(a, b);
  } else if (a->typ == ORC_RUN) {
    if (b->typ == ORC_ARRAY) return xor_array_run(b, a);
    if (b->typ == ORC_RUN) return xor_run_run(a, b);
    return xor_bitmap_run(b, a);
  } else {
    if (b->typ == ORC_ARRAY) return xor_array_bitmap(b, a);
    if (b->typ == ORC_RUN) return xor_bitmap_run(a, b);
    return xor_bitmap_bitmap(a, b);
  }
}

/* Container.BitwiseCompare, roaring.go:5396-5422: equal N and equal bit content,
 * whatever the encodings.  Returns 0 when equal, else 1 + number of differing bits. */
int32_t orc_bitwise_compare(const orc_container* a, const orc_container* b) {
  if (orc_n(a) != orc_n(b)) return -1;
  if (orc_n(a) == 0) return 0;
  uint64_t wa[BN], wb[BN];
  orc_to_words(a, wa);
  orc_to_words(b, wb);
  int32_t d = 0;
  for (int i = 0; i < BN; i++) d += popcnt(wa[i] ^ wb[i]);
  return d ? 1 + d : 0;
}

/* ---- Bitmap level ------------------------------------------------------------------------------------ */

orc_bitmap* orc_bitmap_new(void) { return (orc_bitmap*)xcalloc(1, sizeof(orc_bitmap)); }
void orc_bitmap_free(orc_bitmap* b) {
  if (!b) return;
  for (int32_t i = 0; i < b->len; i++) orc_free(b->cs[i]);
  free(b->keys);
  free(b->cs);
  free(b);
}
void orc_bitmap_put(orc_bitmap* b, uint64_t key, orc_container* c) {
  if (b->len == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 16;
    b->keys = (uint64_t*)realloc(b->keys, (size_t)b->cap * 8);
    b->cs = (orc_container**)realloc(b->cs, (size_t)b->cap * sizeof(*b->cs));
    if (!b->keys || !b->cs) abort();
  }
  b->keys[b->len] = key;
  b->cs[b->len] = c;
  b->len++;
}
int32_t orc_bitmap_len(const orc_bitmap* b) { return b->len; }
uint64_t orc_bitmap_key(const orc_bitmap* b, int32_t i) { return b->keys[i]; }
const orc_container* orc_bitmap_container(const orc_bitmap* b, int32_t i) { return b->cs[i]; }

/* sliceIterator.Next skips nil containers, containers_slice.go:233-255 */
static int32_t it_next(const orc_bitmap* b, int32_t i) {
  i++;
  while (i < b->len && b->cs[i] == NULL) i++;
  return i;
}

/* Bitmap.Count, roaring.go:542; sliceContainers.Count, containers_slice.go:120-126 */
uint64_t orc_bitmap_count(const orc_bitmap* b) {
  uint64_t n = 0;
  for (int32_t i = 0; i < b->len; i++) n += (uint64_t)orc_n(b->cs[i]);
  return n;
}

/* Bitmap.CountRange, roaring.go:573-621 */
uint64_t orc_bitmap_count_range(const orc_bitmap* b, uint64_t start, uint64_t end) {
  uint64_t n = 0;
  if (b->len == 0) return 0;
  uint64_t skey = start >> 16, ekey = end >> 16;
  /* Iterator(skey): position at first key >= skey, found = exact match */
  int32_t i = 0;
  while (i < b->len && b->keys[i] < skey) i++;
  int found = i < b->len && b->keys[i] == skey;
  i = i - 1;
  i = it_next(b, i);
  if (found && skey == ekey) {
    if (i >= b->len) return 0;
    return (uint64_t)orc_count_range(b->cs[i], (int32_t)(start & 0xffff), (int32_t)(end & 0xffff));
  }
  for (; i < b->len; i = it_next(b, i)) {
    uint64_t k = b->keys[i];
    const orc_container* c = b->cs[i];
    if (k > ekey) break;
    if (k == skey) {
      n += (uint64_t)orc_count_range(c, (int32_t)(start & 0xffff), MAXV + 1);
      continue;
    }
    if (k < ekey) {
      n += (uint64_t)orc_n(c);
      continue;
    }
    if (k == ekey) {
      n += (uint64_t)orc_count_range(c, 0, (int32_t)(end & 0xffff));
      break;
    }
  }
  return n;
}

/* Bitmap.IntersectionCount, roaring.go:711-733 */
uint64_t orc_bitmap_intersection_count(const orc_bitmap* a, const orc_bitmap* b) {
  uint64_t n = 0;
  int32_t i = it_next(a, -1), j = it_next(b, -1);
  while (i < a->len && j < b->len) {
    if (a->keys[i] < b->keys[j]) i = it_next(a, i);
    else if (a->keys[i] > b->keys[j]) j = it_next(b, j);
    else {
      n += (uint64_t)orc_intersection_count(a->cs[i], b->cs[j]);
      i = it_next(a, i);
      j = it_next(b, j);
    }
  }
  return n;
}

/* Bitmap.Intersect, roaring.go:736-759 (stores the possibly-nil result under every
 * common key) */
orc_bitmap* orc_bitmap_intersect(const orc_bitmap* a, const orc_bitmap* b) {
  orc_bitmap* o = orc_bitmap_new();
  int32_t i = it_next(a, -1), j = it_next(b, -1);
  while (i < a->len && j < b->len) {
    if (a->keys[i] < b->keys[j]) i = it_next(a, i);
    else if (a->keys[i] > b->keys[j]) j = it_next(b, j);
    else {
      orc_bitmap_put(o, a->keys[i], orc_intersect(a->cs[i], b->cs[j]));
      i = it_next(a, i);
      j = it_next(b, j);
    }
  }
  return o;
}

/* Bitmap.unionIntoTargetSingle, roaring.go:1292-1315 */
static orc_bitmap* bitmap_union_single(const orc_bitmap* a, const orc_bitmap* b) {
  orc_bitmap* o = orc_bitmap_new();
  int32_t i = it_next(a, -1), j = it_next(b, -1);
  while (i < a->len || j < b->len) {
    int hi = i < a->len, hj = j < b->len;
    if (hi && (!hj || a->keys[i] < b->keys[j])) {
      orc_bitmap_put(o, a->keys[i], orc_clone(a->cs[i]));
      i = it_next(a, i);
    } else if (hj && (!hi || a->keys[i] > b->keys[j])) {
      orc_bitmap_put(o, b->keys[j], orc_clone(b->cs[j]));
      j = it_next(b, j);
    } else {
      orc_bitmap_put(o, a->keys[i], orc_union(a->cs[i], b->cs[j]));
      i = it_next(a, i);
      j = it_next(b, j);
    }
  }
  return o;
}

/* Bitmap.Union, roaring.go:1272-1284, and the n-way Bitmap.unionInPlace,
 * roaring.go:1410-1561 with summary stats :7011-7066.  Restated per key (the Go code
 * advances all iterators in lock step over ascending keys, which visits every key of the
 * union exactly once, in the order target-then-others):
 *   target full                          -> unchanged                       (:1455-1461)
 *   any operand full                     -> fullContainer                   (:1465-1474)
 *   no target, exactly one operand       -> that operand                    (:1484-1490)
 *   no target, expectedN >= 512 and first operand not a bitmap
 *                                        -> start from an empty bitmap      (:1500-1505)
 *   no target otherwise                  -> start from first operand        (:1506-1513)
 *   target present, expectedN >= 512     -> target converted to bitmap      (:1521-1529)
 *   then tContainer = tContainer.unionInPlace(operand) for each operand     (:1538)
 *   finally Repair(): recount, drop nil containers                          (:1560)     */
orc_bitmap* orc_bitmap_union(const orc_bitmap* a, const orc_bitmap* const* others, int32_t n_others) {
  if (n_others == 1) return bitmap_union_single(a, others[0]);
  orc_bitmap* o = orc_bitmap_new();
  int32_t ia = it_next(a, -1);
  int32_t* pos = (int32_t*)xmalloc((size_t)(n_others ? n_others : 1) * sizeof(int32_t));
  for (int32_t k = 0; k < n_others; k++) pos[k] = it_next(others[k], -1);
  for (;;) {
    /* smallest pending key */
    int have = 0;
    uint64_t key = 0;
    if (ia < a->len) {
      key = a->keys[ia];
      have = 1;
    }
    for (int32_t k = 0; k < n_others; k++)
      if (pos[k] < others[k]->len && (!have || others[k]->keys[pos[k]] < key)) {
        key = others[k]->keys[pos[k]];
        have = 1;
      }
    if (!have) break;
    const orc_container* tgt = (ia < a->len && a->keys[ia] == key) ? a->cs[ia] : NULL;
    /* operands with this key, in bitmap order */
    int32_t nops = 0, any_full = 0;
    int64_t expected = tgt ? orc_n(tgt) : 0;
    const orc_container* first = NULL;
    for (int32_t k = 0; k < n_others; k++)
      if (pos[k] < others[k]->len && others[k]->keys[pos[k]] == key) {
        const orc_container* c = others[k]->cs[pos[k]];
        if (!first) first = c;
        nops++;
        expected += orc_n(c);
        if (orc_n(c) == MAXV + 1) any_full = 1;
      }
    orc_container* res;
    if (tgt && orc_n(tgt) == MAXV + 1) {
      res = orc_clone(tgt);
    } else if (nops == 0) {
      res = orc_clone(tgt);
    } else if (any_full) {
      res = full_container();
    } else {
      int skip_first = 0;
      if (!tgt) {
        if (nops == 1) {
          res = orc_clone(first);
          skip_first = -1; /* nothing left to union */
        } else if (expected >= 512 && first->typ != ORC_BITMAP) {
          res = orc_new_bitmap(NULL, 0);
        } else {
          res = orc_clone(first);
          skip_first = 1;
        }
      } else {
        if (expected >= 512 && tgt->typ != ORC_BITMAP)
          res = tgt->typ == ORC_ARRAY ? orc_array_to_bitmap(tgt) : orc_run_to_bitmap(tgt);
        else
          res = orc_clone(tgt);
      }
      if (skip_first >= 0) {
        int seen = 0;
        for (int32_t k = 0; k < n_others; k++)
          if (pos[k] < others[k]->len && others[k]->keys[pos[k]] == key) {
            if (skip_first == 1 && !seen) {
              seen = 1;
              continue;
            }
            seen = 1;
            orc_container* t = orc_union_in_place(res, others[k]->cs[pos[k]]);
            orc_free(res);
            res = t;
          }
      }
    }
    if (res) orc_bitmap_put(o, key, res); /* Repair drops nil containers */
    if (ia < a->len && a->keys[ia] == key) ia = it_next(a, ia);
    for (int32_t k = 0; k < n_others; k++)
      if (pos[k] < others[k]->len && others[k]->keys[pos[k]] == key) pos[k] = it_next(others[k], pos[k]);
  }
  free(pos);
  return o;
}

/* Bitmap.singleDifference, roaring.go:1573-1595 */
static orc_bitmap* bitmap_single_difference(const orc_bitmap* a, const orc_bitmap* b) {
  orc_bitmap* o = orc_bitmap_new();
  int32_t i = it_next(a, -1), j = it_next(b, -1);
  while (i < a->len || j < b->len) {
    int hi = i < a->len, hj = j < b->len;
    if (hi && (!hj || a->keys[i] < b->keys[j])) {
      orc_bitmap_put(o, a->keys[i], orc_clone(a->cs[i]));
      i = it_next(a, i);
    } else if (hj && (!hi || a->keys[i] > b->keys[j])) {
      j = it_next(b, j);
    } else {
      orc_bitmap_put(o, a->keys[i], orc_difference(a->cs[i], b->cs[j]));
      i = it_next(a, i);
      j = it_next(b, j);
    }
  }
  return o;
}
/* Bitmap.Difference, roaring.go:1564-1571.  Further subtrahends go through
 * DifferenceInPlace (roaring.go:7090-7172), whose bit content equals subtracting them
 * one after another; that is how it is restated here. */
orc_bitmap* orc_bitmap_difference(const orc_bitmap* a, const orc_bitmap* const* others, int32_t n_others) {
  orc_bitmap* o = bitmap_single_difference(a, others[0]);
  for (int32_t k = 1; k < n_others; k++) {
    orc_bitmap* t = bitmap_single_difference(o, others[k]);
    orc_bitmap_free(o);
    o = t;
  }
  return o;
}

/* Bitmap.Xor, roaring.go:1598-1623 */
orc_bitmap* orc_bitmap_xor(const orc_bitmap* a, const orc_bitmap* b) {
  orc_bitmap* o = orc_bitmap_new();
  int32_t i = it_next(a, -1), j = it_next(b, -1);
  while (i < a->len || j < b->len) {
    int hi = i < a->len, hj = j < b->len;
    if (hi && (!hj || a->keys[i] < b->keys[j])) {
      orc_bitmap_put(o, a->keys[i], orc_clone(a->cs[i]));
      i = it_next(a, i);
    } else if (hj && (!hi || a->keys[i] > b->keys[j])) {
      orc_bitmap_put(o, b->keys[j], orc_clone(b->cs[j]));
      j = it_next(b, j);
    } else {
      orc_bitmap_put(o, a->keys[i], orc_xor(a->cs[i], b->cs[j]));
      i = it_next(a, i);
      j = it_next(b, j);
    }
  }
  return o;
}

/* ---- bulk helpers for the CPU baseline ----------------------------------------------------------- */

/* One dense row pair = 16 calls of intersectionCountBitmapBitmap -> popcountAndSlice
 * (roaring.go:4611, 6928-6939) under Bitmap.IntersectionCount's key merge (:711). */
uint64_t orc_dense_intersection_count(const uint64_t* a, const uint64_t* b, uint64_t n_pairs, uint64_t* out_counts) {
  uint64_t total = 0;
  for (uint64_t p = 0; p < n_pairs; p++) {
    uint64_t n = 0;
    for (int s = 0; s < 16; s++) {
      const uint64_t* x = a + (p * 16 + s) * BN;
      const uint64_t* y = b + (p * 16 + s) * BN;
      uint64_t c = 0;
      for (int i = 0; i < BN; i++) c += popcnt(x[i] & y[i]);
      n += (uint64_t)(int32_t)c;
    }
    if (out_counts) out_counts[p] = n;
    total += n;
  }
  return total;
}

/* Bitmap.Intersect -> intersectBitmapBitmap (roaring.go:4960-4978: AND + popcount fused,
 * fresh [1024]uint64 per output container) followed by Count (:542). */
uint64_t orc_dense_intersect_count(const uint64_t* a, const uint64_t* b, uint64_t n_pairs, uint64_t* out_rows,
                                   uint64_t* out_counts) {
  uint64_t total = 0;
  for (uint64_t p = 0; p < n_pairs; p++) {
    uint64_t n = 0;
    for (int s = 0; s < 16; s++) {
      const uint64_t* x = a + (p * 16 + s) * BN;
      const uint64_t* y = b + (p * 16 + s) * BN;
      uint64_t* o = out_rows + (p * 16 + s) * BN;
      int32_t c = 0;
      for (int i = 0; i < BN; i++) {
        o[i] = x[i] & y[i];
        c += popcnt(o[i]);
      }
      n += (uint64_t)c;
    }
    if (out_counts) out_counts[p] = n;
    total += n;
  }
  return total;
}

/* ---- test hooks: the reference's per-kernel tests call the type-pair kernels directly,
 * bypassing the dispatchers' short-circuits, and sometimes force a stale N
 * (roaring_internal_test.go:621 `b.setN(4097)`). -------------------------------------- */
void orc_set_n(orc_container* c, int32_t n) {
  if (c) c->n = n;
}

/* flip, roaring.go:4221-4258 (array and run containers go through a bitmap) */
orc_container* orc_flip(const orc_container* a) {
  uint64_t w[BN];
  orc_to_words(a, w);
  orc_container* t = orc_new_bitmap(w, -1);
  orc_container* o = flip_bitmap(t);
  orc_free(t);
  return o;
}

/* runAppendInterval on a run list; returns the cardinality delta (roaring.go:5155) */
int32_t orc_run_append_interval(const orc_interval16* base, int32_t len, orc_interval16 v) {
  ivvec r = {0};
  for (int32_t i = 0; i < len; i++) iv_push(&r, base[i]);
  int32_t d = run_append_interval(&r, v);
  free(r.v);
  return d;
}

orc_container* orc_kernel(const char* name, const orc_container* a, const orc_container* b) {
#define K(nm, fn) \
  if (strcmp(name, nm) == 0) return fn(a, b)
  K("intersectArrayArray", intersect_array_array);
  K("intersectArrayRun", intersect_array_run);
  K("intersectRunRun", intersect_run_run);
  K("intersectBitmapRun", intersect_bitmap_run);
  K("intersectArrayBitmap", intersect_array_bitmap);
  K("intersectBitmapBitmap", intersect_bitmap_bitmap);
  K("unionArrayArray", union_array_array);
  K("unionArrayRun", union_array_run);
  K("unionRunRun", union_run_run);
  K("unionBitmapRun", union_bitmap_run);
  K("unionArrayBitmap", union_array_bitmap);
  K("unionBitmapBitmap", union_bitmap_bitmap);
  K("differenceArrayArray", difference_array_array);
  K("differenceArrayRun", difference_array_run);
  K("differenceBitmapRun", difference_bitmap_run);
  K("differenceRunArray", difference_run_array);
  K("differenceRunBitmap", difference_run_bitmap);
  K("differenceRunRun", difference_run_run);
  K("differenceArrayBitmap", difference_array_bitmap);
  K("differenceBitmapArray", difference_bitmap_array);
  K("differenceBitmapBitmap", difference_bitmap_bitmap);
  K("xorArrayArray", xor_array_array);
  K("xorArrayBitmap", xor_array_bitmap);
  K("xorBitmapBitmap", xor_bitmap_bitmap);
  K("xorArrayRun", xor_array_run);
  K("xorRunRun", xor_run_run);
  K("xorBitmapRun", xor_bitmap_run);
#undef K
  return NULL;
}

int32_t orc_count_kernel(const char* name, const orc_container* a, const orc_container* b) {
#define K(nm, fn) \
  if (strcmp(name, nm) == 0) return fn(a, b)
  K("intersectionCountArrayArray", icount_array_array);
  K("intersectionCountArrayRun", icount_array_run);
  K("intersectionCountRunRun", icount_run_run);
  K("intersectionCountBitmapRun", icount_bitmap_run);
  K("intersectionCountArrayBitmap", icount_array_bitmap);
  K("intersectionCountBitmapBitmap", icount_bitmap_bitmap);
#undef K
  return -1;
}

/* ---- multi-threaded bulk baseline ------------------------------------------------------
 * One worker per shard chunk, as the reference runs one goroutine per shard over
 * runtime.NumCPU() pool workers (executor.go:128-147, 6723-6737).  Every worker makes
 * `passes` passes over its own chunk, so thread start-up is paid once, not per pass. */
#include <pthread.h>

typedef struct {
  const uint64_t *a, *b;
  uint64_t n_pairs, passes;
  uint64_t* out_counts;
  uint64_t total;
} mt_job;

static void* mt_worker(void* p) {
  mt_job* j = (mt_job*)p;
  uint64_t t = 0;
  for (uint64_t k = 0; k < j->passes; k++) t = orc_dense_intersection_count(j->a, j->b, j->n_pairs, j->out_counts);
  j->total = t;
  return NULL;
}

uint64_t orc_dense_intersection_count_mt(const uint64_t* a, const uint64_t* b, uint64_t n_pairs, uint64_t* out_counts,
                                         int32_t n_threads, uint64_t passes) {
  if (n_threads < 1) n_threads = 1;
  if ((uint64_t)n_threads > n_pairs) n_threads = (int32_t)(n_pairs ? n_pairs : 1);
  pthread_t* th = (pthread_t*)xmalloc((size_t)n_threads * sizeof(pthread_t));
  mt_job* jobs = (mt_job*)xcalloc((size_t)n_threads, sizeof(mt_job));
  uint64_t per = n_pairs / (uint64_t)n_threads, rem = n_pairs % (uint64_t)n_threads, at = 0;
  for (int32_t t = 0; t < n_threads; t++) {
    uint64_t cnt = per + ((uint64_t)t < rem ? 1 : 0);
    jobs[t].a = a + at * 16 * BN;
    jobs[t].b = b + at * 16 * BN;
    jobs[t].n_pairs = cnt;
    jobs[t].passes = passes;
    jobs[t].out_counts = out_counts ? out_counts + at : NULL;
    at += cnt;
    pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
  }
  uint64_t total = 0;
  for (int32_t t = 0; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    total += jobs[t].total;
  }
  free(th);
  free(jobs);
  return total;
}
