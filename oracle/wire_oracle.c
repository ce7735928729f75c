/*
 * wire_oracle.c — CPU ORACLE (test infrastructure, NOT product code): the roaring
 * serialisation formats the reference reads and writes, restated from roaring/roaring.go:
 *   writer   Bitmap.WriteTo / writeToUnoptimized        :1730-1817
 *            Container.WriteTo, array/bitmap/runWriteTo :4054-4108, size() :4110
 *   readers  NewRoaringIterator                          :2035-2050
 *            newPilosaRoaringIterator + Next             :1984-2031, 2133-2192
 *            newOfficialRoaringIterator + Next           :1945-1982, 2208-2262
 *            readOfficialHeader                          :6948-7006
 * Pilosa format: u32 cookie (12348 | version<<16 | flags<<24), u32 container count, then per
 * container {u64 key, u16 type, u16 N-1}, then per container u32 offset, then the payloads:
 * array N x u16, bitmap 1024 x u64, run u16 count + count x {u16 start, u16 last}.
 * Official format ("3A30" no-run cookie 12346 / "3B30" run cookie 12347): u16 keys, run
 * containers as {start, length-1} pairs, converted to {start, last} on read (:2239-2247).
 * Pinned by tests/test_oracle_wire.py to TestUnmarshalRoaringWithNoErrors'
 * fixtures (roaring_internal_test.go:3793-3836).
 */
#include <stdlib.h>
#include <string.h>

#include "roaring_oracle.h"

#define MAGIC_NUMBER 12348u           /* roaring.go:21 */
#define SERIAL_COOKIE_NO_RUN 12346u   /* roaring.go serialCookieNoRunContainer */
#define SERIAL_COOKIE 12347u          /* roaring.go serialCookie */
#define HEADER_BASE_SIZE 8            /* roaring.go:34 */

static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static void wr16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr32(uint8_t* p, uint32_t v) { wr16(p, (uint16_t)v); wr16(p + 2, (uint16_t)(v >> 16)); }
static void wr64(uint8_t* p, uint64_t v) { wr32(p, (uint32_t)v); wr32(p + 4, (uint32_t)(v >> 32)); }

/* Container.size(), roaring.go:4110-4121 */
static uint64_t container_size(const orc_container* c) {
  if (c->typ == ORC_ARRAY) return (uint64_t)c->len * 2;
  if (c->typ == ORC_RUN) return 2 + (uint64_t)c->len * 4;
  return 8192;
}

/* Bitmap.WriteTo: Optimize() every container, then writeToUnoptimized.  Returns a malloc'd
 * buffer (caller frees with orc_wire_free) and its length. */
uint8_t* orc_roaring_marshal(const orc_bitmap* b, int32_t optimize, uint64_t* out_len) {
  int32_t n = 0;
  orc_container** cs = (orc_container**)calloc((size_t)(b->len > 0 ? b->len : 1), sizeof(orc_container*));
  uint64_t* keys = (uint64_t*)calloc((size_t)(b->len > 0 ? b->len : 1), sizeof(uint64_t));
  uint64_t payload = 0;
  for (int32_t i = 0; i < b->len; i++) {
    const orc_container* c = b->cs[i];
    if (!c || orc_n(c) == 0) continue; /* countNonEmptyContainers / `if c.N() > 0`, :1742, 1768 */
    orc_container* o = optimize ? orc_optimize(c) : orc_clone(c);
    if (!o) continue;
    cs[n] = o;
    keys[n] = b->keys[i];
    payload += container_size(o);
    n++;
  }
  const uint64_t total = HEADER_BASE_SIZE + (uint64_t)n * 16 + payload;
  uint8_t* buf = (uint8_t*)malloc(total ? total : 1);
  wr32(buf, MAGIC_NUMBER); /* version 0, flags 0 */
  wr32(buf + 4, (uint32_t)n);
  uint8_t* h = buf + HEADER_BASE_SIZE;
  for (int32_t i = 0; i < n; i++, h += 12) {
    wr64(h, keys[i]);
    wr16(h + 8, (uint16_t)cs[i]->typ);
    wr16(h + 10, (uint16_t)(orc_n(cs[i]) - 1));
  }
  uint32_t offset = (uint32_t)(HEADER_BASE_SIZE + (uint64_t)n * 16);
  for (int32_t i = 0; i < n; i++, h += 4) {
    wr32(h, offset);
    offset += (uint32_t)container_size(cs[i]);
  }
  for (int32_t i = 0; i < n; i++) {
    const orc_container* c = cs[i];
    if (c->typ == ORC_ARRAY) {
      memcpy(h, c->data, (size_t)c->len * 2);
      h += (size_t)c->len * 2;
    } else if (c->typ == ORC_RUN) {
      wr16(h, (uint16_t)c->len);
      memcpy(h + 2, c->data, (size_t)c->len * 4);
      h += 2 + (size_t)c->len * 4;
    } else {
      memcpy(h, c->data, 8192);
      h += 8192;
    }
  }
  for (int32_t i = 0; i < n; i++) orc_free(cs[i]);
  free(cs);
  free(keys);
  *out_len = total;
  return buf;
}

void orc_wire_free(uint8_t* p) { free(p); }

/* Bitmap.UnmarshalBinary via NewRoaringIterator: returns NULL (and *err = 1) on the error
 * conditions the iterators report, else the bitmap. */
orc_bitmap* orc_roaring_unmarshal(const uint8_t* data, uint64_t len, int32_t* err) {
  *err = 0;
  orc_bitmap* out = orc_bitmap_new();
  if (len < HEADER_BASE_SIZE) goto bad; /* "not long enough to be a roaring header" :2036 */
  const uint32_t magic = rd16(data);
  if (magic == MAGIC_NUMBER) {
    if (data[2] != 0) goto bad; /* storageVersion :1985 */
    const uint64_t keys = rd32(data + 4);
    if (keys == 0) return out;
    if (len < HEADER_BASE_SIZE + keys * 16) goto bad;
    const uint8_t* headers = data + HEADER_BASE_SIZE;
    const uint8_t* offsets = headers + keys * 12;
    uint32_t prev32 = (uint32_t)(HEADER_BASE_SIZE + keys * 16);
    uint64_t chunk = (HEADER_BASE_SIZE + keys * 16) & ~0xFFFFFFFFull;
    for (uint64_t i = 0; i < keys; i++) {
      const uint8_t* h = headers + i * 12;
      const uint64_t key = rd64(h);
      const uint32_t typ = rd16(h + 8);
      const int32_t n = (int32_t)rd16(h + 10) + 1;
      const uint32_t off32 = rd32(offsets + i * 4);
      if (off32 < prev32) chunk += 1ull << 32; /* 4 GiB wrap, :2150-2153 */
      prev32 = off32;
      uint64_t off = chunk + off32;
      uint32_t run_count = 0;
      if (typ == ORC_RUN) {
        if (off + 2 > len) goto bad;
        run_count = rd16(data + off);
        off += 2;
      }
      if (off > len || off < HEADER_BASE_SIZE) goto bad;
      uint64_t size;
      if (typ == ORC_ARRAY) size = (uint64_t)n * 2;
      else if (typ == ORC_BITMAP) size = 8192;
      else if (typ == ORC_RUN) size = (uint64_t)run_count * 4;
      else goto bad;
      if (off + size > len) goto bad;
      orc_container* c;
      if (typ == ORC_ARRAY) {
        uint16_t* v = (uint16_t*)malloc((size_t)n * 2);
        for (int32_t k = 0; k < n; k++) v[k] = rd16(data + off + 2 * (uint64_t)k);
        c = orc_new_array(v, n);
        free(v);
      } else if (typ == ORC_BITMAP) {
        uint64_t* w = (uint64_t*)malloc(8192);
        for (int k = 0; k < 1024; k++) w[k] = rd64(data + off + 8 * (uint64_t)k);
        c = orc_new_bitmap(w, n);
        free(w);
      } else {
        orc_interval16* r = (orc_interval16*)malloc((size_t)(run_count ? run_count : 1) * sizeof(orc_interval16));
        for (uint32_t k = 0; k < run_count; k++) {
          r[k].start = rd16(data + off + 4 * (uint64_t)k);
          r[k].last = rd16(data + off + 4 * (uint64_t)k + 2);
        }
        c = orc_new_run(r, (int32_t)run_count);
        orc_set_n(c, n);
        free(r);
      }
      orc_bitmap_put(out, key, c);
    }
    return out;
  }
  if (magic == SERIAL_COOKIE || magic == SERIAL_COOKIE_NO_RUN) {
    /* readOfficialHeader, roaring.go:6948-7006 */
    const uint32_t cookie = rd32(data);
    uint64_t pos = 4;
    uint32_t size;
    int have_runs = 0;
    const uint8_t* is_run = NULL;
    if (cookie == SERIAL_COOKIE_NO_RUN) {
      size = rd32(data + pos);
      pos += 4;
    } else if ((cookie & 0xFFFF) == SERIAL_COOKIE) {
      have_runs = 1;
      size = (uint32_t)(uint16_t)(cookie >> 16) + 1;
      const uint64_t rb = ((uint64_t)size + 7) / 8;
      if (pos + rb > len) goto bad;
      is_run = data + pos;
      pos += rb;
    } else {
      goto bad;
    }
    const uint64_t header = pos;
    if (size > (1u << 16)) goto bad;
    if (pos + 4ull * size >= len) goto bad; /* "key-cardinality slice overruns buffer" :7000 */
    pos += 4ull * size;
    uint64_t data_off = pos; /* with runs: sequential; without: offsets table follows */
    const uint8_t* offsets = NULL;
    if (!have_runs) {
      if (len < pos + 4ull * size) goto bad;
      offsets = data + pos;
    }
    for (uint32_t i = 0; i < size; i++) {
      const uint64_t key = rd16(data + header + 4ull * i);
      const int32_t n = (int32_t)rd16(data + header + 4ull * i + 2) + 1;
      uint32_t typ = (n < ORC_ARRAY_MAX_SIZE) ? ORC_ARRAY : ORC_BITMAP;
      if (is_run && (is_run[i / 8] & (1u << (i % 8)))) typ = ORC_RUN;
      if (!have_runs) data_off = rd32(offsets + 4ull * i);
      uint32_t run_count = 0;
      if (typ == ORC_RUN) {
        if (data_off + 2 > len) goto bad;
        run_count = rd16(data + data_off);
        data_off += 2;
      }
      if (data_off > len || data_off < HEADER_BASE_SIZE) goto bad;
      const uint64_t sz = typ == ORC_ARRAY ? (uint64_t)n * 2 : typ == ORC_BITMAP ? 8192 : (uint64_t)run_count * 4;
      if (data_off + sz > len) goto bad;
      orc_container* c;
      if (typ == ORC_ARRAY) {
        uint16_t* v = (uint16_t*)malloc((size_t)n * 2);
        for (int32_t k = 0; k < n; k++) v[k] = rd16(data + data_off + 2 * (uint64_t)k);
        c = orc_new_array(v, n);
        free(v);
      } else if (typ == ORC_BITMAP) {
        uint64_t* w = (uint64_t*)malloc(8192);
        for (int k = 0; k < 1024; k++) w[k] = rd64(data + data_off + 8 * (uint64_t)k);
        c = orc_new_bitmap(w, n);
        free(w);
      } else {
        /* official runs are {start, length-1}: Last += Start (:2243-2246) */
        orc_interval16* r = (orc_interval16*)malloc((size_t)(run_count ? run_count : 1) * sizeof(orc_interval16));
        for (uint32_t k = 0; k < run_count; k++) {
          r[k].start = rd16(data + data_off + 4 * (uint64_t)k);
          r[k].last = (uint16_t)(rd16(data + data_off + 4 * (uint64_t)k + 2) + r[k].start);
        }
        c = orc_new_run(r, (int32_t)run_count);
        orc_set_n(c, n);
        free(r);
      }
      data_off += sz;
      orc_bitmap_put(out, key, c);
    }
    return out;
  }
bad:
  orc_bitmap_free(out);
  *err = 1;
  return NULL;
}
