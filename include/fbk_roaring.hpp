// fbk_roaring.hpp — C++ host-side mirror of the reference's operator interface for the hot
// path, layered on the C ABI (fbk.h).  The Go toolchain is absent in this image, so this is
// the compiled-language host side: same names, argument meaning and error behaviour as
//   roaring.Container            roaring/container_stash.go:46-53
//   roaring.Bitmap               roaring/roaring.go:232-248  (slice containers: sorted keys)
//   pilosa.RowSegment / Row      row.go:15-33, 511-523; ops row.go:226-353, 556-610
// Set-ops never return errors in the reference (invariant breaks panic); here a failing
// device call throws fbk::Error (the Go shim would fall back to the pure-Go path).
// All arithmetic runs on the GPU: one batch call per Row operation covering every shard.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "fbk.h"

namespace fbk {

constexpr uint64_t ShardWidth = 1ull << 20;  // shardwidth/helper.go:14

struct Error : std::runtime_error {
  int32_t code;
  Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int32_t rc) {
  if (rc != FBK_OK) throw Error(rc, fbk_last_error(nullptr));
}

struct Interval16 {  // roaring.go:3041
  uint16_t Start, Last;
};

// roaring.Container: exactly one of the three payloads is used.
struct Container {
  uint8_t typ = FBK_TYPE_NIL;
  int32_t n = 0;
  std::vector<uint16_t> array;
  std::vector<uint64_t> bitmap;
  std::vector<Interval16> runs;

  static Container NewContainerArray(std::vector<uint16_t> v) {
    Container c;
    c.typ = FBK_TYPE_ARRAY;
    c.n = int32_t(v.size());
    c.array = std::move(v);
    return c;
  }
  static Container NewContainerBitmapN(std::vector<uint64_t> w, int32_t n) {
    Container c;
    c.typ = FBK_TYPE_BITMAP;
    w.resize(FBK_BITMAP_WORDS, 0);
    c.n = n;
    c.bitmap = std::move(w);
    return c;
  }
  static Container NewContainerRunN(std::vector<Interval16> r, int32_t n) {
    Container c;
    c.typ = FBK_TYPE_RUN;
    c.n = n;
    c.runs = std::move(r);
    return c;
  }
  int32_t N() const { return n; }
  uint32_t len() const {
    return typ == FBK_TYPE_ARRAY ? uint32_t(array.size()) : typ == FBK_TYPE_RUN ? uint32_t(runs.size()) : FBK_BITMAP_WORDS;
  }
  const void* payload() const {
    return typ == FBK_TYPE_ARRAY ? (const void*)array.data() : typ == FBK_TYPE_RUN ? (const void*)runs.data() : (const void*)bitmap.data();
  }
  size_t payload_bytes() const { return typ == FBK_TYPE_ARRAY ? array.size() * 2 : typ == FBK_TYPE_RUN ? runs.size() * 4 : 8192; }
  // every set value, ascending
  void values(uint64_t key, std::vector<uint64_t>& out) const {
    const uint64_t hb = key << 16;
    if (typ == FBK_TYPE_ARRAY) {
      for (uint16_t v : array) out.push_back(hb | v);
    } else if (typ == FBK_TYPE_RUN) {
      for (const Interval16& r : runs)
        for (uint32_t v = r.Start; v <= r.Last; ++v) out.push_back(hb | v);
    } else if (typ == FBK_TYPE_BITMAP) {
      for (uint32_t i = 0; i < FBK_BITMAP_WORDS; ++i)
        for (uint64_t w = bitmap[i]; w; w &= w - 1) out.push_back(hb | (uint64_t(i) * 64 + uint64_t(__builtin_ctzll(w))));
    }
  }
};

// roaring.Bitmap over slice containers: ascending keys (containers_slice.go:5-10).
class Bitmap {
 public:
  std::vector<uint64_t> keys;
  std::vector<Container> containers;

  // NewBitmap(values...): one array container per key (< 4096 values) or a bitmap.
  static Bitmap NewBitmap(std::vector<uint64_t> values) {
    std::sort(values.begin(), values.end());
    values.erase(std::unique(values.begin(), values.end()), values.end());
    Bitmap b;
    size_t i = 0;
    while (i < values.size()) {
      const uint64_t key = values[i] >> 16;
      size_t j = i;
      while (j < values.size() && (values[j] >> 16) == key) ++j;
      if (j - i < 4096) {
        std::vector<uint16_t> a;
        for (size_t k = i; k < j; ++k) a.push_back(uint16_t(values[k] & 0xFFFF));
        b.Put(key, Container::NewContainerArray(std::move(a)));
      } else {
        std::vector<uint64_t> w(FBK_BITMAP_WORDS, 0);
        for (size_t k = i; k < j; ++k) w[(values[k] & 0xFFFF) >> 6] |= 1ull << (values[k] & 63);
        b.Put(key, Container::NewContainerBitmapN(std::move(w), int32_t(j - i)));
      }
      i = j;
    }
    return b;
  }
  void Put(uint64_t key, Container c) {  // keys must arrive ascending
    keys.push_back(key);
    containers.push_back(std::move(c));
  }
  uint64_t Count() const {  // roaring.go:542
    uint64_t n = 0;
    for (const Container& c : containers) n += uint64_t(c.N());
    return n;
  }
  bool Any() const { return Count() != 0; }  // roaring.go:547
  std::vector<uint64_t> Slice() const {      // roaring.go:623
    std::vector<uint64_t> out;
    for (size_t i = 0; i < keys.size(); ++i) containers[i].values(keys[i], out);
    return out;
  }
};

// pilosa.RowSegment (row.go:511-523): the bits of one row in one shard; container keys are
// shard*16 + slot (fragment.row rebases them, fragment.go:318).
struct RowSegment {
  uint64_t shard = 0;
  Bitmap data;
  uint64_t n = 0;
};

class Row {  // row.go:15-33
 public:
  std::vector<RowSegment> Segments;

  static Row NewRow(std::vector<uint64_t> columns) {  // row.go:36-44
    std::map<uint64_t, std::vector<uint64_t>> by_shard;
    for (uint64_t c : columns) by_shard[c / ShardWidth].push_back(c);
    Row r;
    for (auto& kv : by_shard) {
      RowSegment s;
      s.shard = kv.first;
      s.data = Bitmap::NewBitmap(std::move(kv.second));
      s.n = s.data.Count();
      r.Segments.push_back(std::move(s));
    }
    return r;
  }
  std::vector<uint64_t> Columns() const {  // row.go:469
    std::vector<uint64_t> out;
    for (const RowSegment& s : Segments) {
      std::vector<uint64_t> v = s.data.Slice();
      out.insert(out.end(), v.begin(), v.end());
    }
    return out;
  }
  uint64_t Count() const {  // row.go:446
    uint64_t n = 0;
    for (const RowSegment& s : Segments) n += s.n;
    return n;
  }
  bool Any() const { return Count() != 0; }
};

// One GPU.  The Row-level operations below are what executeIntersectShard & friends
// (executor.go:5357, 5382, 2950, 5513) compute shard by shard, issued as ONE batch.
class Device {
 public:
  explicit Device(int device = 0) { check(fbk_open(device, 0, &ctx_)); }
  ~Device() { fbk_close(ctx_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;

  // Row.intersectionCount, row.go:226: non-overlapping shards are ignored.
  uint64_t IntersectionCount(const Row& a, const Row& b) {
    Pairing p = pair_segments(a, b);
    if (p.both.empty()) return 0;
    fbk_batch *ba = upload(p, a, true), *bb = upload(p, b, false);
    std::vector<uint32_t> rows(p.both.size());
    for (size_t i = 0; i < rows.size(); ++i) rows[i] = uint32_t(i);
    std::vector<uint64_t> counts(rows.size());
    int32_t rc = fbk_intersection_count(ctx_, ba, rows.data(), bb, rows.data(), rows.size(), counts.data());
    fbk_batch_free(ctx_, ba);
    fbk_batch_free(ctx_, bb);
    check(rc);
    uint64_t n = 0;
    for (uint64_t c : counts) n += c;
    return n;
  }
  Row Intersect(const Row& a, const Row& b) { return setop(FBK_OP_AND, a, b); }    // row.go:242
  Row Union(const Row& a, const Row& b) { return setop(FBK_OP_OR, a, b); }         // row.go:288
  Row Difference(const Row& a, const Row& b) { return setop(FBK_OP_ANDNOT, a, b); }  // row.go:333
  Row Xor(const Row& a, const Row& b) { return setop(FBK_OP_XOR, a, b); }          // row.go:268
  // Row.Shift, row.go:374-396: n single-column shifts of every segment.  The reference leaves the
  // bit shifted out of a shard's last column in that segment, under a container key of the NEXT
  // shard ("TODO: deal with overflow", row.go:615) — Columns() reports it as the first column of
  // the next shard; here it is carried into the next shard's segment, the same columns.
  Row Shift(const Row& a, int64_t n) {
    if (n < 0) throw Error(FBK_E_INVALID, "cannot shift by negative values");
    Row work = a;
    for (int64_t i = 0; i < n; ++i) work = shift1(work);
    return work;
  }

 private:
  struct Pairing {
    std::vector<std::pair<const RowSegment*, const RowSegment*>> both;  // same shard on both sides
    std::vector<const RowSegment*> only_a, only_b;
  };
  // mergeSegmentIterator, row.go:692-735
  static Pairing pair_segments(const Row& a, const Row& b) {
    Pairing p;
    size_t i = 0, j = 0;
    while (i < a.Segments.size() || j < b.Segments.size()) {
      if (j >= b.Segments.size() || (i < a.Segments.size() && a.Segments[i].shard < b.Segments[j].shard)) {
        p.only_a.push_back(&a.Segments[i++]);
      } else if (i >= a.Segments.size() || a.Segments[i].shard > b.Segments[j].shard) {
        p.only_b.push_back(&b.Segments[j++]);
      } else {
        p.both.emplace_back(&a.Segments[i++], &b.Segments[j++]);
      }
    }
    return p;
  }
  fbk_batch* upload(const Pairing& p, const Row&, bool first) {
    std::vector<fbk_container_desc> descs;
    std::vector<uint8_t> payload;
    for (size_t r = 0; r < p.both.size(); ++r) {
      const RowSegment* s = first ? p.both[r].first : p.both[r].second;
      for (size_t k = 0; k < s->data.keys.size(); ++k) {
        const Container& c = s->data.containers[k];
        if (c.typ == FBK_TYPE_NIL) continue;
        fbk_container_desc d;
        std::memset(&d, 0, sizeof(d));
        d.key = s->data.keys[k];
        d.off = payload.size();
        d.row = uint32_t(r);
        d.len = c.len();
        d.n = c.N();
        d.type = c.typ;
        const uint8_t* src = static_cast<const uint8_t*>(c.payload());
        payload.insert(payload.end(), src, src + c.payload_bytes());
        descs.push_back(d);
      }
    }
    fbk_batch* b = nullptr;
    if (payload.empty()) payload.push_back(0);
    check(fbk_batch_upload(ctx_, descs.data(), descs.size(), uint32_t(p.both.size()), payload.data(), payload.size(), &b));
    return b;
  }
  fbk_batch* upload_segments(const std::vector<const RowSegment*>& segs) {
    Pairing p;
    for (const RowSegment* sg : segs) p.both.emplace_back(sg, sg);
    return upload(p, Row(), true);
  }
  // containers of an output batch -> segments (row r of the batch = shard shards[r])
  void download_segments(fbk_batch* bo, const std::vector<uint64_t>& shards, const std::vector<uint64_t>& counts,
                         std::map<uint64_t, RowSegment>& out) {
    uint32_t n_rows = 0;
    uint64_t nc = 0, pb = 0;
    check(fbk_batch_info(ctx_, bo, &n_rows, &nc, &pb));
    std::vector<fbk_container_desc> descs(nc ? nc : 1);
    std::vector<uint8_t> payload(pb ? pb : 1);
    check(fbk_batch_download(ctx_, bo, descs.data(), nc, payload.data(), pb));
    for (size_t r = 0; r < shards.size(); ++r) {
      RowSegment sg;
      sg.shard = shards[r];
      sg.n = counts[r];
      out[sg.shard] = std::move(sg);
    }
    for (uint64_t i = 0; i < nc; ++i) {
      const fbk_container_desc& d = descs[i];
      RowSegment& sg = out[shards[d.row]];
      const uint8_t* src = payload.data() + d.off;
      if (d.type == FBK_TYPE_ARRAY) {
        std::vector<uint16_t> v(d.len);
        std::memcpy(v.data(), src, size_t(d.len) * 2);
        sg.data.Put(d.key, Container::NewContainerArray(std::move(v)));
      } else if (d.type == FBK_TYPE_RUN) {
        std::vector<Interval16> v(d.len);
        std::memcpy(v.data(), src, size_t(d.len) * 4);
        sg.data.Put(d.key, Container::NewContainerRunN(std::move(v), d.n));
      } else {
        std::vector<uint64_t> v(FBK_BITMAP_WORDS);
        std::memcpy(v.data(), src, 8192);
        sg.data.Put(d.key, Container::NewContainerBitmapN(std::move(v), d.n));
      }
    }
  }
  Row shift1(const Row& a) {
    if (a.Segments.empty()) return a;
    std::vector<const RowSegment*> segs;
    std::map<uint64_t, uint32_t> idx;  // shard -> row of the uploaded batch
    std::set<uint64_t> out_shards;
    for (const RowSegment& sg : a.Segments) {
      idx[sg.shard] = uint32_t(segs.size());
      segs.push_back(&sg);
      out_shards.insert(sg.shard);
      out_shards.insert(sg.shard + 1);  // may receive the carried bit
    }
    std::vector<uint64_t> shards(out_shards.begin(), out_shards.end());
    std::vector<uint32_t> rows(shards.size(), FBK_NO_ROW), carry(shards.size(), FBK_NO_ROW);
    for (size_t i = 0; i < shards.size(); ++i) {
      auto it = idx.find(shards[i]);
      if (it != idx.end()) rows[i] = it->second;
      if (shards[i] > 0 && (it = idx.find(shards[i] - 1)) != idx.end()) carry[i] = it->second;
    }
    fbk_batch *ba = upload_segments(segs), *bo = nullptr;
    std::vector<uint64_t> counts(shards.size());
    int32_t rc = fbk_shift(ctx_, ba, rows.data(), carry.data(), shards.size(), FBK_SETOP_OPTIMIZE, &bo, counts.data());
    fbk_batch_free(ctx_, ba);
    check(rc);
    std::map<uint64_t, RowSegment> out;
    try {
      download_segments(bo, shards, counts, out);
    } catch (...) {
      fbk_batch_free(ctx_, bo);
      throw;
    }
    fbk_batch_free(ctx_, bo);
    Row r;
    for (auto& kv : out)
      if (kv.second.n) r.Segments.push_back(std::move(kv.second));
    return r;
  }
  Row setop(int32_t op, const Row& a, const Row& b) {
    Pairing p = pair_segments(a, b);
    std::map<uint64_t, RowSegment> out;
    if (!p.both.empty()) {
      fbk_batch *ba = upload(p, a, true), *bb = upload(p, b, false), *bo = nullptr;
      std::vector<uint32_t> rows(p.both.size());
      for (size_t i = 0; i < rows.size(); ++i) rows[i] = uint32_t(i);
      std::vector<uint64_t> counts(rows.size());
      int32_t rc = fbk_setop(ctx_, op, ba, rows.data(), bb, rows.data(), rows.size(), FBK_SETOP_OPTIMIZE, &bo, counts.data());
      fbk_batch_free(ctx_, ba);
      fbk_batch_free(ctx_, bb);
      check(rc);
      uint32_t n_rows = 0;
      uint64_t nc = 0, pb = 0;
      check(fbk_batch_info(ctx_, bo, &n_rows, &nc, &pb));
      std::vector<fbk_container_desc> descs(nc ? nc : 1);
      std::vector<uint8_t> payload(pb ? pb : 1);
      rc = fbk_batch_download(ctx_, bo, descs.data(), nc, payload.data(), pb);
      fbk_batch_free(ctx_, bo);
      check(rc);
      for (size_t r = 0; r < p.both.size(); ++r) {
        RowSegment s;
        s.shard = p.both[r].first->shard;
        s.n = counts[r];
        out[s.shard] = std::move(s);
      }
      for (uint64_t i = 0; i < nc; ++i) {
        const fbk_container_desc& d = descs[i];
        RowSegment& s = out[p.both[d.row].first->shard];
        const uint8_t* src = payload.data() + d.off;
        if (d.type == FBK_TYPE_ARRAY) {
          std::vector<uint16_t> v(d.len);
          std::memcpy(v.data(), src, size_t(d.len) * 2);
          s.data.Put(d.key, Container::NewContainerArray(std::move(v)));
        } else if (d.type == FBK_TYPE_RUN) {
          std::vector<Interval16> v(d.len);
          std::memcpy(v.data(), src, size_t(d.len) * 4);
          s.data.Put(d.key, Container::NewContainerRunN(std::move(v), d.n));
        } else {
          std::vector<uint64_t> v(FBK_BITMAP_WORDS);
          std::memcpy(v.data(), src, 8192);
          s.data.Put(d.key, Container::NewContainerBitmapN(std::move(v), d.n));
        }
      }
    }
    // segments present on one side only: dropped by Intersect (row.go:247-250), kept from
    // both sides by Union/Xor (row.go:272-279), kept from `a` by Difference (row.go:345-350)
    if (op != FBK_OP_AND)
      for (const RowSegment* s : p.only_a) out[s->shard] = *s;
    if (op == FBK_OP_OR || op == FBK_OP_XOR)
      for (const RowSegment* s : p.only_b) out[s->shard] = *s;
    Row r;
    for (auto& kv : out) r.Segments.push_back(std::move(kv.second));
    return r;
  }

  fbk_ctx* ctx_ = nullptr;
};

}  // namespace fbk
