// fbk_executor.hpp — C++ host-side mirror of the reference's per-query operators for the hot
// path, layered on the C ABI (fbk.h) and on fbk_roaring.hpp.  The Go toolchain is absent in this
// image, so this is the compiled-language host side the Go executor would be: same operator
// names, argument meaning and result ordering as
//   executeCount / executeBitmapCall(Shard)     executor.go:5839-5892, 1782-1900
//   executeIntersect/Union/Difference/XorShard  executor.go:5357, 5382, 2950, 5513 (left folds)
//   executeRowBSIGroupShard (Row(v > k) ...)     executor.go:5249-5355, field.go:2412-2482
//   executeSum / Min / Max (+ ValCount)          executor.go:2155-2275, 8438-8548
//   executePercentile                            executor.go:1310-1595
//   executeTopK (doTopK + PivotDescending)       executor.go:2357-2412, 2705-2746, bsi.go:18-62
//   executeTopN (counts; the rank cache is not mirrored)   executor.go:2776-2868
//   executeGroupBy (groupByIterator odometer)    executor.go:3918-3990, 8617-8934
// Every shard of a query is evaluated in ONE device call per operator (the reference maps a
// closure over shards, executor.go:6449); the cross-shard reduce is the same associative
// arithmetic (sum of counts, ValCount.Add/Smaller/Larger, concatenation of row segments).
// A tiny in-memory Index (set fields and int fields) stands in for Holder/Field/fragment: it
// only exists to make the operators testable with the reference's own test vectors.
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "fbk_roaring.hpp"

namespace fbk {

struct Pair {  // cache.go Pair
  uint64_t ID = 0, Count = 0;
  bool operator==(const Pair& o) const { return ID == o.ID && Count == o.Count; }
};
struct FieldRow {  // executor.go FieldRow
  std::string Field;
  uint64_t RowID = 0;
  bool operator==(const FieldRow& o) const { return Field == o.Field && RowID == o.RowID; }
};
struct GroupCount {  // executor.go GroupCount
  std::vector<FieldRow> Group;
  uint64_t Count = 0;
  int64_t Agg = 0;
  bool operator==(const GroupCount& o) const { return Group == o.Group && Count == o.Count && Agg == o.Agg; }
};
struct ValCount {  // executor.go:8380; integer fields only
  int64_t Val = 0, Count = 0;
  ValCount Add(const ValCount& o) const { return {Val + o.Val, Count + o.Count}; }  // :8438
  ValCount Smaller(const ValCount& o) const {                                       // :8446-8468
    if (Count == 0 || (o.Val < Val && o.Count > 0)) return o;
    return {Val, Count + (Val == o.Val ? o.Count : 0)};
  }
  ValCount Larger(const ValCount& o) const {  // :8526-8548
    if (Count == 0 || (o.Val > Val && o.Count > 0)) return o;
    return {Val, Count + (Val == o.Val ? o.Count : 0)};
  }
  bool operator==(const ValCount& o) const { return Val == o.Val && Count == o.Count; }
};

// A PQL bitmap call (the subset on the hot path).
struct Call {
  enum Kind { kRow, kRange, kBetween, kIntersect, kUnion, kDifference, kXor, kNot, kAll, kShift } kind = kRow;
  std::string field;
  uint64_t row = 0;    // Row(field=row)
  int32_t op = 0;      // FBK_BSI_* for Row(field <op> value)
  int64_t value = 0, value2 = 0;
  std::vector<Call> children;
  static Call Row(std::string f, uint64_t r) {
    Call c;
    c.kind = kRow;
    c.field = std::move(f);
    c.row = r;
    return c;
  }
  static Call Range(std::string f, int32_t op, int64_t v) {
    Call c;
    c.kind = kRange;
    c.field = std::move(f);
    c.op = op;
    c.value = v;
    return c;
  }
  static Call Between(std::string f, int64_t lo, int64_t hi) {
    Call c;
    c.kind = kBetween;
    c.field = std::move(f);
    c.value = lo;
    c.value2 = hi;
    return c;
  }
  static Call Not(Call child) {  // Not(x) = existence row \ x (executeNotShard, executor.go:5554-5604)
    Call c;
    c.kind = kNot;
    c.children.push_back(std::move(child));
    return c;
  }
  static Call All() {  // All(): the existence row (executeAllCallShard, executor.go:5606)
    Call c;
    c.kind = kAll;
    return c;
  }
  static Call Shift(Call child, int64_t n) {  // Shift(x, n=n): executeShiftShard, executor.go:5818-5836
    Call c;
    c.kind = kShift;
    c.value = n;
    c.children.push_back(std::move(child));
    return c;
  }
  bool ContainsShift() const {
    if (kind == kShift) return true;
    for (const Call& ch : children)
      if (ch.ContainsShift()) return true;
    return false;
  }
  static Call Nary(Kind k, std::vector<Call> ch) {
    Call c;
    c.kind = k;
    c.children = std::move(ch);
    return c;
  }
};

class Index;

// One evaluated bitmap call: for every shard of the index, a row of a device batch.
class RowSet {
 public:
  RowSet() = default;
  RowSet(fbk_ctx* ctx, const fbk_batch* b, std::vector<uint32_t> rows, bool owned) : ctx_(ctx), batch_(b), rows_(std::move(rows)), owned_(owned) {}
  RowSet(RowSet&& o) noexcept { *this = std::move(o); }
  RowSet& operator=(RowSet&& o) noexcept {
    reset();
    ctx_ = o.ctx_;
    batch_ = o.batch_;
    rows_ = std::move(o.rows_);
    owned_ = o.owned_;
    o.batch_ = nullptr;
    o.owned_ = false;
    return *this;
  }
  RowSet(const RowSet&) = delete;
  RowSet& operator=(const RowSet&) = delete;
  ~RowSet() { reset(); }
  const fbk_batch* batch() const { return batch_; }
  const std::vector<uint32_t>& rows() const { return rows_; }

 private:
  void reset() {
    if (owned_ && batch_) fbk_batch_free(ctx_, const_cast<fbk_batch*>(batch_));
    batch_ = nullptr;
    owned_ = false;
  }
  fbk_ctx* ctx_ = nullptr;
  const fbk_batch* batch_ = nullptr;
  std::vector<uint32_t> rows_;
  bool owned_ = false;
};

// In-memory stand-in for Holder / Index / Field / fragment storage.
class Index {
 public:
  explicit Index(int device = 0) { check(fbk_open(device, 0, &ctx_)); }
  ~Index() {
    for (auto& kv : sets_) fbk_batch_free(ctx_, kv.second.batch);
    for (auto& kv : ints_) fbk_batch_free(ctx_, kv.second.batch);
    fbk_close(ctx_);
  }
  Index(const Index&) = delete;
  Index& operator=(const Index&) = delete;

  void CreateSetField(const std::string& name) { sets_[name]; }
  void CreateIntField(const std::string& name, int64_t min, int64_t max) {  // OptFieldTypeInt
    IntField& f = ints_[name];
    f.min = min;
    f.max = max;
    f.base = min > 0 ? min : (max < 0 ? max : 0);  // bsiBase, field.go:2384-2391
  }
  void SetBit(const std::string& field, uint64_t row, uint64_t col) {  // Set(col, field=row)
    sets_.at(field).bits[row].insert(col);
    sets_[kExistence].bits[0].insert(col);  // IndexOptions.TrackExistence: the "_exists" field, row 0
    dirty_ = true;
  }
  void SetValue(const std::string& field, uint64_t col, int64_t value) {  // Set(col, field=value), field.go:1497-1543
    IntField& f = ints_.at(field);
    if (value < f.min || value > f.max) throw Error(FBK_E_INVALID, "value out of range");  // ErrBSIGroupValueTooLow/High
    f.values[col] = value;
    const int64_t bv = value - f.base;
    const uint64_t mag = bv < 0 ? uint64_t(-bv) : uint64_t(bv);
    const uint32_t need = mag ? 64u - uint32_t(__builtin_clzll(mag)) : 0u;  // bitDepthInt64, field.go:2507
    f.bit_depth = std::max(f.bit_depth, need);
    sets_[kExistence].bits[0].insert(col);
    dirty_ = true;
  }
  fbk_ctx* ctx() { return ctx_; }
  static constexpr const char* kExistence = "_exists";  // existenceFieldName, index.go

 private:
  friend class Executor;
  struct SetField {
    std::map<uint64_t, std::set<uint64_t>> bits;  // row id -> columns
    fbk_batch* batch = nullptr;
    std::map<std::pair<uint64_t, uint64_t>, uint32_t> ordinal;  // (shard, row id) -> batch row
    std::vector<uint64_t> row_ids;                               // ascending (Rows(field))
    uint32_t empty_row = 0;                                      // a row with no containers
  };
  struct IntField {
    int64_t min = 0, max = 0, base = 0;
    uint32_t bit_depth = 0;
    std::map<uint64_t, int64_t> values;  // column -> value
    fbk_batch* batch = nullptr;
    std::map<uint64_t, uint32_t> base_row;  // shard -> batch row of the exists plane
    uint32_t empty_base = 0;                // an all-empty BSI fragment for shards without values
  };

  static void append_row(const std::vector<uint64_t>& cols_in_shard, uint64_t key_base, uint32_t row,
                         std::vector<fbk_container_desc>& descs, std::vector<uint8_t>& payload) {
    Bitmap bm = Bitmap::NewBitmap(cols_in_shard);
    for (size_t k = 0; k < bm.keys.size(); ++k) {
      const Container& c = bm.containers[k];
      fbk_container_desc d;
      std::memset(&d, 0, sizeof(d));
      d.key = key_base + bm.keys[k];
      d.off = payload.size();
      d.row = row;
      d.len = c.len();
      d.n = c.N();
      d.type = c.typ;
      const uint8_t* src = static_cast<const uint8_t*>(c.payload());
      payload.insert(payload.end(), src, src + c.payload_bytes());
      descs.push_back(d);
    }
  }

  // (re)build the device-resident fragments: what fragment storage + fbk_batch_upload_rbf give
  // the real system
  void Sync() {
    if (!dirty_) return;
    std::set<uint64_t> shards;
    for (auto& kv : sets_)
      for (auto& rb : kv.second.bits)
        for (uint64_t c : rb.second) shards.insert(c / ShardWidth);
    for (auto& kv : ints_)
      for (auto& cv : kv.second.values) shards.insert(cv.first / ShardWidth);
    shards_.assign(shards.begin(), shards.end());
    for (auto& kv : sets_) {
      SetField& f = kv.second;
      if (f.batch) fbk_batch_free(ctx_, f.batch);
      f.batch = nullptr;
      f.ordinal.clear();
      f.row_ids.clear();
      std::vector<fbk_container_desc> descs;
      std::vector<uint8_t> payload;
      uint32_t n_rows = 0;
      for (auto& rb : f.bits) {
        if (rb.second.empty()) continue;
        f.row_ids.push_back(rb.first);
        std::map<uint64_t, std::vector<uint64_t>> by_shard;
        for (uint64_t c : rb.second) by_shard[c / ShardWidth].push_back(c % ShardWidth);
        for (auto& sc : by_shard) {
          f.ordinal[{sc.first, rb.first}] = n_rows;
          append_row(sc.second, sc.first * 16, n_rows, descs, payload);  // keys rebased to shard*16+slot, fragment.go:318
          ++n_rows;
        }
      }
      f.empty_row = n_rows++;
      if (payload.empty()) payload.push_back(0);
      check(fbk_batch_upload(ctx_, descs.data(), descs.size(), n_rows, payload.data(), payload.size(), &f.batch));
    }
    for (auto& kv : ints_) {
      IntField& f = kv.second;
      if (f.batch) fbk_batch_free(ctx_, f.batch);
      f.batch = nullptr;
      f.base_row.clear();
      std::vector<fbk_container_desc> descs;
      std::vector<uint8_t> payload;
      uint32_t n_rows = 0;
      std::map<uint64_t, std::vector<std::pair<uint64_t, int64_t>>> by_shard;
      for (auto& cv : f.values) by_shard[cv.first / ShardWidth].push_back({cv.first % ShardWidth, cv.second - f.base});
      for (auto& sv : by_shard) {
        f.base_row[sv.first] = n_rows;
        // fragment.positionsForValue, fragment.go:619-657: exists, sign, magnitude planes
        std::vector<std::vector<uint64_t>> planes(f.bit_depth + 2);
        for (auto& cv : sv.second) {
          planes[0].push_back(cv.first);
          if (cv.second < 0) planes[1].push_back(cv.first);
          const uint64_t mag = cv.second < 0 ? uint64_t(-cv.second) : uint64_t(cv.second);
          for (uint32_t i = 0; i < f.bit_depth; ++i)
            if ((mag >> i) & 1) planes[2 + i].push_back(cv.first);
        }
        for (auto& p : planes) append_row(p, 0, n_rows++, descs, payload);
      }
      f.empty_base = n_rows;
      n_rows += f.bit_depth + 2;
      if (payload.empty()) payload.push_back(0);
      check(fbk_batch_upload(ctx_, descs.data(), descs.size(), n_rows, payload.data(), payload.size(), &f.batch));
    }
    dirty_ = false;
  }

  fbk_ctx* ctx_ = nullptr;
  std::map<std::string, SetField> sets_;
  std::map<std::string, IntField> ints_;
  std::vector<uint64_t> shards_;
  bool dirty_ = true;
};

class Executor {
 public:
  explicit Executor(Index& idx) : idx_(idx) {}

 private:
  // The shards a query is evaluated over: the index's shards — plus, when the query contains a
  // Shift, the successor of every shard.  The reference lets the bit shifted out of a shard's last
  // column fall into a container keyed one past the segment, which Row.Columns() reports as the
  // first column of the next shard whether or not that shard holds data (row.go:366-373); here
  // that bit needs a row to land in.  (A shift moves a bit by one column, so the successor shards
  // are enough for any n < ShardWidth.)
  const std::vector<uint64_t>& shards() const { return cur_ ? *cur_ : idx_.shards_; }
  struct Scope {
    Executor& e;
    bool mine = false;
    Scope(Executor& ex, std::initializer_list<const Call*> calls) : e(ex) {
      e.idx_.Sync();
      if (e.cur_) return;
      bool shift = false;
      for (const Call* c : calls) shift = shift || (c && c->ContainsShift());
      if (!shift) return;
      std::set<uint64_t> u(e.idx_.shards_.begin(), e.idx_.shards_.end());
      for (uint64_t sh : e.idx_.shards_) u.insert(sh + 1);
      e.ext_.assign(u.begin(), u.end());
      e.cur_ = &e.ext_;
      mine = true;
    }
    ~Scope() {
      if (mine) e.cur_ = nullptr;
    }
  };
  std::vector<uint64_t> ext_;
  const std::vector<uint64_t>* cur_ = nullptr;

 public:

  // ---- bitmap calls --------------------------------------------------------------------------
  // (a RowSet of a query containing Shift has one row per shard AND successor shard: use Count /
  // Columns rather than the raw rows)
  RowSet Bitmap(const Call& c) {
    Scope sc(*this, {&c});
    return eval(c);
  }
  uint64_t Count(const Call& c) {  // executeCount: sum over shards of Row.Count()
    Scope sc(*this, {&c});
    RowSet r = eval(c);
    return count_rows(r);
  }
  std::vector<uint64_t> Columns(const Call& c) {  // Row.Columns() of the result, ascending
    Scope sc(*this, {&c});
    RowSet r = eval(c);
    return columns(r);
  }

  // ---- Sum / Min / Max -----------------------------------------------------------------------
  ValCount Sum(const std::string& field, const Call* filter = nullptr) {
    Scope sc(*this, {filter});
    const Index::IntField& f = idx_.ints_.at(field);
    const size_t n = shards().size();
    if (n == 0) return {};
    std::vector<uint32_t> base = base_rows(f);
    std::vector<int64_t> sums(n);
    std::vector<uint64_t> counts(n);
    if (filter) {
      RowSet fr = eval(*filter);
      check(fbk_bsi_sum(idx_.ctx_, f.batch, base.data(), uint32_t(n), f.bit_depth, fr.batch(), fr.rows().data(), sums.data(), counts.data()));
    } else {
      check(fbk_bsi_sum(idx_.ctx_, f.batch, base.data(), uint32_t(n), f.bit_depth, nullptr, nullptr, sums.data(), counts.data()));
    }
    ValCount out;
    for (size_t s = 0; s < n; ++s)  // executeSumCountShard: Val = vsum + vcount*Base (executor.go:2205), reduce = Add
      out = out.Add({sums[s] + int64_t(counts[s]) * f.base, int64_t(counts[s])});
    return out;
  }
  ValCount Min(const std::string& field, const Call* filter = nullptr) { return minmax(field, filter, true); }
  ValCount Max(const std::string& field, const Call* filter = nullptr) { return minmax(field, filter, false); }

  // ---- Percentile ----------------------------------------------------------------------------
  // executePercentile, executor.go:1310-1595 (integer fields): binary search between Min and Max
  // on Count(Row(field < x) ∩ filter) / Count(Row(field > x) ∩ filter).  Returns false for the
  // "median of nothing is NULL" case (:1399-1402).
  bool Percentile(const std::string& field, double nth, const Call* filter, ValCount* out) {
    if (nth < 0 || nth > 100.0) throw Error(FBK_E_INVALID, "Percentile(): invalid nth value, should be a number between 0 and 100 inclusive");
    Scope sc(*this, {filter});
    const Index::IntField& f = idx_.ints_.at(field);
    auto count_of = [&](RowSet&& r) {
      if (!filter) return count_rows(r);
      RowSet fr = eval(*filter);
      RowSet both = setop(FBK_OP_AND, r, fr);
      return count_rows(both);
    };
    const uint64_t total = count_of(not_null(f));  // Count(Intersect(filter, Row(field != null)))
    if (total == 0) return false;
    const uint64_t desired_less = uint64_t((double(total) * nth) / 100.0);
    const uint64_t desired_greater = uint64_t((double(total) * (100 - nth)) / 100.0);
    ValCount mn;
    if (desired_greater != 0) {
      mn = Min(field, filter);
      if (desired_less == 0) {
        *out = mn;
        return true;
      }
    }
    const ValCount mx = Max(field, filter);
    if (desired_greater == 0) {
      *out = mx;
      return true;
    }
    int64_t lo = mn.Val, hi = mx.Val, guess = mn.Val;
    while (lo < hi) {
      guess = (lo / 2) + (hi / 2) + (((lo % 2) + (hi % 2)) / 2);  // average without overflow (:1497-1501)
      const uint64_t left = count_of(range(Call::Range(field, FBK_BSI_LT, guess)));
      if (left > desired_less) {
        hi = guess - 1;
        continue;
      }
      const uint64_t right = count_of(range(Call::Range(field, FBK_BSI_GT, guess)));
      if (right > desired_greater) {
        lo = guess + 1;
        continue;
      }
      break;
    }
    *out = ValCount{guess, 1};
    return true;
  }

  // ---- Distinct -----------------------------------------------------------------------------
  // Distinct(field=int field, filter): the distinct values (Base added) in ascending order —
  // the SignedRow{Neg, Pos} of executeDistinct (executor.go:1170-1230, 2034-2153) flattened.
  std::vector<int64_t> Distinct(const std::string& field, const Call* filter = nullptr) {
    Scope sc(*this, {filter});
    const Index::IntField& f = idx_.ints_.at(field);
    std::vector<int64_t> out;
    const size_t n = shards().size();
    if (n == 0) return out;
    std::vector<uint32_t> base = base_rows(f);
    uint64_t cnt = 0, cap = 1024;
    for (;;) {
      out.assign(cap, 0);
      int32_t rc;
      if (filter) {
        RowSet fr = eval(*filter);
        rc = fbk_bsi_distinct(idx_.ctx_, f.batch, base.data(), uint32_t(n), f.bit_depth, fr.batch(), fr.rows().data(), out.data(), cap, &cnt);
      } else {
        rc = fbk_bsi_distinct(idx_.ctx_, f.batch, base.data(), uint32_t(n), f.bit_depth, nullptr, nullptr, out.data(), cap, &cnt);
      }
      if (rc == FBK_E_CAPACITY) {
        cap = cnt;
        continue;
      }
      check(rc);
      break;
    }
    out.resize(cnt);
    for (int64_t& v : out) v += f.base;  // value += int64(offset), executor.go:2126
    return out;
  }

  // ---- TopK / TopN ---------------------------------------------------------------------------
  // TopK(field, k, filter): per-row |row ∩ filter| over all shards (doTopK), then the rows in
  // descending count order, ascending id inside one count, zero counts dropped
  // (BSIData.PivotDescending, bsi.go:18-62).  k == 0: no limit.
  // Counting, the reduce over shards and the ordering all happen on the device (fbk_topk): only
  // the k winners come back.
  std::vector<Pair> TopK(const std::string& field, uint64_t k, const Call* filter = nullptr) {
    Scope sc(*this, {filter});
    const Index::SetField& f = idx_.sets_.at(field);
    const size_t n = shards().size(), nr = f.row_ids.size();
    std::vector<Pair> out;
    if (n == 0 || nr == 0) return out;
    const std::vector<uint32_t> rows_a = field_rows(f);
    const uint32_t cap = uint32_t(k && k < nr ? k : nr);
    std::vector<uint32_t> idx(cap);
    std::vector<uint64_t> cnt(cap);
    uint32_t got = 0;
    if (filter) {
      RowSet fr = eval(*filter);
      check(fbk_topk(idx_.ctx_, f.batch, rows_a.data(), uint32_t(nr), fr.batch(), fr.rows().data(), uint32_t(n), uint32_t(cap == nr ? 0 : cap),
                     idx.data(), cnt.data(), cap, &got));
    } else {
      check(fbk_topk(idx_.ctx_, f.batch, rows_a.data(), uint32_t(nr), nullptr, nullptr, uint32_t(n), uint32_t(cap == nr ? 0 : cap), idx.data(),
                     cnt.data(), cap, &got));
    }
    for (uint32_t i = 0; i < got; ++i) out.push_back({f.row_ids[idx[i]], cnt[i]});  // row_ids ascending: index order = id order
    return out;
  }
  // TopN(field, n, src): the same counts ordered by Pairs.Less (count descending; the reference's
  // tie order is unspecified, cache.go:436, here ascending id); the rank-cache thresholds of
  // fragment.top (fragment.go:1340-1426) are approximations the GPU path does not need.
  std::vector<Pair> TopN(const std::string& field, uint64_t n, const Call* src = nullptr) { return TopK(field, n, src); }

  // ---- GroupBy -------------------------------------------------------------------------------
  // GroupBy(Rows(f0), Rows(f1), ..., filter, aggregate=Sum(field=agg)): odometer order over the
  // ascending row ids of each field (last field fastest), groups with Count == 0 skipped
  // (groupByIterator.Next, executor.go:8880-8934); limit == 0: no limit.
  std::vector<GroupCount> GroupBy(const std::vector<std::string>& fields, const Call* filter = nullptr,
                                  const std::string& agg_field = "", uint64_t limit = 0) {
    if (fields.empty()) throw Error(FBK_E_INVALID, "need at least one child call");  // executor.go:3927
    Scope sc(*this, {filter});
    std::vector<GroupCount> out;
    const size_t n = shards().size();
    if (n == 0) return out;
    std::unique_ptr<RowSet> prefix;  // filter ∩ rows of the fields before the last two
    if (filter) prefix.reset(new RowSet(eval(*filter)));
    std::vector<FieldRow> group;
    group_by_rec(fields, 0, prefix.get(), agg_field, limit, group, out);
    return out;
  }

 private:
  // ---- evaluation of bitmap calls ----
  RowSet leaf_row(const std::string& field, uint64_t row) {
    const Index::SetField& f = idx_.sets_.at(field);
    std::vector<uint32_t> rows;
    for (uint64_t s : shards()) {
      auto it = f.ordinal.find({s, row});
      rows.push_back(it == f.ordinal.end() ? f.empty_row : it->second);
    }
    return RowSet(idx_.ctx_, f.batch, std::move(rows), false);
  }
  RowSet fresh(fbk_batch* b) {
    std::vector<uint32_t> rows(shards().size());
    for (size_t i = 0; i < rows.size(); ++i) rows[i] = uint32_t(i);
    return RowSet(idx_.ctx_, b, std::move(rows), true);
  }
  RowSet setop(int32_t op, const RowSet& a, const RowSet& b) {
    fbk_batch* o = nullptr;
    check(fbk_setop(idx_.ctx_, op, a.batch(), a.rows().data(), b.batch(), b.rows().data(), a.rows().size(), FBK_SETOP_OPTIMIZE, &o, nullptr));
    return fresh(o);
  }
  std::vector<uint32_t> base_rows(const Index::IntField& f) const {
    std::vector<uint32_t> base;
    for (uint64_t s : shards()) {
      auto it = f.base_row.find(s);
      base.push_back(it == f.base_row.end() ? f.empty_base : it->second);
    }
    return base;
  }
  RowSet not_null(const Index::IntField& f) {  // fragment.notNull: the exists plane
    return RowSet(idx_.ctx_, f.batch, base_rows(f), false);
  }
  RowSet empty_set(const Index::IntField& f) {
    return RowSet(idx_.ctx_, f.batch, std::vector<uint32_t>(shards().size(), f.empty_base), false);
  }
  // executeRowBSIGroupShard, executor.go:5249-5355, with bsiGroup.baseValue / baseValueBetween
  RowSet range(const Call& c) {
    const Index::IntField& f = idx_.ints_.at(c.field);
    const int64_t dmin = f.base - (f.bit_depth >= 63 ? INT64_MAX : ((int64_t(1) << f.bit_depth) - 1));  // bitDepthMin
    const int64_t dmax = f.base + (f.bit_depth >= 63 ? INT64_MAX : ((int64_t(1) << f.bit_depth) - 1));  // bitDepthMax
    std::vector<uint32_t> base = base_rows(f);
    fbk_batch* o = nullptr;
    if (c.kind == Call::kBetween) {
      int64_t lo = c.value, hi = c.value2;
      if (hi < dmin || lo > dmax || hi < lo) return empty_set(f);  // baseValueBetween outOfRange
      if (lo <= f.min && hi >= f.max) return not_null(f);
      lo = std::max(lo, dmin);
      hi = std::min(hi, dmax);
      check(fbk_bsi_range_between(idx_.ctx_, f.batch, base.data(), uint32_t(base.size()), f.bit_depth, lo - f.base, hi - f.base,
                                  FBK_SETOP_OPTIMIZE, &o, nullptr));
      return fresh(o);
    }
    const int32_t op = c.op;
    const int64_t value = c.value;
    int64_t bv = 0;
    bool out_of_range = false;
    if (op == FBK_BSI_GT || op == FBK_BSI_GTE) {
      if (value > dmax) out_of_range = true;
      else if (value < dmin) bv = dmin - f.base - (op == FBK_BSI_GT ? 1 : 0);
      else bv = value - f.base;
    } else if (op == FBK_BSI_LT || op == FBK_BSI_LTE) {
      if (value < dmin) out_of_range = true;
      else if (value > dmax) bv = dmax - f.base + (op == FBK_BSI_LT ? 1 : 0);
      else bv = value - f.base;
    } else {
      if (value < dmin || value > dmax) out_of_range = true;
      else bv = value - f.base;
    }
    if (out_of_range && op != FBK_BSI_NEQ) return empty_set(f);
    if ((op == FBK_BSI_LT && value > f.max) || (op == FBK_BSI_LTE && value >= f.max) || (op == FBK_BSI_GT && value < f.min) ||
        (op == FBK_BSI_GTE && value <= f.min))
      return not_null(f);
    if (out_of_range && op == FBK_BSI_NEQ) return not_null(f);
    check(fbk_bsi_range(idx_.ctx_, f.batch, base.data(), uint32_t(base.size()), op, f.bit_depth, bv, FBK_SETOP_OPTIMIZE, &o, nullptr));
    return fresh(o);
  }
  RowSet eval(const Call& c) {
    switch (c.kind) {
      case Call::kRow: return leaf_row(c.field, c.row);
      case Call::kRange:
      case Call::kBetween: return range(c);
      case Call::kAll: return leaf_row(Index::kExistence, 0);
      case Call::kShift: {
        if (c.children.size() != 1) throw Error(FBK_E_INVALID, c.children.empty() ? "Shift() requires an input row" : "Shift() only accepts a single row input");
        if (c.value < 0) throw Error(FBK_E_INVALID, "cannot shift by negative values");  // row.go:375-377
        RowSet acc = eval(c.children[0]);
        const std::vector<uint64_t>& sh = shards();
        for (int64_t i = 0; i < c.value; ++i) {  // Row.Shift: n single-column shifts (row.go:383-393)
          std::vector<uint32_t> carry(sh.size(), FBK_NO_ROW);
          for (size_t k = 1; k < sh.size(); ++k)
            if (sh[k - 1] + 1 == sh[k]) carry[k] = acc.rows()[k - 1];  // last column of the previous shard
          fbk_batch* o = nullptr;
          check(fbk_shift(idx_.ctx_, acc.batch(), acc.rows().data(), carry.data(), sh.size(), FBK_SETOP_OPTIMIZE, &o, nullptr));
          acc = fresh(o);
        }
        return acc;
      }
      case Call::kNot: {
        if (c.children.size() != 1) throw Error(FBK_E_INVALID, "Not() requires exactly one child");
        RowSet ex = leaf_row(Index::kExistence, 0), child = eval(c.children[0]);
        return setop(FBK_OP_ANDNOT, ex, child);
      }
      default: break;
    }
    if (c.children.empty()) throw Error(FBK_E_INVALID, "empty call");  // e.g. "Intersect() requires at least 1 child"
    const int32_t op = c.kind == Call::kIntersect ? FBK_OP_AND : c.kind == Call::kUnion ? FBK_OP_OR : c.kind == Call::kXor ? FBK_OP_XOR : FBK_OP_ANDNOT;
    RowSet acc = eval(c.children[0]);
    for (size_t i = 1; i < c.children.size(); ++i) {  // left fold, child by child
      RowSet next = eval(c.children[i]);
      acc = setop(op, acc, next);
    }
    return acc;
  }
  uint64_t count_rows(const RowSet& r) {
    std::vector<uint64_t> counts(r.rows().size());
    if (!counts.empty()) check(fbk_count(idx_.ctx_, r.batch(), r.rows().data(), counts.size(), counts.data()));
    uint64_t n = 0;
    for (uint64_t c : counts) n += c;
    return n;
  }
  std::vector<uint64_t> columns(const RowSet& r) {
    // gather the result rows into one batch, download, expand
    fbk_batch* o = nullptr;
    check(fbk_setop(idx_.ctx_, FBK_OP_OR, r.batch(), r.rows().data(), r.batch(), r.rows().data(), r.rows().size(), 0, &o, nullptr));
    uint32_t n_rows = 0;
    uint64_t nc = 0, pb = 0;
    check(fbk_batch_info(idx_.ctx_, o, &n_rows, &nc, &pb));
    std::vector<fbk_container_desc> descs(nc ? nc : 1);
    std::vector<uint8_t> payload(pb ? pb : 1);
    int32_t rc = fbk_batch_download(idx_.ctx_, o, descs.data(), nc, payload.data(), pb);
    fbk_batch_free(idx_.ctx_, o);
    check(rc);
    std::vector<uint64_t> out;
    for (uint64_t i = 0; i < nc; ++i) {
      const fbk_container_desc& d = descs[i];
      const uint64_t hb = (shards()[d.row] * 16 + (d.key & 15)) << 16;
      const uint64_t* w = reinterpret_cast<const uint64_t*>(payload.data() + d.off);  // keep-bitmap output: 1024 words
      for (uint32_t k = 0; k < FBK_BITMAP_WORDS; ++k)
        for (uint64_t x = w[k]; x; x &= x - 1) out.push_back(hb | (uint64_t(k) * 64 + uint64_t(__builtin_ctzll(x))));
    }
    std::sort(out.begin(), out.end());
    return out;
  }
  ValCount minmax(const std::string& field, const Call* filter, bool is_min) {
    Scope sc(*this, {filter});
    const Index::IntField& f = idx_.ints_.at(field);
    const size_t n = shards().size();
    ValCount out;
    if (n == 0) return out;
    std::vector<uint32_t> base = base_rows(f);
    std::vector<int64_t> vals(n);
    std::vector<uint64_t> counts(n);
    auto fn = is_min ? fbk_bsi_min : fbk_bsi_max;
    if (filter) {
      RowSet fr = eval(*filter);
      check(fn(idx_.ctx_, f.batch, base.data(), uint32_t(n), f.bit_depth, fr.batch(), fr.rows().data(), vals.data(), counts.data()));
    } else {
      check(fn(idx_.ctx_, f.batch, base.data(), uint32_t(n), f.bit_depth, nullptr, nullptr, vals.data(), counts.data()));
    }
    for (size_t s = 0; s < n; ++s) {
      // Field.MinForShard / MaxForShard: (0, 0) when the shard has no value, else value + Base (field.go:1590, 1620)
      const ValCount v = counts[s] ? ValCount{vals[s] + f.base, int64_t(counts[s])} : ValCount{};
      out = is_min ? out.Smaller(v) : out.Larger(v);
    }
    return out;
  }
  std::vector<uint32_t> field_rows(const Index::SetField& f) {
    const size_t n = shards().size(), nr = f.row_ids.size();
    std::vector<uint32_t> rows(n * nr);
    for (size_t s = 0; s < n; ++s)
      for (size_t i = 0; i < nr; ++i) {
        auto it = f.ordinal.find({shards()[s], f.row_ids[i]});
        rows[s * nr + i] = it == f.ordinal.end() ? f.empty_row : it->second;
      }
    return rows;
  }
  bool emit(const std::vector<FieldRow>& group, uint64_t count, const RowSet* members, const std::string& agg_field, uint64_t limit,
            std::vector<GroupCount>& out) {
    GroupCount g;
    g.Group = group;
    g.Count = count;
    if (!agg_field.empty()) {  // aggregate=Sum(field): Count is the number of columns WITH a value (executor.go:8905-8913)
      const Index::IntField& f = idx_.ints_.at(agg_field);
      std::vector<uint32_t> base = base_rows(f);
      const size_t n = base.size();
      std::vector<int64_t> sums(n);
      std::vector<uint64_t> counts(n);
      check(fbk_bsi_sum(idx_.ctx_, f.batch, base.data(), uint32_t(n), f.bit_depth, members->batch(), members->rows().data(), sums.data(), counts.data()));
      ValCount v;
      for (size_t s = 0; s < n; ++s) v = v.Add({sums[s] + int64_t(counts[s]) * f.base, int64_t(counts[s])});
      g.Count = uint64_t(v.Count);
      g.Agg = v.Val;
    }
    if (g.Count == 0) return true;
    out.push_back(std::move(g));
    return !(limit && out.size() >= limit);
  }
  // fields[level..]: the last two levels are one count-matrix call; earlier levels materialise
  // prefix ∩ row (gbi.rows[i].row.Intersect(gbi.rows[i-1].row), executor.go:8829-8834)
  bool group_by_rec(const std::vector<std::string>& fields, size_t level, const RowSet* prefix, const std::string& agg_field, uint64_t limit,
                    std::vector<FieldRow>& group, std::vector<GroupCount>& out) {
    const size_t n = shards().size();
    const Index::SetField& fa = idx_.sets_.at(fields[level]);
    const size_t na = fa.row_ids.size();
    if (na == 0) return true;
    std::vector<uint32_t> rows_a = field_rows(fa);
    const size_t remaining = fields.size() - level;
    if (remaining == 1) {
      std::vector<uint64_t> tot(na, 0);
      if (prefix) {
        check(fbk_count_matrix(idx_.ctx_, fa.batch, rows_a.data(), uint32_t(na), prefix->batch(), prefix->rows().data(), 1, nullptr, nullptr,
                               uint32_t(n), tot.data(), nullptr));
      } else {
        std::vector<uint64_t> c(n * na);
        check(fbk_count(idx_.ctx_, fa.batch, rows_a.data(), c.size(), c.data()));
        for (size_t s = 0; s < n; ++s)
          for (size_t i = 0; i < na; ++i) tot[i] += c[s * na + i];
      }
      for (size_t i = 0; i < na; ++i) {
        if (!tot[i]) continue;
        group.push_back({fields[level], fa.row_ids[i]});
        bool go = true;
        if (agg_field.empty()) {
          go = emit(group, tot[i], nullptr, agg_field, limit, out);
        } else {
          RowSet r = leaf_row(fields[level], fa.row_ids[i]);
          if (prefix) {
            RowSet m = setop(FBK_OP_AND, r, *prefix);
            go = emit(group, tot[i], &m, agg_field, limit, out);
          } else {
            go = emit(group, tot[i], &r, agg_field, limit, out);
          }
        }
        group.pop_back();
        if (!go) return false;
      }
      return true;
    }
    if (remaining == 2) {
      const Index::SetField& fb = idx_.sets_.at(fields[level + 1]);
      const size_t nb = fb.row_ids.size();
      if (nb == 0) return true;
      std::vector<uint32_t> rows_b = field_rows(fb);
      std::vector<uint64_t> tot(na * nb, 0);
      check(fbk_count_matrix(idx_.ctx_, fa.batch, rows_a.data(), uint32_t(na), fb.batch, rows_b.data(), uint32_t(nb), prefix ? prefix->batch() : nullptr,
                             prefix ? prefix->rows().data() : nullptr, uint32_t(n), tot.data(), nullptr));
      for (size_t i = 0; i < na; ++i)
        for (size_t j = 0; j < nb; ++j) {
          if (!tot[i * nb + j]) continue;
          group.push_back({fields[level], fa.row_ids[i]});
          group.push_back({fields[level + 1], fb.row_ids[j]});
          bool go = true;
          if (agg_field.empty()) {
            go = emit(group, tot[i * nb + j], nullptr, agg_field, limit, out);
          } else {
            RowSet ra = leaf_row(fields[level], fa.row_ids[i]), rb = leaf_row(fields[level + 1], fb.row_ids[j]);
            RowSet m = setop(FBK_OP_AND, ra, rb);
            if (prefix) {
              RowSet m2 = setop(FBK_OP_AND, m, *prefix);
              go = emit(group, tot[i * nb + j], &m2, agg_field, limit, out);
            } else {
              go = emit(group, tot[i * nb + j], &m, agg_field, limit, out);
            }
          }
          group.pop_back();
          group.pop_back();
          if (!go) return false;
        }
      return true;
    }
    for (size_t i = 0; i < na; ++i) {  // three or more fields left: fix this field's row
      RowSet r = leaf_row(fields[level], fa.row_ids[i]);
      group.push_back({fields[level], fa.row_ids[i]});
      bool go;
      if (prefix) {
        RowSet p = setop(FBK_OP_AND, r, *prefix);
        go = count_rows(p) == 0 ? true : group_by_rec(fields, level + 1, &p, agg_field, limit, group, out);
      } else {
        go = group_by_rec(fields, level + 1, &r, agg_field, limit, group, out);
      }
      group.pop_back();
      if (!go) return false;
    }
    return true;
  }

  Index& idx_;
};

}  // namespace fbk
