// Restatement of the reference's executor-level tests against the C++ host mirror
// include/fbk_executor.hpp (every operator runs on the GPU through the C ABI):
//   TestExecutor_Execute_Difference / Intersect / Union / Xor / Count   executor_test.go:1236-1373
//   TestExecutor_ExecuteTopK                                            executor_test.go:1758-1809
//   TestExecutor_Execute_Sum                                            executor_test.go:2813-2871
//   TestExecutor_Execute_MinMax (ColumnID)                              executor_test.go:2510-2570
//   TestExecutor_Execute_GroupBy (Basic / Filter / Aggregate)           executor_test.go:6035-6130
//   Row(f > k) style BSI range calls                                    executor_test.go:3051-3160
//   g++ -std=c++17 -I include tests/cpp/test_executor_api.cpp -L featurebase_amd/csrc -lfbk
#include <algorithm>
#include <cstdio>
#include <vector>

#include "fbk_executor.hpp"

using namespace fbk;
typedef std::vector<uint64_t> V;
static const uint64_t SW = ShardWidth;

static int failures = 0;
#define EXPECT(cond)                                                 \
  do {                                                               \
    if (!(cond)) {                                                   \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      ++failures;                                                    \
    }                                                                \
  } while (0)

static GroupCount G2(const char* f0, uint64_t r0, const char* f1, uint64_t r1, uint64_t count, int64_t agg = 0) {
  GroupCount g;
  g.Group = {{f0, r0}, {f1, r1}};
  g.Count = count;
  g.Agg = agg;
  return g;
}

int main() {
  {  // ---- set operations across shards (executor_test.go:1236-1373) ----
    Index idx;
    idx.CreateSetField("general");
    Executor e(idx);
    auto R = [](uint64_t r) { return Call::Row("general", r); };
    idx.SetBit("general", 10, 1);
    idx.SetBit("general", 10, 2);
    idx.SetBit("general", 10, 3);
    idx.SetBit("general", 11, 2);
    idx.SetBit("general", 11, 4);
    EXPECT(e.Columns(Call::Nary(Call::kDifference, {R(10), R(11)})) == (V{1, 3}));
    idx.SetBit("general", 20, 1);
    idx.SetBit("general", 20, SW + 1);
    idx.SetBit("general", 20, SW + 2);
    idx.SetBit("general", 21, 1);
    idx.SetBit("general", 21, 2);
    idx.SetBit("general", 21, SW + 2);
    EXPECT(e.Columns(Call::Nary(Call::kIntersect, {R(20), R(21)})) == (V{1, SW + 2}));
    idx.SetBit("general", 30, 0);
    idx.SetBit("general", 30, SW + 1);
    idx.SetBit("general", 30, SW + 2);
    idx.SetBit("general", 31, 2);
    idx.SetBit("general", 31, SW + 2);
    EXPECT(e.Columns(Call::Nary(Call::kUnion, {R(30), R(31)})) == (V{0, 2, SW + 1, SW + 2}));
    EXPECT(e.Columns(Call::Nary(Call::kXor, {R(30), R(31)})) == (V{0, 2, SW + 1}));
    idx.SetBit("general", 40, 3);
    idx.SetBit("general", 40, SW + 1);
    idx.SetBit("general", 40, SW + 2);
    EXPECT(e.Count(R(40)) == 3);
    // three-way folds, left to right (executeIntersectShard / executeDifferenceShard)
    EXPECT(e.Count(Call::Nary(Call::kUnion, {R(10), R(20), R(30)})) == 3 + 2 + 1);          // {1,2,3} ∪ {1,SW+1,SW+2} ∪ {0,SW+1,SW+2}
    EXPECT(e.Columns(Call::Nary(Call::kDifference, {R(20), R(21), R(30)})) == (V{}));       // {SW+1} \ {..,SW+1,..}
    EXPECT(e.Columns(Call::Nary(Call::kIntersect, {R(20), R(30), R(40)})) == (V{SW + 1, SW + 2}));
    EXPECT(e.Count(R(99)) == 0);  // a row that does not exist is empty
    // Not() / All() over the tracked existence row: every column set so far is
    // {0,1,2,3,4, SW+1, SW+2}; Not(Row(general=10)) = existence \ {1,2,3}
    EXPECT(e.Columns(Call::All()) == (V{0, 1, 2, 3, 4, SW + 1, SW + 2}));
    EXPECT(e.Columns(Call::Not(R(10))) == (V{0, 4, SW + 1, SW + 2}));
    EXPECT(e.Count(Call::Not(Call::Nary(Call::kUnion, {R(10), R(20), R(30)}))) == 1);  // only column 4 is left
  }
  {  // ---- Shift (TestExecutor_Execute_Shift, executor_test.go:6590-6676) ----
    auto shifted = [](std::initializer_list<uint64_t> cols, int64_t n, int nest = 1) {
      Index idx;
      idx.CreateSetField("general");
      for (uint64_t c : cols) idx.SetBit("general", 10, c);
      Executor e(idx);
      Call c = Call::Row("general", 10);
      for (int i = 0; i < nest; ++i) c = Call::Shift(std::move(c), n);
      return e.Columns(c);
    };
    EXPECT(shifted({0}, 1) == (V{1}));                 // "Shift Bit 0"
    EXPECT(shifted({0}, 1, 2) == (V{2}));              // Shift(Shift(..., n=1), n=1)
    EXPECT(shifted({65535}, 1) == (V{65536}));         // "Shift container boundary"
    EXPECT(shifted({1, SW - 1, SW + 1}, 1) == (V{2, SW, SW + 2}));  // "Shift shard boundary"
    EXPECT(shifted({1, SW - 1, SW + 1}, 1, 2) == (V{3, SW + 1, SW + 3}));
    EXPECT(shifted({SW - 2, SW - 1, SW, SW + 2}, 1) == (V{SW - 1, SW, SW + 1, SW + 3}));  // "no create"
    EXPECT(shifted({SW - 2, SW - 1, SW, SW + 2}, 1, 2) == (V{SW, SW + 1, SW + 2, SW + 4}));
    EXPECT(shifted({1, SW - 1, SW + 1}, 2) == (V{3, SW + 1, SW + 3}));  // n = 2 in one call (Row.Shift loops)
    EXPECT(shifted({SW - 1, 3 * SW - 1}, 1) == (V{SW, 3 * SW}));        // carried into shards that hold no data
    EXPECT(shifted({5}, 0) == (V{5}));
    {
      Index idx;
      idx.CreateSetField("general");
      idx.SetBit("general", 10, SW - 1);
      idx.SetBit("general", 11, SW);
      Executor e(idx);
      // the shifted row meets another row in the successor shard
      EXPECT(e.Count(Call::Nary(Call::kIntersect, {Call::Shift(Call::Row("general", 10), 1), Call::Row("general", 11)})) == 1);
      bool threw = false;
      try {
        e.Columns(Call::Shift(Call::Row("general", 10), -1));
      } catch (const Error&) {
        threw = true;  // "cannot shift by negative values", row.go:375-377
      }
      EXPECT(threw);
    }
  }
  {  // ---- TopK (executor_test.go:1758-1809): rows {0, 10, 20} -> {10: 4, 0: 3} ----
    Index idx;
    idx.CreateSetField("f");
    idx.CreateSetField("x");
    const uint64_t bits[][2] = {{0, 0}, {0, SW + 2}, {10, 2}, {10, SW}, {10, 2 * SW}, {10, SW + 1}, {20, SW}, {0, 1}};
    for (auto& b : bits) idx.SetBit("f", b[0], b[1]);
    Executor e(idx);
    std::vector<Pair> got = e.TopK("f", 2);
    EXPECT(got == (std::vector<Pair>{{10, 4}, {0, 3}}));
    EXPECT(e.TopK("f", 0) == (std::vector<Pair>{{10, 4}, {0, 3}, {20, 1}}));
    // with a filter row (TopK(f, k=3, filter=Row(x=1))): x=1 -> {0, SW, SW+1, 2*SW}
    for (uint64_t c : {uint64_t(0), SW, SW + 1, 2 * SW}) idx.SetBit("x", 1, c);
    Call flt = Call::Row("x", 1);
    EXPECT(e.TopK("f", 3, &flt) == (std::vector<Pair>{{10, 3}, {0, 1}, {20, 1}}));  // ties: ascending id (PivotDescending)
    EXPECT(e.TopN("f", 1, &flt) == (std::vector<Pair>{{10, 3}}));
  }
  {  // ---- Sum / Min / Max / ranges over an int field across 3 shards ----
    // executor_test.go:2813-2871 (Sum) and :2530-2570 (Min/Max): x=0 -> {0, 3, SW+1}, x=1 -> {1}, x=2 -> {SW+2}
    Index idx;
    idx.CreateSetField("x");
    idx.CreateIntField("f", -1110, 1000);
    idx.CreateIntField("foo", 10, 100);
    for (uint64_t c : {uint64_t(0), uint64_t(3), SW + 1}) idx.SetBit("x", 0, c);
    idx.SetBit("x", 1, 1);
    idx.SetBit("x", 2, SW + 2);
    const std::pair<uint64_t, int64_t> fv[] = {{0, 20}, {1, -5}, {2, -5}, {3, 10}, {SW, 30}, {SW + 2, 40}, {5 * SW + 100, 50}, {SW + 1, 60}};
    for (auto& cv : fv) idx.SetValue("f", cv.first, cv.second);
    const std::pair<uint64_t, int64_t> foo[] = {{0, 20}, {SW, 30}, {SW + 2, 40}, {5 * SW + 100, 50}, {SW + 1, 60}};
    for (auto& cv : foo) idx.SetValue("foo", cv.first, cv.second);
    Executor e(idx);
    Call x0 = Call::Row("x", 0), x1 = Call::Row("x", 1), x2 = Call::Row("x", 2);
    EXPECT(e.Sum("foo") == (ValCount{200, 5}));         // Sum(field=foo): base 10, executor_test.go:2846
    EXPECT(e.Sum("foo", &x0) == (ValCount{80, 2}));     // Sum(Row(x=0), field=foo): :2856
    EXPECT(e.Min("f") == (ValCount{-5, 2}));
    EXPECT(e.Min("f", &x0) == (ValCount{10, 1}));
    EXPECT(e.Min("f", &x1) == (ValCount{-5, 1}));
    EXPECT(e.Min("f", &x2) == (ValCount{40, 1}));
    EXPECT(e.Max("f") == (ValCount{60, 1}));
    EXPECT(e.Max("f", &x0) == (ValCount{60, 1}));
    EXPECT(e.Max("f", &x1) == (ValCount{-5, 1}));
    EXPECT(e.Max("f", &x2) == (ValCount{40, 1}));
    // Row(f > 20) etc: the predicate goes through baseValue (field.go:2412)
    EXPECT(e.Columns(Call::Range("f", FBK_BSI_GT, 20)) == (V{SW, SW + 1, SW + 2, 5 * SW + 100}));
    EXPECT(e.Columns(Call::Range("f", FBK_BSI_LT, 0)) == (V{1, 2}));
    EXPECT(e.Columns(Call::Range("f", FBK_BSI_EQ, -5)) == (V{1, 2}));
    EXPECT(e.Columns(Call::Range("f", FBK_BSI_NEQ, -5)) == (V{0, 3, SW, SW + 1, SW + 2, 5 * SW + 100}));
    EXPECT(e.Columns(Call::Range("f", FBK_BSI_GTE, -2000)) == (V{0, 1, 2, 3, SW, SW + 1, SW + 2, 5 * SW + 100}));  // below Min: not-null
    EXPECT(e.Columns(Call::Range("f", FBK_BSI_GT, 5000)) == (V{}));                                             // beyond the bit depth
    EXPECT(e.Columns(Call::Between("f", 10, 40)) == (V{0, 3, SW, SW + 2}));
    EXPECT(e.Columns(Call::Range("foo", FBK_BSI_LTE, 30)) == (V{0, SW}));  // base 10: stored magnitudes are value - 10
    EXPECT(e.Count(Call::Nary(Call::kIntersect, {Call::Range("f", FBK_BSI_GT, 0), x0})) == 3);
    // Distinct(field=f): f = {20, -5, -5, 10, 30, 40, 50, 60}; Distinct(Row(x=0), field=f): cols {0, 3, SW+1}
    EXPECT(e.Distinct("f") == (std::vector<int64_t>{-5, 10, 20, 30, 40, 50, 60}));
    EXPECT(e.Distinct("f", &x0) == (std::vector<int64_t>{10, 20, 60}));
    EXPECT(e.Distinct("foo") == (std::vector<int64_t>{20, 30, 40, 50, 60}));  // Base 10 added back
  }
  {  // ---- GroupBy (executor_test.go:6035-6130) ----
    Index idx;
    idx.CreateSetField("general");
    idx.CreateSetField("sub");
    idx.CreateIntField("v", 0, 1000);
    const uint64_t gen[][2] = {{10, 0}, {10, 1}, {10, SW + 1}, {11, 2}, {11, SW + 2}, {12, 2}, {12, SW + 2}};
    for (auto& b : gen) idx.SetBit("general", b[0], b[1]);
    const uint64_t sub[][2] = {{100, 0}, {100, 1}, {100, 3}, {100, SW + 1}, {110, 2}, {110, 0}};
    for (auto& b : sub) idx.SetBit("sub", b[0], b[1]);
    idx.SetValue("v", 0, 10);
    idx.SetValue("v", 1, 100);
    idx.SetValue("v", SW + 10, 100);
    Executor e(idx);
    std::vector<GroupCount> basic = {G2("general", 10, "sub", 100, 3), G2("general", 10, "sub", 110, 1), G2("general", 11, "sub", 110, 1),
                                     G2("general", 12, "sub", 110, 1)};
    EXPECT(e.GroupBy({"general", "sub"}) == basic);
    Call f10 = Call::Row("general", 10);
    EXPECT(e.GroupBy({"general", "sub"}, &f10) == (std::vector<GroupCount>{basic[0], basic[1]}));
    // aggregate=Sum(field=v): counts are the columns that HAVE a value: (10,100) -> cols {0,1}: 2 / 110; (10,110) -> col 0: 1 / 10
    EXPECT(e.GroupBy({"general", "sub"}, nullptr, "v") ==
           (std::vector<GroupCount>{G2("general", 10, "sub", 100, 2, 110), G2("general", 10, "sub", 110, 1, 10)}));
    EXPECT(e.GroupBy({"general", "sub"}, nullptr, "", 2) == (std::vector<GroupCount>{basic[0], basic[1]}));  // limit=2
    // one field: Rows(general) counts
    std::vector<GroupCount> one = e.GroupBy({"general"});
    EXPECT(one.size() == 3 && one[0].Count == 3 && one[1].Count == 2 && one[2].Count == 2 && one[0].Group[0].RowID == 10);
    // three fields: general x sub x general (odometer, last field fastest)
    std::vector<GroupCount> three = e.GroupBy({"general", "sub", "general"});
    EXPECT(three.size() == 6);  // (10,100,10):3 (10,110,10):1 (11,110,11):1 (11,110,12):1 (12,110,11):1 (12,110,12):1
    if (three.size() == 6) {
      EXPECT(three[0].Count == 3 && three[0].Group[2].RowID == 10);
      EXPECT(three[3].Group[0].RowID == 11 && three[3].Group[2].RowID == 12 && three[3].Count == 1);
    }
    bool threw = false;
    try {
      e.GroupBy({});
    } catch (const Error&) {
      threw = true;
    }
    EXPECT(threw);  // "need at least one child call"
  }
  {  // ---- Percentile (executor_test.go:7587-7760): expectations from the reference's own checker ----
    // Two host-side expectations over plain arrays: `checker` is the reference test's own
    // getExpectedPercentile (executor_test.go:7631-7677); `exec` walks executePercentile's loop
    // (executor.go:1396-1595).  They agree whenever the search ends "balanced"; when the bounds
    // cross instead, executePercentile returns its LAST guess while the checker returns min —
    // a quirk of the reference that parity keeps (the reference's test data never hits it).
    auto checker = [](const std::vector<int64_t>& nums, double nth, bool* balanced) -> int64_t {
      *balanced = true;
      int64_t mn = nums[0], mx = nums[0];
      for (int64_t v : nums) {
        mn = std::min(mn, v);
        mx = std::max(mx, v);
      }
      if (nth == 0.0) return mn;
      if (nth == 100.0) return mx;
      int64_t guess = 0;
      const int less = int((double(nums.size()) * nth) / 100.0), greater = int((double(nums.size()) * (100 - nth)) / 100.0);
      if (less == 0) return mn;
      if (greater == 0) return mx;
      while (mn < mx) {
        guess = ((mx / 2) + (mn / 2)) + (((mx % 2) + (mn % 2)) / 2);
        int l = 0, r = 0;
        for (int64_t v : nums) {
          if (v < guess) ++l;
          else if (v > guess) ++r;
        }
        if (l > less) mx = guess - 1;
        else if (r > greater) mn = guess + 1;
        else return guess;
      }
      *balanced = false;
      return mn;
    };
    auto exec = [](const std::vector<int64_t>& nums, double nth) -> int64_t {
      int64_t mn = nums[0], mx = nums[0];
      for (int64_t v : nums) {
        mn = std::min(mn, v);
        mx = std::max(mx, v);
      }
      const uint64_t less = uint64_t((double(nums.size()) * nth) / 100.0), greater = uint64_t((double(nums.size()) * (100 - nth)) / 100.0);
      if (greater != 0 && less == 0) return mn;
      if (greater == 0) return mx;
      int64_t guess = mn;
      while (mn < mx) {
        guess = (mn / 2) + (mx / 2) + (((mn % 2) + (mx % 2)) / 2);
        uint64_t l = 0, r = 0;
        for (int64_t v : nums) {
          if (v < guess) ++l;
          else if (v > guess) ++r;
        }
        if (l > less) {
          mx = guess - 1;
          continue;
        }
        if (r > greater) {
          mn = guess + 1;
          continue;
        }
        break;
      }
      return guess;
    };
    Index idx;
    idx.CreateSetField("val");
    uint64_t seed = 42;
    auto rnd = [&seed]() {
      seed += 0x9E3779B97F4A7C15ull;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      return z ^ (z >> 31);
    };
    std::vector<std::pair<uint64_t, int64_t>> vals;
    std::vector<int64_t> foo_nums, all_nums;
    int64_t lo = 0, hi = 0;
    for (int i = 0; i < 100; ++i) {  // 100 users spread over 3 shards, values +-uint32, "foo"/"bar" coin flip
      int64_t num = int64_t(rnd() & 0xFFFFFFFFu);
      if (rnd() % 2 == 0) num = -num;
      const uint64_t col = uint64_t(i) * 31 + (uint64_t(i) % 3) * SW;
      const bool foo = rnd() % 2 == 0;
      vals.push_back({col, num});
      idx.SetBit("val", foo ? 0 : 1, col);
      (foo ? foo_nums : all_nums).push_back(num);
      lo = std::min(lo, num);
      hi = std::max(hi, num);
    }
    all_nums.insert(all_nums.end(), foo_nums.begin(), foo_nums.end());
    idx.CreateIntField("net_worth", lo, hi);
    for (auto& cv : vals) idx.SetValue("net_worth", cv.first, cv.second);
    Executor e(idx);
    Call foo = Call::Row("val", 0);
    int balanced_cases = 0;
    for (double nth : {0.0, 10.0, 25.0, 50.0, 75.0, 90.0, 99.0, 100.0}) {
      ValCount got;
      bool bal = false;
      EXPECT(e.Percentile("net_worth", nth, &foo, &got));
      EXPECT(got.Val == exec(foo_nums, nth) && got.Count >= 1);
      const int64_t chk = checker(foo_nums, nth, &bal);
      if (bal) {
        EXPECT(got.Val == chk);
        ++balanced_cases;
      }
      EXPECT(e.Percentile("net_worth", nth, nullptr, &got));
      EXPECT(got.Val == exec(all_nums, nth));
      const int64_t chk2 = checker(all_nums, nth, &bal);
      if (bal) {
        EXPECT(got.Val == chk2);
        ++balanced_cases;
      }
    }
    EXPECT(balanced_cases >= 8);
    ValCount none;
    Call nobody = Call::Row("val", 77);
    EXPECT(!e.Percentile("net_worth", 50, &nobody, &none));  // the median of nothing is NULL
  }
  if (failures) {
    std::printf("%d failure(s)\n", failures);
    return 1;
  }
  std::printf("executor api ok\n");
  return 0;
}
