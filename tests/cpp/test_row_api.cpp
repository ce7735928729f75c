// Restatement of the reference's Row tests (row_test.go:47-137: TestRow_Xor,
// TestRow_Union_Segment, TestRow_Difference_Segment, TestRow_IsEmpty) and the executor
// vectors (executor_test.go:1236-1373) against the C++ host mirror include/fbk_roaring.hpp,
// which runs every Row operation on the GPU through the C ABI.
//   g++ -std=c++17 -I include tests/cpp/test_row_api.cpp -L featurebase_amd/csrc -lfbk -o build/test_row_api
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fbk_roaring.hpp"

using fbk::Row;
using fbk::ShardWidth;
typedef std::vector<uint64_t> V;

static int failures = 0;
#define EXPECT(cond)                                                   \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
      ++failures;                                                      \
    }                                                                  \
  } while (0)

int main() {
  fbk::Device dev(0);
  {  // TestRow_Xor, row_test.go:47-68
    Row r1 = Row::NewRow({0, 1, ShardWidth}), r2 = Row::NewRow({0, 2 * ShardWidth});
    V exp = {1, ShardWidth, 2 * ShardWidth};
    Row res = dev.Xor(r1, r2);
    EXPECT(res.Count() == 3);
    EXPECT(res.Columns() == exp);
    res = dev.Xor(r2, r1);
    EXPECT(res.Count() == 3);
    EXPECT(res.Columns() == exp);
  }
  {  // TestRow_Union_Segment, row_test.go:69-89
    Row r1 = Row::NewRow({0, 1, ShardWidth}), r2 = Row::NewRow({0, 2 * ShardWidth});
    V exp = {0, 1, ShardWidth, 2 * ShardWidth};
    Row res = dev.Union(r1, r2);
    EXPECT(res.Count() == 4);
    EXPECT(res.Columns() == exp);
    res = dev.Union(r2, r1);
    EXPECT(res.Count() == 4);
    EXPECT(res.Columns() == exp);
  }
  {  // TestRow_Difference_Segment, row_test.go:91-103
    Row r1 = Row::NewRow({0, 1, ShardWidth}), r2 = Row::NewRow({0, 2 * ShardWidth});
    V exp = {1, ShardWidth};
    Row res = dev.Difference(r1, r2);
    EXPECT(res.Count() == 2);
    EXPECT(res.Columns() == exp);
  }
  {  // Row.Shift (row.go:374-396) with the columns of TestExecutor_Execute_Shift (executor_test.go:6590-6676)
    const uint64_t SW = ShardWidth;
    EXPECT(dev.Shift(Row::NewRow({0}), 1).Columns() == (V{1}));
    EXPECT(dev.Shift(dev.Shift(Row::NewRow({0}), 1), 1).Columns() == (V{2}));
    EXPECT(dev.Shift(Row::NewRow({65535}), 1).Columns() == (V{65536}));
    EXPECT(dev.Shift(Row::NewRow({1, SW - 1, SW + 1}), 1).Columns() == (V{2, SW, SW + 2}));
    EXPECT(dev.Shift(Row::NewRow({1, SW - 1, SW + 1}), 2).Columns() == (V{3, SW + 1, SW + 3}));
    EXPECT(dev.Shift(Row::NewRow({SW - 2, SW - 1, SW, SW + 2}), 1).Columns() == (V{SW - 1, SW, SW + 1, SW + 3}));
    Row far = dev.Shift(Row::NewRow({SW - 1, 5 * SW - 1}), 1);  // carried into shards that held nothing
    EXPECT(far.Columns() == (V{SW, 5 * SW}));
    EXPECT(far.Count() == 2 && far.Segments.size() == 2 && far.Segments[0].shard == 1 && far.Segments[1].shard == 5);
    EXPECT(dev.Shift(Row::NewRow({7}), 0).Columns() == (V{7}));
    bool threw = false;
    try {
      dev.Shift(Row::NewRow({7}), -1);
    } catch (const fbk::Error&) {
      threw = true;
    }
    EXPECT(threw);
  }
  {  // TestRow_IsEmpty, row_test.go:105-116
    Row r1 = Row::NewRow({1, ShardWidth}), r2 = Row::NewRow({0, 2 * ShardWidth});
    Row res = dev.Intersect(r2, r1);
    EXPECT(r1.Any());
    EXPECT(!res.Any());
    EXPECT(dev.IntersectionCount(r2, r1) == 0);
  }
  {  // executor_test.go:1236-1373 (Difference / Intersect / Union / Xor / Count)
    Row a = Row::NewRow({1, 2, 3}), b = Row::NewRow({2, 4});
    EXPECT(dev.Difference(a, b).Columns() == (V{1, 3}));
    a = Row::NewRow({1, ShardWidth + 1, ShardWidth + 2});
    b = Row::NewRow({1, 2, ShardWidth + 2});
    EXPECT(dev.Intersect(a, b).Columns() == (V{1, ShardWidth + 2}));
    EXPECT(dev.IntersectionCount(a, b) == 2);
    a = Row::NewRow({0, ShardWidth + 1, ShardWidth + 2});
    b = Row::NewRow({2, ShardWidth + 2});
    EXPECT(dev.Union(a, b).Columns() == (V{0, 2, ShardWidth + 1, ShardWidth + 2}));
    EXPECT(dev.Xor(a, b).Columns() == (V{0, 2, ShardWidth + 1}));
    EXPECT(Row::NewRow({3, ShardWidth + 1, ShardWidth + 2}).Count() == 3);
  }
  {  // a dense row (bitmap containers) against a sparse one, result re-encoded by optimize()
    V cols;
    for (uint64_t c = 0; c < 200000; c += 3) cols.push_back(c);
    Row a = Row::NewRow(cols), b = Row::NewRow({0, 3, 4, 65536 * 2 + 2, 199998, 5 * ShardWidth});
    EXPECT(dev.Intersect(a, b).Columns() == (V{0, 3, 199998}));
    EXPECT(dev.IntersectionCount(a, b) == 3);
    Row d = dev.Difference(a, b);
    EXPECT(d.Count() == cols.size() - 3);
    Row u = dev.Union(a, b);
    EXPECT(u.Count() == cols.size() + 3);
  }
  if (failures) {
    std::printf("%d failure(s)\n", failures);
    return 1;
  }
  std::printf("row api ok\n");
  return 0;
}
