// Concurrent callers must OVERLAP on the device: the reference runs ~NumCPU shard workers at once
// (executor.go:6723-6737).  One fbk_ctx is one stream behind one mutex, so every calling thread
// takes its own fbk_ctx_fork (own stream, lock, staging, pool; shared fragment cache).  This test
// runs the same 8 x M small queries (a) from one thread on the root context and (b) from 8 threads
// on 8 forks, checks that all answers are identical, and asserts wall(b) < 0.5 x wall(a).
// It also checks that error messages are per context (fbk_last_error_r), not per thread.
//   g++ -std=c++17 -pthread -I include tests/cpp/test_forks.cpp -L featurebase_amd/csrc -lfbk
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "fbk.h"

static int failures = 0;
#define EXPECT(cond)                                               \
  do {                                                             \
    if (!(cond)) {                                                 \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                  \
    }                                                              \
  } while (0)

static uint64_t next(uint64_t& s) {  // splitmix64
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main() {
  fbk_ctx* root = nullptr;
  if (fbk_open(0, 0, &root) != FBK_OK) {
    std::printf("fbk_open: %s\n", fbk_last_error(nullptr));
    return 1;
  }
  constexpr uint32_t kRows = 128;  // 64 row pairs of 16 bitmap containers = 16 MiB
  std::vector<uint64_t> words(uint64_t(kRows) * 16 * 1024);
  uint64_t seed = 42;
  for (auto& w : words) w = next(seed);
  fbk_batch* batch = nullptr;
  EXPECT(fbk_batch_upload_dense(root, words.data(), kRows, &batch) == FBK_OK);

  constexpr int kThreads = 8, kCalls = 150, kPairs = 16;
  // thread t, call c: pairs (a, b) drawn from the batch's rows
  auto rows_of = [&](int t, int c, uint32_t* ra, uint32_t* rb) {
    uint64_t s = 1000003ull * uint64_t(t) + uint64_t(c);
    for (int i = 0; i < kPairs; ++i) {
      ra[i] = uint32_t(next(s) % kRows);
      rb[i] = uint32_t(next(s) % kRows);
    }
  };
  std::vector<uint64_t> serial(uint64_t(kThreads) * kCalls * kPairs), forked(serial.size());
  auto run = [&](fbk_ctx* ctx, int t, uint64_t* out) {
    uint32_t ra[kPairs], rb[kPairs];
    for (int c = 0; c < kCalls; ++c) {
      rows_of(t, c, ra, rb);
      if (fbk_intersection_count(ctx, batch, ra, batch, rb, kPairs, out + (uint64_t(t) * kCalls + c) * kPairs) != FBK_OK) ++failures;
    }
  };
  for (int t = 0; t < kThreads; ++t) run(root, t, serial.data());  // warm-up (pool, code objects)
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < kThreads; ++t) run(root, t, serial.data());
  const double wall_serial = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  std::vector<fbk_ctx*> forks(kThreads, nullptr);
  for (int t = 0; t < kThreads; ++t) EXPECT(fbk_ctx_fork(root, &forks[t]) == FBK_OK);
  {
    std::vector<std::thread> th;  // warm-up of every fork
    for (int t = 0; t < kThreads; ++t) th.emplace_back(run, forks[t], t, forked.data());
    for (auto& x : th) x.join();
  }
  t0 = std::chrono::steady_clock::now();
  {
    std::vector<std::thread> th;
    for (int t = 0; t < kThreads; ++t) th.emplace_back(run, forks[t], t, forked.data());
    for (auto& x : th) x.join();
  }
  const double wall_forked = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  EXPECT(serial == forked);
  std::printf("serial %.1f ms, 8 forks %.1f ms, ratio %.2f\n", wall_serial * 1e3, wall_forked * 1e3, wall_forked / wall_serial);
  EXPECT(wall_forked < 0.5 * wall_serial);

  // per-context error messages: fork 1 fails; fork 2's message stays empty; the failing call's
  // message is readable from ANOTHER thread (what a rescheduled goroutine does)
  uint32_t bad = 1u << 30;
  uint64_t dummy = 0;
  EXPECT(fbk_intersection_count(forks[1], batch, &bad, batch, &bad, 1, &dummy) == FBK_E_INVALID);
  char buf1[256] = {0}, buf2[256] = {0};
  int32_t code1 = 0, code2 = 0;
  std::thread other([&] {
    fbk_last_error_r(forks[1], buf1, sizeof buf1, &code1);
    fbk_last_error_r(forks[2], buf2, sizeof buf2, &code2);
  });
  other.join();
  EXPECT(code1 == FBK_E_INVALID && std::strstr(buf1, "out of range") != nullptr);
  EXPECT(code2 == 0 && buf2[0] == 0);

  // a root cannot be closed before its forks
  EXPECT(fbk_close(root) == FBK_E_INVALID);
  for (fbk_ctx* f : forks) EXPECT(fbk_close(f) == FBK_OK);
  EXPECT(fbk_batch_free(root, batch) == FBK_OK);
  EXPECT(fbk_close(root) == FBK_OK);
  if (failures) {
    std::printf("%d failure(s)\n", failures);
    return 1;
  }
  std::printf("forks ok\n");
  return 0;
}
