// matrix_xcd_hist.hip — where does a dense count-matrix launch lose time against a longer one?  (not part of the product)
//
// BASELINE configs[3]'s per-rank launch — 1024 shards x (32 A rows + 32 B rows + filter), dense rows, k_count_matrix_mfma<true, 4, 2, nt, 1, 1>,
// 4 slots per block = 4096 blocks — reaches 0.77-0.79 of 8 TB/s where the same kernel over 8192 shards (one block per shard) reaches 0.79-0.80.
// This harness runs the kernel built with -DFBK_MM_STAMPS: every block records its start and end (s_memrealtime, 100 MHz) and the XCD / CU it
// ran on.  From the stamps of one launch:
//   * ramp: slot-time lost before each of the 512 block slots (2 per CU) got its first block; drain: slot-time lost behind each slot's last block;
//   * the blocks' own durations by position in the launch (first wave of blocks, middle, last);
//   * per XCD: blocks executed, when its last block ended relative to the launch's end, CUs seen.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFBK_MM_STAMPS scripts/matrix_xcd_hist.hip -o scripts/matrix_xcd_hist
//   scripts/matrix_xcd_hist [shards=1024] [spb=4] [launches=5] [tickets=0] > profiles/r06_matrix_xcd_hist_1024.txt
// tickets = 1: the launch as the library issues it since round 6 (units by ticket, three tiers, spare blocks: mm_ticket_plan);
// tickets = 2: both forms alternating, launch by launch, on the same memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "../featurebase_amd/csrc/fbk_matrix_mfma.hip.h"

using fbk::u64;

#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                               \
    }                                                        \
  } while (0)

__global__ void k_fill(u64* p, size_t n, u64 seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    u64 z = (i + seed) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    p[i] = z ^ (z >> 31);
  }
}

static double med(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

// one launch's stamps -> the report
static void analyse(const std::vector<u64>& st_all, uint32_t grid, float ms, double bytes, int it, const char* label, uint32_t kSlotsRunning) {
  uint32_t blocks;
  const bool by_ticket = false;
  (void)by_ticket;
    std::vector<u64> st;  // the blocks that ran a unit (a spare block of a ticketed launch leaves no stamp)
    for (uint32_t b = 0; b < grid; ++b)
      if (st_all[4 * (size_t)b + 1])
        for (int k = 0; k < 4; ++k) st.push_back(st_all[4 * (size_t)b + k]);
    blocks = (uint32_t)(st.size() / 4);
    u64 T0 = ~0ull, T1 = 0;
    for (uint32_t b = 0; b < blocks; ++b) T0 = std::min(T0, st[4 * b]), T1 = std::max(T1, st[4 * b + 1]);
    const double dur = (double)(T1 - T0) * 0.01;  // us
    std::vector<double> starts(blocks), ends(blocks), durs(blocks);
    double busy = 0;
    for (uint32_t b = 0; b < blocks; ++b) {
      starts[b] = (double)(st[4 * b] - T0) * 0.01, ends[b] = (double)(st[4 * b + 1] - T0) * 0.01, durs[b] = ends[b] - starts[b];
      busy += durs[b];
    }
    std::vector<uint32_t> by_start(blocks), by_end(blocks);
    for (uint32_t b = 0; b < blocks; ++b) by_start[b] = by_end[b] = b;
    std::sort(by_start.begin(), by_start.end(), [&](uint32_t x, uint32_t y) { return starts[x] < starts[y]; });
    std::sort(by_end.begin(), by_end.end(), [&](uint32_t x, uint32_t y) { return ends[x] < ends[y]; });
    const uint32_t ns = std::min(kSlotsRunning, blocks);
    double ramp = 0, drain = 0;
    for (uint32_t i = 0; i < ns; ++i) ramp += starts[by_start[i]], drain += dur - ends[by_end[blocks - 1 - i]];
    printf("\nlaunch %d (%s, %u units): %.1f us between HIP events (%.3f of 8 TB/s); first block start -> last block end %.1f us (%.3f); slot-time in blocks %.1f %% of %u slots x that\n", it, label, blocks, ms * 1e3,
           bytes / (ms * 1e-3) / 8e12, dur, bytes / (dur * 1e-6) / 8e12, 100.0 * busy / (ns * dur), ns);
    printf("  ramp : the %u slots' first blocks start %.2f us after the first on average (last of them at %.2f us)\n", ns, ramp / ns, starts[by_start[ns - 1]]);
    printf("  drain: the %u slots' last blocks end %.2f us before the launch's end on average (first of them %.2f us before); the last block STARTED %.2f us before the end\n", ns,
           drain / ns, dur - ends[by_end[blocks - ns]], dur - starts[by_start[blocks - 1]]);
    // durations by position in the launch
    const uint32_t groups = std::max(1u, blocks / ns);
    printf("  block durations by start order, groups of %u (median us | GB/s per block slot):", ns);
    for (uint32_t g = 0; g < groups; ++g) {
      std::vector<double> d;
      for (uint32_t i = g * ns; i < std::min(blocks, (g + 1) * ns); ++i) d.push_back(durs[by_start[i]]);
      const double m = med(d);
      if (groups <= 16 || g < 4 || g + 4 >= groups) printf(" %.1f", m);
      else if (g == 4) printf(" ...");
    }
    printf("\n");
    {
      std::vector<double> all(durs);
      std::sort(all.begin(), all.end());
      printf("  all blocks: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us; %u slots x block bytes / median = %.3f of 8 TB/s\n", all[0], all[blocks / 10], all[blocks / 2],
             all[blocks * 9 / 10], all[blocks - 1], ns, ns * (bytes / blocks) / (all[blocks / 2] * 1e-6) / 8e12);
    }
    // per XCD
    printf("  per XCD: blocks | CUs seen | first start | last end before the launch's end (us) | slot-time in blocks (%% of 64 slots x launch)\n");
    for (uint32_t x = 0; x < 8; ++x) {
      uint32_t n = 0;
      double fs = 1e30, le = 0, bz = 0;
      std::set<uint32_t> cus;
      for (uint32_t b = 0; b < blocks; ++b)
        if ((st[4 * b + 2] & 15u) == x) {
          ++n, fs = std::min(fs, starts[b]), le = std::max(le, ends[b]), bz += durs[b];
          cus.insert((uint32_t)(st[4 * b + 3] >> 8) & 0xFFu);  // CU_ID, SH_ID, SE_ID
        }
      if (n) printf("    xcd %u: %5u | %2zu | %7.2f | %7.2f | %.1f\n", x, n, cus.size(), fs, dur - le, 100.0 * bz / (64.0 * dur));
    }
    // the blocks of a few CUs: (start, end) in us, by start
    {
      printf("  blocks of six CUs (xcd.se.sh.cu: start-end us ...):\n");
      std::set<uint32_t> seen;
      for (uint32_t b = 0; b < blocks && seen.size() < 6; b += 37) {
        const uint32_t key = (uint32_t)((st[4 * b + 2] & 15u) << 16) | ((uint32_t)(st[4 * b + 3] >> 8) & 0xFFu);
        if (!seen.insert(key).second) continue;
        printf("    %u.%u.%u.%u:", key >> 16, (key >> 5) & 7u, (key >> 4) & 1u, key & 15u);
        std::vector<std::pair<double, double>> v;
        for (uint32_t x = 0; x < blocks; ++x)
          if (((uint32_t)((st[4 * x + 2] & 15u) << 16) | ((uint32_t)(st[4 * x + 3] >> 8) & 0xFFu)) == key) v.push_back({starts[x], ends[x]});
        std::sort(v.begin(), v.end());
        for (size_t i = 0; i < v.size() && i < 12; ++i) printf(" %.1f-%.1f", v[i].first, v[i].second);
        printf("\n");
      }
    }
    // bytes per time if every block read its unit at a constant pace (units of equal size only: launches by block id)
    printf("  device rate by 5 %% steps of the launch, every block at its own constant pace (fraction of 8 TB/s):");
    for (int k = 0; k < 20; ++k) {
      const double t0 = dur * k / 20.0, t1 = dur * (k + 1) / 20.0;
      double by = 0;
      for (uint32_t b = 0; b < blocks; ++b) {
        const double ov = std::min(t1, ends[b]) - std::max(t0, starts[b]);
        if (ov > 0 && durs[b] > 0) by += ov / durs[b];
      }
      printf(" %.2f", by * (bytes / blocks) / ((t1 - t0) * 1e-6) / 8e12);
    }
    printf("\n");
    // running blocks over time, 20 bins
    printf("  running blocks at 5 %% steps of the launch:");
    for (int k = 0; k <= 20; ++k) {
      const double t = dur * k / 20.0;
      uint32_t r = 0;
      for (uint32_t b = 0; b < blocks; ++b) r += starts[b] <= t && ends[b] > t;
      printf(" %u", r);
    }
    printf("\n");
  }

// the headline kernel: k_icount_dense<16> over `pairs` row pairs (one block per pair), fused per-node total
static int icount_main(int argc, char** argv) {
  const uint32_t pairs = argc > 2 ? atoi(argv[2]) : 1024;
  const int launches = argc > 3 ? atoi(argv[3]) : 5;
  const size_t rowBytes = 16 * 8192;
  uint8_t *A, *B;
  CK(hipMalloc(&A, (size_t)pairs * rowBytes));
  CK(hipMalloc(&B, (size_t)pairs * rowBytes));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, (size_t)pairs * rowBytes / 8, 1ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)B, (size_t)pairs * rowBytes / 8, 77777777ull);
  std::vector<uint32_t> r(pairs);
  for (uint32_t i = 0; i < pairs; ++i) r[i] = i;
  uint32_t *rows, *done;
  u64 *out, *total, *dstamps;
  CK(hipMalloc(&rows, pairs * 4));
  CK(hipMemcpy(rows, r.data(), pairs * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, pairs * 8));
  CK(hipMalloc(&total, 8));
  CK(hipMalloc(&done, 4));
  CK(hipMemset(done, 0, 4));
  CK(hipMalloc(&dstamps, (size_t)pairs * 32));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(fbk::g_mm_stamps), &dstamps, sizeof(dstamps)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 2.0 * pairs * rowBytes + 8.0 * pairs;
  printf("# k_icount_dense<16>: %u row pairs of dense rows, one block per pair (256 KiB), fused total; %.1f MB per launch\n", pairs, bytes * 1e-6);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fbk::k_icount_dense<16>, dim3(pairs), dim3(256), 0, 0, A, rows, B, rows, out, total, done, pairs, (u64*)nullptr);
  CK(hipDeviceSynchronize());
  std::vector<u64> st_all((size_t)pairs * 4);
  std::vector<double> ev;
  for (int it = 0; it < launches; ++it) {
    CK(hipMemsetAsync(dstamps, 0, (size_t)pairs * 32, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fbk::k_icount_dense<16>, dim3(pairs), dim3(256), 0, 0, A, rows, B, rows, out, total, done, pairs, (u64*)nullptr);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(fbk::k_icount_dense<16>, dim3(pairs), dim3(256), 0, 0, A, rows, B, rows, out, total, done, pairs, (u64*)nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st_all.data(), dstamps, (size_t)pairs * 32, hipMemcpyDeviceToHost));
    ev.push_back(ms * 1e3);
    analyse(st_all, pairs, ms, bytes, it, "k_icount_dense<16>", pairs);
  }
  printf("\n# median %.2f us of %zu launches = %.3f of 8 TB/s\n", med(ev), ev.size(), bytes / (med(ev) * 1e-6) / 8e12);
  {  // 200 launches back to back between one pair of events (the bench's protocol)
    u64* none = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(fbk::g_mm_stamps), &none, sizeof(none)));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(fbk::k_icount_dense<16>, dim3(pairs), dim3(256), 0, 0, A, rows, B, rows, out, total, done, pairs, (u64*)nullptr);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("# 200 launches back to back: %.2f us per launch = %.3f of 8 TB/s\n", ms * 1e3 / 200, bytes / (ms * 1e-3 / 200) / 8e12);
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  const uint32_t shards = argc > 1 ? atoi(argv[1]) : 1024, spb = argc > 2 ? atoi(argv[2]) : 4;
  const int launches = argc > 3 ? atoi(argv[3]) : 5;
  const int tickets = argc > 4 ? atoi(argv[4]) : 0;
  if (argc > 1 && !strcmp(argv[1], "icount")) return icount_main(argc, argv);
  const uint32_t nA = 32, nB = 32;
  const size_t rowBytes = 16 * 8192;
  uint8_t *A, *B, *F;
  CK(hipMalloc(&A, (size_t)shards * nA * rowBytes));
  CK(hipMalloc(&B, (size_t)shards * nB * rowBytes));
  CK(hipMalloc(&F, (size_t)shards * rowBytes));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, (size_t)shards * nA * rowBytes / 8, 1ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)B, (size_t)shards * nB * rowBytes / 8, 77777777ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)F, (size_t)shards * rowBytes / 8, 999999999ull);
  std::vector<uint32_t> ra((size_t)shards * nA), rb((size_t)shards * nB), rf(shards);
  for (size_t i = 0; i < ra.size(); ++i) ra[i] = (uint32_t)i;
  for (size_t i = 0; i < rb.size(); ++i) rb[i] = (uint32_t)i;
  for (size_t i = 0; i < rf.size(); ++i) rf[i] = (uint32_t)i;
  uint32_t *rowsA, *rowsB, *rowsF;
  CK(hipMalloc(&rowsA, ra.size() * 4));
  CK(hipMalloc(&rowsB, rb.size() * 4));
  CK(hipMalloc(&rowsF, rf.size() * 4));
  CK(hipMemcpy(rowsA, ra.data(), ra.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(rowsB, rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(rowsF, rf.data(), rf.size() * 4, hipMemcpyHostToDevice));
  u64* out;
  const size_t outBytes = (size_t)shards * nA * nB * 8;
  CK(hipMalloc(&out, outBytes));
  const uint32_t blocks_static = shards * (16 / spb);
  fbk::MmTickets tk{0, 0, 0, 0};
  bool have_plan = fbk::mm_ticket_plan(shards, spb, 512, tk);
  if (argc > 8) {  // explicit tiers: spb1 shards1 spb2 shards2
    const uint32_t spb1 = atoi(argv[5]), s1 = atoi(argv[6]), spb2 = atoi(argv[7]), s2 = atoi(argv[8]);
    tk.tier0_shards = shards - s1 - s2, tk.tier1_shards = s1, tk.tier_spb = spb1 | (spb2 << 8);
    const uint64_t units = uint64_t(tk.tier0_shards) * (16 / spb) + uint64_t(s1) * (16 / spb1) + uint64_t(s2) * (16 / spb2);
    tk.grid = uint32_t((units + units / 8 + 64 + 7) & ~7ull);
    have_plan = true;
  }
  if (tickets && !have_plan) {
    printf("no ticket plan for %u shards at %u slots per block\n", shards, spb);
    return 1;
  }
  uint32_t* dticket;
  CK(hipMalloc(&dticket, 256));
  CK(hipMemset(dticket, 0, 256));
  uint32_t blocks = blocks_static;
  bool by_ticket = false;
  u64* dstamps;
  CK(hipMalloc(&dstamps, (size_t)std::max(blocks_static, tk.grid) * 32));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(fbk::g_mm_stamps), &dstamps, sizeof(dstamps)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&] {
    if (by_ticket)
      hipLaunchKernelGGL((fbk::k_count_matrix_mfma<true, 4, 2, 2, 1, 1>), dim3(tk.grid), dim3(256), 0, 0, A, rowsA, nA, B, rowsB, nB, F, rowsF, shards, spb, out, dticket,
                         tk.tier0_shards, tk.tier1_shards, tk.tier_spb);
    else
      hipLaunchKernelGGL((fbk::k_count_matrix_mfma<true, 4, 2, 2, 1, 1>), dim3(blocks_static), dim3(256), 0, 0, A, rowsA, nA, B, rowsB, nB, F, rowsF, shards, spb, out);
  };
  const double bytes = (double)shards * (nA + nB + 1) * 16 * 8192;
  printf("# k_count_matrix_mfma<F, 4 waves, depth 2, nt, 1x1>: %u shards x (32 + 32 + 1) dense rows, %u slots per block = %u blocks of %.2f MB; %.1f MB per launch\n", shards, spb,
         blocks_static, bytes / blocks_static * 1e-6, bytes * 1e-6);
  if (tickets)
    printf("# by ticket: %u shards in units of %u slots, %u in units of %u, %u in units of %u; grid %u\n", tk.tier0_shards, spb, tk.tier1_shards, tk.tier_spb & 255u,
           shards - tk.tier0_shards - tk.tier1_shards, (tk.tier_spb >> 8) & 255u, tk.grid);
  std::vector<u64> ref;
  for (int i = 0; i < 6; ++i) {
    by_ticket = tickets == 1 || (tickets == 2 && (i & 1));
    CK(hipMemsetAsync(out, 0, outBytes, 0));
    launch();
    if (tickets == 2 && i < 2) {  // both forms give the same matrices
      std::vector<u64> got(outBytes / 8);
      CK(hipMemcpy(got.data(), out, outBytes, hipMemcpyDeviceToHost));
      if (i == 0) ref = got;
      else printf("# by ticket == by block id: %s\n", got == ref ? "yes" : "NO");
    }
  }
  CK(hipDeviceSynchronize());
  std::vector<u64> st_all((size_t)std::max(blocks_static, tk.grid) * 4);
  const uint32_t kSlotsRunning = 512;
  std::vector<double> ev_us[2];
  for (int it = 0; it < launches; ++it) {
    by_ticket = tickets == 1 || (tickets == 2 && (it & 1));
    const uint32_t grid = by_ticket ? tk.grid : blocks_static;
    CK(hipMemsetAsync(out, 0, outBytes, 0));
    CK(hipMemsetAsync(dstamps, 0, (size_t)grid * 32, 0));
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st_all.data(), dstamps, (size_t)grid * 32, hipMemcpyDeviceToHost));
    ev_us[by_ticket ? 1 : 0].push_back(ms * 1e3);
    analyse(st_all, grid, ms, bytes, it, by_ticket ? "by ticket" : "by block id", kSlotsRunning);
  }
  for (int m = 0; m < 2; ++m)
    if (!ev_us[m].empty()) printf("\n# %s: median %.1f us of %zu launches = %.3f of 8 TB/s\n", m ? "by ticket" : "by block id", med(ev_us[m]), ev_us[m].size(), bytes / (med(ev_us[m]) * 1e-6) / 8e12);
  return 0;
}
