// atomic_rate.hip — what does a ticket cost?  (not part of the product)
// `blocks` blocks of 256 threads, thread 0 of each takes `k` tickets from ONE counter (device-scope atomicAdd with return, each waited for
// before the next is issued — a block cannot know its next unit before it has the ticket), or from one of `nctr` counters (block id mod nctr).
// Reports the launch time, the time per ticket seen by one block (latency under that load) and the device-wide ticket rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/atomic_rate.hip -o scripts/atomic_rate && scripts/atomic_rate > profiles/r06_atomic_rate.txt
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                               \
    }                                                        \
  } while (0)

__global__ void __launch_bounds__(256) k_tickets(uint32_t* ctr, uint32_t nctr, uint32_t k, uint32_t* sink) {
  if (threadIdx.x) return;
  uint32_t* c = ctr + 64u * (blockIdx.x % nctr);  // counters 256 bytes apart
  uint32_t acc = 0;
  for (uint32_t i = 0; i < k; ++i) acc += atomicAdd(c, 1u);
  if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main() {
  uint32_t *ctr, *sink;
  CK(hipMalloc(&ctr, 64 * 256));
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("# blocks  counters  tickets/block   launch us   us per ticket (one block)   tickets per us (device)\n");
  for (uint32_t blocks : {1u, 256u, 1024u, 4096u})
    for (uint32_t nctr : {1u, 8u, 64u})
      for (uint32_t k : {1u, 4u, 16u}) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
          CK(hipMemsetAsync(ctr, 0, 64 * 256, 0));
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(k_tickets, dim3(blocks), dim3(256), 0, 0, ctr, nctr, k, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep && ms < best) best = ms;
        }
        printf("%7u %8u %10u %14.2f %18.3f %24.1f\n", blocks, nctr, k, best * 1e3, best * 1e3 / k, (double)blocks * k / (best * 1e3));
      }
  return 0;
}
