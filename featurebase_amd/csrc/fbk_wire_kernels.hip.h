// fbk_wire_kernels.hip.h — device side of the roaring serialisation formats.
// The serialised image (Pilosa format, roaring.go:1730-1817, or the official RoaringBitmap
// format, roaring.go:6948-7006) crosses PCIe as ONE blob; this kernel then moves every
// container payload between the blob and the 16-byte aligned arena in parallel, one
// wavefront per container.  In the blob a payload is only 2-byte aligned (arrays are 2N
// bytes, runs carry a 2-byte count prefix, roaring.go:4100-4107) — in the official format
// with a run bitmap even byte aligned — so the copy granularity is chosen per container.
#pragma once
#include "fbk_kernels.hip.h"

namespace fbk {

struct alignas(8) WireDesc {
  uint64_t src;    // byte offset of the payload in the source buffer
  uint64_t dst;    // byte offset of the payload in the destination buffer
  uint32_t bytes;  // payload bytes
  uint32_t mode;   // 0 copy; 1 copy + official-format run conversion {start, len-1} -> {start, last}
                   // (roaring.go:2239-2247); 2 copy + write the u16 run count at dst-2 (:4100)
};

__global__ void __launch_bounds__(256) k_wire_copy(const uint8_t* __restrict__ src_base, uint8_t* __restrict__ dst_base,
                                                  const WireDesc* __restrict__ descs, uint64_t n) {
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const WireDesc d = descs[i];
  const uint8_t* s = src_base + d.src;
  uint8_t* o = dst_base + d.dst;
  const uint32_t bytes = d.bytes;
  if (d.mode == 2 && lane == 0) {
    const uint32_t runs = bytes >> 2;
    o[-2] = (uint8_t)runs;
    o[-1] = (uint8_t)(runs >> 8);
  }
  const uint64_t al = (reinterpret_cast<uint64_t>(s) | reinterpret_cast<uint64_t>(o));
  if (d.mode == 1) {  // 4-byte records {start, len-1}
    const uint32_t nrec = bytes >> 2;
    if ((al & 3) == 0) {
      const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
      uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
      for (uint32_t k = lane; k < nrec; k += kWave) {
        const uint32_t v = s4[k];
        const uint32_t st = v & 0xFFFFu;
        o4[k] = st | (((st + (v >> 16)) & 0xFFFFu) << 16);
      }
    } else {
      for (uint32_t k = lane; k < nrec; k += kWave) {
        const uint32_t st = s[4 * k] | (s[4 * k + 1] << 8), ln = s[4 * k + 2] | (s[4 * k + 3] << 8);
        const uint32_t la = (st + ln) & 0xFFFFu;
        o[4 * k] = (uint8_t)st;
        o[4 * k + 1] = (uint8_t)(st >> 8);
        o[4 * k + 2] = (uint8_t)la;
        o[4 * k + 3] = (uint8_t)(la >> 8);
      }
    }
    return;
  }
  if ((al & 15) == 0 && (bytes & 15) == 0) {
    const ulonglong2* s16 = reinterpret_cast<const ulonglong2*>(s);
    ulonglong2* o16 = reinterpret_cast<ulonglong2*>(o);
    for (uint32_t k = lane; k < (bytes >> 4); k += kWave) o16[k] = s16[k];
  } else if ((al & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
    uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
    for (uint32_t k = lane; k < (bytes >> 2); k += kWave) o4[k] = s4[k];
    for (uint32_t k = (bytes & ~3u) + lane; k < bytes; k += kWave) o[k] = s[k];
  } else if ((al & 1) == 0) {
    const uint16_t* s2 = reinterpret_cast<const uint16_t*>(s);
    uint16_t* o2 = reinterpret_cast<uint16_t*>(o);
    for (uint32_t k = lane; k < (bytes >> 1); k += kWave) o2[k] = s2[k];
    if ((bytes & 1) && lane == 0) o[bytes - 1] = s[bytes - 1];
  } else {
    for (uint32_t k = lane; k < bytes; k += kWave) o[k] = s[k];
  }
}

}  // namespace fbk
