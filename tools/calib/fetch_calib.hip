// fetch_calib.hip — what does rocprofv3's FETCH_SIZE report for THIS path's access patterns?
// (MI355X_MICROARCH.md §HBM: calibrated only for 16 B/lane streaming reads, where it reports half
// the bytes.)  Every kernel touches every STRIDE-byte slot of a 2 GiB buffer exactly once — far
// beyond the 32 MB of L2 and the 256 MB Infinity Cache — so the distinct bytes are known:
//   stream16            : lane-contiguous 16-byte loads (the guide's reference pattern)
//   gather<BYTES,STRIDE>: one BYTES-wide load per STRIDE-byte slot, slots visited in a
//                         multiplicative-hash order (the 64 lanes of a wave hit 64 unrelated lines):
//                         the bitmap/rank probes (8 B), fieldnorm bytes (1 B), tf words (4 B) and
//                         block records (16 B) of the scan kernels.
// Comparing gather<8,128> (one access per 128-byte line) with gather<8,64> (two accesses per line,
// issued by unrelated waves) tells whether a miss moves 64 or 128 bytes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o tools/calib/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void stream16(const uint4 *buf, uint64_t n16, uint32_t *sink) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n16; i += step) {
    const uint4 v = buf[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int BYTES, int STRIDE>
__global__ void gather(const uint8_t *buf, uint64_t n_slots, uint32_t *sink) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n_slots; i += step) {
    const uint64_t slot = (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (n_slots - 1);  // bijection (n_slots = 2^m)
    const uint8_t *p = buf + slot * (uint64_t)STRIDE;
    if (BYTES == 1) acc += *p;
    if (BYTES == 4) acc += *reinterpret_cast<const uint32_t *>(p);
    if (BYTES == 8) {
      const uint2 v = *reinterpret_cast<const uint2 *>(p);
      acc += v.x ^ v.y;
    }
    if (BYTES == 16) {
      const uint4 v = *reinterpret_cast<const uint4 *>(p);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int BYTES, int STRIDE>
static void run_gather(const uint8_t *buf, uint64_t bytes, uint32_t *sink, const char *name) {
  const uint64_t n_slots = bytes / STRIDE;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  CHECK(hipEventRecord(a));
  gather<BYTES, STRIDE><<<256 * 32, 256>>>(buf, n_slots, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("{\"kernel\": \"gather<%d, %d>\", \"label\": \"%s\", \"accesses\": %llu, \"useful_bytes\": %llu, "
         "\"span_bytes\": %llu, \"ms\": %.3f}\n",
         BYTES, STRIDE, name, (unsigned long long)n_slots, (unsigned long long)(n_slots * BYTES),
         (unsigned long long)bytes, ms);
}

int main() {
  const uint64_t bytes = 2ull << 30;
  uint8_t *buf;
  uint32_t *sink;
  CHECK(hipMalloc((void **)&buf, bytes + 64));
  CHECK(hipMalloc((void **)&sink, 4));
  CHECK(hipMemset(buf, 1, bytes + 64));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    stream16<<<256 * 32, 256>>>((const uint4 *)buf, bytes / 16, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("{\"kernel\": \"stream16\", \"label\": \"16 B/lane coalesced stream\", \"accesses\": %llu, "
           "\"useful_bytes\": %llu, \"span_bytes\": %llu, \"ms\": %.3f}\n",
           (unsigned long long)(bytes / 16), (unsigned long long)bytes, (unsigned long long)bytes, ms);
    run_gather<8, 128>(buf, bytes, sink, "8 B probe, one per 128-B line");
    run_gather<8, 64>(buf, bytes, sink, "8 B probe, one per 64-B half line");
    run_gather<8, 32>(buf, bytes, sink, "8 B probe, one per 32-B sector");
    run_gather<8, 8>(buf, bytes / 4, sink, "8 B probe, every word of the span, hashed order");
    run_gather<1, 128>(buf, bytes, sink, "1 B fieldnorm gather, one per 128-B line");
    run_gather<1, 64>(buf, bytes, sink, "1 B fieldnorm gather, one per 64-B half line");
    run_gather<4, 128>(buf, bytes, sink, "4 B tf word, one per 128-B line");
    run_gather<16, 128>(buf, bytes, sink, "16 B block record, one per 128-B line");
    run_gather<16, 64>(buf, bytes, sink, "16 B block record, one per 64-B half line");
  }
  return 0;
}
