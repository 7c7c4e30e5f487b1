// FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the scoring kernels (MI355X_MICROARCH.md, HBM:
// "other access widths ... uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Every kernel below moves a KNOWN number of useful bytes over an array far larger than L2 + Infinity
// Cache (8 GiB against 32 MiB + 256 MiB), each in its own launch, so rocprofv3 --pmc FETCH_SIZE (or
// WRITE_SIZE, TCC_*) reports one value per pattern.  tools/fetch_size_calibration.sh runs the passes and
// divides.  The program itself prints, per kernel, the useful bytes, the distinct 64-byte and 128-byte
// lines the pattern touches (exact for the streams, expectation for the random patterns) and the HIP-event
// time, so the counter can be put against (a) useful bytes, (b) 64-B sectors, (c) 128-B lines.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_size_calibration.hip -o gpurun_out/fetch_cal
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64 finalizer
  x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// wide coalesced streams: W bytes per lane, lanes adjacent
template <typename T>
__global__ void cal_stream(const T* __restrict__ a, size_t n, uint64_t* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < n; i += stride) {
    T v = a[i];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += w[k];
  }
  if (acc == 0x1234567u) *sink = acc;
}

// scattered loads of W bytes: every lane its own pseudo-random element (the K1d lookups: one 8-byte
// bitmap cell / dense-row value / Bloom word per surviving posting)
template <typename T>
__global__ void cal_random(const T* __restrict__ a, size_t n_elems, size_t n_loads, uint64_t seed, uint64_t* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < n_loads; i += stride) {
    T v = a[mix(i ^ seed) % n_elems];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += w[k];
  }
  if (acc == 0x1234567u) *sink = acc;
}

// scattered 8-byte loads in ascending order per wave (lanes of a wave ask for increasing addresses about
// `gap` elements apart: a doc-ordered list looking up a row - our dominant pattern): neighbours may share
// a 64/128-byte line
__global__ void cal_sorted8(const uint64_t* __restrict__ a, size_t n_elems, size_t n_loads, unsigned gap, uint64_t* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < n_loads; i += stride) {
    // position = i * gap + jitter in [0, gap): ascending, one element per `gap`-element window
    size_t p = (i * (size_t)gap + (mix(i) % gap)) % n_elems;
    acc += a[p];
  }
  if (acc == 0x1234567u) *sink = acc;
}

// (reads a region once so that it is cache resident; not one of the calibrated patterns: its name lacks the cal_ prefix)
__global__ void warm_region(const uint4* __restrict__ a, size_t n, uint64_t* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < n; i += stride) acc += a[i].x;
  if (acc == 0x1234567u) *sink = acc;
}

template <typename T>
__global__ void cal_stream_write(T* __restrict__ a, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  T v; uint32_t* w = reinterpret_cast<uint32_t*>(&v);
  for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = (uint32_t)i + k;
  for (; i < n; i += stride) a[i] = v;
}

__global__ void cal_random_write8(uint64_t* __restrict__ a, size_t n_elems, size_t n_stores, uint64_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n_stores; i += stride) a[mix(i ^ seed) % n_elems] = i;
}

struct Ev { hipEvent_t a, b; Ev() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); } };

static double expect_distinct(double lines, double draws) { return lines * (1.0 - std::exp(-draws / lines)); }

int main(int argc, char** argv) {
  size_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8;
  size_t bytes = gib << 30;
  size_t n_loads = argc > 2 ? strtoull(argv[2], nullptr, 10) : (32u << 20);
  uint4* a; uint64_t* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(sink, 0, 8));
  CK(hipDeviceSynchronize());
  Ev ev;
  const int block = 256, grid = 256 * 32;
  auto timed = [&](const char* name, double useful, double l64, double l128, auto&& launch) {
    // a flush pass between the patterns so that nothing of the previous one is left in L2 / Infinity Cache
    CK(hipMemsetAsync((char*)a + bytes / 2, 1, 1ull << 30, 0));
    CK(hipEventRecord(ev.a, 0)); launch(); CK(hipEventRecord(ev.b, 0)); CK(hipEventSynchronize(ev.b));
    float ms; CK(hipEventElapsedTime(&ms, ev.a, ev.b));
    printf("{\"kernel\": \"%s\", \"useful_bytes\": %.0f, \"bytes_as_64B_lines\": %.0f, \"bytes_as_128B_lines\": %.0f, \"ms\": %.4f, \"useful_GBps\": %.1f}\n",
           name, useful, l64 * 64, l128 * 128, ms, useful / ms / 1e6);
  };
  double L64 = bytes / 64.0, L128 = bytes / 128.0;
  timed("cal_stream<uint4>", bytes, L64, L128, [&] { cal_stream<uint4><<<grid, block>>>(a, bytes / 16, sink); });
  timed("cal_stream<uint2>", bytes, L64, L128, [&] { cal_stream<uint2><<<grid, block>>>((uint2*)a, bytes / 8, sink); });
  timed("cal_stream<uint>", bytes, L64, L128, [&] { cal_stream<uint32_t><<<grid, block>>>((uint32_t*)a, bytes / 4, sink); });
  timed("cal_random<uint2>", n_loads * 8.0, expect_distinct(L64, n_loads), expect_distinct(L128, n_loads),
        [&] { cal_random<uint2><<<grid, block>>>((uint2*)a, bytes / 8, n_loads, 0x51ull, sink); });
  timed("cal_random<uint>", n_loads * 4.0, expect_distinct(L64, n_loads), expect_distinct(L128, n_loads),
        [&] { cal_random<uint32_t><<<grid, block>>>((uint32_t*)a, bytes / 4, n_loads, 0x52ull, sink); });
  timed("cal_random<uint4>", n_loads * 16.0, expect_distinct(L64, n_loads), expect_distinct(L128, n_loads),
        [&] { cal_random<uint4><<<grid, block>>>(a, bytes / 16, n_loads, 0x53ull, sink); });
  for (unsigned gap : {4u, 16u, 64u}) {  // one 8-byte element per 32 / 128 / 512 bytes, ascending
    char name[64]; snprintf(name, sizeof name, "cal_sorted8 gap=%u", gap);
    size_t n = n_loads; if (n * gap > bytes / 8) n = bytes / 8 / gap;
    double span = (double)n * gap * 8;
    double l64 = gap >= 8 ? (double)n : span / 64, l128 = gap >= 16 ? (double)n : span / 128;
    timed(name, n * 8.0, l64, l128, [&] { cal_sorted8<<<grid, block>>>((uint64_t*)a, bytes / 8, n, gap, sink); });
  }
  // the same scattered 8-byte loads confined to regions that fit the Infinity Cache (256 MiB) / one XCD's L2 (4 MiB):
  // does a request that hits there cost a fabric request slot like one that goes to HBM?  (each region is read once
  // first, so it is resident; the flush between patterns is skipped for these)
  for (size_t mib : {128, 32, 2}) {
    const size_t sub = mib << 20;
    char name[64]; snprintf(name, sizeof name, "cal_random<uint2> in %zu MiB", mib);
    warm_region<<<grid, block>>>(a, sub / 16, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev.a, 0));
    cal_random<uint2><<<grid, block>>>((uint2*)a, sub / 8, n_loads, 0x61ull + mib, sink);
    CK(hipEventRecord(ev.b, 0)); CK(hipEventSynchronize(ev.b));
    float ms; CK(hipEventElapsedTime(&ms, ev.a, ev.b));
    printf("{\"kernel\": \"%s\", \"useful_bytes\": %.0f, \"bytes_as_64B_lines\": %.0f, \"bytes_as_128B_lines\": %.0f, \"ms\": %.4f, \"useful_GBps\": %.1f}\n",
           name, n_loads * 8.0, expect_distinct(sub / 64.0, n_loads) * 64, expect_distinct(sub / 128.0, n_loads) * 128, ms, n_loads * 8.0 / ms / 1e6);
  }
  timed("cal_stream_write<uint4>", bytes / 4.0, L64 / 4, L128 / 4, [&] { cal_stream_write<uint4><<<grid, block>>>(a, bytes / 64); });
  timed("cal_random_write8", n_loads * 8.0, expect_distinct(L64, n_loads), expect_distinct(L128, n_loads),
        [&] { cal_random_write8<<<grid, block>>>((uint64_t*)a, bytes / 8, n_loads, 0x54ull); });
  CK(hipDeviceSynchronize());
  return 0;
}
