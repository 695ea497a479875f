// Times the library's radix sorts on key sets shaped like the partitioner's at LUBM-100 (scratch tool, not shipped).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -c profiles/sort_bench.hip -o /tmp/sort_bench.o && hipcc --offload-arch=gfx950 /tmp/sort_bench.o opus_amd/_lib/dx_prims.o -o /tmp/sort_bench
#include <algorithm>
#include <cstdio>
#include <vector>

#include "../opus_amd/csrc/dx_internal.h"

using namespace dx;

static uint64_t mix(uint64_t h) {
  h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; return h ^ (h >> 33);
}

template <typename K, typename F>
static void run(const char* name, DevPool& pool, std::vector<K>& h, int bits, F sort) {
  hipStream_t s = nullptr;
  const size_t n = h.size();
  DevBuf<K> master(pool, n), work(pool, n);
  DX_HIP(hipMemcpy(master.p, h.data(), n * sizeof(K), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  DX_HIP(hipEventCreate(&e0)); DX_HIP(hipEventCreate(&e1));
  float best = 1e9, sum = 0;
  const int reps = 12;
  for (int r = 0; r < reps; ++r) {
    DX_HIP(hipMemcpyAsync(work.p, master.p, n * sizeof(K), hipMemcpyDeviceToDevice, s));
    DX_HIP(hipEventRecord(e0, s));
    sort(work, n, bits, s);
    DX_HIP(hipEventRecord(e1, s));
    DX_HIP(hipStreamSynchronize(s));
    float ms = 0;
    DX_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) { best = std::min(best, ms); sum += ms; }
  }
  std::vector<K> out(n);
  DX_HIP(hipMemcpy(out.data(), work.p, n * sizeof(K), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const bool ok = out == h;
  printf("%-28s n=%zu bits=%d best %.3f ms avg %.3f ms %s\n", name, n, bits, best, sum / (reps - 2), ok ? "sorted-ok" : "WRONG");
  fflush(stdout);
}

int main() {
  DevPool pool;
  {
    const size_t n = 13064756;
    std::vector<uint64_t> h(n);
    for (size_t i = 0; i < n; ++i) {
      const uint64_t s = (i / 6 * 4 + mix(i) % 3) % 3300000, p = mix(i * 7 + 1) % 18, o = mix(i * 13 + 5) % 3300000;
      h[i] = (p << 44) | (s << 22) | o;
    }
    run<uint64_t>("u64 49 bits (triples)", pool, h, 49, [&](DevBuf<uint64_t>& k, size_t n, int b, hipStream_t s) { radix_sort_u64(pool, k, n, b, s); });
  }
  {
    const size_t n = 23000000;
    std::vector<uint32_t> h(n);
    for (size_t i = 0; i < n; ++i) {
      const uint32_t part = (uint32_t)(i * 8 / n), pred = (uint32_t)((i * 8 * 18 / n) % 18), term = (uint32_t)(mix(i) % 3300000);
      h[i] = (part << 28) | (term << 6) | (1u << 5) | pred;
    }
    run<uint32_t>("u32 31 bits (VP index)", pool, h, 31, [&](DevBuf<uint32_t>& k, size_t n, int b, hipStream_t s) { radix_sort_u32(pool, k, n, b, s); });
  }
  {
    const size_t n = 4000000;
    std::vector<u128> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = ((u128)(mix(i) % 40000) << 56) | ((u128)(mix(i + 77) & 0xFFFFFFF) << 28) | (mix(i * 3) & 0xFFFFFFF);
    run<u128>("u128 72 bits", pool, h, 72, [&](DevBuf<u128>& k, size_t n, int b, hipStream_t s) { radix_sort_u128(pool, k, n, b, s); });
  }
  {
    std::vector<uint64_t> h(5000);
    for (size_t i = 0; i < h.size(); ++i) h[i] = mix(i) & 0xFFFFFF;
    run<uint64_t>("u64 24 bits small", pool, h, 24, [&](DevBuf<uint64_t>& k, size_t n, int b, hipStream_t s) { radix_sort_u64(pool, k, n, b, s); });
  }
  return 0;
}
