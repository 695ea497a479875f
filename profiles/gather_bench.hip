// How many scattered sector accesses per second does this GPU serve? The dictionary stage of the partitioner (slot reads,
// atomics, reads of a representative's bytes at any place in the text) and the probes of the query path's joins are
// bound by that rate, not by bandwidth. Every lane reads 16 bytes at pseudo-random 64-byte-aligned places of a buffer,
// `kInFlight` independent loads per round; and the same with an atomic minimum on 8 bytes.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t h) {
  h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
  return h;
}

template <int kInFlight>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ buf, uint64_t sectors /*power of two*/, int rounds, uint64_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t acc = 0;
  for (int r = 0; r < rounds; ++r) {
    uint4 v[kInFlight];
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) v[j] = buf[(mix(t * 1315423911ull + (uint64_t)r * kInFlight + j) & (sectors - 1)) * 4];
#pragma unroll
    for (int j = 0; j < kInFlight; ++j) acc += v[j].x ^ v[j].w;
  }
  if (acc == 0x1234567) out[0] = acc;
}
// a dependent chain: the place of the next read comes out of the one before (what a probe chain or a linked list does)
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ buf, uint64_t sectors, int rounds, uint64_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t at = mix(t) & (sectors - 1), acc = 0;
  for (int r = 0; r < rounds; ++r) {
    const uint4 v = buf[at * 4];
    acc += v.x;
    at = mix(at + v.y + r) & (sectors - 1);
  }
  if (acc == 0x1234567) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_atomic_min(unsigned long long* __restrict__ buf, uint64_t sectors, int rounds) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (int r = 0; r < rounds; ++r) atomicMin(&buf[(mix(t * 2654435761ull + r) & (sectors - 1)) * 8], (unsigned long long)(t + r));
}

int main() {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint64_t* out;
  CK(hipMalloc(&out, 8));
  for (uint64_t mib : {32ull, 128ull, 512ull, 2048ull, 8192ull}) {
    const uint64_t bytes = mib << 20, sectors = bytes / 64;
    void* buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    const int blocks = 256 * 8, rounds = 64;
    auto run = [&](const char* what, auto launch, double per_thread) {
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double n = (double)blocks * 256 * per_thread;
      printf("%5llu MiB  %-28s %8.3f ms  %7.1f G accesses/s  %6.2f TB/s of 64-byte sectors\n", (unsigned long long)mib, what, ms, n / ms / 1e6, n * 64 / ms / 1e9);
    };
    run("reads, 1 in flight per lane", [&] { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, sectors, rounds * 8, out); }, rounds * 8.0);
    run("reads, 4 in flight per lane", [&] { hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, sectors, rounds * 2, out); }, rounds * 8.0);
    run("reads, 8 in flight per lane", [&] { hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, sectors, rounds, out); }, rounds * 8.0);
    run("reads, dependent chain", [&] { hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, sectors, rounds * 2, out); }, rounds * 2.0);
    run("atomicMin on 8 bytes", [&] { hipLaunchKernelGGL(k_atomic_min, dim3(blocks), dim3(256), 0, 0, (unsigned long long*)buf, sectors, rounds); }, rounds * 1.0);
    CK(hipFree(buf));
  }
  return 0;
}
