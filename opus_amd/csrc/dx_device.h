// Device-side helpers: wave64 scans, block scans, the term hash.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dx {

constexpr int kWave = 64;

// ---- term hash ------------------------------------------------------------------------------
// Two 32-bit multiplicative streams over the bytes, finalised into one 64-bit word. The same
// code runs on the host (query constants, class-index entries) and on the device (tokeniser).
// A hash is only a bucket choice: every dictionary hit is confirmed by a full byte compare, and a
// true 64-bit collision between different strings makes the build restart with another seed.
struct TermHash {
  uint32_t a, b;
  __host__ __device__ explicit TermHash(uint64_t seed)
      : a(0x811C9DC5u ^ (uint32_t)seed), b(0x9747B28Cu ^ (uint32_t)(seed >> 32)) {}
  __host__ __device__ void add(uint8_t c) {
    a = (a ^ c) * 0x01000193u;
    b = (b ^ c) * 0x5BD1E995u;
    b ^= b >> 15;
  }
  __host__ __device__ uint64_t finish(uint32_t len) const {
    uint64_t h = ((uint64_t)a << 32) | (uint64_t)(b ^ len);
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 33;
    return h ? h : 1;  // 0 marks an empty dictionary slot
  }
};

__host__ __device__ inline uint64_t mix64(uint64_t h) {
  h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull;
  h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull;
  h ^= h >> 33;
  return h;
}

#ifdef __HIPCC__
__device__ inline int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ inline uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    uint32_t t = __shfl_up(v, d, kWave);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ inline uint64_t wave_incl_scan_u64(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    uint64_t t = __shfl_up(v, d, kWave);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ inline uint32_t wave_reduce_u32(uint32_t v) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
  return v;
}
__device__ inline uint64_t wave_reduce_u64(uint64_t v) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
  return v;
}

// Exclusive scan across a 256-thread block (4 waves); returns the exclusive prefix of `v` and the
// block total in `total`. `lds` must hold 8 T's. Contains two __syncthreads().
template <typename T>
__device__ inline T block_excl_scan_256(T v, T* lds, T& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  T incl;
  if constexpr (sizeof(T) == 8) incl = (T)wave_incl_scan_u64((uint64_t)v, lane);
  else incl = (T)wave_incl_scan_u32((uint32_t)v, lane);
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T x = lds[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  total = tot;
  return base + incl - v;
}
#endif

}  // namespace dx
