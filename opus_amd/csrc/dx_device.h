// Device-side helpers: wave64 scans, block scans, the term hash.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dx {

constexpr int kWave = 64;

// ---- term hash ------------------------------------------------------------------------------
// A 64-bit state of two 32-bit halves fed with the term's bytes four at a time (little-endian words
// counted from the term's first byte; a final partial word is zero-padded): a murmur-style step on one
// half, folded into the other, the halves swapping roles every word; finalised together with the length
// into one 64-bit word. The same code runs on the host (query constants, class-index
// entries) and on the device (tokeniser, lookups). A hash is only a bucket choice: every dictionary
// hit is confirmed by a full byte compare, and a true 64-bit collision between different strings
// makes the build restart with another seed.
struct TermHash {
  uint32_t a, b;
  __host__ __device__ explicit TermHash(uint64_t seed)
      : a(0x9747B28Cu ^ (uint32_t)seed), b(0x85EBCA6Bu ^ (uint32_t)(seed >> 32)) {}
  static __host__ __device__ inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
  __host__ __device__ inline void add_word(uint32_t k) {
    // One multiply per word (v_mul_lo_u32 is quarter rate); x * 5 + c is a shift-add. The step is a
    // bijection of the 64-bit state for a fixed word and of the word for a fixed state, so two strings that
    // differ in one word never meet again, and the word reaches both halves of the state at once.
    uint32_t x = rotl(a ^ (k * 0xCC9E2D51u), 15);
    x = (x << 2) + x + 0xE6546B64u;
    a = b ^ x;
    b = x;
  }
  __host__ __device__ inline uint64_t finish(uint32_t len) const {
    uint64_t h = ((uint64_t)a << 32) | (uint64_t)(b ^ len);
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 33;
    return h ? h : 1;  // 0 marks an empty dictionary slot
  }
};

// Hash of n bytes that can only be read one at a time (host strings, the device's slow paths).
__host__ __device__ inline uint64_t hash_bytes(const uint8_t* p, uint64_t n, uint64_t seed) {
  TermHash h(seed);
  uint64_t i = 0;
  for (; i + 4 <= n; i += 4)
    h.add_word((uint32_t)p[i] | ((uint32_t)p[i + 1] << 8) | ((uint32_t)p[i + 2] << 16) | ((uint32_t)p[i + 3] << 24));
  if (i < n) {
    uint32_t k = 0;
    for (int s = 0; i < n; ++i, s += 8) k |= (uint32_t)p[i] << s;
    h.add_word(k);
  }
  return h.finish((uint32_t)n);
}

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
