// Internal declarations shared by the translation units of libdiaeresis_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/diaeresis_hip.h"

namespace dx {

struct HipError : std::runtime_error {
  hipError_t code;
  HipError(hipError_t c, const char* what, const char* file, int line)
      : std::runtime_error(std::string(what) + ": " + hipGetErrorString(c) + " at " + file + ":" + std::to_string(line)),
        code(c) {}
};
struct DxError : std::runtime_error {
  int code;
  DxError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define DX_HIP(expr)                                                  \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) throw ::dx::HipError(_e, #expr, __FILE__, __LINE__); \
  } while (0)

// Caching device allocator: a partition or query run asks for the same sizes every time it is
// repeated, so freed blocks are kept and handed out again instead of going back to hipMalloc.
class DevPool {
 public:
  DevPool() = default;
  // Fixed arena: hands out pieces of [base, base + bytes) one after another and never calls the HIP allocator;
  // a request that does not fit throws DX_E_RANGE. release() is a no-op, rewind() starts from the base again.
  DevPool(void* base, size_t bytes) : arena_base_(static_cast<uint8_t*>(base)), arena_bytes_(bytes) {}
  DevPool(const DevPool&) = delete;
  DevPool& operator=(const DevPool&) = delete;
  ~DevPool() { trim(); }
  bool is_arena() const { return arena_base_ != nullptr; }
  void rewind() { arena_used_ = 0; }
  void* alloc(size_t bytes);
  void release(void* p);   // back to the cache
  void trim();             // hipFree everything cached (live blocks stay live)
  size_t live_bytes() const { return live_bytes_; }
  uint64_t requests() const { return requests_; }  // buffers handed out by alloc() so far, from the cache or from hipMalloc (not: pieces of an arena)

 private:
  std::multimap<size_t, void*> free_;
  std::unordered_map<void*, size_t> live_;
  size_t live_bytes_ = 0;
  uint64_t requests_ = 0;
  uint8_t* arena_base_ = nullptr;
  size_t arena_bytes_ = 0, arena_used_ = 0;
};

template <typename T>
struct DevBuf {  // RAII view of a pool allocation
  DevPool* pool = nullptr;
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(DevPool& pl, size_t count) : pool(&pl), p((T*)pl.alloc((count ? count : 1) * sizeof(T))), n(count) {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : pool(o.pool), p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { reset(); pool = o.pool; p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { reset(); }
  void reset() { if (p && pool) pool->release(p); p = nullptr; n = 0; }
  size_t bytes() const { return n * sizeof(T); }
};

// ---- primitives (dx_prims.hip) ------------------------------------------------------------
// Exclusive prefix sums; `total` (device pointer, may be null) receives the grand total.
void exclusive_scan_u8_u32(DevPool& pool, const uint8_t* in, uint32_t* out, size_t n, uint32_t* total, hipStream_t s);
void exclusive_scan_u32_u32(DevPool& pool, const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, hipStream_t s);
void exclusive_scan_u32_u64(DevPool& pool, const uint32_t* in, uint64_t* out, size_t n, uint64_t* total, hipStream_t s);
// Stable LSD radix sort of 64-bit keys on bits [0, key_bits); vals (u32) optional. Result ends in
// keys/vals; tmp arrays are taken from the pool.
void radix_sort_u64(DevPool& pool, uint64_t* keys, uint32_t* vals, size_t n, int key_bits, hipStream_t s);
// Keys only, on a pool allocation: `keys` may come back as another allocation of the same size (no final copy).
void radix_sort_u64(DevPool& pool, DevBuf<uint64_t>& keys, size_t n, int key_bits, hipStream_t s);
void radix_sort_u32(DevPool& pool, DevBuf<uint32_t>& keys, size_t n, int key_bits, hipStream_t s);  // keys of at most 32 bits: half the traffic
// 128-bit keys (a dictionary too large for (predicate, subject, object) in 64 bits): twice the traffic per pass
using u128 = unsigned __int128;
void radix_sort_u128(DevPool& pool, DevBuf<u128>& keys, size_t n, int key_bits, hipStream_t s);
inline void radix_sort_keys(DevPool& pool, DevBuf<uint64_t>& keys, size_t n, int key_bits, hipStream_t s) { radix_sort_u64(pool, keys, n, key_bits, s); }
inline void radix_sort_keys(DevPool& pool, DevBuf<u128>& keys, size_t n, int key_bits, hipStream_t s) { radix_sort_u128(pool, keys, n, key_bits, s); }
void fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s);
void fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t s);

// ---- term hashing: one definition for host and device (dx_hash.h) ---------------------------
uint64_t hash_bytes_host(const uint8_t* p, size_t n, uint64_t seed);

}  // namespace dx
