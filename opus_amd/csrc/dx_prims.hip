// Device primitives: caching allocator, exclusive scans, stable LSD radix sort, fills.
#include "dx_device.h"
#include "dx_internal.h"

namespace dx {

// ---- DevPool ----------------------------------------------------------------------------------
static size_t round_alloc(size_t b) {
  const size_t g = (b <= (1u << 20)) ? 4096 : (2u << 20);
  return (b + g - 1) / g * g;
}
void* DevPool::alloc(size_t bytes) {
  const size_t sz = round_alloc(bytes ? bytes : 1);
  auto it = free_.find(sz);
  void* p = nullptr;
  if (it != free_.end()) {
    p = it->second;
    free_.erase(it);
  } else {
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
      trim();  // give cached blocks back and retry once
      e = hipMalloc(&p, sz);
      if (e != hipSuccess) throw DxError(DX_E_NOMEM, "hipMalloc of " + std::to_string(sz) + " bytes failed");
    }
  }
  live_[p] = sz;
  live_bytes_ += sz;
  return p;
}
void DevPool::release(void* p) {
  auto it = live_.find(p);
  if (it == live_.end()) return;
  free_.emplace(it->second, p);
  live_bytes_ -= it->second;
  live_.erase(it);
}
void DevPool::trim() {
  for (auto& kv : free_) (void)hipFree(kv.second);
  free_.clear();
}

// ---- fills -----------------------------------------------------------------------------------------
template <typename T>
__global__ void k_fill(T* p, T v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static unsigned grid_for(size_t n, unsigned per_block, unsigned cap = 4096) {
  size_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}
void fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_fill<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, p, v, n);
}
void fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_fill<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, p, v, n);
}

// ---- exclusive scan ----------------------------------------------------------------------------
// Three launches: per-tile sums, one block scanning the tile sums, per-tile scan with offset.
constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(SC_THREADS) k_scan_tile_sums(const TIn* __restrict__ in, size_t n, TOut* __restrict__ sums) {
  __shared__ TOut lds[8];
  const size_t base = (size_t)blockIdx.x * SC_TILE + (size_t)threadIdx.x * SC_ITEMS;
  TOut v = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i)
    if (base + i < n) v += (TOut)in[base + i];
  TOut total;
  (void)block_excl_scan_256<TOut>(v, lds, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

template <typename TOut>
__global__ void __launch_bounds__(SC_THREADS) k_scan_sums(TOut* __restrict__ sums, size_t nb, TOut* __restrict__ total_out) {
  __shared__ TOut lds[8];
  TOut carry = 0;
  for (size_t base = 0; base < nb; base += SC_THREADS) {
    const size_t i = base + threadIdx.x;
    TOut v = i < nb ? sums[i] : 0, total;
    TOut ex = block_excl_scan_256<TOut>(v, lds, total);
    if (i < nb) sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(SC_THREADS)
k_scan_tiles(const TIn* __restrict__ in, TOut* __restrict__ out, size_t n, const TOut* __restrict__ sums) {
  __shared__ TOut lds[8];
  const size_t base = (size_t)blockIdx.x * SC_TILE + (size_t)threadIdx.x * SC_ITEMS;
  TOut x[SC_ITEMS];
  TOut v = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    x[i] = (base + i < n) ? (TOut)in[base + i] : 0;
    v += x[i];
  }
  TOut total;
  TOut ex = block_excl_scan_256<TOut>(v, lds, total) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += x[i];
  }
}

template <typename TIn, typename TOut>
static void exclusive_scan(DevPool& pool, const TIn* in, TOut* out, size_t n, TOut* total, hipStream_t s) {
  if (n == 0) {
    if (total) DX_HIP(hipMemsetAsync(total, 0, sizeof(TOut), s));
    return;
  }
  const size_t nb = (n + SC_TILE - 1) / SC_TILE;
  DevBuf<TOut> sums(pool, nb);
  hipLaunchKernelGGL((k_scan_tile_sums<TIn, TOut>), dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, n, sums.p);
  hipLaunchKernelGGL((k_scan_sums<TOut>), dim3(1), dim3(SC_THREADS), 0, s, sums.p, nb, total);
  hipLaunchKernelGGL((k_scan_tiles<TIn, TOut>), dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, out, n, sums.p);
  DX_HIP(hipGetLastError());
  // `sums` returns to the pool here; later launches on the same stream are ordered behind the
  // kernels above, so handing the block out again is safe as long as all work shares stream `s`.
}
void exclusive_scan_u8_u32(DevPool& pool, const uint8_t* in, uint32_t* out, size_t n, uint32_t* total, hipStream_t s) {
  exclusive_scan<uint8_t, uint32_t>(pool, in, out, n, total, s);
}
void exclusive_scan_u32_u32(DevPool& pool, const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, hipStream_t s) {
  exclusive_scan<uint32_t, uint32_t>(pool, in, out, n, total, s);
}
void exclusive_scan_u32_u64(DevPool& pool, const uint32_t* in, uint64_t* out, size_t n, uint64_t* total, hipStream_t s) {
  exclusive_scan<uint32_t, uint64_t>(pool, in, out, n, total, s);
}

// ---- radix sort ----------------------------------------------------------------------------------
// 8-bit digits, least significant first. Per pass: (1) per-tile digit histogram in LDS, (2) scan of
// the [digit][tile] matrix, (3) scatter. A tile is 2048 keys; each of the 4 waves of a block owns a
// contiguous 512-key quarter, so (wave, round, lane) order equals key order and ranking by wave-level
// ballots keeps the sort stable.
constexpr int RS_THREADS = 256;
constexpr int RS_KPT = 8;
constexpr int RS_TILE = RS_THREADS * RS_KPT;
constexpr int RS_RADIX = 256;

template <typename K>
__global__ void __launch_bounds__(RS_THREADS)
k_rs_hist(const K* __restrict__ keys, size_t n, int shift, uint32_t* __restrict__ tile_hist, unsigned ntiles) {
  __shared__ uint32_t hist[RS_RADIX];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int j = 0; j < RS_KPT; ++j) {
    const size_t i = base + (size_t)j * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  tile_hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

template <typename K, bool HAS_VALS>
__global__ void __launch_bounds__(RS_THREADS)
k_rs_scatter(const K* __restrict__ kin, const uint32_t* __restrict__ vin, K* __restrict__ kout,
             uint32_t* __restrict__ vout, size_t n, int shift, const uint32_t* __restrict__ tile_off, unsigned ntiles) {
  __shared__ uint32_t whist[4][RS_RADIX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) whist[i][threadIdx.x] = 0;
  __syncthreads();

  const size_t wbase = (size_t)blockIdx.x * RS_TILE + (size_t)w * (64 * RS_KPT);
  K k[RS_KPT];
  uint32_t v[RS_KPT];
  uint32_t rank[RS_KPT];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int j = 0; j < RS_KPT; ++j) {
    const size_t i = wbase + (size_t)j * 64 + lane;
    const bool valid = i < n;
    k[j] = valid ? kin[i] : 0;
    if (HAS_VALS) v[j] = valid ? vin[i] : 0;
    const uint32_t d = (uint32_t)(k[j] >> shift) & 0xFF;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    // lanes of one wave run in lockstep: every peer reads the counter before the leader bumps it
    uint32_t old = 0;
    if (valid) old = whist[w][d];
    const uint32_t below = (uint32_t)__popcll(peers & lt);
    rank[j] = old + below;
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) whist[w][d] = old + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {  // thread t turns the four wave counts of digit t into output bases
    const int d = threadIdx.x;
    uint32_t run = tile_off[(size_t)d * ntiles + blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t c = whist[i][d];
      whist[i][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_KPT; ++j) {
    const size_t i = wbase + (size_t)j * 64 + lane;
    if (i < n) {
      const uint32_t d = (uint32_t)(k[j] >> shift) & 0xFF;
      const uint32_t pos = whist[w][d] + rank[j];
      kout[pos] = k[j];
      if (HAS_VALS) vout[pos] = v[j];
    }
  }
}

// The passes alternate between the caller's array and a scratch array of the same size. When `keys_buf` is
// given (it must be the pool allocation `keys` points at, with n elements) and the result ends in the scratch
// array, the two allocations change places instead of being copied (an odd number of passes over 13 M keys
// would otherwise end with a 100 MB copy).
template <typename K>
static void radix_sort_impl(DevPool& pool, K* keys, DevBuf<K>* keys_buf, uint32_t* vals, size_t n, int key_bits, hipStream_t s) {
  if (n < 2 || key_bits <= 0) return;
  if (n >= (1ull << 32)) throw DxError(DX_E_RANGE, "radix_sort_u64: more than 2^32 keys");
  const unsigned ntiles = (unsigned)((n + RS_TILE - 1) / RS_TILE);
  DevBuf<K> ktmp(pool, keys_buf ? keys_buf->n : n);
  DevBuf<uint32_t> vtmp(pool, vals ? n : 1);
  DevBuf<uint32_t> hist(pool, (size_t)RS_RADIX * ntiles);
  K* kin = keys; K* kout = ktmp.p;
  uint32_t* vin = vals; uint32_t* vout = vtmp.p;
  const int passes = (key_bits + 7) / 8;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * 8;
    hipLaunchKernelGGL(k_rs_hist<K>, dim3(ntiles), dim3(RS_THREADS), 0, s, kin, n, shift, hist.p, ntiles);
    exclusive_scan_u32_u32(pool, hist.p, hist.p, (size_t)RS_RADIX * ntiles, nullptr, s);
    if (vals)
      hipLaunchKernelGGL((k_rs_scatter<K, true>), dim3(ntiles), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n, shift, hist.p, ntiles);
    else
      hipLaunchKernelGGL((k_rs_scatter<K, false>), dim3(ntiles), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n, shift, hist.p, ntiles);
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  DX_HIP(hipGetLastError());
  if (kin != keys) {
    if (keys_buf && !vals) {
      std::swap(*keys_buf, ktmp);  // the stream is in order: the old array goes back to the pool only for later launches
    } else {
      DX_HIP(hipMemcpyAsync(keys, kin, n * sizeof(K), hipMemcpyDeviceToDevice, s));
      if (vals) DX_HIP(hipMemcpyAsync(vals, vin, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
  }
}

void radix_sort_u64(DevPool& pool, uint64_t* keys, uint32_t* vals, size_t n, int key_bits, hipStream_t s) {
  radix_sort_impl<uint64_t>(pool, keys, nullptr, vals, n, key_bits, s);
}
void radix_sort_u64(DevPool& pool, DevBuf<uint64_t>& keys, size_t n, int key_bits, hipStream_t s) {
  radix_sort_impl<uint64_t>(pool, keys.p, &keys, nullptr, n, key_bits, s);
}
void radix_sort_u32(DevPool& pool, DevBuf<uint32_t>& keys, size_t n, int key_bits, hipStream_t s) {
  radix_sort_impl<uint32_t>(pool, keys.p, &keys, nullptr, n, key_bits, s);
}

}  // namespace dx
