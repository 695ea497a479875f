// Device primitives: caching allocator, exclusive scans, stable LSD radix sort, fills.
#include <algorithm>

#include "dx_device.h"
#include "dx_internal.h"

namespace dx {

// ---- DevPool ----------------------------------------------------------------------------------
static size_t round_alloc(size_t b) {
  const size_t g = (b <= (1u << 20)) ? 4096 : (2u << 20);
  return (b + g - 1) / g * g;
}
void* DevPool::alloc(size_t bytes) {
  if (arena_base_) {
    const size_t sz = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
    if (sz > arena_bytes_ - arena_used_)
      throw DxError(DX_E_RANGE, "the statement needs more device scratch than the fixed " + std::to_string(arena_bytes_ >> 20) +
                                    " MiB of the entry point that allocates nothing");
    void* p = arena_base_ + arena_used_;
    arena_used_ += sz;
    return p;
  }
  ++requests_;
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
  if (arena_base_) return;
  auto it = live_.find(p);
  if (it == live_.end()) return;
  free_.emplace(it->second, p);
  live_bytes_ -= it->second;
  live_.erase(it);
}
void DevPool::trim() {
  if (arena_base_) return;
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
// Three launches: per-tile sums, one block scanning the tile sums, per-tile scan with offset (two launches up to
// a thousand tiles, see k_scan_tiles).
// A tile is 4096 items, walked as four stripes of 1024: in every stripe thread t owns the four consecutive
// items 4t..4t+3, so a wave reads 256 (u8) or 1024 (u32) contiguous bytes per load and writes 1 KiB (u32
// out) or 2 KiB (u64 out) of contiguous results per stripe; all loads of a tile are requested before the
// first is used. Inside a tile the order is (stripe, wave, lane): sixteen wave totals go through LDS once.
constexpr int SC_THREADS = 256;
constexpr int SC_STRIPES = 4;
constexpr int SC_STRIPE = SC_THREADS * 4;
constexpr int SC_TILE = SC_STRIPE * SC_STRIPES;
constexpr size_t SC_FUSED_TILES = 1024;  // up to 4 M items

__device__ __forceinline__ void load4(const uint8_t* in, size_t i, size_t n, bool aligned, uint32_t x[4]) {
  if (aligned && i + 4 <= n) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(in + i);
    x[0] = w & 0xFF; x[1] = (w >> 8) & 0xFF; x[2] = (w >> 16) & 0xFF; x[3] = w >> 24;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = i + k < n ? in[i + k] : 0;
  }
}
__device__ __forceinline__ void load4(const uint32_t* in, size_t i, size_t n, bool aligned, uint32_t x[4]) {
  if (aligned && i + 4 <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + i);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = i + k < n ? in[i + k] : 0;
  }
}
__device__ __forceinline__ void store4(uint32_t* out, size_t i, size_t n, bool aligned, const uint32_t y[4]) {
  if (aligned && i + 4 <= n) {
    *reinterpret_cast<uint4*>(out + i) = make_uint4(y[0], y[1], y[2], y[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k < n) out[i + k] = y[k];
  }
}
__device__ __forceinline__ void store4(uint64_t* out, size_t i, size_t n, bool aligned, const uint64_t y[4]) {
  if (aligned && i + 4 <= n) {
    *reinterpret_cast<ulonglong2*>(out + i) = make_ulonglong2(y[0], y[1]);
    *reinterpret_cast<ulonglong2*>(out + i + 2) = make_ulonglong2(y[2], y[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k < n) out[i + k] = y[k];
  }
}
template <typename T> __device__ __forceinline__ T wave_incl_scan(T v, int lane) {
  if constexpr (sizeof(T) == 8) return (T)wave_incl_scan_u64((uint64_t)v, lane);
  else return (T)wave_incl_scan_u32((uint32_t)v, lane);
}

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(SC_THREADS) k_scan_tile_sums(const TIn* in, size_t n, TOut* __restrict__ sums) {
  __shared__ TOut lds[4];
  const bool aligned = (reinterpret_cast<uintptr_t>(in) & (4 * sizeof(TIn) - 1)) == 0;
  const size_t tile0 = (size_t)blockIdx.x * SC_TILE + 4 * (size_t)threadIdx.x;
  uint32_t x[SC_STRIPES][4];
#pragma unroll
  for (int r = 0; r < SC_STRIPES; ++r) load4(in, tile0 + (size_t)r * SC_STRIPE, n, aligned, x[r]);
  TOut v = 0;
#pragma unroll
  for (int r = 0; r < SC_STRIPES; ++r) v += (TOut)x[r][0] + x[r][1] + x[r][2] + x[r][3];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const TOut incl = wave_incl_scan<TOut>(v, lane);
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// One block turns the tile sums into their exclusive prefix: every thread walks a contiguous run of them.
template <typename TOut>
__global__ void __launch_bounds__(SC_THREADS) k_scan_sums(TOut* __restrict__ sums, size_t nb, TOut* __restrict__ total_out) {
  __shared__ TOut lds[8];
  const size_t per = (nb + SC_THREADS - 1) / SC_THREADS;
  const size_t b = std::min(nb, per * threadIdx.x), e = std::min(nb, b + per);
  TOut v = 0;
  for (size_t i = b; i < e; ++i) v += sums[i];
  TOut total;
  TOut run = block_excl_scan_256<TOut>(v, lds, total);
  for (size_t i = b; i < e; ++i) {
    const TOut x = sums[i];
    sums[i] = run;
    run += x;
  }
  if (threadIdx.x == 0 && total_out) *total_out = total;
}

// RAW_SUMS: `sums` still holds the tiles' own sums and every block adds up the ones before it itself (for up to a
// thousand tiles that is four loads per thread, and one launch fewer per scan); the last block writes the total.
template <typename TIn, typename TOut, bool RAW_SUMS>
__global__ void __launch_bounds__(SC_THREADS)
k_scan_tiles(const TIn* in, TOut* out, size_t n, const TOut* __restrict__ sums, TOut* __restrict__ total_out) {  // in == out is allowed
  __shared__ TOut lds[SC_STRIPES * 4];
  __shared__ TOut lds_before[4];
  const bool aligned_in = (reinterpret_cast<uintptr_t>(in) & (4 * sizeof(TIn) - 1)) == 0;
  const bool aligned_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const size_t tile0 = (size_t)blockIdx.x * SC_TILE + 4 * (size_t)threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t x[SC_STRIPES][4];
#pragma unroll
  for (int r = 0; r < SC_STRIPES; ++r) load4(in, tile0 + (size_t)r * SC_STRIPE, n, aligned_in, x[r]);
  TOut mine[SC_STRIPES], incl[SC_STRIPES];
#pragma unroll
  for (int r = 0; r < SC_STRIPES; ++r) {
    mine[r] = (TOut)x[r][0] + x[r][1] + x[r][2] + x[r][3];
    incl[r] = wave_incl_scan<TOut>(mine[r], lane);
    if (lane == 63) lds[r * 4 + w] = incl[r];
  }
  if constexpr (RAW_SUMS) {
    TOut part = 0;
    for (size_t i = threadIdx.x; i < blockIdx.x; i += SC_THREADS) part += sums[i];
    part = wave_incl_scan<TOut>(part, lane);
    if (lane == 63) lds_before[w] = part;
  }
  __syncthreads();
  TOut before;  // everything ahead of (stripe r, wave w) in the tile's order
  if constexpr (RAW_SUMS) {
    before = lds_before[0] + lds_before[1] + lds_before[2] + lds_before[3];
    if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
      TOut t = before;
#pragma unroll
      for (int k = 0; k < SC_STRIPES * 4; ++k) t += lds[k];
      *total_out = t;
    }
  } else {
    before = sums[blockIdx.x];
  }
#pragma unroll
  for (int r = 0; r < SC_STRIPES; ++r) {
    TOut here = before;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const TOut t = lds[r * 4 + k];
      if (k < w) here += t;
      before += t;
    }
    TOut y[4];
    y[0] = here + incl[r] - mine[r];
    y[1] = y[0] + x[r][0];
    y[2] = y[1] + x[r][1];
    y[3] = y[2] + x[r][2];
    store4(out, tile0 + (size_t)r * SC_STRIPE, n, aligned_out, y);
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
  if (nb <= SC_FUSED_TILES) {  // two launches (one for a single tile): the tiles add up the sums before them themselves
    if (nb > 1) hipLaunchKernelGGL((k_scan_tile_sums<TIn, TOut>), dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, n, sums.p);
    hipLaunchKernelGGL((k_scan_tiles<TIn, TOut, true>), dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, out, n, sums.p, total);
  } else {
    hipLaunchKernelGGL((k_scan_tile_sums<TIn, TOut>), dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, n, sums.p);
    hipLaunchKernelGGL((k_scan_sums<TOut>), dim3(1), dim3(SC_THREADS), 0, s, sums.p, nb, total);
    hipLaunchKernelGGL((k_scan_tiles<TIn, TOut, false>), dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, out, n, sums.p, (TOut*)nullptr);
  }
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
// Digits of 8 (or 9, see radix_sort_impl) bits, least significant first. Per pass: (1) per-tile digit histogram in LDS, (2) scan of the [digit][tile] matrix, (3) scatter. A tile is
// 4096 keys; each of the 4 waves of a block owns a contiguous 1024-key quarter, so (wave, round, lane) order equals
// key order and ranking by wave-level ballots keeps the sort stable. (Sixteen keys per lane: eight per lane means
// twice the tiles to scan and runs half as long in the scatter, thirty-two too many registers.)
constexpr int RS_THREADS = 256;
constexpr int RS_KPT = 16;
constexpr int RS_TILE = RS_THREADS * RS_KPT;

// Which tile a block works on. A row of the [digit][tile] count matrix holds sixteen tiles' counts per 64 bytes; blocks
// whose numbers are equal modulo 8 share a second-level cache, so inside every run of 128 block numbers the sixteen
// blocks of one such class take sixteen neighbouring tiles: their 4-byte counts meet in one cache and leave it as one
// line instead of sixteen partial ones (and come back the same way in the scatter). The last, partial run maps 1:1.
__device__ __forceinline__ unsigned rs_tile_of(unsigned b, unsigned ntiles) {
  if (b >= (ntiles & ~127u)) return b;
  const unsigned r = b & 127u;
  return (b & ~127u) + ((r & 7u) << 4) + (r >> 3);
}

// lanes of the wave whose digit equals this lane's (among the lanes with `valid`)
template <int BITS>
__device__ __forceinline__ uint64_t digit_peers(uint32_t d, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const bool bit = (d >> b) & 1;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// the RS_KPT keys of a lane, all requested before the first is used; keys past the end read as 0
template <typename K>
__device__ __forceinline__ void load_keys(const K* __restrict__ keys, size_t first, size_t stride, size_t n, K (&k)[RS_KPT]) {
  if (first + (RS_KPT - 1) * stride < n) {
#pragma unroll
    for (int j = 0; j < RS_KPT; ++j) k[j] = keys[first + (size_t)j * stride];
  } else {
#pragma unroll
    for (int j = 0; j < RS_KPT; ++j) {
      const size_t i = first + (size_t)j * stride;
      k[j] = i < n ? keys[i] : 0;
    }
  }
}

// Keys that arrive sorted by another field (the rows of one predicate, one after another) put all 64 lanes of a
// wave on one counter, and LDS serialises those: a wave whose 64 keys share the digit adds 64 once. Any other wave
// counts with plain LDS atomics (finding the lanes that share a digit by ballots, as the scatter has to for its
// ranks, cost more than the conflicts it saved: 25 us against 14 us per pass over 13 M keys).
// (Counting four tiles per block, to write the [digit][tile] matrix in 16-byte pieces, measured no faster.)
template <typename K, int BITS>
__global__ void __launch_bounds__(RS_THREADS)
k_rs_hist(const K* __restrict__ keys, size_t n, int shift, uint32_t* __restrict__ tile_hist, size_t stride, unsigned ntiles) {
  constexpr int R = 1 << BITS;
  __shared__ uint32_t hist[R];
  for (int d = threadIdx.x; d < R; d += RS_THREADS) hist[d] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned tile = rs_tile_of(blockIdx.x, ntiles);
  const size_t first = (size_t)tile * RS_TILE + threadIdx.x;
  K k[RS_KPT];
  load_keys(keys, first, RS_THREADS, n, k);
#pragma unroll
  for (int j = 0; j < RS_KPT; ++j) {
    const bool valid = first + (size_t)j * RS_THREADS < n;
    const uint32_t d = (uint32_t)(k[j] >> shift) & (R - 1);
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
    if (__all(valid && d == d0)) {
      if (lane == 0) atomicAdd(&hist[d0], 64u);
    } else if (valid) {
      atomicAdd(&hist[d], 1u);
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += RS_THREADS) {
    tile_hist[(size_t)d * stride + tile] = hist[d];
    if (tile == ntiles - 1)  // the padding of the rows takes part in the scan of the matrix
      for (size_t c = ntiles; c < stride; ++c) tile_hist[(size_t)d * stride + c] = 0;
  }
}

// 8-byte keys leave the tile through LDS (4-byte keys gain nothing from it: measured 47 us staged against 42 us direct
// per pass of the VP index's entries; 16-byte keys would need 64 KiB): ranked keys are first put in the tile's sorted order there,
// then thread t writes elements t, t + 256, ..., so that the lanes of a wave write neighbouring addresses of a
// digit's run (8 or 16 keys per digit and tile) instead of 64 separate 4- or 8-byte pieces per store instruction.
template <typename K, bool HAS_VALS, int BITS>
__global__ void __launch_bounds__(RS_THREADS)
k_rs_scatter(const K* __restrict__ kin, const uint32_t* __restrict__ vin, K* __restrict__ kout,
             uint32_t* __restrict__ vout, size_t n, int shift, const uint32_t* __restrict__ tile_off, size_t stride, unsigned ntiles) {
  constexpr bool kStaged = sizeof(K) == 8;
  constexpr int R = 1 << BITS;
  constexpr int kDigitsPerThread = R / RS_THREADS;  // thread t looks after digits t * kDigitsPerThread ...
  __shared__ uint32_t whist[4][R];
  __shared__ uint32_t gbase[kStaged ? R : 1];
  __shared__ uint32_t scan_lds[8];
  __shared__ K stage[kStaged ? RS_TILE : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned tile = rs_tile_of(blockIdx.x, ntiles);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    for (int d = threadIdx.x; d < R; d += RS_THREADS) whist[i][d] = 0;
  __syncthreads();

  const size_t wbase = (size_t)tile * RS_TILE + (size_t)w * (64 * RS_KPT);
  K k[RS_KPT];
  uint32_t v[RS_KPT];
  uint32_t rank[RS_KPT];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  load_keys(kin, wbase + lane, 64, n, k);
  if (HAS_VALS) load_keys(vin, wbase + lane, 64, n, v);
#pragma unroll
  for (int j = 0; j < RS_KPT; ++j) {
    const size_t i = wbase + (size_t)j * 64 + lane;
    const bool valid = i < n;
    const uint32_t d = (uint32_t)(k[j] >> shift) & (R - 1);
    const uint64_t peers = digit_peers<BITS>(d, valid);
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
  if constexpr (!kStaged) {
    // a thread turns the four wave counts of its digits into output bases
    for (int d = threadIdx.x; d < R; d += RS_THREADS) {
      uint32_t run = tile_off[(size_t)d * stride + tile];
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
        const uint32_t d = (uint32_t)(k[j] >> shift) & (R - 1);
        const uint32_t pos = whist[w][d] + rank[j];
        kout[pos] = k[j];
        if (HAS_VALS) vout[pos] = v[j];
      }
    }
  } else {
    {  // where each digit's run begins inside the tile, and what to add to a tile position to get the output position
      uint32_t c[kDigitsPerThread][4], mine = 0;
#pragma unroll
      for (int q = 0; q < kDigitsPerThread; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) { c[q][i] = whist[i][threadIdx.x * kDigitsPerThread + q]; mine += c[q][i]; }
      uint32_t total;
      uint32_t run = block_excl_scan_256<uint32_t>(mine, scan_lds, total);
#pragma unroll
      for (int q = 0; q < kDigitsPerThread; ++q) {
        const int d = threadIdx.x * kDigitsPerThread + q;
        gbase[d] = tile_off[(size_t)d * stride + tile] - run;
#pragma unroll
        for (int i = 0; i < 4; ++i) { whist[i][d] = run; run += c[q][i]; }
      }
    }
    __syncthreads();
    uint32_t lp[RS_KPT];
#pragma unroll
    for (int j = 0; j < RS_KPT; ++j) {
      const size_t i = wbase + (size_t)j * 64 + lane;
      const uint32_t d = (uint32_t)(k[j] >> shift) & (R - 1);
      lp[j] = whist[w][d] + rank[j];
      if (i < n) stage[lp[j]] = k[j];
    }
    __syncthreads();
    const size_t left = n - (size_t)tile * RS_TILE;
    const uint32_t cnt = left < (size_t)RS_TILE ? (uint32_t)left : (uint32_t)RS_TILE;
    uint32_t gp[RS_KPT];
#pragma unroll
    for (int j = 0; j < RS_KPT; ++j) {
      const uint32_t i = (uint32_t)j * RS_THREADS + threadIdx.x;
      if (i < cnt) {
        const K key = stage[i];
        gp[j] = gbase[(uint32_t)(key >> shift) & (R - 1)] + i;
        kout[gp[j]] = key;
      }
    }
    if constexpr (HAS_VALS) {
      uint32_t* vstage = reinterpret_cast<uint32_t*>(stage);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RS_KPT; ++j) {
        const size_t i = wbase + (size_t)j * 64 + lane;
        if (i < n) vstage[lp[j]] = v[j];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RS_KPT; ++j) {
        const uint32_t i = (uint32_t)j * RS_THREADS + threadIdx.x;
        if (i < cnt) vout[gp[j]] = vstage[i];
      }
    }
  }
}

template <typename K, int BITS>
static void radix_sort_pass(DevPool& pool, const K* kin, const uint32_t* vin /*null: keys only*/, K* kout, uint32_t* vout, size_t n, int shift,
                            uint32_t* hist, size_t stride, unsigned ntiles, hipStream_t s) {
  hipLaunchKernelGGL((k_rs_hist<K, BITS>), dim3(ntiles), dim3(RS_THREADS), 0, s, kin, n, shift, hist, stride, ntiles);
  exclusive_scan_u32_u32(pool, hist, hist, (size_t)(1 << BITS) * stride, nullptr, s);
  if (vin)
    hipLaunchKernelGGL((k_rs_scatter<K, true, BITS>), dim3(ntiles), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n, shift, hist, stride, ntiles);
  else
    hipLaunchKernelGGL((k_rs_scatter<K, false, BITS>), dim3(ntiles), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, n, shift, hist, stride, ntiles);
}

// Digits are 8 bits wide, except that the topmost passes take 9 where that saves a pass: 49-bit keys (five predicate
// bits over two 22-bit terms, LUBM-100) sort in five 8-bit passes and one 9-bit pass instead of seven passes. A 9-bit
// pass costs about 1.4 passes of 8 bits (half as long runs in the scatter, a ninth ballot per ranking step), so all
// digits at 9 bits lose (0.67 ms against 0.62 ms for 13 M keys) and one in place of two wins.
template <typename K>
static void radix_sort_impl(DevPool& pool, K* keys, DevBuf<K>* keys_buf, uint32_t* vals, size_t n, int key_bits, hipStream_t s) {
  if (n < 2 || key_bits <= 0) return;
  if (n >= (1ull << 32)) throw DxError(DX_E_RANGE, "radix_sort_u64: more than 2^32 keys");
  int passes = (key_bits + 7) / 8, wide = 0;
  if ((key_bits + 8) / 9 < passes && key_bits - 8 * ((key_bits + 8) / 9) <= 2) {  // one pass saved for at most two wide ones
    passes = (key_bits + 8) / 9;
    wide = key_bits - 8 * passes;  // that many passes take 9 bits
  }
  const unsigned ntiles = (unsigned)((n + RS_TILE - 1) / RS_TILE);
  DevBuf<K> ktmp(pool, keys_buf ? keys_buf->n : n);
  DevBuf<uint32_t> vtmp(pool, vals ? n : 1);
  const size_t stride = ((size_t)ntiles + 15) & ~(size_t)15;  // rows of the count matrix begin on 64-byte boundaries
  DevBuf<uint32_t> hist(pool, (size_t)(wide ? 512 : 256) * stride);
  K* kin = keys; K* kout = ktmp.p;
  uint32_t* vin = vals; uint32_t* vout = vtmp.p;
  int shift = 0;
  for (int p = 0; p < passes; ++p) {
    if (p >= passes - wide) {
      radix_sort_pass<K, 9>(pool, kin, vals ? vin : nullptr, kout, vout, n, shift, hist.p, stride, ntiles, s);
      shift += 9;
    } else {
      radix_sort_pass<K, 8>(pool, kin, vals ? vin : nullptr, kout, vout, n, shift, hist.p, stride, ntiles, s);
      shift += 8;
    }
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  DX_HIP(hipGetLastError());
  // The passes alternate between the caller's array and a scratch array of the same size. When `keys_buf` is
  // given (it must be the pool allocation `keys` points at, with n elements) and the result ends in the scratch
  // array, the two allocations change places instead of being copied (an odd number of passes over 13 M keys
  // would otherwise end with a 100 MB copy).
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
void radix_sort_u128(DevPool& pool, DevBuf<u128>& keys, size_t n, int key_bits, hipStream_t s) {
  radix_sort_impl<u128>(pool, keys.p, &keys, nullptr, n, key_bits, s);
}

}  // namespace dx
