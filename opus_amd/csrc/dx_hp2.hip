// Query hot path: execution of a translated basic graph pattern on vertical-partition tables in HBM.
//
// The reference hands the SQL text to Spark (QueryProcessor.executeFinalQuery, QueryProcessor.scala:184-195):
// per sub-select a cached-table scan with string equality filters, then sort-merge joins with a 200-way
// exchange per join, then count (:173-176). Here every table is a pair of int64 columns, constants are
// resolved to term ids on the device, and the plan is left-deep:
//
//   * the running intermediate is always the hash-join build side (chained table: heads[] + next[],
//     one atomicExch per row, duplicate keys welcome);
//   * every base table is streamed exactly once, coalesced, with its constant filters fused into the
//     probe; matches are appended with one atomic per wave;
//   * the last step of a count query only counts;
//   * nothing returns to the host between steps: row counts live in device counters, kernels clamp to
//     their buffers' capacities, and the host reads all counters once at the end. A step that outgrew
//     its optimistic capacity is detected there (its counter holds the exact size needed) and the
//     statement is run again with that capacity.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <functional>
#include <sstream>

#include "dx_device.h"
#include "dx_sql.h"
#include "dx_state.h"

namespace dx {
namespace {

constexpr int MAXC = 10;
constexpr int MAXSTEPS = 30;
constexpr int kConstBytes = 2560;
constexpr int kMaxConsts = 32;

// An intermediate relation. Its rows are claimed from a counter by the blocks that emit them, and the L2 carries out
// the claims on one word one after another (about 8 ns apiece): an emitting kernel over ten thousand tiles spends a
// hundred microseconds on that alone. A relation that is expected to be large is therefore striped: kStripes counters,
// a block claims from the one its number picks, and stripe x owns the rows of every kStripes-th group of
// kStripeGroup rows -- row i of stripe x lies at (i / G) * (S * G) + x * G + i % G. The relation stays one buffer with
// the same capacity; near its end some rows are missing (the stripes do not end level), and whoever walks it by row
// number (k_build) asks row_is_there(). Rows reached through a join's chains are there by construction.
constexpr uint32_t kStripes = 8, kStripeGroupBits = 10, kStripeStride = 32;  // counters 256 bytes apart: a cache line and a channel each
struct RelDev {
  int64_t* col[MAXC];
  int ncols;
  uint32_t cap;
  const unsigned long long* count;  // stripes > 1: the first of kStripes counters, kStripeStride words apart
  uint32_t stripes;                 // 0 or 1: rows [0, *count)
};
__device__ __forceinline__ uint64_t stripe_row(uint64_t i, uint32_t x) {
  return ((i >> kStripeGroupBits) << (kStripeGroupBits + 3)) + ((uint64_t)x << kStripeGroupBits) + (i & ((1u << kStripeGroupBits) - 1));
}
static_assert(kStripes == 8, "stripe_row shifts by three bits");
// one past the last row that may be there
__device__ __forceinline__ uint32_t rel_rows(const RelDev& r) {
  if (r.stripes <= 1) {
    const unsigned long long c = *r.count;
    return c < r.cap ? (uint32_t)c : r.cap;
  }
  unsigned long long groups = 0;
  for (uint32_t x = 0; x < kStripes; ++x) {
    const unsigned long long g = (r.count[x * kStripeStride] + (1u << kStripeGroupBits) - 1) >> kStripeGroupBits;
    groups = g > groups ? g : groups;
  }
  const unsigned long long e = groups << (kStripeGroupBits + 3);
  return e < r.cap ? (uint32_t)e : r.cap;
}
__device__ __forceinline__ bool row_is_there(const RelDev& r, uint32_t row) {
  if (r.stripes <= 1) return true;  // the caller stays below rel_rows()
  const uint32_t x = (row >> kStripeGroupBits) & (kStripes - 1);
  const uint64_t i = ((uint64_t)(row >> (kStripeGroupBits + 3)) << kStripeGroupBits) | (row & ((1u << kStripeGroupBits) - 1));
  return i < r.count[x * kStripeStride];  // (a wave that walks 64 rows from a multiple of 64 asks about one stripe)
}

struct ArmDev {
  const int64_t* s;
  const int64_t* o;
  uint32_t rows;
  int fs, fo;      // index into consts of the s / o filter, -1 = none
  int nsrc;
  int src[3];      // 0 = s, 1 = o, 2 = constant
  int src_ci[3];
  long long src_val[3];  // the constant's id (projected constants are never looked up: they carry ids of their own)
  int direct;      // the tuple's columns are stored as they are (rows received from other ranks): no filters
  const int64_t* d[3];
};

template <int kBytes, int kCount>
struct ConstArgsT {
  char bytes[kBytes];
  uint32_t off[kCount + 1];
  long long preset[kCount];  // LLONG_MIN = look the string up, anything else = the id to use
  uint64_t hash[kCount];     // term hash of the string, computed on the host
  int n;
};
using ConstArgs = ConstArgsT<kConstBytes, kMaxConsts>;

__device__ __forceinline__ uint64_t hash_key(int64_t a) { return mix64((uint64_t)a * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull); }
__device__ __forceinline__ uint64_t hash_key2(int64_t a, int64_t b) { return mix64(hash_key(a) ^ ((uint64_t)b * 0xC2B2AE3D27D4EB4Full)); }
// Where a key's chain begins in a join's table of heads. Term ids follow the order of first appearance in the text, the
// tables are sorted by subject, and what belongs together was written together: the rows a wave probes with, and the
// rows of the intermediate they meet, carry ids that lie close to each other. So the low `near` bits of the (first) key
// are kept as they are and only the bits above them are mixed: keys inside one block of 2^near ids take neighbouring
// heads (a wave's 64 probes touch a handful of cache lines instead of 64), blocks are scattered over the table like
// hashed keys, and two different keys of one block never share a head. near = 0 mixes everything.
__device__ __forceinline__ uint32_t head_of(int64_t a, int near) {
  const uint64_t lo = (uint64_t)a & ((1ull << near) - 1);
  return (uint32_t)((hash_key(a >> near) << near) | lo);
}
__device__ __forceinline__ uint32_t head_of2(int64_t a, int64_t b, int near, int near_b) {
  const uint64_t lo = (uint64_t)a & ((1ull << near) - 1);
  return (uint32_t)((hash_key2(a >> near, b >> near_b) << near) | lo);
}

// Constants -> term ids, on the device, so the host never waits for them. One wave per constant:
// lane 0 walks the probe chain, then all lanes compare the bytes eight at a time. Returns the id in every lane.
template <class CA>
__device__ __forceinline__ long long resolve_const_value(const CA& ca, int i, int lane, const DictSlot* __restrict__ ht,
                                                         uint64_t mask, const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off) {
  if (ca.preset[i] != (-0x7FFFFFFFFFFFFFFFll - 1)) return ca.preset[i];
  const uint32_t b = ca.off[i], e = ca.off[i + 1], len = e - b;
  const uint64_t h = ca.hash[i];
  long long id = -1;
  if (lane == 0) {
    uint64_t slot = h & mask;
    for (;;) {
      const uint4 sv = *reinterpret_cast<const uint4*>(&ht[slot]);  // key and id in one load
      const uint64_t cur = ((uint64_t)sv.y << 32) | sv.x;
      if (cur == 0) break;
      if (cur == h) { id = (long long)sv.z; break; }  // one slot per hash
      slot = (slot + 1) & mask;
    }
  }
  id = __shfl(id, 0, 64);
  if (id < 0) return -1;
  const uint64_t pb = pool_off[id], pl = pool_off[id + 1] - pb;
  bool same = pl == len;
  if (same) {
    for (uint32_t j = lane; j < len; j += 64) same &= pool[pb + j] == (uint8_t)ca.bytes[b + j];
  }
  same = __all(same);
  return same ? id : -1;
}
template <class CA>
__device__ __forceinline__ void resolve_const(const CA& ca, int i, int lane, uint64_t seed, const DictSlot* __restrict__ ht,
                                              uint64_t mask, const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off,
                                              int64_t* __restrict__ out) {
  const long long v = resolve_const_value(ca, i, lane, ht, mask, pool, pool_off);
  if (lane == 0) out[i] = v;
}
__global__ void __launch_bounds__(64)
k_resolve_consts(ConstArgs ca, uint64_t seed, const DictSlot* __restrict__ ht, uint64_t mask,
                 const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off, int64_t* __restrict__ out) {
  if ((int)blockIdx.x < ca.n) resolve_const(ca, blockIdx.x, threadIdx.x, seed, ht, mask, pool, pool_off, out);
}
// Everything a statement needs before its first scan, in one launch (a statement is a chain of short kernels,
// and the gaps between launches cost as much as the kernels): the row counters are zeroed, the hash-join
// heads of every step are set to "empty", and the first wave of block i resolves constant i.
__global__ void __launch_bounds__(256)
k_query_init(ConstArgs ca, uint64_t seed, const DictSlot* __restrict__ ht, uint64_t mask, const uint8_t* __restrict__ pool,
             const uint64_t* __restrict__ pool_off, int64_t* __restrict__ consts, unsigned long long* __restrict__ counters, int n_counters,
             uint32_t* __restrict__ heads, uint64_t n_heads, unsigned long long* preset_counter = nullptr, unsigned long long preset_value = 0,
             unsigned long long* t_begin = nullptr) {
  if (t_begin && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(t_begin, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((int)blockIdx.x < ca.n && threadIdx.x < 64) resolve_const(ca, blockIdx.x, threadIdx.x, seed, ht, mask, pool, pool_off, consts);
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  if (t < (uint64_t)n_counters) counters[t] = 0;
  if (t == 0 && preset_counter) *preset_counter = preset_value;
  uint4* h4 = reinterpret_cast<uint4*>(heads);  // the arena is a multiple of four words
  for (uint64_t i = t; i < n_heads / 4; i += stride) h4[i] = make_uint4(kNil, kNil, kNil, kNil);
}

// ---- a statement's first and last kernel ------------------------------------------------------------------
// A short statement is a handful of kernels of a few microseconds each, and every kernel boundary costs as much
// as such a kernel. So there is no kernel of its own for the preparation and none for the hand-over:
//   * the statement's FIRST kernel (a scan) carries a Prologue: its
//     blocks resolve the statement's constants for the later kernels, set the hash-join heads of every step to
//     "empty", and each block resolves the one or two constants its own scan filters by for itself (the other
//     blocks' answers are not visible inside one kernel);
//   * the statement's LAST kernel carries an Epilogue: the block that finishes last (a ticket per block) stores
//     the row counters to the pinned host block, then the statement's number, which the host is watching, and
//     zeroes the counters and the ticket for the next statement (they live in the context, zero between
//     statements, so nobody has to clear them first).
constexpr int kProBytes = 1024, kProConsts = 16;
using ProConsts = ConstArgsT<kProBytes, kProConsts>;
struct Prologue {
  int enabled;
  ProConsts ca;
  const DictSlot* ht;
  uint64_t mask;
  const uint8_t* pool;
  const uint64_t* pool_off;
  int64_t* consts;
  uint32_t* heads;
  uint64_t n_heads;                       // multiple of four
  unsigned long long* t_begin;            // the device's wall clock when the statement's first kernel began (its first block)
};
struct Epilogue {
  int enabled;
  unsigned long long* counters;
  int n;
  unsigned long long* ticket;      // [0] main, [1 + i] one per kSubTickets: arrivals and partial sums, zero between statements
  unsigned long long* host_counters;
  unsigned long long* host_seq;
  unsigned long long seq;
  unsigned long long* t_begin;          // written by the statement's first kernel
  unsigned long long* host_clock;       // [0] the device's wall clock at the first kernel's begin, [1] at the last kernel's end
  unsigned long long* stripe_counters;  // step k's stripe x: [x * kStripeStride + k]
  unsigned long long* host_extent;      // per striped step: one past the last row of its relation that is in use
  uint32_t striped;                     // bit k: step k's relation is striped (its row counter is the sum of its stripes')
};

// Values of the arm's two filter constants. In the statement's first kernel they are looked up by this block
// (waves 1 and 2) while wave 0 and the grid as a whole do the statement's preparation. Contains a barrier:
// every thread of the block must call it.
__device__ __forceinline__ void stmt_prologue(const Prologue& pro, int fs, int fo, const int64_t* __restrict__ consts, int64_t& want_s,
                                              int64_t& want_o) {
  if (!pro.enabled) {
    want_s = fs >= 0 ? consts[fs] : 0;
    want_o = fo >= 0 ? consts[fo] : 0;
    return;
  }
  __shared__ long long own[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(pro.t_begin, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wave == 0) {
    for (int c = blockIdx.x; c < pro.ca.n; c += gridDim.x) resolve_const(pro.ca, c, lane, 0, pro.ht, pro.mask, pro.pool, pro.pool_off, pro.consts);
  } else if (wave == 1) {
    const long long v = fs >= 0 ? resolve_const_value(pro.ca, fs, lane, pro.ht, pro.mask, pro.pool, pro.pool_off) : 0;
    if (lane == 0) own[0] = v;
  } else if (wave == 2) {
    const long long v = fo >= 0 ? resolve_const_value(pro.ca, fo, lane, pro.ht, pro.mask, pro.pool, pro.pool_off) : 0;
    if (lane == 0) own[1] = v;
  }
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  uint4* h4 = reinterpret_cast<uint4*>(pro.heads);
  for (uint64_t i = t; i < pro.n_heads / 4; i += stride) h4[i] = make_uint4(kNil, kNil, kNil, kNil);
  __syncthreads();
  want_s = own[0];
  want_o = own[1];
}

// Every thread of the block must call it, after the block's last update of a row counter. `add` (thread 0's
// value counts) is this block's contribution to *add_to, or add_to is null.
// Arrivals are counted in two levels (blocks on kSubTickets words, the last arrival of each word on the main one):
// the L2 carries out atomics on one word one after another, about 8 ns apiece, and a few thousand blocks on one
// word would take longer than the scan they end. A counting kernel's block totals ride in the same words, the
// arrivals above bit kTicketShift and the sum below it, so that one atomic per block does both and the block
// that arrives last holds the total without any ordering between two addresses (a fence here makes every block
// write back and invalidate its L2, which cost more than the scan). Row counters that kernels claim output rows
// from need no fence either: a block has the answers of its own claims before it arrives here.
constexpr int kSubTickets = 64;
constexpr int kTicketShift = 44;
__device__ __forceinline__ void stmt_epilogue(const Epilogue& ep, unsigned long long add = 0, unsigned long long* add_to = nullptr) {
  if (!ep.enabled) {
    if (add_to && threadIdx.x == 0 && add) atomicAdd(add_to, add);
    return;
  }
  __shared__ int last;
  __shared__ unsigned long long total_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned sub = blockIdx.x % kSubTickets;
    const unsigned long long expect = (gridDim.x - sub + kSubTickets - 1) / kSubTickets;  // blocks b < gridDim.x with b % kSubTickets == sub
    const unsigned long long one = 1ull << kTicketShift, low = one - 1;
    int l = 0;
    unsigned long long v = atomicAdd(&ep.ticket[1 + sub], one | (add_to ? add : 0ull)) + (one | (add_to ? add : 0ull));
    if ((v >> kTicketShift) == expect) {  // last of this word: v's low part is the sum of its blocks
      __hip_atomic_store(&ep.ticket[1 + sub], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long words = gridDim.x < (unsigned)kSubTickets ? gridDim.x : (unsigned)kSubTickets;
      v = atomicAdd(&ep.ticket[0], one | (v & low)) + (one | (v & low));
      if ((v >> kTicketShift) == words) {
        l = 1;
        total_s = v & low;
        __hip_atomic_store(&ep.ticket[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    last = l;
  }
  __syncthreads();
  if (!last) return;
  if ((int)threadIdx.x < ep.n) {
    unsigned long long v = __hip_atomic_load(&ep.counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 32 && (ep.striped >> threadIdx.x & 1)) {
      unsigned long long most = 0;
      v = 0;
      for (uint32_t x = 0; x < kStripes; ++x) {
        unsigned long long* c = &ep.stripe_counters[x * kStripeStride + threadIdx.x];
        const unsigned long long n = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v += n;
        most = n > most ? n : most;
      }
      const unsigned long long groups = (most + (1u << kStripeGroupBits) - 1) >> kStripeGroupBits;
      __hip_atomic_store(&ep.host_extent[threadIdx.x], groups << (kStripeGroupBits + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (add_to && add_to == &ep.counters[threadIdx.x]) v += total_s;  // a counter other kernels of the step have added to as well
    __hip_atomic_store(&ep.host_counters[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&ep.counters[threadIdx.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 64) {  // a wave of its own, beside the counters' stores
    __hip_atomic_store(&ep.host_clock[0], __hip_atomic_load(ep.t_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&ep.host_clock[1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(ep.host_seq, ep.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// `pre`: the values of the filtered columns were requested earlier (pre_s / pre_o) and are not loaded again
// v[i] for an index known only at run time, without making the compiler keep v in scratch memory
__device__ __forceinline__ int64_t pick3(const int64_t (&v)[3], int i) { return i == 0 ? v[0] : i == 1 ? v[1] : v[2]; }

__device__ __forceinline__ bool arm_row(const ArmDev& a, const int64_t* __restrict__ consts, int64_t want_s, int64_t want_o, uint32_t i,
                                        int64_t v[3], bool pre = false, int64_t pre_s = 0, int64_t pre_o = 0) {
  if (a.direct) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < a.nsrc) v[k] = a.d[k][i];
    return true;
  }
  // columnar: the filtered column is read first, the other one only for rows that pass
  int64_t s = 0, o = 0;
  if (a.fo >= 0) { o = pre ? pre_o : a.o[i]; if (o != want_o) return false; }
  if (a.fs >= 0) { s = pre ? pre_s : a.s[i]; if (s != want_s) return false; }
  bool need_s = false, need_o = false;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < a.nsrc) { need_s |= a.src[k] == 0; need_o |= a.src[k] == 1; }
  if (need_s && a.fs < 0) s = a.s[i];
  if (need_o && a.fo < 0) o = a.o[i];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < a.nsrc) v[k] = a.src[k] == 0 ? s : a.src[k] == 1 ? o : a.src_val[k];
  return true;
}

// The rows a thread takes from one tile (rows first, first + 256, ...), fetched side by side: the filtered columns of
// all of them first (without a filter: everything the tuple needs), then the other column of the rows that passed.
// Taken row after row, every row waits for its own round trips to memory one after the other, and a kernel
// that is a chain of such waits runs at the memory's latency, not its rate. Returns the rows that pass as a bit
// mask; v[j] is set for those. pre_s / pre_o: values of the filtered columns requested earlier (when `pre`).
template <int kR>
__device__ __forceinline__ uint32_t arm_rows(const ArmDev& a, int64_t want_s, int64_t want_o, uint32_t first, int64_t (&v)[kR][3],
                                            bool pre = false, const int64_t* pre_s = nullptr, const int64_t* pre_o = nullptr) {
  uint32_t pass = 0;
  if (a.direct) {
#pragma unroll
    for (int j = 0; j < kR; ++j) {
      const uint32_t i = first + j * 256;
      if (i < a.rows) {
        pass |= 1u << j;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < a.nsrc) v[j][k] = a.d[k][i];
      }
    }
    return pass;
  }
  bool need_s = false, need_o = false;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < a.nsrc) { need_s |= a.src[k] == 0; need_o |= a.src[k] == 1; }
  const bool filtered = a.fs >= 0 || a.fo >= 0;
  int64_t s[kR], o[kR];
#pragma unroll
  for (int j = 0; j < kR; ++j) {
    const uint32_t i = first + j * 256;
    s[j] = o[j] = 0;
    if (i < a.rows) {
      if (a.fs >= 0) s[j] = pre ? pre_s[j] : a.s[i];
      else if (need_s && !filtered) s[j] = a.s[i];
      if (a.fo >= 0) o[j] = pre ? pre_o[j] : a.o[i];
      else if (need_o && !filtered) o[j] = a.o[i];
    }
  }
#pragma unroll
  for (int j = 0; j < kR; ++j) {
    const uint32_t i = first + j * 256;
    if (i < a.rows && (a.fs < 0 || s[j] == want_s) && (a.fo < 0 || o[j] == want_o)) pass |= 1u << j;
  }
  if (filtered) {
#pragma unroll
    for (int j = 0; j < kR; ++j) {
      const uint32_t i = first + j * 256;
      if (pass >> j & 1) {
        if (need_s && a.fs < 0) s[j] = a.s[i];
        if (need_o && a.fo < 0) o[j] = a.o[i];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kR; ++j)
    if (pass >> j & 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < a.nsrc) v[j][k] = a.src[k] == 0 ? s[j] : a.src[k] == 1 ? o[j] : a.src_val[k];
    }
  return pass;
}

// Filtered scan that only counts (single-relation count queries). The grid is kept to kCountBlocks blocks: the
// kernel ends with one atomic per block on one word, and the L2 carries those out one after another (two
// thousand of them took longer than reading the column). A thread therefore walks many rows, and in the
// usual case -- one constant filter, nothing else read -- it has eight loads in flight at a time.
constexpr unsigned kCountBlocks = 512;
__global__ void __launch_bounds__(256)
k_scan_count(ArmDev a, const int64_t* __restrict__ consts, unsigned long long* __restrict__ total, Prologue pro, Epilogue ep) {
  int64_t want_s, want_o;
  uint32_t c = 0;
  const uint32_t nthreads = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (!a.direct && ((a.fo >= 0) != (a.fs >= 0))) {
    const int64_t* __restrict__ col = a.fo >= 0 ? a.o : a.s;
    uint64_t i = t;  // 64-bit: the stride must not wrap on a table of nearly 2^32 rows
    // the first loads do not depend on the constant: they travel while the prologue looks it up
    int64_t v0[8];
    const bool first_batch = i + 7ull * nthreads < a.rows;
    if (first_batch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v0[j] = col[i + j * nthreads];
    }
    stmt_prologue(pro, a.fs, a.fo, consts, want_s, want_o);
    const int64_t want = a.fo >= 0 ? want_o : want_s;
    if (first_batch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c += v0[j] == want ? 1u : 0u;
      i += 8ull * nthreads;
    }
    for (; i + 7ull * nthreads < a.rows; i += 8ull * nthreads) {
      int64_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = col[i + j * nthreads];
#pragma unroll
      for (int j = 0; j < 8; ++j) c += v[j] == want ? 1u : 0u;
    }
    for (; i < a.rows; i += nthreads) c += col[i] == want ? 1u : 0u;
  } else {
    stmt_prologue(pro, a.direct ? -1 : a.fs, a.direct ? -1 : a.fo, consts, want_s, want_o);
    for (uint64_t i = t; i < a.rows; i += nthreads) {
      int64_t v[3];
      c += arm_row(a, consts, want_s, want_o, (uint32_t)i, v) ? 1u : 0u;
    }
  }
  __shared__ uint32_t lds[8];
  uint32_t block_total;
  (void)block_excl_scan_256<uint32_t>(c, lds, block_total);
  stmt_epilogue(ep, block_total, total);
}

// Output rows are claimed from one counter. Thousands of waves claiming for themselves are served one after
// another by the L2, about 8 ns apiece, which for a selective scan or join is most of the kernel: so a block
// works on a tile of kTileRows rows, counts what it will emit, claims once per tile, then writes.
// Small tables take one row per thread (every block gets a tile at once), large ones four.
constexpr uint64_t kSmallTable = 256ull * 2048;
__device__ __forceinline__ unsigned long long block_claim(uint32_t mine, unsigned long long* counter, uint32_t& before) {
  __shared__ uint32_t lds[8];
  __shared__ unsigned long long base_s;
  uint32_t total;
  before = block_excl_scan_256<uint32_t>(mine, lds, total);
  if (threadIdx.x == 0) base_s = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
  __syncthreads();
  const unsigned long long base = base_s;
  __syncthreads();  // base_s is written again in the next tile
  return base;
}
// The counter a block claims its output rows from, and where the i-th row it was given lies.
__device__ __forceinline__ unsigned long long* claim_counter(const RelDev& out, unsigned long long* counter) {
  return out.stripes > 1 ? counter + (blockIdx.x & (kStripes - 1)) * kStripeStride : counter;
}
__device__ __forceinline__ uint64_t out_row(const RelDev& out, uint64_t i) {
  return out.stripes > 1 ? stripe_row(i, blockIdx.x & (kStripes - 1)) : i;
}

// Filtered scan that appends the projected tuple to `out` (first relation of a plan).
template <int kRowsPerThread>
__global__ void __launch_bounds__(256)
k_scan_emit(ArmDev a, const int64_t* __restrict__ consts, RelDev out, unsigned long long* __restrict__ out_count, Prologue pro, Epilogue ep) {
  constexpr int kTileRows = 256 * kRowsPerThread;
  const uint32_t ntiles = (a.rows + kTileRows - 1) / kTileRows;
  int64_t want_s, want_o;
  // In the statement's first kernel the filtered columns of the block's first tile are requested before the
  // prologue looks the constants up: the two waits overlap.
  int64_t pre_s[kRowsPerThread], pre_o[kRowsPerThread];
  const bool pre = pro.enabled && !a.direct && blockIdx.x < ntiles;
  if (pre) {
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      const uint32_t i = blockIdx.x * kTileRows + j * 256 + threadIdx.x;
      pre_s[j] = (a.fs >= 0 && i < a.rows) ? a.s[i] : 0;
      pre_o[j] = (a.fo >= 0 && i < a.rows) ? a.o[i] : 0;
    }
  }
  stmt_prologue(pro, a.direct ? -1 : a.fs, a.direct ? -1 : a.fo, consts, want_s, want_o);
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // block-uniform
    int64_t v[kRowsPerThread][3];
    uint32_t hits = 0, n = 0;
    const bool use_pre = pre && tile == blockIdx.x;
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      const uint32_t i = tile * kTileRows + j * 256 + threadIdx.x;
      if (i < a.rows && arm_row(a, consts, want_s, want_o, i, v[j], use_pre, pre_s[j], pre_o[j])) { hits |= 1u << j; ++n; }
    }
    uint32_t before;
    unsigned long long idx = block_claim(n, claim_counter(out, out_count), before) + before;
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j)
      if (hits & (1u << j)) {
        const uint64_t pos = out_row(out, idx);
        if (pos < out.cap) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < a.nsrc) out.col[k][pos] = v[j][k];
        }
        ++idx;
      }
  }
  stmt_epilogue(ep);
}

struct BuildArgs {
  RelDev I;
  int nkey;
  int key_col[2];
  uint32_t* heads;
  uint32_t* next;
  uint32_t hmask;
  int near, near_b;
};
__global__ void __launch_bounds__(256) k_build(BuildArgs b) {
  const uint32_t n = rel_rows(b.I);
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    if (!row_is_there(b.I, r)) continue;
    const int64_t k0 = b.I.col[b.key_col[0]][r];
    const uint32_t h = b.nkey == 2 ? head_of2(k0, b.I.col[b.key_col[1]][r], b.near, b.near_b) : head_of(k0, b.near);
    b.next[r] = atomicExch(&b.heads[h & b.hmask], r);
  }
}

struct ProbeArgs {
  ArmDev arm;
  const int64_t* consts;
  RelDev I;
  const uint32_t* heads;
  const uint32_t* next;
  uint32_t hmask;
  int near, near_b; // see head_of
  int nkey;
  int key_src[2];   // position in the arm's tuple
  int key_col[2];   // column of the intermediate
  int nnew;
  int new_src[3];   // positions of the arm's tuple that become new columns
  RelDev out;       // columns: I's, then the new ones
  unsigned long long* out_count;
  int count_only;
  unsigned long long* small_failed;  // k_probe<.., true>: bit `step` is raised when the intermediate does not fit the LDS table
  int step;
};

// Streams one base-table arm and probes the intermediate's hash table. nkey == 0 is the cross product.
// Per tile: every thread walks the chains of its rows once to count its matches, the block claims its output
// rows with one atomic, and the chains are walked again to write (they are short, and in cache by then).
__device__ __forceinline__ bool probe_match(const ProbeArgs& p, uint32_t r, int64_t k0, int64_t k1) {
  return p.nkey == 0 || (p.I.col[p.key_col[0]][r] == k0 && (p.nkey < 2 || p.I.col[p.key_col[1]][r] == k1));
}
// kLds: the intermediate is expected to be small (a few rows after a constant filter, as in most steps of the LUBM
// statements). Then there is no build kernel and no table in HBM: every block enters the intermediate's keys into a
// chained table of its own in LDS (the rows are read once per block, from cache), and the probes never leave
// the CU. An intermediate that turns out larger than kLdsRows raises the step's bit in *p.small_failed, emits
// nothing, and the host runs the statement again with the table in HBM for that step.
constexpr int kLdsRows = 1024, kLdsHeads = 2048;
template <int kRowsPerThread, bool kLds>
__global__ void __launch_bounds__(256) k_probe(ProbeArgs p, Epilogue ep) {
  constexpr int kTileRows = 256 * kRowsPerThread;
  __shared__ int64_t lds_k0[kLds ? kLdsRows : 1], lds_k1[kLds ? kLdsRows : 1];
  __shared__ uint32_t lds_heads[kLds ? kLdsHeads : 1];
  __shared__ uint16_t lds_next[kLds ? kLdsRows : 1];
  const uint32_t nI = rel_rows(p.I);
  const int64_t want_s = (!p.arm.direct && p.arm.fs >= 0) ? p.consts[p.arm.fs] : 0, want_o = (!p.arm.direct && p.arm.fo >= 0) ? p.consts[p.arm.fo] : 0;
  if (kLds) {
    if (nI > (uint32_t)kLdsRows) {  // block-uniform
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the answer is waited for: the flag has landed before this block arrives at the epilogue's ticket
        const unsigned long long before = atomicOr(p.small_failed, 1ull << p.step);
        asm volatile("" ::"v"(before));
      }
      stmt_epilogue(ep);
      return;
    }
    for (uint32_t i = threadIdx.x; i < (uint32_t)kLdsHeads; i += 256) lds_heads[i] = kNil;
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < nI; r += 256) {
      const int64_t k0 = p.I.col[p.key_col[0]][r], k1 = p.nkey == 2 ? p.I.col[p.key_col[1]][r] : 0;
      lds_k0[r] = k0;
      lds_k1[r] = k1;
      const uint32_t h = p.nkey == 2 ? head_of2(k0, k1, p.near, p.near_b) : head_of(k0, p.near);
      lds_next[r] = (uint16_t)atomicExch(&lds_heads[h & (kLdsHeads - 1)], r);  // kNil's low half marks the end
    }
    __syncthreads();
  }
  auto chain_next = [&](uint32_t r) -> uint32_t {
    if (kLds) { const uint32_t nx = lds_next[r]; return nx == 0xFFFFu ? kNil : nx; }
    return p.next[r];
  };
  auto matches = [&](uint32_t r, int64_t k0, int64_t k1) -> bool {
    if (kLds) return lds_k0[r] == k0 && (p.nkey < 2 || lds_k1[r] == k1);
    return probe_match(p, r, k0, k1);
  };
  const uint32_t ntiles = (p.arm.rows + kTileRows - 1) / kTileRows;
  unsigned long long local = 0;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // block-uniform
    int64_t v[kRowsPerThread][3];
    uint32_t head[kRowsPerThread];
    uint32_t n = 0;
    // With the table in LDS the kernel is a stream over the arm: its rows are fetched side by side. With the table in
    // HBM the probes are what takes the time, and more waves per SIMD (fewer registers) serve them better.
    uint32_t pass = 0;
    if constexpr (kLds) {
      pass = arm_rows<kRowsPerThread>(p.arm, want_s, want_o, tile * kTileRows + threadIdx.x, v);
    } else {
#pragma unroll
      for (int j = 0; j < kRowsPerThread; ++j) {
        const uint32_t i = tile * kTileRows + j * 256 + threadIdx.x;
        if (i < p.arm.rows && arm_row(p.arm, p.consts, want_s, want_o, i, v[j])) pass |= 1u << j;
      }
    }
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      head[j] = kNil;
      if (pass >> j & 1) {
        if (p.nkey == 0) {
          head[j] = nI ? 0u : kNil;
        } else {
          const int64_t k0 = pick3(v[j], p.key_src[0]), k1 = p.nkey == 2 ? pick3(v[j], p.key_src[1]) : 0;
          const uint32_t h = p.nkey == 2 ? head_of2(k0, k1, p.near, p.near_b) : head_of(k0, p.near);
          head[j] = kLds ? lds_heads[h & (kLdsHeads - 1)] : p.heads[h & p.hmask];
        }
      }
    }
    uint32_t hit[kRowsPerThread], nhit[kRowsPerThread];  // first matching row of the intermediate, number of matches
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      hit[j] = kNil;
      nhit[j] = 0;
      if (p.nkey == 0) { nhit[j] = head[j] != kNil ? nI : 0u; n += nhit[j]; continue; }
      const int64_t k0 = pick3(v[j], p.key_src[0]), k1 = p.nkey == 2 ? pick3(v[j], p.key_src[1]) : 0;
      for (uint32_t r = head[j]; r != kNil; r = chain_next(r))
        if (matches(r, k0, k1)) {
          if (!nhit[j]) hit[j] = r;
          ++nhit[j];
        }
      n += nhit[j];
    }
    if (p.count_only) { local += n; continue; }
    uint32_t before;
    unsigned long long idx = block_claim(n, claim_counter(p.out, p.out_count), before) + before;
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      if (!nhit[j]) continue;
      const int64_t k0 = p.nkey ? pick3(v[j], p.key_src[0]) : 0, k1 = p.nkey == 2 ? pick3(v[j], p.key_src[1]) : 0;
      // the usual row has one match, remembered from the counting walk; only several matches walk the chain again
      const bool single = p.nkey != 0 && nhit[j] == 1;
      for (uint32_t r = single ? hit[j] : head[j]; r != kNil; r = single ? kNil : (p.nkey == 0 ? (r + 1 < nI ? r + 1 : kNil) : chain_next(r))) {
        if (!single && !matches(r, k0, k1)) continue;
        const uint64_t pos = out_row(p.out, idx);
        if (pos < p.out.cap) {
          for (int c = 0; c < p.I.ncols; ++c) p.out.col[c][pos] = p.I.col[c][r];
          for (int c = 0; c < p.nnew; ++c) p.out.col[p.I.ncols + c][pos] = pick3(v[j], p.new_src[c]);
        }
        ++idx;
      }
    }
  }
  if (p.count_only) {  // one atomic per block
    __shared__ uint64_t lds64[8];
    uint64_t block_total;
    (void)block_excl_scan_256<uint64_t>(local, lds64, block_total);
    stmt_epilogue(ep, block_total, p.out_count);
    return;
  }
  stmt_epilogue(ep);
}

// ---- the same join with the roles exchanged: the table is built over the arm's rows (those that pass its filters)
// and the intermediate is streamed past it. Chosen when the arm is expected to be much smaller than the intermediate:
// entering a row costs about as much as probing with one, and a table over few rows stays in cache. The chain of
// arm row i continues at next[i]; a link is checked against the arm's own columns.
__device__ __forceinline__ void arm_tuple(const ArmDev& a, int64_t want_s, int64_t want_o, uint32_t i, int64_t v[3]) {  // of a row that passes the filters
  if (a.direct) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < a.nsrc) v[k] = a.d[k][i];
    return;
  }
  bool need_s = false, need_o = false;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < a.nsrc) { need_s |= a.src[k] == 0; need_o |= a.src[k] == 1; }
  const int64_t s = a.fs >= 0 ? want_s : need_s ? a.s[i] : 0, o = a.fo >= 0 ? want_o : need_o ? a.o[i] : 0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < a.nsrc) v[k] = a.src[k] == 0 ? s : a.src[k] == 1 ? o : a.src_val[k];
}
template <int kRowsPerThread>
__global__ void __launch_bounds__(256) k_build_arm(ProbeArgs p, uint32_t* __restrict__ heads, uint32_t* __restrict__ next) {
  constexpr int kTileRows = 256 * kRowsPerThread;
  const int64_t want_s = (!p.arm.direct && p.arm.fs >= 0) ? p.consts[p.arm.fs] : 0, want_o = (!p.arm.direct && p.arm.fo >= 0) ? p.consts[p.arm.fo] : 0;
  const uint32_t ntiles = (p.arm.rows + kTileRows - 1) / kTileRows;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t v[kRowsPerThread][3];
    uint32_t pass = 0;
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      const uint32_t i = tile * kTileRows + j * 256 + threadIdx.x;
      if (i < p.arm.rows && arm_row(p.arm, p.consts, want_s, want_o, i, v[j])) pass |= 1u << j;
    }
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      if (!(pass & (1u << j))) continue;
      const uint32_t i = tile * kTileRows + j * 256 + threadIdx.x;
      const int64_t k0 = pick3(v[j], p.key_src[0]);
      const uint32_t h = p.nkey == 2 ? head_of2(k0, pick3(v[j], p.key_src[1]), p.near, p.near_b) : head_of(k0, p.near);
      next[i] = atomicExch(&heads[h & p.hmask], i);
    }
  }
}
// p.heads / p.next: the table over the arm (k_build_arm). Output rows as k_probe's: the intermediate's columns, then the new ones.
template <int kRowsPerThread>
__global__ void __launch_bounds__(256) k_probe_rev(ProbeArgs p, Epilogue ep) {
  constexpr int kTileRows = 256 * kRowsPerThread;
  const uint32_t nI = rel_rows(p.I);
  const int64_t want_s = (!p.arm.direct && p.arm.fs >= 0) ? p.consts[p.arm.fs] : 0, want_o = (!p.arm.direct && p.arm.fo >= 0) ? p.consts[p.arm.fo] : 0;
  auto matches = [&](const int64_t (&v)[3], int64_t k0, int64_t k1) {
    return pick3(v, p.key_src[0]) == k0 && (p.nkey < 2 || pick3(v, p.key_src[1]) == k1);
  };
  const uint32_t ntiles = (nI + kTileRows - 1) / kTileRows;
  unsigned long long local = 0;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // block-uniform
    int64_t k0[kRowsPerThread], k1[kRowsPerThread];
    uint32_t head[kRowsPerThread], hit[kRowsPerThread], nhit[kRowsPerThread];
    uint32_t n = 0;
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      const uint32_t r = tile * kTileRows + j * 256 + threadIdx.x;
      head[j] = kNil;
      k0[j] = k1[j] = 0;
      if (r < nI && row_is_there(p.I, r)) {
        k0[j] = p.I.col[p.key_col[0]][r];
        if (p.nkey == 2) k1[j] = p.I.col[p.key_col[1]][r];
        const uint32_t h = p.nkey == 2 ? head_of2(k0[j], k1[j], p.near, p.near_b) : head_of(k0[j], p.near);
        head[j] = p.heads[h & p.hmask];
      }
    }
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      hit[j] = kNil;
      nhit[j] = 0;
      for (uint32_t i = head[j]; i != kNil; i = p.next[i]) {
        int64_t v[3];
        arm_tuple(p.arm, want_s, want_o, i, v);
        if (matches(v, k0[j], k1[j])) {
          if (!nhit[j]) hit[j] = i;
          ++nhit[j];
        }
      }
      n += nhit[j];
    }
    if (p.count_only) { local += n; continue; }
    uint32_t before;
    unsigned long long idx = block_claim(n, claim_counter(p.out, p.out_count), before) + before;
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      if (!nhit[j]) continue;
      const uint32_t r = tile * kTileRows + j * 256 + threadIdx.x;
      const bool single = nhit[j] == 1;
      for (uint32_t i = single ? hit[j] : head[j]; i != kNil; i = single ? kNil : p.next[i]) {
        int64_t v[3];
        arm_tuple(p.arm, want_s, want_o, i, v);
        if (!single && !matches(v, k0[j], k1[j])) continue;
        const uint64_t pos = out_row(p.out, idx);
        if (pos < p.out.cap) {
          for (int c = 0; c < p.I.ncols; ++c) p.out.col[c][pos] = p.I.col[c][r];
          for (int c = 0; c < p.nnew; ++c) p.out.col[p.I.ncols + c][pos] = pick3(v, p.new_src[c]);
        }
        ++idx;
      }
    }
  }
  if (p.count_only) {  // one atomic per block
    __shared__ uint64_t lds64[8];
    uint64_t block_total;
    (void)block_excl_scan_256<uint64_t>(local, lds64, block_total);
    stmt_epilogue(ep, block_total, p.out_count);
    return;
  }
  stmt_epilogue(ep);
}

// ---- the same join without a table: the arm is stored in ascending order of its subjects, one of the join's key
// columns is that subject, and the intermediate is small beside the arm. Every row of the intermediate finds the
// first row of its subject in the arm by bisection (the upper levels stay in cache) and walks the rows of that
// subject: some twenty dependent reads per row instead of a pass over the whole arm. s_key: which of the key columns
// is the arm's subject. The arm has no constant on its subject. Output rows as k_probe's.
__global__ void __launch_bounds__(256) k_probe_sorted(ProbeArgs p, int s_key, Epilogue ep) {
  const uint32_t nI = rel_rows(p.I);
  const int64_t want_s = 0, want_o = p.arm.fo >= 0 ? p.consts[p.arm.fo] : 0;
  const int64_t* __restrict__ as = p.arm.s;
  const uint32_t rows = p.arm.rows;
  auto matches = [&](const int64_t (&v)[3], int64_t k0, int64_t k1) {
    return pick3(v, p.key_src[0]) == k0 && (p.nkey < 2 || pick3(v, p.key_src[1]) == k1);
  };
  const uint32_t ntiles = (nI + 255) / 256;
  unsigned long long local = 0;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // block-uniform
    const uint32_t r = tile * 256 + threadIdx.x;
    int64_t k0 = 0, k1 = 0;
    uint32_t lo = rows, n = 0;
    if (r < nI && row_is_there(p.I, r)) {
      k0 = p.I.col[p.key_col[0]][r];
      if (p.nkey == 2) k1 = p.I.col[p.key_col[1]][r];
      const int64_t ks = s_key == 0 ? k0 : k1;
      uint32_t a = 0, b = rows;
      while (a < b) {
        const uint32_t m = a + (b - a) / 2;
        if ((uint64_t)as[m] < (uint64_t)ks) a = m + 1; else b = m;  // the order k_count_changes looked at
      }
      lo = a;
      for (uint32_t i = lo; i < rows && as[i] == ks; ++i) {
        if (p.arm.fo >= 0 && p.arm.o[i] != want_o) continue;
        int64_t v[3];
        arm_tuple(p.arm, want_s, want_o, i, v);
        if (matches(v, k0, k1)) ++n;
      }
    }
    if (p.count_only) { local += n; continue; }
    uint32_t before;
    unsigned long long idx = block_claim(n, claim_counter(p.out, p.out_count), before) + before;
    if (!n) continue;
    const int64_t ks = s_key == 0 ? k0 : k1;
    for (uint32_t i = lo; i < rows && as[i] == ks; ++i) {
      if (p.arm.fo >= 0 && p.arm.o[i] != want_o) continue;
      int64_t v[3];
      arm_tuple(p.arm, want_s, want_o, i, v);
      if (!matches(v, k0, k1)) continue;
      const uint64_t pos = out_row(p.out, idx);
      if (pos < p.out.cap) {
        for (int c = 0; c < p.I.ncols; ++c) p.out.col[c][pos] = p.I.col[c][r];
        for (int c = 0; c < p.nnew; ++c) p.out.col[p.I.ncols + c][pos] = pick3(v, p.new_src[c]);
      }
      ++idx;
    }
  }
  if (p.count_only) {  // one atomic per block
    __shared__ uint64_t lds64[8];
    uint64_t block_total;
    (void)block_excl_scan_256<uint64_t>(local, lds64, block_total);
    stmt_epilogue(ep, block_total, p.out_count);
    return;
  }
  stmt_epilogue(ep);
}

// DX_JOIN_LOOKUP: 0 = never, 1 = whenever the arm allows it; otherwise when the arm has at least 2^20 rows and twice as
// many as the intermediate is expected to hold
static int join_lookup_mode() {
  const char* e = getenv("DX_JOIN_LOOKUP");  // (read per statement: the tests switch it)
  return !e ? -1 : atoi(e);
}

// DX_JOIN_REVERSE: 0 = the table is always built over the intermediate, 1 = over the arm whenever that is possible;
// otherwise over the arm when it is expected to hold at most half as many rows as the intermediate
static int join_reverse_mode() {
  const char* e = getenv("DX_JOIN_REVERSE");  // (read per statement: the tests switch it)
  return !e ? -1 : atoi(e);
}

// DX_JOIN_NEAR_BITS: how many low bits of a join key choose the head inside a block of neighbouring heads (0: none)
static int join_near_bits() {
  static const int bits = [] { const char* e = getenv("DX_JOIN_NEAR_BITS"); const int v = e ? atoi(e) : 8; return v < 0 ? 0 : v > 16 ? 16 : v; }();
  return bits;
}

// DX_JOIN_STRIPE_ROWS: a relation expected to hold at least this many rows is striped when a hash join reads it (0: never)
static uint64_t join_stripe_rows() {
  const char* e = getenv("DX_JOIN_STRIPE_ROWS");  // (read per statement: the tests switch it)
  return e ? (uint64_t)strtoull(e, nullptr, 10) : 65536ull;
}

// DX_JOIN_NEAR_B: how many low bits of a two-column key's second column are left out of the mixing, so that rows whose
// second keys lie close together (ids handed out together) start their chains in the same neighbourhood of heads
static int join_near_b_bits() {
  static const int bits = [] { const char* e = getenv("DX_JOIN_NEAR_B"); const int v = e ? atoi(e) : 8; return v < 0 ? 0 : v > 16 ? 16 : v; }();
  return bits;
}

// ---- planning ----------------------------------------------------------------------------------------
struct PlanRel {
  int ncols = 0;
  int cls[3] = {-1, -1, -1};     // variable class of each projected column
  struct Arm {
    const CatalogTable* t;
    int fs = -1, fo = -1;
    int src[3] = {0, 0, 0}, src_ci[3] = {-1, -1, -1};
    long long src_val[3] = {0, 0, 0};  // ids of the projected constants
  };
  std::vector<Arm> arms;
  uint64_t rows_upper = 0;       // rows of the arms held here: bounds the local scan output
  double rows_plan = 0;          // rows of the arms over all ranks: what the cost model sees
  double est = 0;                // estimated rows after the constant filters
  double dist[3] = {1, 1, 1};    // estimated distinct values per projected column
};

struct Plan {
  std::vector<PlanRel> rels;     // in execution order
  std::vector<std::string> const_str;
  std::vector<long long> const_preset;
  std::vector<int> select_cls;
  std::vector<std::string> select_names;
  std::vector<std::string> virtual_terms;
  std::vector<double> est_rows;  // estimated rows after each step of the chosen order
  int n_classes = 0;
};

int find_root(std::vector<int>& uf, int x) { while (uf[x] != x) x = uf[x] = uf[uf[x]]; return x; }

// may_allocate == false (the entry point that allocates nothing): whatever the plan needs beyond the tables
// themselves -- their distinct counts, the duplicate-free copy of a table read by UNION arms -- has to exist
// already, because gathering it takes device memory from the allocator.
// text_hash != 0: the join order found for this statement text over tables with these statistics is remembered.
Plan make_plan(dx_ctx* ctx, const SqlQuery& q, bool may_allocate = true, uint64_t text_hash = 0) {
  Plan plan;
  const long long kLookup = (-0x7FFFFFFFFFFFFFFFll - 1);
  // constants a filter compares with: looked up in the dictionary on the device (at most kMaxConsts distinct ones)
  auto add_const = [&](const std::string& s) -> int {
    for (size_t i = 0; i < plan.const_str.size(); ++i)
      if (plan.const_str[i] == s) return (int)i;
    plan.const_str.push_back(s);
    plan.const_preset.push_back(kLookup);
    return (int)plan.const_str.size() - 1;
  };
  // constants an arm projects ('<partition-_3=predicate>' AS P in every arm of a variable-predicate UNION): never
  // looked up, they carry ids of their own, -(2 + index), and there may be as many as there are arms
  auto add_virtual = [&](const std::string& s) -> long long {
    for (size_t i = 0; i < plan.virtual_terms.size(); ++i)
      if (plan.virtual_terms[i] == s) return -(long long)(2 + i);
    plan.virtual_terms.push_back(s);
    return -(long long)(2 + plan.virtual_terms.size() - 1);
  };
  // variable classes: union-find over (sub, column)
  std::vector<std::pair<int, int>> slots;
  auto slot_of = [&](const std::string& alias, const std::string& col) -> int {
    for (size_t si = 0; si < q.subs.size(); ++si)
      if (q.subs[si].alias == alias)
        for (size_t c = 0; c < q.subs[si].arms[0].proj.size(); ++c)
          if (q.subs[si].arms[0].proj[c].name == col) {
            for (size_t k = 0; k < slots.size(); ++k)
              if (slots[k].first == (int)si && slots[k].second == (int)c) return (int)k;
          }
    throw DxError(DX_E_MALFORMED, "SQL: unknown column " + alias + "." + col);
  };
  for (size_t si = 0; si < q.subs.size(); ++si) {
    const auto& pr = q.subs[si].arms[0].proj;
    if (pr.empty() || pr.size() > 3) throw DxError(DX_E_UNSUPPORTED, "SQL: sub-select with " + std::to_string(pr.size()) + " columns");
    for (size_t c = 0; c < pr.size(); ++c) {
      for (size_t c2 = 0; c2 < c; ++c2)
        if (pr[c2].name == pr[c].name) throw DxError(DX_E_UNSUPPORTED, "SQL: one variable bound to two columns of a sub-select");
      slots.emplace_back((int)si, (int)c);
    }
    for (size_t sj = 0; sj < si; ++sj)
      if (q.subs[sj].alias == q.subs[si].alias) throw DxError(DX_E_MALFORMED, "SQL: duplicate alias " + q.subs[si].alias);
  }
  std::vector<int> uf(slots.size());
  for (size_t i = 0; i < uf.size(); ++i) uf[i] = (int)i;
  for (const auto& c : q.conds) {
    const int a = find_root(uf, slot_of(c.a_alias, c.a_col)), b = find_root(uf, slot_of(c.b_alias, c.b_col));
    if (a != b) uf[a] = b;
  }
  std::vector<int> cls_of(slots.size(), -1);
  for (size_t i = 0; i < slots.size(); ++i) {
    const int r = find_root(uf, (int)i);
    if (cls_of[r] < 0) cls_of[r] = plan.n_classes++;
    cls_of[i] = cls_of[r];
  }
  if (plan.n_classes > MAXC) throw DxError(DX_E_UNSUPPORTED, "SQL: more than " + std::to_string(MAXC) + " variables");
  for (const auto& it : q.select) {
    plan.select_cls.push_back(cls_of[slot_of(it.alias, it.col)]);
    plan.select_names.push_back(it.name);
  }
  // Distinct counts only matter where there is something to order or to size: a statement over one relation
  // never asks for them, the others get those of all their tables with one wait.
  if (q.subs.size() > 1) {
    std::vector<CatalogTable*> need;
    for (const SqlSub& sub : q.subs)
      for (const SqlArm& arm : sub.arms) {
        auto it = ctx->catalog.find(arm.table);
        if (it == ctx->catalog.end()) continue;
        CatalogTable* t = &it->second;
        if (sub.arms.size() > 1) {
          if (!may_allocate && t->several_sources && t->rows >= 2 && !t->distinct_view)
            throw DxError(DX_E_INVAL, "table " + t->view + " has no duplicate-free copy yet: run the statement through dx_query once");
          t = const_cast<CatalogTable*>(distinct_view_of(ctx, *t));  // the view is owned by its table
        }
        if (t->stats_pending) need.push_back(t);
      }
    if (!may_allocate && !need.empty())
      throw DxError(DX_E_INVAL, "table " + need[0]->view + " has no statistics yet: load it with dx_load_table or dx_load_tables");
    table_statistics(ctx, need);
  }
  // relations
  std::vector<PlanRel> rels;
  size_t slot_i = 0;
  for (size_t si = 0; si < q.subs.size(); ++si) {
    const SqlSub& sub = q.subs[si];
    PlanRel r;
    r.ncols = (int)sub.arms[0].proj.size();
    for (int c = 0; c < r.ncols; ++c) r.cls[c] = cls_of[slot_i + c];
    slot_i += r.ncols;
    for (int c = 0; c < r.ncols; ++c)
      for (int c2 = 0; c2 < c; ++c2)
        if (r.cls[c] == r.cls[c2]) throw DxError(DX_E_UNSUPPORTED, "SQL: join condition between two columns of one sub-select");
    struct Cand { PlanRel::Arm a; bool has_s, has_o; std::string s_const, o_const; };
    std::vector<Cand> cands;
    std::vector<std::string> seen;
    for (const SqlArm& arm : sub.arms) {
      std::string sig = arm.table + "\x02" + (arm.has_s ? arm.s_const : "\x03") + "\x02" + (arm.has_o ? arm.o_const : "\x03");
      for (const auto& p : arm.proj) sig += "\x02" + std::to_string(p.src) + p.constant;
      if (std::find(seen.begin(), seen.end(), sig) != seen.end()) continue;  // UNION is a set union: identical arms collapse
      seen.push_back(sig);
      auto it = ctx->catalog.find(arm.table);
      if (it == ctx->catalog.end()) throw DxError(DX_E_NOENT, "SQL: table or view not found: " + arm.table);
      PlanRel::Arm a;
      // a UNION is a set union: rows a shared directory holds twice come out once (a plain sub-select keeps them)
      if (sub.arms.size() > 1 && !may_allocate && it->second.several_sources && it->second.rows >= 2 && !it->second.distinct_view)
        throw DxError(DX_E_INVAL, "table " + it->second.view + " has no duplicate-free copy yet: run the statement through dx_query once");
      a.t = sub.arms.size() > 1 ? distinct_view_of(ctx, it->second) : &it->second;
      for (int c = 0; c < r.ncols; ++c) {
        a.src[c] = arm.proj[c].src;
        if (a.src[c] == 2) { a.src_ci[c] = -1; a.src_val[c] = add_virtual(arm.proj[c].constant); }
      }
      cands.push_back({a, arm.has_s, arm.has_o, arm.s_const, arm.o_const});
    }
    // SQL's UNION returns every row once. Arms that project different constants cannot produce the same row, and an arm
    // whose row fixes both columns of its (duplicate-free) table produces no row twice: that is every UNION the
    // translator writes ('<directory>' AS P in each arm). Arms that can coincide -- the same constants, or none, over
    // different tables or filters; an arm that leaves a column of its table out -- are replaced, group by group, by the
    // duplicate-free table of their rows (distinct_union_of), which then is an arm like any other.
    if (sub.arms.size() > 1) {
      std::vector<std::vector<size_t>> groups;
      for (size_t i = 0; i < cands.size(); ++i) {
        bool placed = false;
        for (auto& g : groups) {
          const Cand& h = cands[g[0]];
          bool same = true;
          for (int c = 0; c < r.ncols; ++c)
            same &= (h.a.src[c] == 2) == (cands[i].a.src[c] == 2) && (h.a.src[c] != 2 || h.a.src_val[c] == cands[i].a.src_val[c]);
          if (same) { g.push_back(i); placed = true; break; }
        }
        if (!placed) groups.push_back({i});
      }
      std::vector<Cand> replaced;
      for (const auto& g : groups) {
        auto fixes_both = [&](const Cand& x) {
          bool fs = x.has_s, fo = x.has_o;
          for (int c = 0; c < r.ncols; ++c) { fs |= x.a.src[c] == 0; fo |= x.a.src[c] == 1; }
          return fs && fo;
        };
        if (g.size() == 1 && fixes_both(cands[g[0]])) { replaced.push_back(cands[g[0]]); continue; }
        if (ctx->comm.world > 1) throw DxError(DX_E_UNSUPPORTED, "SQL: UNION arms whose rows can coincide, on several GPUs");
        std::string signature;
        std::vector<UnionSource> sources;
        std::vector<std::string> lookups;
        for (size_t i : g) {
          const Cand& x = cands[i];
          UnionSource u;
          u.t = x.a.t;
          u.has_s = x.has_s; u.has_o = x.has_o;
          for (int c = 0; c < r.ncols; ++c)
            if (x.a.src[c] != 2) (u.src_a < 0 ? u.src_a : u.src_b) = x.a.src[c];
          signature += x.a.t->view + "\x02" + std::to_string((uintptr_t)x.a.t) + "\x02" + (x.has_s ? x.s_const : "\x03") + "\x02" + (x.has_o ? x.o_const : "\x03") +
                       "\x02" + std::to_string(u.src_a) + std::to_string(u.src_b) + "\x01";
          if (x.has_s) lookups.push_back(x.s_const);
          if (x.has_o) lookups.push_back(x.o_const);
          sources.push_back(u);
        }
        const CatalogTable* table = nullptr;
        auto known = ctx->union_tables.find(signature);
        if (known != ctx->union_tables.end()) {
          table = known->second.get();
        } else {
          if (!may_allocate) throw DxError(DX_E_INVAL, "a UNION of this statement has no duplicate-free copy yet: run the statement through dx_query once");
          std::vector<int64_t> ids(lookups.size());
          if (!lookups.empty()) {
            std::string bytes;
            std::vector<uint64_t> off{0};
            for (const std::string& t : lookups) { bytes += t; off.push_back(bytes.size()); }
            dict_lookup(ctx, bytes.data(), off.data(), lookups.size(), ids.data());
          }
          size_t at = 0;
          for (size_t k = 0; k < g.size(); ++k) {
            if (sources[k].has_s) sources[k].want_s = ids[at++];
            if (sources[k].has_o) sources[k].want_o = ids[at++];
          }
          table = distinct_union_of(ctx, sources, signature);
        }
        if (table->stats_pending) {
          if (!may_allocate) throw DxError(DX_E_INVAL, "a UNION of this statement has no statistics yet: run the statement through dx_query once");
          table_statistics(ctx, {const_cast<CatalogTable*>(table)});
        }
        Cand y = cands[g[0]];
        y.a.t = table;
        y.has_s = y.has_o = false;
        int next_col = 0;
        for (int c = 0; c < r.ncols; ++c)
          if (y.a.src[c] != 2) y.a.src[c] = next_col++;
        replaced.push_back(y);
      }
      cands = std::move(replaced);
    }
    for (const Cand& cd : cands) {
      PlanRel::Arm a = cd.a;
      if (cd.has_s) a.fs = add_const(cd.s_const);
      if (cd.has_o) a.fo = add_const(cd.o_const);
      r.arms.push_back(a);
      r.rows_upper += a.t->rows;
      // uniform-frequency estimate from the table's row and distinct counts (gathered when it was loaded)
      const CatalogTable& t = *a.t;
      const double n = (double)(t.has_plan_stats ? t.plan_rows : t.rows);
      const double ds = (double)std::max<uint64_t>(t.has_plan_stats ? t.plan_distinct_s : t.distinct_s, 1),
                   dob = (double)std::max<uint64_t>(t.has_plan_stats ? t.plan_distinct_o : t.distinct_o, 1);
      r.rows_plan += n;
      double rows = n;
      if (cd.has_s) rows /= ds;
      if (cd.has_o) rows /= dob;
      rows = std::max(rows, n > 0 ? 1.0 : 0.0);
      r.est += rows;
      for (int c = 0; c < r.ncols; ++c) {
        const double d = a.src[c] == 0 ? (cd.has_s ? 1.0 : std::min(ds, rows)) : a.src[c] == 1 ? (cd.has_o ? 1.0 : std::min(dob, rows)) : 1.0;
        r.dist[c] = (&cd == &cands[0]) ? d : r.dist[c] + d;
      }
    }
    rels.push_back(std::move(r));
  }
  if ((int)plan.const_str.size() > kMaxConsts) throw DxError(DX_E_UNSUPPORTED, "SQL: more than 32 constants");
  // Join order: exhaustive search over left-deep orders (a BGP has a handful of sub-selects) for the one with
  // the smallest sum of estimated intermediate sizes; an order may only take a relation that shares a
  // variable with what is bound, unless none is left that does. Size of I |x| R on shared variables K:
  // |I| * |R| / prod over K of max(distinct_I(v), distinct_R(v)).
  {
    const int n = (int)rels.size();
    std::vector<int> best_order, cur;
    std::vector<double> best_rows, cur_rows;
    double best_cost = 1e300;
    // everything the search below looks at, folded into the key of the remembered order
    uint64_t order_key = 0;
    if (text_hash && n > 2) {
      uint64_t h = text_hash ^ 0x9E3779B97F4A7C15ull * (uint64_t)plan.n_classes;
      auto fold = [&](uint64_t v) { h = mix64(h ^ v); };
      auto fold_d = [&](double d) { uint64_t v; memcpy(&v, &d, 8); fold(v); };
      for (const PlanRel& r : rels) {
        fold((uint64_t)r.ncols);
        for (int c = 0; c < r.ncols; ++c) { fold((uint64_t)r.cls[c]); fold_d(r.dist[c]); }
        fold_d(r.est);
        fold_d(r.rows_plan);
      }
      order_key = h ? h : 1;
      auto it = ctx->q_orders.find(order_key);
      if (it != ctx->q_orders.end() && (int)it->second.order.size() == n) {
        for (int i : it->second.order) plan.rels.push_back(rels[i]);
        plan.est_rows = it->second.rows;
        if ((int)plan.rels.size() > MAXSTEPS) throw DxError(DX_E_UNSUPPORTED, "SQL: more than 30 sub-selects");
        return plan;
      }
    }
    std::vector<bool> used(n, false);
    long budget = 200000;  // nodes of the search tree; beyond it the best order found so far is used
    struct State { double rows; std::vector<double> d; };
    std::function<void(const State&, double)> dfs = [&](const State& st, double cost) {
      if (cost >= best_cost || --budget < 0) return;
      if ((int)cur.size() == n) { best_cost = cost; best_order = cur; best_rows = cur_rows; return; }
      bool any_conn = false;
      for (int i = 0; i < n && !cur.empty(); ++i) {
        if (used[i]) continue;
        for (int c = 0; c < rels[i].ncols; ++c) any_conn |= st.d[rels[i].cls[c]] > 0;
      }
      for (int i = 0; i < n; ++i) {
        if (used[i]) continue;
        const PlanRel& r = rels[i];
        bool conn = false;
        for (int c = 0; c < r.ncols; ++c) conn |= st.d[r.cls[c]] > 0;
        if (!cur.empty() && any_conn && !conn) continue;
        State nx = st;
        if (cur.empty()) {
          nx.rows = r.est;
        } else {
          double rows = st.rows * r.est;
          for (int c = 0; c < r.ncols; ++c)
            if (st.d[r.cls[c]] > 0) rows /= std::max(1.0, std::max(st.d[r.cls[c]], r.dist[c]));
          nx.rows = rows;
        }
        for (int c = 0; c < r.ncols; ++c) {
          const double old = st.d[r.cls[c]];
          nx.d[r.cls[c]] = old > 0 ? std::min(old, r.dist[c]) : r.dist[c];
        }
        for (auto& x : nx.d)
          if (x > 0) x = std::max(1.0, std::min(x, std::max(nx.rows, 1.0)));
        used[i] = true;
        cur.push_back(i);
        cur_rows.push_back(nx.rows);
        // the last step of a count query is not materialised, and every step streams its table once
        dfs(nx, cost + ((int)cur.size() == n ? 0.0 : nx.rows) + 0.05 * r.rows_plan);
        cur.pop_back();
        cur_rows.pop_back();
        used[i] = false;
      }
    };
    State st0{0.0, std::vector<double>(plan.n_classes, 0.0)};
    dfs(st0, 0.0);
    for (int i : best_order) plan.rels.push_back(rels[i]);
    plan.est_rows = best_rows;
    if ((int)plan.rels.size() != n) throw DxError(DX_E_INVAL, "planner found no join order");
    if (order_key) {
      if (ctx->q_orders.size() > 4096) ctx->q_orders.clear();
      ctx->q_orders[order_key] = dx_ctx::JoinOrder{best_order, best_rows};
    }
  }
  if ((int)plan.rels.size() > MAXSTEPS) throw DxError(DX_E_UNSUPPORTED, "SQL: more than 30 sub-selects");
  return plan;
}

ArmDev arm_dev(const PlanRel& r, const PlanRel::Arm& a) {
  ArmDev d{};
  d.s = a.t->s; d.o = a.t->o; d.rows = (uint32_t)a.t->rows;
  d.fs = a.fs; d.fo = a.fo; d.nsrc = r.ncols;
  for (int c = 0; c < 3; ++c) { d.src[c] = a.src[c]; d.src_ci[c] = a.src_ci[c]; d.src_val[c] = a.src_val[c]; }
  return d;
}

unsigned scan_grid(uint64_t rows) {
  uint64_t g = (rows + 255) / 256;
  if (g < 1) g = 1;
  return (unsigned)(g > 2048 ? 2048 : g);
}
unsigned tile_grid(uint64_t rows, int rows_per_thread) {  // kernels that work on tiles of 256 * rows_per_thread rows
  uint64_t g = (rows + 256 * rows_per_thread - 1) / (256 * rows_per_thread);
  if (g < 1) g = 1;
  return (unsigned)(g > 4096 ? 4096 : g);
}
void launch_scan_emit(hipStream_t s, uint64_t rows, const ArmDev& a, const int64_t* consts, const RelDev& out, unsigned long long* cnt,
                      const Prologue& pro = Prologue{}, const Epilogue& ep = Epilogue{}, unsigned min_grid = 1) {
  if (rows <= kSmallTable) hipLaunchKernelGGL(k_scan_emit<1>, dim3(std::max(tile_grid(rows, 1), min_grid)), dim3(256), 0, s, a, consts, out, cnt, pro, ep);
  else hipLaunchKernelGGL(k_scan_emit<4>, dim3(std::max(tile_grid(rows, 4), min_grid)), dim3(256), 0, s, a, consts, out, cnt, pro, ep);
}
void launch_probe(hipStream_t s, uint64_t rows, const ProbeArgs& p, const Epilogue& ep = Epilogue{}, bool small = false) {
  if (small) {
    // (fewer blocks, each entering the intermediate once and walking several tiles, measured no faster: 1024 the same, 512 slower)
    if (rows <= kSmallTable) hipLaunchKernelGGL((k_probe<1, true>), dim3(tile_grid(rows, 1)), dim3(256), 0, s, p, ep);
    else hipLaunchKernelGGL((k_probe<4, true>), dim3(tile_grid(rows, 4)), dim3(256), 0, s, p, ep);
  } else {
    if (rows <= kSmallTable) hipLaunchKernelGGL((k_probe<1, false>), dim3(tile_grid(rows, 1)), dim3(256), 0, s, p, ep);
    else hipLaunchKernelGGL((k_probe<4, false>), dim3(tile_grid(rows, 4)), dim3(256), 0, s, p, ep);
  }
}

}  // namespace

// device block of a context: row counters and tickets (see Epilogue)
// ... and, from a multiple of 256 bytes on, the stripes' row counters of striped relations (see RelDev)
constexpr size_t kStripeCountersAt = (MAXSTEPS + 2 + 1 + kSubTickets + kStripeStride - 1) / kStripeStride * kStripeStride;  // in words
constexpr size_t kClockWordAt = kStripeCountersAt + kStripes * kStripeStride;  // ... and the first kernel's begin on the device's wall clock
constexpr size_t kQueryCounterBytes = sizeof(unsigned long long) * (kClockWordAt + 1);
static_assert(MAXSTEPS <= (int)kStripeStride && MAXSTEPS <= 32, "a word per step between two stripes, a bit per step in Epilogue::striped");
// Pinned host block the final counters land in.
constexpr int kShardedWords = 2 * 64 + 8;
struct QueryHostBlock {
  unsigned long long counters[MAXSTEPS + 2];
  unsigned long long extent[MAXSTEPS + 2];    // striped relations: rows of the buffer in use (see Epilogue::host_extent)
  unsigned long long clock[2];                // the device's wall clock: first kernel's begin, last kernel's end
  unsigned long long seq;                     // number of the statement whose counters these are (written last)
  unsigned long long sharded[kShardedWords];  // several GPUs: row counts per destination rank on their way to the host
};

// The pinned block and the two events a statement uses, made once per context (dx_init): no statement creates
// anything, so the entry point that allocates nothing can share this code.
// The statement's last kernel: the row counters go to the pinned host block by stores from the device, followed
// by the statement's number. The host watches that word instead of waiting for the stream's completion signal
// (a copy engine's transfer plus the runtime's wake-up cost more than the statement's kernels for a short one).
__global__ void __launch_bounds__(64)
k_publish(const unsigned long long* __restrict__ counters, int n, unsigned long long* host_counters, unsigned long long* host_seq,
          unsigned long long seq, const unsigned long long* __restrict__ t_begin, unsigned long long* host_clock) {
  if ((int)threadIdx.x < n) __hip_atomic_store(&host_counters[threadIdx.x], counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x == 63) {
    __hip_atomic_store(&host_clock[0], *t_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host_clock[1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Waits until the pinned block carries statement `seq`: by watching the word for a while, then through the stream
// (which also reports a failed kernel). DX_QUERY_WAIT=stream always waits through the stream.
static void wait_published(dx_ctx* ctx, QueryHostBlock* hb, uint64_t seq) {
  static const bool spin = [] { const char* e = getenv("DX_QUERY_WAIT"); return !(e && std::string(e) == "stream"); }();
  if (spin) {
    const auto t0 = std::chrono::steady_clock::now();
    const volatile unsigned long long* w = &hb->seq;
    for (uint32_t i = 0;; ++i) {
      if (*w == seq) { std::atomic_thread_fence(std::memory_order_acquire); return; }
      __builtin_ia32_pause();
      if ((i & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
  }
  DX_HIP(hipStreamSynchronize(ctx->stream));
  DX_HIP(hipGetLastError());
  if (hb->seq != seq) throw DxError(DX_E_HIP, "the statement's counters did not arrive on the host");
}

double result_device_ms(dx_result* r) { return r->device_ms; }

void query_context_setup(dx_ctx* ctx) {
  if (!ctx->q_host) {
    DX_HIP(hipHostMalloc(&ctx->q_host, sizeof(QueryHostBlock), hipHostMallocDefault));
    memset(ctx->q_host, 0, sizeof(QueryHostBlock));
    DX_HIP(hipHostGetDevicePointer(&ctx->q_host_dev, ctx->q_host, 0));
  }
  if (!ctx->q_counters) {  // row counters + the last-block ticket of a statement: zero between statements
    DX_HIP(hipMalloc(&ctx->q_counters, kQueryCounterBytes));
    DX_HIP(hipMemset(ctx->q_counters, 0, kQueryCounterBytes));
  }
  if (ctx->q_clock_khz <= 0) {  // ticks of the device's wall clock (wall_clock64) per millisecond
    int khz = 0;
    DX_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    ctx->q_clock_khz = khz > 0 ? khz : 100000;
  }
}

static bool plan_cache_enabled() {
  const char* e = getenv("DX_NO_PLAN_CACHE");
  return !(e && *e == '1');
}
// the parsed form of a statement text, parsed once per context
static std::shared_ptr<const SqlQuery> parsed_statement(dx_ctx* ctx, const std::string& sql) {
  if (!plan_cache_enabled()) return std::make_shared<const SqlQuery>(parse_sql(sql));
  auto it = ctx->q_parsed.find(sql);
  if (it != ctx->q_parsed.end()) return it->second;
  auto p = std::make_shared<const SqlQuery>(parse_sql(sql));
  if (ctx->q_parsed.size() > 4096) ctx->q_parsed.clear();
  ctx->q_parsed.emplace(sql, p);
  return p;
}

// `pool` supplies every device buffer of the statement's execution: the context's caching allocator, or a
// fixed arena (then may_allocate is false and nothing else may be allocated either).
static dx_result* run_query_in(dx_ctx* ctx, const std::string& sql, int mode, DevPool& pool, bool may_allocate) {
  const auto t_host0 = std::chrono::steady_clock::now();
  const std::shared_ptr<const SqlQuery> parsed = parsed_statement(ctx, sql);
  const SqlQuery& q = *parsed;
  Plan plan = make_plan(ctx, q, may_allocate, plan_cache_enabled() ? (uint64_t)std::hash<std::string>{}(sql) | 1 : 0);
  const double plan_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count();
  if (!ctx->dict.valid()) throw DxError(DX_E_INVAL, "no dictionary loaded");
  for (auto& r : plan.rels)
    for (auto& a : r.arms)
      if (a.t->rows >= (1ull << 32)) throw DxError(DX_E_RANGE, "table with 2^32 rows or more");
  hipStream_t s = ctx->stream;
  const int nsteps = (int)plan.rels.size();

  ConstArgs ca;
  ca.n = (int)plan.const_str.size();
  uint32_t off = 0;
  for (int i = 0; i < ca.n; ++i) {
    ca.off[i] = off;
    ca.preset[i] = plan.const_preset[i];
    if (off + plan.const_str[i].size() > (size_t)kConstBytes) throw DxError(DX_E_UNSUPPORTED, "SQL: constants exceed 2.5 KiB in total");
    memcpy(ca.bytes + off, plan.const_str[i].data(), plan.const_str[i].size());
    ca.hash[i] = hash_bytes((const uint8_t*)plan.const_str[i].data(), plan.const_str[i].size(), ctx->dict.seed);
    off += (uint32_t)plan.const_str[i].size();
  }
  ca.off[ca.n] = off;

  if (!ctx->q_host || !ctx->q_counters) throw DxError(DX_E_INVAL, "context was not set up for queries");
  QueryHostBlock* hb = static_cast<QueryHostBlock*>(ctx->q_host);

  std::vector<uint64_t> caps(nsteps, 0);
  auto* res = new dx_result();
  res->ctx = ctx;
  res->names = plan.select_names;
  res->virtual_terms = plan.virtual_terms;
  try {
    // The statement's preparation rides in its first kernel and its hand-over in its last (see Prologue / Epilogue)
    // whenever the constants fit the first kernel's arguments; otherwise a kernel each does them.
    // steps whose intermediate did not fit the table in LDS the last time this statement ran: remembered per statement text
    const size_t sql_key = std::hash<std::string>{}(sql);
    uint64_t regular_steps = 0;
    {
      auto it = ctx->q_regular_steps.find(sql_key);
      if (it != ctx->q_regular_steps.end()) regular_steps = it->second;
    }
    // Rows after each step: the planner's estimates, or -- when this statement has run over these tables before --
    // what the steps really produced then. Only sizes and the choice between equivalent ways to run a step depend on
    // them, never a result: a wrong one costs a second attempt (a table that does not fit LDS, a buffer too small).
    std::vector<double> rows_hint = plan.est_rows;
    bool rows_seen = false;
    uint64_t tables_signature = 0x9E3779B97F4A7C15ull * (uint64_t)nsteps;
    for (const auto& r : plan.rels)
      for (const auto& a : r.arms) tables_signature = mix64(tables_signature ^ (uint64_t)a.t->rows ^ (uint64_t)(uintptr_t)a.t->s);
    if (plan_cache_enabled() && !getenv("DX_NO_ROW_FEEDBACK")) {
      auto it = ctx->q_seen_rows.find(sql_key);
      if (it != ctx->q_seen_rows.end() && it->second.signature == tables_signature && (int)it->second.rows.size() == nsteps) {
        rows_hint.assign(it->second.rows.begin(), it->second.rows.end());
        rows_seen = true;
      }
    }
    const bool fused = ca.n <= kProConsts && off <= (uint32_t)kProBytes && ctx->q_counters && !getenv("DX_QUERY_UNFUSED");
    for (int attempt = 0; attempt <= 2 * nsteps + 2; ++attempt) {
      const auto t_attempt = std::chrono::steady_clock::now();
      auto t_laid_out = t_attempt, t_queued = t_attempt;
      if (pool.is_arena()) pool.rewind();  // a re-run with larger buffers starts from the arena's base again
      DevBuf<int64_t> consts(pool, kMaxConsts);
      DevBuf<unsigned long long> own_counters;
      if (!fused) own_counters = DevBuf<unsigned long long>(pool, MAXSTEPS + 2);
      unsigned long long* const counters_p = fused ? static_cast<unsigned long long*>(ctx->q_counters) : own_counters.p;
      std::vector<std::vector<DevBuf<int64_t>>> step_cols(nsteps);
      std::vector<DevBuf<uint32_t>> tmp32;
      // The steps are laid out first (buffers, capacities, hash-table sizes) and launched afterwards: the first
      // launch is handed the prologue, the last one the epilogue.
      using Launch = std::function<void(const Prologue&, const Epilogue&, unsigned)>;
      std::vector<Launch> launches;
      uint64_t heads_words = 0;
      DevBuf<uint32_t> heads_arena;
      RelDev I{};
      std::vector<int> I_cls;
      std::vector<uint64_t> step_cap(nsteps, 0);
      unsigned long long* preset_counter = nullptr;  // fused: the first relation is a table used in place
      unsigned long long preset_value = 0;
      // Which steps join how: known before anything is laid out, because a step's output is striped (see RelDev) only
      // when the step that reads it is a hash join with its table in HBM.
      const bool no_lds_tables = getenv("DX_JOIN_NO_LDS") != nullptr;  // every join's table in HBM (the tests ask for it)
      std::vector<int> step_nkey(nsteps, 0);
      std::vector<char> step_small(nsteps, 0);
      {
        std::vector<int> cls(plan.rels[0].cls, plan.rels[0].cls + plan.rels[0].ncols);
        for (int k = 1; k < nsteps; ++k) {
          const PlanRel& r = plan.rels[k];
          for (int c = 0; c < r.ncols; ++c)
            if (std::find(cls.begin(), cls.end(), r.cls[c]) != cls.end()) ++step_nkey[k];
            else cls.push_back(r.cls[c]);
          // an estimate is trusted up to a quarter of the table's room, a count seen before up to all of it
          step_small[k] = fused && step_nkey[k] > 0 && !(regular_steps & (1ull << k)) && k - 1 < (int)rows_hint.size() &&
                          rows_hint[k - 1] <= (rows_seen ? kLdsRows : kLdsRows / 4) && !no_lds_tables;
        }
      }
      uint32_t striped = 0;
      unsigned long long* const stripes_p = counters_p + kStripeCountersAt;
      auto stripe_output = [&](int k) {
        return fused && k + 1 < nsteps && k < (int)rows_hint.size() && step_nkey[k + 1] > 0 && !step_small[k + 1] &&
               join_stripe_rows() > 0 && rows_hint[k] >= (double)join_stripe_rows();
      };
      // the stripes do not end level: room for the fullest one's lead over the average
      auto striped_cap = [](uint64_t cap) { return cap + cap / 8 + ((uint64_t)kStripes << kStripeGroupBits) * 2; };
      for (int k = 0; k < nsteps; ++k) {
        const PlanRel& r = plan.rels[k];
        const bool last = k == nsteps - 1;
        const bool count_only = last && mode == DX_QUERY_COUNT;
        unsigned long long* cnt = counters_p + k;
        if (k == 0) {
          if (count_only) {
            for (const auto& a : r.arms)
              if (a.t->rows) {
                const ArmDev ad = arm_dev(r, a);
                const uint64_t rows = a.t->rows;
                const int64_t* cp = consts.p;
                launches.push_back([=](const Prologue& pro, const Epilogue& ep, unsigned) {
                  hipLaunchKernelGGL(k_scan_count, dim3(std::min(scan_grid(rows), kCountBlocks)), dim3(256), 0, s, ad, cp, cnt, pro, ep);
                });
              }
            break;
          }
          const bool alias = r.arms.size() == 1 && r.arms[0].fs < 0 && r.arms[0].fo < 0 && r.arms[0].src[0] != 2 &&
                             (r.ncols < 2 || r.arms[0].src[1] != 2) && r.ncols <= 2 && mode == DX_QUERY_COUNT;
          I = RelDev{};
          I.ncols = r.ncols;
          I.count = cnt;
          if (alias) {  // unfiltered single table: use its columns in place
            const CatalogTable* t = r.arms[0].t;
            for (int c = 0; c < r.ncols; ++c) I.col[c] = const_cast<int64_t*>(r.arms[0].src[c] == 0 ? t->s : t->o);
            I.cap = (uint32_t)t->rows;
            if (fused) { preset_counter = cnt; preset_value = t->rows; }
            else launches.push_back([=](const Prologue&, const Epilogue&, unsigned) { DX_HIP(hipMemcpyAsync(cnt, &t->rows, 8, hipMemcpyHostToDevice, s)); });
            step_cap[k] = t->rows;
          } else {
            uint64_t cap = std::max<uint64_t>(caps[k] ? caps[k] : r.rows_upper, 1);
            if (stripe_output(k) && striped_cap(cap) < (1ull << 32)) {
              if (!caps[k]) cap = striped_cap(cap);
              striped |= 1u << k;
              I.stripes = kStripes;
              cnt = stripes_p + k;
              I.count = cnt;
            }
            for (int c = 0; c < r.ncols; ++c) { step_cols[k].emplace_back(pool, cap); I.col[c] = step_cols[k].back().p; }
            I.cap = (uint32_t)cap;
            step_cap[k] = cap;
            for (const auto& a : r.arms)
              if (a.t->rows) {
                const ArmDev ad = arm_dev(r, a);
                const uint64_t rows = a.t->rows;
                const int64_t* cp = consts.p;
                const RelDev Iv = I;
                launches.push_back([=](const Prologue& pro, const Epilogue& ep, unsigned min_grid) { launch_scan_emit(s, rows, ad, cp, Iv, cnt, pro, ep, min_grid); });
              }
          }
          I_cls.assign(r.cls, r.cls + r.ncols);
          continue;
        }
        // join with the running intermediate
        ProbeArgs pa{};
        uint64_t my_heads_off = ~0ull;
        pa.consts = consts.p;
        pa.I = I;
        pa.nkey = 0;
        pa.nnew = 0;
        std::vector<int> new_cls;
        for (int c = 0; c < r.ncols; ++c) {
          int at = -1;
          for (size_t j = 0; j < I_cls.size(); ++j)
            if (I_cls[j] == r.cls[c]) at = (int)j;
          if (at >= 0) {
            if (pa.nkey < 2) { pa.key_src[pa.nkey] = c; pa.key_col[pa.nkey] = at; ++pa.nkey; }
            else throw DxError(DX_E_UNSUPPORTED, "SQL: join on three columns of one sub-select");
          } else {
            pa.new_src[pa.nnew++] = c;
            new_cls.push_back(r.cls[c]);
          }
        }
        if ((int)I_cls.size() + pa.nnew > MAXC) throw DxError(DX_E_UNSUPPORTED, "SQL: intermediate wider than 10 columns");
        // An intermediate the planner expects to be small gets its table in LDS, block by block, inside the probe
        // (k_probe<.., true>); if it is not small after all, the step says so and the statement runs again without.
        const bool small = step_small[k] != 0;
        pa.small_failed = counters_p + MAXSTEPS + 1;
        pa.step = k;
        size_t arms_here = 0;
        const PlanRel::Arm* only_arm = nullptr;
        for (const auto& a : r.arms)
          if (a.t->rows) { ++arms_here; only_arm = &a; }
        // the arm's subject among the join's keys, the arm in ascending order of it, and few rows to look up beside the
        // arm's: no table at all (k_probe_sorted)
        int lookup_key = -1;
        if (pa.nkey > 0 && arms_here == 1 && join_lookup_mode() != 0 && only_arm->t->sorted_s && only_arm->fs < 0) {
          for (int c = 0; c < pa.nkey; ++c)
            if (only_arm->src[pa.key_src[c]] == 0) lookup_key = c;
          const double i_rows = k - 1 < (int)rows_hint.size() ? rows_hint[k - 1] : (double)I.cap;
          // (measured at LUBM-1300 and LUBM-100: from 2^20 rows on, with at most half as many rows to look up; 2^18 or 2^19 rows
          // lose to a pass over so small a table, as many rows to look up as the table holds lose to the hash join)
          if (join_lookup_mode() < 0 && !(only_arm->t->rows >= (1u << 20) && 2.0 * i_rows <= (double)only_arm->t->rows)) lookup_key = -1;
        }
        const bool lookup = lookup_key >= 0;
        bool reverse = false;
        if (pa.nkey > 0 && !small && !lookup && arms_here == 1 && join_reverse_mode() != 0) {
          const double i_est = k - 1 < (int)rows_hint.size() ? rows_hint[k - 1] : (double)I.cap;
          reverse = join_reverse_mode() > 0 || 2.0 * std::min<double>(r.est, (double)only_arm->t->rows) <= i_est;
        }
        if (reverse) {
          const uint64_t arm_rows = only_arm->t->rows;
          const uint64_t expect = std::min<uint64_t>(arm_rows, std::max<uint64_t>((uint64_t)std::min(4.0 * r.est, 4.0e9), arm_rows / 16));
          uint32_t hsize = 1024;
          while (hsize < 2ull * expect && hsize < (1u << 31)) hsize <<= 1;
          my_heads_off = heads_words;
          heads_words += hsize;
          tmp32.emplace_back(pool, std::max<uint64_t>(arm_rows, 1));
          pa.next = tmp32.back().p; pa.hmask = hsize - 1; pa.near = join_near_bits(); pa.near_b = join_near_b_bits();
        } else if (pa.nkey > 0 && !small && !lookup) {
          // Chained table: any number of heads is correct, fewer only means longer chains. Two heads per row
          // that is expected (four times the planner's estimate, to be safe), never fewer than a sixteenth of the
          // rows that could arrive: the heads of all steps are filled before the first scan and read at random by the
          // probes, so a table sized for the buffer's capacity costs time in both when the filters are selective.
          uint64_t expect = I.cap;
          if (k - 1 < (int)rows_hint.size())
            expect = rows_seen ? std::min<uint64_t>(I.cap, (uint64_t)std::min(rows_hint[k - 1] + 1.0, 4.0e9))
                               : std::min<uint64_t>(I.cap, std::max<uint64_t>((uint64_t)std::min(4.0 * rows_hint[k - 1], 4.0e9), I.cap / 16));
          uint32_t hsize = 1024;
          while (hsize < 2ull * expect && hsize < (1u << 31)) hsize <<= 1;
          my_heads_off = heads_words;
          heads_words += hsize;
          tmp32.emplace_back(pool, std::max<uint32_t>(I.cap, 1));
          uint32_t* next = tmp32.back().p;
          BuildArgs ba{};
          ba.I = I; ba.nkey = pa.nkey; ba.key_col[0] = pa.key_col[0]; ba.key_col[1] = pa.key_col[1];
          ba.heads = nullptr; ba.next = next; ba.hmask = hsize - 1; ba.near = join_near_bits(); ba.near_b = join_near_b_bits();
          const uint32_t icap = I.cap;
          const uint64_t off_words = my_heads_off;
          DevBuf<uint32_t>* arena = &heads_arena;
          launches.push_back([=](const Prologue&, const Epilogue&, unsigned) mutable {
            ba.heads = arena->p + off_words;
            hipLaunchKernelGGL(k_build, dim3(scan_grid(icap)), dim3(256), 0, s, ba);
          });
          pa.next = next; pa.hmask = hsize - 1; pa.near = join_near_bits(); pa.near_b = join_near_b_bits();
        }
        pa.count_only = count_only ? 1 : 0;
        pa.out_count = cnt;
        RelDev out{};
        out.ncols = (int)I_cls.size() + pa.nnew;
        out.count = cnt;
        if (!count_only) {
          // optimistic capacity: what flows in, or twice the estimate if that is more; a miss costs one re-run
          const uint64_t guess = std::max<uint64_t>(std::max<uint64_t>(I.cap, r.rows_upper),
                                                    (uint64_t)std::min(2.0 * rows_hint[k], 4.0e8));
          uint64_t cap = std::max<uint64_t>(caps[k] ? caps[k] : guess, 1);
          if (stripe_output(k) && striped_cap(cap) < (1ull << 32)) {
            if (!caps[k]) cap = striped_cap(cap);
            striped |= 1u << k;
            out.stripes = kStripes;
            cnt = stripes_p + k;
            out.count = cnt;
            pa.out_count = cnt;
          }
          if (cap >= (1ull << 32)) throw DxError(DX_E_RANGE, "intermediate result with 2^32 rows or more");
          for (int c = 0; c < out.ncols; ++c) { step_cols[k].emplace_back(pool, cap); out.col[c] = step_cols[k].back().p; }
          out.cap = (uint32_t)cap;
          step_cap[k] = cap;
        }
        pa.out = out;
        if (lookup) {
          pa.arm = arm_dev(r, *only_arm);
          const uint32_t icap = I.cap;
          const ProbeArgs pv = pa;
          const int s_key = lookup_key;
          launches.push_back([=](const Prologue&, const Epilogue& ep, unsigned) {
            hipLaunchKernelGGL(k_probe_sorted, dim3(tile_grid(icap, 1)), dim3(256), 0, s, pv, s_key, ep);
          });
          I = out;
          for (int c : new_cls) I_cls.push_back(c);
          continue;
        }
        if (reverse) {
          pa.arm = arm_dev(r, *only_arm);
          const uint64_t arm_rows = only_arm->t->rows;
          const uint64_t off_words = my_heads_off;
          DevBuf<uint32_t>* arena = &heads_arena;
          uint32_t* next = tmp32.back().p;
          const uint32_t icap = I.cap;
          ProbeArgs pv = pa;
          launches.push_back([=](const Prologue&, const Epilogue&, unsigned) {
            if (arm_rows <= kSmallTable) hipLaunchKernelGGL(k_build_arm<1>, dim3(tile_grid(arm_rows, 1)), dim3(256), 0, s, pv, arena->p + off_words, next);
            else hipLaunchKernelGGL(k_build_arm<4>, dim3(tile_grid(arm_rows, 4)), dim3(256), 0, s, pv, arena->p + off_words, next);
          });
          launches.push_back([=](const Prologue&, const Epilogue& ep, unsigned) mutable {
            pv.heads = arena->p + off_words;
            if (icap <= kSmallTable) hipLaunchKernelGGL(k_probe_rev<1>, dim3(tile_grid(icap, 1)), dim3(256), 0, s, pv, ep);
            else hipLaunchKernelGGL(k_probe_rev<4>, dim3(tile_grid(icap, 4)), dim3(256), 0, s, pv, ep);
          });
          I = out;
          for (int c : new_cls) I_cls.push_back(c);
          continue;
        }
        for (const auto& a : r.arms) {
          if (!a.t->rows) continue;
          pa.arm = arm_dev(r, a);
          const uint64_t rows = a.t->rows;
          const uint64_t off_words = my_heads_off;
          DevBuf<uint32_t>* arena = &heads_arena;
          ProbeArgs pv = pa;
          launches.push_back([=](const Prologue&, const Epilogue& ep, unsigned) mutable {
            if (off_words != ~0ull) pv.heads = arena->p + off_words;
            launch_probe(s, rows, pv, ep, small);
          });
        }
        I = out;
        for (int c : new_cls) I_cls.push_back(c);
      }
      heads_arena = DevBuf<uint32_t>(pool, std::max<uint64_t>(heads_words, 4));
      const uint64_t fill_blocks = (heads_words / 4 + 255) / 256;
      const uint64_t seq = ++ctx->q_seq;
      t_laid_out = std::chrono::steady_clock::now();
      unsigned long long* const t_begin = static_cast<unsigned long long*>(ctx->q_counters) + kClockWordAt;
      if (fused && launches.empty()) {  // nothing to scan at all (every table named is empty): every count is zero
        for (auto& c : hb->counters) c = 0;
        hb->clock[0] = hb->clock[1] = 0;
      } else if (fused) {
        QueryHostBlock* hd = static_cast<QueryHostBlock*>(ctx->q_host_dev);
        if (ctx->q_counters_dirty) DX_HIP(hipMemsetAsync(ctx->q_counters, 0, kQueryCounterBytes, s));
        ctx->q_counters_dirty = true;  // until the last kernel has put them back to zero
        Prologue pro{};
        pro.enabled = 1;
        pro.ca.n = ca.n;
        memcpy(pro.ca.bytes, ca.bytes, off);
        for (int i = 0; i < ca.n; ++i) { pro.ca.off[i] = ca.off[i]; pro.ca.preset[i] = ca.preset[i]; pro.ca.hash[i] = ca.hash[i]; }
        pro.ca.off[ca.n] = ca.off[ca.n];
        pro.ht = ctx->dict.ht.p; pro.mask = ctx->dict.ht_mask; pro.pool = ctx->dict.pool.p; pro.pool_off = ctx->dict.pool_off.p;
        pro.consts = consts.p;
        pro.heads = heads_arena.p; pro.n_heads = heads_words;
        pro.t_begin = t_begin;
        Epilogue ep{};
        ep.enabled = 1;
        ep.counters = counters_p; ep.n = MAXSTEPS + 2;
        ep.ticket = counters_p + MAXSTEPS + 2;
        ep.host_counters = hd->counters; ep.host_seq = &hd->seq; ep.seq = seq;
        ep.stripe_counters = stripes_p; ep.host_extent = hd->extent; ep.striped = striped;
        ep.t_begin = t_begin; ep.host_clock = hd->clock;
        const Prologue no_pro{};
        const Epilogue no_ep{};
        const unsigned min_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(fill_blocks, 1024));
        // A table used in place makes the build of step 1 the first kernel, and a kernel cannot set the heads to
        // "empty" while its other blocks already enter rows: then the preparation is a kernel of its own after all.
        const bool own_init = preset_counter != nullptr;
        if (own_init) {
          const unsigned grid = (unsigned)std::max<uint64_t>(std::max<uint64_t>(ca.n, 1), std::min<uint64_t>(fill_blocks, 2048));
          hipLaunchKernelGGL(k_query_init, dim3(grid), dim3(256), 0, s, ca, ctx->dict.seed, ctx->dict.ht.p, ctx->dict.ht_mask,
                             ctx->dict.pool.p, ctx->dict.pool_off.p, consts.p, counters_p, 0, heads_arena.p, heads_words, preset_counter, preset_value, t_begin);
        }
        for (size_t i = 0; i < launches.size(); ++i)
          launches[i](i == 0 && !own_init ? pro : no_pro, i + 1 == launches.size() ? ep : no_ep, min_grid);
        t_queued = std::chrono::steady_clock::now();
        wait_published(ctx, hb, seq);
        ctx->q_counters_dirty = false;
      } else {
        const unsigned grid = (unsigned)std::max<uint64_t>(std::max<uint64_t>(ca.n, 1), std::min<uint64_t>(fill_blocks, 2048));
        hipLaunchKernelGGL(k_query_init, dim3(grid), dim3(256), 0, s, ca, ctx->dict.seed, ctx->dict.ht.p, ctx->dict.ht_mask,
                           ctx->dict.pool.p, ctx->dict.pool_off.p, consts.p, counters_p, MAXSTEPS + 2, heads_arena.p, heads_words,
                           (unsigned long long*)nullptr, 0ull, t_begin);
        const Prologue no_pro{};
        const Epilogue no_ep{};
        for (auto& f : launches) f(no_pro, no_ep, 1);
        QueryHostBlock* hd = static_cast<QueryHostBlock*>(ctx->q_host_dev);
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, counters_p, MAXSTEPS + 2, hd->counters, &hd->seq, (unsigned long long)seq, t_begin, hd->clock);
        wait_published(ctx, hb, seq);
      }
      if (fused && hb->counters[MAXSTEPS + 1]) {  // some step's intermediate did not fit its table in LDS
        regular_steps |= hb->counters[MAXSTEPS + 1];
        if (ctx->q_regular_steps.size() > 4096) ctx->q_regular_steps.clear();
        ctx->q_regular_steps[sql_key] = regular_steps;
        continue;
      }
      bool overflow = false;
      const bool final_counts_only = mode == DX_QUERY_COUNT;
      for (int k = 0; k < nsteps; ++k) {
        if (k == nsteps - 1 && final_counts_only) break;
        // a striped relation needs room up to its fullest stripe's last group (the same again when the statement is re-run:
        // which block claims from which stripe does not change)
        const uint64_t need = (striped >> k & 1) ? hb->extent[k] : hb->counters[k];
        if (need > step_cap[k]) {
          if (need >= (1ull << 32)) throw DxError(DX_E_RANGE, "intermediate result with 2^32 rows or more");
          caps[k] = need;
          overflow = true;
          break;  // later counters were computed from a truncated input
        }
      }
      if (overflow) continue;
      // from the first kernel's begin to the last kernel's end by the device's own wall clock (no events in the queue:
      // each is a packet of its own before and after the kernels, and a call on the host)
      res->device_ms = hb->clock[1] > hb->clock[0] ? (double)(hb->clock[1] - hb->clock[0]) / ctx->q_clock_khz : 0.0;
      res->count = hb->counters[nsteps - 1];
      res->global_count = res->count;
      if (mode == DX_QUERY_BINDINGS) {
        res->has_bindings = true;
        for (size_t c = 0; c < plan.select_cls.size(); ++c) {
          int at = -1;
          for (size_t j = 0; j < I_cls.size(); ++j)
            if (I_cls[j] == plan.select_cls[c]) at = (int)j;
          if (at < 0) throw DxError(DX_E_MALFORMED, "SQL: selected column is not produced by any sub-select");
          DevBuf<int64_t> col(ctx->pool, res->count);  // outlives the statement's buffers: never from an arena
          if (res->count) DX_HIP(hipMemcpyAsync(col.p, I.col[at], res->count * 8, hipMemcpyDeviceToDevice, s));
          res->cols.push_back(std::move(col));
        }
        DX_HIP(hipStreamSynchronize(s));
      }
      res->plan_us = plan_us;  // the text of the plan is put together when it is asked for
      {
        const auto t_done = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        res->host_us[0] = us(t_attempt, t_laid_out);
        res->host_us[1] = us(t_laid_out, t_queued);
        res->host_us[2] = us(t_queued, t_done);
      }
      if (plan_cache_enabled()) {
        if (ctx->q_seen_rows.size() > 4096) ctx->q_seen_rows.clear();
        dx_ctx::SeenRows& seen = ctx->q_seen_rows[sql_key];
        seen.signature = tables_signature;
        seen.rows.assign(hb->counters, hb->counters + nsteps);
      }
      res->plan_steps.resize(nsteps);
      for (int k = 0; k < nsteps; ++k) {
        const PlanRel& r = plan.rels[k];
        for (const auto& a : r.arms) res->plan_steps[k].arms.push_back({a.t->view, a.t->rows, a.fs >= 0, a.fo >= 0});
        res->plan_steps[k].rows_out = hb->counters[k];
      }
      return res;
    }
    throw DxError(DX_E_RANGE, "query did not converge on buffer sizes");
  } catch (...) {
    delete res;
    throw;
  }
}

dx_result* run_query(dx_ctx* ctx, const std::string& sql, int mode) { return run_query_in(ctx, sql, mode, ctx->pool, true); }

// ---- a count that allocates nothing -----------------------------------------------------------------------
// Everything a counting statement needs on the device beyond the tables themselves -- resolved constants, row
// counters, the intermediate relations' columns, the hash-join heads and chains -- is carved out of one
// file-scope device array instead of the caching allocator, for callers that cannot call hipMalloc (HIP graph
// capture, tight serving loops). Each GPU the library runs on holds its own copy of the array at its own
// address until the process ends; all contexts of a GPU share it, and it keeps the last call's contents (the
// statement's first kernel resets the counters and the heads it will use).
constexpr size_t kNoAllocScratchBytes = 32u << 20;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

// With `device` current: the address of that GPU's copy of the array. In host code the array's name is only a
// handle for this call; kernels, memsets and copies take the pointer returned here.
void* noalloc_scratch_of_current_device(size_t* bytes) {
  void* ptr = nullptr;
  DX_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) throw DxError(DX_E_HIP, "hipGetSymbolAddress returned no address for the scratch array");
  *bytes = kNoAllocScratchBytes;
  return ptr;
}

uint64_t run_query_count_noalloc(dx_ctx* ctx, const std::string& sql) {
  if (!ctx->noalloc_scratch) throw DxError(DX_E_INVAL, "context has no scratch array");
  DevPool arena(ctx->noalloc_scratch, ctx->noalloc_scratch_bytes);
  dx_result* r = run_query_in(ctx, sql, DX_QUERY_COUNT, arena, false);
  const uint64_t n = r->count;
  delete r;
  return n;
}


// ---- several GPUs -------------------------------------------------------------------------------------
// One context per GPU; a table lives on the rank that owns its first-level partition. Per join step the two
// inputs must meet: the filtered rows of the new table (the arm, A) and the running intermediate (I), which
// is always spread over the ranks without repetition. Three ways, chosen alike on every rank:
//   * one side is small (by the row counts the ranks exchange): every rank gets all of it (one all-gather) and
//     joins it with its own share of the other side;
//   * otherwise both sides are redistributed by a hash of the join key (the intermediate only if it is not
//     spread by that key already): one all-to-all per relation with all its columns packed row by row;
//   * a cross product sends the whole arm everywhere.
// Every step opens with one exchange of a few words between the hosts: each rank's row counts, its error state,
// and -- when the planner's estimates say that the step will be a redistribution -- the rows it has for every
// peer, counted ahead of the decision, so that a step costs one host exchange and one or two data collectives.
// This is Spark's Exchange hashpartitioning under SortMergeJoin, and its broadcast of a small side
// (QueryProcessor.scala:136-139; QueryProcessorCold.scala:105), with the node's GPUs as shuffle partitions.
// A rank whose local work fails keeps to the protocol until the next collective, where all ranks learn of it
// and return the same error: nobody is left waiting.
namespace {

constexpr int kMaxWorld = 64;
// rows over all ranks up to which a side is sent to every rank instead of hashed (DX_SMALL_SIDE_ROWS overrides it:
// 0 makes every join a redistribution, which is how the tests reach that path on small data)
uint64_t small_side_rows() {
  const char* e = getenv("DX_SMALL_SIDE_ROWS");
  return e && *e ? strtoull(e, nullptr, 10) : 16384;
}

struct DevRel {  // a relation on this rank: columns + row count on the device; the host's count once it is known
  std::vector<DevBuf<int64_t>> own;
  int64_t* col[MAXC] = {};
  int ncols = 0;
  uint64_t cap = 0;
  uint64_t rows = 0;                  // valid when rows_known
  bool rows_known = false;
  DevBuf<unsigned long long> count;   // device copy of the row count
};

struct KeyCols {
  const int64_t* k0;
  const int64_t* k1;  // nullptr for a one-column key
  const unsigned long long* count;
  uint32_t cap;
  int world;
};
__device__ __forceinline__ uint32_t key_owner(const KeyCols& k, uint32_t i) {
  const uint64_t h = k.k1 ? hash_key2(k.k0[i], k.k1[i]) : hash_key(k.k0[i]);
  return (uint32_t)((h >> 32) % (uint32_t)k.world);  // the join's hash table takes the low bits
}
// dest[i] = rank that owns row i's join key; counts[p] += rows going to rank p
__global__ void __launch_bounds__(256)
k_shuffle_count(KeyCols k, uint8_t* __restrict__ dest, unsigned long long* __restrict__ counts) {
  __shared__ uint32_t c[kMaxWorld];
  if (threadIdx.x < kMaxWorld) c[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long n64 = *k.count;
  const uint32_t rows = n64 < k.cap ? (uint32_t)n64 : k.cap;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) {
    const uint32_t d = key_owner(k, i);
    dest[i] = (uint8_t)d;
    atomicAdd(&c[d], 1u);
  }
  __syncthreads();
  if (threadIdx.x < k.world && c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)c[threadIdx.x]);
}

struct RowCols {
  const int64_t* in[MAXC];
  int ncols;
};
// Rows grouped by destination and packed row by row (ncols consecutive words per row): what one all-to-all
// carries. Positions come from one atomic per wave and destination present in it.
__global__ void __launch_bounds__(256)
k_shuffle_scatter(RowCols rc, const uint8_t* __restrict__ dest, uint32_t rows, int world, unsigned long long* __restrict__ cursor,
                  int64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t rounds = (rows + stride - 1) / stride;
  for (uint32_t rd = 0; rd < rounds; ++rd) {
    const uint32_t i = rd * stride + blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < rows;
    const uint32_t d = valid ? dest[i] : 0xFFFFFFFFu;
    unsigned long long pos = 0;
    for (int p = 0; p < world; ++p) {  // wave-uniform
      const uint64_t b = __ballot(d == (uint32_t)p);
      if (!b) continue;
      const int leader = __ffsll((unsigned long long)b) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(&cursor[p], (unsigned long long)__popcll(b));
      base = __shfl(base, leader, 64);
      if (d == (uint32_t)p) pos = base + __popcll(b & lt);
    }
    if (valid)
      for (int c = 0; c < rc.ncols; ++c) out[pos * rc.ncols + c] = rc.in[c][i];
  }
}
// columns -> packed rows (first `rows` rows), and back
__global__ void __launch_bounds__(256) k_pack_rows(RowCols rc, uint32_t rows, int64_t* __restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x)
    for (int c = 0; c < rc.ncols; ++c) out[(uint64_t)i * rc.ncols + c] = rc.in[c][i];
}
struct ColsOut {
  int64_t* out[MAXC];
  int ncols;
};
__global__ void __launch_bounds__(256) k_unpack_rows(const int64_t* __restrict__ in, uint32_t rows, ColsOut co) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x)
    for (int c = 0; c < co.ncols; ++c) co.out[c][i] = in[(uint64_t)i * co.ncols + c];
}
__global__ void k_set_count(unsigned long long* p, unsigned long long v) { *p = v; }

void comm_check(int rc, const char* what) {
  if (rc != 0) throw DxError(DX_E_IO, std::string("collective failed: ") + what);
}

struct ShardedRun {
  dx_ctx* ctx;
  const dx_comm& cm;
  hipStream_t s;
  DevPool& pool;
  LocalError err;  // this rank's outcome so far
  int W;
  uint64_t* pinned;  // the context's pinned block (counters of a statement)
  int n_host_exchanges = 0, n_data_collectives = 0;

  explicit ShardedRun(dx_ctx* c) : ctx(c), cm(c->comm), s(c->stream), pool(c->pool), W(c->comm.world) {
    pinned = reinterpret_cast<uint64_t*>(static_cast<QueryHostBlock*>(ctx->q_host)->sharded);
  }
  void before_collective() {  // a communicator that works outside the context's stream must find the buffers complete
    if (!cm.stream_ordered) DX_HIP(hipStreamSynchronize(s));
  }
  RowCols row_cols(const DevRel& r) const {
    RowCols rc{};
    rc.ncols = r.ncols;
    for (int c = 0; c < r.ncols; ++c) rc.in[c] = r.col[c];
    return rc;
  }
  DevRel make_rel(int ncols, uint64_t cap) {
    DevRel r;
    r.ncols = ncols;
    r.cap = std::max<uint64_t>(cap, 1);
    if (r.cap >= (1ull << 32)) throw DxError(DX_E_RANGE, "relation with 2^32 rows or more on one GPU");
    for (int c = 0; c < ncols; ++c) { r.own.emplace_back(pool, r.cap); r.col[c] = r.own.back().p; }
    r.count = DevBuf<unsigned long long>(pool, 1);
    return r;
  }
  void set_rows(DevRel& r, uint64_t rows) {
    r.rows = rows;
    r.rows_known = true;
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, s, r.count.p, (unsigned long long)rows);  // the kernels read the count on the device
  }

  // Every rank gets all rows of `in` (row counts of all ranks known). Collective.
  DevRel allgather_rel(const DevRel& in, const std::vector<uint64_t>& rows_by_rank) {
    uint64_t total = 0;
    std::vector<uint64_t> counts(W);
    for (int r = 0; r < W; ++r) { counts[r] = rows_by_rank[r] * in.ncols; total += rows_by_rank[r]; }
    if (total >= (1ull << 32)) throw DxError(DX_E_RANGE, "broadcast of 2^32 rows or more");  // same totals on every rank
    DevBuf<int64_t> send(pool, std::max<uint64_t>(in.rows * in.ncols, 1)), recv(pool, std::max<uint64_t>(total * in.ncols, 1));
    if (in.rows) hipLaunchKernelGGL(k_pack_rows, dim3(scan_grid(in.rows)), dim3(256), 0, s, row_cols(in), (uint32_t)in.rows, send.p);
    before_collective();
    if (!cm.allgather) throw DxError(DX_E_INVAL, "the communicator has no allgather");
    comm_check(cm.allgather(cm.user, send.p, in.rows * in.ncols, recv.p, counts.data(), 8), "allgather of a relation");
    ++n_data_collectives;
    DevRel full = make_rel(in.ncols, total);
    ColsOut co{};
    co.ncols = in.ncols;
    for (int c = 0; c < in.ncols; ++c) co.out[c] = full.col[c];
    if (total) hipLaunchKernelGGL(k_unpack_rows, dim3(scan_grid(total)), dim3(256), 0, s, recv.p, (uint32_t)total, co);
    set_rows(full, total);
    return full;
  }

  // Rows of `in` to the ranks `dest` names: send_counts from this rank's count kernel, recv_counts from the peers'.
  DevRel exchange_rel(const DevRel& in, const uint8_t* dest, const std::vector<uint64_t>& send_rows, const std::vector<uint64_t>& recv_rows) {
    uint64_t n_out = 0, n_in = 0;
    std::vector<uint64_t> sc(W), rc(W);
    std::vector<unsigned long long> cur(W);
    for (int p = 0; p < W; ++p) { cur[p] = n_out; n_out += send_rows[p]; n_in += recv_rows[p]; sc[p] = send_rows[p] * in.ncols; rc[p] = recv_rows[p] * in.ncols; }
    DevBuf<int64_t> send(pool, std::max<uint64_t>(n_out * in.ncols, 1)), recv(pool, std::max<uint64_t>(n_in * in.ncols, 1));
    DevBuf<unsigned long long> cursor(pool, kMaxWorld);
    memcpy(pinned, cur.data(), 8 * W);
    DX_HIP(hipMemcpyAsync(cursor.p, pinned, 8 * W, hipMemcpyHostToDevice, s));
    if (n_out) hipLaunchKernelGGL(k_shuffle_scatter, dim3(scan_grid(n_out)), dim3(256), 0, s, row_cols(in), dest, (uint32_t)n_out, W, cursor.p, send.p);
    before_collective();
    comm_check(cm.exchange(cm.user, send.p, sc.data(), recv.p, rc.data(), 8), "exchange of a relation");
    ++n_data_collectives;
    DevRel out = make_rel(in.ncols, n_in);
    ColsOut co{};
    co.ncols = in.ncols;
    for (int c = 0; c < in.ncols; ++c) co.out[c] = out.col[c];
    if (n_in) hipLaunchKernelGGL(k_unpack_rows, dim3(scan_grid(n_in)), dim3(256), 0, s, recv.p, (uint32_t)n_in, co);
    set_rows(out, n_in);
    return out;
  }
};

}  // namespace

// The plan of run_query, executed by all ranks together (every rank passes the same statement).
dx_result* run_query_sharded(dx_ctx* ctx, const std::string& sql, int mode) {
  const dx_comm& cm = ctx->comm;
  if (cm.world > kMaxWorld) throw DxError(DX_E_RANGE, "more than 64 ranks");
  if (!ctx->q_host) throw DxError(DX_E_INVAL, "context was not set up for queries");
  // The statement text is the same on every rank, so a parse error is every rank's error: no collective needed.
  const SqlQuery q = parse_sql(sql);
  const int nsteps = (int)q.subs.size();
  if (nsteps > MAXSTEPS) throw DxError(DX_E_UNSUPPORTED, "SQL: more than 30 sub-selects");
  const int W = cm.world;
  // ---- every table of the statement in partitions of one rank: that rank answers alone
  // A table's name carries its first-level partition ("table00003__3_E_..."), and partition q is kept by rank q % world
  // (dx_load_tables_sharded; the reference's CustomPartitioner, DAP.scala:541-551), so every rank reads the same owner
  // off the statement's text. The partitioning exists to make this the usual case: most statements of a workload touch
  // one first-level partition, no rows move at all, and the only collective is the sum that brings the count (and the
  // owner's error state) to the other ranks.
  {
    int owner = -1;
    bool single = true;
    for (const SqlSub& sub : q.subs)
      for (const SqlArm& arm : sub.arms) {
        const std::string& t = arm.table;
        bool ok = t.size() > 10 && t.compare(0, 5, "table") == 0;
        long part = 0;
        for (int i = 5; ok && i < 10; ++i) { ok = t[i] >= '0' && t[i] <= '9'; part = part * 10 + (t[i] - '0'); }
        if (!ok) { single = false; continue; }
        const int o = (int)(part % W);
        if (owner < 0) owner = o;
        else if (o != owner) single = false;
      }
    if (single && owner >= 0 && !getenv("DX_NO_SINGLE_OWNER")) {
      LocalError err;
      dx_result* res = nullptr;
      // The owner is read off the tables' names; tables placed otherwise (dx_load_table + dx_declare_remote_table with
      // another rule than partition % world) would be scanned as empty here: the owner refuses, and the sum below
      // brings its refusal to every rank.
      if (cm.rank == owner)
        run_local(err, [&] {
          for (const SqlSub& sub : q.subs)
            for (const SqlArm& arm : sub.arms) {
              auto it = ctx->catalog.find(arm.table);
              if (it != ctx->catalog.end() && it->second.remote)
                throw DxError(DX_E_INVAL, "table " + arm.table + " belongs to partition " + arm.table.substr(5, 5) + ", which rank " + std::to_string(owner) +
                                              " of " + std::to_string(W) + " owns (partition % ranks), but it is declared as held by another rank");
            }
          res = run_query(ctx, sql, mode);
        });
      try {
        uint64_t fin[1 + kErrSlots] = {0};
        fin[0] = (err || !res) ? 0 : res->count;
        if (err) fin[1 + err_slot(err.code)] = 1;
        comm_check(cm.allreduce_sum(cm.user, fin, 1 + kErrSlots), "allreduce_sum");
        std::vector<int> codes;
        for (int sl = 0; sl < kErrSlots; ++sl)
          if (fin[1 + sl]) codes.push_back(err_code_of_slot(sl));
        throw_if_any_failed(ctx, err, codes);
        if (!res) {  // not the owner: no rows here
          res = new dx_result();
          res->ctx = ctx;
          for (const auto& it : q.select) res->names.push_back(it.name);
          res->has_bindings = mode == DX_QUERY_BINDINGS;
          if (res->has_bindings)
            for (size_t c = 0; c < q.select.size(); ++c) res->cols.emplace_back(ctx->pool, 1);
          res->plan = "answered by rank " + std::to_string(owner) + " alone: all tables in its partitions\ncollectives: 0 host exchanges, 0 data, 1 sum\n";
        } else {
          res->plan += "answered by this rank alone: all tables in its partitions\ncollectives: 0 host exchanges, 0 data, 1 sum\n";
        }
        res->global_count = fin[0];
        return res;
      } catch (...) {
        delete res;
        throw;
      }
    }
  }
  ShardedRun run(ctx);
  LocalError& err = run.err;
  hipStream_t s = ctx->stream;
  DevPool& pool = ctx->pool;
  const uint64_t kSmallSideRows = small_side_rows();
  // From here on a failure may be this rank's alone (a table it was not told about, a buffer that does not fit):
  // it is carried to the next collective instead of thrown.
  Plan plan;
  ConstArgs ca{};
  DevBuf<int64_t> consts(pool, kMaxConsts);
  run_local(err, [&] {
    plan = make_plan(ctx, q);
    if (!ctx->dict.valid()) throw DxError(DX_E_INVAL, "no dictionary loaded");
    ca.n = (int)plan.const_str.size();
    uint32_t off = 0;
    for (int i = 0; i < ca.n; ++i) {
      ca.off[i] = off;
      ca.preset[i] = plan.const_preset[i];
      if (off + plan.const_str[i].size() > (size_t)kConstBytes) throw DxError(DX_E_UNSUPPORTED, "SQL: constants exceed 2.5 KiB in total");
      memcpy(ca.bytes + off, plan.const_str[i].data(), plan.const_str[i].size());
      ca.hash[i] = hash_bytes((const uint8_t*)plan.const_str[i].data(), plan.const_str[i].size(), ctx->dict.seed);
      off += (uint32_t)plan.const_str[i].size();
    }
    ca.off[ca.n] = off;
    if (ca.n)
      hipLaunchKernelGGL(k_resolve_consts, dim3(ca.n), dim3(64), 0, s, ca, ctx->dict.seed, ctx->dict.ht.p, ctx->dict.ht_mask,
                         ctx->dict.pool.p, ctx->dict.pool_off.p, consts.p);
  });
  // filtered, projected rows of a sub-select's arms that live here; the row count stays on the device
  auto materialise = [&](const PlanRel& r) {
    DevRel a = run.make_rel(r.ncols, r.rows_upper);
    RelDev out{};
    out.ncols = r.ncols;
    out.cap = (uint32_t)a.cap;
    out.count = a.count.p;
    for (int c = 0; c < r.ncols; ++c) out.col[c] = a.col[c];
    DX_HIP(hipMemsetAsync(a.count.p, 0, 8, s));
    for (const auto& arm : r.arms)
      if (arm.t->rows) launch_scan_emit(s, arm.t->rows, arm_dev(r, arm), consts.p, out, a.count.p);
    return a;
  };
  auto read_rows = [&](DevRel& r) {  // row count to the host (clamped to the buffer; the caller checks for overflow itself)
    DX_HIP(hipMemcpyAsync(&run.pinned[0], r.count.p, 8, hipMemcpyDeviceToHost, s));
    DX_HIP(hipStreamSynchronize(s));
    DX_HIP(hipGetLastError());
    r.rows = run.pinned[0];
    r.rows_known = true;
  };
  auto* res = new dx_result();
  res->ctx = ctx;
  try {
    std::ostringstream explain;
    DevRel I;
    std::vector<int> I_cls, I_spread;  // classes of I's columns; classes I is spread by (sorted; empty = as it fell)
    uint64_t local_count = 0;
    for (int k = 0; k < nsteps; ++k) {
      const bool last = k == nsteps - 1;
      const bool count_only = last && mode == DX_QUERY_COUNT;
      DevRel A;
      const PlanRel* r = err ? nullptr : &plan.rels[k];
      run_local(err, [&] {
        explain << (k ? "join " : "scan ");
        for (size_t a = 0; a < r->arms.size(); ++a)
          explain << (a ? " UNION " : "") << r->arms[a].t->view << (r->arms[a].t->remote ? "[remote" : "[local") << ", " << r->arms[a].t->rows << " rows]";
        A = materialise(*r);
      });
      if (k == 0) {
        run_local(err, [&] {
          I = std::move(A);
          I_cls.assign(r->cls, r->cls + r->ncols);
          if (nsteps == 1) { read_rows(I); local_count = I.rows; explain << " -> " << I.rows << " rows here\n"; }
        });
        continue;
      }
      // ---- what joins what
      ProbeArgs pa{};
      std::vector<int> new_cls, key_cls;
      run_local(err, [&] {
        pa.consts = consts.p;
        for (int c = 0; c < r->ncols; ++c) {
          int at = -1;
          for (size_t j = 0; j < I_cls.size(); ++j)
            if (I_cls[j] == r->cls[c]) at = (int)j;
          if (at >= 0) {
            if (pa.nkey < 2) { pa.key_src[pa.nkey] = c; pa.key_col[pa.nkey] = at; ++pa.nkey; key_cls.push_back(r->cls[c]); }
            else throw DxError(DX_E_UNSUPPORTED, "SQL: join on three columns of one sub-select");
          } else {
            pa.new_src[pa.nnew++] = c;
            new_cls.push_back(r->cls[c]);
          }
        }
        if ((int)I_cls.size() + pa.nnew > MAXC) throw DxError(DX_E_UNSUPPORTED, "SQL: intermediate wider than 10 columns");
        // both sides hash the key columns in the same order: ascending variable class
        if (pa.nkey == 2 && key_cls[0] > key_cls[1]) {
          std::swap(key_cls[0], key_cls[1]);
          std::swap(pa.key_src[0], pa.key_src[1]);
          std::swap(pa.key_col[0], pa.key_col[1]);
        }
        if (!I.rows_known) read_rows(I);
      });
      // ---- bring the two sides together. The opening exchange is the same on every path, so a rank that has
      // failed (it may not even have a plan) can take part in it and say so.
      const double estA = err ? 0 : r->est, estI = err ? 0 : (k - 1 < (int)plan.est_rows.size() ? plan.est_rows[k - 1] : 1e18);
      const bool cross = !err && pa.nkey == 0;
      const bool expect_shuffle = !err && !cross && estA > (double)kSmallSideRows && estI > (double)kSmallSideRows;
      const bool move_I = !err && I_spread != key_cls;
      DevBuf<uint8_t> destA(pool, err ? 1 : std::max<uint64_t>(A.cap, 1)), destI(pool, (err || !move_I) ? 1 : std::max<uint64_t>(I.rows, 1));
      DevBuf<unsigned long long> counts(pool, 2 * kMaxWorld + 1);
      std::vector<uint64_t> sendA(W, 0), sendI(W, 0), recvA(W, 0), recvI(W, 0);
      auto count_destinations = [&](bool also_rows_of_A) {  // queues the count kernels and brings their results to the host
        DX_HIP(hipMemsetAsync(counts.p, 0, sizeof(unsigned long long) * (2 * kMaxWorld + 1), s));
        KeyCols ka{A.col[pa.key_src[0]], pa.nkey == 2 ? A.col[pa.key_src[1]] : nullptr, A.count.p, (uint32_t)A.cap, W};
        hipLaunchKernelGGL(k_shuffle_count, dim3(scan_grid(A.rows_known ? A.rows : A.cap)), dim3(256), 0, s, ka, destA.p, counts.p);
        if (move_I && I.rows) {
          KeyCols ki{I.col[pa.key_col[0]], pa.nkey == 2 ? I.col[pa.key_col[1]] : nullptr, I.count.p, (uint32_t)I.cap, W};
          hipLaunchKernelGGL(k_shuffle_count, dim3(scan_grid(I.rows)), dim3(256), 0, s, ki, destI.p, counts.p + kMaxWorld);
        }
        if (also_rows_of_A) DX_HIP(hipMemcpyAsync(counts.p + 2 * kMaxWorld, A.count.p, 8, hipMemcpyDeviceToDevice, s));
        DX_HIP(hipMemcpyAsync(run.pinned, counts.p, sizeof(unsigned long long) * (2 * kMaxWorld + 1), hipMemcpyDeviceToHost, s));
        DX_HIP(hipStreamSynchronize(s));
        DX_HIP(hipGetLastError());
        for (int p = 0; p < W; ++p) { sendA[p] = run.pinned[p]; sendI[p] = run.pinned[kMaxWorld + p]; }
        if (also_rows_of_A) { A.rows = run.pinned[2 * kMaxWorld]; A.rows_known = true; }
      };
      bool counted = false;
      run_local(err, [&] {
        if (std::max<uint64_t>(A.cap, I.rows) >= (1ull << 32)) throw DxError(DX_E_RANGE, "relation of 2^32 rows or more on one GPU");
        if (expect_shuffle) { count_destinations(true); counted = true; }
        else read_rows(A);
      });
      DevRel A2, I2;
      bool have_A2 = false, have_I2 = false;
      {
        // per peer p: [0] rows of my arm, [1] rows of my intermediate, [2] my error code, [3] 1 = words 4 and 5 are valid,
        // [4] rows of my arm for p, [5] rows of my intermediate for p
        std::vector<uint64_t> send(6 * W), recv(6 * W);
        for (int p = 0; p < W; ++p) {
          send[6 * p] = err ? 0 : A.rows;
          send[6 * p + 1] = err ? 0 : I.rows;
          send[6 * p + 2] = (uint64_t)(err ? -err.code : 0);
          send[6 * p + 3] = counted ? 1 : 0;
          send[6 * p + 4] = sendA[p];
          send[6 * p + 5] = sendI[p];
        }
        comm_check(cm.exchange_host(cm.user, send.data(), recv.data(), 6), "exchange of row counts");
        ++run.n_host_exchanges;
        std::vector<int> codes(W);
        std::vector<uint64_t> a_rows(W), i_rows(W);
        uint64_t a_tot = 0, i_tot = 0, inA = 0, inI = 0;
        bool all_counted = true;
        for (int p = 0; p < W; ++p) {
          a_rows[p] = recv[6 * p]; i_rows[p] = recv[6 * p + 1]; codes[p] = -(int)recv[6 * p + 2];
          all_counted &= recv[6 * p + 3] != 0;
          recvA[p] = recv[6 * p + 4]; recvI[p] = recv[6 * p + 5];
          a_tot += a_rows[p]; i_tot += i_rows[p]; inA += recvA[p]; inI += recvI[p];
        }
        throw_if_any_failed(ctx, err, codes);  // from here on every rank has a plan and no failure so far
        // the exact totals decide (the estimates only said whether counting ahead was worth it)
        if (cross || (a_tot <= kSmallSideRows && a_tot <= i_tot)) {
          A2 = run.allgather_rel(A, a_rows);
          have_A2 = true;
        } else if (i_tot <= kSmallSideRows) {
          I2 = run.allgather_rel(I, i_rows);
          have_I2 = true;
          I_spread.clear();  // the result falls where the arm's rows are
        } else {
          // ---- redistribution by the join key
          if (!all_counted) {  // the estimates were off (the same on every rank): count now, exchange the counts
            run_local(err, [&] { count_destinations(false); });
            std::vector<uint64_t> send2(3 * W), recv2(3 * W);
            for (int p = 0; p < W; ++p) { send2[3 * p] = sendA[p]; send2[3 * p + 1] = sendI[p]; send2[3 * p + 2] = (uint64_t)(err ? -err.code : 0); }
            comm_check(cm.exchange_host(cm.user, send2.data(), recv2.data(), 3), "exchange of shuffle counts");
            ++run.n_host_exchanges;
            inA = inI = 0;
            for (int p = 0; p < W; ++p) { recvA[p] = recv2[3 * p]; recvI[p] = recv2[3 * p + 1]; codes[p] = -(int)recv2[3 * p + 2]; inA += recvA[p]; inI += recvI[p]; }
            throw_if_any_failed(ctx, err, codes);
          }
          // Sizes every rank can check for every rank would need the whole matrix of counts; a rank whose own inbox
          // is too large stops alone here, and its peers' next collective fails instead of waiting (documented limit).
          if (inA >= (1ull << 32) || inI >= (1ull << 32)) throw DxError(DX_E_RANGE, "2^32 rows or more arrive on one GPU");
          A2 = run.exchange_rel(A, destA.p, sendA, recvA);
          have_A2 = true;
          if (move_I) {
            I2 = run.exchange_rel(I, destI.p, sendI, recvI);
            have_I2 = true;
            I_spread = key_cls;
          }
        }
      }
      if (!have_A2) { A2 = std::move(A); }
      A = DevRel();
      if (have_I2) I = std::move(I2);
      // ---- local join: I is the build side, the arm's rows are streamed
      DevRel out;
      run_local(err, [&] {
        RelDev Id{};
        Id.ncols = I.ncols;
        Id.cap = (uint32_t)I.rows;
        Id.count = I.count.p;
        for (int c = 0; c < I.ncols; ++c) Id.col[c] = I.col[c];
        pa.I = Id;
        std::vector<DevBuf<uint32_t>> tmp32;
        if (pa.nkey > 0) {
          uint32_t hsize = 1024;
          while (hsize < 2ull * I.rows && hsize < (1u << 31)) hsize <<= 1;
          tmp32.emplace_back(pool, hsize);
          uint32_t* heads = tmp32.back().p;
          tmp32.emplace_back(pool, std::max<uint64_t>(I.rows, 1));
          uint32_t* next = tmp32.back().p;
          DX_HIP(hipMemsetAsync(heads, 0xFF, (size_t)hsize * 4, s));
          BuildArgs ba{};
          ba.I = Id; ba.nkey = pa.nkey; ba.key_col[0] = pa.key_col[0]; ba.key_col[1] = pa.key_col[1];
          ba.heads = heads; ba.next = next; ba.hmask = hsize - 1; ba.near = join_near_bits(); ba.near_b = join_near_b_bits();
          if (I.rows) hipLaunchKernelGGL(k_build, dim3(scan_grid(I.rows)), dim3(256), 0, s, ba);
          pa.heads = heads; pa.next = next; pa.hmask = hsize - 1; pa.near = join_near_bits(); pa.near_b = join_near_b_bits();
        }
        ArmDev ad{};
        ad.rows = (uint32_t)A2.rows;
        ad.fs = ad.fo = -1;
        ad.nsrc = r->ncols;
        ad.direct = 1;
        for (int c = 0; c < r->ncols; ++c) ad.d[c] = A2.col[c];
        pa.arm = ad;
        pa.count_only = count_only ? 1 : 0;
        uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(I.rows, A2.rows), 1);
        for (int attempt = 0;; ++attempt) {
          if (count_only) { out = DevRel(); out.count = DevBuf<unsigned long long>(pool, 1); out.ncols = I.ncols + pa.nnew; }
          else out = run.make_rel(I.ncols + pa.nnew, cap);
          RelDev od{};
          od.ncols = out.ncols;
          od.count = out.count.p;
          od.cap = count_only ? 0 : (uint32_t)out.cap;
          for (int c = 0; c < out.ncols && !count_only; ++c) od.col[c] = out.col[c];
          pa.out = od;
          pa.out_count = out.count.p;
          DX_HIP(hipMemsetAsync(out.count.p, 0, 8, s));
          if (A2.rows) launch_probe(s, A2.rows, pa);
          read_rows(out);
          if (count_only || out.rows <= cap) break;
          if (attempt) throw DxError(DX_E_RANGE, "join output did not fit its exact size");
          cap = out.rows;  // the counter holds the exact size: one more pass
        }
        local_count = out.rows;
        explain << " -> " << out.rows << " rows here\n";
      });
      if (!count_only && !err) {
        I = std::move(out);
        for (int c : new_cls) I_cls.push_back(c);
      }
    }
    // ---- the count over all ranks, and the last chance for a rank to report a failure
    uint64_t fin[1 + kErrSlots] = {0};
    fin[0] = err ? 0 : local_count;
    if (err) fin[1 + err_slot(err.code)] = 1;
    comm_check(cm.allreduce_sum(cm.user, fin, 1 + kErrSlots), "allreduce_sum");
    {
      std::vector<int> codes;  // by slot rather than by rank: the sum says which kinds of failure occurred
      for (int sl = 0; sl < kErrSlots; ++sl)
        if (fin[1 + sl]) codes.push_back(err_code_of_slot(sl));
      throw_if_any_failed(ctx, err, codes);
    }
    res->names = plan.select_names;
    res->virtual_terms = plan.virtual_terms;
    res->count = local_count;
    res->global_count = fin[0];
    if (mode == DX_QUERY_BINDINGS) {
      res->has_bindings = true;
      for (size_t c = 0; c < plan.select_cls.size(); ++c) {
        int at = -1;
        for (size_t j = 0; j < I_cls.size(); ++j)
          if (I_cls[j] == plan.select_cls[c]) at = (int)j;
        if (at < 0) throw DxError(DX_E_MALFORMED, "SQL: selected column is not produced by any sub-select");  // the same on every rank
        DevBuf<int64_t> col(pool, std::max<uint64_t>(res->count, 1));
        if (res->count) DX_HIP(hipMemcpyAsync(col.p, I.col[at], res->count * 8, hipMemcpyDeviceToDevice, s));
        res->cols.push_back(std::move(col));
      }
      DX_HIP(hipStreamSynchronize(s));
    }
    explain << "collectives: " << run.n_host_exchanges << " host exchanges, " << run.n_data_collectives << " data, 1 sum\n";
    res->plan = explain.str();
    return res;
  } catch (...) {
    delete res;
    throw;
  }
}

}  // namespace dx
