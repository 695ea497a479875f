// dx_part_write_layout: the reference's on-disk layout (partitioner/DAP.scala:360-382, 385-509, 101-114) from what a
// partition run left in HBM. The Parquet page bodies are encoded on the GPU:
//
//   * every column chunk (at most 2^20 rows of one column of one file) needs the distinct terms among its values
//     (dictionary page: 4-byte length + bytes per term) and, per row, the index of its term among those, bit-packed
//     (data page). All chunks of all files are done at once: one key (chunk number, term id) per value, one radix sort,
//     one pass that keeps the first of equal keys. What is left is every chunk's dictionary, in id order; a value finds
//     its index by binary search inside its chunk's stretch of that list.
//   * the VP index files (uri, parts; DAP.scala:450-501) are plain pages: per partition the (uri, role, predicate)
//     entries are sorted by (uri, the text the entry stands for), and a prefix sum over the entries' text lengths places
//     every "s<name>" / "o<name>" piece of every row's comma-joined set.
//   * page headers, footers and the other bytes that are not page bodies are made on the host from the pages' sizes
//     (dx_parquet.cpp), uploaded, and copied into their places: the device holds the files' exact bytes, laid one after
//     another ("the image"). The image comes to the host through a ring of pinned buffers, and host threads write what
//     has landed into the files (see Lanes below). The host encodes nothing.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>

#include "dx_device.h"
#include "dx_state.h"

namespace dx {
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void write_all(int fd, const uint8_t* p, uint64_t n, uint64_t at, const std::string& path) {
  while (n) {
    const ssize_t w = pwrite(fd, p, (size_t)std::min<uint64_t>(n, 1ull << 30), (off_t)at);
    if (w < 0) {
      if (errno == EINTR) continue;
      throw DxError(DX_E_IO, "write failed: " + path + ": " + strerror(errno));
    }
    p += w; n -= (uint64_t)w; at += (uint64_t)w;
  }
}

// ---- host threads that write -------------------------------------------------------------------------------------
// The files of a batch are dealt to as many lanes as there are writing threads (the largest file to the lane with the
// fewest bytes so far); in the image a lane's files lie one after another. The image then comes to the host lane by lane
// in turn, 8 MiB at a time, through a ring of pinned buffers, and thread L writes lane L's pieces into lane L's files
// as they land. Two facts of the machines this runs on shape that:
//   * a file that several threads write at once is written by one of them at a time (the file's lock is held for the
//     length of a write call), so every file has one writer -- and a layout's largest files are a large part of it, so
//     all lanes must be on their way from the first piece on, which is why the lanes take turns in the ring;
//   * pinned host memory is mapped into the process page by page when the CPU first touches it (about 1 us per 4 KiB:
//     a thread's first pass over fresh pinned memory runs at 4 GB/s, later passes at memory speed), so the bytes pass
//     through a ring that is a fraction of their size and is touched once, by all threads at once, while the GPU encodes.
struct LaneFile { const std::string* path; uint64_t at, size; };  // at: inside the lane's stretch of the image
struct LaneJob { std::vector<LaneFile> files; uint64_t bytes = 0; };

class Lanes {
 public:
  static constexpr uint64_t kChunk = 8ull << 20;
  Lanes(int device, unsigned lanes, unsigned slots) : device_(device), n_lanes_(lanes), slots_(slots), jobs_(lanes), queue_(lanes), done_(lanes, true) {
    void* p = nullptr;
    if (hipHostMalloc(&p, kChunk * slots, hipHostMallocDefault) != hipSuccess)
      throw DxError(DX_E_NOMEM, "pinned host memory for the ring of " + std::to_string((kChunk * slots) >> 20) + " MiB the layout's files pass through");
    ring_ = static_cast<uint8_t*>(p);
    for (Slot& sl : slots_) {
      if (hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) { release(); throw DxError(DX_E_HIP, "hipEventCreate failed"); }
    }
    for (unsigned l = 0; l < lanes; ++l) threads_.emplace_back([this, l] { run(l); });
  }
  ~Lanes() { stop(); release(); }
  unsigned lanes() const { return n_lanes_; }
  // a new batch: what every lane will receive. The batch before must have ended.
  void begin(std::vector<LaneJob> jobs) {
    std::lock_guard<std::mutex> l(mu_);
    jobs_ = std::move(jobs);
    for (unsigned i = 0; i < n_lanes_; ++i) done_[i] = false;
    ++epoch_;
    cv_.notify_all();
  }
  // the next buffer of the ring, once the piece that was in it has been written
  uint8_t* acquire(unsigned* slot) {
    std::unique_lock<std::mutex> l(mu_);
    const unsigned sl = next_slot_;
    cv_.wait(l, [&] { return !slots_[sl].busy; });
    slots_[sl].busy = true;
    next_slot_ = (next_slot_ + 1) % (unsigned)slots_.size();
    *slot = sl;
    return ring_ + (uint64_t)sl * kChunk;
  }
  hipEvent_t event_of(unsigned slot) { return slots_[slot].ev; }
  // the copy into `slot` and its event are queued: `bytes` more of `lane`'s stretch are on their way
  void publish(unsigned lane, unsigned slot, uint64_t bytes) {
    std::lock_guard<std::mutex> l(mu_);
    queue_[lane].push_back({slot, bytes});
    cv_.notify_all();
  }
  // waits until every lane has written its share of the batch; throws the first failure
  void end() {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { for (bool d : done_) if (!d) return false; return true; });
    if (failure_) { std::exception_ptr f = failure_; failure_ = nullptr; std::rethrow_exception(f); }
  }
  // after a failure on the producing side: the lanes stop waiting for pieces that will not come
  void abandon() {
    { std::lock_guard<std::mutex> l(mu_); abandoned_ = true; }
    cv_.notify_all();
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { for (bool d : done_) if (!d) return false; return true; });
  }
  void stop() {
    { std::lock_guard<std::mutex> l(mu_); closing_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) if (t.joinable()) t.join();
    threads_.clear();
  }
  double busy_ms = 0, first_ms = 0, last_ms = 0;  // read after end()
  uint64_t bytes = 0;

 private:
  struct Slot { hipEvent_t ev = nullptr; bool busy = false; };
  struct Piece { unsigned slot; uint64_t bytes; };
  void release() {
    for (Slot& sl : slots_) if (sl.ev) (void)hipEventDestroy(sl.ev);
    slots_.clear();
    if (ring_) (void)hipHostFree(ring_);
    ring_ = nullptr;
  }
  void run(unsigned lane) {
    (void)hipSetDevice(device_);
    // this thread's share of the ring's pages, mapped before the first piece lands
    for (size_t sl = lane; sl < slots_.size(); sl += n_lanes_) {
      volatile uint8_t* p = ring_ + (uint64_t)sl * kChunk;
      for (uint64_t b = 0; b < kChunk; b += 4096) (void)p[b];
    }
    uint64_t seen = 0;
    for (;;) {
      LaneJob job;
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return closing_ || epoch_ != seen; });
        if (epoch_ == seen) return;
        seen = epoch_;
        job = std::move(jobs_[lane]);
      }
      double busy = 0, first = 0, last = 0;
      uint64_t wrote = 0;
      std::exception_ptr failure;
      size_t cur = 0;  // the file the lane is in
      int fd = -1;
      uint64_t pos = 0;
      auto make_empty = [&](const LaneFile& f) {
        const int e = open(f.path->c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
        if (e < 0) throw DxError(DX_E_IO, "cannot create " + *f.path + ": " + strerror(errno));
        close(e);
      };
      for (;;) {
        Piece pc{};
        {
          std::unique_lock<std::mutex> l(mu_);
          if (pos >= job.bytes) break;
          cv_.wait(l, [&] { return abandoned_ || !queue_[lane].empty(); });
          if (queue_[lane].empty()) break;  // abandoned
          pc = queue_[lane].front();
          queue_[lane].pop_front();
        }
        const hipError_t e = hipEventSynchronize(slots_[pc.slot].ev);
        const double t0 = now_ms();
        if (!failure) {
          try {
            if (e != hipSuccess) throw DxError(DX_E_HIP, std::string("copy of the layout's bytes to the host failed: ") + hipGetErrorString(e));
            const uint8_t* src = ring_ + (uint64_t)pc.slot * kChunk;
            const uint64_t end = pos + pc.bytes;
            uint64_t at = pos;
            while (at < end) {
              while (cur < job.files.size() && job.files[cur].size == 0 && job.files[cur].at <= at) make_empty(job.files[cur++]);
              if (cur >= job.files.size()) break;  // padding behind the lane's last file
              const LaneFile& f = job.files[cur];
              if (at < f.at) { at = std::min(end, f.at); continue; }  // padding between files
              if (fd < 0) {
                fd = open(f.path->c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
                if (fd < 0) throw DxError(DX_E_IO, "cannot create " + *f.path + ": " + strerror(errno));
              }
              const uint64_t n = std::min(end, f.at + f.size) - at;
              write_all(fd, src + (at - pos), n, at - f.at, *f.path);
              wrote += n;
              at += n;
              if (at == f.at + f.size) {
                const int rc = close(fd);
                fd = -1;
                ++cur;
                if (rc != 0) throw DxError(DX_E_IO, "close failed: " + *f.path);
              }
            }
          } catch (...) {
            failure = std::current_exception();
            if (fd >= 0) { close(fd); fd = -1; }
          }
        }
        const double t1 = now_ms();
        busy += t1 - t0;
        if (first == 0) first = t0;
        last = t1;
        pos += pc.bytes;
        std::lock_guard<std::mutex> l(mu_);
        slots_[pc.slot].busy = false;
        cv_.notify_all();
      }
      if (!failure) {  // files of no bytes behind the last piece
        try {
          while (cur < job.files.size() && job.files[cur].size == 0) make_empty(job.files[cur++]);
        } catch (...) { failure = std::current_exception(); }
      }
      if (fd >= 0) close(fd);
      std::lock_guard<std::mutex> l(mu_);
      busy_ms += busy;
      if (first != 0 && (first_ms == 0 || first < first_ms)) first_ms = first;
      last_ms = std::max(last_ms, last);
      bytes += wrote;
      if (failure && !failure_) failure_ = failure;
      queue_[lane].clear();
      done_[lane] = true;
      cv_.notify_all();
    }
  }
  int device_;
  unsigned n_lanes_;
  uint8_t* ring_ = nullptr;
  std::vector<Slot> slots_;
  unsigned next_slot_ = 0;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<LaneJob> jobs_;
  std::vector<std::deque<Piece>> queue_;
  std::vector<bool> done_;
  std::vector<std::thread> threads_;
  std::exception_ptr failure_;
  uint64_t epoch_ = 0;
  bool closing_ = false, abandoned_ = false;
};

// ---- what the kernels see -----------------------------------------------------------------------------------------
enum : uint32_t { kChunkIds = 0, kChunkPidx = 1, kChunkPlain = 2, kChunkParts = 3 };
struct EncChunk {
  const void* src;      // kChunkIds: int64 term ids; kChunkPidx: uint32 predicate indices
  uint64_t first_val;   // place of the chunk's first key among the keys of all values
  uint64_t adj;         // image offset of the chunk's first body byte minus the byte position of its first entry
                        //   (dictionary page body; for plain and parts chunks the data page body)
  uint64_t data_at;     // image offset of the packed indices
  uint32_t n;           // values
  uint32_t kind;
  uint32_t first, count;  // entries [first, first + count) of the entry list; kChunkParts: rows of the VP index
  uint32_t width, pad;
};
struct ChunkSize { uint32_t first, count; uint64_t bytes, start; uint64_t packed_bytes, packed_start; };  // packed: as snappy elements (terms only)
struct EncTile { uint32_t chunk, first; };  // kTileVals values of one chunk
constexpr uint32_t kTileVals = 2048;
struct CopySeg { uint64_t src, dst, bytes; };  // src: offset into the uploaded bytes, or a device address

unsigned blocks_for(uint64_t n, unsigned per = 256) { return (unsigned)std::max<uint64_t>(1, (n + per - 1) / per); }

// key of every value: (chunk number, term id)
__global__ void __launch_bounds__(256)
k_enc_keys(const EncTile* __restrict__ tiles, const EncChunk* __restrict__ chunks, const int64_t* __restrict__ pred_terms, int id_bits,
           uint64_t n_terms, uint64_t* __restrict__ keys, uint32_t* __restrict__ bad) {
  const EncTile t = tiles[blockIdx.x];
  const EncChunk c = chunks[t.chunk];
  const uint64_t hi = (uint64_t)t.chunk << id_bits;
#pragma unroll
  for (int r = 0; r < (int)(kTileVals / 256); ++r) {
    const uint32_t i = t.first + r * 256 + threadIdx.x;
    if (i < c.n) {
      const uint64_t id = c.kind == kChunkPidx ? (uint64_t)pred_terms[static_cast<const uint32_t*>(c.src)[i]] : (uint64_t)static_cast<const int64_t*>(c.src)[i];
      if (id >= n_terms) *bad = 1;
      keys[c.first_val + i] = hi | (id < n_terms ? id : 0);
    }
  }
}

// first of every run of equal keys (sorted input): counts per block of 1024 keys, then the keys themselves
__device__ __forceinline__ uint32_t first_flags4(const uint64_t* __restrict__ keys, uint64_t n, uint64_t i0, int shift, uint64_t (&k)[4]) {
  uint32_t f = 0;
  if (i0 >= n) return 0;
  uint64_t prev = i0 ? keys[i0 - 1] >> shift : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (i0 + j < n) {
      k[j] = keys[i0 + j];
      if (i0 + j == 0 || (k[j] >> shift) != prev) f |= 1u << j;
      prev = k[j] >> shift;
    }
  return f;
}
__global__ void __launch_bounds__(256) k_first_count(const uint64_t* __restrict__ keys, uint64_t n, int shift, uint32_t* __restrict__ block_cnt) {
  __shared__ uint32_t lds[8];
  uint64_t k[4];
  const uint32_t f = first_flags4(keys, n, 4 * ((uint64_t)blockIdx.x * 256 + threadIdx.x), shift, k);
  uint32_t total;
  (void)block_excl_scan_256<uint32_t>(__popc(f), lds, total);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256)
k_first_compact(const uint64_t* __restrict__ keys, uint64_t n, const uint32_t* __restrict__ block_base, uint64_t* __restrict__ out) {
  __shared__ uint32_t lds[8];
  __shared__ uint64_t stage[1024];
  uint64_t k[4];
  const uint32_t f = first_flags4(keys, n, 4 * ((uint64_t)blockIdx.x * 256 + threadIdx.x), 0, k);
  uint32_t total;
  uint32_t at = block_excl_scan_256<uint32_t>(__popc(f), lds, total);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (f & (1u << j)) stage[at++] = k[j];
  __syncthreads();
  const uint32_t base = block_base[blockIdx.x];
  for (uint32_t j = threadIdx.x; j < total; j += 256) out[base + j] = stage[j];
}

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t* __restrict__ a, uint32_t n, uint64_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// bytes every entry takes in a page: length word + term
__global__ void __launch_bounds__(256)
k_entry_lens(const uint64_t* __restrict__ entries, uint32_t n, uint64_t id_mask, const uint64_t* __restrict__ pool_off, uint32_t* __restrict__ len4) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint64_t id = entries[j] & id_mask;
  len4[j] = 4 + (uint32_t)(pool_off[id + 1] - pool_off[id]);
}

// where every chunk's entries are, and how many bytes they take
__global__ void __launch_bounds__(256)
k_chunk_sizes(const EncChunk* __restrict__ chunks, uint32_t n_chunks, const uint64_t* __restrict__ entries, uint32_t n_sorted, int id_bits,
              const uint64_t* __restrict__ dpos, uint32_t n_entries, const uint64_t* __restrict__ dtotal,
              const uint32_t* __restrict__ rowstart, const uint64_t* __restrict__ epos, uint32_t n_vpi, const uint64_t* __restrict__ etotal,
              const uint64_t* __restrict__ cpos /*null: pages are not packed*/, const uint64_t* __restrict__ ctotal,
              const uint64_t* __restrict__ rpos /*null: the parts columns are not packed*/, const uint64_t* __restrict__ rtotal, uint32_t n_rows,
              ChunkSize* __restrict__ out) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_chunks) return;
  const EncChunk c = chunks[k];
  ChunkSize z;
  z.packed_bytes = z.packed_start = 0;
  if (c.kind == kChunkParts) {
    z.first = c.first; z.count = c.count;
    const uint32_t a = rowstart[c.first], b = rowstart[c.first + c.count];
    z.start = a == n_vpi ? *etotal : epos[a];
    z.bytes = (b == n_vpi ? *etotal : epos[b]) - z.start;
    if (rpos) {
      z.packed_start = c.first == n_rows ? *rtotal : rpos[c.first];
      z.packed_bytes = (c.first + c.count == n_rows ? *rtotal : rpos[c.first + c.count]) - z.packed_start;
    }
  } else {
    if (c.kind == kChunkPlain) { z.first = c.first; z.count = c.count; }
    else {
      z.first = lower_bound_u64(entries, n_sorted, (uint64_t)k << id_bits);
      z.count = lower_bound_u64(entries, n_sorted, (uint64_t)(k + 1) << id_bits) - z.first;
    }
    const uint32_t e = z.first + z.count;
    z.start = z.first == n_entries ? *dtotal : dpos[z.first];
    z.bytes = (e == n_entries ? *dtotal : dpos[e]) - z.start;
    if (cpos) {
      z.packed_start = z.first == n_entries ? *ctotal : cpos[z.first];
      z.packed_bytes = (e == n_entries ? *ctotal : cpos[e]) - z.packed_start;
    }
  }
  out[k] = z;
}

// page bodies of length-prefixed terms (dictionary pages, plain pages of ids): four lanes per entry
__global__ void __launch_bounds__(256)
k_enc_terms(const uint64_t* __restrict__ entries, uint32_t n, int id_bits, const EncChunk* __restrict__ chunks, const uint64_t* __restrict__ dpos,
            const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off, uint8_t* __restrict__ image) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t j = g >> 2;
  const uint32_t sub = (uint32_t)g & 3;
  if (j >= n) return;
  const uint64_t key = entries[j];
  const uint64_t id = key & ((1ull << id_bits) - 1);
  const uint64_t b = pool_off[id];
  const uint32_t len = (uint32_t)(pool_off[id + 1] - b);
  uint8_t* dst = image + chunks[key >> id_bits].adj + dpos[j];
  if (sub == 0) __builtin_memcpy(dst, &len, 4);
  dst += 4;
  const uint8_t* src = pool + b;
  if (len >= 16) {
    for (uint32_t at0 = 16 * sub; at0 < len; at0 += 64) {
      const uint32_t at = at0 + 16 <= len ? at0 : len - 16;
      uint4 v;
      __builtin_memcpy(&v, src + at, 16);
      __builtin_memcpy(dst + at, &v, 16);
    }
  } else if (sub == 0) {
    for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
  }
}

// ---- the same pages as snappy streams ----------------------------------------------------------------------------------
// A page of terms is a list of records (4-byte length, bytes), and the terms of a page follow each other in id order, i.e.
// in the order they first appeared in the text: a term shares most of its bytes with the one before it (the students of a
// department, the publications of an author). So a record is written as "the first bytes of the record before it"
// (a copy element reaching back by that record's size) and the rest as a literal; every record's elements depend on its
// neighbour's bytes only, not on how the neighbour was packed, so all records are sized and written side by side:
//   first record of a page, or the record before is 64 KiB or longer:   literal(length word + term)
//   same length as the term before, m common bytes:                      copy(4 + m), literal(rest)
//   another length, m >= 4 common bytes:                                 literal(length word), copy(m), literal(rest)
//   otherwise:                                                           literal(length word + term)
// Copies are the form with a 2-byte distance, at most 64 bytes each. This is not a general compressor: it looks at
// one candidate per record. On LUBM's pages it leaves about a third of the bytes.
struct PackPlan { uint32_t head_lit, copy, tail_lit; };  // literal bytes from the record's start, copied bytes, literal bytes to its end
__device__ __forceinline__ uint32_t lit_bytes(uint32_t n) { return n == 0 ? 0 : n + (n <= 60 ? 1 : n <= 256 ? 2 : n <= 65536 ? 3 : n <= (1u << 24) ? 4 : 5); }
__device__ __forceinline__ uint32_t copy_bytes(uint32_t n) { return 3 * ((n + 63) / 64); }
__device__ __forceinline__ uint32_t pack_bytes(const PackPlan& p) { return lit_bytes(p.head_lit) + copy_bytes(p.copy) + lit_bytes(p.tail_lit); }
__device__ __forceinline__ uint32_t common_prefix(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t n) {
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t x, y;
    __builtin_memcpy(&x, a + i, 8);
    __builtin_memcpy(&y, b + i, 8);
    if (x != y) return i + ((uint32_t)__builtin_ctzll(x ^ y) >> 3);
  }
  for (; i < n && a[i] == b[i]; ++i) {}
  return i;
}
// entry j of the list and the one before it (if it belongs to the same page)
__device__ __forceinline__ PackPlan pack_plan(const uint64_t* __restrict__ entries, uint32_t j, int id_bits, const uint8_t* __restrict__ pool,
                                              const uint64_t* __restrict__ pool_off, uint32_t& len, uint64_t& at, uint32_t& prev_len) {
  const uint64_t key = entries[j], id = key & ((1ull << id_bits) - 1);
  at = pool_off[id];
  len = (uint32_t)(pool_off[id + 1] - at);
  prev_len = 0;
  PackPlan p{4 + len, 0, 0};
  if (j == 0) return p;
  const uint64_t before = entries[j - 1];
  if ((before >> id_bits) != (key >> id_bits)) return p;
  const uint64_t pid = before & ((1ull << id_bits) - 1), pat = pool_off[pid];
  prev_len = (uint32_t)(pool_off[pid + 1] - pat);
  if (4ull + prev_len > 65535) return p;
  const uint32_t m = common_prefix(pool + at, pool + pat, len < prev_len ? len : prev_len);
  if (len == prev_len) return PackPlan{0, 4 + m, len - m};
  if (m >= 4) return PackPlan{4, m, len - m};
  return p;
}
__global__ void __launch_bounds__(256)
k_entry_packed_lens(const uint64_t* __restrict__ entries, uint32_t n, int id_bits, const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off,
                    uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t len, prev_len;
  uint64_t at;
  out[j] = pack_bytes(pack_plan(entries, j, id_bits, pool, pool_off, len, at, prev_len));
}
__device__ __forceinline__ uint8_t* put_literal(uint8_t* d, const uint8_t* __restrict__ a, uint32_t na, const uint8_t* __restrict__ b, uint32_t nb) {
  const uint32_t n = na + nb;  // the literal's bytes are a[0, na) then b[0, nb)
  if (!n) return d;
  const uint32_t v = n - 1;
  if (v < 60) { *d++ = (uint8_t)(v << 2); }
  else {
    const int bytes = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < (1u << 24) ? 3 : 4;
    *d++ = (uint8_t)((59 + bytes) << 2);
    for (int i = 0; i < bytes; ++i) *d++ = (uint8_t)(v >> (8 * i));
  }
  for (uint32_t i = 0; i < na; ++i) *d++ = a[i];
  uint32_t i = 0;
  for (; i + 8 <= nb; i += 8) {
    uint64_t x;
    __builtin_memcpy(&x, b + i, 8);
    __builtin_memcpy(d + i, &x, 8);
  }
  for (; i < nb; ++i) d[i] = b[i];
  return d + nb;
}
__global__ void __launch_bounds__(256)
k_enc_terms_packed(const uint64_t* __restrict__ entries, uint32_t n, int id_bits, const EncChunk* __restrict__ chunks, const uint64_t* __restrict__ cpos,
                   const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off, uint8_t* __restrict__ image) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t len, prev_len;
  uint64_t at;
  const PackPlan p = pack_plan(entries, j, id_bits, pool, pool_off, len, at, prev_len);
  uint8_t* d = image + chunks[entries[j] >> id_bits].adj + cpos[j];
  const uint8_t* term = pool + at;
  uint8_t word[4];
  __builtin_memcpy(word, &len, 4);
  // literal from the record's start: the length word, and the term too unless a copy follows
  if (p.head_lit) d = put_literal(d, word, 4, term, p.head_lit - 4);
  const uint32_t back = 4 + prev_len;
  for (uint32_t left = p.copy; left;) {
    const uint32_t l = left < 64 ? left : 64;
    *d++ = (uint8_t)(2 | ((l - 1) << 2));
    *d++ = (uint8_t)back;
    *d++ = (uint8_t)(back >> 8);
    left -= l;
  }
  if (p.tail_lit) d = put_literal(d, word, 0, term + (len - p.tail_lit), p.tail_lit);
}

// data pages: every thread packs one group of eight indices
__global__ void __launch_bounds__(256)
k_enc_indices(const EncTile* __restrict__ tiles, const EncChunk* __restrict__ chunks, const int64_t* __restrict__ pred_terms,
              const uint64_t* __restrict__ entries, int id_bits, uint8_t* __restrict__ image) {
  const EncTile t = tiles[blockIdx.x];
  const EncChunk c = chunks[t.chunk];
  const uint32_t g = t.first / 8 + threadIdx.x, i0 = g * 8;
  if (i0 >= c.n || c.width == 0) return;
  const uint64_t hi = (uint64_t)t.chunk << id_bits;
  const uint64_t* seg = entries + c.first;
  uint64_t acc = 0;
  int nb = 0;
  uint8_t* d = image + c.data_at + (uint64_t)g * c.width;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint32_t v = 0;
    if (i0 + r < c.n) {
      const uint64_t id = c.kind == kChunkPidx ? (uint64_t)pred_terms[static_cast<const uint32_t*>(c.src)[i0 + r]] : (uint64_t)static_cast<const int64_t*>(c.src)[i0 + r];
      v = lower_bound_u64(seg, c.count, hi | id);
    }
    acc |= (uint64_t)v << nb;
    nb += (int)c.width;
    while (nb >= 8) { *d++ = (uint8_t)acc; acc >>= 8; nb -= 8; }
  }
}

// raw stretches of device arrays into the image, 8 bytes per lane; one block per 16 KiB piece
__global__ void __launch_bounds__(256) k_copy_words(const EncTile* __restrict__ tiles, const CopySeg* __restrict__ segs, uint8_t* __restrict__ image) {
  const EncTile t = tiles[blockIdx.x];
  const CopySeg s = segs[t.chunk];
  const uint8_t* src = reinterpret_cast<const uint8_t*>(s.src);
  uint8_t* dst = image + s.dst;
  const uint64_t b0 = (uint64_t)t.first * 16384, b1 = std::min<uint64_t>(s.bytes, b0 + 16384);
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 7) == 0;
  if (aligned) {
    const uint64_t w1 = b1 & ~7ull;
    for (uint64_t b = b0 + 8ull * threadIdx.x; b < w1; b += 2048) *reinterpret_cast<uint64_t*>(dst + b) = *reinterpret_cast<const uint64_t*>(src + b);
    for (uint64_t b = w1 + threadIdx.x; b < b1; b += 256) dst[b] = src[b];
  } else {
    for (uint64_t b = b0 + threadIdx.x; b < b1; b += 256) dst[b] = src[b];
  }
}
// the bytes the host made (headers, footers): one wave per piece
__global__ void __launch_bounds__(256) k_copy_small(const CopySeg* __restrict__ segs, uint32_t n, const uint8_t* __restrict__ blob, uint8_t* __restrict__ image) {
  const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  const CopySeg s = segs[k];
  for (uint64_t b = threadIdx.x & 63; b < s.bytes; b += 64) image[s.dst + b] = blob[s.src + b];
}

// ---- VP index ------------------------------------------------------------------------------------------------------
// packed (partition | term | role | predicate) words of the partition run -> (partition | term | code), where the codes
// are in the order of the texts they stand for: "o<name>" before "s<name>", names in byte order
__global__ void __launch_bounds__(256)
k_vpi_keys(const uint64_t* __restrict__ a, uint64_t na, const uint64_t* __restrict__ b, uint64_t nb, int bits_pred, int bits_code,
           const uint32_t* __restrict__ name_rank, uint32_t n_names, uint64_t* __restrict__ keys) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= na + nb) return;
  const uint64_t k = i < na ? a[i] : b[i - na];
  const uint32_t pidx = (uint32_t)(k & ((1ull << bits_pred) - 1));
  const uint32_t role = (uint32_t)(k >> bits_pred) & 1;  // 0: subject
  const uint64_t hi = k >> (bits_pred + 1);               // partition | term
  keys[i] = (hi << bits_code) | ((role == 0 ? n_names : 0) + name_rank[pidx]);
}
// per entry: is it the first of its (partition, uri); bytes of its piece of the row's text
__global__ void __launch_bounds__(256)
k_vpi_entry_bytes(const uint64_t* __restrict__ e, uint32_t n, int bits_code, const uint32_t* __restrict__ name_off, uint32_t n_names,
                  uint8_t* __restrict__ head, uint32_t* __restrict__ bytes) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint64_t k = e[j];
  const bool h = j == 0 || (e[j - 1] >> bits_code) != (k >> bits_code);
  uint32_t code = (uint32_t)(k & ((1ull << bits_code) - 1));
  if (code >= n_names) code -= n_names;
  head[j] = h ? 1 : 0;
  bytes[j] = (h ? 4 : 1) + 1 + (name_off[code + 1] - name_off[code]);  // length word or comma, role letter, name
}
__global__ void __launch_bounds__(256)
k_vpi_rowstart(const uint8_t* __restrict__ head, const uint32_t* __restrict__ rowidx, uint32_t n, uint32_t n_rows, uint32_t* __restrict__ rowstart) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < n && head[j]) rowstart[rowidx[j]] = j;
  if (j == 0) rowstart[n_rows] = n;
}
// first row of every partition (rows are numbered over all partitions)
__global__ void k_vpi_rowbase(const uint64_t* __restrict__ e, uint32_t n, int shift, int nparts, const uint32_t* __restrict__ rowidx, uint32_t n_rows,
                              uint32_t* __restrict__ rowbase) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > nparts) return;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((e[mid] >> shift) < (uint64_t)q) lo = mid + 1; else hi = mid;
  }
  rowbase[q] = lo == n ? n_rows : rowidx[lo];
}
// the uri column of the VP index files as entries of plain chunks
__global__ void __launch_bounds__(256)
k_vpi_uri_entries(const uint64_t* __restrict__ e, uint32_t n, const uint8_t* __restrict__ head, const uint32_t* __restrict__ rowidx, int bits_code,
                  int bits_term, const uint32_t* __restrict__ rowbase, const uint32_t* __restrict__ chunk0, int id_bits, uint64_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n || !head[j]) return;
  const uint64_t hi = e[j] >> bits_code;
  const uint32_t q = (uint32_t)(hi >> bits_term), r = rowidx[j];
  const uint64_t term = hi & ((1ull << bits_term) - 1);
  const uint32_t chunk = chunk0[q] + 2 * (uint32_t)((r - rowbase[q]) / kParquetRowGroupRows);
  out[r] = ((uint64_t)chunk << id_bits) | term;
}
// the parts column: every entry writes its piece, four lanes per entry
__global__ void __launch_bounds__(256)
k_vpi_parts(const uint64_t* __restrict__ e, uint32_t n, const uint8_t* __restrict__ head, const uint32_t* __restrict__ rowidx,
            const uint32_t* __restrict__ rowstart, const uint64_t* __restrict__ epos, const uint64_t* __restrict__ etotal, int bits_code, int bits_term,
            const uint32_t* __restrict__ rowbase, const uint32_t* __restrict__ chunk0, const EncChunk* __restrict__ chunks,
            const uint8_t* __restrict__ names, const uint32_t* __restrict__ name_off, uint32_t n_names, uint8_t* __restrict__ image) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t j = (uint32_t)(g >> 2), sub = (uint32_t)g & 3;
  if (j >= n) return;
  const uint64_t k = e[j];
  const bool h = head[j] != 0;
  const uint32_t r = rowidx[j] - (h ? 0 : 1);
  const uint32_t q = (uint32_t)(k >> (bits_code + bits_term));
  uint32_t code = (uint32_t)(k & ((1ull << bits_code) - 1));
  const char letter = code >= n_names ? 's' : 'o';
  if (code >= n_names) code -= n_names;
  const uint32_t chunk = chunk0[q] + 2 * (uint32_t)((r - rowbase[q]) / kParquetRowGroupRows) + 1;
  uint8_t* dst = image + chunks[chunk].adj + epos[j];
  if (sub == 0) {
    if (h) {
      const uint32_t nxt = rowstart[r + 1];
      const uint32_t len = (uint32_t)((nxt == n ? *etotal : epos[nxt]) - epos[j] - 4);
      __builtin_memcpy(dst, &len, 4);
      dst[4] = (uint8_t)letter;
    } else {
      dst[0] = ',';
      dst[1] = (uint8_t)letter;
    }
  }
  dst += h ? 5 : 2;
  const uint8_t* src = names + name_off[code];
  const uint32_t len = name_off[code + 1] - name_off[code];
  if (len >= 16) {
    for (uint32_t at0 = 16 * sub; at0 < len; at0 += 64) {
      const uint32_t at = at0 + 16 <= len ? at0 : len - 16;
      uint4 v;
      __builtin_memcpy(&v, src + at, 16);
      __builtin_memcpy(dst + at, &v, 16);
    }
  } else if (sub == 0) {
    for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
  }
}

// The parts column as snappy elements, by the rules of the pages of terms (k_enc_terms_packed): a row's text is the pieces of
// its entries, the row before it in the file very often has the same pieces or begins with the same ones (the students of a
// department carry the same predicates), so a row is a copy of the first bytes of the row before it and a literal of
// the pieces that are left. Common bytes are counted in whole pieces. Per row: .x = pieces that are copied | bytes the
// row's elements take before the literal's payload << 16, .y = common bytes | form << 30 (0: one literal, 1: same
// length as the row before, 2: another length).
__global__ void __launch_bounds__(256)
k_vpi_rows_plan(const uint64_t* __restrict__ e, uint32_t n, const uint32_t* __restrict__ rowstart, uint32_t n_rows, const uint64_t* __restrict__ epos,
                const uint64_t* __restrict__ etotal, int bits_code, int bits_term, const uint32_t* __restrict__ rowbase,
                const uint32_t* __restrict__ name_off, uint32_t n_names, uint2* __restrict__ plan, uint32_t* __restrict__ packed) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const uint32_t a = rowstart[r], b = rowstart[r + 1];
  const uint64_t end = b == n ? *etotal : epos[b];
  const uint32_t len = (uint32_t)(end - epos[a] - 4);
  const uint32_t q = (uint32_t)(e[a] >> (bits_code + bits_term));
  uint32_t c = 0, m = 0, form = 0;
  const bool first = (r - rowbase[q]) % (uint32_t)kParquetRowGroupRows == 0;
  uint32_t prev_len = 0;
  if (!first) {
    const uint32_t pa = rowstart[r - 1];
    prev_len = (uint32_t)(epos[a] - epos[pa] - 4);
    if (4ull + prev_len <= 65535) {
      const uint64_t code_mask = (1ull << bits_code) - 1;
      const uint32_t most = (b - a) < (a - pa) ? (b - a) : (a - pa);
      while (c < most && (e[a + c] & code_mask) == (e[pa + c] & code_mask)) {
        uint32_t code = (uint32_t)(e[a + c] & code_mask);
        if (code >= n_names) code -= n_names;
        m += (c ? 2 : 1) + (name_off[code + 1] - name_off[code]);  // (comma,) role letter, name
        ++c;
      }
      if (len == prev_len) form = 1;
      else if (m >= 4) form = 2;
      else { c = 0; m = 0; }
    }
  }
  if (form == 0) { c = 0; m = 0; }
  const uint32_t tail = len - m;
  const uint32_t before_payload = form == 0 ? lit_bytes(4 + len) - (4 + len)
                                : form == 1 ? copy_bytes(4 + m) + (lit_bytes(tail) - tail)
                                            : 5 + copy_bytes(m) + (lit_bytes(tail) - tail);
  plan[r] = make_uint2(c | (before_payload << 16), m | (form << 30));
  packed[r] = before_payload + (form == 0 ? 4 + len : tail);
}
__device__ __forceinline__ uint8_t* put_literal_tag(uint8_t* d, uint32_t n) {
  if (!n) return d;
  const uint32_t v = n - 1;
  if (v < 60) { *d++ = (uint8_t)(v << 2); return d; }
  const int bytes = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < (1u << 24) ? 3 : 4;
  *d++ = (uint8_t)((59 + bytes) << 2);
  for (int i = 0; i < bytes; ++i) *d++ = (uint8_t)(v >> (8 * i));
  return d;
}
__global__ void __launch_bounds__(256)
k_vpi_parts_packed(const uint64_t* __restrict__ e, uint32_t n, const uint8_t* __restrict__ head, const uint32_t* __restrict__ rowidx,
                   const uint32_t* __restrict__ rowstart, const uint64_t* __restrict__ epos, const uint64_t* __restrict__ etotal, int bits_code,
                   int bits_term, const uint32_t* __restrict__ rowbase, const uint32_t* __restrict__ chunk0, const EncChunk* __restrict__ chunks,
                   const uint8_t* __restrict__ names, const uint32_t* __restrict__ name_off, uint32_t n_names, const uint2* __restrict__ plan,
                   const uint64_t* __restrict__ rpos, uint8_t* __restrict__ image) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t j = (uint32_t)(g >> 2), sub = (uint32_t)g & 3;
  if (j >= n) return;
  const uint64_t k = e[j];
  const bool h = head[j] != 0;
  const uint32_t r = rowidx[j] - (h ? 0 : 1);
  const uint32_t q = (uint32_t)(k >> (bits_code + bits_term));
  const uint32_t chunk = chunk0[q] + 2 * (uint32_t)((r - rowbase[q]) / kParquetRowGroupRows) + 1;
  const uint2 pl = plan[r];
  const uint32_t c = pl.x & 0xFFFFu, before_payload = pl.x >> 16, m = pl.y & 0x3FFFFFFFu, form = pl.y >> 30;
  const uint32_t a = rowstart[r];
  uint8_t* row = image + chunks[chunk].adj + rpos[r];
  if (h && sub == 1) {  // the row's elements up to the literal's payload (the lane next to the one that writes the first piece)
    const uint32_t nxt = rowstart[r + 1];
    const uint32_t len = (uint32_t)((nxt == n ? *etotal : epos[nxt]) - epos[a] - 4);
    uint8_t* d = row;
    if (form == 0) {
      d = put_literal_tag(d, 4 + len);
    } else {
      const uint32_t back = 4 + (uint32_t)(epos[a] - epos[rowstart[r - 1]] - 4);
      if (form == 2) {
        *d++ = (uint8_t)(3 << 2);  // a literal of four bytes: the length word
        __builtin_memcpy(d, &len, 4);
        d += 4;
      }
      for (uint32_t left = form == 1 ? 4 + m : m; left;) {
        const uint32_t l = left < 64 ? left : 64;
        *d++ = (uint8_t)(2 | ((l - 1) << 2));
        *d++ = (uint8_t)back;
        *d++ = (uint8_t)(back >> 8);
        left -= l;
      }
      d = put_literal_tag(d, len - m);
    }
  }
  if (j - a < c) return;  // this piece is part of what the row copies
  uint32_t code = (uint32_t)(k & ((1ull << bits_code) - 1));
  const char letter = code >= n_names ? 's' : 'o';
  if (code >= n_names) code -= n_names;
  // A row that is one literal carries its length word in the payload, in front of its first piece; otherwise the payload is
  // the row's text from byte m on (the word has been written or copied before it).
  uint8_t* dst;
  uint32_t skip;  // bytes of the piece in front of the name
  if (form == 0) {
    dst = row + before_payload + (epos[j] - epos[a]);
    skip = h ? 5 : 2;
    if (sub == 0) {
      if (h) {
        const uint32_t nxt = rowstart[r + 1];
        const uint32_t len = (uint32_t)((nxt == n ? *etotal : epos[nxt]) - epos[j] - 4);
        __builtin_memcpy(dst, &len, 4);
        dst[4] = (uint8_t)letter;
      } else {
        dst[0] = ',';
        dst[1] = (uint8_t)letter;
      }
    }
  } else {
    const uint32_t at_text = h ? 0u : (uint32_t)(epos[j] - epos[a] - 4);  // where the piece begins in the row's text
    dst = row + before_payload + (at_text - m);
    skip = h ? 1 : 2;
    if (sub == 0) {
      if (h) dst[0] = (uint8_t)letter;
      else { dst[0] = ','; dst[1] = (uint8_t)letter; }
    }
  }
  dst += skip;
  const uint8_t* src = names + name_off[code];
  const uint32_t len = name_off[code + 1] - name_off[code];
  if (len >= 16) {
    for (uint32_t at0 = 16 * sub; at0 < len; at0 += 64) {
      const uint32_t at = at0 + 16 <= len ? at0 : len - 16;
      uint4 v;
      __builtin_memcpy(&v, src + at, 16);
      __builtin_memcpy(dst + at, &v, 16);
    }
  } else if (sub == 0) {
    for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
struct LayoutFile {
  std::string path;
  std::vector<std::string> cols;   // Parquet columns; none: a file of raw bytes
  uint64_t rows = 0;
  struct Col { const int64_t* ids = nullptr; const uint32_t* pidx = nullptr; };
  std::vector<Col> src;            // Parquet with dictionary pages: where every column's values are
  int vpi_partition = -1;          // a VP index file (uri, parts) of this partition: plain pages
  struct Raw { std::string host; const void* dev = nullptr; uint64_t bytes = 0; };
  std::vector<Raw> raw;
  // planning
  std::vector<uint32_t> chunk_ids;  // row group after row group, column after column
  std::vector<uint64_t> group_rows;
  uint64_t at = 0, size = 0;        // place in the image
};

template <typename T>
void upload(DevBuf<T>& d, DevPool& pool, const std::vector<T>& h, hipStream_t s) {
  d = DevBuf<T>(pool, h.size());
  if (!h.empty()) DX_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

struct VpiState {  // the VP index of all partitions as sorted distinct (partition | term | code) entries
  uint32_t n = 0, n_rows = 0, n_names = 0;
  int bits_code = 0, bits_term = 0;
  DevBuf<uint64_t> e, epos, etotal, rpos, rtotal;  // rpos: where every row's elements begin when the parts columns are packed
  DevBuf<uint2> rplan;
  bool parts_packed = false;
  DevBuf<uint8_t> head, names;
  DevBuf<uint32_t> rowidx, rowstart, rowbase, name_off, chunk0;
  std::vector<uint32_t> h_rowbase;
};

void sync(hipStream_t s) { DX_HIP(hipStreamSynchronize(s)); }

// sorts keys[0, n) on their low `bits` bits and keeps the first of equal ones; returns how many are left
uint32_t sort_distinct(DevPool& pool, DevBuf<uint64_t>& keys, uint64_t n, int bits, DevBuf<uint64_t>& out, uint64_t extra_room, hipStream_t s) {
  if (!n) { out = DevBuf<uint64_t>(pool, extra_room); return 0; }
  radix_sort_u64(pool, keys, n, bits, s);
  const unsigned nb = blocks_for(n, 1024);
  DevBuf<uint32_t> cnt(pool, nb), total(pool, 1);
  hipLaunchKernelGGL(k_first_count, dim3(nb), dim3(256), 0, s, keys.p, n, 0, cnt.p);
  exclusive_scan_u32_u32(pool, cnt.p, cnt.p, nb, total.p, s);
  uint32_t m = 0;
  DX_HIP(hipMemcpyAsync(&m, total.p, 4, hipMemcpyDeviceToHost, s));
  sync(s);
  out = DevBuf<uint64_t>(pool, (uint64_t)m + extra_room);
  hipLaunchKernelGGL(k_first_compact, dim3(nb), dim3(256), 0, s, keys.p, n, cnt.p, out.p);
  DX_HIP(hipGetLastError());
  return m;
}

int bits_for(uint64_t n) {  // bits that hold every value below n
  int b = 1;
  while (b < 64 && (1ull << b) < n) ++b;
  return b;
}

}  // namespace

void write_layout(dx_ctx* ctx, const char* base, const char* dataset, const char* local_dir) {
  const double t_begin = now_ms();
  const PartitionState& ps = ctx->part;
  const Dictionary& d = ctx->dict;
  if (!ps.valid || !dataset) throw DxError(DX_E_INVAL, "no partition run / null dataset");
  LayoutTimes& tm = ctx->layout_times;
  tm = LayoutTimes{};
  const std::string B = norm_base(base), D = dataset, N = std::to_string(ps.n_partitions);
  std::string L = local_dir ? local_dir : "";
  if (!L.empty() && L.back() != '/') L += '/';
  const std::string clusters = B + D + "_data/cluster/clusters_" + N + "_bal";
  struct stat stt;
  if (stat(clusters.c_str(), &stt) == 0)  // DAP.scala:86: the stage is skipped when its output exists
    throw DxError(DX_E_INVAL, clusters + " already exists (the reference skips the stage in that case; remove it to rewrite)");
  hipStream_t s = ctx->stream;
  DevPool pool;  // what this call allocates goes back to the device when it returns
  unsigned n_threads = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("DX_WRITE_THREADS")) n_threads = (unsigned)std::max(1, atoi(e));
  tm.threads = (int)n_threads;

  // 256 MiB: two 8 MiB pieces in flight per lane at sixteen lanes. Mapping the ring's pages costs the lanes' threads
  // 49 ms per GiB and competes with everything else the host does meanwhile; with 1 GiB the call took 114 ms at LUBM-100,
  // with 512 / 256 / 128 MiB 101 / 93 / 92 ms.
  uint64_t ring_mib = 256;
  if (const char* e = getenv("DX_LAYOUT_RING_MIB")) ring_mib = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  const unsigned ring_slots = (unsigned)std::max<uint64_t>(2ull * n_threads, (ring_mib << 20) / Lanes::kChunk);
  Lanes lanes(ctx->device, n_threads, ring_slots);  // the threads map the ring's pages while the GPU encodes
  bool batch_open = false;
  std::vector<LayoutFile> files;
  std::vector<std::string> success;  // written when everything else is
  try {
  // ---- the files ---------------------------------------------------------------------------------------------------
  mkdirs(clusters);
  // What the Scala tools read is written as Parquet where they look for it. Beside it, under <dataset>_data/dx_<N>/
  // (a directory none of them lists), the dictionary and the tables' id columns, which dx_open_dataset and
  // dx_load_table read back without decoding strings.
  const std::string side = B + D + "_data/dx_" + N;
  mkdirs(side + "/dictionary");
  // Spark names its files part-<task>-<job uuid>-c000.<codec>.parquet; the partition number is read from the five
  // digits after "part-" (DAP.scala:427-428). One id per call stands in for the job's uuid.
  char job[64];
  snprintf(job, sizeof job, "%08x-dx00-4000-8000-%012llx", (unsigned)time(nullptr), (unsigned long long)(uintptr_t)ctx & 0xFFFFFFFFFFFFull);
  const bool snappy = parquet_writer_uses_snappy();  // Spark's default codec and its file names, unless DX_PARQUET_CODEC=uncompressed
  auto part_file = [&](int k) { return "part-" + part_name(k) + "-" + job + (snappy ? "-c000.snappy.parquet" : "-c000.parquet"); };
  {
    LayoutFile f;
    f.path = side + "/dictionary/offsets.bin";
    f.raw.push_back({"", d.pool_off.p, (d.n_terms + 1) * 8});
    files.push_back(f);
    uint64_t pool_bytes = 0;
    DX_HIP(hipMemcpyAsync(&pool_bytes, d.pool_off.p + d.n_terms, 8, hipMemcpyDeviceToHost, s));
    sync(s);
    f.path = side + "/dictionary/terms.bin";
    f.raw[0] = {"", d.pool.p, pool_bytes};
    files.push_back(f);
  }
  // first-level partitions + statistics.txt_N (DAP.scala:364-381)
  mkdirs(L + D + "_local");
  {
    std::ofstream stats(L + D + "_local/statistics.txt_" + N);
    for (int q = 0; q < ps.n_partitions; ++q) {
      const uint64_t b = ps.part_base[q], n = ps.part_base[q + 1] - b;
      if (!n) continue;  // Spark writes no part file for an empty partition
      LayoutFile f;
      f.path = clusters + "/" + part_file(q);
      f.cols = {"_1", "_2", "_3"};  // (s, o, p): DAP.scala:576,592,628
      f.rows = n;
      f.src = {{ps.row_s.p + b, nullptr}, {ps.row_o.p + b, nullptr}, {nullptr, ps.row_p.p + b}};
      files.push_back(std::move(f));
      const std::string qs = std::to_string(q);
      stats << (qs.size() == 1 ? "0" + qs : qs) << " " << n << "\n";
    }
    if (!stats) throw DxError(DX_E_IO, "writing statistics.txt failed");
  }
  success.push_back(clusters + "/_SUCCESS");
  // VP directories (DAP.scala:419-437) and partition_stat (DAP.scala:101-114)
  std::map<std::string, uint64_t> dir_rows;
  {
    std::map<std::string, int> file_no;
    std::string last_pdir;
    for (const VpTable& t : ps.tables) {
      if (!t.in_layout) continue;
      const std::string rel = part_name(t.partition) + "/_3=" + ps.pred_storage[t.pidx];
      const std::string pdir = clusters + "/" + part_name(t.partition);
      const std::string dir = clusters + "/" + rel, sdir = side + "/cluster/" + rel;
      const int k = file_no[dir]++;
      if (k == 0) { mkdirs(dir); mkdirs(sdir); }
      LayoutFile f;
      f.path = dir + "/" + part_file(k);
      f.cols = {"_1", "_2"};
      f.rows = t.rows;
      f.src = {{ps.row_s.p + t.begin, nullptr}, {ps.row_o.p + t.begin, nullptr}};
      files.push_back(std::move(f));
      LayoutFile c;  // the same rows as id columns: header, subjects, objects
      c.path = sdir + "/part-" + part_name(k) + ".dxcol";
      std::string h(24, '\0');
      const uint64_t two = 2;
      memcpy(&h[0], "DXCOL1\0\0", 8);
      memcpy(&h[8], &t.rows, 8);
      memcpy(&h[16], &two, 8);
      c.raw.push_back({h, nullptr, 24});
      c.raw.push_back({"", ps.row_s.p + t.begin, t.rows * 8});
      c.raw.push_back({"", ps.row_o.p + t.begin, t.rows * 8});
      files.push_back(std::move(c));
      dir_rows[part_name(t.partition)] += t.rows;
      if (pdir != last_pdir) { success.push_back(pdir + "/_SUCCESS"); last_pdir = pdir; }
    }
  }
  {
    std::ofstream pstat(L + D + "_local/partition_stat.txt_" + N + "_bal");
    for (auto& kv : dir_rows) pstat << kv.first << " " << kv.second << "\n";  // `hdfs dfs -ls` lists by name
    for (int q = 0; q < ps.n_partitions; ++q) {
      const uint64_t n = ps.part_base[q + 1] - ps.part_base[q];
      if (n) pstat << part_file(q) << " " << n << "\n";
    }
    // statTables.txt: rows per VP table, the input QueryOptimizer.scala:270-276 reads (the reference has no writer for it)
    std::ofstream st(L + D + "_local/statTables.txt");
    std::map<std::string, uint64_t> rows;
    for (const VpTable& t : ps.tables)
      if (t.in_layout) rows[part_name(t.partition) + "-_3=" + ps.pred_storage[t.pidx]] += t.rows;
    for (auto& kv : rows) st << kv.first << " " << kv.second << "\n";
    if (!pstat || !st) throw DxError(DX_E_IO, "writing partition_stat / statTables failed");
  }
  // node index (DAP.scala:402-412): Types.NodeIndex(uri, uriType)
  {
    const std::string ni = B + D + "_data/node_index_" + N;
    mkdirs(ni);
    LayoutFile f;
    f.path = ni + "/" + part_file(0);
    f.cols = {"uri", "uriType"};
    f.rows = ps.n_node;
    f.src = {{ps.node_s.p, nullptr}, {ps.node_o.p, nullptr}};
    files.push_back(std::move(f));
    success.push_back(ni + "/_SUCCESS");
  }
  // VP index (DAP.scala:450-501): one row per uri with the set of "s" / "o" + sanitised predicate it occurs with,
  // joined by commas (collect_set leaves the order open: byte order of the texts here)
  const size_t first_vpi_file = files.size();
  for (int q = 0; q < ps.n_partitions; ++q) {
    if (ps.part_base[q + 1] == ps.part_base[q]) continue;
    const std::string dir = B + D + "_data/classSubPart_index_" + N + "__" + part_name(q);
    mkdirs(dir);
    LayoutFile f;
    f.path = dir + "/" + part_file(0);
    f.cols = {"uri", "parts"};
    f.vpi_partition = q;
    files.push_back(std::move(f));
    success.push_back(dir + "/_SUCCESS");
  }
  tm.prepare_ms = now_ms() - t_begin;

  // ---- the pages, batch after batch --------------------------------------------------------------------------------
  const int id_bits = bits_for(d.n_terms);
  uint64_t batch_values = 1ull << 27;
  if (const char* e = getenv("DX_LAYOUT_BATCH_VALUES")) batch_values = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  DevBuf<int64_t> d_pred_terms;
  upload(d_pred_terms, pool, ps.pred_terms, s);
    size_t f0 = 0;
    while (f0 < files.size()) {
      const double t_enc0 = now_ms();
      // files [f0, f1): as many as keep the batch's values under the limit; the VP index files go together, last
      size_t f1 = f0;
      uint64_t vals = 0;
      bool with_vpi = false;
      while (f1 < files.size()) {
        const LayoutFile& f = files[f1];
        if (f.vpi_partition >= 0) {
          if (f1 > f0 && f1 == first_vpi_file && vals) break;  // their own batch unless nothing came before
          with_vpi = true;
          ++f1;
          continue;
        }
        const uint64_t v = f.rows * f.src.size();
        if (f1 > f0 && vals + v > batch_values) break;
        vals += v;
        ++f1;
      }
      // column chunks with dictionary pages, their values' places, tiles of values
      std::vector<EncChunk> chunks;
      std::vector<EncTile> tiles;
      uint64_t n_vals = 0;
      for (size_t fi = f0; fi < f1; ++fi) {
        LayoutFile& f = files[fi];
        if (f.src.empty()) continue;
        for (uint64_t r0 = 0; r0 < f.rows; r0 += kParquetRowGroupRows) {
          const uint64_t m = std::min<uint64_t>(kParquetRowGroupRows, f.rows - r0);
          f.group_rows.push_back(m);
          for (const LayoutFile::Col& c : f.src) {
            EncChunk ch{};
            ch.kind = c.pidx ? kChunkPidx : kChunkIds;
            ch.src = c.pidx ? static_cast<const void*>(c.pidx + r0) : static_cast<const void*>(c.ids + r0);
            ch.n = (uint32_t)m;
            ch.first_val = n_vals;
            n_vals += m;
            f.chunk_ids.push_back((uint32_t)chunks.size());
            for (uint32_t v = 0; v < m; v += kTileVals) tiles.push_back({(uint32_t)chunks.size(), v});
            chunks.push_back(ch);
          }
        }
      }
      const uint32_t n_dict_chunks = (uint32_t)chunks.size();
      if (n_vals >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 values in one batch of the layout (lower DX_LAYOUT_BATCH_VALUES)");
      DevBuf<EncChunk> d_chunks;
      DevBuf<EncTile> d_tiles;
      upload(d_chunks, pool, chunks, s);
      upload(d_tiles, pool, tiles, s);
      DevBuf<uint64_t> keys(pool, n_vals), entries;
      DevBuf<uint32_t> bad(pool, 1);
      DX_HIP(hipMemsetAsync(bad.p, 0, 4, s));
      if (!tiles.empty())
        hipLaunchKernelGGL(k_enc_keys, dim3((unsigned)tiles.size()), dim3(256), 0, s, d_tiles.p, d_chunks.p, d_pred_terms.p, id_bits, d.n_terms, keys.p, bad.p);

      // the VP index of all partitions: entries sorted by (partition, uri, text), rows, byte positions
      VpiState vp;
      if (with_vpi) {
        std::vector<std::string> names;  // distinct directory names in byte order
        for (const std::string& nm : ps.pred_storage) names.push_back(nm);
        std::sort(names.begin(), names.end());
        names.erase(std::unique(names.begin(), names.end()), names.end());
        vp.n_names = (uint32_t)names.size();
        std::vector<uint32_t> rank(ps.pred_storage.size()), noff(names.size() + 1, 0);
        std::vector<uint8_t> blob;
        for (size_t i = 0; i < names.size(); ++i) {
          blob.insert(blob.end(), names[i].begin(), names[i].end());
          noff[i + 1] = (uint32_t)blob.size();
        }
        blob.resize(blob.size() + 16, 0);  // the wide copies may read past the last name
        for (size_t i = 0; i < rank.size(); ++i) rank[i] = (uint32_t)(std::lower_bound(names.begin(), names.end(), ps.pred_storage[i]) - names.begin());
        vp.bits_code = bits_for(2ull * std::max<uint32_t>(1, vp.n_names));
        vp.bits_term = ps.vpi_bits_term;
        const int bits_part = bits_for((uint64_t)ps.n_partitions);
        if (vp.bits_code + vp.bits_term + bits_part > 64) throw DxError(DX_E_RANGE, "VP index entries do not fit 64 bits");
        const uint64_t na = ps.vpi_s_base.empty() ? 0 : ps.vpi_s_base.back(), nb = ps.vpi_o_base.empty() ? 0 : ps.vpi_o_base.back();
        if (na + nb >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 VP index entries");
        DevBuf<uint32_t> d_rank;
        upload(d_rank, pool, rank, s);
        upload(vp.name_off, pool, noff, s);
        upload(vp.names, pool, blob, s);
        DevBuf<uint64_t> vkeys(pool, na + nb);
        if (na + nb)
          hipLaunchKernelGGL(k_vpi_keys, dim3(blocks_for(na + nb)), dim3(256), 0, s, ps.vpi_s.p, na, ps.vpi_o.p, nb, ps.vpi_bits_pred, vp.bits_code, d_rank.p,
                             vp.n_names, vkeys.p);
        vp.n = sort_distinct(pool, vkeys, na + nb, vp.bits_code + vp.bits_term + bits_part, vp.e, 0, s);
        vp.head = DevBuf<uint8_t>(pool, vp.n);
        DevBuf<uint32_t> ebytes(pool, vp.n), d_rows(pool, 1);
        vp.rowidx = DevBuf<uint32_t>(pool, vp.n);
        vp.epos = DevBuf<uint64_t>(pool, vp.n);
        vp.etotal = DevBuf<uint64_t>(pool, 1);
        if (vp.n) hipLaunchKernelGGL(k_vpi_entry_bytes, dim3(blocks_for(vp.n)), dim3(256), 0, s, vp.e.p, vp.n, vp.bits_code, vp.name_off.p, vp.n_names, vp.head.p, ebytes.p);
        exclusive_scan_u8_u32(pool, vp.head.p, vp.rowidx.p, vp.n, d_rows.p, s);
        exclusive_scan_u32_u64(pool, ebytes.p, vp.epos.p, vp.n, vp.etotal.p, s);
        DX_HIP(hipMemcpyAsync(&vp.n_rows, d_rows.p, 4, hipMemcpyDeviceToHost, s));
        sync(s);
        vp.rowstart = DevBuf<uint32_t>(pool, (uint64_t)vp.n_rows + 1);
        vp.rowbase = DevBuf<uint32_t>(pool, ps.n_partitions + 1);
        hipLaunchKernelGGL(k_vpi_rowstart, dim3(blocks_for(std::max<uint32_t>(vp.n, 1))), dim3(256), 0, s, vp.head.p, vp.rowidx.p, vp.n, vp.n_rows, vp.rowstart.p);
        hipLaunchKernelGGL(k_vpi_rowbase, dim3(blocks_for(ps.n_partitions + 1, 64)), dim3(64), 0, s, vp.e.p, vp.n, vp.bits_code + vp.bits_term, ps.n_partitions,
                           vp.rowidx.p, vp.n_rows, vp.rowbase.p);
        vp.h_rowbase.resize(ps.n_partitions + 1);
        DX_HIP(hipMemcpyAsync(vp.h_rowbase.data(), vp.rowbase.p, vp.h_rowbase.size() * 4, hipMemcpyDeviceToHost, s));
        sync(s);
      }

      // distinct (chunk, term) keys = every chunk's dictionary; behind them the uri columns of the VP index files
      const int chunk_bits_sorted = bits_for(std::max<uint32_t>(1, n_dict_chunks));
      if (id_bits + chunk_bits_sorted > 64) throw DxError(DX_E_RANGE, "chunk number and term id do not fit 64 bits");
      const uint32_t n_sorted = sort_distinct(pool, keys, n_vals, id_bits + chunk_bits_sorted, entries, vp.n_rows, s);
      keys.reset();
      const uint64_t n_entries64 = (uint64_t)n_sorted + vp.n_rows;
      if (n_entries64 >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 page entries in one batch of the layout");
      const uint32_t n_entries = (uint32_t)n_entries64;
      std::vector<uint32_t> chunk0(ps.n_partitions, 0);
      if (with_vpi) {
        for (size_t fi = f0; fi < f1; ++fi) {
          LayoutFile& f = files[fi];
          if (f.vpi_partition < 0) continue;
          const int q = f.vpi_partition;
          const uint32_t r0 = vp.h_rowbase[q], rows = vp.h_rowbase[q + 1] - r0;
          f.rows = rows;
          chunk0[q] = (uint32_t)chunks.size();
          for (uint64_t g0 = 0; g0 < rows; g0 += kParquetRowGroupRows) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(kParquetRowGroupRows, rows - g0);
            f.group_rows.push_back(m);
            EncChunk u{}, p{};
            u.kind = kChunkPlain; u.n = m; u.first = n_sorted + r0 + (uint32_t)g0; u.count = m;
            p.kind = kChunkParts; p.n = m; p.first = r0 + (uint32_t)g0; p.count = m;
            f.chunk_ids.push_back((uint32_t)chunks.size());
            chunks.push_back(u);
            f.chunk_ids.push_back((uint32_t)chunks.size());
            chunks.push_back(p);
          }
        }
        if (id_bits + bits_for(chunks.size()) > 64) throw DxError(DX_E_RANGE, "chunk number and term id do not fit 64 bits");
        upload(vp.chunk0, pool, chunk0, s);
        if (vp.n)
          hipLaunchKernelGGL(k_vpi_uri_entries, dim3(blocks_for(vp.n)), dim3(256), 0, s, vp.e.p, vp.n, vp.head.p, vp.rowidx.p, vp.bits_code, vp.bits_term,
                             vp.rowbase.p, vp.chunk0.p, id_bits, entries.p + n_sorted);
        upload(d_chunks, pool, chunks, s);
        if (snappy && vp.n_rows && vp.n_names < 32768) {  // (a row's copied pieces are counted in 16 bits)
          vp.parts_packed = true;
          vp.rplan = DevBuf<uint2>(pool, vp.n_rows);
          vp.rpos = DevBuf<uint64_t>(pool, vp.n_rows);
          vp.rtotal = DevBuf<uint64_t>(pool, 1);
          DevBuf<uint32_t> rlen(pool, vp.n_rows);
          hipLaunchKernelGGL(k_vpi_rows_plan, dim3(blocks_for(vp.n_rows)), dim3(256), 0, s, vp.e.p, vp.n, vp.rowstart.p, vp.n_rows, vp.epos.p, vp.etotal.p, vp.bits_code,
                             vp.bits_term, vp.rowbase.p, vp.name_off.p, vp.n_names, vp.rplan.p, rlen.p);
          exclusive_scan_u32_u64(pool, rlen.p, vp.rpos.p, vp.n_rows, vp.rtotal.p, s);
        }
      }
      const uint32_t n_chunks = (uint32_t)chunks.size();
      DevBuf<uint32_t> len4(pool, n_entries);
      DevBuf<uint64_t> dpos(pool, n_entries), dtotal(pool, 1);
      if (n_entries) hipLaunchKernelGGL(k_entry_lens, dim3(blocks_for(n_entries)), dim3(256), 0, s, entries.p, n_entries, (1ull << id_bits) - 1, d.pool_off.p, len4.p);
      exclusive_scan_u32_u64(pool, len4.p, dpos.p, n_entries, dtotal.p, s);
      DevBuf<uint64_t> cpos, ctotal;  // snappy: where every entry's elements begin
      if (snappy) {
        cpos = DevBuf<uint64_t>(pool, n_entries);
        ctotal = DevBuf<uint64_t>(pool, 1);
        if (n_entries) hipLaunchKernelGGL(k_entry_packed_lens, dim3(blocks_for(n_entries)), dim3(256), 0, s, entries.p, n_entries, id_bits, d.pool.p, d.pool_off.p, len4.p);
        exclusive_scan_u32_u64(pool, len4.p, cpos.p, n_entries, ctotal.p, s);
      }
      DevBuf<ChunkSize> d_sizes(pool, n_chunks);
      std::vector<ChunkSize> sizes(n_chunks);
      uint32_t h_bad = 0;
      if (n_chunks) {
        hipLaunchKernelGGL(k_chunk_sizes, dim3(blocks_for(n_chunks)), dim3(256), 0, s, d_chunks.p, n_chunks, entries.p, n_sorted, id_bits, dpos.p, n_entries, dtotal.p,
                           vp.rowstart.p, vp.epos.p, vp.n, vp.etotal.p, snappy ? cpos.p : nullptr, ctotal.p, vp.parts_packed ? vp.rpos.p : nullptr, vp.rtotal.p, vp.n_rows,
                           d_sizes.p);
        DX_HIP(hipMemcpyAsync(sizes.data(), d_sizes.p, n_chunks * sizeof(ChunkSize), hipMemcpyDeviceToHost, s));
      }
      DX_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, s));
      sync(s);
      if (h_bad) throw DxError(DX_E_RANGE, "term id outside the dictionary while writing the layout");

      // every file's size (the Parquet files are laid out from their pages' sizes), then its place in the image: the
      // largest first, because a file is written by one thread from its first byte to its last and the long ones must
      // start early
      std::vector<PqFilePlan> plans(f1 - f0);
      for (size_t fi = f0; fi < f1; ++fi) {
        LayoutFile& f = files[fi];
        if (f.cols.empty()) {
          f.size = 0;
          for (const LayoutFile::Raw& r : f.raw) f.size += r.bytes;
          continue;
        }
        std::vector<PqChunkIn> in;
        for (uint32_t k : f.chunk_ids) {
          PqChunkIn c;
          c.num_values = chunks[k].n;
          if (chunks[k].kind == kChunkPlain || chunks[k].kind == kChunkParts) { c.plain = true; c.plain_bytes = sizes[k].bytes; }
          else { c.n_dict = sizes[k].count; c.dict_bytes = sizes[k].bytes; }
          c.snappy = snappy;
          c.terms_packed = snappy && (chunks[k].kind != kChunkParts || vp.parts_packed) ? sizes[k].packed_bytes : 0;
          in.push_back(c);
        }
        try {
          plans[fi - f0] = plan_parquet_file(f.cols, f.group_rows, in);
        } catch (const DxError& e) {
          throw DxError(e.code, std::string(e.what()) + " in " + f.path);
        }
        f.size = plans[fi - f0].size;
      }
      // the files are dealt to the lanes, the largest to the lane with the fewest bytes so far; a lane's files follow one
      // another in the image
      std::vector<size_t> order(f1 - f0);
      for (size_t i = 0; i < order.size(); ++i) order[i] = f0 + i;
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return files[x].size > files[y].size; });
      const unsigned n_lanes = lanes.lanes();
      std::vector<LaneJob> jobs(n_lanes);
      {
        using Load = std::pair<uint64_t, unsigned>;  // (bytes, lane)
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> least;
        for (unsigned l = 0; l < n_lanes; ++l) least.push({0, l});
        for (size_t fi : order) {
          const Load top = least.top();
          least.pop();
          LaneJob& j = jobs[top.second];
          j.files.push_back({&files[fi].path, j.bytes, files[fi].size});
          j.bytes = (j.bytes + files[fi].size + 63) & ~63ull;
          least.push({j.bytes, top.second});
        }
      }
      uint64_t image_bytes = 0;
      std::vector<uint64_t> lane_at(n_lanes);
      for (unsigned l = 0; l < n_lanes; ++l) {
        lane_at[l] = image_bytes;
        image_bytes += jobs[l].bytes;
      }
      {
        std::unordered_map<const std::string*, uint64_t> at_of;
        for (unsigned l = 0; l < n_lanes; ++l)
          for (const LaneFile& lf : jobs[l].files) at_of[lf.path] = lane_at[l] + lf.at;
        for (size_t fi = f0; fi < f1; ++fi) files[fi].at = at_of[&files[fi].path];
      }
      std::vector<uint8_t> blob;          // the bytes made here, one piece after another
      std::vector<CopySeg> small, words;  // blob -> image; device arrays -> image
      std::vector<EncTile> word_tiles;
      for (size_t fi = f0; fi < f1; ++fi) {
        LayoutFile& f = files[fi];
        if (f.cols.empty()) {
          uint64_t at = 0;
          for (const LayoutFile::Raw& r : f.raw) {
            if (r.dev) {
              for (uint64_t b = 0, t = 0; b < r.bytes; b += 16384, ++t) word_tiles.push_back({(uint32_t)words.size(), (uint32_t)t});
              if (r.bytes > (16384ull << 32)) throw DxError(DX_E_RANGE, "a raw column of more than 64 TiB");
              words.push_back({(uint64_t)(uintptr_t)r.dev, f.at + at, r.bytes});
            } else {
              small.push_back({blob.size(), f.at + at, r.host.size()});
              blob.insert(blob.end(), r.host.begin(), r.host.end());
            }
            at += r.bytes;
          }
        } else {
          const PqFilePlan& plan = plans[fi - f0];
          for (size_t i = 0; i < f.chunk_ids.size(); ++i) {
            EncChunk& c = chunks[f.chunk_ids[i]];
            const ChunkSize& z = sizes[f.chunk_ids[i]];
            const PqChunkOut& o = plan.chunks[i];
            c.first = z.first;
            c.count = z.count;
            c.width = o.width;
            const uint64_t terms_from = snappy && (c.kind != kChunkParts || vp.parts_packed) ? z.packed_start : z.start;  // where the page's bytes begin in the list the kernels count in
            if (c.kind == kChunkPlain || c.kind == kChunkParts) c.adj = f.at + o.data_at - terms_from;
            else { c.adj = f.at + o.dict_at - terms_from; c.data_at = f.at + o.data_at; }
          }
          for (auto& h : plan.host) {
            small.push_back({blob.size(), f.at + h.first, h.second.size()});
            blob.insert(blob.end(), h.second.begin(), h.second.end());
          }
        }
      }
      plans.clear();
      upload(d_chunks, pool, chunks, s);
      DevBuf<uint8_t> image(pool, image_bytes), d_blob;
      DevBuf<CopySeg> d_small, d_words;
      DevBuf<EncTile> d_word_tiles;
      upload(d_blob, pool, blob, s);
      upload(d_small, pool, small, s);
      upload(d_words, pool, words, s);
      upload(d_word_tiles, pool, word_tiles, s);
      if (!small.empty()) hipLaunchKernelGGL(k_copy_small, dim3(blocks_for(small.size(), 4)), dim3(256), 0, s, d_small.p, (uint32_t)small.size(), d_blob.p, image.p);
      if (!word_tiles.empty()) hipLaunchKernelGGL(k_copy_words, dim3((unsigned)word_tiles.size()), dim3(256), 0, s, d_word_tiles.p, d_words.p, image.p);
      if (n_entries && snappy)
        hipLaunchKernelGGL(k_enc_terms_packed, dim3(blocks_for(n_entries)), dim3(256), 0, s, entries.p, n_entries, id_bits, d_chunks.p, cpos.p, d.pool.p, d.pool_off.p, image.p);
      else if (n_entries)
        hipLaunchKernelGGL(k_enc_terms, dim3(blocks_for(4ull * n_entries)), dim3(256), 0, s, entries.p, n_entries, id_bits, d_chunks.p, dpos.p, d.pool.p, d.pool_off.p, image.p);
      if (!tiles.empty())
        hipLaunchKernelGGL(k_enc_indices, dim3((unsigned)tiles.size()), dim3(256), 0, s, d_tiles.p, d_chunks.p, d_pred_terms.p, entries.p, id_bits, image.p);
      if (vp.n && vp.parts_packed)
        hipLaunchKernelGGL(k_vpi_parts_packed, dim3(blocks_for(4ull * vp.n)), dim3(256), 0, s, vp.e.p, vp.n, vp.head.p, vp.rowidx.p, vp.rowstart.p, vp.epos.p,
                           vp.etotal.p, vp.bits_code, vp.bits_term, vp.rowbase.p, vp.chunk0.p, d_chunks.p, vp.names.p, vp.name_off.p, vp.n_names, vp.rplan.p,
                           vp.rpos.p, image.p);
      else if (vp.n)
        hipLaunchKernelGGL(k_vpi_parts, dim3(blocks_for(4ull * vp.n)), dim3(256), 0, s, vp.e.p, vp.n, vp.head.p, vp.rowidx.p, vp.rowstart.p, vp.epos.p, vp.etotal.p,
                           vp.bits_code, vp.bits_term, vp.rowbase.p, vp.chunk0.p, d_chunks.p, vp.names.p, vp.name_off.p, vp.n_names, image.p);
      DX_HIP(hipGetLastError());
      sync(s);
      tm.encode_ms += now_ms() - t_enc0;

      // the image comes to the host lane by lane in turn through the ring; the lanes' threads write what has landed
      const double t_d2h0 = now_ms();
      if (batch_open) { lanes.end(); batch_open = false; }  // the batch before has its files
      std::vector<uint64_t> lane_bytes(n_lanes);
      for (unsigned l = 0; l < n_lanes; ++l) lane_bytes[l] = jobs[l].bytes;
      lanes.begin(std::move(jobs));
      batch_open = true;
      for (uint64_t k = 0;; ++k) {
        bool any = false;
        for (unsigned l = 0; l < n_lanes; ++l) {
          const uint64_t a = k * Lanes::kChunk;
          if (a >= lane_bytes[l]) continue;
          any = true;
          const uint64_t n = std::min(Lanes::kChunk, lane_bytes[l] - a);
          unsigned slot = 0;
          uint8_t* dst = lanes.acquire(&slot);
          DX_HIP(hipMemcpyAsync(dst, image.p + lane_at[l] + a, n, hipMemcpyDeviceToHost, s));
          DX_HIP(hipEventRecord(lanes.event_of(slot), s));
          lanes.publish(l, slot, n);
        }
        if (!any) break;
      }
      sync(s);  // the image goes back to the pool when this batch's variables do
      tm.d2h_ms += now_ms() - t_d2h0;
      f0 = f1;
    }
    if (batch_open) { batch_open = false; lanes.end(); }
  } catch (...) {
    if (batch_open) lanes.abandon();
    lanes.stop();
    // an unfinished layout must not look finished, and a second attempt must not be refused: what this call made goes
    for (const LayoutFile& f : files) (void)unlink(f.path.c_str());
    (void)rename(clusters.c_str(), (clusters + ".incomplete." + std::to_string((unsigned long long)time(nullptr))).c_str());
    throw;
  }
  lanes.stop();
  for (const std::string& p : success) touch(p);
  tm.write_busy_ms = lanes.busy_ms;
  tm.write_wall_ms = lanes.last_ms - lanes.first_ms;
  tm.bytes = lanes.bytes;
  tm.total_ms = now_ms() - t_begin;
}

}  // namespace dx
