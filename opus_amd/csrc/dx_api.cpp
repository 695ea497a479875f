// C ABI of libdiaeresis_hip.so: context, accessors, on-disk layout, query entry points.
#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <unordered_map>

#include "dx_device.h"
#include "dx_sql.h"
#include "dx_state.h"

namespace dx {
uint64_t hash_bytes_host(const uint8_t* p, size_t n, uint64_t seed) { return hash_bytes(p, n, seed); }

namespace {

template <typename F>
int guard(dx_ctx* ctx, F&& f) {
  if (!ctx) return DX_E_INVAL;
  try {
    DX_HIP(hipSetDevice(ctx->device));
    f();
    return DX_OK;
  } catch (const DxError& e) {
    ctx->err = e.what();
    return e.code;
  } catch (const HipError& e) {
    ctx->err = e.what();
    return DX_E_HIP;
  } catch (const std::bad_alloc&) {
    ctx->err = "host allocation failed";
    return DX_E_NOMEM;
  } catch (const std::exception& e) {
    ctx->err = e.what();
    return DX_E_INVAL;
  }
}

// Runs independent pieces of host work (the files of a layout) on a pool of threads, the largest first; the first
// failure is reported after all threads have stopped.
void run_jobs(std::vector<std::pair<uint64_t, std::function<void()>>>& jobs) {
  std::sort(jobs.begin(), jobs.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  unsigned nthreads = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("DX_WRITE_THREADS")) nthreads = (unsigned)std::max(1, atoi(e));
  nthreads = (unsigned)std::min<size_t>(nthreads, std::max<size_t>(jobs.size(), 1));
  std::atomic<size_t> next{0};
  std::mutex mu;
  std::exception_ptr failure;
  auto work = [&] {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= jobs.size()) return;
      try {
        jobs[i].second();
      } catch (...) {
        std::lock_guard<std::mutex> lock(mu);
        if (!failure) failure = std::current_exception();
        next.store(jobs.size());  // no new files after a failure
      }
      jobs[i].second = nullptr;  // the job's buffers go as soon as it is done
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (failure) std::rethrow_exception(failure);
}

std::vector<ClassIndexEntry> parse_class_index(const std::string& text, int nparts) {
  // Loader.loadClassIndexBal (Loader.scala:114-123): `class \t p1,p2,...`; a later line replaces an earlier one
  std::map<std::string, uint64_t> m;
  std::istringstream in(text);
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    const size_t tab = line.find('\t');
    const std::string cls = line.substr(0, tab);
    std::string rest;
    if (tab != std::string::npos)
      for (char c : line.substr(tab + 1))
        if (c != '\t') rest += c;  // tokens.drop(1).mkString
    uint64_t mask = 0;
    std::istringstream ps(rest);
    std::string tok;
    while (std::getline(ps, tok, ',')) {
      char* end = nullptr;
      errno = 0;
      const long v = strtol(tok.c_str(), &end, 10);
      if (tok.empty() || *end || errno) throw DxError(DX_E_MALFORMED, "class index: bad partition id '" + tok + "' for " + cls);
      if (v < 0 || v >= nparts)
        throw DxError(DX_E_RANGE, "class index: partition " + tok + " outside [0, " + std::to_string(nparts) + ") for " + cls);
      mask |= 1ull << v;
    }
    m[cls] = mask;
  }
  std::vector<ClassIndexEntry> out;
  for (auto& kv : m) out.push_back({kv.first, kv.second});
  return out;
}

std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw DxError(DX_E_NOENT, "cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

}  // namespace
void mkdirs(const std::string& path) {
  std::string cur;
  for (size_t i = 0; i < path.size(); ++i) {
    cur += path[i];
    if (path[i] == '/' || i + 1 == path.size()) {
      if (cur == "/" || cur.empty()) continue;
      if (mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) throw DxError(DX_E_IO, "mkdir " + cur + ": " + strerror(errno));
    }
  }
}

namespace {
struct ColHeader {
  char magic[8];
  uint64_t rows;
  uint64_t ncols;
};
const char kMagic[8] = {'D', 'X', 'C', 'O', 'L', '1', 0, 0};

void write_cols(const std::string& path, uint64_t rows, const std::vector<const int64_t*>& host_cols) {
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) throw DxError(DX_E_IO, "cannot create " + path);
  ColHeader h;
  memcpy(h.magic, kMagic, 8);
  h.rows = rows;
  h.ncols = host_cols.size();
  f.write((const char*)&h, sizeof h);
  for (auto* c : host_cols) f.write((const char*)c, (std::streamsize)(rows * 8));
  if (!f) throw DxError(DX_E_IO, "write failed: " + path);
}
}  // namespace
void touch(const std::string& path) { std::ofstream f(path, std::ios::trunc); }

std::string part_name(int i) {
  char b[16];
  snprintf(b, sizeof b, "%05d", i);
  return b;
}
std::string norm_base(const char* base) {
  std::string b = base ? base : "";
  if (!b.empty() && b.back() != '/') b += '/';  // DAP.scala:67-68
  return b;
}
namespace {
std::string view_name(const std::string& key) {  // QueryProcessor.scala:216
  std::string v = "table";
  for (char c : key) {
    if (c == '-') v += '_';
    else if (c == '=') v += "_E_";
    else v += c;
  }
  size_t b = 5, e = v.size();
  while (e > b && (unsigned char)v[e - 1] <= ' ') --e;
  return v.substr(0, e);
}

void fetch(void* dst, const void* src, size_t bytes) {
  if (bytes) DX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
}

// java.lang.String.compareTo on UTF-8 bytes: code-unit order differs from byte order only between
// supplementary characters (lead byte 0xF0-0xF4, surrogates 0xD800-0xDFFF in UTF-16) and U+E000-U+FFFF
// (lead byte 0xEE/0xEF).
inline int utf16_rank(unsigned char c) { return (c == 0xEE || c == 0xEF) ? c + 0x10 : c; }
bool utf16_less(const uint8_t* a, size_t la, const uint8_t* b, size_t lb) {
  const size_t n = std::min(la, lb);
  for (size_t i = 0; i < n; ++i)
    if (a[i] != b[i]) return utf16_rank(a[i]) < utf16_rank(b[i]);
  if (la == lb) return false;
  // a proper prefix is smaller -- but a supplementary character counts two units, which never changes
  // which string is the prefix
  return la < lb;
}

}  // namespace

void throw_if_any_failed(dx_ctx* ctx, const LocalError& mine, const std::vector<int>& codes) {
  (void)ctx;
  int worst = -1;
  for (int c : codes)
    if (c != 0 && (worst < 0 || err_slot(c) < worst)) worst = err_slot(c);
  if (mine && (worst < 0 || err_slot(mine.code) < worst)) worst = err_slot(mine.code);
  if (worst < 0) return;
  const int code = err_code_of_slot(worst);
  if (mine && err_slot(mine.code) == worst) throw AgreedError(code, mine.msg);
  throw AgreedError(code, std::string("another rank failed (error ") + std::to_string(code) + "); this rank stops with it" +
                          (mine ? "; its own failure: " + mine.msg : std::string()));
}

void agree_or_throw(dx_ctx* ctx, const LocalError& err) {
  const dx_comm& cm = ctx->comm;
  if (cm.world <= 1) {
    if (err) throw AgreedError(err.code, err.msg);
    return;
  }
  uint64_t v[kErrSlots] = {0};
  if (err) v[err_slot(err.code)] = 1;
  if (cm.allreduce_sum(cm.user, v, kErrSlots) != 0) throw DxError(DX_E_IO, "collective failed: allreduce_sum (agreement on errors)");
  std::vector<int> codes;
  for (int s = 0; s < kErrSlots; ++s)
    if (v[s]) codes.push_back(err_code_of_slot(s));
  throw_if_any_failed(ctx, err, codes);
}
}  // namespace dx

using namespace dx;

extern "C" {

dx_ctx* dx_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  auto* ctx = new dx_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return nullptr;
  }
  try {  // what the query path would otherwise create on first use; dx_query_count_noalloc must find it ready
    query_context_setup(ctx);
    ctx->noalloc_scratch = noalloc_scratch_of_current_device(&ctx->noalloc_scratch_bytes);
  } catch (const std::exception&) {
    dx_free(ctx);
    return nullptr;
  }
  return ctx;
}

void dx_free(dx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  rccl_comm_destroy(ctx);
  ctx->catalog.clear(); ctx->union_tables.clear();
  ctx->part = PartitionState();
  ctx->dict = Dictionary();
  ctx->qws.reset();
  ctx->pool.trim();
  if (ctx->q_host) (void)hipHostFree(ctx->q_host);
  if (ctx->q_counters) (void)hipFree(ctx->q_counters);
  for (int b = 0; b < kFileStages; ++b) {
    if (ctx->file_stage[b]) (void)hipHostFree(ctx->file_stage[b]);
    if (ctx->file_stage_ev[b]) (void)hipEventDestroy(ctx->file_stage_ev[b]);
  }
  if (ctx->part_host) (void)hipHostFree(ctx->part_host);
  if (ctx->fetch_seq) (void)hipHostFree(ctx->fetch_seq);
  for (hipEvent_t e : ctx->part_events) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* dx_last_error(dx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

const char* dx_part_stage_name(int i) { return (i >= 0 && i < 12) ? kStageNames[i] : ""; }

int dx_partition_device(dx_ctx* ctx, const void* d_text, uint64_t n_bytes, const char* class_index_text, int n_partitions,
                        dx_part_stats* stats) {
  return guard(ctx, [&] {
    if (!class_index_text || (!d_text && n_bytes)) throw DxError(DX_E_INVAL, "null argument");
    if (n_partitions < 1 || n_partitions > DX_MAX_PARTITIONS) throw DxError(DX_E_INVAL, "n_partitions must be in [1, 64]");
    const auto classes = parse_class_index(class_index_text, n_partitions);
    run_partition(ctx, (const uint8_t*)d_text, n_bytes, classes, n_partitions);
    if (stats) *stats = ctx->part.stats;
  });
}

int dx_partition_file(dx_ctx* ctx, const char* instances_path, const char* class_index_path, int n_partitions,
                      dx_part_stats* stats) {
  return guard(ctx, [&] {
    if (!instances_path) throw DxError(DX_E_INVAL, "null path");
    const std::string ci = class_index_path ? read_file(class_index_path) : std::string();
    const int fd = open(instances_path, O_RDONLY);
    if (fd < 0) throw DxError(DX_E_NOENT, std::string("cannot open ") + instances_path);
    struct FdGuard { int fd; ~FdGuard() { close(fd); } } fd_guard{fd};
    struct stat st;
    if (fstat(fd, &st) != 0) throw DxError(DX_E_IO, std::string("cannot stat ") + instances_path);
    const uint64_t n = (uint64_t)st.st_size;
    DevBuf<uint8_t> text(ctx->pool, n + 16);
    // Streamed through pinned staging buffers (kept by the context), so the file never has to fit in host memory.
    // Several threads read: one thread's read() from the page cache runs at about 20 GB/s, less than half of what the
    // link to the device takes. Piece i goes through buffer i % kFileStages; its reader waits for the upload of the
    // piece that used the buffer before, this thread queues the uploads in file order.
    const size_t chunk = 32u << 20;
    for (int b = 0; b < kFileStages; ++b) {
      if (!ctx->file_stage[b]) DX_HIP(hipHostMalloc(&ctx->file_stage[b], chunk, hipHostMallocDefault));
      if (!ctx->file_stage_ev[b]) DX_HIP(hipEventCreateWithFlags(&ctx->file_stage_ev[b], hipEventDisableTiming));
    }
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    unsigned n_readers = (unsigned)std::min<uint64_t>(std::min<unsigned>(kFileStages, std::max(1u, std::thread::hardware_concurrency())), std::max<uint64_t>(n_chunks, 1));
    if (const char* e = getenv("DX_READ_THREADS")) n_readers = (unsigned)std::max(1, std::min(atoi(e), kFileStages));
    std::vector<int> state(n_chunks, 0);  // 1: read into its buffer, 2: its upload is queued, -1: failed
    std::mutex mu;
    std::condition_variable cv;
    bool give_up = false;
    const int device = ctx->device;
    auto reader = [&](unsigned t) {
      (void)hipSetDevice(device);
      for (uint64_t i = t; i < n_chunks; i += n_readers) {
        const int b = (int)(i % kFileStages);
        if (i >= (uint64_t)kFileStages) {
          std::unique_lock<std::mutex> l(mu);
          cv.wait(l, [&] { return give_up || state[i - kFileStages] == 2; });
          if (give_up) return;
          l.unlock();
          if (hipEventSynchronize(ctx->file_stage_ev[b]) != hipSuccess) {
            std::lock_guard<std::mutex> g(mu);
            state[i] = -1;
            cv.notify_all();
            return;
          }
        }
        const uint64_t at = i * chunk, want = std::min<uint64_t>(chunk, n - at);
        uint64_t got = 0;
        while (got < want) {
          const ssize_t r = pread(fd, (char*)ctx->file_stage[b] + got, want - got, (off_t)(at + got));
          if (r < 0 && errno == EINTR) continue;
          if (r <= 0) break;
          got += (uint64_t)r;
        }
        std::lock_guard<std::mutex> g(mu);
        state[i] = got == want ? 1 : -1;
        cv.notify_all();
        if (got != want) return;
      }
    };
    std::vector<std::thread> readers;
    for (unsigned t = 0; t < n_readers; ++t) readers.emplace_back(reader, t);
    bool failed = false;
    try {
      for (uint64_t i = 0; i < n_chunks && !failed; ++i) {
        {
          std::unique_lock<std::mutex> l(mu);
          cv.wait(l, [&] { return state[i] != 0; });
          if (state[i] < 0) { failed = true; break; }
        }
        const int b = (int)(i % kFileStages);
        const uint64_t at = i * chunk, want = std::min<uint64_t>(chunk, n - at);
        DX_HIP(hipMemcpyAsync(text.p + at, ctx->file_stage[b], want, hipMemcpyHostToDevice, ctx->stream));
        DX_HIP(hipEventRecord(ctx->file_stage_ev[b], ctx->stream));
        std::lock_guard<std::mutex> g(mu);
        state[i] = 2;
        cv.notify_all();
      }
    } catch (...) {
      { std::lock_guard<std::mutex> g(mu); give_up = true; }
      cv.notify_all();
      for (auto& t : readers) t.join();
      (void)hipStreamSynchronize(ctx->stream);
      throw;
    }
    { std::lock_guard<std::mutex> g(mu); give_up = true; }
    cv.notify_all();
    for (auto& t : readers) t.join();
    if (failed) {
      (void)hipStreamSynchronize(ctx->stream);  // uploads of earlier pieces may still be running
      throw DxError(DX_E_IO, std::string("short read on ") + instances_path);
    }
    DX_HIP(hipStreamSynchronize(ctx->stream));
    const auto classes = parse_class_index(ci, n_partitions);
    run_partition(ctx, text.p, n, classes, n_partitions);
    if (stats) *stats = ctx->part.stats;
  });
}

// ---- dictionary -----------------------------------------------------------------------------------
int64_t dx_term_count(dx_ctx* ctx) { return ctx && ctx->dict.valid() ? (int64_t)ctx->dict.n_terms : -1; }

static void term_offsets(dx_ctx* ctx, const int64_t* ids, uint64_t n, std::vector<uint64_t>& begin, std::vector<uint64_t>& len) {
  const Dictionary& d = ctx->dict;
  if (!d.valid()) throw DxError(DX_E_INVAL, "no dictionary");
  std::vector<uint64_t> off(d.n_terms + 1);
  fetch(off.data(), d.pool_off.p, (d.n_terms + 1) * 8);
  begin.resize(n);
  len.resize(n);
  for (uint64_t i = 0; i < n; ++i) {
    if (ids[i] < 0 || (uint64_t)ids[i] >= d.n_terms) throw DxError(DX_E_RANGE, "term id out of range: " + std::to_string(ids[i]));
    begin[i] = off[ids[i]];
    len[i] = off[ids[i] + 1] - off[ids[i]];
  }
}

int dx_terms_bytes(dx_ctx* ctx, const int64_t* ids, uint64_t n, uint64_t* total_bytes) {
  return guard(ctx, [&] {
    std::vector<uint64_t> b, l;
    term_offsets(ctx, ids, n, b, l);
    uint64_t t = 0;
    for (auto x : l) t += x;
    *total_bytes = t;
  });
}

int dx_terms_decode(dx_ctx* ctx, const int64_t* ids, uint64_t n, char* bytes, uint64_t cap, uint64_t* offsets) {
  return guard(ctx, [&] {
    std::vector<uint64_t> b, l;
    term_offsets(ctx, ids, n, b, l);
    const Dictionary& d = ctx->dict;
    std::vector<uint8_t> pool(d.pool_off.n ? 0 : 0);
    uint64_t pool_bytes = 0;
    fetch(&pool_bytes, d.pool_off.p + d.n_terms, 8);
    uint64_t total = 0;
    for (auto x : l) total += x;
    if (total > cap) throw DxError(DX_E_RANGE, "decode buffer too small");
    // bulk requests copy the pool once; small ones copy term by term
    if (n > 4096) {
      pool.resize(pool_bytes);
      fetch(pool.data(), d.pool.p, pool_bytes);
    }
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; ++i) {
      offsets[i] = w;
      if (n > 4096) memcpy(bytes + w, pool.data() + b[i], l[i]);
      else fetch(bytes + w, d.pool.p + b[i], l[i]);
      w += l[i];
    }
    offsets[n] = w;
  });
}

int dx_terms_lookup(dx_ctx* ctx, const char* bytes, const uint64_t* offsets, uint64_t n, int64_t* ids) {
  return guard(ctx, [&] { dict_lookup(ctx, bytes, offsets, n, ids); });
}

// ---- partitions, tables, indexes -------------------------------------------------------------------
int dx_part_count(dx_ctx* ctx) { return ctx && ctx->part.valid ? ctx->part.n_partitions : -1; }

int64_t dx_part_rows(dx_ctx* ctx, int q) {
  if (!ctx || !ctx->part.valid || q < 0 || q >= ctx->part.n_partitions) return -1;
  return (int64_t)(ctx->part.part_base[q + 1] - ctx->part.part_base[q]);
}

int dx_part_fetch(dx_ctx* ctx, int q, int64_t* s, int64_t* o, int64_t* p) {
  return guard(ctx, [&] {
    const PartitionState& ps = ctx->part;
    if (!ps.valid || q < 0 || q >= ps.n_partitions) throw DxError(DX_E_INVAL, "no such partition");
    const uint64_t b = ps.part_base[q], n = ps.part_base[q + 1] - b;
    fetch(s, ps.row_s.p + b, n * 8);
    fetch(o, ps.row_o.p + b, n * 8);
    std::vector<uint32_t> pi(n);
    fetch(pi.data(), ps.row_p.p + b, n * 4);
    for (uint64_t i = 0; i < n; ++i) p[i] = ps.pred_terms[pi[i]];
  });
}

int dx_table_count(dx_ctx* ctx) { return ctx && ctx->part.valid ? (int)ctx->part.tables.size() : -1; }

int dx_table_info_at(dx_ctx* ctx, int i, dx_table_info* out) {
  return guard(ctx, [&] {
    const PartitionState& ps = ctx->part;
    if (!ps.valid || i < 0 || i >= (int)ps.tables.size() || !out) throw DxError(DX_E_INVAL, "no such table");
    const VpTable& t = ps.tables[i];
    out->partition = t.partition;
    out->in_layout = t.in_layout ? 1 : 0;
    out->predicate = t.pred_term;
    out->rows = (int64_t)t.rows;
  });
}

int dx_table_fetch(dx_ctx* ctx, int i, int64_t* s, int64_t* o) {
  return guard(ctx, [&] {
    const PartitionState& ps = ctx->part;
    if (!ps.valid || i < 0 || i >= (int)ps.tables.size()) throw DxError(DX_E_INVAL, "no such table");
    const VpTable& t = ps.tables[i];
    fetch(s, ps.row_s.p + t.begin, t.rows * 8);
    fetch(o, ps.row_o.p + t.begin, t.rows * 8);
  });
}

int64_t dx_node_index_rows(dx_ctx* ctx) { return ctx && ctx->part.valid ? (int64_t)ctx->part.n_node : -1; }

int dx_node_index_fetch(dx_ctx* ctx, int64_t* uri, int64_t* type) {
  return guard(ctx, [&] {
    const PartitionState& ps = ctx->part;
    if (!ps.valid) throw DxError(DX_E_INVAL, "no partition run");
    fetch(uri, ps.node_s.p, ps.n_node * 8);
    fetch(type, ps.node_o.p, ps.n_node * 8);
  });
}

int64_t dx_vp_index_rows(dx_ctx* ctx, int q) {
  if (!ctx || !ctx->part.valid || q < 0 || q >= ctx->part.n_partitions) return -1;
  const PartitionState& ps = ctx->part;
  return (int64_t)((ps.vpi_s_base[q + 1] - ps.vpi_s_base[q]) + (ps.vpi_o_base[q + 1] - ps.vpi_o_base[q]));
}

int dx_vp_index_fetch(dx_ctx* ctx, int q, int64_t* uri, int32_t* role, int64_t* predicate) {
  return guard(ctx, [&] {
    const PartitionState& ps = ctx->part;
    if (!ps.valid || q < 0 || q >= ps.n_partitions) throw DxError(DX_E_INVAL, "no such partition");
    const uint64_t bs = ps.vpi_s_base[q], ns = ps.vpi_s_base[q + 1] - bs, bo = ps.vpi_o_base[q], no = ps.vpi_o_base[q + 1] - bo;
    const uint64_t n = ns + no;
    std::vector<uint64_t> k(n);
    if (ns) fetch(k.data(), ps.vpi_s.p + bs, ns * 8);
    if (no) fetch(k.data() + ns, ps.vpi_o.p + bo, no * 8);
    const uint64_t pm = (1ull << ps.vpi_bits_pred) - 1, tm = (1ull << ps.vpi_bits_term) - 1;
    for (uint64_t i = 0; i < n; ++i) {
      predicate[i] = ps.pred_terms[k[i] & pm];
      role[i] = (int32_t)((k[i] >> ps.vpi_bits_pred) & 1);
      uri[i] = (int64_t)((k[i] >> (ps.vpi_bits_pred + 1)) & tm);
    }
  });
}

// ---- instance graph (export path; not on the timed path) -----------------------------------------------
static void instance_graph(dx_ctx* ctx, std::vector<int64_t>& vterms, std::vector<int64_t>* es, std::vector<int64_t>* ed,
                           std::vector<int64_t>* ep) {
  const PartitionState& ps = ctx->part;
  const Dictionary& d = ctx->dict;
  if (!ps.valid) throw DxError(DX_E_INVAL, "no partition run");
  std::vector<uint8_t> flags(d.n_terms);
  fetch(flags.data(), ps.term_flags.p, d.n_terms);
  std::vector<uint64_t> off(d.n_terms + 1);
  fetch(off.data(), d.pool_off.p, (d.n_terms + 1) * 8);
  std::vector<uint8_t> pool(off[d.n_terms]);
  fetch(pool.data(), d.pool.p, pool.size());
  vterms.clear();
  for (uint64_t t = 0; t < d.n_terms; ++t)
    if (flags[t] & 2) vterms.push_back((int64_t)t);
  // sortByKey().zipWithIndex (Preprocessor.scala:142-143): rank in String.compareTo order
  std::sort(vterms.begin(), vterms.end(), [&](int64_t a, int64_t b) {
    return utf16_less(pool.data() + off[a], off[a + 1] - off[a], pool.data() + off[b], off[b + 1] - off[b]);
  });
  if (!es) return;
  std::vector<int64_t> rank(d.n_terms, -1);
  for (size_t i = 0; i < vterms.size(); ++i) rank[vterms[i]] = (int64_t)i;
  auto edges = [&](auto zero, const void* dev_keys) {
    using K = decltype(zero);
    std::vector<K> keys(ps.n_uniq_slots);
    fetch(keys.data(), dev_keys, keys.size() * sizeof(K));
    const K tm = ((K)1 << ps.bits_term) - 1;
    for (size_t i = 0; i < keys.size(); ++i) {
      if (i && keys[i] == keys[i - 1]) continue;
      const uint32_t pidx = (uint32_t)(keys[i] >> (2 * ps.bits_term));
      const uint64_t s = (uint64_t)((keys[i] >> ps.bits_term) & tm), o = (uint64_t)(keys[i] & tm);
      if (pidx == ps.type_pidx || (flags[o] & 1)) continue;
      es->push_back(rank[s]);
      ed->push_back(rank[o]);
      ep->push_back(ps.pred_terms[pidx]);
    }
  };
  if (ps.wide_keys) edges((u128)0, ps.uniq_w.p); else edges((uint64_t)0, ps.uniq.p);
}

int dx_instance_graph_sizes(dx_ctx* ctx, int64_t* n_vertices, int64_t* n_edges) {
  return guard(ctx, [&] {
    if (!ctx->part.valid) throw DxError(DX_E_INVAL, "no partition run");
    // counted from what this context holds: with several ranks that is the rank's share of the edges
    std::vector<int64_t> v, es, ed, ep;
    instance_graph(ctx, v, &es, &ed, &ep);
    *n_vertices = (int64_t)v.size();
    *n_edges = (int64_t)es.size();
  });
}

int dx_instance_graph_fetch(dx_ctx* ctx, int64_t* vertex_terms, int64_t* edge_src, int64_t* edge_dst, int64_t* edge_pred) {
  return guard(ctx, [&] {
    std::vector<int64_t> v, es, ed, ep;
    instance_graph(ctx, v, &es, &ed, &ep);
    memcpy(vertex_terms, v.data(), v.size() * 8);
    memcpy(edge_src, es.data(), es.size() * 8);
    memcpy(edge_dst, ed.data(), ed.size() * 8);
    memcpy(edge_pred, ep.data(), ep.size() * 8);
  });
}

// ---- instance-scale measures (Preprocessor.scala:230-264, 312-339) -----------------------------------------
int dx_measures_compute(dx_ctx* ctx, dx_measure_sizes* out) {
  return guard(ctx, [&] { run_measures(ctx, out); });
}

int dx_measures_classes(dx_ctx* ctx, int64_t* class_term, int64_t* node_freq, int64_t* node_count) {
  return guard(ctx, [&] {
    const MeasureState& ms = ctx->measures;
    if (!ms.valid) throw DxError(DX_E_INVAL, "dx_measures_compute has not run");
    const size_t n = ms.class_term.size() * 8;
    if (n) {
      memcpy(class_term, ms.class_term.data(), n);
      memcpy(node_freq, ms.node_freq.data(), n);
      memcpy(node_count, ms.node_count.data(), n);
    }
  });
}

int dx_measures_cc(dx_ctx* ctx, int64_t* sub_type, int64_t* pred, int64_t* obj_type, int64_t* rows, int64_t* dist) {
  return guard(ctx, [&] {
    const MeasureState& ms = ctx->measures;
    if (!ms.valid) throw DxError(DX_E_INVAL, "dx_measures_compute has not run");
    const size_t n = ms.cc_rows.size() * 8;
    if (n) {
      memcpy(sub_type, ms.cc_sub.data(), n);
      memcpy(pred, ms.cc_pred.data(), n);
      memcpy(obj_type, ms.cc_obj.data(), n);
      memcpy(rows, ms.cc_rows.data(), n);
      memcpy(dist, ms.cc_dist.data(), n);
    }
  });
}

static std::vector<std::string> decode_terms(dx_ctx* ctx, const std::vector<int64_t>& ids) {
  std::vector<std::string> out(ids.size());
  if (ids.empty()) return out;
  uint64_t total = 0;
  int rc = dx_terms_bytes(ctx, ids.data(), ids.size(), &total);
  if (rc) throw DxError(rc, ctx->err);
  std::string bytes(total, '\0');
  std::vector<uint64_t> off(ids.size() + 1);
  rc = dx_terms_decode(ctx, ids.data(), ids.size(), total ? &bytes[0] : nullptr, total, off.data());
  if (rc) throw DxError(rc, ctx->err);
  for (size_t i = 0; i < ids.size(); ++i) out[i] = bytes.substr(off[i], off[i + 1] - off[i]);
  return out;
}

int dx_measures_write(dx_ctx* ctx, const char* base, const char* dataset, const char* local_dir) {
  return guard(ctx, [&] {
    const MeasureState& ms = ctx->measures;
    if (!ms.valid || !dataset) throw DxError(DX_E_INVAL, "dx_measures_compute has not run / null dataset");
    const std::string B = norm_base(base), D = dataset;
    std::string L = local_dir ? local_dir : "";
    if (!L.empty() && L.back() != '/') L += '/';
    const std::vector<std::string> cls = decode_terms(ctx, ms.class_term);
    std::map<int64_t, const std::string*> by_term;
    for (size_t i = 0; i < cls.size(); ++i) by_term[ms.class_term[i]] = &cls[i];
    // schema_cc: what RDD[((String, String, String), Long, Long)].saveAsTextFile writes (Preprocessor.scala:261-262).
    // Each of the three outputs is written only where it does not exist yet (Preprocessor.scala:37-46).
    const std::string cc = B + D + "_data/schema_cc";
    struct stat stt;
    if (stat(cc.c_str(), &stt) != 0) {
      mkdirs(cc);
      const std::vector<std::string> preds = decode_terms(ctx, ms.cc_pred);
      std::ofstream f(cc + "/part-00000", std::ios::binary);
      for (size_t g = 0; g < ms.cc_rows.size(); ++g)
        f << "((" << *by_term[ms.cc_sub[g]] << "," << preds[g] << "," << *by_term[ms.cc_obj[g]] << ")," << ms.cc_rows[g] << "," << ms.cc_dist[g]
          << ")\n";
      if (!f) throw DxError(DX_E_IO, "writing " + cc + " failed");
      touch(cc + "/_SUCCESS");
    }
    // schema_node_freq.txt / schema_node_count.txt: "class\tvalue", descending value (Preprocessor.scala:321-325, 334-338);
    // equal values are ordered by class term id here (the reference leaves them in no particular order)
    mkdirs(L + D + "_local");
    auto write_sorted = [&](const std::string& path, const std::vector<int64_t>& v) {
      if (stat(path.c_str(), &stt) == 0) return;
      std::vector<size_t> idx;
      for (size_t i = 0; i < v.size(); ++i)
        if (v[i] > 0) idx.push_back(i);  // a class that never reaches the reduceByKey has no line
      std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return v[a] > v[b]; });
      std::ofstream f(path, std::ios::binary);
      for (size_t i : idx) f << cls[i] << "\t" << v[i] << "\n";
      if (!f) throw DxError(DX_E_IO, "writing " + path + " failed");
    };
    write_sorted(L + D + "_local/schema_node_freq.txt", ms.node_freq);
    write_sorted(L + D + "_local/schema_node_count.txt", ms.node_count);
  });
}

// ---- on-disk layout -----------------------------------------------------------------------------------
// The layout with the files encoded by host threads: the earlier writer, taken only with DX_LAYOUT_WRITER=host,
// which the GPU encoder does not make.
static void write_layout_host(dx_ctx* ctx, const char* base, const char* dataset, const char* local_dir) {
  {
    const PartitionState& ps = ctx->part;
    const Dictionary& d = ctx->dict;
    if (!ps.valid || !dataset) throw DxError(DX_E_INVAL, "no partition run / null dataset");
    const std::string B = norm_base(base), D = dataset, N = std::to_string(ps.n_partitions);
    std::string L = local_dir ? local_dir : "";
    if (!L.empty() && L.back() != '/') L += '/';
    const std::string clusters = B + D + "_data/cluster/clusters_" + N + "_bal";
    struct stat stt;
    if (stat(clusters.c_str(), &stt) == 0)  // DAP.scala:86: the stage is skipped when its output exists
      throw DxError(DX_E_INVAL, clusters + " already exists (the reference skips the stage in that case; remove it to rewrite)");
    mkdirs(clusters);
    // What the Scala tools read is written as Parquet where they look for it. Beside it, under <dataset>_data/dx_<N>/
    // (a directory none of them lists), the dictionary and the tables' id columns, which dx_open_dataset and
    // dx_load_table read back without decoding strings.
    const std::string side = B + D + "_data/dx_" + N;
    const std::string dd = side + "/dictionary";
    mkdirs(dd);
    std::vector<uint64_t> off(d.n_terms + 1);
    std::vector<uint8_t> pool;
    {
      fetch(off.data(), d.pool_off.p, off.size() * 8);
      pool.resize(off[d.n_terms]);
      fetch(pool.data(), d.pool.p, pool.size());
      std::ofstream f1(dd + "/offsets.bin", std::ios::binary), f2(dd + "/terms.bin", std::ios::binary);
      f1.write((const char*)off.data(), (std::streamsize)(off.size() * 8));
      f2.write((const char*)pool.data(), (std::streamsize)pool.size());
      if (!f1 || !f2) throw DxError(DX_E_IO, "writing dictionary failed");
    }
    const uint64_t nt = d.n_terms;
    // Spark names its files part-<task>-<job uuid>-c000.<codec>.parquet; the partition number is read from the five
    // digits after "part-" (DAP.scala:427-428). One id per call stands in for the job's uuid.
    char job[64];
    snprintf(job, sizeof job, "%08x-dx00-4000-8000-%012llx", (unsigned)time(nullptr), (unsigned long long)(uintptr_t)ctx & 0xFFFFFFFFFFFFull);
    const std::string suffix = parquet_writer_uses_snappy() ? "-c000.snappy.parquet" : "-c000.parquet";
    auto part_file = [&](int k) { return "part-" + part_name(k) + "-" + job + suffix; };
    // The files are independent of each other: this thread fetches each file's id columns from the device and makes
    // the directories, a pool of host threads encodes and writes (one thread took 6.1 s for LUBM-100's 5.6 GB).
    std::vector<std::pair<uint64_t, std::function<void()>>> jobs;  // (values, work)
    std::vector<std::string> success;  // the markers are written when the files are
    // the ids of the files that wait for a thread are held in host memory: at 2^28 values (2 GiB) the pool runs what there is
    auto flush_if_large = [&] {
      uint64_t held = 0;
      for (auto& j : jobs) held += j.first;
      if (held >= (1ull << 28)) { run_jobs(jobs); jobs.clear(); }
    };
    using Ids = std::shared_ptr<std::vector<int64_t>>;
    auto fetch_ids = [&](const int64_t* dev, uint64_t n) {
      Ids v = std::make_shared<std::vector<int64_t>>(n);
      fetch(v->data(), dev, n * 8);
      return v;
    };
    const uint8_t* pool_p = pool.data();
    const uint64_t* off_p = off.data();
    // first-level partitions + statistics.txt_N (DAP.scala:364-381)
    mkdirs(L + D + "_local");
    std::ofstream stats(L + D + "_local/statistics.txt_" + N);
    for (int q = 0; q < ps.n_partitions; ++q) {
      const uint64_t n = ps.part_base[q + 1] - ps.part_base[q];
      if (!n) continue;  // Spark writes no part file for an empty partition
      Ids hs = fetch_ids(ps.row_s.p + ps.part_base[q], n), ho = fetch_ids(ps.row_o.p + ps.part_base[q], n);
      Ids hp = std::make_shared<std::vector<int64_t>>(n);
      {
        std::vector<uint32_t> pi(n);
        fetch(pi.data(), ps.row_p.p + ps.part_base[q], n * 4);
        for (uint64_t i = 0; i < n; ++i) (*hp)[i] = ps.pred_terms[pi[i]];
      }
      const std::string path = clusters + "/" + part_file(q);
      jobs.emplace_back(n * 3, [=] {
        ParquetWriter w(path, {"_1", "_2", "_3"});  // (s, o, p): DAP.scala:576,592,628
        w.append(n, {{hs->data(), nullptr}, {ho->data(), nullptr}, {hp->data(), nullptr}}, pool_p, off_p, nt);
        w.close();
      });
      flush_if_large();
      const std::string qs = std::to_string(q);
      stats << (qs.size() == 1 ? "0" + qs : qs) << " " << n << "\n";
    }
    success.push_back(clusters + "/_SUCCESS");
    // VP directories (DAP.scala:419-437) and partition_stat (DAP.scala:101-114)
    std::map<std::string, uint64_t> dir_rows;
    std::map<std::string, int> file_no;
    for (const VpTable& t : ps.tables) {
      if (!t.in_layout) continue;
      const std::string rel = part_name(t.partition) + "/_3=" + ps.pred_storage[t.pidx];
      const std::string pdir = clusters + "/" + part_name(t.partition);
      const std::string dir = clusters + "/" + rel, sdir = side + "/cluster/" + rel;
      mkdirs(dir);
      mkdirs(sdir);
      Ids hs = fetch_ids(ps.row_s.p + t.begin, t.rows), ho = fetch_ids(ps.row_o.p + t.begin, t.rows);
      const int k = file_no[dir]++;
      const std::string path = dir + "/" + part_file(k), spath = sdir + "/part-" + part_name(k) + ".dxcol";
      const uint64_t rows = t.rows;
      jobs.emplace_back(rows * 2, [=] {
        ParquetWriter w(path, {"_1", "_2"});
        w.append(rows, {{hs->data(), nullptr}, {ho->data(), nullptr}}, pool_p, off_p, nt);
        w.close();
        write_cols(spath, rows, {hs->data(), ho->data()});
      });
      flush_if_large();
      dir_rows[part_name(t.partition)] += t.rows;
      success.push_back(pdir + "/_SUCCESS");
    }
    {
      std::ofstream pstat(L + D + "_local/partition_stat.txt_" + N + "_bal");
      for (auto& kv : dir_rows) pstat << kv.first << " " << kv.second << "\n";  // `hdfs dfs -ls` lists by name
      for (int q = 0; q < ps.n_partitions; ++q) {
        const uint64_t n = ps.part_base[q + 1] - ps.part_base[q];
        if (n) pstat << part_file(q) << " " << n << "\n";
      }
    }
    // statTables.txt: rows per VP table, the input QueryOptimizer.scala:270-276 reads (the reference has no writer for it)
    {
      std::ofstream st(L + D + "_local/statTables.txt");
      std::map<std::string, uint64_t> rows;
      for (const VpTable& t : ps.tables)
        if (t.in_layout) rows[part_name(t.partition) + "-_3=" + ps.pred_storage[t.pidx]] += t.rows;
      for (auto& kv : rows) st << kv.first << " " << kv.second << "\n";
    }
    // node index (DAP.scala:402-412): Types.NodeIndex(uri, uriType)
    {
      const std::string ni = B + D + "_data/node_index_" + N;
      mkdirs(ni);
      Ids hs = fetch_ids(ps.node_s.p, ps.n_node), ho = fetch_ids(ps.node_o.p, ps.n_node);
      const std::string path = ni + "/" + part_file(0);
      const uint64_t n = ps.n_node;
      jobs.emplace_back(n * 2, [=] {
        ParquetWriter w(path, {"uri", "uriType"});
        w.append(n, {{hs->data(), nullptr}, {ho->data(), nullptr}}, pool_p, off_p, nt);
        w.close();
      });
      success.push_back(ni + "/_SUCCESS");
    }
    // VP index (DAP.scala:450-501): one row per uri with the set of "s" / "o" + sanitised predicate it occurs with,
    // joined by commas (collect_set leaves the order open: sorted here)
    for (int q = 0; q < ps.n_partitions; ++q) {
      if (ps.part_base[q + 1] == ps.part_base[q]) continue;
      const uint64_t n = (uint64_t)dx_vp_index_rows(ctx, q);
      Ids u_p = std::make_shared<std::vector<int64_t>>(n), pr_p = std::make_shared<std::vector<int64_t>>(n);
      auto r32_p = std::make_shared<std::vector<int32_t>>(n);
      if (n) {
        int rc = dx_vp_index_fetch(ctx, q, u_p->data(), r32_p->data(), pr_p->data());
        if (rc) throw DxError(rc, ctx->err);
      }
      auto storage_p = std::make_shared<std::unordered_map<int64_t, std::string>>();  // predicate term -> sanitised name
      for (uint32_t i = 0; i < ps.pred_terms.size(); ++i) (*storage_p)[ps.pred_terms[i]] = ps.pred_storage[i];
      const std::string dir = B + D + "_data/classSubPart_index_" + N + "__" + part_name(q);
      mkdirs(dir);
      const std::string path = dir + "/" + part_file(0);
      jobs.emplace_back(n * 4, [=] {
      const std::vector<int64_t>&u = *u_p, &pr = *pr_p;
      const std::vector<int32_t>& r32 = *r32_p;
      const std::unordered_map<int64_t, std::string>& storage = *storage_p;
      std::vector<uint64_t> order(n);
      for (uint64_t i = 0; i < n; ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return u[a] < u[b]; });
      std::vector<int64_t> uris;
      std::vector<std::string> parts;
      for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        std::vector<std::string> set;
        for (; j < n && u[order[j]] == u[order[i]]; ++j) {
          auto it = storage.find(pr[order[j]]);
          if (it == storage.end()) throw DxError(DX_E_INVAL, "VP index names a predicate the partition run does not know");
          set.push_back((r32[order[j]] == 0 ? "s" : "o") + it->second);
        }
        std::sort(set.begin(), set.end());
        set.erase(std::unique(set.begin(), set.end()), set.end());
        std::string joined;
        for (size_t k = 0; k < set.size(); ++k) joined += (k ? "," : "") + set[k];
        uris.push_back(u[order[i]]);
        parts.push_back(std::move(joined));
        i = j;
      }
      ParquetWriter w(path, {"uri", "parts"});
      w.append(uris.size(), {{uris.data(), nullptr}, {nullptr, parts.data()}}, pool_p, off_p, nt);
      w.close();
      });
      success.push_back(dir + "/_SUCCESS");
    }
    run_jobs(jobs);
    for (const std::string& p : success) touch(p);
  }
}

int dx_part_write_layout(dx_ctx* ctx, const char* base, const char* dataset, const char* local_dir) {
  return guard(ctx, [&] {
    const char* w = getenv("DX_LAYOUT_WRITER");  // "host": the earlier writer, pages built by host threads (kept for comparison)
    if (w && std::string(w) == "host") write_layout_host(ctx, base, dataset, local_dir);
    else write_layout(ctx, base, dataset, local_dir);
  });
}

int dx_allocator_stats(dx_ctx* ctx, uint64_t* requests, uint64_t* live_bytes) {
  return guard(ctx, [&] {
    if (requests) *requests = ctx->pool.requests();
    if (live_bytes) *live_bytes = ctx->pool.live_bytes();
  });
}

int dx_layout_times(dx_ctx* ctx, double* out, int n) {
  return guard(ctx, [&] {
    if (!out || n < 0) throw DxError(DX_E_INVAL, "null output");
    const LayoutTimes& t = ctx->layout_times;
    const double v[8] = {t.total_ms, t.prepare_ms, t.encode_ms, t.d2h_ms, t.write_wall_ms, t.write_busy_ms, (double)t.bytes, (double)t.threads};
    for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  });
}

}  // extern "C"

// ---- a layout's Parquet files -> dictionary + resident tables --------------------------------------------------
namespace {
// Distinct strings in the order they are first seen; id = position.
struct HostTerms {
  std::string bytes;
  std::vector<uint64_t> off{0};
  std::vector<uint64_t> hashes;   // per id
  std::vector<uint32_t> slots;    // id + 1, open addressing, a power of two
  HostTerms() : slots(1u << 16, 0) {}
  uint32_t intern(const uint8_t* p, uint32_t len) {
    const uint64_t h = hash_bytes(p, len, 0x9E3779B97F4A7C15ull);
    size_t mask = slots.size() - 1, at = (size_t)h & mask;
    for (;; at = (at + 1) & mask) {
      const uint32_t v = slots[at];
      if (!v) break;
      const uint32_t id = v - 1;
      if (hashes[id] == h && off[id + 1] - off[id] == len && memcmp(bytes.data() + off[id], p, len) == 0) return id;
    }
    if (hashes.size() >= 0xFFFFFFF0ull) throw DxError(DX_E_RANGE, "more than 2^32 terms");
    const uint32_t id = (uint32_t)hashes.size();
    bytes.append(reinterpret_cast<const char*>(p), len);
    off.push_back(bytes.size());
    hashes.push_back(h);
    slots[at] = id + 1;
    if (hashes.size() * 2 > slots.size()) {  // keep it at most half full
      std::vector<uint32_t> bigger(slots.size() * 2, 0);
      mask = bigger.size() - 1;
      for (uint32_t i = 0; i < (uint32_t)hashes.size(); ++i) {
        size_t a = (size_t)hashes[i] & mask;
        while (bigger[a]) a = (a + 1) & mask;
        bigger[a] = i + 1;
      }
      slots.swap(bigger);
    }
    return id;
  }
};
// One table's two columns as ids: a dictionary page is looked up string by string, its rows only through the page's map.
struct ImportSink : ParquetStringSink {
  HostTerms& terms;
  std::vector<uint32_t> map[2];
  std::vector<int64_t> ids[2];
  explicit ImportSink(HostTerms& t) : terms(t) {}
  void dictionary(int col, const std::vector<std::pair<const uint8_t*, uint32_t>>& values) override {
    map[col].resize(values.size());
    for (size_t i = 0; i < values.size(); ++i) map[col][i] = terms.intern(values[i].first, values[i].second);
  }
  void indices(int col, const uint32_t* idx, size_t n) override {
    const size_t at = ids[col].size();
    ids[col].resize(at + n);
    for (size_t i = 0; i < n; ++i) ids[col][at + i] = map[col][idx[i]];
  }
  void value(int col, const uint8_t* bytes, uint32_t len) override { ids[col].push_back(terms.intern(bytes, len)); }
  void end_of_chunk(int) override {}
};
std::vector<std::string> list_dir(const std::string& dir) {
  std::vector<std::string> out;
  DIR* dp = opendir(dir.c_str());
  if (!dp) throw DxError(DX_E_NOENT, "cannot open " + dir);
  while (dirent* e = readdir(dp)) out.emplace_back(e->d_name);
  closedir(dp);
  std::sort(out.begin(), out.end());
  return out;
}
bool is_dir(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
bool ends_with(const std::string& s, const char* suffix) {
  const size_t n = strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

void install_dictionary(dx_ctx* ctx, const char* bytes, const uint64_t* offsets, uint64_t n_terms) {
  ctx->catalog.clear(); ctx->union_tables.clear();
  ctx->table_stats.clear();
  ctx->imported.clear();
  ctx->part = PartitionState();
  Dictionary nd;
  nd.n_terms = n_terms;
  nd.seed = 0x243F6A8885A308D3ull;
  nd.pool_off = DevBuf<uint64_t>(ctx->pool, n_terms + 1);
  nd.pool = DevBuf<uint8_t>(ctx->pool, offsets[n_terms] + 16);
  DX_HIP(hipMemcpy(nd.pool_off.p, offsets, (n_terms + 1) * 8, hipMemcpyHostToDevice));
  if (offsets[n_terms]) DX_HIP(hipMemcpy(nd.pool.p, bytes, offsets[n_terms], hipMemcpyHostToDevice));
  ctx->dict = std::move(nd);
  dict_build_table(ctx);  // fails if two terms are equal
}

// Every vertical-partition table of <base><dataset>_data/cluster/clusters_<N>_bal/<PPPPP>/_3=<predicate>/part-*.parquet
// (columns _1, _2: DAP.scala:433-437), the way QueryProcessor.loadDataset reads them (QueryProcessor.scala:308-339),
// all at once: one dictionary over their strings, the tables resident as id columns under their "PPPPP-_3=<dir>" keys.
void import_layout(dx_ctx* ctx, const std::string& B, const std::string& D, int n_partitions, uint64_t* n_tables, uint64_t* n_rows) {
  const std::string root = B + D + "_data/cluster/clusters_" + std::to_string(n_partitions) + "_bal";
  if (!is_dir(root)) throw DxError(DX_E_NOENT, root + " not found");
  HostTerms terms;
  struct Pending { std::string key; std::vector<int64_t> s, o; bool several_files; };
  std::vector<Pending> pending;
  uint64_t rows_total = 0;
  for (const std::string& part : list_dir(root)) {
    if (part.size() != 5 || part.find_first_not_of("0123456789") != std::string::npos || !is_dir(root + "/" + part)) continue;
    for (const std::string& sub : list_dir(root + "/" + part)) {
      if (sub.compare(0, 3, "_3=") != 0 || !is_dir(root + "/" + part + "/" + sub)) continue;
      const std::string dir = root + "/" + part + "/" + sub;
      ImportSink sink(terms);
      int files = 0;
      for (const std::string& f : list_dir(dir)) {
        if (!ends_with(f, ".parquet") || f[0] == '.' || f[0] == '_') continue;
        read_parquet_strings(dir + "/" + f, {"_1", "_2"}, sink);
        ++files;
      }
      if (sink.ids[0].size() != sink.ids[1].size()) throw DxError(DX_E_MALFORMED, "columns of different lengths in " + dir);
      if (sink.ids[0].size() >= (1ull << 32)) throw DxError(DX_E_RANGE, "table with 2^32 rows or more: " + dir);
      rows_total += sink.ids[0].size();
      pending.push_back({part + "-" + sub, std::move(sink.ids[0]), std::move(sink.ids[1]), files > 1});
    }
  }
  install_dictionary(ctx, terms.bytes.data(), terms.off.data(), terms.hashes.size());
  for (Pending& t : pending) {
    dx_ctx::ImportedTable it;
    it.rows = t.s.size();
    it.several_files = t.several_files;
    it.s = DevBuf<int64_t>(ctx->pool, it.rows);
    it.o = DevBuf<int64_t>(ctx->pool, it.rows);
    if (it.rows) {
      DX_HIP(hipMemcpy(it.s.p, t.s.data(), it.rows * 8, hipMemcpyHostToDevice));
      DX_HIP(hipMemcpy(it.o.p, t.o.data(), it.rows * 8, hipMemcpyHostToDevice));
    }
    ctx->imported.emplace(t.key, std::move(it));
    std::vector<int64_t>().swap(t.s);
    std::vector<int64_t>().swap(t.o);
  }
  ctx->ds_base = B;
  ctx->ds_name = D;
  ctx->ds_partitions = n_partitions;
  if (n_tables) *n_tables = pending.size();
  if (n_rows) *n_rows = rows_total;
}
}  // namespace

extern "C" {

int dx_import_layout(dx_ctx* ctx, const char* base, const char* dataset, int n_partitions, uint64_t* n_tables, uint64_t* n_rows) {
  return guard(ctx, [&] {
    if (!dataset) throw DxError(DX_E_INVAL, "null dataset");
    import_layout(ctx, norm_base(base), dataset, n_partitions, n_tables, n_rows);
  });
}

int dx_open_dataset(dx_ctx* ctx, const char* base, const char* dataset, int n_partitions) {
  return guard(ctx, [&] {
    if (!dataset) throw DxError(DX_E_INVAL, "null dataset");
    const std::string B = norm_base(base), D = dataset;
    const std::string dd = B + D + "_data/dx_" + std::to_string(n_partitions) + "/dictionary";
    struct stat sdd;
    if (stat(dd.c_str(), &sdd) != 0) {  // a layout written by the Scala tools: its Parquet files are all there is
      import_layout(ctx, B, D, n_partitions, nullptr, nullptr);
      return;
    }
    const std::string offs = read_file(dd + "/offsets.bin"), terms = read_file(dd + "/terms.bin");
    if (offs.size() < 8 || offs.size() % 8) throw DxError(DX_E_IO, "corrupt dictionary offsets");
    {  // the kernels index the term pool by these: they must start at 0, never decrease and end at the pool's size
      const uint64_t* o = reinterpret_cast<const uint64_t*>(offs.data());
      const size_t n = offs.size() / 8 - 1;
      if (n >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 terms");
      bool ok = o[0] == 0 && o[n] == terms.size();
      for (size_t i = 0; ok && i < n; ++i) ok = o[i] <= o[i + 1];
      if (!ok) throw DxError(DX_E_IO, "corrupt dictionary: offsets.bin does not describe terms.bin");
    }
    ctx->catalog.clear(); ctx->union_tables.clear();
    ctx->table_stats.clear();
    ctx->imported.clear();
    ctx->part = PartitionState();
    Dictionary nd;
    nd.n_terms = offs.size() / 8 - 1;
    nd.seed = 0x243F6A8885A308D3ull;
    nd.pool_off = DevBuf<uint64_t>(ctx->pool, nd.n_terms + 1);
    nd.pool = DevBuf<uint8_t>(ctx->pool, terms.size() + 16);
    DX_HIP(hipMemcpy(nd.pool_off.p, offs.data(), offs.size(), hipMemcpyHostToDevice));
    if (!terms.empty()) DX_HIP(hipMemcpy(nd.pool.p, terms.data(), terms.size(), hipMemcpyHostToDevice));
    ctx->dict = std::move(nd);
    dict_build_table(ctx);
    ctx->ds_base = B;
    ctx->ds_name = D;
    ctx->ds_partitions = n_partitions;
  });
}

static void load_from_disk(dx_ctx* ctx, const std::string& key, CatalogTable& t) {
  // QueryProcessor.loadDataset (QueryProcessor.scala:308-325): <hdfs><dataset>_data/cluster/clusters_<N>_bal/<partition>/<sub>/*
  if (ctx->ds_name.empty()) throw DxError(DX_E_NOENT, "table " + key + " is not resident and no dataset is open");
  std::string rel = key;
  const size_t dash = key.find('-');
  if (dash != std::string::npos) rel = key.substr(0, dash) + "/" + key.substr(dash + 1);
  const std::string dir = ctx->ds_base + ctx->ds_name + "_data/dx_" + std::to_string(ctx->ds_partitions) + "/cluster/" + rel;  // the id columns beside it
  DIR* dp = opendir(dir.c_str());
  if (!dp) throw DxError(DX_E_NOENT, "cannot open " + dir);
  std::vector<std::string> files;
  while (dirent* e = readdir(dp)) {
    const std::string n = e->d_name;
    if (n.size() > 6 && n.substr(n.size() - 6) == ".dxcol") files.push_back(dir + "/" + n);
  }
  closedir(dp);
  std::sort(files.begin(), files.end());
  std::vector<std::string> blobs;
  uint64_t rows = 0;
  for (auto& f : files) {
    blobs.push_back(read_file(f));
    const std::string& b = blobs.back();
    if (b.size() < sizeof(ColHeader)) throw DxError(DX_E_IO, "corrupt " + f);
    ColHeader h;
    memcpy(&h, b.data(), sizeof h);
    if (memcmp(h.magic, kMagic, 8) || h.ncols < 2 || h.ncols > 3 || h.rows >= (1ull << 32) || b.size() != sizeof h + h.rows * h.ncols * 8)
      throw DxError(DX_E_IO, "corrupt " + f);
    // ids index the dictionary: none may lie outside it
    const int64_t* v = reinterpret_cast<const int64_t*>(b.data() + sizeof h);
    const uint64_t nt = ctx->dict.n_terms;
    for (uint64_t i = 0; i < 2 * h.rows; ++i)
      if ((uint64_t)v[i] >= nt) throw DxError(DX_E_IO, "corrupt " + f + ": term id outside the dictionary");
    rows += h.rows;
  }
  if (rows >= (1ull << 32)) throw DxError(DX_E_RANGE, "table with 2^32 rows or more");
  t.own_s = DevBuf<int64_t>(ctx->pool, rows);
  t.own_o = DevBuf<int64_t>(ctx->pool, rows);
  uint64_t at = 0;
  for (auto& b : blobs) {
    ColHeader h;
    memcpy(&h, b.data(), sizeof h);
    if (h.rows) {
      DX_HIP(hipMemcpy(t.own_s.p + at, b.data() + sizeof h, h.rows * 8, hipMemcpyHostToDevice));
      DX_HIP(hipMemcpy(t.own_o.p + at, b.data() + sizeof h + h.rows * 8, h.rows * 8, hipMemcpyHostToDevice));
    }
    at += h.rows;
  }
  t.rows = rows;
  t.s = t.own_s.p;
  t.o = t.own_o.p;
  t.several_sources = files.size() > 1;
}

// Registers the table (resident slices, or column files from disk) without counting its distinct values yet.
static CatalogTable& register_table(dx_ctx* ctx, const char* partition_key) {
  {
    if (!partition_key) throw DxError(DX_E_INVAL, "null key");
    std::string key = partition_key;
    const std::string view = view_name(key);
    auto it = ctx->catalog.find(view);
    if (it == ctx->catalog.end()) {
      CatalogTable t;
      t.key = key;
      t.view = view;
      bool resident = false;
      PartitionState& ps = ctx->part;
      if (ps.valid) {
        if (!ps.by_key_built) {
          ps.by_key.reserve(ps.tables.size());
          std::vector<std::string> pn;  // partition numbers as the keys spell them
          for (int q = 0; q < ps.n_partitions; ++q) pn.push_back(part_name(q) + "-_3=");
          for (uint32_t i = 0; i < ps.tables.size(); ++i)
            if (ps.tables[i].in_layout) ps.by_key[pn[ps.tables[i].partition] + ps.pred_storage[ps.tables[i].pidx]].push_back(i);
          ps.by_key_built = true;
        }
        auto found = ps.by_key.find(key);
        if (found != ps.by_key.end())
          for (uint32_t i : found->second) {
            const VpTable& vt = ps.tables[i];
            t.slices.push_back({ps.row_s.p + vt.begin, ps.row_o.p + vt.begin, vt.rows});
            resident = true;
          }
      }
      if (!resident) {
        auto im = ctx->imported.find(key);
        if (im != ctx->imported.end()) {  // brought in from Parquet: the columns are in HBM already
          t.slices.push_back({im->second.s.p, im->second.o.p, im->second.rows});
          t.several_sources = im->second.several_files;  // several part files can hold a row twice, like several predicates in one directory
          resident = true;
        }
      }
      if (resident) {
        if (t.slices.size() == 1) {
          t.s = t.slices[0].s; t.o = t.slices[0].o; t.rows = t.slices[0].rows;
        } else {  // predicates whose names sanitise alike share one directory: the view is their concatenation
          for (auto& sl : t.slices) t.rows += sl.rows;
          t.own_s = DevBuf<int64_t>(ctx->pool, t.rows);
          t.own_o = DevBuf<int64_t>(ctx->pool, t.rows);
          uint64_t at = 0;
          for (auto& sl : t.slices) {
            // on the context's stream: ordered with the kernels that read the copy (the stream does not wait for the null stream)
            DX_HIP(hipMemcpyAsync(t.own_s.p + at, sl.s, sl.rows * 8, hipMemcpyDeviceToDevice, ctx->stream));
            DX_HIP(hipMemcpyAsync(t.own_o.p + at, sl.o, sl.rows * 8, hipMemcpyDeviceToDevice, ctx->stream));
            at += sl.rows;
          }
          t.s = t.own_s.p; t.o = t.own_o.p;
          t.several_sources = true;
        }
      } else {
        load_from_disk(ctx, key, t);
      }
      it = ctx->catalog.emplace(view, std::move(t)).first;
      it->second.s = it->second.own_s.p ? it->second.own_s.p : it->second.s;
      it->second.o = it->second.own_o.p ? it->second.own_o.p : it->second.o;
      auto st = ctx->table_stats.find(key);  // statistics survive unloadTables: the data did not change
      if (st == ctx->table_stats.end()) {
        it->second.stats_pending = true;  // counted by the caller, or when a plan first asks
      } else {
        it->second.distinct_s = st->second.first;
        it->second.distinct_o = st->second.second;
        it->second.sorted_s = st->second.sorted_s && it->second.slices.size() == 1;
      }
    }
    return it->second;
  }
}

int dx_load_table(dx_ctx* ctx, const char* partition_key, uint64_t* rows) {
  return guard(ctx, [&] {
    CatalogTable& t = register_table(ctx, partition_key);
    table_statistics(ctx, {&t});
    if (rows) *rows = t.rows;
  });
}

int dx_load_tables(dx_ctx* ctx, const char* const* partition_keys, int n_keys, uint64_t* rows_total) {
  return guard(ctx, [&] {
    if (n_keys < 0 || (n_keys && !partition_keys)) throw DxError(DX_E_INVAL, "null key list");
    std::vector<CatalogTable*> tables;
    uint64_t total = 0;
    for (int i = 0; i < n_keys; ++i) {
      CatalogTable& t = register_table(ctx, partition_keys[i]);  // std::map: references stay valid while others are added
      tables.push_back(&t);
      total += t.rows;
    }
    table_statistics(ctx, tables);  // the statistics of all of them behind one wait
    if (rows_total) *rows_total = total;
  });
}

int dx_dictionary_load(dx_ctx* ctx, const char* bytes, const uint64_t* offsets, uint64_t n_terms) {
  return guard(ctx, [&] {
    if (!offsets || (!bytes && n_terms && offsets[n_terms])) throw DxError(DX_E_INVAL, "null argument");
    if (n_terms >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 terms");
    for (uint64_t i = 0; i < n_terms; ++i)
      if (offsets[i + 1] < offsets[i]) throw DxError(DX_E_INVAL, "offsets must not decrease");
    if (offsets[0] != 0) throw DxError(DX_E_INVAL, "offsets[0] must be 0");
    install_dictionary(ctx, bytes, offsets, n_terms);
    ctx->ds_base.clear();
    ctx->ds_name.clear();
  });
}

int dx_load_table_ids(dx_ctx* ctx, const char* partition_key, const int64_t* s, const int64_t* o, uint64_t rows) {
  return guard(ctx, [&] {
    if (!partition_key || (rows && (!s || !o))) throw DxError(DX_E_INVAL, "null argument");
    if (!ctx->dict.valid()) throw DxError(DX_E_INVAL, "no dictionary loaded");
    if (rows >= (1ull << 32)) throw DxError(DX_E_RANGE, "table with 2^32 rows or more");
    for (uint64_t i = 0; i < rows; ++i)
      if (s[i] < 0 || o[i] < 0 || (uint64_t)s[i] >= ctx->dict.n_terms || (uint64_t)o[i] >= ctx->dict.n_terms)
        throw DxError(DX_E_RANGE, "row " + std::to_string(i) + " names a term id outside the dictionary");
    const std::string key = partition_key, view = view_name(key);
    ctx->catalog.erase(view);
    ctx->table_stats.erase(key);
    CatalogTable t;
    t.key = key;
    t.view = view;
    t.rows = rows;
    t.own_s = DevBuf<int64_t>(ctx->pool, rows);
    t.own_o = DevBuf<int64_t>(ctx->pool, rows);
    if (rows) {
      DX_HIP(hipMemcpy(t.own_s.p, s, rows * 8, hipMemcpyHostToDevice));
      DX_HIP(hipMemcpy(t.own_o.p, o, rows * 8, hipMemcpyHostToDevice));
    }
    auto it = ctx->catalog.emplace(view, std::move(t)).first;
    it->second.s = it->second.own_s.p;
    it->second.o = it->second.own_o.p;
    it->second.stats_pending = true;
  });
}

int dx_set_comm(dx_ctx* ctx, const dx_comm* comm) {
  return guard(ctx, [&] {
    rccl_comm_destroy(ctx);  // leaves a group made by dx_comm_rccl_init
    if (!comm) { ctx->comm = dx_comm{0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}; return; }
    if (comm->world <= 1) {  // a group of one rank: single-GPU execution; the collectives stay callable (dx_comm_selftest)
      ctx->comm = *comm;
      ctx->comm.rank = 0;
      ctx->comm.world = 1;
      return;
    }
    // allgather serves the queries too (a small join side, a cross product, goes to every rank through it)
    if (comm->rank < 0 || comm->rank >= comm->world || !comm->exchange_host || !comm->exchange || !comm->allreduce_sum || !comm->allgather)
      throw DxError(DX_E_INVAL, "dx_set_comm: rank out of range or a collective is missing (exchange_host, exchange, allreduce_sum and allgather are needed)");
    if (comm->world > 64) throw DxError(DX_E_RANGE, "dx_set_comm: more than 64 ranks");
    ctx->comm = *comm;
  });
}

int dx_rccl_unique_id(unsigned char id[DX_RCCL_ID_BYTES]) {
  if (!id) return DX_E_INVAL;
  try {
    rccl_unique_id(id);
    return DX_OK;
  } catch (const DxError& e) {
    return e.code;
  } catch (const std::exception&) {
    return DX_E_IO;
  }
}

int dx_comm_rccl_init(dx_ctx* ctx, const unsigned char id[DX_RCCL_ID_BYTES], int rank, int world) {
  return guard(ctx, [&] {
    if (!id) throw DxError(DX_E_INVAL, "null unique id");
    rccl_comm_create(ctx, id, rank, world, 0);
  });
}

int dx_comm_rccl_init_job(dx_ctx* ctx, const char* path, uint64_t job, int rank, int world, int timeout_ms) {
  return guard(ctx, [&] {
    if (!path) throw DxError(DX_E_INVAL, "null path");
    unsigned char id[DX_RCCL_ID_BYTES];
    const int wait_ms = timeout_ms > 0 ? timeout_ms : 60000;
    const std::string file = rccl_rendezvous_name(path, job);
    if (world > 1) rccl_rendezvous_file(file, rank, wait_ms, id);
    else rccl_unique_id(id);  // a group of one rank has nobody to meet
    try {
      rccl_comm_create(ctx, id, rank, world, wait_ms);
    } catch (...) {
      if (world > 1 && rank == 0) rccl_rendezvous_done(file);
      throw;
    }
    if (world > 1 && rank == 0) rccl_rendezvous_done(file);
  });
}

int dx_comm_rccl_init_file(dx_ctx* ctx, const char* path, int rank, int world, int timeout_ms) {
  return dx_comm_rccl_init_job(ctx, path, 0, rank, world, timeout_ms);
}

int dx_comm_selftest(dx_ctx* ctx) {
  return guard(ctx, [&] {
    const dx_comm& cm = ctx->comm;
    if (!cm.exchange_host || !cm.exchange || !cm.allreduce_sum || !cm.allgather || !cm.allreduce_sum_dev)
      throw DxError(DX_E_INVAL, "no communicator with all five collectives is set");
    const int W = cm.world, R = cm.rank;
    hipStream_t s = ctx->stream;
    auto fail = [&](const std::string& what) { throw DxError(DX_E_IO, "communicator self-test: " + what); };
    auto sync_before = [&] { if (!cm.stream_ordered) DX_HIP(hipStreamSynchronize(s)); };
    {  // host all-to-all, three words per peer
      std::vector<uint64_t> send(3 * W), recv(3 * W, ~0ull);
      for (int p = 0; p < W; ++p) { send[3 * p] = 1000ull * R + p; send[3 * p + 1] = 7; send[3 * p + 2] = p; }
      if (cm.exchange_host(cm.user, send.data(), recv.data(), 3)) fail("exchange_host returned an error");
      for (int p = 0; p < W; ++p)
        if (recv[3 * p] != 1000ull * p + R || recv[3 * p + 1] != 7 || recv[3 * p + 2] != (uint64_t)R) fail("exchange_host: wrong words from rank " + std::to_string(p));
    }
    {  // host sum
      uint64_t v[2] = {(uint64_t)R + 1, 10};
      if (cm.allreduce_sum(cm.user, v, 2)) fail("allreduce_sum returned an error");
      if (v[0] != (uint64_t)W * (W + 1) / 2 || v[1] != 10ull * W) fail("allreduce_sum: wrong sums");
    }
    for (int elem_bytes : {8, 4}) {  // device all-to-all: rank r sends p + 1 + r elements to peer p
      std::vector<uint64_t> sc(W), rc(W);
      uint64_t ns = 0, nr = 0;
      for (int p = 0; p < W; ++p) { sc[p] = p + 1 + R; rc[p] = R + 1 + p; ns += sc[p]; nr += rc[p]; }
      std::vector<uint64_t> h64(ns);
      std::vector<uint32_t> h32(ns);
      uint64_t at = 0;
      for (int p = 0; p < W; ++p)
        for (uint64_t i = 0; i < sc[p]; ++i, ++at) { h64[at] = 1000000ull * R + 1000ull * p + i; h32[at] = (uint32_t)h64[at]; }
      DevBuf<uint8_t> send(ctx->pool, ns * elem_bytes), recv(ctx->pool, nr * elem_bytes);
      DX_HIP(hipMemcpyAsync(send.p, elem_bytes == 8 ? (const void*)h64.data() : (const void*)h32.data(), ns * elem_bytes, hipMemcpyHostToDevice, s));
      DX_HIP(hipMemsetAsync(recv.p, 0xEE, nr * elem_bytes, s));
      sync_before();
      if (cm.exchange(cm.user, send.p, sc.data(), recv.p, rc.data(), elem_bytes)) fail("exchange returned an error");
      std::vector<uint64_t> g64(nr);
      std::vector<uint32_t> g32(nr);
      DX_HIP(hipMemcpyAsync(elem_bytes == 8 ? (void*)g64.data() : (void*)g32.data(), recv.p, nr * elem_bytes, hipMemcpyDeviceToHost, s));
      DX_HIP(hipStreamSynchronize(s));
      at = 0;
      for (int p = 0; p < W; ++p)
        for (uint64_t i = 0; i < rc[p]; ++i, ++at) {
          const uint64_t want = 1000000ull * p + 1000ull * R + i;
          if ((elem_bytes == 8 ? g64[at] : (uint64_t)g32[at]) != (elem_bytes == 8 ? want : (uint64_t)(uint32_t)want))
            fail("exchange (" + std::to_string(elem_bytes) + "-byte elements): wrong data from rank " + std::to_string(p));
        }
    }
    {  // all-gather with uneven shares: rank r contributes r + 2 words
      std::vector<uint64_t> counts(W);
      uint64_t total = 0;
      for (int p = 0; p < W; ++p) { counts[p] = p + 2; total += counts[p]; }
      std::vector<uint64_t> mine(R + 2);
      for (int i = 0; i < R + 2; ++i) mine[i] = 100ull * R + i;
      DevBuf<uint64_t> send(ctx->pool, mine.size()), recv(ctx->pool, total);
      DX_HIP(hipMemcpyAsync(send.p, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, s));
      DX_HIP(hipMemsetAsync(recv.p, 0xEE, total * 8, s));
      sync_before();
      if (cm.allgather(cm.user, send.p, mine.size(), recv.p, counts.data(), 8)) fail("allgather returned an error");
      std::vector<uint64_t> got(total);
      DX_HIP(hipMemcpyAsync(got.data(), recv.p, total * 8, hipMemcpyDeviceToHost, s));
      DX_HIP(hipStreamSynchronize(s));
      uint64_t at = 0;
      for (int p = 0; p < W; ++p)
        for (uint64_t i = 0; i < counts[p]; ++i, ++at)
          if (got[at] != 100ull * p + i) fail("allgather: wrong data from rank " + std::to_string(p));
    }
    for (int elem_bytes : {1, 8}) {  // device sums in place
      const uint64_t n = 300;
      std::vector<uint8_t> h8(n);
      std::vector<uint64_t> h64(n);
      for (uint64_t i = 0; i < n; ++i) { h8[i] = (uint8_t)((i + R) % 3); h64[i] = (i + R) % 3 + (1ull << 40); }
      DevBuf<uint8_t> buf(ctx->pool, n * elem_bytes);
      DX_HIP(hipMemcpyAsync(buf.p, elem_bytes == 1 ? (const void*)h8.data() : (const void*)h64.data(), n * elem_bytes, hipMemcpyHostToDevice, s));
      sync_before();
      if (cm.allreduce_sum_dev(cm.user, buf.p, n, elem_bytes)) fail("allreduce_sum_dev returned an error");
      DX_HIP(hipMemcpyAsync(elem_bytes == 1 ? (void*)h8.data() : (void*)h64.data(), buf.p, n * elem_bytes, hipMemcpyDeviceToHost, s));
      DX_HIP(hipStreamSynchronize(s));
      for (uint64_t i = 0; i < n; ++i) {
        uint64_t want = 0;
        for (int p = 0; p < W; ++p) want += (i + p) % 3 + (elem_bytes == 8 ? (1ull << 40) : 0);
        if ((elem_bytes == 1 ? (uint64_t)h8[i] : h64[i]) != (elem_bytes == 1 ? (want & 0xFF) : want)) fail("allreduce_sum_dev: wrong sum");
      }
    }
  });
}

int dx_load_tables_sharded(dx_ctx* ctx, const char* const* partition_keys, int n_keys, uint64_t* rows_total) {
  return guard(ctx, [&] {
    if (n_keys < 0 || (n_keys && !partition_keys)) throw DxError(DX_E_INVAL, "null key list");
    const dx_comm& cm = ctx->comm;
    std::vector<std::string> keys;  // each table once, in the order given
    for (int i = 0; i < n_keys; ++i) {
      if (!partition_keys[i]) throw DxError(DX_E_INVAL, "null key");
      if (std::find(keys.begin(), keys.end(), partition_keys[i]) == keys.end()) keys.emplace_back(partition_keys[i]);
    }
    // Local part: load what this rank owns, declare the rest. A failure (a missing file) is kept for the agreement below.
    LocalError err;
    std::vector<CatalogTable*> mine;
    std::vector<uint64_t> stats(3 * keys.size(), 0);
    run_local(err, [&] {
      for (const std::string& key : keys) {
        char* end = nullptr;
        const long q = strtol(key.c_str(), &end, 10);  // "00001-_3=..." -> first-level partition 1
        if (end == key.c_str() || *end != '-' || q < 0) throw DxError(DX_E_INVAL, "table key without a partition number: " + key);
        if (cm.world <= 1 || q % cm.world == cm.rank) {
          mine.push_back(&register_table(ctx, key.c_str()));
        } else if (!ctx->catalog.count(view_name(key))) {
          CatalogTable t;
          t.key = key;
          t.view = view_name(key);
          t.remote = true;
          ctx->catalog.emplace(t.view, std::move(t));
        }
      }
      table_statistics(ctx, mine);
      size_t at = 0;
      for (size_t i = 0; i < keys.size(); ++i) {
        auto it = ctx->catalog.find(view_name(keys[i]));
        if (it->second.remote) continue;
        stats[3 * i] = it->second.rows; stats[3 * i + 1] = it->second.distinct_s; stats[3 * i + 2] = it->second.distinct_o;
        ++at;
      }
      (void)at;
    });
    uint64_t total = 0;
    if (cm.world > 1) {
      agree_or_throw(ctx, err);
      // one owner per table: the sum over the ranks is the owner's numbers
      for (size_t at = 0; at < stats.size(); at += 1023) {
        const int n = (int)std::min<size_t>(1023, stats.size() - at);
        if (cm.allreduce_sum(cm.user, stats.data() + at, n) != 0) throw DxError(DX_E_IO, "collective failed: allreduce_sum of the table statistics");
      }
      for (size_t i = 0; i < keys.size(); ++i) {
        CatalogTable& t = ctx->catalog.find(view_name(keys[i]))->second;
        t.has_plan_stats = true;
        t.plan_rows = stats[3 * i]; t.plan_distinct_s = stats[3 * i + 1]; t.plan_distinct_o = stats[3 * i + 2];
      }
    } else if (err) {
      throw DxError(err.code, err.msg);
    }
    for (size_t i = 0; i < keys.size(); ++i) total += stats[3 * i];
    if (rows_total) *rows_total = total;
  });
}

int dx_declare_remote_table(dx_ctx* ctx, const char* partition_key) {
  return guard(ctx, [&] {
    if (!partition_key) throw DxError(DX_E_INVAL, "null key");
    const std::string key = partition_key, view = view_name(key);
    if (ctx->catalog.count(view)) return;
    CatalogTable t;
    t.key = key;
    t.view = view;
    t.remote = true;
    ctx->catalog.emplace(view, std::move(t));
  });
}

int dx_table_stats(dx_ctx* ctx, const char* partition_key, uint64_t out[3]) {
  return guard(ctx, [&] {
    if (!partition_key || !out) throw DxError(DX_E_INVAL, "null argument");
    auto it = ctx->catalog.find(view_name(partition_key));
    if (it == ctx->catalog.end()) throw DxError(DX_E_NOENT, std::string("table not loaded: ") + partition_key);
    table_statistics(ctx, {&it->second});
    out[0] = it->second.rows; out[1] = it->second.distinct_s; out[2] = it->second.distinct_o;
  });
}

int dx_set_table_stats(dx_ctx* ctx, const char* partition_key, const uint64_t in[3]) {
  return guard(ctx, [&] {
    if (!partition_key || !in) throw DxError(DX_E_INVAL, "null argument");
    auto it = ctx->catalog.find(view_name(partition_key));
    if (it == ctx->catalog.end()) throw DxError(DX_E_NOENT, std::string("table neither loaded nor declared: ") + partition_key);
    it->second.has_plan_stats = true;
    it->second.plan_rows = in[0]; it->second.plan_distinct_s = in[1]; it->second.plan_distinct_o = in[2];
  });
}

int dx_unload_all(dx_ctx* ctx) {
  return guard(ctx, [&] { ctx->catalog.clear(); ctx->union_tables.clear(); });
}

// ---- queries -----------------------------------------------------------------------------------------
int dx_query(dx_ctx* ctx, const char* sql, int mode, dx_result** out) {
  return guard(ctx, [&] {
    if (!sql || !out) throw DxError(DX_E_INVAL, "null argument");
    if (mode != DX_QUERY_COUNT && mode != DX_QUERY_BINDINGS) throw DxError(DX_E_INVAL, "unknown query mode");
    *out = ctx->comm.world > 1 ? run_query_sharded(ctx, sql, mode) : run_query(ctx, sql, mode);
  });
}

int dx_query_joined(dx_ctx* ctx, const char* const* statements, int n, int mode, dx_result** out) {
  return guard(ctx, [&] {
    if (!statements || !out || n < 1) throw DxError(DX_E_INVAL, "null argument or no statement");
    if (mode != DX_QUERY_COUNT && mode != DX_QUERY_BINDINGS) throw DxError(DX_E_INVAL, "unknown query mode");
    std::string text;
    for (int i = 0; i < n; ++i) {
      if (!statements[i]) throw DxError(DX_E_INVAL, "null statement");
      if (strchr(statements[i], kStatementSeparator)) throw DxError(DX_E_MALFORMED, "SQL: control character in a statement");
      if (i) text.push_back(kStatementSeparator);
      text += statements[i];
    }
    *out = ctx->comm.world > 1 ? run_query_sharded(ctx, text, mode) : run_query(ctx, text, mode);
  });
}

int dx_query_count_noalloc(dx_ctx* ctx, const char* sql, uint64_t* count) {
  return guard(ctx, [&] {
    if (!sql || !count) throw DxError(DX_E_INVAL, "null argument");
    if (ctx->comm.world > 1) throw DxError(DX_E_UNSUPPORTED, "dx_query_count_noalloc runs on one GPU: the exchange between ranks allocates");
    *count = run_query_count_noalloc(ctx, sql);
  });
}

uint64_t dx_result_count(dx_result* r) { return r ? r->global_count : 0; }
uint64_t dx_result_local_rows(dx_result* r) { return r ? r->count : 0; }
int dx_result_columns(dx_result* r) { return r ? (int)r->names.size() : 0; }
const char* dx_result_column_name(dx_result* r, int i) { return (r && i >= 0 && i < (int)r->names.size()) ? r->names[i].c_str() : ""; }
double dx_result_device_ms(dx_result* r) { return r ? dx::result_device_ms(r) : 0; }

int dx_result_fetch_ids(dx_result* r, int64_t* ids, int sorted) {
  if (!r) return DX_E_INVAL;
  return guard(r->ctx, [&] {
    if (!r->has_bindings) throw DxError(DX_E_INVAL, "result was computed with DX_QUERY_COUNT: no bindings to fetch");
    const uint64_t n = r->count;
    const size_t nc = r->cols.size();
    for (size_t c = 0; c < nc; ++c) fetch(ids + c * n, r->cols[c].p, n * 8);
    if (!sorted || n < 2) return;
    // sort rows by the decoded strings of the columns taken in column-name order
    dx_ctx* ctx = r->ctx;
    const Dictionary& d = ctx->dict;
    std::vector<uint64_t> off(d.n_terms + 1);
    fetch(off.data(), d.pool_off.p, off.size() * 8);
    std::vector<uint8_t> pool(off[d.n_terms]);
    fetch(pool.data(), d.pool.p, pool.size());
    std::vector<size_t> corder(nc);
    for (size_t c = 0; c < nc; ++c) corder[c] = c;
    std::sort(corder.begin(), corder.end(), [&](size_t a, size_t b) { return r->names[a] < r->names[b]; });
    auto term = [&](int64_t id, const uint8_t*& p, size_t& len) {
      if (id >= 0) { p = pool.data() + off[id]; len = off[id + 1] - off[id]; }
      else { const std::string& s = r->virtual_terms[(size_t)(-id - 2)]; p = (const uint8_t*)s.data(); len = s.size(); }
    };
    std::vector<uint64_t> perm(n);
    for (uint64_t i = 0; i < n; ++i) perm[i] = i;
    std::sort(perm.begin(), perm.end(), [&](uint64_t a, uint64_t b) {
      for (size_t c : corder) {
        const int64_t x = ids[c * n + a], y = ids[c * n + b];
        if (x == y) continue;
        const uint8_t *px, *py; size_t lx, ly;
        term(x, px, lx); term(y, py, ly);
        const int cmp = memcmp(px, py, std::min(lx, ly));
        if (cmp) return cmp < 0;
        if (lx != ly) return lx < ly;
      }
      return false;
    });
    std::vector<int64_t> tmp(n);
    for (size_t c = 0; c < nc; ++c) {
      for (uint64_t i = 0; i < n; ++i) tmp[i] = ids[c * n + perm[i]];
      memcpy(ids + c * n, tmp.data(), n * 8);
    }
  });
}

const char* dx_result_virtual_term(dx_result* r, int64_t id) {
  if (!r || id > -2 || (size_t)(-id - 2) >= r->virtual_terms.size()) return nullptr;
  return r->virtual_terms[(size_t)(-id - 2)].c_str();
}

const char* dx_result_plan(dx_result* r) { return r ? r->plan_text().c_str() : ""; }

void dx_result_free(dx_result* r) { delete r; }

}  // extern "C"
