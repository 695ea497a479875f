// Context state: dictionary, partitioner results, table catalogue.
#pragma once
#include <cstring>
#include <memory>

#include "dx_internal.h"

namespace dx {

constexpr uint32_t kNil = 0xFFFFFFFFu;
constexpr int kRepLenBits = 24;  // a term is at most 16 MiB; its first byte sits below 1 TiB

// One slot of the open-addressing term table: 16 bytes, so a probe touches one memory sector. While the
// table is being built `val` holds the smallest (offset << 24 | length) word among the occurrences
// of the slot's hash; afterwards it holds the term id.
struct DictSlot {
  uint64_t key;
  uint64_t val;
};

// Term dictionary resident in HBM. id -> bytes through pool/pool_off; bytes -> id through an
// open-addressing table keyed by the 64-bit term hash (slot -> id), confirmed by byte compare.
struct Dictionary {
  uint64_t n_terms = 0;
  uint64_t seed = 0;
  DevBuf<uint8_t> pool;        // concatenated term bytes, id order
  DevBuf<uint64_t> pool_off;   // n_terms + 1
  DevBuf<DictSlot> ht;         // key = term hash (0 = empty), val = term id
  uint64_t ht_mask = 0;
  bool valid() const { return pool_off.p != nullptr; }
};

struct ClassIndexEntry {
  std::string uri;
  uint64_t mask;
};

struct VpTable {
  int32_t partition;
  uint32_t pidx;          // dense predicate index
  int64_t pred_term;
  uint64_t begin, rows;   // slice of the partition-major row arrays
  bool in_layout;
};

struct PartitionState {
  bool valid = false;
  int n_partitions = 0;
  dx_part_stats stats{};
  // rows of all first-level partitions, partition-major; inside a partition sorted by (predicate, s, o)
  DevBuf<int64_t> row_s, row_o;
  DevBuf<uint32_t> row_p;             // dense predicate index
  std::vector<uint64_t> part_base;    // n_partitions + 1
  std::vector<int64_t> pred_terms;    // pidx -> term id
  std::vector<std::string> pred_strings, pred_storage;  // pidx -> IRI text, sanitised directory name
  std::vector<VpTable> tables;        // non-empty ones, ordered by (partition, pidx)
  // "PPPPP-_3=<directory>" -> the tables behind that key (several when predicates share a directory), made when a table
  // is first asked for by key: a schema of ten thousand predicates has eighty thousand tables, and a statement names hundreds
  std::unordered_map<std::string, std::vector<uint32_t>> by_key;
  bool by_key_built = false;
  // node index
  DevBuf<int64_t> node_s, node_o;
  uint64_t n_node = 0;
  // VP index: sorted unique keys (partition, uri, role, pidx) packed into 64 bits
  // VP index as two lists of packed (partition | term | role | predicate) words: the subject entries in
  // row order (they are distinct as they come: rows are sorted by partition, predicate, subject) and the
  // object entries sorted and de-duplicated. Each has its own per-partition bases (n_partitions + 1).
  DevBuf<uint64_t> vpi_s, vpi_o;
  std::vector<uint64_t> vpi_s_base, vpi_o_base;
  int vpi_bits_term = 0, vpi_bits_pred = 0;
  // sorted distinct input triples packed as (pidx, s, o) plus what is needed to unpack them
  // (64-bit words when predicate index + two term ids fit them, 128-bit words otherwise: wide_keys)
  DevBuf<uint64_t> uniq;
  DevBuf<u128> uniq_w;
  bool wide_keys = false;
  uint64_t n_uniq_slots = 0;          // length of uniq (duplicates are still in, flagged by equality with the predecessor)
  // several GPUs: uniq holds what was routed to this rank; the rank's own triples (packed alike, ids of the common
  // dictionary, unsorted) stay here for what needs every triple once, wherever it was placed
  DevBuf<uint64_t> local_keys;
  DevBuf<u128> local_keys_w;
  uint64_t n_local_keys = 0;
  int bits_term = 0;
  uint32_t type_pidx = kNil;
  DevBuf<uint8_t> term_flags;         // per term: bit0 = literal (starts with '"'), bit1 = vertex, bit2 = has edge
};

// Instance-scale measures (schema_cc, schema_node_freq, schema_node_count) of the last partition run, on the host:
// they are schema-sized.
struct MeasureState {
  bool valid = false;
  std::vector<int64_t> class_term, node_freq, node_count;           // per class, ascending term id
  std::vector<int64_t> cc_sub, cc_pred, cc_obj, cc_rows, cc_dist;   // per (subType, pred, objType) group
  uint64_t n_type_entries = 0, n_cc_rows = 0;
};

constexpr int kFileStages = 8;  // pinned 32 MiB buffers between a file and the device

// Where the time of the last dx_part_write_layout went (wall clock of the calling thread unless said otherwise).
struct LayoutTimes {
  double total_ms = 0;
  double prepare_ms = 0;     // directories, text files, the list of files
  double encode_ms = 0;      // page bodies on the GPU (with the host's planning of the files between its two passes)
  double d2h_ms = 0;         // the files' bytes coming to pinned host memory; the writing threads run meanwhile
  double write_wall_ms = 0;  // first write started .. last write done
  double write_busy_ms = 0;  // summed over the writing threads
  uint64_t bytes = 0;        // written through the image (everything but the small text files)
  int threads = 0;
};

struct CatalogTable {
  std::string key;      // "00001-_3=<pred>"
  std::string view;     // "table00001__3_E_<pred>"
  uint64_t rows = 0;
  uint64_t distinct_s = 0, distinct_o = 0;  // feed the join-order estimates; counted when a plan first needs them
  bool stats_pending = false;               // loaded, not counted yet
  bool sorted_s = false;                    // checked when the statistics are counted: s never decreases from one row to the next
  bool unsorted_s = false;                  // found so: the distinct subjects are counted like the objects, through a byte per term
  // A directory that several predicates share (their names sanitise alike) is read as one table and can hold a
  // row twice. Spark's UNION removes such rows, a plain sub-select keeps them: UNION arms read this copy instead.
  bool several_sources = false;             // more than one slice / file behind the view
  std::unique_ptr<CatalogTable> distinct_view;
  // several GPUs: a table another rank holds has no rows here; the planner works on the numbers of the
  // whole table, which the host sets on every rank alike
  bool remote = false;
  bool has_plan_stats = false;
  uint64_t plan_rows = 0, plan_distinct_s = 0, plan_distinct_o = 0;
  // a view is the concatenation of one or more row slices (several predicates can sanitise to one name)
  struct Slice { const int64_t* s; const int64_t* o; uint64_t rows; };
  std::vector<Slice> slices;
  // owned storage when loaded from disk or when slices had to be concatenated
  DevBuf<int64_t> own_s, own_o;
  const int64_t* s = nullptr;
  const int64_t* o = nullptr;
};

// One arm of a UNION as far as its rows go: the table, the constants its columns are compared with (ids; -1: a constant the
// dictionary does not hold, which matches nothing), and which of its columns become the first and the second output
// column (0 = s, 1 = o, -1 = none).
struct UnionSource {
  const CatalogTable* t = nullptr;
  bool has_s = false, has_o = false;
  int64_t want_s = 0, want_o = 0;
  int src_a = -1, src_b = -1;
};

}  // namespace dx

namespace dx { struct RcclComm; struct SqlQuery; }

struct dx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  dx::DevPool pool;
  dx::Dictionary dict;
  dx::PartitionState part;
  dx::MeasureState measures;
  dx::LayoutTimes layout_times;
  std::map<std::string, dx::CatalogTable> catalog;   // by view name
  struct TableStats { uint64_t first = 0, second = 0; bool sorted_s = false; };
  std::map<std::string, TableStats> table_stats;  // by table key: distinct s, distinct o, whether s is in ascending order
  // Vertical-partition tables brought in from a layout's Parquet files (dx_import_layout), by table key: resident id
  // columns that dx_load_table hands out without copying, like the tables of a partition run.
  struct ImportedTable { dx::DevBuf<int64_t> s, o; uint64_t rows = 0; bool several_files = false; };
  std::map<std::string, ImportedTable> imported;
  // dataset opened from disk
  std::string ds_base, ds_name;
  int ds_partitions = 0;
  // query workspace
  dx::DevBuf<uint8_t> qws;
  double q_clock_khz = 0;               // ticks of the device's wall clock per millisecond
  void* q_host = nullptr;               // pinned block the final counters of a statement land in
  void* q_host_dev = nullptr;           // the same block as the device addresses it
  uint64_t q_seq = 0;                   // statements run on this context so far
  void* file_stage[dx::kFileStages] = {};          // pinned buffers dx_partition_file streams a file through
  hipEvent_t file_stage_ev[dx::kFileStages] = {};
  void* q_counters = nullptr;           // device: a statement's row counters and its last-block ticket, zero between statements
  // duplicate-free results of groups of UNION arms that may hold a row more than once (see distinct_union_of), by the
  // group's description; emptied whenever the catalogue is
  std::unordered_map<std::string, std::unique_ptr<dx::CatalogTable>> union_tables;
  std::unordered_map<size_t, uint64_t> q_regular_steps;  // per statement text: join steps that need their hash table in HBM
  struct SeenRows { uint64_t signature = 0; std::vector<uint64_t> rows; };
  std::unordered_map<size_t, SeenRows> q_seen_rows;       // per statement text: rows after each step the last time it ran (over the tables `signature` stands for)
  // Statements come again: the parsed form is kept per text, and the join order the search found per (text, numbers the
  // search looked at), so a repeated statement over tables with the same statistics skips both (DX_NO_PLAN_CACHE=1: never).
  std::unordered_map<std::string, std::shared_ptr<const dx::SqlQuery>> q_parsed;
  struct JoinOrder { std::vector<int> order; std::vector<double> rows; };
  std::unordered_map<uint64_t, JoinOrder> q_orders;
  bool q_counters_dirty = false;        // a statement failed between its first and its last kernel: clear before the next
  std::vector<hipEvent_t> part_events;  // the partitioner's stage timers, created once
  int part_two_pass_calls = 0;          // calls that still skip the fused tokeniser: an input had a tile with more records than it has slots
  uint8_t* part_host = nullptr;         // pinned block the partitioner's small results come back through
  uint8_t* part_host_dev = nullptr;     // the same block as the device addresses it
  size_t part_host_used = 0;
  unsigned long long* fetch_seq = nullptr;      // pinned word: number of the last batch of small results that has landed
  unsigned long long* fetch_seq_dev = nullptr;
  uint64_t fetch_seq_next = 0;
  // this GPU's copy of the file-scope scratch array behind dx_query_count_noalloc (address looked up in dx_init)
  void* noalloc_scratch = nullptr;
  size_t noalloc_scratch_bytes = 0;
  // several GPUs: collectives handed in by the host; world == 1 means single-GPU execution
  dx_comm comm{0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  dx::RcclComm* native_comm = nullptr;  // the communicator made by dx_comm_rccl_init (comm.user points at it)
};

namespace dx {
// Counts and other small results come back through one pinned block. A copy into pinned memory is queued on the
// stream like a kernel; a copy into pageable memory makes the host wait for each one on its own (20-40 us apiece,
// with the GPU idle). add() queues, sync() waits for the stream and hands the values out; collect() hands out
// after a synchronisation that has happened anyway. The block is handed out from its start again at every
// partition call; what does not fit is copied the slow way.
constexpr size_t kPartHostBytes = 256 << 10;
// Batched mode (the partitioner): small results are not copied one by one (a copy engine transfer or a copy kernel
// each, ~6 us apiece with its gap) but written into the pinned block by ONE kernel per batch, which then stores the
// batch's number into a pinned word; sync() watches that word instead of waiting for the stream's completion signal
// (whose wake-up costs ~10 us more). A batch is queued by flush() or sync(): the device buffers named by add() must
// live until then.
constexpr int kFetchBatch = 8;
constexpr size_t kFetchSmall = 8 << 10;
struct FetchBatch {
  const void* src[kFetchBatch];
  uint32_t off[kFetchBatch], bytes[kFetchBatch];
  int n;
};
void launch_fetch_batch(dx_ctx* ctx, const FetchBatch& b, unsigned long long seq);  // dx_hp1.hip
void wait_fetch_seq(dx_ctx* ctx, unsigned long long seq);                            // dx_hp1.hip
struct HostFetch {
  dx_ctx* ctx;
  struct Item { void* dst; size_t off, bytes; };
  std::vector<Item> items;
  bool batched;
  FetchBatch pending{};
  unsigned long long last_seq = 0;  // the batch queued last, 0: something else was queued after it (or nothing at all)
  explicit HostFetch(dx_ctx* c, bool batch = false) : ctx(c), batched(batch) { pending.n = 0; }
  void ensure_block() {
    if (ctx->part_host) return;
    void* p = nullptr;
    DX_HIP(hipHostMalloc(&p, kPartHostBytes, hipHostMallocDefault));
    ctx->part_host = static_cast<uint8_t*>(p);
    ctx->part_host_used = 0;
    void* d = nullptr;
    DX_HIP(hipHostGetDevicePointer(&d, p, 0));
    ctx->part_host_dev = static_cast<uint8_t*>(d);
    DX_HIP(hipHostMalloc(&p, 64, hipHostMallocDefault));
    ctx->fetch_seq = static_cast<unsigned long long*>(p);
    *ctx->fetch_seq = 0;
    DX_HIP(hipHostGetDevicePointer(&d, p, 0));
    ctx->fetch_seq_dev = static_cast<unsigned long long*>(d);
  }
  void add(void* dst, const void* dev, size_t bytes) {
    if (!bytes) return;
    ensure_block();
    const size_t off = (ctx->part_host_used + 15) & ~(size_t)15;
    if (off + bytes > kPartHostBytes) {
      flush();
      DX_HIP(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
      last_seq = 0;
      return;
    }
    ctx->part_host_used = off + bytes;
    items.push_back({dst, off, bytes});
    if (batched && bytes <= kFetchSmall) {
      if (pending.n == kFetchBatch) flush();
      pending.src[pending.n] = dev;
      pending.off[pending.n] = (uint32_t)off;
      pending.bytes[pending.n] = (uint32_t)bytes;
      ++pending.n;
      return;
    }
    flush();  // keeps the order of the stream: what was asked for first lands first
    DX_HIP(hipMemcpyAsync(ctx->part_host + off, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    last_seq = 0;
  }
  // queues the batch collected so far
  void flush() {
    if (!pending.n) return;
    last_seq = ++ctx->fetch_seq_next;
    launch_fetch_batch(ctx, pending, last_seq);
    pending.n = 0;
  }
  // Host to device through the same pinned block: the bytes are copied into it now (the caller's memory is free
  // again at once) and travel with the stream; a copy from pageable memory is staged by the runtime call by call.
  void upload(void* dev, const void* src, size_t bytes) {
    if (!bytes) return;
    ensure_block();
    const size_t off = (ctx->part_host_used + 15) & ~(size_t)15;
    if (off + bytes > kPartHostBytes) {
      DX_HIP(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, ctx->stream));
      DX_HIP(hipStreamSynchronize(ctx->stream));  // src may go away when the caller returns
      return;
    }
    std::memcpy(ctx->part_host + off, src, bytes);
    DX_HIP(hipMemcpyAsync(dev, ctx->part_host + off, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->part_host_used = off + bytes;
  }
  void collect() {
    for (const Item& it : items) std::memcpy(it.dst, ctx->part_host + it.off, it.bytes);
    items.clear();
  }
  void sync() {
    flush();
    if (last_seq) wait_fetch_seq(ctx, last_seq);  // the stream is in order: everything queued before the batch has landed too
    else DX_HIP(hipStreamSynchronize(ctx->stream));
    last_seq = 0;
    collect();
  }
};
}  // namespace dx

struct dx_result {
  dx_ctx* ctx = nullptr;
  uint64_t count = 0;         // rows held by this context
  uint64_t global_count = 0;  // rows over all ranks (== count on one GPU)
  double device_ms = 0;       // first kernel's begin to last kernel's end, by the device's wall clock
  bool has_bindings = false;
  std::vector<std::string> names;
  std::vector<dx::DevBuf<int64_t>> cols;   // one per selected column (bindings mode)
  std::vector<std::string> virtual_terms;  // projected constants: id -(2 + index)
  // Join order with the row count after every step. A statement leaves the facts (plan_steps) and the text is put
  // together when it is first asked for: formatting it costs microseconds that a short statement does not have.
  std::string plan;
  struct PlanArm { std::string view; uint64_t rows; bool fs, fo; };
  struct PlanStep { std::vector<PlanArm> arms; uint64_t rows_out; };
  std::vector<PlanStep> plan_steps;
  double plan_us = -1;
  double host_us[3] = {0, 0, 0};  // the last attempt's host time: buffers and arguments laid out, kernels queued, waited for the counters
  const std::string& plan_text() {
    if (plan_us >= 0) {
      std::string t = "parse+plan " + std::to_string((long)plan_us) + " us, lay out " + std::to_string((long)host_us[0]) + " us, queue " +
                      std::to_string((long)host_us[1]) + " us, wait " + std::to_string((long)host_us[2]) + " us\n";
      for (size_t k = 0; k < plan_steps.size(); ++k) {
        t += k ? "join " : "scan ";
        for (size_t a = 0; a < plan_steps[k].arms.size(); ++a) {
          const PlanArm& arm = plan_steps[k].arms[a];
          t += (a ? " UNION " : "") + arm.view + "[" + std::to_string(arm.rows) + " rows" + (arm.fs ? ", s filter" : "") + (arm.fo ? ", o filter" : "") + "]";
        }
        t += " -> " + std::to_string(plan_steps[k].rows_out) + " rows\n";
      }
      plan = t + plan;
      plan_us = -1;
      plan_steps.clear();
    }
    return plan;
  }
};

namespace dx {
// dx_parquet_read.cpp: the string columns of one Parquet file, handed over the way the file holds them: per column chunk
// a dictionary and runs of indices into it (or single values where a page is not dictionary-encoded). Columns are
// delivered chunk by chunk, column after column inside a row group; rows of different columns line up by position.
struct ParquetStringSink {
  virtual ~ParquetStringSink() = default;
  virtual void dictionary(int col, const std::vector<std::pair<const uint8_t*, uint32_t>>& values) = 0;  // valid during the call
  virtual void indices(int col, const uint32_t* idx, size_t n) = 0;       // rows that name entries of the column's last dictionary
  virtual void value(int col, const uint8_t* bytes, uint32_t len) = 0;    // one row of a page without dictionary
  virtual void end_of_chunk(int col) = 0;
};
void read_parquet_strings(const std::string& path, const std::vector<std::string>& columns, ParquetStringSink& sink);

// dx_parquet.cpp: files of REQUIRED string columns as the reference's Spark jobs write them (schemas of
// utils/Types.scala:7-16). A column is given as term ids over a string pool, or as strings.
bool parquet_writer_uses_snappy();  // pages are snappy streams like Spark's own (file names say so too) unless DX_PARQUET_CODEC=uncompressed
class ParquetWriter {
 public:
  struct Column { const int64_t* ids; const std::string* strings; };
  struct Impl;
  ParquetWriter(const std::string& path, const std::vector<std::string>& column_names);
  ~ParquetWriter();
  ParquetWriter(const ParquetWriter&) = delete;
  ParquetWriter& operator=(const ParquetWriter&) = delete;
  // n more rows; pool / pool_off / n_terms describe the strings behind the id columns
  void append(uint64_t n, const std::vector<Column>& cols, const uint8_t* pool, const uint64_t* pool_off, uint64_t n_terms);
  void close();  // footer; the destructor only releases the file

 private:
  Impl* p;
};
// The same format for page bodies that are encoded on the GPU (dx_layout.hip). Per column chunk the caller states the
// sizes of its pages: a dictionary page of n_dict entries in dict_bytes plus bit-packed indices, or (plain) one page of
// plain_bytes of length-prefixed values. The plan gives the file's size, every byte of it that is not a page body
// (as (file offset, bytes) pieces in file order) and the file offsets of the bodies.
constexpr uint64_t kParquetRowGroupRows = 1u << 20;
// snappy: the pages are snappy streams. The page of terms (a dictionary page, or a plain page that is a list of terms)
// arrives from the device as `terms_packed` bytes of literals and copies, 0 for a plain page that the device writes
// as it is (it then travels as one literal); the packed indices of a data page always travel as one literal.
struct PqChunkIn {
  uint64_t num_values = 0, n_dict = 0, dict_bytes = 0;
  bool plain = false;
  uint64_t plain_bytes = 0;
  bool snappy = false;
  uint64_t terms_packed = 0;
};
struct PqChunkOut { uint64_t dict_at = 0, data_at = 0, data_bytes = 0; uint32_t width = 0; };
struct PqFilePlan { uint64_t size = 0; std::vector<std::pair<uint64_t, std::string>> host; std::vector<PqChunkOut> chunks; };
// chunks: row group after row group, inside a group column after column
PqFilePlan plan_parquet_file(const std::vector<std::string>& names, const std::vector<uint64_t>& group_rows, const std::vector<PqChunkIn>& chunks);

// dx_layout.hip: the reference's on-disk layout from what a partition run left in HBM (dx_part_write_layout)
void write_layout(dx_ctx* ctx, const char* base, const char* dataset, const char* local_dir);
// dx_api.cpp: small helpers of the layout code
void mkdirs(const std::string& path);
void touch(const std::string& path);
std::string part_name(int i);
std::string norm_base(const char* base);

// dx_hp1.hip
void run_partition(dx_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, const std::vector<ClassIndexEntry>& classes,
                   int n_partitions);
void dict_lookup(dx_ctx* ctx, const char* bytes, const uint64_t* offsets, uint64_t n, int64_t* ids_host);
void dict_build_table(dx_ctx* ctx);  // (re)builds ht_* from pool/pool_off
// distinct subjects / objects of every table in the list that still lacks them: all kernels queued, one wait
void table_statistics(dx_ctx* ctx, const std::vector<CatalogTable*>& tables);
// the table without repeated rows (built on first use), or the table itself when it has one source
const CatalogTable* distinct_view_of(dx_ctx* ctx, CatalogTable& t);
// The rows the arms produce together, every row once, as a table of its own (first output column as s, second as o;
// a missing one is 0 throughout): what SQL's UNION makes of arms whose rows can coincide. Kept under `signature`.
const CatalogTable* distinct_union_of(dx_ctx* ctx, const std::vector<UnionSource>& arms, const std::string& signature);
// dx_measures.hip
void run_measures(dx_ctx* ctx, dx_measure_sizes* out);
// dx_hp2.hip
dx_result* run_query(dx_ctx* ctx, const std::string& sql, int mode);
double result_device_ms(dx_result* r);
void query_context_setup(dx_ctx* ctx);                      // pinned block + events of the query path, once per context
void* noalloc_scratch_of_current_device(size_t* bytes);     // the device array behind dx_query_count_noalloc
uint64_t run_query_count_noalloc(dx_ctx* ctx, const std::string& sql);
dx_result* run_query_sharded(dx_ctx* ctx, const std::string& sql, int mode);
extern const char* const kStageNames[12];
// dx_comm_rccl.cpp
void rccl_unique_id(unsigned char* out);
void rccl_comm_create(dx_ctx* ctx, const unsigned char* id_bytes, int rank, int world, int timeout_ms);
void rccl_comm_destroy(dx_ctx* ctx);
std::string rccl_rendezvous_name(const char* path, uint64_t job);
void rccl_rendezvous_file(const std::string& file, int rank, int timeout_ms, unsigned char* id);
void rccl_rendezvous_done(const std::string& file);
const char* rccl_comm_last_error(dx_ctx* ctx);

// ---- agreement between ranks ---------------------------------------------------------------------------
// A rank that fails between two collectives must not simply return: its peers would wait for it in the next
// one for ever. Every rank therefore brings the outcome of its local work to the next collective, and all
// of them leave with the same error code when any of them failed.
constexpr int kErrSlots = 8;
inline int err_slot(int code) {
  switch (code) {
    case DX_E_NOENT: return 0;
    case DX_E_IO: return 1;
    case DX_E_NOMEM: return 2;
    case DX_E_INVAL: return 3;
    case DX_E_RANGE: return 4;
    case DX_E_MALFORMED: return 5;
    case DX_E_UNSUPPORTED: return 6;
    default: return 7;
  }
}
inline int err_code_of_slot(int slot) {
  static const int codes[kErrSlots] = {DX_E_NOENT, DX_E_IO, DX_E_NOMEM, DX_E_INVAL, DX_E_RANGE, DX_E_MALFORMED, DX_E_UNSUPPORTED, DX_E_HIP};
  return codes[slot];
}
// An error every rank of the group leaves with (thrown by the two functions below): it must not be brought to
// another agreement.
struct AgreedError : DxError {
  AgreedError(int c, const std::string& m) : DxError(c, m) {}
};
struct LocalError {
  int code = 0;
  std::string msg;
  explicit operator bool() const { return code != 0; }
};
// Runs f unless an earlier step of this rank has already failed; a failure is kept, not thrown.
template <typename F>
void run_local(LocalError& err, F&& f) {
  if (err) return;
  try {
    f();
  } catch (const DxError& e) {
    err.code = e.code; err.msg = e.what();
  } catch (const HipError& e) {
    err.code = DX_E_HIP; err.msg = e.what();
  } catch (const std::bad_alloc&) {
    err.code = DX_E_NOMEM; err.msg = "host allocation failed";
  }
}
// Collective: every rank passes its own outcome; if any rank failed, all of them throw the error of the lowest
// failing slot (a rank that did not fail itself reports which error its peers had).
void agree_or_throw(dx_ctx* ctx, const LocalError& err);
// The same decision from codes that have already been exchanged (one per rank; 0 = fine).
void throw_if_any_failed(dx_ctx* ctx, const LocalError& mine, const std::vector<int>& codes_by_rank);
}  // namespace dx
