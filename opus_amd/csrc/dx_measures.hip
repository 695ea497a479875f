// Instance-scale inputs of the schema clustering as HIP kernels for gfx950: what Preprocessor.computeCC
// (Preprocessor.scala:230-264), computeSchemaNodeFreq (:312-326) and computeSchemaNodeCount (:328-339)
// compute over GraphX triplets, restated over the sorted packed triples the partitioner leaves in HBM.
//
//   * The triples are sorted by (predicate, s, o), so the rdf:type triples are one block, sorted by
//     (subject, class): a run of equal keys is one entry of the subject's type list and its length is the
//     entry's multiplicity (the reference's Seq[String] keeps a type stated twice as two entries).
//   * type_off[v] = first entry of term v: the type lists as CSR over term ids.
//   * An edge is a distinct triple outside that block with a non-literal object (Preprocessor.scala:153-167).
//       node count[c] = sum over edges, over the entries (c, m) of both endpoints, of m            (:329-333)
//       node freq[c]  = sum over entries (v, c, m) whose v is an endpoint of some edge, of m          (:313-320)
//       cc            = rows (subType, pred, objType, inst2) of every edge with both endpoints typed, one per
//                       pair of entries, weight m_s * m_o; sorted; per (subType, pred, objType) the weight sum
//                       and the number of distinct inst2                                               (:232-261)
//   * Classes are few (14 for LUBM, 10^2..10^3 elsewhere) and millions of edges add to them: a counter in HBM
//     hit by every wave is served by one L2 channel one atomic at a time, so each block adds into a histogram
//     in LDS and touches HBM once per class at its end.
#include <algorithm>

#include "dx_device.h"
#include "dx_state.h"

namespace dx {

namespace {

constexpr uint8_t kFlagLiteral = 1, kFlagEdge = 4;
constexpr int MS_BINS = 8192;     // classes per histogram pass (32 KiB of LDS)
constexpr unsigned MS_GRID = 512;  // every block ends with one atomic per class it met: few blocks, or the class counters queue up

inline unsigned blocks(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per ? (n + per - 1) / per : 1); }
inline int bits_for(uint64_t n) {
  int b = 1;
  while ((1ull << b) < n) ++b;
  return b;
}

// lower bounds of two keys in the sorted key array (thread 0 and thread 1)
template <typename K>
__global__ void k_ms_bounds(const K* __restrict__ keys, uint32_t n, K k0, K k1, uint32_t* __restrict__ out) {
  if (threadIdx.x > 1) return;
  const K k = threadIdx.x ? k1 : k0;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (keys[mid] < k) lo = mid + 1; else hi = mid;
  }
  out[threadIdx.x] = lo;
}

template <typename K>
__global__ void __launch_bounds__(256) k_ms_heads(const K* __restrict__ keys, uint32_t n, uint8_t* __restrict__ head) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one entry per distinct (subject, class) of the type block; at[j] = its first position in the block
template <typename K>
__global__ void __launch_bounds__(256)
k_ms_type_entries(const K* __restrict__ keys, const uint8_t* __restrict__ head, const uint32_t* __restrict__ pos, uint32_t n,
                  int bits_term, uint32_t* __restrict__ ent_s, uint32_t* __restrict__ ent_c, uint32_t* __restrict__ ent_at,
                  uint8_t* __restrict__ class_used) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const K tm = ((K)1 << bits_term) - 1;
  const uint32_t j = pos[i];
  const uint32_t c = (uint32_t)(keys[i] & tm);
  ent_s[j] = (uint32_t)((keys[i] >> bits_term) & tm);
  ent_c[j] = c;
  ent_at[j] = i;
  class_used[c] = 1;  // the same byte from every lane that sees the class: no atomic needed
}

__global__ void __launch_bounds__(256)
k_ms_list_classes(const uint8_t* __restrict__ class_used, const uint32_t* __restrict__ class_idx, const uint64_t* __restrict__ pool_off,
                  uint32_t n_terms, uint32_t* __restrict__ class_term, uint8_t* __restrict__ class_ok) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms || !class_used[t]) return;
  class_term[class_idx[t]] = t;
  class_ok[class_idx[t]] = pool_off[t + 1] != pool_off[t];  // computeCC drops rows with an empty string (:241)
}

// type_off[v] = number of entries whose subject is below v, for v in [0, n_terms]
__global__ void __launch_bounds__(256)
k_ms_type_off(const uint32_t* __restrict__ ent_s, uint32_t n_ent, uint32_t n_terms, uint32_t* __restrict__ type_off) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v > n_terms) return;
  uint32_t lo = 0, hi = n_ent;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (ent_s[mid] < v) lo = mid + 1; else hi = mid;
  }
  type_off[v] = lo;
}

// the block's histogram over the classes [c0, c0 + MS_BINS), added to HBM once per class that occurred
struct ClassBins {
  uint32_t* h;
  uint32_t c0, width;
  __device__ void clear() {
    for (uint32_t b = threadIdx.x; b < width; b += blockDim.x) h[b] = 0;
    __syncthreads();
  }
  __device__ void add(uint32_t c, uint32_t m) {
    const uint32_t b = c - c0;
    if (b < width) atomicAdd(&h[b], m);
  }
  __device__ void flush(unsigned long long* __restrict__ out) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < width; b += blockDim.x)
      if (h[b]) atomicAdd(&out[c0 + b], (unsigned long long)h[b]);
  }
};

__global__ void __launch_bounds__(256)
k_ms_node_freq(const uint32_t* __restrict__ ent_s, const uint32_t* __restrict__ ent_c, const uint32_t* __restrict__ ent_at, uint32_t n_ent,
               const uint8_t* __restrict__ flags, const uint32_t* __restrict__ class_idx, uint32_t c0, uint32_t n_classes,
               unsigned long long* __restrict__ freq) {
  __shared__ uint32_t lds[MS_BINS];
  ClassBins bins{lds, c0, n_classes - c0 < (uint32_t)MS_BINS ? n_classes - c0 : (uint32_t)MS_BINS};
  bins.clear();
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_ent; j += gridDim.x * blockDim.x)
    if (flags[ent_s[j]] & kFlagEdge) bins.add(class_idx[ent_c[j]], ent_at[j + 1] - ent_at[j]);
  bins.flush(freq);
}

struct Triple { uint32_t pidx, s, o; };
template <typename K>
__device__ __forceinline__ Triple unpack(K k, int bits_term) {
  const K tm = ((K)1 << bits_term) - 1;
  return {(uint32_t)(k >> (2 * bits_term)), (uint32_t)((k >> bits_term) & tm), (uint32_t)(k & tm)};
}
// true for the first copy of a triple that is an edge of the instance graph
template <typename K>
__device__ __forceinline__ bool is_edge(const K* __restrict__ keys, uint32_t i, int bits_term, uint32_t type_pidx,
                                        const uint8_t* __restrict__ flags, Triple& t) {
  const K k = keys[i];
  if (i && keys[i - 1] == k) return false;
  t = unpack(k, bits_term);
  return t.pidx != type_pidx && !(flags[t.o] & kFlagLiteral);
}

// node count per class, and per edge the number of cc rows it expands to (pairs of entries with non-empty class names)
template <typename K>
__global__ void __launch_bounds__(256)
k_ms_edges(const K* __restrict__ keys, uint32_t n, int bits_term, uint32_t type_pidx, const uint8_t* __restrict__ flags,
           const uint32_t* __restrict__ type_off, const uint32_t* __restrict__ ent_c, const uint32_t* __restrict__ ent_at,
           const uint32_t* __restrict__ class_idx, const uint8_t* __restrict__ class_ok, uint32_t c0, uint32_t n_classes,
           unsigned long long* __restrict__ count, uint32_t* __restrict__ cc_rows /*null after the first pass*/) {
  __shared__ uint32_t lds[MS_BINS];
  ClassBins bins{lds, c0, n_classes - c0 < (uint32_t)MS_BINS ? n_classes - c0 : (uint32_t)MS_BINS};
  bins.clear();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    Triple t;
    uint32_t rows = 0;
    if (is_edge(keys, i, bits_term, type_pidx, flags, t)) {
      uint32_t ok_s = 0, ok_o = 0;
      for (uint32_t j = type_off[t.s]; j < type_off[t.s + 1]; ++j) {
        const uint32_t c = class_idx[ent_c[j]];
        bins.add(c, ent_at[j + 1] - ent_at[j]);
        ok_s += class_ok[c];
      }
      for (uint32_t j = type_off[t.o]; j < type_off[t.o + 1]; ++j) {
        const uint32_t c = class_idx[ent_c[j]];
        bins.add(c, ent_at[j + 1] - ent_at[j]);
        ok_o += class_ok[c];
      }
      rows = ok_s * ok_o;
    }
    if (cc_rows) cc_rows[i] = rows;
  }
  bins.flush(count);
}

// the cc rows: key = (subType, pred, objType, inst2) packed, value = product of the two multiplicities
template <typename K>
__global__ void __launch_bounds__(256)
k_ms_cc_rows(const K* __restrict__ keys, uint32_t n, int bits_term, uint32_t type_pidx, const uint8_t* __restrict__ flags,
             const uint32_t* __restrict__ type_off, const uint32_t* __restrict__ ent_c, const uint32_t* __restrict__ ent_at,
             const uint32_t* __restrict__ class_idx, const uint8_t* __restrict__ class_ok, const uint32_t* __restrict__ cc_rows,
             const uint64_t* __restrict__ row_pos, int bits_class, int bits_pred, uint64_t* __restrict__ out_key,
             uint32_t* __restrict__ out_w) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !cc_rows[i]) return;
  const Triple t = unpack(keys[i], bits_term);
  uint64_t at = row_pos[i];
  for (uint32_t a = type_off[t.s]; a < type_off[t.s + 1]; ++a) {
    const uint32_t ca = class_idx[ent_c[a]];
    if (!class_ok[ca]) continue;
    const uint32_t ma = ent_at[a + 1] - ent_at[a];
    for (uint32_t b = type_off[t.o]; b < type_off[t.o + 1]; ++b) {
      const uint32_t cb = class_idx[ent_c[b]];
      if (!class_ok[cb]) continue;
      out_key[at] = (((((uint64_t)ca << bits_pred) | t.pidx) << bits_class | cb) << bits_term) | t.o;
      out_w[at] = ma * (ent_at[b + 1] - ent_at[b]);
      ++at;
    }
  }
}

__global__ void __launch_bounds__(256)
k_ms_cc_flags(const uint64_t* __restrict__ key, uint32_t n, int bits_term, uint8_t* __restrict__ head_group, uint8_t* __restrict__ head_inst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head_group[i] = (i == 0 || (key[i] >> bits_term) != (key[i - 1] >> bits_term)) ? 1 : 0;
  head_inst[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_ms_cc_groups(const uint64_t* __restrict__ key, const uint8_t* __restrict__ head_group, const uint32_t* __restrict__ gpos, uint32_t n,
               int bits_term, uint64_t* __restrict__ g_key, uint32_t* __restrict__ g_start) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head_group[i]) return;
  g_key[gpos[i]] = key[i] >> bits_term;
  g_start[gpos[i]] = i;
}

__global__ void __launch_bounds__(256)
k_ms_cc_finish(const uint32_t* __restrict__ g_start, uint32_t n_groups, uint32_t n, const uint64_t* __restrict__ wpos,
               const uint32_t* __restrict__ dpos, uint64_t* __restrict__ rows, uint64_t* __restrict__ dist) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t i0 = g_start[g], i1 = g + 1 < n_groups ? g_start[g + 1] : n;
  rows[g] = wpos[i1] - wpos[i0];
  dist[g] = (uint64_t)(dpos[i1] - dpos[i0]);
}

}  // namespace

// K: the key word of the partition run (uint64_t, or u128 for a dictionary too large for it)
template <typename K>
static void run_measures_t(dx_ctx* ctx, dx_measure_sizes* out, const DevBuf<K>& uniq, const DevBuf<K>& local_keys) {
  const PartitionState& ps = ctx->part;
  const Dictionary& d = ctx->dict;
  if (!ps.valid || (!uniq.p && ps.n_uniq_slots)) throw DxError(DX_E_INVAL, "the measures need the state of a partition run in this context");
  hipStream_t s = ctx->stream;
  DevPool& pool = ctx->pool;
  MeasureState& ms = ctx->measures;
  ms = MeasureState();
  HostFetch fetch(ctx);
  const size_t fetch_mark = ctx->part_host_used;  // the pinned block is handed out from here again when this call is over
  // Several GPUs: the measures count every input triple once, wherever (and whether) it was placed, and a type
  // stated on two ranks counts twice. Every rank therefore gathers all ranks' own triples -- 8 bytes apiece, 11 GB
  // for the largest configuration, against 288 GB of HBM -- sorts them and computes the (schema-sized) result for
  // itself: the same numbers on every rank, no partial results to merge.
  const dx_comm& cm = ctx->comm;
  DevBuf<K> gathered;
  const K* keys = uniq.p;
  uint64_t n_keys = ps.n_uniq_slots;
  if (cm.world > 1) {
    if (!cm.allgather || !cm.exchange_host) throw DxError(DX_E_INVAL, "the communicator lacks the collectives of the measures");
    if (!local_keys.p && ps.n_local_keys) throw DxError(DX_E_INVAL, "the partition run was not made with this communicator");
    std::vector<uint64_t> mine(cm.world, ps.n_local_keys), counts(cm.world, 0);
    if (cm.exchange_host(cm.user, mine.data(), counts.data(), 1)) throw DxError(DX_E_IO, "collective failed: exchange_host");
    n_keys = 0;
    for (uint64_t c : counts) n_keys += c;
    gathered = DevBuf<K>(pool, n_keys);
    DX_HIP(hipStreamSynchronize(s));
    if (cm.allgather(cm.user, local_keys.p, ps.n_local_keys, gathered.p, counts.data(), (int)sizeof(K))) throw DxError(DX_E_IO, "collective failed: allgather of the triples");
    radix_sort_keys(pool, gathered, n_keys, bits_for(ps.pred_terms.size() ? ps.pred_terms.size() : 1) + 2 * ps.bits_term, s);
    keys = gathered.p;
  }
  if (n_keys >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 triples for the measures on one GPU");
  const uint32_t n = (uint32_t)n_keys, n_terms = (uint32_t)d.n_terms;
  const int bt = ps.bits_term;
  const uint32_t np = (uint32_t)ps.pred_terms.size();
  hipEvent_t e0, e1;
  DX_HIP(hipEventCreate(&e0));
  DX_HIP(hipEventCreate(&e1));
  struct Ev { hipEvent_t a, b; ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } ev{e0, e1};
  DX_HIP(hipEventRecord(e0, s));
  auto finish = [&] {
    DX_HIP(hipEventRecord(e1, s));
    DX_HIP(hipEventSynchronize(e1));
    float msec = 0;
    DX_HIP(hipEventElapsedTime(&msec, e0, e1));
    ctx->part_host_used = fetch_mark;
    ms.valid = true;
    if (out) {
      out->n_classes = ms.class_term.size();
      out->n_cc = ms.cc_rows.size();
      out->n_type_entries = ms.n_type_entries;
      out->n_cc_rows = ms.n_cc_rows;
      out->device_ms = msec;
    }
  };
  if (!n || ps.type_pidx == kNil) { finish(); return; }

  // ---- the rdf:type block and the type lists
  DevBuf<uint32_t> bounds(pool, 2);
  const K k0 = (K)ps.type_pidx << (2 * bt);
  const bool last_pred = ps.type_pidx + 1 >= np;  // the block then runs to the end of the keys
  const K k1 = last_pred ? ~(K)0 : (K)(ps.type_pidx + 1) << (2 * bt);
  hipLaunchKernelGGL((k_ms_bounds<K>), dim3(1), dim3(64), 0, s, keys, n, k0, k1, bounds.p);
  uint32_t hb[2] = {0, 0};
  fetch.add(hb, bounds.p, 8);
  fetch.sync();
  const uint32_t tb = hb[0], nt = (last_pred ? n : hb[1]) - hb[0];
  if (!nt) { finish(); return; }
  const K* tkeys = keys + tb;
  DevBuf<uint8_t> head(pool, nt);
  DevBuf<uint32_t> pos(pool, nt), d_tot(pool, 2);
  hipLaunchKernelGGL((k_ms_heads<K>), dim3(blocks(nt)), dim3(256), 0, s, tkeys, nt, head.p);
  exclusive_scan_u8_u32(pool, head.p, pos.p, nt, d_tot.p, s);
  uint32_t n_ent = 0;
  fetch.add(&n_ent, d_tot.p, 4);
  fetch.sync();
  ms.n_type_entries = n_ent;
  DevBuf<uint32_t> ent_s(pool, n_ent), ent_c(pool, n_ent), ent_at(pool, (size_t)n_ent + 1);
  DevBuf<uint8_t> class_used(pool, n_terms);
  DevBuf<uint32_t> class_idx(pool, n_terms);
  DX_HIP(hipMemsetAsync(class_used.p, 0, n_terms, s));
  hipLaunchKernelGGL((k_ms_type_entries<K>), dim3(blocks(nt)), dim3(256), 0, s, tkeys, head.p, pos.p, nt, bt, ent_s.p, ent_c.p, ent_at.p,
                     class_used.p);
  DX_HIP(hipMemcpyAsync(ent_at.p + n_ent, &nt, 4, hipMemcpyHostToDevice, s));
  exclusive_scan_u8_u32(pool, class_used.p, class_idx.p, n_terms, d_tot.p + 1, s);
  uint32_t n_classes = 0;
  fetch.add(&n_classes, d_tot.p + 1, 4);
  fetch.sync();  // also: nt has been read by the copy above
  head.reset(); pos.reset();
  DevBuf<uint32_t> class_term(pool, n_classes);
  DevBuf<uint8_t> class_ok(pool, n_classes);
  hipLaunchKernelGGL(k_ms_list_classes, dim3(blocks(n_terms)), dim3(256), 0, s, class_used.p, class_idx.p, d.pool_off.p, n_terms, class_term.p,
                     class_ok.p);
  DevBuf<uint32_t> type_off(pool, (size_t)n_terms + 1);
  hipLaunchKernelGGL(k_ms_type_off, dim3(blocks((uint64_t)n_terms + 1)), dim3(256), 0, s, ent_s.p, n_ent, n_terms, type_off.p);

  // ---- node frequency, node count, and how many cc rows every edge expands to
  DevBuf<unsigned long long> freq(pool, n_classes), count(pool, n_classes);
  DX_HIP(hipMemsetAsync(freq.p, 0, (size_t)n_classes * 8, s));
  DX_HIP(hipMemsetAsync(count.p, 0, (size_t)n_classes * 8, s));
  DevBuf<uint32_t> cc_rows(pool, n);
  for (uint32_t c0 = 0; c0 < n_classes; c0 += MS_BINS) {
    hipLaunchKernelGGL(k_ms_node_freq, dim3(std::min(blocks(n_ent), MS_GRID)), dim3(256), 0, s, ent_s.p, ent_c.p, ent_at.p, n_ent,
                       ps.term_flags.p, class_idx.p, c0, n_classes, freq.p);
    hipLaunchKernelGGL((k_ms_edges<K>), dim3(std::min(blocks(n), MS_GRID)), dim3(256), 0, s, keys, n, bt, ps.type_pidx, ps.term_flags.p,
                       type_off.p, ent_c.p, ent_at.p, class_idx.p, class_ok.p, c0, n_classes, count.p, c0 == 0 ? cc_rows.p : (uint32_t*)nullptr);
  }

  // ---- cc: expand, sort, reduce
  DevBuf<uint64_t> row_pos(pool, n), d_tot64(pool, 1);
  exclusive_scan_u32_u64(pool, cc_rows.p, row_pos.p, n, d_tot64.p, s);
  uint64_t R = 0;
  fetch.add(&R, d_tot64.p, 8);
  std::vector<uint32_t> h_class(n_classes);
  std::vector<unsigned long long> h_freq(n_classes), h_count(n_classes);
  fetch.add(h_class.data(), class_term.p, (size_t)n_classes * 4);
  fetch.add(h_freq.data(), freq.p, (size_t)n_classes * 8);
  fetch.add(h_count.data(), count.p, (size_t)n_classes * 8);
  fetch.sync();
  ms.class_term.assign(h_class.begin(), h_class.end());
  ms.node_freq.assign(h_freq.begin(), h_freq.end());
  ms.node_count.assign(h_count.begin(), h_count.end());
  ms.n_cc_rows = R;
  if (R >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 (subType, pred, objType, instance) rows on one GPU");
  const int bits_class = bits_for(n_classes), bits_pred = bits_for(np ? np : 1);
  if (2 * bits_class + bits_pred + bt > 64) throw DxError(DX_E_RANGE, "classes, predicates and terms do not fit a 64-bit packed key");
  if (R) {
    const uint32_t r = (uint32_t)R;
    DevBuf<uint64_t> key(pool, r);
    DevBuf<uint32_t> w(pool, r);
    hipLaunchKernelGGL((k_ms_cc_rows<K>), dim3(blocks(n)), dim3(256), 0, s, keys, n, bt, ps.type_pidx, ps.term_flags.p, type_off.p, ent_c.p,
                       ent_at.p, class_idx.p, class_ok.p, cc_rows.p, row_pos.p, bits_class, bits_pred, key.p, w.p);
    radix_sort_u64(pool, key.p, w.p, r, 2 * bits_class + bits_pred + bt, s);
    DevBuf<uint8_t> head_g(pool, r), head_i(pool, r);
    DevBuf<uint32_t> gpos(pool, (size_t)r + 1), dpos(pool, (size_t)r + 1);
    DevBuf<uint64_t> wpos(pool, (size_t)r + 1);
    hipLaunchKernelGGL(k_ms_cc_flags, dim3(blocks(r)), dim3(256), 0, s, key.p, r, bt, head_g.p, head_i.p);
    exclusive_scan_u8_u32(pool, head_g.p, gpos.p, r, gpos.p + r, s);
    exclusive_scan_u8_u32(pool, head_i.p, dpos.p, r, dpos.p + r, s);
    exclusive_scan_u32_u64(pool, w.p, wpos.p, r, wpos.p + r, s);
    uint32_t G = 0;
    fetch.add(&G, gpos.p + r, 4);
    fetch.sync();
    DevBuf<uint64_t> g_key(pool, G), g_rows(pool, G), g_dist(pool, G);
    DevBuf<uint32_t> g_start(pool, G);
    hipLaunchKernelGGL(k_ms_cc_groups, dim3(blocks(r)), dim3(256), 0, s, key.p, head_g.p, gpos.p, r, bt, g_key.p, g_start.p);
    hipLaunchKernelGGL(k_ms_cc_finish, dim3(blocks(G)), dim3(256), 0, s, g_start.p, G, r, wpos.p, dpos.p, g_rows.p, g_dist.p);
    std::vector<uint64_t> hk(G), hr(G), hd(G);
    fetch.add(hk.data(), g_key.p, (size_t)G * 8);
    fetch.add(hr.data(), g_rows.p, (size_t)G * 8);
    fetch.add(hd.data(), g_dist.p, (size_t)G * 8);
    fetch.sync();
    const uint64_t cm = (1ull << bits_class) - 1, pm = (1ull << bits_pred) - 1;
    for (uint32_t g = 0; g < G; ++g) {
      ms.cc_obj.push_back((int64_t)h_class[hk[g] & cm]);
      ms.cc_pred.push_back(ps.pred_terms[(hk[g] >> bits_class) & pm]);
      ms.cc_sub.push_back((int64_t)h_class[(hk[g] >> (bits_class + bits_pred)) & cm]);
      ms.cc_rows.push_back((int64_t)hr[g]);
      ms.cc_dist.push_back((int64_t)hd[g]);
    }
  }
  DX_HIP(hipGetLastError());
  finish();
}

void run_measures(dx_ctx* ctx, dx_measure_sizes* out) {
  const PartitionState& ps = ctx->part;
  if (ps.wide_keys) run_measures_t<u128>(ctx, out, ps.uniq_w, ps.local_keys_w);
  else run_measures_t<uint64_t>(ctx, out, ps.uniq, ps.local_keys);
}

}  // namespace dx
