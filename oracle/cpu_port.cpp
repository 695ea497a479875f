// CPU restatement of the partitioner's post-DAP stage in C++ (TEST INFRASTRUCTURE ONLY).
//
// Second, faster checker beside oracle/dap_oracle.py and the CPU baseline that bench.py times on the
// GPU box's host cores. It follows the reference's structure -- vertices with type and label lists,
// distinct edges, GraphX-style triplets, three emission streams, union + distinct, split by partition
// and predicate -- on interned strings (Preprocessor.scala:116-192, partitioner/DAP.scala:360-437,
// 567-629). Nothing in the product path may link or load it.
//
// PARITY UNPINNED: like the Python restatement, it is pinned only by the hand-derived example in
// tests/golden/appendix_c.json; the reference ships no expected outputs and cannot run offline.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <string_view>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace {

using sv = std::string_view;
const char* kType = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>";

inline bool jws(unsigned char c) { return c == ' ' || c == '\t' || c == 0x0B || c == 0x0C; }
inline sv trim(sv s) {
  while (!s.empty() && (unsigned char)s.front() <= 0x20) s.remove_prefix(1);
  while (!s.empty() && (unsigned char)s.back() <= 0x20) s.remove_suffix(1);
  return s;
}
sv drop_right_units(sv s, int units, bool& bad) {
  while (units > 0 && !s.empty()) {
    size_t q = s.size() - 1;
    while (q > 0 && ((unsigned char)s[q] & 0xC0) == 0x80) --q;
    const int u = (((unsigned char)s[q] & 0xF8) == 0xF0) ? 2 : 1;
    if (u > units) { bad = true; break; }
    units -= u;
    s = s.substr(0, q);
  }
  return s;
}

struct Result {
  uint64_t n_triples = 0, n_vertices = 0, n_edges = 0, n_node_index = 0, n_tables = 0;
  std::vector<uint64_t> part_rows;
  std::vector<uint64_t> part_digest;  // order-independent digest of the partition's (s, o, p) strings
  // instance-scale measures, strings as their FNV-1a hashes: (class, node freq, node count) and
  // (subType, pred, objType, rows, distinct object instances)
  std::vector<uint64_t> classes, cc;
  int error = 0;
};

uint64_t fnv(sv s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}
uint64_t mix(uint64_t h) { h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; return h ^ (h >> 33); }

Result* run(const char* text, uint64_t n, const char* class_index, int nparts, double* seconds) {
  const auto t0 = std::chrono::steady_clock::now();
  auto* res = new Result();
  res->part_rows.assign(nparts, 0);
  res->part_digest.assign(nparts, 0);
  // ---- parse (Preprocessor.scala:119-122)
  std::unordered_map<sv, uint32_t> intern;
  std::vector<sv> strs;
  auto id_of = [&](sv s) {
    auto it = intern.find(s);
    if (it != intern.end()) return it->second;
    const uint32_t id = (uint32_t)strs.size();
    strs.push_back(s);
    intern.emplace(s, id);
    return id;
  };
  struct Triple { uint32_t s, p, o; };
  std::vector<Triple> triples;
  const uint32_t type_id = id_of(sv(kType));
  for (uint64_t i = 0; i < n;) {
    uint64_t e = i;
    while (e < n && text[e] != '\n' && text[e] != '\r') ++e;
    sv line(text + i, e - i);
    uint64_t next = e;
    if (next < n) next += (text[next] == '\r' && next + 1 < n && text[next + 1] == '\n') ? 2 : 1;
    i = next;
    if (!line.empty() && line[0] == '#') continue;
    size_t p = 0;
    while (p < line.size() && !jws(line[p])) ++p;
    if (p >= line.size()) { res->error = 1; return res; }
    sv t0 = line.substr(0, p);
    while (p < line.size() && jws(line[p])) ++p;
    size_t b1 = p;
    while (p < line.size() && !jws(line[p])) ++p;
    if (p >= line.size()) { res->error = 1; return res; }
    sv t1 = line.substr(b1, p - b1);
    while (p < line.size() && jws(line[p])) ++p;
    bool bad = false;
    sv t2 = trim(drop_right_units(line.substr(p), 2, bad));
    if (bad) { res->error = 1; return res; }
    triples.push_back({id_of(trim(t0)), id_of(trim(t1)), id_of(t2)});
  }
  res->n_triples = triples.size();
  // ---- class index (Loader.scala:114-123)
  std::unordered_map<uint32_t, uint64_t> cls_mask;
  {
    sv ci(class_index);
    while (!ci.empty()) {
      size_t nl = ci.find('\n');
      sv line = ci.substr(0, nl);
      ci = nl == sv::npos ? sv() : ci.substr(nl + 1);
      if (line.empty()) continue;
      size_t tab = line.find('\t');
      sv cls = line.substr(0, tab);
      uint64_t mask = 0;
      if (tab != sv::npos) {
        sv rest = line.substr(tab + 1);
        while (!rest.empty()) {
          size_t c = rest.find(',');
          sv tok = rest.substr(0, c);
          rest = c == sv::npos ? sv() : rest.substr(c + 1);
          if (!tok.empty()) mask |= 1ull << std::stoi(std::string(tok));
        }
      }
      auto it = intern.find(cls);
      if (it != intern.end()) cls_mask[it->second] = mask;
    }
  }
  auto P = [&](uint32_t c) { auto it = cls_mask.find(c); return it == cls_mask.end() ? 0ull : it->second; };
  // ---- vertices (Preprocessor.scala:131-149)
  struct V { std::vector<uint32_t> types; std::vector<std::pair<uint32_t, uint32_t>> labels; bool is_vertex = false; };
  std::vector<V> attr(strs.size());
  auto literal = [&](uint32_t id) { return !strs[id].empty() && strs[id][0] == '"'; };
  for (const auto& t : triples) {
    if (t.p == type_id) { attr[t.s].is_vertex = true; attr[t.s].types.push_back(t.o); }
    else if (literal(t.o)) { attr[t.s].is_vertex = true; attr[t.s].labels.push_back({t.p, t.o}); }
    else { attr[t.s].is_vertex = true; attr[t.o].is_vertex = true; }
  }
  for (auto& a : attr) res->n_vertices += a.is_vertex;
  // ---- edges (Preprocessor.scala:153-167)
  std::vector<Triple> edges;
  for (const auto& t : triples)
    if (t.p != type_id && !literal(t.o)) edges.push_back(t);
  auto key = [](const Triple& t) { return std::make_tuple(t.p, t.s, t.o); };
  std::sort(edges.begin(), edges.end(), [&](const Triple& a, const Triple& b) { return key(a) < key(b); });
  edges.erase(std::unique(edges.begin(), edges.end(), [&](const Triple& a, const Triple& b) { return key(a) == key(b); }), edges.end());
  res->n_edges = edges.size();
  // ---- partitionInstances (DAP.scala:575-604): rows as (partition, p, s, o)
  using Row = std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>;
  std::vector<Row> rows;
  auto vmask = [&](const V& v) { uint64_t m = 0; for (uint32_t c : v.types) m |= P(c); return m; };
  auto emit = [&](uint64_t m, uint32_t s, uint32_t o, uint32_t p) {
    for (int q = 0; q < nparts; ++q)
      if ((m >> q) & 1) rows.emplace_back((uint32_t)q, p, s, o);
  };
  for (const auto& e : edges) {
    const V &src = attr[e.s], &dst = attr[e.o];
    if (!src.types.empty() || !dst.types.empty()) {
      emit(vmask(src) | vmask(dst), e.s, e.o, e.p);                                    // :575-580
      for (uint32_t c : src.types) emit(P(c), e.s, c, type_id);                          // :583-593
      for (uint32_t c : dst.types) emit(P(c), e.o, c, type_id);
    }
    const bool has = (!src.labels.empty() && !src.types.empty()) || (!dst.labels.empty() && !dst.types.empty());
    if (has) {                                                                           // :595-602
      for (const V* v : {&src, &dst}) {
        if (v->types.empty()) continue;
        const uint32_t uri = v == &src ? e.s : e.o;
        const uint64_t m = vmask(*v);
        for (const auto& l : v->labels) emit(m, uri, l.second, l.first);
      }
    }
  }
  std::sort(rows.begin(), rows.end());
  rows.erase(std::unique(rows.begin(), rows.end()), rows.end());                          // :604
  // ---- split by partition and predicate (DAP.scala:364-367, 433-437), node index (:402-412)
  std::vector<std::pair<uint32_t, uint32_t>> node;
  uint32_t last_q = ~0u, last_p = ~0u;
  for (const auto& r : rows) {
    const uint32_t q = std::get<0>(r), p = std::get<1>(r), s = std::get<2>(r), o = std::get<3>(r);
    ++res->part_rows[q];
    res->part_digest[q] += mix(fnv(strs[s]) ^ mix(fnv(strs[o]) + 0x9E3779B97F4A7C15ull) ^ mix(fnv(strs[p]) + 0x632BE59BD9B4E019ull));
    if (q != last_q || p != last_p) { ++res->n_tables; last_q = q; last_p = p; }
    if (p == type_id) node.emplace_back(s, o);
  }
  std::sort(node.begin(), node.end());
  node.erase(std::unique(node.begin(), node.end()), node.end());
  res->n_node_index = node.size();
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // ---- instance-scale measures, outside the timed part (Preprocessor.scala:230-264, 312-339)
  {
    std::map<uint32_t, std::pair<uint64_t, uint64_t>> per_class;  // class -> (freq, count)
    std::vector<uint8_t> endpoint(strs.size(), 0);
    using CcRow = std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>;  // subType, pred, objType, inst2
    std::vector<CcRow> cc_rows;
    for (const auto& e : edges) {
      const V &src = attr[e.s], &dst = attr[e.o];
      if (src.types.empty() && dst.types.empty()) continue;
      for (uint32_t c : src.types) ++per_class[c].second;                                 // :329-333
      for (uint32_t c : dst.types) ++per_class[c].second;
      if (!src.types.empty()) endpoint[e.s] = 1;                                           // :313-317
      if (!dst.types.empty()) endpoint[e.o] = 1;
      for (uint32_t st : src.types)                                                        // :232-241, 356-366
        for (uint32_t ot : dst.types)
          if (!strs[st].empty() && !strs[e.p].empty() && !strs[ot].empty()) cc_rows.emplace_back(st, e.p, ot, e.o);
    }
    for (size_t v = 0; v < strs.size(); ++v)
      if (endpoint[v])
        for (uint32_t c : attr[v].types) ++per_class[c].first;                            // :318-320
    for (auto& kv : per_class) {
      res->classes.push_back(fnv(strs[kv.first]));
      res->classes.push_back(kv.second.first);
      res->classes.push_back(kv.second.second);
    }
    std::sort(cc_rows.begin(), cc_rows.end());
    for (size_t i = 0; i < cc_rows.size();) {
      size_t j = i;
      uint64_t rows_n = 0, dist = 0;
      while (j < cc_rows.size() && std::get<0>(cc_rows[j]) == std::get<0>(cc_rows[i]) && std::get<1>(cc_rows[j]) == std::get<1>(cc_rows[i]) &&
             std::get<2>(cc_rows[j]) == std::get<2>(cc_rows[i])) {
        ++rows_n;
        if (j == i || std::get<3>(cc_rows[j]) != std::get<3>(cc_rows[j - 1])) ++dist;     // :246-251
        ++j;
      }
      res->cc.push_back(fnv(strs[std::get<0>(cc_rows[i])]));
      res->cc.push_back(fnv(strs[std::get<1>(cc_rows[i])]));
      res->cc.push_back(fnv(strs[std::get<2>(cc_rows[i])]));
      res->cc.push_back(rows_n);
      res->cc.push_back(dist);
      i = j;
    }
  }
  return res;
}

}  // namespace

extern "C" {
// Runs the restatement; *seconds receives the wall time of the partitioning itself (the measures are computed after it).
void* cpu_port_run(const char* text, uint64_t n, const char* class_index_text, int nparts, double* seconds) {
  return run(text, n, class_index_text, nparts, seconds);
}
// Instance-scale measures of the run: sizes = [classes, cc groups]; classes = 3 words each (FNV-1a of the class, node
// freq, node count), cc = 5 words each (FNV-1a of subType, pred, objType; rows; distinct object instances).
void cpu_port_measure_sizes(void* h, uint64_t* sizes) {
  const Result* r = (const Result*)h;
  sizes[0] = r->classes.size() / 3;
  sizes[1] = r->cc.size() / 5;
}
void cpu_port_measures(void* h, uint64_t* classes, uint64_t* cc) {
  const Result* r = (const Result*)h;
  if (!r->classes.empty()) memcpy(classes, r->classes.data(), r->classes.size() * 8);
  if (!r->cc.empty()) memcpy(cc, r->cc.data(), r->cc.size() * 8);
}
// out: [error, n_triples, n_vertices, n_edges, n_node_index, n_tables, rows[nparts], digest[nparts]]
void cpu_port_stats(void* h, uint64_t* out) {
  const Result* r = (const Result*)h;
  out[0] = (uint64_t)r->error; out[1] = r->n_triples; out[2] = r->n_vertices; out[3] = r->n_edges;
  out[4] = r->n_node_index; out[5] = r->n_tables;
  for (size_t i = 0; i < r->part_rows.size(); ++i) { out[6 + i] = r->part_rows[i]; out[6 + r->part_rows.size() + i] = r->part_digest[i]; }
}
void cpu_port_free(void* h) { delete (Result*)h; }
// Digest of rows given as ids into a string pool (term i = pool[offs[i], offs[i+1])): the same sum run() forms
// for a partition, so that a checker can compare id-encoded rows with this restatement without decoding them.
uint64_t cpu_port_rows_digest(const uint8_t* pool, const uint64_t* offs, uint64_t n_terms, const int64_t* s, const int64_t* o,
                              const int64_t* p, uint64_t n_rows) {
  std::vector<uint64_t> h(n_terms);
  for (uint64_t i = 0; i < n_terms; ++i) h[i] = fnv(sv((const char*)pool + offs[i], offs[i + 1] - offs[i]));
  uint64_t d = 0;
  for (uint64_t r = 0; r < n_rows; ++r)
    d += mix(h[s[r]] ^ mix(h[o[r]] + 0x9E3779B97F4A7C15ull) ^ mix(h[p[r]] + 0x632BE59BD9B4E019ull));
  return d;
}
}
