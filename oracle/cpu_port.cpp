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
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
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
  // kept for the query leg: the interned strings and every vertical-partition table as rows of (s, o) ids,
  // keyed by "<partition as five digits>-_3=<sanitised predicate>" (the tokens of a translated file's partitions line)
  std::vector<sv> strs;
  std::unordered_map<sv, uint32_t> ids;
  std::map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> tables;
};

// DAP.scala:512-514 (uriToStorage): every code point outside [A-Za-z0-9] becomes '_', then lower case
std::string uri_to_storage(sv s) {
  std::string out;
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      out += (c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : ((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9')) ? (char)c : '_';
      ++i;
    } else {
      out += '_';
      i += (c & 0xE0) == 0xC0 ? 2 : (c & 0xF0) == 0xE0 ? 3 : (c & 0xF8) == 0xF0 ? 4 : 1;
    }
  }
  return out;
}
size_t utf16_units(sv s) {
  size_t u = 0;
  for (unsigned char c : s) { if ((c & 0xC0) != 0x80) ++u; if ((c & 0xF8) == 0xF0) ++u; }
  return u;
}
std::string table_key(uint32_t q, sv pred) {
  char part[8];
  snprintf(part, sizeof part, "%05u", q);
  return std::string(part) + "-_3=" + uri_to_storage(pred);
}

uint64_t fnv(sv s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}
uint64_t mix(uint64_t h) { h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; return h ^ (h >> 33); }

Result* run(const char* text, uint64_t n, const char* class_index, int nparts, double* seconds, bool keep_tables = false) {
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
  if (keep_tables) {  // outside the timed part: the tables a query leg reads (predicates of 150 units or more have none, DAP.scala:434)
    for (const auto& r : rows) {
      const uint32_t q = std::get<0>(r), p = std::get<1>(r);
      if (utf16_units(strs[p]) >= 150) continue;
      res->tables[table_key(q, strs[p])].emplace_back(std::get<2>(r), std::get<3>(r));
    }
    res->strs = strs;
    res->ids = intern;
  }
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


// ---- the same restatement on T threads ------------------------------------------------------------------
// One dictionary shared by all threads (64 shards by hash, a lock each), one pass per stage with the work split by
// text range or by a hash of the item, so that every "distinct" of the reference is a sort + unique inside one
// thread's bucket. Nothing here depends on order: the statistics and the order-independent digests equal run()'s.
template <class F>
void parallel_for(int T, F f) {
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([=] { f(t); });
  f(0);
  for (auto& x : th) x.join();
}

Result* run_mt(const char* text, uint64_t n, const char* class_index, int nparts, int T, double* seconds, bool keep_tables) {
  const auto t0 = std::chrono::steady_clock::now();
  if (T < 1) T = 1;
  auto* res = new Result();
  res->part_rows.assign(nparts, 0);
  res->part_digest.assign(nparts, 0);
  // ---- text ranges cut at line ends
  std::vector<uint64_t> cut(T + 1, n);
  cut[0] = 0;
  for (int t = 1; t < T; ++t) {
    uint64_t c = n / T * t;
    if (c < cut[t - 1]) c = cut[t - 1];
    while (c < n && c > 0 && text[c - 1] != '\n' && text[c - 1] != '\r') ++c;
    if (c < n && c > 0 && text[c - 1] == '\r' && text[c] == '\n') ++c;
    cut[t] = c;
  }
  // ---- parse (Preprocessor.scala:119-122) into string views, then intern through the sharded dictionary
  constexpr int kShards = 64;
  struct Shard { std::mutex m; std::unordered_map<sv, uint32_t> map; std::vector<sv> strs; };
  std::vector<Shard> shard(kShards);
  struct Triple { uint32_t s, p, o; };
  std::vector<std::vector<Triple>> tri(T);
  std::atomic<int> error{0};
  auto intern = [&](sv x) -> uint32_t {  // (shard << 26) | index in shard; made dense below
    const uint64_t h = std::hash<sv>{}(x);
    Shard& sh = shard[h >> 58];
    std::lock_guard<std::mutex> g(sh.m);
    auto it = sh.map.find(x);
    if (it != sh.map.end()) return (uint32_t)((h >> 58) << 26) | it->second;
    const uint32_t id = (uint32_t)sh.strs.size();
    sh.strs.push_back(x);
    sh.map.emplace(x, id);
    return (uint32_t)((h >> 58) << 26) | id;
  };
  parallel_for(T, [&](int t) {
    auto& out = tri[t];
    sv last_s, last_p;
    uint32_t last_s_id = 0, last_p_id = 0;
    for (uint64_t i = cut[t]; i < cut[t + 1];) {
      uint64_t e = i;
      while (e < n && text[e] != '\n' && text[e] != '\r') ++e;
      sv line(text + i, e - i);
      uint64_t next = e;
      if (next < n) next += (text[next] == '\r' && next + 1 < n && text[next + 1] == '\n') ? 2 : 1;
      i = next;
      if (!line.empty() && line[0] == '#') continue;
      size_t p = 0;
      while (p < line.size() && !jws(line[p])) ++p;
      if (p >= line.size()) { error = 1; return; }
      sv f0 = line.substr(0, p);
      while (p < line.size() && jws(line[p])) ++p;
      size_t b1 = p;
      while (p < line.size() && !jws(line[p])) ++p;
      if (p >= line.size()) { error = 1; return; }
      sv f1 = line.substr(b1, p - b1);
      while (p < line.size() && jws(line[p])) ++p;
      bool bad = false;
      sv f2 = trim(drop_right_units(line.substr(p), 2, bad));
      if (bad) { error = 1; return; }
      f0 = trim(f0); f1 = trim(f1);
      // a subject's triples follow one another and predicates repeat: the previous line's ids are tried first
      if (f0 != last_s || out.empty()) { last_s = f0; last_s_id = intern(f0); }
      if (f1 != last_p || out.empty()) { last_p = f1; last_p_id = intern(f1); }
      out.push_back({last_s_id, last_p_id, intern(f2)});
    }
  });
  if (error) { res->error = 1; return res; }
  std::vector<uint32_t> base(kShards + 1, 0);
  for (int k = 0; k < kShards; ++k) base[k + 1] = base[k] + (uint32_t)shard[k].strs.size();
  const uint32_t n_terms = base[kShards];
  std::vector<sv> strs(n_terms);
  parallel_for(T, [&](int t) {
    for (int k = t; k < kShards; k += T) std::copy(shard[k].strs.begin(), shard[k].strs.end(), strs.begin() + base[k]);
    for (auto& x : tri[t]) {
      x.s = base[x.s >> 26] + (x.s & 0x3FFFFFFu);
      x.p = base[x.p >> 26] + (x.p & 0x3FFFFFFu);
      x.o = base[x.o >> 26] + (x.o & 0x3FFFFFFu);
    }
  });
  auto dense_of = [&](sv x, uint32_t& id) -> bool {
    const uint64_t h = std::hash<sv>{}(x);
    auto it = shard[h >> 58].map.find(x);
    if (it == shard[h >> 58].map.end()) return false;
    id = base[h >> 58] + it->second;
    return true;
  };
  for (auto& v : tri) res->n_triples += v.size();
  uint32_t type_id = ~0u;
  const bool has_type = dense_of(sv(kType), type_id);
  (void)has_type;
  // ---- class index (Loader.scala:114-123)
  std::vector<uint64_t> Pm(n_terms, 0);
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
      uint32_t id;
      if (dense_of(cls, id)) Pm[id] = mask;  // a class listed twice: the later line wins, as in a Scala Map
    }
  }
  // ---- vertices (Preprocessor.scala:131-149): lists guarded by striped locks (a subject may occur in two threads' ranges)
  struct V { std::vector<uint32_t> types; std::vector<std::pair<uint32_t, uint32_t>> labels; };
  std::vector<V> attr(n_terms);
  std::vector<uint8_t> is_vertex(n_terms, 0);
  constexpr int kStripes = 4096;
  std::vector<std::mutex> stripe(kStripes);
  auto literal = [&](uint32_t id) { return !strs[id].empty() && strs[id][0] == '"'; };
  parallel_for(T, [&](int t) {
    for (const auto& x : tri[t]) {
      if (x.p == type_id) {
        std::lock_guard<std::mutex> g(stripe[x.s % kStripes]);
        is_vertex[x.s] = 1; attr[x.s].types.push_back(x.o);
      } else if (literal(x.o)) {
        std::lock_guard<std::mutex> g(stripe[x.s % kStripes]);
        is_vertex[x.s] = 1; attr[x.s].labels.push_back({x.p, x.o});
      } else {
        reinterpret_cast<std::atomic<uint8_t>*>(&is_vertex[x.s])->store(1, std::memory_order_relaxed);
        reinterpret_cast<std::atomic<uint8_t>*>(&is_vertex[x.o])->store(1, std::memory_order_relaxed);
      }
    }
  });
  {
    std::vector<uint64_t> cnt(T, 0);
    parallel_for(T, [&](int t) { for (uint32_t i = t; i < n_terms; i += T) cnt[t] += is_vertex[i]; });
    for (auto c : cnt) res->n_vertices += c;
  }
  // ---- edges (Preprocessor.scala:153-167): distinct inside the bucket of their hash
  auto hash3 = [](uint32_t a, uint32_t b, uint32_t c) { return mix(((uint64_t)a << 32 | b) * 0x9E3779B97F4A7C15ull + c); };
  std::vector<std::vector<std::vector<Triple>>> ebk(T, std::vector<std::vector<Triple>>(T));
  parallel_for(T, [&](int t) {
    for (const auto& x : tri[t])
      if (x.p != type_id && !literal(x.o)) ebk[t][hash3(x.p, x.s, x.o) % T].push_back(x);
  });
  std::vector<std::vector<Triple>> edges(T);
  auto key = [](const Triple& t) { return std::make_tuple(t.p, t.s, t.o); };
  // ---- partitionInstances (DAP.scala:575-604): rows (partition, p, s, o), distinct inside the bucket of their hash
  using Row = std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>;
  std::vector<std::vector<std::vector<Row>>> rbk(T, std::vector<std::vector<Row>>(T));
  auto vmask = [&](const V& v) { uint64_t m = 0; for (uint32_t c : v.types) m |= Pm[c]; return m; };
  parallel_for(T, [&](int t) {
    auto& ed = edges[t];
    for (int src = 0; src < T; ++src) ed.insert(ed.end(), ebk[src][t].begin(), ebk[src][t].end());
    std::sort(ed.begin(), ed.end(), [&](const Triple& a, const Triple& b) { return key(a) < key(b); });
    ed.erase(std::unique(ed.begin(), ed.end(), [&](const Triple& a, const Triple& b) { return key(a) == key(b); }), ed.end());
    auto emit = [&](uint64_t m, uint32_t s, uint32_t o, uint32_t p) {
      for (int q = 0; q < nparts; ++q)
        if ((m >> q) & 1) rbk[t][hash3(p, s, o) % T].emplace_back((uint32_t)q, p, s, o);
    };
    for (const auto& e : ed) {
      const V &src = attr[e.s], &dst = attr[e.o];
      if (!src.types.empty() || !dst.types.empty()) {
        emit(vmask(src) | vmask(dst), e.s, e.o, e.p);                                    // :575-580
        for (uint32_t c : src.types) emit(Pm[c], e.s, c, type_id);                        // :583-593
        for (uint32_t c : dst.types) emit(Pm[c], e.o, c, type_id);
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
  });
  for (auto& e : edges) res->n_edges += e.size();
  std::vector<std::vector<Row>> rows(T);
  std::vector<std::vector<uint64_t>> prow(T, std::vector<uint64_t>(nparts, 0)), pdig(T, std::vector<uint64_t>(nparts, 0));
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> tabs(T), node(T);
  parallel_for(T, [&](int t) {
    auto& r = rows[t];
    for (int src = 0; src < T; ++src) r.insert(r.end(), rbk[src][t].begin(), rbk[src][t].end());
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());                                     // :604
    for (const auto& x : r) {
      const uint32_t q = std::get<0>(x), p = std::get<1>(x), s = std::get<2>(x), o = std::get<3>(x);
      ++prow[t][q];
      pdig[t][q] += mix(fnv(strs[s]) ^ mix(fnv(strs[o]) + 0x9E3779B97F4A7C15ull) ^ mix(fnv(strs[p]) + 0x632BE59BD9B4E019ull));
      if (tabs[t].empty() || tabs[t].back() != std::make_pair(q, p)) tabs[t].emplace_back(q, p);
      if (p == type_id) node[t].emplace_back(s, o);
    }
    // the node index is distinct over the partitions (DAP.scala:402-412): a row's (s, o) lands in the bucket of (p, s, o),
    // the same for every partition it is in
    std::sort(node[t].begin(), node[t].end());
    node[t].erase(std::unique(node[t].begin(), node[t].end()), node[t].end());
  });
  {
    std::vector<std::pair<uint32_t, uint32_t>> all;
    for (int t = 0; t < T; ++t) {
      for (int q = 0; q < nparts; ++q) { res->part_rows[q] += prow[t][q]; res->part_digest[q] += pdig[t][q]; }
      all.insert(all.end(), tabs[t].begin(), tabs[t].end());
      res->n_node_index += node[t].size();
    }
    std::sort(all.begin(), all.end());
    res->n_tables = std::unique(all.begin(), all.end()) - all.begin();
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (keep_tables) {
    for (int t = 0; t < T; ++t)
      for (const auto& x : rows[t]) {
        const uint32_t q = std::get<0>(x), p = std::get<1>(x);
        if (utf16_units(strs[p]) >= 150) continue;
        res->tables[table_key(q, strs[p])].emplace_back(std::get<2>(x), std::get<3>(x));
      }
    res->strs = strs;
    res->ids.reserve(strs.size() * 2);
    for (uint32_t i = 0; i < n_terms; ++i) res->ids.emplace(strs[i], i);
  }
  return res;
}

// ---- query leg: the translated statement's sub-selects as scans with constant filters and hash joins ---------
// What Spark does with the SQL text (QueryProcessor.scala:184-195: cached-table scans, equality filters, equi-joins
// on the shared variables, count), as a left-deep chain of hash joins: the smallest filtered relation first, then
// always the smallest relation that shares a variable with what is bound (a cross product only if none does).
// Two ways of keying the joins: on the terms' ids (what an id-encoded engine compares), or on the terms' strings
// (what the reference compares: its tables hold strings), with the same plan. The probe side is split over T threads.
struct SubSel {
  std::string table;
  bool has_s = false, has_o = false;
  std::string s_const, o_const;
  int s_var = -1, o_var = -1;  // variable number of the projected column, -1 = not projected
};
struct Rel {
  std::vector<int> vars;
  std::vector<uint32_t> data;  // row-major, vars.size() ids per row
  size_t rows() const { return vars.empty() ? 0 : data.size() / vars.size(); }
};

uint64_t run_query(const Result* R, const std::vector<SubSel>& subs, int T, bool string_keyed, int* err) {
  *err = 0;
  if (T < 1) T = 1;
  const auto& strs = R->strs;
  std::vector<Rel> rel(subs.size());
  // scans (Spark: InMemoryTableScan + Filter)
  for (size_t i = 0; i < subs.size(); ++i) {
    const SubSel& q = subs[i];
    auto it = R->tables.find(q.table);
    if (it == R->tables.end()) { *err = 2; return 0; }
    const auto& tab = it->second;
    Rel& r = rel[i];
    if (q.s_var >= 0) r.vars.push_back(q.s_var);
    if (q.o_var >= 0 && q.o_var != q.s_var) r.vars.push_back(q.o_var);
    std::vector<std::vector<uint32_t>> part(T);
    // keyed on strings the filter compares strings, as the reference's tables hold them; keyed on ids the constant is
    // looked up once and ids are compared (a constant the dictionary does not know matches nothing)
    uint32_t s_id = ~0u, o_id = ~0u;
    if (!string_keyed) {
      if (q.has_s) { auto f = R->ids.find(sv(q.s_const)); if (f != R->ids.end()) s_id = f->second; }
      if (q.has_o) { auto f = R->ids.find(sv(q.o_const)); if (f != R->ids.end()) o_id = f->second; }
    }
    parallel_for(T, [&](int t) {
      const size_t b = tab.size() * t / T, e = tab.size() * (t + 1) / T;
      for (size_t k = b; k < e; ++k) {
        const uint32_t s = tab[k].first, o = tab[k].second;
        if (q.has_s && !(string_keyed ? strs[s] == sv(q.s_const) : s == s_id)) continue;
        if (q.has_o && !(string_keyed ? strs[o] == sv(q.o_const) : o == o_id)) continue;
        if (q.s_var >= 0 && q.o_var == q.s_var && s != o) continue;
        if (q.s_var >= 0) part[t].push_back(s);
        if (q.o_var >= 0 && q.o_var != q.s_var) part[t].push_back(o);
        if (r.vars.empty()) part[t].push_back(0);
      }
    });
    if (r.vars.empty()) r.vars.push_back(-1 - (int)i);  // nothing projected: one column that joins with nothing
    for (auto& p : part) r.data.insert(r.data.end(), p.begin(), p.end());
  }
  // left-deep order
  std::vector<bool> used(subs.size(), false);
  size_t first = 0;
  for (size_t i = 1; i < rel.size(); ++i) if (rel[i].rows() < rel[first].rows()) first = i;
  Rel cur = rel[first];
  used[first] = true;
  for (size_t step = 1; step < rel.size(); ++step) {
    int best = -1; bool best_conn = false;
    for (size_t i = 0; i < rel.size(); ++i) {
      if (used[i]) continue;
      bool conn = false;
      for (int v : rel[i].vars) conn |= std::find(cur.vars.begin(), cur.vars.end(), v) != cur.vars.end();
      if (best < 0 || (conn && !best_conn) || (conn == best_conn && rel[i].rows() < rel[best].rows())) { best = (int)i; best_conn = conn; }
    }
    const Rel& nx = rel[best];
    used[best] = true;
    std::vector<int> kc, kn, newcols;  // key columns in cur / in nx, columns of nx that are new
    for (size_t c = 0; c < nx.vars.size(); ++c) {
      auto f = std::find(cur.vars.begin(), cur.vars.end(), nx.vars[c]);
      if (f != cur.vars.end()) { kc.push_back((int)(f - cur.vars.begin())); kn.push_back((int)c); } else newcols.push_back((int)c);
    }
    // build on the smaller side, probe with the larger one split over the threads
    const bool build_cur = cur.rows() <= nx.rows();
    const Rel &B = build_cur ? cur : nx, &P = build_cur ? nx : cur;
    const std::vector<int> &kb = build_cur ? kc : kn, &kp = build_cur ? kn : kc;
    const size_t wb = B.vars.size(), wp = P.vars.size();
    auto key_of = [&](const Rel& r, size_t row, const std::vector<int>& k) -> uint64_t {
      uint64_t h = 0x9E3779B97F4A7C15ull;
      for (int c : k) {
        const uint32_t id = r.data[row * r.vars.size() + c];
        h = mix(h ^ (string_keyed ? (uint64_t)std::hash<sv>{}(strs[id]) : (uint64_t)id));
      }
      return h;
    };
    auto keys_equal = [&](size_t rb, size_t rp) -> bool {
      for (size_t j = 0; j < kb.size(); ++j) {
        const uint32_t a = B.data[rb * wb + kb[j]], b = P.data[rp * wp + kp[j]];
        if (string_keyed ? strs[a] != strs[b] : a != b) return false;
      }
      return true;
    };
    std::unordered_multimap<uint64_t, uint32_t> ht;
    ht.reserve(B.rows() * 2 + 16);
    for (size_t r = 0; r < B.rows(); ++r) ht.emplace(key_of(B, r, kb), (uint32_t)r);
    Rel out;
    out.vars = cur.vars;
    for (int c : newcols) out.vars.push_back(nx.vars[c]);
    const bool last = step + 1 == rel.size();
    std::vector<std::vector<uint32_t>> part(T);
    std::vector<uint64_t> cnt(T, 0);
    parallel_for(T, [&](int t) {
      const size_t b = P.rows() * t / T, e = P.rows() * (t + 1) / T;
      for (size_t rp = b; rp < e; ++rp) {
        auto range = ht.equal_range(key_of(P, rp, kp));
        for (auto it2 = range.first; it2 != range.second; ++it2) {
          const size_t rb = it2->second;
          if (!keys_equal(rb, rp)) continue;
          ++cnt[t];
          if (last) continue;  // the count is all that is asked for
          const size_t rc = build_cur ? rb : rp, rn = build_cur ? rp : rb;
          for (size_t c = 0; c < cur.vars.size(); ++c) part[t].push_back(cur.data[rc * cur.vars.size() + c]);
          for (int c : newcols) part[t].push_back(nx.data[rn * nx.vars.size() + c]);
        }
      }
    });
    if (last) { uint64_t n = 0; for (auto c : cnt) n += c; return n; }
    for (auto& p : part) out.data.insert(out.data.end(), p.begin(), p.end());
    cur = std::move(out);
  }
  return cur.rows();
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
// The same on `threads` threads; keep_tables != 0 keeps the vertical-partition tables for cpu_port_query (the text
// must then stay where it is until cpu_port_free: the tables' strings point into it).
void* cpu_port_run_mt(const char* text, uint64_t n, const char* class_index_text, int nparts, int threads, int keep_tables, double* seconds) {
  return threads <= 1 ? run(text, n, class_index_text, nparts, seconds, keep_tables != 0)
                      : run_mt(text, n, class_index_text, nparts, threads, seconds, keep_tables != 0);
}
// Rows of a kept table, or -1.
int64_t cpu_port_table_rows(void* h, const char* key) {
  const Result* r = (const Result*)h;
  auto it = r->tables.find(key);
  return it == r->tables.end() ? -1 : (int64_t)it->second.size();
}
// Count of a translated statement given as its sub-selects: table key, constant filters (null = none), variable
// numbers of the projected s / o columns (-1 = not projected). Joins are natural joins on equal variable numbers.
// *seconds = wall time of the execution (scans + joins + count), tables resident. Returns -1 for an unknown table.
int64_t cpu_port_query(void* h, int n_subs, const char** tables, const char** s_const, const char** o_const, const int* s_var,
                       const int* o_var, int threads, int string_keyed, double* seconds) {
  std::vector<SubSel> subs(n_subs);
  for (int i = 0; i < n_subs; ++i) {
    subs[i].table = tables[i];
    if (s_const[i]) { subs[i].has_s = true; subs[i].s_const = s_const[i]; }
    if (o_const[i]) { subs[i].has_o = true; subs[i].o_const = o_const[i]; }
    subs[i].s_var = s_var[i];
    subs[i].o_var = o_var[i];
  }
  int err = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const uint64_t n = run_query((const Result*)h, subs, threads, string_keyed != 0, &err);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return err ? -1 : (int64_t)n;
}
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
