// Host-only driver of the library's Parquet writer (dx_parquet.cpp) for the CPU tests: writes one file whose
// first two columns are ids over a small pool of terms and whose third column is given as strings.
//   parquet_writer_harness <out.parquet> <rows> <distinct terms> [plan | plan-snappy | plan-packed]
// With "plan" the file is put together the way the GPU encoder's host side does it: plan_parquet_file lays the file out
// from the pages' sizes, and the page bodies (here made by plain loops) are dropped into the places it names; the third
// column is then a plain page. "plan-snappy": the pages are snappy streams whose bodies travel as one literal each.
// "plan-packed": the pages of terms are packed the way the device packs them (each record as a copy of the first bytes
// of the record before it plus a literal: the same rules, restated here), so that the rules meet a reader on the CPU.
// Row i holds (term[i % T], term[(7 * i + 3) % T], "row<i>"); term k is "<t" + k + ">" + k times 'x' (so lengths
// vary, the empty-ish and the long one included), term 1 is non-ASCII.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dx_state.h"

static void put_term(std::string& out, const char* p, size_t n) {
  const uint32_t len = (uint32_t)n;
  out.append(reinterpret_cast<const char*>(&len), 4);
  out.append(p, n);
}

// A page of (4-byte length, bytes) records as snappy elements, by the device's rules (k_enc_terms_packed in dx_layout.hip).
static void put_lit(std::string& out, const std::string& bytes) {
  if (bytes.empty()) return;
  const size_t n = bytes.size() - 1;
  if (n < 60) out.push_back((char)(n << 2));
  else {
    const int k = n < (1u << 8) ? 1 : n < (1u << 16) ? 2 : n < (1u << 24) ? 3 : 4;
    out.push_back((char)((59 + k) << 2));
    for (int i = 0; i < k; ++i) out.push_back((char)(n >> (8 * i)));
  }
  out += bytes;
}
static std::string pack_terms(const std::vector<std::string>& terms) {
  std::string out;
  for (size_t j = 0; j < terms.size(); ++j) {
    const std::string& t = terms[j];
    const uint32_t len = (uint32_t)t.size();
    const std::string word(reinterpret_cast<const char*>(&len), 4);
    size_t head = 4 + t.size(), copy = 0, tail = 0, back = 0;
    if (j > 0 && 4 + terms[j - 1].size() <= 65535) {
      const std::string& p = terms[j - 1];
      back = 4 + p.size();
      size_t m = 0;
      while (m < t.size() && m < p.size() && t[m] == p[m]) ++m;
      if (t.size() == p.size()) { head = 0; copy = 4 + m; tail = t.size() - m; }
      else if (m >= 4) { head = 4; copy = m; tail = t.size() - m; }
    }
    if (head) put_lit(out, head == 4 ? word : word + t);
    for (size_t left = copy; left;) {
      const size_t l = left < 64 ? left : 64;
      out.push_back((char)(2 | ((l - 1) << 2)));
      out.push_back((char)back);
      out.push_back((char)(back >> 8));
      left -= l;
    }
    if (tail) put_lit(out, t.substr(t.size() - tail));
  }
  return out;
}

// the file through the planner: per row group and column the distinct ids in ascending order are the dictionary
static int write_planned(const char* path, uint64_t rows, const std::vector<int64_t>& a, const std::vector<int64_t>& b, const std::vector<std::string>& c,
                         const std::string& pool, const std::vector<uint64_t>& off, bool snappy, bool packed) {
  std::vector<uint64_t> group_rows;
  std::vector<dx::PqChunkIn> in;
  std::vector<std::string> dict_body, data_body;
  for (uint64_t r0 = 0; r0 < rows; r0 += dx::kParquetRowGroupRows) {
    const uint64_t m = std::min<uint64_t>(dx::kParquetRowGroupRows, rows - r0);
    group_rows.push_back(m);
    for (const std::vector<int64_t>* col : {&a, &b}) {
      std::vector<int64_t> u(col->begin() + r0, col->begin() + r0 + m);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      std::string db;
      std::vector<std::string> terms;
      for (int64_t id : u) { put_term(db, pool.data() + off[id], off[id + 1] - off[id]); terms.push_back(pool.substr(off[id], off[id + 1] - off[id])); }
      dx::PqChunkIn ci;
      ci.num_values = m; ci.n_dict = u.size(); ci.dict_bytes = db.size();
      ci.snappy = snappy;
      if (packed) { db = pack_terms(terms); ci.terms_packed = db.size(); }
      in.push_back(ci);
      int width = 0;
      while ((1ull << width) < u.size()) ++width;
      std::string packed((m + 7) / 8 * width, '\0');
      for (uint64_t i = 0; i < m; ++i) {
        const uint64_t v = std::lower_bound(u.begin(), u.end(), (*col)[r0 + i]) - u.begin();
        for (int k = 0; k < width; ++k)
          if (v >> k & 1) packed[(i * width + k) >> 3] |= (char)(1 << ((i * width + k) & 7));
      }
      dict_body.push_back(db);
      data_body.push_back(packed);
    }
    std::string plain;
    for (uint64_t i = 0; i < m; ++i) put_term(plain, c[r0 + i].data(), c[r0 + i].size());
    dx::PqChunkIn ci;
    ci.num_values = m; ci.plain = true; ci.plain_bytes = plain.size();
    ci.snappy = snappy;
    if (packed) { plain = pack_terms(std::vector<std::string>(c.begin() + r0, c.begin() + r0 + m)); ci.terms_packed = plain.size(); }
    in.push_back(ci);
    dict_body.push_back("");
    data_body.push_back(plain);
  }
  const dx::PqFilePlan plan = dx::plan_parquet_file({"_1", "_2", "parts"}, group_rows, in);
  std::string file(plan.size, '\xEE');
  size_t covered = 0;
  for (auto& h : plan.host) { memcpy(&file[h.first], h.second.data(), h.second.size()); covered += h.second.size(); }
  for (size_t k = 0; k < in.size(); ++k) {
    const dx::PqChunkOut& o = plan.chunks[k];
    if (!in[k].plain) {
      memcpy(&file[o.dict_at], dict_body[k].data(), dict_body[k].size());
      covered += dict_body[k].size();
      if (o.data_bytes != data_body[k].size()) { fprintf(stderr, "packed size differs\n"); return 1; }
    }
    memcpy(&file[o.data_at], data_body[k].data(), data_body[k].size());
    covered += data_body[k].size();
  }
  if (covered != plan.size) { fprintf(stderr, "the plan leaves %zu bytes of the file to nobody\n", (size_t)plan.size - covered); return 1; }
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(file.data(), 1, file.size(), f) != file.size() || fclose(f) != 0) return 1;
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 4 && argc != 5) return 2;
  const uint64_t rows = strtoull(argv[2], nullptr, 10), T = strtoull(argv[3], nullptr, 10);
  std::string pool;
  std::vector<uint64_t> off{0};
  for (uint64_t k = 0; k < T; ++k) {
    std::string t = k == 1 ? std::string("\"h\xC3\xA9llo \xF0\x9F\x99\x82\"@fr") : "<t" + std::to_string(k) + ">" + std::string(k % 97, 'x');
    pool += t;
    off.push_back(pool.size());
  }
  std::vector<int64_t> a(rows), b(rows);
  std::vector<std::string> c(rows);
  for (uint64_t i = 0; i < rows; ++i) { a[i] = (int64_t)(i % T); b[i] = (int64_t)((7 * i + 3) % T); c[i] = "row" + std::to_string(i); }
  try {
    if (argc == 5) {
      const std::string mode = argv[4];
      return write_planned(argv[1], rows, a, b, c, pool, off, mode != "plan", mode == "plan-packed");
    }
    dx::ParquetWriter w(argv[1], {"_1", "_2", "parts"});
    // two appends: row groups continue across calls
    const uint64_t half = rows / 2;
    w.append(half, {{a.data(), nullptr}, {b.data(), nullptr}, {nullptr, c.data()}}, (const uint8_t*)pool.data(), off.data(), T);
    w.append(rows - half, {{a.data() + half, nullptr}, {b.data() + half, nullptr}, {nullptr, c.data() + half}}, (const uint8_t*)pool.data(), off.data(), T);
    w.close();
  } catch (const dx::DxError& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
