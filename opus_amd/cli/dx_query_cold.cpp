// dx-query-cold: one translated query, bindings written out, behind the reference's QueryProcessorCold.
//
//   dx-query-cold <partitions> <translated query file> <dataset> <balance> <base path>
//
// Same positional arguments and results file as org.ics.isl.QueryProcessorCold
// (query_processor/QueryProcessorCold.scala:28-94): load the tables of the file's `partitions` line, run its
// statement, write the bindings as CSV under <base><dataset>/result_cold/ (replacing what is there, :56) and
// append Query / Time / Result_count / partitions to results/<dataset>/cold_<N>_bal/results_<TP>.txt
// (cold_<N>/ when balance != 1). Time covers what the reference times, result.write.csv (:119-127): since
// Spark evaluates lazily that is the execution of the statement plus the writing. Time_us is added because GPU
// times are often below a millisecond.
// CSV as Spark's writer leaves it by default: no header, columns in the order of the SELECT list, ',' between
// fields, a field holding ',', '"' or a line break enclosed in '"' with inner '"' written as '\"', an empty
// string as "". Rows are sorted (Spark's order is whatever its tasks produce).
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "diaeresis_hip.h"

static std::vector<std::string> split(const std::string& s, char c) {
  std::vector<std::string> out;
  std::string cur;
  for (char ch : s) {
    if (ch == c) { out.push_back(cur); cur.clear(); } else cur += ch;
  }
  out.push_back(cur);
  return out;
}

static void remove_tree(const std::string& path) {  // HdfsUtils.removeDirInHDFS on a local path
  DIR* d = opendir(path.c_str());
  if (!d) return;
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n == "." || n == "..") continue;
    const std::string p = path + "/" + n;
    struct stat st;
    if (lstat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) remove_tree(p); else unlink(p.c_str());
  }
  closedir(d);
  rmdir(path.c_str());
}

static void csv_field(std::string& out, const char* b, size_t n) {
  if (n == 0) { out += "\"\""; return; }
  bool quote = false;
  for (size_t i = 0; i < n; ++i) quote |= b[i] == ',' || b[i] == '"' || b[i] == '\n' || b[i] == '\r';
  if (!quote) { out.append(b, n); return; }
  out += '"';
  for (size_t i = 0; i < n; ++i) {
    if (b[i] == '"') out += '\\';
    out += b[i];
  }
  out += '"';
}

int main(int argc, char** argv) {
  if (argc - 1 != 5) {  // QueryProcessorCold.scala:31-34
    printf("Arguments: number of partitions, query file ,dataset name, balance, hdfs base path\n");
    return 255;
  }
  const int n = atoi(argv[1]);
  const std::string query_path = argv[2], dataset = argv[3];
  const int bal = atoi(argv[4]);
  std::string base = argv[5];
  if (!base.empty() && base.back() != '/') base += '/';
  const std::string result_dir = "results/" + dataset + "/cold_" + std::to_string(n) + (bal == 1 ? "_bal/" : "/");
  mkdir("results", 0777);
  mkdir(("results/" + dataset).c_str(), 0777);
  mkdir(result_dir.c_str(), 0777);
  const std::string name = query_path.substr(query_path.find_last_of('/') + 1);
  const std::string cold = base + dataset + "/result_cold";
  remove_tree(cold);
  printf("%s\n", query_path.c_str());

  std::ifstream f(query_path);
  if (!f) { fprintf(stderr, "dx-query-cold: cannot open %s\n", query_path.c_str()); return 1; }
  std::string line;
  std::vector<std::string> partitions, statements, keys_seen;  // the reference's query map keeps one statement per key
  int tp = -1;
  while (std::getline(f, line)) {  // :65-80
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.rfind("TP", 0) == 0) tp = atoi(split(line, ' ')[1].c_str());
    else if (line.rfind("partitions", 0) == 0) { auto t = split(line, ' '); partitions = t.size() > 1 ? split(t[1], ',') : std::vector<std::string>(); }
    else if (line.rfind(">>>>>", 0) != 0) {
      const size_t sp = line.find(' ');
      const std::string key = line.substr(0, sp), sql = sp == std::string::npos ? "" : line.substr(sp + 1);
      size_t at = 0;
      while (at < keys_seen.size() && keys_seen[at] != key) ++at;
      if (at == keys_seen.size()) { keys_seen.push_back(key); statements.push_back(sql); }
      else statements[at] = sql;
    }
  }
  if (statements.empty()) {
    fprintf(stderr, "dx-query-cold: %s holds no statement\n", name.c_str());
    return 1;
  }

  dx_ctx* ctx = dx_init(0);
  if (!ctx) { fprintf(stderr, "dx-query-cold: no HIP device (this tool has no CPU path)\n"); return 1; }
  auto fail = [&](const char* what) {
    fprintf(stderr, "dx-query-cold: %s: %s\n", what, dx_last_error(ctx));
    dx_free(ctx);
    return 1;
  };
  if (dx_open_dataset(ctx, base.c_str(), dataset.c_str(), n) != DX_OK) return fail("open");
  {  // preloadTables (:120)
    std::vector<const char*> keys;
    for (const auto& p : partitions) keys.push_back(p.c_str());
    uint64_t loaded = 0;
    if (dx_load_tables(ctx, keys.data(), (int)keys.size(), &loaded) != DX_OK) return fail("load");
  }
  const auto t1 = std::chrono::steady_clock::now();
  dx_result* r = nullptr;
  {  // one statement, or several whose results are joined on the columns they share (executeFinalQuery / customJoin)
    std::vector<const char*> sqls;
    for (const auto& q : statements) sqls.push_back(q.c_str());
    if (dx_query_joined(ctx, sqls.data(), (int)sqls.size(), DX_QUERY_BINDINGS, &r) != DX_OK) return fail("query");
  }
  const uint64_t rows = dx_result_count(r);
  const int cols = dx_result_columns(r);
  std::vector<int64_t> ids(rows * (uint64_t)cols + 1);
  if (rows && cols && dx_result_fetch_ids(r, ids.data(), 1) != DX_OK) { dx_result_free(r); return fail("fetch"); }
  // the terms of all columns in one call; constants projected by the statement are not dictionary terms
  std::vector<int64_t> real;
  for (uint64_t i = 0; i < rows * (uint64_t)cols; ++i)
    if (ids[i] >= 0) real.push_back(ids[i]);
  uint64_t total = 0;
  if (!real.empty() && dx_terms_bytes(ctx, real.data(), real.size(), &total) != DX_OK) { dx_result_free(r); return fail("decode"); }
  std::string bytes(total, '\0');
  std::vector<uint64_t> off(real.size() + 1, 0);
  if (!real.empty() && dx_terms_decode(ctx, real.data(), real.size(), &bytes[0], total, off.data()) != DX_OK) { dx_result_free(r); return fail("decode"); }
  std::vector<uint64_t> where(rows * (uint64_t)cols, 0);  // position of cell i among the real terms
  {
    uint64_t k = 0;
    for (uint64_t i = 0; i < rows * (uint64_t)cols; ++i)
      if (ids[i] >= 0) where[i] = k++;
  }
  mkdir((base + dataset).c_str(), 0777);
  mkdir(cold.c_str(), 0777);
  {
    std::ofstream out(cold + "/part-00000-dx-c000.csv", std::ios::binary);
    std::string row;
    for (uint64_t i = 0; i < rows; ++i) {
      row.clear();
      for (int c = 0; c < cols; ++c) {
        if (c) row += ',';
        const uint64_t cell = (uint64_t)c * rows + i;  // column-major
        if (ids[cell] >= 0) {
          csv_field(row, bytes.data() + off[where[cell]], off[where[cell] + 1] - off[where[cell]]);
        } else {
          const char* v = dx_result_virtual_term(r, ids[cell]);
          const std::string s = v ? v : "";
          csv_field(row, s.data(), s.size());
        }
      }
      row += '\n';
      out << row;
    }
    if (!out) { dx_result_free(r); return fail("write"); }
  }
  { std::ofstream ok(cold + "/_SUCCESS"); }
  const auto t2 = std::chrono::steady_clock::now();
  const long long us = std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count();
  std::string plist;
  for (size_t i = 0; i < partitions.size(); ++i) plist += (i ? "," : "") + partitions[i];
  std::ofstream res(result_dir + "results_" + std::to_string(tp) + ".txt", std::ios::app);  // :82-91
  res << "Query: " << name << "\n"
      << "Time: " << us / 1000 << "\n"
      << "Time_us: " << us << "\n"
      << "Result_count: " << rows << "\n"
      << "partitions: " << plist << "\n";
  dx_result_free(r);
  dx_free(ctx);
  return 0;
}
