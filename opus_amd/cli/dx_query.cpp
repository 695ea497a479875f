// dx-query: the query processor CLI behind scripts/run_query.sh of the reference.
//
//   dx-query <dataset> <partitions> <base path> <translated queries dir>
//
// Same positional arguments, translated-file format and results file as org.ics.isl.QueryProcessor
// (query_processor/QueryProcessor.scala:30-110): for every file under the directory, load the vertical
// partition tables named on its `partitions` line, run the SQL once untimed if it is the first query of
// the process, time the count, append Query / Time / Time_load / Load_data / Result_count / partitions
// to results/<dataset>/<N>_bal/results_<TP>_bal.txt. Time_us is added next to Time (which is in whole
// milliseconds like the reference's) because GPU times are below one millisecond.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "diaeresis_hip.h"

static void list_files(const std::string& dir, std::vector<std::string>& out) {  // QueryProcessor.scala:353-361
  std::vector<std::string> files, dirs;
  DIR* d = opendir(dir.c_str());
  if (!d) return;
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n == "." || n == "..") continue;
    struct stat st;
    const std::string p = dir + "/" + n;
    if (stat(p.c_str(), &st) != 0) continue;
    (S_ISDIR(st.st_mode) ? dirs : files).push_back(p);
  }
  closedir(d);
  std::sort(files.begin(), files.end());
  std::sort(dirs.begin(), dirs.end());
  out.insert(out.end(), files.begin(), files.end());
  for (auto& s : dirs) list_files(s, out);
}

static std::vector<std::string> split(const std::string& s, char c) {
  std::vector<std::string> out;
  std::string cur;
  for (char ch : s) {
    if (ch == c) { out.push_back(cur); cur.clear(); } else cur += ch;
  }
  out.push_back(cur);
  return out;
}

int main(int argc, char** argv) {
  if (argc - 1 != 4) {  // QueryProcessor.scala:33-37
    printf("-->Missing Arguments\n");
    printf("-->Arguments: number of partitions, dataset name, hdfs base path, sql query path\n");
    return 255;
  }
  const std::string dataset = argv[1];
  const int n = atoi(argv[2]);
  std::string base = argv[3];
  const std::string input = argv[4];
  if (!base.empty() && base.back() != '/') base += '/';
  const std::string result_dir = "results/" + dataset + "/" + std::to_string(n) + "_bal/";
  mkdir("results", 0777);
  mkdir(("results/" + dataset).c_str(), 0777);
  mkdir(result_dir.c_str(), 0777);

  dx_ctx* ctx = dx_init(0);
  if (!ctx) { fprintf(stderr, "dx-query: no HIP device (this tool has no CPU path)\n"); return 1; }
  if (dx_open_dataset(ctx, base.c_str(), dataset.c_str(), n) != DX_OK) {
    fprintf(stderr, "dx-query: %s\n", dx_last_error(ctx));
    return 1;
  }
  std::vector<std::string> files;
  list_files(input, files);
  bool first_run = true;
  int failures = 0;
  for (const std::string& path : files) {
    const std::string name = path.substr(path.find_last_of('/') + 1);
    std::ifstream f(path);
    std::string line;
    std::vector<std::string> partitions;
    int tp = -1;
    std::vector<std::pair<std::string, std::string>> query_map;  // key -> statement, in the order the keys first appear
    while (std::getline(f, line)) {  // QueryProcessor.scala:73-88
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.rfind("TP", 0) == 0) tp = atoi(split(line, ' ')[1].c_str());
      else if (line.rfind("partitions", 0) == 0) { auto t = split(line, ' '); partitions = t.size() > 1 ? split(t[1], ',') : std::vector<std::string>(); }
      else if (line.rfind(">>>>>", 0) != 0) {
        const size_t sp = line.find(' ');
        const std::string key = line.substr(0, sp), text = sp == std::string::npos ? "" : line.substr(sp + 1);
        size_t at = 0;
        while (at < query_map.size() && query_map[at].first != key) ++at;
        if (at == query_map.size()) query_map.emplace_back(key, text);
        else query_map[at].second = text;
      }
    }
    printf("Query: %s\n", name.c_str());
    if (query_map.empty()) {
      fprintf(stderr, "dx-query: %s holds no statement\n", name.c_str());
      ++failures;
      continue;
    }
    // one statement, or several whose results are joined on the columns they share (executeFinalQuery, :184-208)
    std::vector<const char*> sqls;
    for (const auto& kv : query_map) sqls.push_back(kv.second.c_str());
    auto run = [&](dx_result** out) { return dx_query_joined(ctx, sqls.data(), (int)sqls.size(), DX_QUERY_COUNT, out); };
    // preloadTables (:210-236)
    uint64_t load_rows = 0;
    const auto l0 = std::chrono::steady_clock::now();
    bool ok = true;
    {
      std::vector<const char*> keys;
      for (const auto& p : partitions) keys.push_back(p.c_str());
      if (dx_load_tables(ctx, keys.data(), (int)keys.size(), &load_rows) != DX_OK) { fprintf(stderr, "dx-query: %s\n", dx_last_error(ctx)); ok = false; }
    }
    const auto load_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - l0).count();
    dx_result* r = nullptr;
    if (ok && first_run) {  // :161-170
      if (run(&r) == DX_OK) dx_result_free(r);
      first_run = false;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (ok && run(&r) != DX_OK) { fprintf(stderr, "dx-query: %s\n", dx_last_error(ctx)); ok = false; }
    const auto t2 = std::chrono::steady_clock::now();
    if (!ok) { ++failures; dx_unload_all(ctx); continue; }
    const long long us = std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count();
    std::ofstream out(result_dir + "results_" + std::to_string(tp) + "_bal.txt", std::ios::app);  // :90-106
    std::string plist;
    for (size_t i = 0; i < partitions.size(); ++i) plist += (i ? "," : "") + partitions[i];
    out << "Query: " << name << "\n"
        << "Time: " << us / 1000 << "\n"
        << "Time_us: " << us << "\n"
        << "Time_load: " << load_ms << "\n"
        << "Load_data: " << load_rows << "\n"
        << "Result_count: " << dx_result_count(r) << "\n"
        << "partitions: " << plist << "\n";
    printf("Time: %lld\n\n", us / 1000);
    dx_result_free(r);
    dx_unload_all(ctx);  // :180
  }
  dx_free(ctx);
  return failures ? 1 : 0;
}
