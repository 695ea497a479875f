// Host-only driver of the library's Parquet reader (dx_parquet_read.cpp) for the CPU tests: reads the named string
// columns of one file and writes them out row by row, so that the test can compare them with what it wrote.
//   parquet_reader_harness <in.parquet> <out.txt> <column> [<column> ...]
// Output: one line per row and column, "<column index>\t<length>\t<bytes>\n" in the order the reader delivers them
// (column after column inside a row group); then a line "chunks <n> dictionaries <n> indexed <n> plain <n>".
#include <cstdio>
#include <string>
#include <vector>

#include "dx_state.h"

struct Dump : dx::ParquetStringSink {
  FILE* f;
  std::vector<std::vector<std::string>> dict;
  size_t chunks = 0, dicts = 0, indexed = 0, plain = 0;
  Dump(FILE* out, size_t ncols) : f(out), dict(ncols) {}
  void row(int col, const char* p, size_t len) {
    fprintf(f, "%d\t%zu\t", col, len);
    fwrite(p, 1, len, f);
    fputc('\n', f);
  }
  void dictionary(int col, const std::vector<std::pair<const uint8_t*, uint32_t>>& values) override {
    dict[col].clear();
    for (auto& v : values) dict[col].emplace_back(reinterpret_cast<const char*>(v.first), v.second);
    ++dicts;
  }
  void indices(int col, const uint32_t* idx, size_t n) override {
    for (size_t i = 0; i < n; ++i) row(col, dict[col][idx[i]].data(), dict[col][idx[i]].size());
    indexed += n;
  }
  void value(int col, const uint8_t* bytes, uint32_t len) override { row(col, reinterpret_cast<const char*>(bytes), len); ++plain; }
  void end_of_chunk(int) override { ++chunks; }
};

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::vector<std::string> cols(argv + 3, argv + argc);
  FILE* f = fopen(argv[2], "wb");
  if (!f) return 2;
  int rc = 0;
  try {
    Dump d(f, cols.size());
    dx::read_parquet_strings(argv[1], cols, d);
    fprintf(f, "chunks %zu dictionaries %zu indexed %zu plain %zu\n", d.chunks, d.dicts, d.indexed, d.plain);
  } catch (const dx::DxError& e) {
    fprintf(stderr, "%d %s\n", e.code, e.what());
    rc = 1;
  } catch (const std::exception& e) {  // the library's entry points turn these into error codes as well
    fprintf(stderr, "exception %s\n", e.what());
    rc = 1;
  }
  fclose(f);
  return rc;
}
