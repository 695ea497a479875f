// Host-only driver of the library's Parquet writer (dx_parquet.cpp) for the CPU tests: writes one file whose
// first two columns are ids over a small pool of terms and whose third column is given as strings.
//   parquet_writer_harness <out.parquet> <rows> <distinct terms>
// Row i holds (term[i % T], term[(7 * i + 3) % T], "row<i>"); term k is "<t" + k + ">" + k times 'x' (so lengths
// vary, the empty-ish and the long one included), term 1 is non-ASCII.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dx_state.h"

int main(int argc, char** argv) {
  if (argc != 4) return 2;
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
