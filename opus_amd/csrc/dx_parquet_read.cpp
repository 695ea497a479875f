// A minimal Parquet reader for the reference's on-disk layout (host code, no GPU work).
//
// What spark.read.load(path).as[(String, String)] does for a vertical-partition directory
// (QueryProcessor.scala:322-333): every part-*.parquet file of the directory, its flat string columns by name.
// Handles what Spark's writer (parquet-mr) and this library's own writer produce for such tables: BYTE_ARRAY
// columns, REQUIRED or OPTIONAL (definition levels; a null is an error here, the tables hold none), data pages of
// the first and the second version, PLAIN and dictionary encodings (PLAIN_DICTIONARY / RLE_DICTIONARY, with the
// fall-back to PLAIN pages inside a chunk), UNCOMPRESSED and SNAPPY. Anything else is reported, not guessed at.
//
// The values reach the caller the way the file holds them -- a dictionary per column chunk and indices into it --
// so that a caller who maps strings to ids looks every distinct string of a chunk up once, not every row.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

#include "dx_state.h"

namespace dx {
namespace {

[[noreturn]] void bad(const std::string& path, const std::string& what) { throw DxError(DX_E_MALFORMED, "parquet: " + what + " in " + path); }

// ---- Thrift compact protocol, reading side ----------------------------------------------------------------
struct ThriftIn {
  const uint8_t* p;
  const uint8_t* end;
  const std::string& path;
  std::vector<int> last{0};
  ThriftIn(const uint8_t* b, const uint8_t* e, const std::string& pa) : p(b), end(e), path(pa) {}
  uint8_t byte() {
    if (p >= end) bad(path, "truncated metadata");
    return *p++;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      const uint8_t b = byte();
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return v;
    }
    bad(path, "overlong varint");
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  std::string binary() {
    const uint64_t n = varint();
    if (n > (uint64_t)(end - p)) bad(path, "truncated string");
    std::string s(reinterpret_cast<const char*>(p), n);
    p += n;
    return s;
  }
  // next field of the current struct: false at its end
  bool field(int& id, int& type) {
    const uint8_t b = byte();
    if (b == 0) return false;
    type = b & 0x0F;
    const int delta = b >> 4;
    id = delta ? last.back() + delta : (int)zigzag();
    last.back() = id;
    return true;
  }
  void begin_struct() { last.push_back(0); }
  void end_struct() { last.pop_back(); }
  void list_header(int& elem_type, uint64_t& n) {
    const uint8_t b = byte();
    elem_type = b & 0x0F;
    n = b >> 4;
    if (n == 15) n = varint();
  }
  void skip(int type) {
    switch (type) {
      case 1: case 2: return;                  // booleans live in the field header
      case 3: (void)byte(); return;
      case 4: case 5: case 6: (void)varint(); return;
      case 7: if (end - p < 8) bad(path, "truncated double"); p += 8; return;
      case 8: { const uint64_t n = varint(); if (n > (uint64_t)(end - p)) bad(path, "truncated binary"); p += n; return; }
      case 9: case 10: {
        int et; uint64_t n;
        list_header(et, n);
        for (uint64_t i = 0; i < n; ++i) {
          if (et == 1 || et == 2) (void)byte();  // a list's booleans take a byte each
          else skip(et);
        }
        return;
      }
      case 11: {
        const uint64_t n = varint();
        if (n) {
          const uint8_t kv = byte();
          for (uint64_t i = 0; i < n; ++i) { skip(kv >> 4); skip(kv & 0x0F); }
        }
        return;
      }
      case 12: {
        begin_struct();
        int id, t;
        while (field(id, t)) skip(t);
        end_struct();
        return;
      }
      default: bad(path, "unknown thrift type " + std::to_string(type));
    }
  }
};

constexpr int kByteArray = 6;
constexpr int kPlain = 0, kPlainDictionary = 2, kRleDictionary = 8;
constexpr int kUncompressed = 0, kSnappy = 1;
constexpr int kDataPage = 0, kDictionaryPage = 2, kDataPageV2 = 3;

struct SchemaLeaf { std::string name; int type = -1; int repetition = 0; };
struct ChunkInfo {
  int type = -1, codec = 0;
  std::vector<std::string> path_in_schema;
  int64_t num_values = 0, data_page_offset = -1, dictionary_page_offset = -1, total_compressed = 0;
};
struct RowGroupInfo { std::vector<ChunkInfo> cols; int64_t rows = 0; };
struct FileMeta { std::vector<SchemaLeaf> leaves; std::vector<RowGroupInfo> groups; int64_t rows = 0; };

ChunkInfo read_column_meta(ThriftIn& t) {
  ChunkInfo c;
  t.begin_struct();
  int id, ty;
  while (t.field(id, ty)) {
    if (id == 1 && ty == 5) c.type = (int)t.zigzag();
    else if (id == 3 && ty == 9) {
      int et; uint64_t n;
      t.list_header(et, n);
      for (uint64_t i = 0; i < n; ++i) c.path_in_schema.push_back(t.binary());
    } else if (id == 4 && ty == 5) c.codec = (int)t.zigzag();
    else if (id == 5 && ty == 6) c.num_values = t.zigzag();
    else if (id == 7 && ty == 6) c.total_compressed = t.zigzag();
    else if (id == 9 && ty == 6) c.data_page_offset = t.zigzag();
    else if (id == 11 && ty == 6) c.dictionary_page_offset = t.zigzag();
    else t.skip(ty);
  }
  t.end_struct();
  return c;
}

FileMeta read_footer(const uint8_t* file, size_t size, const std::string& path) {
  if (size < 12 || memcmp(file, "PAR1", 4) != 0 || memcmp(file + size - 4, "PAR1", 4) != 0) bad(path, "not a Parquet file");
  uint32_t flen;
  memcpy(&flen, file + size - 8, 4);
  if ((uint64_t)flen + 12 > size) bad(path, "footer length beyond the file");
  ThriftIn t(file + size - 8 - flen, file + size - 8, path);
  FileMeta m;
  int id, ty;
  while (t.field(id, ty)) {
    if (id == 2 && ty == 9) {  // schema: the root, then its children in order (flat tables only)
      int et; uint64_t n;
      t.list_header(et, n);
      for (uint64_t i = 0; i < n; ++i) {
        SchemaLeaf leaf;
        int children = 0;
        t.begin_struct();
        int fid, fty;
        while (t.field(fid, fty)) {
          if (fid == 1 && fty == 5) leaf.type = (int)t.zigzag();
          else if (fid == 3 && fty == 5) leaf.repetition = (int)t.zigzag();
          else if (fid == 4 && fty == 8) leaf.name = t.binary();
          else if (fid == 5 && fty == 5) children = (int)t.zigzag();
          else t.skip(fty);
        }
        t.end_struct();
        if (i == 0) continue;  // the root
        if (children) bad(path, "nested column " + leaf.name);
        if (leaf.repetition == 2) bad(path, "repeated column " + leaf.name);
        m.leaves.push_back(leaf);
      }
    } else if (id == 3 && ty == 6) {
      m.rows = t.zigzag();
    } else if (id == 4 && ty == 9) {
      int et; uint64_t n;
      t.list_header(et, n);
      for (uint64_t g = 0; g < n; ++g) {
        RowGroupInfo rg;
        t.begin_struct();
        int gid, gty;
        while (t.field(gid, gty)) {
          if (gid == 1 && gty == 9) {
            int cet; uint64_t cn;
            t.list_header(cet, cn);
            for (uint64_t c = 0; c < cn; ++c) {
              ChunkInfo ci;
              t.begin_struct();
              int cid, cty;
              while (t.field(cid, cty)) {
                if (cid == 3 && cty == 12) ci = read_column_meta(t);
                else t.skip(cty);
              }
              t.end_struct();
              rg.cols.push_back(std::move(ci));
            }
          } else if (gid == 3 && gty == 6) {
            rg.rows = t.zigzag();
          } else {
            t.skip(gty);
          }
        }
        t.end_struct();
        m.groups.push_back(std::move(rg));
      }
    } else {
      t.skip(ty);
    }
  }
  return m;
}

struct PageHeader {
  int type = -1, encoding = 0;
  int32_t uncompressed = 0, compressed = 0, num_values = 0, num_nulls = 0;
  int32_t def_bytes = 0, rep_bytes = 0;
  bool v2_compressed = true;
};
PageHeader read_page_header(ThriftIn& t) {
  PageHeader h;
  int id, ty;
  while (t.field(id, ty)) {
    if (id == 1 && ty == 5) h.type = (int)t.zigzag();
    else if (id == 2 && ty == 5) h.uncompressed = (int32_t)t.zigzag();
    else if (id == 3 && ty == 5) h.compressed = (int32_t)t.zigzag();
    else if ((id == 5 || id == 7 || id == 8) && ty == 12) {
      const int which = id;
      t.begin_struct();
      int fid, fty;
      while (t.field(fid, fty)) {
        if (fid == 1 && fty == 5) h.num_values = (int32_t)t.zigzag();
        else if (which == 5 && fid == 2 && fty == 5) h.encoding = (int)t.zigzag();
        else if (which == 7 && fid == 2 && fty == 5) h.encoding = (int)t.zigzag();
        else if (which == 8 && fid == 2 && fty == 5) h.num_nulls = (int32_t)t.zigzag();
        else if (which == 8 && fid == 4 && fty == 5) h.encoding = (int)t.zigzag();
        else if (which == 8 && fid == 5 && fty == 5) h.def_bytes = (int32_t)t.zigzag();
        else if (which == 8 && fid == 6 && fty == 5) h.rep_bytes = (int32_t)t.zigzag();
        else if (which == 8 && fid == 7 && (fty == 1 || fty == 2)) h.v2_compressed = fty == 1;
        else t.skip(fty);
      }
      t.end_struct();
    } else {
      t.skip(ty);
    }
  }
  return h;
}

// ---- Snappy (raw format): a varint length, then literals and copies ------------------------------------------
void snappy_uncompress(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect, const std::string& path) {
  const uint8_t* p = src;
  const uint8_t* end = src + n;
  uint64_t len = 0;
  for (int shift = 0;; shift += 7) {
    if (p >= end || shift > 35) bad(path, "snappy: bad length");
    const uint8_t b = *p++;
    len |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) break;
  }
  if (len != expect) bad(path, "snappy: page size differs from its header");
  out.resize(len);
  size_t at = 0;
  while (p < end) {
    const uint8_t tag = *p++;
    if ((tag & 3) == 0) {  // literal
      size_t l = tag >> 2;
      if (l >= 60) {
        const int extra = (int)l - 59;
        if (end - p < extra) bad(path, "snappy: truncated literal length");
        l = 0;
        for (int i = 0; i < extra; ++i) l |= (size_t)p[i] << (8 * i);
        p += extra;
      }
      ++l;
      if ((size_t)(end - p) < l || at + l > len) bad(path, "snappy: literal runs past the end");
      memcpy(out.data() + at, p, l);
      p += l;
      at += l;
    } else {
      size_t l, off;
      if ((tag & 3) == 1) {
        if (p >= end) bad(path, "snappy: truncated copy");
        l = 4 + ((tag >> 2) & 7);
        off = ((size_t)(tag >> 5) << 8) | *p++;
      } else if ((tag & 3) == 2) {
        if (end - p < 2) bad(path, "snappy: truncated copy");
        l = 1 + (tag >> 2);
        off = (size_t)p[0] | ((size_t)p[1] << 8);
        p += 2;
      } else {
        if (end - p < 4) bad(path, "snappy: truncated copy");
        l = 1 + (tag >> 2);
        off = (size_t)p[0] | ((size_t)p[1] << 8) | ((size_t)p[2] << 16) | ((size_t)p[3] << 24);
        p += 4;
      }
      if (off == 0 || off > at || at + l > len) bad(path, "snappy: copy outside the output");
      for (size_t i = 0; i < l; ++i) out[at + i] = out[at + i - off];  // may overlap itself: byte by byte
      at += l;
    }
  }
  if (at != len) bad(path, "snappy: output shorter than announced");
}

// ---- RLE / bit-packed hybrid ------------------------------------------------------------------------------------
// Decodes `count` values of `width` bits from [p, end) into out (appending).
void hybrid_decode(const uint8_t* p, const uint8_t* end, int width, size_t count, std::vector<uint32_t>& out, const std::string& path) {
  if (width < 0 || width > 32) bad(path, "bit width " + std::to_string(width));
  const size_t start = out.size();
  out.resize(start + count);
  uint32_t* dst = out.data() + start;
  size_t done = 0;
  if (width == 0) { std::fill(dst, dst + count, 0u); return; }
  const int vbytes = (width + 7) / 8;
  while (done < count) {
    uint64_t header = 0;
    for (int shift = 0;; shift += 7) {
      if (p >= end || shift > 35) bad(path, "truncated run header");
      const uint8_t b = *p++;
      header |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) break;
    }
    if (header & 1) {  // bit-packed: groups of eight values; a page's last group may be cut short by its writer
      const uint64_t groups = header >> 1;
      const uint64_t avail = (uint64_t)(end - p);
      const uint64_t n = std::min<uint64_t>(groups * 8, count - done);
      if ((n * (uint64_t)width + 7) / 8 > avail) bad(path, "truncated bit-packed run");
      const uint64_t mask = width == 32 ? 0xFFFFFFFFull : ((1ull << width) - 1);
      uint64_t bit = 0;
      for (uint64_t i = 0; i < n; ++i, bit += width) {
        const uint64_t byte = bit >> 3;
        uint64_t word = 0;
        memcpy(&word, p + byte, std::min<uint64_t>(8, avail - byte));
        dst[done + i] = (uint32_t)((word >> (bit & 7)) & mask);
      }
      done += n;
      p += std::min<uint64_t>(groups * (uint64_t)width, avail);
    } else {  // run of one value
      const uint64_t run = header >> 1;
      if (end - p < vbytes) bad(path, "truncated run value");
      uint32_t v = 0;
      memcpy(&v, p, vbytes);
      p += vbytes;
      const uint64_t n = std::min<uint64_t>(run, count - done);
      std::fill(dst + done, dst + done + n, v);
      done += n;
      if (run == 0) bad(path, "empty run");
    }
  }
}

struct Mapped {
  const uint8_t* p = nullptr;
  size_t n = 0;
  explicit Mapped(const std::string& path) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw DxError(DX_E_NOENT, "cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw DxError(DX_E_IO, "cannot stat " + path); }
    n = (size_t)st.st_size;
    if (n) {
      void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); throw DxError(DX_E_IO, "cannot map " + path); }
      p = static_cast<const uint8_t*>(m);
    }
    close(fd);
  }
  ~Mapped() { if (p) munmap(const_cast<uint8_t*>(p), n); }
  Mapped(const Mapped&) = delete;
  Mapped& operator=(const Mapped&) = delete;
};

// length-prefixed byte arrays (PLAIN): calls f(bytes, len) for `count` of them
template <typename F>
void plain_byte_arrays(const uint8_t* p, const uint8_t* end, size_t count, const std::string& path, F f) {
  for (size_t i = 0; i < count; ++i) {
    if (end - p < 4) bad(path, "truncated value length");
    uint32_t len;
    memcpy(&len, p, 4);
    p += 4;
    if ((size_t)(end - p) < len) bad(path, "value runs past its page");
    f(p, len);
    p += len;
  }
}

}  // namespace

void read_parquet_strings(const std::string& path, const std::vector<std::string>& columns, ParquetStringSink& sink) {
  Mapped file(path);
  const FileMeta meta = read_footer(file.p, file.n, path);
  std::vector<int> leaf_of(columns.size(), -1);
  for (size_t c = 0; c < columns.size(); ++c) {
    for (size_t l = 0; l < meta.leaves.size(); ++l)
      if (meta.leaves[l].name == columns[c]) leaf_of[c] = (int)l;
    if (leaf_of[c] < 0) bad(path, "no column " + columns[c]);
    if (meta.leaves[leaf_of[c]].type != kByteArray) bad(path, "column " + columns[c] + " is not a string column");
  }
  std::vector<uint8_t> page;
  std::vector<uint32_t> levels, indices;
  std::vector<std::pair<const uint8_t*, uint32_t>> dict;
  std::vector<uint8_t> dict_bytes;  // a decompressed dictionary page has to outlive the chunk's data pages
  for (const RowGroupInfo& rg : meta.groups) {
    if (rg.cols.size() != meta.leaves.size()) bad(path, "row group with a different number of columns than the schema");
    for (size_t c = 0; c < columns.size(); ++c) {
      const ChunkInfo& ch = rg.cols[leaf_of[c]];
      const bool optional = meta.leaves[leaf_of[c]].repetition == 1;
      if (ch.type != kByteArray) bad(path, "column chunk of another type than its schema element");
      if (ch.codec != kUncompressed && ch.codec != kSnappy) bad(path, "compression codec " + std::to_string(ch.codec) + " (only uncompressed and snappy are read)");
      if (ch.num_values <= 0) { sink.end_of_chunk((int)c); continue; }  // a file without rows still has a row group
      int64_t at = ch.data_page_offset;
      if (ch.dictionary_page_offset > 0 && ch.dictionary_page_offset < at) at = ch.dictionary_page_offset;
      if (at < 4 || (uint64_t)at >= file.n) bad(path, "column chunk outside the file");
      int64_t left = ch.num_values;
      dict.clear();
      bool have_dict = false;
      while (left > 0) {
        ThriftIn t(file.p + at, file.p + file.n, path);
        const PageHeader h = read_page_header(t);
        const uint8_t* body = t.p;
        if (h.compressed < 0 || h.uncompressed < 0 || (uint64_t)(body - file.p) + (uint64_t)h.compressed > file.n) bad(path, "page outside the file");
        at = (body - file.p) + h.compressed;
        if (h.type == kDictionaryPage) {
          if (h.encoding != kPlain && h.encoding != kPlainDictionary) bad(path, "dictionary page encoding " + std::to_string(h.encoding));
          const uint8_t* b = body;
          const uint8_t* e = body + h.compressed;
          if (ch.codec == kSnappy) {
            snappy_uncompress(body, h.compressed, dict_bytes, h.uncompressed, path);
            b = dict_bytes.data();
            e = b + dict_bytes.size();
          }
          dict.clear();
          if (h.num_values < 0 || (int64_t)h.num_values * 4 > e - b) bad(path, "dictionary page with " + std::to_string(h.num_values) + " values");
          plain_byte_arrays(b, e, h.num_values, path, [&](const uint8_t* p, uint32_t len) { dict.emplace_back(p, len); });
          have_dict = true;
          sink.dictionary((int)c, dict);
          continue;
        }
        if (h.type != kDataPage && h.type != kDataPageV2) continue;  // index pages and the like
        // a page of parquet-mr holds some ten thousand values, one of this library's writer 2^20: a damaged header must not size buffers
        if (h.num_values < 0 || h.num_values > (1 << 26) || h.num_values > left) bad(path, "page with " + std::to_string(h.num_values) + " values");
        const uint8_t* vals;
        const uint8_t* vend;
        size_t present = (size_t)h.num_values;
        if (h.type == kDataPage) {
          const uint8_t* b = body;
          const uint8_t* e = body + h.compressed;
          if (ch.codec == kSnappy) {
            snappy_uncompress(body, h.compressed, page, h.uncompressed, path);
            b = page.data();
            e = b + page.size();
          }
          if (optional) {  // definition levels: 4-byte length, then the hybrid encoding at bit width 1
            if (e - b < 4) bad(path, "truncated definition levels");
            uint32_t n;
            memcpy(&n, b, 4);
            b += 4;
            if ((size_t)(e - b) < n) bad(path, "definition levels run past their page");
            levels.clear();
            hybrid_decode(b, b + n, 1, h.num_values, levels, path);
            for (uint32_t v : levels)
              if (!v) bad(path, "null in column " + columns[c] + " (the tables of the layout hold none)");
            b += n;
          }
          vals = b;
          vend = e;
        } else {
          if (h.rep_bytes) bad(path, "repetition levels in a flat column");
          if (h.num_nulls) bad(path, "null in column " + columns[c] + " (the tables of the layout hold none)");
          const uint8_t* b = body + h.def_bytes;  // levels are never compressed in a page of the second version
          const int32_t data_c = h.compressed - h.def_bytes, data_u = h.uncompressed - h.def_bytes;
          if (h.def_bytes < 0 || data_c < 0 || data_u < 0) bad(path, "definition levels longer than their page");
          const uint8_t* e = b + data_c;
          if (ch.codec == kSnappy && h.v2_compressed) {
            snappy_uncompress(b, data_c, page, data_u, path);
            b = page.data();
            e = b + page.size();
          }
          vals = b;
          vend = e;
        }
        if (h.encoding == kPlain) {
          plain_byte_arrays(vals, vend, present, path, [&](const uint8_t* p, uint32_t len) { sink.value((int)c, p, len); });
        } else if (h.encoding == kPlainDictionary || h.encoding == kRleDictionary) {
          if (!have_dict) bad(path, "dictionary-encoded page without a dictionary page");
          if (vend - vals < 1) bad(path, "truncated index page");
          indices.clear();
          hybrid_decode(vals + 1, vend, vals[0], present, indices, path);
          for (uint32_t i : indices)
            if (i >= dict.size()) bad(path, "dictionary index out of range");
          sink.indices((int)c, indices.data(), indices.size());
        } else {
          bad(path, "value encoding " + std::to_string(h.encoding) + " (only PLAIN and dictionary encodings are read)");
        }
        left -= h.num_values;
      }
      sink.end_of_chunk((int)c);
    }
  }
}

}  // namespace dx
