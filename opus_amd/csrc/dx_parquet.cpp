// A minimal Parquet writer for the reference's on-disk layout (host code, no GPU work).
//
// Everything the Scala tools write with DataFrame.write.parquet is a flat table of string columns
// (partitioner/DAP.scala:364-367 `_1,_2,_3`; :433-437 `_1,_2`; :405-412 `uri,uriType`; :496-499 `uri,parts`;
// case classes in utils/Types.scala:7-16). This writer produces such files so that spark.read.load (Loader.scala:19-38,
// QueryProcessor.scala:308-339) and any other Parquet reader accept them: format version 1, one row group per
// kRowGroupRows rows, REQUIRED BYTE_ARRAY columns annotated UTF8, per column chunk one dictionary page (PLAIN) and
// one data page of bit-packed dictionary indices (the PLAIN_DICTIONARY encoding of parquet-mr's first-version
// pages), Thrift compact footer. Terms repeat heavily inside a column, so the dictionary pages keep the files near
// the size of Spark's own. Pages are snappy streams, Spark's default codec, unless DX_PARQUET_CODEC=uncompressed
// (a third of the bytes on disk for LUBM's tables, and the encoding takes about twice as long).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dx_state.h"

namespace dx {
namespace {

// ---- Thrift compact protocol, as much of it as the footer and the page headers need ----------------------
struct Thrift {
  std::string out;
  std::vector<int> last;  // last field id of every open struct
  Thrift() { last.push_back(0); }
  void varint(uint64_t v) {
    while (v >= 0x80) { out.push_back((char)(v | 0x80)); v >>= 7; }
    out.push_back((char)v);
  }
  static uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
  void field(int id, int type) {
    const int delta = id - last.back();
    if (delta > 0 && delta <= 15) out.push_back((char)((delta << 4) | type));
    else { out.push_back((char)type); varint(zigzag(id)); }
    last.back() = id;
  }
  void i32(int id, int32_t v) { field(id, 5); varint(zigzag(v)); }
  void i64(int id, int64_t v) { field(id, 6); varint(zigzag(v)); }
  void str(int id, const std::string& s) { field(id, 8); varint(s.size()); out += s; }
  void begin_struct(int id) { field(id, 12); last.push_back(0); }
  void end_struct() { out.push_back(0); last.pop_back(); }
  void list(int id, int elem_type, size_t n) {
    field(id, 9);
    if (n < 15) out.push_back((char)((n << 4) | elem_type));
    else { out.push_back((char)(0xF0 | elem_type)); varint(n); }
  }
  // list elements carry no field header
  void elem_struct_begin() { last.push_back(0); }
  void elem_i32(int32_t v) { varint(zigzag(v)); }
  void elem_str(const std::string& s) { varint(s.size()); out += s; }
};

constexpr int kByteArray = 6, kRequired = 0, kUtf8 = 0, kPlain = 0, kPlainDictionary = 2, kRle = 3, kUncompressed = 0, kSnappy = 1;
constexpr int kDataPage = 0, kDictionaryPage = 2;
constexpr uint64_t kRowGroupRows = kParquetRowGroupRows;

struct ChunkMeta {
  int64_t dict_offset, data_offset, total_size, total_uncompressed, num_values;
};

void put_u32(std::string& s, uint32_t v) { s.append(reinterpret_cast<const char*>(&v), 4); }

// ---- Snappy, raw format: the uncompressed length, then literals and copies found with a 4-byte hash inside 64 KiB blocks ----
inline uint32_t load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
void snappy_literal(std::string& out, const uint8_t* p, size_t len) {
  if (!len) return;
  const size_t n = len - 1;
  if (n < 60) {
    out.push_back((char)(n << 2));
  } else {
    const int bytes = n < (1u << 8) ? 1 : n < (1u << 16) ? 2 : n < (1u << 24) ? 3 : 4;
    out.push_back((char)((59 + bytes) << 2));
    for (int i = 0; i < bytes; ++i) out.push_back((char)(n >> (8 * i)));
  }
  out.append(reinterpret_cast<const char*>(p), len);
}
void snappy_copy(std::string& out, size_t offset, size_t len) {  // len >= 4, offset < 65536
  auto two = [&](size_t l) { out.push_back((char)(2 | ((l - 1) << 2))); out.push_back((char)offset); out.push_back((char)(offset >> 8)); };
  while (len >= 68) { two(64); len -= 64; }
  if (len > 64) { two(60); len -= 60; }
  if (len < 12 && offset < 2048) { out.push_back((char)(1 | ((len - 4) << 2) | ((offset >> 8) << 5))); out.push_back((char)offset); }
  else two(len);
}
void snappy_compress(const std::string& in, std::string& out) {
  out.clear();
  for (uint64_t v = in.size();;) {  // the length as a varint
    if (v >= 0x80) { out.push_back((char)(v | 0x80)); v >>= 7; } else { out.push_back((char)v); break; }
  }
  constexpr size_t kBlock = 1 << 16;
  std::vector<uint16_t> table(1 << 14);
  const uint8_t* base = reinterpret_cast<const uint8_t*>(in.data());
  for (size_t b0 = 0; b0 < in.size(); b0 += kBlock) {
    const uint8_t* p = base + b0;
    const size_t len = std::min(kBlock, in.size() - b0);
    std::fill(table.begin(), table.end(), (uint16_t)0);
    size_t ip = 0, lit = 0;
    uint32_t skip = 32;  // after 32 positions without a match the step grows: bit-packed indices do not compress
    while (ip + 4 <= len) {
      const uint32_t w = load32(p + ip);
      const uint32_t h = (w * 0x1E35A7BDu) >> 18;
      const size_t cand = table[h];
      table[h] = (uint16_t)ip;
      if (cand < ip && load32(p + cand) == w) {
        snappy_literal(out, p + lit, ip - lit);
        size_t m = 4;
        while (ip + m < len && p[cand + m] == p[ip + m]) ++m;
        snappy_copy(out, ip - cand, m);
        ip += m;
        lit = ip;
        skip = 32;
      } else {
        ip += skip++ >> 5;
      }
    }
    snappy_literal(out, p + lit, len - lit);
  }
}

bool snappy_pages() {  // Spark's default codec is this writer's too; DX_PARQUET_CODEC=uncompressed leaves the pages as they are
  const char* e = getenv("DX_PARQUET_CODEC");
  return !(e && (std::string(e) == "uncompressed" || std::string(e) == "none"));
}

std::string page_header(int type, size_t size, size_t compressed_size, uint64_t num_values, bool dictionary_encoded) {
  Thrift t;
  t.i32(1, type);
  t.i32(2, (int32_t)size);
  t.i32(3, (int32_t)compressed_size);
  if (type == kDataPage) {
    t.begin_struct(5);
    t.i32(1, (int32_t)num_values);
    t.i32(2, dictionary_encoded ? kPlainDictionary : kPlain);
    t.i32(3, kRle);
    t.i32(4, kRle);
    t.end_struct();
  } else {
    t.begin_struct(7);
    t.i32(1, (int32_t)num_values);
    t.i32(2, kPlainDictionary);
    t.end_struct();
  }
  t.out.push_back(0);  // end of PageHeader
  return t.out;
}

}  // namespace

bool parquet_writer_uses_snappy() { return snappy_pages(); }

struct ParquetWriter::Impl {
  FILE* f = nullptr;
  std::string path;
  std::vector<std::string> names;
  uint64_t at = 0, rows = 0;
  struct Group { uint64_t rows; std::vector<ChunkMeta> cols; };
  std::vector<Group> groups;
  // id -> index in the current chunk's dictionary, valid where stamp[id] == epoch
  std::vector<uint32_t> local, stamp;
  uint32_t epoch = 0;
  void write(const std::string& s) {
    if (s.empty()) return;
    if (fwrite(s.data(), 1, s.size(), f) != s.size()) throw DxError(DX_E_IO, "write failed: " + path);
    at += s.size();
  }
  bool snappy = snappy_pages();
  std::string packed;
  // one page: header, then the body as the file's codec wants it; returns the bytes the page would take uncompressed
  int64_t write_page(int type, const std::string& body, uint64_t num_values, bool dictionary_encoded) {
    const std::string* out = &body;
    if (snappy) { snappy_compress(body, packed); out = &packed; }
    if (out->size() >= (1ull << 31)) throw DxError(DX_E_RANGE, "page of 2 GiB or more in " + path);
    const std::string h = page_header(type, body.size(), out->size(), num_values, dictionary_encoded);
    write(h);
    write(*out);
    return (int64_t)(h.size() + body.size());
  }
};

ParquetWriter::ParquetWriter(const std::string& path, const std::vector<std::string>& column_names) : p(new Impl) {
  p->path = path;
  p->names = column_names;
  p->f = fopen(path.c_str(), "wb");
  if (!p->f) { delete p; p = nullptr; throw DxError(DX_E_IO, "cannot create " + path); }
  p->write("PAR1");
}

ParquetWriter::~ParquetWriter() {
  if (p) {
    if (p->f) fclose(p->f);
    delete p;
  }
}

// One chunk of a column whose values are given as term ids: dictionary page over the ids that occur, then indices.
static ChunkMeta write_id_chunk(ParquetWriter::Impl* p, const int64_t* ids, uint64_t n, const uint8_t* pool, const uint64_t* pool_off,
                                uint64_t n_terms) {
  if (p->local.size() < n_terms) { p->local.assign(n_terms, 0); p->stamp.assign(n_terms, 0); p->epoch = 0; }
  if (++p->epoch == 0) { std::fill(p->stamp.begin(), p->stamp.end(), 0); p->epoch = 1; }
  std::string dict;
  std::vector<uint32_t> idx(n);
  uint32_t n_dict = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t id = (uint64_t)ids[i];
    if (id >= n_terms) throw DxError(DX_E_RANGE, "term id outside the dictionary while writing " + p->path);
    if (p->stamp[id] != p->epoch) {
      p->stamp[id] = p->epoch;
      p->local[id] = n_dict++;
      const uint64_t b = pool_off[id], len = pool_off[id + 1] - b;
      put_u32(dict, (uint32_t)len);
      dict.append(reinterpret_cast<const char*>(pool + b), len);
    }
    idx[i] = p->local[id];
  }
  if (dict.size() >= (1ull << 31)) throw DxError(DX_E_RANGE, "dictionary page of 2 GiB or more in " + p->path);
  int width = 0;
  while ((1ull << width) < n_dict) ++width;
  // data page: bit width, then one bit-packed run of ceil(n / 8) groups of eight values, least significant bit first
  const uint64_t groups = (n + 7) / 8;
  std::string data;
  data.push_back((char)width);
  {
    Thrift v;
    v.varint((groups << 1) | 1);
    data += v.out;
  }
  const size_t base = data.size();
  data.resize(base + groups * (uint64_t)width, 0);
  if (width) {
    uint8_t* d = reinterpret_cast<uint8_t*>(&data[base]);
    uint64_t bit = 0;
    for (uint64_t i = 0; i < n; ++i, bit += width) {
      const uint64_t v = idx[i];
      uint64_t byte = bit >> 3;
      const int sh = (int)(bit & 7);
      uint64_t word = v << sh;  // width <= 32, sh <= 7: fits
      for (int k = 0; k < 5 && word; ++k, word >>= 8) d[byte + k] |= (uint8_t)word;
    }
  }
  ChunkMeta m{};
  m.num_values = (int64_t)n;
  m.dict_offset = (int64_t)p->at;
  m.total_uncompressed = p->write_page(kDictionaryPage, dict, n_dict, true);
  m.data_offset = (int64_t)p->at;
  m.total_uncompressed += p->write_page(kDataPage, data, n, true);
  m.total_size = (int64_t)p->at - m.dict_offset;
  return m;
}

// One chunk of a column given as strings: a PLAIN data page.
static ChunkMeta write_string_chunk(ParquetWriter::Impl* p, const std::string* vals, uint64_t n) {
  std::string data;
  for (uint64_t i = 0; i < n; ++i) { put_u32(data, (uint32_t)vals[i].size()); data += vals[i]; }
  if (data.size() >= (1ull << 31)) throw DxError(DX_E_RANGE, "data page of 2 GiB or more in " + p->path);
  ChunkMeta m{};
  m.num_values = (int64_t)n;
  m.dict_offset = -1;
  m.data_offset = (int64_t)p->at;
  m.total_uncompressed = p->write_page(kDataPage, data, n, false);
  m.total_size = (int64_t)p->at - m.data_offset;
  return m;
}

void ParquetWriter::append(uint64_t n, const std::vector<Column>& cols, const uint8_t* pool, const uint64_t* pool_off, uint64_t n_terms) {
  if (cols.size() != p->names.size()) throw DxError(DX_E_INVAL, "parquet: column count differs from the schema");
  for (uint64_t b = 0; b < n; b += kRowGroupRows) {
    const uint64_t m = std::min<uint64_t>(kRowGroupRows, n - b);
    Impl::Group g;
    g.rows = m;
    for (const Column& c : cols)
      g.cols.push_back(c.ids ? write_id_chunk(p, c.ids + b, m, pool, pool_off, n_terms) : write_string_chunk(p, c.strings + b, m));
    p->groups.push_back(std::move(g));
  }
  p->rows += n;
}

// The file's footer: FileMetaData in the compact protocol, its length, the magic.
struct FooterGroup { uint64_t rows; std::vector<ChunkMeta> cols; };
static std::string footer_bytes(const std::vector<std::string>& names, uint64_t rows, const std::vector<FooterGroup>& groups, bool snappy) {
  Thrift t;
  t.i32(1, 1);  // version
  t.list(2, 12, names.size() + 1);
  t.elem_struct_begin();  // root
  t.str(4, "spark_schema");
  t.i32(5, (int32_t)names.size());
  t.end_struct();
  for (const std::string& nm : names) {
    t.elem_struct_begin();
    t.i32(1, kByteArray);
    t.i32(3, kRequired);
    t.str(4, nm);
    t.i32(6, kUtf8);
    t.end_struct();
  }
  t.i64(3, (int64_t)rows);
  t.list(4, 12, groups.size());
  for (const auto& g : groups) {
    t.elem_struct_begin();
    t.list(1, 12, g.cols.size());
    int64_t total = 0;
    for (size_t c = 0; c < g.cols.size(); ++c) {
      const ChunkMeta& m = g.cols[c];
      total += m.total_uncompressed;
      t.elem_struct_begin();  // ColumnChunk
      t.i64(2, m.dict_offset >= 0 ? m.dict_offset : m.data_offset);
      t.begin_struct(3);      // ColumnMetaData
      t.i32(1, kByteArray);
      if (m.dict_offset >= 0) { t.list(2, 5, 2); t.elem_i32(kPlainDictionary); t.elem_i32(kRle); }
      else { t.list(2, 5, 2); t.elem_i32(kPlain); t.elem_i32(kRle); }
      t.list(3, 8, 1);
      t.elem_str(names[c]);
      t.i32(4, snappy ? kSnappy : kUncompressed);
      t.i64(5, m.num_values);
      t.i64(6, m.total_uncompressed);
      t.i64(7, m.total_size);
      t.i64(9, m.data_offset);
      if (m.dict_offset >= 0) t.i64(11, m.dict_offset);
      t.end_struct();
      t.end_struct();
    }
    t.i64(2, total);  // total_byte_size: the uncompressed bytes of the row group's column data
    t.i64(3, (int64_t)g.rows);
    t.end_struct();
  }
  t.str(6, snappy ? "diaeresis-hip (parquet format 1, plain dictionary pages, snappy)" : "diaeresis-hip (parquet format 1, plain dictionary pages, uncompressed)");
  t.out.push_back(0);
  std::string out = t.out;
  put_u32(out, (uint32_t)t.out.size());
  out += "PAR1";
  return out;
}

void ParquetWriter::close() {
  std::vector<FooterGroup> groups;
  for (auto& g : p->groups) groups.push_back({g.rows, g.cols});
  p->write(footer_bytes(p->names, p->rows, groups, p->snappy));
  if (fclose(p->f) != 0) { p->f = nullptr; throw DxError(DX_E_IO, "close failed: " + p->path); }
  p->f = nullptr;
}

// ---- a file whose page bodies are made elsewhere ---------------------------------------------------------------
// The same file format as ParquetWriter's, uncompressed, for a caller that has the page bodies encoded on the GPU: from
// the sizes of every column chunk's pages this lays the file out, writes everything that is not a page body (magic, page
// headers, the bit width and run header in front of the packed indices, footer) and says where each body belongs.
// A snappy stream begins with the length of what it unpacks to; a body that is sent as it is becomes one literal.
static std::string snappy_preamble(uint64_t unpacked) {
  Thrift v;
  v.varint(unpacked);
  return v.out;
}
static std::string snappy_literal_tag(uint64_t len) {  // the element header of a literal of len bytes (none for an empty one)
  std::string out;
  if (!len) return out;
  const uint64_t n = len - 1;
  if (n < 60) {
    out.push_back((char)(n << 2));
  } else {
    const int bytes = n < (1u << 8) ? 1 : n < (1u << 16) ? 2 : n < (1u << 24) ? 3 : 4;
    out.push_back((char)((59 + bytes) << 2));
    for (int i = 0; i < bytes; ++i) out.push_back((char)(n >> (8 * i)));
  }
  return out;
}

PqFilePlan plan_parquet_file(const std::vector<std::string>& names, const std::vector<uint64_t>& group_rows, const std::vector<PqChunkIn>& chunks) {
  if (chunks.size() != names.size() * group_rows.size()) throw DxError(DX_E_INVAL, "parquet plan: chunk count differs from row groups x columns");
  bool snappy = false;
  for (const PqChunkIn& c : chunks) snappy |= c.snappy;
  for (const PqChunkIn& c : chunks)
    if (c.snappy != snappy) throw DxError(DX_E_INVAL, "parquet plan: one codec per file");
  PqFilePlan plan;
  plan.chunks.resize(chunks.size());
  uint64_t at = 0, cur_at = 0;
  std::string cur;
  auto host = [&](const std::string& b) { if (cur.empty()) cur_at = at; cur += b; at += b.size(); };
  auto device = [&](uint64_t bytes) {
    if (!cur.empty()) { plan.host.emplace_back(cur_at, std::move(cur)); cur.clear(); }
    const uint64_t here = at;
    at += bytes;
    return here;
  };
  host("PAR1");
  std::vector<FooterGroup> groups;
  uint64_t rows = 0;
  size_t k = 0;
  for (uint64_t gr : group_rows) {
    FooterGroup g;
    g.rows = gr;
    rows += gr;
    for (size_t c = 0; c < names.size(); ++c, ++k) {
      const PqChunkIn& in = chunks[k];
      PqChunkOut& out = plan.chunks[k];
      ChunkMeta m{};
      m.num_values = (int64_t)in.num_values;
      if (in.plain) {
        if (in.plain_bytes >= (1ull << 31)) throw DxError(DX_E_RANGE, "data page of 2 GiB or more");
        m.dict_offset = -1;
        m.data_offset = (int64_t)at;
        size_t header_bytes;
        if (!snappy) {
          const std::string h = page_header(kDataPage, in.plain_bytes, in.plain_bytes, in.num_values, false);
          header_bytes = h.size();
          host(h);
          out.data_at = device(in.plain_bytes);
          out.data_bytes = in.plain_bytes;
        } else {
          // the length it unpacks to, then either what the device packed or the body as one literal
          const std::string pre = snappy_preamble(in.plain_bytes) + (in.terms_packed ? std::string() : snappy_literal_tag(in.plain_bytes));
          const uint64_t from_device = in.terms_packed ? in.terms_packed : in.plain_bytes;
          if (pre.size() + from_device >= (1ull << 31)) throw DxError(DX_E_RANGE, "data page of 2 GiB or more");
          const std::string h = page_header(kDataPage, in.plain_bytes, pre.size() + from_device, in.num_values, false);
          header_bytes = h.size();
          host(h);
          host(pre);
          out.data_at = device(from_device);
          out.data_bytes = from_device;
        }
        m.total_uncompressed = (int64_t)(header_bytes + in.plain_bytes);
        m.total_size = (int64_t)at - m.data_offset;
      } else {
        if (in.dict_bytes >= (1ull << 31)) throw DxError(DX_E_RANGE, "dictionary page of 2 GiB or more");
        int width = 0;
        while ((1ull << width) < in.n_dict) ++width;
        const uint64_t n8 = (in.num_values + 7) / 8;
        std::string prefix;  // bit width, then the header of one bit-packed run of n8 groups of eight values
        prefix.push_back((char)width);
        { Thrift v; v.varint((n8 << 1) | 1); prefix += v.out; }
        const uint64_t body = prefix.size() + n8 * (uint64_t)width;
        if (body >= (1ull << 31)) throw DxError(DX_E_RANGE, "data page of 2 GiB or more");
        m.dict_offset = (int64_t)at;
        size_t header_bytes;
        if (!snappy) {
          const std::string hd = page_header(kDictionaryPage, in.dict_bytes, in.dict_bytes, in.n_dict, true);
          host(hd);
          out.dict_at = device(in.dict_bytes);
          m.data_offset = (int64_t)at;
          const std::string hp = page_header(kDataPage, body, body, in.num_values, true);
          host(hp);
          host(prefix);
          header_bytes = hd.size() + hp.size();
        } else {
          const std::string pre_d = snappy_preamble(in.dict_bytes) + (in.terms_packed || !in.dict_bytes ? std::string() : snappy_literal_tag(in.dict_bytes));
          const uint64_t from_device = in.terms_packed ? in.terms_packed : in.dict_bytes;
          const std::string hd = page_header(kDictionaryPage, in.dict_bytes, pre_d.size() + from_device, in.n_dict, true);
          host(hd);
          host(pre_d);
          out.dict_at = device(from_device);
          m.data_offset = (int64_t)at;
          const std::string pre_p = snappy_preamble(body) + snappy_literal_tag(body);
          const std::string hp = page_header(kDataPage, body, pre_p.size() + body, in.num_values, true);
          host(hp);
          host(pre_p);
          host(prefix);
          header_bytes = hd.size() + hp.size();
        }
        out.width = (uint32_t)width;
        out.data_bytes = n8 * (uint64_t)width;
        out.data_at = device(out.data_bytes);
        m.total_uncompressed = (int64_t)(header_bytes + in.dict_bytes + body);
        m.total_size = (int64_t)at - m.dict_offset;
      }
      g.cols.push_back(m);
    }
    groups.push_back(std::move(g));
  }
  host(footer_bytes(names, rows, groups, snappy));
  if (!cur.empty()) plan.host.emplace_back(cur_at, std::move(cur));
  plan.size = at;
  return plan;
}

}  // namespace dx
