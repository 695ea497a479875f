// Partitioner hot path (post-DAP stage) as HIP kernels for gfx950.
//
// What the reference computes with string-keyed Spark shuffles (Preprocessor.scala:116-192,
// partitioner/DAP.scala:360-509, 567-629) is restated here as integer set algebra:
//
//   * every term (subject, predicate, object) gets a dense int64 id through an open-addressing hash
//     table in HBM (64-bit hash as bucket choice, full byte compare as proof);
//   * P(C) = bitmask of first-level partitions of class C (class index);
//     M(v) = OR of P(C) over the types of vertex v; E(v) = v is an endpoint of an object-property triple;
//   * a distinct input triple (s, p, o) is placed in the partitions of
//         M(s) | M(o)          if it is an object-property triple        (DAP.scala:575-580)
//         E(s) ? P(o) : 0      if p is rdf:type                           (DAP.scala:583-593)
//         E(s) ? M(s) : 0      if o is a literal                          (DAP.scala:595-602)
//     which is the closed form of the reference's three triplet streams: each of them iterates over
//     edges, so a vertex contributes only if it has an edge, type triples go to the partitions of
//     their own class, label triples to the partitions of all classes of their subject;
//   * sorting the distinct triples by (predicate, s, o) once and splitting them stably by partition
//     bit yields every vertical-partition table as a contiguous, sorted slice.
#include <algorithm>
#include <atomic>
#include <chrono>

#include "dx_device.h"
#include "dx_state.h"

namespace dx {

namespace {

constexpr uint8_t kFlagLiteral = 1, kFlagVertex = 2, kFlagEdge = 4, kFlagNode = 8;
constexpr int kTypeLen = 49;
__constant__ __attribute__((aligned(16))) char c_rdf_type[52] = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>";

__device__ __forceinline__ bool is_eol(uint8_t c) { return c == '\n' || c == '\r'; }
__device__ __forceinline__ bool is_jws(uint8_t c) { return c == ' ' || c == '\t' || c == 0x0B || c == 0x0C; }

// ---- 1. record starts -----------------------------------------------------------------------------
// A line starts at byte 0 and after every terminator (LF, CR LF, or a lone CR: Hadoop's LineRecordReader
// behind sc.textFile); a record is a line that does not start with '#' (Preprocessor.scala:120).
// The text is cut into 32 KiB tiles. k_line_scan counts the records that START in each tile; after a
// scan of those counts the tokeniser (one block per tile) knows the index of its first record.
constexpr int LS_THREADS = 256;
constexpr int LS_TILE = 32768;
constexpr int LS_PER_THREAD = LS_TILE / LS_THREADS;  // 128 bytes

// 0x80 in every byte of v that is <= 0x20 (exact, no cross-byte carries)
__device__ __forceinline__ uint32_t le20_mask(uint32_t v) {
  const uint32_t t = (v & 0x7F7F7F7Fu) + 0x5F5F5F5Fu;  // bit 7 set iff the low seven bits are >= 0x21
  return ~(t | v) & 0x80808080u;
}
// 0x80 in every byte of v that lies in [0x0A, 0x0D]: LF and CR plus the rare VT / FF
__device__ __forceinline__ uint32_t eol_candidates(uint32_t v) {
  const uint32_t lo7 = v & 0x7F7F7F7Fu;
  const uint32_t gt0d = lo7 + 0x72727272u;  // bit 7 set iff low seven bits >= 0x0E
  const uint32_t ge0a = lo7 + 0x76767676u;  // bit 7 set iff low seven bits >= 0x0A
  return ge0a & ~gt0d & ~v & 0x80808080u;
}
// 0x80 in every byte of v equal to c (exact)
__device__ __forceinline__ uint32_t eq_mask(uint32_t v, uint32_t c4) {
  const uint32_t x = v ^ c4;
  return ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) | 0x7F7F7F7Fu);
}
// the four 0x80 flags of a word -> 4 bits
__device__ __forceinline__ uint32_t flags_to_nibble(uint32_t f) {
  const uint32_t y = f >> 7;
  return (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xFu;
}

// One pass over the text, branch-free SWAR on 16 bytes per lane: writes for every 16-byte chunk the
// 16-bit mask of the bytes that start a record, and per 32 KiB tile the number of records starting in
// it. With T = terminator flags (LF | CR) and H = '#' flags, a byte starts a record iff the byte before
// it is a terminator and it is itself neither a terminator nor '#':  S = (T shifted by one byte) & ~(T | H).
// (CR LF: the LF is a terminator, so it is not a start, and the byte after it is.) A start that is
// itself a terminator is an empty line, on which the reference throws.
//
// Each wave walks 8 KiB of its tile, 1 KiB per step. The flags of the byte before a lane's chunk come
// from the lane below (a DPP shift, no LDS); lane 0 takes them from lane 63 of the step before, and at
// the start of the wave's stretch from one byte load.
constexpr int LS_WAVE_BYTES = LS_TILE / (LS_THREADS / 64);  // 8 KiB
constexpr int LS_STEPS = LS_WAVE_BYTES / 1024;

struct LineCarry {
  uint32_t t, l;  // 0x80000000 if the byte before the wave's current step is a terminator / an LF
};

// value of `v` in the lane below; lane 0 gets `first`
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t first) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /*wave shift right by 1*/, 0xF, 0xF, false);
}

__device__ __forceinline__ uint32_t line_scan_chunk(const uint32_t (&w)[4], uint64_t i0, uint64_t n, LineCarry& carry,
                                                    uint16_t* __restrict__ chunk_mask, unsigned long long* __restrict__ err) {
  uint32_t L[4], R[4], T[4], H[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    L[d] = eq_mask(w[d], 0x0A0A0A0Au);
    R[d] = eq_mask(w[d], 0x0D0D0D0Du);
    H[d] = eq_mask(w[d], 0x23232323u);
    T[d] = L[d] | R[d];
  }
  const uint32_t tin = from_lane_below(T[3], carry.t);
  const uint32_t lin = from_lane_below(L[3], carry.l);
  carry.t = (uint32_t)__builtin_amdgcn_readlane((int)T[3], 63);
  carry.l = (uint32_t)__builtin_amdgcn_readlane((int)L[3], 63);
  // Starts first; empty lines (a terminator right after an LF, or a CR after a CR) need two terminators in a row,
  // so the work of telling them from CR LF is done only by lanes that see such a pair.
  uint32_t S[4];
  uint32_t tp = tin, pair = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t Tp = __builtin_amdgcn_alignbit(T[d], tp, 24);  // flags moved up by one byte, carry from the word before
    S[d] = Tp & ~(T[d] | H[d]);
    pair |= Tp & T[d];
    tp = T[d];
  }
  uint32_t m16 = flags_to_nibble(S[0]) | (flags_to_nibble(S[1]) << 4) | (flags_to_nibble(S[2]) << 8) | (flags_to_nibble(S[3]) << 12);
  const uint32_t keep = i0 + 16 <= n ? 0xFFFFu : (i0 < n ? ((1u << (uint32_t)(n - i0)) - 1u) : 0u);  // bytes past the end start nothing
  m16 &= keep;
  if (pair) {
    uint32_t E[4];
    uint32_t lp = lin;
    tp = tin;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t Tp = __builtin_amdgcn_alignbit(T[d], tp, 24);
      const uint32_t Lp = __builtin_amdgcn_alignbit(L[d], lp, 24);
      E[d] = (Lp & T[d]) | ((Tp & ~Lp) & R[d]);  // LF then terminator, or CR then CR
      tp = T[d];
      lp = L[d];
    }
    const uint32_t e16 =
        (flags_to_nibble(E[0]) | (flags_to_nibble(E[1]) << 4) | (flags_to_nibble(E[2]) << 8) | (flags_to_nibble(E[3]) << 12)) & keep;
    if (e16) atomicMin(err, (unsigned long long)(i0 + __ffs((int)e16) - 1));
  }
  chunk_mask[i0 >> 4] = (uint16_t)m16;  // every chunk of every tile, so the tokeniser never reads an unwritten mask
  return __popc(m16);
}

__global__ void __launch_bounds__(LS_THREADS)
k_line_scan(const uint8_t* __restrict__ text, uint64_t n, uint16_t* __restrict__ chunk_mask, uint32_t* __restrict__ tile_counts,
            unsigned long long* __restrict__ err) {
  __shared__ uint32_t lds[8];
  const int lane = threadIdx.x & 63;
  const uint64_t tile0 = (uint64_t)blockIdx.x * LS_TILE;
  const uint64_t wave0 = tile0 + (uint64_t)(threadIdx.x >> 6) * LS_WAVE_BYTES;
  // the byte before the wave's stretch (the text starts as if after a line feed)
  uint8_t pb = '\n';
  if (wave0 != 0) pb = wave0 - 1 < n ? text[wave0 - 1] : (uint8_t)'x';
  LineCarry carry;
  carry.t = (pb == '\n' || pb == '\r') ? 0x80000000u : 0u;
  carry.l = pb == '\n' ? 0x80000000u : 0u;
  uint32_t mine = 0;
  if (tile0 + LS_TILE <= n) {  // whole tile inside the text: all of the wave's loads go out before the first is used
    uint4 v[LS_STEPS];
#pragma unroll
    for (int it = 0; it < LS_STEPS; ++it) v[it] = *reinterpret_cast<const uint4*>(text + wave0 + (uint64_t)it * 1024 + (uint64_t)lane * 16);
#pragma unroll
    for (int it = 0; it < LS_STEPS; ++it) {
      const uint32_t w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
      mine += line_scan_chunk(w, wave0 + (uint64_t)it * 1024 + (uint64_t)lane * 16, n, carry, chunk_mask, err);
    }
  } else {
    for (int it = 0; it < LS_STEPS; ++it) {
      const uint64_t i0 = wave0 + (uint64_t)it * 1024 + (uint64_t)lane * 16;
      uint32_t w[4] = {0x78787878u, 0x78787878u, 0x78787878u, 0x78787878u};
      if (i0 + 16 <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(text + i0);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
      } else if (i0 < n) {
        for (int j = 0; j < 4; ++j)
          for (int k = 0; k < 4; ++k)
            if (i0 + j * 4 + k < n) w[j] = (w[j] & ~(0xFFu << (8 * k))) | ((uint32_t)text[i0 + j * 4 + k] << (8 * k));
      }
      mine += line_scan_chunk(w, i0, n, carry, chunk_mask, err);
    }
  }
  uint32_t total;
  (void)block_excl_scan_256<uint32_t>(mine, lds, total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

// ---- 2. tokenise + hash -------------------------------------------------------------------------------
// Field split = java.lang.String.split("\\s+", 3) (Preprocessor.scala:121); subject and predicate are
// trimmed; the object is the third field minus its last two UTF-16 units, trimmed (:122). Lines with
// fewer than three fields make the reference throw: reported through err.
struct RecordOut {
  uint64_t rep[3];   // (byte offset << 24) | length
  uint64_t hash[3];
  uint8_t kind;      // 0 object-property triple, 1 rdf:type, 2 literal object
  bool bad;
};

__device__ __forceinline__ void trim_range(const uint8_t* t, uint64_t& b, uint64_t& e) {
  while (b < e && t[b] <= 0x20) ++b;
  while (e > b && t[e - 1] <= 0x20) --e;
}
__device__ __forceinline__ uint64_t hash_range(const uint8_t* t, uint64_t b, uint64_t e, uint64_t seed) {
  return hash_bytes(t + b, e - b, seed);
}

// Reference-exact tokeniser of one record read byte by byte from global memory. The fast path below
// hands a record over whenever it meets anything unusual (control characters, non-ASCII bytes where
// the last two units are dropped, a line running past the staged window).
__device__ __noinline__ void tokenise_record_slow(const uint8_t* __restrict__ text, uint64_t n, uint64_t start, uint64_t seed,
                                                  RecordOut& out) {
  uint64_t p = start;
  uint64_t b[3], e[3];
  out.bad = false;
  b[0] = p;
  while (p < n && !is_eol(text[p]) && !is_jws(text[p])) ++p;
  e[0] = p;
  if (p >= n || is_eol(text[p])) out.bad = true;
  if (!out.bad) {
    while (p < n && is_jws(text[p])) ++p;
    b[1] = p;
    while (p < n && !is_eol(text[p]) && !is_jws(text[p])) ++p;
    e[1] = p;
    if (p >= n || is_eol(text[p])) out.bad = true;
  }
  if (out.bad) {
    for (int k = 0; k < 3; ++k) { out.rep[k] = 0; out.hash[k] = 1; }
    out.kind = 0;
    return;
  }
  while (p < n && is_jws(text[p])) ++p;
  b[2] = p;
  while (p < n && !is_eol(text[p])) ++p;
  e[2] = p;
  // dropRight(2): two UTF-16 code units from the end (a 4-byte UTF-8 sequence is two units)
  int units = 2;
  while (units > 0 && e[2] > b[2]) {
    uint64_t q = e[2] - 1;
    while (q > b[2] && (text[q] & 0xC0) == 0x80) --q;
    const int u = ((text[q] & 0xF8) == 0xF0) ? 2 : 1;
    if (u > units) { out.bad = true; break; }  // would split a surrogate pair
    units -= u;
    e[2] = q;
  }
  for (int k = 0; k < 3; ++k) trim_range(text, b[k], e[k]);
  for (int k = 0; k < 3; ++k) {
    const uint64_t len = e[k] - b[k];
    if (len >= (1ull << kRepLenBits)) out.bad = true;
    out.rep[k] = (b[k] << kRepLenBits) | (len & ((1ull << kRepLenBits) - 1));
    out.hash[k] = hash_range(text, b[k], e[k], seed);
  }
  uint8_t kd = 0;
  if (e[1] - b[1] == kTypeLen) {
    bool eq = true;
    for (int i = 0; i < kTypeLen; ++i) eq &= (text[b[1] + i] == (uint8_t)c_rdf_type[i]);
    if (eq) kd = 1;
  }
  if (kd == 0 && e[2] > b[2] && text[b[2]] == '"') kd = 2;
  out.kind = kd;
}

// Fast path: one block stages a 32 KiB tile (+ 1 KiB so that lines starting near its end are whole)
// into LDS with coalesced 16-byte loads, finds the record starts there, and each thread tokenises
// one record out of LDS four bytes at a time (unaligned words via v_alignbyte, delimiter search by SWAR,
// hashing fused into the scan of subject and predicate). The window, the start list and the tables of the
// tile-level term folding below add up to just under 40 KiB, so that four blocks share a CU's 160 KiB.
constexpr int TK_THREADS = 256;
constexpr int TK_OVER = 1024;
constexpr int TK_WIN = LS_TILE + TK_OVER;      // bytes staged per block
constexpr int TK_PRE = 16;                     // bytes staged before the tile (only the last one is looked at)
constexpr int TK_ROUND = TK_THREADS;           // records per round, one per thread (a tile of ordinary lines has ~200)
constexpr int TK_BUCKETS = 768;                // direct-mapped table of the round's terms
constexpr uint32_t kNoEntry = 0xFFFFFFFFu;     // the occurrence of a malformed record: it has no dictionary entry

__device__ __forceinline__ uint32_t lds_word(const uint32_t* win, uint32_t pos) {  // 4 bytes at any byte offset
  const uint32_t i = pos >> 2;
  return __builtin_amdgcn_alignbyte(win[i + 1], win[i], pos & 3);
}
__device__ __forceinline__ uint8_t lds_byte(const uint32_t* win, uint32_t pos) {
  return (uint8_t)(win[pos >> 2] >> ((pos & 3) * 8));
}

// Scans a subject/predicate field starting at pos and feeds its bytes to h; returns the position of the
// byte that ended it (the first byte <= 0x20). The loop has one exit and no bound of its own: the staged
// window is followed by line-feed padding, so it stops at the latest there. Four LDS words are requested
// per round (one wait for four reads: the loop is bound by LDS latency, not by instruction issue); the
// upper word of each unaligned read is carried over as the lower word of the next.
__device__ __forceinline__ uint32_t scan_token(const uint32_t* win, uint32_t pos, TermHash& h) {
  uint32_t i = pos >> 2;
  const uint32_t sh = pos & 3;
  uint32_t lo = win[i];
  uint32_t v0, v1, v2, v3, m0, m1, m2, m3;
  for (;;) {
    const uint32_t w1 = win[i + 1], w2 = win[i + 2], w3 = win[i + 3], w4 = win[i + 4];
    v0 = __builtin_amdgcn_alignbyte(w1, lo, sh);
    v1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
    v2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
    v3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
    m0 = le20_mask(v0); m1 = le20_mask(v1); m2 = le20_mask(v2); m3 = le20_mask(v3);
    if (m0 | m1 | m2 | m3) break;
    h.add_word(v0); h.add_word(v1); h.add_word(v2); h.add_word(v3);
    i += 4;
    lo = w4;
  }
  uint32_t v = v0, m = m0, base = 0;
  if (!m0) {
    h.add_word(v0);
    v = v1; m = m1; base = 1;
    if (!m1) {
      h.add_word(v1);
      v = v2; m = m2; base = 2;
      if (!m2) {
        h.add_word(v2);
        v = v3; m = m3; base = 3;
      }
    }
  }
  const uint32_t k = (uint32_t)(__ffs((int)m) - 1) >> 3;
  if (k) h.add_word(v & ((1u << (8 * k)) - 1u));
  return 4 * (i + base) + sh + k;
}

// ---- dictionary slot claim, shared by the fused tokeniser and the entry-level build -----------------------
constexpr uint32_t kMaxProbes = 1u << 10;  // a longer chain means the table is too full: the build restarts larger
constexpr uint64_t kNoHolder = ~0ull;      // k_ht_clear's representative word: nobody has claimed the slot's term yet

__device__ __forceinline__ uint64_t diff_global(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t len) {
  uint64_t diff = 0;
  uint64_t j = 0;
  for (; j + 8 <= len; j += 8) {
    uint64_t x, y;
    __builtin_memcpy(&x, a + j, 8); __builtin_memcpy(&y, b + j, 8);
    diff |= x ^ y;
  }
  for (; j < len; ++j) diff |= (uint64_t)(a[j] ^ b[j]);
  return diff;
}
// The same for two terms somewhere in the text, 64 bytes of each side requested at a time as four independent
// 16-byte loads (one round trip instead of eight: the loop is bound by memory latency); chunks past the end are
// pulled back to the last 16 bytes, which compares some bytes twice and never reads outside the terms.
__device__ __forceinline__ uint32_t diff_global_wide(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t len) {
  if (len < 16) return diff_global(a, b, len) != 0;
  uint32_t diff = 0;
  const uint32_t last = len - 16;
  for (uint32_t j = 0; j < len; j += 64) {
    uint4 x[4], y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t off = j + 16 * c < last ? j + 16 * c : last;
      __builtin_memcpy(&x[c], a + off, 16);
      __builtin_memcpy(&y[c], b + off, 16);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) diff |= (x[c].x ^ y[c].x) | (x[c].y ^ y[c].y) | (x[c].z ^ y[c].z) | (x[c].w ^ y[c].w);
  }
  return diff;
}

// len bytes at window offset pa against len bytes at window offset pb, both in LDS (two occurrences of one tile).
// Four words of each side are requested per round: the loop is bound by LDS latency.
__device__ __forceinline__ uint32_t diff_lds_lds(const uint32_t* win, uint32_t pa, uint32_t pb, uint32_t len) {
  uint32_t diff = 0;
  uint32_t ia = pa >> 2, ib = pb >> 2;
  const uint32_t sa = pa & 3, sb = pb & 3;
  uint32_t la = win[ia], lb = win[ib];
  uint32_t j = 0;
  for (; j + 16 <= len; j += 16, ia += 4, ib += 4) {
    const uint32_t a1 = win[ia + 1], a2 = win[ia + 2], a3 = win[ia + 3], a4 = win[ia + 4];
    const uint32_t b1 = win[ib + 1], b2 = win[ib + 2], b3 = win[ib + 3], b4 = win[ib + 4];
    diff |= (__builtin_amdgcn_alignbyte(a1, la, sa) ^ __builtin_amdgcn_alignbyte(b1, lb, sb)) |
            (__builtin_amdgcn_alignbyte(a2, a1, sa) ^ __builtin_amdgcn_alignbyte(b2, b1, sb)) |
            (__builtin_amdgcn_alignbyte(a3, a2, sa) ^ __builtin_amdgcn_alignbyte(b3, b2, sb)) |
            (__builtin_amdgcn_alignbyte(a4, a3, sa) ^ __builtin_amdgcn_alignbyte(b4, b3, sb));
    la = a4;
    lb = b4;
  }
  for (; j + 4 <= len; j += 4) {
    const uint32_t ha = win[++ia], hb = win[++ib];
    diff |= __builtin_amdgcn_alignbyte(ha, la, sa) ^ __builtin_amdgcn_alignbyte(hb, lb, sb);
    la = ha;
    lb = hb;
  }
  if (j < len) {
    const uint32_t x = __builtin_amdgcn_alignbyte(win[ia + 1], la, sa) ^ __builtin_amdgcn_alignbyte(win[ib + 1], lb, sb);
    diff |= x & ((1u << (8 * (len - j))) - 1u);
  }
  return diff;
}

// Claims or finds the slot of hash h and lowers its representative to `mine` if that is earlier in the text.
// Returns the slot; `holder` = a representative that was in the slot before this call took effect (some other
// occurrence that claims the same string), or kNoHolder for the very first claim. Every occurrence but that
// first one is handed a holder and compares its bytes with it, so by induction over the order in which the
// atomics landed all occurrences of a slot carry the same string, or a collision is reported.
__device__ __forceinline__ uint32_t claim_slot(DictSlot* __restrict__ ht, uint64_t mask, uint64_t h, uint64_t mine, uint64_t& holder,
                                               uint32_t* __restrict__ overflow) {
  uint64_t slot = h & mask;
  holder = kNoHolder;
  for (uint32_t probes = 0; probes < kMaxProbes; ++probes) {
    // key and representative share one 16-byte slot: one load answers "is this my slot" and "am I earlier"
    const uint4 sv = *reinterpret_cast<const uint4*>(&ht[slot]);
    uint64_t cur = ((uint64_t)sv.y << 32) | sv.x;
    if (cur == 0) {
      cur = atomicCAS((unsigned long long*)&ht[slot].key, 0ull, (unsigned long long)h);
      if (cur == 0 || cur == h) {
        holder = atomicMin((unsigned long long*)&ht[slot].val, (unsigned long long)mine);
        return (uint32_t)slot;
      }
    } else if (cur == h) {
      const uint64_t r = ((uint64_t)sv.w << 32) | sv.z;
      if (r == kNoHolder) {  // claimed a moment ago, representative not published yet: the atomic's answer decides
        holder = atomicMin((unsigned long long*)&ht[slot].val, (unsigned long long)mine);
      } else {               // any published representative serves as holder; lowering it needs no answer
        holder = r;
        if (r > mine) atomicMin((unsigned long long*)&ht[slot].val, (unsigned long long)mine);
      }
      return (uint32_t)slot;
    }
    slot = (slot + 1) & mask;
  }
  *overflow = 1;
  return 0;
}

// Stages [tile0 - TK_PRE, tile0 + TK_WIN) of the text into `win`, clipped to the text; bytes outside it and
// the eight padding words after the window read as '\n'. Ends with a block barrier.
// LDS-DMA: every wave-instruction moves 64 x 16 contiguous bytes straight into the window (no VGPR staging,
// all copies of a wave in flight at once). Only the spans touching the ends of the text are assembled by hand.
__device__ __forceinline__ void stage_window(uint32_t* win, const uint8_t* __restrict__ text, uint64_t n, uint64_t tile0) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  constexpr uint32_t kStageBytes = TK_PRE + TK_WIN;
#pragma unroll
  for (uint32_t it = 0; it < (kStageBytes + TK_THREADS * 16 - 1) / (TK_THREADS * 16); ++it) {
    const uint32_t wave_c = it * (TK_THREADS * 16) + wv * 1024;  // window offset of this wave's 1 KiB span
    if (wave_c >= kStageBytes) continue;
    const int64_t wave_g = (int64_t)tile0 - TK_PRE + wave_c;
    const uint32_t c = wave_c + ln * 16;
    if (wave_g >= 0 && (uint64_t)wave_g + 1024 <= n && wave_c + 1024 <= kStageBytes) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(text + wave_g + ln * 16),
                                       (__attribute__((address_space(3))) void*)&win[wave_c >> 2], 16, 0, 0);
    } else if (c < kStageBytes) {
      const int64_t g = wave_g + ln * 16;
      uint32_t w[4];
      for (int j = 0; j < 4; ++j) {
        w[j] = 0;
        for (int k = 0; k < 4; ++k) {
          const int64_t gi = g + j * 4 + k;
          const uint8_t ch = (gi >= 0 && (uint64_t)gi < n) ? text[gi] : (uint8_t)'\n';
          w[j] |= (uint32_t)ch << (8 * k);
        }
      }
      *reinterpret_cast<uint4*>(&win[c >> 2]) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  if (threadIdx.x < 8) win[(TK_PRE + TK_WIN) / 4 + threadIdx.x] = 0x0A0A0A0Au;
  __builtin_amdgcn_s_waitcnt(0);  // the DMA copies count on vmcnt
  __syncthreads();
}

// One block per tile. Every record of the tile is tokenised out of LDS by one thread; then the terms of the
// round are folded: a direct-mapped table in LDS names, per bucket, the earliest occurrence that fell into it,
// and every other occurrence of the bucket compares its bytes with that one (both are in the window). An
// occurrence that equals an earlier one of its round only records that occurrence's dictionary entry; the rest
// (the leaders: ~140 of a LUBM tile's ~550 occurrences) become the tile's dictionary entries, written densely
// at the tile's own place, [3 * first_rec, 3 * first_rec + leaders): entry indices grow with the position in
// the text. Only those entries go to the table in HBM afterwards (k_entry_claim), in a kernel without the
// window in LDS and therefore at full occupancy. Two different terms in one bucket just make two leaders.
//
// FUSED: the block also finds the record starts of its tile itself (the rule of k_line_scan, applied to the staged
// window), so the text is read from HBM once instead of twice. The index of a tile's first record is then not
// known when the tile is tokenised: the tile's records and entries go to a place of their own, slots
// [tile * slots_per_tile, + records of the tile), and are packed once the counts have been scanned (k_encode_tiles).
// A tile with more records than slots reports overflow, and the caller falls back to the two-pass form.
constexpr uint32_t kSlotsPerTile = 512;  // 64 bytes per record on average inside a tile; LUBM lines average 178

// Record starts among the 16 bytes of a chunk (see k_line_scan for the rule): `tin` / `lin` are the terminator / LF
// flags of the word before the chunk. Empty lines are reported through err.
__device__ __forceinline__ uint32_t chunk_starts(const uint32_t (&w)[4], uint32_t tin, uint32_t lin, uint32_t keep, uint64_t i0,
                                                 unsigned long long* __restrict__ err, uint32_t& not_plain) {
  uint32_t L[4], R[4], T[4], H[4], S[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    L[d] = eq_mask(w[d], 0x0A0A0A0Au);
    R[d] = eq_mask(w[d], 0x0D0D0D0Du);
    H[d] = eq_mask(w[d], 0x23232323u);
    T[d] = L[d] | R[d];
  }
  uint32_t tp = tin, pair = 0, hash_after = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t Tp = __builtin_amdgcn_alignbit(T[d], tp, 24);
    S[d] = Tp & ~(T[d] | H[d]);
    pair |= Tp & T[d];
    hash_after |= Tp & H[d];
    tp = T[d];
  }
  // anything but lines that one LF ends and a record follows: a CR, a VT or FF (bytes 0x0B / 0x0C are no terminators but
  // sit in the same range), a comment line, two terminators in a row
  not_plain = R[0] | R[1] | R[2] | R[3] | pair | hash_after | (tin & ~lin) |
              ((eol_candidates(w[0]) ^ L[0]) | (eol_candidates(w[1]) ^ L[1]) | (eol_candidates(w[2]) ^ L[2]) | (eol_candidates(w[3]) ^ L[3]));
  const uint32_t m16 = (flags_to_nibble(S[0]) | (flags_to_nibble(S[1]) << 4) | (flags_to_nibble(S[2]) << 8) | (flags_to_nibble(S[3]) << 12)) & keep;
  if (pair) {
    uint32_t E[4];
    uint32_t lp = lin;
    tp = tin;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t Tp = __builtin_amdgcn_alignbit(T[d], tp, 24);
      const uint32_t Lp = __builtin_amdgcn_alignbit(L[d], lp, 24);
      E[d] = (Lp & T[d]) | ((Tp & ~Lp) & R[d]);  // LF then terminator, or CR then CR
      tp = T[d];
      lp = L[d];
    }
    const uint32_t e16 =
        (flags_to_nibble(E[0]) | (flags_to_nibble(E[1]) << 4) | (flags_to_nibble(E[2]) << 8) | (flags_to_nibble(E[3]) << 12)) & keep;
    if (e16) atomicMin(err, (unsigned long long)(i0 + __ffs((int)e16) - 1));
  }
  return m16;
}

template <bool FUSED>
__global__ void __launch_bounds__(TK_THREADS)
k_tokenise(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ chunk_mask /*two-pass form*/,
           const uint32_t* __restrict__ tile_first /*two-pass form: scanned counts, ntiles+1*/, uint64_t seed, uint64_t type_hash,
           uint64_t n_rec /*records, or slots in the fused form: the stride of t_entry's fields*/,
           ulonglong2* __restrict__ ent /*[entry]: x = (offset << 24) | length, y = hash*/,
           uint32_t* __restrict__ t_entry /*[field][record]: the occurrence's entry*/, uint32_t* __restrict__ tile_leaders /*[tile]*/,
           uint8_t* __restrict__ kind, unsigned long long* __restrict__ err,
           uint32_t* __restrict__ tile_counts /*fused form, out: records that start in the tile*/, uint32_t* __restrict__ overflow /*fused form*/) {
  __shared__ __attribute__((aligned(16))) uint32_t win[(TK_PRE + TK_WIN) / 4 + 8];
  __shared__ __attribute__((aligned(16))) uint32_t fold_lds[3 * TK_ROUND + TK_BUCKETS];
  uint32_t* const occ_info = fold_lds;             // window offset | length << 16 of the round's occurrences
  uint32_t* const table = fold_lds + 3 * TK_ROUND; // earliest occurrence per bucket; afterwards the leaders' ranks (16 bits each)
  __shared__ uint16_t starts[TK_ROUND];
  __shared__ uint32_t scan_lds[8];
  // fused form: every line of the tile ends in one LF and is followed by a record (no CR, VT, FF, comment or empty line):
  // a record then ends right before the next one starts, and its end need not be searched for
  __shared__ uint32_t plain_tile;
  const uint64_t tile0 = (uint64_t)blockIdx.x * LS_TILE;
  uint32_t first_rec, n_here;
  if constexpr (FUSED) {
    first_rec = blockIdx.x * kSlotsPerTile;
    n_here = 0;
  } else {
    first_rec = tile_first[blockIdx.x];
    n_here = tile_first[blockIdx.x + 1] - first_rec;
    if (n_here == 0) {  // block-uniform
      if (threadIdx.x == 0) tile_leaders[blockIdx.x] = 0;
      return;
    }
  }
  // stage [tile0 - TK_PRE, tile0 + TK_WIN) clipped to the text; bytes outside read as '\n'
  const uint64_t avail = n - tile0;
  const uint32_t win_bytes = avail < (uint64_t)TK_WIN ? (uint32_t)avail : (uint32_t)TK_WIN;
  // record starts of this tile, in order. Positions are window offsets (tile byte i is at TK_PRE + i).
  const uint32_t seg = TK_PRE + threadIdx.x * LS_PER_THREAD;  // thread t owns starts at [seg, seg + 128)
  uint32_t smask[4] = {0, 0, 0, 0};  // bit i of the 128: a record starts at byte i of this thread's segment
  if constexpr (!FUSED) {  // requested before the window is staged: the masks travel together with the text
    const uint32_t seg_tile = threadIdx.x * LS_PER_THREAD;
    if (seg_tile < win_bytes) {
      const uint4 mv = *reinterpret_cast<const uint4*>(chunk_mask + ((tile0 + seg_tile) >> 4));  // 8 chunk masks = 128 bits
      smask[0] = mv.x; smask[1] = mv.y; smask[2] = mv.z; smask[3] = mv.w;
    }
  }
  stage_window(win, text, n, tile0);
  if constexpr (FUSED) {
    // Chunk c of the tile (16 bytes) is looked at by thread c % 256: neighbouring lanes read neighbouring chunks, which
    // LDS serves without bank conflicts; the 16-bit masks go through LDS to the thread that owns the chunk's segment.
    uint16_t* cmask = reinterpret_cast<uint16_t*>(fold_lds);  // 2048 masks = 4 KiB, gone before the folding tables are used
    const uint32_t tile_bytes = avail < (uint64_t)LS_TILE ? (uint32_t)avail : (uint32_t)LS_TILE;
    // A chunk can hold a start (or an empty line) only if it, or the byte before it, holds a byte in [0x0A, 0x0D]: one
    // chunk in ten on ordinary lines. A cheap test picks those, and only they get the exact treatment (a lane has
    // eight chunks; the lanes of a wave loop as often as the lane with the most such chunks, three times or so).
    uint32_t need = 0;
#pragma unroll
    for (int q = 0; q < LS_TILE / 16 / TK_THREADS; ++q) {
      const uint32_t c = q * TK_THREADS + threadIdx.x;
      const uint32_t off = TK_PRE + c * 16;  // window offset of the chunk
      const uint4 v = *reinterpret_cast<const uint4*>(&win[off >> 2]);
      const uint32_t before = win[(off >> 2) - 1] | 0x00FFFFFFu;  // only its last byte counts
      // "some byte below 0x0E" (exact as a yes / no for the sixteen bytes and the one before; tabs pass it too)
      const uint32_t cand = (((v.x - 0x0E0E0E0Eu) & ~v.x) | ((v.y - 0x0E0E0E0Eu) & ~v.y) | ((v.z - 0x0E0E0E0Eu) & ~v.z) | ((v.w - 0x0E0E0E0Eu) & ~v.w) |
                             ((before - 0x0E0E0E0Eu) & ~before)) & 0x80808080u;
      if (cand) need |= 1u << q;
      cmask[c] = 0;
    }
    if (threadIdx.x == 0) plain_tile = 1;
    auto exact = [&](uint32_t c) -> uint32_t {
      const uint32_t off = TK_PRE + c * 16;
      const uint4 v = *reinterpret_cast<const uint4*>(&win[off >> 2]);
      const uint32_t before = win[(off >> 2) - 1];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      const uint32_t lin = eq_mask(before, 0x0A0A0A0Au), tin = lin | eq_mask(before, 0x0D0D0D0Du);
      const uint32_t i = c * 16;  // tile offset
      const uint32_t keep = i + 16 <= tile_bytes ? 0xFFFFu : (i < tile_bytes ? ((1u << (tile_bytes - i)) - 1u) : 0u);
      uint32_t not_plain;
      cmask[c] = (uint16_t)chunk_starts(w, tin & 0x80000000u, lin & 0x80000000u, keep, tile0 + i, err, not_plain);
      return not_plain;
    };
    // The chunks that passed are listed and then worked through by all threads side by side: a tile of ordinary lines
    // has some two hundred of them, a tenth of its chunks, but the lane with the most has three, and a wave would
    // take as many turns as that lane.
    uint16_t* const list = cmask + LS_TILE / 16;  // what is left of the folding tables' space behind the masks
    constexpr uint32_t kListMax = (sizeof(fold_lds) - (LS_TILE / 16) * 2) / 2;
    uint32_t listed;
    uint32_t at = block_excl_scan_256<uint32_t>((uint32_t)__popc(need), scan_lds, listed);
    uint32_t odd_bytes = 0;
    if (listed <= kListMax) {  // block-uniform
      for (uint32_t m = need; m; m &= m - 1) list[at++] = (uint16_t)(((uint32_t)__ffs((int)m) - 1) * TK_THREADS + threadIdx.x);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < listed; i += TK_THREADS) odd_bytes |= exact(list[i]);
    } else {
      for (uint32_t m = need; m; m &= m - 1) odd_bytes |= exact(((uint32_t)__ffs((int)m) - 1) * TK_THREADS + threadIdx.x);
    }
    if (odd_bytes) plain_tile = 0;
    __syncthreads();
    const uint4 mv = *reinterpret_cast<const uint4*>(cmask + threadIdx.x * (LS_PER_THREAD / 16));
    smask[0] = mv.x; smask[1] = mv.y; smask[2] = mv.z; smask[3] = mv.w;
  }
  uint32_t at0;
  {
    const uint32_t cnt = __popc(smask[0]) + __popc(smask[1]) + __popc(smask[2]) + __popc(smask[3]);
    uint32_t total;
    at0 = block_excl_scan_256<uint32_t>(cnt, scan_lds, total);  // its barriers end the reads of the masks in LDS
    if constexpr (FUSED) {
      n_here = total;  // block-uniform
      if (threadIdx.x == 0) tile_counts[blockIdx.x] = n_here;
      if (n_here == 0 || n_here > kSlotsPerTile) {
        if (threadIdx.x == 0) {
          tile_leaders[blockIdx.x] = 0;
          if (n_here) *overflow = 1;
        }
        return;
      }
    } else if (__syncthreads_or(total != n_here)) {  // the masks and the counts come from one kernel: cannot differ
      if (threadIdx.x == 0) { atomicMin(err, (unsigned long long)tile0); tile_leaders[blockIdx.x] = 0; }
      for (uint32_t lr = threadIdx.x; lr < n_here; lr += TK_THREADS) {
        const uint64_t r = (uint64_t)first_rec + lr;
        for (int k = 0; k < 3; ++k) t_entry[k * n_rec + r] = kNoEntry;
        kind[r] = 0;
      }
      return;
    }
  }
  const uint32_t limit = TK_PRE + win_bytes;           // one past the last staged text byte
  const bool window_is_eof = tile0 + win_bytes >= n;   // staged window reaches the end of the text
  uint32_t leaders_before = 0;                         // leaders of the tile's earlier rounds
  // One record per thread and round; only a tile of very short lines needs a second round.
  for (uint32_t base = 0; base < n_here; base += TK_ROUND) {  // block-uniform
    if (base) __syncthreads();
    {
      uint32_t at = at0 - base;  // wraps below the round's range and fails the bound check there
#pragma unroll
      for (int wd = 0; wd < 4; ++wd) {
        uint32_t m = smask[wd];
        while (m) {
          const int bit = __ffs((int)m) - 1;
          m &= m - 1;
          if (at < (uint32_t)TK_ROUND) starts[at] = (uint16_t)(seg + wd * 32 + bit);
          ++at;
        }
      }
      for (uint32_t i = threadIdx.x; i < (uint32_t)TK_BUCKETS; i += TK_THREADS) table[i] = ~0u;
    }
    __syncthreads();
    const uint32_t lr = base + threadIdx.x;
    const bool has = lr < n_here;
    RecordOut out;
    out.bad = false;
    out.kind = 0;
    for (int k = 0; k < 3; ++k) { out.rep[k] = 0; out.hash[k] = 0; }
    bool odd = false;
    uint32_t b[3] = {0, 0, 0}, len[3] = {0, 0, 0};
    if (has) {
      const uint32_t p0 = starts[threadIdx.x];
      const uint64_t gstart = tile0 + (p0 - TK_PRE);
      // Fast path: "<s> <p> <o> ." shaped lines. Anything else -- a field ended by something that is not
      // blank, control characters, a line that is too short or runs past the window, non-ASCII bytes where
      // the last two units are cut -- is left to the reference-exact slow path, which also reports errors.
      uint32_t e[3];
      TermHash hs(seed), hp(seed), ho(seed);
      b[0] = p0;
      e[0] = scan_token(win, p0, hs);
      uint32_t p = e[0];
      odd |= !is_jws(lds_byte(win, p)) | (p >= limit);
      ++p;
      if (lds_byte(win, p) <= 0x20) {  // more than one blank between the fields: rare
        while (p < limit && is_jws(lds_byte(win, p))) ++p;
        odd |= lds_byte(win, p) <= 0x20;
      }
      b[1] = p;
      e[1] = scan_token(win, p, hp);
      p = e[1];
      odd |= !is_jws(lds_byte(win, p)) | (p >= limit);
      ++p;
      if (lds_byte(win, p) <= 0x20) {
        while (p < limit && is_jws(lds_byte(win, p))) ++p;
        odd |= (lds_byte(win, p) <= 0x20) && !is_eol(lds_byte(win, p));  // an empty object is fine
      }
      b[2] = p;
      bool end_known = false;
      if constexpr (FUSED) {
        // In a tile of plain lines the record ends at the LF right before the next record's start (which this block has
        // found itself); only the tile's last record, whose line ends beyond the tile, is searched as below.
        if (plain_tile && threadIdx.x + 1 < (uint32_t)TK_ROUND && lr + 1 < n_here) {
          e[2] = (uint32_t)starts[threadIdx.x + 1] - 1;
          end_known = true;
        }
      }
      if (!end_known) {  // end of line: first LF / CR at or after the object's first byte, four words per round
        uint32_t i = p >> 2;
        const uint32_t sh = p & 3;
        uint32_t lo = win[i];
        uint32_t v0, v1, v2, v3, m0, m1, m2, m3;
        for (;;) {
          const uint32_t w1 = win[i + 1], w2 = win[i + 2], w3 = win[i + 3], w4 = win[i + 4];
          v0 = __builtin_amdgcn_alignbyte(w1, lo, sh);
          v1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
          v2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
          v3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
          m0 = eol_candidates(v0); m1 = eol_candidates(v1); m2 = eol_candidates(v2); m3 = eol_candidates(v3);
          if (m0 | m1 | m2 | m3) break;
          i += 4;
          lo = w4;
        }
        uint32_t v = v0, m = m0, wbase = 0;
        if (!m0) { v = v1; m = m1; wbase = 1; if (!m1) { v = v2; m = m2; wbase = 2; if (!m2) { v = v3; m = m3; wbase = 3; } } }
        const uint32_t k = (uint32_t)(__ffs((int)m) - 1) >> 3;
        odd |= !is_eol((uint8_t)(v >> (8 * k)));  // a VT / FF inside the object
        e[2] = 4 * (i + wbase) + sh + k;
      }
      odd |= (e[2] > limit) | ((e[2] == limit) & !window_is_eof);  // the line runs past the staged window
      if (!odd) {
        // dropRight(2) then trim
        if (e[2] - b[2] >= 2) {
          odd |= ((lds_byte(win, e[2] - 1) | lds_byte(win, e[2] - 2)) & 0x80) != 0;
          e[2] -= 2;
        } else {
          e[2] = b[2];
        }
        while (e[2] > b[2] && lds_byte(win, e[2] - 1) <= 0x20) --e[2];
        while (b[2] < e[2] && lds_byte(win, b[2]) <= 0x20) ++b[2];
      }
      if (odd) {
        // The slow path is a call: what it fills in lives in memory. A record of its own keeps the fast path's results
        // in registers (filled in place, they were stored to and loaded from scratch memory for every record: 64 bytes
        // per lane, more HBM writes than the kernel's real output).
        RecordOut slow;
        tokenise_record_slow(text, n, gstart, seed, slow);
        out.bad = slow.bad;
        out.kind = slow.kind;
        for (int k = 0; k < 3; ++k) { out.rep[k] = slow.rep[k]; out.hash[k] = slow.hash[k]; }
      } else {
        {  // object hash, four words per round
          uint32_t q = b[2], i = q >> 2;
          const uint32_t sh = q & 3;
          uint32_t lo = win[i];
          for (; q + 16 <= e[2]; q += 16, i += 4) {
            const uint32_t w1 = win[i + 1], w2 = win[i + 2], w3 = win[i + 3], w4 = win[i + 4];
            ho.add_word(__builtin_amdgcn_alignbyte(w1, lo, sh));
            ho.add_word(__builtin_amdgcn_alignbyte(w2, w1, sh));
            ho.add_word(__builtin_amdgcn_alignbyte(w3, w2, sh));
            ho.add_word(__builtin_amdgcn_alignbyte(w4, w3, sh));
            lo = w4;
          }
          for (; q + 4 <= e[2]; q += 4, ++i) {
            const uint32_t hi = win[i + 1];
            ho.add_word(__builtin_amdgcn_alignbyte(hi, lo, sh));
            lo = hi;
          }
          if (q < e[2]) ho.add_word(__builtin_amdgcn_alignbyte(win[i + 1], lo, sh) & ((1u << (8 * (e[2] - q))) - 1u));
        }
        for (int k = 0; k < 3; ++k) len[k] = e[k] - b[k];
        out.hash[0] = hs.finish(len[0]);
        out.hash[1] = hp.finish(len[1]);
        out.hash[2] = ho.finish(len[2]);
        for (int k = 0; k < 3; ++k) out.rep[k] = ((tile0 + (b[k] - TK_PRE)) << kRepLenBits) | len[k];
        uint8_t kd = 0;
        if (len[1] == kTypeLen && out.hash[1] == type_hash) {  // hash first, then the proof, a word at a time
          const uint32_t* tw = reinterpret_cast<const uint32_t*>(c_rdf_type);
          bool eq = true;
#pragma unroll
          for (int i = 0; i < kTypeLen / 4; ++i) eq &= lds_word(win, b[1] + 4 * i) == tw[i];
          eq &= lds_byte(win, b[1] + kTypeLen - 1) == (uint8_t)c_rdf_type[kTypeLen - 1];
          if (eq) kd = 1;
        }
        if (kd == 0 && len[2] && lds_byte(win, b[2]) == '"') kd = 2;
        out.kind = kd;
      }
      if (out.bad) atomicMin(err, (unsigned long long)gstart);
    }
    // ---- the round's terms folded onto their earliest occurrence
    const bool good = has && !out.bad;
    const uint32_t my = 3 * threadIdx.x;  // occurrence numbers grow with the position in the text
    bool in_table[3], lead[3];
    uint32_t bucket[3], owner[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      in_table[k] = good && !odd && len[k] < 65536u;  // bytes in the window (a slow-path record's need not be)
      lead[k] = good;
      owner[k] = my + k;
      bucket[k] = (uint32_t)(((out.hash[k] >> 32) * (uint64_t)TK_BUCKETS) >> 32);
      if (in_table[k]) {
        occ_info[my + k] = b[k] | (len[k] << 16);
        atomicMin(&table[bucket[k]], my + k);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (!in_table[k]) continue;
      const uint32_t o = table[bucket[k]];
      if (o == my + k) continue;
      const uint32_t info = occ_info[o];
      if ((info >> 16) == len[k] && diff_lds_lds(win, b[k], info & 0xFFFFu, len[k]) == 0) { lead[k] = false; owner[k] = o; }
    }
    const uint32_t mine_cnt = (uint32_t)lead[0] + (uint32_t)lead[1] + (uint32_t)lead[2];
    uint32_t total;
    const uint32_t ex = block_excl_scan_256<uint32_t>(mine_cnt, scan_lds, total);  // its barriers end the reads of `table`
    uint16_t* rank16 = reinterpret_cast<uint16_t*>(table);
    uint32_t rk[3] = {ex, ex + (uint32_t)lead[0], ex + (uint32_t)lead[0] + (uint32_t)lead[1]};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (lead[k] && in_table[k]) rank16[my + k] = (uint16_t)rk[k];
    __syncthreads();
    const uint64_t ebase = 3ull * first_rec + leaders_before;
    if (has) {
      const uint64_t r = (uint64_t)first_rec + lr;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (good && !lead[k]) rk[k] = rank16[owner[k]];
        if (lead[k]) ent[ebase + rk[k]] = make_ulonglong2(out.rep[k], out.hash[k]);
        t_entry[k * n_rec + r] = good ? (uint32_t)(ebase + rk[k]) : kNoEntry;  // field-major: coalesced
      }
      kind[r] = good ? out.kind : (uint8_t)0;
    }
    leaders_before += total;
  }
  if (threadIdx.x == 0) tile_leaders[blockIdx.x] = leaders_before;
}

// ---- 3. dictionary ---------------------------------------------------------------------------------
// The kernels over the tiles' entries give one wave to a tile: its entries are [3 * first_rec, + leaders).
constexpr int EN_TILES_PER_BLOCK = 1;

// Every entry claims / finds the slot of its hash and lowers the slot's representative to the smallest
// (offset, length) word, i.e. the first occurrence in the text: ids do not depend on timing.
// An entry that finds an earlier representative r published in its slot can never be the first occurrence: it
// is compared with r byte by byte at once, both read from the text (flag 0). An entry that lowered the
// representative, or claimed the slot, may be the first (flag 2): k_entry_first decides when all claims have
// landed, and compares it with the final representative if it is not. So every entry but the representative has
// been compared with an entry of its slot that lies earlier in the text: with the comparisons inside the tiles
// all occurrences that share a slot carry one string, or a collision is reported.
constexpr uint8_t kMaybeFirst = 2;
__global__ void __launch_bounds__(64 * EN_TILES_PER_BLOCK)
k_entry_claim(const uint32_t* __restrict__ tile_first, const uint32_t* __restrict__ tile_leaders, unsigned ntiles,
              const uint8_t* __restrict__ text, const ulonglong2* __restrict__ ent, DictSlot* __restrict__ ht, uint64_t mask,
              uint32_t* __restrict__ e_slot, uint8_t* __restrict__ first, uint32_t* __restrict__ coll /*[0] collision, [1] overflow*/) {
  const unsigned tile = blockIdx.x * EN_TILES_PER_BLOCK + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const uint64_t base = 3ull * tile_first[tile];
  const uint32_t L = tile_leaders[tile];
  for (uint32_t i = threadIdx.x & 63; i < L; i += 64) {
    uint8_t f = 0;
    {
      const ulonglong2 en = ent[base + i];
      const uint64_t mine = en.x, h = en.y;
      uint64_t slot = h & mask, holder = kNoHolder;
      uint32_t probes = 0;
      for (; probes < kMaxProbes; ++probes) {
        const uint4 sv = *reinterpret_cast<const uint4*>(&ht[slot]);  // key and representative in one load
        uint64_t cur = ((uint64_t)sv.y << 32) | sv.x;
        if (cur == 0) {
          cur = atomicCAS((unsigned long long*)&ht[slot].key, 0ull, (unsigned long long)h);
          if (cur == 0 || cur == h) {
            atomicMin((unsigned long long*)&ht[slot].val, (unsigned long long)mine);
            f = kMaybeFirst;
            break;
          }
        } else if (cur == h) {
          const uint64_t r = ((uint64_t)sv.w << 32) | sv.z;
          if (r == kNoHolder || r > mine) {
            atomicMin((unsigned long long*)&ht[slot].val, (unsigned long long)mine);
            f = kMaybeFirst;
          } else {
            holder = r;
          }
          break;
        }
        slot = (slot + 1) & mask;
      }
      if (probes == kMaxProbes) { coll[1] = 1; slot = 0; }
      e_slot[base + i] = (uint32_t)slot;
      if (holder != kNoHolder) {
        const uint64_t lm = mine & ((1ull << kRepLenBits) - 1);
        if (lm != (holder & ((1ull << kRepLenBits) - 1)) ||
            diff_global_wide(text + (mine >> kRepLenBits), text + (holder >> kRepLenBits), (uint32_t)lm)) coll[0] = 1;
      }
    }
    first[base + i] = f;
  }
}

// first[x] = 1 when entry x is its slot's representative (the earliest occurrence of its term in the text).
// Only the entries k_entry_claim left undecided are looked at. tile_firsts[tile] = the tile's number of them.
__global__ void __launch_bounds__(64 * EN_TILES_PER_BLOCK)
k_entry_first(const uint32_t* __restrict__ tile_first, const uint32_t* __restrict__ tile_leaders, unsigned ntiles,
              const uint8_t* __restrict__ text, const ulonglong2* __restrict__ ent, const uint32_t* __restrict__ e_slot,
              const DictSlot* __restrict__ ht, uint8_t* __restrict__ first, uint32_t* __restrict__ tile_firsts,
              uint32_t* __restrict__ tile_bytes /*[tile]: bytes of the tile's representatives, i.e. its share of the string pool*/,
              uint32_t* __restrict__ collision) {
  // A third of a tile's entries are undecided. The wave first lists them (their numbers inside the tile, in LDS), then
  // works through the list with all lanes busy: one chain of dependent loads per 64 undecided entries instead of one
  // per 64 entries of any kind.
  constexpr uint32_t kList = 512;
  __shared__ uint16_t list[EN_TILES_PER_BLOCK][kList];
  const unsigned tile = blockIdx.x * EN_TILES_PER_BLOCK + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint64_t base = 3ull * tile_first[tile];
  const uint32_t L = tile_leaders[tile];
  uint32_t mine_cnt = 0, mine_bytes = 0;
  for (uint32_t i0 = 0; i0 < L;) {  // wave-uniform
    uint32_t m = 0;                  // entries listed
    for (; i0 < L && m + 64 <= kList; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool maybe = i < L && first[base + i] == kMaybeFirst;
      const uint64_t b = __ballot(maybe);
      if (maybe) list[w][m + (uint32_t)__popcll(b & lt)] = (uint16_t)i;  // a 32 KiB tile has fewer than 2^16 entries
      m += (uint32_t)__popcll(b);
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; j < m; j += 64) {
      const uint32_t i = list[w][j];
      const uint64_t mine = ent[base + i].x, r = ht[e_slot[base + i]].val;
      const uint8_t f = mine == r;
      if (!f) {
        const uint64_t lm = mine & ((1ull << kRepLenBits) - 1);
        if (lm != (r & ((1ull << kRepLenBits) - 1)) || diff_global_wide(text + (mine >> kRepLenBits), text + (r >> kRepLenBits), (uint32_t)lm)) *collision = 1;
      }
      first[base + i] = f;
      mine_cnt += f;
      if (f) mine_bytes += (uint32_t)(mine & ((1ull << kRepLenBits) - 1));
    }
    __builtin_amdgcn_wave_barrier();
  }
  mine_cnt = wave_reduce_u32(mine_cnt);
  mine_bytes = wave_reduce_u32(mine_bytes);  // a tile's terms are shorter than its 33 KiB window, or slow-path terms of under 16 MiB each: fits
  if (lane == 0) { tile_firsts[tile] = mine_cnt; tile_bytes[tile] = mine_bytes; }
}

// The representatives get their ids (their rank among the representatives, i.e. ids grow with the place of a
// term's first occurrence in the text); the slot now maps hash -> id.
__global__ void __launch_bounds__(64 * EN_TILES_PER_BLOCK)
k_entry_assign(const uint32_t* __restrict__ tile_first, const uint32_t* __restrict__ tile_leaders, unsigned ntiles,
               const ulonglong2* __restrict__ ent, uint32_t* __restrict__ e_slot, const uint8_t* __restrict__ first,
               const uint32_t* __restrict__ firsts_before /*[tile]: representatives in the tiles before it*/, DictSlot* __restrict__ ht,
               uint32_t* __restrict__ term_len, uint64_t* __restrict__ term_src) {
  const unsigned tile = blockIdx.x * EN_TILES_PER_BLOCK + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const uint64_t base = 3ull * tile_first[tile];
  const uint32_t L = tile_leaders[tile];
  const int lane = threadIdx.x & 63;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t run = firsts_before[tile];
  for (uint32_t i0 = 0; i0 < L; i0 += 64) {  // wave-uniform trip count: the ballot sees all 64 lanes
    const uint32_t i = i0 + lane;
    const bool f = i < L && first[base + i];
    const uint64_t b = __ballot(f);
    if (f) {
      const uint32_t id = run + (uint32_t)__popcll(b & lt);
      const uint64_t rep = ent[base + i].x;
      ht[e_slot[base + i]].val = id;
      e_slot[base + i] = id;  // a representative knows its id already: k_entry_to_id leaves it alone
      term_len[id] = (uint32_t)(rep & ((1ull << kRepLenBits) - 1));
      term_src[id] = rep >> kRepLenBits;
    }
    run += (uint32_t)__popcll(b);
  }
}

// entry -> id, in place of the entry's slot (the representatives got theirs from k_entry_assign)
__global__ void __launch_bounds__(64 * EN_TILES_PER_BLOCK)
k_entry_to_id(const uint32_t* __restrict__ tile_first, const uint32_t* __restrict__ tile_leaders, unsigned ntiles,
              const DictSlot* __restrict__ ht, const uint8_t* __restrict__ first, uint32_t* __restrict__ e_slot_then_id) {
  const unsigned tile = blockIdx.x * EN_TILES_PER_BLOCK + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const uint64_t base = 3ull * tile_first[tile];
  const uint32_t L = tile_leaders[tile];
  for (uint32_t i = threadIdx.x & 63; i < L; i += 64)
    if (!first[base + i]) e_slot_then_id[base + i] = (uint32_t)ht[e_slot_then_id[base + i]].val;
}

__global__ void __launch_bounds__(256) k_ht_clear(DictSlot* __restrict__ ht, uint64_t cap) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x)
    *reinterpret_cast<uint4*>(&ht[i]) = make_uint4(0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);
}

// Every occurrence claims / finds the slot of its hash and lowers the slot's representative to the
// smallest (offset, length) word, i.e. the first occurrence in the text: ids do not depend on timing.
// Occurrences are laid out field-major (all subjects, all predicates, all objects, each in text
// order), so a run of records with one subject is a run of equal hashes in neighbouring lanes: only the
// first lane of a run goes to memory, the rest of the run (later in the text, so never the minimum)
// takes its slot over a shuffle.
__global__ void __launch_bounds__(256)
k_dict_insert(const uint64_t* __restrict__ t_hash, const uint64_t* __restrict__ t_rep, uint64_t n_occ,
              DictSlot* __restrict__ ht, uint64_t mask, uint32_t* __restrict__ t_slot, uint32_t* __restrict__ overflow) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < n_occ;
  const uint64_t h = valid ? t_hash[i] : 0, mine = valid ? t_rep[i] : ~0ull;
  const uint64_t h_left = __shfl_up(h, 1, 64);
  const bool leader = valid && (lane == 0 || h_left != h);
  uint64_t slot = 0;
  if (leader) {
    uint64_t holder;
    slot = claim_slot(ht, mask, h, mine, holder, overflow);
  }
  const uint64_t leaders = __ballot(leader);
  const uint64_t at_or_below = leaders & (~0ull >> (63 - lane));
  const int src = at_or_below ? 63 - __clzll((long long)at_or_below) : lane;
  const uint32_t run_slot = __shfl((uint32_t)slot, src, 64);
  if (valid) t_slot[i] = run_slot;
}

// One occurrence in twelve is a first one: a lane looks at four flags at once and does the work of those set.
// (Sixteen per lane measured slower: too few lanes left to hide the dependent loads of the ones that have work.)
__global__ void __launch_bounds__(256)
k_dict_assign(const uint64_t* __restrict__ t_rep, const uint32_t* __restrict__ t_slot, const uint8_t* __restrict__ first,
              const uint32_t* __restrict__ first_scan, uint64_t n_occ, DictSlot* __restrict__ ht,
              uint32_t* __restrict__ term_len, uint64_t* __restrict__ term_src) {
  const uint64_t i0 = 4 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
  if (i0 >= n_occ) return;
  uint32_t m = 0;
  if (i0 + 4 <= n_occ) {
    m = *reinterpret_cast<const uint32_t*>(first + i0);
  } else {
    for (uint64_t i = i0; i < n_occ; ++i) m |= (uint32_t)(first[i] != 0) << (8 * (i - i0));
  }
  while (m) {
    const int bit = __ffs((int)m) - 1;  // flags are bytes holding 0 or 1
    m &= m - 1;
    const uint64_t i = i0 + (bit >> 3);
    const uint32_t id = first_scan[i];
    const uint64_t rep = t_rep[i];
    ht[t_slot[i]].val = id;  // the representative word has done its job: the slot now maps hash -> id
    term_len[id] = (uint32_t)(rep & ((1ull << kRepLenBits) - 1));
    term_src[id] = rep >> kRepLenBits;
  }
}

// Encoded records (s, p, o as term ids) + per-term facts that follow from the record kinds. Each fact has
// a byte array of its own, written with plain stores of 1: every writer of a byte stores the same value, so
// no atomics are needed (read-modify-write atomics are carried out lane by lane in the L2, and the records
// of one subject sit in neighbouring lanes). The check before each store only saves stores.
__global__ void __launch_bounds__(256)
k_encode(const uint32_t* __restrict__ t_entry, const uint32_t* __restrict__ e_id, const uint8_t* __restrict__ kind,
         uint32_t n_rec, uint32_t* __restrict__ S, uint32_t* __restrict__ P, uint32_t* __restrict__ O,
         uint8_t* __restrict__ is_vertex, uint8_t* __restrict__ has_edge, uint8_t* __restrict__ is_literal,
         uint8_t* __restrict__ is_pred) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  // an occurrence names its tile's entry for the term, the entry carries the id
  const uint32_t s = e_id[t_entry[r]], p = e_id[t_entry[(uint64_t)n_rec + r]], o = e_id[t_entry[2ull * n_rec + r]];
  S[r] = s; P[r] = p; O[r] = o;
  auto set = [](uint8_t* a, uint32_t id) { if (!a[id]) a[id] = 1; };
  set(is_pred, p);
  const uint8_t k = kind[r];
  set(is_vertex, s);
  if (k == 0) { set(has_edge, s); set(is_vertex, o); set(has_edge, o); }
  else if (k == 2) set(is_literal, o);
}

// The same for the fused tokeniser's layout: the records of tile t sit in slots [t * kSlotsPerTile, + records of t) and move
// to their dense places [rec_first[t], rec_first[t + 1]). One wave per tile.
__global__ void __launch_bounds__(64 * EN_TILES_PER_BLOCK)
k_encode_tiles(const uint32_t* __restrict__ rec_first, unsigned ntiles, const uint32_t* __restrict__ t_entry, uint64_t n_slots,
               const uint32_t* __restrict__ e_id, const uint8_t* __restrict__ kind_slots, uint32_t* __restrict__ S, uint32_t* __restrict__ P,
               uint32_t* __restrict__ O, uint8_t* __restrict__ kind, uint8_t* __restrict__ is_vertex, uint8_t* __restrict__ has_edge,
               uint8_t* __restrict__ is_literal, uint8_t* __restrict__ is_pred) {
  const unsigned tile = blockIdx.x * EN_TILES_PER_BLOCK + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const uint32_t r0 = rec_first[tile], n_here = rec_first[tile + 1] - r0;
  const uint64_t slot0 = (uint64_t)tile * kSlotsPerTile;
  auto set = [](uint8_t* a, uint32_t id) { if (!a[id]) a[id] = 1; };
  for (uint32_t lr = threadIdx.x & 63; lr < n_here; lr += 64) {
    const uint64_t at = slot0 + lr;
    const uint32_t s = e_id[t_entry[at]], p = e_id[t_entry[n_slots + at]], o = e_id[t_entry[2 * n_slots + at]];
    const uint32_t r = r0 + lr;
    S[r] = s; P[r] = p; O[r] = o;
    const uint8_t k = kind_slots[at];
    kind[r] = k;
    set(is_pred, p);
    set(is_vertex, s);
    if (k == 0) { set(has_edge, s); set(is_vertex, o); set(has_edge, o); }
    else if (k == 2) set(is_literal, o);
  }
}

__global__ void __launch_bounds__(256) k_iota_times(uint32_t* __restrict__ out, uint64_t n, uint32_t mult) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint32_t)i * mult;
}

__global__ void __launch_bounds__(256)
k_merge_flags(const uint8_t* __restrict__ is_vertex, const uint8_t* __restrict__ has_edge, const uint8_t* __restrict__ is_literal,
              uint32_t n_terms, uint8_t* __restrict__ flags) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_terms)
    flags[t] = (is_vertex[t] ? kFlagVertex : 0) | (has_edge[t] ? kFlagEdge : 0) | (is_literal[t] ? kFlagLiteral : 0);
}

// Copies term bytes from the input text into the id-ordered pool: four lanes per term, 16 bytes per lane and
// step (unaligned wide loads and stores), the last chunk overlapping the one before instead of a byte tail.
// Neighbouring lane groups write neighbouring pool ranges.
__global__ void __launch_bounds__(256)
k_pool_copy(const uint8_t* __restrict__ text, const uint64_t* __restrict__ term_src, const uint64_t* __restrict__ pool_off,
            uint64_t n_terms, uint8_t* __restrict__ pool) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t t = g >> 2;
  const uint32_t sub = (uint32_t)g & 3;
  if (t >= n_terms) return;
  const uint64_t b = pool_off[t], len = pool_off[t + 1] - b;
  const uint8_t* src = text + term_src[t];
  uint8_t* dst = pool + b;
  if (len >= 16) {
    for (uint64_t j = 16ull * sub; j < len; j += 64) {
      const uint64_t at = j + 16 <= len ? j : len - 16;
      uint4 v;
      __builtin_memcpy(&v, src + at, 16);
      __builtin_memcpy(dst + at, &v, 16);
    }
  } else if (sub == 0) {
    for (uint64_t j = 0; j < len; ++j) dst[j] = src[j];
  }
}

// Looks strings up: hash, probe, byte-compare against the pool.
__global__ void __launch_bounds__(64)
k_lookup(const uint8_t* __restrict__ q, const uint64_t* __restrict__ q_off, uint32_t nq, uint64_t seed,
         const DictSlot* __restrict__ ht, uint64_t mask,
         const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off, int64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint64_t b = q_off[i], e = q_off[i + 1];
  const uint64_t h = hash_range(q, b, e, seed);
  uint64_t slot = h & mask;
  int64_t res = -1;
  for (;;) {
    const uint64_t cur = ht[slot].key;
    if (cur == 0) break;
    if (cur == h) {
      const uint32_t id = (uint32_t)ht[slot].val;
      const uint64_t pb = pool_off[id], pl = pool_off[id + 1] - pb;
      bool same = pl == e - b;
      for (uint64_t j = 0; same && j < pl; ++j) same = pool[pb + j] == q[b + j];
      if (same) res = id;
      break;  // one slot per hash: a different string with this hash is not in the dictionary
    }
    slot = (slot + 1) & mask;
  }
  out[i] = res;
}

// Rebuilds the lookup table from the pool (dictionary loaded from disk).
__global__ void __launch_bounds__(256)
k_table_from_pool(const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off, uint64_t n_terms, uint64_t seed,
                  DictSlot* __restrict__ ht, uint64_t mask, uint32_t* __restrict__ collision) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const uint64_t h = hash_range(pool, pool_off[t], pool_off[t + 1], seed);
  uint64_t slot = h & mask;
  for (;;) {
    const uint64_t cur = atomicCAS((unsigned long long*)&ht[slot].key, 0ull, (unsigned long long)h);
    if (cur == 0) { ht[slot].val = t; break; }
    if (cur == h) { *collision = 1; break; }  // two pool strings are distinct by construction
    slot = (slot + 1) & mask;
  }
}

// ---- 4. class masks, vertex masks -----------------------------------------------------------------------
__global__ void k_set_class_masks(const int64_t* __restrict__ ids, const uint64_t* __restrict__ masks, uint32_t n,
                                  uint64_t* __restrict__ Pm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ids[i] >= 0) atomicOr((unsigned long long*)&Pm[ids[i]], (unsigned long long)masks[i]);
}

__global__ void __launch_bounds__(256)
k_vertex_masks(const uint32_t* __restrict__ S, const uint32_t* __restrict__ O, const uint8_t* __restrict__ kind, uint32_t n_rec,
               const uint64_t* __restrict__ Pm, uint64_t* __restrict__ M) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rec || kind[r] != 1) return;
  const uint64_t m = Pm[O[r]];
  if (m && (M[S[r]] & m) != m) atomicOr((unsigned long long*)&M[S[r]], (unsigned long long)m);
}

// pred_terms[pred_index[t]] = t for every term used as a predicate; also counts vertices
__global__ void __launch_bounds__(256)
k_list_preds(const uint8_t* __restrict__ is_pred, const uint32_t* __restrict__ pred_index, const uint8_t* __restrict__ flags,
             uint32_t n_terms, uint32_t* __restrict__ pred_terms, unsigned long long* __restrict__ n_vertices) {
  __shared__ uint32_t lds[8];
  uint32_t v = 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_terms; t += gridDim.x * blockDim.x) {
    if (is_pred[t]) pred_terms[pred_index[t]] = t;
    v += (flags[t] & kFlagVertex) ? 1u : 0u;
  }
  uint32_t total;
  (void)block_excl_scan_256<uint32_t>(v, lds, total);
  if (threadIdx.x == 0 && total) atomicAdd(n_vertices, (unsigned long long)total);
}

// ---- 5. sort keys, placement masks -----------------------------------------------------------------------
// K: the key word, uint64_t when predicate index and two term ids fit 64 bits (every LUBM size), u128 beyond
// (tens of thousands of predicates over hundreds of millions of terms).
template <typename K>
__global__ void __launch_bounds__(256)
k_pack_keys(const uint32_t* __restrict__ S, const uint32_t* __restrict__ P, const uint32_t* __restrict__ O, uint32_t n_rec,
            const uint32_t* __restrict__ pred_index, int bits_term, K* __restrict__ keys) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  keys[r] = ((K)pred_index[P[r]] << (2 * bits_term)) | ((K)S[r] << bits_term) | (K)O[r];
}

struct Unpacked { uint32_t pidx, s, o; };
template <typename K>
__device__ __forceinline__ Unpacked unpack(K k, int bits_term) {
  const K tm = ((K)1 << bits_term) - 1;
  return {(uint32_t)(k >> (2 * bits_term)), (uint32_t)((k >> bits_term) & tm), (uint32_t)(k & tm)};
}

// mask[i] = partitions of sorted triple i (0 for a duplicate of its predecessor); also counts
// distinct triples, distinct edges and node-index rows.
template <typename K>
__global__ void __launch_bounds__(256)
k_place(const K* __restrict__ keys, uint32_t n, int bits_term, uint32_t type_pidx, const uint8_t* __restrict__ flags,
        const uint64_t* __restrict__ Pm, const uint64_t* __restrict__ M, uint64_t owned /*partitions this rank keeps*/,
        uint64_t* __restrict__ mask, uint8_t* __restrict__ node_flag, unsigned long long* __restrict__ counters /*0 uniq, 1 edges, 2 rows over all partitions*/) {
  // Grid-stride: the counters are hit once per block at the end, and the grid is about a thousand blocks.
  // (Atomics of many blocks on one word are served one after another by a single L2 channel.)
  // A triple's mask hangs on three loads one after another (key, term flags, vertex masks), so a thread works on
  // four triples at a time and has the four chains in flight together; every load is predicated, none sits in a branch.
  uint32_t uq = 0, ed = 0;
  uint64_t rows = 0;  // a triple is one row in each of its partitions: the split's 32-bit positions must hold the sum
  const uint64_t stride = (uint64_t)gridDim.x * 1024;
  for (uint64_t b = (uint64_t)blockIdx.x * 1024 + threadIdx.x; b < n; b += stride) {
    K k[4], kp[4];
    bool fresh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t i = b + 256 * j;
      const bool in = i < n;
      k[j] = in ? keys[i] : K{};
      kp[j] = (in && i) ? keys[i - 1] : K{};
      fresh[j] = in && (i == 0 || kp[j] != k[j]);
    }
    Unpacked t[4];
    uint8_t fs[4], fo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[j] = unpack(k[j], bits_term);
      fs[j] = fresh[j] ? flags[t[j].s] : 0;
      fo[j] = fresh[j] ? flags[t[j].o] : 0;
    }
    uint64_t pm[4], ms[4], mo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool is_type = t[j].pidx == type_pidx, lit = (fo[j] & kFlagLiteral) != 0, edge = (fs[j] & kFlagEdge) != 0;
      pm[j] = (fresh[j] && is_type && edge) ? Pm[t[j].o] : 0;
      ms[j] = (fresh[j] && !is_type && (!lit || edge)) ? M[t[j].s] : 0;
      mo[j] = (fresh[j] && !is_type && !lit) ? M[t[j].o] : 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t i = b + 256 * j;
      if (i >= n) continue;
      const bool is_type = t[j].pidx == type_pidx;
      uint64_t m = is_type ? pm[j] : (ms[j] | mo[j]);
      bool nf = is_type && m != 0;
      if (fresh[j]) {
        // With several ranks a triple reaches every owner of one of its partitions: the owner of the lowest
        // one counts it and keeps its node-index row (one GPU: owned = all ones, so always). A triple that no
        // partition takes has been sent to one rank only.
        const bool mine = m == 0 || ((m & (~m + 1)) & owned) != 0;
        if (mine) {
          ++uq;
          if (!is_type && !(fo[j] & kFlagLiteral)) ++ed;
        }
        nf = nf && mine;
        m &= owned;
      }
      mask[i] = m;
      node_flag[i] = nf ? 1 : 0;
      rows += (uint64_t)__popcll(m);
    }
  }
  __shared__ uint32_t lds[8];
  uint32_t tu, te;
  (void)block_excl_scan_256<uint32_t>(uq, lds, tu);
  (void)block_excl_scan_256<uint32_t>(ed, lds, te);
  if (threadIdx.x == 0) {
    if (tu) atomicAdd(&counters[0], (unsigned long long)tu);
    if (te) atomicAdd(&counters[1], (unsigned long long)te);
  }
  __shared__ uint64_t lds64[8];
  uint64_t tr;
  (void)block_excl_scan_256<uint64_t>(rows, lds64, tr);
  if (threadIdx.x == 0 && tr) atomicAdd(&counters[2], (unsigned long long)tr);
}

// ---- 6. stable split by partition bit -----------------------------------------------------------------------
// counts[q][tile] = triples of the tile placed in partition q; after the scan of that matrix the
// scatter writes row i of partition q at counts[q][tile] + (rank of i among the tile's rows with bit q).
constexpr int SP_THREADS = 256;
constexpr int SP_TILE = 2048;  // each wave owns 512 consecutive triples

__global__ void __launch_bounds__(SP_THREADS)
k_split_count(const uint64_t* __restrict__ mask, uint32_t n, int nparts, uint32_t* __restrict__ counts, unsigned ntiles) {
  __shared__ uint32_t c[DX_MAX_PARTITIONS];
  if (threadIdx.x < DX_MAX_PARTITIONS) c[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * SP_TILE;
  for (int j = 0; j < SP_TILE / SP_THREADS; ++j) {
    const uint32_t i = base + j * SP_THREADS + threadIdx.x;
    const uint64_t m = i < n ? mask[i] : 0;
    for (int q = 0; q < nparts; ++q) {
      const uint64_t b = __ballot((m >> q) & 1);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(&c[q], (uint32_t)__popcll(b));
    }
  }
  __syncthreads();
  if (threadIdx.x < nparts) counts[(size_t)threadIdx.x * ntiles + blockIdx.x] = c[threadIdx.x];
}

template <typename K>
__global__ void __launch_bounds__(SP_THREADS)
k_split_scatter(const K* __restrict__ keys, const uint64_t* __restrict__ mask, uint32_t n, int nparts, int bits_term,
                const uint32_t* __restrict__ offs, unsigned ntiles, int64_t* __restrict__ row_s, int64_t* __restrict__ row_o,
                uint32_t* __restrict__ row_p) {
  __shared__ uint32_t wcnt[4][DX_MAX_PARTITIONS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t wbase = blockIdx.x * SP_TILE + w * (SP_TILE / 4);
  // pass 1: per-wave totals per partition
  uint32_t tot = 0;  // lane q accumulates partition q (nparts <= 64)
  for (int j = 0; j < SP_TILE / 4 / 64; ++j) {
    const uint32_t i = wbase + j * 64 + lane;
    const uint64_t m = i < n ? mask[i] : 0;
    for (int q = 0; q < nparts; ++q) {
      const uint32_t c = (uint32_t)__popcll(__ballot((m >> q) & 1));
      if (lane == q) tot += c;
    }
  }
  if (lane < DX_MAX_PARTITIONS) wcnt[w][lane] = tot;
  __syncthreads();
  if (threadIdx.x < nparts) {
    uint32_t run = offs[(size_t)threadIdx.x * ntiles + blockIdx.x];
    for (int i = 0; i < 4; ++i) { const uint32_t c = wcnt[i][threadIdx.x]; wcnt[i][threadIdx.x] = run; run += c; }
  }
  __syncthreads();
  // pass 2: write
  uint32_t basev = lane < DX_MAX_PARTITIONS ? wcnt[w][lane] : 0;  // lane q holds the running base of partition q
  for (int j = 0; j < SP_TILE / 4 / 64; ++j) {
    const uint32_t i = wbase + j * 64 + lane;
    const uint64_t m = i < n ? mask[i] : 0;
    Unpacked t{0, 0, 0};
    if (m) t = unpack(keys[i], bits_term);
    for (int q = 0; q < nparts; ++q) {
      const bool bit = (m >> q) & 1;
      const uint64_t b = __ballot(bit);
      if (!b) continue;  // wave-uniform
      const uint32_t qbase = __shfl(basev, q, 64);
      if (bit) {
        const uint32_t pos = qbase + (uint32_t)__popcll(b & lt);
        row_s[pos] = (int64_t)t.s;
        row_o[pos] = (int64_t)t.o;
        row_p[pos] = t.pidx;
      }
      if (lane == q) basev += (uint32_t)__popcll(b);
    }
  }
}

// First-level partition of row i (rows are partition-major). The search starts from the partition of the
// block's first row, which is the same for every lane and therefore runs on the scalar unit; a lane then
// moves on only if its block straddles a partition boundary.
__device__ __forceinline__ int partition_of_row(const uint32_t* __restrict__ part_base, int nparts, uint32_t i) {
  const uint32_t first = blockIdx.x * blockDim.x;
  int q = 0;
  while (q + 1 < nparts && first >= part_base[q + 1]) ++q;
  while (q + 1 < nparts && i >= part_base[q + 1]) ++q;
  return q;
}

// table_start[q * np + pidx] = first row of that table (rows are sorted by pidx inside a partition)
__global__ void __launch_bounds__(256)
k_table_starts(const uint32_t* __restrict__ row_p, const uint32_t* __restrict__ part_base /*nparts+1*/, int nparts, uint32_t np,
               uint32_t* __restrict__ start) {
  // four consecutive rows per lane (one 16-byte load), grid-stride: a block per 256 rows would spend its time being launched.
  // A table starts where the predicate changes or a partition begins. The second kind is written by the first lanes
  // of block 0; the first kind is met a few hundred times among millions of rows, and only there is the row's
  // partition looked up.
  const uint32_t total = part_base[nparts];
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)nparts) {
    const uint32_t b = part_base[threadIdx.x];
    if (b < part_base[threadIdx.x + 1]) start[(size_t)threadIdx.x * np + row_p[b]] = b;
  }
  for (uint32_t i0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x); i0 < total; i0 += 4 * gridDim.x * blockDim.x) {
    uint32_t p[4];
    if (i0 + 4 <= total) {
      const uint4 v = *reinterpret_cast<const uint4*>(row_p + i0);
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    } else {
      for (int k = 0; k < 4; ++k) p[k] = i0 + k < total ? row_p[i0 + k] : 0;
    }
    uint32_t prev = i0 ? row_p[i0 - 1] : p[0];
    for (int k = 0; k < 4 && i0 + k < total; ++k) {
      if (prev != p[k]) {
        const uint32_t i = i0 + k;
        int q = 0;
        while (q + 1 < nparts && i >= part_base[q + 1]) ++q;
        start[(size_t)q * np + p[k]] = i;
      }
      prev = p[k];
    }
  }
}

// ---- 7. node index, VP index ------------------------------------------------------------------------------
template <typename K>
__global__ void __launch_bounds__(256)
k_node_index(const K* __restrict__ keys, const uint8_t* __restrict__ node_flag, const uint32_t* __restrict__ pos, uint32_t n,
             int bits_term, int64_t* __restrict__ ns, int64_t* __restrict__ no, uint8_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !node_flag[i]) return;
  const Unpacked t = unpack(keys[i], bits_term);
  ns[pos[i]] = t.s;
  no[pos[i]] = t.o;
  // plain byte store: nothing else writes the flags while this kernel runs, and every writer of this byte
  // (the other node-index rows of the same uri) stores the same value
  const uint8_t f = flags[t.s];
  if (!(f & kFlagNode)) flags[t.s] = f | kFlagNode;
}

// The VP index of a partition (DAP.scala:479-497) is the set of (uri, role, predicate) with uri a typed
// uri (one that has a node-index row) standing as subject ('s') or object ('o') of a row of that predicate,
// plus (class, 'o', rdf:type) for every type row. Rows are sorted by (partition, predicate, s, o), so:
//  - 's' entries are distinct as soon as a row whose (predicate, s) equals the previous row's is skipped,
//    and they come out in (partition, predicate, s) order: no sort;
//  - 'o' entries of non-type rows need a sort to drop duplicates;
//  - the classes of type rows are few and repeat millions of times: a type row (s, C) lies in every
//    partition of the class index entry of C, so marking C as used (a plain byte store) is enough, and the
//    entries (C, 'o', rdf:type) of the partitions of C are written afterwards.
// Pass 1 writes the per-row decisions es/eo and counts them per block of 1024 rows, pass 2 (after the scan of those
// counts) writes the entries.
__global__ void __launch_bounds__(256)
k_vpi_flags(const int64_t* __restrict__ row_s, const int64_t* __restrict__ row_o, const uint32_t* __restrict__ row_p,
            const uint32_t* __restrict__ part_base, int nparts, uint32_t type_pidx, const uint8_t* __restrict__ flags,
            uint8_t* __restrict__ es_out, uint8_t* __restrict__ eo_out, uint8_t* __restrict__ class_used,
            uint32_t* __restrict__ block_s /*[block]: 's' entries of the block's 1024 rows*/, uint32_t* __restrict__ block_o,
            int drop_repeats /*partition, predicate index and object fit the set's 64-bit word: 6 + 26 + 32 bits*/) {
  // four consecutive rows per lane: 16- and 32-byte loads, one 4-byte store per output instead of four 1-byte ones
  __shared__ uint32_t lds[8];
  // 'o' entries repeat at short range (the students of a department follow one another and share its courses): a
  // block keeps one of each (partition, predicate, object) among its 1024 rows, whichever row enters this set
  // first, and the sort that makes the entries distinct has that much less to carry.
  constexpr uint32_t kSeen = 2048;  // at most half full
  __shared__ unsigned long long seen[kSeen];
  for (uint32_t j = threadIdx.x; j < kSeen; j += 256) seen[j] = ~0ull;
  __syncthreads();
  const uint32_t total = part_base[nparts];
  const uint32_t i0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  uint32_t es4 = 0, eo4 = 0;
  if (i0 < total) {
  uint32_t s[4], o[4], p[4];
  const bool whole = i0 + 4 <= total;
  if (whole) {
    const longlong2 s01 = *reinterpret_cast<const longlong2*>(row_s + i0), s23 = *reinterpret_cast<const longlong2*>(row_s + i0 + 2);
    const longlong2 o01 = *reinterpret_cast<const longlong2*>(row_o + i0), o23 = *reinterpret_cast<const longlong2*>(row_o + i0 + 2);
    const uint4 pv = *reinterpret_cast<const uint4*>(row_p + i0);
    s[0] = (uint32_t)s01.x; s[1] = (uint32_t)s01.y; s[2] = (uint32_t)s23.x; s[3] = (uint32_t)s23.y;
    o[0] = (uint32_t)o01.x; o[1] = (uint32_t)o01.y; o[2] = (uint32_t)o23.x; o[3] = (uint32_t)o23.y;
    p[0] = pv.x; p[1] = pv.y; p[2] = pv.z; p[3] = pv.w;
  } else {
    for (int k = 0; k < 4; ++k) {
      const bool in = i0 + k < total;
      s[k] = in ? (uint32_t)row_s[i0 + k] : 0;
      o[k] = in ? (uint32_t)row_o[i0 + k] : 0;
      p[k] = in ? row_p[i0 + k] : 0;
    }
  }
  uint32_t prev_s = i0 ? (uint32_t)row_s[i0 - 1] : 0, prev_p = i0 ? row_p[i0 - 1] : 0;
  // the partition of the block's first row is the same for every lane (scalar unit); a lane moves on from there
  int q = 0;
  {
    const uint32_t first = 4 * blockIdx.x * blockDim.x;
    while (q + 1 < nparts && first >= part_base[q + 1]) ++q;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = i0 + k;
    if (i < total) {
      while (q + 1 < nparts && i >= part_base[q + 1]) ++q;
      bool es = (flags[s[k]] & kFlagNode) != 0;
      if (es && i > part_base[q] && prev_p == p[k] && prev_s == s[k]) es = false;
      bool eo = false;
      if (p[k] == type_pidx) {
        if (!class_used[o[k]]) class_used[o[k]] = 1;
      } else {
        eo = (flags[o[k]] & kFlagNode) != 0;
        if (eo && drop_repeats) {
          const unsigned long long key = ((unsigned long long)q << 58) | ((unsigned long long)p[k] << 32) | o[k];
          for (uint32_t at = (uint32_t)(mix64(key) >> 40) & (kSeen - 1);; at = (at + 1) & (kSeen - 1)) {
            const unsigned long long was = atomicCAS(&seen[at], ~0ull, key);
            if (was == ~0ull) break;
            if (was == key) { eo = false; break; }
          }
        }
      }
      es4 |= (uint32_t)es << (8 * k);
      eo4 |= (uint32_t)eo << (8 * k);
    }
    prev_s = s[k];
    prev_p = p[k];
  }
  if (whole) {
    *reinterpret_cast<uint32_t*>(es_out + i0) = es4;
    *reinterpret_cast<uint32_t*>(eo_out + i0) = eo4;
  } else {
    for (int k = 0; k < 4 && i0 + k < total; ++k) {
      es_out[i0 + k] = (uint8_t)(es4 >> (8 * k));
      eo_out[i0 + k] = (uint8_t)(eo4 >> (8 * k));
    }
  }
  }
  // the flags are bytes holding 0 or 1: a multiply adds the four of a word up in its top byte
  uint32_t ts, to;
  (void)block_excl_scan_256<uint32_t>((es4 * 0x01010101u) >> 24, lds, ts);
  (void)block_excl_scan_256<uint32_t>((eo4 * 0x01010101u) >> 24, lds, to);
  if (threadIdx.x == 0) { block_s[blockIdx.x] = ts; block_o[blockIdx.x] = to; }
}

// Pass 2: the entries' places follow from the scanned block counts and a scan inside the block (same 1024 rows per
// block, four per lane, as pass 1). The block's entries are put together in LDS and leave it in order, neighbouring
// lanes writing neighbouring entries.
template <typename K>  // K: the object entries' word, 32 bits when partition | term | role | predicate fit
__global__ void __launch_bounds__(256)
k_vpi_write(const int64_t* __restrict__ row_s, const int64_t* __restrict__ row_o, const uint32_t* __restrict__ row_p,
            const uint32_t* __restrict__ part_base, int nparts, int bits_term, int bits_pred, const uint8_t* __restrict__ es,
            const uint8_t* __restrict__ eo, const uint32_t* __restrict__ block_s /*scanned*/, const uint32_t* __restrict__ block_o,
            uint64_t* __restrict__ out_s, K* __restrict__ out_o) {
  __shared__ uint32_t lds[8];
  __shared__ uint64_t stage_s[1024];
  __shared__ K stage_o[1024];
  const uint32_t total = part_base[nparts];
  const uint32_t i0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  uint32_t es4 = 0, eo4 = 0;
  const bool whole = i0 + 4 <= total;
  if (whole) {
    es4 = *reinterpret_cast<const uint32_t*>(es + i0);
    eo4 = *reinterpret_cast<const uint32_t*>(eo + i0);
  } else {
    for (int k = 0; k < 4 && i0 + k < total; ++k) {
      es4 |= (uint32_t)es[i0 + k] << (8 * k);
      eo4 |= (uint32_t)eo[i0 + k] << (8 * k);
    }
  }
  uint32_t ts, to;
  uint32_t ps = block_excl_scan_256<uint32_t>((es4 * 0x01010101u) >> 24, lds, ts);
  uint32_t po = block_excl_scan_256<uint32_t>((eo4 * 0x01010101u) >> 24, lds, to);
  if (es4 | eo4) {
    uint32_t s[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0}, p[4];
    if (whole) {
      if (es4) {
        const longlong2 s01 = *reinterpret_cast<const longlong2*>(row_s + i0), s23 = *reinterpret_cast<const longlong2*>(row_s + i0 + 2);
        s[0] = (uint32_t)s01.x; s[1] = (uint32_t)s01.y; s[2] = (uint32_t)s23.x; s[3] = (uint32_t)s23.y;
      }
      if (eo4) {
        const longlong2 o01 = *reinterpret_cast<const longlong2*>(row_o + i0), o23 = *reinterpret_cast<const longlong2*>(row_o + i0 + 2);
        o[0] = (uint32_t)o01.x; o[1] = (uint32_t)o01.y; o[2] = (uint32_t)o23.x; o[3] = (uint32_t)o23.y;
      }
      const uint4 pv = *reinterpret_cast<const uint4*>(row_p + i0);
      p[0] = pv.x; p[1] = pv.y; p[2] = pv.z; p[3] = pv.w;
    } else {
      for (int k = 0; k < 4; ++k) {
        const bool in = i0 + k < total;
        s[k] = in ? (uint32_t)row_s[i0 + k] : 0;
        o[k] = in ? (uint32_t)row_o[i0 + k] : 0;
        p[k] = in ? row_p[i0 + k] : 0;
      }
    }
    int q = partition_of_row(part_base, nparts, i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool a = (es4 >> (8 * k)) & 1, b = (eo4 >> (8 * k)) & 1;
      if (!a && !b) continue;
      const uint32_t i = i0 + k;
      while (q + 1 < nparts && i >= part_base[q + 1]) ++q;
      const uint64_t qq = (uint64_t)q << (bits_term + 1 + bits_pred);
      if (a) stage_s[ps++] = qq | ((uint64_t)s[k] << (1 + bits_pred)) | p[k];
      if (b) stage_o[po++] = (K)(qq | ((uint64_t)o[k] << (1 + bits_pred)) | (1ull << bits_pred) | p[k]);
    }
  }
  __syncthreads();
  const uint32_t bs = block_s[blockIdx.x], bo = block_o[blockIdx.x];
  for (uint32_t j = threadIdx.x; j < ts; j += 256) out_s[bs + j] = stage_s[j];
  for (uint32_t j = threadIdx.x; j < to; j += 256) out_o[bo + j] = stage_o[j];
}

// counts (out == nullptr) or appends the (class, 'o', rdf:type) entries of the classes in use
template <typename K>
__global__ void __launch_bounds__(256)
k_vpi_type_entries(const uint8_t* __restrict__ class_used, const uint64_t* __restrict__ Pm, uint64_t owned /*partitions this rank keeps*/,
                   uint32_t n_terms, uint32_t type_pidx, int bits_term, int bits_pred, unsigned long long* __restrict__ counter,
                   K* __restrict__ out) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_terms; t += gridDim.x * blockDim.x) {
    if (!class_used[t]) continue;  // all but the class terms
    unsigned long long m = Pm[t] & owned;
    if (!m) continue;
    const unsigned long long at = atomicAdd(counter, (unsigned long long)__popcll(m));
    if (!out) continue;
    for (uint32_t k = 0; m; ++k, m &= m - 1) {
      const uint64_t q = (uint64_t)(__ffsll((long long)m) - 1);
      out[at + k] = (K)((q << (bits_term + 1 + bits_pred)) | ((uint64_t)t << (1 + bits_pred)) | (1ull << bits_pred) | type_pidx);
    }
  }
}

// base[q] = place of the first 's' entry at or after row part_base[q] for q < nparts (an empty partition takes the next
// one's), base[nparts] = total
__global__ void k_vpi_s_bases(const uint32_t* __restrict__ block_s /*scanned*/, const uint8_t* __restrict__ es,
                              const uint32_t* __restrict__ part_base, int nparts, uint32_t n_rows, const uint32_t* __restrict__ total,
                              uint32_t* __restrict__ base) {
  // one block of 256 threads per partition: the flags between the start of the row's block of 1024 and the row, four per thread
  __shared__ uint32_t lds[8];
  const int q = blockIdx.x;
  const uint32_t r = part_base[q];
  if (q == nparts || r >= n_rows) {  // block-uniform
    if (threadIdx.x == 0) base[q] = *total;
    return;
  }
  const uint32_t first = r & ~1023u;
  uint32_t mine = 0;
  for (uint32_t i = first + 4 * threadIdx.x; i < first + 4 * threadIdx.x + 4 && i < r; ++i) mine += es[i];
  uint32_t sum;
  (void)block_excl_scan_256<uint32_t>(mine, lds, sum);
  if (threadIdx.x == 0) base[q] = block_s[r / 1024] + sum;
}

template <typename K>
__global__ void __launch_bounds__(256)
k_flag_distinct(const K* __restrict__ keys, uint32_t n, uint8_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
template <typename K>
__global__ void __launch_bounds__(256)
k_compact_u64(const K* __restrict__ in, const uint8_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n,
              uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) out[pos[i]] = in[i];
}
// Distinct keys of a sorted array without a flag and a position per key: blocks of 1024 keys count their distinct keys
// (a key differs from the one before it), the counts are scanned, and the second kernel finds the places inside
// the block again, puts the block's distinct keys together in LDS and writes them out side by side.
template <typename K>
__device__ __forceinline__ uint32_t distinct_flags4(const K* __restrict__ keys, uint32_t n, uint32_t i0, K (&k)[4]) {
  uint32_t f = 0;
  if (i0 >= n) return 0;
  K prev = i0 ? keys[i0 - 1] : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (i0 + j < n) {
      k[j] = keys[i0 + j];
      if (i0 + j == 0 || k[j] != prev) f |= 1u << j;
      prev = k[j];
    }
  }
  return f;
}
template <typename K>
__global__ void __launch_bounds__(256)
k_distinct_count(const K* __restrict__ keys, uint32_t n, uint32_t* __restrict__ block_cnt) {
  __shared__ uint32_t lds[8];
  K k[4];
  const uint32_t f = distinct_flags4(keys, n, 4 * (blockIdx.x * blockDim.x + threadIdx.x), k);
  uint32_t total;
  (void)block_excl_scan_256<uint32_t>(__popc(f), lds, total);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}
template <typename K>
__global__ void __launch_bounds__(256)
k_distinct_compact(const K* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ block_base /*scanned counts*/, uint64_t* __restrict__ out) {
  __shared__ uint32_t lds[8];
  __shared__ K stage[1024];
  K k[4];
  const uint32_t f = distinct_flags4(keys, n, 4 * (blockIdx.x * blockDim.x + threadIdx.x), k);
  uint32_t total;
  uint32_t at = block_excl_scan_256<uint32_t>(__popc(f), lds, total);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (f & (1u << j)) stage[at++] = k[j];
  __syncthreads();
  const uint32_t base = block_base[blockIdx.x];
  for (uint32_t j = threadIdx.x; j < total; j += 256) out[base + j] = (uint64_t)stage[j];
}

// base[q] = first index whose partition field is >= q (keys sorted)
__global__ void k_vpi_bases(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ n_keys, int shift, int nparts,
                            uint32_t* __restrict__ base) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > nparts) return;
  uint32_t lo = 0, hi = *n_keys;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((keys[mid] >> shift) < (uint64_t)q) lo = mid + 1; else hi = mid;
  }
  base[q] = lo;
}

// ---- several GPUs: one dictionary for all ranks, triples routed to the owners of their partitions -----------
// The ranks' local term pools laid one after another are treated as a text of E entries: entry e of rank r is
// bytes [base[r] + off_r[j], base[r] + off_r[j+1]). The dictionary kernels above then give every distinct
// string one id, in order of first appearance in that concatenation: the same ids on every rank.
__global__ void __launch_bounds__(256)
k_entry_spans(const uint8_t* __restrict__ gpool, const uint64_t* __restrict__ goff /*each rank's n_r + 1 offsets, concatenated*/,
              const uint64_t* __restrict__ entry_base /*[W+1] first entry of rank r*/, const uint64_t* __restrict__ byte_base /*[W]*/,
              int world, uint64_t n_entries, uint64_t seed, uint64_t* __restrict__ e_rep, uint64_t* __restrict__ e_hash) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  int r = 0;
  while (r + 1 < world && e >= entry_base[r + 1]) ++r;
  const uint64_t j = e - entry_base[r];
  const uint64_t* off = goff + entry_base[r] + r;  // rank r's offsets start after r earlier (n + 1)-arrays
  const uint64_t b = byte_base[r] + off[j], len = off[j + 1] - off[j];
  e_rep[e] = (b << kRepLenBits) | len;
  e_hash[e] = hash_range(gpool, b, b + len, seed);
}

// first[e] = 1 when entry e is its slot's representative; the others are compared with it byte by byte
__global__ void __launch_bounds__(256)
k_entry_verify(const uint8_t* __restrict__ gpool, const uint64_t* __restrict__ e_rep, const uint32_t* __restrict__ e_slot, uint64_t n_entries,
               const DictSlot* __restrict__ ht, uint8_t* __restrict__ first, uint32_t* __restrict__ collision) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const uint64_t mine = e_rep[e], r = ht[e_slot[e]].val;
  first[e] = mine == r;
  if (mine == r) return;
  const uint64_t lm = mine & ((1ull << kRepLenBits) - 1);
  if (lm != (r & ((1ull << kRepLenBits) - 1)) || diff_global(gpool + (mine >> kRepLenBits), gpool + (r >> kRepLenBits), lm)) *collision = 1;
}

__global__ void __launch_bounds__(256)
k_entry_ids(const uint32_t* __restrict__ e_slot, const DictSlot* __restrict__ ht, uint64_t n_entries, uint32_t* __restrict__ gid) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_entries) gid[e] = (uint32_t)ht[e_slot[e]].val;
}

__global__ void __launch_bounds__(256)
k_remap(uint32_t* __restrict__ ids, const uint32_t* __restrict__ map, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] = map[ids[i]];
}

// the per-term facts of k_encode, from records that already carry their final ids
__global__ void __launch_bounds__(256)
k_facts(const uint32_t* __restrict__ S, const uint32_t* __restrict__ P, const uint32_t* __restrict__ O, const uint8_t* __restrict__ kind,
        uint32_t n_rec, uint8_t* __restrict__ is_vertex, uint8_t* __restrict__ has_edge, uint8_t* __restrict__ is_literal,
        uint8_t* __restrict__ is_pred) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  const uint32_t s = S[r], p = P[r], o = O[r];
  auto set = [](uint8_t* a, uint32_t id) { if (!a[id]) a[id] = 1; };
  set(is_pred, p);
  const uint8_t k = kind[r];
  set(is_vertex, s);
  if (k == 0) { set(has_edge, s); set(is_vertex, o); set(has_edge, o); }
  else if (k == 2) set(is_literal, o);
}

__global__ void __launch_bounds__(256) k_nonzero_u8(uint8_t* __restrict__ a, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = a[i] ? 1 : 0;
}

// M[t] = OR over the ranks' partial masks (gathered one rank after another)
__global__ void __launch_bounds__(256)
k_or_gathered(const uint64_t* __restrict__ gathered, int world, uint64_t n, uint64_t* __restrict__ M) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t m = 0;
  for (int r = 0; r < world; ++r) m |= gathered[(uint64_t)r * n + i];
  M[i] = m;
}

// Partitions of a triple (the closed form of k_place, without the duplicate test) folded into the set of
// ranks that own one of them: bit d of the result = rank d holds a partition q of the triple, q % world == d.
__device__ __forceinline__ uint64_t placement(const Unpacked& t, uint32_t type_pidx, const uint8_t* __restrict__ flags,
                                              const uint64_t* __restrict__ Pm, const uint64_t* __restrict__ M) {
  if (t.pidx == type_pidx) return (flags[t.s] & kFlagEdge) ? Pm[t.o] : 0;
  if (flags[t.o] & kFlagLiteral) return (flags[t.s] & kFlagEdge) ? M[t.s] : 0;
  return M[t.s] | M[t.o];
}
__device__ __forceinline__ uint64_t owners_of(uint64_t m, int world) {
  uint64_t d = 0;
  for (; m; m &= m - 1) d |= 1ull << ((__ffsll((long long)m) - 1) % world);
  return d;
}
template <typename K>
__global__ void __launch_bounds__(256)
k_route_count(const K* __restrict__ keys, uint32_t n, int bits_term, uint32_t type_pidx, const uint8_t* __restrict__ flags,
              const uint64_t* __restrict__ Pm, const uint64_t* __restrict__ M, int world, uint64_t* __restrict__ owners,
              unsigned long long* __restrict__ counts) {
  __shared__ uint32_t c[DX_MAX_PARTITIONS];
  if (threadIdx.x < DX_MAX_PARTITIONS) c[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const K key = keys[i];
    uint64_t d = owners_of(placement(unpack(key, bits_term), type_pidx, flags, Pm, M), world);
    // a triple that no partition takes still counts among the distinct triples (and edges) of the input: it goes to
    // one rank, chosen by its own bits, where its copies from other ranks' texts meet it
    if (!d) d = 1ull << (mix64((uint64_t)key ^ (uint64_t)(key >> 32 >> 32)) % (uint64_t)world);
    owners[i] = d;
    for (uint64_t x = d; x; x &= x - 1) atomicAdd(&c[__ffsll((long long)x) - 1], 1u);
  }
  __syncthreads();
  if (threadIdx.x < world && c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)c[threadIdx.x]);
}
template <typename K>
__global__ void __launch_bounds__(256)
k_route_scatter(const K* __restrict__ keys, const uint64_t* __restrict__ owners, uint32_t n, int world,
                unsigned long long* __restrict__ cursor, K* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t rounds = (n + stride - 1) / stride;
  for (uint32_t rd = 0; rd < rounds; ++rd) {
    const uint32_t i = rd * stride + blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t d = i < n ? owners[i] : 0;
    const K k = i < n ? keys[i] : (K)0;
    for (int p = 0; p < world; ++p) {  // wave-uniform
      const bool go = (d >> p) & 1;
      const uint64_t b = __ballot(go);
      if (!b) continue;
      const int leader = __ffsll((unsigned long long)b) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(&cursor[p], (unsigned long long)__popcll(b));
      base = __shfl(base, leader, 64);
      if (go) out[base + __popcll(b & lt)] = k;
    }
  }
}

// the node bit of the flags as a byte array and back (summed over the ranks in between)
__global__ void __launch_bounds__(256) k_node_bit_out(const uint8_t* __restrict__ flags, uint64_t n, uint8_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (flags[i] & kFlagNode) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_node_bit_in(const uint8_t* __restrict__ in, uint64_t n, uint8_t* __restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && in[i]) flags[i] |= kFlagNode;
}

// src[i] / len[i] = where term ids[i] starts in the pool and how long it is
__global__ void __launch_bounds__(256)
k_term_spans(const uint32_t* __restrict__ ids, uint32_t n, const uint64_t* __restrict__ pool_off, uint64_t* __restrict__ src,
             uint32_t* __restrict__ len) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t b = pool_off[ids[i]];
  src[i] = b;
  len[i] = (uint32_t)(pool_off[ids[i] + 1] - b);
}

int bits_for(uint64_t n) {  // bits needed for values in [0, n)
  int b = 1;
  while ((1ull << b) < n) ++b;
  return b;
}
inline unsigned blocks(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per ? (n + per - 1) / per : 1); }

// event i of the context's pool, created on first use (creating and destroying a dozen events per call costs
// more host time than some of the stages they time)
hipEvent_t pooled_event(dx_ctx* ctx, size_t i) {
  while (ctx->part_events.size() <= i) {
    hipEvent_t e;
    DX_HIP(hipEventCreate(&e));
    ctx->part_events.push_back(e);
  }
  return ctx->part_events[i];
}
// out[i] = in[i * stride]
// one block adds up a short array (the tiles' entry counts)
__global__ void __launch_bounds__(1024) k_sum_u32(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
  __shared__ uint32_t lds[16];
  uint32_t v = 0;
  for (uint32_t i = 4 * threadIdx.x; i + 4 <= n; i += 4096) { const uint4 q = *reinterpret_cast<const uint4*>(in + i); v += q.x + q.y + q.z + q.w; }
  if (threadIdx.x == 0) for (uint32_t i = n & ~3u; i < n; ++i) v += in[i];
  v = wave_reduce_u32(v);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += lds[w]; *out = t; }
}

// *out += the sum of in[0, n): each block adds up a slice and contributes it with one atomic (*out starts at zero)
__global__ void __launch_bounds__(1024) k_sum_u32_u64(const uint32_t* __restrict__ in, uint32_t n, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long lds[16];
  unsigned long long v = 0;
  for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) v += in[i];
  v = wave_reduce_u64(v);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < 16; ++w) t += lds[w];
    if (t) atomicAdd(out, t);
  }
}

__global__ void k_pick_strided(const uint32_t* __restrict__ in, size_t stride, int n, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(size_t)i * stride];
}

struct StageTimer {
  dx_ctx* ctx;
  std::vector<hipEvent_t> ev;
  explicit StageTimer(dx_ctx* c) : ctx(c) { mark(); }
  void mark() { hipEvent_t e = pooled_event(ctx, 4 + ev.size()); DX_HIP(hipEventRecord(e, ctx->stream)); ev.push_back(e); }
};

}  // namespace

// Stages 10 and 11 are the two big kernels inside "dictionary", timed on their own.
const char* const kStageNames[12] = {"line_scan", "tokenise_hash", "dictionary", "string_pool", "vertex_masks", "sort_triples",
                                     "place", "split_partitions", "node_index", "vp_index", "dict_clear_kernel", "dict_claim_kernel"};

// One block writes a batch of small device results into the pinned host block and then the batch's number into the
// pinned word the host watches (HostFetch, batched mode).
__global__ void __launch_bounds__(256)
k_fetch_batch(FetchBatch b, uint8_t* __restrict__ host, unsigned long long* __restrict__ host_seq, unsigned long long seq) {
  for (int k = 0; k < b.n; ++k) {
    const uint8_t* src = static_cast<const uint8_t*>(b.src[k]);
    uint8_t* dst = host + b.off[k];
    const uint32_t bytes = b.bytes[k];
    if (((reinterpret_cast<uintptr_t>(src) | bytes) & 3) == 0) {  // dst offsets are multiples of 16
      for (uint32_t i = threadIdx.x; i < bytes / 4; i += 256)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(dst) + i, reinterpret_cast<const uint32_t*>(src)[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      for (uint32_t i = threadIdx.x; i < bytes; i += 256)
        __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_fetch_batch(dx_ctx* ctx, const FetchBatch& b, unsigned long long seq) {
  hipLaunchKernelGGL(k_fetch_batch, dim3(1), dim3(256), 0, ctx->stream, b, ctx->part_host_dev, ctx->fetch_seq_dev, seq);
}

// Watches the pinned word for a while, then waits through the stream (which also reports a failed kernel).
void wait_fetch_seq(dx_ctx* ctx, unsigned long long seq) {
  const auto t0 = std::chrono::steady_clock::now();
  const volatile unsigned long long* w = ctx->fetch_seq;
  for (uint32_t i = 0;; ++i) {
    if (*w >= seq) { std::atomic_thread_fence(std::memory_order_acquire); return; }
    __builtin_ia32_pause();
    if ((i & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
  }
  DX_HIP(hipStreamSynchronize(ctx->stream));
  DX_HIP(hipGetLastError());
  if (*w < seq) throw DxError(DX_E_HIP, "the partitioner's results did not arrive on the host");
}

void dict_build_table(dx_ctx* ctx) {
  Dictionary& d = ctx->dict;
  hipStream_t s = ctx->stream;
  uint64_t cap = 1024;
  while (cap < d.n_terms * 2) cap <<= 1;
  DevBuf<uint32_t> coll(ctx->pool, 1);
  for (int attempt = 0; attempt < 8; ++attempt) {
    d.ht = DevBuf<DictSlot>(ctx->pool, cap);
    d.ht_mask = cap - 1;
    DX_HIP(hipMemsetAsync(d.ht.p, 0, cap * sizeof(DictSlot), s));
    DX_HIP(hipMemsetAsync(coll.p, 0, 4, s));
    if (d.n_terms)
      hipLaunchKernelGGL(k_table_from_pool, dim3(blocks(d.n_terms)), dim3(256), 0, s, d.pool.p, d.pool_off.p, d.n_terms, d.seed,
                         d.ht.p, d.ht_mask, coll.p);
    uint32_t c = 0;
    DX_HIP(hipMemcpyAsync(&c, coll.p, 4, hipMemcpyDeviceToHost, s));
    DX_HIP(hipStreamSynchronize(s));
    if (!c) return;
    d.seed = mix64(d.seed + 0x9E3779B97F4A7C15ull);
  }
  throw DxError(DX_E_RANGE, "dictionary: hash collisions persisted over 8 seeds");
}

void dict_lookup(dx_ctx* ctx, const char* bytes, const uint64_t* offsets, uint64_t n, int64_t* ids_host) {
  if (n == 0) return;
  Dictionary& d = ctx->dict;
  if (!d.valid()) throw DxError(DX_E_INVAL, "no dictionary: run dx_partition_* or dx_open_dataset first");
  hipStream_t s = ctx->stream;
  const uint64_t nbytes = offsets[n];
  DevBuf<uint8_t> q(ctx->pool, nbytes + 1);
  DevBuf<uint64_t> qo(ctx->pool, n + 1);
  DevBuf<int64_t> out(ctx->pool, n);
  HostFetch fetch(ctx);
  const size_t mark = ctx->part_host_used;
  fetch.upload(q.p, bytes, nbytes);
  fetch.upload(qo.p, offsets, (n + 1) * 8);
  hipLaunchKernelGGL(k_lookup, dim3(blocks(n, 64)), dim3(64), 0, s, q.p, qo.p, (uint32_t)n, d.seed, d.ht.p,
                     d.ht_mask, d.pool.p, d.pool_off.p, out.p);
  fetch.add(ids_host, out.p, n * 8);
  fetch.sync();
  ctx->part_host_used = mark;  // nothing of this call is left in the block
}

static size_t utf16_units(const std::string& s) {
  size_t u = 0;
  for (unsigned char c : s) {
    if ((c & 0xC0) != 0x80) ++u;
    if ((c & 0xF8) == 0xF0) ++u;
  }
  return u;
}
static std::string uri_to_storage(const std::string& s) {  // DAP.scala:512-514: replaceAll matches by code point
  std::string out;
  for (size_t i = 0; i < s.size();) {
    unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      if ((c >= 'A' && c <= 'Z')) out += (char)(c - 'A' + 'a');
      else if ((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9')) out += (char)c;
      else out += '_';
      ++i;
    } else {
      const int len = (c & 0xE0) == 0xC0 ? 2 : (c & 0xF0) == 0xE0 ? 3 : (c & 0xF8) == 0xF0 ? 4 : 1;
      out += '_';  // one '_' per code point, a supplementary character included (java.util.regex works on code points)
      i += len;
    }
  }
  return out;
}

// Distinct values of the two columns of a table (load time, untimed by the reference as well: it is part
// of caching the table, QueryProcessor.scala:222-232). Subjects are counted on the stored order when the
// table is one sorted slice; any other column is counted through a byte per term.
__global__ void __launch_bounds__(256)
k_count_changes(const uint64_t* __restrict__ v, uint64_t n, unsigned long long* __restrict__ out, unsigned long long* __restrict__ descents) {
  __shared__ uint32_t lds[8];
  uint32_t c = 0, down = 0;  // down: places where the order the count relies on does not hold
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    c += (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
    down += (i != 0 && v[i] < v[i - 1]) ? 1u : 0u;
  }
  uint32_t total;
  (void)block_excl_scan_256<uint32_t>(c, lds, total);
  if (threadIdx.x == 0 && total) atomicAdd(out, (unsigned long long)total);
  if (__syncthreads_or(down != 0) && threadIdx.x == 0) atomicAdd(descents, 1ull);
}

// Distinct values of a column without sorting it: one byte per term, set by plain stores (every writer stores
// the same value, so no atomics and no lane-by-lane traffic on the bytes of frequent terms), then counted.
__global__ void __launch_bounds__(256)
k_mark_terms(const int64_t* __restrict__ col, uint64_t n, uint64_t n_terms, uint8_t* __restrict__ seen) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = (uint64_t)col[i];
    if (v < n_terms && !seen[v]) seen[v] = 1;
  }
}
__global__ void __launch_bounds__(256)
k_count_marked(const uint8_t* __restrict__ seen, uint64_t n_padded /*multiple of 16, padding bytes are zero*/, unsigned long long* __restrict__ out) {
  __shared__ uint32_t lds[8];
  uint32_t c = 0;
  for (uint64_t i = 16 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x); i < n_padded; i += 16 * (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = *reinterpret_cast<const uint4*>(seen + i);  // sixteen marks at a time; a mark is the byte 1
    c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
  uint32_t total;
  (void)block_excl_scan_256<uint32_t>(c, lds, total);
  if (threadIdx.x == 0 && total) atomicAdd(out, (unsigned long long)total);  // few blocks: the counter is one word in one L2 channel
}

// Small tables, all in one launch, one block each: the distinct values of a column are counted in a set in LDS (open
// addressing over 32-bit ids), a sorted subject column by its changes. A heterogeneous schema has tens of thousands of
// tables of a few dozen rows, and a statement with a variable predicate names hundreds of them: three launches and a
// pass over a byte per dictionary term for each of them was milliseconds per statement.
constexpr uint32_t kSmallStatsRows = 16384, kSmallStatsSlots = 32768;
struct SmallTable { const int64_t* s; const int64_t* o; uint32_t rows, sorted_s; };
__global__ void __launch_bounds__(256) k_stats_small(const SmallTable* __restrict__ tabs, const uint32_t* __restrict__ where, unsigned long long* __restrict__ out,
                                                      unsigned long long* __restrict__ descents) {
  __shared__ uint32_t set[kSmallStatsSlots];
  __shared__ uint32_t lds[8];
  const SmallTable t = tabs[blockIdx.x];
  for (int col = 1; col >= 0; --col) {
    const int64_t* __restrict__ v = col ? t.o : t.s;
    uint32_t c = 0;
    if (col == 0 && t.sorted_s) {
      uint32_t down = 0;
      for (uint32_t i = threadIdx.x; i < t.rows; i += 256) {
        c += (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
        down += (i != 0 && (uint64_t)v[i] < (uint64_t)v[i - 1]) ? 1u : 0u;
      }
      if (__syncthreads_or(down != 0) && threadIdx.x == 0) descents[where[blockIdx.x]] = 1;
    } else {
      for (uint32_t i = threadIdx.x; i < kSmallStatsSlots; i += 256) set[i] = 0xFFFFFFFFu;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < t.rows; i += 256) {
        const uint32_t x = (uint32_t)v[i];  // term ids are below 2^32 - 1
        uint32_t h = (uint32_t)mix64(x) & (kSmallStatsSlots - 1);
        for (;;) {
          const uint32_t old = atomicCAS(&set[h], 0xFFFFFFFFu, x);
          if (old == 0xFFFFFFFFu) { ++c; break; }
          if (old == x) break;
          h = (h + 1) & (kSmallStatsSlots - 1);
        }
      }
    }
    uint32_t total;
    (void)block_excl_scan_256<uint32_t>(c, lds, total);  // its barriers also end the use of the set
    if (threadIdx.x == 0) out[2 * where[blockIdx.x] + col] = total;
  }
}

void table_statistics(dx_ctx* ctx, const std::vector<CatalogTable*>& tables) {
  hipStream_t s = ctx->stream;
  std::vector<CatalogTable*> todo;
  {
    std::unordered_map<CatalogTable*, bool> seen_tables;
    for (CatalogTable* t : tables)
      if (t->stats_pending && seen_tables.emplace(t, true).second) todo.push_back(t);
  }
  if (todo.empty()) return;
  if (!ctx->dict.valid()) throw DxError(DX_E_INVAL, "no dictionary loaded");
  const uint64_t n_terms = ctx->dict.n_terms;
  DevBuf<unsigned long long> cnt(ctx->pool, 3 * todo.size());  // distinct s, distinct o per table, then a word per table for the order of s
  DX_HIP(hipMemsetAsync(cnt.p, 0, 24 * todo.size(), s));
  std::vector<SmallTable> small;
  std::vector<uint32_t> small_where;
  for (size_t k = 0; k < todo.size(); ++k) {
    const CatalogTable& t = *todo[k];
    if (t.rows == 0 || t.rows > kSmallStatsRows) continue;
    small.push_back({t.s, t.o, (uint32_t)t.rows, t.slices.size() == 1 && !t.unsorted_s ? 1u : 0u});
    small_where.push_back((uint32_t)k);
  }
  DevBuf<SmallTable> d_small(ctx->pool, small.size());
  DevBuf<uint32_t> d_where(ctx->pool, small.size());
  if (!small.empty()) {
    DX_HIP(hipMemcpyAsync(d_small.p, small.data(), small.size() * sizeof(SmallTable), hipMemcpyHostToDevice, s));
    DX_HIP(hipMemcpyAsync(d_where.p, small_where.data(), small.size() * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_stats_small, dim3((unsigned)small.size()), dim3(256), 0, s, d_small.p, d_where.p, cnt.p, cnt.p + 2 * todo.size());
  }
  const uint64_t n_padded = (n_terms + 15) / 16 * 16 + 16;
  DevBuf<uint8_t> seen;  // one map for all large tables: the stream runs them one after another
  for (size_t k = 0; k < todo.size(); ++k) {
    const CatalogTable& t = *todo[k];
    if (t.rows <= kSmallStatsRows) continue;
    if (!seen.p) seen = DevBuf<uint8_t>(ctx->pool, n_padded);
    // grids of a few hundred blocks: every block ends with one atomic on the same counter
    const unsigned g = (unsigned)std::min<uint64_t>(256, (t.rows + 255) / 256);
    const bool sorted_s = t.slices.size() == 1 && !t.unsorted_s;  // one slice of a partition: rows are in (s, o) order (looked at while counting)
    for (int col = 0; col < 2; ++col) {
      const int64_t* src = col == 0 ? t.s : t.o;
      if (col == 0 && sorted_s) {
        hipLaunchKernelGGL(k_count_changes, dim3(g), dim3(256), 0, s, (const uint64_t*)src, t.rows, cnt.p + 2 * k + col, cnt.p + 2 * todo.size() + k);
        continue;
      }
      DX_HIP(hipMemsetAsync(seen.p, 0, n_padded, s));
      hipLaunchKernelGGL(k_mark_terms, dim3(g), dim3(256), 0, s, src, t.rows, n_terms, seen.p);
      hipLaunchKernelGGL(k_count_marked, dim3((unsigned)std::min<uint64_t>(256, (n_padded / 16 + 255) / 256)), dim3(256), 0, s, seen.p, n_padded,
                         cnt.p + 2 * k + col);
    }
  }
  std::vector<CatalogTable*> recount;
  std::vector<unsigned long long> h(3 * todo.size(), 0);
  DX_HIP(hipMemcpyAsync(h.data(), cnt.p, 24 * todo.size(), hipMemcpyDeviceToHost, s));
  DX_HIP(hipStreamSynchronize(s));
  for (size_t k = 0; k < todo.size(); ++k) {
    CatalogTable& t = *todo[k];
    t.distinct_s = h[2 * k];
    t.distinct_o = h[2 * k + 1];
    // a table that is one slice had its subjects' order looked at while they were counted
    t.sorted_s = t.rows > 0 && t.slices.size() == 1 && !t.unsorted_s && h[2 * todo.size() + k] == 0;
    t.stats_pending = false;
    if (t.slices.size() == 1 && !t.unsorted_s && h[2 * todo.size() + k] != 0) {
      // a table from somebody else's files, in no particular order: its subjects' changes were no count of them
      t.unsorted_s = true;
      t.stats_pending = true;
      recount.push_back(&t);
      continue;
    }
    if (!t.key.empty()) ctx->table_stats[t.key] = {t.distinct_s, t.distinct_o, t.sorted_s};  // survive unloadTables: the data does not change
  }
  if (!recount.empty()) table_statistics(ctx, recount);
}

namespace {
__global__ void __launch_bounds__(256)
k_pack_pairs(const int64_t* __restrict__ s, const int64_t* __restrict__ o, uint64_t n, uint64_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ((uint64_t)s[i] << 32) | (uint32_t)o[i];
}
__global__ void __launch_bounds__(256)
k_unpack_distinct_pairs(const uint64_t* __restrict__ in, const uint8_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n,
                        int64_t* __restrict__ s, int64_t* __restrict__ o) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  s[pos[i]] = (int64_t)(in[i] >> 32);
  o[pos[i]] = (int64_t)(in[i] & 0xFFFFFFFFu);
}
// rows of one UNION arm that pass its constants, as packed (first column, second column) words behind *count
__global__ void __launch_bounds__(256)
k_union_pairs(const int64_t* __restrict__ s, const int64_t* __restrict__ o, uint64_t n, int has_s, int64_t want_s, int has_o, int64_t want_o,
              int src_a, int src_b, uint64_t* __restrict__ out, unsigned long long* __restrict__ count) {
  __shared__ uint32_t lds[8];
  __shared__ unsigned long long base_s;
  for (uint64_t i0 = (uint64_t)blockIdx.x * 256; i0 < n; i0 += (uint64_t)gridDim.x * 256) {  // block-uniform
    const uint64_t i = i0 + threadIdx.x;
    bool take = i < n;
    int64_t vs = 0, vo = 0;
    if (take) {
      vs = s[i];
      vo = o[i];
      take = (!has_s || vs == want_s) && (!has_o || vo == want_o);
    }
    uint32_t total;
    const uint32_t before = block_excl_scan_256<uint32_t>(take ? 1u : 0u, lds, total);
    if (threadIdx.x == 0) base_s = total ? atomicAdd(count, (unsigned long long)total) : 0ull;
    __syncthreads();
    if (take) {
      const uint64_t a = src_a < 0 ? 0 : (uint64_t)(src_a == 0 ? vs : vo), b = src_b < 0 ? 0 : (uint64_t)(src_b == 0 ? vs : vo);
      out[base_s + before] = (a << 32) | (uint32_t)b;
    }
    __syncthreads();
  }
}
}  // namespace

const CatalogTable* distinct_union_of(dx_ctx* ctx, const std::vector<UnionSource>& arms, const std::string& signature) {
  auto hit = ctx->union_tables.find(signature);
  if (hit != ctx->union_tables.end()) return hit->second.get();
  if (ctx->dict.n_terms >= (1ull << 32)) throw DxError(DX_E_RANGE, "dictionary too large for a duplicate-free UNION");
  hipStream_t s = ctx->stream;
  DevPool& pool = ctx->pool;
  uint64_t most = 0;
  for (const UnionSource& a : arms) most += a.t->rows;
  if (most >= (1ull << 32)) throw DxError(DX_E_RANGE, "UNION arms of 2^32 rows or more");
  DevBuf<uint64_t> keys(pool, std::max<uint64_t>(most, 1));
  DevBuf<unsigned long long> count(pool, 1);
  DX_HIP(hipMemsetAsync(count.p, 0, 8, s));
  for (const UnionSource& a : arms)
    if (a.t->rows)
      hipLaunchKernelGGL(k_union_pairs, dim3(std::min<unsigned>(blocks(a.t->rows), 4096u)), dim3(256), 0, s, a.t->s, a.t->o, a.t->rows, a.has_s ? 1 : 0, a.want_s,
                         a.has_o ? 1 : 0, a.want_o, a.src_a, a.src_b, keys.p, count.p);
  unsigned long long n64 = 0;
  DX_HIP(hipMemcpyAsync(&n64, count.p, 8, hipMemcpyDeviceToHost, s));
  DX_HIP(hipStreamSynchronize(s));
  const uint32_t n = (uint32_t)n64;
  auto v = std::make_unique<CatalogTable>();
  v->view = "(UNION, every row once)";
  uint32_t kept = 0;
  if (n) {
    radix_sort_u64(pool, keys, n, 32 + bits_for(ctx->dict.n_terms ? ctx->dict.n_terms : 1), s);
    DevBuf<uint8_t> flag(pool, n);
    DevBuf<uint32_t> pos(pool, n), d_total(pool, 1);
    hipLaunchKernelGGL(k_flag_distinct<uint64_t>, dim3(blocks(n)), dim3(256), 0, s, keys.p, n, flag.p);
    exclusive_scan_u8_u32(pool, flag.p, pos.p, n, d_total.p, s);
    DX_HIP(hipMemcpyAsync(&kept, d_total.p, 4, hipMemcpyDeviceToHost, s));
    DX_HIP(hipStreamSynchronize(s));
    v->own_s = DevBuf<int64_t>(pool, kept);
    v->own_o = DevBuf<int64_t>(pool, kept);
    hipLaunchKernelGGL(k_unpack_distinct_pairs, dim3(blocks(n)), dim3(256), 0, s, keys.p, flag.p, pos.p, n, v->own_s.p, v->own_o.p);
    DX_HIP(hipStreamSynchronize(s));
  } else {
    v->own_s = DevBuf<int64_t>(pool, 1);
    v->own_o = DevBuf<int64_t>(pool, 1);
  }
  v->rows = kept;
  v->s = v->own_s.p;
  v->o = v->own_o.p;
  v->slices.push_back({v->s, v->o, kept});
  v->stats_pending = true;
  const CatalogTable* out = v.get();
  ctx->union_tables.emplace(signature, std::move(v));
  return out;
}

const CatalogTable* distinct_view_of(dx_ctx* ctx, CatalogTable& t) {
  if (!t.several_sources || t.rows < 2) return &t;
  if (t.distinct_view) return t.distinct_view.get();
  if (t.rows >= (1ull << 32) || ctx->dict.n_terms >= (1ull << 32)) throw DxError(DX_E_RANGE, "table too large for a distinct copy");
  hipStream_t s = ctx->stream;
  DevPool& pool = ctx->pool;
  const uint32_t n = (uint32_t)t.rows;
  DevBuf<uint64_t> keys(pool, n);
  hipLaunchKernelGGL(k_pack_pairs, dim3(blocks(n)), dim3(256), 0, s, t.s, t.o, (uint64_t)n, keys.p);
  radix_sort_u64(pool, keys, n, 32 + bits_for(ctx->dict.n_terms ? ctx->dict.n_terms : 1), s);
  DevBuf<uint8_t> flag(pool, n);
  DevBuf<uint32_t> pos(pool, n), d_total(pool, 1);
  hipLaunchKernelGGL(k_flag_distinct<uint64_t>, dim3(blocks(n)), dim3(256), 0, s, keys.p, n, flag.p);
  exclusive_scan_u8_u32(pool, flag.p, pos.p, n, d_total.p, s);
  uint32_t kept = 0;
  DX_HIP(hipMemcpyAsync(&kept, d_total.p, 4, hipMemcpyDeviceToHost, s));
  DX_HIP(hipStreamSynchronize(s));
  auto v = std::make_unique<CatalogTable>();
  v->key.clear();  // not a table of its own: its statistics are not kept by key
  v->view = t.view;
  v->rows = kept;
  v->own_s = DevBuf<int64_t>(pool, kept);
  v->own_o = DevBuf<int64_t>(pool, kept);
  hipLaunchKernelGGL(k_unpack_distinct_pairs, dim3(blocks(n)), dim3(256), 0, s, keys.p, flag.p, pos.p, n, v->own_s.p, v->own_o.p);
  DX_HIP(hipStreamSynchronize(s));
  v->s = v->own_s.p;
  v->o = v->own_o.p;
  v->slices.push_back({v->s, v->o, kept});  // sorted by (s, o): the subject column can be counted by its changes
  v->stats_pending = true;
  v->remote = t.remote;
  t.distinct_view = std::move(v);
  return t.distinct_view.get();
}

namespace {
void comm_ok(int rc, const char* what) {
  if (rc != 0) throw DxError(DX_E_IO, std::string("collective failed: ") + what);
}
// every rank's value of a host number, in rank order
std::vector<uint64_t> gather_u64(const dx_comm& cm, uint64_t v) {
  std::vector<uint64_t> send(cm.world, v), recv(cm.world, 0);
  comm_ok(cm.exchange_host(cm.user, send.data(), recv.data(), 1), "exchange_host");
  return recv;
}

// Replaces the rank's dictionary (built over its own part of the text) by the dictionary of all ranks'
// terms and rewrites the records' term ids accordingly. On return ctx->dict and n_terms are the same on every rank.
void globalise_dictionary(dx_ctx* ctx, uint32_t n_rec, uint32_t* S, uint32_t* P, uint32_t* O, uint32_t& n_terms) {
  const dx_comm& cm = ctx->comm;
  hipStream_t s = ctx->stream;
  DevPool& pool = ctx->pool;
  Dictionary& d = ctx->dict;
  const int W = cm.world;
  uint64_t local_bytes = 0;
  DX_HIP(hipMemcpyAsync(&local_bytes, d.pool_off.p + n_terms, 8, hipMemcpyDeviceToHost, s));
  DX_HIP(hipStreamSynchronize(s));
  const std::vector<uint64_t> n_r = gather_u64(cm, n_terms), bytes_r = gather_u64(cm, local_bytes);
  std::vector<uint64_t> entry_base(W + 1, 0), byte_base(W + 1, 0), off_counts(W);
  for (int r = 0; r < W; ++r) {
    entry_base[r + 1] = entry_base[r] + n_r[r];
    byte_base[r + 1] = byte_base[r] + bytes_r[r];
    off_counts[r] = n_r[r] + 1;
  }
  const uint64_t E = entry_base[W], total_bytes = byte_base[W];
  if (E >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 dictionary entries over all ranks");
  if (total_bytes >= (1ull << (64 - kRepLenBits))) throw DxError(DX_E_RANGE, "term pools larger than 1 TiB");
  DevBuf<uint8_t> gpool(pool, total_bytes + 16);
  DevBuf<uint64_t> goff(pool, E + W);
  comm_ok(cm.allgather(cm.user, d.pool.p, local_bytes, gpool.p, bytes_r.data(), 1), "allgather of term bytes");
  comm_ok(cm.allgather(cm.user, d.pool_off.p, (uint64_t)n_terms + 1, goff.p, off_counts.data(), 8), "allgather of term offsets");
  DevBuf<uint64_t> bases(pool, 2 * (W + 1));
  DX_HIP(hipMemcpyAsync(bases.p, entry_base.data(), (W + 1) * 8, hipMemcpyHostToDevice, s));
  DX_HIP(hipMemcpyAsync(bases.p + W + 1, byte_base.data(), (W + 1) * 8, hipMemcpyHostToDevice, s));
  DevBuf<uint64_t> e_rep(pool, E), e_hash(pool, E);
  DevBuf<uint32_t> e_slot(pool, E), first_scan(pool, E), coll(pool, 2);
  DevBuf<uint8_t> first(pool, E);
  DevBuf<uint32_t> d_total(pool, 1);
  uint64_t cap = 1024;
  while (cap < E + E / 2) cap <<= 1;
  DevBuf<DictSlot> ht(pool, cap);
  uint64_t seed = 0x243F6A8885A308D3ull;
  uint32_t n_global = 0;
  for (int attempt = 0;; ++attempt) {
    if (E) hipLaunchKernelGGL(k_entry_spans, dim3(blocks(E)), dim3(256), 0, s, gpool.p, goff.p, bases.p, bases.p + W + 1, W, E, seed, e_rep.p, e_hash.p);
    hipLaunchKernelGGL(k_ht_clear, dim3(2048), dim3(256), 0, s, ht.p, cap);
    DX_HIP(hipMemsetAsync(coll.p, 0, 8, s));
    if (E) {
      hipLaunchKernelGGL(k_dict_insert, dim3(blocks(E)), dim3(256), 0, s, e_hash.p, e_rep.p, E, ht.p, cap - 1, e_slot.p, coll.p + 1);
      hipLaunchKernelGGL(k_entry_verify, dim3(blocks(E)), dim3(256), 0, s, gpool.p, e_rep.p, e_slot.p, E, ht.p, first.p, coll.p);
    }
    exclusive_scan_u8_u32(pool, first.p, first_scan.p, E, d_total.p, s);
    uint32_t h_coll[2] = {0, 0};
    DX_HIP(hipMemcpyAsync(h_coll, coll.p, 8, hipMemcpyDeviceToHost, s));
    DX_HIP(hipMemcpyAsync(&n_global, d_total.p, 4, hipMemcpyDeviceToHost, s));
    DX_HIP(hipStreamSynchronize(s));
    // every rank sees the same entries, so every rank takes the same turns here
    if (h_coll[1]) {
      if (cap >= 8ull * E && cap >= 4096) throw DxError(DX_E_RANGE, "dictionary: probe chains too long at 8 slots per entry");
      cap <<= 1;
      ht = DevBuf<DictSlot>(pool, cap);
      continue;
    }
    if (!h_coll[0]) break;
    if (attempt >= 7) throw DxError(DX_E_RANGE, "dictionary: hash collisions persisted over 8 seeds");
    seed = mix64(seed + 0x9E3779B97F4A7C15ull);
  }
  DevBuf<uint32_t> term_len(pool, (size_t)n_global + 1);
  DevBuf<uint64_t> term_src(pool, n_global);
  DX_HIP(hipMemsetAsync(term_len.p, 0, ((size_t)n_global + 1) * 4, s));
  DevBuf<uint32_t> gid(pool, E);
  if (E) {
    hipLaunchKernelGGL(k_dict_assign, dim3(blocks((E + 3) / 4)), dim3(256), 0, s, e_rep.p, e_slot.p, first.p, first_scan.p, E, ht.p, term_len.p, term_src.p);
    hipLaunchKernelGGL(k_entry_ids, dim3(blocks(E)), dim3(256), 0, s, e_slot.p, ht.p, E, gid.p);
  }
  Dictionary g;
  g.n_terms = n_global;
  g.seed = seed;
  g.pool_off = DevBuf<uint64_t>(pool, (size_t)n_global + 1);
  DevBuf<uint64_t> d_total64(pool, 1);
  exclusive_scan_u32_u64(pool, term_len.p, g.pool_off.p, (size_t)n_global + 1, d_total64.p, s);
  uint64_t g_bytes = 0;
  DX_HIP(hipMemcpyAsync(&g_bytes, d_total64.p, 8, hipMemcpyDeviceToHost, s));
  DX_HIP(hipStreamSynchronize(s));
  g.pool = DevBuf<uint8_t>(pool, g_bytes + 16);
  if (n_global) hipLaunchKernelGGL(k_pool_copy, dim3(blocks(4ull * n_global)), dim3(256), 0, s, gpool.p, term_src.p, g.pool_off.p, (uint64_t)n_global, g.pool.p);
  g.ht = std::move(ht);
  g.ht_mask = cap - 1;
  // my local id j is entry entry_base[rank] + j
  const uint32_t* map = gid.p + entry_base[cm.rank];
  if (n_rec) {
    hipLaunchKernelGGL(k_remap, dim3(blocks(n_rec)), dim3(256), 0, s, S, map, (uint64_t)n_rec);
    hipLaunchKernelGGL(k_remap, dim3(blocks(n_rec)), dim3(256), 0, s, P, map, (uint64_t)n_rec);
    hipLaunchKernelGGL(k_remap, dim3(blocks(n_rec)), dim3(256), 0, s, O, map, (uint64_t)n_rec);
  }
  DX_HIP(hipStreamSynchronize(s));
  DX_HIP(hipGetLastError());
  ctx->dict = std::move(g);
  n_terms = n_global;
}
}  // namespace

static void run_partition_body(dx_ctx* ctx, const uint8_t* text, uint64_t n_bytes, const std::vector<ClassIndexEntry>& classes, int nparts);

// With a communicator the call is collective, and a rank must not leave it alone: the body passes checkpoints
// (agreements on "no rank has failed") before each group of collectives and at its end, and a rank whose own
// work fails in between brings its error to the checkpoint its peers reach next. All ranks return the same code.
void run_partition(dx_ctx* ctx, const uint8_t* text, uint64_t n_bytes, const std::vector<ClassIndexEntry>& classes, int nparts) {
  if (ctx->comm.world <= 1) {
    run_partition_body(ctx, text, n_bytes, classes, nparts);
    return;
  }
  LocalError err;
  try {
    run_partition_body(ctx, text, n_bytes, classes, nparts);
    return;
  } catch (const AgreedError&) {
    ctx->part = PartitionState();
    throw;
  } catch (const DxError& e) {
    err.code = e.code; err.msg = e.what();
  } catch (const HipError& e) {
    err.code = DX_E_HIP; err.msg = e.what();
  } catch (const std::bad_alloc&) {
    err.code = DX_E_NOMEM; err.msg = "host allocation failed";
  }
  ctx->part = PartitionState();
  agree_or_throw(ctx, err);
}

// runs a statement with K = the key word of this partition run and keys = its sorted key array
#define DX_WITH_KEYS(...)                                                                 \
  do {                                                                                    \
    if (ps.wide_keys) { using K = u128; const K* keys = ps.uniq_w.p; __VA_ARGS__; }       \
    else { using K = uint64_t; const K* keys = ps.uniq.p; __VA_ARGS__; }                  \
  } while (0)

static void run_partition_body(dx_ctx* ctx, const uint8_t* text, uint64_t n_bytes, const std::vector<ClassIndexEntry>& classes,
                               int nparts) {
  hipStream_t s = ctx->stream;
  DevPool& pool = ctx->pool;
  if (nparts < 1 || nparts > DX_MAX_PARTITIONS) throw DxError(DX_E_INVAL, "n_partitions must be in [1, 64]");
  if (n_bytes >= (1ull << (64 - kRepLenBits))) throw DxError(DX_E_RANGE, "input larger than 1 TiB");
  ctx->catalog.clear(); ctx->union_tables.clear();
  ctx->table_stats.clear();
  ctx->imported.clear();
  ctx->part = PartitionState();
  ctx->measures = MeasureState();
  ctx->dict = Dictionary();
  PartitionState& ps = ctx->part;
  ps.n_partitions = nparts;
  dx_part_stats& st = ps.stats;
  st.n_bytes = n_bytes;
  StageTimer tm(ctx);
  ctx->part_host_used = 0;
  HostFetch fetch(ctx, /*batch=*/true);  // every device buffer handed to fetch.add() below lives until the next fetch.sync()

  // ---- 1.-3. record starts, tokeniser, dictionary
  const unsigned ntiles = (unsigned)((n_bytes + LS_TILE - 1) / LS_TILE);
  DevBuf<unsigned long long> err(pool, 1);
  DevBuf<uint32_t> d_total(pool, 4);
  DevBuf<uint32_t> coll(pool, 4);  // [0] true hash collision, [1] table overflow, [2] a tile with more records than slots
  DevBuf<uint32_t> tile_counts(pool, ntiles + 1);   // records starting in each tile; scanned: the tile's first record
  DevBuf<uint32_t> tile_leaders(pool, ntiles + 1);  // dictionary entries of each tile
  DevBuf<uint32_t> tile_firsts(pool, ntiles + 1);   // representatives per tile, then (scanned) before each tile
  DevBuf<uint32_t> tile_bytes(pool, ntiles + 1);    // bytes of each tile's representatives
  DevBuf<unsigned long long> d_pool_bytes(pool, 1);
  uint64_t pool_bytes = 0;                          // all terms' bytes: comes back with the number of terms
  DevBuf<uint32_t> slot_first;                      // fused form: where each tile's slots begin
  DevBuf<uint16_t> chunk_mask;                      // two-pass form: one 16-bit start mask per 16 bytes of text
  // Dictionary entries: the tokeniser folds the occurrences of a tile onto their first one there, and only those
  // (a quarter of the occurrences on LUBM) go to the table. Tile t's entries sit at [3 * first slot of t, + leaders of t);
  // the slots of a tile are its records (two-pass form) or a fixed number per tile (fused form).
  DevBuf<ulonglong2> ent;
  DevBuf<uint32_t> t_entry, e_slot;
  DevBuf<uint8_t> first, kind_slots;
  DevBuf<DictSlot> ht;
  uint64_t cap = 0;
  uint32_t n_rec = 0, n_terms = 0;
  uint64_t n_slots = 0;
  uint64_t seed = 0x243F6A8885A308D3ull;
  float ms_tok = 0, ms_clear = 0, ms_claim = 0, ms_scan = 0;
  const unsigned en_blocks = (ntiles + EN_TILES_PER_BLOCK - 1) / EN_TILES_PER_BLOCK;
  // The fused form reads the text once; it needs every tile to hold at most kSlotsPerTile records and 32-bit entry
  // numbers over all slots. A context that met a tile with more records takes the two-pass form for its next calls.
  // DX_TWO_PASS=1 always takes the two-pass form (how the tests reach it on ordinary inputs).
  bool fused = ctx->part_two_pass_calls == 0 && 3ull * ntiles * kSlotsPerTile < (1ull << 32) &&
               !(getenv("DX_TWO_PASS") && *getenv("DX_TWO_PASS") == '1');
  if (ctx->part_two_pass_calls) --ctx->part_two_pass_calls;
  const uint32_t* entry_first = nullptr;  // what the kernels over the tiles' entries take as "first slot of the tile"
  bool scanned = false;
  for (;;) {
    const hipEvent_t e0 = pooled_event(ctx, 0), e1 = pooled_event(ctx, 1), e2 = pooled_event(ctx, 2), e3 = pooled_event(ctx, 3);
    DX_HIP(hipMemsetAsync(err.p, 0xFF, 8, s));
    DX_HIP(hipMemsetAsync(coll.p, 0, 16, s));
    const uint64_t type_hash = hash_bytes((const uint8_t*)"<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>", kTypeLen, seed);
    if (!fused && !scanned && n_bytes) {
      // two-pass form, first pass: start masks and counts; the host learns the number of records before the tokeniser runs
      const hipEvent_t ls0 = pooled_event(ctx, 20), ls1 = pooled_event(ctx, 21);
      chunk_mask = DevBuf<uint16_t>(pool, (size_t)ntiles * (LS_TILE / 16) + 64);
      DX_HIP(hipEventRecord(ls0, s));
      DX_HIP(hipMemsetAsync(tile_counts.p + ntiles, 0, 4, s));
      hipLaunchKernelGGL(k_line_scan, dim3(ntiles), dim3(LS_THREADS), 0, s, text, n_bytes, chunk_mask.p, tile_counts.p, err.p);
      exclusive_scan_u32_u32(pool, tile_counts.p, tile_counts.p, (size_t)ntiles + 1, d_total.p, s);
      DX_HIP(hipEventRecord(ls1, s));
      fetch.add(&n_rec, d_total.p, 4);
      fetch.sync();
      DX_HIP(hipEventSynchronize(ls1));
      DX_HIP(hipEventElapsedTime(&ms_scan, ls0, ls1));
      scanned = true;
    }
    if (!fused) {
      if (3ull * n_rec >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32/3 records on one GPU");
      n_slots = n_rec;
      entry_first = tile_counts.p;
    } else {
      n_slots = (uint64_t)ntiles * kSlotsPerTile;
      if (!slot_first.p) {
        slot_first = DevBuf<uint32_t>(pool, ntiles + 1);
        hipLaunchKernelGGL(k_iota_times, dim3(blocks((uint64_t)ntiles + 1)), dim3(256), 0, s, slot_first.p, (uint64_t)ntiles + 1, kSlotsPerTile);
      }
      entry_first = slot_first.p;
    }
    if (ent.n != 3 * n_slots || !ent.p) {
      ent = DevBuf<ulonglong2>(pool, 3 * n_slots);
      t_entry = DevBuf<uint32_t>(pool, 3 * n_slots);
      e_slot = DevBuf<uint32_t>(pool, 3 * n_slots);
      first = DevBuf<uint8_t>(pool, 3 * n_slots);
      kind_slots = DevBuf<uint8_t>(pool, n_slots);
    }
    // ---- tokenise + hash + folding inside the tiles in one pass over the text
    DX_HIP(hipEventRecord(e1, s));
    if (fused) {
      if (n_bytes) {
        DX_HIP(hipMemsetAsync(tile_counts.p + ntiles, 0, 4, s));
        hipLaunchKernelGGL(k_tokenise<true>, dim3(ntiles), dim3(TK_THREADS), 0, s, text, n_bytes, (const uint16_t*)nullptr, (const uint32_t*)nullptr,
                           seed, type_hash, n_slots, ent.p, t_entry.p, tile_leaders.p, kind_slots.p, err.p, tile_counts.p, coll.p + 2);
      }
    } else if (n_rec) {
      hipLaunchKernelGGL(k_tokenise<false>, dim3(ntiles), dim3(TK_THREADS), 0, s, text, n_bytes, chunk_mask.p, tile_counts.p, seed, type_hash,
                         n_slots, ent.p, t_entry.p, tile_leaders.p, kind_slots.p, err.p, (uint32_t*)nullptr, (uint32_t*)nullptr);
    }
    DX_HIP(hipEventRecord(e2, s));
    if (fused) {  // the counts, scanned; the host sizes the table by the number of records
      if (n_bytes) {
        exclusive_scan_u32_u32(pool, tile_counts.p, tile_counts.p, (size_t)ntiles + 1, d_total.p, s);
        uint32_t h_over = 0;
        unsigned long long h_err = ~0ull;
        fetch.add(&n_rec, d_total.p, 4);
        fetch.add(&h_over, coll.p + 2, 4);
        fetch.add(&h_err, err.p, 8);
        fetch.sync();
        if (h_over) {  // some tile holds more records than it has slots: this input takes the two-pass form
          fused = false;
          ctx->part_two_pass_calls = 8;  // the next inputs are likely to be of the same kind: do not try each of them twice
          ent.reset();
          continue;
        }
        if (h_err != ~0ull)
          throw DxError(DX_E_MALFORMED, "malformed N-Triples input at byte " + std::to_string(h_err) +
                                            " (empty line, fewer than three fields, or a term of 16 MiB or more)");
        if (3ull * n_rec >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32/3 records on one GPU");
      }
    }
    // Table capacity: at least half a slot per record to start with (LUBM-like data has ~0.26 distinct terms per
    // record, so the table is at most half full and, at LUBM-100, small enough to stay in the last-level cache while
    // the entries gather from it); a probe chain that runs too long reports overflow and the build restarts twice
    // as large (at 8 slots per record it is at most three eighths full and chains of 2^10 do not occur).
    if (!cap) {
      cap = 1024;
      while (cap < n_rec / 2) cap <<= 1;
    }
    if (ht.n != cap) ht = DevBuf<DictSlot>(pool, cap);
    // ---- the tiles' entries claim their slots, and every entry that is not its term's first occurrence is
    // compared with that one
    DX_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_ht_clear, dim3(2048), dim3(256), 0, s, ht.p, cap);
    DX_HIP(hipEventRecord(e3, s));
    const hipEvent_t c0 = pooled_event(ctx, 22), c1 = pooled_event(ctx, 23);
    DX_HIP(hipEventRecord(c0, s));
    if (n_rec)
      hipLaunchKernelGGL(k_entry_claim, dim3(en_blocks), dim3(64 * EN_TILES_PER_BLOCK), 0, s, entry_first, tile_leaders.p, ntiles, text, ent.p,
                         ht.p, cap - 1, e_slot.p, first.p, coll.p);
    DX_HIP(hipEventRecord(c1, s));
    if (n_rec)
      hipLaunchKernelGGL(k_entry_first, dim3(en_blocks), dim3(64 * EN_TILES_PER_BLOCK), 0, s, entry_first, tile_leaders.p, ntiles, text,
                         ent.p, e_slot.p, ht.p, first.p, tile_firsts.p, tile_bytes.p, coll.p);
    if (n_rec) DX_HIP(hipMemsetAsync(tile_firsts.p + ntiles, 0, 4, s));
    if (n_rec) exclusive_scan_u32_u32(pool, tile_firsts.p, tile_firsts.p, (size_t)ntiles + 1, d_total.p, s);
    else DX_HIP(hipMemsetAsync(d_total.p, 0, 4, s));
    if (n_rec) hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(1024), 0, s, tile_leaders.p, ntiles, d_total.p + 1);
    DX_HIP(hipMemsetAsync(d_pool_bytes.p, 0, 8, s));
    if (n_rec) hipLaunchKernelGGL(k_sum_u32_u64, dim3(std::min(blocks(ntiles, 1024), 64u)), dim3(1024), 0, s, tile_bytes.p, ntiles, d_pool_bytes.p);
    fetch.add(&pool_bytes, d_pool_bytes.p, 8);
    unsigned long long h_err = ~0ull;
    uint32_t h_coll[2] = {0, 0};
    fetch.add(&h_err, err.p, 8);
    fetch.add(h_coll, coll.p, 8);
    fetch.add(&n_terms, d_total.p, 4);
    if (n_rec) fetch.add(&st.n_dict_entries, d_total.p + 1, 4);
    fetch.sync();
    DX_HIP(hipEventSynchronize(c1));  // passed on the device already (the results behind it are here); the runtime has to see it too
    DX_HIP(hipEventElapsedTime(&ms_clear, e0, e3)); DX_HIP(hipEventElapsedTime(&ms_tok, e1, e2)); DX_HIP(hipEventElapsedTime(&ms_claim, c0, c1));
    if (h_err != ~0ull)
      throw DxError(DX_E_MALFORMED, "malformed N-Triples input at byte " + std::to_string(h_err) +
                                        " (empty line, fewer than three fields, or a term of 16 MiB or more)");
    if (h_coll[1]) {  // table too full for its probe bound
      if (cap >= 8ull * n_rec && cap >= 4096) throw DxError(DX_E_RANGE, "dictionary: probe chains too long at 8 slots per record");
      cap <<= 1;
      continue;
    }
    if (!h_coll[0]) break;
    if (st.hash_retries >= 7) throw DxError(DX_E_RANGE, "dictionary: hash collisions persisted over 8 seeds");
    seed = mix64(seed + 0x9E3779B97F4A7C15ull);
    ++st.hash_retries;
  }
  st.n_triples = n_rec;
  tm.mark();  // boundary of the first stage; the times of the stages up to the encode come from their own events
  const uint64_t n_occ = 3ull * n_rec;
  chunk_mask.reset();
  st.n_terms = n_terms;
  Dictionary& d = ctx->dict;
  d.n_terms = n_terms;
  d.seed = seed;
  DevBuf<uint32_t> term_len(pool, (size_t)n_terms + 1);
  DevBuf<uint64_t> term_src(pool, n_terms);
  DevBuf<uint32_t> S(pool, n_rec), P(pool, n_rec), O(pool, n_rec);
  DevBuf<uint8_t> kind;  // per record: 0 object-property triple, 1 rdf:type, 2 literal object
  if (fused) kind = DevBuf<uint8_t>(pool, n_rec);
  ps.term_flags = DevBuf<uint8_t>(pool, ((size_t)n_terms + 4) & ~(size_t)3);
  size_t nt_pad = ((size_t)n_terms + 4) & ~(size_t)3;
  DevBuf<uint8_t> fact(pool, 4 * nt_pad);  // is_pred, is_vertex, has_edge, is_literal
  uint8_t* is_pred = fact.p;
  DX_HIP(hipMemsetAsync(fact.p, 0, fact.bytes(), s));
  DX_HIP(hipMemsetAsync(term_len.p, 0, ((size_t)n_terms + 1) * 4, s));
  if (n_occ) {
    hipLaunchKernelGGL(k_entry_assign, dim3(en_blocks), dim3(64 * EN_TILES_PER_BLOCK), 0, s, entry_first, tile_leaders.p, ntiles, ent.p,
                       e_slot.p, first.p, tile_firsts.p, ht.p, term_len.p, term_src.p);
    hipLaunchKernelGGL(k_entry_to_id, dim3(en_blocks), dim3(64 * EN_TILES_PER_BLOCK), 0, s, entry_first, tile_leaders.p, ntiles, ht.p, first.p, e_slot.p);
    if (fused)
      hipLaunchKernelGGL(k_encode_tiles, dim3(en_blocks), dim3(64 * EN_TILES_PER_BLOCK), 0, s, tile_counts.p, ntiles, t_entry.p, n_slots, e_slot.p,
                         kind_slots.p, S.p, P.p, O.p, kind.p, fact.p + nt_pad, fact.p + 2 * nt_pad, fact.p + 3 * nt_pad, is_pred);
    else
      hipLaunchKernelGGL(k_encode, dim3(blocks(n_rec)), dim3(256), 0, s, t_entry.p, e_slot.p, kind_slots.p, n_rec, S.p, P.p, O.p,
                         fact.p + nt_pad, fact.p + 2 * nt_pad, fact.p + 3 * nt_pad, is_pred);
  }
  hipLaunchKernelGGL(k_merge_flags, dim3(blocks(n_terms)), dim3(256), 0, s, fact.p + nt_pad, fact.p + 2 * nt_pad, fact.p + 3 * nt_pad,
                     n_terms, ps.term_flags.p);
  // the build-time table doubles as the lookup table
  d.ht = std::move(ht);
  d.ht_mask = cap - 1;
  tm.mark();  // boundary placed after the encode; tokenise/dictionary times come from their own events
  if (fused) kind_slots.reset(); else kind = std::move(kind_slots);  // two-pass form: the slots are the records
  t_entry.reset(); e_slot.reset(); first.reset(); tile_firsts.reset(); tile_bytes.reset(); slot_first.reset();

  // ---- 4. string pool
  d.pool_off = DevBuf<uint64_t>(pool, (size_t)n_terms + 1);
  DevBuf<uint64_t> d_total64(pool, 1);
  exclusive_scan_u32_u64(pool, term_len.p, d.pool_off.p, (size_t)n_terms + 1, d_total64.p, s);
  d.pool = DevBuf<uint8_t>(pool, pool_bytes + 16);  // the sum of the lengths is known since the representatives were counted: no wait here
  if (n_terms) hipLaunchKernelGGL(k_pool_copy, dim3(blocks(4ull * n_terms)), dim3(256), 0, s, text, term_src.p, d.pool_off.p,
                                  (uint64_t)n_terms, d.pool.p);
  tm.mark();
  ent.reset(); term_src.reset(); term_len.reset();

  // ---- several GPUs: one dictionary for all ranks; the per-term facts are recomputed over its ids and
  // summed over the ranks (a fact holds if it holds anywhere). The time lands in the vertex_masks stage.
  const dx_comm& cm = ctx->comm;
  const bool sharded = cm.world > 1;
  uint64_t owned = ~0ull;  // first-level partitions this rank keeps
  // Agreement that no rank has failed so far (a collective; does nothing on one GPU). A rank that fails between two
  // of these brings its error to the next one from run_partition's handler.
  auto checkpoint = [&] { if (sharded) agree_or_throw(ctx, LocalError{}); };
  if (sharded) {
    if (!cm.allgather || !cm.allreduce_sum_dev) throw DxError(DX_E_INVAL, "the communicator lacks the partitioner's collectives");
    checkpoint();  // malformed input, dictionary limits, anything of the local passes above
    owned = 0;
    for (int q = 0; q < nparts; ++q)
      if (q % cm.world == cm.rank) owned |= 1ull << q;
    globalise_dictionary(ctx, n_rec, S.p, P.p, O.p, n_terms);
    st.n_terms = n_terms;
    nt_pad = ((size_t)n_terms + 4) & ~(size_t)3;
    fact = DevBuf<uint8_t>(pool, 4 * nt_pad);
    is_pred = fact.p;
    DX_HIP(hipMemsetAsync(fact.p, 0, fact.bytes(), s));
    if (n_rec)
      hipLaunchKernelGGL(k_facts, dim3(blocks(n_rec)), dim3(256), 0, s, S.p, P.p, O.p, kind.p, n_rec, fact.p + nt_pad, fact.p + 2 * nt_pad,
                         fact.p + 3 * nt_pad, is_pred);
    DX_HIP(hipStreamSynchronize(s));
    comm_ok(cm.allreduce_sum_dev(cm.user, fact.p, 4 * nt_pad, 1), "allreduce of the term facts");
    hipLaunchKernelGGL(k_nonzero_u8, dim3(blocks(4 * nt_pad)), dim3(256), 0, s, fact.p, (uint64_t)(4 * nt_pad));
    ps.term_flags = DevBuf<uint8_t>(pool, nt_pad);
    hipLaunchKernelGGL(k_merge_flags, dim3(blocks(n_terms)), dim3(256), 0, s, fact.p + nt_pad, fact.p + 2 * nt_pad, fact.p + 3 * nt_pad,
                       n_terms, ps.term_flags.p);
  }

  // ---- 5. class masks and vertex masks
  DevBuf<uint64_t> Pm(pool, n_terms), M(pool, n_terms);
  DX_HIP(hipMemsetAsync(Pm.p, 0, Pm.bytes() ? Pm.bytes() : 8, s));
  DX_HIP(hipMemsetAsync(M.p, 0, M.bytes() ? M.bytes() : 8, s));
  // The class URIs and rdf:type are looked up on the device and the ids stay there (the mask kernel reads them);
  // the host only wants rdf:type's id, which comes back with the predicate count's wait below.
  int64_t type_term = -1;
  {
    std::string bytes;
    std::vector<uint64_t> offs{0};
    std::vector<uint64_t> masks;
    for (auto& c : classes) { bytes += c.uri; offs.push_back(bytes.size()); masks.push_back(c.mask); }
    bytes += "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>";
    offs.push_back(bytes.size());
    const size_t nq = offs.size() - 1;
    DevBuf<uint8_t> q(pool, bytes.size() + 1);
    DevBuf<uint64_t> qo(pool, nq + 1);
    DevBuf<int64_t> dids(pool, nq);
    fetch.upload(q.p, bytes.data(), bytes.size());
    fetch.upload(qo.p, offs.data(), (nq + 1) * 8);
    hipLaunchKernelGGL(k_lookup, dim3(blocks(nq, 64)), dim3(64), 0, s, q.p, qo.p, (uint32_t)nq, d.seed, d.ht.p, d.ht_mask, d.pool.p,
                       d.pool_off.p, dids.p);
    fetch.add(&type_term, dids.p + (nq - 1), 8);
    if (!classes.empty() && n_terms) {
      DevBuf<uint64_t> dmask(pool, classes.size());
      fetch.upload(dmask.p, masks.data(), classes.size() * 8);
      hipLaunchKernelGGL(k_set_class_masks, dim3(blocks(classes.size())), dim3(256), 0, s, dids.p, dmask.p, (uint32_t)classes.size(), Pm.p);
    }
    fetch.flush();  // dids ends here
  }
  if (n_rec) hipLaunchKernelGGL(k_vertex_masks, dim3(blocks(n_rec)), dim3(256), 0, s, S.p, O.p, kind.p, n_rec, Pm.p, M.p);
  if (sharded && n_terms) {
    // A vertex's types may be declared in another rank's part of the text: the ranks' masks are OR-ed. Rank r does it
    // for the r-th slice of the id range: every rank sends it that slice of its own masks (one all-to-all), it folds
    // the W slices, and the folded slices are gathered: 2 x 8 bytes per term over the links instead of W x 8.
    const int W = cm.world;
    const uint64_t per = ((uint64_t)n_terms + W - 1) / W;
    auto slice_len = [&](int r) { const uint64_t b = std::min<uint64_t>(n_terms, per * r), e = std::min<uint64_t>(n_terms, per * (r + 1)); return e - b; };
    const uint64_t mine = slice_len(cm.rank);
    std::vector<uint64_t> send(W), recv(W, mine), lens(W);
    for (int r = 0; r < W; ++r) send[r] = lens[r] = slice_len(r);
    DevBuf<uint64_t> slices(pool, std::max<uint64_t>(1, (uint64_t)W * mine)), folded(pool, std::max<uint64_t>(1, mine));
    DX_HIP(hipStreamSynchronize(s));
    comm_ok(cm.exchange(cm.user, M.p, send.data(), slices.p, recv.data(), 8), "exchange of the vertex masks' slices");
    if (mine) hipLaunchKernelGGL(k_or_gathered, dim3(blocks(mine)), dim3(256), 0, s, slices.p, W, mine, folded.p);
    DX_HIP(hipStreamSynchronize(s));
    comm_ok(cm.allgather(cm.user, folded.p, mine, M.p, lens.data(), 8), "allgather of the folded vertex masks");
    DX_HIP(hipStreamSynchronize(s));
  }
  tm.mark();

  // ---- 6. predicates, sort keys
  DevBuf<uint32_t> pred_index(pool, (size_t)n_terms + 1);
  exclusive_scan_u8_u32(pool, is_pred, pred_index.p, n_terms, d_total.p, s);
  // The list of predicates is written before the host knows how many there are (room for every term), and the
  // first kPredsAtOnce of them come back with the count: one wait instead of two for any ordinary schema.
  constexpr uint32_t kPredsAtOnce = 1024;
  DevBuf<uint32_t> d_preds(pool, (size_t)n_terms + 1);
  uint32_t np = 0;
  {
    DevBuf<unsigned long long> d_nv(pool, 1);
    DX_HIP(hipMemsetAsync(d_nv.p, 0, 8, s));
    if (n_terms)
      hipLaunchKernelGGL(k_list_preds, dim3(std::min(blocks(n_terms), 1024u)), dim3(256), 0, s, is_pred, pred_index.p, ps.term_flags.p, n_terms, d_preds.p, d_nv.p);
    const uint32_t at_once = (uint32_t)std::min<uint64_t>(kPredsAtOnce, n_terms);
    std::vector<uint32_t> hp(at_once);
    unsigned long long nv = 0;
    fetch.add(&np, d_total.p, 4);
    fetch.add(hp.data(), d_preds.p, (size_t)at_once * 4);
    fetch.add(&nv, d_nv.p, 8);
    fetch.sync();
    if (np > at_once) {
      hp.resize(np);
      fetch.add(hp.data() + at_once, d_preds.p + at_once, (size_t)(np - at_once) * 4);
      fetch.sync();
    }
    hp.resize(np);
    st.n_predicates = np;
    ps.pred_terms.assign(hp.begin(), hp.end());
    st.n_vertices = nv;
  }
  // The predicates' strings are wanted by the host at the very end. Their lengths are gathered now and their bytes
  // packed after the split, so that both come back with synchronisations that happen anyway (a schema with ten
  // thousand predicates would otherwise queue twenty thousand small copies, or wait twice more).
  std::vector<uint64_t> pred_off((size_t)np + 1, 0);
  std::string pred_bytes;
  DevBuf<uint64_t> d_pred_src(pool, np), d_pred_off(pool, (size_t)np + 1);
  if (np) {
    DevBuf<uint32_t> d_len(pool, (size_t)np + 1);
    DevBuf<uint64_t> d_tot(pool, 1);
    DX_HIP(hipMemsetAsync(d_len.p + np, 0, 4, s));
    hipLaunchKernelGGL(k_term_spans, dim3(blocks(np)), dim3(256), 0, s, d_preds.p, np, d.pool_off.p, d_pred_src.p, d_len.p);
    exclusive_scan_u32_u64(pool, d_len.p, d_pred_off.p, (size_t)np + 1, d_tot.p, s);
    fetch.add(pred_off.data(), d_pred_off.p, ((size_t)np + 1) * 8);
  }
  d_preds.reset();
  const int bits_term = bits_for(n_terms ? n_terms : 1), bits_pred = bits_for(np ? np : 1);
  if (bits_for(nparts) + bits_term + 1 + bits_pred > 64)
    throw DxError(DX_E_RANGE, "dictionary too large for 64-bit packed VP index entries on one GPU");
  // (predicate index, subject, object) in one word: 64 bits hold every LUBM size (5 + 2 x 28 bits at LUBM-1300); a
  // schema of tens of thousands of predicates over hundreds of millions of terms takes 128-bit words, sorted by the
  // same passes with twice the bytes. DX_FORCE_WIDE_KEYS=1 takes that path on any input (how the tests reach it).
  ps.wide_keys = bits_pred + 2 * bits_term > 64 || (getenv("DX_FORCE_WIDE_KEYS") && *getenv("DX_FORCE_WIDE_KEYS") == '1');
  ps.bits_term = bits_term;
  for (uint32_t i = 0; i < np; ++i)
    if (ps.pred_terms[i] == type_term) ps.type_pidx = i;
  ps.n_uniq_slots = n_rec;
  auto pack_route_sort = [&](auto& uniq, auto& local_keys) {
    using KBuf = std::remove_reference_t<decltype(uniq)>;
    using K = std::remove_pointer_t<decltype(uniq.p)>;
    uniq = KBuf(pool, n_rec);
    if (n_rec) hipLaunchKernelGGL((k_pack_keys<K>), dim3(blocks(n_rec)), dim3(256), 0, s, S.p, P.p, O.p, n_rec, pred_index.p, bits_term, uniq.p);
    S.reset(); P.reset(); O.reset(); kind.reset(); fact.reset();
    if (sharded) {
      // every triple goes to the ranks that own one of its partitions (CustomPartitioner, DAP.scala:541-551);
      // what arrives is sorted, de-duplicated across ranks and placed like the triples of a single GPU
      DevBuf<uint64_t> owners(pool, n_rec);
      DevBuf<unsigned long long> counts(pool, 2 * DX_MAX_PARTITIONS);
      DX_HIP(hipMemsetAsync(counts.p, 0, sizeof(unsigned long long) * 2 * DX_MAX_PARTITIONS, s));
      if (n_rec)
        hipLaunchKernelGGL((k_route_count<K>), dim3(std::min(blocks(n_rec), 4096u)), dim3(256), 0, s, uniq.p, n_rec, bits_term, ps.type_pidx,
                           ps.term_flags.p, Pm.p, M.p, cm.world, owners.p, counts.p);
      std::vector<unsigned long long> h(cm.world);
      fetch.add(h.data(), counts.p, sizeof(unsigned long long) * cm.world);
      fetch.sync();
      std::vector<uint64_t> send(cm.world), recv(cm.world);
      std::vector<unsigned long long> cur(cm.world);
      unsigned long long run = 0;
      for (int p = 0; p < cm.world; ++p) { send[p] = h[p]; cur[p] = run; run += h[p]; }
      if (run >= (1ull << 32)) throw DxError(DX_E_RANGE, "more than 2^32 routed triples on one GPU");  // reaches the checkpoint below from the handler
      DX_HIP(hipMemcpyAsync(counts.p + DX_MAX_PARTITIONS, cur.data(), sizeof(unsigned long long) * cm.world, hipMemcpyHostToDevice, s));
      KBuf outbox(pool, run);
      if (n_rec)
        hipLaunchKernelGGL((k_route_scatter<K>), dim3(std::min(blocks(n_rec), 4096u)), dim3(256), 0, s, uniq.p, owners.p, n_rec, cm.world,
                           counts.p + DX_MAX_PARTITIONS, outbox.p);
      DX_HIP(hipStreamSynchronize(s));
      DX_HIP(hipGetLastError());
      checkpoint();
      comm_ok(cm.exchange_host(cm.user, send.data(), recv.data(), 1), "exchange_host");
      uint64_t n_in = 0;
      for (int p = 0; p < cm.world; ++p) n_in += recv[p];
      {  // an inbox of 2^32 triples is this rank's problem alone: all ranks hear of it before the triples move
        LocalError e;
        if (n_in >= (1ull << 32)) { e.code = DX_E_RANGE; e.msg = "more than 2^32 triples arrive on one GPU"; }
        agree_or_throw(ctx, e);
      }
      KBuf inbox(pool, n_in);
      comm_ok(cm.exchange(cm.user, outbox.p, send.data(), inbox.p, recv.data(), (int)sizeof(K)), "exchange of the triples");
      local_keys = std::move(uniq);
      ps.n_local_keys = n_rec;
      uniq = std::move(inbox);
      n_rec = (uint32_t)n_in;
      ps.n_uniq_slots = n_rec;
    }
    radix_sort_keys(pool, uniq, n_rec, bits_pred + 2 * bits_term, s);
  };
  if (ps.wide_keys) pack_route_sort(ps.uniq_w, ps.local_keys_w); else pack_route_sort(ps.uniq, ps.local_keys);
  tm.mark();

  // ---- 7. placement masks
  DevBuf<uint64_t> mask(pool, n_rec);
  DevBuf<uint8_t> node_flag(pool, n_rec);
  DevBuf<unsigned long long> counters(pool, 3);
  DX_HIP(hipMemsetAsync(counters.p, 0, 24, s));
  if (n_rec)
    DX_WITH_KEYS(hipLaunchKernelGGL((k_place<K>), dim3(std::min(blocks(n_rec), 1024u)), dim3(256), 0, s, keys, n_rec, bits_term, ps.type_pidx, ps.term_flags.p, Pm.p, M.p,
                       owned, mask.p, node_flag.p, counters.p));
  tm.mark();

  // ---- 8. stable split by partition
  const unsigned sp_tiles = (unsigned)((n_rec + SP_TILE - 1) / SP_TILE) ? (unsigned)((n_rec + SP_TILE - 1) / SP_TILE) : 1;
  DevBuf<uint32_t> sp_counts(pool, (size_t)nparts * sp_tiles + 1);
  DX_HIP(hipMemsetAsync(sp_counts.p, 0, sp_counts.bytes(), s));
  if (n_rec) hipLaunchKernelGGL(k_split_count, dim3(sp_tiles), dim3(SP_THREADS), 0, s, mask.p, n_rec, nparts, sp_counts.p, sp_tiles);
  exclusive_scan_u32_u32(pool, sp_counts.p, sp_counts.p, (size_t)nparts * sp_tiles + 1, nullptr, s);
  std::vector<uint32_t> h_part_base(nparts + 1);
  DevBuf<uint32_t> d_part_base(pool, nparts + 1);  // where each partition's rows begin: column 0 of every row of the scanned matrix
  hipLaunchKernelGGL(k_pick_strided, dim3(1), dim3(128), 0, s, sp_counts.p, (size_t)sp_tiles, nparts + 1, d_part_base.p);
  fetch.add(h_part_base.data(), d_part_base.p, (nparts + 1) * 4);
  unsigned long long h_counters[3] = {0, 0, 0};
  fetch.add(h_counters, counters.p, 24);
  // the node index's positions do not depend on the split: their count comes back with the same synchronisation
  DevBuf<uint32_t> node_pos(pool, n_rec);
  exclusive_scan_u8_u32(pool, node_flag.p, node_pos.p, n_rec, d_total.p, s);
  uint32_t n_node = 0;
  fetch.add(&n_node, d_total.p, 4);
  fetch.sync();
  // Row positions of the split are 32-bit; the scan above has already wrapped if this GPU holds more rows than that,
  // so stop here, before anything is scattered by those positions. Under a communicator every rank learns of it.
  uint64_t too_many_rows = h_counters[2] >= (1ull << 32) ? 1 : 0;
  if (sharded) {  // placed distinct triples / edges of the whole input, each counted by one rank
    uint64_t v[3] = {h_counters[0], h_counters[1], too_many_rows};
    comm_ok(cm.allreduce_sum(cm.user, v, 3), "allreduce_sum");
    h_counters[0] = v[0];
    h_counters[1] = v[1];
    too_many_rows = v[2];
  }
  if (too_many_rows) {  // with several ranks the sum above has told all of them
    if (sharded) throw AgreedError(DX_E_RANGE, "more than 2^32 rows over the first-level partitions of one GPU");
    throw DxError(DX_E_RANGE, "more than 2^32 rows over the first-level partitions of one GPU");
  }
  st.n_unique = h_counters[0];
  st.n_edges = h_counters[1];
  const uint32_t n_rows = h_part_base[nparts];
  st.n_rows = n_rows;
  ps.part_base.assign(h_part_base.begin(), h_part_base.end());
  ps.row_s = DevBuf<int64_t>(pool, n_rows);
  ps.row_o = DevBuf<int64_t>(pool, n_rows);
  ps.row_p = DevBuf<uint32_t>(pool, n_rows);
  if (n_rec)
    DX_WITH_KEYS(hipLaunchKernelGGL((k_split_scatter<K>), dim3(sp_tiles), dim3(SP_THREADS), 0, s, keys, mask.p, n_rec, nparts, bits_term, sp_counts.p,
                       sp_tiles, ps.row_s.p, ps.row_o.p, ps.row_p.p));
  if (np && pred_off[np]) {
    DevBuf<uint8_t> packed(pool, pred_off[np] + 16);
    hipLaunchKernelGGL(k_pool_copy, dim3(blocks(4ull * np)), dim3(256), 0, s, d.pool.p, d_pred_src.p, d_pred_off.p, (uint64_t)np, packed.p);
    pred_bytes.resize(pred_off[np]);
    fetch.add(&pred_bytes[0], packed.p, pred_off[np]);
    fetch.flush();  // packed ends here
  }
  std::vector<uint32_t> h_tstart;
  {
    const size_t nt = (size_t)nparts * (np ? np : 1);
    DevBuf<uint32_t> tstart(pool, nt);
    DX_HIP(hipMemsetAsync(tstart.p, 0xFF, nt * 4, s));
    if (n_rows) hipLaunchKernelGGL(k_table_starts, dim3(std::min(blocks(n_rows, 1024), 2048u)), dim3(256), 0, s, ps.row_p.p, d_part_base.p, nparts, np, tstart.p);
    h_tstart.resize(nt);
    fetch.add(h_tstart.data(), tstart.p, nt * 4);  // wanted at the very end: comes back with a later synchronisation
    fetch.flush();  // tstart ends here
  }
  tm.mark();

  // ---- 9. node index
  ps.n_node = n_node;
  st.n_node_index = n_node;
  ps.node_s = DevBuf<int64_t>(pool, n_node);
  ps.node_o = DevBuf<int64_t>(pool, n_node);
  if (n_rec) DX_WITH_KEYS(hipLaunchKernelGGL((k_node_index<K>), dim3(blocks(n_rec)), dim3(256), 0, s, keys, node_flag.p, node_pos.p, n_rec, bits_term,
                                ps.node_s.p, ps.node_o.p, ps.term_flags.p));
  if (sharded && n_terms) {  // "this uri has a node-index row" has to hold on every rank, whoever keeps the row
    DevBuf<uint8_t> node_bit(pool, n_terms);
    hipLaunchKernelGGL(k_node_bit_out, dim3(blocks(n_terms)), dim3(256), 0, s, ps.term_flags.p, (uint64_t)n_terms, node_bit.p);
    DX_HIP(hipStreamSynchronize(s));
    comm_ok(cm.allreduce_sum_dev(cm.user, node_bit.p, n_terms, 1), "allreduce of the node flags");
    hipLaunchKernelGGL(k_node_bit_in, dim3(blocks(n_terms)), dim3(256), 0, s, node_bit.p, (uint64_t)n_terms, ps.term_flags.p);
    DX_HIP(hipStreamSynchronize(s));
  }
  tm.mark();
  node_pos.reset(); node_flag.reset(); mask.reset();

  // ---- 10. VP index
  ps.vpi_bits_term = bits_term;
  ps.vpi_bits_pred = bits_pred;
  ps.vpi_s_base.assign(nparts + 1, 0);
  ps.vpi_o_base.assign(nparts + 1, 0);
  std::vector<uint32_t> hvb(nparts + 1, 0);
  uint64_t n_s_host = 0;
  if (n_rows) {
    DevBuf<uint8_t> es(pool, n_rows), eo(pool, n_rows);
    const unsigned vpi_blocks = blocks(((uint64_t)n_rows + 3) / 4);  // 1024 rows each
    DevBuf<uint32_t> block_s(pool, vpi_blocks + 1), block_o(pool, vpi_blocks + 1);
    DevBuf<uint8_t> class_used(pool, n_terms);
    DevBuf<unsigned long long> n_type_ent(pool, 2);
    DevBuf<uint32_t> totals(pool, 2);
    DX_HIP(hipMemsetAsync(class_used.p, 0, class_used.bytes(), s));
    DX_HIP(hipMemsetAsync(n_type_ent.p, 0, 16, s));
    hipLaunchKernelGGL(k_vpi_flags, dim3(vpi_blocks), dim3(256), 0, s, ps.row_s.p, ps.row_o.p, ps.row_p.p, d_part_base.p, nparts,
                       ps.type_pidx, ps.term_flags.p, es.p, eo.p, class_used.p, block_s.p, block_o.p, bits_pred <= 26 ? 1 : 0);
    exclusive_scan_u32_u32(pool, block_s.p, block_s.p, vpi_blocks, totals.p, s);
    exclusive_scan_u32_u32(pool, block_o.p, block_o.p, vpi_blocks, totals.p + 1, s);
    hipLaunchKernelGGL(k_vpi_type_entries<uint64_t>, dim3(std::min(blocks(n_terms), 1024u)), dim3(256), 0, s, class_used.p, Pm.p, owned, n_terms, ps.type_pidx,
                       bits_term, bits_pred, n_type_ent.p, (uint64_t*)nullptr);
    DevBuf<uint32_t> sb(pool, nparts + 1);
    hipLaunchKernelGGL(k_vpi_s_bases, dim3(nparts + 1), dim3(256), 0, s, block_s.p, es.p, d_part_base.p, nparts, n_rows, totals.p, sb.p);
    uint32_t h_tot[2] = {0, 0};
    unsigned long long h_type_ent = 0;
    std::vector<uint32_t> hsb(nparts + 1);
    fetch.add(h_tot, totals.p, 8);
    fetch.add(&h_type_ent, n_type_ent.p, 8);
    fetch.add(hsb.data(), sb.p, (nparts + 1) * 4);
    fetch.sync();
    const uint32_t n_s = h_tot[0], n_o_raw = h_tot[1] + (uint32_t)h_type_ent;
    for (int q = 0; q <= nparts; ++q) ps.vpi_s_base[q] = hsb[q];
    ps.vpi_s = DevBuf<uint64_t>(pool, n_s);
    // The object entries are written, sorted and made distinct in words just wide enough for them: LUBM-100 needs
    // 31 bits (3 partition + 22 term + 1 role + 5 predicate), and half-width keys halve the traffic of the sort.
    const int ent_bits = bits_for(nparts) + bits_term + 1 + bits_pred;
    auto object_entries = [&](auto zero) {
      using K = decltype(zero);
      DevBuf<K> ent(pool, n_o_raw);
      hipLaunchKernelGGL(k_vpi_write<K>, dim3(vpi_blocks), dim3(256), 0, s, ps.row_s.p, ps.row_o.p, ps.row_p.p, d_part_base.p, nparts, bits_term,
                         bits_pred, es.p, eo.p, block_s.p, block_o.p, ps.vpi_s.p, ent.p);
      if (h_type_ent)
        hipLaunchKernelGGL(k_vpi_type_entries<K>, dim3(std::min(blocks(n_terms), 1024u)), dim3(256), 0, s, class_used.p, Pm.p, owned, n_terms, ps.type_pidx,
                           bits_term, bits_pred, n_type_ent.p + 1, ent.p + h_tot[1]);
      es.reset(); eo.reset(); block_s.reset(); block_o.reset();
      if constexpr (sizeof(K) == 4) radix_sort_u32(pool, ent, n_o_raw, ent_bits, s);
      else radix_sort_u64(pool, ent, n_o_raw, ent_bits, s);
      const unsigned dblocks = blocks(((uint64_t)n_o_raw + 3) / 4);  // 1024 entries each
      DevBuf<uint32_t> dcnt(pool, (size_t)dblocks + 1);
      if (n_o_raw) hipLaunchKernelGGL(k_distinct_count<K>, dim3(dblocks), dim3(256), 0, s, ent.p, n_o_raw, dcnt.p);
      exclusive_scan_u32_u32(pool, dcnt.p, dcnt.p, n_o_raw ? dblocks : 0, d_total.p, s);
      // room for every entry: how many are distinct comes back with the call's last wait (the array is a few bytes per
      // row of the partitions; an exact size would cost a host round trip here)
      ps.vpi_o = DevBuf<uint64_t>(pool, n_o_raw);
      if (n_o_raw) hipLaunchKernelGGL(k_distinct_compact<K>, dim3(dblocks), dim3(256), 0, s, ent.p, n_o_raw, dcnt.p, ps.vpi_o.p);
    };
    if (ent_bits <= 32) object_entries((uint32_t)0); else object_entries((uint64_t)0);
    DevBuf<uint32_t> vb(pool, nparts + 1);
    hipLaunchKernelGGL(k_vpi_bases, dim3(1), dim3(128), 0, s, ps.vpi_o.p, d_total.p, bits_term + 1 + bits_pred, nparts, vb.p);
    fetch.add(hvb.data(), vb.p, (nparts + 1) * 4);
    fetch.flush();  // vb ends here
    n_s_host = n_s;
  }
  tm.mark();
  fetch.sync();
  DX_HIP(hipGetLastError());
  checkpoint();  // every rank got through: all return success together
  if (n_rows) {
    for (int q = 0; q <= nparts; ++q) ps.vpi_o_base[q] = hvb[q];
    st.n_vp_index = n_s_host + hvb[nparts];  // the last base is the number of distinct object entries
  }

  // ---- host-side catalogue of tables
  {
    for (uint32_t i = 0; i < np; ++i) {
      ps.pred_strings.emplace_back(pred_bytes.substr(pred_off[i], pred_off[i + 1] - pred_off[i]));
      ps.pred_storage.push_back(uri_to_storage(ps.pred_strings[i]));
    }
    std::vector<uint8_t> pred_in_layout(np);
    for (uint32_t p = 0; p < np; ++p) pred_in_layout[p] = utf16_units(ps.pred_strings[p]) < 150;  // DAP.scala:434
    for (int q = 0; q < nparts; ++q) {  // tables in (partition, predicate index) order
      const size_t first = ps.tables.size();
      uint64_t end = h_part_base[q + 1];
      for (int64_t p = (int64_t)np - 1; p >= 0; --p) {  // a table ends where the next one present begins
        const uint32_t stt = h_tstart[(size_t)q * np + p];
        if (stt == kNil) continue;
        VpTable t;
        t.partition = q; t.pidx = (uint32_t)p; t.pred_term = ps.pred_terms[p];
        t.begin = stt; t.rows = end - stt;
        t.in_layout = pred_in_layout[p] != 0;
        ps.tables.push_back(t);
        end = stt;
      }
      std::reverse(ps.tables.begin() + first, ps.tables.end());
    }
    st.n_tables = (uint32_t)ps.tables.size();
  }
  // ---- stage times
  DX_HIP(hipEventSynchronize(tm.ev.back()));  // passed on the device before the last results left it
  std::vector<float> gaps;
  for (size_t i = 1; i < tm.ev.size(); ++i) {
    float ms = 0;
    DX_HIP(hipEventElapsedTime(&ms, tm.ev[i - 1], tm.ev[i]));
    gaps.push_back(ms);
  }
  // event layout: [0]start [1]record starts+tokenise+dictionary [2]ids+encode [3]pool [4]masks [5]sort [6]place [7]split [8]node [9]vpi
  st.ms_stage[0] = ms_scan;  // the two-pass form's first pass; the fused form has none
  st.ms_stage[1] = ms_tok;
  st.ms_stage[2] = gaps[0] + gaps[1] - ms_scan - ms_tok;
  for (int i = 3; i <= 9; ++i) st.ms_stage[i] = gaps[i - 1];
  float total = 0;
  DX_HIP(hipEventElapsedTime(&total, tm.ev.front(), tm.ev.back()));
  st.ms_total = total;
  st.ms_stage[10] = ms_clear;
  st.ms_stage[11] = ms_claim;
  ps.valid = true;
}

}  // namespace dx
