// Recursive-descent parser for the SQL the reference's translator writes.
//
// Grammar (QueryTranslator.scala:363-392 for the outer statement, :417-543 for the sub-selects;
// QueryOptimizer.scala:163-233 only reorders FROM and rewrites WHERE inside the same grammar):
//
//   query  := SELECT item {, item} FROM sub {, sub} [WHERE cond {AND cond}]
//   item   := alias . col AS name
//   sub    := ( arm ) AS alias  |  ( ( arm ) UNION ( arm ) {UNION ( arm )} ) AS alias
//   arm    := SELECT proj {, proj} FROM table [WHERE filt {AND filt}]      (arms may carry their own parentheses)
//   proj   := (s | o | 'constant') AS name
//   filt   := (s | o) == 'constant'
//   cond   := alias . col = alias . col
//
// Keywords are matched case-insensitively, like Spark SQL does. Quoted constants are raw N-Triples
// terms and cannot contain a single quote (QueryTranslator.isValidQuery, :637-643).
#include "dx_sql.h"

#include <cctype>

#include "dx_internal.h"

namespace dx {
namespace {

struct Tok {
  enum Kind { Ident, Str, Sym, End } kind;
  std::string text;
};

class Lexer {
 public:
  explicit Lexer(const std::string& s) : s_(s) { advance(); }
  const Tok& peek() const { return cur_; }
  Tok next() { Tok t = cur_; advance(); return t; }
  bool is_sym(const char* sym) const { return cur_.kind == Tok::Sym && cur_.text == sym; }
  bool is_kw(const char* kw) const {
    if (cur_.kind != Tok::Ident) return false;
    const std::string& t = cur_.text;
    size_t i = 0;
    for (; kw[i] && i < t.size(); ++i)
      if (std::toupper((unsigned char)t[i]) != kw[i]) return false;
    return !kw[i] && i == t.size();
  }
  [[noreturn]] void fail(const std::string& what) const {
    throw DxError(DX_E_MALFORMED, "SQL parse error near offset " + std::to_string(pos_) + ": " + what);
  }
  void expect_sym(const char* sym) { if (!is_sym(sym)) fail(std::string("expected '") + sym + "'"); advance(); }
  void expect_kw(const char* kw) { if (!is_kw(kw)) fail(std::string("expected ") + kw); advance(); }
  std::string expect_ident() { if (cur_.kind != Tok::Ident) fail("expected identifier"); return next().text; }

 private:
  void advance() {
    while (pos_ < s_.size() && std::isspace((unsigned char)s_[pos_])) ++pos_;
    if (pos_ >= s_.size()) { cur_ = {Tok::End, ""}; return; }
    const char c = s_[pos_];
    if (c == '\'') {
      const size_t e = s_.find('\'', pos_ + 1);
      if (e == std::string::npos) fail("unterminated string constant");
      cur_ = {Tok::Str, s_.substr(pos_ + 1, e - pos_ - 1)};
      pos_ = e + 1;
    } else if (std::isalnum((unsigned char)c) || c == '_') {
      size_t e = pos_;
      while (e < s_.size() && (std::isalnum((unsigned char)s_[e]) || s_[e] == '_')) ++e;
      cur_ = {Tok::Ident, s_.substr(pos_, e - pos_)};
      pos_ = e;
    } else if (c == '=' && pos_ + 1 < s_.size() && s_[pos_ + 1] == '=') {
      cur_ = {Tok::Sym, "=="};
      pos_ += 2;
    } else if (c == '(' || c == ')' || c == ',' || c == '.' || c == '=') {
      cur_ = {Tok::Sym, std::string(1, c)};
      ++pos_;
    } else {
      fail(std::string("unexpected character '") + c + "'");
    }
  }
  const std::string& s_;
  size_t pos_ = 0;
  Tok cur_{Tok::End, ""};
};

SqlArm parse_arm(Lexer& lx) {
  SqlArm arm;
  lx.expect_kw("SELECT");
  for (;;) {
    SqlArm::Proj p;
    if (lx.peek().kind == Tok::Str) {
      p.src = 2;
      p.constant = lx.next().text;
    } else {
      const std::string col = lx.expect_ident();
      if (col == "s") p.src = 0;
      else if (col == "o") p.src = 1;
      else if (col == "p")
        throw DxError(DX_E_UNSUPPORTED, "projection of column p: vertical-partition tables only have s and o "
                                        "(the reference's Spark job fails on this statement as well)");
      else lx.fail("unknown column '" + col + "'");
    }
    lx.expect_kw("AS");
    p.name = lx.expect_ident();
    arm.proj.push_back(p);
    if (lx.is_sym(",")) { lx.next(); continue; }
    break;
  }
  lx.expect_kw("FROM");
  arm.table = lx.expect_ident();
  if (lx.is_kw("WHERE")) {
    lx.next();
    for (;;) {
      const std::string col = lx.expect_ident();
      if (!lx.is_sym("==") && !lx.is_sym("=")) lx.fail("expected '=='");
      lx.next();
      if (lx.peek().kind != Tok::Str) lx.fail("expected quoted constant");
      const std::string c = lx.next().text;
      if (col == "s") { if (arm.has_s && arm.s_const != c) arm.s_const = "\x01never"; else arm.s_const = c; arm.has_s = true; }
      else if (col == "o") { if (arm.has_o && arm.o_const != c) arm.o_const = "\x01never"; else arm.o_const = c; arm.has_o = true; }
      else lx.fail("filter on unknown column '" + col + "'");
      if (lx.is_kw("AND")) { lx.next(); continue; }
      break;
    }
  }
  return arm;
}

SqlSub parse_sub(Lexer& lx) {
  SqlSub sub;
  lx.expect_sym("(");
  if (lx.is_sym("(")) {
    for (;;) {
      lx.expect_sym("(");
      sub.arms.push_back(parse_arm(lx));
      lx.expect_sym(")");
      if (lx.is_kw("UNION")) { lx.next(); continue; }
      break;
    }
  } else {
    sub.arms.push_back(parse_arm(lx));
  }
  lx.expect_sym(")");
  lx.expect_kw("AS");
  sub.alias = lx.expect_ident();
  for (size_t i = 1; i < sub.arms.size(); ++i) {
    if (sub.arms[i].proj.size() != sub.arms[0].proj.size()) lx.fail("UNION arms project different column counts");
    // SQL names the columns of a UNION after its first arm
    for (size_t j = 0; j < sub.arms[i].proj.size(); ++j) sub.arms[i].proj[j].name = sub.arms[0].proj[j].name;
  }
  return sub;
}

}  // namespace

// The statements of one query file, joined the way QueryProcessor.executeFinalQuery joins their results
// (query_processor/QueryProcessor.scala:184-208, 276-278): left to right, each time on the output columns the two
// sides share (Dataset.join(right, commonCols, "inner")), or as a cross product when they share none. A statement's
// result is a bag and so is the join, so the join of the results equals ONE statement over all sub-selects of all
// statements with an equality added for every shared output column; that statement is what this returns. Its output
// columns come in Spark's order: the shared ones, then the left side's others, then the right side's others.
static SqlQuery join_statements(const std::vector<SqlQuery>& parts) {
  SqlQuery acc;
  for (size_t k = 0; k < parts.size(); ++k) {
    SqlQuery r = parts[k];
    const std::string prefix = "q" + std::to_string(k) + "_";  // the statements number their sub-selects alike (tab0, tab1, ...)
    for (SqlSub& sub : r.subs) sub.alias = prefix + sub.alias;
    for (SqlQuery::Item& it : r.select) it.alias = prefix + it.alias;
    for (SqlQuery::Cond& c : r.conds) { c.a_alias = prefix + c.a_alias; c.b_alias = prefix + c.b_alias; }
    if (k == 0) { acc = std::move(r); continue; }
    std::vector<SqlQuery::Item> shared, left_only, right_only;
    for (const SqlQuery::Item& l : acc.select) {
      const SqlQuery::Item* same = nullptr;
      for (const SqlQuery::Item& x : r.select)
        if (x.name == l.name) same = &x;
      if (same) {
        shared.push_back(l);
        acc.conds.push_back({l.alias, l.col, same->alias, same->col});
      } else {
        left_only.push_back(l);
      }
    }
    for (const SqlQuery::Item& x : r.select) {
      bool in_left = false;
      for (const SqlQuery::Item& l : acc.select) in_left |= l.name == x.name;
      if (!in_left) right_only.push_back(x);
    }
    acc.select = shared;
    acc.select.insert(acc.select.end(), left_only.begin(), left_only.end());
    acc.select.insert(acc.select.end(), right_only.begin(), right_only.end());
    for (SqlSub& sub : r.subs) acc.subs.push_back(std::move(sub));
    for (SqlQuery::Cond& c : r.conds) acc.conds.push_back(std::move(c));
  }
  return acc;
}

static SqlQuery parse_one(const std::string& text);

SqlQuery parse_sql(const std::string& text) {
  if (text.find(kStatementSeparator) == std::string::npos) return parse_one(text);
  std::vector<SqlQuery> parts;
  size_t at = 0;
  for (;;) {
    const size_t end = text.find(kStatementSeparator, at);
    parts.push_back(parse_one(text.substr(at, end == std::string::npos ? std::string::npos : end - at)));
    if (end == std::string::npos) break;
    at = end + 1;
  }
  return join_statements(parts);
}

static SqlQuery parse_one(const std::string& text) {
  Lexer lx(text);
  SqlQuery q;
  lx.expect_kw("SELECT");
  for (;;) {
    SqlQuery::Item it;
    it.alias = lx.expect_ident();
    lx.expect_sym(".");
    it.col = lx.expect_ident();
    lx.expect_kw("AS");
    it.name = lx.expect_ident();
    q.select.push_back(it);
    if (lx.is_sym(",")) { lx.next(); continue; }
    break;
  }
  lx.expect_kw("FROM");
  for (;;) {
    q.subs.push_back(parse_sub(lx));
    if (lx.is_sym(",")) { lx.next(); continue; }
    break;
  }
  if (lx.is_kw("WHERE")) {
    lx.next();
    for (;;) {
      SqlQuery::Cond c;
      c.a_alias = lx.expect_ident(); lx.expect_sym("."); c.a_col = lx.expect_ident();
      if (!lx.is_sym("=") && !lx.is_sym("==")) lx.fail("expected '='");
      lx.next();
      c.b_alias = lx.expect_ident(); lx.expect_sym("."); c.b_col = lx.expect_ident();
      q.conds.push_back(c);
      if (lx.is_kw("AND")) { lx.next(); continue; }
      break;
    }
  }
  if (lx.peek().kind != Tok::End) lx.fail("trailing input");
  return q;
}

}  // namespace dx
