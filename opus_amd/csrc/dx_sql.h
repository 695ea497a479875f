// Parser for the translated-query SQL subset (everything QueryTranslator / QueryOptimizer emit).
#pragma once
#include <string>
#include <vector>

namespace dx {

// One `(SELECT ... FROM table [WHERE ...])`. A projected column comes from s, o or a quoted constant.
struct SqlArm {
  std::string table;
  struct Proj { int src; std::string constant; std::string name; };  // src: 0 = s, 1 = o, 2 = constant
  std::vector<Proj> proj;
  bool has_s = false, has_o = false;
  std::string s_const, o_const;
};
// `( arm ) AS tabK` or `( (arm) UNION (arm) ... ) AS tabK`
struct SqlSub {
  std::string alias;
  std::vector<SqlArm> arms;
};
struct SqlQuery {
  struct Item { std::string alias, col, name; };
  struct Cond { std::string a_alias, a_col, b_alias, b_col; };
  std::vector<Item> select;
  std::vector<SqlSub> subs;
  std::vector<Cond> conds;
};

// Several statements in one text are separated by this character (dx_query_joined puts them together that way); what
// comes back is then the one statement that joins their results as the reference's customJoin does.
constexpr char kStatementSeparator = '\x1e';
// Throws DxError(DX_E_MALFORMED / DX_E_UNSUPPORTED).
SqlQuery parse_sql(const std::string& text);

}  // namespace dx
