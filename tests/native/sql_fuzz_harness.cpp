// Host-only harness for the parser of the translated-query SQL subset: one statement per input line, one verdict per
// output line. Built by tests/test_cpu_sql_fuzz.py with g++ and AddressSanitizer; the parser is plain C++.
#include <iostream>
#include <string>

#include "dx_internal.h"
#include "dx_sql.h"

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    try {
      const dx::SqlQuery q = dx::parse_sql(line);
      std::cout << "OK " << q.select.size() << " " << q.subs.size() << " " << q.conds.size() << "\n";
    } catch (const dx::DxError& e) {
      std::cout << "ERR " << e.code << "\n";
    }
  }
  return 0;
}
