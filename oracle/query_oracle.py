"""CPU restatement of the query processor's execution path (TEST INFRASTRUCTURE ONLY).

The reference hands the translated SQL text to Spark SQL
(QueryProcessor.executeFinalQuery, QueryProcessor.scala:184-195) over temp views
named after the vertical-partition tables (preloadTables :210-236, loadDataset
:308-339) and reports `result.count` (:173-176). Spark SQL itself
(org.apache.spark:spark-sql_2.12:3.0.1, pom.xml:80-85) is a third-party
dependency that is not in the reference tree and cannot run offline, so this
checker executes the same SQL text, verbatim, with SQLite over the same tables:
inner joins, equality on strings, bag semantics, UNION as set union -- the part
of SQL both engines agree on.

PARITY UNPINNED: the reference holds no expected counts or bindings for any
query (its only test is one `assertTrue(true)`, AppTest.scala:13-16); results are checked against
this restatement only.
"""
from __future__ import annotations

import re
import sqlite3


def parse_translated_file(text: str):
    """QueryProcessor.main, :68-88. Returns (query_map, partitions, tp_num)."""
    partitions, tp_num, query_map = [], -1, {}
    for line in re.split("\r\n|\n|\r", text):
        if line.startswith("TP"):
            tp_num = int(line.split(" ")[1])
        elif line.startswith("partitions"):
            partitions = line.split(" ")[1].split(",")
        elif not line.startswith(">>>>>"):
            tokens = line.split(" ")
            query_map[tokens[0]] = " ".join(tokens[1:])
    return query_map, partitions, tp_num


def clean_table_name(partition: str) -> str:
    """QueryProcessor.scala:216."""
    return "table" + partition.replace("-", "_").replace("=", "_E_").strip()


def _strip_union_parens(sql: str) -> str:
    """Spark accepts `( (SELECT ..) UNION (SELECT ..) ) AS t`; SQLite wants the
    arms of a compound select unparenthesised."""
    return re.sub(r"\(\s*(\(SELECT.*?\)(?:\s*UNION\s*\(SELECT.*?\))+)\s*\)",
                  lambda m: "(" + re.sub(r"\)\s*UNION\s*\(", " UNION ", m.group(1).strip()[1:-1]) + ")", sql)


class SqliteProcessor:
    """preloadTables + executeFinalQuery over in-memory (s, o) string tables."""

    def __init__(self, vp: dict, column_type: str = "TEXT"):
        # vp: {(partition_name, '_3=<pred>'): [(s, o), ...]}; column_type "INTEGER" for tables of term ids
        self.vp = vp
        self.column_type = column_type
        self.db = sqlite3.connect(":memory:")
        self.loaded = set()

    def preload(self, partitions):
        rows_total = 0
        for part in partitions:
            name = clean_table_name(part)
            pname, sub = part.split("-", 1)
            rows = self.vp.get((pname, sub))
            if rows is None:
                raise FileNotFoundError(part)  # spark.read.load on a missing path throws
            if name not in self.loaded:
                self.db.execute('CREATE TABLE "%s" (s %s, o %s)' % (name, self.column_type, self.column_type))
                self.db.executemany('INSERT INTO "%s" VALUES (?, ?)' % name, rows)
                # indexes and statistics change SQLite's plan, never its answer: without them its join order
                # for the six-pattern triangles takes minutes on a few hundred thousand rows
                self.db.execute('CREATE INDEX "%s_s" ON "%s" (s)' % (name, name))
                self.db.execute('CREATE INDEX "%s_o" ON "%s" (o)' % (name, name))
                self.loaded.add(name)
                self.analysed = False
            rows_total += len(rows)
        return rows_total

    def execute(self, query_map: dict):
        """Returns (columns, rows) -- rows as a list of tuples of raw terms."""
        if not getattr(self, "analysed", True):
            self.db.execute("ANALYZE")
            self.analysed = True
        if len(query_map) == 1:
            cur = self.db.execute(_strip_union_parens(next(iter(query_map.values()))))
            cols = [d[0] for d in cur.description]
            return cols, cur.fetchall()
        # QueryProcessor.executeFinalQuery (QueryProcessor.scala:184-195) with several statements: each is executed, and the
        # results are folded left to right with customJoin (:197-208): an inner join on the columns both sides have
        # (commonColumns, :276-278), or a cross join when they have none. Done here on materialised results, the way
        # it reads; bags throughout (neither side is made distinct).
        acc_cols, acc_rows = None, None
        for sql in query_map.values():
            cur = self.db.execute(_strip_union_parens(sql))
            cols = [d[0] for d in cur.description]
            rows = cur.fetchall()
            if acc_cols is None:
                acc_cols, acc_rows = cols, rows
                continue
            common = [c for c in acc_cols if c in cols]
            li = [acc_cols.index(c) for c in common]
            ri = [cols.index(c) for c in common]
            l_rest = [i for i, c in enumerate(acc_cols) if c not in common]
            r_rest = [i for i, c in enumerate(cols) if c not in common]
            by_key = {}
            for r in rows:
                by_key.setdefault(tuple(r[i] for i in ri), []).append(r)
            out = []
            for l in acc_rows:
                key = tuple(l[i] for i in li)
                for r in by_key.get(key, ()):
                    out.append(tuple(key) + tuple(l[i] for i in l_rest) + tuple(r[i] for i in r_rest))
            acc_cols = common + [acc_cols[i] for i in l_rest] + [cols[i] for i in r_rest]
            acc_rows = out
        return acc_cols, acc_rows


def run_query(vp: dict, translated_text: str):
    """Returns {'count', 'columns', 'rows' (sorted by the column names' order)}."""
    qm, partitions, tp = parse_translated_file(translated_text)
    proc = SqliteProcessor(vp)
    loaded = proc.preload(partitions)
    cols, rows = proc.execute(qm)
    order = sorted(range(len(cols)), key=lambda i: cols[i])
    cols_sorted = [cols[i] for i in order]
    rows_sorted = sorted(tuple(r[i] for i in order) for r in rows)
    return {"count": len(rows), "columns": cols_sorted, "rows": rows_sorted, "load_data": loaded, "tp": tp}


def run_query_ids(vp_ids: dict, translated_text: str, term_id, max_rows: int = 0):
    """The same statement over tables of term ids (for inputs too large to decode into Python strings):
    every quoted constant of the SQL text is replaced by term_id(constant) (-1 for a term that is not in the
    dictionary: it then matches nothing, as the string would). Only for statements whose quoted strings are
    all comparison constants (no projected constants). Returns {'count', 'load_data', 'tp'} plus, for results of at
    most max_rows rows, 'columns' and the sorted 'rows' of ids."""
    qm, partitions, tp = parse_translated_file(translated_text)
    proc = SqliteProcessor(vp_ids, column_type="INTEGER")
    loaded = proc.preload(partitions)
    qm = {k: re.sub(r"'([^']*)'", lambda m: str(int(term_id(m.group(1)))), v) for k, v in qm.items()}
    sql = _strip_union_parens(next(iter(qm.values())))
    proc.db.execute("ANALYZE")
    count = proc.db.execute("SELECT COUNT(*) FROM (%s)" % sql).fetchone()[0]
    out = {"count": count, "load_data": loaded, "tp": tp}
    if count <= max_rows:  # bindings as id tuples, columns in name order, sorted
        cur = proc.db.execute(sql)
        cols = [d[0] for d in cur.description]
        order = sorted(range(len(cols)), key=lambda i: cols[i])
        out["columns"] = [cols[i] for i in order]
        out["rows"] = sorted(tuple(r[i] for i in order) for r in cur.fetchall())
    return out
