"""Host-side stand-in for the reference's SPARQL front end (no JVM needed).

The reference keeps SPARQL parsing, partition discovery and SQL translation in
Scala on the host (QueryParser.scala, QueryTranslator.scala); the GPU engine only
consumes the translated-query text files they write. This module restates that
front end so translated files can be produced where no JVM exists. It follows

  * QueryParser.parseQuery / extractTriplePatterns / extractQueryMappings
    (QueryParser.scala:18-124, cleanTp :135-155),
  * QueryTranslator.buildQueryIndex / mapSubPartitions (QueryTranslator.scala:126-241),
  * QueryTranslator.translateTp and the SQL templates (:292-393, :417-543),
  * the file format written by QueryTranslator.main (:93-108).

Iteration orders that the reference leaves to scala.collection.mutable.HashMap
(variable visiting order, `partitions` line order) are reproduced by emulating
Scala 2.12's HashTable index function, so the output is byte-identical to the
translated Q2 example the reference carries (QueryOptimizer.scala:282).
"""
from __future__ import annotations

import re
from typing import Callable, Iterable

RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"


# --------------------------------------------------------------------------
# Scala 2.12 collection orders
# --------------------------------------------------------------------------
def java_string_hash(s: str) -> int:
    h = 0
    for unit in _utf16_units(s):
        h = (31 * h + unit) & 0xFFFFFFFF
    return h


def _utf16_units(s: str):
    b = s.encode("utf-16-be", "surrogatepass")
    return [(b[i] << 8) | b[i + 1] for i in range(0, len(b), 2)]


def _byteswap32(v: int) -> int:  # scala.util.hashing.byteswap32
    hc = (v * 0x9E3775CD) & 0xFFFFFFFF
    hc = int.from_bytes(hc.to_bytes(4, "big"), "little")
    return (hc * 0x9E3775CD) & 0xFFFFFFFF


def _table_index(hcode: int, size: int) -> int:  # mutable.HashTable.index (2.12)
    ones = size - 1
    seed = bin(ones).count("1")
    exponent = 32 - ones.bit_length()
    i = _byteswap32(hcode & 0xFFFFFFFF)
    rot = seed % 32
    rotated = ((i >> rot) | (i << (32 - rot))) & 0xFFFFFFFF
    return (rotated >> exponent) & ones


def scala_mutable_hashmap_order(keys: Iterable, hash_of: Callable[[object], int]) -> list:
    """Iteration order of a scala.collection.mutable.HashMap holding `keys`
    (inserted in the given order): buckets from the highest index down, newest
    entry first within a bucket; the table starts at 16 and doubles above 75 %."""
    keys = list(keys)
    size = 16
    while len(keys) > size * 3 // 4:
        size *= 2
    buckets: dict[int, list] = {}
    for k in keys:
        buckets.setdefault(_table_index(hash_of(k), size), []).insert(0, k)
    out = []
    for idx in range(size - 1, -1, -1):
        out.extend(buckets.get(idx, []))
    return out


def _improve_immutable(h: int) -> int:  # immutable.HashMap.improve (2.12)
    h &= 0xFFFFFFFF
    h = (h + (~(h << 9) & 0xFFFFFFFF)) & 0xFFFFFFFF
    h ^= h >> 14
    h = (h + (h << 4)) & 0xFFFFFFFF
    return h ^ (h >> 10)


def scala_immutable_order(keys: Iterable, hash_of: Callable[[object], int]) -> list:
    """Iteration order of a scala.collection.immutable Map / Set built by adding
    `keys` in the given order: Map1..Map4 / Set1..Set4 keep insertion order, larger
    ones are hash tries walked five hash bits at a time from the low end."""
    keys = list(dict.fromkeys(keys))
    if len(keys) <= 4:
        return keys
    return sorted(keys, key=lambda k: [(_improve_immutable(hash_of(k)) >> s) & 31 for s in range(0, 35, 5)])


def _int_order(keys):
    return scala_mutable_hashmap_order(keys, lambda k: k)


def _str_order(keys):
    return scala_mutable_hashmap_order(keys, java_string_hash)


# --------------------------------------------------------------------------
# QueryParser
# --------------------------------------------------------------------------
def is_variable(s: str) -> bool:
    return "?" in s


def is_literal(s: str) -> bool:
    return '"' in s


def remove_non_alnum(s: str) -> str:
    return re.sub("[^a-zA-Z0-9]", "", s)


def clean_tp(tokens):
    out = []
    for t in tokens:
        if '"' in t or is_variable(t):
            out.append(t)
        elif not t.startswith("<") and not t.endswith("<"):
            out.append("<" + t + ">")
        elif not t.startswith("<"):
            out.append("<" + t)
        elif not t.endswith(">"):
            out.append(t + ">")
        else:
            out.append(t)
    return out


def find_tp_vars(tp):
    return [remove_non_alnum(x) for x in tp if "?" in x]


class ParsedQuery:
    def __init__(self, text: str):
        # Source.fromFile(f).getLines.mkString: line terminators vanish
        self.query = "".join(re.split("\r\n|\n|\r", text))
        self.triple_patterns = list(enumerate(self._extract_tps()))
        self.mappings, self.is_indexed = self._extract_mappings()

    def _extract_tps(self):
        s = "".join(m.group(1) for m in re.finditer(r"(\{.*?\})", self.query))[1:-1]
        if not s:
            return []
        parts = re.split(r" \.", s)
        while parts and parts[-1] == "":  # java.lang.String.split drops trailing empty strings
            parts.pop()
        tps = []
        for part in parts:
            if part.startswith("#"):
                continue
            x = clean_tp([t.strip() for t in re.split(r"\s+", part.strip())])
            if len(x) < 3:
                raise ValueError("malformed triple pattern: %r" % part)
            # The reference tests x(1) against "a" / "rdf:type" AFTER cleanTp has wrapped
            # bare tokens in <>, so the normalisation never fires (QueryParser.scala:64-72);
            # kept as written.
            if x[1] in ("a", "rdf:type"):
                tps.append((x[0], RDF_TYPE, x[2]))
            else:
                tps.append((x[0], x[1], x[2]))
        return tps

    def _extract_mappings(self):
        tp_map: dict[int, str] = {}
        type_map: dict[str, str] = {}
        for i, (s, p, o) in self.triple_patterns:
            if not is_variable(s) and not is_literal(s):
                tp_map[i] = s
            if not is_variable(o) and not is_literal(o):
                tp_map[i] = o
                if p == RDF_TYPE:
                    type_map[s] = o
        for i, (s, p, o) in self.triple_patterns:
            if is_variable(s) and is_variable(o):
                if s in type_map:
                    tp_map[i] = type_map[s]
                if o in type_map:
                    tp_map[i] = type_map[o]
            elif is_variable(s):
                if s in type_map and p != RDF_TYPE:
                    tp_map[i] = type_map[s]
            elif is_variable(o):
                if o in type_map:
                    tp_map[i] = type_map[o]
        indexed = sorted(tp_map) == [i for i, _ in self.triple_patterns]
        return tp_map, indexed


# --------------------------------------------------------------------------
# QueryTranslator
# --------------------------------------------------------------------------
def uri_to_storage(uri: str) -> str:
    return re.sub("[^A-Za-z0-9]", "_", uri).lower()


def _pad5(p: str) -> str:
    return "0" * (5 - len(p)) + p


def table_name(partition: str) -> str:
    return partition.replace("-", "_").replace("=", "_E_").strip()


def build_query_index(pq: ParsedQuery, node_types: Callable[[str], list], class_index: dict, statistics: dict,
                      sub_parts: Callable[[str, str], list] | None = None):
    """QueryTranslator.buildQueryIndex + mapSubPartitions (QueryTranslator.scala:126-241). `node_types(uri)`
    returns the distinct classes of uri in the node index; `sub_parts(partition, uri)` the entries of uri in
    that partition's VP index ("s<dir>" / "o<dir>", the `parts` column of classSubPart_index split at commas),
    which only triple patterns with a variable predicate need (the reference has the loading of that index
    commented out, :47-57, so there such patterns fail; here they work when the index is handed in). Returns
    {tp id: [partition-_3=pred, ...]} in the reference's iteration order, or None when some triple pattern
    cannot be placed (the reference prints 'cannot be indexed')."""
    order = _int_order(list(pq.mappings))
    index = {}
    for tp_id in order:
        uri = pq.mappings[tp_id]
        cls_parts = class_index.get(uri, [])
        if cls_parts:
            cands = [(_pad5(p), statistics[_pad5(p)]) for p in cls_parts]
            index[tp_id] = min(cands, key=lambda t: t[1])[0]  # minBy keeps the first minimum
        else:
            types = node_types(uri)
            if not types:
                return None
            part = None
            for t in types:
                b = class_index.get(t, ["-1"])
                if part is None:
                    part = list(b)
                else:
                    inter = [x for x in part if x in b]
                    part = inter if inter else part + list(b)
            if "-1" in part:
                return None
            cands = [(_pad5(p), statistics.get(_pad5(p), -1)) for p in part]
            index[tp_id] = min(cands, key=lambda t: t[1])[0]
    tps = dict(pq.triple_patterns)
    out = {}
    # `.map` on a value typed scala.collection.Map builds an immutable Map
    # (QueryTranslator.scala:11 imports scala.collection.Map), hence the order below.
    for tp_id in scala_immutable_order(list(index), lambda k: k):
        s, p, o = tps[tp_id]
        if is_variable(p):
            if sub_parts is None:
                raise NotImplementedError(
                    "variable-predicate triple patterns need the per-partition VP index, whose loading the "
                    "reference has commented out (QueryTranslator.scala:47-57)")
            part = index[tp_id]
            if not is_variable(s):      # :195-209
                out[tp_id] = [part + "-_3=" + uri_to_storage(e[1:]) for e in sub_parts(part, s) if e[:1] == "s"]
            elif not is_variable(o) and not is_literal(o):   # :210-224
                out[tp_id] = [part + "-_3=" + uri_to_storage(e[1:]) for e in sub_parts(part, o) if e[:1] == "o"]
            else:
                out[tp_id] = [part]     # :227-229
            continue
        out[tp_id] = [index[tp_id] + "-_3=" + uri_to_storage(p)]
    return {k: out[k] for k in scala_immutable_order(list(out), lambda k: k)}


def _single_var(partition: str, tp) -> str:
    s, p, o = tp
    if "?" in s:
        return "(SELECT s AS %s FROM %s WHERE o == '%s')" % (remove_non_alnum(s), partition, o)
    if "?" in p:
        return "(SELECT p AS %s FROM %s WHERE s == '%s' AND o == '%s')" % (remove_non_alnum(p), partition, s, o)
    return "(SELECT o AS %s FROM %s WHERE s == '%s')" % (remove_non_alnum(o), partition, s)


def _double_var(partition: str, tp) -> str:
    s, p, o = tp
    if "?" in s and "?" in p:
        return "(SELECT s AS %s, p AS %s FROM %s WHERE o == '%s')" % (
            remove_non_alnum(s), remove_non_alnum(p), partition, o)
    if "?" in p and "?" in o:
        return "(SELECT p AS %s, o AS %s FROM %s WHERE s == '%s')" % (
            remove_non_alnum(p), remove_non_alnum(o), partition, s)
    return "(SELECT s AS %s, o AS %s FROM %s)" % (remove_non_alnum(s), remove_non_alnum(o), partition)


def _three_var(partition: str, tp) -> str:
    s, _, o = tp
    return "(SELECT s AS %s, o AS %s FROM %s)" % (remove_non_alnum(s), remove_non_alnum(o), partition)


def translate_sub(partition: str, tp) -> str:
    n = sum(1 for x in tp if "?" in x)
    return _single_var(partition, tp) if n == 1 else _double_var(partition, tp) if n == 2 else _three_var(partition, tp)


def translate_sub_union(partitions: list, tp) -> str:
    """The Array[String] overloads (QueryTranslator.scala:466-542): one UNION arm per sub-partition, the
    variable predicate bound to the constant '<' + partition key + '>'. The overloads' other branches
    interpolate the array itself into the SQL text (a JVM object name), which no engine can run."""
    s, p, o = tp
    n = sum(1 for x in tp if "?" in x)
    if not partitions:
        raise NotImplementedError("variable predicate without any sub-partition: the reference emits ' )'")
    tab = lambda key: "table" + table_name(key)  # noqa: E731
    if n == 1 and "?" in p:
        v = remove_non_alnum(p)
        q = "( " + "".join(" (SELECT '<%s>' AS %s FROM %s WHERE s == '%s' AND o == '%s') UNION" % (k, v, tab(k), s, o) for k in partitions)
        return q[:-5] + " )"
    if n == 2 and "?" in s and "?" in p:
        v1, v2 = remove_non_alnum(s), remove_non_alnum(p)
        q = "( " + "".join("(SELECT s AS %s, '<%s>' AS %s FROM %s WHERE o == '%s')UNION" % (v1, k, v2, tab(k), o) for k in partitions)
        return q[:-5] + " )"
    if n == 2 and "?" in p and "?" in o:
        v1, v2 = remove_non_alnum(p), remove_non_alnum(o)
        q = "( " + "".join("(SELECT o AS %s, '<%s>' AS %s FROM %s WHERE s == '%s')UNION" % (v2, k, v1, tab(k), s) for k in partitions)
        return q[:-5] + " )"
    raise NotImplementedError("variable predicate with this combination of variables: the reference prints a JVM array name")


def _var_table_mappings(sub: str, table: str) -> dict:
    vars_ = sub[sub.index("SELECT") + 6: sub.index("FROM")]
    out = {}
    for v in vars_.split(","):
        out[v[v.index("AS") + 2:].strip()] = table
    return out


def translate_tp(pq: ParsedQuery, query_index: dict) -> str:
    """QueryTranslator.translateTp (:292-393): one SQL statement for the BGP."""
    tps = dict(pq.triple_patterns)
    aggr: dict[str, list[int]] = {}
    for i, tp in pq.triple_patterns:
        for v in find_tp_vars(tp):
            aggr.setdefault(v, []).append(i)
    visit = _str_order(list(aggr))
    visit = sorted(visit, key=lambda v: -len(aggr[v]))  # stable, like sortWith(_._2 > _._2)
    used, sub_queries = [], []
    select_vars: list[str] = []
    var_table: dict[str, str] = {}
    join_tables: dict[str, list[str]] = {}
    for v in visit:
        for i in aggr[v]:
            tp = tps[i]
            tab = "tab%d" % i
            if is_variable(tp[1]):
                sub = translate_sub_union(query_index[i], tp) + " AS " + tab
            else:
                sub = translate_sub("table" + table_name(query_index[i][0]), tp) + " AS " + tab
            join_tables.setdefault(v, []).append(tab)
            if i not in used:
                var_table.update(_var_table_mappings(sub, tab))
                for x in find_tp_vars(tp):
                    if x not in select_vars:
                        select_vars.append(x)
                sub_queries.append(sub)
                used.append(i)
    # selectVariables is an immutable Set and joinTables an immutable Map. With more than four
    # variables the visiting order above is itself unspecified in the reference (its keys hash an
    # Array by identity), so only the text for <= 4 variables is reproducible byte for byte.
    select_vars = scala_immutable_order(select_vars, java_string_hash)
    q = "SELECT " + ",".join("%s.%s AS %s" % (var_table[v], v, v) for v in select_vars) + " FROM "
    q += ", ".join(sub_queries)
    conds = []
    for v in scala_immutable_order(list(join_tables), java_string_hash):
        tables = join_tables[v]
        for a, b in zip(tables, tables[1:]):
            conds.append("%s.%s=%s.%s" % (a, v, b, v))
    if conds:
        q += " WHERE " + " AND ".join(conds) + " "
    return q


def translate_query(name: str, sparql_text: str, node_types, class_index: dict, statistics: dict, sub_parts=None):
    """Returns the text of translated_queries_N_bal/<name>, or None when the query
    cannot be indexed (QueryTranslator.scala:92-114)."""
    pq = ParsedQuery(sparql_text)
    if not pq.is_indexed:
        return None
    qi = build_query_index(pq, node_types, class_index, statistics, sub_parts)
    if qi is None:
        return None
    sql = translate_tp(pq, qi)
    partitions = []
    for parts in qi.values():
        for p in parts:
            if p not in partitions:
                partitions.append(p)
    return ">>>>> %s\nX %s\npartitions %s\nTP %d" % (name, sql, ",".join(partitions), len(pq.triple_patterns))


def load_class_index(text: str) -> dict:
    """Loader.loadClassIndexBal (Loader.scala:114-123)."""
    out = {}
    for line in text.splitlines():
        tok = line.split("\t")
        out[tok[0]] = "".join(tok[1:]).split(",")
    return out


def load_partition_stats(text: str) -> dict:
    """Loader.loadPartitionStatsBal (Loader.scala:236-247): keeps `00...` lines."""
    out = {}
    for line in text.splitlines():
        if line.startswith("00"):
            tok = line.split(" ")
            out[tok[0]] = int(tok[1])
    return out


# ---- command line: the stand-in behind scripts/run_translator.sh ------------------------------------------
def load_layout_indexes(base: str, dataset: str, n: int):
    """node_types(uri) and sub_parts(partition, uri) over a layout on disk: the Parquet files
    Loader.loadNodeIndex / loadSubPartIndex read (Loader.scala:266-276), whoever wrote them -- the Scala tools,
    dx_part_write_layout or parquet_io.export_layout."""
    import os

    from . import parquet_io
    data_dir = os.path.join(base, dataset + "_data")
    node = {}
    t = parquet_io.read_table(os.path.join(data_dir, "node_index_%d" % n))
    for u, c in zip(t.column("uri").to_pylist(), t.column("uriType").to_pylist()):
        node.setdefault(u, []).append(c)
    sub_cache = {}

    def sub_parts(part, uri):
        if part not in sub_cache:
            d = os.path.join(data_dir, "classSubPart_index_%d__%s" % (n, part))
            tt = parquet_io.read_table(d)
            sub_cache[part] = dict(zip(tt.column("uri").to_pylist(), tt.column("parts").to_pylist()))
        return [x for x in sub_cache[part].get(uri, "").split(",") if x]
    return (lambda uri: node.get(uri, [])), sub_parts


def main(argv=None) -> int:
    """QueryTranslator.main (QueryTranslator.scala:25-119): <dataset> <partitions> <base path> <sparql query path>.
    Reads <dataset>_local/class_index.txt_<N>_bal and partition_stat.txt_<N>_bal from the working directory, the
    node index (and, for variable predicates, the VP index) from the layout, and writes
    <sparql query path>/translated_queries_<N>_bal/<query file name> for every query that can be indexed."""
    import os
    import sys
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 4:
        print("-->Missing Arguments")
        print("-->Arguments: dataset name, number of partitions, hdfs base path, sparql query path")
        return 255
    dataset, n, base, input_path = argv[0], int(argv[1]), argv[2], argv[3]
    if not base.endswith("/"):
        base += "/"
    node_types, sub_parts = load_layout_indexes(base, dataset, n)
    class_index = load_class_index(open(dataset + "_local/class_index.txt_%d_bal" % n).read())
    statistics = load_partition_stats(open(dataset + "_local/partition_stat.txt_%d_bal" % n).read())
    out_dir = os.path.join(input_path, "translated_queries_%d_bal" % n)
    files = []
    for root, dirs, names in os.walk(input_path):
        dirs[:] = sorted(d for d in dirs if not d.startswith("translated_queries_"))
        files += [os.path.join(root, f) for f in sorted(names)]
    os.makedirs(out_dir, exist_ok=True)
    for path in files:
        name = os.path.basename(path)
        text = translate_query(name, open(path).read(), node_types, class_index, statistics, sub_parts)
        if text is None:
            print("Query cannot be indexed:" + name)
            continue
        print("queryName: " + name)
        with open(os.path.join(out_dir, name), "w") as f:
            f.write(text)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
