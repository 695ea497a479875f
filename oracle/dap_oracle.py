"""CPU restatement of the partitioner's post-DAP stage (TEST INFRASTRUCTURE ONLY).

This module is the checker for the HIP partitioner: a plain-Python, string-level
restatement of what the reference computes with Spark RDD / GraphX / SQL
operators. It deliberately follows the reference's *structure* (vertices with
type and label lists, GraphX triplets, three emission streams, union, distinct)
instead of the bitmask formulation the GPU path uses, so that the two are
independent derivations of the same result. Nothing in the product path may
import it.

PARITY UNPINNED: the reference ships no tests, golden vectors or expected
outputs for this path (src/test/scala/org/ics/isl/AppTest.scala:13-16 asserts
`true`), and it cannot be built or run offline (Scala/Spark from Maven Central,
pom.xml:28-100). The restatement is pinned only by a hand-derived known-answer
example (tests/golden/appendix_c.json) worked out from the cited source lines.

Third-party semantics relied on (not vendored in the reference):
org.apache.spark:spark-core_2.12:3.1.1, spark-graphx_2.12:3.0.1,
spark-sql_2.12:3.0.1 -- `reduceByKey`, `sortByKey`, `zipWithIndex`, `join`,
`distinct`, `Graph.triplets`, `DataFrameWriter.partitionBy`, `collect_set` are
used with their documented set/bag semantics.
"""
from __future__ import annotations

import re
from collections import defaultdict

RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"  # Constatnts.scala:6

_JAVA_WS = re.compile("[ \t\n\x0b\f\r]+")  # java.util.regex "\\s+"


def java_trim(s: str) -> str:
    """java.lang.String.trim(): strips code units <= U+0020 from both ends."""
    b, e = 0, len(s)
    while b < e and s[b] <= " ":
        b += 1
    while e > b and s[e - 1] <= " ":
        e -= 1
    return s[b:e]


def utf16_key(s: str) -> bytes:
    """Sort key equal to java.lang.String.compareTo order (UTF-16 code units)."""
    return s.encode("utf-16-be", "surrogatepass")


def utf16_len(s: str) -> int:
    """java.lang.String.length(): number of UTF-16 code units."""
    return len(s) + sum(1 for ch in s if ord(ch) > 0xFFFF)


def drop_right_utf16(s: str, n: int) -> str:
    """scala StringOps.dropRight(n) on a Java string (n UTF-16 code units)."""
    units = s.encode("utf-16-be", "surrogatepass")
    keep = max(0, len(units) // 2 - n)
    return units[: keep * 2].decode("utf-16-be", "surrogatepass")


def split_lines(text: str) -> list[str]:
    """Hadoop LineRecordReader as used by sc.textFile: LF, CRLF and CR all end a
    line; no trailing empty record after a final terminator."""
    lines = re.split("\r\n|\n|\r", text)
    if lines and lines[-1] == "":
        lines.pop()
    return lines


def parse_instances(text: str) -> list[tuple[str, str, str]]:
    """Preprocessor.scala:119-122: skip '#' lines, split("\\s+", 3), object is
    the rest of the line minus its last two characters, trimmed. A line with
    fewer than three fields raises, as the reference's t(1)/t(2) would."""
    out = []
    for line in split_lines(text):
        if line.startswith("#"):
            continue
        t = _JAVA_WS.split(line, maxsplit=2)
        if len(t) < 3:
            raise ValueError("malformed N-Triples line: %r" % line[:80])
        out.append((java_trim(t[0]), java_trim(t[1]), java_trim(drop_right_utf16(t[2], 2))))
    return out


def is_literal(s: str) -> bool:  # Preprocessor.scala:114
    return s.startswith('"')


def create_instance_graph(triples):
    """Preprocessor.scala:131-167.

    Returns (vertices, edges): vertices is a list of (id, uri, types, labels)
    in id order -- id is the rank of the uri in String.compareTo order
    (sortByKey + zipWithIndex, :142-143); edges is the set of (srcId, dstId, p)
    for triples with p != rdf:type and a non-literal object (:153-167)."""
    attrs: dict[str, list[tuple[str, str]]] = defaultdict(list)
    for s, p, o in triples:
        if p == RDF_TYPE:
            attrs[s].append(("rdf:type", o))
        elif is_literal(o):
            attrs[s].append((p, o))
        else:
            attrs[s]  # noqa: B018  (s, Seq())
            attrs[o]  # noqa: B018  (o, Seq())
    vertices = []
    ids = {}
    for i, uri in enumerate(sorted(attrs, key=utf16_key)):
        labels = attrs[uri]
        types = [o for (k, o) in labels if k == "rdf:type"]
        rest = [(k, o) for (k, o) in labels if k != "rdf:type"]
        vertices.append((i, uri, types, rest))
        ids[uri] = i
    edges = set()
    for s, p, o in triples:
        if p != RDF_TYPE and not is_literal(o):
            edges.add((ids[s], ids[o], p))
    return vertices, edges


def load_class_index(text: str) -> dict[str, list[str]]:
    """Loader.scala:114-123: `class \\t p1,p2,...`; later lines override."""
    index = {}
    for line in text.splitlines():
        tokens = line.split("\t")
        index[tokens[0]] = "".join(tokens[1:]).split(",")
    return index


def _distinct(seq):
    seen, out = set(), []
    for x in seq:
        if x not in seen:
            seen.add(x)
            out.append(x)
    return out


def partition_instances(vertices, edges, class_index):
    """DAP.scala:567-622 (+ populateLabels :624-629).

    Returns the set of ((s, o, p), partition_string)."""
    vattr = {vid: (uri, types, labels) for (vid, uri, types, labels) in vertices}
    triplets = [(vattr[s], vattr[d], p) for (s, d, p) in edges]
    lookup = lambda c: class_index.get(c, [])  # noqa: E731  brIndex.value.getOrElse(_, Array())

    partitioned = set()  # :575-580
    for src, dst, p in triplets:
        if not (src[1] or dst[1]):
            continue
        t = (src[0], dst[0], p)
        types = _distinct(list(src[1]) + list(dst[1]))
        parts = _distinct([q for c in types for q in lookup(c)])
        for q in parts:
            partitioned.add((t, q))

    type_triples = set()  # :583-586
    for src, dst, p in triplets:
        if not (src[1] or dst[1]):
            continue
        for c in src[1]:
            type_triples.add((src[0], RDF_TYPE, c))
        for c in dst[1]:
            type_triples.add((dst[0], RDF_TYPE, c))
    partition_type = set()  # :588-593, stored column order (s, o, p)
    for (s, p, c) in type_triples:
        for q in lookup(c):
            partition_type.add(((s, c, p), q))

    label_triples = set()  # :595-602
    for src, dst, p in triplets:
        if not ((src[2] and src[1]) or (dst[2] and dst[1])):
            continue
        for attr in (src, dst):
            uri, types, labels = attr
            for (lp, lo) in labels:
                if not types:  # .filter(!_._2.isEmpty) on the per-type lookup array
                    continue
                parts = _distinct([q for c in types for q in lookup(c)])
                for q in parts:
                    label_triples.add(((uri, lo, lp), q))

    return partition_type | partitioned | label_triples  # :604


def uri_to_storage(uri: str) -> str:
    """DAP.scala:512-514."""
    return re.sub("[^A-Za-z0-9]", "_", uri).lower()


def partition_clusters(final_clusters, n_partitions: int):
    """DAP.scala:360-382 with CustomPartitioner (:541-551).

    Returns (parts, statistics): parts[i] is the set of (s, o, p) rows of
    first-level partition i; statistics is the set of `statistics.txt_N` lines."""
    parts = [set() for _ in range(n_partitions)]
    counts = defaultdict(int)
    for (t, q) in final_clusters:
        i = int(q)  # key.toString.toInt
        if not 0 <= i < n_partitions:
            raise ValueError("partition id %r outside [0, %d)" % (q, n_partitions))
        parts[i].add(t)
        counts[q] += 1
    stats = set()
    for q, c in counts.items():
        stats.add(("0" + q if len(q) == 1 else q) + " " + str(c))
    return parts, stats


def partition_name(i: int) -> str:
    return "%05d" % i  # part-0000i-<uuid>... -> substring(start, start+5), DAP.scala:427-428


def sub_partition_clusters(parts):
    """DAP.scala:385-509.

    Returns (node_index, vp, vp_index):
      node_index  set of (uri, type)                                   (:402-412)
      vp          {(partition_name, '_3=<storage pred>'): list of (s, o)}   (:419-437)
                  a list, because two predicates that sanitise to the same name
                  share a directory and may both contribute the same (s, o)
      vp_index    {partition_name: {uri: frozenset of 's<storage>'/'o<storage>'}}  (:450-501)
    """
    cluster = set()
    for rows in parts:
        cluster |= rows
    node_index = {(s, o) for (s, o, p) in cluster if p == RDF_TYPE}
    typed = {u for (u, _) in node_index}

    vp = defaultdict(list)
    vp_index = {}
    for i, rows in enumerate(parts):
        if not rows:
            continue  # Spark writes no part file for an empty partition, so no directory follows
        name = partition_name(i)
        for (s, o, p) in rows:
            if utf16_len(p) < 150:
                vp[(name, "_3=" + uri_to_storage(p))].append((s, o))
        idx = defaultdict(set)
        for (s, o, p) in rows:
            if s in typed:
                idx[s].add("s" + uri_to_storage(p))
            if o in typed:
                idx[o].add("o" + uri_to_storage(p))
            if p == RDF_TYPE:
                idx[o].add("o" + uri_to_storage(RDF_TYPE))
        vp_index[name] = {u: frozenset(v) for u, v in idx.items()}
    return node_index, dict(vp), vp_index


def compute_partition_statistics(parts, vp):
    """DAP.scala:101-114: one `<entry> <rows>` line per entry of clusters_N_bal/.

    Part-file names carry a run-specific uuid, so only the directory lines
    (`0000i <rows>`, the ones Loader.scala:236-247 keeps) are returned, as a
    dict name -> rows; part files have len(parts[i]) rows."""
    stat = defaultdict(int)
    for (name, _), rows in vp.items():
        stat[name] += len(rows)
    for i, rows in enumerate(parts):
        if rows:
            stat.setdefault(partition_name(i), 0)
    return dict(stat)


def compute_cc(vertices, edges):
    """Preprocessor.computeCC (Preprocessor.scala:230-264; combineTripleTypes :356-366).

    For every instance edge whose endpoints both carry types, one row per (subject type, object type)
    pair -- the type lists keep duplicates, so a type triple stated twice counts twice -- grouped by
    (subType, pred, objType). Returns {(subType, pred, objType): (rows, distinct object instances)};
    the reference saves these as "((subType,pred,objType),rows,dist)" text lines in no particular order."""
    vattr = {vid: (uri, types, labels) for (vid, uri, types, labels) in vertices}
    rows = defaultdict(int)
    inst2 = defaultdict(set)
    for (s, d, p) in edges:
        src, dst = vattr[s], vattr[d]
        if not (src[1] and dst[1]):
            continue
        for st in src[1]:
            for ot in dst[1]:
                if st and p and ot:  # :241
                    rows[(st, p, ot)] += 1
                    inst2[(st, p, ot)].add(dst[0])
    return {k: (rows[k], len(inst2[k])) for k in rows}


def compute_schema_node_freq(vertices, edges):
    """Preprocessor.computeSchemaNodeFreq (Preprocessor.scala:312-326): over the distinct typed endpoints
    of edges, the number of times each class occurs in their type lists. {class: freq}; the reference
    writes "class\tfreq" lines by descending freq (ties in no particular order)."""
    vattr = {vid: (uri, types, labels) for (vid, uri, types, labels) in vertices}
    nodes = set()
    for (s, d, p) in edges:
        if vattr[s][1] or vattr[d][1]:
            for v in (s, d):
                if vattr[v][1]:
                    nodes.add(v)  # .distinct on (uri, types, labels): one per vertex
    freq = defaultdict(int)
    for v in nodes:
        for c in vattr[v][1]:
            freq[c] += 1
    return dict(freq)


def compute_schema_node_count(vertices, edges):
    """Preprocessor.computeSchemaNodeCount (Preprocessor.scala:328-339): every edge with a typed endpoint
    counts once for each entry of its endpoints' type lists. {class: count}."""
    vattr = {vid: (uri, types, labels) for (vid, uri, types, labels) in vertices}
    count = defaultdict(int)
    for (s, d, p) in edges:
        if vattr[s][1] or vattr[d][1]:
            for c in list(vattr[s][1]) + list(vattr[d][1]):
                count[c] += 1
    return dict(count)


def run_dap(text: str, class_index_text: str, n_partitions: int):
    """The whole post-DAP stage on in-memory inputs; returns a dict of artefacts."""
    triples = parse_instances(text)
    vertices, edges = create_instance_graph(triples)
    class_index = load_class_index(class_index_text)
    final = partition_instances(vertices, edges, class_index)
    parts, stats = partition_clusters(final, n_partitions)
    node_index, vp, vp_index = sub_partition_clusters(parts)
    return {
        "n_triples": len(triples),
        "vertices": vertices,
        "edges": edges,
        "parts": parts,
        "statistics": stats,
        "node_index": node_index,
        "vp": vp,
        "vp_index": vp_index,
        "partition_stat": compute_partition_statistics(parts, vp),
        "schema_cc": compute_cc(vertices, edges),
        "schema_node_freq": compute_schema_node_freq(vertices, edges),
        "schema_node_count": compute_schema_node_count(vertices, edges),
    }
