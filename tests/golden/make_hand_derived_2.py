"""Writes tests/golden/hand_derived_2.json: a second known-answer example for the partitioner's post-DAP stage.

Every value below was worked out by hand from the reference's rules (Preprocessor.scala:131-167 vertices and edges,
230-264 schema_cc, 312-339 node frequency / count; DAP.scala:575-604 placement, 360-382 part files, 402-437 node index
and vertical partitions, 479-499 VP index); nothing here calls the oracle or the engine. The script only spells the two
long predicate names out and serialises.

Covers: a vertex with two types whose classes lie in different partitions (<m1>: <A> in 0, <B> in 1); a class replicated
to two partitions whose instance carries a label (<C> in 0 and 2, <c1> "cee"); predicates of exactly 149 and 150 UTF-16
units (`p.size < 150` keeps the first in the VP directories and drops the second; both stay in the part files and both
reach the VP index, which is built from the part files); two predicates whose names sanitise to one directory
(<http://e/Has-Part>, <http://e/has_part>); a typed object that is itself a class (<C> is an instance of <Meta> and the
object of <http://e/about>: it enters the VP index through the join on the object, DAP.scala:483-485, with
"o_http___e_about_", and through the rule for classes, :488-489, with "o<type>"; <A>, <B>, <Meta> enter through the second
rule alone); an isolated typed vertex (<z1>), whose triples vanish; a duplicate input line; a class-pair counted twice with
one distinct object (schema_cc ((<A>,<knows>,<C>),2,1)).
"""
import json
import os

T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
TS = "_http___www_w3_org_1999_02_22_rdf_syntax_ns_type_"
L149 = "<http://e/" + "p" * 138 + ">"   # 149 units: kept in the VP directories
L150 = "<http://e/" + "q" * 139 + ">"   # 150 units: not in the VP directories
S149 = "_http___e_" + "p" * 138 + "_"
S150 = "_http___e_" + "q" * 139 + "_"
assert len(L149) == 149 and len(L150) == 150
K, N, HP1, HP2, AB = "<knows>", "<name>", "<http://e/Has-Part>", "<http://e/has_part>", "<http://e/about>"
HPS, ABS = "_http___e_has_part_", "_http___e_about_"

lines = [
    ("<m1>", T, "<A>"), ("<m1>", T, "<B>"), ("<c1>", T, "<C>"), ("<c1>", N, '"cee"'), ("<m1>", K, "<c1>"),
    ("<m1>", L150, "<c1>"), ("<m1>", L149, "<c1>"), ("<m1>", HP1, "<c1>"), ("<c1>", HP2, "<m1>"), ("<c1>", AB, "<C>"),
    ("<C>", T, "<Meta>"), ("<m1>", N, '"em"'), ("<z1>", T, "<A>"), ("<m1>", K, "<c1>"), ("<m2>", T, "<A>"), ("<m2>", K, "<c1>"),
]
five = [("<m1>", "<c1>", K), ("<m1>", "<c1>", L150), ("<m1>", "<c1>", L149), ("<m1>", "<c1>", HP1), ("<c1>", "<m1>", HP2)]  # go to {0, 1, 2}
g = {
    "_comment": __doc__.strip(),
    "n_partitions": 3,
    "class_index": "<A>\t0\n<B>\t1\n<C>\t0,2\n<Meta>\t2\n",
    "instances": "".join("%s %s %s .\n" % t for t in lines),
    "n_triples": 16,
    # ids = rank in String.compareTo order: "<C>" < "<c1>" < "<m1>" < "<m2>" < "<z1>"; the classes <A>, <B>, <Meta> are no vertices
    "vertices": [
        [0, "<C>", ["<Meta>"], []],
        [1, "<c1>", ["<C>"], [[N, '"cee"']]],
        [2, "<m1>", ["<A>", "<B>"], [[N, '"em"']]],
        [3, "<m2>", ["<A>"], []],
        [4, "<z1>", ["<A>"], []],
    ],
    "edges": [[2, 1, K], [2, 1, L150], [2, 1, L149], [2, 1, HP1], [1, 2, HP2], [1, 0, AB], [3, 1, K]],
    # stored as (s, o, p)
    "parts": [
        five + [("<c1>", "<C>", AB), ("<m2>", "<c1>", K), ("<m1>", "<A>", T), ("<c1>", "<C>", T), ("<m2>", "<A>", T), ("<c1>", '"cee"', N), ("<m1>", '"em"', N)],
        five + [("<m1>", "<B>", T), ("<m1>", '"em"', N)],
        five + [("<c1>", "<C>", AB), ("<m2>", "<c1>", K), ("<c1>", "<C>", T), ("<C>", "<Meta>", T), ("<c1>", '"cee"', N)],
    ],
    "statistics": ["00 12", "01 7", "02 10"],
    "node_index": [["<m1>", "<A>"], ["<m1>", "<B>"], ["<c1>", "<C>"], ["<C>", "<Meta>"], ["<m2>", "<A>"]],
    "vp": {
        "00000/_3=_knows_": [["<m1>", "<c1>"], ["<m2>", "<c1>"]],
        "00000/_3=" + S149: [["<m1>", "<c1>"]],
        "00000/_3=" + HPS: [["<m1>", "<c1>"], ["<c1>", "<m1>"]],
        "00000/_3=" + ABS: [["<c1>", "<C>"]],
        "00000/_3=" + TS: [["<m1>", "<A>"], ["<c1>", "<C>"], ["<m2>", "<A>"]],
        "00000/_3=_name_": [["<c1>", '"cee"'], ["<m1>", '"em"']],
        "00001/_3=_knows_": [["<m1>", "<c1>"]],
        "00001/_3=" + S149: [["<m1>", "<c1>"]],
        "00001/_3=" + HPS: [["<m1>", "<c1>"], ["<c1>", "<m1>"]],
        "00001/_3=" + TS: [["<m1>", "<B>"]],
        "00001/_3=_name_": [["<m1>", '"em"']],
        "00002/_3=_knows_": [["<m1>", "<c1>"], ["<m2>", "<c1>"]],
        "00002/_3=" + S149: [["<m1>", "<c1>"]],
        "00002/_3=" + HPS: [["<m1>", "<c1>"], ["<c1>", "<m1>"]],
        "00002/_3=" + ABS: [["<c1>", "<C>"]],
        "00002/_3=" + TS: [["<c1>", "<C>"], ["<C>", "<Meta>"]],
        "00002/_3=_name_": [["<c1>", '"cee"']],
    },
    "partition_stat": {"00000": 11, "00001": 6, "00002": 9},   # rows of the directories: the 150-unit predicate's row is missing
    "vp_index": {
        "00000": {
            "<m1>": ["s_knows_", "s" + S150, "s" + S149, "s" + HPS, "o" + HPS, "s" + TS, "s_name_"],
            "<c1>": ["o_knows_", "o" + S150, "o" + S149, "o" + HPS, "s" + HPS, "s" + ABS, "s" + TS, "s_name_"],
            "<m2>": ["s_knows_", "s" + TS],
            "<C>": ["o" + ABS, "o" + TS],
            "<A>": ["o" + TS],
        },
        "00001": {
            "<m1>": ["s_knows_", "s" + S150, "s" + S149, "s" + HPS, "o" + HPS, "s" + TS, "s_name_"],
            "<c1>": ["o_knows_", "o" + S150, "o" + S149, "o" + HPS, "s" + HPS],
            "<B>": ["o" + TS],
        },
        "00002": {
            "<m1>": ["s_knows_", "s" + S150, "s" + S149, "s" + HPS, "o" + HPS],
            "<c1>": ["o_knows_", "o" + S150, "o" + S149, "o" + HPS, "s" + HPS, "s" + ABS, "s" + TS, "s_name_"],
            "<m2>": ["s_knows_"],
            "<C>": ["o" + ABS, "o" + TS, "s" + TS],
            "<Meta>": ["o" + TS],
        },
    },
    # ((subType, pred, objType), rows, distinct objects): edges with both ends typed, every pair of their types
    "schema_cc": [
        [["<A>", K, "<C>"], 2, 1], [["<B>", K, "<C>"], 1, 1], [["<A>", L150, "<C>"], 1, 1], [["<B>", L150, "<C>"], 1, 1],
        [["<A>", L149, "<C>"], 1, 1], [["<B>", L149, "<C>"], 1, 1], [["<A>", HP1, "<C>"], 1, 1], [["<B>", HP1, "<C>"], 1, 1],
        [["<C>", HP2, "<A>"], 1, 1], [["<C>", HP2, "<B>"], 1, 1], [["<C>", AB, "<Meta>"], 1, 1],
    ],
    # distinct typed ends of edges per class: <m1> (A, B), <m2> (A), <c1> (C), <C> (Meta)
    "schema_node_freq": {"<A>": 2, "<B>": 1, "<C>": 1, "<Meta>": 1},
    # per edge with a typed end, the types of both ends: five edges <m1>-<c1> (A, B, C each), <c1>-<C> (C, Meta), <m2>-<c1> (A, C)
    "schema_node_count": {"<A>": 6, "<B>": 5, "<C>": 7, "<Meta>": 1},
}
g["parts"] = [[list(t) for t in p] for p in g["parts"]]
assert [len(p) for p in g["parts"]] == [12, 7, 10]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hand_derived_2.json"), "w") as f:
    json.dump(g, f, indent=1)
