"""Writes tests/golden/hand_derived_queries.json: known answers for the query path over the vertical-partition tables of
tests/golden/hand_derived_2.json (three first-level partitions; the tables are listed in that file under "vp").

Every statement below is written by hand in the form the reference's translator emits (QueryTranslator.scala:363-392,
417-543; one example of the reference's own is tests/golden/translated_Q2_reference_example.txt), and every answer was
worked out by hand from the tables with the semantics of QueryProcessor.executeFinalQuery (QueryProcessor.scala:184-208):
inner joins on equal strings, bags (a binding produced twice is counted twice), UNION as a set, several statements of one
file folded left to right on the columns they share. Nothing here calls the oracle or the engine; the script only
spells the table names out and serialises. `load_data` is the reference's Load_data: the rows of the tables a file names
(QueryProcessor.scala:222-232), each table once per mention in the `partitions` line.

The tables used (subject, object):
  00000 knows     (<m1>,<c1>) (<m2>,<c1>)            00001 knows  (<m1>,<c1>)
  00000 has_part  (<m1>,<c1>) (<c1>,<m1>)            00002 type   (<c1>,<C>) (<C>,<Meta>)
  00000 type      (<m1>,<A>) (<c1>,<C>) (<m2>,<A>)
  00000 name      (<c1>,"cee") (<m1>,"em")
"""
import json
import os

TYPE = "_http___www_w3_org_1999_02_22_rdf_syntax_ns_type_"


def key(part, sub):           # an entry of the `partitions` line
    return "%s-_3=%s" % (part, sub)


def table(part, sub):         # QueryProcessor.scala:216
    return "table" + key(part, sub).replace("-", "_").replace("=", "_E_")


K0, K1 = ("00000", "_knows_"), ("00001", "_knows_")
H0, T0, N0, T2 = ("00000", "_http___e_has_part_"), ("00000", TYPE), ("00000", "_name_"), ("00002", TYPE)


def file_text(name, partitions, statements):
    return ">>>>> %s\n%s\npartitions %s\nTP %d" % (
        name, "\n".join("%s %s" % s for s in statements), ",".join(key(*p) for p in partitions), len(partitions))


cases = []

# 1. ?x <knows> ?y . ?x rdf:type <A>  -- both knows rows have a subject of type <A>
cases.append({
    "name": "two_patterns_constant_object",
    "text": file_text("two_patterns_constant_object", [K0, T0], [("X",
        "SELECT tab0.X AS X,tab0.Y AS Y FROM (SELECT s AS X, o AS Y FROM %s) AS tab0, (SELECT s AS X FROM %s WHERE o == '<A>') AS tab1 "
        "WHERE tab0.X=tab1.X " % (table(*K0), table(*T0)))]),
    "columns": ["X", "Y"], "rows": [["<m1>", "<c1>"], ["<m2>", "<c1>"]], "count": 2, "load_data": 2 + 3})

# 2. bags: the objects of knows are <c1> twice, <c1> is a <C>: the binding Y=<c1> comes out twice and both count
cases.append({
    "name": "a_binding_produced_twice_counts_twice",
    "text": file_text("a_binding_produced_twice_counts_twice", [K0, T0], [("X",
        "SELECT tab0.Y AS Y FROM (SELECT o AS Y FROM %s) AS tab0, (SELECT s AS Y FROM %s WHERE o == '<C>') AS tab1 WHERE tab0.Y=tab1.Y "
        % (table(*K0), table(*T0)))]),
    "columns": ["Y"], "rows": [["<c1>"], ["<c1>"]], "count": 2, "load_data": 2 + 3})

# 3. UNION is a set: (<m1>,<c1>) is in both partitions' knows tables and comes out once
cases.append({
    "name": "union_of_two_partitions_is_a_set",
    "text": file_text("union_of_two_partitions_is_a_set", [K0, K1], [("X",
        "SELECT tab0.X AS X,tab0.Y AS Y FROM ( (SELECT s AS X, o AS Y FROM %s)UNION(SELECT s AS X, o AS Y FROM %s) ) AS tab0 "
        % (table(*K0), table(*K1)))]),
    "columns": ["X", "Y"], "rows": [["<m1>", "<c1>"], ["<m2>", "<c1>"]], "count": 2, "load_data": 2 + 1})

# 4. constants on both sides of a two-pattern chain: <m1> has_part ?y . ?y has_part <m1>  ->  ?y = <c1>
cases.append({
    "name": "constant_subject_and_constant_object",
    "text": file_text("constant_subject_and_constant_object", [H0, H0], [("X",
        "SELECT tab0.Y AS Y FROM (SELECT o AS Y FROM %s WHERE s == '<m1>') AS tab0, (SELECT s AS Y FROM %s WHERE o == '<m1>') AS tab1 "
        "WHERE tab0.Y=tab1.Y " % (table(*H0), table(*H0)))]),
    "columns": ["Y"], "rows": [["<c1>"]], "count": 1, "load_data": 2 + 2})

# 5. a constant the data does not hold: nothing
cases.append({
    "name": "unknown_constant",
    "text": file_text("unknown_constant", [K0, T0], [("X",
        "SELECT tab0.X AS X FROM (SELECT s AS X FROM %s) AS tab0, (SELECT s AS X FROM %s WHERE o == '<Nope>') AS tab1 WHERE tab0.X=tab1.X "
        % (table(*K0), table(*T0)))]),
    "columns": ["X"], "rows": [], "count": 0, "load_data": 2 + 3})

# 6. tables of two first-level partitions: ?x <knows> ?y (partition 1) . ?y rdf:type <C> (partition 2)
cases.append({
    "name": "join_across_two_partitions",
    "text": file_text("join_across_two_partitions", [K1, T2], [("X",
        "SELECT tab0.X AS X,tab0.Y AS Y FROM (SELECT s AS X, o AS Y FROM %s) AS tab0, (SELECT s AS Y FROM %s WHERE o == '<C>') AS tab1 "
        "WHERE tab0.Y=tab1.Y " % (table(*K1), table(*T2)))]),
    "columns": ["X", "Y"], "rows": [["<m1>", "<c1>"]], "count": 1, "load_data": 1 + 2})

# 7. a three-pattern star with a literal: ?x <knows> ?y . ?x <name> ?n . ?x rdf:type <A>: only <m1> has a name
cases.append({
    "name": "star_with_a_literal",
    "text": file_text("star_with_a_literal", [K0, N0, T0], [("X",
        "SELECT tab0.X AS X,tab0.Y AS Y,tab1.N AS N FROM (SELECT s AS X, o AS Y FROM %s) AS tab0, (SELECT s AS X, o AS N FROM %s) AS tab1, "
        "(SELECT s AS X FROM %s WHERE o == '<A>') AS tab2 WHERE tab0.X=tab1.X AND tab1.X=tab2.X " % (table(*K0), table(*N0), table(*T0)))]),
    "columns": ["N", "X", "Y"], "rows": [['"em"', "<m1>", "<c1>"]], "count": 1, "load_data": 2 + 2 + 3})

# 8. two statements in one file (customJoin on the shared column X): knows(X,Y) with name(X,N) -> (<m1>,<c1>,"em")
cases.append({
    "name": "two_statements_joined_on_their_common_column",
    "text": file_text("two_statements_joined_on_their_common_column", [K0, N0], [
        ("A", "SELECT tab0.X AS X,tab0.Y AS Y FROM (SELECT s AS X, o AS Y FROM %s) AS tab0 " % table(*K0)),
        ("B", "SELECT tab0.X AS X,tab0.N AS N FROM (SELECT s AS X, o AS N FROM %s) AS tab0 " % table(*N0))]),
    "columns": ["N", "X", "Y"], "rows": [['"em"', "<m1>", "<c1>"]], "count": 1, "load_data": 2 + 2})

# 9. two statements without a common column: the cross product, 2 x 2 rows
cases.append({
    "name": "two_statements_without_a_common_column",
    "text": file_text("two_statements_without_a_common_column", [K0, N0], [
        ("A", "SELECT tab0.X AS X FROM (SELECT s AS X FROM %s) AS tab0 " % table(*K0)),
        ("B", "SELECT tab0.N AS N FROM (SELECT o AS N FROM %s) AS tab0 " % table(*N0))]),
    "columns": ["N", "X"], "rows": [['"cee"', "<m1>"], ['"cee"', "<m2>"], ['"em"', "<m1>"], ['"em"', "<m2>"]], "count": 4,
    "load_data": 2 + 2})

out = {"_comment": __doc__.strip(), "layout": "hand_derived_2.json", "cases": cases}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hand_derived_queries.json")
json.dump(out, open(path, "w"), indent=1, ensure_ascii=False)
print("wrote", path, len(cases), "cases")
