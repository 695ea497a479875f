#!/usr/bin/env python3
"""Reduces the reference's DBpedia and SWDF query files to their shapes: every constant becomes C<i>, every
predicate P<j> (rdf:type stays), every variable ?v<k>, numbered in order of appearance. Only the structure is
kept: which positions are constants, which patterns share a variable, where the predicate is a variable.

  python tests/golden/make_query_shapes.py /root/reference/queries > tests/golden/query_shapes.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"


def patterns_of(text):
    """Triple patterns of a query file, split the way the reference's parser does but ignoring blank
    remainders (the DBpedia files end their last pattern with " . " before the brace, on which
    QueryParser.scala:63-72 itself indexes past the end of a one-element array)."""
    import re
    q = "".join(re.split("\r\n|\n|\r", text))
    body = "".join(m.group(1) for m in re.finditer(r"(\{.*?\})", q))[1:-1]
    out = []
    for part in re.split(r" \.", body):
        toks = part.split()
        if len(toks) >= 3 and not part.startswith("#"):
            out.append((toks[0], RDF_TYPE if toks[1] in ("a", "rdf:type") else toks[1], " ".join(toks[2:])))
    return out


def shape_of(text):
    names = {}

    def sym(tok, kind):
        if tok == RDF_TYPE:
            return "a"
        if "?" in tok:
            kind = "?v"
        elif '"' in tok:
            kind = "L"
        key = (kind, tok)
        if key not in names:
            names[key] = "%s%d" % (kind, sum(1 for k in names if k[0] == kind))
        return names[key]

    return [[sym(s, "C"), sym(p, "P"), sym(o, "C")] for (s, p, o) in patterns_of(text)]


def main(qroot):
    out = {}
    for folder in ("dbpedia", "swdf/swdf(b)", "swdf/swdf(u)"):
        counts = {}
        d = os.path.join(qroot, folder)
        for f in sorted(os.listdir(d)):
            try:
                sh = shape_of(open(os.path.join(d, f)).read())
            except Exception:  # noqa: BLE001  (a file the reference's parser rejects as well)
                continue
            key = json.dumps(sh)
            counts[key] = counts.get(key, 0) + 1
        out[folder] = [{"patterns": json.loads(k), "files": v} for k, v in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/queries")
