"""Seeded generators of heterogeneous-schema N-Triples: SWDF-shaped (about a hundred predicates, many
small vertical partitions, multi-typed subjects, literal-heavy) and DBpedia-shaped (thousands of
predicates with Zipf frequencies, hub objects, long and look-alike predicate IRIs).

The reference names both datasets only by folder (`queries/swdf*`, `queries/dbpedia`); the data is not
in its tree, so these are synthetic inputs of the same shape, not copies of either dataset.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"


@dataclass(frozen=True)
class Shape:
    ns: str
    n_pred: int
    zipf_pred: float
    n_class: int
    literal_share: float     # share of predicates that are datatype properties
    type_share: float        # share of triples that are rdf:type
    triples_per_entity: float
    zipf_object: float       # skew of edge targets (hubs)
    untyped_share: float
    odd_line_share: float    # tabs, runs of blanks, CRLF


SHAPES = {
    "swdf": Shape("http://data.semanticweb.org/", 120, 1.1, 48, 0.55, 0.14, 9.0, 1.05, 0.08, 0.02),
    "dbpedia": Shape("http://dbpedia.org/", 10000, 1.0, 320, 0.5, 0.10, 14.0, 1.2, 0.15, 0.01),
}

_LITERALS = ['"%d"', '"%d"^^<http://www.w3.org/2001/XMLSchema#integer>', '"name %d"@en', '"Straße  %d été"@de',
             '"東京 %d"@ja', '"%d-01-01"^^<http://www.w3.org/2001/XMLSchema#date>', '"a \\"quoted\\" value %d"',
             '"tab\there %d"', '"\U0001F600 %d"']


def _zipf_weights(n: int, a: float) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), a)
    return w / w.sum()


def predicates(shape: str) -> list:
    sh = SHAPES[shape]
    out = []
    for i in range(sh.n_pred):
        if i % 97 == 41:      # 150 characters or more: kept in the partitions, left out of the VP directories
            out.append("<%sontology/%s%d>" % (sh.ns, "long" * 40, i))
        elif i % 89 == 7:     # differs from its neighbour only in characters the directory name folds
            out.append("<%sproperty/P-%d>" % (sh.ns, i - 1))
        elif i % 89 == 6:
            out.append("<%sproperty/p_%d>" % (sh.ns, i))
        else:
            out.append("<%sproperty/p%d>" % (sh.ns, i))
    return out


def generate(shape: str, n_triples: int, seed: int = 0, n_partitions: int = 8):
    """Returns (text, class_index_text, n_partitions). Deterministic in (shape, n_triples, seed, n_partitions)."""
    sh = SHAPES[shape]
    rng = np.random.default_rng(seed)
    n_ent = max(16, int(n_triples / sh.triples_per_entity))
    preds = predicates(shape)
    is_lit = rng.random(sh.n_pred) < sh.literal_share
    w_pred = _zipf_weights(sh.n_pred, sh.zipf_pred)
    w_cls = _zipf_weights(sh.n_class, 0.9)
    w_obj = _zipf_weights(n_ent, sh.zipf_object)

    n_type = int(n_triples * sh.type_share)
    n_rest = n_triples - n_type
    typed = rng.random(n_ent) >= sh.untyped_share
    typed_ids = np.flatnonzero(typed)
    t_subj = typed_ids[rng.integers(0, typed_ids.size, size=n_type)]
    t_cls = rng.choice(sh.n_class, size=n_type, p=w_cls)
    r_subj = np.minimum((n_ent * np.power(rng.random(n_rest), 1.6)).astype(np.int64), n_ent - 1)
    r_pred = rng.choice(sh.n_pred, size=n_rest, p=w_pred)
    r_obj = rng.choice(n_ent, size=n_rest, p=w_obj)
    r_val = rng.integers(0, 50, size=n_rest)
    r_tpl = rng.integers(0, len(_LITERALS), size=n_rest)

    ent = lambda i: "<%sresource/E%d>" % (sh.ns, i)  # noqa: E731
    lines = ["%s %s <%sontology/C%d> ." % (ent(s), RDF_TYPE, sh.ns, c) for s, c in zip(t_subj.tolist(), t_cls.tolist())]
    for s, p, o, v, t in zip(r_subj.tolist(), r_pred.tolist(), r_obj.tolist(), r_val.tolist(), r_tpl.tolist()):
        if is_lit[p]:
            lines.append("%s %s %s ." % (ent(s), preds[p], _LITERALS[t] % v))
        else:
            lines.append("%s %s %s ." % (ent(s), preds[p], ent(o)))
    order = rng.permutation(len(lines))
    odd = rng.random(len(lines)) < sh.odd_line_share
    out = []
    for k in order.tolist():
        ln = lines[k]
        if odd[k]:
            a, b, c = ln.split(" ", 2)
            ln = (a + "\t" + b + "   " + c) if k & 1 else (a + "  \t" + b + " " + c + "\r")
        out.append(ln)
    text = "\n".join(out) + "\n"

    ci = []
    for c in range(sh.n_class):
        if c % 7 == 5:
            continue  # classes the schema clustering never saw
        k = 1 + int(rng.random() < 0.3)
        ps = sorted(set(rng.integers(0, n_partitions, size=k).tolist()))
        ci.append("<%sontology/C%d>\t%s\n" % (sh.ns, c, ",".join(str(x) for x in ps)))
    return text, "".join(ci), n_partitions
