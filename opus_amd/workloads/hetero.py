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


def _fill(template: bytes, fields, n: int) -> np.ndarray:
    """n copies of `template` (one row each) with zero-padded decimal fields (column, width, values) filled in."""
    buf = np.tile(np.frombuffer(template, dtype=np.uint8), (n, 1))
    for col, width, values in fields:
        v = np.asarray(values, dtype=np.int64)
        for k in range(width):
            buf[:, col + width - 1 - k] = 48 + (v // 10 ** k) % 10
    return buf


def generate_large(shape: str, n_triples: int, seed: int = 0, n_partitions: int = 8, sample_lines: int = 150_000):
    """The same shapes at benchmark size, built with numpy: returns (text as a uint8 array, class index text,
    n_partitions, the first `sample_lines` lines as a str). Predicate frequencies, the share of rdf:type triples, literal
    and object properties, hub objects and untyped subjects follow SHAPES; names have a fixed width (the irregular names of
    `predicates()` -- 150 characters and more, names that share a directory -- are left to the small inputs of the tests).
    A subject's triples follow one another in the text, as in a dump grouped by subject."""
    sh = SHAPES[shape]
    rng = np.random.default_rng(seed)
    n_ent = max(16, int(n_triples / sh.triples_per_entity))
    is_lit = rng.random(sh.n_pred) < sh.literal_share
    w_pred = _zipf_weights(sh.n_pred, sh.zipf_pred)
    w_cls = _zipf_weights(sh.n_class, 0.9)
    n_type = int(n_triples * sh.type_share)
    n_rest = n_triples - n_type
    typed_ids = np.flatnonzero(rng.random(n_ent) >= sh.untyped_share)
    subj = np.concatenate([typed_ids[rng.integers(0, typed_ids.size, size=n_type)],
                           np.minimum((n_ent * np.power(rng.random(n_rest), 1.6)).astype(np.int64), n_ent - 1)])
    pred = np.concatenate([np.full(n_type, -1, dtype=np.int64), rng.choice(sh.n_pred, size=n_rest, p=w_pred)])
    # hub objects: a power law over the entities (inverse-CDF sampling; rng.choice with ten million weights is slow)
    obj = np.minimum((n_ent * np.power(rng.random(n_triples), 1.0 / max(1e-6, 2.0 - sh.zipf_object) * 2.5)).astype(np.int64), n_ent - 1)
    cls = rng.choice(sh.n_class, size=n_triples, p=w_cls)
    val = rng.integers(0, 100, size=n_triples)
    kind = np.where(pred < 0, 0, np.where(is_lit[np.maximum(pred, 0)], 2, 1))  # 0 type, 1 object property, 2 literal
    order = np.argsort(subj, kind="stable")
    subj, pred, obj, cls, val, kind = subj[order], pred[order], obj[order], cls[order], val[order], kind[order]
    ns = sh.ns.encode()
    t_type = b"<" + ns + b"resource/E00000000> " + RDF_TYPE.encode() + b" <" + ns + b"ontology/C000> .\n"
    t_edge = b"<" + ns + b"resource/E00000000> <" + ns + b"property/p00000> <" + ns + b"resource/E00000000> .\n"
    t_lit = b"<" + ns + b"resource/E00000000> <" + ns + b"property/p00000> \"name 00\"@en .\n"
    e_col = 1 + len(ns) + len(b"resource/E")
    lens = np.array([len(t_type), len(t_edge), len(t_lit)], dtype=np.int64)[kind]
    off = np.concatenate([[0], np.cumsum(lens)])
    text = np.empty(int(off[-1]), dtype=np.uint8)
    step = 400_000
    for a in range(0, n_triples, step):
        b = min(n_triples, a + step)
        k = kind[a:b]
        for which, tpl in ((0, t_type), (1, t_edge), (2, t_lit)):
            sel = np.flatnonzero(k == which) + a
            if sel.size == 0:
                continue
            if which == 0:
                rows = _fill(tpl, [(e_col, 8, subj[sel]), (tpl.index(b"ontology/C") + 10, 3, cls[sel])], sel.size)
            elif which == 1:
                rows = _fill(tpl, [(e_col, 8, subj[sel]), (tpl.index(b"property/p") + 10, 5, pred[sel]),
                                   (tpl.rindex(b"resource/E") + 10, 8, obj[sel])], sel.size)
            else:
                rows = _fill(tpl, [(e_col, 8, subj[sel]), (tpl.index(b"property/p") + 10, 5, pred[sel]), (tpl.index(b"\"name ") + 6, 2, val[sel])], sel.size)
            idx = off[sel][:, None] + np.arange(len(tpl))[None, :]
            text[idx] = rows
    ci = []
    for c in range(sh.n_class):
        if c % 7 == 5:
            continue  # classes the schema clustering never saw
        k = 1 + int(rng.random() < 0.3)
        ps = sorted(set(rng.integers(0, n_partitions, size=k).tolist()))
        ci.append("<%sontology/C%03d>\t%s\n" % (sh.ns, c, ",".join(str(x) for x in ps)))
    sample = text[:int(off[min(sample_lines, n_triples)])].tobytes().decode()
    return text, "".join(ci), n_partitions, sample
