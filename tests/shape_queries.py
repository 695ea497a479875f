"""Turns the query shapes of tests/golden/query_shapes.json into SPARQL over a given data set: constants and
predicates are chosen so that the pattern has answers in the data (random search over the triples, seeded)."""
import random
from collections import defaultdict

RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"


class TripleIndex:
    def __init__(self, triples):
        self.triples = list(triples)
        self.by_s = defaultdict(list)
        self.by_o = defaultdict(list)
        self.by_p = defaultdict(list)
        for t in self.triples:
            self.by_s[t[0]].append(t)
            self.by_o[t[2]].append(t)
            self.by_p[t[1]].append(t)

    def candidates(self, s, p, o, rng, limit=40):
        """Triples matching the bound positions (None = free), a bounded random sample."""
        if s is not None:
            pool = self.by_s.get(s, [])
        elif o is not None:
            pool = self.by_o.get(o, [])
        elif p is not None:
            pool = self.by_p.get(p, [])
        else:
            pool = self.triples
        hits = [t for t in (pool if len(pool) <= 4000 else rng.sample(pool, 4000))
                if (s is None or t[0] == s) and (p is None or t[1] == p) and (o is None or t[2] == o)]
        rng.shuffle(hits)
        return hits[:limit]


def instantiate(shape, index: TripleIndex, rng: random.Random, tries=30):
    """One assignment of the shape's symbols to terms of the data under which every pattern is a triple of the
    data, or None. 'a' is rdf:type; P symbols are predicates other than rdf:type; L symbols are literals."""
    patterns = shape
    order = sorted(range(len(patterns)), key=lambda i: -sum(1 for x in patterns[i] if x[0] == "C"))

    def ok(sym, term):
        if sym == "a":
            return term == RDF_TYPE
        if sym[0] == "P":
            return term != RDF_TYPE
        if sym[0] == "L":
            return term.startswith('"')
        if sym[0] == "C":
            return not term.startswith('"')
        return True

    def search(k, bind):
        if k == len(order):
            return bind
        s, p, o = patterns[order[k]]
        get = lambda x: RDF_TYPE if x == "a" else bind.get(x)  # noqa: E731
        for t in index.candidates(get(s), get(p), get(o), rng):
            b2 = dict(bind)
            good = True
            for sym, term in zip((s, p, o), t):
                if sym == "a":
                    good &= term == RDF_TYPE
                elif sym in b2:
                    good &= b2[sym] == term
                elif not ok(sym, term):
                    good = False
                else:
                    b2[sym] = term
                if not good:
                    break
            if good:
                # distinct P symbols stand for distinct predicates
                ps = [v for key, v in b2.items() if key[0] == "P"]
                if len(ps) != len(set(ps)):
                    continue
                r = search(k + 1, b2)
                if r is not None:
                    return r
        return None

    for _ in range(tries):
        r = search(0, {})
        if r is not None:
            return r
    return None


def sparql(shape, bind) -> str:
    term = lambda x: RDF_TYPE if x == "a" else (x.replace("?v", "?x") if x.startswith("?") else bind[x])  # noqa: E731
    variables = sorted({x.replace("?v", "?x") for tp in shape for x in tp if x.startswith("?")})
    body = " . ".join(" ".join(term(x) for x in tp) for tp in shape)
    return "SELECT %s WHERE { %s }" % (" ".join(variables), body)
