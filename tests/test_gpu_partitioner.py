"""Parity of the HIP partitioner with the CPU restatement (bit-exact, string level)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def engine():
    from opus_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _run(engine, text, ci, n):
    from oracle import dap_oracle
    from parity_util import assert_same_as_oracle, engine_artifacts
    st = engine.partition_text(text.encode("utf-8", "surrogatepass") if isinstance(text, str) else text, ci, n)
    ref = dap_oracle.run_dap(text if isinstance(text, str) else bytes(text).decode(), ci, n)
    got = engine_artifacts(engine)
    assert st.n_triples == ref["n_triples"]
    assert st.n_vertices == len(ref["vertices"])
    assert st.n_edges == len(ref["edges"])
    assert st.n_rows == sum(len(p) for p in ref["parts"])
    assert_same_as_oracle(got, ref)
    return st, got, ref


@pytest.mark.parametrize("name", ["appendix_c.json", "hand_derived_2.json"])
def test_golden_known_answer(engine, name):
    """The hand-derived examples (worked out from the reference's rules, not by running anything) against the GPU
    itself, not only through the oracle."""
    g = json.load(open(os.path.join(HERE, "golden", name)))
    st, got, ref = _run(engine, g["instances"], g["class_index"], g["n_partitions"])
    assert [set(p) for p in got["parts"]] == [{tuple(t) for t in p} for p in g["parts"]]
    assert all(len(p) == len(set(p)) for p in got["parts"])
    assert set(got["node_index"]) == {tuple(x) for x in g["node_index"]} and len(got["node_index"]) == len(g["node_index"])
    assert {k[0] + "/" + k[1]: sorted(v) for k, v in got["vp"].items()} == {k: sorted(tuple(x) for x in v) for k, v in g["vp"].items()}
    assert got["vertices"] == [v[1] for v in g["vertices"]]
    assert got["edges"] == {tuple(e) for e in g["edges"]}
    assert got["statistics"] == set(g["statistics"]) and got["partition_stat"] == g["partition_stat"]
    assert {p: {u: set(s) for u, s in d.items()} for p, d in got["vp_index"].items()} == {p: {u: set(s) for u, s in d.items()} for p, d in g["vp_index"].items()}
    if "schema_cc" in g:
        assert got["schema_cc"] == {tuple(k): (rows, dist) for k, rows, dist in g["schema_cc"]}
        assert got["schema_node_freq"] == g["schema_node_freq"] and got["schema_node_count"] == g["schema_node_count"]


def test_lubm1(engine, lubm1):
    from parity_util import assert_same_as_oracle, engine_artifacts
    st = engine.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
    assert st.n_triples == lubm1["n_triples"]
    assert_same_as_oracle(engine_artifacts(engine), lubm1["oracle"])


def test_both_forms_of_the_tokeniser(lubm1, monkeypatch):
    """The tokeniser that finds the record starts itself (one pass over the text) and the two-pass form (line scan
    first) give the oracle's result; an input with more records in a tile than the one-pass form has slots falls back
    by itself, stays with the two-pass form for its next calls and returns to the one-pass form afterwards."""
    from opus_amd.engine import Engine
    from oracle import dap_oracle
    from parity_util import assert_same_as_oracle, engine_artifacts
    e = Engine(0)
    try:
        def form(st):
            return "two-pass" if st.ms_stage[0] > 0 else "one-pass"
        st = e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
        assert form(st) == "one-pass"
        assert_same_as_oracle(engine_artifacts(e), lubm1["oracle"])
        monkeypatch.setenv("DX_TWO_PASS", "1")
        st = e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
        assert form(st) == "two-pass" and st.n_triples == lubm1["n_triples"]
        assert_same_as_oracle(engine_artifacts(e), lubm1["oracle"])
        monkeypatch.delenv("DX_TWO_PASS")
        # 600 records of 13 bytes in the first tile, then ordinary lines
        text = "".join("<s%03d> <p> <o> .\n" % (i % 700) for i in range(3000)) + lubm1["text"]
        ref = dap_oracle.run_dap(text, lubm1["class_index"], 4)
        st = e.partition_text(text.encode(), lubm1["class_index"], 4)
        assert form(st) == "two-pass" and st.n_triples == ref["n_triples"]
        assert_same_as_oracle(engine_artifacts(e), ref)
        for _ in range(8):
            st = e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
            assert form(st) == "two-pass"
        st = e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
        assert form(st) == "one-pass"
        assert_same_as_oracle(engine_artifacts(e), lubm1["oracle"])
    finally:
        e.close()


EDGE_CASES = {
    "duplicates_and_comments": (
        "# a comment line\n<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n<a> <p> <b> .\n<a> <p> <b> .\n"
        "<a> <p> <b> .\n#another\n<b> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <B> .\n<a> <l> \"x\" .\n<a> <l> \"x\" .\n",
        "<A>\t0\n<B>\t1,2\n", 3),
    "crlf_tabs_no_final_newline": (
        "<a>\t<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>   <A> .\r\n<a>  <p>\t<b> .\r\n<b> <q> \"lit with  spaces\"@en .\r\n"
        "<b> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .",
        "<A>\t0,1\n", 2),
    "class_missing_from_index_and_untyped": (
        "<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <Z> .\n<a> <p> <b> .\n<b> <p> <c> .\n<c> <n> \"v\" .\n"
        "<c> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n<d> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n"
        "<d> <n> \"isolated\" .\n",
        "<A>\t1\n", 2),
    "multi_typed_and_literal_datatypes": (
        "<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <B> .\n"
        "<a> <p> <b> .\n<a> <n> \"1\"^^<http://www.w3.org/2001/XMLSchema#integer> .\n<b> <n> \"été \U0001F600\" .\n"
        "<b> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <B> .\n<ü> <p> <a> .\n<￮> <p> <a> .\n<\U00010400> <p> <a> .\n",
        "<A>\t0\n<B>\t2,3\n", 4),
    "long_predicate_left_out_of_vp": (
        "<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n<a> <" + "p" * 148 + "> <b> .\n<a> <" + "q" * 147 + "> <b> .\n",
        "<A>\t0\n", 1),
    "predicates_sharing_a_directory": (
        "<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n<a> <http://x/P-1> <b> .\n<a> <http://x/p_1> <b> .\n"
        "<a> <http://x/p.1> <c> .\n",
        "<A>\t0\n", 1),
    "empty_object_and_leading_space": (
        "<a> <p> \n <s> <o> .\n<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n", "<A>\t0\n", 1),
    # far more records per 32 KiB of text than an ordinary tile holds
    "dense_short_lines": ("".join("s%d p o%d .\n" % (i % 700, i % 13) for i in range(12000)) + "s1 a A .\n", "A\t0\n", 1),
    # three new terms per record: more terms than the dictionary table's first size has slots, so it regrows
    "all_terms_distinct": ("".join("<s%d> <p%d> <o%d> .\n" % (i, i, i) for i in range(3000)) +
                           "<s1> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n", "<A>\t0,1\n", 2),
    # type lists keep duplicates (a type stated twice counts twice in the measures); classes named "" are dropped from schema_cc
    "repeated_types_and_empty_class": (
        "<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n"
        "<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <B> .\n<b> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <B> .\n"
        "<b> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> \n<a> <p> <b> .\n<a> <p> <c> .\n<c> <q> <b> .\n<a> <p> <b> .\n<b> <q> <b> .\n",
        "<A>\t0\n<B>\t1\n", 2),
    # more classes than one pass of the measures' per-block histogram holds (8192): the class range is walked in windows
    "more_classes_than_a_histogram_pass": (
        "".join("<s%d> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <C%d> .\n<s%d> <p%d> <s%d> .\n" % (i, i % 9001, i, i % 3, (i * 7 + 1) % 9500)
                for i in range(9500)),
        "<C1>\t0\n<C9000>\t1\n<C17>\t0,1\n", 2),
    "empty_input": ("", "<A>\t0\n", 2),
    "only_comments": ("# nothing\n# here\n", "<A>\t0\n", 2),
}


@pytest.mark.parametrize("name", sorted(EDGE_CASES))
def test_edge_cases(engine, name):
    text, ci, n = EDGE_CASES[name]
    _run(engine, text, ci, n)


@pytest.mark.parametrize("text", ["<a> <p>\n", "<a>\n", "<a> <p> <b> .\n\n<a> <p> <c> .\n", "<a>   \n"])
def test_malformed_lines_are_errors(engine, text):
    """The reference throws on lines with fewer than three fields (Preprocessor.scala:121-122)."""
    from opus_amd.engine import DxError
    from oracle import dap_oracle
    with pytest.raises((ValueError, IndexError)):
        dap_oracle.run_dap(text, "<A>\t0\n", 1)
    with pytest.raises(DxError) as ei:
        engine.partition_text(text.encode(), "<A>\t0\n", 1)
    assert ei.value.code == -74


def test_directory_name_of_a_predicate_with_a_supplementary_character(engine):
    """uriToStorage is replaceAll("[^A-Za-z0-9]", "_") (DAP.scala:512-514): java.util.regex matches by code point,
    so a character outside the BMP becomes one '_', not one per UTF-16 unit. The table must be loadable under the
    key the translator derives from the predicate's text."""
    from oracle import dap_oracle
    pred = "<http://x/p\U00010400q\u00e9>"
    text = "<a> <http://www.w3.org/1999/02/22-rdf-syntax-ns#type> <A> .\n<a> %s <b> .\n<b> %s <c> .\n" % (pred, pred)
    _, got, ref = _run(engine, text, "<A>\t0\n", 1)
    name = dap_oracle.uri_to_storage(pred)
    assert name == "_http___x_p_q__"
    key = "00000-_3=" + name
    assert ("00000", "_3=" + name) in ref["vp"]
    engine.unload_all()
    assert engine.load_table(key) == 1   # (a, b): b has no type, so (b, c) is placed nowhere
    view = "table" + key.replace("-", "_").replace("=", "_E_")
    assert engine.query("SELECT tab0.X AS X FROM (SELECT s AS X FROM %s) AS tab0 " % view).count == 1


def test_class_index_partition_out_of_range(engine):
    from opus_amd.engine import DxError
    with pytest.raises(DxError) as ei:
        engine.partition_text(b"<a> <p> <b> .\n", "<A>\t5\n", 2)
    assert ei.value.code == -34


def test_random_small_graphs(engine):
    """Seeded random multi-typed graphs with heavy duplication and many tiny predicates."""
    rng = np.random.default_rng(7)
    T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    for trial in range(4):
        nv, ncls, npred, nparts = 60, 6, 9, int(rng.integers(1, 7))
        lines = []
        for _ in range(500):
            k = rng.integers(0, 3)
            s = "<v%d>" % rng.integers(0, nv)
            if k == 0:
                lines.append("%s %s <C%d> ." % (s, T, rng.integers(0, ncls)))
            elif k == 1:
                lines.append('%s <l%d> "val%d" .' % (s, rng.integers(0, 3), rng.integers(0, 20)))
            else:
                lines.append("%s <e%d> <v%d> ." % (s, rng.integers(0, npred), rng.integers(0, nv)))
        ci = "".join("<C%d>\t%s\n" % (c, ",".join(str(x) for x in sorted(set(rng.integers(0, nparts, size=rng.integers(1, 3)).tolist()))))
                     for c in range(ncls - 1))  # the last class is not indexed
        _run(engine, "\n".join(lines) + "\n", ci, nparts)


def _term_hash(bs: bytes, seed: int) -> int:
    """The library's 64-bit term hash (TermHash in opus_amd/csrc/dx_device.h), restated to build a collision."""
    m = 0xFFFFFFFF
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & m  # noqa: E731
    a, b = 0x9747B28C ^ (seed & m), 0x85EBCA6B ^ (seed >> 32)
    for i in range(0, len(bs), 4):
        k = int.from_bytes(bs[i:i + 4].ljust(4, b"\0"), "little")
        x = (rotl(a ^ ((k * 0xCC9E2D51) & m), 15) * 5 + 0xE6546B64) & m
        a, b = b ^ x, x
    h = ((a << 32) | (b ^ len(bs))) & ((1 << 64) - 1)
    for c in (0xFF51AFD7ED558CCD, 0xC4CEB9FE1A85EC53):
        h ^= h >> 33
        h = (h * c) & ((1 << 64) - 1)
    h ^= h >> 33
    return h or 1


def test_true_hash_collision_is_detected_and_resolved(engine):
    """Two different terms with the same 64-bit hash under the first seed (constructed from the hash's algebra):
    the byte comparison must notice, and the build must start over with another seed and still be exact."""
    s1, s2 = "x?'uZsx5J<MG", "Ybabp)p>`j;p"
    seed0 = 0x243F6A8885A308D3
    assert s1 != s2 and _term_hash(s1.encode(), seed0) == _term_hash(s2.encode(), seed0)
    T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    text = ("%s %s <A> .\n%s %s <B> .\n%s <p> %s .\n%s <q> \"lit\" .\n<c> <p> %s .\n<c> %s <A> .\n" %
            (s1, T, s2, T, s1, s2, s2, s1, T))
    st, got, ref = _run(engine, text, "<A>\t0\n<B>\t1\n", 2)
    assert st.hash_retries >= 1
    assert s1 in got["strs"] and s2 in got["strs"]


def test_measure_files(engine, tmp_path):
    """schema_cc / schema_node_freq.txt / schema_node_count.txt as Loader.loadCC, loadSchemaNodeFreq and
    loadSchemaNodeCount read them (Loader.scala:318-322, 348-365)."""
    from oracle import dap_oracle
    text, ci, n = EDGE_CASES["repeated_types_and_empty_class"]
    engine.partition_text(text.encode(), ci, n)
    sizes, _, _ = engine.measures()
    ref = dap_oracle.run_dap(text, ci, n)
    assert sizes["n_cc"] == len(ref["schema_cc"])
    engine.write_measures(str(tmp_path), "ds", str(tmp_path))
    cc = {}
    for line in open(tmp_path / "ds_data" / "schema_cc" / "part-00000"):
        x = line.rstrip("\n").split(",")  # loadCC: split(","), drop(2) / dropRight(1) around the brackets
        cc[(x[0][2:], x[1], x[2][:-1])] = (int(x[3]), int(x[4][:-1]))
    assert cc == ref["schema_cc"]
    assert os.path.exists(tmp_path / "ds_data" / "schema_cc" / "_SUCCESS")
    for name, key in (("schema_node_freq.txt", "schema_node_freq"), ("schema_node_count.txt", "schema_node_count")):
        rows = [ln.rstrip("\n").split("\t") for ln in open(tmp_path / "ds_local" / name)]
        assert {r[0]: int(r[1]) for r in rows} == ref[key]
        assert [int(r[1]) for r in rows] == sorted((int(r[1]) for r in rows), reverse=True)
    # what exists is left alone (Preprocessor.scala:37-46), what is missing is written again
    os.remove(tmp_path / "ds_local" / "schema_node_count.txt")
    with open(tmp_path / "ds_local" / "schema_node_freq.txt", "w") as f:
        f.write("kept\t1\n")
    engine.write_measures(str(tmp_path), "ds", str(tmp_path))
    assert open(tmp_path / "ds_local" / "schema_node_freq.txt").read() == "kept\t1\n"
    assert os.path.exists(tmp_path / "ds_local" / "schema_node_count.txt")


def _fuzz_text(rng, n_lines):
    """Lines built from adversarial pieces: runs of blanks and tabs (and VT / FF, which java's \\s splits on) between
    the fields, CR / CRLF / LF endings, comments, non-ASCII and non-BMP characters next to the two units that are cut
    from the object, literals with blanks and quotes, very short and very long terms (longer than the tokeniser's
    overhang and longer than a whole tile), repeated triples."""
    T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    blanks = [" ", "\t", "  ", " \t ", "\x0b", "\x0c", " \x0c\t"]
    ends = ["\n", "\r\n", "\r", "\n"]
    subj = ["<s%d>" % i for i in range(12)] + ["<é%d>" % i for i in range(3)] + ["<\U00010400x>", "_:b1", "<" + "L" * 3000 + ">", "<" + "M" * 40000 + ">"]
    cls = ["<A>", "<B>", "<C>", "<Ünï>", "<" + "K" * 1500 + ">"]
    preds = ["<p%d>" % i for i in range(5)] + ["<" + "q" * 160 + ">", "<pé>"]
    lits = ['"x"', '"two words"', '"quo\\"te"@en', '"1"^^<http://www.w3.org/2001/XMLSchema#integer>', '"été"', '"\U0001F600"', '""', '"' + "z" * 2500 + '"']
    tails = [" .", " .", "  .", "\t.", " . ", "..", " ,"]
    out = []
    for _ in range(n_lines):
        k = rng.integers(0, 10)
        if k == 0:
            out.append("# comment " + str(rng.integers(0, 99)) + ends[rng.integers(0, 4)])
            continue
        s = subj[rng.integers(0, len(subj))]
        if k <= 3:
            p, o = T, cls[rng.integers(0, len(cls))]
        elif k <= 6:
            p, o = preds[rng.integers(0, len(preds))], lits[rng.integers(0, len(lits))]
        else:
            p, o = preds[rng.integers(0, len(preds))], subj[rng.integers(0, len(subj))]
        lead = " " if rng.integers(0, 12) == 0 else ""   # a line starting with a blank: an empty first field
        line = lead + s + blanks[rng.integers(0, len(blanks))] + p + blanks[rng.integers(0, len(blanks))] + o + tails[rng.integers(0, len(tails))]
        out.append(line + ends[rng.integers(0, 4)])
        if rng.integers(0, 6) == 0:
            out.append(out[-1])
    text = "".join(out)
    return text[:-1] if rng.integers(0, 2) and text.endswith("\n") and not text.endswith("\r\n") else text


@pytest.mark.parametrize("seed", range(6))
def test_fuzzed_lines_against_oracle(engine, seed):
    from oracle import dap_oracle
    rng = np.random.default_rng(1000 + seed)
    text = _fuzz_text(rng, 400)
    ci = "<A>\t0,2\n<B>\t1\n<Ünï>\t3\n<" + "K" * 1500 + ">\t1,3\n"
    try:
        dap_oracle.run_dap(text, ci, 4)
    except (ValueError, IndexError):
        pytest.skip("this seed produced a line the reference throws on")
    _run(engine, text, ci, 4)


@pytest.mark.parametrize("seed", range(4))
def test_control_characters_inside_and_around_fields(engine, seed):
    """Fields made of arbitrary ASCII including NUL and the other control characters: java's split breaks only on
    [ \\t\\n\\x0B\\f\\r], trim strips every unit <= U+0020 from the ends of a field, and a control character in the
    middle belongs to the term."""
    rng = np.random.default_rng(50 + seed)
    T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    ctrl = [chr(c) for c in (0, 1, 2, 7, 8, 0x0e, 0x1b, 0x1c, 0x1f, 0x7f)]
    body = list("abcXYZ019<>:/#_-\"'") + ctrl
    seps = [" ", "\t", "\x0b", "\x0c", "  ", "\t \x0c"]

    def tok(n):
        t = "".join(body[rng.integers(0, len(body))] for _ in range(n))
        return t if t.strip("".join(chr(c) for c in range(0x21))) else "x" + t   # not empty after trim

    names = [tok(int(rng.integers(1, 9))) for _ in range(25)]
    lines = []
    for _ in range(300):
        s = names[rng.integers(0, len(names))]
        k = rng.integers(0, 3)
        if k == 0:
            p, o = T, "<C%d>" % rng.integers(0, 4)
        elif k == 1:
            p, o = "<p%d>" % rng.integers(0, 3), names[rng.integers(0, len(names))]
        else:
            p, o = tok(int(rng.integers(1, 6))), '"' + tok(int(rng.integers(0, 10))) + " " + tok(2) + '"'
        pad = ctrl[rng.integers(0, len(ctrl))] if rng.integers(0, 3) == 0 else ""
        lines.append(s + pad + seps[rng.integers(0, len(seps))] + p + seps[rng.integers(0, len(seps))] + pad + o + pad + " .")
    text = "\n".join(lines) + "\n"
    ci = "<C0>\t0\n<C1>\t1\n<C2>\t0,1\n"
    _run(engine, text, ci, 2)


@pytest.mark.parametrize("boundary", [16, 1024, 8192, 32768, 65536])
def test_line_ends_on_every_kind_of_boundary(engine, boundary):
    """LF, CR LF and a lone CR placed just before, across and just after the edges the line scan works in (a lane's 16
    bytes, a wave's 1 KiB step, a wave's 8 KiB stretch, a 32 KiB tile), each followed by a comment line or a triple
    that starts exactly there."""
    T = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
    ci = "<A>\t0\n<B>\t1\n"
    head = "<s0> " + T + " <A> .\n<s1> " + T + " <B> .\n<s0> <p> <s1> .\n"
    for shift in (-2, -1, 0, 1):
        for end in ("\n", "\r\n", "\r"):
            for follower in ("# a comment\n<s2> <p> <s0> .\n", "<s3> <q> \"lit\" .\n# tail\n"):
                # a triple with a long literal whose line terminator begins at byte boundary + shift
                stem = head + "<s1> <name> \""
                edge = -(-(len(stem) + 8) // boundary) * boundary   # the first such edge the stem leaves room for
                fill = edge + shift - len(stem) - len("\" .")
                text = stem + "x" * fill + "\" ." + end + follower + "<s1> <p> <s0> ." + end
                assert text.index(end, len(stem)) == edge + shift
                _run(engine, text, ci, 2)
