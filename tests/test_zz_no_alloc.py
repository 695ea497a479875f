"""dx_query_count_noalloc (the entry point that allocates nothing) against dx_query, through the C ABI.

The statement's device scratch comes from one fixed array per GPU, which keeps the previous call's
contents: the checks therefore run back to back on one context, larger inputs first, and repeat a
statement after a smaller one has left its rows behind.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def loaded(lubm1):
    from opus_amd import lubm
    from opus_amd.engine import Engine
    from opus_amd.host import translator as T
    from oracle.query_oracle import parse_translated_file
    from parity_util import engine_artifacts
    e = Engine(0)
    e.partition_text(lubm1["bytes"], lubm1["class_index"], 4)
    art = engine_artifacts(e)
    ni = {}
    for u, c in art["node_index"]:
        ni.setdefault(u, []).append(c)
    ci = T.load_class_index(lubm1["class_index"])
    plans = {}
    for name, text in lubm.queries(0).items():
        qm, partitions, _ = parse_translated_file(T.translate_query(name, text, lambda u: ni.get(u, []), ci, art["partition_stat"]))
        plans[name] = (next(iter(qm.values())), partitions)
    yield e, plans
    e.close()


def _input_rows(e, partitions):
    e.unload_all()
    return e.load_tables(partitions)


def test_counts_equal_dx_query_on_calls_in_a_row(loaded, lubm1):
    from oracle.query_oracle import run_query  # noqa: F401  (the oracle stays the checker of dx_query elsewhere)
    e, plans = loaded
    # order the 13 statements by the rows they read, largest first: every call after the first runs over a
    # scratch array that still holds a larger statement's counters, heads and rows
    sized = sorted(((_input_rows(e, parts), name) for name, (sql, parts) in plans.items()), reverse=True)
    assert sized[0][0] > sized[-1][0]
    seen_nonzero = 0
    for rows, name in sized:
        sql, parts = plans[name]
        assert _input_rows(e, parts) == rows
        want = e.query(sql).count
        got = e.query_count_noalloc(sql)
        assert got == want, (name, got, want)
        seen_nonzero += want > 0
    assert seen_nonzero >= 8
    # the largest again, right after the smallest
    big_sql, big_parts = plans[sized[0][1]]
    small_sql, small_parts = plans[sized[-1][1]]
    _input_rows(e, small_parts)
    small = e.query_count_noalloc(small_sql)
    assert small == e.query(small_sql).count
    _input_rows(e, big_parts)
    assert e.query_count_noalloc(big_sql) == e.query(big_sql).count


@pytest.mark.parametrize("unfused", [False, True])
def test_a_call_takes_nothing_from_any_allocator(loaded, monkeypatch, unfused):
    """The entry point's one promise: between entering and leaving it the context's allocator hands out no buffer (not
    even a cached one), holds the same bytes, and the device's free memory is what it was (no hipMalloc / hipFree by
    anybody in the process). Checked for statements of every length, with the statement's preparation riding in its
    first kernel and as kernels of its own (DX_QUERY_UNFUSED), and against dx_query, which does allocate."""
    import torch
    e, plans = loaded
    if unfused:
        monkeypatch.setenv("DX_QUERY_UNFUSED", "1")
    for name in ("Q9.txt", "Q2.txt", "Q4.txt", "Q1.txt", "Q6.txt"):
        sql, parts = plans[name]
        _input_rows(e, parts)
        want = e.query(sql).count               # also gathers whatever statistics the plan needs, as the contract asks
        e.query_count_noalloc(sql)              # the first call may still size things on the host
        torch.cuda.synchronize()
        before = (e.allocator_stats(), torch.cuda.mem_get_info()[0])
        for _ in range(3):
            assert e.query_count_noalloc(sql) == want
        torch.cuda.synchronize()
        assert (e.allocator_stats(), torch.cuda.mem_get_info()[0]) == before, name
        r = e.query(sql)                        # the allocating entry point does move the counter
        assert e.allocator_stats()[0] > before[0][0]
        r.close()


def test_two_contexts_of_one_gpu_share_the_array_one_after_another(loaded):
    from opus_amd.engine import Engine
    e, plans = loaded
    sql, parts = plans["Q9.txt"]
    _input_rows(e, parts)
    want = e.query(sql).count
    e2 = Engine(0)
    try:
        terms = ["<a>", "<b>", "<c>", "<p>"]
        data = np.frombuffer("".join(terms).encode(), dtype=np.uint8)
        offs = np.cumsum([0] + [len(t) for t in terms]).astype(np.uint64)
        e2.dictionary_load(data, offs)
        e2.load_table_ids("00000-_3=_p_", np.array([0, 1, 1], dtype=np.int64), np.array([1, 2, 0], dtype=np.int64))
        e2.load_tables(["00000-_3=_p_"])
        small = "SELECT tab0.X AS X,tab1.Z AS Z FROM (SELECT s AS X, o AS Y FROM table00000__3_E__p_) AS tab0, " \
                "(SELECT s AS Y, o AS Z FROM table00000__3_E__p_) AS tab1 WHERE tab0.Y=tab1.Y "
        assert e2.query_count_noalloc(small) == e2.query(small).count == 3  # a->b->{c,a}, b->a->b
    finally:
        e2.close()
    assert e.query_count_noalloc(sql) == want


def test_rejections(loaded):
    from opus_amd.engine import DxError, Engine
    e, plans = loaded
    e2 = Engine(0)
    try:
        n = 3 << 20  # two 24 MiB columns of intermediate: more than the 32 MiB array
        terms = ["<t%d>" % i for i in range(4)]
        data = np.frombuffer("".join(terms).encode(), dtype=np.uint8)
        offs = np.cumsum([0] + [len(t) for t in terms]).astype(np.uint64)
        e2.dictionary_load(data, offs)
        s = (np.arange(n, dtype=np.int64) % 4)
        o = ((np.arange(n, dtype=np.int64) // 4) % 4)
        e2.load_table_ids("00000-_3=_big_", s, o)
        sql = "SELECT tab0.X AS X,tab0.Y AS Y,tab1.Z AS Z FROM (SELECT s AS X, o AS Y FROM table00000__3_E__big_ WHERE s == '<t1>') AS tab0, " \
              "(SELECT s AS Y, o AS Z FROM table00000__3_E__big_ WHERE o == '<t2>') AS tab1 WHERE tab0.Y=tab1.Y "
        with pytest.raises(DxError) as ei:   # statistics not gathered yet: dx_load_table_ids does not count them
            e2.query_count_noalloc(sql)
        assert ei.value.code == -22
        e2.load_tables(["00000-_3=_big_"])
        with pytest.raises(DxError) as ei:
            e2.query_count_noalloc(sql)
        assert ei.value.code == -34 and "32 MiB" in str(ei.value)
        assert e2.query(sql).count > 0       # the allocating entry point takes it, and the context is still usable
        # a statement that fits runs on the same context afterwards
        one = "SELECT tab0.X AS X FROM (SELECT s AS X FROM table00000__3_E__big_ WHERE o == '<t3>') AS tab0 "
        assert e2.query_count_noalloc(one) == e2.query(one).count == n // 4
    finally:
        e2.close()
