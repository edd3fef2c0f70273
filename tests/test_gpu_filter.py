"""Config 5 on the GPU: Infiscript post-filter + NumberOfDocumentsInFilter + facet aggregation on device-resident columns, against the oracle
(oracle/filter.hpp, pinned by the reference's BytecodeVM / FilterParser / Ternary / Faceting tests in tests/test_oracle_filter_kats.py)."""
import numpy as np
import pytest

from infidex_amd import SearchEngine, Document, Query
from infidex_amd.engine import InfidexError
from tests import oracle_lib as O
from tests.test_oracle_filter_kats import VM_KATS, BCL_KATS
from tools.synth import Synth

pytestmark = pytest.mark.gpu
GENRES = ["Action", "Comedy", "Drama", "Horror", "Sci-Fi", "Romance", "Thriller", "Western", "Fantasy", "Mystery", "Crime", "Animation"]


def columns(n, seed=5):
    rng = np.random.default_rng(seed)
    year = rng.integers(1950, 2025, n).astype(np.int64)
    rating = np.round(rng.uniform(1.0, 10.0, n), 1)
    genre = [GENRES[i] for i in rng.integers(0, len(GENRES), n)]
    return year, rating, genre


@pytest.fixture(scope="module")
def pair():
    s = Synth(2, docs=40000)
    arena, offs = s.docs()
    e = SearchEngine.create_default(device=0); e.index_flat(None, arena, offs, s.field_weights)
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    year, rating, genre = columns(40000)
    for x in (e, o):
        x.set_column("year", year, facetable=True); x.set_column("rating", rating, facetable=False); x.set_column("genre", genre, facetable=True)
    qa, qo = s.queries(120, qseed=31, fuzz=0.3)
    return e, o, Synth.texts(qa, qo), (year, rating, genre)


@pytest.mark.parametrize("flt", ["year >= 2000 AND rating > 7.0", "genre IN ('Drama', 'crime') OR year < 1960", "NOT (rating <= 5) AND genre != 'Horror'",
                                 "year BETWEEN 1990 AND 1999", "genre STARTS WITH 'S' OR genre LIKE '%er'", "rating >= 9.5 ? genre = 'Action' : year >= 2020",
                                 "rating = 7", "rating = '7.0'", "nosuchfield IS NULL AND year > 2010", None])
def test_filter_and_facets_match_the_oracle(pair, flt):
    e, o, texts, cols = pair
    res = e.search_filtered(texts, 20, filter=flt, enable_facets=True)
    for q, r in zip(texts, res):
        w = o.search_filtered(q, 20, filter=flt, enable_facets=True)
        assert [x.document_id for x in r.records] == w["keys"], (flt, q)
        assert r.total_in_filter == w["in_filter"], (flt, r.total_in_filter, w["in_filter"])
        assert (r.facets or {}) == w["facets"], (flt, q, r.facets, w["facets"])


def test_vm_known_answers_through_the_device():
    """Every VM known answer of the reference (restated in test_oracle_filter_kats.VM_KATS) evaluated by the product: one document whose fields are
    the KAT's, the expression as the post-filter of a query that returns it."""
    for expr, fields, want in VM_KATS + BCL_KATS:
        e = SearchEngine.create_default(device=0); e.index_documents([Document(1, "alpha bravo"), Document(2, "charlie delta")])
        for name, v in fields.items():
            if v is None:
                continue                                    # a null field = no column value: the engine treats an absent column as null
            if isinstance(v, float): e.set_column(name, np.array([v, v], np.float64))
            elif isinstance(v, int): e.set_column(name, np.array([v, v], np.int64))
            else: e.set_column(name, [v, v])
        r = e.search(Query("alpha bravo", 10, filter=expr))
        assert ([x.document_id for x in r.records] == [1]) is want, (expr, fields, r.records)
        assert r.total_in_filter == (2 if want else 0), (expr, r.total_in_filter)


def test_syntax_errors_and_unsupported():
    e = SearchEngine.create_default(device=0); e.index_documents([Document(1, "alpha bravo")]); e.set_column("a", ["1"])
    for bad in ["score >= 90 ? 'high'", "genre = ", "(a = '1'", "a # '1'", ""]:
        with pytest.raises(InfidexError):
            e.search(Query("alpha", 10, filter=bad))
    with pytest.raises(InfidexError):
        e.search(Query("alpha", 10, filter="a MATCHES '^1'"))
    assert [x.document_id for x in e.search(Query("alpha", 10, filter="a = '1'")).records] == [1]      # the session recovers after an error


def test_in_filter_count_follows_deletions():
    """Filter.NumberOfDocumentsInFilter counts the documents that are not Deleted, and a filter used after a deletion counts again."""
    docs = [Document(k, "alpha bravo %d" % k) for k in range(1, 9)]
    year = np.array([1990, 1995, 2000, 2005, 2010, 2015, 2020, 2025], np.int64)
    e = SearchEngine.create_default(device=0); e.index_documents(docs); e.set_column("year", year, facetable=True)
    o = O.OracleEngine.create_default(); o.index([(d.document_key, d.fields) for d in docs]); o.set_column("year", year, facetable=True)
    for step in range(2):
        r = e.search(Query("alpha", 10, filter="year >= 2000")); w = o.search_filtered("alpha", 10, filter="year >= 2000")
        assert r.total_in_filter == w["in_filter"] == (6 if step == 0 else 4)
        assert [x.document_id for x in r.records] == w["keys"]
        e.delete_documents([3, 8]); o.delete_keys([3, 8])
    e.restore_documents()
    assert e.search(Query("alpha", 10, filter="year >= 2000")).total_in_filter == 6
    # a column added later: filters compiled before it existed treated the field as null and are compiled again
    assert e.search(Query("alpha", 10, filter="tag = 'x'")).total_in_filter == 0
    e.set_column("tag", ["x", "y"] * 4)
    assert e.search(Query("alpha", 10, filter="tag = 'x'")).total_in_filter == 4


def test_book_library_cases_of_the_reference():
    """FacetingTests.cs:108-560 through the device: the reference's own assertions (tests/book_library.py) on the product's rows and facets, and row for
    row / facet for facet equality with the oracle — multi-field documents, a string column compared numerically (year >= '2000'), OR / IN / nested
    filters, NumberOfDocumentsInFilter."""
    from infidex_amd.engine import Field
    from tests import book_library as BL
    from tests.test_oracle_filter_kats import _oracle_books
    keys, texts, cols = BL.book_fields()
    e = SearchEngine.create_default(device=0)
    e.index_documents([Document(k, [Field(n, t, w) for n, t, w in zip(("title", "author", "genre", "description"), ts, BL.BOOK_WEIGHTS)]) for k, ts in zip(keys, texts)])
    for name, (vals, fac) in cols.items():
        e.set_column(name, vals, facetable=fac)
    o = _oracle_books()
    for case in BL.CASES:
        name, _, query, k, flt, *_ = case
        r = e.search(Query(query, k, filter=flt, enable_facets=True))
        got = [x.document_id for x in r.records]
        BL.check_case(case, got, r.facets)
        w = o.search_filtered(query, k, filter=flt, enable_facets=True)
        assert got == w["keys"] and (r.facets or {}) == w["facets"] and r.total_in_filter == w["in_filter"], (name, got, w)
    assert e.search(Query("magic", 20)).facets is None                                     # Facets_NotReturnedWhenDisabled


def test_sharded_filter_and_facets_match_the_oracle():
    """Config 5 on document shards (SearchEngine.cs:298-316 after the merge): the post-filter and the facet counts run on the merged rows in phase 4 on
    every rank, Filter.NumberOfDocumentsInFilter is the sum of the shards' device counts.  Rows, counts and facets must be the oracle's."""
    from infidex_amd.engine import pack_texts
    from infidex_amd.sharded import create_sharded_engine, ShardSession, simulate_shards_dev, simulate_set_filter
    n = 150000
    s = Synth(4, docs=n); arena, offs = s.docs()
    year, rating, genre = columns(n, seed=9)
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    W = 3
    engs = [create_sharded_engine(r, W, 0) for r in range(W)]
    for x in engs + [o]:
        if x is not o:
            x.index_flat(None, arena, offs, s.field_weights)
        x.set_column("year", year, facetable=True); x.set_column("rating", rating, facetable=False); x.set_column("genre", genre, facetable=True)
    sess = [ShardSession(e) for e in engs]
    qa, qo = s.queries(80, qseed=77)
    texts = Synth.texts(qa, qo); a, off = pack_texts(texts)
    for flt in ["year >= 2000 AND rating > 7.0", "genre IN ('Drama', 'crime') OR year < 1960", None]:
        nin = simulate_set_filter(sess, flt, True)
        res = simulate_shards_dev(sess, a, off, 20)
        for r in res[1:]:
            for x, y in zip(r, res[0]):
                assert np.array_equal(x, y)
        keys, scores, ties, counts, flags = res[0]
        for i, q in enumerate(texts):
            w = o.search_filtered(q, 20, filter=flt, enable_facets=True)
            assert keys[i, :int(counts[i])].tolist() == w["keys"], (flt, q)
            assert nin == w["in_filter"], (flt, nin, w["in_filter"])
            for ss in sess:
                assert (ss.facets(i) or {}) == w["facets"], (flt, q)
    simulate_set_filter(sess, None, False)
