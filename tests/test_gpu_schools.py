"""The reference's real-data relevance suite (SchoolSearchParityTests.cs, tests/school_kats.py) on the GPU path, and full parity with
the oracle on the same corpus: real Czech text with diacritics, 7 629 documents, three synonym pairs."""
import pytest

from infidex_amd import SearchEngine, Document
from tests import oracle_lib as O
from tests import school_kats as S
from tests.test_gpu_parity import compare_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def schools():
    names = S.load_names()
    e = SearchEngine.create_default(device=0, want_features=True)
    o = O.OracleEngine.create_default()
    for a, b in S.SYNONYMS:
        e.add_synonym(a, b); o.add_synonym(a, b)
    e.index_documents([Document(i, n) for i, n in enumerate(names)])
    o.index([(i, n) for i, n in enumerate(names)])
    return names, e, o


def test_school_suite_on_the_gpu(schools):
    names, e, _ = schools

    def search(q, k):
        r = e.search(q, k)
        return [(x.document_id, float(x.score)) for x in r.records]
    S.check_all(search, names)


def test_school_queries_match_the_oracle(schools):
    names, e, o = schools
    qs = ["mateřská škola lázně bělohrad", "bělohrad lázně mateřská škola", "bel", "belo", "belohradska", "sciozlí", "scio škola ve zlíně",
          "sciozlínskáškola", "sciozlín", "scioškola br", "scioškola če", "škola zlín s", "tyršovka česká lípa", "zlínská scioškola",
          "zlímská scioškola", "scio škola a", "škola scio z", "gympl praha", "zs brno", "ss technická ostrava", "ZŠ a MŠ Kolín",
          "základní umělecká škola", "gymnázium jana nerudy", "materska skola", "střední průmyslová škola elektrotechnická"]
    for k in (20, 50):
        st = compare_batch(e, o, qs, k)
        assert st["set_mismatch"] == 0 and st["feat_mismatch"] == 0, st
        assert st["order_unclassified"] == 0 and st["order_mismatch"] <= 1, st      # a flip, if any, is a classified 2^-6 near-tie
