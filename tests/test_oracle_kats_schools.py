"""Oracle vs the reference's real-data suite (SchoolSearchParityTests.cs) — see tests/school_kats.py."""
import pytest

from tests import oracle_lib as O
from tests import school_kats as S


@pytest.fixture(scope="module")
def school_oracle():
    names = S.load_names()
    o = O.OracleEngine.create_default()
    for a, b in S.SYNONYMS:
        o.add_synonym(a, b)
    o.index([(i, n) for i, n in enumerate(names)])
    return names, o


def test_school_suite_on_the_oracle(school_oracle):
    names, o = school_oracle

    def search(q, k):
        r = o.search(q, k)
        return list(zip(r["keys"], [float(x) for x in r["scores"]]))
    S.check_all(search, names)
