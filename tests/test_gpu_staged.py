"""GPU parity cases written when no GPU time was left to run them (round 4): they are SKIPPED unless INFX_RUN_STAGED=1, so the recorded GPU suite holds only tests
that have run green on an MI355X.  First thing to run in the next round: `INFX_RUN_STAGED=1 python -m pytest tests/test_gpu_staged.py -m gpu`; what passes moves into
test_gpu_parity.py."""
import os

import numpy as np
import pytest

from infidex_amd import SearchEngine, Document
from tests import oracle_lib as O

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("INFX_RUN_STAGED") != "1", reason="staged: not yet run on a GPU (set INFX_RUN_STAGED=1)")]

FINAL_ATOL = 2.0 ** -6 + 1e-6

ASTRAL_DOCS = [(1, "\U0001F50Dab zeta"), (2, "\U0001F50Eab yotta"), (3, "plain \U0001F50Dab"), (4, "x\U0001F50D \U0001F50Ex \U0001F50Dab"), (5, "\U0001F50D"),
               (6, "\U00020000\U00020001 cjk\U00020001"), (7, "�ab already replaced"), (8, "x\U00020000 end"), (20, "emoji \U0001F600\U0001F601 party \U0001F600"),
               (21, "\U0001D49C\U0001D4B7\U0001D4B8 math script"), (22, "mixed a\U0001F50Db c\U0001F50Ed"), (23, "\ud83d lone high"), (24, "lone low \udd0d tail")]
ASTRAL_QUERIES = ["\U0001F50Dab", "\U0001F50Eab zeta", "a\U0001F50Db", "emoji \U0001F600", "\U0001D49C\U0001D4B7\U0001D4B8", "x\U0001F50D", "\U0001F50Dxb", "cjk\U00020001",
                  "party \U0001F601\U0001F600", "\U0001F50Dab\U0001F50E", "\ud83d lone", "low \udd0d", "math script", "mixed"]


def test_characters_outside_the_bmp_on_gpu():
    """Surrogate pairs and lone surrogates through the whole device pipeline: rows and scores equal the oracle's (the host side is covered on CPU by
    test_host_parity.py::test_characters_outside_the_bmp_index_and_plan_like_the_oracle)."""
    o = O.OracleEngine.create_default(); o.index(ASTRAL_DOCS)
    e = SearchEngine.create_default(device=0); e.index_documents([Document(k, t) for k, t in ASTRAL_DOCS])
    for q, r in zip(ASTRAL_QUERIES, e.search_batch(ASTRAL_QUERIES, 10)):
        w = o.search(q, 10)
        if w["unsupported"]:
            assert r.records == [] or len(r.records) == 0, q
            continue
        assert [x.document_id for x in r.records] == w["keys"], (q, r.records, w)
        assert np.allclose([x.score for x in r.records], w["scores"], rtol=0, atol=FINAL_ATOL), (q, r.records, w)
