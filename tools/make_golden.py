#!/usr/bin/env python3
"""Generates tests/golden/*.json — committed input/output vectors of the hot path.

* reference_kats.json: corpora, queries and expected DocumentId lists that the reference's OWN tests hold
  (src/Infidex.Tests/ReferenceMatchingTests.cs:39-98, QueryTests.cs:150-277, SearchEngineTests.cs:37-54) — data, restated.
* synth_cfg{2,3}_small.json: seeded synthetic corpora (tools/synth.py, BASELINE configs 2 and 3 scaled down) with the oracle's
  results (keys, scores, tiebreakers) for a fixed query set.  The oracle is pinned by the reference KATs (tests/test_oracle_kats.py);
  these vectors pin the oracle against regressions and let the GPU path be checked against committed data.

Run from the repo root:  python tools/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import oracle_lib as O          # noqa: E402
from tests.test_oracle_kats import TEN_DOCS  # noqa: E402
from tools.synth import Synth              # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def reference_kats():
    return {
        "source": "src/Infidex.Tests/ReferenceMatchingTests.cs:39-98 (corpus and exact result lists)",
        "docs": [[k, t] for k, t in TEN_DOCS],
        "max_results": 10,
        "cases": [
            {"query": "batman", "first": 6},
            {"query": "qick fux", "keys": [5, 1]},
            {"query": "battamam", "keys": [6]},
            {"query": "new york", "keys": [8]},
            {"query": "speeding", "keys": [7]},
        ],
    }


def synth_vectors(cfg, docs, nq, k, qseed):
    s = Synth(cfg, docs=docs)
    arena, offs = s.docs()
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    qa, qo = s.queries(nq, qseed=qseed)
    texts = Synth.texts(qa, qo)
    cases = []
    for q in texts:
        r = o.search(q, k, 500)
        cases.append({"query": q, "keys": [int(x) for x in r["keys"]], "scores": [float(x) for x in r["scores"]],
                      "used_coverage": bool(r["used_coverage"])})
    return {"generator": "tools/make_golden.py", "config": cfg, "docs": docs, "query_seed": qseed, "max_results": k, "coverage_depth": 500,
            "corpus": "tools.synth.Synth(config, docs=docs).docs()  (SplitMix64-seeded, deterministic)", "cases": cases}


def main():
    os.makedirs(OUT, exist_ok=True)
    json.dump(reference_kats(), open(os.path.join(OUT, "reference_kats.json"), "w"), indent=1)
    json.dump(synth_vectors(2, 3000, 60, 10, 101), open(os.path.join(OUT, "synth_cfg2_small.json"), "w"), indent=0)
    json.dump(synth_vectors(3, 2000, 60, 20, 102), open(os.path.join(OUT, "synth_cfg3_small.json"), "w"), indent=0)
    print("written:", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
