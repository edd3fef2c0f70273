"""Classification of a query whose final top-k DocumentId set differs between the HIP path and the oracle.

Used by tests/test_gpu_scale.py and by bench.py's reported set comparison.  A difference is a "tie-at-cut-off" only if the Stage-1
top-`depth` sets differ and every document in their symmetric difference scores within SCORE_RTOL*4 of the oracle's cut-off score
(the reference's own arithmetic is position dependent there: quirk Q9 + the BCL heap order).  Anything else is
"other" and is a parity failure.
"""
import numpy as np

SCORE_RTOL = 2e-6 * 32


def classify(engine, oracle, queries, k, depth=500):
    """queries: texts whose final sets differ.  Returns a list of dicts {query, kind, detail}."""
    out = []
    engine.set_introspection(True)
    try:
        for q in queries:
            res = engine.search_batch([q], k, depth)[0]
            r = oracle.search(q, k, depth)
            ok, osc = oracle.last_stage1()
            gk, gsc = engine.last_stage1(0)
            od = dict(zip(ok.tolist(), osc.tolist())); gd = dict(zip(gk.tolist(), gsc.tolist()))
            got = [x.document_id for x in res.records]
            if set(got) == set(r["keys"]):
                out.append({"query": q, "kind": "identical-on-rerun", "detail": ""})
                continue
            sym = set(od) ^ set(gd)
            cut = min(osc) if len(osc) else 0.0
            if sym and all(abs(od.get(d, gd.get(d)) - cut) <= SCORE_RTOL * 4 * max(abs(cut), 1.0) for d in sym):
                exact = sum(1 for d in sym if np.float32(od.get(d, gd.get(d))) == np.float32(cut))
                out.append({"query": q, "kind": "tie-at-cut-off",
                            "detail": f"stage-1 symmetric difference {len(sym)} docs at cut {cut!r} ({exact} bit-equal to it); final diff {sorted(set(got) ^ set(r['keys']))}"})
            else:
                out.append({"query": q, "kind": "other",
                            "detail": f"stage-1 symmetric difference {len(sym)} docs, cut {cut!r}; final got {got} want {r['keys']}"})
    finally:
        engine.set_introspection(False)
    return out


FINAL_SCORE_TOL = 2.0 ** -6 + 1e-6     # fp32 quantisation of (float)precedence + semantic once precedence >= 2^17 (FusionScorer.cs:218)


def assert_final_rows_match_oracle(keys, scores, counts, oracle, texts, k, depth=500, what=""):
    """Final rows of a product batch against the oracle, query by query: identical DocumentId SETS; identical ORDER unless the rows that moved are
    2^-6 near-ties — every document's score must then lie within FINAL_SCORE_TOL of the oracle's score for it (an order flip between rows whose
    scores differ by more than the quantisation step fails).  Returns (identical order, classified flips)."""
    same = flips = 0
    for i, q in enumerate(texts):
        r = oracle.search(q, k, depth)
        got = keys[i, :int(counts[i])].tolist()
        assert set(got) == set(r["keys"]), (what, q, got, r["keys"])
        gs = dict(zip(got, scores[i, :len(got)].tolist())); os_ = dict(zip(r["keys"], r["scores"]))
        assert all(abs(gs[d] - os_[d]) <= FINAL_SCORE_TOL for d in got), (what, q, gs, os_)
        if got == r["keys"]:
            same += 1
            continue
        flips += 1
        for pos, (a, b) in enumerate(zip(got, r["keys"])):       # a flipped position holds two documents whose oracle scores are one quantisation step apart at most
            if a != b:
                assert abs(os_[a] - os_[b]) <= FINAL_SCORE_TOL, (what, q, pos, a, b, os_[a], os_[b])
    return same, flips
