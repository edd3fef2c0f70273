"""Several documents under one DocumentKey (the reference's segmented documents, SegmentTrackingTests.cs:92-210, 324-345): product vs oracle."""
import numpy as np
from infidex_amd import SearchEngine
from infidex_amd.engine import Document
from tests import oracle_lib as O

CASES = [
    ([(1, "Introduction to the topic of animals"), (1, "The quick brown fox jumps over the lazy dog"), (1, "Conclusion and summary of findings")], ["fox", "summary animals"]),
    ([(1, "Introduction chapter one"), (1, "Batman fights crime in Gotham City"), (1, "Conclusion chapter one"), (2, "Batman and Robin save the day"),
      (2, "The end of their adventure"), (3, "Superman flies faster than a speeding bullet")], ["batman", "chapter one", "batman robin"]),
    ([(1, "The cat sat on the mat"), (1, "The dog ran through the park"), (1, "The bird flew in the sky")], ["batman", "the dog"]),
    ([(1, "Chapter 1 introduction"), (1, "The hero begins his journey"), (2, "The hero saves the day"), (3, "A story about courage")], ["hero", "hero journey"]),
    ([(1, f"Segment {i} text content") if i != 5 else (1, "This segment contains batman") for i in range(10)], ["batman", "segment text"]),
]
same = diff = 0
for docs, queries in CASES:
    o = O.OracleEngine.create_default(); o.index(docs)
    e = SearchEngine.create_default(device=0); e.index_documents([Document(k, t) for k, t in docs])
    for q, r in zip(queries, e.search_batch(queries, 10)):
        w = o.search(q, 10)
        got = [(x.document_id, round(float(x.score), 3)) for x in r.records]
        want = list(zip(w["keys"], [round(float(s), 3) for s in w["scores"]]))
        ok = got == want
        same += ok; diff += (not ok)
        print("OK  " if ok else "DIFF", repr(q), "product", got, "oracle", want)
print("same", same, "diff", diff)
