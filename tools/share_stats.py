"""Term sharing inside a 1000-query batch (host planning only, no GPU): how many (query, index term) pairs, how many distinct terms, how much of the work the
most shared terms cover — the numbers behind DESIGN.md section 7 g1 (the dense MFMA formulation needs shared terms).  python tools/share_stats.py"""
import numpy as np, collections, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from infidex_amd import SearchEngine
from tools.synth import Synth
s = Synth(4, docs=1_000_000)
arena, offs = s.docs()
e = SearchEngine.create_default(device=-1)
e.index_flat(None, arena, offs, s.field_weights)
qa, qo = s.queries(1000, qseed=1000)
cnt = collections.Counter(); tot = 0; per = []
for q in Synth.texts(qa, qo):
    p = e.plan(q)
    ids = [int(t) for t in p["term_ids"] if t >= 0]
    per.append(len(ids)); tot += len(ids); cnt.update(ids)
top = cnt.most_common()
print("queries 1000, (query, index term) pairs", tot, "distinct terms", len(cnt), "mean terms/query %.1f" % np.mean(per))
for k in (16, 64, 256, 1024):
    c = sum(v for _, v in top[:k]); print("top-%d terms cover %d pairs = %.1f %%; the k-th is used by %d queries" % (k, c, 100.0 * c / tot, top[min(k, len(top)) - 1][1]))
# dense tile for the top-64 terms: fraction of (query, term) cells that are nonzero
k = 64; ids = [t for t, _ in top[:k]]
print("Q block [1000 x 64] of the top-64 terms: nonzeros %.2f %%" % (100.0 * sum(v for _, v in top[:k]) / (1000 * 64)))
