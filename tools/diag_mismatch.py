"""Diagnose top-k set differences between the GPU path and the oracle on a synthetic corpus (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.synth import Synth
from tests import oracle_lib as O
from infidex_amd import SearchEngine

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
s = Synth(4, docs=docs)
arena, offs = s.docs()
e = SearchEngine.create_default(device=0); e.index_flat(None, arena, offs, s.field_weights)
o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
qa, qo = s.queries(nq, qseed=1000)
qs = Synth.texts(qa, qo)
res = e.search_batch(qs, 20)
bad = 0
for qi, q in enumerate(qs):
    r = o.search(q, 20)
    g = [x.document_id for x in res[qi].records]
    if set(g) == set(r["keys"]):
        continue
    bad += 1
    ok, osc = o.last_stage1(); gk, gsc = e.last_stage1(qi)
    same_s1 = set(ok.tolist()) == set(gk.tolist())
    cut_o = float(osc.min()) if len(osc) else 0.0
    ties_at_cut = int((osc == cut_o).sum())
    gs = sorted([round(x.score, 4) for x in res[qi].records], reverse=True); os_ = sorted([round(float(x), 4) for x in r["scores"]], reverse=True)
    print(f"--- q{qi} '{q}' stage1_same={same_s1} n_s1=({len(ok)},{len(gk)}) cut={cut_o:.6f} ties_at_cut={ties_at_cut} "
          f"s1_multiset_equal={np.array_equal(np.sort(osc), np.sort(gsc))} final_scores_equal={gs == os_} stats={o.last_stats()}")
    if bad <= 6:
        print("   gpu:", [(x.document_id, round(x.score, 3), x.tiebreaker) for x in res[qi].records][:8])
        print("   orc:", list(zip(r["keys"], np.round(r["scores"], 3).tolist(), r["ties"].tolist()))[:8])
        d = set(ok.tolist()) ^ set(gk.tolist())
        od = dict(zip(ok.tolist(), osc.tolist())); gd = dict(zip(gk.tolist(), gsc.tolist()))
        print("   s1 symdiff:", [(k, od.get(k), gd.get(k)) for k in list(d)[:6]])
print("mismatching queries:", bad, "of", len(qs))
