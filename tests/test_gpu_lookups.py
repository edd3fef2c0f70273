"""Dictionary lookups of query planning on the device (csrc/lookup.hip.inc) against the oracle:
   k_ld1  == FstIndex.MatchWithinEditDistance1  (oracle trie walk: count and the first 1024 term ids, in order)
   k_wm   == WordMatcherLookup.Execute          (oracle/wordmatcher.hpp: the union of the exact / LD1 / affix doc-id sets)
on the reference's KAT corpus, the school names (diacritics, synonyms), random Unicode corpora and the synthetic configurations; and the whole search
with the lookups on the device equals the search with them on the host."""
import os
import subprocess
import sys

import numpy as np
import pytest

from infidex_amd import SearchEngine, Document
from infidex_amd.engine import normalize as _norm
from tests import oracle_lib as O
from tests import school_kats as SK
from tests import unicode_corpus as U
from tools.synth import Synth

pytestmark = pytest.mark.gpu


def _pair(docs, synonyms=()):
    e = SearchEngine.create_default(device=0)
    o = O.OracleEngine.create_default()
    for a, b in synonyms:
        e.add_synonym(a, b); o.add_synonym(a, b)
    e.index_documents([Document(k, t) for k, t in docs])
    o.index(docs)
    assert e.device_lookups()
    return e, o


def _check_words(e, o, words, handed_back):
    for w in words:
        c2, m2 = o.match_ld1(w)
        c1, m1 = e.match_ld1_device(w)
        if c1 < 0:                      # the kernel may hand a word back (work lists outgrown / > 64 characters); the product then runs the host walk
            handed_back.append(w)
            c1, m1 = e.match_ld1(w)
        assert c1 == c2 and np.array_equal(m1, m2), w


def _check_queries(e, o, queries):
    n = 0
    for q in queries:
        st = _norm(_norm(q.strip(), lower=True))        # the search text the pipeline hands to WordMatcherLookup
        got = e.wordmatcher_device(st)
        if got is None:
            continue
        assert np.array_equal(got, o.wordmatcher(st)), q
        assert np.array_equal(got, e.wordmatcher(st)), q
        n += 1
    return n


def test_kat_corpus_and_the_q13_quirk():
    from __graft_entry__ import TEN_DOCS
    e, o = _pair(TEN_DOCS)
    back = []
    _check_words(e, o, ["qick", "battamam", "speding", "glitters", "gothm", "jurney", "thousnd", "abcd", "zzzzzzzz", "fox", "a" * 70], back)
    assert back == ["a" * 70]
    assert _check_queries(e, o, ["qick fux", "battamam", "new york", "speeding", "the fox", "bat", "man", "quick brown fox jumps", "spider-man", "in", "t"]) >= 10
    # WordMatcher.cs:166-196 (quirk Q13): the affix walk yields one document per distinct word — the LAST that contains it
    e2, o2 = _pair([(0, "batman one"), (1, "batman two"), (2, "batman three")])
    assert o2.wm_lookup("bat", affix=True).tolist() == [2]
    assert e2.wordmatcher_device("bat").tolist() == o2.wordmatcher("bat").tolist() == [2]
    assert e2.wordmatcher_device("man").tolist() == o2.wordmatcher("man").tolist()
    assert e2.wordmatcher_device("batman").tolist() == o2.wordmatcher("batman").tolist() == [0, 1, 2]


def test_school_names():
    names = SK.load_names()
    e, o = _pair([(i, n) for i, n in enumerate(names)], SK.SYNONYMS)
    rng = np.random.default_rng(5)
    words, queries = [], []
    for i in rng.integers(0, len(names), 150):
        toks = [t for t in _norm(names[int(i)], lower=True).split(" ") if t]
        queries.append(" ".join(toks[:3]))
        for t in toks[:3]:
            if len(t) >= 4:
                j = int(rng.integers(1, len(t)))
                words += [t[:j] + t[j + 1:], t[:j] + "x" + t[j:], t[:j] + "q" + t[j + 1:]]
    back = []
    _check_words(e, o, words[:400], back)
    assert len(back) <= len(words[:400]) // 20, back
    assert _check_queries(e, o, queries + ["mateřská škola lázně bělohrad", "bel", "belo", "sciozlí", "zs brno", "gymnázium jana nerudy", "ma", "sk"]) >= 100


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_unicode_corpora(seed):
    docs, queries = U.make(seed)
    e, o = _pair(docs)
    back = []
    words = [w for q in queries for w in _norm(q, lower=True).replace("\n", " ").split(" ") if len(w) >= 1][:200]
    _check_words(e, o, words, back)
    assert _check_queries(e, o, queries) >= len(queries) // 2


@pytest.mark.parametrize("config,docs", [(2, 60000), (3, 40000)])
def test_synthetic_configs_words_and_queries(config, docs):
    s = Synth(config, docs=docs)
    arena, offs = s.docs()
    e = SearchEngine.create_default(device=0); e.index_flat(None, arena, offs, s.field_weights)
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    qa, qo = s.queries(300, qseed=17, fuzz=1.0)
    qs = Synth.texts(qa, qo)
    back = []
    words = sorted({w for q in qs for w in q.split() if len(w) >= 4})
    _check_words(e, o, words[:500], back)
    assert len(back) <= 5, back
    assert _check_queries(e, o, qs[:200] + ["th", "an", "qu"]) >= 200


def test_search_with_device_lookups_equals_search_with_host_lookups(tmp_path):
    """The same batches through an engine whose planning lookups run on the GPU (default) and one that keeps them on the host (INFX_HOST_LOOKUPS=1):
    identical rows, bit for bit; the default engine really used the device (lookup_stats)."""
    script = r'''
import sys, json, numpy as np
from infidex_amd import SearchEngine
from infidex_amd.engine import pack_texts
from tools.synth import Synth
s = Synth(3, docs=50000); arena, offs = s.docs()
e = SearchEngine.create_default(device=0); e.index_flat(None, arena, offs, s.field_weights)
qa, qo = s.queries(600, qseed=91, fuzz=0.6)
texts = Synth.texts(qa, qo) + ["qu", "", "zzzzqq", "the of and", "a b c d e f g h i j k l m n o p q r s t u v w x y z aa bb cc dd ee ff"]
a, o = pack_texts(texts)
k, sc, t, c, f = e.search_packed(a, o, 20)
np.savez(sys.argv[1], k=k, sc=sc.view(np.uint32), t=t, c=c, f=f, stats=np.array(list(e.lookup_stats().values()), np.int64), dev=np.array([int(e.device_lookups())]))
'''
    res = []
    for host in ("0", "1"):
        env = dict(os.environ); env["INFX_HOST_LOOKUPS"] = host; env["INFX_DEVICE_LOOKUPS"] = "1"      # "0": dictionaries uploaded, every lookup on the device; "1": no upload, host
        env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = str(tmp_path / f"lk{host}.npz")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=env, timeout=900)
        res.append(np.load(out))
    dev, host = res
    assert dev["dev"][0] == 1 and host["dev"][0] == 0
    assert dev["stats"][0] > 100 and dev["stats"][2] > 500 and host["stats"][0] == 0 and host["stats"][2] == 0, (dev["stats"], host["stats"])
    assert dev["stats"][1] <= 3                                    # words the LD1 kernel handed back to the host walk
    for key in ("k", "sc", "t", "c", "f"):
        assert np.array_equal(dev[key], host[key]), key
