"""Eight document shards — the world size the driver's scaling run ends at — against the ORACLE, on one GPU (eight engines of one process, the phases in lock
step, exchange buffers as device tensors: the RCCL code path minus the collectives).  No multi-GPU box is reachable from here, so this is the widest the sharded
machinery has been exercised against the reference algorithm: config-4 shape (fuzzy 2-3 word queries, exact Stage-1 replay across the shards: first-pass lists
of eight ranks, the global ambiguity test, chunk chains that cross seven shard boundaries, owner-side heaps) and config-5 shape (Infiscript filter + facets over
the merged rows, NumberOfDocumentsInFilter summed over eight device counts).  600 k documents = 10 containers of 65 536 ids: every rank owns at least one."""
import os

import numpy as np
import pytest

from infidex_amd.engine import pack_texts
from tests import oracle_lib as O
from tools.synth import Synth

pytestmark = pytest.mark.gpu

N_DOCS = 600_000
W = 8
K = 20


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    from infidex_amd.sharded import create_sharded_engine, ShardSession
    s = Synth(4, docs=N_DOCS); arena, offs = s.docs()
    engs = [create_sharded_engine(r, W, 0) for r in range(W)]
    # one host-index build for the "node": rank 0 indexes and saves, the other seven read the arrays back and upload their own shard
    cache = str(tmp_path_factory.mktemp("hostcache"))
    engs[0].set_build_threads(8); engs[0].index_flat(None, arena, offs, s.field_weights)
    path = os.path.join(cache, "host_index.bin"); engs[0].save_host_index(path)
    for g in engs[1:]:
        g.index_from_host_cache(path)
    os.remove(path)
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    return s, engs, [ShardSession(g) for g in engs], o


def test_every_rank_owns_whole_containers(world):
    s, engs, sess, o = world
    info = [g.shard_info() for g in engs]                      # (first document, documents)
    assert all(b % 65536 == 0 and n > 0 for b, n in info), info
    assert sum(n for _, n in info) == N_DOCS and [b for b, _ in info] == sorted(b for b, _ in info)


def test_eight_shards_equal_the_oracle_config4(world):
    from infidex_amd.sharded import simulate_shards_dev
    from tests.parity_classify import assert_final_rows_match_oracle
    s, engs, sess, o = world
    qa, qo = s.queries(400, qseed=808)
    texts = Synth.texts(qa, qo) + ["qu", "", "zzzzqq", "the of and"]
    a, off = pack_texts(texts)
    res = simulate_shards_dev(sess, a, off, K)
    for r in res[1:]:
        for x, y in zip(r, res[0]):
            assert np.array_equal(x, y)                           # all eight ranks hold the same rows
    keys, scores, ties, counts, flags = res[0]
    sample = list(range(0, len(texts), 2))
    same, flips = assert_final_rows_match_oracle(keys[sample], scores[sample], counts[sample], o, [texts[i] for i in sample], K, what="8 shards at 600k docs")
    replays = sum(x.s.last_timings()["exact_replays"] for x in sess)
    print("8 shards vs oracle:", same, "identical order,", flips, "near-tie flips of", len(sample), "; queries replayed on their owners:", replays)
    assert replays > 0


def test_eight_shards_equal_the_oracle_config5(world):
    from infidex_amd.sharded import simulate_shards_dev, simulate_set_filter
    from tools.synth import config5_columns
    s, engs, sess, o = world
    year, rating, genre = config5_columns(N_DOCS)
    for x in engs + [o]:
        x.set_column("year", year, facetable=True); x.set_column("rating", rating, facetable=False); x.set_column("genre", genre, facetable=True)
    qa, qo = s.queries(60, qseed=55)
    texts = Synth.texts(qa, qo); a, off = pack_texts(texts)
    try:
        for flt in ["year >= 2000 AND rating > 7.0", None]:
            nin = simulate_set_filter(sess, flt, True)
            res = simulate_shards_dev(sess, a, off, K)
            for r in res[1:]:
                for x, y in zip(r, res[0]):
                    assert np.array_equal(x, y)
            keys, scores, ties, counts, flags = res[0]
            for i, q in enumerate(texts):
                w = o.search_filtered(q, K, filter=flt, enable_facets=True)
                assert keys[i, :int(counts[i])].tolist() == w["keys"], (flt, q)
                assert nin == w["in_filter"], (flt, nin, w["in_filter"])
                for ss in sess:
                    assert (ss.facets(i) or {}) == w["facets"], (flt, q)
    finally:
        simulate_set_filter(sess, None, False)
