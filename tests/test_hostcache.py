"""One host-index build per node (include/infidex_engine.h: infx_engine_save_host_index / _index_from_host_cache; infidex_amd/sharded.py:
index_flat_per_node).  No GPU needed: the host index and query planning are host code (device = -1 engines)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from infidex_amd import SearchEngine
from infidex_amd import engine as E
from tools.synth import Synth


def _queries(s, n=150, seed=5):
    qa, qo = s.queries(n, qseed=seed, fuzz=0.5)
    return Synth.texts(qa, qo) + ["qu", "", "zzzzqq", "the of and"]


def _same_planning(a, b, texts):
    assert a.index_stats() == b.index_stats()
    for t in texts:
        pa, pb = a.plan(t), b.plan(t)
        for k in pa:
            assert np.array_equal(pa[k], pb[k]), (t, k)
        assert np.array_equal(a.wordmatcher(t), b.wordmatcher(t)), t


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    s = Synth(3, docs=8000)
    arena, offs = s.docs()
    keys = np.arange(8000, dtype=np.int64) * 7 + 3           # DocumentKey != internal id: the key map has to come back too
    e = SearchEngine.create_default(device=-1, threads=4)
    e.add_synonym("street", "road")
    e.index_flat(keys, arena, offs, s.field_weights)
    path = str(tmp_path_factory.mktemp("hc") / "host.bin")
    e.save_host_index(path)
    return s, e, path


def test_cached_host_index_plans_like_the_built_one(built):
    s, e, path = built
    f = SearchEngine.create_default(device=-1, threads=2)
    f.add_synonym("street", "road")
    f.index_from_host_cache(path)
    _same_planning(e, f, _queries(s))
    # deletions go through the key map the loader rebuilds from the stored DocumentKeys
    assert e.delete_documents([3, 10, 17]) == f.delete_documents([3, 10, 17]) == 3
    with pytest.raises(E.InfidexError):
        f.index_from_host_cache(path)                        # already indexed


def test_foreign_truncated_or_mismatching_cache_is_refused(built, tmp_path):
    s, e, path = built
    blob = open(path, "rb").read()
    mid = len(blob) // 2
    flipped = blob[:mid] + bytes([blob[mid] ^ 0x10]) + blob[mid + 1:]          # one bit inside an array: the lengths still add up, the checksum does not
    cases = {"truncated": blob[:len(blob) // 2], "foreign": b"INFDX2" + blob[6:], "tail": blob[:-8] + b"\0" * 8, "empty": b"", "flipped": flipped}
    for name, data in cases.items():
        p = str(tmp_path / (name + ".bin"))
        open(p, "wb").write(data)
        f = SearchEngine.create_default(device=-1, threads=2)
        f.add_synonym("street", "road")
        with pytest.raises(E.InfidexError):
            f.index_from_host_cache(p)
    other_cfg = SearchEngine.create_default(device=-1, threads=2)      # no synonym map: the stored texts were canonicalised with one
    with pytest.raises(E.InfidexError) as ei:
        other_cfg.index_from_host_cache(path)
    assert "configuration" in str(ei.value)
    missing = SearchEngine.create_default(device=-1, threads=2)
    with pytest.raises(E.InfidexError):
        missing.index_from_host_cache(str(tmp_path / "nope.bin"))


def test_cache_file_is_created_exclusively(built, tmp_path):
    """A link planted at the writer's temporary path is replaced, not written through; a link at the final path is not read through."""
    s, e, path = built
    victim = tmp_path / "victim.txt"
    victim.write_bytes(b"do not touch")
    target = str(tmp_path / "host.bin")
    os.symlink(str(victim), target + ".tmp")
    e.save_host_index(target)
    assert victim.read_bytes() == b"do not touch"
    assert not os.path.islink(target) and os.path.getsize(target) == os.path.getsize(path)
    assert (os.stat(target).st_mode & 0o777) == 0o600
    link = str(tmp_path / "link.bin")
    os.symlink(target, link)
    f = SearchEngine.create_default(device=-1, threads=2)
    f.add_synonym("street", "road")
    with pytest.raises(E.InfidexError):
        f.index_from_host_cache(link)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank(rank, world, port, cache_dir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infidex_amd.sharded import create_sharded_engine, index_flat_per_node
    s = Synth(2, docs=70000)                                  # two 65 536-id containers: rank 0 owns one, rank 1 the other
    arena, offs = s.docs()
    eng = create_sharded_engine(rank, world, -1, threads=2)
    index_flat_per_node(eng, dist.barrier, rank, 4, None, arena, offs, s.field_weights, tag=str(port), cache_dir=cache_dir)
    texts = _queries(s, 60)
    plans = [eng.plan(t) for t in texts]
    sig = [(p["mode"], p["prefix_set"], p["n_and"], p["term_ids"].tolist(), p["df"].tolist(), p["idf"].view(np.uint32).tolist()) for p in plans]
    dist.barrier()                                            # the leader removes the file after the helper's second barrier: look once it has
    q.put((rank, eng.index_stats(), eng.shard_info(), sig, sorted(os.listdir(cache_dir))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_build(tmp_path):
    """world_size 2 over gloo: rank 0 builds and saves, rank 1 reads the cache; both plan identically, own different shards, the file is gone afterwards."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_rank, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in ps:
        p.join(60)
    (_, st0, sh0, sig0, ls0), (_, st1, sh1, sig1, ls1) = got
    assert st0 == st1 and sig0 == sig1
    assert sh0 == (0, 65536) and sh1 == (65536, 70000 - 65536)      # whole containers: the leader and the follower own different shards
    assert ls0 == [] and ls1 == []                             # removed by the leader after the second barrier

    # and the same planning as an engine that simply indexed the corpus itself
    s = Synth(2, docs=70000)
    arena, offs = s.docs()
    solo = SearchEngine.create_default(device=-1, threads=4)
    solo.index_flat(None, arena, offs, s.field_weights)
    texts = _queries(s, 60)
    sig = [(p["mode"], p["prefix_set"], p["n_and"], p["term_ids"].tolist(), p["df"].tolist(), p["idf"].view(np.uint32).tolist()) for p in (solo.plan(t) for t in texts)]
    assert sig == sig0


def _failing_rank(rank, world, port, cache_dir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infidex_amd.sharded import create_sharded_engine, index_flat_per_node
    s = Synth(2, docs=3000)
    arena, offs = s.docs()
    eng = create_sharded_engine(rank, world, -1, threads=2)
    if rank == 0:
        offs = offs.copy(); offs[7] = offs[9] + 5                       # descending field offsets: the leader's index_flat refuses them
    try:
        index_flat_per_node(eng, dist.barrier, rank, 2, None, arena, offs, s.field_weights, tag=str(port), cache_dir=cache_dir)
        q.put((rank, "ok", sorted(os.listdir(cache_dir))))
    except Exception as e:                  # noqa: BLE001
        q.put((rank, type(e).__name__, sorted(os.listdir(cache_dir))))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_leader_fails_every_rank_instead_of_hanging(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_failing_rank, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in ps:
        p.join(60)
    assert got[0][1] != "ok" and got[1][1] != "ok", got          # both raised, nobody waited for the other
    assert got[0][2] == []                                       # nothing left behind
