"""world_size-2 gloo test (CPU) of the collective plumbing of the sharded path: count all-reduce, top-k all-gather and the
all-reduce of disjoint Stage-2 records behave as infidex_amd/sharded.py assumes (the phases themselves need a GPU and are
covered by tests/test_gpu_parity.py::test_sharded_equals_unsharded)."""
import os
import socket
import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infidex_amd.sharded import TorchComm
    c = TorchComm(dist)
    nd, depth = 5, 7
    rng = np.random.default_rng(100 + rank)
    # Exchange 1: histograms add up
    counts = rng.integers(0, 1000, (nd, 136)).astype(np.uint32)
    g = c.allreduce_sum_i32(counts)
    # Exchange 2: per-shard top-k (doc, score bits) gathered in rank order
    hits = np.zeros((nd, depth, 2), np.int32)
    hits[..., 0] = rng.integers(rank * 1000, (rank + 1) * 1000, (nd, depth))
    hits[..., 1] = rng.random((nd, depth)).astype(np.float32).view(np.int32)
    ah = c.allgather(hits)
    hc = np.full(nd, depth - rank, np.uint32)
    ahc = c.allgather(hc)
    # disjoint records: every candidate is owned by exactly one rank
    ncand = 11
    outs = np.zeros((ncand, 3), np.int32)
    own = np.arange(ncand) % world == rank
    outs[own] = rng.integers(-2**31, 2**31 - 1, (int(own.sum()), 3), dtype=np.int64).astype(np.int32)
    m = c.allreduce_sum_i32(outs)
    q.put((rank, counts, g, hits, ah, ahc, outs, m))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_collectives():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in ps:
        p.join(60)
    (r0, c0, g0, h0, ah0, ahc0, o0, m0), (r1, c1, g1, h1, ah1, ahc1, o1, m1) = got
    assert np.array_equal(g0, c0 + c1) and np.array_equal(g1, g0)
    assert np.array_equal(ah0[0], h0) and np.array_equal(ah0[1], h1) and np.array_equal(ah1, ah0)
    assert ahc0.tolist() == [[7] * 5, [6] * 5]
    # merged = own record from whichever rank owned it, bit-exact
    exp = np.where((np.arange(11) % 2 == 0)[:, None], o0, o1)
    assert np.array_equal(m0, exp) and np.array_equal(m1, exp)
    # the merge rule of phase 3: best `depth` of the union in (score desc, doc asc) order
    sc = ah0[..., 1].view(np.float32)
    for j in range(5):
        pairs = [(float(sc[w, j, k]), int(ah0[w, j, k, 0])) for w in range(2) for k in range(int(ahc0[w, j]))]
        top = sorted(pairs, key=lambda x: (-x[0], x[1]))[:7]
        assert len(top) == 7 and all(top[i][0] >= top[i + 1][0] for i in range(6))


def _prefetch_worker(rank, world, port, q):
    """Sharded planning exchange on host-only engines (no GPU needed for the host lookups): each rank computes the LD1 expansions and the
    WordMatcher descriptors of its half of the batch, the ranks all-gather the blobs on the planning group and import each other's."""
    import ctypes as C
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infidex_amd import SearchEngine
    from infidex_amd.engine import _p
    from infidex_amd.sharded import TorchComm
    from tools.synth import Synth
    s = Synth(4, docs=30000); arena, offs = s.docs()
    e = SearchEngine.create_default(device=-1); e.index_flat(None, arena, offs, s.field_weights)
    from infidex_amd.engine import pack_texts
    # blank, short ("unsupported"), beyond the Stage-2 envelope (too long / too many distinct words), outside the BMP: every kind of plan record crosses
    # (600 characters / 40 distinct words: the long coverage record; 2100 characters / 140 words: beyond that, an error status crosses)
    extras = ["", "   ", "qu", "x" * 600, " ".join("w%dq" % i for i in range(40)), "y" * 2100, " ".join("v%dq" % i for i in range(140)), "caf\u00e9 \U0001F50D na\u00efve", "ab cd ef"]
    qa, qo = s.queries(200, qseed=5, fuzz=0.6)
    qa, qo = pack_texts(Synth.texts(qa, qo) + extras)
    nq = len(qo) - 1
    sess = C.c_void_p(); assert e.L.infx_engine_default_session(e.h, C.byref(sess)) == 0
    L = e.L; L.infx_session_prefetch_collect.restype = C.c_int64; L.infx_session_prefetch_pending.restype = C.c_int64

    def collect(b, en):
        n = L.infx_session_prefetch_collect(sess, nq, _p(qa, C.c_uint16), _p(qo, C.c_uint64), b, en, 500); assert n >= 8
        blob = np.zeros(n, np.uint8); assert L.infx_session_prefetch_blob(sess, _p(blob, C.c_uint8), C.c_int64(n)) == 0
        return blob
    def digest():
        out = np.zeros(nq, np.uint64); used = C.c_uint32(0)
        assert L.infx_session_plan_digest(sess, nq, _p(qa, C.c_uint16), _p(qo, C.c_uint64), 500, _p(out, C.c_uint64), C.byref(used)) == 0
        return out, int(used.value)
    d_self, used_self = digest()                                   # every query planned here
    c = TorchComm(dist); g = c.planning_group()
    begin, end = nq * rank // world, nq * (rank + 1) // world
    mine = collect(begin, end)
    blobs = c.allgather_bytes(mine, group=g)
    assert np.array_equal(blobs[rank], mine)
    for r, b in enumerate(blobs):
        if r != rank:
            assert L.infx_session_prefetch_import(sess, _p(np.ascontiguousarray(b), C.c_uint8), C.c_int64(b.size)) == 0
    pending = int(L.infx_session_prefetch_pending(sess))
    d_x, used_x = digest()                                         # own slice from the collect, the peer's slice from its blob
    # another batch (other texts, same size): the pending entries are not believed
    qa2, qo2 = s.queries(200, qseed=6, fuzz=0.6)
    qa2, qo2 = pack_texts(Synth.texts(qa2, qo2) + extras)
    out2 = np.zeros(nq, np.uint64); used2 = C.c_uint32(7)
    assert L.infx_session_plan_digest(sess, nq, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), 500, _p(out2, C.c_uint64), C.byref(used2)) == 0
    same_text = sum(1 for i in range(nq) if np.array_equal(qa[qo[i]:qo[i + 1]], qa2[qo2[i]:qo2[i + 1]]))
    # a truncated blob is rejected, not half-imported silently; so is one with a flipped bit in the plan section (checksum)
    bad = np.ascontiguousarray(blobs[1 - rank][: max(9, blobs[1 - rank].size // 2)])
    rc_bad = L.infx_session_prefetch_import(sess, _p(bad, C.c_uint8), C.c_int64(bad.size))
    flip = np.ascontiguousarray(blobs[1 - rank]).copy(); flip[flip.size - 40] ^= 4
    rc_flip = L.infx_session_prefetch_import(sess, _p(flip, C.c_uint8), C.c_int64(flip.size))
    d_after, used_after = digest()                                 # the refused imports left the entries alone
    assert np.array_equal(d_self, d_x) and np.array_equal(d_self, d_after), "plans through the exchange differ from the plans made here"
    assert used_self == 0 and used_x == nq and used_after == nq and int(used2.value) == same_text and rc_flip != 0, (used_self, used_x, used_after, int(used2.value), same_text, rc_flip)
    # the peer's slice, recomputed here after the import (LD1 expansions now come from the fuzzy cache): byte-identical to what the peer sent
    ob, oe = nq * (1 - rank) // world, nq * (2 - rank) // world
    again = collect(ob, oe)
    q.put((rank, mine.tobytes(), blobs[1 - rank].tobytes(), again.tobytes(), pending, rc_bad))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_planning_exchange_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_prefetch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in ps:
        p.join(60)
    (_, mine0, peer0, again0, pend0, bad0), (_, mine1, peer1, again1, pend1, bad1) = got
    assert peer0 == mine1 and peer1 == mine0                      # variable-length byte all-gather on the planning group
    assert again0 == mine1 and again1 == mine0                    # same index + same text => same lookups, whoever computes them
    assert pend0 > 0 and pend1 > 0                                # WordMatcher descriptor sets of the peer's queries wait for phase 0
    assert bad0 != 0 and bad1 != 0
    assert len(mine0) > 1000 and len(mine1) > 1000


class _FakeSession:
    """Stands in for a ShardSession in the chained replay: continues the state as rank r would (state[q] = state[q] * 10 + r + 1 for needed q)."""

    def __init__(self, rank):
        self.rank = rank

    def phase2d(self, need, state):
        for q in range(len(need)):
            if need[q]:
                state[q, 0] = state[q, 0] * 10 + self.rank + 1


def _distx_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infidex_amd.sharded import TorchComm, _DistX, _HostBufs
    X = _DistX(TorchComm(dist), _HostBufs())
    hits = np.full((3, 4, 2), rank + 1, np.int32)
    g = X.allgather([hits])[0]
    nxt = np.asarray([0.5 + rank, 0.0, 2.0], np.float32)
    gn = X.allgather([nxt])[0]
    m = X.max_int([100 + 37 * rank])
    blob = np.full(16, 7 + rank, np.uint8)
    gb = X.allgather([blob])[0]
    cnt = np.full((2, 136), rank + 1, np.int32)
    gc = X.allreduce_sum([cnt])[0]
    uc = np.asarray([3 + rank, 5], np.uint32)
    guc = X.allreduce_sum([uc])[0]
    # the sequential chain: rank 0's step, then rank 1's, on the queries that need it; every rank ends with the last shard's state
    state = X.chain([_FakeSession(rank)], np.asarray([1, 0, 1], np.uint32), np.zeros((3, 6), np.uint32))
    # the infx_comm the C++ driver calls back into (host buffers, raw pointers): in-place sum, rank-ordered gather
    import ctypes as C
    import types
    from infidex_amd.sharded import native_comm
    cc, keep = native_comm(types.SimpleNamespace(L=None), TorchComm(dist))
    a = np.asarray([1 + rank, 0xFFFFFFF0, 7], np.uint32); send = np.full(5, 10 + rank, np.uint8); recv = np.zeros(5 * world, np.uint8)
    rc1 = cc.allreduce_sum_u32(cc.ctx, a.ctypes.data_as(C.c_void_p), 3, None)
    rc2 = cc.allgather(cc.ctx, send.ctypes.data_as(C.c_void_p), recv.ctypes.data_as(C.c_void_p), 5, None)
    q.put((rank, g, gn, m, gb, gc, guc, state, (rc1, rc2, a, recv, cc.rank, cc.nranks, cc.device_buffers)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_object_of_the_sharded_driver_world2():
    """The collectives _run_batch issues (infidex_amd/sharded.py), through the real-rank exchange object on gloo: shapes, rank order, the padded-size
    maximum, and the order of the chained sequential replay."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_distx_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in ps:
        p.join(60)
    for rank, g, gn, m, gb, gc, guc, state, nat in got:
        rc1, rc2, a, recv, crank, cworld, cdev = nat
        assert rc1 == 0 and rc2 == 0 and crank == rank and cworld == 2 and cdev == 0
        assert a.tolist() == [3, 0xFFFFFFE0, 14]                 # uint32 sums wrap like the int32 sums the transport computes
        assert recv.tolist() == [10] * 5 + [11] * 5
        assert g.shape == (2, 3, 4, 2) and (g[0] == 1).all() and (g[1] == 2).all()
        assert gn.dtype == np.float32 and gn[:, 0].tolist() == [0.5, 1.5]
        assert m == 137
        assert gb.shape == (2, 16) and gb[0, 0] == 7 and gb[1, 0] == 8
        assert (gc == 3).all() and gc.dtype == np.int32
        assert guc.tolist() == [7, 10] and guc.dtype == np.uint32
        assert state[:, 0].tolist() == [12, 0, 12]            # rank 0 first, then rank 1


def test_shards_are_whole_containers():
    """Shard boundaries are multiples of 65 536 documents (whole Roaring containers, Bm25Scorer.cs:195-280 chunks never straddle shards); shards
    cover the corpus; more shards than containers leaves the last ones empty."""
    from infidex_amd.sharded import create_sharded_engine
    from tools.synth import Synth
    s = Synth(2, docs=150000); arena, offs = s.docs()
    for W in (1, 2, 3, 5):
        spans = []
        for r in range(W):
            e = create_sharded_engine(r, W, -1); e.index_flat(None, arena, offs, s.field_weights)
            spans.append(e.shard_info())
        assert spans[0][0] == 0 and sum(n for _, n in spans) == 150000
        for (b, n), (b2, _) in zip(spans, spans[1:]):
            assert b + n == b2
        assert all(b % 65536 == 0 or b == 150000 for b, _ in spans), spans
        if W == 5:
            assert sum(1 for _, n in spans if n == 0) == 2


def test_plan_exchange_edge_batches_in_one_process():
    """The plan exchange between two host-only engines of one process (no transport): empty batches, single queries, empty slices, a slice that is the whole
    batch — the importer's digests equal the ones it computes itself, and every exchanged query is counted."""
    import ctypes as C
    from infidex_amd import SearchEngine
    from infidex_amd.engine import _p, pack_texts
    from tools.synth import Synth
    s = Synth(4, docs=5000); arena, offs = s.docs()
    a = SearchEngine.create_default(device=-1); a.index_flat(None, arena, offs, s.field_weights)
    b = SearchEngine.create_default(device=-1); b.index_flat(None, arena, offs, s.field_weights)
    c = SearchEngine.create_default(device=-1); c.index_flat(None, arena, offs, s.field_weights)      # never sees an exchange: the reference digests
    L = a.L; L.infx_session_prefetch_collect.restype = C.c_int64
    sa = C.c_void_p(); sb = C.c_void_p(); sc = C.c_void_p()
    assert L.infx_engine_default_session(a.h, C.byref(sa)) == 0 and L.infx_engine_default_session(b.h, C.byref(sb)) == 0 and L.infx_engine_default_session(c.h, C.byref(sc)) == 0
    qa, qo = s.queries(6, qseed=2, fuzz=0.5)
    texts = Synth.texts(qa, qo)
    for batch, (begin, end) in [([], (0, 0)), (texts[:1], (0, 1)), (texts[:1], (0, 0)), (texts[:1], (1, 1)), (texts, (0, 6)), (texts, (2, 5)), (texts, (6, 6)), (texts[:2] + [""], (2, 3))]:
        qa2, qo2 = pack_texts(batch); nq = len(batch)
        ref = np.zeros(max(nq, 1), np.uint64); used = C.c_uint32(9)
        assert L.infx_session_plan_digest(sc, nq, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), 500, _p(ref, C.c_uint64), C.byref(used)) == 0 and used.value == 0
        n = L.infx_session_prefetch_collect(sa, nq, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), begin, end, 500); assert n > 0, (batch, begin, end)
        blob = np.zeros(n, np.uint8); assert L.infx_session_prefetch_blob(sa, _p(blob, C.c_uint8), C.c_int64(n)) == 0
        assert L.infx_session_prefetch_collect(sb, nq, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), 0, 0, 500) > 0      # the importer's own (empty) slice: a rank collects before it imports
        assert L.infx_session_prefetch_import(sb, _p(blob, C.c_uint8), C.c_int64(n)) == 0, a.L.infx_engine_last_error()
        got = np.zeros(max(nq, 1), np.uint64)
        assert L.infx_session_plan_digest(sb, nq, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), 500, _p(got, C.c_uint64), C.byref(used)) == 0
        assert used.value == end - begin and np.array_equal(got[:nq], ref[:nq]), (batch, begin, end, used.value)
        # another depth than the plans were made for: nothing of the exchange is used
        assert L.infx_session_plan_digest(sb, nq, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), 200, _p(got, C.c_uint64), C.byref(used)) == 0 and used.value == 0
    # a slice outside the batch is refused by the collector
    qa2, qo2 = pack_texts(texts)
    assert L.infx_session_prefetch_collect(sa, 6, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), 4, 7, 500) < 0
    assert L.infx_session_prefetch_collect(sa, 6, _p(qa2, C.c_uint16), _p(qo2, C.c_uint64), 5, 4, 500) < 0


def _blob_hash(b: bytes) -> int:      # bytes_hash of csrc/host/engine.cpp (the plan section's checksum), restated
    M = (1 << 64) - 1
    h = (0x9E3779B97F4A7C15 ^ len(b)) & M
    i = 0
    while i + 8 <= len(b):
        h = ((h ^ int.from_bytes(b[i:i + 8], "little")) * 0xFF51AFD7ED558CCD) & M; h ^= h >> 32; i += 8
    for c in b[i:]:
        h = ((h ^ c) * 1099511628211) & M
    return h ^ (h >> 29)


def _plan_section_start(blob: bytes) -> int:      # skips the LD1 and WordMatcher sections (layout: infx_session_prefetch_collect)
    import struct
    o = 8
    (n,) = struct.unpack_from("<I", blob, o); o += 4
    for _ in range(n):
        (wl,) = struct.unpack_from("<H", blob, o); o += 2 + 2 * wl
        (nm,) = struct.unpack_from("<I", blob, o); o += 4 + 4 * nm
    (n,) = struct.unpack_from("<I", blob, o); o += 4
    for _ in range(n):
        (tl,) = struct.unpack_from("<H", blob, o); o += 2 + 2 * tl
        (nl,) = struct.unpack_from("<I", blob, o); o += 4 + 16 * nl
        (no,) = struct.unpack_from("<I", blob, o); o += 4 + 4 * no
    return o


def test_damaged_plan_records_are_refused_not_believed():
    """The plan section is checksummed (a damaged blob is refused at import); behind the checksum every record is still range-checked where it is parsed.
    800 single-byte corruptions of the plan section WITH a recomputed checksum: import or the parse (through infx_session_plan_digest) reports an error or the
    record happens to stay well-formed — nothing crashes, and a clean blob is accepted afterwards."""
    import ctypes as C
    import random
    from infidex_amd import SearchEngine
    from infidex_amd.engine import _p, pack_texts
    from tools.synth import Synth
    s = Synth(4, docs=5000); arena, offs = s.docs()
    a = SearchEngine.create_default(device=-1); a.index_flat(None, arena, offs, s.field_weights)
    b = SearchEngine.create_default(device=-1); b.index_flat(None, arena, offs, s.field_weights)
    L = a.L; L.infx_session_prefetch_collect.restype = C.c_int64
    sa = C.c_void_p(); sb = C.c_void_p()
    assert L.infx_engine_default_session(a.h, C.byref(sa)) == 0 and L.infx_engine_default_session(b.h, C.byref(sb)) == 0
    qa, qo = s.queries(40, qseed=4, fuzz=0.5)
    qa, qo = pack_texts(Synth.texts(qa, qo) + ["", "qu", " ".join("w%dq" % i for i in range(40)), "x" * 600]); nq = len(qo) - 1
    n = L.infx_session_prefetch_collect(sa, nq, _p(qa, C.c_uint16), _p(qo, C.c_uint64), 0, nq, 500); assert n > 0
    blob = np.zeros(n, np.uint8); assert L.infx_session_prefetch_blob(sa, _p(blob, C.c_uint8), C.c_int64(n)) == 0
    raw = blob.tobytes(); start = _plan_section_start(raw)
    assert _blob_hash(raw[start:-8]) == int.from_bytes(raw[-8:], "little")      # the restated checksum is the library's
    rng = random.Random(1); refused_import = refused_parse = accepted = 0
    out = np.zeros(nq, np.uint64); used = C.c_uint32(0)
    for _ in range(800):
        m = bytearray(raw); pos = rng.randrange(start, len(raw) - 8); m[pos] ^= 1 << rng.randrange(8)
        m[-8:] = _blob_hash(bytes(m[start:-8])).to_bytes(8, "little")
        mb = np.frombuffer(bytes(m), np.uint8)
        L.infx_session_prefetch_collect(sb, nq, _p(qa, C.c_uint16), _p(qo, C.c_uint64), 0, 0, 500)      # the importer's own (empty) slice: starts every round clean
        if L.infx_session_prefetch_import(sb, _p(mb, C.c_uint8), C.c_int64(mb.size)) != 0:
            refused_import += 1
            continue
        if L.infx_session_plan_digest(sb, nq, _p(qa, C.c_uint16), _p(qo, C.c_uint64), 500, _p(out, C.c_uint64), C.byref(used)) != 0:
            refused_parse += 1
        else:
            accepted += 1
    assert refused_import + refused_parse + accepted == 800 and refused_import + refused_parse > 0
    # without the recomputed checksum every one of them is refused at import
    m = bytearray(raw); m[start + 40] ^= 1
    mb = np.frombuffer(bytes(m), np.uint8); assert L.infx_session_prefetch_import(sb, _p(mb, C.c_uint8), C.c_int64(mb.size)) != 0
    L.infx_session_prefetch_collect(sb, nq, _p(qa, C.c_uint16), _p(qo, C.c_uint64), 0, 0, 500)
    assert L.infx_session_prefetch_import(sb, _p(blob, C.c_uint8), C.c_int64(n)) == 0
    assert L.infx_session_plan_digest(sb, nq, _p(qa, C.c_uint16), _p(qo, C.c_uint64), 500, _p(out, C.c_uint64), C.byref(used)) == 0 and used.value == nq
    print("damaged plan records:", refused_import, "refused at import,", refused_parse, "at the parse,", accepted, "still well-formed")
