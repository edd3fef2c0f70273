"""The C++ driver of the sharded phases (infx_session_sharded_finish) with DEVICE exchange buffers and more than one shard — the combination the RCCL
deployment runs — without RCCL: two document shards on one GPU in one process, one thread per shard, and an infx_comm (include/infidex_engine.h) whose
all-reduce / all-gather callbacks rendezvous through a threading.Barrier and move the device buffers with hipMemcpy.  (RCCL itself cannot put two ranks
on one GPU; with one rank it is covered by `INFX_FORCE_SHARDED=1 python bench.py`, and the two-process test uses host buffers over gloo.)  The rows must
be those of the phase-by-phase simulation on the same shards, which the other sharded tests compare with the oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

from tools.synth import Synth

pytestmark = pytest.mark.gpu


def test_native_driver_with_device_buffers_on_two_shards():
    from infidex_amd.sharded import create_sharded_engine, ShardSession, simulate_shards, _CComm, _ALLREDUCE, _ALLGATHER
    from infidex_amd.engine import pack_texts, _p
    hip = C.CDLL("libamdhip64.so")
    H2D, D2H = 1, 2
    W = 2
    s = Synth(2, docs=140000)                                 # three 65 536-id containers: shard 0 holds one, shard 1 two
    arena, offs = s.docs()
    engs = [create_sharded_engine(r, W, 0) for r in range(W)]
    for e in engs:
        e.index_flat(None, arena, offs, s.field_weights)
    sess = [ShardSession(e) for e in engs]
    qa, qo = s.queries(160, qseed=47, fuzz=0.3)
    qs = Synth.texts(qa, qo) + ["qu", "", "zzzzqq"]
    a2, o2 = pack_texts(qs)
    expected = simulate_shards(sess, a2, o2, 10)

    barrier = threading.Barrier(W, timeout=60)
    slots = [None] * W

    def exchange(rank, host):                                 # every rank deposits its array and gets all of them, in rank order
        slots[rank] = host
        barrier.wait()
        got = [slots[r] for r in range(W)]
        barrier.wait()                                        # everyone has read the slots before the next collective overwrites them
        return got

    def make_comm(rank):
        def _ar(ctx, buf, count, stream):
            try:
                n = int(count)
                if n:
                    assert hip.hipStreamSynchronize(C.c_void_p(stream)) == 0
                    h = np.empty(n, np.uint32)
                    assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(buf), C.c_size_t(n * 4), D2H) == 0
                    parts = exchange(rank, h)
                    tot = parts[0].copy()
                    for p in parts[1:]:
                        tot += p                               # uint32: wraps like the device sum
                    assert hip.hipMemcpy(C.c_void_p(buf), tot.ctypes.data_as(C.c_void_p), C.c_size_t(n * 4), H2D) == 0
                return 0
            except Exception:                                  # never let an exception cross the C frames
                barrier.abort()
                return 3

        def _ag(ctx, send, recv, nbytes, stream):
            try:
                n = int(nbytes)
                if n:
                    assert hip.hipStreamSynchronize(C.c_void_p(stream)) == 0
                    h = np.empty(n, np.uint8)
                    assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(send), C.c_size_t(n), D2H) == 0
                    allb = np.ascontiguousarray(np.concatenate(exchange(rank, h)))
                    assert hip.hipMemcpy(C.c_void_p(recv), allb.ctypes.data_as(C.c_void_p), C.c_size_t(n * W), H2D) == 0
                return 0
            except Exception:
                barrier.abort()
                return 3
        far, fag = _ALLREDUCE(_ar), _ALLGATHER(_ag)
        cc = _CComm()
        cc.ctx = C.c_void_p(1); cc.rank = rank; cc.nranks = W; cc.device_buffers = 1
        cc.allreduce_sum_u32 = far; cc.allgather = fag
        return cc, (far, fag)

    comms = [make_comm(r) for r in range(W)]
    out = [None] * W
    errs = []

    def rank_main(r):
        try:
            ss = sess[r]
            ss.phase0(a2, o2, 500)
            nq, mr = ss.nq, 10
            ss.max_results = mr
            keys = np.full((nq, mr), -1, np.int64); scores = np.zeros((nq, mr), np.float32)
            ties = np.zeros((nq, mr), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
            ss.e._check(ss.L.infx_session_sharded_finish(ss.s.h, C.byref(comms[r][0]), mr, 1, _p(keys, C.c_int64), _p(scores, C.c_float),
                                                         _p(ties, C.c_uint8), _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
            out[r] = (keys, scores, ties, counts, flags)
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
            barrier.abort()

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(120)
    assert not errs, errs
    for r in range(W):
        for got, exp in zip(out[r], expected[r]):
            assert np.array_equal(got, exp)
    replays = sum(x.s.last_timings()["exact_replays"] for x in sess)
    assert replays > 0                                        # ambiguous cuts were replayed across the shards through the device buffers


def test_three_sessions_in_flight_issue_their_collectives_in_one_order_on_every_rank():
    """Two shards x three pipeline sessions each, six batches in flight on session i mod 3, every thread delayed at random — and ONE exchange channel per
    rank pair, shared by the three sessions, that only works when both ranks issue the collectives of their sessions in the same global order (each deposit
    carries (session, collective number) and the ranks must present equal tags).  That order is what the collective ring (infx_engine_coll_ring, CollSeq in
    csrc/host/engine.cpp) guarantees RCCL, where a different order of different communicators' kernels on two ranks can hang the job."""
    import random
    import time
    from infidex_amd.sharded import create_sharded_engine, ShardSession, simulate_shards, _CComm, _ALLREDUCE, _ALLGATHER
    from infidex_amd.engine import pack_texts, _p
    hip = C.CDLL("libamdhip64.so")
    H2D, D2H = 1, 2
    W, K, NB = 2, 3, 6
    s = Synth(2, docs=140000)
    arena, offs = s.docs()
    engs = [create_sharded_engine(r, W, 0) for r in range(W)]
    for e in engs:
        e.index_flat(None, arena, offs, s.field_weights)
    sess = [[ShardSession(e) for _ in range(K)] for e in engs]
    batches = []
    for b in range(NB):
        qa, qo = s.queries(60 + 10 * b, qseed=300 + b, fuzz=0.4)
        batches.append(pack_texts(Synth.texts(qa, qo)))
    expected = [simulate_shards([sess[r][0] for r in range(W)], a, o, 10) for a, o in batches]

    barrier = threading.Barrier(W, timeout=90)
    slots = [None] * W
    mismatches = []

    def exchange(rank, tag, host):
        slots[rank] = (tag, host)
        barrier.wait()
        got = [slots[r] for r in range(W)]
        if any(g[0] != tag for g in got):
            mismatches.append((rank, tag, [g[0] for g in got]))
        barrier.wait()
        return [g[1] for g in got]

    def make_comm(rank, k):
        n_issued = [0]

        def tag():
            n_issued[0] += 1
            return (k, n_issued[0])

        def _ar(ctx, buf, count, stream):
            try:
                n = int(count)
                if n:
                    assert hip.hipStreamSynchronize(C.c_void_p(stream)) == 0
                    h = np.empty(n, np.uint32)
                    assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(buf), C.c_size_t(n * 4), D2H) == 0
                    parts = exchange(rank, tag(), h)
                    if any(p.size != n for p in parts):
                        raise RuntimeError("size mismatch")
                    tot = parts[0].copy()
                    for p in parts[1:]:
                        tot += p
                    assert hip.hipMemcpy(C.c_void_p(buf), tot.ctypes.data_as(C.c_void_p), C.c_size_t(n * 4), H2D) == 0
                return 0
            except Exception:
                barrier.abort()
                return 3

        def _ag(ctx, send, recv, nbytes, stream):
            try:
                n = int(nbytes)
                if n:
                    assert hip.hipStreamSynchronize(C.c_void_p(stream)) == 0
                    h = np.empty(n, np.uint8)
                    assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(send), C.c_size_t(n), D2H) == 0
                    parts = exchange(rank, tag(), h)
                    if any(p.size != n for p in parts):
                        raise RuntimeError("size mismatch")
                    allb = np.ascontiguousarray(np.concatenate(parts))
                    assert hip.hipMemcpy(C.c_void_p(recv), allb.ctypes.data_as(C.c_void_p), C.c_size_t(n * W), H2D) == 0
                return 0
            except Exception:
                barrier.abort()
                return 3
        far, fag = _ALLREDUCE(_ar), _ALLGATHER(_ag)
        cc = _CComm()
        cc.ctx = C.c_void_p(1); cc.rank = rank; cc.nranks = W; cc.device_buffers = 1
        cc.allreduce_sum_u32 = far; cc.allgather = fag
        return cc, (far, fag)

    comms = [[make_comm(r, k) for k in range(K)] for r in range(W)]
    for r in range(W):
        hs = (C.c_void_p * K)(*[sess[r][k].s.h for k in range(K)])
        engs[r]._check(engs[r].L.infx_engine_coll_ring(engs[r].h, K, hs))
    out = [[None] * NB for _ in range(W)]
    errs = []

    def session_main(r, k):
        rng = random.Random(1000 * r + k)
        ss = sess[r][k]
        try:
            for b in range(k, NB, K):
                time.sleep(rng.random() * 0.05)                 # scramble which session reaches its collectives first on this rank
                a, o = batches[b]
                ss.phase0(a, o, 500)
                time.sleep(rng.random() * 0.03)
                nq, mr = ss.nq, 10
                ss.max_results = mr
                keys = np.full((nq, mr), -1, np.int64); scores = np.zeros((nq, mr), np.float32)
                ties = np.zeros((nq, mr), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
                ss.e._check(ss.L.infx_session_sharded_finish(ss.s.h, C.byref(comms[r][k][0]), mr, 1, _p(keys, C.c_int64), _p(scores, C.c_float),
                                                             _p(ties, C.c_uint8), _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
                out[r][b] = (keys, scores, ties, counts, flags)
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
            barrier.abort()
        finally:
            ss.L.infx_session_coll_retire(ss.s.h)

    ths = [threading.Thread(target=session_main, args=(r, k)) for r in range(W) for k in range(K)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not errs, errs
    assert not mismatches, mismatches[:3]
    for b in range(NB):
        for r in range(W):
            for got, exp in zip(out[r][b], expected[b][r]):
                assert np.array_equal(got, exp), (b, r)
    for r in range(W):                                           # every session issued collectives, and both ranks issued the same number
        st = np.zeros((K, 4), np.uint64)
        for k in range(K):
            sess[r][k].L.infx_session_coll_stats(sess[r][k].s.h, _p(st[k], C.c_uint64))
        assert (st[:, 0] > 0).all() and (st[:, 1] > 0).all()
        if r == 0:
            ref = st.copy()
        assert np.array_equal(st[:, :2], ref[:, :2])


def test_sharded_search_rejects_a_depth_below_max_depth():
    """the chained sequential replay exchanges heaps laid out for max_depth entries; a sharded batch with a smaller depth used to work until
    one query needed the chain and then failed on every rank.  It is refused up front."""
    from infidex_amd.sharded import create_sharded_engine, ShardSession
    from infidex_amd.engine import pack_texts, InfidexError
    s = Synth(2, docs=70000)
    arena, offs = s.docs()
    e = create_sharded_engine(0, 2, 0)
    e.index_flat(None, arena, offs, s.field_weights)
    ss = ShardSession(e)
    a, o = pack_texts(["alpha beta"])
    with pytest.raises(InfidexError):
        ss.phase0(a, o, 100)
    ss.phase0(a, o, 500)
