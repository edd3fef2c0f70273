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
