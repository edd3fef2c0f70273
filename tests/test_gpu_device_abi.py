"""The device ABI (include/infidex_hip.h) driven directly through ctypes, the way a C# P/Invoke host would: the index arrays come from the ORACLE's
builder, uploaded with infx_upload_*; Stage 1 = infx_stage1_batch with term lists (ids, idf, maxScore, tier roles), Stage 2 = infx_stage2_batch with
the oracle's candidate rows.  Nothing goes through the product's search path (infx_engine_search_batch): a host-only engine (device = -1) supplies
only what the reference's host would compute itself (tier roles of the terms, the prepared coverage query)."""
import ctypes as C

import numpy as np
import pytest

from infidex_amd import SearchEngine
from infidex_amd.engine import load_library, normalize, _u16, _p
from tests import oracle_lib as O
from tools.synth import Synth

pytestmark = pytest.mark.gpu
NCLASS, NFEAT, DEPTH = 136, 32, 500
SCORE_RTOL = 2e-6 * 32


class Cfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("range_docs", C.c_int32), ("max_depth", C.c_int32), ("flags", C.c_int32)]


class Term(C.Structure):
    _fields_ = [("term_id", C.c_int32), ("extra_off", C.c_uint32), ("extra_len", C.c_uint32), ("idf", C.c_float), ("max_score", C.c_float),
                ("role", C.c_uint8), ("rank", C.c_uint8), ("reserved", C.c_uint16)]


class Query(C.Structure):
    _fields_ = [("term_off", C.c_uint32), ("num_terms", C.c_uint32), ("mode", C.c_int32), ("prefix_set", C.c_int32), ("depth", C.c_int32),
                ("n_and", C.c_int32), ("df_s1", C.c_int32), ("df_s2", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("doc", C.c_int32), ("score", C.c_float)]


class CovCand(C.Structure):
    _fields_ = [("query", C.c_uint32), ("doc", C.c_int32), ("base_score", C.c_float), ("want_lcs", C.c_int32)]


class CovOut(C.Structure):
    _fields_ = [("score", C.c_float), ("tiebreaker", C.c_uint8), ("word_hits", C.c_uint8), ("lcs", C.c_uint8), ("status", C.c_uint8), ("word_hits_full", C.c_int32)]


def chk(L, rc):
    assert rc == 0, (rc, L.infx_last_error().decode())


def test_device_abi_with_oracle_built_inputs():
    L = load_library()
    s = Synth(2, docs=20000)
    arena, offs = s.docs()
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    host = SearchEngine.create_default(device=-1); host.index_flat(None, arena, offs, s.field_weights)      # planning only: Search on it fails (no GPU path)
    ex = o.export_index(); N = o.num_docs; T = o.num_terms; avgdl = np.float32(o.avgdl)
    # Stage-2 text = lower(Normalize(IndexedText)) (SegmentProcessor.cs:42-75)
    raw = arena.tobytes().decode("utf-16-le")
    docs = [raw[int(offs[d]):int(offs[d + 1])] for d in range(N)]
    norm = [normalize(t, lower=True) for t in docs]
    tarr = [_u16(t) for t in norm]; toffs = np.zeros(N + 1, np.uint64); toffs[1:] = np.cumsum([len(a) for a in tarr]); text = np.concatenate(tarr)
    idx = C.c_void_p(); st = C.c_void_p()
    cfg = Cfg(0, 0, DEPTH, 0)
    chk(L, L.infx_create(C.byref(cfg), C.byref(idx)))
    keys = np.arange(N, dtype=np.int64)
    chk(L, L.infx_upload_docs(idx, N, _p(ex["doc_len"], C.c_float), C.c_float(avgdl), _p(keys, C.c_int64), None, _p(toffs, C.c_uint64), _p(text, C.c_uint16)))
    chk(L, L.infx_upload_postings(idx, T, _p(ex["post_off"], C.c_uint64), _p(ex["post_doc"], C.c_int32), _p(ex["post_w"], C.c_uint8), _p(ex["df"], C.c_int32)))
    z = np.zeros(1, np.uint64)
    chk(L, L.infx_upload_prefix_docsets(idx, 0, _p(z, C.c_uint64), None))
    chk(L, L.infx_stream_create(idx, C.byref(st)))
    qa, qo = s.queries(600, qseed=17, fuzz=0.0)
    texts = Synth.texts(qa, qo)
    # maxScore of a term (VectorModel.cs:525-531), fp32 like the reference
    f32 = np.float32
    minDl = f32(f32(1) - f32(0.75)) + f32(f32(0.75) * f32(f32(1) / avgdl)); maxCore = f32(f32(255) * f32(f32(1.2) + f32(1))) / f32(f32(255) + f32(f32(1.2) * minDl))
    used, qs, terms = [], [], []
    for q in texts:
        p = host.plan(q)
        if p["flags"] or p["prefix_set"] >= 0 or len(p["term_ids"]) == 0 or (p["term_ids"] < 0).any():
            continue
        qs.append(Query(len(terms), len(p["term_ids"]), p["mode"], -1, DEPTH, p["n_and"], p["df_s1"], p["df_s2"]))
        for i in range(len(p["term_ids"])):
            idf = f32(p["idf"][i]); terms.append(Term(int(p["term_ids"][i]), 0, 0, float(idf), float(f32(idf * f32(maxCore + f32(1)))), int(p["roles"][i]), int(p["ranks"][i]), 0))
        used.append(q)
    nq = len(qs); assert nq >= 60          # queries with a prefix DocSet need infx_upload_prefix_docsets with the host's set numbering: left to the engine tests
    QA = (Query * nq)(*qs); TA = (Term * len(terms))(*terms)
    hits = (Hit * (nq * DEPTH))(); hc = np.zeros(nq, np.uint32)
    chk(L, L.infx_stage1_batch(st, nq, QA, len(terms), TA, 0, None, hits, _p(hc, C.c_uint32)))
    o.set_trace(True)
    cov_rows, cov_q, want = [], [], []
    sz = L.infx_sizeof_cov_query()
    for i, q in enumerate(used):
        o.search(q, 10, DEPTH)
        ok, osc = o.last_stage1()
        got = {hits[i * DEPTH + k].doc: hits[i * DEPTH + k].score for k in range(int(hc[i]))}
        assert set(got) == set(ok.tolist()), (q, sorted(set(got) ^ set(ok.tolist())))               # internal id == DocumentKey here; exact replay: the SET is the oracle's
        for d, sc in zip(ok.tolist(), osc.tolist()):
            assert abs(got[d] - sc) <= SCORE_RTOL * max(abs(sc), 1e-9), (q, d, got[d], sc)
        ids, base, sc2, ties, feat = o.last_trace()
        if len(ok) < 2 or len(ids) == 0 or len(cov_q) >= 48:
            continue
        buf = (C.c_uint8 * sz)()
        a = _u16(q)
        assert host.L.infx_engine_prepare_cov_query(host.h, _p(a, C.c_uint16), len(a), buf) == 0
        first2 = ok.tolist()[:2]                                                                       # docIndex 0 / 1: the first two Stage-1 keys (quirk Q7)
        for k in range(len(ids)):
            cov_rows.append(CovCand(len(cov_q), int(ids[k]), float(base[k]), 1 if int(ids[k]) in first2 else 0)); want.append((q, float(sc2[k]), int(ties[k]), feat[k]))
        cov_q.append(bytes(buf))
    assert len(cov_q) >= 12
    qbuf = (C.c_uint8 * (sz * len(cov_q))).from_buffer_copy(b"".join(cov_q))
    CA = (CovCand * len(cov_rows))(*cov_rows); outs = (CovOut * len(cov_rows))(); fo = np.zeros((len(cov_rows), NFEAT), np.int32)
    chk(L, L.infx_stage2_batch(st, len(cov_q), qbuf, len(cov_rows), CA, outs, _p(fo, C.c_int32)))
    for k, (q, sc, tie, feat) in enumerate(want):
        assert outs[k].status == 0
        assert np.array_equal(fo[k, :O.N_INT_FEAT], feat[:O.N_INT_FEAT]), (q, cov_rows[k].doc, fo[k, :O.N_INT_FEAT].tolist(), feat[:O.N_INT_FEAT].tolist())
        assert outs[k].tiebreaker == tie and abs(outs[k].score - sc) <= 2.0 ** -6 + 1e-6, (q, outs[k].score, sc)
    L.infx_stream_destroy(st); L.infx_destroy(idx)


@pytest.mark.parametrize("flags", [1, 0])          # INFX_CFG_NO_EXACT_REPLAY: the first pass alone; 0: with the replay behind it
def test_member_list_virtual_term_counts_every_document_once(flags):
    """infx_term.reserved == 1 (a fuzzy virtual term given as its LD1 member term ids): the union is formed inside the accumulate launch and every
    document counts with tf == 1 (RoaringPostingsEnum.Freq, Indexing/RoaringPostingsEnum.cs:21) — also where a member's own posting has tf >= 2.
    Expected values: the same query with the union materialised by the caller (reserved == 0), and BM25+ restated in numpy fp32 over the oracle's arrays."""
    L = load_library()
    s = Synth(2, docs=20000)
    arena, offs = s.docs()
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    ex = o.export_index(); N = o.num_docs; T = o.num_terms; avgdl = np.float32(o.avgdl)
    idx = C.c_void_p(); st = C.c_void_p()
    cfg = Cfg(0, 0, DEPTH, flags)
    chk(L, L.infx_create(C.byref(cfg), C.byref(idx)))
    keys = np.arange(N, dtype=np.int64)
    chk(L, L.infx_upload_docs(idx, N, _p(ex["doc_len"], C.c_float), C.c_float(avgdl), _p(keys, C.c_int64), None, None, None))
    chk(L, L.infx_upload_postings(idx, T, _p(ex["post_off"], C.c_uint64), _p(ex["post_doc"], C.c_int32), _p(ex["post_w"], C.c_uint8), _p(ex["df"], C.c_int32)))
    z = np.zeros(1, np.uint64)
    chk(L, L.infx_upload_prefix_docsets(idx, 0, _p(z, C.c_uint64), None))
    chk(L, L.infx_stream_create(idx, C.byref(st)))
    po, pd, pw = ex["post_off"], ex["post_doc"], ex["post_w"]
    # members: terms that hold postings with tf >= 2, moderately long lists, overlapping documents between members
    cand = [t for t in range(T) if 200 <= po[t + 1] - po[t] <= 4000 and pw[int(po[t]):int(po[t + 1])].max() >= 2]
    assert len(cand) >= 6
    f32 = np.float32
    norm = (f32(1.2) * (f32(f32(1) - f32(0.75)) + f32(f32(0.75) / avgdl) * ex["doc_len"].astype(np.float32))).astype(np.float32)      # Bm25Scorer.cs:413
    qs, terms, extra, want = [], [], [], []
    for g in range(0, 6, 3):
        members = cand[g:g + 3]
        docs = np.unique(np.concatenate([pd[int(po[t]):int(po[t + 1])] for t in members])).astype(np.int32)
        df = len(docs)
        idf = f32(np.log(f32(f32(N - df) + f32(0.5)) / f32(f32(df) + f32(0.5)) + f32(1)))
        sc = (idf * (f32(1) * f32(2.2) / (f32(1) + norm[docs]) + f32(1))).astype(np.float32)
        want.append(dict(zip(docs.tolist(), sc.tolist())))
        for reserved in (1, 0):      # the same query twice: member list, then the caller's union
            qs.append(Query(len(terms), 1, 2, -1, DEPTH, 1, 1, 0))      # INFX_MODE_DISJ, one eligible term of rank 0
            if reserved:
                terms.append(Term(-1, len(extra), len(members), float(idf), float(idf * 3.2), 16, 0, 1)); extra += members
            else:
                terms.append(Term(-1, len(extra), len(docs), float(idf), float(idf * 3.2), 16, 0, 0)); extra += docs.tolist()
    nq = len(qs)
    QA = (Query * nq)(*qs); TA = (Term * len(terms))(*terms); EX = np.asarray(extra, np.int32)
    hits = (Hit * (nq * DEPTH))(); hc = np.zeros(nq, np.uint32)
    chk(L, L.infx_stage1_batch(st, nq, QA, len(terms), TA, len(EX), _p(EX, C.c_int32), hits, _p(hc, C.c_uint32)))
    for i in range(nq):
        got = {hits[i * DEPTH + k].doc: hits[i * DEPTH + k].score for k in range(int(hc[i]))}
        w = want[i // 2]
        assert len(got) == min(DEPTH, len(w))
        cut = sorted(w.values(), reverse=True)[len(got) - 1]
        for d, sc in got.items():
            assert d in w and abs(sc - w[d]) <= SCORE_RTOL * w[d], (i, d, sc, w.get(d))      # tf == 1 for every document of the union
            assert w[d] >= cut * (1 - 1e-5)
        if i % 2 == 1:                                                                         # member list == caller-built union, row for row
            prev = {hits[(i - 1) * DEPTH + k].doc: hits[(i - 1) * DEPTH + k].score for k in range(int(hc[i - 1]))}
            assert prev == got
    L.infx_stream_destroy(st); L.infx_destroy(idx)
