"""INFS segment reader (infx_segment_*, infx_engine_verify_segment; csrc/host/infs.h) against files in the reference's Flush format written by
tests/infs_writer.py: the reference's own known answers (SegmentTests.cs), round trips over the ORACLE's index of synthetic and Unicode corpora, every block-size
regime of BlockPostingsWriter (dense lists of 256-posting blocks, sparse lists cut by the density rule, short last groups of GroupVarInt), refusals of corrupted
files, and the cross-check of a segment against the product's index of the same documents."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from infidex_amd import SearchEngine
from infidex_amd.engine import InfidexError, _p, load_library
from tests import infs_writer as W
from tests import oracle_lib as O
from tools.synth import Synth


def _read(path):
    L = load_library()
    h = C.c_void_p()
    rc = L.infx_segment_open(path.encode(), C.byref(h))
    if rc:
        raise InfidexError(rc, (L.infx_engine_last_error() or b"").decode("utf-8", "replace"))
    try:
        dc = C.c_int32(); nt = C.c_int32(); npost = C.c_int64(); nch = C.c_int64()
        assert L.infx_segment_info(h, C.byref(dc), C.byref(nt), C.byref(npost), C.byref(nch)) == 0
        T, P = nt.value, npost.value
        toff = np.zeros(T + 1, np.uint32); tch = np.zeros(max(1, nch.value), np.uint16); poff = np.zeros(T + 1, np.uint64)
        docs = np.zeros(max(1, P), np.int32); w = np.zeros(max(1, P), np.uint8)
        assert L.infx_segment_export(h, _p(toff, C.c_uint32), _p(tch, C.c_uint16), _p(poff, C.c_uint64), _p(docs, C.c_int32), _p(w, C.c_uint8)) == 0
        terms = [tch[int(toff[i]):int(toff[i + 1])].tobytes().decode("utf-16-le") for i in range(T)]
        return dc.value, terms, poff, docs[:P], w[:P]
    finally:
        L.infx_segment_close(h)


def test_segment_tests_known_answers(tmp_path):
    """SegmentTests.cs:11-51 (WriteAndReadSegment_ShouldWork) and the merged segment of :54-120 (MergeSegments_ShouldWork) as the merger writes it."""
    p = str(tmp_path / "a.seg")
    W.write(p, [("apple", [1, 3], [10, 20]), ("banana", [2], [5])], 5)
    dc, terms, poff, docs, w = _read(p)
    assert dc == 5 and terms == ["apple", "banana"]
    assert docs[int(poff[0]):int(poff[1])].tolist() == [1, 3] and w[int(poff[0]):int(poff[1])].tolist() == [10, 20]
    assert docs[int(poff[1]):int(poff[2])].tolist() == [2] and w[int(poff[1]):int(poff[2])].tolist() == [5]
    assert "orange" not in terms
    p2 = str(tmp_path / "m.seg")
    W.write(p2, [("common", [1, 5], [10, 30]), ("unique1", [2], [20]), ("unique2", [8], [40])], 10)
    dc, terms, poff, docs, w = _read(p2)
    assert dc == 10 and terms == ["common", "unique1", "unique2"]
    assert docs.tolist() == [1, 5, 2, 8] and w.tolist() == [10, 30, 20, 40]


def test_block_regimes_and_varint_groups(tmp_path):
    rng = np.random.default_rng(11)
    N = 3_000_000
    lists = {
        "dense": np.arange(0, 5000, 1),                                     # 256-posting blocks
        "sparse": np.sort(rng.choice(N, 3000, replace=False)),              # density rule: blocks of 64
        "mixed": np.unique(np.concatenate([np.arange(100, 400), rng.choice(N, 500, replace=False)])),
        "one": np.asarray([N - 1]), "five": np.asarray([0, 1, 70000, 70001, 17_000_000 % N]),
        "wide": np.asarray([2 ** 24 + 5, 2 ** 24 + 6 + 2 ** 16]),           # 4-byte and 3-byte deltas
    }
    lists["five"] = np.unique(lists["five"])
    terms = [(k, v.astype(np.int64).tolist(), rng.integers(1, 256, v.size).tolist()) for k, v in lists.items()]
    p = str(tmp_path / "b.seg")
    W.write(p, terms, 2 ** 25)
    dc, names, poff, docs, w = _read(p)
    assert names == sorted(lists)
    want = {k: (d, ww) for k, d, ww in terms}
    for i, k in enumerate(names):
        a, b = int(poff[i]), int(poff[i + 1])
        assert docs[a:b].tolist() == want[k][0] and w[a:b].tolist() == want[k][1], k


def _oracle_terms(o, lo=0, hi=None):
    ex = o.export_index(); out = []
    for t in range(o.num_terms):
        if ex["df"][t] <= 0:
            continue
        a, b = int(ex["post_off"][t]), int(ex["post_off"][t + 1])
        d = ex["post_doc"][a:b]; ww = ex["post_w"][a:b]
        if hi is not None:
            m = (d >= lo) & (d < hi); d = d[m] - lo; ww = ww[m]
        if d.size:
            out.append((o.term_text(t), d.tolist(), ww.tolist()))
    return out


def test_round_trip_of_an_oracle_index_and_cross_check_with_the_product(tmp_path):
    s = Synth(3, docs=6000); arena, offs = s.docs()
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    terms = _oracle_terms(o)
    p = str(tmp_path / "full.seg")
    W.write(p, terms, 6000)
    dc, names, poff, docs, w = _read(p)
    assert dc == 6000 and len(names) == len(terms) > 3000
    by = {t: (d, ww) for t, d, ww in terms}
    assert names == sorted(by, key=lambda x: [ord(c) for c in x])
    for i in range(0, len(names), 7):
        a, b = int(poff[i]), int(poff[i + 1])
        assert docs[a:b].tolist() == by[names[i]][0] and w[a:b].tolist() == by[names[i]][1]
    e = SearchEngine.create_default(device=-1)                              # the reader and the cross-check need no GPU
    e.index_flat(None, arena, offs, s.field_weights)
    chk = np.zeros(3, np.int64)
    e._check(e.L.infx_engine_verify_segment(e.h, p.encode(), 0, _p(chk, C.c_int64)))
    assert chk.tolist() == [6000, len(terms), sum(len(t[1]) for t in terms)]
    # a second flush: documents [4000, 6000) with ids relative to 4000 (VectorModel.Flush, VectorModel.cs:804-815)
    part = _oracle_terms(o, 4000, 6000)
    p2 = str(tmp_path / "part.seg")
    W.write(p2, part, 2000)
    e._check(e.L.infx_engine_verify_segment(e.h, p2.encode(), 4000, _p(chk, C.c_int64)))
    assert chk.tolist() == [2000, len(part), sum(len(t[1]) for t in part)]
    assert e.L.infx_engine_verify_segment(e.h, p2.encode(), 0, None) == 5              # the same segment against the wrong documents: INFX_EUNSUPPORTED
    assert e.L.infx_engine_verify_segment(e.h, p2.encode(), 5000, None) == 5             # ... outside the corpus
    bad = list(part); t0 = bad[len(bad) // 2]; bad[len(bad) // 2] = (t0[0], t0[1], [t0[2][0] % 255 + 1] + t0[2][1:])
    p3 = str(tmp_path / "bad.seg"); W.write(p3, bad, 2000)
    assert e.L.infx_engine_verify_segment(e.h, p3.encode(), 4000, None) == 5             # one weight byte off
    p4 = str(tmp_path / "short.seg"); W.write(p4, part[:-3], 2000)
    assert e.L.infx_engine_verify_segment(e.h, p4.encode(), 4000, None) == 5             # terms missing from the segment


def test_unicode_terms(tmp_path):
    from tests import unicode_corpus as U
    docs, _ = U.make(5)
    o = O.OracleEngine.create_default(); o.index(docs)
    terms = _oracle_terms(o)
    p = str(tmp_path / "u.seg"); W.write(p, terms, len(docs))
    dc, names, poff, docs_, w = _read(p)
    assert set(names) == {t[0] for t in terms} and names == sorted(names, key=lambda x: x.encode("utf-16-le").hex() and [ord(c) for c in x])


def test_corrupted_files_are_refused(tmp_path):
    rng = np.random.default_rng(3)
    terms = [(f"t{i:03d}", np.sort(rng.choice(100000, int(rng.integers(1, 600)), replace=False)).tolist(), None) for i in range(40)]
    terms = [(t, d, rng.integers(1, 200, len(d)).tolist()) for t, d, _ in terms]
    p = str(tmp_path / "c.seg")
    raw = W.write(p, terms, 100000)
    _read(p)
    n = len(raw)
    post_start, fst_start, off_start = struct.unpack("<qqq", raw[-24:])

    def refused(mut):
        q = str(tmp_path / "x.seg"); open(q, "wb").write(bytes(mut))
        with pytest.raises(InfidexError):
            _read(q)
    refused(raw[:30])                                                        # truncated
    for pos in (0, 4, 8, n - 24, n - 16, n - 8):                            # magic, version, term count, the three footer offsets
        m = bytearray(raw); m[pos] ^= 0x21; refused(m)
    m = bytearray(raw); m[12:16] = struct.pack("<i", 50); refused(m)        # document count below the doc ids
    m = bytearray(raw); m[post_start + 16 + 5] ^= 0x7F; refused(m)          # a varint byte of the first block (deltas -> ids no longer match the skip table)
    m = bytearray(raw); m[post_start + 4] ^= 0x03; refused(m)               # block count of the first list
    m = bytearray(raw); m[fst_start + 10 + 4 + 2] ^= 0x05; refused(m)       # the root's arc count
    m = bytearray(raw); m[off_start + 16 + 1] ^= 0x10; refused(m)           # a word of the Elias-Fano high bits
    # the weight of a posting: the skip table's max weight no longer holds when the largest weight of a block is lowered ... or the file still parses (a smaller weight
    # below the maximum is undetectable by construction) — the engine-side cross-check covers that case (test above)
    assert os.path.getsize(p) == n


def test_flipped_bits_never_get_past_the_bounds_checks(tmp_path):
    """600 single-bit flips anywhere in a segment file: the reader refuses the file or — where the bit is one the format leaves free (a weight below its block's
    maximum, slack of the Elias-Fano arrays) — reads it; it never crashes and never reads outside the file.  Whatever it reads satisfies the invariants the
    export promises: ascending document ids below the document count in every list, term texts in ordinal order."""
    rng = np.random.default_rng(5)
    terms = [(f"w{i:03d}" + "x" * int(rng.integers(0, 4)), np.sort(rng.choice(200000, int(rng.integers(1, 900)), replace=False)).tolist(), None) for i in range(60)]
    terms = sorted([(t, d, rng.integers(1, 255, len(d)).tolist()) for t, d, _ in terms])
    raw = W.write(str(tmp_path / "base.seg"), terms, 200000)
    refused = read = 0
    q = str(tmp_path / "f.seg")
    for k in range(600):
        m = bytearray(raw)
        at = int(rng.integers(0, len(raw)))
        m[at] ^= 1 << int(rng.integers(0, 8))
        open(q, "wb").write(bytes(m))
        try:
            dc, names, poff, docs, w = _read(q)
        except InfidexError as ex:
            assert ex.code == 1; refused += 1
            continue
        read += 1
        assert names == sorted(names) and len(set(names)) == len(names)
        for i in range(len(names)):
            d = docs[int(poff[i]):int(poff[i + 1])]
            assert d.size == 0 or (np.all(np.diff(d) > 0) and 0 <= d[0] and d[-1] < dc)
    print("refused", refused, "read", read)
    assert refused > 300


def _flush_like(o, tmp_path, cuts):
    """Segment files as a sequence of Flush calls would leave them: documents [cuts[i], cuts[i + 1]) each, ids relative to the segment's first document."""
    paths, bases = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        p = str(tmp_path / f"flush{lo}.seg"); W.write(p, _oracle_terms(o, lo, hi), hi - lo)
        paths.append(p); bases.append(lo)
    return paths, bases


def test_an_engine_populated_from_segments_holds_the_index_of_the_documents(tmp_path):
    """Two flushed segments + a live tail -> infx_engine_index_from_segments: the flushed ranges' (document, weight) postings come from the files and the host index
    is, array for array, the one infx_engine_index_documents builds from the same documents.  A segment written from other documents is refused and leaves the
    engine reusable."""
    s = Synth(2, docs=6000); arena, offs = s.docs()
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    paths, bases = _flush_like(o, tmp_path, [0, 2500, 4700])                      # tail: documents [4700, 6000)
    e = SearchEngine.create_default(device=-1); e.index_flat_from_segments(None, arena, offs, s.field_weights, paths, bases)
    ref = SearchEngine.create_default(device=-1); ref.index_flat(None, arena, offs, s.field_weights)
    a, b = e.export_index(), ref.export_index()
    assert e.index_stats() == ref.index_stats()
    for k in ("df", "post_off", "post_doc", "post_w", "doc_len"):
        assert np.array_equal(a[k], b[k]), k
    assert a["avgdl"] == b["avgdl"]
    qa, qo = s.queries(40, qseed=3)
    for q in Synth.texts(qa, qo):
        pa, pb = e.plan(q), ref.plan(q)
        assert np.array_equal(pa["term_ids"], pb["term_ids"]) and np.array_equal(pa["idf"], pb["idf"]), q
    # no segments at all: everything is the live tail
    e0 = SearchEngine.create_default(device=-1); e0.index_flat_from_segments(None, arena, offs, s.field_weights, [], [])
    assert np.array_equal(e0.export_index()["post_doc"], b["post_doc"])
    # segments of OTHER documents / with a gap / out of order: refused, and the engine can still be indexed
    bad = SearchEngine.create_default(device=-1)
    with pytest.raises(Exception):
        bad.index_flat_from_segments(None, arena, offs, s.field_weights, paths[1:], bases[1:])       # does not start at document 0
    with pytest.raises(Exception):
        bad.index_flat_from_segments(None, arena, offs, s.field_weights, paths, [0, 2400])           # gap / overlap
    other = str(tmp_path / "other.seg"); W.write(other, [("zzzzzzzz", [0, 1], [1, 1])], 2500)
    with pytest.raises(Exception):
        bad.index_flat_from_segments(None, arena, offs, s.field_weights, [other], [0])               # a term the documents do not produce
    bad.index_flat(None, arena, offs, s.field_weights)
    assert np.array_equal(bad.export_index()["post_doc"], b["post_doc"])


def test_segments_keep_the_df_counter_and_the_stop_terms_of_an_unflushed_index(tmp_path):
    """What a segment file cannot give back: the reference's df counter counts a document twice when a term's weight byte saturates inside it
    (Term.cs:118-146, quirk Q5), and the segment writer drops the lists of terms that were stop terms AT FLUSH TIME (SegmentWriter skips df <= 0) — a term that
    becomes a stop term later still has its list in the earlier segment.  The import accumulates every document (so df and stop decisions are those of the unflushed
    index) and lets the segments replace the postings of their ranges: with a low stop-term limit and documents that repeat a word hundreds of times, the host
    index is still, array for array, the one infx_engine_index_documents builds."""
    from infidex_amd.engine import pack_texts
    rng = np.random.default_rng(11)
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet"]
    docs = []
    for i in range(900):
        k = int(rng.integers(2, 6)); t = " ".join(rng.choice(words, k))
        if i % 37 == 5:
            t = t + " " + " ".join(["golf"] * 300)             # the weight byte of "golf" (and of its n-grams) saturates inside this document
        if i in (40, 300, 500, 650, 880):
            t = t + " " + " ".join(["zulu"] * 140)             # a RARE word that saturates: it stays a live term whose df counts these documents several times
        docs.append(t)
    arena, offs = pack_texts(docs); fw = np.zeros(1, np.int32)
    LIMIT = 260                                                   # stop-term limit: the frequent terms cross it in the middle of the corpus
    cuts = [0, 150, 420, 700]
    paths, bases = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):                       # every flush sees the index as it was THEN: documents [0, hi)
        o = O.OracleEngine(enable_coverage=True, word_matcher=True, stop_term_limit=LIMIT)
        a2, o2 = pack_texts(docs[:hi]); o.add_flat(None, a2, o2, fw); o.finalize()
        p = str(tmp_path / f"flush{lo}.seg"); W.write(p, _oracle_terms(o, lo, hi), hi - lo); paths.append(p); bases.append(lo)
    full = O.OracleEngine(enable_coverage=True, word_matcher=True, stop_term_limit=LIMIT); full.add_flat(None, arena, offs, fw); full.finalize()
    exo = full.export_index()
    assert int((exo["df"] < 0).sum()) > 3                         # the full corpus has stop terms ...
    first = _read(paths[0])[1]
    stop_names = {full.term_text(t) for t in range(full.num_terms) if exo["df"][t] < 0}
    assert stop_names & set(first)                                # ... whose lists the FIRST flush still wrote
    e = SearchEngine(device=-1, stop_term_limit=LIMIT); e.index_flat_from_segments(None, arena, offs, fw, paths, bases)
    ref = SearchEngine(device=-1, stop_term_limit=LIMIT); ref.index_flat(None, arena, offs, fw)
    a, b = e.export_index(), ref.export_index()
    assert e.index_stats() == ref.index_stats()
    for k in ("df", "post_off", "post_doc", "post_w", "doc_len"):
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(b[k], exo[k]), k                    # and both are the oracle's index
    assert a["avgdl"] == b["avgdl"]
    # the saturation is real: some list's df exceeds its posting count
    plen = np.diff(exo["post_off"]); live = exo["df"] > 0
    assert np.any(exo["df"][live] > plen[live])


@pytest.mark.gpu
def test_an_engine_populated_only_from_segments_searches_like_the_oracle(tmp_path):
    """30 000 documents: two flushed segments + a live tail, no infx_engine_index_documents — searches equal the oracle's on the same documents."""
    from tests.parity_classify import assert_final_rows_match_oracle
    from infidex_amd.engine import pack_texts
    s = Synth(2, docs=30000); arena, offs = s.docs()
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    paths, bases = _flush_like(o, tmp_path, [0, 12000, 26000])
    e = SearchEngine.create_default(device=0); e.index_flat_from_segments(None, arena, offs, s.field_weights, paths, bases)
    qa, qo = s.queries(200, qseed=11)
    texts = Synth.texts(qa, qo)
    a2, o2 = pack_texts(texts)
    k, sc, t, c, f = e.search_packed(a2, o2, 10)
    assert_final_rows_match_oracle(k, sc, c, o, texts, 10, what="engine populated from two segments + a live tail")


@pytest.mark.gpu
def test_a_verified_segment_is_the_index_the_gpu_searches(tmp_path):
    """Flush-style segments of a corpus (two flushes) verified against the GPU engine's index of the same documents; the segment's CSR, mapped from term ordinals to
    the engine's term ids, is exactly the posting arrays the engine uploaded (infx_upload_postings layout); searches on that engine equal the oracle's."""
    from tests.parity_classify import assert_final_rows_match_oracle
    from infidex_amd.engine import pack_texts
    s = Synth(2, docs=30000); arena, offs = s.docs()
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    e = SearchEngine.create_default(device=0); e.index_flat(None, arena, offs, s.field_weights)
    chk = np.zeros(3, np.int64)
    for lo, hi in ((0, 18000), (18000, 30000)):
        part = _oracle_terms(o, lo, hi)
        p = str(tmp_path / f"f{lo}.seg"); W.write(p, part, hi - lo)
        e._check(e.L.infx_engine_verify_segment(e.h, p.encode(), lo, _p(chk, C.c_int64)))
        assert chk[0] == hi - lo and chk[1] == len(part)
    whole = str(tmp_path / "w.seg"); W.write(whole, _oracle_terms(o), 30000)
    dc, names, poff, docs, w = _read(whole)
    ex = e.export_index()
    for i in range(0, len(names), 11):                                       # ordinal -> engine term id by text; slices must be the uploaded arrays
        t = o.term_id(names[i])
        a, b = int(ex["post_off"][t]), int(ex["post_off"][t + 1])
        x, y = int(poff[i]), int(poff[i + 1])
        assert np.array_equal(ex["post_doc"][a:b], docs[x:y]) and np.array_equal(ex["post_w"][a:b], w[x:y])
    qa, qo = s.queries(120, qseed=5)
    texts = Synth.texts(qa, qo)
    a2, o2 = pack_texts(texts)
    k, sc, t, c, f = e.search_packed(a2, o2, 10)
    assert_final_rows_match_oracle(k, sc, c, o, texts, 10, what="engine behind verified segments")
