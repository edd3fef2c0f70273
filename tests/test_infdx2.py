"""INFDX2 reader (infx_engine_load_index, csrc/host/infdx2.h) against files in the reference's Save format written by tests/infdx2_writer.py from the
ORACLE's index of the same corpus (documents + every non-stop term with its postings and weight bytes): the product must accept the file — i.e. its own
builder reproduces every stored posting — and refuse corrupted files and files whose postings it would not reproduce."""
import numpy as np
import pytest

from infidex_amd import SearchEngine
from infidex_amd.engine import InfidexError
from tests import infdx2_writer as W
from tests import oracle_lib as O
from tests.test_oracle_kats import TEN_DOCS
from tools.synth import Synth


def _oracle_file(path, docs, corrupt=None, bump_weight=False, deleted=(), tamper=None, mutate_terms=None):
    o = O.OracleEngine.create_default(); o.index(docs)
    ex = o.export_index()
    terms = []
    for t in range(o.num_terms):
        if ex["df"][t] <= 0:
            continue                                              # WriteTerms keeps DocumentFrequency > 0 (stop terms are not stored)
        a, b = int(ex["post_off"][t]), int(ex["post_off"][t + 1])
        post = list(zip(ex["post_doc"][a:b].tolist(), ex["post_w"][a:b].tolist()))
        terms.append((o.term_text(t), int(ex["df"][t]), post))
    if mutate_terms is not None:
        mutate_terms(terms)
    if bump_weight:
        text, df, post = terms[len(terms) // 2]; terms[len(terms) // 2] = (text, df, [(post[0][0], post[0][1] + 1)] + post[1:])
    # the derived sections as the reference derives them: term FST over EVERY term of the collection, short-query index, metadata cache, WordMatcher
    derived, trailer = W.derived_sections([o.term_text(t) for t in range(o.num_terms)], [t for _, t in docs], O.normalize)
    if tamper is not None:
        derived, trailer = tamper(derived, trailer)
    W.write(path, [(i, k, t, k in deleted) for i, (k, t) in enumerate(docs)], terms, derived=derived, trailer=trailer)
    if corrupt is not None:
        raw = bytearray(open(path, "rb").read()); raw[corrupt] ^= 0x40; open(path, "wb").write(bytes(raw))
    return o, len(terms), sum(len(p) for _, _, p in terms)


def test_reader_accepts_what_the_reference_format_holds(tmp_path):
    s = Synth(2, docs=3000); arena, offs = s.docs()
    raw = arena.tobytes().decode("utf-16-le")
    docs = [(1000 + d, raw[int(offs[d]):int(offs[d + 1])]) for d in range(3000)] + [(5000 + k, t) for k, t in TEN_DOCS] + [(9001, "Žďár nad Sázavou škola"), (9002, "")]
    p = str(tmp_path / "idx.infdx2")
    o, nterms, npost = _oracle_file(p, docs)
    e = SearchEngine.create_default(device=-1)                   # host-only engine: the reader and its cross-check need no GPU
    assert e.load_index(p) == (len(docs), nterms, npost) and nterms > 1000 and npost > 50000
    st = e.index_stats(); assert st["docs"] == len(docs)
    assert e.plan("batman robin")["term_ids"].size > 0


@pytest.mark.parametrize("what", ["magic", "header", "data", "truncated", "weights"])
def test_reader_refuses_foreign_corrupted_and_inconsistent_files(tmp_path, what):
    docs = [(k, t) for k, t in TEN_DOCS]
    p = str(tmp_path / "bad.infdx2")
    _oracle_file(p, docs, corrupt={"magic": 2, "header": 12, "data": 60}.get(what), bump_weight=(what == "weights"))
    if what == "truncated":
        raw = open(p, "rb").read(); open(p, "wb").write(raw[:len(raw) // 2])
    e = SearchEngine.create_default(device=-1)
    with pytest.raises(InfidexError) as ex:
        e.load_index(p)
    assert ex.value.code == (5 if what == "weights" else 1), (what, str(ex.value))      # INFX_EUNSUPPORTED: postings this builder would not produce; INFX_EINVAL otherwise


def _sections(docs):
    """Byte ranges of the three derived sections inside `derived` (the writer concatenates them) for the tamper tests."""
    o = O.OracleEngine.create_default(); o.index(docs)
    texts = [t for _, t in docs]
    fst = W.fst_section([(o.term_text(t), t) for t in range(o.num_terms)])
    sq = W.short_query_section([O.normalize(t, True) for t in texts])
    return len(fst), len(sq)


TAMPER_DOCS = [(k, t) for k, t in TEN_DOCS] + [(77, "Žďár nad Sázavou škola"), (78, "the batman returns again and again")]


@pytest.mark.parametrize("what", ["fst_output", "fst_missing_term", "short_position", "short_extra_list", "meta_count", "meta_first", "wm_absent", "wm_exact_doc",
                                  "wm_ld1_key", "wm_affix_last_doc", "wm_trailing", "derived_trailing"])
def test_reader_refuses_derived_sections_that_disagree_with_the_documents(tmp_path, what):
    """Every derived section is checked against the index rebuilt from the stored documents (csrc/host/infdx2_verify.h): a file whose FST, short-query
    index, metadata cache or WordMatcher says something else than its documents is refused, and the engine stays usable."""
    import struct
    nfst, nsq = _sections(TAMPER_DOCS)
    o = O.OracleEngine.create_default(); o.index(TAMPER_DOCS)
    texts = [t for _, t in TAMPER_DOCS]
    terms = [o.term_text(t) for t in range(o.num_terms)]
    meta = [O.normalize(t.lower(), False) for t in texts]

    def tamper(derived, trailer):
        d = bytearray(derived); t = bytearray(trailer)
        if what == "fst_output":                               # two terms swap their collection indexes
            pairs = [(x, i) for i, x in enumerate(terms)]; pairs[3], pairs[4] = (pairs[3][0], 4), (pairs[4][0], 3)
            d[:nfst] = W.fst_section(pairs)
        elif what == "fst_missing_term":
            fst = W.fst_section([(x, i) for i, x in enumerate(terms[:-1])]); d[:nfst] = fst
        elif what == "short_position":                         # first posting of the first single-character list: position + 1
            at = nfst + 4 + 2 + 4 + 4; d[at:at + 2] = struct.pack("<H", struct.unpack_from("<H", d, at)[0] + 1)
        elif what == "short_extra_list":                       # one more 2-character prefix nobody's text holds
            sq = bytearray(d[nfst:nfst + nsq]); idx = W.short_query_section([O.normalize(x, True) for x in texts] + ["qz"])
            d[nfst:nfst + nsq] = idx                          # (document id = len(texts): out of range as well)
        elif what == "meta_count":
            d[-2:] = struct.pack("<H", struct.unpack_from("<H", d, len(d) - 2)[0] + 1)
        elif what == "meta_first":
            d[nfst + nsq:] = W.metadata_section(["zzz " + meta[0]] + meta[1:])
        elif what == "wm_absent":
            t = bytearray(b"\x00")
        elif what == "wm_exact_doc":                           # a word of the first document is spelt differently in the stored dictionaries
            w0 = next(w for w in W.words_of(meta[0]) if 3 <= len(w) <= 8)
            t = bytearray(W.wordmatcher_section([meta[0].replace(w0, w0[:-1] + ("q" if w0[-1] != "q" else "x"), 1)] + meta[1:]))
        elif what == "wm_ld1_key":
            t = bytearray(W.wordmatcher_section(meta, max_ld1=7))
        elif what == "wm_affix_last_doc":                      # the affix FST leads to the FIRST occurrence instead of the last
            full = W.wordmatcher_section(meta)
            occ = [(w, d) for d, m in enumerate(meta) for w in W.words_of(m) if len(w) >= 3]
            first = {}
            for i, (w, _) in enumerate(occ):
                first.setdefault(w, i)
            good = W.fst_section([(w, i) for i, (w, _) in enumerate(occ)]); bad = W.fst_section([(w, first[w]) for w, _ in occ])
            assert good in full and len(good) == len(bad) and good != bad
            t = bytearray(full.replace(good, bad))
        elif what == "wm_trailing":
            t += b"\x00"
        elif what == "derived_trailing":
            d += b"\x00\x00"
        return bytes(d), bytes(t)

    p = str(tmp_path / "bad.infdx2")
    _oracle_file(p, TAMPER_DOCS, tamper=tamper)
    e = SearchEngine.create_default(device=-1)
    with pytest.raises(InfidexError) as ex:
        e.load_index(p)
    assert ex.value.code == 5, (what, str(ex.value))           # INFX_EUNSUPPORTED
    print(what, "->", str(ex.value))
    section = {"fst": "FST", "short": "short-query", "meta": "metadata", "wm": ("WordMatcher", "affix", "symmetric-delete", "exact-word"), "derived": "data section"}[what.split("_")[0]]
    assert any(x in str(ex.value) for x in ((section,) if isinstance(section, str) else section)), (what, str(ex.value))
    # the refused file left the engine unindexed: the intact file loads into the same instance
    good = str(tmp_path / "good.infdx2")
    _oracle_file(good, TAMPER_DOCS)
    assert e.load_index(good)[0] == len(TAMPER_DOCS)


def test_flipped_bytes_in_the_derived_sections_never_get_past_the_bounds_checks(tmp_path):
    """400 single-byte corruptions — half in the WordMatcher section (no checksum covers it), half in the derived sections with the data checksum recomputed (a
    crafted file): each load either refuses the file or — when the byte is one nothing depends on (a Roaring offset, an arc's redundant output) — accepts it;
    none may crash or read out of bounds."""
    import struct
    rng = np.random.default_rng(11)
    base = str(tmp_path / "base.infdx2")
    _oracle_file(base, TAMPER_DOCS)
    raw = open(base, "rb").read()
    dlen = struct.unpack_from("<I", raw, 26)[0]
    data_at, data_end = 30, 30 + dlen
    nfst, nsq = _sections(TAMPER_DOCS)
    derived_len = len(W.derived_sections([""], [t for _, t in TAMPER_DOCS], O.normalize)[0]) - len(W.fst_section([("", 0)])) + nfst
    refused = accepted = 0
    for k in range(400):
        b = bytearray(raw)
        if k % 2 == 0:
            at = int(rng.integers(data_end + 4, len(raw)))
            b[at] ^= 1 << int(rng.integers(0, 8))
        else:
            at = int(rng.integers(data_end - derived_len, data_end))
            b[at] ^= 1 << int(rng.integers(0, 8))
            b[data_end:data_end + 4] = struct.pack("<I", W.checksum_bytes(bytes(b[data_at:data_end])))
        p = str(tmp_path / "f.infdx2")
        open(p, "wb").write(bytes(b))
        e = SearchEngine.create_default(device=-1)
        try:
            e.load_index(p); accepted += 1
        except InfidexError as ex:
            assert ex.code in (1, 5); refused += 1
    print("refused", refused, "accepted", accepted)
    assert refused >= 396


def test_reference_persistence_kats(tmp_path):
    """PersistenceTests.cs:13-64 (SaveAndLoadIndex_PreservesData) and :152-196 (SaveAndLoadIndex_UnicodeSurrogateCharacters), as far as they reach this path:
    the saved index of the two-document corpus answers "fox" -> [1], "dog" -> [2] (the oracle, as the reference asserts before AND after the round trip) and loads
    with the vocabulary it was saved with; the index of the single document U+1F50D — whose first 3-gram ends in HALF the surrogate pair, which BinaryWriter writes
    as U+FFFD — loads too (the reference's own Load accepts it), again with the same vocabulary.  The query of the second test is two UTF-16 units long: the
    ShortQueryProcessor path, outside this hot path (the product flags it unsupported)."""
    docs = [(1, "The quick brown fox"), (2, "jumps over the lazy dog")]
    p = str(tmp_path / "test_index.bin")
    o, nterms, _ = _oracle_file(p, docs)
    assert o.search("fox", 10)["keys"] == [1] and o.search("dog", 10)["keys"] == [2]
    e = SearchEngine.create_default(device=-1)
    assert e.load_index(p)[:2] == (2, nterms) and e.index_stats()["terms"] == o.num_terms

    docs = [(1, "\U0001F50D")]
    p = str(tmp_path / "surrogates_index.bin")
    o, nterms, _ = _oracle_file(p, docs)
    assert nterms == 2 and any("\ufffd" in W.lossy(o.term_text(t)) for t in range(o.num_terms))      # the file really holds a replaced half pair
    e = SearchEngine.create_default(device=-1)
    assert e.load_index(p)[:2] == (1, 2) and e.index_stats()["terms"] == o.num_terms == 2


ASTRAL_DOCS = [(1, "\U0001F50Dab zeta"), (2, "\U0001F50Eab yotta"), (3, "plain \U0001F50Dab"), (4, "x\U0001F50D \U0001F50Ex \U0001F50Dab"), (5, "\U0001F50D"),
               (6, "\U00020000\U00020001 cjk\U00020001"), (7, "\ufffdab already replaced"), (8, "x\U00020000 end")]          # "x\ud83d" / "x\ud840": two token prefixes that are stored as "x\ufffd"


def test_keys_cut_out_of_surrogate_pairs_are_matched_through_the_lossy_utf8(tmp_path):
    """Characters outside the BMP: 3-gram windows, 1..3-unit token prefixes and single-unit deletions cut surrogate pairs apart, BinaryWriter stores the halves as
    U+FFFD, and several different keys collapse onto one stored text ("\udd0dab" and "\udd0eab" -> "\ufffdab", next to a document that really holds "\ufffdab").
    The reader matches them in their order of first appearance; the file loads, and it stops loading when two colliding terms swap their postings."""
    p = str(tmp_path / "astral.infdx2")
    o, nterms, npost = _oracle_file(p, ASTRAL_DOCS)
    texts = [W.lossy(o.term_text(t)) for t in range(o.num_terms)]
    assert len(set(texts)) < len(texts)                                # at least two terms share their stored text
    e = SearchEngine.create_default(device=-1)
    assert e.load_index(p) == (len(ASTRAL_DOCS), nterms, npost)

    # swap the postings of two terms with the same stored text: the same bytes for the names, each other's lists under them
    def swap(terms):
        first = {}
        for i, (text, df, post) in enumerate(terms):
            x = W.lossy(text)
            if x in first and (terms[first[x]][1], terms[first[x]][2]) != (df, post):
                j = first[x]
                terms[i], terms[j] = (terms[i][0], terms[j][1], terms[j][2]), (terms[j][0], df, post)
                return
            first.setdefault(x, i)
        raise AssertionError("no two colliding terms with different postings in this corpus")
    bad = str(tmp_path / "astral_bad.infdx2")
    _oracle_file(bad, ASTRAL_DOCS, mutate_terms=swap)
    e = SearchEngine.create_default(device=-1)
    with pytest.raises(InfidexError) as ex:
        e.load_index(bad)
    assert ex.value.code == 5, str(ex.value)


def test_roaring_bitmaps_of_every_container_kind(tmp_path):
    """A word that occurs in > 4096 documents of one 65 536-id container is stored as a bitmap container, a rarer one as an array container, and document
    ids beyond 65 535 open a second container: all three decode to the rebuilt lists."""
    docs = [(i, "common word" + (" rare" if i % 1000 == 0 else "") + (" beyond" if i >= 65536 else "")) for i in range(70000)]
    p = str(tmp_path / "big.infdx2")
    _, nterms, npost = _oracle_file(p, docs)
    e = SearchEngine.create_default(device=-1, threads=4)
    assert e.load_index(p) == (len(docs), nterms, npost)


@pytest.mark.gpu
def test_loaded_index_searches_like_the_indexed_one(tmp_path):
    s = Synth(2, docs=20000); arena, offs = s.docs()
    raw = arena.tobytes().decode("utf-16-le")
    docs = [(d, raw[int(offs[d]):int(offs[d + 1])]) for d in range(20000)]
    p = str(tmp_path / "idx.infdx2")
    o, _, _ = _oracle_file(p, docs, deleted={7, 11})
    a = SearchEngine.create_default(device=0); a.load_index(p)
    o.delete_keys([7, 11])
    qa, qo = s.queries(100, qseed=3, fuzz=0.3)
    texts = Synth.texts(qa, qo)
    for q, r in zip(texts, a.search_batch(texts, 10)):
        w = o.search(q, 10)
        assert [x.document_id for x in r.records] == w["keys"], q
        assert not ({7, 11} & {x.document_id for x in r.records})
