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


def _oracle_file(path, docs, corrupt=None, bump_weight=False, deleted=()):
    o = O.OracleEngine.create_default(); o.index(docs)
    ex = o.export_index()
    terms = []
    for t in range(o.num_terms):
        if ex["df"][t] <= 0:
            continue                                              # WriteTerms keeps DocumentFrequency > 0 (stop terms are not stored)
        a, b = int(ex["post_off"][t]), int(ex["post_off"][t + 1])
        post = list(zip(ex["post_doc"][a:b].tolist(), ex["post_w"][a:b].tolist()))
        terms.append((o.term_text(t), int(ex["df"][t]), post))
    if bump_weight:
        text, df, post = terms[len(terms) // 2]; terms[len(terms) // 2] = (text, df, [(post[0][0], post[0][1] + 1)] + post[1:])
    W.write(path, [(i, k, t, k in deleted) for i, (k, t) in enumerate(docs)], terms, derived=b"\xAB" * 37, trailer=b"\x01" + b"\xCD" * 11)
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
