"""Pins the oracle (CPU restatement, oracle/) against the reference's OWN known-answer tests.

Every case cites the reference test it restates (paths relative to /root/reference/src/Infidex.Tests).
The reference is C# and cannot run here (no .NET in the image), so these KATs are what pins the oracle.
CPU-only: runs under `-m "not gpu"`.
"""
import numpy as np
import pytest

from tests import oracle_lib as O

TEN_DOCS = [  # ReferenceMatchingTests.cs:20-31 == Infidex.Example/Example.cs:19-31
    (1, "The quick brown fox jumps over the lazy dog"),
    (2, "A journey of a thousand miles begins with a single step"),
    (3, "To be or not to be, that is the question"),
    (4, "All that glitters is not gold"),
    (5, "The fox was quick and clever in the forest"),
    (6, "Batman and Robin fight crime in Gotham City"),
    (7, "Superman flies faster than a speeding bullet"),
    (8, "Spider-Man swings through New York City"),
    (9, "Wonder Woman protects the innocent"),
    (10, "The Flash runs at incredible speeds"),
]


@pytest.fixture(scope="module")
def ten():
    e = O.OracleEngine.create_default()
    e.index(TEN_DOCS)
    return e


# ---- ReferenceMatchingTests.cs:39-98 : EXACT result lists -------------------------------------------------
def test_ref_batman_first_is_6(ten):
    r = ten.search("batman", 10)
    assert len(r["keys"]) >= 1 and r["keys"][0] == 6


def test_ref_qick_fux_exact(ten):
    assert ten.search("qick fux", 10)["keys"] == [5, 1]


def test_ref_battamam_exact(ten):
    assert ten.search("battamam", 10)["keys"] == [6]


def test_ref_new_york_exact(ten):
    assert ten.search("new york", 10)["keys"] == [8]


def test_ref_speeding_exact(ten):
    assert ten.search("speeding", 10)["keys"] == [7]


# ---- BASELINE config 1: README 3-doc corpus, "quik fox" (README.md:50-52) -----------------------------------
def test_config1_quickstart():
    e = O.OracleEngine.create_default()
    e.index([(1, "The quick brown fox jumps over the lazy dog"),
             (2, "A journey of a thousand miles begins with a single step"),
             (3, "To be or not to be that is the question")])
    r = e.search("quik fox", 10)
    assert r["keys"][0] == 1


# ---- SearchEngineTests.cs ----------------------------------------------------------------------------------
def test_se_fox_finds_1_and_4():  # :11-35
    e = O.OracleEngine.create_default()
    e.index([(1, "The quick brown fox jumps over the lazy dog"),
             (2, "A journey of a thousand miles begins with a single step"),
             (3, "To be or not to be that is the question"),
             (4, "The fox was quick and clever")])
    keys = e.search("fox", 10)["keys"]
    assert 1 in keys and 4 in keys


def test_se_exact_match_high_score():  # :37-54
    e = O.OracleEngine.create_default()
    e.index([(1, "hello world"), (2, "goodbye world"), (3, "hello there")])
    r = e.search("hello world", 10)
    assert r["keys"][0] == 1 and r["scores"][0] > 200


def test_se_fuzzy_batmam():  # :57-74
    e = O.OracleEngine.create_default()
    e.index([(1, "batman and robin"), (2, "superman flies high"), (3, "spiderman swings")])
    r = e.search("batmam", 10)
    assert r["keys"] and r["keys"][0] == 1


def test_se_empty_query():  # :77-90
    e = O.OracleEngine.create_default()
    e.index([(1, "hello world")])
    assert e.search("", 10)["keys"] == []


def test_se_no_matches():  # :92-106
    e = O.OracleEngine.create_default()
    e.index([(1, "hello world"), (2, "goodbye world")])
    r = e.search("xyzabc", 10)
    assert len(r["keys"]) == 0 or r["scores"][0] < 50


def test_se_multiword_ranks():  # :109-129
    e = O.OracleEngine.create_default()
    e.index([(1, "the quick brown fox"), (2, "the lazy brown dog"), (3, "a quick decision"), (4, "quick brown")])
    r = e.search("quick brown", 10)
    assert r["keys"][0] in (4, 1)


def test_se_minimal_engine():  # :150-164 (CreateMinimal: no coverage)
    e = O.OracleEngine.create_minimal()
    e.index([(1, "hello world"), (2, "goodbye world")])
    r = e.search("hello", 10)
    assert r["keys"] and r["keys"][0] == 1 and not r["used_coverage"]


# ---- QueryTests.cs:150-277 : exact COUNTS + ordering ---------------------------------------------------------
def test_qt_identical_docs_exactly_5():
    e = O.OracleEngine.create_default()
    e.index([(i, "batman saves the day") for i in range(20)])
    assert len(e.search("batman", 5)["keys"]) == 5


def test_qt_varied_docs_exactly_8():
    e = O.OracleEngine.create_default()
    e.index([(i, f"batman saves the day story {i}") for i in range(20)])
    assert len(e.search("batman", 8)["keys"]) == 8


BATMAN20 = [
    "Batman is a superhero appearing in American comic books published by DC Comics.",
    "The character was created by Bob Kane and Bill Finger, and first appeared in Detective Comics #27.",
    "Batman's secret identity is Bruce Wayne, a wealthy American playboy, philanthropist, and industrialist.",
    "He resides in Gotham City and operates out of the Batcave.",
    "His archenemy is the Joker, a criminal mastermind with a clown-like appearance.",
    "Other notable villains include Penguin, Riddler, Catwoman, and Two-Face.",
    "Batman comic books by DC Comics are very popular.",
    "Batman Arkham games are popular among gamers.",
    "The Dark Knight is a critically acclaimed Batman movie.",
    "Christian Bale played Batman in Christopher Nolan's trilogy.",
    "Batman drives the Batmobile through city streets.",
    "Batman has many enemies like Joker and Harley Quinn.",
    "Robin is Batman's sidekick.",
    "Alfred Pennyworth is Batman's loyal butler.",
    "Commissioner Gordon often works with Batman.",
    "The Justice League includes Batman, Superman, and Wonder Woman.",
    "Batman uses various gadgets and martial arts.",
    "Batman animated series is beloved by many fans.",
    "Zack Snyder directed Batman v Superman.",
    "Robert Pattinson is the latest actor to portray Batman.",
]


def test_qt_different_docs_exactly_12():
    e = O.OracleEngine.create_default()
    e.index(list(enumerate(BATMAN20)))
    assert len(e.search("batman", 12)["keys"]) == 12


DARK20 = [
    "Batman is a superhero appearing in American comic books.",
    "The character was created by Bob Kane and Bill Finger.",
    "Bruce Wayne is Batman's secret identity.",
    "He operates out of the Batcave in Gotham City.",
    "The Joker is Batman's archenemy and nemesis.",
    "The Dark Knight Rises",
    "Other villains include Penguin and Riddler.",
    "Batman comic books are published by DC Comics.",
    "The Dark Knight Rises is an epic conclusion",
    "Batman uses gadgets and martial arts skills.",
    "Christian Bale portrayed Batman in the trilogy.",
    "The Dark Knight was a critically acclaimed film.",
    "Robin is Batman's trusted sidekick and partner.",
    "Alfred Pennyworth is Batman's loyal butler.",
    "Commissioner Gordon works with Batman regularly.",
    "The Justice League includes Batman and Superman.",
    "Batman animated series is beloved by fans.",
    "Zack Snyder directed Batman v Superman movie.",
    "Robert Pattinson is the latest Batman actor.",
    "The Batmobile is Batman's iconic vehicle.",
]


def test_qt_dark_knight_rises_order():
    e = O.OracleEngine.create_default()
    e.index(list(enumerate(DARK20)))
    r = e.search("dark knight rises", 10)
    assert r["keys"][0] == 5
    assert 8 in r["keys"][:3]
    s = r["scores"]
    assert all(s[i - 1] >= s[i] for i in range(1, len(s)))


# ---- FuzzyRegressionTests.cs:31-58 ---------------------------------------------------------------------------
def test_fuzzy_the_matrx():
    e = O.OracleEngine.create_default()
    e.index([(1, "The Mat"), (2, "The Matrix"), (3, "The Matriarx"), (4, "The Match"), (5, "The Meatrix")])
    r = e.search("the matrx", 10)
    sc = dict(zip(r["keys"], r["scores"]))
    assert 2 in sc
    assert sc[2] > sc.get(1, 0.0)   # default(ScoreEntry).Score == 0 when doc 1 is absent


# ---- CoverageEngineTests.cs:18-119 ---------------------------------------------------------------------------
def test_cov_exact_match():
    cov, f, _, _ = O.coverage_standalone("hello world", "this is hello world text")
    assert cov > 200 and f["WordHits"] == 2


def test_cov_no_match():
    cov, _, _, _ = O.coverage_standalone("xyz abc", "hello world test")
    assert cov < 100


def test_cov_partial_match():
    cov, f, _, _ = O.coverage_standalone("hello world test", "hello world")
    assert cov > 100 and f["WordHits"] == 2


def test_cov_fuzzy_typo():
    cov, f, _, _ = O.coverage_standalone("batmam", "batman is a superhero")
    assert cov > 150 and f["WordHits"] > 0


def test_cov_joined_words():
    cov, _, _, _ = O.coverage_standalone("new york", "I live in newyork city")
    assert cov > 100


def test_cov_prefix():
    cov, _, _, _ = O.coverage_standalone("bat", "batman is a superhero")
    assert cov > 50


def test_cov_empty_query():
    cov, f, _, _ = O.coverage_standalone("", "hello world")
    assert cov == 0 and f["WordHits"] == 0


# ---- BugReproductionTests.cs:13-67 (fixed word-IDF cache, bm25 = 0.5) ------------------------------------------
def test_bug_matrix_rev_prefers_revisited():
    idf = {"the": 1.574, "matrix": 9.544, "rev": 9.515}
    _, _, s_rel, _ = O.coverage_standalone("the matrix rev", "The Matrix Reloaded", 0.0, 0.5, idf)
    _, _, s_rev, _ = O.coverage_standalone("the matrix rev", "The Matrix Revisited", 0.0, 0.5, idf)
    assert s_rev > s_rel


# ---- LevenshteinDistanceTests.cs ------------------------------------------------------------------------------
@pytest.mark.parametrize("a,b,d", [
    ("hello", "hello", 0), ("hello", "hallo", 1), ("bat", "brat", 1), ("batman", "batma", 1), ("abc", "xyz", 3),
    ("", "", 0), ("hello", "", 5), ("", "hello", 5), ("kitten", "sitting", 3), ("saturday", "sunday", 3),
])
def test_levenshtein(a, b, d):
    assert O.levenshtein(a, b) == d


def test_levenshtein_within():
    assert O.levenshtein("batman", "batmam", 1) <= 1      # IsWithinDistance :50
    assert O.levenshtein("batman", "ratmin", 1) > 1       # :57
    long1 = "a" * 100
    long2 = "a" * 50 + "b" + "a" * 49
    assert O.levenshtein(long1, long2) == 1               # :60-68


# ---- StringMetrics.Lcs worked examples (Metrics/StringMetrics.cs:25-27 comments) ---------------------------------
def test_lcs_examples():
    assert O.lcs("battamam", "batman", 1) == 4
    assert O.lcs("speeding", "speeds", 1) == 6
    assert O.lcs("fox", "the quick fox", 0) == 3


# ---- WordMatcherTests.cs ------------------------------------------------------------------------------------------
def test_wm_exact():
    e = O.OracleEngine.create_default()
    e.index([(0, "hello world test"), (1, "goodbye world")])
    assert e.wm_lookup("world").tolist() == [0, 1]


def test_wm_ld1():
    e = O.OracleEngine.create_default()
    e.index([(0, "batman is here")])
    r = e.wm_lookup("batmam")
    assert r is not None and 0 in r.tolist()


def test_wm_affix_prefix():
    e = O.OracleEngine.create_default()
    e.index([(0, "batman superman spiderman")])
    r = e.wm_lookup("bat", affix=True)
    assert r is not None and 0 in r.tolist()


def test_wm_affix_keeps_only_last_doc_quirk_q13():
    # WordMatcher.cs:166-196: _fstIndex is null while loading => the trie keeps the LAST occurrence's term id
    e = O.OracleEngine.create_default()
    e.index([(0, "batman one"), (1, "batman two"), (2, "batman three")])
    assert e.wm_lookup("bat", affix=True).tolist() == [2]


# ---- TokenizerTests.cs / quirk Q2 : "quik fox" -> 10 raw tokens / 9 distinct -------------------------------------------
def test_query_terms_quik_fox(ten):
    ten.search("quik fox", 10)
    t, df, idf, mx = ten.last_terms()
    # known terms only reach Bm25Scorer: 'fox' (word == its own 3-gram, deduped), trigrams of "  quik fox" that exist
    texts = [ten.term_text(i) for i in t if i >= 0]
    assert "fox" in texts and texts.count("fox") == 1
    assert list(t) == sorted(t)          # ascending termId order (quirk Q8)


def test_normalizer_and_case():
    assert O.normalize("Mateřská  škola\tBělohrad", lower=True) == "materska skola belohrad"
    assert O.normalize("a   b") == "a b"


def test_index_tf_bytes_quirk_q3():
    # Term.cs:71-122: Med weight 1.25 => byte sequence 1,2,3 == occurrence count; High 1.5 => 2,4,6 (banker's rounding)
    e = O.OracleEngine.create_default()
    e.add(1, [("abc abc abc", O.MED)])
    e.add(2, [("abc abc abc", O.HIGH)])
    e.finalize()
    ix = e.export_index()
    t = e.term_id("abc")
    lo, hi = int(ix["post_off"][t]), int(ix["post_off"][t + 1])
    # 'abc' occurs 3x as 3-gram + 3x as word per doc => 6 adds
    assert ix["post_doc"][lo:hi].tolist() == [0, 1]
    assert ix["post_w"][lo:hi].tolist() == [6, 12]


def test_stop_term_quirk_q5():
    e = O.OracleEngine(True, True, stop_term_limit=5)
    for i in range(8):
        e.add(i, "zzz common")
    e.finalize()
    ix = e.export_index()
    assert ix["df"][e.term_id("zzz")] == -1
    r = e.search("zzz", 10)
    assert r["keys"] == [] or r["keys"] is not None    # stop term silently dropped, no crash


def test_deleted_documents_are_skipped_not_reindexed():
    """Document.Deleted (Core/Document.cs, DocumentCollection.cs:200-212): the flag is checked on the query path only — Bm25Scorer.cs:322-323
    (flush), SearchPipeline.cs:404-406 / 463-465 (coverage), :532-537 (docIndex).  Index statistics are untouched, so the scores of the
    surviving documents do not move."""
    o = O.OracleEngine.create_default(); o.index(TEN_DOCS)
    before = o.search("batman", 10)
    assert before["keys"][0] == 6
    assert o.delete_keys([6]) == 1 and o.delete_keys([6]) == 0
    after = o.search("batman", 10)
    assert 6 not in after["keys"]
    # the other rows keep their scores: df / avgdl still include the deleted document
    b = dict(zip(before["keys"], before["scores"])); a = dict(zip(after["keys"], after["scores"]))
    assert all(abs(a[k] - b[k]) < 1e-6 for k in a if k in b)
    # a WordMatcher-only hit (typo query) on a deleted document disappears as well
    assert o.search("battamam", 10)["keys"] == []
    r = o.search("qick fux", 10)["keys"]
    o.delete_keys(r[:1])
    assert o.search("qick fux", 10)["keys"] == r[1:]


# ---- SynonymTests.cs:96-143 (search-level expectations; the reference builds these two engines with indexSizes [4, 5, 6] — the expectation tested is the
#      synonym behaviour, which the default configuration must show as well: both spellings are canonicalised to one root at index and at query time) ----
def test_synonyms_find_both_terms():  # :96-122
    o = O.OracleEngine.create_default(); o.add_synonym("car", "automobile")
    o.index([(1, "I drive a car to work"), (2, "This automobile is fast"), (3, "The truck is big")])
    keys = o.search("car", 10)["keys"]
    assert len(keys) >= 2 and {1, 2} <= set(keys)


def test_synonyms_work_in_both_directions():  # :124-143
    o = O.OracleEngine.create_default(); o.add_synonym("car", "automobile")
    o.index([(1, "I drive a car to work"), (2, "This automobile is fast")])
    keys = o.search("automobile", 10)["keys"]
    assert len(keys) >= 2 and {1, 2} <= set(keys)


# ---- SegmentTrackingTests.cs:92-210, 324-345: several documents under one DocumentKey (segments) -> one row per key ------------------------------
# (Document.SegmentNumber's index-time effect — continuation segments are tokenised without the start padding, Tokenizer.cs — is not restated: these
#  expectations do not depend on it.  What they pin: ConsolidateSegments keeps one row per key, Stage 2 scores a key through GetDocumentByPublicKey.)
SEGMENT_CASES = [
    ([(1, "Introduction to the topic of animals"), (1, "The quick brown fox jumps over the lazy dog"), (1, "Conclusion and summary of findings")], "fox", [1]),          # :92-115
    ([(1, "Introduction chapter one"), (1, "Batman fights crime in Gotham City"), (1, "Conclusion chapter one"), (2, "Batman and Robin save the day"),
      (2, "The end of their adventure"), (3, "Superman flies faster than a speeding bullet")], "batman", [1, 2]),                                                      # :118-148
    ([(1, "The cat sat on the mat"), (1, "The dog ran through the park"), (1, "The bird flew in the sky")], "batman", []),                                               # :151-166
    ([(1, "The cat sat on the mat"), (2, "The dog ran through the park"), (3, "The bird flew in the sky")], "batman", []),                                               # :169-184
    ([(1, "Chapter 1 introduction"), (1, "The hero begins his journey"), (2, "The hero saves the day"), (3, "A story about courage")], "hero", [1, 2]),                 # :187-211
    ([(1, f"Segment {i} text content") if i != 5 else (1, "This segment contains batman") for i in range(10)], "batman", [1]),                                          # :324-345
]


@pytest.mark.parametrize("case", range(len(SEGMENT_CASES)))
def test_segmented_documents_give_one_row_per_key(case):
    docs, q, want = SEGMENT_CASES[case]
    o = O.OracleEngine.create_default(); o.index(docs)
    r = o.search(q, 10)
    assert sorted(r["keys"]) == want and len(r["keys"]) == len(want)
    assert all(s > 0 for s in r["scores"])


def test_quirk_q18_the_lcs_of_a_second_evaluation_comes_back_from_a_byte():
    """SearchPipeline.cs:492-503: the LCS of docIndex 0 / 1 is computed on the document's first evaluation and stored as (byte)Math.Min(lcs, 255); a document that
    is evaluated twice (WordMatcher overlap row, then its Stage-1 row) reads the span back the second time.  For a query of more than 255 characters that the
    document contains, the first evaluation sees the full length, the second 255 — the oracle's trace shows both (the device reproduces it: k_stage2's `split`)."""
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet", "kilo", "lima", "mike", "november", "oscar", "papa"]
    q = " ".join(w + str(i) for i, w in enumerate(words * 3))                 # 48 words, > 255 characters
    assert len(q) > 255
    docs = [(0, q), (1, "alpha0 bravo1 something else"), (2, "unrelated text")]
    o = O.OracleEngine.create_default(); o.index(docs); o.set_trace(True)
    r = o.search(q, 10)
    assert r["keys"][0] == 0
    tids, tbase, tsc, tties, tfeat = o.last_trace()
    lcs_i = O.FEAT_NAMES.index("Lcs")
    seen = [int(tfeat[i, lcs_i]) for i in range(len(tids)) if int(tids[i]) == 0]
    assert seen == [len(q), 255], seen                                        # first evaluation: the whole query; second: the byte
