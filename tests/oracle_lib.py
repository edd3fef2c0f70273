"""ctypes wrapper over oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference algorithm (oracle/README.md). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

_lib = None


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "_build/liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_finalize.argtypes = [C.c_void_p]
        L.orc_set_trace.argtypes = [C.c_void_p, C.c_int]
        L.orc_avgdl.restype = C.c_float
        L.orc_avgdl.argtypes = [C.c_void_p]
        for f in ("orc_num_docs", "orc_num_terms", "orc_num_postings"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_timed_batch.restype = C.c_double
        _lib = L
    return _lib


def u16(s):
    """Python str -> (uint16 numpy array) of UTF-16 code units."""
    return np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


FEAT_NAMES = ["CoverageScore", "TermsCount", "TermsWithAnyMatch", "TermsFullyMatched", "TermsStrictMatched",
              "TermsPrefixMatched", "FirstMatchIndex", "WordHits", "DocTokenCount", "LongestPrefixRun",
              "SuffixPrefixRun", "PhraseSpan", "PrecedingStrictCount", "LastTokenHasPrefix", "LastTermIsTypeAhead",
              "UnfilteredQueryTokenCount", "LexicalPrefixLast", "AllPrecedingExact", "IsPerfectDocLexical",
              "HasStemEvidence", "HasAnchorStem", "TrailingMatchDensity", "SingleTermLexicalSim",
              "SingleCharLastTokenBoost", "Lcs", "SumCi_bits", "IdfCoverage_bits", "TotalIdf_bits", "MissingIdf_bits",
              "LastTermCi_bits", "WeightedCoverage_bits", "_pad"]
NFEAT = 32
N_INT_FEAT = 25   # features [0, 25) are the integer "coverage counts" that must match bit-exactly

HIGH, MED, LOW = 0, 1, 2


class OracleEngine:
    """Mirrors the reference's SearchEngine.CreateDefault() / CreateMinimal() for the hot path."""

    def __init__(self, enable_coverage=True, word_matcher=True, stop_term_limit=0):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create(int(enable_coverage), int(word_matcher), int(stop_term_limit)))

    @classmethod
    def create_default(cls):
        return cls(True, True)

    @classmethod
    def create_minimal(cls):
        return cls(False, False)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def add_synonym(self, a, b):
        """SynonymMap.AddSynonym — before indexing."""
        ua, ub = u16(a), u16(b)
        self.L.orc_add_synonym(self.h, _p(ua, C.c_uint16), len(ua), _p(ub, C.c_uint16), len(ub))

    def add(self, key, text_or_fields):
        """text_or_fields: str (single 'content' field, Weight.Med) or list of (text, weight)."""
        fields = [(text_or_fields, MED)] if isinstance(text_or_fields, str) else list(text_or_fields)
        arrs = [u16(t) for t, _ in fields]
        ptrs = (C.POINTER(C.c_uint16) * len(arrs))(*[_p(a, C.c_uint16) for a in arrs])
        lens = (C.c_int32 * len(arrs))(*[len(a) for a in arrs])
        ws = (C.c_int32 * len(arrs))(*[w for _, w in fields])
        self.L.orc_add_document(self.h, C.c_int64(key), len(arrs), ptrs, lens, ws)

    def add_flat(self, keys, arena, offs, field_weights=(MED,)):
        fw = np.asarray(field_weights, dtype=np.int32)
        keys = None if keys is None else np.ascontiguousarray(keys, dtype=np.int64)
        n = (len(offs) - 1) // len(fw)
        self.L.orc_add_documents_flat(self.h, C.c_int64(n), _p(keys, C.c_int64), _p(arena, C.c_uint16),
                                      _p(offs, C.c_uint64), len(fw), _p(fw, C.c_int32))

    def index(self, docs):
        for k, t in docs:
            self.add(k, t)
        self.finalize()

    def finalize(self):
        self.L.orc_finalize(self.h)

    def set_trace(self, on=True):
        self.L.orc_set_trace(self.h, int(on))

    def search(self, text, max_results=10, depth=500, enable_coverage=True):
        q = u16(text)
        cap = max(max_results, 1)
        keys = np.zeros(cap, np.int64); scores = np.zeros(cap, np.float32); ties = np.zeros(cap, np.uint8)
        flags = C.c_int32(0)
        n = self.L.orc_search(self.h, _p(q, C.c_uint16), len(q), max_results, depth, int(enable_coverage),
                              _p(keys, C.c_int64), _p(scores, C.c_float), _p(ties, C.c_uint8), cap, C.byref(flags))
        return {"keys": keys[:n].tolist(), "scores": scores[:n].copy(), "ties": ties[:n].copy(),
                "unsupported": bool(flags.value & 1), "used_coverage": bool(flags.value & 2)}

    def last_stage1(self, cap=4096):
        keys = np.zeros(cap, np.int64); scores = np.zeros(cap, np.float32)
        n = self.L.orc_last_stage1(self.h, _p(keys, C.c_int64), _p(scores, C.c_float), cap)
        return keys[:n].copy(), scores[:n].copy()

    def last_trace(self, cap=8192):
        ids = np.zeros(cap, np.int32); base = np.zeros(cap, np.float32); sc = np.zeros(cap, np.float32)
        ties = np.zeros(cap, np.uint8); feat = np.zeros((cap, NFEAT), np.int32)
        n = self.L.orc_last_trace(self.h, _p(ids, C.c_int32), _p(base, C.c_float), _p(sc, C.c_float), _p(ties, C.c_uint8),
                                  _p(feat, C.c_int32), cap)
        return ids[:n].copy(), base[:n].copy(), sc[:n].copy(), ties[:n].copy(), feat[:n].copy()

    def last_terms(self, cap=256):
        t = np.zeros(cap, np.int32); df = np.zeros(cap, np.int32); idf = np.zeros(cap, np.float32); mx = np.zeros(cap, np.float32)
        n = self.L.orc_last_terms(self.h, _p(t, C.c_int32), _p(df, C.c_int32), _p(idf, C.c_float), _p(mx, C.c_float), cap)
        return t[:n].copy(), df[:n].copy(), idf[:n].copy(), mx[:n].copy()

    def last_stats(self):
        o = np.zeros(3, np.int64)
        self.L.orc_last_stats(self.h, _p(o, C.c_int64))
        return {"candidates": int(o[0]), "postings_touched": int(o[1]), "mode": int(o[2])}

    # ---- index introspection ----
    @property
    def num_docs(self):
        return int(self.L.orc_num_docs(self.h))

    @property
    def num_terms(self):
        return int(self.L.orc_num_terms(self.h))

    @property
    def avgdl(self):
        return float(self.L.orc_avgdl(self.h))

    def export_index(self):
        T = self.num_terms; P = int(self.L.orc_num_postings(self.h)); N = self.num_docs
        df = np.zeros(T, np.int32); off = np.zeros(T + 1, np.uint64); pd = np.zeros(P, np.int32)
        pw = np.zeros(P, np.uint8); dl = np.zeros(N, np.float32)
        self.L.orc_export_index(self.h, _p(df, C.c_int32), _p(off, C.c_uint64), _p(pd, C.c_int32), _p(pw, C.c_uint8), _p(dl, C.c_float))
        return {"df": df, "post_off": off, "post_doc": pd, "post_w": pw, "doc_len": dl}

    def term_text(self, t):
        buf = np.zeros(256, np.uint16)
        n = self.L.orc_term_text(self.h, int(t), _p(buf, C.c_uint16), 256)
        return buf[:n].tobytes().decode("utf-16-le", errors="surrogatepass")

    def term_id(self, s):
        a = u16(s)
        return int(self.L.orc_term_id(self.h, _p(a, C.c_uint16), len(a)))

    def prefix_docset(self, p, cap=1 << 20):
        a = u16(p); out = np.zeros(cap, np.int32)
        n = self.L.orc_prefix_docset(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), cap)
        return out[:min(n, cap)].copy()

    def match_ld1(self, q, cap=1024):
        a = u16(q); out = np.zeros(cap, np.int32)
        c = self.L.orc_match_ld1(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), cap)
        return c, out[:min(c, cap)].copy()

    def wordmatcher(self, q, cap=1 << 22):
        a = u16(q); out = np.zeros(cap, np.int32)
        n = self.L.orc_wordmatcher(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), cap)
        return out[:min(n, cap)].copy()

    def wm_lookup(self, word, affix=False, cap=1 << 20):
        a = u16(word); out = np.zeros(cap, np.int32)
        n = self.L.orc_wm_lookup(self.h, _p(a, C.c_uint16), len(a), int(affix), _p(out, C.c_int32), cap)
        return None if n < 0 else out[:min(n, cap)].copy()

    def timed_batch(self, queries, max_results=10, depth=500, threads=1, want_latency=False):
        arrs = [u16(q) for q in queries]
        offs = np.zeros(len(arrs) + 1, np.uint64)
        offs[1:] = np.cumsum([len(a) for a in arrs])
        arena = np.concatenate(arrs) if arrs else np.zeros(0, np.uint16)
        keys = np.full((len(arrs), max_results), -1, np.int64)
        lat = np.zeros(len(arrs), np.float64) if want_latency else None
        self.L.orc_timed_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint64), C.c_int32,
                                           C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        secs = self.L.orc_timed_batch(self.h, len(arrs), _p(arena, C.c_uint16), _p(offs, C.c_uint64), max_results, depth,
                                      threads, _p(keys, C.c_int64), _p(lat, C.c_double))
        return secs, keys, lat


def stage_times(reset=True):
    """Milliseconds the oracle spent per stage, summed over its threads, since the last reset: planning, candidate selection, BM25+ scoring, WordMatcher, coverage."""
    out = np.zeros(5, np.float64)
    lib().orc_stage_times(_p(out, C.c_double), int(reset))
    return dict(zip(("planning", "candidate_selection", "bm25_scoring", "wordmatcher", "coverage"), out.tolist()))


def levenshtein(a, b, max_err=2**31 - 1, ignore_case=False):
    x, y = u16(a), u16(b)
    return lib().orc_levenshtein(_p(x, C.c_uint16), len(x), _p(y, C.c_uint16), len(y), max_err, int(ignore_case))


def damerau(a, b, max_d, ignore_case=False):
    x, y = u16(a), u16(b)
    return lib().orc_damerau(_p(x, C.c_uint16), len(x), _p(y, C.c_uint16), len(y), max_d, int(ignore_case))


def lcs(a, b, tol):
    x, y = u16(a), u16(b)
    return lib().orc_lcs(_p(x, C.c_uint16), len(x), _p(y, C.c_uint16), len(y), tol)


def case_tables():
    """char.ToLowerInvariant / ToUpperInvariant / IsLetter as the oracle applies them (oracle/unicode_tables.hpp), 65536 entries each."""
    lo = np.zeros(65536, np.uint16); up = np.zeros(65536, np.uint16); le = np.zeros(65536, np.uint8)
    lib().orc_case_tables(_p(lo, C.c_uint16), _p(up, C.c_uint16), _p(le, C.c_uint8))
    return {"lower": lo, "upper": up, "letter": le}


def normalize(s, lower=False):
    x = u16(s); out = np.zeros(len(x) + 8, np.uint16)
    n = lib().orc_normalize(_p(x, C.c_uint16), len(x), int(lower), _p(out, C.c_uint16), len(out))
    return out[:n].tobytes().decode("utf-16-le", errors="surrogatepass")


def coverage_standalone(query, doc, lcs_sum=0.0, bm25=0.0, word_idf=None):
    q, d = u16(query), u16(doc)
    feat = np.zeros(NFEAT, np.int32); score = C.c_float(0); tie = C.c_uint8(0)
    L = lib()
    L.orc_coverage_standalone.argtypes = [C.POINTER(C.c_uint16), C.c_int32, C.POINTER(C.c_uint16), C.c_int32, C.c_double, C.c_float,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                          C.c_int32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    words = list((word_idf or {}).items())
    warrs = [u16(w) for w, _ in words]
    woffs = np.zeros(len(warrs) + 1, np.uint64)
    if warrs:
        woffs[1:] = np.cumsum([len(a) for a in warrs])
    warena = np.concatenate(warrs) if warrs else np.zeros(1, np.uint16)
    wvals = np.asarray([v for _, v in words] or [0.0], np.float32)
    cov = L.orc_coverage_standalone(_p(q, C.c_uint16), len(q), _p(d, C.c_uint16), len(d), lcs_sum, bm25,
                                    _p(feat, C.c_int32), C.byref(score), C.byref(tie),
                                    len(words), _p(warena, C.c_uint16), _p(woffs, C.c_uint64), _p(wvals, C.c_float))
    return cov, dict(zip(FEAT_NAMES, feat.tolist())), score.value, tie.value


# ---- Infiscript filter / facets (config 5) ------------------------------------------------------------------------------
def filter_eval(expr, fields):
    """fields: dict name -> None | int | float | str.  Returns True / False; raises ValueError on a parse error, NotImplementedError for
    a construct the oracle does not restate (MATCHES)."""
    L = lib()
    names = list(fields.keys())
    kinds = (C.c_int32 * len(names))(); ints = (C.c_int64 * len(names))(); dbls = (C.c_double * len(names))()
    cn = (C.c_char_p * len(names))(*[n.encode() for n in names]); cs = (C.c_char_p * len(names))()
    for i, n in enumerate(names):
        v = fields[n]
        if v is None: kinds[i] = 0
        elif isinstance(v, bool): kinds[i] = 3; cs[i] = (b"True" if v else b"False")
        elif isinstance(v, int): kinds[i] = 1; ints[i] = v
        elif isinstance(v, float): kinds[i] = 2; dbls[i] = v
        else: kinds[i] = 3; cs[i] = str(v).encode()
    L.orc_filter_eval.restype = C.c_int32
    r = L.orc_filter_eval(expr.encode(), len(names), cn, kinds, ints, dbls, cs)
    if r == -1: raise ValueError("FilterParseException: " + expr)
    if r == -2: raise NotImplementedError(expr)
    return bool(r)


def double_to_string(x):
    buf = C.create_string_buffer(64); lib().orc_double_to_string(C.c_double(x), buf, 64); return buf.value.decode()


def _oe_set_column(self, name, values, facetable=False):
    """A non-indexed document field per internal doc id: int64 / float64 numpy array or a list of str."""
    n = len(values)
    if isinstance(values, np.ndarray) and values.dtype.kind in "iu":
        v = np.ascontiguousarray(values, np.int64); self.L.orc_set_column(self.h, name.encode(), 1, int(facetable), C.c_int64(n), _p(v, C.c_int64), None, None, None)
    elif isinstance(values, np.ndarray) and values.dtype.kind == "f":
        v = np.ascontiguousarray(values, np.float64); self.L.orc_set_column(self.h, name.encode(), 2, int(facetable), C.c_int64(n), None, _p(v, C.c_double), None, None)
    else:
        bs = [str(x).encode() for x in values]; offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(b) for b in bs]); arena = b"".join(bs) + b"\0"
        self.L.orc_set_column(self.h, name.encode(), 3, int(facetable), C.c_int64(n), None, None, C.c_char_p(arena), _p(offs, C.c_uint64))


def _oe_search_filtered(self, text, max_results=10, depth=500, enable_coverage=True, filter=None, enable_facets=False):
    import json
    q = u16(text); cap = max(max_results, 1)
    keys = np.zeros(cap, np.int64); scores = np.zeros(cap, np.float32); ties = np.zeros(cap, np.uint8)
    flags = C.c_int32(0); nin = C.c_int32(0)
    self.L.orc_search_filtered.restype = C.c_int32
    n = self.L.orc_search_filtered(self.h, _p(q, C.c_uint16), len(q), max_results, depth, int(enable_coverage), filter.encode() if filter is not None else None,
                                   int(enable_facets), _p(keys, C.c_int64), _p(scores, C.c_float), _p(ties, C.c_uint8), cap, C.byref(flags), C.byref(nin))
    if n == -1: raise ValueError("FilterParseException: " + str(filter))
    if n == -2: raise NotImplementedError(str(filter))
    buf = C.create_string_buffer(1 << 16); self.L.orc_last_facets_json(self.h, buf, 1 << 16)
    return {"keys": keys[:n].tolist(), "scores": scores[:n].copy(), "ties": ties[:n].copy(), "used_coverage": bool(flags.value & 2),
            "in_filter": int(nin.value), "facets": {k: [(a, int(b)) for a, b in v] for k, v in json.loads(buf.value.decode() or "{}").items()}}


def _oe_delete_keys(self, keys):
    """Document.Deleted = true for every document with one of these DocumentKeys (index statistics untouched)."""
    k = np.ascontiguousarray(keys, np.int64); self.L.orc_delete_keys.restype = C.c_int32
    return int(self.L.orc_delete_keys(self.h, _p(k, C.c_int64), C.c_int64(len(k))))


def _oe_restore_all(self):
    self.L.orc_restore_all.argtypes = [C.c_void_p]; self.L.orc_restore_all.restype = None
    self.L.orc_restore_all(self.h)


OracleEngine.delete_keys = _oe_delete_keys
OracleEngine.restore_all = _oe_restore_all
OracleEngine.set_column = _oe_set_column
OracleEngine.search_filtered = _oe_search_filtered
