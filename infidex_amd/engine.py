"""ctypes binding of libinfidex_hip.so with the reference's public names for the hot path.

Mirrors (reference paths under src/Infidex): SearchEngine.cs (CreateDefault/CreateMinimal/IndexDocuments/Search),
Api/Query.cs, Api/Result.cs, Core/Document.cs, Api/Weight.cs, Core/ScoreEntry.cs.
"""
import ctypes as C
import os
from dataclasses import dataclass, field as _dc_field
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libinfidex_hip.so")

INFX_NFEAT = 32
STATUS = {0: "INFX_OK", 1: "INFX_EINVAL", 2: "INFX_ENOMEM", 3: "INFX_EHIP", 4: "INFX_ECAPACITY", 5: "INFX_EUNSUPPORTED"}


class InfidexError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class Weight:      # Api/Weight.cs:7-26
    High, Med, Low = 0, 1, 2


@dataclass
class Field:       # Api/Field.cs (name, value, weight)
    name: str
    value: str
    weight: int = Weight.Med


@dataclass
class Document:    # Core/Document.cs:76-88 (single text => one 'content' field, Weight.Med)
    document_key: int
    fields: Union[str, Sequence[Field]]

    def field_list(self) -> List[Field]:
        return [Field("content", self.fields, Weight.Med)] if isinstance(self.fields, str) else list(self.fields)


@dataclass
class Query:       # Api/Query.cs:9-40
    text: str
    max_number_of_records_to_return: int = 10
    coverage_depth: int = 500
    enable_coverage: bool = True
    filter: Optional[str] = None           # Infiscript expression (Filter.Parse), applied to the returned rows (ResultProcessor.ApplyFilter)
    enable_facets: bool = False


@dataclass
class ScoreEntry:  # Core/ScoreEntry.cs
    score: float
    document_id: int
    tiebreaker: int = 0


@dataclass
class Result:      # Api/Result.cs
    records: List[ScoreEntry] = _dc_field(default_factory=list)
    unsupported: bool = False
    used_coverage: bool = False
    stage1_fallback: bool = False
    skipped_candidates: bool = False       # a candidate document exceeded the Stage-2 envelope (INFX_MAX_DOC_TOKENS) and was left out
    facets: Optional[dict] = None          # field -> [(value, count)] (count desc, value asc), Api/Result.cs Facets
    total_in_filter: int = 0               # Filter.NumberOfDocumentsInFilter


class _Cfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("range_docs", C.c_int32), ("max_depth", C.c_int32), ("threads", C.c_int32),
                ("enable_coverage", C.c_int32), ("word_matcher", C.c_int32), ("stop_term_limit", C.c_int32),
                ("want_features", C.c_int32), ("no_exact_replay", C.c_int32)]


_lib = None


def load_library():
    """Loads the in-tree HIP extension; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        path = os.environ.get("INFX_LIB") or LIB_PATH       # INFX_LIB: the experiments build of the same library (A/B test of the k_accumulate designs)
        if not os.path.exists(path):
            raise InfidexError(3, f"{path} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.infx_engine_last_error.restype = C.c_char_p
        L.infx_last_error.restype = C.c_char_p
        L.infx_engine_wordmatcher.restype = C.c_int64
        L.infx_engine_last_stage2.restype = C.c_int64
        _lib = L
    return _lib


def _u16(s: str) -> np.ndarray:
    return np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def pack_texts(texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    arrs = [_u16(t) for t in texts]
    offs = np.zeros(len(arrs) + 1, np.uint64)
    if arrs:
        offs[1:] = np.cumsum([len(a) for a in arrs])
    arena = np.concatenate(arrs) if arrs and offs[-1] > 0 else np.zeros(1, np.uint16)
    return arena, offs


class SearchEngine:
    def __init__(self, enable_coverage=True, word_matcher=True, device: int = 0, range_docs: int = 0, max_depth: int = 500,
                 threads: int = 0, stop_term_limit: int = 0, want_features: bool = False, exact_replay: bool = True):
        self.L = load_library()
        cfg = _Cfg(device, range_docs, max_depth, threads, int(enable_coverage), int(word_matcher), stop_term_limit, int(want_features), int(not exact_replay))
        h = C.c_void_p()
        self._check(self.L.infx_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.device = device

    # SearchEngine.cs:78-94
    @classmethod
    def create_default(cls, **kw):
        return cls(enable_coverage=True, word_matcher=True, **kw)

    @classmethod
    def create_minimal(cls, **kw):
        return cls(enable_coverage=False, word_matcher=False, **kw)

    def _check(self, rc):
        if rc != 0:
            msg = self.L.infx_engine_last_error()
            raise InfidexError(rc, msg.decode("utf-8", "replace") if msg else "")

    def close(self):
        if getattr(self, "h", None):
            self.L.infx_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- indexing ----
    def index_documents(self, docs: Sequence[Document]):
        docs = list(docs)
        if not docs:
            return self.index_flat(None, np.zeros(1, np.uint16), np.zeros(1, np.uint64), (Weight.Med,))
        fl0 = docs[0].field_list()
        weights = [f.weight for f in fl0]
        texts = []
        for d in docs:
            fl = d.field_list()
            if [f.weight for f in fl] != weights:
                raise InfidexError(1, "all documents of one IndexDocuments call must share the field schema")
            texts.extend(str(f.value) for f in fl)
        arena, offs = pack_texts(texts)
        keys = np.asarray([d.document_key for d in docs], np.int64)
        return self.index_flat(keys, arena, offs, weights)

    def index_flat(self, keys, arena, offs, field_weights=(Weight.Med,)):
        fw = np.asarray(field_weights, np.int32)
        n = (len(offs) - 1) // len(fw)
        keys = None if keys is None else np.ascontiguousarray(keys, np.int64)
        self._keep = (keys, arena, offs)
        self._check(self.L.infx_engine_index_documents(self.h, C.c_int64(n), _p(keys, C.c_int64), _p(arena, C.c_uint16),
                                                       _p(offs, C.c_uint64), len(fw), _p(fw, C.c_int32)))
        self._keep = None

    def index_flat_from_segments(self, keys, arena, offs, field_weights, segment_paths, doc_bases):
        """An engine populated from flushed INFS segments + a live tail (VectorModel.Flush): the posting lists of the flushed document ranges come from the
        files, the documents supply everything else (infx_engine_index_from_segments); the corpus is then searched as one index."""
        fw = np.asarray(field_weights, np.int32)
        n = (len(offs) - 1) // len(fw)
        keys = None if keys is None else np.ascontiguousarray(keys, np.int64)
        paths = (C.c_char_p * len(segment_paths))(*[str(p).encode() for p in segment_paths])
        bases = np.ascontiguousarray(doc_bases, np.int32)
        self._check(self.L.infx_engine_index_from_segments(self.h, C.c_int64(n), _p(keys, C.c_int64), _p(arena, C.c_uint16), _p(offs, C.c_uint64), len(fw), _p(fw, C.c_int32),
                                                            len(segment_paths), paths, _p(bases, C.c_int32)))

    def load_index(self, path: str):
        """SearchEngine.Load (SearchEngine.cs:399-441) of an INFDX2 file: indexes the stored documents and verifies every stored term / posting against
        the index just built (infx_engine_load_index).  Returns (documents, stored terms compared, stored postings compared)."""
        c = np.zeros(3, np.int64)
        self._check(self.L.infx_engine_load_index(self.h, str(path).encode(), _p(c, C.c_int64)))
        return int(c[0]), int(c[1]), int(c[2])

    # ---- one host-index build per node (document shards; include/infidex_engine.h) ----
    def set_build_threads(self, threads: int):
        """Threads of the index build only (a node's leader rank builds with every core while planning keeps the rank's share)."""
        self._check(self.L.infx_engine_set_build_threads(self.h, int(threads)))

    def save_host_index(self, path: str):
        """Writes the host index (dictionaries, postings, WordMatcher lists, texts) to a node-local file for the node's other ranks."""
        self._check(self.L.infx_engine_save_host_index(self.h, str(path).encode()))

    def index_from_host_cache(self, path: str):
        """Instead of index_flat / index_documents: reads the host index a leader rank saved and uploads this rank's shard."""
        self._check(self.L.infx_engine_index_from_host_cache(self.h, str(path).encode()))

    # ---- Document.Deleted (DocumentCollection.DeleteDocumentsByKey, Core/DocumentCollection.cs:200-212) ----
    def delete_documents(self, keys) -> int:
        """Marks every document with one of these DocumentKeys as deleted (index statistics are not rebuilt, as in the reference until the next
        re-index); returns how many documents were newly marked.  Exclusive: no search may be in flight."""
        k = np.ascontiguousarray(list(keys) if not isinstance(keys, np.ndarray) else keys, np.int64)
        marked = C.c_int64(0)
        self._check(self.L.infx_engine_delete_documents(self.h, _p(k, C.c_int64), C.c_int64(len(k)), C.byref(marked)))
        return int(marked.value)

    def shard_info(self):
        """(first internal id, number of documents) of the doc range this engine's GPU holds (the whole corpus when unsharded)."""
        b = C.c_int32(0); n = C.c_int32(0)
        self._check(self.L.infx_engine_shard_info(self.h, C.byref(b), C.byref(n)))
        return int(b.value), int(n.value)

    def restore_documents(self):
        """Clears every Deleted flag."""
        self._check(self.L.infx_engine_restore_documents(self.h))

    # ---- non-indexed document fields (DocumentFields) as columns: filterable / facetable (config 5) ----
    def set_column(self, name, values, facetable=False):
        """One value per indexed document, in indexing order: int64 / float64 numpy array or a sequence of str."""
        n = len(values)
        if isinstance(values, np.ndarray) and values.dtype.kind in "iu":
            v = np.ascontiguousarray(values, np.int64)
            self._check(self.L.infx_engine_add_column(self.h, name.encode(), 1, int(facetable), C.c_int64(n), _p(v, C.c_int64), None, None, None))
        elif isinstance(values, np.ndarray) and values.dtype.kind == "f":
            v = np.ascontiguousarray(values, np.float64)
            self._check(self.L.infx_engine_add_column(self.h, name.encode(), 2, int(facetable), C.c_int64(n), None, _p(v, C.c_double), None, None))
        else:
            bs = [str(x).encode() for x in values]
            offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(b) for b in bs])
            arena = b"".join(bs) + b"\0"
            self._check(self.L.infx_engine_add_column(self.h, name.encode(), 3, int(facetable), C.c_int64(n), None, None, C.c_char_p(arena), _p(offs, C.c_uint64)))

    def facets_of(self, sh, nq, i):
        """Facets of query i of the last batch (nq queries) searched on session handle sh with Query.EnableFacets: {field: [(value, count)]}, counts over
        the returned rows, (count desc, value asc) — Core/FacetBuilder.cs:19-105."""
        facets = {}
        for k in range(self.L.infx_engine_facet_column_count(sh)):
            col = C.c_int32(0); codes = np.zeros(128, np.uint32); cnts = np.zeros(128, np.uint32)
            m = self.L.infx_engine_last_facets(sh, nq, i, k, C.byref(col), _p(codes, C.c_uint32), _p(cnts, C.c_uint32), 128)
            if m < 0:
                self._check(1)
            if m > 0:
                nb = C.create_string_buffer(256); self.L.infx_engine_column_info(self.h, col.value, nb, 256, None, None)
                vals = []
                for j in range(m):
                    vb = C.create_string_buffer(1024); self.L.infx_engine_column_value(self.h, col.value, int(codes[j]), vb, 1024)
                    vals.append((vb.value.decode(), int(cnts[j])))
                facets[nb.value.decode()] = vals
        return facets

    def _default_session(self):
        h = C.c_void_p(); self._check(self.L.infx_engine_default_session(self.h, C.byref(h))); return h

    def search_filtered(self, texts: Sequence[str], max_results=10, depth=500, enable_coverage=True, filter=None, enable_facets=False, session=None):
        """Search(Query) with Query.Filter / Query.EnableFacets for a batch sharing one filter: post-filter and facet counts run on the device."""
        sh = session.h if session is not None else self._default_session()
        nin = C.c_uint32(0)
        self._check(self.L.infx_engine_set_filter(sh, filter.encode() if filter is not None else None, int(enable_facets), C.byref(nin)))
        try:
            arena, offs = pack_texts(texts)
            keys, scores, ties, counts, flags = (session or self).search_packed(arena, offs, max_results, depth, enable_coverage)
            nq = len(texts)
            out = []
            for i in range(nq):
                recs = [ScoreEntry(float(scores[i, k]), int(keys[i, k]), int(ties[i, k])) for k in range(int(counts[i]))]
                facets = self.facets_of(sh, nq, i) if enable_facets else None
                out.append(Result(recs, bool(flags[i] & 1), bool(flags[i] & 2), bool(flags[i] & 4), bool(flags[i] & 8), facets, int(nin.value)))
            return out
        finally:
            self.L.infx_engine_set_filter(sh, None, 0, None)

    # ---- search ----
    def search(self, query: Union[Query, str], max_results: Optional[int] = None) -> Result:
        q = query if isinstance(query, Query) else Query(query, max_results or 10)
        if q.filter is not None or q.enable_facets:
            return self.search_filtered([q.text], q.max_number_of_records_to_return, q.coverage_depth, q.enable_coverage, q.filter, q.enable_facets)[0]
        return self.search_batch([q.text], q.max_number_of_records_to_return, q.coverage_depth, q.enable_coverage)[0]

    def search_batch_raw(self, texts: Sequence[str], max_results=10, depth=500, enable_coverage=True):
        arena, offs = pack_texts(texts)
        return self.search_packed(arena, offs, max_results, depth, enable_coverage)

    def search_packed(self, arena, offs, max_results=10, depth=500, enable_coverage=True):
        nq = len(offs) - 1
        keys = np.full((nq, max_results), -1, np.int64); scores = np.zeros((nq, max_results), np.float32)
        ties = np.zeros((nq, max_results), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
        self._check(self.L.infx_engine_search_batch(self.h, nq, _p(arena, C.c_uint16), _p(offs, C.c_uint64), max_results, depth,
                                                    int(enable_coverage), _p(keys, C.c_int64), _p(scores, C.c_float),
                                                    _p(ties, C.c_uint8), _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
        return keys, scores, ties, counts, flags

    def search_batch(self, texts: Sequence[str], max_results=10, depth=500, enable_coverage=True) -> List[Result]:
        keys, scores, ties, counts, flags = self.search_batch_raw(texts, max_results, depth, enable_coverage)
        out = []
        for i in range(len(texts)):
            recs = [ScoreEntry(float(scores[i, k]), int(keys[i, k]), int(ties[i, k])) for k in range(int(counts[i]))]
            out.append(Result(recs, bool(flags[i] & 1), bool(flags[i] & 2), bool(flags[i] & 4), bool(flags[i] & 8)))
        return out

    def last_timings(self):
        host = np.zeros(5, np.float64); kern = np.zeros(5, np.float32); alg = np.zeros(6, np.uint64)
        self._check(self.L.infx_engine_last_timings(self.h, _p(host, C.c_double), _p(kern, C.c_float), _p(alg, C.c_uint64)))
        return {"plan_ms": host[0], "stage1_ms": host[1], "prep2_ms": host[2], "stage2_ms": host[3], "post_ms": host[4],
                "k_accumulate_ms": float(kern[0]), "k_select_ms": float(kern[1]), "k_stage2_ms": float(kern[2]),
                "k_prep2_ms": float(kern[3]), "k_finalize_ms": float(kern[4]),
                "alg_bytes": int(alg[0]), "stage2_candidates": int(alg[1]), "stage2_text_bytes": int(alg[2]),
                "streamed_bytes": int(alg[3]), "stage1_candidates": int(alg[4]), "exact_replays": int(alg[5])}

    # ---- introspection (parity tests) ----
    def index_stats(self):
        n = C.c_int64(); t = C.c_int64(); p = C.c_int64(); a = C.c_float()
        self._check(self.L.infx_engine_index_stats(self.h, C.byref(n), C.byref(t), C.byref(p), C.byref(a)))
        return {"docs": n.value, "terms": t.value, "postings": p.value, "avgdl": a.value}

    def export_index(self):
        s = self.index_stats()
        T, P, N = s["terms"], s["postings"], s["docs"]
        df = np.zeros(T, np.int32); off = np.zeros(T + 1, np.uint64); pd = np.zeros(max(P, 1), np.int32)
        pw = np.zeros(max(P, 1), np.uint8); dl = np.zeros(max(N, 1), np.float32)
        self._check(self.L.infx_engine_export_index(self.h, _p(df, C.c_int32), _p(off, C.c_uint64), _p(pd, C.c_int32), _p(pw, C.c_uint8), _p(dl, C.c_float)))
        return {"df": df, "post_off": off, "post_doc": pd[:P], "post_w": pw[:P], "doc_len": dl[:N], "avgdl": s["avgdl"]}

    def term_text(self, t):
        buf = np.zeros(256, np.uint16)
        n = self.L.infx_engine_term_text(self.h, int(t), _p(buf, C.c_uint16), 256)
        return buf[:max(n, 0)].tobytes().decode("utf-16-le", errors="surrogatepass")

    def add_synonym(self, a, b):
        """SynonymMap.AddSynonym — before index_documents."""
        ua, ub = _u16(a), _u16(b)
        self._check(self.L.infx_engine_add_synonym(self.h, _p(ua, C.c_uint16), len(ua), _p(ub, C.c_uint16), len(ub)))

    def match_ld1_forward(self, q, cap=1024):
        a = _u16(q); out = np.zeros(cap, np.int32)
        c = self.L.infx_engine_match_ld1_forward(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), cap)
        return c, out[:min(c, cap)].copy()

    def match_ld1(self, q, cap=1024):
        a = _u16(q); out = np.zeros(cap, np.int32)
        c = self.L.infx_engine_match_ld1(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), cap)
        return c, out[:min(c, cap)].copy()

    def plan(self, text, depth=500, cap=256):
        a = _u16(text)
        t = np.zeros(cap, np.int32); df = np.zeros(cap, np.int32); idf = np.zeros(cap, np.float32)
        roles = np.zeros(cap, np.uint8); ranks = np.zeros(cap, np.uint8); meta = np.zeros(5, np.int32); flags = C.c_int32(0)
        n = self.L.infx_engine_plan(self.h, _p(a, C.c_uint16), len(a), depth, _p(t, C.c_int32), _p(df, C.c_int32), _p(idf, C.c_float),
                                    _p(roles, C.c_uint8), _p(ranks, C.c_uint8), cap, _p(meta, C.c_int32), C.byref(flags))
        return {"term_ids": t[:n].copy(), "df": df[:n].copy(), "idf": idf[:n].copy(), "roles": roles[:n].copy(), "ranks": ranks[:n].copy(),
                "mode": int(meta[0]), "prefix_set": int(meta[1]), "n_and": int(meta[2]), "df_s1": int(meta[3]), "df_s2": int(meta[4]),
                "flags": flags.value}

    def fuzzy_cache_size(self) -> int:
        """Entries of the LD1 expansion cache (LRU, at most 1000 like the reference's)."""
        self.L.infx_engine_fuzzy_cache_size.restype = C.c_int64
        return int(self.L.infx_engine_fuzzy_cache_size(self.h))

    def wordmatcher(self, text, cap=1 << 22):
        a = _u16(text); out = np.zeros(cap, np.int32)
        n = self.L.infx_engine_wordmatcher(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), C.c_int64(cap))
        return out[:min(n, cap)].copy()

    # ---- the planning lookups as the device answers them (infidex_engine.h; tests/test_gpu_lookups.py) ----
    def device_lookups(self) -> bool:
        return self.L.infx_engine_device_lookups(self.h) == 1

    def lookup_stats(self):
        out = np.zeros(4, np.int64)
        self._check(self.L.infx_engine_lookup_stats(self.h, _p(out, C.c_int64)))
        return dict(ld1_device=int(out[0]), ld1_host=int(out[1]), wm_device=int(out[2]), wm_host=int(out[3]))

    def match_ld1_device(self, q, cap=1024):
        """(count, first `cap` term ids) like match_ld1; count -1 / -2: the kernel handed the word back to the host walk."""
        a = _u16(q); out = np.zeros(cap, np.int32)
        c = self.L.infx_engine_match_ld1_device(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), cap)
        if c <= -100:
            self._check(-100 - c if c < -100 else 1)
        return c, out[:max(0, min(c, cap))].copy()

    def wordmatcher_device(self, text, cap=1 << 22):
        """Ascending unique doc ids of the lists k_wm emits for the (already prepared) search text; None: not admissible for the device lookup."""
        a = _u16(text); out = np.zeros(cap, np.int32)
        self.L.infx_engine_wordmatcher_device.restype = C.c_int64
        n = self.L.infx_engine_wordmatcher_device(self.h, _p(a, C.c_uint16), len(a), _p(out, C.c_int32), C.c_int64(cap))
        if n == -2:
            return None
        if n < 0:
            self._check(1)
        return out[:min(n, cap)].copy()

    def prefix_pop(self, p):
        a = _u16(p)
        return int(self.L.infx_engine_prefix_pop(self.h, _p(a, C.c_uint16), len(a)))

    def set_introspection(self, on=True):
        """Parity tooling: keep the Stage-1 rows / Stage-2 candidates of the last batch for last_stage1() / last_stage2()."""
        self._check(self.L.infx_engine_set_introspection(self.h, int(bool(on))))

    def last_stage1(self, qi, cap=4096):
        keys = np.zeros(cap, np.int64); sc = np.zeros(cap, np.float32)
        n = self.L.infx_engine_last_stage1(self.h, qi, _p(keys, C.c_int64), _p(sc, C.c_float), cap)
        n = max(n, 0)
        return keys[:n].copy(), sc[:n].copy()

    def last_stage2(self, cap=1 << 20):
        qo = np.zeros(cap, np.uint32); docs = np.zeros(cap, np.int32); base = np.zeros(cap, np.float32); sc = np.zeros(cap, np.float32)
        ties = np.zeros(cap, np.uint8); feat = np.zeros((cap, INFX_NFEAT), np.int32)
        n = self.L.infx_engine_last_stage2(self.h, _p(qo, C.c_uint32), _p(docs, C.c_int32), _p(base, C.c_float), _p(sc, C.c_float),
                                           _p(ties, C.c_uint8), _p(feat, C.c_int32), C.c_int64(cap))
        n = min(int(n), cap)
        return qo[:n].copy(), docs[:n].copy(), base[:n].copy(), sc[:n].copy(), ties[:n].copy(), feat[:n].copy()


class Session:
    """One in-flight batch (own HIP stream + scratch) on a SearchEngine; use one per host thread to overlap the host-side
    preparation of a batch with the GPU stages of another (infidex_engine.h)."""

    def __init__(self, engine: "SearchEngine"):
        self.engine = engine
        self.L = engine.L
        h = C.c_void_p()
        engine._check(self.L.infx_engine_session_create(engine.h, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.infx_engine_session_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_packed(self, arena, offs, max_results=10, depth=500, enable_coverage=True):
        nq = len(offs) - 1
        keys = np.full((nq, max_results), -1, np.int64); scores = np.zeros((nq, max_results), np.float32)
        ties = np.zeros((nq, max_results), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
        self.engine._check(self.L.infx_engine_session_search_batch(self.h, nq, _p(arena, C.c_uint16), _p(offs, C.c_uint64), max_results, depth,
                                                                   int(enable_coverage), _p(keys, C.c_int64), _p(scores, C.c_float),
                                                                   _p(ties, C.c_uint8), _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
        return keys, scores, ties, counts, flags

    def set_filter(self, expr=None, enable_facets=False):
        """Installs Query.Filter / Query.EnableFacets on this session (None clears); returns Filter.NumberOfDocumentsInFilter."""
        nin = C.c_uint32(0)
        self.engine._check(self.L.infx_engine_set_filter(self.h, expr.encode() if expr is not None else None, int(enable_facets), C.byref(nin)))
        return int(nin.value)

    def last_timings(self, kernels=True):
        """Host phase times of the session's last batch; kernels=True adds the kernel durations (HIP events on the session's stream).  Resolving those costs a
        handful of HIP API calls, which queue behind the launches of other sessions: a throughput run asks for them on a sample of its batches only."""
        host = np.zeros(5, np.float64); kern = np.zeros(5, np.float32); alg = np.zeros(6, np.uint64)
        self.engine._check(self.L.infx_engine_session_last_timings(self.h, _p(host, C.c_double), _p(kern, C.c_float) if kernels else None, _p(alg, C.c_uint64)))
        pb = np.zeros(4, np.float64)
        self.engine._check(self.L.infx_engine_session_plan_breakdown(self.h, _p(pb, C.c_double)))
        out = {"plan_ms": host[0], "stage1_ms": host[1], "prep2_ms": host[2], "stage2_ms": host[3], "post_ms": host[4],
               "plan_tokens_ms": float(pb[0]), "plan_ld1_device_ms": float(pb[1]), "plan_union_device_ms": float(pb[2]), "plan_finish_ms": float(pb[3]),
               "alg_bytes": int(alg[0]), "stage2_candidates": int(alg[1]), "stage2_text_bytes": int(alg[2]),
               "streamed_bytes": int(alg[3]), "stage1_candidates": int(alg[4]), "exact_replays": int(alg[5])}
        if kernels:
            rms = C.c_float(0); why = np.zeros(3, np.uint32); parts = np.zeros(4, np.float32)
            self.engine._check(self.L.infx_engine_session_last_replay(self.h, C.byref(rms), _p(why, C.c_uint32)))
            self.engine._check(self.L.infx_engine_session_replay_breakdown(self.h, _p(parts, C.c_float)))
            out.update({"k_replay_ms": float(rms.value), "flag_plateau": int(why[0]), "flag_band": int(why[1]), "flag_unknown": int(why[2]),
                        "k_ex_scan_ms": float(parts[0]), "k_ex_chunk_ms": float(parts[1]), "k_ex_heap_ms": float(parts[2]), "k_exact1_ms": float(parts[3]),
                        "k_accumulate_ms": float(kern[0]), "k_select_ms": float(kern[1]), "k_stage2_ms": float(kern[2]),
                        "k_prep2_ms": float(kern[3]), "k_finalize_ms": float(kern[4])})
        return out


def normalize(s, lower=False):
    L = load_library()
    x = _u16(s); out = np.zeros(len(x) + 8, np.uint16)
    n = L.infx_engine_normalize(_p(x, C.c_uint16), len(x), int(lower), _p(out, C.c_uint16), len(out))
    return out[:n].tobytes().decode("utf-16-le", errors="surrogatepass")
