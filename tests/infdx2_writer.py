"""Writer of the reference's INFDX2 index file format (SearchEngine.Save -> IndexPersistence.Save, src/Infidex/Indexing/IndexPersistence.cs:33-99,
WriteDocuments :304-320, WriteTerms :355-383, checksums :268-299), restated for the tests of the product's reader (infx_engine_load_index).
TEST INFRASTRUCTURE.  No file written by the reference itself exists in its repository (and no .NET runtime here), so the reader's parity is pinned to
this restatement of the writer, not to a reference-produced file: PARITY UNPINNED for f4.

The derived sections are restated too, each from the text the reference derives it from:
  term FST            FstSerializer.Write of FstBuilder.Build (Fst/FstSerializer.cs:16-37, FstBuilder.cs:80-166): every term of the collection -> its index
  short-query index   PositionalPrefixIndex.IndexDocument / Write (ShortQuery/PositionalPrefixIndex.cs:56-118, 249-274) over lower(normalize(IndexedText))
  metadata cache      VectorModel.BuildDocumentMetadataCache (VectorModel.cs:251-312) + DocumentMetadataCache.Write over normalize(lower(IndexedText))
  WordMatcher         WordMatcher.Load / FinalizeIndex / Save (WordMatcher/WordMatcher.cs:80-193, 391-455) over normalize(lower(IndexedText)), document
                      sets in the portable Roaring format (Internalized/Roaring/RoaringArray.cs:396-469)."""
import struct
from collections import deque

# ConfigurationParameters.cs:58-62 (TokenizerSetup default delimiters)
DELIMS = set(" -/.,:;'`\u2013\u2014*&\\_(){}[]\t")


def _rotl7(c):
    return ((c << 7) | (c >> 25)) & 0xFFFFFFFF


def checksum_words(vals):
    c = 0x12345678
    for v in vals:
        c = _rotl7(c ^ (v & 0xFFFFFFFF))
    return c


def checksum_bytes(data: bytes):
    c = 0x12345678
    for i in range(0, len(data), 4):
        c = _rotl7(c ^ int.from_bytes(data[i:i + 4], "little"))
    return c


def _units(s: str):
    """UTF-16 code units of a Python string (which may hold half a surrogate pair as a code point of its own)."""
    raw = s.encode("utf-16-le", "surrogatepass")
    return struct.unpack("<%dH" % (len(raw) // 2), raw)


def _from_units(units) -> str:
    return struct.pack("<%dH" % len(units), *units).decode("utf-16-le", "surrogatepass")


def lossy(s: str) -> str:
    """What Encoding.UTF8 (replacement fallback) makes of a .NET string: half a surrogate pair on its own becomes U+FFFD."""
    u = _units(s); out = []; i = 0
    while i < len(u):
        c = u[i]
        if 0xD800 <= c <= 0xDBFF and i + 1 < len(u) and 0xDC00 <= u[i + 1] <= 0xDFFF:
            out += [c, u[i + 1]]; i += 2
        else:
            out.append(0xFFFD if 0xD800 <= c <= 0xDFFF else c); i += 1
    return struct.pack("<%dH" % len(out), *out).decode("utf-16-le")


def _string(s: str) -> bytes:        # BinaryWriter.Write(string): 7-bit encoded UTF-8 byte length + bytes
    b = lossy(s).encode("utf-8"); n = len(b); out = bytearray()
    while True:
        if n >= 0x80:
            out.append((n & 0x7F) | 0x80); n >>= 7
        else:
            out.append(n); break
    return bytes(out) + b


def write(path, docs, terms, derived=b"", trailer=b"\x00", flags=0b10011):
    """docs: [(id, key, indexed_text, deleted)], terms: [(text, df, [(doc, weight)])] — non-stop terms only.  `derived` stands for the FST, short-query
    index and metadata-cache sections (flags HasFst | HasShortQueryIndex | HasDocumentMetadataCache), `trailer` for the WordMatcher section."""
    data = bytearray(struct.pack("<i", len(docs)))
    for i, key, text, deleted in docs:
        data += struct.pack("<iq", i, key) + _string(text) + _string("") + struct.pack("<iiB", 0, 0, 1 if deleted else 0)
    data += struct.pack("<i", len(terms))
    for text, df, post in terms:
        data += _string(text) + struct.pack("<ii", df, len(post))
        for d, w in post:
            data += struct.pack("<iB", d, w)
    data += derived
    head = b"INFDX2" + struct.pack("<IIII", 2, flags, len(docs), len(terms)) + struct.pack("<I", checksum_words([2, flags, len(docs), len(terms)]))
    with open(path, "wb") as f:
        f.write(head + struct.pack("<I", len(data)) + bytes(data) + struct.pack("<I", checksum_bytes(bytes(data))) + trailer)


def words_of(text):
    """string.Split(delimiters, RemoveEmptyEntries) / the token loops of the short-query index and the metadata cache: maximal runs of non-delimiters."""
    out, cur = [], []
    for ch in text:
        if ch in DELIMS:
            if cur:
                out.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    if cur:
        out.append("".join(cur))
    return out


def _u16len(s):
    return len(s.encode("utf-16-le", "surrogatepass")) // 2


def _compact_trie(pairs, reverse=False):
    """FstBuilder.AddToTrie / AddReversed (the last output of a repeated word wins) + CompactTrie (BFS, children ordered by label).  Labels are UTF-16 units."""
    root = {"c": {}, "f": False, "o": -1}
    for w, o in pairs:
        units = _units(w)
        cur = root
        for u in (reversed(units) if reverse else units):
            cur = cur["c"].setdefault(u, {"c": {}, "f": False, "o": -1})
        cur["f"] = True; cur["o"] = o
    nodes, arcs = [], []
    index = {id(root): 0}; q = deque([root]); nxt = 1
    while q:
        b = q.popleft()
        nodes.append((len(arcs), len(b["c"]) & 0xFFFF, b["f"], b["o"]))
        for label in sorted(b["c"]):
            ch = b["c"][label]
            index[id(ch)] = nxt; nxt += 1; q.append(ch)
            arcs.append((label, index[id(ch)], ch["o"] if ch["f"] else -1, ch["f"]))
    return nodes, arcs


def fst_section(pairs, forward_only_count=None):
    """pairs: (text, output) in Add order; termCount = number of Add calls."""
    out = bytearray(struct.pack("<IHi", 0x46535432, 1, len(pairs) if forward_only_count is None else forward_only_count))
    for rev in (False, True):
        nodes, arcs = _compact_trie(pairs, rev)
        out += struct.pack("<i", len(nodes))
        for a, n, f, o in nodes:
            out += struct.pack("<iH?i", a, n, f, o)
        out += struct.pack("<i", len(arcs))
        for lb, tg, o, f in arcs:
            out += struct.pack("<Hii?", lb, tg, o, f)
        out += struct.pack("<i", 0)
    return bytes(out)


def short_query_section(index_texts):
    """index_texts[d] = lower(normalize(IndexedText)) of document d.  Postings (doc, (ushort) token index, wordStart = true), sorted by (doc, position);
    single characters in character order (the array walk), longer prefixes in first-insertion order (Dictionary enumeration)."""
    single, multi = {}, {}
    for d, text in enumerate(index_texts):
        for k, tok in enumerate(words_of(text)):
            units = _units(tok)
            for L in range(1, min(3, len(units)) + 1):
                key = units[:L]
                (single if L == 1 else multi).setdefault(key, []).append((d, k & 0xFFFF))
    def plist(ps):
        ps = sorted(ps)
        return struct.pack("<i", len(ps)) + b"".join(struct.pack("<iH?", d, k, True) for d, k in ps)
    out = bytearray(struct.pack("<i", len(single)))
    for key in sorted(single):
        out += struct.pack("<H", key[0]) + plist(single[key])
    out += struct.pack("<i", len(multi))
    for key, ps in multi.items():
        out += _string(_from_units(key)) + plist(ps)
    return bytes(out)


def metadata_section(meta_texts, deleted=()):
    """meta_texts[d] = normalize(lower(IndexedText)); deleted / empty documents hold DocumentMetadata.Empty."""
    out = bytearray(struct.pack("<i", len(meta_texts)))
    for d, text in enumerate(meta_texts):
        toks = [] if d in deleted else words_of(text)
        out += _string(toks[0] if toks else "") + struct.pack("<H", min(len(toks), 65535))
    return bytes(out)


def roaring(docs):
    """RoaringBitmap.Serialize of ascending ids: no run containers (cookie 12346), array containers up to 4096 values, bitmaps beyond."""
    groups = {}
    for v in docs:
        groups.setdefault(v >> 16, []).append(v & 0xFFFF)
    keys = sorted(groups)
    out = bytearray(struct.pack("<Ii", 12346, len(keys)))
    for k in keys:
        out += struct.pack("<HH", k, len(groups[k]) - 1)
    off = 4 + 4 + 8 * len(keys)
    for k in keys:
        out += struct.pack("<i", off); off += 8192 if len(groups[k]) > 4096 else 2 * len(groups[k])
    for k in keys:
        vals = groups[k]
        if len(vals) > 4096:
            bits = bytearray(8192)
            for v in vals:
                bits[v >> 3] |= 1 << (v & 7)
            out += bits
        else:
            out += struct.pack("<%dH" % len(vals), *vals)
    return bytes(out)


def wordmatcher_section(wm_texts, min_exact=2, max_exact=8, min_ld1=3, max_ld1=8):
    """wm_texts[d] = normalize(lower(IndexedText)).  bool present | exact dictionary | symmetric-delete dictionary | bool hasFst | FST | occurrence map."""
    exact, ld1, occ = {}, {}, []
    def add(index, key, d):
        docs = index.setdefault(key, [])
        if not docs or docs[-1] != d:
            docs.append(d)
    for d, text in enumerate(wm_texts):
        for w in words_of(text):
            n = _u16len(w)
            units = _units(w)
            if min_exact <= n <= max_exact:
                add(exact, w, d)
            if min_ld1 <= n <= max_ld1:
                for i in range(n):
                    v = units[:i] + units[i + 1:]
                    add(ld1, _from_units(v), d)
            if n >= min_ld1:
                occ.append((w, d))                 # _fstIndex is null while indexing: every occurrence gets a new id, the FST keeps the last (Q13)
    out = bytearray(b"\x01")
    for index in (exact, ld1):
        out += struct.pack("<i", len(index))
        for key, docs in index.items():
            blob = roaring(docs)
            out += _string(key) + struct.pack("<i", len(blob)) + blob
    out += b"\x01" + fst_section([(w, i) for i, (w, _) in enumerate(occ)])
    out += struct.pack("<i", len(occ))
    for i, (_, d) in enumerate(occ):
        blob = roaring([d])
        out += struct.pack("<ii", i, len(blob)) + blob
    return bytes(out)


def derived_sections(term_texts, indexed_texts, normalize, deleted=()):
    """(derived, trailer) for write(): term_texts in collection order (stop terms included), indexed_texts[d] = IndexedText, normalize(s, lower_after) = the
    oracle's TextNormalizer (+ ToLowerInvariant afterwards when asked)."""
    index_texts = [normalize(t, True) for t in indexed_texts]
    meta_texts = [normalize(t.lower(), False) for t in indexed_texts]
    derived = fst_section([(t, i) for i, t in enumerate(term_texts)]) + short_query_section(index_texts) + metadata_section(meta_texts, set(deleted))
    return derived, wordmatcher_section(meta_texts)
