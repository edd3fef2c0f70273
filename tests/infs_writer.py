"""Writer of the reference's INFS segment file format (SearchEngine.Flush -> VectorModel.Flush -> SegmentWriter.WriteSegment,
src/Infidex/Indexing/Segments/SegmentWriter.cs:13-94), restated for the tests of the product's reader (csrc/host/infs.h, infx_segment_*):
  postings    BlockPostingsWriter.Write   Indexing/Segments/BlockPostingsWriter.cs:24-161   (blocks of 64..256 postings, density rule :66-76, skip table :105-112)
              GroupVarInt.Write           Indexing/Compression/GroupVarInt.cs:56-115          (tag byte of four 2-bit lengths; a short last group writes fewer values)
  term index  FstBuilder.Build / CompactTrie  Indexing/Fst/FstBuilder.cs:80-166 (plain tries, BFS order, children sorted by label) + FstSerializer.Write Fst/FstSerializer.cs:16-37,69-111
  offsets     EliasFano.Encode / Write    Indexing/Compression/EliasFano.cs:29-102 with DArray.Build / Write (DArray.cs:36-122,196-208), CompactArray.Write (CompactArray.cs:103-119)
TEST INFRASTRUCTURE.  No file written by the reference itself exists in its repository (and no .NET runtime here), so the reader's parity is pinned to this
restatement of the writer and to the two known answers of SegmentTests.cs, not to a reference-produced file: PARITY UNPINNED for f2(a)."""
import struct
from collections import deque


def _byte_count(v):
    return 1 if v < (1 << 8) else 2 if v < (1 << 16) else 3 if v < (1 << 24) else 4


def group_varint(values):
    out = bytearray(); i = 0; n = len(values)
    while i < n:
        rem = n - i
        g = [values[i + k] if k < rem else 0 for k in range(4)]
        cnt = min(rem, 4)
        lens = [_byte_count(v) for v in g]
        out.append(((lens[0] - 1) << 6) | ((lens[1] - 1) << 4) | ((lens[2] - 1) << 2) | (lens[3] - 1))
        for k in range(cnt):
            out += int(g[k]).to_bytes(4, "little")[:lens[k]]
        i += cnt
    return bytes(out)


def block_postings(buf: bytearray, docs, weights):
    """Appends one posting list at the end of buf (absolute file offsets = len(buf) positions: buf IS the file so far)."""
    start = len(buf)
    buf += struct.pack("<iiq", 0, 0, 0)
    total = 0; blocks = []            # (min, max, offset, maxw, count)
    cur_d, cur_w = [], []

    def flush():
        off = len(buf)
        deltas = []; prev = 0
        for d in cur_d:
            deltas.append(d - prev); prev = d
        data = group_varint(deltas)
        buf.extend(struct.pack("<i", len(data))); buf.extend(data); buf.extend(bytes(cur_w))
        blocks.append((cur_d[0], cur_d[-1], off, max(cur_w), len(cur_d)))
        cur_d.clear(); cur_w.clear()

    for d, w in zip(docs, weights):
        cur_d.append(d); cur_w.append(w)
        if len(cur_d) >= 256 or (len(cur_d) >= 64 and d - cur_d[0] > len(cur_d) * 8):
            flush()
        total += 1
    if cur_d:
        flush()
    if total == 0:
        del buf[start + 4:]            # the reference leaves the 16-byte placeholder in place and rewrites only the count; see note below
        buf += struct.pack("<iq", 0, 0)
        return
    skip = len(buf)
    for mn, mx, off, mw, cnt in blocks:
        buf += struct.pack("<iiqBi", mn, mx, off, mw, cnt)
    buf[start:start + 16] = struct.pack("<iiq", total, len(blocks), skip)


def _compact_trie(words_outputs, reverse=False):
    root = {"c": {}, "f": False, "o": -1}
    for w, o in words_outputs:
        cur = root
        for ch in (reversed(w) if reverse else w):
            cur = cur["c"].setdefault(ch, {"c": {}, "f": False, "o": -1})
        cur["f"] = True; cur["o"] = o
    nodes, arcs = [], []
    index = {id(root): 0}; q = deque([root]); nxt = 1
    while q:
        b = q.popleft()
        node = (len(arcs), len(b["c"]), b["f"], b["o"])
        for label in sorted(b["c"], key=ord):
            ch = b["c"][label]
            if id(ch) not in index:
                index[id(ch)] = nxt; nxt += 1; q.append(ch)
            arcs.append((ord(label), index[id(ch)], ch["o"] if ch["f"] else -1, ch["f"]))
        nodes.append(node)
    return nodes, arcs


def fst(terms):
    """terms: sorted list of texts; output = ordinal.  FstSerializer.Write."""
    out = bytearray(struct.pack("<IHi", 0x46535432, 1, len(terms)))
    for rev in (False, True):
        nodes, arcs = _compact_trie([(t, i) for i, t in enumerate(terms)], rev)
        out += struct.pack("<i", len(nodes))
        for a, n, f, o in nodes:
            out += struct.pack("<iH?i", a, n, f, o)
        out += struct.pack("<i", len(arcs))
        for lb, tg, o, f in arcs:
            out += struct.pack("<Hii?", lb, tg, o, f)
        out += struct.pack("<i", 0)
    return bytes(out)


def _compact_set(data, width, index, value):
    pos = index * width; block = pos >> 6; shift = pos & 63
    data[block] |= (value << shift) & 0xFFFFFFFFFFFFFFFF
    if shift + width > 64:
        data[block + 1] |= value >> (64 - shift)


def elias_fano(values):
    n = len(values); u = values[-1]
    l = 0
    if u > n:
        l = (u // n).bit_length()                      # Log2(u / n) + 1
    max_h = u >> l
    hb_len = max_h + n
    words = [0] * ((hb_len + 63) // 64)
    low = [0] * ((n * l + 63) // 64)
    for i, v in enumerate(values):
        if l > 0:
            _compact_set(low, l, i, v & ((1 << l) - 1))
        p = (v >> l) + i
        words[p >> 6] |= 1 << (p & 63)
    # DArray.Build(select1)
    block_inv, sub_inv, overflow = [], [], []
    cur = []

    def flush():
        fst_, lst = cur[0], cur[-1]
        if lst - fst_ < (1 << 16):
            block_inv.append(fst_ & 0x7FFFFFFFFFFFFFFF)
            for i in range(0, len(cur), 32):
                sub_inv.append(cur[i] - fst_)
        else:
            block_inv.append(len(overflow) | 0x8000000000000000)
            overflow.extend(cur)
            for i in range(0, len(cur), 32):
                sub_inv.append(0)
        cur.clear()
    for i, w in enumerate(words):
        if i == len(words) - 1 and hb_len % 64:
            w &= (1 << (hb_len % 64)) - 1
        while w:
            tz = (w & -w).bit_length() - 1
            cur.append(i * 64 + tz)
            if len(cur) == 1024:
                flush()
            w &= w - 1
    if cur:
        flush()
    out = bytearray(struct.pack("<iiii", n, l, hb_len, len(words)))
    for w in words:
        out += struct.pack("<Q", w)
    out += struct.pack("<i", len(block_inv)) + b"".join(struct.pack("<Q", x) for x in block_inv)
    out += struct.pack("<i", len(sub_inv)) + b"".join(struct.pack("<H", x) for x in sub_inv)
    out += struct.pack("<i", len(overflow)) + b"".join(struct.pack("<q", x) for x in overflow)
    out += struct.pack("<iii", l, n, len(low)) + b"".join(struct.pack("<Q", x) for x in low)      # CompactArray(lowBitsData, l, n).Write
    return bytes(out)


def write(path, terms, doc_count, doc_id_offset=0):
    """terms: [(text, [doc ids], [weight bytes])] — the non-stop terms with df > 0, any order (the writer sorts them ordinally, SegmentWriter.cs:15-18)."""
    terms = sorted(terms, key=lambda t: [ord(c) for c in t[0]])
    buf = bytearray(struct.pack("<Iiii", 0x494E4653, 1, len(terms), doc_count))
    postings_start = len(buf)
    offsets = []
    for text, docs, weights in terms:
        offsets.append(len(buf))
        block_postings(buf, [d - doc_id_offset for d in docs] if doc_id_offset > 0 else docs, weights)
    fst_start = len(buf)
    buf += fst([t[0] for t in terms])
    offsets_start = len(buf)
    if offsets:
        buf += elias_fano(offsets)
    buf += struct.pack("<qqq", postings_start, fst_start, offsets_start)
    with open(path, "wb") as f:
        f.write(bytes(buf))
    return bytes(buf)
