"""Writer of the reference's INFDX2 index file format (SearchEngine.Save -> IndexPersistence.Save, src/Infidex/Indexing/IndexPersistence.cs:33-99,
WriteDocuments :304-320, WriteTerms :355-383, checksums :268-299), restated for the tests of the product's reader (infx_engine_load_index).
TEST INFRASTRUCTURE.  No file written by the reference itself exists in its repository (and no .NET runtime here), so the reader's parity is pinned to
this restatement of the writer, not to a reference-produced file: PARITY UNPINNED for f4."""
import struct


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


def _string(s: str) -> bytes:        # BinaryWriter.Write(string): 7-bit encoded UTF-8 byte length + bytes
    b = s.encode("utf-8"); n = len(b); out = bytearray()
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
