// INFDX2 index files (the reference's SearchEngine.Save / Load: Indexing/IndexPersistence.cs:33-206, 269-399) -> the product's host index.
//
// File: "INFDX2" | u32 version (2) | u32 flags | u32 docCount | u32 termCount | u32 headerChecksum | u32 dataLength | data | u32 dataChecksum | [WordMatcher ...]
// data: documents (i32 count; per document i32 id, i64 DocumentKey, string IndexedText, string clientInformation, i32 segment, i32 jsonIndex, u8 deleted)
//       terms     (i32 count of the non-stop terms; per term string text, i32 documentFrequency, i32 postingCount, postingCount x {i32 docId, u8 weight})
//       [FST] [short-query index] [document metadata cache]   — derived structures; the product rebuilds them from the documents and host/infdx2_verify.h
//       checks the stored ones against what it built (as it does with the WordMatcher section behind the checksum)
// strings are BinaryWriter strings: 7-bit-encoded UTF-8 byte length, then the bytes.  Checksums: IndexPersistence.cs:268-299.
//
// What Load gives the reference is documents { DocumentKey, one "content" field = IndexedText, Weight.Med } (ReadDocuments :322-349) plus the stored
// postings.  The product indexes those documents with its own builder — which is what produced the stored postings when the corpus was indexed as single
// Med-weight fields (Document(key, text)) — and then CHECKS the file's term section against what it built, term by term, posting by posting: a file that
// was written from differently weighted fields (its tf bytes then differ) is refused rather than searched with other weights than the reference would use.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>

namespace infx {
namespace infdx2 {

struct Doc { int32_t id; int64_t key; std::u16string text; bool deleted; };
struct TermRec { std::u16string text; int32_t df; std::vector<int32_t> docs; std::vector<uint8_t> w; };
struct File { uint32_t flags = 0, docCount = 0, termCount = 0; std::vector<Doc> docs; std::vector<TermRec> terms; std::string error;
              std::vector<uint8_t> blob;                       // the whole file: the derived sections are verified in place (infdx2_verify.h)
              size_t derivedAt = 0, dataEnd = 0, trailerAt = 0; };      // [derivedAt, dataEnd) = FST / short-query index / metadata cache; trailerAt = the WordMatcher flag

inline uint32_t rotl7(uint32_t c) { return (c << 7) | (c >> 25); }
inline uint32_t checksum_words(const uint32_t* v, size_t n) { uint32_t c = 0x12345678u; for (size_t i = 0; i < n; i++) { c ^= v[i]; c = rotl7(c); } return c; }
inline uint32_t checksum_bytes(const uint8_t* d, size_t n) {
    uint32_t c = 0x12345678u;
    for (size_t i = 0; i < n; i += 4) { uint32_t v = 0; const size_t r = std::min<size_t>(4, n - i); for (size_t j = 0; j < r; j++) v |= (uint32_t)d[i + j] << (j * 8); c ^= v; c = rotl7(c); }
    return c;
}

// BinaryWriter.Write(string) encodes with Encoding.UTF8, whose replacement fallback writes U+FFFD for a UTF-16 unit that is half of a surrogate pair on its
// own.  Such strings exist in every index of a text with characters outside the BMP: a 3-gram window, a 1..3-unit token prefix or a single-unit deletion cuts
// pairs apart.  The reference's own round trip of such an index (PersistenceTests.cs:152-196, one document: U+1F50D) loads fine — its Load compares nothing —
// so the stored keys are compared with the rebuilt ones THROUGH that mapping: lossy(rebuilt) == stored, several rebuilt keys that collapse onto one stored text
// are matched in their order of first appearance (the order Dictionary<,> enumerates them in, i.e. the order they were written).
inline bool has_lone_surrogate(const char16_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const char16_t c = p[i];
        if (c >= 0xD800 && c <= 0xDBFF) { if (i + 1 < n && p[i + 1] >= 0xDC00 && p[i + 1] <= 0xDFFF) { i++; continue; } return true; }
        if (c >= 0xDC00 && c <= 0xDFFF) return true;
    }
    return false;
}
inline std::u16string lossy(const char16_t* p, size_t n) {
    std::u16string o; o.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const char16_t c = p[i];
        if (c >= 0xD800 && c <= 0xDBFF && i + 1 < n && p[i + 1] >= 0xDC00 && p[i + 1] <= 0xDFFF) { o.push_back(c); o.push_back(p[++i]); }
        else o.push_back((c >= 0xD800 && c <= 0xDFFF) ? (char16_t)0xFFFD : c);
    }
    return o;
}

struct Rd {
    const uint8_t* p; const uint8_t* e; bool ok = true;
    template <class T> T get() { T v{}; if ((size_t)(e - p) < sizeof(T)) { ok = false; return v; } std::memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    bool str(std::u16string& out) {                       // BinaryReader.ReadString: 7-bit encoded byte length + UTF-8
        uint32_t len = 0; int shift = 0;
        for (;;) { if (p >= e || shift > 28) { ok = false; return false; } const uint8_t b = *p++; len |= (uint32_t)(b & 0x7F) << shift; if (!(b & 0x80)) break; shift += 7; }
        if ((size_t)(e - p) < len) { ok = false; return false; }
        out.clear(); out.reserve(len);
        const uint8_t* q = p; const uint8_t* qe = p + len;
        while (q < qe) {
            uint32_t c = *q++; int extra = c >= 0xF0 ? 3 : (c >= 0xE0 ? 2 : (c >= 0xC0 ? 1 : 0));
            if (extra) { c &= (0x3Fu >> extra); for (int k = 0; k < extra && q < qe; k++) c = (c << 6) | (*q++ & 0x3Fu); }
            if (c >= 0x10000) { c -= 0x10000; out.push_back((char16_t)(0xD800 + (c >> 10))); out.push_back((char16_t)(0xDC00 + (c & 0x3FF))); } else out.push_back((char16_t)c);
        }
        p += len; return true;
    }
};

inline bool read_file(const std::string& path, File& F) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { F.error = "cannot open " + path; return false; }
    F.blob.assign((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const std::vector<uint8_t>& buf = F.blob;
    Rd R{buf.data(), buf.data() + buf.size()};
    if (buf.size() < 30 || std::memcmp(buf.data(), "INFDX2", 6) != 0) { F.error = "invalid index magic: expected INFDX2"; return false; }
    R.p += 6;
    const uint32_t version = R.get<uint32_t>(); F.flags = R.get<uint32_t>(); F.docCount = R.get<uint32_t>(); F.termCount = R.get<uint32_t>();
    const uint32_t hsum = R.get<uint32_t>();
    if (version != 2) { F.error = "unsupported index version " + std::to_string(version) + " (expected 2)"; return false; }
    const uint32_t hv[4] = {version, F.flags, F.docCount, F.termCount};
    if (hsum != checksum_words(hv, 4)) { F.error = "header checksum mismatch - index file may be corrupted"; return false; }
    const uint32_t dlen = R.get<uint32_t>();
    if (!R.ok || (size_t)(R.e - R.p) < (size_t)dlen + 4) { F.error = "index data section truncated"; return false; }
    const uint8_t* data = R.p; uint32_t dsum; std::memcpy(&dsum, data + dlen, 4);
    if (dsum != checksum_bytes(data, dlen)) { F.error = "data checksum mismatch - index file may be corrupted"; return false; }
    Rd D{data, data + dlen};
    const int32_t nd = D.get<int32_t>();
    if (!D.ok || nd < 0 || (uint32_t)nd != F.docCount) { F.error = "document count mismatch between header and data"; return false; }
    F.docs.resize((size_t)nd);
    std::u16string info;
    for (auto& d : F.docs) {
        d.id = D.get<int32_t>(); d.key = D.get<int64_t>(); D.str(d.text); D.str(info); (void)D.get<int32_t>(); (void)D.get<int32_t>(); d.deleted = D.get<uint8_t>() != 0;
        if (!D.ok) { F.error = "document section truncated"; return false; }
    }
    const int32_t nt = D.get<int32_t>();
    if (!D.ok || nt < 0) { F.error = "term section truncated"; return false; }
    F.terms.resize((size_t)nt);
    for (auto& t : F.terms) {
        D.str(t.text); t.df = D.get<int32_t>(); const int32_t pc = D.get<int32_t>();
        if (!D.ok || pc < 0 || (size_t)(D.e - D.p) < (size_t)pc * 5) { F.error = "term section truncated"; return false; }
        t.docs.resize((size_t)pc); t.w.resize((size_t)pc);
        for (int32_t i = 0; i < pc; i++) { std::memcpy(&t.docs[i], D.p, 4); t.w[i] = D.p[4]; D.p += 5; }
    }
    F.derivedAt = (size_t)(D.p - buf.data()); F.dataEnd = (size_t)(data + dlen - buf.data()); F.trailerAt = F.dataEnd + 4;
    return true;        // the FST / short-query index / metadata cache bytes (rest of the data) and the WordMatcher section after the checksum: infdx2_verify.h
}

}  // namespace infdx2
}  // namespace infx
