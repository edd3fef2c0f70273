// The derived sections of an INFDX2 file, checked against the index rebuilt from the file's documents ("a corrupt derived section
// passes").  SearchEngine.Load READS these sections and searches with them (IndexPersistence.cs:183-203, SearchEngine.cs:430-439); the product rebuilds its own
// from the stored documents, so a file is only as good as the reference's Load of it would be if every stored structure says what the rebuilt one says:
//
//   term FST            FstSerializer.Write (Fst/FstSerializer.cs:16-37): every term of the TermCollection — stop terms included — with its collection index
//                       (VectorModel.cs:227-237); forward and reverse trie must enumerate exactly the rebuilt dictionary with output == term id.
//   short-query index   PositionalPrefixIndex.Write (ShortQuery/PositionalPrefixIndex.cs:249-274): per 1..3-character token prefix the (document, token
//                       position, wordStart) postings sorted by (document, position) (PrefixPosting.cs:29-35); regenerated from the index texts, entry by entry.
//   metadata cache      DocumentMetadataCache.Write (Coverage/DocumentMetadataCache.cs:119-133): first token and token count (<= 65535) of
//                       normalize(lower(IndexedText)) per document, empty for deleted / empty documents (VectorModel.cs:251-312).
//   WordMatcher         WordMatcher.Save (WordMatcher/WordMatcher.cs:391-455) behind the data checksum: the exact and the symmetric-delete dictionary (key ->
//                       RoaringBitmap of documents, the portable Roaring format of Internalized/Roaring/RoaringArray.cs:396-469), the affix FST (word -> id of its
//                       LAST occurrence, quirk Q13: WordMatcher.cs:167-193) and the occurrence -> document map.
//
// Every offset and count is bounds-checked before use; nothing here allocates more than the file's size allows.
#pragma once
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>
#include "index.h"
#include "infdx2.h"
#include "infs.h"

namespace infx {
namespace infdx2 {

inline bool rd_string(infs::Rd& r, std::u16string& out) {              // BinaryReader.ReadString
    uint32_t len = 0; int shift = 0;
    for (;;) { const uint8_t b = r.get<uint8_t>(); if (!r.ok || shift > 28) { r.ok = false; return false; } len |= (uint32_t)(b & 0x7F) << shift; if (!(b & 0x80)) break; shift += 7; }
    if (r.p > r.n || r.n - r.p < len) { r.ok = false; return false; }
    Rd q{r.b + r.p, r.b + r.p + len};
    // reuse the UTF-8 decoder of the document / term strings: a length-prefixed string of `len` bytes
    out.clear(); out.reserve(len);
    const uint8_t* p = q.p; const uint8_t* e = q.e;
    while (p < e) {
        uint32_t c = *p++; const int extra = c >= 0xF0 ? 3 : (c >= 0xE0 ? 2 : (c >= 0xC0 ? 1 : 0));
        if (extra) { c &= (0x3Fu >> extra); for (int k = 0; k < extra && p < e; k++) c = (c << 6) | (*p++ & 0x3Fu); }
        if (c >= 0x10000) { c -= 0x10000; out.push_back((char16_t)(0xD800 + (c >> 10))); out.push_back((char16_t)(0xDC00 + (c & 0x3FF))); } else out.push_back((char16_t)c);
    }
    r.p += len; return true;
}

// a serialized RoaringBitmap of `len` bytes at r.p -> ascending document ids appended to `out`; false when malformed or when an id is outside [0, N)
inline bool rd_roaring(infs::Rd& r, size_t len, int32_t N, std::vector<int32_t>& out) {
    if (r.p > r.n || r.n - r.p < len) { r.ok = false; return false; }
    struct Q : infs::Rd { size_t p0; Q(const uint8_t* b, size_t n, size_t p) : infs::Rd(b, n, p), p0(p) {} } q(r.b, r.p + len, r.p);
    r.p += len;
    const uint32_t cookie = q.get<uint32_t>();
    const bool hasRun = (cookie & 0xFFFF) == 12347;
    if (!q.ok || (!hasRun && cookie != 12346)) return false;
    const uint32_t size = hasRun ? (cookie >> 16) + 1 : q.get<uint32_t>();
    if (!q.ok || size > 65536 || (size_t)size * 4 > q.n - q.p) return false;
    const size_t runMapAt = q.p;
    if (hasRun && !q.skip((size + 7) / 8)) return false;
    const size_t keysAt = q.p;
    if (!q.skip((size_t)size * 4)) return false;
    const bool hasOffsets = !hasRun || size >= 4;
    const size_t offsAt = q.p, blobAt = q.p0;
    if (hasOffsets && !q.skip((size_t)size * 4)) return false;                   // the reference's reader skips them; checked here against the layout they describe
    int64_t prev = -1;
    for (uint32_t k = 0; k < size; k++) {
        if (hasOffsets) { uint32_t o; std::memcpy(&o, q.b + offsAt + (size_t)k * 4, 4); if ((size_t)o != q.p - blobAt) return false; }
        uint16_t key, cm1; std::memcpy(&key, q.b + keysAt + (size_t)k * 4, 2); std::memcpy(&cm1, q.b + keysAt + (size_t)k * 4 + 2, 2);
        const uint32_t card = 1u + cm1; const int64_t hi = (int64_t)key << 16;
        if (k && hi <= prev - (prev & 0xFFFF)) return false;              // container keys ascend
        const bool isRun = hasRun && ((q.b[runMapAt + k / 8] >> (k % 8)) & 1);
        const size_t before = out.size();
        if (isRun) {
            const uint16_t nruns = q.get<uint16_t>();
            if (!q.ok || (size_t)nruns * 4 > q.n - q.p) return false;
            int64_t last = -1;
            for (uint16_t j = 0; j < nruns; j++) {
                const uint16_t v = q.get<uint16_t>(), l = q.get<uint16_t>();
                if ((int64_t)v <= last || (uint32_t)v + l > 65535u) return false;
                if (hi + v + l >= (int64_t)N) return false;
                for (uint32_t x = v; x <= (uint32_t)v + l; x++) out.push_back((int32_t)(hi + x));
                last = (int64_t)v + l;
            }
        } else if (card > 4096) {
            if (q.n - q.p < 8192) return false;
            for (uint32_t wd = 0; wd < 1024; wd++) {
                uint64_t bits; std::memcpy(&bits, q.b + q.p + (size_t)wd * 8, 8);
                while (bits) { const int bit = __builtin_ctzll(bits); bits &= bits - 1; const int64_t id = hi + wd * 64 + bit; if (id >= (int64_t)N) return false; out.push_back((int32_t)id); }
            }
            q.p += 8192;
        } else {
            if (q.n - q.p < (size_t)card * 2) return false;
            int32_t last = -1;
            for (uint32_t i = 0; i < card; i++) { uint16_t v; std::memcpy(&v, q.b + q.p + (size_t)i * 2, 2); if ((int32_t)v <= last || hi + v >= (int64_t)N) return false; last = v; out.push_back((int32_t)(hi + v)); }
            q.p += (size_t)card * 2;
        }
        if (out.size() - before != card) return false;                    // the cardinality of the key table is the container's
        prev = out.back();
    }
    return q.ok && q.p == q.n;                                            // the blob is exactly one bitmap
}

inline bool rd_fst(infs::Rd& r, infs::Trie& fw, infs::Trie& rv, int32_t& termCount, std::string& err) {
    const uint32_t magic = r.get<uint32_t>(); const uint16_t ver = r.get<uint16_t>(); termCount = r.get<int32_t>();
    if (!r.ok || magic != 0x46535432u || ver != 1 || termCount < 0) { err = "FST header"; return false; }
    return infs::read_trie(r, fw, err) && infs::read_trie(r, rv, err);
}

// stored key -> id of the rebuilt key it stands for: the same text; else, for a stored text that is the lossy image (infdx2.h: lossy) of rebuilt keys with half a
// surrogate pair in them, the next of those in id order = order of first appearance = the order the reference wrote them in.  Every rebuilt key is handed out once.
struct LossyMatcher {
    const KeyTable& K; std::vector<uint8_t> seen;
    std::unordered_map<std::u16string, std::pair<std::vector<uint32_t>, size_t>> side; bool built = false;
    explicit LossyMatcher(const KeyTable& k) : K(k), seen(k.size(), 0) {}
    template <class Ok> int64_t match(const std::u16string& stored, Ok eligible) {
        bool replaced = false; for (char16_t c : stored) if (c == 0xFFFD) { replaced = true; break; }
        if (!replaced) {                                     // no U+FFFD in it: nothing collapses onto this text
            const int64_t id = K.find(uview(stored.data(), stored.size()));
            if (id >= 0 && !seen[(size_t)id] && eligible((uint32_t)id)) { seen[(size_t)id] = 1; return id; }
            return -1;
        }
        if (!built) {                                        // candidates of a text: the keys whose lossy image it is and the key that really reads like that, in id order
            for (uint32_t k = 0; k < (uint32_t)K.size(); k++) { const uview key = K.key(k); if (has_lone_surrogate(key.data(), key.size())) side[lossy(key.data(), key.size())].first.push_back(k); }
            for (auto& kv : side) { const int64_t id = K.find(uview(kv.first.data(), kv.first.size())); if (id >= 0) { kv.second.first.push_back((uint32_t)id); std::sort(kv.second.first.begin(), kv.second.first.end()); } }
            built = true;
        }
        auto it = side.find(stored);
        if (it == side.end()) {
            const int64_t id = K.find(uview(stored.data(), stored.size()));
            if (id >= 0 && !seen[(size_t)id] && eligible((uint32_t)id)) { seen[(size_t)id] = 1; return id; }
            return -1;
        }
        auto& q = it->second;
        while (q.second < q.first.size()) { const uint32_t c = q.first[q.second++]; if (!seen[c] && eligible(c)) { seen[c] = 1; return (int64_t)c; } }
        return -1;
    }
    int64_t match(const std::u16string& stored) { return match(stored, [](uint32_t) { return true; }); }
};

// returns nullptr when every section the file holds agrees with the rebuilt index, else what differs (static text)
inline const char* check_derived(const File& F, const HostIndex& ix) {
    const HostConfig& cfg = ix.cfg;
    const int32_t N = ix.N;
    infs::Rd r(F.blob.data(), F.dataEnd, F.derivedAt);
    std::string err;
    // ---- term FST ---------------------------------------------------------------------------------------------------
    if (F.flags & 1u) {
        infs::Trie fw, rv; int32_t tc = 0;
        if (!rd_fst(r, fw, rv, tc, err)) return "the stored term FST is malformed";
        const size_t K = ix.terms.K();
        std::vector<std::u16string> terms; std::vector<int32_t> outs;
        if (!infs::enumerate_trie(fw, terms, outs, err)) return "the stored term FST is malformed";
        if (terms.size() != K || (size_t)tc != K) return "the stored term FST holds another number of terms than the index rebuilt from the documents";
        for (size_t i = 0; i < terms.size(); i++)
            if (ix.terms.keys.find(uview((const u16*)terms[i].data(), terms[i].size())) != (int64_t)outs[i]) return "a term of the stored FST maps to another term index than in the rebuilt index";
        terms.clear(); outs.clear();
        if (!infs::enumerate_trie(rv, terms, outs, err)) return "the stored term FST is malformed";
        if (terms.size() != K) return "the stored reverse term FST holds another number of terms than the rebuilt index";
        for (size_t i = 0; i < terms.size(); i++) {
            std::reverse(terms[i].begin(), terms[i].end());
            if (ix.terms.keys.find(uview((const u16*)terms[i].data(), terms[i].size())) != (int64_t)outs[i]) return "a term of the stored reverse FST maps to another term index than in the rebuilt index";
        }
    }
    // ---- short-query index --------------------------------------------------------------------------------------------
    if (F.flags & 2u) {
        struct List { size_t at; uint32_t n, cur; };
        std::vector<List> all;                                             // in file order
        std::unordered_map<uint64_t, std::vector<uint32_t>> byKey;          // key: length << 48 | up to three UTF-16 units -> lists stored under that text (file order)
        auto keyOf = [](const u16* p, int L) { uint64_t k = (uint64_t)L << 48; for (int i = 0; i < L; i++) k |= (uint64_t)p[i] << (16 * i); return k; };
        uint64_t stored = 0;
        auto take = [&](uint64_t key, bool mayRepeat) -> bool {
            const int32_t n = r.get<int32_t>();
            if (!r.ok || n < 0 || (size_t)n > (r.n - r.p) / 7) return false;
            auto& v = byKey[key];
            if (!v.empty() && !mayRepeat) return false;                    // a prefix appears once — unless its text is the lossy image of several (infdx2.h: lossy)
            v.push_back((uint32_t)all.size()); all.push_back(List{r.p, (uint32_t)n, 0u});
            stored += (uint64_t)n; r.p += (size_t)n * 7; return true;
        };
        const int32_t n1 = r.get<int32_t>();
        if (!r.ok || n1 < 0 || n1 > 65536) return "the stored short-query index is malformed";
        for (int32_t i = 0; i < n1; i++) { const u16 c = r.get<uint16_t>(); if (!r.ok || !take(keyOf(&c, 1), false)) return "the stored short-query index is malformed"; }
        const int32_t nm = r.get<int32_t>();
        if (!r.ok || nm < 0 || (size_t)nm > (r.n - r.p) / 5) return "the stored short-query index is malformed";
        std::u16string pf;
        for (int32_t i = 0; i < nm; i++) {
            if (!rd_string(r, pf) || pf.size() < 2 || pf.size() > 3) return "the stored short-query index is malformed";
            bool rep = false; for (char16_t c : pf) if (c == 0xFFFD) rep = true;
            if (!take(keyOf((const u16*)pf.data(), (int)pf.size()), rep)) return "the stored short-query index is malformed";
        }
        // regenerate: tokens of the index text in order, prefixes of 1..3 characters, position = (ushort) token index (PositionalPrefixIndex.cs:56-118)
        std::vector<std::pair<uint64_t, uint16_t>> want;
        uint64_t made = 0;
        std::unordered_map<uint64_t, uint32_t> listOf;                      // regenerated prefix -> its stored list, fixed at the prefix's first appearance
        std::unordered_map<uint64_t, uint32_t> nextUnder;                   // stored text -> how many of its lists are taken
        auto list_for = [&](uint64_t key) -> int64_t {
            auto f = listOf.find(key);
            if (f != listOf.end()) return f->second;
            const int L = (int)(key >> 48); u16 u[3]; for (int i = 0; i < L; i++) u[i] = (u16)(key >> (16 * i));
            uint64_t sk = key;
            if (L >= 2 && has_lone_surrogate(u, (size_t)L)) { const std::u16string ls = lossy(u, (size_t)L); sk = keyOf(ls.data(), L); }      // single characters are written as numbers
            auto v = byKey.find(sk);
            if (v == byKey.end()) return -1;
            uint32_t& nx = nextUnder[sk];
            if (nx >= v->second.size()) return -1;
            const uint32_t li = v->second[nx++];
            listOf.emplace(key, li); return li;
        };
        for (int32_t d = 0; d < N; d++) {
            const uview t(ix.text.data() + ix.textOff[d], (size_t)(ix.textOff[d + 1] - ix.textOff[d]));
            want.clear(); uint32_t tok = 0;
            for_each_word(t, [&](int off, int len) { const int mx = std::min(len, 3); for (int L = 1; L <= mx; L++) want.push_back({keyOf(t.data() + off, L), (uint16_t)tok}); tok++; });
            // lists are fixed in order of first appearance (token order), the entries of a document are then matched per list in position order
            for (auto& w : want) if (list_for(w.first) < 0) return "the stored short-query index lacks a (prefix, document, position) entry of the index texts";
            std::sort(want.begin(), want.end());
            for (auto& w : want) {
                List& Lst = all[(size_t)list_for(w.first)];
                if (Lst.cur >= Lst.n) return "the stored short-query index lacks a (prefix, document, position) entry of the index texts";
                const uint8_t* e = F.blob.data() + Lst.at + (size_t)Lst.cur++ * 7;
                int32_t doc; uint16_t pos; std::memcpy(&doc, e, 4); std::memcpy(&pos, e + 4, 2);
                if (doc != d || pos != w.second || e[6] != 1) return "a posting of the stored short-query index differs from the index texts";
            }
            made += want.size();
        }
        if (made != stored) return "the stored short-query index holds postings the index texts do not produce";
    }
    // ---- document metadata cache --------------------------------------------------------------------------------------
    if (F.flags & 16u) {
        const int32_t n = r.get<int32_t>();
        if (!r.ok || n != N) return "the stored document metadata cache holds another number of documents";
        std::u16string first; ustr text, tmp;
        for (int32_t d = 0; d < N; d++) {
            if (!rd_string(r, first)) return "the stored document metadata cache is malformed";
            const uint16_t count = r.get<uint16_t>();
            if (!r.ok) return "the stored document metadata cache is malformed";
            const Doc& D = F.docs[(size_t)d];
            if (D.deleted && first.empty() && count == 0) continue;       // deleted before the cache was built: Empty (VectorModel.cs:263-267)
            text.assign((const u16*)D.text.data(), D.text.size()); lower_inplace(text); normalize_into(text, tmp); text.swap(tmp);
            if (cfg.syn.has()) cfg.syn.canonicalize(text);
            uint32_t tokens = 0; int fo = 0, fl = 0;
            for_each_word(text, [&](int off, int len) { if (tokens == 0) { fo = off; fl = len; } tokens++; });
            const uint16_t wantCount = (uint16_t)std::min<uint32_t>(tokens, 65535u);
            const bool sameFirst = first.size() == (size_t)fl && (fl == 0 || std::memcmp(first.data(), text.data() + fo, (size_t)fl * 2) == 0 ||
                                                                    (has_lone_surrogate(text.data() + fo, (size_t)fl) && lossy(text.data() + fo, (size_t)fl) == first));
            if (count != wantCount || !sameFirst)
                return "an entry of the stored document metadata cache differs from the stored document text";
        }
    }
    if (r.p != F.dataEnd) return "the data section holds bytes behind its last section";
    // ---- WordMatcher (behind the checksum) ------------------------------------------------------------------------------
    infs::Rd w(F.blob.data(), F.blob.size(), F.trailerAt);
    const uint8_t hasWm = w.get<uint8_t>();
    if (!w.ok) return "the file ends before the WordMatcher flag";
    if (hasWm > 1) return "the WordMatcher flag is neither 0 nor 1";
    if (hasWm && !cfg.wordMatcher) return "the file holds WordMatcher data but the engine is configured without a WordMatcher";      // SearchEngine.cs:436-437
    if (!hasWm && cfg.wordMatcher) return "the file lacks the WordMatcher data the engine is configured with";                       // SearchEngine.cs:438-439
    if (hasWm) {
        std::u16string key; std::vector<int32_t> docs;
        for (int which = 0; which < 2; which++) {
            const Csr& C = which ? ix.wmLd1 : ix.wmExact;
            const int32_t n = w.get<int32_t>();
            if (!w.ok || n < 0 || (size_t)n != C.K()) return which ? "the stored symmetric-delete dictionary holds another number of keys than the rebuilt one" : "the stored exact-word dictionary holds another number of keys than the rebuilt one";
            LossyMatcher M(C.keys);
            for (int32_t i = 0; i < n; i++) {
                if (!rd_string(w, key)) return "the stored WordMatcher dictionaries are malformed";
                const int32_t len = w.get<int32_t>();
                docs.clear();
                if (!w.ok || len < 0 || !rd_roaring(w, (size_t)len, N, docs)) return "a document set of the stored WordMatcher dictionaries is malformed";
                const int64_t id = M.match(key);
                if (id < 0) return "a key of the stored WordMatcher dictionaries does not exist in the rebuilt one";
                const uint64_t b = C.off[(size_t)id], m = C.off[(size_t)id + 1] - b;
                if (m != docs.size() || !std::equal(docs.begin(), docs.end(), C.doc.begin() + (ptrdiff_t)b)) return "a document set of the stored WordMatcher dictionaries differs from the rebuilt one";
            }
        }
        const uint8_t hasFst = w.get<uint8_t>();
        if (!w.ok || hasFst != 1) return "the stored WordMatcher holds no affix FST (the engine is configured with SupportAffix)";
        infs::Trie fw, rv; int32_t occ = 0;
        if (!rd_fst(w, fw, rv, occ, err)) return "the stored affix FST is malformed";
        const int32_t nmap = w.get<int32_t>();
        if (!w.ok || nmap < 0 || nmap != occ || (size_t)nmap > (w.n - w.p) / 8) return "the stored affix occurrence map is malformed";
        std::vector<int32_t> docOf((size_t)nmap, -1);
        for (int32_t i = 0; i < nmap; i++) {
            const int32_t id = w.get<int32_t>(), len = w.get<int32_t>();
            docs.clear();
            if (!w.ok || id < 0 || id >= nmap || docOf[(size_t)id] >= 0 || len < 0 || !rd_roaring(w, (size_t)len, N, docs) || docs.size() != 1) return "the stored affix occurrence map is malformed";
            docOf[(size_t)id] = docs[0];
        }
        if (w.p != w.n) return "the file holds bytes behind the WordMatcher section";
        for (int pass = 0; pass < 2; pass++) {
            std::vector<std::u16string> words; std::vector<int32_t> outs;
            if (!infs::enumerate_trie(pass ? rv : fw, words, outs, err)) return "the stored affix FST is malformed";
            if (words.size() != ix.affixFwd.size()) return "the stored affix FST holds another number of words than the rebuilt one";
            for (size_t i = 0; i < words.size(); i++) {
                if (pass) std::reverse(words[i].begin(), words[i].end());
                const int64_t id = ix.words.find(uview((const u16*)words[i].data(), words[i].size()));
                if (id < 0 || (int)words[i].size() < cfg.wmMinLD1) return "a word of the stored affix FST does not exist in the rebuilt one";
                if (outs[i] < 0 || outs[i] >= nmap || docOf[(size_t)outs[i]] != ix.wordLastDoc[(size_t)id]) return "a word of the stored affix FST leads to another document than in the rebuilt one (last occurrence, WordMatcher.cs:167-193)";
            }
        }
    } else if (w.p != w.n) return "the file holds bytes behind the WordMatcher flag";
    return nullptr;
}

}  // namespace infdx2
}  // namespace infx
