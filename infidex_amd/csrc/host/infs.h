// INFS segment files (the reference's SearchEngine.Flush: VectorModel.Flush -> SegmentWriter.WriteSegment, Indexing/Segments/SegmentWriter.cs:13-94) -> flat
// posting arrays in the layout infx_upload_postings takes.  SURVEY 8 f2(a): the import path from flushed segments.
//
// File:  u32 "INFS" 0x494E4653 | i32 version (1) | i32 termCount | i32 docCount
//        postings, one list per term in ORDINAL order of the term text          BlockPostingsWriter.cs:24-161
//            i32 totalCount | i32 numBlocks | i64 skipTableOffset                (an empty list: the 16 bytes are zero)
//            blocks: i32 byteLength | GroupVarInt deltas (first delta = first doc id) | count weight bytes      GroupVarInt.cs:56-115
//            skip table: per block i32 minDoc | i32 maxDoc | i64 blockOffset | u8 maxWeight | i32 count
//        term index "FST2": u32 magic | u16 version | i32 termCount | forward trie | reverse trie; a trie = i32 nodes x {i32 arcStart, u16 arcCount, u8 final,
//            i32 output} | i32 arcs x {u16 label, i32 target, i32 output, u8 final} | i32 root; BFS order, children sorted by label, output = term ordinal
//                                                                                Fst/FstBuilder.cs:80-166, Fst/FstSerializer.cs:16-111
//        offsets: Elias-Fano of the lists' file offsets                          Compression/EliasFano.cs:29-117 (+ DArray select index, CompactArray low bits)
//        footer: i64 postingsStart | i64 fstStart | i64 offsetsStart
// All offsets are absolute file positions.  Doc ids are segment-local (the writer subtracts its docIdOffset).
//
// The reader trusts nothing: every offset and count is bounds-checked, the skip table is checked against the decoded blocks (min / max doc, max weight, counts),
// doc ids must ascend and stay below docCount, the trie must enumerate exactly termCount terms with ordinals 0..T-1 in ordinal text order, the reverse trie must hold
// the same number of terms, the Elias-Fano offsets must ascend inside the postings section and its select index must agree with the bit vector.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include <algorithm>

namespace infx {
namespace infs {

struct Segment {
    int32_t docCount = 0;
    std::vector<std::u16string> terms;           // ordinal order == the file's term ordinals
    std::vector<uint64_t> off;                   // T + 1
    std::vector<int32_t> doc; std::vector<uint8_t> w;
    std::vector<uint32_t> blocksPerTerm;         // how the writer blocked each list (diagnostics)
    std::string error;
};

struct Rd {
    const uint8_t* b; size_t n; size_t p; bool ok = true;
    Rd(const uint8_t* base, size_t len, size_t pos) : b(base), n(len), p(pos) {}
    template <class T> T get() { T v{}; if (p > n || n - p < sizeof(T)) { ok = false; p = n; return v; } std::memcpy(&v, b + p, sizeof(T)); p += sizeof(T); return v; }
    bool skip(size_t k) { if (p > n || n - p < k) { ok = false; p = n; return false; } p += k; return true; }
};

struct Trie { std::vector<int32_t> arcStart; std::vector<uint16_t> arcCount; std::vector<uint8_t> fin; std::vector<int32_t> out;
              std::vector<uint16_t> label; std::vector<int32_t> target; int32_t root = 0; };

inline bool read_trie(Rd& r, Trie& t, std::string& err) {
    const int32_t nn = r.get<int32_t>();
    if (!r.ok || nn < 1 || (size_t)nn > (r.n - r.p) / 11) { err = "term index: bad node count"; return false; }
    t.arcStart.resize(nn); t.arcCount.resize(nn); t.fin.resize(nn); t.out.resize(nn);
    for (int32_t i = 0; i < nn; i++) { t.arcStart[i] = r.get<int32_t>(); t.arcCount[i] = r.get<uint16_t>(); t.fin[i] = r.get<uint8_t>(); t.out[i] = r.get<int32_t>(); }
    const int32_t na = r.get<int32_t>();
    if (!r.ok || na < 0 || (size_t)na > (r.n - r.p) / 11) { err = "term index: bad arc count"; return false; }
    t.label.resize(na); t.target.resize(na);
    std::vector<int32_t> arcOut((size_t)na); std::vector<uint8_t> arcFin((size_t)na);      // copies of the target's output / final flag (FstBuilder.cs:154-160): checked below
    for (int32_t i = 0; i < na; i++) { t.label[i] = r.get<uint16_t>(); t.target[i] = r.get<int32_t>(); arcOut[i] = r.get<int32_t>(); arcFin[i] = r.get<uint8_t>(); }
    t.root = r.get<int32_t>();
    if (!r.ok || t.root < 0 || t.root >= nn) { err = "term index: truncated"; return false; }
    for (int32_t i = 0; i < nn; i++) {
        if (t.arcCount[i] && (t.arcStart[i] < 0 || (int64_t)t.arcStart[i] + t.arcCount[i] > na)) { err = "term index: arc range out of bounds"; return false; }
        for (int k = 1; k < t.arcCount[i]; k++) if (t.label[t.arcStart[i] + k] <= t.label[t.arcStart[i] + k - 1]) { err = "term index: children not sorted by label"; return false; }
    }
    for (int32_t i = 0; i < na; i++) if (t.target[i] <= 0 || t.target[i] >= nn) { err = "term index: arc target out of bounds"; return false; }
    for (int32_t i = 0; i < nn; i++) if (t.fin[i] > 1 || (!t.fin[i] && t.out[i] != -1)) { err = "term index: final flag neither 0 nor 1, or an output on a non-final node"; return false; }
    // the layout CompactTrie produces (FstBuilder.cs:110-166): breadth-first, root first, a node's arcs behind those of the nodes before it, arc k leading to node k + 1
    if (t.root != 0 || na != nn - 1) { err = "term index: not in FstBuilder's breadth-first layout"; return false; }
    for (int32_t i = 0, run = 0; i < nn; i++) { if (t.arcStart[i] != run) { err = "term index: not in FstBuilder's breadth-first layout"; return false; } run += t.arcCount[i]; }
    for (int32_t i = 0; i < na; i++) if (t.target[i] != i + 1) { err = "term index: not in FstBuilder's breadth-first layout"; return false; }
    for (int32_t i = 0; i < na; i++) {          // redundant copies of the node's fields: a file where they disagree was not written by FstBuilder
        const int32_t c = t.target[i];
        if (arcFin[i] != t.fin[c] || arcOut[i] != (t.fin[c] ? t.out[c] : -1)) { err = "term index: an arc disagrees with its target node"; return false; }
    }
    return true;
}
// terms of a trie in ordinal order (pre-order, children ascending) with their outputs; fails on a cycle (more nodes visited than exist)
inline bool enumerate_trie(const Trie& t, std::vector<std::u16string>& terms, std::vector<int32_t>& outs, std::string& err) {
    struct Fr { int32_t node; uint16_t next; };
    std::vector<Fr> st{{t.root, 0}}; std::u16string cur; size_t visited = 0;
    if (t.fin[t.root]) { err = "term index: the empty string is a term"; return false; }
    while (!st.empty()) {
        Fr& f = st.back();
        if (f.next >= t.arcCount[f.node]) { st.pop_back(); if (!cur.empty()) cur.pop_back(); continue; }
        const int32_t a = t.arcStart[f.node] + f.next++;
        const int32_t ch = t.target[a];
        if (++visited > t.arcStart.size()) { err = "term index: not a tree"; return false; }
        cur.push_back((char16_t)t.label[a]);
        if (t.fin[ch]) { terms.push_back(cur); outs.push_back(t.out[ch]); }
        st.push_back({ch, 0});
    }
    if (visited + 1 != t.arcStart.size() || visited != t.label.size()) { err = "term index: nodes or arcs outside the tree"; return false; }
    return true;
}

inline bool decode_list(const uint8_t* b, size_t n, uint64_t at, uint64_t secBegin, uint64_t secEnd, int32_t docCount, Segment& S, uint32_t& nBlocksOut, std::string& err) {
    Rd r(b, n, (size_t)at);
    const int32_t total = r.get<int32_t>();
    if (!r.ok || total < 0) { err = "postings: bad count"; return false; }
    nBlocksOut = 0;
    if (total == 0) return true;
    const int32_t nb = r.get<int32_t>(); const int64_t skipAt = r.get<int64_t>();
    if (!r.ok || nb < 1 || skipAt < (int64_t)secBegin || (uint64_t)skipAt + (uint64_t)nb * 21 > secEnd) { err = "postings: skip table out of bounds"; return false; }
    Rd sk(b, n, (size_t)skipAt);
    int64_t seen = 0; int32_t prevDoc = -1;
    for (int32_t k = 0; k < nb; k++) {
        const int32_t mn = sk.get<int32_t>(), mx = sk.get<int32_t>(); const int64_t bo = sk.get<int64_t>(); const uint8_t mw = sk.get<uint8_t>(); const int32_t cnt = sk.get<int32_t>();
        if (!sk.ok || cnt < 1 || cnt > 256 || bo < (int64_t)secBegin || (uint64_t)bo + 4 > secEnd) { err = "postings: bad block entry"; return false; }
        Rd br(b, n, (size_t)bo);
        const int32_t len = br.get<int32_t>();
        if (!br.ok || len < 0 || (uint64_t)bo + 4 + (uint64_t)len + (uint64_t)cnt > secEnd) { err = "postings: block out of bounds"; return false; }
        const uint8_t* p = b + bo + 4; const uint8_t* pe = p + len;
        int32_t docv = 0; uint8_t maxw = 0; const size_t base = S.doc.size();
        for (int32_t i = 0; i < cnt;) {                                   // GroupVarInt: tag, then up to four little-endian values of 1..4 bytes
            if (p >= pe) { err = "postings: varint data truncated"; return false; }
            const uint8_t tag = *p++;
            for (int g = 0; g < 4 && i < cnt; g++, i++) {
                const int l = ((tag >> (6 - 2 * g)) & 3) + 1;
                if (pe - p < l) { err = "postings: varint data truncated"; return false; }
                uint32_t v = 0; for (int q = 0; q < l; q++) v |= (uint32_t)p[q] << (8 * q);
                p += l;
                if (v > 0x7FFFFFFFu || (int64_t)docv + (int64_t)v > 0x7FFFFFFF) { err = "postings: doc id overflow"; return false; }
                docv += (int32_t)v;                                        // first delta of a block is the absolute doc id (the writer restarts prev at 0)
                if (docv <= prevDoc || docv >= docCount) { err = "postings: doc ids must ascend and stay below the segment's document count"; return false; }
                prevDoc = docv; S.doc.push_back(docv);
            }
        }
        if (p != pe) { err = "postings: varint length mismatch"; return false; }
        for (int32_t i = 0; i < cnt; i++) { S.w.push_back(pe[i]); maxw = std::max(maxw, pe[i]); }
        if (S.doc[base] != mn || S.doc.back() != mx || maxw != mw) { err = "postings: skip table disagrees with the block (min / max doc, max weight)"; return false; }
        seen += cnt;
    }
    if (seen != total) { err = "postings: block counts do not add up"; return false; }
    nBlocksOut = (uint32_t)nb;
    return true;
}

inline bool parse(const uint8_t* b, size_t n, Segment& S) {
    S = Segment();
    auto fail = [&](const char* m) { S.error = m; return false; };
    if (n < 16 + 24) return fail("file too short");
    Rd h(b, n, 0);
    if (h.get<uint32_t>() != 0x494E4653u) return fail("not an INFS segment (magic)");
    if (h.get<int32_t>() != 1) return fail("unsupported segment version");
    const int32_t T = h.get<int32_t>(); S.docCount = h.get<int32_t>();
    if (T < 0 || S.docCount < 0) return fail("negative counts");
    Rd f(b, n, n - 24);
    const int64_t postingsStart = f.get<int64_t>(), fstStart = f.get<int64_t>(), offsetsStart = f.get<int64_t>();
    if (postingsStart != 16 || fstStart < postingsStart || offsetsStart < fstStart || (uint64_t)offsetsStart > n - 24) return fail("footer: sections out of order");
    // term index
    Rd r(b, (size_t)offsetsStart, (size_t)fstStart);
    if (r.get<uint32_t>() != 0x46535432u || r.get<uint16_t>() != 1) return fail("term index: bad magic / version");
    if (r.get<int32_t>() != T) return fail("term index: term count differs from the header");
    Trie fw, rv;
    if (!read_trie(r, fw, S.error) || !read_trie(r, rv, S.error)) return false;
    if (r.p != (size_t)offsetsStart) return fail("term index: trailing bytes");
    std::vector<int32_t> outs, routs; std::vector<std::u16string> rterms;
    if (!enumerate_trie(fw, S.terms, outs, S.error)) return false;
    if ((int64_t)S.terms.size() != T) return fail("term index: the trie does not hold termCount terms");
    for (int32_t i = 0; i < T; i++) if (outs[i] != i) return fail("term index: outputs are not the ordinals of the sorted terms");
    if (!enumerate_trie(rv, rterms, routs, S.error)) return false;
    if ((int64_t)rterms.size() != T) return fail("term index: the reverse trie holds another number of terms");
    {   // the reverse trie maps reversed(term) -> the same ordinal
        for (int32_t i = 0; i < T; i++) {
            if (routs[i] < 0 || routs[i] >= T) return fail("term index: reverse output out of range");
            const std::u16string& t = S.terms[routs[i]];
            if (t.size() != rterms[i].size() || !std::equal(t.rbegin(), t.rend(), rterms[i].begin())) return fail("term index: reverse trie disagrees with the forward trie");
        }
    }
    // offsets (Elias-Fano)
    std::vector<uint64_t> offs((size_t)T);
    if (T > 0) {
        Rd e(b, n - 24, (size_t)offsetsStart);
        const int32_t cnt = e.get<int32_t>(), l = e.get<int32_t>(), hbLen = e.get<int32_t>(), hbWords = e.get<int32_t>();
        if (!e.ok || cnt != T || l < 0 || l > 63 || hbLen < 0 || hbWords != (hbLen + 63) / 64 || (size_t)hbWords > (e.n - e.p) / 8) return fail("offsets: bad Elias-Fano header");
        const size_t hbAt = e.p; e.skip((size_t)hbWords * 8);
        const int32_t nBlk = e.get<int32_t>(); if (!e.ok || nBlk < 0 || (size_t)nBlk > (e.n - e.p) / 8) return fail("offsets: select index truncated");
        std::vector<uint64_t> blk((size_t)nBlk); for (auto& x : blk) x = e.get<uint64_t>();
        const int32_t nSub = e.get<int32_t>(); if (!e.ok || nSub < 0 || (size_t)nSub > (e.n - e.p) / 2) return fail("offsets: select index truncated");
        std::vector<uint16_t> sub((size_t)nSub); for (auto& x : sub) x = e.get<uint16_t>();
        const int32_t nOvf = e.get<int32_t>(); if (!e.ok || nOvf < 0 || (size_t)nOvf > (e.n - e.p) / 8) return fail("offsets: select index truncated");
        std::vector<int64_t> ovf((size_t)nOvf); for (auto& x : ovf) x = e.get<int64_t>();
        const int32_t cw = e.get<int32_t>(), cc = e.get<int32_t>(), cl = e.get<int32_t>();
        if (!e.ok || cw != l || cc != T || cl < 0 || (size_t)cl != ((size_t)T * (size_t)l + 63) / 64 || (size_t)cl > (e.n - e.p) / 8) return fail("offsets: bad low-bits array");
        const size_t lowAt = e.p; e.skip((size_t)cl * 8);
        if (!e.ok || e.p != n - 24) return fail("offsets: trailing bytes");
        auto word = [&](size_t at, size_t i) { uint64_t v; std::memcpy(&v, b + at + 8 * i, 8); return v; };
        size_t i = 0;
        for (int32_t wv = 0; wv < hbWords; wv++) {
            uint64_t x = word(hbAt, (size_t)wv);
            if (wv == hbWords - 1 && (hbLen & 63)) x &= (1ull << (hbLen & 63)) - 1;
            while (x) {
                const int tz = __builtin_ctzll(x); x &= x - 1;
                const uint64_t pos = (uint64_t)wv * 64 + (uint64_t)tz;
                if (i >= (size_t)T) return fail("offsets: more high bits than values");
                uint64_t low = 0;
                if (l) { const uint64_t bp = (uint64_t)i * (uint64_t)l; const size_t bi = (size_t)(bp >> 6); const int sh = (int)(bp & 63); low = word(lowAt, bi) >> sh; if (sh + l > 64) low |= word(lowAt, bi + 1) << (64 - sh); low &= (1ull << l) - 1; }
                offs[i] = ((pos - i) << l) | low;
                // the select index (DArray.Select) must land on this bit: block inventory for every 1024th, sub-block inventory for every 32nd value
                if ((i & 31) == 0) {
                    const size_t bk = i >> 10;
                    if (bk >= blk.size() || (i >> 5) >= sub.size()) return fail("offsets: select index too short");
                    const bool isOvf = (blk[bk] >> 63) != 0; const uint64_t bp = blk[bk] & 0x7FFFFFFFFFFFFFFFull;
                    const uint64_t sel = isOvf ? ((bp + (i & 1023) < ovf.size()) ? (uint64_t)ovf[bp + (i & 1023)] : ~0ull) : bp + sub[i >> 5];
                    if (sel != pos) return fail("offsets: select index disagrees with the bit vector");
                }
                i++;
            }
        }
        if (i != (size_t)T) return fail("offsets: fewer high bits than values");
        for (int32_t k = 0; k < T; k++) if (offs[k] < (uint64_t)postingsStart || offs[k] + 4 > (uint64_t)fstStart || (k && offs[k] <= offs[k - 1])) return fail("offsets: not ascending inside the postings section");
    }
    // postings
    S.off.assign((size_t)T + 1, 0); S.blocksPerTerm.assign((size_t)T, 0);
    for (int32_t k = 0; k < T; k++) {
        if (!decode_list(b, n, offs[k], (uint64_t)postingsStart, (uint64_t)fstStart, S.docCount, S, S.blocksPerTerm[k], S.error)) return false;
        S.off[k + 1] = S.doc.size();
    }
    return true;
}

inline bool read_file(const char* path, Segment& S) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { S = Segment(); S.error = "cannot open the segment file"; return false; }
    const std::streamoff n = f.tellg(); f.seekg(0);
    std::vector<uint8_t> buf((size_t)n);
    if (n && !f.read((char*)buf.data(), n)) { S = Segment(); S.error = "cannot read the segment file"; return false; }
    return parse(buf.data(), buf.size(), S);
}

} // namespace infs
} // namespace infx
