// Node-local cache of the HOST index (host/index.h: HostIndex).  Document-sharded operation runs one process per GPU and every process needs the whole
// host index (global df / avgdl / N, the term and word dictionaries, prefix populations, the WordMatcher lists: SURVEY 8e "replicated on every GPU"); W
// processes building it at once on the node's shared cores take W times the CPU work.  Instead the node's leader builds it once with every core and
// writes it here (a file under /dev/shm); the other processes read it back — plain arrays, a couple of seconds — and upload their own shard.
// The file is a transient hand-off between processes of ONE build on ONE node: raw little-endian arrays behind a header that pins the layout version,
// the configuration the index was built with, the index fingerprint and a checksum of the payload; a truncated, foreign or corrupted file is refused (the checksum is an unkeyed multiply-xor hash: it detects damage, not tampering — the security boundary is the private 0700 directory with O_EXCL | O_NOFOLLOW and the ownership check).
// The writer creates its file exclusively (O_EXCL | O_NOFOLLOW, mode 0600): a link somebody planted at the predictable path of a world-writable
// directory is not written through.  It is not the reference's INFDX2 format (host/infdx2.h reads that).
#pragma once
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include "index.h"

namespace hostcache {
using namespace infx;
static const char MAGIC[8] = {'I', 'N', 'F', 'X', 'H', 'I', 'C', '1'};
static const char TAIL[8] = {'1', 'C', 'I', 'H', 'X', 'F', 'N', 'I'};

// checksum of the payload: a multiply-xor hash over 8-byte words, chunk by chunk as the arrays stream through (the chunk boundaries are the same for
// writer and reader: both walk the same ar() sequence)
struct Sum {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    void add(const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p; uint64_t x = h; size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); x = (x ^ w) * 0xFF51AFD7ED558CCDull; x ^= x >> 29; }
        uint64_t w = 0; if (i < n) std::memcpy(&w, b + i, n - i);
        x = (x ^ w ^ ((uint64_t)n << 56)) * 0xC4CEB9FE1A85EC53ull; x ^= x >> 32;
        h = x;
    }
};
struct Writer {
    FILE* f; bool ok = true; uint64_t bytes = 0; Sum sum;
    void raw(const void* p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; bytes += n; sum.add(p, n); }
    template <class T> void pod(T& v) { raw(&v, sizeof(T)); }
    template <class T> void vec(std::vector<T>& v) { uint64_t n = v.size(); pod(n); raw(v.data(), (size_t)n * sizeof(T)); }
};
struct Reader {
    FILE* f; uint64_t left; bool ok = true; Sum sum;      // left: bytes the file still holds (no length field can ask for more)
    void raw(void* p, size_t n) { if (!ok) return; if (n > left || (n && fread(p, 1, n, f) != n)) { ok = false; return; } left -= n; sum.add(p, n); }
    template <class T> void pod(T& v) { raw(&v, sizeof(T)); }
    template <class T> void vec(std::vector<T>& v) { uint64_t n = 0; pod(n); if (!ok || n > left / sizeof(T)) { ok = false; return; } v.resize((size_t)n); raw(v.data(), (size_t)n * sizeof(T)); }
};
template <class A> void ar(A& a, KeyTable& k) { a.vec(k.arena); a.vec(k.keyOff); a.vec(k.keyLen); a.vec(k.slotHash); a.vec(k.slotId); a.pod(k.mask); }
template <class A> void ar(A& a, Csr& c) { ar(a, c.keys); a.vec(c.off); a.vec(c.doc); a.vec(c.w); a.vec(c.meta); }
template <class A> void ar(A& a, HostIndex& ix) {
    a.pod(ix.N); a.vec(ix.docKey); a.vec(ix.docLen); a.pod(ix.avgdl); a.vec(ix.textOff); a.vec(ix.text);
    ar(a, ix.terms); a.vec(ix.df); a.vec(ix.sortedTerms);
    a.vec(ix.trie); a.vec(ix.edgeStart); a.vec(ix.edgeLabel); a.vec(ix.edgeChild);
    a.vec(ix.rEdgeStart); a.vec(ix.rEdgeLabel); a.vec(ix.rEdgeChild); a.vec(ix.rTerm);
    ar(a, ix.prefixKeys); a.vec(ix.prefixPop); a.vec(ix.prefixSetId); a.vec(ix.psOff); a.vec(ix.psDocs);
    ar(a, ix.wmExact); ar(a, ix.wmLd1); a.vec(ix.affixFwd); a.vec(ix.affixRev);
    ar(a, ix.words); a.vec(ix.wordDf); a.vec(ix.wordLastDoc); a.vec(ix.wordIdf);
    ar(a, ix.icWords); a.vec(ix.icWordIdf);
}
// what the arrays depend on besides the documents: FNV-1a over the configuration fields build_index reads (threads excluded: the build is
// deterministic in the thread count) and the synonym pairs
inline uint64_t config_signature(const HostConfig& c, uint64_t synHash) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    const int32_t iv[] = {c.ngram, c.startPad, c.stopPad, c.stopTermLimit, c.enableCoverage ? 1 : 0, c.wordMatcher ? 1 : 0, c.wmMinExact, c.wmMaxExact, c.wmMinLD1, c.wmMaxLD1, c.maxDepth,
                          (int32_t)sizeof(HostIndex::TrieNode), (int32_t)sizeof(size_t)};
    mix(iv, sizeof(iv)); mix(c.fieldWeights, sizeof(c.fieldWeights)); mix(&synHash, sizeof(synHash));
    return h;
}
struct Header { char magic[8]; uint32_t version, keysAreIds; uint64_t configSig, fingerprint, payloadBytes, payloadSum; };
constexpr uint32_t VERSION = 3;      // 3: OrdinalIgnoreCase word classes (icWords)

// returns an empty string on success, else what went wrong
inline std::string save(const char* path, HostIndex& ix, bool keysAreIds, uint64_t configSig, uint64_t fingerprint) {
    const std::string tmp = std::string(path) + ".tmp";
    unlink(tmp.c_str());                                   // a stale file of a crashed run (unlink does not follow a link)
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return "cannot create " + tmp + " (exclusive creation: " + std::strerror(errno) + ")";
    FILE* f = fdopen(fd, "wb");
    if (!f) { close(fd); unlink(tmp.c_str()); return "cannot create " + tmp; }
    Header h{}; std::memcpy(h.magic, MAGIC, 8); h.version = VERSION; h.keysAreIds = keysAreIds ? 1u : 0u; h.configSig = configSig; h.fingerprint = fingerprint; h.payloadBytes = 0;
    Writer w{f};
    w.raw(&h, sizeof(h));
    w.sum = Sum{};                                          // the checksum covers the payload only
    ar(w, ix);
    const uint64_t payload = w.bytes - sizeof(h), psum = w.sum.h;
    w.raw(TAIL, 8);
    bool ok = w.ok;
    if (ok && fseek(f, 0, SEEK_SET) == 0) { h.payloadBytes = payload; h.payloadSum = psum; ok = fwrite(&h, 1, sizeof(h), f) == sizeof(h); } else ok = false;
    if (fclose(f) != 0) ok = false;
    if (!ok) { remove(tmp.c_str()); return "short write to " + tmp + " (is the file system full?)"; }
    if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return std::string("cannot rename the cache file to ") + path; }
    return "";
}
inline std::string load(const char* path, HostIndex& ix, bool& keysAreIds, uint64_t configSig, uint64_t (*fingerprintOf)(const HostIndex&)) {
    const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    FILE* f = fd < 0 ? nullptr : fdopen(fd, "rb");
    if (!f) { if (fd >= 0) close(fd); return std::string("cannot open ") + path; }
    std::string err;
    const HostConfig keep = ix.cfg;
    do {
        if (fseek(f, 0, SEEK_END) != 0) { err = "cannot seek"; break; }
        const long long size = ftell(f);
        if (size < (long long)(sizeof(Header) + 8) || fseek(f, 0, SEEK_SET) != 0) { err = "file too short for a host-index cache"; break; }
        Header h{};
        if (fread(&h, 1, sizeof(h), f) != sizeof(h) || std::memcmp(h.magic, MAGIC, 8) != 0) { err = "not a host-index cache (magic)"; break; }
        if (h.version != VERSION) { err = "host-index cache of another layout version"; break; }
        if (h.configSig != configSig) { err = "host-index cache was built with another configuration (n-gram / stop-term / WordMatcher / synonym settings must match on every rank)"; break; }
        if (h.payloadBytes != (uint64_t)size - sizeof(Header) - 8) { err = "host-index cache is truncated"; break; }
        Reader r{f, h.payloadBytes};
        ar(r, ix);
        char tail[8];
        if (!r.ok || r.left != 0 || fread(tail, 1, 8, f) != 8 || std::memcmp(tail, TAIL, 8) != 0) { err = "host-index cache is corrupt (lengths do not add up)"; break; }
        if (r.sum.h != h.payloadSum) { err = "host-index cache is corrupt (payload checksum)"; break; }
        if (fingerprintOf(ix) != h.fingerprint) { err = "host-index cache does not match its own fingerprint"; break; }
        // cheap structural checks before anything indexes into the arrays
        const size_t N = (size_t)ix.N, T = ix.terms.K();
        if (ix.N < 0 || ix.docKey.size() != N || ix.docLen.size() != N || ix.textOff.size() != N + 1 || ix.terms.off.size() != T + 1 || ix.df.size() != T ||
            ix.terms.off.back() != ix.terms.doc.size() || ix.terms.w.size() != ix.terms.doc.size() || ix.textOff.back() != ix.text.size() ||
            ix.psOff.empty() || ix.psOff.back() != ix.psDocs.size()) { err = "host-index cache is inconsistent"; break; }
        keysAreIds = h.keysAreIds != 0;
    } while (false);
    fclose(f);
    if (!err.empty()) { ix = HostIndex{}; ix.cfg = keep; }
    return err;
}
}   // namespace hostcache
