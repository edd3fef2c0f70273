// Deterministic synthetic corpora / query streams for the BASELINE configs.
// Test + bench infrastructure (not part of the product library). PRNG = SplitMix64; every document is generated from
// its own stream seeded by (seed, docId), so the corpus does not depend on the thread count.
//   vocabulary: V words, length uniform 3..10, letters from English unigram frequencies
//   documents : field k has U[minW_k, maxW_k] words, rank ~ Zipf(1.07), single spaces, each word lower/Title case 50/50
//   queries   : sample a doc, pick distinct words of length >= 4, optionally apply one edit (never at position 0)
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>
#include <cmath>
#include <thread>
#include <algorithm>

namespace {
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};
struct Synth {
    uint64_t seed; int V;
    std::vector<std::string> words;
    std::vector<double> cdf;
    uint32_t zipf(Rng& r) const { double u = r.uni() * cdf.back(); return (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()); }
};
const double LETTER_FREQ[26] = {8.167,1.492,2.782,4.253,12.702,2.228,2.015,6.094,6.966,0.153,0.772,4.025,2.406,6.749,7.507,1.929,0.095,5.987,6.327,9.056,2.758,0.978,2.360,0.150,1.974,0.074};

void gen_doc_words(const Synth& S, int64_t doc, int fieldCount, const int* minW, const int* maxW, std::vector<std::vector<uint32_t>>& fw, std::vector<std::vector<uint8_t>>& caps) {
    Rng r(S.seed ^ ((uint64_t)(doc + 1) * 0xD1B54A32D192ED03ull));
    fw.resize(fieldCount); caps.resize(fieldCount);
    for (int f = 0; f < fieldCount; f++) {
        int n = minW[f] + (int)r.below((uint32_t)(maxW[f] - minW[f] + 1));
        fw[f].resize(n); caps[f].resize(n);
        for (int i = 0; i < n; i++) { fw[f][i] = S.zipf(r); caps[f][i] = (uint8_t)(r.next() & 1); }
    }
}
}

extern "C" {

void* synth_create(uint64_t seed, int V) {
    Synth* S = new Synth(); S->seed = seed; S->V = V;
    Rng r(seed * 0x2545F4914F6CDD1Dull + 1);
    double lc[26]; double acc = 0; for (int i = 0; i < 26; i++) { acc += LETTER_FREQ[i]; lc[i] = acc; }
    S->words.resize(V);
    for (int w = 0; w < V; w++) {
        int len = 3 + (int)r.below(8);
        std::string s(len, 'a');
        for (int i = 0; i < len; i++) { double u = r.uni() * acc; int c = 0; while (c < 25 && lc[c] < u) c++; s[i] = (char)('a' + c); }
        S->words[w] = s;
    }
    S->cdf.resize(V); double z = 0;
    for (int i = 0; i < V; i++) { z += 1.0 / std::pow((double)(i + 1), 1.07); S->cdf[i] = z; }
    return S;
}
void synth_destroy(void* h) { delete (Synth*)h; }

// pass 1 (arena == null): fills offs (n*fieldCount+1). pass 2: fills arena using offs.
void synth_docs(void* h, int64_t n, int fieldCount, const int* minW, const int* maxW, uint64_t* offs, uint16_t* arena, int threads) {
    const Synth& S = *(Synth*)h;
    if (threads < 1) threads = 1;
    std::vector<uint32_t> lens;
    if (!arena) lens.resize((size_t)n * fieldCount);
    auto work = [&](int64_t b, int64_t e) {
        std::vector<std::vector<uint32_t>> fw; std::vector<std::vector<uint8_t>> caps;
        for (int64_t d = b; d < e; d++) {
            gen_doc_words(S, d, fieldCount, minW, maxW, fw, caps);
            for (int f = 0; f < fieldCount; f++) {
                if (!arena) {
                    uint32_t L = 0; for (size_t i = 0; i < fw[f].size(); i++) L += (uint32_t)S.words[fw[f][i]].size() + (i ? 1 : 0);
                    lens[(size_t)d * fieldCount + f] = L;
                } else {
                    uint16_t* p = arena + offs[(size_t)d * fieldCount + f];
                    for (size_t i = 0; i < fw[f].size(); i++) {
                        if (i) *p++ = ' ';
                        const std::string& w = S.words[fw[f][i]];
                        for (size_t k = 0; k < w.size(); k++) *p++ = (uint16_t)((k == 0 && caps[f][i]) ? (w[k] - 32) : w[k]);
                    }
                }
            }
        }
    };
    std::vector<std::thread> th; int64_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) { int64_t b = t * per, e = std::min(n, b + per); if (b < e) th.emplace_back(work, b, e); }
    for (auto& x : th) x.join();
    if (!arena) { offs[0] = 0; for (size_t i = 0; i < lens.size(); i++) offs[i + 1] = offs[i] + lens[i]; }
}

// queries: pass 1 (arena == null) fills offs (nq+1); pass 2 fills arena. fields_mask: bit f set => words may come from field f.
// need_field: a field index that must contribute at least one word (or -1).
void synth_queries(void* h, int64_t nq, int64_t nDocs, int fieldCount, const int* minW, const int* maxW, int wordsMin, int wordsMax,
                   double fuzzFrac, int fields_mask, int need_field, uint64_t qseed, uint64_t* offs, uint16_t* arena) {
    const Synth& S = *(Synth*)h;
    std::vector<std::vector<uint32_t>> fw; std::vector<std::vector<uint8_t>> caps;
    uint64_t pos = 0;
    if (!arena) offs[0] = 0;
    for (int64_t qi = 0; qi < nq; qi++) {
        Rng r(qseed ^ ((uint64_t)(qi + 1) * 0xA24BAED4963EE407ull));
        std::vector<std::string> picked;
        for (int attempt = 0; attempt < 64 && picked.empty(); attempt++) {
            int64_t d = (int64_t)(r.next() % (uint64_t)nDocs);
            gen_doc_words(S, d, fieldCount, minW, maxW, fw, caps);
            int want = wordsMin + (int)r.below((uint32_t)(wordsMax - wordsMin + 1));
            std::vector<std::pair<int, uint32_t>> pool;   // (field, word id), distinct words of length >= 4
            for (int f = 0; f < fieldCount; f++) if (fields_mask & (1 << f)) for (uint32_t w : fw[f]) {
                if (S.words[w].size() < 4) continue;
                bool dup = false; for (auto& p : pool) if (p.second == w) { dup = true; break; }
                if (!dup) pool.push_back({f, w});
            }
            if ((int)pool.size() < want) continue;
            std::vector<std::pair<int, uint32_t>> sel;
            if (need_field >= 0) {
                std::vector<size_t> nf; for (size_t i = 0; i < pool.size(); i++) if (pool[i].first == need_field) nf.push_back(i);
                if (nf.empty()) continue;
                size_t k = nf[r.below((uint32_t)nf.size())]; sel.push_back(pool[k]); pool.erase(pool.begin() + k);
            }
            while ((int)sel.size() < want) { size_t k = r.below((uint32_t)pool.size()); sel.push_back(pool[k]); pool.erase(pool.begin() + k); }
            for (auto& s : sel) picked.push_back(S.words[s.second]);
            if (r.uni() < fuzzFrac) {
                std::string& w = picked[r.below((uint32_t)picked.size())];
                int kind = (int)r.below(4); int L = (int)w.size();
                int p = 1 + (int)r.below((uint32_t)(L - 1));
                char c = (char)('a' + r.below(26));
                if (kind == 0) { if (w[p] == c) c = (char)('a' + (c - 'a' + 1) % 26); w[p] = c; }
                else if (kind == 1) w.erase(p, 1);
                else if (kind == 2) w.insert(w.begin() + p, c);
                else { if (p + 1 < L) std::swap(w[p], w[p + 1]); else std::swap(w[p - 1 > 0 ? p - 1 : 1], w[p]); }
            }
        }
        if (picked.empty()) picked.push_back("zzzz");
        std::string q; for (size_t i = 0; i < picked.size(); i++) { if (i) q.push_back(' '); q += picked[i]; }
        if (!arena) offs[qi + 1] = offs[qi] + q.size();
        else { for (char ch : q) arena[pos++] = (uint16_t)(unsigned char)ch; }
    }
}

} // extern "C"
