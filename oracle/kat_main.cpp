// ORACLE — TEST INFRASTRUCTURE ONLY. Quick CLI: index the reference's 10-doc corpus and print results.
#include "pipeline.hpp"
#include <cstdio>
using namespace orc;
int main(int argc, char** argv) {
    Engine e;
    const char* docs[] = {
        "The quick brown fox jumps over the lazy dog",
        "A journey of a thousand miles begins with a single step",
        "To be or not to be, that is the question",
        "All that glitters is not gold",
        "The fox was quick and clever in the forest",
        "Batman and Robin fight crime in Gotham City",
        "Superman flies faster than a speeding bullet",
        "Spider-Man swings through New York City",
        "Wonder Woman protects the innocent",
        "The Flash runs at incredible speeds"};
    for (int i = 0; i < 10; i++) e.add_document(i + 1, utf8_to_u16(docs[i]));
    e.finalize();
    e.keepTrace = true;
    const char* qs[] = {"batman", "qick fux", "battamam", "new york", "speeding", "fox", "quik fox"};
    for (const char* q : qs) {
        if (argc > 1 && std::string(argv[1]) != q) continue;
        QueryParams qp;
        SearchOutput o = e.search(utf8_to_u16(q), qp);
        printf("query '%s' -> %zu results (stage1 %zu, cov %d):", q, o.records.size(), o.stage1.size(), (int)o.usedCoverage);
        for (auto& r : o.records) printf(" [%lld %.4f t%d]", (long long)r.key, r.score, r.tie);
        printf("\n");
        if (argc > 1) {
            for (auto& s : o.stage1) printf("   s1 key=%lld score=%.5f\n", (long long)s.key, s.score);
            for (auto& t : o.trace) printf("   cand id=%d base=%.4f score=%.4f tie=%d wh=%d any=%d/%d full=%d strict=%d pref=%d first=%d lcs=%d cov=%d sumCi=%.3f\n",
                t.internalId, t.baseScore, t.score, t.tie, t.f.WordHits, t.f.TermsWithAnyMatch, t.f.TermsCount, t.f.TermsFullyMatched, t.f.TermsStrictMatched, t.f.TermsPrefixMatched, t.f.FirstMatchIndex, t.lcs, t.f.CoverageScore, t.f.SumCi);
        }
    }
    return 0;
}
