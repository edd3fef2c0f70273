"""Random text with diacritics, mixed case, every delimiter of ConfigurationParameters.cs:58-62, digits, odd whitespace, repeated words, empty
and one-character documents, duplicate texts — shared by the host-parity (CPU) and GPU-parity tests."""
import random

ALPHABET = list("abcdefghijklmnoprstuvzáčďéěíňóřšťúůýžäöüßÀÉÎÕÜñçøåæœABCDEFGHIJKLMNOPRSTUVZÁČĎÉĚÍŇÓŘŠŤÚŮÝŽ0123456789")
DELIMS = [" ", " ", " ", " ", "-", "/", ".", ",", ":", ";", "'", "`", "–", "—", "*", "&", "\\", "_", "(", ")", "{", "}", "[", "]", "\t",
          " ", "  ", "\n", "§", "!"]


def make(seed, ndocs=400, nqueries=60):
    rng = random.Random(100 + seed)
    vocab = ["".join(rng.choice(ALPHABET) for _ in range(rng.choice([1, 2, 3, 4, 5, 6, 8, 11]))) for _ in range(120)]
    docs = []
    for i in range(ndocs):
        n = rng.choice([0, 1, 1, 3, 5, 8, 13, 40])
        t = "".join(rng.choice(vocab) + rng.choice(DELIMS) for _ in range(n))
        if rng.random() < 0.1 and docs:
            t = docs[rng.randrange(len(docs))][1]          # duplicate text
        if rng.random() < 0.05:
            t = " " + t.upper() + "\t"
        docs.append((i, t))
    queries = []
    for _ in range(nqueries):
        q = "".join(rng.choice(vocab) + rng.choice(DELIMS) for _ in range(rng.choice([1, 2, 3, 5])))
        if rng.random() < 0.3 and len(q) > 3:
            j = rng.randrange(len(q)); q = q[:j] + rng.choice(ALPHABET) + q[j + 1:]
        queries.append(q)
    return docs, queries
