"""ctypes wrapper of tools/synth.cpp — deterministic synthetic corpora for the BASELINE configs."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "synth.cpp")
LIB = os.path.join(HERE, "_build", "libsynth.so")
SEED0 = 0x1F1DE5

# (n_docs, vocab, fields[(minW, maxW, weight)], query words (min,max), fuzz fraction, need_field, top-k)
CONFIGS = {
    2: dict(docs=100_000, vocab=50_000, fields=[(4, 12, 1)], qwords=(2, 2), fuzz=0.0, need_field=-1, k=10),
    3: dict(docs=1_000_000, vocab=200_000, fields=[(2, 6, 0), (8, 24, 2)], qwords=(3, 3), fuzz=1.0, need_field=0, k=20),
    4: dict(docs=10_000_000, vocab=1_000_000, fields=[(4, 12, 1)], qwords=(2, 3), fuzz=0.3, need_field=-1, k=20),
    # config 5 = config 4 + non-indexed fields year / rating / genre, Filter.Parse("year >= 2000 AND rating > 7.0"), EnableFacets
    5: dict(docs=10_000_000, vocab=1_000_000, fields=[(4, 12, 1)], qwords=(2, 3), fuzz=0.3, need_field=-1, k=20, filter="year >= 2000 AND rating > 7.0"),
}
GENRES = ["Action", "Comedy", "Drama", "Horror", "Sci-Fi", "Romance", "Thriller", "Western", "Fantasy", "Mystery", "Crime", "Animation"]


def config5_columns(n, seed=0x1F1DE5 + 5):
    """year in U[1950, 2024] (int), rating in U[1.0, 10.0] (1 decimal), genre in 12 values; year and genre facetable."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return rng.integers(1950, 2025, n).astype(np.int64), np.round(rng.uniform(1.0, 10.0, n), 1), [GENRES[i] for i in rng.integers(0, len(GENRES), n)]


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRC, "-lpthread"])
    return LIB


class Synth:
    def __init__(self, config: int, docs: int = None, vocab: int = None, threads: int = None):
        cfg = dict(CONFIGS[config])
        if docs is not None:
            # scaled-down variants keep the vocabulary/doc ratio of the config
            cfg["vocab"] = vocab if vocab is not None else max(1000, int(cfg["vocab"] * docs / cfg["docs"]))
            cfg["docs"] = docs
        self.cfg = cfg
        self.config = config
        self.L = C.CDLL(build())
        self.L.synth_create.restype = C.c_void_p
        self.h = C.c_void_p(self.L.synth_create(C.c_uint64(SEED0 + config), cfg["vocab"]))
        self.threads = threads or os.cpu_count() or 1
        self.minW = np.asarray([f[0] for f in cfg["fields"]], np.int32)
        self.maxW = np.asarray([f[1] for f in cfg["fields"]], np.int32)
        self.field_weights = [f[2] for f in cfg["fields"]]

    def docs(self):
        n, fc = self.cfg["docs"], len(self.cfg["fields"])
        offs = np.zeros(n * fc + 1, np.uint64)
        ip = C.POINTER(C.c_int32); up = C.POINTER(C.c_uint64); hp = C.POINTER(C.c_uint16)
        self.L.synth_docs(self.h, C.c_int64(n), fc, self.minW.ctypes.data_as(ip), self.maxW.ctypes.data_as(ip), offs.ctypes.data_as(up), None, self.threads)
        arena = np.zeros(max(int(offs[-1]), 1), np.uint16)
        self.L.synth_docs(self.h, C.c_int64(n), fc, self.minW.ctypes.data_as(ip), self.maxW.ctypes.data_as(ip), offs.ctypes.data_as(up), arena.ctypes.data_as(hp), self.threads)
        return arena, offs

    def queries(self, nq, qseed=1, fuzz=None):
        cfg = self.cfg
        n, fc = cfg["docs"], len(cfg["fields"])
        offs = np.zeros(nq + 1, np.uint64)
        ip = C.POINTER(C.c_int32); up = C.POINTER(C.c_uint64); hp = C.POINTER(C.c_uint16)
        fz = cfg["fuzz"] if fuzz is None else fuzz
        args = (self.h, C.c_int64(nq), C.c_int64(n), fc, self.minW.ctypes.data_as(ip), self.maxW.ctypes.data_as(ip), cfg["qwords"][0], cfg["qwords"][1],
                C.c_double(fz), (1 << fc) - 1, cfg["need_field"], C.c_uint64(qseed))
        self.L.synth_queries(*args, offs.ctypes.data_as(up), None)
        arena = np.zeros(max(int(offs[-1]), 1), np.uint16)
        self.L.synth_queries(*args, offs.ctypes.data_as(up), arena.ctypes.data_as(hp))
        return arena, offs

    @staticmethod
    def texts(arena, offs):
        return [arena[int(offs[i]):int(offs[i + 1])].tobytes().decode("utf-16-le") for i in range(len(offs) - 1)]
