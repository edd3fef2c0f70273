"""infidex_amd — MI355X-native drop-in for the query-time scoring hot path of lofcz/Infidex.

Python is plumbing only (ctypes over libinfidex_hip.so); the product is the HIP kernels + the C ABI
(include/infidex_hip.h) + the C++ host engine (include/infidex_engine.h). There is no CPU scoring path:
Search raises if the HIP extension or a GPU is missing.
"""
from .engine import (SearchEngine, Session, Query, Document, Field, Weight, Result, ScoreEntry, InfidexError,
                     load_library, LIB_PATH)

__all__ = ["SearchEngine", "Session", "Query", "Document", "Field", "Weight", "Result", "ScoreEntry", "InfidexError",
           "load_library", "LIB_PATH"]
