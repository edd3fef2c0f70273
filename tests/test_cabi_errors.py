"""Error behaviour of the C ABI (include/*.h): no exceptions or crashes across the boundary — status codes + infx_last_error (SURVEY 8b "Errors").
No GPU needed."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from infidex_amd import SearchEngine, LIB_PATH
from infidex_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HANDLE = re.compile(r"infx_(index|stream|engine|session|filter)\s*\*\s*\w+$")


def _handle_functions():
    out = []
    for hdr in ("infidex_hip.h", "infidex_engine.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(int32_t|int64_t)\s+(infx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
            ret, name, args = m.groups()
            params = [a.strip() for a in args.split(",")] if args.strip() and args.strip() != "void" else []
            if params and HANDLE.search(params[0]) and "**" not in params[0]:
                out.append((ret, name, len(params)))
    return out


def test_every_entry_point_rejects_a_null_handle():
    """Every function whose first parameter is a handle returns an error (non-zero status, or -1 for the count-returning ones) for a NULL handle with
    all other arguments zero — in a child process, so that a crash is a test failure and not the end of the test run."""
    fns = _handle_functions()
    assert len(fns) >= 90
    code = r'''
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
bad = []
for spec in sys.argv[2:]:
    ret, name, n = spec.split(":")
    f = getattr(L, name); f.restype = C.c_int64 if ret == "int64_t" else C.c_int32
    sys.stdout.write(name + "\n"); sys.stdout.flush()
    if f(*([None] * int(n))) == 0: bad.append(name)
print("ACCEPTED", bad)
'''
    r = subprocess.run([sys.executable, "-c", code, LIB_PATH] + [f"{a}:{b}:{c}" for a, b, c in fns], capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0, f"crashed in {lines[-1] if lines else '?'} (exit {r.returncode})"
    assert lines[-1] == "ACCEPTED []", lines[-1]


def test_engine_argument_errors_are_status_codes_with_messages():
    e = SearchEngine.create_default(device=-1, threads=1)
    L = e.L
    a = E._u16("alpha beta gamma"); offs = np.asarray([0, len(a)], np.uint64); fw = np.asarray([1], np.int32)
    assert L.infx_engine_index_documents(e.h, C.c_int64(-1), None, None, None, 1, E._p(fw, C.c_int32)) == 1          # INFX_EINVAL
    assert L.infx_engine_index_documents(e.h, C.c_int64(1), None, E._p(a, C.c_uint16), E._p(offs, C.c_uint64), 0, E._p(fw, C.c_int32)) == 1
    e.index_flat(None, a, offs)
    with pytest.raises(E.InfidexError) as ei:
        e.index_flat(None, a, offs)                                     # an engine instance is indexed once
    assert ei.value.code == 1 and "already indexed" in str(ei.value)
    with pytest.raises(E.InfidexError) as ei:
        e.load_index("/nonexistent/file.infdx2")
    assert ei.value.code != 0
    with pytest.raises(E.InfidexError) as ei:
        e.save_host_index("/nonexistent_dir/x.bin")
    assert "cannot create" in str(ei.value)
    assert L.infx_engine_set_shard(e.h, 3, 2) != 0                       # rank >= nranks
    assert e.delete_documents([12345]) == 0                              # unknown DocumentKey: nothing marked, no error (DeleteDocumentsByKey semantics)
    with pytest.raises(E.InfidexError) as ei:
        e.search("alpha")
    assert ei.value.code == 3                                            # INFX_EHIP: the scoring path has no CPU fallback
    msg = C.c_char_p(L.infx_engine_last_error())
    assert msg.value and b"GPU" in msg.value


def test_device_abi_without_a_gpu_reports_ehip():
    L = C.CDLL(LIB_PATH)
    L.infx_last_error.restype = C.c_char_p

    class Cfg(C.Structure):
        _fields_ = [("device", C.c_int32), ("range_docs", C.c_int32), ("max_depth", C.c_int32), ("flags", C.c_int32)]
    idx = C.c_void_p()
    rc = L.infx_create(C.byref(Cfg(0, 0, 500, 0)), C.byref(idx))
    if rc == 0:                                                          # a GPU is present (this test also runs on the GPU box)
        L.infx_destroy(idx)
        return
    assert rc == 3 and b"no CPU fallback" in L.infx_last_error()
    assert L.infx_create(None, C.byref(idx)) == 1 and L.infx_create(C.byref(Cfg(0, 0, 500, 0)), None) == 1
    L.infx_destroy(None); L.infx_stream_destroy(None); L.infx_filter_destroy(None)      # destroying nothing is a no-op
