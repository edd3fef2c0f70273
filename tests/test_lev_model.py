"""The distance-1 fast path of k_stage2's fuzzy matchers (s2_dam1, infidex_amd/csrc/lev.hip.inc) as host code: the SAME source file compiled with g++ and
checked against the banded dynamic programme + the reference's transposition rule (s2_damerau(a, b, 1): LevenshteinDistance.cs:181-341) on every pair of
strings over a three-letter alphabet up to length 6 and on random strings with planted edits.  The device code is checked on the GPU by the parity suites."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_distance_one_fast_path_equals_the_banded_damerau(tmp_path):
    exe = str(tmp_path / "lev_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(HERE, "models", "lev_model.cpp"), "-o", exe])
    out = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout[-2000:] + out.stderr[-2000:]
