"""The register-resident PriorityQueue of the exact replay (RHeap, infidex_amd/csrc/exact3.hip.inc) as a host model: same layout (four siblings
per lane, sorted by (priority, sibling index); levels pinned to registers), same operation, lanes as arrays — checked node by node against the
plain 4-ary heap with the BCL's sift rules on tie-heavy random streams, at every depth where a level boundary moves.  The device code itself is
checked on the GPU (tests/test_gpu_scale.py::test_parallel_exact_replay_equals_the_sequential_kernel, INFX_EX_HEAP_LDS A/B)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_register_heap_model_equals_the_array_heap(tmp_path):
    exe = str(tmp_path / "rheap_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-include", "cstring", os.path.join(HERE, "models", "rheap_model.cpp"), "-o", exe])
    out = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
