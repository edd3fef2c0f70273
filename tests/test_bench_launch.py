"""`python bench.py --gpus N` must run N ranks: started bare it launches itself under torch.distributed.run (one rank per GPU), started by the launcher it
checks WORLD_SIZE against --gpus.  The launch path is exercised here on CPU (gloo, INFX_BENCH_LAUNCH_ONLY=1: ranks, process group, one all-reduce, result
line); the whole sharded bench at W = 2 runs in tests/test_gpu_sharded_ranks.py::test_bench_gpus_2 on a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ); env.update(env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bare_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"INFX_BENCH_LAUNCH_ONLY": "1", "INFX_DIST_BACKEND": "gloo", "MASTER_PORT": "29611"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["ranks_in_all_reduce"] == 2 and d["gpus_arg"] == 2


def test_world_size_must_equal_gpus():
    env = dict(os.environ); env.update({"INFX_BENCH_LAUNCH_ONLY": "1", "INFX_DIST_BACKEND": "gloo"})
    for k in ("RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29612",
                        os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 4 but the launcher started WORLD_SIZE=2" in (r.stderr + r.stdout)
