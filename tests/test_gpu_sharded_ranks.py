"""Two real ranks (two processes, torch.distributed gloo, both on GPU 0) run the document-sharded phases with their collectives — including the exact
Stage-1 replay across the shards — and the result must be the ORACLE's.
This is the N>1 path of bench.py / infidex_amd/sharded.py minus RCCL (two ranks cannot share one GPU under RCCL); the RCCL tensors path is
covered on one GPU by simulate_shards_dev (test_gpu_parity.py / test_gpu_scale.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RANK_SCRIPT = r'''
import os, sys, numpy as np
import torch, torch.distributed as dist
torch.cuda.init()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from infidex_amd.sharded import create_sharded_engine, ShardedSearcher, TorchComm, index_flat_per_node
from infidex_amd.engine import pack_texts
from tools.synth import Synth
s = Synth(4, docs=140000); arena, offs = s.docs()
# one host-index build for the node: rank 0 indexes and saves the host index, rank 1 reads it back and uploads its own shard
eng = create_sharded_engine(rank, world, 0)
index_flat_per_node(eng, dist.barrier, rank, 4, None, arena, offs, s.field_weights, tag=os.environ["MASTER_PORT"], cache_dir=os.path.dirname(sys.argv[1]))
qa, qo = s.queries(300, qseed=91)
texts = Synth.texts(qa, qo) + ["qu", "", "zzzzqq"]
searcher = ShardedSearcher(eng, TorchComm(dist))
a, o = pack_texts(texts)
batches = [(a, o), pack_texts(texts[:100]), pack_texts(texts[100:])]
res = list(searcher.search_stream(batches, 20, 500))             # three batches in flight: a session, a stream and a communicator each
single = searcher.search_packed(a, o, 20, 500)
# plan exchange on (INFX_PLAN_EXCHANGE=1 here; by default only worlds of >= 4 ranks with few CPUs per rank switch it on: each rank plans its half of the batch — text preparation, term lookups, coverage query contexts — and the blobs are exchanged
# on the planning group) and off (every rank plans the whole batch) must agree
assert eng.device_lookups() and searcher.partition_planning and searcher.native and len(searcher.sessions) == 3
used, peers = searcher.plan_exchange_stats()
nq_ = len(o) - 1; mine_ = nq_ * (rank + 1) // world - nq_ * rank // world
assert used == nq_ and peers == nq_ - mine_, (used, peers, nq_, mine_)      # every plan of the last batch came through the exchange, the peer's half from its blob
off = ShardedSearcher(eng, TorchComm(dist), partition_planning=False, native=False)      # Python-driven phases, every rank plans the whole batch
r_off = off.search_packed(a, o, 20, 500)
for x, y in zip(single, r_off):
    assert np.array_equal(x, y)      # the C++ driver (callbacks into gloo here, RCCL in production) and the Python driver agree bit for bit
if rank == 0:
    k, sc, t, c, f = res[0]
    np.savez(sys.argv[1], k=k, sc=sc, t=t, c=c, f=f, k1=res[1][0], c1=res[1][3], k2=res[2][0], c2=res[2][3], ks=single[0], cs=single[3])
dist.barrier(); dist.destroy_process_group()
'''


def test_two_ranks_equal_the_oracle(tmp_path):
    from infidex_amd.engine import pack_texts
    from tests import oracle_lib as O
    from tests.parity_classify import assert_final_rows_match_oracle
    from tools.synth import Synth
    out = str(tmp_path / "r0.npz")
    env = dict(os.environ); env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["INFX_THREADS"] = "4"; env["INFX_PLAN_EXCHANGE"] = "1"      # (off by default in a world of two: forced on, this test is its GPU coverage)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           "-c", RANK_SCRIPT, out] if False else None
    script = str(tmp_path / "rank.py"); open(script, "w").write(RANK_SCRIPT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631", script, out]
    subprocess.run(cmd, check=True, env=env, timeout=900)
    r = np.load(out)
    s = Synth(4, docs=140000); arena, offs = s.docs()            # 3 containers: rank 0 holds one, rank 1 two
    o = O.OracleEngine.create_default(); o.add_flat(None, arena, offs, s.field_weights); o.finalize()
    qa, qo = s.queries(300, qseed=91)
    texts = Synth.texts(qa, qo) + ["qu", "", "zzzzqq"]
    k, c = r["k"], r["c"]
    same, flips = assert_final_rows_match_oracle(k, r["sc"], c, o, texts, 20, what="2 real ranks")
    print("2 ranks vs oracle:", same, "identical order,", flips, "near-tie flips")
    assert np.array_equal(r["cs"], c) and np.array_equal(r["ks"], k)                      # search_packed == search_stream
    assert np.array_equal(r["k1"], k[:100]) and np.array_equal(r["c1"], c[:100]) and np.array_equal(r["k2"], k[100:]) and np.array_equal(r["c2"], c[100:])


def test_bench_gpus_2():
    """`python bench.py --gpus 2` — started bare, as the driver starts it — runs the document-sharded bench on TWO ranks (gloo here: two ranks cannot share
    one GPU under RCCL) and reports n_gpus == 2 with the same collectives issued on both ranks."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.update({"INFX_DIST_BACKEND": "gloo", "MASTER_PORT": "29641", "INFX_THREADS": "4", "INFX_PLAN_EXCHANGE": "1"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--docs", "140000", "--steps", "2", "--warmup", "1", "--batch", "200", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    cs = d["collectives_per_rank"]
    px = [c.pop("plan_exchange") for c in cs]
    assert len(cs) == 2 and cs[0] == cs[1], cs
    assert all(p["on"] and p["queries_planned_from_exchange"] == 200 and p["of_them_imported_from_peers"] == 100 for p in px), px      # each rank planned half of the last batch


def test_bench_gpus_8():
    """`python bench.py --gpus 8`, started bare as the driver starts it, on EIGHT ranks (gloo: eight ranks share this one GPU) over 600 k documents (10 containers:
    every rank owns one or two): the line reports n_gpus == 8, the eight ranks issued the same collectives, and the plan exchange — on by default in a world of
    eight with two planner threads per rank — fed every query of the last batch, seven eighths of them from peers.  What the driver's first 8-GPU run exercises
    beyond this is RCCL itself."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.update({"INFX_DIST_BACKEND": "gloo", "MASTER_PORT": "29651", "INFX_THREADS": "2"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "INFX_PLAN_EXCHANGE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--docs", "600000", "--steps", "2", "--warmup", "1", "--batch", "160", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    cs = d["collectives_per_rank"]
    px = [c.pop("plan_exchange") for c in cs]
    assert len(cs) == 8 and all(c == cs[0] for c in cs), cs
    if px[0]["on"]:          # (a box whose quota leaves a rank more than four CPUs plans every query everywhere: the rule of ShardedSearcher)
        assert all(p["on"] and p["queries_planned_from_exchange"] == 160 and p["of_them_imported_from_peers"] == 140 for p in px), px
    assert len(d["stage_ms_per_step_per_rank"]) == 8
