"""world_size-2 gloo test (CPU) of the collective plumbing of the sharded path: count all-reduce, top-k all-gather and the
all-reduce of disjoint Stage-2 records behave as infidex_amd/sharded.py assumes (the phases themselves need a GPU and are
covered by tests/test_gpu_parity.py::test_sharded_equals_unsharded)."""
import os
import socket
import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from infidex_amd.sharded import TorchComm
    c = TorchComm(dist)
    nd, depth = 5, 7
    rng = np.random.default_rng(100 + rank)
    # Exchange 1: histograms add up
    counts = rng.integers(0, 1000, (nd, 136)).astype(np.uint32)
    g = c.allreduce_sum_i32(counts)
    # Exchange 2: per-shard top-k (doc, score bits) gathered in rank order
    hits = np.zeros((nd, depth, 2), np.int32)
    hits[..., 0] = rng.integers(rank * 1000, (rank + 1) * 1000, (nd, depth))
    hits[..., 1] = rng.random((nd, depth)).astype(np.float32).view(np.int32)
    ah = c.allgather(hits)
    hc = np.full(nd, depth - rank, np.uint32)
    ahc = c.allgather(hc)
    # disjoint records: every candidate is owned by exactly one rank
    ncand = 11
    outs = np.zeros((ncand, 3), np.int32)
    own = np.arange(ncand) % world == rank
    outs[own] = rng.integers(-2**31, 2**31 - 1, (int(own.sum()), 3), dtype=np.int64).astype(np.int32)
    m = c.allreduce_sum_i32(outs)
    q.put((rank, counts, g, hits, ah, ahc, outs, m))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_collectives():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in ps:
        p.join(60)
    (r0, c0, g0, h0, ah0, ahc0, o0, m0), (r1, c1, g1, h1, ah1, ahc1, o1, m1) = got
    assert np.array_equal(g0, c0 + c1) and np.array_equal(g1, g0)
    assert np.array_equal(ah0[0], h0) and np.array_equal(ah0[1], h1) and np.array_equal(ah1, ah0)
    assert ahc0.tolist() == [[7] * 5, [6] * 5]
    # merged = own record from whichever rank owned it, bit-exact
    exp = np.where((np.arange(11) % 2 == 0)[:, None], o0, o1)
    assert np.array_equal(m0, exp) and np.array_equal(m1, exp)
    # the merge rule of phase 3: best `depth` of the union in (score desc, doc asc) order
    sc = ah0[..., 1].view(np.float32)
    for j in range(5):
        pairs = [(float(sc[w, j, k]), int(ah0[w, j, k, 0])) for w in range(2) for k in range(int(ahc0[w, j]))]
        top = sorted(pairs, key=lambda x: (-x[0], x[1]))[:7]
        assert len(top) == 7 and all(top[i][0] >= top[i + 1][0] for i in range(6))
