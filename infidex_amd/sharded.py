"""Document-sharded search across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests of the plumbing).
Every rank indexes the whole corpus on the host (global df / avgdl / N, exactly like the reference's single index), uploads
its contiguous doc range, and runs each batch as four phases of the C++ engine with three small collectives in between:

    phase0  text prep, LD1 member lists, k_union_count -> all-reduce(sum)  |union| of new fuzzy virtual terms (Exchange 1b)
    phase1  idf/roles + k_accumulate                 -> all-reduce(sum)  class histograms   (Exchange 1: tier decisions, quirk Q11)
    phase2  k_select with the GLOBAL counts          -> all-gather       per-shard top-`depth` (Exchange 2: the north-star collective)
    phase3  merge, Stage-2 prep, k_stage2 on OWNED candidates -> all-reduce(sum) of the disjoint 12-byte records
    phase4  final ordering / truncation (identical on every rank)
"""
import ctypes as C
from typing import List, Sequence

import numpy as np

from .engine import SearchEngine, Session, _p, INFX_NFEAT  # noqa: F401

INFX_NCLASS = 136


class TorchComm:
    """numpy <-> torch.distributed adaptor. With backend nccl the tensors live on this rank's GPU (RCCL), with gloo on the CPU."""

    def __init__(self, dist, device=None):
        import torch
        self.torch = torch
        self.dist = dist
        # torch's CPU ops would otherwise wake an OpenMP team as wide as the host (256 threads here) that spins after every tiny
        # copy: under a container CPU quota that burns the budget and the whole process is throttled for tens of milliseconds
        try:
            torch.set_num_threads(1)
        except Exception:
            pass
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                         if dist.get_backend() == "nccl" else torch.device("cpu"))

    def allreduce_sum_i32(self, a: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(a.view(np.int32))).clone().to(self.device)   # never alias the caller's array
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().view(a.dtype).reshape(a.shape)

    def allgather(self, a: np.ndarray) -> np.ndarray:
        """Returns an array of shape (world, *a.shape); a must have the same shape on every rank."""
        t = self.torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(self.device)
        out = self.torch.empty(self.world * t.numel(), dtype=t.dtype, device=self.device)   # flat: accepted by both RCCL and gloo
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().view(a.dtype).reshape((self.world,) + a.shape)

    def planning_group(self):
        """A second communicator (always gloo, CPU tensors) for the planner thread's exchange: its collectives are issued from another
        thread than the batch collectives, so they must not share a communicator with them.  Collective: every rank calls it once."""
        if not hasattr(self, "_plan_group"):
            self._plan_group = self.dist.new_group(backend="gloo")
        return self._plan_group

    def allgather_bytes(self, blob: np.ndarray, group=None):
        """All-gather of variable-length byte strings (uint8 arrays) on `group`; returns the list of every rank's bytes."""
        torch = self.torch
        n = torch.tensor([blob.size], dtype=torch.int64)
        sizes = torch.zeros(self.world, dtype=torch.int64)
        self.dist.all_gather_into_tensor(sizes, n, group=group)
        m = int(sizes.max().item())
        pad = torch.zeros(max(m, 1), dtype=torch.uint8)
        if blob.size:
            pad[:blob.size] = torch.from_numpy(blob)
        out = torch.empty(self.world * max(m, 1), dtype=torch.uint8)
        self.dist.all_gather_into_tensor(out, pad, group=group)
        o = out.numpy().reshape(self.world, max(m, 1))
        return [o[r, :int(sizes[r].item())] for r in range(self.world)]

    def max_i64(self, v: int) -> int:
        t = self.torch.tensor([v], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())


def create_sharded_engine(rank: int, world: int, device: int, **kw) -> SearchEngine:
    eng = SearchEngine.create_default(device=device, **kw)
    eng._check(eng.L.infx_engine_set_shard(eng.h, rank, world))
    return eng


class ShardSession:
    """Phase-level access to one engine session (used by ShardedSearcher and by the single-process shard simulation)."""

    def __init__(self, engine: SearchEngine):
        self.e = engine
        self.L = engine.L
        self.s = Session(engine)

    def prefetch_collect(self, arena, offs, begin, end, depth):
        """This rank's share of the batch's index-wide host lookups (LD1 expansions, WordMatcher descriptors of queries [begin, end)) as bytes."""
        nq = len(offs) - 1
        self.L.infx_session_prefetch_collect.restype = C.c_int64
        n = self.L.infx_session_prefetch_collect(self.s.h, nq, _p(arena, C.c_uint16), _p(offs, C.c_uint64), begin, end, depth)
        if n < 0:
            self.e._check(-1)
        blob = np.zeros(max(int(n), 1), np.uint8)
        self.e._check(self.L.infx_session_prefetch_blob(self.s.h, _p(blob, C.c_uint8), C.c_int64(blob.size)))
        return blob[:int(n)]

    def prefetch_import(self, blob):
        b = np.ascontiguousarray(blob, np.uint8)
        self.e._check(self.L.infx_session_prefetch_import(self.s.h, _p(b, C.c_uint8), C.c_int64(b.size)))

    def phase0(self, arena, offs, depth):
        nu = C.c_uint32(0)
        self.nq = len(offs) - 1
        self.depth = depth
        self.e._check(self.L.infx_session_phase0(self.s.h, self.nq, _p(arena, C.c_uint16), _p(offs, C.c_uint64), depth, C.byref(nu)))
        uc = np.zeros(max(nu.value, 1), np.uint32)
        if nu.value:
            self.e._check(self.L.infx_session_union_counts(self.s.h, _p(uc, C.c_uint32)))
        return uc[:nu.value]

    def phase1(self, global_union_counts):
        nd = C.c_uint32(0)
        guc = np.ascontiguousarray(global_union_counts, np.uint32)
        if guc.size == 0:
            guc = np.zeros(1, np.uint32)
        self.e._check(self.L.infx_session_phase1(self.s.h, _p(guc, C.c_uint32), C.byref(nd)))
        self.nd = nd.value
        counts = np.zeros((max(self.nd, 1), INFX_NCLASS), np.uint32)
        if self.nd:
            self.e._check(self.L.infx_session_counts(self.s.h, _p(counts, C.c_uint32)))
        return counts[:self.nd]

    def phase2(self, global_counts):
        hits = np.zeros((max(self.nd, 1), self.depth, 2), np.int32)     # (doc:int32, score bits:int32)
        hc = np.zeros(max(self.nd, 1), np.uint32)
        gc = np.ascontiguousarray(global_counts, np.uint32)
        if gc.size == 0:
            gc = np.zeros((1, INFX_NCLASS), np.uint32)
        self.e._check(self.L.infx_session_phase2(self.s.h, _p(gc, C.c_uint32), hits.ctypes.data_as(C.c_void_p), _p(hc, C.c_uint32)))
        return hits[:self.nd], hc[:self.nd]

    def phase3(self, all_hits, all_hc, max_results, enable_coverage=True):
        W = all_hits.shape[0]
        ah = np.ascontiguousarray(all_hits, np.int32)
        ac = np.ascontiguousarray(all_hc, np.uint32)
        if ah.size == 0:
            ah = np.zeros((W, 1, self.depth, 2), np.int32); ac = np.zeros((W, 1), np.uint32)
        ncand = C.c_uint64(0)
        self.max_results = max_results
        self.e._check(self.L.infx_session_phase3(self.s.h, W, ah.ctypes.data_as(C.c_void_p), _p(ac, C.c_uint32), max_results, int(enable_coverage), C.byref(ncand)))
        self.ncand = ncand.value
        outs = np.zeros((max(self.ncand, 1), 3), np.int32)
        if self.ncand:
            self.e._check(self.L.infx_session_outs(self.s.h, _p(outs, C.c_int32)))
        return outs[:self.ncand]

    # ---- device-resident exchange buffers (torch CUDA tensors; RCCL works on them in place) ----
    def phase1_dev(self, global_union_counts, counts_t):
        """phase 1 with the class histograms written into counts_t (nq x INFX_NCLASS int32 CUDA tensor): Exchange 1 all-reduces it in place."""
        nd = C.c_uint32(0)
        guc = np.ascontiguousarray(global_union_counts, np.uint32)
        if guc.size == 0:
            guc = np.zeros(1, np.uint32)
        self.e._check(self.L.infx_session_phase1x(self.s.h, _p(guc, C.c_uint32), C.c_void_p(counts_t.data_ptr()), C.byref(nd)))
        self.nd = nd.value

    def phase2_dev(self, global_counts, hits_t, hc_t):
        if hasattr(global_counts, "data_ptr"):       # device tensor: read back from HBM by the engine, no host round trip
            self.e._check(self.L.infx_session_phase2x(self.s.h, C.c_void_p(global_counts.data_ptr()), C.c_void_p(hits_t.data_ptr()), C.c_void_p(hc_t.data_ptr())))
            return
        gc = np.ascontiguousarray(global_counts, np.uint32)
        if gc.size == 0:
            gc = np.zeros((1, INFX_NCLASS), np.uint32)
        self.e._check(self.L.infx_session_phase2x(self.s.h, _p(gc, C.c_uint32), C.c_void_p(hits_t.data_ptr()), C.c_void_p(hc_t.data_ptr())))

    def phase3_dev(self, all_hits_t, all_hc_t, outs_t, max_results, enable_coverage=True):
        W = all_hits_t.shape[0]
        self.max_results = max_results
        self.e._check(self.L.infx_session_phase3x(self.s.h, W, C.c_void_p(all_hits_t.data_ptr()), C.c_void_p(all_hc_t.data_ptr()), max_results, int(enable_coverage),
                                                  C.c_void_p(outs_t.data_ptr())))

    def phase4_dev(self, merged_t):
        nq, mr = self.nq, self.max_results
        keys = np.full((nq, mr), -1, np.int64); scores = np.zeros((nq, mr), np.float32)
        ties = np.zeros((nq, mr), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
        self.e._check(self.L.infx_session_phase4(self.s.h, C.c_void_p(merged_t.data_ptr()), _p(keys, C.c_int64), _p(scores, C.c_float), _p(ties, C.c_uint8),
                                                 _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
        return keys, scores, ties, counts, flags

    def phase4(self, merged_outs):
        nq, mr = self.nq, self.max_results
        keys = np.full((nq, mr), -1, np.int64); scores = np.zeros((nq, mr), np.float32)
        ties = np.zeros((nq, mr), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
        mo = np.ascontiguousarray(merged_outs, np.int32)
        if mo.size == 0:
            mo = np.zeros((1, 3), np.int32)
        self.e._check(self.L.infx_session_phase4(self.s.h, _p(mo, C.c_int32), _p(keys, C.c_int64), _p(scores, C.c_float), _p(ties, C.c_uint8),
                                                 _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
        return keys, scores, ties, counts, flags


class ShardedSearcher:
    """One rank of a document-sharded deployment. All ranks must call search_packed / search_stream with the same batches."""

    def __init__(self, engine: SearchEngine, comm: TorchComm, partition_planning: bool = True):
        self.partition_planning = partition_planning and comm.world > 1
        self.plan_group = comm.planning_group() if self.partition_planning else None      # collective: same call on every rank
        self.sessions = [ShardSession(engine), ShardSession(engine)]
        self.sess = self.sessions[0]
        self.comm = comm
        self.last = self.sess

    def _prefetch(self, s, arena, offs, depth):
        """Sharded planning: this rank runs the expensive index-wide host lookups (LD1 expansion of unknown words, WordMatcher descriptors —
        ~85 % of the host time per query at 10 M documents) for its 1/W slice of the batch only; the ranks all-gather the results and import
        each other's before phase 0.  Results are unchanged: every rank holds the whole host index, the lookups are pure functions of the text."""
        c = self.comm
        if c.world <= 1 or not self.partition_planning:
            return
        nq = len(offs) - 1
        begin, end = nq * c.rank // c.world, nq * (c.rank + 1) // c.world
        mine = s.prefetch_collect(arena, offs, begin, end, depth)
        for r, b in enumerate(c.allgather_bytes(mine, group=self.plan_group)):
            if r != c.rank and b.size:
                s.prefetch_import(b)

    def search_packed(self, arena, offs, max_results=10, depth=500, enable_coverage=True):
        s = self.sessions[0]
        self._prefetch(s, arena, offs, depth)
        uc = s.phase0(arena, offs, depth)
        return self._finish(s, uc, max_results, depth, enable_coverage)

    def search_stream(self, batches, max_results=10, depth=500, enable_coverage=True, stamps=None):
        """Pipelined stream of batches [(arena, offs), ...]: a planner thread runs phase 0 of batch i+1 (text preparation, LD1
        expansion, k_union — host work and kernels on the OTHER session's stream, no collective) while this thread drives the
        collective phases of batch i.  Collectives are issued by this thread only, in batch order, so their order is identical on
        every rank.  Yields the results in order; stamps (optional list) receives (t_plan_start, t_done) per batch."""
        import queue
        import threading
        import time
        free = [threading.Semaphore(1), threading.Semaphore(1)]
        q = queue.Queue(maxsize=2)

        def planner():
            for i, (a, o) in enumerate(batches):
                j = i % 2
                free[j].acquire()
                t0 = time.time()
                try:
                    self._prefetch(self.sessions[j], a, o, depth)
                    q.put((j, self.sessions[j].phase0(a, o, depth), t0, None))
                except Exception as ex:      # surfaced by the consumer
                    q.put((j, None, t0, ex))
                    return

        th = threading.Thread(target=planner, daemon=True)
        th.start()
        for _ in range(len(batches)):
            j, uc, t0, ex = q.get()
            if ex is not None:
                raise ex
            res = self._finish(self.sessions[j], uc, max_results, depth, enable_coverage)
            if stamps is not None:
                stamps.append((t0, time.time()))
            free[j].release()
            yield res
        th.join()

    def _finish(self, s, uc, max_results, depth, enable_coverage):
        import os, time
        c = self.comm
        self.last = s
        dbg = os.environ.get("INFX_DEBUG") is not None
        T = [time.time()]
        def mark():
            if dbg: T.append(time.time())
        guc = c.allreduce_sum_i32(uc) if uc.size else uc                                          # Exchange 1b: df of new fuzzy unions
        if c.device.type == "cuda" and c.dist.get_backend() == "nccl":
            # RCCL path: class histograms, hit lists and Stage-2 rows stay in HBM; every collective runs in place on the tensor the kernels wrote
            torch = c.torch; nq = s.nq
            counts_t = torch.zeros((max(nq, 1), INFX_NCLASS), dtype=torch.int32, device=c.device)
            torch.cuda.current_stream().synchronize()
            mark(); s.phase1_dev(guc, counts_t); mark()
            c.dist.all_reduce(counts_t, op=c.dist.ReduceOp.SUM)                                   # Exchange 1 (tier decisions need GLOBAL cardinalities, Q11)
            gcounts = counts_t
            nd = max(s.nd, 1)
            hits_t = torch.zeros((nd, depth, 2), dtype=torch.int32, device=c.device); hc_t = torch.zeros(nd, dtype=torch.int32, device=c.device)
            torch.cuda.current_stream().synchronize()      # the zero fills run on torch's stream, the engine writes on its own: order them
            mark(); s.phase2_dev(gcounts, hits_t, hc_t); mark()
            all_hits_t = torch.empty((c.world, nd, depth, 2), dtype=torch.int32, device=c.device)
            all_hc_t = torch.empty((c.world, nd), dtype=torch.int32, device=c.device)
            c.dist.all_gather_into_tensor(all_hits_t, hits_t)                                     # Exchange 2 (RCCL all-gather of top-k over xGMI)
            c.dist.all_gather_into_tensor(all_hc_t, hc_t)
            outs_t = torch.zeros((max(nq, 1) * 2 * depth, 3), dtype=torch.int32, device=c.device)
            torch.cuda.current_stream().synchronize()
            mark(); s.phase3_dev(all_hits_t, all_hc_t, outs_t, max_results, enable_coverage); mark()
            c.dist.all_reduce(outs_t, op=c.dist.ReduceOp.SUM)                                     # disjoint Stage-2 rows
            torch.cuda.current_stream().synchronize()
            mark(); r = s.phase4_dev(outs_t); mark()
            if dbg and c.rank == 0:
                import sys
                print("[infx-shard] ms: allreduce-uc+alloc %.2f phase1 %.2f allreduce-counts(device)+alloc %.2f phase2 %.2f allgather %.2f phase3 %.2f allreduce-rows %.2f phase4 %.2f" % tuple((T[i + 1] - T[i]) * 1e3 for i in range(8)), file=sys.stderr)
            return r
        counts = s.phase1(guc)
        gcounts = c.allreduce_sum_i32(counts) if counts.size else counts                       # Exchange 1
        hits, hc = s.phase2(gcounts)
        all_hits = c.allgather(hits) if hits.size else hits.reshape((c.world,) + hits.shape)      # Exchange 2 (all-gather of top-k)
        all_hc = c.allgather(hc) if hc.size else hc.reshape((c.world,) + hc.shape)
        outs = s.phase3(all_hits, all_hc, max_results, enable_coverage)
        merged = c.allreduce_sum_i32(outs) if outs.size else outs                              # disjoint Stage-2 records
        return s.phase4(merged)

    def last_timings(self):
        return self.last.s.last_timings()


def simulate_shards_dev(sessions: Sequence[ShardSession], arena, offs, max_results=10, depth=500, enable_coverage=True, device="cuda:0"):
    """simulate_shards with the exchange buffers as torch CUDA tensors (the RCCL code path minus the collectives, which are
    replaced by torch.stack / sum on the same device)."""
    import torch
    ucs = [s.phase0(arena, offs, depth) for s in sessions]
    guc = np.sum(np.stack(ucs).astype(np.uint64), axis=0).astype(np.uint32) if ucs[0].size else ucs[0]
    counts = [s.phase1(guc) for s in sessions]
    g = np.sum(np.stack(counts).astype(np.uint64), axis=0).astype(np.uint32)
    nd = max(sessions[0].nd, 1); nq = sessions[0].nq
    hits, hcs = [], []
    for s in sessions:
        h = torch.zeros((nd, depth, 2), dtype=torch.int32, device=device); c = torch.zeros(nd, dtype=torch.int32, device=device)
        torch.cuda.synchronize()                            # zero fills (torch stream) before the engine's writes (its own stream)
        s.phase2_dev(g, h, c); hits.append(h); hcs.append(c)
    all_hits = torch.stack(hits).contiguous(); all_hc = torch.stack(hcs).contiguous()
    outs = []
    for s in sessions:
        o = torch.zeros((max(nq, 1) * 2 * depth, 3), dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        s.phase3_dev(all_hits, all_hc, o, max_results, enable_coverage); outs.append(o)
    merged = torch.stack(outs).sum(dim=0, dtype=torch.int32).contiguous()
    torch.cuda.synchronize()
    return [s.phase4_dev(merged) for s in sessions]


def simulate_shards(sessions: Sequence[ShardSession], arena, offs, max_results=10, depth=500, enable_coverage=True):
    """Single-process lock-step simulation of W shards (e.g. W engines on ONE GPU): same phases, numpy instead of RCCL.
    Used by the GPU parity test to check that the sharded path reproduces the unsharded results."""
    ucs = [s.phase0(arena, offs, depth) for s in sessions]
    guc = np.sum(np.stack(ucs).astype(np.uint64), axis=0).astype(np.uint32) if ucs[0].size else ucs[0]
    counts = [s.phase1(guc) for s in sessions]
    g = np.sum(np.stack(counts).astype(np.uint64), axis=0).astype(np.uint32)
    ph2 = [s.phase2(g) for s in sessions]
    all_hits = np.stack([h for h, _ in ph2]); all_hc = np.stack([c for _, c in ph2])
    outs = [s.phase3(all_hits, all_hc, max_results, enable_coverage) for s in sessions]
    merged = np.sum(np.stack(outs).astype(np.int64), axis=0).astype(np.int32) if outs[0].size else outs[0]
    res = [s.phase4(merged) for s in sessions]
    return res
