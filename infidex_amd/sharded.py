"""Document-sharded search across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests of the plumbing).
Every rank indexes the whole corpus on the host (global df / avgdl / N, exactly like the reference's single index), uploads its
contiguous doc range — whole 65 536-id Roaring containers, so the reference's chunked walk (Bm25Scorer.cs:195-280) never straddles
two shards — and runs each batch as phases of the C++ engine with small collectives in between:

    phase0   text prep, LD1 member lists, k_union_count     -> all-reduce(sum)  |union| of new fuzzy virtual terms   (Exchange 1b)
    phase1   idf/roles + k_accumulate                       -> all-reduce(sum)  class histograms (tier decisions, Q11) (Exchange 1)
    phase2a  k_select with the GLOBAL counts: first-pass top-`depth` + best score left out -> all-gather               (Exchange 2a)
    phase2b  global ambiguity test + this shard's part of the exact replay (chunks, exact scores, validity intervals)
                                                            -> all-gather of the packed candidates                     (Exchange 2c)
    phase2c  the owner of a flagged query (q mod W) replays the reference's heap over shard 0's chunks, shard 1's, ...
                                                            -> all-gather of the final per-rank lists                  (Exchange 2b: the
                                                               north-star collective — RCCL all-gather of per-shard top-k over xGMI)
    [phase2d sequential chain for queries the parallel replay could not certify: rare]
    phase3   merge, Stage-2 prep, k_stage2 on OWNED candidates -> all-reduce(sum) of the disjoint 12-byte records
    phase4   final ordering / truncation (identical on every rank)

The same driver (`_run_batch`) serves real ranks (one local session, torch.distributed collectives) and the single-process
simulation of W shards on one GPU (W local sessions, the "collectives" are numpy / torch ops) that the parity tests use.
"""
import ctypes as C
from typing import List, Sequence

import os
import numpy as np

from .engine import SearchEngine, Session, _p, INFX_NFEAT  # noqa: F401

INFX_NCLASS = 136
CHAIN = 0xFFFFFFFF


def _vp(x):
    """void* of a numpy array or a torch tensor (host or device)."""
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return x.ctypes.data_as(C.c_void_p)


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

    def allreduce_min_i32(self, a: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(a.view(np.int32))).clone().to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return t.cpu().numpy().view(a.dtype).reshape(a.shape)

    def max_i64(self, v: int) -> int:
        t = self.torch.tensor([v], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())


def create_sharded_engine(rank: int, world: int, device: int, **kw) -> SearchEngine:
    eng = SearchEngine.create_default(device=device, **kw)
    eng._check(eng.L.infx_engine_set_shard(eng.h, rank, world))
    return eng


def _job_dir(cache_dir: str, tag: str, create: bool) -> str:
    """Private directory of one job's hand-off under `cache_dir` (mode 0700, owned by this user, not a link): the path is predictable, the directory
    is world-writable, so nothing is written through what somebody else may have put there."""
    d = os.path.join(cache_dir, f"infx_{os.getuid()}_{tag}")
    if create:
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f"{d}: not a private directory of this user - refusing to hand the host index over through it")
    return d


def index_flat_per_node(eng: SearchEngine, barrier, local_rank: int, node_cpus: int, keys, arena, offs, field_weights, tag: str, cache_dir: str = "/dev/shm"):
    """One host-index build per NODE instead of one per rank.  The node's leader (local rank 0) indexes the documents with every core of the node and
    saves the host index into a private directory under `cache_dir`; the node's other ranks wait (`barrier()`: any barrier over the ranks, e.g.
    torch.distributed.barrier), read the arrays back and upload their own shard; the leader removes file and directory after a second barrier.
    `tag` must be the same on the ranks of a node and unique per job (e.g. MASTER_PORT).  Every rank ends up with the same host index as if it had
    called index_flat itself.  A rank that fails still takes part in both barriers and raises afterwards: a leader that cannot index or save leaves
    no file, so its followers fail to open it and raise as well - nobody waits for a peer that has gone."""
    err = None
    path = None
    try:
        if local_rank == 0:
            path = os.path.join(_job_dir(cache_dir, tag, create=True), "host_index.bin")
            eng.set_build_threads(node_cpus)
            eng.index_flat(keys, arena, offs, field_weights)
            eng.save_host_index(path)
    except BaseException as e:              # noqa: BLE001 - re-raised below, after the barriers the peers are waiting in
        err = e
    try:
        barrier()
        if err is None and local_rank != 0:
            try:
                d = _job_dir(cache_dir, tag, create=False)
                eng.index_from_host_cache(os.path.join(d, "host_index.bin"))
            except BaseException as e:      # noqa: BLE001
                err = e
        barrier()
    finally:
        if local_rank == 0:
            for f in ("host_index.bin", "host_index.bin.tmp"):
                try:
                    os.remove(os.path.join(cache_dir, f"infx_{os.getuid()}_{tag}", f))
                except OSError:
                    pass
            try:
                os.rmdir(os.path.join(cache_dir, f"infx_{os.getuid()}_{tag}"))
            except OSError:
                pass
    if err is not None:
        raise err


class ShardSession:
    """Phase-level access to one engine session.  Exchange buffers are numpy arrays (host) or torch CUDA tensors (RCCL works on them in place)."""

    def __init__(self, engine: SearchEngine):
        self.e = engine
        self.L = engine.L
        self.s = Session(engine)
        self.nq = self.nd = 0
        self.depth = 500

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

    def set_filter(self, expr=None, enable_facets=False) -> int:
        """Query.Filter / Query.EnableFacets on this session; returns THIS SHARD's share of Filter.NumberOfDocumentsInFilter (sum over the shards)."""
        return self.s.set_filter(expr, enable_facets)

    def facets(self, i):
        """Facets of query i of the last batch (phase 4 ran the post-filter and counted the facet values of the kept rows; identical on every rank)."""
        return self.e.facets_of(self.s.h, self.nq, i)

    def phase0(self, arena, offs, depth):
        nu = C.c_uint32(0)
        self.nq = len(offs) - 1
        self.depth = depth
        self.e._check(self.L.infx_session_phase0(self.s.h, self.nq, _p(arena, C.c_uint16), _p(offs, C.c_uint64), depth, C.byref(nu)))
        uc = np.zeros(max(nu.value, 1), np.uint32)
        if nu.value:
            self.e._check(self.L.infx_session_union_counts(self.s.h, _p(uc, C.c_uint32)))
        return uc[:nu.value]

    def phase1(self, global_union_counts, counts):
        """counts: (>= nq) x INFX_NCLASS int32, host or device: this shard's class histograms are written into its first nd rows."""
        nd = C.c_uint32(0)
        guc = np.ascontiguousarray(global_union_counts, np.uint32)
        if guc.size == 0:
            guc = np.zeros(1, np.uint32)
        self.e._check(self.L.infx_session_phase1x(self.s.h, _p(guc, C.c_uint32), _vp(counts), C.byref(nd)))
        self.nd = nd.value

    def phase2a(self, global_counts, hits, hc, nxt):
        self.e._check(self.L.infx_session_phase2a(self.s.h, _vp(global_counts), _vp(hits), _vp(hc), _vp(nxt)))

    def phase2b(self, W, all_hits, all_hc, all_next) -> int:
        nb = C.c_uint64(0)
        self.e._check(self.L.infx_session_phase2b(self.s.h, W, _vp(all_hits), _vp(all_hc), _vp(all_next), C.byref(nb)))
        return int(nb.value)

    def phase2b_blob(self, dst, padded):
        self.e._check(self.L.infx_session_phase2b_blob(self.s.h, _vp(dst), C.c_uint64(padded)))

    def phase2c(self, W, all_blobs, padded, hits, hc):
        self.e._check(self.L.infx_session_phase2c(self.s.h, W, _vp(all_blobs), C.c_uint64(padded), _vp(hits), _vp(hc)))

    def phase2d(self, need, state):
        need = np.ascontiguousarray(need, np.uint32)
        self.e._check(self.L.infx_session_phase2d(self.s.h, _p(need, C.c_uint32), _vp(state)))

    def phase3(self, W, all_hits, all_hc, outs, max_results, enable_coverage=True):
        self.max_results = max_results
        self.e._check(self.L.infx_session_phase3x(self.s.h, W, _vp(all_hits), _vp(all_hc), max_results, int(enable_coverage), _vp(outs)))

    def phase4(self, merged):
        nq, mr = self.nq, self.max_results
        keys = np.full((nq, mr), -1, np.int64); scores = np.zeros((nq, mr), np.float32)
        ties = np.zeros((nq, mr), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
        self.e._check(self.L.infx_session_phase4(self.s.h, _vp(merged), _p(keys, C.c_int64), _p(scores, C.c_float), _p(ties, C.c_uint8),
                                                 _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
        return keys, scores, ties, counts, flags


# ---- exchanges: what differs between real ranks and the in-process simulation ---------------------------------------------------------
class _HostBufs:
    """numpy exchange buffers (gloo, or the host-buffer simulation)."""
    device = None

    def new(self, shape, dtype):
        return np.zeros(shape, dtype)

    def host(self, x):
        return x

    def like_host(self, a):
        return np.ascontiguousarray(a)

    def sync(self):
        pass


class _DevBufs:
    """torch CUDA exchange buffers (RCCL, or the device-tensor simulation).  Allocations and fills run on torch's stream, the engine works on
    its own: sync() orders them (the engine calls are synchronous themselves)."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self._dt = {np.dtype(np.int32): torch.int32, np.dtype(np.uint32): torch.int32, np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8}

    def new(self, shape, dtype):
        return self.torch.zeros(shape, dtype=self._dt[np.dtype(dtype)], device=self.device)

    def host(self, x):
        return x.cpu().numpy()

    def like_host(self, a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        return self.torch.from_numpy(a).to(self.device)

    def sync(self):
        self.torch.cuda.synchronize(self.device)


class _LocalX:
    """W shards simulated in one process: the participants' buffers are combined with numpy / torch ops instead of collectives."""

    def __init__(self, W, bufs):
        self.world = W; self.b = bufs; self.ranks = list(range(W))

    def allreduce_sum(self, xs):
        if self.b.device is None or isinstance(xs[0], np.ndarray):
            r = np.sum(np.stack(xs).astype(np.int64), axis=0).astype(xs[0].dtype)
        else:
            r = self.b.torch.stack(xs).sum(dim=0, dtype=xs[0].dtype).contiguous()
        return [r] * len(xs)

    def allgather(self, xs):
        r = np.stack(xs) if self.b.device is None else self.b.torch.stack(xs).contiguous()
        return [r] * len(xs)

    def max_int(self, vs):
        return max(vs)

    def chain(self, sessions, need, state):
        for s in sessions:                               # shard 0, 1, ...: one heap continued from shard to shard
            s.phase2d(need, state)
        return state


class _DistX:
    """One real rank: torch.distributed collectives (RCCL on device tensors, gloo on numpy)."""

    def __init__(self, comm: TorchComm, bufs):
        self.c = comm; self.world = comm.world; self.b = bufs; self.ranks = [comm.rank]

    def allreduce_sum(self, xs):
        x = xs[0]
        if self.b.device is None or isinstance(x, np.ndarray):
            return [self.c.allreduce_sum_i32(x) if x.size else x]
        self.c.dist.all_reduce(x, op=self.c.dist.ReduceOp.SUM)         # in place on the tensor the kernels wrote
        return [x]

    def allgather(self, xs):
        x = xs[0]
        if self.b.device is None:
            return [self.c.allgather(x)]
        out = self.b.torch.empty((self.world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        self.c.dist.all_gather_into_tensor(out, x.contiguous())
        return [out]

    def max_int(self, vs):
        return self.c.max_i64(int(vs[0]))

    def chain(self, sessions, need, state):
        for r in range(self.world):                      # W sequential steps: rank r continues the heap rank r-1 left
            if r == self.c.rank:
                sessions[0].phase2d(need, state)
            state = self.c.allgather(state)[r].copy()
        return state


def _run_batch(sessions: Sequence[ShardSession], X, ucs, max_results, depth, enable_coverage):
    """All phases after phase 0 for the local participants `sessions` (X.ranks) of a world of X.world shards."""
    B = X.b; W = X.world; s0 = sessions[0]; nq = s0.nq
    gucs = X.allreduce_sum([np.ascontiguousarray(u, np.uint32) for u in ucs]) if ucs[0].size else ucs          # Exchange 1b: df of new fuzzy unions (host: idf is computed there)
    counts = [B.new((max(nq, 1), INFX_NCLASS), np.int32) for _ in sessions]
    B.sync()
    for s, g, c in zip(sessions, gucs, counts):
        s.phase1(g, c)
    gcounts = X.allreduce_sum(counts)                                                                          # Exchange 1: tier decisions need GLOBAL cardinalities (Q11)
    nd = max(s0.nd, 1)
    hits = [B.new((nd, depth, 2), np.int32) for _ in sessions]; hcs = [B.new((nd,), np.int32) for _ in sessions]; nxt = [B.new((nd,), np.float32) for _ in sessions]
    B.sync()
    for s, g, h, c, n in zip(sessions, gcounts, hits, hcs, nxt):
        s.phase2a(g, h, c, n)
    ah, ac, an = X.allgather(hits), X.allgather(hcs), X.allgather(nxt)                                         # Exchange 2a: first-pass lists + best score left out
    B.sync()
    sizes = [s.phase2b(W, h, c, n) for s, h, c, n in zip(sessions, ah, ac, an)]
    pad = (X.max_int(sizes) + 15) & ~15
    blobs = [B.new((pad,), np.uint8) for _ in sessions]
    B.sync()
    for s, b in zip(sessions, blobs):
        s.phase2b_blob(b, pad)
    ab = X.allgather(blobs)                                                                                    # Exchange 2c: candidates the heap could still take
    B.sync()
    for s, b, h, c in zip(sessions, ab, hits, hcs):
        s.phase2c(W, b, pad, h, c)                                                                             # hits / hcs now hold this rank's FINAL contribution
    fh, fc = X.allgather(hits), X.allgather(hcs)                                                               # Exchange 2b: the all-gather of per-shard top-k
    B.sync()
    fc_h = B.host(fc[0]).view(np.uint32)
    need = (fc_h == CHAIN).any(axis=0)
    if need.any():                                                                                             # rare: the literal sequential replay, shard after shard
        state = X.chain(sessions, need.astype(np.uint32), np.zeros((nd, 2 + 2 * depth), np.uint32))
        fh_h = B.host(fh[0]).copy(); fc_h = fc_h.copy()
        for q in np.nonzero(need)[0]:
            n = int(state[q, 0])
            fh_h[:, q] = 0; fc_h[:, q] = 0
            fh_h[0, q, :n, 0] = state[q, 2:2 + n].view(np.int32); fh_h[0, q, :n, 1] = state[q, 2 + depth:2 + depth + n].view(np.int32); fc_h[0, q] = n
        fh = [B.like_host(fh_h)] * len(sessions); fc = [B.like_host(fc_h)] * len(sessions)
        B.sync()
    outs = [B.new((max(nq, 1) * 2 * depth, 3), np.int32) for _ in sessions]
    B.sync()
    for s, h, c, o in zip(sessions, fh, fc, outs):
        s.phase3(W, h, c, o, max_results, enable_coverage)
    merged = X.allreduce_sum(outs)                                                                             # disjoint Stage-2 rows
    B.sync()
    return [s.phase4(m) for s, m in zip(sessions, merged)]


_ALLREDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
_ALLGATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class _CComm(C.Structure):      # infx_comm (include/infidex_engine.h)
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("nranks", C.c_int32), ("device_buffers", C.c_int32), ("reserved", C.c_int32),
                ("allreduce_sum_u32", _ALLREDUCE), ("allgather", _ALLGATHER)]


def native_comm(engine, comm: TorchComm, session=None, group=None):
    """The infx_comm the C++ driver of the sharded phases (infx_session_sharded_finish) talks through.
    nccl backend: RCCL INSIDE the library — rank 0 creates a ncclUniqueId, torch.distributed only ships its 128 bytes, and `session` (a ShardSession) joins
    a communicator of its own; every collective of a batch is then issued by the library on the session's HIP stream over HBM buffers.
    Other backends (gloo in the tests): host buffers + callbacks into torch.distributed on `group`.  Collective: same call order on every rank."""
    cc = _CComm()
    if comm.dist.get_backend() == "nccl":
        L = engine.L; torch = comm.torch
        idb = np.zeros(128, np.uint8)
        if comm.rank == 0:
            engine._check(L.infx_engine_rccl_unique_id(_p(idb, C.c_uint8)))
        t = torch.from_numpy(idb).to(comm.device)
        comm.dist.broadcast(t, src=0)
        idb = np.ascontiguousarray(t.cpu().numpy())
        err = None
        try:
            if session is not None:
                engine._check(L.infx_session_comm_rccl(session.s.h, _p(idb, C.c_uint8), C.byref(cc)))
            else:
                engine._check(L.infx_engine_comm_rccl(engine.h, _p(idb, C.c_uint8), C.byref(cc)))
        except Exception as ex:      # noqa: BLE001 — agreed on below: the world falls back together or not at all
            err = ex
        ok = comm.allreduce_min_i32(np.array([0 if err is not None else 1], np.int32))
        if int(ok[0]) == 1:
            return cc, ()
        # RCCL inside the library could not be brought up on some rank (never seen; W > 1 over RCCL has not run anywhere yet): the batch collectives go through
        # torch.distributed on a gloo group with host buffers instead — slower, but every rank answers, and the line says so (collectives_per_rank.transport)
        import sys
        print(f"[infidex] rank {comm.rank}: in-library RCCL unavailable ({err if err is not None else 'a peer failed'}); falling back to gloo host-buffer collectives", file=sys.stderr)
        cc = _CComm()
        group = comm.dist.new_group(backend="gloo")
        comm.rccl_fallback = True
    torch = comm.torch; dist = comm.dist; world = comm.world

    def _ar(ctx, buf, count, stream):
        try:
            if count:
                dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array((C.c_int32 * count).from_address(buf))), op=dist.ReduceOp.SUM, group=group)      # in place; uint32 sums wrap like int32
            return 0
        except Exception:      # never let an exception cross the C frames
            return 3

    def _ag(ctx, send, recv, nbytes, stream):
        try:
            if nbytes:
                s_ = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(send)))
                r_ = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * (nbytes * world)).from_address(recv)))
                dist.all_gather_into_tensor(r_, s_, group=group)
            return 0
        except Exception:
            return 3
    far, fag = _ALLREDUCE(_ar), _ALLGATHER(_ag)
    cc.ctx = C.c_void_p(1); cc.rank = comm.rank; cc.nranks = world; cc.device_buffers = 0
    cc.allreduce_sum_u32 = far; cc.allgather = fag
    return cc, (far, fag)          # keep the callback objects alive as long as the struct is used


class ShardedSearcher:
    """One rank of a document-sharded deployment. All ranks must call search_packed / search_stream with the same batches.
    native=True (default): planning, the phases and every collective between them run inside library calls (infx_session_phase0 +
    infx_session_sharded_finish), `sessions` batches in flight per rank — each pipeline session has its own HIP stream and its own communicator
    (RCCL inside the library; a torch.distributed group per session with other backends), batch i runs on session i mod `sessions` on every rank, so
    the order of the collectives on each communicator is the same everywhere.
    native=False: the same phases driven from Python one batch at a time (_run_batch — also what the in-process shard simulation of the tests uses)."""

    def __init__(self, engine: SearchEngine, comm: TorchComm, partition_planning: bool = True, native=None, sessions: int = 3):
        import os
        self.native = (os.environ.get("INFX_SHARD_NATIVE", "1") != "0") if native is None else bool(native)
        self.comm = comm
        # the plan exchange: rank r plans its 1/W slice of every batch, the ranks all-gather the plans (INFX_PLAN_EXCHANGE=0 or partition_planning=False:
        # every rank plans every query).  With the dictionaries on the device (the default) the blobs carry the token-level plans and the
        # coverage query contexts; with host lookups also the LD1 expansions and the WordMatcher descriptors.
        # Default (INFX_PLAN_EXCHANGE unset): on only where its projected gain is large and its measured cost small — W >= 4 ranks sharing a node whose CPU quota
        # leaves a rank at most four planner threads (per-rank host planning 10 us -> 5.3 + 4.7 / W us per query).  The only wall-clock
        # measurement in the tree, two ranks on one GPU, is slower with it (27.5 k vs 30.0 k queries/s, profiles/r05_bench_two_ranks_10m_plan_exchange_*.json),
        # so small worlds plan every query on every rank.  INFX_PLAN_EXCHANGE=1 / 0 forces it on / off.
        env = os.environ.get("INFX_PLAN_EXCHANGE", "")
        if env in ("0", "1"):
            want = env == "1"
        else:
            cpus = int(engine.L.infx_engine_effective_cpus())      # hardware threads capped by affinity AND the cgroup CPU quota (the boxes' quota is 16 of 256)
            want = comm.world >= 4 and cpus // max(1, comm.world) <= 4
        self.partition_planning = bool(partition_planning and comm.world > 1 and want)
        K = max(1, int(sessions)) if self.native else 1
        self.sessions = [ShardSession(engine) for _ in range(K)]
        self.sess = self.sessions[0]
        self.last = self.sess
        # per session: the batch communicator and (sharded planning) a gloo group for the planner's byte exchange — collective set-up, same order everywhere
        self.ccomms, self._keep, self.plan_groups = [], [], []
        gloo = comm.dist.get_backend() != "nccl"
        for s in self.sessions:
            if self.native:
                cc, keep = native_comm(engine, comm, session=s, group=(comm.dist.new_group(backend="gloo") if gloo and K > 1 else None))
                self.ccomms.append(cc); self._keep.append(keep)
            g = None
            if self.partition_planning:      # (collective: every rank creates the same groups in the same order)
                try:
                    g = comm.dist.new_group(backend="gloo")
                except Exception as ex:      # no gloo transport on this host (the same on every rank of a node): every rank plans every query, as without the exchange
                    import sys
                    print(f"[infidex] plan exchange off: gloo group could not be created ({ex})", file=sys.stderr)
                    self.partition_planning = False
            self.plan_groups.append(g)
        if comm.world > 1 and partition_planning and want:
            # the fallback is a decision of the WORLD, not of a rank: one rank without its groups while the others exchange would mismatch every collective behind it
            # (and hang until INFX_COMM_TIMEOUT_S).  All ranks reach this all-reduce — group creation above is attempted by all or none — and take the minimum.
            import numpy as _np
            ok = _np.array([1 if self.partition_planning else 0], _np.int32)
            ok = comm.allreduce_min_i32(ok)
            if int(ok[0]) == 0 and self.partition_planning:
                import sys
                print("[infidex] plan exchange off: another rank could not create its gloo groups", file=sys.stderr)
            self.partition_planning = bool(int(ok[0]))
        if not self.partition_planning:
            self.plan_groups = [None] * len(self.sessions)
        self.plan_group = self.plan_groups[0]
        on_dev = comm.device.type == "cuda" and comm.dist.get_backend() == "nccl"
        self.X = _DistX(comm, _DevBufs(comm.device) if on_dev else _HostBufs())
        self.in_filter = 0

    def _prefetch(self, k, arena, offs, depth):
        """Plan exchange: this rank plans its 1/W slice of the batch only (text preparation, term lookups, coverage query contexts; with host
        lookups also the LD1 expansions and WordMatcher descriptors); the ranks all-gather the results and import each other's before phase 0.
        Results are unchanged: every rank holds the whole host index, and what is exchanged is a pure function of (index, query text)."""
        import time
        c = self.comm; s = self.sessions[k]
        s.plan_exchange_ms = (0.0, 0.0, 0.0)
        if c.world <= 1 or not self.partition_planning:
            return
        nq = len(offs) - 1
        begin, end = nq * c.rank // c.world, nq * (c.rank + 1) // c.world
        t0 = time.time()
        mine = s.prefetch_collect(arena, offs, begin, end, depth)
        t1 = time.time()
        blobs = c.allgather_bytes(mine, group=self.plan_groups[k])
        t2 = time.time()
        for r, b in enumerate(blobs):
            if r != c.rank and b.size:
                s.prefetch_import(b)
        s.plan_exchange_ms = (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (time.time() - t2))      # planning the own slice, all-gather (incl. waiting for the peers), import

    def _one(self, k, arena, offs, max_results, depth, enable_coverage):
        s = self.sessions[k]
        self._prefetch(k, arena, offs, depth)
        uc = s.phase0(arena, offs, depth)
        self.last = s
        if not self.native:
            return _run_batch([s], self.X, [uc], max_results, depth, enable_coverage)[0]
        nq, mr = s.nq, max_results
        s.max_results = mr
        keys = np.full((nq, mr), -1, np.int64); scores = np.zeros((nq, mr), np.float32)
        ties = np.zeros((nq, mr), np.uint8); counts = np.zeros(nq, np.uint32); flags = np.zeros(nq, np.uint32)
        s.e._check(s.L.infx_session_sharded_finish(s.s.h, C.byref(self.ccomms[k]), mr, int(enable_coverage), _p(keys, C.c_int64), _p(scores, C.c_float),
                                                   _p(ties, C.c_uint8), _p(counts, C.c_uint32), _p(flags, C.c_uint32)))
        return keys, scores, ties, counts, flags

    def _ring(self, n):
        """Declares the sessions whose collectives the coming batches issue (infx_engine_coll_ring): ring order = session order, identical on every rank."""
        if not self.native:
            return
        hs = (C.c_void_p * max(1, n))(*[self.sessions[k].s.h for k in range(n)])
        self.sessions[0].e._check(self.sessions[0].L.infx_engine_coll_ring(self.sessions[0].e.h, n, hs))

    def plan_exchange_stats(self, k=None):
        """(queries of session k's last phase 0 planned from the exchange, of them imported from peers); k = None: the last session used."""
        s = self.last if k is None else self.sessions[k]
        o = np.zeros(2, np.uint32); s.e._check(s.L.infx_session_plan_exchange_stats(s.s.h, _p(o, C.c_uint32)))
        return int(o[0]), int(o[1])

    def coll_stats(self):
        """Collectives this rank issued through the native driver so far: calls and payload bytes, summed over the pipeline sessions."""
        tot = np.zeros(4, np.uint64)
        for s in self.sessions:
            o = np.zeros(4, np.uint64); s.L.infx_session_coll_stats(s.s.h, _p(o, C.c_uint64)); tot += o
        return dict(allreduce_calls=int(tot[0]), allgather_calls=int(tot[1]), allreduce_bytes=int(tot[2]), allgather_bytes=int(tot[3]))

    def search_packed(self, arena, offs, max_results=10, depth=500, enable_coverage=True):
        self._ring(1)
        return self._one(0, arena, offs, max_results, depth, enable_coverage)

    def search_stream(self, batches, max_results=10, depth=500, enable_coverage=True, stamps=None, timings=None):
        """Stream of batches [(arena, offs), ...] with len(self.sessions) batches in flight: worker thread k takes batches k, k + K, ... through
        session k (planning, kernels and collectives of one batch overlap the others'; the library calls release the GIL).  Yields the results in
        batch order; stamps (optional list) receives (t_start, t_done) per batch, timings the session's last_timings() per batch."""
        import threading
        import time
        K = len(self.sessions); n = len(batches)
        done = [threading.Event() for _ in range(n)]; out = [None] * n; err = []
        st = [None] * n; tm = [None] * n
        self._ring(min(K, n))      # the sessions that take part take turns with their collectives (same order on every rank)

        def worker(k):
            try:
                for i in range(k, n, K):
                    t0 = time.time()
                    out[i] = self._one(k, batches[i][0], batches[i][1], max_results, depth, enable_coverage)
                    st[i] = (t0, time.time()); tm[i] = self.sessions[k].s.last_timings()
                    px = getattr(self.sessions[k], "plan_exchange_ms", (0.0, 0.0, 0.0))
                    tm[i]["plan_exchange_own_slice_ms"], tm[i]["plan_exchange_allgather_ms"], tm[i]["plan_exchange_import_ms"] = px
                    done[i].set()
            except Exception as ex:      # surfaced by the consumer
                err.append(ex)
                for e_ in done:
                    e_.set()
            finally:
                if self.native:
                    self.sessions[k].L.infx_session_coll_retire(self.sessions[k].s.h)      # this session's collectives of the stream are all issued

        ths = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(min(K, n))]
        for t in ths:
            t.start()
        for i in range(n):
            done[i].wait()
            if err:
                raise err[0]
            if stamps is not None:
                stamps.append(st[i])
            if timings is not None:
                timings.append(tm[i])
            yield out[i]
        for t in ths:
            t.join()

    def set_filter(self, expr=None, enable_facets=False) -> int:
        """Query.Filter (Infiscript text, None = no filter) and Query.EnableFacets for the following searches (ResultProcessor.ApplyFilter on the merged
        rows + FacetBuilder, SearchEngine.cs:298-316).  Collective: same call on every rank.  Returns Filter.NumberOfDocumentsInFilter — every rank
        counts its own documents on the device, the counts are summed."""
        mine = [s.set_filter(expr, enable_facets) for s in self.sessions][0]
        self.in_filter = int(self.comm.allreduce_sum_i32(np.asarray([mine], np.uint32))[0]) if self.comm.world > 1 else int(mine)
        return self.in_filter

    def last_facets(self, i):
        return self.last.facets(i)

    def last_timings(self):
        return self.last.s.last_timings()


def simulate_set_filter(sessions: Sequence[ShardSession], expr=None, enable_facets=False) -> int:
    """ShardedSearcher.set_filter for the in-process simulation: installs the filter on every shard's session, sums the per-shard counts."""
    return int(sum(s.set_filter(expr, enable_facets) for s in sessions))


def simulate_shards_dev(sessions: Sequence[ShardSession], arena, offs, max_results=10, depth=500, enable_coverage=True, device="cuda:0"):
    """simulate_shards with the exchange buffers as torch CUDA tensors (the RCCL code path minus the collectives, which are
    replaced by torch.stack / sum on the same device)."""
    ucs = [s.phase0(arena, offs, depth) for s in sessions]
    return _run_batch(sessions, _LocalX(len(sessions), _DevBufs(device)), ucs, max_results, depth, enable_coverage)


def simulate_shards(sessions: Sequence[ShardSession], arena, offs, max_results=10, depth=500, enable_coverage=True):
    """Single-process lock-step simulation of W shards (e.g. W engines on ONE GPU): same phases, numpy instead of RCCL.
    Used by the GPU parity tests to check the sharded path against the oracle."""
    ucs = [s.phase0(arena, offs, depth) for s in sessions]
    return _run_batch(sessions, _LocalX(len(sessions), _HostBufs()), ucs, max_results, depth, enable_coverage)
