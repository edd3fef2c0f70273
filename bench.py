#!/usr/bin/env python
"""bench.py — queries/sec of the Infidex query hot path on MI355X (BASELINE.json metric).

Workload (N=1): BASELINE config 4 — 10 M synthetic single-field docs (SURVEY.md §8d generator), streams of 1 000-query
batches (2- and 3-word queries, 30 % fuzzed), CoverageDepth 500, top-k = 20.  One "step" = one 1 000-query batch through
the whole hot path (Stage-1 planning -> k_accumulate/k_select -> Stage-2 preparation -> k_stage2 -> final ordering and
truncation) with the index resident in HBM.

N>1 (launched by torch.distributed.run, one rank per GPU): see DESIGN.md "Multi-GPU".

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (k_accumulate, HBM-bound) and
`cpu_baseline` (the oracle = restated reference algorithm in C++, NOT the .NET binary, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Batches in flight run on separate HIP streams: give the runtime enough hardware queues for them (default 4) BEFORE HIP initialises, so the
# light, latency-bound kernels of one batch (top-k selection, exact replay) overlap the streaming kernels of another (measured: +10 % q/s).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0     # the measured copy rate of the same guide: SURVEY 8(d) asks for the fraction of it next to the fraction of the spec


PMC_FILE = "r06_pmc.json"


def kernel_sha16():
    """sha256 (16 hex digits) of what decides the traffic of the Stage-1 accumulation (k_accumulate_sparse + k_accumulate): the kernel sources and their launch code (grid, XCD-aware block map, split point, LDS sizes)."""
    import hashlib
    csrc = os.path.join(ROOT, "infidex_amd", "csrc")
    h = hashlib.sha256(open(os.path.join(csrc, "stage1.hip.inc"), "rb").read())
    h.update(open(os.path.join(csrc, "stage1_sparse.hip.inc"), "rb").read())
    src = open(os.path.join(csrc, "infidex_hip.hip")).read()
    a = src.index("template <int R> static void launch_acc("); b = src.index("// Longest-queries-first order for k_select", a)
    h.update(src[a:b].encode())
    return h.hexdigest()[:16]


def roofline_by_kernel(roof):
    """Every kernel that takes >= 5 % of the GPU time of a batch, priced like k_accumulate: algorithmic bytes of the launch (DESIGN.md section 4 defines them per
    kernel) / duration (HIP events on the launch stream, one batch in flight) against the HBM peak, with what actually bounds it."""
    def m(key):
        vals = [t[key] for t in roof if key in t]
        return float(np.mean(vals)) if vals else 0.0
    rows = m("stage1_candidates"); s2rows = m("stage2_candidates"); s2bytes = m("stage2_text_bytes"); repl = m("exact_replays")
    nq = m("queries") or 1000.0
    spec = [
        # kernel, duration key, algorithmic bytes, bound, what the bytes are
        ("k_select", "k_select_ms", rows * 9.0, "latency", "arena rows x (class 1 B + score 4 B + doc id 4 B), each read once (the radix select reads class + score once per pass: 2-3 passes); incl. k_select_order and the multi-workgroup sweeps of the largest queries (k_selg_*)"),
        ("k_ex_scan", "k_ex_scan_ms", m("replay_rows") * 12.0, "serial-latency", "k_ex_walk x2 + k_ex_prefix + k_ex_theta: rows of the flagged queries x (class 1 B + score 4 B + directory / candidate-list 7 B); the time is k_ex_theta's ~1 400 serial sorted insertions per query on one wave"),
        ("k_ex_chunk", "k_ex_chunk_ms", m("replay_rows") * (4.0 + 8.0 + 4.0 + 4.0), "latency", "k_ex_chunk<1|4|16>: candidate rows of the flagged queries x (list entry 4 B + hit mask 8 B + tf exceptions 4 B + doc length 4 B); three dependent loads + the term loop per chunk task"),
        ("k_ex_heap", "k_ex_heap_ms", m("replay_rows") * 8.0, "serial-latency", "emitted candidates x (doc 4 B + score 4 B); time = ~2 k dependent 4-ary heap operations per query on one wave (register-resident heap: ~1 200 cycles each)"),
        ("k_prep2", "k_prep2_ms", nq * 500 * 8.0 + s2rows * 16.0, "latency", "Stage-1 rows in (8 B) + candidate rows out (16 B) + WordMatcher list probes (binary searches)"),
        ("k_stage2", "k_stage2_ms", s2bytes + s2rows * (16.0 + 12.0), "issue", "UTF-16 text of every scored row + candidate row in (16 B) + result out (12 B); integer string code, one lane per row"),
    ]
    out = []
    tot = sum(m(k) for k in ("k_accumulate_ms", "k_select_ms", "k_replay_ms", "k_prep2_ms", "k_stage2_ms", "k_finalize_ms")) or 1.0
    for name, key, bytes_, bound, what in spec:
        ms = m(key)
        if ms <= 0:
            continue
        ach = bytes_ / (ms * 1e-3) / 1e9
        out.append({"kernel": name, "avg_launch_ms": ms, "share_of_gpu_time": ms / tot, "algorithmic_bytes_per_launch": bytes_, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "bound": bound, "bytes_are": what})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)      # >= 3 s of timed GPU work at ~14 ms per 1000-query batch
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--docs", type=int, default=0, help="0 = the full size of the config (config 4: 10 M)")
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--range-docs", type=int, default=0)
    ap.add_argument("--build-threads", type=int, default=0)
    # measured at 10 M docs (queries/s, p50 / p95 batch latency ms): 4: 53.3 k, 48 / 106; 5: 63.9 k, 60 / 119; 6: 65.9 k, 85 / 126; 7: 66.2 k, 76 / 177; 8: 59.1 k, 86 / 375 —
    # the replay kernels of a batch run one workgroup per flagged query and leave most CUs to other batches' full-width kernels
    # round 4 (workspaces never reallocated mid-stream, turnstile on the wide phase; 100 steps, queries/s and p50 / p95 batch latency ms): 3: 67.7 k, 42 / 56;
    # 4: 72.7 k, 52 / 75; 5: 74.6 k, 63 / 88; 6: 73.7-77.1 k, 75 / 111; 8: 74.0 k, 97 / 169 — five keeps the rate of six at 15 % less latency
    ap.add_argument("--sessions", type=int, default=4, help="batches in flight (host threads, one engine session each)")
    ap.add_argument("--shard-sessions", type=int, default=4, help="sharded runs: batches in flight per rank (pipeline sessions, each with its own stream and communicator); W = 1 forced-sharded, 40 steps: 3: 81.5 k, 4: 87.8 k, 5: 92.3 k, 6: 85.8 k queries/s")
    ap.add_argument("--distinct-batches", type=int, default=0, help="distinct synthetic query batches the steps cycle through; 0 (default) = warmup + steps, "
                    "i.e. no batch — and so no misspelt word beyond what the Zipf stream itself repeats — occurs twice: planning is measured cold")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent full-index replicas instead of document shards")
    ap.add_argument("--long-steps", type=int, default=200, help="single-GPU runs: a second, longer stream of this many fresh batches after the timed region, reported as value_long "
                                                                "(the driver's K steps time a quarter of a second, less than the box-to-box spread); 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries in the CPU-baseline sample (0 = auto, ~10-30 s)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    # --gpus N is the number of ranks.  Started bare (no WORLD_SIZE: `python bench.py --gpus N`), the script launches itself under torch.distributed.run with
    # one rank per GPU — the form the driver uses for N > 1 — and relays the ranks' output; started BY torch.distributed.run it checks that the launcher's
    # world size is the one asked for, so a line can never report another N than the one it ran on.
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import subprocess
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", port,
               os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and os.environ.get("INFX_FORCE_SHARDED") != "1":
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} ranks; start `python bench.py --gpus N` "
                 f"(it launches the ranks itself) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")

    # stdout carries ONE line, the result: everything else a library prints there (RCCL's start-up banner, C stdio) goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    node_local_rank = local_rank          # rank within the node (the device index below may wrap when ranks share a GPU in tests)
    dist = None
    force_sharded = os.environ.get("INFX_FORCE_SHARDED") == "1"      # exercise the sharded / RCCL code path with a single rank
    if world > 1 or force_sharded:
        import torch
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("INFX_DIST_BACKEND", "nccl")     # "gloo" lets two ranks share one GPU when testing the sharded flow
        if os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"]) == os.environ["WORLD_SIZE"]:
            # all ranks on this node: the gloo groups of the run (plan exchange, the per-session groups of a gloo run) talk over loopback — gloo otherwise looks the
            # container's hostname up to pick an interface, and that name need not resolve
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        if torch.cuda.is_available() or backend == "nccl":       # (the launch-only CPU test runs over gloo without a device)
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
    if os.environ.get("INFX_BENCH_LAUNCH_ONLY") == "1":      # tests/test_bench_launch.py: the launch path alone (ranks, process group, one collective), no GPU work
        n = 1
        if dist is not None:
            import torch
            t = torch.ones(1, device="cuda" if dist.get_backend() == "nccl" else "cpu"); dist.all_reduce(t); n = int(t.item())
        if rank == 0:
            os.write(result_fd, (json.dumps({"launch_only": True, "n_gpus": world, "ranks_in_all_reduce": n, "gpus_arg": args.gpus}) + "\n").encode())
        if dist is not None:
            dist.destroy_process_group()
        return
    from tools.synth import Synth, CONFIGS
    from infidex_amd import SearchEngine, Session, build as _build
    _build.build()

    def quota_cpus():      # the same rule as infx_engine_effective_cpus (affinity mask and cgroup CPU quota), before the library is loaded
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                n = min(n, max(1, -(-int(q) // int(per))))
        except Exception:
            try:
                q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0 and per > 0:
                    n = min(n, max(1, -(-q // per)))
            except Exception:
                pass
        return max(1, n)
    if world > 1 and "INFX_THREADS" not in os.environ:
        # one process per GPU shares the host: size every process's planner pool to its share of the CPUs
        os.environ["INFX_THREADS"] = str(max(1, quota_cpus() // world))
    from infidex_amd.engine import load_library
    ncpu = int(load_library().infx_engine_effective_cpus())     # hardware threads capped by affinity and the cgroup CPU quota
    bthreads = args.build_threads or max(1, min(64, ncpu if "INFX_THREADS" in os.environ else ncpu // max(1, world)))
    full = CONFIGS[args.config]["docs"]
    if not args.docs:
        args.docs = full
    syn = Synth(args.config, docs=(None if args.docs == full else args.docs), threads=bthreads)
    k = syn.cfg["k"]
    # one host-index build per node (N > 1): the node's leader generates and indexes the corpus with every core and hands the host index to the other
    # ranks through /dev/shm (infidex_amd/sharded.py: index_flat_per_node); INFX_SHARED_HOST_INDEX=0: every rank builds its own, as before
    share_build = world > 1 and os.environ.get("INFX_SHARED_HOST_INDEX", "1") != "0"
    leader = (not share_build) or node_local_rank == 0
    t0 = time.time()
    if share_build and leader:
        syn = Synth(args.config, docs=(None if args.docs == full else args.docs), threads=args.build_threads or max(1, min(64, quota_cpus())))
    arena, offs = syn.docs() if leader else (None, None)
    t_gen = time.time() - t0

    # ---- CPU baseline (rank 0, N=1 only): the oracle's index is built AFTER the timed GPU region (it would compete for the CPU quota) ----
    want_cpu = (not args.no_cpu_baseline) and rank == 0 and world == 1
    orc_box = {}

    # ---- product: index + upload ---------------------------------------------------------------------------------------
    t0 = time.time()
    sharded = (world > 1 or force_sharded) and not args.replicas
    if sharded:
        # north-star layout: the 10 M-doc index is document-sharded over the GPUs, every rank answers the SAME query stream,
        # per-shard top-k merged by an RCCL all-gather (infidex_amd/sharded.py)
        from infidex_amd.sharded import create_sharded_engine, ShardedSearcher, TorchComm
        eng = create_sharded_engine(rank, world, local_rank, threads=bthreads, range_docs=args.range_docs)
    else:
        eng = SearchEngine.create_default(device=local_rank, threads=bthreads, range_docs=args.range_docs)
    if share_build:
        from infidex_amd.sharded import index_flat_per_node
        index_flat_per_node(eng, dist.barrier, node_local_rank, max(1, min(64, quota_cpus())), None, arena, offs, syn.field_weights, tag=os.environ.get("MASTER_PORT", "0"))
    else:
        eng.index_flat(None, arena, offs, syn.field_weights)
    flt = syn.cfg.get("filter")                       # config 5: Query.Filter + Query.EnableFacets on device-resident columns
    cols5 = None
    if flt:
        from tools.synth import config5_columns
        cols5 = config5_columns(args.docs)
        eng.set_column("year", cols5[0], facetable=True); eng.set_column("rating", cols5[1], facetable=False); eng.set_column("genre", cols5[2], facetable=True)
    t_index = time.time() - t0

    nsteps = args.warmup + args.steps
    ndist = nsteps if args.distinct_batches <= 0 else max(1, min(nsteps, args.distinct_batches))
    qa, qo = syn.queries(ndist * args.batch, qseed=1000 + (0 if sharded else rank))
    dbatches = []
    for s in range(ndist):
        lo, hi = s * args.batch, (s + 1) * args.batch
        o2 = (qo[lo:hi + 1] - qo[lo]).astype(np.uint64)
        dbatches.append((np.ascontiguousarray(qa[int(qo[lo]):int(qo[hi])]) if qo[hi] > qo[lo] else np.zeros(1, np.uint16), o2))
    batches = [dbatches[s % ndist] for s in range(nsteps)]
    # set-up batches (another seed, never part of the warm-up or the timed stream): every session runs one before the warm-up so that its device
    # workspaces and pinned staging are allocated outside the measurement — with fewer warm-up steps than sessions some would otherwise grow theirs
    # (hipMalloc / hipHostMalloc) inside the timed region
    sqa, sqo = syn.queries(args.batch, qseed=77000 + (0 if sharded else rank))      # document shards answer the SAME batch on every rank (round 5: the per-rank seed made the ranks' collectives differ in size — found by the first two-rank run of this script)
    setup_batch = (np.ascontiguousarray(sqa[:int(sqo[-1])]) if sqo[-1] > 0 else np.zeros(1, np.uint16), sqo.astype(np.uint64))

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # A stream of batches: `--sessions` host threads, each with its own engine session (HIP stream + scratch), pull batches from
    # a shared cursor, so the host-side preparation of one batch overlaps the GPU stages of another.  Every batch still runs the
    # complete hot path; results do not depend on the interleaving (tests/test_gpu_parity.py::test_batching_is_transparent).
    if sharded:
        nsess = max(1, min(args.shard_sessions, args.steps))
        searcher = ShardedSearcher(eng, TorchComm(dist), sessions=nsess)
        list(searcher.search_stream([setup_batch] * len(searcher.sessions), k, 500))       # every pipeline session allocates its workspaces
        list(searcher.search_stream(batches[:args.warmup], k, 500))
        sync()
        tim, lat, results = [], [], []
        t_start = time.time()
        stamps = []
        # planner thread: phase 0 of the next batch overlaps the collective phases of the current one (sharded.py search_stream)
        def lockstep(bs):
            for a_, o_ in bs:
                t0_ = time.time(); r_ = searcher.search_packed(a_, o_, k, 500); stamps.append((t0_, time.time())); yield r_
        lock = os.environ.get("INFX_SHARD_LOCKSTEP") == "1"
        stream = lockstep(batches[args.warmup:nsteps]) if lock else searcher.search_stream(batches[args.warmup:nsteps], k, 500, stamps=stamps, timings=tim)
        for keys, scores, ties, counts, flags in stream:
            if lock:
                tim.append(searcher.last_timings())
            results.append((keys, counts))
        lat = [(b - a) * 1000.0 for a, b in stamps]
        first_keys = (results[0][0].copy(), results[0][1].copy())
        nsess = len(searcher.sessions)
        sessions = None
    else:
        nsess = max(1, min(args.sessions, args.steps))
        sessions = [Session(eng) for _ in range(nsess)]
        in_filter = None
        if flt:
            tf0 = time.time(); in_filter = [se.set_filter(flt, True) for se in sessions][0]; t_filter_first_use = time.time() - tf0
        for se in sessions:
            se.search_packed(setup_batch[0], setup_batch[1], k, 500)
        for s in range(args.warmup):
            sessions[s % nsess].search_packed(batches[s][0], batches[s][1], k, 500)
        sync()
        tim = [None] * nsteps
        lat = [0.0] * nsteps
        results = [None] * nsteps
        cursor = {"next": args.warmup}
        lock = threading.Lock()
        errors = []

        def worker(sess):
            try:
                while True:
                    with lock:
                        s = cursor["next"]
                        if s >= nsteps:
                            return
                        cursor["next"] = s + 1
                    ts = time.time()
                    keys, scores, ties, counts, flags = sess.search_packed(batches[s][0], batches[s][1], k, 500)
                    lat[s] = (time.time() - ts) * 1000.0
                    stampsT[s] = (sessions.index(sess), ts, time.time())
                    tim[s] = sess.last_timings(kernels=(s % 4 == 0))      # kernel durations on every fourth batch: resolving them is HIP API traffic
                    results[s] = (keys, counts)
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)

        stampsT = [None] * nsteps
        t_start = time.time()
        ths = [threading.Thread(target=worker, args=(se,)) for se in sessions]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise errors[0]
        if os.environ.get("INFX_BENCH_TIMELINE") == "1":       # per batch: session, submit and completion time (ms from the start of the timed region), host plan / device wait
            for i in range(args.warmup, nsteps):
                se_, a_, b_ = stampsT[i]
                print(f"[timeline] batch {i - args.warmup:3d} session {se_} submit {1000 * (a_ - t_start):7.1f} done {1000 * (b_ - t_start):7.1f} plan {tim[i]['plan_ms']:5.1f} wait {tim[i]['stage2_ms']:5.1f}", file=sys.stderr)
        tim = tim[args.warmup:]
        lat = lat[args.warmup:]
        first_keys = (results[args.warmup][0].copy(), results[args.warmup][1].copy())
    sync()
    elapsed = time.time() - t_start
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # sharded: all ranks cooperate on ONE stream (strong scaling of the 10 M corpus); replicas: every rank answers its own stream
    total_queries = args.steps * args.batch * (1 if sharded else world)
    qps = total_queries / elapsed

    # long leg (after the timed region, same process, same sessions): `--long-steps` FRESH batches through the same worker loop — the steady-state rate once the
    # fill and drain of the session pipeline no longer weigh (value_long); not the headline, which stays the driver's K steps
    value_long = None
    if not sharded and world == 1 and args.long_steps > 0:
        lqa, lqo = syn.queries(args.long_steps * args.batch, qseed=5000)
        lb = []
        for s_ in range(args.long_steps):
            lo, hi = s_ * args.batch, (s_ + 1) * args.batch
            lb.append((np.ascontiguousarray(lqa[int(lqo[lo]):int(lqo[hi])]) if lqo[hi] > lqo[lo] else np.zeros(1, np.uint16), (lqo[lo:hi + 1] - lqo[lo]).astype(np.uint64)))
        cur2 = {"next": 0}; lock2 = threading.Lock(); err2 = []; lat2 = [0.0] * len(lb)

        def worker2(sess):
            try:
                while True:
                    with lock2:
                        s_ = cur2["next"]
                        if s_ >= len(lb):
                            return
                        cur2["next"] = s_ + 1
                    t_ = time.time(); sess.search_packed(lb[s_][0], lb[s_][1], k, 500); lat2[s_] = (time.time() - t_) * 1000.0
            except Exception as ex:  # noqa: BLE001
                err2.append(ex)
        tl0 = time.time()
        th2 = [threading.Thread(target=worker2, args=(se,)) for se in sessions]
        for t in th2:
            t.start()
        for t in th2:
            t.join()
        if err2:
            raise err2[0]
        tl = time.time() - tl0
        value_long = {"value": args.long_steps * args.batch / tl, "unit": "queries/s", "steps": args.long_steps, "ms_per_step": tl / args.long_steps * 1000.0,
                      "p50_batch_latency_ms": float(np.median(lat2)), "p95_batch_latency_ms": float(np.percentile(lat2, 95)),
                      "note": "same process and sessions, fresh batches, run after the timed region; the headline `value` is the driver's K-step form"}

    # roofline leg: the same batches once more on ONE session (no concurrent kernels), HIP events on the launch stream
    roof = []
    if sharded:
        roof = tim
    else:
        for s in range(args.warmup, min(nsteps, args.warmup + 3)):
            sessions[0].search_packed(batches[s][0], batches[s][1], k, 500)
            roof.append(sessions[0].last_timings())
    # interactive use: one query per call (the reference's Search(Query)), on one session
    single_ms = None
    if not sharded:
        from infidex_amd.engine import pack_texts
        texts1 = Synth.texts(batches[args.warmup][0], batches[args.warmup][1])[:40]
        ls = []
        for q1 in texts1:
            a1, o1 = pack_texts([q1])
            t1 = time.time(); sessions[0].search_packed(a1, o1, k, 500); ls.append((time.time() - t1) * 1000.0)
        single_ms = float(np.median(ls[8:]))
    STAGE_KEYS = ("plan_ms", "stage1_ms", "prep2_ms", "stage2_ms", "post_ms", "k_accumulate_ms", "k_select_ms", "k_replay_ms", "k_prep2_ms", "k_stage2_ms", "k_finalize_ms")
    for t in list(tim) + list(roof):                   # the library times rules + select + replay as one span: report select and replay apart
        if "k_select_only" not in t and "k_select_ms" in t:
            t["k_select_only"] = True; t["k_select_ms"] = max(0.0, t["k_select_ms"] - t.get("k_replay_ms", 0.0))
        t.setdefault("queries", args.batch)
        if "exact_replays" in t and "stage1_candidates" in t:      # candidate rows of the queries the replay handles, estimated from the flagged share
            t.setdefault("replay_rows", t["stage1_candidates"] * t["exact_replays"] / max(1, args.batch))
    stage_me = {kk: float(np.mean([t[kk] for t in tim if kk in t])) for kk in STAGE_KEYS if any(kk in t for t in tim)}
    for kk in ("plan_tokens_ms", "plan_ld1_device_ms", "plan_union_device_ms", "plan_finish_ms",      # where plan_ms went (fused sessions)
               "plan_exchange_own_slice_ms", "plan_exchange_allgather_ms", "plan_exchange_import_ms"):      # sharded: the plan exchange in front of phase 0 (not part of plan_ms)
        if tim and all(kk in t for t in tim):
            stage_me[kk] = float(np.mean([t[kk] for t in tim]))
    stage_ranks = None
    if dist is not None and world > 1:          # per-phase milliseconds of EVERY rank (host phases are replicated, device phases shrink with W)
        stage_ranks = [None] * world
        dist.all_gather_object(stage_ranks, stage_me)
    acc_ms = float(np.mean([t["k_accumulate_ms"] for t in roof]))
    alg = float(np.mean([t["alg_bytes"] for t in roof]))
    streamed = float(np.mean([t["streamed_bytes"] for t in roof]))
    achieved = alg / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
    qw = str(syn.cfg["qwords"][0]) if syn.cfg["qwords"][0] == syn.cfg["qwords"][1] else f"{syn.cfg['qwords'][0]}-{syn.cfg['qwords'][1]}"
    lk = eng.lookup_stats() if hasattr(eng, "lookup_stats") else None
    out = {
        "metric": ("queries/sec, 10M-doc corpus, top-k=20 (whole hot path, index resident in HBM)" if args.config == 4 and args.docs == full else
                   f"queries/sec, config {args.config}, {args.docs} docs, top-k={k} (whole hot path, index resident in HBM)"),
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1000.0, "higher_is_better": True, "scaling": ("strong" if sharded else "weak"),
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE config {args.config}: {syn.cfg['docs']} docs, vocab {syn.cfg['vocab']}, {len(syn.cfg['fields'])} field(s), {args.batch}-query batches, "
                               f"{qw}-word queries {int(syn.cfg['fuzz'] * 100)}% fuzzed, depth 500, top-{k}" + (f", filter {flt!r} + facets" if flt else ""),
                   "docs": syn.cfg["docs"], "batch": args.batch, "top_k": k, "coverage_depth": 500,
                   "sessions_in_flight": nsess,
                   "parallelism": "single GPU" if world == 1 else (f"{world} document shards, count all-reduce + RCCL all-gather of per-shard top-500 + owner-scored Stage 2" if sharded
                                                                    else f"{world} independent replicas (one index per GPU, query stream split)")},
        "p50_batch_latency_ms": float(np.median(lat)), "p95_batch_latency_ms": float(np.percentile(lat, 95)),
        "p50_single_query_latency_ms": single_ms,
        "value_long": value_long,
        # host phases of a batch (per session; sessions overlap): planning (text prep, term lookup, LD1 expansion, idf/roles),
        # Stage-1 host part (phase API only), Stage-2 preparation (fused pipeline: WordMatcher descriptors + PrepareQuery),
        # the wait for the device (fused: the whole device pipeline behind one synchronisation), host post-processing
        "stage_ms_per_step": stage_me,
        # "bound": what limits the kernel by the counters of profiles/ (instruction issue), NOT what it is priced against: `peak` stays the HBM roofline the
        # path is bounded by in principle (byte streaming, no MFMA work), so `frac` is the achieved fraction of the HBM roofline by algorithmic bytes
        "roofline": {"kernel": "Stage-1 accumulation = k_accumulate_sparse + k_accumulate (two kernels split the (query, stripe) pairs of a launch by candidate count; one after the other on the "
                               "stream, timed as one span)", "bound": "issue", "priced_against": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS, "traffic": None,
                     "limiter": "not bandwidth: the accumulation is priced against the HBM roofline (byte streaming, no MFMA work) by SURVEY 8(d)'s algorithmic bytes, but the streaming "
                                "kernel (dense stripes: ~3.7 ms with all 4096 wave slots busy since its blocks run heaviest query first) is bound by VALU / SALU issue and latency of its (posting list, "
                                "doc range) visits, and the sparse kernel (stripes of <= 96 candidates: ~2.0 ms) by the lane-loads of its dependent lookups - it does not stream its lists at "
                                "all (profiles/r06_final_10m.md, r06_accumulate_blocks.md, r06_pmc.json)",
                     "algorithmic_bytes_per_launch": alg, "avg_launch_ms": acc_ms,
                     "streamed_bytes_per_launch": streamed, "stage1_candidates_per_launch": float(np.mean([t["stage1_candidates"] for t in roof])),
                     "stage2_rows_per_launch": float(np.mean([t["stage2_candidates"] for t in roof])), "exact_replays_per_launch": float(np.mean([t["exact_replays"] for t in roof])), "streamed_GBps": streamed / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0,
                     "other_kernels_ms": {kk: float(np.mean([t[kk + "_ms"] for t in roof])) for kk in ("k_select", "k_replay", "k_prep2", "k_stage2", "k_finalize")},
                     "replay_flag_reasons_per_launch": {kk: float(np.mean([t["flag_" + kk] for t in roof])) for kk in ("plateau", "band", "unknown")},
                     "note": "achieved = SURVEY 8(d) algorithmic bytes / duration of the two accumulation kernels together (HIP events on the launch stream around both, uncontended launch); the rocprofv3 kernel-trace averages of k_accumulate_sparse and k_accumulate add up to it"},
        "roofline_by_kernel": roofline_by_kernel(roof),
        "planning_lookups": lk,
        "setup_s": {"corpus_gen": t_gen, "index_build_and_upload": t_index, "host_threads": bthreads},
    }
    if stage_ranks is not None:
        out["stage_ms_per_step_per_rank"] = stage_ranks
    if sharded:
        # what every rank issued through the native driver (set-up + warm-up + timed batches): W ranks with equal counts = RCCL saw W ranks in step
        cs = searcher.coll_stats()
        cs["batches"] = len(searcher.sessions) + nsteps
        cs["transport"] = ("gloo host buffers (in-library RCCL could not be brought up: fallback)" if getattr(searcher.comm, "rccl_fallback", False)
                           else ("RCCL inside the library, HBM buffers, on the sessions' streams" if dist is not None and dist.get_backend() == "nccl" else "torch.distributed callbacks (gloo), host buffers"))
        # the plan exchange: queries of this rank's last batch planned from exchanged plans (own slice + peers') / imported from peers (W = 1: nothing to exchange)
        px = searcher.plan_exchange_stats()
        cs["plan_exchange"] = {"on": bool(searcher.partition_planning), "queries_planned_from_exchange": px[0], "of_them_imported_from_peers": px[1]}
        if dist is not None and world > 1:
            allcs = [None] * world
            dist.all_gather_object(allcs, cs)
        else:
            allcs = [cs]
        out["collectives_per_rank"] = allcs
        out["config"]["collective_order"] = "ring over the pipeline sessions (infx_engine_coll_ring): one collective per turn, same order on every rank"
    if flt and not sharded:
        out["config"]["filter"] = flt; out["config"]["facets"] = ["year", "genre"]
        out["filter"] = {"documents_in_filter": in_filter, "first_use_s_incl_compile_and_device_count": t_filter_first_use}
    # HBM traffic of the dominant kernel comes from a separate rocprofv3 --pmc pass (counters cannot be read from inside the process).  The committed
    # measurement is attached only to the workload it was taken on (config 4, full size, 1000-query batches, one GPU) and only if it was taken on THIS
    # build of the kernel: sha256 over csrc/stage1.hip.inc AND the launch code of csrc/infidex_hip.hip (launch_acc: grid, block map, LDS size).
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        ksha = kernel_sha16()
        if args.config == 4 and args.docs == full and args.batch == 1000 and not sharded and world == 1 and pmc.get("kernel_source_sha16") == ksha:
            out["roofline"]["traffic"] = pmc["hbm_read_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "HBM read bytes per launch (FETCH_SIZE x2, " + pmc["source"] + ")"
            out["roofline"]["traffic_frac_of_peak"] = pmc["hbm_read_bytes_per_launch"] / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if acc_ms > 0 else None
        else:
            out["roofline"]["traffic_note"] = f"profiles/{PMC_FILE} was measured on another build of the kernel or another workload (config 4 at full size only): not attached"
    except Exception:
        pass
    if want_cpu:
        from tests import oracle_lib as O
        tb = time.time()
        o = O.OracleEngine.create_default()
        o.add_flat(None, arena, offs, syn.field_weights)
        o.finalize()
        orc_box["build_s"] = time.time() - tb
        cthreads = args.cpu_threads or max(1, min(64, ncpu))
        sample = args.cpu_sample
        texts = Synth.texts(batches[args.warmup][0], batches[args.warmup][1])
        if sample <= 0:
            probe = texts[:8]
            ps, _, _ = o.timed_batch(probe, k, 500, threads=1)
            per_q = ps / len(probe)
            sample = int(max(32, min(len(texts), 20.0 * cthreads / max(per_q, 1e-6))))
        sample = min(sample, len(texts))
        O.stage_times(reset=True)
        secs, okeys, _ = o.timed_batch(texts[:sample], k, 500, threads=cthreads)
        stage_tot = O.stage_times(reset=True)              # thread-milliseconds per stage over the sample
        secs1, _, lat1 = o.timed_batch(texts[:min(sample, 24)], k, 500, threads=1, want_latency=True)
        stage1t = O.stage_times(reset=True)
        # identical top-k DocumentId sets on the sample (parity is asserted in tests/; reported here)
        gk, gc = first_keys
        if flt:        # config 5: the rows are post-filtered; compare against the oracle's filtered search (sequential, smaller sample) incl. facets
            o.set_column("year", cols5[0], facetable=True); o.set_column("rating", cols5[1], facetable=False); o.set_column("genre", cols5[2], facetable=True)
            sample = min(sample, 96)
            want = [o.search_filtered(texts[i], k, 500, filter=flt, enable_facets=True) for i in range(sample)]
            okeys = [np.asarray(w["keys"], np.int64) for w in want]
            got = eng.search_filtered(texts[:sample], k, 500, filter=flt, enable_facets=True)
            out["filter"]["facets_identical"] = f"{sum(1 for g, w in zip(got, want) if (g.facets or {}) == w['facets'])}/{sample}"
            out["filter"]["documents_in_filter_oracle"] = want[0]["in_filter"] if want else None
        differ = [i for i in range(sample) if set(gk[i, :gc[i]].tolist()) != set(x for x in okeys[i].tolist() if x >= 0)]
        same = sample - len(differ)
        from tests.parity_classify import classify
        cls = classify(eng, o, [texts[i] for i in differ], k) if (differ and not flt) else []
        out["cpu_baseline"] = {"value": sample / secs, "unit": "queries/s", "cores": cthreads, "kind": "port",
                               "sample": f"first {sample} queries of the first timed batch on the same {args.docs}-document config-{args.config} index, one in-flight query per thread; "
                                         f"oracle = C++ restatement of the reference algorithm (not the .NET binary)",
                               "single_thread_qps": min(sample, 24) / secs1, "single_thread_p50_ms": float(np.median(lat1)),
                               # where a query's time goes in the port: thread-milliseconds per query and stage (all threads of the sample run / the single-thread run)
                               "stage_ms": {kk: v / sample for kk, v in stage_tot.items()},
                               "stage_ms_single_thread": {kk: v / min(sample, 24) for kk, v in stage1t.items()},
                               "index_build_s": orc_box["build_s"], "identical_topk_sets": f"{same}/{sample}",
                               "parity": {"identical": same, "tie_at_cut_off": sum(1 for c in cls if c["kind"] == "tie-at-cut-off"),
                                          "identical_on_rerun": sum(1 for c in cls if c["kind"] == "identical-on-rerun"),
                                          "other": [c for c in cls if c["kind"] == "other"],
                                          "unpinned": "product and oracle replay the same restatement of .NET's PriorityQueue (4-ary heap) / introsort / logf; no .NET "
                                                      "runtime exists here, so agreement with a real .NET run on exact ties is not checked"}}
        out["speedup_vs_cpu_baseline"] = qps / (sample / secs)
    if rank == 0:
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)        # RCCL's start-up banner sits in the C stdio buffer: out it goes (to stderr) before the result line
        except Exception:
            pass
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
